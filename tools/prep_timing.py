"""Cost of evp()'s preparation phase on the device vs on the host, gx1 size:
the whole cice_evp_hip_prep call (copies included), its device part, and the oracle's C
restatement of the same phase on one host core (what the reference does in Fortran)."""
import sys, time, pathlib
R = str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + '/tests', R + '/oracle']
import numpy as np
import oracle
from cice_amd import evp, synth, decomp
wl = sys.argv[1] if len(sys.argv) > 1 else "gx1"
spec = synth.GRIDS[wl]; nx, ny = spec["nx"], spec["ny"]
g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
pr = synth.make_primary(g, "full", seed=4)
dc = decomp.single_block(nx, ny, "cyclic", "closed")
sc = lambda a, fill=0.0: dc.scatter(np.ascontiguousarray(a), 0, fill=fill)
geo = {k: sc(g[k], 1.0 if k != "uarear" else 0.0) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
static = {k: sc(v, (1.0 if k in ("tarea", "uarea") else 0)) for k, v in pr["static"].items()}
t = {k: sc(v) for k, v in pr["t"].items()}
state = {k: sc(v) for k, v in pr["state"].items()}
scal = synth.evp_scalars(120)
ppd = dict(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11, dyn_mass_min=1e-10)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                  geo["uarear"], geo["tarea"], keepalive=keep)
core.set_prep_geometry(*[static[k] for k in ("tmask", "umask", "hm", "tarea", "uarea", "fcor_blk")])
pp = evp.PrepParams(**ppd, ssh_stress_coupled=0)
for a in list(t.values()) + [state[k] for k in evp.FIELDS[:12]] + [state["uvel"], state["vvel"]]:
    core.pin_host(a)
core.prep(pp, t, state)
t0 = time.perf_counter()
for _ in range(10):
    core.prep(pp, t, state)
t_call = (time.perf_counter() - t0) / 10
tim = core.timings()
blks = dc.local_blocks(0)
dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, 1, nx, ny, dc.ew, dc.ns, [b.ilo for b in blks], [b.ihi for b in blks],
                          [b.jlo for b in blks], [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
z = np.zeros(dc.shape(0))
st2 = dict(state, strintxU=z, strintyU=z, strocnxU=z, strocnyU=z)
opp = oracle.PrepParams(**ppd, cosw=1.0, sinw=0.0, ssh_coupled=0)
oracle.prep(dom, opp, static, t, st2)
t0 = time.perf_counter()
for _ in range(5):
    oracle.prep(dom, opp, static, t, st2)
t_cpu = (time.perf_counter() - t0) / 5
# whole evp() call through the wider entry points, PCIe included: prep -> strength -> loop -> download
strength = sc(np.where(pr["static"]["tmask"] != 0, 2.75e4 * pr["t"]["vice"] * np.exp(-20.0 * (1.0 - pr["t"]["aice"])), 0.0))
core.pin_host(strength)
outbuf = {k: np.zeros(dc.shape(0)) for k in evp.OUTPUTS}
core.pin_host(*outbuf.values())
def whole(resident):
    st = {k: v for k, v in state.items() if not (resident and k in evp.FIELDS[:12])}
    dst = {k: v for k, v in outbuf.items() if not (resident and k in evp.FIELDS[:12])}
    core.prep(pp, t, state); core.set_strength(strength); core.subcycle(120); core.download_into(outbuf)   # warm, stresses on device
    t0 = time.perf_counter()
    for _ in range(10):
        core.prep(pp, t, st)
        core.set_strength(strength)
        core.subcycle(120)
        core.download_into(dst)
    return (time.perf_counter() - t0) / 10
t_full, t_res = whole(False), whole(True)
print("EVPCALL", wl, "prep+loop+download ms: all arrays %.3f, stresses resident %.3f" % (1e3 * t_full, 1e3 * t_res))
print("PREP", wl, "call_ms %.3f" % (1e3 * t_call), "h2d_ms %.3f" % tim["h2d_ms"], "device_ms %.3f" % tim["prep_ms"],
      "oracle_1core_ms %.2f" % (1e3 * t_cpu))
core.finalize()
