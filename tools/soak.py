"""Soak test of the tagged-record hand-off: the on-chip resident kernel must reproduce the streaming
kernel's result bit for bit on EVERY repetition (a torn or stale record would show up as a
difference or as a time-out).  python tools/soak.py [gx1|tx1|gx3] [reps]"""
import sys, time, os, pathlib
R = str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + '/tests', R + '/oracle']
import numpy as np
from cice_amd import evp, synth, decomp
wl = sys.argv[1] if len(sys.argv) > 1 else "gx1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
case = sys.argv[3] if len(sys.argv) > 3 else "full"       # "caps": ice edges (tiles next to open water)
bs = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else None
spec = synth.GRIDS[wl]; nx, ny = spec["nx"], spec["ny"]; ns = spec.get("ns", "closed")
g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=ns))
st = synth.make_state(g, case=case, seed=5, warm=True)
dc = decomp.single_block(nx, ny, "cyclic", ns) if bs is None else decomp.Decomp(nx, ny, bs[0], bs[1], "cyclic", ns, 1)
geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
tm = dc.scatter(st["iceTmask"], 0, fill=0); um = dc.scatter(st["iceUmask"], 0, fill=0)
scal = synth.evp_scalars(120)
def make(resident):
    os.environ["CICE_EVP_HIP_RESIDENT"] = "1" if resident else "0"
    d, keep = evp.make_dims(dc, 0)
    return evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
core = make(False)
core.upload(fields, tm, um); core.subcycle(120); want = core.download(); core.finalize()
core = make(True)
core.upload(fields, tm, um)
assert core.timings()["tile_variant"] >= 1000   # gen 1 (10xx) or gen 2 (20xx) resident kernel
bad = 0
t0 = time.time()
for r in range(reps):
    core.upload(fields, tm, um) if r % 50 == 0 else None     # mostly back-to-back launches from the evolving state
    if r % 50 == 0:
        core.subcycle(120)
        got = core.download()
        for k in ("uvel", "vvel", "stressp_1", "stress12_4"):
            if not np.array_equal(got[k], want[k]):
                bad += 1
    else:
        core.subcycle(120)
core.sync()
print("SOAK", wl, case, bs, "reps", reps, "subcycles", reps * 120, "mismatches", bad, "seconds %.1f" % (time.time() - t0), "variant", core.timings()["tile_variant"])
core.finalize()
sys.exit(1 if bad else 0)
