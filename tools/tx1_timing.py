"""tx1-sized tripole grid (360x240) on one GPU: per-subcycle time of the streaming path with the
seam kernel, for DESIGN.md.  Synthetic state (cice_amd.synth), ndte=240."""
import sys, time, pathlib
R = str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + '/tests', R + '/oracle']
import numpy as np
from cice_amd import evp, synth, decomp
spec = synth.GRIDS["tx1"]; nx, ny = spec["nx"], spec["ny"]
g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="tripole"))
st = synth.make_state(g, case="full", seed=3, warm=True)
bs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (nx, ny)
dc = decomp.Decomp(nx, ny, bs[0], bs[1], "cyclic", "tripole", 1)
geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
tm = dc.scatter(st["iceTmask"], 0, fill=0); um = dc.scatter(st["iceUmask"], 0, fill=0)
scal = synth.evp_scalars(240)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep)
core.upload(fields, tm, um)
for _ in range(2): core.subcycle(240)
core.sync()
t0 = time.perf_counter()
for _ in range(10): core.subcycle(240)
core.sync()
t = time.perf_counter() - t0
tt = core.timings()
print("TX1", bs, "us/subcycle %.2f" % (1e6 * t / 2400), "cell-updates/s %.3g" % (nx * ny * 2400 / t), "launches/subcycle", tt["launches_per_subcycle"], "variant", tt["tile_variant"])
core.finalize()
