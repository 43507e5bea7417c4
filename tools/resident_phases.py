"""Where a subcycle of the on-chip resident kernel goes, per wave: shader-cycle stamps collected under
CICE_EVP_HIP_RES_PROF=1 (16 x 16 tiles).  Usage: python tools/resident_phases.py [gx1|gx3|p2] [ndte]"""
import os, sys, pathlib
R = str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + "/tests", R + "/oracle"]
os.environ["CICE_EVP_HIP_RES_PROF"] = "1"
os.environ.setdefault("CICE_EVP_HIP_RESIDENT", "1"); os.environ.setdefault("CICE_EVP_HIP_RES_GEN", "2")
os.environ.setdefault("CICE_EVP_HIP_RES_LOGW", "4"); os.environ.setdefault("CICE_EVP_HIP_TYB", "4")
import numpy as np
from cice_amd import evp, synth
from test_gpu_parity import synth_case
wl = sys.argv[1] if len(sys.argv) > 1 else "gx1"
ndte = int(sys.argv[2]) if len(sys.argv) > 2 else 120
spec = synth.GRIDS[wl]
scal = synth.evp_scalars(120)
dc, geo, fields, tm, um = synth_case(wl, "full", seed=1, warm=True)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep, testing=True)
core.upload(fields, tm, um)
for _ in range(3): core.subcycle(ndte)
core.sync()
p = core.debug_prof().astype(np.float64)
tt = core.timings()
core.finalize()
used = p[:, :, :5].sum(axis=(1, 2)) > 0
p = p[used]
ph = p[:, :, :5] / ndte                     # cycles per subcycle
names = ["poll", "stress", "wait B1", "stepu+publish", "wait B2"]
nact = p[:, 0, 7].astype(int)
rank = p[:, 0, 5].astype(int)
print("RESULT", wl, "tiles", len(p), "us/subcycle (event)", 1e3 * tt["loop_ms"] / ndte)
def show(sel, what):
    if not sel.any(): return
    late, early = ph[sel][:, 0, :], ph[sel][:, 1:, :].reshape(-1, 5)
    print(f"  {what}: {int(sel.sum())} tiles; total cycles/subcycle rim wave {late.sum(axis=1).mean():.0f}")
    print("     rim wave      " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, late.mean(axis=0))))
    print("     other waves   " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, early.mean(axis=0))))
show(nact == 4, "full tiles")
for r in range(3):
    show((nact == 4) & (rank == r), f"full tiles, arrival rank {r} on their CU")
show((nact > 0) & (nact < 4), "partial tiles")
show(nact == 0, "ice-free tiles")
