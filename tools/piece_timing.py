#!/usr/bin/env python3
"""One per-rank piece of a split domain timed alone on one GPU (tools/predict_scaling.py's method, one row at a time):

  [CICE_EVP_HIP_... switches in the environment]  python tools/piece_timing.py NX NY [nsub [reps]] [--selfx] [--comm]

--comm: create the RCCL communicator with the rank itself (needed by CICE_EVP_HIP_MARCH_SELFX=1 and HALO=rccl);
--selfx: route every on-device ghost copy through the remote transport (CICE_EVP_HIP_SELF_EXCHANGE=1).
Prints one line: us per subcycle (wall, median of reps calls), tile variant, transport, marching-kernel info.
"""
import os
import sys
import time
from pathlib import Path

R = str(Path(__file__).resolve().parents[1])
sys.path[:0] = [R, R + "/tests", R + "/oracle"]
import numpy as np
from cice_amd import decomp, evp, synth

if os.environ.get("EVP_TIMING_LIB"):          # A/B on one box: time another build of the library
    evp.LIB_PATH = Path(os.environ["EVP_TIMING_LIB"]).resolve()

args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = {a for a in sys.argv[1:] if a.startswith("--")}
nx, ny = int(args[0]), int(args[1])
nsub = int(args[2]) if len(args) > 2 else 96
reps = int(args[3]) if len(args) > 3 else 5
if "--selfx" in flags:
    os.environ["CICE_EVP_HIP_SELF_EXCHANGE"] = "1"
g = synth.derive_geometry(synth.make_grid(nx, ny, 1.0e4, ns="closed"))
st = synth.make_state(g, case="full", seed=1, warm=True)
dc = decomp.Decomp(nx, ny, nx, ny, "cyclic", "closed", 1)
geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
tm, um = dc.scatter(st["iceTmask"], 0, fill=0), dc.scatter(st["iceUmask"], 0, fill=0)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(120), strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                  geo["uarear"], geo["tarea"], keepalive=keep)
try:
    if "--comm" in flags or "--selfx" in flags:
        core.comm_init(core.comm_unique_id())
    core.upload(fields, tm, um)
    core.subcycle(nsub)
    core.sync()
    each = []
    for _ in range(reps):
        t0 = time.perf_counter()
        core.subcycle(nsub)
        core.sync()
        each.append(1e6 * (time.perf_counter() - t0) / nsub)
    tt = core.timings()
    mi = core.march_info()
    out = core.download()
    import hashlib
    h = hashlib.sha256(np.ascontiguousarray(out["uvel"]).tobytes() + np.ascontiguousarray(out["stressp_1"]).tobytes()).hexdigest()[:12]
finally:
    core.finalize()
sw = " ".join(f"{k[13:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CICE_EVP_HIP_"))
print(f"PIECE {nx}x{ny} nsub={nsub} [{sw}]: {np.median(each):.2f} us/subcycle (min {min(each):.2f}) variant {tt['tile_variant']} "
      f"transport {tt['halo_transport']} march {mi.get('last_call')} strips {mi.get('strips')} seg {mi.get('segments')}x{mi.get('seglen')} "
      f"state {h}", flush=True)
