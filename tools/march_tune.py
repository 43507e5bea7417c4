"""Two-subcycle marching kernel on 3600x2400 (or argv[1]: a workload name or NXxNY): us per subcycle for a list of settings.
   python tools/march_tune.py [s01] "SEG=48 ORDER=0" "SEG=96 ORDER=1" ...   (CICE_EVP_HIP_MARCH_<key>)"""
import sys, os, time
import pathlib; R=str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0]=[R, R+'/tests', R+'/oracle']
import numpy as np
from cice_amd import evp, synth, decomp
args = sys.argv[1:]
import re
if args and re.fullmatch(r"\d+x\d+", args[0]):          # any size, 0.1-degree-class spacing
    wl = args.pop(0)
    spec = dict(nx=int(wl.split("x")[0]), ny=int(wl.split("x")[1]), dx0=2.8e4)
else:
    wl = args.pop(0) if args and args[0] in synth.GRIDS else "s01"
    spec = synth.GRIDS[wl]
nx, ny = spec["nx"], spec["ny"]
g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
st = synth.make_state(g, case="full", seed=20260928, warm=True)
dc = decomp.Decomp(nx, ny, nx, ny, "cyclic", "closed", 1)
geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
tm = dc.scatter(st["iceTmask"], 0, fill=0); um = dc.scatter(st["iceUmask"], 0, fill=0)
scal = synth.evp_scalars(480)
ndte = int(os.environ.get("TUNE_NDTE", "48"))
ref = None
for setting in (args or ["SEG=0"]):
    kv = dict(x.split("=") for x in setting.split())
    for k, v in kv.items():
        os.environ[("CICE_EVP_HIP_" if k in ("MARCH", "RESIDENT") else "CICE_EVP_HIP_MARCH_") + k] = v
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep)
    core.upload(fields, tm, um)
    core.subcycle(ndte); core.sync()
    ts = []
    for _ in range(int(os.environ.get("TUNE_REPS", "7"))):
        t0 = time.perf_counter()
        core.subcycle(ndte); core.sync()
        ts.append(time.perf_counter() - t0)
    t = 3 * float(np.median(ts))
    tmin = min(ts)
    out = core.download()
    cs = float(np.abs(out["uvel"]).sum())
    if ref is None: ref = cs
    print("RESULT", wl, setting, "us/subcycle median %.1f min %.1f" % (1e6 * t / (3 * ndte), 1e6 * tmin / ndte), "same" if cs == ref else "DIFFERENT", core.march_info(), flush=True)
    core.finalize()
    for k in kv:
        os.environ.pop(("CICE_EVP_HIP_" if k in ("MARCH", "RESIDENT") else "CICE_EVP_HIP_MARCH_") + k, None)
