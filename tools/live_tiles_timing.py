"""What "only the tiles that hold ice run" buys the resident B-grid kernel: loop time per subcycle (HIP events, median of a few
calls of 120 subcycles) with the library's own choice against the streaming kernel forced, for ice everywhere and on the polar caps.
    python tools/live_tiles_timing.py"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["CICE_EVP_HIP_MARCH"] = "0"
from cice_amd import decomp, evp, synth  # noqa: E402


def one(nx, ny, dx0, case, resident):
    if resident is None:
        os.environ.pop("CICE_EVP_HIP_RESIDENT", None)
    else:
        os.environ["CICE_EVP_HIP_RESIDENT"] = resident
    g = synth.derive_geometry(synth.make_grid(nx, ny, dx0, ns="closed"))
    st = synth.make_state(g, case=case, seed=3, warm=True)
    dc = decomp.Decomp(nx, ny, nx, ny, "cyclic", "closed", 1)
    geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k in ("HTE", "HTN", "dxT", "dyT", "tarea") else 0.0))
           for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
    fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
    tm, um = dc.scatter(st["iceTmask"], 0, fill=0), dc.scatter(st["iceUmask"], 0, fill=0)
    scal = synth.evp_scalars(120)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    try:
        core.upload(fields, tm, um)
        ms = []
        for _ in range(8):
            core.subcycle(120)
            core.sync()
            ms.append(core.timings()["loop_ms"])
        t = core.timings()
        return 1e3 * float(np.median(ms[2:])) / 120, t["tile_variant"], t["resident_tiles_run"], t["resident_tiles"], float(tm.mean())
    finally:
        core.finalize()


for (nx, ny, dx0) in ((320, 384, 1.1e5), (560, 400, 5.0e4), (720, 540, 5.0e4)):
    for case in ("full", "caps"):
        a = one(nx, ny, dx0, case, None)
        b = one(nx, ny, dx0, case, "0")
        print(f"{nx}x{ny} {case:5s} (T-cells with ice {a[4]:.2f}): default {a[0]:7.2f} us per subcycle (variant {a[1]}, tiles {a[2]} of {a[3]})   "
              f"streaming forced {b[0]:7.2f} (variant {b[1]})")
