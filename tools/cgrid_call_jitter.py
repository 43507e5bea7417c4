#!/usr/bin/env python3
"""Per-call host wall time of cice_evp_hip_cgrid_run on gx1 (the bench's `host_prepared_cgrid_run`), every call printed:
finds which call of a series is slow and whether the loop (device) or the copies (host) took the time.

  python tools/cgrid_call_jitter.py [calls [gc]]      gc=0: Python's garbage collector off during the series
"""
import ctypes as C
import gc
import sys
import time
from pathlib import Path

R = str(Path(__file__).resolve().parents[1])
sys.path[:0] = [R, R + "/tests", R + "/oracle"]
import numpy as np
from cice_amd import decomp, evp, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if len(sys.argv) > 2 and sys.argv[2] == "0":
    gc.disable()
ndte = 120
spec = synth.GRIDS["gx1"]
nx, ny = spec["nx"], spec["ny"]
g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
cg = synth.cgrid_geometry(g)
state, inputs, masks = synth.cgrid_state(g, cg, case="full", seed=20260928, warm=True)
dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", "closed")
static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
own = lambda a, dt=np.float64: np.array(a, dtype=dt, order="C", copy=True)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(ndte), strict=True), static["dyE"], static["dxN"], static["dxT"],
                  static["dyT"], 1.0 / static["uarea"], static["tarea"], keepalive=keep)
try:
    core.cgrid_set_geometry(static)
    work = {k: (own(state[k]) if k in state else np.zeros(core.shape)) for k in evp.CGRID_FIELDS}
    inp = {k: own(inputs[k]) for k in evp.CGRID_INPUTS}
    mk = {k: own(masks[k], np.int32) for k in evp.CGRID_MASKS}
    core.pin_host(*work.values(), *inp.values())
    ftab = (evp._f64p * 19)(*[evp._dp(work[k]) for k in evp.CGRID_FIELDS])
    itab = (evp._f64p * 23)(*[evp._dp(inp[k]) for k in evp.CGRID_INPUTS])
    mtab = [evp._ip(mk[k]) for k in evp.CGRID_MASKS]
    L = core.lib
    rows = []
    for k in range(calls):
        t0 = time.perf_counter()
        evp._check(L, L.cice_evp_hip_cgrid_run(ndte, 0, ftab, itab, *mtab), "cgrid_run")
        ms = 1e3 * (time.perf_counter() - t0)
        rows.append((ms, core.cgrid_timings()["loop_ms"]))
    for k, (ms, loop) in enumerate(rows):
        print(f"call {k:3d}: {ms:8.3f} ms   loop {loop:7.3f} ms" + ("   <--" if ms > 2 * np.median([r[0] for r in rows]) else ""))
finally:
    core.finalize()
