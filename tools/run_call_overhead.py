"""Per-call cost of cice_evp_hip_run (H2D + ndte subcycles + D2H) on gx1-sized arrays,
pageable vs page-locked host memory."""
import sys, time, pathlib
R = str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + "/tests", R + "/oracle"]
import numpy as np
from cice_amd import evp, synth
from test_gpu_parity import synth_case
scal = synth.evp_scalars(120)
dc, geo, fields, tm, um = synth_case("gx1", "full", seed=1, warm=True)
d, keep = evp.make_dims(dc, 0)
for pin, resident in ((False, False), (True, False), (True, True)):
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep)
    work = {k: np.array(fields[k], dtype=np.float64, order="C", copy=True) for k in evp.FIELDS}
    tmc, umc = np.ascontiguousarray(tm, np.int32), np.ascontiguousarray(um, np.int32)
    if pin:
        core.pin_host(*work.values())
    core.set_option(evp.OPT_STRESS_RESIDENT, 1 if resident else 0)
    core.run_inplace(work, tmc, umc, 120)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        core.run_inplace(work, tmc, umc, 120)
    t = (time.perf_counter() - t0) / n
    tt = core.timings()
    print(f"RESULT pinned={pin} stresses_resident={resident}: {1e3*t:.2f} ms per evp call  (H2D {tt['h2d_ms']:.2f} ms, subcycles {tt['loop_ms']:.2f} ms, D2H {tt['d2h_ms']:.2f} ms)")
    core.finalize()
