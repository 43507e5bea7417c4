#!/usr/bin/env python3
"""Option A: what one evp() costs when the preparation phase runs on the device too --
cice_evp_hip_prep (11 T-grid fields + velocities in, stresses resident) + set_strength + ndte subcycles + download of
the 6 outputs; host wall time per call and the library's own split (H2D, preparation kernels, loop)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from cice_amd import decomp, evp, synth  # noqa: E402


def main():
    spec = synth.GRIDS["gx1"]
    nx, ny = spec["nx"], spec["ny"]
    g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
    pr = synth.make_primary(g, "full", seed=9)
    st = synth.make_state(g, case="full", seed=7, warm=True)
    dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", "closed")
    sc = lambda x, fill=0.0: np.ascontiguousarray(dc.scatter(np.ascontiguousarray(x), 0, fill=fill))
    geo = {k: sc(g[k], 1.0 if k != "uarear" else 0.0) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(120), strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    lib = core.lib
    try:
        static = {k: sc(v, (1.0 if k in ("tarea", "uarea") else 0)) for k, v in pr["static"].items()}
        core.set_prep_geometry(*[static[k] for k in ("tmask", "umask", "hm", "tarea", "uarea", "fcor_blk")])
        pp = evp.PrepParams(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11, dyn_mass_min=1e-10,
                            ssh_stress_coupled=0)
        t = {k: sc(v) for k, v in pr["t"].items()}
        state = {k: sc(v) for k, v in pr["state"].items()}
        strength = sc(st["strength"])
        out = {k: np.zeros(core.shape) for k in ("strintxU", "strintyU", "taubxU", "taubyU", "uvel", "vvel")}
        core.pin_host(*t.values(), state["uvel"], state["vvel"], strength, *out.values(), *[state[k] for k in evp.FIELDS[:12]])
        f64p = C.POINTER(C.c_double)
        dp = lambda a: a.ctypes.data_as(f64p)
        ttab = (f64p * 11)(*[dp(t[k]) for k in evp.PREP_T])
        first = (f64p * len(evp.FIELDS))(*[(dp(state[k]) if k in state and k != "iceUmask" else None) for k in evp.FIELDS])
        later = (f64p * len(evp.FIELDS))(*[(dp(state[k]) if k in ("uvel", "vvel") else None) for k in evp.FIELDS])
        tm = np.zeros(core.shape, np.int32)
        um = np.ascontiguousarray(state["iceUmask"], np.int32)
        otab = (f64p * len(evp.FIELDS))(*[(dp(out[k]) if k in out else None) for k in evp.FIELDS])
        i32p = C.POINTER(C.c_int32)

        def call(tab):
            parts = []
            t0 = time.perf_counter()
            rc = lib.cice_evp_hip_prep(C.byref(pp), ttab, tab, tm.ctypes.data_as(i32p), um.ctypes.data_as(i32p), None, None, None, None)
            assert rc == 0, rc
            parts.append(time.perf_counter())
            assert lib.cice_evp_hip_set_strength(dp(strength)) == 0
            parts.append(time.perf_counter())
            assert lib.cice_evp_hip_subcycle(C.c_int32(120)) == 0
            assert lib.cice_evp_hip_download(otab) == 0
            parts.append(time.perf_counter())
            return [1e3 * (b - a) for a, b in zip([t0] + parts[:-1], parts)]

        call(first)
        call(later)
        ts = np.array([call(later) for _ in range(12)])
        med = np.median(ts, axis=0)
        tt = core.timings()
        print(f"PREPCALL gx1: {med.sum():.3f} ms per evp() call = prep {med[0]:.3f} + set_strength {med[1]:.3f} + loop/download {med[2]:.3f} "
              f"(library: H2D {tt['h2d_ms']:.3f}, preparation kernels {tt['prep_ms']:.3f}, loop {tt['loop_ms']:.3f}, D2H {tt['d2h_ms']:.3f} ms)")
    finally:
        core.finalize()


if __name__ == "__main__":
    main()
