#!/usr/bin/env python3
"""C-grid EVP subcycle on the GPU: microseconds per subcycle on the named synthetic grids (HIP events around the
captured loop).  usage: tools/cgrid_timing.py [gx1 s01 ...] [--ndte 120] [--reps 5]"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from cice_amd import decomp, evp, synth  # noqa: E402

import os  # noqa: E402
if os.environ.get("EVP_TIMING_LIB"):          # A/B on one box: time another build of the library (e.g. the previous commit's)
    evp.LIB_PATH = Path(os.environ["EVP_TIMING_LIB"]).resolve()


def case(grid, case_="full", bs=None, seabed=False):
    spec = synth.GRIDS[grid]
    ns = spec.get("ns", "closed")
    g = synth.derive_geometry(synth.make_grid(spec["nx"], spec["ny"], spec["dx0"], ns=ns))
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case=case_, seed=3, seabed=seabed)
    bx, by = bs if bs else (spec["nx"], spec["ny"])
    dc = decomp.Decomp(spec["nx"], spec["ny"], bx, by, "cyclic", ns, 1)
    return (dc,) + synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("grids", nargs="*", default=["gx1"])
    ap.add_argument("--ndte", type=int, default=120)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--case", default="full")
    ap.add_argument("--visc", default="avg_zeta", choices=["avg_zeta", "avg_strength"])
    ap.add_argument("--bs", default="", help="block size BXxBY (default: one block)")
    ap.add_argument("--seabed", action="store_true", help="seabed stress on shallow cells (the resident kernel's SLOW variant)")
    a = ap.parse_args()
    for grid in a.grids:
        dc, static, state, inputs, masks = case(grid, a.case, tuple(int(v) for v in a.bs.split("x")) if a.bs else None, a.seabed)
        d, keep = evp.make_dims(dc, 0)
        scal = synth.evp_scalars(a.ndte)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        try:
            core.cgrid_set_geometry(static)
            core.cgrid_upload(state, inputs, masks, visc_method=a.visc)
            ts = []
            for r in range(a.reps + 1):
                core.cgrid_subcycle(a.ndte)
                core.cgrid_sync()
                ts.append(core.cgrid_timings()["loop_ms"])
            best = min(ts[1:])
            ncell = dc.nx_global * dc.ny_global
            nact = int(masks["iceTmask"].sum())
            tt = core.cgrid_timings()
            print(f"CGRID {grid} {a.case} {a.visc}{' seabed' if a.seabed else ''} one_launch={tt['one_launch_subcycles']} resident={tt['resident_subcycles']} windows_with_ice={tt['resident_windows_with_ice']}/{tt['resident_windows']} geometry_derived={tt['geometry_derived']} marched={tt['marched_items']} items x {tt['marched_segment_rows']} rows ({tt['marched_cells']} cells; {tt['marched_edge_windows']} windows beside): {best * 1e3 / a.ndte:.2f} us/subcycle (best of {a.reps}; first {ts[0] * 1e3 / a.ndte:.2f}), "
                  f"{ncell / (best * 1e-3 / a.ndte):.3e} cell-updates/s, active T {nact}/{ncell}", flush=True)
        finally:
            core.finalize()


if __name__ == "__main__":
    main()
