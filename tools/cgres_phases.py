#!/usr/bin/env python3
"""Where a subcycle of the on-chip resident C-grid kernel (cg_res) goes, per wave: shader cycles summed over a launch
(test build, CICE_EVP_HIP_CGRID_PROF=1).   python tools/cgres_phases.py [gx1|gx3] [--ndte 120]"""
import argparse, os, sys
from pathlib import Path
os.environ["CICE_EVP_HIP_CGRID_PROF"] = "1"
os.environ.setdefault("CICE_EVP_HIP_CGRID_RESIDENT", "1")
R = str(Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + "/tests", R + "/oracle"]
import numpy as np
from cice_amd import decomp, evp, synth
ap = argparse.ArgumentParser(); ap.add_argument("grid", nargs="?", default="gx1"); ap.add_argument("--ndte", type=int, default=120)
a = ap.parse_args()
spec = synth.GRIDS[a.grid]
ns = spec.get("ns", "closed")
g = synth.derive_geometry(synth.make_grid(spec["nx"], spec["ny"], spec["dx0"], ns=ns))
cg = synth.cgrid_geometry(g)
state, inputs, masks = synth.cgrid_state(g, cg, case="full", seed=3)
dc = decomp.Decomp(spec["nx"], spec["ny"], spec["nx"], spec["ny"], "cyclic", ns, 1)
static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(a.ndte), strict=True), static["dyE"], static["dxN"], static["dxT"],
                  static["dyT"], 1.0 / static["uarea"], static["tarea"], keepalive=keep, testing=True)
try:
    core.cgrid_set_geometry(static)
    core.cgrid_upload(state, inputs, masks)
    for _ in range(3):
        core.cgrid_subcycle(a.ndte); core.cgrid_sync()
    tt = core.cgrid_timings()
    P = core.debug_cgres_prof().astype(np.float64)
finally:
    core.finalize()
n = tt["resident_subcycles"]
print(f"{a.grid}: {len(P)} windows, {tt['loop_ms'] * 1e3 / a.ndte:.2f} us per subcycle, resident subcycles {n}")
names = ["poll", "bar0", "S", "bar1", "T", "bar2", "U+bar3", "C"]
P = P / max(n, 1)
for w in range(4):
    print(f" wave {w}: " + "  ".join(f"{nm} {np.median(P[:, w, k]):6.0f}" for k, nm in enumerate(names)) + f"   sum {np.median(P[:, w, :].sum(axis=1)):7.0f} cycles/subcycle")
if ns == "tripole":      # the windows at the fold (the last window row) against the rest
    nf = -(-spec["nx"] // 13)
    for label, Q in (("fold windows", P[-nf:]), ("other windows", P[:-nf])):
        for w in range(4):
            print(f" {label} wave {w}: " + "  ".join(f"{nm} {np.median(Q[:, w, k]):6.0f}" for k, nm in enumerate(names)) + f"   sum {np.median(Q[:, w, :].sum(axis=1)):7.0f}")
tot = P.sum(axis=2)
print(" all waves: median sum", np.median(tot), " (100 MHz?? no: shader clock; at 2.4 GHz 1 us = 2400 cycles)")
