#!/usr/bin/env python3
"""Where a window of the C grid's one-launch kernel (cg_one) spends its time: per-window cycle stamps of one wave with
owned cells (test build, CICE_EVP_HIP_CGRID_PROF=1; evp_cgrid.hip: mark()).

  python tools/cgrid_phases.py [gx1|s01|...] [--ndte 12]

Prints, over the windows of the LAST launch: the median / 10 % / 90 % length in shader-clock cycles of
  S     start -> first barrier reached       (level S: loads + strain rates)
  b1    waiting at the first barrier          (the slowest wave of the workgroup)
  T     first barrier -> second reached       (level T)
  b2    waiting at the second barrier
  C     second barrier -> level C arithmetic done (loads + stress12U at three corners + both momentum equations)
  st    -> end (the stores)
and how many windows a CU ran and how they overlapped in time (windows in flight per CU, from start / end stamps:
below 1 = the CU sits idle between two workgroups)."""
import argparse
import os
import sys
from pathlib import Path

os.environ["CICE_EVP_HIP_CGRID_PROF"] = "1"
R = str(Path(__file__).resolve().parents[1])
sys.path[:0] = [R, R + "/tests", R + "/oracle"]
import numpy as np
from cice_amd import decomp, evp, synth

ap = argparse.ArgumentParser()
ap.add_argument("grid", nargs="?", default="s01")
ap.add_argument("--ndte", type=int, default=12)
a = ap.parse_args()
spec = synth.GRIDS[a.grid]
g = synth.derive_geometry(synth.make_grid(spec["nx"], spec["ny"], spec["dx0"], ns="closed"))
cg = synth.cgrid_geometry(g)
state, inputs, masks = synth.cgrid_state(g, cg, case="full", seed=3)
dc = decomp.Decomp(spec["nx"], spec["ny"], spec["nx"], spec["ny"], "cyclic", "closed", 1)
static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(a.ndte), strict=True), static["dyE"], static["dxN"], static["dxT"],
                  static["dyT"], 1.0 / static["uarea"], static["tarea"], keepalive=keep, testing=True)
try:
    core.cgrid_set_geometry(static)
    core.cgrid_upload(state, inputs, masks)
    core.cgrid_subcycle(a.ndte)
    core.cgrid_sync()
    core.cgrid_subcycle(a.ndte)
    core.cgrid_sync()
    tt = core.cgrid_timings()
    P = core.debug_cgrid_prof().astype(np.int64)
finally:
    core.finalize()
P = P[P[:, 0] > 0]
print(f"{a.grid}: {len(P)} windows stamped, {tt['loop_ms'] * 1e3 / a.ndte:.1f} us per subcycle, geometry_derived={tt['geometry_derived']}")
names = ["S", "b1", "T", "b2", "C", "st"]
for k, n in enumerate(names):
    dt = P[:, k + 1] - P[:, k]
    print(f"  {n:3s} median {int(np.median(dt)):7d}   10% {int(np.percentile(dt, 10)):7d}   90% {int(np.percentile(dt, 90)):7d} cycles")
tot = P[:, 6] - P[:, 0]
print(f"  window start -> end: median {int(np.median(tot))}, 10% {int(np.percentile(tot, 10))}, 90% {int(np.percentile(tot, 90))}")
hw = P[:, 7]
cu = ((hw >> 32) & 15) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 15)       # XCC, SE, CU
ids, cnt = np.unique(cu, return_counts=True)
print(f"  CUs seen {len(ids)}, windows per CU min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}")
# windows in flight per CU: average over time
infl = []
for c in ids[:64]:
    w = P[cu == c]
    ev = np.concatenate([np.stack([w[:, 0], np.ones(len(w), np.int64)], 1), np.stack([w[:, 6], -np.ones(len(w), np.int64)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    lvl = np.cumsum(ev[:, 1])[:-1]
    dtv = np.diff(ev[:, 0])
    infl.append((lvl * dtv).sum() / max(1, dtv.sum()))
print(f"  windows in flight per CU, time average between its first window's start and its last one's end: {np.mean(infl):.2f}")
