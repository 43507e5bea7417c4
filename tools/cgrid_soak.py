#!/usr/bin/env python3
"""Soak of the C-grid loop's schedules: the one-launch kernel (five arrays ping-pong, buffers change roles whenever a call
runs an odd number of subcycles) must reproduce the three-launch schedule bit for bit on EVERY repetition of a long
sequence of calls with varying subcycle counts, uploads in between and both visc_methods.
  python tools/cgrid_soak.py [gx3|p2|gx1] [reps]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from cice_amd import decomp, evp, synth  # noqa: E402


def main():
    grid = sys.argv[1] if len(sys.argv) > 1 else "gx3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    spec = synth.GRIDS[grid]
    g = synth.derive_geometry(synth.make_grid(spec["nx"], spec["ny"], spec["dx0"], ns="closed"))
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case="full", seed=3, warm=True)
    dc = decomp.Decomp(spec["nx"], spec["ny"], -(-spec["nx"] // 2), -(-spec["ny"] // 2), "cyclic", "closed", 1)
    static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
    scal = synth.evp_scalars(120)
    rng = np.random.default_rng(1)
    plan = []                                   # (upload?, visc, subcycles)
    for r in range(reps):
        plan.append((r == 0 or rng.random() < 0.2, "avg_strength" if (r // 40) % 2 else "avg_zeta", int(rng.choice([1, 2, 3, 5, 8, 13, 120]))))

    def run(one):
        os.environ["CICE_EVP_HIP_CGRID_ONE"] = one
        d, keep = evp.make_dims(dc, 0)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        sums, n_one = [], 0
        try:
            core.cgrid_set_geometry(static)
            visc_now = None
            for up, visc, n in plan:
                if up or visc != visc_now:
                    core.cgrid_upload(state, inputs, masks, visc_method=visc)
                    visc_now = visc
                core.cgrid_subcycle(n)
                n_one += core.cgrid_timings()["one_launch_subcycles"]
                out = core.cgrid_download()
                sums.append(tuple(out[k].tobytes() for k in ("uvelE", "vvelN", "stresspT", "stress12T", "stress12U", "shearU", "zetax2T")))
        finally:
            core.finalize()
        return sums, n_one

    t0 = time.time()
    a, na = run("0")
    b, nb = run("1")
    bad = sum(1 for x, y in zip(a, b) if x != y)
    print(f"CGRID_SOAK {grid}: {reps} calls, {sum(p[2] for p in plan)} subcycles, {nb} of them as one launch ({na} with the switch off), "
          f"{bad} calls differ, {time.time() - t0:.1f} s: {'OK' if bad == 0 and nb > 0 and na == 0 else 'FAILED'}")
    return 0 if bad == 0 and nb > 0 and na == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
