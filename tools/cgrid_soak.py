#!/usr/bin/env python3
"""Soak of the C-grid loop's schedules: the on-chip resident kernel (round 5) and the one-launch kernel (five arrays ping-pong, buffers change roles whenever a call
runs an odd number of subcycles) must reproduce the three-launch schedule bit for bit on EVERY repetition of a long
sequence of calls with varying subcycle counts, uploads in between and both visc_methods.
  python tools/cgrid_soak.py [gx3|p2|gx1|tx1] [reps]"""
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
    ns = spec.get("ns", "closed")               # (tx1: tripole -- five phases + fold steps against the resident kernel's FOLD variant)
    g = synth.derive_geometry(synth.make_grid(spec["nx"], spec["ny"], spec["dx0"], ns=ns))
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case="full", seed=3, warm=True)
    nb_ = int(os.environ.get("SOAK_BLOCKS", "2"))
    dc = decomp.Decomp(spec["nx"], spec["ny"], -(-spec["nx"] // nb_), -(-spec["ny"] // nb_), "cyclic", ns, 1)
    static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
    scal = synth.evp_scalars(120)
    rng = np.random.default_rng(1)
    plan = []                                   # (upload?, visc, subcycles)
    for r in range(reps):
        plan.append((r == 0 or rng.random() < 0.2, "avg_strength" if (r // 40) % 2 else "avg_zeta", int(rng.choice([1, 2, 3, 5, 8, 13, 120]))))

    def run(one, resident="0"):
        os.environ["CICE_EVP_HIP_CGRID_ONE"] = one
        os.environ["CICE_EVP_HIP_CGRID_RESIDENT"] = resident
        if resident != "0":
            del os.environ["CICE_EVP_HIP_CGRID_RESIDENT"]        # the library's own choice (probe at the first eligible call)
        d, keep = evp.make_dims(dc, 0)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        sums, n_one, n_res = [], 0, 0
        try:
            core.cgrid_set_geometry(static)
            visc_now = None
            for up, visc, n in plan:
                if up or visc != visc_now:
                    core.cgrid_upload(state, inputs, masks, visc_method=visc)
                    visc_now = visc
                core.cgrid_subcycle(n)
                n_one += core.cgrid_timings()["one_launch_subcycles"]
                n_res += core.cgrid_timings()["resident_subcycles"]
                out = core.cgrid_download()
                sums.append(tuple(out[k].tobytes() for k in ("uvelE", "vvelN", "stresspT", "stress12T", "stress12U", "shearU", "zetax2T")))
        finally:
            core.finalize()
        return sums, n_one, n_res

    t0 = time.time()
    a, na, _ = run("0")
    b, nb, _ = run("1")
    c, nc, nr = run("1", "auto")                 # round 5: the on-chip resident kernel where it is eligible, mixed with the others
    bad = sum(1 for x, y in zip(a, b) if x != y) + sum(1 for x, y in zip(a, c) if x != y)
    ok = bad == 0 and (nb > 0 or ns == "tripole") and na == 0 and nr > 0
    print(f"CGRID_SOAK {grid}: {reps} calls, {sum(p[2] for p in plan)} subcycles, {nb} of them as one launch ({na} with the switch off); third run: "
          f"{nr} inside the resident kernel + {nc} as one launch; {bad} calls differ, {time.time() - t0:.1f} s: {'OK' if ok else 'FAILED'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
