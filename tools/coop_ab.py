"""A/B on one box, one process: the resident B-grid kernel with its rim T-cells updated by one thread each (the default) and by
four lanes each (COOP, cice_amd/csrc/evp_resident2.hip) -- CICE_EVP_HIP_RES_COOP is read at every launch of the test build.
Same state, same number of subcycles, alternating rounds; outputs compared bit for bit, loop times by HIP events.

    python tools/coop_ab.py [gx1|tx1|gx3] [rounds] [ndte]"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("CICE_EVP_HIP_RES_LOGW", "4")
os.environ["CICE_EVP_HIP_RESIDENT"] = "1"
os.environ["CICE_EVP_HIP_RES_COOP"] = "0"
from cice_amd import decomp, evp, synth  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "gx1"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ndte = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    spec = synth.GRIDS[wl]
    nx, ny = spec["nx"], spec["ny"]
    ns = spec.get("ns", "closed")
    g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=ns))
    st = synth.make_state(g, case="full", seed=7, warm=True)
    dc = decomp.Decomp(nx, ny, nx, ny, "cyclic", ns, 1)
    fold = g if ns != "closed" else None
    geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k in ("HTE", "HTN", "dxT", "dyT", "tarea") else 0.0))
           for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
    fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
    tm, um = dc.scatter(st["iceTmask"], 0, fill=0), dc.scatter(st["iceUmask"], 0, fill=0)
    scal = synth.evp_scalars(ndte)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    try:
        res = {"0": [], "1": []}
        outs = {}
        for r in range(rounds):
            for coop in ("0", "1"):
                os.environ["CICE_EVP_HIP_RES_COOP"] = coop
                out = core.run(fields, tm, um, ndte=ndte)
                t = core.timings()
                res[coop].append(t["loop_ms"])
                if r == 0:
                    outs[coop] = {k: np.array(v, copy=True) for k, v in out.items()}
                assert t["tile_variant"] >= 2000 and t["resident_fallbacks"] == 0, t
        bad = [k for k in outs["0"] if not np.array_equal(np.ascontiguousarray(outs["0"][k]).view(np.uint64),
                                                          np.ascontiguousarray(outs["1"][k]).view(np.uint64))]
        for coop in ("0", "1"):
            v = np.array(res[coop][1:] or res[coop])
            print(f"COOP={coop}: loop ms median {np.median(v):.4f} min {v.min():.4f}  -> {1e3 * np.median(v) / ndte:.3f} us per subcycle")
        print("bitwise identical" if not bad else f"DIFFER: {bad}", "| max|u|", float(np.abs(outs['0']['uvel']).max()))
        if bad:
            k = bad[0]
            dd = outs["0"][k] != outs["1"][k]
            print(k, int(dd.sum()), "cells differ; first:", np.argwhere(dd)[:5].tolist(),
                  float(np.abs(outs["0"][k] - outs["1"][k]).max()))
        return 1 if bad else 0
    finally:
        core.finalize()


if __name__ == "__main__":
    sys.exit(main())
