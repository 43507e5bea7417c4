"""Debug aid: which cells of the tripole fixtures differ from the reference only in the sign of zero, per kernel path."""
import os, sys, pathlib
R = str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + "/tests", R + "/oracle"]
import numpy as np
from common import GoldenCase
from test_gpu_parity import hip_from_case, post_evp
for envs in ({"CICE_EVP_HIP_SEAM_FIN": "1", "CICE_EVP_HIP_RESIDENT": "0"}, {"CICE_EVP_HIP_RESIDENT": "0"}, {}):
    for k in ("CICE_EVP_HIP_SEAM_FIN", "CICE_EVP_HIP_RESIDENT"):
        os.environ.pop(k, None)
    os.environ.update(envs)
    c = GoldenCase("trip_cyc_2x2_full")
    core = hip_from_case(c, strict=True)
    try:
        for icall in range(1, c.ncalls + 1):
            dyn, tm, um = c.inputs(icall)
            for nsub in c.nsub_list:
                out = post_evp(c, core.run(dyn, tm, um, ndte=nsub))
                want = c.expected(icall, nsub)
                for f in ("uvel", "vvel"):
                    g, w = out[f], want[f]
                    ne = np.argwhere(g.view(np.uint64) != w.view(np.uint64))
                    print(envs, "call", icall, "nsub", nsub, f, "differ:", [(tuple(q), float(g[tuple(q)]), float(w[tuple(q)]), bool(np.signbit(g[tuple(q)])), bool(np.signbit(w[tuple(q)])),
                                                                           int(um[tuple(q)]), float(dyn[f][tuple(q)]), bool(np.signbit(dyn[f][tuple(q)]))) for q in ne[:8]])
    finally:
        core.finalize()
