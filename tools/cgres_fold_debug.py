"""Debug aid (GPU): the on-chip resident C-grid kernel on the tripole fixtures, forced on; per field the number of cells whose
bits differ from the reference's arrays and where the first few are.  python tools/cgres_fold_debug.py [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import oracle  # noqa: E402
from common import GoldenCase  # noqa: E402
from test_gpu_cgrid import cgrid_core  # noqa: E402
from cice_amd import evp  # noqa: E402


def main():
    names = sys.argv[1:] or ["cgrid_trip_2x2_full", "cgrid_trip_1blk_patchy_avgstrength"]
    os.environ["CICE_EVP_HIP_VERBOSE"] = "1"
    for name in names:
        c = GoldenCase(name)
        dom = c.oracle_domain()
        for forced in ("1",):
            os.environ["CICE_EVP_HIP_CGRID_RESIDENT"] = forced
            core = cgrid_core(c)
            try:
                for icall in range(1, c.ncalls + 1):
                    state, inputs, masks = c.cgrid_inputs(icall)
                    for nsub in c.nsub_list:
                        try:
                            out = core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
                        except evp.EvpHipError as e:
                            print(name, "REFUSED:", e)
                            break
                        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
                        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
                        want = c.cgrid_expected(icall, nsub)
                        res = core.cgrid_timings()["resident_subcycles"]
                        bad = 0
                        for k in sorted(want):
                            if k not in out:
                                continue
                            a = np.ascontiguousarray(out[k]).view(np.uint64)
                            b = np.ascontiguousarray(want[k]).view(np.uint64)
                            d = np.argwhere(a != b)
                            if len(d):
                                bad += 1
                                first = [tuple(int(x) + (0 if q == 0 else 1) for q, x in enumerate(r)) for r in d[:6]]
                                rows = sorted(set(int(r[1]) + 1 for r in d))
                                print(f"  {name} call {icall} nsub {nsub} res {res}: {k}: {len(d)} differ, rows {rows[:8]}, first (blk, j, i) {first}")
                        print(f"{name} call {icall} nsub {nsub} resident {res}: {'OK' if not bad else str(bad) + ' fields differ'}")
                    else:
                        continue
                    break
            finally:
                core.finalize()


if __name__ == "__main__":
    main()
