#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/<tag>/) to the tracked summary
bench.py quotes: per kernel/workload pair the average launch duration of the dominant kernel
(kernel trace), its hardware counters per launch (PMC passes), and what follows from them --
HBM-side bytes per launch (FETCH_SIZE corrected by the factor measured on known byte counts in the
same access width, tools/pmc_calib.hip; WRITE_SIZE likewise), VALU-busy SIMD cycles, effective clock.

  python tools/pmc_summary.py gpurun_out/<tag> profiles/r02_pmc_summary.json
"""
from __future__ import annotations

import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

N_SIMD = 1024            # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
N_XCD = 8
MAX_CLOCK_HZ = 2.4e9

DOMINANT = {"gx1res": "evp_resident", "gx1str": "evp_subcycle_tile", "s01str": "evp_subcycle_tile", "s01march": "evp_march"}


def _is(name: str, match) -> bool:
    """`match`: a substring, or "re:<pattern>"."""
    return bool(re.search(match[3:], name)) if match.startswith("re:") else match in name


def counters(path: Path, match):
    """{counter: (average per launch, launches)} and the average duration of kernels whose name contains `match`."""
    agg, dur = defaultdict(list), []
    kname = None
    with open(path) as f:
        for r in csv.DictReader(f):
            if not _is(r["Kernel_Name"], match):
                continue
            kname = r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {c: (sum(v) / len(v), len(v)) for c, v in agg.items()}
    return out, (sum(dur) / len(dur) * 1e-3 if dur else None), kname


def kernel_stats(path: Path, match):
    with open(path) as f:
        for r in csv.DictReader(f):
            if _is(r["Name"], match):
                return dict(name=r["Name"], calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) * 1e-3,
                            min_us=float(r["MinNs"]) * 1e-3, max_us=float(r["MaxNs"]) * 1e-3)
    return None


def calibration(d: Path):
    """bytes really moved / bytes the counter reports, per calibration kernel."""
    known = {}
    plain = d / "calib_plain.log"
    if not plain.exists():
        return None
    for line in plain.read_text().splitlines():
        m = re.match(r"CALIB (\w+) bytes_read (\d+) bytes_written (\d+)", line)
        if m:
            known[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    out = {}
    for cname, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
        f = d / f"calib_{cname}_counter_collection.csv"
        if not f.exists():
            continue
        for k, b in known.items():
            c, _, _ = counters(f, k)
            if cname in c and c[cname][0] > 0 and b[idx] > 0:
                out[f"{cname}:{k}"] = b[idx] / (c[cname][0] * 1024.0)
    return out


def main():
    d, dst = Path(sys.argv[1]), Path(sys.argv[2])
    cal = calibration(d) or {}
    # 8 B per lane coalesced, as every array access of the streaming EVP kernel
    f_fetch = cal.get("FETCH_SIZE:calib_read8", 2.0)
    f_write = cal.get("WRITE_SIZE:calib_write8", 1.0)
    res = {"source": str(d), "calibration": {"measured": cal, "fetch_factor_used": f_fetch, "write_factor_used": f_write,
                                             "note": "bytes = counter [KB] x 1024 x factor; factors from tools/pmc_calib.hip "
                                                     "(1 GiB streams of 8 B per lane)"},
           "kernels": {}}
    for key, match in DOMINANT.items():
        st = d / f"{key}_trace_kernel_stats.csv"
        if not st.exists():
            continue
        entry = {"kernel_trace": kernel_stats(st, match)}
        cnt = {}
        for p in ("sq1", "sq2", "fetch", "write"):
            f = d / f"{key}_{p}_counter_collection.csv"
            if f.exists():
                c, dur_us, kname = counters(f, match)
                for k, (v, n) in c.items():
                    cnt[k] = {"avg_per_launch": v, "launches": n, "pass": p, "pass_avg_us": dur_us}
                entry["kernel"] = kname
        entry["counters"] = cnt
        g = lambda k: cnt[k]["avg_per_launch"] if k in cnt else None
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            entry["hbm_bytes_per_launch"] = (g("FETCH_SIZE") * f_fetch + g("WRITE_SIZE") * f_write) * 1024.0
            entry["hbm_read_bytes_per_launch"] = g("FETCH_SIZE") * f_fetch * 1024.0
            entry["hbm_write_bytes_per_launch"] = g("WRITE_SIZE") * f_write * 1024.0
        if g("SQ_ACTIVE_INST_VALU") is not None:
            # SQ_* cycle counters tick in quad-cycles (MI355X_MICROARCH.md, per-instruction constants)
            busy = g("SQ_ACTIVE_INST_VALU") * 4.0
            entry["valu_busy_simd_cycles_per_launch"] = busy
            dur = cnt["SQ_ACTIVE_INST_VALU"]["pass_avg_us"] * 1e-6
            entry["valu_busy_frac_of_max_clock_in_pmc_pass"] = busy / (N_SIMD * dur * MAX_CLOCK_HZ)
            if g("SQ_WAVE_CYCLES"):
                entry["wave_cycles_share"] = {k: g(k) / g("SQ_WAVE_CYCLES") for k in
                                              ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if g(k)}
            if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
                entry["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
        if g("GRBM_GUI_ACTIVE") is not None:
            dur = cnt["GRBM_GUI_ACTIVE"]["pass_avg_us"] * 1e-6
            # GRBM_GUI_ACTIVE covers the dispatch of a launch, not only the kernel's own duration: for launches of a few
            # microseconds the quotient exceeded the chip's 2.4 GHz (round 3: 3.48 "GHz" for the 11.8-us streaming launch).
            # Reported only where the launch is long enough for the overhead to vanish and the value is physically possible.
            ghz = g("GRBM_GUI_ACTIVE") / N_XCD / dur * 1e-9
            entry["effective_clock_ghz"] = ghz if (dur >= 200e-6 and ghz <= MAX_CLOCK_HZ * 1e-9 * 1.01) else None
        bj = d / f"{key}_bench_under_trace.json"
        if bj.exists() and bj.read_text().strip():
            try:
                b = json.loads(bj.read_text())
                entry["bench_line_under_trace"] = {"us_per_subcycle": b["config"]["us_per_subcycle"],
                                                   "kernel_us": b["roofline"]["kernel_us"],
                                                   "tile_variant": b["config"]["tile_variant"]}
            except Exception:  # noqa: BLE001
                pass
        res["kernels"][key] = entry
    # C-grid subcycle (tools/cgrid_timing.py): the three kernels of the fused schedule, duration and HBM-side bytes each
    CG = {"A_avg_strain": "cg_avg_strain", "B_stress_t": "cg_stress_t", "C_stress_u_step": "cg_stress_u_step"}
    for key in ("cgx1", "cgs01", "cgx1one", "cgs01one", "cgx1res", "cgtx1res", "cgtx1"):
        st = d / f"{key}_trace_kernel_stats.csv"
        if not st.exists():
            continue
        entry = {}
        # (one launch per subcycle: on large domains the marched kernel cg_strip -- the windows along the block edges ride in its launch;
        # matched: the instantiations with LAST = false, i.e. every subcycle of a call but the first and the last --; elsewhere cg_one alone)
        for tag, match in ({"resident": "cg_res<"} if key.endswith("res") else {"marched": "re:cg_strip<(true|false), false,", "one_launch": "cg_one"} if key.endswith("one") else CG).items():
            if kernel_stats(st, match) is None:
                continue
            e = {"kernel_trace": kernel_stats(st, match)}
            for p, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
                f = d / f"{key}_{p}_counter_collection.csv"
                if f.exists():
                    c, dur_us, kname = counters(f, match)
                    if cname in c:
                        e[cname + "_KB_per_launch"] = c[cname][0]
            if "FETCH_SIZE_KB_per_launch" in e and "WRITE_SIZE_KB_per_launch" in e:
                e["hbm_bytes_per_launch"] = (e["FETCH_SIZE_KB_per_launch"] * f_fetch + e["WRITE_SIZE_KB_per_launch"] * f_write) * 1024.0
            sq = {}
            for p in ("sq1", "sq2"):
                f = d / f"{key}_{p}_counter_collection.csv"
                if f.exists():
                    c, dur_us, kname = counters(f, match)
                    sq.update({k: v for k, (v, n) in c.items()})
                    sq.update({k + "_launches": n for k, (v, n) in c.items()})
                    if p == "sq1":
                        sq["_pass_avg_us"] = dur_us
            if sq.get("SQ_WAVE_CYCLES"):
                e["sq"] = {"valu_insts_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"] if sq.get("SQ_WAVES") else None,
                           "waves_per_launch": sq.get("SQ_WAVES"),
                           "valu_busy_simd_cycles_per_launch": (sq["SQ_ACTIVE_INST_VALU"] * 4.0) if sq.get("SQ_ACTIVE_INST_VALU") else None,
                           "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"),
                           "wave_cycles_share": {k: sq[k] / sq["SQ_WAVE_CYCLES"] for k in
                                                 ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if sq.get(k)},
                           "valu_busy_frac_of_max_clock_in_pmc_pass": sq["SQ_ACTIVE_INST_VALU"] * 4.0 /
                                                                      (N_SIMD * sq["_pass_avg_us"] * 1e-6 * MAX_CLOCK_HZ)
                           if sq.get("SQ_ACTIVE_INST_VALU") else None,
                           "vmem_rd_insts_per_wave": sq["SQ_INSTS_VMEM_RD"] / sq["SQ_WAVES"] if sq.get("SQ_INSTS_VMEM_RD") and sq.get("SQ_WAVES") else None,
                           "vmem_wr_insts_per_wave": sq["SQ_INSTS_VMEM_WR"] / sq["SQ_WAVES"] if sq.get("SQ_INSTS_VMEM_WR") and sq.get("SQ_WAVES") else None}
            entry[tag] = e
        tot_us = sum(e["kernel_trace"]["avg_us"] for e in entry.values() if e.get("kernel_trace"))
        tot_b = sum(e.get("hbm_bytes_per_launch", 0.0) for e in entry.values())
        entry["per_subcycle"] = {"kernel_us_sum": tot_us, "hbm_bytes": tot_b or None,
                                 "hbm_GBps_over_kernel_time": (tot_b / (tot_us * 1e-6) / 1e9) if tot_b and tot_us else None}
        res["kernels"][key] = entry
    dst.write_text(json.dumps(res, indent=1))
    for k, e in res["kernels"].items():
        if k.startswith("cg"):
            print(k, e["per_subcycle"], {t: v.get("kernel_trace", {}) and v["kernel_trace"].get("avg_us") for t, v in e.items() if t != "per_subcycle"})
            continue
        print(k, {q: e.get(q) for q in ("hbm_bytes_per_launch", "valu_busy_simd_cycles_per_launch", "effective_clock_ghz")},
              e["kernel_trace"])


if __name__ == "__main__":
    main()
