#!/usr/bin/env python3
"""Checksums of the state bench.py must hold after N steps (N x ndte subcycles) of its synthetic
workloads, computed by the CPU oracle (oracle/evp_oracle.c, pinned bit for bit to the compiled
reference on the fixtures of this directory).  bench.py hashes the same fields after its timed region
and reports "verified".  Data only: tests/golden/bench_checksums.json.

  python tests/golden/make_bench_checksums.py [gx3 gx1 s01]
"""
from __future__ import annotations

import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))

import oracle  # noqa: E402
from cice_amd import decomp, evp, synth  # noqa: E402

OUT = Path(__file__).resolve().parent / "bench_checksums.json"
FIELDS = ("uvel", "vvel", "stressp_1")          # bench.py VERIFY_FIELDS
CONFIGS = {   # workload: (case, ndte, ns, checkpoints)
    "gx3": ("full", 120, "closed", [1, 3, 5, 12, 23, 25, 50]),
    "gx1": ("full", 120, "closed", [1, 3, 5, 12, 23, 25, 50]),
    "s01": ("full", 480, "closed", [3]),
    "tx1": ("full", 240, "tripole", [12]),
    "gx1@240": ("full", 240, "closed", [3, 5, 12]),      # BASELINE configs[2]: gx1 at ndte = 240 (bench.py N > 1)
    "gx1@caps": ("caps", 120, "closed", [12]),            # SURVEY 8(d)'s second ice case (bench.py `caps`)
}


def run(workload):
    case, ndte, ns, cps = CONFIGS[workload]
    workload = workload.split("@")[0]
    spec = synth.GRIDS[workload]
    nx, ny = spec["nx"], spec["ny"]
    g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=ns))
    st = synth.make_state(g, case=case, seed=20260928, warm=True)      # bench.py measure()
    dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", ns)
    geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
    fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
    tm = dc.scatter(st["iceTmask"], 0, fill=0)
    um = dc.scatter(st["iceUmask"], 0, fill=0)
    scal = synth.evp_scalars(ndte)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), nx, ny, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    m = oracle.metrics(dom, scal["deltaminEVP"], geo["HTE"], geo["HTN"], geo["tarea"])
    if ns == "tripole":      # what bench.py hands to cice_evp_hip_set_metrics must be what the oracle uses
        dxhy, dyhx = synth.bgrid_fold_metrics(dc, 0, g)
        assert np.array_equal(dxhy, m["dxhy"]) and np.array_equal(dyhx, m["dyhx"])
    static = dict(m, dxT=geo["dxT"], dyT=geo["dyT"], uarear=geo["uarear"])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i",
                                                      "capping", "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    res, done = {}, 0
    dyn = dict(fields)
    for n in cps:
        t0 = time.time()
        out = oracle.subcycle(dom, prm, (n - done) * ndte, dyn, static, tm, um)
        done = n
        dyn.update({k: out[k] for k in evp.OUTPUTS})
        glob = {k: dc.gather({0: out[k]}) for k in FIELDS}
        h = hashlib.sha256()
        for k in FIELDS:
            h.update(np.ascontiguousarray(glob[k], dtype="<f8").tobytes())
        res[str(n)] = dict(sha256=h.hexdigest(), sum_abs_u=float(np.abs(glob["uvel"]).sum()),
                           max_abs_u=float(np.abs(glob["uvel"]).max()))
        print(f"{workload} N={n}: {res[str(n)]['sha256'][:16]} max|u| {res[str(n)]['max_abs_u']:.6f} ({time.time() - t0:.1f} s)", flush=True)
    return f"{workload}/{case}/ndte{ndte}/{ns}/strict", res


CGRID_FIELDS = ("uvelE", "vvelN", "stresspT", "stress12U")      # bench.py CGRID_VERIFY_FIELDS
CGRID_CONFIGS = {"gx3": ("full", 120, [1, 4]), "gx1": ("full", 120, [1, 4]), "s01": ("full", 12, [2]), "s01@120": ("full", 120, [2]), "tx1": ("full", 120, [4])}


def cgrid_inputs(workload, case):
    """The C-grid workload of bench.py (shared: bench.py imports nothing from here, it builds the same through synth)."""
    spec = synth.GRIDS[workload]
    nx, ny = spec["nx"], spec["ny"]
    ns = spec.get("ns", "closed")          # (tx1: tripole)
    g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=ns))
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case=case, seed=20260928, warm=True)
    dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", ns)
    return dc, synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)


def run_cgrid(workload):
    case, ndte, cps = CGRID_CONFIGS[workload]
    workload = workload.split("@")[0]
    dc, (static, state, inputs, masks) = cgrid_inputs(workload, case)
    scal = synth.evp_scalars(ndte)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i",
                                                      "capping", "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    res, done = {}, 0
    cur = dict(state)
    for n in cps:
        t0 = time.time()
        out = oracle.cgrid_subcycle(dom, prm, (n - done) * ndte, cur, inputs, static, masks)
        done = n
        cur = {k: out[k] for k in oracle.C_FIELDS}      # work arrays carry over too (same as the resident GPU state)
        h = hashlib.sha256()
        for k in CGRID_FIELDS:
            h.update(np.ascontiguousarray(dc.gather({0: out[k]}), dtype="<f8").tobytes())
        res[str(n)] = dict(sha256=h.hexdigest(), max_abs_uE=float(np.abs(out["uvelE"]).max()))
        print(f"cgrid {workload} N={n}: {res[str(n)]['sha256'][:16]} max|uE| {res[str(n)]['max_abs_uE']:.6f} ({time.time() - t0:.1f} s)", flush=True)
    return f"cgrid/{workload}/{case}/ndte{ndte}/{dc.ns}/strict", res


if __name__ == "__main__":
    allres = json.loads(OUT.read_text()) if OUT.exists() else {}
    args = sys.argv[1:] or ["gx3", "gx1"]
    for w in [x[6:] for x in args if x.startswith("cgrid:")]:
        key, res = run_cgrid(w)
        allres[key] = res
        OUT.write_text(json.dumps(allres, indent=1, sort_keys=True))
    for w in [x for x in args if not x.startswith("cgrid:")]:
        key, res = run(w)
        allres[key] = res
        OUT.write_text(json.dumps(allres, indent=1, sort_keys=True))
