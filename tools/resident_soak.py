#!/usr/bin/env python3
"""Soak run of the benchmarked kernel: N launches of 120 subcycles on the gx1 workload of bench.py, back to back, in
groups of 50 from a fresh upload; after every group the state is hashed and compared with the committed checksum of
50 steps (tests/golden/bench_checksums.json, made by the CPU oracle).  A stale record tag, a missed hand-off or a
parity slip between launches would show as a mismatch.   usage: tools/resident_soak.py [launches=20000]"""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from cice_amd import decomp, evp, synth  # noqa: E402

golden = json.load(open(ROOT / "tests" / "golden" / "bench_checksums.json"))
spec = synth.GRIDS["gx1"]; nx, ny = spec["nx"], spec["ny"]
g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
st = synth.make_state(g, case="full", seed=20260928, warm=True)
dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", "closed")
geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
tm = dc.scatter(st["iceTmask"], 0, fill=0); um = dc.scatter(st["iceUmask"], 0, fill=0)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(120), strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t0 = time.time()
bad = 0
for rep in range(N // 50):
    core.upload(fields, tm, um)
    for _ in range(50):
        core.subcycle(120)
    out = core.download()
    h = hashlib.sha256()
    for k in ("uvel", "vvel", "stressp_1"):
        h.update(np.ascontiguousarray(dc.gather({0: out[k]}), dtype="<f8").tobytes())
    ok = h.hexdigest() == golden["gx1/full/ndte120/closed/strict"]["50"]["sha256"]
    bad += (not ok)
print("SOAK launches", N, "groups", N // 50, "bad", bad, "variant", core.timings()["tile_variant"], "%.1f s" % (time.time() - t0))
core.finalize()
