"""GPU (-m gpu): the several-subcycles-per-pass kernel (cice_amd/csrc/evp_march.hip: one wave marches north over a strip
of 64 columns, (stress -> stepu) x 4 per row -- or x 2, x 3 --, neighbours by wave shuffles, a device-private rectangle
layout) against the golden fixtures frozen from the reference's evp() and against the CPU oracle -- bit for bit in
strict mode.  The path is the default from 450k cells per rank; here it is forced on small grids
(CICE_EVP_HIP_MARCH=1, on-chip resident kernel off) with short segments so that every overlap rule is exercised:
strips (<= 56 owned columns of 64 lanes), segments, cyclic wrap images, several blocks per rank, padded blocks, closed
east-west boundaries, subcycle counts that leave 1, 2 or 3 after the passes of four, revised EVP / seabed stress /
fractional capping (the non-LEAN variants).  CICE_EVP_HIP_MARCH_K (test build) = subcycles per full pass; by default the library
takes four from 46 rows per segment, three from 23, two below (evp_host_march.cpp)."""
import numpy as np
import pytest

import oracle  # noqa: F401
from cice_amd import decomp, evp, synth
from common import GOLDEN_CASES, GoldenCase, assert_bitwise
from test_gpu_parity import SIG, VEL, hip_from_case, run_hip, run_oracle, synth_case

pytestmark = pytest.mark.gpu

NON_TRIPOLE = [n for n in GOLDEN_CASES if not n.startswith("trip")]


@pytest.fixture
def march(monkeypatch):
    monkeypatch.setenv("CICE_EVP_HIP_MARCH", "1")
    monkeypatch.setenv("CICE_EVP_HIP_RESIDENT", "0")
    return monkeypatch


def npasses(ndte, k=4):
    """Passes the marching path needs for ndte subcycles: passes of k, one pass of the remaining 2 .. k-1; a single remaining
    subcycle goes through the one-subcycle kernel (evp_host_march.cpp: march_run)."""
    q, rem = divmod(ndte, k)
    return q + (1 if rem >= 2 else 0)


@pytest.mark.parametrize("seg,kpass", [(0, 4), (5, 4), (5, 3), (0, 2)])
@pytest.mark.parametrize("name", NON_TRIPOLE)
def test_march_golden_strict_bitwise(name, seg, kpass, march):
    """Every non-tripole fixture of the reference (1 .. 6 blocks, padded blocks, cyclic and closed E-W, classic and
    revised EVP, capping 0 / 0.5 / 1, Ktens, seabed stress), 1 / 2 / 10 / 120 subcycles, two calls; four (the default),
    three and two subcycles per pass."""
    if seg:
        march.setenv("CICE_EVP_HIP_MARCH_SEG", str(seg))
    march.setenv("CICE_EVP_HIP_MARCH_K", str(kpass))
    c = GoldenCase(name)
    core = hip_from_case(c, strict=True)
    try:
        for icall in range(1, c.ncalls + 1):
            dyn, tm, um = c.inputs(icall)
            for nsub in c.nsub_list:
                out = core.run(dyn, tm, um, ndte=nsub)
                assert_bitwise(out, c.expected(icall, nsub), f"{name} call {icall} nsub {nsub} (march)")
                info = core.march_info()
                assert info["mode"] == 1 and info["last_call"] and info["declined"] == 0 and info["kpass"] == kpass, info
        assert core.march_info()["passes"] > 0
    finally:
        core.finalize()


@pytest.mark.parametrize("grid,case,bs,warm,seg", [("gx3", "full", None, False, 0), ("gx3", "caps", (25, 29), True, 7),
                                                    ("gx3", "full", (50, 58), True, 16), ("gx1", "full", None, True, 0),
                                                    ("gx1", "caps", (80, 96), False, 48)])
def test_march_synthetic_vs_oracle_strict_bitwise(grid, case, bs, warm, seg, march):
    """gx3 / gx1-sized synthetic grids (curvilinear metrics, land, cyclic E-W: the wrap images carry the state across
    the seam) against the CPU oracle; 14 subcycles = three passes of four and one of two."""
    if seg:
        march.setenv("CICE_EVP_HIP_MARCH_SEG", str(seg))
    dc, geo, fields, tm, um = synth_case(grid, case, seed=20260928, warm=warm, bs=bs)
    scal = synth.evp_scalars(120)
    got = run_hip(dc, geo, fields, tm, um, scal, strict=True, ndte=14)
    want = run_oracle(dc, geo, fields, tm, um, scal, 14)
    assert np.abs(want["uvel"]).max() > 1e-4
    assert_bitwise(got, want, f"{grid}/{case} march vs oracle")


def test_march_equals_streaming_kernel_at_odd_counts_and_across_calls(march):
    """upload / subcycle(60) / subcycle(57) / subcycle(2) / subcycle(1) / download: 57 = 4 x 14 + 1 starts with one subcycle of
    the one-subcycle kernel, 2 is one pass of two; the block-layout state is current after every call."""
    march.setenv("CICE_EVP_HIP_MARCH_K", "4")          # (a grid this small gets two subcycles per pass on its own)
    c = GoldenCase("pop_cyc_3x2pad_caps")
    core = hip_from_case(c, strict=True)
    try:
        dyn, tm, um = c.inputs(1)
        core.upload(dyn, tm, um)
        core.subcycle(60)
        core.subcycle(57)
        core.subcycle(2)
        core.subcycle(1)
        core.sync()
        assert_bitwise(core.download(), c.expected(1, 120), "march: upload/subcycle x4/download")
        info = core.march_info()
        assert info["kpass"] == 4 and info["passes"] == 15 + 14 + 1 and info["subcycles"] == 60 + 56 + 2, info
    finally:
        core.finalize()


def test_march_fused_mode_equals_streaming_fused(march):
    """Fused build: contraction is per source expression, so the marching kernel produces the bits of every other kernel
    that inlines evp_cell.inc."""
    scal = synth.evp_scalars(120)
    dc, geo, fields, tm, um = synth_case("gx3", "full", seed=4, warm=True, bs=(50, 58))
    a = run_hip(dc, geo, fields, tm, um, scal, strict=False, ndte=20)
    march.setenv("CICE_EVP_HIP_MARCH", "0")
    b = run_hip(dc, geo, fields, tm, um, scal, strict=False, ndte=20)
    assert_bitwise(a, b, "fused: march vs streaming")


@pytest.mark.parametrize("seed,grid,bs,holes", [(11, "gx3", None, 0.3), (12, "gx3", (50, 58), 0.6), (14, "gx3", None, 1.1)])
def test_march_random_masks_vs_oracle(seed, grid, bs, holes, march):
    """Random, mutually independent holes in iceTmask / iceUmask on the GLOBAL grid (ghost cells are images, as at the
    reference's boundary): every mask-dependent branch of the march -- ice next to none in either direction, whole
    strips without ice -- against the oracle."""
    scal = synth.evp_scalars(120)
    spec = synth.GRIDS[grid]
    nx, ny = spec["nx"], spec["ny"]
    g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
    st = synth.make_state(g, case="full", seed=seed, warm=True)
    rng = np.random.default_rng(seed)
    tmg = (st["iceTmask"] * (rng.random((ny, nx)) > holes)).astype(np.int32)
    umg = (st["iceUmask"] * (rng.random((ny, nx)) > holes)).astype(np.int32)
    for k in evp.FIELDS[:12]:
        st[k] = st[k] * tmg
    for k in ("uvel", "vvel", "uvel_init", "vvel_init"):
        st[k] = st[k] * umg
    bsz = bs or (nx, ny)
    dc = decomp.Decomp(nx, ny, bsz[0], bsz[1], "cyclic", "closed", 1)
    geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k in ("HTE", "HTN", "dxT", "dyT", "tarea") else 0.0))
           for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
    fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
    tm = dc.scatter(tmg, 0, fill=0)
    um = dc.scatter(umg, 0, fill=0)
    march.setenv("CICE_EVP_HIP_MARCH_SEG", "9")
    got = run_hip(dc, geo, fields, tm, um, scal, strict=True, ndte=10)
    want = run_oracle(dc, geo, fields, tm, um, scal, 10)
    assert_bitwise(got, want, f"march, random masks seed {seed}")


@pytest.mark.parametrize("overlap", [0, 1, "direct"])
@pytest.mark.parametrize("grid,case,bs,seg,own,ext", [("gx3", "full", None, 0, 0, 2), ("gx3", "caps", (25, 29), 9, 17, 0),
                                                       ("gx3", "caps", (50, 58), 9, 23, 4), ("gx1", "full", None, 40, 0, 2)])
def test_march_ring_exchanged_over_rccl_with_the_rank_itself(grid, case, bs, seg, own, ext, overlap, march):
    """Several ranks: the passes are followed by an exchange of the four-cell ring (pack -> ncclSend / ncclRecv -> unpack,
    duplicates included; march_plan.cpp).  One GPU can run all of it by treating the cyclic seam of the domain as a rank
    boundary -- the rank is its own east and west neighbour (CICE_EVP_HIP_MARCH_SELFX=1): no wrap inside the strips, the
    halo columns live on what RCCL delivers.  ext: the rank also holds (and advances redundantly) `ext` columns of its
    neighbour -- here of itself -- on either side, so that the ring is exchanged every (ext + 4)-th subcycle only.
    overlap = 1 (CICE_EVP_HIP_MARCH_OVERLAP): the cells the neighbour waits for are advanced first by an early launch on
    the second stream, pack + send / recv run there while the pass itself runs on the compute stream (round 4).
    "direct" (CICE_EVP_HIP_MARCH_DIRECT=1): no library -- the pack kernel stores into the neighbour's inbox (here: its own),
    flags instead of send / recv; the first exchange runs both ways and must agree bit for bit before it is used.
    Against the oracle, bit for bit; the list logic for 2 and 4 ranks is
    tests/test_multirank_cpu.py::test_march_ring_between_ranks_known_answer."""
    march.setenv("CICE_EVP_HIP_MARCH_OVERLAP", "0" if overlap == "direct" else str(overlap))
    march.setenv("CICE_EVP_HIP_MARCH_DIRECT", "1" if overlap == "direct" else "0")
    march.setenv("CICE_EVP_HIP_MARCH_SELFX", "1")
    march.setenv("CICE_EVP_HIP_MARCH_EXT", str(ext))
    if seg:
        march.setenv("CICE_EVP_HIP_MARCH_SEG", str(seg))
    if own:
        march.setenv("CICE_EVP_HIP_MARCH_OWN", str(own))
    dc, geo, fields, tm, um = synth_case(grid, case, seed=8, warm=True, bs=bs)
    scal = synth.evp_scalars(120)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    try:
        core.comm_init(core.comm_unique_id())
        got = core.run(fields, tm, um, ndte=14)
        info = core.march_info()
        assert info["mode"] == 1 and info["last_call"] and info["passes"] == npasses(14, info["kpass"]), info
        assert info["ring"] == ("direct stores (HIP IPC)" if overlap == "direct" else "rccl"), info
        if overlap == "direct":          # a second call: every exchange through the inboxes now
            got = core.run(fields, tm, um, ndte=14)
    finally:
        core.finalize()
    want = run_oracle(dc, geo, fields, tm, um, scal, 14)
    assert_bitwise(got, want, f"{grid}/{case}: march, seam through RCCL, ext {ext}")


def test_march_declines_a_state_whose_ghost_cells_are_not_images(march):
    """The rectangle holds every cell once; the reference keeps per-block ghost storage and computes the T-cells of the
    north / east fringe from it.  A caller whose ghost values differ from the cells they image (here: ice punched out
    of the east ghost column of the mask, and a ghost stress changed) gets the one-subcycle kernels -- and the
    reference's answer for exactly that input."""
    scal = synth.evp_scalars(120)
    dc, geo, fields, tm, um = synth_case("gx3", "full", seed=6, warm=True, bs=(50, 58))
    tm = tm.copy()
    fields = {k: v.copy() for k, v in fields.items()}
    tm[:, :, -1] = 0                      # east ghost column of every block: no longer the image of column ilo
    fields["stressp_1"][:, 5:20, -1] *= 1.5
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    try:
        got = core.run(fields, tm, um, ndte=8)
        info = core.march_info()
        assert info["mode"] == 1 and info["declined"] == 1 and not info["last_call"], info
    finally:
        core.finalize()
    want = run_oracle(dc, geo, fields, tm, um, scal, 8)
    assert_bitwise(got, want, "declined call: one-subcycle kernels on the caller's own ghost values")


def test_march_is_the_default_on_a_large_grid_and_invariant_under_the_cut():
    """1440 x 720 (1.04M cells: above the threshold, nothing forced): the marching kernel runs by default; cutting the
    domain into other strips' segments or into CICE blocks changes nothing (size-independent property; the oracle
    comparison at 3600 x 2400 is bench.py's committed checksum)."""
    import os
    scal = synth.evp_scalars(480)
    nx, ny = 1440, 720
    g = synth.derive_geometry(synth.make_grid(nx, ny, 2.8e4, ns="closed"))
    st = synth.make_state(g, case="full", seed=3, warm=True)
    ref = None
    for bs, seg in (((nx, ny), None), ((nx, ny), "37"), ((360, 240), None)):
        if seg:
            os.environ["CICE_EVP_HIP_MARCH_SEG"] = seg
        try:
            dc = decomp.Decomp(nx, ny, bs[0], bs[1], "cyclic", "closed", 1)
            geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k in ("HTE", "HTN", "dxT", "dyT", "tarea") else 0.0))
                   for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
            fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
            tm = dc.scatter(st["iceTmask"], 0, fill=0)
            um = dc.scatter(st["iceUmask"], 0, fill=0)
            d, keep = evp.make_dims(dc, 0)
            core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                              geo["uarear"], geo["tarea"], keepalive=keep)
            try:
                out = core.run(fields, tm, um, ndte=6)
                info = core.march_info()
                assert info["mode"] == 1 and info["last_call"] and info["passes"] == npasses(6, info["kpass"]), info
            finally:
                core.finalize()
        finally:
            os.environ.pop("CICE_EVP_HIP_MARCH_SEG", None)
        glob = {k: dc.gather({0: out[k]}) for k in VEL + SIG + ["strintxU", "taubxU"]}
        if ref is None:
            ref = glob
            assert np.abs(glob["uvel"]).max() > 1e-5
        else:
            assert_bitwise(glob, ref, f"1440x720 cut {bs} seg {seg}")
    # and the one-subcycle kernel gives the same bits
    os.environ["CICE_EVP_HIP_MARCH"] = "0"
    try:
        dc = decomp.Decomp(nx, ny, nx, ny, "cyclic", "closed", 1)
        geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k in ("HTE", "HTN", "dxT", "dyT", "tarea") else 0.0))
               for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
        fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
        out = run_hip(dc, geo, fields, dc.scatter(st["iceTmask"], 0, fill=0), dc.scatter(st["iceUmask"], 0, fill=0),
                      scal, strict=True, ndte=6)
    finally:
        os.environ.pop("CICE_EVP_HIP_MARCH", None)
    assert_bitwise({k: dc.gather({0: out[k]}) for k in ref}, ref, "1440x720: march vs one-subcycle kernel")


def test_s01_full_size_march_vs_oracle_bitwise():
    """BASELINE's largest configuration (3600 x 2400 = 8.6M cells, one block, nothing forced: the marching kernel is the
    default there) against the CPU oracle itself, 15 subcycles = three passes of four + one of three: every
    output field on every cell, bit for bit.  (bench.py verifies the same workload after 3 x 480 subcycles against a
    committed checksum of the oracle's state.)"""
    scal = synth.evp_scalars(480)
    dc, geo, fields, tm, um = synth_case("s01", "full", seed=2, warm=True)
    want = run_oracle(dc, geo, fields, tm, um, scal, 15)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    try:
        got = core.run(fields, tm, um, ndte=15)
        info = core.march_info()
        assert info["mode"] == 1 and info["last_call"] and info["passes"] == 4 and info["subcycles"] == 15 and info["declined"] == 0, info
        path = core.describe_path()
        assert "four subcycles per pass" in path and "marching path: on" in path and "one rank" in path, path
    finally:
        core.finalize()
    assert np.abs(want["uvel"]).max() > 1e-3
    assert_bitwise(got, want, "3600x2400 march vs oracle, 15 subcycles")


def test_march_plan_is_built_for_a_rank_of_several(tmp_path):
    """The marching path decides at the first cice_evp_hip_subcycle, long after cice_evp_hip_init has returned: the
    global block table handed to init must still be there then (it was dropped once: every multi-rank domain silently got
    the one-subcycle kernels).  One process plays rank 0 of a 2 x 1 split without a communicator: the plan must get as far
    as asking for one -- RCCL refuses two ranks on one device, so the exchange itself is exercised with the rank itself
    (test_march_ring_exchanged_over_rccl_with_the_rank_itself) and on the CPU (tests/test_multirank_cpu.py)."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path[:0] = ["%s", "%s/tests", "%s/oracle"]
import numpy as np
from cice_amd import decomp, evp, synth
nx, ny = 240, 120
g = synth.derive_geometry(synth.make_grid(nx, ny, 2.8e4, ns="closed"))
st = synth.make_state(g, case="full", seed=3, warm=True)
dc = decomp.per_rank_blocks(nx, ny, 2, "cyclic", "closed", proc_shape=(2, 1))
geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(120), strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                  geo["uarear"], geo["tarea"], keepalive=keep)
del keep, d                      # the caller's table is gone, as CICE's temporaries are after dyn_evp_hip_init
import gc; gc.collect()
try:
    core.upload(fields, dc.scatter(st["iceTmask"], 0, fill=0), dc.scatter(st["iceUmask"], 0, fill=0))
    core.subcycle(2)
    core.sync()
except evp.EvpHipError as e:
    print("EXPECTED_ERROR", str(e)[:200])
''' % ((str(__import__("pathlib").Path(__file__).resolve().parents[1]),) * 3)
    import os
    env = dict(os.environ, CICE_EVP_HIP_MARCH="1", CICE_EVP_HIP_RESIDENT="0", CICE_EVP_HIP_VERBOSE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "global block table required" not in r.stderr, r.stderr[-1500:]
    assert "no RCCL communicator" in r.stderr, (r.stdout[-800:], r.stderr[-1500:])


@pytest.mark.parametrize("seed", list(range(101, 113)) + [2056, 17197] + [int(s) for s in __import__("os").environ.get("MARCH_SWEEP_SEEDS", "").split() if s])
def test_march_random_geometry_vs_oracle(seed, march):
    """Geometry sweep: random domain sizes (not multiples of the strip width), block splits with padded last blocks, strip
    widths, segment lengths, closed / cyclic east-west boundaries, random ice holes, even and odd subcycle counts -- and,
    on cyclic domains every other seed, the seam exchanged as a ring with a random redundant rim (the several-rank form
    with the rank itself as neighbour).  Every output field against the oracle, bit for bit.
    (Seed 2056: 144 columns held in strips of 13 left one column to the last strip, and the first column beyond the
    rectangle, which the exchange fills, sat in two strips but was delivered to one: march_plan.cpp now keeps two.)"""
    rng = np.random.default_rng(seed)
    nx, ny = int(rng.integers(66, 210)), int(rng.integers(30, 110))
    ew = "cyclic" if seed % 3 else "closed"
    nbx, nby = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    bsx, bsy = -(-nx // nbx), -(-ny // nby)
    own = int(rng.choice([0, 13, 29, 47, 56]))
    seg = int(rng.integers(5, 45))
    ndte = int(rng.choice([6, 9, 12, 7, 15]))
    kpass = int(rng.choice([4, 4, 3, 2]))
    march.setenv("CICE_EVP_HIP_MARCH_K", str(kpass))
    selfx = ew == "cyclic" and seed % 2 == 0
    ext = int(rng.choice([0, 2, 4, 6, 8])) if selfx else 0
    march.setenv("CICE_EVP_HIP_MARCH_SEG", str(seg))
    if own:
        march.setenv("CICE_EVP_HIP_MARCH_OWN", str(own))
    if selfx:
        march.setenv("CICE_EVP_HIP_MARCH_SELFX", "1")
        march.setenv("CICE_EVP_HIP_MARCH_EXT", str(ext))
        march.setenv("CICE_EVP_HIP_MARCH_OVERLAP", str((seed // 2) % 2))       # every other of them: exchange overlapped with the pass
        march.setenv("CICE_EVP_HIP_MARCH_DIRECT", str((seed // 4) % 2))        # ... and without RCCL (where not overlapped)
        march.setenv("CICE_EVP_HIP_MARCH_BANDSEG", str(int(rng.integers(2, 12))))
    g = synth.derive_geometry(synth.make_grid(nx, ny, 3.0e4, ns="closed"))
    st = synth.make_state(g, case="full", seed=seed, warm=True)
    holes = float(rng.choice([0.0, 0.2, 0.7]))
    tmg = (st["iceTmask"] * (rng.random((ny, nx)) >= holes)).astype(np.int32)
    umg = (st["iceUmask"] * (rng.random((ny, nx)) >= holes)).astype(np.int32)
    for k in evp.FIELDS[:12]:
        st[k] = st[k] * tmg
    for k in ("uvel", "vvel", "uvel_init", "vvel_init"):
        st[k] = st[k] * umg
    dc = decomp.Decomp(nx, ny, bsx, bsy, ew, "closed", 1)
    geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k in ("HTE", "HTN", "dxT", "dyT", "tarea") else 0.0))
           for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
    fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
    tm, um = dc.scatter(tmg, 0, fill=0), dc.scatter(umg, 0, fill=0)
    scal = synth.evp_scalars(120)
    what = f"seed {seed}: {nx}x{ny} {ew}, blocks {bsx}x{bsy}, own {own}, seg {seg}, ndte {ndte}, selfx {selfx} ext {ext}, k {kpass}, holes {holes}"
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    try:
        if selfx:
            core.comm_init(core.comm_unique_id())
        got = core.run(fields, tm, um, ndte=ndte)
        info = core.march_info()
        path = core.describe_path()
        if info["mode"] == 0 and own and "no strip width" in path:
            # a forced narrow strip on a width it cannot tile (with the cyclic wrap inside, the last strip needs eight columns -- four
            # duplicated to either neighbour; seed 17197: 172 columns in strips of <= 13): the library says so and the streaming
            # kernel runs; the bits are compared all the same
            pass
        else:
            assert info["mode"] == 1 and info["last_call"] and info["declined"] == 0 and info["passes"] == npasses(ndte, kpass), (what, info, path)
    finally:
        core.finalize()
    want = run_oracle(dc, geo, fields, tm, um, scal, ndte)
    assert_bitwise(got, want, what)
