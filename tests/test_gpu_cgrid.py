"""GPU (-m gpu): the C-grid EVP subcycle through the C ABI (cice_evp_hip_cgrid_*) against
  (1) the fixtures frozen from the reference's own evp() with grid_ice = 'C' -- bit-exact, every array of the
      loop on every cell (ghost cells included; the loop's eight halo updates are fused into the kernels),
  (2) the CPU oracle on a seeded synthetic gx3-sized case, single block and 2 x 2 blocks -- bit-exact."""
import numpy as np
import pytest

import oracle
from cice_amd import evp
from common import CGRID_CASES, CGRID_TFOLD_CASES, GoldenCase, assert_bitwise, bits_equal

pytestmark = pytest.mark.gpu


def cgrid_core(c: GoldenCase):
    d, keep = c.hip_dims()
    prm = evp.make_params(c.scal_dict(), strict=True)
    ua = c.d["uarea"]
    uarear = np.where(ua > 0, 1.0 / np.where(ua > 0, ua, 1.0), 0.0)
    # cice_evp_hip_init wants the B-grid geometry too; on the C grid HTE = dyE and HTN = dxN (ice_grid.F90)
    core = evp.EvpHip(d, prm, c.d["dyE"], c.d["dxN"], c.d["dxT"], c.d["dyT"], uarear, c.d["tarea"], keepalive=keep)
    core.cgrid_set_geometry(c.cgrid_static())
    return core


def expected_loop_only(c, dom, icall, nsub):
    """The fixture holds the state after the whole evp(); the one post-loop step that touches the loop's arrays is
    the halo update of strintxE / strintyN (ice_dyn_evp.F90:1437-1440), which stays with the caller: compare
    those two on the cells the loop writes (their ghost cells keep the caller's values)."""
    return c.cgrid_expected(icall, nsub)


@pytest.mark.parametrize("name", CGRID_CASES + CGRID_TFOLD_CASES)
def test_cgrid_golden_bitwise(name):
    c = GoldenCase(name)
    dom = c.oracle_domain()
    core = cgrid_core(c)
    try:
        for icall in range(1, c.ncalls + 1):
            state, inputs, masks = c.cgrid_inputs(icall)
            for nsub in c.nsub_list:
                out = core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
                oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
                oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
                assert_bitwise(out, c.cgrid_expected(icall, nsub), f"{name} call {icall} nsub {nsub} (HIP C grid)")
    finally:
        core.finalize()


@pytest.mark.parametrize("name", CGRID_CASES + CGRID_TFOLD_CASES)
def test_cgrid_dyn_finish_on_device_bitwise(name):
    """dyn_finish at N and E points (ice_dyn_evp.F90:1408-1436), the last thing evp() computes from the C-grid loop's
    velocities, on the device from the loop's resident final state and the operands it already holds: strocnxN / strocnyN /
    strocnxE / strocnyE equal the arrays the reference's evp() left (fixtures regenerated with them in round 4), bit for bit
    on the cells of dyn_prep2's N / E lists; every other cell keeps the value it was handed (sentinel)."""
    from test_oracle_golden import cgrid_dyn_finish_lists
    c = GoldenCase(name)
    core = cgrid_core(c)
    keys = ("strocnxN", "strocnyN", "strocnxE", "strocnyE")
    try:
        for icall in range(1, c.ncalls + 1):
            state, inputs, masks = c.cgrid_inputs(icall)
            lists = cgrid_dyn_finish_lists(c, masks)
            for nsub in c.nsub_list:
                core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
                want = {k: c.d[f"o{icall:02d}n{nsub:04d}_{k}"] for k in keys}
                sent = {k: np.where(lists[k[-1]], 0.0, 7.25) for k in keys}
                got = core.cgrid_dyn_finish(prev=sent)
                for k in keys:
                    on = lists[k[-1]]
                    assert bits_equal(got[k][on], want[k][on]), f"{name} call {icall} nsub {nsub} {k}"
                    assert (got[k][~on] == 7.25).all(), f"{k}: a cell off the list was written"
        assert np.abs(want["strocnxE"]).max() > 0 and np.abs(want["strocnyN"]).max() > 0
    finally:
        core.finalize()


@pytest.mark.parametrize("name", CGRID_CASES + CGRID_TFOLD_CASES)
def test_cgrid_deformations_t_on_device_bitwise(name):
    """deformationsC_T (ice_dyn_shared.F90:1968-2074), which evp() runs right after the C-grid loop, on the device from
    the loop's resident final state: divu, shear, vort, rdg_conv, rdg_shear equal the arrays the reference's evp() left
    (fixtures regenerated with them), bit for bit on every cell -- the T-cells of the list are recomputed, every other
    cell must keep the value it was handed (inout arrays: handed the reference's own values there, and a sentinel test
    below shows nothing off the list is touched)."""
    c = GoldenCase(name)
    core = cgrid_core(c)
    keys = ("divu", "shear", "vort", "rdg_conv", "rdg_shear")
    try:
        for icall in range(1, c.ncalls + 1):
            state, inputs, masks = c.cgrid_inputs(icall)
            for nsub in c.nsub_list:
                core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
                want = {k: c.d[f"o{icall:02d}n{nsub:04d}_{k}"] for k in keys}
                tm = masks["iceTmask"] != 0
                sent = {k: np.where(tm, 0.0, 7.25) for k in keys}          # list cells are rewritten, the rest must stay 7.25
                got = core.cgrid_deformations(c.d["tarear"], prev=sent)
                blk = c.blk
                onlist = np.zeros_like(tm)
                for b in range(c.nblocks):
                    ilo, ihi, jlo, jhi = [int(v) for v in blk[b, :4]]
                    onlist[b, jlo - 1:jhi + 1, ilo - 1:ihi + 1] = tm[b, jlo - 1:jhi + 1, ilo - 1:ihi + 1]
                for k in keys:
                    assert bits_equal(got[k][onlist], want[k][onlist]), f"{name} call {icall} nsub {nsub} {k}"
                    assert (got[k][~onlist] == sent[k][~onlist]).all(), f"{k}: a cell off the T list was written"
        assert np.abs(want["divu"]).max() > 0 and np.abs(want["vort"]).max() > 0
    finally:
        core.finalize()


def test_cgrid_resident_kernel_bitwise(monkeypatch):
    """The on-chip resident C-grid kernel (evp_cgrid_res.hip: all subcycles of a call but the first after an upload in ONE launch,
    state in registers and LDS, face velocities traded between windows as tagged records) forced on: every fixture it is
    eligible for -- one rank, no T-fold, the default-configuration shortcuts -- bit-identical to the
    reference's arrays, ghost cells included, in one call and across calls; a cut it can never run refuses loudly, a call whose
    operands stand in the way (seabed stress) falls back to the one-launch kernel.  Forced off, the one-launch kernel gives the
    same bits."""
    ran, refused = [], []
    for name in CGRID_CASES:
        c = GoldenCase(name)
        dom = c.oracle_domain()
        for forced in ("1", "0"):
            monkeypatch.setenv("CICE_EVP_HIP_CGRID_RESIDENT", forced)
            core = cgrid_core(c)
            try:
                for icall in range(1, c.ncalls + 1):
                    state, inputs, masks = c.cgrid_inputs(icall)
                    for nsub in c.nsub_list:
                        try:
                            out = core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
                        except evp.EvpHipError as e:
                            assert forced == "1" and "not applicable" in str(e), (name, str(e))
                            refused.append((name, str(e)))
                            break
                        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
                        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
                        assert_bitwise(out, c.cgrid_expected(icall, nsub), f"{name} call {icall} nsub {nsub} (C grid, resident {forced})")
                        res = core.cgrid_timings()["resident_subcycles"]
                        if forced == "1" and nsub >= 4:
                            # forced on, the kernel runs -- the seabed-stress fixtures too (round 6: the SLOW variant reads the
                            # six operands the default-configuration short cuts drop from their arrays at level C)
                            assert res == nsub - 1, (name, nsub, res)
                            ran.append(name)
                        if forced == "0":
                            assert res == 0
                    else:
                        continue
                    break
            finally:
                core.finalize()
    print("resident C-grid kernel ran on:", sorted(set(ran)), "refused:", sorted(set(r[0] for r in refused)))
    assert len(set(ran)) >= 1, (ran, refused)
    assert any(n.startswith("cgrid_trip_") for n in ran), (ran, refused)      # ... the FOLD variant among them
    assert any("seabed" in n for n in ran), (ran, refused)                    # ... and the SLOW one (seabed stress)


@pytest.mark.parametrize("case", ["caps", "full"])
def test_cgrid_resident_kernel_survives_lagging_windows(case, monkeypatch):
    """The two-buffer record scheme of the resident C-grid kernel must not depend on windows keeping pace by luck: with every
    fourth window delayed by 10 us per subcycle (test build, CICE_EVP_HIP_CGRID_RES_DEBUG=8) -- next to open water in the 'caps'
    case -- the result is still the oracle's, bit for bit, and no wait gives up.  Revised EVP rides along in the 'full' case."""
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_RESIDENT", "1")
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_RES_DEBUG", "8")
    dc, g, static, state, inputs, masks = synth_cgrid("gx3", case=case, seed=23)
    got, want = run_both(dc, g, static, state, inputs, masks, 40, scal_kw=(dict(revised_evp=True) if case == "full" else None))
    assert_bitwise(got, want, f"C grid, resident kernel with lagging windows, {case}")
    assert np.abs(want["uvelE"]).max() > 1e-4


@pytest.mark.parametrize("bs,case,visc,lag,revised", [(None, "full", "avg_zeta", False, False), ((90, 60), "caps", "avg_strength", True, False),
                                                      ((360, 46), "full", "avg_zeta", True, False), ((75, 240), "caps", "avg_zeta", False, False),
                                                      ((180, 120), "full", "avg_strength", False, True), (None, "caps", "avg_strength", False, True),
                                                      # (revised EVP with avg_zeta: the variant with the most LDS holds two workgroups per CU -- a smaller grid)
                                                      ("tx3", "full", "avg_zeta", False, True), ("tx3", "caps", "avg_zeta", True, False)])
def test_cgrid_resident_kernel_on_a_tripole_grid_vs_oracle(bs, case, visc, lag, revised, monkeypatch):
    """tx1 (360 x 240, u-fold): the resident kernel's FOLD variant forced on -- the windows at the fold carry a mirrored mini-tile in
    source orientation and build every value ON the fold (vvelN, uvelN, uvelU, vvelU, shearU, stress12U) from both sides' raw values
    -- bit-identical to the oracle, ghost cells included: one block, blocks cut in both directions (the mirrored cells of a window in
    another block; a block at the fold with fewer than eleven rows), the poles inside a window and at a window's edge; with every
    fourth window lagging (test hook) the same bits; classic and revised EVP.  The first subcycle of the call runs as the five phases."""
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_RESIDENT", "1")
    if lag:
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_RES_DEBUG", "8")
    from cice_amd import synth
    grid = "tx1"
    if bs == "tx3":
        grid, bs = "tx3", None
    dc, g, static, state, inputs, masks = synth_cgrid(grid, case=case, bs=bs, seed=41)
    ndte = 24
    scal = synth.evp_scalars(120, **(dict(revised_evp=True) if revised else {}))
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    want = oracle.cgrid_subcycle(dom, prm, ndte, state, inputs, static, masks, visc_method=visc)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        got = core.cgrid_run(ndte, state, inputs, masks, visc_method=visc)
        t = core.cgrid_timings()
        assert t["resident_subcycles"] == ndte - 1 and t["resident_fallbacks"] == 0, t
        assert_bitwise(got, want, f"C grid, tripole, resident kernel, blocks {bs}, {case}, {visc}, lag {lag}")
        # the same state again without an upload in between: every subcycle of the call inside the launch
        core.cgrid_subcycle(6)
        got2 = core.cgrid_download()
        assert core.cgrid_timings()["resident_subcycles"] == 6
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_RESIDENT", "0")
    finally:
        core.finalize()
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        core.cgrid_upload(state, inputs, masks, visc_method=visc)
        core.cgrid_subcycle(ndte)
        core.cgrid_subcycle(6)
        ref2 = core.cgrid_download()
        assert core.cgrid_timings()["resident_subcycles"] == 0
    finally:
        core.finalize()
    assert_bitwise(got2, ref2, "C grid, tripole: a second call on the device state, resident against five phases")
    assert np.abs(want["uvelE"]).max() > 1e-4


def test_cgrid_resident_kernel_runs_only_the_windows_with_ice(monkeypatch):
    """720 x 270 (1176 windows: more than the chip holds) with ice on the polar caps only: the windows none of whose positions carry ice
    do not run and are not polled (cg_res_live at every upload), the 659 that do fit the chip -- the resident kernel takes the call,
    bit-identical to the oracle and to the same call with every window forced to run (test switch) or on the one-launch kernel."""
    from cice_amd import synth
    dc, g, static, state, inputs, masks = synth_cgrid("q8", case="caps", seed=47)
    ndte = 12
    scal = synth.evp_scalars(120)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    want = oracle.cgrid_subcycle(dom, prm, ndte, state, inputs, static, masks, visc_method="avg_zeta")
    d, keep = evp.make_dims(dc, 0)

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        try:
            core.cgrid_set_geometry(static)
            out = core.cgrid_run(ndte, state, inputs, masks)
            return out, core.cgrid_timings()
        finally:
            core.finalize()
            for k in env:
                monkeypatch.delenv(k)

    got, t = run(CICE_EVP_HIP_CGRID_RESIDENT="1")
    assert t["resident_windows"] > 768 and 0 < t["resident_windows_with_ice"] <= 768, t
    assert t["resident_subcycles"] == ndte - 1 and t["resident_fallbacks"] == 0, t
    assert_bitwise(got, want, "C grid, 720 x 270 caps, resident kernel on the windows with ice")
    off, t0 = run(CICE_EVP_HIP_CGRID_RESIDENT="0")
    assert t0["resident_subcycles"] == 0 and t0["one_launch_subcycles"] > 0
    assert_bitwise(off, want, "C grid, 720 x 270 caps, one-launch kernel")
    # every window forced to run: too many for the chip.  Demanding the kernel does not make that an error (round-5 advice: only what
    # can never change -- tables, geometry, rank layout -- is; how many windows a call wants resident is a condition of the call):
    # the call goes through the one-launch kernel, as it would once the resident kernel had run
    allw, ta = run(CICE_EVP_HIP_CGRID_RESIDENT="1", CICE_EVP_HIP_CGRID_RES_CULL="0")
    assert ta["resident_subcycles"] == 0 and ta["one_launch_subcycles"] > 0, ta
    assert_bitwise(allw, want, "C grid, 720 x 270 caps, resident kernel demanded with every window: fell back")


@pytest.mark.parametrize("grid", ["gx3", "tx1"])
def test_cgrid_run_recovers_when_a_window_is_not_resident(grid, monkeypatch):
    """(gx3; tx1: the FOLD variant, the repeat runs as five phases + fold steps)  cice_evp_hip_cgrid_run on a GPU that is not the rank's alone: one window of the resident kernel never shows up (test
    hook, real launches only), the waits on its records give up (bounded), nothing is written back, the download leaves the
    caller's arrays alone -- the call is repeated with the per-subcycle kernels and returns the oracle's answer; later calls
    stay off the resident kernel."""
    from cice_amd import synth
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_RES_DEBUG", "16")
    dc, g, static, state, inputs, masks = synth_cgrid(grid, case="full", seed=29)
    scal = synth.evp_scalars(120)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    want = oracle.cgrid_subcycle(dom, prm, 24, state, inputs, static, masks, visc_method="avg_zeta")
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        got = core.cgrid_run(24, state, inputs, masks)
        t = core.cgrid_timings()
        assert t["resident_fallbacks"] == 1 and t["resident_subcycles"] == 0, t
        assert_bitwise(got, want, "C grid after the fall-back")
        got = core.cgrid_run(24, state, inputs, masks)
        t = core.cgrid_timings()
        assert t["resident_fallbacks"] == 1 and t["resident_subcycles"] == 0, t
        assert_bitwise(got, want, "C grid, next call")
    finally:
        core.finalize()


def test_cgrid_split_calls_equal_one_call():
    """upload / subcycle(a) / subcycle(b) / download == run(a + b): the resident state carries over, and the
    one-off zero fill of the first subcycle is not repeated."""
    c = GoldenCase("cgrid_cyc_2x2_patchy")
    core = cgrid_core(c)
    try:
        state, inputs, masks = c.cgrid_inputs(1)
        core.cgrid_upload(state, inputs, masks)
        core.cgrid_subcycle(1)
        core.cgrid_subcycle(1)
        core.cgrid_subcycle(118)
        out = core.cgrid_download()
        dom = c.oracle_domain()
        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
        assert_bitwise(out, c.cgrid_expected(1, 120), "split calls")
        t = core.cgrid_timings()
        assert t["nsub"] == 118 and t["loop_ms"] > 0
    finally:
        core.finalize()


def test_cgrid_fails_loudly_without_geometry():
    c = GoldenCase("trip_cyc_1blk_patchy")
    d, keep = c.hip_dims()
    core = evp.EvpHip(d, evp.make_params(c.scal_dict(), strict=True), c.d["HTE"], c.d["HTN"], c.d["dxT"], c.d["dyT"],
                      c.d["uarear"], c.d["tarea"], keepalive=keep)
    try:
        z = np.zeros(core.shape)
        with pytest.raises(evp.EvpHipError, match="geometry not set"):
            core.cgrid_upload({k: z for k in evp.CGRID_FIELDS}, {k: z for k in evp.CGRID_INPUTS},
                              {k: z.astype(np.int32) for k in evp.CGRID_MASKS})
    finally:
        core.finalize()


def synth_cgrid(grid_name, case="full", bs=None, seed=3, seabed=False):
    from cice_amd import decomp, synth
    spec = synth.GRIDS[grid_name]
    ns = spec.get("ns", "closed")
    g = synth.derive_geometry(synth.make_grid(spec["nx"], spec["ny"], spec["dx0"], ns=ns))
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case=case, seed=seed, seabed=seabed)
    nx, ny = spec["nx"], spec["ny"]
    bs = bs or (nx, ny)
    dc = decomp.Decomp(nx, ny, bs[0], bs[1], "cyclic", ns, 1)
    static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
    return dc, g, static, state, inputs, masks


def run_both(dc, g, static, state, inputs, masks, ndte, visc_method="avg_zeta", scal_kw=None, scal_over=None, info=None):
    from cice_amd import synth
    scal = synth.evp_scalars(120, **(scal_kw or {}))
    scal.update(scal_over or {})
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    want = oracle.cgrid_subcycle(dom, prm, ndte, state, inputs, static, masks, visc_method=visc_method)
    d, keep = evp.make_dims(dc, 0)
    ua = static["uarea"]
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / ua, static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        got = core.cgrid_run(ndte, state, inputs, masks, visc_method=visc_method)
        if info is not None:
            info.update(core.cgrid_timings())
    finally:
        core.finalize()
    return got, want


@pytest.mark.parametrize("grid,bs,case,visc,seabed", [("gx3", None, "full", "avg_zeta", False),
                                                      ("gx3", (30, 40), "caps", "avg_strength", True),   # padded blocks
                                                      ("gx1", None, "full", "avg_zeta", True),
                                                      ("gx1", (80, 96), "caps", "avg_zeta", False),
                                                      # tripole: N faces with ice ON the fold, fold step after every phase
                                                      ("tx1", None, "full", "avg_zeta", True),
                                                      ("tx1", (90, 60), "full", "avg_strength", False)])
def test_cgrid_synthetic_vs_oracle_bitwise(grid, bs, case, visc, seabed):
    """Seeded synthetic workloads at the BASELINE sizes, one block and several (padded) blocks: every array of the
    loop equal to the oracle's, bit for bit, after 24 subcycles."""
    args = synth_cgrid(grid, case=case, bs=bs, seabed=seabed)
    got, want = run_both(*args, ndte=24, visc_method=visc)
    assert_bitwise(got, want, f"C grid {grid} {bs} {case} {visc}")
    assert np.abs(want["uvelE"]).max() > 1e-3 and np.isfinite(want["stresspT"]).all()
    assert int(args[5]["iceEmask"].sum()) > 0.2 * args[5]["iceEmask"].size * (0.3 if case == "caps" else 1.0)


@pytest.mark.parametrize("grid,bs,visc,revised", [("gx3", None, "avg_zeta", False), ("gx3", (50, 58), "avg_strength", True),
                                                   ("tx1", None, "avg_zeta", False), ("tx1", (90, 60), "avg_strength", True)])
def test_cgrid_resident_kernel_general_momentum_step(grid, bs, visc, revised, monkeypatch):
    """The resident C-grid kernel's SLOW variant (round 6): every operand the default-configuration short cuts drop is given a
    value of its own -- seabed stress TbE / TbN on a third of the cells, rheofact 0 on a tenth, waterxE / wateryN turned against
    the ocean currents, an ocean turning angle in the scalars -- the kernel is demanded (CICE_EVP_HIP_CGRID_RESIDENT=1) and must
    have run all subcycles but the first; every array of the loop against the oracle, bit for bit (closed and tripole grids, one
    block and several, both visc_methods, classic and revised EVP)."""
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_RESIDENT", "1")
    dc, g, static, state, inputs, masks = synth_cgrid(grid, case="full", bs=bs, seed=31, seabed=True)
    rng = np.random.default_rng(31)
    inputs = {k: v.copy() for k, v in inputs.items()}
    ang = np.deg2rad(17.0)
    cw, sw = float(np.cos(ang)), float(np.sin(ang))
    for t, wat in (("E", "waterxE"), ("N", "wateryN")):
        shp = inputs[f"Tb{t}"].shape
        inputs[f"rheofact{t}"] = inputs[f"rheofact{t}"] * (rng.random(shp) > 0.1)
        inputs[f"Tb{t}"] = np.where(rng.random(shp) > 0.66, rng.uniform(0.0, 0.5, shp), 0.0) * (inputs[f"ai{t}"] > 0)
    # the reference's waterx / watery with a turning angle (ice_dyn_shared.F90:819-820) at the E and N points
    inputs["waterxE"] = inputs["uocnE"] * cw - inputs["vocnE"] * sw
    inputs["wateryN"] = inputs["vocnN"] * cw + inputs["uocnN"] * sw
    # (the ghost cells of the perturbed fields must stay images of the cells they mirror: the synthetic inputs are scattered from
    # global fields, so redo that for the ones drawn per block above)
    from cice_amd import synth
    for k in ("rheofactE", "rheofactN", "TbE", "TbN"):
        loc = next((l for l, names in synth.CGRID_LOC.items() if k in names), "center")
        inputs[k] = dc.scatter(dc.gather({0: inputs[k]}), 0, fill=0.0, fold=(loc, 1.0))
    kw = dict(revised_evp=True, arlx=300.0, brlx=300.0) if revised else {}
    scal = synth.evp_scalars(120, **kw)
    scal.update(cosw=cw, sinw=sw)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    ndte = 24
    want = oracle.cgrid_subcycle(dom, prm, ndte, state, inputs, static, masks, visc_method=visc)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        got = core.cgrid_run(ndte, state, inputs, masks, visc_method=visc)
        t = core.cgrid_timings()
    finally:
        core.finalize()
    assert t["resident_subcycles"] == ndte - 1 and t["resident_fallbacks"] == 0, t
    assert_bitwise(got, want, f"C grid {grid} {bs} {visc}: resident kernel, general momentum step")
    assert np.abs(want["taubxE"]).max() > 0 and np.abs(want["uvelE"]).max() > 1e-3


def test_cgrid_both_schedules_agree(monkeypatch):
    """The five-launch schedule (CICE_EVP_HIP_CGRID_FUSED=0) and the fused three-launch one give the same bits
    (the golden tests run the default; this one pins the other against the same fixture)."""
    c = GoldenCase("cgrid_cyccyc_2x2_cap0_ktens")
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_FUSED", "0")
    core = cgrid_core(c)
    try:
        state, inputs, masks = c.cgrid_inputs(1)
        dom = c.oracle_domain()
        for nsub in c.nsub_list:
            out = core.cgrid_run(nsub, state, inputs, masks)
            oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
            oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
            assert_bitwise(out, c.cgrid_expected(1, nsub), f"five-phase schedule nsub {nsub}")
    finally:
        core.finalize()


@pytest.mark.parametrize("name", ["cgrid_cyccyc_2x2_cap0_ktens", "cgrid_closed_2x2_revp", "cgrid_cyc_2x2_patchy",
                                  "cgrid_cyc_1blk_seabed", "cgrid_cyc_3x2pad_cap05_avgstrength"])
@pytest.mark.parametrize("one", ["0", "1", "1:0", "1:2"])
def test_cgrid_one_launch_schedule_agrees(name, one, monkeypatch):
    """One launch per subcycle (cg_one: three levels in one workgroup, neighbours recomputed, five arrays ping-pong)
    against the three-launch form of the fused schedule, both pinned on the reference's arrays -- in one call and in
    split calls with odd counts (the buffers change roles between calls)."""
    c = GoldenCase(name)
    one, _, shape = one.partition(":")          # "1:2": the 64x16 window (the default on these small grids is 32x8)
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_ONE", one)
    if shape:
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_ONE_SHAPE", shape)
    core = cgrid_core(c)
    try:
        state, inputs, masks = c.cgrid_inputs(1)
        dom = c.oracle_domain()
        nsub = max(c.nsub_list)
        visc = str(c.d["visc_method"])
        out = core.cgrid_run(nsub, state, inputs, masks, visc_method=visc)
        t = core.cgrid_timings()
        # (the on-chip resident kernel takes the one-launch kernel's subcycles where it is eligible: classic EVP, avg_zeta, defaults)
        assert t["one_launch_subcycles"] + t["resident_subcycles"] == ((nsub - 1) if one == "1" else 0), t
        assert t["geometry_derived"], "the reference's own start-up arrays satisfy the identities the derived view rests on"
        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
        assert_bitwise(out, c.cgrid_expected(1, nsub), f"CICE_EVP_HIP_CGRID_ONE={one} nsub {nsub}")
        core.cgrid_upload(state, inputs, masks, visc_method=visc)
        done = 0
        for k in (1, 3, 2, nsub - 6):
            core.cgrid_subcycle(k)
            done += k
        assert done == nsub
        out = core.cgrid_download()
        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
        assert_bitwise(out, c.cgrid_expected(1, nsub), f"CICE_EVP_HIP_CGRID_ONE={one} split calls")
    finally:
        core.finalize()


@pytest.mark.parametrize("name", ["cgrid_cyccyc_2x2_cap0_ktens", "cgrid_closed_2x2_revp", "cgrid_cyc_1blk_seabed",
                                  "cgrid_cyc_3x2pad_cap05_avgstrength"])
@pytest.mark.parametrize("shape", ["0", "2", "three_launches"])
def test_cgrid_all_static_arrays_loaded_agrees(name, shape, monkeypatch):
    """cg_one and the three fused kernels derive 15 of the 23 static arrays from the eight dx / dy arrays by default (the other
    tests); with CICE_EVP_HIP_CGRID_GEO=0 they load all 23, as before: the same bits, against the reference's arrays."""
    c = GoldenCase(name)
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_GEO", "0")
    if shape == "three_launches":
        if str(c.d["visc_method"]) != "avg_zeta":
            pytest.skip("avg_strength without cg_one runs the five phases, which always load")
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_ONE", "0")
    else:
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_ONE_SHAPE", shape)
    core = cgrid_core(c)
    try:
        state, inputs, masks = c.cgrid_inputs(1)
        dom = c.oracle_domain()
        nsub = max(c.nsub_list)
        visc = str(c.d["visc_method"])
        out = core.cgrid_run(nsub, state, inputs, masks, visc_method=visc)
        t = core.cgrid_timings()
        assert t["one_launch_subcycles"] == (0 if shape == "three_launches" else nsub - 1) and not t["geometry_derived"], t
        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
        assert_bitwise(out, c.cgrid_expected(1, nsub), "CICE_EVP_HIP_CGRID_GEO=0")
    finally:
        core.finalize()


@pytest.mark.parametrize("what", ["DminTarea", "earear", "tarea_ghost", "ratiodyEr", "uvm"])
def test_cgrid_derived_geometry_is_refused_when_an_identity_fails(what, capfd, monkeypatch):
    """One static array off the reference's start-up formula by one bit in one cell (an interior cell, a ghost cell; a
    land mask that is neither 0 nor 1): the library keeps all 23 arrays in use, says why, and still equals the oracle
    run on the same arrays."""
    monkeypatch.setenv("CICE_EVP_HIP_VERBOSE", "1")
    dc, g, static, state, inputs, masks = synth_cgrid("gx3", bs=(50, 58))
    static = {k: v.copy() for k, v in static.items()}
    if what == "tarea_ghost":
        static["tarea"][1, 0, 7] = np.nextafter(static["tarea"][1, 0, 7], np.inf)
    elif what == "uvm":
        static["uvm"][2, 20, 20] = 0.5
    else:
        static[what][3, 30, 25] = np.nextafter(static[what][3, 30, 25], np.inf)
    scal = __import__("cice_amd.synth", fromlist=["x"]).evp_scalars(120)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    want = oracle.cgrid_subcycle(dom, prm, 9, state, inputs, static, masks)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        got = core.cgrid_run(9, state, inputs, masks)
        t = core.cgrid_timings()
    finally:
        core.finalize()
    assert t["one_launch_subcycles"] == 8 and not t["geometry_derived"], t
    err = capfd.readouterr().err
    assert "all 23 static arrays stay in use" in err and "differs from the reference's start-up formula" in err, err
    assert_bitwise(got, want, f"identity broken: {what}")


def test_cgrid_derived_geometry_on_the_synthetic_workloads():
    """The bench's synthetic block arrays follow the reference's start-up formulas (cice_amd/synth.py: cgrid_scatter), ghost
    cells included, so the bench measures the derived view -- several blocks, closed and cyclic edges."""
    for bs in (None, (50, 58)):
        args = synth_cgrid("gx3", bs=bs)
        dc, static = args[0], args[2]
        d, keep = evp.make_dims(dc, 0)
        scal = __import__("cice_amd.synth", fromlist=["x"]).evp_scalars(120)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        try:
            core.cgrid_set_geometry(static)
            assert core.cgrid_timings()["geometry_derived"], bs
        finally:
            core.finalize()


def random_cgrid_case(seed, nx, ny, bs, ew, ns, holes):
    """Arbitrary (not physical) operands: random positive lengths / areas, random velocities, stresses and forcing,
    and -- the point -- random, mutually independent ice masks with holes, so that every mask-dependent branch of
    the kernels (ice flags of ghost corners, pushes, fold entries without ice, T-cells of the extra row) is hit."""
    from cice_amd import decomp
    rng = np.random.default_rng(seed)
    dc = decomp.Decomp(nx, ny, bs[0], bs[1], ew, ns, 1)
    pos = lambda lo, hi: rng.uniform(lo, hi, (ny, nx))
    sym = lambda a: rng.uniform(-a, a, (ny, nx))
    hm = (rng.random((ny, nx)) > 0.08).astype(np.float64)
    hm[:1, :] = 0.0
    if ns != "tripole":
        hm[-1:, :] = 0.0
    e = lambda a: np.roll(a, -1, axis=1)
    n = lambda a: np.vstack([a[1:], a[-1:, ::-1] if ns == "tripole" else np.zeros((1, nx))])
    cg = {k: pos(0.8e5, 1.2e5) for k in ("dxT", "dyT", "dxU", "dyU", "dxE", "dyE", "dxN", "dyN")}
    for a in ("uarea", "tarea", "earea", "narea"):
        cg[a] = pos(0.8e10, 1.2e10)
    cg["earear"], cg["narear"] = 1.0 / cg["earea"], 1.0 / cg["narea"]
    cg["hm"] = hm
    cg["epm"], cg["npm"] = np.minimum(hm, e(hm)), np.minimum(hm, n(hm))
    cg["uvm"] = np.minimum(np.minimum(hm, e(hm)), np.minimum(n(hm), e(n(hm))))
    cg["DminTarea"] = 1e-11 * cg["tarea"]
    for r in ("ratiodxN", "ratiodyE"):
        cg[r] = -pos(0.9, 1.1)
        cg[r + "r"] = 1.0 / cg[r]
    ice = lambda m: ((m > 0.5) & (rng.random((ny, nx)) > holes)).astype(np.int32)
    masks = {"iceTmask": ice(hm), "iceUmask": ice(cg["uvm"]), "iceEmask": ice(cg["epm"]), "iceNmask": ice(cg["npm"])}
    state = {k: sym(0.3) for k in ("uvelE", "vvelE", "uvelN", "vvelN", "uvel", "vvel")}
    state.update({k: sym(2e3) * masks["iceTmask"] for k in ("stresspT", "stressmT", "stress12T")})
    state["stress12U"] = sym(2e3) * masks["iceUmask"]
    state.update({k: sym(0.05) for k in ("strintxE", "strintyN", "taubxE", "taubyN")})
    inputs = {"strength": pos(1e3, 4e4)}
    for t in "EN":
        inputs.update({f"cdn_ocn{t}": pos(0.004, 0.007), f"ai{t}": pos(0.2, 1.0), f"uocn{t}": sym(0.2), f"vocn{t}": sym(0.2),
                       f"fm{t}": sym(0.05), f"Tb{t}": pos(0.0, 0.5) * (rng.random((ny, nx)) > 0.7),
                       f"rheofact{t}": (rng.random((ny, nx)) > 0.1).astype(np.float64)})
    inputs.update(waterxE=sym(0.2), wateryN=sym(0.2), forcexE=sym(0.1), forceyN=sym(0.1), emassdti=pos(0.1, 0.6),
                  nmassdti=pos(0.1, 0.6), uvelE_init=sym(0.3), vvelN_init=sym(0.3))
    from cice_amd import synth
    return (dc, None) + synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)


@pytest.mark.parametrize("seed,nx,ny,bs,ew,ns,holes,visc,revised", [
    (1, 70, 40, (70, 40), "cyclic", "closed", 0.3, "avg_zeta", False),
    (2, 70, 40, (24, 14), "cyclic", "closed", 0.6, "avg_zeta", True),        # padded blocks, revised EVP
    (3, 50, 30, (25, 10), "closed", "closed", 0.3, "avg_strength", False),
    (4, 48, 36, (16, 12), "cyclic", "cyclic", 0.5, "avg_zeta", False),
    (5, 64, 30, (64, 30), "cyclic", "tripole", 0.4, "avg_zeta", False),      # ice and holes ON the fold
    (6, 48, 24, (12, 8), "cyclic", "tripole", 0.5, "avg_strength", True),
    (7, 40, 30, (40, 30), "cyclic", "closed", 1.1, "avg_zeta", False),       # no ice at all
    # fold lists of 1204 and 2404 entries per field: the two-entries-per-thread fold kernel and the one that stages in memory
    # (the cases above and every tripole fixture take the one-entry-per-thread kernel; evp_cgrid.hip: cg_fold_reg / cg_fold_one)
    (8, 600, 10, (600, 10), "cyclic", "tripole", 0.4, "avg_zeta", False),
    (9, 1200, 8, (1200, 8), "cyclic", "tripole", 0.4, "avg_zeta", False),
])
def test_cgrid_random_masks_vs_oracle_bitwise(seed, nx, ny, bs, ew, ns, holes, visc, revised):
    args = random_cgrid_case(seed, nx, ny, bs, ew, ns, holes)
    kw = dict(revised_evp=True, arlx=300.0, brlx=300.0) if revised else {}
    got, want = run_both(*args, ndte=7, visc_method=visc, scal_kw=kw)
    assert_bitwise(got, want, f"random masks seed {seed}")
    assert np.isfinite(want["uvelE"]).all() and np.isfinite(want["stresspT"]).all()
    if holes < 1.0:
        assert np.abs(want["uvelE"] - args[3]["uvelE"]).max() > 0


def stir_momentum(inputs, masks, rng):
    """Leave the default configuration: an ocean turning angle's waterx / watery, rheofact of 0 on some faces (seabed stress comes from
    synth.cgrid_state(seabed=True)) -- the general momentum step of every kernel."""
    shape = inputs["waterxE"].shape
    inputs["waterxE"] = inputs["waterxE"] * (1.0 + 0.1 * rng.random(shape))
    inputs["wateryN"] = inputs["wateryN"] * (1.0 - 0.1 * rng.random(shape))
    for t in "EN":
        inputs[f"rheofact{t}"] = inputs[f"rheofact{t}"] * (rng.random(shape) > 0.1)


def marched_case(seed, nx, ny, bs, case, holes, land, general=False):
    """A synthetic workload in the default configuration (the reference's start-up identities hold for the geometry, waterx == uocn,
    Tb == 0, rheofact == 1 on ice: what the marched kernel is for) with the branches stirred up: random land cells inside the ocean
    (coastal corners: the boundary-condition ratios), random, mutually independent holes in the four ice masks."""
    from cice_amd import decomp, synth
    rng = np.random.default_rng(seed)
    g0 = synth.make_grid(nx, ny, 2.0e4, ns="closed")
    g0["kmt"] = g0["kmt"] * (rng.random((ny, nx)) >= land)
    g = synth.derive_geometry(g0)
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case=case, seed=seed, seabed=general)
    if general:
        stir_momentum(inputs, masks, rng)
    for k in masks:
        masks[k] = masks[k] * (rng.random((ny, nx)) >= holes).astype(np.int32)
    for k in ("stresspT", "stressmT", "stress12T"):
        state[k] = state[k] * masks["iceTmask"]
    state["stress12U"] = state["stress12U"] * masks["iceUmask"]
    dc = decomp.Decomp(nx, ny, bs[0], bs[1], "cyclic", "closed", 1)
    return (dc, g) + synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)


@pytest.mark.parametrize("seed,nx,ny,bs,case,holes,land,revised,seg,shape", [
    (21, 200, 48, (200, 48), "full", 0.2, 0.02, False, None, "2"),       # one block: 64 x 16 windows, default segments
    (22, 330, 40, (330, 40), "caps", 0.4, 0.05, True, "7", "1"),         # 64 x 8 windows, segments of 7 rows, revised EVP
    (23, 400, 64, (200, 32), "full", 0.3, 0.0, False, "5", "1"),         # 2 x 2 blocks, each with a marched interior
    (24, 140, 90, (140, 90), "full", 1.1, 0.0, False, "1", "2"),         # no ice at all; one row per segment
    (25, 260, 72, (260, 72), "full", 0.0, 0.1, False, "64", "2"),        # ice everywhere, many islands; one segment per strip
    (26, 190, 50, (190, 50), "caps", 0.1, 0.01, False, "3", "2"),        # the last strip shifted west (68 columns: 60 + 8)
    # the rectangle's top window row would read dyU of row ny_global, which the reference extrapolates: that row goes back to cg_one
    (27, 200, 46, (200, 46), "full", 0.2, 0.02, False, None, "2"),
    # seabed stress, an ocean turning angle's waterx, rheofact = 0 on some faces: the general momentum step (classic / revised EVP)
    (28, 280, 60, (280, 60), "full", 0.2, 0.03, False, "9", "2"),
    (29, 300, 44, (150, 44), "caps", 0.3, 0.0, True, None, "1"),
    # visc_method = avg_strength (deltaU, a row late, feeds the corner viscosities of every subcycle); the second with the general momentum step
    (30, 240, 52, (240, 52), "full", 0.2, 0.03, False, None, "2"),
    (31, 300, 40, (150, 40), "caps", 0.1, 0.02, True, "6", "2"),
])
def test_cgrid_marched_interior_vs_oracle_bitwise(seed, nx, ny, bs, case, holes, land, revised, seg, shape, monkeypatch):
    """The one-launch schedule with the interior of each block marched (evp_cgrid.hip: cg_strip; the default on the 0.1-degree
    class only) forced onto small blocks: every array of the loop equal to the oracle's, bit for bit, and equal to the windowed
    kernel's -- the once-per-call arrays of the last subcycle (deltaU a row late) included; the first subcycle runs as before."""
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_RESIDENT", "0")
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_ONE_SHAPE", shape)
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP", "1")
    if seg:
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_SEG", seg)
    if seed in (22, 23):          # the windows kept along the edges: 64 x 8 / 64 x 16 instead of 32 x 8
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_EDGE", "1" if seed == 22 else "2")
    from cice_amd import synth
    dc, _, static, state, inputs, masks = marched_case(seed, nx, ny, bs, case, holes, land, general=seed in (28, 29, 31))
    kw = dict(revised_evp=True, arlx=300.0, brlx=300.0) if revised else {}
    scal = synth.evp_scalars(120, **kw)
    if seed in (28, 29, 31):
        scal.update(cosw=np.cos(0.4), sinw=np.sin(0.4))
    visc = "avg_strength" if seed in (30, 31) else "avg_zeta"
    d, keep = evp.make_dims(dc, 0)

    def run():
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        try:
            core.cgrid_set_geometry(static)
            out = core.cgrid_run(9, state, inputs, masks, visc_method=visc)
            return out, core.cgrid_timings()
        finally:
            core.finalize()
    if seed == 21:                # the last subcycle of the call by the windowed kernel, as before
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_LAST", "0")
    if seed == 25:                # the six lengths loaded, not formed in the kernel
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_LEN", "0")
    if seed == 26:                # a grid whose lengths are not the reference's means: the check refuses, all eight stay loaded
        static = dict(static)
        static["dxT"] = static["dxT"] * (1.0 + 1e-9)
        for k, (a, b_) in dict(tarea=("dxT", "dyT")).items():
            static[k] = static[a] * static[b_]
        static["DminTarea"] = scal["deltaminEVP"] * static["tarea"]
    got, tt = run()
    assert tt["marched_items"] > 0 and tt["one_launch_subcycles"] == 8 and tt["geometry_derived"], tt
    assert tt["marched_lengths_derived"] == (seed not in (25, 26)), tt
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    want = oracle.cgrid_subcycle(dom, prm, 9, state, inputs, static, masks, visc_method=visc)
    assert_bitwise(got, want, f"marched interior seed {seed}")
    if holes < 1.0:
        assert np.abs(want["uvelE"] - state["uvelE"]).max() > 0
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP", "0")
    windowed, tw = run()
    assert tw["marched_items"] == 0 and tw["one_launch_subcycles"] == 8, tw
    assert_bitwise(got, windowed, f"marched interior against the windowed kernel, seed {seed}")


@pytest.mark.parametrize("seed", [3101, 3102, 3103, 3104] + [int(s) for s in __import__("os").environ.get("CGRID_STRIP_SWEEP_SEEDS", "").split() if s])
def test_cgrid_marched_interior_random_cuts_vs_oracle(seed, monkeypatch):
    """A sweep over what shapes the marched kernel's plan: domain and block size (one to four blocks, padded or not, cyclic or closed
    in x), segment length, the shape of the windows kept, lengths formed or loaded, the last subcycle marched or not, ice cover, islands,
    classic / revised EVP -- all drawn from the seed; every array equal to the oracle's, bit for bit.  CGRID_STRIP_SWEEP_SEEDS adds seeds."""
    rng = np.random.default_rng(seed)
    nbx, nby = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    scale = int(__import__("os").environ.get("CGRID_STRIP_SWEEP_SCALE", "1"))       # (wide runs: larger blocks, several strips and segments)
    bx, by = int(rng.integers(130, 260)) * scale, int(rng.integers(24, 70)) * scale
    nx, ny = nbx * bx - int(rng.integers(0, 20)) * (nbx > 1), nby * by - int(rng.integers(0, 6)) * (nby > 1)
    case, holes, land = ("full", "caps")[int(rng.integers(0, 2))], float(rng.choice([0.0, 0.2, 0.5])), float(rng.choice([0.0, 0.03, 0.1]))
    revised = bool(rng.integers(0, 2))
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_RESIDENT", "0")
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_ONE_SHAPE", str(int(rng.integers(1, 3))))
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP", "1")
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_EDGE", str(int(rng.choice([0, 0, 0, 1, 2]))))
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_LEN", str(int(rng.integers(0, 2))))
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_LAST", str(int(rng.integers(0, 2))))
    if rng.integers(0, 2):
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_STRIP_SEG", str(int(rng.integers(1, 24))))
    from cice_amd import decomp, synth
    g0 = synth.make_grid(nx, ny, 2.0e4, ns="closed")
    g0["kmt"] = g0["kmt"] * (rng.random((ny, nx)) >= land)
    g = synth.derive_geometry(g0)
    cg = synth.cgrid_geometry(g)
    general = bool(rng.integers(0, 3) == 0)
    visc = ("avg_zeta", "avg_zeta", "avg_strength")[int(rng.integers(0, 3))]
    state, inputs, masks = synth.cgrid_state(g, cg, case=case, seed=seed, seabed=general)
    if general:
        stir_momentum(inputs, masks, rng)
    for k in masks:
        masks[k] = masks[k] * (rng.random((ny, nx)) >= holes).astype(np.int32)
    for k in ("stresspT", "stressmT", "stress12T"):
        state[k] = state[k] * masks["iceTmask"]
    state["stress12U"] = state["stress12U"] * masks["iceUmask"]
    dc = decomp.Decomp(nx, ny, bx, by, "cyclic", "closed", 1)
    static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
    kw = dict(revised_evp=True, arlx=300.0, brlx=300.0) if revised else {}
    info = {}
    got, want = run_both(dc, g, static, state, inputs, masks, ndte=int(rng.integers(3, 8)), scal_kw=kw, info=info, visc_method=visc,
                         scal_over=dict(cosw=np.cos(0.3), sinw=np.sin(0.3)) if general else None)
    assert_bitwise(got, want, f"marched interior, random cut {seed}: {nx} x {ny} in blocks of {bx} x {by}")
    # (windows of 64 positions along the edges leave blocks under ~190 columns without a rectangle: cg_one runs those alone)
    assert info["marched_items"] > 0 or __import__("os").environ["CICE_EVP_HIP_CGRID_STRIP_EDGE"] != "0", info
    print(f"STRIP_SWEEP seed {seed}: {nx} x {ny} / {bx} x {by}, items {info['marched_items']} x {info['marched_segment_rows']} rows, "
          f"cells {info['marched_cells']}, lengths formed {info['marched_lengths_derived']}, general momentum step {general}, {visc}")


def test_cgrid_default_configuration_shortcuts_are_bit_neutral(monkeypatch):
    """cg_stress_u_step<true> (taken when waterx == uocn, Tb == +0 and rheofact == 1 hold bit for bit on every ice
    cell of a call: the reference's default configuration) against the general kernel (CICE_EVP_HIP_CGRID_FAST=0) and
    the fixture, same bits; a call with seabed stress must not take it (it is compared with its fixture elsewhere)."""
    c = GoldenCase("cgrid_cyc_2x2_patchy")
    dom = c.oracle_domain()
    state, inputs, masks = c.cgrid_inputs(1)
    assert bits_equal(inputs["waterxE"][masks["iceEmask"] != 0], inputs["uocnE"][masks["iceEmask"] != 0])
    assert not inputs["TbE"].any() and (inputs["rheofactE"][masks["iceEmask"] != 0] == 1.0).all()
    outs = []
    for fast in ("1", "0"):
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_FAST", fast)
        core = cgrid_core(c)
        try:
            out = core.cgrid_run(120, state, inputs, masks)
        finally:
            core.finalize()
        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
        assert_bitwise(out, c.cgrid_expected(1, 120), f"CICE_EVP_HIP_CGRID_FAST={fast}")
        outs.append(out)
    assert_bitwise(outs[0], outs[1], "shortcut vs general kernel")


@pytest.mark.parametrize("transport", ["rccl", "direct"])
@pytest.mark.parametrize("name", ["cgrid_cyc_2x2_patchy", "cgrid_cyc_3x2pad_cap05_avgstrength"])
def test_cgrid_single_rank_self_exchange(name, transport, monkeypatch):
    """The C-grid loop's remote-halo path on one GPU: CICE_EVP_HIP_SELF_EXCHANGE routes every ghost copy through the
    exchange with the rank itself, so nothing is pushed and every exchange point of the schedule fills all ghost cells
    -- through pack -> ncclGroup{ncclSend, ncclRecv} -> unpack (rccl, enqueued eagerly) or through the mailbox kernel
    (direct, inside the captured graph).  Fused and five-phase schedules; same bits as the fixtures."""
    monkeypatch.setenv("CICE_EVP_HIP_SELF_EXCHANGE", "1")
    monkeypatch.setenv("CICE_EVP_HIP_HALO", transport)
    c = GoldenCase(name)
    dom = c.oracle_domain()
    d, keep = c.hip_dims()
    ua = c.d["uarea"]
    core = evp.EvpHip(d, evp.make_params(c.scal_dict(), strict=True), c.d["dyE"], c.d["dxN"], c.d["dxT"], c.d["dyT"],
                      np.where(ua > 0, 1.0 / np.where(ua > 0, ua, 1.0), 0.0), c.d["tarea"], keepalive=keep)
    try:
        core.comm_init(core.comm_unique_id())
        assert core.timings()["halo_transport"] == ("rccl" if transport == "rccl" else "mailbox")
        core.cgrid_set_geometry(c.cgrid_static())
        state, inputs, masks = c.cgrid_inputs(1)
        for nsub in (1, 120):
            out = core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
            oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
            oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
            assert_bitwise(out, c.cgrid_expected(1, nsub), f"{name} through {transport} self exchange, nsub {nsub}")
    finally:
        core.finalize()


def test_cgrid_s01_full_size_invariances(monkeypatch):
    """The 0.1-degree-class size (3600 x 2400 = 8.6M cells), 3 subcycles -- where the oracle would need a minute per
    run, size-independent properties instead: one block == 2 x 2 blocks with pushed ghost images == the five-launch
    schedule == 2 x 2 blocks with every ghost cell routed through the mailbox exchange (self exchange), bit for bit
    on every array of the loop (the bench line additionally checks this size against a committed oracle checksum)."""
    from cice_amd import synth
    keys = [k for k in evp.CGRID_FIELDS if k not in ("etax2U", "deltaU")]     # (never exchanged: ghost cells differ by layout)
    ref = None
    for bs, fused, selfx in ((None, "1", False), ((1800, 1200), "1", False), (None, "0", False), ((1800, 1200), "1", True)):
        monkeypatch.setenv("CICE_EVP_HIP_CGRID_FUSED", fused)
        if selfx:
            monkeypatch.setenv("CICE_EVP_HIP_SELF_EXCHANGE", "1")
            monkeypatch.setenv("CICE_EVP_HIP_HALO", "direct")
        dc, g, static, state, inputs, masks = synth_cgrid("s01", bs=bs, seed=2)
        d, keep = evp.make_dims(dc, 0)
        core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(480), strict=True), static["dyE"], static["dxN"], static["dxT"],
                          static["dyT"], 1.0 / static["uarea"], static["tarea"], keepalive=keep)
        try:
            if selfx:
                core.comm_init(core.comm_unique_id())
            core.cgrid_set_geometry(static)
            out = core.cgrid_run(3, state, inputs, masks)
        finally:
            core.finalize()
        glob = {k: dc.gather({0: out[k]}) for k in keys}
        del out, static, state, inputs
        assert np.isfinite(glob["uvelE"]).all() and np.abs(glob["uvelE"]).max() > 1e-4
        if ref is None:
            ref = glob
        else:
            assert_bitwise(glob, ref, f"C grid s01 blocks={bs} fused={fused} mailbox={selfx}")


def test_cgrid_many_small_blocks_and_zero_subcycles():
    """Edge cases of the layout: gx3 cut into 100 blocks of 10 x 12 cells (smaller than a 64-wide workgroup row, most
    ghost cells images of other blocks, padded blocks at the far edges) against the oracle; and ndte = 0 hands back
    exactly what was handed in (no launch, no ghost repair, work arrays zero as evp() leaves them at entry)."""
    args = synth_cgrid("gx3", case="caps", bs=(10, 12), seed=9, seabed=True)
    got, want = run_both(*args, ndte=6)
    assert_bitwise(got, want, "100 small blocks")
    assert args[0].nx_block == 12 and len(args[0].local_blocks(0)) == 100
    dc, g, static, state, inputs, masks = args
    from cice_amd import synth
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(120), strict=True), static["dyE"], static["dxN"], static["dxT"],
                      static["dyT"], 1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        out = core.cgrid_run(0, state, inputs, masks)
    finally:
        core.finalize()
    for k in evp.CGRID_FIELDS[:14]:
        assert bits_equal(out[k], state[k]), k
    for k in evp.CGRID_FIELDS[14:]:
        assert not out[k].any(), k


@pytest.mark.parametrize("seed", [61, 62, 63, 64, 65, 66])
def test_cgrid_random_scalars_vs_oracle(seed):
    """Property test over the EVP scalars on the C grid: capping in {0, 1, fractional}, Ktens, yield-curve ratios,
    classic / revised EVP, an ocean turning angle (cosw != 1, sinw != 0: no fixture has one), both visc methods, random
    operands and masks -- against the oracle, bit for bit."""
    rng = np.random.default_rng(seed)
    kw = dict(capping=float(rng.choice([0.0, 1.0, rng.uniform(0.1, 0.9)])), Ktens=float(rng.choice([0.0, rng.uniform(0.05, 0.5)])),
              e_yieldcurve=float(rng.uniform(1.2, 2.5)), e_plasticpot=float(rng.uniform(1.2, 2.5)))
    if rng.random() < 0.5:
        kw.update(revised_evp=True, arlx=float(rng.uniform(100, 400)), brlx=float(rng.uniform(100, 400)))
    over = {}
    if rng.random() < 0.6:
        ang = np.deg2rad(rng.uniform(5.0, 25.0))
        over = dict(cosw=float(np.cos(ang)), sinw=float(np.sin(ang)))
    ns = "tripole" if seed % 3 == 0 else "closed"
    args = random_cgrid_case(seed, 48, 30, (16, 10) if seed % 2 else (48, 30), "cyclic", ns, 0.3)
    got, want = run_both(*args, ndte=6, visc_method=("avg_strength" if seed % 2 else "avg_zeta"), scal_kw=kw, scal_over=over)
    assert_bitwise(got, want, f"C grid random scalars seed {seed}: {kw} {over} {ns}")
    assert np.isfinite(want["uvelE"]).all()


# ---- the preparation phase of evp() on the C grid, on the device (cice_evp_hip_cgrid_prep) ----------------------------

def hip_prep_params(c: GoldenCase):
    d = c.prep_scal_dict()
    return evp.PrepParams(dt=d["dt"], rhoi=d["rhoi"], rhos=d["rhos"], gravit=d["gravit"], dyn_area_min=d["dyn_area_min"],
                          dyn_mass_min=d["dyn_mass_min"], ssh_stress_coupled=d["ssh_coupled"])


def device_prep(core, c, icall, state, visc=None):
    """cice_evp_hip_cgrid_prep -> seabed factors -> _prep_finish for call `icall` of a fixture; returns the new masks."""
    t, st, _ = c.cgrid_prep_inputs(icall)
    masks_prev = {k: st[k] for k in ("iceUmask", "iceEmask", "iceNmask")}
    masks = core.cgrid_prep(hip_prep_params(c), t, state if state is not None else None, masks_prev)
    s = c.scal
    if s[23] != 0.0 and s[29] != 0.0:          # probabilistic (one thickness category in the harness)
        core.cgrid_seabed_prob(c.d["hwater"], t["aice"][:, None], t["vice"][:, None], s[26], s[17], s[19], s[30], s[31])
    elif s[23] != 0.0:
        core.cgrid_seabed_lkd(c.d["hwater"], s[24], s[25], s[26], s[27])
    core.cgrid_prep_finish(c.d[f"in{icall:02d}_strength"], visc or str(c.d["visc_method"]))
    return masks


@pytest.mark.parametrize("name", CGRID_CASES + CGRID_TFOLD_CASES)
def test_cgrid_prep_on_device_bitwise(name):
    """Everything the C-grid loop reads, computed on the device from the T-grid state and forcing (11 arrays in instead
    of 14 + 23) and compared with what the reference's own preparation left (the in* arrays of the fixtures): the four ice
    masks, the 12 state arrays and 22 per-call inputs, every cell, bit for bit (TbE / TbN: the device exp(), <= 4 ulp).
    Then the loop from that state against the reference's outputs.  Second call of three fixtures: cells gain / lose ice."""
    c = GoldenCase(name)
    dom = c.oracle_domain()
    core = cgrid_core(c)
    try:
        core.cgrid_set_prep_geometry(c.cgrid_prep_static())
        for icall in range(1, c.ncalls + 1):
            _, st, _ = c.cgrid_prep_inputs(icall)
            masks = device_prep(core, c, icall, {k: st[k] for k in oracle.C_FIELDS[:12]})
            want_state, want_in, want_masks = c.cgrid_inputs(icall)
            for k in oracle.C_MASKS:
                assert bits_equal(masks[k] != 0, want_masks[k] != 0), f"{name} call {icall} {k}"
            got = {k: core.cgrid_fetch(k) for k in list(want_state) + [k for k in want_in if k != "strength"]}
            oracle.halo_update(dom, got["strintxE"], "Eface", "vector")      # (the fixture's post-loop exchange, see above)
            oracle.halo_update(dom, got["strintyN"], "Nface", "vector")
            for k in ("TbE", "TbN"):
                w = want_in[k]
                # LKD: one exp() per face (<= 1 ulp); probabilistic: sums of 100 x 100 exp / log terms (a few ulp of the sum)
                assert np.allclose(got.pop(k), w, rtol=1e-15 if c.scal[29] == 0.0 else 1e-12, atol=0.0), k
                assert c.scal[23] == 0.0 or np.abs(w).max() > 0
            assert_bitwise(got, {k: v for k, v in {**want_state, **want_in}.items() if k in got},
                           f"{name} call {icall} (device preparation)")
            if c.scal[23] != 0.0:
                continue                      # the loop below would start from TbE / TbN that may differ in the last bit
            nsub = c.nsub_list[-1]
            core.cgrid_subcycle(nsub)
            out = core.cgrid_download()
            oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
            oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
            assert_bitwise(out, c.cgrid_expected(icall, nsub), f"{name} call {icall} nsub {nsub} (device preparation + loop)")
    finally:
        core.finalize()


def test_cgrid_prep_keeps_the_state_on_the_device_between_calls():
    """Second call with state12 = NULL: the velocities and stresses the first call's loop left on the device are the ones
    evp() would be entered with -- same inputs for the loop, same results, and nothing but the 11 T-grid arrays and the
    strength travelled in."""
    c = GoldenCase("cgrid_cyc_2x2_patchy")
    dom = c.oracle_domain()
    core = cgrid_core(c)
    try:
        core.cgrid_set_prep_geometry(c.cgrid_prep_static())
        _, st, _ = c.cgrid_prep_inputs(1)
        device_prep(core, c, 1, {k: st[k] for k in oracle.C_FIELDS[:12]})
        nsub = c.nsub_list[-1]
        core.cgrid_subcycle(nsub)                      # the reference's state before call 2 = after its last run of call 1
        masks = device_prep(core, c, 2, None)
        want_state, want_in, want_masks = c.cgrid_inputs(2)
        for k in oracle.C_MASKS:
            assert bits_equal(masks[k] != 0, want_masks[k] != 0), k
        got = {k: core.cgrid_fetch(k) for k in want_state}
        oracle.halo_update(dom, got["strintxE"], "Eface", "vector")
        oracle.halo_update(dom, got["strintyN"], "Nface", "vector")
        assert_bitwise(got, want_state, "call 2 from the resident state")
        core.cgrid_subcycle(nsub)
        out = core.cgrid_download()
        oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
        oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
        assert_bitwise(out, c.cgrid_expected(2, nsub), "call 2 from the resident state, loop")
    finally:
        core.finalize()


def test_cgrid_prep_fails_loudly_out_of_order():
    c = GoldenCase("cgrid_cyc_1blk_seabed")
    core = cgrid_core(c)
    try:
        t, st, _ = c.cgrid_prep_inputs(1)
        with pytest.raises(evp.EvpHipError, match="set_prep_geometry"):
            core.cgrid_prep(hip_prep_params(c), t, {k: st[k] for k in oracle.C_FIELDS[:12]}, st)
        core.cgrid_set_prep_geometry(c.cgrid_prep_static())
        with pytest.raises(evp.EvpHipError, match="first call must upload"):
            core.cgrid_prep(hip_prep_params(c), t, None, st)
        with pytest.raises(evp.EvpHipError, match="cgrid_prep first"):
            core.cgrid_prep_finish(c.d["in01_strength"])
    finally:
        core.finalize()


@pytest.mark.parametrize("grid,bs,case,coupled", [("gx1", None, "full", False), ("gx1", (80, 96), "caps", True),
                                                  ("tx1", (90, 60), "full", True), ("gx3", (25, 29), "caps", False)])
def test_cgrid_prep_synthetic_vs_oracle_bitwise(grid, bs, case, coupled):
    """The device preparation on gx1 / tx1 (tripole) / gx3-sized synthetic cases, one block and many, geostrophic and
    coupled sea-surface tilt, previous masks that make cells gain and lose ice: the four masks, the loop's 12 state
    arrays and 22 inputs against the oracle's restatement (itself pinned on the fixtures), every cell, bit for bit."""
    from cice_amd import decomp, synth
    spec = synth.GRIDS[grid]
    nx, ny = spec["nx"], spec["ny"]
    ns = spec.get("ns", "closed")
    g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=ns))
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case=case, seed=9, warm=True)
    t, st7, prev = synth.cgrid_prep_inputs(g, cg, case=case, seed=17, coupled=coupled)
    bsx, bsy = bs if bs else (nx, ny)
    dc = decomp.Decomp(nx, ny, bsx, bsy, "cyclic", ns, 1)
    static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
    tb = {k: dc.scatter(v, 0, fold=("center", -1.0 if k in ("uocn", "vocn", "ss_tltx", "ss_tlty", "strairxT", "strairyT") else 1.0))
          for k, v in t.items()}
    loc = {"umaskCD": "NEcorner", "emask": "Eface", "nmask": "Nface", "fcor_blk": "NEcorner", "fcorE_blk": "Eface", "fcorN_blk": "Nface"}
    static.update({k: dc.scatter(v, 0, fill=0, fold=(loc.get(k, "center"), 1.0)) for k, v in st7.items()})
    prevb = {k: dc.scatter(v, 0, fill=0) for k, v in prev.items()}
    scal = synth.evp_scalars(120)
    ppd = dict(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11, dyn_mass_min=1e-10)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    want = oracle.cgrid_prep(dom, oracle.PrepParams(**ppd, cosw=scal["cosw"], sinw=scal["sinw"], ssh_coupled=int(coupled)), static,
                             tb, dict(state, **prevb))
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        core.cgrid_set_prep_geometry(static)
        got = core.cgrid_prep(evp.PrepParams(**ppd, ssh_stress_coupled=int(coupled)), tb, state, prevb)
        for k in oracle.C_MASKS:
            assert bits_equal(got[k] != 0, want[k] != 0), k
        for k in ("iceEmask", "iceNmask"):          # the case must make faces gain and lose ice
            new, old = want[k] != 0, prevb[k] != 0
            assert (new & ~old).any() and (old & ~new).any(), k
        keys = oracle.C_FIELDS[:14] + [k for k in oracle.C_INPUTS if k != "strength"]
        assert_bitwise({k: core.cgrid_fetch(k) for k in keys}, {k: want[k] for k in keys}, f"{grid} {bs} device preparation vs oracle")
        assert np.abs(want["forcexE"]).max() > 0 and np.abs(want["uvelN"]).max() > 0
    finally:
        core.finalize()


@pytest.mark.parametrize("seed", list(range(201, 209)) + [936] + [int(s) for s in __import__("os").environ.get("CGRID_PREP_SWEEP_SEEDS", "").split() if s])
def test_cgrid_prep_and_loop_random_geometry_vs_oracle(seed):
    """Geometry sweep of the C-grid path from the T-grid state on: random domain sizes, block splits with padded last blocks,
    closed and tripole north boundaries, geostrophic / coupled tilt, previous masks that make faces gain and lose ice --
    device preparation against the oracle's (masks, 14 state arrays, 22 inputs), then 6 subcycles of the loop from the
    device-prepared state against the oracle's loop from the oracle-prepared one.  Bit for bit.
    (Seed 936: a corner on a block edge loses its ice -- its ghost image keeps the old stress12U until the reference's first
    exchange copies the zero over it; the five-launch schedule used to push ice cells only.)"""
    from cice_amd import decomp, synth
    rng = np.random.default_rng(seed)
    trip = seed % 3 == 0
    nx, ny = 2 * int(rng.integers(20, 70)), int(rng.integers(24, 80))
    nbx, nby = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    if trip:
        nby = 1 if ny < 40 else nby                 # (the blocks next to the fold hold at least two rows anyway)
    bsx, bsy = -(-nx // nbx), -(-ny // nby)
    ns = "tripole" if trip else "closed"
    coupled = bool(seed % 2)
    visc = "avg_strength" if seed % 4 == 1 else "avg_zeta"
    g = synth.derive_geometry(synth.make_grid(nx, ny, 9.0e4, ns=ns))
    cg = synth.cgrid_geometry(g)
    state, inputs, masks = synth.cgrid_state(g, cg, case="full", seed=seed, warm=True)
    t, st7, prev = synth.cgrid_prep_inputs(g, cg, case="full", seed=seed + 1000, coupled=coupled)
    dc = decomp.Decomp(nx, ny, bsx, bsy, "cyclic", ns, 1)
    static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
    vec = ("uocn", "vocn", "ss_tltx", "ss_tlty", "strairxT", "strairyT")
    tb = {k: dc.scatter(v, 0, fold=("center", -1.0 if k in vec else 1.0)) for k, v in t.items()}
    loc = {"umaskCD": "NEcorner", "emask": "Eface", "nmask": "Nface", "fcor_blk": "NEcorner", "fcorE_blk": "Eface", "fcorN_blk": "Nface"}
    static.update({k: dc.scatter(v, 0, fill=0, fold=(loc.get(k, "center"), 1.0)) for k, v in st7.items()})
    prevb = {k: dc.scatter(v, 0, fill=0) for k, v in prev.items()}
    scal = synth.evp_scalars(120)
    ppd = dict(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11, dyn_mass_min=1e-10)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), dc.nx_global, dc.ny_global, dc.ew, dc.ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    what = f"seed {seed}: {nx}x{ny} {ns}, blocks {bsx}x{bsy}, coupled {coupled}, {visc}"
    want = oracle.cgrid_prep(dom, oracle.PrepParams(**ppd, cosw=scal["cosw"], sinw=scal["sinw"], ssh_coupled=int(coupled)), static,
                             tb, dict(state, **prevb))
    strength = inputs["strength"]
    prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                      "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})
    win = {k: want[k] for k in oracle.C_INPUTS}
    win["strength"] = strength
    wloop = oracle.cgrid_subcycle(dom, prm, 6, {k: want[k] for k in oracle.C_FIELDS[:14]}, win, static,
                                  {k: want[k] for k in oracle.C_MASKS}, visc_method=visc)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                      1.0 / static["uarea"], static["tarea"], keepalive=keep)
    try:
        core.cgrid_set_geometry(static)
        core.cgrid_set_prep_geometry(static)
        got = core.cgrid_prep(evp.PrepParams(**ppd, ssh_stress_coupled=int(coupled)), tb, state, prevb)
        for k in oracle.C_MASKS:
            assert bits_equal(got[k] != 0, want[k] != 0), (what, k)
        keys = oracle.C_FIELDS[:14] + [k for k in oracle.C_INPUTS if k != "strength"]
        assert_bitwise({k: core.cgrid_fetch(k) for k in keys}, {k: want[k] for k in keys}, what + " (preparation)")
        core.cgrid_prep_finish(strength, visc)
        core.cgrid_subcycle(6)
        out = core.cgrid_download()
        assert_bitwise(out, wloop, what + " (loop from the device-prepared state)")
        assert np.abs(wloop["uvelE"]).max() > 1e-4
    finally:
        core.finalize()


# ---------------------------------------------------------------------------------------------------------------
# Geometry sweep pinned on the REFERENCE itself: the prebuilt harness (oracle/_ref, the reference's unmodified evp()
# with grid_ice = 'C') runs a random domain / block split / boundary kind / option set on the box, its dump becomes a
# fixture on the fly, and the loop, the device preparation and deformationsC_T are compared with the reference's own
# arrays exactly as for the committed fixtures.
# ---------------------------------------------------------------------------------------------------------------
def reference_cgrid_case(tmp_path, nx, ny, bs, ew, ns, **kw):
    import run_ref
    from cice_amd import synth
    import common
    if not run_ref.have_ref("strict"):
        pytest.skip("oracle/_ref/evp_ref_harness_strict not present")
    g = synth.make_grid(nx, ny, dx0=1.1e5, ns=("tripole" if ns == "tripoleT" else ns))
    run_ref.write_pop_grid(tmp_path / "grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
    run_ref.write_kmt(tmp_path / "kmt.bin", g["kmt"])
    d, txt = run_ref.run_harness(nx, ny, bs[0], bs[1], ew=ew, ns=ns, variant="strict", h_ndte=kw.pop("h_ndte", 120),
                                 grid_kind=("tripolefile" if ns in ("tripole", "tripoleT") else "popfile"),
                                 grid_files=(tmp_path / "grid.bin", tmp_path / "kmt.bin"), h_grid_ice="C", **kw)
    np.savez(tmp_path / "ccase.npz", **d, ew=np.array(ew), ns=np.array(ns), visc_method=np.array(kw.get("h_visc_method", "avg_zeta")))
    old = common.GOLDEN
    common.GOLDEN = tmp_path
    try:
        return GoldenCase("ccase")
    finally:
        common.GOLDEN = old


@pytest.mark.parametrize("bs", [(320, 384), (160, 192)])
def test_cgrid_gx1_size_vs_reference_harness(tmp_path, bs, monkeypatch):
    """The C-grid loop at BASELINE's gx1 size (320 x 384, ndte = 120) against the reference ITSELF: operands captured from,
    outputs compared with, the reference's own evp() with grid_ice = 'C' (unmodified sources, strict build) run here on the
    box, as one block and as 2 x 2 blocks.  Every schedule that can run this grid: the default (the on-chip resident kernel
    cg_res where it is eligible: all subcycles of a call but the first in one launch), forced one launch per subcycle (cg_one; and
    with the interior of each block marched: cg_strip), forced three launches -- bit-identical on all 19 arrays, ghost cells included, after 120 subcycles and after 7."""
    c = reference_cgrid_case(tmp_path, 320, 384, bs, "cyclic", "closed", icecase="full", nsub_list=[7, 120], ncalls=1, h_ndte=120)
    dom = c.oracle_domain()
    state, inputs, masks = c.cgrid_inputs(1)
    assert (masks["iceTmask"] != 0).sum() > 80000
    ran_resident = False
    for what, envs in (("default", {}), ("one launch per subcycle", {"CICE_EVP_HIP_CGRID_RESIDENT": "0"}),
                       ("one launch per subcycle, the interior marched", {"CICE_EVP_HIP_CGRID_RESIDENT": "0", "CICE_EVP_HIP_CGRID_STRIP": "1"}),
                       ("three launches", {"CICE_EVP_HIP_CGRID_RESIDENT": "0", "CICE_EVP_HIP_CGRID_ONE": "0"})):
        for k in ("CICE_EVP_HIP_CGRID_RESIDENT", "CICE_EVP_HIP_CGRID_ONE", "CICE_EVP_HIP_CGRID_STRIP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in envs.items():
            monkeypatch.setenv(k, v)
        core = cgrid_core(c)
        try:
            for nsub in (120, 7):
                out = core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
                oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
                oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
                assert_bitwise(out, c.cgrid_expected(1, nsub), f"gx1-size C grid {bs} {what} nsub {nsub} vs the reference")
                t = core.cgrid_timings()
                if what == "default":
                    ran_resident = ran_resident or t["resident_subcycles"] == nsub - 1
                else:
                    assert t["resident_subcycles"] == 0
                assert (t["marched_items"] > 0) == ("marched" in what), t
        finally:
            core.finalize()
    assert np.abs(out["uvelE"]).max() > 1e-3
    if bs == (320, 384):
        assert ran_resident, "one block of 320 x 384 on one rank is what the resident kernel is for"


@pytest.mark.parametrize("seed", list(range(501, 509)) + list(range(901, 909)) + [int(s) for s in __import__("os").environ.get("CGRID_REF_SWEEP_SEEDS", "").split() if s])
def test_cgrid_geometry_sweep_vs_reference(seed, tmp_path, monkeypatch):
    monkeypatch.setenv("CICE_EVP_HIP_CGRID_ONE_SHAPE", str(seed % 3))      # all three windows of the one-launch kernel (32x8, 64x8, 64x16)
    rng = np.random.default_rng(seed)
    ns = ["closed", "tripole", "cyclic"][seed % 3] if seed % 7 else "tripole"
    if seed >= 900:
        ns = "tripoleT"                       # (end of round 4) T-fold lists of all four locations, 1 - 3 blocks across
    ew = "closed" if (ns == "closed" and seed % 2) else "cyclic"
    nx, ny = 2 * int(rng.integers(10, 36)), int(rng.integers(14, 48))
    nbx, nby = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    bs = (-(-nx // nbx), -(-ny // nby))
    kw = dict(icecase=str(rng.choice(["full", "patchy", "caps"])), nsub_list=[1, 7], ncalls=2, h_ndte=7, h_evolve=True)
    if rng.random() < 0.3:
        kw["h_visc_method"] = "avg_strength"
    if rng.random() < 0.3:
        kw["h_seabed"] = True
    if rng.random() < 0.3:
        kw.update(h_capping=float(rng.choice([0.0, 0.5])), h_Ktens=0.1)
    if rng.random() < 0.3:
        kw["h_ssh"] = "coupled"
    what = f"seed {seed}: {nx}x{ny} ew {ew} ns {ns}, blocks {bs[0]}x{bs[1]}, {kw}"
    c = reference_cgrid_case(tmp_path, nx, ny, bs, ew, ns, **kw)
    dom = c.oracle_domain()
    core = cgrid_core(c)
    try:
        core.cgrid_set_prep_geometry(c.cgrid_prep_static())
        for icall in range(1, c.ncalls + 1):
            state, inputs, masks = c.cgrid_inputs(icall)
            for nsub in c.nsub_list:
                out = core.cgrid_run(nsub, state, inputs, masks, visc_method=str(c.d["visc_method"]))
                oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
                oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
                assert_bitwise(out, c.cgrid_expected(icall, nsub), f"{what}: call {icall} nsub {nsub} (loop)")
                if __import__("os").environ.get("CGRID_REF_SWEEP_LOG"):      # (which kernel ran: one line per call, for sweeps by hand)
                    with open(__import__("os").environ["CGRID_REF_SWEEP_LOG"], "a") as fh:
                        fh.write(f"{seed} {ns} {nx}x{ny} blocks {nbx}x{nby} nsub {nsub} resident {core.cgrid_timings()['resident_subcycles']}\n")
            if c.scal[23] != 0.0:
                continue                      # (seabed stress: the device exp() may differ from the host's in the last bit of TbE / TbN)
            # the same call from the T-grid state: preparation on the device, then the loop from the device-prepared state
            _, st, _ = c.cgrid_prep_inputs(icall)
            masks_d = device_prep(core, c, icall, {k: st[k] for k in oracle.C_FIELDS[:12]})
            for k in oracle.C_MASKS:
                assert bits_equal(masks_d[k] != 0, masks[k] != 0), (what, icall, k)
            core.cgrid_subcycle(c.nsub_list[-1])
            out = core.cgrid_download()
            oracle.halo_update(dom, out["strintxE"], "Eface", "vector")
            oracle.halo_update(dom, out["strintyN"], "Nface", "vector")
            assert_bitwise(out, c.cgrid_expected(icall, c.nsub_list[-1]), f"{what}: call {icall} (device preparation + loop)")
        assert np.abs(out["uvelE"]).max() > 1e-4, what
        assert core.cgrid_timings()["resident_fallbacks"] == 0, (what, core.cgrid_timings())     # nothing repeated in silence
    finally:
        core.finalize()
