"""GPU (-m gpu): the drop-in itself.  oracle/_ref/evp_hip_dropin_harness is the
reference's own, unmodified evp() driver (compiled in place from the reference tree)
linked with cice_amd/fortran/ice_dyn_evp1d_hip.F90 -- the build-owned module that
takes the place of the reference's alternative EVP core at ice_dyn_evp.F90:846-856 --
and libcice_evp_hip.so.  In one process it runs, from identical state, the HIP core
(through Fortran -> ISO_C_BINDING -> C ABI -> HIP) and the reference's standard_2d
path, and dumps both; they must be bit-identical, ghost cells included.

The binary is prebuilt by oracle/ref/build_ref.sh (needs the reference tree) and
travels to the GPU box; the test skips when it is absent."""
import numpy as np
import pytest

import run_ref
from cice_amd import synth
from common import bits_equal

pytestmark = pytest.mark.gpu

FIELDS = (["uvel", "vvel", "strintxU", "strintyU", "taubxU", "taubyU"] +
          [f"stress{k}_{c}" for k in ("p", "m", "12") for c in range(1, 5)])
# outputs of the untouched remainder of evp() (deformations, dyn_finish) fed by the HIP velocities
DOWNSTREAM = ["divu", "shear", "strocnxU"]

CASES = [
    # nx, ny, bx, by, ew, kwargs
    (40, 36, 20, 18, "cyclic", dict(grid_kind="rect", icecase="full")),
    (60, 44, 20, 15, "cyclic", dict(grid_kind="popfile", icecase="patchy")),
    (100, 116, 50, 29, "cyclic", dict(grid_kind="popfile", icecase="caps", h_seabed=True)),
    (64, 48, 64, 48, "closed", dict(grid_kind="popfile", icecase="full", h_revised=True)),
    # tripole north boundary: fold row inside the loop and -- in the Option-A body -- the 12 x
    # ice_HaloUpdate_stress symmetrisation on the device (cice_evp_hip_stress_halo)
    (72, 40, 36, 20, "cyclic", dict(grid_kind="tripolefile", icecase="full", ns="tripole")),
    (48, 36, 48, 36, "cyclic", dict(grid_kind="tripolefile", icecase="patchy", ns="tripole", h_capping=0.5)),
]


@pytest.mark.parametrize("resident", [False, True], ids=["no_hooks", "resident_opt_in"])
@pytest.mark.parametrize("nx,ny,bx,by,ew,kw", CASES)
def test_reference_driver_with_hip_core_bitwise(tmp_path, nx, ny, bx, by, ew, kw, resident):
    """resident=False (the shim's default): the harness calls NO fetch / invalidate hook -- an unpatched host.  The
    twelve intent(inout) stress arrays of dyn_evp1d_run (ice_dyn_evp1d.F90:121-135) must hold the reference's values
    in ice_flux after every evp() call (what ice_restart_driver.F90:187-200 writes), closed and tripole grids alike
    (on a tripole grid evp()'s own 12 x ice_HaloUpdate_stress then runs on current host arrays).
    resident=True: the host opted in (dyn_evp_hip_keep_stresses_resident) and calls the two hooks."""
    if resident and (nx, ny) not in ((40, 36), (72, 40), (48, 36)):
        pytest.skip("opt-in variant: one closed-north and the two tripole cases")
    run_bgrid_dropin(tmp_path, nx, ny, bx, by, ew, kw, resident)


def run_bgrid_dropin(tmp_path, nx, ny, bx, by, ew, kw, resident, ndte=120):
    if not run_ref.have_ref("hip_dropin"):
        pytest.skip("oracle/_ref/evp_hip_dropin_harness not built (needs the reference tree)")
    kw = dict(kw)
    ns = kw.pop("ns", "closed")
    grid_files = None
    if kw["grid_kind"] != "rect":
        g = synth.make_grid(nx, ny, dx0=1.1e5, ns=ns)
        run_ref.write_pop_grid(tmp_path / "grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
        run_ref.write_kmt(tmp_path / "kmt.bin", g["kmt"])
        grid_files = (tmp_path / "grid.bin", tmp_path / "kmt.bin")
    d, txt = run_ref.run_harness(nx, ny, bx, by, ew=ew, ns=ns, variant="hip_dropin", h_ndte=ndte,
                                 ncalls=2, nsub_list=[1, ndte], hipmode=True, hipbody=True,
                                 hipresident=resident, grid_files=grid_files, **kw)
    checked = 0
    for icall in (1, 2):
        for nsub in (1, ndte):
            for f in FIELDS + DOWNSTREAM:
                hip = d[f"h{icall:02d}n{nsub:04d}_{f}"]
                ref = d[f"o{icall:02d}n{nsub:04d}_{f}"]
                assert bits_equal(hip, ref), (
                    f"call {icall} nsub {nsub} {f}: {int((hip != ref).sum())} cells differ, "
                    f"max|d|={np.abs(hip - ref).max():.3e}")
                checked += 1
            # Option A: the B-grid body of evp() -- preparation phase + loop --
            # through Fortran dyn_evp_hip_evp_body -> cice_evp_hip_prep / _set_strength / _subcycle
            # / _download, the ice strength computed by a host callback in between
            for f in FIELDS:
                body = d[f"b{icall:02d}n{nsub:04d}_{f}"]
                ref = d[f"o{icall:02d}n{nsub:04d}_{f}"]
                assert bits_equal(body, ref), (
                    f"Option A body, call {icall} nsub {nsub} {f}: {int((body != ref).sum())} cells differ, "
                    f"max|d|={np.abs(body - ref).max():.3e}")
                checked += 1
            if resident:
                # Option A with resident stresses, two bodies in a row: the second call neither uploads nor downloads the 12
                # stresses (the host copies were overwritten with garbage in between) -- against evp() called twice
                for f in ("uvel", "vvel", "stressp_1", "stressm_2", "stress12_3", "stress12_4"):
                    two = d[f"c{icall:02d}n{nsub:04d}_{f}"]
                    ref = d[f"p{icall:02d}n{nsub:04d}_{f}"]
                    assert bits_equal(two, ref), (
                        f"Option A, resident stresses, second body in a row, call {icall} nsub {nsub} {f}: "
                        f"{int((two != ref).sum())} cells differ, max|d|={np.abs(two - ref).max():.3e}")
    assert np.abs(d[f"o02n{ndte:04d}_uvel"]).max() > (1e-3 if ndte >= 100 else 1e-5)
    assert checked == 2 * 2 * (2 * len(FIELDS) + len(DOWNSTREAM))


@pytest.mark.parametrize("seed", list(range(601, 607)) + [int(s) for s in __import__("os").environ.get("DROPIN_SWEEP_SEEDS", "").split() if s])
def test_reference_driver_with_hip_core_geometry_sweep(tmp_path, seed):
    """The same through random geometries: domain size, block split (padded last blocks, several blocks next to the
    fold), closed / cyclic east-west, closed / tripole north, ice case, options -- the Fortran shim builds the dims, the
    block table and the halo plan for whatever decomposition the reference's driver hands it."""
    rng = np.random.default_rng(seed)
    trip = seed % 3 == 0
    nx, ny = 2 * int(rng.integers(12, 50)), int(rng.integers(16, 60))
    nbx, nby = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    bx, by = -(-nx // nbx), -(-ny // nby)
    kw = dict(grid_kind="tripolefile" if trip else "popfile", icecase=str(rng.choice(["full", "patchy", "caps"])))
    if trip:
        kw["ns"] = "tripole"
    if rng.random() < 0.3:
        kw["h_seabed"] = True
    if rng.random() < 0.3:
        kw["h_revised"] = True
    if rng.random() < 0.3:
        kw["h_capping"] = 0.5
    ew = "closed" if (not trip and seed % 2) else "cyclic"
    run_bgrid_dropin(tmp_path, nx, ny, bx, by, ew, kw, resident=bool(seed % 5 == 0), ndte=int(rng.choice([5, 12])))


CGRID_LOOP_FIELDS = ["uvelE", "vvelE", "uvelN", "vvelN", "uvel", "vvel", "stresspT", "stressmT", "stress12T", "stress12U",
                     "taubxE", "taubyN", "zetax2T", "etax2T", "etax2U", "shearU", "deltaU"]
# deformationsC_T on the device (dyn_evp_hip_cgrid_deformations), from the state the HIP loop left there
CGRID_DOWNSTREAM = ["divu", "shear", "vort", "rdg_conv", "rdg_shear",
                    # dyn_finish at N and E points on the device (dyn_evp_hip_cgrid_dyn_finish), same state
                    "strocnxN", "strocnyN", "strocnxE", "strocnyE"]
CGRID_CASES = [
    (40, 36, 20, 18, "cyclic", dict(icecase="full")),
    (60, 44, 20, 15, "cyclic", dict(icecase="patchy", h_visc_method="avg_strength", h_capping=0.5)),
    (64, 48, 64, 48, "closed", dict(icecase="full", h_revised=True, h_seabed=True)),
    (72, 40, 36, 20, "cyclic", dict(icecase="full", ns="tripole")),          # fold step after every phase
    (72, 40, 24, 20, "cyclic", dict(icecase="patchy", ns="tripoleT")),       # T-fold lists (end of round 4), three blocks across
]


@pytest.mark.parametrize("prep", ["reference_preparation", "device_preparation"])
@pytest.mark.parametrize("nx,ny,bx,by,ew,kw", CGRID_CASES)
def test_reference_driver_with_hip_cgrid_loop_bitwise(tmp_path, nx, ny, bx, by, ew, kw, prep):
    """C grid: the reference's own driver and preparation (evp() with ndte = 0), then the subcycle loop through the
    Fortran entry a patched evp() calls -- dyn_evp_hip_cgrid_run(<ice_dyn_evp's private arrays>) -> ISO_C_BINDING
    -> cice_evp_hip_cgrid_run -> HIP -- against the reference's evp() with grid_ice = 'C' from the same state, in
    the same process.  Every array the loop writes, every cell, ghost cells included; strintxE / strintyN after the
    halo update evp() gives them once the loop is over (ice_dyn_evp.F90:1437-1440; applied here with the oracle).
    device_preparation: the reference's evp() does not run at all for the HIP result -- dyn_evp_hip_cgrid_evp_body does
    the preparation (device; ice strength and seabed factors by the host's routines from the returned masks) and the
    loop from the state evp() would be entered with; the ice cover changes between the two calls and the sea surface
    slopes (ssh_stress = 'coupled')."""
    run_cgrid_dropin(tmp_path, nx, ny, bx, by, ew, kw, prep)


def run_cgrid_dropin(tmp_path, nx, ny, bx, by, ew, kw, prep, ndte=120):
    if not run_ref.have_ref("hip_dropin"):
        pytest.skip("oracle/_ref/evp_hip_dropin_harness not built (needs the reference tree)")
    kw = dict(kw)
    if prep == "device_preparation":
        kw.update(hipbody=True, h_evolve=True, h_ssh="coupled")
    ns = kw.pop("ns", "closed")
    g = synth.make_grid(nx, ny, dx0=1.1e5, ns=("tripole" if ns == "tripoleT" else ns))
    run_ref.write_pop_grid(tmp_path / "grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
    run_ref.write_kmt(tmp_path / "kmt.bin", g["kmt"])
    d, txt = run_ref.run_harness(nx, ny, bx, by, ew=ew, ns=ns, variant="hip_dropin", h_ndte=ndte, ncalls=2,
                                 nsub_list=[1, ndte], hipmode=True, h_grid_ice="C",
                                 grid_kind=("tripolefile" if ns in ("tripole", "tripoleT") else "popfile"),
                                 grid_files=(tmp_path / "grid.bin", tmp_path / "kmt.bin"), **kw)
    import oracle
    dom = oracle.OracleDomain.from_dump(d, ew, ns)
    checked = 0
    for icall in (1, 2):
        for nsub in (1, ndte):
            for f in CGRID_LOOP_FIELDS + ["strintxE", "strintyN"] + CGRID_DOWNSTREAM:
                hip = d[f"h{icall:02d}n{nsub:04d}_{f}"]
                ref = d[f"o{icall:02d}n{nsub:04d}_{f}"]
                if f.startswith("strint"):     # evp()'s own halo update after the loop (on a tripole grid it also
                    hip = oracle.halo_update(dom, np.ascontiguousarray(hip.copy()),   # averages the N-face row ON the fold)
                                             "Eface" if f == "strintxE" else "Nface", "vector")
                if f.startswith("strocn") and prep == "device_preparation":
                    # (a host that skips its own preparation zeroes the ocean stresses off the ice itself, DESIGN section 9:
                    # compared on the faces dyn_finish writes)
                    on = d[f"in{icall:02d}_ice{f[-1]}mask"] != 0
                    hip, ref = np.where(on, hip, 0.0), np.where(on, ref, 0.0)
                assert bits_equal(hip, ref), (
                    f"C grid call {icall} nsub {nsub} {f}: {int((hip != ref).sum())} cells differ, "
                    f"max|d|={np.abs(hip - ref).max():.3e}")
                checked += 1
    assert np.abs(d[f"o02n{ndte:04d}_uvelE"]).max() > (1e-3 if ndte >= 100 else 1e-5) and checked == 2 * 2 * (len(CGRID_LOOP_FIELDS) + 2 + len(CGRID_DOWNSTREAM))
    assert np.abs(d[f"o02n{ndte:04d}_divu"]).max() > 0


@pytest.mark.parametrize("seed", list(range(701, 707)) + [int(s) for s in __import__("os").environ.get("DROPIN_SWEEP_SEEDS", "").split() if s])
def test_reference_driver_with_hip_cgrid_loop_geometry_sweep(tmp_path, seed):
    """C grid through random geometries and options (both visc_methods, seabed stress, revised EVP, closed / cyclic /
    tripole boundaries, padded blocks), the reference's preparation or the device's."""
    rng = np.random.default_rng(seed)
    trip = seed % 3 == 0
    nx, ny = 2 * int(rng.integers(12, 40)), int(rng.integers(16, 50))
    nbx, nby = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    bx, by = -(-nx // nbx), -(-ny // nby)
    kw = dict(icecase=str(rng.choice(["full", "patchy", "caps"])))
    if trip:
        kw["ns"] = "tripole"
    if rng.random() < 0.3:
        kw["h_visc_method"] = "avg_strength"
    if rng.random() < 0.3:
        kw["h_seabed"] = True
    if rng.random() < 0.3:
        kw["h_revised"] = True
    ew = "closed" if (not trip and seed % 2) else "cyclic"
    run_cgrid_dropin(tmp_path, nx, ny, bx, by, ew, kw, "device_preparation" if seed % 2 else "reference_preparation",
                     ndte=int(rng.choice([5, 12])))


@pytest.mark.parametrize("resident", [False, True], ids=["no_hooks", "resident_opt_in"])
@pytest.mark.parametrize("nx,ny,bx,by,kw", [(72, 40, 36, 20, dict(icecase="full")), (48, 36, 48, 36, dict(icecase="patchy", h_capping=0.5)),
                                            (90, 30, 18, 15, dict(icecase="full"))])      # five blocks across the top row
def test_reference_driver_with_hip_core_on_a_tripoleT_grid(tmp_path, nx, ny, bx, by, kw, resident, monkeypatch):
    """ns_boundary_type = 'tripoleT' (T-fold; ice_domain.F90:260), Option B: the reference's unmodified evp() -- its own
    preparation, its own 12 x ice_HaloUpdate_stress after the loop -- with the HIP core behind dyn_evp1d_run.  The loop's
    velocity halo follows the T-fold rule (top U row = image of row NY-1).  Every output array of the whole evp(), every
    cell, against the reference's standard path in the same process.  resident_opt_in (late round 4): the host keeps the
    stresses on the device between calls; evp()'s twelve ice_HaloUpdate_stress calls then act on stale host arrays and the
    library applies the same step to the device copy (cice_evp_hip_stress_halo, T-fold rule incl. the north-west corner
    ghost cells) -- what the fetch hook brings back must equal the reference's arrays on every cell."""
    if not run_ref.have_ref("hip_dropin"):
        pytest.skip("oracle/_ref/evp_hip_dropin_harness not built (needs the reference tree)")
    monkeypatch.setenv("CICE_EVP_HIP_VERBOSE", "1")        # the shim reports once which kernel / transport the library settled on
    g = synth.make_grid(nx, ny, dx0=1.1e5, ns="tripole")
    run_ref.write_pop_grid(tmp_path / "grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
    run_ref.write_kmt(tmp_path / "kmt.bin", g["kmt"])
    d, txt = run_ref.run_harness(nx, ny, bx, by, ew="cyclic", ns="tripoleT", variant="hip_dropin", h_ndte=120,
                                 ncalls=2, nsub_list=[1, 120], hipmode=True, hipbody=True, grid_kind="tripolefile",
                                 hipresident=resident, grid_files=(tmp_path / "grid.bin", tmp_path / "kmt.bin"), **kw)
    checked = 0
    for icall in (1, 2):
        for nsub in (1, 120):
            for f in FIELDS + DOWNSTREAM:
                hip = d[f"h{icall:02d}n{nsub:04d}_{f}"]
                ref = d[f"o{icall:02d}n{nsub:04d}_{f}"]
                assert bits_equal(hip, ref), (
                    f"tripoleT call {icall} nsub {nsub} {f}: {int((hip != ref).sum())} cells differ, "
                    f"max|d|={np.abs(hip - ref).max():.3e}")
                checked += 1
            # Option A (end of round 4): the whole B-grid body of evp() through dyn_evp_hip_evp_body -- the preparation with
            # the T-fold rule of the cell-centre fields, the loop, the symmetrisation, all on the device
            for f in FIELDS:
                body = d[f"b{icall:02d}n{nsub:04d}_{f}"]
                ref = d[f"o{icall:02d}n{nsub:04d}_{f}"]
                assert bits_equal(body, ref), (
                    f"tripoleT, Option A body, call {icall} nsub {nsub} {f}: {int((body != ref).sum())} cells differ, "
                    f"max|d|={np.abs(body - ref).max():.3e}")
                checked += 1
    assert np.abs(d["o02n0120_uvel"]).max() > 1e-3 and checked == 2 * 2 * (2 * len(FIELDS) + len(DOWNSTREAM))
    # (the library's own choice: since round 6 the on-chip resident kernel with the T-fold inside; the streaming kernel with its list
    # copies where the probe prefers it)
    assert ("(dyn_evp_hip) task 0: rank 0 of 1: kernel = on-chip resident" in txt or
            "(dyn_evp_hip) task 0: rank 0 of 1: kernel = one subcycle per launch (streaming)" in txt), txt[-1500:]
