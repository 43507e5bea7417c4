"""GPU (-m gpu): the drop-in under the reference's MPI path -- the Fortran shim's `nprocs > 1` branch executed.

oracle/_ref/evp_hip_dropin_harness_mpi = the reference's unmodified evp() driver with its comm/mpi modules (compiled in
place against the image's MPICH) + cice_amd/fortran shim + libcice_evp_hip.so, started as 2-4 MPI tasks by mpiexec.
The reference's own init_domain_distribution deals the blocks out (cartesian and roundrobin); dyn_evp_hip_init builds
the owner table from distrb_info%blockLocation / blockLocalID (ice_distribution.F90:24-37), ships the bootstrap data
with CICE's own broadcast_array (comm/mpi/ice_broadcast.F90) and calls cice_evp_hip_halo_import; from then on the
velocity halo of every subcycle travels between the tasks' kernels, not through MPI.  The box has ONE GPU, so the
tasks share it (CICE_EVP_HIP_BOOTSTRAP=blobs: HIP-IPC mailboxes; RCCL refuses two ranks on one device) -- the stand-in
a 1-GPU box allows, as in test_gpu_zz_multiprocess.py.  Every task's arrays after the HIP core must equal what the
reference's standard_2d path (halo through MPI) leaves in the same task, ghost cells included, and the assembled
global fields must equal the reference's serial build.  (Named to sort after test_gpu_parity.py, like
test_gpu_zz_multiprocess.py: several processes time-slicing one GPU must not be able to hide the single-process results
under `pytest -x`.)"""
import os

import numpy as np
import pytest

import run_ref
from cice_amd import synth
from common import bits_equal

pytestmark = pytest.mark.gpu

B_FIELDS = (["uvel", "vvel", "strintxU", "strintyU", "taubxU", "taubyU"] +
            [f"stress{k}_{c}" for k in ("p", "m", "12") for c in range(1, 5)])
B_DOWNSTREAM = ["divu", "shear", "strocnxU"]

CASES = [
    # nx, ny, bx, by, ew, ns, nprocs, distribution, body (Option A too), kwargs
    (40, 36, 20, 18, "cyclic", "closed", 2, "cartesian", False, dict(grid_kind="rect", icecase="full")),
    (60, 44, 20, 15, "cyclic", "closed", 2, "roundrobin", False, dict(grid_kind="popfile", icecase="patchy")),
    (100, 116, 50, 29, "closed", "closed", 4, "cartesian", False, dict(grid_kind="popfile", icecase="caps", h_seabed=True)),
    (72, 40, 36, 20, "cyclic", "tripole", 2, "cartesian", False, dict(grid_kind="tripolefile", icecase="full")),
    (72, 40, 18, 20, "cyclic", "tripole", 4, "roundrobin", False, dict(grid_kind="tripolefile", icecase="patchy", h_capping=0.5)),
    (72, 40, 36, 20, "cyclic", "closed", 2, "cartesian", True, dict(grid_kind="popfile", icecase="full")),
    # ns_boundary_type = 'tripoleT' over MPI tasks: the fold row cut in y only, in x, five blocks across dealt round-robin
    (72, 40, 36, 20, "cyclic", "tripoleT", 2, "cartesian", False, dict(grid_kind="tripolefile", icecase="full")),
    (72, 40, 18, 40, "cyclic", "tripoleT", 2, "cartesian", False, dict(grid_kind="tripolefile", icecase="patchy")),
    (90, 30, 18, 15, "cyclic", "tripoleT", 3, "roundrobin", False, dict(grid_kind="tripolefile", icecase="full", h_capping=0.5)),
]


CGRID_FIELDS = ["uvelE", "vvelE", "uvelN", "vvelN", "uvel", "vvel", "stresspT", "stressmT", "stress12T", "stress12U",
                "taubxE", "taubyN", "zetax2T", "etax2T", "etax2U", "shearU", "deltaU"]
CGRID_DOWNSTREAM = ["divu", "shear", "vort", "rdg_conv", "rdg_shear", "strocnxN", "strocnyN", "strocnxE", "strocnyE"]


def run_case(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, body, kw, ndte=24, cgrid=False, maskhalo=False):
    FIELDS, DOWNSTREAM = (CGRID_FIELDS, CGRID_DOWNSTREAM) if cgrid else (B_FIELDS, B_DOWNSTREAM)
    if cgrid:
        kw = dict(kw, h_grid_ice="C")
    if not (run_ref.have_ref("hip_dropin_mpi") and run_ref.have_mpiexec()):
        pytest.skip("oracle/_ref/evp_hip_dropin_harness_mpi or mpiexec not available")
    files = None
    if kw["grid_kind"] != "rect":
        g = synth.make_grid(nx, ny, dx0=1.1e5, ns=("tripole" if ns == "tripoleT" else ns))
        run_ref.write_pop_grid(tmp_path / "grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
        run_ref.write_kmt(tmp_path / "kmt.bin", g["kmt"])
        files = (tmp_path / "grid.bin", tmp_path / "kmt.bin")
    common = dict(ew=ew, ns=ns, h_ndte=ndte, ncalls=2, nsub_list=[1, ndte], grid_files=files, maskhalo_dyn=maskhalo, **kw)
    env = {"CICE_EVP_HIP_BOOTSTRAP": "blobs", "CICE_EVP_HIP_HALO_TIMEOUT_MS": "20000",
           "CICE_EVP_HIP_DEVICE": "0", "CICE_EVP_HIP_VERBOSE": "1"}
    last = None
    for attempt in (1, 2):          # tasks time-slicing one GPU: one retry, as in test_gpu_zz_multiprocess.py
        try:
            par, txt = run_ref.run_harness(nx, ny, bx, by, variant="hip_dropin_mpi", nprocs=nprocs,
                                           distribution_type=dist, hipmode=True, hipbody=body, extra_env=env,
                                           workdir=tmp_path / f"par{attempt}", timeout=600, **common)
            last = None
            break
        except RuntimeError as e:
            last = e
    if last is not None:
        try:
            os.makedirs("gpurun_out", exist_ok=True)
            with open(f"gpurun_out/dropin_mpi_FAIL_{nx}x{ny}_{bx}x{by}_{nprocs}_{dist}_{ns}.log", "w") as f:
                f.write(str(last)[-20000:])
        except OSError:
            pass
        raise last
    checked = 0
    for r, d in enumerate(par):
        for icall in (1, 2):
            for nsub in (1, ndte):
                for f in FIELDS + DOWNSTREAM:
                    hip, ref = d[f"h{icall:02d}n{nsub:04d}_{f}"], d[f"o{icall:02d}n{nsub:04d}_{f}"]
                    assert bits_equal(hip, ref), (f"task {r} call {icall} nsub {nsub} {f}: "
                                                  f"{int((hip != ref).sum())} cells differ, max|d|={np.abs(hip - ref).max():.3e}")
                    checked += 1
                if body:
                    for f in FIELDS:
                        b, ref = d[f"b{icall:02d}n{nsub:04d}_{f}"], d[f"o{icall:02d}n{nsub:04d}_{f}"]
                        assert bits_equal(b, ref), f"Option A body, task {r} call {icall} nsub {nsub} {f}"
                        checked += 1
    assert checked == nprocs * 4 * (len(FIELDS) * (2 if body else 1) + len(DOWNSTREAM))
    # ... and the whole thing against the reference's serial build on the same domain
    if run_ref.have_ref("strict"):
        ser, _ = run_ref.run_harness(nx, ny, bx, by, variant="strict", workdir=tmp_path / "ser", **common)
        for f in FIELDS:
            k = f"n{ndte:04d}_{f}"
            assert bits_equal(run_ref.global_field(par, "h02" + k), run_ref.global_field(ser, "o02" + k)), f
    if cgrid:
        assert np.nanmax(np.abs(run_ref.global_field(par, f"h02n{ndte:04d}_uvelE"))) > 1e-5
    assert np.nanmax(np.abs(run_ref.global_field(par, f"h02n{ndte:04d}_uvel"))) > 1e-5
    return txt


@pytest.mark.parametrize("seed", list(range(901, 909)) + [int(s) for s in os.environ.get("DROPIN_MPI_SWEEP_SEEDS", "").split() if s])
def test_reference_mpi_driver_with_hip_core_geometry_sweep(tmp_path, seed):
    """Random domain sizes, block splits (padded last blocks), 2-4 MPI tasks, cartesian / roundrobin / sectrobin /
    rake-free distributions of the reference's own making, closed / cyclic / tripole boundaries, options."""
    rng = np.random.default_rng(seed)
    trip = seed % 3 == 0
    nx, ny = 2 * int(rng.integers(12, 50)), int(rng.integers(16, 60))
    nbx, nby = int(rng.integers(2, 5)), int(rng.integers(1, 4))
    bx, by = -(-nx // nbx), -(-ny // nby)
    nblk = (-(-nx // bx)) * (-(-ny // by))
    nprocs = int(rng.integers(2, min(4, nblk) + 1))
    dist = str(rng.choice(["cartesian", "roundrobin", "sectrobin"]))
    kw = dict(grid_kind="tripolefile" if trip else "popfile", icecase=str(rng.choice(["full", "patchy", "caps"])))
    if rng.random() < 0.3:
        kw["h_seabed"] = True
    if rng.random() < 0.3:
        kw["h_revised"] = True
    if rng.random() < 0.3:
        kw["h_capping"] = 0.5
    ew = "closed" if (not trip and seed % 2) else "cyclic"
    run_case(tmp_path, nx, ny, bx, by, ew, "tripole" if trip else "closed", nprocs, dist, False, kw, ndte=int(rng.choice([5, 12])))


@pytest.mark.parametrize("seed", list(range(3001, 3011)) + [int(s) for s in os.environ.get("DROPIN_MPI_WIDE_SEEDS", "").split() if s])
def test_reference_mpi_driver_with_hip_core_wide_sweep(tmp_path, seed):
    """The sweep above widened: C grid as well (closed north: across ranks the C grid wants the fold's blocks on one rank),
    the masked halo, Option A (device preparation), tasks without blocks, spacecurve distributions."""
    rng = np.random.default_rng(seed)
    cgrid = bool(rng.random() < 0.4)
    trip = (not cgrid) and seed % 3 == 0
    nx, ny = 2 * int(rng.integers(12, 50)), int(rng.integers(16, 60))
    nbx, nby = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    bx, by = -(-nx // nbx), -(-ny // nby)
    nblk = (-(-nx // bx)) * (-(-ny // by))
    if nblk < 2:
        bx = -(-nx // 2)
        nblk = 2 * (-(-ny // by))
    nprocs = int(rng.integers(2, 5))
    if nprocs > nblk:
        nprocs = nblk
    dist = str(rng.choice(["cartesian", "roundrobin", "sectrobin", "sectcart", "rake"]))
    if dist == "sectcart" and (nprocs % 2 or (-(-nx // bx)) % 2):      # (create_distrb_sectcart wants an even nblocks_x)
        dist = "cartesian"
    kw = dict(grid_kind="tripolefile" if trip else "popfile", icecase=str(rng.choice(["full", "patchy", "caps"])))
    if rng.random() < 0.3:
        kw["h_seabed"] = True
    if rng.random() < 0.3:
        kw["h_revised"] = True
    if rng.random() < 0.3:
        kw["h_capping"] = 0.5
    if cgrid and rng.random() < 0.3:
        kw["h_visc_method"] = "avg_strength"
    body = (not cgrid) and bool(rng.random() < 0.3)
    maskhalo = bool(rng.random() < 0.4)
    ew = "closed" if (not trip and seed % 2) else "cyclic"
    run_case(tmp_path, nx, ny, bx, by, ew, "tripole" if trip else "closed", nprocs, dist, body, kw,
             ndte=int(rng.choice([5, 12])), cgrid=cgrid, maskhalo=maskhalo)


@pytest.mark.parametrize("cgrid", [False, True], ids=["B", "C"])
@pytest.mark.parametrize("nx,ny,bx,by,ew,ns,nprocs,dist,kw", [
    (54, 52, 14, 26, "cyclic", "tripole", 3, "cartesian", dict(grid_kind="tripolefile", icecase="full")),    # 4 x 2 blocks: 4, 4, 0
    (90, 21, 30, 21, "closed", "closed", 2, "cartesian", dict(grid_kind="popfile", icecase="patchy")),        # 3 x 1 blocks: 3, 0
    (68, 54, 17, 27, "cyclic", "closed", 3, "cartesian", dict(grid_kind="popfile", icecase="caps", h_revised=True)),
    # tripoleT with a blockless task (round-5 advice): settle_stress_residency's global_minval is a collective over ALL tasks
    # of the distribution -- the task without blocks used to skip it and the others waited for ever
    (54, 52, 14, 26, "cyclic", "tripoleT", 3, "cartesian", dict(grid_kind="tripolefile", icecase="full")),
])
def test_reference_mpi_driver_with_a_task_that_holds_no_blocks(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, kw, cgrid):
    """The reference's cartesian distribution hands a task NOTHING when its processor grid does not divide the block grid
    (ice_distribution.F90 create_distrb_cart) and carries on; so does the drop-in: the task without blocks is a bystander
    of the bootstrap (cice_evp_hip_init with nblocks = 0, an empty blob in the all-gather) and the shim's routines
    return at once there.  Found by the geometry sweep below (seeds 2043, 2057, 2084, 2119, 2129, 2134, 2136)."""
    if cgrid and ns in ("tripole", "tripoleT"):
        pytest.skip("C grid on a tripole grid: the blocks next to the fold must lie on one rank (INTEGRATION.md); this cut splits them in x")
    run_case(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, False, kw, cgrid=cgrid)


@pytest.mark.parametrize("cgrid", [False, True], ids=["B", "C"])
@pytest.mark.parametrize("nx,ny,bx,by,ew,ns,nprocs,dist,kw", [
    (60, 48, 20, 12, "cyclic", "closed", 3, "roundrobin", dict(grid_kind="popfile", icecase="caps")),
    (72, 40, 18, 20, "cyclic", "tripole", 2, "cartesian", dict(grid_kind="tripolefile", icecase="patchy")),
])
def test_reference_mpi_driver_with_hip_core_masked_halo(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, kw, cgrid):
    """maskhalo_dyn = .true. under MPI: evp() builds halo_info_mask (ice_dyn_evp.F90:739-770) and the reference's loop
    exchanges only strips that hold ice; the shim rebuilds the same mask from iceUmask (B grid) / the dilated iceTmask (C
    grid) with the reference's own ice_HaloUpdate across the MPI tasks and hands it to cice_evp_hip_halo_mask -- the
    branch `maskhalo_dyn .and. get_num_procs() > 1` of dyn_evp_hip_run / cgrid_halo_mask.  Ice on part of the domain,
    blocks without ice among the neighbours."""
    txt = run_case(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, False, kw, cgrid=cgrid, maskhalo=True)
    assert "maskhalo_dyn          =      T" in txt


@pytest.mark.parametrize("nx,ny,bx,by,ew,ns,nprocs,dist,kw", [
    (48, 40, 24, 20, "cyclic", "closed", 2, "cartesian", dict(grid_kind="popfile", icecase="full")),
    (60, 44, 20, 15, "closed", "closed", 3, "roundrobin", dict(grid_kind="popfile", icecase="patchy", h_visc_method="avg_strength")),
    (72, 40, 36, 20, "cyclic", "tripole", 2, "cartesian", dict(grid_kind="tripolefile", icecase="full")),
])
def test_reference_mpi_driver_with_hip_cgrid_loop_bitwise(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, kw):
    """C grid across MPI tasks: dyn_evp_hip_cgrid_run called by every task of the reference's driver (the eight halo
    updates of a subcycle between the tasks' kernels), against the reference's evp() with grid_ice = 'C' whose halo
    goes through MPI -- every array the loop writes, every cell of every task, ghost cells included."""
    run_case(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, False, kw, cgrid=True)


@pytest.mark.parametrize("nx,ny,bx,by,ew,ns,nprocs,dist,body,kw", CASES)
def test_reference_mpi_driver_with_hip_core_bitwise(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, body, kw):
    txt = run_case(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, body, kw)
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/dropin_mpi_{nx}x{ny}_{nprocs}_{dist}.log", "w") as f:
            f.write(txt[-20000:])
    except OSError:
        pass
