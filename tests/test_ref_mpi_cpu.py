"""CPU: the reference's own MPI path (comm/mpi/ice_boundary.F90, ice_communicate, ice_broadcast, ice_gather_scatter ...
compiled unmodified, in place, against the image's MPICH: oracle/ref/build_ref.sh `mpi`) run here under mpiexec.

What this pins: the fixtures and every reference-made comparison of this suite come from the serial-communication
build (comm/serial).  Here the SAME evp() is run with its blocks distributed over 2 and 4 MPI tasks -- ghost cells
travelling through the reference's MPI_ISEND / MPI_IRECV halo (ice_boundary.F90:1066-1760), tripole fold included --
and every output must equal the serial build's bit for bit, cell by global cell.  The oracle and the HIP path are
compared with the serial build elsewhere; with this, "the reference" means its MPI path too (north_star: the
reference's MPI/Fortran path).  The binaries are prebuilt (they travel to the GPU box); nothing reads /root/reference here."""
import numpy as np
import pytest

import run_ref
from cice_amd import synth
from common import bits_equal

CASES = [
    # nx, ny, bx, by, ew, ns, nprocs, distribution, harness kwargs
    (40, 36, 20, 18, "cyclic", "closed", 2, "cartesian", dict(grid_kind="rect", icecase="full")),
    (60, 44, 20, 15, "cyclic", "closed", 2, "roundrobin", dict(grid_kind="popfile", icecase="patchy")),
    (60, 44, 20, 15, "closed", "closed", 4, "cartesian", dict(grid_kind="popfile", icecase="caps", h_revised=True)),
    (72, 40, 36, 20, "cyclic", "tripole", 2, "cartesian", dict(grid_kind="tripolefile", icecase="full")),
    (72, 40, 18, 10, "cyclic", "tripole", 4, "roundrobin", dict(grid_kind="tripolefile", icecase="patchy", h_capping=0.5)),
    (72, 40, 18, 10, "cyclic", "tripoleT", 4, "cartesian", dict(grid_kind="tripolefile", icecase="full")),
    (48, 40, 24, 20, "cyclic", "closed", 4, "cartesian", dict(grid_kind="popfile", icecase="full", h_grid_ice="C")),
    (72, 40, 36, 10, "cyclic", "tripole", 2, "roundrobin", dict(grid_kind="tripolefile", icecase="full", h_grid_ice="C")),
    # a task that holds no blocks (4 x 2 blocks on 3 tasks, cartesian: 4, 4, 0): the reference carries on
    (54, 52, 14, 26, "cyclic", "tripole", 3, "cartesian", dict(grid_kind="tripolefile", icecase="full")),
    # maskhalo_dyn under MPI (ice_HaloMask, ice_boundary.F90:889-1062): strips without ice are not exchanged -- same bits
    (60, 48, 20, 12, "cyclic", "closed", 3, "roundrobin", dict(grid_kind="popfile", icecase="caps", maskhalo=True)),
    (72, 40, 18, 20, "cyclic", "tripole", 2, "cartesian", dict(grid_kind="tripolefile", icecase="patchy", h_grid_ice="C", maskhalo=True)),
]


def _grid(tmp_path, nx, ny, ns, kw):
    if kw["grid_kind"] == "rect":
        return None
    g = synth.make_grid(nx, ny, dx0=1.1e5, ns=ns)
    run_ref.write_pop_grid(tmp_path / "grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
    run_ref.write_kmt(tmp_path / "kmt.bin", g["kmt"])
    return tmp_path / "grid.bin", tmp_path / "kmt.bin"


@pytest.mark.parametrize("nx,ny,bx,by,ew,ns,nprocs,dist,kw", CASES)
def test_reference_mpi_path_equals_its_serial_build_bitwise(tmp_path, nx, ny, bx, by, ew, ns, nprocs, dist, kw):
    if not (run_ref.have_ref("mpistrict") and run_ref.have_ref("strict") and run_ref.have_mpiexec()):
        pytest.skip("oracle/_ref MPI build or mpiexec not available")
    kw = dict(kw)
    maskhalo = kw.pop("maskhalo", False)
    files = _grid(tmp_path, nx, ny, ns, kw)
    common = dict(ew=ew, ns=ns, h_ndte=24, ncalls=2, nsub_list=[1, 24], grid_files=files, **kw)
    ser, _ = run_ref.run_harness(nx, ny, bx, by, variant="strict", workdir=tmp_path / "ser", **common)
    par, txt = run_ref.run_harness(nx, ny, bx, by, variant="mpistrict", nprocs=nprocs, distribution_type=dist,
                                   maskhalo_dyn=maskhalo, workdir=tmp_path / "par", **common)
    assert ("maskhalo_dyn          =      T" in txt) == bool(maskhalo)
    assert len(par) == nprocs and all(int(d["dims"][2]) >= 0 for d in par)
    assert sum(int(d["dims"][2]) for d in par) == int(ser["dims"][2])       # the same blocks, dealt out
    outs = [k for k in ser if k[0] == "o" and k[3] == "n"]
    assert len(outs) >= 2 * 2 * 10
    moved = False
    for k in outs:
        a, b = run_ref.global_field(ser, k), run_ref.global_field(par, k)
        assert bits_equal(a, b), f"{k}: {int((a != b).sum())} cells differ between {nprocs} MPI tasks and the serial build"
        moved = moved or (k.endswith("uvel") or k.endswith("uvelE")) and np.nanmax(np.abs(a)) > 1e-6
    assert moved
