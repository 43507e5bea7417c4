"""TEST INFRASTRUCTURE -- run the compiled reference harness (oracle/_ref) and
read its dump.  Only tests/, tests/golden/make_golden.py, smoke() and the
bench's cpu_baseline leg may import this module; the product never does.

The harness binary is built by oracle/ref/build_ref.sh from the unmodified
reference sources in /root/reference (in place).  It travels to the GPU box as
a prebuilt file under oracle/_ref/, so it may be *executed* there, but nothing
here reads /root/reference at run time.
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF_DIR = HERE.parent / "_ref"


MPIEXEC = os.environ.get("CICE_REF_MPIEXEC", "/opt/conda/bin/mpiexec")   # the image's MPICH 3.3.2 (hydra)


def harness_path(variant: str = "strict") -> Path:
    if variant == "hip_dropin":
        # the reference's unmodified evp() driver + cice_amd/fortran shim + libcice_evp_hip.so
        return REF_DIR / "evp_hip_dropin_harness"
    if variant == "hip_dropin_mpi":
        # the same with the reference's comm/mpi modules: several MPI tasks drive the shim's nprocs > 1 branch
        return REF_DIR / "evp_hip_dropin_harness_mpi"
    return REF_DIR / f"evp_ref_harness_{variant}"


def is_mpi_variant(variant: str) -> bool:
    return variant.startswith("mpi") or variant.endswith("_mpi")


def have_mpiexec() -> bool:
    return os.path.exists(MPIEXEC) and os.access(MPIEXEC, os.X_OK)


def have_ref(variant: str = "strict") -> bool:
    p = harness_path(variant)
    return p.exists() and os.access(p, os.X_OK)


def read_dump(path) -> dict:
    """Parse the record stream written by evp_dumpio (see ice_dyn_evp1d_capture.F90):
    name(32 chars) | type code int32 (1=f64, 2=i32) | d1 d2 d3 int32 | payload (Fortran order)."""
    out = {}
    buf = Path(path).read_bytes()
    off = 0
    while off < len(buf):
        name = buf[off:off + 32].decode("ascii").strip()
        tcode, d1, d2, d3 = np.frombuffer(buf, dtype="<i4", count=4, offset=off + 32)
        off += 48
        n = int(d1) * int(d2) * int(d3)
        dt = np.dtype("<f8") if tcode == 1 else np.dtype("<i4")
        a = np.frombuffer(buf, dtype=dt, count=n, offset=off).copy()
        off += n * dt.itemsize
        if d2 == 1 and d3 == 1:
            out[name] = a
        else:
            # Fortran (d1,d2,d3) column-major  ->  numpy [d3][d2][d1] C-order (i fastest)
            out[name] = a.reshape(int(d3), int(d2), int(d1))
    return out


def _fmt(v):
    if isinstance(v, bool):
        return ".true." if v else ".false."
    if isinstance(v, str):
        return f"'{v}'"
    if isinstance(v, float):
        return repr(v).replace("e", "d") if "e" in repr(v) else repr(v) + "d0"
    if isinstance(v, (list, tuple)):
        return ", ".join(_fmt(x) for x in v)
    return str(v)


def write_pop_grid(path, ULAT, ULON, HTN_cm, HTE_cm, ANGLE=None):
    """POP-format binary grid file read by the reference's popgrid
    (/root/reference/cicecore/cicedyn/infrastructure/ice_grid.F90:1000-1061):
    7 direct-access records of nx_global*ny_global float64 (native endian here):
    ULAT ULON HTN HTE HUS HUW ANGLE.  Arrays are [ny][nx]."""
    ny, nx = ULAT.shape
    z = np.zeros((ny, nx))
    recs = [ULAT, ULON, HTN_cm, HTE_cm, HTN_cm, HTE_cm, z if ANGLE is None else ANGLE]
    with open(path, "wb") as f:
        for r in recs:
            f.write(np.ascontiguousarray(r, dtype="<f8").tobytes())


def write_kmt(path, kmt):
    with open(path, "wb") as f:
        f.write(np.ascontiguousarray(kmt, dtype="<i4").tobytes())


def run_harness(nx_global, ny_global, block_size_x, block_size_y, *,
                ew="cyclic", ns="closed", maskhalo_dyn=False, variant="strict",
                threads=1, workdir=None, grid_files=None, keep=False, timeout=3600,
                nprocs=1, distribution_type="cartesian", processor_shape="slenderX2", extra_env=None,
                mpiexec_args="",
                **harness):
    """Run one harness case, return (dump dict, stdout text).  nprocs > 1 (mpi* variants, started under the image's
    mpiexec): the reference distributes the blocks over that many MPI tasks (ice_domain.F90 init_domain_distribution)
    and the first return value is the LIST of per-task dumps in task order (global_field() assembles them)."""
    exe = harness_path(variant)
    if nprocs > 1 and not is_mpi_variant(variant):
        raise ValueError("nprocs > 1 needs one of the mpi variants")
    if not have_ref(variant):
        raise FileNotFoundError(f"{exe} not built (run oracle/ref/build_ref.sh)")
    tmp = None
    if workdir is None:
        tmp = tempfile.TemporaryDirectory(prefix="evpref_")
        workdir = tmp.name
    wd = Path(workdir)
    wd.mkdir(parents=True, exist_ok=True)
    (wd / "ice_in").write_text(
        "&domain_nml\n"
        f"  nprocs = {nprocs}\n"
        f"  nx_global = {nx_global}\n  ny_global = {ny_global}\n"
        f"  block_size_x = {block_size_x}\n  block_size_y = {block_size_y}\n"
        f"  max_blocks = -1\n  processor_shape = '{processor_shape}'\n"
        f"  distribution_type = '{distribution_type}'\n  distribution_wght = 'blockall'\n"
        f"  ew_boundary_type = '{ew}'\n  ns_boundary_type = '{ns}'\n"
        f"  maskhalo_dyn = {_fmt(bool(maskhalo_dyn))}\n"
        "  maskhalo_remap = .false.\n  maskhalo_bound = .false.\n"
        "  add_mpi_barriers = .false.\n  debug_blocks = .false.\n/\n")
    if grid_files is not None:
        gpath, kpath = grid_files
        harness.setdefault("h_grid_file", str(gpath))
        harness.setdefault("h_kmt_file", str(kpath))
    lines = ["&harness_nml"]
    for k, v in harness.items():
        lines.append(f"  {k} = {_fmt(v)}")
    lines.append("/\n")
    (wd / "harness_in").write_text("\n".join(lines))
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(threads)
    env.setdefault("OMP_SCHEDULE", "static,1")
    env.setdefault("OMP_STACKSIZE", "256M")
    # one 320x384 block needs a large stack (automatic arrays in evp()): SURVEY Appendix A.3
    if extra_env:
        env.update(extra_env)
    if is_mpi_variant(variant):
        if not have_mpiexec():
            raise FileNotFoundError(f"{MPIEXEC} not found")
        cmd = f"ulimit -s unlimited 2>/dev/null; exec '{MPIEXEC}' {mpiexec_args} -n {nprocs} '{exe}'"
    else:
        cmd = f"ulimit -s unlimited 2>/dev/null; exec '{exe}'"
    r = subprocess.run(["bash", "-c", cmd], cwd=wd, env=env, capture_output=True,
                       text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"reference harness failed rc={r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
    dumpname = harness.get("dumpfile", "dump.bin")
    if nprocs > 1:
        d = [read_dump(wd / f"{dumpname}.{r}") if (wd / f"{dumpname}.{r}").exists() else {} for r in range(nprocs)]
    else:
        d = read_dump(wd / dumpname) if (wd / dumpname).exists() else {}
    text = r.stdout
    if tmp is not None and not keep:
        tmp.cleanup()
    return d, text


def global_field(dumps, name, ghosts=False):
    """One [ny_global][nx_global] array out of a dump (or the list of per-task dumps of an MPI run): every block's
    interior cells put where blkinfo says they lie.  Cells no block covers (eliminated land blocks) stay NaN for
    fp64 fields and -1 for integer ones."""
    if isinstance(dumps, dict):
        dumps = [dumps]
    first = next(d for d in dumps if d)
    nxg, nyg = int(first["dims"][4]), int(first["dims"][5])
    sample = first[name]
    out = np.full((nyg, nxg), np.nan) if sample.dtype == np.float64 else np.full((nyg, nxg), -1, dtype=sample.dtype)
    for d in dumps:
        if not d:
            continue
        nb = int(d["dims"][2])
        blk = np.asarray(d["blkinfo"]).reshape(nb, 8)
        a = d[name]
        for b in range(nb):
            ilo, ihi, jlo, jhi, _, _, ig0, jg0 = (int(v) for v in blk[b])
            out[jg0 - 1:jg0 + (jhi - jlo), ig0 - 1:ig0 + (ihi - ilo)] = a[b, jlo - 1:jhi, ilo - 1:ihi]
    return out


def parse_wall(text):
    """`TIMING ... wall_s_total_evp_calls T` -- every task prints one; the slowest counts."""
    v = [float(x) for x in re.findall(r"wall_s_total_evp_calls\s+([0-9.Ee+-]+)", text)]
    return max(v) if v else None


def parse_timer(text, name="evp"):
    """Pull `Timer N: <name>  T seconds` out of ice_timer_print_all output."""
    m = re.search(r"Timer\s+\d+:\s+" + re.escape(name) + r"\s+([0-9.Ee+-]+)\s+seconds", text)
    return float(m.group(1)) if m else None
