#!/usr/bin/env python3
"""EVP subcycle benchmark (BASELINE.json metric: EVP subcycle cell-updates/sec,
gx1 fp64, at 1/2/4/8 GPUs; % of the roofline).

  python bench.py --gpus N --steps K --warmup W        (N > 1: under torch.distributed.run, or bare -- it then re-executes
                                                        itself under torch.distributed.run, one rank per GPU, 127.0.0.1)

A "step" is one evp() call's worth of the hot path: ndte subcycles
(stress + stepu + velocity halo, ice_dyn_evp.F90:859-913) on a resident synthetic
state.  Workload at every N: the gx1-sized grid (320x384, fp64, ndte=120, ice on
every ocean cell), block-decomposed over the N GPUs (strong scaling of the
headline config; `--workload s01` selects the synthetic 3600x2400 grid).
Rank 0 prints ONE JSON line.

What the line carries besides the contract's fields:
  roofline      the bound that binds the timed kernel.  On-chip resident kernel: fp64 VALU issue
                (VALU-busy SIMD-cycles per launch from the committed rocprofv3 PMC pass of this
                command / live kernel time / 1024 SIMDs x 2.4 GHz); the 368 B/cell figure is kept as a
                labelled yardstick (`hbm_equivalent`).  Streaming kernel: HBM (`streaming` holds its
                fractions on gx1 and on 3600x2400, measured live in this run, with PMC traffic).
  verified      the state after the timed region (+ a few untimed steps up to the next checkpoint)
                hashed and compared with tests/golden/bench_checksums.json (made by the CPU oracle).
  median_ms_per_step   N = 1: the same loop ten more times, one call at a time (HIP events of the library): SURVEY 8(d)'s median
                next to the contract's K-step mean (`ms_per_step`, `value`).
  caps          gx1 with ice on the polar caps only (SURVEY 8(d)'s second, realistic ice case): cell-updates/s over all cells and
                over the cells with ice, the tiles the resident kernel ran (only the ones that hold ice), verified
  cgrid         the C-grid subcycle on gx1, on 3600x2400 and on the tripole grid tx1: microseconds per subcycle, verified
                against committed oracle checksums; `kernel` says what ran (gx1: the on-chip resident kernel cg_res, every
                subcycle of a call but the first after an upload in one launch; 3600x2400: one launch per subcycle, HBM
                fraction on its 289 B per cell).
  rccl_control  N > 1: the headline workload once more with every remote ghost cell carried by RCCL point-to-point.
  attempts      N > 1 only: every sync point is an agreement over the ranks; a failed or unverified attempt is
                repeated by all ranks with the resident kernel off, then with RCCL only (config.attempts).
  configs2_gx1_ndte240, tripole, secondary
                N > 1: the other BASELINE configs on the same ranks -- gx1 at ndte = 240 (library default, and forced
                onto RCCL point-to-point), tx1 with the tripole seam (cut in y), 3600x2400 at ndte = 480 -- each
                verified against its committed checksum, each with every rank's own view (per_rank).
  cpu_baseline  the reference's own evp() timed on this box's cores (its MPI path under mpiexec, and its 2-d path and
                1-d core under OpenMP; the fastest is `value`, all three are in `paths`), and --
                the checker's job -- the HIP path run on the inputs the reference captured in this
                same run, compared bit for bit with the reference's outputs (`reference_parity`).
"""
from __future__ import annotations

import argparse
import contextlib
import ctypes as C
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

B_ALG = 368.0            # algorithmic bytes per cell-subcycle (32 reads + 14 writes, fp64)
B_PASS_MARCH = 328.0     # marching kernel: 27 reads + 14 writes (fp64) per cell and PASS of four (or three, two) subcycles
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD = 1024            # 256 CUs x 4 SIMDs
MAX_CLOCK_HZ = 2.4e9
FP64_VALU_PEAK_TFLOPS = 78.6        # dense fp64 vector peak with FMA; 39.3 without (strict mode never contracts)
# nominal arithmetic of one cell-subcycle counted on cice_amd/csrc/evp_cell.inc (classic EVP, capping 1): stress 376
# mul/add + 4 sqrt + 4 div, stepu 46 mul/add + 1 sqrt + 2 div -- every operation counted as one flop
ALG_FLOP = 433.0
CHECKPOINTS = [1, 3, 5, 12, 23, 25, 50]     # total steps after which tests/golden/bench_checksums.json holds a hash
VERIFY_FIELDS = ("uvel", "vvel", "stressp_1")
CGRID_VERIFY_FIELDS = ("uvelE", "vvelN", "stresspT", "stress12U")   # tests/golden/make_bench_checksums.py
CGRID_B_ALG = 648.0      # C grid: 81 fp64 array touches per cell and subcycle in the fused schedule
CGRID_B_ALG_GEO = 563.0  # ... 70 + three mask bytes where the fused kernels derive 15 of the 23 static arrays (as cg_one does)
EXTRAS = ("s01", "streaming", "tripole", "caps", "cgrid", "per_call", "configs2", "rccl_control", "ring_variants")
# what a plain run measures besides the headline: everything on one GPU; at N > 1 the 3600x2400 grid on the library's default
# path and ONE control (the headline workload forced onto RCCL point-to-point) -- the other forms are asked for by name
EXTRAS_DEFAULT = {1: ("s01", "streaming", "tripole", "caps", "cgrid", "per_call"), 2: ("s01", "rccl_control")}
CGRID_B_ALG_ONE = 408.0  # ... 51 in the one-launch kernel (cg_one: the default on one rank without a fold)
CGRID_B_ALG_ONE_GEO = 289.0  # ... 36 + one mask byte where cg_one derives 15 of its 23 static arrays from the 8 dx / dy arrays


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gx1", choices=["gx3", "gx1", "s01"])
    ap.add_argument("--case", default="full", choices=["full", "caps"])
    ap.add_argument("--ndte", type=int, default=None)
    ap.add_argument("--fused", action="store_true",
                    help="allow FMA contraction (default: strict fp64, bit-identical to the reference built without FMA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the extra measurements reported under 'secondary', 'tripole', 'roofline.streaming', 'per_call_ms'")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline wall time per code path")
    ap.add_argument("--extras", default="auto",
                    help="comma list of the extra measurements to run, or 'all'.  Default ('auto') at N = 1: s01, streaming, "
                         "tripole, cgrid, per_call; at N > 1: s01 (3600x2400 at ndte = 480, library default) and rccl_control (the "
                         "headline workload forced onto RCCL point-to-point).  By name only, N > 1: configs2 (gx1 at ndte = 240, "
                         "default and forced RCCL), tripole (tx1 over the ranks), ring_variants (s01 again with the ring exchange "
                         "overlapped / as direct IPC stores)")
    a = ap.parse_args()
    a.extras = set(EXTRAS if a.extras == "all" else EXTRAS_DEFAULT[min(a.gpus, 2)] if a.extras == "auto"
                   else [x for x in a.extras.split(",") if x])
    unknown = a.extras - set(EXTRAS)
    if unknown:
        ap.error(f"unknown --extras {sorted(unknown)} (known: {', '.join(EXTRAS)})")
    return a


def state_hash(glob: dict) -> str:
    h = hashlib.sha256()
    for k in VERIFY_FIELDS:
        h.update(np.ascontiguousarray(glob[k], dtype="<f8").tobytes())
    return h.hexdigest()


def golden_key(workload, case, ndte, ns):
    return f"{workload}/{case}/ndte{ndte}/{ns}/strict"


def load_golden():
    try:
        return json.loads((ROOT / "tests" / "golden" / "bench_checksums.json").read_text())
    except Exception:  # noqa: BLE001
        return {}


# What this chip sustains in fp64 wave64 instructions per SIMD with every SIMD busy (tools/fp64_rate_probe.hip; the clock
# drops to ~2.0 GHz under an all-fp64 load): ns per v_mul_f64 / v_add_f64 at 1, 2, 4 waves per SIMD; v_rcp_f64 / v_sqrt_f64
FP64_NS_PER_INST = {1: 2.40, 2: 2.15, 4: 2.00}
FP64_NS_QUARTER_RATE = 6.9
QUARTER_RATE_PER_CELL_SUBCYCLE = 11          # 6 IEEE divisions (one v_rcp_f64 each) + 5 square roots


def measured_issue(e, t_kernel, sub_per_launch, active_cells):
    """The resident kernel against the fp64 instruction throughput the chip really sustains (two ice-holding waves per SIMD)."""
    if not e or "SQ_INSTS_VALU" not in e.get("counters", {}):
        return None
    insts = e["counters"]["SQ_INSTS_VALU"]["avg_per_launch"]
    quarter = QUARTER_RATE_PER_CELL_SUBCYCLE * active_cells / 64.0 * sub_per_launch
    t_floor = (insts * FP64_NS_PER_INST[2] + quarter * (FP64_NS_QUARTER_RATE - FP64_NS_PER_INST[2])) * 1e-9 / N_SIMD
    return {"valu_wave_instructions_per_launch": insts, "ns_per_wave_instruction_per_simd": FP64_NS_PER_INST,
            "ns_per_quarter_rate_instruction": FP64_NS_QUARTER_RATE, "floor_us_per_launch": 1e6 * t_floor,
            "frac_of_measured_rate": t_floor / t_kernel, "source": "profiles/r03_fp64_rate_probe.txt (tools/fp64_rate_probe.hip)",
            "note": "the kernel's VALU instruction count (PMC) priced at what one SIMD issues per ns with two waves on it, every SIMD "
                    "of the chip busy -- 4 cycles per fp64 instruction at the ~2.0 GHz the chip holds under that load, not at 2.4 GHz; "
                    "`frac` above stays on the nominal clock"}


def load_pmc():
    """The newest tracked PMC summary (tools/profile_gpu.sh + tools/pmc_summary.py on this command)."""
    files = sorted((ROOT / "profiles").glob("r*_pmc_summary.json"))
    if not files:
        return None, None
    try:
        return json.loads(files[-1].read_text()), f"profiles/{files[-1].name}"
    except Exception:  # noqa: BLE001
        return None, None


# ----------------------------------------------------------------------------------------------
# cpu_baseline leg (rank 0, N = 1): the only part of this file that touches oracle/
# ----------------------------------------------------------------------------------------------
def cpu_baseline(workload, case, ndte, target_s, strict):
    """The reference's own evp() (oracle/_ref: the unmodified reference sources compiled with amdflang
    -O2 -fopenmp, linked against the Icepack interface stub) timed on this box's host cores by its own
    timer_evp -- the standard 2-d path and the 1-d "shared_mem_1d" core -- plus, as the checker, the HIP
    path on the very inputs the reference captured here, compared bit for bit with its outputs.
    Falls back to the C restatement ("port") if the prebuilt reference binaries did not travel."""
    from cice_amd import synth
    spec = synth.GRIDS[workload]
    nx, ny = spec["nx"], spec["ny"]
    cores = len(os.sched_getaffinity(0))
    sys.path.insert(0, str(ROOT / "oracle" / "ref"))
    sys.path.insert(0, str(ROOT / "oracle"))
    sys.path.insert(0, str(ROOT / "tests"))
    try:
        import run_ref
        import tempfile
        if not run_ref.have_ref("fast"):
            raise FileNotFoundError("oracle/_ref/evp_ref_harness_fast")
        g = synth.make_grid(nx, ny, spec["dx0"], ns="closed")
        td = tempfile.mkdtemp(prefix="evpcpu_")
        run_ref.write_pop_grid(td + "/grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
        run_ref.write_kmt(td + "/kmt.bin", g["kmt"])

        def ref_run(variant, bx, by, threads, ncalls, nprocs=1, calibrate=False, **kw):
            mpi = dict(nprocs=nprocs, distribution_type="roundrobin", mpiexec_args=kw.pop("mpiexec_args", "")) if nprocs > 1 else {}
            d, txt = run_ref.run_harness(nx, ny, bx, by, ew="cyclic", ns="closed", variant=variant,
                                         threads=threads, grid_kind="popfile", icecase=case,
                                         grid_files=(td + "/grid.bin", td + "/kmt.bin"),
                                         h_ndte=ndte, ncalls=1, nsub_list=[ndte], dump_arrays=False,
                                         ntiming=ncalls, timeout=900, **mpi, **kw)
            if nprocs > 1 and calibrate:     # two calls: timer_evp prints with 0.01 s resolution -- the wall clock around them
                return run_ref.parse_wall(txt)
            return run_ref.parse_timer(txt, "evp")       # timer_evp is cleared before the ntiming calls (MPI: max over tasks)

        # candidates: the 2-d path is OpenMP over blocks (threads <= blocks); the 1-d core over the cell vector
        cands = []
        for bdiv in (8, 16):
            bx, by = max(nx // bdiv, 8), max(ny // bdiv, 8)
            nblk = (nx // bx) * (ny // by)
            cands.append(dict(path="standard_2d", variant="fast", bx=bx, by=by, threads=min(cores, nblk), kw={}))
        if run_ref.have_ref("fast1d"):      # (256 threads on its flat vector measured 40x slower than 16: try a few team sizes)
            for th in sorted({min(cores, 16), min(cores, 64)}):
                cands.append(dict(path="shared_mem_1d", variant="fast1d", bx=max(nx // 8, 8), by=max(ny // 8, 8),
                                  threads=th, kw=dict(time_1d=True)))
        # the reference's MPI path (comm/mpi compiled against the image's MPICH, oracle/ref/build_ref.sh `mpi`): one block
        # per MPI task, as many tasks as blocks -- what north_star names as the baseline ("reference MPI/Fortran path")
        if run_ref.have_ref("mpifast") and run_ref.have_mpiexec():
            for (dx_, dy_) in ((4, 4), (8, 4), (8, 8), (16, 8)):
                if dx_ * dy_ <= cores and nx % dx_ == 0 and ny % dy_ == 0:
                    for bind in ("", "-bind-to core"):      # hydra leaves the tasks to the scheduler unless told otherwise
                        cands.append(dict(path="mpi", variant="mpifast", bx=nx // dx_, by=ny // dy_, threads=1, nprocs=dx_ * dy_,
                                          kw=dict(mpiexec_args=bind)))
        results = []
        for c in cands:
            try:
                t_cal = ref_run(c["variant"], c["bx"], c["by"], c["threads"], 2, nprocs=c.get("nprocs", 1), calibrate=True, **dict(c["kw"]))
            except Exception as e:  # noqa: BLE001
                if c["path"] != "mpi":
                    raise
                print(f"[bench] reference MPI path with {c.get('nprocs')} tasks failed ({type(e).__name__}: {str(e)[:300]})", file=sys.stderr)
                continue
            if not t_cal or t_cal <= 0:
                continue
            c["per_call"] = t_cal / 2.0
            results.append(c)
        if not results:
            raise RuntimeError("no timing from the reference harness")
        best2d = min((c for c in results if c["path"] == "standard_2d"), key=lambda c: c["per_call"], default=None)
        best1d = min((c for c in results if c["path"] == "shared_mem_1d"), key=lambda c: c["per_call"], default=None)
        bestmpi = min((c for c in results if c["path"] == "mpi"), key=lambda c: c["per_call"], default=None)
        timed = []
        for c in [c for c in (best2d, best1d, bestmpi) if c]:
            ncalls = int(max(2, min(2000, target_s / max(c["per_call"], 1e-4))))
            t = ref_run(c["variant"], c["bx"], c["by"], c["threads"], ncalls, nprocs=c.get("nprocs", 1), **dict(c["kw"]))
            if t and t > 0:
                timed.append(dict(path=c["path"], value=nx * ny * ndte * ncalls / t, cores=c.get("nprocs", c["threads"]),
                                  blocks=f"{(nx // c['bx']) * (ny // c['by'])} x {c['bx']}x{c['by']}",
                                  evp_calls=ncalls, timer_evp_s=t,
                                  parallelism=(f"{c['nprocs']} MPI tasks (MPICH 3.3.2, shared memory; mpiexec {c['kw'].get('mpiexec_args') or 'unbound'}), one block each" if c["path"] == "mpi"
                                               else f"{c['threads']} OpenMP threads")))
        if not timed:
            raise RuntimeError("no timing from the reference harness")
        top = max(timed, key=lambda r: r["value"])
        build = ("amdflang -O2, comm/mpi against the image's MPICH 3.3.2; its own timer_evp, slowest task"
                 if top["path"] == "mpi" else "amdflang -O2 -fopenmp, comm/serial; its own timer_evp")
        out = dict(value=top["value"], unit="cell-updates/s", cores=top["cores"], kind="reference",
                   sample=f"reference evp() ({top['path']}) compiled from the unmodified sources ({build}), "
                          f"{nx}x{ny} in {top['blocks']} blocks, ndte={ndte}, {top['evp_calls']} evp() calls in "
                          f"{top['timer_evp_s']:.2f}s (subcycle loop incl. halo updates + deformations), "
                          f"{top['parallelism']} on {cores} host cores; fastest of the code paths in `paths` "
                          f"(standard_2d and shared_mem_1d: comm/serial + OpenMP; mpi: the reference's comm/mpi halo, MPI_ISEND/IRECV)",
                   paths=timed, host_cores=cores,
                   calibration=[dict(path=c["path"], tasks=c.get("nprocs", 1), threads=c["threads"], mpiexec_args=c["kw"].get("mpiexec_args"),
                                     blocks=f"{c['bx']}x{c['by']}", s_per_call=c["per_call"]) for c in results])
        if strict and run_ref.have_ref("strict"):
            out["reference_parity"] = reference_parity(nx, ny, ndte, case, td)
        try:      # the same code with grid_ice = 'C' (next-tier row f-4), timed the same way
            bx, by = best2d["bx"], best2d["by"]
            tC = ref_run("fast", bx, by, best2d["threads"], 4, h_grid_ice="C")
            if tC and tC > 0:
                out["cgrid"] = dict(value=nx * ny * ndte * 4 / tC, unit="cell-updates/s", cores=best2d["threads"], kind="reference",
                                    sample=f"reference evp() with grid_ice='C', {nx}x{ny} in {(nx // bx) * (ny // by)} blocks of {bx}x{by}, "
                                           f"ndte={ndte}, 4 calls, timer_evp={tC:.2f}s")
            if bestmpi:      # ... and on its MPI path, with the task count that was fastest for the B grid
                mb = dict(bestmpi["kw"])
                tM = ref_run("mpifast", bestmpi["bx"], bestmpi["by"], 1, 16, nprocs=bestmpi["nprocs"], h_grid_ice="C", **mb)
                if tM and tM > 0:
                    vM = nx * ny * ndte * 16 / tM
                    omp = out.get("cgrid")
                    if not omp or vM > omp["value"]:
                        out["cgrid"] = dict(value=vM, unit="cell-updates/s", cores=bestmpi["nprocs"], kind="reference",
                                            sample=f"reference evp() with grid_ice='C' on its MPI path, {nx}x{ny} as {bestmpi['nprocs']} MPI tasks of one "
                                                   f"{bestmpi['bx']}x{bestmpi['by']} block each (mpiexec {mb.get('mpiexec_args') or 'unbound'}), ndte={ndte}, "
                                                   f"16 calls, timer_evp={tM:.2f}s",
                                            openmp=omp)
        except Exception as e:  # noqa: BLE001
            out["cgrid"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        return out
    except Exception as e:  # noqa: BLE001
        print(f"[bench] reference CPU baseline unavailable ({type(e).__name__}: {e}); using the C port", file=sys.stderr)
    # port: the oracle's C restatement with OpenMP
    import oracle  # noqa: F401
    from test_gpu_parity import run_oracle, synth_case
    scal = synth.evp_scalars(ndte)
    dc, geo, fields, tm, um = synth_case(workload, case, seed=1, bs=(max(nx // 8, 8), max(ny // 8, 8)))
    nsub = 24
    run_oracle(dc, geo, fields, tm, um, scal, 2)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < target_s and reps < 50:
        run_oracle(dc, geo, fields, tm, um, scal, nsub)
        reps += 1
    t = time.perf_counter() - t0
    return dict(value=nx * ny * nsub * reps / t, unit="cell-updates/s", cores=cores, kind="port",
                sample=f"oracle/evp_oracle.c (gcc -O2 -fopenmp), {nx}x{ny}, {nsub * reps} subcycles incl. "
                       f"setup copies, {cores} threads")


def reference_parity(nx, ny, ndte, case, td):
    """Checker: the reference's evp() (strict build) run HERE, one block, its subcycle inputs captured at
    the drop-in boundary; the HIP path (whatever kernel the autotuner picks, i.e. the timed one) on those
    inputs; every output field compared bit for bit with the reference's."""
    import run_ref
    import common
    from common import GoldenCase
    from cice_amd import evp
    d, _ = run_ref.run_harness(nx, ny, nx, ny, ew="cyclic", ns="closed", variant="strict", h_ndte=ndte, ncalls=1,
                               nsub_list=[ndte], grid_kind="popfile", icecase=case,
                               grid_files=(td + "/grid.bin", td + "/kmt.bin"), timeout=900)
    np.savez(td + "/case.npz", **d, ew=np.array("cyclic"), ns=np.array("closed"))
    old = common.GOLDEN
    common.GOLDEN = Path(td)
    try:
        c = GoldenCase("case")
    finally:
        common.GOLDEN = old
    dd, keep = c.hip_dims()
    core = evp.EvpHip(dd, evp.make_params(c.scal_dict(), strict=True), c.d["HTE"], c.d["HTN"], c.d["dxT"], c.d["dyT"],
                      c.d["uarear"], c.d["tarea"], keepalive=keep)
    try:
        dyn, tm, um = c.inputs(1)
        out = core.run(dyn, tm, um, ndte=ndte)
        tv = core.timings()["tile_variant"]
    finally:
        core.finalize()
    want = c.expected(1, ndte)
    bad = [k for k, w in want.items() if not np.array_equal(np.ascontiguousarray(out[k], dtype=np.float64).view(np.uint64),
                                                            np.ascontiguousarray(w, dtype=np.float64).view(np.uint64))]
    return dict(bitwise=not bad, fields_compared=len(want), fields_differing=bad, subcycles=ndte,
                tile_variant=tv, max_abs_u=float(np.abs(out["uvel"]).max()),
                how="reference evp() (strict build, 1 block) run in this leg; its captured subcycle inputs -> "
                    "cice_evp_hip_run -> outputs vs the reference's, all 18 fields incl. ghost cells")


# ----------------------------------------------------------------------------------------------
@contextlib.contextmanager
def quiet_interpreter():
    """Per-call timings: collect the interpreter's garbage first (earlier measurements leave multi-GB arrays whose release
    took 45-56 ms of one call in ten when a collection happened to fall into it) and keep the collector off meanwhile,
    as timeit does."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def main():
    a = parse()
    a.strict = not a.fused
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # launched bare (`python bench.py --gpus N`): become the launcher the contract names -- one rank per GPU under
        # torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]])
    import torch
    import torch.distributed as dist
    from cice_amd import decomp, evp, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    # Rehearsal on a 1-GPU box (tests only): CICE_EVP_BENCH_REHEARSAL=1 runs the N ranks as N processes
    # on device 0 over gloo, the mailbox halo bootstrapped by hand (RCCL refuses two ranks per device)
    rehearsal = world > 1 and os.environ.get("CICE_EVP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("CICE_EVP_HIP_DEVICE", str(local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("CICE_EVP_HIP_VERBOSE", "1")     # why a transport was not taken goes to stderr
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    golden = load_golden()

    class AttemptFailed(RuntimeError):
        """Some rank's device work failed (a wait gave up, a launch error): every rank leaves the attempt together."""

    def agree(ok: bool) -> bool:
        """The ranks' verdicts on the work since the last agreement, MIN over ranks.  Every rank runs the same sequence
        of these whether or not its own work failed, so a failure never leaves ranks in different collectives."""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def measure(workload, case, ndte, steps, warmup, ns="closed", env=None, verify=True, proc_shape=None, again_env=None, median_calls=0):
        """One timed pass: `warmup` untimed + `steps` timed evp() subcycle loops of `workload`,
        block-decomposed over the ranks; barrier + sync on both sides, MAX over ranks."""
        saved = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            spec = synth.GRIDS[workload]
            nx, ny = spec["nx"], spec["ny"]
            g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=ns))
            st = synth.make_state(g, case=case, seed=20260928, warm=True)
            dc = decomp.per_rank_blocks(nx, ny, world, "cyclic", ns, proc_shape)
            geo = {k: dc.scatter(g[k], rank, fill=(1.0 if k != "uarear" else 0.0))
                   for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
            fields = {k: dc.scatter(st[k], rank) for k in evp.FIELDS}
            tm = dc.scatter(st["iceTmask"], rank, fill=0)
            um = dc.scatter(st["iceUmask"], rank, fill=0)
            n_active = int(st["iceTmask"].sum())
            fold_metrics = synth.bgrid_fold_metrics(dc, rank, g) if ns == "tripole" else None
            del g, st

            scal = synth.evp_scalars(ndte)
            d, keep = evp.make_dims(dc, rank)
            # (rehearsal on one GPU: the marching path's ring exchanges go through the test build's host transport --
            # gloo underneath -- because RCCL refuses two ranks per device; a real run uses the product library and RCCL)
            core = evp.EvpHip(d, evp.make_params(scal, strict=a.strict), geo["HTE"], geo["HTN"], geo["dxT"],
                              geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep, testing=(True if rehearsal else None))
            try:
                def rehearsal_transport():
                    def xchg(ranks, ns_, nr_, send, recv):
                        ops, so, ro, keepalive = [], 0, 0, []
                        for q, n_s, n_r in zip(ranks, ns_, nr_):
                            if q == rank:
                                recv[ro:ro + n_r] = send[so:so + n_s]
                            else:
                                if n_s:
                                    ts = torch.from_numpy(np.ascontiguousarray(send[so:so + n_s]))
                                    keepalive.append(ts)
                                    ops.append(dist.P2POp(dist.isend, ts, q))
                                if n_r:
                                    tr = torch.empty(n_r, dtype=torch.float64)
                                    keepalive.append((tr, ro, n_r))
                                    ops.append(dist.P2POp(dist.irecv, tr, q))
                            so += n_s
                            ro += n_r
                        if ops:
                            for w in dist.batch_isend_irecv(ops):
                                w.wait()
                        for item in keepalive:
                            if isinstance(item, tuple):
                                tr, o, n = item
                                recv[o:o + n] = tr.numpy()

                    def reduce(op, v):
                        t = torch.tensor([v], dtype=torch.int64)
                        dist.all_reduce(t, op=dist.ReduceOp.MIN if op == 0 else dist.ReduceOp.MAX)
                        return int(t.item())
                    core.set_test_transport(xchg, reduce)

                def setup():
                    if ns == "tripole":      # CICE's own dxhy / dyhx: their north ghost row is a sign-flipped mirror image
                        core.set_metrics(**dict(zip(("dxhy", "dyhx"), fold_metrics)))
                    if world > 1 and rehearsal:
                        blobs = [None] * world
                        dist.all_gather_object(blobs, core.halo_export())
                        core.halo_import(blobs)
                        rehearsal_transport()
                    elif world > 1:
                        uid = [core.comm_unique_id() if rank == 0 else None]
                        dist.broadcast_object_list(uid, src=0)
                        core.comm_init(uid[0])
                    core.upload(fields, tm, um)

                def sync_point(work):
                    """`work`, then wait for the device; the barrier is an agreement on "still fine" (MIN over ranks)."""
                    err = None
                    try:
                        work()
                        core.sync()
                        torch.cuda.synchronize()
                    except Exception as e:  # noqa: BLE001
                        err = e
                    t_done = time.perf_counter()
                    if not agree(err is None):
                        raise AttemptFailed(f"rank {rank}: {err}" if err else f"rank {rank}: another rank failed")
                    return t_done

                def warm():
                    setup()          # (transport set-up and its probes belong to the attempt: a failure there is agreed on too)
                    for _ in range(warmup):
                        core.subcycle(ndte)

                def timed():
                    core.mark(0)
                    for _ in range(steps):
                        core.subcycle(ndte)
                    core.mark(1)

                sync_point(warm)
                t0 = time.perf_counter()
                t1 = sync_point(timed)
                dt = t1 - t0
                if world > 1:
                    tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt.item())
                # HIP events on the library's stream around the timed region (rank 0's share)
                tm_ev = core.timings()
                kt = core.time_kernels(200)
                # N > 1: every rank's view of the timed region, so that a first run on real xGMI can be read -- its own wall
                # time, the time its stream spent between the marks, its kernel alone (no exchange: HIP events around
                # repeated launches on scratch state), which kernel / transport it ended up with, what it sends per exchange
                per_rank = None
                if world > 1:
                    mi = core.march_info()
                    mine = dict(rank=rank, wall_ms=1e3 * (t1 - t0), stream_ms=float(tm_ev["marks_ms"]),
                                us_per_subcycle_stream=1e3 * float(tm_ev["marks_ms"]) / (steps * ndte),
                                kernel_alone_us_per_launch=1e3 * float(kt["stencil_ms"]), halo_kernel_us=1e3 * float(kt["halo_ms"]),
                                back_to_back_us=1e3 * float(kt["stencil_period_ms"]),
                                tile_variant=int(tm_ev["tile_variant"]), halo_transport=tm_ev["halo_transport"],
                                launches_per_subcycle=float(tm_ev["launches_per_subcycle"]),
                                halo_send_cells=int(tm_ev["halo_send_cells"]), halo_recv_cells=int(tm_ev["halo_recv_cells"]),
                                resident_fallbacks=int(tm_ev["resident_fallbacks"]),
                                two_subcycle_kernel=dict(ran=bool(mi["last_call"]), passes=mi["passes"], declined=mi["declined"],
                                                         strips=mi["strips"], segments=mi["segments"], rows_per_segment=mi["seglen"]),
                                probes_us=dict(streaming=1e3 * float(tm_ev["stream_probe_ms"]), resident=1e3 * float(tm_ev["resident_probe_ms"])),
                                local_cells=int(sum(b.gnx * b.gny for b in dc.local_blocks(rank))), path=core.describe_path(),
                                # what RCCL itself reports (ncclCommCount / UserRank / CuDevice) and the device's PCI bus id: "did RCCL
                                # see N ranks on N different GPUs" without trusting a label of ours
                                **{k: v for k, v in core.comm_info().items() if k != "have_comm"})
                    per_rank = [None] * world
                    dist.all_gather_object(per_rank, mine)
                # SURVEY 8(d) asks for the median over >= 10 evp() calls next to the contract's K-step mean: the same loop again,
                # one call at a time (HIP events of the library around each loop; one GPU only -- across ranks every step would
                # need its own agreement)
                each_ms = []
                if world == 1 and median_calls:
                    for _ in range(median_calls):
                        core.subcycle(ndte)
                        core.sync()
                        each_ms.append(float(core.timings()["loop_ms"]))
                # ---- what was timed, checked: continue (untimed) to the next checkpoint, hash the state ----
                total = warmup + steps + len(each_ms)
                key = golden_key(workload, case, ndte, ns)
                target = next((n for n in CHECKPOINTS if n >= total and str(n) in golden.get(key, {})), None)
                ver = dict(verified=None, why="no committed checksum for this configuration", key=key)
                if not a.strict:
                    ver["why"] = "fused mode is not bit-comparable with the oracle (tolerance: DESIGN.md)"
                elif verify and target is not None:
                    for _ in range(target - total):
                        core.subcycle(ndte)
                    core.sync()
                out = core.download()
                fallbacks = core.timings()["resident_fallbacks"]
                # the same ranks, state and communicator once more under switches the library reads at every call
                # (again_env): one untimed + `steps` timed loops continuing from where the verified run ended
                again = None
                for env_k in ([again_env] if isinstance(again_env, dict) else (again_env or [])):
                    saved2 = {k: os.environ.get(k) for k in env_k}
                    os.environ.update(env_k)
                    try:
                        sync_point(lambda: core.subcycle(ndte))
                        ta = time.perf_counter()
                        tb = sync_point(lambda: [core.subcycle(ndte) for _ in range(steps)])
                        dta = tb - ta
                        if world > 1:
                            tt2 = torch.tensor([dta], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                            dist.all_reduce(tt2, op=dist.ReduceOp.MAX)
                            dta = float(tt2.item())
                        o2 = core.download()
                        again = (again or []) + [dict(dt=dta, finite=bool(np.isfinite(o2["uvel"]).all()), env=dict(env_k),
                                                      ring=core.march_info().get("ring"))]
                    finally:
                        for k, v in saved2.items():
                            if v is None:
                                os.environ.pop(k, None)
                            else:
                                os.environ[k] = v
                if a.strict and verify and target is not None:
                    mine = {k: out[k] for k in VERIFY_FIELDS}
                    parts = [mine]
                    if world > 1:
                        parts = [None] * world
                        dist.all_gather_object(parts, mine)
                    if rank == 0:
                        glob = {k: dc.gather({r: parts[r][k] for r in range(world)}) for k in VERIFY_FIELDS}
                        want = golden[key][str(target)]
                        got = state_hash(glob)
                        ver = dict(verified=bool(got == want["sha256"]), total_steps=target, subcycles=target * ndte,
                                   fields=list(VERIFY_FIELDS), sha256=got[:16], expected=want["sha256"][:16], key=key,
                                   sum_abs_u=float(np.abs(glob["uvel"]).sum()), expected_sum_abs_u=want["sum_abs_u"],
                                   how="sha256 over the global interior arrays after all timed launches (+ untimed steps up "
                                       "to the checkpoint) vs tests/golden/bench_checksums.json (CPU oracle, pinned bit for "
                                       "bit to the reference)")
            finally:
                core.finalize()
            return dict(nx=nx, ny=ny, ndte=ndte, dc=dc, tm=tm, n_active=n_active, dt=dt, tm_ev=tm_ev, kt=kt, per_rank=per_rank,
                        steps=steps, warmup=warmup, ver=ver, fallbacks=fallbacks, again=again, each_ms=each_ms,
                        finite=bool(np.isfinite(out["uvel"]).all() and np.isfinite(out["stressp_1"]).all()),
                        umax=float(np.abs(out["uvel"]).max()))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def measure_with_fallbacks(*args, **kw):
        """N > 1: the preferred path (whatever the library's collective start-up probes settle on) first; if some rank's
        device work fails at run time, or the result does not verify, all ranks retry together with the on-chip resident
        kernel off, then with RCCL point-to-point only.  What was tried is reported under `attempts`."""
        base = dict(kw.pop("env", None) or {})
        ladder = [({}, "library default")]
        if world > 1:
            ladder += [({"CICE_EVP_HIP_RESIDENT": "0"}, "streaming kernels (resident kernel across GPUs off)"),
                       ({"CICE_EVP_HIP_RESIDENT": "0", "CICE_EVP_HIP_MARCH": "0"},
                        "one-subcycle streaming kernel (resident kernel and marching kernel off)"),
                       ({"CICE_EVP_HIP_RESIDENT": "0", "CICE_EVP_HIP_MARCH": "0", "CICE_EVP_HIP_HALO": "rccl"},
                        "one-subcycle streaming kernel, RCCL point-to-point only")]
        attempts = []
        for extra_env, label in ladder:
            try:
                Mx = measure(*args, env={**base, **extra_env}, **kw)
                bad = (not Mx["finite"]) or (Mx["ver"].get("verified") is False)
                # (verification is computed on rank 0; every rank learns the verdict)
                if agree(not bad) or world == 1:
                    attempts.append({"path": label, "ok": not bad})
                    Mx["attempts"] = attempts
                    return Mx
                attempts.append({"path": label, "ok": False, "why": "state after the timed region does not match the committed checksum"})
            except AttemptFailed as e:
                attempts.append({"path": label, "ok": False, "why": str(e)[:300]})
                if rank == 0:
                    print(f"[bench] attempt '{label}' failed: {e}", file=sys.stderr)
        raise RuntimeError(f"no path produced a verified result: {attempts}")

    def cgrid_measure(workload, case, ndte, steps, warmup):
        """The C-grid subcycle (evp()'s loop for grid_ice = 'C') on one GPU: `steps` timed loops of
        `ndte` subcycles on the resident state, HIP events around each captured loop; the state after warmup + steps
        loops is hashed against the oracle's committed checksum."""
        spec = synth.GRIDS[workload]
        nx, ny = spec["nx"], spec["ny"]
        ns = spec.get("ns", "closed")           # (tx1: the tripole grid of BASELINE configs[3])
        g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=ns))
        cg = synth.cgrid_geometry(g)
        state, inputs, masks = synth.cgrid_state(g, cg, case=case, seed=20260928, warm=True)
        dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", ns)
        static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
        n_active = int(masks["iceTmask"].sum())
        scal = synth.evp_scalars(ndte)
        d, keep = evp.make_dims(dc, 0)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        try:
            core.cgrid_set_geometry(static)
            core.cgrid_upload(state, inputs, masks)
            for _ in range(warmup):
                core.cgrid_subcycle(ndte)
            core.cgrid_sync()
            t0 = time.perf_counter()
            ev_ms = 0.0
            for _ in range(steps):
                core.cgrid_subcycle(ndte)
                core.cgrid_sync()
                ev_ms += core.cgrid_timings()["loop_ms"]
            wall = time.perf_counter() - t0
            tt_ = core.cgrid_timings()
            one = tt_["one_launch_subcycles"] > 0
            res_n = tt_["resident_subcycles"]
            res_probe_us = 1e3 * tt_["resident_probe_ms"]
            geo = tt_["geometry_derived"]
            marched = tt_["marched_items"]
            out = core.cgrid_download()
        finally:
            core.finalize()
        h = hashlib.sha256()
        for k in CGRID_VERIFY_FIELDS:
            h.update(np.ascontiguousarray(dc.gather({0: out[k]}), dtype="<f8").tobytes())
        want = golden.get(f"cgrid/{workload}/{case}/ndte{ndte}/{ns}/strict", {}).get(str(warmup + steps))
        # kernels per subcycle: cg_res (evp_cgrid_res.hip: every subcycle of a call but the first after an upload inside ONE launch),
        # cg_one, or the fused schedule (evp_cgrid.hip)
        launches = (1.0 / res_n) if res_n else (1 if one else 10 if ns == "tripole" else 3)
        # the resident kernel is not bound by HBM: its share of the chip's fp64 issue slots, from the committed PMC pass of the same kernel
        # on the same grid (tools/profile_gpu.sh cgx1res / cgtx1res: launches of 120 subcycles) over the live time -- withheld unless the
        # pass ran this kernel variant with this many waves
        issue = None
        if res_n:
            pmc, pmc_file = load_pmc()
            e = ((pmc or {}).get("kernels", {}).get({"gx1": "cgx1res", "tx1": "cgtx1res"}.get(workload, ""), {}) or {}).get("resident")
            sq = (e or {}).get("sq") or {}
            busy, waves = sq.get("valu_busy_simd_cycles_per_launch"), sq.get("waves_per_launch")
            live_waves = 4 * tt_["resident_windows_with_ice"]         # (only the windows that hold ice are launched)
            want_kernel = "cg_res<false, false, true>" if ns == "tripole" else "cg_res<false, false, false>"
            if busy and waves == live_waves and want_kernel in (e["kernel_trace"]["name"] or ""):
                t_k = ev_ms * 1e-3 / steps            # one launch = one call of res_n subcycles
                issue = {"bound": "fp64_valu", "frac": busy * (res_n / 120.0) / t_k / (N_SIMD * MAX_CLOCK_HZ), "unit": "share of VALU-busy SIMD-cycles",
                         "valu_busy_simd_cycles_per_subcycle": busy / 120.0, "valu_insts_per_wave_per_subcycle": (sq.get("valu_insts_per_launch") or 0) / waves / 120.0,
                         "pmc_source": f"{pmc_file}#{'cgtx1res' if ns == 'tripole' else 'cgx1res'}",
                         "note": "SQ_ACTIVE_INST_VALU x 4 of the committed pass (launches of 120 subcycles) / live kernel time / (1024 SIMDs x 2.4 GHz); "
                                 "the rest of a subcycle is the hand-off between windows (tools/cgres_phases.py)"}
        one = one or bool(res_n)
        t_sub = ev_ms * 1e-3 / (steps * ndte)
        alg = ((CGRID_B_ALG_ONE_GEO if geo else CGRID_B_ALG_ONE) if one else (CGRID_B_ALG_GEO if geo else CGRID_B_ALG)) * nx * ny
        return {"workload": f"{workload} {nx}x{ny} C-grid EVP ndte={ndte}, case={case}, strict fp64, one GPU",
                "value": nx * ny * ndte * steps / wall, "unit": "cell-updates/s", "steps": steps, "warmup": warmup,
                "us_per_subcycle": 1e6 * t_sub, "us_per_subcycle_wall": 1e6 * wall / (steps * ndte),
                "launches_per_subcycle": launches, "active_T_cells": n_active,
                "kernel": ("cg_res (on-chip resident: all subcycles of a call in one launch" + (", FOLD variant)" if ns == "tripole" else ")") if res_n
                           else "cg_strip (the block's interior marched: one wave per strip of 60 columns x segment of rows) with cg_one's windows along the "
                                "block's edges in the same launch" if (one and marched) else "cg_one" if one
                           else "five phases + five fold steps" if ns == "tripole" else "fused schedule, three launches"),
                "marched_items": marched or None, "marched_segment_rows": (tt_["marched_segment_rows"] if marched else None),
                "marched_cells": (tt_["marched_cells"] if marched else None), "windows_beside_marched": (tt_["marched_edge_windows"] if marched else None),
                "resident_subcycles_per_call": res_n, "resident_probe_us_per_subcycle": (res_probe_us if res_n else None),
                "resident_windows_with_ice": (tt_["resident_windows_with_ice"] if res_n else None), "resident_windows": (tt_["resident_windows"] if res_n else None),
                "verified": (h.hexdigest() == want["sha256"]) if want else None,
                "checked_against": "tests/golden/bench_checksums.json (oracle/evp_oracle.c, pinned to the reference's evp() with grid_ice='C')" if want else None,
                "finite": bool(np.isfinite(out["uvelE"]).all()), "max_abs_uE": float(np.abs(out["uvelE"]).max()),
                "fp64_issue": issue,
                "roofline": {"bound": "hbm", "achieved": alg / t_sub / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / t_sub / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_subcycle": alg,
                             "geometry_derived": bool(geo),
                             "frac_on_408B_yardstick": (CGRID_B_ALG_ONE * nx * ny / t_sub / 1e9 / HBM_PEAK_GBS) if one else None,
                             "note": (("289 B per cell and subcycle = 36 fp64 array touches + one mask byte: the one-launch kernel cg_one with 15 "
                                       "of its 23 static arrays (areas, reciprocal areas, boundary ratios, DminTarea, land masks) derived in "
                                       "the kernel from the eight dx / dy arrays -- identities the host verified bit for bit; 408 B with all "
                                       "51 arrays loaded (frac_on_408B_yardstick)" if (geo and one) else
                                       "408 B per cell and subcycle = 51 fp64 array touches of the one-launch kernel cg_one") +
                                      " (shearU, etax2T and the T-cell stresses stay in LDS between its three levels)" if one else
                                      "648 B per cell and subcycle = 81 fp64 array touches of the three fused kernels") +
                                     " (DESIGN.md section 9); on gx1 the 64 MB working set is Infinity-Cache resident" +
                                     ("; with the on-chip resident kernel the state and operands stay in registers / LDS for the whole call and "
                                      "HBM does not bound it: the figure is the yardstick of what a streaming kernel would have to move" if res_n else "")}}

    def cgrid_per_call(workload, case, ndte):
        """What a C-grid host waits for per evp() call, two ways: its own preparation + cice_evp_hip_cgrid_run (14 state +
        23 input arrays and 4 masks in, 19 arrays out), or the preparation on the device (11 T-grid arrays + the ice
        strength in, the loop's state resident between calls, 14 arrays out).  Caller's arrays page-locked."""
        spec = synth.GRIDS[workload]
        nx, ny = spec["nx"], spec["ny"]
        g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
        cg = synth.cgrid_geometry(g)
        state, inputs, masks = synth.cgrid_state(g, cg, case=case, seed=20260928, warm=True)
        t, st7, prev = synth.cgrid_prep_inputs(g, cg, case=case, seed=20260928)
        dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", "closed")
        static, state, inputs, masks = synth.cgrid_scatter(dc, 0, cg, state, inputs, masks)
        own = lambda a, dt=np.float64: np.array(a, dtype=dt, order="C", copy=True)
        tb = {k: own(dc.scatter(v, 0)) for k, v in t.items()}
        static.update({k: dc.scatter(v, 0, fill=0) for k, v in st7.items()})
        d, keep = evp.make_dims(dc, 0)
        core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(ndte), strict=True), static["dyE"], static["dxN"], static["dxT"],
                          static["dyT"], 1.0 / static["uarea"], static["tarea"], keepalive=keep)
        res = {}
        try:
            core.cgrid_set_geometry(static)
            core.cgrid_set_prep_geometry(static)
            work = {k: (own(state[k]) if k in state else np.zeros(core.shape)) for k in evp.CGRID_FIELDS}
            inp = {k: own(inputs[k]) for k in evp.CGRID_INPUTS}
            mk = {k: own(masks[k], np.int32) for k in evp.CGRID_MASKS}
            core.pin_host(*work.values(), *inp.values(), *tb.values())
            ftab = (evp._f64p * 19)(*[evp._dp(work[k]) for k in evp.CGRID_FIELDS])
            f14 = (evp._f64p * 19)(*([evp._dp(work[k]) for k in evp.CGRID_FIELDS[:14]] + [None] * 5))
            itab = (evp._f64p * 23)(*[evp._dp(inp[k]) for k in evp.CGRID_INPUTS])
            ttab = (evp._f64p * 11)(*[evp._dp(tb[k]) for k in evp.PREP_T])
            s12 = (evp._f64p * 12)(*[evp._dp(work[k]) for k in evp.CGRID_FIELDS[:12]])
            pp = evp.PrepParams(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11, dyn_mass_min=1e-10,
                                ssh_stress_coupled=0)
            L = core.lib

            def host_prepared():
                evp._check(L, L.cice_evp_hip_cgrid_run(ndte, 0, ftab, itab, *[evp._ip(mk[k]) for k in evp.CGRID_MASKS]), "cgrid_run")

            first = [True]

            def device_prepared():
                st = s12 if first[0] else None
                first[0] = False
                evp._check(L, L.cice_evp_hip_cgrid_prep(C.byref(pp), ttab, st, *[evp._ip(mk[k]) for k in evp.CGRID_MASKS]), "cgrid_prep")
                evp._check(L, L.cice_evp_hip_cgrid_prep_finish(evp._dp(inp["strength"]), 0), "cgrid_prep_finish")
                evp._check(L, L.cice_evp_hip_cgrid_subcycle(ndte), "cgrid_subcycle")
                evp._check(L, L.cice_evp_hip_cgrid_download(f14), "cgrid_download")

            for label, fn in (("host_prepared_cgrid_run", host_prepared), ("device_prepared_state_resident", device_prepared)):
                for _ in range(2):
                    fn()
                each = []
                with quiet_interpreter():
                    for _ in range(10):
                        t1 = time.perf_counter()
                        fn()
                        each.append(1e3 * (time.perf_counter() - t1))
                tt = core.cgrid_timings()
                res[label] = dict(ms_per_call=float(np.median(each)), loop_ms=tt["loop_ms"], slowest_of_10=max(each))
                if label.startswith("device"):
                    res[label]["prep_kernels_ms"] = tt["prep_ms"]
        finally:
            core.finalize()
        res["note"] = ("median host wall time per evp()-equivalent call over 10 calls, arrays page-locked: host_prepared = upload of 14 state "
                       "+ 23 input arrays + 4 masks, ndte subcycles, download of 19; device_prepared = 11 T-grid arrays + strength in, "
                       "dyn_prep1/2 and the averages on the device (prep_kernels_ms), the loop's state resident between calls, 14 arrays out "
                       "(the host's own preparation time is NOT in the first figure: the reference spends it on top)")
        return res

    def per_call_cost(workload, case, ndte):
        """What CICE waits for per evp(): cice_evp_hip_run = H2D of 32 fields + loop + D2H of 18, page-locked arrays."""
        spec = synth.GRIDS[workload]
        nx, ny = spec["nx"], spec["ny"]
        g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
        st = synth.make_state(g, case=case, seed=20260928, warm=True)
        dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", "closed")
        geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
        work = {k: np.array(dc.scatter(st[k], 0), dtype=np.float64, order="C", copy=True) for k in evp.FIELDS}
        tmc = np.ascontiguousarray(dc.scatter(st["iceTmask"], 0, fill=0), np.int32)
        umc = np.ascontiguousarray(dc.scatter(st["iceUmask"], 0, fill=0), np.int32)
        d, keep = evp.make_dims(dc, 0)
        core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(ndte), strict=a.strict), geo["HTE"], geo["HTN"], geo["dxT"],
                          geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep)
        res = {}
        try:
            core.pin_host(*work.values())
            # default first: dyn_evp1d_run's contract (the 12 intent(inout) stresses travel in and out at every call)
            for label, resident in (("default_stresses_copied_each_call", 0), ("opt_in_stresses_resident", 1)):
                core.set_option(evp.OPT_STRESS_RESIDENT, resident)
                for _ in range(2):
                    core.run_inplace(work, tmc, umc, ndte)
                n = 10
                each = []
                with quiet_interpreter():
                    for _ in range(n):
                        t1 = time.perf_counter()
                        core.run_inplace(work, tmc, umc, ndte)
                        each.append(1e3 * (time.perf_counter() - t1))
                t = float(np.median(each)) * 1e-3
                if os.environ.get("CICE_EVP_BENCH_DEBUG"):
                    print(f"[bench] per-call {label}: " + " ".join(f"{x:.2f}" for x in each), file=sys.stderr)
                tt = core.timings()
                res[label] = dict(cice_evp_hip_run=1e3 * t, h2d=tt["h2d_ms"], loop=tt["loop_ms"], d2h=tt["d2h_ms"],
                                  slowest_of_10=max(each))
        finally:
            core.finalize()
        res["note"] = ("median host wall time per cice_evp_hip_run call over 10 calls (upload + ndte subcycles + download), caller's arrays page-locked "
                       "once (one gather + one scatter launch per call); default = what an unpatched host gets (stresses current in "
                       "its arrays after every call); opt_in_stresses_resident = for hosts that installed the fetch / invalidate "
                       "hooks (dyn_evp_hip_keep_stresses_resident): 20 fields in, 6 out, the 12 stresses stay on the device")
        return res

    def per_call_option_a(workload, case, ndte):
        """What a host that hands over the whole evp() body waits for per call (INTEGRATION.md Option A,
        dyn_evp_hip_evp_body): cice_evp_hip_prep (11 T-grid arrays + uvel, vvel in; dyn_prep1 / T->U averages / dyn_prep2 on
        the device; the 12 stresses resident) + the host's ice strength via _set_strength + ndte subcycles + download --
        once as the shim does it by default (the 12 stresses in and out with everything else: CICE's arrays current after
        every call), once as it does for a host that opted in to resident stresses (dyn_evp_hip_keep_stresses_resident: they
        stay on the device, 6 outputs back; restart / history through the fetch hook)."""
        spec = synth.GRIDS[workload]
        nx, ny = spec["nx"], spec["ny"]
        g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns="closed"))
        pr = synth.make_primary(g, case, seed=9)
        st = synth.make_state(g, case=case, seed=7, warm=True)
        dc = decomp.per_rank_blocks(nx, ny, 1, "cyclic", "closed")
        sc = lambda x, fill=0.0: np.ascontiguousarray(dc.scatter(np.ascontiguousarray(x), 0, fill=fill))
        geo = {k: sc(g[k], 1.0 if k != "uarear" else 0.0) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
        d, keep = evp.make_dims(dc, 0)
        core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(ndte), strict=a.strict), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                          geo["uarear"], geo["tarea"], keepalive=keep)
        lib = core.lib
        res = {}
        try:
            static = {k: sc(v, (1.0 if k in ("tarea", "uarea") else 0)) for k, v in pr["static"].items()}
            core.set_prep_geometry(*[static[k] for k in ("tmask", "umask", "hm", "tarea", "uarea", "fcor_blk")])
            pp = evp.PrepParams(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11, dyn_mass_min=1e-10,
                                ssh_stress_coupled=0)
            t = {k: sc(v) for k, v in pr["t"].items()}
            state = {k: sc(v) for k, v in pr["state"].items()}
            strength = sc(st["strength"])
            out = {k: np.zeros(core.shape) for k in ("strintxU", "strintyU", "taubxU", "taubyU", "uvel", "vvel")}
            core.pin_host(*t.values(), state["uvel"], state["vvel"], strength, *out.values(), *[state[k] for k in evp.FIELDS[:12]])
            ttab = (evp._f64p * 11)(*[evp._dp(t[k]) for k in evp.PREP_T])
            first = (evp._f64p * len(evp.FIELDS))(*[(evp._dp(state[k]) if k in state and k != "iceUmask" else None) for k in evp.FIELDS])
            later = (evp._f64p * len(evp.FIELDS))(*[(evp._dp(state[k]) if k in ("uvel", "vvel") else None) for k in evp.FIELDS])
            later_sig = (evp._f64p * len(evp.FIELDS))(*[(evp._dp(state[k]) if k in ("uvel", "vvel") or k in evp.FIELDS[:12] else None)
                                                        for k in evp.FIELDS])
            tm = np.zeros(core.shape, np.int32)
            um = np.ascontiguousarray(state["iceUmask"], np.int32)
            o6 = (evp._f64p * len(evp.FIELDS))(*[(evp._dp(out[k]) if k in out else None) for k in evp.FIELDS])
            o18 = (evp._f64p * len(evp.FIELDS))(*[(evp._dp(out[k]) if k in out else evp._dp(state[k]) if k in evp.FIELDS[:12] else None)
                                                  for k in evp.FIELDS])

            def call(tab, otab):
                parts = [time.perf_counter()]
                evp._check(lib, lib.cice_evp_hip_prep(C.byref(pp), ttab, tab, evp._ip(tm), evp._ip(um), None, None, None, None), "prep")
                parts.append(time.perf_counter())
                evp._check(lib, lib.cice_evp_hip_set_strength(evp._dp(strength)), "set_strength")
                parts.append(time.perf_counter())
                evp._check(lib, lib.cice_evp_hip_subcycle(C.c_int32(ndte)), "subcycle")
                evp._check(lib, lib.cice_evp_hip_download(otab), "download")
                parts.append(time.perf_counter())
                return [1e3 * (y - x) for x, y in zip(parts[:-1], parts[1:])]

            call(first, o18)
            for label, itab, otab in (("shim_default_stresses_in_and_out", later_sig, o18),
                                      ("shim_resident_stresses_6_arrays_back", later, o6)):
                call(itab, otab)
                with quiet_interpreter():
                    ts = np.array([call(itab, otab) for _ in range(10)])
                med = np.median(ts, axis=0)
                tt = core.timings()
                res[label] = dict(ms_per_call=float(np.median(ts.sum(axis=1))), prep_call=float(med[0]), set_strength=float(med[1]),
                                  loop_and_download=float(med[2]), slowest_of_10=float(ts.sum(axis=1).max()),
                                  library=dict(h2d=tt["h2d_ms"], prep_kernels=tt["prep_ms"], loop=tt["loop_ms"], d2h=tt["d2h_ms"]))
        finally:
            core.finalize()
        res["note"] = ("median host wall time per evp() body over 10 calls through Option A (device preparation): 13 arrays in (11 T-grid "
                       "fields, uvel, vvel) + strength, dyn_prep1 / averages / dyn_prep2 + ndte subcycles on the device, arrays page-locked; "
                       "the host's own ice-strength computation (Icepack) is NOT in the figure")
        return res

    ndte = a.ndte or {"gx3": 120, "gx1": 120, "s01": 480}[a.workload]
    M = measure_with_fallbacks(a.workload, a.case, ndte, a.steps, a.warmup, median_calls=(10 if (a.gpus == 1 and a.workload != "s01") else 0))
    nx, ny, dc, tm, n_active, dt, tm_ev, kt = (M[k] for k in ("nx", "ny", "dc", "tm", "n_active", "dt", "tm_ev", "kt"))
    # extras must never cost the primary line: a failure is reported inside the JSON instead
    M2 = M3 = MS = None
    extra = {}
    extra_err = {}
    want = (lambda k: a.secondary and k in a.extras)

    def rank_block(Mx, label):
        """One extra workload at N > 1 as the JSON line reports it (rank 0 only uses it)."""
        c = Mx["nx"] * Mx["ny"]
        return {"workload": label, "value": c * Mx["ndte"] * Mx["steps"] / Mx["dt"], "unit": "cell-updates/s",
                "steps": Mx["steps"], "warmup": Mx["warmup"], "ms_per_step": 1e3 * Mx["dt"] / Mx["steps"],
                "us_per_subcycle": 1e6 * Mx["dt"] / (Mx["steps"] * Mx["ndte"]),
                "decomposition": f"{Mx['dc'].proc_shape[0]}x{Mx['dc'].proc_shape[1]} ranks, "
                                 f"{Mx['dc'].block_size_x}x{Mx['dc'].block_size_y} cells each",
                "tile_variant": Mx["tm_ev"]["tile_variant"], "halo_transport": Mx["tm_ev"]["halo_transport"],
                "launches_per_subcycle": Mx["tm_ev"]["launches_per_subcycle"],
                "verified": Mx["ver"].get("verified"), "verification": Mx["ver"], "finite": Mx["finite"],
                "attempts": Mx.get("attempts"), "per_rank": Mx.get("per_rank")}

    M2o = None
    if want("s01") and a.workload != "s01":
        try:      # the 0.1-degree-class grid the strong-scaling target is stated on (streaming kernel, HBM-bound)
            # (N > 1: timed a second time on the same ranks and state with the ring exchange overlapped with the interior of the
            # pass (CICE_EVP_HIP_MARCH_OVERLAP=1: early launch of the (strip, row) units the neighbours wait for, pack + RCCL send /
            # recv on the second stream, every other unit on the compute stream -- a loss on one GPU, meant for real xGMI), and a
            # third time with the ring not through RCCL but as stores into the neighbours' HIP-IPC-mapped inboxes; all three are
            # reported)
            M2 = measure_with_fallbacks("s01", "full", 480, 2, 1,
                                        again_env=([{"CICE_EVP_HIP_MARCH_OVERLAP": "1"}, {"CICE_EVP_HIP_MARCH_DIRECT": "1"}]
                                                   if (world > 1 and "ring_variants" in a.extras) else None))
            M2o = M2.get("again")
        except Exception as e:  # noqa: BLE001
            extra_err["secondary"] = f"{type(e).__name__}: {e}"[:300]
    if want("streaming") and world == 1 and 1000 <= tm_ev["tile_variant"] < 3000:
        try:      # the same workload through the streaming kernel: its HBM fraction next to the resident kernel's
            MS = measure(a.workload, a.case, ndte, 5, 2, env={"CICE_EVP_HIP_RESIDENT": "0"})
        except Exception as e:  # noqa: BLE001
            extra_err["streaming"] = f"{type(e).__name__}: {e}"[:300]
    if want("rccl_control") and world > 1:
        # the control next to the library's default path: the same workload with every remote ghost cell carried by RCCL
        # point-to-point (ncclSend / ncclRecv per neighbour and subcycle, streaming kernel) -- what north_star names
        if rehearsal:
            extra["rccl_control"] = {"skipped": "rehearsal on one GPU: RCCL refuses two ranks per device"}
        else:
            try:
                Mx = measure_with_fallbacks(a.workload, a.case, ndte, max(2, a.steps // 4), 1, env={"CICE_EVP_HIP_HALO": "rccl"})
                extra["rccl_control"] = rank_block(Mx, f"{a.workload} {Mx['nx']}x{Mx['ny']} B-grid EVP ndte={ndte}, case={a.case}, {world} GPUs, "
                                                       "halo forced onto RCCL point-to-point")
            except Exception as e:  # noqa: BLE001
                extra_err["rccl_control"] = f"{type(e).__name__}: {e}"[:300]
    if want("configs2") and a.workload == "gx1" and world > 1:
        # BASELINE configs[2]: gx1 at ndte = 240 over the N GPUs -- once on the library's default path (the on-chip kernel
        # trading tagged records through IPC-mapped buffers when every rank fits), once forced onto what the config names:
        # block-decomposition halo over RCCL point-to-point every subcycle (streaming kernel; the exchange on the second
        # stream where the per-rank domain is large enough to overlap)
        c2 = {}
        for label, env2 in (("library_default", {}), ("rccl_point_to_point_forced", {"CICE_EVP_HIP_HALO": "rccl"})):
            if env2 and rehearsal:
                c2[label] = {"skipped": "rehearsal on one GPU: RCCL refuses two ranks per device"}
                continue
            try:
                Mx = measure_with_fallbacks("gx1", "full", 240, 4, 1, env=env2)
                c2[label] = rank_block(Mx, f"gx1 {Mx['nx']}x{Mx['ny']} B-grid EVP ndte=240, case=full, {world} GPUs")
            except Exception as e:  # noqa: BLE001
                c2[label] = {"error": f"{type(e).__name__}: {e}"[:300]}
        extra["configs2_gx1_ndte240"] = c2
    if want("tripole") and a.workload == "gx1" and world > 1:
        # BASELINE configs[3]: the tripole grid over the N GPUs in its most square cut (8 GPUs: 4 x 2, the fold row split in
        # x: seam partners on different ranks trade their raw records through the peers' buffers, round 4)
        try:
            Mx = measure_with_fallbacks("tx1", "full", 240, 10, 2, ns="tripole")
            extra["tripole"] = rank_block(Mx, f"tx1 {Mx['nx']}x{Mx['ny']} tripole B-grid EVP ndte=240, case=full, {world} GPUs")
        except Exception as e:  # noqa: BLE001
            extra_err["tripole"] = f"{type(e).__name__}: {e}"[:300]
    if want("tripole") and a.workload == "gx1" and world == 1:
        try:      # the tripole grid of configs[3] (fold row averaged inside the resident kernel)
            M3 = measure("tx1", "full", 240, 10, 2, ns="tripole")
        except Exception as e:  # noqa: BLE001
            extra_err["tripole"] = f"{type(e).__name__}: {e}"[:300]
    Mcaps = None
    if want("caps") and a.workload == "gx1" and a.case == "full" and world == 1:
        try:      # SURVEY 8(d)'s second ice case: ice on the polar caps (the realistic cover); only the tiles that hold ice run
            Mcaps = measure("gx1", "caps", 120, 10, 2)
        except Exception as e:  # noqa: BLE001
            extra_err["caps"] = f"{type(e).__name__}: {e}"[:300]
    if want("cgrid") and a.workload == "gx1" and world == 1:
        try:      # next-tier row f-4: the C-grid subcycle on the same grid, and on the 0.1-degree-class one (HBM-bound)
            extra["cgrid"] = cgrid_measure("gx1", "full", 120, 3, 1)
            extra["cgrid"]["s01"] = cgrid_measure("s01", "full", 120, 1, 1)     # (ndte = 120 as the headline: a call's first and last subcycle weigh 1 / 120 each)
            extra["cgrid"]["tx1"] = cgrid_measure("tx1", "full", 120, 3, 1)
            extra["cgrid"]["per_call_ms"] = cgrid_per_call("gx1", "full", 120)
        except Exception as e:  # noqa: BLE001
            extra_err["cgrid"] = f"{type(e).__name__}: {e}"[:300]
    if want("per_call") and a.workload == "gx1" and world == 1:
        try:
            extra["per_call_ms"] = per_call_cost(a.workload, a.case, ndte)
        except Exception as e:  # noqa: BLE001
            extra_err["per_call_ms"] = f"{type(e).__name__}: {e}"[:300]
        try:
            extra.setdefault("per_call_ms", {})["option_a_device_preparation"] = per_call_option_a(a.workload, a.case, ndte)
        except Exception as e:  # noqa: BLE001
            extra_err["per_call_option_a"] = f"{type(e).__name__}: {e}"[:300]

    if rank == 0:
        pmc, pmc_file = load_pmc()

        def hbm_block(Mx, pmc_key):
            """HBM roofline of a streaming-kernel measurement.  One-subcycle kernel: one launch = one subcycle of rank 0's
            sub-domain, 368 B per cell.  Marching kernel (tile_variant >= 3000): one launch = one PASS = four
            subcycles (three or two on short segments; the library reports launches per subcycle); its algorithmic bytes are
            stated per pass -- 27 fp64 reads + 14 writes per cell = 328 B -- and the 368 B-per-cell-subcycle figure stays in
            the block as the labelled yardstick.  Round 6 measured that this kernel is bound by instruction issue, not by HBM
            (one wave per SIMD issues one instruction per ~2.4 ns whatever it is): `issue` prices the committed PMC pass's
            instruction counts at that rate."""
            my = sum(b.gnx * b.gny for b in Mx["dc"].local_blocks(0))
            march = Mx["tm_ev"]["tile_variant"] >= 3000
            if march:
                pmc_key = {"s01str": "s01march"}.get(pmc_key, pmc_key + "march")
            lps = float(Mx["tm_ev"]["launches_per_subcycle"] or 0.0)
            spl = (1.0 / lps if lps > 0 else 4.0) if march else 1      # subcycles per launch (mean over the call's passes)
            tk = Mx["tm_ev"]["marks_ms"] * 1e-3 / (Mx["steps"] * Mx["ndte"] / spl)
            alg = (B_PASS_MARCH if march else B_ALG) * my
            e = (pmc or {}).get("kernels", {}).get(pmc_key) if world == 1 else None
            traffic = e.get("hbm_bytes_per_launch") if e else None
            blk = {"bound": "hbm", "achieved": alg / tk / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": alg / tk / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                   "kernel": "evp_marchk" if march else "evp_subcycle_tile",
                   "kernel_us": 1e6 * tk, "alg_bytes_per_launch": alg, "subcycles_per_launch": spl,
                   "tile_variant": Mx["tm_ev"]["tile_variant"],
                   "launches_per_subcycle": Mx["tm_ev"]["launches_per_subcycle"]}
            if march:
                blk["alg_bytes_per_cell_per_launch"] = B_PASS_MARCH
                blk["yardstick_368B_per_cell_subcycle"] = {
                    "GBps_equivalent": B_ALG * my * spl / tk / 1e9, "frac_of_hbm_peak": B_ALG * my * spl / tk / 1e9 / HBM_PEAK_GBS,
                    "note": "what a kernel that streams every field once per SUBCYCLE would have to move for the same result "
                            "(SURVEY 8d); this kernel streams them once per pass of several subcycles, so the figure may exceed the peak"}
                c = (e or {}).get("counters", {})
                kinds = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
                if all(k in c for k in kinds) and c.get("SQ_WAVES", {}).get("avg_per_launch") and "evp_marchk" in (e.get("kernel") or ""):
                    waves = c["SQ_WAVES"]["avg_per_launch"]
                    n_inst = sum(c[k]["avg_per_launch"] for k in kinds)
                    floor = n_inst / waves * FP64_NS_PER_INST[1] * 1e-9          # every wave alone on its SIMD, all in parallel
                    blk["issue"] = {"bound": "instruction issue (one wave per SIMD)", "wave_instructions_per_launch": n_inst,
                                    "waves_per_launch": waves, "instructions_per_wave": n_inst / waves,
                                    "valu_share": c["SQ_INSTS_VALU"]["avg_per_launch"] / n_inst,
                                    "ns_per_instruction_one_wave_per_simd": FP64_NS_PER_INST[1],
                                    "floor_us_per_launch": 1e6 * floor, "frac": floor / tk,
                                    "live_ns_per_instruction": 1e9 * tk / (n_inst / waves),
                                    "pmc_source": f"{pmc_file}#{pmc_key}",
                                    "note": "instructions of all kinds a wave issues per launch (committed PMC pass of this command) x "
                                            "2.4 ns, the rate tools/fp64_rate_probe.hip measures for one wave per SIMD "
                                            "(profiles/r03_fp64_rate_probe.txt), over the live launch time: what binds this kernel"}
            if traffic:
                blk["measured_traffic_GBps"] = traffic / tk / 1e9
                blk["pmc_source"] = f"{pmc_file}#{pmc_key}"
            try:      # what a plain streaming kernel of the same array shape (30 in, 16 out) reaches on this box, live
                ceil = evp.stream_probe(my) / 1e9
                blk["stream_ceiling"] = {"GBps": ceil, "frac_of_ceiling": alg / tk / 1e9 / ceil,
                                         "note": "cice_evp_hip_stream_probe: 30 fp64 arrays in, 16 out, same cell count, "
                                                 "no arithmetic; the practical HBM rate for this access shape"}
            except Exception as e:  # noqa: BLE001
                blk["stream_ceiling"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            return blk

        cells = nx * ny
        ms_step = 1e3 * dt / a.steps
        value = cells * ndte * a.steps / dt
        my_cells = sum(b.gnx * b.gny for b in dc.local_blocks(0))
        my_active = int((tm[:, 1:-1, 1:-1] != 0).sum())
        resident = 1000 <= tm_ev["tile_variant"] < 3000
        # dominant kernel and its average launch duration from HIP events on the library's
        # stream over the timed region: the streaming kernel is launched once per subcycle
        # (graph-captured), the on-chip resident kernel once per step (all ndte subcycles)
        sub_per_launch = ndte if resident else 1
        n_launch = a.steps * (1 if resident else ndte)
        t_kernel = tm_ev["marks_ms"] * 1e-3 / n_launch
        alg_bytes = B_ALG * my_cells * sub_per_launch
        kshown = "evp_resident2_tile" if resident else "evp_subcycle_tile"
        if resident:
            e = (pmc or {}).get("kernels", {}).get("gx1res") if (a.workload == "gx1" and a.case == "full" and a.strict and world == 1) else None
            same = bool(e and e.get("bench_line_under_trace", {}).get("tile_variant") == tm_ev["tile_variant"])
            # The committed counters may only price THIS launch if they were taken on the same kernel doing the same work:
            # the kernel name, the wave count (tiles x 4 waves, from the tile shape the live run reports) and the VALU
            # instructions per 64 active cells and subcycle (595 in strict mode) are checked against the live run;
            # anything else and the counter-based figures are withheld rather than divided by a live time they do not fit.
            pmc_refused = None
            if e:
                logw = tm_ev["tile_variant"] % 10
                W, H = 1 << logw, 256 >> logw
                # (only the tiles that hold ice run: the library reports how many; older libraries ran them all)
                live_waves = 4 * (int(tm_ev.get("resident_tiles_run") or 0) or
                                  sum((-(-b.gnx // (W - 1))) * (-(-b.gny // (H - 1))) for b in dc.local_blocks(0)))
                waves = e.get("counters", {}).get("SQ_WAVES", {}).get("avg_per_launch")
                insts = e.get("counters", {}).get("SQ_INSTS_VALU", {}).get("avg_per_launch")
                per64 = insts / (my_active / 64.0 * sub_per_launch) if insts else None
                if kshown not in (e.get("kernel") or ""):
                    pmc_refused = f"committed counters are of '{e.get('kernel')}', the timed kernel is {kshown}"
                elif not same:
                    pmc_refused = "committed counters were taken on another tile variant"
                elif tm_ev["tile_variant"] >= 2000 and waves != live_waves:
                    pmc_refused = f"committed pass ran {waves} waves per launch, this run launches {live_waves}"
                elif per64 is None or not (450.0 <= per64 <= 800.0):
                    pmc_refused = f"committed SQ_INSTS_VALU = {per64} per 64 active cells and subcycle (expected ~595-700)"
                if pmc_refused:
                    e = None
            busy = e.get("valu_busy_simd_cycles_per_launch") if e else None
            peak = N_SIMD * MAX_CLOCK_HZ
            flops = ALG_FLOP * my_active * sub_per_launch / t_kernel / 1e12
            roof = {"bound": "fp64_valu",
                    "achieved": (busy / t_kernel) if busy else None, "peak": peak, "unit": "VALU-busy SIMD-cycles/s",
                    "frac": (busy / t_kernel / peak) if busy else None,
                    "traffic": e.get("hbm_bytes_per_launch") if e else None,
                    "kernel": kshown, "kernel_us": 1e6 * t_kernel, "subcycles_per_launch": sub_per_launch,
                    "pmc_source": f"{pmc_file}#gx1res" if e else None, "pmc_same_tile_variant": same, "pmc_refused": pmc_refused,
                    "valu_busy_simd_cycles_per_launch": busy,
                    # the other two readings of the same launch, side by side with `frac` (labelled; neither bounds it)
                    "frac_of_hbm_peak_on_368B_yardstick": alg_bytes / t_kernel / 1e9 / HBM_PEAK_GBS,
                    "frac_of_fp64_flop_peak_without_fma": flops / (FP64_VALU_PEAK_TFLOPS / 2),
                    "nominal_flops": {"flop_per_active_cell_subcycle": ALG_FLOP, "achieved_TFLOPs": flops,
                                      "peak_TFLOPs_without_fma": FP64_VALU_PEAK_TFLOPS / 2, "frac": flops / (FP64_VALU_PEAK_TFLOPS / 2)},
                    "hbm_equivalent": {"alg_bytes_per_launch": alg_bytes, "GBps_equivalent": alg_bytes / t_kernel / 1e9,
                                       "frac_of_hbm_peak": alg_bytes / t_kernel / 1e9 / HBM_PEAK_GBS,
                                       "active_cells_only_GBps": B_ALG * my_active * sub_per_launch / t_kernel / 1e9,
                                       "note": "yardstick only: 368 B x cells x subcycles / time is what a kernel that streams every "
                                               "field once per subcycle would have to move; this kernel keeps the stresses and the "
                                               "per-call operands in registers/LDS for all subcycles of a launch and moves `traffic` "
                                               "bytes instead, so the figure may exceed the HBM peak -- HBM does not bound it"},
                    "measured_fp64_issue_rate": measured_issue(e, t_kernel, sub_per_launch, my_active),
                    "note": "frac = VALU-busy SIMD-cycles per launch (SQ_ACTIVE_INST_VALU x 4 from the committed PMC pass of "
                            "this command, pmc_source) / live kernel time (HIP events on the kernel's stream over the timed "
                            "region) / (1024 SIMDs x 2.4 GHz): the share of the chip's fp64 issue slots the launch used"}
        else:
            roof = hbm_block(M, {"gx1": "gx1str", "s01": "s01str"}.get(a.workload, ""))
            if roof["subcycles_per_launch"] == 1:
                roof["achieved_active_cells_only"] = B_ALG * my_active / t_kernel / 1e9
                roof["note"] = ("achieved = 368 B x grid cells of rank 0 (SURVEY 8d: all cells of the domain, ice or not) / average "
                                "launch duration from HIP events on the kernel's stream over the timed region")
            else:
                roof["note"] = ("achieved = 328 B (27 fp64 reads + 14 writes: every field once per PASS of several subcycles) x grid "
                                "cells of rank 0 / average launch duration from HIP events on the kernel's stream over the "
                                "timed region (includes, per call, one gather and one scatter launch between the CICE block "
                                "layout and the kernel's private layout)")
        streaming = {}
        if MS is not None:
            streaming[a.workload] = hbm_block(MS, {"gx1": "gx1str"}.get(a.workload, ""))
            streaming[a.workload]["verified"] = MS["ver"].get("verified")
            if a.workload in ("gx1", "gx3"):
                streaming[a.workload]["note"] = "working set (45 MB on gx1) is Infinity-Cache resident: fabric requests, not DRAM"
        if M2 is not None:
            streaming["s01"] = hbm_block(M2, "s01str")
        if streaming:
            roof["streaming"] = streaming
        res = {
            "metric": "EVP subcycle cell-updates/sec (gx1 fp64)" if a.workload == "gx1"
                      else f"EVP subcycle cell-updates/sec ({a.workload} fp64)",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "median_ms_per_step": (float(np.median(M["each_ms"])) if M.get("each_ms") else None),
            "verified": M["ver"].get("verified"),
            "config": {"workload": f"{a.workload} {nx}x{ny} B-grid EVP ndte={ndte}, case={a.case}, "
                                   f"{'strict fp64 (no FMA contraction; bit-identical to the reference)' if a.strict else 'fp64 with FMA contraction'}",
                       "cells": cells, "active_T_cells": n_active, "ndte": ndte,
                       "decomposition": f"{dc.proc_shape[0]}x{dc.proc_shape[1]} ranks, "
                                        f"{dc.block_size_x}x{dc.block_size_y} cells each",
                       "us_per_subcycle": 1e3 * ms_step / ndte, "tile_variant": tm_ev["tile_variant"],
                       "launches_per_subcycle": tm_ev["launches_per_subcycle"],
                       "halo_transport": tm_ev["halo_transport"],
                       "autotune_probe_us": {"streaming": 1e3 * tm_ev["stream_probe_ms"], "resident": 1e3 * tm_ev["resident_probe_ms"]},
                       "resident_fallbacks": M["fallbacks"], "attempts": M.get("attempts"), "per_rank": M.get("per_rank"),
                       "finite": M["finite"], "max_abs_u": M["umax"]},      # (scalars of the other workloads are added below)
            "verification": M["ver"],
            "roofline": roof,
        }
        if M2 is not None:
            c2 = M2["nx"] * M2["ny"]
            res["secondary"] = {
                "workload": f"s01 {M2['nx']}x{M2['ny']} B-grid EVP ndte=480, case=full (0.1-degree class), strong scaling",
                "value": c2 * 480 * 2 / M2["dt"], "unit": "cell-updates/s", "steps": 2, "warmup": 1,
                "ms_per_step": 1e3 * M2["dt"] / 2, "us_per_subcycle": 1e6 * M2["dt"] / (2 * 480),
                "decomposition": f"{M2['dc'].proc_shape[0]}x{M2['dc'].proc_shape[1]} ranks, "
                                 f"{M2['dc'].block_size_x}x{M2['dc'].block_size_y} cells each",
                "tile_variant": M2["tm_ev"]["tile_variant"], "halo_transport": M2["tm_ev"]["halo_transport"],
                "launches_per_subcycle": M2["tm_ev"]["launches_per_subcycle"],
                "roofline_frac_rank0": streaming["s01"]["frac"],
                "verified": M2["ver"].get("verified"), "finite": M2["finite"],
                "attempts": M2.get("attempts"), "per_rank": M2.get("per_rank")}
            for Mo in (M2o or []):
                direct = "CICE_EVP_HIP_MARCH_DIRECT" in Mo["env"]
                res["secondary"]["ring_exchange_direct_ipc" if direct else "ring_exchange_overlapped"] = {
                    "value": c2 * 480 * 2 / Mo["dt"], "us_per_subcycle": 1e6 * Mo["dt"] / (2 * 480), "finite": Mo["finite"],
                    "ring": Mo.get("ring"),
                    "verified": None, "why_unverified": "continues from the verified run's state (no checksum that far); bit-identity of this "
                                                        "form of the exchange is what tests/test_gpu_march.py and the multi-process tests pin",
                    "note": ("CICE_EVP_HIP_MARCH_DIRECT=1: no RCCL -- the pack kernel stores into the neighbours' HIP-IPC-mapped inboxes, "
                             "flags instead of send / recv (off by default: no faster where the transfer is a device copy, never measured "
                             "over xGMI); 'ring' says whether the trial exchange let it be used") if direct else
                            ("CICE_EVP_HIP_MARCH_OVERLAP=1: the (strip, row) units other ranks wait for advanced first on the second stream, "
                             "pack + RCCL send / recv overlapped with the rest of the pass (off by default: a loss where the transfer is "
                             "a device copy -- profiles/r06_ring_rank.txt)")}
        for k_, v_ in extra_err.items():
            res[k_] = {"error": v_}
        res.update(extra)
        # the other workloads' headline scalars once more, flat, inside `config`: a reader that keeps only the contract's keys
        # (the driver's record does) still sees them
        if M2 is not None:
            res["config"].update(s01_us_per_subcycle=res["secondary"]["us_per_subcycle"], s01_verified=res["secondary"]["verified"],
                                 s01_subcycles_per_launch=(1.0 / M2["tm_ev"]["launches_per_subcycle"] if M2["tm_ev"]["launches_per_subcycle"] else None),
                                 s01_frac_hbm_peak_on_328B_per_pass=streaming["s01"]["frac"],
                                 s01_frac_issue_floor=(streaming["s01"].get("issue") or {}).get("frac"))
        cg = res.get("cgrid") if isinstance(res.get("cgrid"), dict) else {}
        for key in ("gx1", "tx1", "s01"):
            v = cg if key == "gx1" else (cg.get(key) if isinstance(cg.get(key), dict) else None)      # (gx1's figures are the block itself)
            if v and v.get("us_per_subcycle") is not None:
                res["config"][f"cgrid_{key}_us_per_subcycle"] = v["us_per_subcycle"]
                res["config"][f"cgrid_{key}_verified"] = v.get("verified")
        if M3 is not None:
            res["tripole"] = {
                "workload": "tx1 360x240 tripole B-grid EVP ndte=240, case=full, one GPU",
                "value": M3["nx"] * M3["ny"] * 240 * 10 / M3["dt"], "unit": "cell-updates/s", "steps": 10, "warmup": 2,
                "us_per_subcycle": 1e6 * M3["dt"] / (10 * 240), "tile_variant": M3["tm_ev"]["tile_variant"],
                "verified": M3["ver"].get("verified"), "finite": M3["finite"]}
        if Mcaps is not None:
            tcaps = Mcaps["dt"] / (10 * 120)
            res["caps"] = {
                "workload": "gx1 320x384 B-grid EVP ndte=120, case=caps (ice on the polar caps: SURVEY 8(d)'s realistic cover), one GPU",
                "value": Mcaps["nx"] * Mcaps["ny"] / tcaps, "unit": "cell-updates/s",
                "active_cell_updates_per_s": Mcaps["n_active"] / tcaps, "active_T_cells": Mcaps["n_active"],
                "steps": 10, "warmup": 2, "us_per_subcycle": 1e6 * tcaps, "tile_variant": Mcaps["tm_ev"]["tile_variant"],
                "resident_tiles_run": Mcaps["tm_ev"].get("resident_tiles_run"), "resident_tiles": Mcaps["tm_ev"].get("resident_tiles"),
                "verified": Mcaps["ver"].get("verified"), "finite": Mcaps["finite"],
                "note": "the on-chip resident kernel launches only the tiles that hold ice"}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.workload, a.case, ndte, a.cpu_seconds, a.strict)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
