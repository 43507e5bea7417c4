"""GPU (-m gpu), run last: several ranks as several PROCESSES on the one GPU of the test box.

The stand-in for a multi-GPU node that a 1-GPU box allows: HIP IPC handles exchanged over gloo,
every rank's kernels trading records / mailbox stores with the other processes' kernels through
IPC-mapped memory (tools/mailbox_2proc.py), and bench.py's N>1 path (CICE_EVP_BENCH_REHEARSAL=1).
Kept in a file of their own, sorted after test_gpu_parity.py, so that under `pytest -x` a hiccup of
this time-sliced set-up cannot hide the results of the single-process parity tests."""
import os
from pathlib import Path

import pytest

from common import FULL_SUITE, wide

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world,workload,shape,resident", wide([(8, "gx3", "2x4", True), (4, "tx1", "4x1", True), (4, "tx1", "2x2", False),
                                                                (2, "tx1", "2x1", "prep_stream"), (2, "360x240:tripoleT", "1x2", False),
                                                                (4, "100x116:tripoleT", "4x1", False), (4, "gx3", "2x2", "prep")]) + [
                                                           (2, "gx3", "", True), (4, "gx3", "2x2", True),
                                                           (2, "gx1", "1x2", False),
                                                           (2, "tx1", "1x2", True),
                                                           (2, "gx3", "1x2", "blocks"),
                                                           # fold row split in x: the on-chip kernel, seam partners on different
                                                           # ranks trading their raw records through the peers' buffers (round 4)
                                                           (2, "tx1", "2x1", True), (4, "tx1", "2x2", True),
                                                           # ranks of the fold row that hold neither a pole point nor a whole pair
                                                           # (3 x 1, 4 x 1: found two ranks running without any fold handling), and
                                                           # the natural cut of tx1 on eight GPUs
                                                           (3, "tx1", "3x1", True), (8, "tx1", "4x2", True),
                                                           (2, "tx1", "2x1", False),
                                                           # evp()'s preparation phase on a tripole grid cut in y
                                                           (2, "tx1", "1x2", "prep"),
                                                           # ... and with the fold row split in x: T-grid ghost cells across the
                                                           # fold and the stress symmetrisation through shifted copies
                                                           (4, "tx1", "2x2", "prep_stream"),
                                                           # ... and the same preparation followed by the on-chip kernel
                                                           (2, "tx1", "2x1", "prep"),
                                                           # ns_boundary_type = 'tripoleT' over several ranks (late round 4): the top
                                                           # physical row is an image of row NY-1 -- receive lists name interior
                                                           # cells, the exchange follows the launch (streaming kernel): the fold
                                                           # row cut in x, in y only, both, odd sizes
                                                           (2, "360x240:tripoleT", "2x1", False),
                                                           (4, "360x240:tripoleT", "2x2", False), (3, "126x60:tripoleT", "3x1", False),
                                                           # round 6: the T-fold INSIDE the on-chip kernel where the top row lies on one
                                                           # rank (cut in y); a top row split in x leaves some ranks unable -- the
                                                           # ranks agree on the streaming kernel ("auto": nothing forced)
                                                           (2, "360x240:tripoleT", "1x2", True), (4, "360x240:tripoleT", "2x2", "auto"),
                                                           # ... and its preparation phase on the device, ranks cut in y (end of
                                                           # round 4: the T-fold rule of the cell-centre fields stays on the top rank)
                                                           (2, "120x80:tripoleT", "1x2", "prep_stream")])
def test_mailbox_halo_between_processes_on_one_gpu(world, workload, shape, resident):
    """The mailbox transport across PROCESS boundaries (HIP IPC handles exchanged over gloo,
    peers' inboxes mapped, flags raised from the other process's kernels): `world` ranks share
    the one GPU of this box, each owning one block; every rank's sub-domain, ghost cells
    included, equals the single-rank run bit for bit (tools/mailbox_2proc.py).  The last two cases cut the
    tripole seam row in x (px = 2): seam pairs and the seam row's east-west neighbours on different ranks, raw
    partner values travelling into the staging slots, every seam cell finalised after the exchange."""
    import subprocess
    import sys as _sys
    root = Path(__file__).resolve().parents[1]
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(root / "tools" / "mailbox_2proc.py"), "--workload", workload, "--ndte", "24"]
    if shape:
        cmd += ["--shape", shape]
    # gx3 pieces fit the one GPU several times over: the resident kernels of all ranks are
    # co-resident and trade tagged records across process boundaries; gx1 halves do not fit
    # twice, so that case pins the streaming kernel + mailbox exchange
    env = dict(os.environ, CICE_EVP_HIP_HALO_TIMEOUT_MS="20000")
    if resident == "blocks":
        # several CICE blocks per rank AND neighbours on other ranks, resident kernel
        cmd += ["--blocks-per-rank", "2x2", "--expect-resident", "--timing"]
    elif resident == "prep_stream":
        cmd += ["--prep"]
        env["CICE_EVP_HIP_RESIDENT"] = "0"
    elif resident == "prep":
        # from the primary model state: evp()'s preparation phase on every rank, its T-grid halos
        # crossing the ranks through the same transport, then the loop (f-2 on a split domain)
        cmd += ["--prep", "--expect-resident"]
    elif resident == "auto":
        pass                                        # the library's own (collective) choice
    elif resident:
        cmd += ["--expect-resident", "--timing"]    # --timing: 5 x 120 + 7 more subcycles, launches back to back
    else:
        env["CICE_EVP_HIP_RESIDENT"] = "0"
    # Several processes time-slicing ONE GPU is a stand-in for several GPUs, not a supported
    # configuration: a run can trip over the previous test's processes still being torn down.  One
    # retry (with a fresh rendezvous port); every failure is kept under gpurun_out/ for inspection.
    for attempt in (1, 2):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        if r.returncode == 0 and "MAILBOX_2PROC OK" in r.stdout:
            break
        try:
            (root / "gpurun_out").mkdir(exist_ok=True)
            (root / "gpurun_out" / f"mailbox_fail_{world}_{workload}_{shape}_{attempt}.log").write_text(r.stdout + "\n---\n" + r.stderr)
        except OSError:
            pass
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
    assert r.returncode == 0 and "MAILBOX_2PROC OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_multi_rank_falls_back_together():
    """bench.py at N > 1 when the preferred path fails at RUN time (test hook: one workgroup of the resident kernel
    never shows up, so its neighbours' waits give up on both ranks): every sync point is an agreement, all ranks
    leave the attempt together, retry with the streaming kernel and report what was tried."""
    import json
    import subprocess
    import sys as _sys
    root = Path(__file__).resolve().parents[1]
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(root / "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "gx3", "--no-secondary"]
    env = dict(os.environ, CICE_EVP_BENCH_REHEARSAL="1", CICE_EVP_HIP_HALO_TIMEOUT_MS="3000", CICE_EVP_HIP_RES_DEBUG="16")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    att = d["config"]["attempts"]
    assert len(att) == 2 and not att[0]["ok"] and att[1]["ok"], att
    assert d["config"]["tile_variant"] < 1000 and d["verified"] is True and d["config"]["finite"]


def test_bench_self_launches_when_started_bare():
    """The literal command the driver may type, no launcher around it: `python bench.py --gpus 2 --steps 2 --warmup 1`.  bench.py
    re-executes itself under torch.distributed.run (one rank per GPU; here both ranks on the one GPU, CICE_EVP_BENCH_REHEARSAL=1)
    and rank 0 prints exactly one JSON line: the gx1 N-rank figure on top, the 3600x2400 grid under `secondary` with the ranks'
    own views, the forced-RCCL control named (it cannot run with two ranks on one device), no CPU-baseline leg."""
    import json
    import subprocess
    import sys as _sys
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CICE_EVP_BENCH_REHEARSAL="1", CICE_EVP_HIP_HALO_TIMEOUT_MS="20000")
    r = subprocess.run([_sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=root,
                       capture_output=True, text=True, timeout=1500, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["verified"] is True
    assert d["metric"].startswith("EVP subcycle cell-updates/sec (gx1") and d["cpu_baseline"] is None
    assert [q["rank"] for q in d["config"]["per_rank"]] == [0, 1]
    assert all(q["halo_transport"] in ("mailbox", "rccl") for q in d["config"]["per_rank"])
    sec = d["secondary"]
    assert sec["verified"] is True and sec["finite"] and sec["value"] > 0 and [q["rank"] for q in sec["per_rank"]] == [0, 1]
    assert "ring_exchange_overlapped" not in sec and "configs2_gx1_ndte240" not in d and "tripole" not in d
    assert "skipped" in d["rccl_control"]


def test_bench_multi_rank_rehearsal():
    """bench.py's N>1 path (decomposition, collective set-up, barrier + MAX-over-ranks timing, the
    JSON line) rehearsed on the one GPU of this box: 2 ranks as 2 processes over gloo with the
    mailbox halo bootstrapped by hand (CICE_EVP_BENCH_REHEARSAL=1; the driver's real N>1 runs use
    RCCL and one GPU per rank).  The line of an N > 1 run must speak for BASELINE configs[2] (gx1 at
    ndte = 240: library default and forced RCCL point-to-point) and configs[3] (tx1, tripole) too."""
    import json
    import subprocess
    import sys as _sys
    root = Path(__file__).resolve().parents[1]
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(root / "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "gx1", "--extras", "configs2,tripole,s01" + (",ring_variants" if FULL_SUITE else "")]
    env = dict(os.environ, CICE_EVP_BENCH_REHEARSAL="1", CICE_EVP_HIP_HALO_TIMEOUT_MS="20000")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["halo_transport"] == "mailbox" and d["config"]["finite"] and d["verified"] is True
    # (gx1 as two processes on ONE GPU: 2 x 286 tiles exceed the 512 co-resident workgroups of the remote variant, so the
    # collective probe sends both ranks to the streaming kernel + mailbox here; on two GPUs each rank has its chip to itself.
    # The tripole block below fits and must run the on-chip kernel.)
    assert d["config"]["tile_variant"] >= 2000 or d["config"]["tile_variant"] < 1000
    assert d["cpu_baseline"] is None and "roofline" in d
    # every rank's own view of the timed region (what a first run on real xGMI is read with)
    pr = d["config"]["per_rank"]
    assert [q["rank"] for q in pr] == [0, 1] and all(q["halo_transport"] == "mailbox" and q["halo_send_cells"] > 0 for q in pr)
    assert all(q["stream_ms"] > 0 and q["wall_ms"] >= 0.5 * q["stream_ms"] and q["local_cells"] > 0 for q in pr)
    # what RCCL itself reports per rank and the device's PCI bus id (the rehearsal has no communicator: -1, but the fields and a
    # bus id are there; on a node rccl_nranks == N and N different bus ids are what a reader checks)
    assert all({"rccl_nranks", "rccl_rank", "rccl_device", "hip_device", "device_bus_id"} <= set(q) for q in pr), pr
    assert all(q["rccl_nranks"] == -1 and len(q["device_bus_id"]) >= 7 for q in pr), pr
    # configs[2]: gx1 at ndte = 240, verified against its own committed checksum; the forced-RCCL leg cannot run here
    c2 = d["configs2_gx1_ndte240"]
    lib = c2["library_default"]
    assert lib["verified"] is True and lib["finite"] and lib["us_per_subcycle"] > 0, lib
    assert lib["verification"]["key"] == "gx1/full/ndte240/closed/strict" and [q["rank"] for q in lib["per_rank"]] == [0, 1]
    assert "skipped" in c2["rccl_point_to_point_forced"]
    # configs[4]: 3600 x 2400 on the marching kernel (ring over the test transport), verified; and once more on the same
    # state with the exchange overlapped with the pass
    sec = d["secondary"]
    assert sec["verified"] is True and sec["finite"] and sec["tile_variant"] >= 3000, sec
    if FULL_SUITE:      # (the other two forms of the ring on the same state; in the default run tools/mailbox_2proc.py --march pins them)
        assert sec["ring_exchange_overlapped"]["finite"] and sec["ring_exchange_overlapped"]["us_per_subcycle"] > 0
        dx = sec["ring_exchange_direct_ipc"]
        assert dx["finite"] and dx["us_per_subcycle"] > 0 and dx["ring"] == "direct stores (HIP IPC)", dx
    # configs[3]: the tripole grid in its natural (most square) cut -- here 2 x 1, the fold row split in x --, the on-chip kernel
    # on both ranks, seam partners trading raw records across the rank boundary
    tp = d["tripole"]
    assert tp["verified"] is True and tp["finite"] and tp["decomposition"].startswith("2x1 ranks"), tp     # the fold row split in x
    assert tp["tile_variant"] >= 2000 and [q["rank"] for q in tp["per_rank"]] == [0, 1]


@pytest.mark.parametrize("world,workload,shape,extra", wide([(2, "gx3", "", []), (2, "gx1", "1x2", ["--maskhalo", "--case", "caps"]),
                                                             (2, "gx3", "2x1", ["--prep", "--case", "caps"]), (2, "120x80:tripoleT", "1x2", [])]) + [
                                                        (4, "gx3", "2x2", ["--blocks-per-rank", "2x2", "--timing"]),
                                                        (2, "gx1", "1x2", ["--visc", "avg_strength"]),
                                                        (4, "gx1", "2x2", ["--timing"]),
                                                        # maskhalo_dyn for the C-grid loop: every in-loop exchange through the
                                                        # masked halo (five-point dilation of iceTmask); ice in two caps only
                                                        (4, "gx3", "2x2", ["--maskhalo", "--case", "caps", "--blocks-per-rank", "2x2"]),
                                                        # tripole grid cut in y: the rank with the fold rows does the
                                                        # fold steps, every rank the five-phase schedule
                                                        (2, "tx1", "1x2", []), (3, "tx1", "1x3", ["--blocks-per-rank", "2x1"]),
                                                        # the preparation phase on the device on every rank (T-grid halos and
                                                        # the E / N velocity averages across ranks), then the loop
                                                        (4, "gx1", "2x2", ["--prep", "--blocks-per-rank", "2x1"]),
                                                        (2, "tx1", "1x2", ["--prep"]),
                                                        # tripoleT, ranks cut in y (end of round 4): T-fold lists on the top rank
                                                        (2, "120x80:tripoleT", "1x2", ["--prep"])])
def test_cgrid_across_processes_on_one_gpu(world, workload, shape, extra):
    """The C-grid subcycle split over `world` ranks (processes sharing this box's GPU): ghost cells that mirror
    cells of other ranks are filled through the mailbox transport after every producing launch -- five exchange
    points per subcycle in the fused schedule, seven in the five-phase one (avg_strength) -- and every rank's
    arrays, ghost cells included, equal the single-rank run bit for bit (tools/mailbox_2proc.py --cgrid)."""
    import subprocess
    import sys as _sys
    root = Path(__file__).resolve().parents[1]
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(root / "tools" / "mailbox_2proc.py"), "--cgrid", "--workload", workload, "--ndte", "24"] + extra
    if shape:
        cmd += ["--shape", shape]
    env = dict(os.environ, CICE_EVP_HIP_HALO_TIMEOUT_MS="20000")
    for attempt in (1, 2):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        if r.returncode == 0 and "MAILBOX_2PROC OK" in r.stdout:
            break
        try:
            (root / "gpurun_out").mkdir(exist_ok=True)
            (root / "gpurun_out" / f"cgrid_mp_fail_{world}_{workload}_{shape}_{attempt}.log").write_text(r.stdout + "\n---\n" + r.stderr)
        except OSError:
            pass
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
    assert r.returncode == 0 and "MAILBOX_2PROC OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("world,workload,shape,extra,ext", wide([(4, "gx1", "4x1", [], "0"), (2, "gx3", "2x1", [], "6")]) + [
                                                            (2, "gx1", "2x1", [], "4"), (2, "gx1", "1x2", ["--timing"], "2"),
                                                            (4, "gx1", "2x2", ["--blocks-per-rank", "2x1"], "4"),
                                                            # pieces narrower than one strip, an odd split, a wide rim
                                                            (3, "gx3", "3x1", ["--timing"], "2"),
                                                            (4, "gx3", "2x2", [], "8")])
def test_two_subcycle_kernel_across_processes_on_one_gpu(world, workload, shape, extra, ext):
    """The marching path (several subcycles per pass) in its several-rank form, as `world` processes sharing this box's GPU: every
    rank plans from the global block table (rectangles of all ranks, ring lists in one canonical order), holds its piece plus a
    redundant rim, exchanges the four-cell ring every (ext + 4)-th subcycle and agrees with the others on path and verdicts.  The
    exchanges and agreements go through the library's test transport (host buffers + torch.distributed gloo: RCCL refuses
    two ranks per device) -- plan, pack / unpack, schedule and kernels are the product's.  Every rank's velocities,
    stresses and diagnostics, ghost cells of the velocities included, equal the single-rank run bit for bit; --timing
    adds back-to-back calls and an odd subcycle count (one subcycle of the one-subcycle kernel with the mailbox halo)."""
    import subprocess
    import sys as _sys
    root = Path(__file__).resolve().parents[1]
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(root / "tools" / "mailbox_2proc.py"), "--march", "--workload", workload, "--ndte", "24", "--shape", shape] + extra
    env = dict(os.environ, CICE_EVP_HIP_HALO_TIMEOUT_MS="20000", CICE_EVP_HIP_MARCH="1", CICE_EVP_HIP_RESIDENT="0",
               CICE_EVP_HIP_MARCH_SEG="24", CICE_EVP_HIP_MARCH_EXT=ext,
               # four, three or two subcycles per pass (every rank the same: the ring is exchanged between the passes)
               CICE_EVP_HIP_MARCH_K=str(2 + int(ext) // 2 % 3),
               # every other layout with the exchange overlapped (early launch of the cells the neighbours wait for: N-S and
               # E-W cuts, corners, several blocks per rank)
               CICE_EVP_HIP_MARCH_OVERLAP=str(int(ext) // 2 % 2 if shape != "2x2" else 1),
               # ... and the layouts that are not overlapped without a library: the pack kernel stores into the other PROCESS's inbox
               # (HIP IPC), flags instead of send / recv, after a trial exchange that must agree with the transport's bits
               CICE_EVP_HIP_MARCH_DIRECT=str(1 - (int(ext) // 2 % 2 if shape != "2x2" else 1)))
    for attempt in (1, 2):          # processes time-slicing ONE GPU: one retry with a fresh rendezvous port, as in the tests above
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        if r.returncode == 0 and "MAILBOX_2PROC OK" in r.stdout:
            break
        try:
            (root / "gpurun_out").mkdir(exist_ok=True)
            (root / "gpurun_out" / f"march_mp_fail_{world}_{shape}_{attempt}.log").write_text(r.stdout + "\n---\n" + r.stderr)
        except OSError:
            pass
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
    assert r.returncode == 0 and "MAILBOX_2PROC OK" in r.stdout, (r.stdout[-2500:], r.stderr[-3000:])
