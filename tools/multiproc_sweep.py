#!/usr/bin/env python3
"""Several ranks as several processes on ONE GPU (tools/mailbox_2proc.py) over random geometries: domain size, rank layout,
blocks per rank, kernel / path (resident with remote neighbours, streaming + mailbox, two subcycles per pass through the
test transport, C grid, device preparation).  Every rank's sub-domain, ghost cells included, must equal the single-rank
run bit for bit.  Not part of the test suite (each case spawns 2-4 processes): run by hand, gpurun_out/ keeps failures.

  usage: tools/multiproc_sweep.py [--seeds 1 2 3 ...] [--first 1 --count 12]"""
import argparse
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="*", default=None)
    ap.add_argument("--first", type=int, default=1)
    ap.add_argument("--count", type=int, default=12)
    ap.add_argument("--tripole-resident", action="store_true",
                    help="every case: the on-chip kernel on a tripole grid, the fold row split over 1 to 4 ranks in x (round 4)")
    ap.add_argument("--tripoleT-next", action="store_true",
                    help="every case: a tripoleT grid cut in y only (fold rows on the top rank) -- device preparation, C-grid loop, "
                         "C-grid device preparation (end of round 4)")
    a = ap.parse_args()
    seeds = a.seeds if a.seeds else list(range(a.first, a.first + a.count))
    nbad = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        mode = ["resident", "streaming", "march", "cgrid", "prep", "cgrid_prep"][seed % 6]
        trip = mode in ("resident", "streaming", "prep") and seed % 4 == 0
        world, shape = [(2, "2x1"), (2, "1x2"), (4, "2x2")][int(rng.integers(0, 3))]
        if a.tripole_resident:
            mode, trip = ("resident" if seed % 3 else "prep"), True
            world, shape = [(2, "2x1"), (3, "3x1"), (4, "4x1"), (4, "2x2"), (2, "1x2"), (6, "3x2")][int(rng.integers(0, 6))]
        if a.tripoleT_next:
            mode, trip = ["prep", "cgrid", "cgrid_prep"][seed % 3], False
            world, shape = [(2, "1x2"), (3, "1x3"), (4, "1x4")][int(rng.integers(0, 3))]
        px, py = [int(v) for v in shape.split("x")]
        nx = 2 * px * int(rng.integers(14, 50))      # even, and the same number of columns on every rank
        ny = py * int(rng.integers(16, 60))
        tfold = (not a.tripole_resident) and mode == "streaming" and seed % 4 == 1      # ns_boundary_type = 'tripoleT' (late round 4)
        tfold = tfold or a.tripoleT_next
        wl = f"{nx}x{ny}" + (":tripole" if trip else (":tripoleT" if tfold else ""))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port()), str(ROOT / "tools" / "mailbox_2proc.py"), "--workload", wl, "--shape", shape,
               "--ndte", str(int(rng.choice([6, 9, 24])))]
        env = dict(os.environ, CICE_EVP_HIP_HALO_TIMEOUT_MS="20000")
        if rng.random() < 0.4 and mode in ("resident", "streaming", "cgrid"):
            cmd += ["--blocks-per-rank", "2x2" if rng.random() < 0.5 else "2x1"]
        if mode == "resident":
            cmd += ["--expect-resident"]
        elif mode == "streaming":
            env["CICE_EVP_HIP_RESIDENT"] = "0"
        elif mode == "march":
            cmd += ["--march"]
            env.update(CICE_EVP_HIP_MARCH="1", CICE_EVP_HIP_RESIDENT="0", CICE_EVP_HIP_MARCH_SEG=str(int(rng.integers(5, 30))))
            env["CICE_EVP_HIP_MARCH_EXT"] = str(int(rng.choice([0, 2, 4])))
            env["CICE_EVP_HIP_MARCH_OWN"] = str(int(rng.choice([13, 29, 60])))
            form = int(rng.integers(0, 3))       # the ring through the transport / overlapped with the pass / as stores into the peers' inboxes
            env["CICE_EVP_HIP_MARCH_OVERLAP"] = "1" if form == 1 else "0"
            env["CICE_EVP_HIP_MARCH_DIRECT"] = "1" if form == 2 else "0"
        elif mode == "cgrid":
            cmd += ["--cgrid"] + (["--visc", "avg_strength"] if rng.random() < 0.4 else []) + (["--maskhalo"] if rng.random() < 0.4 else [])
        elif mode == "prep":
            cmd += ["--prep"]
            if rng.random() < 0.5 or tfold:         # (tripoleT: the streaming kernel is the only one eligible anyway)
                env["CICE_EVP_HIP_RESIDENT"] = "0"
        elif mode == "cgrid_prep":
            cmd += ["--cgrid", "--prep"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        ok = r.returncode == 0 and "MAILBOX_2PROC OK" in r.stdout
        print(f"seed {seed}: {mode} {wl} as {shape} {' '.join(cmd[cmd.index('--ndte'):])} "
              f"{ {k: v for k, v in env.items() if k.startswith('CICE_EVP_HIP_MARCH') or k == 'CICE_EVP_HIP_RESIDENT'} }: {'OK' if ok else 'FAILED'}", flush=True)
        if not ok:
            nbad += 1
            out = ROOT / "gpurun_out" / "multiproc_sweep"
            out.mkdir(parents=True, exist_ok=True)
            (out / f"seed{seed}.log").write_text(" ".join(cmd) + "\n" + r.stdout[-6000:] + "\n---\n" + r.stderr[-6000:])
            print("   ", (r.stdout.strip().splitlines() or [""])[-1][:300], "|", (r.stderr.strip().splitlines() or [""])[-1][:300], flush=True)
    print(f"MULTIPROC_SWEEP {len(seeds) - nbad}/{len(seeds)} OK")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
