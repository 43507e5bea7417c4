#!/usr/bin/env python3
"""Predicted strong-scaling curve from ONE GPU: the per-rank piece an N-way split of a workload produces is
timed alone on the GPU with every ghost copy that would cross ranks routed through the remote-halo transport to
the rank itself (CICE_EVP_HIP_SELF_EXCHANGE=1).  What a 1-GPU box cannot show is the xGMI hop itself; the table is
the prediction the first real N > 1 run is to be held against.

The tripole grid tx1 is NOT emulated that way (round 6): with one rank standing in for all, every ghost copy of the piece goes
through the self-exchange and the piece drops to the streaming kernel + three launches per subcycle -- 14-16 us per subcycle, a
figure no real run shows.  Its N > 1 rows are real multi-process runs on the one GPU instead: N ranks as N processes in the
natural cut (2x1, 2x2, 4x2), the on-chip resident kernel across processes, records and mailbox over HIP IPC
(tools/mailbox_2proc.py --timing; every rank bit-identical to the single-rank run or the row is an error).  The processes share
the GPU, so the row says what the hand-offs between ranks cost, not what eight chips would gain.

  python tools/predict_scaling.py [out.json [workload ...]]
"""
import json
import os
import sys
import time
from pathlib import Path

R = str(Path(__file__).resolve().parents[1])
sys.path[:0] = [R, R + "/tests", R + "/oracle"]
import numpy as np
from cice_amd import decomp, evp, synth

# workload: (nx, ny, ns, ndte, {N: [(px, py), ...]})
WORK = {
    "gx1": (320, 384, "closed", 120, {1: [(1, 1)], 2: [(2, 1), (1, 2)], 4: [(4, 1), (2, 2)], 8: [(8, 1), (4, 2)]}),
    "tx1": (360, 240, "tripole", 240, {1: [(1, 1)], 2: [(2, 1)], 4: [(2, 2)], 8: [(4, 2)]}),       # N > 1: real processes, see above
    "s01": (3600, 2400, "closed", 480, {1: [(1, 1)], 2: [(2, 1)], 4: [(4, 1), (2, 2)], 8: [(8, 1), (4, 2)]}),
}
TRANSPORTS = {
    "resident-remote": {"CICE_EVP_HIP_HALO": "direct"},
    "stream+mailbox": {"CICE_EVP_HIP_HALO": "direct", "CICE_EVP_HIP_RESIDENT": "0", "CICE_EVP_HIP_MARCH": "0"},
    "stream+rccl": {"CICE_EVP_HIP_HALO": "rccl", "CICE_EVP_HIP_RESIDENT": "0", "CICE_EVP_HIP_MARCH": "0"},
    # several subcycles per pass, the four-cell ring exchanged every eighth subcycle over RCCL send/recv (here: the E-W seam of the piece,
    # with the rank itself -- pack, ncclGroup, unpack are those of a real run; a 4x2 piece has two more sides)
    "march+rccl": {"CICE_EVP_HIP_HALO": "rccl", "CICE_EVP_HIP_RESIDENT": "0", "CICE_EVP_HIP_MARCH": "1", "CICE_EVP_HIP_MARCH_SELFX": "1"},
}
KEYS = ["CICE_EVP_HIP_HALO", "CICE_EVP_HIP_RESIDENT", "CICE_EVP_HIP_SELF_EXCHANGE", "CICE_EVP_HIP_MARCH", "CICE_EVP_HIP_MARCH_SELFX"]


core_info = {}


def time_multiproc(wl, N, px, py, ndte=24):
    """N ranks as N processes on this GPU (tools/mailbox_2proc.py --timing): us per subcycle (slowest rank), tile variant, launches."""
    import ast
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in KEYS}
    env["CICE_EVP_HIP_HALO_TIMEOUT_MS"] = "20000"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={N}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), R + "/tools/mailbox_2proc.py", "--workload", wl, "--shape", f"{px}x{py}", "--ndte", str(ndte), "--timing"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("MAILBOX_2PROC")]
    if r.returncode != 0 or not line or " OK " not in line[-1]:
        raise RuntimeError((line[-1] if line else r.stdout[-300:] + r.stderr[-600:])[:400])
    import re
    res = ast.literal_eval(re.sub(r"np\.float64\(([^)]*)\)", r"\1", line[-1][line[-1].index("["):]))
    return max(q[2] for q in res), res[0][4], res[0][3]


def time_piece(nx, ny, ns, dx0, nsub, envs, selfx):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(envs)
    if selfx and "CICE_EVP_HIP_MARCH_SELFX" not in envs:
        os.environ["CICE_EVP_HIP_SELF_EXCHANGE"] = "1"
    g = synth.derive_geometry(synth.make_grid(nx, ny, dx0, ns=ns))
    st = synth.make_state(g, case="full", seed=1, warm=True)
    dc = decomp.Decomp(nx, ny, nx, ny, "cyclic", ns, 1)
    geo = {k: dc.scatter(g[k], 0, fill=(1.0 if k != "uarear" else 0.0)) for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
    fields = {k: dc.scatter(st[k], 0) for k in evp.FIELDS}
    tm, um = dc.scatter(st["iceTmask"], 0, fill=0), dc.scatter(st["iceUmask"], 0, fill=0)
    d, keep = evp.make_dims(dc, 0)
    core = evp.EvpHip(d, evp.make_params(synth.evp_scalars(120), strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"],
                      geo["uarear"], geo["tarea"], keepalive=keep)
    try:
        if selfx:
            core.comm_init(core.comm_unique_id())
        core.upload(fields, tm, um)
        core.subcycle(nsub)
        core.sync()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            core.subcycle(nsub)
        core.sync()
        t = (time.perf_counter() - t0) / (reps * nsub)
        tt = core.timings()
        global core_info
        core_info = core.march_info()
    finally:
        core.finalize()
    return 1e6 * t, tt


def main():
    out = []
    only = set(sys.argv[2:])
    for wl, (NX, NY, ns, ndte, layouts) in WORK.items():
        if only and wl not in only:
            continue
        dx0 = synth.GRIDS[wl]["dx0"]
        for N, shapes in layouts.items():
            for px, py in shapes:
                nx, ny = NX // px, NY // py
                if wl == "tx1" and N > 1:
                    try:
                        us, tv, lps = time_multiproc(wl, N, px, py)
                        rec = dict(workload=wl, n=N, layout=f"{px}x{py}", piece=f"{nx}x{ny}", transport="processes on ONE GPU: resident kernel across ranks, records + mailbox over HIP IPC",
                                   us_per_subcycle=us, tile_variant=tv, launches_per_subcycle=lps, march=False,
                                   note="tools/mailbox_2proc.py --timing; the N processes share the GPU",
                                   predicted_cell_updates_per_s=NX * NY / (us * 1e-6))
                    except Exception as e:  # noqa: BLE001
                        rec = dict(workload=wl, n=N, layout=f"{px}x{py}", piece=f"{nx}x{ny}", transport="processes on ONE GPU", error=str(e)[:300])
                    out.append(rec)
                    print("RESULT", rec, flush=True)
                    continue
                # a piece below the top row of a tripole grid has closed north/south neighbours here; the top piece keeps the fold
                for tname, envs in TRANSPORTS.items():
                    if N == 1 and tname != "resident-remote":
                        continue
                    if tname == "march+rccl" and wl != "s01":
                        continue
                    nsub = min(ndte, 120) if nx * ny < 500000 else 96
                    try:
                        us, tt = time_piece(nx, ny, ns, dx0, nsub, {} if N == 1 else envs, N > 1)
                    except Exception as e:  # noqa: BLE001
                        out.append(dict(workload=wl, n=N, layout=f"{px}x{py}", piece=f"{nx}x{ny}", transport=tname, error=str(e)[:200]))
                        print("RESULT", out[-1], flush=True)
                        continue
                    label = tname if N > 1 else "none"
                    if N > 1 and core_info.get("last_call") and tname != "march+rccl":
                        # pieces beyond the chip run the marching kernel by default; the one-GPU self-exchange hook of the
                        # one-subcycle kernels does not reach it: this row is the piece's compute time without any exchange
                        label = "march, no exchange (compute only)"
                    rec = dict(workload=wl, n=N, layout=f"{px}x{py}", piece=f"{nx}x{ny}", transport=label,
                               us_per_subcycle=us, tile_variant=tt["tile_variant"], halo_transport=tt["halo_transport"],
                               launches_per_subcycle=tt["launches_per_subcycle"], halo_cells=tt["halo_send_cells"], march=core_info.get("last_call"),
                               predicted_cell_updates_per_s=NX * NY / (us * 1e-6))
                    out.append(rec)
                    print("RESULT", rec, flush=True)
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
