"""CPU, world_size 2 (gloo): the multi-GPU halo path minus the device.

Each rank builds its own halo plan from the global block table through the C ABI
(host-only cice_evp_hip_plan_build), packs its send list, exchanges the buffers
point-to-point (torch.distributed gloo here, RCCL ncclSend/ncclRecv on the GPUs), unpacks
into its recv list and applies its local copies -- exactly the steps of halo_uv() in
cice_amd/csrc/evp_api.cpp.  The result must equal the known answer by global index
(the reference's halochk method, drivers/unittest/halochk/halochk.F90:232-247), which
proves that sender and receiver lists are in the same order without set-up traffic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cice_amd import decomp, evp

CASES = [
    # nx, ny, bx, by, ew, ns, nranks, proc_shape
    (40, 36, 20, 18, "cyclic", "closed", 2, (2, 1)),
    (40, 36, 10, 12, "cyclic", "closed", 2, (1, 2)),     # several blocks per rank, padded in y
    (37, 29, 10, 10, "closed", "closed", 2, (2, 1)),     # padded blocks in x and y
    (48, 24, 24, 24, "cyclic", "cyclic", 2, (2, 1)),     # each rank is its own N/S neighbour
]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nx, ny, bx, by, ew, ns, nranks, shape = case
        dc = decomp.Decomp(nx, ny, bx, by, ew, ns, nranks, shape)
        d, keep = evp.make_dims(dc, rank)
        plan = evp.halo_plan(d)
        gi = np.arange(1, nx + 1)[None, :] + 1000.0 * np.arange(1, ny + 1)[:, None]
        want = dc.scatter(gi, rank, fill=-999.0)          # ghosts by global index
        a = want.copy()
        for b in dc.local_blocks(rank):                   # wipe ghosts, keep interiors
            m = np.ones((dc.ny_block, dc.nx_block), bool)
            m[1:1 + b.gny, 1:1 + b.gnx] = False
            a[b.local][m] = -999.0
        flat = a.reshape(-1)
        # pack
        sendbuf = torch.from_numpy(flat[plan["send_src"]].copy())
        recvbuf = torch.zeros(len(plan["recv_dst"]), dtype=torch.float64)
        ops, so, ro = [], 0, 0
        for p, ns_, nr_ in zip(plan["peer_rank"], plan["peer_nsend"], plan["peer_nrecv"]):
            if ns_:
                ops.append(dist.P2POp(dist.isend, sendbuf[so:so + ns_], int(p)))
            if nr_:
                ops.append(dist.P2POp(dist.irecv, recvbuf[ro:ro + nr_], int(p)))
            so += ns_
            ro += nr_
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        # unpack + local copies
        flat[plan["recv_dst"]] = recvbuf.numpy()
        src = plan["local_src"]
        flat[plan["local_dst"]] = np.where(src >= 0, plan["local_sign"] * flat[np.maximum(src, 0)], 0.0)
        ok = bool(np.array_equal(a, want))
        nbad = int((a != want).sum())
        # mailbox semantics: the SENDER knows where each value lands in the receiver's array
        # (send_dst), the receiver knows which global cell each ghost mirrors (recv_gid)
        b2 = want.copy()
        for blk in dc.local_blocks(rank):
            m = np.ones((dc.ny_block, dc.nx_block), bool)
            m[1:1 + blk.gny, 1:1 + blk.gnx] = False
            b2[blk.local][m] = -999.0
        flat2 = b2.reshape(-1)
        vals = torch.from_numpy(flat2[plan["send_src"]].copy())
        dsts = torch.from_numpy(plan["send_dst"].astype(np.int64).copy())
        rvals = torch.zeros(len(plan["recv_dst"]), dtype=torch.float64)
        rdsts = torch.zeros(len(plan["recv_dst"]), dtype=torch.int64)
        ops, so, ro = [], 0, 0
        for p, ns_, nr_ in zip(plan["peer_rank"], plan["peer_nsend"], plan["peer_nrecv"]):
            if ns_:
                ops.append(dist.P2POp(dist.isend, vals[so:so + ns_], int(p), tag=1))
                ops.append(dist.P2POp(dist.isend, dsts[so:so + ns_], int(p), tag=2))
            if nr_:
                ops.append(dist.P2POp(dist.irecv, rvals[ro:ro + nr_], int(p), tag=1))
                ops.append(dist.P2POp(dist.irecv, rdsts[ro:ro + nr_], int(p), tag=2))
            so += ns_
            ro += nr_
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        flat2[rdsts.numpy()] = rvals.numpy()             # "remote store" at the sender's addresses
        flat2[plan["local_dst"]] = np.where(src >= 0, plan["local_sign"] * flat2[np.maximum(src, 0)], 0.0)
        ok = ok and bool(np.array_equal(b2, want)) and bool(np.array_equal(rdsts.numpy(), plan["recv_dst"]))
        gid = plan["recv_gid"]
        known = (gid % nx + 1) + 1000.0 * (gid // nx + 1)
        ok = ok and bool(np.array_equal(known, want.reshape(-1)[plan["recv_dst"]]))
        q.put((rank, ok, nbad, int(len(plan["send_src"])), int(len(plan["recv_dst"]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", CASES)
def test_two_rank_halo_exchange_known_answer(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, nbad, nsend, nrecv in sorted(res):
        assert ok, f"rank {rank}: {nbad} ghost cells wrong"
        assert nsend > 0 and nrecv > 0
    # what one rank sends the other receives
    r0, r1 = sorted(res)
    assert r0[3] == r1[4] and r0[4] == r1[3]
