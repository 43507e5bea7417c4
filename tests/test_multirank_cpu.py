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
from common import bits_equal

CASES = [
    # nx, ny, bx, by, ew, ns, nranks, proc_shape
    (40, 36, 20, 18, "cyclic", "closed", 2, (2, 1)),
    (40, 36, 10, 12, "cyclic", "closed", 2, (1, 2)),     # several blocks per rank, padded in y
    (37, 29, 10, 10, "closed", "closed", 2, (2, 1)),     # padded blocks in x and y
    (48, 24, 24, 24, "cyclic", "cyclic", 2, (2, 1)),     # each rank is its own N/S neighbour
]


TRIPOLE_CASES = [
    # nx, ny, bx, by, nranks, proc_shape: the tripole seam row cut in x (tx1-on-8-GPUs-style layouts)
    (40, 24, 20, 12, 2, (2, 1)),
    (40, 24, 10, 12, 2, (2, 1)),      # two blocks per rank along the seam
    (36, 20, 9, 10, 4, (4, 1)),
    (40, 24, 20, 12, 4, (2, 2)),
    (44, 20, 11, 5, 4, (2, 2)),       # several blocks per rank
]


def _tripole_worker(rank, world, port, case, q):
    """Velocity halo on a tripole grid whose seam is split across ranks: exchange (ghost cells + raw seam values
    into the staging slots), local copies, general seam step -- against the oracle's single-rank update of the
    same global field, block by block."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
        import oracle
        nx, ny, bx, by, nranks, shape = case
        dc = decomp.Decomp(nx, ny, bx, by, "cyclic", "tripole", nranks, shape)
        d, keep = evp.make_dims(dc, rank)
        plan = evp.halo_plan(d)
        glob = np.random.default_rng(17).standard_normal((ny, nx))
        # the known answer: all blocks on ONE rank, updated by the oracle (pinned to the reference's tripole fixtures)
        one = decomp.Decomp(nx, ny, bx, by, "cyclic", "tripole", 1)
        ob = one.local_blocks(0)
        dom = oracle.OracleDomain(one.nx_block, one.ny_block, len(ob), nx, ny, "cyclic", "tripole",
                                  [b.ilo for b in ob], [b.ihi for b in ob], [b.jlo for b in ob], [b.jhi for b in ob],
                                  [b.gi0 for b in ob], [b.gj0 for b in ob])
        ref = oracle.halo_update(dom, np.ascontiguousarray(one.scatter(glob, 0, fill=0.0)), "NEcorner", "vector")
        mine = dc.local_blocks(rank)
        a = np.ascontiguousarray(dc.scatter(glob, rank, fill=0.0))
        for b in mine:                                    # ghosts start empty
            m = np.ones((dc.ny_block, dc.nx_block), bool)
            m[1:1 + b.gny, 1:1 + b.gnx] = False
            a[b.local][m] = 0.0
        flat = np.concatenate([a.reshape(-1), np.zeros(plan["tail"])])
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
        src = plan["local_src"]
        flat[plan["local_dst"]] = np.where(src >= 0, plan["local_sign"] * flat[np.maximum(src, 0)], 0.0)
        flat[plan["recv_dst"]] = plan["recv_sign"] * recvbuf.numpy()
        fa, fb, fc = plan["fin_a"], plan["fin_b"], plan["fin_coef"].astype(np.float64)
        res = np.where(fb >= 0, fc * (0.5 * (flat[fa] + (-1.0) * flat[np.maximum(fb, 0)])), fc * flat[fa])
        flat[plan["fin_dst"]] = res
        got = flat[:a.size].reshape(a.shape)
        nbad = 0
        for b in mine:                                    # same block in the one-rank decomposition
            k = next(o.local for o in ob if o.gi0 == b.gi0 and o.gj0 == b.gj0)
            nbad += int((got[b.local] != ref[k]).sum())
        q.put((rank, nbad, int(plan["tail"]), int(len(plan["fin_dst"]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", TRIPOLE_CASES)
def test_tripole_seam_split_across_ranks_known_answer(case):
    world = case[4]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tripole_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nbad, tail, nfin in sorted(res):
        assert nbad == 0, f"rank {rank}: {nbad} cells differ from the oracle's tripole halo update"
    assert sum(r[2] for r in res) > 0 and sum(r[3] for r in res) >= case[0]      # staging slots were needed; every seam cell finalised


TFOLD_RANK_CASES = [
    # nx, ny, bx, by, nranks, proc_shape: ns_boundary_type = 'tripoleT' -- the top physical row is itself an image (of row NY-1)
    (40, 24, 20, 12, 2, (2, 1)),      # the fold row cut in x
    (40, 24, 10, 6, 2, (1, 2)),       # cut in y only, several blocks per rank
    (36, 20, 9, 10, 4, (4, 1)),
    (44, 20, 11, 5, 4, (2, 2)),
]


def _tfold_worker(rank, world, port, case, q):
    """tripoleT split over ranks: receive lists name INTERIOR cells of the top row next to ghost cells.  The plan's lists +
    gloo against the oracle's T-fold halo update (NE-corner vector field) of the same global field on one rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
        import oracle
        nx, ny, bx, by, nranks, shape = case
        dc = decomp.Decomp(nx, ny, bx, by, "cyclic", "tripoleT", nranks, shape)
        d, keep = evp.make_dims(dc, rank)
        plan = evp.halo_plan(d)
        one = decomp.Decomp(nx, ny, bx, by, "cyclic", "tripoleT", 1)
        ob = one.local_blocks(0)
        dom = oracle.OracleDomain(one.nx_block, one.ny_block, len(ob), nx, ny, "cyclic", "tripoleT",
                                  [b.ilo for b in ob], [b.ihi for b in ob], [b.jlo for b in ob], [b.jhi for b in ob],
                                  [b.gi0 for b in ob], [b.gj0 for b in ob])
        glob = np.random.default_rng(29).standard_normal((ny, nx))
        ref = oracle.halo_update(dom, np.ascontiguousarray(one.scatter(glob, 0, fill=0.0)), "NEcorner", "vector")
        a = np.ascontiguousarray(dc.scatter(glob, rank, fill=0.0))
        mine = dc.local_blocks(rank)
        for b in mine:                                   # wipe the ghost cells; the interior (top row included) keeps its raw values
            m = np.ones((dc.ny_block, dc.nx_block), bool)
            m[1:1 + b.gny, 1:1 + b.gnx] = False
            a[b.local][m] = 0.0
        flat = a.reshape(-1)
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
        src = plan["local_src"]
        loc = np.where(src >= 0, plan["local_sign"] * flat[np.maximum(src, 0)], 0.0)      # (sources are rows NY-1 / NY-2: never a destination)
        flat[plan["recv_dst"]] = plan["recv_sign"] * recvbuf.numpy()
        flat[plan["local_dst"]] = loc
        nbad, n_int = 0, 0
        for b in mine:
            k = next(o.local for o in ob if o.gi0 == b.gi0 and o.gj0 == b.gj0)
            nbad += int((a[b.local] != ref[k]).sum())
        plane = dc.ny_block * dc.nx_block
        for dcell in plan["recv_dst"]:
            jj, ii = divmod(int(dcell) % plane, dc.nx_block)
            blk = mine[int(dcell) // plane]
            n_int += int(1 <= jj <= blk.gny and 1 <= ii <= blk.gnx)
        q.put((rank, nbad, n_int, int(len(plan["recv_dst"]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", TFOLD_RANK_CASES)
def test_tripoleT_split_over_ranks_known_answer(case):
    world = case[4]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tfold_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nbad, n_int, nrecv in sorted(res):
        assert nbad == 0, f"rank {rank}: {nbad} cells differ from the oracle's T-fold halo update"
    if case[5][0] > 1:       # the fold row cut in x: some rank receives into interior cells of its top row
        assert sum(r[2] for r in res) > 0


def _centre_fold_worker(rank, world, port, case, q):
    """Cell-centre fields on a tripole grid whose fold row is split over ranks: the ghost cells across the fold whose source
    another rank owns are filled by running the NE-corner exchange on a copy shifted by one cell (halo_plan.h) -- plan
    lists + gloo here, against the oracle's centre-rule halo update of the same global field (scalar and vector kind)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
        import oracle
        nx, ny, bx, by, nranks, shape = case
        dc = decomp.Decomp(nx, ny, bx, by, "cyclic", "tripole", nranks, shape)
        d, keep = evp.make_dims(dc, rank)
        plan = evp.halo_plan(d)
        fs = evp.fold_split_plan()
        one = decomp.Decomp(nx, ny, bx, by, "cyclic", "tripole", 1)
        ob = one.local_blocks(0)
        dom = oracle.OracleDomain(one.nx_block, one.ny_block, len(ob), nx, ny, "cyclic", "tripole",
                                  [b.ilo for b in ob], [b.ihi for b in ob], [b.jlo for b in ob], [b.jhi for b in ob],
                                  [b.gi0 for b in ob], [b.gj0 for b in ob])
        mine = dc.local_blocks(rank)

        def exchange(flat):
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
            flat[plan["recv_dst"]] = plan["recv_sign"] * recvbuf.numpy()

        nbad = 0
        for kind in ("scalar", "vector"):
            glob = np.random.default_rng(23).standard_normal((ny, nx))
            ref = oracle.halo_update(dom, np.ascontiguousarray(one.scatter(glob, 0, fill=0.0)), "center", kind)
            a = np.ascontiguousarray(dc.scatter(glob, rank, fill=0.0))
            for b in mine:
                m = np.ones((dc.ny_block, dc.nx_block), bool)
                m[1:1 + b.gny, 1:1 + b.gnx] = False
                a[b.local][m] = 0.0
            flat = np.concatenate([a.reshape(-1), np.zeros(plan["tail"])])
            # 1. ghosts with a source on this rank (centre rule), 2. the plain exchange: right for every remote ghost that is
            # not across the fold, 3. the exchange of the shifted copy for those that are
            src = plan["center_src"]
            vs = plan["center_vsign"] if kind == "vector" else np.ones_like(plan["center_vsign"])
            flat[plan["center_dst"]] = np.where(src >= 0, vs * flat[np.maximum(src, 0)], 0.0)
            exchange(flat)
            flat[fs["seam_dst"]] = flat[fs["seam_slot"]]      # east-west ghosts of row NY owned elsewhere: raw values from the staging slots
            flat[plan["center_dst"]] = np.where(src >= 0, vs * flat[np.maximum(src, 0)], 0.0)   # (the corner-rule exchange overwrote some)
            cp = np.zeros_like(flat)
            cp[fs["shift_cells"]] = flat[fs["shift_cells"] + dc.nx_block + 1]
            ls = plan["local_src"]                            # the whole NE-corner update of the copy: local part, remote part
            cp[plan["local_dst"]] = np.where(ls >= 0, plan["local_sign"] * cp[np.maximum(ls, 0)], 0.0)
            exchange(cp)
            flat[fs["center_dst"]] = (1.0 if kind == "vector" else -1.0) * cp[fs["center_dst"]]
            got = flat[:a.size].reshape(a.shape)
            for b in mine:
                k = next(o.local for o in ob if o.gi0 == b.gi0 and o.gj0 == b.gj0)
                nbad += int((got[b.local] != ref[k]).sum())
        q.put((rank, nbad, int(len(fs["center_dst"])), int(fs["fold_split"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", TRIPOLE_CASES)
def test_centre_fields_across_a_split_fold_known_answer(case):
    world = case[4]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_centre_fold_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nbad, ndst, split in sorted(res):
        assert nbad == 0, f"rank {rank}: {nbad} cells differ from the oracle's centre-rule halo update"
        assert split == 1
    assert sum(r[2] for r in res) > 0


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
        ok = bool(bits_equal(a, want))
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
        ok = ok and bool(bits_equal(b2, want)) and bool(bits_equal(rdsts.numpy(), plan["recv_dst"]))
        gid = plan["recv_gid"]
        known = (gid % nx + 1) + 1000.0 * (gid // nx + 1)
        ok = ok and bool(bits_equal(known, want.reshape(-1)[plan["recv_dst"]]))
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


CGRID_RANK_CASES = [
    # nx, ny, bx, by, ew, nranks, proc_shape, visc_method
    (40, 36, 20, 18, "cyclic", 2, (2, 1), "avg_zeta"),
    (40, 36, 10, 12, "cyclic", 2, (1, 2), "avg_strength"),      # several (padded) blocks per rank
    (44, 40, 22, 20, "closed", 4, (2, 2), "avg_zeta"),
]


def _cgrid_worker(rank, world, port, case, q):
    """The C-grid subcycle loop on the blocks of ONE rank (the oracle's arithmetic, block by block), with every halo
    update of the loop done the way the GPU path does it across ranks: ghost cells this rank owns the source of by the
    plan's local lists, the others by a point-to-point exchange of the plan's send / recv lists (gloo here, mailbox
    stores or RCCL on the GPUs) -- the same lists at every exchange point, whatever the field's location.  Each rank's
    blocks, ghost cells of the exchanged fields included, must equal the single-rank run."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
        import oracle
        from cice_amd import synth
        nx, ny, bx, by, ew, nranks, shape, visc = case
        g = synth.make_grid(nx, ny, 1.1e5, ns="closed")
        g["ew"] = ew
        g = synth.derive_geometry(g)
        cg = synth.cgrid_geometry(g)
        st, inp, mk = synth.cgrid_state(g, cg, seed=5, seabed=True)
        scal = synth.evp_scalars(120)
        prm = oracle.make_params(**{k: scal[k] for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping",
                                                          "Ktens", "deltaminEVP", "u0", "cosw", "sinw", "rhow")})

        def domain(dc, r):
            ob = dc.local_blocks(r)
            return oracle.OracleDomain(dc.nx_block, dc.ny_block, len(ob), nx, ny, ew, "closed", [b.ilo for b in ob],
                                       [b.ihi for b in ob], [b.jlo for b in ob], [b.jhi for b in ob],
                                       [b.gi0 for b in ob], [b.gj0 for b in ob])

        # the known answer: all blocks on one rank, the oracle's own halo update
        one = decomp.Decomp(nx, ny, bx, by, ew, "closed", 1)
        s1 = synth.cgrid_scatter(one, 0, cg, st, inp, mk)
        ref = oracle.cgrid_subcycle(domain(one, 0), prm, 5, s1[1], s1[2], s1[0], s1[3], visc_method=visc)
        # this rank's share, the exchange done here
        dc = decomp.Decomp(nx, ny, bx, by, ew, "closed", nranks, shape)
        d, keep = evp.make_dims(dc, rank)
        plan = evp.halo_plan(d)
        n = len(dc.local_blocks(rank)) * dc.ny_block * dc.nx_block
        calls = []

        def halo(aptr, loc, kind):
            flat = np.ctypeslib.as_array(aptr, shape=(n,))
            calls.append(loc)
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
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            src = plan["local_src"]
            vals = np.where(src >= 0, flat[np.maximum(src, 0)], 0.0)
            flat[plan["local_dst"]] = vals
            flat[plan["recv_dst"]] = recvbuf.numpy()

        oracle.set_halo_callback(halo)
        try:
            sN = synth.cgrid_scatter(dc, rank, cg, st, inp, mk)
            got = oracle.cgrid_subcycle(domain(dc, rank), prm, 5, sN[1], sN[2], sN[0], sN[3], visc_method=visc)
        finally:
            oracle.set_halo_callback(None)
        exchanged = ("uvelE", "vvelE", "uvelN", "vvelN", "uvel", "vvel", "stresspT", "stressmT", "stress12U", "zetax2T",
                     "etax2T", "shearU")
        nbad = 0
        mine = dc.local_blocks(rank)
        ob = one.local_blocks(0)
        for k in oracle.C_FIELDS:
            for b in mine:
                kb = next(o.local for o in ob if o.gi0 == b.gi0 and o.gj0 == b.gj0)
                if k in exchanged:
                    nbad += int((got[k][b.local] != ref[k][kb]).sum())
                else:
                    nbad += int((got[k][b.local][1:1 + b.gny, 1:1 + b.gnx] != ref[k][kb][1:1 + b.gny, 1:1 + b.gnx]).sum())
        q.put((rank, nbad, len(calls), float(np.abs(ref["uvelE"]).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", CGRID_RANK_CASES)
def test_cgrid_loop_split_over_ranks_known_answer(case):
    world = case[5]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cgrid_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nbad, ncalls, umax in sorted(res):
        assert nbad == 0, f"rank {rank}: {nbad} values differ from the single-rank run"
        assert ncalls == 5 * 12 and umax > 1e-3         # 12 fields exchanged per subcycle (eight ice_HaloUpdate calls)


# ---------------------------------------------------------------------------------------------------------------
# Several-subcycles-per-pass path (cice_amd/csrc/evp_march.hip): the exchange of the FOUR-cell ring (P = EVP_MARCH_PAD) between
# the ranks' rectangles (march_plan.cpp), once per pass of four.  world_size-2/4 gloo processes: every rank builds its plan from the
# global block table alone, packs, exchanges point-to-point, unpacks -- the known answer is the global cell number.
# ---------------------------------------------------------------------------------------------------------------
MARCH_CASES = [
    # nx, ny, bx, by, ew, nranks, proc_shape, own_max, wrap_inside, ext
    (130, 40, 65, 40, "cyclic", 2, (2, 1), 56, True, 0),      # x split: the cyclic seam and the inner cut, both between ranks
    (130, 40, 130, 20, "cyclic", 2, (1, 2), 56, True, 0),     # y slabs: every rank wraps inside and trades halo ROWS (duplicates!)
    (96, 48, 48, 24, "cyclic", 4, (2, 2), 20, True, 0),       # 2 x 2, narrow strips, corner cells from the diagonal neighbour
    (96, 48, 24, 24, "closed", 4, (2, 2), 56, True, 0),       # closed E-W, two blocks per rank
    (75, 30, 75, 15, "cyclic", 2, (1, 2), 56, False, 0),      # test hook: the cyclic seam exchanged with the rank itself
    # ext = 2 / 4: every rank holds (and advances redundantly) a rim of its neighbours' cells; ring of ext + 4 per exchange
    (130, 40, 65, 40, "cyclic", 2, (2, 1), 56, True, 2),
    (130, 40, 130, 20, "cyclic", 2, (1, 2), 56, True, 2),
    (96, 48, 48, 24, "cyclic", 4, (2, 2), 20, True, 4),
    (96, 48, 24, 24, "closed", 4, (2, 2), 56, True, 2),
    (75, 30, 75, 15, "cyclic", 2, (1, 2), 56, False, 2),
    # 61 + 2 + 2 = 65 columns held = 4 x 16 + 1: strips of 16 would leave ONE column to the last strip, and the first columns
    # beyond the rectangle would sit in two strips (found by a geometry sweep on the GPU: seed 2056, when the ring was two cells
    # wide and the rule "at least two") -- the last strip keeps at least P = 4 columns: 13 is chosen (5 x 13 = 65)
    (122, 30, 61, 30, "cyclic", 2, (2, 1), 16, True, 2),
]


def _march_worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nx, ny, bx, by, ew, nranks, shape, own_max, wrap_inside, ext = case
        dc = decomp.Decomp(nx, ny, bx, by, ew, "closed", nranks, shape)
        d, keep = evp.make_dims(dc, rank)
        P = evp.march_plan(d, own_max, wrap_inside, ext)
        own, ns, nxr, nyr, gx0, gy0 = P["own"], P["nstrips"], P["nxr"], P["nyr"], P["gx0"], P["gy0"]   # what the rank HOLDS
        ew_, ee_, es_, en_ = P["ext"]
        PW = 4                     # EVP_MARCH_PAD: overlap lanes, halo rows, cells of the ring
        rows = nyr + 2 * PW

        def home(x, y):            # where the rank holds column x (may lie PW cells beyond the rectangle), row y
            s = min(max(x, 0) // own, ns - 1)
            return ((y + PW) * ns + s) * 64 + (x - s * own + PW)

        buf = np.full(rows * ns * 64, -1.0)
        for y in range(es_, nyr - en_):              # its OWN cells
            for x in range(ew_, nxr - ee_):
                buf[home(x, y)] = (gy0 + y) * nx + (gx0 + x) % nx
        sendbuf = torch.from_numpy(buf[P["send_pos"]].copy())
        assert (sendbuf >= 0).all(), "a send entry that is not an owned cell"
        recvbuf = torch.zeros(len(P["recv_pos1"]), dtype=torch.float64)
        ops, so, ro, selfcopy = [], 0, 0, None
        for p, ns_, nr_ in zip(P["peer_rank"], P["peer_nsend"], P["peer_nrecv"]):
            if int(p) == rank:     # the rank itself (seam exchanged with itself): same order on both sides by construction
                assert ns_ == nr_
                selfcopy = (so, ro, int(ns_))
            else:
                if ns_:
                    ops.append(dist.P2POp(dist.isend, sendbuf[so:so + ns_], int(p)))
                if nr_:
                    ops.append(dist.P2POp(dist.irecv, recvbuf[ro:ro + nr_], int(p)))
            so += ns_
            ro += nr_
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if selfcopy:
            recvbuf[selfcopy[1]:selfcopy[1] + selfcopy[2]] = sendbuf[selfcopy[0]:selfcopy[0] + selfcopy[2]]
        buf[P["recv_pos1"]] = recvbuf.numpy()
        has2 = P["recv_pos2"] >= 0
        buf[P["recv_pos2"][has2]] = recvbuf.numpy()[has2]
        # known answer: every cell of the PW-cell ring that exists in the global domain
        nbad = nring = 0
        for y in range(-PW, nyr + PW):
            for x in range(-PW, nxr + PW):
                if ew_ <= x < nxr - ee_ and es_ <= y < nyr - en_:
                    continue
                gx, gy = gx0 + x, gy0 + y
                if not (0 <= gy < ny):
                    continue
                if not (0 <= gx < nx):
                    if ew != "cyclic":
                        continue
                    if P["wrapx"]:
                        continue       # read through the wrap of the strips, never stored
                    gx %= nx
                nring += 1
                nbad += int(buf[home(x, y)] != gy * nx + gx)
                # ... and in EVERY lane the kernel reads it from: the PW overlap lanes on either side of every strip
                for s_ in range(ns):
                    cnt = min(own, nxr - s_ * own)
                    for l in list(range(PW)) + list(range(cnt + PW, cnt + 2 * PW)):
                        if s_ * own - PW + l == x:
                            nbad += int(buf[((y + PW) * ns + s_) * 64 + l] != gy * nx + gx)
        q.put((rank, nbad, nring, int(has2.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", MARCH_CASES)
def test_march_ring_between_ranks_known_answer(case):
    world = case[5]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_march_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, nbad, nring, ndup in res:
        assert nbad == 0 and nring > 0, (rank, nbad, nring)
    if case[8] and case[6][0] == 1 and case[9] == 0:        # y slabs wrapping inside: halo-row cells next to a strip edge have duplicates
        assert all(r[3] > 0 for r in res)


def test_march_plan_refuses_a_rank_too_thin_next_to_a_closed_boundary():
    """ADVICE (round 3): the redundant rim + ring of a rank must not reach past a CLOSED global boundary through a
    neighbour thinner than ext + P cells (P = 4, the width of the ring) -- those positions would stay zero here while the owner
    advances the same cells from the caller's boundary ghost values.  The plan refuses the layout on every rank alike (the
    one-subcycle kernels run it); with a cyclic dimension or a wide enough neighbour it is accepted."""
    def plan(nx, bx, ew, ext, rank):
        dc = decomp.Decomp(nx, 24, bx, 24, ew, "closed", 2, (2, 1))
        d, keep = evp.make_dims(dc, rank)
        return evp.march_plan(d, 0, True, ext)

    for rank in (0, 1):
        with pytest.raises(evp.EvpHipError, match="closer to a closed boundary"):
            plan(40, 36, "closed", 4, rank)           # blocks of 36 + 4 columns: the east rank is 4 < ext + 4 wide
        with pytest.raises(evp.EvpHipError, match="closer to a closed boundary|too small"):
            plan(39, 36, "closed", 0, rank)           # 36 + 3: thinner than the ring itself
        assert plan(40, 36, "closed", 0, rank)["ext"] is not None      # no redundant rim, four columns for a four-cell ring
        assert plan(40, 36, "cyclic", 4, rank)["nxr"] > 0               # cyclic: no closed boundary in x
        assert plan(48, 24, "closed", 4, rank)["nxr"] == 24 + 4         # 24 + 24: wide enough


@pytest.mark.parametrize("shape", [(2, 1), (3, 1), (4, 1), (2, 2), (4, 2)])
def test_fold_row_split_over_ranks_lists_of_the_on_chip_kernel(shape):
    """Round 4: the on-chip kernel on a tripole grid whose fold row is split in x.  Host-only consistency of what every rank
    derives from the global block table without talking to the others:
      * a seam cell with its pair partner on another rank polls a staging slot -- the partner's rank must send exactly that
        column's raw value to exactly that slot (the kernel stores raw records there);
      * every ghost image of a seam cell of another rank is fed by that rank with the right sign, same order on both sides;
      * the send list's signs are the receiver's (remote images across the fold carry their sign on the producer's side);
      * the first n_ghost entries of corresponding send / recv lists are ghost cells, what follows are staging slots.
    (3 x 1, 4 x 1, 4 x 2: ranks that hold neither a pole point nor a whole pair -- the layouts that ran without fold handling
    before tripole_seam() looked at the general seam list.)"""
    NX, NY = 48, 20
    px, py = shape
    dc = decomp.per_rank_blocks(NX, NY, px * py, "cyclic", "tripole", shape)
    nr = px * py
    plans, ncell = {}, {}
    for r in range(nr):
        d, keep = evp.make_dims(dc, r)
        plans[r] = evp.halo_plan(d)
        ncell[r] = int(np.prod(dc.shape(r)))
    nxb, plane = dc.nx_block, dc.nx_block * dc.ny_block

    def gcol(r, c):
        b = dc.local_blocks(r)[c // plane]
        return b.gi0 + ((c % plane) % nxb - 1), b.gj0 + ((c % plane) // nxb - 1)

    def owner(ig, jg):
        return next(b.owner for b in dc.blocks if b.gi0 <= ig < b.gi0 + b.gnx and b.gj0 <= jg < b.gj0 + b.gny)

    def peer_slices(P, q):
        so, ro = int(P["peer_nsend"][:q].sum()), int(P["peer_nrecv"][:q].sum())
        return slice(so, so + int(P["peer_nsend"][q])), slice(ro, ro + int(P["peer_nrecv"][q]))

    npairs = nimg = 0
    for r in range(nr):
        P, n = plans[r], ncell[r]
        # (1) remote pair partners
        for dst, a, b in zip(P["fin_dst"], P["fin_a"], P["fin_b"]):
            if b < 0 or (dst != a and dst != b) or (a < n and b < n):
                continue
            slot = int(b if dst == a else a)
            ig, jg = gcol(r, int(dst))
            assert jg == NY and slot >= n
            o = owner(NX - ig, NY)
            Q = plans[o]
            q = list(Q["peer_rank"]).index(r)
            ss, _ = peer_slices(Q, q)
            k = [k for k in range(ss.start, ss.stop) if Q["send_dst"][k] == slot]
            assert len(k) == 1 and k[0] - ss.start >= Q["peer_counts4"][q][0], (shape, r, ig)
            assert gcol(o, int(Q["send_src"][k[0]])) == (NX - ig, NY)
            npairs += 1
        # (2) seam images and (3), (4) list heads
        oo = io = 0
        for q, pr in enumerate(P["peer_rank"]):
            ss, rs = peer_slices(P, q)
            Q = plans[int(pr)]
            qq = list(Q["peer_rank"]).index(r)
            sq, rq = peer_slices(Q, qq)
            ng_s, ng_r, n_out, n_in = [int(v) for v in P["peer_counts4"][q]]
            assert ng_s == Q["peer_counts4"][qq][1] and ng_r == Q["peer_counts4"][qq][0]
            assert (P["send_dst"][ss][:ng_s] < ncell[int(pr)]).all() and (P["send_dst"][ss][ng_s:] >= ncell[int(pr)]).all()
            assert bits_equal(P["send_sign"][ss], Q["recv_sign"][rq]) and bits_equal(P["send_dst"][ss], Q["recv_dst"][rq])
            oq = int(Q["peer_counts4"][:qq, 3].sum())
            mine = P["fimg_out"][oo:oo + n_out]
            theirs = Q["fimg_in"][oq:oq + int(Q["peer_counts4"][qq][3])]
            assert len(mine) == len(theirs)
            for (src, dstc, sg), (d2, col, sg2) in zip(mine, theirs):
                assert dstc == d2 and sg == sg2 and gcol(r, int(src)) == (int(col), NY)
                nimg += 1
            oo += n_out
            io += n_in
    assert npairs > 0 and nimg > 0
