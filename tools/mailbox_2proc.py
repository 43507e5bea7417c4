#!/usr/bin/env python3
"""N ranks as N processes on ONE GPU exchanging the velocity halo through the mailbox
transport (HIP IPC mapped inboxes + flag handshake), bootstrapped over gloo -- the only
multi-process GPU configuration a 1-GPU box allows (RCCL refuses two ranks on one device).
Each rank compares its sub-domain, bit for bit, with a single-rank run of the whole domain.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
         --master-port 29533 tools/mailbox_2proc.py [--workload gx3] [--ndte 24] [--timing]
"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gx3")
    ap.add_argument("--ndte", type=int, default=24)
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--shape", default="")          # e.g. 2x1
    ap.add_argument("--blocks-per-rank", default="", help="e.g. 2x2: split every rank's sub-domain into CICE blocks")
    ap.add_argument("--soak", type=int, default=0, help="N more launches of 120 subcycles before the comparison")
    ap.add_argument("--prep", action="store_true",
                    help="start from the primary model state: evp()'s preparation phase on the device on every "
                         "rank (T-grid halos across ranks through the same transport), then the loop")
    ap.add_argument("--maskhalo", action="store_true",
                    help="with --cgrid: hand the loop the masked halo evp() builds when maskhalo_dyn (five-point dilation of "
                         "iceTmask, ice_dyn_evp.F90:739-770); ghost cells outside the mask stay stale, everything else must not change")
    ap.add_argument("--march", action="store_true",
                    help="force the marching kernel on every rank; its ring exchanges and rank agreements go "
                         "through the library's test transport (host buffers + gloo here: RCCL refuses two ranks per device)")
    ap.add_argument("--case", default="full")
    ap.add_argument("--cgrid", action="store_true",
                    help="the C-grid subcycle (cice_evp_hip_cgrid_*): ghost cells other ranks own filled through the "
                         "same transport after every producing launch")
    ap.add_argument("--visc", default="avg_zeta")
    ap.add_argument("--expect-resident", action="store_true",
                    help="fail unless the on-chip resident kernel with remote neighbours ran")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from cice_amd import decomp, evp, synth

    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    os.environ["CICE_EVP_HIP_DEVICE"] = "0"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)

    if a.workload in synth.GRIDS:
        spec = synth.GRIDS[a.workload]
    else:                                       # "NXxNY" or "NXxNY:tripole": any size (tools/multiproc_sweep.py)
        size, _, bnd = a.workload.partition(":")
        spec = dict(nx=int(size.split("x")[0]), ny=int(size.split("x")[1]), dx0=1.1e5, ns=bnd or "closed")
    nx, ny = spec["nx"], spec["ny"]
    ns_bnd = spec.get("ns", "closed")          # tx1: tripole north boundary (rank layouts with px = 1 only)
    # ("NXxNY:tripoleT": the T-fold -- the same grid as the u-fold's; only the halo rule differs, and the top row is an image)
    g = synth.derive_geometry(synth.make_grid(nx, ny, spec["dx0"], ns=("tripole" if ns_bnd == "tripoleT" else ns_bnd)))
    st = synth.make_state(g, case="full", seed=7, warm=True)
    pr = synth.make_primary(g, "full", seed=9) if a.prep else None
    scal = synth.evp_scalars(120)

    def run_cgrid(dc, r, exchange):
        cg = synth.cgrid_geometry(g)
        state, inputs, masks = synth.cgrid_state(g, cg, case=a.case, seed=7, warm=True, seabed=True)
        tmg = np.asarray(masks["iceTmask"]) != 0          # global: the halo mask of the C-grid loop
        hmg = tmg | np.roll(tmg, 1, 1) | np.roll(tmg, -1, 1)
        hmg[1:] |= tmg[:-1]
        hmg[:-1] |= tmg[1:]
        static, state, inputs, masks = synth.cgrid_scatter(dc, r, cg, state, inputs, masks)
        d, keep = evp.make_dims(dc, r)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), static["dyE"], static["dxN"], static["dxT"], static["dyT"],
                          1.0 / static["uarea"], static["tarea"], keepalive=keep)
        try:
            if exchange:
                blobs = [None] * world
                dist.all_gather_object(blobs, core.halo_export())
                core.halo_import(blobs)
            hm = dc.scatter(hmg.astype(np.int32), r, fill=0)
            if exchange and a.maskhalo:
                core.halo_mask(hm)
            core.cgrid_set_geometry(static)
            if a.prep:
                # the preparation phase on the device on every rank: T-grid halos, E / N velocities and their averages
                # across ranks through the transport.  Ghost cells of the fields the preparation halo-updates itself are
                # handed over WRONG on purpose (aice, vice, vsno are read as given: ice_dyn_evp.F90:362-371)
                t11, st7, prev = synth.cgrid_prep_inputs(g, cg, case=a.case, seed=17)
                vec = ("uocn", "vocn", "ss_tltx", "ss_tlty", "strairxT", "strairyT")
                tb = {k: np.array(dc.scatter(v, r, fold=("center", -1.0 if k in vec else 1.0)), dtype=np.float64, order="C", copy=True)
                      for k, v in t11.items()}
                for k in tb:
                    if k in ("aice", "vice", "vsno"):
                        continue
                    for b in dc.local_blocks(r):
                        ring = np.ones(tb[k][b.local].shape, dtype=bool)
                        ring[1:1 + b.gny, 1:1 + b.gnx] = False
                        tb[k][b.local][ring] = 1.5 * tb[k][b.local][ring] + 0.125
                loc = {"umaskCD": "NEcorner", "emask": "Eface", "nmask": "Nface", "fcor_blk": "NEcorner", "fcorE_blk": "Eface", "fcorN_blk": "Nface"}
                static.update({k: dc.scatter(v, r, fill=0, fold=(loc.get(k, "center"), 1.0)) for k, v in st7.items()})
                prevb = {k: dc.scatter(v, r, fill=0) for k, v in prev.items()}
                core.cgrid_set_prep_geometry(static)
                pp = evp.PrepParams(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11, dyn_mass_min=1e-10,
                                    ssh_stress_coupled=0)
                newm = core.cgrid_prep(pp, tb, state, prevb)
                if exchange and a.maskhalo:
                    pass                  # (the mask above was built from the synthetic loop masks: not used together with --prep)
                core.cgrid_prep_finish(inputs["strength"], a.visc)
            else:
                core.cgrid_upload(state, inputs, masks, visc_method=a.visc)
            core.cgrid_subcycle(a.ndte)
            t = None
            if a.timing:
                core.cgrid_sync()
                if exchange:
                    dist.barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    core.cgrid_subcycle(40)
                core.cgrid_sync()
                t = (time.perf_counter() - t0) / 120 * 1e6
                core.cgrid_subcycle(7)
            out = core.cgrid_download()
            out["_halomask"] = hm
            if a.prep:                    # what the preparation produced, too
                for k in ("aiE", "forcexE", "emassdti", "aiN", "forceyN", "uocnN", "uvelE_init"):
                    out["prep_" + k] = core.cgrid_fetch(k)
                for k, v in newm.items():
                    out["prep_" + k] = v.astype(np.float64)
            return out, core.timings(), t
        finally:
            core.finalize()

    if a.cgrid:
        ref = None
        for turn in range(world):
            if turn == rank:
                ref, _, _ = run_cgrid(decomp.single_block(nx, ny, "cyclic", ns_bnd), 0, False)
            dist.barrier()
        shape = tuple(int(v) for v in a.shape.split("x")) if a.shape else None
        dcN = decomp.per_rank_blocks(nx, ny, world, "cyclic", ns_bnd, proc_shape=shape)
        if a.blocks_per_rank:
            sx, sy = (int(v) for v in a.blocks_per_rank.split("x"))
            dcN = decomp.Decomp(nx, ny, -(-dcN.block_size_x // sx), -(-dcN.block_size_y // sy), "cyclic", ns_bnd, world,
                                dcN.proc_shape)
        got, tim, t_us = run_cgrid(dcN, rank, True)
        assert tim["halo_transport"] == "mailbox", tim
        exchanged = ("uvelE", "vvelE", "uvelN", "vvelN", "uvel", "vvel", "stresspT", "stressmT", "stress12U", "zetax2T",
                     "etax2T", "shearU")
        bad = []
        for k in list(evp.CGRID_FIELDS) + [q for q in ref if q.startswith("prep_")]:
            want = dcN.scatter(ref[k][0][1:-1, 1:-1], rank)
            for b in dcN.local_blocks(rank):
                w = want[b.local][1:1 + b.gny, 1:1 + b.gnx]
                h = got[k][b.local][1:1 + b.gny, 1:1 + b.gnx]
                if not np.array_equal(w, h):
                    bad.append((k, int((w != h).sum()), float(np.abs(w - h).max())))
                if k in exchanged:      # ghost cells that mirror a cell (all but those beyond the closed north / south edge)
                    j0 = 1 if b.gj0 == 1 else 0
                    j1 = b.gny + 1 if b.gj0 + b.gny - 1 == ny else b.gny + 2
                    w2 = want[b.local][j0:j1, 0:b.gnx + 2]
                    h2 = got[k][b.local][j0:j1, 0:b.gnx + 2]
                    if a.maskhalo:      # ghost cells outside the mask are not refreshed (stale, as in the reference)
                        keep = got["_halomask"][b.local][j0:j1, 0:b.gnx + 2] != 0
                        w2, h2 = w2[keep], h2[keep]
                    if not np.array_equal(w2, h2):
                        bad.append((k + " ghosts", int((w2 != h2).sum()), float(np.abs(w2 - h2).max())))
        res = [None] * world
        dist.all_gather_object(res, (rank, bad, t_us, tim["halo_send_cells"]))
        if rank == 0:
            ok = all(not r[1] for r in res)
            print("MAILBOX_2PROC", "OK" if ok else "FAIL", "cgrid", a.workload, f"world={world}", res, flush=True)
            if not ok:
                sys.exit(1)
        dist.destroy_process_group()
        return

    def run(dc, r, exchange):
        geo = {k: dc.scatter(g[k], r, fill=(1.0 if k != "uarear" else 0.0))
               for k in ("HTE", "HTN", "dxT", "dyT", "tarea", "uarear")}
        fields = {k: dc.scatter(st[k], r) for k in evp.FIELDS}
        tm = dc.scatter(st["iceTmask"], r, fill=0)
        um = dc.scatter(st["iceUmask"], r, fill=0)
        d, keep = evp.make_dims(dc, r)
        # (the test transport of --march exists in the test build only; everything else runs the product library unless the
        # environment holds one of the test build's switches)
        core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"],
                          geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep, testing=(True if a.march else None))
        try:
            if exchange:
                blobs = [None] * world
                dist.all_gather_object(blobs, core.halo_export())
                core.halo_import(blobs)
            if exchange and a.march:
                def xchg(ranks, ns, nr, send, recv):
                    ops, so, ro = [], 0, 0
                    keepalive = []
                    for q, n_s, n_r in zip(ranks, ns, nr):
                        if q == rank:                       # (self-exchange across a seam this rank spans)
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
                    tt = torch.tensor([v], dtype=torch.int64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MIN if op == 0 else dist.ReduceOp.MAX)
                    return int(tt.item())
                core.set_test_transport(xchg, reduce)
            if a.prep:
                sc = lambda x, fill=0.0: dc.scatter(np.ascontiguousarray(x), r, fill=fill)
                static = {k: sc(v, (1.0 if k in ("tarea", "uarea") else 0)) for k, v in pr["static"].items()}
                core.set_prep_geometry(*[static[k] for k in ("tmask", "umask", "hm", "tarea", "uarea", "fcor_blk")])
                pp = evp.PrepParams(dt=3600.0, rhoi=917.0, rhos=330.0, gravit=9.80616, dyn_area_min=1e-11,
                                    dyn_mass_min=1e-10, ssh_stress_coupled=0)
                # T-grid fields the preparation halo-updates itself are handed over with WRONG ghost cells on purpose
                # (aice, vice, vsno are read as given, ice_dyn_evp.F90:362-371)
                tf = {k: np.array(sc(v), dtype=np.float64, order="C", copy=True) for k, v in pr["t"].items()}
                for k in tf:
                    if k in ("aice", "vice", "vsno"):
                        continue
                    for b in dc.local_blocks(r):
                        ring = np.ones(tf[k][b.local].shape, dtype=bool)
                        ring[1:1 + b.gny, 1:1 + b.gnx] = False
                        tf[k][b.local][ring] = 1.5 * tf[k][b.local][ring] + 0.125
                tmk, umk, _ = core.prep(pp, tf, {k: sc(v) for k, v in pr["state"].items()})
                core.set_strength(fields["strength"])
                extra = {k: core.prep_fetch(k) for k in ("forcexU", "umassdti", "uvel_init", "aiU")}
                extra["iceTmask"] = tmk.astype(np.float64)
            else:
                core.upload(fields, tm, um)
                extra = {}
            core.subcycle(a.ndte)
            t = None
            if a.timing:      # the same launches in the reference run: back-to-back loops are compared too
                core.sync()
                if exchange:
                    dist.barrier()
                t0 = time.perf_counter()
                for _ in range(5):
                    core.subcycle(120)
                core.sync()
                t = (time.perf_counter() - t0) / 600 * 1e6
                core.subcycle(7)          # an odd count: the record-buffer parity flips between launches
            for _ in range(a.soak):
                core.subcycle(120)
            if ns_bnd == "tripole" and not a.timing:
                core.stress_halo()          # evp()'s 12 x ice_HaloUpdate_stress on the resident stresses (any rank layout)
            out = core.download()
            out.update(extra)
            out["_march"] = core.march_info()
            return out, core.timings(), t
        finally:
            core.finalize()

    # The single-rank reference run assumes it has the GPU to itself (its resident kernel needs all its
    # tiles co-resident; a tx1-sized domain fills the chip): the processes take turns.
    ref = None
    for turn in range(world):
        if turn == rank:
            ref, _, _ = run(decomp.single_block(nx, ny, "cyclic", ns_bnd), 0, False)
        dist.barrier()
    shape = tuple(int(v) for v in a.shape.split("x")) if a.shape else None
    dcN = decomp.per_rank_blocks(nx, ny, world, "cyclic", ns_bnd, proc_shape=shape)
    if a.blocks_per_rank:
        sx, sy = (int(v) for v in a.blocks_per_rank.split("x"))
        dcN = decomp.Decomp(nx, ny, -(-dcN.block_size_x // sx), -(-dcN.block_size_y // sy), "cyclic", ns_bnd, world,
                            dcN.proc_shape)
    got, tim, t_us = run(dcN, rank, True)
    assert tim["halo_transport"] == "mailbox", tim
    bad = []
    if a.march:
        mi = got["_march"]
        if not (mi["mode"] == 1 and mi["last_call"] and mi["declined"] == 0 and mi["passes"] > 0):
            bad.append(("marching kernel did not run", mi))
        want_ring = "direct stores (HIP IPC)" if os.environ.get("CICE_EVP_HIP_MARCH_DIRECT") == "1" and os.environ.get("CICE_EVP_HIP_MARCH_OVERLAP", "0") != "1" else None
        if want_ring and mi["ring"] != want_ring:
            bad.append(("ring of the marching path not exchanged through the inboxes", mi))
    for k in ("uvel", "vvel", "stressp_1", "stressm_3", "stress12_4", "strintxU", "taubyU",
              "forcexU", "umassdti", "uvel_init", "aiU", "iceTmask"):
        if k not in got:
            continue
        want = dcN.scatter(ref[k][0][1:-1, 1:-1], rank)
        for b in dcN.local_blocks(rank):
            w = want[b.local][1:1 + b.gny, 1:1 + b.gnx]
            h = got[k][b.local][1:1 + b.gny, 1:1 + b.gnx]
            if not np.array_equal(w, h):
                bad.append((k, float(np.abs(w - h).max())))
        if k.startswith("stress") and ns_bnd == "tripole" and not a.timing:
            # the ghost row beyond the fold after the symmetrisation: the partner array's top row, mirrored (centre rule)
            fam, q = k.rsplit("_", 1)
            partner = f"{fam}_{((int(q) - 1) ^ 2) + 1}"
            wantp = dcN.scatter(ref[partner][0][1:-1, 1:-1], rank, fold=("center", 1.0))
            for b in dcN.local_blocks(rank):
                if b.gj0 + b.gny - 1 != ny:
                    continue
                w = wantp[b.local][b.gny + 1, :b.gnx + 2]
                h = got[k][b.local][b.gny + 1, :b.gnx + 2]
                if not np.array_equal(w, h):
                    bad.append((k + " fold ghost row", int((w != h).sum()), float(np.abs(w - h).max())))
        if k in ("uvel", "vvel"):      # ghost cells too (post-condition of the drop-in boundary)
            w2, h2 = want.copy(), got[k].copy()
            if ns_bnd in ("tripole", "tripoleT"):    # (scatter() does not know the fold: leave the folded ghost row out)
                for b in dcN.local_blocks(rank):
                    if b.gj0 + b.gny - 1 == ny:           # (row gny + 1: a padded block has spare rows above it)
                        w2[b.local][b.gny + 1:, :] = 0.0
                        h2[b.local][b.gny + 1:, :] = 0.0
            if not np.array_equal(w2, h2):
                bad.append((k + " ghosts", float(np.abs(w2 - h2).max())))
    res = [None] * world
    dist.all_gather_object(res, (rank, bad, t_us, tim["launches_per_subcycle"], tim["tile_variant"]))
    if rank == 0:
        ok = all(not r[1] for r in res)
        if a.expect_resident:
            ok = ok and all(r[4] >= 2000 for r in res)
        print("MAILBOX_2PROC", "OK" if ok else "FAIL", a.workload, f"world={world}", res, flush=True)
        if not ok:
            sys.exit(1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
