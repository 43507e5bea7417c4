// Host side: the velocity halo (local images, tripole seam, remote transports) and the
// subcycle loop as enqueued work (serial, boundary-first + second stream, riding exchange).
#include "evp_host.h"

namespace evp_host {

void fill_direct(EvpDirect &D) { fill_direct(D, false); }

void fill_direct(EvpDirect &D, bool masked)
{
    State::Direct &X = S.direct;
    char *base = (char *)X.mailbox;
    masked = masked && S.msk.on;
    D.n_send = masked ? S.msk.n_send : S.n_send;
    D.n_recv = masked ? S.msk.n_recv : S.n_recv;
    D.n_recv_slots = S.n_recv;
    D.npeers = (int)S.plan.peers.size();
    D.send_src = masked ? S.msk.send_src : S.h_send_src;
    D.send_addr = masked ? S.msk.send_addr : X.send_addr;
    D.send_pstride = masked ? S.msk.send_pstride : X.send_pstride;
    D.recv_dst = masked ? S.msk.recv_dst : S.h_recv_dst;
    D.recv_sign = (const signed char *)(masked ? S.msk.recv_sign : S.h_recv_sign);
    D.recv_slot = masked ? S.msk.recv_slot : nullptr;
    D.flags_in = (unsigned *)base;
    D.seq = (unsigned *)(base + DIRECT_SEQ_OFF);
    D.err = (int *)(base + DIRECT_ERR_OFF);
    D.inbox = (double *)(base + X.inbox_off);
    const double tmo_ms = env("CICE_EVP_HIP_HALO_TIMEOUT_MS") ? std::atof(env("CICE_EVP_HIP_HALO_TIMEOUT_MS")) : 30000.0;
    D.timeout_ticks = (unsigned long long)(tmo_ms * 1.0e5);     // 100 MHz wall clock
    D.peer_flag = X.peer_flag;
    const int dbg = env_test("CICE_EVP_HIP_HALO_DEBUG") ? std::atoi(env_test("CICE_EVP_HIP_HALO_DEBUG")) : 0;
    D.dbg = dbg;
}

// Ghost cells whose source lives on another rank, for a pair of arrays laid out like uvel/vvel
// (the velocities of the loop; pairs of T-grid fields in the preparation phase on grids without
// a tripole fold, where cell-centre and corner fields mirror the same cells)
// masked: the in-loop velocity exchange on the entries ice_HaloMask keeps (every other exchange -- pre-loop
// velocities, T-grid fields -- uses the full lists, as the reference does with halo_info vs halo_info_mask)
int halo_remote_pair(double *a, double *bb, bool masked, bool has_tail)
{
    masked = masked && S.msk.on;
    // staging slots of the split tripole seam lie BEHIND the cells of an array (offsets >= S.n): the velocity buffers are
    // allocated with that tail (S.nuv), and so are the arrays a caller vouches for (has_tail); any other target would be
    // written out of bounds
    if (S.plan.tail > 0 && !has_tail && !((a == S.u[0] && bb == S.v[0]) || (a == S.u[1] && bb == S.v[1])))
        return fail(-3, "remote halo of a non-velocity array on a rank layout that splits the tripole seam row (staging slots)");
    if (!S.plan.peers.empty() && S.direct.on) {
        EvpDirect D;
        fill_direct(D, masked);
        evp_launch_halo_direct(D, a, bb, S.stream);
    } else if (!S.plan.peers.empty()) {
        if (!S.have_comm) return fail(-2, "remote halo needed but neither cice_evp_hip_comm_init nor cice_evp_hip_halo_import was called");
        evp_launch_halo_pack(a, bb, masked ? S.msk.send_src : S.h_send_src, S.sendbuf, masked ? S.msk.n_send : S.n_send, S.stream);
        size_t so = 0, ro = 0, q = 0;
        NCCLC(ncclGroupStart());
        for (const HaloPeer &p : S.plan.peers) {
            const size_t ns = masked ? (size_t)S.msk.peer_nsend[q] : p.send_src.size();
            const size_t nr = masked ? (size_t)S.msk.peer_nrecv[q] : p.recv_dst.size();
            if (ns) NCCLC(ncclSend(S.sendbuf + 2 * so, 2 * ns, ncclDouble, p.rank, S.comm, S.stream));
            if (nr) NCCLC(ncclRecv(S.recvbuf + 2 * ro, 2 * nr, ncclDouble, p.rank, S.comm, S.stream));
            so += ns;
            ro += nr;
            ++q;
        }
        NCCLC(ncclGroupEnd());
        evp_launch_halo_unpack(a, bb, masked ? S.msk.recv_dst : S.h_recv_dst,
                               (const signed char *)(masked ? S.msk.recv_sign : S.h_recv_sign), S.recvbuf,
                               masked ? S.msk.n_recv : S.n_recv, S.stream);
    }
    return 0;
}

static int foldx_setup();

int fold_seam_ghosts(double *a, double *b)
{
    if (foldx_setup()) return -1;
    State::FoldX &F = S.foldx;
    if (F.n_seam) evp_launch_halo_local(a, b, F.seam_dst, F.seam_slot, (const signed char *)F.seam_one, F.n_seam, S.stream);
    return 0;
}

int fold_remote_pair(const double *srcA, const double *srcB, double *dstA, double *dstB, int kind, double fa, double fb)
{
    if (foldx_setup()) return -1;
    State::FoldX &F = S.foldx;
    // every rank takes part in the exchange (its top-row cells may be somebody's sources) whether or not it has
    // destinations of its own
    evp_launch_fold_shift2(srcA, srcB, F.scr[0], F.scr[1], F.cells, F.n_cells, S.d.nx_block, S.stream);
    // the whole NE-corner update of the copies: a destination's corner-rule source may sit on this rank even though its
    // centre-rule source does not (the column next to a rank boundary)
    evp_launch_halo_local(F.scr[0], F.scr[1], S.h_local_dst, S.h_local_src, (const signed char *)S.h_local_sign, S.n_local, S.stream);
    if (int rc = halo_remote_pair(F.scr[0], F.scr[1], false, true)) return rc;
    evp_launch_fold_extract2(dstA, dstB, F.scr[0], F.scr[1], F.dst[kind], F.n_dst[kind], fa, fb, S.stream);
    return 0;
}

static int foldx_setup()
{
    State::FoldX &F = S.foldx;
    if (!F.ready) {
        const HaloPlan &P = S.plan;
        auto up = [&](int32_t *&dp, int &n, const std::vector<int32_t> &v) -> int {
            n = (int)v.size();
            if (n) {
                HIPC(hipMalloc((void **)&dp, v.size() * sizeof(int32_t)));
                HIPC(hipMemcpy(dp, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            }
            return 0;
        };
        int n2 = 0;
        if (up(F.cells, F.n_cells, P.fold_shift_cells) || up(F.dst[0], F.n_dst[0], P.center_foldr_dst) ||
            up(F.dst[1], F.n_dst[1], P.stress_foldr_dst) || up(F.seam_dst, F.n_seam, P.center_seam_dst) ||
            up(F.seam_slot, n2, P.center_seam_slot)) return -1;
        if (F.n_seam) {
            std::vector<int8_t> one(F.n_seam, 1);
            HIPC(hipMalloc((void **)&F.seam_one, one.size()));
            HIPC(hipMemcpy(F.seam_one, one.data(), one.size(), hipMemcpyHostToDevice));
        }
        for (auto &p : F.scr) {
            if (alloc_d(&p, S.nuv)) return -1;
            HIPC(hipMemsetAsync(p, 0, S.nuv * sizeof(double), S.stream));
        }
        F.ready = true;
    }
    return 0;
}

// velocity halo of buffer `b` (ice_dyn_evp.F90:908-910)
int halo_uv(int b, bool masked)
{
    const bool pushed = S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH);
    if (!pushed)
        evp_launch_halo_local(S.u[b], S.v[b], S.h_local_dst, S.h_local_src,
                              (const signed char *)S.h_local_sign, S.n_local, S.stream);
    const bool general = S.plan.tail > 0 || (env_test("CICE_EVP_HIP_SEAM_FIN") && std::atoi(env_test("CICE_EVP_HIP_SEAM_FIN")));
    if (!general) {
        // tripole seam of the top physical row, every pair and every ghost image of a seam cell on this rank: the
        // remote exchange below never involves seam-row cells
        evp_launch_halo_seam(S.u[b], S.v[b], masked ? S.u[b ^ 1] : nullptr, masked ? S.v[b ^ 1] : nullptr, S.h_seam_a, S.h_seam_b, S.n_seam, S.h_seam_pole, S.n_pole,
                             S.h_late_dst, S.h_late_src, (const signed char *)S.h_late_sign, S.n_late, S.stream);
        if (int rc = halo_remote_pair(S.u[b], S.v[b], masked)) return rc;
        return 0;
    }
    // any rank layout: the exchange first (it also carries the RAW seam values other ranks hold into the staging
    // slots behind the arrays), then every seam cell and every ghost image of one is finalised from raw values
    if (int rc = halo_remote_pair(S.u[b], S.v[b], masked)) return rc;
    evp_launch_halo_seam_fin(S.u[b], S.v[b], masked ? S.u[b ^ 1] : nullptr, masked ? S.v[b ^ 1] : nullptr, S.h_fin_dst, S.h_fin_a, S.h_fin_b, (const signed char *)S.h_fin_coef, S.n_fin, S.stream);
    return 0;
}

// Boundary-first + second stream pays when the interior kernel is long enough to hide the
// exchange; on small per-rank domains the extra host calls (events, two launches) cost more
// than they hide (measured: 50 vs 26 us per subcycle on gx1, eager).  CICE_EVP_HIP_OVERLAP=1/0 forces.
bool use_overlap()
{
    const bool seam = (S.n_seam + S.n_pole + S.n_late) > 0;
    if (!S.overlap || S.plan.peers.empty() || seam || S.plan.tfold || !(S.have_comm || S.direct.on)) return false;
    if (env_test("CICE_EVP_HIP_OVERLAP")) return std::atoi(env_test("CICE_EVP_HIP_OVERLAP")) != 0;
    size_t cells = 0;
    for (int b = 0; b < S.d.nblocks; ++b)
        cells += (size_t)(S.ihi[b] - S.ilo[b] + 1) * (S.jhi[b] - S.jlo[b] + 1);
    return cells >= 400000;
}

// The mailbox exchange can ride in the subcycle launch (no tripole seam step in between).
bool use_riding_exchange()
{
    if (!S.direct.on || S.plan.peers.empty() || (S.n_seam + S.n_pole + S.n_late) > 0 || S.plan.tfold) return false;
    // Pays when the interior tiles outlast the exchange (measured per subcycle, riding vs separate
    // kernel: 4 x 1800x1200 blocks 571 vs 627 us, 720x540 27.9 vs 34.1, 720x270 19.4 vs 22.1); on a
    // domain that is one wave of workgroups there is nothing to overlap with and the separate kernel
    // is quicker (gx1, 2 x 320x192: 25 vs 20.8 us).
    if (env_test("CICE_EVP_HIP_HALO_RIDE")) return std::atoi(env_test("CICE_EVP_HIP_HALO_RIDE")) != 0;
    size_t cells = 0;
    for (int b = 0; b < S.d.nblocks; ++b)
        cells += (size_t)(S.ihi[b] - S.ilo[b] + 1) * (S.jhi[b] - S.jlo[b] + 1);
    return cells >= 160000;
}

// Which tiles of `variant` hold U-cells that some other rank mirrors (send list)?
int get_tile_split(int variant, State::TileSplit **out)
{
    auto it = S.splits.find(variant);
    if (it != S.splits.end()) { *out = &it->second; return 0; }
    int tyb, gx, gy;
    evp_tile_geometry(S.max_ni, S.max_nj, variant, &tyb, &gx, &gy);
    const int ntiles = gx * gy * S.d.nblocks;
    std::vector<char> is_b((size_t)ntiles, 0);
    const int nx = S.d.nx_block;
    for (const HaloPeer &p : S.plan.peers)
        for (int32_t src : p.send_src) {
            const int b = (int)(src / S.plane);
            const int rem = (int)(src % S.plane);
            const int j = rem / nx + 1, i = rem % nx + 1;
            const int bx = (i - S.ilo[b]) / 63, by = (j - S.jlo[b]) / (tyb - 1);
            if (bx < 0 || bx >= gx || by < 0 || by >= gy) continue;
            is_b[((size_t)b * gy + by) * gx + bx] = 1;       // row-major tile id (xcdmap 0/2 decoding)
        }
    std::vector<int> lb, li;
    for (int t = 0; t < ntiles; ++t) (is_b[t] ? lb : li).push_back(t);
    State::TileSplit ts;
    ts.nb = (int)lb.size();
    ts.ni = (int)li.size();
    if (ts.nb) {
        HIPC(hipMalloc((void **)&ts.d_boundary, lb.size() * sizeof(int)));
        HIPC(hipMemcpy(ts.d_boundary, lb.data(), lb.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (ts.ni) {
        HIPC(hipMalloc((void **)&ts.d_interior, li.size() * sizeof(int)));
        HIPC(hipMemcpy(ts.d_interior, li.data(), li.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    {   // boundary tiles first, then the rest: order of the launch that carries the exchange workgroup
        // interior tiles in XCD-chunked order: workgroup w runs on XCD w % 8, so give each XCD one
        // contiguous run of the (row-major) interior sequence -- neighbouring tiles share an L2
        std::vector<int> all(lb);
        const size_t n = li.size(), per = (n + 7) / 8, w0 = lb.size() + 1;    // +1: the exchange workgroup
        std::vector<int> chunked;
        for (size_t w = 0; chunked.size() < n; ++w) {
            const size_t x = (w0 + w) & 7, q = x * per + (w >> 3);
            if ((w >> 3) < per && q < n) chunked.push_back(li[q]);
            if (w > 16 * (n + 8)) break;
        }
        if (chunked.size() != n) chunked = li;
        all.insert(all.end(), chunked.begin(), chunked.end());
        HIPC(hipMalloc((void **)&ts.d_all, all.size() * sizeof(int)));
        HIPC(hipMemcpy(ts.d_all, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    *out = &S.splits.emplace(variant, ts).first->second;
    return 0;
}

int enqueue_loop(int ndte, int cur0)
{
    int cur = cur0;
    const bool strict = S.prm.strict != 0;
    const int cap = cap_mode();
    // Boundary strips first, RCCL exchange on a second stream while the interior tiles run
    // (the tripole seam needs every tile of the top row first, so it keeps the serial order).
    const bool overlap = use_overlap();
    if (use_riding_exchange()) {
        // mailbox halo: one launch per subcycle; the tiles other ranks wait for run first, one
        // extra workgroup exchanges their velocities while the interior tiles are computed
        const int variant = S.tyb % 100;
        State::TileSplit *ts = nullptr;
        if (int rc = get_tile_split(variant, &ts)) return rc;
        const bool pushed = S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH);
        for (int k = 0; k < ndte; ++k) {
            EvpArgs A;
            fill_args(A, cur, k == ndte - 1);
            A.tile_list = ts->d_all; A.tile_count = ts->nb + ts->ni;
            A.dx = (S.msk.on && S.direct.d_dx_m) ? S.direct.d_dx_m : S.direct.d_dx; A.dx_count = S.direct.d_cnt; A.dx_fseq = S.direct.d_cnt + 16; A.dx_nb = ts->nb;
            evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, variant, strict, cap, S.stream);
            if (!pushed)
                evp_launch_halo_local(S.u[cur ^ 1], S.v[cur ^ 1], S.h_local_dst, S.h_local_src,
                                      (const signed char *)S.h_local_sign, S.n_local, S.stream);
            cur ^= 1;
        }
        HIPC(hipGetLastError());
        return 0;
    }
    if (!overlap) {
        for (int k = 0; k < ndte; ++k) {
            EvpArgs A;
            fill_args(A, cur, k == ndte - 1);
            evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, S.tyb, strict, cap, S.stream);
            if (int rc = halo_uv(cur ^ 1, true)) return rc;
            cur ^= 1;
        }
        HIPC(hipGetLastError());
        return 0;
    }
    // the split kernels decode tile ids row-major: use the row-major flavour of the variant
    const int variant = S.tyb % 100;
    State::TileSplit *ts = nullptr;
    if (int rc = get_tile_split(variant, &ts)) return rc;
    for (int k = 0; k < ndte; ++k) {
        const int nxt = cur ^ 1;
        EvpArgs A;
        fill_args(A, cur, k == ndte - 1);
        if (k > 0) HIPC(hipStreamWaitEvent(S.stream, S.ev_halo, 0));   // ghosts of u_in complete
        // 1. tiles whose cells other ranks need
        A.tile_list = ts->d_boundary; A.tile_count = ts->nb;
        evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, variant, strict, cap, S.stream);
        const bool mk = S.msk.on;
        if (!S.direct.on) evp_launch_halo_pack(S.u[nxt], S.v[nxt], mk ? S.msk.send_src : S.h_send_src, S.sendbuf, mk ? S.msk.n_send : S.n_send, S.stream);
        HIPC(hipEventRecord(S.ev_pack, S.stream));
        // 2. everything else, concurrently with the exchange
        A.tile_list = ts->d_interior; A.tile_count = ts->ni;
        evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, variant, strict, cap, S.stream);
        if (!(S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH)))
            evp_launch_halo_local(S.u[nxt], S.v[nxt], S.h_local_dst, S.h_local_src,
                                  (const signed char *)S.h_local_sign, S.n_local, S.stream);
        // 3. RCCL point-to-point over xGMI on the communication stream
        HIPC(hipStreamWaitEvent(S.stream_comm, S.ev_pack, 0));
        if (S.direct.on) {      // mailbox exchange: stores into the peers' inboxes, no library call
            EvpDirect D;
            fill_direct(D, true);
            evp_launch_halo_direct(D, S.u[nxt], S.v[nxt], S.stream_comm);
        } else {
            size_t so = 0, ro = 0, q = 0;
            NCCLC(ncclGroupStart());
            for (const HaloPeer &p : S.plan.peers) {
                const size_t ns = mk ? (size_t)S.msk.peer_nsend[q] : p.send_src.size();
                const size_t nr = mk ? (size_t)S.msk.peer_nrecv[q] : p.recv_dst.size();
                if (ns) NCCLC(ncclSend(S.sendbuf + 2 * so, 2 * ns, ncclDouble, p.rank, S.comm, S.stream_comm));
                if (nr) NCCLC(ncclRecv(S.recvbuf + 2 * ro, 2 * nr, ncclDouble, p.rank, S.comm, S.stream_comm));
                so += ns;
                ro += nr;
                ++q;
            }
            NCCLC(ncclGroupEnd());
            evp_launch_halo_unpack(S.u[nxt], S.v[nxt], mk ? S.msk.recv_dst : S.h_recv_dst,
                                   (const signed char *)(mk ? S.msk.recv_sign : S.h_recv_sign), S.recvbuf,
                                   mk ? S.msk.n_recv : S.n_recv, S.stream_comm);
        }
        HIPC(hipEventRecord(S.ev_halo, S.stream_comm));
        cur = nxt;
    }
    HIPC(hipStreamWaitEvent(S.stream, S.ev_halo, 0));   // the compute stream owns the final state
    HIPC(hipGetLastError());
    return 0;
}

}  // namespace evp_host
