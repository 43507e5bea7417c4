// Host side of the multi-GPU transports: mailbox halo over HIP IPC (kernel: evp_halo_direct.hip),
// its probes, the probe of the resident kernel across GPUs, and the RCCL bootstrap.
#include "evp_host.h"

namespace evp_host {

// ---- mailbox halo: set-up over HIP IPC (kernel: evp_halo_direct.hip) ------------------------
// What a rank tells the others: how to map its mailbox and where each peer's entries land.
struct HaloBlob {
    uint32_t magic, version;
    int32_t rank, npeers;
    uint64_t host_id;
    int64_t pid;
    uint64_t base;                 // mailbox address in the exporting process
    uint64_t inbox_off, n_recv;
    uint64_t rec_off, rec_stride;  // record buffers of the resident kernel inside the mailbox (0: none)
    uint64_t raw_off, raw_stride;  // ... and its raw seam records (tripole fold row split over ranks; 0: none)
    int32_t can_res, pad_;         // this rank can run the resident kernel with remote neighbours
    hipIpcMemHandle_t handle;
    struct { int32_t rank, recv_off, count, flag_idx; } peer[EVP_DIRECT_MAXPEER];
};
static_assert(sizeof(HaloBlob) <= CICE_EVP_HIP_HALO_BLOB, "HaloBlob must fit CICE_EVP_HIP_HALO_BLOB");
constexpr uint32_t HALO_BLOB_MAGIC = 0x45565048u;   // "EVPH"

uint64_t host_identity()
{
    char name[256] = {0};
    (void)gethostname(name, sizeof name - 1);
    uint64_t h = 1469598103934665603ull;
    for (const char *c = name; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
    return h;
}

int direct_export(HaloBlob &B)
{
    State::Direct &X = S.direct;
    const int np = (int)S.plan.peers.size();
    if (np > EVP_DIRECT_MAXPEER) return fail(-8, "mailbox halo: %d peers > %d", np, EVP_DIRECT_MAXPEER);
    // resident kernel across GPUs: its record buffers must be writable by the neighbours, so they
    // live in the mailbox allocation (one IPC handle)
    bool want_res = resident_possible(true) && !S.plan.peers.empty() &&
                    !(env("CICE_EVP_HIP_RESIDENT") && std::atoi(env("CICE_EVP_HIP_RESIDENT")) == 0) &&
                    !(env_test("CICE_EVP_HIP_RES_REMOTE") && std::atoi(env_test("CICE_EVP_HIP_RES_REMOTE")) == 0);
    size_t rec_off = 0;
    const size_t rec_stride = S.n * 32;            // one 32-byte record pair per cell of every block
    const size_t raw_stride = (S.n + (size_t)S.plan.tail) * 32;   // raw seam records: the cells + the staging slots of remote partners
    if (!X.mailbox) {
        X.inbox_off = DIRECT_INBOX_OFF;
        X.bytes = X.inbox_off + 2 * 2 * (size_t)std::max(S.n_recv, 1) * sizeof(double);
        X.bytes = (X.bytes + 255) & ~(size_t)255;
        if (want_res) { rec_off = X.bytes; X.bytes += 2 * rec_stride; }
        X.rec_off = rec_off;
        if (want_res && S.plan.tail > 0) { X.raw_off = X.bytes; X.bytes += 2 * raw_stride; }
        // fine-grained: stores of another GPU become visible to loads here without a kernel boundary
        // (no coarse-grained fallback: without this property a peer's stores are only guaranteed
        // to be seen at kernel boundaries, and the transport would be wrong on a real node)
        HIPC(hipExtMallocWithFlags(&X.mailbox, X.bytes, hipDeviceMallocFinegrained));
        HIPC(hipMemset(X.mailbox, 0, X.bytes));
    }
    int can_res = 0;
    if (want_res && X.rec_off) {
        if (S.res2_rec_owned)
            for (auto &q : S.res2_rec) { if (q) (void)hipFree(q); q = nullptr; }
        S.res2_rec_owned = false;
        S.res2_rec[0] = (char *)X.mailbox + X.rec_off;
        S.res2_rec[1] = (char *)X.mailbox + X.rec_off + rec_stride;
        if (X.raw_off) {
            if (S.res2_raw_owned)
                for (auto &q : S.res2_rec_raw) { if (q) (void)hipFree(q); q = nullptr; }
            S.res2_raw_owned = false;
            S.res2_rec_raw[0] = (char *)X.mailbox + X.raw_off;
            S.res2_rec_raw[1] = (char *)X.mailbox + X.raw_off + raw_stride;
        }
        const int forced_w = env_test("CICE_EVP_HIP_RES_LOGW") ? std::atoi(env_test("CICE_EVP_HIP_RES_LOGW")) : 0;
        for (int logw : {4, 5, 6}) {
            if (forced_w && logw != forced_w) continue;
            if (resident2_setup(logw)) continue;
            if (resident2_fits(true)) { can_res = 1; break; }
        }
        g_err.clear();
    }
    std::memset(&B, 0, sizeof B);
    B.can_res = can_res;
    B.rec_off = X.rec_off;
    B.rec_stride = rec_stride;
    B.raw_off = X.raw_off;
    B.raw_stride = raw_stride;
    B.magic = HALO_BLOB_MAGIC;
    B.version = 1;
    B.rank = S.d.rank;
    B.npeers = np;
    B.host_id = host_identity();
    B.pid = (int64_t)getpid();
    B.base = (uint64_t)(uintptr_t)X.mailbox;
    B.inbox_off = X.inbox_off;
    B.n_recv = (uint64_t)S.n_recv;
    HIPC(hipIpcGetMemHandle(&B.handle, X.mailbox));
    int ro = 0;
    for (int q = 0; q < np; ++q) {
        const HaloPeer &p = S.plan.peers[q];
        B.peer[q].rank = p.rank;
        B.peer[q].recv_off = ro;
        B.peer[q].count = (int)p.recv_dst.size();
        B.peer[q].flag_idx = q;
        ro += (int)p.recv_dst.size();
    }
    X.exported = true;
    return 0;
}

// Map every peer's mailbox and build the device tables.  Local decision only (no communication).
int direct_import(const HaloBlob *blobs, int nranks)
{
    State::Direct &X = S.direct;
    if (!X.exported) return fail(-8, "mailbox halo: import before export");
    if (nranks != S.d.nranks) return fail(-8, "mailbox halo: %d blobs for %d ranks", nranks, S.d.nranks);
    const int np = (int)S.plan.peers.size();
    std::vector<double *> send_addr((size_t)std::max(S.n_send, 1), nullptr);
    std::vector<unsigned> send_pstride((size_t)std::max(S.n_send, 1), 0u);
    std::vector<unsigned *> peer_flag((size_t)std::max(np, 1));
    std::map<int, char *> mapped;
    size_t so = 0;
    for (int q = 0; q < np; ++q) {
        const HaloPeer &p = S.plan.peers[q];
        if (p.rank < 0 || p.rank >= nranks) return fail(-8, "mailbox halo: peer rank %d out of range", p.rank);
        const HaloBlob &B = blobs[p.rank];
        if (B.magic != HALO_BLOB_MAGIC || B.version != 1 || B.rank != p.rank)
            return fail(-8, "mailbox halo: bad blob of rank %d", p.rank);
        if (B.host_id != host_identity()) return fail(-8, "mailbox halo: rank %d is on another host", p.rank);
        int e = -1;
        for (int k = 0; k < B.npeers; ++k)
            if (B.peer[k].rank == S.d.rank) e = k;
        if (e < 0 || B.peer[e].count != (int)p.send_src.size())
            return fail(-8, "mailbox halo: rank %d expects %d cells from this rank, plan sends %d", p.rank,
                        e < 0 ? -1 : B.peer[e].count, (int)p.send_src.size());
        char *base = nullptr;
        if (B.pid == (int64_t)getpid()) base = (char *)(uintptr_t)B.base;       // same process (self-exchange)
        else if (mapped.count(p.rank)) base = mapped[p.rank];
        else {
            void *ptr = nullptr;
            HIPC(hipIpcOpenMemHandle(&ptr, B.handle, hipIpcMemLazyEnablePeerAccess));
            X.opened.push_back(ptr);
            base = (char *)ptr;
        }
        mapped[p.rank] = base;
        peer_flag[q] = (unsigned *)base + (size_t)B.peer[e].flag_idx * EVP_DIRECT_FLAG_STRIDE;
        for (size_t k = 0; k < p.send_src.size(); ++k) {
            send_addr[so + k] = (double *)(base + B.inbox_off) + 2 * ((size_t)B.peer[e].recv_off + k);
            send_pstride[so + k] = (unsigned)(2 * B.n_recv);
        }
        so += p.send_src.size();
    }
    auto up = [&](auto *&dptr, const auto &v) -> int {
        using T = typename std::remove_reference<decltype(v[0])>::type;
        if (!dptr) HIPC(hipMalloc((void **)&dptr, v.size() * sizeof(T)));
        HIPC(hipMemcpy((void *)dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    };
    if (up(X.send_addr, send_addr) || up(X.send_pstride, send_pstride) || up(X.peer_flag, peer_flag)) return -1;
    S.msk.h_send_addr = send_addr;          // kept for cice_evp_hip_halo_mask (compacted copies of these tables)
    S.msk.h_send_pstride = send_pstride;
    // resident kernel with neighbours on other GPUs: only if EVERY rank can run it
    bool all_res = true;
    for (int r = 0; r < nranks; ++r) all_res = all_res && blobs[r].magic == HALO_BLOB_MAGIC && blobs[r].can_res != 0;
    S.res_remote = false;
    if (all_res && np > 0) {
        std::vector<void *> prec((size_t)np), praw((size_t)np, nullptr);
        std::vector<size_t> pstr((size_t)np), prawstr((size_t)np, 0);
        std::vector<int2> rimg(S.n * 3, make_int2(-1, -1));     // per cell: up to three images on other ranks
        std::vector<int2> rraw;                                 // per seam cell: where its raw record goes on other ranks
        const bool split_fold = S.plan.tail > 0 || [&] { for (const HaloPeer &p : S.plan.peers) if (p.n_ghost_send < (int)p.send_src.size() || !p.fimg_src.empty()) return true; return false; }();
        if (split_fold) rraw.assign(S.n * 3, make_int2(-1, -1));
        bool ok = true;
        auto add = [&](std::vector<int2> &tab, size_t c, int2 v) {
            int e = 0;
            while (e < 3 && tab[c * 3 + e].x >= 0) ++e;
            if (e == 3) { ok = false; return; }
            tab[c * 3 + e] = v;
        };
        for (int q = 0; q < np && ok; ++q) {
            const HaloPeer &p = S.plan.peers[q];
            const HaloBlob &B = blobs[p.rank];
            prec[q] = mapped[p.rank] + B.rec_off;
            pstr[q] = (size_t)B.rec_stride;
            if (B.raw_off) { praw[q] = mapped[p.rank] + B.raw_off; prawstr[q] = (size_t)B.raw_stride; }
            if (env_test("CICE_EVP_HIP_RES_REMOTE_BREAK")) {      // test hook: records go nowhere -> the probe must fail
                void *dummy = nullptr;
                HIPC(hipMalloc(&dummy, 2 * (size_t)B.rec_stride));
                prec[q] = dummy;                             // (leaked on purpose: test processes only)
            }
            // ghost cells of the peer: the FINAL velocity (negative across the tripole fold)
            for (int k = 0; k < p.n_ghost_send && ok; ++k)
                add(rimg, (size_t)p.send_src[k], make_int2(q | (p.send_sign[k] < 0 ? 256 : 0), p.send_dst[k]));
            // ... its ghost images of this rank's seam cells likewise (after the averaging)
            for (size_t k = 0; k < p.fimg_src.size() && ok; ++k)
                add(rimg, (size_t)p.fimg_src[k], make_int2(q | (p.fimg_sign[k] < 0 ? 256 : 0), p.fimg_dst[k]));
            // raw seam values: into the peer's rec_raw buffer at the staging slot it polls
            // (the plan also sends a raw value to every rank that finalises a ghost image from it in the streaming path; the
            // on-chip kernel delivers those images as final values, so only the rank that owns the pair partner needs it)
            for (size_t k = (size_t)p.n_ghost_send; k < p.send_src.size() && ok; ++k) {
                const size_t c = (size_t)p.send_src[k];
                const int b = (int)(c / S.plane), i = (int)((c % S.plane) % S.d.nx_block) + 1;
                const int ig = S.iglob0[b] + (i - S.ilo[b]);
                const int pc = S.d.nx_global - ig, NY = S.d.ny_global;
                int owner = -1;
                for (size_t t = 0; t < S.gtab[0].size(); ++t)
                    if (pc >= S.gtab[0][t] && pc < S.gtab[0][t] + S.gtab[2][t] && NY >= S.gtab[1][t] && NY < S.gtab[1][t] + S.gtab[3][t])
                        owner = S.gtab[4][t];
                if (owner != p.rank) continue;
                if (!praw[q]) { ok = false; break; }
                add(rraw, c, make_int2(q, p.send_dst[k]));
            }
        }
        if (env("CICE_EVP_HIP_VERBOSE") && split_fold) {
            size_t nraw = 0, nimg = 0;
            for (const int2 &v : rraw) nraw += v.x >= 0;
            for (const int2 &v : rimg) nimg += v.x >= 0;
            std::fprintf(stderr, "[cice_evp_hip] rank %d: fold row split over ranks: %zu remote images, %zu raw seam records sent, %d staging slots, tables %s\n",
                         (int)S.d.rank, nimg, nraw, S.plan.tail, ok ? "fit" : "DO NOT FIT (more than three destinations of one cell)");
            for (int q = 0; q < np; ++q)
                std::fprintf(stderr, "[cice_evp_hip] rank %d: peer %d: %d ghost + %zu raw sends, %d ghost + %zu raw receives, %zu / %zu seam images out / in, raw buffer %s\n",
                             (int)S.d.rank, S.plan.peers[q].rank, S.plan.peers[q].n_ghost_send, S.plan.peers[q].send_src.size() - S.plan.peers[q].n_ghost_send,
                             S.plan.peers[q].n_ghost_recv, S.plan.peers[q].recv_dst.size() - S.plan.peers[q].n_ghost_recv,
                             S.plan.peers[q].fimg_src.size(), S.plan.peers[q].fimg_recv_dst.size(), praw[q] ? "mapped" : "none");
        }
        if (ok) {
            if (up(S.res2_rimg, rimg) || up(S.res2_peer_rec, prec) || up(S.res2_peer_rstride, pstr)) return -1;
            if (split_fold && (up(S.res2_rraw, rraw) || up(S.res2_peer_raw, praw) || up(S.res2_peer_raw_stride, prawstr))) return -1;
            S.res_remote = true;
        }
    }
    EvpDirect D;
    fill_direct(D);
    if (!X.d_dx) HIPC(hipMalloc((void **)&X.d_dx, sizeof(EvpDirect)));
    HIPC(hipMemcpy(X.d_dx, &D, sizeof(EvpDirect), hipMemcpyHostToDevice));
    if (!X.d_cnt) HIPC(hipMalloc((void **)&X.d_cnt, 32 * sizeof(unsigned)));
    HIPC(hipMemset(X.d_cnt, 0, 32 * sizeof(unsigned)));
    return 0;
}

// Probe exchange (collective): every interior cell carries its global cell number, every ghost
// must come back holding the number of the cell it mirrors (halochk.F90:232-247's method).
// Uses the velocity buffers before any state has been uploaded, and leaves them zeroed.
int direct_probe()
{
    std::vector<double> hu(S.nuv, 0.0), hv(S.nuv, 0.0);      // (staging slots of the tripole seam step included)
    const int nx = S.d.nx_block;
    for (int b = 0; b < S.d.nblocks; ++b)
        for (int j = S.jlo[b]; j <= S.jhi[b]; ++j)
            for (int i = S.ilo[b]; i <= S.ihi[b]; ++i) {
                const size_t c = b * S.plane + (size_t)(j - 1) * nx + (i - 1);
                const double gid = (double)((S.iglob0[b] + (i - S.ilo[b]) - 1) +
                                            (size_t)S.d.nx_global * (S.jglob0[b] + (j - S.jlo[b]) - 1));
                hu[c] = gid + 1.0;
                hv[c] = -2.0 * (gid + 1.0);
            }
    HIPC(hipMemcpyAsync(S.u[0], hu.data(), S.nuv * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemcpyAsync(S.v[0], hv.data(), S.nuv * sizeof(double), hipMemcpyHostToDevice, S.stream));
    EvpDirect D;
    fill_direct(D);
    for (int rep = 0; rep < 3; ++rep)           // both inbox parities, and a repeat
        evp_launch_halo_direct(D, S.u[0], S.v[0], S.stream);
    HIPC(hipMemcpyAsync(hu.data(), S.u[0], S.nuv * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipMemcpyAsync(hv.data(), S.v[0], S.nuv * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    int err = 0;
    HIPC(hipMemcpyAsync(&err, D.err, sizeof(int), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemsetAsync(S.u[0], 0, S.n * sizeof(double), S.stream));
    HIPC(hipMemsetAsync(S.v[0], 0, S.n * sizeof(double), S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    if (err) {
        HIPC(hipMemset(D.err, 0, sizeof(int)));
        return fail(-8, "mailbox halo probe: peer %d never signalled", S.plan.peers[err - 1].rank);
    }
    for (const HaloPeer &p : S.plan.peers)
        for (size_t k = 0; k < p.recv_dst.size(); ++k) {
            const double want = (double)p.recv_sign[k] * ((double)p.recv_gid[k] + 1.0);
            if (hu[p.recv_dst[k]] != want || hv[p.recv_dst[k]] != -2.0 * want)
                return fail(-8, "mailbox halo probe: ghost %d from rank %d holds %.17g, expected %.17g",
                            (int)p.recv_dst[k], p.rank, hu[p.recv_dst[k]], want);
        }
    return 0;
}

int direct_check_error()
{
    if (!S.direct.on) return 0;
    int e = 0;
    HIPC(hipMemcpy(&e, (char *)S.direct.mailbox + DIRECT_ERR_OFF, sizeof(int), hipMemcpyDeviceToHost));
    if (e) return fail(-8, "mailbox halo: rank %d never signalled within the time-out (CICE_EVP_HIP_HALO_TIMEOUT_MS)",
                       S.plan.peers[e - 1].rank);
    return 0;
}

// Probe of the resident kernel with neighbours on other GPUs (collective): no ice anywhere, every
// interior cell holds its global cell number as "velocity"; three subcycles later every ghost that
// mirrors another rank's cell must hold that cell's number -- carried there by tagged records only.
int resident_remote_probe()
{
    std::vector<double> hu(S.n, 0.0), hv(S.n, 0.0);
    const int nx = S.d.nx_block;
    for (int blk = 0; blk < S.d.nblocks; ++blk)
        for (int j = S.jlo[blk]; j <= S.jhi[blk]; ++j)
            for (int i = S.ilo[blk]; i <= S.ihi[blk]; ++i) {
                const size_t c = blk * S.plane + (size_t)(j - 1) * nx + (i - 1);
                const double gid = (double)((S.iglob0[blk] + (i - S.ilo[blk]) - 1) +
                                            (size_t)S.d.nx_global * (S.jglob0[blk] + (j - S.jlo[blk]) - 1));
                hu[c] = gid + 1.0;
                hv[c] = -2.0 * (gid + 1.0);
            }
    for (int b = 0; b < 2; ++b) {
        HIPC(hipMemcpyAsync(S.u[b], hu.data(), S.n * sizeof(double), hipMemcpyHostToDevice, S.stream));
        HIPC(hipMemcpyAsync(S.v[b], hv.data(), S.n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    }
    HIPC(hipMemsetAsync(S.mask, 0, S.n, S.stream));
    S.res_timeout_ms = 10000.0;
    int rc = launch_resident2(3, 0, false);
    S.res_timeout_ms = 0;
    if (rc) return rc;
    S.res_launched = true;
    std::vector<double> gu(S.n), gv(S.n);
    HIPC(hipMemcpyAsync(gu.data(), S.u[1], S.n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipMemcpyAsync(gv.data(), S.v[0], S.n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    rc = resident_check_error();
    S.res_mode = -1;            // (resident_check_error parks the mode on failure; decided again at upload)
    for (int b = 0; b < 2; ++b) {
        HIPC(hipMemsetAsync(S.u[b], 0, S.n * sizeof(double), S.stream));
        HIPC(hipMemsetAsync(S.v[b], 0, S.n * sizeof(double), S.stream));
    }
    HIPC(hipStreamSynchronize(S.stream));
    if (rc) return rc;
    const int NX = S.d.nx_global, NY = S.d.ny_global;
    for (const HaloPeer &p : S.plan.peers) {
        for (int k = 0; k < p.n_ghost_recv; ++k) {       // (the entries behind them are staging slots of the streaming path)
            const double want = (double)p.recv_sign[k] * ((double)p.recv_gid[k] + 1.0);
            if (gu[p.recv_dst[k]] != want || gv[p.recv_dst[k]] != -2.0 * want)
                return fail(-8, "resident kernel probe: ghost %d from rank %d holds %.17g, expected %.17g",
                            (int)p.recv_dst[k], p.rank, gu[p.recv_dst[k]], want);
        }
        // images of seam-row cells of other ranks: the owner's value AFTER the fold step -- the pair (lo, hi = NX - lo) holds
        // (xavg, -xavg), xavg = 0.5 * (x_lo - x_hi), from the first subcycle on; a pole point has changed sign three times
        for (size_t k = 0; k < p.fimg_recv_dst.size(); ++k) {
            const int col = p.fimg_recv_col[k];
            auto val = [&](int ig) { return (double)((ig - 1) + (size_t)NX * (NY - 1)) + 1.0; };
            double fin;
            if (col == NX / 2 || col == NX) fin = -val(col);
            else {
                const int lo = std::min(col, NX - col), hi = NX - lo;
                const double xavg = 0.5 * (val(lo) + (-1.0) * val(hi));
                fin = col == lo ? xavg : -1.0 * xavg;
            }
            const double want = (double)p.fimg_recv_sign[k] * fin;
            if (gu[p.fimg_recv_dst[k]] != want || gv[p.fimg_recv_dst[k]] != -2.0 * want)
                return fail(-8, "resident kernel probe: ghost %d (image of seam column %d of rank %d) holds %.17g, expected %.17g",
                            (int)p.fimg_recv_dst[k], col, p.rank, gu[p.fimg_recv_dst[k]], want);
        }
    }
    return 0;
}

// 0 auto, 1 RCCL only, 2 mailbox required
int halo_choice()
{
    const char *h = env("CICE_EVP_HIP_HALO");
    if (!h) return 0;
    if (!std::strcmp(h, "rccl")) return 1;
    if (!std::strcmp(h, "direct")) return 2;
    return 0;
}

}  // namespace evp_host

using namespace evp_host;

extern "C" {

int cice_evp_hip_comm_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    NCCLC(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

// the blob of a rank without blocks: nobody's peer, no objection to anything
static void bystander_blob(HaloBlob &B)
{
    std::memset(&B, 0, sizeof B);
    B.can_res = 1;
    B.magic = HALO_BLOB_MAGIC;
    B.version = 1;
    B.rank = S.d.rank;
    B.npeers = 0;
    B.host_id = host_identity();
    B.pid = (int64_t)getpid();
}

int cice_evp_hip_comm_info(int32_t *out, int32_t n, char *bus_id, int32_t nb)
{
    int32_t v[5] = {S.have_comm ? 1 : 0, -1, -1, -1, (int32_t)S.device};
    if (S.have_comm) {
        int c = -1, r = -1, dv = -1;
        NCCLC(ncclCommCount(S.comm, &c));
        NCCLC(ncclCommUserRank(S.comm, &r));
        NCCLC(ncclCommCuDevice(S.comm, &dv));
        v[1] = c; v[2] = r; v[3] = dv;
    }
    for (int k = 0; out && k < n && k < 5; ++k) out[k] = v[k];
    if (bus_id && nb > 0) {
        bus_id[0] = 0;
        if (S.ready || S.bystander) HIPC(hipDeviceGetPCIBusId(bus_id, nb, S.device));
    }
    return 0;
}

int cice_evp_hip_comm_init(const void *id128)
{
    if (!S.ready && !S.bystander) return fail(-1, "not initialised");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    HIPC(hipSetDevice(S.device));
    NCCLC(ncclCommInitRank(&S.comm, S.d.nranks, id, S.d.rank));
    S.have_comm = true;
    if (halo_choice() == 1) { S.direct.why = "CICE_EVP_HIP_HALO=rccl"; return 0; }
    // Mailbox halo: every step below is followed by an agreement (all-reduce of "still fine"),
    // so that either all ranks switch to it or all stay on RCCL.
    const int nr = S.d.nranks;
    char *d_blobs = nullptr;
    int *d_ok = nullptr;
    HIPC(hipMalloc((void **)&d_blobs, (size_t)nr * CICE_EVP_HIP_HALO_BLOB));
    HIPC(hipMalloc((void **)&d_ok, sizeof(int)));
    auto agree = [&](int mine, int &all) -> int {
        HIPC(hipMemcpy(d_ok, &mine, sizeof(int), hipMemcpyHostToDevice));
        NCCLC(ncclAllReduce(d_ok, d_ok, 1, ncclInt, ncclMin, S.comm, S.stream));
        HIPC(hipStreamSynchronize(S.stream));
        HIPC(hipMemcpy(&all, d_ok, sizeof(int), hipMemcpyDeviceToHost));
        return 0;
    };
    std::vector<char> mine(CICE_EVP_HIP_HALO_BLOB, 0), all((size_t)nr * CICE_EVP_HIP_HALO_BLOB, 0);
    int ok = 1, all_ok = 0;
    if (S.bystander) bystander_blob(*reinterpret_cast<HaloBlob *>(mine.data()));
    else ok = direct_export(*reinterpret_cast<HaloBlob *>(mine.data())) == 0;
    std::string why = ok ? "" : g_err;
    HIPC(hipMemcpy(d_blobs + (size_t)S.d.rank * CICE_EVP_HIP_HALO_BLOB, mine.data(), CICE_EVP_HIP_HALO_BLOB, hipMemcpyHostToDevice));
    NCCLC(ncclAllGather(d_blobs + (size_t)S.d.rank * CICE_EVP_HIP_HALO_BLOB, d_blobs, CICE_EVP_HIP_HALO_BLOB, ncclChar, S.comm, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemcpy(all.data(), d_blobs, all.size(), hipMemcpyDeviceToHost));
    if (ok && !S.bystander) {
        ok = direct_import(reinterpret_cast<const HaloBlob *>(all.data()), nr) == 0;
        if (!ok) why = g_err;
    }
    if (agree(ok, all_ok)) return -1;
    if (all_ok) {
        ok = S.bystander ? 1 : direct_probe() == 0;
        if (!ok) why = g_err;
        if (agree(ok, all_ok)) return -1;
    }
    if (all_ok) {           // resident kernel across GPUs: its own probe, same agreement
        // EVERY rank takes part in this agreement, whatever its own verdict so far: a rank without neighbours on other
        // ranks (or without blocks) has nothing to probe and votes yes -- a vote only some ranks enter would leave the
        // others in the all-reduce
        int res_ok = (S.bystander || !S.res_remote) ? 1 : resident_remote_probe() == 0, res_all = 0;
        const std::string why_res = res_ok ? "" : g_err;
        if (agree(res_ok, res_all)) return -1;
        if (!res_all && S.res_remote) {
            S.res_remote = false;
            if (env("CICE_EVP_HIP_VERBOSE"))
                std::fprintf(stderr, "[cice_evp_hip] rank %d: resident kernel across GPUs off (%s)\n", S.d.rank,
                             why_res.empty() ? "another rank's probe failed" : why_res.c_str());
        }
    }
    (void)hipFree(d_blobs);
    (void)hipFree(d_ok);
    S.direct.on = all_ok != 0;
    S.direct.why = S.direct.on ? "" : (why.empty() ? "another rank could not set it up" : why);
    if (!S.direct.on && halo_choice() == 2)
        return fail(-8, "CICE_EVP_HIP_HALO=direct but the mailbox halo is unavailable: %s", S.direct.why.c_str());
    if (!S.direct.on && env("CICE_EVP_HIP_VERBOSE"))
        std::fprintf(stderr, "[cice_evp_hip] rank %d: mailbox halo off (%s), using RCCL\n", S.d.rank, S.direct.why.c_str());
    g_err.clear();
    return 0;
}

// ice_HaloMask (ice_boundary.F90:889-1062; evp() builds it when maskhalo_dyn, ice_dyn_evp.F90:739-770): a reduced
// halo for the velocity updates INSIDE the subcycle loop.  halomask: the reference's own array -- 1 where
// iceUmask, ghost cells updated -- so that sender (mask of the source cell, :975-985) and receiver (mask of the
// ghost cell, :1018-1028) drop the same entries; copies inside a rank are never masked (:925-945) and messages
// across the tripole fold are always kept (:979, 1022).  NULL = back to the full halo.  Bit-neutral: a dropped
// ghost cell mirrors an ice-free cell, whose velocity is and stays 0 (dyn_prep2 + the full pre-loop update).
// The on-chip resident kernel across GPUs ignores it: its records are published every subcycle, ice or not.
int cice_evp_hip_halo_mask(const int32_t *halomask)
{
    if (!S.ready) return fail(-1, "not initialised");
    State::Masked &M = S.msk;
    for (auto &kv : S.graphs) (void)hipGraphExecDestroy(kv.second);     // captured loops bake the list lengths in
    S.graphs.clear();
    // The call is collective and both sides of every message must drop the same entries, so the on/off decision may
    // only depend on what every rank computes alike: the plan's global flag (some rank exchanges across the tripole
    // fold or through seam staging slots: the reference never masks those messages, ice_boundary.F90:979,1022, and
    // here the whole in-loop exchange then stays unmasked on ALL ranks), never this rank's own lists.
    if (!halomask || S.plan.any_fold_exchange || S.plan.tfold) { M.on = false; return 0; }   // (tripoleT: interior cells of the top row are destinations too)
    if (S.plan.peers.empty()) { M.on = false; return 0; }           // no messages at all: nothing to agree on
    std::vector<int32_t> ss, rd, rslot;
    std::vector<int8_t> rs;
    std::vector<double *> sa;
    std::vector<unsigned> sp;
    M.peer_nsend.assign(S.plan.peers.size(), 0);
    M.peer_nrecv.assign(S.plan.peers.size(), 0);
    size_t so = 0, ro = 0;
    for (size_t q = 0; q < S.plan.peers.size(); ++q) {
        const HaloPeer &p = S.plan.peers[q];
        for (size_t k = 0; k < p.send_src.size(); ++k)
            if (halomask[p.send_src[k]] != 0) {
                ss.push_back(p.send_src[k]);
                if (!M.h_send_addr.empty()) { sa.push_back(M.h_send_addr[so + k]); sp.push_back(M.h_send_pstride[so + k]); }
                ++M.peer_nsend[q];
            }
        for (size_t k = 0; k < p.recv_dst.size(); ++k)
            if (halomask[p.recv_dst[k]] != 0) {
                rd.push_back(p.recv_dst[k]);
                rs.push_back(p.recv_sign[k]);
                rslot.push_back((int32_t)(ro + k));
                ++M.peer_nrecv[q];
            }
        so += p.send_src.size();
        ro += p.recv_dst.size();
    }
    auto up = [&](auto *&dptr, const auto &v, size_t cap) -> int {
        using T = typename std::remove_reference<decltype(v[0])>::type;
        if (!dptr) HIPC(hipMalloc((void **)&dptr, std::max<size_t>(cap, 1) * sizeof(T)));
        if (!v.empty()) HIPC(hipMemcpyAsync((void *)dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, S.stream));
        return 0;
    };
    if (up(M.send_src, ss, S.n_send) || up(M.recv_dst, rd, S.n_recv) || up(M.recv_slot, rslot, S.n_recv) ||
        up(M.recv_sign, rs, S.n_recv)) return -1;
    if (!M.h_send_addr.empty() && (up(M.send_addr, sa, S.n_send) || up(M.send_pstride, sp, S.n_send))) return -1;
    M.n_send = (int)ss.size();
    M.n_recv = (int)rd.size();
    M.on = true;
    if (S.direct.on) {
        EvpDirect D;
        fill_direct(D, true);
        if (!S.direct.d_dx_m) HIPC(hipMalloc((void **)&S.direct.d_dx_m, sizeof(EvpDirect)));
        HIPC(hipMemcpyAsync(S.direct.d_dx_m, &D, sizeof(EvpDirect), hipMemcpyHostToDevice, S.stream));
    }
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

int cice_evp_hip_halo_export(void *blob)
{
    if (!S.ready && !S.bystander) return fail(-1, "not initialised");
    if (!blob) return fail(-1, "null blob");
    HIPC(hipSetDevice(S.device));
    std::vector<char> tmp(CICE_EVP_HIP_HALO_BLOB, 0);
    if (S.bystander) bystander_blob(*reinterpret_cast<HaloBlob *>(tmp.data()));
    else if (int rc = direct_export(*reinterpret_cast<HaloBlob *>(tmp.data()))) return rc;
    std::memcpy(blob, tmp.data(), CICE_EVP_HIP_HALO_BLOB);
    return 0;
}

int cice_evp_hip_halo_import(const void *blobs, int32_t nranks)
{
    if (!S.ready && !S.bystander) return fail(-1, "not initialised");
    if (!blobs) return fail(-1, "null blobs");
    if (S.bystander)       // nobody's peer: nothing to map, nothing to probe (the probe runs between peers)
        return nranks == S.d.nranks ? 0 : fail(-8, "mailbox halo: %d blobs for %d ranks", nranks, S.d.nranks);
    HIPC(hipSetDevice(S.device));
    std::vector<HaloBlob> B((size_t)std::max(nranks, 0));
    for (int r = 0; r < nranks; ++r)
        std::memcpy(&B[r], (const char *)blobs + (size_t)r * CICE_EVP_HIP_HALO_BLOB, sizeof(HaloBlob));
    if (int rc = direct_import(B.data(), nranks)) return rc;
    if (int rc = direct_probe()) return rc;     // collective; a failure here is fatal for the caller
    S.direct.on = true;
    if (S.res_remote)
        if (int rc = resident_remote_probe()) return rc;
    return 0;
}

}  // extern "C"
