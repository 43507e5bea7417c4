// =====================================================================
// Host side of the two-subcycles-per-pass kernel (evp_march.hip): the device-private rectangle layout, the
// conversions between it and the CICE block layout, and the loop.
//
// Rectangle layout: all blocks of the rank assembled into ONE array per field -- (nxr + 2 halo columns each side,
// rounded up to the strips' reach) x (nyr + 2 halo rows each side), i fastest.  Every cell of the rank exists once:
// the reference's redundant copies (ghost cells, the T-cells of the north / east fringe every block computes for
// itself, ice_dyn_shared.F90:740-749) are images of it.  Halo columns image owned columns when the rectangle spans a
// cyclic E-W dimension; on closed sides their first layer holds the caller's ghost values (what stress reads there,
// constant during the loop), everything else is 0 with mask 0.
//
// A call of cice_evp_hip_subcycle(ndte) through this path:
//   [ndte odd: one subcycle with the one-subcycle kernel]  gather -> consistency check (first call after an upload)
//   -> ndte/2 passes (ping-pong between two sets of u, v, 12 stresses) -> scatter.
// Not eligible (the one-subcycle kernels keep running): several ranks, tripole / cyclic north-south boundary, blocks
// that do not tile a rectangle (eliminated land blocks), metric terms handed over as arrays (tripole), a caller
// whose ghost values are not images of one global state.
// =====================================================================
#include "evp_host.h"

namespace evp_host {

namespace {

struct MarchBuf {
    double *u[2] = {}, *v[2] = {}, *sig[2][12] = {};
    double *stat[5] = {};        // dxT dyT HTE HTN uarear
    double *in[13] = {};         // strength vrelfac uocn vocn forcex forcey umassdti fm | waterx watery TbU uvel_init vvel_init
    double *diag[4] = {};
    uint8_t *mask = nullptr;
    unsigned *bad = nullptr;
    int *blkid = nullptr;
    int2 *org = nullptr;
};
MarchBuf B;

template <class T> void F(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

}  // namespace

void march_free()
{
    for (int k = 0; k < 2; ++k) {
        F(B.u[k]); F(B.v[k]);
        for (auto &p : B.sig[k]) F(p);
    }
    for (auto &p : B.stat) F(p);
    for (auto &p : B.in) F(p);
    for (auto &p : B.diag) F(p);
    F(B.mask); F(B.bad); F(B.blkid); F(B.org);
    S.march = State::March{};
}

// Can this rank's sub-domain be held as one rectangle?  Fills S.march.G (host fields) when it can.
static bool march_geometry(std::string &why)
{
    State::March &M = S.march;
    const cice_evp_hip_dims &d = S.d;
    if (d.nranks > 1 || !S.plan.peers.empty()) { why = "several ranks"; return false; }
    if (d.nghost != 1) { why = "nghost != 1"; return false; }
    if (d.ns_boundary_type == CICE_EVP_BND_TRIPOLE || d.ns_boundary_type == CICE_EVP_BND_CYCLIC) { why = "north-south boundary is not closed"; return false; }
    if (!(S.flags & EVP_F_METRICS) || (S.flags & EVP_F_DXHY_ARRAY)) { why = "metric terms come from arrays"; return false; }
    if (d.nblocks < 1) { why = "no blocks"; return false; }
    // blocks must tile [gx0, gx0+nxr) x [gy0, gy0+nyr) with full blocks of bsx x bsy (the last column / row may be smaller)
    int gx0 = 1 << 30, gy0 = 1 << 30, gx1 = 0, gy1 = 0;
    for (int b = 0; b < d.nblocks; ++b) {
        gx0 = std::min(gx0, S.iglob0[b]); gy0 = std::min(gy0, S.jglob0[b]);
        gx1 = std::max(gx1, S.iglob0[b] + (S.ihi[b] - S.ilo[b])); gy1 = std::max(gy1, S.jglob0[b] + (S.jhi[b] - S.jlo[b]));
    }
    const int nxr = gx1 - gx0 + 1, nyr = gy1 - gy0 + 1;
    int bsx = 0, bsy = 0;
    for (int b = 0; b < d.nblocks; ++b) {
        if (S.iglob0[b] == gx0) bsx = std::max(bsx, S.ihi[b] - S.ilo[b] + 1);
        if (S.jglob0[b] == gy0) bsy = std::max(bsy, S.jhi[b] - S.jlo[b] + 1);
    }
    if (bsx <= 0 || bsy <= 0) { why = "no block at the origin of the rectangle"; return false; }
    const int nbx = (nxr + bsx - 1) / bsx, nby = (nyr + bsy - 1) / bsy;
    if ((long)nbx * nby != d.nblocks) { why = "blocks do not tile a rectangle"; return false; }
    M.blkid_h.assign((size_t)nbx * nby, -1);
    M.org_h.assign(d.nblocks, int2{0, 0});
    for (int b = 0; b < d.nblocks; ++b) {
        const int ox = S.iglob0[b] - gx0, oy = S.jglob0[b] - gy0;
        if (ox % bsx || oy % bsy) { why = "block origins off the block grid"; return false; }
        const int bi = ox / bsx, bj = oy / bsy;
        const int wx = std::min(bsx, nxr - ox), wy = std::min(bsy, nyr - oy);
        if (S.ihi[b] - S.ilo[b] + 1 != wx || S.jhi[b] - S.jlo[b] + 1 != wy) { why = "irregular block sizes"; return false; }
        if (M.blkid_h[(size_t)bj * nbx + bi] >= 0) { why = "two blocks at one place"; return false; }
        M.blkid_h[(size_t)bj * nbx + bi] = b;
        M.org_h[b] = int2{ox, oy};
    }
    const bool wrapx = d.ew_boundary_type == CICE_EVP_BND_CYCLIC;
    if (wrapx && nxr != d.nx_global) { why = "cyclic dimension not spanned by this rank"; return false; }
    if (nxr < 4 || nyr < 1) { why = "rectangle too small"; return false; }
    EvpMarchGeo &G = M.G;
    G.nxr = nxr; G.nyr = nyr;
    M.nstrips = (nxr + EVP_MARCH_OWN - 1) / EVP_MARCH_OWN;
    G.ldx = ((M.nstrips * EVP_MARCH_OWN + 2 * EVP_MARCH_PAD + 7) / 8) * 8;
    G.rows = nyr + 2 * EVP_MARCH_PAD;
    G.nxb = d.nx_block; G.nyb = d.ny_block; G.plane = (int)S.plane; G.nblocks = d.nblocks;
    G.bsx = bsx; G.bsy = bsy; G.nbx = nbx; G.nby = nby;
    G.ilo = d.nghost + 1;
    G.wrapx = wrapx ? 1 : 0;
    M.nel = (size_t)G.ldx * G.rows;
    if (M.nel * 8 >= (1ull << 32)) { why = "rectangle beyond 32-bit byte offsets"; return false; }
    // segments: about one round of resident waves (256 CUs x 12) so that every wave marches one long segment
    int seglen = env("CICE_EVP_HIP_MARCH_SEG") ? std::atoi(env("CICE_EVP_HIP_MARCH_SEG")) : 0;
    if (seglen <= 0) {
        const int want_seg = std::max(1, 3072 / M.nstrips);
        seglen = std::max(16, (nyr + want_seg - 1) / want_seg);
    }
    M.seglen = std::min(seglen, nyr);
    M.nseg = (nyr + M.seglen - 1) / M.seglen;
    M.nitems = M.nseg * M.nstrips;
    return true;
}

static int march_alloc()
{
    State::March &M = S.march;
    auto A = [&](double *&p) -> int {
        if (p) return 0;
        HIPC(hipMalloc((void **)&p, M.nel * sizeof(double)));
        HIPC(hipMemsetAsync(p, 0, M.nel * sizeof(double), S.stream));
        return 0;
    };
    for (int k = 0; k < 2; ++k) {
        if (A(B.u[k]) || A(B.v[k])) return -1;
        for (auto &p : B.sig[k])
            if (A(p)) return -1;
    }
    for (auto &p : B.stat) if (A(p)) return -1;
    for (auto &p : B.in) if (A(p)) return -1;
    for (auto &p : B.diag) if (A(p)) return -1;
    if (!B.mask) {
        HIPC(hipMalloc((void **)&B.mask, M.nel));
        HIPC(hipMemsetAsync(B.mask, 0, M.nel, S.stream));
    }
    if (!B.bad) HIPC(hipMalloc((void **)&B.bad, sizeof(unsigned)));
    if (!B.blkid) {
        HIPC(hipMalloc((void **)&B.blkid, M.blkid_h.size() * sizeof(int)));
        HIPC(hipMemcpy(B.blkid, M.blkid_h.data(), M.blkid_h.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPC(hipMalloc((void **)&B.org, M.org_h.size() * sizeof(int2)));
        HIPC(hipMemcpy(B.org, M.org_h.data(), M.org_h.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    M.G.blkid = B.blkid;
    M.G.blk_org = B.org;
    M.G.blk = S.blk;
    return 0;
}

static int read_bad(unsigned &bad)
{
    HIPC(hipMemcpyAsync(&bad, B.bad, sizeof bad, hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

// static fields into the rectangle, once; are their ghost values images of one global field?
static int march_statics()
{
    State::March &M = S.march;
    EvpMarchTab T{};
    double *src[5] = {S.stat[0], S.stat[1], S.hte, S.htn, S.stat[9]};
    for (int k = 0; k < 5; ++k) { T.blk[k] = src[k]; T.rect[k] = B.stat[k]; T.rect2[k] = nullptr; }
    T.n = 5;
    evp_launch_march_gather(M.G, T, nullptr, nullptr, S.stream);
    EvpMarchTab C{};
    C.n = 4;               // dxT dyT (fringe) | HTE HTN (fringe + column ilo-1 / row jlo-1)
    for (int k = 0; k < 4; ++k) { C.blk[k] = src[k]; C.rect[k] = B.stat[k]; }
    HIPC(hipMemsetAsync(B.bad, 0, sizeof(unsigned), S.stream));
    evp_launch_march_check(M.G, C, nullptr, nullptr, 0, 2, B.bad, S.stream);
    unsigned bad = 0;
    if (read_bad(bad)) return -1;
    M.stat_ok = bad == 0;
    M.stat_done = true;
    return 0;
}

// Decide once per init (after the first upload: EVP_F_METRICS is known then) whether this rank uses the path.
bool march_wanted()
{
    State::March &M = S.march;
    if (M.mode >= 0) return M.mode == 1;
    M.mode = 0;
    const int want = env("CICE_EVP_HIP_MARCH") ? std::atoi(env("CICE_EVP_HIP_MARCH")) : -1;
    if (want == 0) return false;
    std::string why;
    if (!march_geometry(why)) {
        M.why = why;
        if (env("CICE_EVP_HIP_VERBOSE")) std::fprintf(stderr, "[cice_evp_hip] two-subcycle kernel off: %s\n", why.c_str());
        return false;
    }
    // worth it when the domain is far beyond what stays on the chip (the on-chip resident kernel is chosen before this
    // is asked): from ~1M cells the strips fill the GPU
    if (want < 0 && (long)M.G.nxr * M.G.nyr < 1000000L) { M.why = "domain below 1M cells"; return false; }
    M.mode = 1;
    return true;
}

static void march_args(EvpMarch &A, int cur, int last)
{
    const State::March &M = S.march;
    const cice_evp_hip_params &q = S.prm;
    A.p = {q.arlx1i, q.denom1, q.brlx, q.revp, q.e_factor, q.epp2i, q.capping, q.Ktens, q.u0, q.cosw, q.sinw, q.rhow};
    A.deltaminEVP = q.deltaminEVP;
    A.ldx = M.G.ldx; A.nxr = M.G.nxr; A.nyr = M.G.nyr;
    A.nstrips = M.nstrips; A.nseg = M.nseg; A.seglen = M.seglen; A.nitems = M.nitems;
    A.wrapx = M.G.wrapx;
    A.last = last;
    A.flags = S.flags & S.flags_allowed;
    A.mask = B.mask;
    A.u_in = B.u[cur]; A.v_in = B.v[cur]; A.u_out = B.u[cur ^ 1]; A.v_out = B.v[cur ^ 1];
    for (int k = 0; k < 12; ++k) { A.sig_in[k] = B.sig[cur][k]; A.sig_out[k] = B.sig[cur ^ 1][k]; }
    A.dxT = B.stat[0]; A.dyT = B.stat[1]; A.HTE = B.stat[2]; A.HTN = B.stat[3]; A.uarear = B.stat[4];
    A.strength = B.in[0]; A.vrelfac = B.in[1]; A.uocn = B.in[2]; A.vocn = B.in[3]; A.forcex = B.in[4]; A.forcey = B.in[5];
    A.umassdti = B.in[6]; A.fm = B.in[7]; A.waterx = B.in[8]; A.watery = B.in[9]; A.TbU = B.in[10];
    A.uvel_init = B.in[11]; A.vvel_init = B.in[12];
    A.strintx = B.diag[0]; A.strinty = B.diag[1]; A.taubx = B.diag[2]; A.tauby = B.diag[3];
}

// All ndte subcycles of a call.  Returns 0 when done (S.cur advanced like the one-subcycle loop would), < 0 on error.
int march_run(int ndte)
{
    State::March &M = S.march;
    if (march_alloc()) return -1;
    if (!M.stat_done && march_statics()) return -1;
    int cur = S.cur;
    int left = ndte;
    auto fallback = [&](const char *why) -> int {
        ++M.declined;
        M.why = why;
        if (env("CICE_EVP_HIP_VERBOSE")) std::fprintf(stderr, "[cice_evp_hip] two-subcycle kernel declined this call: %s\n", why);
        if (left > 0)
            if (int rc = enqueue_loop(left, cur)) return rc;
        S.cur = cur ^ (left & 1);
        return 0;
    };
    if (!M.stat_ok) return fallback("ghost values of the static grid fields are not images of one global field");
    if (left & 1) {          // the odd one first, in the block layout
        if (int rc = enqueue_loop(1, cur)) return rc;
        cur ^= 1;
        --left;
    }
    if (left == 0) { S.cur = cur; return 0; }
    const unsigned fl = S.flags & S.flags_allowed;
    // ---- gather the state and the per-call inputs ----
    {
        EvpMarchTab T{};
        auto add = [&](double *blk, double *rect, double *rect2) { T.blk[T.n] = blk; T.rect[T.n] = rect; T.rect2[T.n] = rect2; ++T.n; };
        add(S.u[cur], B.u[0], B.u[1]);
        add(S.v[cur], B.v[0], B.v[1]);
        for (int k = 0; k < 12; ++k) add(S.sig[cur][k], B.sig[0][k], B.sig[1][k]);
        add(S.in[F_STRENGTH], B.in[0], nullptr);
        add(S.vrelfac, B.in[1], nullptr);
        add(S.in[F_UOCN], B.in[2], nullptr); add(S.in[F_VOCN], B.in[3], nullptr);
        add(S.in[F_FORCEX], B.in[4], nullptr); add(S.in[F_FORCEY], B.in[5], nullptr);
        add(S.in[F_UMASSDTI], B.in[6], nullptr); add(S.in[F_FM], B.in[7], nullptr);
        if (!(fl & EVP_F_WATER_IS_OCN)) { add(S.in[F_WATERX], B.in[8], nullptr); add(S.in[F_WATERY], B.in[9], nullptr); }
        if (!(fl & EVP_F_TBU_ZERO)) add(S.in[F_TBU], B.in[10], nullptr);
        if (S.prm.revp != 0.0) { add(S.in[F_UVEL_INIT], B.in[11], nullptr); add(S.in[F_VVEL_INIT], B.in[12], nullptr); }
        evp_launch_march_gather(M.G, T, S.mask, B.mask, S.stream);
    }
    if (M.checked_seq != S.upload_seq) {
        // first call on this uploaded state: are the caller's ghost values images of one global state?
        EvpMarchTab C{};
        C.blk[0] = S.u[cur]; C.rect[0] = B.u[0];
        C.blk[1] = S.v[cur]; C.rect[1] = B.v[0];
        for (int k = 0; k < 12; ++k) { C.blk[2 + k] = S.sig[cur][k]; C.rect[2 + k] = B.sig[0][k]; }
        C.blk[14] = S.in[F_STRENGTH]; C.rect[14] = B.in[0];
        C.n = 15;
        HIPC(hipMemsetAsync(B.bad, 0, sizeof(unsigned), S.stream));
        evp_launch_march_check(M.G, C, S.mask, B.mask, 2, 13, B.bad, S.stream);
        unsigned bad = 0;
        if (read_bad(bad)) return -1;
        if (bad) return fallback("ghost cells of the uploaded state are not images of one global state");
        M.checked_seq = S.upload_seq;
    }
    // ---- the passes ----
    const int npass = left / 2;
    int rc = 0;
    for (int k = 0; k < npass; ++k) {
        EvpMarch A;
        march_args(A, rc, k == npass - 1);
        evp_launch_march(A, S.prm.strict != 0, cap_mode(), S.stream);
        rc ^= 1;
    }
    HIPC(hipGetLastError());
    M.passes += npass;
    // ---- back to the block layout ----
    {
        EvpMarchTab T{};
        auto add = [&](double *blk, double *rect) { T.blk[T.n] = blk; T.rect[T.n] = rect; ++T.n; };
        add(S.u[cur], B.u[rc]); add(S.v[cur], B.v[rc]);
        for (int k = 0; k < 12; ++k) add(S.sig[cur][k], B.sig[rc][k]);
        add(S.in[F_STRINTX], B.diag[0]); add(S.in[F_STRINTY], B.diag[1]);
        add(S.in[F_TAUBX], B.diag[2]); add(S.in[F_TAUBY], B.diag[3]);
        evp_launch_march_scatter(M.G, T, S.mask, 2, 12, S.stream);
    }
    HIPC(hipGetLastError());
    S.cur = cur;            // an even number of subcycles later: same ping-pong buffer of the block layout
    return 0;
}

}  // namespace evp_host
