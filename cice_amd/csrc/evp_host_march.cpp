// =====================================================================
// Host side of the several-subcycles-per-pass kernel (evp_march.hip): the device-private rectangle layout, the
// conversions between it and the CICE block layout, and the loop.
//
// Strip-major layout: all blocks of the rank are assembled into ONE rectangle of nxr x nyr cells, cut into strips of
// `own` <= 56 columns; per (row, strip) one contiguous block [field][64 lanes] -- with P = EVP_MARCH_PAD = 4, lanes
// P .. P+own-1 are the strip's own columns, the others duplicate its neighbours' (or, beyond a closed side, hold the
// caller's ghost values / zeros).
// Buffers: state (u, v, 12 stresses; two copies), constants of a call (13 fields), optional operands (5), diagnostics
// (4); rows -P .. nyr+P+2 (P halo rows below, P above, two spare rows the prefetch may touch, one dump row).  Every
// cell of the rank exists once: the reference's redundant copies (ghost cells, the T-cells of the north / east fringe
// every block computes for itself, ice_dyn_shared.F90:740-749) are images of it.  The byte mask stays row-major.
//
// A call of cice_evp_hip_subcycle(ndte) through this path:
//   [ndte = 4q + 1: one subcycle with the one-subcycle kernel]  gather -> consistency check (first call after an upload)
//   -> q passes of four subcycles and one of the remaining two or three (ping-pong between two sets of u, v, 12
//   stresses) -> scatter.
// Not eligible (the one-subcycle kernels keep running): tripole / cyclic north-south boundary, blocks that do not tile a
// rectangle per rank (eliminated land blocks), metric terms handed over as arrays (tripole), a rank too thin next to a
// closed boundary for its redundant rim (march_plan.cpp), a caller whose ghost values are not images of one global
// state.  Several ranks: the ring of ext + P cells travels over RCCL send / recv once per ext + P subcycles (section 6).
// =====================================================================
#include "evp_host.h"
#include "march_plan.h"

static_assert(MARCH_PLAN_PAD == EVP_MARCH_PAD, "march_plan.h and evp_device.h must agree on the width of the overlap");

namespace evp_host {

namespace {

struct MarchBuf {
    double *st[2] = {};          // state blocks: u v sig x 12
    double *cst = nullptr;       // dxT dyT strength HTE HTN vrelfac uocn vocn forcex forcey umassdti fm uarear
    double *opt = nullptr;       // waterx watery TbU uvel_init vvel_init
    double *diag = nullptr;      // strintx strinty taubx tauby
    uint8_t *mask = nullptr;     // row-major
    unsigned *bad = nullptr, *dup = nullptr;
    int *blkid = nullptr;
    int2 *org = nullptr;
    // the two-cell ring between ranks
    int *send_pos = nullptr, *recv_pos1 = nullptr, *recv_pos2 = nullptr, *send_midx = nullptr, *recv_midx = nullptr;
    double *sendbuf = nullptr, *recvbuf = nullptr;
    // the early launch of an exchange pass: work items that cover every cell other ranks receive from this one
    int4 *band_items = nullptr, *rest_items = nullptr;      // ... and every other (strip, row) of the rectangle
    int nband = 0, nrest = 0;
    hipEvent_t ev_in = nullptr, ev_main = nullptr, ev_done = nullptr;
    // the same ring as stores into the peers' HIP-IPC-mapped inboxes (march_direct_setup)
    void *dx_area = nullptr;     // fine-grained: flag lines | count | err | inbox [2][n_recv * NF]
    std::vector<void *> dx_mapped;
    EvpMarchDirect dx{};
    unsigned dx_seq = 0;
    EvpRingCuts cut_send{}, cut_recv{};      // where each neighbour's block begins in the send / receive list
};
MarchPlan PL;
// slots of the constants block (evp_march.hip: C_*), of the optional block (O_*)
enum { C_DXT = 0, C_DYT, C_STRENGTH, C_HTE, C_HTN, C_VRELFAC, C_UOCN, C_VOCN, C_FORCEX, C_FORCEY, C_UMASSDTI, C_FM, C_UAREAR };
enum { O_WATERX = 0, O_WATERY, O_TBU, O_UINIT, O_VINIT };
MarchBuf B;

template <class T> void F(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

}  // namespace

void march_free()
{
    F(B.st[0]); F(B.st[1]); F(B.cst); F(B.opt); F(B.diag);
    F(B.mask); F(B.bad); F(B.dup); F(B.blkid); F(B.org);
    F(B.send_pos); F(B.recv_pos1); F(B.recv_pos2); F(B.send_midx); F(B.recv_midx); F(B.sendbuf); F(B.recvbuf);
    F(B.band_items); F(B.rest_items);
    B.nband = B.nrest = 0;
    for (void *m : B.dx_mapped) (void)hipIpcCloseMemHandle(m);
    B.dx_mapped.clear();
    F(B.dx_area);
    B.dx = EvpMarchDirect{};
    B.dx_seq = 0;
    for (hipEvent_t *e : {&B.ev_in, &B.ev_main, &B.ev_done}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    PL = MarchPlan();
    S.march = State::March{};
}

// Can this rank's sub-domain be held as one rectangle?  Fills S.march.G (host fields) when it can.
static bool march_geometry(std::string &why)
{
    State::March &M = S.march;
    const cice_evp_hip_dims &d = S.d;
    // the rectangles of all ranks, this rank's strips and the exchange lists: the same verdict on every rank
    const int own_max = env_test("CICE_EVP_HIP_MARCH_OWN") ? std::atoi(env_test("CICE_EVP_HIP_MARCH_OWN")) : EVP_MARCH_OWN;
    const bool wrap_inside = !(env_test("CICE_EVP_HIP_MARCH_SELFX") && std::atoi(env_test("CICE_EVP_HIP_MARCH_SELFX")));
    // cells a rank holds beyond its own on every side with a neighbour: the ring of ext + P cells is then exchanged every
    // (ext + P)-th subcycle only (march_plan.h); 4 = after every second pass of four.  (Rounds 4-5, two subcycles per pass and a
    // two-cell ring, 3600 x 2400 as 4x2 pieces, one-GPU rehearsal: 54.3 / 52.3 / 52.0 / 51.2 us per subcycle with ext 0 / 2 / 4 / 6
    // against 47.0 without any exchange.)
    int ext = env_test("CICE_EVP_HIP_MARCH_EXT") ? std::max(0, std::atoi(env_test("CICE_EVP_HIP_MARCH_EXT")) & ~1) : -1;
    if (ext >= 0) {
        if (!build_march_plan(d, own_max, wrap_inside, ext, PL)) { why = PL.error; return false; }
    } else {
        // Default (round 6): the widest rim of 12 / 8 / 4 cells that costs no rank a strip more than a rim of 4 does and at most 2 %
        // more rows -- the exchange then comes every 16th / 12th / 8th subcycle for the same bytes per subcycle.  (The 8 x 1 pieces of
        // 3600 x 2400: 450 + 2 x 12 = 474 columns still fit the 9 strips that 458 need; measured on one GPU, ring exchanged with the
        // rank itself, an exchange costs ~38 us: 4.8 us per subcycle at every 8th.)  Every rank reaches the same value from the global
        // block table; a layout the plan refuses (a rank too thin next to a closed boundary) tries the next narrower rim.
        bool ok = false;
        std::string first_error;
        for (int e : {12, 8, 4, 0}) {
            if (!build_march_plan(d, own_max, wrap_inside, e, PL)) {
                if (first_error.empty()) first_error = PL.error;
                continue;
            }
            bool fits = true;
            const bool ew_cyc = d.ew_boundary_type == CICE_EVP_BND_CYCLIC;
            for (const MarchRect &R : PL.all) {
                if (!R.ok || e <= 4) continue;
                const bool span = wrap_inside && ew_cyc && R.nxr == d.nx_global;      // wraps inside: no rim in x
                const int w = (R.gx0 > 0 || ew_cyc) ? 1 : 0, ea = (R.gx0 + R.nxr < d.nx_global || ew_cyc) ? 1 : 0;
                const int sn = R.gy0 > 0 ? 1 : 0, nn = R.gy0 + R.nyr < d.ny_global ? 1 : 0;
                const int own_eff = std::min(own_max > 0 ? own_max : EVP_MARCH_OWN, EVP_MARCH_OWN);
                const int w4 = R.nxr + (span ? 0 : 4 * (w + ea)), we = R.nxr + (span ? 0 : e * (w + ea));
                if ((we + own_eff - 1) / own_eff > (w4 + own_eff - 1) / own_eff) fits = false;
                if ((long)(R.nyr + e * (sn + nn)) * 100 > (long)(R.nyr + 4 * (sn + nn)) * 102) fits = false;
            }
            if (fits) { ext = e; ok = true; break; }
        }
        if (!ok) { why = first_error.empty() ? PL.error : first_error; return false; }
    }
    M.ring_valid = ext + EVP_MARCH_PAD;
    // Subcycles per pass.  Round 6 measured what binds the kernel: a wave issues one instruction per 2.4 ns whatever it is, and a
    // row costs ~640 instructions per level plus ~350 that do not depend on the number of levels (loads, stores, addressing, the
    // register pipelines); a segment of sl rows marches sl + 2K - 1, the warm-up rows with part of the levels idle.  Measured on
    // the 8 x 1 and 4 x 2 pieces of 3600 x 2400 (22- and 20-row segments, profiles/r06_ring_rank.txt): 58.0 / 50.0 / 49.1 and
    // 54.9 / 46.8 / 46.0 us per subcycle with two / three / four subcycles per pass; 3600 x 2400 itself (160 rows) 362 / 315 / 293.
    // Four from 12 rows per segment, three from 8, two below.  Every rank must run the same passes (the ring is exchanged between
    // them), so the rule looks at the SHORTEST segment of any rank, which every rank works out from the global block table alone.
    // The test build can ask for a value.
    {
        int slmin = 1 << 30;
        for (const MarchRect &R : PL.all) {
            if (!R.ok) continue;
            const int grow = d.nranks > 1 ? 2 * ext : 0;
            const int ns = (R.nxr + grow + EVP_MARCH_OWN - 1) / EVP_MARCH_OWN;
            const int nseg = std::max(1, 1024 / ns);
            slmin = std::min(slmin, std::max(6, (R.nyr + grow + nseg - 1) / nseg));
        }
        M.kpass = std::min(EVP_MARCH_KMAX, slmin >= 12 ? 4 : slmin >= 8 ? 3 : 2);
    }
    if (env_test("CICE_EVP_HIP_MARCH_K")) M.kpass = std::min(EVP_MARCH_KMAX, std::max(2, std::atoi(env_test("CICE_EVP_HIP_MARCH_K"))));
    if (PL.peers.size() > (size_t)EVP_MARCH_DIRECT_MAXPEER) { why = "more ring neighbours than the exchange lists hold"; return false; }
    if (!PL.peers.empty() && !S.have_comm && !S.test_xchg) { why = "cells of other ranks needed but no RCCL communicator (cice_evp_hip_comm_init)"; return false; }
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
    if (nxr != PL.owned.nxr || nyr != PL.owned.nyr || gx0 - 1 != PL.owned.gx0 || gy0 - 1 != PL.owned.gy0) { why = "local blocks disagree with the global block table"; return false; }
    for (auto &o : M.org_h) { o.x += PL.ext_w; o.y += PL.ext_s; }        // block origins in the rectangle the rank HOLDS
    const bool wrapx = PL.wrapx;
    EvpMarchGeo &G = M.G;
    G.nxr = PL.me.nxr; G.nyr = PL.me.nyr;                                  // held: own cells + the redundant rim
    G.ext_w = PL.ext_w; G.ext_s = PL.ext_s; G.nxo = nxr; G.nyo = nyr;
    G.nxb = d.nx_block; G.nyb = d.ny_block; G.plane = (int)S.plane; G.nblocks = d.nblocks;
    G.bsx = bsx; G.bsy = bsy; G.nbx = nbx; G.nby = nby;
    G.ilo = d.nghost + 1;
    G.wrapx = wrapx ? 1 : 0;
    G.gx0 = PL.me.gx0; G.gy0 = PL.me.gy0; G.nxg = d.nx_global; G.nyg = d.ny_global;
    G.ew_cyclic = d.ew_boundary_type == CICE_EVP_BND_CYCLIC ? 1 : 0;
    G.own = PL.me.own; G.nstrips = PL.me.nstrips;
    M.dup_h.assign(PL.dup.size(), EVP_MARCH_NODUP);
    for (size_t k = 0; k < PL.dup.size(); ++k)
        if (PL.dup[k] >= 0) M.dup_h[k] = (unsigned)((((size_t)(PL.dup[k] >> 8)) * EVP_MARCH_S_NF * 64 + (PL.dup[k] & 255)) * 8);
    M.nstrips = G.nstrips;
    G.ldx = ((G.nstrips * G.own + 64 + 2 * EVP_MARCH_PAD + 7) / 8) * 8;
    G.rows = G.nyr + 2 * EVP_MARCH_PAD + 3;      // y = -P .. nyr+P+2: halo, two rows the prefetch may touch, the dump row
    M.nblk = (size_t)G.rows * G.nstrips;
    if (M.nblk * EVP_MARCH_S_NF * 512 >= (1ull << 32)) { why = "state buffer beyond 32-bit byte offsets"; return false; }
    // segments: one wave per SIMD (1024 of them), all resident at once -- measured at 3600 x 2400: 16 segments (960
    // waves) 318 us per subcycle, 24 (1440: some SIMDs get two) 400, 34 (2040: two each) 378, 12 (720) 372
    int seglen = env_test("CICE_EVP_HIP_MARCH_SEG") ? std::atoi(env_test("CICE_EVP_HIP_MARCH_SEG")) : 0;
    if (seglen <= 0) {
        // (medium domains, 0.45M .. 1M cells: shorter segments keep the count near 1000 -- 1080 x 720: 14-row segments 39 us
        // per subcycle against 53 for the one-subcycle kernel; each segment recomputes ~3.5 rows of warm-up)
        // (as many segments as fit 1024 waves: 3600 x 2400 has 60 strips -- 17 segments = 1020 waves = 255 workgroups, one per CU,
        // 302.8 us per subcycle; 16 segments (960 waves, 16 CUs idle) 310.1; 18 (1080: some SIMDs get two) 472 -- round 5, same box)
        const int want_seg = std::max(1, 1024 / M.nstrips);
        seglen = std::max(6, (G.nyr + want_seg - 1) / want_seg);
    }
    M.seglen = std::min(seglen, G.nyr);
    M.nseg = (G.nyr + M.seglen - 1) / M.seglen;
    M.nitems = M.nseg * M.nstrips;
    return true;
}

static int march_alloc()
{
    State::March &M = S.march;
    auto A = [&](double *&p, int nf) -> int {
        if (p) return 0;
        const size_t bytes = M.nblk * (size_t)nf * 512;
        HIPC(hipMalloc((void **)&p, bytes));
        HIPC(hipMemsetAsync(p, 0, bytes, S.stream));
        return 0;
    };
    if (A(B.st[0], EVP_MARCH_S_NF) || A(B.st[1], EVP_MARCH_S_NF) || A(B.cst, EVP_MARCH_C_NF) || A(B.opt, EVP_MARCH_O_NF) ||
        A(B.diag, EVP_MARCH_D_NF)) return -1;
    if (!B.mask) {
        HIPC(hipMalloc((void **)&B.mask, (size_t)M.G.rows * M.G.ldx));
        HIPC(hipMemsetAsync(B.mask, 0, (size_t)M.G.rows * M.G.ldx, S.stream));
    }
    if (!B.bad) HIPC(hipMalloc((void **)&B.bad, sizeof(unsigned)));
    if (!B.blkid) {
        HIPC(hipMalloc((void **)&B.blkid, M.blkid_h.size() * sizeof(int)));
        HIPC(hipMemcpy(B.blkid, M.blkid_h.data(), M.blkid_h.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPC(hipMalloc((void **)&B.org, M.org_h.size() * sizeof(int2)));
        HIPC(hipMemcpy(B.org, M.org_h.data(), M.org_h.size() * sizeof(int2), hipMemcpyHostToDevice));
        HIPC(hipMalloc((void **)&B.dup, M.dup_h.size() * sizeof(unsigned)));
        HIPC(hipMemcpy(B.dup, M.dup_h.data(), M.dup_h.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    if (PL.n_send + PL.n_recv > 0 && !B.sendbuf) {
        std::vector<int> sp, r1, r2, sm, rm;
        for (const MarchPeer &p : PL.peers) {
            sp.insert(sp.end(), p.send_pos.begin(), p.send_pos.end());
            r1.insert(r1.end(), p.recv_pos1.begin(), p.recv_pos1.end());
            r2.insert(r2.end(), p.recv_pos2.begin(), p.recv_pos2.end());
            for (size_t k = 0; k < p.send_pos.size(); ++k)
                sm.push_back((int)((size_t)((p.send_pos[k] >> 6) / M.G.nstrips) * M.G.ldx + EVP_MARCH_PAD + p.send_col[k]));
            for (size_t k = 0; k < p.recv_pos1.size(); ++k)
                rm.push_back((int)((size_t)p.recv_row[k] * M.G.ldx + EVP_MARCH_PAD + p.recv_col[k]));
        }
        auto up = [&](int *&dp, const std::vector<int> &v) -> int {
            HIPC(hipMalloc((void **)&dp, std::max<size_t>(v.size(), 1) * sizeof(int)));
            if (!v.empty()) HIPC(hipMemcpy(dp, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
            return 0;
        };
        if (up(B.send_pos, sp) || up(B.recv_pos1, r1) || up(B.recv_pos2, r2) || up(B.send_midx, sm) || up(B.recv_midx, rm)) return -1;
        B.cut_send.n = B.cut_recv.n = (int)PL.peers.size();
        int cs = 0, cr = 0;
        for (size_t q = 0; q < PL.peers.size(); ++q) {
            B.cut_send.start[q] = cs; B.cut_recv.start[q] = cr;
            cs += (int)PL.peers[q].send_pos.size(); cr += (int)PL.peers[q].recv_pos1.size();
        }
        B.cut_send.start[PL.peers.size()] = cs; B.cut_recv.start[PL.peers.size()] = cr;
        HIPC(hipMalloc((void **)&B.sendbuf, std::max<size_t>(PL.n_send, 1) * EVP_MARCH_S_NF * sizeof(double)));
        HIPC(hipMalloc((void **)&B.recvbuf, std::max<size_t>(PL.n_recv, 1) * EVP_MARCH_S_NF * sizeof(double)));
        // Work items of the EARLY launch of an exchange pass (march_run): per strip the rows that hold cells some other rank
        // receives, cut into short segments -- from the send lists themselves, so every sent cell is covered whatever the
        // layout.  Short segments (a third of the regular length, at least 6 rows) so that the launch, the pack and the
        // transfer end before the pass they overlap with does; each costs 2K - 1 rows of warm-up like any segment.  The pass
        // itself then runs over every OTHER (strip, row) of the rectangle (rest_items): the two launches of an exchange pass
        // store into disjoint cells (round-4 advice: they used to overlap on the band, storing the same bits twice).
        const int nstr = M.G.nstrips;
        std::vector<std::vector<char>> rows((size_t)nstr, std::vector<char>((size_t)M.G.nyr, 0));
        for (int pos : sp) {
            const int blk = pos >> 6, strip = blk % nstr, y = blk / nstr - EVP_MARCH_PAD;
            if (y >= 0 && y < M.G.nyr) rows[(size_t)strip][(size_t)y] = 1;
        }
        int bseg = env_test("CICE_EVP_HIP_MARCH_BANDSEG") ? std::atoi(env_test("CICE_EVP_HIP_MARCH_BANDSEG")) : 0;
        if (bseg <= 0) bseg = std::max(6, M.seglen / 3);
        std::vector<int4> items;
        for (int st = 0; st < nstr; ++st)
            for (int y = 0; y < M.G.nyr;) {
                if (!rows[(size_t)st][(size_t)y]) { ++y; continue; }
                int y1 = y;
                while (y1 < M.G.nyr && y1 - y < bseg && rows[(size_t)st][(size_t)y1]) ++y1;
                items.push_back(make_int4(st, y, y1, 0));
                y = y1;
            }
        B.nband = (int)items.size();
        if (B.nband > 0) {
            HIPC(hipMalloc((void **)&B.band_items, items.size() * sizeof(int4)));
            HIPC(hipMemcpy(B.band_items, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice));
        }
        std::vector<int4> rest;
        for (int st = 0; st < nstr; ++st)
            for (int y = 0; y < M.G.nyr;) {
                if (rows[(size_t)st][(size_t)y]) { ++y; continue; }
                int y1 = y;
                while (y1 < M.G.nyr && y1 - y < M.seglen && !rows[(size_t)st][(size_t)y1]) ++y1;
                rest.push_back(make_int4(st, y, y1, 0));
                y = y1;
            }
        B.nrest = (int)rest.size();
        if (B.nrest > 0) {
            HIPC(hipMalloc((void **)&B.rest_items, rest.size() * sizeof(int4)));
            HIPC(hipMemcpy(B.rest_items, rest.data(), rest.size() * sizeof(int4), hipMemcpyHostToDevice));
        }
        for (hipEvent_t *e : {&B.ev_in, &B.ev_main, &B.ev_done})
            if (!*e) HIPC(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    M.G.blkid = B.blkid;
    M.G.blk_org = B.org;
    M.G.blk = S.blk;
    return 0;
}

// The two-cell ring of one strip-major buffer (all nf fields of every halo cell) from the ranks that own the cells:
// pack -> ncclGroupStart{ncclSend, ncclRecv per neighbour}ncclGroupEnd -> unpack (incl. the duplicates), on the
// library's stream.  Once per PASS of two subcycles for the state, once per call for the constants and the mask.
// (test hook) the same exchange through host buffers and the caller's callback
static int hook_exchange(int nf, hipStream_t st)
{
    std::vector<int32_t> ranks;
    std::vector<int64_t> ns, nr;
    for (const MarchPeer &p : PL.peers) {
        ranks.push_back(p.rank);
        ns.push_back((int64_t)p.send_pos.size() * nf);
        nr.push_back((int64_t)p.recv_pos1.size() * nf);
    }
    S.test_send.resize((size_t)PL.n_send * nf + 1);
    S.test_recv.resize((size_t)PL.n_recv * nf + 1);
    HIPC(hipMemcpyAsync(S.test_send.data(), B.sendbuf, (size_t)PL.n_send * nf * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (S.test_xchg(S.test_user, (int32_t)ranks.size(), ranks.data(), ns.data(), nr.data(), S.test_send.data(), S.test_recv.data()))
        return fail(-2, "test transport: the exchange callback failed");
    HIPC(hipMemcpyAsync(B.recvbuf, S.test_recv.data(), (size_t)PL.n_recv * nf * sizeof(double), hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return 0;
}

// pack + transfer of the ring of `buf` on stream `st` (what is left is the unpack)
static int march_send_recv(double *buf, int nf, hipStream_t st)
{
    evp_launch_march_pack(buf, nf, B.send_pos, PL.n_send, B.cut_send, B.sendbuf, st);
    if (S.test_xchg) return hook_exchange(nf, st);
    size_t so = 0, ro = 0;
    NCCLC(ncclGroupStart());
    for (const MarchPeer &p : PL.peers) {
        const size_t ns = p.send_pos.size(), nr = p.recv_pos1.size();
        if (ns) NCCLC(ncclSend(B.sendbuf + so * nf, ns * nf, ncclDouble, p.rank, S.comm, st));
        if (nr) NCCLC(ncclRecv(B.recvbuf + ro * nf, nr * nf, ncclDouble, p.rank, S.comm, st));
        so += ns; ro += nr;
    }
    NCCLC(ncclGroupEnd());
    return 0;
}

static int agree_max(unsigned &v);

// ---- the ring without a communication library ------------------------------------------------------------------------
// What a rank tells each of its ring neighbours: how to map its inbox and where that neighbour's entries land in it.
struct MarchBlob {
    uint32_t magic, nf;
    int32_t from, to;
    uint64_t host_id;
    int64_t pid;
    uint64_t base, inbox_off, inbox_pstride, flag_off;      // bytes
    int64_t off_cells, count;                               // where the receiver of this blob writes, how many cells are expected
    hipIpcMemHandle_t handle;
};
constexpr int MARCH_BLOB_DOUBLES = 32;
static_assert(sizeof(MarchBlob) <= MARCH_BLOB_DOUBLES * sizeof(double), "MarchBlob must fit its slot");
constexpr uint32_t MARCH_BLOB_MAGIC = 0x4d524348u;      // "MRCH"

// Collective over the ranks of the ring (every rank with march neighbours runs it in the same call): inbox + flags in
// fine-grained memory, one blob per neighbour through the transport that is there anyway (RCCL send / recv, or the test
// hook), map, vote.  Any rank that cannot (another host, no IPC, switched off) makes everybody stay with the library.
static int march_direct_setup()
{
    State::March &M = S.march;
    M.direct = 0;
    const int np = (int)PL.peers.size();
    // Opt-in (CICE_EVP_HIP_MARCH_DIRECT=1), and only if every rank asks: on one GPU (the ring exchanged with the rank itself,
    // 450 x 2400 and 900 x 1200 pieces) it is no faster than RCCL -- 54.7 against 52.6 and 52.7 against 52.1 us per subcycle: the
    // pack kernel's uncached 8-byte stores take as long (13.4 us) as RCCL's pack + copy kernel (6.4 + 8.3) -- and what it does over
    // xGMI has never been measured; bench.py --gpus N times the 3600 x 2400 block this way too.
    {
        const bool want = env_test("CICE_EVP_HIP_MARCH_DIRECT") && std::atoi(env_test("CICE_EVP_HIP_MARCH_DIRECT")) &&
                          !(env("CICE_EVP_HIP_HALO") && !std::strcmp(env("CICE_EVP_HIP_HALO"), "rccl"));
        unsigned no = want ? 0u : 1u;
        HIPC(hipMemcpyAsync(B.bad, &no, sizeof no, hipMemcpyHostToDevice, S.stream));
        if (agree_max(no)) return -1;
        if (no) {
            M.direct_why = want ? "not asked for on every rank" : "not asked for (CICE_EVP_HIP_MARCH_DIRECT=1)";
            return 0;
        }
    }
    bool ok = np > 0 && np <= EVP_MARCH_DIRECT_MAXPEER;
    if (!ok) M.direct_why = "more ring neighbours than the direct exchange holds";
    const size_t flag_bytes = (size_t)EVP_MARCH_DIRECT_MAXPEER * 64, count_off = flag_bytes, err_off = flag_bytes + 64;
    const size_t inbox_off = flag_bytes + 128;
    const size_t par_doubles = ((size_t)std::max(PL.n_recv, 1) * EVP_MARCH_S_NF + 31) & ~(size_t)31;
    std::vector<double> out((size_t)np * MARCH_BLOB_DOUBLES, 0.0), in((size_t)np * MARCH_BLOB_DOUBLES, 0.0);
    if (ok) {
        const size_t bytes = inbox_off + 2 * par_doubles * sizeof(double);
        if (hipExtMallocWithFlags(&B.dx_area, bytes, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(B.dx_area, 0, bytes) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
            M.direct_why = "no fine-grained device memory for the inbox";
        }
    }
    hipIpcMemHandle_t handle{};
    if (ok && hipIpcGetMemHandle(&handle, B.dx_area) != hipSuccess) {
        (void)hipGetLastError();
        ok = false;
        M.direct_why = "hipIpcGetMemHandle failed";
    }
    size_t ro = 0;
    for (int q = 0; q < np; ++q) {
        MarchBlob b{};
        b.magic = ok ? MARCH_BLOB_MAGIC : 0u;
        b.nf = EVP_MARCH_S_NF;
        b.from = S.d.rank; b.to = PL.peers[q].rank;
        b.host_id = host_identity();
        b.pid = (int64_t)getpid();
        b.base = (uint64_t)(uintptr_t)B.dx_area;
        b.inbox_off = inbox_off; b.inbox_pstride = par_doubles * sizeof(double); b.flag_off = (size_t)q * 64;
        b.off_cells = (int64_t)ro; b.count = (int64_t)PL.peers[q].recv_pos1.size();
        b.handle = handle;
        std::memcpy(&out[(size_t)q * MARCH_BLOB_DOUBLES], &b, sizeof b);
        ro += PL.peers[q].recv_pos1.size();
    }
    // the blobs travel like a ring of 32 doubles per neighbour
    if (S.test_xchg) {
        std::vector<int32_t> ranks;
        std::vector<int64_t> cnt((size_t)np, MARCH_BLOB_DOUBLES);
        for (const MarchPeer &p : PL.peers) ranks.push_back(p.rank);
        if (S.test_xchg(S.test_user, np, ranks.data(), cnt.data(), cnt.data(), out.data(), in.data()))
            return fail(-2, "test transport: the exchange callback failed");
    } else {
        double *d_out = nullptr, *d_in = nullptr;
        const size_t nb = (size_t)np * MARCH_BLOB_DOUBLES * sizeof(double);
        HIPC(hipMalloc((void **)&d_out, nb));
        HIPC(hipMalloc((void **)&d_in, nb));
        HIPC(hipMemcpyAsync(d_out, out.data(), nb, hipMemcpyHostToDevice, S.stream));
        NCCLC(ncclGroupStart());
        for (int q = 0; q < np; ++q) {
            NCCLC(ncclSend(d_out + (size_t)q * MARCH_BLOB_DOUBLES, MARCH_BLOB_DOUBLES, ncclDouble, PL.peers[q].rank, S.comm, S.stream));
            NCCLC(ncclRecv(d_in + (size_t)q * MARCH_BLOB_DOUBLES, MARCH_BLOB_DOUBLES, ncclDouble, PL.peers[q].rank, S.comm, S.stream));
        }
        NCCLC(ncclGroupEnd());
        HIPC(hipMemcpyAsync(in.data(), d_in, nb, hipMemcpyDeviceToHost, S.stream));
        HIPC(hipStreamSynchronize(S.stream));
        (void)hipFree(d_out); (void)hipFree(d_in);
    }
    EvpMarchDirect D{};
    D.npeers = np;
    for (int q = 0; q < np && ok; ++q) {
        MarchBlob b;
        std::memcpy(&b, &in[(size_t)q * MARCH_BLOB_DOUBLES], sizeof b);
        const MarchPeer &p = PL.peers[q];
        if (b.magic != MARCH_BLOB_MAGIC) { ok = false; M.direct_why = "rank " + std::to_string(p.rank) + " cannot take part"; break; }
        if (b.from != p.rank || b.to != S.d.rank || b.nf != EVP_MARCH_S_NF || b.count != (int64_t)p.send_pos.size()) {
            ok = false;
            M.direct_why = "rank " + std::to_string(p.rank) + " expects another ring from this rank than the plan sends";
            break;
        }
        if (b.host_id != host_identity()) { ok = false; M.direct_why = "rank " + std::to_string(p.rank) + " is on another host"; break; }
        char *base = nullptr;
        if (b.pid == (int64_t)getpid()) base = (char *)(uintptr_t)b.base;       // this very process (the ring exchanged with oneself)
        else {
            void *ptr = nullptr;
            if (hipIpcOpenMemHandle(&ptr, b.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError();
                ok = false;
                M.direct_why = "hipIpcOpenMemHandle failed for rank " + std::to_string(p.rank);
                break;
            }
            B.dx_mapped.push_back(ptr);
            base = (char *)ptr;
        }
        D.dst[q] = (double *)(base + b.inbox_off) + (size_t)b.off_cells * EVP_MARCH_S_NF;
        D.dst_pstride[q] = (size_t)(b.inbox_pstride / sizeof(double));
        D.peer_flag[q] = (unsigned *)(base + b.flag_off);
    }
    if (ok) {
        char *mine = (char *)B.dx_area;
        D.flags_in = (unsigned *)mine;
        D.count = (unsigned *)(mine + count_off);
        D.err = (int *)(mine + err_off);
        D.inbox = (const double *)(mine + inbox_off);
        D.inbox_pstride = par_doubles;
        const double tmo_ms = env("CICE_EVP_HIP_HALO_TIMEOUT_MS") ? std::atof(env("CICE_EVP_HIP_HALO_TIMEOUT_MS")) : 30000.0;
        D.timeout_ticks = (unsigned long long)(tmo_ms * 1.0e5);
        B.dx = D;
    }
    // everybody or nobody
    unsigned vote = ok ? 0u : 1u;
    HIPC(hipMemcpyAsync(B.bad, &vote, sizeof vote, hipMemcpyHostToDevice, S.stream));
    if (agree_max(vote)) return -1;
    if (vote && ok) M.direct_why = "another rank cannot take part";
    M.direct = vote ? 0 : 2;         // 2: the first exchange runs both ways and compares
    if (env("CICE_EVP_HIP_VERBOSE"))
        std::fprintf(stderr, "[cice_evp_hip] rank %d: ring of the marching path %s%s\n", (int)S.d.rank,
                     M.direct ? "as stores into the neighbours' HIP-IPC-mapped inboxes" : "through RCCL send / recv: ",
                     M.direct ? "" : M.direct_why.c_str());
    return 0;
}

int march_direct_error()
{
    if (S.march.direct <= 0 || !B.dx.err) return 0;
    int e = 0;
    HIPC(hipMemcpy(&e, B.dx.err, sizeof e, hipMemcpyDeviceToHost));
    if (e) return fail(-8, "marching path: ring neighbour rank %d never signalled within the time-out (CICE_EVP_HIP_HALO_TIMEOUT_MS)",
                       (e - 1 < (int)PL.peers.size()) ? PL.peers[e - 1].rank : -1);
    return 0;
}

static int march_exchange(double *buf, double *buf2, int nf)
{
    if (PL.peers.empty()) return 0;
    State::March &M = S.march;
#ifndef CICE_EVP_HIP_TESTING
    // Product build: the ring goes through RCCL send / recv; its direct-store form is a switch of the test build (until a
    // multi-GPU node has ranked the two), so there is nothing to vote on -- and no collective that a rank without ring
    // neighbours would miss.
    M.direct = 0;
    M.direct_why = "test build only";
#else
    // Test build.  The switch is read at every state exchange and a change re-opens the set-up, which is COLLECTIVE over the
    // ranks that have ring neighbours (blobs, a vote): every such rank must see the same value at the same exchange -- the
    // tests and bench.py's ring_variants set it in all processes alike.  Mutually exclusive with CICE_EVP_HIP_MARCH_OVERLAP
    // (the overlapped path ignores it).
    if (nf == EVP_MARCH_S_NF) {
        const int asked = (env_test("CICE_EVP_HIP_MARCH_DIRECT") && std::atoi(env_test("CICE_EVP_HIP_MARCH_DIRECT"))) ? 1 : 0;
        if (M.direct >= 0 && asked != M.direct_asked) {       // (bench.py times one state both ways: the switch changed between two calls)
            HIPC(hipStreamSynchronize(S.stream));
            for (void *m : B.dx_mapped) (void)hipIpcCloseMemHandle(m);
            B.dx_mapped.clear();
            F(B.dx_area);
            B.dx = EvpMarchDirect{};
            B.dx_seq = 0;
            M.direct = -1;
        }
        M.direct_asked = asked;
        if (M.direct < 0)
            if (int rc = march_direct_setup()) return rc;
    }
#endif
    if (nf == EVP_MARCH_S_NF && M.direct == 1) {
        const unsigned seq = ++B.dx_seq;
        evp_launch_march_pack_direct(buf, nf, B.send_pos, PL.n_send, B.cut_send, B.dx, seq, S.stream);
        evp_launch_march_unpack_direct(buf, buf2, nf, B.recv_pos1, B.recv_pos2, PL.n_recv, B.cut_recv, B.dx, seq, nullptr, nullptr, S.stream);
        return 0;
    }
    if (int rc = march_send_recv(buf, nf, S.stream)) return rc;
    evp_launch_march_unpack(buf, buf2, nf, B.recv_pos1, B.recv_pos2, PL.n_recv, B.cut_recv, B.recvbuf, S.stream);
    if (nf == EVP_MARCH_S_NF && M.direct == 2) {
        // once: the same ring through the inboxes as well, compared bit for bit with what the library delivered
        const unsigned seq = ++B.dx_seq;
        HIPC(hipMemsetAsync(B.bad, 0, sizeof(unsigned), S.stream));
        evp_launch_march_pack_direct(buf, nf, B.send_pos, PL.n_send, B.cut_send, B.dx, seq, S.stream);
        evp_launch_march_unpack_direct(buf, buf2, nf, B.recv_pos1, B.recv_pos2, PL.n_recv, B.cut_recv, B.dx, seq, B.recvbuf, B.bad, S.stream);
        unsigned bad = 0;
        if (agree_max(bad)) return -1;
        int e = 0;
        HIPC(hipMemcpy(&e, B.dx.err, sizeof e, hipMemcpyDeviceToHost));
        unsigned tmo = e ? 1u : 0u;
        HIPC(hipMemcpyAsync(B.bad, &tmo, sizeof tmo, hipMemcpyHostToDevice, S.stream));
        if (agree_max(tmo)) return -1;
        if (e) HIPC(hipMemset(B.dx.err, 0, sizeof(int)));
        M.direct = (bad || tmo) ? 0 : 1;
        if (!M.direct) M.direct_why = tmo ? "a neighbour never signalled in the trial exchange" : "the trial exchange delivered other bits than the library";
        if (env("CICE_EVP_HIP_VERBOSE"))
            std::fprintf(stderr, "[cice_evp_hip] rank %d: trial of the direct ring exchange: %s\n", (int)S.d.rank,
                         M.direct ? "identical to RCCL, in use from now on" : M.direct_why.c_str());
    }
    return 0;
}

static int march_exchange_mask()
{
    if (PL.peers.empty()) return 0;
    evp_launch_march_pack_mask(B.mask, B.send_midx, PL.n_send, B.sendbuf, S.stream);
    if (S.test_xchg) {
        if (int rc = hook_exchange(1, S.stream)) return rc;
        evp_launch_march_unpack_mask(B.mask, B.recv_midx, PL.n_recv, B.recvbuf, S.stream);
        return 0;
    }
    size_t so = 0, ro = 0;
    NCCLC(ncclGroupStart());
    for (const MarchPeer &p : PL.peers) {
        const size_t ns = p.send_pos.size(), nr = p.recv_pos1.size();
        if (ns) NCCLC(ncclSend(B.sendbuf + so, ns, ncclDouble, p.rank, S.comm, S.stream));
        if (nr) NCCLC(ncclRecv(B.recvbuf + ro, nr, ncclDouble, p.rank, S.comm, S.stream));
        so += ns; ro += nr;
    }
    NCCLC(ncclGroupEnd());
    evp_launch_march_unpack_mask(B.mask, B.recv_midx, PL.n_recv, B.recvbuf, S.stream);
    return 0;
}

// max over ranks of a device counter (the ranks must take the same path)
static int agree_max(unsigned &v)
{
    if (S.test_reduce && !PL.peers.empty() && S.d.nranks > 1) {
        HIPC(hipMemcpyAsync(&v, B.bad, sizeof v, hipMemcpyDeviceToHost, S.stream));
        HIPC(hipStreamSynchronize(S.stream));
        if (S.test_reduce(S.test_user, 1, &v)) return fail(-2, "test transport: the reduce callback failed");
        return 0;
    }
    if (!PL.peers.empty() && S.d.nranks > 1) NCCLC(ncclAllReduce(B.bad, B.bad, 1, ncclUint32, ncclMax, S.comm, S.stream));
    HIPC(hipMemcpyAsync(&v, B.bad, sizeof v, hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

namespace {
struct TabBuilder {
    EvpMarchTab T{};
    void add(double *blk, double *pk, double *pk2, int nf, int slot)
    {
        T.blk[T.n] = blk; T.pk[T.n] = pk; T.pk2[T.n] = pk2; T.nf[T.n] = nf; T.slot[T.n] = slot; ++T.n;
    }
};
}  // namespace

// static fields into the constants block, once; are their ghost values images of one global field?
static int march_statics()
{
    State::March &M = S.march;
    TabBuilder G;
    double *src[5] = {S.stat[0], S.stat[1], S.hte, S.htn, S.stat[9]};
    const int slot[5] = {C_DXT, C_DYT, C_HTE, C_HTN, C_UAREAR};
    for (int k = 0; k < 5; ++k) G.add(src[k], B.cst, nullptr, EVP_MARCH_C_NF, slot[k]);
    evp_launch_march_gather(M.G, G.T, nullptr, nullptr, S.stream);
    if (march_exchange(B.cst, nullptr, EVP_MARCH_C_NF)) return -1;      // (the per-call slots travel too: overwritten at every call)
    TabBuilder C;              // dxT dyT (fringe) | HTE HTN (fringe + column ilo-1 / row jlo-1)
    for (int k = 0; k < 4; ++k) C.add(src[k], B.cst, nullptr, EVP_MARCH_C_NF, slot[k]);
    HIPC(hipMemsetAsync(B.bad, 0, sizeof(unsigned), S.stream));
    evp_launch_march_check(M.G, C.T, nullptr, nullptr, 0, 2, B.bad, S.stream);
    unsigned bad = 0;
    if (agree_max(bad)) return -1;
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
    // a rank without blocks never gets here, so the agreement below would wait for it for ever: every rank sees it in the
    // global block table and stays off the path, without a vote
    if (S.d.nranks > 1 && !S.gtab[4].empty()) {
        std::vector<char> has((size_t)S.d.nranks, 0);
        for (int o : S.gtab[4])
            if (o >= 0 && o < S.d.nranks) has[(size_t)o] = 1;
        for (char h : has)
            if (!h) { M.why = "a rank holds no blocks"; return false; }
    }
    const int want = env("CICE_EVP_HIP_MARCH") ? std::atoi(env("CICE_EVP_HIP_MARCH")) : -1;
    // (a rank that was told CICE_EVP_HIP_MARCH=0 still takes part in the agreement below, voting no: an environment that
    // differs between the ranks then switches the path off everywhere instead of leaving the others in a collective)
    std::string why;
    bool ok = want != 0 && march_geometry(why);
    if (want == 0) why = "CICE_EVP_HIP_MARCH=0";
    if (S.d.nranks > 1 && S.test_reduce) {
        int32_t h = ok ? 1 : 0;
        if (S.test_reduce(S.test_user, 0, &h)) return false;
        if (ok && !h) { ok = false; why = "another rank cannot use it"; }
    } else if (S.d.nranks > 1 && S.have_comm) {
        // the choice must be the same on every rank (what decides it is partly local: e.g. whether the metric terms can be
        // recomputed from the edge lengths is verified on each rank's own cells)
        int *dflag = nullptr;
        int h = ok ? 1 : 0;
        if (hipMalloc((void **)&dflag, sizeof(int)) != hipSuccess) return false;
        bool fine = hipMemcpyAsync(dflag, &h, sizeof h, hipMemcpyHostToDevice, S.stream) == hipSuccess &&
                    ncclAllReduce(dflag, dflag, 1, ncclInt32, ncclMin, S.comm, S.stream) == ncclSuccess &&
                    hipMemcpyAsync(&h, dflag, sizeof h, hipMemcpyDeviceToHost, S.stream) == hipSuccess &&
                    hipStreamSynchronize(S.stream) == hipSuccess;
        (void)hipFree(dflag);
        if (ok && (!fine || !h)) { ok = false; why = "another rank cannot use it"; }
    } else if (S.d.nranks > 1) {
        ok = false;
        if (why.empty()) why = "several ranks without an RCCL communicator";
    }
    if (!ok) {
        M.why = why;
        if (env("CICE_EVP_HIP_VERBOSE")) std::fprintf(stderr, "[cice_evp_hip] marching kernel off: %s\n", why.c_str());
        return false;
    }
    // worth it when the domain is far beyond what stays on the chip (the on-chip resident kernel is chosen before this
    // is asked): measured against the one-subcycle kernel 24 vs 26.5 us per subcycle at 389k cells, 39 vs 53 at 778k
    if (want < 0 && (long)S.d.nx_global * S.d.ny_global < 450000L * std::max(1, (int)S.d.nranks)) { M.why = "below 450k cells per rank"; return false; }
    M.mode = 1;
    return true;
}

static void march_args(EvpMarch &A, int cur, int last)
{
    const State::March &M = S.march;
    const cice_evp_hip_params &q = S.prm;
    A.p = {q.arlx1i, q.denom1, q.brlx, q.revp, q.e_factor, q.epp2i, q.capping, q.Ktens, q.u0, q.cosw, q.sinw, q.rhow};
    A.deltaminEVP = q.deltaminEVP;
    A.nxr = M.G.nxr; A.nyr = M.G.nyr; A.ldx = M.G.ldx; A.own = M.G.own;
    A.nstrips = M.nstrips; A.nseg = M.nseg; A.seglen = M.seglen; A.nitems = M.nitems;
    A.wrapx = M.G.wrapx;
    A.last = last;
    A.kpass = M.kpass;
    A.order = env_test("CICE_EVP_HIP_MARCH_ORDER") ? std::atoi(env_test("CICE_EVP_HIP_MARCH_ORDER")) : 1;
    A.flags = S.flags & S.flags_allowed;
    A.mask = B.mask;
    A.st_in = B.st[cur]; A.st_out = B.st[cur ^ 1];
    A.cst = B.cst;
    const bool need_opt = !(A.flags & EVP_F_WATER_IS_OCN) || !(A.flags & EVP_F_TBU_ZERO) || q.revp != 0.0;
    A.opt = need_opt ? B.opt : nullptr;
    A.diag = B.diag;
    A.dup = B.dup;
    A.items = nullptr;
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
        if (env("CICE_EVP_HIP_VERBOSE")) std::fprintf(stderr, "[cice_evp_hip] marching kernel declined this call: %s\n", why);
        if (left > 0)
            if (int rc = enqueue_loop(left, cur)) return rc;
        S.cur = cur ^ (left & 1);
        return 0;
    };
    if (!M.stat_ok) return fallback("ghost values of the static grid fields are not images of one global field");
    // ndte = q passes of kpass subcycles + one pass of the remaining 2 .. kpass-1; a single remaining subcycle goes first, through
    // the one-subcycle kernel in the block layout
    std::vector<int> sizes;
    {
        const int kp = M.kpass;
        int q = left / kp, rem = left % kp;
        if (rem == 1) {
            if (int rc = enqueue_loop(1, cur)) return rc;
            cur ^= 1;
            --left;
            rem = 0;
            S.cur = cur;     // the device state HAS advanced: an error further down must not leave S.cur on the old copy
        }
        sizes.assign((size_t)q, kp);
        if (rem) sizes.push_back(rem);
    }
    M.call_passes = 0; M.call_subcycles = 0;
    if (left == 0) { S.cur = cur; return 0; }
    const unsigned fl = S.flags & S.flags_allowed;
    // ---- gather the state and the per-call inputs ----
    {
        TabBuilder G;
        G.add(S.u[cur], B.st[0], B.st[1], EVP_MARCH_S_NF, 0);
        G.add(S.v[cur], B.st[0], B.st[1], EVP_MARCH_S_NF, 1);
        for (int k = 0; k < 12; ++k) G.add(S.sig[cur][k], B.st[0], B.st[1], EVP_MARCH_S_NF, 2 + k);
        G.add(S.in[F_STRENGTH], B.cst, nullptr, EVP_MARCH_C_NF, C_STRENGTH);
        G.add(S.vrelfac, B.cst, nullptr, EVP_MARCH_C_NF, C_VRELFAC);
        G.add(S.in[F_UOCN], B.cst, nullptr, EVP_MARCH_C_NF, C_UOCN); G.add(S.in[F_VOCN], B.cst, nullptr, EVP_MARCH_C_NF, C_VOCN);
        G.add(S.in[F_FORCEX], B.cst, nullptr, EVP_MARCH_C_NF, C_FORCEX); G.add(S.in[F_FORCEY], B.cst, nullptr, EVP_MARCH_C_NF, C_FORCEY);
        G.add(S.in[F_UMASSDTI], B.cst, nullptr, EVP_MARCH_C_NF, C_UMASSDTI); G.add(S.in[F_FM], B.cst, nullptr, EVP_MARCH_C_NF, C_FM);
        if (!(fl & EVP_F_WATER_IS_OCN)) {
            G.add(S.in[F_WATERX], B.opt, nullptr, EVP_MARCH_O_NF, O_WATERX); G.add(S.in[F_WATERY], B.opt, nullptr, EVP_MARCH_O_NF, O_WATERY);
        }
        if (!(fl & EVP_F_TBU_ZERO)) G.add(S.in[F_TBU], B.opt, nullptr, EVP_MARCH_O_NF, O_TBU);
        if (S.prm.revp != 0.0) {
            G.add(S.in[F_UVEL_INIT], B.opt, nullptr, EVP_MARCH_O_NF, O_UINIT); G.add(S.in[F_VVEL_INIT], B.opt, nullptr, EVP_MARCH_O_NF, O_VINIT);
        }
        evp_launch_march_gather(M.G, G.T, S.mask, B.mask, S.stream);
        // the two-cell ring of everything: other ranks' cells (once per call for the constants and the mask)
        if (march_exchange(B.cst, nullptr, EVP_MARCH_C_NF)) return -1;
        const bool need_opt = !(fl & EVP_F_WATER_IS_OCN) || !(fl & EVP_F_TBU_ZERO) || S.prm.revp != 0.0;
        if (need_opt && march_exchange(B.opt, nullptr, EVP_MARCH_O_NF)) return -1;
        if (march_exchange_mask()) return -1;
        if (march_exchange(B.st[0], B.st[1], EVP_MARCH_S_NF)) return -1;
    }
    if (M.checked_seq != S.upload_seq || !PL.peers.empty()) {
        // first call on this uploaded state: are the caller's ghost values images of one global state?
        TabBuilder C;
        C.add(S.u[cur], B.st[0], nullptr, EVP_MARCH_S_NF, 0);
        C.add(S.v[cur], B.st[0], nullptr, EVP_MARCH_S_NF, 1);
        for (int k = 0; k < 12; ++k) C.add(S.sig[cur][k], B.st[0], nullptr, EVP_MARCH_S_NF, 2 + k);
        C.add(S.in[F_STRENGTH], B.cst, nullptr, EVP_MARCH_C_NF, C_STRENGTH);
        HIPC(hipMemsetAsync(B.bad, 0, sizeof(unsigned), S.stream));
        evp_launch_march_check(M.G, C.T, S.mask, B.mask, 2, 13, B.bad, S.stream);
        unsigned bad = 0;
        if (agree_max(bad)) return -1;
        if (bad) return fallback("ghost cells of the uploaded state are not images of one global state (here or on another rank)");
        M.checked_seq = S.upload_seq;
    }
    // ---- the passes ----
    // Exchange passes (the redundant rim and the ring have been used up; the last): the (strip, row) units that hold cells other
    // ranks are waiting for are advanced FIRST, by an early launch of the same kernel over short segments on the second stream;
    // their pack and the RCCL send / recv follow there while the rest of the pass -- every other (strip, row), rest_items -- runs
    // on the compute stream: the transfer is overlapped with the interior of the pass instead of following it (ice_HaloUpdate ->
    // RCCL point-to-point on a second HIP stream over interior compute).  The two launches read the same input state and store
    // into disjoint cells (a cell's result does not depend on the segment it is computed in).  Only the unpack waits for the
    // pass: the pass still writes its (spent) redundant rim where the received cells go.
    const int npass = (int)sizes.size();
    int rc = 0;
    int valid = M.ring_valid;        // cells beyond the rank's own that hold the current state (the gather's exchange just filled them)
    // Opt-in, in the PRODUCT library too (CICE_EVP_HIP_MARCH_OVERLAP=1; every rank alike): where it can be measured -- one GPU, the
    // ring exchanged with the rank itself, profiles/r06_ring_rank.txt -- it still loses, 64.2 against 57.3 us per subcycle on the
    // 450 x 2400 piece and 61.2 against 56.9 on 900 x 1200, although the two launches no longer advance the band twice (rounds 4-5:
    // 57.7 against 51.2): the pass is bound by instruction issue with every SIMD holding one wave for the whole pass, so there is
    // no idle resource to hide a transfer under, and "early" means SHORT segments for the band -- 2K - 1 rows of warm-up on 7 stored.
    // On one GPU the transfer is a 7-us device copy; over xGMI it is 1.6 MB per neighbour every eighth subcycle, and only a node
    // can say whether hiding that is worth 6 us per subcycle.  bench.py --gpus N --extras ring_variants times both.
    const bool overlap = !PL.peers.empty() && B.nband > 0 && M.direct != 1 &&
                         env("CICE_EVP_HIP_MARCH_OVERLAP") && std::atoi(env("CICE_EVP_HIP_MARCH_OVERLAP")) &&
                         !(env_test("CICE_EVP_HIP_MARCH_DIRECT") && std::atoi(env_test("CICE_EVP_HIP_MARCH_DIRECT")));
    for (int k = 0; k < npass; ++k) {
        // the ring of the new state travels after this pass when the next one needs more valid cells than are left, and after the
        // last one (the way back to the block layout reads the ghost cells from it)
        // (a pass advances the cells the rank holds -- its own + ext -- and only reads the P-cell ring around them: whatever is left of
        // the ring afterwards is a state older)
        valid = std::min(valid - sizes[(size_t)k], M.ring_valid - EVP_MARCH_PAD);
        const bool exch = !PL.peers.empty() && (k == npass - 1 || valid < sizes[(size_t)k + 1]);
        if (exch) valid = M.ring_valid;
        EvpMarch A;
        march_args(A, rc, k == npass - 1);
        A.kpass = sizes[(size_t)k];
        if (exch && overlap) {
            HIPC(hipEventRecord(B.ev_in, S.stream));
            HIPC(hipStreamWaitEvent(S.stream_comm, B.ev_in, 0));
            EvpMarch E = A;
            E.items = B.band_items;
            E.nitems = B.nband;                                  // (the last pass: the band's diagnostics are this launch's business)
            evp_launch_march(E, S.prm.strict != 0, cap_mode(), S.stream_comm);
            if (int e = march_send_recv(B.st[rc ^ 1], EVP_MARCH_S_NF, S.stream_comm)) return e;
            A.items = B.rest_items;                              // the pass itself: everything else
            A.nitems = B.nrest;
        }
        if (A.nitems > 0) evp_launch_march(A, S.prm.strict != 0, cap_mode(), S.stream);
        rc ^= 1;
        if (exch && overlap) {
            HIPC(hipEventRecord(B.ev_main, S.stream));
            HIPC(hipStreamWaitEvent(S.stream_comm, B.ev_main, 0));
            evp_launch_march_unpack(B.st[rc], nullptr, EVP_MARCH_S_NF, B.recv_pos1, B.recv_pos2, PL.n_recv, B.cut_recv, B.recvbuf, S.stream_comm);
            HIPC(hipEventRecord(B.ev_done, S.stream_comm));
            HIPC(hipStreamWaitEvent(S.stream, B.ev_done, 0));
        } else if (exch) {
            if (march_exchange(B.st[rc], nullptr, EVP_MARCH_S_NF)) return -1;
        }
    }
    HIPC(hipGetLastError());
    M.passes += npass;
    M.subcycles += left;
    M.call_passes = npass; M.call_subcycles = left;
    // ---- back to the block layout ----
    {
        TabBuilder T;
        T.add(S.u[cur], B.st[rc], nullptr, EVP_MARCH_S_NF, 0); T.add(S.v[cur], B.st[rc], nullptr, EVP_MARCH_S_NF, 1);
        for (int k = 0; k < 12; ++k) T.add(S.sig[cur][k], B.st[rc], nullptr, EVP_MARCH_S_NF, 2 + k);
        T.add(S.in[F_STRINTX], B.diag, nullptr, EVP_MARCH_D_NF, 0); T.add(S.in[F_STRINTY], B.diag, nullptr, EVP_MARCH_D_NF, 1);
        T.add(S.in[F_TAUBX], B.diag, nullptr, EVP_MARCH_D_NF, 2); T.add(S.in[F_TAUBY], B.diag, nullptr, EVP_MARCH_D_NF, 3);
        evp_launch_march_scatter(M.G, T.T, S.mask, 2, 12, S.stream);
    }
    HIPC(hipGetLastError());
    S.cur = cur;            // the passes leave the block layout's ping-pong buffer where it was: the scatter wrote the new state there
    return 0;
}

}  // namespace evp_host

#ifdef CICE_EVP_HIP_TESTING
extern "C" int cice_evp_hip_set_test_transport(cice_evp_hip_test_xchg_fn xchg, cice_evp_hip_test_reduce_fn reduce, void *user)
{
    using namespace evp_host;
    if ((xchg == nullptr) != (reduce == nullptr)) return fail(-1, "test transport: give both callbacks or none");
    S.test_xchg = xchg; S.test_reduce = reduce; S.test_user = user;
    return 0;
}

// Host-only: the plan of dims->rank without touching a device (CPU tests).
extern "C" int cice_evp_hip_march_plan(const cice_evp_hip_dims *dims, int32_t own_max, int32_t wrap_inside, int32_t ext, int32_t *geo14,
                                       int32_t *peer_rank, int32_t *peer_nsend, int32_t *peer_nrecv, int32_t *send_pos,
                                       int32_t *recv_pos1, int32_t *recv_pos2)
{
    using namespace evp_host;
    if (!dims) return fail(-1, "null dims");
    MarchPlan P;
    if (!build_march_plan(*dims, own_max > 0 ? own_max : EVP_MARCH_OWN, wrap_inside != 0, ext, P)) return fail(-3, "march plan: %s", P.error.c_str());
    if (geo14) {
        const int32_t g[14] = {P.me.gx0, P.me.gy0, P.me.nxr, P.me.nyr, P.me.own, P.me.nstrips, (int32_t)P.peers.size(), P.n_send,
                               P.n_recv, P.wrapx ? 1 : 0, P.ext_w, P.ext_e, P.ext_s, P.ext_n};
        for (int k = 0; k < 14; ++k) geo14[k] = g[k];
    }
    size_t so = 0, ro = 0;
    for (size_t q = 0; q < P.peers.size(); ++q) {
        const MarchPeer &p = P.peers[q];
        if (peer_rank) peer_rank[q] = p.rank;
        if (peer_nsend) peer_nsend[q] = (int32_t)p.send_pos.size();
        if (peer_nrecv) peer_nrecv[q] = (int32_t)p.recv_pos1.size();
        for (size_t k = 0; k < p.send_pos.size(); ++k)
            if (send_pos) send_pos[so + k] = p.send_pos[k];
        for (size_t k = 0; k < p.recv_pos1.size(); ++k) {
            if (recv_pos1) recv_pos1[ro + k] = p.recv_pos1[k];
            if (recv_pos2) recv_pos2[ro + k] = p.recv_pos2[k];
        }
        so += p.send_pos.size(); ro += p.recv_pos1.size();
    }
    return 0;
}
#endif  // CICE_EVP_HIP_TESTING
