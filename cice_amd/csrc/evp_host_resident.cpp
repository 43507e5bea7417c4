// Host side of the on-chip resident kernel (evp_resident2.hip): ring tables, residency checks,
// launches, error word, and the choices made at the first upload.
#include "evp_host.h"

namespace evp_host {

// ---- on-chip resident subcycle -------------------------------------------------------
// (a rank of a fold row split in x may hold neither a pole point nor a pair with both halves: its seam cells are the plan's
// general list then -- 3 x 1 and 4 x 1 cuts of tx1 showed it: two of the ranks ran without any fold handling)
bool tripole_seam() { return (S.n_seam + S.n_pole + S.n_late) > 0 || S.plan.tail > 0 || !S.plan.fin_dst.empty(); }

// tripoleT (T-fold), round 6: the top physical row of U-cells is the image of row NY-1 (halo_plan.cpp: resolve) -- the plan's local
// copies whose DESTINATION is an interior cell.  The resident kernel treats such a cell like a seam cell whose new value is -1 x
// the source cell's new value (EvpResident2::tfold); the copies are taken out of the ghost-image tables here.
static bool tfold_image(size_t k)
{
    if (!S.plan.tfold) return false;
    const int dst = S.plan.local_dst[k];
    const int b = dst / (int)S.plane, r = dst % (int)S.plane, j = r / S.d.nx_block + 1, i = r % S.d.nx_block + 1;
    return i >= S.ilo[b] && i <= S.ihi[b] && j >= S.jlo[b] && j <= S.jhi[b];
}

bool resident_possible(bool with_peers)
{
    if (!with_peers && !S.plan.peers.empty()) return false;
    // tripoleT: the top row's images are interior cells -- inside the kernel (round 6) where every cell a top-row cell of this rank
    // is the image of lies on this rank too (one rank; several ranks cut in y only); a top row split in x has images whose sources
    // are other ranks' cells (the receive lists name interior cells then): the streaming kernel, images rewritten after every launch
    if (S.plan.tfold) {
        for (const HaloPeer &p : S.plan.peers)
            for (int k = 0; k < p.n_ghost_recv; ++k) {
                const int dst = p.recv_dst[k];
                const int b = dst / (int)S.plane, r = dst % (int)S.plane, j = r / S.d.nx_block + 1, i = r % S.d.nx_block + 1;
                if (i >= S.ilo[b] && i <= S.ihi[b] && j >= S.jlo[b] && j <= S.jhi[b]) return false;
            }
    }
    // tripole seam pairs across ranks: with neighbours on other GPUs the partners trade their raw records through the
    // peers' rec_raw buffers (round 4); every other caller gets the streaming kernel + exchange + seam step
    if (S.plan.tail > 0 && !with_peers) return false;
    if (tripole_seam() || S.plan.tfold || S.d.nblocks > 1) {
        // tagged-record kernel only: the fold row is averaged inside the kernel, ghost images come
        // from a per-cell table (at most three per cell, no eliminated source block)
        std::map<int, int> nimg;
        for (size_t k = 0; k < S.plan.local_dst.size(); ++k) {
            if (S.plan.local_src[k] < 0) return false;
            if (tfold_image(k)) continue;
            if (++nimg[S.plan.local_src[k]] > 3) return false;
        }
        return true;
    }
    if (S.n_local > 0 && !S.push_ok) return false;
    return true;
}

// ---- evp_resident2.hip: ring lists and publish map of a tile shape ------
// For every tile: the cells of its LDS velocity tile that it reads but does not produce itself
// (ring + ghost/truncation cells), each with the record to poll and the U-cell that produces
// it; and the map of U-cells some other tile mirrors (those publish a record each subcycle).
// Geometry only -- independent of the ice masks.
int resident2_setup(int logw)
{
    if (S.res2_ring && S.res2_logw == logw) return 0;
    auto F = [](auto *&p) { if (p) (void)hipFree((void *)p); p = nullptr; };
    F(S.res2_ring); F(S.res2_cnt); F(S.res2_pub); F(S.res2_perm); F(S.res2_late); F(S.res2_nact); F(S.res2_nlate);
    F(S.res2_live); F(S.res2_celltile);
    S.res2_nlive = 0;
    S.res2_cls_h.clear();
    S.res2_order_stale = true;
    S.res2_logw = logw;
    const int W = 1 << logw, H = 256 / W, LW = W + 1;
    int gx, gy;
    evp_resident_geometry(S.max_ni, S.max_nj, logw, &gx, &gy);
    const int nb = S.d.nblocks;
    const int ntiles = gx * gy * nb;
    const int nx = S.d.nx_block, ny = S.d.ny_block;
    const size_t plane = S.plane, ncell = S.n;
    std::vector<int> ghost_src(ncell, -1);
    for (size_t k = 0; k < S.plan.local_dst.size(); ++k)
        if (S.plan.local_src[k] >= 0 && !tfold_image(k)) ghost_src[S.plan.local_dst[k]] = S.plan.local_src[k];
    for (const HaloPeer &p : S.plan.peers) {          // produced on another rank: -2 (always refreshed)
        for (int k = 0; k < p.n_ghost_recv; ++k) ghost_src[p.recv_dst[k]] = -2;     // (what follows are staging slots, not cells)
        for (int32_t d : p.fimg_recv_dst) ghost_src[d] = -2;                        // images of seam cells: the owner's FINAL value
    }
    std::vector<char> on_seam(ncell, 0);              // cells of the tripole fold row (change every subcycle, ice or not)
    for (int32_t c : S.plan.seam_a) on_seam[c] = 1;
    for (int32_t c : S.plan.seam_b) on_seam[c] = 1;
    for (int32_t c : S.plan.seam_pole) on_seam[c] = 1;
    // T-fold: the top-row cells and the cells they are images of change every subcycle, ice or not; the sources publish
    std::vector<int> tfold_src;
    for (size_t k = 0; k < S.plan.local_dst.size(); ++k)
        if (S.plan.local_src[k] >= 0 && tfold_image(k)) {
            on_seam[S.plan.local_dst[k]] = 1;
            on_seam[S.plan.local_src[k]] = 1;
            tfold_src.push_back(S.plan.local_src[k]);
        }
    std::vector<int4> ring((size_t)ntiles * EVP_RES2_RING, make_int4(-1, 0, -1, 0));
    std::vector<int> cnt((size_t)ntiles, 0);
    std::vector<int> celltile(ncell, -1);
    std::vector<char> always((size_t)ntiles, 0);
    std::vector<uint8_t> pub(ncell, 0);
    std::vector<char> seen((size_t)(H + 1) * LW);
    // 16 x 16 tiles: rim wave / interior waves (evp_resident2.hip).  cls: which T-cells of a tile read a
    // ring velocity that has a producer (geometry); the thread -> cell permutation itself also depends
    // on the ice mask and is built by resident2_order
    const bool permuted = logw == 4;
    std::vector<uint8_t> cls(permuted ? (size_t)ntiles * 256 : 0);
    std::vector<char> ringli((size_t)(H + 1) * LW);
    for (int b = 0; b < nb; ++b) {
        const int ilo = S.ilo[b], ihi = S.ihi[b], jlo = S.jlo[b], jhi = S.jhi[b];
        const int cb = (int)(b * plane);
        for (int by = 0; by < gy; ++by)
            for (int bx = 0; bx < gx; ++bx) {
                const int t = (b * gy + by) * gx + bx;
                const int i0 = ilo + bx * (W - 1), j0 = jlo + by * (H - 1);
                std::fill(seen.begin(), seen.end(), 0);
                for (int trow = 0; trow < H - 1; ++trow)                     // the U-cells this tile owns
                    for (int tcol = 0; tcol < W - 1; ++tcol) {
                        const int i = i0 + tcol, j = j0 + trow;
                        if (i > ihi || j > jhi) continue;
                        const int cp = cb + (j - 1) * nx + (i - 1);
                        celltile[cp] = t;
                        if (on_seam[cp]) always[t] = 1;
                    }
                for (int trow = 0; trow < H; ++trow)
                    for (int tcol = 0; tcol < W; ++tcol) {
                        const int i = i0 + tcol, j = j0 + trow;
                        if (i > ihi + 1 || j > jhi + 1) continue;          // T-cell not computed
                        for (int q = 0; q < 4; ++q) {
                            const int di = -(q & 1), dj = -(q >> 1);
                            const int pc = tcol + di, pr = trow + dj, pi = i + di, pj = j + dj;
                            const bool interior = pi >= ilo && pi <= ihi && pj >= jlo && pj <= jhi;
                            const bool here = interior && pc >= 0 && pc <= W - 2 && pr >= 0 && pr <= H - 2;
                            if (here) continue;
                            const int li = (pr + 1) * LW + (pc + 1);
                            if (seen[li]) continue;
                            seen[li] = 1;
                            if (pi < 1 || pi > nx || pj < 1 || pj > ny) continue;
                            const int cp = cb + (pj - 1) * nx + (pi - 1);
                            const int src = interior ? cp : ghost_src[cp];
                            if (src == -1) continue;                   // nobody produces it: constant for the whole call
                            if (cnt[t] >= EVP_RES2_RING) return fail(-6, "resident2: ring list overflow");
                            const int always = (src >= 0 && on_seam[src]) ? 1 : 0;
                            ring[(size_t)t * EVP_RES2_RING + cnt[t]++] = make_int4(cp, li, src, always);
                            if (interior) pub[cp] = 1;
                        }
                    }
                if (permuted) {
                    std::fill(ringli.begin(), ringli.end(), 0);
                    for (int e = 0; e < cnt[t]; ++e) ringli[ring[(size_t)t * EVP_RES2_RING + e].y] = 1;
                    for (int trow = 0; trow < H; ++trow)
                        for (int tcol = 0; tcol < W; ++tcol) {
                            const int i = i0 + tcol, j = j0 + trow;
                            const bool computed = i <= ihi + 1 && j <= jhi + 1;
                            const int li = (trow + 1) * LW + (tcol + 1);
                            const bool is_late = computed && (ringli[li] || ringli[li - 1] || ringli[li - LW] || ringli[li - LW - 1]);
                            cls[(size_t)t * 256 + trow * W + tcol] = is_late ? 2 : computed ? 1 : 0;
                        }
                }
            }
    }
    for (int c : tfold_src) pub[(size_t)c] = 1;
    S.res2_ntiles = ntiles;
    S.res2_always_h = always;
    HIPC(hipMalloc((void **)&S.res2_celltile, celltile.size() * sizeof(int)));
    HIPC(hipMemcpy(S.res2_celltile, celltile.data(), celltile.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPC(hipMalloc((void **)&S.res2_live, (size_t)ntiles));
    HIPC(hipMemset(S.res2_live, 1, (size_t)ntiles));
    // ghost images from a per-cell table whenever they are not confined to the edge of ONE block:
    // tripole grids (ghost row NY+1 mirrors row NY-1) and several blocks per rank
    if ((tripole_seam() || S.plan.tfold || nb > 1) && !S.res2_img3) {
        std::vector<int> img3(ncell * 3, -1);
        for (size_t k = 0; k < S.plan.local_dst.size(); ++k) {
            const int src = S.plan.local_src[k];
            if (src < 0 || tfold_image(k)) continue;
            const int enc = S.plan.local_dst[k] * 2 + (S.plan.local_sign[k] < 0 ? 1 : 0);
            int e = 0;
            while (e < 3 && img3[(size_t)src * 3 + e] >= 0) ++e;
            if (e == 3) return fail(-6, "resident2: more than three ghost images of one cell");
            img3[(size_t)src * 3 + e] = enc;
        }
        HIPC(hipMalloc((void **)&S.res2_img3, img3.size() * sizeof(int)));
        HIPC(hipMemcpy(S.res2_img3, img3.data(), img3.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if ((tripole_seam() || S.plan.tfold) && !S.res2_seam) {
        std::vector<int> seam((size_t)nx * nb, 0);       // per block and column of the fold row
        auto slot = [&](int32_t off) { return (size_t)(off / plane) * nx + (off % plane) % nx; };
        for (size_t k = 0; k < S.plan.local_dst.size(); ++k)         // T-fold: the cell of row NY-1 a top-row cell is the image of
            if (S.plan.local_src[k] >= 0 && tfold_image(k)) seam[slot(S.plan.local_dst[k])] = S.plan.local_src[k] * 4 + 1;
        for (size_t k = 0; k < S.plan.seam_a.size(); ++k) {
            seam[slot(S.plan.seam_a[k])] = S.plan.seam_b[k] * 4 + 1;
            seam[slot(S.plan.seam_b[k])] = S.plan.seam_a[k] * 4 + 2;
        }
        for (int32_t pcell : S.plan.seam_pole) seam[slot(pcell)] = 3;
        // pairs with the partner on another rank (fold row split in x): the plan's general seam list names, for each of this
        // rank's seam cells, the RAW operands a (low column) and b (high column) -- a local cell or a staging slot >= ncell,
        // which is where the partner's raw record lands in this rank's rec_raw buffer
        for (size_t k = 0; k < S.plan.fin_dst.size(); ++k) {
            const int32_t dst = S.plan.fin_dst[k], a = S.plan.fin_a[k], b = S.plan.fin_b[k];
            if (b < 0 || (dst != a && dst != b)) continue;                   // pole / unpaired cell, or a ghost image
            if ((size_t)a < ncell && (size_t)b < ncell) continue;            // both local: set above
            seam[slot(dst)] = (dst == a) ? b * 4 + 1 : a * 4 + 2;
        }
        HIPC(hipMalloc((void **)&S.res2_seam, seam.size() * sizeof(int)));
        HIPC(hipMemcpy(S.res2_seam, seam.data(), seam.size() * sizeof(int), hipMemcpyHostToDevice));
        for (auto &q : S.res2_rec_raw)
            if (!q) {
                if (!S.res2_raw_owned) return fail(-6, "resident2: raw seam records missing from the mailbox");
                HIPC(hipMalloc(&q, (ncell + (size_t)S.plan.tail) * 32));
                HIPC(hipMemset(q, 0, (ncell + (size_t)S.plan.tail) * 32));
            }
    }
    HIPC(hipMalloc((void **)&S.res2_ring, ring.size() * sizeof(int4)));
    HIPC(hipMemcpy(S.res2_ring, ring.data(), ring.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPC(hipMalloc((void **)&S.res2_cnt, cnt.size() * sizeof(int)));
    HIPC(hipMemcpy(S.res2_cnt, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPC(hipMalloc((void **)&S.res2_pub, pub.size()));
    HIPC(hipMemcpy(S.res2_pub, pub.data(), pub.size(), hipMemcpyHostToDevice));
    if (permuted) {
        HIPC(hipMalloc((void **)&S.res2_perm, (size_t)ntiles * 256));
        HIPC(hipMalloc((void **)&S.res2_late, (size_t)ntiles));
        HIPC(hipMalloc((void **)&S.res2_nact, (size_t)ntiles));
        HIPC(hipMalloc((void **)&S.res2_nlate, (size_t)ntiles));
        if (!S.res2_cuload) {
            HIPC(hipMalloc((void **)&S.res2_cuload, 2048 * 8 * sizeof(int)));
            HIPC(hipMemset(S.res2_cuload, 0, 2048 * 8 * sizeof(int)));
        }
        S.res2_cls_h = cls;
        S.res2_cnt_h = cnt;
    }
    for (auto &p : S.res2_rec)
        if (!p) {
            if (!S.res2_rec_owned) return fail(-6, "resident2: record buffers missing from the mailbox");
            HIPC(hipMalloc(&p, ncell * 32));
            HIPC(hipMemset(p, 0, ncell * 32));
        }
    if (!S.res_err) {
        HIPC(hipMalloc((void **)&S.res_err, 8 * sizeof(int)));
        HIPC(hipMemset(S.res_err, 0, 8 * sizeof(int)));
    }
    return 0;
}

bool resident2_fits(bool remote)
{
    // (asked at every cice_evp_hip_subcycle since only the tiles with ice have to fit: the device's answer is kept per
    // kernel variant, the question then costs a comparison)
    struct Key { int dev, logw, strict, capm; unsigned fl; bool remote; long cap; };
    static std::vector<Key> known;
    // Decided once, but TBU_ZERO / WATER_IS_OCN are re-derived from the data at every upload / prep and
    // the LDS need of a workgroup grows when they drop (up to 6 KB): size with the flag combination
    // that needs the most LDS, so that a later call can never have fewer workgroups per CU than the
    // tile count was admitted against.
    const unsigned fl = S.flags & S.flags_allowed & ~(EVP_F_WATER_IS_OCN | EVP_F_TBU_ZERO);
    long cap = -1;
    for (const Key &k : known)
        if (k.dev == S.device && k.logw == S.res2_logw && k.strict == (S.prm.strict != 0) && k.capm == cap_mode() && k.fl == fl && k.remote == remote)
            cap = k.cap;
    if (cap < 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, S.device) != hipSuccess) return false;
        const int per_cu = std::min(evp_resident2_max_blocks_per_cu(S.prm.strict != 0, cap_mode(), fl, S.res2_logw, remote), 8);
        cap = (long)per_cu * prop.multiProcessorCount;
        known.push_back(Key{S.device, S.res2_logw, S.prm.strict != 0, cap_mode(), fl, remote, cap});
    }
    // what has to be co-resident is the tiles that run: all of them with neighbours on other ranks or before the masks are
    // known, the ones that hold ice otherwise (resident2_order)
    const int need = (remote || !S.plan.peers.empty() || S.res2_order_stale || !S.res2_order || S.res2_order_for != S.res2_logw)
                         ? S.res2_ntiles : std::max(S.res2_nlive, 1);
    return S.res2_ntiles > 0 && (long)need * 10 <= cap * 9;
}


// the same question for the masks of the current upload (the launch order, and with it the list of tiles that run, is brought up
// to date first); with neighbours on other ranks the answer never changes after the start-up agreement
bool resident2_fits_now()
{
    if (!S.plan.peers.empty() || S.res_remote) return true;
    if (resident_tables() || resident2_order()) { g_err.clear(); return false; }
    return resident2_fits(false);
}

// What depends on the ice masks, rebuilt when they or the tile shape change:
//  * 16 x 16 tiles: the thread -> cell permutation.  Ice cells that read ring velocities first (they
//    and the ring poll share the wave that takes chunk 0), then the other ice cells, then the rest: a
//    tile costs ceil(ice cells / 64) waves instead of four.
//  * the launch order.  All tiles advance in lock step and a SIMD issues for one wave at a time, so the
//    CU with the most ice-holding waves paces the grid.  Workgroup w lands on CU w mod #CUs (observed
//    breadth-first placement, tools/wg_placement.hip; used for speed only): longest-processing-time
//    assignment of tiles to CUs with the slots each CU gets, heaviest tile of a CU first.
//    CICE_EVP_HIP_RES_ORDER=0 keeps the natural order.
int resident2_order()
{
    if (S.res2_order && !S.res2_order_stale && S.res2_order_for == S.res2_logw) return 0;
    const int W = 1 << S.res2_logw, H = 256 / W;
    int gx, gy;
    evp_resident_geometry(S.max_ni, S.max_nj, S.res2_logw, &gx, &gy);
    const int ntiles = gx * gy * S.d.nblocks, nx = S.d.nx_block;
    const bool permuted = !S.res2_cls_h.empty() && S.res2_logw == 4;
    std::vector<int> cost((size_t)ntiles);        // 1024 * ice-holding waves + ice cells
    std::vector<uint8_t> live((size_t)ntiles, 1);
    std::vector<uint8_t> perm(permuted ? (size_t)ntiles * 256 : 0), late(permuted ? (size_t)ntiles : 0),
                         nact(permuted ? (size_t)ntiles : 0), nlt(permuted ? (size_t)ntiles : 0);
    bool coop_ok = permuted;
    for (int t = 0; t < ntiles; ++t) {
        const int b = t / (gx * gy), bx = (t % (gx * gy)) % gx, by = (t % (gx * gy)) / gx;
        const int i0 = S.ilo[b] + bx * (W - 1), j0 = S.jlo[b] + by * (H - 1);
        auto ice = [&](int pos) -> bool {           // T-cell or U-cell of this position takes part in the loop
            const int i = i0 + (pos & (W - 1)), j = j0 + pos / W;
            if (i > S.ihi[b] + 1 || j > S.jhi[b] + 1) return false;
            return (S.hmask[b * S.plane + (size_t)(j - 1) * nx + (i - 1)] & 3u) != 0;
        };
        int n = 0, waves = 0;
        if (permuted) {
            const uint8_t *cl = &S.res2_cls_h[(size_t)t * 256];
            uint8_t *pm = &perm[(size_t)t * 256];
            int k = 0, nlate = 0;
            for (int pass = 0; pass < 3; ++pass)
                for (int pos = 0; pos < 256; ++pos) {
                    const bool on = cl[pos] != 0 && ice(pos);
                    const int which = on ? (cl[pos] == 2 ? 0 : 1) : 2;
                    if (which != pass) continue;
                    pm[k++] = (uint8_t)pos;
                    if (pass == 0) ++nlate;
                    if (pass < 2) ++n;
                }
            waves = (n + 63) / 64;
            nact[t] = (uint8_t)waves;
            late[t] = (uint8_t)std::min(4, (std::max(nlate, S.res2_cnt_h[t]) + 63) / 64);
            // COOP: the first 64 ice cells of the list -- every rim cell and as many of the others as the 64 quads take -- are
            // updated by quads, so that the wave that polls the ring holds no cell of its own to update afterwards (a wave's
            // pass through the stress update costs the same for 4 active lanes as for 64)
            nlt[t] = (uint8_t)std::min(n, 64);
            if (nlate > 64) coop_ok = false;
        } else {
            int wave_on[4] = {0, 0, 0, 0};
            for (int pos = 0; pos < 256; ++pos)
                if (ice(pos)) { ++n; wave_on[pos >> 6] = 1; }
            waves = wave_on[0] + wave_on[1] + wave_on[2] + wave_on[3];
        }
        cost[t] = 1024 * waves + n;
        // runs: holds ice (T- or U-cell), or cells of the tripole fold row; every tile when other ranks read this one's
        // records; never a tile without U-cells (it leaves at once anyway)
        const bool no_ucell = i0 > S.ihi[b] || j0 > S.jhi[b];
        live[t] = !no_ucell && (n > 0 || S.res2_always_h[t] || !S.plan.peers.empty()) ? 1 : 0;
        // (test build, RES_DEBUG bit 512: the tiles without U-cells run as they used to -- the hazard they were, evp_resident2.hip)
        if (no_ucell && env_test("CICE_EVP_HIP_RES_DEBUG") && (std::atoi(env_test("CICE_EVP_HIP_RES_DEBUG")) & 512)) live[t] = 1;
    }
    std::vector<int> run;
    for (int t = 0; t < ntiles; ++t)
        if (live[t]) run.push_back(t);
    const int nrun = (int)run.size();
    if (permuted) {
        HIPC(hipMemcpyAsync(S.res2_perm, perm.data(), perm.size(), hipMemcpyHostToDevice, S.stream));
        HIPC(hipMemcpyAsync(S.res2_late, late.data(), late.size(), hipMemcpyHostToDevice, S.stream));
        HIPC(hipMemcpyAsync(S.res2_nact, nact.data(), nact.size(), hipMemcpyHostToDevice, S.stream));
        HIPC(hipMemcpyAsync(S.res2_nlate, nlt.data(), nlt.size(), hipMemcpyHostToDevice, S.stream));
    }
    S.res2_coop_ok = coop_ok;
    const bool off = env_test("CICE_EVP_HIP_RES_ORDER") && !std::atoi(env_test("CICE_EVP_HIP_RES_ORDER"));
    std::vector<int> order((size_t)ntiles, 0);
    for (int w = 0; w < nrun; ++w) order[w] = run[w];
    if (!off) {
        hipDeviceProp_t prop;
        int ncu = 256;
        if (hipGetDeviceProperties(&prop, S.device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        std::vector<int> by_cost = run;
        std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int c) { return cost[a] > cost[c]; });
        // CU c receives the workgroups c, c + ncu, c + 2 ncu, ...: slots[c] of them
        std::vector<int> slots((size_t)ncu), load((size_t)ncu, 0);
        std::vector<std::vector<int>> mine((size_t)ncu);
        for (int c = 0; c < ncu; ++c) slots[c] = nrun / ncu + (c < nrun % ncu ? 1 : 0);
        for (int tile : by_cost) {                 // heaviest first, to the least loaded CU with a free slot
            int best = -1;                           // (ties: the CU with fewer slots, so that the CUs with one more workgroup stay light)
            for (int c = 0; c < ncu; ++c) {
                if ((int)mine[c].size() >= slots[c]) continue;
                if (best < 0 || load[c] < load[best] || (load[c] == load[best] && slots[c] < slots[best])) best = c;
            }
            mine[best].push_back(tile);
            load[best] += cost[tile] >> 10;
        }
        for (int c = 0; c < ncu; ++c)
            for (size_t k = 0; k < mine[c].size(); ++k) order[c + (int)k * ncu] = mine[c][k];
    }
    if (S.res2_order && S.res2_order_for != S.res2_logw) { (void)hipFree(S.res2_order); S.res2_order = nullptr; }
    if (!S.res2_order) HIPC(hipMalloc((void **)&S.res2_order, order.size() * sizeof(int)));
    HIPC(hipMemcpyAsync(S.res2_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemcpyAsync(S.res2_live, live.data(), live.size(), hipMemcpyHostToDevice, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    S.res2_nlive = nrun;
    S.res2_order_for = S.res2_logw;
    S.res2_order_stale = false;
    return 0;
}

int launch_resident2(int ndte, int cur0, bool dry)
{
    if (ndte >= 4096) return fail(-6, "resident2: ndte must be < 4096");
    if (int rc = resident_tables()) return rc;
    if (int rc = resident2_order()) return rc;
    EvpArgs A;
    fill_args(A, cur0, 1);
    EvpResident2 R;
    R.ndte = ndte;
    R.cur0 = dry ? 0 : cur0;
    R.dry = dry ? 1 : 0;
    S.res2_epoch = (S.res2_epoch + 1u) & 0xFFFFFu;
    if (S.res2_epoch == 0) S.res2_epoch = 1;
    R.tag_base = S.res2_epoch << 12;
    R.par0 = S.res2_par;
    S.res2_par = (S.res2_par + ndte + 1) & 1;     // never start in the buffer the previous launch ended in
    R.nblocks = S.d.nblocks;
    R.order = S.res2_order;
    R.nlaunch = S.res2_nlive;
    R.live = S.plan.peers.empty() ? S.res2_live : nullptr;
    R.celltile = S.res2_celltile;
    R.perm = S.res2_perm;
    R.late_waves = S.res2_late;
    R.nact = S.res2_nact;
    // rim T-cells by corners (COOP): where the variant is built, every tile's rim list fits its 64 quads and the chip still
    // takes all tiles at once with the variant's larger LDS share (the same rule as resident2_fits)
    {
        const int want = env_test("CICE_EVP_HIP_RES_COOP") ? std::atoi(env_test("CICE_EVP_HIP_RES_COOP")) : EVP_RES2_COOP_DEFAULT;
        bool can = want != 0 && S.res2_coop_ok && S.res2_logw == 4 && !S.res_remote &&
                   evp_resident2_coop_built(S.prm.strict != 0, cap_mode(), S.res2_logw, false);
        if (can) {
            hipDeviceProp_t prop;
            const int per_cu = std::min(evp_resident2_max_blocks_per_cu(S.prm.strict != 0, cap_mode(), A.flags, 4, false, true), 8);
            can = hipGetDeviceProperties(&prop, S.device) == hipSuccess &&
                  (long)std::max(S.res2_nlive, 1) * 10 <= (long)per_cu * prop.multiProcessorCount * 9;
        }
        S.res2_coop = can ? 1 : 0;
        R.nlate = can ? S.res2_nlate : nullptr;
    }
    R.cuload = S.res2_cuload;
    R.prof = nullptr;
    if (S.res2_logw == 4 && env_test("CICE_EVP_HIP_RES_PROF") && std::atoi(env_test("CICE_EVP_HIP_RES_PROF"))) {
        if (!S.res2_prof) HIPC(hipMalloc((void **)&S.res2_prof, (size_t)S.res2_ntiles * 32 * sizeof(unsigned long long)));
        HIPC(hipMemsetAsync(S.res2_prof, 0, (size_t)S.res2_ntiles * 32 * sizeof(unsigned long long), S.stream));
        R.prof = S.res2_prof;
    }
    const int dbg2 = env_test("CICE_EVP_HIP_RES_DEBUG") ? std::atoi(env_test("CICE_EVP_HIP_RES_DEBUG")) : 0;
    R.dbg = dbg2;
    R.seam = S.res2_seam;
    R.tfold = S.plan.tfold ? 1 : 0;
    R.img3 = S.res2_img3;
    R.rec_raw[0] = S.res2_rec_raw[0];
    R.rec_raw[1] = S.res2_rec_raw[1];
    R.rimg = S.res_remote ? S.res2_rimg : nullptr;
    R.peer_rec = S.res2_peer_rec;
    R.peer_rstride = S.res2_peer_rstride;
    R.rraw = S.res_remote ? S.res2_rraw : nullptr;
    R.peer_raw = S.res2_peer_raw;
    R.peer_raw_stride = S.res2_peer_raw_stride;
    const double tmo_ms = env("CICE_EVP_HIP_HALO_TIMEOUT_MS") ? std::atof(env("CICE_EVP_HIP_HALO_TIMEOUT_MS")) : 30000.0;
    R.timeout_ticks = (unsigned long long)((S.res_timeout_ms > 0 ? S.res_timeout_ms : tmo_ms) * 1.0e5);
    R.spin_limit = 4000000u;
    R.err = S.res_err;
    R.pubmap = S.res2_pub;
    R.ring = S.res2_ring;
    R.ring_cnt = S.res2_cnt;
    R.rec[0] = S.res2_rec[0];
    R.rec[1] = S.res2_rec[1];
    if (dry) {   // inputs come from the current state, nothing is written back
        R.u[0] = S.u[cur0]; R.v[0] = S.v[cur0]; R.u[1] = S.u[cur0]; R.v[1] = S.v[cur0];
    } else {
        R.u[0] = S.u[0]; R.v[0] = S.v[0]; R.u[1] = S.u[1]; R.v[1] = S.v[1];
    }
    R.tab = S.res_tab + (dry ? 28 * (1 + cur0) : 0);
    evp_launch_resident2(A, R, S.max_ni, S.max_nj, S.res2_logw, S.prm.strict != 0, cap_mode(), S.stream);
    HIPC(hipGetLastError());
    return 0;
}

int resident_tables()
{
    if (!S.res_tab) {
        // three pointer tables, uploaded once: [0] real run, [1]/[2] dry probe reading sig[0]/sig[1]
        double *tab[3][28];
        for (int v = 0; v < 3; ++v) {
            for (int k = 0; k < 12; ++k) {
                tab[v][k] = S.sig[v == 0 ? 0 : v - 1][k];
                tab[v][12 + k] = S.sig[v == 0 ? 1 : v - 1][k];
            }
            tab[v][24] = S.in[F_STRINTX]; tab[v][25] = S.in[F_STRINTY];
            tab[v][26] = S.in[F_TAUBX]; tab[v][27] = S.in[F_TAUBY];
        }
        HIPC(hipMalloc((void **)&S.res_tab, sizeof tab));
        HIPC(hipMemcpy(S.res_tab, tab, sizeof tab, hipMemcpyHostToDevice));
    }
    return 0;
}

int resident_check_error()
{
    if (!S.res_launched) return 0;
    S.res_launched = false;
    int ev[8] = {0};
    HIPC(hipMemcpy(ev, S.res_err, sizeof ev, hipMemcpyDeviceToHost));
    const int e = ev[0];
    if (e) {
        HIPC(hipMemset(S.res_err, 0, sizeof ev));
        S.res_mode = 0;
        return fail(-7, "resident EVP kernel: a wait gave up (%s; tile %d, subcycle %d, cell %d, tag seen %#x, wanted %#x)%s",
                    e == 1 ? "record of this GPU" : e == 2 ? "record of another rank" : e == 3 ? "fold-row partner" : "?",
                    ev[1], ev[2], ev[3], (unsigned)ev[4], (unsigned)ev[5],
                    e == 2 ? " -- CICE_EVP_HIP_HALO_TIMEOUT_MS bounds the wait for other ranks"
                           : " -- workgroups not co-resident?");
    }
    return 0;
}


// Choices made once the first state is on the device (tile shape of the streaming kernel,
// streaming vs on-chip resident kernel); shared by cice_evp_hip_upload and cice_evp_hip_prep.
int tune_after_upload()
{
    float ms = 0;
    if (!S.tyb_forced && !S.tuned) {
        // pick the tile height once per init by timing a few launches of each variant on
        // the real state (results are identical for every tile shape; only speed differs).
        // The launches write the ping-pong "next" buffers, which the first real subcycle
        // overwrites, so the state is not advanced.
        // tile heights whose wave count fills the 4 SIMDs evenly (4, 8) plus 3; odd wave counts
        // (5, 9) leave one SIMD with twice the work and measured 1.5-2x slower
        const int cand[9] = {4, 8, 3, 104, 108, 103, 204, 208, 203};
        float best = 1e30f;
        int best_t = 5;
        EvpArgs A;
        fill_args(A, S.cur, 0);
        for (int c : cand) {
            for (int rep = 0; rep < 2; ++rep) {   // first pass warms caches / code
                HIPC(hipEventRecord(S.ev2, S.stream));
                for (int k = 0; k < 8; ++k)
                    evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, c, S.prm.strict != 0, cap_mode(), S.stream);
                HIPC(hipEventRecord(S.ev3, S.stream));
                HIPC(hipStreamSynchronize(S.stream));
                HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
            }
            if (ms < best) { best = ms; best_t = c; }
        }
        S.tyb = best_t;
        S.tuned = true;
        S.t_stream_probe_ms = best / 8.0;
        for (auto &kv : S.graphs) (void)hipGraphExecDestroy(kv.second);
        S.graphs.clear();
    }
    // on-chip resident subcycle: use it when it fits and a dry probe on scratch velocities
    // (same work, nothing written back) runs clean and faster than the streaming kernel
    if (S.res_mode < 0) {
        S.res_mode = 0;
        int want = -1;
        if (env("CICE_EVP_HIP_RESIDENT")) want = std::atoi(env("CICE_EVP_HIP_RESIDENT"));
        if (want != 0 && S.res_remote && S.direct.on) {
            // neighbours on other GPUs: tile shape fixed at export, no timing probes (every launch
            // of this kernel is collective across ranks)
            S.res_mode = 1;
        } else if (want != 0 && resident_possible()) {
            const int forced_w = env_test("CICE_EVP_HIP_RES_LOGW") ? std::atoi(env_test("CICE_EVP_HIP_RES_LOGW")) : 0;
            float best = 1e30f;
            int best_w = 0;
            bool any_fit = false;
            for (int logw : {5, 4, 6}) {
                if (forced_w && logw != forced_w) continue;
                if (resident2_setup(logw)) { if (want == 1) return -6; continue; }
                if (resident2_order()) { if (want == 1) return -6; continue; }      // (which tiles hold ice: what has to fit)
                if (!resident2_fits()) continue;
                any_fit = true;
                if (want == 1 && forced_w) { best = 0.0f; best_w = logw; break; }
                // steady-state cost per subcycle = slope between a short and a long dry run
                // (launch, prologue and epilogue are paid once per evp() call)
                const int nshort = 8, nlong = 40;
                float tres = 1e30f, tl[2] = {0, 0};
                bool ok = true;
                for (int rep = 0; rep < 3 && ok; ++rep) {
                    const int np = (rep == 2) ? nlong : nshort;      // rep 0 warms up
                    HIPC(hipEventRecord(S.ev2, S.stream));
                    if (int rc = launch_resident2(np, S.cur, true)) return rc;
                    HIPC(hipEventRecord(S.ev3, S.stream));
                    HIPC(hipStreamSynchronize(S.stream));
                    S.res_launched = true;
                    if (resident_check_error()) { ok = false; g_err.clear(); break; }   // a tolerated probe failure is not the caller's error
                    HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
                    if (rep >= 1) tl[rep - 1] = ms;
                }
                if (ok) tres = (tl[1] - tl[0]) / (nlong - nshort);
                if (ok && tres < best) { best = tres; best_w = logw; }
            }
            S.t_res_probe_ms = best_w ? best : -1.0;
            if (best_w && (want == 1 || S.t_stream_probe_ms <= 0.0 || best < S.t_stream_probe_ms)) {
                if (int rc = resident2_setup(best_w)) return rc;
                S.res_mode = 1;
            } else if (want == 1) {
                return fail(-6, any_fit ? "resident EVP kernel requested but its probe failed"
                                        : "resident EVP kernel requested but its workgroups cannot be co-resident");
            }
        } else if (want == 1) {
            return fail(-6, "resident EVP kernel requested but not applicable (one block per rank, no remote halo, no tripole)");
        }
    }
    return 0;
}

}  // namespace evp_host
