// =====================================================================
// Two EVP subcycles per pass over HBM: the "marching" kernel (gfx950, wave64, fp64) for per-rank domains
// that do not fit on the chip (3600 x 2400 and the like).
//
// Why: the one-subcycle streaming kernel (evp_kernels.hip) already moves no byte twice and runs at the box's mixed
// read/write streaming rate; per subcycle it must read 27 and write 14 doubles per cell.  The only way below that is
// fewer sweeps: this kernel advances the state by TWO subcycles of the reference's loop
// (ice_dyn_evp.F90:859-913: stress :867, stepu :889, halo :908) while touching every array once.
//
// Data: a device-private "rectangle" layout (evp_host_march.cpp): all blocks of the rank assembled into one array
// per field, row stride ldx, two halo columns / rows on every side (cyclic wrap images, neighbours' cells, or zeros).
// The CICE-layout arrays are gathered into it before the loop and scattered back after it.
//
// Work item = ONE WAVE marching north over a strip of 64 columns x seglen rows; lane = column.  No workgroup barrier,
// no LDS-shared data: a workgroup is just four independent waves.
//   row r of the march:   S1  stress of subcycle k+1 on T-row r      (velocities U(k) of rows r-1, r)
//                         U1  stepu  of subcycle k+1 on U-row r-1    (stress divergence from T-rows r-1, r)
//                         S2  stress of subcycle k+2 on T-row r-1    (velocities U(k+1) of rows r-2, r-1)
//                         U2  stepu  of subcycle k+2 on U-row r-2    -> stored
//   * neighbours in i (uvel(i-1,j), HTE(i-1,j), str(i+1,j,.)) come from the adjacent lane by wavefront shuffles,
//     neighbours in j from values the wave carries from its previous row in registers;
//   * the 12 stresses of subcycle k+1 wait for S2 in a per-wave LDS stash (2 rows x 12 x 64 doubles = 12 KB);
//   * validity shrinks by one lane per stage: S1 lanes 1..63, U1 1..62, S2 2..62, U2 2..61 -- a strip owns 60
//     output columns, neighbouring strips / segments recompute the overlap (1.07 x 1.06 redundant work at
//     3600 x 2400 with 48-row segments), bit-identical by construction: same operands, same operation order
//     (evp_cell.inc), nothing depends on scheduling.
// Algorithmic HBM bytes per cell and PASS: 27 reads + 14 writes = 328 B, i.e. 164 B per cell-subcycle against the
// 368 B yardstick of SURVEY 8(d).
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "evp_device.h"
#include "evp_math.h"

namespace {

// neighbour lanes: one DPP move per 32-bit half (wave_shr:1 / wave_shl:1 span all 64 lanes on gfx9-family hardware)
__device__ __forceinline__ double lane_up(double v)       // lane l <- lane l-1 (lane 0: undefined, never used)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_dn(double v)       // lane l <- lane l+1 (lane 63: undefined, never used)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// LEAN: the host has verified waterx == uocn, watery == vocn, TbU == 0 on every ice U-cell and MODE == 3 (classic
// EVP, revp == 0): eight momentum operands per U-cell instead of thirteen stay alive between U1 and U2.
template <bool STRICT, int MODE, bool LEAN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void evp_march2(EvpMarch A)
{
    using MM = Math<STRICT>;
    using SI = typename MM::SI;
    using UI = typename MM::UI;
    using UO = typename MM::UO;
    __shared__ double stash[4][2][12][64];
    // scalar base + 32-bit byte offset: one offset register serves every array (a 64-bit per-lane address per array,
    // kept across the loop, cost 90 registers and spilled)
    auto LD = [](const double *p, unsigned off) -> double {
        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p) + off);
    };
    auto ST = [](double *p, unsigned off, double v) {
        *reinterpret_cast<double *>(reinterpret_cast<char *>(p) + off) = v;
    };

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= A.nitems) return;                      // whole waves; the kernel has no barrier
    const int strip = item % A.nstrips, seg = item / A.nstrips;
    const int x = strip * EVP_MARCH_OWN - 2 + lane;    // this lane's column (T-cell and U-cell of every stage)
    const int Y0 = seg * A.seglen;
    const int Y1 = min(Y0 + A.seglen, A.nyr);          // owned output rows [Y0, Y1)
    const unsigned flags = A.flags;
    const bool water_is_ocn = LEAN || (flags & EVP_F_WATER_IS_OCN);
    const bool tbu_zero = LEAN || (flags & EVP_F_TBU_ZERO);
    const bool revised = !LEAN && A.p.revp != 0.0;
    const bool own_x = lane >= 2 && lane <= 61 && x < A.nxr;
    // E-W cyclic wrap inside this rank: the halo columns are images of owned columns and are kept current by the
    // lanes that own their sources
    int img = 0;
    if (A.wrapx && own_x) {
        if (x < 2) img = A.nxr;
        else if (x >= A.nxr - 2) img = -A.nxr;
    }
    const unsigned rowb = (unsigned)A.ldx * 8u;                     // bytes per row
    const int e0 = (Y0 - 2 + EVP_MARCH_PAD) * A.ldx + EVP_MARCH_PAD + x;      // element (x, Y0-2)
    unsigned ob = (unsigned)e0 * 8u;
    unsigned em = (unsigned)e0;                                     // element index for the byte mask
    const int imgb = img * 8;

    // ---- carried state: what the rows below have left for this one ----
    double u_p = LD(A.u_in, ob), v_p = LD(A.v_in, ob);                        // U(k), row r-1
    double htn_p = LD(A.HTN, ob), htn_pp = 0, hte_p = 0;                 // HTN rows r-1, r-2; HTE row r-1
    double dxT_p = 0, dyT_p = 0, strength_p = 0;                    // T-row r-1 (statics of S2)
    unsigned m_p = 0, m_pp = 0;                                     // masks of rows r-1, r-2
    double c1_sx0 = 0, c1_sx1 = 0, c1_sy0 = 0, c1_sy2 = 0;          // str(k+1) of T-row r-1 -> U-row r-1
    double u1_p = 0, v1_p = 0;                                      // U(k+1), row r-2
    double c2_sx0 = 0, c2_sx1 = 0, c2_sy0 = 0, c2_sy2 = 0;          // str(k+2) of T-row r-2 -> U-row r-2

    for (int r = Y0 - 1; r <= Y1 + 1; ++r) {
        ob += rowb; em += (unsigned)A.ldx;                          // element (x, r)
        const unsigned m = A.mask[em];
        const double u0 = LD(A.u_in, ob), v0 = LD(A.v_in, ob);
        const double hte = LD(A.HTE, ob), htn = LD(A.HTN, ob);

        // ---- S1: stress(k+1) on T(x, r) -------------------------------------------------------
        const bool act1 = (m & 1u) && lane >= 1;
        double str1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) str1[k] = 0.0;
        double dxT = 0, dyT = 0, strength = 0;
        {
            const double uL = lane_up(u0), vL = lane_up(v0), hteL = lane_up(hte);
            const double uL_p = lane_up(u_p), vL_p = lane_up(v_p);
            if (act1) {
                double s[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) s[k] = LD(A.sig_in[k], ob);
                dxT = LD(A.dxT, ob); dyT = LD(A.dyT, ob); strength = LD(A.strength, ob);
                SI a;
                a.dxT = dxT; a.dyT = dyT; a.strength = strength;
                a.u_ij = u0; a.u_im = uL; a.u_jm = u_p; a.u_mm = uL_p;
                a.v_ij = v0; a.v_im = vL; a.v_jm = v_p; a.v_mm = vL_p;
                MM::metrics(hte, hteL, htn, htn_p, A.deltaminEVP, a);
                MM::template stress<MODE>(A.p, a, s, str1);
#pragma unroll
                for (int k = 0; k < 12; ++k) stash[wv][r & 1][k][lane] = s[k];
            }
        }
        // T-row r is row "j+1" of U-row r-1 and row "j" of U-row r (ice_dyn_shared.F90:948-951)
        const double n1_sx3 = lane_dn(str1[3]), n1_sy3 = lane_dn(str1[7]);
        const double t1_sx1 = lane_dn(str1[1]), t1_sy2 = lane_dn(str1[6]);

        // ---- U1: stepu(k+1) on U(x, r-1) ------------------------------------------------------
        // (U2 reads the same operands again one row later: from L2 / Infinity Cache, not from registers -- carrying
        // them across S2 spilled 150 B per lane)
        auto momentum = [&](unsigned eu, double uold, double vold, double sx0, double sx1, double sx2, double sx3,
                            double sy0, double sy1, double sy2, double sy3, UO &o) {
            UI w;
            w.vrelfac = LD(A.vrelfac, eu);
            w.uocn = LD(A.uocn, eu); w.vocn = LD(A.vocn, eu);
            w.forcex = LD(A.forcex, eu); w.forcey = LD(A.forcey, eu);
            w.Umassdti = LD(A.umassdti, eu); w.fm = LD(A.fm, eu); w.uarear = LD(A.uarear, eu);
            if (water_is_ocn) { w.waterx = w.uocn; w.watery = w.vocn; }
            else { w.waterx = LD(A.waterx, eu); w.watery = LD(A.watery, eu); }
            w.TbU = tbu_zero ? 0.0 : LD(A.TbU, eu);
            w.uvel_init = revised ? LD(A.uvel_init, eu) : 0.0;
            w.vvel_init = revised ? LD(A.vvel_init, eu) : 0.0;
            w.uold = uold; w.vold = vold;
            w.sx0 = sx0; w.sx1 = sx1; w.sx2 = sx2; w.sx3 = sx3;
            w.sy0 = sy0; w.sy1 = sy1; w.sy2 = sy2; w.sy3 = sy3;
            if (tbu_zero) MM::template stepu<MODE, false>(A.p, w, o);
            else MM::template stepu<MODE, true>(A.p, w, o);
        };
        double u1 = u_p, v1 = v_p;                                  // off the ice: the velocity stays what it is
        const bool isU1 = (m_p & 2u) && lane >= 1 && lane <= 62 && r >= Y0;
        if (isU1) {
            UO o;
            momentum(ob - rowb, u_p, v_p, c1_sx0, c1_sx1, str1[2], n1_sx3, c1_sy0, str1[5], c1_sy2, n1_sy3, o);
            u1 = o.u; v1 = o.v;
        }

        // ---- S2: stress(k+2) on T(x, r-1) -----------------------------------------------------
        const bool act2 = (m_p & 1u) && lane >= 2 && lane <= 62 && r - 1 >= Y0;
        double str2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) str2[k] = 0.0;
        {
            const double u1L = lane_up(u1), v1L = lane_up(v1), hteL_p = lane_up(hte_p);
            const double u1L_p = lane_up(u1_p), v1L_p = lane_up(v1_p);
            if (act2) {
                double s[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) s[k] = stash[wv][(r - 1) & 1][k][lane];
                SI b;
                b.dxT = dxT_p; b.dyT = dyT_p; b.strength = strength_p;
                b.u_ij = u1; b.u_im = u1L; b.u_jm = u1_p; b.u_mm = u1L_p;
                b.v_ij = v1; b.v_im = v1L; b.v_jm = v1_p; b.v_mm = v1L_p;
                MM::metrics(hte_p, hteL_p, htn_p, htn_pp, A.deltaminEVP, b);
                MM::template stress<MODE>(A.p, b, s, str2);
                if (own_x && r - 1 < Y1) {
                    const unsigned es = ob - rowb;
#pragma unroll
                    for (int k = 0; k < 12; ++k) ST(A.sig_out[k], es, s[k]);
                    if (img) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) ST(A.sig_out[k], es + imgb, s[k]);
                    }
                }
            }
        }
        const double n2_sx3 = lane_dn(str2[3]), n2_sy3 = lane_dn(str2[7]);
        const double t2_sx1 = lane_dn(str2[1]), t2_sy2 = lane_dn(str2[6]);

        // ---- U2: stepu(k+2) on U(x, r-2) ------------------------------------------------------
        const bool isU2 = (m_pp & 2u) && own_x && r - 2 >= Y0;
        if (isU2) {
            UO o;
            momentum(ob - 2u * rowb, u1_p, v1_p, c2_sx0, c2_sx1, str2[2], n2_sx3, c2_sy0, str2[5], c2_sy2, n2_sy3, o);
            const unsigned eo = ob - 2u * rowb;
            ST(A.u_out, eo, o.u); ST(A.v_out, eo, o.v);
            if (img) { ST(A.u_out, eo + imgb, o.u); ST(A.v_out, eo + imgb, o.v); }
            if (A.last) {
                ST(A.strintx, eo, o.strintx); ST(A.strinty, eo, o.strinty);
                ST(A.taubx, eo, o.taubx); ST(A.tauby, eo, o.tauby);
            }
        }

        // ---- hand the row over to the next one ------------------------------------------------
        u_p = u0; v_p = v0;
        htn_pp = htn_p; htn_p = htn; hte_p = hte;
        dxT_p = dxT; dyT_p = dyT; strength_p = strength;
        m_pp = m_p; m_p = m;
        c1_sx0 = str1[0]; c1_sx1 = t1_sx1; c1_sy0 = str1[4]; c1_sy2 = t1_sy2;
        u1_p = u1; v1_p = v1;
        c2_sx0 = str2[0]; c2_sx1 = t2_sx1; c2_sy0 = str2[4]; c2_sy2 = t2_sy2;
    }
}

// ---------------------------------------------------------------------
// CICE block layout <-> rectangle layout (evp_host_march.cpp owns the geometry)
// ---------------------------------------------------------------------
struct CellMap {
    int e;          // element of the rectangle arrays, -1: no such element
    bool live;      // the element is a real cell of the rank's rectangle, or a halo element that images one
};

// block cell (b; i, j 1-based) -> rectangle element
__device__ __forceinline__ CellMap block_to_rect(const EvpMarchGeo &G, int b, int i, int j)
{
    const int2 o = G.blk_org[b];                 // rectangle coordinates of the block's first interior cell
    const int xs = o.x + (i - G.ilo), ys = o.y + (j - G.ilo);
    CellMap c;
    c.e = -1;
    c.live = false;
    if (xs < -EVP_MARCH_PAD || xs >= G.nxr + EVP_MARCH_PAD || ys < -EVP_MARCH_PAD || ys >= G.nyr + EVP_MARCH_PAD) return c;
    c.e = (ys + EVP_MARCH_PAD) * G.ldx + EVP_MARCH_PAD + xs;
    c.live = ((xs >= 0 && xs < G.nxr) || G.wrapx) && (ys >= 0 && ys < G.nyr);
    return c;
}

// rectangle element (x, y incl. halo) -> element of the block-layout arrays it takes its value from; -1: none (0).
// inside: interior cell of a block; wrap halo: the interior cell it images; first halo layer of a closed side: the
// ghost cell of the edge block (the caller's value there is what the reference reads, ice_dyn_evp.F90:867)
__device__ __forceinline__ int rect_to_block(const EvpMarchGeo &G, int x, int y, bool &is_cell)
{
    int xs = x, ys = y;
    if (G.wrapx) { if (xs < 0) xs += G.nxr; else if (xs >= G.nxr) xs -= G.nxr; }
    is_cell = xs >= 0 && xs < G.nxr && ys >= 0 && ys < G.nyr;
    if (xs < -1 || xs > G.nxr || ys < -1 || ys > G.nyr) return -1;
    const int bi = min(max(xs, 0) / G.bsx, G.nbx - 1), bj = min(max(ys, 0) / G.bsy, G.nby - 1);
    const int b = G.blkid[bj * G.nbx + bi];
    if (b < 0) return -1;
    const int i = G.ilo + (xs - bi * G.bsx), j = G.ilo + (ys - bj * G.bsy);      // 1-based; ilo-1 / ihi+1 on closed sides
    if (i < 1 || i > G.nxb || j < 1 || j > G.nyb) return -1;
    return b * G.plane + (j - 1) * G.nxb + (i - 1);
}

__global__ __launch_bounds__(256) void march_gather(EvpMarchGeo G, EvpMarchTab T, const uint8_t *__restrict__ mask_blk,
                                                    uint8_t *__restrict__ mask_rect)
{
    const int x = blockIdx.x * 256 + threadIdx.x - EVP_MARCH_PAD, y = blockIdx.y - EVP_MARCH_PAD;
    if (x + EVP_MARCH_PAD >= G.ldx) return;
    const int e = (y + EVP_MARCH_PAD) * G.ldx + EVP_MARCH_PAD + x;
    bool is_cell;
    const int s = rect_to_block(G, x, y, is_cell);
    for (int f = 0; f < T.n; ++f) {
        const double v = s >= 0 ? T.blk[f][s] : 0.0;
        T.rect[f][e] = v;
        if (T.rect2[f]) T.rect2[f][e] = v;
    }
    if (mask_rect) mask_rect[e] = (s >= 0 && is_cell) ? (mask_blk[s] & 3u) : 0;
}

// Is the caller's block-layout state the image of ONE global state?  The reference computes the T-cells of the
// north / east fringe (ihi+1, jhi+1) redundantly on every block from that block's own ghost storage
// (ice_dyn_shared.F90:740-749) and reads ghost velocities as the caller left them; the rectangle holds every cell
// once.  Both give the same bits iff the ghost values equal the cells they image, which CICE maintains (halo
// updates of iceTmask, strength and the velocities before the loop; stresses evolve alike on both copies).  A
// caller for which this does not hold (synthetic tests with independent holes in ghost masks) gets the one-
// subcycle kernels, which keep per-block ghost storage.  `which`: 1 per-call fields, 2 static fields.
__global__ __launch_bounds__(256) void march_check(EvpMarchGeo G, EvpMarchTab T, const uint8_t *__restrict__ mask_blk,
                                                   const uint8_t *__restrict__ mask_rect, int nuv, int nfringe,
                                                   unsigned *__restrict__ bad)
{
    // T.blk/T.rect: [0, nuv) compared on the whole ghost ring, [nuv, nuv + nfringe) on the fringe T-cells,
    // the rest (HTE, HTN) on the fringe and on column ilo-1 / row jlo-1
    const int i = blockIdx.x * 256 + threadIdx.x + 1, j = blockIdx.y + 1, b = blockIdx.z;
    const int4 r = G.blk[b];
    if (i < r.x - 1 || i > r.y + 1 || j < r.z - 1 || j > r.w + 1) return;
    const bool ghost = i < r.x || i > r.y || j < r.z || j > r.w;
    if (!ghost) return;
    const bool fringe = (i == r.y + 1 || j == r.w + 1) && i >= r.x && j >= r.z;
    const int s = b * G.plane + (j - 1) * G.nxb + (i - 1);
    const CellMap c = block_to_rect(G, b, i, j);
    unsigned nbad = 0;
    auto differs = [](double p, double q) { return __double_as_longlong(p) != __double_as_longlong(q); };
    // velocities: every ghost cell the stress of an owned T-cell reads (closed sides too: the rectangle took the value
    // from ONE of the ghost cells that image the position; they must all agree)
    if (c.e >= 0)
        for (int f = 0; f < nuv; ++f) nbad += differs(T.blk[f][s], T.rect[f][c.e]);
    if (c.live) {
        if (fringe) {
            for (int f = nuv; f < T.n; ++f) nbad += differs(T.blk[f][s], T.rect[f][c.e]);
            if (mask_blk) nbad += ((mask_blk[s] ^ mask_rect[c.e]) & 1u);
        } else if (i == r.x - 1 || j == r.z - 1) {
            for (int f = nuv + nfringe; f < T.n; ++f) nbad += differs(T.blk[f][s], T.rect[f][c.e]);
        }
    } else if (fringe && mask_blk) {
        nbad += (mask_blk[s] & 1u);          // a T-cell beyond a closed boundary that the reference would compute
    }
    if (nbad) atomicAdd(bad, nbad);
}

// rectangle -> block layout after the loop.  Velocities: every cell with a live source (interior, ghost images of
// cells of this rank: what the reference's halo update leaves, ice_dyn_evp.F90:908-910); stresses: the T-cells the
// reference updates (ilo..ihi+1 x jlo..jhi+1 where iceTmask); strintx/y, taubx/y: interior ice U-cells.
__global__ __launch_bounds__(256) void march_scatter(EvpMarchGeo G, EvpMarchTab T, const uint8_t *__restrict__ mask_blk,
                                                     int nuv, int nsig)
{
    const int i = blockIdx.x * 256 + threadIdx.x + 1, j = blockIdx.y + 1, b = blockIdx.z;
    const int4 r = G.blk[b];
    if (i < r.x - 1 || i > r.y + 1 || j < r.z - 1 || j > r.w + 1) return;
    const CellMap c = block_to_rect(G, b, i, j);
    if (!c.live) return;
    const int s = b * G.plane + (j - 1) * G.nxb + (i - 1);
    const unsigned m = mask_blk[s];
    for (int f = 0; f < nuv; ++f) T.blk[f][s] = T.rect[f][c.e];
    if ((m & 1u) && i >= r.x && j >= r.z)
        for (int f = nuv; f < nuv + nsig; ++f) T.blk[f][s] = T.rect[f][c.e];
    if ((m & 2u) && i >= r.x && i <= r.y && j >= r.z && j <= r.w)
        for (int f = nuv + nsig; f < T.n; ++f) T.blk[f][s] = T.rect[f][c.e];
}

}  // namespace

void evp_launch_march(const EvpMarch &A, bool strict, int mode, hipStream_t st)
{
    const dim3 grid((unsigned)((A.nitems + 3) / 4)), block(256);
    const bool lean = mode == 3 && (A.flags & EVP_F_WATER_IS_OCN) && (A.flags & EVP_F_TBU_ZERO) &&
                      !(std::getenv("CICE_EVP_HIP_MARCH_LEAN") && !std::atoi(std::getenv("CICE_EVP_HIP_MARCH_LEAN")));
#define EVP_MARCH_LAUNCH(S, M, L) hipLaunchKernelGGL((evp_march2<S, M, L>), grid, block, 0, st, A)
    if (strict) {
        if (lean) EVP_MARCH_LAUNCH(true, 3, true);
        else if (mode == 3) EVP_MARCH_LAUNCH(true, 3, false);
        else if (mode == 1) EVP_MARCH_LAUNCH(true, 1, false);
        else if (mode == 0) EVP_MARCH_LAUNCH(true, 0, false);
        else EVP_MARCH_LAUNCH(true, -1, false);
    } else {
        if (lean) EVP_MARCH_LAUNCH(false, 3, true);
        else if (mode == 3) EVP_MARCH_LAUNCH(false, 3, false);
        else if (mode == 1) EVP_MARCH_LAUNCH(false, 1, false);
        else if (mode == 0) EVP_MARCH_LAUNCH(false, 0, false);
        else EVP_MARCH_LAUNCH(false, -1, false);
    }
#undef EVP_MARCH_LAUNCH
}

void evp_launch_march_gather(const EvpMarchGeo &G, const EvpMarchTab &T, const uint8_t *mask_blk, uint8_t *mask_rect,
                             hipStream_t st)
{
    hipLaunchKernelGGL(march_gather, dim3((unsigned)((G.ldx + 255) / 256), (unsigned)G.rows), dim3(256), 0, st, G, T,
                       mask_blk, mask_rect);
}

void evp_launch_march_check(const EvpMarchGeo &G, const EvpMarchTab &T, const uint8_t *mask_blk, const uint8_t *mask_rect,
                            int nuv, int nfringe, unsigned *bad, hipStream_t st)
{
    hipLaunchKernelGGL(march_check, dim3((unsigned)((G.nxb + 255) / 256), (unsigned)G.nyb, (unsigned)G.nblocks), dim3(256),
                       0, st, G, T, mask_blk, mask_rect, nuv, nfringe, bad);
}

void evp_launch_march_scatter(const EvpMarchGeo &G, const EvpMarchTab &T, const uint8_t *mask_blk, int nuv, int nsig,
                              hipStream_t st)
{
    hipLaunchKernelGGL(march_scatter, dim3((unsigned)((G.nxb + 255) / 256), (unsigned)G.nyb, (unsigned)G.nblocks),
                       dim3(256), 0, st, G, T, mask_blk, nuv, nsig);
}
