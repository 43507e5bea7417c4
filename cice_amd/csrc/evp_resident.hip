// =====================================================================
// On-chip resident EVP subcycle (gfx950): ONE launch runs all `ndte` subcycles.
//
// The streaming kernel (evp_kernels.hip) re-reads ~25 MB and re-writes ~12 MB of
// state and per-call constants every subcycle although only the velocities change
// hands between cells.  For domains that fit on the chip (gx3/gx1/tx1-sized: a few
// hundred bytes per cell against 128 MB of vector registers) this kernel keeps, for
// the whole call, in the registers of the thread that owns the cell:
//     the 12 stress components, the metric terms, strength      (T-cell)
//     the momentum-equation operands                            (U-cell)
// Per subcycle a thread only reads the 4+4 velocities around its T-cell and writes
// its U-cell's new (u,v).  Velocities travel through L2 with write-through (sc1)
// stores and L1-bypassing (sc1) loads; tiles synchronise with the neighbour tiles
// that produce the velocities they read -- one monotonic flag per tile, no grid
// barrier.  Same arithmetic (evp_cell.inc), same tile shape and same ownership rules
// as the streaming kernel => same bits.
//
// Requires every workgroup to be co-resident (checked by the host from the occupancy
// query with a margin); every spin is bounded and raises an error word instead of
// hanging.
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evp_math.h"

namespace {

constexpr int RTY = 4;   // T rows per tile = waves per workgroup (one wave per SIMD)

__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool STRICT, int CAP, int LOGW>
__global__ __launch_bounds__(64 * RTY, 4) void evp_resident_tile(EvpArgs A, EvpResident R)
{
    using MM = Math<STRICT>;
    // LDS (dynamic, 16-byte aligned base, carve offsets multiples of 16):
    //   s_str [4][RTY][64]     partials a U-cell takes from the T-row above it (str 3,6,4,8)
    //   s_tc  [4][RTY][64]     T-cell constants read once per subcycle (strength, DminTarea, dxhy, dyhx)
    //   s_uc  [nu][RTY-1][64]  momentum-equation operands of the tile's U-cells (nu = 8..11)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *s_str = smem;                  // [4][256]
    double *s_tc = smem + 4 * 256;         // [4][256]
    double *s_uc = smem + 8 * 256;         // [nu][256]
    __shared__ int s_bad;

    // Tile = W x H T-cells, W*H = 256 (one per thread): lane -> (column, sub-row), wave -> row
    // group.  W = 64 is one row per wave (512-byte segments); W = 32 / 16 are squarer tiles:
    // fewer T-cells recomputed on the N/E fringe (256/(W-1)/(H-1): 1.35 / 1.18 / 1.14 per
    // U-cell) and less waste in the last tile column.
    constexpr int W = 1 << LOGW;
    constexpr int H = 256 / W;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int t = ty * 64 + tx;            // linear thread id == row*W + column
    const int tcol = tx & (W - 1);
    const int trow = t >> LOGW;
    // workgroup w runs on XCD w % 8 (observed dispatch rule; speed only): give each XCD one
    // contiguous band of tile rows, so that most neighbour tiles exchange velocities through
    // the XCD's own L2 instead of across the fabric
    int tile = blockIdx.x;
    if (A.xcdmap) {
        const int per = (A.ntiles + 7) >> 3;
        tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (tile >= A.ntiles) return;
    }
    const int bx = tile % A.gx;
    const int by = tile / A.gx;
    const int4 r = A.blk[0];
    const int i = r.x + bx * (W - 1) + tcol;
    const int j = r.z + by * (H - 1) + trow;
    const int nx = A.nx;
    const int c = (j - 1) * nx + (i - 1);
    const unsigned flags = A.flags;
    const bool water = !(flags & EVP_F_WATER_IS_OCN);
    const bool tbu = !(flags & EVP_F_TBU_ZERO);

    const bool inT = (i <= r.y + 1) && (j <= r.w + 1);
    unsigned m = 0;
    if (inT) m = A.mask[c];
    const bool actT = inT && (m & 1u);
    const bool isU = (tcol < W - 1) && (trow < H - 1) && (i <= r.y) && (j <= r.w) && (m & 2u);
    const bool own = (tcol < W - 1 || i == r.y + 1) && (trow < H - 1 || j == r.w + 1);

    // ---- state that stays on the CU for the whole call -------------------------------------
    typename MM::SI a;
    double s[12];
    if (actT) {
#pragma unroll
        for (int k = 0; k < 12; ++k) s[k] = R.tab[R.cur0 * 12 + k][c];
        a.dxT = A.dxT[c]; a.dyT = A.dyT[c];
        a.strength = A.strength[c];
        if ((flags & EVP_F_METRICS) && !(flags & EVP_F_DXHY_ARRAY)) {
            MM::metrics(A.HTE[c], A.HTE[c - 1], A.HTN[c], A.HTN[c - nx], A.deltaminEVP, a);
        } else {
            a.dxhy = A.dxhy[c]; a.dyhx = A.dyhx[c];
            a.cxp = A.cxp[c]; a.cyp = A.cyp[c]; a.cxm = A.cxm[c]; a.cym = A.cym[c];
            a.DminTarea = A.DminTarea[c];
        }
        s_tc[0 * 256 + t] = a.strength;
        s_tc[1 * 256 + t] = a.DminTarea;
        s_tc[2 * 256 + t] = a.dxhy;
        s_tc[3 * 256 + t] = a.dyhx;
    }
    double u_own = 0.0, v_own = 0.0;
    if (isU) {
        s_uc[0 * 256 + t] = A.vrelfac[c];
        s_uc[1 * 256 + t] = A.uocn[c];
        s_uc[2 * 256 + t] = A.vocn[c];
        s_uc[3 * 256 + t] = A.forcex[c];
        s_uc[4 * 256 + t] = A.forcey[c];
        s_uc[5 * 256 + t] = A.umassdti[c];
        s_uc[6 * 256 + t] = A.fm[c];
        s_uc[7 * 256 + t] = A.uarear[c];
        int row = 8;
        if (water) { s_uc[row * 256 + t] = A.waterx[c]; s_uc[(row + 1) * 256 + t] = A.watery[c]; row += 2; }
        if (tbu) s_uc[row * 256 + t] = A.TbU[c];
        u_own = R.u[R.cur0][c];
        v_own = R.v[R.cur0][c];
    }
    // ghost images of this U-cell (cyclic wrap), looked up once: at most three (corner cell)
    int img0 = -1, img1 = -1, img2 = -1;
    if (isU && (flags & EVP_F_PUSH) && (i == r.x || i == r.y || j == r.z || j == r.w)) {
        const int slots[4] = {(i == r.x) ? (j - r.z) : -1, (i == r.y) ? A.push_nj + (j - r.z) : -1,
                              (j == r.z) ? 2 * A.push_nj + (i - r.x) : -1,
                              (j == r.w) ? 2 * A.push_nj + A.push_ni + (i - r.x) : -1};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (slots[e] < 0) continue;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int v = A.push[slots[e] * 2 + w];
                if (v < 0) continue;
                if (img0 < 0) img0 = v;
                else if (img1 < 0) img1 = v;
                else img2 = v;
            }
        }
    }
    if (t == 0) s_bad = 0;
    __syncthreads();

    // ---- the subcycle loop (ice_dyn_evp.F90:859-913) ------------------------------------------
    for (int k = 0; k < R.ndte; ++k) {
        const int rb = (R.cur0 + k) & 1;
        const double *ur = R.u[rb];
        const double *vr = R.v[rb];
        double *uw = R.u[rb ^ 1];
        double *vw = R.v[rb ^ 1];

        if (k > 0 && !(R.dbg & 1)) {
            // wait until every tile this tile exchanges velocities with has finished
            // subcycle k-1 (a flag counts the subcycles its tile has completed)
            if (ty == 0 && tx < EVP_RES_NNB) {
                const int nb = R.nbr[tile * EVP_RES_NNB + tx];
                if (nb >= 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(R.flags + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > R.spin_limit ||
                            ((spins & 255u) == 0 &&
                             __hip_atomic_load(R.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            __hip_atomic_store(R.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s_bad = 1;
                            break;
                        }
                    }
                }
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __syncthreads();
            if (s_bad) return;   // uniform: every thread of the workgroup leaves together
        }

        double str[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) str[e] = 0.0;
        if (actT) {
            if (isU) { a.u_ij = u_own; a.v_ij = v_own; }
            else { a.u_ij = ld_sc1(ur + c); a.v_ij = ld_sc1(vr + c); }
            a.u_im = ld_sc1(ur + c - 1); a.v_im = ld_sc1(vr + c - 1);
            a.u_jm = ld_sc1(ur + c - nx); a.v_jm = ld_sc1(vr + c - nx);
            a.u_mm = ld_sc1(ur + c - nx - 1); a.v_mm = ld_sc1(vr + c - nx - 1);
            a.strength = s_tc[0 * 256 + t]; a.DminTarea = s_tc[1 * 256 + t];
            a.dxhy = s_tc[2 * 256 + t]; a.dyhx = s_tc[3 * 256 + t];
            MM::template stress<CAP>(A.p, a, s, str);
        }
        // partials for the U-cells of the row below go through LDS; the east neighbour's
        // partials (str 2,7) come from the next lane of the same wave
        s_str[0 * 256 + t] = str[2];
        s_str[1 * 256 + t] = str[5];
        s_str[2 * 256 + t] = str[3];
        s_str[3 * 256 + t] = str[7];
        const double sx1 = __shfl_down(str[1], 1);
        const double sy2 = __shfl_down(str[6], 1);
        __syncthreads();

        if (isU) {
            typename MM::UI q;
            typename MM::UO o;
            q.uold = u_own; q.vold = v_own;
            q.vrelfac = s_uc[0 * 256 + t]; q.uocn = s_uc[1 * 256 + t]; q.vocn = s_uc[2 * 256 + t];
            q.forcex = s_uc[3 * 256 + t]; q.forcey = s_uc[4 * 256 + t]; q.Umassdti = s_uc[5 * 256 + t];
            q.fm = s_uc[6 * 256 + t]; q.uarear = s_uc[7 * 256 + t];
            int row = 8;
            if (water) { q.waterx = s_uc[row * 256 + t]; q.watery = s_uc[(row + 1) * 256 + t]; row += 2; }
            else { q.waterx = q.uocn; q.watery = q.vocn; }
            q.TbU = tbu ? s_uc[row * 256 + t] : 0.0;
            q.uvel_init = A.p.revp != 0.0 ? A.uvel_init[c] : 0.0;
            q.vvel_init = A.p.revp != 0.0 ? A.vvel_init[c] : 0.0;
            q.sx0 = str[0]; q.sx1 = sx1;
            q.sx2 = s_str[0 * 256 + t + W]; q.sx3 = s_str[2 * 256 + t + W + 1];
            q.sy0 = str[4]; q.sy1 = s_str[1 * 256 + t + W];
            q.sy2 = sy2; q.sy3 = s_str[3 * 256 + t + W + 1];
            if (flags & EVP_F_TBU_ZERO) MM::template stepu<CAP, false>(A.p, q, o);
            else MM::template stepu<CAP, true>(A.p, q, o);
            u_own = o.u; v_own = o.v;
            st_sc1(uw + c, o.u);
            st_sc1(vw + c, o.v);
            if (img0 >= 0) { const double sg = (img0 & 1) ? -1.0 : 1.0; st_sc1(uw + (img0 >> 1), sg * o.u); st_sc1(vw + (img0 >> 1), sg * o.v); }
            if (img1 >= 0) { const double sg = (img1 & 1) ? -1.0 : 1.0; st_sc1(uw + (img1 >> 1), sg * o.u); st_sc1(vw + (img1 >> 1), sg * o.v); }
            if (img2 >= 0) { const double sg = (img2 & 1) ? -1.0 : 1.0; st_sc1(uw + (img2 >> 1), sg * o.u); st_sc1(vw + (img2 >> 1), sg * o.v); }
            if (k == R.ndte - 1 && !R.dry) {
                R.tab[24][c] = o.strintx; R.tab[25][c] = o.strinty;
                R.tab[26][c] = o.taubx; R.tab[27][c] = o.tauby;
            }
        }
        // publish: all velocity stores of this workgroup are out of the CU, then the flag
        if (!(R.dbg & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0)
            __hip_atomic_store(R.flags + tile, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- write the stresses back (both ping-pong copies: the host flips only u,v) -----------
    if (actT && own && !R.dry) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            R.tab[k][c] = s[k];
            R.tab[12 + k][c] = s[k];
        }
    }
}

}  // namespace

static size_t resident_lds_bytes(unsigned flags)
{
    int nu = 8 + ((flags & EVP_F_WATER_IS_OCN) ? 0 : 2) + ((flags & EVP_F_TBU_ZERO) ? 0 : 1);
    return sizeof(double) * 256 * (size_t)(8 + nu);
}

template <int LOGW>
static int occ(bool strict, int cap, size_t lds)
{
    int nb = 0;
#define EVP_OCC(S, C) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, evp_resident_tile<S, C, LOGW>, 64 * RTY, lds)
    hipError_t e;
    if (strict) e = cap == 3 ? EVP_OCC(true, 3) : cap == 1 ? EVP_OCC(true, 1) : cap == 0 ? EVP_OCC(true, 0) : EVP_OCC(true, -1);
    else e = cap == 3 ? EVP_OCC(false, 3) : cap == 1 ? EVP_OCC(false, 1) : cap == 0 ? EVP_OCC(false, 0) : EVP_OCC(false, -1);
#undef EVP_OCC
    return e == hipSuccess ? nb : 0;
}

int evp_resident_max_blocks_per_cu(bool strict, int cap, unsigned flags, int logw)
{
    const size_t lds = resident_lds_bytes(flags);
    return logw == 4 ? occ<4>(strict, cap, lds) : logw == 5 ? occ<5>(strict, cap, lds) : occ<6>(strict, cap, lds);
}

void evp_resident_geometry(int max_ni, int max_nj, int logw, int *gx, int *gy)
{
    const int W = 1 << logw, H = 256 / W;
    *gx = (max_ni + W - 2) / (W - 1);
    *gy = (max_nj + H - 2) / (H - 1);
}

template <int LOGW>
static void launch(const EvpArgs &A, const EvpResident &R, bool strict, int cap, hipStream_t st)
{
    dim3 grid(A.xcdmap ? ((A.ntiles + 7) / 8) * 8 : A.ntiles), block(64, RTY);
    const size_t lds = resident_lds_bytes(A.flags);
#define EVP_LAUNCH(S, C) hipLaunchKernelGGL((evp_resident_tile<S, C, LOGW>), grid, block, lds, st, A, R)
    if (strict) {
        if (cap == 3) EVP_LAUNCH(true, 3);
        else if (cap == 1) EVP_LAUNCH(true, 1);
        else if (cap == 0) EVP_LAUNCH(true, 0);
        else EVP_LAUNCH(true, -1);
    } else {
        if (cap == 3) EVP_LAUNCH(false, 3);
        else if (cap == 1) EVP_LAUNCH(false, 1);
        else if (cap == 0) EVP_LAUNCH(false, 0);
        else EVP_LAUNCH(false, -1);
    }
#undef EVP_LAUNCH
}

void evp_launch_resident(const EvpArgs &A0, const EvpResident &R, int max_ni, int max_nj, int logw,
                         bool strict, int cap, hipStream_t st)
{
    EvpArgs A = A0;
    evp_resident_geometry(max_ni, max_nj, logw, &A.gx, &A.gy);
    A.ntiles = A.gx * A.gy;
    A.xcdmap = R.xcdmap;
    if (logw == 4) launch<4>(A, R, strict, cap, st);
    else if (logw == 5) launch<5>(A, R, strict, cap, st);
    else launch<6>(A, R, strict, cap, st);
}
