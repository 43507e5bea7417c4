// =====================================================================
// On-chip resident EVP subcycle: tagged hand-off.
//
// (A first generation, retired in round 5, published a tile's velocities and then a flag; the
// neighbours polled the flag, passed a barrier and only then loaded the velocities: per subcycle
// one store drain, one flag round trip and one data round trip sat on the critical path,
// ~55 % of its time by shader-clock stamps.)  Here the data IS the flag: a velocity that another tile needs is published as one aligned 16-byte granule
//      { tag, value.lo, value.hi, tag }        tag = launch epoch | subcycle
// written by ONE write-through (sc1) dwordx4 store; the reader re-reads the granule
// (sc1, L1-bypassing) until both tags carry the subcycle it waits for -- a torn
// granule cannot pass (MI355X_MICROARCH.md, "R2": observed untorn, checked anyway).
// No flags, no store drain, no fences.  Velocities used inside the tile never leave
// the CU: they live in an LDS tile with a one-cell ring; only the ring is polled
// (<= 2(W+H) threads, one granule pair each).  Records are double-buffered by subcycle
// parity; a tile can be at most one subcycle ahead of a neighbour, so two suffice.
//
// Same arithmetic and ownership rules as the streaming kernel => same bits.  Every spin is bounded and raises the error word.
//
// Rim wave / interior waves (PERM, the 16 x 16 tile).  A subcycle of a tile is a dependent chain:
// (tripoleT, round 6: a top-row U-cell is the image of a cell of row NY-1; it takes -1 x that cell's new value after its own momentum
// step, through the record that cell publishes -- EvpResident2::tfold, the seam table names the source)
// neighbours publish -> ring poll -> stress -> barrier -> momentum step -> publish.  Only the T-cells
// on the rim of the tile read ring velocities (60 of 256 in a full 16 x 16 tile), and the ring has
// at most 64 entries: a host-built permutation puts those T-cells -- and the ring poll -- into
// wave 0, all other T-cells into waves 1-3.  Waves 1-3 start their stress update straight after the
// barrier that ends the previous momentum step, from velocities that never left the CU, while wave 0
// waits for the neighbours' records: the hand-off latency (about 1 us of the 5.5 us a subcycle took
// on gx1, three tiles per CU time-sharing the SIMDs) overlaps with three quarters of the tile's
// arithmetic instead of preceding all of it.  The lane -> cell map is arbitrary here because no
// array access of the loop is a global one; the eight stress-divergence partials travel through LDS
// by cell position.  Same arithmetic per cell => same bits.
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

// Debug hooks (lagging tiles, a workgroup that never shows up, A/B switches) and the per-phase cycle stamps exist in the
// TEST build's object of this file only (-DCICE_EVP_HIP_TESTING, linked into libcice_evp_hip_testing.so): the product
// kernel does not test R.dbg / R.prof at all.
#ifdef CICE_EVP_HIP_TESTING
#define RES_DBG(R) ((R).dbg)
#define RES_PROF(R) ((R).prof != nullptr)
#else
#define RES_DBG(R) 0
#define RES_PROF(R) false
#endif

#include "evp_math.h"

namespace {

constexpr int RTY = 4;   // waves per workgroup (one per SIMD)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// two adjacent granules (u then v) of one cell, write-through / L1-bypassing
__device__ __forceinline__ void st_rec2(void *p, v4u a, v4u b)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1"
                 :
                 : "v"(p), "v"(a), "v"(b)
                 : "memory");
}
__device__ __forceinline__ void ld_rec2(const void *p, v4u &a, v4u &b)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(p)
                 : "memory");
}
// the same at system scope: records another GPU writes into this one's buffer / this one into a peer's
__device__ __forceinline__ void st_rec2_sys(void *p, v4u a, v4u b)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1"
                 :
                 : "v"(p), "v"(a), "v"(b)
                 : "memory");
}
__device__ __forceinline__ void ld_rec2_sys(const void *p, v4u &a, v4u &b)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(p)
                 : "memory");
}
__device__ __forceinline__ v4u pack_rec(double x, unsigned tag)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    v4u r;
    r.x = tag; r.y = (unsigned)bits; r.z = (unsigned)(bits >> 32); r.w = tag;
    return r;
}
__device__ __forceinline__ double unpack_rec(v4u r)
{
    return __longlong_as_double((long long)(((unsigned long long)r.z << 32) | r.y));
}

// quad permute of a double (one DPP move per 32-bit half): lane l of every four adjacent lanes reads lane (l ^ X)
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
constexpr int QP_X1 = 0xB1, QP_X2 = 0x4E, QP_X3 = 0x1B;      // quad_perm [1,0,3,2], [2,3,0,1], [3,2,1,0]

// REMOTE: some ghost cells mirror cells of OTHER ranks (GPUs of the same node).  Their records
// live in this rank's record buffer and are written by the producing GPU with plain 16-byte
// stores over xGMI (the buffer is mapped there through HIP IPC); an edge cell with remote images
// stores its record into the peers' buffers as well.  Nothing else changes: the halo exchange is
// a remote store plus the ring poll that is there anyway.
// COOP (16 x 16 tiles, strict arithmetic, the default scalars -- CAP == 3 -- only): the T-cells that read ring velocities are
// not updated by one thread each after the ring poll -- 700 instructions of one wave on the critical path of every
// subcycle, while the other three waves of the tile wait at the barrier -- but by FOUR lanes each, one per corner, spread over
// all four waves: a lane works out the strain rates, Delta, the viscosities and the three stresses of its corner (the corner's
// operands and signs selected per lane, the operations and their order those of stress_cell: a - b*c == a + (-b)*c, x + y == y + x
// bit for bit), fetches the other three corners' stresses with quad-permute moves, and forms the two stress-divergence partials
// that carry its corner's coefficients.  The cell's stresses live three per lane in those quads for the whole call.
template <bool STRICT, int CAP, int LOGW, bool REMOTE, bool COOP = false>
__global__ __launch_bounds__(64 * RTY, REMOTE ? 2 : 3) void evp_resident2_tile(EvpArgs A, EvpResident2 R)
{
    static_assert(!COOP || (STRICT && CAP == 3 && LOGW == 4), "COOP: 16 x 16 tiles, strict build, default scalars");
    using MM = Math<STRICT>;
    constexpr int W = 1 << LOGW;
    constexpr int H = 256 / W;
    constexpr int LW = W + 1;                 // LDS velocity tile: (H+1) x (W+1), origin (-1,-1)
    constexpr int NUV = ((H + 1) * LW + 7) & ~7;
    constexpr bool PERM = (LOGW == 4);        // rim wave / interior waves (see the header)
    constexpr int NSTR = PERM ? 6 : 4;        // planes of stress-divergence partials that travel through LDS
    constexpr int SW = PERM ? W + 1 : W;      // row stride of a plane (PERM: odd, a column of cells is not one bank)
    constexpr int SP = SW * H;                // plane size
    // LDS (dynamic): s_str[NSTR][SP] | s_tc[4][256] | s_u[NUV] | s_v[NUV] | s_uc[nu][256]
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *s_str = smem;
    double *s_tc = smem + NSTR * SP;
    double *s_u = s_tc + 4 * 256;
    double *s_v = s_u + NUV;
    double *s_uc = s_v + NUV;
    // COOP: per rim T-cell its ten operands (s_rc[k * 64 + g]) and the two partials its own U-cell's thread keeps (s_r04), behind s_uc
    double *s_rc = nullptr, *s_r04 = nullptr;
    __shared__ int s_bad;

    __shared__ int s_chunk[4], s_simd[4], s_cu;

    const int tx = threadIdx.x, ty = threadIdx.y;
    const int t = ty * 64 + tx;
    // launch order = heaviest tiles first (host-sorted by active waves, see resident2_order): the
    // workgroups that end up third on a CU are then the cheap ones (land, partial edge tiles)
    const int tile = R.order ? R.order[blockIdx.x] : (int)blockIdx.x;
    bool no_ucell;
    {
        // A tile that owns no U-cell leaves at once.  The tile grid is uniform over the blocks of a rank (sized for the
        // largest one), so a block that is a row or column shorter than the largest gets a last tile row / column that
        // starts beyond its last U-cell: all such a tile holds is the T-row jhi+1 (or column ihi+1), which the tile below
        // (left of) it computes and stores as well.  It would poll its neighbours' records every subcycle while nobody
        // polls any of its own -- a reader without back-pressure on its producers: they may run two subcycles ahead of
        // it and overwrite a record it has not read, after which it waits for ever (seen as "a wait gave up ... tag
        // seen 0x1002, wanted 0x1000" in the probe of a 3 x 3-block decomposition whose top blocks are one row short;
        // on a single rank the call then fell back to the streaming kernel, across ranks it failed).
        const int pb = A.gx * A.gy;
        const int4 rb = A.blk[tile / pb];
        // (test build, RES_DEBUG bit 512: keep them, bit 256: and let them lag -- to show the hazard, tests/test_gpu_parity.py)
        no_ucell = rb.x + ((tile % pb) % A.gx) * (W - 1) > rb.y || rb.z + ((tile % pb) / A.gx) * (H - 1) > rb.w;
        if (no_ucell && !(RES_DBG(R) & 512)) return;
    }
    // PERM: which quarter ("chunk") of the tile's permuted cell list this wave takes.  The host packs
    // the ice cells of a tile into its first chunks (a coastal tile then costs one or two waves, not
    // four).  A SIMD issues for one wave at a time and all tiles advance in lock step, so the SIMD with
    // the most ice-holding waves paces the whole grid: the workgroups that share a CU put their active
    // chunks on the CU's least loaded SIMDs (the wave -> SIMD map is the hardware's: read from HW_ID;
    // a small per-CU record under a lock, once per launch).  Speed only: any chunk -> wave map is correct.
    int tq = t;
    int cu_rank = 0;                          // how many workgroups had reached this CU before this one
    if (PERM && R.cuload && !(RES_DBG(R) & 64)) {
        const int wave = t >> 6;
        if ((t & 63) == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            s_simd[wave] = (int)((hw >> 4) & 3u);
            if (wave == 0) s_cu = (int)(((xcc & 7u) << 8) | ((hw >> 8) & 0xffu));   // XCC | SE, SH, CU
        }
        __syncthreads();
        if (t == 0) {
            int *L = R.cuload + 8 * s_cu;         // [0] lock, [1] launch stamp, [2..5] ice-holding waves per SIMD, [6] workgroups so far
            const int nact = R.nact[tile];        // active chunks of this tile: 0 .. nact-1
            int chunk_of[4] = {0, 1, 2, 3};
            unsigned spins = 0;
            bool locked = false;
            while (spins++ < 200000u) {
                if (atomicCAS(&L[0], 0, 1) == 0) { locked = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (locked) {
                int ld[4];
                const int stamp = __hip_atomic_load(&L[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool fresh = stamp != (int)R.tag_base;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ld[q] = fresh ? 0 : __hip_atomic_load(&L[2 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int rank = fresh ? 0 : __hip_atomic_load(&L[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_simd[0] |= rank << 8;               // (the arrival rank travels to the other threads with it)
                bool used[4] = {false, false, false, false};
                for (int ck = 0; ck < 4; ++ck) {      // active chunks first, each to the free wave on the least loaded SIMD
                    int best = -1;
                    for (int w = 0; w < 4; ++w) {
                        if (used[w]) continue;
                        if (best < 0 || ld[s_simd[w] & 3] < ld[s_simd[best] & 3]) best = w;
                    }
                    used[best] = true;
                    chunk_of[best] = ck;
                    if (ck < nact) ld[s_simd[best] & 3] += 1;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) __hip_atomic_store(&L[2 + q], ld[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&L[6], rank + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&L[1], (int)R.tag_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&L[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) s_chunk[w] = chunk_of[w];
        }
        __syncthreads();
        tq = s_chunk[wave] * 64 + (t & 63);
        cu_rank = s_simd[0] >> 8;
        // (tried, no measurable effect on gx1: one s_setprio level per workgroup on all SIMDs of its CU, in
        // either order; s_setprio 3 for the rim wave between its poll and the barrier; rim waves of the
        // workgroups of a CU on different SIMDs; s_sleep 0/3/8 in the poll loop -- all within +-2 %)
    }
    // which T-cell of the tile this thread owns: row-major (lane = column) unless permuted
    const int pos = PERM ? (int)R.perm[tile * 256 + tq] : t;      // == trow*W + tcol
    const int tcol = pos & (W - 1);
    const int trow = pos >> LOGW;
    const int sp = trow * SW + tcol;          // this cell in a plane of s_str
    // PERM: waves below `late_waves` hold the T-cells that read ring velocities and do the ring poll;
    // with exactly one such wave (every 16 x 16 tile: rim <= 60 cells, ring <= 64 entries) the others
    // never wait for it before their stress update ("split")
    const int late_waves = PERM ? (int)R.late_waves[tile] : 4;
    const bool split = PERM && late_waves <= 1 && !(RES_DBG(R) & 32);
    if ((RES_DBG(R) & 16) && tile == 1 && !R.dry) return;   // test hook: one workgroup "never becomes resident" (after the probes)
    const int per_blk = A.gx * A.gy;
    const int bz = tile / per_blk;                // CICE block of this rank
    const int bx = (tile % per_blk) % A.gx;
    const int by = (tile % per_blk) / A.gx;
    const int4 r = A.blk[bz];
    const int i0 = r.x + bx * (W - 1), j0 = r.z + by * (H - 1);
    const int i = i0 + tcol;
    const int j = j0 + trow;
    const int nx = A.nx, ny = A.ny;
    const int cb = bz * (int)A.plane;             // first cell of the block in every (nx, ny, nblocks) array
    const int c = cb + (j - 1) * nx + (i - 1);
    const int li = (trow + 1) * LW + (tcol + 1);   // this cell in the LDS velocity tile
    const unsigned flags = A.flags;
    const bool water = !(flags & EVP_F_WATER_IS_OCN);
    const bool tbu = !(flags & EVP_F_TBU_ZERO);
    if (COOP) {
        s_rc = s_uc + (8 + (water ? 2 : 0) + (tbu ? 1 : 0)) * 256;
        s_r04 = s_rc + 10 * 64;
    }

    const bool inT = (i <= r.y + 1) && (j <= r.w + 1);
    unsigned m = 0;
    if (inT) m = A.mask[c];
    const bool actT = inT && (m & 1u);
    const bool isU = (tcol < W - 1) && (trow < H - 1) && (i <= r.y) && (j <= r.w) && (m & 2u);
    const bool own = (tcol < W - 1 || i == r.y + 1) && (trow < H - 1 || j == r.w + 1);
    // Every U-cell another tile's ring mirrors publishes a record every subcycle, ice or not, and
    // every ring entry that has a producer is polled every subcycle: a tile then cannot run more
    // than one subcycle ahead of ANY tile that still has to read its records (the two-buffer
    // record scheme depends on that; making either side depend on the ice mask would let a tile
    // next to open water run ahead of its reader and overwrite a record that was never read).
    const bool ownU = (tcol < W - 1) && (trow < H - 1) && (i <= r.y) && (j <= r.w);
    const bool pub = ownU && (R.pubmap[c] != 0);
    const int par0 = R.par0;                      // record buffer of subcycle index 0 in this launch
    // COOP: the first nlate entries of the tile's permuted cell list are its rim T-cells with ice (resident2_order); the
    // thread that holds such a cell leaves its stress update to the quad g = its list index, lanes 4g .. 4g+3 of the workgroup
    const int nlate = COOP ? (int)R.nlate[tile] : 0;
    const bool rim_owner = COOP && tq < nlate;
    const bool actN = actT && !rim_owner;         // this thread updates its own T-cell
    const int g = t >> 2, qc = t & 3;             // quad, corner: 0 NE, 1 NW, 2 SW, 3 SE (the order of the stress arrays)
    const bool quad = COOP && g < nlate;
    int gc = 0;
    unsigned g_vel = 0, g_out = 0;                // packed LDS indices: the three velocity cells (10 bits each), the two outputs (16 bits each)
    bool gown = false;
    double sr0 = 0.0, sr1 = 0.0, sr2 = 0.0;       // stressp_q, stressm_q, stress12_q of the quad's cell
    if (quad) {
        const int gpos = (int)R.perm[tile * 256 + g];
        const int gcol = gpos & (W - 1), grow = gpos >> LOGW;
        const int gi = i0 + gcol, gj = j0 + grow;
        gc = cb + (gj - 1) * nx + (gi - 1);
        gown = (gcol < W - 1 || gi == r.y + 1) && (grow < H - 1 || gj == r.w + 1);
        const int gli = (grow + 1) * LW + (gcol + 1);
        const int ij = gli, im = gli - 1, jm = gli - LW, mm = gli - LW - 1;
        // the corner's operands: first / second velocity of the u-pair, second of the v-pair (the first is the same cell)
        const int g_oa = qc == 0 ? ij : qc == 1 ? im : qc == 2 ? mm : jm;
        const int g_ob = qc == 0 ? im : qc == 1 ? ij : qc == 2 ? jm : mm;
        const int g_od = qc == 0 ? jm : qc == 1 ? mm : qc == 2 ? im : ij;
        g_vel = (unsigned)g_oa | ((unsigned)g_ob << 10) | ((unsigned)g_od << 20);
        // where the lane's two partials go: NE str(1), str(5) -> the cell's own thread (s_r04); NW str(2), str(7) -> planes 4, 5;
        // SW str(4), str(8) -> planes 2, 3; SE str(3), str(6) -> planes 0, 1
        const int gsp = grow * SW + gcol;
        const int off04 = (int)(s_r04 - s_str);   // (indices relative to s_str: the NE lane's two go to s_r04)
        const int g_ox = qc == 0 ? off04 + g : (qc == 1 ? 4 : qc == 2 ? 2 : 0) * SP + gsp;
        const int g_oy = qc == 0 ? off04 + 64 + g : (qc == 1 ? 5 : qc == 2 ? 3 : 1) * SP + gsp;
        g_out = (unsigned)g_ox | ((unsigned)g_oy << 16);
        sr0 = R.tab[R.cur0 * 12 + qc][gc];
        sr1 = R.tab[R.cur0 * 12 + 4 + qc][gc];
        sr2 = R.tab[R.cur0 * 12 + 8 + qc][gc];
        if (qc == 0) {
            typename MM::SI b;
            b.dxT = A.dxT[gc]; b.dyT = A.dyT[gc];
            b.strength = A.strength[gc];
            if ((flags & EVP_F_METRICS) && !(flags & EVP_F_DXHY_ARRAY)) {
                MM::metrics(A.HTE[gc], A.HTE[gc - 1], A.HTN[gc], A.HTN[gc - nx], A.deltaminEVP, b);
            } else {
                b.dxhy = A.dxhy[gc]; b.dyhx = A.dyhx[gc];
                b.cxp = A.cxp[gc]; b.cyp = A.cyp[gc]; b.cxm = A.cxm[gc]; b.cym = A.cym[gc];
                b.DminTarea = A.DminTarea[gc];
            }
            s_rc[0 * 64 + g] = b.strength; s_rc[1 * 64 + g] = b.DminTarea;
            s_rc[2 * 64 + g] = b.dxT; s_rc[3 * 64 + g] = b.dyT;
            s_rc[4 * 64 + g] = b.cxp; s_rc[5 * 64 + g] = b.cyp; s_rc[6 * 64 + g] = b.cxm; s_rc[7 * 64 + g] = b.cym;
            s_rc[8 * 64 + g] = b.dxhy; s_rc[9 * 64 + g] = b.dyhx;
        }
    }

    // ---- state that stays on the CU for the whole call -------------------------------------
    typename MM::SI a;
    double s[12];
    if (actN) {
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
    }
    // the velocity tile incl. its ring starts from the input arrays (ghost cells are valid on entry)
    {
        const double *u0 = R.u[R.cur0], *v0 = R.v[R.cur0];
        for (int q = t; q < (H + 1) * LW; q += 256) {
            const int pr = q / LW - 1, pc = q % LW - 1;
            const int gi = i0 + pc, gj = j0 + pr;
            double uu = 0.0, vv = 0.0;
            if (gi >= 1 && gi <= nx && gj >= 1 && gj <= ny) {
                const int cp = cb + (gj - 1) * nx + (gi - 1);
                uu = u0[cp]; vv = v0[cp];
            }
            s_u[q] = uu; s_v[q] = vv;
        }
    }
    // ghost images of this U-cell (cyclic wrap), looked up once: at most three (corner cell)
    int img0 = -1, img1 = -1, img2 = -1;
    // tripole seam (row jhi lies on the fold): this thread's U-cell takes part in the pair average
    // after every momentum step, ice or not (ice_boundary.F90:1630-1649) -- role 1/2: low/high index
    // of a pair, 3: pole point; partner = the other cell of the pair
    int seam_role = 0, seam_partner = -1;
    if (R.seam && ownU && j == r.w) {
        const int sv = R.seam[bz * nx + i - 1];
        seam_role = sv & 3;
        seam_partner = sv >> 2;
    }
    const bool isSeam = seam_role != 0;
    if (R.img3) {
        // tripole: ghost images are not confined to the block edge (ghost row NY+1 mirrors row NY-1):
        // per-cell table, at most three images
        if (ownU) { img0 = R.img3[3 * c]; img1 = R.img3[3 * c + 1]; img2 = R.img3[3 * c + 2]; }
    } else
    if (ownU && (flags & EVP_F_PUSH) && (i == r.x || i == r.y || j == r.z || j == r.w)) {
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
    // images of this U-cell on other ranks: record address (parity 0) in the peer's buffer
    char *rp0 = nullptr, *rp1 = nullptr, *rp2 = nullptr;
    size_t rs0 = 0, rs1 = 0, rs2 = 0;
    double rg0 = 1.0, rg1 = 1.0, rg2 = 1.0;       // an image across the tripole fold takes the negative
    // fold row split over ranks: where this seam cell's RAW record goes on other ranks
    char *rq0 = nullptr, *rq1 = nullptr, *rq2 = nullptr;
    size_t rqs0 = 0, rqs1 = 0, rqs2 = 0;
    if (REMOTE) {
        if (ownU) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int2 v = R.rimg[3 * (size_t)c + e];      // per-cell table: {peer index (| 256: negative), ghost cell at that peer}
                if (v.x < 0) continue;
                const int pe = v.x & 255;
                char *ptr = (char *)R.peer_rec[pe] + 32 * (size_t)v.y;
                const size_t st = R.peer_rstride[pe];
                const double sg = (v.x & 256) ? -1.0 : 1.0;
                if (!rp0) { rp0 = ptr; rs0 = st; rg0 = sg; }
                else if (!rp1) { rp1 = ptr; rs1 = st; rg1 = sg; }
                else { rp2 = ptr; rs2 = st; rg2 = sg; }
            }
            if (R.rraw && isSeam) {
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const int2 v = R.rraw[3 * (size_t)c + e];
                    if (v.x < 0) continue;
                    char *ptr = (char *)R.peer_raw[v.x] + 32 * (size_t)v.y;
                    const size_t st = R.peer_raw_stride[v.x];
                    if (!rq0) { rq0 = ptr; rqs0 = st; }
                    else if (!rq1) { rq1 = ptr; rqs1 = st; }
                    else { rq2 = ptr; rqs2 = st; }
                }
            }
        }
    }
    // the partner of a seam pair on another rank: its raw record arrives in THIS rank's rec_raw buffer at a staging slot
    const bool seam_remote = REMOTE && isSeam && seam_role != 3 && seam_partner >= (int)(A.plane * (size_t)R.nblocks);
    const bool rpub = REMOTE && rp0 != nullptr;    // publishes every subcycle, ice or not
    // my ring entry: which cell of the LDS ring do I refresh, from which record
    int ring_cp = -1, ring_li = 0;
    bool ring_remote = false;
    if (tq < R.ring_cnt[tile]) {     // (PERM: ring entries and the T-cells that read them share the wave that took chunk 0)
        const int4 e = R.ring[tile * EVP_RES2_RING + tq];   // x: cell whose record is polled, y: LDS index, z: producing U-cell
        // has a producer on this GPU: refreshed every subcycle -- unless the producer's tile holds no ice and does not run
        // (EvpResident2::live): its cells keep the values the velocity tile was filled with above
        if (e.z >= 0 && (!R.live || R.live[R.celltile[e.z]])) { ring_cp = e.x; ring_li = e.y; }
        if (REMOTE && e.z == -2) { ring_cp = e.x; ring_li = e.y; ring_remote = true; }   // produced on another rank
    }
    // first wait that gives up records where and on what: err[0] kind (1 ring of this GPU, 2 ring of
    // another rank, 3 fold-row partner, 4 final ghosts), [1] tile, [2] subcycle, [3] cell, [4] tag seen, [5] tag wanted
    auto give_up_note = [&](int kind, int k, int cell, unsigned seen, unsigned wanted) {
        if (atomicCAS(R.err, 0, kind) == 0) {
            R.err[1] = tile; R.err[2] = k; R.err[3] = cell; R.err[4] = (int)seen; R.err[5] = (int)wanted;
        }
    };
    auto publish_remote = [&](int par, double uu, double vv, unsigned tag) {
        st_rec2_sys(rp0 + (size_t)par * rs0, pack_rec(rg0 * uu, tag), pack_rec(rg0 * vv, tag));
        if (rp1) st_rec2_sys(rp1 + (size_t)par * rs1, pack_rec(rg1 * uu, tag), pack_rec(rg1 * vv, tag));
        if (rp2) st_rec2_sys(rp2 + (size_t)par * rs2, pack_rec(rg2 * uu, tag), pack_rec(rg2 * vv, tag));
    };
    // one poll of a record written by another rank: bounded by wall-clock time (ranks reach
    // evp() at different moments), not by a spin count
    auto poll_remote = [&](const v4u *rec, unsigned want, v4u &ra, v4u &rb, int kind = 2, int cell = -1) -> bool {
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            ld_rec2_sys(rec, ra, rb);
            if (ra.x == want && ra.w == want && rb.x == want && rb.w == want) return true;
            if ((++spins & 255u) == 0 &&
                (wall_clock64() - t0 > R.timeout_ticks ||
                 __hip_atomic_load(R.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                give_up_note(kind, (int)(want - R.tag_base), cell >= 0 ? cell : ring_cp, ra.x, want);
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    if (t == 0) s_bad = 0;
    __syncthreads();

    // bound of a wait on a record of this GPU: spin count (workgroups not co-resident -> fail fast) on
    // a rank without remote neighbours, wall-clock time otherwise
    unsigned long long t_wait0 = 0;
    auto gave_up = [&](unsigned spins, unsigned long long t0) -> bool {
        if (REMOTE) {
            if ((spins & 255u) != 0) return false;
            return wall_clock64() - t0 > R.timeout_ticks ||
                   __hip_atomic_load(R.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        }
        return spins > R.spin_limit ||
               ((spins & 255u) == 0 && __hip_atomic_load(R.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
    };
    double u_own = 0.0, v_own = 0.0;
    if (ownU) { u_own = s_u[li]; v_own = s_v[li]; }
    // initial records (subcycle tag 0) so that the neighbours' first ring refresh finds them
    {
        const unsigned tag = R.tag_base;
        v4u *r0 = (v4u *)R.rec[par0 & 1];
        if (rpub) publish_remote(par0 & 1, u_own, v_own, tag);
        if (pub) st_rec2(r0 + 2 * (size_t)c, pack_rec(u_own, tag), pack_rec(v_own, tag));
        if (ownU) {
            if (img0 >= 0) { const double sg = (img0 & 1) ? -1.0 : 1.0; st_rec2(r0 + 2 * (size_t)(img0 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
            if (img1 >= 0) { const double sg = (img1 & 1) ? -1.0 : 1.0; st_rec2(r0 + 2 * (size_t)(img1 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
            if (img2 >= 0) { const double sg = (img2 & 1) ? -1.0 : 1.0; st_rec2(r0 + 2 * (size_t)(img2 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
        }
    }

    // ---- the subcycle loop (ice_dyn_evp.F90:859-913) ------------------------------------------
    // optional phase stamps of every wave (CICE_EVP_HIP_RES_PROF=1, tools/resident_phases.py): shader cycles
    // spent in [ring poll | stress | wait at the barrier before the momentum step | momentum step, seam,
    // publish | wait at the barrier that ends the subcycle]
    const bool prof = RES_PROF(R);
    unsigned long long pc0 = 0, pacc0 = 0, pacc1 = 0, pacc2 = 0, pacc3 = 0, pacc4 = 0;
    if (prof) pc0 = __builtin_readcyclecounter();
#define EVP_STAMP(acc)                                                  \
    if (prof) {                                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        acc += now_ - pc0;                                              \
        pc0 = now_;                                                     \
    }
    // Tried in round 4, measured on gx1 (tools/resident_phases.py), none moved the subcycle by more than
    // 2 %: s_setprio 3 for the rim wave / 0 for interior stress / 2 for every wave's momentum step; the ring records of the
    // next subcycle requested before the momentum step (the hand-off is on a dependency CYCLE between neighbours, not on one
    // tile's critical path: nothing is there earlier); fine-grained / uncached memory for the record buffers; no s_sleep.
    // Two polls in flight per lane (a second poll a few hundred cycles behind the first, each re-issued once looked at; one
    // asm block on fixed registers, compares included -- the compiler copies a poll's registers at a loop merge while the
    // load is in flight, in every form it was given): bit-identical, and SLOWER -- poll phase 4476 -> 5131 cycles, 4.69 ->
    // 4.90 us per subcycle (gpurun_out/r4i): twice the poll traffic costs more than the earlier sighting saves.
    for (int k = 0; k < R.ndte; ++k) {
        const unsigned want = R.tag_base + (unsigned)k;       // tag of the velocities subcycle k reads
        const v4u *rd = (const v4u *)R.rec[(k + par0) & 1];
        v4u *wr = (v4u *)R.rec[((k + par0) & 1) ^ 1];

        if (((RES_DBG(R) & 8) && (tile & 3) == 1) || ((RES_DBG(R) & 256) && no_ucell)) {      // robustness test: every fourth tile lags by ~10 us per subcycle
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < 1000ull) __builtin_amdgcn_s_sleep(8);
        }
        // refresh the ring of the velocity tile from the neighbours' records (split: only wave 0 holds
        // ring entries and T-cells that read them; nobody else waits here)
        if (REMOTE && ring_remote) {
            v4u ra, rb;
            if (!poll_remote(rd + 2 * (size_t)ring_cp, want, ra, rb)) s_bad = 1;
            s_u[ring_li] = unpack_rec(ra);
            s_v[ring_li] = unpack_rec(rb);
        } else if (ring_cp >= 0 && !(RES_DBG(R) & 2)) {
            v4u ra, rb;
            unsigned spins = 0;
            if (REMOTE) t_wait0 = wall_clock64();
            for (;;) {
                ld_rec2(rd + 2 * (size_t)ring_cp, ra, rb);
                if ((ra.x == want && ra.w == want && rb.x == want && rb.w == want) || (RES_DBG(R) & 1)) break;
                // a local neighbour may itself be waiting for another rank: with remote neighbours
                // every wait is bounded by wall-clock time, not by a spin count
                if (gave_up(++spins, t_wait0)) {
                    give_up_note(1, k, ring_cp, ra.x, want);
                    s_bad = 1;
                    break;
                }
                if (RES_DBG(R) & 4) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
            }
            s_u[ring_li] = unpack_rec(ra);
            s_v[ring_li] = unpack_rec(rb);
        }
        if (!split) {
            __syncthreads();
            if (s_bad) return;   // uniform: every thread of the workgroup leaves together
        } else {
            // ring cells are written and read by wave 0 only: LDS operations of one wave complete in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        EVP_STAMP(pacc0)

        double str[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) str[e] = 0.0;
        if (actN) {
            a.u_ij = s_u[li]; a.v_ij = s_v[li];
            a.u_im = s_u[li - 1]; a.v_im = s_v[li - 1];
            a.u_jm = s_u[li - LW]; a.v_jm = s_v[li - LW];
            a.u_mm = s_u[li - LW - 1]; a.v_mm = s_v[li - LW - 1];
            a.strength = s_tc[0 * 256 + t]; a.DminTarea = s_tc[1 * 256 + t];
            a.dxhy = s_tc[2 * 256 + t]; a.dyhx = s_tc[3 * 256 + t];
            MM::template stress<CAP>(A.p, a, s, str);
        }
        EVP_STAMP(pacc1)
        double sx1, sy2;
        if (PERM) {      // by cell position: the U-cell's thread need not sit next to its T neighbours
            if (!rim_owner) {      // (COOP: a rim cell's partials come from its quad)
                s_str[0 * SP + sp] = str[2];
                s_str[1 * SP + sp] = str[5];
                s_str[2 * SP + sp] = str[3];
                s_str[3 * SP + sp] = str[7];
                s_str[4 * SP + sp] = str[1];
                s_str[5 * SP + sp] = str[6];
            }
            __syncthreads();
            if (split && s_bad) return;   // set by wave 0 before the barrier
            if (COOP) {
                EVP_STAMP(pacc2)      // (stamps: the wait at this barrier counts as "wait B1", the quads' work as "stress")
                // the ring is in LDS for everybody now: the rim T-cells, one corner per lane, all four waves
                if (quad) {
                    typename MM::CI cn;
                    double KX, K12X, KYP, K12Y;
                    MM::corner_operands(qc, s_rc[2 * 64 + g], s_rc[3 * 64 + g], s_rc[4 * 64 + g], s_rc[5 * 64 + g], s_rc[6 * 64 + g],
                                        s_rc[7 * 64 + g], cn, KX, K12X, KYP, K12Y);
                    unsigned pv = g_vel, po = g_out;
                    asm volatile("" : "+v"(pv), "+v"(po));      // (unpacked here, every subcycle: not five registers for the whole call)
                    const int g_oa = (int)(pv & 1023u), g_ob = (int)((pv >> 10) & 1023u), g_od = (int)(pv >> 20);
                    cn.ua = s_u[g_oa]; cn.va = s_v[g_oa];
                    cn.ub = s_u[g_ob]; cn.vb = s_v[g_ob];
                    cn.ud = s_u[g_od]; cn.vd = s_v[g_od];
                    cn.strength = s_rc[0 * 64 + g]; cn.DminTarea = s_rc[1 * 64 + g];
                    MM::template corner<CAP>(A.p, cn, sr0, sr1, sr2);
                    double X, Y;
                    MM::partials(sr0, quad_perm<QP_X1>(sr0), quad_perm<QP_X2>(sr0), quad_perm<QP_X3>(sr0),
                                 sr1, quad_perm<QP_X1>(sr1), quad_perm<QP_X2>(sr1), quad_perm<QP_X3>(sr1),
                                 sr2, quad_perm<QP_X1>(sr2), quad_perm<QP_X2>(sr2), quad_perm<QP_X3>(sr2),
                                 KX, K12X, KYP, K12Y, s_rc[8 * 64 + g], s_rc[9 * 64 + g], X, Y);
                    s_str[po & 0xffffu] = X;
                    s_str[po >> 16] = Y;
                }
                EVP_STAMP(pacc1)
                __syncthreads();
                if (rim_owner) { str[0] = s_r04[tq]; str[4] = s_r04[64 + tq]; }
            }
            sx1 = s_str[4 * SP + sp + 1];   // (sp + 1 < SP also for the last cell of the last row: SW = W + 1)
            sy2 = s_str[5 * SP + sp + 1];
        } else {
            s_str[0 * SP + sp] = str[2];
            s_str[1 * SP + sp] = str[5];
            s_str[2 * SP + sp] = str[3];
            s_str[3 * SP + sp] = str[7];
            sx1 = __shfl_down(str[1], 1);
            sy2 = __shfl_down(str[6], 1);
            __syncthreads();
        }
        EVP_STAMP(pacc2)

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
            q.sx2 = s_str[0 * SP + sp + SW]; q.sx3 = s_str[2 * SP + sp + SW + 1];
            q.sy0 = str[4]; q.sy1 = s_str[1 * SP + sp + SW];
            q.sy2 = sy2; q.sy3 = s_str[3 * SP + sp + SW + 1];
            if (tbu) MM::template stepu<CAP, true>(A.p, q, o);
            else MM::template stepu<CAP, false>(A.p, q, o);
            u_own = o.u; v_own = o.v;
            if (k == R.ndte - 1 && !R.dry) {
                R.tab[24][c] = o.strintx; R.tab[25][c] = o.strinty;
                R.tab[26][c] = o.taubx; R.tab[27][c] = o.tauby;
            }
        }
        // T-fold: a top-row cell waits for the record of the cell it is the image of -- which may sit in the SAME wave (next to the
        // pole columns) and publishes further down: every cell that is not such an image publishes first
        bool published = false;
        if (R.tfold && !isSeam) {
            if (ownU) {
                const unsigned tag = want + 1u;
                if (pub) st_rec2(wr + 2 * (size_t)c, pack_rec(u_own, tag), pack_rec(v_own, tag));
                if (img0 >= 0) { const double sg = (img0 & 1) ? -1.0 : 1.0; st_rec2(wr + 2 * (size_t)(img0 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
                if (img1 >= 0) { const double sg = (img1 & 1) ? -1.0 : 1.0; st_rec2(wr + 2 * (size_t)(img1 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
                if (img2 >= 0) { const double sg = (img2 & 1) ? -1.0 : 1.0; st_rec2(wr + 2 * (size_t)(img2 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
            }
            if (rpub) publish_remote(((k + par0) & 1) ^ 1, u_own, v_own, want + 1u);
            published = true;
        }
        if (isSeam) {
            // what the halo update does to the fold row after every subcycle: pair (a, b) <- (xavg, -xavg),
            // xavg = 0.5*(x_a + isign*x_b), isign = -1; pole points change sign.  The partner's value of
            // THIS subcycle travels as a tagged record of its own (rec_raw), like any other hand-off.
            const double isign = -1.0;
            const unsigned tag = want + 1u;
            if (R.tfold) {
                // T-fold: the top physical row is the image of row NY-1 (a(i, NY) <- -a(NX-i+1, NY-1), nothing averaged).  The cell's own
                // momentum step above has left its strintx / taubx; its velocity is the source cell's FINAL value of this subcycle,
                // which that cell publishes like any velocity another tile mirrors
                v4u ra, rb;
                unsigned spins = 0;
                if (REMOTE) t_wait0 = wall_clock64();
                for (;;) {
                    ld_rec2(wr + 2 * (size_t)seam_partner, ra, rb);
                    if (ra.x == tag && ra.w == tag && rb.x == tag && rb.w == tag) {
                        u_own = isign * unpack_rec(ra); v_own = isign * unpack_rec(rb);
                        break;
                    }
                    if (gave_up(++spins, t_wait0)) {
                        give_up_note(3, k, seam_partner, ra.x, tag);
                        s_bad = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            } else if (seam_role == 3) {
                u_own = isign * u_own; v_own = isign * v_own;
            } else {
                v4u *rw = (v4u *)R.rec_raw[((k + par0) & 1) ^ 1];
                st_rec2(rw + 2 * (size_t)c, pack_rec(u_own, tag), pack_rec(v_own, tag));
                if (REMOTE && rq0) {       // ... and into the rec_raw buffers of the ranks that hold the partner (or an image)
                    const int parw = ((k + par0) & 1) ^ 1;
                    const v4u a = pack_rec(u_own, tag), b = pack_rec(v_own, tag);
                    st_rec2_sys(rq0 + (size_t)parw * rqs0, a, b);
                    if (rq1) st_rec2_sys(rq1 + (size_t)parw * rqs1, a, b);
                    if (rq2) st_rec2_sys(rq2 + (size_t)parw * rqs2, a, b);
                }
                v4u ra, rb;
                unsigned spins = 0;
                bool ok = true;
                if (REMOTE) t_wait0 = wall_clock64();
                if (seam_remote) {
                    ok = poll_remote(rw + 2 * (size_t)seam_partner, tag, ra, rb, 3, seam_partner);
                    if (!ok) s_bad = 1;
                } else
                for (;;) {
                    ld_rec2(rw + 2 * (size_t)seam_partner, ra, rb);
                    if (ra.x == tag && ra.w == tag && rb.x == tag && rb.w == tag) break;
                    if (gave_up(++spins, t_wait0)) {
                        give_up_note(3, k, seam_partner, ra.x, tag);
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (ok) {
                    const double pu = unpack_rec(ra), pv = unpack_rec(rb);
                    if (seam_role == 1) {
                        u_own = 0.5 * (u_own + isign * pu);
                        v_own = 0.5 * (v_own + isign * pv);
                    } else {
                        u_own = isign * (0.5 * (pu + isign * u_own));
                        v_own = isign * (0.5 * (pv + isign * v_own));
                    }
                }
            }
        }
        if (isU || isSeam) { s_u[li] = u_own; s_v[li] = v_own; }   // read by the next stress phase (after the ring barrier)
        if (ownU && !published) {
            const unsigned tag = want + 1u;
            if (pub) st_rec2(wr + 2 * (size_t)c, pack_rec(u_own, tag), pack_rec(v_own, tag));
            if (img0 >= 0) { const double sg = (img0 & 1) ? -1.0 : 1.0; st_rec2(wr + 2 * (size_t)(img0 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
            if (img1 >= 0) { const double sg = (img1 & 1) ? -1.0 : 1.0; st_rec2(wr + 2 * (size_t)(img1 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
            if (img2 >= 0) { const double sg = (img2 & 1) ? -1.0 : 1.0; st_rec2(wr + 2 * (size_t)(img2 >> 1), pack_rec(sg * u_own, tag), pack_rec(sg * v_own, tag)); }
        }
        if (rpub && !published) publish_remote(((k + par0) & 1) ^ 1, u_own, v_own, want + 1u);
        // no publish step: the records carry their own tags
        EVP_STAMP(pacc3)
        if (split) __syncthreads();   // the tile's own new velocities are in LDS before anybody's next stress update
        EVP_STAMP(pacc4)
    }
#undef EVP_STAMP
    if (prof && (t & 63) == 0) {
        unsigned long long *o = R.prof + ((size_t)tile * 4 + (tq >> 6)) * 8;
        o[0] = pacc0; o[1] = pacc1; o[2] = pacc2; o[3] = pacc3; o[4] = pacc4;
        o[5] = (unsigned long long)cu_rank | ((unsigned long long)(PERM && R.cuload ? s_cu : 0) << 8) | ((unsigned long long)(s_simd[t >> 6] & 3) << 24);
        o[6] = (unsigned long long)(t >> 6); o[7] = (unsigned long long)R.nact[tile];
    }
    // ghost cells that mirror another rank's cells: fetch the final velocities (the caller
    // relies on current ghosts, ice_dyn_evp.F90:920-934)
    if (REMOTE && ring_remote) {
        v4u ra, rb;
        const v4u *rd = (const v4u *)R.rec[(R.ndte + par0) & 1];
        const bool ok = poll_remote(rd + 2 * (size_t)ring_cp, R.tag_base + (unsigned)R.ndte, ra, rb);
        if (ok && !R.dry) {
#pragma unroll
            for (int b = 0; b < 2; ++b) { R.u[b][ring_cp] = unpack_rec(ra); R.v[b][ring_cp] = unpack_rec(rb); }
        }
    }

    // ---- write the state back --------------------------------------------------------------
    if (!R.dry) {
        if (actN && own) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                R.tab[k][c] = s[k];
                R.tab[12 + k][c] = s[k];
            }
        }
        if (quad && gown) {
            R.tab[qc][gc] = sr0; R.tab[12 + qc][gc] = sr0;
            R.tab[4 + qc][gc] = sr1; R.tab[16 + qc][gc] = sr1;
            R.tab[8 + qc][gc] = sr2; R.tab[20 + qc][gc] = sr2;
        }
        if (isU || isSeam) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                double *uu = R.u[b], *vv = R.v[b];
                uu[c] = u_own; vv[c] = v_own;
                if (img0 >= 0) { const double sg = (img0 & 1) ? -1.0 : 1.0; uu[img0 >> 1] = sg * u_own; vv[img0 >> 1] = sg * v_own; }
                if (img1 >= 0) { const double sg = (img1 & 1) ? -1.0 : 1.0; uu[img1 >> 1] = sg * u_own; vv[img1 >> 1] = sg * v_own; }
                if (img2 >= 0) { const double sg = (img2 & 1) ? -1.0 : 1.0; uu[img2 >> 1] = sg * u_own; vv[img2 >> 1] = sg * v_own; }
            }
        }
    }
}

size_t lds_bytes(unsigned flags, int logw, bool coop = false)
{
    const int W = 1 << logw, H = 256 / W;
    const int nuv = ((H + 1) * (W + 1) + 7) & ~7;
    const int nu = 8 + ((flags & EVP_F_WATER_IS_OCN) ? 0 : 2) + ((flags & EVP_F_TBU_ZERO) ? 0 : 1);
    const size_t nstr = logw == 4 ? 6 * (size_t)(W + 1) * H : 4 * (size_t)256;   // PERM planes: row stride W + 1
    return sizeof(double) * (nstr + (size_t)256 * (4 + nu) + 2 * (size_t)nuv + (coop ? 12 * (size_t)64 : 0));
}

template <int LOGW, bool REMOTE>
int occ(bool strict, int cap, size_t lds)
{
    int nb = 0;
#define EVP_OCC(S, C) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, evp_resident2_tile<S, C, LOGW, REMOTE>, 64 * RTY, lds)
    hipError_t e;
    if (strict) e = cap == 3 ? EVP_OCC(true, 3) : cap == 1 ? EVP_OCC(true, 1) : cap == 0 ? EVP_OCC(true, 0) : EVP_OCC(true, -1);
    else e = cap == 3 ? EVP_OCC(false, 3) : cap == 1 ? EVP_OCC(false, 1) : cap == 0 ? EVP_OCC(false, 0) : EVP_OCC(false, -1);
#undef EVP_OCC
    return e == hipSuccess ? nb : 0;
}

template <int LOGW, bool REMOTE>
void launch(const EvpArgs &A, const EvpResident2 &R, bool strict, int cap, hipStream_t st)
{
    dim3 grid(R.nlaunch > 0 ? R.nlaunch : A.ntiles), block(64, RTY);
    const size_t lds = lds_bytes(A.flags, LOGW);
#define EVP_LAUNCH(S, C) hipLaunchKernelGGL((evp_resident2_tile<S, C, LOGW, REMOTE>), grid, block, lds, st, A, R)
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

}  // namespace

void evp_resident_geometry(int max_ni, int max_nj, int logw, int *gx, int *gy)
{
    const int W = 1 << logw, H = 256 / W;
    *gx = (max_ni + W - 2) / (W - 1);
    *gy = (max_nj + H - 2) / (H - 1);
}

// (measured slower than one thread per rim cell wherever two or three tiles share a CU -- HISTORY.md, round 5 -- so the variant is
// compiled into the test build only, as an A/B switch: CICE_EVP_HIP_RES_COOP=1, tools/coop_ab.py)
#ifdef CICE_EVP_HIP_TESTING
bool evp_resident2_coop_built(bool strict, int cap, int logw, bool remote) { return strict && cap == 3 && logw == 4 && !remote; }
#else
bool evp_resident2_coop_built(bool, int, int, bool) { return false; }
#endif

int evp_resident2_max_blocks_per_cu(bool strict, int cap, unsigned flags, int logw, bool remote, bool coop)
{
    if (coop) {
#ifdef CICE_EVP_HIP_TESTING
        if (!evp_resident2_coop_built(strict, cap, logw, remote)) return 0;
        int nb = 0;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, evp_resident2_tile<true, 3, 4, false, true>, 64 * RTY, lds_bytes(flags, 4, true)) == hipSuccess ? nb : 0;
#else
        return 0;
#endif
    }
    const size_t lds = lds_bytes(flags, logw);
    if (remote) return logw == 4 ? occ<4, true>(strict, cap, lds) : logw == 5 ? occ<5, true>(strict, cap, lds) : occ<6, true>(strict, cap, lds);
    return logw == 4 ? occ<4, false>(strict, cap, lds) : logw == 5 ? occ<5, false>(strict, cap, lds) : occ<6, false>(strict, cap, lds);
}

void evp_launch_resident2(const EvpArgs &A0, const EvpResident2 &R, int max_ni, int max_nj, int logw,
                          bool strict, int cap, hipStream_t st)
{
    EvpArgs A = A0;
    evp_resident_geometry(max_ni, max_nj, logw, &A.gx, &A.gy);
    A.ntiles = A.gx * A.gy * (R.nblocks > 0 ? R.nblocks : 1);
    const bool remote = R.rimg != nullptr;
#ifdef CICE_EVP_HIP_TESTING
    if (R.nlate && evp_resident2_coop_built(strict, cap, logw, remote)) {
        dim3 grid(R.nlaunch > 0 ? R.nlaunch : A.ntiles), block(64, RTY);
        hipLaunchKernelGGL((evp_resident2_tile<true, 3, 4, false, true>), grid, block, lds_bytes(A.flags, 4, true), st, A, R);
        return;
    }
#endif
    if (remote) {
        if (logw == 4) launch<4, true>(A, R, strict, cap, st);
        else if (logw == 5) launch<5, true>(A, R, strict, cap, st);
        else launch<6, true>(A, R, strict, cap, st);
    } else {
        if (logw == 4) launch<4, false>(A, R, strict, cap, st);
        else if (logw == 5) launch<5, false>(A, R, strict, cap, st);
        else launch<6, false>(A, R, strict, cap, st);
    }
}
