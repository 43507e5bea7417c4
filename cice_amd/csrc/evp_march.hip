// =====================================================================
// Several EVP subcycles per pass over HBM: the "marching" kernel (gfx950, wave64, fp64) for per-rank domains
// that do not fit on the chip (3600 x 2400 and the like).
//
// Why: the one-subcycle streaming kernel (evp_kernels.hip) already moves no byte twice and runs at the box's mixed
// read/write streaming rate; per subcycle it must read 27 and write 14 doubles per cell.  The only way below that is
// fewer sweeps: this kernel advances the state by K = 2, 3 or 4 subcycles of the reference's loop
// (ice_dyn_evp.F90:859-913: stress :867, stepu :889, halo :908) while touching every field once.  K = 2 was rounds
// 3-5 (0.58-0.61 of the HBM peak on its own bytes, the fp64 pipes busy half of the time: memory and arithmetic take
// turns inside the one wave a SIMD holds); K = 4 (round 6) halves the bytes per subcycle again and makes the pass
// arithmetic-bound -- the loads of a row have four subcycles of arithmetic to hide behind instead of two.
//
// Work item = ONE WAVE marching north over a strip of 64 columns x seglen rows; lane = column.  No workgroup barrier,
// no LDS-shared data: a workgroup is just four independent waves.
//   row r of the march, level L = 1 .. K (S_L = stress of subcycle k+L, U_L = stepu of subcycle k+L):
//                         S_L  on T-row r-(L-1)    (velocities U(k+L-1) of rows r-L, r-(L-1))
//                         U_L  on U-row r-L        (stress divergence from T-rows r-L, r-(L-1));  U_K -> stored
//   * neighbours in i (uvel(i-1,j), HTE(i-1,j), str(i+1,j,.)) come from the adjacent lane by wavefront shuffles (DPP),
//     neighbours in j from values the wave carries from its previous row in registers;
//   * the 12 stresses of subcycle k+L wait for S_(L+1) in a per-wave LDS stash (2 rows x 12 x 64 doubles = 12 KB per
//     level boundary: 36 KB per wave, 144 KB per workgroup at K = 4 -- one workgroup per CU, one wave per SIMD);
//   * validity shrinks by one lane per stage: S_L lanes L..64-L, U_L lanes L..63-L -- a strip owns <= 64 - 2P = 56
//     columns (P = EVP_MARCH_PAD = 4 lanes of overlap on either side, whatever K), neighbouring strips / segments
//     recompute the overlap, bit-identical by construction: same operands, same operation order (evp_cell.inc),
//     nothing depends on scheduling.
//
// Data: a device-private, STRIP-MAJOR layout (evp_host_march.cpp).  Per (row, strip) one contiguous block
// [field][64 lanes]: the state (u, v, 12 stresses: 7 KB, two copies for ping-pong), the constants of a call (dxT, dyT,
// strength, HTE, HTN and eight momentum operands: 6.5 KB), optional operands, diagnostics.  A wave's row is then a few
// long contiguous runs instead of 41 x 512 B scattered over 41 arrays -- measured with a copy of this access pattern
// (tools/march_stream.hip): 5.6 TB/s packed against 2.6-4.8 TB/s for separate arrays.  The 2P overlap lanes of a
// block duplicate columns its neighbours own; an owner stores its P edge columns into the neighbour's block too.
// With a cyclic E-W dimension inside the rank the strips wrap around (lane -> column modulo nxr): no halo columns.
// Algorithmic HBM bytes per cell and PASS: 27 reads + 14 writes = 328 B, i.e. 82 B per cell-subcycle at K = 4 (164 at
// K = 2) against the 368 B yardstick of SURVEY 8(d).
// =====================================================================
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include <cstdlib>

#include "evp_device.h"
#include "evp_math.h"

namespace {

constexpr int PADW = EVP_MARCH_PAD;

// field slots of the packed blocks (64 doubles = 512 bytes each)
enum : unsigned { S_U = 0, S_V = 1, S_SIG = 2, S_NF = 14 };
enum : unsigned { C_DXT = 0, C_DYT, C_STRENGTH, C_HTE, C_HTN, C_VRELFAC, C_UOCN, C_VOCN, C_FORCEX, C_FORCEY, C_UMASSDTI, C_FM,
                  C_UAREAR, C_NF };
enum : unsigned { O_WATERX = 0, O_WATERY, O_TBU, O_UINIT, O_VINIT, O_NF };
enum : unsigned { D_NF = 4 };
static_assert(S_NF == EVP_MARCH_S_NF && C_NF == EVP_MARCH_C_NF && O_NF == EVP_MARCH_O_NF && D_NF == EVP_MARCH_D_NF, "block sizes");

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

// What a wave needs to know about its work item; everything here is wave-uniform except lane-indexed members.
struct Item {
    int lane, wv, strip, Y0, Y1;
    int xw;                 // this lane's column (wrapped into the rectangle when E-W is cyclic inside the rank)
    bool own_x;             // the lane owns its column (stores its results)
    unsigned lane8;         // lane * 8
    unsigned dupd;          // byte offset, from the start of a state ROW, of the duplicate of this lane's column in a
                            // neighbouring block (field 0); EVP_MARCH_NODUP: none
};

__device__ __forceinline__ bool march_item(const EvpMarch &A, Item &I)
{
    I.lane = threadIdx.x & 63;
    I.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup -> four neighbouring strips of one segment.  order bit0: workgroups go to the XCDs round-robin (observed
    // dispatch rule, used for speed only); give every XCD one contiguous run of the item sequence.  bit1: segment fastest.
    int wg = blockIdx.x;
    if (A.order & 1) {
        const int per = (gridDim.x + 7) >> 3;
        wg = (wg & 7) * per + (wg >> 3);
    }
    const int item = wg * 4 + I.wv;
    if (item >= A.nitems) return false;                // whole waves; the kernels have no barrier
    if (A.items) {                                     // explicit list (wave-uniform)
        const int4 it = A.items[item];
        I.strip = __builtin_amdgcn_readfirstlane(it.x);
        I.Y0 = __builtin_amdgcn_readfirstlane(it.y);
        I.Y1 = __builtin_amdgcn_readfirstlane(it.z);
    } else {
        int seg;
        if (A.order & 2) { seg = item % A.nseg; I.strip = item / A.nseg; }
        else { I.strip = item % A.nstrips; seg = item / A.nstrips; }
        I.Y0 = seg * A.seglen;
        I.Y1 = min(I.Y0 + A.seglen, A.nyr);
    }
    const int x = I.strip * A.own - PADW + I.lane;
    I.own_x = I.lane >= PADW && I.lane < PADW + A.own && x < A.nxr;
    I.xw = x;
    if (A.wrapx) {
        if (I.xw < 0) I.xw += A.nxr;
        else if (I.xw >= A.nxr) I.xw -= A.nxr;
    }
    I.lane8 = (unsigned)I.lane * 8u;
    I.dupd = (unsigned)A.dup[I.strip * 64 + I.lane];
    return true;
}

// Buffer (MUBUF) accesses: resource descriptor of the whole buffer + wave-uniform byte offset of the row (SGPR) +
// per-lane byte offset within the row (VGPR) + the field's constant offset.  The descriptor's bounds check is what makes
// "every memory instruction unconditional" free: a lane with nothing to load or store gets an offset beyond the buffer --
// the hardware returns 0 / drops the store without touching memory (the range check looks at the per-lane offset only).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
using Rsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned OOB = 0xfffff000u;                    // + any field offset: still below 2^32, beyond every buffer
__device__ __forceinline__ Rsrc make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// field k of the block row at soff: 512 bytes per field; the instruction's immediate reaches 4095, the rest rides on soff
__device__ __forceinline__ double FLD(Rsrc r, unsigned voff, unsigned soff, unsigned k)
{
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff + (k & 7u) * 512u, soff + (k >> 3) * 4096u, 0);
    return __longlong_as_double(((long long)v.y << 32) | v.x);
}
__device__ __forceinline__ void FST(Rsrc r, unsigned voff, unsigned soff, unsigned k, double x)
{
    const long long b = __double_as_longlong(x);
    v2u v;
    v.x = (unsigned)(b & 0xffffffffll);
    v.y = (unsigned)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff + (k & 7u) * 512u, soff + (k >> 3) * 4096u, 0);
}

// LEAN: the host has verified waterx == uocn, watery == vocn, TbU == 0 on every ice U-cell and MODE == 3 (classic
// EVP, revp == 0): the optional operands are not read.
// ---------------------------------------------------------------------
// Every load in flight ONE ROW AHEAD of its use (software prefetch into registers), and every
// vector-memory instruction of the loop issued unconditionally: a lane that has nothing to load or store is given an
// out-of-range offset (dropped by the buffer bounds check) instead of being branched around.  The point is s_waitcnt: vmcnt counts in
// issue order, and across a conditional memory instruction the compiler has to assume the worst and drain the queue
// -- with predicated loads, issued where they are used, the march exposed three memory latencies per row; here the only
// wait of a row is for loads issued a whole row of arithmetic earlier (and never for the stores in between).
// What a level needs of older rows -- the geometry of T-row r-(L-1), the momentum operands and masks of U-row r-L --
// rides along in register pipelines, one stage per row.  K = 2: 248 VGPRs, two waves per SIMD allowed; K = 3, 4: one wave
// per SIMD with the 512-entry register file (VGPRs + AGPRs) to itself.
// ---------------------------------------------------------------------
template <int K, bool STRICT, int MODE, bool LEAN, bool LAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K <= 2 ? 2 : 1, K <= 2 ? 2 : 1))) void evp_marchk(EvpMarch A)
{
    static_assert(K >= 2 && K <= EVP_MARCH_KMAX, "subcycles per pass");
    using MM = Math<STRICT>;
    using SI = typename MM::SI;
    using UI = typename MM::UI;
    using UO = typename MM::UO;
    __shared__ double stash[4][K - 1][2][12][64];
    Item I;
    if (!march_item(A, I)) return;
    const int lane = I.lane, wv = I.wv, Y0 = I.Y0, Y1 = I.Y1;
    const unsigned flags = A.flags;
    const bool water_is_ocn = LEAN || (flags & EVP_F_WATER_IS_OCN);
    const bool tbu_zero = LEAN || (flags & EVP_F_TBU_ZERO);
    const bool revised = !LEAN && A.p.revp != 0.0;
    const bool own_x = I.own_x;
    const unsigned l8 = I.lane8;
    const unsigned rows = (unsigned)(A.nyr + 2 * PADW + 3);
    const unsigned srow = (unsigned)A.nstrips * (S_NF * 512), crow = (unsigned)A.nstrips * (C_NF * 512),
                   orow = (unsigned)A.nstrips * (O_NF * 512), drow = (unsigned)A.nstrips * (D_NF * 512);     // bytes per row of blocks
    const Rsrc rSin = make_rsrc(A.st_in, rows * srow), rSout = make_rsrc(A.st_out, rows * srow), rC = make_rsrc(A.cst, rows * crow),
               rO = make_rsrc(A.opt, A.opt ? rows * orow : 0u), rD = make_rsrc(A.diag, rows * drow);
    // row offsets (uniform) of the row below the first one of the march and the lane's offsets within a row of each buffer
    const unsigned row0 = (unsigned)(Y0 - K + PADW);
    unsigned sS = row0 * srow, sC = row0 * crow, sO = row0 * orow, sD = row0 * drow;
    const unsigned vS = (unsigned)I.strip * (S_NF * 512) + l8, vC = (unsigned)I.strip * (C_NF * 512) + l8,
                   vO = (unsigned)I.strip * (O_NF * 512) + l8, vD = (unsigned)I.strip * (D_NF * 512) + l8;
    unsigned em = row0 * (unsigned)A.ldx + (unsigned)(PADW + I.xw);      // mask element of (column xw, row Y0-K)

    struct Row { double u, v, hte, htn, s[12], dxT, dyT, strength; };
    auto load_row = [&](unsigned sr, unsigned cr, unsigned mm, Row &R) {          // sr, cr: row offsets
        R.u = FLD(rSin, vS, sr, S_U); R.v = FLD(rSin, vS, sr, S_V);
        R.hte = FLD(rC, vC, cr, C_HTE); R.htn = FLD(rC, vC, cr, C_HTN);
        const bool act = (mm & 1u) && lane >= 1;
        const unsigned os = act ? vS : OOB, oc = act ? vC : OOB;
#pragma unroll
        for (int k = 0; k < 12; ++k) R.s[k] = FLD(rSin, os, sr, S_SIG + k);
        R.dxT = FLD(rC, oc, cr, C_DXT); R.dyT = FLD(rC, oc, cr, C_DYT); R.strength = FLD(rC, oc, cr, C_STRENGTH);
    };
    auto load_us = [&](unsigned cr, unsigned orr, bool act, UI &w) {
        const unsigned oc = act ? vC : OOB;
        w.vrelfac = FLD(rC, oc, cr, C_VRELFAC);
        w.uocn = FLD(rC, oc, cr, C_UOCN); w.vocn = FLD(rC, oc, cr, C_VOCN);
        w.forcex = FLD(rC, oc, cr, C_FORCEX); w.forcey = FLD(rC, oc, cr, C_FORCEY);
        w.Umassdti = FLD(rC, oc, cr, C_UMASSDTI); w.fm = FLD(rC, oc, cr, C_FM); w.uarear = FLD(rC, oc, cr, C_UAREAR);
        if (!LEAN) {                                                              // (uniform conditions)
            const unsigned oo = act ? vO : OOB;
            if (!water_is_ocn) { w.waterx = FLD(rO, oo, orr, O_WATERX); w.watery = FLD(rO, oo, orr, O_WATERY); }
            if (!tbu_zero) w.TbU = FLD(rO, oo, orr, O_TBU);
            if (revised) { w.uvel_init = FLD(rO, oo, orr, O_UINIT); w.vvel_init = FLD(rO, oo, orr, O_VINIT); }
        }
    };
    auto momentum = [&](const UI &us, double uold, double vold, double sx0, double sx1, double sx2, double sx3,
                        double sy0, double sy1, double sy2, double sy3, UO &o) {
        UI w = us;
        if (water_is_ocn) { w.waterx = us.uocn; w.watery = us.vocn; }
        if (tbu_zero) w.TbU = 0.0;
        if (!revised) { w.uvel_init = 0.0; w.vvel_init = 0.0; }
        w.uold = uold; w.vold = vold;
        w.sx0 = sx0; w.sx1 = sx1; w.sx2 = sx2; w.sx3 = sx3;
        w.sy0 = sy0; w.sy1 = sy1; w.sy2 = sy2; w.sy3 = sy3;
        if (tbu_zero) MM::template stepu<MODE, false>(A.p, w, o);
        else MM::template stepu<MODE, true>(A.p, w, o);
    };

    // ---- carried state (index = level; [0] of the velocities is U(k) itself) ----
    double up[K], vp[K];                 // U(k+L) on row r-L-1: the row below the one level L+1's stress works on
    double cs[K][4];                     // str of level L+1 on T-row r-L-1: sx0, sx1 of the lane to the east, sy0, sy2 of that lane
    struct Geo { double hte, dxT, dyT, strength; };
    Geo gq[K];                           // geometry of T-row r-j (gq[0] = the current row, filled per iteration)
    double htnq[K + 1];                  // HTN of row r-j
    UI usq[K];                           // momentum operands of U-row r-1-j
    unsigned mk[K + 1];                  // masks of row r-j
#pragma unroll
    for (int j = 0; j < K; ++j) {
        up[j] = 0.0; vp[j] = 0.0;
        cs[j][0] = cs[j][1] = cs[j][2] = cs[j][3] = 0.0;
        gq[j] = Geo{0.0, 0.0, 0.0, 0.0};
        htnq[j] = 0.0;
        usq[j] = UI{};
        mk[j] = 0u;
    }
    htnq[K] = 0.0; mk[K] = 0u;
    up[0] = FLD(rSin, vS, sS, S_U); vp[0] = FLD(rSin, vS, sS, S_V);        // U(k), row Y0-K
    htnq[0] = FLD(rC, vC, sC, C_HTN);                                       // (shifted into htnq[1] by the first iteration)
    // in flight when the loop starts: the first row of the march, momentum operands nobody uses, the masks of that row and the next
    unsigned m_n = A.mask[em + (unsigned)A.ldx], m_nn = A.mask[em + 2u * (unsigned)A.ldx];
    Row N{};
    UI usN{};
    load_us(sC, sO, false, usN);
    load_row(sS + srow, sC + crow, m_n, N);

    for (int r = Y0 - (K - 1); r <= Y1 - 1 + K; ++r) {
        sS += srow; sC += crow; sO += orow; sD += drow; em += (unsigned)A.ldx;          // blocks of row r
        // ---- the pipelines move up one row ----
#pragma unroll
        for (int j = K; j >= 1; --j) { htnq[j] = htnq[j - 1]; mk[j] = mk[j - 1]; }
#pragma unroll
        for (int j = K - 1; j >= 1; --j) { gq[j] = gq[j - 1]; usq[j] = usq[j - 1]; }
        const Row C = N;
        usq[0] = usN;                                               // momentum operands of U-row r-1
        mk[0] = m_n;
        htnq[0] = C.htn;
        gq[0] = Geo{C.hte, C.dxT, C.dyT, C.strength};
        m_n = m_nn;
        // everything the NEXT row consumes, requested now
        {
            const bool isU1n = (mk[0] & 2u) && lane >= 1 && lane <= 62 && r + 1 >= Y0 - K + 2;
            load_us(sC, sO, isU1n, usN);                            // U-row r
            m_nn = A.mask[em + 2u * (unsigned)A.ldx];               // (spare rows on top of the arrays)
            load_row(sS + srow, sC + crow, m_n, N);                 // row r+1
        }

        double cu = C.u, cv = C.v;                                  // U(k+L-1) on T-row r-(L-1), the level's "row above"
#pragma unroll
        for (int L = 1; L <= K; ++L) {
            const int tr = r - (L - 1), ur = r - L;
            // ---- S_L: stress(k+L) on T(x, tr) ----
            const bool act = (mk[L - 1] & 1u) && lane >= L && lane <= 64 - L && tr >= Y0 - (K - L);
            double str[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) str[k] = 0.0;
            double s[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) s[k] = 0.0;
            {
                const double uL = lane_up(cu), vL = lane_up(cv), hteL = lane_up(gq[L - 1].hte);
                const double uL_p = lane_up(up[L - 1]), vL_p = lane_up(vp[L - 1]);
                if (act) {
                    if (L == 1) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) s[k] = C.s[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 12; ++k) s[k] = stash[wv][L - 2][tr & 1][k][lane];
                    }
                    SI a;
                    a.dxT = gq[L - 1].dxT; a.dyT = gq[L - 1].dyT; a.strength = gq[L - 1].strength;
                    a.u_ij = cu; a.u_im = uL; a.u_jm = up[L - 1]; a.u_mm = uL_p;
                    a.v_ij = cv; a.v_im = vL; a.v_jm = vp[L - 1]; a.v_mm = vL_p;
                    MM::metrics(gq[L - 1].hte, hteL, htnq[L - 1], htnq[L], A.deltaminEVP, a);
                    MM::template stress<MODE>(A.p, a, s, str);
                    if (L < K) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) stash[wv][L < K ? L - 1 : 0][tr & 1][k][lane] = s[k];
                    }
                }
            }
            if (L == K) {
                // new stresses of T(x, tr): into the owner's block, and -- the P columns on either edge of a strip -- into the
                // neighbouring block that duplicates them (a second store instruction with few live lanes)
                const bool st = act && own_x && tr < Y1;
                const unsigned o1 = st ? vS : OOB;
                const unsigned o2 = (st && I.dupd != EVP_MARCH_NODUP) ? I.dupd : OOB;
                const unsigned so = sS - (unsigned)(K - 1) * srow;
#pragma unroll
                for (int k = 0; k < 12; ++k) FST(rSout, o1, so, S_SIG + k, s[k]);
#pragma unroll
                for (int k = 0; k < 12; ++k) FST(rSout, o2, so, S_SIG + k, s[k]);
            }
            const double n_sx3 = lane_dn(str[3]), n_sy3 = lane_dn(str[7]);
            const double t_sx1 = lane_dn(str[1]), t_sy2 = lane_dn(str[6]);

            // ---- U_L: stepu(k+L) on U(x, ur) ----
            if (L < K) {
                double nu = up[L - 1], nv = vp[L - 1];
                const bool isU = (mk[L] & 2u) && lane >= L && lane <= 63 - L && ur >= Y0 - (K - L);
                if (isU) {
                    UO o;
                    momentum(usq[L - 1], up[L - 1], vp[L - 1], cs[L - 1][0], cs[L - 1][1], str[2], n_sx3, cs[L - 1][2], str[5], cs[L - 1][3], n_sy3, o);
                    nu = o.u; nv = o.v;
                }
                // hand the level's rows over: what was "above" becomes "below" for the next row
                up[L - 1] = cu; vp[L - 1] = cv;
                cu = nu; cv = nv;
            } else {
                const bool isU = (mk[L] & 2u) && own_x && ur >= Y0;
                UO o;
                o.u = 0.0; o.v = 0.0; o.strintx = 0.0; o.strinty = 0.0; o.taubx = 0.0; o.tauby = 0.0;
                if (isU)
                    momentum(usq[L - 1], up[L - 1], vp[L - 1], cs[L - 1][0], cs[L - 1][1], str[2], n_sx3, cs[L - 1][2], str[5], cs[L - 1][3], n_sy3, o);
                const unsigned o1 = isU ? vS : OOB;
                const unsigned od = (isU && I.dupd != EVP_MARCH_NODUP) ? I.dupd : OOB;
                const unsigned so = sS - (unsigned)K * srow;
                FST(rSout, o1, so, S_U, o.u); FST(rSout, o1, so, S_V, o.v);
                FST(rSout, od, so, S_U, o.u); FST(rSout, od, so, S_V, o.v);
                if (LAST) {
                    const unsigned q = isU ? vD : OOB;
                    const unsigned sd = sD - (unsigned)K * drow;
                    FST(rD, q, sd, 0, o.strintx); FST(rD, q, sd, 1, o.strinty);
                    FST(rD, q, sd, 2, o.taubx); FST(rD, q, sd, 3, o.tauby);
                }
                up[L - 1] = cu; vp[L - 1] = cv;
            }
            cs[L - 1][0] = str[0]; cs[L - 1][1] = t_sx1; cs[L - 1][2] = str[4]; cs[L - 1][3] = t_sy2;
        }
    }
}

// ---------------------------------------------------------------------
// Measured and removed (round 6, profiles/r06_march_k_sweep.txt): the same pass as a pipeline ACROSS the four waves of a workgroup
// (wave w = level w + 1 of one work item, rows handed on through 61 KB of LDS with one workgroup barrier per row, two workgroups per
// CU).  Bit-identical, 192 VGPRs, no register pipelines -- and 348 us per subcycle against 284 for the kernel above on the same box:
// 760-820 instructions per level and row against 723, and the lock step of four waves costs more than two waves per SIMD win.
// ---------------------------------------------------------------------
// ---------------------------------------------------------------------
// CICE block layout <-> strip-major layout (evp_host_march.cpp owns the geometry)
// ---------------------------------------------------------------------
// cell (x, y) of the rectangle incl. the P halo layers -> (block index of its row/strip, lane) of the strip that OWNS the
// column (or, for halo columns of a closed side, the edge strip that holds it)
__device__ __forceinline__ void cell_to_packed(const EvpMarchGeo &G, int x, int y, long &blk, int &lane)
{
    const int s = min(max(x, 0) / G.own, G.nstrips - 1);
    lane = x - s * G.own + PADW;
    blk = (long)(y + EVP_MARCH_PAD) * G.nstrips + s;
}

struct CellMap {
    long blk; int lane;   // owner position in the packed buffers
    int em;               // element of the row-major byte mask
    bool ok;              // the position exists
    bool live;            // it is a real cell of the rank's rectangle (interior, or the cell a wrap ghost images)
};

// block cell (b; i, j 1-based) -> packed position
__device__ __forceinline__ CellMap block_to_packed(const EvpMarchGeo &G, int b, int i, int j)
{
    const int2 o = G.blk_org[b];                 // rectangle coordinates of the block's first interior cell
    int xs = o.x + (i - G.ilo);
    const int ys = o.y + (j - G.ilo);
    CellMap c{};
    if (xs < -EVP_MARCH_PAD || xs >= G.nxr + EVP_MARCH_PAD || ys < -EVP_MARCH_PAD || ys >= G.nyr + EVP_MARCH_PAD) return c;
    if (G.wrapx) {                               // a ghost cell across the cyclic seam takes the cell it images
        if (xs < 0) xs += G.nxr;
        else if (xs >= G.nxr) xs -= G.nxr;
    }
    cell_to_packed(G, xs, ys, c.blk, c.lane);
    c.em = (ys + EVP_MARCH_PAD) * G.ldx + EVP_MARCH_PAD + xs;
    c.ok = c.lane >= 0 && c.lane < 64;
    // live: the position holds a real cell of the GLOBAL domain -- one of this rank's, or (two-cell ring kept current by
    // the exchange) one of another rank's
    const int gx = G.gx0 + xs, gy = G.gy0 + ys;
    c.live = c.ok && gy >= 0 && gy < G.nyg && ((gx >= 0 && gx < G.nxg) || G.ew_cyclic);
    return c;
}

// column x, row y of the rectangle (incl. halo) -> element of the block-layout arrays it takes its value from; -1: none
// (0).  inside: interior cell of a block; beyond a cyclic seam: the interior cell it images; first halo layer of a
// closed side: the ghost cell of the edge block (the caller's value there is what the reference reads,
// ice_dyn_evp.F90:867)
__device__ __forceinline__ int rect_to_block(const EvpMarchGeo &G, int x, int y, bool &is_cell)
{
    int xs = x - G.ext_w, ys = y - G.ext_s;          // relative to the rank's own cells
    if (G.wrapx) { if (xs < 0) xs += G.nxo; else if (xs >= G.nxo) xs -= G.nxo; }
    is_cell = xs >= 0 && xs < G.nxo && ys >= 0 && ys < G.nyo;
    if (xs < -1 || xs > G.nxo || ys < -1 || ys > G.nyo) return -1;
    const int bi = min(max(xs, 0) / G.bsx, G.nbx - 1), bj = min(max(ys, 0) / G.bsy, G.nby - 1);
    const int b = G.blkid[bj * G.nbx + bi];
    if (b < 0) return -1;
    const int i = G.ilo + (xs - bi * G.bsx), j = G.ilo + (ys - bj * G.bsy);      // 1-based; ilo-1 / ihi+1 on closed sides
    if (i < 1 || i > G.nxb || j < 1 || j > G.nyb) return -1;
    return b * G.plane + (j - 1) * G.nxb + (i - 1);
}

// every lane of every block of every row (all duplicates included) <- the block-layout arrays
__global__ __launch_bounds__(256) void march_gather(EvpMarchGeo G, EvpMarchTab T, const uint8_t *__restrict__ mask_blk,
                                                    uint8_t *__restrict__ mask_rect)
{
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6), row = blockIdx.y;
    if (s >= G.nstrips) return;
    const int x = s * G.own - PADW + lane, y = row - EVP_MARCH_PAD;
    bool is_cell;
    const int src = rect_to_block(G, x, y, is_cell);
    const size_t blk = (size_t)row * G.nstrips + s;
    for (int f = 0; f < T.n; ++f) {
        const double v = src >= 0 ? T.blk[f][src] : 0.0;
        const size_t e = (blk * T.nf[f] + T.slot[f]) * 64 + lane;
        T.pk[f][e] = v;
        if (T.pk2[f]) T.pk2[f][e] = v;
    }
    // the byte mask stays row-major; one lane per column writes it (the owner, or the edge strips for the halo columns)
    if (mask_rect) {
        const bool mine = (lane >= PADW && lane < PADW + G.own) || (s == 0 && lane < PADW) || (s == G.nstrips - 1 && lane >= PADW + G.own);
        if (mine && x + EVP_MARCH_PAD < G.ldx && x >= -EVP_MARCH_PAD)
            mask_rect[(size_t)row * G.ldx + EVP_MARCH_PAD + x] = (src >= 0 && is_cell) ? (mask_blk[src] & 3u) : 0;
    }
}

// Is the caller's block-layout state the image of ONE global state?  The reference computes the T-cells of the
// north / east fringe (ihi+1, jhi+1) redundantly on every block from that block's own ghost storage
// (ice_dyn_shared.F90:740-749) and reads ghost velocities as the caller left them; the packed layout holds every cell
// once (plus exact duplicates).  Both give the same bits iff the ghost values equal the cells they image, which CICE
// maintains (halo updates of iceTmask, strength and the velocities before the loop; stresses evolve alike on both
// copies).  A caller for which this does not hold (synthetic tests with independent holes in ghost masks) gets the
// one-subcycle kernels, which keep per-block ghost storage.
__global__ __launch_bounds__(256) void march_check(EvpMarchGeo G, EvpMarchTab T, const uint8_t *__restrict__ mask_blk,
                                                   const uint8_t *__restrict__ mask_rect, int nuv, int nfringe,
                                                   unsigned *__restrict__ bad)
{
    // T: [0, nuv) compared on the whole ghost ring, [nuv, nuv + nfringe) on the fringe T-cells, the rest (HTE, HTN) on
    // the fringe and on column ilo-1 / row jlo-1
    const int i = blockIdx.x * 256 + threadIdx.x + 1, j = blockIdx.y + 1, b = blockIdx.z;
    const int4 r = G.blk[b];
    if (i < r.x - 1 || i > r.y + 1 || j < r.z - 1 || j > r.w + 1) return;
    const bool ghost = i < r.x || i > r.y || j < r.z || j > r.w;
    if (!ghost) return;
    const bool fringe = (i == r.y + 1 || j == r.w + 1) && i >= r.x && j >= r.z;
    const int s = b * G.plane + (j - 1) * G.nxb + (i - 1);
    const CellMap c = block_to_packed(G, b, i, j);
    unsigned nbad = 0;
    auto differs = [](double p, double q) { return __double_as_longlong(p) != __double_as_longlong(q); };
    auto pk = [&](int f) { return T.pk[f][((size_t)c.blk * T.nf[f] + T.slot[f]) * 64 + c.lane]; };
    // velocities: every ghost cell the stress of an owned T-cell reads (closed sides too: the packed layout took the
    // value from ONE of the ghost cells that image the position; they must all agree)
    if (c.ok)
        for (int f = 0; f < nuv; ++f) nbad += differs(T.blk[f][s], pk(f));
    if (c.live) {
        if (fringe) {
            for (int f = nuv; f < T.n; ++f) nbad += differs(T.blk[f][s], pk(f));
            if (mask_blk) nbad += ((mask_blk[s] ^ mask_rect[c.em]) & 1u);
        } else if (i == r.x - 1 || j == r.z - 1) {
            for (int f = nuv + nfringe; f < T.n; ++f) nbad += differs(T.blk[f][s], pk(f));
        }
    } else if (fringe && mask_blk) {
        nbad += (mask_blk[s] & 1u);          // a T-cell beyond a closed boundary that the reference would compute
    }
    if (nbad) atomicAdd(bad, nbad);
}

// packed -> block layout after the loop.  Velocities: every cell with a live source (interior, ghost images of cells
// of this rank: what the reference's halo update leaves, ice_dyn_evp.F90:908-910); stresses: the T-cells the reference
// updates (ilo..ihi+1 x jlo..jhi+1 where iceTmask); strintx/y, taubx/y: interior ice U-cells.
__global__ __launch_bounds__(256) void march_scatter(EvpMarchGeo G, EvpMarchTab T, const uint8_t *__restrict__ mask_blk,
                                                     int nuv, int nsig)
{
    const int i = blockIdx.x * 256 + threadIdx.x + 1, j = blockIdx.y + 1, b = blockIdx.z;
    const int4 r = G.blk[b];
    if (i < r.x - 1 || i > r.y + 1 || j < r.z - 1 || j > r.w + 1) return;
    const CellMap c = block_to_packed(G, b, i, j);
    if (!c.live) return;
    const int s = b * G.plane + (j - 1) * G.nxb + (i - 1);
    const unsigned m = mask_blk[s];
    auto pk = [&](int f) { return T.pk[f][((size_t)c.blk * T.nf[f] + T.slot[f]) * 64 + c.lane]; };
    for (int f = 0; f < nuv; ++f) T.blk[f][s] = pk(f);
    if ((m & 1u) && i >= r.x && j >= r.z)
        for (int f = nuv; f < nuv + nsig; ++f) T.blk[f][s] = pk(f);
    if ((m & 2u) && i >= r.x && i <= r.y && j >= r.z && j <= r.w)
        for (int f = nuv + nsig; f < T.n; ++f) T.blk[f][s] = pk(f);
}

// ---- the two-cell ring between ranks (march_plan.h): list-driven pack / unpack around ncclSend / ncclRecv ----
// element t of the wire -> (peer q, field f, list entry e); t - C.start[q] * nf is the element's place in the peer's block
struct RingElem { int q, f, e, local; };
__device__ __forceinline__ RingElem ring_elem(const EvpRingCuts &C, int nf, int t)
{
    int q = 0;
    while (q + 1 < C.n && t >= C.start[q + 1] * nf) ++q;
    RingElem r;
    r.q = q;
    r.local = t - C.start[q] * nf;
    const int nq = C.start[q + 1] - C.start[q];
    r.f = r.local / nq;
    r.e = C.start[q] + (r.local - r.f * nq);
    return r;
}
__global__ __launch_bounds__(256) void march_pack(const double *__restrict__ buf, int nf, const int *__restrict__ pos, int n, EvpRingCuts C,
                                                  double *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * nf) return;
    const RingElem r = ring_elem(C, nf, t);
    const int p = pos[r.e];
    out[t] = buf[((size_t)(p >> 6) * nf + r.f) * 64 + (p & 63)];
}
__device__ __forceinline__ void ring_store(double *__restrict__ buf, double *__restrict__ buf2, int nf, const int *__restrict__ pos1,
                                           const int *__restrict__ pos2, const RingElem &r, double v)
{
    const int p = pos1[r.e], q = pos2[r.e];
    const size_t e1 = ((size_t)(p >> 6) * nf + r.f) * 64 + (p & 63);
    buf[e1] = v;
    if (buf2) buf2[e1] = v;
    if (q >= 0) {                                    // the column's duplicate in the neighbouring strip's block
        const size_t e2 = ((size_t)(q >> 6) * nf + r.f) * 64 + (q & 63);
        buf[e2] = v;
        if (buf2) buf2[e2] = v;
    }
}
__global__ __launch_bounds__(256) void march_unpack(double *__restrict__ buf, double *__restrict__ buf2, int nf,
                                                    const int *__restrict__ pos1, const int *__restrict__ pos2, int n, EvpRingCuts C,
                                                    const double *__restrict__ in)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * nf) return;
    ring_store(buf, buf2, nf, pos1, pos2, ring_elem(C, nf, t), in[t]);
}
// ---- ... and without a library: stores into the peers' inboxes (evp_device.h: EvpMarchDirect) ----
__global__ __launch_bounds__(256) void march_pack_direct(const double *__restrict__ buf, int nf, const int *__restrict__ pos, int n,
                                                         EvpRingCuts C, EvpMarchDirect D, unsigned seq)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n * nf) {
        const RingElem r = ring_elem(C, nf, t);
        const int p = pos[r.e];
        const double v = buf[((size_t)(p >> 6) * nf + r.f) * 64 + (p & 63)];
        double *dst = D.dst[r.q] + (size_t)(seq & 1u) * D.dst_pstride[r.q] + r.local;
        __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // The stores are write-through at system scope: once the memory system has acknowledged them (vmcnt 0) they are where a
    // peer's loads look, whatever this GPU's caches hold -- no release FENCE here: at system scope that writes the whole L2
    // back, once per workgroup (measured: 120 us per exchange instead of 10).  The counter lives in the same uncached
    // memory; the workgroup that brings it to the launch's size knows every other one's stores have been acknowledged.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(D.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(D.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int q = 0; q < D.npeers; ++q) __hip_atomic_store(D.peer_flag[q], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// The wait for the peers' flags is a launch of its own, ONE wave: the unpack kernel behind it on the stream then never spins.
// (As one kernel -- every workgroup of the unpack polling before it read its share -- a large ring filled the whole GPU with
// spinning workgroups: harmless on a GPU of the rank's own, a deadlock until the time-out when two ranks rehearse on ONE GPU
// and the peer's pack kernel finds no CU to run on.  Round 6: 3600 x 2400 as two processes, ring of eight cells.)
__global__ __launch_bounds__(64) void march_wait_direct(EvpMarchDirect D, unsigned seq)
{
    if ((int)threadIdx.x < D.npeers && __hip_atomic_load(D.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        const unsigned *f = D.flags_in + (size_t)threadIdx.x * 16;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            const unsigned have = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(have - seq) >= 0) break;
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0 &&
                (wall_clock64() - t0 > D.timeout_ticks || __hip_atomic_load(D.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                atomicCAS(D.err, 0, 1 + (int)threadIdx.x);       // which peer never arrived
                break;
            }
        }
    }
}
__global__ __launch_bounds__(256) void march_unpack_direct(double *__restrict__ buf, double *__restrict__ buf2, int nf,
                                                           const int *__restrict__ pos1, const int *__restrict__ pos2, int n, EvpRingCuts C,
                                                           EvpMarchDirect D, unsigned seq, const double *__restrict__ verify,
                                                           unsigned *__restrict__ bad)
{
    // (no acquire fence -- it would invalidate the L2 once per workgroup: the inbox is read past the caches, after the flags)
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * nf) return;
    const double v = __hip_atomic_load(D.inbox + (size_t)(seq & 1u) * D.inbox_pstride + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (verify) {
        if (__double_as_longlong(v) != __double_as_longlong(verify[t])) atomicAdd(bad, 1u);
        return;
    }
    ring_store(buf, buf2, nf, pos1, pos2, ring_elem(C, nf, t), v);
}
__global__ __launch_bounds__(256) void march_pack_mask(const uint8_t *__restrict__ mask, const int *__restrict__ idx, int n,
                                                       double *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) out[t] = (double)mask[idx[t]];
}
__global__ __launch_bounds__(256) void march_unpack_mask(uint8_t *__restrict__ mask, const int *__restrict__ idx, int n,
                                                         const double *__restrict__ in)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) mask[idx[t]] = (uint8_t)in[t];
}

}  // namespace

template <int K>
static void launch_march_k(const EvpMarch &A, bool strict, int mode, bool lean, dim3 grid, dim3 block, hipStream_t st)
{
#define EVP_MARCH_LAUNCH(S, M, L)                                                                    \
    do {                                                                                             \
        if (A.last) hipLaunchKernelGGL((evp_marchk<K, S, M, L, true>), grid, block, 0, st, A);       \
        else hipLaunchKernelGGL((evp_marchk<K, S, M, L, false>), grid, block, 0, st, A);             \
    } while (0)
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

void evp_launch_march(const EvpMarch &A, bool strict, int mode, hipStream_t st)
{
    const unsigned nwg = (unsigned)((A.nitems + 3) / 4);
    const dim3 grid((A.order & 1) ? ((nwg + 7) / 8) * 8 : nwg), block(256);
    const bool lean = mode == 3 && (A.flags & EVP_F_WATER_IS_OCN) && (A.flags & EVP_F_TBU_ZERO) &&
                      !(evp_env_test("CICE_EVP_HIP_MARCH_LEAN") && !std::atoi(evp_env_test("CICE_EVP_HIP_MARCH_LEAN")));
    switch (A.kpass) {
    case 2: launch_march_k<2>(A, strict, mode, lean, grid, block, st); break;
#if EVP_MARCH_KMAX >= 4
    case 3: launch_march_k<3>(A, strict, mode, lean, grid, block, st); break;
    default: launch_march_k<4>(A, strict, mode, lean, grid, block, st); break;
#else
    default: launch_march_k<3>(A, strict, mode, lean, grid, block, st); break;
#endif
    }
}

void evp_launch_march_gather(const EvpMarchGeo &G, const EvpMarchTab &T, const uint8_t *mask_blk, uint8_t *mask_rect,
                             hipStream_t st)
{
    hipLaunchKernelGGL(march_gather, dim3((unsigned)((G.nstrips + 3) / 4), (unsigned)G.rows), dim3(256), 0, st, G, T,
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

void evp_launch_march_pack(const double *buf, int nf, const int *pos, int n, const EvpRingCuts &C, double *out, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(march_pack, dim3((unsigned)(((size_t)n * nf + 255) / 256)), dim3(256), 0, st, buf, nf, pos, n, C, out);
}
void evp_launch_march_unpack(double *buf, double *buf2, int nf, const int *pos1, const int *pos2, int n, const EvpRingCuts &C, const double *in,
                             hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(march_unpack, dim3((unsigned)(((size_t)n * nf + 255) / 256)), dim3(256), 0, st, buf, buf2, nf, pos1, pos2, n, C, in);
}
void evp_launch_march_pack_direct(const double *buf, int nf, const int *pos, int n, const EvpRingCuts &C, const EvpMarchDirect &D, unsigned seq,
                                  hipStream_t st)
{
    const unsigned nwg = (unsigned)std::max<size_t>(1, ((size_t)n * nf + 255) / 256);      // (at least one: the flags must rise)
    hipLaunchKernelGGL(march_pack_direct, dim3(nwg), dim3(256), 0, st, buf, nf, pos, n, C, D, seq);
}
void evp_launch_march_unpack_direct(double *buf, double *buf2, int nf, const int *pos1, const int *pos2, int n, const EvpRingCuts &C,
                                    const EvpMarchDirect &D, unsigned seq, const double *verify, unsigned *bad, hipStream_t st)
{
    const unsigned nwg = (unsigned)std::max<size_t>(1, ((size_t)n * nf + 255) / 256);
    hipLaunchKernelGGL(march_wait_direct, dim3(1), dim3(64), 0, st, D, seq);
    hipLaunchKernelGGL(march_unpack_direct, dim3(nwg), dim3(256), 0, st, buf, buf2, nf, pos1, pos2, n, C, D, seq, verify, bad);
}
void evp_launch_march_pack_mask(const uint8_t *mask, const int *idx, int n, double *out, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(march_pack_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mask, idx, n, out);
}
void evp_launch_march_unpack_mask(uint8_t *mask, const int *idx, int n, const double *in, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(march_unpack_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mask, idx, n, in);
}
