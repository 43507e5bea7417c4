// =====================================================================
// C-grid EVP subcycle, on-chip resident (round 5): ALL subcycles of a call in ONE launch for domains that fit the chip
// (gx1, tx1-sized: up to ~130k cells on one rank).  Reference: evp()'s loop for grid_ice = 'C', ice_dyn_evp.F90:936-1121
// (strain_rates_U ice_dyn_shared.F90:2341-2444, stressC_T :1758, stressC_U :1898, div_stress_Ex / _Ny :2195-2416,
// stepu_C / stepv_C ice_dyn_shared.F90:1090, 1189, the grid_average_X2Y variants ice_grid.F90:4388-4606).
//
// cg_one (evp_cgrid.hip) is one launch per subcycle: every level of every subcycle fetches its operands from L2 /
// Infinity Cache again and a subcycle is four dependent memory round trips (18 us on gx1, 0.24 of the HBM roof although
// nothing needs to move).  Here the B grid's answer (evp_resident2.hip): the state and the operands of a window stay in
// registers and LDS for the whole call, and the only thing that travels between workgroups is the new face velocity of the
// cells another window's rim mirrors -- one aligned 32-byte record {tag, uvelE, tag | tag, vvelN, tag} per cell and subcycle,
// written with write-through stores and polled (L1-bypassing) by the reader until both tags carry the subcycle it waits for.
//
// Window = 16 x 16 positions, one thread each, the inner 13 x 13 owned (cg_one's window, same table of source cells:
// halo_plan.cpp build_window_table with one extra row / column).  Levels as in cg_one, separated by workgroup barriers:
//   S  strain_rates_U (shearU alone except for deltaU in the last subcycle) at every position
//   T  stressC_T at tx, ty >= 1          U  corner viscosity + stressC_U at tx, ty <= 14
//   C  div_stress + stepu_C / stepv_C on the owned cells; publish.
// Rim positions recompute what the neighbouring window computes AND keep their own copy of its history (stresspT, stressmT,
// stress12U in registers): same arithmetic on the same inputs => same bits, so the velocities are the only hand-off.
// A position's operands are those of the cell its value comes from (the table); a neighbour's are the window neighbour's --
// the host has verified that every ghost cell's static arrays equal its source's bit for bit (cgres_geometry_images_ok),
// else this kernel is not used.
//
// Lock step: every position outside the owned range that has a producer is polled every subcycle, every cell some other
// window mirrors publishes every subcycle, ice or not (geometry only, never the masks): a window cannot run more than one
// subcycle ahead of a window that still reads its records, so two record buffers (subcycle parity) suffice.  Every spin is
// bounded and raises the error word.  fp64, strict: no FMA contraction, the reference's operation order.
//
// Windows without ice do not run (per call: cg_res_live, R.order / R.live): a window none of whose positions -- rim, mirrored rows and
// the ghost T cells it serves included -- carries ice in any of the four masks changes nothing in a call; its cells keep the values
// every reader loaded at the start, so a reader does not poll a cell of such a window.  What is launched, and has to be co-resident, is
// the list of windows with ice: a grid ten times the chip's size runs here when a tenth of its windows hold ice.
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "evp_device.h"

#pragma clang fp contract(off)

// per-phase cycle stamps exist in the TEST build's object of this file only (tools/cgres_phases.py)
#ifdef CICE_EVP_HIP_TESTING
#define CGRES_PROF(R) ((R).prof != nullptr)
#define CGRES_DBG(R) ((R).dbg)
#else
#define CGRES_PROF(R) false
#define CGRES_DBG(R) 0
#endif

namespace {

constexpr int X = 16, Y = 16, LW = X + 1, NP = LW * (Y + 1);

typedef unsigned v4u __attribute__((ext_vector_type(4)));

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

// visc_replpress, ice_dyn_shared.F90:2446-2475 (the one-division form for capping = 1: see evp_cgrid.hip)
__device__ __forceinline__ void visc_replpress(const EvpScalars &p, double strength, double DminArea, double Delta,
                                               double &zetax2, double &etax2, double &rep_prs)
{
    double tmpcalc;
    if (p.capping == 1.0 && DminArea > 0.0 && fabs(strength) <= 1.7976931348623157e308 && Delta <= 1.7976931348623157e308)
        tmpcalc = strength / fmax(Delta, DminArea);
    else
        tmpcalc = p.capping * (strength / fmax(Delta, DminArea)) + (1.0 - p.capping) * (strength / (Delta + DminArea));
    zetax2 = (1.0 + p.Ktens) * tmpcalc;
    rep_prs = (1.0 - p.Ktens) * tmpcalc * Delta;
    etax2 = p.epp2i * zetax2;
}

// grid_average_X2YA with two / four weights (ice_grid.F90:4388-4606), operands already in registers
__device__ __forceinline__ double avg2(double a0, double w0, double a1, double w1)
{
    const double wtmp = (w0 + w1);
    if (wtmp == 0.0) return 0.0;
    return (a0 * w0 + a1 * w1) / wtmp;
}
__device__ __forceinline__ double avg4(double a0, double w0, double a1, double w1, double a2, double w2, double a3, double w3)
{
    const double wtmp = (w0 + w1 + w2 + w3);
    if (wtmp == 0.0) return 0.0;
    return (a0 * w0 + a1 * w1 + a2 * w2 + a3 * w3) / wtmp;
}

struct TOut { double zetax2, etax2, sp, sm, shearT; };
// stressC_T (ice_dyn_evp.F90:1758-1860); the four corner values of shearU handed in
__device__ __forceinline__ TOut t_stress(const EvpScalars &p, double uEo, double uEw, double vNo, double vNs, double dyEo, double dyEw,
                                         double dxNo, double dxNs, double dxT2, double dyT2, double uao, double uas, double uasw,
                                         double uaw, double uareaavgr, double strength, double DminT, double shO, double shS,
                                         double shSW, double shW, double spo, double smo, double relax)
{
    const double divT = dyEo * uEo - dyEw * uEw + dxNo * vNo - dxNs * vNs;
    const double tensionT = (dyT2) * (uEo / dyEo - uEw / dyEw) - (dxT2) * (vNo / dxNo - vNs / dxNs);
    const double shearTsqr = (shO * shO * uao + shS * shS * uas + shSW * shSW * uasw + shW * shW * uaw) * uareaavgr;
    TOut r;
    r.shearT = (shO * uao + shS * uas + shSW * uasw + shW * uaw) * uareaavgr;
    const double DeltaT = sqrt(divT * divT + p.e_factor * (tensionT * tensionT + shearTsqr));
    double rep_prs;
    visc_replpress(p, strength, DminT, DeltaT, r.zetax2, r.etax2, rep_prs);
    r.sp = (spo * relax + p.arlx1i * (r.zetax2 * divT - rep_prs)) * p.denom1;
    r.sm = (smo * relax + p.arlx1i * r.etax2 * tensionT) * p.denom1;
    return r;
}

__device__ __forceinline__ void push1(const EvpCgrid &A, size_t c, double *f, double v)
{
    const int s = A.img_slot[c];
    if (s < 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int d = A.img_dst[3 * s + k];
        if (d >= 0) f[d] = v;
    }
}

// AVGS: visc_method = 'avg_strength' -- the corner viscosities come from the T -> U average of the strength (A.strengthU, once
// per call) and the corner's own Delta (ice_dyn_evp.F90:992-996), so level S works out the whole of strain_rates_U in every
// subcycle and hands deltaU to level U through the plane etax2T travels in otherwise.
// REVP: revised EVP (revp = 1) -- revp * uvelE_init / revp * vvelN_init are operands of the momentum step (two more planes by
// owned cell); under classic EVP they are zeros of the initial velocity's sign and travel as two bits.
// The planes read inside the window only are (X+1) wide like the others wherever LDS allows it (53.3 KB per workgroup for three
// per CU): X wide they measured 3-4 % slower on one box (levels U and C: gx1 6.5 -> 6.8 us per subcycle, gx3 4.17 -> 4.30), so
// only the variant that needs the two extra planes AND the T -> U weights (revised EVP with avg_zeta) packs them.
//
// FOLD: a tripole (u-fold) grid, ice_boundary.F90:1626-1722.  The windows at the fold (tl.w bit 0) own the rows up to the fold row NY at
// tile row tf and hold, in the three tile rows above it, a MIRRORED mini-tile in SOURCE orientation (global rows NY-2, NY-1, NY; the table:
// halo_plan.cpp build_fold_window_table): recomputing a ghost cell in ITS orientation would sum in another order, recomputing the cell it
// mirrors in that cell's own orientation does not.  The two rows that face each other across the fold (tile rows tf and tf+3, "cls" 1 / 3)
// read "north" through a remap: E-face / corner type fields at column 15 - tx of the other row, centre / N-face type at 16 - tx (N faces
// and corners one row further), vectors with the sign changed.  Everything ON the fold (N faces and NE corners of row NY: vvelN, uvelN,
// uvelU, vvelU, shearU, stress12U) is what the reference's halo update makes of it in every subcycle, ice or not -- the average of
// the two raw values, x_lo and x_hi of the pair of columns: s * 0.5 * (x_lo + isign * x_hi), s = 1 for the lower column, isign for the higher
// (evp_cgrid.hip: cg_fold_gather) -- built from both sides' raw values, which every window at the fold computes itself.
// SLOW (round 6): the call's operands do not satisfy the default-configuration short cuts -- seabed stress (TbE / TbN != 0), waterx /
// watery other than the ocean currents (an ocean turning angle), rheofact != 1 on some ice cell.  The six operands the short cuts
// drop are then read from their arrays at level C, every subcycle, by the thread that owns the cell (the LDS has no room for six
// more compact planes and the registers none for six more doubles across a subcycle: they come from L2, the wait is once per
// subcycle), and the momentum step is the general one of stepu_C / stepv_C (ice_dyn_shared.F90:1090-1283), as in cg_one<FAST = false>.
template <bool AVGS, bool REVP, bool FOLD, bool SLOW>
__global__ __launch_bounds__(X *Y, 3) void cg_res(EvpCgrid A, EvpCgRes R)
{
    constexpr int PW = (REVP && !AVGS) ? X : LW;         // row stride of the window-only planes
    constexpr int NQ = PW * Y;
    constexpr int NPC = REVP ? 18 : 16;
    // planes a level reads one position beyond the window (velocities, the averaging weights, dyE / dxN for the boundary ratios):
    // (Y+1) x (X+1); planes read inside the window only: Y x X, index = thread index
    __shared__ double s_uE[NP], s_vN[NP], s_ea[NP], s_na[NP], s_dyE[NP], s_dxN[NP];
    __shared__ double s_ua[NQ], s_ta[AVGS ? 1 : NQ], s_sh[NQ], s_eta[NQ], s_sp[NQ], s_sm[NQ], s_s12[NQ];
    // per-call operands of the momentum step, by owned cell (read at level C only): 0-5 uocnE vocnE facE emassdti fmE forcexE,
    // 6-11 the same at N, 12-15 earear 1/dxE narear 1/dyN, REVP: 16-17 revp * uvelE_init, revp * vvelN_init.
    __shared__ double s_pc[NPC][13 * 13];
    // the table of source cells lives, during the prologue, where stress12U's plane is afterwards (ints; the plane is filled once
    // every thread has read the table for the last time)
    int *const s_src = reinterpret_cast<int *>(&s_s12[0]);
    static_assert(sizeof(double) * NQ >= sizeof(int) * NP, "s_src does not fit its alias");
    __shared__ uint8_t s_gm[NP];           // land masks of the position's cell: bit0 epm, 1 npm, 2 uvm, 3 hm; bit 4: its window runs
    __shared__ int s_bad;
    // FOLD: raw values of the two rows at the fold, [.][0] the window's own fold row, [.][1] the mirrored one: uvelU, vvelU, uvelN, vvelE
    // (level S), the new vvelN (level C)
    __shared__ double s_fr[FOLD ? 5 : 1][2][LW];
    auto FR = [](int k) constexpr { return FOLD ? k : 0; };       // (planes 1 .. 4 exist in fold windows only: the accesses below are dead code elsewhere)

    const int t = threadIdx.x;
    const int tx = t & (X - 1), ty = t / X;
    int li = ty * LW + tx;
    const int tile = R.order ? R.order[blockIdx.x] : (int)blockIdx.x;        // (the windows with ice of this call)
    // test hooks (test build only; CICE_EVP_HIP_CGRID_RES_DEBUG): 8 = every fourth window lags 10 us per subcycle (the results must not
    // change), 16 = window 1 never shows up in a real launch (every wait on its records gives up: the caller must hear about it)
    if ((CGRES_DBG(R) & 16) && tile == 1 && !R.dry) return;
    const int4 tl = R.tiles[tile];
    const int4 q = A.blk[tl.x];
    const int i = tl.y - 2 + tx, j = tl.z - 2 + ty;
    const int nx = A.nx;
    const EvpScalars &p = A.p;
    const double relax = 1.0 - p.arlx1i * p.revp;
    // FOLD: cls 0 an ordinary position, 1 the window's fold row (global row NY), 2 the mirrored rows NY-2 / NY-1, 3 the mirrored fold
    // row, 4 unused; jmax: the last owned row (a window under the fold windows stops where they start)
    const bool foldwin = FOLD && (tl.w & 1);
    const int tf = FOLD ? (tl.w >> 8) & 255 : 0;
    const int jmax = FOLD ? (tl.w >> 16) : q.w;
    const int cls = !foldwin ? 0 : ty < tf ? 0 : ty == tf ? 1 : ty < tf + 3 ? 2 : ty == tf + 3 ? 3 : 4;
    const bool onf = FOLD && (cls == 1 || cls == 3);
    const int orow = cls == 1 ? tf + 3 : tf;          // the row across the fold
    const int frow = cls == 3 ? 1 : 0;
    // "north" of a position in a row at the fold: E-face / corner type at column 15 - tx of the other row, centre type at 16 - tx;
    // one step east there is one column to the west (hx)
    const int nE = onf ? orow * LW + 15 - tx : li + LW;
    const int nC = onf ? orow * LW + 16 - tx : li + LW;
    const int hx = onf ? -1 : 1;
    // fb: bit 0 the corner's column is the lower one of its pair, 1 the corner is a pole (its own partner), 2 the N face's column is the
    // lower one, 3 the same for the N face one column to the east (in tile orientation)
    unsigned fb = 0;
    if (onf) {
        const int4 t2 = R.tiles2[tile];
        const int NXg = t2.y;
        int ig = ((cls == 1 ? t2.x + tx : NXg - t2.x - 15 + tx) - 1) % NXg;
        if (ig < 0) ig += NXg;
        ig += 1;
        const int ige = ig == NXg ? 1 : ig + 1;
        fb = (ig < NXg / 2 ? 1u : 0u) | ((ig == NXg / 2 || ig == NXg) ? 2u : 0u) | (ig <= NXg / 2 ? 4u : 0u) | (ige <= NXg / 2 ? 8u : 0u);
    }
    auto G = [&](int k) { return R.gbase + (size_t)k * R.stride; };
    auto IN = [&](int k) { return R.inbase + (size_t)k * R.stride; };

    // ---- the window's tiles: source cell, velocities, the operands neighbours read --------------------------------
    for (int e = t; e < NP; e += X * Y) {
        const int lr = R.tab[(size_t)tile * NP + e];
        const size_t c = (size_t)(lr < 0 ? -1 - lr : lr);
        s_src[e] = lr;
        s_uE[e] = R.uE_in[c];
        s_vN[e] = R.vN_in[c];
        s_ea[e] = G(CG_EAREA)[c];
        s_na[e] = G(CG_NAREA)[c];
        s_dyE[e] = G(CG_DYE)[c];
        s_dxN[e] = G(CG_DXN)[c];
        if (e % LW < X && e / LW < Y) {
            s_ua[(e / LW) * PW + e % LW] = G(CG_UAREA)[c];
            if (!AVGS) s_ta[(e / LW) * PW + e % LW] = G(CG_TAREA)[c];
        }
        s_gm[e] = (uint8_t)(R.gmask[c] | ((!R.live || R.live[c]) ? 16u : 0u));       // bit 4: the cell's window runs in this call
    }
    if (t == 0) s_bad = 0;
    const int lr = R.tab[(size_t)tile * NP + li];
    const bool stat = lr < 0;
    const size_t L = (size_t)(stat ? -1 - lr : lr);
    const unsigned m = A.mask[L];
    const bool own = tx >= 2 && tx <= X - 2 && ty >= 2 && ty <= Y - 2 && i <= q.y && j <= jmax;
    // FOLD: level T and U in the mirrored rows NY-1 and NY only (row NY-2 feeds their shearU), nothing above the mini-tile
    const bool rowTU = !foldwin || ty <= tf || ty == tf + 2 || ty == tf + 3;
    const bool compS = !stat && (m & 2u) && cls != 4;
    const bool compT = tx >= 1 && ty >= 1 && !stat && (m & 1u) && rowTU;
    const bool compU = !stat && tx <= X - 2 && (foldwin ? rowTU : ty <= Y - 2);
    const bool pub = own && R.pubmap[L] != 0;
    // FOLD: the mirrored partners of the owned cells of the fold row work out their raw vvelN too (the N-face half of level C)
    const bool pown = FOLD && cls == 3 && tx >= 2 && tx <= X - 2;
    // what a position that does not compute a level shows its neighbours: the array's value, unchanged through the loop
    // (a corner ON the fold without ice: strain_rates_U's zero fill, which the halo update then averages with its partner)
    int pt = ty * PW + tx;               // this position in the window-only planes
    const int nCp = onf ? orow * PW + (tx ? 16 - tx : 15) : pt + PW;     // centre type "north" / the corner's partner, in those planes (tx = 0: unused)
    const int pU = onf ? orow * PW + 15 - tx : pt;
    s_sh[pt] = (onf && !compS && !stat) ? 0.0 : A.f[CF_SHEARU][L];
    s_eta[pt] = A.f[CF_ETA][L];
    double sp = R.sp_in[L], sm = R.sm_in[L], s12v = R.s12_in[L];
    s_sp[pt] = sp;
    s_sm[pt] = sm;
    double s12T = own ? A.f[CF_S12T][L] : 0.0;
    __syncthreads();

    // ---- operands for the whole call: registers for what every subcycle reads first, LDS (by owned cell) for what only the
    // momentum step reads, one word of bits for the 0 / 1 land masks -------------------------------------------------------
    const size_t cE = (size_t)(s_src[li + 1] < 0 ? -1 - s_src[li + 1] : s_src[li + 1]);
    // (a row at the fold: the array neighbour -- the ghost row beyond the fold holds the grid's own values)
    const size_t cN = onf ? L + nx : (size_t)(s_src[li + LW] < 0 ? -1 - s_src[li + LW] : s_src[li + LW]);
    // mb: bit 0 epm, 1 npm, 2 uvm of the cell; 3 npm of its east, 4 epm of its north neighbour; 5-8 hm of the cell, east, north,
    // north-east; classic EVP: 9 / 10 the signs of revp * uvelE_init, revp * vvelN_init
    unsigned mb;
    {
        const unsigned gmo = s_gm[li], gme = s_gm[li + 1], gmnE = s_gm[nE], gmn = s_gm[nC], gmne = s_gm[nC + hx];
        mb = (gmo & 7u) | ((gme & 2u) << 2) | ((gmnE & 1u) << 4) | ((gmo & 8u) << 2) | ((gme & 8u) << 3) | ((gmn & 8u) << 4) | ((gmne & 8u) << 5);
    }
    auto bit = [&](unsigned k) -> double { return (mb >> k) & 1u ? 1.0 : 0.0; };
    double dxU = 0.0, dyU = 0.0, ddyN = 0.0, ddxE = 0.0;
    if (!stat) {
        dxU = G(CG_DXU)[L]; dyU = G(CG_DYU)[L];
        ddyN = G(CG_DYN)[cE] - G(CG_DYN)[L];
        ddxE = G(CG_DXE)[cN] - G(CG_DXE)[L];
    }
    double dxT2 = 0.0, dyT2 = 0.0, uareaavgr = 0.0, strength = 0.0, DminT = 0.0;
    if (compT) {
        const double dxT = G(CG_DXT)[L], dyT = G(CG_DYT)[L];
        dxT2 = dxT * dxT; dyT2 = dyT * dyT;
        uareaavgr = 1.0 / (s_ua[pt] + s_ua[pt - PW] + s_ua[pt - PW - 1] + s_ua[pt - 1]);
        strength = IN(CI_STRENGTH)[L];
        DminT = G(CG_DMINT)[L];
    }
    double wtmpU = 0.0, strU = 0.0;      // avg_zeta: the sum of the weights of the T -> U average; AVGS: DminUarea, strengthU
    if (compU) {
        if (AVGS) { wtmpU = A.deltaminEVP * G(CG_UAREA)[L]; strU = A.strengthU[L]; }
        else wtmpU = (bit(5) * s_ta[pt] + bit(6) * s_ta[pt + 1] + bit(7) * s_ta[nCp] + bit(8) * s_ta[nCp + hx]);
    }
    double hdyEr = 0, dyT2e = 0, dxU2s = 0;
    double hdxNr = 0, dxT2n = 0, dyU2w = 0;
    // owned cells only (FOLD: the partners of the fold row's owned cells take the row of s_pc after the last owned one)
    int oi = (pown ? tf - 1 : ty - 2) * 13 + (tx - 2);
    double zE0 = 0.0, zN0 = 0.0;                     // revp * uvelE_init, revp * vvelN_init (stored once the source table is done with)
    if (own) {
        const size_t cS = (size_t)(s_src[li - LW] < 0 ? -1 - s_src[li - LW] : s_src[li - LW]);
        const size_t cW = (size_t)(s_src[li - 1] < 0 ? -1 - s_src[li - 1] : s_src[li - 1]);
        const double dyE = G(CG_DYE)[L], dxE = G(CG_DXE)[L], dxN = G(CG_DXN)[L], dyN = G(CG_DYN)[L];
        const double dyTe = G(CG_DYT)[cE], dxUs = G(CG_DXU)[cS], dxTn = G(CG_DXT)[cN], dyUw = G(CG_DYU)[cW];
        const double dxTo = G(CG_DXT)[L], dyTo = G(CG_DYT)[L];
        dxT2 = dxTo * dxTo; dyT2 = dyTo * dyTo;      // (compT holds the same values; an owned cell without ice still needs them)
        s_pc[12][oi] = G(CG_EAREAR)[L]; hdyEr = 0.5 / dyE; s_pc[13][oi] = 1.0 / dxE;
        dyT2e = dyTe * dyTe; dxU2s = dxUs * dxUs;
        s_pc[14][oi] = G(CG_NAREAR)[L]; hdxNr = 0.5 / dxN; s_pc[15][oi] = 1.0 / dyN;
        dxT2n = dxTn * dxTn; dyU2w = dyUw * dyUw;
        s_pc[0][oi] = IN(CI_UOCNE)[L]; s_pc[1][oi] = IN(CI_VOCNE)[L]; s_pc[2][oi] = A.facE[L]; s_pc[3][oi] = IN(CI_EMASSDTI)[L];
        s_pc[4][oi] = IN(CI_FME)[L]; s_pc[5][oi] = IN(CI_FORCEXE)[L];
        s_pc[6][oi] = IN(CI_UOCNN)[L]; s_pc[7][oi] = IN(CI_VOCNN)[L]; s_pc[8][oi] = A.facN[L]; s_pc[9][oi] = IN(CI_NMASSDTI)[L];
        s_pc[10][oi] = IN(CI_FMN)[L]; s_pc[11][oi] = IN(CI_FORCEYN)[L];
        zE0 = p.revp * IN(CI_UE_INIT)[L];
        zN0 = p.revp * IN(CI_VN_INIT)[L];
        if (!REVP) {       // classic EVP: zeros -- their signs as bits 9 / 10
            mb |= (__double_as_longlong(zE0) < 0 ? 1u : 0u) << 9;
            mb |= (__double_as_longlong(zN0) < 0 ? 1u : 0u) << 10;
        }
    }
    if (pown) {
        const size_t cW = (size_t)(s_src[li - 1] < 0 ? -1 - s_src[li - 1] : s_src[li - 1]);
        const double dxN = G(CG_DXN)[L], dyN = G(CG_DYN)[L], dxTn = G(CG_DXT)[cN], dyUw = G(CG_DYU)[cW], dxTo = G(CG_DXT)[L];
        dxT2 = dxTo * dxTo;
        s_pc[14][oi] = G(CG_NAREAR)[L]; hdxNr = 0.5 / dxN; s_pc[15][oi] = 1.0 / dyN;
        dxT2n = dxTn * dxTn; dyU2w = dyUw * dyUw;
        s_pc[6][oi] = IN(CI_UOCNN)[L]; s_pc[7][oi] = IN(CI_VOCNN)[L]; s_pc[8][oi] = A.facN[L]; s_pc[9][oi] = IN(CI_NMASSDTI)[L];
        s_pc[10][oi] = IN(CI_FMN)[L]; s_pc[11][oi] = IN(CI_FORCEYN)[L];
        zN0 = p.revp * IN(CI_VN_INIT)[L];
        mb |= (__double_as_longlong(zN0) < 0 ? 1u : 0u) << 10;
    }
    // The extra row / column of the reference's T list (ghost cells ihi+1, jhi+1: of what stressC_T computes there only stress12T
    // survives the exchange) is kept up by the window that owns the neighbouring interior cell -- by the threads of its column
    // tx = 0 and its row ty = 0, which have no T position of their own: thread (0, ty) serves the ghost cell of column ihi+1 in
    // row ty, thread (tx, 0) the one of row jhi+1 in column tx.  Such a thread runs level T like everybody else, at the ghost
    // position (tli), with the ghost cell's own strength, DminTarea and history; the planes hold the static operands of the
    // cell the position's value comes from, which equal the ghost cell's (cgres: images verified).
    // FOLD: the ghost row beyond the fold (row NY+1; the window's corner cell included) is a row of MIRRORED cells: stressC_T there
    // reads what the halo update left in the ghost cells -- the mirrored values, in the ghost cell's orientation (foldghost, below).
    int tli = li, t6 = pt;         // the position level T is evaluated at (index into the velocity tile and into the window-only planes)
    bool ghostT = false, foldghost = false;
    int fgx = 0;
    size_t g = 0;
    if ((tx == 0) != (ty == 0)) {
        // column ihi+1 (incl. the corner) / row jhi+1 as window positions
        const int gx_ = tx == 0 ? q.y + 1 - (tl.y - 2) : tx, gy_ = tx == 0 ? ty : q.w + 1 - (tl.z - 2);
        if (gx_ >= 1 && gx_ <= X - 1 && gy_ >= 1 && gy_ <= Y - 1) {
            const int gi = tl.y - 2 + gx_, gj = tl.z - 2 + gy_;
            const bool colg = gi == q.y + 1 && gj >= q.z && gj <= q.w + 1, rowg = gj == q.w + 1 && gi >= q.x && gi <= q.y;
            if ((tx == 0 && colg) || (ty == 0 && rowg)) {
                const int ii = min(gi, q.y), jj = min(gj, q.w);
                if (ii >= tl.y && ii <= tl.y + X - 4 && jj >= tl.z && jj <= tl.z + Y - 4 && jj <= jmax) {
                    g = (size_t)tl.x * A.plane + (size_t)(gj - 1) * nx + (gi - 1);
                    if (A.mask[g] & 1u) {
                        ghostT = true;
                        tli = gy_ * LW + gx_;
                        t6 = gy_ * PW + gx_;
                        if (foldwin && gj == q.w + 1) { foldghost = true; fgx = gx_; }
                    }
                }
            }
        }
    }
    if (ghostT) {
        const double dxT = G(CG_DXT)[g], dyT = G(CG_DYT)[g];
        dxT2 = dxT * dxT; dyT2 = dyT * dyT;
        if (foldghost) {        // corners of the ghost cell: NE, NW mirrored from row NY-1, SE, SW the fold row's own
            const int po = (tf + 2) * PW + 15 - fgx, ps = tf * PW + fgx;
            uareaavgr = 1.0 / (s_ua[po] + s_ua[ps] + s_ua[ps - 1] + s_ua[po + 1]);
        } else
        uareaavgr = 1.0 / (s_ua[t6] + s_ua[t6 - PW] + s_ua[t6 - PW - 1] + s_ua[t6 - 1]);
        strength = IN(CI_STRENGTH)[g];
        DminT = G(CG_DMINT)[g];
        s12T = A.f[CF_S12T][g];
        sp = 0.0; sm = 0.0;         // (stressC_T's own stresspT / stressmT there are overwritten by the exchange: not kept)
    }
    // level T in a row at the fold reads the corners ON the fold averaged: the partner of the evaluation position's own corner
    const bool doT = compT || ghostT;
    // FOLD: everything a thread knows about its place at the fold in ONE register -- bits 0-2 cls, 3 foldghost, 4 pown, 5-8 fb, 9-13 the ghost
    // cell's column; the remapped indices are worked out from it where a level needs them (a dozen integer operations against a dozen
    // registers held through the whole loop: the kernel has none to spare, tools/cgres_phases.py)
    const int fpk = FOLD ? (cls | (foldghost ? 8 : 0) | (pown ? 16 : 0) | ((int)fb << 5) | (fgx << 9)) : 0;
    struct FoldIdx { bool onf, pown, ghost; int cls, frow, nE, nCp, pU, hx, gx; unsigned fb; };
    auto fold_idx = [&](int f) -> FoldIdx {
        FoldIdx r;
        if (FOLD) asm volatile("" : "+v"(f));        // (not to be hoisted out of the subcycle loop)
        r.cls = f & 7;
        r.onf = FOLD && (r.cls == 1 || r.cls == 3);
        r.pown = FOLD && (f & 16);
        r.ghost = FOLD && (f & 8);
        r.fb = (unsigned)(f >> 5) & 15u;
        r.gx = (f >> 9) & 31;
        r.frow = r.cls == 3 ? 1 : 0;
        const int orw = r.cls == 1 ? tf + 3 : tf;
        r.nE = r.onf ? orw * LW + 15 - tx : li + LW;
        r.nCp = r.onf ? orw * PW + (tx ? 16 - tx : 15) : pt + PW;
        r.pU = r.onf ? orw * PW + 15 - tx : pt;
        r.hx = r.onf ? -1 : 1;
        return r;
    };
    const bool keepS12T = (own && compT) || ghostT;

    // ---- ring: the positions of the velocity tile this window does not produce -- at most one per thread, dealt round-robin to
    // the four waves (a thread with two entries polled them one after the other: two memory round trips on the critical path of
    // every subcycle; tools/cgres_phases.py showed the wave that held them all waiting twice as long as the others).
    // Entry e of the (Y+1) x (X+1) tile is polled if it lies outside the owned range and has a producer (not static);
    // (X, Y), the one entry no level reads, is left out
    // ... and lies within reach of the owned cells (EVP_CGRES_REACH positions beyond the last owned column / row; the rule of
    // halo_plan.h: cgres_in_reach, which also makes the publishers' map): a narrow window at a block's edge would otherwise poll far
    // into windows that do not poll it back, and the two-slot record protocol needs every dependency to be mutual
    const int last_ex = min(X - 2, 2 + q.y - tl.y), last_ey = min(Y - 2, 2 + jmax - tl.z);
    auto ring_src = [&](int e) -> int {
        const int ex = e % LW, ey = e / LW;
        const int gi = tl.y - 2 + ex, gj = tl.z - 2 + ey;
        const bool mine = ex >= 2 && ex <= X - 2 && ey >= 2 && ey <= Y - 2 && gi <= q.y && gj <= jmax;
        const bool reach = foldwin || (ex <= last_ex + EVP_CGRES_REACH && ey <= last_ey + EVP_CGRES_REACH);
        const int sc = s_src[e];
        return (mine || !reach || sc < 0 || !(s_gm[e] & 16u)) ? -1 : sc;        // (a cell of a window without ice: the value loaded above stays)
    };
    // All entries sit in wave 0, two per lane, both requested before either is looked at: ONE polling wave per workgroup.
    // (Measured on gx1, tools/cgres_phases.py: the entries dealt round-robin to all four waves, one per thread -- every wave of
    // every workgroup spinning -- 8.2 us per subcycle against 6.4 with most of them in one wave: the price of a poll sits in the
    // consumer CU's memory queue, which the workgroups that still compute on that CU need for their own publishes.)
    // A narrow window at a block's edge can have more than 128 entries: the rest goes to wave 1, then 2, 3 the same way.
    int ring0 = -1, ring_e = 0, ring1 = -1, ring_e1 = 0;
    {
        const int r0 = (t >> 6) * 128 + (t & 63), r1 = r0 + 64;
        int cnt = 0;
        for (int e = 0; e < NP - 1; ++e) {
            const int sc = ring_src(e);
            if (sc < 0) continue;
            if (cnt == r0) { ring0 = sc; ring_e = e; }
            if (cnt == r1) { ring1 = sc; ring_e1 = e; }
            ++cnt;
        }
    }
    auto give_up_note = [&](int k, int cell, unsigned seen, unsigned wanted) {
        if (atomicCAS(R.err, 0, 1) == 0) {
            R.err[1] = tile; R.err[2] = k; R.err[3] = cell; R.err[4] = (int)seen; R.err[5] = (int)wanted;
        }
    };
    // both entries of a lane: four loads in flight, then the tags; an entry that has arrived is not requested again
    auto poll2 = [&](const v4u *rd, unsigned want, int k) {
        v4u ra, rb, rc, rd2;
        bool need0 = ring0 >= 0, need1 = ring1 >= 0;
        unsigned spins = 0;
        while (need0 || need1) {
            if (need0 && need1) {
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                             "global_load_dwordx4 %2, %5, off sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd2)
                             : "v"(rd + 2 * (size_t)ring0), "v"(rd + 2 * (size_t)ring1)
                             : "memory");
            } else if (need0) {
                ld_rec2(rd + 2 * (size_t)ring0, ra, rb);
            } else {
                ld_rec2(rd + 2 * (size_t)ring1, rc, rd2);
            }
            if (need0 && ra.x == want && ra.w == want && rb.x == want && rb.w == want) {
                s_uE[ring_e] = unpack_rec(ra);
                s_vN[ring_e] = unpack_rec(rb);
                need0 = false;
            }
            if (need1 && rc.x == want && rc.w == want && rd2.x == want && rd2.w == want) {
                s_uE[ring_e1] = unpack_rec(rc);
                s_vN[ring_e1] = unpack_rec(rd2);
                need1 = false;
            }
            if (!(need0 || need1)) break;
            ++spins;
            if (spins > R.spin_limit || ((spins & 255u) == 0 && __hip_atomic_load(R.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                give_up_note(k, need0 ? ring0 : ring1, need0 ? ra.x : rc.x, want);
                s_bad = 1;
                return;
            }
            if (R.long_sleep) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
        }
    };
    __syncthreads();               // (every thread has read the source table for the last time: its plane takes stress12U)
    s_s12[pt] = s12v;
    if (REVP && (own || pown)) { s_pc[NPC - 2][oi] = zE0; s_pc[NPC - 1][oi] = zN0; }
    // initial records (tag of subcycle 0) so that the neighbours' first poll finds them; also the proof that they are resident
    if (pub) st_rec2((v4u *)R.rec[R.par0 & (EVP_CGRES_SLOTS - 1)] + 2 * L, pack_rec(s_uE[li], R.tag_base), pack_rec(s_vN[li], R.tag_base));
    __syncthreads();

    // One subcycle.  LAST (a compile-time constant: the loop body proper carries none of it) = the subcycle that ends the call, in
    // which the arrays nothing inside the loop reads are stored: shearU, deltaU, zetax2T, etax2T, etax2U, strintxE/yN, taubxE/yN.
    // Returns false when a wait gave up (every thread of the workgroup then leaves).
    // (test build) shader cycles per phase, accumulated over the launch by one lane per wave:
    // 0 poll | 1 barrier after it | 2 S | 3 barrier | 4 T | 5 barrier | 6 U + barrier | 7 C
    const bool prof = CGRES_PROF(R);
    unsigned long long pc0 = 0, pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (prof) pc0 = __builtin_readcyclecounter();
#define CG_STAMP(q)                                                     \
    if (prof) {                                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        pacc[q] += now_ - pc0;                                          \
        pc0 = now_;                                                     \
    }
    auto subcycle = [&](auto LASTC, int k) -> bool {
        constexpr bool LAST = decltype(LASTC)::value;
        const unsigned want = R.tag_base + (unsigned)k;
        // (four slots: a window that is read without reading back may be up to three subcycles ahead of its reader -- the host has
        // checked that it cannot be more, halo_plan.cpp: cgres_dependencies -- and still not overwrite what that reader waits for)
        const v4u *rd = (const v4u *)R.rec[(k + R.par0) & (EVP_CGRES_SLOTS - 1)];
        v4u *wr = (v4u *)R.rec[(k + 1 + R.par0) & (EVP_CGRES_SLOTS - 1)];
        // (the operand planes never change inside the loop: without an index the compiler cannot see through it hoists every one
        // of their loads into registers -- which is exactly what they are in LDS to avoid)
        // (made opaque IN PLACE: a copy per index would cost five more registers through the whole subcycle)
        asm volatile("" : "+v"(li), "+v"(tli), "+v"(oi), "+v"(pt), "+v"(t6));
        const int lo = li, to = tli, oo = oi, o6 = pt, to6 = t6;
        if ((CGRES_DBG(R) & 8) && (tile & 3) == 1) {
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < 1000ull) __builtin_amdgcn_s_sleep(8);
        }
        if (ring0 >= 0 || ring1 >= 0) poll2(rd, want, k);
        CG_STAMP(0)
        __syncthreads();
        if (s_bad) return false;
        CG_STAMP(1)

        // ---- S ----
        double uNo = 0.0, vEo = 0.0;
        const double uEo = s_uE[li], vNo = s_vN[li];
        // AVGS: deltaU in every subcycle, at every corner level U evaluates (its own uvelN / vvelE need the west / south neighbour)
        const bool fullS = AVGS ? (compS && tx >= 1 && ty >= 1) : (LAST && own);
        if (FOLD) {
            // the rows at the fold first work out the raw uvelU, vvelU, uvelN (all ON the fold) and vvelE (read across it) of their
            // positions, every position, ice or not (the reference's averages cover every cell), for both sides to pick up
            // (4 quantities x 2 rows x 16 columns = 128 single averages, one per thread of the waves 0 and 1 -- wave 0 the two-point, wave 1
            // the four-point ones: the threads of the two rows themselves would run the four divisions one after the other, with the
            // rest of the workgroup waiting at the barrier: tools/cgres_phases.py tx1, level S of the fold windows 4800 cycles against 2650)
            if (foldwin && t < 128) {
                int tt = t;
                asm volatile("" : "+v"(tt));             // (not to be hoisted out of the subcycle loop: registers)
                const int r = (tt >> 4) & 1, c = tt & 15, q = (tt >> 5) & 1;
                const int ps = (r ? tf + 3 : tf) * LW + c;            // the position ...
                const int pn = (r ? tf : tf + 3) * LW + 15 - c;       // ... and its E-face type "north" across the fold (vectors: sign changed)
                const unsigned gm = s_gm[ps];
                double v;
                if (tt < 64) {       // q = 0 uvelU = avg2(uvelE, uvelE north) * uvm, q = 1 vvelU = avg2(vvelN, vvelN east) * uvm
                    const double a0 = q ? s_vN[ps] : s_uE[ps], w0 = q ? s_na[ps] : s_ea[ps];
                    const double a1 = q ? s_vN[ps + 1] : -s_uE[pn], w1 = q ? s_na[ps + 1] : s_ea[pn];
                    v = avg2(a0, w0, a1, w1) * ((gm >> 2) & 1u ? 1.0 : 0.0);
                } else {             // q = 0 uvelN = avg4(W, O, NW, N) * npm, q = 1 vvelE = avg4(S, SE, O, E) * epm
                    const double a0 = q ? s_vN[ps - LW] : s_uE[ps - 1], w0 = q ? s_na[ps - LW] : s_ea[ps - 1];
                    const double a1 = q ? s_vN[ps - LW + 1] : s_uE[ps], w1 = q ? s_na[ps - LW + 1] : s_ea[ps];
                    const double a2 = q ? s_vN[ps] : -s_uE[pn + 1], w2 = q ? s_na[ps] : s_ea[pn + 1];
                    const double a3 = q ? s_vN[ps + 1] : -s_uE[pn], w3 = q ? s_na[ps + 1] : s_ea[pn];
                    v = avg4(a0, w0, a1, w1, a2, w2, a3, w3) * ((gm >> (q ? 0 : 1)) & 1u ? 1.0 : 0.0);
                    if (!q && c == 0) v = 0.0;
                }
                s_fr[(tt >> 6) * 2 + q][r][c] = v;
            }
            if (foldwin) __syncthreads();
        }
        // a vector ON the fold from the raw values of the pair of columns: lower column 0.5 * (x_lo - x_hi), higher column the negative
        // of that (cg_fold_gather: s * (0.5 * (x_a + isign * x_b))); a pole is its own partner: s * x_a
        auto fold_vec = [&](double me, double partner, bool lower) -> double {
            const double v = 0.5 * ((lower ? me : partner) + (-1.0) * (lower ? partner : me));
            return lower ? v : (-1.0) * v;
        };
        const FoldIdx FS = fold_idx(fpk);
        if (compS || own || FS.pown) {
            const bool onf = FS.onf;
            const int nE = FS.nE, frow = FS.frow;
            const unsigned fb = FS.fb;
            const double epc = bit(0), npc = bit(1), npe = bit(3), epn = bit(4);
            const double uEn = onf ? -s_uE[nE] : s_uE[li + LW], vNe = s_vN[li + 1];
            const double eao = s_ea[lo], ean = s_ea[FOLD ? nE : lo + LW], nao = s_na[lo], nae = s_na[lo + 1];
            double uU = 0.0, vU = 0.0;
            if (onf) {
                uNo = fold_vec(s_fr[FR(2)][frow][tx], s_fr[FR(2)][1 - frow][16 - tx], fb & 4u);
                vEo = s_fr[FR(3)][frow][tx];
                if (fb & 2u) { uU = (-1.0) * s_fr[0][frow][tx]; vU = (-1.0) * s_fr[FR(1)][frow][tx]; }
                else {
                    uU = fold_vec(s_fr[0][frow][tx], s_fr[0][1 - frow][15 - tx], fb & 1u);
                    vU = fold_vec(s_fr[FR(1)][frow][tx], s_fr[FR(1)][1 - frow][15 - tx], fb & 1u);
                }
            } else {
                if (own || (AVGS && fullS)) {
                    uNo = avg4(s_uE[li - 1], s_ea[lo - 1], uEo, eao, s_uE[li + LW - 1], s_ea[lo + LW - 1], uEn, ean) * npc;
                    vEo = avg4(s_vN[li - LW], s_na[lo - LW], s_vN[li - LW + 1], s_na[lo - LW + 1], vNo, nao, vNe, nae) * epc;
                }
                if (compS) {
                    const double uvm = bit(2);
                    uU = avg2(uEo, eao, uEn, ean) * uvm;
                    vU = avg2(vNo, nao, vNe, nae) * uvm;
                }
            }
            if (compS) {
                // The four boundary-condition ratios only ever meet the factor (npc - npe) or (epc - epn), +0 away from a coast, and
                // (+0 * mask) * ratio * velocity has the same bits for any finite negative ratio (the host has checked that all are):
                // -1 there; the lanes of a coastal corner work them out from the reference's start-up identities
                // ratiodxN = -dxN(i+1,j)/dxN(i,j), ratiodyE = -dyE(i,j+1)/dyE(i,j) and their reciprocals (ice_dyn_evp.F90:235-238;
                // verified bit for bit on the caller's arrays by derive_geometry_check)
                double rxN = -1.0, rxNr = -1.0, ryE = -1.0, ryEr = -1.0;
                if (npc != npe) { rxN = -(s_dxN[lo + 1] / s_dxN[lo]); rxNr = 1.0 / rxN; }
                if (epc != epn) { ryE = -(s_dyE[FOLD ? nE : lo + LW] / s_dyE[lo]); ryEr = 1.0 / ryE; }
                const double uEijp1 = uEn * epn + (epc - epn) * epc * ryE * uEo;
                const double uEij = uEo * epc + (epn - epc) * epn * ryEr * uEn;
                const double vNip1j = vNe * npe + (npc - npe) * npc * rxN * vNo;
                const double vNij = vNo * npc + (npe - npc) * npe * rxNr * vNe;
                const double sh = dxU * (uEijp1 - uEij) - uU * ddxE + dyU * (vNip1j - vNij) - vU * ddyN;
                s_sh[pt] = sh;
                if (fullS) {             // deltaU is wanted (avg_zeta: once per call, for the caller): the rest of strain_rates_U
                    double uNe, vEn;
                    if (onf) {           // uvelN one column to the east: ON the fold as well; vvelE beyond the fold: the mirrored cell's, sign changed
                        uNe = fold_vec(s_fr[FR(2)][frow][tx + 1], s_fr[FR(2)][1 - frow][15 - tx], fb & 8u);
                        vEn = -s_fr[FR(3)][1 - frow][15 - tx];
                    } else {
                        uNe = avg4(uEo, eao, s_uE[li + 1], s_ea[lo + 1], uEn, ean, s_uE[li + LW + 1], s_ea[lo + LW + 1]) * npe;
                        vEn = avg4(vNo, nao, vNe, nae, s_vN[li + LW], s_na[lo + LW], s_vN[li + LW + 1], s_na[lo + LW + 1]) * epn;
                    }
                    const double uNip1j = uNe * npe + (npc - npe) * npc * rxN * uNo;
                    const double uNij = uNo * npc + (npe - npc) * npe * rxNr * uNe;
                    const double vEijp1 = vEn * epn + (epc - epn) * epc * ryE * vEo;
                    const double vEij = vEo * epc + (epn - epc) * epn * ryEr * vEn;
                    const double dv = dyU * (uNip1j - uNij) + uU * ddyN + dxU * (vEijp1 - vEij) + vU * ddxE;
                    const double tn = dyU * (uNip1j - uNij) - uU * ddyN - dxU * (vEijp1 - vEij) + vU * ddxE;
                    const double delta = sqrt(dv * dv + p.e_factor * (tn * tn + sh * sh));
                    if (AVGS) s_eta[pt] = delta;
                    if (LAST && own && !R.dry) {
                        A.f[CF_SHEARU][L] = sh;
                        A.f[CF_DELTAU][L] = delta;
                        if (m & 16u) push1(A, L, A.f[CF_SHEARU], sh);
                    }
                }
            }
        }
        // (FOLD: a corner of the fold row without ice: the zero strain_rates_U fills in, for the fold step after the launch)
        if (FOLD && LAST && own && FS.cls == 1 && !compS && !R.dry) A.f[CF_SHEARU][L] = 0.0;
        CG_STAMP(2)
        __syncthreads();
        CG_STAMP(3)

        // ---- T ---- (at the thread's own position, or at the ghost position it serves)
        if (doT) {
            TOut r;
            if (FOLD) {
                // ONE evaluation with selected operands (the ghost cells beyond the fold are served by threads of wave 0: two branches would
                // run one after the other there, with the workgroup waiting).  The ghost cell (i, NY+1) is evaluated in its own orientation
                // on what the halo updates leave in its ghost neighbours: uvelE(i), uvelE(i-1) = -uvelE of row NY at the mirrored columns,
                // vvelN(i) = -vvelN of row NY-1, vvelN south of it = the fold row's own (averaged); shearU NE, NW = row NY-1's, SE, SW = the
                // fold row's averaged with their partners.  An ordinary position in a row at the fold: its own and its west corner averaged.
                const FoldIdx FT = fold_idx(fpk);
                const bool gh = FT.ghost;
                const int gx = FT.gx;
                const int a = (tf + 3) * LW + 15 - gx, c = (tf + 2) * LW + 16 - gx, d = tf * LW + gx;
                const int po = (tf + 2) * PW + 15 - gx, ps = tf * PW + gx, pp = (tf + 3) * PW + 15 - gx;
                const int iE0 = gh ? a : tli, iE1 = gh ? a + 1 : tli - 1, iN0 = gh ? c : tli, iN1 = gh ? d : tli - LW;
                const int p0 = gh ? po : t6, p1 = gh ? ps : t6 - PW, p2 = gh ? ps - 1 : t6 - PW - 1, p3 = gh ? po + 1 : t6 - 1;
                const int pT = (FT.cls == 1 ? tf + 3 : tf) * PW + 15 - (t6 - ty * PW);       // the partner of the evaluation position's own corner
                const bool f0 = !gh && FT.onf;
                double uE0 = s_uE[iE0], uE1 = s_uE[iE1], vN0 = s_vN[iN0];
                if (gh) { uE0 = -uE0; uE1 = -uE1; vN0 = -vN0; }
                double sh0 = s_sh[p0], sh1 = s_sh[p1], sh2 = s_sh[p2], sh3 = s_sh[p3];
                const double x0 = s_sh[f0 ? pT : p0], x3 = s_sh[f0 ? pT + 1 : p3], x1 = s_sh[gh ? pp : p1], x2 = s_sh[gh ? pp + 1 : p2];
                if (f0) { sh0 = 0.5 * (sh0 + x0); sh3 = 0.5 * (sh3 + x3); }
                if (gh) { sh1 = 0.5 * (sh1 + x1); sh2 = 0.5 * (sh2 + x2); }
                r = t_stress(p, uE0, uE1, vN0, s_vN[iN1], s_dyE[iE0], s_dyE[iE1], s_dxN[iN0], s_dxN[iN1], dxT2, dyT2, s_ua[p0], s_ua[p1], s_ua[p2], s_ua[p3],
                             uareaavgr, strength, DminT, sh0, sh1, sh2, sh3, sp, sm, relax);
            } else {
                r = t_stress(p, s_uE[tli], s_uE[tli - 1], s_vN[tli], s_vN[tli - LW], s_dyE[to], s_dyE[to - 1], s_dxN[to], s_dxN[to - LW], dxT2, dyT2,
                             s_ua[to6], s_ua[to6 - PW], s_ua[to6 - PW - 1], s_ua[to6 - 1], uareaavgr, strength, DminT, s_sh[t6], s_sh[t6 - PW],
                             s_sh[t6 - PW - 1], s_sh[t6 - 1], sp, sm, relax);
            }
            if (compT) {
                sp = r.sp; sm = r.sm;
                if (!AVGS) s_eta[pt] = r.etax2;       // (AVGS: the plane holds deltaU, level U does not read etax2T)
                s_sp[pt] = sp;
                s_sm[pt] = sm;
            }
            if (keepS12T) s12T = (s12T * relax + p.arlx1i * 0.5 * r.etax2 * r.shearT) * p.denom1;
            if (LAST && own && compT && !R.dry) {
                A.f[CF_ZETA][L] = r.zetax2;
                A.f[CF_ETA][L] = r.etax2;
                if (m & 16u) {
                    push1(A, L, A.f[CF_ZETA], r.zetax2);
                    push1(A, L, A.f[CF_ETA], r.etax2);
                }
            }
        }
        CG_STAMP(4)
        __syncthreads();
        CG_STAMP(5)

        // ---- U ----
        double etaU = 0.0;
        const FoldIdx FU = fold_idx(fpk);
        if (compU) {
            const bool onf = FU.onf;
            const int nCp = FU.nCp, hx = FU.hx, pU = FU.pU;
            double e2;
            if (AVGS) {
                double z, rp;
                visc_replpress(p, strU, wtmpU, s_eta[pt], z, e2, rp);
            } else {
                const double wU = FOLD ? (bit(5) * s_ta[o6] + bit(6) * s_ta[o6 + 1] + bit(7) * s_ta[nCp] + bit(8) * s_ta[nCp + hx]) : wtmpU;
                e2 = wU == 0.0 ? 0.0
                                  : (bit(5) * s_eta[pt] * s_ta[o6] + bit(6) * s_eta[pt + 1] * s_ta[o6 + 1] + bit(7) * s_eta[nCp] * s_ta[FOLD ? nCp : o6 + PW] +
                                     bit(8) * s_eta[nCp + hx] * s_ta[FOLD ? nCp + hx : o6 + PW + 1]) / wU;
            }
            etaU = e2;
            // (a corner ON the fold: shearU as the halo update left it -- averaged with its partner)
            const double shU = onf ? 0.5 * (s_sh[pt] + s_sh[pU]) : s_sh[pt];
            const double upd = (s12v * relax + p.arlx1i * 0.5 * e2 * shU) * p.denom1;
            if (m & 2u) {
                s12v = upd;
                s_s12[pt] = upd;
            } else if (onf) {
                s_s12[pt] = s12v;        // (without ice: the raw value is the history, which the average below keeps changing)
            }
        }
        __syncthreads();
        // FOLD: stress12U ON the fold -- the history takes the average of the two raw values, ice or not
        if (FU.onf && compU) s12v = 0.5 * (s12v + s_s12[FU.pU]);
        CG_STAMP(6)

        // ---- C ---- (FOLD: the mirrored partners of the fold row's owned cells run the N-face half for the raw vvelN)
        double vout = 0.0;
        const FoldIdx FC = fold_idx(fpk);
        if (own || FC.pown) {
            const bool onf = FC.onf;
            const int pU = FC.pU, nCp = FC.nCp, frow = FC.frow;
            const double s12c = s12v, s12s = s_s12[pt - PW];
            const double s12w = onf ? 0.5 * (s_s12[pt - 1] + s_s12[pU + 1]) : s_s12[pt - 1];       // (the west corner of a fold-row cell: ON the fold)
            const double spc = sp, smc = sm;
            double unew, vnew, strintx, strinty, taubx, tauby;
            // SLOW: the six operands the short cuts drop, from their arrays (the cell's own: L is the position's source cell)
            double g_rheoE = 1.0, g_rheoN = 1.0, g_wxE = 0.0, g_wyN = 0.0, g_tbE = 0.0, g_tbN = 0.0;
            if (SLOW) {
                g_rheoE = IN(CI_RHEOE)[L]; g_rheoN = IN(CI_RHEON)[L];
                g_wxE = IN(CI_WATERXE)[L]; g_wyN = IN(CI_WATERYN)[L];
                g_tbE = IN(CI_TBE)[L]; g_tbN = IN(CI_TBN)[L];
            }
            // (FOLD: the partner threads run the E-face half too, on operands nobody set and for nobody to read -- one basic block with the
            // N-face half, as in the plain variant)
            {
                const double spe = s_sp[pt + 1], sme = s_sm[pt + 1];
                const double uocnE = s_pc[0][oo], vocnE = s_pc[1][oo], facE = s_pc[2][oo], massE = s_pc[3][oo], fmE = s_pc[4][oo], forcexE = s_pc[5][oo];
                const double zE = REVP ? s_pc[NPC - 2][oo] : ((mb >> 9) & 1u ? -0.0 : 0.0);     // revp * uvelE_init
                const double strintx_core = 0.5 * s_dyE[lo] * (spe - spc) + (FOLD ? 0.5 / s_dyE[lo] : hdyEr) * ((dyT2e)*sme - (dyT2)*smc) + s_pc[13][oo] * ((dxU * dxU) * s12c - (dxU2s)*s12s);
                strintx = s_pc[12][oo] * (strintx_core);
                if (SLOW) strintx = (g_rheoE * s_pc[12][oo]) * (strintx_core);
                const double uold = uEo, vold = vEo;
                const double du = uocnE - uold, dv = vocnE - vold;
                const double vrel = facE * sqrt(du * du + dv * dv);
                const double taux = vrel * (SLOW ? g_wxE : uocnE);
                double Cb = 0.0;
                if (SLOW) Cb = g_tbE / (sqrt(uold * uold + vold * vold) + p.u0);
                const double cca = (p.brlx + p.revp) * massE + vrel * p.cosw + Cb;
                const double ccb = fmE + copysign(1.0, fmE) * vrel * p.sinw;
                const double cc1 = strintx + forcexE + taux + massE * (p.brlx * uold + zE);
                unew = (ccb * vold + cc1) / cca;
                taubx = -unew * Cb;
            }
            {
                const double spn = s_sp[nCp], smn = s_sm[nCp];
                const double uocnN = s_pc[6][oo], vocnN = s_pc[7][oo], facN = s_pc[8][oo], massN = s_pc[9][oo], fmN = s_pc[10][oo], forceyN = s_pc[11][oo];
                const double zN = REVP ? s_pc[NPC - 1][oo] : ((mb >> 10) & 1u ? -0.0 : 0.0);    // revp * vvelN_init
                const double strinty_core = 0.5 * s_dxN[lo] * (spn - spc) - hdxNr * ((dxT2n)*smn - (dxT2)*smc) + s_pc[15][oo] * ((dyU * dyU) * s12c - (dyU2w)*s12w);
                strinty = (SLOW ? g_rheoN * s_pc[14][oo] : s_pc[14][oo]) * (strinty_core);
                const double uold = uNo, vold = vNo;
                const double du = uocnN - uold, dv = vocnN - vold;
                const double vrel = facN * sqrt(du * du + dv * dv);
                const double tauy = vrel * (SLOW ? g_wyN : vocnN);
                double Cb = 0.0;
                if (SLOW) Cb = g_tbN / (sqrt(uold * uold + vold * vold) + p.u0);
                const double cca = (p.brlx + p.revp) * massN + vrel * p.cosw + Cb;
                const double ccb = fmN + copysign(1.0, fmN) * vrel * p.sinw;
                const double cc2 = strinty + forceyN + tauy + massN * (p.brlx * vold + zN);
                vnew = (-ccb * uold + cc2) / cca;
                tauby = -vnew * Cb;
            }
            // (nobody reads the velocity tile between the barrier above and the one after the next poll)
            vout = (m & 8u) ? vnew : vNo;
            if (onf) s_fr[FR(4)][frow][tx] = vout;        // raw: both sides of the fold, for the average below
            if (own) {
                const double uout = (m & 4u) ? unew : uEo;
                s_uE[li] = uout;
                if (!onf) {
                    s_vN[li] = vout;
                    if (pub) st_rec2(wr + 2 * L, pack_rec(uout, want + 1u), pack_rec(vout, want + 1u));
                }
                if (LAST && !R.dry) {
                    if (!AVGS) A.f[CF_ETAU][L] = etaU;      // (avg_strength: the reference never stores etax2U)
                    if (m & 4u) { A.f[CF_STRX][L] = strintx; A.f[CF_TAUBX][L] = taubx; }
                    if (m & 8u) { A.f[CF_STRY][L] = strinty; A.f[CF_TAUBY][L] = tauby; }
                }
            }
        }
        if (FOLD && foldwin) {
            // vvelN ON the fold: what the halo update makes of the two raw values, on the tile and in the record
            __syncthreads();
            if (own && FC.cls == 1) {
                const double v = fold_vec(vout, s_fr[FR(4)][1][16 - tx], FC.fb & 4u);
                s_vN[li] = v;
                if (pub) st_rec2(wr + 2 * L, pack_rec(s_uE[li], want + 1u), pack_rec(v, want + 1u));
            }
        }
        CG_STAMP(7)
        return true;
    };
    for (int k = 0; k < R.nsub - 1; ++k)
        if (!subcycle(std::false_type{}, k)) return;
    if (!subcycle(std::true_type{}, R.nsub - 1)) return;
#undef CG_STAMP
    if (prof && (t & 63) == 0) {
        unsigned long long *o = R.prof + ((size_t)tile * 4 + (t >> 6)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = pacc[e];
    }

    // ---- the state goes back (both ping-pong allocations: whichever schedule runs next finds it) -------------------
    if (R.dry) return;
    if (own) {
        const double uo = s_uE[li], vo = s_vN[li];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (m & 1u) {
                R.sp_out[b][L] = sp;
                R.sm_out[b][L] = sm;
                if (m & 16u) { push1(A, L, R.sp_out[b], sp); push1(A, L, R.sm_out[b], sm); }
            }
            if ((m & 2u) || cls == 1) {          // (ON the fold: the average changes a corner without ice too)
                R.s12_out[b][L] = s12v;
                if (m & 16u) push1(A, L, R.s12_out[b], s12v);
            }
            if (m & 4u) {
                R.uE_out[b][L] = uo;
                if (m & 16u) push1(A, L, R.uE_out[b], uo);
            }
            if ((m & 8u) || cls == 1) {
                R.vN_out[b][L] = vo;
                if (m & 16u) push1(A, L, R.vN_out[b], vo);
            }
        }
        if (m & 1u) A.f[CF_S12T][L] = s12T;
    }
    if (ghostT) {        // (the ghost cell's index again, from the position it was evaluated at: not worth two registers through the loop)
        const int gx_ = tli % LW, gy_ = tli / LW;
        A.f[CF_S12T][(size_t)tl.x * A.plane + (size_t)(tl.z - 2 + gy_ - 1) * nx + (tl.y - 2 + gx_ - 1)] = s12T;
    }
}

}  // namespace

// Per call (after the mask byte is composed): which windows hold ice.  One workgroup per window: any of the four ice masks set in the
// source cell of any position, or in any array cell of the window's footprint in its block (the ghost T cells of column ihi+1 / row
// jhi+1 it serves are array cells there).  live_win[w] = 0 / 1; live_cell[c] = live_win of the window that owns c.
namespace {
__global__ __launch_bounds__(256) void cg_res_live(EvpCgrid A, const int *__restrict__ tab, const int4 *__restrict__ tiles, int fold, int *live_win,
                                                    uint8_t *live_cell)
{
    const int w = blockIdx.x, t = threadIdx.x;
    const int4 tl = tiles[w];
    const int4 q = A.blk[tl.x];
    int any = 0;
    for (int e = t; e < NP; e += 256) {
        const int lr = tab[(size_t)w * NP + e];
        if (lr >= 0) any |= A.mask[lr] & 15u;
        const int i = tl.y - 2 + e % LW, j = tl.z - 2 + e / LW;
        if (i >= 1 && i <= A.nx && j >= 1 && j <= A.ny) any |= A.mask[(size_t)tl.x * A.plane + (size_t)(j - 1) * A.nx + (i - 1)] & 15u;
    }
    const int live = __syncthreads_or(any) ? 1 : 0;
    if (t == 0) live_win[w] = live;
    const int jmax = fold ? (tl.w >> 16) : q.w;
    for (int e = t; e < NP; e += 256) {
        const int ex = e % LW, ey = e / LW, i = tl.y - 2 + ex, j = tl.z - 2 + ey;
        if (ex >= 2 && ex <= X - 2 && ey >= 2 && ey <= Y - 2 && i <= q.y && j <= jmax) live_cell[tab[(size_t)w * NP + e]] = (uint8_t)live;
    }
}
}  // namespace

void evp_launch_cgrid_res_live(const EvpCgrid &A, const int *tab, const int4 *tiles, int ntiles, int fold, int *live_win, uint8_t *live_cell, hipStream_t st)
{
    if (ntiles > 0) hipLaunchKernelGGL(cg_res_live, dim3(ntiles), dim3(256), 0, st, A, tab, tiles, fold, live_win, live_cell);
}

// Per call: the loop's state in pairs of ghost cells OUTSIDE the domain that the kernel treats as one position (pairs: host-built,
// evp_host_cgrid.cpp build_res_tables) must agree bit for bit -- uvelE, vvelN, stresspT, stressmT, stress12U.  Bit 8 of *flags else.
namespace {
struct Five { const double *f[5]; };
__global__ void cg_res_pair_check(Five F, const int2 *__restrict__ pairs, int n, unsigned *flags)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int2 pr = pairs[k];
    bool bad = false;
#pragma unroll
    for (int a = 0; a < 5; ++a) bad |= __double_as_longlong(F.f[a][pr.x]) != __double_as_longlong(F.f[a][pr.y]);
    if (bad) atomicOr(flags, 256u);
}
}  // namespace

void evp_launch_cgrid_res_pair_check(const double *const *five, const int2 *pairs, int n, unsigned *flags, hipStream_t st)
{
    if (n <= 0) return;
    Five F;
    for (int a = 0; a < 5; ++a) F.f[a] = five[a];
    hipLaunchKernelGGL(cg_res_pair_check, dim3((n + 255) / 256), dim3(256), 0, st, F, pairs, n, flags);
}

namespace {
// Workgroups of one variant a CU holds at once: what the runtime says, and not more than the LDS allows when a workgroup's share is
// rounded up to 2 KB (the revised-EVP avg_zeta FOLD variant, 54 248 B, was reported as three per CU and ran as two: its waits gave up)
template <class K>
int blocks_per_cu(K kernel)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, X * Y, 0) != hipSuccess) return 0;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kernel)) != hipSuccess) return 0;
    const size_t share = (fa.sharedSizeBytes + 2047) / 2048 * 2048;
    // the CU's LDS as the device reports it (160 KB on gfx950), not a constant of this file
    int dev = 0, lds = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds <= 0)
        return 0;
    if (share) nb = std::min<int>(nb, (int)((size_t)lds / share));
    return nb;
}
}  // namespace

int evp_cgrid_res_max_blocks_per_cu(int avg_strength, int revised, int fold, int slow)
{
#define CGRES_PICK(F)                                                                                                          \
    do {                                                                                                                       \
        if (fold) {                                                                                                            \
            if (slow) { if (avg_strength) { if (revised) F((cg_res<true, true, true, true>)); else F((cg_res<true, false, true, true>)); }      \
                        else { if (revised) F((cg_res<false, true, true, true>)); else F((cg_res<false, false, true, true>)); } }               \
            else { if (avg_strength) { if (revised) F((cg_res<true, true, true, false>)); else F((cg_res<true, false, true, false>)); }         \
                   else { if (revised) F((cg_res<false, true, true, false>)); else F((cg_res<false, false, true, false>)); } }                  \
        } else {                                                                                                               \
            if (slow) { if (avg_strength) { if (revised) F((cg_res<true, true, false, true>)); else F((cg_res<true, false, false, true>)); }    \
                        else { if (revised) F((cg_res<false, true, false, true>)); else F((cg_res<false, false, false, true>)); } }             \
            else { if (avg_strength) { if (revised) F((cg_res<true, true, false, false>)); else F((cg_res<true, false, false, false>)); }       \
                   else { if (revised) F((cg_res<false, true, false, false>)); else F((cg_res<false, false, false, false>)); } }                \
        }                                                                                                                      \
    } while (0)
    const int revised_ = revised;
    (void)revised_;
#define CGRES_OCC(K) return blocks_per_cu(K)
    CGRES_PICK(CGRES_OCC);
#undef CGRES_OCC
    return 0;
}

void evp_launch_cgrid_res(const EvpCgrid &A, const EvpCgRes &R, hipStream_t st)
{
    const dim3 grid(R.ntiles), block(X * Y);
    const bool revised = A.p.revp != 0.0;
    const int fold = R.fold, slow = R.slow, avg_strength = A.avg_strength;
#define CGRES_LAUNCH(K) hipLaunchKernelGGL(K, grid, block, 0, st, A, R)
    CGRES_PICK(CGRES_LAUNCH);
#undef CGRES_LAUNCH
#undef CGRES_PICK
}
