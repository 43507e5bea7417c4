// =====================================================================
// C-grid EVP subcycle for gfx950: u at the east faces (E), v at the north faces (N), stresses at
// cell centres (T) and corners (U).  Reference: evp()'s loop for grid_ice = 'C', ice_dyn_evp.F90:938-1099, with
// strain_rates_U / strain_rates_Tdt / stepu_C / stepv_C (ice_dyn_shared.F90:2291-2444, 1090-1290), stressC_T /
// stressC_U / div_stress_Ex / div_stress_Ny (ice_dyn_evp.F90:1758-1972, 2195-2416) and the grid_average_X2Y
// variants the loop calls (ice_grid.F90:4159-4606).
//
// One subcycle is five dependent stencil phases; between two phases every cell's neighbours must be complete, so
// each phase is one launch.  What the reference does with eight ice_HaloUpdate calls per subcycle is fused into
// the phases: the thread that produces a cell also stores it into the ghost cells that mirror it ("images",
// at most three: a corner cell of a doubly periodic block), so the next phase reads plain i+-1, j+-1 neighbours
// and no halo kernel runs inside the loop.  Fields the reference never exchanges (etax2U, deltaU, stress12T,
// strintxE/yN, taubxE/yN) are not pushed: their ghost cells end up exactly as the reference leaves them.
//
// Three schedules.  "one" (cg_one, further down; the default on one rank without a fold -- from 300 000 cells per rank with the interior
// of every block marched by cg_strip, at the end of this file, and cg_one's windows along the block edges): ONE launch per subcycle, the three
// dependent levels inside a workgroup, neighbouring positions recomputed.  "phases": the five phases as five launches (any
// visc_method; tripole grids, with a fold step after each).  "fused" (visc_method = avg_zeta; several ranks):
// three launches per subcycle --
//   A  face->face / face->corner averages of the PREVIOUS subcycle's velocities recomputed where strain_rates_U
//      needs them (own cell and the east / north neighbour), so that phase 4 disappears from the loop and runs once
//      after it; uvelN, vvelE are stored for stepu_C / stepv_C, the corner velocities stay in registers;
//   B  stressC_T as before;
//   C  stressC_U recomputed at the three corners div_stress_Ex / _Ny read (own, south, west: stress12U ping-pongs
//      between two buffers), then stepu_C / stepv_C.
// Arrays that nothing inside the loop reads (zetax2T, etax2U, deltaU, strintxE/yN, taubxE/yN) are stored in the last
// subcycle of a call only.  81 instead of 99 doubles moved per cell and subcycle, 3 instead of 5 launches.
//
// fp64, strict: no FMA contraction, operations in the reference's order -- bit-identical to the reference
// compiled with -O2 -ffp-contract=off (tests/test_gpu_cgrid.py against the committed fixtures).
// HBM-bound on large grids (about 90 doubles moved per cell and subcycle), launch-bound on gx1-sized ones.
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evp_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int TX = 64, TY = 4;

__device__ __forceinline__ void push(const EvpCgrid &A, size_t c, unsigned m, int field, double v)
{
    (void)m;
    const int s = A.img_slot[c];
    if (s < 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int d = A.img_dst[3 * s + k];
        if (d >= 0) A.f[field][d] = v;
    }
}

// visc_replpress, ice_dyn_shared.F90:2446-2475
__device__ __forceinline__ void visc_replpress(const EvpScalars &p, double strength, double DminArea, double Delta,
                                               double &zetax2, double &etax2, double &rep_prs)
{
    // capping = 1 (the default, capping_method 'max'): the second quotient is multiplied by (1 - 1) = +0 and the product added --
    // a zero of the quotient's sign, which is the first quotient's sign too, so the sum IS the first quotient, bit for bit,
    // whenever the second one is finite (DminArea > 0 makes both denominators positive; a finite strength; a Delta that is
    // a number: with Delta = NaN the reference's sum is NaN -- 0 x NaN -- while fmax would return DminArea, so a blown-up
    // state takes the reference's expression and stays visible).  One division instead of two; any other case takes the
    // reference's expression.
    double tmpcalc;
    if (p.capping == 1.0 && DminArea > 0.0 && fabs(strength) <= 1.7976931348623157e308 && Delta <= 1.7976931348623157e308)
        tmpcalc = strength / fmax(Delta, DminArea);
    else
        tmpcalc = p.capping * (strength / fmax(Delta, DminArea)) + (1.0 - p.capping) * (strength / (Delta + DminArea));
    zetax2 = (1.0 + p.Ktens) * tmpcalc;
    rep_prs = (1.0 - p.Ktens) * tmpcalc * Delta;
    etax2 = p.epp2i * zetax2;
}

// array k of a table that is one allocation: base + k * stride (a scalar multiply-add where it is used, instead of
// one kernel-argument pointer per array held in scalar registers from the top of the kernel)
struct Slab {
    static constexpr bool derived = false;
    const double *base;
    size_t stride;
    __device__ __forceinline__ const double *operator[](int k) const { return base + (size_t)k * stride; }
};
// The same table with 15 of its 23 arrays DERIVED from the other eight instead of loaded (cg_one is HBM-bound on large
// grids, and nearly half of its bytes per cell are static geometry): the reference computes them once at start-up as
//   tarea = dxT*dyT, uarea = dxU*dyU, narea = dxN*dyN, earea = dxE*dyE          (ice_grid.F90:681-684)
//   earear = 1/earea, narear = 1/narea where the area is > 0, else 0            (ice_grid.F90:706-715)
//   ratiodxN = -dxN(i+1,j)/dxN(i,j), ratiodyE = -dyE(i,j+1)/dyE(i,j), ratiodxNr = 1/ratiodxN, ratiodyEr = 1/ratiodyE
//                                                                               (ice_dyn_evp.F90:235-238)
//   DminTarea = deltaminEVP*tarea                                               (ice_dyn_shared.F90: init_dyn_shared)
//   hm, uvm, npm, epm: land masks, 0 or 1                                       (ice_grid.F90:979, 3382-3385)
// cice_evp_hip_cgrid_set_geometry checks every one of these identities BIT FOR BIT on the caller's arrays (all cells the
// kernel can read) and hands out this view only if all hold, so a derived value IS the array's value; the masks travel
// as four bits of one byte.  k is a constant at every use: the switch folds.
struct DSlab {
    static constexpr bool derived = true;
    const double *base;
    size_t stride;
    const uint8_t *gm;              // bit 0 epm, 1 npm, 2 uvm, 3 hm
    int nx;
    double dmin;
    struct Acc {
        const DSlab &S;
        int k;
        __device__ __forceinline__ double raw(int a, size_t p) const { return S.base[(size_t)a * S.stride + p]; }
        __device__ __forceinline__ double operator[](size_t p) const
        {
            switch (k) {
            case CG_TAREA: return raw(CG_DXT, p) * raw(CG_DYT, p);
            case CG_UAREA: return raw(CG_DXU, p) * raw(CG_DYU, p);
            case CG_NAREA: return raw(CG_DXN, p) * raw(CG_DYN, p);
            case CG_EAREA: return raw(CG_DXE, p) * raw(CG_DYE, p);
            case CG_EAREAR: { const double a = raw(CG_DXE, p) * raw(CG_DYE, p); return a > 0.0 ? 1.0 / a : 0.0; }
            case CG_NAREAR: { const double a = raw(CG_DXN, p) * raw(CG_DYN, p); return a > 0.0 ? 1.0 / a : 0.0; }
            case CG_DMINT: return S.dmin * (raw(CG_DXT, p) * raw(CG_DYT, p));
            case CG_RXN: return -(raw(CG_DXN, p + 1) / raw(CG_DXN, p));
            case CG_RXNR: return 1.0 / -(raw(CG_DXN, p + 1) / raw(CG_DXN, p));
            case CG_RYE: return -(raw(CG_DYE, p + S.nx) / raw(CG_DYE, p));
            case CG_RYER: return 1.0 / -(raw(CG_DYE, p + S.nx) / raw(CG_DYE, p));
            case CG_EPM: return (S.gm[p] & 1u) ? 1.0 : 0.0;
            case CG_NPM: return (S.gm[p] & 2u) ? 1.0 : 0.0;
            case CG_UVM: return (S.gm[p] & 4u) ? 1.0 : 0.0;
            case CG_HM: return (S.gm[p] & 8u) ? 1.0 : 0.0;
            default: return raw(k, p);
            }
        }
    };
    __device__ __forceinline__ Acc operator[](int k) const { return Acc{*this, k}; }
};
// the fused (three-launch) kernels see the static table through the same two views: the kernel argument's pointer array, or
// the derived one (A.gmask set: the arrays are one allocation, A.g[k] = A.g[0] + k * A.gstride)
struct PtrTab {
    static constexpr bool derived = false;
    const double *const *g;
    __device__ __forceinline__ const double *operator[](int k) const { return g[k]; }
};
template <bool GEO> struct AGeo;
template <> struct AGeo<false> {
    static __device__ __forceinline__ PtrTab make(const EvpCgrid &A) { return PtrTab{A.g}; }
};
template <> struct AGeo<true> {
    static __device__ __forceinline__ DSlab make(const EvpCgrid &A) { return DSlab{A.g[0], A.gstride, A.gmask, A.nx, A.deltaminEVP}; }
};

struct Cell { int i, j, b; size_t o; int4 q; bool in; };
__device__ __forceinline__ Cell cell(const EvpCgrid &A)
{
    Cell c;
    int bx = blockIdx.x, by = blockIdx.y;
    c.b = blockIdx.z;
    if (A.xcd_rows > 0) {
        // Workgroups go to the 8 XCDs round-robin by their linear id, and every XCD has its own L2: with the plain
        // 2-D numbering the workgroup above and the one below always sit on different XCDs, so the rows they share are
        // fetched twice.  1-D launch: bands of xcd_rows workgroup rows, dealt to the XCDs group after group (a group =
        // 8 bands), short enough that a band's workgroups are resident together.
        const int gx = (A.nx + TX - 1) / TX, gy = (A.ny + TY - 1) / TY;
        const int R = A.xcd_rows;
        const int ngroups = (gy + 8 * R - 1) / (8 * R);
        const int per = gx * R * 8 * ngroups;            // workgroups per block of the domain
        const int L = blockIdx.x;
        c.b = L / per;
        const int r = L - c.b * per;
        const int k = r >> 3;
        const int g = k / (gx * R), kk = k - g * (gx * R);
        bx = kk % gx;
        by = (g * 8 + (r & 7)) * R + kk / gx;
    }
    c.i = bx * TX + threadIdx.x + 1;                 // 1-based, as the reference
    c.j = by * TY + threadIdx.y + 1;
    c.in = c.i <= A.nx && c.j <= A.ny;
    c.q = A.blk[c.b];
    c.o = (size_t)c.b * A.plane + (size_t)(c.j - 1) * A.nx + (c.i - 1);
    return c;
}

// ---- phase 0: strain_rates_U (strain rates * area at the corners); shearU is exchanged (:965-967) ----
// strain_rates_U proper (ice_dyn_shared.F90:2341-2444) on values in registers
struct StrainIn {
    double uNo, uNe, vEo, vEn, uEo, uEn, vNo, vNe, uU, vU;
};
template <class GT>
__device__ __forceinline__ void strain_u(const EvpCgrid &A, const GT &G, size_t o, const StrainIn &v, double &sh, double &delta)
{
    const size_t e = o + 1, n = o + A.nx;
    const auto epm = G[CG_EPM], npm = G[CG_NPM];
    const double dxU = G[CG_DXU][o], dyU = G[CG_DYU][o];
    const double ddyN = G[CG_DYN][e] - G[CG_DYN][o], ddxE = G[CG_DXE][n] - G[CG_DXE][o];
    const double npc = npm[o], npe = npm[e], epc = epm[o], epn = epm[n];
    // The four boundary-condition ratios only ever meet the factor (npc - npe) or (epc - epn), which is +0 away from a coast:
    // (+0 * mask) * ratio * velocity has the same bits for ANY finite negative ratio.  With the derived view (the host has
    // checked that every ratio is finite and negative) they are therefore worked out -- two divisions each pair -- only by
    // the waves that hold a coastal corner; everybody else takes -1.
    double rxN, rxNr, ryE, ryEr;
    if (GT::derived) {
        const bool nd = npc != npe, ed = epc != epn;
        rxN = nd ? G[CG_RXN][o] : -1.0; rxNr = nd ? G[CG_RXNR][o] : -1.0;
        ryE = ed ? G[CG_RYE][o] : -1.0; ryEr = ed ? G[CG_RYER][o] : -1.0;
    } else {
        rxN = G[CG_RXN][o]; rxNr = G[CG_RXNR][o]; ryE = G[CG_RYE][o]; ryEr = G[CG_RYER][o];
    }
    const double uNip1j = v.uNe * npe + (npc - npe) * npc * rxN * v.uNo;
    const double uNij = v.uNo * npc + (npe - npc) * npe * rxNr * v.uNe;
    const double vEijp1 = v.vEn * epn + (epc - epn) * epc * ryE * v.vEo;
    const double vEij = v.vEo * epc + (epn - epc) * epn * ryEr * v.vEn;
    const double dv = dyU * (uNip1j - uNij) + v.uU * ddyN + dxU * (vEijp1 - vEij) + v.vU * ddxE;
    const double tn = dyU * (uNip1j - uNij) - v.uU * ddyN - dxU * (vEijp1 - vEij) + v.vU * ddxE;
    const double uEijp1 = v.uEn * epn + (epc - epn) * epc * ryE * v.uEo;
    const double uEij = v.uEo * epc + (epn - epc) * epn * ryEr * v.uEn;
    const double vNip1j = v.vNe * npe + (npc - npe) * npc * rxN * v.vNo;
    const double vNij = v.vNo * npc + (npe - npc) * npe * rxNr * v.vNe;
    sh = dxU * (uEijp1 - uEij) - v.uU * ddxE + dyU * (vNip1j - vNij) - v.vU * ddyN;
    delta = sqrt(dv * dv + A.p.e_factor * (tn * tn + sh * sh));
}

// The shear alone (same operations as in strain_u): inside the loop only shearU is read -- divergence, tension and
// Delta at the corners, and with them the averages at the east and north neighbour (uNe, vEn) and the corner's own
// uNo, vEo, feed deltaU only, which visc_method = avg_zeta stores for the caller in the last subcycle of a call.
template <class GT>
__device__ __forceinline__ double shear_u(const EvpCgrid &A, const GT &G, size_t o, double uEo, double uEn, double vNo, double vNe,
                                          double uU, double vU)
{
    const size_t e = o + 1, n = o + A.nx;
    const auto epm = G[CG_EPM], npm = G[CG_NPM];
    const double dxU = G[CG_DXU][o], dyU = G[CG_DYU][o];
    const double ddyN = G[CG_DYN][e] - G[CG_DYN][o], ddxE = G[CG_DXE][n] - G[CG_DXE][o];
    const double npc = npm[o], npe = npm[e], epc = epm[o], epn = epm[n];
    // The four boundary-condition ratios only ever meet the factor (npc - npe) or (epc - epn), which is +0 away from a coast:
    // (+0 * mask) * ratio * velocity has the same bits for ANY finite negative ratio.  With the derived view (the host has
    // checked that every ratio is finite and negative) they are therefore worked out -- two divisions each pair -- only by
    // the waves that hold a coastal corner; everybody else takes -1.
    double rxN, rxNr, ryE, ryEr;
    if (GT::derived) {
        const bool nd = npc != npe, ed = epc != epn;
        rxN = nd ? G[CG_RXN][o] : -1.0; rxNr = nd ? G[CG_RXNR][o] : -1.0;
        ryE = ed ? G[CG_RYE][o] : -1.0; ryEr = ed ? G[CG_RYER][o] : -1.0;
    } else {
        rxN = G[CG_RXN][o]; rxNr = G[CG_RXNR][o]; ryE = G[CG_RYE][o]; ryEr = G[CG_RYER][o];
    }
    const double uEijp1 = uEn * epn + (epc - epn) * epc * ryE * uEo;
    const double uEij = uEo * epc + (epn - epc) * epn * ryEr * uEn;
    const double vNip1j = vNe * npe + (npc - npe) * npc * rxN * vNo;
    const double vNij = vNo * npc + (npe - npc) * npe * rxNr * vNe;
    return dxU * (uEijp1 - uEij) - uU * ddxE + dyU * (vNip1j - vNij) - vU * ddyN;
}

// grid_average_X2YA at cell p (ice_grid.F90:4388-4606): 'NW' (E -> N), 'SE' (N -> E), 'N' (E -> U), 'E' (N -> U)
template <class P, class W>
__device__ __forceinline__ double avg_nw(const P &a, const W &w, size_t p, int nx)
{
    const double wtmp = (w[p - 1] + w[p] + w[p + nx - 1] + w[p + nx]);
    if (wtmp == 0.0) return 0.0;
    return (a[p - 1] * w[p - 1] + a[p] * w[p] + a[p + nx - 1] * w[p + nx - 1] + a[p + nx] * w[p + nx]) / wtmp;
}
template <class P, class W>
__device__ __forceinline__ double avg_se(const P &a, const W &w, size_t p, int nx)
{
    const double wtmp = (w[p - nx] + w[p - nx + 1] + w[p] + w[p + 1]);
    if (wtmp == 0.0) return 0.0;
    return (a[p - nx] * w[p - nx] + a[p - nx + 1] * w[p - nx + 1] + a[p] * w[p] + a[p + 1] * w[p + 1]) / wtmp;
}
template <class P, class W>
__device__ __forceinline__ double avg_2(const P &a, const W &w, size_t p, size_t q)
{
    const double wtmp = (w[p] + w[q]);
    if (wtmp == 0.0) return 0.0;
    return (a[p] * w[p] + a[q] * w[q]) / wtmp;
}

// ---- fused A: what phase 4 of the previous subcycle would have stored, recomputed where this subcycle's
// strain_rates_U reads it.  The east / north neighbour may be a ghost cell: its value in the reference is the copy
// of the owner's average, which the same formula gives here from the (pushed) ghost velocities and the static
// ghost weights. ----
template <bool GEO>
__global__ __launch_bounds__(TX *TY) void cg_avg_strain(EvpCgrid A, int last)
{
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const size_t o = c.o, e = o + 1, n = o + A.nx;
    const unsigned m = A.mask[o];
    const auto G = AGeo<GEO>::make(A);
    const double *uE = A.f[CF_UE], *vN = A.f[CF_VN];
    const auto ea = G[CG_EAREA], na = G[CG_NAREA];
    const auto npm = G[CG_NPM], epm = G[CG_EPM];
    const double uNo = avg_nw(uE, ea, o, A.nx) * npm[o];
    const double vEo = avg_se(vN, na, o, A.nx) * epm[o];
    A.f[CF_UN][o] = uNo;                         // stepv_C / stepu_C of this subcycle read them (own cell)
    A.f[CF_VE][o] = vEo;
    // no early exit for cells without ice: every load below is in bounds, and issuing them all before the first
    // wait is what matters on grids this small (two waves per SIMD); only the stores are conditional
    const double uvm = G[CG_UVM][o];
    const double uU = avg_2(uE, ea, o, n) * uvm;
    const double vU = avg_2(vN, na, o, e) * uvm;
    double sh, delta = 0.0;
    if (last) {                                  // deltaU is wanted (nothing inside the loop reads it): the whole of strain_rates_U
        StrainIn v;
        v.uNo = uNo; v.vEo = vEo; v.uU = uU; v.vU = vU;
        v.uNe = avg_nw(uE, ea, e, A.nx) * npm[e];
        v.vEn = avg_se(vN, na, n, A.nx) * epm[n];
        v.uEo = uE[o]; v.uEn = uE[n]; v.vNo = vN[o]; v.vNe = vN[e];
        strain_u(A, G, o, v, sh, delta);
    } else {
        sh = shear_u(A, G, o, uE[o], uE[n], vN[o], vN[e], uU, vU);
    }
    if (!(m & 2u)) return;
    A.f[CF_SHEARU][o] = sh;
    if (last) A.f[CF_DELTAU][o] = delta;
    if (m & 16u) push(A, o, m, CF_SHEARU, sh);
}

__global__ __launch_bounds__(TX *TY) void cg_strain_u(EvpCgrid A)
{
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const size_t o = c.o, e = o + 1, n = o + A.nx;
    const unsigned m = A.mask[o];
    if (!(m & 2u)) {
        // strain_rates_U zero-fills shearU before computing the ice cells; on a tripole grid the fold step of the
        // previous subcycle may have stored an average into a fold-row cell without ice
        if (A.tripole) A.f[CF_SHEARU][o] = 0.0;
        return;
    }
    const double *uE = A.f[CF_UE], *vE = A.f[CF_VE], *uN = A.f[CF_UN], *vN = A.f[CF_VN];
    const double uU = A.f[CF_UU][o], vU = A.f[CF_VU][o];
    const double *epm = A.g[CG_EPM], *npm = A.g[CG_NPM];
    const double dxU = A.g[CG_DXU][o], dyU = A.g[CG_DYU][o];
    const double ddyN = A.g[CG_DYN][e] - A.g[CG_DYN][o], ddxE = A.g[CG_DXE][n] - A.g[CG_DXE][o];
    const double rxN = A.g[CG_RXN][o], rxNr = A.g[CG_RXNR][o], ryE = A.g[CG_RYE][o], ryEr = A.g[CG_RYER][o];
    const double npc = npm[o], npe = npm[e], epc = epm[o], epn = epm[n];
    const double uNip1j = uN[e] * npe + (npc - npe) * npc * rxN * uN[o];
    const double uNij = uN[o] * npc + (npe - npc) * npe * rxNr * uN[e];
    const double vEijp1 = vE[n] * epn + (epc - epn) * epc * ryE * vE[o];
    const double vEij = vE[o] * epc + (epn - epc) * epn * ryEr * vE[n];
    const double dv = dyU * (uNip1j - uNij) + uU * ddyN + dxU * (vEijp1 - vEij) + vU * ddxE;
    const double tn = dyU * (uNip1j - uNij) - uU * ddyN - dxU * (vEijp1 - vEij) + vU * ddxE;
    const double uEijp1 = uE[n] * epn + (epc - epn) * epc * ryE * uE[o];
    const double uEij = uE[o] * epc + (epn - epc) * epn * ryEr * uE[n];
    const double vNip1j = vN[e] * npe + (npc - npe) * npc * rxN * vN[o];
    const double vNij = vN[o] * npc + (npe - npc) * npe * rxNr * vN[e];
    const double sh = dxU * (uEijp1 - uEij) - uU * ddxE + dyU * (vNip1j - vNij) - vU * ddyN;
    A.f[CF_SHEARU][o] = sh;
    A.f[CF_DELTAU][o] = sqrt(dv * dv + A.p.e_factor * (tn * tn + sh * sh));
    if (m & 16u) push(A, o, m, CF_SHEARU, sh);
}

// ---- phase 1: stressC_T on ilo..ihi+1 x jlo..jhi+1 (the reference's T list, ice_dyn_shared.F90:729-738).
// zetax2T, etax2T, stresspT, stressmT are exchanged right after (:988-990): interior cells store and push them, the
// extra row and column (ghost cells) only keep what is never exchanged, stress12T. ----
template <bool ALWAYS, bool GEO>
__global__ __launch_bounds__(TX *TY) void cg_stress_t(EvpCgrid A, int last)
{
    const auto G = AGeo<GEO>::make(A);
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y + 1 || c.j < c.q.z || c.j > c.q.w + 1) return;
    const size_t o = c.o, w = o - 1, s = o - A.nx, sw = s - 1;
    const unsigned m = A.mask[o];
    const bool own = c.i <= c.q.y && c.j <= c.q.w;
    const double *uE = A.f[CF_UE], *vN = A.f[CF_VN], *shU = A.f[CF_SHEARU];
    const auto dyE = G[CG_DYE], dxN = G[CG_DXN], uarea = G[CG_UAREA];
    const double dxT = G[CG_DXT][o], dyT = G[CG_DYT][o];
    const double divT = dyE[o] * uE[o] - dyE[w] * uE[w] + dxN[o] * vN[o] - dxN[s] * vN[s];
    const double tensionT = (dyT * dyT) * (uE[o] / dyE[o] - uE[w] / dyE[w]) - (dxT * dxT) * (vN[o] / dxN[o] - vN[s] / dxN[s]);
    const double uareaavgr = 1.0 / (uarea[o] + uarea[s] + uarea[sw] + uarea[w]);
    const double shearTsqr = (shU[o] * shU[o] * uarea[o] + shU[s] * shU[s] * uarea[s] + shU[sw] * shU[sw] * uarea[sw] +
                              shU[w] * shU[w] * uarea[w]) * uareaavgr;
    const double shearT = (shU[o] * uarea[o] + shU[s] * uarea[s] + shU[sw] * uarea[sw] + shU[w] * uarea[w]) * uareaavgr;
    const double DeltaT = sqrt(divT * divT + A.p.e_factor * (tensionT * tensionT + shearTsqr));
    double zetax2, etax2, rep_prs;
    visc_replpress(A.p, A.in[CI_STRENGTH][o], G[CG_DMINT][o], DeltaT, zetax2, etax2, rep_prs);
    const double relax = 1.0 - A.p.arlx1i * A.p.revp;
    const double s12 = (A.f[CF_S12T][o] * relax + A.p.arlx1i * 0.5 * etax2 * shearT) * A.p.denom1;
    const double sp = (A.f[CF_SP][o] * relax + A.p.arlx1i * (zetax2 * divT - rep_prs)) * A.p.denom1;
    const double sm = (A.f[CF_SM][o] * relax + A.p.arlx1i * etax2 * tensionT) * A.p.denom1;
    if (!(m & 1u)) return;                       // loads above are unconditional (in bounds), stores are not
    A.f[CF_S12T][o] = s12;
    if (!own) return;
    const bool zeta = ALWAYS || last;            // zetax2T: nothing in the loop reads it
    if (zeta) A.f[CF_ZETA][o] = zetax2;
    A.f[CF_ETA][o] = etax2;
    A.f[CF_SP][o] = sp;
    A.f[CF_SM][o] = sm;
    if (m & 16u) {
        if (zeta) push(A, o, m, CF_ZETA, zetax2);
        push(A, o, m, CF_ETA, etax2);
        push(A, o, m, CF_SP, sp);
        push(A, o, m, CF_SM, sm);
    }
}

// T -> U average, grid_average_X2YS('NE', work, tarea, hm): ice_grid.F90:4190-4209
template <class GT>
__device__ __forceinline__ double avg_t2u_g(const EvpCgrid &A, const GT &G, const double *w1, size_t o)
{
    const auto hm = G[CG_HM], ta = G[CG_TAREA];
    const size_t e = o + 1, n = o + A.nx, ne = n + 1;
    const double wtmp = (hm[o] * ta[o] + hm[e] * ta[e] + hm[n] * ta[n] + hm[ne] * ta[ne]);
    if (wtmp == 0.0) return 0.0;
    return (hm[o] * w1[o] * ta[o] + hm[e] * w1[e] * ta[e] + hm[n] * w1[n] * ta[n] + hm[ne] * w1[ne] * ta[ne]) / wtmp;
}
__device__ __forceinline__ double avg_t2u(const EvpCgrid &A, const double *w1, size_t o) { return avg_t2u_g(A, PtrTab{A.g}, w1, o); }

// ---- phase 2: viscosity at the corners (:992-996) and stressC_U; stress12U is exchanged (:1011-1013) ----
__global__ __launch_bounds__(TX *TY) void cg_stress_u(EvpCgrid A)
{
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const size_t o = c.o;
    const unsigned m = A.mask[o];
    double etax2U;
    if (A.avg_strength) {
        if (!(m & 2u)) return;
        double z, r;
        visc_replpress(A.p, A.strengthU[o], A.deltaminEVP * A.g[CG_UAREA][o], A.f[CF_DELTAU][o], z, etax2U, r);
    } else {
        etax2U = avg_t2u(A, A.f[CF_ETA], o);
        A.f[CF_ETAU][o] = etax2U;                       // every interior cell, as grid_average_X2YS does
        if (!(m & 2u)) return;
    }
    const double relax = 1.0 - A.p.arlx1i * A.p.revp;
    const double s12 = (A.f[CF_S12U][o] * relax + A.p.arlx1i * 0.5 * etax2U * A.f[CF_SHEARU][o]) * A.p.denom1;
    A.f[CF_S12U][o] = s12;
    if (m & 16u) push(A, o, m, CF_S12U, s12);
}

// ---- phase 3: div_stress_Ex + stepu_C at E, div_stress_Ny + stepv_C at N; uvelE, vvelN are exchanged (:1063-1068) ----
__global__ __launch_bounds__(TX *TY) void cg_step(EvpCgrid A)
{
    constexpr bool FAST = false;      // (the shortcuts live in the fused schedule's cg_stress_u_step<true>)
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const size_t o = c.o, e = o + 1, n = o + A.nx, s = o - A.nx, w = o - 1;
    const unsigned m = A.mask[o];
    if (!(m & 12u)) return;
    const double *sp = A.f[CF_SP], *sm = A.f[CF_SM], *s12 = A.f[CF_S12U];
    const double spc = sp[o], smc = sm[o], s12c = s12[o];
    const EvpScalars &p = A.p;
    if (m & 4u) {
        const double *dyT = A.g[CG_DYT], *dxU = A.g[CG_DXU];
        const double dyE = A.g[CG_DYE][o], dxE = A.g[CG_DXE][o];
        const double strintx = A.in[CI_RHEOE][o] * A.g[CG_EAREAR][o] *
                               (0.5 * dyE * (sp[e] - spc) + (0.5 / dyE) * ((dyT[e] * dyT[e]) * sm[e] - (dyT[o] * dyT[o]) * smc) +
                                (1.0 / dxE) * ((dxU[o] * dxU[o]) * s12c - (dxU[s] * dxU[s]) * s12[s]));
        const double uold = A.f[CF_UE][o], vold = A.f[CF_VE][o];
        const double uocn = A.in[CI_UOCNE][o];
        const double du = uocn - uold, dv = A.in[CI_VOCNE][o] - vold;
        const double vrel = (FAST ? A.facE[o] : A.in[CI_AIE][o] * p.rhow * A.in[CI_CWE][o]) * sqrt(du * du + dv * dv);
        const double taux = vrel * (FAST ? uocn : A.in[CI_WATERXE][o]);
        double Cb = 0.0;
        if (!FAST) {
            const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
            Cb = A.in[CI_TBE][o] / ccc;
        }
        const double massdti = A.in[CI_EMASSDTI][o], fm = A.in[CI_FME][o];
        const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
        const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
        const double cc1 = strintx + A.in[CI_FORCEXE][o] + taux + massdti * (p.brlx * uold + p.revp * A.in[CI_UE_INIT][o]);
        const double unew = (ccb * vold + cc1) / cca;
        A.f[CF_STRX][o] = strintx;
        A.f[CF_UE][o] = unew;
        A.f[CF_TAUBX][o] = -unew * Cb;
        if (m & 16u) push(A, o, m, CF_UE, unew);
    }
    if (m & 8u) {
        const double *dxT = A.g[CG_DXT], *dyU = A.g[CG_DYU];
        const double dxN = A.g[CG_DXN][o], dyN = A.g[CG_DYN][o];
        const double strinty = A.in[CI_RHEON][o] * A.g[CG_NAREAR][o] *
                               (0.5 * dxN * (sp[n] - spc) - (0.5 / dxN) * ((dxT[n] * dxT[n]) * sm[n] - (dxT[o] * dxT[o]) * smc) +
                                (1.0 / dyN) * ((dyU[o] * dyU[o]) * s12c - (dyU[w] * dyU[w]) * s12[w]));
        const double uold = A.f[CF_UN][o], vold = A.f[CF_VN][o];
        const double du = A.in[CI_UOCNN][o] - uold, dv = A.in[CI_VOCNN][o] - vold;
        const double vrel = A.in[CI_AIN][o] * p.rhow * A.in[CI_CWN][o] * sqrt(du * du + dv * dv);
        const double tauy = vrel * A.in[CI_WATERYN][o];
        const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
        const double Cb = A.in[CI_TBN][o] / ccc;
        const double massdti = A.in[CI_NMASSDTI][o], fm = A.in[CI_FMN][o];
        const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
        const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
        const double cc2 = strinty + A.in[CI_FORCEYN][o] + tauy + massdti * (p.brlx * vold + p.revp * A.in[CI_VN_INIT][o]);
        const double vnew = (-ccb * uold + cc2) / cca;
        A.f[CF_STRY][o] = strinty;
        A.f[CF_VN][o] = vnew;
        A.f[CF_TAUBY][o] = -vnew * Cb;
        if (m & 16u) push(A, o, m, CF_VN, vnew);
    }
}

// stress12U after this subcycle at corner p (own cell or a neighbour, possibly a ghost cell): stressC_U with the
// T -> U average of etax2T, from the previous subcycle's value in A.s12_in; unchanged where there is no ice
template <class GT>
__device__ __forceinline__ double s12u_new(const EvpCgrid &A, const GT &G, size_t p, bool ice, double relax, double *etaU)
{
    const double old = A.s12_in[p];
    const double e2 = avg_t2u_g(A, G, A.f[CF_ETA], p);
    if (etaU) *etaU = e2;
    const double upd = (old * relax + A.p.arlx1i * 0.5 * e2 * A.f[CF_SHEARU][p]) * A.p.denom1;
    return ice ? upd : old;
}

// ---- fused C: phases 2 and 3 in one launch (visc_method = avg_zeta) ----
// FAST: the operands the reference's default configuration makes redundant are not read -- waterxE == uocnE and
// wateryN == vocnN bit for bit (cosw = 1, sinw = 0), TbE = TbN = +0 (no seabed stress), rheofactE = rheofactN = 1 --
// established per call on every ice cell (cg_call_setup); aiX*rhow*Cw comes premultiplied (same operation order).
// Bit-neutral: x*1.0, x + (+0.0) and 0.0/c are exact, taub = -u*(+0.0) is still formed.
template <bool FAST, bool GEO>
__global__ __launch_bounds__(TX *TY * 2) void cg_stress_u_step(EvpCgrid A, int last)
{
    const auto G = AGeo<GEO>::make(A);
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const size_t o = c.o, e = o + 1, n = o + A.nx, s = o - A.nx, w = o - 1;
    const unsigned m = A.mask[o];
    const double relax = 1.0 - A.p.arlx1i * A.p.revp;
    // blockDim.z == 2: the E face and the N face of a cell in different waves (half the dependent chain per wave, twice
    // the waves in flight: grids of gx1's size are latency-, not bandwidth-bound); blockDim.z == 1: one thread does both
    const bool doE = blockDim.z == 1 || threadIdx.z == 0, doN = blockDim.z == 1 || threadIdx.z == 1;
    double etaU;
    const double s12c = s12u_new(A, G, o, (m & 2u) != 0, relax, &etaU);
    const double s12s = doE ? s12u_new(A, G, s, (A.mask[s] & 32u) != 0, relax, nullptr) : 0.0;
    const double s12w = doN ? s12u_new(A, G, w, (A.mask[w] & 32u) != 0, relax, nullptr) : 0.0;
    const double *sp = A.f[CF_SP], *sm = A.f[CF_SM];
    const double spc = sp[o], smc = sm[o];
    const EvpScalars &p = A.p;
    // both faces computed for every interior cell (all loads in bounds and issued together); stores by mask
    double unew = 0.0, vnew = 0.0, strintx = 0.0, strinty = 0.0, taubx = 0.0, tauby = 0.0;
    if (doE) {
        const auto dyT = G[CG_DYT], dxU = G[CG_DXU];
        const double dyE = G[CG_DYE][o], dxE = G[CG_DXE][o];
        strintx = (FAST ? G[CG_EAREAR][o] : A.in[CI_RHEOE][o] * G[CG_EAREAR][o]) *
                  (0.5 * dyE * (sp[e] - spc) + (0.5 / dyE) * ((dyT[e] * dyT[e]) * sm[e] - (dyT[o] * dyT[o]) * smc) +
                   (1.0 / dxE) * ((dxU[o] * dxU[o]) * s12c - (dxU[s] * dxU[s]) * s12s));
        const double uold = A.f[CF_UE][o], vold = A.f[CF_VE][o];
        const double uocn = A.in[CI_UOCNE][o];
        const double du = uocn - uold, dv = A.in[CI_VOCNE][o] - vold;
        const double vrel = (FAST ? A.facE[o] : A.in[CI_AIE][o] * p.rhow * A.in[CI_CWE][o]) * sqrt(du * du + dv * dv);
        const double taux = vrel * (FAST ? uocn : A.in[CI_WATERXE][o]);
        double Cb = 0.0;
        if (!FAST) {
            const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
            Cb = A.in[CI_TBE][o] / ccc;
        }
        const double massdti = A.in[CI_EMASSDTI][o], fm = A.in[CI_FME][o];
        const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
        const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
        const double cc1 = strintx + A.in[CI_FORCEXE][o] + taux + massdti * (p.brlx * uold + p.revp * A.in[CI_UE_INIT][o]);
        unew = (ccb * vold + cc1) / cca;
        taubx = -unew * Cb;
    }
    if (doN) {
        const auto dxT = G[CG_DXT], dyU = G[CG_DYU];
        const double dxN = G[CG_DXN][o], dyN = G[CG_DYN][o];
        strinty = (FAST ? G[CG_NAREAR][o] : A.in[CI_RHEON][o] * G[CG_NAREAR][o]) *
                  (0.5 * dxN * (sp[n] - spc) - (0.5 / dxN) * ((dxT[n] * dxT[n]) * sm[n] - (dxT[o] * dxT[o]) * smc) +
                   (1.0 / dyN) * ((dyU[o] * dyU[o]) * s12c - (dyU[w] * dyU[w]) * s12w));
        const double uold = A.f[CF_UN][o], vold = A.f[CF_VN][o];
        const double vocn = A.in[CI_VOCNN][o];
        const double du = A.in[CI_UOCNN][o] - uold, dv = vocn - vold;
        const double vrel = (FAST ? A.facN[o] : A.in[CI_AIN][o] * p.rhow * A.in[CI_CWN][o]) * sqrt(du * du + dv * dv);
        const double tauy = vrel * (FAST ? vocn : A.in[CI_WATERYN][o]);
        double Cb = 0.0;
        if (!FAST) {
            const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
            Cb = A.in[CI_TBN][o] / ccc;
        }
        const double massdti = A.in[CI_NMASSDTI][o], fm = A.in[CI_FMN][o];
        const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
        const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
        const double cc2 = strinty + A.in[CI_FORCEYN][o] + tauy + massdti * (p.brlx * vold + p.revp * A.in[CI_VN_INIT][o]);
        vnew = (-ccb * uold + cc2) / cca;
        tauby = -vnew * Cb;
    }
    if (doE) {
        if (m & 2u) {
            A.f[CF_S12U][o] = s12c;
            if (m & 16u) push(A, o, m, CF_S12U, s12c);
        }
        if (last) A.f[CF_ETAU][o] = etaU;
    }
    if (doE && (m & 4u)) {
        A.f[CF_UE][o] = unew;
        if (last) {
            A.f[CF_STRX][o] = strintx;
            A.f[CF_TAUBX][o] = taubx;
        }
        if (m & 16u) push(A, o, m, CF_UE, unew);
    }
    if (doN && (m & 8u)) {
        A.f[CF_VN][o] = vnew;
        if (last) {
            A.f[CF_STRY][o] = strinty;
            A.f[CF_TAUBY][o] = tauby;
        }
        if (m & 16u) push(A, o, m, CF_VN, vnew);
    }
}

// ---- once per call: the leading factor of vrel, and whether the default-configuration shortcuts of cg_stress_u_step
// <true> hold bit for bit on every ice cell ----
__global__ __launch_bounds__(TX *TY) void cg_call_setup(EvpCgrid A, double *facE, double *facN, unsigned *flags)
{
    const Cell c = cell(A);
    if (!c.in) return;
    const size_t o = c.o;
    facE[o] = A.in[CI_AIE][o] * A.p.rhow * A.in[CI_CWE][o];
    facN[o] = A.in[CI_AIN][o] * A.p.rhow * A.in[CI_CWN][o];
    if (c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const unsigned m = A.mask[o];
    unsigned bad = 0;
    auto bits = [](double x) { return (unsigned long long)__double_as_longlong(x); };
    if (m & 4u) {
        if (bits(A.in[CI_WATERXE][o]) != bits(A.in[CI_UOCNE][o])) bad |= 1u;
        if (bits(A.in[CI_TBE][o]) != 0ull) bad |= 2u;
        if (A.in[CI_RHEOE][o] != 1.0) bad |= 4u;
    }
    if (m & 8u) {
        if (bits(A.in[CI_WATERYN][o]) != bits(A.in[CI_VOCNN][o])) bad |= 1u;
        if (bits(A.in[CI_TBN][o]) != 0ull) bad |= 2u;
        if (A.in[CI_RHEON][o] != 1.0) bad |= 4u;
    }
    if (bad) atomicOr(flags, bad);
}

// ---- copy a field into the ghost images of its interior cells (what one ice_HaloUpdate does) ----
__global__ __launch_bounds__(TX *TY) void cg_fill_images(EvpCgrid A, int field)
{
    const Cell c = cell(A);
    if (!c.in) return;
    if (A.mask[c.o] & 16u) push(A, c.o, 16u, field, A.f[field][c.o]);
}

// ---- phase 4: the other component at each face and the corner velocities (:1070-1094):
// uvelN = E2N('NW', earea) * npm, vvelE = N2E('SE', narea) * epm, uvel = E2U('N', earea) * uvm,
// vvel = N2U('E', narea) * uvm (grid_average_X2YA, ice_grid.F90:4388-4606); all four are exchanged ----
__global__ __launch_bounds__(TX *TY) void cg_average(EvpCgrid A)
{
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const size_t o = c.o, e = o + 1, n = o + A.nx, s = o - A.nx, w = o - 1;
    const unsigned m = A.mask[o];
    const double *uE = A.f[CF_UE], *vN = A.f[CF_VN], *ea = A.g[CG_EAREA], *na = A.g[CG_NAREA];
    const double eo = ea[o], no = na[o], uo = uE[o], vo = vN[o];
    double uN = 0.0, vE = 0.0, uU = 0.0, vU = 0.0, wtmp;
    wtmp = (ea[w] + eo + ea[n - 1] + ea[n]);
    if (wtmp != 0.0) uN = (uE[w] * ea[w] + uo * eo + uE[n - 1] * ea[n - 1] + uE[n] * ea[n]) / wtmp;
    wtmp = (na[s] + na[s + 1] + no + na[e]);
    if (wtmp != 0.0) vE = (vN[s] * na[s] + vN[s + 1] * na[s + 1] + vo * no + vN[e] * na[e]) / wtmp;
    wtmp = (eo + ea[n]);
    if (wtmp != 0.0) uU = (uo * eo + uE[n] * ea[n]) / wtmp;
    wtmp = (no + na[e]);
    if (wtmp != 0.0) vU = (vo * no + vN[e] * na[e]) / wtmp;
    const double uvm = A.g[CG_UVM][o];
    uN = uN * A.g[CG_NPM][o];
    vE = vE * A.g[CG_EPM][o];
    uU = uU * uvm;
    vU = vU * uvm;
    A.f[CF_UN][o] = uN;
    A.f[CF_VE][o] = vE;
    A.f[CF_UU][o] = uU;
    A.f[CF_VU][o] = vU;
    if (m & 16u) {
        push(A, o, m, CF_UN, uN);
        push(A, o, m, CF_VE, vE);
        push(A, o, m, CF_UU, uU);
        push(A, o, m, CF_VU, vU);
    }
}

// ---- once per call: strengthU = T2U('S')(strength) for visc_method = 'avg_strength' (:993) ----
__global__ __launch_bounds__(TX *TY) void cg_strength_u(EvpCgrid A, double *out)
{
    const Cell c = cell(A);
    if (!c.in) return;
    const bool interior = c.i >= c.q.x && c.i <= c.q.y && c.j >= c.q.z && c.j <= c.q.w;
    out[c.o] = interior ? avg_t2u(A, A.in[CI_STRENGTH], c.o) : 0.0;
}

// ---- once per call with ndte >= 1: grid_average_X2YA zero-fills its whole output before the interior is
// computed (ice_grid.F90:4412), so whatever uvelN, vvelE, uvel, vvel held outside the interior (ghost cells without
// a source, padding of short blocks) is zero from the first subcycle on; ghost cells with a source are pushed ----
__global__ __launch_bounds__(TX *TY) void cg_zero_outside(EvpCgrid A)
{
    const Cell c = cell(A);
    if (!c.in) return;
    if (c.i >= c.q.x && c.i <= c.q.y && c.j >= c.q.z && c.j <= c.q.w) return;
    A.f[CF_UN][c.o] = 0.0;
    A.f[CF_VE][c.o] = 0.0;
    A.f[CF_UU][c.o] = 0.0;
    A.f[CF_VU][c.o] = 0.0;
}

// ---- tripole fold, pass 1 (values from raw sources) and pass 2 (stores); see EvpCgFold ----
// ---- deformationsC_T (ice_dyn_shared.F90:1968-2074; strain_rates_Tdtsd :2171-2243), which evp() calls right after
// the loop (ice_dyn_evp.F90:1106-1119): on the T-cells of dyn_prep2's list, from the loop's final face velocities and
// shearU; every other cell keeps what the array holds ----
__global__ __launch_bounds__(TX *TY) void cg_deformations_t(EvpCgrid A, const double *__restrict__ tarear, double *__restrict__ divu,
                                                            double *__restrict__ shear, double *__restrict__ vort,
                                                            double *__restrict__ rdg_conv, double *__restrict__ rdg_shear)
{
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y + 1 || c.j < c.q.z || c.j > c.q.w + 1) return;
    const size_t o = c.o, w = o - 1, s = o - A.nx, sw = s - 1;
    if (!(A.mask[o] & 1u)) return;
    const double *uE = A.f[CF_UE], *vE = A.f[CF_VE], *uN = A.f[CF_UN], *vN = A.f[CF_VN], *shU = A.f[CF_SHEARU];
    const double *dyE = A.g[CG_DYE], *dxN = A.g[CG_DXN], *uarea = A.g[CG_UAREA];
    const double dxT = A.g[CG_DXT][o], dyT = A.g[CG_DYT][o];
    const double divT = dyE[o] * uE[o] - dyE[w] * uE[w] + dxN[o] * vN[o] - dxN[s] * vN[s];
    const double tensionT = (dyT * dyT) * (uE[o] / dyE[o] - uE[w] / dyE[w]) - (dxT * dxT) * (vN[o] / dxN[o] - vN[s] / dxN[s]);
    const double shearT = (dxT * dxT) * (uN[o] / dxN[o] - uN[s] / dxN[s]) + (dyT * dyT) * (vE[o] / dyE[o] - vE[w] / dyE[w]);
    const double shearTsqr = (shU[o] * shU[o] * uarea[o] + shU[s] * shU[s] * uarea[s] + shU[sw] * shU[sw] * uarea[sw] +
                              shU[w] * shU[w] * uarea[w]) / (uarea[o] + uarea[s] + uarea[sw] + uarea[w]);
    const double DeltaT = sqrt(divT * divT + A.p.e_factor * (tensionT * tensionT + shearTsqr));
    const double tr = tarear[o];
    const double dv = divT * tr;
    divu[o] = dv;
    const double tmp = DeltaT * tr;
    rdg_conv[o] = -fmin(dv, 0.0);
    rdg_shear[o] = 0.5 * (tmp - fabs(dv));
    shear[o] = tr * sqrt(tensionT * tensionT + shearT * shearT);
    vort[o] = tr * ((dyE[o] * vE[o] - dyE[w] * vE[w]) - (dxN[o] * uN[o] - dxN[s] * uN[s]));
}

// ---- dyn_finish at N and E points (ice_dyn_shared.F90:1291-1365; call sites ice_dyn_evp.F90:1408-1436): the ice-ocean stress
// from the loop's final face velocities, on the cells of dyn_prep2's N / E lists; every other cell keeps what the array holds.
// which: 0 = N points (uvelN, vvelN, cdn_ocnN, aiN, uocnN, vocnN, fmN), 1 = E points ----
__global__ __launch_bounds__(TX *TY) void cg_dyn_finish(EvpCgrid A, int which, double *__restrict__ strocnx, double *__restrict__ strocny)
{
    const Cell c = cell(A);
    if (!c.in || c.i < c.q.x || c.i > c.q.y || c.j < c.q.z || c.j > c.q.w) return;
    const size_t o = c.o;
    if (!(A.mask[o] & (which ? 4u : 8u))) return;
    const double Cw = A.in[which ? CI_CWE : CI_CWN][o], aiX = A.in[which ? CI_AIE : CI_AIN][o];
    const double uocn = A.in[which ? CI_UOCNE : CI_UOCNN][o], vocn = A.in[which ? CI_VOCNE : CI_VOCNN][o];
    const double fm = A.in[which ? CI_FME : CI_FMN][o];
    const double u = A.f[which ? CF_UE : CF_UN][o], v = A.f[which ? CF_VE : CF_VN][o];
    const double du = uocn - u, dv = vocn - v;
    double vrel = A.p.rhow * Cw * sqrt(du * du + dv * dv);
    vrel = vrel * aiX;
    strocnx[o] = vrel * ((uocn - u) * A.p.cosw - (vocn - v) * A.p.sinw * copysign(1.0, fm));
    strocny[o] = vrel * ((vocn - v) * A.p.cosw + (uocn - u) * A.p.sinw * copysign(1.0, fm));
}

__global__ void cg_fold_gather(EvpCgFold F)
{
    const int q = blockIdx.y;
    const EvpCgFoldList &L = F.L[F.loc[q]];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L.n) return;
    const double *x = F.x[q];
    const int a = L.a[k], b = L.b[k];
    const double isign = F.isign[q];
    const double s = L.flip[k] ? isign : 1.0;
    const double xa = a >= 0 ? x[a] : 0.0;
    double v;
    if (b >= 0 || b == -2) v = s * (0.5 * (xa + isign * (b >= 0 ? x[b] : 0.0)));     // -2: partner's block eliminated (land)
    else v = s * xa;
    F.tmp[(size_t)q * F.maxn + k] = v;
}
__global__ void cg_fold_scatter(EvpCgFold F)
{
    const int q = blockIdx.y;
    const EvpCgFoldList &L = F.L[F.loc[q]];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L.n) return;
    F.x[q][L.dst[k]] = F.tmp[(size_t)q * F.maxn + k];
}

// ---- the mask byte from the caller's four logical arrays (as 32-bit words, non-zero = true): pass 0 composes bits 0-4
// and bit5 of interior iceU cells, pass 1 hands bit5 on to the ghost cells that mirror them (one writer per ghost) ----
__global__ __launch_bounds__(TX *TY) void cg_mask_compose(EvpCgrid A, const int *m4, uint8_t *mask, int pass)
{
    const Cell c = cell(A);
    if (!c.in) return;
    const size_t o = c.o, n = (size_t)A.plane * A.nblocks;
    const bool interior = c.i >= c.q.x && c.i <= c.q.y && c.j >= c.q.z && c.j <= c.q.w;
    if (pass == 0) {
        const unsigned u = m4[n + o] ? 2u : 0u;
        mask[o] = (uint8_t)((m4[o] ? 1u : 0u) | u | (m4[2 * n + o] ? 4u : 0u) | (m4[3 * n + o] ? 8u : 0u) |
                            (A.img_slot[o] >= 0 ? 16u : 0u) | ((interior && u) ? 32u : 0u));
    } else if (interior && (mask[o] & 2u)) {
        const int s = A.img_slot[o];
        if (s < 0) return;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int d = A.img_dst[3 * s + k];
            if (d >= 0) mask[d] |= 32u;
        }
    }
}

// the same in ONE launch when the lists are short enough for one workgroup to order "all reads, then all writes" with a
// barrier (every thread keeps to its own entries in both passes; tmp holds its values in between)
__global__ __launch_bounds__(1024) void cg_fold_one(EvpCgFold F)
{
    for (int q = 0; q < F.nfields; ++q) {
        const EvpCgFoldList &L = F.L[F.loc[q]];
        const double *x = F.x[q];
        const double isign = F.isign[q];
        for (int k = threadIdx.x; k < L.n; k += 1024) {
            const int a = L.a[k], b = L.b[k];
            const double s = L.flip[k] ? isign : 1.0;
            const double xa = a >= 0 ? x[a] : 0.0;
            double v;
            if (b >= 0 || b == -2) v = s * (0.5 * (xa + isign * (b >= 0 ? x[b] : 0.0)));
            else v = s * xa;
            F.tmp[(size_t)q * F.maxn + k] = v;
        }
    }
    __syncthreads();
    for (int q = 0; q < F.nfields; ++q) {
        const EvpCgFoldList &L = F.L[F.loc[q]];
        for (int k = threadIdx.x; k < L.n; k += 1024) F.x[q][L.dst[k]] = F.tmp[(size_t)q * F.maxn + k];
    }
}

// ... and with lists of at most PER x 1024 entries every thread keeps its entries in registers between the two passes: two
// dependent memory round trips (indices, then operands) before the barrier instead of four (the tmp store and its reload),
// on a launch whose whole duration is latency (tx1: five of these per subcycle)
template <int PER>
__global__ __launch_bounds__(1024) void cg_fold_reg(EvpCgFold F)
{
    double v[4 * PER];
    int dd[4 * PER];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool on = q < F.nfields;                       // (uniform)
        const EvpCgFoldList L = F.L[on ? F.loc[q] : 0];
        const double *x = F.x[on ? q : 0];
        const double isign = F.isign[on ? q : 0];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = u * 1024 + (int)threadIdx.x, id = q * PER + u;
            dd[id] = -1;
            v[id] = 0.0;
            if (on && k < L.n) {
                const int a = L.a[k], b = L.b[k];
                const double s = L.flip[k] ? isign : 1.0;
                dd[id] = L.dst[k];
                const double xa = a >= 0 ? x[a] : 0.0;
                if (b >= 0 || b == -2) v[id] = s * (0.5 * (xa + isign * (b >= 0 ? x[b] : 0.0)));
                else v[id] = s * xa;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = q * PER + u;
            if (dd[id] >= 0) F.x[q][dd[id]] = v[id];
        }
}

// ---- ranks > 1: iceU of interior cells as 0/1 doubles (exchanged like a field), and back into bit5 of the mask for the
// ghost cells that mirror cells of other ranks ----
__global__ __launch_bounds__(TX *TY) void cg_umask_to_double(EvpCgrid A, double *d)
{
    const Cell c = cell(A);
    if (!c.in) return;
    const bool interior = c.i >= c.q.x && c.i <= c.q.y && c.j >= c.q.z && c.j <= c.q.w;
    d[c.o] = (interior && (A.mask[c.o] & 2u)) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(TX *TY) void cg_bit5_from_double(EvpCgrid A, const double *d, uint8_t *mask)
{
    const Cell c = cell(A);
    if (!c.in) return;
    if (d[c.o] != 0.0) mask[c.o] |= 32u;
}

// ---- ghost cells whose neighbour block was eliminated (land): ice_HaloUpdate fills them with zero at every
// exchange (ice_boundary.F90, fill value); nothing pushes into them, so they are zeroed once, in the first subcycle ----
__global__ void cg_zero_cells(EvpCgrid A, const int *cells, int n)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int c = cells[k];
    const int fld[9] = {CF_UE, CF_VE, CF_UN, CF_VN, CF_UU, CF_VU, CF_SP, CF_SM, CF_S12U};
#pragma unroll
    for (int q = 0; q < 9; ++q) A.f[fld[q]][c] = 0.0;
}


// =====================================================================
// One launch per subcycle (grids small enough to be launch- and latency-bound; one rank, no fold, avg_zeta).
//
// A workgroup of ONE_X x ONE_Y threads covers a window of the grid, one thread per position; the inner
// (ONE_X-3) x (ONE_Y-3) positions are the cells it owns.  Four levels, three workgroup barriers, nothing leaves the chip
// in between:
//   S  strain_rates_U at every position of the window (from the PREVIOUS subcycle's face velocities, read from
//      global memory around the cell itself, averages recomputed as in cg_avg_strain)              -> shearU in LDS
//   T  stressC_T at the positions with tx, ty >= 1 (shearU of the four corners from LDS)            -> etax2T,
//      stresspT, stressmT in LDS
//   U  viscosity at the corner + stressC_U at the positions with tx <= X-2, ty <= Y-2 (etax2T of the four T neighbours from
//      LDS)                                                                                        -> stress12U in LDS
//   C  div_stress + stepu_C / stepv_C on the owned cells (stress12U of the own, south and west corner from LDS).
//      (Until late round 4 level C evaluated stressC_U at its three corners itself -- each corner three times per window: a
//      fourth level costs a barrier and saves two T -> U averages or, visc_method = avg_strength, two visc_replpress per cell:
//      same box, libraries alternating: 3600 x 2400 avg_strength 876 -> 826 us, avg_zeta 784 -> 769, gx1 17.7 -> 17.3; ten
//      registers fewer.  Handing level T's west operands over between lanes instead of loading them: 2-4 % slower, taken out.)
// Positions outside the owned cells recompute what a neighbouring workgroup also computes, so no workgroup waits
// for another one; what a workgroup reads of its neighbours is the previous subcycle's state only, which is why
// uvelE, vvelN, stresspT, stressmT (and stress12U, as before) ping-pong between two buffers.
//
// A position is not an array cell: two steps beyond the owned cells of a block the array has ended, and one step
// beyond them sits a ghost cell whose value in the reference is the copy of another cell's.  The table (built once by
// the host from the halo plan) gives for every position the cell the reference's value COMES FROM -- the cell itself,
// the interior cell a ghost cell mirrors (any block of the rank), or, for a ghost cell nothing is copied into (closed
// boundary, eliminated neighbour block), the ghost cell itself marked "static": its arrays are read, never computed.
// Level S and T values are computed AT that cell (its own metrics, its own array neighbours, whose ghost cells hold
// pushed level-0 state), which is bit for bit what the owner stores and the exchange copies.
// Arrays nothing inside the loop reads any more (shearU, etax2T and the ones listed at the top) are stored in the last
// subcycle of a call only; stress12T of the extra row / column ihi+1, jhi+1 (ghost cells the reference also computes and
// no exchange overwrites) is kept up by the workgroup that owns the neighbouring interior cell, with the ghost cell's own
// metrics and history.
// =====================================================================
// The marched kernel's views (cg_strip): an array = a wave-uniform base + a 32-bit byte offset per lane, so that ONE register per
// lane (the cell's offset) addresses every array -- in a loop over rows the compiler otherwise keeps a 64-bit per-lane address for
// each of the ~50 arrays (250 registers, two waves per SIMD).  The tables (23 x n, 23 x n doubles) must end below 4 GB: the host checks.
struct P32 {
    const char *base;
    __device__ __forceinline__ double operator[](size_t i) const { return *(const double *)(base + (size_t)((unsigned)i * 8u)); }
};
struct P32W {
    char *base;
    __device__ __forceinline__ double &operator[](size_t i) const { return *(double *)(base + (size_t)((unsigned)i * 8u)); }
};
struct P32K {      // array k of a table: base + (k * stride + i) * 8, the sum in 32 bits
    const char *base;
    unsigned koff;
    __device__ __forceinline__ double operator[](size_t i) const { return *(const double *)(base + (size_t)((unsigned)i * 8u + koff)); }
};
struct Slab32 {
    static constexpr bool derived = false;
    const char *base;
    unsigned stride8;
    __device__ __forceinline__ P32K operator[](int k) const { return P32K{base, (unsigned)k * stride8}; }
};
struct DSlab32 {   // DSlab with 32-bit offsets
    static constexpr bool derived = true;
    const char *base;
    unsigned stride8;
    const uint8_t *gm;
    int nx;
    double dmin;
    struct Acc {
        const DSlab32 &S;
        int k;
        __device__ __forceinline__ double raw(int a, size_t p) const { return *(const double *)(S.base + (size_t)((unsigned)p * 8u + (unsigned)a * S.stride8)); }
        __device__ __forceinline__ unsigned bits(size_t p) const { return S.gm[(size_t)(unsigned)p]; }
        __device__ __forceinline__ double operator[](size_t p) const
        {
            switch (k) {
            case CG_TAREA: return raw(CG_DXT, p) * raw(CG_DYT, p);
            case CG_UAREA: return raw(CG_DXU, p) * raw(CG_DYU, p);
            case CG_NAREA: return raw(CG_DXN, p) * raw(CG_DYN, p);
            case CG_EAREA: return raw(CG_DXE, p) * raw(CG_DYE, p);
            case CG_EAREAR: { const double a = raw(CG_DXE, p) * raw(CG_DYE, p); return a > 0.0 ? 1.0 / a : 0.0; }
            case CG_NAREAR: { const double a = raw(CG_DXN, p) * raw(CG_DYN, p); return a > 0.0 ? 1.0 / a : 0.0; }
            case CG_DMINT: return S.dmin * (raw(CG_DXT, p) * raw(CG_DYT, p));
            case CG_RXN: return -(raw(CG_DXN, p + 1) / raw(CG_DXN, p));
            case CG_RXNR: return 1.0 / -(raw(CG_DXN, p + 1) / raw(CG_DXN, p));
            case CG_RYE: return -(raw(CG_DYE, p + S.nx) / raw(CG_DYE, p));
            case CG_RYER: return 1.0 / -(raw(CG_DYE, p + S.nx) / raw(CG_DYE, p));
            case CG_EPM: return (bits(p) & 1u) ? 1.0 : 0.0;
            case CG_NPM: return (bits(p) & 2u) ? 1.0 : 0.0;
            case CG_UVM: return (bits(p) & 4u) ? 1.0 : 0.0;
            case CG_HM: return (bits(p) & 8u) ? 1.0 : 0.0;
            default: return raw(k, p);
            }
        }
    };
    __device__ __forceinline__ Acc operator[](int k) const { return Acc{*this, k}; }
};
template <bool GEO> struct GeoView32;
template <> struct GeoView32<false> {
    static __device__ __forceinline__ Slab32 make(const EvpCgrid &, const EvpCgOne &T) { return Slab32{(const char *)T.gbase, (unsigned)T.stride * 8u}; }
};
template <> struct GeoView32<true> {
    static __device__ __forceinline__ DSlab32 make(const EvpCgrid &A, const EvpCgOne &T)
    {
        return DSlab32{(const char *)T.gbase, (unsigned)T.stride * 8u, T.gmask, A.nx, A.deltaminEVP};
    }
};
template <bool GEO> struct GeoView;
template <> struct GeoView<false> {
    static __device__ __forceinline__ Slab make(const EvpCgrid &, const EvpCgOne &T) { return Slab{T.gbase, T.stride}; }
};
template <> struct GeoView<true> {
    static __device__ __forceinline__ DSlab make(const EvpCgrid &A, const EvpCgOne &T)
    {
        return DSlab{T.gbase, T.stride, T.gmask, A.nx, A.deltaminEVP};
    }
};
struct TStress { double zetax2, etax2, sp, sm, shearT; };
// stressC_T at cell o (ice_dyn_evp.F90:1758-1860) with the four corner values of shearU handed in; spo, smo: previous
template <class GT, class IT, class P>
__device__ __forceinline__ TStress t_stress(const EvpCgrid &A, const GT &G, const IT &IN, const P &uE, const P &vN, size_t o,
                                            double shO, double shS, double shSW, double shW, double spo, double smo)
{
    const size_t w = o - 1, s = o - A.nx, sw = s - 1;
    const auto dyE = G[CG_DYE], dxN = G[CG_DXN], uarea = G[CG_UAREA];
    const double dxT = G[CG_DXT][o], dyT = G[CG_DYT][o];
    const double divT = dyE[o] * uE[o] - dyE[w] * uE[w] + dxN[o] * vN[o] - dxN[s] * vN[s];
    const double tensionT = (dyT * dyT) * (uE[o] / dyE[o] - uE[w] / dyE[w]) - (dxT * dxT) * (vN[o] / dxN[o] - vN[s] / dxN[s]);
    const double uareaavgr = 1.0 / (uarea[o] + uarea[s] + uarea[sw] + uarea[w]);
    const double shearTsqr = (shO * shO * uarea[o] + shS * shS * uarea[s] + shSW * shSW * uarea[sw] + shW * shW * uarea[w]) * uareaavgr;
    TStress r;
    r.shearT = (shO * uarea[o] + shS * uarea[s] + shSW * uarea[sw] + shW * uarea[w]) * uareaavgr;
    const double DeltaT = sqrt(divT * divT + A.p.e_factor * (tensionT * tensionT + shearTsqr));
    double rep_prs;
    visc_replpress(A.p, IN[CI_STRENGTH][o], G[CG_DMINT][o], DeltaT, r.zetax2, r.etax2, rep_prs);
    const double relax = 1.0 - A.p.arlx1i * A.p.revp;
    r.sp = (spo * relax + A.p.arlx1i * (r.zetax2 * divT - rep_prs)) * A.p.denom1;
    r.sm = (smo * relax + A.p.arlx1i * r.etax2 * tensionT) * A.p.denom1;
    return r;
}

// MODE 0: avg_zeta, not the last subcycle of a call (shearU alone at level S); 1: avg_zeta, last subcycle (deltaU is stored);
// 2: avg_strength (deltaU feeds the corner viscosities in every subcycle)
// GEO: the derived view of the static table (DSlab above)
// (The same 64 x 16 window on a workgroup of 512 threads, two positions per thread one after the other inside each level, so
// that TWO workgroups in different phases share a CU where the 1024-thread one is alone: 3600 x 2400 804 us against 798,
// avg_strength 952 against 874 -- the workgroup's phases are not what the CU waits for.  Taken out again.)
template <bool FAST, int ONE_X, int ONE_Y, int MODE, bool GEO>
__device__ __forceinline__ void cg_one_window(const EvpCgrid &A, const EvpCgOne &T, int last, int t, int tx, int ty)
{
    constexpr bool AVGS = MODE == 2;
    __shared__ double s_sh[ONE_Y][ONE_X], s_un[ONE_Y][ONE_X], s_ve[ONE_Y][ONE_X];
    __shared__ double s_eta[ONE_Y][ONE_X], s_sp[ONE_Y][ONE_X], s_sm[ONE_Y][ONE_X];
    __shared__ double s_dl[ONE_Y][ONE_X];            // deltaU (visc_method = avg_strength: the corner viscosities come from it)
    __shared__ double s_s12[ONE_Y][ONE_X];           // stress12U of this subcycle
    // workgroups go to the XCDs round-robin: XCD x gets the x-th contiguous run of the (space-ordered) window list
    // (A launch of only as many workgroups as are resident at once, each looping over several windows with the LDS arrays
    // doubled -- the 1024-thread workgroup of the 64 x 16 window is alone on its CU, and tools/cgrid_phases.py shows the CU
    // idle for a fifth of the time between two of them -- was built and measured: the loop keeps the ~70 array pointers
    // alive in scalar registers that spill into vector ones, 109 -> 140 registers, one wave per SIMD less; 3600 x 2400
    // 786 -> 937 us, gx1 16.9 -> 25.9.  Taken out.  Forcing MORE waves per SIMD with a register cap -- 80 registers for the 512-thread
    // shapes, 64 for this one -- spills 38 / 69 registers to scratch: 3600 x 2400 1432 / 1690 us.)
    if (t >= T.ntiles) return;
    const int4 tl = T.tiles[t];                          // block, first owned i, first owned j (1-based)
    const int4 q = A.blk[tl.x];
    const int i = tl.y - 2 + tx, j = tl.z - 2 + ty;      // the position in the block's own numbering (may lie outside its array)
    // (a window whose positions are all cells of its block's array, each its own source -- nearly all of them on a large
    // block -- is marked regular and skips the table: one dependent load less at the head of the kernel)
    const int lr = tl.w ? (int)((size_t)tl.x * A.plane + (size_t)(j - 1) * A.nx + (i - 1)) : T.tab[(size_t)t * (ONE_X * ONE_Y) + ty * ONE_X + tx];
    const bool stat = lr < 0;
    const size_t L = (size_t)(stat ? -1 - lr : lr);
    const unsigned m = A.mask[L];
    const bool own = tx >= 2 && tx <= ONE_X - 2 && ty >= 2 && ty <= ONE_Y - 2 && i <= q.y && j <= q.w;
    const double *uE = T.uE_in, *vN = T.vN_in;
    const Slab IN{T.inbase, T.stride};
    const auto G = GeoView<GEO>::make(A, T);
    // (test build) phase stamps of one wave with owned cells (thread (2, 2)): 0 start, 1 / 2 before / after the first barrier,
    // 3 / 4 the second, 5 level C's arithmetic done, 6 end; 7: XCC and CU the window ran on
    const bool stamp = T.prof && tx == 2 && ty == 2;
    auto mark = [&](int k) {
        if (stamp) T.prof[(size_t)t * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    mark(0);
    if (stamp) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        T.prof[(size_t)t * 8 + 7] = ((unsigned long long)(xcc & 15u) << 32) | hw;
    }
    const double relax = 1.0 - A.p.arlx1i * A.p.revp;
    const int nx = A.nx;

    // ---- S ----
    {
        double sh, delta = 0.0, uNo = 0.0, vEo = 0.0;
        if (stat) {
            sh = A.f[CF_SHEARU][L];
        } else {
            const size_t o = L, e = o + 1, n = o + nx;
            const auto ea = G[CG_EAREA], na = G[CG_NAREA], npm = G[CG_NPM], epm = G[CG_EPM];
            const double uvm = G[CG_UVM][o];
            if (MODE != 0) {                             // deltaU is wanted: the whole of strain_rates_U
                StrainIn v;
                v.uNo = uNo = avg_nw(uE, ea, o, nx) * npm[o];
                v.vEo = vEo = avg_se(vN, na, o, nx) * epm[o];
                v.uU = avg_2(uE, ea, o, n) * uvm;
                v.vU = avg_2(vN, na, o, e) * uvm;
                v.uNe = avg_nw(uE, ea, e, nx) * npm[e];
                v.vEn = avg_se(vN, na, n, nx) * epm[n];
                v.uEo = uE[o]; v.uEn = uE[n]; v.vNo = vN[o]; v.vNe = vN[e];
                strain_u(A, G, o, v, sh, delta);
            } else {                                     // shearU alone: two of the six averages (four more on owned cells, for level C)
                if (own) {
                    uNo = avg_nw(uE, ea, o, nx) * npm[o];
                    vEo = avg_se(vN, na, o, nx) * epm[o];
                }
                sh = shear_u(A, G, o, uE[o], uE[n], vN[o], vN[e], avg_2(uE, ea, o, n) * uvm, avg_2(vN, na, o, e) * uvm);
            }
            if (!(m & 2u)) sh = A.f[CF_SHEARU][o];       // strain_rates_U leaves cells without ice alone
            else if (own && last) {
                A.f[CF_SHEARU][o] = sh;
                A.f[CF_DELTAU][o] = delta;
                if (m & 16u) push(A, o, m, CF_SHEARU, sh);
            }
        }
        s_sh[ty][tx] = sh;
        s_un[ty][tx] = uNo;
        s_ve[ty][tx] = vEo;
        if (AVGS) s_dl[ty][tx] = delta;
    }
    mark(1);
    __syncthreads();
    mark(2);

    // ---- T ----
    if (tx >= 1 && ty >= 1) {
        const double shO = s_sh[ty][tx], shS = s_sh[ty - 1][tx], shSW = s_sh[ty - 1][tx - 1], shW = s_sh[ty][tx - 1];
        double eta, sp = T.sp_in[L], sm = T.sm_in[L];
        if (stat || !(m & 1u)) {
            eta = A.f[CF_ETA][L];
        } else {
            const TStress r = t_stress(A, G, IN, uE, vN, L, shO, shS, shSW, shW, sp, sm);
            eta = r.etax2; sp = r.sp; sm = r.sm;
            if (own) {
                A.f[CF_S12T][L] = (A.f[CF_S12T][L] * relax + A.p.arlx1i * 0.5 * r.etax2 * r.shearT) * A.p.denom1;
                A.f[CF_SP][L] = sp;
                A.f[CF_SM][L] = sm;
                if (last) {
                    A.f[CF_ZETA][L] = r.zetax2;
                    A.f[CF_ETA][L] = eta;
                }
                if (m & 16u) {
                    push(A, L, m, CF_SP, sp);
                    push(A, L, m, CF_SM, sm);
                    if (last) {
                        push(A, L, m, CF_ZETA, r.zetax2);
                        push(A, L, m, CF_ETA, eta);
                    }
                }
            }
        }
        s_eta[ty][tx] = eta;
        s_sp[ty][tx] = sp;
        s_sm[ty][tx] = sm;
        // the extra row / column of the reference's T list: ghost cells; only stress12T survives the exchange
        if ((i == q.y + 1 && j >= q.z && j <= q.w + 1) || (j == q.w + 1 && i >= q.x && i <= q.y)) {
            const int ii = min(i, q.y), jj = min(j, q.w);
            if (ii >= tl.y && ii <= tl.y + ONE_X - 4 && jj >= tl.z && jj <= tl.z + ONE_Y - 4) {
                const size_t g = (size_t)tl.x * A.plane + (size_t)(j - 1) * nx + (i - 1);
                if (A.mask[g] & 1u) {
                    const TStress r = t_stress(A, G, IN, uE, vN, g, shO, shS, shSW, shW, 0.0, 0.0);
                    A.f[CF_S12T][g] = (A.f[CF_S12T][g] * relax + A.p.arlx1i * 0.5 * r.etax2 * r.shearT) * A.p.denom1;
                }
            }
        }
    }
    mark(3);
    __syncthreads();
    mark(4);

    // ---- C ----
    // (The two bottom rows of positions are rim: their waves have nothing left to do here.  Letting them touch the lines of the
    // window that comes to this CU next -- one dword per 128-byte line of its 16 rows of all 29 arrays, so that the next
    // workgroup's first loads find them on their way -- was built and measured: level S did not get shorter (14.5k cycles
    // against 13.8k), level C more than twice as long: 3600 x 2400 1134 us against 799.  The loads are not waiting for a cold
    // miss, the memory system is busy; more requests make it worse.)
    // ---- U ---- stress12U (stressC_U, with the T -> U average of etax2T or the corner's own viscosity) ONCE per position that has
    // its four T neighbours in the window, AT the cell the position's value comes from -- what the owner stores and the
    // exchange copies; level C reads the three corners it needs (own, south, west) from LDS instead of evaluating each itself
    double etaU = 0.0;
    {
        double s12v = A.s12_in[L];
        if (!stat && tx <= ONE_X - 2 && ty <= ONE_Y - 2) {
            double e2;
            if (AVGS) {
                // visc_method = avg_strength: viscosity from the T -> U average of the strength and the corner's own Delta
                // (ice_dyn_evp.F90:992-996)
                double z, r;
                visc_replpress(A.p, A.strengthU[L], A.deltaminEVP * G[CG_UAREA][L], s_dl[ty][tx], z, e2, r);
            } else {
                // T -> U average of etax2T (avg_t2u) from the values in LDS
                const auto hm = G[CG_HM], ta = G[CG_TAREA];
                const size_t pp = L, pe = pp + 1, pn = pp + nx, pne = pn + 1;
                const double wtmp = (hm[pp] * ta[pp] + hm[pe] * ta[pe] + hm[pn] * ta[pn] + hm[pne] * ta[pne]);
                e2 = wtmp == 0.0 ? 0.0
                                 : (hm[pp] * s_eta[ty][tx] * ta[pp] + hm[pe] * s_eta[ty][tx + 1] * ta[pe] + hm[pn] * s_eta[ty + 1][tx] * ta[pn] +
                                    hm[pne] * s_eta[ty + 1][tx + 1] * ta[pne]) / wtmp;
            }
            etaU = e2;
            const double upd = (s12v * relax + A.p.arlx1i * 0.5 * e2 * s_sh[ty][tx]) * A.p.denom1;
            if (m & 2u) s12v = upd;
        }
        s_s12[ty][tx] = s12v;
    }
    __syncthreads();
    if (!own) return;
    {
        const size_t o = L, e = o + 1, n = o + nx, s = o - nx, w = o - 1;
        const double s12c = s_s12[ty][tx], s12s = s_s12[ty - 1][tx], s12w = s_s12[ty][tx - 1];
        const double spc = s_sp[ty][tx], smc = s_sm[ty][tx];
        const double spe = s_sp[ty][tx + 1], sme = s_sm[ty][tx + 1], spn = s_sp[ty + 1][tx], smn = s_sm[ty + 1][tx];
        const EvpScalars &p = A.p;
        double unew, vnew, strintx, strinty, taubx, tauby;
        {
            const auto dyT = G[CG_DYT], dxU = G[CG_DXU];
            const double dyE = G[CG_DYE][o], dxE = G[CG_DXE][o];
            strintx = (FAST ? G[CG_EAREAR][o] : IN[CI_RHEOE][o] * G[CG_EAREAR][o]) *
                      (0.5 * dyE * (spe - spc) + (0.5 / dyE) * ((dyT[e] * dyT[e]) * sme - (dyT[o] * dyT[o]) * smc) +
                       (1.0 / dxE) * ((dxU[o] * dxU[o]) * s12c - (dxU[s] * dxU[s]) * s12s));
            const double uold = uE[o], vold = s_ve[ty][tx];
            const double uocn = IN[CI_UOCNE][o];
            const double du = uocn - uold, dv = IN[CI_VOCNE][o] - vold;
            const double vrel = (FAST ? A.facE[o] : IN[CI_AIE][o] * p.rhow * IN[CI_CWE][o]) * sqrt(du * du + dv * dv);
            const double taux = vrel * (FAST ? uocn : IN[CI_WATERXE][o]);
            double Cb = 0.0;
            if (!FAST) {
                const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
                Cb = IN[CI_TBE][o] / ccc;
            }
            const double massdti = IN[CI_EMASSDTI][o], fm = IN[CI_FME][o];
            const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
            const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
            const double cc1 = strintx + IN[CI_FORCEXE][o] + taux + massdti * (p.brlx * uold + p.revp * IN[CI_UE_INIT][o]);
            unew = (ccb * vold + cc1) / cca;
            taubx = -unew * Cb;
        }
        {
            const auto dxT = G[CG_DXT], dyU = G[CG_DYU];
            const double dxN = G[CG_DXN][o], dyN = G[CG_DYN][o];
            strinty = (FAST ? G[CG_NAREAR][o] : IN[CI_RHEON][o] * G[CG_NAREAR][o]) *
                      (0.5 * dxN * (spn - spc) - (0.5 / dxN) * ((dxT[n] * dxT[n]) * smn - (dxT[o] * dxT[o]) * smc) +
                       (1.0 / dyN) * ((dyU[o] * dyU[o]) * s12c - (dyU[w] * dyU[w]) * s12w));
            const double uold = s_un[ty][tx], vold = vN[o];
            const double vocn = IN[CI_VOCNN][o];
            const double du = IN[CI_UOCNN][o] - uold, dv = vocn - vold;
            const double vrel = (FAST ? A.facN[o] : IN[CI_AIN][o] * p.rhow * IN[CI_CWN][o]) * sqrt(du * du + dv * dv);
            const double tauy = vrel * (FAST ? vocn : IN[CI_WATERYN][o]);
            double Cb = 0.0;
            if (!FAST) {
                const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
                Cb = IN[CI_TBN][o] / ccc;
            }
            const double massdti = IN[CI_NMASSDTI][o], fm = IN[CI_FMN][o];
            const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
            const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
            const double cc2 = strinty + IN[CI_FORCEYN][o] + tauy + massdti * (p.brlx * vold + p.revp * IN[CI_VN_INIT][o]);
            vnew = (-ccb * uold + cc2) / cca;
            tauby = -vnew * Cb;
        }
        if (m & 2u) {
            A.f[CF_S12U][o] = s12c;
            if (m & 16u) push(A, o, m, CF_S12U, s12c);
        }
        mark(5);
        if (last && !AVGS) A.f[CF_ETAU][o] = etaU;   // (avg_strength: the reference never stores etax2U)
        if (m & 4u) {
            A.f[CF_UE][o] = unew;
            if (last) {
                A.f[CF_STRX][o] = strintx;
                A.f[CF_TAUBX][o] = taubx;
            }
            if (m & 16u) push(A, o, m, CF_UE, unew);
        }
        if (m & 8u) {
            A.f[CF_VN][o] = vnew;
            if (last) {
                A.f[CF_STRY][o] = strinty;
                A.f[CF_TAUBY][o] = tauby;
            }
            if (m & 16u) push(A, o, m, CF_VN, vnew);
        }
        mark(6);
    }
}


template <bool FAST, int ONE_X, int ONE_Y, int MODE, bool GEO>
__global__ __launch_bounds__(ONE_X *ONE_Y) void cg_one(EvpCgrid A, EvpCgOne T, int last)
{
    // workgroups go to the XCDs round-robin: XCD x gets the x-th contiguous run of the (space-ordered) window list
    const int t = T.plain ? (int)blockIdx.x : (int)(blockIdx.x & 7u) * T.per_xcd + (int)(blockIdx.x >> 3);
    cg_one_window<FAST, ONE_X, ONE_Y, MODE, GEO>(A, T, last, t, (int)threadIdx.x, (int)threadIdx.y);
}

// =====================================================================
// cg_one's subcycle for the INTERIOR of a large block, marched (round 6).  cg_one's window is a workgroup of 64 x 16 positions that
// owns 61 x 13 cells, four levels behind three barriers, every operand fetched where it is used (some 130 loads per position, most
// of them of neighbours another lane also fetches, behind branches): on 3600 x 2400 it runs at 0.38 of the HBM peak, and what
// it waits for is instruction issue and exposed load latency, not bytes.  Here ONE WAVE takes a strip of 64
// positions (60 owned columns) and walks north over a segment of rows [ja, jb]; per iteration
//   S (row j) -> T (row j) -> U (row j-1) -> C (row j-1),
// every operand loaded ONCE per row by the lane that holds the cell, unconditionally and a whole iteration ahead of its use (33 loads
// per row), east / west neighbours by DPP lane shifts, south neighbours and the level hand-offs carried in registers; products and
// quotients that several neighbours use (velocity x area, hm x etax2T x tarea, uE / dyE ...) are formed once by their own lane and
// shifted -- the same operation on the same operands, so the same bits.  No LDS, no barrier, two waves per SIMD.
// Same arithmetic AT the same cells on the same inputs (the previous subcycle's buffers) as cg_one: which kernel owns a cell does
// not show in its bits.  The derived view of the static table only (else cg_one); template variants for the lengths formed in the
// kernel (LEN), the last subcycle of a call (LAST), the general momentum step (FAST = false) and visc_method = avg_strength (AVGS),
// described above the kernel.  Only "regular" positions (interior cells of the block, each its own source): the host hands this
// kernel the rectangle of the block that the regular windows of 32 x 8 cover (halo_plan.cpp: strip_zones / strip_items) and keeps
// the windows along the block's edges for cg_one -- which run as further workgroups of the same launch.
// Lanes: S on 0..62 (lane 63 only loads: the east neighbour's operands), T on 1..62, U on 1..61, C -- the owned cells -- on 2..61.
// =====================================================================
__device__ __forceinline__ double cg_lane_up(double v)    // lane l <- lane l-1 (lane 0: undefined, never used)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double cg_lane_dn(double v)    // lane l <- lane l+1 (lane 63: undefined, never used)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ unsigned cg_lane_dn_u(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}
// an array as a wave-uniform base + a 32-bit byte offset per lane: one register per lane addresses every array
__device__ __forceinline__ double cg_ld(const void *base, unsigned off) { return *(const double *)((const char *)base + (size_t)off); }
__device__ __forceinline__ unsigned cg_ldb(const uint8_t *base, unsigned cell) { return base[(size_t)cell]; }
__device__ __forceinline__ void cg_st(void *base, unsigned off, double v) { *(double *)((char *)base + (size_t)off) = v; }

// In the same launch, numbered after the marched kernel's workgroups: the windows cg_one keeps along the block's edges (E), as
// workgroups of 256 threads (32 x 8 positions).  They take the places the first marched workgroups to finish leave, instead of a
// launch of their own behind it (51 us on 3600 x 2400 for 2.6 % of the cells: every window a round trip of four dependent levels).
// (The marched part is written into the kernel itself: as an inlined device function it comes out 11 registers fatter and spills.)
// LEN: the six lengths dxT, dyT, dxU, dyU, dxE, dyN are formed from dxN (= HTN) and dyE (= HTE) by the reference's start-up
// formulas (ice_grid.F90:3063-3280: two- and four-point means; the host has verified them bit for bit on every cell the kernel
// touches) instead of loaded: 27 loads per row instead of 33, 48 B per cell less.  (Lane 0 then lacks a west neighbour: S on 1..62,
// T on 2..62, U on 2..61, owned cells on 3..61.)
// LAST: the last subcycle of a call also stores what the caller reads once per call -- shearU, zetax2T, etax2T, etax2U, strintxE / yN,
// taubxE / yN (values the subcycle forms anyway) and deltaU, for which strain_rates_U's divergence and tension are worked out too, ONE ROW
// LATE: they take the N-face average at the east neighbour (a lane shift of the owned rows' uvelN) and the E-face average of the row to
// the north (the next iteration's vvelE); nothing inside the loop reads deltaU with visc_method = avg_zeta.
// FAST (as in cg_one): the default configuration's short cuts hold on every ice cell of the call (waterx == uocn, Tb == 0, rheofact == 1);
// without it the momentum step is the general one: ten more operands per cell (rheofact, aice x cdn_ocn in place of their per-call
// product, waterx / watery, Tb at both faces), two more square roots and divisions.
// (the general momentum step does not fit 256 registers, and a wave that spills to scratch waits for its prefetched rows at every
// reload -- vmcnt counts in order: 983 us on 3600 x 2400, slower than cg_one; it gets a SIMD's register file to itself instead: 658 us
// against cg_one's 905.  The last subcycle's instantiation, 32 bytes of scratch, is no faster that way: 754 us against 717)
// AVGS: visc_method = avg_strength -- the corner's viscosity from the T -> U average of the strength (made once per call) and the
// corner's own Delta (ice_dyn_evp.F90:992-996): deltaU of row j-1, one row late as in LAST, arrives exactly when level U of that row runs.
template <bool LEN, bool LAST, bool FAST, bool AVGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((FAST && !AVGS) ? 2 : 1, (FAST && !AVGS) ? 2 : 1))) void cg_strip(EvpCgrid A, EvpCgOne T, EvpCgStrip Z, EvpCgOne E)
{
    if ((int)blockIdx.x >= 8 * Z.per_xcd) {
        cg_one_window<FAST, 32, 8, AVGS ? 2 : (LAST ? 1 : 0), true>(A, E, LAST ? 1 : 0, (int)blockIdx.x - 8 * Z.per_xcd, (int)(threadIdx.x & 31u), (int)(threadIdx.x >> 5));
        return;
    }
    const int lane = (int)(threadIdx.x & 63u);
    // workgroups go to the XCDs round-robin: XCD x takes the x-th contiguous run of the item list (x fastest, then segments)
    const int wg = (int)(blockIdx.x & 7u) * Z.per_xcd + (int)(blockIdx.x >> 3);
    const int it = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
    if (it >= Z.nitems) return;
    const int *w = Z.items + (size_t)it * 6;               // block, column of lane 2, first and last owned row (1-based), first and last owned lane
    const int blk = __builtin_amdgcn_readfirstlane(w[0]), col = __builtin_amdgcn_readfirstlane(w[1]);
    const int ja = __builtin_amdgcn_readfirstlane(w[2]), jb = __builtin_amdgcn_readfirstlane(w[3]);
    const int own_lo = __builtin_amdgcn_readfirstlane(w[4]), own_hi = __builtin_amdgcn_readfirstlane(w[5]);
    const unsigned nx = (unsigned)A.nx, nx8 = nx * 8u;
    const bool ownx = lane >= own_lo && lane <= own_hi;
    const unsigned st8 = (unsigned)T.stride * 8u;
    const void *gb = T.gbase, *ib = T.inbase;
    const uint8_t *gm = T.gmask;
    const EvpScalars &p = A.p;
    const double relax = 1.0 - p.arlx1i * p.revp;
    const double dmin = A.deltaminEVP;
    const bool revised = p.revp != 0.0;
    // cell of (lane, row j0): the row the first iteration calls j.  Three groups of loads run ahead of the arithmetic by
    // different distances: A (what level S reads of the row north of its own) two rows, B (the rest of S and T) one row, C (level
    // C's momentum operands, used one row behind) within the iteration.
    // (LEN: one iteration earlier -- dxE of row ja - 2, which level S starts with, takes HTN of row ja - 3)
    const int j0 = ja - (LEN ? 5 : 4);
    unsigned cell = (unsigned)((size_t)blk * A.plane + (size_t)(j0 - 1) * nx + (size_t)(col - 3 + lane));
    auto G = [&](int k, unsigned off) { return cg_ld(gb, off + (unsigned)k * st8); };
    auto I = [&](int k, unsigned off) { return cg_ld(ib, off + (unsigned)k * st8); };

    // ---- registers that travel with the rows: suffix N = row j+1, 0 = row j, 1 = row j-1, 2 = row j-2 ----
    double uEN = 0, dxEN = 0, dyEN = 0; unsigned gN = 0;                     // group A as it arrives
    double hN = 0, h0 = 0, h1 = 0, ehN = 0, wdyEN = 0;                       // LEN: HTN of rows j+1, j, j-1; lane-shifted HTN / HTE of row j+1
    double uE0 = 0, uE1 = 0, dxE0 = 0, dyE0 = 0, ea0 = 0, dxE1 = 0, dyE1 = 0, ea1 = 0, eaNc = 0, PNc = 0;
    unsigned g0 = 0;
    double vN1 = 0, dxN1 = 0, dyN1 = 0, na1 = 0, Q1 = 0, DV1 = 0, VQ1 = 0, ua1 = 0, XU1 = 0, XU2 = 0, YU1 = 0, XT1 = 0, YT1 = 0, W1 = 0;
    double sh1 = 0, SS1 = 0, SU1 = 0, un1 = 0, ve1 = 0, R1 = 0, sp1 = 0, sm1 = 0, s12_2 = 0;
    unsigned m1 = 0;
    double dxU1 = 0, dyU1 = 0, uU1 = 0, vU1 = 0; unsigned g1 = 0;           // LAST, AVGS: what deltaU of row j-1 still needs
    // loads in flight (issued one iteration, used the next)
    double L_uE = 0, L_dxE = 0, L_dyE = 0; unsigned L_g = 0;                 // A: row j+2 (LEN: L_dxE carries HTN)
    double L_vN = 0, L_dxN = 0, L_dyN = 0, L_dxU = 0, L_dyU = 0, L_dxT = 0, L_dyT = 0, L_str = 0, L_sp = 0, L_sm = 0, L_s12t = 0,
           L_eta = 0, L_shu = 0;                                             // B: row j+1
    unsigned L_m = 0, m_next = 0;                                            // ice masks: row j+2 in flight, row j+1 arrived

    for (int j = j0; j <= jb + 1; ++j, cell += nx) {
        // ---- what arrived: A = row j+1, B = row j, C = row j-1 (requested during the previous iteration) ----
        const double a_uE = L_uE, a_dxE = L_dxE, a_dyE = L_dyE; const unsigned a_g = L_g;
        const double b_vN = L_vN, b_dxN = L_dxN, b_dyN = L_dyN, b_dxU = L_dxU, b_dyU = L_dyU, b_dxT = L_dxT, b_dyT = L_dyT,
                     b_str = L_str, b_sp = L_sp, b_sm = L_sm, b_s12t = L_s12t, b_eta = L_eta, b_shu = L_shu;
        const unsigned m = m_next;                          // ice masks of row j
        m_next = L_m;
        // ---- everything the next iteration consumes, requested now ----
        {
            const unsigned cN = cell + nx, cNN = cN + nx;             // rows j+1, j+2
            const unsigned oN = cN * 8u, oNN = oN + nx8;
            L_uE = cg_ld(T.uE_in, oNN); L_dxE = G(LEN ? CG_DXN : CG_DXE, oNN); L_dyE = G(CG_DYE, oNN); L_g = cg_ldb(gm, cNN);
            L_m = cg_ldb(A.mask, cNN);
            L_vN = cg_ld(T.vN_in, oN); L_str = I(CI_STRENGTH, oN);
            if (!LEN) {
                L_dxN = G(CG_DXN, oN); L_dyN = G(CG_DYN, oN); L_dxU = G(CG_DXU, oN); L_dyU = G(CG_DYU, oN);
                L_dxT = G(CG_DXT, oN); L_dyT = G(CG_DYT, oN);
            }
            L_sp = cg_ld(T.sp_in, oN); L_sm = cg_ld(T.sm_in, oN); L_s12t = cg_ld(A.f[CF_S12T], oN);
            // what cells without ice keep (read by very few lanes: the others fetch the array's first line)
            L_eta = cg_ld(A.f[CF_ETA], (m_next & 1u) ? 0u : oN);
            L_shu = cg_ld(A.f[CF_SHEARU], (m_next & 2u) ? 0u : oN);
        }
        // group C (row j-1) is used at the bottom of THIS iteration: levels S, T and U cover its latency
        const unsigned o1 = (cell - nx) * 8u;
        const double c_s12 = cg_ld(A.s12_in, o1);
        const double c_uoE = I(CI_UOCNE, o1), c_voE = I(CI_VOCNE, o1), c_emE = I(CI_EMASSDTI, o1), c_fmE = I(CI_FME, o1), c_fxE = I(CI_FORCEXE, o1);
        const double c_voN = I(CI_VOCNN, o1), c_uoN = I(CI_UOCNN, o1), c_emN = I(CI_NMASSDTI, o1), c_fmN = I(CI_FMN, o1), c_fyN = I(CI_FORCEYN, o1);
        double c_fcE = 0.0, c_fcN = 0.0, c_aiE = 0.0, c_cwE = 0.0, c_wxE = 0.0, c_tbE = 0.0, c_rhE = 0.0, c_aiN = 0.0, c_cwN = 0.0, c_wyN = 0.0, c_tbN = 0.0, c_rhN = 0.0;
        if (FAST) {
            c_fcE = cg_ld(A.facE, o1); c_fcN = cg_ld(A.facN, o1);
        } else {
            c_aiE = I(CI_AIE, o1); c_cwE = I(CI_CWE, o1); c_wxE = I(CI_WATERXE, o1); c_tbE = I(CI_TBE, o1); c_rhE = I(CI_RHEOE, o1);
            c_aiN = I(CI_AIN, o1); c_cwN = I(CI_CWN, o1); c_wyN = I(CI_WATERYN, o1); c_tbN = I(CI_TBN, o1); c_rhN = I(CI_RHEON, o1);
        }
        // classic EVP (revp == 0): revp * uvelE_init is a zero and the initial velocities are not read (as in the B-grid kernels,
        // evp_cell.inc: the sum keeps the add of +0; a zero of the other sign could only show where uold is -0 exactly)
        double c_uiE = 0.0, c_viN = 0.0;
        if (revised) { c_uiE = I(CI_UE_INIT, o1); c_viN = I(CI_VN_INIT, o1); }
        const double c_strU = AVGS ? cg_ld(A.strengthU, o1) : 0.0;
        // ---- rows move up: last iteration's "north" is this iteration's own row ----
        uE1 = uE0; uE0 = uEN; uEN = a_uE;
        dxE1 = dxE0; dyE1 = dyE0; ea1 = ea0;
        dxE0 = dxEN; dyE0 = dyEN; ea0 = eaNc; dyEN = a_dyE;
        const double eh0 = ehN, wdyE0 = wdyEN;              // (LEN) HTN of the east, HTE of the west neighbour, row j
        if (LEN) {
            h1 = h0; h0 = hN; hN = a_dxE;
            ehN = cg_lane_dn(hN); wdyEN = cg_lane_up(dyEN);
            dxEN = 0.25 * (hN + ehN + h0 + eh0);            // dxE = mean of the four HTN around the E face (ice_grid.F90:3139-3146)
        } else {
            dxEN = a_dxE;
        }
        const double P0 = PNc;                              // uE * earea of row j: last iteration's row j+1
        if (LAST || AVGS) g1 = g0;
        g0 = gN; gN = a_g;
        const double eaN = dxEN * dyEN;                     // earea = dxE * dyE (ice_grid.F90:684), row j+1
        const double PN = uEN * eaN;
        // (LEN: dxN = HTN; dyN = mean of the four HTE around the N face; dxU, dyU, dxT, dyT = two-point means, ice_grid.F90:3099, 3207, 3120, 3237)
        const double vN0 = b_vN, dxN0 = LEN ? h0 : b_dxN, dyN0 = LEN ? 0.25 * (dyE0 + wdyE0 + dyEN + wdyEN) : b_dyN,
                     dxU0 = LEN ? 0.5 * (h0 + eh0) : b_dxU, dyU0 = LEN ? 0.5 * (dyE0 + dyEN) : b_dyU,
                     dxT0 = LEN ? 0.5 * (h0 + h1) : b_dxT, dyT0 = LEN ? 0.5 * (dyE0 + wdyE0) : b_dyT;
        const double na0 = dxN0 * dyN0, ua0 = dxU0 * dyU0, ta0 = dxT0 * dyT0;
        const double Q0 = vN0 * na0;
        const double hm0 = (g0 & 8u) ? 1.0 : 0.0;
        const double W0 = hm0 * ta0;

        double sh = 0.0, uNo = 0.0, vEo = 0.0, eta = 0.0, sp = 0.0, sm = 0.0, R0 = 0.0, SS0 = 0.0, SU0 = 0.0, DV0 = 0.0, VQ0 = 0.0;
        double uU0 = 0.0, vU0 = 0.0, dlt = 0.0;
        if (j >= ja - 2) {
            // ---- S (row j): strain_rates_U's shear at the corner (ice_dyn_shared.F90:2341-2444), the two averages of level C ----
            const double ea0W = cg_lane_up(ea0), eaNW = cg_lane_up(eaN), P0W = cg_lane_up(P0), PNW = cg_lane_up(PN);
            const double na0E = cg_lane_dn(na0), Q0E = cg_lane_dn(Q0), na1E = cg_lane_dn(na1), Q1E = cg_lane_dn(Q1);
            const double vN0E = cg_lane_dn(vN0), dyN0E = cg_lane_dn(dyN0), dxN0E = cg_lane_dn(dxN0);
            const unsigned g0E = cg_lane_dn_u(g0);
            const double epc = (g0 & 1u) ? 1.0 : 0.0, npc = (g0 & 2u) ? 1.0 : 0.0, uvm = (g0 & 4u) ? 1.0 : 0.0;
            const double npe = (g0E & 2u) ? 1.0 : 0.0, epn = (gN & 1u) ? 1.0 : 0.0;
            {
                const double wt = (ea0 + eaN);
                const double uU = (wt == 0.0 ? 0.0 : (P0 + PN) / wt) * uvm;          // avg_2(uE, earea, o, n)
                const double wv = (na0 + na0E);
                const double vU = (wv == 0.0 ? 0.0 : (Q0 + Q0E) / wv) * uvm;          // avg_2(vN, narea, o, e)
                const double ddyN = dyN0E - dyN0, ddxE = dxEN - dxE0;
                double rxN = -1.0, rxNr = -1.0, ryE = -1.0, ryEr = -1.0;
                if (npc != npe) { rxN = -(dxN0E / dxN0); rxNr = 1.0 / -(dxN0E / dxN0); }
                if (epc != epn) { ryE = -(dyEN / dyE0); ryEr = 1.0 / -(dyEN / dyE0); }
                const double uEo = uE0, uEn = uEN, vNo = vN0, vNe = vN0E;
                const double uEijp1 = uEn * epn + (epc - epn) * epc * ryE * uEo;
                const double uEij = uEo * epc + (epn - epc) * epn * ryEr * uEn;
                const double vNip1j = vNe * npe + (npc - npe) * npc * rxN * vNo;
                const double vNij = vNo * npc + (npe - npc) * npe * rxNr * vNe;
                sh = dxU0 * (uEijp1 - uEij) - uU * ddxE + dyU0 * (vNip1j - vNij) - vU * ddyN;
                if (!(m & 2u)) sh = b_shu;                 // strain_rates_U leaves cells without ice alone
                else if (LAST && ownx && j >= ja && j <= jb) cg_st(A.f[CF_SHEARU], cell * 8u, sh);
                uU0 = uU; vU0 = vU;
            }
            if (j >= ja - (AVGS ? 1 : 0) && j <= jb + ((LAST || AVGS) ? 1 : 0)) {     // (owned rows: what level C reads one row behind; LAST, AVGS: what deltaU takes too)
                const double wn = (ea0W + ea0 + eaNW + eaN);
                uNo = (wn == 0.0 ? 0.0 : (P0W + P0 + PNW + PN) / wn) * npc;           // avg_nw(uE, earea, o)
                const double ws = (na1 + na1E + na0 + na0E);
                vEo = (ws == 0.0 ? 0.0 : (Q1 + Q1E + Q0 + Q0E) / ws) * epc;           // avg_se(vN, narea, o)
            }
            if (LAST || AVGS) {
                // ---- deltaU of row j-1 (strain_rates_U, ice_dyn_shared.F90:2341-2444: the divergence and the tension at the corner) ----
                const double uNe = cg_lane_dn(un1), dxN1E = cg_lane_dn(dxN1), dyN1E = cg_lane_dn(dyN1);
                const unsigned g1E = cg_lane_dn_u(g1);
                if (AVGS ? (j >= ja) : (j > ja && j <= jb + 1 && ownx && (m1 & 2u))) {
                    const double epc = (g1 & 1u) ? 1.0 : 0.0, npc = (g1 & 2u) ? 1.0 : 0.0;
                    const double npe = (g1E & 2u) ? 1.0 : 0.0, epn = (g0 & 1u) ? 1.0 : 0.0;
                    double rxN = -1.0, rxNr = -1.0, ryE = -1.0, ryEr = -1.0;
                    if (npc != npe) { rxN = -(dxN1E / dxN1); rxNr = 1.0 / -(dxN1E / dxN1); }
                    if (epc != epn) { ryE = -(dyE0 / dyE1); ryEr = 1.0 / -(dyE0 / dyE1); }
                    const double uNo_ = un1, vEo_ = ve1, vEn = vEo;
                    const double ddyN = dyN1E - dyN1, ddxE = dxE0 - dxE1;
                    const double uNip1j = uNe * npe + (npc - npe) * npc * rxN * uNo_;
                    const double uNij = uNo_ * npc + (npe - npc) * npe * rxNr * uNe;
                    const double vEijp1 = vEn * epn + (epc - epn) * epc * ryE * vEo_;
                    const double vEij = vEo_ * epc + (epn - epc) * epn * ryEr * vEn;
                    const double dv = dyU1 * (uNip1j - uNij) + uU1 * ddyN + dxU1 * (vEijp1 - vEij) + vU1 * ddxE;
                    const double tn = dyU1 * (uNip1j - uNij) - uU1 * ddyN - dxU1 * (vEijp1 - vEij) + vU1 * ddxE;
                    dlt = sqrt(dv * dv + p.e_factor * (tn * tn + sh1 * sh1));
                    if (LAST && j > ja && j <= jb + 1 && ownx && (m1 & 2u)) cg_st(A.f[CF_DELTAU], (cell - nx) * 8u, dlt);
                }
            }
            // ---- T (row j): stressC_T (ice_dyn_evp.F90:1758-1860) ----
            {
                const double DU0 = dyE0 * uE0, UQ0 = uE0 / dyE0;
                DV0 = dxN0 * vN0; VQ0 = vN0 / dxN0;
                SS0 = sh * sh * ua0; SU0 = sh * ua0;
                const double DU0W = cg_lane_up(DU0), UQ0W = cg_lane_up(UQ0), ua0W = cg_lane_up(ua0), ua1W = cg_lane_up(ua1);
                const double SS0W = cg_lane_up(SS0), SS1W = cg_lane_up(SS1), SU0W = cg_lane_up(SU0), SU1W = cg_lane_up(SU1);
                sp = b_sp; sm = b_sm; eta = b_eta;
                if (m & 1u) {
                    const double divT = DU0 - DU0W + DV0 - DV1;
                    const double tensionT = (dyT0 * dyT0) * (UQ0 - UQ0W) - (dxT0 * dxT0) * (VQ0 - VQ1);
                    const double uareaavgr = 1.0 / (ua0 + ua1 + ua1W + ua0W);
                    const double shearTsqr = (SS0 + SS1 + SS1W + SS0W) * uareaavgr;
                    const double shearT = (SU0 + SU1 + SU1W + SU0W) * uareaavgr;
                    const double DeltaT = sqrt(divT * divT + p.e_factor * (tensionT * tensionT + shearTsqr));
                    double zetax2, etax2, rep_prs;
                    visc_replpress(p, b_str, dmin * ta0, DeltaT, zetax2, etax2, rep_prs);
                    eta = etax2;
                    sp = (b_sp * relax + p.arlx1i * (zetax2 * divT - rep_prs)) * p.denom1;
                    sm = (b_sm * relax + p.arlx1i * etax2 * tensionT) * p.denom1;
                    if (ownx && j >= ja && j <= jb) {
                        const unsigned o0 = cell * 8u;
                        cg_st(A.f[CF_S12T], o0, (b_s12t * relax + p.arlx1i * 0.5 * etax2 * shearT) * p.denom1);
                        cg_st(A.f[CF_SP], o0, sp);
                        cg_st(A.f[CF_SM], o0, sm);
                        if (LAST) {
                            cg_st(A.f[CF_ZETA], o0, zetax2);
                            cg_st(A.f[CF_ETA], o0, eta);
                        }
                    }
                }
                R0 = hm0 * eta * ta0;                       // the position's term of the T -> U average of etax2T
            }
        }
        if (j >= ja) {
            // ---- U (row j-1): stressC_U with the T -> U average of etax2T (ice_dyn_evp.F90:1862-1972, ice_grid.F90 grid_average_X2Y 'NE') ----
            double e2;
            if (AVGS) {
                double z, r;
                visc_replpress(p, c_strU, dmin * ua1, dlt, z, e2, r);
            } else {
                const double W1E = cg_lane_dn(W1), W0E = cg_lane_dn(W0), R1E = cg_lane_dn(R1), R0E = cg_lane_dn(R0);
                const double wtmp = (W1 + W1E + W0 + W0E);
                e2 = wtmp == 0.0 ? 0.0 : (R1 + R1E + R0 + R0E) / wtmp;
            }
            double s12 = c_s12;
            const double upd = (s12 * relax + p.arlx1i * 0.5 * e2 * sh1) * p.denom1;
            if (m1 & 2u) s12 = upd;
            // ---- C (row j-1): div_stress_Ex / _Ny, stepu_C / stepv_C (ice_dyn_evp.F90:2195-2416, ice_dyn_shared.F90:1090-1290) ----
            const double s12w = cg_lane_up(s12), spe = cg_lane_dn(sp1), sme = cg_lane_dn(sm1);
            const double YT1E = cg_lane_dn(YT1), YU1W = cg_lane_up(YU1);
            if (j > ja && ownx) {
                const unsigned oc = (cell - nx) * 8u;
                const double s12c = s12, s12s = s12_2;
                const double spc = sp1, smc = sm1, spn = sp, smn = sm;
                double unew, vnew, strintx_ = 0.0, strinty_ = 0.0, taubx_ = 0.0, tauby_ = 0.0;
                {
                    const double dyE = dyE1, dxE = dxE1;
                    const double earear = ea1 > 0.0 ? 1.0 / ea1 : 0.0;
                    const double strintx = (FAST ? earear : c_rhE * earear) *
                                           (0.5 * dyE * (spe - spc) + (0.5 / dyE) * (YT1E * sme - YT1 * smc) + (1.0 / dxE) * (XU1 * s12c - XU2 * s12s));
                    const double uold = uE1, vold = ve1;
                    const double uocn = c_uoE;
                    const double du = uocn - uold, dv = c_voE - vold;
                    const double vrel = (FAST ? c_fcE : c_aiE * p.rhow * c_cwE) * sqrt(du * du + dv * dv);
                    const double taux = vrel * (FAST ? uocn : c_wxE);
                    double Cb = 0.0;
                    if (!FAST) {
                        const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
                        Cb = c_tbE / ccc;
                    }
                    const double massdti = c_emE, fm = c_fmE;
                    const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
                    const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
                    const double cc1 = strintx + c_fxE + taux + massdti * (p.brlx * uold + p.revp * c_uiE);
                    unew = (ccb * vold + cc1) / cca;
                    strintx_ = strintx; taubx_ = -unew * Cb;
                }
                {
                    const double dxN = dxN1, dyN = dyN1;
                    const double narear = na1 > 0.0 ? 1.0 / na1 : 0.0;
                    const double XT0 = dxT0 * dxT0;
                    const double strinty = (FAST ? narear : c_rhN * narear) *
                                           (0.5 * dxN * (spn - spc) - (0.5 / dxN) * (XT0 * smn - XT1 * smc) + (1.0 / dyN) * (YU1 * s12c - YU1W * s12w));
                    const double uold = un1, vold = vN1;
                    const double vocn = c_voN;
                    const double du = c_uoN - uold, dv = vocn - vold;
                    const double vrel = (FAST ? c_fcN : c_aiN * p.rhow * c_cwN) * sqrt(du * du + dv * dv);
                    const double tauy = vrel * (FAST ? vocn : c_wyN);
                    double Cb = 0.0;
                    if (!FAST) {
                        const double ccc = sqrt(uold * uold + vold * vold) + p.u0;
                        Cb = c_tbN / ccc;
                    }
                    const double massdti = c_emN, fm = c_fmN;
                    const double cca = (p.brlx + p.revp) * massdti + vrel * p.cosw + Cb;
                    const double ccb = fm + copysign(1.0, fm) * vrel * p.sinw;
                    const double cc2 = strinty + c_fyN + tauy + massdti * (p.brlx * vold + p.revp * c_viN);
                    vnew = (-ccb * uold + cc2) / cca;
                    strinty_ = strinty; tauby_ = -vnew * Cb;
                }
                if (m1 & 2u) cg_st(A.f[CF_S12U], oc, s12c);
                if (LAST && !AVGS) cg_st(A.f[CF_ETAU], oc, e2);     // (avg_strength: the reference never stores etax2U)
                if (m1 & 4u) {
                    cg_st(A.f[CF_UE], oc, unew);
                    if (LAST) { cg_st(A.f[CF_STRX], oc, strintx_); cg_st(A.f[CF_TAUBX], oc, taubx_); }
                }
                if (m1 & 8u) {
                    cg_st(A.f[CF_VN], oc, vnew);
                    if (LAST) { cg_st(A.f[CF_STRY], oc, strinty_); cg_st(A.f[CF_TAUBY], oc, tauby_); }
                }
            }
            s12_2 = s12;
        }
        // ---- the rows move on ----
        eaNc = eaN; PNc = PN;
        vN1 = vN0; dxN1 = dxN0; dyN1 = dyN0; na1 = na0; Q1 = Q0; DV1 = DV0; VQ1 = VQ0; ua1 = ua0;
        XU2 = XU1; XU1 = dxU0 * dxU0; YU1 = dyU0 * dyU0; XT1 = dxT0 * dxT0; YT1 = dyT0 * dyT0; W1 = W0;
        sh1 = sh; SS1 = SS0; SU1 = SU0; un1 = uNo; ve1 = vEo; R1 = R0; sp1 = sp; sm1 = sm; m1 = m;
        if (LAST || AVGS) { dxU1 = dxU0; dyU1 = dyU0; uU1 = uU0; vU1 = vU0; }
    }
}

}  // namespace

void evp_launch_cgrid_zero_cells(const EvpCgrid &A, const int *cells, int n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(cg_zero_cells, dim3((n + 255) / 256), dim3(256), 0, st, A, cells, n);
}

static dim3 cg_grid(const EvpCgrid &A)
{
    const int gx = (A.nx + TX - 1) / TX, gy = (A.ny + TY - 1) / TY;
    if (A.xcd_rows > 0) {
        const int ngroups = (gy + 8 * A.xcd_rows - 1) / (8 * A.xcd_rows);
        return dim3((unsigned)(gx * A.xcd_rows * 8 * ngroups * A.nblocks), 1, 1);
    }
    return dim3(gx, gy, A.nblocks);
}

void evp_launch_cgrid_call_setup(const EvpCgrid &A, double *facE, double *facN, unsigned *flags, hipStream_t st)
{
    const dim3 grid = cg_grid(A), block(TX, TY);
    hipLaunchKernelGGL(cg_call_setup, grid, block, 0, st, A, facE, facN, flags);
}

void evp_launch_cgrid_mask(const EvpCgrid &A, const int *m4, hipStream_t st)
{
    const dim3 grid = cg_grid(A), block(TX, TY);
    hipLaunchKernelGGL(cg_mask_compose, grid, block, 0, st, A, m4, const_cast<uint8_t *>(A.mask), 0);
    hipLaunchKernelGGL(cg_mask_compose, grid, block, 0, st, A, m4, const_cast<uint8_t *>(A.mask), 1);
}

void evp_launch_cgrid_fold(const EvpCgFold &F, hipStream_t st)
{
    if (F.nfields <= 0 || F.maxn <= 0) return;
    if (F.maxn <= 1024) {            // one workgroup, one launch, values in registers
        hipLaunchKernelGGL(cg_fold_reg<1>, dim3(1), dim3(1024), 0, st, F);
        return;
    }
    if (F.maxn <= 2048) {
        hipLaunchKernelGGL(cg_fold_reg<2>, dim3(1), dim3(1024), 0, st, F);
        return;
    }
    if (F.maxn <= 8192) {            // one workgroup, one launch
        hipLaunchKernelGGL(cg_fold_one, dim3(1), dim3(1024), 0, st, F);
        return;
    }
    const dim3 grid((F.maxn + 255) / 256, F.nfields);
    hipLaunchKernelGGL(cg_fold_gather, grid, dim3(256), 0, st, F);
    hipLaunchKernelGGL(cg_fold_scatter, grid, dim3(256), 0, st, F);
}

void evp_launch_cgrid_umask(const EvpCgrid &A, double *scratch, int back, hipStream_t st)
{
    const dim3 grid = cg_grid(A), block(TX, TY);
    if (!back) hipLaunchKernelGGL(cg_umask_to_double, grid, block, 0, st, A, scratch);
    else hipLaunchKernelGGL(cg_bit5_from_double, grid, block, 0, st, A, scratch, const_cast<uint8_t *>(A.mask));
}

void evp_launch_cgrid_phase(const EvpCgrid &A, int phase, int last, hipStream_t st)
{
    const dim3 grid = cg_grid(A), block(TX, TY);
    switch (phase) {
    case 0: hipLaunchKernelGGL(cg_strain_u, grid, block, 0, st, A); break;
    // (A.gmask set: the fused kernels derive 15 of the 23 static arrays like cg_one does; the five-phase kernels always load)
    case 1: hipLaunchKernelGGL((cg_stress_t<true, false>), grid, block, 0, st, A, last); break;
    case 10:
        if (A.gmask) hipLaunchKernelGGL((cg_stress_t<false, true>), grid, block, 0, st, A, last);
        else hipLaunchKernelGGL((cg_stress_t<false, false>), grid, block, 0, st, A, last);
        break;
    case 7:
        if (A.gmask) hipLaunchKernelGGL(cg_avg_strain<true>, grid, block, 0, st, A, last);
        else hipLaunchKernelGGL(cg_avg_strain<false>, grid, block, 0, st, A, last);
        break;
    case 8:
        if (A.gmask) hipLaunchKernelGGL((cg_stress_u_step<false, true>), grid, dim3(TX, TY, A.split_faces ? 2 : 1), 0, st, A, last);
        else hipLaunchKernelGGL((cg_stress_u_step<false, false>), grid, dim3(TX, TY, A.split_faces ? 2 : 1), 0, st, A, last);
        break;
    case 11:
        if (A.gmask) hipLaunchKernelGGL((cg_stress_u_step<true, true>), grid, dim3(TX, TY, A.split_faces ? 2 : 1), 0, st, A, last);
        else hipLaunchKernelGGL((cg_stress_u_step<true, false>), grid, dim3(TX, TY, A.split_faces ? 2 : 1), 0, st, A, last);
        break;
    case 9: hipLaunchKernelGGL(cg_fill_images, grid, block, 0, st, A, last); break;
    case 2: hipLaunchKernelGGL(cg_stress_u, grid, block, 0, st, A); break;
    case 3: hipLaunchKernelGGL(cg_step, grid, block, 0, st, A); break;
    case 4: hipLaunchKernelGGL(cg_average, grid, block, 0, st, A); break;
    case 5: hipLaunchKernelGGL(cg_strength_u, grid, block, 0, st, A, const_cast<double *>(A.strengthU)); break;
    default: hipLaunchKernelGGL(cg_zero_outside, grid, block, 0, st, A); break;
    }
}

void evp_launch_cgrid_deformations(const EvpCgrid &A, const double *tarear, double *divu, double *shear, double *vort,
                                   double *rdg_conv, double *rdg_shear, hipStream_t st)
{
    hipLaunchKernelGGL(cg_deformations_t, cg_grid(A), dim3(TX, TY), 0, st, A, tarear, divu, shear, vort, rdg_conv, rdg_shear);
}

void evp_launch_cgrid_dyn_finish(const EvpCgrid &A, int which, double *strocnx, double *strocny, hipStream_t st)
{
    hipLaunchKernelGGL(cg_dyn_finish, cg_grid(A), dim3(TX, TY), 0, st, A, which, strocnx, strocny);
}

void evp_launch_cgrid_one(const EvpCgrid &A, const EvpCgOne &T, int fast, int last, hipStream_t st)
{
    const dim3 grid((unsigned)(8 * T.per_xcd)), block(T.ox, T.oy);
    const int mode = A.avg_strength ? 2 : (last ? 1 : 0);
#define CG_ONE(F, X, Y, M)                                                                       \
    do {                                                                                         \
        if (T.gmask) hipLaunchKernelGGL((cg_one<F, X, Y, M, true>), grid, block, 0, st, A, T, last); \
        else hipLaunchKernelGGL((cg_one<F, X, Y, M, false>), grid, block, 0, st, A, T, last);     \
    } while (0)
#define CG_ONE_M(F, X, Y)                 \
    do {                                  \
        if (mode == 0) CG_ONE(F, X, Y, 0); \
        else if (mode == 1) CG_ONE(F, X, Y, 1); \
        else CG_ONE(F, X, Y, 2);          \
    } while (0)
    if (T.ox == 32 && T.oy == 8) {
        if (fast) CG_ONE_M(true, 32, 8);
        else CG_ONE_M(false, 32, 8);
    } else if (T.oy == 8) {
        if (fast) CG_ONE_M(true, 64, 8);
        else CG_ONE_M(false, 64, 8);
    } else {
        if (fast) CG_ONE_M(true, 64, 16);
        else CG_ONE_M(false, 64, 16);
    }
#undef CG_ONE_M
#undef CG_ONE
}

void evp_launch_cgrid_strip(const EvpCgrid &A, const EvpCgOne &T, const EvpCgStrip &Z, const EvpCgOne *E, int fast, int last, hipStream_t st)
{
    if (Z.nitems <= 0) return;
    // E: windows of 32 x 8 positions to run in the same launch (NULL: none)
    EvpCgOne none = T;
    none.ntiles = 0;
    const EvpCgOne &W = (E && E->ntiles > 0) ? *E : none;
    const dim3 grid((unsigned)(8 * Z.per_xcd + W.ntiles)), block(256);
#define CG_STRIP(L, X, F) do { if (A.avg_strength) hipLaunchKernelGGL((cg_strip<L, X, F, true>), grid, block, 0, st, A, T, Z, W); \
                               else hipLaunchKernelGGL((cg_strip<L, X, F, false>), grid, block, 0, st, A, T, Z, W); } while (0)
#define CG_STRIP_F(L, X) do { if (fast) CG_STRIP(L, X, true); else CG_STRIP(L, X, false); } while (0)
    if (Z.lengths) {
        if (last) CG_STRIP_F(true, true);
        else CG_STRIP_F(true, false);
    } else {
        if (last) CG_STRIP_F(false, true);
        else CG_STRIP_F(false, false);
    }
#undef CG_STRIP_F
#undef CG_STRIP
}
