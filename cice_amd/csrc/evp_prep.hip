// =====================================================================
// Preparation phase of evp() on the device: everything the
// reference does between entering evp() and its subcycle loop on the B grid
// (dynamics/ice_dyn_evp.F90:383-840), except icepack_ice_strength (Icepack) and
// the seabed stress factor (calls exp(); both stay with the host):
//   prep1a/prep1b   dyn_prep1                  ice_dyn_shared.F90:496-576
//   halo_center     ice_HaloUpdate, centre      ice_dyn_evp.F90:413-428, 466-470
//   prep_average    grid_average_X2Y T->U       ice_grid.F90:4183-4204 ('S'), 4650-4666 ('F')
//   prep2           dyn_prep2                   ice_dyn_shared.F90:586-839
// Not a hot path (once per evp() call): one thread per cell, operation order of the
// reference, no FMA contraction in either build mode so that it is bit-identical to it.
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evp_device.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ bool cell_of(const EvpPrep &P, int &i, int &j, int &bz, size_t &c)
{
    i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    j = blockIdx.y + 1;
    bz = blockIdx.z;
    if (i > P.nx) return false;
    c = (size_t)bz * P.plane + (size_t)(j - 1) * P.nx + (i - 1);
    return true;
}

// dyn_prep1, first loop (:545-558): mass and the "has ice" predicate on every cell
__global__ void prep1a(EvpPrep P)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const bool tm = P.tmask[c] != 0;
    const double tmass = tm ? (P.rhoi * P.t[1][c] + P.rhos * P.t[2][c]) : 0.0;
    P.tmass[c] = tmass;
    P.tmphm[c] = (uint8_t)(tm && (P.t[0][c] > P.dyn_area_min) && (tmass > P.dyn_mass_min));
}

// both loops of dyn_prep1 in one launch: the "has ice" predicate of the 3 x 3 neighbourhood is recomputed from the
// T-grid fields (same expression, same operands: ghost cells carry the caller's values in either form)
__global__ void prep1(EvpPrep P)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const bool tm = P.tmask[c] != 0;
    P.tmass[c] = tm ? (P.rhoi * P.t[1][c] + P.rhos * P.t[2][c]) : 0.0;
    const int4 r = P.blk[bz];
    double m = 0.0;
    if (i >= r.x && i <= r.y && j >= r.z && j <= r.w && tm) {
        bool any = false;
        for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) {
                const size_t q = c + (ptrdiff_t)dj * P.nx + di;
                const bool tq = P.tmask[q] != 0;
                const double tmass = tq ? (P.rhoi * P.t[1][q] + P.rhos * P.t[2][q]) : 0.0;
                any = any || (tq && (P.t[0][q] > P.dyn_area_min) && (tmass > P.dyn_mass_min));
            }
        m = any ? 1.0 : 0.0;
    }
    P.maskd[c] = m;
}

// dyn_prep1, second loop (:560-575): extent mask = any of the 3x3 neighbourhood, physical cells only
__global__ void prep1b(EvpPrep P)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const int4 r = P.blk[bz];
    double m = 0.0;
    if (i >= r.x && i <= r.y && j >= r.z && j <= r.w && P.tmask[c]) {
        unsigned any = 0;
        for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) any |= P.tmphm[c + (ptrdiff_t)dj * P.nx + di];
        m = any ? 1.0 : 0.0;
    }
    P.maskd[c] = m;
}

// ghost cells of cell-centre fields; is_vec: the vector kinds change sign across the tripole fold
__global__ void halo_center(EvpPrepHalo H)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= H.n) return;
    const int d = H.dst[t], s = H.src[t];
    const double vs = (double)H.vsign[t];
    for (int k = 0; k < H.narr; ++k) {
        double x = 0.0;
        if (s >= 0) x = (H.is_vec[k] ? vs : 1.0) * H.a[k][s];
        H.a[k][d] = x;
    }
}

// T -> U averages on the physical cells, 0 elsewhere (work2(:,:,:) = c0 first)
__device__ __forceinline__ void prep_average_cell(const EvpPrep &P, int i, int j, int bz, size_t c)
{
    const int4 r = P.blk[bz];
    const bool in = i >= r.x && i <= r.y && j >= r.z && j <= r.w;
    double o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (in) {
        const size_t c1 = c + 1, c2 = c + P.nx, c3 = c + P.nx + 1;
        const double m0 = P.hm[c], m1 = P.hm[c1], m2 = P.hm[c2], m3 = P.hm[c3];
        const double w0 = P.tarea[c], w1 = P.tarea[c1], w2 = P.tarea[c2], w3 = P.tarea[c3];
        const double wtmp = (m0 * w0 + m1 * w1 + m2 * w2 + m3 * w3);
        // state-masked: tmass, aice_init, cdn_ocn, uocn, vocn, ss_tltx, ss_tlty
        const double *src[7] = {P.tmass, P.t[3], P.t[4], P.t[5], P.t[6], P.t[7], P.t[8]};
        if (wtmp != 0.0)
            for (int k = 0; k < 7; ++k) {
                const double *a = src[k];
                o[k] = (m0 * a[c] * w0 + m1 * a[c1] * w1 + m2 * a[c2] * w2 + m3 * a[c3] * w3) / wtmp;
            }
        // flux: wind stress
        for (int k = 0; k < 2; ++k) {
            const double *a = P.t[9 + k];
            o[7 + k] = 0.25 * (a[c] * w0 + a[c1] * w1 + a[c2] * w2 + a[c3] * w3) / P.uarea[c];
        }
    }
    P.umass[c] = o[0]; P.aiU[c] = o[1]; P.cdn_ocnU[c] = o[2]; P.uocnU[c] = o[3]; P.vocnU[c] = o[4];
    P.ss_tltxU[c] = o[5]; P.ss_tltyU[c] = o[6]; P.strairxU[c] = o[7]; P.strairyU[c] = o[8];
}
__global__ void prep_average(EvpPrep P)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    prep_average_cell(P, i, j, bz, c);
}

// dyn_prep2 (:697-838)
__device__ __forceinline__ void prep2_cell(const EvpPrep &P, int i, int j, int bz, size_t c)
{
    const int4 r = P.blk[bz];
    const bool iceT = P.maskd[c] != 0.0;
    if (!iceT)
        for (int k = 0; k < 12; ++k) P.sig[k][c] = 0.0;
    double waterx = 0.0, watery = 0.0, forcex = 0.0, forcey = 0.0, umassdti = 0.0;
    bool iceU = false;
    // the seabed stress diagnostics are zeroed on every cell (:704-712); the subcycle writes them on
    // ice U-cells only, so a cell that has lost its ice must not keep the previous call's value
    P.taubx[c] = 0.0; P.tauby[c] = 0.0;
    if (i >= r.x && i <= r.y && j >= r.z && j <= r.w) {
        const double umass = P.umass[c], aiU = P.aiU[c];
        const bool old = P.umask_old[c] != 0;
        iceU = P.umask[c] && (aiU > P.dyn_area_min) && (umass > P.dyn_mass_min);
        double u = P.uvel[c], v = P.vvel[c];
        if (iceU) {
            if (!old) { u = P.uocnU[c]; v = P.vocnU[c]; }
        } else {
            u = 0.0; v = 0.0;
            P.strintx[c] = 0.0; P.strinty[c] = 0.0;      // :776-781
        }
        P.uvel[c] = u; P.vvel[c] = v;
        P.uvel_init[c] = u; P.vvel_init[c] = v;
        if (iceU) {
            umassdti = umass / P.dt;
            const double fm = P.fcor[c] * umass;
            P.fm[c] = fm;
            const double sg = copysign(1.0, fm);
            const double uo = P.uocnU[c], vo = P.vocnU[c];
            waterx = uo * P.cosw - vo * P.sinw * sg;
            watery = vo * P.cosw + uo * P.sinw * sg;
            double tx, ty;
            if (P.ssh_coupled) {
                tx = -P.gravit * umass * P.ss_tltxU[c];
                ty = -P.gravit * umass * P.ss_tltyU[c];
            } else {
                tx = -fm * vo;
                ty = fm * uo;
            }
            P.strtltx[c] = tx; P.strtlty[c] = ty;
            forcex = P.strairxU[c] + tx;
            forcey = P.strairyU[c] + ty;
            // does the subcycle kernel's waterx == uocn shortcut hold bit for bit?
            if (__double_as_longlong(waterx) != __double_as_longlong(uo) ||
                __double_as_longlong(watery) != __double_as_longlong(vo))
                atomicOr(P.flagword, 1u);
        }
    }
    P.waterx[c] = waterx; P.watery[c] = watery; P.forcex[c] = forcex; P.forcey[c] = forcey;
    P.umassdti[c] = umassdti;
    P.mask[c] = (uint8_t)((iceT ? 1 : 0) | (iceU ? 2 : 0));
}
__global__ void prep2(EvpPrep P)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    prep2_cell(P, i, j, bz, c);
}
// the averages of a U-cell are read back by dyn_prep2 at that very cell only: one launch, one thread does both
// (its own stores are visible to it)
__global__ void prep_average_prep2(EvpPrep P)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    prep_average_cell(P, i, j, bz, c);
    prep2_cell(P, i, j, bz, c);
}

__global__ void words_to_bytes(const int32_t *__restrict__ w, uint8_t *__restrict__ b, size_t n)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) b[k] = w[k] != 0;
}

// seabed_stress_factor_LKD (ice_dyn_shared.F90:1386-1460) on the ice U-cells, 0 elsewhere (dyn_prep2 zeroes
// TbU first, :706).  exp() is the device library's (<= 1 ulp): the one operation of this file whose last
// bit may differ from the host libm the reference calls -- tolerance stated in DESIGN.md.
__global__ void seabed_lkd(EvpPrep P, const double *__restrict__ hwater, double *__restrict__ TbU,
                           double k1, double k2, double alphab, double threshold_hw, unsigned *flagword)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const int4 r = P.blk[bz];
    double tb = 0.0;
    if (i >= r.x && i <= r.y && j >= r.z && j <= r.w && (P.mask[c] & 2u)) {
        const size_t c1 = c + 1, c2 = c + P.nx, c3 = c + P.nx + 1;
        // grid_neighbor_min / _max at the U point (ice_grid.F90:4974, 5005): min/max(a, b, c, d) left to right
        const double hwu = fmin(fmin(fmin(hwater[c], hwater[c1]), hwater[c2]), hwater[c3]);
        const double docalc = hwu < threshold_hw ? 1.0 : 0.0;
        const double *aice = P.t[0], *vice = P.t[1];
        const double au = fmax(fmax(fmax(aice[c], aice[c1]), aice[c2]), aice[c3]);
        const double hu = fmax(fmax(fmax(vice[c], vice[c1]), vice[c2]), vice[c3]);
        const double hcu = au * hwu / k1;
        tb = docalc * k2 * fmax(0.0, (hu - hcu)) * exp(-alphab * (1.0 - au));
        if (tb != 0.0) atomicOr(flagword, 2u);
    }
    TbU[c] = tb;
}

// seabed_stress_factor_prob (ice_dyn_shared.F90:1475-1683), step 1: the factor at the T point of every ice T-cell of
// dyn_prep2's list (ilo..ihi+1 x jlo..jhi+1) with atot > 0.05 and hwater < 50 m: a log-normal ice-thickness distribution
// (100 categories of 0.5 m) against a normal bathymetry distribution (100 categories over +-3 sigma_b).  Same expressions
// in the same order as the reference; exp() / log() are the device library's (<= 1 ulp from the host libm's), so the
// result is within a few ulp, not bit-identical -- tolerance stated in DESIGN.md.  aicen / vicen: (nx, ny, ncat, nblocks).
__global__ void seabed_prob_T(EvpPrep P, const double *__restrict__ hwater, const double *__restrict__ aicen,
                              const double *__restrict__ vicen, int ncat, double alphab, double rhoi, double rhow,
                              double gravit, double pi, double puny, double *__restrict__ Tbt)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const int4 r = P.blk[bz];
    double out = 0.0;
    if (i >= r.x && i <= r.y + 1 && j >= r.z && j <= r.w + 1 && (P.mask[c] & 1u)) {
        constexpr int NI = 100, NB = 100;
        const double max_depth = 50.0, mu_s = 0.1, sigma_b = 2.5, c0 = 0.0, c1 = 1.0, c2 = 2.0, c3 = 3.0, c6 = 6.0, p5 = 0.5;
        const size_t cl = c - (size_t)bz * P.plane;
        double atot = 0.0;
        for (int n = 0; n < ncat; ++n) atot += aicen[((size_t)bz * ncat + n) * P.plane + cl];
        if (atot > 0.05 && hwater[c] < max_depth) {
            const double mu_b = hwater[c];
            const double wid_i = max_depth / NI, wid_b = c6 * sigma_b / NB;
            double y_n[NB], P_y[NB];
            for (int k = 1; k <= NB; ++k) {
                y_n[k - 1] = (mu_b - c3 * sigma_b) + ((double)k - p5) * (c6 * sigma_b / NB);
                const double dy = y_n[k - 1] - mu_b;
                const double b_n = exp(-(dy * dy) / (c2 * (sigma_b * sigma_b))) / (sigma_b * sqrt(c2 * pi));
                P_y[k - 1] = b_n * wid_b;
            }
            double m_i = 0.0, v_i = c0;
            for (int n = 0; n < ncat; ++n) m_i += vicen[((size_t)bz * ncat + n) * P.plane + cl];
            for (int n = 0; n < ncat; ++n) {
                const double vc = vicen[((size_t)bz * ncat + n) * P.plane + cl], ac = aicen[((size_t)bz * ncat + n) * P.plane + cl];
                v_i = v_i + vc * vc / (fmax(ac, puny));
            }
            v_i = fmax((v_i - m_i * m_i), puny);
            const double mu_i = log(m_i / sqrt(c1 + v_i / (m_i * m_i)));
            const double sigma_i = sqrt(log(c1 + v_i / (m_i * m_i)));
            double x_kmax = exp(mu_i + sqrt(c2 * sigma_i) * 1.9430);
            const double cut = wid_i * ((double)NI - p5);      // x_k(ncat_i): the loop that would lower it never runs (:1583-1589)
            x_kmax = fmin(cut, x_kmax);
            double tsum = 0.0;
            for (int n = 1; n <= NI; ++n) {
                const double x_k = wid_i * ((double)n - p5);
                const double lx = log(x_k) - mu_i;
                const double g_k = exp(-(lx * lx) / (c2 * (sigma_i * sigma_i))) / (x_k * sigma_i * sqrt(c2 * pi));
                double P_x = g_k * wid_i;
                if (x_k > x_kmax) P_x = c0;
                int ii = 0;
                for (int k = 0; k < NB; ++k) ii += (y_n[k] <= rhoi * x_k / rhow) ? 1 : 0;
                double tb = c0;
                if (ii != 0) {
                    double sm = 0.0;
                    for (int k = 0; k < ii; ++k) sm += P_y[k] * (rhoi * x_k - rhow * y_n[k]);
                    tb = fmax(mu_s * gravit * P_x * sm, c0);
                }
                tsum += tb;
            }
            out = tsum * exp(-alphab * (c1 - atot));
        }
    }
    Tbt[c] = out;
}
// step 2 (:1648-1655): TbU = grid_neighbor_max(Tbt, 'U') on the ice U-cells, 0 elsewhere (dyn_prep2 zeroed TbU, :706)
__global__ void seabed_prob_U(EvpPrep P, const double *__restrict__ Tbt, double *__restrict__ TbU, unsigned *flagword)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const int4 r = P.blk[bz];
    double tb = 0.0;
    if (i >= r.x && i <= r.y && j >= r.z && j <= r.w && (P.mask[c] & 2u)) {
        tb = fmax(fmax(fmax(Tbt[c], Tbt[c + 1]), Tbt[c + P.nx]), Tbt[c + P.nx + 1]);
        if (tb != 0.0) atomicOr(flagword, 2u);
    }
    TbU[c] = tb;
}

// Cell-centre fields across a tripole fold whose row is split over ranks (halo_plan.h): the shifted copies the
// NE-corner exchange is run on -- a2[c] = a[c + nx + 1] on this rank's cells of row NY-1 -- and, after the exchange, the
// ghost cells it filled: a[d] = fa * a2[d] (fa = -1 undoes the exchange's sign for scalar kinds, +1 keeps it for vectors)
__global__ void fold_shift2(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ a2,
                            double *__restrict__ b2, const int *__restrict__ cells, int n, int nx)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int c = cells[t];
    a2[c] = a[c + nx + 1];
    b2[c] = b[c + nx + 1];
}
__global__ void fold_extract2(double *__restrict__ a, double *__restrict__ b, const double *__restrict__ a2,
                              const double *__restrict__ b2, const int *__restrict__ dst, int n, double fa, double fb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int d = dst[t];
    a[d] = fa * a2[d];
    b[d] = fb * b2[d];
}

dim3 cell_grid(const EvpPrep &P, int nblocks) { return dim3((P.nx + 63) / 64, P.ny, nblocks); }

}  // namespace

void evp_launch_prep1(const EvpPrep &P, int nblocks, hipStream_t st)
{
    hipLaunchKernelGGL(prep1, cell_grid(P, nblocks), dim3(64), 0, st, P);
}
void evp_launch_prep_average_prep2(const EvpPrep &P, int nblocks, hipStream_t st)
{
    hipLaunchKernelGGL(prep_average_prep2, cell_grid(P, nblocks), dim3(64), 0, st, P);
}
void evp_launch_words_to_bytes(const int32_t *w, uint8_t *b, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(words_to_bytes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, b, n);
}

void evp_launch_halo_center(const EvpPrepHalo &H, hipStream_t st)
{
    if (H.n <= 0 || H.narr <= 0) return;
    hipLaunchKernelGGL(halo_center, dim3((H.n + 255) / 256), dim3(256), 0, st, H);
}

void evp_launch_prep_average(const EvpPrep &P, int nblocks, hipStream_t st)
{
    hipLaunchKernelGGL(prep_average, cell_grid(P, nblocks), dim3(64), 0, st, P);
}

void evp_launch_prep2(const EvpPrep &P, int nblocks, hipStream_t st)
{
    hipLaunchKernelGGL(prep2, cell_grid(P, nblocks), dim3(64), 0, st, P);
}

void evp_launch_seabed_lkd(const EvpPrep &P, int nblocks, const double *hwater, double *TbU, double k1, double k2,
                           double alphab, double threshold_hw, unsigned *flagword, hipStream_t st)
{
    hipLaunchKernelGGL(seabed_lkd, cell_grid(P, nblocks), dim3(64), 0, st, P, hwater, TbU, k1, k2, alphab, threshold_hw, flagword);
}

void evp_launch_seabed_prob(const EvpPrep &P, int nblocks, const double *hwater, const double *aicen, const double *vicen,
                            int ncat, double alphab, double rhoi, double rhow, double gravit, double pi, double puny,
                            double *Tbt, double *TbU, unsigned *flagword, hipStream_t st)
{
    hipLaunchKernelGGL(seabed_prob_T, cell_grid(P, nblocks), dim3(64), 0, st, P, hwater, aicen, vicen, ncat, alphab, rhoi, rhow,
                       gravit, pi, puny, Tbt);
    hipLaunchKernelGGL(seabed_prob_U, cell_grid(P, nblocks), dim3(64), 0, st, P, Tbt, TbU, flagword);
}

// step 1 alone (the C grid takes the maximum over faces instead: evp_cgrid_prep.hip)
void evp_launch_seabed_prob_t(const EvpPrep &P, int nblocks, const double *hwater, const double *aicen, const double *vicen,
                              int ncat, double alphab, double rhoi, double rhow, double gravit, double pi, double puny,
                              double *Tbt, hipStream_t st)
{
    hipLaunchKernelGGL(seabed_prob_T, cell_grid(P, nblocks), dim3(64), 0, st, P, hwater, aicen, vicen, ncat, alphab, rhoi, rhow,
                       gravit, pi, puny, Tbt);
}

void evp_launch_fold_shift2(const double *a, const double *b, double *a2, double *b2, const int *cells, int n, int nx, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(fold_shift2, dim3((n + 255) / 256), dim3(256), 0, st, a, b, a2, b2, cells, n, nx);
}
void evp_launch_fold_extract2(double *a, double *b, const double *a2, const double *b2, const int *dst, int n, double fa, double fb,
                              hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(fold_extract2, dim3((n + 255) / 256), dim3(256), 0, st, a, b, a2, b2, dst, n, fa, fb);
}
