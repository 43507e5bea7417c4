// =====================================================================
// Preparation phase of evp() for grid_ice = 'C' on the device (SURVEY 8 f-2 for the C grid): what the reference does
// between dyn_prep1 and its subcycle loop at U, N and E points (dynamics/ice_dyn_evp.F90:430-453, 479-490, 563-691) --
//   grid_average_X2Y 'S' T -> U / E / N    infrastructure/ice_grid.F90:4190-4209, 4290-4306, 4332-4348
//   grid_average_X2Y 'F' T -> E / N        ice_grid.F90:4728-4744, 4766-4782
//   dyn_prep2 (X = U, N, E; rheofactX)     dynamics/ice_dyn_shared.F90:697-838
//   stresses zeroed off the ice            ice_dyn_evp.F90:676-691
// -- in one launch, one thread per cell: every average a face / corner reads back is its own.  dyn_prep1, the T-grid halo
// updates (evp_prep.hip) and the velocity averages / exchanges that follow (evp_cgrid.hip: cg_average, images, fold) are
// the existing kernels.  Once per evp() call, not a hot path: operation order of the reference, no FMA contraction, so
// that the loop starts from bit-identical inputs (tests/test_gpu_cgrid.py against the committed fixtures).
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evp_device.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ bool cell_of(const EvpCgPrep &P, int &i, int &j, int &bz, size_t &c)
{
    i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    j = blockIdx.y + 1;
    bz = blockIdx.z;
    if (i > P.nx) return false;
    c = (size_t)bz * P.plane + (size_t)(j - 1) * P.nx + (i - 1);
    return true;
}

// grid_average_X2YS over the cells c, q (two-point, E / N) -- 0 where the weight sum vanishes
__device__ __forceinline__ double avg_s2(const double *a, double mw0, double mw1, double m0, double m1, double w0, double w1,
                                         size_t c, size_t q)
{
    const double wtmp = (mw0 + mw1);
    if (wtmp == 0.0) return 0.0;
    return (m0 * a[c] * w0 + m1 * a[q] * w1) / wtmp;
}

__global__ void cg_prep(EvpCgPrep P)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const int4 r = P.blk[bz];
    const bool in = i >= r.x && i <= r.y && j >= r.z && j <= r.w;
    const bool iceT = P.maskd[c] != 0.0;
    P.m4[c] = iceT ? 1 : 0;
    // products per location (E = 1, N = 2): cdn, ai, uocn, vocn (whole-array zero fill of grid_average_X2Y), and
    // what dyn_prep2 zeroes on every cell
    double cdn[3] = {0, 0, 0}, ai[3] = {0, 0, 0}, uo[3] = {0, 0, 0}, vo[3] = {0, 0, 0};
    double water[3] = {0, 0, 0}, force[3] = {0, 0, 0}, massdti[3] = {0, 0, 0};
    bool ice[3] = {false, false, false};
    if (in) {
        const size_t ce = c + 1, cn = c + P.nx, cne = c + P.nx + 1;
        const double m0 = P.hm[c], m1 = P.hm[ce], m2 = P.hm[cn], m3 = P.hm[cne];
        const double w0 = P.tarea[c], w1 = P.tarea[ce], w2 = P.tarea[cn], w3 = P.tarea[cne];
        const double *tm = P.tmass, *a_init = P.t[3];
        for (int L = 0; L < 3; ++L) {
            double mass, aiX;
            const size_t q = L == 1 ? ce : cn;
            const double mq = L == 1 ? m1 : m2, wq = L == 1 ? w1 : w2;
            if (L == 0) {
                const double wtmp = (m0 * w0 + m1 * w1 + m2 * w2 + m3 * w3);
                mass = 0.0; aiX = 0.0;
                if (wtmp != 0.0) {
                    mass = (m0 * tm[c] * w0 + m1 * tm[ce] * w1 + m2 * tm[cn] * w2 + m3 * tm[cne] * w3) / wtmp;
                    aiX = (m0 * a_init[c] * w0 + m1 * a_init[ce] * w1 + m2 * a_init[cn] * w2 + m3 * a_init[cne] * w3) / wtmp;
                }
            } else {
                const double mw0 = m0 * w0, mw1 = mq * wq;
                mass = avg_s2(tm, mw0, mw1, m0, mq, w0, wq, c, q);
                aiX = avg_s2(a_init, mw0, mw1, m0, mq, w0, wq, c, q);
                cdn[L] = avg_s2(P.t[4], mw0, mw1, m0, mq, w0, wq, c, q);
                uo[L] = avg_s2(P.t[5], mw0, mw1, m0, mq, w0, wq, c, q);
                vo[L] = avg_s2(P.t[6], mw0, mw1, m0, mq, w0, wq, c, q);
                ai[L] = aiX;
            }
            int32_t *mX = P.m4 + (size_t)(1 + L) * P.n;
            const bool old = mX[c] != 0;
            const bool iceX = P.xmask[L][c] && (aiX > P.dyn_area_min) && (mass > P.dyn_mass_min);
            mX[c] = iceX ? 1 : 0;
            ice[L] = iceX;
            if (L == 0) continue;                 // at U points only the mask outlives the preparation (uvel, vvel: averages below)
            double *uX = P.f[L == 1 ? CF_UE : CF_UN], *vX = P.f[L == 1 ? CF_VE : CF_VN];
            double u = uX[c], v = vX[c];
            if (iceX) {
                if (!old) { u = uo[L]; v = vo[L]; }
            } else {
                u = 0.0; v = 0.0;
                P.f[L == 1 ? CF_STRX : CF_STRY][c] = 0.0;
            }
            uX[c] = u; vX[c] = v;
            P.in[L == 1 ? CI_UE_INIT : CI_VN_INIT][c] = L == 1 ? u : v;
            if (!iceX) continue;
            P.in[L == 1 ? CI_RHEOE : CI_RHEON][c] = aiX > 1e-3 ? 1.0 : 0.0;       // rheo_area_min, ice_dyn_shared.F90:67
            massdti[L] = mass / P.dt;
            const double fm = P.fcor[L][c] * mass;
            P.in[L == 1 ? CI_FME : CI_FMN][c] = fm;
            const double sg = copysign(1.0, fm);
            const double wx = uo[L] * P.cosw - vo[L] * P.sinw * sg;
            const double wy = vo[L] * P.cosw + uo[L] * P.sinw * sg;
            // the component this face carries: x at E, y at N
            const double *strair = P.t[L == 1 ? 9 : 10];
            const double air = 0.5 * (strair[c] * w0 + strair[q] * wq) / (L == 1 ? P.earea[c] : P.narea[c]);
            double tlt;
            if (P.ssh_coupled) {
                const double ss = avg_s2(P.t[L == 1 ? 7 : 8], m0 * w0, mq * wq, m0, mq, w0, wq, c, q);
                tlt = -P.gravit * mass * ss;
            } else {
                tlt = L == 1 ? -fm * vo[L] : fm * uo[L];
            }
            water[L] = L == 1 ? wx : wy;
            force[L] = air + tlt;
        }
    }
    P.in[CI_CWE][c] = cdn[1]; P.in[CI_AIE][c] = ai[1]; P.in[CI_UOCNE][c] = uo[1]; P.in[CI_VOCNE][c] = vo[1];
    P.in[CI_CWN][c] = cdn[2]; P.in[CI_AIN][c] = ai[2]; P.in[CI_UOCNN][c] = uo[2]; P.in[CI_VOCNN][c] = vo[2];
    P.in[CI_WATERXE][c] = water[1]; P.in[CI_FORCEXE][c] = force[1]; P.in[CI_EMASSDTI][c] = massdti[1];
    P.in[CI_WATERYN][c] = water[2]; P.in[CI_FORCEYN][c] = force[2]; P.in[CI_NMASSDTI][c] = massdti[2];
    P.in[CI_TBE][c] = 0.0; P.in[CI_TBN][c] = 0.0;
    P.f[CF_TAUBX][c] = 0.0; P.f[CF_TAUBY][c] = 0.0;
    // ice_dyn_evp.F90:676-691, every cell: the U mask of a ghost cell is whatever the caller's array holds there
    if (!iceT) { P.f[CF_SP][c] = 0.0; P.f[CF_SM][c] = 0.0; P.f[CF_S12T][c] = 0.0; }
    if (!(in ? ice[0] : (P.m4[P.n + c] != 0))) P.f[CF_S12U][c] = 0.0;
}

__global__ void cg_seabed_lkd(EvpCgPrep P, const uint8_t *__restrict__ mask, const double *__restrict__ hwater, double k1,
                              double k2, double alphab, double threshold_hw)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const int4 r = P.blk[bz];
    const bool in = i >= r.x && i <= r.y && j >= r.z && j <= r.w;
    const double *aice = P.t[0], *vice = P.t[1];
    for (int L = 1; L <= 2; ++L) {
        double tb = 0.0;
        if (in && (mask[c] & (L == 1 ? 4u : 8u))) {
            const size_t q = L == 1 ? c + 1 : c + P.nx;
            const double hwu = fmin(hwater[c], hwater[q]);
            const double docalc = hwu < threshold_hw ? 1.0 : 0.0;
            const double au = fmax(aice[c], aice[q]);
            const double hu = fmax(vice[c], vice[q]);
            const double hcu = au * hwu / k1;
            tb = docalc * k2 * fmax(0.0, (hu - hcu)) * exp(-alphab * (1.0 - au));
        }
        P.in[L == 1 ? CI_TBE : CI_TBN][c] = tb;
    }
}

__global__ void cg_seabed_prob_faces(EvpCgPrep P, const uint8_t *__restrict__ mask, const double *__restrict__ Tbt)
{
    int i, j, bz; size_t c;
    if (!cell_of(P, i, j, bz, c)) return;
    const int4 r = P.blk[bz];
    const bool in = i >= r.x && i <= r.y && j >= r.z && j <= r.w;
    P.in[CI_TBE][c] = (in && (mask[c] & 4u)) ? fmax(Tbt[c], Tbt[c + 1]) : 0.0;
    P.in[CI_TBN][c] = (in && (mask[c] & 8u)) ? fmax(Tbt[c], Tbt[c + P.nx]) : 0.0;
}

dim3 cell_grid(const EvpCgPrep &P, int nblocks) { return dim3((P.nx + 63) / 64, P.ny, nblocks); }

}  // namespace

void evp_launch_cgrid_prep(const EvpCgPrep &P, int nblocks, hipStream_t st)
{
    hipLaunchKernelGGL(cg_prep, cell_grid(P, nblocks), dim3(64), 0, st, P);
}
void evp_launch_cgrid_seabed_lkd(const EvpCgPrep &P, int nblocks, const uint8_t *mask, const double *hwater, double k1, double k2,
                                 double alphab, double threshold_hw, hipStream_t st)
{
    hipLaunchKernelGGL(cg_seabed_lkd, cell_grid(P, nblocks), dim3(64), 0, st, P, mask, hwater, k1, k2, alphab, threshold_hw);
}
void evp_launch_cgrid_seabed_prob_faces(const EvpCgPrep &P, int nblocks, const uint8_t *mask, const double *Tbt, hipStream_t st)
{
    hipLaunchKernelGGL(cg_seabed_prob_faces, cell_grid(P, nblocks), dim3(64), 0, st, P, mask, Tbt);
}
