// =====================================================================
// HIP kernels of the MI355X-native EVP subcycle (gfx950, wave64, fp64).
//
// One launch per subcycle on a single GPU: a fused stress + stepu kernel over
// dense masked tiles (replaces the reference's two sweeps over compressed index
// lists, ice_dyn_evp.F90:867-901) whose edge threads also write the ghost-cell
// images of the velocities they produce (replaces dyn_haloUpdate ->
// ice_HaloUpdate for on-device neighbours, ice_dyn_evp.F90:908-910).
//
// Jacobi semantics: stress of every T-cell uses the velocities of the
// previous subcycle (ice_dyn_evp.F90:867-901), so u,v are ping-ponged between
// two buffers.  The 12 stress components are ping-ponged as well: a tile
// recomputes the T-cells on its north/east fringe (owned by the neighbouring
// tile) and needs their OLD stresses while the owner is writing the new ones.
// =====================================================================
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include <cstdlib>

#include "evp_device.h"
#include "evp_halo_direct.h"
#include "evp_math.h"

namespace {

// ---------------------------------------------------------------------
// Fused stress + stepu, tile version.
//
// blockDim = (64, TYB).  A workgroup owns a tile of 64 x TYB T-cells whose
// south-west corner is (ilo + 63*bx, jlo + (TYB-1)*by) of CICE block `bz`; it
// produces the 63 x (TYB-1) U-cells at the same (i,j).  Lane = i so that every
// array access of a wave is one contiguous 512-byte row segment.
//
//   phase 0  all loads of both phases are issued up front (one latency exposure)
//   phase 1  each thread: stress update of T(i,j) -> 12 new stresses (stored by
//            the owning tile only) and the 8 partials str(i,j,1:8) -> LDS
//   phase 2  each thread with tx<63, ty<TYB-1: stepu of U(i,j) from
//            str(i,j,1|5) str(i+1,j,2|7) str(i,j+1,3|6) str(i+1,j+1,4|8)
//            (ice_dyn_shared.F90:948-951) read back from LDS; threads on a block
//            edge also store u,v into the ghost cells that mirror their cell.
// ---------------------------------------------------------------------
template <int TYB, bool STRICT, int CAP, bool PRE>
__global__ __launch_bounds__(64 * TYB) void evp_subcycle_tile(EvpArgs A)
{
    using MM = Math<STRICT>;
    __shared__ double s_str[8][TYB][64];

    const int tx = threadIdx.x, ty = threadIdx.y;
    // XCD-aware tile order: workgroup w runs on XCD w % 8 (observed dispatch rule, used
    // for speed only).  Give every XCD one contiguous run of the tile sequence, ordered
    // with the tile row index fastest, so that vertically adjacent tiles -- which share
    // the T-row recomputed on the fringe -- are read through the same L2 close in time.
    int t;
    bool bnd = false;                                // tile whose cells the exchange workgroup sends
    if (A.dx) {
        // Mailbox halo riding in this launch: tiles that produce cells other ranks mirror come
        // first, then ONE workgroup that waits for them and runs the exchange while the
        // interior tiles (the rest of the grid) are computed.
        int w = blockIdx.x;
        if (w == A.dx_nb) {
            const unsigned target = (*A.dx_fseq + 1u) * (unsigned)A.dx_nb;
            evp_mailbox::exchange(*A.dx, A.u_out, A.v_out, ty * 64 + tx, 64 * TYB, A.dx_count, target);
            if (tx == 0 && ty == 0) *A.dx_fseq += 1u;
            return;
        }
        bnd = w < A.dx_nb;
        if (w > A.dx_nb) --w;
        t = A.tile_list[w];
    } else if (A.tile_list) {
        t = A.tile_list[blockIdx.x];                 // explicit subset (boundary / interior tiles)
    } else {
        const int w = blockIdx.x;
        const int per = (A.ntiles + 7) >> 3;
        t = A.xcdmap ? (w & 7) * per + (w >> 3) : w;
        if (t >= A.ntiles) return;
    }
    int bx, by;
    if (A.xcdmap == 1) {            // column runs: vertically adjacent tiles back to back
        by = t % A.gy;
        bx = (t / A.gy) % A.gx;
    } else {                        // row-major: horizontally adjacent tiles back to back
        bx = t % A.gx;
        by = (t / A.gx) % A.gy;
    }
    const int bz = t / (A.gy * A.gx);
    const int4 r = A.blk[bz];                       // ilo, ihi, jlo, jhi (1-based)
    const int i = r.x + bx * 63 + tx;               // 1-based local indices
    const int j = r.z + by * (TYB - 1) + ty;
    const int nx = A.nx;
    // 32-bit element index (every array is far below 2^31 elements): loads become
    // scalar-base + 32-bit-offset instead of a 64-bit address add per array
    const int c = bz * (int)A.plane + (j - 1) * nx + (i - 1);
    const unsigned ob = (unsigned)c * 8u;            // byte offset of this cell in every array
    const unsigned orow = (unsigned)nx * 8u;          // one row down
    auto LD = [](const double *p, unsigned off) -> double {
        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p) + off);
    };
    auto ST = [](double *p, unsigned off, double v) {
        *reinterpret_cast<double *>(reinterpret_cast<char *>(p) + off) = v;
    };
    const unsigned flags = A.flags;

    const bool inT = (i <= r.y + 1) && (j <= r.w + 1);
    unsigned m = 0;
    if (inT) m = A.mask[(unsigned)c];
    const bool actT = inT && (m & 1u);
    const bool isU = (tx < 63) && (ty < TYB - 1) && (i <= r.y) && (j <= r.w) && (m & 2u);

    // ---- phase 0: loads -------------------------------------------------
    double u_ij = 0.0, v_ij = 0.0;
    if (actT || isU) {
        u_ij = LD(A.u_in, ob);
        v_ij = LD(A.v_in, ob);
    }
    typename MM::SI a;
    double s[12];
    if (actT) {
#pragma unroll
        for (int k = 0; k < 12; ++k) s[k] = LD(A.sig_in[k], ob);
        a.u_ij = u_ij; a.v_ij = v_ij;
        a.u_im = LD(A.u_in, ob - 8u); a.v_im = LD(A.v_in, ob - 8u);
        a.u_jm = LD(A.u_in, ob - orow); a.v_jm = LD(A.v_in, ob - orow);
        a.u_mm = LD(A.u_in, ob - orow - 8u); a.v_mm = LD(A.v_in, ob - orow - 8u);
        a.dxT = LD(A.dxT, ob); a.dyT = LD(A.dyT, ob);
        a.strength = LD(A.strength, ob);
    }
    double hte = 0, hte_im = 0, htn = 0, htn_jm = 0;
    if (actT) {
        if (flags & EVP_F_METRICS) {
            hte = LD(A.HTE, ob); hte_im = LD(A.HTE, ob - 8u);
            htn = LD(A.HTN, ob); htn_jm = LD(A.HTN, ob - orow);
            if (flags & EVP_F_DXHY_ARRAY) { a.dxhy = LD(A.dxhy, ob); a.dyhx = LD(A.dyhx, ob); }
        } else {
            a.dxhy = LD(A.dxhy, ob); a.dyhx = LD(A.dyhx, ob);
            a.cxp = LD(A.cxp, ob); a.cyp = LD(A.cyp, ob); a.cxm = LD(A.cxm, ob); a.cym = LD(A.cym, ob);
            a.DminTarea = LD(A.DminTarea, ob);
        }
    }
    typename MM::UI q;
    auto load_u = [&]() {
        q.uold = u_ij; q.vold = v_ij;
        q.vrelfac = LD(A.vrelfac, ob);
        q.uocn = LD(A.uocn, ob); q.vocn = LD(A.vocn, ob);
        q.forcex = LD(A.forcex, ob); q.forcey = LD(A.forcey, ob);
        q.Umassdti = LD(A.umassdti, ob); q.fm = LD(A.fm, ob); q.uarear = LD(A.uarear, ob);
        if (flags & EVP_F_WATER_IS_OCN) { q.waterx = q.uocn; q.watery = q.vocn; }
        else { q.waterx = LD(A.waterx, ob); q.watery = LD(A.watery, ob); }
        q.TbU = (flags & EVP_F_TBU_ZERO) ? 0.0 : LD(A.TbU, ob);
        q.uvel_init = A.p.revp != 0.0 ? LD(A.uvel_init, ob) : 0.0;
        q.vvel_init = A.p.revp != 0.0 ? LD(A.vvel_init, ob) : 0.0;
    };
    if (PRE && isU) load_u();   // PRE: momentum operands in flight during the stress phase

    // ---- phase 1: stress ---------------------------------------------------
    double str[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) str[k] = 0.0;
    if (actT) {
        if (flags & EVP_F_METRICS) {
            const double dxhy_a = a.dxhy, dyhx_a = a.dyhx;
            MM::metrics(hte, hte_im, htn, htn_jm, A.deltaminEVP, a);
            if (flags & EVP_F_DXHY_ARRAY) { a.dxhy = dxhy_a; a.dyhx = dyhx_a; }
        }
        MM::template stress<CAP>(A.p, a, s, str);
        // the tile that holds this T-cell off its north/east fringe owns it; the
        // ghost row/column ihi+1 / jhi+1 has no further tile and is owned here
        const bool own = (tx < 63 || i == r.y + 1) && (ty < TYB - 1 || j == r.w + 1);
        if (own) {
#pragma unroll
            for (int k = 0; k < 12; ++k) ST(A.sig_out[k], ob, s[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_str[k][ty][tx] = str[k];
    __syncthreads();

    // ---- phase 2: momentum ----------------------------------------------------
    if (isU) {
        typename MM::UO o;
        if (!PRE) load_u();
        q.sx0 = s_str[0][ty][tx]; q.sx1 = s_str[1][ty][tx + 1];
        q.sx2 = s_str[2][ty + 1][tx]; q.sx3 = s_str[3][ty + 1][tx + 1];
        q.sy0 = s_str[4][ty][tx]; q.sy1 = s_str[5][ty + 1][tx];
        q.sy2 = s_str[6][ty][tx + 1]; q.sy3 = s_str[7][ty + 1][tx + 1];
        if (flags & EVP_F_TBU_ZERO) MM::template stepu<CAP, false>(A.p, q, o);
        else MM::template stepu<CAP, true>(A.p, q, o);
        if (bnd) {   // write-through: visible to the exchange workgroup on another XCD
            __hip_atomic_store(reinterpret_cast<double *>(reinterpret_cast<char *>(A.u_out) + ob), o.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<double *>(reinterpret_cast<char *>(A.v_out) + ob), o.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ST(A.u_out, ob, o.u); ST(A.v_out, ob, o.v);
        }
        if (A.last) {
            ST(A.strintx, ob, o.strintx); ST(A.strinty, ob, o.strinty);
            ST(A.taubx, ob, o.taubx); ST(A.tauby, ob, o.tauby);
        }
        if ((flags & EVP_F_PUSH) && (i == r.x || i == r.y || j == r.z || j == r.w)) {
            // ghost images of this cell (cyclic wrap / neighbouring block on this GPU)
            const int nslot = 2 * (A.push_nj + A.push_ni);
            const int *tab = A.push + (size_t)bz * nslot * 2;
            int slots[4];
            slots[0] = (i == r.x) ? (j - r.z) : -1;
            slots[1] = (i == r.y) ? A.push_nj + (j - r.z) : -1;
            slots[2] = (j == r.z) ? 2 * A.push_nj + (i - r.x) : -1;
            slots[3] = (j == r.w) ? 2 * A.push_nj + A.push_ni + (i - r.x) : -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (slots[e] < 0) continue;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int v = tab[slots[e] * 2 + w];
                    if (v >= 0) {
                        const double sg = (v & 1) ? -1.0 : 1.0;
                        A.u_out[v >> 1] = sg * o.u;
                        A.v_out[v >> 1] = sg * o.v;
                    }
                }
            }
        }
    }
    if (bnd) {       // check in: every velocity of this tile has reached memory
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tx == 0 && ty == 0) __hip_atomic_fetch_add(A.dx_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------
// Next tier (f-1): deformations and dyn_finish on the final velocities.
// One thread per cell, row-major 2-D launch over (nx, ny, nblocks).
// ---------------------------------------------------------------------
template <bool STRICT>
__global__ void deformations_kernel(EvpArgs A, const double *__restrict__ dxU, const double *__restrict__ dyU,
                                    const double *__restrict__ tarear, double *__restrict__ divu,
                                    double *__restrict__ shear, double *__restrict__ vort,
                                    double *__restrict__ rdg_conv, double *__restrict__ rdg_shear)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int bz = blockIdx.z;
    if (i > A.nx) return;
    const int nx = A.nx;
    const size_t c = (size_t)bz * A.plane + (size_t)(j - 1) * nx + (i - 1);
    const int4 r = A.blk[bz];
    double o_divu = 0, o_shear = 0, o_vort = 0, o_conv = 0, o_rshear = 0;   // zero off the ice (ice_dyn_evp.F90:385-393)
    const bool inT = i >= r.x && i <= r.y + 1 && j >= r.z && j <= r.w + 1;
    if (inT && (A.mask[c] & 1u)) {
        if (STRICT) {
            evp_strict::StressIn a; evp_strict::DeformOut o;
            a.u_ij = A.u_in[c]; a.v_ij = A.v_in[c]; a.u_im = A.u_in[c - 1]; a.v_im = A.v_in[c - 1];
            a.u_jm = A.u_in[c - nx]; a.v_jm = A.v_in[c - nx]; a.u_mm = A.u_in[c - nx - 1]; a.v_mm = A.v_in[c - nx - 1];
            a.dxT = A.dxT[c]; a.dyT = A.dyT[c]; a.cxp = A.cxp[c]; a.cyp = A.cyp[c]; a.cxm = A.cxm[c]; a.cym = A.cym[c];
            evp_strict::deform_cell(A.p, a, tarear[c], dyU[c], dyU[c - 1], dyU[c - nx], dyU[c - nx - 1],
                                    dxU[c], dxU[c - 1], dxU[c - nx], dxU[c - nx - 1], o);
            o_divu = o.divu; o_shear = o.shear; o_vort = o.vort; o_conv = o.rdg_conv; o_rshear = o.rdg_shear;
        } else {
            evp_fused::StressIn a; evp_fused::DeformOut o;
            a.u_ij = A.u_in[c]; a.v_ij = A.v_in[c]; a.u_im = A.u_in[c - 1]; a.v_im = A.v_in[c - 1];
            a.u_jm = A.u_in[c - nx]; a.v_jm = A.v_in[c - nx]; a.u_mm = A.u_in[c - nx - 1]; a.v_mm = A.v_in[c - nx - 1];
            a.dxT = A.dxT[c]; a.dyT = A.dyT[c]; a.cxp = A.cxp[c]; a.cyp = A.cyp[c]; a.cxm = A.cxm[c]; a.cym = A.cym[c];
            evp_fused::deform_cell(A.p, a, tarear[c], dyU[c], dyU[c - 1], dyU[c - nx], dyU[c - nx - 1],
                                   dxU[c], dxU[c - 1], dxU[c - nx], dxU[c - nx - 1], o);
            o_divu = o.divu; o_shear = o.shear; o_vort = o.vort; o_conv = o.rdg_conv; o_rshear = o.rdg_shear;
        }
    }
    divu[c] = o_divu; shear[c] = o_shear; vort[c] = o_vort; rdg_conv[c] = o_conv; rdg_shear[c] = o_rshear;
}

template <bool STRICT>
__global__ void dyn_finish_kernel(EvpArgs A, double *__restrict__ strocnx, double *__restrict__ strocny)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int bz = blockIdx.z;
    const int4 r = A.blk[bz];
    if (i < r.x || i > r.y || j < r.z || j > r.w) return;
    const size_t c = (size_t)bz * A.plane + (size_t)(j - 1) * A.nx + (i - 1);
    if (!(A.mask[c] & 2u)) return;
    double sx, sy;
    if (STRICT) evp_strict::finish_cell(A.p, A.Cw[c], A.aiX[c], A.uocn[c], A.vocn[c], A.u_in[c], A.v_in[c], A.fm[c], sx, sy);
    else evp_fused::finish_cell(A.p, A.Cw[c], A.aiX[c], A.uocn[c], A.vocn[c], A.u_in[c], A.v_in[c], A.fm[c], sx, sy);
    strocnx[c] = sx;
    strocny[c] = sy;
}

// (aiX*rhow)*Cw once per call: the leading factors of vrel in stepu
// (ice_dyn_shared.F90:933), multiplied in the reference's order.
__global__ void vrelfac_kernel(const double *__restrict__ aiX, const double *__restrict__ Cw,
                               double rhow, double *__restrict__ out, size_t n)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = aiX[t] * rhow * Cw[t];
}

// ---------------------------------------------------------------------
// Ghost-cell gather for cells whose source lives on this device
// (ice_boundary.F90:1372-1409 local copies; cyclic wrap; tripole fold sign).
// src < 0: eliminated land block -> fill 0 (srcBlock == 0, :1397-1407).
// ---------------------------------------------------------------------
__global__ void halo_local_uv(double *__restrict__ u, double *__restrict__ v,
                              const int *__restrict__ dst, const int *__restrict__ src,
                              const signed char *__restrict__ sign, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int s = src[t], d = dst[t];
    double uu = 0.0, vv = 0.0;
    if (s >= 0) {
        const double sg = (double)sign[t];
        uu = sg * u[s];
        vv = sg * v[s];
    }
    u[d] = uu;
    v[d] = vv;
}

// Tripole seam of the velocity (NE-corner, vector): one workgroup.
//   pairs:  (x_a, x_b) <- (xavg, -xavg), xavg = 0.5*(x_a + isign*x_b), isign = -1
//           (ice_boundary.F90:1630-1649)
//   poles:  x <- -x  (the pole points mirror onto themselves, copy-out :1689-1722)
//   late :  ghost copies whose source is a seam-row cell, repeated with the new values
// u2, v2 (may be null): the OTHER ping-pong buffer, which takes every value stored here as well.  The subcycle kernel writes
// ice cells only, so a seam-row cell without ice would otherwise meet, in the buffer the next-but-one subcycle writes, its
// value of two updates ago -- and a pole point changes sign with EVERY update (a zero there showed the wrong sign after
// 4k + 2 subcycles: found when the tests began to compare bit patterns, round 5).
__global__ void halo_seam_uv(double *__restrict__ u, double *__restrict__ v, double *__restrict__ u2, double *__restrict__ v2,
                             const int *__restrict__ pa, const int *__restrict__ pb, int npair,
                             const int *__restrict__ pole, int npole,
                             const int *__restrict__ ldst, const int *__restrict__ lsrc,
                             const signed char *__restrict__ lsign, int nlate)
{
    const double isign = -1.0;
    for (int k = threadIdx.x; k < npair; k += blockDim.x) {
        const int a = pa[k], b = pb[k];
        const double xu = 0.5 * (u[a] + isign * u[b]);
        const double xv = 0.5 * (v[a] + isign * v[b]);
        u[a] = xu; u[b] = isign * xu;
        v[a] = xv; v[b] = isign * xv;
        if (u2) { u2[a] = xu; u2[b] = isign * xu; v2[a] = xv; v2[b] = isign * xv; }
    }
    for (int k = threadIdx.x; k < npole; k += blockDim.x) {
        const int a = pole[k];
        const double xu = isign * u[a], xv = isign * v[a];
        u[a] = xu;
        v[a] = xv;
        if (u2) { u2[a] = xu; v2[a] = xv; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nlate; k += blockDim.x) {
        const double sg = (double)lsign[k];
        const double xu = sg * u[lsrc[k]], xv = sg * v[lsrc[k]];
        u[ldst[k]] = xu;
        v[ldst[k]] = xv;
        if (u2) { u2[ldst[k]] = xu; v2[ldst[k]] = xv; }
    }
}

// The same seam step for any rank layout (halo_plan.h, fin lists): every entry reads RAW values -- local cells or
// staging slots the exchange has just filled -- and only then are the results stored (own seam cells are both
// operands and destinations).  One workgroup; up to FIN_PER entries per thread are held in registers.
constexpr int FIN_PER = 8;
__global__ __launch_bounds__(1024) void halo_seam_fin(double *__restrict__ u, double *__restrict__ v, double *__restrict__ u2,
                                                      double *__restrict__ v2, const int *__restrict__ dst, const int *__restrict__ fa,
                                                      const int *__restrict__ fb, const signed char *__restrict__ coef, int n)
{
    const double isign = -1.0;
    double ru[FIN_PER], rv[FIN_PER];
#pragma unroll
    for (int e = 0; e < FIN_PER; ++e) {
        const int k = threadIdx.x + e * 1024;
        ru[e] = rv[e] = 0.0;
        if (k < n) {
            const int a = fa[k], b = fb[k];
            const double c = (double)coef[k];
            if (b >= 0) {      // pair: (x_a, x_b) <- (xavg, isign*xavg), xavg = 0.5*(x_a + isign*x_b)  (ice_boundary.F90:1630-1649)
                const double xu = 0.5 * (u[a] + isign * u[b]);
                const double xv = 0.5 * (v[a] + isign * v[b]);
                ru[e] = c * xu; rv[e] = c * xv;
            } else {
                ru[e] = c * u[a]; rv[e] = c * v[a];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < FIN_PER; ++e) {
        const int k = threadIdx.x + e * 1024;
        if (k < n) {
            u[dst[k]] = ru[e]; v[dst[k]] = rv[e];
            if (u2) { u2[dst[k]] = ru[e]; v2[dst[k]] = rv[e]; }      // (as in halo_seam_uv: both ping-pong buffers)
        }
    }
}

// ice_HaloUpdate_stress x 12 (ice_dyn_evp.F90:1321-1389): component k of a family takes the
// mirrored top row of its partner (1<->3, 2<->4 == index ^ 2) into its tripole ghost row.
// Reads interior rows, writes ghost rows: the 12 updates are independent.
struct SigTable { double *p[12]; };
__global__ void halo_stress_sym(SigTable T, const int *__restrict__ dst, const int *__restrict__ src, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int d = dst[t], s = src[t];
#pragma unroll
    for (int k = 0; k < 12; ++k) T.p[k][d] = (s >= 0) ? T.p[k ^ 2][s] : 0.0;
}

// tripoleT (halo_plan.h): the arrays _1, _2 of each family take the partner's mirrored cell of the top physical row; the
// east-west ghost cells of that row of _3, _4 become images of their own array.  Reads touch interior cells of _3 / _4 only,
// which nothing here writes.
__global__ void halo_stress_tfold(SigTable T, const int *__restrict__ dst, const int *__restrict__ src, int n,
                                  const int *__restrict__ odst, const int *__restrict__ osrc, int no)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const int d = dst[t], s = src[t];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (!(k & 2)) T.p[k][d] = (s >= 0) ? T.p[k ^ 2][s] : 0.0;
    }
    if (t < no) {
        const int d = odst[t], s = osrc[t];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (k & 2) T.p[k][d] = (s >= 0) ? T.p[k][s] : 0.0;
    }
}

// pack / unpack of remote halo cells (ice_boundary.F90:1260-1284, 1419-1449)
__global__ void halo_pack_uv(const double *__restrict__ u, const double *__restrict__ v,
                             const int *__restrict__ src, double *__restrict__ buf, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int s = src[t];
    buf[2 * (size_t)t] = u[s];
    buf[2 * (size_t)t + 1] = v[s];
}

__global__ void halo_unpack_uv(double *__restrict__ u, double *__restrict__ v,
                               const int *__restrict__ dst, const signed char *__restrict__ sign,
                               const double *__restrict__ buf, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int d = dst[t];
    const double sg = (double)sign[t];
    u[d] = sg * buf[2 * (size_t)t];
    v[d] = sg * buf[2 * (size_t)t + 1];
}

template <int TYB>
void launch_tile(const EvpArgs &A, dim3 grid, hipStream_t st, bool strict, int cap)
{
    dim3 block(64, TYB);
    const bool pre = evp_env_test("CICE_EVP_HIP_PREFETCH") && std::atoi(evp_env_test("CICE_EVP_HIP_PREFETCH"));
#define EVP_LAUNCH(S, C)                                                                           \
    do {                                                                                           \
        if (pre) hipLaunchKernelGGL((evp_subcycle_tile<TYB, S, C, true>), grid, block, 0, st, A);  \
        else hipLaunchKernelGGL((evp_subcycle_tile<TYB, S, C, false>), grid, block, 0, st, A);     \
    } while (0)
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

// ---------------------------------------------------------------------
// host-callable launchers (declared in evp_device.h)
// ---------------------------------------------------------------------
void evp_launch_subcycle(const EvpArgs &A0, int max_ni, int max_nj, int nblocks, int variant,
                         bool strict, int cap, hipStream_t st)
{
    // a tile produces 63 x (TYB-1) U-cells and holds one more T row/column, so
    // ceil(ni/63) x ceil(nj/(TYB-1)) tiles also cover the T-cells ihi+1 / jhi+1
    EvpArgs A = A0;
    int tyb = variant % 100;           // variant = tile height + 100 * (XCD-contiguous tile order)
    A.xcdmap = variant / 100;        // 0: plain row-major, 1: XCD-chunked column runs, 2: XCD-chunked row-major
    if (tyb < 2 || tyb > 9) tyb = 5;
    A.gx = (max_ni + 62) / 63;
    A.gy = (max_nj + tyb - 2) / (tyb - 1);
    A.ntiles = A.gx * A.gy * nblocks;
    const int per = (A.ntiles + 7) / 8;
    dim3 grid(per * 8);
    if (A.tile_list) {
        if (A.tile_count <= 0) return;
        grid = dim3(A.tile_count + (A.dx ? 1 : 0));      // + the exchange workgroup
    }
    switch (tyb) {
    case 2: launch_tile<2>(A, grid, st, strict, cap); break;
    case 3: launch_tile<3>(A, grid, st, strict, cap); break;
    case 4: launch_tile<4>(A, grid, st, strict, cap); break;
    case 6: launch_tile<6>(A, grid, st, strict, cap); break;
    case 7: launch_tile<7>(A, grid, st, strict, cap); break;
    case 8: launch_tile<8>(A, grid, st, strict, cap); break;
    case 9: launch_tile<9>(A, grid, st, strict, cap); break;
    default: launch_tile<5>(A, grid, st, strict, cap); break;
    }
}

void evp_launch_deformations(const EvpArgs &A, int nblocks, bool strict, const double *dxU, const double *dyU,
                              const double *tarear, double *divu, double *shear, double *vort,
                              double *rdg_conv, double *rdg_shear, hipStream_t st)
{
    dim3 grid((A.nx + 63) / 64, A.ny, nblocks), block(64);
    if (strict) hipLaunchKernelGGL(deformations_kernel<true>, grid, block, 0, st, A, dxU, dyU, tarear, divu, shear, vort, rdg_conv, rdg_shear);
    else hipLaunchKernelGGL(deformations_kernel<false>, grid, block, 0, st, A, dxU, dyU, tarear, divu, shear, vort, rdg_conv, rdg_shear);
}

void evp_launch_dyn_finish(const EvpArgs &A, int nblocks, bool strict, double *strocnx, double *strocny,
                           hipStream_t st)
{
    dim3 grid((A.nx + 63) / 64, A.ny, nblocks), block(64);
    if (strict) hipLaunchKernelGGL(dyn_finish_kernel<true>, grid, block, 0, st, A, strocnx, strocny);
    else hipLaunchKernelGGL(dyn_finish_kernel<false>, grid, block, 0, st, A, strocnx, strocny);
}

void evp_tile_geometry(int max_ni, int max_nj, int variant, int *tyb_out, int *gx, int *gy)
{
    int tyb = variant % 100;
    if (tyb < 2 || tyb > 9) tyb = 5;
    *tyb_out = tyb;
    *gx = (max_ni + 62) / 63;
    *gy = (max_nj + tyb - 2) / (tyb - 1);
}

void evp_launch_vrelfac(const double *aiX, const double *Cw, double rhow, double *out, size_t n,
                        hipStream_t st)
{
    hipLaunchKernelGGL(vrelfac_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, aiX, Cw, rhow, out, n);
}

void evp_launch_halo_local(double *u, double *v, const int *dst, const int *src,
                           const signed char *sign, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(halo_local_uv, dim3((n + 255) / 256), dim3(256), 0, st, u, v, dst, src, sign, n);
}

void evp_launch_halo_seam(double *u, double *v, double *u2, double *v2, const int *pa, const int *pb, int npair, const int *pole,
                          int npole, const int *ldst, const int *lsrc, const signed char *lsign,
                          int nlate, hipStream_t st)
{
    if (npair <= 0 && npole <= 0 && nlate <= 0) return;
    hipLaunchKernelGGL(halo_seam_uv, dim3(1), dim3(1024), 0, st, u, v, u2, v2, pa, pb, npair, pole, npole, ldst,
                       lsrc, lsign, nlate);
}

int evp_halo_seam_fin_capacity() { return FIN_PER * 1024; }

void evp_launch_halo_seam_fin(double *u, double *v, double *u2, double *v2, const int *dst, const int *fa, const int *fb,
                              const signed char *coef, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(halo_seam_fin, dim3(1), dim3(1024), 0, st, u, v, u2, v2, dst, fa, fb, coef, n);
}

void evp_launch_halo_stress(double *const *sig12, const int *dst, const int *src, int n, hipStream_t st)
{
    if (n <= 0) return;
    SigTable T;
    for (int k = 0; k < 12; ++k) T.p[k] = sig12[k];
    hipLaunchKernelGGL(halo_stress_sym, dim3((n + 255) / 256), dim3(256), 0, st, T, dst, src, n);
}

void evp_launch_halo_stress_tfold(double *const *sig12, const int *dst, const int *src, int n, const int *odst, const int *osrc, int no,
                                  hipStream_t st)
{
    if (n <= 0 && no <= 0) return;
    SigTable T;
    for (int k = 0; k < 12; ++k) T.p[k] = sig12[k];
    hipLaunchKernelGGL(halo_stress_tfold, dim3((std::max(n, no) + 255) / 256), dim3(256), 0, st, T, dst, src, n, odst, osrc, no);
}

void evp_launch_halo_pack(const double *u, const double *v, const int *src, double *buf, int n,
                          hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(halo_pack_uv, dim3((n + 255) / 256), dim3(256), 0, st, u, v, src, buf, n);
}

void evp_launch_halo_unpack(double *u, double *v, const int *dst, const signed char *sign,
                            const double *buf, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(halo_unpack_uv, dim3((n + 255) / 256), dim3(256), 0, st, u, v, dst, sign, buf, n);
}
