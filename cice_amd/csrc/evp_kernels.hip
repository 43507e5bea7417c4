// =====================================================================
// HIP kernels of the MI355X-native EVP subcycle (gfx950, wave64, fp64).
//
// One launch per subcycle: a fused stress + stepu kernel over dense masked
// tiles (replaces the reference's two sweeps over compressed index lists,
// ice_dyn_evp.F90:867-901), followed -- for ghost cells that cannot be folded
// into the stencil loads -- by a tiny halo gather kernel (replaces
// dyn_haloUpdate -> ice_HaloUpdate, ice_dyn_evp.F90:908-910).
//
// Jacobi semantics: stress of every T-cell uses the velocities of the
// previous subcycle (ice_dyn_evp.F90:867-901), so u,v are ping-ponged between
// two buffers.  The 12 stress components are ping-ponged as well: a tile
// recomputes the T-cells on its north/east fringe (owned by the neighbouring
// tile) and needs their OLD stresses while the owner is writing the new ones.
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evp_device.h"

namespace {

// shared/ice_constants.F90:79-85
__device__ constexpr double p027 = 1.0 / 36.0;
__device__ constexpr double p055 = 1.0 / 18.0;
__device__ constexpr double p111 = 1.0 / 9.0;
__device__ constexpr double p166 = 1.0 / 6.0;
__device__ constexpr double p222 = 2.0 / 9.0;
__device__ constexpr double p25 = 0.25;
__device__ constexpr double p333 = 1.0 / 3.0;
__device__ constexpr double p5 = 0.5;

}  // namespace

#pragma clang fp contract(off)
namespace evp_strict {
#include "evp_cell.inc"
}
#pragma clang fp contract(fast)
namespace evp_fused {
#include "evp_cell.inc"
}

namespace {

// ---------------------------------------------------------------------
// Fused stress + stepu, tile version.
//
// blockDim = (64, TYB).  A workgroup owns a tile of 64 x TYB T-cells whose
// south-west corner is (ilo + 63*bx, jlo + (TYB-1)*by) of CICE block `bz`; it
// produces the 63 x (TYB-1) U-cells at the same (i,j).  Lane = i so that every
// array access of a wave is one contiguous 512-byte row segment.
//
//   phase 1  each thread: stress update of T(i,j) -> 12 new stresses (stored by
//            the owning tile only) and the 8 partials str(i,j,1:8) -> LDS
//   phase 2  each thread with tx<63, ty<TYB-1: stepu of U(i,j) from
//            str(i,j,1|5) str(i+1,j,2|7) str(i,j+1,3|6) str(i+1,j+1,4|8)
//            (ice_dyn_shared.F90:948-951) read back from LDS
// ---------------------------------------------------------------------
template <int TYB, bool STRICT, int CAP>
__global__ __launch_bounds__(64 * TYB) void evp_subcycle_tile(EvpArgs A)
{
    __shared__ double s_str[8][TYB][64];

    const int tx = threadIdx.x, ty = threadIdx.y;
    const int bz = blockIdx.z;
    const int4 r = A.blk[bz];                       // ilo, ihi, jlo, jhi (1-based)
    const int i = r.x + blockIdx.x * 63 + tx;       // 1-based local indices
    const int j = r.z + blockIdx.y * (TYB - 1) + ty;
    const int nx = A.nx;
    const size_t base = (size_t)bz * A.plane;
    const size_t c = base + (size_t)(j - 1) * nx + (i - 1);

    const bool inT = (i <= r.y + 1) && (j <= r.w + 1);
    unsigned m = 0;
    if (inT) m = A.mask[c];

    double str[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) str[k] = 0.0;

    double u_ij = 0.0, v_ij = 0.0;
    if (inT && (m & 3u)) {
        u_ij = A.u_in[c];
        v_ij = A.v_in[c];
    }

    if (inT && (m & 1u)) {
        double s[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) s[k] = A.sig_in[k][c];
        if (STRICT) {
            evp_strict::StressIn a;
            a.u_ij = u_ij; a.v_ij = v_ij;
            a.u_im = A.u_in[c - 1]; a.v_im = A.v_in[c - 1];
            a.u_jm = A.u_in[c - nx]; a.v_jm = A.v_in[c - nx];
            a.u_mm = A.u_in[c - nx - 1]; a.v_mm = A.v_in[c - nx - 1];
            a.dxT = A.dxT[c]; a.dyT = A.dyT[c]; a.dxhy = A.dxhy[c]; a.dyhx = A.dyhx[c];
            a.cxp = A.cxp[c]; a.cyp = A.cyp[c]; a.cxm = A.cxm[c]; a.cym = A.cym[c];
            a.DminTarea = A.DminTarea[c]; a.strength = A.strength[c];
            evp_strict::stress_cell<CAP>(A.p, a, s, str);
        } else {
            evp_fused::StressIn a;
            a.u_ij = u_ij; a.v_ij = v_ij;
            a.u_im = A.u_in[c - 1]; a.v_im = A.v_in[c - 1];
            a.u_jm = A.u_in[c - nx]; a.v_jm = A.v_in[c - nx];
            a.u_mm = A.u_in[c - nx - 1]; a.v_mm = A.v_in[c - nx - 1];
            a.dxT = A.dxT[c]; a.dyT = A.dyT[c]; a.dxhy = A.dxhy[c]; a.dyhx = A.dyhx[c];
            a.cxp = A.cxp[c]; a.cyp = A.cyp[c]; a.cxm = A.cxm[c]; a.cym = A.cym[c];
            a.DminTarea = A.DminTarea[c]; a.strength = A.strength[c];
            evp_fused::stress_cell<CAP>(A.p, a, s, str);
        }
        // the tile that holds this T-cell off its north/east fringe owns it; the
        // ghost row/column ihi+1 / jhi+1 has no further tile and is owned here
        const bool own = (tx < 63 || i == r.y + 1) && (ty < TYB - 1 || j == r.w + 1);
        if (own) {
#pragma unroll
            for (int k = 0; k < 12; ++k) A.sig_out[k][c] = s[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_str[k][ty][tx] = str[k];
    __syncthreads();

    const bool isU = (tx < 63) && (ty < TYB - 1) && (i <= r.y) && (j <= r.w) && (m & 2u);
    if (isU) {
        if (STRICT) {
            evp_strict::StepuIn a;
            evp_strict::StepuOut o;
            a.uold = u_ij; a.vold = v_ij;
            a.Cw = A.Cw[c]; a.aiX = A.aiX[c]; a.uocn = A.uocn[c]; a.vocn = A.vocn[c];
            a.waterx = A.waterx[c]; a.watery = A.watery[c]; a.forcex = A.forcex[c];
            a.forcey = A.forcey[c]; a.Umassdti = A.umassdti[c]; a.fm = A.fm[c];
            a.uarear = A.uarear[c]; a.TbU = A.TbU[c];
            a.uvel_init = A.p.revp != 0.0 ? A.uvel_init[c] : 0.0;
            a.vvel_init = A.p.revp != 0.0 ? A.vvel_init[c] : 0.0;
            a.sx0 = s_str[0][ty][tx]; a.sx1 = s_str[1][ty][tx + 1];
            a.sx2 = s_str[2][ty + 1][tx]; a.sx3 = s_str[3][ty + 1][tx + 1];
            a.sy0 = s_str[4][ty][tx]; a.sy1 = s_str[5][ty + 1][tx];
            a.sy2 = s_str[6][ty][tx + 1]; a.sy3 = s_str[7][ty + 1][tx + 1];
            evp_strict::stepu_cell(A.p, a, o);
            A.u_out[c] = o.u; A.v_out[c] = o.v;
            if (A.last) {
                A.strintx[c] = o.strintx; A.strinty[c] = o.strinty;
                A.taubx[c] = o.taubx; A.tauby[c] = o.tauby;
            }
        } else {
            evp_fused::StepuIn a;
            evp_fused::StepuOut o;
            a.uold = u_ij; a.vold = v_ij;
            a.Cw = A.Cw[c]; a.aiX = A.aiX[c]; a.uocn = A.uocn[c]; a.vocn = A.vocn[c];
            a.waterx = A.waterx[c]; a.watery = A.watery[c]; a.forcex = A.forcex[c];
            a.forcey = A.forcey[c]; a.Umassdti = A.umassdti[c]; a.fm = A.fm[c];
            a.uarear = A.uarear[c]; a.TbU = A.TbU[c];
            a.uvel_init = A.p.revp != 0.0 ? A.uvel_init[c] : 0.0;
            a.vvel_init = A.p.revp != 0.0 ? A.vvel_init[c] : 0.0;
            a.sx0 = s_str[0][ty][tx]; a.sx1 = s_str[1][ty][tx + 1];
            a.sx2 = s_str[2][ty + 1][tx]; a.sx3 = s_str[3][ty + 1][tx + 1];
            a.sy0 = s_str[4][ty][tx]; a.sy1 = s_str[5][ty + 1][tx];
            a.sy2 = s_str[6][ty][tx + 1]; a.sy3 = s_str[7][ty + 1][tx + 1];
            evp_fused::stepu_cell(A.p, a, o);
            A.u_out[c] = o.u; A.v_out[c] = o.v;
            if (A.last) {
                A.strintx[c] = o.strintx; A.strinty[c] = o.strinty;
                A.taubx[c] = o.taubx; A.tauby[c] = o.tauby;
            }
        }
    }
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
#define EVP_LAUNCH(S, C) hipLaunchKernelGGL((evp_subcycle_tile<TYB, S, C>), grid, block, 0, st, A)
    if (strict) {
        if (cap == 1) EVP_LAUNCH(true, 1);
        else if (cap == 0) EVP_LAUNCH(true, 0);
        else EVP_LAUNCH(true, -1);
    } else {
        if (cap == 1) EVP_LAUNCH(false, 1);
        else if (cap == 0) EVP_LAUNCH(false, 0);
        else EVP_LAUNCH(false, -1);
    }
#undef EVP_LAUNCH
}

}  // namespace

// ---------------------------------------------------------------------
// host-callable launchers (declared in evp_device.h)
// ---------------------------------------------------------------------
void evp_launch_subcycle(const EvpArgs &A, int max_ni, int max_nj, int nblocks, int tyb,
                         bool strict, int cap, hipStream_t st)
{
    // a tile produces 63 x (TYB-1) U-cells and holds one more T row/column, so
    // ceil(ni/63) x ceil(nj/(TYB-1)) tiles also cover the T-cells ihi+1 / jhi+1
    const int gx = (max_ni + 62) / 63;
    if (tyb == 9) {
        dim3 grid(gx, (max_nj + 7) / 8, nblocks);
        launch_tile<9>(A, grid, st, strict, cap);
    } else {
        dim3 grid(gx, (max_nj + 3) / 4, nblocks);
        launch_tile<5>(A, grid, st, strict, cap);
    }
}

void evp_launch_halo_local(double *u, double *v, const int *dst, const int *src,
                           const signed char *sign, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(halo_local_uv, dim3((n + 255) / 256), dim3(256), 0, st, u, v, dst, src, sign, n);
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
