// =====================================================================
// Host <-> device traffic of one evp() call as TWO launches instead of ~50 copies.
//
// cice_evp_hip_run moves 20-32 fields in and 6-18 out, every one a separate host array of the
// caller (CICE's module arrays).  As hipMemcpyAsync calls each ~1 MB copy pays its own set-up on
// the copy engine (measured 30 GB/s for the whole batch on gx1).  Arrays the caller has page-locked
// (cice_evp_hip_pin_host -> hipHostRegisterMapped) are mapped into the device's address space, so one
// kernel can gather all of them over PCIe in a single launch, and one kernel scatters the outputs
// back: 16-byte accesses, consecutive lanes on consecutive addresses, every CU streaming.
// Also here: what dyn_prep2 does to stresses that stay on the device between calls (zero off the ice).
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evp_device.h"

namespace {

__global__ __launch_bounds__(256) void copy_many(EvpCopyTab T)
{
    const int a = blockIdx.y;
    const double *__restrict__ src = T.src[a];
    double *__restrict__ dst = T.dst[a];
    const size_t n = T.len;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (T.vec2) {
        const size_t n2 = n >> 1;
        const double2 *__restrict__ s2 = reinterpret_cast<const double2 *>(src);
        double2 *__restrict__ d2 = reinterpret_cast<double2 *>(dst);
        for (; i < n2; i += stride) d2[i] = s2[i];
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
    } else {
        for (; i < n; i += stride) dst[i] = src[i];
    }
}

struct SigTab { double *p[24]; };
// dyn_prep2 zeroes the 12 stress components wherever iceTmask is false (ice_dyn_shared.F90:712-727) on the
// host arrays before the loop; for stresses that never left the device the same, on both ping-pong copies
__global__ void zero_sig_off_mask(SigTab T, const uint8_t *__restrict__ mask, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (mask[i] & 1u)) return;
#pragma unroll
    for (int k = 0; k < 24; ++k) T.p[k][i] = 0.0;
}

}  // namespace

void evp_launch_copy_many(const EvpCopyTab &T, hipStream_t st)
{
    if (T.n <= 0 || T.len == 0) return;
    // enough workgroups to keep the link busy, few enough that one array is a handful of 64-KB bursts per workgroup
    const unsigned per = (unsigned)std::min<size_t>(64, (T.len / 2 + 255) / 256);
    hipLaunchKernelGGL(copy_many, dim3(per ? per : 1, T.n), dim3(256), 0, st, T);
}

void evp_launch_zero_sig_off_mask(double *const *sig0, double *const *sig1, const uint8_t *mask, size_t n, hipStream_t st)
{
    SigTab T;
    for (int k = 0; k < 12; ++k) { T.p[k] = sig0[k]; T.p[12 + k] = sig1[k]; }
    hipLaunchKernelGGL(zero_sig_off_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, T, mask, n);
}
