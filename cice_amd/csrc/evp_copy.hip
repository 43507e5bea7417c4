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
#include <vector>

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

// scatter of arrays a kernel wrote on masked cells only: the other cells of the destination keep what they hold
__global__ __launch_bounds__(256) void copy_many_masked(EvpCopyTab T, const uint8_t *__restrict__ mask, unsigned bit)
{
    const int a = blockIdx.y;
    const double *__restrict__ src = T.src[a];
    double *__restrict__ dst = T.dst[a];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.len; i += stride)
        if (mask[i] & bit) dst[i] = src[i];
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

// Plain streaming with the array shape of one B-grid subcycle (30 arrays in, 16 out, 368 B per cell, every element
// touched once, consecutive lanes on consecutive addresses, next to no arithmetic): what HBM gives a kernel of this
// shape on this box -- the practical ceiling the streaming subcycle kernel is compared with (bench.py).
struct StreamTab { const double *in[30]; double *out[16]; };
__global__ __launch_bounds__(256) void stream_30_16(StreamTab T, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 30; ++k) s += T.in[k][i];
#pragma unroll
    for (int k = 0; k < 16; ++k) T.out[k][i] = s + k;
}

}  // namespace

// returns seconds per launch (best of `reps` groups of 10), or a negative HIP error code
double evp_stream_probe(size_t ncells, int reps, hipStream_t st)
{
    StreamTab T{};
    std::vector<void *> owned;
    auto fail = [&](hipError_t e) {
        for (void *p : owned) (void)hipFree(p);
        return -(double)(int)e;
    };
    for (auto &p : T.in) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, ncells * sizeof(double));
        if (e != hipSuccess) return fail(e);
        owned.push_back(q);
        (void)hipMemsetAsync(q, 0, ncells * sizeof(double), st);
        p = (const double *)q;
    }
    for (auto &p : T.out) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, ncells * sizeof(double));
        if (e != hipSuccess) return fail(e);
        owned.push_back(q);
        p = (double *)q;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)((ncells + 255) / 256);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream_30_16, dim3(grid), dim3(256), 0, st, T, ncells);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0, st);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(stream_30_16, dim3(grid), dim3(256), 0, st, T, ncells);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (void *p : owned) (void)hipFree(p);
    return (double)best * 1e-4;
}

void evp_launch_copy_many(const EvpCopyTab &T, hipStream_t st)
{
    if (T.n <= 0 || T.len == 0) return;
    // enough workgroups to keep the link busy, few enough that one array is a handful of 64-KB bursts per workgroup
    const unsigned per = (unsigned)std::min<size_t>(64, (T.len / 2 + 255) / 256);
    hipLaunchKernelGGL(copy_many, dim3(per ? per : 1, T.n), dim3(256), 0, st, T);
}

void evp_launch_copy_many_masked(const EvpCopyTab &T, const uint8_t *mask, unsigned bit, hipStream_t st)
{
    if (T.n <= 0 || T.len == 0) return;
    const unsigned per = (unsigned)std::min<size_t>(64, (T.len + 255) / 256);
    hipLaunchKernelGGL(copy_many_masked, dim3(per ? per : 1, T.n), dim3(256), 0, st, T, mask, bit);
}

void evp_launch_zero_sig_off_mask(double *const *sig0, double *const *sig1, const uint8_t *mask, size_t n, hipStream_t st)
{
    SigTab T;
    for (int k = 0; k < 12; ++k) { T.p[k] = sig0[k]; T.p[12 + k] = sig1[k]; }
    hipLaunchKernelGGL(zero_sig_off_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, T, mask, n);
}
