// Known byte counts in the streaming EVP kernel's access width (one fp64 = 8 B per lane, 512 B per
// wave and row), far larger than the 256 MB Infinity Cache: what rocprofv3's FETCH_SIZE / WRITE_SIZE
// report for them calibrates the counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a wide
// coalesced stream on gfx950; other widths and WRITE_SIZE are to be calibrated in one's own pattern).
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib && /tmp/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read8(const double *__restrict__ a, double *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += a[i];
    if (s == 1.2345e300) out[0] = s;          // keeps the loads alive, never true
}
__global__ void calib_write8(double *__restrict__ a, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = (double)i;
}
__global__ void calib_copy8(const double *__restrict__ a, double *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i] + 1.0;
}
int main()
{
    const size_t n = (size_t)1 << 27;         // 1 GiB per array
    double *a, *b;
    (void)hipMalloc(&a, n * 8); (void)hipMalloc(&b, n * 8);
    (void)hipMemset(a, 0, n * 8); (void)hipMemset(b, 0, n * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, a, b, n); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("CALIB calib_read8 bytes_read %zu bytes_written 0 ms %.4f TB/s %.3f\n", n * 8, ms, n * 8 / ms / 1e9);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, a, n); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("CALIB calib_write8 bytes_read 0 bytes_written %zu ms %.4f TB/s %.3f\n", n * 8, ms, n * 8 / ms / 1e9);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(calib_copy8, dim3(4096), dim3(256), 0, 0, a, b, n); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("CALIB calib_copy8 bytes_read %zu bytes_written %zu ms %.4f TB/s %.3f\n", n * 8, n * 8, ms, 2 * n * 8 / ms / 1e9);
    }
    return 0;
}
