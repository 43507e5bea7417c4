// What HBM bandwidth can a kernel on this MI355X actually reach?  The 8 TB/s in the roofline is the pin rate;
// this measures plain streaming kernels (no arithmetic to speak of) with the access shape of the EVP streaming
// kernel: many separate fp64 arrays, every element touched once, consecutive lanes on consecutive addresses.
//   copy        1 array in, 1 out                      (the classic)
//   read        NIN arrays in, one value out per workgroup
//   write       NOUT arrays out
//   evp-shaped  30 arrays in, 16 out (the array counts of one B-grid subcycle: 368 B per cell)
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_stream.hip -o /tmp/hbm_stream ; run: /tmp/hbm_stream [Mcells]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Tab { const double *in[32]; double *out[16]; };

template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void stream(Tab T, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NIN; ++k) s += T.in[k][i];
    if (NOUT == 0) {
        if (s == 123.456) T.out[0][i] = s;          // never true: keeps the loads
    } else {
#pragma unroll
        for (int k = 0; k < NOUT; ++k) T.out[k][i] = s + k;
    }
}

template <int NIN, int NOUT>
static int run(const char *name, Tab &T, size_t n, hipStream_t st)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n + 255) / 256);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((stream<NIN, NOUT>), dim3(grid), dim3(256), 0, st, T, n);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, st));
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((stream<NIN, NOUT>), dim3(grid), dim3(256), 0, st, T, n);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)(NIN + NOUT) * 8.0 * (double)n;
    printf("HBMSTREAM %-11s %2d in %2d out  %8.1f us/launch  %7.3f TB/s  (%.2f of 8 TB/s)\n", name, NIN, NOUT,
           best * 100.0, bytes / (best * 1e-4) / 1e12, bytes / (best * 1e-4) / 8e12);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t n = (size_t)((argc > 1 ? atof(argv[1]) : 8.64) * 1e6);     // 3600 x 2400 by default
    Tab T{};
    for (auto &p : T.in) { double *q; CHECK(hipMalloc((void **)&q, n * 8)); CHECK(hipMemset(q, 0, n * 8)); p = q; }
    for (auto &p : T.out) { CHECK(hipMalloc((void **)&p, n * 8)); CHECK(hipMemset(p, 0, n * 8)); }
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    printf("HBMSTREAM cells %zu (%.1f MB per array)\n", n, n * 8 / 1e6);
    if (run<1, 1>("copy", T, n, st)) return 1;
    if (run<32, 0>("read", T, n, st)) return 1;
    if (run<0, 16>("write", T, n, st)) return 1;
    if (run<30, 16>("evp-shaped", T, n, st)) return 1;
    if (run<8, 8>("8+8", T, n, st)) return 1;
    return 0;
}
