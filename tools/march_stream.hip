// Does the data layout limit the marching kernel's HBM rate?  Same work distribution as evp_march.hip (one wave per
// strip of 64 columns x SEG rows, marching north, NIN doubles in and NOUT out per cell and row, next to no arithmetic):
//   soa     NIN + NOUT separate arrays, a wave touches 512 contiguous bytes of each per row (the kernel's layout)
//   packed  strip-major blocks: per (row, strip) one contiguous block of NIN x 64 doubles in, NOUT x 64 out
// build: hipcc --offload-arch=gfx950 -O3 tools/march_stream.hip -o /tmp/march_stream ; run: /tmp/march_stream [seg] [waves_per_wg]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NIN = 27, NOUT = 14, NX = 3600, NY = 2400, NSTRIP = 60, LDX = 3848;
struct Tab { const double *in[NIN]; double *out[NOUT]; const double *pin; double *pout; int seg, nseg; };

template <bool PACKED, int DEPTH>
__global__ __launch_bounds__(256) void march(Tab T)
{
    const int lane = threadIdx.x & 63, item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= NSTRIP * T.nseg) return;
    const int strip = item % NSTRIP, seg = item / NSTRIP;
    const int y0 = seg * T.seg, y1 = min(y0 + T.seg, NY);
    double carry = 0.0;
    for (int y = y0; y < y1; ++y) {
        double s = carry;
        if (PACKED) {
            const double *b = T.pin + ((size_t)y * NSTRIP + strip) * (NIN * 64) + lane;
#pragma unroll
            for (int k = 0; k < NIN; ++k) s += b[k * 64];
            double *o = T.pout + ((size_t)y * NSTRIP + strip) * (NOUT * 64) + lane;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) o[k * 64] = s + k;
        } else {
            const size_t e = (size_t)y * LDX + strip * 64 + lane;
#pragma unroll
            for (int k = 0; k < NIN; ++k) s += T.in[k][e];
#pragma unroll
            for (int k = 0; k < NOUT; ++k) T.out[k][e] = s + k;
        }
        carry = s * 1e-30;
    }
}

template <bool PACKED>
static int run(const char *name, Tab &T, hipStream_t st)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((NSTRIP * T.nseg + 3) / 4);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((march<PACKED, 1>), dim3(grid), dim3(256), 0, st, T);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, st));
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((march<PACKED, 1>), dim3(grid), dim3(256), 0, st, T);
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)(NIN + NOUT) * 8.0 * 64.0 * NSTRIP * NY;
    printf("MARCHSTREAM %-7s seg %3d (%4d waves)  %8.1f us/launch  %6.3f TB/s\n", name, T.seg, NSTRIP * T.nseg, best * 200.0,
           bytes / (best * 2e-4) / 1e12);
    return 0;
}

int main(int argc, char **argv)
{
    Tab T{};
    const size_t n = (size_t)LDX * NY;
    for (auto &p : T.in) { double *q; CHECK(hipMalloc((void **)&q, n * 8)); CHECK(hipMemset(q, 0, n * 8)); p = q; }
    for (auto &p : T.out) { CHECK(hipMalloc((void **)&p, n * 8)); CHECK(hipMemset(p, 0, n * 8)); }
    double *q;
    CHECK(hipMalloc((void **)&q, (size_t)NY * NSTRIP * NIN * 64 * 8)); CHECK(hipMemset(q, 0, (size_t)NY * NSTRIP * NIN * 64 * 8)); T.pin = q;
    CHECK(hipMalloc((void **)&T.pout, (size_t)NY * NSTRIP * NOUT * 64 * 8));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int seg : {24, 48, 72}) {
        T.seg = seg; T.nseg = (NY + seg - 1) / seg;
        if (run<false>("soa", T, st)) return 1;
        if (run<true>("packed", T, st)) return 1;
    }
    return 0;
}
