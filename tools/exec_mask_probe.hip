// Does a wave64 fp64 instruction finish sooner when only 16 (or 32) of its lanes are active?  One wave per SIMD-free CU,
// a dependent chain of v_fma_f64 (and of v_rcp_f64 / v_sqrt_f64), shader-clock cycles per instruction for EXEC = 64, 32, 16
// and 1 active lanes.  hipcc --offload-arch=gfx950 -O3 tools/exec_mask_probe.hip -o /tmp/exec_mask_probe && /tmp/exec_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(double *out, unsigned long long *cyc, int nact, int kind)
{
    double x = 1.0 + threadIdx.x * 1e-9, a = 1.0000001, b = 1e-9;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < nact) {
        t0 = __builtin_readcyclecounter();
        if (kind == 0) {
#pragma unroll 1
            for (int k = 0; k < 1000; ++k) {
#pragma unroll
                for (int u = 0; u < 16; ++u) x = __builtin_fma(x, a, b);
            }
        } else {
#pragma unroll 1
            for (int k = 0; k < 1000; ++k) {
#pragma unroll
                for (int u = 0; u < 16; ++u) x = __builtin_amdgcn_rcp(x) + 0.5;
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    for (int kind = 0; kind < 2; ++kind)
        for (int nact : {64, 48, 32, 16, 1}) {
            chain<<<1, 64>>>(out, cyc, nact, kind);
            chain<<<1, 64>>>(out, cyc, nact, kind);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s active lanes %2d: %.2f cycles per dependent instruction%s\n", kind ? "v_rcp_f64 + v_add_f64" : "v_fma_f64", nact, (double)h / 16000.0, kind ? " pair" : "");
        }
    return 0;
}
