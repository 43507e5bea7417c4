// Measurement aid: how many SIMD cycles does one wave64 fp64 VALU instruction occupy on this GPU, per opcode, and do
// two / four waves on one SIMD overlap?  (The EVP kernels are strict fp64 without FMA contraction: v_mul_f64 and
// v_add_f64 dominate; the roofline's "VALU-busy" counter SQ_ACTIVE_INST_VALU assumes 4 cycles per instruction.)
// Each thread runs N instructions on 8 independent register chains (a chain's next instruction comes 8 instructions later); grid = waves_per_simd x 4 x 256 CUs
// waves of 64.  Prints cycles per instruction per SIMD (shader clock from s_memtime / wall time).
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_rate_probe.hip -o /tmp/fp64p && /tmp/fp64p
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void chain(double *out, int iters, double a0, double b0)
{
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = a0 + k + threadIdx.x * 1e-9;
    const double b = b0, c = 1e-30;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 128; ++kk) {          // 128 instructions per loop trip: the branch does not count
            const int k = kk & 7;
            if (OP == 0) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(b));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[k]));
            if (OP == 4) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x[k]));
            if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(*(float *)&x[k]) : "v"(1.0f));
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    if (s == 123.456) out[0] = s;
}

template <int OP>
int run(const char *name, int waves_per_simd)
{
    double *out; CK(hipMalloc((void **)&out, 8));
    const int iters = 2000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const dim3 grid(ncu * waves_per_simd), block(256);           // 4 waves per workgroup = one per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain<OP>, grid, block, 0, 0, out, 100, 1.0, 1.0000001);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain<OP>, grid, block, 0, 0, out, iters, 1.0, 1.0000001);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = (double)iters * 128 * waves_per_simd;
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    std::printf("%-10s waves/SIMD %d: %.3f ns per wave-instruction per SIMD = %.2f cycles at %.2f GHz (nominal)\n", name, waves_per_simd,
                1e6 * ms / insts_per_simd, 1e6 * ms / insts_per_simd * clk_khz * 1e-6, clk_khz * 1e-6);
    CK(hipFree(out));
    return 0;
}

int main()
{
    for (int w : {1, 2, 4}) {
        if (run<0>("v_mul_f64", w) || run<1>("v_add_f64", w) || run<2>("v_fma_f64", w)) return 1;
    }
    for (int w : {1, 2}) {
        if (run<3>("v_rcp_f64", w) || run<4>("v_sqrt_f64", w) || run<5>("v_mul_f32", w)) return 1;
    }
    return 0;
}
