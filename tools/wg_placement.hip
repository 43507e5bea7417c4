// Where does the hardware put the workgroups of a persistent launch?  Same shape as the resident
// EVP kernel (256 threads, ~44 KB LDS, all workgroups alive together): prints how many workgroups
// each CU received and which launch indices share a CU.   hipcc --offload-arch=gfx950 -O2 wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 3) void probe(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = hw;
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 572;
    unsigned *d; hipMalloc(&d, 2 * n * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 44 * 1024, 0, d, 20000);   // 200 us
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int w = 0; w < n; ++w) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID layout
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(w);
    }
    std::map<size_t, int> hist;
    for (auto &kv : cu) hist[kv.second.size()]++;
    printf("workgroups %d on %zu distinct CUs;", n, cu.size());
    for (auto &kv : hist) printf(" %d CUs x %zu", kv.second, kv.first);
    printf("\n");
    int shown = 0;
    for (auto &kv : cu)
        if (kv.second.size() >= 3 && shown++ < 6) {
            printf("  cu %05x:", kv.first);
            for (int w : kv.second) printf(" %d", w);
            printf("\n");
        }
    for (auto &kv : cu)
        if (shown++ < 12) {
            printf("  cu %05x:", kv.first);
            for (int w : kv.second) printf(" %d", w);
            printf("\n");
        }
    return 0;
}
