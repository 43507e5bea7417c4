// Measurement aid: what does a grid-wide barrier inside ONE persistent launch cost on this GPU, against a kernel boundary
// inside a captured graph?  (The C-grid subcycle is three dependent launches of ~8 us on gx1: would one persistent launch
// with three barriers per subcycle be faster?)  Each "phase" reads two neighbouring workgroups' previous output and
// writes 10 x 256 doubles per workgroup (about what a C-grid phase stores per 64x4 tile), then synchronises:
//   mode 0: one kernel launch per phase, captured in a graph;   mode 1: one launch, grid barrier between phases
// (monotonic counter, agent-scope release / acquire -- the L2 write-back and invalidate a kernel boundary also does).
// Every spin is bounded: a barrier that does not complete sets an error word instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp [nwg] [nphase]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NARR = 10;

__device__ __forceinline__ void phase_body(double *const *a, int nwg, int wg, int tid, int ph)
{
    const int l = (wg + nwg - 1) % nwg, r = (wg + 1) % nwg;
    const double *src = a[(ph + NARR - 1) % NARR];
    const double x = src[(size_t)l * 256 + tid] + src[(size_t)r * 256 + tid] + src[(size_t)wg * 256 + tid];
#pragma unroll
    for (int k = 0; k < NARR; ++k) a[(ph + k) % NARR][(size_t)wg * 256 + tid] = 0.25 * x + k;
}

struct Arr { double *p[NARR]; };

__global__ void one_phase(Arr A, int nwg, int ph) { phase_body(A.p, nwg, blockIdx.x, threadIdx.x, ph); }

__global__ void persistent(Arr A, int nwg, int nphase, unsigned *ctr, unsigned *err)
{
    const int wg = blockIdx.x, tid = threadIdx.x;
    for (int ph = 0; ph < nphase; ++ph) {
        phase_body(A.p, nwg, wg, tid, ph);
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(ph + 1) * (unsigned)nwg;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 2000000) { atomicExch(err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (*(volatile unsigned *)err) return;
    }
}

int main(int argc, char **argv)
{
    const int nwg = argc > 1 ? std::atoi(argv[1]) : 480, nphase = argc > 2 ? std::atoi(argv[2]) : 360;
    Arr A;
    for (int k = 0; k < NARR; ++k) { CK(hipMalloc((void **)&A.p[k], (size_t)nwg * 256 * 8)); CK(hipMemset(A.p[k], 0, (size_t)nwg * 256 * 8)); }
    unsigned *ctr, *err;
    CK(hipMalloc((void **)&ctr, 4)); CK(hipMalloc((void **)&err, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int maxb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, persistent, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    std::printf("workgroups %d, phases %d, co-resident capacity %d x %d CUs\n", nwg, nphase, maxb, prop.multiProcessorCount);
    if (nwg > maxb * prop.multiProcessorCount) { std::printf("too many workgroups to be co-resident\n"); return 1; }
    // mode 0: graph of nphase launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int ph = 0; ph < nphase; ++ph) hipLaunchKernelGGL(one_phase, dim3(nwg), dim3(256), 0, st, A, nwg, ph);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("graph of launches : %.2f us per phase\n", 1e3 * ms / nphase);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(persistent, dim3(nwg), dim3(256), 0, st, A, nwg, nphase, ctr, err);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        std::printf("persistent+barrier: %.2f us per phase%s\n", 1e3 * ms / nphase, h ? "  (A BARRIER GAVE UP)" : "");
    }
    return 0;
}
