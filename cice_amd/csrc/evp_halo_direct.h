// Device body of the mailbox halo exchange (see evp_halo_direct.hip for the protocol),
// shared by the stand-alone kernel and by the exchange workgroup that rides inside the
// subcycle launch (evp_kernels.hip).  Executed by ONE workgroup of `nthr` threads.
#pragma once
#include <hip/hip_runtime.h>

#include "evp_device.h"

namespace evp_mailbox {

__device__ __forceinline__ void st_sys(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ld_sys(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// `wait_count`/`wait_target`: when non-null, first wait (bounded) until the producers of the
// cells to send -- the boundary tiles of the same launch -- have all checked in.
__device__ __forceinline__ void exchange(const EvpDirect &D, double *__restrict__ u, double *__restrict__ v,
                                         int tid, int nthr, const unsigned *wait_count, unsigned wait_target)
{
    const unsigned s = *D.seq + 1u;          // this exchange's sequence number (same on every rank)
    const unsigned par = s & 1u;
    const bool dead = *D.err != 0;           // an earlier wait gave up: do not wait again

    if (wait_count) {
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            while (!dead) {
                const unsigned have = __hip_atomic_load(wait_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(have - wait_target) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 && wall_clock64() - t0 > D.timeout_ticks) {
                    atomicCAS(D.err, 0, 1000);           // own boundary tiles never checked in
                    break;
                }
            }
        }
        __syncthreads();
    }

    // 1. remote stores into the peers' inboxes; UNR entries per thread in flight at a time
    constexpr int UNR = 4;
    for (int base = 0; base < D.n_send; base += nthr * UNR) {
        int src[UNR];
        double *dst[UNR];
        double uu[UNR], vv[UNR];
#pragma unroll
        for (int e = 0; e < UNR; ++e) {
            const int k = base + e * nthr + tid;
            src[e] = -1;
            if (k < D.n_send) {
                src[e] = D.send_src[k];
                dst[e] = D.send_addr[k] + (size_t)par * D.send_pstride[k];
            }
        }
#pragma unroll
        for (int e = 0; e < UNR; ++e)
            if (src[e] >= 0) { uu[e] = ld_sys(u + src[e]); vv[e] = ld_sys(v + src[e]); }
#pragma unroll
        for (int e = 0; e < UNR; ++e)
            if (src[e] >= 0) { st_sys(dst[e], uu[e]); st_sys(dst[e] + 1, vv[e]); }
    }
    if (!(D.dbg & 1)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope
    __syncthreads();
    // 2. + 3. raise my number at the peers, wait for theirs
    for (int q = tid; q < D.npeers; q += nthr) {
        __hip_atomic_store(D.peer_flag[q], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned *f = D.flags_in + (size_t)q * EVP_DIRECT_FLAG_STRIDE;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (!dead) {
            const unsigned have = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(have - s) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && wall_clock64() - t0 > D.timeout_ticks) {
                atomicCAS(D.err, 0, 1 + q);              // which peer never arrived
                break;
            }
        }
    }
    __syncthreads();
    if (!(D.dbg & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // 4. inbox -> ghost cells
    const double *in = D.inbox + (size_t)par * 2 * (size_t)D.n_recv_slots;
    for (int base = 0; base < D.n_recv; base += nthr * UNR) {
        int d[UNR];
        double sg[UNR], uu[UNR], vv[UNR];
#pragma unroll
        for (int e = 0; e < UNR; ++e) {
            const int k = base + e * nthr + tid;
            d[e] = -1;
            if (k < D.n_recv) {
                d[e] = D.recv_dst[k];
                sg[e] = (double)D.recv_sign[k];
                const size_t slot = D.recv_slot ? (size_t)D.recv_slot[k] : (size_t)k;
                uu[e] = ld_sys(in + 2 * slot);
                vv[e] = ld_sys(in + 2 * slot + 1);
            }
        }
#pragma unroll
        for (int e = 0; e < UNR; ++e)
            if (d[e] >= 0) { u[d[e]] = sg[e] * uu[e]; v[d[e]] = sg[e] * vv[e]; }
    }
    if (tid == 0) *D.seq = s;      // every thread read seq before the first barrier
}

}  // namespace evp_mailbox
