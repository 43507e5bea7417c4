// =====================================================================
// Velocity halo between GPUs of one node without a communication library:
// the "mailbox" exchange.
//
// Replaces, per subcycle, pack -> ncclSend/ncclRecv -> unpack (and with them
// ice_HaloUpdate's MPI round, ice_boundary.F90:1221-1449) by ONE kernel of one
// workgroup that
//   1. stores the velocities other ranks mirror straight into those ranks'
//      inboxes -- peer memory mapped through HIP IPC, i.e. plain stores that
//      travel over xGMI -- with system-scope write-through stores,
//   2. release-fences and raises its sequence number in every peer's flag slot,
//   3. waits (bounded) until every peer's sequence number has reached its own,
//   4. copies its inbox into the ghost cells (sign applied, as the unpack does).
// The inbox is double buffered by sequence parity; a rank can be at most one
// exchange ahead of a neighbour (it needs that neighbour's flag to finish an
// exchange), so parity p is never overwritten before it has been read.
// No host involvement: the kernel is captured into the subcycle hipGraph.
// =====================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evp_device.h"
#include "evp_halo_direct.h"

namespace {

__global__ __launch_bounds__(1024) void halo_direct_uv(EvpDirect D, double *__restrict__ u, double *__restrict__ v)
{
    evp_mailbox::exchange(D, u, v, threadIdx.x, blockDim.x, nullptr, 0u);
}

}  // namespace

void evp_launch_halo_direct(const EvpDirect &D, double *u, double *v, hipStream_t st)
{
    hipLaunchKernelGGL(halo_direct_uv, dim3(1), dim3(1024), 0, st, D, u, v);
}
