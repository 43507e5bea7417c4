// Host-side halo plan: which ghost cell of which local block takes its value
// from where.  Pure C++ (no HIP) so that it can be built and tested on a
// machine without a GPU.
//
// Replaces, for the one field pair the subcycle exchanges (uvel,vvel at NE
// corners, vector kind), what ice_HaloCreate precomputes as address lists
// (infrastructure/comm/mpi/ice_boundary.F90:171-885, type ice_halo :88-126):
//   srcLocalAddr/dstLocalAddr -> local (dst, src, sign) lists
//   sendAddr / recvAddr       -> per-peer send_src / recv_dst lists
// Instead of walking block neighbours direction by direction the plan is
// derived from the *meaning* of a ghost cell: it mirrors the interior cell
// that holds the same global (i,j) (cyclic wrap, tripole fold), or nothing
// (closed/open outer boundary: left untouched, as ice_HaloUpdate does when no
// fillValue is given, :1176-1189).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/cice_evp_hip.h"

struct HaloBlock {
    int gi0, gj0, gnx, gny;   // interior rectangle in global index space (1-based origin)
    int owner, local;         // owning rank (-1: eliminated land block), local block index
};

struct HaloPeer {
    int rank = -1;
    std::vector<int32_t> send_src;    // offsets into the local (nx,ny,nblocks) array
    std::vector<int32_t> recv_dst;
    std::vector<int8_t> recv_sign;
};

struct HaloPlan {
    int nx_block = 0, ny_block = 0, nblocks = 0;
    std::vector<int32_t> local_dst, local_src;   // src = -1: fill with 0
    std::vector<int8_t> local_sign;
    std::vector<HaloPeer> peers;                  // ascending rank
    // tripole u-fold seam of NE-corner vector fields: pairs averaged with sign flip
    // (ice_boundary.F90:1630-1649); offsets into the local array, both local.
    std::vector<int32_t> seam_a, seam_b;
    std::string error;
};

// Returns false and sets plan.error on inconsistent input.
bool build_halo_plan(const cice_evp_hip_dims &d, HaloPlan &plan);
