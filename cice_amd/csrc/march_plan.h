// Host-only geometry and exchange plan of the several-subcycles-per-pass path (evp_march.hip) -- pure C++, no HIP, so
// that it can be built and tested on a machine without a GPU (tests/test_multirank_cpu.py).
//
// Every rank's sub-domain must be ONE rectangle of the global index space (CICE's cartesian distributions:
// ice_distribution.F90 create_distrb_cart; any number of blocks per rank as long as they tile a rectangle).  The rank
// holds it in the strip-major layout of evp_host_march.cpp: strips of `own` columns, per (row, strip) a block of 64
// lanes -- with P = MARCH_PLAN_PAD (4): lanes P .. P+own-1 own their columns, lanes 0 .. P-1 and P+own .. 2P+own-1 duplicate
// the neighbouring strips' edge columns (or, for the first / last strip, hold the P halo columns beyond the rectangle).
//
// A pass of the marching kernel that advances the state by k <= P subcycles needs it k cells beyond the rectangle on
// every side.  What replaces the reference's ice_HaloUpdate there (ice_boundary.F90:1066-1760; one-cell ring, every
// subcycle) is one exchange of a P-cell ring every few passes: like build_halo_plan, the lists are derived from the meaning of a halo cell --
// it images the cell with the same global index (cyclic wrap) -- by every rank for every rank, in one canonical
// order, so that sender and receiver agree without any set-up communication.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/cice_evp_hip.h"

#ifdef EVP_MARCH_PAD
#define MARCH_PLAN_PAD EVP_MARCH_PAD
#else
#define MARCH_PLAN_PAD 4       // == EVP_MARCH_PAD (evp_device.h; evp_host_march.cpp asserts it)
#endif

struct MarchRect {
    int gx0 = 0, gy0 = 0;      // global index (0-based) of the first owned cell
    int nxr = 0, nyr = 0;      // owned cells
    int own = 0, nstrips = 0;
    bool ok = false;
};

struct MarchPeer {
    int rank = -1;
    // position of a cell in a rank's strip-major buffers: (storage row * nstrips + strip) * 64 + lane; element of field f
    // in a buffer with NF fields per block: ((pos >> 6) * NF + f) * 64 + (pos & 63)
    std::vector<int32_t> send_pos;               // in MY layout: the owner's position of the cell
    std::vector<int32_t> send_col;               // its column x (for the row-major byte mask), row = (pos >> 6) / nstrips
    std::vector<int32_t> recv_pos1, recv_pos2;   // in MY layout: where the value goes; pos2 = its duplicate or -1
    std::vector<int32_t> recv_col, recv_row;     // column x / storage row of the halo cell (byte mask)
};

struct MarchPlan {
    MarchRect me;                                // the rectangle this rank HOLDS: its own cells plus `ext` more on every side that
                                                 // has a neighbour (computed redundantly, so that the ring is exchanged less often)
    MarchRect owned;                             // the cells this rank owns (gx0, gy0, nxr, nyr)
    int ext_w = 0, ext_e = 0, ext_s = 0, ext_n = 0;
    std::vector<MarchRect> all;                  // every rank's OWN rectangle (index = rank; ok = false: rank holds nothing)
    bool wrapx = false;                          // E-W cyclic wrap handled inside the rank (it spans the whole dimension)
    std::vector<int32_t> dup;                    // [nstrips][64]: (strip << 8) | lane of the duplicate of an owner lane's column, -1 none
    std::vector<MarchPeer> peers;                // ascending rank; may contain this rank itself (self-exchange across the seam)
    int n_send = 0, n_recv = 0;
    std::string error;                           // non-empty: this domain cannot use the path (the same verdict on every rank)
};

// own_max: widest strip (<= 64 - 2P = 56); wrap_inside: let a rank that spans a cyclic E-W dimension wrap internally
// (false: the seam is exchanged like any other rank boundary -- with the rank itself; a test hook).
// ext (even, >= 0): every rank also holds -- and advances redundantly -- `ext` cells beyond its own on every side that
// has a neighbour.  One exchange then brings the ring of ext + P cells around the rank's own cells up to date, and
// passes advancing ext + P subcycles in all can follow before the next one: every subcycle costs one cell of validity.
bool build_march_plan(const cice_evp_hip_dims &d, int own_max, bool wrap_inside, int ext, MarchPlan &P);
