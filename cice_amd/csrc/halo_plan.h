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
    std::vector<int32_t> send_dst;    // the ghost cell each entry fills, as an offset into the PEER's array
    std::vector<int32_t> recv_dst;
    std::vector<int8_t> recv_sign;
    std::vector<int32_t> recv_gid;    // global cell number (ig-1) + NX*(jg-1) each ghost mirrors (probe exchange)
    std::vector<int8_t> send_sign;    // factor the receiver applies to entry k (= its recv_sign; +1 for raw seam entries)
    // the first n_ghost_* entries of the send / recv lists are ghost cells; what follows (tripole fold row split over ranks)
    // are RAW seam values travelling into staging slots n_local + t of the receiver
    int n_ghost_send = 0, n_ghost_recv = 0;
    // On-chip kernel with the fold row split over ranks: a ghost image of a seam-row cell takes the FINAL (averaged) value
    // straight from the cell's owner, as one more tagged record after the owner's own averaging -- instead of being finalised
    // from raw pair values as the streaming path does.  fimg_*: cells of this rank (seam row) with the ghost cell at the
    // peer they feed and its sign; fimg_recv_*: this rank's ghost cells fed that way, the global column of the seam cell.
    std::vector<int32_t> fimg_src, fimg_dst;
    std::vector<int8_t> fimg_sign;
    std::vector<int32_t> fimg_recv_dst, fimg_recv_col;
    std::vector<int8_t> fimg_recv_sign;
};

struct HaloPlan {
    int nx_block = 0, ny_block = 0, nblocks = 0;
    std::vector<int32_t> local_dst, local_src;   // src = -1: fill with 0
    std::vector<int8_t> local_sign;
    std::vector<HaloPeer> peers;                  // ascending rank
    // Tripole u-fold seam of NE-corner vector fields (ice_boundary.F90:1630-1649,
    // 1689-1722).  The top physical row lies on the fold: U(i,NY) and U(NX-i,NY) are one
    // point, so after every update the pair is replaced by (xavg, -xavg),
    // xavg = 0.5*(x_i - x_{NX-i}), i = 1..NX/2-1; the two pole points i = NX/2, NX mirror
    // onto themselves and change sign.  Offsets into the local array (both cells of a
    // pair must live on this rank).  `late_*`: ghost copies whose source is a seam-row
    // cell; they are repeated after the seam step so that they carry the averaged value.
    std::vector<int32_t> seam_a, seam_b, seam_pole;
    std::vector<int32_t> late_dst, late_src;
    std::vector<int8_t> late_sign;
    // The same step for ANY rank layout (seam pairs, or a ghost cell and the seam cell it mirrors, on different
    // ranks): one entry per cell this rank must finalise after the velocity exchange -- its own seam-row cells and
    // its ghost images of seam-row cells.  x[dst] = coef * 0.5*(x[a] + (-1)*x[b])   (b >= 0: pair average)
    //                                     x[dst] = coef * x[a]                     (b == -1: pole / unpaired cell)
    // a, b are offsets of RAW values of this subcycle: local cells, or staging slots n_local + t that the exchange
    // fills with the raw value of a seam cell of another rank (`tail` slots; they travel as extra entries of the
    // peers' send / recv lists, after the ghost cells).  All reads precede all writes.
    std::vector<int32_t> fin_dst, fin_a, fin_b;
    std::vector<int8_t> fin_coef;
    int tail = 0;
    // Computed identically on EVERY rank (the enumeration below runs over all ranks): some rank's in-loop velocity
    // exchange carries an entry across the tripole fold (sign -1 from another rank) or a staging slot.  Collective
    // decisions (cice_evp_hip_halo_mask) must hang on this, never on a rank's own lists.
    bool any_fold_exchange = false;
    bool stress_remote = false;                   // the stress symmetrisation needs a top-row cell of another rank
    // tripole: which ranks hold the two physical rows next to the fold (NY-1, NY)?  0: none here, 1: all of them here,
    // 2: shared with other ranks (the C-grid fold step then cannot be done on one rank)
    int fold_rows = 0;
    // ice_HaloUpdate_stress (ice_boundary.F90:7441-7826; evp() after the subcycle loop,
    // ice_dyn_evp.F90:1321-1389): ghost row NY+1 of a cell-centre scalar takes the mirrored top
    // physical row of its PARTNER array, a1(ig, NY+1) <- a2(NX-ig+1, NY); ghost cells whose source
    // block was eliminated are set to 0 (src = -1).  Offsets into the local array.
    std::vector<int32_t> stress_dst, stress_src;
    // tripoleT: the same twelve calls rewrite the top PHYSICAL row instead -- a1(ig, NY) <- a2(NX-ig+2, NY), east-west ghost
    // columns of that row included, the ghost row above untouched (T-fold offsets of ice_boundary.F90:7700-7740; pinned on
    // the reference's own arrays, tests/golden/tript_*.npz).  The calls come in pairs (a1, a2), (a2, a1) and the mirror is an
    // involution, so the second call of a pair puts a2's interior values back where they were: what changes is a1 = the
    // arrays _1 and _2 of each family (stress_dst / stress_src above: cell of row NY <- partner array's mirrored cell), and the
    // east-west ghost cells of row NY of _3 and _4, which end up as plain images of their own array (stress_own_*).
    std::vector<int32_t> stress_own_dst, stress_own_src;
    // ... and one cell of the ghost row above: the north-west corner ghost cell of a top-row block (global column ig) takes
    // a2(NX-ig+2, NY-1) -- in BOTH arrays of a pair, each from the other -- unless ig is NX/2 or NX (0).  Found by the geometry
    // sweep against the reference itself (three blocks across the top row are needed to see it) and pinned there on 60 random
    // layouts; nothing else of that row is touched.
    std::vector<int32_t> stress_corner_dst, stress_corner_src;
    // Ghost cells of CELL-CENTRE fields (the T-grid inputs of evp()'s preparation phase,
    // ice_dyn_evp.F90:413-428, 466-470): same as the velocity lists except across the tripole
    // fold, where a centre cell mirrors column NX-ig+1 of row NY-k+1 (ice_boundary.F90:1689-1722,
    // ioffset -1, joffset 0); center_vsign is the factor for vector kinds (-1 across the fold),
    // scalars always copy.  center_remote: some source lives on another rank (not listed).
    std::vector<int32_t> center_dst, center_src;
    std::vector<int8_t> center_vsign;
    bool tfold = false;                           // ns_boundary_type 'tripoleT'
    // tripoleT, cell-centre fields (the preparation phase): the top PHYSICAL row lies on the fold -- made symmetric pair by
    // pair (i <-> NX-i+2 for i = 2..NX/2; i = 1 and NX/2+1 mirror onto themselves) and rewritten from its mirror, east-west
    // ghost columns included; the ghost row above takes row NY-1 at column NX-i+2 (ice_boundary.F90:1563-1583, 1686-1722).
    // As entries of a two-pass fold step (evp_device.h: EvpCgFoldList): x[dst] = s * 0.5*(x[a] + isign*x[b]) or s * x[a]
    // (b = -1), s = flip ? isign : 1.  center_tf_remote: an operand lives on another rank or in an eliminated block (the
    // device preparation then stays with the host).  The lists above hold the other ghost cells (rows below NY).
    std::vector<int32_t> center_tf_dst, center_tf_a, center_tf_b;
    std::vector<uint8_t> center_tf_flip;
    bool center_tf_remote = false;
    bool center_remote = false;
    bool center_fold_remote = false;              // ... and one of them lies across the tripole fold (centre mirror rule, other rank)
    // Centre-field ghost cells of this rank whose source lies across the fold on ANOTHER rank (fold row split in x), and the
    // same for the stress symmetrisation.  They are filled through the ordinary NE-corner exchange run on a SHIFTED copy:
    // with a'(i, NY-1) = a(i+1, NY) on the owner, the corner rule ghost(ig, NY+1) <- -a'(NX-ig, NY-1) delivers
    // -a(NX-ig+1, NY), the centre rule's source (sign fixed afterwards).  fold_shift_cells: this rank's interior cells of
    // row NY-1 (the shifted copy is built there).
    std::vector<int32_t> center_foldr_dst, stress_foldr_dst, fold_shift_cells;
    // ... and this rank's east-west ghost cells of row NY whose source another rank owns: the velocity exchange does not
    // fill them (images of seam-row cells are finalised from RAW values), but it brings the raw value of the source into
    // a staging slot -- which for a cell-centre field IS the ghost value: a[dst] = a[slot] after the exchange
    std::vector<int32_t> center_seam_dst, center_seam_slot;
    bool fold_split = false;                      // the blocks holding row NY have more than one owner: the same on every rank
                                                  // (the exchanges above are collective)
    std::string error;
};

// Returns false and sets plan.error on inconsistent input.
bool build_halo_plan(const cice_evp_hip_dims &d, HaloPlan &plan);

// Window table of the C grid's one-launch kernel (evp_cgrid.hip: cg_one; host only).  Windows of ox x oy positions, the
// inner (ox-3) x (oy-3) owned cells, cover every block's interior row by row (in strips of `strip` windows in x).  Per
// window 4 ints in `tiles` -- block, first owned i, first owned j (1-based, array numbering), 1 if the window is regular
// (every position an array cell of that block and its own source) -- and ox*oy entries in `tab`: for the position
// (tx, ty) = cell (i0-2+tx, j0-2+ty) of the block's numbering, which may lie outside its array, the cell whose value the
// reference has there: >= 0 an interior cell (itself, or the one a ghost cell mirrors according to the plan's local
// copies; further out the walk continues from the mirrored cell, neighbour by neighbour, x first), or -1 - c for a ghost
// cell c nothing is copied into (closed boundary, eliminated neighbour block): its arrays are read, never computed.
// extra = 1: (ox+1) x (oy+1) positions per window, same owned range and window stride (the resident kernel's velocity tile).
void build_window_table(const cice_evp_hip_dims &d, const HaloPlan &plan, int ox, int oy, int strip, std::vector<int32_t> &tiles,
                        std::vector<int32_t> &tab, int extra = 0);

// The same for a tripole (u-fold) grid (17 x 17 positions): the top window row of the blocks at the fold carries a mirrored
// mini-tile in source orientation above the fold row; see halo_plan.cpp.  tiles2: (G0, NX, 0, 0) per window.
bool build_fold_window_table(const cice_evp_hip_dims &d, const HaloPlan &plan, std::vector<int32_t> &tiles, std::vector<int32_t> &tiles2,
                             std::vector<int32_t> &tab, std::string &why);

// Which positions of a resident window's 17 x 17 velocity tile are hand-offs (evp_cgrid_res.hip polls them, their owner publishes
// them): not owned, with a producing cell, and within reach of the window's owned cells -- at most CGRES_REACH positions beyond
// the last owned column and row (an owned cell's divergence reads level U beside it, that level T one further, that the velocities
// one further again: two; three is the kernel's own margin).  Everything further out is worked out from whatever the tile was filled
// with and read by nobody.  Without the bound a narrow window at a block's edge (one or two owned columns) polled up to 14 columns
// into its neighbour and beyond -- cells of a window that does not poll IT: that window could run two subcycles ahead and overwrite
// the record slot the narrow one was still waiting for (round-5 advice).  Fold windows keep every position (their mirrored mini-tile
// runs against the column index).
constexpr int CGRES_REACH = 3;
constexpr int CGRES_SLOTS = 4;             // record slots per cell, by subcycle modulo (evp_device.h: EVP_CGRES_SLOTS)
inline bool cgres_in_reach(int ex, int ey, int last_ex, int last_ey, bool foldwin)
{
    return foldwin || (ex <= last_ex + CGRES_REACH && ey <= last_ey + CGRES_REACH);
}
// ---- the marched C-grid kernel's share of a rank (evp_cgrid.hip: cg_strip) -- host only, CPU-tested ----
// A rectangle of a block that the regular windows of a window table cover (regular: every position an interior cell of the block,
// its own source): first owned column / row of its first and last window column / row.
struct StripZone { int b, i0, i1, j0, j1; };
// per block the rectangle of its regular windows, if they form one, it is at least a strip wide and none of its cells has a ghost
// image (img_slot: per cell, < 0 = none; may be null)
void strip_zones(const cice_evp_hip_dims &d, const std::vector<int32_t> &tiles, int ex, int ey, const int *img_slot, std::vector<StripZone> &zones);
// work items of the rectangles, x 6 ints each: block, column of lane 2, first and last owned row, first and last owned lane.  lo0: first lane
// that may own a column (2; 3 where the kernel forms the lengths), the last is 61; strips of 62 - lo0 columns, the last one shifted west
// so that lane 62 stays inside the rectangle + 1; segments of `seg` rows (0: the fewest rows >= seg_min with at most `slots` items).
// Returns the rows per segment.
int strip_items(const std::vector<StripZone> &zones, int ex, int ey, int lo0, long slots, int seg_min, int seg, std::vector<int32_t> &items);
// 1 for every window of `tiles` that lies inside one of the rectangles (the marched kernel owns its cells), 0: cg_one keeps it
void strip_windows(const std::vector<StripZone> &zones, const std::vector<int32_t> &tiles, std::vector<uint8_t> &in_zone);

// The hand-off graph of the resident windows (tiles / tab as build_window_table(..., 16, 16, ., extra = 1) or
// build_fold_window_table made them): window w READS window p when it polls a cell p owns.  A window cannot start subcycle j + 1
// before every window it reads has finished subcycle j, so a window p is never more than len subcycles ahead of w, len = the
// shortest chain p reads ... reads w.  The exact-tag record protocol with CGRES_SLOTS slots per cell is safe for the hand-off
// w reads p iff that chain is at most CGRES_SLOTS - 1 long (p's record of subcycle j is overwritten by that of j + CGRES_SLOTS):
// 1 when the hand-off is mutual -- nearly all are --, 2 or 3 for the one-way ones of narrow windows and of fold windows whose
// mirror images do not line up.  pub (may be NULL): [ncell] 1 = the cell is polled by some window, i.e. its owner publishes it.
// Returns the number of UNSAFE hand-offs (no chain back within CGRES_SLOTS - 1); *n_edges, *n_oneway (may be NULL): hand-offs in
// all, and those that are not mutual.
int cgres_dependencies(const cice_evp_hip_dims &d, bool tripole, const std::vector<int32_t> &tiles, const std::vector<int32_t> &tab,
                       std::vector<uint8_t> *pub, int *n_edges, int *n_oneway);

// C grid on a tripole (u-fold) grid: the fold step of one field location (0 centre, 1 NE corner, 2 E face, 3 N face),
// by the meaning of the cells (ice_boundary.F90:1626-1722): one entry for every cell of every local block -- interior
// or ghost -- in the top physical row NY (locations with points ON the fold: NE corner, N face) or in the ghost row
// NY+1 (every location).    x[dst] = s * 0.5*(x[a] + isign*x[b])   b >= 0 (or -2: partner's block eliminated, 0)
//                           x[dst] = s * x[a]                      b == -1 (a == -1: source eliminated, 0)
// s = flip ? isign : 1; isign = -1 for vector kinds.  Sources are interior cells of THIS rank (the blocks holding rows
// NY-1 and NY must all be local).  Host only.
struct FoldList {
    std::vector<int32_t> dst, a, b;
    std::vector<uint8_t> flip;
};
void build_fold_list(const cice_evp_hip_dims &d, int loc, FoldList &L);
