/* =====================================================================
 * cice_evp_hip_testing.h -- entry points of the TEST build only.
 *
 * libcice_evp_hip_testing.so = the sources of libcice_evp_hip.so compiled with -DCICE_EVP_HIP_TESTING: the same
 * kernels and host code, plus what tests and tools need and a production host must never find by accident:
 *   - host-only introspection of the halo / seam / marching / window plans (CPU known-answer tests),
 *   - read-outs of the resident kernel's per-CU placement record and phase stamps (tools/),
 *   - a test transport that routes the marching path's exchanges through host callbacks (several ranks as
 *     processes sharing ONE GPU, where RCCL refuses to run),
 *   - the experiment / fault-injection environment switches (evp_host.h: env_test) -- in the production build they
 *     read as unset.
 * The production ABI is include/cice_evp_hip.h.
 * ===================================================================== */
#ifndef CICE_EVP_HIP_TESTING_H
#define CICE_EVP_HIP_TESTING_H

#include "cice_evp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host only (no device needed): the fold step of field location `loc` (0 centre, 1 NE corner, 2 E face, 3 N face) on a
 * tripole grid -- x[dst] = s*0.5*(x[a] + isign*x[b]) (b >= 0; -2: partner eliminated) or s*x[a] (b == -1), s = flip ? isign : 1.
 * First call with NULL lists for the count.                                                                       */
int cice_evp_hip_cgrid_fold_plan(const cice_evp_hip_dims *dims, int32_t loc, int32_t *count, int32_t *dst, int32_t *a,
                                 int32_t *b, int32_t *flip);
/* Host only (no device needed): the window table of the C grid's one-launch kernel for windows of ox x oy positions --
 * per window {block, first owned i, first owned j (1-based), regular} in tiles4 and ox*oy entries in tab: the cell whose
 * value the reference has at that position (>= 0), or -1 - c for a ghost cell c nothing is copied into.  First call with
 * NULL arrays for the count.  (What the device kernel reads; checked on the CPU against the decomposition's global numbering.) */
int cice_evp_hip_cgrid_window_plan(const cice_evp_hip_dims *dims, int32_t ox, int32_t oy, int32_t *ntiles, int32_t *tiles4, int32_t *tab);
/* The same with (ox + extra) x (oy + extra) positions per window, extra = 0 or 1 (same owned range and window stride): the table of the
 * on-chip resident C-grid kernel, whose level S reads one position beyond the window to the east and north (evp_cgrid_res.hip).
 * extra = 2: that kernel's table on a tripole (u-fold) grid, 17 x 17 positions (ox, oy unused): the windows at the fold own up to
 * 11 rows and carry a mirrored mini-tile above the fold row, tiles4[3] = fold flag | tf << 8 | last owned row << 16 (halo_plan.cpp:
 * build_fold_window_table); -5 when a mirrored cell is not on this rank.                                                          */
int cice_evp_hip_cgrid_window_plan_ext(const cice_evp_hip_dims *dims, int32_t ox, int32_t oy, int32_t extra, int32_t *ntiles,
                                       int32_t *tiles4, int32_t *tab);
/* Host only: the hand-off graph of the on-chip resident C-grid kernel's windows (closed / cyclic grids: the table of extra = 1;
 * tripole: of extra = 2).  A window depends on another when it polls a cell that one owns -- every non-owned position with a
 * producer within 3 positions of its last owned column / row (halo_plan.h: cgres_in_reach; fold windows: every position).  n_edges:
 * dependencies, n_oneway: those that are not mutual, n_unsafe: those of them with no chain of at most 3 dependencies back (the
 * window read could then be four subcycles ahead of its reader and overwrite one of the kernel's four record slots per cell that
 * the reader still waits for): the library does not use the kernel unless n_unsafe == 0.                                          */
int cice_evp_hip_cgrid_window_deps(const cice_evp_hip_dims *dims, int32_t *n_windows, int32_t *n_edges, int32_t *n_oneway, int32_t *n_unsafe);
/* Host only: how the one-launch C-grid schedule of large domains shares a rank's cells between the marched kernel (cg_strip) and the
 * windowed kernel (halo_plan.h: strip_zones, strip_items, strip_windows, on the window table of ex x ey positions).  items6: per work
 * item block, column of lane 2, first and last owned row (1-based), first and last owned lane (lane l holds column items6[1] - 2 + l);
 * tiles4 / in_zone: the windows (block, first owned i, first owned j, regular) and 1 where the marched kernel owns the window's cells.
 * lo0: first lane that may own a column (2, or 3 where the kernel forms the lengths); slots, seg_min, seg as in strip_items.  Pass NULL
 * arrays to learn the counts.                                                                                                        */
int cice_evp_hip_cgrid_strip_plan(const cice_evp_hip_dims *dims, int32_t ex, int32_t ey, int32_t lo0, int32_t slots, int32_t seg_min, int32_t seg,
                                  int32_t *n_items, int32_t *items6, int32_t items_cap, int32_t *n_windows, int32_t *tiles4, uint8_t *in_zone, int32_t windows_cap,
                                  int32_t *seg_rows);
/* Test hook: route the exchanges and the rank agreements of the marching path through HOST buffers and the caller's
 * callbacks instead of RCCL (which refuses two ranks on one device), so that its several-rank form can be run as
 * processes sharing one GPU (tools/mailbox_2proc.py --march: torch.distributed gloo underneath).  xchg: per peer q
 * (ascending rank) send_count[q] doubles starting at send + sum of the counts before, likewise recv; returns 0.  reduce:
 * op 0 = min of one int32, 1 = max of one uint32, in place.  NULL callbacks switch the hook off.  Not for production.  */
typedef int (*cice_evp_hip_test_xchg_fn)(void *user, int32_t npeers, const int32_t *peer_rank, const int64_t *send_count,
                                         const int64_t *recv_count, const double *send, double *recv);
typedef int (*cice_evp_hip_test_reduce_fn)(void *user, int32_t op, void *value);
int cice_evp_hip_set_test_transport(cice_evp_hip_test_xchg_fn xchg, cice_evp_hip_test_reduce_fn reduce, void *user);
/* Host-only (CPU tests): geometry and exchange lists of the marching path for dims->rank.  Every rank's sub-domain
 * must be one rectangle; the rank HOLDS its own cells plus `ext` (even) more on every side that has a neighbour, in strips
 * of `own` <= own_max columns (position of a cell = (storage row * nstrips + strip) * 64 + lane), and after one exchange
 * of the ring of ext + 2 cells ext/2 + 1 passes can follow.  geo14 = {gx0, gy0, nxr, nyr of what it holds, own, nstrips,
 * peers, cells sent, cells received, wraps inside, ext west / east / south / north}; per peer (ascending rank; the rank
 * itself when wrap_inside = 0 on a cyclic dimension it spans) the cells it sends / receives, recv_pos2 = the duplicate
 * position or -1.  Lists may be NULL.                                                                                */
int cice_evp_hip_march_plan(const cice_evp_hip_dims *dims, int32_t own_max, int32_t wrap_inside, int32_t ext, int32_t *geo14,
                            int32_t *peer_rank, int32_t *peer_nsend, int32_t *peer_nrecv, int32_t *send_pos,
                            int32_t *recv_pos1, int32_t *recv_pos2);
/* Per-CU record of the last on-chip resident launch with 16 x 16 tiles (tools): n <= 2048*8 ints, per CU
 * (index = XCC<<8 | HW_ID[15:8]) {lock, launch stamp, ice-holding waves on SIMD 0..3, 0, 0}.        */
int cice_evp_hip_debug_cuload(int32_t *out, int32_t n);
/* Phase stamps of the last resident launch made under CICE_EVP_HIP_RES_PROF=1 (tools/resident_phases.py):
 * per tile and chunk of 64 cells 8 x uint64 = shader cycles in {ring poll, stress, barrier wait, momentum
 * step + publish, barrier wait}, arrival rank of the workgroup on its CU, hardware wave, active chunks.    */
int cice_evp_hip_debug_prof(uint64_t *out, int32_t ntiles_max);
/* C grid, one-launch kernel: 8 stamps per window of the last launch (CICE_EVP_HIP_CGRID_PROF=1 when the geometry was set):
 * shader-clock cycles at 0 start, 1 / 2 before / after the first workgroup barrier, 3 / 4 the second, 5 level C's arithmetic
 * done, 6 end; 7 = XCC id << 32 | HW_ID.  Returns the number of windows.  tools/cgrid_phases.py */
int cice_evp_hip_debug_cgrid_prof(uint64_t *out, int32_t ntiles_max);
/* The same for the on-chip resident C-grid kernel (cg_res): 32 values per window = 4 waves x 8 phases, shader cycles summed over the
 * last launch: poll | barrier | S | barrier | T | barrier | U + barrier | C.  Returns the number of windows.                       */
int cice_evp_hip_debug_cgres_prof(uint64_t *out, int32_t ntiles_max);
/* Host-only: build the plan for `dims` without touching a device (CPU tests). */
int cice_evp_hip_plan_build(const cice_evp_hip_dims *dims);
int cice_evp_hip_halo_plan(int32_t *counts4, int32_t *local_dst, int32_t *local_src,
                           int32_t *local_sign, int32_t *peer_rank, int32_t *peer_nsend,
                           int32_t *peer_nrecv, int32_t *send_src, int32_t *recv_dst);

/* Tripole (u-fold) part of the plan: counts3 = {seam pairs, pole cells, late copies}.
 * After every velocity update the pair (a,b) of the seam row becomes (xavg, -xavg),
 * pole cells change sign, then the late copies are (re)applied
 * (ice_boundary.F90:1630-1649, 1689-1722).  Lists may be NULL.                  */
int cice_evp_hip_seam_plan(int32_t *counts3, int32_t *seam_a, int32_t *seam_b, int32_t *seam_pole,
                           int32_t *late_dst, int32_t *late_src, int32_t *late_sign);
/* What the mailbox transports use on top of cice_evp_hip_halo_plan's lists, same peer order and
 * lengths as send_src / recv_dst: send_dst = the ghost cell each sent value fills, as an offset
 * into the PEER's array (remote stores need it; no set-up traffic -- every rank enumerates every
 * rank's ghosts); recv_gid = global cell number (ig-1)+nx_global*(jg-1) each received ghost mirrors
 * (probe exchanges).  Lists may be NULL.                                                         */
int cice_evp_hip_peer_plan(int32_t *send_dst, int32_t *recv_gid);
/* What the on-chip kernel uses when the tripole fold row is split over ranks: per peer counts4 = {ghost entries at the head
 * of the send list, of the recv list (raw seam values for the staging slots follow them), seam images out, in}; send_sign =
 * the factor the receiver applies, send-list order; out3 = (seam cell here, ghost cell at the peer, sign) per image the owner
 * feeds with its FINAL value; in3 = (ghost cell here, global column of the seam cell, sign).  Lists may be NULL.        */
int cice_evp_hip_fold_images_plan(int32_t *counts4, int32_t *send_sign, int32_t *out3, int32_t *in3);
/* Factor applied to each received value (+1, or -1 for a ghost cell across the tripole fold), recv-list order. */
int cice_evp_hip_peer_signs(int32_t *recv_sign);
/* Ghost-cell lists of cell-centre fields (T-grid inputs of cice_evp_hip_prep) whose source is on
 * this rank: a[dst] <- (vector kind ? vsign : 1) * a[src], src = -1: 0.  Returns 1 (not an error)
 * when some ghost needs another rank.  Lists may be NULL.                                       */
int cice_evp_hip_center_plan(int32_t *count, int32_t *dst, int32_t *src, int32_t *vsign);
/* Lists behind cice_evp_hip_stress_halo: a1[dst] <- a2[src] for every partner pair (src = -1:
 * fill 0, ice_boundary.F90:7643-7645).  Lists may be NULL.                                     */
int cice_evp_hip_stress_plan(int32_t *count, int32_t *dst, int32_t *src);
/* Lists of the shifted-copy exchange that serves centre-kind fields across a tripole fold whose row is split over ranks
 * (cice_amd/csrc/halo_plan.h): which 0 = cells of row NY-1 where a'(c) = a(c + nx_block + 1) is built, 1 = centre-field
 * ghost cells taken from the exchanged copy, 2 = the stress symmetrisation's; 3, 4 = east-west ghost cells of row NY
 * owned elsewhere and the staging slots (offsets behind the array) a plain exchange leaves their values in.  Returns 1
 * when the fold row is split.                                                                                       */
int cice_evp_hip_fold_split_plan(int32_t which, int32_t *count, int32_t *cells);
/* Flags of the plan, up to n of them: [0] some rank's in-loop velocity exchange crosses the tripole fold or uses
 * seam staging slots -- computed identically on every rank, what collective decisions (cice_evp_hip_halo_mask)
 * hang on; [1] fold_rows (0 none here, 1 all here, 2 shared); [2] stress symmetrisation needs another rank;
 * [3] a cell-centre ghost needs another rank; [4] ... across the fold.  Returns the number written.           */
int cice_evp_hip_plan_flags(int32_t *flags, int32_t n);


#ifdef __cplusplus
}
#endif
#endif /* CICE_EVP_HIP_TESTING_H */
