// Device-side argument blocks and launcher prototypes shared by
// evp_kernels.hip (kernels) and evp_api.cpp (C ABI / state management).
#pragma once
// test-build-only environment switch (evp_host_common.cpp): NULL in the production library
const char *evp_env_test(const char *key);
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

// EVP scalars as the kernels see them (subset of cice_evp_hip_params)
struct EvpScalars {
    double arlx1i, denom1, brlx, revp;
    double e_factor, epp2i;
    double capping, Ktens;
    double u0, cosw, sinw;
    double rhow;
};

// One subcycle's worth of pointers.  Every array is (nx, ny, nblocks) fp64,
// i fastest -- the memory image of CICE's (nx_block,ny_block,max_blocks).
struct EvpArgs {
    EvpScalars p;
    int nx, ny;
    size_t plane;              // nx*ny
    int gx, gy, ntiles;        // tile grid (filled by evp_launch_subcycle)
    int xcdmap;                // 1: XCD-contiguous tile order
    const int *tile_list;      // non-NULL: run only these tiles (boundary-first / interior split)
    int tile_count;
    int last;                  // write strintx/y, taubx/y (needed after the last subcycle only)
    const int4 *blk;           // per block: ilo, ihi, jlo, jhi (1-based)
    const uint8_t *mask;       // bit0 = iceTmask, bit1 = iceUmask
    // ping-pong state
    const double *u_in, *v_in;
    double *u_out, *v_out;
    const double *sig_in[12];  // stressp_1..4, stressm_1..4, stress12_1..4
    double *sig_out[12];
    // static metric terms (init_dyn_shared, ice_dyn_shared.F90:384-441)
    const double *dxT, *dyT, *dxhy, *dyhx, *cxp, *cyp, *cxm, *cym, *DminTarea, *uarear;
    const double *HTE, *HTN;   // for EVP_F_METRICS (metric terms recomputed per cell)
    double deltaminEVP;
    // per-call inputs (dyn_evp1d_run argument list, ice_dyn_evp1d.F90:121-153)
    const double *strength, *Cw, *aiX, *uocn, *vocn, *waterx, *watery, *forcex, *forcey;
    const double *umassdti, *fm, *TbU, *uvel_init, *vvel_init;
    const double *vrelfac;     // (aiX*rhow)*Cw, formed once per call (same rounding as stepu :933)
    unsigned flags;            // EVP_F_*
    // halo push table (see halo_plan.h): per block 2*(nj_max+ni_max) edge slots x 2 entries
    const int *push;
    int push_ni, push_nj;
    // diagnostics of the last subcycle
    double *strintx, *strinty, *taubx, *tauby;
    // mailbox halo riding in the launch (evp_halo_direct.h); dx == NULL: off.  tile_list then
    // holds ALL tiles, the dx_nb boundary tiles first; workgroup dx_nb is the exchange workgroup.
    const struct EvpDirect *dx;
    unsigned *dx_count;        // boundary tiles checked in so far (never reset)
    unsigned *dx_fseq;         // launches with a riding exchange so far
    int dx_nb;
};

// On-chip resident subcycle (evp_resident2.hip): velocities another tile needs travel as tagged 16-byte records
// logw: log2 of the tile width in T-cells (6, 5, 4 -> tiles of 64x4, 32x8, 16x16 T-cells)
void evp_resident_geometry(int max_ni, int max_nj, int logw, int *gx, int *gy);
#define EVP_RES2_RING 256
struct EvpResident2 {
    int ndte;
    int cur0;                  // ping-pong buffer holding the input velocities / stresses
    int dry;                   // 1: residency + timing probe, nothing written back
    unsigned tag_base;         // launch epoch << 12; a record of subcycle k carries tag_base + k
    unsigned spin_limit;
    int *err;
    const uint8_t *pubmap;     // per cell: 1 = some other tile's ring mirrors this U-cell
    const int4 *ring;          // [ntiles][EVP_RES2_RING]: x record cell, y LDS index, z producing U-cell (-1 none)
    const int *ring_cnt;       // [ntiles]
    void *rec[2];              // [2][ncell] x {u granule, v granule} (2 x 16 bytes), by subcycle parity
    double *u[2], *v[2];       // plain arrays: input from [cur0], final state to both
    // pointer table in device memory (keeps 28 pointers out of the kernel's SGPR budget):
    // [0..11] sig buffer 0, [12..23] sig buffer 1, [24..27] strintx strinty taubx tauby
    double *const *tab;
    int nblocks;               // CICE blocks of this rank (tiles = nblocks x gx x gy)
    const int *order;          // [ntiles] tile run by workgroup w (NULL: identity)
    // Only the tiles that hold ice run (one rank, no neighbours on other ranks): nlaunch workgroups, order[0 .. nlaunch-1] their
    // tiles.  A tile without ice changes nothing in a call -- its U-cells keep the values every reader loaded at the start -- so a
    // ring entry whose producer tile is not live is not polled (live[celltile[producer cell]]).  live == NULL: every tile runs.
    int nlaunch;               // workgroups to launch (0: all tiles)
    const uint8_t *live;       // [ntiles]
    const int *celltile;       // [ncell] tile that owns the U-cell, -1 for ghost cells
    // 16 x 16 tiles only (rim wave / interior waves, see evp_resident2.hip): which T-cell of the tile a
    // thread owns, the T-cells that read ring velocities first; how many waves hold such cells / ring entries
    const uint8_t *perm;       // [ntiles][256] cell position trow*16 + tcol of (permuted) thread index
    const uint8_t *late_waves; // [ntiles]
    const uint8_t *nact;       // [ntiles] chunks (64 entries of perm) that hold ice cells: the first nact
    const uint8_t *nlate;      // [ntiles] COOP: rim T-cells with ice = the first nlate entries of perm (<= 64); NULL: the variant is off
    int *cuload;               // [2048][8] per-CU record of the launch: lock, stamp, ice-holding waves per SIMD
    unsigned long long *prof;  // NULL, or [ntiles][4 chunks][8]: cycles per phase (tools)
    int dbg;                   // timing experiments only (CICE_EVP_HIP_RES_DEBUG; WRONG results): 1 no tag check, 2 no ring loads, 4 longer sleep; 8 = every fourth tile lags 10 us per subcycle (results stay right); 16 = tile 1 never runs (every wait on it gives up); A/B switches, results stay right: 32 = 16 x 16 tiles without the rim-wave split, 64 = without the per-CU SIMD balancing
    int par0;                  // which of rec[0/1] holds the records of subcycle index 0 of THIS launch
                               // (flips so that a launch never starts in the buffer the previous one ended in)
    // neighbours on other GPUs (ring entries with z == -2 are produced there); rimg == NULL: none
    const int2 *rimg;          // [ncell][3] images on other ranks: {peer index | 256 if the image takes the negative, ghost cell at that peer}, x = -1 none
    void *const *peer_rec;     // [npeers] the peer's record buffer (parity 0) as mapped here
    const size_t *peer_rstride;// [npeers] bytes between the two parities of that buffer
    unsigned long long timeout_ticks;   // bound of a wait on another rank (100 MHz wall clock)
    // tripole grid (the top physical row lies on the fold); all NULL otherwise
    const int *seam;           // [nx] per column of the fold row: partner cell * 4 + role (1 low, 2 high, 3 pole), 0 none
    int tfold;                 // 1: tripoleT (T-fold): a seam entry names the cell of row NY-1 the top-row cell is the image of (role 1);
                               // after its momentum step the cell takes -1 x that cell's NEW value (no average: ice_boundary.F90:1686-1722)
    const int *img3;           // [ncell][3] ghost images of every U-cell: dst * 2 + (sign < 0), -1 none
    void *rec_raw[2];          // records of the pre-average velocities of the fold row, by subcycle parity
    // fold row split over ranks: a seam cell whose partner lives on another rank stores its raw record into that rank's
    // rec_raw buffer as well (slot = a staging index >= ncell there) and polls the partner's in its own buffer at the slot
    // the seam table names (partner index >= ncell)
    const int2 *rraw;          // [ncell][3] {peer index, slot at that peer}, x = -1 none; NULL: no split fold
    void *const *peer_raw;     // [npeers] the peer's rec_raw buffer (parity 0) as mapped here
    const size_t *peer_raw_stride;
};
int evp_resident2_max_blocks_per_cu(bool strict, int cap, unsigned flags, int logw, bool remote, bool coop = false);
bool evp_resident2_coop_built(bool strict, int cap, int logw, bool remote);   // the rim-cells-by-corners variant exists for this combination
void evp_launch_resident2(const EvpArgs &A, const EvpResident2 &R, int max_ni, int max_nj, int logw,
                          bool strict, int cap, hipStream_t st);

// Several (2 .. 4) subcycles per pass over a device-private strip-major layout (evp_march.hip, evp_host_march.cpp)
#ifndef EVP_MARCH_PAD          // (a build-time A/B: -DEVP_MARCH_PAD=3 makes three the most subcycles per pass, strips of 58)
#define EVP_MARCH_PAD 4        // width of the overlap: lanes on either side of a strip's own columns, halo rows below and above,
                               // halo columns of the row-major byte mask, cells of the ring between ranks = the most subcycles
                               // one pass can advance (validity shrinks by one cell per side and subcycle)
#endif
#define EVP_MARCH_KMAX EVP_MARCH_PAD
#define EVP_MARCH_OWN (64 - 2 * EVP_MARCH_PAD)       // most columns a 64-lane strip can own
#define EVP_MARCH_S_NF 14      // fields per block: state (u v sig x 12), constants, optional operands, diagnostics
#define EVP_MARCH_C_NF 13
#define EVP_MARCH_O_NF 5
#define EVP_MARCH_D_NF 4
#define EVP_MARCH_NODUP 0xffffffffu
struct EvpMarch {
    EvpScalars p;
    double deltaminEVP;
    int nxr, nyr;              // cells of the rank's rectangle
    int ldx;                   // row stride of the row-major byte mask
    int own;                   // columns a strip owns (<= EVP_MARCH_OWN)
    int nstrips, nseg, seglen, nitems;
    int wrapx;                 // the rectangle spans a cyclic E-W dimension: strips wrap around
    int last;                  // write strintx/y, taubx/y (last pass of a call)
    int kpass;                 // subcycles this pass advances the state by (2 .. EVP_MARCH_KMAX)
    int order;                 // work item order: bit0 one contiguous run per XCD, bit1 segment index fastest
    unsigned flags;            // EVP_F_WATER_IS_OCN / EVP_F_TBU_ZERO
    const uint8_t *mask;       // row-major [rows][ldx]: bit0 iceTmask, bit1 iceUmask (0 where there is no cell)
    // strip-major buffers: [row][strip][field][64 lanes]
    const double *st_in;       // u v stressp_1..4 stressm_1..4 stress12_1..4
    double *st_out;
    const double *cst;         // dxT dyT strength HTE HTN vrelfac uocn vocn forcex forcey umassdti fm uarear
    const double *opt;         // waterx watery TbU uvel_init vvel_init, or NULL when none of them is read
    double *diag;              // strintx strinty taubx tauby
    const unsigned *dup;       // [nstrips][64]: where, from the start of a state row, the duplicate of the lane's column lives
    const int4 *items;         // NULL: item = (strip, segment) of the regular cut; else nitems explicit {strip, Y0, Y1, 0} (the
                               // early launch of the cells other ranks are waiting for, evp_host_march.cpp)
};
struct EvpMarchGeo {
    int nxr, nyr, ldx, rows;   // rectangle: owned cells; mask row stride; rows of every buffer (nyr + halo + spare)
    int own, nstrips;          // strips of `own` owned columns
    int nxb, nyb, plane;       // CICE block arrays (nx_block, ny_block, their product)
    int nblocks;
    int bsx, bsy, nbx, nby;    // interior size of a full block, blocks of the rank in x / y
    int ilo;                   // first interior index of a block (nghost + 1)
    int wrapx;
    int gx0, gy0, nxg, nyg, ew_cyclic;   // the rectangle in the global index space (0-based), the global domain
    int ext_w, ext_s, nxo, nyo;          // several ranks: the rank's OWN cells are [ext_w, ext_w+nxo) x [ext_s, ext_s+nyo) of the rectangle
    const int *blkid;          // [nby][nbx] local block index
    const int2 *blk_org;       // [nblocks] rectangle coordinates of the first interior cell
    const int4 *blk;           // [nblocks] ilo ihi jlo jhi
};
#define EVP_MARCH_TAB 40
struct EvpMarchTab {
    double *blk[EVP_MARCH_TAB];     // block-layout arrays
    double *pk[EVP_MARCH_TAB];      // strip-major buffer the field lives in
    double *pk2[EVP_MARCH_TAB];     // gather: second copy (ping-pong partner) or NULL
    int nf[EVP_MARCH_TAB];          // fields per block of that buffer
    int slot[EVP_MARCH_TAB];        // the field's slot
    int n;
};
void evp_launch_march(const EvpMarch &A, bool strict, int mode, hipStream_t st);
void evp_launch_march_gather(const EvpMarchGeo &G, const EvpMarchTab &T, const uint8_t *mask_blk, uint8_t *mask_rect,
                             hipStream_t st);
void evp_launch_march_check(const EvpMarchGeo &G, const EvpMarchTab &T, const uint8_t *mask_blk, const uint8_t *mask_rect,
                            int nuv, int nfringe, unsigned *bad, hipStream_t st);
void evp_launch_march_scatter(const EvpMarchGeo &G, const EvpMarchTab &T, const uint8_t *mask_blk, int nuv, int nsig,
                              hipStream_t st);
// Wire format of the ring: the peers' blocks one after the other, inside a block field by field ([f][entry]: consecutive
// lanes take consecutive entries of ONE field -- the ring cells of a row sit next to each other in a 64-lane block of the
// packed layout, so a wave gathers runs of 48 bytes and more instead of 64 single doubles 512 bytes apart).
#define EVP_MARCH_DIRECT_MAXPEER 16
struct EvpRingCuts {
    int n;                                           // peers
    int start[EVP_MARCH_DIRECT_MAXPEER + 1];         // entries [start[q], start[q+1]) of the list belong to peer q
};
void evp_launch_march_pack(const double *buf, int nf, const int *pos, int n, const EvpRingCuts &C, double *out, hipStream_t st);
// The same ring exchange WITHOUT a communication library (ranks on the GPUs of one node): the pack kernel stores every
// entry straight into the receiving rank's inbox -- peer memory mapped through HIP IPC, plain stores over xGMI -- and the
// last of its workgroups raises this rank's sequence number in every peer's flag slot; the unpack kernel waits (bounded)
// for the peers' numbers and reads its own inbox.  Inboxes are double buffered by sequence parity (a rank cannot be two
// exchanges ahead of a neighbour: it needs that neighbour's flag to finish one).  evp_host_march.cpp: march_direct_*.
struct EvpMarchDirect {
    int npeers;
    double *dst[EVP_MARCH_DIRECT_MAXPEER];           // where they land in that peer's inbox (parity 0), as mapped here
    size_t dst_pstride[EVP_MARCH_DIRECT_MAXPEER];    // doubles to parity 1 of that inbox
    unsigned *peer_flag[EVP_MARCH_DIRECT_MAXPEER];   // my flag slot at that peer, as mapped here
    unsigned *flags_in;           // own flag slots, one per peer, 16 unsigneds apart
    unsigned *count;              // workgroups of the running pack kernel that have stored their share
    int *err;                     // != 0: a wait gave up (1 + index of the peer)
    const double *inbox;          // own inbox [2 parities][n_recv * nf]
    size_t inbox_pstride;         // doubles
    unsigned long long timeout_ticks;   // of the 100 MHz wall clock
};
void evp_launch_march_pack_direct(const double *buf, int nf, const int *pos, int n, const EvpRingCuts &C, const EvpMarchDirect &D, unsigned seq,
                                  hipStream_t st);
// verify != NULL: do not unpack, compare the inbox with verify[] (what the library transport delivered) and count differences in *bad
void evp_launch_march_unpack_direct(double *buf, double *buf2, int nf, const int *pos1, const int *pos2, int n, const EvpRingCuts &C,
                                    const EvpMarchDirect &D, unsigned seq, const double *verify, unsigned *bad, hipStream_t st);
void evp_launch_march_unpack(double *buf, double *buf2, int nf, const int *pos1, const int *pos2, int n, const EvpRingCuts &C, const double *in,
                             hipStream_t st);
void evp_launch_march_pack_mask(const uint8_t *mask, const int *idx, int n, double *out, hipStream_t st);
void evp_launch_march_unpack_mask(uint8_t *mask, const int *idx, int n, const double *in, hipStream_t st);

// Mailbox halo between GPUs of one node (evp_halo_direct.hip)
#define EVP_DIRECT_MAXPEER 32
#define EVP_DIRECT_FLAG_STRIDE 16          // unsigneds: one 64-byte line per flag
struct EvpDirect {
    int n_send, n_recv, npeers;
    const int *send_src;          // [n_send] local source cells, the plan's peers concatenated
    double *const *send_addr;     // [n_send] where the entry lands in its peer's inbox (parity 0), as mapped here
    const unsigned *send_pstride; // [n_send] doubles to the other parity of that inbox
    const int *recv_dst;          // [n_recv] local ghost cells
    const signed char *recv_sign;
    const int *recv_slot;         // [n_recv] inbox slot of the entry (NULL: its index) -- masked halos keep the unmasked layout
    int n_recv_slots;             // slots per parity of the inbox
    double *inbox;                // own inbox [2 parities][n_recv][u,v]
    unsigned *flags_in;           // own flag slots, one per peer, EVP_DIRECT_FLAG_STRIDE apart
    unsigned *seq;                // exchanges completed so far
    int *err;                     // != 0: a wait gave up (1 + index of the peer)
    unsigned long long timeout_ticks;   // of the 100 MHz wall clock
    int dbg;                      // timing experiments only (CICE_EVP_HIP_HALO_DEBUG): 1 no release, 2 no acquire, 4 no flags, 8 empty
    unsigned *const *peer_flag;   // [npeers] my flag slot in the peer's mailbox as mapped here
};
void evp_launch_halo_direct(const EvpDirect &D, double *u, double *v, hipStream_t st);

// Preparation phase of evp() on the device (evp_prep.hip)
struct EvpPrep {
    int nx, ny;
    size_t plane;
    const int4 *blk;
    const uint8_t *tmask, *umask, *umask_old;
    const double *hm, *tarea, *uarea, *fcor;
    double *t[11];             // aice vice vsno aice_init cdn_ocn uocn vocn ss_tltx ss_tlty strairxT strairyT (device copies)
    double *tmass, *umass, *maskd;   // maskd: iceTmask as 0/1 (halo-updated like a field)
    uint8_t *tmphm;
    double *ss_tltxU, *ss_tltyU, *strairxU, *strairyU, *strtltx, *strtlty;
    double *aiU, *cdn_ocnU, *uocnU, *vocnU, *umassdti, *fm, *waterx, *watery, *forcex, *forcey;
    double *uvel_init, *vvel_init, *uvel, *vvel;
    double *sig[12];
    double *strintx, *strinty, *taubx, *tauby;   // zeroed as dyn_prep2 does (:704-712 everywhere, :776-781 off the ice)
    uint8_t *mask;             // out: bit0 iceTmask, bit1 iceUmask
    unsigned *flagword;        // out: bit0 = waterx/watery differ from uocnU/vocnU somewhere
    double dt, rhoi, rhos, gravit, dyn_area_min, dyn_mass_min, cosw, sinw;
    int ssh_coupled;
};
struct EvpPrepHalo {
    double *a[10];
    unsigned char is_vec[10];
    int narr;
    const int *dst, *src;
    const signed char *vsign;
    int n;
};
void evp_launch_prep1(const EvpPrep &P, int nblocks, hipStream_t st);
void evp_launch_halo_center(const EvpPrepHalo &H, hipStream_t st);
void evp_launch_prep_average(const EvpPrep &P, int nblocks, hipStream_t st);
void evp_launch_prep2(const EvpPrep &P, int nblocks, hipStream_t st);
void evp_launch_prep_average_prep2(const EvpPrep &P, int nblocks, hipStream_t st);
void evp_launch_words_to_bytes(const int32_t *w, uint8_t *b, size_t n, hipStream_t st);
void evp_launch_seabed_prob(const EvpPrep &P, int nblocks, const double *hwater, const double *aicen, const double *vicen,
                            int ncat, double alphab, double rhoi, double rhow, double gravit, double pi, double puny,
                            double *Tbt, double *TbU, unsigned *flagword, hipStream_t st);
void evp_launch_fold_shift2(const double *a, const double *b, double *a2, double *b2, const int *cells, int n, int nx, hipStream_t st);
void evp_launch_fold_extract2(double *a, double *b, const double *a2, const double *b2, const int *dst, int n, double fa, double fb,
                              hipStream_t st);
void evp_launch_seabed_prob_t(const EvpPrep &P, int nblocks, const double *hwater, const double *aicen, const double *vicen,
                              int ncat, double alphab, double rhoi, double rhow, double gravit, double pi, double puny,
                              double *Tbt, hipStream_t st);
void evp_launch_seabed_lkd(const EvpPrep &P, int nblocks, const double *hwater, double *TbU, double k1, double k2,
                           double alphab, double threshold_hw, unsigned *flagword, hipStream_t st);

enum : unsigned {
    EVP_F_METRICS = 1u,     // recompute cxp..DminTarea from HTE,HTN,dxT,dyT (tarea == dxT*dyT verified)
    EVP_F_WATER_IS_OCN = 2u,// waterxU==uocnU and wateryU==vocnU bit for bit on every active U-cell
    EVP_F_TBU_ZERO = 4u,    // TbU == 0 on every active U-cell (seabed_stress off)
    EVP_F_VRELFAC = 8u,     // use the pre-multiplied drag factor
    EVP_F_PUSH = 16u,
    EVP_F_DXHY_ARRAY = 32u, // with EVP_F_METRICS: dxhy, dyhx still come from their arrays (tripole ghost row)       // edge U-cells also write their ghost images (halo fused into the kernel)
};

// one launch for many same-sized array copies (evp_copy.hip); pointers may be device-mapped host memory
#define EVP_COPY_MAX 40
struct EvpCopyTab {
    const double *src[EVP_COPY_MAX];
    double *dst[EVP_COPY_MAX];
    int n;                     // arrays
    size_t len;                // doubles per array
    int vec2;                  // every pointer 16-byte aligned: double2 accesses
};
void evp_launch_copy_many(const EvpCopyTab &T, hipStream_t st);
double evp_stream_probe(size_t ncells, int reps, hipStream_t st);
void evp_launch_copy_many_masked(const EvpCopyTab &T, const uint8_t *mask, unsigned bit, hipStream_t st);
void evp_launch_zero_sig_off_mask(double *const *sig0, double *const *sig1, const uint8_t *mask, size_t n, hipStream_t st);

void evp_launch_vrelfac(const double *aiX, const double *Cw, double rhow, double *out, size_t n,
                        hipStream_t st);
void evp_launch_subcycle(const EvpArgs &A, int max_ni, int max_nj, int nblocks, int variant,
                         bool strict, int cap, hipStream_t st);
void evp_launch_deformations(const EvpArgs &A, int nblocks, bool strict, const double *dxU, const double *dyU,
                              const double *tarear, double *divu, double *shear, double *vort,
                              double *rdg_conv, double *rdg_shear, hipStream_t st);
void evp_launch_dyn_finish(const EvpArgs &A, int nblocks, bool strict, double *strocnx, double *strocny,
                           hipStream_t st);
// tile geometry of a variant (tile height, tiles in x / y per block)
void evp_tile_geometry(int max_ni, int max_nj, int variant, int *tyb, int *gx, int *gy);
void evp_launch_halo_local(double *u, double *v, const int *dst, const int *src,
                           const signed char *sign, int n, hipStream_t st);
// u2, v2: the other ping-pong buffer (takes the same stores; null: none)
void evp_launch_halo_seam(double *u, double *v, double *u2, double *v2, const int *pa, const int *pb, int npair, const int *pole,
                          int npole, const int *ldst, const int *lsrc, const signed char *lsign,
                          int nlate, hipStream_t st);
int evp_halo_seam_fin_capacity();
void evp_launch_halo_seam_fin(double *u, double *v, double *u2, double *v2, const int *dst, const int *fa, const int *fb,
                              const signed char *coef, int n, hipStream_t st);
void evp_launch_halo_stress(double *const *sig12, const int *dst, const int *src, int n, hipStream_t st);
// tripoleT: the top physical row of _1 / _2 from the partner's mirrored cell, east-west ghost cells of that row of _3 / _4 from
// their own array (halo_plan.h)
void evp_launch_halo_stress_tfold(double *const *sig12, const int *dst, const int *src, int n, const int *odst, const int *osrc, int no,
                                  hipStream_t st);
void evp_launch_halo_pack(const double *u, const double *v, const int *src, double *buf, int n,
                          hipStream_t st);
void evp_launch_halo_unpack(double *u, double *v, const int *dst, const signed char *sign,
                            const double *buf, int n, hipStream_t st);

// ---------------------------------------------------------------------
// C-grid EVP subcycle (evp_cgrid.hip; reference ice_dyn_evp.F90:938-1099).  Field order of the tables =
// include/cice_evp_hip.h (CICE_EVP_HIP_CGRID_*).
// ---------------------------------------------------------------------
enum { CG_NF = 19, CG_NIN = 23, CG_NG = 23 };
enum {   // f[]
    CF_UE = 0, CF_VE, CF_UN, CF_VN, CF_UU, CF_VU, CF_SP, CF_SM, CF_S12T, CF_S12U, CF_STRX, CF_STRY, CF_TAUBX,
    CF_TAUBY, CF_ZETA, CF_ETA, CF_ETAU, CF_SHEARU, CF_DELTAU
};
enum {   // in[]
    CI_STRENGTH = 0, CI_CWE, CI_AIE, CI_UOCNE, CI_VOCNE, CI_WATERXE, CI_FORCEXE, CI_EMASSDTI, CI_FME, CI_UE_INIT,
    CI_TBE, CI_RHEOE, CI_CWN, CI_AIN, CI_UOCNN, CI_VOCNN, CI_WATERYN, CI_FORCEYN, CI_NMASSDTI, CI_FMN, CI_VN_INIT,
    CI_TBN, CI_RHEON
};
enum {   // g[]
    CG_DXT = 0, CG_DYT, CG_DXU, CG_DYU, CG_DXE, CG_DYE, CG_DXN, CG_DYN, CG_UAREA, CG_TAREA, CG_EAREA, CG_NAREA,
    CG_EAREAR, CG_NAREAR, CG_EPM, CG_NPM, CG_UVM, CG_HM, CG_DMINT, CG_RXN, CG_RXNR, CG_RYE, CG_RYER
};
struct EvpCgrid {
    double *f[CG_NF];
    const double *in[CG_NIN];
    const double *g[CG_NG];
    const uint8_t *gmask;         // non-null: the fused kernels derive 15 of the 23 static arrays (as cg_one does; the host verified the
    size_t gstride;               // identities); the four land masks as bits; g[k] = g[0] + k * gstride
    const double *strengthU;      // visc_method = 'avg_strength': T->U average of the strength (once per call)
    const uint8_t *mask;          // bit0 iceT, bit1 iceU, bit2 iceE, bit3 iceN, bit4: the cell has ghost images,
                                  // bit5: iceU of the cell, or of the interior cell this ghost cell mirrors
    const double *s12_in;         // fused path: stress12U ping-pong (read s12_in, write f[CF_S12U])
    const double *facE, *facN;    // (aiE*rhow)*cdn_ocnE, (aiN*rhow)*cdn_ocnN: the leading factor of vrel, once per call
    const int *img_slot;          // per cell: row of img_dst, or -1
    const int *img_dst;           // 3 per row: ghost cells of this rank that mirror the cell (-1: none)
    const int4 *blk;
    EvpScalars p;
    double deltaminEVP;
    int nx, ny, nblocks, avg_strength;
    int split_faces;              // fused step kernel: E and N face of a cell in different waves (small grids)
    int xcd_rows;                 // > 0: workgroup rows per XCD band (1-D launch, vertically adjacent workgroups on one XCD)
    int tripole;                  // the fold step writes into cells without ice: what the reference re-zeroes every subcycle is re-zeroed
    size_t plane;
};
// One launch per subcycle (evp_cgrid.hip: cg_one).  A window = ox x oy positions (one workgroup: 32 x 8, 64 x 8 or 64 x 16), the
// inner (ox-3) x (oy-3) of them owned cells; tab: per window and position the cell whose value the reference has there
// (>= 0), or -1 - ghost cell for a ghost cell nothing is copied into (its arrays are read, not computed).
struct EvpCgOne {
    const int *tab;
    const int4 *tiles;            // block, first owned i, first owned j (1-based), 1 if the window is regular (no table needed)
    int ntiles, per_xcd;          // windows; windows per XCD (launch = 8 * per_xcd workgroups)
    int ox, oy;
    int plain;                    // 1: window = workgroup id (A/B switch; default: contiguous runs of the list per XCD)
    const double *uE_in, *vN_in, *sp_in, *sm_in;   // previous subcycle's buffers (A.f[...] = this subcycle's)
    const double *gbase, *inbase; // the static and the per-call tables as one allocation each: array k = base + k * stride
    size_t stride;                // (70 pointers as kernel arguments do not fit the scalar registers)
    unsigned long long *prof;     // test build, CICE_EVP_HIP_CGRID_PROF=1: 8 cycle stamps per window (tools/cgrid_phases.py)
    const uint8_t *gmask;         // non-null: 15 of the 23 static arrays are derived in the kernel (identities verified by the
                                  // host, evp_host_cgrid.cpp: derive_geometry_check); the four land masks as bits of this byte
};
void evp_launch_cgrid_one(const EvpCgrid &A, const EvpCgOne &T, int fast, int last, hipStream_t st);
// The interior of a large block, marched (evp_cgrid.hip: cg_strip): one wave per item = strip of 64 positions (up to 60 owned
// columns) x segment of rows; T carries the buffers and tables as for cg_one (its window list is not used).  The derived view of
// the static table (T.gmask) only; fast, last, A.avg_strength: as for cg_one.
struct EvpCgStrip {
    const int *items;             // x 6: block, column of lane 2, first and last owned row (1-based), first and last owned lane
    int nitems, per_xcd;          // items; workgroups (of four items) per XCD (launch = 8 * per_xcd workgroups)
    int lengths;                  // 1: dxT, dyT, dxU, dyU, dxE, dyN formed in the kernel from dxN, dyE (verified by the host); the items own lanes >= 3
};
void evp_launch_cgrid_strip(const EvpCgrid &A, const EvpCgOne &T, const EvpCgStrip &Z, const EvpCgOne *E, int fast, int last, hipStream_t st);
// All subcycles of a call in one launch, state on the chip (evp_cgrid_res.hip: cg_res).  Windows of 16 x 16 positions, the inner
// 13 x 13 owned; tab: per window the source cell of its 17 x 17 positions (one row / column more than cg_one's: what level S reads
// of its north / east neighbour), as in EvpCgOne.  The velocities another window's rim mirrors travel as tagged 32-byte records.
#define EVP_CGRES_REACH 3      // == CGRES_REACH (halo_plan.h): positions beyond the last owned column / row a resident window polls
#define EVP_CGRES_SLOTS 4      // == CGRES_SLOTS: record slots per cell; the record of subcycle j sits in slot (j + par0) mod 4, so a window
                               // may be up to three subcycles ahead of one that reads it (halo_plan.h: cgres_dependencies proves it is not more)
struct EvpCgRes {
    const int *tab;               // [ntiles][17 * 17]
    const int4 *tiles;            // block, first owned i, first owned j (1-based), fold: fold window | tf << 8 | last owned row << 16
    const int4 *tiles2;           // fold (tripole grids): global column of tile column 0, NX, -, -   (halo_plan.cpp: build_fold_window_table)
    int fold;                     // 1: tripole (u-fold) grid, the kernel's FOLD variant
    int slow;                     // 1: the SLOW variant (seabed stress, waterx != uocn, rheofact != 1 on some ice cell: general momentum step)
    const int *order;             // [ntiles] window run by workgroup w (NULL: identity): the windows that hold ice in this call
    int ntiles;                   // ... and how many there are
    const uint8_t *live;          // per cell: its window runs in this call (NULL: all do)
    int nsub;                     // subcycles of this launch; the last one ends the call (the once-per-call arrays are stored in it)
    int dry;                      // 1: residency + timing probe, nothing written back
    int par0;                     // which of rec[0 .. 3] holds the records of subcycle index 0 of this launch
    unsigned tag_base;            // launch epoch << 12; a record of subcycle k carries tag_base + k
    unsigned spin_limit;
    int *err;                     // [8] first wait that gave up: 1, window, subcycle, cell, tag seen, tag wanted
    const uint8_t *pubmap;        // per cell: 1 = some other window's rim mirrors this cell
    void *rec[EVP_CGRES_SLOTS];   // [slots][ncell] x {uvelE granule, vvelN granule} (2 x 16 bytes), by subcycle modulo the slots
    const double *uE_in, *vN_in, *sp_in, *sm_in, *s12_in;               // the state on entry
    double *uE_out[2], *vN_out[2], *sp_out[2], *sm_out[2], *s12_out[2];   // ... and where it goes on exit (both allocations of each)
    const double *gbase, *inbase; // static / per-call tables: array k = base + k * stride
    size_t stride;
    const uint8_t *gmask;         // the four land masks as bits (derive_geometry_check passed: required)
    unsigned long long *prof;     // test build, CICE_EVP_HIP_CGRID_PROF=1: [ntiles][4 waves][8] cycles per phase (tools/cgres_phases.py)
    int long_sleep;               // A/B (test build): 512 instead of 64 cycles between two looks at a record
    int dbg;                      // test hooks (test build): 8 every fourth window lags, 16 window 1 never runs (real launches)
};
int evp_cgrid_res_max_blocks_per_cu(int avg_strength, int revised, int fold, int slow);
void evp_launch_cgrid_res_pair_check(const double *const *five, const int2 *pairs, int n, unsigned *flags, hipStream_t st);
void evp_launch_cgrid_res(const EvpCgrid &A, const EvpCgRes &R, hipStream_t st);
void evp_launch_cgrid_res_live(const EvpCgrid &A, const int *tab, const int4 *tiles, int ntiles, int fold, int *live_win, uint8_t *live_cell, hipStream_t st);
// phase: 0 strain_rates_U, 1 stressC_T, 2 T->U viscosity + stressC_U, 3 div_stress + stepu_C/stepv_C,
//        4 face->face and face->corner velocity averages, 5 strengthU (once per call), 6 zero what the reference's
//        whole-array zero fills leave zero outside the interior (uvelN, vvelE, uvel, vvel; once per call)
//        7 fused: averages + strain_rates_U, 8 fused: viscosity at the corners + stressC_U + div_stress + stepu_C/stepv_C,
//        9 copy field `last` into its ghost images
// last: the launch belongs to the last subcycle of a call (arrays nobody reads inside the loop are stored only then)
void evp_launch_cgrid_deformations(const EvpCgrid &A, const double *tarear, double *divu, double *shear, double *vort,
                                   double *rdg_conv, double *rdg_shear, hipStream_t st);
void evp_launch_cgrid_dyn_finish(const EvpCgrid &A, int which, double *strocnx, double *strocny, hipStream_t st);
void evp_launch_cgrid_phase(const EvpCgrid &A, int phase, int last, hipStream_t st);
// Tripole fold of the C-grid fields (u-fold; ice_boundary.F90:1626-1722): entries (dst, a, b, flip) per field location
// -- x[dst] = s * 0.5*(x[a] + isign*x[b])  (b >= 0: a point ON the fold, averaged with its partner)
//    x[dst] = s * x[a]                     (b < 0: a ghost cell mirroring a cell across the fold; a < 0: 0)
// with s = flip ? isign : 1 and isign = -1 for vector kinds, +1 for scalars.  Two passes (all reads, then all writes).
struct EvpCgFoldList { const int *dst, *a, *b; const unsigned char *flip; int n; };
struct EvpCgFold {
    double *x[4];              // up to four fields per step
    int loc[4];                // their location: 0 centre, 1 NE corner, 2 E face, 3 N face
    double isign[4];
    int nfields;
    EvpCgFoldList L[4];        // per location
    double *tmp;               // nfields x max n
    int maxn;
};
void evp_launch_cgrid_fold(const EvpCgFold &F, hipStream_t st);
// m4: iceTmask | iceUmask | iceEmask | iceNmask, n 32-bit words each
void evp_launch_cgrid_mask(const EvpCgrid &A, const int *m4, hipStream_t st);
void evp_launch_cgrid_umask(const EvpCgrid &A, double *scratch, int back, hipStream_t st);
// per call: facE / facN, and which of the default-configuration shortcuts hold bit for bit on every ice cell --
// flags bit0: waterxE != uocnE or wateryN != vocnN somewhere, bit1: a TbE / TbN that is not +0, bit2: a rheofact != 1
void evp_launch_cgrid_call_setup(const EvpCgrid &A, double *facE, double *facN, unsigned *flags, hipStream_t st);
void evp_launch_cgrid_zero_cells(const EvpCgrid &A, const int *cells, int n, hipStream_t st);

// Preparation phase of evp() for grid_ice = 'C' on the device (evp_cgrid_prep.hip; ice_dyn_evp.F90:430-453, 479-490,
// 563-691): the T -> U / E / N averages and dyn_prep2 at U, N and E points in one launch, after dyn_prep1 and the T-grid
// halo updates of the B-grid preparation (evp_prep.hip: prep1, halo_center).
struct EvpCgPrep {
    int nx, ny;
    size_t plane, n;              // n = plane * nblocks (stride of the four mask words in m4)
    const int4 *blk;
    const double *t[11];          // T-grid fields after their halo update (order of EvpPrep::t)
    const double *tmass, *maskd;  // dyn_prep1's products (maskd: iceTmask as 0/1, halo-updated)
    const double *hm, *tarea, *uarea, *earea, *narea;
    const uint8_t *xmask[3];      // umaskCD, emask, nmask
    const double *fcor[3];        // fcor_blk, fcorE_blk, fcorN_blk
    int32_t *m4;                  // iceTmask | iceUmask | iceEmask | iceNmask words: U/E/N in = previous call's, out = new
    double *f[14];                // the loop's state (first 14 of CF_*)
    double *in[CG_NIN];           // the loop's per-call inputs (CI_*; strength untouched)
    double dt, gravit, dyn_area_min, dyn_mass_min, cosw, sinw;
    int ssh_coupled;
};
void evp_launch_cgrid_prep(const EvpCgPrep &P, int nblocks, hipStream_t st);
// seabed_stress_factor_LKD at E and N points (grid_location 'E' / 'N'): TbE, TbN from aice, vice, hwater and mask bits 2, 3
void evp_launch_cgrid_seabed_lkd(const EvpCgPrep &P, int nblocks, const uint8_t *mask, const double *hwater, double k1, double k2,
                                 double alphab, double threshold_hw, hipStream_t st);
// seabed_stress_factor_prob, C-grid tail (ice_dyn_shared.F90:1656-1676): TbE / TbN = max of Tbt over the two T-cells of a face
void evp_launch_cgrid_seabed_prob_faces(const EvpCgPrep &P, int nblocks, const uint8_t *mask, const double *Tbt, hipStream_t st);
