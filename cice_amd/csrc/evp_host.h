// =====================================================================
// Host side of the MI355X-native EVP core, shared declarations: the device
// state behind the C ABI (include/cice_evp_hip.h) and the helpers its
// translation units call on each other.
//   evp_api.cpp            C ABI: init / upload / subcycle / download / run / introspection
//   evp_host_common.cpp    state, copies, static metric terms, lists, kernel arguments
//   evp_host_loop.cpp      velocity halo, tile split, the subcycle loop as enqueued work
//   evp_host_resident.cpp  on-chip resident kernels: set-up, launch, checks, autotuning
//   evp_host_mailbox.cpp   mailbox halo over HIP IPC, RCCL bootstrap, probes
//   evp_host_prep.cpp      preparation phase of evp() (f-2)
//   evp_host_cgrid.cpp     C-grid subcycle (f-4): state, ghost images, the loop
//
// HBM layout: structure-of-arrays; every field is one contiguous fp64 array
// (nx_block, ny_block, nblocks), i fastest -- the memory image of the CICE
// module arrays, so H2D/D2H are straight copies of blocks 1..nblocks.
// State that the subcycle rewrites (uvel, vvel, 12 stresses) exists twice
// (ping-pong, see evp_kernels.hip); everything else once.
// =====================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <algorithm>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/cice_evp_hip.h"
#include "../../include/cice_evp_hip_testing.h"   // (declarations; the definitions exist under CICE_EVP_HIP_TESTING only)
#include "evp_device.h"
#include "halo_plan.h"

namespace evp_host {

extern std::string g_err;
int fail(int code, const char *fmt, ...);

#define HIPC(call)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((int)e_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                              \
    } while (0)

#define NCCLC(call)                                                                              \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess)                                                                   \
            return fail(1000 + (int)r_, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

// order of the 32-entry field table == argument order of cice_evp_hip_run
enum Field {
    F_SIG0 = 0,   // 0..11 stressp_1..4, stressm_1..4, stress12_1..4
    F_STRENGTH = 12, F_CW, F_AIX, F_UOCN, F_VOCN, F_WATERX, F_WATERY, F_FORCEX, F_FORCEY,
    F_UMASSDTI, F_FM, F_STRINTX, F_STRINTY, F_TBU, F_TAUBX, F_TAUBY, F_UVEL, F_VVEL,
    F_UVEL_INIT, F_VVEL_INIT, F_COUNT
};

#define EVP_RES2_COOP_DEFAULT 0      // rim T-cells by corners in the resident B-grid kernel: the product's choice where it is possible

struct State {
    bool ready = false;
    // A rank that holds NO blocks (the reference allows it: a cartesian distribution whose processor grid does not divide
    // the block grid, land-block elimination): it takes part in the bootstrap's collectives -- RCCL communicator, blob
    // all-gather, the agreements -- and in nothing else; every other entry point refuses it ("rank holds no blocks": the
    // host has nothing to hand over there, cice_amd/fortran skips its calls).
    bool bystander = false;
    bool uploaded = false;
    cice_evp_hip_dims d{};
    cice_evp_hip_params prm{};
    std::vector<int32_t> ilo, ihi, jlo, jhi, iglob0, jglob0;
    std::vector<int32_t> gtab[6];   // the global block table (gi0 gj0 gnx gny gowner glocal): plans built after init need it
    int device = 0;
    size_t plane = 0, n = 0;     // nx*ny, nx*ny*nblocks
    int max_ni = 0, max_nj = 0;
    int tyb = 4;
    bool tyb_forced = false, tuned = false;
    hipStream_t stream = nullptr, stream_comm = nullptr;
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    bool overlap = true;
    // tiles that produce cells other ranks need (run first) / all other tiles, per tile variant
    struct TileSplit { int *d_boundary = nullptr, *d_interior = nullptr, *d_all = nullptr; int nb = 0, ni = 0; };
    std::map<int, TileSplit> splits;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, evm[2] = {nullptr, nullptr};
    bool marked[2] = {false, false};

    // device arrays
    double *stat[10] = {};      // dxT dyT dxhy dyhx cxp cyp cxm cym DminTarea uarear
    double *in[F_COUNT] = {};   // per-call inputs + diagnostics (entries of ping-ponged fields unused)
    double *u[2] = {}, *v[2] = {};
    double *sig[2][12] = {};
    double *sig_snap[12] = {};                     // pre-call stresses for the replay of a resident-kernel call (keep_sig)
    double *hte = nullptr, *htn = nullptr;   // edge lengths for in-kernel metric terms
    double *vrelfac = nullptr;               // (aiX*rhow)*Cw, rebuilt at every upload
    double *post_geo[3] = {};                // dxU dyU tarear (next tier f-1)
    double *post_out[7] = {};                // divu shear vort rdg_conv rdg_shear strocnx strocny
    bool have_post_geo = false;
    uint8_t *mask = nullptr;
    int4 *blk = nullptr;
    int cur = 0;
    unsigned flags = 0;          // EVP_F_* in effect
    unsigned flags_allowed = ~0u;
    int *push = nullptr;         // halo push table (device)
    int push_ni = 0, push_nj = 0;
    bool push_ok = false;

    HaloPlan plan;
    int32_t *h_local_dst = nullptr, *h_local_src = nullptr;
    int8_t *h_local_sign = nullptr;
    int n_local = 0;
    // remote halo
    ncclComm_t comm = nullptr;
    bool have_comm = false;
    // test hook (cice_evp_hip_set_test_transport): the marching path's exchanges and agreements through host
    // buffers and caller-supplied callbacks instead of RCCL, so that its several-rank form can run as processes sharing
    // ONE GPU (RCCL refuses two ranks per device)
    cice_evp_hip_test_xchg_fn test_xchg = nullptr;
    cice_evp_hip_test_reduce_fn test_reduce = nullptr;
    void *test_user = nullptr;
    std::vector<double> test_send, test_recv;
    int32_t *h_seam_a = nullptr, *h_seam_b = nullptr, *h_seam_pole = nullptr, *h_late_dst = nullptr,
            *h_late_src = nullptr;
    int8_t *h_late_sign = nullptr;
    int n_seam = 0, n_pole = 0, n_late = 0;
    int32_t *h_fin_dst = nullptr, *h_fin_a = nullptr, *h_fin_b = nullptr;   // general seam step (HaloPlan::fin_*)
    int8_t *h_fin_coef = nullptr;
    int n_fin = 0;
    size_t nuv = 0;              // elements of a velocity buffer: n + staging slots of the seam step
    int32_t *h_stress_dst = nullptr, *h_stress_src = nullptr;
    int n_stress = 0;
    int32_t *h_stress_own_dst = nullptr, *h_stress_own_src = nullptr;     // tripoleT (halo_plan.h)
    int n_stress_own = 0;
    int32_t *h_stress_corner_dst = nullptr, *h_stress_corner_src = nullptr;
    int n_stress_corner = 0;
    // preparation phase on the device (evp_prep.hip)
    struct FoldX {                 // device side of HaloPlan::center_foldr_dst / stress_foldr_dst / fold_shift_cells
        bool ready = false;
        int32_t *cells = nullptr, *dst[2] = {nullptr, nullptr}, *seam_dst = nullptr, *seam_slot = nullptr;
        int8_t *seam_one = nullptr;
        int n_cells = 0, n_dst[2] = {0, 0}, n_seam = 0;
        double *scr[2] = {nullptr, nullptr};
    } foldx;
    struct Prep {
        bool geo = false;
        uint8_t *tmask = nullptr, *umask = nullptr, *umask_old = nullptr, *tmphm = nullptr;
        int32_t *umask_old32 = nullptr;          // the caller's words as uploaded (reduced to bytes on the device)
        double *hm = nullptr, *tarea = nullptr, *uarea = nullptr, *fcor = nullptr, *hwater = nullptr;
        double *aicen = nullptr, *vicen = nullptr, *tbt = nullptr;     // seabed_stress_factor_prob: category arrays, factor at T points
        int ncat = 0;
        double *t[11] = {};
        double *tmass = nullptr, *umass = nullptr, *maskd = nullptr;
        double *ss_tltxU = nullptr, *ss_tltyU = nullptr, *strairxU = nullptr, *strairyU = nullptr,
               *strtltx = nullptr, *strtlty = nullptr;
        unsigned *flagword = nullptr;
        int32_t *c_dst = nullptr, *c_src = nullptr;
        int8_t *c_vsign = nullptr;
        int n_center = 0;
        // tripoleT: the fold step of the cell-centre fields (halo_plan.h: center_tf_*)
        int32_t *tf_dst = nullptr, *tf_a = nullptr, *tf_b = nullptr;
        uint8_t *tf_flip = nullptr;
        double *tf_tmp = nullptr;
        int n_tf = 0;
        std::vector<uint8_t> h8;
        double t_ms = 0;
    } prep;
    int32_t *h_send_src = nullptr, *h_recv_dst = nullptr;
    int8_t *h_recv_sign = nullptr;
    double *sendbuf = nullptr, *recvbuf = nullptr;
    int n_send = 0, n_recv = 0;
    // masked halo of the in-loop velocity exchange (ice_HaloMask, ice_boundary.F90:889-1062): the entries whose
    // halomask is set, compacted; the unmasked lists serve every other exchange
    struct Masked {
        bool on = false;
        int n_send = 0, n_recv = 0;
        std::vector<int> peer_nsend, peer_nrecv;
        int32_t *send_src = nullptr, *recv_dst = nullptr, *recv_slot = nullptr;
        int8_t *recv_sign = nullptr;
        double **send_addr = nullptr;
        unsigned *send_pstride = nullptr;
        std::vector<double *> h_send_addr;        // host copies of the unmasked mailbox tables (kept at import)
        std::vector<unsigned> h_send_pstride;
    } msk;

    // mailbox halo (evp_halo_direct.hip): peers' inboxes mapped through HIP IPC
    struct Direct {
        bool on = false;             // use it for the remote halo
        bool exported = false;
        void *mailbox = nullptr;     // [flags][seq][err][inbox x 2 parities]
        size_t bytes = 0, inbox_off = 0, rec_off = 0, raw_off = 0;
        std::vector<void *> opened;  // hipIpcOpenMemHandle results
        EvpDirect *d_dx = nullptr;   // device copy of the argument block (exchange riding in the subcycle launch)
        EvpDirect *d_dx_m = nullptr; // the same with the masked lists (cice_evp_hip_halo_mask)
        unsigned *d_cnt = nullptr;   // [0] boundary tiles checked in, [16] launches with a riding exchange
        double **send_addr = nullptr;
        unsigned *send_pstride = nullptr;
        unsigned **peer_flag = nullptr;
        std::string why;             // why it is off
    } direct;

    // captured loops.  A capture stores the kernel arguments BY VALUE, so everything a launch bakes in
    // that can change between calls is part of the key: (ndte, cur) and the EVP_F_* flags in effect
    // (TBU_ZERO / WATER_IS_OCN are re-derived from the data at every upload / prep)
    std::map<std::tuple<int, int, unsigned>, hipGraphExec_t> graphs;
    bool use_graph = true;

    // on-chip resident subcycle (evp_resident2.hip)
    int res_mode = -1;           // -1 undecided, 0 off, 1 on
    bool res_forced = false;
    int *res_err = nullptr;
    double **res_tab = nullptr;  // device pointer table (EvpResident2::tab)
    int res_ntiles = 0;
    int4 *res2_ring = nullptr;
    int *res2_cnt = nullptr;
    uint8_t *res2_pub = nullptr;
    // 16 x 16 tiles (rim wave / interior waves): thread -> cell map, waves that wait for the ring, chunks with ice
    uint8_t *res2_perm = nullptr, *res2_late = nullptr, *res2_nact = nullptr, *res2_nlate = nullptr;
    // only the tiles that hold ice run (resident2_order): how many, which (device flags), the tile of every U-cell, the tiles
    // that always run (cells of the tripole fold row change in every subcycle, ice or not)
    bool res_ran = true;           // the last cice_evp_hip_subcycle ran inside the resident kernel (false: that call's ice did not fit)
    int res2_nlive = 0;
    uint8_t *res2_live = nullptr;
    int *res2_celltile = nullptr;
    std::vector<char> res2_always_h;
    int res2_coop = -1;            // rim T-cells by corners (evp_resident2.hip COOP): -1 undecided, 0 off, 1 on
    bool res2_coop_ok = false;     // ... possible for the current tables (every tile's rim list fits 64 quads)
    int *res2_cuload = nullptr;                           // per-CU record of a launch (EvpResident2::cuload)
    unsigned long long *res2_prof = nullptr;              // phase stamps (CICE_EVP_HIP_RES_PROF=1)
    std::vector<uint8_t> res2_cls_h;                      // per tile and cell position: 0 not computed, 1 reads no ring velocity, 2 does
    std::vector<int> res2_cnt_h;                          // ring entries per tile (host copy)
    void *res2_rec[2] = {nullptr, nullptr};
    int res2_logw = 0, res2_ntiles = 0;
    unsigned res2_epoch = 0;
    int res2_par = 0;            // record buffer in which the next launch starts (EvpResident2::par0)
    bool res2_rec_owned = true;  // false: the record buffers live inside the mailbox allocation
    // resident kernel with neighbours on other GPUs (records stored into peers' buffers over xGMI)
    bool res_remote = false;     // agreed by all ranks at mailbox import
    double res_timeout_ms = 0;   // > 0: overrides the wait bound of the next resident launches (probe)
    int *res2_order = nullptr;                        // launch order of the tiles (heaviest first), per upload
    int res2_order_for = -1;                          // logw the order was built for
    bool res2_order_stale = true;                     // masks changed since it was built
    int *res2_seam = nullptr, *res2_img3 = nullptr;   // tripole: fold-row roles, per-cell ghost images
    void *res2_rec_raw[2] = {nullptr, nullptr};       // tripole: records of the pre-average fold-row velocities
    bool res2_raw_owned = true;                       // false: they live inside the mailbox allocation (fold row split over ranks)
    int2 *res2_rraw = nullptr;                        // EvpResident2::rraw / peer_raw / peer_raw_stride
    void **res2_peer_raw = nullptr;
    size_t *res2_peer_raw_stride = nullptr;
    int2 *res2_rimg = nullptr;
    void **res2_peer_rec = nullptr;
    size_t *res2_peer_rstride = nullptr;
    bool res_launched = false;   // an un-checked launch is in flight
    int res_fallbacks = 0;       // calls repeated with the streaming kernel after a resident launch gave up
    double t_res_probe_ms = 0, t_stream_probe_ms = 0;

    double t_loop_ms = 0, t_h2d_ms = 0, t_d2h_ms = 0;
    int t_nsub = 0;
    std::vector<uint8_t> hmask, hmask_prev;   // masks of this / the previous upload (what depends on them is rebuilt only when they change)
    struct Pinned { size_t bytes; void *dev; };    // dev: the range as the device sees it (NULL: not mapped)
    std::map<const void *, Pinned> pinned;         // host ranges registered by cice_evp_hip_pin_host
    // stresses that stay on the device between calls of cice_evp_hip_run (CICE_EVP_HIP_OPT_STRESS_RESIDENT)
    // several subcycles per pass over a device-private rectangle layout (evp_march.hip, evp_host_march.cpp)
    struct March {
        int mode = -1;                 // -1 undecided, 0 off, 1 on
        EvpMarchGeo G{};
        std::vector<int> blkid_h;
        std::vector<int2> org_h;
        int nstrips = 0, nseg = 0, seglen = 0, nitems = 0;
        int kpass = EVP_MARCH_KMAX;    // subcycles a full pass advances the state by
        int ring_valid = EVP_MARCH_PAD;// cells beyond the rank's own that are current after an exchange of the ring (ext + P):
                                       // that many subcycles can follow before the next exchange
        long subcycles = 0;            // subcycles advanced by passes since init
        int call_passes = 0, call_subcycles = 0;   // ... of the last call (launches per subcycle for the timing read-out)
        size_t nblk = 0;               // (row, strip) blocks per buffer
        std::vector<unsigned> dup_h;   // [nstrips][64] duplicate positions (EvpMarch::dup)
        bool stat_done = false, stat_ok = false;
        unsigned checked_seq = ~0u;    // upload_seq whose ghost-cell consistency has been verified
        long passes = 0;               // passes run since init
        int declined = 0;              // calls handed to the one-subcycle kernels (consistency check failed)
        bool last_call = false;        // the last cice_evp_hip_subcycle went through this path
        std::string why;
        int direct = -1;               // the ring between ranks as stores into HIP-IPC-mapped inboxes: -1 not tried, 0 off, 1 on, 2 being verified
        std::string direct_why;        // ... and why it is off
        int direct_asked = -1;         // value of CICE_EVP_HIP_MARCH_DIRECT the decision was taken on (a change re-opens it)
    } march;
    unsigned upload_seq = 0;                       // bumped whenever the caller hands new state / inputs to the device
    int fault_calls = 0;                           // test hook counter (fault_hook, evp_api.cpp)
    bool opt_sig_resident = false;
    bool sig_valid = false;                        // sig[cur] holds what the caller's arrays would hold
    bool lean_diag = false;                        // this call: strintx/y, taubx/y not uploaded; written back on ice U-cells only
};

extern State S;

inline const char *env(const char *k) { return std::getenv(k); }
// experiment / fault-injection / A-B switches: read in the TEST build (-DCICE_EVP_HIP_TESTING -> libcice_evp_hip_testing.so)
// only; the production library does not even carry their names (a macro, so that the literal is never emitted): what it
// reads is DEVICE, VERBOSE, HALO, HALO_TIMEOUT_MS, RESIDENT, MARCH, CGRID_ONE, CGRID_RESIDENT (include/cice_evp_hip.h)
#ifdef CICE_EVP_HIP_TESTING
#define env_test(k) (std::getenv(k))
#else
#define env_test(k) (static_cast<const char *>(nullptr))
#endif

// mailbox layout: EVP_DIRECT_MAXPEER flag lines, then seq, err, then the inbox
constexpr size_t DIRECT_SEQ_OFF = (size_t)EVP_DIRECT_MAXPEER * EVP_DIRECT_FLAG_STRIDE * sizeof(unsigned);
constexpr size_t DIRECT_ERR_OFF = DIRECT_SEQ_OFF + 64;
constexpr size_t DIRECT_INBOX_OFF = DIRECT_ERR_OFF + 64;

uint64_t host_identity();   // evp_host_mailbox.cpp
// evp_host_common.cpp
int alloc_d(double **p, size_t n);
void free_all();
int h2d(double *dst, const double *src);
int d2h(double *dst, const double *src);
// batched variants: arrays the caller page-locked travel in ONE gather / scatter launch, the rest as copies
struct CopyBatch { std::vector<std::pair<double *, const double *>> items; };
int h2d_batch(CopyBatch &B);
int d2h_batch(CopyBatch &B);
bool batch_mapped(const CopyBatch &B, bool to_device);
int d2h_batch_masked(CopyBatch &B, unsigned bit);
int derive_metrics(const double *HTE, const double *HTN, const double *dxT, const double *dyT,
                   const double *uarear, const double *tarea);
int upload_lists();
int build_push_table();
void fill_args(EvpArgs &A, int cur, int last);
int cap_mode();
// evp_host_loop.cpp
void fill_direct(EvpDirect &D);
int halo_remote_pair(double *a, double *bb, bool masked = false, bool has_tail = false);
// centre-kind fields across a tripole fold whose row is split over ranks (halo_plan.h): dstA[d] <- fa * (the exchange of the
// shifted copy of srcA)[d] for d in the list, likewise B.  kind 0: the centre-field ghost cells of the preparation phase,
// 1: the ghost row of the stress symmetrisation
int fold_remote_pair(const double *srcA, const double *srcB, double *dstA, double *dstB, int kind, double fa, double fb);
// after a plain exchange of two centre-kind arrays (with staging tail): their east-west ghost cells of row NY <- the raw
// values the exchange left in the staging slots
int fold_seam_ghosts(double *a, double *b);
void fill_direct(EvpDirect &D, bool masked);
int halo_uv(int b, bool masked = false);
bool use_overlap();
bool use_riding_exchange();
int get_tile_split(int variant, State::TileSplit **out);
int enqueue_loop(int ndte, int cur0);
// evp_host_resident.cpp
bool tripole_seam();
bool resident_possible(bool with_peers = false);
int resident2_setup(int logw);
bool resident2_fits(bool remote = false);
bool resident2_fits_now();
int resident2_order();
int resident_tables();
int launch_resident2(int ndte, int cur0, bool dry);
int resident_check_error();
int tune_after_upload();
// evp_host_march.cpp
bool march_wanted();
int march_run(int ndte);
void march_free();
int march_direct_error();     // a ring neighbour never signalled (direct exchange of the marching path)
// evp_host_mailbox.cpp
int direct_check_error();
// evp_host_cgrid.cpp
void cgrid_free();

}  // namespace evp_host
