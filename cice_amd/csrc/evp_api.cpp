// =====================================================================
// C ABI of the MI355X-native EVP core (see include/cice_evp_hip.h) and the
// device-state management behind it.
//
// HBM layout: structure-of-arrays; every field is one contiguous fp64 array
// (nx_block, ny_block, nblocks), i fastest -- the memory image of the CICE
// module arrays, so H2D/D2H are straight copies of blocks 1..nblocks.
// State that the subcycle rewrites (uvel, vvel, 12 stresses) exists twice
// (ping-pong, see evp_kernels.hip); everything else once.
// =====================================================================
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/cice_evp_hip.h"
#include "evp_device.h"
#include "halo_plan.h"

// host arithmetic of derive_metrics must round every operation (no FMA)
#pragma clang fp contract(off)

namespace {

std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code ? code : -1;
}

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

struct State {
    bool ready = false;
    bool uploaded = false;
    cice_evp_hip_dims d{};
    cice_evp_hip_params prm{};
    std::vector<int32_t> ilo, ihi, jlo, jhi, iglob0, jglob0;
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
    int32_t *h_seam_a = nullptr, *h_seam_b = nullptr, *h_seam_pole = nullptr, *h_late_dst = nullptr,
            *h_late_src = nullptr;
    int8_t *h_late_sign = nullptr;
    int n_seam = 0, n_pole = 0, n_late = 0;
    int32_t *h_stress_dst = nullptr, *h_stress_src = nullptr;
    int n_stress = 0;
    // preparation phase on the device (evp_prep.hip)
    struct Prep {
        bool geo = false;
        uint8_t *tmask = nullptr, *umask = nullptr, *umask_old = nullptr, *tmphm = nullptr;
        double *hm = nullptr, *tarea = nullptr, *uarea = nullptr, *fcor = nullptr;
        double *t[11] = {};
        double *tmass = nullptr, *umass = nullptr, *maskd = nullptr;
        double *ss_tltxU = nullptr, *ss_tltyU = nullptr, *strairxU = nullptr, *strairyU = nullptr,
               *strtltx = nullptr, *strtlty = nullptr;
        unsigned *flagword = nullptr;
        int32_t *c_dst = nullptr, *c_src = nullptr;
        int8_t *c_vsign = nullptr;
        int n_center = 0;
        std::vector<uint8_t> h8;
        double t_ms = 0;
    } prep;
    int32_t *h_send_src = nullptr, *h_recv_dst = nullptr;
    int8_t *h_recv_sign = nullptr;
    double *sendbuf = nullptr, *recvbuf = nullptr;
    int n_send = 0, n_recv = 0;

    // mailbox halo (evp_halo_direct.hip): peers' inboxes mapped through HIP IPC
    struct Direct {
        bool on = false;             // use it for the remote halo
        bool exported = false;
        void *mailbox = nullptr;     // [flags][seq][err][inbox x 2 parities]
        size_t bytes = 0, inbox_off = 0, rec_off = 0;
        std::vector<void *> opened;  // hipIpcOpenMemHandle results
        EvpDirect *d_dx = nullptr;   // device copy of the argument block (exchange riding in the subcycle launch)
        unsigned *d_cnt = nullptr;   // [0] boundary tiles checked in, [16] launches with a riding exchange
        double **send_addr = nullptr;
        unsigned *send_pstride = nullptr;
        unsigned **peer_flag = nullptr;
        std::string why;             // why it is off
    } direct;

    std::map<std::pair<int, int>, hipGraphExec_t> graphs;   // (ndte, cur) -> captured loop
    bool use_graph = true;

    // on-chip resident subcycle (evp_resident.hip)
    int res_mode = -1;           // -1 undecided, 0 off, 1 on
    bool res_forced = false;
    int *res_flags = nullptr, *res_nbr = nullptr, *res_err = nullptr;
    double **res_tab = nullptr;  // device pointer table (EvpResident::tab)
    double *res_scratch[4] = {}; // u,v ping-pong copies for the dry probe
    int res_ntiles = 0, res_logw = 6;
    int res_gen = 1;             // 1: flags (evp_resident.hip), 2: tagged records (evp_resident2.hip)
    int4 *res2_ring = nullptr;
    int *res2_cnt = nullptr;
    uint8_t *res2_pub = nullptr;
    void *res2_rec[2] = {nullptr, nullptr};
    int res2_logw = 0, res2_ntiles = 0;
    unsigned res2_epoch = 0;
    int res2_par = 0;            // record buffer in which the next launch starts (EvpResident2::par0)
    bool res2_rec_owned = true;  // false: the record buffers live inside the mailbox allocation
    // resident kernel with neighbours on other GPUs (records stored into peers' buffers over xGMI)
    bool res_remote = false;     // agreed by all ranks at mailbox import
    double res_timeout_ms = 0;   // > 0: overrides the wait bound of the next resident launches (probe)
    int2 *res2_rimg = nullptr;
    void **res2_peer_rec = nullptr;
    size_t *res2_peer_rstride = nullptr;
    bool res_launched = false;   // an un-checked launch is in flight
    double t_res_probe_ms = 0, t_stream_probe_ms = 0;

    double t_loop_ms = 0, t_h2d_ms = 0, t_d2h_ms = 0;
    int t_nsub = 0;
    std::vector<uint8_t> hmask;
    std::map<const void *, size_t> pinned;   // host ranges registered by cice_evp_hip_pin_host
};

State S;

const char *env(const char *k) { return std::getenv(k); }

// mailbox layout: EVP_DIRECT_MAXPEER flag lines, then seq, err, then the inbox
constexpr size_t DIRECT_SEQ_OFF = (size_t)EVP_DIRECT_MAXPEER * EVP_DIRECT_FLAG_STRIDE * sizeof(unsigned);
constexpr size_t DIRECT_ERR_OFF = DIRECT_SEQ_OFF + 64;
constexpr size_t DIRECT_INBOX_OFF = DIRECT_ERR_OFF + 64;
void fill_direct(EvpDirect &D);

int alloc_d(double **p, size_t n)
{
    HIPC(hipMalloc((void **)p, n * sizeof(double)));
    HIPC(hipMemsetAsync(*p, 0, n * sizeof(double), S.stream));
    return 0;
}

void free_all()
{
    auto F = [](auto *&p) {
        if (p) (void)hipFree((void *)p);
        p = nullptr;
    };
    for (auto &p : S.stat) F(p);
    for (auto &p : S.in) F(p);
    for (int k = 0; k < 2; ++k) {
        F(S.u[k]);
        F(S.v[k]);
        for (auto &p : S.sig[k]) F(p);
    }
    F(S.hte);
    F(S.htn);
    F(S.vrelfac);
    F(S.res_flags); F(S.res_nbr); F(S.res_err); F(S.res_tab);
    F(S.res2_ring); F(S.res2_cnt); F(S.res2_pub);
    if (S.res2_rec_owned) { F(S.res2_rec[0]); F(S.res2_rec[1]); }
    S.res2_rec[0] = S.res2_rec[1] = nullptr;
    S.res2_rec_owned = true;
    S.res_remote = false;
    S.res2_par = 0; S.res2_epoch = 0;
    F(S.res2_rimg); F(S.res2_peer_rec); F(S.res2_peer_rstride);
    for (auto &p : S.res_scratch) F(p);
    for (auto &p : S.post_geo) F(p);
    for (auto &p : S.post_out) F(p);
    F(S.push);
    F(S.mask);
    F(S.blk);
    F(S.h_local_dst);
    F(S.h_local_src);
    F(S.h_local_sign);
    F(S.h_seam_a); F(S.h_seam_b); F(S.h_seam_pole); F(S.h_late_dst); F(S.h_late_src); F(S.h_late_sign);
    F(S.h_stress_dst); F(S.h_stress_src);
    {
        State::Prep &Q = S.prep;
        F(Q.tmask); F(Q.umask); F(Q.umask_old); F(Q.tmphm); F(Q.hm); F(Q.tarea); F(Q.uarea); F(Q.fcor);
        for (auto &q : Q.t) F(q);
        F(Q.tmass); F(Q.umass); F(Q.maskd); F(Q.ss_tltxU); F(Q.ss_tltyU); F(Q.strairxU); F(Q.strairyU);
        F(Q.strtltx); F(Q.strtlty); F(Q.flagword); F(Q.c_dst); F(Q.c_src); F(Q.c_vsign);
        S.prep = State::Prep();
    }
    F(S.h_send_src);
    F(S.h_recv_dst);
    F(S.h_recv_sign);
    F(S.sendbuf);
    F(S.recvbuf);
    for (void *q : S.direct.opened) (void)hipIpcCloseMemHandle(q);
    F(S.direct.mailbox); F(S.direct.d_dx); F(S.direct.d_cnt); F(S.direct.send_addr); F(S.direct.send_pstride); F(S.direct.peer_flag);
    S.direct = State::Direct();
    for (auto &kv : S.graphs) (void)hipGraphExecDestroy(kv.second);
    S.graphs.clear();
    if (S.ev0) (void)hipEventDestroy(S.ev0);
    if (S.ev1) (void)hipEventDestroy(S.ev1);
    if (S.ev2) (void)hipEventDestroy(S.ev2);
    if (S.ev3) (void)hipEventDestroy(S.ev3);
    for (auto &e : S.evm) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    S.ev0 = S.ev1 = S.ev2 = S.ev3 = nullptr;
    if (S.have_comm) (void)ncclCommDestroy(S.comm);
    S.have_comm = false;
    for (auto &kv : S.pinned) (void)hipHostUnregister(const_cast<void *>(kv.first));
    S.pinned.clear();
    for (auto &kv : S.splits) {
        if (kv.second.d_boundary) (void)hipFree(kv.second.d_boundary);
        if (kv.second.d_interior) (void)hipFree(kv.second.d_interior);
        if (kv.second.d_all) (void)hipFree(kv.second.d_all);
    }
    S.splits.clear();
    if (S.ev_pack) (void)hipEventDestroy(S.ev_pack);
    if (S.ev_halo) (void)hipEventDestroy(S.ev_halo);
    S.ev_pack = S.ev_halo = nullptr;
    if (S.stream_comm) (void)hipStreamDestroy(S.stream_comm);
    S.stream_comm = nullptr;
    if (S.stream) (void)hipStreamDestroy(S.stream);
    S.stream = nullptr;
}

// Copies blocks 1..nblocks of a host (nx,ny,max_blocks) array: contiguous prefix.
int h2d(double *dst, const double *src)
{
    HIPC(hipMemcpyAsync(dst, src, S.n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    return 0;
}
int d2h(double *dst, const double *src)
{
    HIPC(hipMemcpyAsync(dst, src, S.n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    return 0;
}

// Static metric terms, host arithmetic in the reference's operation order
// (init_dyn_shared, ice_dyn_shared.F90:384-388, 401-441).  dxhy/dyhx are
// evaluated directly on the N/E ghost T-cells from the HTE/HTN ghost values
// (which CICE defines from the global arrays, ice_grid.F90:662-666) instead of
// through a halo update: same operands, same result for every cell that can
// hold ice.
int derive_metrics(const double *HTE, const double *HTN, const double *dxT, const double *dyT,
                   const double *uarear, const double *tarea)
{
    const int nx = S.d.nx_block;
    const size_t plane = S.plane;
    std::vector<std::vector<double>> m(7, std::vector<double>(S.n, 0.0));   // cxp cyp cxm cym dxhy dyhx Dmin
    const double p5 = 0.5, c1p5 = 1.5;
    for (int b = 0; b < S.d.nblocks; ++b) {
        const double *hte = HTE + b * plane, *htn = HTN + b * plane;
        for (size_t k = 0; k < plane; ++k) m[6][b * plane + k] = S.prm.deltaminEVP * tarea[b * plane + k];
        for (int j = S.jlo[b]; j <= S.jhi[b] + 1; ++j)
            for (int i = S.ilo[b]; i <= S.ihi[b] + 1; ++i) {
                const size_t c = (size_t)(j - 1) * nx + (i - 1);
                const size_t g = b * plane + c;
                m[0][g] = (c1p5 * htn[c] - p5 * htn[c - nx]);        // cxp
                m[1][g] = (c1p5 * hte[c] - p5 * hte[c - 1]);         // cyp
                m[2][g] = -(c1p5 * htn[c - nx] - p5 * htn[c]);       // cxm
                m[3][g] = -(c1p5 * hte[c - 1] - p5 * hte[c]);        // cym
                m[4][g] = p5 * (hte[c] - hte[c - 1]);                // dxhy
                m[5][g] = p5 * (htn[c] - htn[c - nx]);               // dyhx
            }
    }
    if (h2d(S.stat[0], dxT) || h2d(S.stat[1], dyT) || h2d(S.stat[9], uarear)) return -1;
    if (h2d(S.hte, HTE) || h2d(S.htn, HTN)) return -1;
    // in-kernel metric terms need tarea == dxT*dyT bit for bit (ice_grid.F90:681)
    bool same = true;
    for (size_t k = 0; k < S.n && same; ++k) same = (tarea[k] == dxT[k] * dyT[k]);
    if (same) S.flags |= EVP_F_METRICS;
    // on the tripole ghost row dxhy/dyhx are mirrored interior values (halo update with sign,
    // ice_dyn_shared.F90:412-417), not a local difference: keep them as arrays there
    if (S.d.ns_boundary_type == CICE_EVP_BND_TRIPOLE) S.flags |= EVP_F_DXHY_ARRAY;
    const int order[7] = {4, 5, 6, 7, 2, 3, 8};   // stat slots of cxp cyp cxm cym dxhy dyhx Dmin
    for (int k = 0; k < 7; ++k)
        if (h2d(S.stat[order[k]], m[k].data())) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

int upload_lists()
{
    const HaloPlan &P = S.plan;
    S.n_local = (int)P.local_dst.size();
    if (S.n_local) {
        HIPC(hipMalloc((void **)&S.h_local_dst, S.n_local * sizeof(int32_t)));
        HIPC(hipMalloc((void **)&S.h_local_src, S.n_local * sizeof(int32_t)));
        HIPC(hipMalloc((void **)&S.h_local_sign, S.n_local));
        HIPC(hipMemcpy(S.h_local_dst, P.local_dst.data(), S.n_local * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(S.h_local_src, P.local_src.data(), S.n_local * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(S.h_local_sign, P.local_sign.data(), S.n_local, hipMemcpyHostToDevice));
    }
    auto up32 = [&](const std::vector<int32_t> &v, int32_t *&dptr) -> int {
        if (v.empty()) return 0;
        HIPC(hipMalloc((void **)&dptr, v.size() * sizeof(int32_t)));
        HIPC(hipMemcpy(dptr, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        return 0;
    };
    S.n_seam = (int)P.seam_a.size();
    S.n_pole = (int)P.seam_pole.size();
    S.n_late = (int)P.late_dst.size();
    if (up32(P.seam_a, S.h_seam_a) || up32(P.seam_b, S.h_seam_b) || up32(P.seam_pole, S.h_seam_pole) ||
        up32(P.late_dst, S.h_late_dst) || up32(P.late_src, S.h_late_src)) return -1;
    S.n_stress = (int)P.stress_dst.size();
    if (up32(P.stress_dst, S.h_stress_dst) || up32(P.stress_src, S.h_stress_src)) return -1;
    if (S.n_late) {
        HIPC(hipMalloc((void **)&S.h_late_sign, S.n_late));
        HIPC(hipMemcpy(S.h_late_sign, P.late_sign.data(), S.n_late, hipMemcpyHostToDevice));
    }
    std::vector<int32_t> ss, rd;
    std::vector<int8_t> rs;
    for (const HaloPeer &p : P.peers) {
        ss.insert(ss.end(), p.send_src.begin(), p.send_src.end());
        rd.insert(rd.end(), p.recv_dst.begin(), p.recv_dst.end());
        rs.insert(rs.end(), p.recv_sign.begin(), p.recv_sign.end());
    }
    S.n_send = (int)ss.size();
    S.n_recv = (int)rd.size();
    if (S.n_send) {
        HIPC(hipMalloc((void **)&S.h_send_src, ss.size() * sizeof(int32_t)));
        HIPC(hipMemcpy(S.h_send_src, ss.data(), ss.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPC(hipMalloc((void **)&S.sendbuf, 2 * ss.size() * sizeof(double)));
    }
    if (S.n_recv) {
        HIPC(hipMalloc((void **)&S.h_recv_dst, rd.size() * sizeof(int32_t)));
        HIPC(hipMalloc((void **)&S.h_recv_sign, rs.size()));
        HIPC(hipMemcpy(S.h_recv_dst, rd.data(), rd.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(S.h_recv_sign, rs.data(), rs.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc((void **)&S.recvbuf, 2 * rd.size() * sizeof(double)));
    }
    return 0;
}

// Inverse of the local part of the halo plan: for every interior edge cell the
// ghost cells that mirror it, so that the thread producing the cell can store the
// images itself.  Per block 2*(nj+ni) edge slots (W, E, S, N) x 2 entries; an entry is
// dst*2 + (sign<0), or -1.  Falls back to the gather kernel if an image does not fit.
int build_push_table()
{
    const HaloPlan &P = S.plan;
    S.push_ok = false;
    S.push_ni = S.max_ni;
    S.push_nj = S.max_nj;
    const int nslot = 2 * (S.push_nj + S.push_ni);
    std::vector<int> tab((size_t)S.d.nblocks * nslot * 2, -1);
    const int nx = S.d.nx_block;
    bool ok = true;
    for (size_t k = 0; k < P.local_dst.size() && ok; ++k) {
        const int src = P.local_src[k];
        if (src < 0) { ok = false; break; }
        const int b = (int)(src / S.plane);
        const int rem = (int)(src % S.plane);
        const int j = rem / nx + 1, i = rem % nx + 1;
        int cand[4];
        cand[0] = (i == S.ilo[b]) ? (j - S.jlo[b]) : -1;
        cand[1] = (i == S.ihi[b]) ? S.push_nj + (j - S.jlo[b]) : -1;
        cand[2] = (j == S.jlo[b]) ? 2 * S.push_nj + (i - S.ilo[b]) : -1;
        cand[3] = (j == S.jhi[b]) ? 2 * S.push_nj + S.push_ni + (i - S.ilo[b]) : -1;
        const int enc = P.local_dst[k] * 2 + (P.local_sign[k] < 0 ? 1 : 0);
        bool placed = false;
        for (int e = 0; e < 4 && !placed; ++e) {
            if (cand[e] < 0) continue;
            for (int w = 0; w < 2 && !placed; ++w) {
                int &slot = tab[((size_t)b * nslot + cand[e]) * 2 + w];
                if (slot < 0) { slot = enc; placed = true; }
            }
        }
        if (!placed) ok = false;
    }
    if (!ok || P.local_dst.empty()) return 0;
    HIPC(hipMalloc((void **)&S.push, tab.size() * sizeof(int)));
    HIPC(hipMemcpy(S.push, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    S.push_ok = true;
    return 0;
}

void fill_args(EvpArgs &A, int cur, int last)
{
    const cice_evp_hip_params &q = S.prm;
    A.p = {q.arlx1i, q.denom1, q.brlx, q.revp, q.e_factor, q.epp2i, q.capping, q.Ktens,
           q.u0, q.cosw, q.sinw, q.rhow};
    A.nx = S.d.nx_block;
    A.ny = S.d.ny_block;
    A.plane = S.plane;
    A.last = last;
    A.tile_list = nullptr;
    A.tile_count = 0;
    A.dx = nullptr; A.dx_count = nullptr; A.dx_fseq = nullptr; A.dx_nb = 0;
    A.blk = S.blk;
    A.mask = S.mask;
    A.u_in = S.u[cur];
    A.v_in = S.v[cur];
    A.u_out = S.u[cur ^ 1];
    A.v_out = S.v[cur ^ 1];
    for (int k = 0; k < 12; ++k) {
        A.sig_in[k] = S.sig[cur][k];
        A.sig_out[k] = S.sig[cur ^ 1][k];
    }
    A.dxT = S.stat[0]; A.dyT = S.stat[1]; A.dxhy = S.stat[2]; A.dyhx = S.stat[3];
    A.cxp = S.stat[4]; A.cyp = S.stat[5]; A.cxm = S.stat[6]; A.cym = S.stat[7];
    A.DminTarea = S.stat[8]; A.uarear = S.stat[9];
    A.HTE = S.hte; A.HTN = S.htn; A.deltaminEVP = q.deltaminEVP;
    A.vrelfac = S.vrelfac;
    A.flags = S.flags & S.flags_allowed;
    if (!S.push_ok) A.flags &= ~EVP_F_PUSH;
    A.push = S.push; A.push_ni = S.push_ni; A.push_nj = S.push_nj;
    A.strength = S.in[F_STRENGTH]; A.Cw = S.in[F_CW]; A.aiX = S.in[F_AIX];
    A.uocn = S.in[F_UOCN]; A.vocn = S.in[F_VOCN]; A.waterx = S.in[F_WATERX];
    A.watery = S.in[F_WATERY]; A.forcex = S.in[F_FORCEX]; A.forcey = S.in[F_FORCEY];
    A.umassdti = S.in[F_UMASSDTI]; A.fm = S.in[F_FM]; A.TbU = S.in[F_TBU];
    A.uvel_init = S.in[F_UVEL_INIT]; A.vvel_init = S.in[F_VVEL_INIT];
    A.strintx = S.in[F_STRINTX]; A.strinty = S.in[F_STRINTY];
    A.taubx = S.in[F_TAUBX]; A.tauby = S.in[F_TAUBY];
}

int cap_mode()
{
    if (S.prm.capping == 1.0) return 1;
    if (S.prm.capping == 0.0) return 0;
    return -1;
}

void fill_direct(EvpDirect &D)
{
    State::Direct &X = S.direct;
    char *base = (char *)X.mailbox;
    D.n_send = S.n_send;
    D.n_recv = S.n_recv;
    D.npeers = (int)S.plan.peers.size();
    D.send_src = S.h_send_src;
    D.send_addr = X.send_addr;
    D.send_pstride = X.send_pstride;
    D.recv_dst = S.h_recv_dst;
    D.recv_sign = (const signed char *)S.h_recv_sign;
    D.flags_in = (unsigned *)base;
    D.seq = (unsigned *)(base + DIRECT_SEQ_OFF);
    D.err = (int *)(base + DIRECT_ERR_OFF);
    D.inbox = (double *)(base + X.inbox_off);
    static const double tmo_ms = env("CICE_EVP_HIP_HALO_TIMEOUT_MS") ? std::atof(env("CICE_EVP_HIP_HALO_TIMEOUT_MS")) : 30000.0;
    D.timeout_ticks = (unsigned long long)(tmo_ms * 1.0e5);     // 100 MHz wall clock
    D.peer_flag = X.peer_flag;
    static const int dbg = env("CICE_EVP_HIP_HALO_DEBUG") ? std::atoi(env("CICE_EVP_HIP_HALO_DEBUG")) : 0;
    D.dbg = dbg;
}

// Ghost cells whose source lives on another rank, for a pair of arrays laid out like uvel/vvel
// (the velocities of the loop; pairs of T-grid fields in the preparation phase on grids without
// a tripole fold, where cell-centre and corner fields mirror the same cells)
int halo_remote_pair(double *a, double *bb)
{
    if (!S.plan.peers.empty() && S.direct.on) {
        EvpDirect D;
        fill_direct(D);
        evp_launch_halo_direct(D, a, bb, S.stream);
    } else if (!S.plan.peers.empty()) {
        if (!S.have_comm) return fail(-2, "remote halo needed but neither cice_evp_hip_comm_init nor cice_evp_hip_halo_import was called");
        evp_launch_halo_pack(a, bb, S.h_send_src, S.sendbuf, S.n_send, S.stream);
        size_t so = 0, ro = 0;
        NCCLC(ncclGroupStart());
        for (const HaloPeer &p : S.plan.peers) {
            if (!p.send_src.empty())
                NCCLC(ncclSend(S.sendbuf + 2 * so, 2 * p.send_src.size(), ncclDouble, p.rank, S.comm, S.stream));
            if (!p.recv_dst.empty())
                NCCLC(ncclRecv(S.recvbuf + 2 * ro, 2 * p.recv_dst.size(), ncclDouble, p.rank, S.comm, S.stream));
            so += p.send_src.size();
            ro += p.recv_dst.size();
        }
        NCCLC(ncclGroupEnd());
        evp_launch_halo_unpack(a, bb, S.h_recv_dst, (const signed char *)S.h_recv_sign, S.recvbuf,
                               S.n_recv, S.stream);
    }
    return 0;
}

// velocity halo of buffer `b` (ice_dyn_evp.F90:908-910)
int halo_uv(int b)
{
    const bool pushed = S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH);
    if (!pushed)
        evp_launch_halo_local(S.u[b], S.v[b], S.h_local_dst, S.h_local_src,
                              (const signed char *)S.h_local_sign, S.n_local, S.stream);
    // tripole seam of the top physical row (all on this rank, enforced by the plan); the remote
    // exchange below never involves seam-row cells
    evp_launch_halo_seam(S.u[b], S.v[b], S.h_seam_a, S.h_seam_b, S.n_seam, S.h_seam_pole, S.n_pole,
                         S.h_late_dst, S.h_late_src, (const signed char *)S.h_late_sign, S.n_late, S.stream);
    if (int rc = halo_remote_pair(S.u[b], S.v[b])) return rc;
    return 0;
}

// ---- on-chip resident subcycle -------------------------------------------------------
// Host side of evp_resident.hip: which tiles exchange velocities (producers == readers by
// symmetry: the 8 surrounding tiles, with cyclic wrap through the ghost-cell images).
bool resident_possible(bool with_peers = false)
{
    if (S.d.nblocks != 1 || (!with_peers && !S.plan.peers.empty())) return false;
    if ((S.n_seam + S.n_pole + S.n_late) > 0) return false;          // tripole seam: streaming path
    if (S.n_local > 0 && !S.push_ok) return false;
    return true;
}

int resident_setup(int logw)
{
    if (S.res_nbr && S.res_logw == logw) return 0;
    if (S.res_nbr) { (void)hipFree(S.res_nbr); S.res_nbr = nullptr; }
    if (S.res_flags) { (void)hipFree(S.res_flags); S.res_flags = nullptr; }
    S.res_logw = logw;
    const int W = 1 << logw, H = 256 / W;
    int gx, gy;
    evp_resident_geometry(S.max_ni, S.max_nj, logw, &gx, &gy);
    const int ntiles = gx * gy;
    const int nx = S.d.nx_block, ny = S.d.ny_block;
    const int ilo = S.ilo[0], ihi = S.ihi[0], jlo = S.jlo[0], jhi = S.jhi[0];
    // producer of every cell of the (single) block: tile id, or -1 (never written)
    std::vector<int> prod((size_t)nx * ny, -1);
    for (int j = jlo; j <= jhi; ++j)
        for (int i = ilo; i <= ihi; ++i)
            prod[(size_t)(j - 1) * nx + (i - 1)] = ((j - jlo) / (H - 1)) * gx + (i - ilo) / (W - 1);
    for (size_t k = 0; k < S.plan.local_dst.size(); ++k)
        if (S.plan.local_src[k] >= 0) prod[S.plan.local_dst[k]] = prod[S.plan.local_src[k]];
    std::vector<int> nbr((size_t)ntiles * EVP_RES_NNB, -1);
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx) {
            const int t = by * gx + bx;
            int cnt = 0;
            // velocities read by the T-cells of this tile: i0-1..i0+W-1, j0-1..j0+H-1
            const int i0 = ilo + bx * (W - 1), j0 = jlo + by * (H - 1);
            for (int j = j0 - 1; j <= j0 + H - 1; ++j)
                for (int i = i0 - 1; i <= i0 + W - 1; ++i) {
                    if (i < 1 || i > nx || j < 1 || j > ny) continue;
                    const int p = prod[(size_t)(j - 1) * nx + (i - 1)];
                    if (p < 0 || p == t) continue;
                    bool seen = false;
                    for (int e = 0; e < cnt; ++e) seen |= nbr[(size_t)t * EVP_RES_NNB + e] == p;
                    if (seen) continue;
                    if (cnt >= EVP_RES_NNB) return fail(-6, "resident: too many neighbour tiles");
                    nbr[(size_t)t * EVP_RES_NNB + cnt++] = p;
                }
        }
    // symmetry (a reader must also be waited for before its input is overwritten)
    for (int t = 0; t < ntiles; ++t)
        for (int e = 0; e < EVP_RES_NNB; ++e) {
            const int p = nbr[(size_t)t * EVP_RES_NNB + e];
            if (p < 0) continue;
            bool back = false;
            int cntp = 0;
            for (int f = 0; f < EVP_RES_NNB; ++f) {
                back |= nbr[(size_t)p * EVP_RES_NNB + f] == t;
                cntp += nbr[(size_t)p * EVP_RES_NNB + f] >= 0;
            }
            if (!back) {
                if (cntp >= EVP_RES_NNB) return fail(-6, "resident: too many neighbour tiles");
                nbr[(size_t)p * EVP_RES_NNB + cntp] = t;
            }
        }
    S.res_ntiles = ntiles;
    HIPC(hipMalloc((void **)&S.res_nbr, nbr.size() * sizeof(int)));
    HIPC(hipMemcpy(S.res_nbr, nbr.data(), nbr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPC(hipMalloc((void **)&S.res_flags, (size_t)ntiles * sizeof(int)));
    if (!S.res_err) {
        HIPC(hipMalloc((void **)&S.res_err, sizeof(int)));
        HIPC(hipMemset(S.res_err, 0, sizeof(int)));
    }
    return 0;
}

// ---- second generation (evp_resident2.hip): ring lists and publish map of a tile shape ------
// For every tile: the cells of its LDS velocity tile that it reads but does not produce itself
// (ring + ghost/truncation cells), each with the record to poll and the U-cell that produces
// it; and the map of U-cells some other tile mirrors (those publish a record each subcycle).
// Geometry only -- independent of the ice masks.
int resident2_setup(int logw)
{
    if (S.res2_ring && S.res2_logw == logw) return 0;
    auto F = [](auto *&p) { if (p) (void)hipFree((void *)p); p = nullptr; };
    F(S.res2_ring); F(S.res2_cnt); F(S.res2_pub);
    S.res2_logw = logw;
    const int W = 1 << logw, H = 256 / W, LW = W + 1;
    int gx, gy;
    evp_resident_geometry(S.max_ni, S.max_nj, logw, &gx, &gy);
    const int ntiles = gx * gy;
    const int nx = S.d.nx_block, ny = S.d.ny_block;
    const int ilo = S.ilo[0], ihi = S.ihi[0], jlo = S.jlo[0], jhi = S.jhi[0];
    std::vector<int> ghost_src((size_t)nx * ny, -1);
    for (size_t k = 0; k < S.plan.local_dst.size(); ++k)
        if (S.plan.local_src[k] >= 0) ghost_src[S.plan.local_dst[k]] = S.plan.local_src[k];
    for (const HaloPeer &p : S.plan.peers)            // produced on another rank: -2 (always refreshed)
        for (int32_t d : p.recv_dst) ghost_src[d] = -2;
    std::vector<int4> ring((size_t)ntiles * EVP_RES2_RING, make_int4(-1, 0, -1, 0));
    std::vector<int> cnt((size_t)ntiles, 0);
    std::vector<uint8_t> pub((size_t)nx * ny, 0);
    std::vector<char> seen((size_t)(H + 1) * LW);
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx) {
            const int t = by * gx + bx;
            const int i0 = ilo + bx * (W - 1), j0 = jlo + by * (H - 1);
            std::fill(seen.begin(), seen.end(), 0);
            for (int trow = 0; trow < H; ++trow)
                for (int tcol = 0; tcol < W; ++tcol) {
                    const int i = i0 + tcol, j = j0 + trow;
                    if (i > ihi + 1 || j > jhi + 1) continue;          // T-cell not computed
                    for (int q = 0; q < 4; ++q) {
                        const int di = -(q & 1), dj = -(q >> 1);
                        const int pc = tcol + di, pr = trow + dj, pi = i + di, pj = j + dj;
                        const bool interior = pi >= ilo && pi <= ihi && pj >= jlo && pj <= jhi;
                        const bool here = interior && pc >= 0 && pc <= W - 2 && pr >= 0 && pr <= H - 2;
                        if (here) continue;
                        const int li = (pr + 1) * LW + (pc + 1);
                        if (seen[li]) continue;
                        seen[li] = 1;
                        if (pi < 1 || pi > nx || pj < 1 || pj > ny) continue;
                        const int cp = (pj - 1) * nx + (pi - 1);
                        const int src = interior ? cp : ghost_src[cp];
                        if (cnt[t] >= EVP_RES2_RING) return fail(-6, "resident2: ring list overflow");
                        ring[(size_t)t * EVP_RES2_RING + cnt[t]++] = make_int4(cp, li, src, 0);
                        if (interior) pub[cp] = 1;
                    }
                }
        }
    S.res2_ntiles = ntiles;
    HIPC(hipMalloc((void **)&S.res2_ring, ring.size() * sizeof(int4)));
    HIPC(hipMemcpy(S.res2_ring, ring.data(), ring.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPC(hipMalloc((void **)&S.res2_cnt, cnt.size() * sizeof(int)));
    HIPC(hipMemcpy(S.res2_cnt, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPC(hipMalloc((void **)&S.res2_pub, pub.size()));
    HIPC(hipMemcpy(S.res2_pub, pub.data(), pub.size(), hipMemcpyHostToDevice));
    for (auto &p : S.res2_rec)
        if (!p) {
            if (!S.res2_rec_owned) return fail(-6, "resident2: record buffers missing from the mailbox");
            HIPC(hipMalloc(&p, (size_t)nx * ny * 32));
            HIPC(hipMemset(p, 0, (size_t)nx * ny * 32));
        }
    if (!S.res_err) {
        HIPC(hipMalloc((void **)&S.res_err, sizeof(int)));
        HIPC(hipMemset(S.res_err, 0, sizeof(int)));
    }
    return 0;
}

bool resident2_fits(bool remote = false)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, S.device) != hipSuccess) return false;
    // remote: decided before any field has been seen -> the flag combination with the largest LDS need
    const unsigned fl = remote ? (S.flags & S.flags_allowed & ~(EVP_F_WATER_IS_OCN | EVP_F_TBU_ZERO)) : (S.flags & S.flags_allowed);
    const int per_cu = std::min(evp_resident2_max_blocks_per_cu(S.prm.strict != 0, cap_mode(), fl, S.res2_logw, remote), 8);
    const long cap = (long)per_cu * prop.multiProcessorCount;
    return S.res2_ntiles > 0 && (long)S.res2_ntiles * 10 <= cap * 9;
}

int resident_tables();

int launch_resident2(int ndte, int cur0, bool dry)
{
    if (ndte >= 4096) return fail(-6, "resident2: ndte must be < 4096");
    if (int rc = resident_tables()) return rc;
    EvpArgs A;
    fill_args(A, cur0, 1);
    EvpResident2 R;
    R.ndte = ndte;
    R.cur0 = dry ? 0 : cur0;
    R.dry = dry ? 1 : 0;
    S.res2_epoch = (S.res2_epoch + 1u) & 0xFFFFFu;
    if (S.res2_epoch == 0) S.res2_epoch = 1;
    R.tag_base = S.res2_epoch << 12;
    R.par0 = S.res2_par;
    S.res2_par = (S.res2_par + ndte + 1) & 1;     // never start in the buffer the previous launch ended in
    R.rimg = S.res_remote ? S.res2_rimg : nullptr;
    R.rimg_ni = S.max_ni; R.rimg_nj = S.max_nj;
    R.peer_rec = S.res2_peer_rec;
    R.peer_rstride = S.res2_peer_rstride;
    static const double tmo_ms = env("CICE_EVP_HIP_HALO_TIMEOUT_MS") ? std::atof(env("CICE_EVP_HIP_HALO_TIMEOUT_MS")) : 30000.0;
    R.timeout_ticks = (unsigned long long)((S.res_timeout_ms > 0 ? S.res_timeout_ms : tmo_ms) * 1.0e5);
    R.spin_limit = 4000000u;
    R.err = S.res_err;
    R.pubmap = S.res2_pub;
    R.ring = S.res2_ring;
    R.ring_cnt = S.res2_cnt;
    R.rec[0] = S.res2_rec[0];
    R.rec[1] = S.res2_rec[1];
    if (dry) {   // inputs come from the current state, nothing is written back
        R.u[0] = S.u[cur0]; R.v[0] = S.v[cur0]; R.u[1] = S.u[cur0]; R.v[1] = S.v[cur0];
    } else {
        R.u[0] = S.u[0]; R.v[0] = S.v[0]; R.u[1] = S.u[1]; R.v[1] = S.v[1];
    }
    R.tab = S.res_tab + (dry ? 28 * (1 + cur0) : 0);
    evp_launch_resident2(A, R, S.max_ni, S.max_nj, S.res2_logw, S.prm.strict != 0, cap_mode(), S.stream);
    HIPC(hipGetLastError());
    return 0;
}

// every workgroup must be resident at once: occupancy query x CUs, with a margin
bool resident_fits()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, S.device) != hipSuccess) return false;
    const int per_cu = std::min(evp_resident_max_blocks_per_cu(S.prm.strict != 0, cap_mode(), S.flags & S.flags_allowed, S.res_logw), 8);
    const long cap = (long)per_cu * prop.multiProcessorCount;
    return S.res_ntiles > 0 && (long)S.res_ntiles * 10 <= cap * 9;
}

int resident_tables()
{
    if (!S.res_tab) {
        // three pointer tables, uploaded once: [0] real run, [1]/[2] dry probe reading sig[0]/sig[1]
        double *tab[3][28];
        for (int v = 0; v < 3; ++v) {
            for (int k = 0; k < 12; ++k) {
                tab[v][k] = S.sig[v == 0 ? 0 : v - 1][k];
                tab[v][12 + k] = S.sig[v == 0 ? 1 : v - 1][k];
            }
            tab[v][24] = S.in[F_STRINTX]; tab[v][25] = S.in[F_STRINTY];
            tab[v][26] = S.in[F_TAUBX]; tab[v][27] = S.in[F_TAUBY];
        }
        HIPC(hipMalloc((void **)&S.res_tab, sizeof tab));
        HIPC(hipMemcpy(S.res_tab, tab, sizeof tab, hipMemcpyHostToDevice));
    }
    return 0;
}

int launch_resident(int ndte, int cur0, bool dry)
{
    EvpArgs A;
    fill_args(A, cur0, 1);
    EvpResident R;
    R.ndte = ndte;
    R.cur0 = dry ? 0 : cur0;
    R.dry = dry ? 1 : 0;
    R.spin_limit = 4000000u;
    R.xcdmap = env("CICE_EVP_HIP_RES_XCD") ? std::atoi(env("CICE_EVP_HIP_RES_XCD")) : 0;
    R.dbg = env("CICE_EVP_HIP_RES_DEBUG") ? std::atoi(env("CICE_EVP_HIP_RES_DEBUG")) : 0;
    R.flags = S.res_flags;
    R.nbr = S.res_nbr;
    R.err = S.res_err;
    if (dry) {
        R.u[0] = S.res_scratch[0]; R.v[0] = S.res_scratch[1];
        R.u[1] = S.res_scratch[2]; R.v[1] = S.res_scratch[3];
    } else {
        R.u[0] = S.u[0]; R.v[0] = S.v[0]; R.u[1] = S.u[1]; R.v[1] = S.v[1];
    }
    if (int rc = resident_tables()) return rc;
    R.tab = S.res_tab + (dry ? 28 * (1 + cur0) : 0);
    HIPC(hipMemsetAsync(S.res_flags, 0, (size_t)S.res_ntiles * sizeof(int), S.stream));
    evp_launch_resident(A, R, S.max_ni, S.max_nj, S.res_logw, S.prm.strict != 0, cap_mode(), S.stream);
    HIPC(hipGetLastError());
    return 0;
}

int resident_check_error()
{
    if (!S.res_launched) return 0;
    S.res_launched = false;
    int e = 0;
    HIPC(hipMemcpy(&e, S.res_err, sizeof(int), hipMemcpyDeviceToHost));
    if (e) {
        HIPC(hipMemset(S.res_err, 0, sizeof(int)));
        S.res_mode = 0;
        if (e == 2)
            return fail(-7, "resident EVP kernel: a record of another rank never arrived within the time-out "
                            "(CICE_EVP_HIP_HALO_TIMEOUT_MS)");
        return fail(-7, "resident EVP kernel: a neighbour-flag wait timed out (workgroups not co-resident?)");
    }
    return 0;
}


// ---- mailbox halo: set-up over HIP IPC (kernel: evp_halo_direct.hip) ------------------------
// What a rank tells the others: how to map its mailbox and where each peer's entries land.
struct HaloBlob {
    uint32_t magic, version;
    int32_t rank, npeers;
    uint64_t host_id;
    int64_t pid;
    uint64_t base;                 // mailbox address in the exporting process
    uint64_t inbox_off, n_recv;
    uint64_t rec_off, rec_stride;  // record buffers of the resident kernel inside the mailbox (0: none)
    int32_t can_res, pad_;         // this rank can run the resident kernel with remote neighbours
    hipIpcMemHandle_t handle;
    struct { int32_t rank, recv_off, count, flag_idx; } peer[EVP_DIRECT_MAXPEER];
};
static_assert(sizeof(HaloBlob) <= CICE_EVP_HIP_HALO_BLOB, "HaloBlob must fit CICE_EVP_HIP_HALO_BLOB");
constexpr uint32_t HALO_BLOB_MAGIC = 0x45565048u;   // "EVPH"

uint64_t host_identity()
{
    char name[256] = {0};
    (void)gethostname(name, sizeof name - 1);
    uint64_t h = 1469598103934665603ull;
    for (const char *c = name; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
    return h;
}

int direct_export(HaloBlob &B)
{
    State::Direct &X = S.direct;
    const int np = (int)S.plan.peers.size();
    if (np > EVP_DIRECT_MAXPEER) return fail(-8, "mailbox halo: %d peers > %d", np, EVP_DIRECT_MAXPEER);
    // resident kernel across GPUs: its record buffers must be writable by the neighbours, so they
    // live in the mailbox allocation (one IPC handle)
    bool want_res = resident_possible(true) && !S.plan.peers.empty() &&
                    !(env("CICE_EVP_HIP_RESIDENT") && std::atoi(env("CICE_EVP_HIP_RESIDENT")) == 0) &&
                    !(env("CICE_EVP_HIP_RES_REMOTE") && std::atoi(env("CICE_EVP_HIP_RES_REMOTE")) == 0);
    size_t rec_off = 0;
    const size_t rec_stride = S.plane * 32;
    if (!X.mailbox) {
        X.inbox_off = DIRECT_INBOX_OFF;
        X.bytes = X.inbox_off + 2 * 2 * (size_t)std::max(S.n_recv, 1) * sizeof(double);
        X.bytes = (X.bytes + 255) & ~(size_t)255;
        if (want_res) { rec_off = X.bytes; X.bytes += 2 * rec_stride; }
        X.rec_off = rec_off;
        // fine-grained: stores of another GPU become visible to loads here without a kernel boundary
        // (no coarse-grained fallback: without this property a peer's stores are only guaranteed
        // to be seen at kernel boundaries, and the transport would be wrong on a real node)
        HIPC(hipExtMallocWithFlags(&X.mailbox, X.bytes, hipDeviceMallocFinegrained));
        HIPC(hipMemset(X.mailbox, 0, X.bytes));
    }
    int can_res = 0;
    if (want_res && X.rec_off) {
        if (S.res2_rec_owned)
            for (auto &q : S.res2_rec) { if (q) (void)hipFree(q); q = nullptr; }
        S.res2_rec_owned = false;
        S.res2_rec[0] = (char *)X.mailbox + X.rec_off;
        S.res2_rec[1] = (char *)X.mailbox + X.rec_off + rec_stride;
        const int forced_w = env("CICE_EVP_HIP_RES_LOGW") ? std::atoi(env("CICE_EVP_HIP_RES_LOGW")) : 0;
        for (int logw : {4, 5, 6}) {
            if (forced_w && logw != forced_w) continue;
            if (resident2_setup(logw)) continue;
            if (resident2_fits(true)) { can_res = 1; break; }
        }
        g_err.clear();
    }
    std::memset(&B, 0, sizeof B);
    B.can_res = can_res;
    B.rec_off = X.rec_off;
    B.rec_stride = rec_stride;
    B.magic = HALO_BLOB_MAGIC;
    B.version = 1;
    B.rank = S.d.rank;
    B.npeers = np;
    B.host_id = host_identity();
    B.pid = (int64_t)getpid();
    B.base = (uint64_t)(uintptr_t)X.mailbox;
    B.inbox_off = X.inbox_off;
    B.n_recv = (uint64_t)S.n_recv;
    HIPC(hipIpcGetMemHandle(&B.handle, X.mailbox));
    int ro = 0;
    for (int q = 0; q < np; ++q) {
        const HaloPeer &p = S.plan.peers[q];
        B.peer[q].rank = p.rank;
        B.peer[q].recv_off = ro;
        B.peer[q].count = (int)p.recv_dst.size();
        B.peer[q].flag_idx = q;
        ro += (int)p.recv_dst.size();
    }
    X.exported = true;
    return 0;
}

// Map every peer's mailbox and build the device tables.  Local decision only (no communication).
int direct_import(const HaloBlob *blobs, int nranks)
{
    State::Direct &X = S.direct;
    if (!X.exported) return fail(-8, "mailbox halo: import before export");
    if (nranks != S.d.nranks) return fail(-8, "mailbox halo: %d blobs for %d ranks", nranks, S.d.nranks);
    const int np = (int)S.plan.peers.size();
    std::vector<double *> send_addr((size_t)std::max(S.n_send, 1), nullptr);
    std::vector<unsigned> send_pstride((size_t)std::max(S.n_send, 1), 0u);
    std::vector<unsigned *> peer_flag((size_t)std::max(np, 1));
    std::map<int, char *> mapped;
    size_t so = 0;
    for (int q = 0; q < np; ++q) {
        const HaloPeer &p = S.plan.peers[q];
        if (p.rank < 0 || p.rank >= nranks) return fail(-8, "mailbox halo: peer rank %d out of range", p.rank);
        const HaloBlob &B = blobs[p.rank];
        if (B.magic != HALO_BLOB_MAGIC || B.version != 1 || B.rank != p.rank)
            return fail(-8, "mailbox halo: bad blob of rank %d", p.rank);
        if (B.host_id != host_identity()) return fail(-8, "mailbox halo: rank %d is on another host", p.rank);
        int e = -1;
        for (int k = 0; k < B.npeers; ++k)
            if (B.peer[k].rank == S.d.rank) e = k;
        if (e < 0 || B.peer[e].count != (int)p.send_src.size())
            return fail(-8, "mailbox halo: rank %d expects %d cells from this rank, plan sends %d", p.rank,
                        e < 0 ? -1 : B.peer[e].count, (int)p.send_src.size());
        char *base = nullptr;
        if (B.pid == (int64_t)getpid()) base = (char *)(uintptr_t)B.base;       // same process (self-exchange)
        else if (mapped.count(p.rank)) base = mapped[p.rank];
        else {
            void *ptr = nullptr;
            HIPC(hipIpcOpenMemHandle(&ptr, B.handle, hipIpcMemLazyEnablePeerAccess));
            X.opened.push_back(ptr);
            base = (char *)ptr;
        }
        mapped[p.rank] = base;
        peer_flag[q] = (unsigned *)base + (size_t)B.peer[e].flag_idx * EVP_DIRECT_FLAG_STRIDE;
        for (size_t k = 0; k < p.send_src.size(); ++k) {
            send_addr[so + k] = (double *)(base + B.inbox_off) + 2 * ((size_t)B.peer[e].recv_off + k);
            send_pstride[so + k] = (unsigned)(2 * B.n_recv);
        }
        so += p.send_src.size();
    }
    auto up = [&](auto *&dptr, const auto &v) -> int {
        using T = typename std::remove_reference<decltype(v[0])>::type;
        if (!dptr) HIPC(hipMalloc((void **)&dptr, v.size() * sizeof(T)));
        HIPC(hipMemcpy((void *)dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    };
    if (up(X.send_addr, send_addr) || up(X.send_pstride, send_pstride) || up(X.peer_flag, peer_flag)) return -1;
    // resident kernel with neighbours on other GPUs: only if EVERY rank can run it
    bool all_res = true;
    for (int r = 0; r < nranks; ++r) all_res = all_res && blobs[r].magic == HALO_BLOB_MAGIC && blobs[r].can_res != 0;
    S.res_remote = false;
    if (all_res && np > 0) {
        std::vector<void *> prec((size_t)np);
        std::vector<size_t> pstr((size_t)np);
        const int nslot = 2 * (S.max_nj + S.max_ni);
        std::vector<int2> rimg((size_t)nslot * 2, make_int2(-1, -1));
        const int nx = S.d.nx_block;
        bool ok = true;
        for (int q = 0; q < np && ok; ++q) {
            const HaloPeer &p = S.plan.peers[q];
            const HaloBlob &B = blobs[p.rank];
            prec[q] = mapped[p.rank] + B.rec_off;
            pstr[q] = (size_t)B.rec_stride;
            if (env("CICE_EVP_HIP_RES_REMOTE_BREAK")) {      // test hook: records go nowhere -> the probe must fail
                void *dummy = nullptr;
                HIPC(hipMalloc(&dummy, 2 * (size_t)B.rec_stride));
                prec[q] = dummy;                             // (leaked on purpose: test processes only)
            }
            for (size_t k = 0; k < p.send_src.size() && ok; ++k) {
                const int rem = (int)(p.send_src[k] % S.plane);
                const int j = rem / nx + 1, i = rem % nx + 1;
                const int cand[4] = {(i == S.ilo[0]) ? (j - S.jlo[0]) : -1,
                                     (i == S.ihi[0]) ? S.max_nj + (j - S.jlo[0]) : -1,
                                     (j == S.jlo[0]) ? 2 * S.max_nj + (i - S.ilo[0]) : -1,
                                     (j == S.jhi[0]) ? 2 * S.max_nj + S.max_ni + (i - S.ilo[0]) : -1};
                bool placed = false;
                for (int e = 0; e < 4 && !placed; ++e) {
                    if (cand[e] < 0) continue;
                    for (int w = 0; w < 2 && !placed; ++w) {
                        int2 &slot = rimg[(size_t)cand[e] * 2 + w];
                        if (slot.x < 0) { slot = make_int2(q, p.send_dst[k]); placed = true; }
                    }
                }
                ok = placed;
            }
        }
        if (ok) {
            if (up(S.res2_rimg, rimg) || up(S.res2_peer_rec, prec) || up(S.res2_peer_rstride, pstr)) return -1;
            S.res_remote = true;
        }
    }
    EvpDirect D;
    fill_direct(D);
    if (!X.d_dx) HIPC(hipMalloc((void **)&X.d_dx, sizeof(EvpDirect)));
    HIPC(hipMemcpy(X.d_dx, &D, sizeof(EvpDirect), hipMemcpyHostToDevice));
    if (!X.d_cnt) HIPC(hipMalloc((void **)&X.d_cnt, 32 * sizeof(unsigned)));
    HIPC(hipMemset(X.d_cnt, 0, 32 * sizeof(unsigned)));
    return 0;
}

// Probe exchange (collective): every interior cell carries its global cell number, every ghost
// must come back holding the number of the cell it mirrors (halochk.F90:232-247's method).
// Uses the velocity buffers before any state has been uploaded, and leaves them zeroed.
int direct_probe()
{
    std::vector<double> hu(S.n, 0.0), hv(S.n, 0.0);
    const int nx = S.d.nx_block;
    for (int b = 0; b < S.d.nblocks; ++b)
        for (int j = S.jlo[b]; j <= S.jhi[b]; ++j)
            for (int i = S.ilo[b]; i <= S.ihi[b]; ++i) {
                const size_t c = b * S.plane + (size_t)(j - 1) * nx + (i - 1);
                const double gid = (double)((S.iglob0[b] + (i - S.ilo[b]) - 1) +
                                            (size_t)S.d.nx_global * (S.jglob0[b] + (j - S.jlo[b]) - 1));
                hu[c] = gid + 1.0;
                hv[c] = -2.0 * (gid + 1.0);
            }
    HIPC(hipMemcpyAsync(S.u[0], hu.data(), S.n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemcpyAsync(S.v[0], hv.data(), S.n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    EvpDirect D;
    fill_direct(D);
    for (int rep = 0; rep < 3; ++rep)           // both inbox parities, and a repeat
        evp_launch_halo_direct(D, S.u[0], S.v[0], S.stream);
    HIPC(hipMemcpyAsync(hu.data(), S.u[0], S.n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipMemcpyAsync(hv.data(), S.v[0], S.n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    int err = 0;
    HIPC(hipMemcpyAsync(&err, D.err, sizeof(int), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemsetAsync(S.u[0], 0, S.n * sizeof(double), S.stream));
    HIPC(hipMemsetAsync(S.v[0], 0, S.n * sizeof(double), S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    if (err) {
        HIPC(hipMemset(D.err, 0, sizeof(int)));
        return fail(-8, "mailbox halo probe: peer %d never signalled", S.plan.peers[err - 1].rank);
    }
    for (const HaloPeer &p : S.plan.peers)
        for (size_t k = 0; k < p.recv_dst.size(); ++k) {
            const double want = (double)p.recv_sign[k] * ((double)p.recv_gid[k] + 1.0);
            if (hu[p.recv_dst[k]] != want || hv[p.recv_dst[k]] != -2.0 * want)
                return fail(-8, "mailbox halo probe: ghost %d from rank %d holds %.17g, expected %.17g",
                            (int)p.recv_dst[k], p.rank, hu[p.recv_dst[k]], want);
        }
    return 0;
}

int direct_check_error()
{
    if (!S.direct.on) return 0;
    int e = 0;
    HIPC(hipMemcpy(&e, (char *)S.direct.mailbox + DIRECT_ERR_OFF, sizeof(int), hipMemcpyDeviceToHost));
    if (e) return fail(-8, "mailbox halo: rank %d never signalled within the time-out (CICE_EVP_HIP_HALO_TIMEOUT_MS)",
                       S.plan.peers[e - 1].rank);
    return 0;
}

// Probe of the resident kernel with neighbours on other GPUs (collective): no ice anywhere, every
// interior cell holds its global cell number as "velocity"; three subcycles later every ghost that
// mirrors another rank's cell must hold that cell's number -- carried there by tagged records only.
int resident_remote_probe()
{
    std::vector<double> hu(S.n, 0.0), hv(S.n, 0.0);
    const int nx = S.d.nx_block;
    for (int j = S.jlo[0]; j <= S.jhi[0]; ++j)
        for (int i = S.ilo[0]; i <= S.ihi[0]; ++i) {
            const size_t c = (size_t)(j - 1) * nx + (i - 1);
            const double gid = (double)((S.iglob0[0] + (i - S.ilo[0]) - 1) +
                                        (size_t)S.d.nx_global * (S.jglob0[0] + (j - S.jlo[0]) - 1));
            hu[c] = gid + 1.0;
            hv[c] = -2.0 * (gid + 1.0);
        }
    for (int b = 0; b < 2; ++b) {
        HIPC(hipMemcpyAsync(S.u[b], hu.data(), S.n * sizeof(double), hipMemcpyHostToDevice, S.stream));
        HIPC(hipMemcpyAsync(S.v[b], hv.data(), S.n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    }
    HIPC(hipMemsetAsync(S.mask, 0, S.n, S.stream));
    S.res_timeout_ms = 10000.0;
    int rc = launch_resident2(3, 0, false);
    S.res_timeout_ms = 0;
    if (rc) return rc;
    S.res_launched = true;
    std::vector<double> gu(S.n), gv(S.n);
    HIPC(hipMemcpyAsync(gu.data(), S.u[1], S.n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipMemcpyAsync(gv.data(), S.v[0], S.n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    rc = resident_check_error();
    S.res_mode = -1;            // (resident_check_error parks the mode on failure; decided again at upload)
    for (int b = 0; b < 2; ++b) {
        HIPC(hipMemsetAsync(S.u[b], 0, S.n * sizeof(double), S.stream));
        HIPC(hipMemsetAsync(S.v[b], 0, S.n * sizeof(double), S.stream));
    }
    HIPC(hipStreamSynchronize(S.stream));
    if (rc) return rc;
    for (const HaloPeer &p : S.plan.peers)
        for (size_t k = 0; k < p.recv_dst.size(); ++k) {
            const double want = (double)p.recv_sign[k] * ((double)p.recv_gid[k] + 1.0);
            if (gu[p.recv_dst[k]] != want || gv[p.recv_dst[k]] != -2.0 * want)
                return fail(-8, "resident kernel probe: ghost %d from rank %d holds %.17g, expected %.17g",
                            (int)p.recv_dst[k], p.rank, gu[p.recv_dst[k]], want);
        }
    return 0;
}

// 0 auto, 1 RCCL only, 2 mailbox required
int halo_choice()
{
    const char *h = env("CICE_EVP_HIP_HALO");
    if (!h) return 0;
    if (!std::strcmp(h, "rccl")) return 1;
    if (!std::strcmp(h, "direct")) return 2;
    return 0;
}

// Boundary-first + second stream pays when the interior kernel is long enough to hide the
// exchange; on small per-rank domains the extra host calls (events, two launches) cost more
// than they hide (measured: 50 vs 26 us per subcycle on gx1, eager).  CICE_EVP_HIP_OVERLAP=1/0 forces.
bool use_overlap()
{
    const bool seam = (S.n_seam + S.n_pole + S.n_late) > 0;
    if (!S.overlap || S.plan.peers.empty() || seam || !(S.have_comm || S.direct.on)) return false;
    if (env("CICE_EVP_HIP_OVERLAP")) return std::atoi(env("CICE_EVP_HIP_OVERLAP")) != 0;
    size_t cells = 0;
    for (int b = 0; b < S.d.nblocks; ++b)
        cells += (size_t)(S.ihi[b] - S.ilo[b] + 1) * (S.jhi[b] - S.jlo[b] + 1);
    return cells >= 400000;
}

// The mailbox exchange can ride in the subcycle launch (no tripole seam step in between).
bool use_riding_exchange()
{
    if (!S.direct.on || S.plan.peers.empty() || (S.n_seam + S.n_pole + S.n_late) > 0) return false;
    // Pays when the interior tiles outlast the exchange (measured, 4 x 1800x1200 blocks: 571 us
    // riding, 598 two streams, 627 separate kernel); on a domain that is one wave of workgroups
    // there is nothing to overlap with and the separate kernel is quicker (gx1: 20.8 vs 25 us).
    if (env("CICE_EVP_HIP_HALO_RIDE")) return std::atoi(env("CICE_EVP_HIP_HALO_RIDE")) != 0;
    size_t cells = 0;
    for (int b = 0; b < S.d.nblocks; ++b)
        cells += (size_t)(S.ihi[b] - S.ilo[b] + 1) * (S.jhi[b] - S.jlo[b] + 1);
    return cells >= 400000;
}

// Which tiles of `variant` hold U-cells that some other rank mirrors (send list)?
int get_tile_split(int variant, State::TileSplit **out)
{
    auto it = S.splits.find(variant);
    if (it != S.splits.end()) { *out = &it->second; return 0; }
    int tyb, gx, gy;
    evp_tile_geometry(S.max_ni, S.max_nj, variant, &tyb, &gx, &gy);
    const int ntiles = gx * gy * S.d.nblocks;
    std::vector<char> is_b((size_t)ntiles, 0);
    const int nx = S.d.nx_block;
    for (const HaloPeer &p : S.plan.peers)
        for (int32_t src : p.send_src) {
            const int b = (int)(src / S.plane);
            const int rem = (int)(src % S.plane);
            const int j = rem / nx + 1, i = rem % nx + 1;
            const int bx = (i - S.ilo[b]) / 63, by = (j - S.jlo[b]) / (tyb - 1);
            if (bx < 0 || bx >= gx || by < 0 || by >= gy) continue;
            is_b[((size_t)b * gy + by) * gx + bx] = 1;       // row-major tile id (xcdmap 0/2 decoding)
        }
    std::vector<int> lb, li;
    for (int t = 0; t < ntiles; ++t) (is_b[t] ? lb : li).push_back(t);
    State::TileSplit ts;
    ts.nb = (int)lb.size();
    ts.ni = (int)li.size();
    if (ts.nb) {
        HIPC(hipMalloc((void **)&ts.d_boundary, lb.size() * sizeof(int)));
        HIPC(hipMemcpy(ts.d_boundary, lb.data(), lb.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (ts.ni) {
        HIPC(hipMalloc((void **)&ts.d_interior, li.size() * sizeof(int)));
        HIPC(hipMemcpy(ts.d_interior, li.data(), li.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    {   // boundary tiles first, then the rest: order of the launch that carries the exchange workgroup
        // interior tiles in XCD-chunked order: workgroup w runs on XCD w % 8, so give each XCD one
        // contiguous run of the (row-major) interior sequence -- neighbouring tiles share an L2
        std::vector<int> all(lb);
        const size_t n = li.size(), per = (n + 7) / 8, w0 = lb.size() + 1;    // +1: the exchange workgroup
        std::vector<int> chunked;
        for (size_t w = 0; chunked.size() < n; ++w) {
            const size_t x = (w0 + w) & 7, q = x * per + (w >> 3);
            if ((w >> 3) < per && q < n) chunked.push_back(li[q]);
            if (w > 16 * (n + 8)) break;
        }
        if (chunked.size() != n) chunked = li;
        all.insert(all.end(), chunked.begin(), chunked.end());
        HIPC(hipMalloc((void **)&ts.d_all, all.size() * sizeof(int)));
        HIPC(hipMemcpy(ts.d_all, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    *out = &S.splits.emplace(variant, ts).first->second;
    return 0;
}

int enqueue_loop(int ndte, int cur0)
{
    int cur = cur0;
    const bool strict = S.prm.strict != 0;
    const int cap = cap_mode();
    // Boundary strips first, RCCL exchange on a second stream while the interior tiles run
    // (the tripole seam needs every tile of the top row first, so it keeps the serial order).
    const bool overlap = use_overlap();
    if (use_riding_exchange()) {
        // mailbox halo: one launch per subcycle; the tiles other ranks wait for run first, one
        // extra workgroup exchanges their velocities while the interior tiles are computed
        const int variant = S.tyb % 100;
        State::TileSplit *ts = nullptr;
        if (int rc = get_tile_split(variant, &ts)) return rc;
        const bool pushed = S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH);
        for (int k = 0; k < ndte; ++k) {
            EvpArgs A;
            fill_args(A, cur, k == ndte - 1);
            A.tile_list = ts->d_all; A.tile_count = ts->nb + ts->ni;
            A.dx = S.direct.d_dx; A.dx_count = S.direct.d_cnt; A.dx_fseq = S.direct.d_cnt + 16; A.dx_nb = ts->nb;
            evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, variant, strict, cap, S.stream);
            if (!pushed)
                evp_launch_halo_local(S.u[cur ^ 1], S.v[cur ^ 1], S.h_local_dst, S.h_local_src,
                                      (const signed char *)S.h_local_sign, S.n_local, S.stream);
            cur ^= 1;
        }
        HIPC(hipGetLastError());
        return 0;
    }
    if (!overlap) {
        for (int k = 0; k < ndte; ++k) {
            EvpArgs A;
            fill_args(A, cur, k == ndte - 1);
            evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, S.tyb, strict, cap, S.stream);
            if (int rc = halo_uv(cur ^ 1)) return rc;
            cur ^= 1;
        }
        HIPC(hipGetLastError());
        return 0;
    }
    // the split kernels decode tile ids row-major: use the row-major flavour of the variant
    const int variant = S.tyb % 100;
    State::TileSplit *ts = nullptr;
    if (int rc = get_tile_split(variant, &ts)) return rc;
    for (int k = 0; k < ndte; ++k) {
        const int nxt = cur ^ 1;
        EvpArgs A;
        fill_args(A, cur, k == ndte - 1);
        if (k > 0) HIPC(hipStreamWaitEvent(S.stream, S.ev_halo, 0));   // ghosts of u_in complete
        // 1. tiles whose cells other ranks need
        A.tile_list = ts->d_boundary; A.tile_count = ts->nb;
        evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, variant, strict, cap, S.stream);
        if (!S.direct.on) evp_launch_halo_pack(S.u[nxt], S.v[nxt], S.h_send_src, S.sendbuf, S.n_send, S.stream);
        HIPC(hipEventRecord(S.ev_pack, S.stream));
        // 2. everything else, concurrently with the exchange
        A.tile_list = ts->d_interior; A.tile_count = ts->ni;
        evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, variant, strict, cap, S.stream);
        if (!(S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH)))
            evp_launch_halo_local(S.u[nxt], S.v[nxt], S.h_local_dst, S.h_local_src,
                                  (const signed char *)S.h_local_sign, S.n_local, S.stream);
        // 3. RCCL point-to-point over xGMI on the communication stream
        HIPC(hipStreamWaitEvent(S.stream_comm, S.ev_pack, 0));
        if (S.direct.on) {      // mailbox exchange: stores into the peers' inboxes, no library call
            EvpDirect D;
            fill_direct(D);
            evp_launch_halo_direct(D, S.u[nxt], S.v[nxt], S.stream_comm);
        } else {
            size_t so = 0, ro = 0;
            NCCLC(ncclGroupStart());
            for (const HaloPeer &p : S.plan.peers) {
                if (!p.send_src.empty())
                    NCCLC(ncclSend(S.sendbuf + 2 * so, 2 * p.send_src.size(), ncclDouble, p.rank, S.comm, S.stream_comm));
                if (!p.recv_dst.empty())
                    NCCLC(ncclRecv(S.recvbuf + 2 * ro, 2 * p.recv_dst.size(), ncclDouble, p.rank, S.comm, S.stream_comm));
                so += p.send_src.size();
                ro += p.recv_dst.size();
            }
            NCCLC(ncclGroupEnd());
            evp_launch_halo_unpack(S.u[nxt], S.v[nxt], S.h_recv_dst, (const signed char *)S.h_recv_sign, S.recvbuf,
                                   S.n_recv, S.stream_comm);
        }
        HIPC(hipEventRecord(S.ev_halo, S.stream_comm));
        cur = nxt;
    }
    HIPC(hipStreamWaitEvent(S.stream, S.ev_halo, 0));   // the compute stream owns the final state
    HIPC(hipGetLastError());
    return 0;
}

}  // namespace

// =====================================================================
extern "C" {

int cice_evp_hip_abi_version(void) { return CICE_EVP_HIP_ABI_VERSION; }

int cice_evp_hip_last_error(char *buf, int32_t buflen)
{
    if (buf && buflen > 0) {
        std::strncpy(buf, g_err.c_str(), (size_t)buflen - 1);
        buf[buflen - 1] = 0;
    }
    return (int)g_err.size();
}

int cice_evp_hip_finalize(void)
{
    if (S.stream) (void)hipStreamSynchronize(S.stream);
    free_all();
    S = State();
    return 0;
}

int cice_evp_hip_init(const cice_evp_hip_dims *dims, const cice_evp_hip_params *params,
                      const double *HTE, const double *HTN, const double *dxT, const double *dyT,
                      const double *uarear, const double *tarea)
{
    if (!dims || !params || !HTE || !HTN || !dxT || !dyT || !uarear || !tarea)
        return fail(-1, "cice_evp_hip_init: null argument");
    if (S.ready) cice_evp_hip_finalize();
    if (dims->nghost != 1) return fail(-1, "nghost must be 1");
    if (dims->nblocks < 1 || dims->nblocks > dims->max_blocks) return fail(-1, "bad nblocks/max_blocks");

    S.d = *dims;
    S.prm = *params;
    const int nb = dims->nblocks;
    S.ilo.assign(dims->ilo, dims->ilo + nb);
    S.ihi.assign(dims->ihi, dims->ihi + nb);
    S.jlo.assign(dims->jlo, dims->jlo + nb);
    S.jhi.assign(dims->jhi, dims->jhi + nb);
    S.iglob0.assign(dims->iglob0, dims->iglob0 + nb);
    S.jglob0.assign(dims->jglob0, dims->jglob0 + nb);
    S.d.ilo = S.ilo.data(); S.d.ihi = S.ihi.data(); S.d.jlo = S.jlo.data(); S.d.jhi = S.jhi.data();
    S.d.iglob0 = S.iglob0.data(); S.d.jglob0 = S.jglob0.data();

    if (!build_halo_plan(*dims, S.plan)) return fail(-3, "halo plan: %s", S.plan.error.c_str());
    if (env("CICE_EVP_HIP_SELF_EXCHANGE") && std::atoi(env("CICE_EVP_HIP_SELF_EXCHANGE")) && dims->nranks == 1) {
        // test hook: route the on-device ghost copies through pack -> ncclSend/ncclRecv (to
        // self) -> unpack, so that the remote-halo code path runs on a single GPU
        HaloPeer self;
        self.rank = dims->rank;
        std::vector<int32_t> kd, ks;
        std::vector<int8_t> kg;
        for (size_t k = 0; k < S.plan.local_dst.size(); ++k) {
            if (S.plan.local_src[k] < 0) {
                kd.push_back(S.plan.local_dst[k]); ks.push_back(-1); kg.push_back(1);
                continue;
            }
            self.send_src.push_back(S.plan.local_src[k]);
            self.send_dst.push_back(S.plan.local_dst[k]);
            self.recv_dst.push_back(S.plan.local_dst[k]);
            self.recv_sign.push_back(S.plan.local_sign[k]);
            {
                const int32_t so = S.plan.local_src[k];
                const size_t pl = (size_t)dims->nx_block * dims->ny_block;
                const int b = (int)(so / pl), rem = (int)(so % pl);
                const int j = rem / dims->nx_block + 1, i = rem % dims->nx_block + 1;
                const int ig = dims->iglob0[b] + (i - dims->ilo[b]), jg = dims->jglob0[b] + (j - dims->jlo[b]);
                self.recv_gid.push_back((int32_t)((ig - 1) + (size_t)dims->nx_global * (jg - 1)));
            }
        }
        S.plan.local_dst = kd; S.plan.local_src = ks; S.plan.local_sign = kg;
        S.plan.peers.push_back(self);
    }
    // the global block table is only needed while planning
    S.d.gi0 = S.d.gj0 = S.d.gnx = S.d.gny = S.d.gowner = S.d.glocal = nullptr;

    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(-4, "no HIP device");
    int dev = 0;
    if (env("CICE_EVP_HIP_DEVICE")) dev = std::atoi(env("CICE_EVP_HIP_DEVICE"));
    else if (env("LOCAL_RANK")) dev = std::atoi(env("LOCAL_RANK")) % ndev;
    else dev = dims->rank % ndev;
    S.device = dev;
    HIPC(hipSetDevice(dev));
    HIPC(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    HIPC(hipStreamCreateWithFlags(&S.stream_comm, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&S.ev_pack, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&S.ev_halo, hipEventDisableTiming));
    S.overlap = !(env("CICE_EVP_HIP_NO_OVERLAP") && std::atoi(env("CICE_EVP_HIP_NO_OVERLAP")));
    HIPC(hipEventCreate(&S.ev0));
    HIPC(hipEventCreate(&S.ev1));
    HIPC(hipEventCreate(&S.ev2));
    HIPC(hipEventCreate(&S.ev3));
    HIPC(hipEventCreate(&S.evm[0]));
    HIPC(hipEventCreate(&S.evm[1]));

    S.plane = (size_t)dims->nx_block * dims->ny_block;
    S.n = S.plane * nb;
    S.max_ni = S.max_nj = 0;
    std::vector<int4> hb(nb);
    for (int b = 0; b < nb; ++b) {
        hb[b] = make_int4(S.ilo[b], S.ihi[b], S.jlo[b], S.jhi[b]);
        S.max_ni = std::max(S.max_ni, S.ihi[b] - S.ilo[b] + 1);
        S.max_nj = std::max(S.max_nj, S.jhi[b] - S.jlo[b] + 1);
    }
    S.tyb = 4;
    if (env("CICE_EVP_HIP_TYB")) {
        const int t = std::atoi(env("CICE_EVP_HIP_TYB"));   // tile height [+100: XCD-contiguous order]
        S.tyb = (t % 100 >= 2 && t % 100 <= 9) ? t : 5 + 100 * (t / 100);
    }
    S.use_graph = !(env("CICE_EVP_HIP_NOGRAPH") && std::atoi(env("CICE_EVP_HIP_NOGRAPH")));

    for (auto &p : S.stat)
        if (alloc_d(&p, S.n)) return -1;
    for (int f = F_STRENGTH; f < F_COUNT; ++f) {
        if (f == F_UVEL || f == F_VVEL) continue;
        if (alloc_d(&S.in[f], S.n)) return -1;
    }
    for (int k = 0; k < 2; ++k) {
        if (alloc_d(&S.u[k], S.n) || alloc_d(&S.v[k], S.n)) return -1;
        for (auto &p : S.sig[k])
            if (alloc_d(&p, S.n)) return -1;
    }
    if (alloc_d(&S.hte, S.n) || alloc_d(&S.htn, S.n) || alloc_d(&S.vrelfac, S.n)) return -1;
    S.flags = EVP_F_VRELFAC;
    S.flags_allowed = ~0u;
    if (env("CICE_EVP_HIP_FLAGS")) S.flags_allowed = (unsigned)std::strtoul(env("CICE_EVP_HIP_FLAGS"), nullptr, 0);
    HIPC(hipMalloc((void **)&S.mask, S.n));
    HIPC(hipMemsetAsync(S.mask, 0, S.n, S.stream));
    HIPC(hipMalloc((void **)&S.blk, nb * sizeof(int4)));
    HIPC(hipMemcpy(S.blk, hb.data(), nb * sizeof(int4), hipMemcpyHostToDevice));
    if (upload_lists()) return -1;
    if (build_push_table()) return -1;
    if (S.push_ok) S.flags |= EVP_F_PUSH;
    if (derive_metrics(HTE, HTN, dxT, dyT, uarear, tarea)) return -1;
    S.hmask.resize(S.n);
    S.tyb_forced = env("CICE_EVP_HIP_TYB") != nullptr;
    S.ready = true;
    S.uploaded = false;
    S.cur = 0;
    return 0;
}

int cice_evp_hip_set_metrics(const double *cxp, const double *cyp, const double *cxm,
                             const double *cym, const double *dxhy, const double *dyhx,
                             const double *DminTarea)
{
    if (!S.ready) return fail(-1, "not initialised");
    const double *src[7] = {dxhy, dyhx, cxp, cyp, cxm, cym, DminTarea};   // stat slots 2..8
    for (int k = 0; k < 7; ++k)
        if (src[k] && h2d(S.stat[2 + k], src[k])) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

// Choices made once the first state is on the device (tile shape of the streaming kernel,
// streaming vs on-chip resident kernel); shared by cice_evp_hip_upload and cice_evp_hip_prep.
static int tune_after_upload()
{
    float ms = 0;
    if (!S.tyb_forced && !S.tuned) {
        // pick the tile height once per init by timing a few launches of each variant on
        // the real state (results are identical for every tile shape; only speed differs).
        // The launches write the ping-pong "next" buffers, which the first real subcycle
        // overwrites, so the state is not advanced.
        // tile heights whose wave count fills the 4 SIMDs evenly (4, 8) plus 3; odd wave counts
        // (5, 9) leave one SIMD with twice the work and measured 1.5-2x slower
        const int cand[9] = {4, 8, 3, 104, 108, 103, 204, 208, 203};
        float best = 1e30f;
        int best_t = 5;
        EvpArgs A;
        fill_args(A, S.cur, 0);
        for (int c : cand) {
            for (int rep = 0; rep < 2; ++rep) {   // first pass warms caches / code
                HIPC(hipEventRecord(S.ev2, S.stream));
                for (int k = 0; k < 8; ++k)
                    evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, c, S.prm.strict != 0, cap_mode(), S.stream);
                HIPC(hipEventRecord(S.ev3, S.stream));
                HIPC(hipStreamSynchronize(S.stream));
                HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
            }
            if (ms < best) { best = ms; best_t = c; }
        }
        S.tyb = best_t;
        S.tuned = true;
        S.t_stream_probe_ms = best / 8.0;
        for (auto &kv : S.graphs) (void)hipGraphExecDestroy(kv.second);
        S.graphs.clear();
    }
    // on-chip resident subcycle: use it when it fits and a dry probe on scratch velocities
    // (same work, nothing written back) runs clean and faster than the streaming kernel
    if (S.res_mode < 0) {
        S.res_mode = 0;
        int want = -1;
        if (env("CICE_EVP_HIP_RESIDENT")) want = std::atoi(env("CICE_EVP_HIP_RESIDENT"));
        if (want != 0 && S.res_remote && S.direct.on) {
            // neighbours on other GPUs: tile shape fixed at export, no timing probes (every launch
            // of this kernel is collective across ranks)
            S.res_gen = 2;
            S.res_mode = 1;
        } else if (want != 0 && resident_possible()) {
            const int forced_w = env("CICE_EVP_HIP_RES_LOGW") ? std::atoi(env("CICE_EVP_HIP_RES_LOGW")) : 0;
            const int forced_g = env("CICE_EVP_HIP_RES_GEN") ? std::atoi(env("CICE_EVP_HIP_RES_GEN")) : 0;
            float best = 1e30f;
            int best_w = 0, best_g = 0;
            bool any_fit = false, done = false;
            for (int gen : {2, 1}) {
                if (done || (forced_g && gen != forced_g)) continue;
                for (int logw : {5, 4, 6}) {
                    if (forced_w && logw != forced_w) continue;
                    if (gen == 1) {
                        if (resident_setup(logw)) { if (want == 1) return -6; continue; }
                        if (!resident_fits()) continue;
                        for (auto &p : S.res_scratch)
                            if (!p && alloc_d(&p, S.n)) return -1;
                    } else {
                        if (resident2_setup(logw)) { if (want == 1) return -6; continue; }
                        if (!resident2_fits()) continue;
                    }
                    any_fit = true;
                    if (want == 1 && forced_w && forced_g) { best = 0.0f; best_w = logw; best_g = gen; done = true; break; }
                    // steady-state cost per subcycle = slope between a short and a long dry run
                    // (launch, prologue and epilogue are paid once per evp() call)
                    const int nshort = 8, nlong = 40;
                    float tres = 1e30f, tl[2] = {0, 0};
                    bool ok = true;
                    for (int rep = 0; rep < 3 && ok; ++rep) {
                        const int np = (rep == 2) ? nlong : nshort;      // rep 0 warms up
                        if (gen == 1)
                            for (int q = 0; q < 4; ++q)
                                HIPC(hipMemcpyAsync(S.res_scratch[q], (q & 1) ? S.v[S.cur] : S.u[S.cur],
                                                    S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
                        HIPC(hipEventRecord(S.ev2, S.stream));
                        if (int rc = (gen == 1 ? launch_resident(np, S.cur, true) : launch_resident2(np, S.cur, true))) return rc;
                        HIPC(hipEventRecord(S.ev3, S.stream));
                        HIPC(hipStreamSynchronize(S.stream));
                        S.res_launched = true;
                        if (resident_check_error()) { ok = false; break; }
                        HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
                        if (rep >= 1) tl[rep - 1] = ms;
                    }
                    if (ok) tres = (tl[1] - tl[0]) / (nlong - nshort);
                    if (ok && tres < best) { best = tres; best_w = logw; best_g = gen; }
                }
            }
            S.t_res_probe_ms = best_w ? best : -1.0;
            if (best_w && (want == 1 || S.t_stream_probe_ms <= 0.0 || best < S.t_stream_probe_ms)) {
                if (int rc = (best_g == 1 ? resident_setup(best_w) : resident2_setup(best_w))) return rc;
                S.res_gen = best_g;
                S.res_mode = 1;
            } else if (want == 1) {
                return fail(-6, any_fit ? "resident EVP kernel requested but its probe failed"
                                        : "resident EVP kernel requested but its workgroups cannot be co-resident");
            }
        } else if (want == 1) {
            return fail(-6, "resident EVP kernel requested but not applicable (one block per rank, no remote halo, no tripole)");
        }
    }
    return 0;
}

int cice_evp_hip_upload(const double *const *f, const int32_t *iceTmask, const int32_t *iceUmask)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!f || !iceTmask || !iceUmask) return fail(-1, "null argument");
    HIPC(hipEventRecord(S.ev2, S.stream));
    S.cur = 0;
    for (int k = 0; k < 12; ++k) {
        if (!f[k]) return fail(-1, "null stress field %d", k);
        if (h2d(S.sig[0][k], f[k])) return -1;
        HIPC(hipMemcpyAsync(S.sig[1][k], S.sig[0][k], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    }
    for (int fi = F_STRENGTH; fi < F_COUNT; ++fi) {
        if (fi == F_UVEL || fi == F_VVEL) continue;
        if ((fi == F_UVEL_INIT || fi == F_VVEL_INIT) && !f[fi]) continue;   // only read when revp = 1
        if (!f[fi]) return fail(-1, "null field %d", fi);
        if (h2d(S.in[fi], f[fi])) return -1;
    }
    if (S.prm.revp != 0.0 && (!f[F_UVEL_INIT] || !f[F_VVEL_INIT]))
        return fail(-1, "uvel_init/vvel_init required for revised EVP");
    if (!f[F_UVEL] || !f[F_VVEL]) return fail(-1, "null velocity field");
    if (h2d(S.u[0], f[F_UVEL]) || h2d(S.v[0], f[F_VVEL])) return -1;
    HIPC(hipMemcpyAsync(S.u[1], S.u[0], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    HIPC(hipMemcpyAsync(S.v[1], S.v[0], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    bool water_is_ocn = true, tbu_zero = true;
    {
        const double *wx = f[F_WATERX], *wy = f[F_WATERY], *uo = f[F_UOCN], *vo = f[F_VOCN], *tb = f[F_TBU];
        for (size_t k = 0; k < S.n; ++k) {
            const bool um = iceUmask[k] != 0;
            S.hmask[k] = (uint8_t)((iceTmask[k] != 0 ? 1 : 0) | (um ? 2 : 0));
            if (um) {
                // bit-for-bit identical operands (cosw=1, sinw=0: ice_dyn_shared.F90:69-70,819-820)
                if (std::memcmp(&wx[k], &uo[k], 8) != 0 || std::memcmp(&wy[k], &vo[k], 8) != 0) water_is_ocn = false;
                if (tb[k] != 0.0) tbu_zero = false;
            }
        }
    }
    S.flags &= ~(EVP_F_WATER_IS_OCN | EVP_F_TBU_ZERO);
    if (water_is_ocn) S.flags |= EVP_F_WATER_IS_OCN;
    if (tbu_zero) S.flags |= EVP_F_TBU_ZERO;
    evp_launch_vrelfac(S.in[F_AIX], S.in[F_CW], S.prm.rhow, S.vrelfac, S.n, S.stream);
    HIPC(hipMemcpyAsync(S.mask, S.hmask.data(), S.n, hipMemcpyHostToDevice, S.stream));
    HIPC(hipEventRecord(S.ev3, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
    S.t_h2d_ms = ms;
    S.uploaded = true;
    return tune_after_upload();
}

int cice_evp_hip_subcycle(int32_t ndte)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (ndte < 0) return fail(-1, "ndte < 0");
    if (ndte == 0) return 0;
    HIPC(hipEventRecord(S.ev0, S.stream));
    if (S.res_mode == 1) {
        if (int rc = (S.res_gen == 2 ? launch_resident2(ndte, S.cur, false) : launch_resident(ndte, S.cur, false))) return rc;
        S.res_launched = true;
        HIPC(hipEventRecord(S.ev1, S.stream));
        S.cur ^= (ndte & 1);
        S.t_nsub = ndte;
        return 0;
    }
    // RCCL p2p inside a captured graph: opt-in (CICE_EVP_HIP_GRAPH_RCCL=1) until measured on a multi-GPU node
    static const bool graph_rccl = env("CICE_EVP_HIP_GRAPH_RCCL") && std::atoi(env("CICE_EVP_HIP_GRAPH_RCCL"));
    const bool graph_ok = S.use_graph && (S.plan.peers.empty() || graph_rccl || S.direct.on);
    if (graph_ok) {
        const auto key = std::make_pair((int)ndte, S.cur);
        auto it = S.graphs.find(key);
        if (it == S.graphs.end()) {
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            if (use_overlap() || use_riding_exchange()) {   // device allocations are not allowed while capturing
                State::TileSplit *ts = nullptr;
                if (int rc = get_tile_split(S.tyb % 100, &ts)) return rc;
            }
            HIPC(hipStreamBeginCapture(S.stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_loop(ndte, S.cur);
            hipError_t e = hipStreamEndCapture(S.stream, &g);
            if (rc) return rc;
            if (e != hipSuccess) return fail((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
            HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            HIPC(hipGraphDestroy(g));
            it = S.graphs.emplace(key, ge).first;
            // the event recorded before the capture is stale for timing; re-record
            HIPC(hipEventRecord(S.ev0, S.stream));
        }
        HIPC(hipGraphLaunch(it->second, S.stream));
    } else {
        if (int rc = enqueue_loop(ndte, S.cur)) return rc;
    }
    HIPC(hipEventRecord(S.ev1, S.stream));
    S.cur ^= (ndte & 1);
    S.t_nsub = ndte;
    return 0;
}

// Tripole: force the stresses symmetric across the seam on the resident state, as evp() does
// on the host arrays after the subcycle loop (12 x ice_HaloUpdate_stress, ice_dyn_evp.F90:1321-1389).
int cice_evp_hip_stress_halo(void)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    evp_launch_halo_stress(S.sig[S.cur], S.h_stress_dst, S.h_stress_src, S.n_stress, S.stream);
    HIPC(hipGetLastError());
    return 0;
}


// ---- next tier: the preparation phase of evp() on the device ----------------
int cice_evp_hip_set_prep_geometry(const int32_t *tmask, const int32_t *umask, const double *hm,
                                   const double *tarea, const double *uarea, const double *fcor_blk)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!tmask || !umask || !hm || !tarea || !uarea || !fcor_blk) return fail(-1, "null argument");
    State::Prep &Q = S.prep;
    if (S.plan.center_remote && S.d.ns_boundary_type == CICE_EVP_BND_TRIPOLE)
        return fail(-9, "device preparation: the T-grid halo across ranks is not implemented for tripole grids; "
                        "keep evp()'s host preparation and use cice_evp_hip_run on this configuration");
    auto B = [&](uint8_t *&p) -> int { if (!p) HIPC(hipMalloc((void **)&p, S.n)); return 0; };
    if (B(Q.tmask) || B(Q.umask) || B(Q.umask_old) || B(Q.tmphm)) return -1;
    auto D = [&](double *&p) -> int { return p ? 0 : alloc_d(&p, S.n); };
    if (D(Q.hm) || D(Q.tarea) || D(Q.uarea) || D(Q.fcor) || D(Q.tmass) || D(Q.umass) || D(Q.maskd) ||
        D(Q.ss_tltxU) || D(Q.ss_tltyU) || D(Q.strairxU) || D(Q.strairyU) || D(Q.strtltx) || D(Q.strtlty)) return -1;
    for (auto &q : Q.t)
        if (D(q)) return -1;
    if (!Q.flagword) HIPC(hipMalloc((void **)&Q.flagword, sizeof(unsigned)));
    Q.h8.resize(S.n);
    for (size_t k = 0; k < S.n; ++k) Q.h8[k] = tmask[k] != 0;
    HIPC(hipMemcpy(Q.tmask, Q.h8.data(), S.n, hipMemcpyHostToDevice));
    for (size_t k = 0; k < S.n; ++k) Q.h8[k] = umask[k] != 0;
    HIPC(hipMemcpy(Q.umask, Q.h8.data(), S.n, hipMemcpyHostToDevice));
    if (h2d(Q.hm, hm) || h2d(Q.tarea, tarea) || h2d(Q.uarea, uarea) || h2d(Q.fcor, fcor_blk)) return -1;
    const HaloPlan &P = S.plan;
    Q.n_center = (int)P.center_dst.size();
    if (Q.n_center && !Q.c_dst) {
        HIPC(hipMalloc((void **)&Q.c_dst, Q.n_center * sizeof(int32_t)));
        HIPC(hipMalloc((void **)&Q.c_src, Q.n_center * sizeof(int32_t)));
        HIPC(hipMalloc((void **)&Q.c_vsign, Q.n_center));
        HIPC(hipMemcpy(Q.c_dst, P.center_dst.data(), Q.n_center * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Q.c_src, P.center_src.data(), Q.n_center * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Q.c_vsign, P.center_vsign.data(), Q.n_center, hipMemcpyHostToDevice));
    }
    HIPC(hipStreamSynchronize(S.stream));
    Q.geo = true;
    return 0;
}

int cice_evp_hip_prep(const cice_evp_hip_prep_params *pp, const double *const *tfields11,
                      const double *const *fields32, int32_t *iceTmask, int32_t *iceUmask,
                      double *strintxU, double *strintyU, double *strocnxU, double *strocnyU)
{
    if (!S.ready) return fail(-1, "not initialised");
    State::Prep &Q = S.prep;
    if (!Q.geo) return fail(-1, "cice_evp_hip_set_prep_geometry was not called");
    if (!pp || !tfields11 || !fields32 || !iceTmask || !iceUmask) return fail(-1, "null argument");
    for (int k = 0; k < 11; ++k)
        if (!tfields11[k]) return fail(-1, "null T-grid field %d", k);
    for (int k = 0; k < 12; ++k)
        if (!fields32[k]) return fail(-1, "null stress field %d", k);
    if (!fields32[F_UVEL] || !fields32[F_VVEL]) return fail(-1, "null velocity field");
    HIPC(hipEventRecord(S.ev2, S.stream));
    S.cur = 0;
    for (int k = 0; k < 11; ++k)
        if (h2d(Q.t[k], tfields11[k])) return -1;
    for (int k = 0; k < 12; ++k)
        if (h2d(S.sig[0][k], fields32[k])) return -1;
    if (h2d(S.u[0], fields32[F_UVEL]) || h2d(S.v[0], fields32[F_VVEL])) return -1;
    bool tbu_zero = true;
    if (fields32[F_TBU]) {
        if (h2d(S.in[F_TBU], fields32[F_TBU])) return -1;
        for (size_t k = 0; k < S.n && tbu_zero; ++k) tbu_zero = fields32[F_TBU][k] == 0.0;
    } else {
        HIPC(hipMemsetAsync(S.in[F_TBU], 0, S.n * sizeof(double), S.stream));
    }
    for (size_t k = 0; k < S.n; ++k) Q.h8[k] = iceUmask[k] != 0;
    HIPC(hipMemcpyAsync(Q.umask_old, Q.h8.data(), S.n, hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemsetAsync(Q.flagword, 0, sizeof(unsigned), S.stream));
    HIPC(hipEventRecord(S.ev3, S.stream));

    EvpPrep P{};
    P.nx = S.d.nx_block; P.ny = S.d.ny_block; P.plane = S.plane; P.blk = S.blk;
    P.tmask = Q.tmask; P.umask = Q.umask; P.umask_old = Q.umask_old;
    P.hm = Q.hm; P.tarea = Q.tarea; P.uarea = Q.uarea; P.fcor = Q.fcor;
    for (int k = 0; k < 11; ++k) P.t[k] = Q.t[k];
    P.tmass = Q.tmass; P.umass = Q.umass; P.maskd = Q.maskd; P.tmphm = Q.tmphm;
    P.ss_tltxU = Q.ss_tltxU; P.ss_tltyU = Q.ss_tltyU; P.strairxU = Q.strairxU; P.strairyU = Q.strairyU;
    P.strtltx = Q.strtltx; P.strtlty = Q.strtlty;
    P.aiU = S.in[F_AIX]; P.cdn_ocnU = S.in[F_CW]; P.uocnU = S.in[F_UOCN]; P.vocnU = S.in[F_VOCN];
    P.umassdti = S.in[F_UMASSDTI]; P.fm = S.in[F_FM]; P.waterx = S.in[F_WATERX]; P.watery = S.in[F_WATERY];
    P.forcex = S.in[F_FORCEX]; P.forcey = S.in[F_FORCEY];
    P.uvel_init = S.in[F_UVEL_INIT]; P.vvel_init = S.in[F_VVEL_INIT];
    P.uvel = S.u[0]; P.vvel = S.v[0];
    for (int k = 0; k < 12; ++k) P.sig[k] = S.sig[0][k];
    P.mask = S.mask; P.flagword = Q.flagword;
    P.dt = pp->dt; P.rhoi = pp->rhoi; P.rhos = pp->rhos; P.gravit = pp->gravit;
    P.dyn_area_min = pp->dyn_area_min; P.dyn_mass_min = pp->dyn_mass_min;
    P.cosw = S.prm.cosw; P.sinw = S.prm.sinw; P.ssh_coupled = pp->ssh_stress_coupled;

    evp_launch_prep1(P, S.d.nblocks, S.stream);
    auto halo = [&](std::initializer_list<std::pair<double *, bool>> arrs) {
        EvpPrepHalo H{};
        for (const auto &a : arrs) { H.a[H.narr] = a.first; H.is_vec[H.narr] = a.second; ++H.narr; }
        H.dst = Q.c_dst; H.src = Q.c_src; H.vsign = (const signed char *)Q.c_vsign; H.n = Q.n_center;
        evp_launch_halo_center(H, S.stream);
    };
    // ice_dyn_evp.F90:413-428: iceTmask; tmass, aice_init, cdn_ocn (scalars); uocn, vocn, ss_tltx/y (vectors)
    halo({{Q.maskd, false}, {Q.tmass, false}, {Q.t[3], false}, {Q.t[4], false},
          {Q.t[5], true}, {Q.t[6], true}, {Q.t[7], true}, {Q.t[8], true}});
    halo({{Q.t[9], true}, {Q.t[10], true}});                 // :466-469 (calc_strair branch)
    if (S.plan.center_remote) {
        // neighbours on other ranks (no tripole fold here: centre and corner fields mirror the same
        // cells, so the velocity exchange carries pairs of T-grid fields)
        double *pairs[5][2] = {{Q.maskd, Q.tmass}, {Q.t[3], Q.t[4]}, {Q.t[5], Q.t[6]}, {Q.t[7], Q.t[8]}, {Q.t[9], Q.t[10]}};
        for (auto &pr : pairs)
            if (int rc = halo_remote_pair(pr[0], pr[1])) return rc;
    }
    evp_launch_prep_average(P, S.d.nblocks, S.stream);
    evp_launch_prep2(P, S.d.nblocks, S.stream);
    // ghost velocities before the loop (:729-732): the same exchange as inside the loop
    {
        const bool pushed = S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH);
        if (pushed)      // the in-kernel push only exists inside the subcycle kernel: use the gather lists here
            evp_launch_halo_local(S.u[0], S.v[0], S.h_local_dst, S.h_local_src,
                                  (const signed char *)S.h_local_sign, S.n_local, S.stream);
        if (int rc = halo_uv(0)) return rc;
    }
    for (int k = 0; k < 12; ++k)
        HIPC(hipMemcpyAsync(S.sig[1][k], S.sig[0][k], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    HIPC(hipMemcpyAsync(S.u[1], S.u[0], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    HIPC(hipMemcpyAsync(S.v[1], S.v[0], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    evp_launch_vrelfac(S.in[F_AIX], S.in[F_CW], S.prm.rhow, S.vrelfac, S.n, S.stream);
    HIPC(hipEventRecord(S.ev1, S.stream));
    // masks and the shortcut flag back to the host
    unsigned flagword = 0;
    HIPC(hipMemcpyAsync(S.hmask.data(), S.mask, S.n, hipMemcpyDeviceToHost, S.stream));
    HIPC(hipMemcpyAsync(&flagword, Q.flagword, sizeof(unsigned), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
    S.t_h2d_ms = ms;
    HIPC(hipEventElapsedTime(&ms, S.ev3, S.ev1));
    Q.t_ms = ms;
    for (size_t k = 0; k < S.n; ++k) iceTmask[k] = (S.hmask[k] & 1u) ? 1 : 0;
    // dyn_prep2 writes iceUmask on the physical cells only (:761-764) and zeroes the stress
    // divergence / ocean stress off the ice there (:776-781); they are the caller's arrays
    {
        const int nx = S.d.nx_block;
        for (int b = 0; b < S.d.nblocks; ++b)
            for (int j = S.jlo[b]; j <= S.jhi[b]; ++j)
                for (int i = S.ilo[b]; i <= S.ihi[b]; ++i) {
                    const size_t c = b * S.plane + (size_t)(j - 1) * nx + (i - 1);
                    iceUmask[c] = (S.hmask[c] & 2u) ? 1 : 0;
                    if (S.hmask[c] & 2u) continue;
                    if (strintxU) strintxU[c] = 0.0;
                    if (strintyU) strintyU[c] = 0.0;
                    if (strocnxU) strocnxU[c] = 0.0;
                    if (strocnyU) strocnyU[c] = 0.0;
                }
    }
    S.flags &= ~(EVP_F_WATER_IS_OCN | EVP_F_TBU_ZERO);
    if (!(flagword & 1u)) S.flags |= EVP_F_WATER_IS_OCN;
    if (tbu_zero) S.flags |= EVP_F_TBU_ZERO;
    S.uploaded = true;
    return tune_after_upload();
}

// Address of a caller's array, for hosts whose language will not hand out the address of an
// object without a TARGET-like attribute (CICE's module arrays): the Fortran shim builds the
// pointer tables of cice_evp_hip_prep / _download with it.
void *cice_evp_hip_addr(const void *array) { return const_cast<void *>(array); }

// ice strength, computed by the host (icepack_ice_strength + its halo update, ice_dyn_evp.F90:541-552,
// 727-728) from the masks cice_evp_hip_prep returned
int cice_evp_hip_set_strength(const double *strength)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (!strength) return fail(-1, "null argument");
    return h2d(S.in[F_STRENGTH], strength);
}

// products of the preparation phase that stay on the device, for hosts that need them
// (coupling diagnostics) and for the tests
int cice_evp_hip_prep_fetch(int32_t which, double *dst)
{
    if (!S.ready || !S.uploaded || !S.prep.geo) return fail(-1, "no prepared state");
    if (!dst) return fail(-1, "null argument");
    State::Prep &Q = S.prep;
    const double *tab[20] = {S.in[F_AIX], S.in[F_CW], S.in[F_UOCN], S.in[F_VOCN], S.in[F_UMASSDTI], S.in[F_FM],
                             S.in[F_WATERX], S.in[F_WATERY], S.in[F_FORCEX], S.in[F_FORCEY], S.in[F_UVEL_INIT],
                             S.in[F_VVEL_INIT], Q.strtltx, Q.strtlty, Q.strairxU, Q.strairyU, Q.tmass, Q.umass,
                             S.u[S.cur], S.v[S.cur]};
    if (which < 0 || which >= 20) return fail(-1, "prep_fetch: which = %d", (int)which);
    if (d2h(dst, tab[which])) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

// Record a HIP event on the library's stream: which = 0 (begin) or 1 (end) of a caller's
// timed region; the elapsed time is reported by cice_evp_hip_get_timings()[6].
int cice_evp_hip_mark(int32_t which)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (which < 0 || which > 1) return fail(-1, "mark index");
    HIPC(hipEventRecord(S.evm[which], S.stream));
    S.marked[which] = true;
    return 0;
}

// ---- next tier: deformations and dyn_finish on the resident final state ----
int cice_evp_hip_set_post_geometry(const double *dxU, const double *dyU, const double *tarear)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!dxU || !dyU || !tarear) return fail(-1, "null argument");
    const double *src[3] = {dxU, dyU, tarear};
    for (int k = 0; k < 3; ++k) {
        if (!S.post_geo[k] && alloc_d(&S.post_geo[k], S.n)) return -1;
        if (h2d(S.post_geo[k], src[k])) return -1;
    }
    for (auto &p : S.post_out)
        if (!p && alloc_d(&p, S.n)) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    S.have_post_geo = true;
    return 0;
}

int cice_evp_hip_deformations(double *divu, double *shear, double *vort, double *rdg_conv, double *rdg_shear)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (!S.have_post_geo) return fail(-1, "cice_evp_hip_set_post_geometry not called");
    EvpArgs A;
    fill_args(A, S.cur, 0);
    evp_launch_deformations(A, S.d.nblocks, S.prm.strict != 0, S.post_geo[0], S.post_geo[1], S.post_geo[2],
                            S.post_out[0], S.post_out[1], S.post_out[2], S.post_out[3], S.post_out[4], S.stream);
    double *dst[5] = {divu, shear, vort, rdg_conv, rdg_shear};
    for (int k = 0; k < 5; ++k)
        if (dst[k] && d2h(dst[k], S.post_out[k])) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

int cice_evp_hip_dyn_finish(double *strocnxU, double *strocnyU)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (!strocnxU || !strocnyU) return fail(-1, "null argument");
    for (int k = 5; k < 7; ++k)
        if (!S.post_out[k] && alloc_d(&S.post_out[k], S.n)) return -1;
    // inout: cells outside the ice keep the caller's values (dyn_prep2 zeroes them, :776-784)
    if (h2d(S.post_out[5], strocnxU) || h2d(S.post_out[6], strocnyU)) return -1;
    EvpArgs A;
    fill_args(A, S.cur, 0);
    evp_launch_dyn_finish(A, S.d.nblocks, S.prm.strict != 0, S.post_out[5], S.post_out[6], S.stream);
    if (d2h(strocnxU, S.post_out[5]) || d2h(strocnyU, S.post_out[6])) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

// Page-lock a caller-owned host array so that the H2D/D2H copies of cice_evp_hip_run become
// direct DMA (pageable copies of gx1's 50 arrays cost ~14 ms per call, pinned ~1.5 ms).  For
// arrays that live as long as the library is initialised -- CICE's module arrays.  Idempotent;
// unregistered by cice_evp_hip_finalize.
int cice_evp_hip_pin_host(const void *ptr, int64_t bytes)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!ptr || bytes <= 0) return fail(-1, "bad argument");
    auto it = S.pinned.find(ptr);
    if (it != S.pinned.end() && it->second >= (size_t)bytes) return 0;
    if (it != S.pinned.end()) {
        (void)hipHostUnregister(const_cast<void *>(ptr));
        S.pinned.erase(it);
    }
    HIPC(hipHostRegister(const_cast<void *>(ptr), (size_t)bytes, hipHostRegisterDefault));
    S.pinned[ptr] = (size_t)bytes;
    return 0;
}

int cice_evp_hip_sync(void)
{
    if (!S.ready) return fail(-1, "not initialised");
    HIPC(hipStreamSynchronize(S.stream));
    if (int rc = direct_check_error()) return rc;
    return resident_check_error();
}

int cice_evp_hip_download(double *const *f)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    HIPC(hipStreamSynchronize(S.stream));
    if (int rc = direct_check_error()) return rc;
    if (int rc = resident_check_error()) return rc;
    HIPC(hipEventRecord(S.ev2, S.stream));
    for (int k = 0; k < 12; ++k)
        if (f[k] && d2h(f[k], S.sig[S.cur][k])) return -1;
    const int outs[4] = {F_STRINTX, F_STRINTY, F_TAUBX, F_TAUBY};
    for (int o : outs)
        if (f[o] && d2h(f[o], S.in[o])) return -1;
    if (f[F_UVEL] && d2h(f[F_UVEL], S.u[S.cur])) return -1;
    if (f[F_VVEL] && d2h(f[F_VVEL], S.v[S.cur])) return -1;
    HIPC(hipEventRecord(S.ev3, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
    S.t_d2h_ms = ms;
    if (S.t_nsub > 0 && hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess) S.t_loop_ms = ms;
    return 0;
}

int cice_evp_hip_run(double *stressp_1, double *stressp_2, double *stressp_3, double *stressp_4,
                     double *stressm_1, double *stressm_2, double *stressm_3, double *stressm_4,
                     double *stress12_1, double *stress12_2, double *stress12_3, double *stress12_4,
                     const double *strength, const double *cdn_ocnU, const double *aiU,
                     const double *uocnU, const double *vocnU, const double *waterxU,
                     const double *wateryU, const double *forcexU, const double *forceyU,
                     const double *umassdti, const double *fmU, double *strintxU, double *strintyU,
                     const double *TbU, double *taubxU, double *taubyU, double *uvel, double *vvel,
                     const double *uvel_init, const double *vvel_init, const int32_t *iceTmask,
                     const int32_t *iceUmask, int32_t ndte)
{
    double *f[F_COUNT] = {stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2,
                          stressm_3, stressm_4, stress12_1, stress12_2, stress12_3, stress12_4,
                          (double *)strength, (double *)cdn_ocnU, (double *)aiU, (double *)uocnU,
                          (double *)vocnU, (double *)waterxU, (double *)wateryU, (double *)forcexU,
                          (double *)forceyU, (double *)umassdti, (double *)fmU, strintxU, strintyU,
                          (double *)TbU, taubxU, taubyU, uvel, vvel, (double *)uvel_init,
                          (double *)vvel_init};
    if (int rc = cice_evp_hip_upload(f, iceTmask, iceUmask)) return rc;
    if (int rc = cice_evp_hip_subcycle(ndte)) return rc;
    // only the documented outputs travel back
    double *o[F_COUNT] = {};
    for (int k = 0; k < 12; ++k) o[k] = f[k];
    o[F_STRINTX] = strintxU; o[F_STRINTY] = strintyU; o[F_TAUBX] = taubxU; o[F_TAUBY] = taubyU;
    o[F_UVEL] = uvel; o[F_VVEL] = vvel;
    return cice_evp_hip_download(o);
}

int cice_evp_hip_get_timings(double *out, int32_t n)
{
    float ms = 0;
    if (S.ready && S.t_nsub > 0 && hipEventQuery(S.ev1) == hipSuccess &&
        hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess)
        S.t_loop_ms = ms;
    double marks_ms = -1.0;
    if (S.ready && S.marked[0] && S.marked[1] && hipEventQuery(S.evm[1]) == hipSuccess &&
        hipEventElapsedTime(&ms, S.evm[0], S.evm[1]) == hipSuccess)
        marks_ms = ms;
    const double v[11] = {S.t_loop_ms, S.t_h2d_ms, S.t_d2h_ms, (double)S.t_nsub,
                         S.res_mode == 1 ? 1.0 / std::max(S.t_nsub, 1) :
                         1.0 + ((S.n_local > 0 && !(S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH))) ? 1.0 : 0.0) +
                             (S.plan.peers.empty() ? 0.0 : (S.direct.on ? (use_riding_exchange() ? 0.0 : 1.0) : 2.0)) + ((S.n_seam + S.n_pole + S.n_late) > 0 ? 1.0 : 0.0),
                         (double)(S.res_mode == 1 ? (S.res_gen == 2 ? 2000 + S.res2_logw : 1000 + S.res_logw) : S.tyb), marks_ms, S.t_stream_probe_ms, S.t_res_probe_ms,
                          S.plan.peers.empty() ? 0.0 : (S.direct.on ? 2.0 : 1.0), S.prep.t_ms};
    for (int k = 0; k < n && k < 11; ++k) out[k] = v[k];
    return 0;
}

// Per-launch durations of the two kernels of a subcycle, measured with HIP events
// on the library's own stream (the stream the kernels are launched on).  Works on
// the resident state without advancing it: the launches write the ping-pong
// "next" buffers, which the next real subcycle overwrites anyway.
int cice_evp_hip_time_kernels(int32_t nrep, double *out3)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (nrep < 1) nrep = 1;
    std::vector<hipEvent_t> ev(2 * (size_t)nrep + 2);
    for (auto &e : ev) HIPC(hipEventCreate(&e));
    const bool strict = S.prm.strict != 0;
    const int cap = cap_mode();
    EvpArgs A;
    fill_args(A, S.cur, 0);
    double sum[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
        for (int r = 0; r < nrep; ++r) {
            HIPC(hipEventRecord(ev[2 * r], S.stream));
            if (which == 0) evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, S.tyb, strict, cap, S.stream);
            else evp_launch_halo_local(S.u[S.cur ^ 1], S.v[S.cur ^ 1], S.h_local_dst, S.h_local_src,
                                       (const signed char *)S.h_local_sign, S.n_local, S.stream);
            HIPC(hipEventRecord(ev[2 * r + 1], S.stream));
        }
        HIPC(hipStreamSynchronize(S.stream));
        for (int r = 0; r < nrep; ++r) {
            float ms = 0;
            HIPC(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]));
            sum[which] += ms;
        }
    }
    // back-to-back period of the stencil kernel (launch gap included)
    HIPC(hipEventRecord(ev[2 * nrep], S.stream));
    for (int r = 0; r < nrep; ++r)
        evp_launch_subcycle(A, S.max_ni, S.max_nj, S.d.nblocks, S.tyb, strict, cap, S.stream);
    HIPC(hipEventRecord(ev[2 * nrep + 1], S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, ev[2 * nrep], ev[2 * nrep + 1]));
    out3[0] = sum[0] / nrep;
    out3[1] = S.n_local > 0 ? sum[1] / nrep : 0.0;
    out3[2] = ms / nrep;
    for (auto &e : ev) (void)hipEventDestroy(e);
    return 0;
}

int cice_evp_hip_comm_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    NCCLC(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

int cice_evp_hip_comm_init(const void *id128)
{
    if (!S.ready) return fail(-1, "not initialised");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    HIPC(hipSetDevice(S.device));
    NCCLC(ncclCommInitRank(&S.comm, S.d.nranks, id, S.d.rank));
    S.have_comm = true;
    if (halo_choice() == 1) { S.direct.why = "CICE_EVP_HIP_HALO=rccl"; return 0; }
    // Mailbox halo: every step below is followed by an agreement (all-reduce of "still fine"),
    // so that either all ranks switch to it or all stay on RCCL.
    const int nr = S.d.nranks;
    char *d_blobs = nullptr;
    int *d_ok = nullptr;
    HIPC(hipMalloc((void **)&d_blobs, (size_t)nr * CICE_EVP_HIP_HALO_BLOB));
    HIPC(hipMalloc((void **)&d_ok, sizeof(int)));
    auto agree = [&](int mine, int &all) -> int {
        HIPC(hipMemcpy(d_ok, &mine, sizeof(int), hipMemcpyHostToDevice));
        NCCLC(ncclAllReduce(d_ok, d_ok, 1, ncclInt, ncclMin, S.comm, S.stream));
        HIPC(hipStreamSynchronize(S.stream));
        HIPC(hipMemcpy(&all, d_ok, sizeof(int), hipMemcpyDeviceToHost));
        return 0;
    };
    std::vector<char> mine(CICE_EVP_HIP_HALO_BLOB, 0), all((size_t)nr * CICE_EVP_HIP_HALO_BLOB, 0);
    int ok = direct_export(*reinterpret_cast<HaloBlob *>(mine.data())) == 0, all_ok = 0;
    std::string why = ok ? "" : g_err;
    HIPC(hipMemcpy(d_blobs + (size_t)S.d.rank * CICE_EVP_HIP_HALO_BLOB, mine.data(), CICE_EVP_HIP_HALO_BLOB, hipMemcpyHostToDevice));
    NCCLC(ncclAllGather(d_blobs + (size_t)S.d.rank * CICE_EVP_HIP_HALO_BLOB, d_blobs, CICE_EVP_HIP_HALO_BLOB, ncclChar, S.comm, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemcpy(all.data(), d_blobs, all.size(), hipMemcpyDeviceToHost));
    if (ok) {
        ok = direct_import(reinterpret_cast<const HaloBlob *>(all.data()), nr) == 0;
        if (!ok) why = g_err;
    }
    if (agree(ok, all_ok)) return -1;
    if (all_ok) {
        ok = direct_probe() == 0;
        if (!ok) why = g_err;
        if (agree(ok, all_ok)) return -1;
    }
    if (all_ok && S.res_remote) {           // resident kernel across GPUs: its own probe, same agreement
        int res_ok = resident_remote_probe() == 0, res_all = 0;
        const std::string why_res = res_ok ? "" : g_err;
        if (agree(res_ok, res_all)) return -1;
        if (!res_all) {
            S.res_remote = false;
            if (env("CICE_EVP_HIP_VERBOSE"))
                std::fprintf(stderr, "[cice_evp_hip] rank %d: resident kernel across GPUs off (%s)\n", S.d.rank,
                             why_res.empty() ? "another rank's probe failed" : why_res.c_str());
        }
    }
    (void)hipFree(d_blobs);
    (void)hipFree(d_ok);
    S.direct.on = all_ok != 0;
    S.direct.why = S.direct.on ? "" : (why.empty() ? "another rank could not set it up" : why);
    if (!S.direct.on && halo_choice() == 2)
        return fail(-8, "CICE_EVP_HIP_HALO=direct but the mailbox halo is unavailable: %s", S.direct.why.c_str());
    if (!S.direct.on && env("CICE_EVP_HIP_VERBOSE"))
        std::fprintf(stderr, "[cice_evp_hip] rank %d: mailbox halo off (%s), using RCCL\n", S.d.rank, S.direct.why.c_str());
    g_err.clear();
    return 0;
}

int cice_evp_hip_halo_export(void *blob)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!blob) return fail(-1, "null blob");
    HIPC(hipSetDevice(S.device));
    std::vector<char> tmp(CICE_EVP_HIP_HALO_BLOB, 0);
    if (int rc = direct_export(*reinterpret_cast<HaloBlob *>(tmp.data()))) return rc;
    std::memcpy(blob, tmp.data(), CICE_EVP_HIP_HALO_BLOB);
    return 0;
}

int cice_evp_hip_halo_import(const void *blobs, int32_t nranks)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!blobs) return fail(-1, "null blobs");
    HIPC(hipSetDevice(S.device));
    std::vector<HaloBlob> B((size_t)std::max(nranks, 0));
    for (int r = 0; r < nranks; ++r)
        std::memcpy(&B[r], (const char *)blobs + (size_t)r * CICE_EVP_HIP_HALO_BLOB, sizeof(HaloBlob));
    if (int rc = direct_import(B.data(), nranks)) return rc;
    if (int rc = direct_probe()) return rc;     // collective; a failure here is fatal for the caller
    S.direct.on = true;
    if (S.res_remote)
        if (int rc = resident_remote_probe()) return rc;
    return 0;
}

// Host-only: build the halo plan for `dims` without touching a device (tests).
int cice_evp_hip_plan_build(const cice_evp_hip_dims *dims)
{
    if (!dims) return fail(-1, "null dims");
    if (!build_halo_plan(*dims, S.plan)) return fail(-3, "halo plan: %s", S.plan.error.c_str());
    return 0;
}

int cice_evp_hip_halo_plan(int32_t *counts4, int32_t *local_dst, int32_t *local_src,
                           int32_t *local_sign, int32_t *peer_rank, int32_t *peer_nsend,
                           int32_t *peer_nrecv, int32_t *send_src, int32_t *recv_dst)
{
    const HaloPlan &P = S.plan;
    size_t ns = 0, nr = 0;
    for (const HaloPeer &p : P.peers) {
        ns += p.send_src.size();
        nr += p.recv_dst.size();
    }
    if (counts4) {
        counts4[0] = (int32_t)P.local_dst.size();
        counts4[1] = (int32_t)P.peers.size();
        counts4[2] = (int32_t)ns;
        counts4[3] = (int32_t)nr;
    }
    for (size_t k = 0; k < P.local_dst.size(); ++k) {
        if (local_dst) local_dst[k] = P.local_dst[k];
        if (local_src) local_src[k] = P.local_src[k];
        if (local_sign) local_sign[k] = P.local_sign[k];
    }
    size_t so = 0, ro = 0;
    for (size_t q = 0; q < P.peers.size(); ++q) {
        const HaloPeer &p = P.peers[q];
        if (peer_rank) peer_rank[q] = p.rank;
        if (peer_nsend) peer_nsend[q] = (int32_t)p.send_src.size();
        if (peer_nrecv) peer_nrecv[q] = (int32_t)p.recv_dst.size();
        if (send_src) std::copy(p.send_src.begin(), p.send_src.end(), send_src + so);
        if (recv_dst) std::copy(p.recv_dst.begin(), p.recv_dst.end(), recv_dst + ro);
        so += p.send_src.size();
        ro += p.recv_dst.size();
    }
    return 0;
}

// Tripole part of the plan built by the last init / plan_build (tests): counts3 =
// {pairs, poles, late copies}; lists may be NULL.
int cice_evp_hip_peer_plan(int32_t *send_dst, int32_t *recv_gid)
{
    const HaloPlan &P = S.plan;
    size_t so = 0, ro = 0;
    for (const HaloPeer &p : P.peers) {
        for (size_t k = 0; k < p.send_dst.size(); ++k)
            if (send_dst) send_dst[so + k] = p.send_dst[k];
        for (size_t k = 0; k < p.recv_gid.size(); ++k)
            if (recv_gid) recv_gid[ro + k] = p.recv_gid[k];
        so += p.send_dst.size();
        ro += p.recv_gid.size();
    }
    return 0;
}

int cice_evp_hip_stress_plan(int32_t *count, int32_t *dst, int32_t *src)
{
    const HaloPlan &P = S.plan;
    if (count) *count = (int32_t)P.stress_dst.size();
    for (size_t k = 0; k < P.stress_dst.size(); ++k) {
        if (dst) dst[k] = P.stress_dst[k];
        if (src) src[k] = P.stress_src[k];
    }
    return 0;
}

int cice_evp_hip_seam_plan(int32_t *counts3, int32_t *seam_a, int32_t *seam_b, int32_t *seam_pole,
                           int32_t *late_dst, int32_t *late_src, int32_t *late_sign)
{
    const HaloPlan &P = S.plan;
    if (counts3) {
        counts3[0] = (int32_t)P.seam_a.size();
        counts3[1] = (int32_t)P.seam_pole.size();
        counts3[2] = (int32_t)P.late_dst.size();
    }
    for (size_t k = 0; k < P.seam_a.size(); ++k) {
        if (seam_a) seam_a[k] = P.seam_a[k];
        if (seam_b) seam_b[k] = P.seam_b[k];
    }
    for (size_t k = 0; k < P.seam_pole.size(); ++k)
        if (seam_pole) seam_pole[k] = P.seam_pole[k];
    for (size_t k = 0; k < P.late_dst.size(); ++k) {
        if (late_dst) late_dst[k] = P.late_dst[k];
        if (late_src) late_src[k] = P.late_src[k];
        if (late_sign) late_sign[k] = P.late_sign[k];
    }
    return 0;
}

}  // extern "C"
