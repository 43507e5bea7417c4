// Host side, common part: error text, the device state, copies, static metric terms
// (init_dyn_shared), index lists, ghost-image push table, kernel argument block.
#include "evp_host.h"

// host arithmetic of derive_metrics must round every operation (no FMA)
#pragma clang fp contract(off)

// (the kernel objects are compiled once and linked into both libraries: their two switches come through this function)
#ifdef CICE_EVP_HIP_TESTING
const char *evp_env_test(const char *key) { return std::getenv(key); }
#else
const char *evp_env_test(const char *) { return nullptr; }
#endif

namespace evp_host {

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

State S;

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
    for (auto &p : S.sig_snap) F(p);
    F(S.hte);
    F(S.htn);
    F(S.vrelfac);
    F(S.res_err); F(S.res_tab);
    F(S.res2_ring); F(S.res2_cnt); F(S.res2_pub); F(S.res2_perm); F(S.res2_late); F(S.res2_nact); F(S.res2_nlate); F(S.res2_live); F(S.res2_celltile); F(S.res2_cuload); F(S.res2_prof);
    if (S.res2_rec_owned) { F(S.res2_rec[0]); F(S.res2_rec[1]); }
    S.res2_rec[0] = S.res2_rec[1] = nullptr;
    S.res2_rec_owned = true;
    S.res_remote = false;
    S.res2_par = 0; S.res2_epoch = 0;
    F(S.res2_rimg); F(S.res2_peer_rec); F(S.res2_peer_rstride);
    F(S.res2_order); F(S.res2_seam); F(S.res2_img3);
    if (S.res2_raw_owned) { F(S.res2_rec_raw[0]); F(S.res2_rec_raw[1]); }
    S.res2_rec_raw[0] = S.res2_rec_raw[1] = nullptr;
    S.res2_raw_owned = true;
    F(S.res2_rraw); F(S.res2_peer_raw); F(S.res2_peer_raw_stride);
    for (auto &p : S.post_geo) F(p);
    for (auto &p : S.post_out) F(p);
    F(S.push);
    F(S.mask);
    F(S.blk);
    F(S.h_local_dst);
    F(S.h_local_src);
    F(S.h_local_sign);
    F(S.h_seam_a); F(S.h_seam_b); F(S.h_seam_pole); F(S.h_late_dst); F(S.h_late_src); F(S.h_late_sign);
    F(S.h_fin_dst); F(S.h_fin_a); F(S.h_fin_b); F(S.h_fin_coef);
    F(S.h_stress_dst); F(S.h_stress_src); F(S.h_stress_own_dst); F(S.h_stress_own_src); F(S.h_stress_corner_dst); F(S.h_stress_corner_src);
    {
        State::Prep &Q = S.prep;
        F(Q.tmask); F(Q.umask); F(Q.umask_old); F(Q.umask_old32); F(Q.tmphm); F(Q.hm); F(Q.tarea); F(Q.uarea); F(Q.fcor); F(Q.hwater); F(Q.aicen); F(Q.vicen); F(Q.tbt); Q.ncat = 0;
        for (auto &q : Q.t) F(q);
        F(Q.tmass); F(Q.umass); F(Q.maskd); F(Q.ss_tltxU); F(Q.ss_tltyU); F(Q.strairxU); F(Q.strairyU);
        F(Q.strtltx); F(Q.strtlty); F(Q.flagword); F(Q.c_dst); F(Q.c_src); F(Q.c_vsign); F(Q.tf_dst); F(Q.tf_a); F(Q.tf_b); F(Q.tf_flip); F(Q.tf_tmp);
        S.prep = State::Prep();
    }
    F(S.h_send_src);
    F(S.h_recv_dst);
    F(S.h_recv_sign);
    F(S.msk.send_src); F(S.msk.recv_dst); F(S.msk.recv_slot); F(S.msk.recv_sign); F(S.msk.send_addr); F(S.msk.send_pstride);
    S.msk = State::Masked();
    F(S.sendbuf);
    F(S.recvbuf);
    for (void *q : S.direct.opened) (void)hipIpcCloseMemHandle(q);
    F(S.direct.mailbox); F(S.direct.d_dx); F(S.direct.d_dx_m); F(S.direct.d_cnt); F(S.direct.send_addr); F(S.direct.send_pstride); F(S.direct.peer_flag);
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
    {
        State::FoldX &F = S.foldx;
        if (F.cells) (void)hipFree(F.cells);
        if (F.seam_dst) (void)hipFree(F.seam_dst);
        if (F.seam_slot) (void)hipFree(F.seam_slot);
        if (F.seam_one) (void)hipFree(F.seam_one);
        for (auto &p : F.dst) if (p) (void)hipFree(p);
        for (auto &p : F.scr) if (p) (void)hipFree(p);
        F = State::FoldX();
    }
    for (auto &kv : S.pinned) (void)hipHostUnregister(const_cast<void *>(kv.first));
    S.sig_valid = false;
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

// Device view of a caller's array if it lies inside a range registered (and mapped) by cice_evp_hip_pin_host
static void *mapped_view(const void *host, size_t bytes)
{
    auto it = S.pinned.upper_bound(host);
    if (it == S.pinned.begin()) return nullptr;
    --it;
    const char *base = (const char *)it->first;
    if (!it->second.dev || (const char *)host < base || (const char *)host + bytes > base + it->second.bytes) return nullptr;
    return (char *)it->second.dev + ((const char *)host - base);
}

static int run_batch(CopyBatch &B, bool to_device)
{
    const bool off = env_test("CICE_EVP_HIP_GATHER") && !std::atoi(env_test("CICE_EVP_HIP_GATHER"));
    EvpCopyTab T{};
    T.len = S.n;
    T.vec2 = 1;
    auto flush = [&]() {
        if (T.n > 0) evp_launch_copy_many(T, S.stream);
        T.n = 0;
        T.vec2 = 1;
    };
    for (auto &it : B.items) {
        double *dst = it.first;
        const double *src = it.second;
        const void *host = to_device ? (const void *)src : (const void *)dst;
        void *view = off ? nullptr : mapped_view(host, S.n * sizeof(double));
        if (!view) {
            if (to_device ? h2d(dst, src) : d2h(dst, src)) return -1;
            continue;
        }
        T.src[T.n] = to_device ? (const double *)view : src;
        T.dst[T.n] = to_device ? dst : (double *)view;
        if ((((uintptr_t)T.src[T.n]) | ((uintptr_t)T.dst[T.n])) & 15u) T.vec2 = 0;
        if (++T.n == EVP_COPY_MAX) flush();
    }
    flush();
    HIPC(hipGetLastError());
    B.items.clear();
    return 0;
}
// every host array of the batch lies in a range the caller page-locked and mapped (and the gather path is on)
bool batch_mapped(const CopyBatch &B, bool to_device)
{
    if (env_test("CICE_EVP_HIP_GATHER") && !std::atoi(env_test("CICE_EVP_HIP_GATHER"))) return false;
    for (auto &it : B.items)
        if (!mapped_view(to_device ? (const void *)it.second : (const void *)it.first, S.n * sizeof(double))) return false;
    return true;
}
// device -> mapped host arrays, only the cells whose mask byte has `bit` set (batch_mapped(B, false) must hold)
int d2h_batch_masked(CopyBatch &B, unsigned bit)
{
    EvpCopyTab T{};
    T.len = S.n;
    for (auto &it : B.items) {
        T.src[T.n] = it.second;
        T.dst[T.n] = (double *)mapped_view(it.first, S.n * sizeof(double));
        if (++T.n == EVP_COPY_MAX) { evp_launch_copy_many_masked(T, S.mask, bit, S.stream); T.n = 0; }
    }
    evp_launch_copy_many_masked(T, S.mask, bit, S.stream);
    HIPC(hipGetLastError());
    B.items.clear();
    return 0;
}
int h2d_batch(CopyBatch &B) { return run_batch(B, true); }
int d2h_batch(CopyBatch &B) { return run_batch(B, false); }

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
    if (S.d.ns_boundary_type == CICE_EVP_BND_TRIPOLE || S.d.ns_boundary_type == CICE_EVP_BND_TRIPOLET) S.flags |= EVP_F_DXHY_ARRAY;
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
    S.n_stress_own = (int)P.stress_own_dst.size();
    if (up32(P.stress_own_dst, S.h_stress_own_dst) || up32(P.stress_own_src, S.h_stress_own_src)) return -1;
    S.n_stress_corner = (int)P.stress_corner_dst.size();
    if (up32(P.stress_corner_dst, S.h_stress_corner_dst) || up32(P.stress_corner_src, S.h_stress_corner_src)) return -1;
    if (S.n_late) {
        HIPC(hipMalloc((void **)&S.h_late_sign, S.n_late));
        HIPC(hipMemcpy(S.h_late_sign, P.late_sign.data(), S.n_late, hipMemcpyHostToDevice));
    }
    S.n_fin = (int)P.fin_dst.size();
    // the list is only run on layouts that split the seam row (halo_uv: general form) or when forced for tests
    const bool fin_used = P.tail > 0 || (env_test("CICE_EVP_HIP_SEAM_FIN") && std::atoi(env_test("CICE_EVP_HIP_SEAM_FIN")));
    if (fin_used && S.n_fin > evp_halo_seam_fin_capacity()) return fail(-3, "tripole seam: %d cells to finalise on one rank (limit %d)", S.n_fin, evp_halo_seam_fin_capacity());
    if (up32(P.fin_dst, S.h_fin_dst) || up32(P.fin_a, S.h_fin_a) || up32(P.fin_b, S.h_fin_b)) return -1;
    if (S.n_fin) {
        HIPC(hipMalloc((void **)&S.h_fin_coef, S.n_fin));
        HIPC(hipMemcpy(S.h_fin_coef, P.fin_coef.data(), S.n_fin, hipMemcpyHostToDevice));
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

// Arithmetic variant of the kernels (template MODE, evp_math.h): 3 = capping == 1 and the reference's
// default scalars -- classic EVP (revp == 0), Ktens == 0, cosw == 1, sinw == 0 -- whose products with
// exactly 1.0 / sums with exactly 0.0 the kernels then leave out, bit-neutral (CICE_EVP_HIP_SIMPLE=0: off)
int cap_mode()
{
    const bool simple_ok = !(env_test("CICE_EVP_HIP_SIMPLE") && !std::atoi(env_test("CICE_EVP_HIP_SIMPLE")));
    if (simple_ok && S.prm.capping == 1.0 && S.prm.revp == 0.0 && S.prm.Ktens == 0.0 && S.prm.cosw == 1.0 &&
        S.prm.sinw == 0.0)
        return 3;
    if (S.prm.capping == 1.0) return 1;
    if (S.prm.capping == 0.0) return 0;
    return -1;
}

}  // namespace evp_host
