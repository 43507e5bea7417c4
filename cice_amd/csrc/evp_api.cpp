// =====================================================================
// C ABI of the MI355X-native EVP core (see include/cice_evp_hip.h): life cycle, the
// per-call entry points, the post-loop kernels (f-1, f-3) and introspection.  The device
// state and the helpers live in evp_host*.cpp (overview in evp_host.h).
// =====================================================================
#include "evp_host.h"

using namespace evp_host;

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

// host only: the C-grid fold step of one location for the decomposition in `dims` (counts first, then the lists)
#ifdef CICE_EVP_HIP_TESTING
int cice_evp_hip_cgrid_fold_plan(const cice_evp_hip_dims *dims, int32_t loc, int32_t *count, int32_t *dst, int32_t *a,
                                 int32_t *b, int32_t *flip)
{
    if (!dims || !count || loc < 0 || loc > 3) return fail(-1, "bad argument");
    if (dims->ns_boundary_type != CICE_EVP_BND_TRIPOLE && dims->ns_boundary_type != CICE_EVP_BND_TRIPOLET) return fail(-1, "not a tripole grid");
    FoldList L;
    build_fold_list(*dims, loc, L);
    *count = (int32_t)L.dst.size();
    if (dst && a && b && flip)
        for (size_t k = 0; k < L.dst.size(); ++k) { dst[k] = L.dst[k]; a[k] = L.a[k]; b[k] = L.b[k]; flip[k] = L.flip[k]; }
    return 0;
}

int cice_evp_hip_cgrid_window_plan(const cice_evp_hip_dims *dims, int32_t ox, int32_t oy, int32_t *ntiles, int32_t *tiles4, int32_t *tab)
{
    return cice_evp_hip_cgrid_window_plan_ext(dims, ox, oy, 0, ntiles, tiles4, tab);
}

int cice_evp_hip_cgrid_window_plan_ext(const cice_evp_hip_dims *dims, int32_t ox, int32_t oy, int32_t extra, int32_t *ntiles, int32_t *tiles4,
                                       int32_t *tab)
{
    if (!dims || !ntiles || ox < 4 || oy < 4 || extra < 0 || extra > 2) return fail(-1, "bad argument");
    HaloPlan P;
    if (!build_halo_plan(*dims, P)) return fail(-3, "halo plan: %s", P.error.c_str());
    std::vector<int32_t> t4, tb;
    if (extra == 2) {                       // the resident kernel's windows on a tripole grid (17 x 17 positions; ox, oy unused)
        std::vector<int32_t> t2;
        std::string why;
        if (!build_fold_window_table(*dims, P, t4, t2, tb, why)) return fail(-5, "fold windows: %s", why.c_str());
    } else {
        build_window_table(*dims, P, ox, oy, 1 << 20, t4, tb, extra);
    }
    *ntiles = (int32_t)(t4.size() / 4);
    if (tiles4) std::copy(t4.begin(), t4.end(), tiles4);
    if (tab) std::copy(tb.begin(), tb.end(), tab);
    return 0;
}

int cice_evp_hip_cgrid_window_deps(const cice_evp_hip_dims *dims, int32_t *n_windows, int32_t *n_edges, int32_t *n_oneway, int32_t *n_unsafe)
{
    if (!dims || !n_edges || !n_oneway || !n_unsafe) return fail(-1, "bad argument");
    HaloPlan P;
    if (!build_halo_plan(*dims, P)) return fail(-3, "halo plan: %s", P.error.c_str());
    const bool tripole = dims->ns_boundary_type == CICE_EVP_BND_TRIPOLE;
    std::vector<int32_t> t4, t2, tb;
    if (tripole) {
        std::string why;
        if (!build_fold_window_table(*dims, P, t4, t2, tb, why)) return fail(-5, "fold windows: %s", why.c_str());
    } else {
        build_window_table(*dims, P, 16, 16, 1 << 20, t4, tb, 1);
    }
    int ne = 0, n1 = 0;
    *n_unsafe = cgres_dependencies(*dims, tripole, t4, tb, nullptr, &ne, &n1);
    *n_edges = ne;
    *n_oneway = n1;
    if (n_windows) *n_windows = (int32_t)(t4.size() / 4);
    return 0;
}
int cice_evp_hip_cgrid_strip_plan(const cice_evp_hip_dims *dims, int32_t ex, int32_t ey, int32_t lo0, int32_t slots, int32_t seg_min, int32_t seg,
                                  int32_t *n_items, int32_t *items6, int32_t items_cap, int32_t *n_windows, int32_t *tiles4, uint8_t *in_zone, int32_t windows_cap,
                                  int32_t *seg_rows)
{
    if (!dims || !n_items || !n_windows) return fail(-1, "bad argument");
    HaloPlan P;
    if (!build_halo_plan(*dims, P)) return fail(-3, "halo plan: %s", P.error.c_str());
    std::vector<int32_t> t4, tb, it;
    build_window_table(*dims, P, ex, ey, 1 << 20, t4, tb);
    // (ghost images: the sources of the rank's own ghost copies)
    std::vector<int> img((size_t)dims->nblocks * dims->nx_block * dims->ny_block, -1);
    for (size_t k = 0; k < P.local_src.size(); ++k)
        if (P.local_src[k] >= 0) img[(size_t)P.local_src[k]] = 0;        // (-1: a ghost cell filled with 0, no source)
    std::vector<StripZone> zones;
    strip_zones(*dims, t4, ex, ey, img.data(), zones);
    const int s = strip_items(zones, ex, ey, lo0, slots, seg_min, seg, it);
    std::vector<uint8_t> in;
    strip_windows(zones, t4, in);
    *n_items = (int32_t)(it.size() / 6);
    *n_windows = (int32_t)(t4.size() / 4);
    if (seg_rows) *seg_rows = s;
    if (items6) {
        if ((size_t)items_cap * 6 < it.size()) return fail(-1, "room for %d items, there are %d", items_cap, *n_items);
        std::copy(it.begin(), it.end(), items6);
    }
    if (tiles4 && in_zone) {
        if ((size_t)windows_cap * 4 < t4.size()) return fail(-1, "room for %d windows, there are %d", windows_cap, *n_windows);
        std::copy(t4.begin(), t4.end(), tiles4);
        std::copy(in.begin(), in.end(), in_zone);
    }
    return 0;
}
#endif  // CICE_EVP_HIP_TESTING

int cice_evp_hip_stream_probe(int64_t ncells, double *bytes_per_second)
{
    if (ncells <= 0 || !bytes_per_second) return fail(-1, "bad argument");
    const double sec = evp_stream_probe((size_t)ncells, 5, nullptr);
    if (sec <= 0) return fail((int)-sec, "stream probe failed: %s", hipGetErrorString((hipError_t)(int)-sec));
    *bytes_per_second = 46.0 * 8.0 * (double)ncells / sec;
    return 0;
}

// HIP device of this rank: CICE_EVP_HIP_DEVICE, else the rank's index on its node as the launcher states it
// (torch.distributed.run, Open MPI, MVAPICH2, MPICH / Intel MPI hydra, Slurm), else rank modulo the device count --
// right on one node, and on several when the launcher places ranks block-wise.
static int pick_device(int rank, int ndev)
{
    if (env("CICE_EVP_HIP_DEVICE")) return std::atoi(env("CICE_EVP_HIP_DEVICE"));
    for (const char *name : {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"})
        if (const char *v = std::getenv(name))
            if (*v >= '0' && *v <= '9') return std::atoi(v) % ndev;
    return rank % ndev;
}

int cice_evp_hip_finalize(void)
{
    if (S.stream) (void)hipStreamSynchronize(S.stream);
    cgrid_free();
    march_free();
    free_all();
    S = State();
    return 0;
}

int cice_evp_hip_init(const cice_evp_hip_dims *dims, const cice_evp_hip_params *params,
                      const double *HTE, const double *HTN, const double *dxT, const double *dyT,
                      const double *uarear, const double *tarea)
{
    if (!dims || !params) return fail(-1, "cice_evp_hip_init: null argument");
    // also after an init that failed midway (S.ready still false): release whatever it had created
    cice_evp_hip_finalize();
    if (dims->nghost != 1) return fail(-1, "nghost must be 1");
    if (dims->nblocks == 0 && dims->nranks > 1) {
        // no blocks on this rank (shared/ice_distribution.F90 hands a task nothing when the processor grid does not divide
        // the block grid): a bystander of the bootstrap, see State::bystander
        S.d = *dims;
        S.d.ilo = S.d.ihi = S.d.jlo = S.d.jhi = S.d.iglob0 = S.d.jglob0 = nullptr;
        S.d.gi0 = S.d.gj0 = S.d.gnx = S.d.gny = S.d.gowner = S.d.glocal = nullptr;
        S.d.nblocks_tot = 0;
        S.prm = *params;
        int ndev = 0;
        HIPC(hipGetDeviceCount(&ndev));
        if (ndev < 1) return fail(-4, "no HIP device");
        S.device = pick_device(dims->rank, ndev);
        HIPC(hipSetDevice(S.device));
        HIPC(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
        S.bystander = true;
        return 0;
    }
    if (!HTE || !HTN || !dxT || !dyT || !uarear || !tarea) return fail(-1, "cice_evp_hip_init: null argument");
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
    if (env_test("CICE_EVP_HIP_SELF_EXCHANGE") && std::atoi(env_test("CICE_EVP_HIP_SELF_EXCHANGE")) && dims->nranks == 1) {
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
            self.send_sign.push_back(S.plan.local_sign[k]);
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
        self.n_ghost_send = (int)self.send_src.size();
        self.n_ghost_recv = (int)self.recv_dst.size();
        S.plan.peers.push_back(self);
    }
    // the global block table is needed again by plans built later (the marching path decides at the first
    // cice_evp_hip_subcycle): keep a copy -- the caller's arrays need not outlive this call
    {
        const int32_t *src[6] = {dims->gi0, dims->gj0, dims->gnx, dims->gny, dims->gowner, dims->glocal};
        const bool have = dims->nblocks_tot > 0 && src[0] && src[1] && src[2] && src[3] && src[4] && src[5];
        for (int k = 0; k < 6; ++k) {
            S.gtab[k].clear();
            if (have) S.gtab[k].assign(src[k], src[k] + dims->nblocks_tot);
        }
        S.d.gi0 = have ? S.gtab[0].data() : nullptr; S.d.gj0 = have ? S.gtab[1].data() : nullptr;
        S.d.gnx = have ? S.gtab[2].data() : nullptr; S.d.gny = have ? S.gtab[3].data() : nullptr;
        S.d.gowner = have ? S.gtab[4].data() : nullptr; S.d.glocal = have ? S.gtab[5].data() : nullptr;
        if (!have) S.d.nblocks_tot = 0;
    }

    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(-4, "no HIP device");
    const int dev = pick_device(dims->rank, ndev);
    S.device = dev;
    HIPC(hipSetDevice(dev));
    HIPC(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    HIPC(hipStreamCreateWithFlags(&S.stream_comm, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&S.ev_pack, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&S.ev_halo, hipEventDisableTiming));
    S.overlap = !(env_test("CICE_EVP_HIP_NO_OVERLAP") && std::atoi(env_test("CICE_EVP_HIP_NO_OVERLAP")));
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
    if (env_test("CICE_EVP_HIP_TYB")) {
        const int t = std::atoi(env_test("CICE_EVP_HIP_TYB"));   // tile height [+100: XCD-contiguous order]
        S.tyb = (t % 100 >= 2 && t % 100 <= 9) ? t : 5 + 100 * (t / 100);
    }
    S.use_graph = !(env_test("CICE_EVP_HIP_NOGRAPH") && std::atoi(env_test("CICE_EVP_HIP_NOGRAPH")));

    for (auto &p : S.stat)
        if (alloc_d(&p, S.n)) return -1;
    for (int f = F_STRENGTH; f < F_COUNT; ++f) {
        if (f == F_UVEL || f == F_VVEL) continue;
        if (alloc_d(&S.in[f], S.n)) return -1;
    }
    S.nuv = S.n + (size_t)S.plan.tail;          // + staging slots for raw seam values of other ranks (tripole, any layout)
    for (int k = 0; k < 2; ++k) {
        if (alloc_d(&S.u[k], S.nuv) || alloc_d(&S.v[k], S.nuv)) return -1;
        for (auto &p : S.sig[k])
            if (alloc_d(&p, S.n)) return -1;
    }
    if (alloc_d(&S.hte, S.n) || alloc_d(&S.htn, S.n) || alloc_d(&S.vrelfac, S.n)) return -1;
    S.flags = EVP_F_VRELFAC;
    S.flags_allowed = ~0u;
    if (env_test("CICE_EVP_HIP_FLAGS")) S.flags_allowed = (unsigned)std::strtoul(env_test("CICE_EVP_HIP_FLAGS"), nullptr, 0);
    HIPC(hipMalloc((void **)&S.mask, S.n));
    HIPC(hipMemsetAsync(S.mask, 0, S.n, S.stream));
    HIPC(hipMalloc((void **)&S.blk, nb * sizeof(int4)));
    HIPC(hipMemcpy(S.blk, hb.data(), nb * sizeof(int4), hipMemcpyHostToDevice));
    if (upload_lists()) return -1;
    if (build_push_table()) return -1;
    if (S.push_ok) S.flags |= EVP_F_PUSH;
    if (derive_metrics(HTE, HTN, dxT, dyT, uarear, tarea)) return -1;
    S.hmask.resize(S.n);
    S.tyb_forced = env_test("CICE_EVP_HIP_TYB") != nullptr;
    S.ready = true;
    S.uploaded = false;
    S.cur = 0;
    S.fault_calls = 0;
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

static int upload_impl(const double *const *f, const int32_t *iceTmask, const int32_t *iceUmask, bool keep_sig)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!f || !iceTmask || !iceUmask) return fail(-1, "null argument");
    // (every argument check comes before the first copy is enqueued)
    if (!f[F_WATERX] || !f[F_WATERY] || !f[F_TBU]) return fail(-1, "null field (waterxU / wateryU / TbU)");
    HIPC(hipEventRecord(S.ev2, S.stream));
    CopyBatch B;
    if (!keep_sig) {
        S.cur = 0;
        for (int k = 0; k < 12; ++k) {
            if (!f[k]) return fail(-1, "null stress field %d", k);
            B.items.push_back({S.sig[0][k], f[k]});
        }
    }
    const int cur = S.cur;      // keep_sig: the stresses of the previous call live in sig[cur]
    for (int fi = F_STRENGTH; fi < F_COUNT; ++fi) {
        if (fi == F_UVEL || fi == F_VVEL) continue;
        if ((fi == F_UVEL_INIT || fi == F_VVEL_INIT) && (!f[fi] || S.prm.revp == 0.0)) continue;   // only read when revp = 1
        if (fi == F_WATERX || fi == F_WATERY || fi == F_TBU) continue;   // below: only where the kernels will read them
        // lean call (cice_evp_hip_run on page-locked arrays): the loop writes these four on ice U-cells only and the
        // download writes back those cells only, so the caller's values elsewhere never need to travel
        if (S.lean_diag && (fi == F_STRINTX || fi == F_STRINTY || fi == F_TAUBX || fi == F_TAUBY)) continue;
        // (strintx/y, taubx/y are outputs of the loop on ice U-cells only; elsewhere the caller's values -- zeroed by
        // dyn_prep2 -- must survive the download, so they travel in as well)
        if (!f[fi]) return fail(-1, "null field %d", fi);
        B.items.push_back({S.in[fi], f[fi]});
    }
    if (S.prm.revp != 0.0 && (!f[F_UVEL_INIT] || !f[F_VVEL_INIT]))
        return fail(-1, "uvel_init/vvel_init required for revised EVP");
    if (!f[F_UVEL] || !f[F_VVEL]) return fail(-1, "null velocity field");
    B.items.push_back({S.u[cur], f[F_UVEL]});
    B.items.push_back({S.v[cur], f[F_VVEL]});
    if (h2d_batch(B)) return -1;
    {   // the second ping-pong copy of what was uploaded: one launch
        EvpCopyTab T{};
        T.len = S.n;
        T.vec2 = 1;
        if (!keep_sig)
            for (int k = 0; k < 12; ++k) { T.src[T.n] = S.sig[0][k]; T.dst[T.n] = S.sig[1][k]; ++T.n; }
        T.src[T.n] = S.u[cur]; T.dst[T.n] = S.u[cur ^ 1]; ++T.n;
        T.src[T.n] = S.v[cur]; T.dst[T.n] = S.v[cur ^ 1]; ++T.n;
        for (int k = 0; k < T.n; ++k)
            if ((((uintptr_t)T.src[k]) | ((uintptr_t)T.dst[k])) & 15u) T.vec2 = 0;
        evp_launch_copy_many(T, S.stream);
    }
    // While that batch travels the host scans the caller's arrays for what the kernels will not read in this call:
    // waterx / watery where they equal uocn / vocn bit for bit on every ice U-cell, TbU where it is zero there (with
    // uvel_init / vvel_init under classic EVP: five of the 32 arrays of the default configuration stay on the host).
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
        CopyBatch B2;
        if (!(water_is_ocn && (S.flags_allowed & EVP_F_WATER_IS_OCN))) {
            B2.items.push_back({S.in[F_WATERX], wx});
            B2.items.push_back({S.in[F_WATERY], wy});
        }
        if (!(tbu_zero && (S.flags_allowed & EVP_F_TBU_ZERO))) B2.items.push_back({S.in[F_TBU], tb});
        if (!B2.items.empty() && h2d_batch(B2)) return -1;
    }
    if (S.lean_diag) {
        // bit 7: the cells the loop really writes strintx/y, taubx/y on -- iceUmask on INTERIOR cells.  A caller may
        // hold iceUmask on ghost or padding cells too (halo-updated masks): the masked scatter of the download must
        // not hand never-written device values back there.  Kernels test bits 0 and 1 only.
        const int nxb = S.d.nx_block;
        for (int b = 0; b < S.d.nblocks; ++b)
            for (int j = S.d.jlo[b]; j <= S.d.jhi[b]; ++j)
                for (int i = S.d.ilo[b]; i <= S.d.ihi[b]; ++i) {
                    const size_t k = (size_t)b * S.plane + (size_t)(j - 1) * nxb + (i - 1);
                    if (S.hmask[k] & 2u) S.hmask[k] |= 0x80u;
                }
    }
    S.flags &= ~(EVP_F_WATER_IS_OCN | EVP_F_TBU_ZERO);
    if (water_is_ocn) S.flags |= EVP_F_WATER_IS_OCN;
    if (tbu_zero) S.flags |= EVP_F_TBU_ZERO;
    evp_launch_vrelfac(S.in[F_AIX], S.in[F_CW], S.prm.rhow, S.vrelfac, S.n, S.stream);
    HIPC(hipMemcpyAsync(S.mask, S.hmask.data(), S.n, hipMemcpyHostToDevice, S.stream));
    if (keep_sig) evp_launch_zero_sig_off_mask(S.sig[0], S.sig[1], S.mask, S.n, S.stream);
    HIPC(hipEventRecord(S.ev3, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
    S.t_h2d_ms = ms;
    S.uploaded = true;
    ++S.upload_seq;
    if (S.hmask_prev != S.hmask) { S.res2_order_stale = true; S.hmask_prev = S.hmask; }
    return tune_after_upload();
}

int cice_evp_hip_upload(const double *const *f, const int32_t *iceTmask, const int32_t *iceUmask)
{
    S.sig_valid = false;
    return upload_impl(f, iceTmask, iceUmask, false);
}

int cice_evp_hip_subcycle(int32_t ndte)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (ndte < 0) return fail(-1, "ndte < 0");
    if (ndte == 0) return 0;
    HIPC(hipEventRecord(S.ev0, S.stream));
    // (the tiles that hold ice must all be on the chip at once: checked against THIS call's masks -- a call with more ice than that
    // goes through the kernels below, the next one is asked again)
    S.res_ran = S.res_mode == 1 && resident2_fits_now();
    if (S.res_ran) {
        if (int rc = launch_resident2(ndte, S.cur, false)) return rc;
        S.res_launched = true;
        HIPC(hipEventRecord(S.ev1, S.stream));
        S.cur ^= (ndte & 1);
        S.t_nsub = ndte;
        return 0;
    }
    S.march.last_call = false;
    if (march_wanted()) {
        // large per-rank domain: several subcycles per pass over HBM (evp_march.hip); falls back to the loop below by
        // itself when the uploaded state does not qualify
        const int declined0 = S.march.declined;
        if (int rc = march_run(ndte)) return rc;
        S.march.last_call = S.march.declined == declined0;
        HIPC(hipEventRecord(S.ev1, S.stream));
        S.t_nsub = ndte;
        return 0;
    }
    // RCCL p2p inside a captured graph: opt-in (CICE_EVP_HIP_GRAPH_RCCL=1) until measured on a multi-GPU node
    const bool graph_rccl = env_test("CICE_EVP_HIP_GRAPH_RCCL") && std::atoi(env_test("CICE_EVP_HIP_GRAPH_RCCL"));
    const bool graph_ok = S.use_graph && (S.plan.peers.empty() || graph_rccl || S.direct.on);
    if (graph_ok) {
        const auto key = std::make_tuple((int)ndte, S.cur, S.flags & S.flags_allowed);
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

// 1 if cice_evp_hip_stress_halo can do its job on this rank layout (tripole: always; tripoleT: the top row on one rank and no
// eliminated block in it; other boundaries: there is nothing to do, 1), else 0 -- for a host that wants to keep the stresses
// on the device between calls and must know whether evp()'s twelve ice_HaloUpdate_stress calls can be left to the library.
int cice_evp_hip_stress_halo_available(void)
{
    if (!S.ready) return 0;
    return (S.plan.tfold && S.plan.stress_remote) ? 0 : 1;
}

// Tripole: force the stresses symmetric across the seam on the resident state, as evp() does
// on the host arrays after the subcycle loop (12 x ice_HaloUpdate_stress, ice_dyn_evp.F90:1321-1389).
int cice_evp_hip_stress_halo(void)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (S.plan.tfold) {
        if (S.plan.stress_remote)
            return fail(-9, "tripoleT: with the top row split over ranks (or next to an eliminated block) the stress symmetrisation "
                            "stays with the host (evp() applies it to its own arrays, ice_dyn_evp.F90:1321-1389)");
        evp_launch_halo_stress_tfold(S.sig[S.cur], S.h_stress_dst, S.h_stress_src, S.n_stress, S.h_stress_own_dst, S.h_stress_own_src,
                                     S.n_stress_own, S.stream);
        // the north-west corner ghost cells: both arrays of a pair, each from the other's row NY-1 (which nothing above writes)
        evp_launch_halo_stress(S.sig[S.cur], S.h_stress_corner_dst, S.h_stress_corner_src, S.n_stress_corner, S.stream);
        HIPC(hipGetLastError());
        return 0;
    }
    evp_launch_halo_stress(S.sig[S.cur], S.h_stress_dst, S.h_stress_src, S.n_stress, S.stream);
    // partners on other ranks (fold row split in x): a1's ghost row <- a2's top row through the exchange of a shifted copy
    // (halo_plan.h); collective -- a rank without destinations of its own still serves its top row.  Scalars: factor -1.
    if (S.plan.fold_split)
        for (int fam = 0; fam < 12; fam += 4)
            for (int q = fam; q < fam + 2; ++q) {
                double *a1 = S.sig[S.cur][q], *a2 = S.sig[S.cur][q ^ 2];
                if (int rc = fold_remote_pair(a2, a1, a1, a2, 1, -1.0, -1.0)) return rc;
            }
    HIPC(hipGetLastError());
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
    if (it != S.pinned.end() && it->second.bytes >= (size_t)bytes) return 0;
    if (it != S.pinned.end()) {
        (void)hipHostUnregister(const_cast<void *>(ptr));
        S.pinned.erase(it);
    }
    // mapped: the device can address the range itself, so the per-call traffic becomes one gather and one
    // scatter launch (evp_copy.hip) instead of one copy per array
    void *dev = nullptr;
    if (hipHostRegister(const_cast<void *>(ptr), (size_t)bytes, hipHostRegisterMapped) == hipSuccess) {
        if (hipHostGetDevicePointer(&dev, const_cast<void *>(ptr), 0) != hipSuccess) dev = nullptr;
    } else {
        (void)hipGetLastError();
        HIPC(hipHostRegister(const_cast<void *>(ptr), (size_t)bytes, hipHostRegisterDefault));
    }
    S.pinned[ptr] = State::Pinned{(size_t)bytes, dev};
    return 0;
}

// Options of the per-call entry points.  CICE_EVP_HIP_OPT_STRESS_RESIDENT (1): the 12 stress components
// stay on the device between calls of cice_evp_hip_run -- evp() is their only writer (ice_dyn_evp.F90), so
// they are uploaded on the first call only (and after cice_evp_hip_invalidate_stresses), zeroed off
// iceTmask on the device as dyn_prep2 does on the host, and not downloaded; the caller fetches them when
// something else needs them (restart / history: cice_evp_hip_fetch_stresses).
int cice_evp_hip_set_option(int32_t key, int32_t value)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (key == CICE_EVP_HIP_OPT_STRESS_RESIDENT) {
        S.opt_sig_resident = value != 0;
        if (!S.opt_sig_resident) S.sig_valid = false;
        return 0;
    }
    return fail(-1, "unknown option %d", (int)key);
}

int cice_evp_hip_invalidate_stresses(void)
{
    S.sig_valid = false;
    return 0;
}

int cice_evp_hip_fetch_stresses(double *const *sig12)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (!sig12) return fail(-1, "null argument");
    HIPC(hipStreamSynchronize(S.stream));
    if (int rc = resident_check_error()) return rc;
    CopyBatch B;
    for (int k = 0; k < 12; ++k)
        if (sig12[k]) B.items.push_back({sig12[k], S.sig[S.cur][k]});
    if (d2h_batch(B)) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

int cice_evp_hip_sync(void)
{
    if (!S.ready) return fail(-1, "not initialised");
    HIPC(hipStreamSynchronize(S.stream));
    if (int rc = direct_check_error()) return rc;
    if (int rc = march_direct_error()) return rc;
    return resident_check_error();
}

int cice_evp_hip_download(double *const *f)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    HIPC(hipStreamSynchronize(S.stream));
    if (int rc = direct_check_error()) return rc;
    if (int rc = march_direct_error()) return rc;
    if (int rc = resident_check_error()) return rc;
    HIPC(hipEventRecord(S.ev2, S.stream));
    CopyBatch B;
    for (int k = 0; k < 12; ++k)
        if (f[k]) B.items.push_back({f[k], S.sig[S.cur][k]});
    const int outs[4] = {F_STRINTX, F_STRINTY, F_TAUBX, F_TAUBY};
    if (S.lean_diag) {
        CopyBatch M;
        for (int o : outs)
            if (f[o]) M.items.push_back({f[o], S.in[o]});
        if (d2h_batch_masked(M, 0x80u)) return -1;          // bit7: interior && iceUmask = the cells the loop wrote
    } else {
        for (int o : outs)
            if (f[o]) B.items.push_back({f[o], S.in[o]});
    }
    if (f[F_UVEL]) B.items.push_back({f[F_UVEL], S.u[S.cur]});
    if (f[F_VVEL]) B.items.push_back({f[F_VVEL], S.v[S.cur]});
    if (d2h_batch(B)) return -1;
    HIPC(hipEventRecord(S.ev3, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
    S.t_d2h_ms = ms;
    if (S.t_nsub > 0 && hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess) S.t_loop_ms = ms;
    return 0;
}

// Test hook: the N-th call that reaches it reports a failure that did not happen (N = value of the variable), so
// that recovery paths can be exercised on a healthy GPU.
static bool fault_hook(const char *name)
{
    const char *e = env_test(name);
    return e && ++S.fault_calls == std::atoi(e);      // counted from cice_evp_hip_init
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
    // stresses that never left the device (CICE_EVP_HIP_OPT_STRESS_RESIDENT): no upload, dyn_prep2's zeroing on the device
    const bool keep_sig = S.opt_sig_resident && S.sig_valid && S.uploaded;
    {
        CopyBatch D;
        for (double *p : {strintxU, strintyU, taubxU, taubyU})
            if (p) D.items.push_back({p, nullptr});
        S.lean_diag = D.items.size() == 4 && batch_mapped(D, false) &&
                      !(env_test("CICE_EVP_HIP_LEAN") && !std::atoi(env_test("CICE_EVP_HIP_LEAN")));
    }
    struct LeanOff { ~LeanOff() { S.lean_diag = false; } } lean_off;      // only for the duration of this call
    if (int rc = upload_impl(f, iceTmask, iceUmask, keep_sig)) return rc;
    // The resident kernel assumes the GPU to itself.  If that did not hold (another process or a long kernel on the
    // device: a wait gave up) the call is repeated with the streaming kernel, which is then kept.  Without resident
    // stresses nothing has been written back yet and the caller's inputs are intact.  With them (keep_sig) the only
    // copy of the pre-call stresses is sig[cur], which the resident kernel overwrites at its end (both ping-pong
    // copies): keep a device snapshot for the replay.
    const bool may_replay = S.res_mode == 1 && S.plan.peers.empty();
    const int cur0 = S.cur;
    if (may_replay && keep_sig) {
        EvpCopyTab T{};
        T.len = S.n;
        T.vec2 = 1;
        for (int k = 0; k < 12; ++k) {
            if (!S.sig_snap[k]) HIPC(hipMalloc((void **)&S.sig_snap[k], S.n * sizeof(double)));
            T.src[T.n] = S.sig[cur0][k]; T.dst[T.n] = S.sig_snap[k]; ++T.n;
            if ((((uintptr_t)T.src[k]) | ((uintptr_t)T.dst[k])) & 15u) T.vec2 = 0;
        }
        evp_launch_copy_many(T, S.stream);
    }
    S.sig_valid = false;                      // until this call has succeeded
    if (int rc = cice_evp_hip_subcycle(ndte)) return rc;
    if (may_replay) {
        HIPC(hipStreamSynchronize(S.stream));
        if (resident_check_error() != 0 || fault_hook("CICE_EVP_HIP_FAULT_REPLAY")) {
            if (env("CICE_EVP_HIP_VERBOSE"))
                std::fprintf(stderr, "[cice_evp_hip] %s -- repeating the call with the streaming kernel\n", g_err.c_str());
            g_err.clear();
            ++S.res_fallbacks;
            S.res_mode = 0;
            if (keep_sig) {
                EvpCopyTab T{};
                T.len = S.n;
                T.vec2 = 1;
                for (int k = 0; k < 12; ++k) {
                    T.src[T.n] = S.sig_snap[k]; T.dst[T.n] = S.sig[cur0][k]; ++T.n;
                    if ((((uintptr_t)T.src[k]) | ((uintptr_t)T.dst[k])) & 15u) T.vec2 = 0;
                }
                evp_launch_copy_many(T, S.stream);
                S.cur = cur0;
            }
            if (int rc = upload_impl(f, iceTmask, iceUmask, keep_sig)) return rc;
            if (int rc = cice_evp_hip_subcycle(ndte)) return rc;
            // the resident kernel is off from here on (res_mode = 0): the snapshot has no further use
            HIPC(hipStreamSynchronize(S.stream));
            for (auto &q : S.sig_snap) { if (q) (void)hipFree(q); q = nullptr; }
        }
    }
    // only the documented outputs travel back
    double *o[F_COUNT] = {};
    if (!S.opt_sig_resident)
        for (int k = 0; k < 12; ++k) o[k] = f[k];
    o[F_STRINTX] = strintxU; o[F_STRINTY] = strintyU; o[F_TAUBX] = taubxU; o[F_TAUBY] = taubyU;
    o[F_UVEL] = uvel; o[F_VVEL] = vvel;
    if (int rc = cice_evp_hip_download(o)) return rc;          // incl. the error words of the kernels
    S.sig_valid = S.opt_sig_resident;                          // only a call that succeeded leaves valid resident stresses
    return 0;
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
    const bool res = S.res_mode == 1 && S.res_ran;
    const double v[16] = {S.t_loop_ms, S.t_h2d_ms, S.t_d2h_ms, (double)S.t_nsub,
                         res ? 1.0 / std::max(S.t_nsub, 1) : S.march.last_call ? (double)S.march.call_passes / std::max(S.march.call_subcycles, 1) :
                         1.0 + ((S.n_local > 0 && !(S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH))) ? 1.0 : 0.0) +
                             (S.plan.peers.empty() ? 0.0 : (S.direct.on ? (use_riding_exchange() ? 0.0 : 1.0) : 2.0)) + ((S.n_seam + S.n_pole + S.n_late) > 0 ? 1.0 : 0.0),
                         (double)(res ? 2000 + S.res2_logw : (S.march.last_call ? 3000 + S.march.seglen : S.tyb)), marks_ms, S.t_stream_probe_ms, S.t_res_probe_ms,
                          S.plan.peers.empty() ? 0.0 : (S.direct.on ? 2.0 : 1.0), S.prep.t_ms, (double)S.res_fallbacks,
                          (double)(S.msk.on ? S.msk.n_send : S.n_send), (double)(S.msk.on ? S.msk.n_recv : S.n_recv),
                          (double)(res ? (S.res2_nlive > 0 ? S.res2_nlive : S.res2_ntiles) : 0), (double)(S.res_mode == 1 ? S.res2_ntiles : 0)};
    for (int k = 0; k < n && k < 16; ++k) out[k] = v[k];
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

#ifdef CICE_EVP_HIP_TESTING
// Per-CU record of the last resident launch (16 x 16 tiles): 2048 CUs x {lock, stamp, ice-holding waves
// on SIMD 0..3, 0, 0}; for tools that check how evenly the workgroups spread their waves.
int cice_evp_hip_debug_cuload(int32_t *out, int32_t n)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!S.res2_cuload) return fail(-1, "no resident launch with 16 x 16 tiles yet");
    if (!out || n < 0 || n > 2048 * 8) return fail(-1, "bad argument");
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemcpy(out, S.res2_cuload, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

// Phase stamps of the last resident launch made with CICE_EVP_HIP_RES_PROF=1: [tiles][4 chunks][8] x u64.
int cice_evp_hip_debug_prof(uint64_t *out, int32_t ntiles_max)
{
    if (!S.ready || !S.res2_prof) return fail(-1, "no profiled resident launch (CICE_EVP_HIP_RES_PROF=1, 16 x 16 tiles)");
    if (!out || ntiles_max < S.res2_ntiles) return fail(-1, "bad argument: need room for %d tiles", S.res2_ntiles);
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemcpy(out, S.res2_prof, (size_t)S.res2_ntiles * 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return S.res2_ntiles > 0 ? 0 : -1;
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

// The lists the on-chip kernel uses when the tripole fold row is split over ranks (halo_plan.h): per peer (ascending rank,
// as cice_evp_hip_halo_plan) counts4 = {ghost entries at the head of the send list, of the recv list, seam images out, in};
// send_sign in send-list order; the seam images out as (src, dst at the peer, sign), in as (dst, global column, sign).
int cice_evp_hip_fold_images_plan(int32_t *counts4, int32_t *send_sign, int32_t *out3, int32_t *in3)
{
    const HaloPlan &P = S.plan;
    size_t q = 0, so = 0, oo = 0, io = 0;
    for (const HaloPeer &p : P.peers) {
        if (counts4) {
            counts4[4 * q] = p.n_ghost_send; counts4[4 * q + 1] = p.n_ghost_recv;
            counts4[4 * q + 2] = (int32_t)p.fimg_src.size(); counts4[4 * q + 3] = (int32_t)p.fimg_recv_dst.size();
        }
        for (size_t k = 0; k < p.send_sign.size(); ++k)
            if (send_sign) send_sign[so + k] = p.send_sign[k];
        for (size_t k = 0; k < p.fimg_src.size(); ++k)
            if (out3) { out3[3 * (oo + k)] = p.fimg_src[k]; out3[3 * (oo + k) + 1] = p.fimg_dst[k]; out3[3 * (oo + k) + 2] = p.fimg_sign[k]; }
        for (size_t k = 0; k < p.fimg_recv_dst.size(); ++k)
            if (in3) { in3[3 * (io + k)] = p.fimg_recv_dst[k]; in3[3 * (io + k) + 1] = p.fimg_recv_col[k]; in3[3 * (io + k) + 2] = p.fimg_recv_sign[k]; }
        so += p.send_sign.size(); oo += p.fimg_src.size(); io += p.fimg_recv_dst.size();
        ++q;
    }
    return 0;
}

// recv_sign of cice_evp_hip_halo_plan's recv list (-1: the ghost lies across the tripole fold), same order
int cice_evp_hip_peer_signs(int32_t *recv_sign)
{
    size_t ro = 0;
    for (const HaloPeer &p : S.plan.peers) {
        for (size_t k = 0; k < p.recv_sign.size(); ++k)
            if (recv_sign) recv_sign[ro + k] = p.recv_sign[k];
        ro += p.recv_sign.size();
    }
    return 0;
}

int cice_evp_hip_center_plan(int32_t *count, int32_t *dst, int32_t *src, int32_t *vsign)
{
    const HaloPlan &P = S.plan;
    if (count) *count = (int32_t)P.center_dst.size();
    for (size_t k = 0; k < P.center_dst.size(); ++k) {
        if (dst) dst[k] = P.center_dst[k];
        if (src) src[k] = P.center_src[k];
        if (vsign) vsign[k] = P.center_vsign[k];
    }
    return P.center_remote ? 1 : 0;
}

// Lists of the shifted-copy exchange (halo_plan.h): which 0 = this rank's cells of row NY-1 where the copy is built,
// 1 = centre-field ghost cells filled from it, 2 = ghost-row cells of the stress symmetrisation filled from it; 3 / 4 =
// east-west ghost cells of row NY and the staging slots that hold their values after a plain exchange.
// Returns 1 when the blocks holding row NY have more than one owner (the exchanges are collective then), else 0.
int cice_evp_hip_fold_split_plan(int32_t which, int32_t *count, int32_t *cells)
{
    const HaloPlan &P = S.plan;
    // (5 / 6: tripoleT -- east-west ghost cells of the top row that the stress symmetrisation leaves as images of their own array,
    // and the cells they mirror)
    const std::vector<int32_t> &v = which == 0 ? P.fold_shift_cells : (which == 1 ? P.center_foldr_dst :
                                    (which == 2 ? P.stress_foldr_dst : (which == 3 ? P.center_seam_dst :
                                    (which == 4 ? P.center_seam_slot : (which == 5 ? P.stress_own_dst : (which == 6 ? P.stress_own_src :
                                    (which == 7 ? P.stress_corner_dst : P.stress_corner_src)))))));
    if (count) *count = (int32_t)v.size();
    if (cells)
        for (size_t k = 0; k < v.size(); ++k) cells[k] = v[k];
    return P.fold_split ? 1 : 0;
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
#endif  // CICE_EVP_HIP_TESTING

// General form of the seam step (any rank layout): counts2 = {entries, staging slots}; lists may be NULL.
int cice_evp_hip_seam_fin_plan(int32_t *counts2, int32_t *dst, int32_t *a, int32_t *b, int32_t *coef)
{
    const HaloPlan &P = S.plan;
    if (counts2) { counts2[0] = (int32_t)P.fin_dst.size(); counts2[1] = (int32_t)P.tail; }
    for (size_t k = 0; k < P.fin_dst.size(); ++k) {
        if (dst) dst[k] = P.fin_dst[k];
        if (a) a[k] = P.fin_a[k];
        if (b) b[k] = P.fin_b[k];
        if (coef) coef[k] = P.fin_coef[k];
    }
    return P.stress_remote ? 1 : 0;
}

int cice_evp_hip_march_info(int32_t *out, int32_t n)
{
    const State::March &M = S.march;
    const int32_t v[10] = {M.mode, (int32_t)std::min<long>(M.passes, 0x7fffffffL), M.declined, M.nstrips, M.nseg, M.seglen,
                           M.last_call ? 1 : 0, M.direct, M.kpass, (int32_t)std::min<long>(M.subcycles, 0x7fffffffL)};
    for (int k = 0; out && k < n && k < 10; ++k) out[k] = v[k];
    return 0;
}

// One line on what the last cice_evp_hip_subcycle ran and why the alternatives did not -- for logs (the Fortran shim
// prints it once after the first call when CICE_EVP_HIP_VERBOSE is set) and for reading a first multi-GPU run.
int cice_evp_hip_describe_path(char *buf, int32_t n)
{
    if (!buf || n < 2) return fail(-1, "describe_path: no buffer");
    const State::March &M = S.march;
    const char *kernel = S.res_mode == 1 ? (S.res_remote ? "on-chip resident (tagged records, neighbours on other ranks)"
                                                       : "on-chip resident (tagged records)")
                         : (M.last_call ? (M.kpass == 4 ? "four subcycles per pass (marching)" : M.kpass == 3 ? "three subcycles per pass (marching)" : "two subcycles per pass (marching)")
                                      : "one subcycle per launch (streaming)");
    const char *transport = S.plan.peers.empty() ? "none (one rank)" : (S.direct.on ? "mailbox over HIP IPC" : (S.have_comm ? "RCCL send/recv" : "not set up"));
    const char *ring = (M.mode == 1 && !S.plan.peers.empty())
                           ? (M.direct == 1 ? " (ring between ranks: stores into HIP-IPC-mapped inboxes)"
                                            : (M.direct == 2 ? " (ring between ranks: RCCL send/recv, direct stores on trial)" : " (ring between ranks: RCCL send/recv)"))
                           : "";
    std::snprintf(buf, (size_t)n, "rank %d of %d: kernel = %s; halo transport = %s%s%s; marching path: %s%s%s%s; blocks %d, cells per exchange %d",
                  (int)S.d.rank, (int)std::max(1, (int)S.d.nranks), kernel, transport,
                  (!S.plan.peers.empty() && !S.direct.on && !S.direct.why.empty()) ? " (mailbox off: " : "",
                  (!S.plan.peers.empty() && !S.direct.on && !S.direct.why.empty()) ? (S.direct.why + ")").c_str() : "",
                  M.mode == 1 ? "on" : (M.mode == 0 ? "off" : "undecided"), (M.mode == 0 && !M.why.empty()) ? " -- " : "",
                  (M.mode == 0 && !M.why.empty()) ? M.why.c_str() : "", ring, (int)S.d.nblocks, (int)(S.msk.on ? S.msk.n_send : S.n_send));
    return 0;
}

#ifdef CICE_EVP_HIP_TESTING
int cice_evp_hip_plan_flags(int32_t *flags, int32_t n)
{
    const HaloPlan &P = S.plan;
    const int32_t v[5] = {P.any_fold_exchange, P.fold_rows, P.stress_remote, P.center_remote, P.center_fold_remote};
    int32_t k = 0;
    for (; flags && k < n && k < 5; ++k) flags[k] = v[k];
    return k;
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
#endif  // CICE_EVP_HIP_TESTING

}  // extern "C"
