// Host side of the preparation phase of evp() on the device (kernels: evp_prep.hip).
#include "evp_host.h"

using namespace evp_host;

extern "C" {

// ---- next tier: the preparation phase of evp() on the device ----------------
int cice_evp_hip_set_prep_geometry(const int32_t *tmask, const int32_t *umask, const double *hm,
                                   const double *tarea, const double *uarea, const double *fcor_blk)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!tmask || !umask || !hm || !tarea || !uarea || !fcor_blk) return fail(-1, "null argument");
    State::Prep &Q = S.prep;
    // T-grid ghost cells owned by other ranks travel with the velocity exchange (same cells for centre and corner
    // fields) -- except across the tripole fold, where centre fields mirror other cells than corner fields do
    // (tripoleT: the centre rule rewrites the top physical row -- a fold step of its own after the plain ghost copies, one rank)
    // (ranks cut in y only are fine: what travels between them are plain ghost rows, the fold step stays on the top rank)
    if (S.plan.tfold && S.plan.center_tf_remote)
        return fail(-9, "device preparation on a tripoleT grid: the top row's mirror cells live on other ranks (or in an eliminated "
                        "block) here; keep evp()'s host preparation (cice_evp_hip_run)");
    auto B = [&](uint8_t *&p) -> int { if (!p) HIPC(hipMalloc((void **)&p, S.n)); return 0; };
    if (B(Q.tmask) || B(Q.umask) || B(Q.umask_old) || B(Q.tmphm)) return -1;
    HIPC(hipMalloc((void **)&Q.umask_old32, S.n * sizeof(int32_t)));
    // (with the staging tail of a split tripole seam row: these arrays go through the velocity exchange too)
    auto D = [&](double *&p) -> int { return p ? 0 : alloc_d(&p, S.nuv); };
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
    Q.n_tf = (int)P.center_tf_dst.size();
    if (Q.n_tf && !Q.tf_dst) {
        const size_t nb = (size_t)Q.n_tf * sizeof(int32_t);
        HIPC(hipMalloc((void **)&Q.tf_dst, nb));
        HIPC(hipMalloc((void **)&Q.tf_a, nb));
        HIPC(hipMalloc((void **)&Q.tf_b, nb));
        HIPC(hipMalloc((void **)&Q.tf_flip, (size_t)Q.n_tf));
        HIPC(hipMalloc((void **)&Q.tf_tmp, (size_t)4 * Q.n_tf * sizeof(double)));
        HIPC(hipMemcpy(Q.tf_dst, P.center_tf_dst.data(), nb, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Q.tf_a, P.center_tf_a.data(), nb, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Q.tf_b, P.center_tf_b.data(), nb, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Q.tf_flip, P.center_tf_flip.data(), (size_t)Q.n_tf, hipMemcpyHostToDevice));
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
    // stresses: all 12 given, or all 12 NULL = keep the copy the previous call left on the device
    // (nothing between two evp() calls touches them: ice_dyn_evp.F90 is their only writer)
    int nsig = 0;
    for (int k = 0; k < 12; ++k) nsig += fields32[k] != nullptr;
    if (nsig != 0 && nsig != 12) return fail(-1, "stress fields: give all 12 or none");
    if (nsig == 0 && !S.uploaded) return fail(-1, "no stresses on the device yet: the first call must upload them");
    if (!fields32[F_UVEL] || !fields32[F_VVEL]) return fail(-1, "null velocity field");
    HIPC(hipEventRecord(S.ev2, S.stream));
    // everything that travels in, as ONE gather launch over the arrays the caller page-locked (cice_evp_hip_pin_host)
    CopyBatch B;
    for (int k = 0; k < 11; ++k) B.items.push_back({Q.t[k], tfields11[k]});
    for (int k = 0; k < 12; ++k) {
        if (nsig) {
            B.items.push_back({S.sig[0][k], fields32[k]});
        } else if (S.cur != 0) {
            HIPC(hipMemcpyAsync(S.sig[0][k], S.sig[S.cur][k], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
        }
    }
    S.cur = 0;
    B.items.push_back({S.u[0], fields32[F_UVEL]});
    B.items.push_back({S.v[0], fields32[F_VVEL]});
    bool tbu_zero = true;
    if (fields32[F_TBU]) {
        B.items.push_back({S.in[F_TBU], fields32[F_TBU]});
        for (size_t k = 0; k < S.n && tbu_zero; ++k) tbu_zero = fields32[F_TBU][k] == 0.0;
    } else {
        HIPC(hipMemsetAsync(S.in[F_TBU], 0, S.n * sizeof(double), S.stream));
    }
    if (h2d_batch(B)) return -1;
    // the previous call's iceUmask: the caller's 32-bit logical words, reduced to bytes on the device
    HIPC(hipMemcpyAsync(Q.umask_old32, iceUmask, S.n * sizeof(int32_t), hipMemcpyHostToDevice, S.stream));
    evp_launch_words_to_bytes(Q.umask_old32, Q.umask_old, S.n, S.stream);
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
    P.strintx = S.in[F_STRINTX]; P.strinty = S.in[F_STRINTY]; P.taubx = S.in[F_TAUBX]; P.tauby = S.in[F_TAUBY];
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
    // and :466-469 (calc_strair branch): strairxT, strairyT -- one launch for all ten
    halo({{Q.maskd, false}, {Q.tmass, false}, {Q.t[3], false}, {Q.t[4], false},
          {Q.t[5], true}, {Q.t[6], true}, {Q.t[7], true}, {Q.t[8], true}, {Q.t[9], true}, {Q.t[10], true}});
    if (S.plan.tfold && Q.n_tf) {
        // tripoleT: rows NY (on the fold: symmetrised, rewritten from its mirror) and NY+1 of the same ten fields, after the
        // plain ghost copies (whose sources are rows the fold step does not write); the two-pass fold launch of the C grid
        // (evp_cgrid.hip: cg_fold_reg / cg_fold_one), four fields at a time
        const std::pair<double *, bool> arrs[10] = {{Q.maskd, false}, {Q.tmass, false}, {Q.t[3], false}, {Q.t[4], false}, {Q.t[5], true},
                                                    {Q.t[6], true}, {Q.t[7], true}, {Q.t[8], true}, {Q.t[9], true}, {Q.t[10], true}};
        for (int k0 = 0; k0 < 10; k0 += 4) {
            EvpCgFold F{};
            for (int k = k0; k < std::min(k0 + 4, 10); ++k) {
                F.x[F.nfields] = arrs[k].first;
                F.loc[F.nfields] = 0;
                F.isign[F.nfields] = arrs[k].second ? -1.0 : 1.0;
                ++F.nfields;
            }
            F.L[0] = {Q.tf_dst, Q.tf_a, Q.tf_b, Q.tf_flip, Q.n_tf};
            F.tmp = Q.tf_tmp;
            F.maxn = Q.n_tf;
            evp_launch_cgrid_fold(F, S.stream);
        }
    }
    if (S.plan.center_remote) {
        // neighbours on other ranks (no tripole fold here: centre and corner fields mirror the same
        // cells, so the velocity exchange carries pairs of T-grid fields)
        double *pairs[5][2] = {{Q.maskd, Q.tmass}, {Q.t[3], Q.t[4]}, {Q.t[5], Q.t[6]}, {Q.t[7], Q.t[8]}, {Q.t[9], Q.t[10]}};
        for (auto &pr : pairs) {
            if (int rc = halo_remote_pair(pr[0], pr[1], false, true)) return rc;
            if (S.plan.fold_split)       // east-west ghost cells of row NY owned elsewhere: the raw values in the staging slots
                if (int rc = fold_seam_ghosts(pr[0], pr[1])) return rc;
        }
        if (S.plan.fold_split) {
            // the plain exchange follows the NE-corner rule: ghost cells across the fold whose CENTRE-rule source is on this
            // rank have just been overwritten with a corner-rule value from elsewhere -- the local list again (its sources
            // are interior cells)
            halo({{Q.maskd, false}, {Q.tmass, false}, {Q.t[3], false}, {Q.t[4], false},
                  {Q.t[5], true}, {Q.t[6], true}, {Q.t[7], true}, {Q.t[8], true}, {Q.t[9], true}, {Q.t[10], true}});
            // ... and where the fold row is split in x, the ghost cells whose centre-rule source lies across the fold on
            // another rank: the exchange of a shifted copy (halo_plan.h); scalars -1 (undo the exchange's sign), vectors +1
            const double fac[5][2] = {{-1, -1}, {-1, -1}, {1, 1}, {1, 1}, {1, 1}};
            for (int q = 0; q < 5; ++q)
                if (int rc = fold_remote_pair(pairs[q][0], pairs[q][1], pairs[q][0], pairs[q][1], 0, fac[q][0], fac[q][1])) return rc;
        }
    }
    evp_launch_prep_average_prep2(P, S.d.nblocks, S.stream);      // T -> U averages and dyn_prep2 in one launch
    // ghost velocities before the loop (:729-732): the same exchange as inside the loop
    {
        const bool pushed = S.push_ok && (S.flags & S.flags_allowed & EVP_F_PUSH);
        if (pushed)      // the in-kernel push only exists inside the subcycle kernel: use the gather lists here
            evp_launch_halo_local(S.u[0], S.v[0], S.h_local_dst, S.h_local_src,
                                  (const signed char *)S.h_local_sign, S.n_local, S.stream);
        if (int rc = halo_uv(0)) return rc;
    }
    {   // the second ping-pong copy of the state: one launch for the fourteen arrays
        EvpCopyTab T{};
        T.len = S.n;
        T.vec2 = 1;
        for (int k = 0; k < 12; ++k) { T.src[T.n] = S.sig[0][k]; T.dst[T.n] = S.sig[1][k]; ++T.n; }
        T.src[T.n] = S.u[0]; T.dst[T.n] = S.u[1]; ++T.n;
        T.src[T.n] = S.v[0]; T.dst[T.n] = S.v[1]; ++T.n;
        for (int k = 0; k < T.n; ++k)
            if ((((uintptr_t)T.src[k]) | ((uintptr_t)T.dst[k])) & 15u) T.vec2 = 0;
        evp_launch_copy_many(T, S.stream);
    }
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
    ++S.upload_seq;
    if (S.hmask_prev != S.hmask) { S.res2_order_stale = true; S.hmask_prev = S.hmask; }
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
    ++S.upload_seq;
    return h2d(S.in[F_STRENGTH], strength);
}

// Seabed stress factor, computed by the host AFTER the preparation (seabed_stress_factor_LKD/_prob need
// the new iceUmask: evp() calls them between dyn_prep2 and the loop, ice_dyn_evp.F90:770-826) -- or
// on the device by cice_evp_hip_seabed_lkd.  Re-derives the "TbU == 0 everywhere" shortcut.
int cice_evp_hip_set_tbu(const double *TbU)
{
    if (!S.ready || !S.uploaded) return fail(-1, "state not uploaded");
    if (!TbU) return fail(-1, "null argument");
    if (h2d(S.in[F_TBU], TbU)) return -1;
    bool zero = true;
    for (size_t k = 0; k < S.n && zero; ++k) zero = !(S.hmask[k] & 2u) || TbU[k] == 0.0;
    HIPC(hipStreamSynchronize(S.stream));      // the caller may reuse its array
    S.flags &= ~EVP_F_TBU_ZERO;
    if (zero) S.flags |= EVP_F_TBU_ZERO;
    return 0;
}

// Seabed stress factor on the device, LKD method (seabed_stress_factor_LKD, ice_dyn_shared.F90:1386-1460; call
// site ice_dyn_evp.F90:783-790): from the aice / vice the last cice_evp_hip_prep uploaded (their ghost cells are
// refreshed here), the water depth (hwater: ice_flux, static unless the host couples it -- NULL keeps the copy
// of the previous call) and the masks that preparation produced.  Call between _prep and _subcycle instead of
// cice_evp_hip_set_tbu.  One device exp() per ice U-cell: <= 1 ulp from the host libm's.
int cice_evp_hip_seabed_lkd(const double *hwater, double k1, double k2, double alphab, double threshold_hw)
{
    if (!S.ready || !S.uploaded || !S.prep.geo) return fail(-1, "no prepared state (cice_evp_hip_prep first)");
    State::Prep &Q = S.prep;
    if (S.plan.center_remote) return fail(-9, "device seabed stress factor: T-grid ghosts on other ranks are not refreshed here; "
                                              "compute TbU on the host (cice_evp_hip_set_tbu)");
    if (!Q.hwater) {
        if (!hwater) return fail(-1, "hwater needed on the first call");
        if (alloc_d(&Q.hwater, S.n)) return -1;
    }
    if (hwater && h2d(Q.hwater, hwater)) return -1;
    EvpPrepHalo H{};
    H.a[0] = Q.t[0]; H.a[1] = Q.t[1]; H.a[2] = Q.hwater; H.narr = 3;     // scalars at cell centres
    H.dst = Q.c_dst; H.src = Q.c_src; H.vsign = (const signed char *)Q.c_vsign; H.n = Q.n_center;
    evp_launch_halo_center(H, S.stream);
    EvpPrep P{};
    P.nx = S.d.nx_block; P.ny = S.d.ny_block; P.plane = S.plane; P.blk = S.blk;
    P.t[0] = Q.t[0]; P.t[1] = Q.t[1]; P.mask = S.mask;
    HIPC(hipMemsetAsync(Q.flagword, 0, sizeof(unsigned), S.stream));
    evp_launch_seabed_lkd(P, S.d.nblocks, Q.hwater, S.in[F_TBU], k1, k2, alphab, threshold_hw, Q.flagword, S.stream);
    unsigned fw = 0;
    HIPC(hipMemcpyAsync(&fw, Q.flagword, sizeof(unsigned), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    S.flags &= ~EVP_F_TBU_ZERO;
    if (!(fw & 2u)) S.flags |= EVP_F_TBU_ZERO;
    return 0;
}

// Seabed stress factor on the device, probabilistic method (seabed_stress_factor_prob, ice_dyn_shared.F90:1475-1683;
// call site ice_dyn_evp.F90:791-800): from the category concentrations / volumes aicen, vicen (ice_state:
// (nx_block, ny_block, ncat, max_blocks), ghost cells current as in the host model), the water depth (NULL keeps the
// copy of the previous call) and the masks the last cice_evp_hip_prep produced.  Call between _prep and _subcycle
// instead of cice_evp_hip_set_tbu.  exp() / log() are the device library's: TbU within a few ulp of the reference's.
int cice_evp_hip_seabed_prob(const double *hwater, const double *aicen, const double *vicen, int32_t ncat, double alphab,
                             double rhoi, double gravit, double pi, double puny)
{
    if (!S.ready || !S.uploaded || !S.prep.geo) return fail(-1, "no prepared state (cice_evp_hip_prep first)");
    if (!aicen || !vicen || ncat < 1) return fail(-1, "bad argument");
    State::Prep &Q = S.prep;
    if (!Q.hwater) {
        if (!hwater) return fail(-1, "hwater needed on the first call");
        if (alloc_d(&Q.hwater, S.n)) return -1;
    }
    if (hwater && h2d(Q.hwater, hwater)) return -1;
    if (Q.ncat != ncat) {
        if (Q.aicen) { (void)hipFree(Q.aicen); Q.aicen = nullptr; }
        if (Q.vicen) { (void)hipFree(Q.vicen); Q.vicen = nullptr; }
        if (alloc_d(&Q.aicen, S.n * (size_t)ncat) || alloc_d(&Q.vicen, S.n * (size_t)ncat)) return -1;
        Q.ncat = ncat;
    }
    if (!Q.tbt && alloc_d(&Q.tbt, S.n)) return -1;
    // the caller's arrays are (nx, ny, ncat, max_blocks): blocks 1..nblocks are contiguous
    HIPC(hipMemcpyAsync(Q.aicen, aicen, S.n * (size_t)ncat * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemcpyAsync(Q.vicen, vicen, S.n * (size_t)ncat * sizeof(double), hipMemcpyHostToDevice, S.stream));
    EvpPrep P{};
    P.nx = S.d.nx_block; P.ny = S.d.ny_block; P.plane = S.plane; P.blk = S.blk;
    P.mask = S.mask;
    HIPC(hipMemsetAsync(Q.flagword, 0, sizeof(unsigned), S.stream));
    evp_launch_seabed_prob(P, S.d.nblocks, Q.hwater, Q.aicen, Q.vicen, ncat, alphab, rhoi, S.prm.rhow, gravit, pi, puny, Q.tbt,
                           S.in[F_TBU], Q.flagword, S.stream);
    unsigned fw = 0;
    HIPC(hipMemcpyAsync(&fw, Q.flagword, sizeof(unsigned), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    S.flags &= ~EVP_F_TBU_ZERO;
    if (!(fw & 2u)) S.flags |= EVP_F_TBU_ZERO;
    ++S.upload_seq;
    return 0;
}

// products of the preparation phase that stay on the device, for hosts that need them
// (coupling diagnostics) and for the tests
int cice_evp_hip_prep_fetch(int32_t which, double *dst)
{
    if (!S.ready || !S.uploaded || !S.prep.geo) return fail(-1, "no prepared state");
    if (!dst) return fail(-1, "null argument");
    State::Prep &Q = S.prep;
    const double *tab[21] = {S.in[F_AIX], S.in[F_CW], S.in[F_UOCN], S.in[F_VOCN], S.in[F_UMASSDTI], S.in[F_FM],
                             S.in[F_WATERX], S.in[F_WATERY], S.in[F_FORCEX], S.in[F_FORCEY], S.in[F_UVEL_INIT],
                             S.in[F_VVEL_INIT], Q.strtltx, Q.strtlty, Q.strairxU, Q.strairyU, Q.tmass, Q.umass,
                             S.u[S.cur], S.v[S.cur], S.in[F_TBU]};
    if (which < 0 || which >= 21) return fail(-1, "prep_fetch: which = %d", (int)which);
    if (d2h(dst, tab[which])) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

}  // extern "C"
