// =====================================================================
// C-grid EVP subcycle behind the C ABI (include/cice_evp_hip.h, cice_evp_hip_cgrid_*): device state, ghost-image
// table, the loop as a captured graph of three (fused schedule) or five launches per subcycle (evp_cgrid.hip).
//
// Replaces evp()'s loop for grid_ice = 'C' (ice_dyn_evp.F90:938-1099).  The caller has run the reference's own
// preparation (dyn_prep1/2 at U, N and E points, seabed stress, the grid averages of the forcing) and hands over
// what the loop reads; it gets back what the loop writes.  Cyclic / closed / open boundaries; any number of blocks
// per rank (their ghost cells are images like any other).  Ghost cells that mirror cells of OTHER ranks are filled
// by the B-grid path's velocity exchange (halo_remote_pair: mailbox stores over xGMI, or RCCL point-to-point) run on
// pairs of the loop's arrays after the launch that produces them -- the same points at which the reference calls
// ice_HaloUpdate.  Tripole (u-fold) grids: a fold step per exchange point from host-built lists (halo_plan.cpp:
// build_fold_list), the blocks next to the fold on one rank.
// =====================================================================
#include "evp_host.h"

namespace evp_host {

struct CGridState {
    bool geo = false, uploaded = false;
    double *f[CG_NF] = {}, *in[CG_NIN] = {}, *g[CG_NG] = {};
    double *strengthU = nullptr;
    double *umaskd = nullptr;    // ranks > 1: iceU as a field, to learn the flags of ghost cells other ranks own
    double *fac[2] = {nullptr, nullptr};   // leading factor of vrel at E / N (once per call)
    unsigned *d_flags = nullptr;
    bool fast = false;           // the shortcuts of cg_stress_u_step<true> hold on every ice cell of this call
    double *s12alt = nullptr;    // second stress12U buffer of the fused schedule (f[CF_S12U] always holds the current one)
    int flip = 0;                // which of the two allocations f[CF_S12U] is (part of the graph key)
    uint8_t *mask = nullptr;
    int *img_slot = nullptr, *img_dst = nullptr;
    // tripole fold: per field location the cells of the fold row / the ghost row beyond it and their sources
    bool tripole = false;
    struct Fold { int *dst = nullptr, *a = nullptr, *b = nullptr; unsigned char *flip = nullptr; int n = 0; } fold[4];
    double *fold_tmp = nullptr;
    int fold_maxn = 0;
    int *zero_cells = nullptr;   // ghost cells of eliminated (land) neighbour blocks
    int n_zero = 0;
    std::vector<int> h_img_slot, h_img_dst;
    int32_t *mask4 = nullptr;    // the caller's four logical masks as uploaded (4 x n words)
    int avg_strength = 0;
    bool first = true;           // no subcycle has run since the upload
    std::map<std::pair<int, int>, hipGraphExec_t> graphs;   // (ndte, flip << 3 | fused << 2 | first << 1 | avg_strength)
    double t_loop_ms = 0;
    int t_nsub = 0;
    double *tarear = nullptr, *post[5] = {};   // deformationsC_T: 1/tarea (static), divu shear vort rdg_conv rdg_shear
};
static CGridState CG;

void cgrid_free()
{
    auto F = [](auto *&p) {
        if (p) (void)hipFree((void *)p);
        p = nullptr;
    };
    for (auto &p : CG.f) F(p);
    for (auto &p : CG.in) F(p);
    for (auto &p : CG.g) F(p);
    F(CG.tarear); for (auto &p : CG.post) F(p);
    F(CG.strengthU); F(CG.s12alt); F(CG.umaskd); F(CG.fac[0]); F(CG.fac[1]); F(CG.d_flags); F(CG.mask); F(CG.mask4); F(CG.img_slot); F(CG.img_dst); F(CG.zero_cells); F(CG.fold_tmp);
    for (auto &f : CG.fold) { F(f.dst); F(f.a); F(f.b); F(f.flip); }
    for (auto &kv : CG.graphs) (void)hipGraphExecDestroy(kv.second);
    CG.graphs.clear();
    CG = CGridState();
}

static void fill(EvpCgrid &A)
{
    const cice_evp_hip_params &q = S.prm;
    for (int k = 0; k < CG_NF; ++k) A.f[k] = CG.f[k];
    for (int k = 0; k < CG_NIN; ++k) A.in[k] = CG.in[k];
    for (int k = 0; k < CG_NG; ++k) A.g[k] = CG.g[k];
    A.strengthU = CG.strengthU;
    A.s12_in = nullptr;
    A.facE = CG.fac[0];
    A.facN = CG.fac[1];
    A.mask = CG.mask;
    A.img_slot = CG.img_slot;
    A.img_dst = CG.img_dst;
    A.blk = S.blk;
    A.p = {q.arlx1i, q.denom1, q.brlx, q.revp, q.e_factor, q.epp2i, q.capping, q.Ktens, q.u0, q.cosw, q.sinw, q.rhow};
    A.deltaminEVP = q.deltaminEVP;
    A.nx = S.d.nx_block;
    A.ny = S.d.ny_block;
    A.nblocks = S.d.nblocks;
    A.avg_strength = CG.avg_strength;
    A.tripole = CG.tripole ? 1 : 0;
    {   // two waves per cell in the fused step kernel while the grid is small enough to be latency-bound
        const char *e = env("CICE_EVP_HIP_CGRID_SPLIT");
        A.split_faces = e ? (std::atoi(e) != 0) : (S.n <= 600000);
    }
    {   // XCD-banded workgroup numbering (evp_cgrid.hip: cell()); CICE_EVP_HIP_CGRID_XCD=0: plain 2-D launch
        const bool on = !(env("CICE_EVP_HIP_CGRID_XCD") && !std::atoi(env("CICE_EVP_HIP_CGRID_XCD")));
        const int gy = (S.d.ny_block + 3) / 4;      // TY = 4 rows per workgroup
        const int gx = (S.d.nx_block + 63) / 64;    // TX = 64
        const int rows = env("CICE_EVP_HIP_CGRID_XCD") && std::atoi(env("CICE_EVP_HIP_CGRID_XCD")) > 1 ? std::atoi(env("CICE_EVP_HIP_CGRID_XCD")) : std::max(1, 256 / gx);
        A.xcd_rows = on ? std::min((gy + 7) / 8, rows) : 0;
    }
    A.plane = S.plane;
}

static bool remote() { return !S.plan.peers.empty(); }
// ghost cells owned by other ranks, for two of the loop's arrays (no-op on one rank).  Every exchange inside the loop
// goes through the masked halo when the host has handed one over (maskhalo_dyn: evp() builds it for the C grid as the
// five-point dilation of iceTmask, ice_dyn_evp.F90:739-770, and passes halo_info_mask to every dyn_haloUpdate of the
// loop, :965-1096); copies inside a rank -- the pushes of the kernels -- are never masked, as in ice_HaloMask.
#define XCHG(a, b)                                             \
    do {                                                       \
        if (remote())                                          \
            if (int rc_ = halo_remote_pair((a), (b), true)) return rc_; \
    } while (0)

// tripole: the fold step of up to four fields after the launch that produced them (what their ice_HaloUpdate does
// at the fold: points ON it averaged with their partners, ghost row beyond it mirrored); loc 0 centre, 1 NE corner,
// 2 E face, 3 N face; vec: vector kind (sign change across the fold)
struct FoldField { double *x; int loc; bool vec; };
static void fold(std::initializer_list<FoldField> fs)
{
    if (!CG.tripole) return;
    EvpCgFold F{};
    for (const FoldField &f : fs) {
        F.x[F.nfields] = f.x;
        F.loc[F.nfields] = f.loc;
        F.isign[F.nfields] = f.vec ? -1.0 : 1.0;
        ++F.nfields;
    }
    for (int l = 0; l < 4; ++l) F.L[l] = {CG.fold[l].dst, CG.fold[l].a, CG.fold[l].b, CG.fold[l].flip, CG.fold[l].n};
    F.tmp = CG.fold_tmp;
    F.maxn = CG.fold_maxn;
    evp_launch_cgrid_fold(F, S.stream);
}

static bool fused_schedule()
{
    if (CG.avg_strength) return false;           // needs deltaU at the neighbours: five phases
    if (CG.tripole) return false;                // recomputing a neighbour across the fold would sum in mirrored order
    return !(env("CICE_EVP_HIP_CGRID_FUSED") && !std::atoi(env("CICE_EVP_HIP_CGRID_FUSED")));
}

// five launches per subcycle, any visc_method
static int enqueue_phases(const EvpCgrid &A, int ndte, bool first)
{
    for (int k = 0; k < ndte; ++k) {
        evp_launch_cgrid_phase(A, 0, 1, S.stream);
        // the first strain_rates_U still reads the caller's ghost values of uvelN / vvelE
        if (first && k == 0) {
            evp_launch_cgrid_phase(A, 6, 1, S.stream);
            evp_launch_cgrid_zero_cells(A, CG.zero_cells, CG.n_zero, S.stream);
        }
        XCHG(A.f[CF_SHEARU], A.f[CF_SHEARU]);
        fold({{A.f[CF_SHEARU], 1, false}});
        evp_launch_cgrid_phase(A, 1, 1, S.stream);
        XCHG(A.f[CF_ETA], A.f[CF_ZETA]);
        XCHG(A.f[CF_SP], A.f[CF_SM]);
        fold({{A.f[CF_ZETA], 0, false}, {A.f[CF_ETA], 0, false}, {A.f[CF_SP], 0, false}, {A.f[CF_SM], 0, false}});
        evp_launch_cgrid_phase(A, 2, 1, S.stream);
        XCHG(A.f[CF_S12U], A.f[CF_S12U]);
        fold({{A.f[CF_S12U], 1, false}});
        evp_launch_cgrid_phase(A, 3, 1, S.stream);
        XCHG(A.f[CF_UE], A.f[CF_VN]);
        fold({{A.f[CF_UE], 2, true}, {A.f[CF_VN], 3, true}});
        evp_launch_cgrid_phase(A, 4, 1, S.stream);
        XCHG(A.f[CF_UN], A.f[CF_VE]);
        XCHG(A.f[CF_UU], A.f[CF_VU]);
        fold({{A.f[CF_UN], 3, true}, {A.f[CF_VE], 2, true}, {A.f[CF_UU], 1, true}, {A.f[CF_VU], 1, true}});
    }
    return 0;
}

// three launches per subcycle + one after the loop (evp_cgrid.hip); stress12U ping-pongs, returns with the
// current values in `cur` (the caller swaps the pointers when ndte is odd)
static int enqueue_fused(EvpCgrid A, int ndte, bool first)
{
    double *cur = CG.f[CF_S12U], *other = CG.s12alt;
    if (first) {
        // the caller's ghost cells of stress12U are whatever dyn_prep left there (zero: iceUmask is never set on
        // ghost cells, ice_dyn_evp.F90:683-690); the reference repairs them with the first halo update, the fused
        // kernel recomputes neighbours from their previous value: make the previous values ghost-consistent first
        evp_launch_cgrid_phase(A, 9, CF_S12U, S.stream);
        XCHG(cur, cur);
        // (ghost cells of eliminated land blocks: zero in both buffers; no ice cell reads them before the first exchange)
        evp_launch_cgrid_zero_cells(A, CG.zero_cells, CG.n_zero, S.stream);
        HIPC(hipMemcpyAsync(other, cur, S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    }
    for (int k = 0; k < ndte; ++k) {
        const int last = (k == ndte - 1);
        A.f[CF_S12U] = cur;
        if (first && k == 0) {
            evp_launch_cgrid_phase(A, 0, 1, S.stream);
            evp_launch_cgrid_phase(A, 6, 1, S.stream);
        } else {
            evp_launch_cgrid_phase(A, 7, last, S.stream);
        }
        XCHG(A.f[CF_SHEARU], A.f[CF_SHEARU]);
        evp_launch_cgrid_phase(A, 10, last, S.stream);
        XCHG(A.f[CF_ETA], A.f[CF_ZETA]);         // (zetax2T is stored in the last subcycle only; harmless before)
        XCHG(A.f[CF_SP], A.f[CF_SM]);
        A.s12_in = cur;
        A.f[CF_S12U] = other;
        evp_launch_cgrid_phase(A, CG.fast ? 11 : 8, last, S.stream);
        XCHG(other, other);
        XCHG(A.f[CF_UE], A.f[CF_VN]);
        std::swap(cur, other);
    }
    A.f[CF_S12U] = cur;
    evp_launch_cgrid_phase(A, 4, 1, S.stream);
    XCHG(A.f[CF_UN], A.f[CF_VE]);
    XCHG(A.f[CF_UU], A.f[CF_VU]);
    return 0;
}

// Fold lists (halo_plan.cpp: build_fold_list) on the device
static int build_fold_lists()
{
    if (S.d.nx_global % 2) return fail(-4, "tripole: nx_global must be even");
    cice_evp_hip_dims d = S.d;
    d.ilo = S.ilo.data(); d.ihi = S.ihi.data(); d.jlo = S.jlo.data(); d.jhi = S.jhi.data();
    d.iglob0 = S.iglob0.data(); d.jglob0 = S.jglob0.data();
    CG.fold_maxn = 0;
    for (int loc = 0; loc < 4; ++loc) {
        FoldList L;
        build_fold_list(d, loc, L);
        const std::vector<int32_t> &dst = L.dst, &a = L.a, &bb = L.b;
        const std::vector<uint8_t> &flip = L.flip;
        CGridState::Fold &Fd = CG.fold[loc];
        Fd.n = (int)dst.size();
        CG.fold_maxn = std::max(CG.fold_maxn, Fd.n);
        if (!Fd.n) continue;
        HIPC(hipMalloc((void **)&Fd.dst, dst.size() * sizeof(int)));
        HIPC(hipMalloc((void **)&Fd.a, dst.size() * sizeof(int)));
        HIPC(hipMalloc((void **)&Fd.b, dst.size() * sizeof(int)));
        HIPC(hipMalloc((void **)&Fd.flip, dst.size()));
        HIPC(hipMemcpy(Fd.dst, dst.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Fd.a, a.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Fd.b, bb.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(Fd.flip, flip.data(), dst.size(), hipMemcpyHostToDevice));
    }
    if (CG.fold_maxn) HIPC(hipMalloc((void **)&CG.fold_tmp, (size_t)4 * CG.fold_maxn * sizeof(double)));
    return 0;
}

}  // namespace evp_host

using namespace evp_host;

extern "C" {

int cice_evp_hip_cgrid_set_geometry(const double *const *static23)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!static23) return fail(-1, "null argument");
    const HaloPlan &P = S.plan;
    const bool tripole = S.d.ns_boundary_type == CICE_EVP_BND_TRIPOLE;
    if (S.d.ns_boundary_type > CICE_EVP_BND_TRIPOLE) return fail(-4, "C-grid EVP: tripoleT is not supported");
    if (tripole && P.fold_rows == 2)
        return fail(-4, "C-grid EVP on a tripole grid: the blocks next to the fold (rows NY-1, NY) must all be on one rank "
                        "(split the domain in y only); here they are shared with other ranks");
    cgrid_free();
    CG.tripole = tripole;
    for (auto &p : CG.f)
        if (alloc_d(&p, S.n)) return -1;
    for (auto &p : CG.in)
        if (alloc_d(&p, S.n)) return -1;
    for (int k = 0; k < CG_NG; ++k) {
        if (!static23[k]) return fail(-1, "null static array %d", k);
        if (alloc_d(&CG.g[k], S.n) || h2d(CG.g[k], static23[k])) return -1;
    }
    if (alloc_d(&CG.strengthU, S.n) || alloc_d(&CG.s12alt, S.n) || alloc_d(&CG.fac[0], S.n) || alloc_d(&CG.fac[1], S.n)) return -1;
    HIPC(hipMalloc((void **)&CG.d_flags, sizeof(unsigned)));
    if (!S.plan.peers.empty() && alloc_d(&CG.umaskd, S.n)) return -1;
    HIPC(hipMalloc((void **)&CG.mask, S.n));
    HIPC(hipMalloc((void **)&CG.mask4, 4 * S.n * sizeof(int32_t)));
    // ghost images: for every interior cell the ghost cells of this rank that mirror it (what ice_HaloUpdate copies)
    CG.h_img_slot.assign(S.n, -1);
    std::vector<int> dst, zero;
    for (size_t k = 0; k < P.local_dst.size(); ++k) {
        const int src = P.local_src[k];
        if (tripole) {       // ghost row beyond the fold: location-dependent, done by the fold step, not by an image
            const int db = (int)(P.local_dst[k] / S.plane);
            const int dj = (int)((P.local_dst[k] % S.plane) / S.d.nx_block) + 1;
            if (S.jglob0[db] + (dj - S.jlo[db]) > S.d.ny_global) continue;
        }
        if (src < 0) {                              // neighbour block eliminated (land): the reference fills with zero
            zero.push_back(P.local_dst[k]);
            continue;
        }
        int &slot = CG.h_img_slot[src];
        if (slot < 0) {
            slot = (int)(dst.size() / 3);
            dst.insert(dst.end(), 3, -1);
        }
        int w = 0;
        while (w < 3 && dst[3 * slot + w] >= 0) ++w;
        if (w == 3) return fail(-4, "C-grid EVP: a cell with more than three ghost images");
        dst[3 * slot + w] = P.local_dst[k];
    }
    if (dst.empty()) dst.assign(3, -1);
    CG.h_img_dst = dst;
    HIPC(hipMalloc((void **)&CG.img_slot, S.n * sizeof(int)));
    HIPC(hipMalloc((void **)&CG.img_dst, dst.size() * sizeof(int)));
    HIPC(hipMemcpyAsync(CG.img_slot, CG.h_img_slot.data(), S.n * sizeof(int), hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemcpyAsync(CG.img_dst, dst.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice, S.stream));
    if (tripole && P.fold_rows == 1)             // (ranks without the fold rows run the same schedule with empty lists)
        if (int rc = build_fold_lists()) return rc;
    CG.n_zero = (int)zero.size();
    if (CG.n_zero) {
        HIPC(hipMalloc((void **)&CG.zero_cells, zero.size() * sizeof(int)));
        HIPC(hipMemcpyAsync(CG.zero_cells, zero.data(), zero.size() * sizeof(int), hipMemcpyHostToDevice, S.stream));
    }
    HIPC(hipStreamSynchronize(S.stream));
    CG.geo = true;
    return 0;
}

int cice_evp_hip_cgrid_upload(const double *const *state14, const double *const *inputs23, const int32_t *iceTmask,
                              const int32_t *iceUmask, const int32_t *iceEmask, const int32_t *iceNmask,
                              int32_t visc_method)
{
    if (!S.ready || !CG.geo) return fail(-1, "C-grid EVP: geometry not set");
    if (!state14 || !inputs23 || !iceTmask || !iceUmask || !iceEmask || !iceNmask) return fail(-1, "null argument");
    if (visc_method != 0 && visc_method != 1) return fail(-1, "visc_method %d (0 avg_zeta, 1 avg_strength)", visc_method);
    CopyBatch B;
    for (int k = 0; k < 14; ++k) {
        if (!state14[k]) return fail(-1, "null state array %d", k);
        B.items.push_back({CG.f[k], state14[k]});
    }
    for (int k = 0; k < CG_NIN; ++k) {
        if (!inputs23[k]) return fail(-1, "null input array %d", k);
        B.items.push_back({CG.in[k], inputs23[k]});
    }
    if (h2d_batch(B)) return -1;
    // evp() zeroes its work arrays at entry (ice_dyn_evp.F90:351-361)
    for (int k = CF_ZETA; k < CG_NF; ++k) HIPC(hipMemsetAsync(CG.f[k], 0, S.n * sizeof(double), S.stream));
    // the mask byte is composed on the device from the caller's four logical arrays (bit5: iceU of an interior cell,
    // handed on to the ghost cells that mirror it -- the caller's iceUmask is not maintained on ghost cells: dyn_prep2
    // sets it on ilo..ihi x jlo..jhi only, ice_dyn_shared.F90:740-745)
    {
        const int32_t *m[4] = {iceTmask, iceUmask, iceEmask, iceNmask};
        for (int k = 0; k < 4; ++k)
            HIPC(hipMemcpyAsync(CG.mask4 + (size_t)k * S.n, m[k], S.n * sizeof(int32_t), hipMemcpyHostToDevice, S.stream));
        EvpCgrid A;
        fill(A);
        evp_launch_cgrid_mask(A, CG.mask4, S.stream);
    }
    CG.avg_strength = visc_method;
    unsigned h_flags = ~0u;
    {
        EvpCgrid A;
        fill(A);
        HIPC(hipMemsetAsync(CG.d_flags, 0, sizeof(unsigned), S.stream));
        evp_launch_cgrid_call_setup(A, CG.fac[0], CG.fac[1], CG.d_flags, S.stream);
        HIPC(hipMemcpyAsync(&h_flags, CG.d_flags, sizeof(unsigned), hipMemcpyDeviceToHost, S.stream));
    }
    if (remote()) {                              // bit5 of ghost cells other ranks own
        EvpCgrid A;
        fill(A);
        evp_launch_cgrid_umask(A, CG.umaskd, 0, S.stream);
        if (int rc = halo_remote_pair(CG.umaskd, CG.umaskd)) return rc;
        evp_launch_cgrid_umask(A, CG.umaskd, 1, S.stream);
    }
    if (visc_method == 1) {
        EvpCgrid A;
        fill(A);
        evp_launch_cgrid_phase(A, 5, 1, S.stream);
    }
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(S.stream));       // the caller may change its arrays after this returns
    CG.fast = h_flags == 0 && !(env("CICE_EVP_HIP_CGRID_FAST") && !std::atoi(env("CICE_EVP_HIP_CGRID_FAST")));
    CG.uploaded = true;
    CG.first = true;
    return 0;
}

int cice_evp_hip_cgrid_subcycle(int32_t ndte)
{
    if (!CG.uploaded) return fail(-1, "C-grid EVP: nothing uploaded");
    if (ndte < 0) return fail(-1, "ndte < 0");
    if (ndte == 0) return 0;
    EvpCgrid A;
    fill(A);
    const bool fused = fused_schedule();
    auto enqueue = [&]() -> int {
        if (fused) return enqueue_fused(A, ndte, CG.first);
        return enqueue_phases(A, ndte, CG.first);
    };
    HIPC(hipEventRecord(S.ev0, S.stream));
    if (S.use_graph && (!remote() || S.direct.on)) {     // RCCL point-to-point is enqueued eagerly (as the B-grid loop does)
        const std::pair<int, int> key(ndte, (CG.fast ? 16 : 0) | (CG.flip << 3) | (fused ? 4 : 0) | (CG.first ? 2 : 0) | CG.avg_strength);
        auto it = CG.graphs.find(key);
        if (it == CG.graphs.end()) {
            hipGraph_t gr = nullptr;
            hipGraphExec_t ex = nullptr;
            HIPC(hipStreamBeginCapture(S.stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue();
            HIPC(hipStreamEndCapture(S.stream, &gr));
            if (rc) return rc;
            HIPC(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
            (void)hipGraphDestroy(gr);
            it = CG.graphs.emplace(key, ex).first;
        }
        HIPC(hipGraphLaunch(it->second, S.stream));
    } else if (enqueue()) {
        return -1;
    }
    if (fused && (ndte & 1)) {                   // the current stress12U is in the other allocation now
        std::swap(CG.f[CF_S12U], CG.s12alt);
        CG.flip ^= 1;
    }
    HIPC(hipEventRecord(S.ev1, S.stream));
    HIPC(hipGetLastError());
    CG.first = false;
    CG.t_nsub = ndte;
    return 0;
}

int cice_evp_hip_cgrid_download(double *const *fields19)
{
    if (!CG.uploaded) return fail(-1, "C-grid EVP: nothing uploaded");
    if (!fields19) return fail(-1, "null argument");
    CopyBatch B;
    for (int k = 0; k < CG_NF; ++k)
        if (fields19[k]) B.items.push_back({fields19[k], CG.f[k]});
    if (d2h_batch(B)) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    if (CG.t_nsub && hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess) CG.t_loop_ms = ms;
    return 0;
}

// deformationsC_T (ice_dyn_shared.F90:1968-2074; evp() calls it right after the C-grid loop, ice_dyn_evp.F90:1106-1119) on
// the device, from the loop's resident final state: divu, shear, vort, rdg_conv, rdg_shear on the T-cells of dyn_prep2's
// list; the five arrays are inout (every other cell keeps the caller's value).  tarear: ice_grid's array, taken at the
// first call (NULL afterwards keeps it).
int cice_evp_hip_cgrid_deformations(const double *tarear, double *divu, double *shear, double *vort, double *rdg_conv,
                                    double *rdg_shear)
{
    if (!CG.uploaded) return fail(-1, "C-grid EVP: nothing uploaded");
    double *host[5] = {divu, shear, vort, rdg_conv, rdg_shear};
    for (double *p : host)
        if (!p) return fail(-1, "null argument");
    if (!CG.tarear) {
        if (!tarear) return fail(-1, "tarear needed on the first call");
        if (alloc_d(&CG.tarear, S.n)) return -1;
    }
    if (tarear && h2d(CG.tarear, tarear)) return -1;
    CopyBatch U;
    for (int k = 0; k < 5; ++k) {
        if (!CG.post[k] && alloc_d(&CG.post[k], S.n)) return -1;
        U.items.push_back({CG.post[k], host[k]});
    }
    if (h2d_batch(U)) return -1;
    EvpCgrid A;
    fill(A);
    evp_launch_cgrid_deformations(A, CG.tarear, CG.post[0], CG.post[1], CG.post[2], CG.post[3], CG.post[4], S.stream);
    HIPC(hipGetLastError());
    CopyBatch D;
    for (int k = 0; k < 5; ++k) D.items.push_back({host[k], CG.post[k]});
    if (d2h_batch(D)) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

int cice_evp_hip_cgrid_sync(void)
{
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    if (CG.t_nsub && hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess) CG.t_loop_ms = ms;
    return 0;
}

int cice_evp_hip_cgrid_run(int32_t ndte, int32_t visc_method, double *const *fields19, const double *const *inputs23,
                           const int32_t *iceTmask, const int32_t *iceUmask, const int32_t *iceEmask,
                           const int32_t *iceNmask)
{
    if (cice_evp_hip_cgrid_upload(fields19, inputs23, iceTmask, iceUmask, iceEmask, iceNmask, visc_method)) return -1;
    if (cice_evp_hip_cgrid_subcycle(ndte)) return -1;
    return cice_evp_hip_cgrid_download(fields19);
}

int cice_evp_hip_cgrid_timings(double *out, int32_t n)
{
    if (!out || n < 2) return fail(-1, "need room for 2 values");
    out[0] = CG.t_loop_ms;
    out[1] = (double)CG.t_nsub;
    return 0;
}

}  // extern "C"
