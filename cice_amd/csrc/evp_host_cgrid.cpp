// =====================================================================
// C-grid EVP subcycle behind the C ABI (include/cice_evp_hip.h, cice_evp_hip_cgrid_*): device state, ghost-image
// table, the loop as a captured graph of one (cg_one: one rank, no fold), three (fused schedule) or five launches per
// subcycle (evp_cgrid.hip).
//
// Replaces evp()'s loop for grid_ice = 'C' (ice_dyn_evp.F90:938-1099).  The caller has run the reference's own
// preparation (dyn_prep1/2 at U, N and E points, seabed stress, the grid averages of the forcing) and hands over
// what the loop reads; it gets back what the loop writes.  Cyclic / closed / open boundaries; any number of blocks
// per rank (their ghost cells are images like any other).  Ghost cells that mirror cells of OTHER ranks are filled
// by the B-grid path's velocity exchange (halo_remote_pair: mailbox stores over xGMI, or RCCL point-to-point) run on
// pairs of the loop's arrays after the launch that produces them -- the same points at which the reference calls
// ice_HaloUpdate.  Tripole grids, u-fold and T-fold (tripoleT): a fold step per exchange point from host-built lists
// (halo_plan.cpp: build_fold_list / build_fold_list_tfold), the blocks next to the fold on one rank.
// =====================================================================
#include <cmath>
#include <set>

#include "evp_host.h"

namespace evp_host {

struct CGridState {
    bool geo = false, uploaded = false;
    double *f[CG_NF] = {}, *in[CG_NIN] = {}, *g[CG_NG] = {};
    double *gslab = nullptr, *inslab = nullptr;   // g[k] = gslab + k * n, in[k] = inslab + k * n: one allocation per table (cg_one addresses them as base + k * stride)
    double *strengthU = nullptr;
    double *umaskd = nullptr;    // ranks > 1: iceU as a field, to learn the flags of ghost cells other ranks own
    double *fac[2] = {nullptr, nullptr};   // leading factor of vrel at E / N (once per call)
    unsigned *d_flags = nullptr;
    bool fast = false;           // the shortcuts of cg_stress_u_step<true> hold on every ice cell of this call
    double *s12alt = nullptr;    // second stress12U buffer of the fused schedule (f[CF_S12U] always holds the current one)
    int flip = 0;                // which of the two allocations f[CF_S12U] is (part of the graph key)
    // one launch per subcycle (evp_cgrid.hip: cg_one): window table, second buffers of uvelE, vvelN, stresspT, stressmT
    struct One {
        int *tab = nullptr;
        int4 *tiles = nullptr;
        int ntiles = 0, per_xcd = 0, ox = 0, oy = 0;
        double *alt[4] = {};
        unsigned long long *prof = nullptr;   // test build: phase stamps (CICE_EVP_HIP_CGRID_PROF=1)
        int flip = 0;            // which allocation f[CF_UE], f[CF_VN], f[CF_SP], f[CF_SM] are (part of the graph key)
        // the interior of large blocks marched (evp_cgrid.hip: cg_strip): its items, and the windows cg_one keeps (the block edges)
        int *items = nullptr;
        int4 *tiles_e = nullptr;
        int *tab_e = nullptr;
        int nitems = 0, ntiles_e = 0, strip_seg = 0, ex = 0, ey = 0;   // (ex, ey: shape of the windows kept)
        int strip_len = 0;       // 1: the marched kernel forms six of the eight lengths from dxN, dyE
        long strip_cells = 0;    // cells the marched kernel owns
    } one;
    // all subcycles of a call in one launch, state on the chip (evp_cgrid_res.hip: cg_res)
    struct Res {
        int *tab = nullptr;
        int4 *tiles = nullptr, *tiles2 = nullptr;     // (tiles2: tripole grids, halo_plan.cpp build_fold_window_table)
        int ntiles = 0;
        uint8_t *pubmap = nullptr;
        uint8_t *gmask = nullptr;    // tripole grids: the land masks as bits (elsewhere CG.gmask, which the one-launch kernels share)
        void *rec = nullptr;         // EVP_CGRES_SLOTS x S.n records of 32 bytes
        int *err = nullptr;
        int *d_order = nullptr;      // the windows that hold ice in this call (cg_res_live at every upload), n_live of them
        int *live_win = nullptr;     // [ntiles] 0 / 1
        uint8_t *live_cell = nullptr;   // per cell: its window runs in this call
        int n_live = 0;
        std::vector<int> h_live;
        unsigned epoch = 0;
        int par = 0;
        bool images_ok = false;      // every ghost cell's static arrays equal its source's bit for bit
        int2 *pairs = nullptr;       // (array neighbour of a position's cell, the ghost cell outside the domain the table names for the
        int npairs = 0;              // neighbouring position) where the two differ: must hold the same values (static: checked once)
        bool pairs_state_ok = true;  // ... and the same state in this call (checked on the device at every upload)
        std::string why;             // ... or why the kernel is not eligible on this rank
        int mode = -1;               // -1 undecided (first eligible call probes), 0 off, 1 on
        bool launched = false;       // a launch whose error word has not been looked at
        long cap4[8] = {};           // windows that can be resident at once (occupancy x CUs), by kernel variant (avg_strength | revised << 1 | slow << 2)
        double t_probe_ms = -1.0;    // probe: ms per subcycle
        int last_nsub = 0;           // subcycles of the last call that ran inside it
        int fallbacks = 0;           // cice_evp_hip_cgrid_run calls repeated without it after a wait gave up
        unsigned long long *prof = nullptr;   // test build: phase stamps (CICE_EVP_HIP_CGRID_PROF=1)
    } res;
    uint8_t *mask = nullptr;
    uint8_t *gmask = nullptr;    // the four land masks as bits (cg_one's derived view of the static table); null: an identity failed
    std::string geo_why;         // ... and which one
    int *img_slot = nullptr, *img_dst = nullptr;
    // tripole fold: per field location the cells of the fold row / the ghost row beyond it and their sources
    bool tripole = false;
    bool tfold = false;                  // ... of the T-fold kind (tripoleT)
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
    int t_one = 0;               // subcycles of the last call that ran as one launch each (cg_one)
    double *tarear = nullptr, *post[5] = {};   // deformationsC_T: 1/tarea (static), divu shear vort rdg_conv rdg_shear
    // preparation phase on the device (cice_evp_hip_cgrid_prep)
    struct Prep {
        bool geo = false, pending = false;     // pending: prepared, cice_evp_hip_cgrid_prep_finish not yet called
        uint8_t *tmask = nullptr, *xmask[3] = {};
        double *fcor[3] = {}, *t[11] = {}, *tmass = nullptr, *maskd = nullptr;
        int *c_dst = nullptr, *c_src = nullptr;
        signed char *c_vsign = nullptr;
        int n_center = 0;
        double *hwater = nullptr, *tbt = nullptr, *aicen = nullptr, *vicen = nullptr;
        int ncat = 0;
        std::vector<int32_t> h4;               // the four mask words as they come back
        cice_evp_hip_prep_params pp{};
        double t_ms = 0;
    } prep;
};
static CGridState CG;

void cgrid_free()
{
    auto F = [](auto *&p) {
        if (p) (void)hipFree((void *)p);
        p = nullptr;
    };
    for (auto &p : CG.f) F(p);
    F(CG.gslab); F(CG.inslab); F(CG.one.prof);
    for (auto &p : CG.in) p = nullptr;
    for (auto &p : CG.g) p = nullptr;
    F(CG.tarear); for (auto &p : CG.post) F(p);
    F(CG.one.tab); F(CG.one.tiles); F(CG.one.items); F(CG.one.tiles_e); F(CG.one.tab_e); for (auto &p : CG.one.alt) F(p);
    CG.one = CGridState::One{};
    F(CG.res.tab); F(CG.res.tiles); F(CG.res.tiles2); F(CG.res.pubmap); F(CG.res.gmask); F(CG.res.rec); F(CG.res.err); F(CG.res.pairs); F(CG.res.prof); F(CG.res.d_order); F(CG.res.live_win); F(CG.res.live_cell);
    CG.res = CGridState::Res{};
    {
        CGridState::Prep &Q = CG.prep;
        F(Q.tmask); for (auto &p : Q.xmask) F(p);
        for (auto &p : Q.fcor) F(p);
        for (auto &p : Q.t) F(p);
        F(Q.tmass); F(Q.maskd); F(Q.c_dst); F(Q.c_src); F(Q.c_vsign); F(Q.hwater); F(Q.tbt); F(Q.aicen); F(Q.vicen);
    }
    F(CG.strengthU); F(CG.s12alt); F(CG.umaskd); F(CG.fac[0]); F(CG.fac[1]); F(CG.d_flags); F(CG.mask); F(CG.gmask); F(CG.mask4); F(CG.img_slot); F(CG.img_dst); F(CG.zero_cells); F(CG.fold_tmp);
    for (auto &f : CG.fold) { F(f.dst); F(f.a); F(f.b); F(f.flip); }
    for (auto &kv : CG.graphs) (void)hipGraphExecDestroy(kv.second);
    CG.graphs.clear();
    CG = CGridState();
}

static bool geo_derived();
static void fill(EvpCgrid &A)
{
    const cice_evp_hip_params &q = S.prm;
    for (int k = 0; k < CG_NF; ++k) A.f[k] = CG.f[k];
    for (int k = 0; k < CG_NIN; ++k) A.in[k] = CG.in[k];
    for (int k = 0; k < CG_NG; ++k) A.g[k] = CG.g[k];
    A.gmask = geo_derived() ? CG.gmask : nullptr;
    A.gstride = S.n;
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
        const char *e = env_test("CICE_EVP_HIP_CGRID_SPLIT");
        A.split_faces = e ? (std::atoi(e) != 0) : (S.n <= 600000);
    }
    {   // XCD-banded workgroup numbering (evp_cgrid.hip: cell()); CICE_EVP_HIP_CGRID_XCD=0: plain 2-D launch
        const bool on = !(env_test("CICE_EVP_HIP_CGRID_XCD") && !std::atoi(env_test("CICE_EVP_HIP_CGRID_XCD")));
        const int gy = (S.d.ny_block + 3) / 4;      // TY = 4 rows per workgroup
        const int gx = (S.d.nx_block + 63) / 64;    // TX = 64
        const int rows = env_test("CICE_EVP_HIP_CGRID_XCD") && std::atoi(env_test("CICE_EVP_HIP_CGRID_XCD")) > 1 ? std::atoi(env_test("CICE_EVP_HIP_CGRID_XCD")) : std::max(1, 256 / gx);
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

static bool one_launch();
static bool fused_schedule()
{
    if (CG.avg_strength && !one_launch()) return false;   // the three-launch kernels need deltaU at the neighbours: five phases
    if (CG.tripole) return false;                // recomputing a neighbour across the fold would sum in mirrored order
    return !(env_test("CICE_EVP_HIP_CGRID_FUSED") && !std::atoi(env_test("CICE_EVP_HIP_CGRID_FUSED")));
}

// The kernels push a cell into its ghost images only where there is ice; the reference's ice_HaloUpdate copies every
// cell.  The difference shows once per call: a cell WITHOUT ice whose ghost images hold something else than the cell
// itself -- a corner that lost its ice since the last step (dyn_prep2 zeroes stress12U on ilo..ihi x jlo..jhi only,
// iceUmask is never set on ghost cells: the ghost image keeps the old stress) -- is repaired by the reference's first
// exchange of the field.  Cells without ice do not change during a call, so one unconditional copy in the first
// subcycle, right after the launch that produces the field, leaves every ghost cell as the reference has it.
static void first_exchange_copies_everything(const EvpCgrid &A, std::initializer_list<int> fields)
{
    for (int f : fields) evp_launch_cgrid_phase(A, 9, f, S.stream);
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
        if (first && k == 0) first_exchange_copies_everything(A, {CF_SP, CF_SM});
        XCHG(A.f[CF_ETA], A.f[CF_ZETA]);
        XCHG(A.f[CF_SP], A.f[CF_SM]);
        fold({{A.f[CF_ZETA], 0, false}, {A.f[CF_ETA], 0, false}, {A.f[CF_SP], 0, false}, {A.f[CF_SM], 0, false}});
        evp_launch_cgrid_phase(A, 2, 1, S.stream);
        if (first && k == 0) first_exchange_copies_everything(A, {CF_S12U});
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

static int res_launch(const EvpCgrid &A, int nsub, bool dry, double *const cur5[5], double *const alt5[5]);
// tripole (u-fold) on one rank: the first subcycles as five launches + fold steps, the last nres inside ONE launch of the on-chip
// resident kernel's FOLD variant, then the fold step of everything that launch leaves (ghost row beyond the fold; the points ON the fold
// come out averaged already, and the step leaves an averaged pair as it is) and the velocity averages after the loop
static int enqueue_phases_resident(const EvpCgrid &A, int ndte, bool first, int nres);

// one launch per subcycle (cg_one) for every subcycle but the first after an upload (which still reads the caller's
// uvelN, vvelE, uvel, vvel): one rank, no fold.  Measured, us per subcycle, three launches -> one:
// gx3 12.5 -> 9.5, 300x240 17.6 -> 12.8, gx1 22.4 -> 17.6, 720x270 28.3 -> 21.9, 720x540 44.1 -> 39.3 (window shapes:
// build_one_tables), 1440x1080 196.6 -> 178.0, 3600x2400 1013 -> 902; avg_strength against its five launches:
// gx1 30.1 -> 18.6, 1440x1080 235 -> 191, 3600x2400 1257 -> 968.  CICE_EVP_HIP_CGRID_ONE=0 switches it off
// cg_one with 15 of its 23 static arrays derived in the kernel from the other eight (evp_cgrid.hip: DSlab): allowed when
// every identity holds bit for bit on the caller's arrays (derive_geometry_check, once per cice_evp_hip_cgrid_set_geometry).
// CICE_EVP_HIP_CGRID_GEO=0 keeps all 23 arrays in use.
static bool geo_derived()
{
    if (!CG.gmask) return false;
    if (const char *e = env_test("CICE_EVP_HIP_CGRID_GEO")) return std::atoi(e) != 0;
    return true;
}
// Checks, on every cell the kernels can read (the blocks' cells with their ghost ring; the ratios and nothing else need
// a neighbour: interior cells), that the caller's derived arrays are what the reference's start-up computes from dx / dy
// (the list: evp_cgrid.hip above DSlab).  Compared as BITS.  Returns the four masks as bits, or an empty vector + why.
// areas = false: the land masks and the boundary ratios only (what the on-chip resident kernel derives on a tripole grid).
static std::vector<uint8_t> derive_geometry_check(const double *const *g, std::string &why, bool areas = true)
{
    const int nxb = S.d.nx_block;
    std::vector<uint8_t> gm(S.n, 0);
    auto same = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0; };
    auto bad = [&](const char *what, int b, int i, int j) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s differs from the reference's start-up formula at block %d cell (%d, %d)", what, b, i, j);
        why = buf;
        return std::vector<uint8_t>();
    };
    const double dmin = S.prm.deltaminEVP;
    for (int b = 0; b < S.d.nblocks; ++b)
        for (int j = S.jlo[b] - 1; j <= S.jhi[b] + 1; ++j)
            for (int i = S.ilo[b] - 1; i <= S.ihi[b] + 1; ++i) {
                const size_t p = (size_t)b * S.plane + (size_t)(j - 1) * nxb + (i - 1);
                const double ta = g[CG_DXT][p] * g[CG_DYT][p], ua = g[CG_DXU][p] * g[CG_DYU][p];
                const double na = g[CG_DXN][p] * g[CG_DYN][p], ea = g[CG_DXE][p] * g[CG_DYE][p];
                if (areas) {
                    if (!same(ta, g[CG_TAREA][p])) return bad("tarea", b, i, j);
                    if (!same(ua, g[CG_UAREA][p])) return bad("uarea", b, i, j);
                    if (!same(na, g[CG_NAREA][p])) return bad("narea", b, i, j);
                    if (!same(ea, g[CG_EAREA][p])) return bad("earea", b, i, j);
                    if (!same(ea > 0.0 ? 1.0 / ea : 0.0, g[CG_EAREAR][p])) return bad("earear", b, i, j);
                    if (!same(na > 0.0 ? 1.0 / na : 0.0, g[CG_NAREAR][p])) return bad("narear", b, i, j);
                    if (!same(dmin * ta, g[CG_DMINT][p])) return bad("DminTarea", b, i, j);
                }
                unsigned bits = 0;
                const int mk[4] = {CG_EPM, CG_NPM, CG_UVM, CG_HM};
                for (int q = 0; q < 4; ++q) {
                    const double m = g[mk[q]][p];
                    if (same(m, 1.0)) bits |= 1u << q;
                    else if (!same(m, 0.0)) return bad("a land mask (neither 0 nor 1)", b, i, j);
                }
                gm[p] = (uint8_t)bits;
                if (i >= S.ilo[b] && i <= S.ihi[b] && j >= S.jlo[b] && j <= S.jhi[b]) {
                    const double rx = -(g[CG_DXN][p + 1] / g[CG_DXN][p]), ry = -(g[CG_DYE][p + nxb] / g[CG_DYE][p]);
                    if (!same(rx, g[CG_RXN][p]) || !same(1.0 / rx, g[CG_RXNR][p])) return bad("ratiodxN / ratiodxNr", b, i, j);
                    if (!same(ry, g[CG_RYE][p]) || !same(1.0 / ry, g[CG_RYER][p])) return bad("ratiodyE / ratiodyEr", b, i, j);
                    // (finite and negative: the kernels put -1 in their place wherever the factor next to them is +0)
                    if (!(rx < 0.0 && ry < 0.0 && std::isfinite(rx) && std::isfinite(ry) && std::isfinite(1.0 / rx) && std::isfinite(1.0 / ry)))
                        return bad("a boundary ratio (not finite and negative)", b, i, j);
                }
            }
    return gm;
}

static const int ONE_FIELDS[4] = {CF_UE, CF_VN, CF_SP, CF_SM};
static bool one_launch()
{
    if (!CG.one.tab || remote()) return false;
    if (const char *e = env("CICE_EVP_HIP_CGRID_ONE")) return std::atoi(e) != 0;
    return true;
}
static int one_subcycles(int ndte, bool first) { return one_launch() ? ndte - (first ? 1 : 0) : 0; }
static int res_launch(const EvpCgrid &A, int nsub, bool dry, double *const cur5[5], double *const alt5[5]);
static int build_res_tables(const double *const *static23);

// three launches per subcycle + one after the loop (evp_cgrid.hip); stress12U ping-pongs, returns with the
// current values in `cur` (the caller swaps the pointers when ndte is odd)
// nres > 0: the last nres subcycles of the call run inside ONE launch of the on-chip resident kernel (evp_cgrid_res.hip), which
// reads the current allocation of each ping-pong array and leaves the final state in both
static int enqueue_fused(EvpCgrid A, int ndte, bool first, int nres = 0)
{
    double *cur = CG.f[CF_S12U], *other = CG.s12alt;
    const bool one = one_launch();
    double *c4[4], *o4[4];
    for (int q = 0; q < 4; ++q) {
        c4[q] = CG.f[ONE_FIELDS[q]];
        o4[q] = CG.one.alt[q];
    }
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
    for (int k = 0; k < ndte - nres; ++k) {
        const int last = (k == ndte - 1);
        A.f[CF_S12U] = cur;
        if (one && !(first && k == 0)) {
            EvpCgOne T{CG.one.tab, CG.one.tiles, CG.one.ntiles, CG.one.per_xcd, CG.one.ox, CG.one.oy,
                       (env_test("CICE_EVP_HIP_CGRID_ONE_XCD") && !std::atoi(env_test("CICE_EVP_HIP_CGRID_ONE_XCD"))) ? 1 : 0, c4[0], c4[1], c4[2], c4[3], CG.gslab, CG.inslab, S.n,
                       CG.one.prof, geo_derived() ? CG.gmask : nullptr};
            for (int q = 0; q < 4; ++q) A.f[ONE_FIELDS[q]] = o4[q];
            A.s12_in = cur;
            A.f[CF_S12U] = other;
            if (CG.one.nitems > 0 && T.gmask && !(last && env_test("CICE_EVP_HIP_CGRID_STRIP_LAST") && !std::atoi(env_test("CICE_EVP_HIP_CGRID_STRIP_LAST")))) {
                // the interior of the blocks marched, the windows along their edges as before: both read the previous
                // subcycle's buffers only and own disjoint cells
                EvpCgStrip Z{CG.one.items, CG.one.nitems, ((CG.one.nitems + 3) / 4 + 7) / 8, CG.one.strip_len};
                EvpCgOne E = T;
                E.tab = CG.one.tab_e; E.tiles = CG.one.tiles_e; E.ntiles = CG.one.ntiles_e; E.per_xcd = (CG.one.ntiles_e + 7) / 8;
                E.ox = CG.one.ex; E.oy = CG.one.ey;
                E.prof = nullptr;
                // (the edge windows on the second stream beside the marched kernel: measured no gain, 565 us against 552 -- a
                // 1024-thread workgroup does not fit beside the marched kernel's waves on a CU anyway)
                // windows of 32 x 8 ride in the marched kernel's launch; other shapes (A/B) get a launch of their own behind it
                const bool ride = E.ox == 32 && E.oy == 8 && !(env_test("CICE_EVP_HIP_CGRID_STRIP_RIDE") && !std::atoi(env_test("CICE_EVP_HIP_CGRID_STRIP_RIDE")));
                evp_launch_cgrid_strip(A, T, Z, ride ? &E : nullptr, CG.fast ? 1 : 0, last, S.stream);
                if (!ride && E.ntiles > 0) evp_launch_cgrid_one(A, E, CG.fast ? 1 : 0, last, S.stream);
            } else if (T.ntiles > 0) {
                evp_launch_cgrid_one(A, T, CG.fast ? 1 : 0, last, S.stream);
            }
            std::swap(cur, other);
            for (int q = 0; q < 4; ++q) std::swap(c4[q], o4[q]);
            continue;
        }
        if (first && k == 0 && CG.avg_strength) {
            // (only with cg_one for the rest: fused_schedule) the first subcycle as the five launches, stress12U in place
            evp_launch_cgrid_phase(A, 0, 1, S.stream);
            evp_launch_cgrid_phase(A, 6, 1, S.stream);
            evp_launch_cgrid_phase(A, 1, 1, S.stream);
            first_exchange_copies_everything(A, {CF_SP, CF_SM});
            evp_launch_cgrid_phase(A, 2, 1, S.stream);
            first_exchange_copies_everything(A, {CF_S12U});
            evp_launch_cgrid_phase(A, 3, 1, S.stream);
            for (int q = 0; q < 4; ++q)
                HIPC(hipMemcpyAsync(o4[q], c4[q], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
            HIPC(hipMemcpyAsync(other, cur, S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
            continue;
        }
        if (first && k == 0) {
            evp_launch_cgrid_phase(A, 0, 1, S.stream);
            evp_launch_cgrid_phase(A, 6, 1, S.stream);
        } else {
            evp_launch_cgrid_phase(A, 7, last, S.stream);
        }
        XCHG(A.f[CF_SHEARU], A.f[CF_SHEARU]);
        evp_launch_cgrid_phase(A, 10, last, S.stream);
        if (first && k == 0) first_exchange_copies_everything(A, {CF_SP, CF_SM});   // (stress12U: above, before the loop)
        XCHG(A.f[CF_ETA], A.f[CF_ZETA]);         // (zetax2T is stored in the last subcycle only; harmless before)
        XCHG(A.f[CF_SP], A.f[CF_SM]);
        A.s12_in = cur;
        A.f[CF_S12U] = other;
        evp_launch_cgrid_phase(A, CG.fast ? 11 : 8, last, S.stream);
        XCHG(other, other);
        XCHG(A.f[CF_UE], A.f[CF_VN]);
        std::swap(cur, other);
        if (one && first && k == 0)          // cells no subcycle writes (no ice, ghost cells nothing is copied into): the same in both buffers
            for (int q = 0; q < 4; ++q)
                HIPC(hipMemcpyAsync(o4[q], c4[q], S.n * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    }
    A.f[CF_S12U] = cur;
    for (int q = 0; q < 4; ++q) A.f[ONE_FIELDS[q]] = c4[q];
    if (nres > 0) {
        double *cur5[5] = {c4[0], c4[1], c4[2], c4[3], cur};
        double *alt5[5] = {o4[0], o4[1], o4[2], o4[3], other};
        if (int rc = res_launch(A, nres, false, cur5, alt5)) return rc;
    }
    evp_launch_cgrid_phase(A, 4, 1, S.stream);
    XCHG(A.f[CF_UN], A.f[CF_VE]);
    XCHG(A.f[CF_UU], A.f[CF_VU]);
    return 0;
}

static int enqueue_phases_resident(const EvpCgrid &A, int ndte, bool first, int nres)
{
    if (ndte > nres)
        if (int rc = enqueue_phases(A, ndte - nres, first)) return rc;
    double *cur5[5] = {CG.f[CF_UE], CG.f[CF_VN], CG.f[CF_SP], CG.f[CF_SM], CG.f[CF_S12U]};
    double *alt5[5] = {CG.one.alt[0], CG.one.alt[1], CG.one.alt[2], CG.one.alt[3], CG.s12alt};
    if (int rc = res_launch(A, nres, false, cur5, alt5)) return rc;
    fold({{A.f[CF_SHEARU], 1, false}, {A.f[CF_S12U], 1, false}});
    fold({{A.f[CF_ZETA], 0, false}, {A.f[CF_ETA], 0, false}, {A.f[CF_SP], 0, false}, {A.f[CF_SM], 0, false}});
    fold({{A.f[CF_UE], 2, true}, {A.f[CF_VN], 3, true}});
    evp_launch_cgrid_phase(A, 4, 1, S.stream);
    fold({{A.f[CF_UN], 3, true}, {A.f[CF_VE], 2, true}, {A.f[CF_UU], 1, true}, {A.f[CF_VU], 1, true}});
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

// Window table of the one-launch kernel: for every position of every window the cell whose value the reference has
// there.  Within one cell of the owned range that is the array cell itself -- interior cell, or the interior cell a ghost
// cell mirrors (halo plan), or the ghost cell marked static (nothing is copied into it); further out the walk goes on
// from the mirrored cell, neighbour by neighbour.
static int build_one_tables(const double *const *static23)
{
    const HaloPlan &P = S.plan;
    // the window: 32x8 on the smallest grids (more workgroups than 64x8 gives), 64x8 where that gives every CU one or two
    // workgroups (gx1: 462), 64x16 from there on -- the fewest recomputed positions and re-read rows per owned cell, which is
    // what counts once there are several windows per CU (720x270: 24.6 / 21.9 us with 32x8 / 64x16; 3600x2400: 1083 / 1032 /
    // 925 with 32x8 / 64x8 / 64x16).  CICE_EVP_HIP_CGRID_ONE_SHAPE=0 / 1 / 2 picks one
    int shape = S.n > 160000 ? 2 : (S.n > 40000 ? 1 : 0);
    if (const char *e = env_test("CICE_EVP_HIP_CGRID_ONE_SHAPE")) shape = std::min(2, std::max(0, std::atoi(e)));
    const int OX = shape ? 64 : 32, OY = shape == 2 ? 16 : 8;
    // Order of the windows = order of the workgroups on an XCD (each XCD takes a contiguous run of the list): row by row.
    // (Strips of a few windows in x, top to bottom, so that vertical neighbours run together and share their 2-3 common rows
    // in L2, were measured: 3600 x 2400 1083 us row by row, 1171 / 1139 / 1106 / 1082 in strips of 4 / 8 / 16 / 32 windows --
    // narrow strips cost more in DRAM locality than the shared rows save.  CICE_EVP_HIP_CGRID_ONE_STRIP=<n> for A/B.)
    int strip = 1 << 20;
    if (const char *e = env_test("CICE_EVP_HIP_CGRID_ONE_STRIP")) strip = std::max(1, std::atoi(e));
    cice_evp_hip_dims d = S.d;
    d.ilo = S.ilo.data(); d.ihi = S.ihi.data(); d.jlo = S.jlo.data(); d.jhi = S.jhi.data();
    d.iglob0 = S.iglob0.data(); d.jglob0 = S.jglob0.data();
    std::vector<int32_t> tab, tiles;
    build_window_table(d, P, OX, OY, strip, tiles, tab);       // halo_plan.cpp (host only: CPU known-answer test)
    CGridState::One &O = CG.one;
    O.ntiles = (int)(tiles.size() / 4);
    O.ox = OX;
    O.oy = OY;
    O.per_xcd = (O.ntiles + 7) / 8;
    HIPC(hipMalloc((void **)&O.tab, tab.size() * sizeof(int)));
    HIPC(hipMalloc((void **)&O.tiles, tiles.size() * sizeof(int32_t)));
    HIPC(hipMemcpyAsync(O.tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemcpyAsync(O.tiles, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice, S.stream));
    for (auto &p : O.alt)
        if (alloc_d(&p, S.n)) return -1;
    if (env_test("CICE_EVP_HIP_CGRID_PROF") && std::atoi(env_test("CICE_EVP_HIP_CGRID_PROF"))) {
        HIPC(hipMalloc((void **)&O.prof, (size_t)O.ntiles * 8 * sizeof(unsigned long long)));
        HIPC(hipMemsetAsync(O.prof, 0, (size_t)O.ntiles * 8 * sizeof(unsigned long long), S.stream));
    }
    // ---- the marched kernel's share (cg_strip): per block the rectangle its regular windows cover, if they form one ----
    // default: large domains (the rectangle at least 300 000 cells and half of the rank's); CICE_EVP_HIP_CGRID_STRIP=0 / 1 (test
    // build) switches it off / on wherever a regular window exists, CICE_EVP_HIP_CGRID_STRIP_SEG=<rows> sets the segment length
    {
        int want = shape == 2 ? 2 : 0;              // 2: auto
        if (const char *e = env_test("CICE_EVP_HIP_CGRID_STRIP")) want = std::atoi(e) ? 1 : 0;
        long interior = 0;
        for (int b = 0; b < d.nblocks; ++b) interior += (long)(d.ihi[b] - d.ilo[b] + 1) * (d.jhi[b] - d.jlo[b] + 1);
        if (want == 2 && interior < 300000) want = 0;
        // the windows cg_one keeps beside the marched kernel -- a frame one window deep along the block's edges -- are cut
        // smaller than the ones it covers a whole domain with: 32 x 8 positions (29 x 5 owned), 256 threads, four workgroups per CU
        // in one round instead of two rounds of 1024-thread ones (3600 x 2400: the frame 51 us -> see DESIGN.md section 7)
        int eshape = 0;
        if (const char *e = env_test("CICE_EVP_HIP_CGRID_STRIP_EDGE")) eshape = std::min(2, std::max(0, std::atoi(e)));
        const int EX = eshape ? 64 : 32, EY = eshape == 2 ? 16 : 8;
        std::vector<int32_t> etiles, etab;
        if (want && (EX != OX || EY != OY)) {
            build_window_table(d, P, EX, EY, 1 << 20, etiles, etab);
            tiles.swap(etiles);
            tab.swap(etab);
        }
        const int nt = (int)(tiles.size() / 4), sx = EX - 3, sy = EY - 3;
        using Zone = StripZone;                        // (halo_plan.h: the planning is host-only code with a CPU test)
        std::vector<Zone> zones;
        long zcells = 0;
        std::vector<uint8_t> in_zone((size_t)nt, 0);
        if ((double)S.n * 8.0 * std::max((int)CG_NG, (int)CG_NIN) >= 4294967296.0) want = 0;     // (the kernel's 32-bit offsets into the tables)
        if (want) strip_zones(d, tiles, EX, EY, CG.h_img_slot.empty() ? nullptr : CG.h_img_slot.data(), zones);
        for (const Zone &z : zones) zcells += (long)(z.i1 - z.i0 + sx) * (z.j1 - z.j0 + sy);
        // (default: where the work items fill enough of the chip -- measured against cg_one alone: 720 x 270 23-48 us (segments of 4-16 rows)
        // against 22, 720 x 540 32 (8 rows) against 40, 1440 x 1080 95 against 154, 3600 x 2400 455-489 against 794)
        if (want == 2 && (2 * zcells < interior || zcells < 300000)) zones.clear();
        // The six lengths the reference's start-up forms from HTN (= dxN) and HTE (= dyE) -- dxU, dyU, dxT, dyT two-point means, dxE, dyN
        // four-point means (ice_grid.F90:3063-3280) -- checked BIT FOR BIT on every cell the marched kernel would form them for (each
        // rectangle with three columns and rows around it); where all hold the kernel forms them itself instead of loading them.  A
        // grid whose lengths were made otherwise (or a rectangle that reaches a row the reference extrapolates) keeps all eight loaded.
        bool lengths = !zones.empty() && static23 != nullptr;
        if (const char *e = env_test("CICE_EVP_HIP_CGRID_STRIP_LEN")) lengths = lengths && std::atoi(e) != 0;
        if (lengths) {
            auto same = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0; };
            const double *const *g = static23;
            const int nxb = d.nx_block, nyb = d.ny_block;
            auto holds = [&](const Zone &z) {
                const int ia = std::max(2, z.i0 - 3), ib = std::min(nxb - 1, z.i1 + sx - 1 + 2);
                const int ja = std::max(2, z.j0 - 2), jb = std::min(nyb - 1, z.j1 + sy - 1 + 2);
                const double *N = g[CG_DXN], *E = g[CG_DYE];
                for (int j = ja; j <= jb; ++j)
                    for (int i = ia; i <= ib; ++i) {
                        const size_t p0 = (size_t)z.b * nxb * nyb + (size_t)(j - 1) * nxb + (i - 1);
                        if (!(same(g[CG_DXU][p0], 0.5 * (N[p0] + N[p0 + 1])) && same(g[CG_DXT][p0], 0.5 * (N[p0] + N[p0 - nxb])) &&
                              same(g[CG_DXE][p0], 0.25 * (N[p0] + N[p0 + 1] + N[p0 - nxb] + N[p0 - nxb + 1])) &&
                              same(g[CG_DYU][p0], 0.5 * (E[p0] + E[p0 + nxb])) && same(g[CG_DYT][p0], 0.5 * (E[p0] + E[p0 - 1])) &&
                              same(g[CG_DYN][p0], 0.25 * (E[p0] + E[p0 - 1] + E[p0 + nxb] + E[p0 + nxb - 1]))))
                            return false;
                    }
                return true;
            };
            // (a rectangle whose outermost window row reaches a row the reference extrapolates -- j = 1, j = ny_global -- gives that
            // row of windows back to cg_one)
            std::vector<Zone> cut = zones;
            for (Zone &z : cut) {
                bool ok = false;
                for (int v = 0; v < 4 && !ok; ++v) {
                    Zone t = z;
                    if (v & 1) t.j1 -= sy;
                    if (v & 2) t.j0 += sy;
                    if (t.j1 < t.j0) continue;
                    if (holds(t)) { z = t; ok = true; }
                }
                lengths = lengths && ok;
            }
            if (lengths) {
                zones = cut;
                zcells = 0;
                for (const Zone &z : zones) zcells += (long)(z.i1 - z.i0 + sx) * (z.j1 - z.j0 + sy);
            }
        }
        if (!zones.empty()) {
            // strips of 60 owned columns (lanes 2 .. 61 of the wave; 59, lanes 3 .. 61, where the kernel forms the lengths; the last
            // strip of a rectangle is shifted west so that its lanes stay inside it and owns what is left); segments: about two
            // waves per SIMD resident at once over all strips (256 CUs x 8)
            // (measured, 3600 x 2400: 2006 items of 70 rows 567 us per subcycle; 2065 items -- 17 more than fit at once -- 693;
            // 2950 x 48 597, 4720 x 30 603, 11741 x 12 624: one round of work, as long as possible.  Shortest segment: 16 rows from a
            // million cells -- 1440 x 1080: 1608 items of 16 rows 95 us, 1992 of 13 104 --, 8 below -- 720 x 540: 804 items of 8 rows
            // 32 us, 408 of 16 49)
            long slots = 2048;
            if (const char *e = env_test("CICE_EVP_HIP_CGRID_STRIP_ITEMS")) slots = std::max(1, std::atoi(e));
            int seg_forced = 0;
            if (const char *e = env_test("CICE_EVP_HIP_CGRID_STRIP_SEG")) seg_forced = std::max(1, std::atoi(e));
            std::vector<int32_t> items;
            const int seg = strip_items(zones, EX, EY, lengths ? 3 : 2, slots, zcells >= 1000000 ? 16 : 8, seg_forced, items);
            strip_windows(zones, tiles, in_zone);
            std::vector<int32_t> tiles_e, tab_e;
            const size_t per = (size_t)EX * EY;
            for (int w = 0; w < nt; ++w)
                if (!in_zone[(size_t)w]) {
                    tiles_e.insert(tiles_e.end(), tiles.begin() + 4 * w, tiles.begin() + 4 * w + 4);
                    tab_e.insert(tab_e.end(), tab.begin() + (size_t)w * per, tab.begin() + (size_t)(w + 1) * per);
                }
            O.nitems = (int)(items.size() / 6);
            O.ntiles_e = (int)(tiles_e.size() / 4);
            O.ex = EX; O.ey = EY;
            O.strip_len = lengths ? 1 : 0;
            O.strip_seg = seg;
            O.strip_cells = zcells;
            HIPC(hipMalloc((void **)&O.items, items.size() * sizeof(int32_t)));
            HIPC(hipMemcpy(O.items, items.data(), items.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            if (O.ntiles_e) {
                HIPC(hipMalloc((void **)&O.tiles_e, tiles_e.size() * sizeof(int32_t)));
                HIPC(hipMalloc((void **)&O.tab_e, tab_e.size() * sizeof(int32_t)));
                HIPC(hipMemcpy(O.tiles_e, tiles_e.data(), tiles_e.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                HIPC(hipMemcpy(O.tab_e, tab_e.data(), tab_e.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            }
        }
    }
    HIPC(hipStreamSynchronize(S.stream));       // (the host vectors go out of scope)
    return 0;
}

// ---- the on-chip resident kernel (evp_cgrid_res.hip) ---------------------------------------------------------------
// Tables: cg_one's window table for 16 x 16 windows with one more row / column of positions, the map of cells some other
// window's rim mirrors (they publish a record every subcycle), the record buffers.  And the one property of the caller's
// static arrays the kernel relies on: it takes a NEIGHBOUR's operands from the cell the neighbouring position's value comes
// from, where the one-launch kernels read the array neighbour of the own cell -- the same thing if every ghost cell's static
// arrays are copies of its source's, which is checked here bit for bit on the sixteen arrays concerned (true of a grid whose
// static fields were halo-updated, as CICE's are; false, for instance, across a tripole fold).
static int build_res_tables(const double *const *static23)
{
    const HaloPlan &P = S.plan;
    CGridState::Res &Q = CG.res;
    Q.images_ok = true;
    const bool tripole = CG.tripole;
    for (size_t k = 0; k < P.local_dst.size() && Q.images_ok; ++k) {
        if (P.local_src[k] < 0) continue;
        if (tripole) {       // (the ghost row beyond the fold: by field location, below)
            const int db = (int)(P.local_dst[k] / S.plane);
            const int dj = (int)((P.local_dst[k] % S.plane) / S.d.nx_block) + 1;
            if (S.jglob0[db] + (dj - S.jlo[db]) > S.d.ny_global) continue;
        }
        // (the arrays the kernel reads at a NEIGHBOUR's position: the eight lengths, the four areas, the four land masks; the
        // reciprocal areas, DminTarea and the boundary ratios are read at the own cell only or not at all)
        for (int a : {CG_DXT, CG_DYT, CG_DXU, CG_DYU, CG_DXE, CG_DYE, CG_DXN, CG_DYN, CG_UAREA, CG_TAREA, CG_EAREA, CG_NAREA, CG_EPM, CG_NPM,
                      CG_UVM, CG_HM})
            if (std::memcmp(static23[a] + P.local_dst[k], static23[a] + P.local_src[k], sizeof(double)) != 0) {
                Q.images_ok = false;
                Q.why = "static array " + std::to_string(a) + " differs between ghost cell " + std::to_string(P.local_dst[k]) + " and its source";
                break;
            }
    }
    if (!Q.images_ok) return 0;
    cice_evp_hip_dims d = S.d;
    d.ilo = S.ilo.data(); d.ihi = S.ihi.data(); d.jlo = S.jlo.data(); d.jhi = S.jhi.data();
    d.iglob0 = S.iglob0.data(); d.jglob0 = S.jglob0.data();
    if (tripole) {
        // The kernel's FOLD variant takes the operands of a cell beyond the fold from the cell it mirrors: every ghost cell of the row
        // NY+1 must hold, array by array, what the cell its field location maps it to holds (true of a grid whose static fields went
        // through ice_HaloUpdate with their field_loc, as CICE's do).
        // (only the arrays the kernel reads THROUGH the remap: tarea, hm | uarea | dyE, earea, epm | dxN; everything else of a cell beyond
        // the fold is read from the array's own ghost row -- which need not be a mirror image: CICE computes e.g. dxE there from the ghost HTN)
        static const int by_loc[4][4] = {{CG_TAREA, CG_HM, -1, -1}, {CG_UAREA, -1, -1, -1}, {CG_DYE, CG_EAREA, CG_EPM, -1}, {CG_DXN, -1, -1, -1}};
        for (int loc = 0; loc < 4 && Q.images_ok; ++loc) {
            FoldList L;
            build_fold_list(d, loc, L);
            for (size_t k = 0; k < L.dst.size() && Q.images_ok; ++k) {
                if (L.b[k] != -1 || L.a[k] < 0 || L.a[k] == L.dst[k]) continue;      // (points ON the fold keep their own values)
                for (int a : by_loc[loc])
                    if (a >= 0 && std::memcmp(static23[a] + L.dst[k], static23[a] + L.a[k], sizeof(double)) != 0) {
                        Q.images_ok = false;
                        Q.why = "static array " + std::to_string(a) + " differs between the cell " + std::to_string(L.dst[k]) + " beyond the fold and the cell it mirrors";
                        break;
                    }
            }
        }
        if (!Q.images_ok) return 0;
    }
    constexpr int RX = 16, RY = 16, NPOS = (RX + 1) * (RY + 1);
    {   // a domain that can never be resident (3600 x 2400: 51k windows) gets no tables and no record buffers
        long nw = 0;
        for (int b = 0; b < S.d.nblocks; ++b)
            nw += (long)((S.ihi[b] - S.ilo[b] + RX - 3) / (RX - 3)) * ((S.jhi[b] - S.jlo[b] + RY - 3) / (RY - 3));
        if (nw > 16384) {     // (only the windows that hold ice have to be resident, per call: cg_res_live)
            Q.why = "more windows than can ever be resident at once (" + std::to_string(nw) + ")";
            return 0;
        }
    }
    std::vector<int32_t> tab, tiles, tiles2;
    if (tripole) {
        if (!build_fold_window_table(d, P, tiles, tiles2, tab, Q.why)) return 0;
        if ((long)tiles.size() / 4 > 16384) { Q.why = "more windows than can ever be resident at once"; return 0; }
        // the land masks as bits, the boundary ratios' identities (the five-phase kernels of these grids load all 23 arrays: no gmask yet)
        const std::vector<uint8_t> gm = derive_geometry_check(static23, Q.why, false);
        if (gm.empty()) return 0;
        HIPC(hipMalloc((void **)&Q.gmask, S.n));
        HIPC(hipMemcpy(Q.gmask, gm.data(), S.n, hipMemcpyHostToDevice));
        for (auto &p : CG.one.alt)
            if (!p && alloc_d(&p, S.n)) return -1;
    } else {
        build_window_table(d, P, RX, RY, 1 << 20, tiles, tab, 1);
    }
    Q.ntiles = (int)(tiles.size() / 4);
    // the last owned row of a window, the last row of positions that matter to its owned cells (fold windows: the fold row)
    auto jmax_of = [&](int w) { return tripole ? tiles[4 * w + 3] >> 16 : S.jhi[tiles[4 * w]]; };
    // who publishes what, and is every hand-off mutual?  (halo_plan.cpp: cgres_dependencies -- the rule the kernel's ring applies)
    static_assert(CGRES_REACH == EVP_CGRES_REACH && CGRES_SLOTS == EVP_CGRES_SLOTS, "halo_plan.h and evp_device.h must agree on the windows' ring");
    std::vector<uint8_t> pub;
    {
        int n_edges = 0, n_oneway = 0;
        const int unsafe = cgres_dependencies(d, tripole, tiles, tab, &pub, &n_edges, &n_oneway);
        if (unsafe > 0) {
            // a window that is read by one it cannot be held back by within three subcycles could overwrite the record slot that
            // reader still waits for: static ineligibility (the one-launch kernel runs such a cut)
            Q.why = std::to_string(unsafe) + " of " + std::to_string(n_edges) + " hand-offs between windows have no chain back within " +
                    std::to_string(CGRES_SLOTS - 1) + " subcycles";
            return 0;
        }
    }
    // A neighbour's operands and state come from the cell the table names for the NEIGHBOURING POSITION; the one-launch kernels
    // (and the reference) read the array neighbour of the position's own cell.  Inside the domain the two are the same cell or an
    // image of it.  Outside a closed boundary they can be two different ghost cells (every block keeps its own): the kernel is
    // only right if both hold the same values -- the static arrays are compared here, the loop's state at every upload.
    {
        std::vector<int2> pairs;
        std::set<std::pair<int, int>> seen;
        const int nxb = S.d.nx_block;
        for (int w = 0; w < Q.ntiles && Q.images_ok; ++w) {
            // (tripole: a fold window's rows up to the fold row, the others' up to two rows beyond the last owned one)
            const int tymax = !tripole ? RY : (tiles[4 * w + 3] & 1) ? ((tiles[4 * w + 3] >> 8) & 255) + 1 : std::min(RY, jmax_of(w) - tiles[4 * w + 2] + 5);
            for (int ty = 0; ty < tymax; ++ty)
                for (int tx = 0; tx < RX; ++tx) {
                    const int a = tab[(size_t)w * NPOS + ty * (RX + 1) + tx];
                    if (a < 0) continue;                  // a position outside the domain computes nothing
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if ((!dx && !dy) || tx + dx < 0 || ty + dy < 0 || ty + dy >= tymax) continue;
                            const int c = tab[(size_t)w * NPOS + (ty + dy) * (RX + 1) + (tx + dx)];
                            if (c >= 0) continue;         // inside the domain: the cell itself or an image of it
                            const int b = a + dx + dy * nxb, g = -1 - c;
                            if (b == g || !seen.insert({b, g}).second) continue;
                            pairs.push_back(make_int2(b, g));
                            for (int k : {CG_DXT, CG_DYT, CG_DXU, CG_DYU, CG_DXE, CG_DYE, CG_DXN, CG_DYN, CG_UAREA, CG_TAREA, CG_EAREA, CG_NAREA,
                                          CG_EPM, CG_NPM, CG_UVM, CG_HM})
                                if (std::memcmp(static23[k] + b, static23[k] + g, sizeof(double)) != 0) {
                                    Q.images_ok = false;
                                    Q.why = "static array " + std::to_string(k) + " differs between the ghost cells " + std::to_string(b) + " and " +
                                            std::to_string(g) + " outside the domain";
                                }
                        }
                }
        }
        if (!Q.images_ok) return 0;
        Q.npairs = (int)pairs.size();
        if (Q.npairs) {
            HIPC(hipMalloc((void **)&Q.pairs, pairs.size() * sizeof(int2)));
            HIPC(hipMemcpy(Q.pairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice));
        }
    }
    HIPC(hipMalloc((void **)&Q.tab, tab.size() * sizeof(int)));
    HIPC(hipMalloc((void **)&Q.tiles, tiles.size() * sizeof(int32_t)));
    HIPC(hipMalloc((void **)&Q.pubmap, S.n));
    HIPC(hipMalloc((void **)&Q.rec, (size_t)EVP_CGRES_SLOTS * S.n * 32));
    HIPC(hipMalloc((void **)&Q.err, 8 * sizeof(int)));
    HIPC(hipMemcpy(Q.tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(Q.tiles, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (tripole) {
        HIPC(hipMalloc((void **)&Q.tiles2, tiles2.size() * sizeof(int32_t)));
        HIPC(hipMemcpy(Q.tiles2, tiles2.data(), tiles2.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    HIPC(hipMemcpy(Q.pubmap, pub.data(), S.n, hipMemcpyHostToDevice));
    HIPC(hipMemset(Q.rec, 0, (size_t)EVP_CGRES_SLOTS * S.n * 32));
    HIPC(hipMemset(Q.err, 0, 8 * sizeof(int)));
    {   // until the first upload says otherwise: every window runs
        std::vector<int> ident(Q.ntiles);
        for (int w = 0; w < Q.ntiles; ++w) ident[w] = w;
        HIPC(hipMalloc((void **)&Q.d_order, ident.size() * sizeof(int)));
        HIPC(hipMalloc((void **)&Q.live_win, ident.size() * sizeof(int)));
        HIPC(hipMalloc((void **)&Q.live_cell, S.n));
        HIPC(hipMemcpy(Q.d_order, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPC(hipMemset(Q.live_cell, 1, S.n));
        Q.n_live = Q.ntiles;
    }
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, S.device));
    for (int v = 0; v < 8; ++v) Q.cap4[v] = (long)evp_cgrid_res_max_blocks_per_cu(v & 1, (v >> 1) & 1, tripole ? 1 : 0, v >> 2) * prop.multiProcessorCount;
    return 0;
}

// eligible in this call: one rank, no fold (either visc_method, classic or revised EVP), the default-configuration shortcuts hold on every ice cell, the static
// identities hold (the kernel takes -1 for a boundary ratio away from a coast), every window co-resident
static bool res_cull() { return !(env_test("CICE_EVP_HIP_CGRID_RES_CULL") && !std::atoi(env_test("CICE_EVP_HIP_CGRID_RES_CULL"))); }
// per_call (may be NULL): set when what stands in the way holds for THIS call only (the call's masks, operands, state)
static bool res_eligible(std::string *why = nullptr, bool *per_call = nullptr)
{
    auto no = [&](const char *w) { if (why) *why = w; return false; };
    const CGridState::Res &Q = CG.res;
    if (per_call) *per_call = false;
    if (CG.tripole) {
        // a u-fold on one rank: the kernel's FOLD variant (the first subcycle of a call runs as the five phases)
        if (CG.tfold || remote()) return no("a T-fold, or several ranks on a tripole grid");
    } else if (!CG.one.tab || remote() || !fused_schedule() || !one_launch()) {
        return no("several ranks, a block too small, or the one-launch schedule switched off");
    }
    if (!Q.tab) return no(Q.why.empty() ? "tables not built" : Q.why.c_str());
    if (CG.tripole ? !Q.gmask : !geo_derived()) return no("a start-up identity of the static arrays does not hold");
    if (per_call) *per_call = true;
    if (!Q.pairs_state_ok) return no("ghost cells outside the domain that the kernel treats as one position hold different state");
    if (Q.n_live < 1) return no("no window holds ice");
    // (seabed stress, waterx / watery other than the ocean currents, rheofact != 1 on some ice cell: the kernel's SLOW variant, round 6)
    if ((long)(res_cull() ? Q.n_live : Q.ntiles) > Q.cap4[(CG.avg_strength ? 1 : 0) | (S.prm.revp != 0.0 ? 2 : 0) | (CG.fast ? 0 : 4)]) return no("more windows with ice than can be resident at once");
    return true;
}

static int res_launch(const EvpCgrid &A, int nsub, bool dry, double *const cur5[5], double *const alt5[5])
{
    CGridState::Res &Q = CG.res;
    EvpCgRes R{};
    R.tab = Q.tab; R.tiles = Q.tiles; R.order = nullptr; R.ntiles = Q.ntiles;
    R.tiles2 = Q.tiles2; R.fold = CG.tripole ? 1 : 0;
    R.slow = CG.fast ? 0 : 1;
    R.long_sleep = env_test("CICE_EVP_HIP_CGRID_RES_SLEEP") && std::atoi(env_test("CICE_EVP_HIP_CGRID_RES_SLEEP")) ? 1 : 0;
    R.dbg = env_test("CICE_EVP_HIP_CGRID_RES_DEBUG") ? std::atoi(env_test("CICE_EVP_HIP_CGRID_RES_DEBUG")) : 0;
    // the windows that hold ice in this call (finish_upload: cg_res_live); CICE_EVP_HIP_CGRID_RES_CULL=0 (test build) runs them all
    if (res_cull()) {
        R.order = Q.d_order; R.ntiles = Q.n_live; R.live = Q.live_cell;
    }
    R.nsub = nsub; R.dry = dry ? 1 : 0;
    Q.epoch = (Q.epoch + 1u) & 0xFFFFFu;
    if (Q.epoch == 0) Q.epoch = 1;
    R.tag_base = Q.epoch << 12;
    R.par0 = Q.par;
    Q.par = (Q.par + nsub) & (EVP_CGRES_SLOTS - 1);
    R.spin_limit = (R.dbg & 16) ? 200000u : 4000000u;       // (the fault test need not wait seconds for the bound)
    R.err = Q.err;
    R.pubmap = Q.pubmap;
    for (int k = 0; k < EVP_CGRES_SLOTS; ++k) R.rec[k] = (char *)Q.rec + (size_t)k * S.n * 32;
    R.uE_in = cur5[0]; R.vN_in = cur5[1]; R.sp_in = cur5[2]; R.sm_in = cur5[3]; R.s12_in = cur5[4];
    R.uE_out[0] = cur5[0]; R.uE_out[1] = alt5[0]; R.vN_out[0] = cur5[1]; R.vN_out[1] = alt5[1];
    R.sp_out[0] = cur5[2]; R.sp_out[1] = alt5[2]; R.sm_out[0] = cur5[3]; R.sm_out[1] = alt5[3];
    R.s12_out[0] = cur5[4]; R.s12_out[1] = alt5[4];
    R.gbase = CG.gslab; R.inbase = CG.inslab; R.stride = S.n;
    R.gmask = CG.tripole ? Q.gmask : CG.gmask;
    if (!dry && !Q.prof && env_test("CICE_EVP_HIP_CGRID_PROF") && std::atoi(env_test("CICE_EVP_HIP_CGRID_PROF")))
        HIPC(hipMalloc((void **)&Q.prof, (size_t)Q.ntiles * 32 * sizeof(unsigned long long)));
    R.prof = dry ? nullptr : Q.prof;
    evp_launch_cgrid_res(A, R, S.stream);
    HIPC(hipGetLastError());
    Q.launched = true;
    return 0;
}

static int res_check_error()
{
    CGridState::Res &Q = CG.res;
    if (!Q.launched) return 0;
    Q.launched = false;
    int ev[8] = {0};
    HIPC(hipMemcpy(ev, Q.err, sizeof ev, hipMemcpyDeviceToHost));
    if (ev[0]) {
        HIPC(hipMemset(Q.err, 0, sizeof ev));
        Q.mode = 0;
        return fail(-7, "resident C-grid kernel: a wait gave up (window %d, subcycle %d, cell %d, tag seen %#x, wanted %#x) -- workgroups not co-resident?",
                    ev[1], ev[2], ev[3], (unsigned)ev[4], (unsigned)ev[5]);
    }
    return 0;
}

// First eligible call: a dry run (the same work on the uploaded state, nothing written back) proves that every window is
// resident and gives the steady-state cost per subcycle.  CICE_EVP_HIP_CGRID_RESIDENT=0 / 1 forces the verdict.
static int res_decide(const EvpCgrid &A)
{
    CGridState::Res &Q = CG.res;
    if (Q.mode >= 0) return 0;
    int want = -1;
    if (const char *e = env("CICE_EVP_HIP_CGRID_RESIDENT")) want = std::atoi(e);
    std::string why;
    bool per_call = false;
    if (want == 0 || !res_eligible(&why, &per_call)) {
        // forced on: only what can never change (tables, geometry, rank layout) is an error; a condition of this call's masks,
        // operands or state is a fall-back for this call, as it is once the kernel has run (res_subcycles)
        if (want == 1 && !per_call) return fail(-6, "resident C-grid kernel requested but not applicable: %s", why.c_str());
        if (env("CICE_EVP_HIP_VERBOSE") && want != 0) std::fprintf(stderr, "[cice_evp_hip] C grid: on-chip resident kernel not used: %s\n", why.c_str());
        // (per-call conditions -- visc_method, the shortcuts -- may hold in a later call: stay undecided unless switched off)
        if (want == 0) Q.mode = 0;
        return 0;
    }
    if (want == 1) {       // forced on: no probe launches (profiles see real launches only); a window that is not resident shows at the next sync
        Q.mode = 1;
        return 0;
    }
    double *cur5[5] = {CG.f[CF_UE], CG.f[CF_VN], CG.f[CF_SP], CG.f[CF_SM], CG.f[CF_S12U]};
    double *alt5[5] = {CG.one.alt[0], CG.one.alt[1], CG.one.alt[2], CG.one.alt[3], CG.s12alt};
    const int nshort = 8, nlong = 40;
    float tl[2] = {0, 0}, ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        const int np = rep == 2 ? nlong : nshort;
        HIPC(hipEventRecord(S.ev2, S.stream));
        if (int rc = res_launch(A, np, true, cur5, alt5)) return rc;
        HIPC(hipEventRecord(S.ev3, S.stream));
        HIPC(hipStreamSynchronize(S.stream));
        if (res_check_error()) {
            if (want == 1) return -7;
            g_err.clear();
            Q.mode = 0;
            if (env("CICE_EVP_HIP_VERBOSE")) std::fprintf(stderr, "[cice_evp_hip] C grid: on-chip resident kernel failed its probe, not used\n");
            return 0;
        }
        HIPC(hipEventElapsedTime(&ms, S.ev2, S.ev3));
        if (rep >= 1) tl[rep - 1] = ms;
    }
    Q.t_probe_ms = (tl[1] - tl[0]) / (nlong - nshort);
    Q.mode = 1;
    return 0;
}

static int res_subcycles(int ndte, bool first)
{
    if (CG.res.mode != 1 || (!CG.tripole && !one_launch()) || !res_eligible()) return 0;
    const int n = ndte - (first ? 1 : 0);
    return (n >= 3 && n <= 4000) ? n : 0;
}

int finish_upload(int32_t visc_method);

}  // namespace evp_host

using namespace evp_host;

extern "C" {

int cice_evp_hip_cgrid_set_geometry(const double *const *static23)
{
    if (!S.ready) return fail(-1, "not initialised");
    if (!static23) return fail(-1, "null argument");
    const HaloPlan &P = S.plan;
    // (tripoleT: the same five launches + fold steps; only the fold lists differ -- all four locations rewrite the top
    // physical row there, halo_plan.cpp: build_fold_list_tfold)
    const bool tfold = S.d.ns_boundary_type == CICE_EVP_BND_TRIPOLET;
    const bool tripole = S.d.ns_boundary_type == CICE_EVP_BND_TRIPOLE || tfold;
    if (tripole && P.fold_rows == 2)
        return fail(-4, "C-grid EVP on a tripole grid: the blocks next to the fold (rows NY-1, NY; tripoleT: NY-2 .. NY) must all "
                        "be on one rank (split the domain in y only); here they are shared with other ranks");
    cgrid_free();
    CG.tripole = tripole;
    CG.tfold = tfold;
    for (auto &p : CG.f)
        if (alloc_d(&p, S.n)) return -1;
    if (alloc_d(&CG.inslab, (size_t)CG_NIN * S.n) || alloc_d(&CG.gslab, (size_t)CG_NG * S.n)) return -1;
    for (int k = 0; k < CG_NIN; ++k) CG.in[k] = CG.inslab + (size_t)k * S.n;
    for (int k = 0; k < CG_NG; ++k) {
        if (!static23[k]) return fail(-1, "null static array %d", k);
        CG.g[k] = CG.gslab + (size_t)k * S.n;
        if (h2d(CG.g[k], static23[k])) return -1;
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
            if (S.jglob0[db] + (dj - S.jlo[db]) > S.d.ny_global - (tfold ? 1 : 0)) continue;    // (tripoleT: the top physical row too)
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
    if (!tripole && S.plan.peers.empty() && S.d.nx_block >= 3 && S.d.ny_block >= 3) {
        if (int rc = build_one_tables(static23)) return rc;
        if (int rc = build_res_tables(static23)) return rc;
    }
    if (tripole && !tfold && S.plan.peers.empty() && P.fold_rows == 1 && S.d.ew_boundary_type == CICE_EVP_BND_CYCLIC)
        if (int rc = build_res_tables(static23)) return rc;
    if (!tripole) {      // (the five-phase kernels of tripole grids always load all 23)
        const std::vector<uint8_t> gm = derive_geometry_check(static23, CG.geo_why);
        if (!gm.empty()) {
            HIPC(hipMalloc((void **)&CG.gmask, S.n));
            HIPC(hipMemcpy(CG.gmask, gm.data(), S.n, hipMemcpyHostToDevice));
        } else if (env("CICE_EVP_HIP_VERBOSE")) {
            std::fprintf(stderr, "[cice_evp_hip] C grid: all 23 static arrays stay in use: %s\n", CG.geo_why.c_str());
        }
    }
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
    return finish_upload(visc_method);
}

}  // extern "C"

namespace evp_host {
// what follows the arrival of the loop's inputs, however they arrived (cice_evp_hip_cgrid_upload: from the host;
// cice_evp_hip_cgrid_prep: computed here)
int finish_upload(int32_t visc_method)
{
    // evp() zeroes its work arrays at entry (ice_dyn_evp.F90:351-361)
    for (int k = CF_ZETA; k < CG_NF; ++k) HIPC(hipMemsetAsync(CG.f[k], 0, S.n * sizeof(double), S.stream));
    CG.avg_strength = visc_method;
    unsigned h_flags = ~0u;
    {
        EvpCgrid A;
        fill(A);
        HIPC(hipMemsetAsync(CG.d_flags, 0, sizeof(unsigned), S.stream));
        evp_launch_cgrid_call_setup(A, CG.fac[0], CG.fac[1], CG.d_flags, S.stream);
        if (CG.res.npairs) {
            const double *five[5] = {CG.f[CF_UE], CG.f[CF_VN], CG.f[CF_SP], CG.f[CF_SM], CG.f[CF_S12U]};
            evp_launch_cgrid_res_pair_check(five, CG.res.pairs, CG.res.npairs, CG.d_flags, S.stream);
        }
        HIPC(hipMemcpyAsync(&h_flags, CG.d_flags, sizeof(unsigned), hipMemcpyDeviceToHost, S.stream));
        if (CG.res.tab) {        // which windows of the resident kernel hold ice in this call
            CGridState::Res &Q = CG.res;
            HIPC(hipMemsetAsync(Q.live_cell, 0, S.n, S.stream));
            evp_launch_cgrid_res_live(A, Q.tab, Q.tiles, Q.ntiles, CG.tripole ? 1 : 0, Q.live_win, Q.live_cell, S.stream);
            Q.h_live.resize(Q.ntiles);
            HIPC(hipMemcpyAsync(Q.h_live.data(), Q.live_win, (size_t)Q.ntiles * sizeof(int), hipMemcpyDeviceToHost, S.stream));
        }
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
    CG.fast = (h_flags & 255u) == 0 && !(env_test("CICE_EVP_HIP_CGRID_FAST") && !std::atoi(env_test("CICE_EVP_HIP_CGRID_FAST")));
    CG.res.pairs_state_ok = (h_flags & 256u) == 0;
    if (CG.res.tab) {
        CGridState::Res &Q = CG.res;
        std::vector<int> ord;
        for (int w = 0; w < Q.ntiles; ++w)
            if (Q.h_live[w]) ord.push_back(w);
        Q.n_live = (int)ord.size();
        if (Q.n_live) HIPC(hipMemcpy(Q.d_order, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    CG.uploaded = true;
    CG.first = true;
    return 0;
}
}  // namespace evp_host

extern "C" {

int cice_evp_hip_cgrid_subcycle(int32_t ndte)
{
    if (!CG.uploaded) return fail(-1, "C-grid EVP: nothing uploaded");
    if (ndte < 0) return fail(-1, "ndte < 0");
    if (ndte == 0) return 0;
    EvpCgrid A;
    fill(A);
    const bool fused = fused_schedule();
    if (int rc = res_check_error()) return rc;
    if (int rc = res_decide(A)) return rc;       // (first call: the on-chip resident kernel's probe; refuses loudly when forced on a rank it cannot serve)
    const int nres = (fused || CG.tripole) ? res_subcycles(ndte, CG.first) : 0;
    auto enqueue = [&]() -> int {
        if (fused) return enqueue_fused(A, ndte, CG.first, nres);
        if (nres > 0) return enqueue_phases_resident(A, ndte, CG.first, nres);
        return enqueue_phases(A, ndte, CG.first);
    };
    HIPC(hipEventRecord(S.ev0, S.stream));
    // (the resident launch carries a fresh epoch in its arguments: enqueued eagerly, with the few launches around it)
    if (nres == 0 && S.use_graph && (!remote() || S.direct.on)) {     // RCCL point-to-point is enqueued eagerly (as the B-grid loop does)
        const std::pair<int, int> key(ndte, (geo_derived() ? 128 : 0) | (fused && one_launch() ? 64 : 0) | (CG.one.flip << 5) | (CG.fast ? 16 : 0) | (CG.flip << 3) |
                                                (fused ? 4 : 0) | (CG.first ? 2 : 0) | CG.avg_strength);
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
    if (fused && ((ndte - nres - ((CG.first && CG.avg_strength) ? 1 : 0)) & 1)) {   // the current stress12U is in the other allocation now
        // (every subcycle swaps the two, except a first one run as five launches: visc_method = avg_strength)
        std::swap(CG.f[CF_S12U], CG.s12alt);
        CG.flip ^= 1;
    }
    if (fused && ((one_subcycles(ndte, CG.first) - nres) & 1)) {      // and so are uvelE, vvelN, stresspT, stressmT
        for (int q = 0; q < 4; ++q) std::swap(CG.f[ONE_FIELDS[q]], CG.one.alt[q]);
        CG.one.flip ^= 1;
    }
    HIPC(hipEventRecord(S.ev1, S.stream));
    HIPC(hipGetLastError());
    CG.t_one = fused ? one_subcycles(ndte, CG.first) - nres : 0;
    CG.res.last_nsub = nres;
    CG.first = false;
    CG.t_nsub = ndte;
    return 0;
}

int cice_evp_hip_cgrid_download(double *const *fields19)
{
    if (!CG.uploaded) return fail(-1, "C-grid EVP: nothing uploaded");
    if (!fields19) return fail(-1, "null argument");
    if (CG.res.launched) {       // a resident launch whose waits gave up has written nothing back: the caller's arrays stay untouched
        HIPC(hipStreamSynchronize(S.stream));
        if (int rc = res_check_error()) return rc;
    }
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

// dyn_finish at N and E points (ice_dyn_shared.F90:1291-1365; evp() calls it for both after the C-grid loop,
// ice_dyn_evp.F90:1408-1436) on the device, from the loop's resident final face velocities and the per-call operands the
// loop already holds (cdn_ocnX, aiX, uocnX, vocnX, fmX): the four arrays are inout -- written on the cells of dyn_prep2's
// N / E lists, every other cell keeps the caller's value.
int cice_evp_hip_cgrid_dyn_finish(double *strocnxN, double *strocnyN, double *strocnxE, double *strocnyE)
{
    if (!CG.uploaded) return fail(-1, "C-grid EVP: nothing uploaded");
    double *host[4] = {strocnxN, strocnyN, strocnxE, strocnyE};
    for (double *p : host)
        if (!p) return fail(-1, "null argument");
    CopyBatch U;
    for (int k = 0; k < 4; ++k) {
        if (!CG.post[k] && alloc_d(&CG.post[k], S.n)) return -1;
        U.items.push_back({CG.post[k], host[k]});
    }
    if (h2d_batch(U)) return -1;
    EvpCgrid A;
    fill(A);
    evp_launch_cgrid_dyn_finish(A, 0, CG.post[0], CG.post[1], S.stream);
    evp_launch_cgrid_dyn_finish(A, 1, CG.post[2], CG.post[3], S.stream);
    HIPC(hipGetLastError());
    CopyBatch D;
    for (int k = 0; k < 4; ++k) D.items.push_back({host[k], CG.post[k]});
    if (d2h_batch(D)) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

int cice_evp_hip_cgrid_sync(void)
{
    HIPC(hipStreamSynchronize(S.stream));
    float ms = 0;
    if (CG.t_nsub && hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess) CG.t_loop_ms = ms;
    return res_check_error();
}

int cice_evp_hip_cgrid_run(int32_t ndte, int32_t visc_method, double *const *fields19, const double *const *inputs23,
                           const int32_t *iceTmask, const int32_t *iceUmask, const int32_t *iceEmask,
                           const int32_t *iceNmask)
{
    if (cice_evp_hip_cgrid_upload(fields19, inputs23, iceTmask, iceUmask, iceEmask, iceNmask, visc_method)) return -1;
    if (cice_evp_hip_cgrid_subcycle(ndte)) return -1;
    const int rc = cice_evp_hip_cgrid_download(fields19);
    if (rc == -7) {
        // The on-chip resident kernel gave up (a window was not resident: the GPU is not this rank's alone).  Its waits are
        // bounded, a launch that gives up writes nothing back and the download has not touched the caller's arrays: the call
        // is repeated from them with the per-subcycle kernels, which later calls use too (the verdict is kept).
        ++CG.res.fallbacks;
        if (env("CICE_EVP_HIP_VERBOSE")) std::fprintf(stderr, "[cice_evp_hip] C grid: %s -- repeating the call without the resident kernel\n", g_err.c_str());
        g_err.clear();
        if (cice_evp_hip_cgrid_upload(fields19, inputs23, iceTmask, iceUmask, iceEmask, iceNmask, visc_method)) return -1;
        if (cice_evp_hip_cgrid_subcycle(ndte)) return -1;
        return cice_evp_hip_cgrid_download(fields19);
    }
    return rc;
}

// ---- the preparation phase on the device (SURVEY 8 f-2 for grid_ice = 'C'; kernels: evp_prep.hip prep1 / halo_center,
// evp_cgrid_prep.hip, evp_cgrid.hip cg_average) ----------------------------------------------------------------------
int cice_evp_hip_cgrid_set_prep_geometry(const int32_t *tmask, const int32_t *umaskCD, const int32_t *emask,
                                         const int32_t *nmask, const double *fcor_blk, const double *fcorE_blk,
                                         const double *fcorN_blk)
{
    if (!S.ready || !CG.geo) return fail(-1, "C-grid EVP: geometry not set");
    if (!tmask || !umaskCD || !emask || !nmask || !fcor_blk || !fcorE_blk || !fcorN_blk) return fail(-1, "null argument");
    if (S.plan.center_fold_remote || (S.plan.tfold && S.plan.center_tf_remote))
        return fail(-9, "device preparation: T-grid ghost cells across the tripole fold live on other ranks here; keep "
                        "evp()'s host preparation (cice_evp_hip_cgrid_run) on this configuration");
    CGridState::Prep &Q = CG.prep;
    std::vector<uint8_t> h8(S.n);
    auto B = [&](uint8_t *&p, const int32_t *src) -> int {
        if (!p) HIPC(hipMalloc((void **)&p, S.n));
        for (size_t k = 0; k < S.n; ++k) h8[k] = src[k] != 0;
        HIPC(hipMemcpy(p, h8.data(), S.n, hipMemcpyHostToDevice));
        return 0;
    };
    if (B(Q.tmask, tmask) || B(Q.xmask[0], umaskCD) || B(Q.xmask[1], emask) || B(Q.xmask[2], nmask)) return -1;
    const double *fc[3] = {fcor_blk, fcorE_blk, fcorN_blk};
    for (int k = 0; k < 3; ++k)
        if ((!Q.fcor[k] && alloc_d(&Q.fcor[k], S.n)) || h2d(Q.fcor[k], fc[k])) return -1;
    for (auto &p : Q.t)
        if (!p && alloc_d(&p, S.n)) return -1;
    if ((!Q.tmass && alloc_d(&Q.tmass, S.n)) || (!Q.maskd && alloc_d(&Q.maskd, S.n))) return -1;
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
    // the loop's inputs persist between calls as the reference's module arrays do (a cell off the ice keeps e.g. its fmE)
    for (auto &p : CG.in) HIPC(hipMemsetAsync(p, 0, S.n * sizeof(double), S.stream));
    HIPC(hipMemsetAsync(CG.mask4, 0, 4 * S.n * sizeof(int32_t), S.stream));
    HIPC(hipStreamSynchronize(S.stream));
    Q.geo = true;
    return 0;
}

static void fill_prep(EvpCgPrep &P)
{
    CGridState::Prep &Q = CG.prep;
    P.nx = S.d.nx_block; P.ny = S.d.ny_block; P.plane = S.plane; P.n = S.n; P.blk = S.blk;
    for (int k = 0; k < 11; ++k) P.t[k] = Q.t[k];
    P.tmass = Q.tmass; P.maskd = Q.maskd;
    P.hm = CG.g[CG_HM]; P.tarea = CG.g[CG_TAREA]; P.uarea = CG.g[CG_UAREA]; P.earea = CG.g[CG_EAREA]; P.narea = CG.g[CG_NAREA];
    for (int k = 0; k < 3; ++k) { P.xmask[k] = Q.xmask[k]; P.fcor[k] = Q.fcor[k]; }
    P.m4 = CG.mask4;
    for (int k = 0; k < 14; ++k) P.f[k] = CG.f[k];
    for (int k = 0; k < CG_NIN; ++k) P.in[k] = CG.in[k];
    P.dt = Q.pp.dt; P.gravit = Q.pp.gravit; P.dyn_area_min = Q.pp.dyn_area_min; P.dyn_mass_min = Q.pp.dyn_mass_min;
    P.cosw = S.prm.cosw; P.sinw = S.prm.sinw; P.ssh_coupled = Q.pp.ssh_stress_coupled;
}

int cice_evp_hip_cgrid_prep(const cice_evp_hip_prep_params *pp, const double *const *tfields11,
                            const double *const *state12, int32_t *iceTmask, int32_t *iceUmask, int32_t *iceEmask,
                            int32_t *iceNmask)
{
    if (!S.ready || !CG.geo) return fail(-1, "C-grid EVP: geometry not set");
    CGridState::Prep &Q = CG.prep;
    if (!Q.geo) return fail(-1, "cice_evp_hip_cgrid_set_prep_geometry was not called");
    if (!pp || !tfields11 || !iceTmask || !iceUmask || !iceEmask || !iceNmask) return fail(-1, "null argument");
    for (int k = 0; k < 11; ++k)
        if (!tfields11[k]) return fail(-1, "null T-grid field %d", k);
    // the loop's state: given, or NULL = what the previous call left on the device (evp() is its only writer)
    int nstate = 0;
    if (state12)
        for (int k = 0; k < 12; ++k) nstate += state12[k] != nullptr;
    if (nstate != 0 && nstate != 12) return fail(-1, "state arrays: give all 12 or none");
    if (nstate == 0 && !CG.uploaded) return fail(-1, "no C-grid state on the device yet: the first call must upload it");
    Q.pp = *pp;
    HIPC(hipEventRecord(S.ev2, S.stream));
    CopyBatch B;
    for (int k = 0; k < 11; ++k) B.items.push_back({Q.t[k], tfields11[k]});
    if (nstate)
        for (int k = 0; k < 12; ++k) B.items.push_back({CG.f[k], state12[k]});
    if (h2d_batch(B)) return -1;
    {   // the previous call's ice masks at U, E, N points (ghost cells included: they travel back unchanged)
        const int32_t *m[3] = {iceUmask, iceEmask, iceNmask};
        for (int k = 0; k < 3; ++k)
            HIPC(hipMemcpyAsync(CG.mask4 + (size_t)(k + 1) * S.n, m[k], S.n * sizeof(int32_t), hipMemcpyHostToDevice, S.stream));
    }
    HIPC(hipEventRecord(S.ev3, S.stream));
    {   // dyn_prep1 and the T-grid halo updates: the B-grid preparation's kernels
        EvpPrep P{};
        P.nx = S.d.nx_block; P.ny = S.d.ny_block; P.plane = S.plane; P.blk = S.blk;
        P.tmask = Q.tmask;
        for (int k = 0; k < 11; ++k) P.t[k] = Q.t[k];
        P.tmass = Q.tmass; P.maskd = Q.maskd;
        P.rhoi = pp->rhoi; P.rhos = pp->rhos; P.dyn_area_min = pp->dyn_area_min; P.dyn_mass_min = pp->dyn_mass_min;
        evp_launch_prep1(P, S.d.nblocks, S.stream);
        EvpPrepHalo H{};
        std::pair<double *, bool> arrs[10] = {{Q.maskd, false}, {Q.tmass, false}, {Q.t[3], false}, {Q.t[4], false}, {Q.t[5], true},
                                              {Q.t[6], true}, {Q.t[7], true}, {Q.t[8], true}, {Q.t[9], true}, {Q.t[10], true}};
        for (const auto &a : arrs) { H.a[H.narr] = a.first; H.is_vec[H.narr] = a.second; ++H.narr; }
        H.dst = Q.c_dst; H.src = Q.c_src; H.vsign = Q.c_vsign; H.n = Q.n_center;
        evp_launch_halo_center(H, S.stream);
        if (CG.tfold) {
            // tripoleT: the centre rule rewrites the top physical row (made symmetric, then mirrored) and fills the ghost row
            // from row NY-1: the fold step of location 0, after the plain ghost copies (whose sources it does not write)
            fold({{arrs[0].first, 0, arrs[0].second}, {arrs[1].first, 0, arrs[1].second}, {arrs[2].first, 0, arrs[2].second}, {arrs[3].first, 0, arrs[3].second}});
            fold({{arrs[4].first, 0, arrs[4].second}, {arrs[5].first, 0, arrs[5].second}, {arrs[6].first, 0, arrs[6].second}, {arrs[7].first, 0, arrs[7].second}});
            fold({{arrs[8].first, 0, arrs[8].second}, {arrs[9].first, 0, arrs[9].second}});
        }
        if (S.plan.center_remote) {
            double *pairs[5][2] = {{Q.maskd, Q.tmass}, {Q.t[3], Q.t[4]}, {Q.t[5], Q.t[6]}, {Q.t[7], Q.t[8]}, {Q.t[9], Q.t[10]}};
            for (auto &pr : pairs)
                if (int rc = halo_remote_pair(pr[0], pr[1])) return rc;
        }
    }
    EvpCgPrep P{};
    fill_prep(P);
    evp_launch_cgrid_prep(P, S.d.nblocks, S.stream);
    EvpCgrid A;
    fill(A);
    evp_launch_cgrid_mask(A, CG.mask4, S.stream);
    // ice_dyn_evp.F90:703-731: exchange uvelE, vvelN; the other component at each face and the corner velocities; exchange
    // them -- the tail of a subcycle (phase 4), on every cell (nothing is masked yet)
    evp_launch_cgrid_phase(A, 9, CF_UE, S.stream);
    evp_launch_cgrid_phase(A, 9, CF_VN, S.stream);
    if (remote())
        if (int rc = halo_remote_pair(A.f[CF_UE], A.f[CF_VN])) return rc;
    fold({{A.f[CF_UE], 2, true}, {A.f[CF_VN], 3, true}});
    evp_launch_cgrid_phase(A, 6, 1, S.stream);
    evp_launch_cgrid_phase(A, 4, 1, S.stream);
    if (remote()) {
        if (int rc = halo_remote_pair(A.f[CF_UN], A.f[CF_VE])) return rc;
        if (int rc = halo_remote_pair(A.f[CF_UU], A.f[CF_VU])) return rc;
    }
    fold({{A.f[CF_UN], 3, true}, {A.f[CF_VE], 2, true}, {A.f[CF_UU], 1, true}, {A.f[CF_VU], 1, true}});
    HIPC(hipEventRecord(S.ev1, S.stream));
    Q.h4.resize(4 * S.n);
    HIPC(hipMemcpyAsync(Q.h4.data(), CG.mask4, 4 * S.n * sizeof(int32_t), hipMemcpyDeviceToHost, S.stream));
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(S.stream));
    int32_t *out[4] = {iceTmask, iceUmask, iceEmask, iceNmask};
    for (int k = 0; k < 4; ++k) std::memcpy(out[k], Q.h4.data() + (size_t)k * S.n, S.n * sizeof(int32_t));
    float ms = 0;
    if (hipEventElapsedTime(&ms, S.ev3, S.ev1) == hipSuccess) Q.t_ms = ms;
    Q.pending = true;
    return 0;
}

// seabed stress factors TbE / TbN on the device, LKD method (seabed_stress_factor_LKD with grid_location = 'E' / 'N',
// ice_dyn_shared.F90:1386-1460; call site ice_dyn_evp.F90:803-815), from the aice / vice of the last
// cice_evp_hip_cgrid_prep (ghost cells as given) and the masks it produced.  Between _cgrid_prep and _cgrid_prep_finish.  One device exp() per
// ice face: <= 1 ulp from the host libm's.
int cice_evp_hip_cgrid_seabed_lkd(const double *hwater, double k1, double k2, double alphab, double threshold_hw)
{
    CGridState::Prep &Q = CG.prep;
    if (!Q.pending) return fail(-1, "no prepared C-grid state (cice_evp_hip_cgrid_prep first)");
    if (!Q.hwater) {
        if (!hwater) return fail(-1, "hwater needed on the first call");
        if (alloc_d(&Q.hwater, S.n)) return -1;
    }
    if (hwater && h2d(Q.hwater, hwater)) return -1;
    // aice, vice, hwater are read at (i+1, j) / (i, j+1) as the caller handed them over, ghost cells included -- the
    // reference reads its module arrays the same way (their ghost cells are current in the host model)
    EvpCgPrep P{};
    fill_prep(P);
    evp_launch_cgrid_seabed_lkd(P, S.d.nblocks, CG.mask, Q.hwater, k1, k2, alphab, threshold_hw, S.stream);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

// ... probabilistic method (seabed_stress_factor_prob with TbE / TbN, ice_dyn_shared.F90:1475-1683; call site
// ice_dyn_evp.F90:816-827): the T-point factor as on the B grid, then the maximum over the two T-cells of every ice face
int cice_evp_hip_cgrid_seabed_prob(const double *hwater, const double *aicen, const double *vicen, int32_t ncat, double alphab,
                                   double rhoi, double gravit, double pi, double puny)
{
    CGridState::Prep &Q = CG.prep;
    if (!Q.pending) return fail(-1, "no prepared C-grid state (cice_evp_hip_cgrid_prep first)");
    if (!aicen || !vicen || ncat < 1) return fail(-1, "bad argument");
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
    HIPC(hipMemcpyAsync(Q.aicen, aicen, S.n * (size_t)ncat * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIPC(hipMemcpyAsync(Q.vicen, vicen, S.n * (size_t)ncat * sizeof(double), hipMemcpyHostToDevice, S.stream));
    EvpPrep PB{};
    PB.nx = S.d.nx_block; PB.ny = S.d.ny_block; PB.plane = S.plane; PB.blk = S.blk;
    PB.mask = CG.mask;                         // bit0 = iceTmask in both layouts
    evp_launch_seabed_prob_t(PB, S.d.nblocks, Q.hwater, Q.aicen, Q.vicen, ncat, alphab, rhoi, S.prm.rhow, gravit, pi, puny, Q.tbt,
                             S.stream);
    EvpCgPrep P{};
    fill_prep(P);
    evp_launch_cgrid_seabed_prob_faces(P, S.d.nblocks, CG.mask, Q.tbt, S.stream);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

// ... or computed by the host after the preparation (the reference's own routines, libm exp(): what the Fortran entry does)
int cice_evp_hip_cgrid_set_tb(const double *TbE, const double *TbN)
{
    if (!CG.prep.pending) return fail(-1, "no prepared C-grid state (cice_evp_hip_cgrid_prep first)");
    if (!TbE || !TbN) return fail(-1, "null argument");
    if (h2d(CG.in[CI_TBE], TbE) || h2d(CG.in[CI_TBN], TbN)) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

// ice strength (the host's: icepack_ice_strength + its halo update, ice_dyn_evp.F90:596-608, 727-728) and the per-call
// set-up of the loop; cice_evp_hip_cgrid_subcycle / _download follow as after cice_evp_hip_cgrid_upload
int cice_evp_hip_cgrid_prep_finish(const double *strength, int32_t visc_method)
{
    CGridState::Prep &Q = CG.prep;
    if (!Q.pending) return fail(-1, "no prepared C-grid state (cice_evp_hip_cgrid_prep first)");
    if (!strength) return fail(-1, "null argument");
    if (visc_method != 0 && visc_method != 1) return fail(-1, "visc_method %d (0 avg_zeta, 1 avg_strength)", visc_method);
    if (h2d(CG.in[CI_STRENGTH], strength)) return -1;
    Q.pending = false;
    return finish_upload(visc_method);
}

// what the preparation left on the device, for hosts that need it (dyn_finish at E / N points reads aiX, fmX, uocnX,
// vocnX) and for the tests: table 0 = the loop's 19 state / work arrays, 1 = its 23 inputs
int cice_evp_hip_cgrid_fetch(int32_t table, int32_t index, double *dst)
{
    if (!S.ready || !CG.geo) return fail(-1, "C-grid EVP: geometry not set");
    if (!dst) return fail(-1, "null argument");
    const double *src = nullptr;
    if (table == 0 && index >= 0 && index < CG_NF) src = CG.f[index];
    if (table == 1 && index >= 0 && index < CG_NIN) src = CG.in[index];
    if (!src) return fail(-1, "cgrid_fetch: table %d index %d", (int)table, (int)index);
    if (d2h(dst, src)) return -1;
    HIPC(hipStreamSynchronize(S.stream));
    return 0;
}

#ifdef CICE_EVP_HIP_TESTING
// Phase stamps of the last cg_one launch (CICE_EVP_HIP_CGRID_PROF=1 at cice_evp_hip_cgrid_set_geometry): [windows][8] x u64.
// Returns the number of windows, < 0 on error.
int cice_evp_hip_debug_cgres_prof(uint64_t *out, int32_t ntiles_max)
{
    if (!S.ready || !CG.res.prof) return fail(-1, "no profiled resident C-grid launch (CICE_EVP_HIP_CGRID_PROF=1)");
    if (!out || ntiles_max < CG.res.ntiles) return fail(-1, "bad argument: need room for %d windows", CG.res.ntiles);
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemcpy(out, CG.res.prof, (size_t)CG.res.ntiles * 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CG.res.ntiles;
}

int cice_evp_hip_debug_cgrid_prof(uint64_t *out, int32_t ntiles_max)
{
    if (!S.ready || !CG.one.prof) return fail(-1, "no profiled one-launch kernel (CICE_EVP_HIP_CGRID_PROF=1)");
    if (!out || ntiles_max < CG.one.ntiles) return fail(-1, "bad argument: need room for %d windows", CG.one.ntiles);
    HIPC(hipStreamSynchronize(S.stream));
    HIPC(hipMemcpy(out, CG.one.prof, (size_t)CG.one.ntiles * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CG.one.ntiles;
}
#endif

int cice_evp_hip_cgrid_timings(double *out, int32_t n)
{
    if (!out || n < 2) return fail(-1, "need room for 2 values");
    out[0] = CG.t_loop_ms;
    out[1] = (double)CG.t_nsub;
    if (n >= 3) out[2] = CG.prep.t_ms;           // device time of the last cice_evp_hip_cgrid_prep (kernels, without the copies)
    if (n >= 4) out[3] = (double)CG.t_one;       // subcycles of the last call run as one launch each
    if (n >= 5) out[4] = geo_derived() ? 1.0 : 0.0;   // ... with 15 of the 23 static arrays derived in the kernel
    if (n >= 6) out[5] = (double)CG.res.last_nsub;    // subcycles of the last call inside ONE launch of the on-chip resident kernel
    if (n >= 7) out[6] = CG.res.t_probe_ms;           // ... and what its probe measured per subcycle, ms (-1: no probe ran)
    if (n >= 8) out[7] = (double)CG.res.fallbacks;    // cice_evp_hip_cgrid_run calls repeated without it after one of its waits gave up
    if (n >= 9) out[8] = (double)CG.res.n_live;       // windows of the resident kernel that hold ice in this call (only they run) ...
    if (n >= 10) out[9] = (double)CG.res.ntiles;      // ... of so many
    if (n >= 11) out[10] = (double)CG.one.nitems;     // the one-launch schedule's marched kernel (cg_strip): work items (0: not in use) ...
    if (n >= 12) out[11] = (double)CG.one.strip_cells;   // ... the cells it owns ...
    if (n >= 13) out[12] = (double)CG.one.ntiles_e;   // ... and the windows cg_one keeps beside it
    if (n >= 14) out[13] = (double)CG.one.strip_seg;  // ... rows per segment
    if (n >= 15) out[14] = (double)CG.one.strip_len;  // ... 1: it forms six of the eight lengths from dxN, dyE
    return 0;
}

}  // extern "C"
