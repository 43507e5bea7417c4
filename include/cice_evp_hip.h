/* =====================================================================
 * cice_evp_hip.h -- C ABI of the MI355X-native EVP dynamics core.
 *
 * Drop-in boundary: the place where CICE's evp() already hands the whole EVP
 * subcycle to an alternative core,
 *     call dyn_evp1d_run(stressp_1 ... iceUmask)
 *         cicecore/cicedyn/dynamics/ice_dyn_evp.F90:846-856   (run)
 *         cicecore/cicedyn/dynamics/ice_dyn_evp.F90:153-155   (init)
 *     argument contract: ice_dyn_evp1d.F90:121-153
 * and what that region computes in the standard path:
 *     do ksub=1,ndte { stress ; stepu } ; halo(uvel,vvel)   ice_dyn_evp.F90:859-913
 *
 * Conventions
 *  - every array pointer is caller-owned HOST memory laid out as Fortran
 *    (nx_block, ny_block, max_blocks) column-major real(8) -- CICE's own
 *    module arrays are passed straight through, no copies on the Fortran side;
 *    only blocks 1..nblocks are touched;
 *  - masks are Fortran logical(4) / int32: non-zero = .true.;
 *  - all functions return 0 on success, a non-zero code otherwise, and never
 *    exit(): the Fortran wrapper turns non-zero into abort_ice(...) as the
 *    reference's fail-stop convention requires (comm/mpi/ice_exit.F90);
 *    cice_evp_hip_last_error() returns the message;
 *  - one host thread per process calls in (evp() is called outside OpenMP
 *    regions, general/ice_step_mod.F90:1007); the library is not re-entrant.
 * ===================================================================== */
#ifndef CICE_EVP_HIP_H
#define CICE_EVP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CICE_EVP_HIP_ABI_VERSION 1

/* boundary types: ice_domain.F90 domain_nml ew_boundary_type / ns_boundary_type */
enum {
    CICE_EVP_BND_CLOSED = 0,
    CICE_EVP_BND_OPEN = 1,
    CICE_EVP_BND_CYCLIC = 2,
    CICE_EVP_BND_TRIPOLE = 3, /* u-fold; ns only */
    CICE_EVP_BND_TRIPOLET = 4 /* T-fold ('tripoleT'); ns only; B-grid subcycle loop (cice_evp_hip_run / _upload / _subcycle /
                               * _download), any rank layout (the streaming kernel; the images of the top row are interior
                               * cells, so the exchange always follows the launch); cice_evp_hip_stress_halo and the
                               * preparation phase (cice_evp_hip_prep: the centre rule of the T-fold rewrites the top physical
                               * row, ice_boundary.F90:1563-1583) with the top row on one rank; the C grid (cice_evp_hip_cgrid_*:
                               * five launches + fold steps from the T-fold lists of the four field locations, device
                               * preparation included) with the rows NY-2 .. NY on one rank */
};

/* Block decomposition of this process (type(block), ice_blocks.F90:21-41;
 * nblocks/blocks_ice, ice_domain.F90:42-78) plus -- for nranks > 1 -- the
 * global block table every rank can build with get_block_parameter and
 * distrb_info%blockLocation (ice_distribution.F90:24-37).                 */
typedef struct {
    int32_t nx_block, ny_block; /* block array extents incl. ghost cells           */
    int32_t nblocks;            /* blocks owned by this process                   */
    int32_t max_blocks;         /* 3rd extent of the host arrays (>= nblocks)     */
    int32_t nghost;             /* must be 1 (ice_blocks.F90:47)                  */
    int32_t nx_global, ny_global;
    int32_t ew_boundary_type, ns_boundary_type;
    int32_t rank, nranks;       /* position in the EVP process group              */
    /* per local block, length nblocks; 1-based local indices like type(block)    */
    const int32_t *ilo, *ihi, *jlo, *jhi;
    const int32_t *iglob0, *jglob0; /* i_glob(ilo), j_glob(jlo)                   */
    /* global block table, length nblocks_tot; may be NULL when nranks == 1.
     * A block is the interior rectangle [gi0, gi0+gnx) x [gj0, gj0+gny) of the
     * global index space owned by `gowner` (-1: eliminated land block) as its
     * `glocal`-th local block (0-based).                                          */
    int32_t nblocks_tot;
    const int32_t *gi0, *gj0, *gnx, *gny, *gowner, *glocal;
} cice_evp_hip_dims;

/* EVP scalars: set_evp_parameters (ice_dyn_shared.F90:453-486), module
 * parameters u0,cosw,sinw (:68-70), capping (ice_init.F90:1553-1558), Ktens,
 * deltaminEVP, and rhow = icepack_query_parameters(rhow_out) (:920).           */
typedef struct {
    int32_t ndte;     /* default subcycle count                                     */
    int32_t strict;   /* 1: no FMA contraction (bit-comparable with the CPU oracle
                         and with the reference built -ffp-contract=off); 0: fused  */
    double arlx1i, denom1, brlx, revp;
    double e_factor, epp2i;
    double capping, Ktens, deltaminEVP;
    double u0, cosw, sinw;
    double rhow;
} cice_evp_hip_params;

/* ---- life cycle --------------------------------------------------------- */

/* Replaces dyn_evp1d_init (ice_dyn_evp1d.F90:73-117).  Static grid arrays are
 * CICE's ice_grid module arrays (HTE,HTN,dxT,dyT,uarear,tarea).  Derives the
 * metric terms of init_dyn_shared (ice_dyn_shared.F90:384-441) and uploads
 * them; builds the halo plan; selects the device (HIP device = local rank
 * unless CICE_EVP_HIP_DEVICE is set).
 * dims->nblocks == 0 with nranks > 1: a task the reference's distribution gave no block
 * (shared/ice_distribution.F90; evp() then loops over nothing).  Accepted: the rank is a
 * bystander -- cice_evp_hip_comm_unique_id / _comm_init / _halo_export / _halo_import /
 * _finalize work and keep the other ranks' collectives complete; every other entry
 * point refuses it, the host has nothing to hand over there.  With such a rank in the job
 * the marching path stays off on every rank.                                  */
int cice_evp_hip_init(const cice_evp_hip_dims *dims, const cice_evp_hip_params *params,
                      const double *HTE, const double *HTN, const double *dxT, const double *dyT,
                      const double *uarear, const double *tarea);

/* Optional: overwrite the derived metric terms with the caller's own arrays
 * (ice_dyn_shared module arrays cxp,cyp,cxm,cym,dxhy,dyhx,DminTarea); NULL = keep.
 * On tripole grids pass at least dxhy,dyhx: their north ghost row is a mirrored
 * interior value (halo update with sign, ice_dyn_shared.F90:412-417) which the library
 * can only reproduce by local differencing when the grid is symmetric about the seam. */
int cice_evp_hip_set_metrics(const double *cxp, const double *cyp, const double *cxm,
                             const double *cym, const double *dxhy, const double *dyhx,
                             const double *DminTarea);

/* Replaces dyn_evp1d_run (ice_dyn_evp1d.F90:121-319): H2D, `ndte` subcycles on
 * the device, D2H.  Same argument order as the reference routine, plus
 * uvel_init/vvel_init (ice_dyn_shared module arrays; read only when revp=1) and
 * the subcycle count.  On entry the ghost cells of uvel, vvel mirror their sources, as they do at
 * the reference's call site (evp() halo-updates them right before the loop, ice_dyn_evp.F90:729-732):
 * inside the loop only cells with ice are exchanged.  On exit: 12 stresses, uvel, vvel (ghost cells
 * current), strintxU/yU, taubxU/yU hold the state after the last subcycle.      */
int cice_evp_hip_run(double *stressp_1, double *stressp_2, double *stressp_3, double *stressp_4,
                     double *stressm_1, double *stressm_2, double *stressm_3, double *stressm_4,
                     double *stress12_1, double *stress12_2, double *stress12_3, double *stress12_4,
                     const double *strength, const double *cdn_ocnU, const double *aiU,
                     const double *uocnU, const double *vocnU, const double *waterxU,
                     const double *wateryU, const double *forcexU, const double *forceyU,
                     const double *umassdti, const double *fmU, double *strintxU, double *strintyU,
                     const double *TbU, double *taubxU, double *taubyU, double *uvel, double *vvel,
                     const double *uvel_init, const double *vvel_init, const int32_t *iceTmask,
                     const int32_t *iceUmask, int32_t ndte);

int cice_evp_hip_finalize(void);

/* Optional: page-lock a caller-owned host array that outlives the library's
 * initialisation (CICE's module arrays), so that the copies of cice_evp_hip_run are
 * direct DMA.  Idempotent per pointer; released by cice_evp_hip_finalize.          */
int cice_evp_hip_pin_host(const void *ptr, int64_t bytes);

/* Options of the per-call entry points.
 * CICE_EVP_HIP_OPT_STRESS_RESIDENT: the 12 stress components stay on the device between calls of
 * cice_evp_hip_run.  evp() is their only writer, so they are uploaded on the first call only (and after
 * cice_evp_hip_invalidate_stresses, e.g. when a restart file was read into the host arrays), zeroed off
 * iceTmask on the device as dyn_prep2 does on the host arrays (ice_dyn_shared.F90:712-727), and NOT
 * written back: the host arrays are stale until cice_evp_hip_fetch_stresses, which a host calls where
 * something else reads them (restart: ice_restart_driver.F90:187-200; history: principal stresses).
 * 24 of the 50 per-call array transfers go away.                                                  */
#define CICE_EVP_HIP_OPT_STRESS_RESIDENT 1
int cice_evp_hip_set_option(int32_t key, int32_t value);
int cice_evp_hip_fetch_stresses(double *const *sig12);
int cice_evp_hip_invalidate_stresses(void);

/* ---- resident-state entry points (same work as _run, split in three so that
 *      a caller can keep the state in HBM across calls) ----------------------- */

/* fields: pointer table in the order of the cice_evp_hip_run arguments
 * stressp_1 .. vvel_init (32 entries); masks as in _run.                      */
int cice_evp_hip_upload(const double *const *fields32, const int32_t *iceTmask,
                        const int32_t *iceUmask);
/* ndte subcycles (stress + stepu + velocity halo) on the resident state.      */
int cice_evp_hip_subcycle(int32_t ndte);
/* D2H of the 18 output fields into a 32-entry table (NULL entries skipped).   */
int cice_evp_hip_download(double *const *fields32);
/* HIP event on the library's stream: which=0 begins, which=1 ends a caller's timed
 * region (elapsed ms: cice_evp_hip_get_timings()[6]).                           */
int cice_evp_hip_mark(int32_t which);
/* Block the host until all device work of this library has finished.          */
int cice_evp_hip_sync(void);

/* ---- next tier: the two kernels evp() runs on the final velocities ----
 * deformations (ice_dyn_shared.F90:1756-1860, call site ice_dyn_evp.F90:920-934) and
 * dyn_finish (ice_dyn_shared.F90:1291-1365, call site ice_dyn_evp.F90:1395-1406),
 * computed on the device from the resident state after cice_evp_hip_subcycle/_run.
 * set_post_geometry: ice_grid arrays dxU, dyU, tarear (once).  deformations: outputs,
 * zero where iceTmask is false.  dyn_finish: strocnxU/yU are inout.               */
/* stress_halo (rest of SURVEY 8 f-3): on a tripole grid, force the 12 resident stress components
 * symmetric across the seam -- what evp() does with 12 x ice_HaloUpdate_stress on the host arrays
 * right after the subcycle loop (ice_dyn_evp.F90:1321-1389; ice_boundary.F90:7441-7826): the
 * ghost row NY+1 of component c takes the mirrored top physical row of its partner (1<->3, 2<->4).
 * On a tripoleT grid the same twelve calls (T-fold offsets, ice_boundary.F90:7697-7722) rewrite the top
 * PHYSICAL row of components 1 and 2 from the partner's mirrored cell, leave 3 and 4 with plain images in
 * the east-west ghost cells of that row, and touch one cell of the ghost row: the north-west corner ghost
 * cell of every top-row block (pinned on the reference's own output).
 * Call between cice_evp_hip_subcycle and cice_evp_hip_download; no-op on other grids.            */
int cice_evp_hip_stress_halo(void);
/* 1 if cice_evp_hip_stress_halo can do that on this rank layout (tripole: always; tripoleT: the top row on one rank and no
 * eliminated block in it), else 0: a host that keeps the stresses resident asks before it leaves the step to the library.
 * The answer is THIS rank's (a rank without top-row blocks says 1 whatever the others say); that is sound rank by rank because
 * the tripoleT symmetrisation on the device never crosses ranks -- a host whose restart / diagnostics code wants one answer
 * for the whole job reduces it itself (MIN).                                                                              */
int cice_evp_hip_stress_halo_available(void);
int cice_evp_hip_set_post_geometry(const double *dxU, const double *dyU, const double *tarear);
int cice_evp_hip_deformations(double *divu, double *shear, double *vort, double *rdg_conv,
                              double *rdg_shear);
int cice_evp_hip_dyn_finish(double *strocnxU, double *strocnyU);

/* ---- next tier: the preparation phase of evp() on the device ---------------
 * Everything evp() does between its entry and the subcycle loop on the B grid
 * (ice_dyn_evp.F90:383-840): dyn_prep1 (ice_dyn_shared.F90:496-576), the ice_HaloUpdate calls on
 * iceTmask and the T-grid fields (:413-428, 466-470), grid_average_X2Y T->U in its state-masked
 * and flux flavours (ice_grid.F90:4183-4204, 4650-4666), dyn_prep2 (ice_dyn_shared.F90:586-839)
 * and the pre-loop velocity halo (:729-732) -- except icepack_ice_strength and the seabed
 * stress factor TbU, which the host keeps (Icepack; exp()).  Call order per evp():
 *   cice_evp_hip_prep -> host: ice strength from the returned iceTmask -> cice_evp_hip_set_strength
 *   -> cice_evp_hip_subcycle -> cice_evp_hip_download (+ _deformations, _dyn_finish, _stress_halo).
 * set_prep_geometry (once): tmask, umask (logical as int32), hm, tarea, uarea (ice_grid),
 * fcor_blk (ice_dyn_shared).  T-grid fields, in this order: aice, vice, vsno, aice_init, cdn_ocn,
 * uocn, vocn, ss_tltx, ss_tlty, strairxT, strairyT (their ghost cells need not be current).
 * fields32: the table of cice_evp_hip_upload; read here: the 12 stresses, uvel, vvel, TbU (NULL = 0).
 * All 12 stress pointers NULL = keep the stresses the previous call left on the device (evp() is
 * their only writer; pair it with NULL stress entries in cice_evp_hip_download and fetch them only
 * when a restart or history file needs them): 24 of the 49 per-call array transfers go away.
 * iceUmask: in = mask of the previous call (new ice starts at the ocean velocity), out = new mask;
 * strintxU/strintyU/strocnxU/strocnyU (may be NULL): zeroed off the ice on the host arrays.
 * On a split domain the T-grid halos use the same transport as the velocities (collective call);
 * on a tripole domain too, whatever the rank layout (a fold row split over ranks in x: through the
 * exchange of a shifted copy, cice_amd/csrc/halo_plan.h).                                        */
typedef struct cice_evp_hip_prep_params {
    double dt;                 /* dynamics time step (dyn_prep2's Xmass/dt)                    */
    double rhoi, rhos, gravit; /* icepack_query_parameters                                      */
    double dyn_area_min, dyn_mass_min;
    int32_t ssh_stress_coupled;/* ssh_stress == 'coupled' (else 'geostrophic')                  */
} cice_evp_hip_prep_params;
int cice_evp_hip_set_prep_geometry(const int32_t *tmask, const int32_t *umask, const double *hm,
                                   const double *tarea, const double *uarea, const double *fcor_blk);
int cice_evp_hip_prep(const cice_evp_hip_prep_params *pp, const double *const *tfields11,
                      const double *const *fields32, int32_t *iceTmask, int32_t *iceUmask,
                      double *strintxU, double *strintyU, double *strocnxU, double *strocnyU);
int cice_evp_hip_set_strength(const double *strength);
/* Seabed stress factor TbU for the coming subcycle loop, when the host computes it: evp() does so
 * AFTER dyn_prep2 (which zeroes it, ice_dyn_shared.F90:706) from the new iceUmask
 * (seabed_stress_factor_LKD / _prob, ice_dyn_evp.F90:770-826), so it cannot travel with
 * cice_evp_hip_prep.  Call between cice_evp_hip_prep and cice_evp_hip_subcycle.                  */
int cice_evp_hip_set_tbu(const double *TbU);
/* ... or on the device, LKD method (seabed_stress_factor_LKD, ice_dyn_shared.F90:1386-1460): k1, k2, alphab,
 * threshold_hw are ice_dyn_shared's namelist parameters, hwater ice_flux's water depth (NULL: the copy of the
 * previous call).  Uses the aice / vice and the masks of the last cice_evp_hip_prep.  One exp() per ice U-cell from
 * the device's math library: TbU may differ from the host's libm result in the last bit (tolerance).  */
int cice_evp_hip_seabed_lkd(const double *hwater, double k1, double k2, double alphab, double threshold_hw);
/* The same for seabed_stress_method = 'probabilistic' (seabed_stress_factor_prob, ice_dyn_shared.F90:1475-1683; B grid):
 * aicen, vicen = ice_state's category arrays (nx_block, ny_block, ncat, max_blocks) with current ghost cells; rhoi,
 * gravit, pi, puny from icepack_query_parameters (rhow is the one given at init).  exp() / log() are the device
 * library's: TbU within a few ulp of the reference's, not bit-identical.                                          */
int cice_evp_hip_seabed_prob(const double *hwater, const double *aicen, const double *vicen, int32_t ncat, double alphab,
                             double rhoi, double gravit, double pi, double puny);
/* Returns its argument.  Lets a Fortran host take the address of a module array that lacks the
 * TARGET attribute (type(*), dimension(*) dummy) to fill the pointer tables above.              */
void *cice_evp_hip_addr(const void *array);
/* which: 0 aiU 1 cdn_ocnU 2 uocnU 3 vocnU 4 umassdti 5 fmU 6 waterxU 7 wateryU 8 forcexU 9 forceyU
 * 10 uvel_init 11 vvel_init 12 strtltxU 13 strtltyU 14 strairxU 15 strairyU 16 tmass 17 umass
 * 18 uvel 19 vvel 20 TbU (as the subcycle loop will see them)                                         */
int cice_evp_hip_prep_fetch(int32_t which, double *dst);

/* ---- C-grid EVP subcycle (second half): evp()'s loop for grid_ice = 'C',
 * cicedyn/dynamics/ice_dyn_evp.F90:938-1099 -- strain_rates_U, stressC_T, stressC_U, div_stress_Ex / _Ny,
 * stepu_C / stepv_C, the E<->N and face->corner velocity averages, and the eight ice_HaloUpdate calls of every
 * subcycle (fused into the kernels; ghost cells owned by other ranks through the halo transport of the B-grid path).
 * Call after cice_evp_hip_init (dims, scalars) and, for nranks > 1, cice_evp_hip_comm_init or _halo_import.
 * Cyclic, closed or open boundaries, tripole u-fold (the blocks holding rows NY-1, NY on one rank) and T-fold ('tripoleT',
 * ice_boundary.F90:1563-1622; rows NY-2 .. NY on one rank); any number of blocks.
 *
 * static23 (ice_grid / ice_dyn_evp arrays, once):
 *   dxT dyT dxU dyU dxE dyE dxN dyN uarea tarea earea narea earear narear epm npm uvm hm DminTarea
 *   ratiodxN ratiodxNr ratiodyE ratiodyEr
 * fields19 (in/out; entries 14..18 are evp()'s work arrays: out only, zero at entry as evp() zeroes them :351-361):
 *   uvelE vvelE uvelN vvelN uvel vvel stresspT stressmT stress12T stress12U strintxE strintyN taubxE taubyN
 *   zetax2T etax2T etax2U shearU deltaU
 * inputs23 (what the reference's preparation leaves in ice_dyn_evp's module arrays):
 *   strength cdn_ocnE aiE uocnE vocnE waterxE forcexE emassdti fmE uvelE_init TbE rheofactE
 *   cdn_ocnN aiN uocnN vocnN wateryN forceyN nmassdti fmN vvelN_init TbN rheofactN
 * visc_method: 0 'avg_zeta', 1 'avg_strength' (ice_dyn_evp.F90:992-996).
 * On entry ghost cells mirror their sources, as evp()'s preparation leaves them (inside the loop only cells with ice
 * are exchanged).  Not done here (it is after the loop, :1437-1440): the ice_HaloUpdate of strintxE / strintyN.    */
int cice_evp_hip_cgrid_set_geometry(const double *const *static23);
int cice_evp_hip_cgrid_run(int32_t ndte, int32_t visc_method, double *const *fields19,
                           const double *const *inputs23, const int32_t *iceTmask, const int32_t *iceUmask,
                           const int32_t *iceEmask, const int32_t *iceNmask);
/* The preparation phase of evp() for grid_ice = 'C' on the device (ice_dyn_evp.F90:383-735, calc_strair branch, forcing on the
 * T grid): dyn_prep1, the T-grid halo updates, the state-masked / flux averages T -> U, E, N (grid_average_X2Y 'S' / 'F'),
 * dyn_prep2 at U, N and E points, the stresses zeroed off the ice, the velocity exchanges and face -> face / face -> corner
 * averages -- everything the loop reads except the ice strength (Icepack's) is computed where the loop will read it:
 * 11 T-grid arrays travel in instead of 14 + 23.
 *   cice_evp_hip_cgrid_set_prep_geometry (once, after _cgrid_set_geometry): tmask, umaskCD, emask, nmask (ice_grid logicals
 *     as int32), fcor_blk, fcorE_blk, fcorN_blk (ice_dyn_shared)
 *   cice_evp_hip_cgrid_prep: tfields11 as for cice_evp_hip_prep; state12 = uvelE vvelE uvelN vvelN uvel vvel stresspT
 *     stressmT stress12T stress12U strintxE strintyN as evp() is entered with them, or NULL = what the previous call left
 *     on the device (evp() is their only writer); iceUmask / iceEmask / iceNmask: in = the previous call's masks, out = new
 *     (physical cells only, as dyn_prep2); iceTmask: out.  Not done for the host's copies: strintyE, strintxN and the
 *     ocean stresses zeroed off the ice (ice_dyn_shared.F90:776-781) -- diagnostics the loop does not read.
 *   host: ice strength from the returned iceTmask; optionally cice_evp_hip_cgrid_seabed_lkd / _prob (TbE, TbN; otherwise 0)
 *   cice_evp_hip_cgrid_prep_finish(strength, visc_method) -> cice_evp_hip_cgrid_subcycle -> cice_evp_hip_cgrid_download
 *   cice_evp_hip_cgrid_fetch(table, index, dst): table 0 = fields19, 1 = inputs23 as they are on the device (dyn_finish at
 *     E / N points reads aiX, fmX, uocnX, vocnX on the host).
 * A tripole fold row split over ranks in x is refused (centre-field mirror across the fold on another rank).          */
int cice_evp_hip_cgrid_set_prep_geometry(const int32_t *tmask, const int32_t *umaskCD, const int32_t *emask,
                                         const int32_t *nmask, const double *fcor_blk, const double *fcorE_blk,
                                         const double *fcorN_blk);
int cice_evp_hip_cgrid_prep(const cice_evp_hip_prep_params *pp, const double *const *tfields11,
                            const double *const *state12, int32_t *iceTmask, int32_t *iceUmask, int32_t *iceEmask,
                            int32_t *iceNmask);
int cice_evp_hip_cgrid_seabed_lkd(const double *hwater, double k1, double k2, double alphab, double threshold_hw);
int cice_evp_hip_cgrid_seabed_prob(const double *hwater, const double *aicen, const double *vicen, int32_t ncat, double alphab,
                                   double rhoi, double gravit, double pi, double puny);
int cice_evp_hip_cgrid_set_tb(const double *TbE, const double *TbN);   /* seabed factors computed by the host after _cgrid_prep */
int cice_evp_hip_cgrid_prep_finish(const double *strength, int32_t visc_method);
int cice_evp_hip_cgrid_fetch(int32_t table, int32_t index, double *dst);
/* the same in three steps (state14 = the first 14 entries of fields19) */
int cice_evp_hip_cgrid_upload(const double *const *state14, const double *const *inputs23,
                              const int32_t *iceTmask, const int32_t *iceUmask, const int32_t *iceEmask,
                              const int32_t *iceNmask, int32_t visc_method);
int cice_evp_hip_cgrid_subcycle(int32_t ndte);
int cice_evp_hip_cgrid_download(double *const *fields19);
int cice_evp_hip_cgrid_sync(void);
/* deformationsC_T (ice_dyn_shared.F90:1968-2074), which evp() calls right after the C-grid loop (ice_dyn_evp.F90:1106-1119),
 * on the device from the loop's resident final state: divu, shear, vort, rdg_conv, rdg_shear (ice_state / ice_flux arrays,
 * inout: written on the ice T-cells of dyn_prep2's list only).  tarear = ice_grid's 1/tarea, taken at the first call.  */
int cice_evp_hip_cgrid_deformations(const double *tarear, double *divu, double *shear, double *vort, double *rdg_conv,
                                    double *rdg_shear);
/* dyn_finish at N and E points (ice_dyn_shared.F90:1291-1365; ice_dyn_evp.F90:1408-1436, right after the one at U points) on
 * the device, from the loop's resident final face velocities and the per-call operands it already holds: strocnxN, strocnyN,
 * strocnxE, strocnyE (ice_flux arrays, inout: written on the cells of dyn_prep2's N / E lists only).                       */
int cice_evp_hip_cgrid_dyn_finish(double *strocnxN, double *strocnyN, double *strocnxE, double *strocnyE);
/* out[0] = ms of the last cgrid_subcycle (HIP events), out[1] = its ndte, [2] (n >= 3) = device ms of the last cgrid_prep,
 * [3] (n >= 4) = how many of those subcycles ran as one launch each (the default schedule on one rank without a fold),
 * [4] (n >= 5) = 1 if that kernel derives 15 of the 23 static arrays from the eight dx / dy arrays (allowed when
 * cice_evp_hip_cgrid_set_geometry found the reference's start-up identities to hold bit for bit; CICE_EVP_HIP_CGRID_GEO=0
 * keeps all 23 in use; in the test build only),
 * [5] (n >= 6) subcycles of the last call that ran inside ONE launch of the on-chip resident kernel (evp_cgrid_res.hip: the state
 * of a call in registers and LDS, face velocities traded between windows as tagged records; one rank, no T-fold, every window
 * that holds ice co-resident -- seabed stress / an ocean turning angle / rheofact != 1 take its general momentum step --: up to ~130k cells with ice everywhere, larger domains
 * when only part of them is covered; CICE_EVP_HIP_CGRID_RESIDENT=0 / 1 forbids / requires it), [6] (n >= 7) what its start-up probe
 * measured per subcycle, ms (-1: no probe ran), [7] (n >= 8) calls of cice_evp_hip_cgrid_run that were repeated with the per-subcycle
 * kernels after one of its (bounded) waits gave up, [8], [9] (n >= 10) its windows that hold ice in this call -- only they run -- and
 * all its windows, [10] .. [14] (n >= 15) the one-launch schedule's marched kernel (evp_cgrid.hip: cg_strip -- the interior of
 * large blocks, one wave per strip of 61 columns and segment of rows): work items (0: not in use), cells it owns, windows the
 * windowed kernel keeps beside it (the block edges), rows per segment, 1 if it forms dxT, dyT, dxU, dyU, dxE, dyN from dxN and dyE
 * (the reference's start-up means, verified bit for bit on the caller's arrays)  */
int cice_evp_hip_cgrid_timings(double *out, int32_t n);

/* ---- multi-GPU: RCCL point-to-point halo over xGMI ---------------------------- */
/* 128-byte ncclUniqueId made by rank 0 and distributed by the host program
 * (MPI_Bcast in CICE; torch.distributed in bench.py).                          */
int cice_evp_hip_comm_unique_id(void *id128);
int cice_evp_hip_comm_init(const void *id128);
/* What RCCL itself says about the communicator of this rank, for logs and for reading a first multi-GPU run (bench.py prints it
 * per rank): out[0] 1 = a communicator exists, [1] ncclCommCount (ranks RCCL sees), [2] ncclCommUserRank, [3] ncclCommCuDevice,
 * [4] the HIP device this rank runs on; bus_id (may be NULL): that device's PCI bus id ("0000:c1:00.0"), nb bytes.            */
int cice_evp_hip_comm_info(int32_t *out, int32_t n, char *bus_id, int32_t nb);
/* Mailbox halo inside one node: neighbouring ranks store ghost values straight into each
 * other's HIP-IPC-mapped inboxes from a kernel (plain stores over xGMI + a flag handshake),
 * so the whole subcycle loop -- exchange included -- is one hipGraph.  Replaces the same
 * ice_HaloUpdate round as the RCCL path (ice_dyn_evp.F90:908-910).  cice_evp_hip_comm_init
 * sets it up by itself (handles travel through an ncclAllGather, a probe exchange of global
 * cell numbers -- the halochk method, drivers/unittest/halochk/halochk.F90:232-247 -- must
 * pass on every rank, else all ranks stay on RCCL).  A host without RCCL does the same by
 * hand: every rank exports CICE_EVP_HIP_HALO_BLOB bytes, the host all-gathers them
 * (rank order) and every rank imports the nranks blobs; import runs the probe exchange
 * (collective).  CICE_EVP_HIP_HALO=rccl|direct overrides the choice.                      */
/* Masked halo for the velocity updates inside the subcycle loop = ice_HaloMask (ice_boundary.F90:889-1062; built by
 * evp() when maskhalo_dyn, ice_dyn_evp.F90:739-770).  halomask: (nx_block, ny_block, max_blocks) int32, 1 where
 * iceUmask, ghost cells updated (the reference's own array), NULL = full halo again.  Exchanged cells whose mask is
 * 0 are dropped on both sides; copies inside a rank and exchanges across the tripole fold are never masked.  Call
 * once per evp() after the masks are known (collective in the sense that every rank must pass its own mask).    */
int cice_evp_hip_halo_mask(const int32_t *halomask);
#define CICE_EVP_HIP_HALO_BLOB 1024
int cice_evp_hip_halo_export(void *blob);
int cice_evp_hip_halo_import(const void *blobs, int32_t nranks);

/* ---- introspection ------------------------------------------------------------ */
int cice_evp_hip_abi_version(void);
int cice_evp_hip_last_error(char *buf, int32_t buflen);
/* out[0]=last subcycle-loop ms (HIP events), [1]=H2D ms, [2]=D2H ms,
 * [3]=subcycles in that loop, [4]=kernel launches per subcycle,
 * [5]=tile variant in use (tile height + 100 * tile-order mode),
 * [6]=ms between cice_evp_hip_mark(0) and cice_evp_hip_mark(1), -1 if unset,
 * [7]=streaming probe ms/subcycle, [8]=resident probe ms/subcycle,
 * [9]=remote halo transport: 0 none, 1 RCCL p2p, 2 mailbox (direct stores over xGMI),
 * [10]=device time of the last cice_evp_hip_prep (kernels + halos, without the copies), ms,
 * [11]=cice_evp_hip_run calls repeated with the streaming kernel after the resident one gave up,
 * [12], [13]=cells this rank sends / receives per velocity exchange of the loop (after cice_evp_hip_halo_mask),
 * [14], [15]=on-chip resident kernel: tiles that ran in the last loop (only the tiles that hold ice run; 0: that loop did not
 *            go through it, e.g. because its ice needed more tiles than the chip holds at once) / tiles of the domain  */
int cice_evp_hip_get_timings(double *out, int32_t n);
/* Halo plan of this rank, for tests: counts[0]=local copies, [1]=#peers,
 * [2]=total send cells, [3]=total recv cells.  Lists may be NULL.
 * local_dst/local_src are 0-based offsets into a (nx_block,ny_block,nblocks)
 * array; sign is +-1 (tripole vector fold).                                    */
/* Per-launch kernel durations by HIP events on the library's stream, state not
 * advanced: out3[0]=fused stress+stepu kernel ms, [1]=halo gather kernel ms,
 * [2]=back-to-back period of the fused kernel ms (launch gap included).        */
int cice_evp_hip_time_kernels(int32_t nrep, double *out3);
/* Measurement aid (no state needed): bytes/s of a plain streaming kernel with the array shape of one B-grid subcycle
 * (30 fp64 arrays in, 16 out, `ncells` elements each, every element touched once) -- what HBM gives a kernel of this
 * shape on this device; the yardstick next to the 8 TB/s pin rate in bench.py's roofline block.                   */
/* The several-subcycles-per-pass ("marching") path for per-rank domains beyond the chip (evp_march.hip): out[0] mode (-1 undecided,
 * 0 off, 1 on), [1] passes run since init, [2] calls it handed back to the one-subcycle kernels (the uploaded ghost
 * values were not images of one global state), [3] strips, [4] segments, [5] rows per segment, [6] 1 = the last
 * cice_evp_hip_subcycle ran through it, [7] (n >= 8) how its ring travels between ranks: -1 not set up, 0 RCCL send / recv,
 * 1 stores into the neighbours' HIP-IPC-mapped inboxes (opt-in: CICE_EVP_HIP_MARCH_DIRECT=1 on every rank, inside one
 * node; in use once a trial exchange has delivered the same bits as RCCL on every rank), 2 on trial; [8] (n >= 10) subcycles
 * a full pass advances the state by (4), [9] subcycles advanced by passes since init.
 * CICE_EVP_HIP_MARCH=0/1 forces the path off / on (default: from 450k cells per rank).  Several ranks:
 * CICE_EVP_HIP_MARCH_OVERLAP=1 (every rank alike) advances the cells other ranks wait for first, on a second stream, and
 * overlaps their RCCL send / recv with the rest of the pass (default: pack, send / recv, unpack after the pass).        */
int cice_evp_hip_march_info(int32_t *out, int32_t n);
/* One line of text on what the last cice_evp_hip_subcycle ran (kernel, halo transport, the marching path and why it is
 * off when it is), for logs.                                                                                       */
int cice_evp_hip_describe_path(char *buf, int32_t n);
int cice_evp_hip_stream_probe(int64_t ncells, double *bytes_per_second);
/* The seam step for any rank layout (pairs, or a ghost cell and the seam cell it mirrors, on different ranks):
 * counts2 = {entries, staging slots}.  After the velocity exchange x[dst] = coef * 0.5*(x[a] - x[b]) (b >= 0) or
 * coef * x[a] (b = -1), a / b = local cells or staging slots n_local + t filled by the exchange with RAW seam values
 * of other ranks (they are the last entries of the peers' send / recv lists).  Returns 1 (not an error) when the
 * stress symmetrisation would need another rank (then cice_evp_hip_stress_halo refuses).  Lists may be NULL.   */
int cice_evp_hip_seam_fin_plan(int32_t *counts2, int32_t *dst, int32_t *a, int32_t *b, int32_t *coef);
/* (The plan introspection entry points of the CPU tests, the phase / placement read-outs of the tools and the test
 * transport are NOT part of this library: they are declared in cice_evp_hip_testing.h and exist only in the test build,
 * libcice_evp_hip_testing.so, compiled with -DCICE_EVP_HIP_TESTING from the same sources.)                          */

#ifdef __cplusplus
}
#endif
#endif /* CICE_EVP_HIP_H */
