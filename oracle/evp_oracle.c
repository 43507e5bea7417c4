/* =====================================================================
 * TEST INFRASTRUCTURE -- CPU restatement (plain C, IEEE fp64, no FMA
 * contraction) of the reference's B-grid EVP subcycle.  It is the checker
 * for the HIP path and the "port" CPU baseline; it is never linked into,
 * loaded by, or called from the product (cice_amd/, libcice_evp_hip.so).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY PIN: this file is checked bit-for-bit against the reference's own
 * evp() compiled unmodified from /root/reference with
 * `amdflang -O2 -ffp-contract=off` (oracle/ref/build_ref.sh) through the
 * golden fixtures in tests/golden/ (tests/test_oracle_golden.py).
 *
 * Every function cites the reference lines it follows
 * (paths relative to /root/reference/cicecore/cicedyn/).
 *
 * Array layout: Fortran (nx_block, ny_block, nblocks) column-major, i.e.
 * C index  blk*nx*ny + (j-1)*nx + (i-1)  with 1-based i,j as in the reference.
 * ===================================================================== */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* numeric constants: shared/ice_constants.F90:79-85 */
static const double p027 = 1.0 / 36.0;  /* p027 = p055*p5  */
static const double p055 = 1.0 / 18.0;  /* p055 = p111*p5  */
static const double p111 = 1.0 / 9.0;
static const double p166 = 1.0 / 6.0;
static const double p222 = 2.0 / 9.0;
static const double p25 = 0.25;
static const double p333 = 1.0 / 3.0;
static const double p5 = 0.5;
static const double c1p5 = 1.5;

enum { BND_CLOSED = 0, BND_OPEN = 1, BND_CYCLIC = 2, BND_TRIPOLE = 3, BND_TRIPOLET = 4 };

typedef struct {
    int nx_block, ny_block, nblocks, nghost;
    int nx_global, ny_global;
    int ew_type, ns_type;
    /* per block, length nblocks each; 1-based local indices as in type(block)
       (infrastructure/ice_blocks.F90:21-41) */
    const int *ilo, *ihi, *jlo, *jhi;
    const int *iglob0, *jglob0; /* global index of (ilo), (jlo) */
} evp_oracle_domain;

typedef struct {
    /* set_evp_parameters: dynamics/ice_dyn_shared.F90:453-486 */
    double arlx1i, denom1, brlx, revp, e_factor, epp2i;
    double capping, Ktens, deltaminEVP;
    double u0, cosw, sinw; /* ice_dyn_shared.F90:68-70,86 */
    double rhow;           /* Icepack constant, ice_dyn_shared.F90:920 */
} evp_oracle_params;

#define IX(i, j) ((size_t)((j)-1) * nx + (size_t)((i)-1))

/* ---------------------------------------------------------------------
 * set_evp_parameters   dynamics/ice_dyn_shared.F90:453-486
 * ------------------------------------------------------------------- */
void evp_oracle_set_parameters(int ndte, double dt, int revised_evp, double elasticDamp,
                               double arlx_in, double brlx_in, double e_yieldcurve,
                               double e_plasticpot, double *out /* arlx, arlx1i, brlx, denom1,
                               revp, epp2i, e_factor, dtei, ecci */)
{
    double dtei = (double)ndte / dt;
    double epp2i = 1.0 / (e_plasticpot * e_plasticpot);
    double e_factor = (e_yieldcurve * e_yieldcurve) / (e_plasticpot * e_plasticpot * e_plasticpot * e_plasticpot);
    double ecci = 1.0 / (e_yieldcurve * e_yieldcurve);
    double revp, denom1, arlx1i, arlx = arlx_in, brlx = brlx_in;
    if (revised_evp) {
        revp = 1.0;
        denom1 = 1.0;
        arlx1i = 1.0 / arlx;
    } else {
        revp = 0.0;
        arlx = 2.0 * elasticDamp * (double)ndte;
        arlx1i = 1.0 / arlx;
        brlx = (double)ndte;
        denom1 = 1.0 / (1.0 + arlx1i);
    }
    out[0] = arlx; out[1] = arlx1i; out[2] = brlx; out[3] = denom1; out[4] = revp;
    out[5] = epp2i; out[6] = e_factor; out[7] = dtei; out[8] = ecci;
}

/* ---------------------------------------------------------------------
 * Ghost-cell update by *semantics* of ice_HaloUpdate for the boundary types
 * cyclic / closed / open (infrastructure/comm/serial/ice_boundary.F90:1066-1760):
 * a ghost cell takes the value of the interior cell that holds the same global
 * (i,j); ghost cells outside a closed/open outer boundary are left untouched
 * when no fillValue is given (ewfillouter/nsfillouter = .false., :1176-1182)
 * and set to `fill` when one is (:1184-1189).  Tripole: see halo_tripole below.
 * ------------------------------------------------------------------- */
static void gather_global(const evp_oracle_domain *d, const double *a, double *g)
{
    const int nx = d->nx_block, ny = d->ny_block;
    for (int b = 0; b < d->nblocks; ++b) {
        const double *ab = a + (size_t)b * nx * ny;
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                int ig = d->iglob0[b] + (i - d->ilo[b]);
                int jg = d->jglob0[b] + (j - d->jlo[b]);
                g[(size_t)(jg - 1) * d->nx_global + (ig - 1)] = ab[IX(i, j)];
            }
    }
}

/* Test hook: when set, every halo update of this file is handed to the callback instead (array, field_loc, field_type)
 * -- tests/test_multirank_cpu.py runs the C-grid loop on the blocks of ONE rank of a multi-rank decomposition and does
 * the exchange itself (plan lists + torch.distributed gloo), to check lists and exchange points without a GPU. */
typedef void (*evp_oracle_halo_cb)(double *a, int field_loc, int field_type);
static evp_oracle_halo_cb g_halo_cb = 0;
void evp_oracle_set_halo_callback(evp_oracle_halo_cb cb) { g_halo_cb = cb; }

/* field_loc: 0 = center, 1 = NE corner, 2 = E face, 3 = N face.  field_type: 0 = scalar, 1 = vector (sign -1 over
 * the tripole fold).  u-fold rules per location (ice_boundary.F90:1626-1683): offsets (ioffset, joffset) =
 * center (0,0), NE corner (1,1), E face (1,0), N face (0,1); points ON the fold -- top physical row of NE-corner and
 * N-face fields -- are forced symmetric first: NE corner pairs i <-> NX-i (i = 1..NX/2-1; i = NX/2 and NX mirror onto
 * themselves), N face pairs i <-> NX+1-i (i = 1..NX/2). */
void evp_oracle_halo_update(const evp_oracle_domain *d, double *a, int field_loc, int field_type,
                            int have_fill, double fill)
{
    if (g_halo_cb && !have_fill) {
        g_halo_cb(a, field_loc, field_type);
        return;
    }
    const int nx = d->nx_block, ny = d->ny_block;
    const int NX = d->nx_global, NY = d->ny_global;
    double *g = (double *)malloc(sizeof(double) * (size_t)NX * NY);
    gather_global(d, a, g);
    const int tripole = (d->ns_type == BND_TRIPOLE);
    const int tfold = (d->ns_type == BND_TRIPOLET);
    const double isign = (field_type == 1) ? -1.0 : 1.0;

    if (tfold) {
        /* tripoleT (T-fold; ice_boundary.F90:1563-1622 offsets, :1686-1722 copy-out with the addresses of :8135-8159: local
         * column i of rows jhi and jhi+1 takes buffer column NX - iGlobal(i) + 1 - ioffset of buffer rows 3 - joffset and
         * 2 - joffset, where buffer rows 1..3 are the global rows NY-2..NY).  NE-corner fields (offsets 0, 1; no pair
         * averaging): the top physical row is the image of row NY-1, the ghost row that of row NY-2 --
         *   a(i, NY) <- isign * a(NX-i+1, NY-1),   a(i, NY+1) <- isign * a(NX-i+1, NY-2)    (ghost columns included).
         * Cell-centre fields (offsets -1, 0; the preparation phase's T-grid updates, ice_dyn_evp.F90:413-428, 466-469): the top
         * physical row lies ON the fold and is made symmetric first (:1568-1583: pairs i <-> NX-i+2 for i = 2..NX/2,
         * xavg = 0.5*(x1 + isign*x2), stored as xavg and isign*xavg; i = 1 and i = NX/2+1 mirror onto themselves), then
         *   a(i, NY) <- isign * avg(NX-i+2, NY),   a(i, NY+1) <- isign * a(NX-i+2, NY-1)    (column NX+1 is column 1).
         * E-face fields (offsets 0, 0; the C grid's uvelE, vvelE): on the fold like the centres, pairs i <-> NX+1-i for
         * i = 1..NX/2 (:1593-1608), then a(i, NY) <- isign * avg(NX-i+1, NY), a(i, NY+1) <- isign * a(NX-i+1, NY-1).
         * N-face fields (offsets -1, 1; vvelN, uvelN): no averaging, a(i, NY) <- isign * a(NX-i+2, NY-1),
         * a(i, NY+1) <- isign * a(NX-i+2, NY-2). */
        if (field_loc == 0 || field_loc == 2) {
            double *top = g + (size_t)(NY - 1) * NX;
            const int i0 = field_loc == 0 ? 2 : 1, off = field_loc == 0 ? 2 : 1;
            for (int i = i0; i <= NX / 2; ++i) {
                const int idst = NX - i + off;
                const double x1 = top[i - 1], x2 = top[idst - 1];
                const double xavg = 0.5 * (x1 + isign * x2);
                top[i - 1] = xavg;
                top[idst - 1] = isign * xavg;
            }
        }
        for (int b = 0; b < d->nblocks; ++b) {
            double *ab = a + (size_t)b * nx * ny;
            const int ilo = d->ilo[b], ihi = d->ihi[b], jlo = d->jlo[b], jhi = d->jhi[b];
            for (int j = jlo - d->nghost; j <= jhi + d->nghost; ++j)
                for (int i = ilo - d->nghost; i <= ihi + d->nghost; ++i) {
                    const int interior = (i >= ilo && i <= ihi && j >= jlo && j <= jhi);
                    int ig = d->iglob0[b] + (i - ilo);
                    int jg = d->jglob0[b] + (j - jlo);
                    if (ig < 1 || ig > NX) {
                        if (d->ew_type != BND_CYCLIC) continue;
                        ig = (ig < 1) ? ig + NX : ig - NX;
                    }
                    if (jg == NY || jg == NY + 1) {
                        /* column NX - ig + 1 - ioffset; rows 3 - joffset / 2 - joffset of the buffer (= global NY-2 .. NY) */
                        int is = NX - ig + ((field_loc == 0 || field_loc == 3) ? 2 : 1);
                        if (is > NX) is -= NX;
                        const int down = (field_loc == 1 || field_loc == 3) ? 1 : 0;      /* joffset */
                        ab[IX(i, j)] = isign * g[(size_t)((jg == NY ? NY - 1 : NY - 2) - down) * NX + (is - 1)];
                        continue;
                    }
                    if (interior || jg < 1) continue;                    /* closed south */
                    ab[IX(i, j)] = g[(size_t)(jg - 1) * NX + (ig - 1)];
                }
        }
        free(g);
        return;
    }
    if (tripole && field_loc == 3) {
        /* N face: (:1664-1677) */
        double *top = g + (size_t)(NY - 1) * NX;
        for (int i = 1; i <= NX / 2; ++i) {
            int idst = NX + 1 - i;
            double x1 = top[i - 1], x2 = top[idst - 1];
            double xavg = 0.5 * (x1 + isign * x2);
            top[i - 1] = xavg;
            top[idst - 1] = isign * xavg;
        }
        /* copy-out: array(i,NY) = isign*buf(NX+1-i, NY): (xavg, isign*xavg) come back as themselves */
    }
    if (tripole && field_loc == 1) {
        /* u-fold, NE-corner field: the top physical row lies on the fold and is
           degenerate; enforce symmetry by averaging the two copies
           (ice_boundary.F90:1630-1649), then the averaged row replaces the top
           physical row of the array (copy-out with offsets 1,1: :1632-1633,1689-1722). */
        double *top = g + (size_t)(NY - 1) * NX;
        for (int i = 1; i <= NX / 2 - 1; ++i) {
            int idst = NX - i;
            double x1 = top[i - 1], x2 = top[idst - 1];
            double xavg = 0.5 * (x1 + isign * x2);
            top[i - 1] = xavg;
            top[idst - 1] = isign * xavg;
        }
        /* copy-out: array(i,NY) = isign*buf(mirror(i),NY), mirror(i) = NX-i (0 -> NX).  An
           averaged pair is returned as (xavg, isign*xavg); the two pole points i = NX/2 and
           i = NX mirror onto themselves and simply change sign.  Ghost columns of the seam
           row receive the same final values. */
        top[NX / 2 - 1] = isign * top[NX / 2 - 1];
        top[NX - 1] = isign * top[NX - 1];
    }

    for (int b = 0; b < d->nblocks; ++b) {
        double *ab = a + (size_t)b * nx * ny;
        const int ilo = d->ilo[b], ihi = d->ihi[b], jlo = d->jlo[b], jhi = d->jhi[b];
        for (int j = jlo - d->nghost; j <= jhi + d->nghost; ++j)
            for (int i = ilo - d->nghost; i <= ihi + d->nghost; ++i) {
                int interior = (i >= ilo && i <= ihi && j >= jlo && j <= jhi);
                int ig = d->iglob0[b] + (i - ilo);
                int jg = d->jglob0[b] + (j - jlo);
                if (interior) {
                    if (tripole && (field_loc == 1 || field_loc == 3) && jg == NY) /* seam row written back */
                        ab[IX(i, j)] = g[(size_t)(NY - 1) * NX + (ig - 1)];
                    continue;
                }
                int outside = 0;
                double sgn = 1.0;
                if (ig < 1 || ig > NX) {
                    if (d->ew_type == BND_CYCLIC) ig = (ig < 1) ? ig + NX : ig - NX;
                    else outside = 1;
                }
                if (jg < 1) {
                    if (d->ns_type == BND_CYCLIC) jg += NY;
                    else outside = 1;
                } else if (jg > NY) {
                    if (d->ns_type == BND_CYCLIC) jg -= NY;
                    else if (tripole && !outside) {
                        /* u-fold mirror (ice_blocks.F90:423-424; ice_boundary.F90:1689-1722).
                           center:   ghost(ig, NY+k)  <- sign * a(NX-ig+1, NY-k+1)
                           NEcorner: ghost(ig, NY+k)  <- sign * a(NX-ig  , NY-k  )   (offsets 1,1) */
                        int k = jg - NY;
                        if (field_loc == 0) { ig = NX - ig + 1; jg = NY - k + 1; }
                        else if (field_loc == 1) { ig = NX - ig; jg = NY - k; }
                        else if (field_loc == 2) { ig = NX - ig; jg = NY - k + 1; }      /* E face: offsets (1,0) */
                        else { ig = NX - ig + 1; jg = NY - k; }                          /* N face: offsets (0,1) */
                        if (ig < 1) ig += NX;
                        if (ig > NX) ig -= NX;
                        sgn = isign;
                    } else outside = 1;
                }
                if (outside) {
                    if (have_fill) ab[IX(i, j)] = fill;
                    continue;
                }
                ab[IX(i, j)] = sgn * g[(size_t)(jg - 1) * NX + (ig - 1)];
            }
    }
    free(g);
}

/* ---------------------------------------------------------------------
 * Tripole symmetrisation of the stresses after the subcycle loop
 * (dynamics/ice_dyn_evp.F90:1321-1389 -> ice_HaloUpdate_stress,
 * infrastructure/comm/serial/ice_boundary.F90:7440-7825): the north ghost row of
 * array1 takes the mirrored top physical row of array2 (cell-centre, scalar, u-fold:
 * ghost(i, NY+1) <- a2(NX-i+1, NY)).  OUTSIDE the replaced region (evp() runs it on
 * the host arrays after the core returns); restated so that whole-evp() fixtures can
 * be compared on every cell.
 * ------------------------------------------------------------------- */
void evp_oracle_tripole_stress(const evp_oracle_domain *d, double *a1, const double *a2)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const int NX = d->nx_global, NY = d->ny_global;
    if (d->ns_type != BND_TRIPOLE) return;
    double *g = (double *)malloc(sizeof(double) * (size_t)NX * NY);
    gather_global(d, a2, g);
    for (int b = 0; b < d->nblocks; ++b) {
        double *ab = a1 + (size_t)b * nx * ny;
        const int ilo = d->ilo[b], ihi = d->ihi[b], jlo = d->jlo[b], jhi = d->jhi[b];
        const int j = jhi + 1;
        if (d->jglob0[b] + (j - jlo) != NY + 1) continue;
        for (int i = ilo - d->nghost; i <= ihi + d->nghost; ++i) {
            int ig = d->iglob0[b] + (i - ilo);
            if (ig < 1) ig += NX;
            if (ig > NX) ig -= NX;
            int im = NX - ig + 1;
            ab[IX(i, j)] = g[(size_t)(NY - 1) * NX + (im - 1)];
        }
    }
    free(g);
}

/* ---------------------------------------------------------------------
 * Static metric terms   dynamics/ice_dyn_shared.F90:384-388 (DminTarea),
 * :401-424 (dxhy, dyhx + halo with fillValue=c1), :426-441 (cxp, cyp, cxm, cym)
 * ------------------------------------------------------------------- */
void evp_oracle_metrics(const evp_oracle_domain *d, double deltaminEVP, const double *HTE,
                        const double *HTN, const double *tarea, double *cxp, double *cyp,
                        double *cxm, double *cym, double *dxhy, double *dyhx, double *DminTarea)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t nb = (size_t)nx * ny;
    for (int b = 0; b < d->nblocks; ++b) {
        const double *hte = HTE + b * nb, *htn = HTN + b * nb;
        for (int j = 1; j <= ny; ++j)
            for (int i = 1; i <= nx; ++i)
                DminTarea[b * nb + IX(i, j)] = deltaminEVP * tarea[b * nb + IX(i, j)];
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                dxhy[b * nb + IX(i, j)] = p5 * (hte[IX(i, j)] - hte[IX(i - 1, j)]);
                dyhx[b * nb + IX(i, j)] = p5 * (htn[IX(i, j)] - htn[IX(i, j - 1)]);
            }
        for (int j = d->jlo[b]; j <= d->jhi[b] + 1; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b] + 1; ++i) {
                cyp[b * nb + IX(i, j)] = (c1p5 * hte[IX(i, j)] - p5 * hte[IX(i - 1, j)]);
                cxp[b * nb + IX(i, j)] = (c1p5 * htn[IX(i, j)] - p5 * htn[IX(i, j - 1)]);
                cym[b * nb + IX(i, j)] = -(c1p5 * hte[IX(i - 1, j)] - p5 * hte[IX(i, j)]);
                cxm[b * nb + IX(i, j)] = -(c1p5 * htn[IX(i, j - 1)] - p5 * htn[IX(i, j)]);
            }
    }
    evp_oracle_halo_update(d, dxhy, 0, 1, 1, 1.0);
    evp_oracle_halo_update(d, dyhx, 0, 1, 1, 1.0);
}

/* ---------------------------------------------------------------------
 * visc_replpress   dynamics/ice_dyn_shared.F90:2446-2475
 * ------------------------------------------------------------------- */
static inline void visc_replpress(const evp_oracle_params *p, double strength, double DminArea,
                                  double Delta, double *zetax2, double *etax2, double *rep_prs)
{
    double tmpcalc = p->capping * (strength / fmax(Delta, DminArea)) +
                     (1.0 - p->capping) * (strength / (Delta + DminArea));
    *zetax2 = (1.0 + p->Ktens) * tmpcalc;
    *rep_prs = (1.0 - p->Ktens) * tmpcalc * Delta;
    *etax2 = p->epp2i * (*zetax2);
}

/* ---------------------------------------------------------------------
 * stress (one T-cell)   dynamics/ice_dyn_evp.F90:1539-1741
 *   strain_rates        dynamics/ice_dyn_shared.F90:2127-2161
 * str1..str8 are the eight planes of the reference's str(:,:,1:8).
 * ------------------------------------------------------------------- */
static inline void stress_cell(const evp_oracle_params *p, int nx, int i, int j,
                               const double *uvel, const double *vvel, const double *dxT,
                               const double *dyT, const double *dxhy, const double *dyhx,
                               const double *cxp, const double *cyp, const double *cxm,
                               const double *cym, const double *DminTarea, const double *strength,
                               double *sp1, double *sp2, double *sp3, double *sp4, double *sm1,
                               double *sm2, double *sm3, double *sm4, double *s121, double *s122,
                               double *s123, double *s124, double *str, size_t plane)
{
    const size_t c = IX(i, j);
    const double u_ij = uvel[IX(i, j)], u_im = uvel[IX(i - 1, j)], u_jm = uvel[IX(i, j - 1)],
                 u_mm = uvel[IX(i - 1, j - 1)];
    const double v_ij = vvel[IX(i, j)], v_im = vvel[IX(i - 1, j)], v_jm = vvel[IX(i, j - 1)],
                 v_mm = vvel[IX(i - 1, j - 1)];
    const double cxp_ = cxp[c], cyp_ = cyp[c], cxm_ = cxm[c], cym_ = cym[c];
    const double dxt = dxT[c], dyt = dyT[c];

    /* divergence  =  e_11 + e_22   (:2127-2134) */
    double divune = cyp_ * u_ij - dyt * u_im + cxp_ * v_ij - dxt * v_jm;
    double divunw = cym_ * u_im + dyt * u_ij + cxp_ * v_im - dxt * v_mm;
    double divusw = cym_ * u_mm + dyt * u_jm + cxm_ * v_mm + dxt * v_im;
    double divuse = cyp_ * u_jm - dyt * u_mm + cxm_ * v_jm + dxt * v_ij;
    /* tension strain rate  =  e_11 - e_22   (:2137-2144) */
    double tensionne = -cym_ * u_ij - dyt * u_im + cxm_ * v_ij + dxt * v_jm;
    double tensionnw = -cyp_ * u_im + dyt * u_ij + cxm_ * v_im + dxt * v_mm;
    double tensionsw = -cyp_ * u_mm + dyt * u_jm + cxp_ * v_mm - dxt * v_im;
    double tensionse = -cym_ * u_jm - dyt * u_mm + cxp_ * v_jm - dxt * v_ij;
    /* shearing strain rate  =  2*e_12   (:2147-2154) */
    double shearne = -cym_ * v_ij - dyt * v_im - cxm_ * u_ij - dxt * u_jm;
    double shearnw = -cyp_ * v_im + dyt * v_ij - cxm_ * u_im - dxt * u_mm;
    double shearsw = -cyp_ * v_mm + dyt * v_jm - cxp_ * u_mm + dxt * u_im;
    double shearse = -cym_ * v_jm - dyt * v_mm - cxp_ * u_jm + dxt * u_ij;
    /* Delta   (:2158-2161) */
    double Deltane = sqrt(divune * divune + p->e_factor * (tensionne * tensionne + shearne * shearne));
    double Deltanw = sqrt(divunw * divunw + p->e_factor * (tensionnw * tensionnw + shearnw * shearnw));
    double Deltasw = sqrt(divusw * divusw + p->e_factor * (tensionsw * tensionsw + shearsw * shearsw));
    double Deltase = sqrt(divuse * divuse + p->e_factor * (tensionse * tensionse + shearse * shearse));

    double zetax2ne, etax2ne, rep_prsne, zetax2nw, etax2nw, rep_prsnw;
    double zetax2sw, etax2sw, rep_prssw, zetax2se, etax2se, rep_prsse;
    visc_replpress(p, strength[c], DminTarea[c], Deltane, &zetax2ne, &etax2ne, &rep_prsne);
    visc_replpress(p, strength[c], DminTarea[c], Deltanw, &zetax2nw, &etax2nw, &rep_prsnw);
    visc_replpress(p, strength[c], DminTarea[c], Deltasw, &zetax2sw, &etax2sw, &rep_prssw);
    visc_replpress(p, strength[c], DminTarea[c], Deltase, &zetax2se, &etax2se, &rep_prsse);

    /* the stresses (ice_dyn_evp.F90:1585-1610); (1) NE, (2) NW, (3) SW, (4) SE */
    const double arlx1i = p->arlx1i, revp = p->revp, denom1 = p->denom1;
    sp1[c] = (sp1[c] * (1.0 - arlx1i * revp) + arlx1i * (zetax2ne * divune - rep_prsne)) * denom1;
    sp2[c] = (sp2[c] * (1.0 - arlx1i * revp) + arlx1i * (zetax2nw * divunw - rep_prsnw)) * denom1;
    sp3[c] = (sp3[c] * (1.0 - arlx1i * revp) + arlx1i * (zetax2sw * divusw - rep_prssw)) * denom1;
    sp4[c] = (sp4[c] * (1.0 - arlx1i * revp) + arlx1i * (zetax2se * divuse - rep_prsse)) * denom1;

    sm1[c] = (sm1[c] * (1.0 - arlx1i * revp) + arlx1i * etax2ne * tensionne) * denom1;
    sm2[c] = (sm2[c] * (1.0 - arlx1i * revp) + arlx1i * etax2nw * tensionnw) * denom1;
    sm3[c] = (sm3[c] * (1.0 - arlx1i * revp) + arlx1i * etax2sw * tensionsw) * denom1;
    sm4[c] = (sm4[c] * (1.0 - arlx1i * revp) + arlx1i * etax2se * tensionse) * denom1;

    s121[c] = (s121[c] * (1.0 - arlx1i * revp) + arlx1i * p5 * etax2ne * shearne) * denom1;
    s122[c] = (s122[c] * (1.0 - arlx1i * revp) + arlx1i * p5 * etax2nw * shearnw) * denom1;
    s123[c] = (s123[c] * (1.0 - arlx1i * revp) + arlx1i * p5 * etax2sw * shearsw) * denom1;
    s124[c] = (s124[c] * (1.0 - arlx1i * revp) + arlx1i * p5 * etax2se * shearse) * denom1;

    /* combinations of the stresses for the momentum equation (:1646-1689) */
    double ssigpn = sp1[c] + sp2[c];
    double ssigps = sp3[c] + sp4[c];
    double ssigpe = sp1[c] + sp4[c];
    double ssigpw = sp2[c] + sp3[c];
    double ssigp1 = (sp1[c] + sp3[c]) * p055;
    double ssigp2 = (sp2[c] + sp4[c]) * p055;

    double ssigmn = sm1[c] + sm2[c];
    double ssigms = sm3[c] + sm4[c];
    double ssigme = sm1[c] + sm4[c];
    double ssigmw = sm2[c] + sm3[c];
    double ssigm1 = (sm1[c] + sm3[c]) * p055;
    double ssigm2 = (sm2[c] + sm4[c]) * p055;

    double ssig12n = s121[c] + s122[c];
    double ssig12s = s123[c] + s124[c];
    double ssig12e = s121[c] + s124[c];
    double ssig12w = s122[c] + s123[c];
    double ssig121 = (s121[c] + s123[c]) * p111;
    double ssig122 = (s122[c] + s124[c]) * p111;

    double csigpne = p111 * sp1[c] + ssigp2 + p027 * sp3[c];
    double csigpnw = p111 * sp2[c] + ssigp1 + p027 * sp4[c];
    double csigpsw = p111 * sp3[c] + ssigp2 + p027 * sp1[c];
    double csigpse = p111 * sp4[c] + ssigp1 + p027 * sp2[c];

    double csigmne = p111 * sm1[c] + ssigm2 + p027 * sm3[c];
    double csigmnw = p111 * sm2[c] + ssigm1 + p027 * sm4[c];
    double csigmsw = p111 * sm3[c] + ssigm2 + p027 * sm1[c];
    double csigmse = p111 * sm4[c] + ssigm1 + p027 * sm2[c];

    double csig12ne = p222 * s121[c] + ssig122 + p055 * s123[c];
    double csig12nw = p222 * s122[c] + ssig121 + p055 * s124[c];
    double csig12sw = p222 * s123[c] + ssig122 + p055 * s121[c];
    double csig12se = p222 * s124[c] + ssig121 + p055 * s122[c];

    double str12ew = p5 * dxt * (p333 * ssig12e + p166 * ssig12w);
    double str12we = p5 * dxt * (p333 * ssig12w + p166 * ssig12e);
    double str12ns = p5 * dyt * (p333 * ssig12n + p166 * ssig12s);
    double str12sn = p5 * dyt * (p333 * ssig12s + p166 * ssig12n);

    const double dxhy_ = dxhy[c], dyhx_ = dyhx[c];
    /* for dF/dx (u momentum) (:1698-1717) */
    double strp_tmp = p25 * dyt * (p333 * ssigpn + p166 * ssigps);
    double strm_tmp = p25 * dyt * (p333 * ssigmn + p166 * ssigms);
    str[0 * plane + c] = -strp_tmp - strm_tmp - str12ew + dxhy_ * (-csigpne + csigmne) + dyhx_ * csig12ne;
    str[1 * plane + c] = strp_tmp + strm_tmp - str12we + dxhy_ * (-csigpnw + csigmnw) + dyhx_ * csig12nw;
    strp_tmp = p25 * dyt * (p333 * ssigps + p166 * ssigpn);
    strm_tmp = p25 * dyt * (p333 * ssigms + p166 * ssigmn);
    str[2 * plane + c] = -strp_tmp - strm_tmp + str12ew + dxhy_ * (-csigpse + csigmse) + dyhx_ * csig12se;
    str[3 * plane + c] = strp_tmp + strm_tmp + str12we + dxhy_ * (-csigpsw + csigmsw) + dyhx_ * csig12sw;
    /* for dF/dy (v momentum) (:1722-1739) */
    strp_tmp = p25 * dxt * (p333 * ssigpe + p166 * ssigpw);
    strm_tmp = p25 * dxt * (p333 * ssigme + p166 * ssigmw);
    str[4 * plane + c] = -strp_tmp + strm_tmp - str12ns - dyhx_ * (csigpne + csigmne) + dxhy_ * csig12ne;
    str[5 * plane + c] = strp_tmp - strm_tmp - str12sn - dyhx_ * (csigpse + csigmse) + dxhy_ * csig12se;
    strp_tmp = p25 * dxt * (p333 * ssigpw + p166 * ssigpe);
    strm_tmp = p25 * dxt * (p333 * ssigmw + p166 * ssigme);
    str[6 * plane + c] = -strp_tmp + strm_tmp + str12ns - dyhx_ * (csigpnw + csigmnw) + dxhy_ * csig12nw;
    str[7 * plane + c] = strp_tmp - strm_tmp + str12sn - dyhx_ * (csigpsw + csigmsw) + dxhy_ * csig12sw;
}

/* ---------------------------------------------------------------------
 * stepu (one U-cell)   dynamics/ice_dyn_shared.F90:925-966
 * ------------------------------------------------------------------- */
static inline void stepu_cell(const evp_oracle_params *p, int nx, int i, int j, const double *Cw,
                              const double *aiX, const double *str, size_t plane,
                              const double *uocn, const double *vocn, const double *waterx,
                              const double *watery, const double *forcex, const double *forcey,
                              const double *Umassdti, const double *fm, const double *uarear,
                              double *strintx, double *strinty, double *taubx, double *tauby,
                              const double *uvel_init, const double *vvel_init, double *uvel,
                              double *vvel, const double *TbU)
{
    const size_t c = IX(i, j);
    double uold = uvel[c];
    double vold = vvel[c];
    /* (magnitude of relative ocean current)*rhow*drag*aice */
    double vrel = aiX[c] * p->rhow * Cw[c] *
                  sqrt((uocn[c] - uold) * (uocn[c] - uold) + (vocn[c] - vold) * (vocn[c] - vold));
    double taux = vrel * waterx[c];
    double tauy = vrel * watery[c];
    double Cb = TbU[c] / (sqrt(uold * uold + vold * vold) + p->u0);
    double cca = (p->brlx + p->revp) * Umassdti[c] + vrel * p->cosw + Cb;
    double ccb = fm[c] + copysign(1.0, fm[c]) * vrel * p->sinw;
    double ab2 = cca * cca + ccb * ccb;
    /* divergence of the internal stress tensor (:948-951) */
    strintx[c] = uarear[c] * (str[0 * plane + IX(i, j)] + str[1 * plane + IX(i + 1, j)] +
                              str[2 * plane + IX(i, j + 1)] + str[3 * plane + IX(i + 1, j + 1)]);
    strinty[c] = uarear[c] * (str[4 * plane + IX(i, j)] + str[5 * plane + IX(i, j + 1)] +
                              str[6 * plane + IX(i + 1, j)] + str[7 * plane + IX(i + 1, j + 1)]);
    double cc1 = strintx[c] + forcex[c] + taux + Umassdti[c] * (p->brlx * uold + p->revp * uvel_init[c]);
    double cc2 = strinty[c] + forcey[c] + tauy + Umassdti[c] * (p->brlx * vold + p->revp * vvel_init[c]);
    uvel[c] = (cca * cc1 + ccb * cc2) / ab2;
    vvel[c] = (cca * cc2 - ccb * cc1) / ab2;
    taubx[c] = -uvel[c] * Cb;
    tauby[c] = -vvel[c] * Cb;
}

/* ---------------------------------------------------------------------
 * The subcycle loop   dynamics/ice_dyn_evp.F90:859-913
 *   do ksub = 1,ndte { for all blocks: stress ; stepu }  ; halo(uvel,vvel) NEcorner/vector
 * Dense sweep with masks instead of the reference's compressed index lists;
 * the index-list contract is dyn_prep2's (ice_dyn_shared.F90:740-789):
 *   T-cells: ilo..ihi+1 x jlo..jhi+1 where iceTmask ;  U-cells: ilo..ihi x jlo..jhi where iceUmask.
 * Pointer table `f` (each (nx_block,ny_block,nblocks)):
 *  0-11 stressp_1..4, stressm_1..4, stress12_1..4 (inout)
 *  12 strength 13 cdn_ocnU 14 aiU 15 uocnU 16 vocnU 17 waterxU 18 wateryU 19 forcexU 20 forceyU
 *  21 umassdti 22 fmU 23 strintxU 24 strintyU 25 TbU 26 taubxU 27 taubyU 28 uvel 29 vvel
 *  30 uvel_init 31 vvel_init
 * Static table `g`: 0 dxT 1 dyT 2 dxhy 3 dyhx 4 cxp 5 cyp 6 cxm 7 cym 8 DminTarea 9 uarear
 * ------------------------------------------------------------------- */
void evp_oracle_subcycle(const evp_oracle_domain *d, const evp_oracle_params *p, int ndte,
                         double *const *f, const double *const *g, const int32_t *iceTmask,
                         const int32_t *iceUmask)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t nb = (size_t)nx * ny;
    double *str = (double *)malloc(sizeof(double) * 8 * nb * (size_t)d->nblocks);

    for (int ksub = 0; ksub < ndte; ++ksub) {
#pragma omp parallel
        {
#pragma omp for schedule(static) collapse(2)
            for (int b = 0; b < d->nblocks; ++b)
                for (int j = 1; j <= ny; ++j) {
                    double *strb = str + 8 * nb * b;
                    /* str(:,:,:) = c0   (ice_dyn_evp.F90:1537) */
                    for (int k = 0; k < 8; ++k)
                        memset(strb + k * nb + IX(1, j), 0, sizeof(double) * nx);
                    if (j < d->jlo[b] || j > d->jhi[b] + 1) continue;
                    for (int i = d->ilo[b]; i <= d->ihi[b] + 1; ++i) {
                        if (!iceTmask[b * nb + IX(i, j)]) continue;
                        stress_cell(p, nx, i, j, f[28] + b * nb, f[29] + b * nb, g[0] + b * nb,
                                    g[1] + b * nb, g[2] + b * nb, g[3] + b * nb, g[4] + b * nb,
                                    g[5] + b * nb, g[6] + b * nb, g[7] + b * nb, g[8] + b * nb,
                                    f[12] + b * nb, f[0] + b * nb, f[1] + b * nb, f[2] + b * nb,
                                    f[3] + b * nb, f[4] + b * nb, f[5] + b * nb, f[6] + b * nb,
                                    f[7] + b * nb, f[8] + b * nb, f[9] + b * nb, f[10] + b * nb,
                                    f[11] + b * nb, strb, nb);
                    }
                }
            /* implicit barrier: all str planes complete before stepu reads neighbours */
#pragma omp for schedule(static) collapse(2)
            for (int b = 0; b < d->nblocks; ++b)
                for (int j = 1; j <= ny; ++j) {
                    if (j < d->jlo[b] || j > d->jhi[b]) continue;
                    const double *strb = str + 8 * nb * b;
                    for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                        if (!iceUmask[b * nb + IX(i, j)]) continue;
                        stepu_cell(p, nx, i, j, f[13] + b * nb, f[14] + b * nb, strb, nb,
                                   f[15] + b * nb, f[16] + b * nb, f[17] + b * nb, f[18] + b * nb,
                                   f[19] + b * nb, f[20] + b * nb, f[21] + b * nb, f[22] + b * nb,
                                   g[9] + b * nb, f[23] + b * nb, f[24] + b * nb, f[26] + b * nb,
                                   f[27] + b * nb, f[30] + b * nb, f[31] + b * nb, f[28] + b * nb,
                                   f[29] + b * nb, f[25] + b * nb);
                    }
                }
        }
        /* dyn_haloUpdate(uvel, vvel; field_loc_NEcorner, field_type_vector)  (:908-910) */
        evp_oracle_halo_update(d, f[28], 1, 1, 0, 0.0);
        evp_oracle_halo_update(d, f[29], 1, 1, 0, 0.0);
    }
    free(str);
}

/* ---------------------------------------------------------------------
 * "Next" tier: the two kernels evp() runs on the final
 * velocities right after the subcycle loop.
 *
 * deformations   dynamics/ice_dyn_shared.F90:1756-1860
 *   T-cells ilo..ihi+1 x jlo..jhi+1 where iceTmask (same list as stress);
 *   the five outputs are zero elsewhere (evp() zero-fills them, ice_dyn_evp.F90:385-393).
 * ------------------------------------------------------------------- */
void evp_oracle_deformations(const evp_oracle_domain *d, const evp_oracle_params *p,
                             const double *uvel, const double *vvel, const double *dxT,
                             const double *dyT, const double *dxU, const double *dyU,
                             const double *cxp, const double *cyp, const double *cxm,
                             const double *cym, const double *tarear, const int32_t *iceTmask,
                             double *vort, double *shear, double *divu, double *rdg_conv,
                             double *rdg_shear)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t nb = (size_t)nx * ny;
    memset(vort, 0, sizeof(double) * nb * d->nblocks);
    memset(shear, 0, sizeof(double) * nb * d->nblocks);
    memset(divu, 0, sizeof(double) * nb * d->nblocks);
    memset(rdg_conv, 0, sizeof(double) * nb * d->nblocks);
    memset(rdg_shear, 0, sizeof(double) * nb * d->nblocks);
    for (int b = 0; b < d->nblocks; ++b) {
        const double *u = uvel + b * nb, *v = vvel + b * nb;
        for (int j = d->jlo[b]; j <= d->jhi[b] + 1; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b] + 1; ++i) {
                const size_t c = b * nb + IX(i, j);
                if (!iceTmask[c]) continue;
                const double u_ij = u[IX(i, j)], u_im = u[IX(i - 1, j)], u_jm = u[IX(i, j - 1)],
                             u_mm = u[IX(i - 1, j - 1)];
                const double v_ij = v[IX(i, j)], v_im = v[IX(i - 1, j)], v_jm = v[IX(i, j - 1)],
                             v_mm = v[IX(i - 1, j - 1)];
                const double cxp_ = cxp[c], cyp_ = cyp[c], cxm_ = cxm[c], cym_ = cym[c];
                const double dxt = dxT[c], dyt = dyT[c];
                /* strain_rates, ice_dyn_shared.F90:2127-2161 */
                double divune = cyp_ * u_ij - dyt * u_im + cxp_ * v_ij - dxt * v_jm;
                double divunw = cym_ * u_im + dyt * u_ij + cxp_ * v_im - dxt * v_mm;
                double divusw = cym_ * u_mm + dyt * u_jm + cxm_ * v_mm + dxt * v_im;
                double divuse = cyp_ * u_jm - dyt * u_mm + cxm_ * v_jm + dxt * v_ij;
                double tensionne = -cym_ * u_ij - dyt * u_im + cxm_ * v_ij + dxt * v_jm;
                double tensionnw = -cyp_ * u_im + dyt * u_ij + cxm_ * v_im + dxt * v_mm;
                double tensionsw = -cyp_ * u_mm + dyt * u_jm + cxp_ * v_mm - dxt * v_im;
                double tensionse = -cym_ * u_jm - dyt * u_mm + cxp_ * v_jm - dxt * v_ij;
                double shearne = -cym_ * v_ij - dyt * v_im - cxm_ * u_ij - dxt * u_jm;
                double shearnw = -cyp_ * v_im + dyt * v_ij - cxm_ * u_im - dxt * u_mm;
                double shearsw = -cyp_ * v_mm + dyt * v_jm - cxp_ * u_mm + dxt * u_im;
                double shearse = -cym_ * v_jm - dyt * v_mm - cxp_ * u_jm + dxt * u_ij;
                double Deltane = sqrt(divune * divune + p->e_factor * (tensionne * tensionne + shearne * shearne));
                double Deltanw = sqrt(divunw * divunw + p->e_factor * (tensionnw * tensionnw + shearnw * shearnw));
                double Deltasw = sqrt(divusw * divusw + p->e_factor * (tensionsw * tensionsw + shearsw * shearsw));
                double Deltase = sqrt(divuse * divuse + p->e_factor * (tensionse * tensionse + shearse * shearse));
                /* :1827-1849 */
                divu[c] = p25 * (divune + divunw + divuse + divusw) * tarear[c];
                double tmp = p25 * (Deltane + Deltanw + Deltase + Deltasw) * tarear[c];
                rdg_conv[c] = -fmin(divu[c], 0.0);
                rdg_shear[c] = p5 * (tmp - fabs(divu[c]));
                double tsum = tensionne + tensionnw + tensionse + tensionsw;
                double ssum = shearne + shearnw + shearse + shearsw;
                shear[c] = p25 * tarear[c] * sqrt(tsum * tsum + ssum * ssum);
                const double *dyu = dyU + b * nb, *dxu = dxU + b * nb;
                double dvdxn = dyu[IX(i, j)] * v_ij - dyu[IX(i - 1, j)] * v_im;
                double dvdxs = dyu[IX(i, j - 1)] * v_jm - dyu[IX(i - 1, j - 1)] * v_mm;
                double dudye = dxu[IX(i, j)] * u_ij - dxu[IX(i, j - 1)] * u_jm;
                double dudyw = dxu[IX(i - 1, j)] * u_im - dxu[IX(i - 1, j - 1)] * u_mm;
                vort[c] = p5 * tarear[c] * (dvdxn + dvdxs - dudye - dudyw);
            }
    }
}

/* ---------------------------------------------------------------------
 * dyn_finish   dynamics/ice_dyn_shared.F90:1291-1365
 *   ice-ocean stress from the final velocities on U-cells where iceUmask
 *   (ilo..ihi x jlo..jhi); strocnx/y are inout (other cells keep their value).
 * ------------------------------------------------------------------- */
void evp_oracle_dyn_finish(const evp_oracle_domain *d, const evp_oracle_params *p, const double *Cw,
                           const double *uvel, const double *vvel, const double *uocn,
                           const double *vocn, const double *aiX, const double *fm,
                           const int32_t *iceUmask, double *strocnx, double *strocny)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t nb = (size_t)nx * ny;
    for (int b = 0; b < d->nblocks; ++b)
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const size_t c = b * nb + IX(i, j);
                if (!iceUmask[c]) continue;
                double du = uocn[c] - uvel[c], dv = vocn[c] - vvel[c];
                double vrel = p->rhow * Cw[c] * sqrt(du * du + dv * dv);
                vrel = vrel * aiX[c];
                strocnx[c] = vrel * ((uocn[c] - uvel[c]) * p->cosw - (vocn[c] - vvel[c]) * p->sinw * copysign(1.0, fm[c]));
                strocny[c] = vrel * ((vocn[c] - vvel[c]) * p->cosw + (uocn[c] - uvel[c]) * p->sinw * copysign(1.0, fm[c]));
            }
}

/* =====================================================================
 * Preparation phase of evp(): everything between the entry of
 * evp() and the subcycle loop on the B grid, dynamics/ice_dyn_evp.F90:383-840,
 * except icepack_ice_strength (Icepack) and the seabed stress factor (exp()):
 *   dyn_prep1            ice_dyn_shared.F90:496-576
 *   halo updates         ice_dyn_evp.F90:413-428, 466-470, 727-733
 *   grid_average_X2Y     infrastructure/ice_grid.F90:3983-3984 ('S' T->U = X2YS 'NE' with
 *                        tarea, hm: :4183-4204) and :3957-3958 ('F' T->U = X2YF 'NE' with
 *                        tarea, uarea: :4650-4666)
 *   dyn_prep2            ice_dyn_shared.F90:586-839
 * T-grid inputs are const: their ghost cells are refreshed on private copies,
 * as the reference refreshes them on its module arrays.
 * ===================================================================== */
typedef struct {
    double dt, rhoi, rhos, gravit, dyn_area_min, dyn_mass_min, cosw, sinw;
    int ssh_coupled;   /* ssh_stress == 'coupled' (else 'geostrophic') */
} evp_oracle_prep_params;

enum { /* indices into the T-grid input table */
    PT_AICE = 0, PT_VICE, PT_VSNO, PT_AICE_INIT, PT_CDN_OCN, PT_UOCN, PT_VOCN, PT_SS_TLTX, PT_SS_TLTY,
    PT_STRAIRX, PT_STRAIRY, PT_COUNT
};
enum { /* indices into the U-grid output table */
    PU_AIU = 0, PU_CDN_OCNU, PU_UOCNU, PU_VOCNU, PU_UMASSDTI, PU_FM, PU_WATERX, PU_WATERY, PU_FORCEX,
    PU_FORCEY, PU_UVEL_INIT, PU_VVEL_INIT, PU_STRTLTX, PU_STRTLTY, PU_STRAIRXU, PU_STRAIRYU,
    PU_TMASS, PU_UMASS, PU_COUNT
};

static void avg_T2U_S(const evp_oracle_domain *d, const double *w1, const double *tarea, const double *hm,
                      double *w2)
{
    const int nx = d->nx_block;
    const size_t nb = (size_t)nx * d->ny_block;
    memset(w2, 0, sizeof(double) * nb * d->nblocks);
    for (int b = 0; b < d->nblocks; ++b) {
        const double *a = w1 + b * nb, *wt = tarea + b * nb, *m = hm + b * nb;
        double *o = w2 + b * nb;
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const double wtmp = (m[IX(i, j)] * wt[IX(i, j)] + m[IX(i + 1, j)] * wt[IX(i + 1, j)] +
                                     m[IX(i, j + 1)] * wt[IX(i, j + 1)] + m[IX(i + 1, j + 1)] * wt[IX(i + 1, j + 1)]);
                if (wtmp != 0.0)
                    o[IX(i, j)] = (m[IX(i, j)] * a[IX(i, j)] * wt[IX(i, j)] +
                                   m[IX(i + 1, j)] * a[IX(i + 1, j)] * wt[IX(i + 1, j)] +
                                   m[IX(i, j + 1)] * a[IX(i, j + 1)] * wt[IX(i, j + 1)] +
                                   m[IX(i + 1, j + 1)] * a[IX(i + 1, j + 1)] * wt[IX(i + 1, j + 1)]) / wtmp;
            }
    }
}

static void avg_T2U_F(const evp_oracle_domain *d, const double *w1, const double *tarea, const double *uarea,
                      double *w2)
{
    const int nx = d->nx_block;
    const size_t nb = (size_t)nx * d->ny_block;
    memset(w2, 0, sizeof(double) * nb * d->nblocks);
    for (int b = 0; b < d->nblocks; ++b) {
        const double *a = w1 + b * nb, *wt = tarea + b * nb, *w2a = uarea + b * nb;
        double *o = w2 + b * nb;
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i)
                o[IX(i, j)] = p25 * (a[IX(i, j)] * wt[IX(i, j)] + a[IX(i + 1, j)] * wt[IX(i + 1, j)] +
                                     a[IX(i, j + 1)] * wt[IX(i, j + 1)] + a[IX(i + 1, j + 1)] * wt[IX(i + 1, j + 1)]) /
                              w2a[IX(i, j)];
    }
}

/* sig: 12 stress arrays (inout); uvel,vvel,iceUmask,strintx/y,strocnx/y: inout; iceTmask, U: out */
void evp_oracle_prep(const evp_oracle_domain *d, const evp_oracle_prep_params *p, const int *tmask,
                     const int *umask, const double *hm, const double *tarea, const double *uarea,
                     const double *fcor, const double *const *T, double *const *sig, double *uvel,
                     double *vvel, int *iceUmask, double *strintx, double *strinty, double *strocnx,
                     double *strocny, int *iceTmask, double *const *U)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t nb = (size_t)nx * ny, n = nb * d->nblocks;
    double *t[PT_COUNT];
    for (int k = 0; k < PT_COUNT; ++k) {
        t[k] = (double *)malloc(sizeof(double) * n);
        memcpy(t[k], T[k], sizeof(double) * n);
    }
    double *tmass = U[PU_TMASS], *umass = U[PU_UMASS];
    double *maskd = (double *)malloc(sizeof(double) * n);
    unsigned char *tmphm = (unsigned char *)malloc(n);

    /* dyn_prep1 (:545-575): mass on every cell, extent mask on the physical cells */
    for (int b = 0; b < d->nblocks; ++b) {
        const size_t o = b * nb;
        for (int j = 1; j <= ny; ++j)
            for (int i = 1; i <= nx; ++i) {
                const size_t c = o + IX(i, j);
                tmass[c] = tmask[c] ? (p->rhoi * t[PT_VICE][c] + p->rhos * t[PT_VSNO][c]) : 0.0;
                tmphm[c] = tmask[c] && (t[PT_AICE][c] > p->dyn_area_min) && (tmass[c] > p->dyn_mass_min);
                iceTmask[c] = 0;
            }
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                int any = 0;
                for (int dj = -1; dj <= 1; ++dj)
                    for (int di = -1; di <= 1; ++di) any |= tmphm[o + IX(i + di, j + dj)];
                iceTmask[o + IX(i, j)] = any && tmask[o + IX(i, j)];
            }
    }
    /* ice_HaloUpdate(iceTmask, center, scalar)  ice_dyn_evp.F90:413-416 */
    for (size_t c = 0; c < n; ++c) maskd[c] = (double)iceTmask[c];
    evp_oracle_halo_update(d, maskd, 0, 0, 0, 0.0);
    for (size_t c = 0; c < n; ++c) iceTmask[c] = maskd[c] != 0.0;

    /* halo updates of the T-grid fields (:422-428): scalars, then vectors */
    evp_oracle_halo_update(d, tmass, 0, 0, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_AICE_INIT], 0, 0, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_CDN_OCN], 0, 0, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_UOCN], 0, 1, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_VOCN], 0, 1, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_SS_TLTX], 0, 1, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_SS_TLTY], 0, 1, 0, 0.0);

    /* T -> U, state-masked (:430-436) */
    double *ss_tltxU = (double *)malloc(sizeof(double) * n), *ss_tltyU = (double *)malloc(sizeof(double) * n);
    avg_T2U_S(d, tmass, tarea, hm, umass);
    avg_T2U_S(d, t[PT_AICE_INIT], tarea, hm, U[PU_AIU]);
    avg_T2U_S(d, t[PT_CDN_OCN], tarea, hm, U[PU_CDN_OCNU]);
    avg_T2U_S(d, t[PT_UOCN], tarea, hm, U[PU_UOCNU]);
    avg_T2U_S(d, t[PT_VOCN], tarea, hm, U[PU_VOCNU]);
    avg_T2U_S(d, t[PT_SS_TLTX], tarea, hm, ss_tltxU);
    avg_T2U_S(d, t[PT_SS_TLTY], tarea, hm, ss_tltyU);
    /* wind stress, calc_strair branch (:465-470): halo, then flux average */
    evp_oracle_halo_update(d, t[PT_STRAIRX], 0, 1, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_STRAIRY], 0, 1, 0, 0.0);
    avg_T2U_F(d, t[PT_STRAIRX], tarea, uarea, U[PU_STRAIRXU]);
    avg_T2U_F(d, t[PT_STRAIRY], tarea, uarea, U[PU_STRAIRYU]);

    /* dyn_prep2 (:697-838) */
    for (int b = 0; b < d->nblocks; ++b) {
        const size_t o = b * nb;
        for (int j = 1; j <= ny; ++j)
            for (int i = 1; i <= nx; ++i) {
                const size_t c = o + IX(i, j);
                U[PU_WATERX][c] = U[PU_WATERY][c] = U[PU_FORCEX][c] = U[PU_FORCEY][c] = U[PU_UMASSDTI][c] = 0.0;
                if (!iceTmask[c])
                    for (int k = 0; k < 12; ++k) sig[k][c] = 0.0;
            }
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const size_t c = o + IX(i, j);
                const int old = iceUmask[c];
                iceUmask[c] = umask[c] && (U[PU_AIU][c] > p->dyn_area_min) && (umass[c] > p->dyn_mass_min);
                if (iceUmask[c]) {
                    if (!old) { uvel[c] = U[PU_UOCNU][c]; vvel[c] = U[PU_VOCNU][c]; }
                } else {
                    uvel[c] = vvel[c] = 0.0;
                    strintx[c] = strinty[c] = strocnx[c] = strocny[c] = 0.0;
                }
                U[PU_UVEL_INIT][c] = uvel[c];
                U[PU_VVEL_INIT][c] = vvel[c];
                if (!iceUmask[c]) continue;
                U[PU_UMASSDTI][c] = umass[c] / p->dt;
                const double fm = fcor[c] * umass[c];
                U[PU_FM][c] = fm;
                const double sgn = copysign(1.0, fm);
                U[PU_WATERX][c] = U[PU_UOCNU][c] * p->cosw - U[PU_VOCNU][c] * p->sinw * sgn;
                U[PU_WATERY][c] = U[PU_VOCNU][c] * p->cosw + U[PU_UOCNU][c] * p->sinw * sgn;
                if (p->ssh_coupled) {
                    U[PU_STRTLTX][c] = -p->gravit * umass[c] * ss_tltxU[c];
                    U[PU_STRTLTY][c] = -p->gravit * umass[c] * ss_tltyU[c];
                } else {
                    U[PU_STRTLTX][c] = -fm * U[PU_VOCNU][c];
                    U[PU_STRTLTY][c] = fm * U[PU_UOCNU][c];
                }
                U[PU_FORCEX][c] = U[PU_STRAIRXU][c] + U[PU_STRTLTX][c];
                U[PU_FORCEY][c] = U[PU_STRAIRYU][c] + U[PU_STRTLTY][c];
            }
    }
    /* velocity halo before the loop (:729-732) */
    evp_oracle_halo_update(d, uvel, 1, 1, 0, 0.0);
    evp_oracle_halo_update(d, vvel, 1, 1, 0, 0.0);

    for (int k = 0; k < PT_COUNT; ++k) free(t[k]);
    free(maskd); free(tmphm); free(ss_tltxU); free(ss_tltyU);
}

/* =====================================================================
 * Seabed stress factor, LKD method: seabed_stress_factor_LKD, dynamics/ice_dyn_shared.F90:1386-1460
 * (grid_neighbor_min / _max at the U point: infrastructure/ice_grid.F90:4974, 5005); call site
 * ice_dyn_evp.F90:783-790, after dyn_prep2 has zeroed TbU (:706).  exp() is libm's, as in the reference.
 * aice / vice / hwater with current ghost cells (the caller's, like the reference's module arrays).
 * ===================================================================== */
void evp_oracle_seabed_lkd(const evp_oracle_domain *d, double k1, double k2, double alphab, double threshold_hw,
                           const double *aice, const double *vice, const double *hwater,
                           const int32_t *iceUmask, double *TbU)
{
    const int nx = d->nx_block;
    const size_t plane = (size_t)nx * d->ny_block;
    for (int b = 0; b < d->nblocks; ++b) {
        for (size_t c = b * plane; c < (b + 1) * plane; ++c) TbU[c] = 0.0;
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const size_t c = b * plane + (size_t)(j - 1) * nx + (i - 1);
                if (!iceUmask[c]) continue;
                const size_t c1 = c + 1, c2 = c + nx, c3 = c + nx + 1;
                const double hwu = fmin(fmin(fmin(hwater[c], hwater[c1]), hwater[c2]), hwater[c3]);
                const double docalc_tbu = hwu < threshold_hw ? 1.0 : 0.0;
                const double au = fmax(fmax(fmax(aice[c], aice[c1]), aice[c2]), aice[c3]);
                const double hu = fmax(fmax(fmax(vice[c], vice[c1]), vice[c2]), vice[c3]);
                const double hcu = au * hwu / k1;
                TbU[c] = docalc_tbu * k2 * fmax(0.0, (hu - hcu)) * exp(-alphab * (1.0 - au));
            }
    }
}

/* =====================================================================
 * Seabed stress factor, probabilistic method: seabed_stress_factor_prob, dynamics/ice_dyn_shared.F90:1475-1683
 * (B grid: TbU = grid_neighbor_max(Tbt, 'U'), :1648-1655).  Per ice T-cell with atot > 0.05 and hwater < 50 m: a
 * log-normal ice-thickness distribution (100 categories of 0.5 m) against a normal bathymetry distribution (100
 * categories over +-3 sigma_b, sigma_b = 2.5 m); exp() / log() are libm's, as in the reference.  The loop that would
 * cut x_kmax at the last ice-holding category (`do n = ncat,-1,1`, :1583-1589) never executes, so cut = x_k(100).
 * aicen / vicen: (nx, ny, ncat, nblocks), i fastest.
 * ===================================================================== */
/* loc 1: U (B grid), 2: E, 3: N (C grid, :1656-1676) */
static void seabed_prob_at(const evp_oracle_domain *d, int loc, int ncat, double alphab, double rhoi, double rhow, double gravit,
                           double pi, double puny, const double *aicen, const double *vicen, const double *hwater,
                           const int32_t *iceTmask, const int32_t *iceUmask, double *TbU)
{
    enum { NI = 100, NB = 100 };
    const double max_depth = 50.0, mu_s = 0.1, sigma_b = 2.5, c0 = 0.0, c1 = 1.0, c2 = 2.0, c3 = 3.0, c6 = 6.0, p5 = 0.5;
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t plane = (size_t)nx * ny;
    double *Tbt = (double *)calloc(plane, sizeof(double));
    for (int b = 0; b < d->nblocks; ++b) {
        for (size_t c = 0; c < plane; ++c) Tbt[c] = 0.0;
        /* T-cells of the list dyn_prep2 builds: ilo..ihi+1 x jlo..jhi+1 where iceTmask (ice_dyn_shared.F90:740-749) */
        for (int j = d->jlo[b]; j <= d->jhi[b] + 1; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b] + 1; ++i) {
                const size_t cl = (size_t)(j - 1) * nx + (i - 1), c = b * plane + cl;
                if (!iceTmask[c]) continue;
                double atot = 0.0;
                for (int n = 0; n < ncat; ++n) atot += aicen[((size_t)b * ncat + n) * plane + cl];
                if (!(atot > 0.05 && hwater[c] < max_depth)) continue;
                const double mu_b = hwater[c];
                const double wid_i = max_depth / NI, wid_b = c6 * sigma_b / NB;
                double x_k[NI], y_n[NB], P_x[NI], P_y[NB];
                for (int k = 1; k <= NI; ++k) x_k[k - 1] = wid_i * ((double)k - p5);
                for (int k = 1; k <= NB; ++k) y_n[k - 1] = (mu_b - c3 * sigma_b) + ((double)k - p5) * (c6 * sigma_b / NB);
                double m_i = 0.0, v_i = c0;
                for (int n = 0; n < ncat; ++n) m_i += vicen[((size_t)b * ncat + n) * plane + cl];
                for (int n = 0; n < ncat; ++n) {
                    const double vc = vicen[((size_t)b * ncat + n) * plane + cl], ac = aicen[((size_t)b * ncat + n) * plane + cl];
                    v_i = v_i + vc * vc / (fmax(ac, puny));
                }
                v_i = fmax((v_i - m_i * m_i), puny);
                const double mu_i = log(m_i / sqrt(c1 + v_i / (m_i * m_i)));
                const double sigma_i = sqrt(log(c1 + v_i / (m_i * m_i)));
                double x_kmax = exp(mu_i + sqrt(c2 * sigma_i) * 1.9430);
                const double cut = x_k[NI - 1];
                x_kmax = fmin(cut, x_kmax);
                for (int k = 0; k < NI; ++k) {
                    const double lx = log(x_k[k]) - mu_i;
                    const double g_k = exp(-(lx * lx) / (c2 * (sigma_i * sigma_i))) / (x_k[k] * sigma_i * sqrt(c2 * pi));
                    P_x[k] = g_k * wid_i;
                }
                for (int k = 0; k < NB; ++k) {
                    const double dy = y_n[k] - mu_b;
                    const double b_n = exp(-(dy * dy) / (c2 * (sigma_b * sigma_b))) / (sigma_b * sqrt(c2 * pi));
                    P_y[k] = b_n * wid_b;
                }
                for (int k = 0; k < NI; ++k)
                    if (x_k[k] > x_kmax) P_x[k] = c0;
                double tsum = 0.0;
                for (int n = 0; n < NI; ++n) {
                    int ii = 0;
                    for (int k = 0; k < NB; ++k) ii += (y_n[k] <= rhoi * x_k[n] / rhow) ? 1 : 0;
                    double tb = c0;
                    if (ii != 0) {
                        double sm = 0.0;
                        for (int k = 0; k < ii; ++k) sm += P_y[k] * (rhoi * x_k[n] - rhow * y_n[k]);
                        tb = fmax(mu_s * gravit * P_x[n] * sm, c0);
                    }
                    tsum += tb;
                }
                Tbt[cl] = tsum * exp(-alphab * (c1 - atot));
            }
        for (size_t c = b * plane; c < (b + 1) * plane; ++c) TbU[c] = 0.0;
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const size_t cl = (size_t)(j - 1) * nx + (i - 1), c = b * plane + cl;
                if (!iceUmask[c]) continue;
                /* grid_neighbor_max (ice_grid.F90:5004-5009): U max(a(i,j), a(i+1,j), a(i,j+1), a(i+1,j+1)); E, N: two cells */
                if (loc == 1) TbU[c] = fmax(fmax(fmax(Tbt[cl], Tbt[cl + 1]), Tbt[cl + nx]), Tbt[cl + nx + 1]);
                else TbU[c] = fmax(Tbt[cl], Tbt[loc == 2 ? cl + 1 : cl + nx]);
            }
    }
    free(Tbt);
}
void evp_oracle_seabed_prob(const evp_oracle_domain *d, int ncat, double alphab, double rhoi, double rhow, double gravit,
                            double pi, double puny, const double *aicen, const double *vicen, const double *hwater,
                            const int32_t *iceTmask, const int32_t *iceUmask, double *TbU)
{
    seabed_prob_at(d, 1, ncat, alphab, rhoi, rhow, gravit, pi, puny, aicen, vicen, hwater, iceTmask, iceUmask, TbU);
}
/* grid_ice = 'C': TbE, TbN (call site ice_dyn_evp.F90:816-827) */
void evp_oracle_seabed_prob_c(const evp_oracle_domain *d, int ncat, double alphab, double rhoi, double rhow, double gravit,
                              double pi, double puny, const double *aicen, const double *vicen, const double *hwater,
                              const int32_t *iceTmask, const int32_t *iceEmask, const int32_t *iceNmask, double *TbE,
                              double *TbN)
{
    seabed_prob_at(d, 2, ncat, alphab, rhoi, rhow, gravit, pi, puny, aicen, vicen, hwater, iceTmask, iceEmask, TbE);
    seabed_prob_at(d, 3, ncat, alphab, rhoi, rhow, gravit, pi, puny, aicen, vicen, hwater, iceTmask, iceNmask, TbN);
}

/* =====================================================================
 * C-grid EVP subcycle: evp()'s loop for grid_ice = 'C',
 * dynamics/ice_dyn_evp.F90:938-1099.  u lives at the east face (E), v at the north
 * face (N); stresses at cell centres (T) and corners (U).  One subcycle =
 *   strain_rates_U -> halo(shearU) -> stressC_T -> halo(zetax2T, etax2T, stresspT, stressmT)
 *   -> T->U average of etax2T (or of the strength) -> stressC_U -> halo(stress12U)
 *   -> div_stress_Ex / div_stress_Ny -> stepu_C / stepv_C -> halo(uvelE), halo(vvelN)
 *   -> uvelN = E->N average, vvelE = N->E average (times the face masks) -> their halos
 *   -> uvel, vvel = face -> corner averages (times uvm) -> halo.
 *
 * Field table `f` (in/out):
 *   0 uvelE 1 vvelE 2 uvelN 3 vvelN 4 uvel 5 vvel 6 stresspT 7 stressmT 8 stress12T 9 stress12U
 *  10 strintxE 11 strintyN 12 taubxE 13 taubyN 14 zetax2T 15 etax2T 16 etax2U 17 shearU 18 deltaU
 *   (14-18 are work arrays of evp(), zeroed at its entry, :351-361; kept because the reference
 *    keeps them and the ridging diagnostics read shearU afterwards)
 * Input table `in`:
 *   0 strength 1 cdn_ocnE 2 aiE 3 uocnE 4 vocnE 5 waterxE 6 forcexE 7 emassdti 8 fmE 9 uvelE_init
 *  10 TbE 11 rheofactE 12 cdn_ocnN 13 aiN 14 uocnN 15 vocnN 16 wateryN 17 forceyN 18 nmassdti
 *  19 fmN 20 vvelN_init 21 TbN 22 rheofactN
 * Static table `g`:
 *   0 dxT 1 dyT 2 dxU 3 dyU 4 dxE 5 dyE 6 dxN 7 dyN 8 uarea 9 tarea 10 earea 11 narea 12 earear
 *  13 narear 14 epm 15 npm 16 uvm 17 hm 18 DminTarea 19 ratiodxN 20 ratiodxNr 21 ratiodyE 22 ratiodyEr
 * ===================================================================== */

/* grid_average_X2YA, the four directions the loop uses (infrastructure/ice_grid.F90:4388-4606):
 * dir 0 'NW' (E->N), 1 'SE' (N->E), 2 'N' (E->U), 3 'E' (N->U).  The whole output array is zeroed first
 * (:4412), interior cells with a non-zero weight sum are computed, then the result is multiplied by `mask`
 * on every cell (ice_dyn_evp.F90:1074-1075, 1088-1089). */
static void avg_A(const evp_oracle_domain *d, int dir, const double *w1, const double *wght, const double *mask,
                  double *w2)
{
    const int nx = d->nx_block;
    const size_t nb = (size_t)nx * d->ny_block;
    memset(w2, 0, sizeof(double) * nb * d->nblocks);
    for (int b = 0; b < d->nblocks; ++b) {
        const double *a = w1 + b * nb, *w = wght + b * nb;
        double *o = w2 + b * nb;
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                double wtmp;
                switch (dir) {
                case 0: /* NW */
                    wtmp = (w[IX(i - 1, j)] + w[IX(i, j)] + w[IX(i - 1, j + 1)] + w[IX(i, j + 1)]);
                    if (wtmp != 0.0)
                        o[IX(i, j)] = (a[IX(i - 1, j)] * w[IX(i - 1, j)] + a[IX(i, j)] * w[IX(i, j)] +
                                       a[IX(i - 1, j + 1)] * w[IX(i - 1, j + 1)] + a[IX(i, j + 1)] * w[IX(i, j + 1)]) / wtmp;
                    break;
                case 1: /* SE */
                    wtmp = (w[IX(i, j - 1)] + w[IX(i + 1, j - 1)] + w[IX(i, j)] + w[IX(i + 1, j)]);
                    if (wtmp != 0.0)
                        o[IX(i, j)] = (a[IX(i, j - 1)] * w[IX(i, j - 1)] + a[IX(i + 1, j - 1)] * w[IX(i + 1, j - 1)] +
                                       a[IX(i, j)] * w[IX(i, j)] + a[IX(i + 1, j)] * w[IX(i + 1, j)]) / wtmp;
                    break;
                case 2: /* N */
                    wtmp = (w[IX(i, j)] + w[IX(i, j + 1)]);
                    if (wtmp != 0.0)
                        o[IX(i, j)] = (a[IX(i, j)] * w[IX(i, j)] + a[IX(i, j + 1)] * w[IX(i, j + 1)]) / wtmp;
                    break;
                default: /* E */
                    wtmp = (w[IX(i, j)] + w[IX(i + 1, j)]);
                    if (wtmp != 0.0)
                        o[IX(i, j)] = (a[IX(i, j)] * w[IX(i, j)] + a[IX(i + 1, j)] * w[IX(i + 1, j)]) / wtmp;
                    break;
                }
            }
    }
    const double *m = mask;
    for (size_t k = 0; k < nb * d->nblocks; ++k) w2[k] = w2[k] * m[k];
}

void evp_oracle_cgrid_subcycle(const evp_oracle_domain *d, const evp_oracle_params *p, int ndte,
                               int avg_strength, double *const *f, const double *const *in,
                               const double *const *g, const int32_t *iceTmask, const int32_t *iceUmask,
                               const int32_t *iceEmask, const int32_t *iceNmask)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t nb = (size_t)nx * ny, ntot = nb * d->nblocks;
    double *divergU = (double *)malloc(sizeof(double) * ntot), *tensionU = (double *)malloc(sizeof(double) * ntot);
    double *strengthU = (double *)calloc(ntot, sizeof(double));
    const double relax = 1.0 - p->arlx1i * p->revp;      /* (c1-arlx1i*revp) */

    for (int ksub = 0; ksub < ndte; ++ksub) {
        /* ---- strain_rates_U  (ice_dyn_shared.F90:2341-2444): strain rates * area at the corners ---- */
        memset(divergU, 0, sizeof(double) * ntot);
        memset(tensionU, 0, sizeof(double) * ntot);
        memset(f[17], 0, sizeof(double) * ntot);
        memset(f[18], 0, sizeof(double) * ntot);
        for (int b = 0; b < d->nblocks; ++b) {
            const double *uE = f[0] + b * nb, *vE = f[1] + b * nb, *uN = f[2] + b * nb, *vN = f[3] + b * nb;
            const double *uU = f[4] + b * nb, *vU = f[5] + b * nb;
            const double *dxU = g[2] + b * nb, *dyU = g[3] + b * nb, *dxE = g[4] + b * nb, *dyN = g[7] + b * nb;
            const double *epm = g[14] + b * nb, *npm = g[15] + b * nb;
            const double *rxN = g[19] + b * nb, *rxNr = g[20] + b * nb, *ryE = g[21] + b * nb, *ryEr = g[22] + b * nb;
            for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
                for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                    if (!iceUmask[b * nb + IX(i, j)]) continue;
                    const size_t c = IX(i, j), e = IX(i + 1, j), n = IX(i, j + 1);
                    double uNip1j = uN[e] * npm[e] + (npm[c] - npm[e]) * npm[c] * rxN[c] * uN[c];
                    double uNij = uN[c] * npm[c] + (npm[e] - npm[c]) * npm[e] * rxNr[c] * uN[e];
                    double vEijp1 = vE[n] * epm[n] + (epm[c] - epm[n]) * epm[c] * ryE[c] * vE[c];
                    double vEij = vE[c] * epm[c] + (epm[n] - epm[c]) * epm[n] * ryEr[c] * vE[n];
                    double dv = dyU[c] * (uNip1j - uNij) + uU[c] * (dyN[e] - dyN[c]) + dxU[c] * (vEijp1 - vEij) +
                                vU[c] * (dxE[n] - dxE[c]);
                    double tn = dyU[c] * (uNip1j - uNij) - uU[c] * (dyN[e] - dyN[c]) - dxU[c] * (vEijp1 - vEij) +
                                vU[c] * (dxE[n] - dxE[c]);
                    double uEijp1 = uE[n] * epm[n] + (epm[c] - epm[n]) * epm[c] * ryE[c] * uE[c];
                    double uEij = uE[c] * epm[c] + (epm[n] - epm[c]) * epm[n] * ryEr[c] * uE[n];
                    double vNip1j = vN[e] * npm[e] + (npm[c] - npm[e]) * npm[c] * rxN[c] * vN[c];
                    double vNij = vN[c] * npm[c] + (npm[e] - npm[c]) * npm[e] * rxNr[c] * vN[e];
                    double sh = dxU[c] * (uEijp1 - uEij) - uU[c] * (dxE[n] - dxE[c]) + dyU[c] * (vNip1j - vNij) -
                                vU[c] * (dyN[e] - dyN[c]);
                    divergU[b * nb + c] = dv;
                    tensionU[b * nb + c] = tn;
                    f[17][b * nb + c] = sh;
                    f[18][b * nb + c] = sqrt(dv * dv + p->e_factor * (tn * tn + sh * sh));
                }
        }
        evp_oracle_halo_update(d, f[17], 1, 0, 0, 0.0);      /* shearU: NE corner, scalar (ice_dyn_evp.F90:965-967) */

        /* ---- stressC_T  (ice_dyn_evp.F90:1758-1880; strain_rates_Tdt ice_dyn_shared.F90:2291-2339) ---- */
        for (int b = 0; b < d->nblocks; ++b) {
            const double *uE = f[0] + b * nb, *vN = f[3] + b * nb, *shU = f[17] + b * nb;
            const double *dxT = g[0] + b * nb, *dyT = g[1] + b * nb, *dyE = g[5] + b * nb, *dxN = g[6] + b * nb;
            const double *uarea = g[8] + b * nb, *Dmin = g[18] + b * nb, *strength = in[0] + b * nb;
            for (int j = d->jlo[b]; j <= d->jhi[b] + 1; ++j)
                for (int i = d->ilo[b]; i <= d->ihi[b] + 1; ++i) {
                    if (!iceTmask[b * nb + IX(i, j)]) continue;
                    const size_t c = IX(i, j), w = IX(i - 1, j), s = IX(i, j - 1), sw = IX(i - 1, j - 1);
                    double divT = dyE[c] * uE[c] - dyE[w] * uE[w] + dxN[c] * vN[c] - dxN[s] * vN[s];
                    double tensionT = (dyT[c] * dyT[c]) * (uE[c] / dyE[c] - uE[w] / dyE[w]) -
                                      (dxT[c] * dxT[c]) * (vN[c] / dxN[c] - vN[s] / dxN[s]);
                    double uareaavgr = 1.0 / (uarea[c] + uarea[s] + uarea[sw] + uarea[w]);
                    double shearTsqr = (shU[c] * shU[c] * uarea[c] + shU[s] * shU[s] * uarea[s] +
                                        shU[sw] * shU[sw] * uarea[sw] + shU[w] * shU[w] * uarea[w]) * uareaavgr;
                    double shearT = (shU[c] * uarea[c] + shU[s] * uarea[s] + shU[sw] * uarea[sw] + shU[w] * uarea[w]) *
                                    uareaavgr;
                    double DeltaT = sqrt(divT * divT + p->e_factor * (tensionT * tensionT + shearTsqr));
                    double zetax2, etax2, rep_prs;
                    visc_replpress(p, strength[c], Dmin[c], DeltaT, &zetax2, &etax2, &rep_prs);
                    f[14][b * nb + c] = zetax2;
                    f[15][b * nb + c] = etax2;
                    f[6][b * nb + c] = (f[6][b * nb + c] * relax + p->arlx1i * (zetax2 * divT - rep_prs)) * p->denom1;
                    f[7][b * nb + c] = (f[7][b * nb + c] * relax + p->arlx1i * etax2 * tensionT) * p->denom1;
                    f[8][b * nb + c] = (f[8][b * nb + c] * relax + p->arlx1i * p5 * etax2 * shearT) * p->denom1;
                }
        }
        for (int k = 14; k <= 15; ++k) evp_oracle_halo_update(d, f[k], 0, 0, 0, 0.0);   /* (:988-990) */
        for (int k = 6; k <= 7; ++k) evp_oracle_halo_update(d, f[k], 0, 0, 0, 0.0);

        /* ---- viscosity at the corners (:992-996) and stressC_U (:1898-1972) ---- */
        if (avg_strength) avg_T2U_S(d, in[0], g[9], g[17], strengthU);
        else avg_T2U_S(d, f[15], g[9], g[17], f[16]);
        for (int b = 0; b < d->nblocks; ++b)
            for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
                for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                    const size_t c = b * nb + IX(i, j);
                    if (!iceUmask[c]) continue;
                    double etax2U = f[16][c];
                    if (avg_strength) {
                        double z, r;
                        visc_replpress(p, strengthU[c], p->deltaminEVP * g[8][c], f[18][c], &z, &etax2U, &r);
                    }
                    f[9][c] = (f[9][c] * relax + p->arlx1i * p5 * etax2U * f[17][c]) * p->denom1;
                }
        evp_oracle_halo_update(d, f[9], 1, 0, 0, 0.0);       /* stress12U (:1011-1013) */

        /* ---- div_stress_Ex, div_stress_Ny (:2195-2239, :2371-2416), stepu_C, stepv_C (ice_dyn_shared.F90:1090-1290) ---- */
        for (int b = 0; b < d->nblocks; ++b) {
            const double *sp = f[6] + b * nb, *sm = f[7] + b * nb, *s12 = f[9] + b * nb;
            const double *dxT = g[0] + b * nb, *dyT = g[1] + b * nb, *dxU = g[2] + b * nb, *dyU = g[3] + b * nb;
            const double *dxE = g[4] + b * nb, *dyE = g[5] + b * nb, *dxN = g[6] + b * nb, *dyN = g[7] + b * nb;
            for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
                for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                    const size_t c = IX(i, j), e = IX(i + 1, j), n = IX(i, j + 1), s = IX(i, j - 1), w = IX(i - 1, j);
                    const size_t o = b * nb + c;
                    if (iceEmask[o])
                        f[10][o] = in[11][o] * g[12][o] *
                                   (p5 * dyE[c] * (sp[e] - sp[c]) +
                                    (p5 / dyE[c]) * ((dyT[e] * dyT[e]) * sm[e] - (dyT[c] * dyT[c]) * sm[c]) +
                                    (1.0 / dxE[c]) * ((dxU[c] * dxU[c]) * s12[c] - (dxU[s] * dxU[s]) * s12[s]));
                    if (iceNmask[o])
                        f[11][o] = in[22][o] * g[13][o] *
                                   (p5 * dxN[c] * (sp[n] - sp[c]) -
                                    (p5 / dxN[c]) * ((dxT[n] * dxT[n]) * sm[n] - (dxT[c] * dxT[c]) * sm[c]) +
                                    (1.0 / dyN[c]) * ((dyU[c] * dyU[c]) * s12[c] - (dyU[w] * dyU[w]) * s12[w]));
                }
        }
        for (int b = 0; b < d->nblocks; ++b)
            for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
                for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                    const size_t o = b * nb + IX(i, j);
                    if (iceEmask[o]) {       /* stepu_C: u at E, with the v interpolated to E last subcycle */
                        double uold = f[0][o], vold = f[1][o];
                        double vrel = in[2][o] * p->rhow * in[1][o] *
                                      sqrt((in[3][o] - uold) * (in[3][o] - uold) + (in[4][o] - vold) * (in[4][o] - vold));
                        double taux = vrel * in[5][o];
                        double ccc = sqrt(uold * uold + vold * vold) + p->u0;
                        double Cb = in[10][o] / ccc;
                        double cca = (p->brlx + p->revp) * in[7][o] + vrel * p->cosw + Cb;
                        double ccb = in[8][o] + copysign(1.0, in[8][o]) * vrel * p->sinw;
                        double cc1 = f[10][o] + in[6][o] + taux + in[7][o] * (p->brlx * uold + p->revp * in[9][o]);
                        f[0][o] = (ccb * vold + cc1) / cca;
                        f[12][o] = -f[0][o] * Cb;
                    }
                    if (iceNmask[o]) {       /* stepv_C: v at N, with the u interpolated to N last subcycle */
                        double uold = f[2][o], vold = f[3][o];
                        double vrel = in[13][o] * p->rhow * in[12][o] *
                                      sqrt((in[14][o] - uold) * (in[14][o] - uold) + (in[15][o] - vold) * (in[15][o] - vold));
                        double tauy = vrel * in[16][o];
                        double ccc = sqrt(uold * uold + vold * vold) + p->u0;
                        double Cb = in[21][o] / ccc;
                        double cca = (p->brlx + p->revp) * in[18][o] + vrel * p->cosw + Cb;
                        double ccb = in[19][o] + copysign(1.0, in[19][o]) * vrel * p->sinw;
                        double cc2 = f[11][o] + in[17][o] + tauy + in[18][o] * (p->brlx * vold + p->revp * in[20][o]);
                        f[3][o] = (-ccb * uold + cc2) / cca;
                        f[13][o] = -f[3][o] * Cb;
                    }
                }
        evp_oracle_halo_update(d, f[0], 2, 1, 0, 0.0);       /* uvelE: E face, vector (:1063-1065) */
        evp_oracle_halo_update(d, f[3], 3, 1, 0, 0.0);       /* vvelN: N face, vector (:1066-1068) */

        /* ---- the other component at each face (:1070-1082), corner velocities (:1084-1094) ---- */
        avg_A(d, 0, f[0], g[10], g[15], f[2]);               /* uvelN = E2N(uvelE) * npm */
        avg_A(d, 1, f[3], g[11], g[14], f[1]);               /* vvelE = N2E(vvelN) * epm */
        evp_oracle_halo_update(d, f[2], 3, 1, 0, 0.0);
        evp_oracle_halo_update(d, f[1], 2, 1, 0, 0.0);
        avg_A(d, 2, f[0], g[10], g[16], f[4]);               /* uvel = E2U(uvelE) * uvm */
        avg_A(d, 3, f[3], g[11], g[16], f[5]);               /* vvel = N2U(vvelN) * uvm */
        evp_oracle_halo_update(d, f[4], 1, 1, 0, 0.0);
        evp_oracle_halo_update(d, f[5], 1, 1, 0, 0.0);
    }
    free(divergU);
    free(tensionU);
    free(strengthU);
}


/* =====================================================================
 * deformationsC_T, dynamics/ice_dyn_shared.F90:1968-2074 (strain_rates_Tdtsd :2171-2243, strain_rates_Tdt :2251-2311);
 * evp() calls it right after the C-grid loop (ice_dyn_evp.F90:1106-1119).  On the T-cells of dyn_prep2's list
 * (ilo..ihi+1 x jlo..jhi+1 where iceTmask): divu, shear, vort, rdg_conv, rdg_shear; every other cell keeps its value.
 * ===================================================================== */
void evp_oracle_deformations_c_t(const evp_oracle_domain *d, double e_factor, const double *uvelE, const double *vvelE,
                                 const double *uvelN, const double *vvelN, const double *dxN, const double *dyE,
                                 const double *dxT, const double *dyT, const double *tarear, const double *uarea,
                                 const double *shearU, const int32_t *iceTmask, double *vort, double *shear, double *divu,
                                 double *rdg_conv, double *rdg_shear)
{
    const int nx = d->nx_block;
    const size_t nb = (size_t)nx * d->ny_block;
    for (int b = 0; b < d->nblocks; ++b)
        for (int j = d->jlo[b]; j <= d->jhi[b] + 1; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b] + 1; ++i) {
                const size_t c = b * nb + IX(i, j), w = c - 1, s = c - nx, sw = s - 1;
                if (!iceTmask[c]) continue;
                /* strain_rates_Tdt: divergence and tension (x area) */
                const double divT = dyE[c] * uvelE[c] - dyE[w] * uvelE[w] + dxN[c] * vvelN[c] - dxN[s] * vvelN[s];
                const double tensionT = (dyT[c] * dyT[c]) * (uvelE[c] / dyE[c] - uvelE[w] / dyE[w]) -
                                        (dxT[c] * dxT[c]) * (vvelN[c] / dxN[c] - vvelN[s] / dxN[s]);
                /* strain_rates_Tdtsd: shear at the T point from the N / E velocities */
                const double shearT = (dxT[c] * dxT[c]) * (uvelN[c] / dxN[c] - uvelN[s] / dxN[s]) +
                                      (dyT[c] * dyT[c]) * (vvelE[c] / dyE[c] - vvelE[w] / dyE[w]);
                const double shearTsqr = (shearU[c] * shearU[c] * uarea[c] + shearU[s] * shearU[s] * uarea[s] +
                                          shearU[sw] * shearU[sw] * uarea[sw] + shearU[w] * shearU[w] * uarea[w]) /
                                         (uarea[c] + uarea[s] + uarea[sw] + uarea[w]);
                const double DeltaT = sqrt(divT * divT + e_factor * (tensionT * tensionT + shearTsqr));
                divu[c] = divT * tarear[c];
                const double tmp = DeltaT * tarear[c];
                rdg_conv[c] = -fmin(divu[c], 0.0);
                rdg_shear[c] = 0.5 * (tmp - fabs(divu[c]));
                shear[c] = tarear[c] * sqrt(tensionT * tensionT + shearT * shearT);
                vort[c] = tarear[c] * ((dyE[c] * vvelE[c] - dyE[w] * vvelE[w]) - (dxN[c] * uvelN[c] - dxN[s] * uvelN[s]));
            }
}

/* =====================================================================
 * Preparation phase of evp() for grid_ice = 'C' (dynamics/ice_dyn_evp.F90:383-735 and 770-840, calc_strair branch, ocean
 * and atmosphere forcing on the T grid): dyn_prep1 (:383-416), the T-grid halo updates (:418-428, 471-474), the state-masked
 * averages T -> U / E / N (:430-453; grid_average_X2YS 'NE' / 'E' / 'N', ice_grid.F90:4190-4209, 4290-4306, 4332-4348),
 * the flux averages of the wind stress T -> N / E (:485-488; grid_average_X2YF, ice_grid.F90:4728-4782), dyn_prep2 at
 * U, N and E points (:563-674; ice_dyn_shared.F90:697-838, rheofactX :805-811), the zeroing of the T / U stresses off the
 * ice (:676-691), then the velocity exchanges and face -> face / face -> corner averages (:703-731).  Not here: ice
 * strength (Icepack) and its halo update, the seabed stress factors (evp_oracle_seabed_lkd_c below).
 *
 * g: the 23 static arrays of evp_oracle_cgrid_subcycle (uarea 8, tarea 9, earea 10, narea 11, epm 14, npm 15, uvm 16, hm 17).
 * masks4: tmask, umaskCD, emask, nmask.  fcor3: fcor_blk, fcorE_blk, fcorN_blk.  T: the 11 T-grid fields of evp_oracle_prep.
 * f: the first 14 arrays of the loop's state table (inout): uvelE vvelE uvelN vvelN uvel vvel stresspT stressmT stress12T
 * stress12U strintxE strintyN taubxE taubyN.  in: the loop's 23 per-call inputs (out; [0] strength is not touched).
 * iceUmask, iceEmask, iceNmask: in = the previous call's, out = new; iceTmask: out.
 * ===================================================================== */
static void avg_T2X_S(const evp_oracle_domain *d, int dir, const double *w1, const double *wt, const double *m, double *w2)
{
    const int nx = d->nx_block;
    const size_t nb = (size_t)nx * d->ny_block;
    memset(w2, 0, sizeof(double) * nb * d->nblocks);
    for (int b = 0; b < d->nblocks; ++b)
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const size_t c = b * nb + IX(i, j), q = dir == 2 ? c + 1 : c + nx;       /* 2: 'E' (i+1), 3: 'N' (j+1) */
                const double wtmp = (m[c] * wt[c] + m[q] * wt[q]);
                if (wtmp != 0.0) w2[c] = (m[c] * w1[c] * wt[c] + m[q] * w1[q] * wt[q]) / wtmp;
            }
}
static void avg_T2X_F(const evp_oracle_domain *d, int dir, const double *w1, const double *wt, const double *wt2, double *w2)
{
    const int nx = d->nx_block;
    const size_t nb = (size_t)nx * d->ny_block;
    memset(w2, 0, sizeof(double) * nb * d->nblocks);
    for (int b = 0; b < d->nblocks; ++b)
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const size_t c = b * nb + IX(i, j), q = dir == 2 ? c + 1 : c + nx;
                w2[c] = p5 * (w1[c] * wt[c] + w1[q] * wt[q]) / wt2[c];
            }
}

void evp_oracle_cgrid_prep(const evp_oracle_domain *d, const evp_oracle_prep_params *p, const int32_t *const *masks4,
                           const double *const *fcor3, const double *const *g, const double *const *T, double *const *f,
                           double *const *in, int32_t *iceTmask, int32_t *iceUmask, int32_t *iceEmask, int32_t *iceNmask)
{
    const int nx = d->nx_block, ny = d->ny_block;
    const size_t nb = (size_t)nx * ny, n = nb * d->nblocks;
    const int32_t *tmask = masks4[0];
    const double *uarea = g[8], *tarea = g[9], *earea = g[10], *narea = g[11], *epm = g[14], *npm = g[15], *uvm = g[16],
                 *hm = g[17];
    const double rheo_area_min = 1e-3;                     /* ice_dyn_shared.F90:67 */
    double *t[PT_COUNT];
    for (int k = 0; k < PT_COUNT; ++k) {
        t[k] = (double *)malloc(sizeof(double) * n);
        memcpy(t[k], T[k], sizeof(double) * n);
    }
    double *tmass = (double *)malloc(sizeof(double) * n), *maskd = (double *)malloc(sizeof(double) * n);
    unsigned char *tmphm = (unsigned char *)malloc(n);
    /* dyn_prep1 */
    for (int b = 0; b < d->nblocks; ++b) {
        const size_t o = b * nb;
        for (int j = 1; j <= ny; ++j)
            for (int i = 1; i <= nx; ++i) {
                const size_t c = o + IX(i, j);
                tmass[c] = tmask[c] ? (p->rhoi * t[PT_VICE][c] + p->rhos * t[PT_VSNO][c]) : 0.0;
                tmphm[c] = tmask[c] && (t[PT_AICE][c] > p->dyn_area_min) && (tmass[c] > p->dyn_mass_min);
                iceTmask[c] = 0;
            }
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                int any = 0;
                for (int dj = -1; dj <= 1; ++dj)
                    for (int di = -1; di <= 1; ++di) any |= tmphm[o + IX(i + di, j + dj)];
                iceTmask[o + IX(i, j)] = any && tmask[o + IX(i, j)];
            }
    }
    for (size_t c = 0; c < n; ++c) maskd[c] = (double)iceTmask[c];
    evp_oracle_halo_update(d, maskd, 0, 0, 0, 0.0);
    for (size_t c = 0; c < n; ++c) iceTmask[c] = maskd[c] != 0.0;
    evp_oracle_halo_update(d, tmass, 0, 0, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_AICE_INIT], 0, 0, 0, 0.0);
    evp_oracle_halo_update(d, t[PT_CDN_OCN], 0, 0, 0, 0.0);
    for (int k = PT_UOCN; k <= PT_STRAIRY; ++k) evp_oracle_halo_update(d, t[k], 0, 1, 0, 0.0);

    /* the averages: per location X in {U, E, N}: Xmass, aiX, cdn_ocnX, uocnX, vocnX, ss_tltxX, ss_tltyX, strairxX, strairyX */
    double *A[3][9];
    for (int L = 0; L < 3; ++L)
        for (int k = 0; k < 9; ++k) A[L][k] = (double *)malloc(sizeof(double) * n);
    const double *src[7] = {tmass, t[PT_AICE_INIT], t[PT_CDN_OCN], t[PT_UOCN], t[PT_VOCN], t[PT_SS_TLTX], t[PT_SS_TLTY]};
    for (int k = 0; k < 7; ++k) {
        avg_T2U_S(d, src[k], tarea, hm, A[0][k]);
        avg_T2X_S(d, 2, src[k], tarea, hm, A[1][k]);
        avg_T2X_S(d, 3, src[k], tarea, hm, A[2][k]);
    }
    for (int k = 0; k < 2; ++k) {
        avg_T2U_F(d, t[PT_STRAIRX + k], tarea, uarea, A[0][7 + k]);
        avg_T2X_F(d, 2, t[PT_STRAIRX + k], tarea, earea, A[1][7 + k]);
        avg_T2X_F(d, 3, t[PT_STRAIRX + k], tarea, narea, A[2][7 + k]);
    }
    /* dyn_prep2 at U (only its mask and velocities matter to the C-grid loop), N, E */
    double *scratch = (double *)malloc(sizeof(double) * n);       /* products of the U pass nothing reads afterwards */
    for (int L = 0; L < 3; ++L) {
        const int32_t *Xmask = masks4[1 + L];
        int32_t *iceX = L == 0 ? iceUmask : (L == 1 ? iceEmask : iceNmask);
        const double *Xmass = A[L][0], *aiX = A[L][1], *uocnX = A[L][3], *vocnX = A[L][4], *fcor = fcor3[L];
        double *uX = L == 0 ? f[4] : (L == 1 ? f[0] : f[2]), *vX = L == 0 ? f[5] : (L == 1 ? f[1] : f[3]);
        /* E: 1 cdn 2 ai 3 uocn 4 vocn 5 waterx 6 forcex 7 massdti 8 fm 9 u_init 10 Tb 11 rheofact; N: + 11 */
        const int o0 = L == 1 ? 1 : 12;
        double *cdn = L ? in[o0] : scratch, *ai = L ? in[o0 + 1] : scratch, *uo = L ? in[o0 + 2] : scratch,
               *vo = L ? in[o0 + 3] : scratch, *water = L ? in[o0 + 4] : scratch, *force = L ? in[o0 + 5] : scratch,
               *massdti = L ? in[o0 + 6] : scratch, *fm = L ? in[o0 + 7] : scratch, *init = L ? in[o0 + 8] : scratch,
               *Tb = L ? in[o0 + 9] : scratch, *rheo = L ? in[o0 + 10] : scratch;
        double *strint = L == 1 ? f[10] : (L == 2 ? f[11] : scratch), *taub = L == 1 ? f[12] : (L == 2 ? f[13] : scratch);
        if (L) {
            memcpy(cdn, A[L][2], sizeof(double) * n);
            memcpy(ai, aiX, sizeof(double) * n);
            memcpy(uo, uocnX, sizeof(double) * n);
            memcpy(vo, vocnX, sizeof(double) * n);
        }
        for (int b = 0; b < d->nblocks; ++b) {
            const size_t o = b * nb;
            for (int j = 1; j <= ny; ++j)
                for (int i = 1; i <= nx; ++i) {
                    const size_t c = o + IX(i, j);
                    water[c] = 0.0; force[c] = 0.0; massdti[c] = 0.0; Tb[c] = 0.0; taub[c] = 0.0;   /* :701-712 */
                }
            for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
                for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                    const size_t c = o + IX(i, j);
                    const int old = iceX[c];
                    iceX[c] = Xmask[c] && (aiX[c] > p->dyn_area_min) && (Xmass[c] > p->dyn_mass_min);
                    if (iceX[c]) {
                        if (!old) { uX[c] = uocnX[c]; vX[c] = vocnX[c]; }
                    } else {
                        uX[c] = vX[c] = 0.0;
                        strint[c] = 0.0;            /* (strintyE / strintxN and the ocean stresses: not the loop's) */
                    }
                    init[c] = L == 2 ? vX[c] : uX[c];
                    if (!iceX[c]) continue;
                    rheo[c] = aiX[c] > rheo_area_min ? 1.0 : 0.0;
                    massdti[c] = Xmass[c] / p->dt;
                    const double fmc = fcor[c] * Xmass[c];
                    fm[c] = fmc;
                    const double sgn = copysign(1.0, fmc);
                    const double wx = uocnX[c] * p->cosw - vocnX[c] * p->sinw * sgn;
                    const double wy = vocnX[c] * p->cosw + uocnX[c] * p->sinw * sgn;
                    double tx, ty;
                    if (p->ssh_coupled) {
                        tx = -p->gravit * Xmass[c] * A[L][5][c];
                        ty = -p->gravit * Xmass[c] * A[L][6][c];
                    } else {
                        tx = -fmc * vocnX[c];
                        ty = fmc * uocnX[c];
                    }
                    water[c] = L == 2 ? wy : wx;
                    force[c] = L == 2 ? (A[L][8][c] + ty) : (A[L][7][c] + tx);
                }
        }
    }
    /* :676-691: T stresses off the T mask, U stresses off the U mask, on every cell (iceUmask is never set on ghosts) */
    for (size_t c = 0; c < n; ++c) {
        if (!iceUmask[c]) f[9][c] = 0.0;
        if (!iceTmask[c]) f[6][c] = f[7][c] = f[8][c] = 0.0;
    }
    /* :703-731 */
    evp_oracle_halo_update(d, f[0], 2, 1, 0, 0.0);
    evp_oracle_halo_update(d, f[3], 3, 1, 0, 0.0);
    avg_A(d, 0, f[0], earea, npm, f[2]);
    avg_A(d, 1, f[3], narea, epm, f[1]);
    evp_oracle_halo_update(d, f[2], 3, 1, 0, 0.0);
    evp_oracle_halo_update(d, f[1], 2, 1, 0, 0.0);
    avg_A(d, 2, f[0], earea, uvm, f[4]);
    avg_A(d, 3, f[3], narea, uvm, f[5]);
    evp_oracle_halo_update(d, f[4], 1, 1, 0, 0.0);     /* :735-738 */
    evp_oracle_halo_update(d, f[5], 1, 1, 0, 0.0);
    for (int k = 0; k < PT_COUNT; ++k) free(t[k]);
    for (int L = 0; L < 3; ++L)
        for (int k = 0; k < 9; ++k) free(A[L][k]);
    free(tmass); free(maskd); free(tmphm); free(scratch);
}

/* seabed_stress_factor_LKD at E or N points (grid_location = 'E' / 'N', ice_dyn_shared.F90:1386-1460 with
 * grid_neighbor_min / _max over the two T-cells sharing the face, ice_grid.F90:4975-4978, 5006-5009); call site
 * ice_dyn_evp.F90:803-815.  dir 2: E (i, i+1), 3: N (j, j+1). */
void evp_oracle_seabed_lkd_c(const evp_oracle_domain *d, int dir, double k1, double k2, double alphab, double threshold_hw,
                             const double *aice, const double *vice, const double *hwater, const int32_t *iceXmask,
                             double *TbX)
{
    const int nx = d->nx_block;
    const size_t plane = (size_t)nx * d->ny_block;
    for (int b = 0; b < d->nblocks; ++b) {
        for (size_t c = b * plane; c < (b + 1) * plane; ++c) TbX[c] = 0.0;
        for (int j = d->jlo[b]; j <= d->jhi[b]; ++j)
            for (int i = d->ilo[b]; i <= d->ihi[b]; ++i) {
                const size_t c = b * plane + (size_t)(j - 1) * nx + (i - 1), q = dir == 2 ? c + 1 : c + nx;
                if (!iceXmask[c]) continue;
                const double hwu = fmin(hwater[c], hwater[q]);
                const double docalc_tbu = hwu < threshold_hw ? 1.0 : 0.0;
                const double au = fmax(aice[c], aice[q]);
                const double hu = fmax(vice[c], vice[q]);
                const double hcu = au * hwu / k1;
                TbX[c] = docalc_tbu * k2 * fmax(0.0, (hu - hcu)) * exp(-alphab * (1.0 - au));
            }
    }
}
