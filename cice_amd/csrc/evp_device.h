// Device-side argument blocks and launcher prototypes shared by
// evp_kernels.hip (kernels) and evp_api.cpp (C ABI / state management).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    // per-call inputs (dyn_evp1d_run argument list, ice_dyn_evp1d.F90:121-153)
    const double *strength, *Cw, *aiX, *uocn, *vocn, *waterx, *watery, *forcex, *forcey;
    const double *umassdti, *fm, *TbU, *uvel_init, *vvel_init;
    // diagnostics of the last subcycle
    double *strintx, *strinty, *taubx, *tauby;
};

void evp_launch_subcycle(const EvpArgs &A, int max_ni, int max_nj, int nblocks, int tyb,
                         bool strict, int cap, hipStream_t st);
void evp_launch_halo_local(double *u, double *v, const int *dst, const int *src,
                           const signed char *sign, int n, hipStream_t st);
void evp_launch_halo_pack(const double *u, const double *v, const int *src, double *buf, int n,
                          hipStream_t st);
void evp_launch_halo_unpack(double *u, double *v, const int *dst, const signed char *sign,
                            const double *buf, int n, hipStream_t st);
