// Device-side arithmetic shared by evp_kernels.hip (streaming tiles) and
// evp_resident2.hip (on-chip resident subcycle): constants, the two builds of the
// per-cell formulas (strict / fused) and a traits struct selecting between them.
#pragma once
#include <hip/hip_runtime.h>
#include "evp_device.h"

namespace {

// shared/ice_constants.F90:79-85
__device__ constexpr double p027 = 1.0 / 36.0;
__device__ constexpr double p055 = 1.0 / 18.0;
__device__ constexpr double p111 = 1.0 / 9.0;
__device__ constexpr double p166 = 1.0 / 6.0;
__device__ constexpr double p222 = 2.0 / 9.0;
__device__ constexpr double p25 = 0.25;
__device__ constexpr double p333 = 1.0 / 3.0;
__device__ constexpr double p5 = 0.5;
__device__ constexpr double c1p5 = 1.5;

}  // namespace

#pragma clang fp contract(off)
namespace evp_strict {
#include "evp_cell.inc"
}
#pragma clang fp contract(on)
namespace evp_fused {
#include "evp_cell.inc"
}

namespace {

template <bool STRICT> struct Math;
template <> struct Math<true> {
    using SI = evp_strict::StressIn;
    using UI = evp_strict::StepuIn;
    using UO = evp_strict::StepuOut;
    // MODE: 1 capping == 1, 0 capping == 0, -1 general capping; 3 = capping == 1 AND the reference's default
    // scalars (revp == 0, Ktens == 0, cosw == 1, sinw == 0): products with exactly 1.0 dropped (evp_cell.inc)
    template <int MODE>
    static __device__ __forceinline__ void stress(const EvpScalars &p, const SI &a, double (&s)[12], double (&str)[8])
    {
        evp_strict::stress_cell<(MODE == 3 ? 1 : MODE), MODE == 3>(p, a, s, str);
    }
    template <int MODE = -1, bool TBU = true>
    static __device__ __forceinline__ void stepu(const EvpScalars &p, const UI &a, UO &o) { evp_strict::stepu_cell<MODE == 3, TBU>(p, a, o); }
    static __device__ __forceinline__ void metrics(double hte, double hte_im, double htn, double htn_jm, double dmin, SI &a)
    {
        evp_strict::metrics_cell(hte, hte_im, htn, htn_jm, dmin, a);
    }
    // stress_cell one corner per lane (evp_resident2.hip, COOP)
    using CI = evp_strict::CornerIn;
    template <int MODE>
    static __device__ __forceinline__ void corner(const EvpScalars &p, const CI &c, double &sp, double &sm, double &s12)
    {
        evp_strict::stress_corner<(MODE == 3 ? 1 : MODE), MODE == 3>(p, c, sp, sm, s12);
    }
    static __device__ __forceinline__ void corner_operands(int q, double dxT, double dyT, double cxp, double cyp, double cxm, double cym,
                                                           CI &c, double &KX, double &K12X, double &KYP, double &K12Y)
    {
        evp_strict::corner_operands(q, dxT, dyT, cxp, cyp, cxm, cym, c, KX, K12X, KYP, K12Y);
    }
    static __device__ __forceinline__ void partials(double P0, double P1, double P2, double P3, double M0, double M1, double M2, double M3,
                                                    double T0, double T1, double T2, double T3, double KX, double K12X, double KYP,
                                                    double K12Y, double dxhy, double dyhx, double &X, double &Y)
    {
        evp_strict::stress_corner_partials(P0, P1, P2, P3, M0, M1, M2, M3, T0, T1, T2, T3, KX, K12X, KYP, K12Y, dxhy, dyhx, X, Y);
    }
};
template <> struct Math<false> {
    using SI = evp_fused::StressIn;
    using UI = evp_fused::StepuIn;
    using UO = evp_fused::StepuOut;
    // MODE: 1 capping == 1, 0 capping == 0, -1 general capping; 3 = capping == 1 AND the reference's default
    // scalars (revp == 0, Ktens == 0, cosw == 1, sinw == 0): products with exactly 1.0 dropped (evp_cell.inc)
    template <int MODE>
    static __device__ __forceinline__ void stress(const EvpScalars &p, const SI &a, double (&s)[12], double (&str)[8])
    {
        evp_fused::stress_cell<(MODE == 3 ? 1 : MODE), MODE == 3>(p, a, s, str);
    }
    template <int MODE = -1, bool TBU = true>
    static __device__ __forceinline__ void stepu(const EvpScalars &p, const UI &a, UO &o) { evp_fused::stepu_cell<MODE == 3, TBU>(p, a, o); }
    static __device__ __forceinline__ void metrics(double hte, double hte_im, double htn, double htn_jm, double dmin, SI &a)
    {
        evp_fused::metrics_cell(hte, hte_im, htn, htn_jm, dmin, a);
    }
    // stress_cell one corner per lane (evp_resident2.hip, COOP)
    using CI = evp_fused::CornerIn;
    template <int MODE>
    static __device__ __forceinline__ void corner(const EvpScalars &p, const CI &c, double &sp, double &sm, double &s12)
    {
        evp_fused::stress_corner<(MODE == 3 ? 1 : MODE), MODE == 3>(p, c, sp, sm, s12);
    }
    static __device__ __forceinline__ void corner_operands(int q, double dxT, double dyT, double cxp, double cyp, double cxm, double cym,
                                                           CI &c, double &KX, double &K12X, double &KYP, double &K12Y)
    {
        evp_fused::corner_operands(q, dxT, dyT, cxp, cyp, cxm, cym, c, KX, K12X, KYP, K12Y);
    }
    static __device__ __forceinline__ void partials(double P0, double P1, double P2, double P3, double M0, double M1, double M2, double M3,
                                                    double T0, double T1, double T2, double T3, double KX, double K12X, double KYP,
                                                    double K12Y, double dxhy, double dyhx, double &X, double &Y)
    {
        evp_fused::stress_corner_partials(P0, P1, P2, P3, M0, M1, M2, M3, T0, T1, T2, T3, KX, K12X, KYP, K12Y, dxhy, dyhx, X, Y);
    }
};

}  // namespace
