"""TEST INFRASTRUCTURE -- ctypes front end of oracle/libevp_oracle.so (the CPU
restatement of the reference EVP subcycle).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
(cice_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libevp_oracle.so"

BND = {"closed": 0, "open": 1, "cyclic": 2, "tripole": 3, "tripoleT": 4}

# order of the dynamic-field pointer table of evp_oracle_subcycle
DYN_FIELDS = [
    "stressp_1", "stressp_2", "stressp_3", "stressp_4",
    "stressm_1", "stressm_2", "stressm_3", "stressm_4",
    "stress12_1", "stress12_2", "stress12_3", "stress12_4",
    "strength", "cdn_ocnU", "aiU", "uocnU", "vocnU", "waterxU", "wateryU",
    "forcexU", "forceyU", "umassdti", "fmU", "strintxU", "strintyU", "TbU",
    "taubxU", "taubyU", "uvel", "vvel", "uvel_init", "vvel_init",
]
STATIC_FIELDS = ["dxT", "dyT", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "DminTarea", "uarear"]
OUT_FIELDS = DYN_FIELDS[:12] + ["strintxU", "strintyU", "taubxU", "taubyU", "uvel", "vvel"]


class Domain(C.Structure):
    _fields_ = [("nx_block", C.c_int), ("ny_block", C.c_int), ("nblocks", C.c_int),
                ("nghost", C.c_int), ("nx_global", C.c_int), ("ny_global", C.c_int),
                ("ew_type", C.c_int), ("ns_type", C.c_int),
                ("ilo", C.POINTER(C.c_int)), ("ihi", C.POINTER(C.c_int)),
                ("jlo", C.POINTER(C.c_int)), ("jhi", C.POINTER(C.c_int)),
                ("iglob0", C.POINTER(C.c_int)), ("jglob0", C.POINTER(C.c_int))]


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping", "Ktens",
                 "deltaminEVP", "u0", "cosw", "sinw", "rhow")]


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < (HERE / "evp_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B", "libevp_oracle.so"], check=True,
                       capture_output=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.evp_oracle_subcycle.restype = None
        _lib.evp_oracle_metrics.restype = None
        _lib.evp_oracle_halo_update.restype = None
        _lib.evp_oracle_set_parameters.restype = None
    return _lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleDomain:
    """Block geometry as the reference's `type(block)` holds it
    (ice_blocks.F90:21-41): per block ilo,ihi,jlo,jhi (1-based) and the global
    index of the first interior cell."""

    def __init__(self, nx_block, ny_block, nblocks, nx_global, ny_global, ew, ns,
                 ilo, ihi, jlo, jhi, iglob0, jglob0, nghost=1):
        self.arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (ilo, ihi, jlo, jhi, iglob0, jglob0)]
        self.c = Domain(nx_block, ny_block, nblocks, nghost, nx_global, ny_global,
                        BND[ew], BND[ns], *[_ip(a) for a in self.arrs])
        self.shape = (nblocks, ny_block, nx_block)

    @classmethod
    def from_dump(cls, d, ew, ns):
        nxb, nyb, nblk, ngh, nxg, nyg = [int(v) for v in d["dims"][:6]]
        bi = np.asarray(d["blkinfo"]).reshape(nblk, 8)
        return cls(nxb, nyb, nblk, nxg, nyg, ew, ns, bi[:, 0], bi[:, 1], bi[:, 2], bi[:, 3],
                   bi[:, 6], bi[:, 7], nghost=ngh)


def make_params(**kw) -> Params:
    p = Params()
    for k, v in kw.items():
        setattr(p, k, float(v))
    return p


def params_from_scalars(s) -> Params:
    """scalars record of the reference harness dump (evp_ref_harness.F90)."""
    return make_params(arlx1i=s[0], denom1=s[1], brlx=s[2], revp=s[3], e_factor=s[4], epp2i=s[5],
                       capping=s[6], Ktens=s[7], deltaminEVP=s[8], u0=s[9], cosw=s[10], sinw=s[11],
                       rhow=s[12])


def set_parameters(ndte, dt, revised_evp=False, elasticDamp=0.36, arlx=300.0, brlx=300.0,
                   e_yieldcurve=2.0, e_plasticpot=2.0):
    out = np.zeros(9)
    lib().evp_oracle_set_parameters(C.c_int(ndte), C.c_double(dt), C.c_int(int(revised_evp)),
                                    C.c_double(elasticDamp), C.c_double(arlx), C.c_double(brlx),
                                    C.c_double(e_yieldcurve), C.c_double(e_plasticpot), _dp(out))
    return dict(zip(["arlx", "arlx1i", "brlx", "denom1", "revp", "epp2i", "e_factor", "dtei", "ecci"], out))


def metrics(dom: OracleDomain, deltaminEVP, HTE, HTN, tarea):
    names = ["cxp", "cyp", "cxm", "cym", "dxhy", "dyhx", "DminTarea"]
    out = {n: np.zeros(dom.shape) for n in names}
    HTE, HTN, tarea = [np.ascontiguousarray(a, dtype=np.float64) for a in (HTE, HTN, tarea)]
    lib().evp_oracle_metrics(C.byref(dom.c), C.c_double(deltaminEVP), _dp(HTE), _dp(HTN), _dp(tarea),
                             *[_dp(out[n]) for n in names])
    return out


def halo_update(dom: OracleDomain, a, field_loc="NEcorner", field_type="vector", fill=None):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    loc = {"center": 0, "NEcorner": 1, "Eface": 2, "Nface": 3}[field_loc]
    lib().evp_oracle_halo_update(C.byref(dom.c), _dp(a), C.c_int(loc),
                                 C.c_int(1 if field_type == "vector" else 0),
                                 C.c_int(0 if fill is None else 1), C.c_double(0.0 if fill is None else fill))
    return a


def tripole_stress_sym(dom: OracleDomain, out: dict) -> dict:
    """ice_HaloUpdate_stress x12 as evp() applies it after the loop (ice_dyn_evp.F90:1364-1387)."""
    lib().evp_oracle_tripole_stress.restype = None
    pairs = [(1, 3), (3, 1), (2, 4), (4, 2)]
    for fam in ("stressp", "stressm", "stress12"):
        for a, b in pairs:
            lib().evp_oracle_tripole_stress(C.byref(dom.c), _dp(out[f"{fam}_{a}"]), _dp(out[f"{fam}_{b}"]))
    return out


def subcycle(dom: OracleDomain, params: Params, ndte: int, dyn: dict, static: dict,
             iceTmask, iceUmask) -> dict:
    """Run ndte subcycles; `dyn` maps DYN_FIELDS -> arrays (copied, originals untouched).
    Returns the dict of all dynamic fields after the loop."""
    work = {k: np.array(dyn[k], dtype=np.float64, order="C", copy=True) for k in DYN_FIELDS}
    st = {k: np.ascontiguousarray(static[k], dtype=np.float64) for k in STATIC_FIELDS}
    tm = np.ascontiguousarray(iceTmask, dtype=np.int32)
    um = np.ascontiguousarray(iceUmask, dtype=np.int32)
    fptr = (C.POINTER(C.c_double) * len(DYN_FIELDS))(*[_dp(work[k]) for k in DYN_FIELDS])
    gptr = (C.POINTER(C.c_double) * len(STATIC_FIELDS))(*[_dp(st[k]) for k in STATIC_FIELDS])
    lib().evp_oracle_subcycle(C.byref(dom.c), C.byref(params), C.c_int(ndte), fptr, gptr,
                              tm.ctypes.data_as(C.POINTER(C.c_int32)),
                              um.ctypes.data_as(C.POINTER(C.c_int32)))
    return work


def deformations(dom: OracleDomain, params: Params, uvel, vvel, static: dict, geo: dict, iceTmask) -> dict:
    """deformations (ice_dyn_shared.F90:1756-1860).  geo: dxU, dyU, tarear."""
    lib().evp_oracle_deformations.restype = None
    names = ["vort", "shear", "divu", "rdg_conv", "rdg_shear"]
    out = {n: np.zeros(dom.shape) for n in names}
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in
         (uvel, vvel, static["dxT"], static["dyT"], geo["dxU"], geo["dyU"], static["cxp"], static["cyp"],
          static["cxm"], static["cym"], geo["tarear"])]
    tm = np.ascontiguousarray(iceTmask, dtype=np.int32)
    lib().evp_oracle_deformations(C.byref(dom.c), C.byref(params), *[_dp(x) for x in a],
                                  tm.ctypes.data_as(C.POINTER(C.c_int32)), *[_dp(out[n]) for n in names])
    return out


def dyn_finish(dom: OracleDomain, params: Params, dyn: dict, uvel, vvel, iceUmask, strocnx, strocny) -> dict:
    """dyn_finish (ice_dyn_shared.F90:1291-1365); strocnx/y are inout."""
    lib().evp_oracle_dyn_finish.restype = None
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in
         (dyn["cdn_ocnU"], uvel, vvel, dyn["uocnU"], dyn["vocnU"], dyn["aiU"], dyn["fmU"])]
    um = np.ascontiguousarray(iceUmask, dtype=np.int32)
    sx = np.array(strocnx, dtype=np.float64, order="C", copy=True)
    sy = np.array(strocny, dtype=np.float64, order="C", copy=True)
    lib().evp_oracle_dyn_finish(C.byref(dom.c), C.byref(params), *[_dp(x) for x in a],
                                um.ctypes.data_as(C.POINTER(C.c_int32)), _dp(sx), _dp(sy))
    return dict(strocnxU=sx, strocnyU=sy)


def dyn_finish_at(dom: OracleDomain, params: Params, Cw, aiX, uocn, vocn, fm, u, v, icemask, strocnx, strocny):
    """dyn_finish (ice_dyn_shared.F90:1291-1365) with the operands of any staggering: evp() calls the same routine at N and E
    points on the C grid (ice_dyn_evp.F90:1408-1436).  strocnx/y are inout; returns the two arrays."""
    lib().evp_oracle_dyn_finish.restype = None
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (Cw, u, v, uocn, vocn, aiX, fm)]
    m = np.ascontiguousarray(icemask, dtype=np.int32)
    sx = np.array(strocnx, dtype=np.float64, order="C", copy=True)
    sy = np.array(strocny, dtype=np.float64, order="C", copy=True)
    lib().evp_oracle_dyn_finish(C.byref(dom.c), C.byref(params), *[_dp(x) for x in a],
                                m.ctypes.data_as(C.POINTER(C.c_int32)), _dp(sx), _dp(sy))
    return sx, sy


# ---- preparation phase of evp() ----------------------------------------------
PREP_T = ["aice", "vice", "vsno", "aice_init", "cdn_ocn", "uocn", "vocn", "ss_tltx", "ss_tlty",
          "strairxT", "strairyT"]
PREP_U = ["aiU", "cdn_ocnU", "uocnU", "vocnU", "umassdti", "fmU", "waterxU", "wateryU", "forcexU",
          "forceyU", "uvel_init", "vvel_init", "strtltxU", "strtltyU", "strairxU", "strairyU",
          "tmass", "umass"]


class PrepParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("dt", "rhoi", "rhos", "gravit", "dyn_area_min", "dyn_mass_min",
                                          "cosw", "sinw")] + [("ssh_coupled", C.c_int)]


def prep(dom: OracleDomain, pp: PrepParams, static: dict, tfields: dict, state: dict) -> dict:
    """evp()'s preparation phase (ice_dyn_evp.F90:383-840 minus ice strength and seabed stress).
    static: tmask, umask (int), hm, tarea, uarea, fcor_blk.  tfields: PREP_T.  state: the 12
    stresses, uvel, vvel, iceUmask (previous), strintxU/yU, strocnxU/yU (all copied).
    Returns every product: PREP_U, iceTmask, iceUmask and the updated state."""
    lib().evp_oracle_prep.restype = None
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    tm, um = i32(static["tmask"]), i32(static["umask"])
    st = [f64(static[k]) for k in ("hm", "tarea", "uarea", "fcor_blk")]
    T = [f64(tfields[k]) for k in PREP_T]
    out = {k: np.array(state[k], dtype=np.float64, order="C", copy=True)
           for k in DYN_FIELDS[:12] + ["uvel", "vvel", "strintxU", "strintyU", "strocnxU", "strocnyU"]}
    out["iceUmask"] = np.array(state["iceUmask"], dtype=np.int32, order="C", copy=True)
    out["iceTmask"] = np.zeros(dom.shape, dtype=np.int32)
    for k in PREP_U:
        out[k] = np.zeros(dom.shape)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    Tptr = (C.POINTER(C.c_double) * len(T))(*[_dp(a) for a in T])
    Sptr = (C.POINTER(C.c_double) * 12)(*[_dp(out[k]) for k in DYN_FIELDS[:12]])
    Uptr = (C.POINTER(C.c_double) * len(PREP_U))(*[_dp(out[k]) for k in PREP_U])
    lib().evp_oracle_prep(C.byref(dom.c), C.byref(pp), ip(tm), ip(um), *[_dp(a) for a in st], Tptr, Sptr,
                          _dp(out["uvel"]), _dp(out["vvel"]), ip(out["iceUmask"]), _dp(out["strintxU"]),
                          _dp(out["strintyU"]), _dp(out["strocnxU"]), _dp(out["strocnyU"]),
                          ip(out["iceTmask"]), Uptr)
    return out


def seabed_lkd(dom: OracleDomain, k1, k2, alphab, threshold_hw, aice, vice, hwater, iceUmask):
    """seabed_stress_factor_LKD (ice_dyn_shared.F90:1386-1460): TbU on the ice U-cells, 0 elsewhere."""
    lib().evp_oracle_seabed_lkd.restype = None
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a, v, h = f64(aice), f64(vice), f64(hwater)
    um = np.ascontiguousarray(iceUmask, dtype=np.int32)
    out = np.zeros(dom.shape)
    lib().evp_oracle_seabed_lkd(C.byref(dom.c), C.c_double(k1), C.c_double(k2), C.c_double(alphab),
                                C.c_double(threshold_hw), _dp(a), _dp(v), _dp(h),
                                um.ctypes.data_as(C.POINTER(C.c_int32)), _dp(out))
    return out


def seabed_prob(dom: OracleDomain, alphab, rhoi, rhow, gravit, pi, puny, aicen, vicen, hwater, iceTmask, iceUmask):
    """seabed_stress_factor_prob (ice_dyn_shared.F90:1475-1683), B grid: TbU on the ice U-cells, 0 elsewhere.
    aicen / vicen: [nblocks][ncat][ny][nx]."""
    lib().evp_oracle_seabed_prob.restype = None
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a, v, h = f64(aicen), f64(vicen), f64(hwater)
    ncat = a.shape[1]
    i32 = lambda m: np.ascontiguousarray(m, dtype=np.int32)
    tm, um = i32(iceTmask), i32(iceUmask)
    out = np.zeros(dom.shape)
    lib().evp_oracle_seabed_prob(C.byref(dom.c), C.c_int(ncat), C.c_double(alphab), C.c_double(rhoi), C.c_double(rhow),
                                 C.c_double(gravit), C.c_double(pi), C.c_double(puny), _dp(a), _dp(v), _dp(h),
                                 tm.ctypes.data_as(C.POINTER(C.c_int32)), um.ctypes.data_as(C.POINTER(C.c_int32)), _dp(out))
    return out


def seabed_prob_c(dom: OracleDomain, alphab, rhoi, rhow, gravit, pi, puny, aicen, vicen, hwater, iceTmask, iceEmask, iceNmask):
    """seabed_stress_factor_prob for grid_ice = 'C' (ice_dyn_shared.F90:1475-1683, tail :1656-1676): (TbE, TbN)."""
    lib().evp_oracle_seabed_prob_c.restype = None
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a, v, h = f64(aicen), f64(vicen), f64(hwater)
    i32 = lambda m: np.ascontiguousarray(m, dtype=np.int32)
    tm, em, nm = i32(iceTmask), i32(iceEmask), i32(iceNmask)
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))
    tbe, tbn = np.zeros(dom.shape), np.zeros(dom.shape)
    lib().evp_oracle_seabed_prob_c(C.byref(dom.c), C.c_int(a.shape[1]), C.c_double(alphab), C.c_double(rhoi), C.c_double(rhow),
                                   C.c_double(gravit), C.c_double(pi), C.c_double(puny), _dp(a), _dp(v), _dp(h), ip(tm), ip(em),
                                   ip(nm), _dp(tbe), _dp(tbn))
    return tbe, tbn


# ---- C-grid subcycle -----------------------------------------------------------
C_FIELDS = ["uvelE", "vvelE", "uvelN", "vvelN", "uvel", "vvel", "stresspT", "stressmT", "stress12T", "stress12U",
            "strintxE", "strintyN", "taubxE", "taubyN", "zetax2T", "etax2T", "etax2U", "shearU", "deltaU"]
C_WORK = C_FIELDS[14:]       # evp() zeroes these at its entry (ice_dyn_evp.F90:351-361): absent from `state` = zeros
C_INPUTS = ["strength", "cdn_ocnE", "aiE", "uocnE", "vocnE", "waterxE", "forcexE", "emassdti", "fmE", "uvelE_init",
            "TbE", "rheofactE", "cdn_ocnN", "aiN", "uocnN", "vocnN", "wateryN", "forceyN", "nmassdti", "fmN",
            "vvelN_init", "TbN", "rheofactN"]
C_STATIC = ["dxT", "dyT", "dxU", "dyU", "dxE", "dyE", "dxN", "dyN", "uarea", "tarea", "earea", "narea", "earear",
            "narear", "epm", "npm", "uvm", "hm", "DminTarea", "ratiodxN", "ratiodxNr", "ratiodyE", "ratiodyEr"]
C_MASKS = ["iceTmask", "iceUmask", "iceEmask", "iceNmask"]


def cgrid_subcycle(dom: OracleDomain, params: Params, ndte: int, state: dict, inputs: dict, static: dict,
                   masks: dict, visc_method: str = "avg_zeta") -> dict:
    """evp()'s subcycle loop for grid_ice = 'C' (ice_dyn_evp.F90:938-1099).  Everything is copied; returns C_FIELDS."""
    lib().evp_oracle_cgrid_subcycle.restype = None
    work = {k: (np.array(state[k], dtype=np.float64, order="C", copy=True) if k in state else np.zeros(dom.shape))
            for k in C_FIELDS}
    inp = [np.ascontiguousarray(inputs[k], dtype=np.float64) for k in C_INPUTS]
    st = [np.ascontiguousarray(static[k], dtype=np.float64) for k in C_STATIC]
    mk = [np.ascontiguousarray(masks[k], dtype=np.int32) for k in C_MASKS]
    fptr = (C.POINTER(C.c_double) * len(C_FIELDS))(*[_dp(work[k]) for k in C_FIELDS])
    iptr = (C.POINTER(C.c_double) * len(inp))(*[_dp(a) for a in inp])
    gptr = (C.POINTER(C.c_double) * len(st))(*[_dp(a) for a in st])
    lib().evp_oracle_cgrid_subcycle(C.byref(dom.c), C.byref(params), C.c_int(ndte),
                                    C.c_int(1 if visc_method == "avg_strength" else 0), fptr, iptr, gptr,
                                    *[m.ctypes.data_as(C.POINTER(C.c_int32)) for m in mk])
    return work


def deformations_c_t(dom: OracleDomain, e_factor, fields: dict, static: dict, tarear, iceTmask, prev: dict | None = None) -> dict:
    """deformationsC_T (ice_dyn_shared.F90:1968-2074) from the loop's final uvelE/vvelE/uvelN/vvelN/shearU; `prev`: the
    five arrays as they were (cells off the T list keep them; zeros when absent)."""
    lib().evp_oracle_deformations_c_t.restype = None
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    out = {k: (np.array(prev[k], dtype=np.float64, order="C", copy=True) if prev else np.zeros(dom.shape))
           for k in ("vort", "shear", "divu", "rdg_conv", "rdg_shear")}
    a = [f64(fields[k]) for k in ("uvelE", "vvelE", "uvelN", "vvelN")] + [f64(static[k]) for k in ("dxN", "dyE", "dxT", "dyT")] + \
        [f64(tarear), f64(static["uarea"]), f64(fields["shearU"])]
    tm = np.ascontiguousarray(iceTmask, dtype=np.int32)
    lib().evp_oracle_deformations_c_t(C.byref(dom.c), C.c_double(e_factor), *[_dp(x) for x in a],
                                      tm.ctypes.data_as(C.POINTER(C.c_int32)),
                                      *[_dp(out[k]) for k in ("vort", "shear", "divu", "rdg_conv", "rdg_shear")])
    return out


C_PREP_MASKS = ["tmask", "umaskCD", "emask", "nmask"]
C_PREP_FCOR = ["fcor_blk", "fcorE_blk", "fcorN_blk"]


def cgrid_prep(dom: OracleDomain, pp: PrepParams, static: dict, tfields: dict, state: dict, prev_inputs: dict | None = None) -> dict:
    """evp()'s preparation phase for grid_ice = 'C' (ice_dyn_evp.F90:383-735) minus ice strength and seabed stress.
    static: C_STATIC + C_PREP_MASKS + C_PREP_FCOR; tfields: PREP_T; state: the first 14 of C_FIELDS (absent = zeros) and the
    previous iceUmask / iceEmask / iceNmask; prev_inputs: C_INPUTS as the previous call left them (cells off the ice keep
    e.g. their old fmE).  Returns the updated state, C_INPUTS[1:] and the four masks."""
    lib().evp_oracle_cgrid_prep.restype = None
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    mk = [i32(static[k]) for k in C_PREP_MASKS]
    fc = [f64(static[k]) for k in C_PREP_FCOR]
    st = [f64(static[k]) for k in C_STATIC]
    T = [f64(tfields[k]) for k in PREP_T]
    out = {k: (np.array(state[k], dtype=np.float64, order="C", copy=True) if k in state else np.zeros(dom.shape))
           for k in C_FIELDS[:14]}
    for k in C_INPUTS:
        out[k] = (np.array(prev_inputs[k], dtype=np.float64, order="C", copy=True) if prev_inputs and k in prev_inputs
                  else np.zeros(dom.shape))
    for k in ("iceUmask", "iceEmask", "iceNmask"):
        out[k] = np.array(state[k], dtype=np.int32, order="C", copy=True) if k in state else np.zeros(dom.shape, dtype=np.int32)
    out["iceTmask"] = np.zeros(dom.shape, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    mptr = (C.POINTER(C.c_int32) * 4)(*[ip(a) for a in mk])
    cptr = (C.POINTER(C.c_double) * 3)(*[_dp(a) for a in fc])
    gptr = (C.POINTER(C.c_double) * len(st))(*[_dp(a) for a in st])
    Tptr = (C.POINTER(C.c_double) * len(T))(*[_dp(a) for a in T])
    fptr = (C.POINTER(C.c_double) * 14)(*[_dp(out[k]) for k in C_FIELDS[:14]])
    iptr = (C.POINTER(C.c_double) * len(C_INPUTS))(*[_dp(out[k]) for k in C_INPUTS])
    lib().evp_oracle_cgrid_prep(C.byref(dom.c), C.byref(pp), mptr, cptr, gptr, Tptr, fptr, iptr, ip(out["iceTmask"]),
                                ip(out["iceUmask"]), ip(out["iceEmask"]), ip(out["iceNmask"]))
    return out


def seabed_lkd_c(dom: OracleDomain, loc: str, k1, k2, alphab, threshold_hw, aice, vice, hwater, iceXmask):
    """seabed_stress_factor_LKD with grid_location = 'E' / 'N' (ice_dyn_shared.F90:1386-1460)."""
    lib().evp_oracle_seabed_lkd_c.restype = None
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a, v, h = f64(aice), f64(vice), f64(hwater)
    xm = np.ascontiguousarray(iceXmask, dtype=np.int32)
    out = np.zeros(dom.shape)
    lib().evp_oracle_seabed_lkd_c(C.byref(dom.c), C.c_int({"E": 2, "N": 3}[loc]), C.c_double(k1), C.c_double(k2),
                                  C.c_double(alphab), C.c_double(threshold_hw), _dp(a), _dp(v), _dp(h),
                                  xm.ctypes.data_as(C.POINTER(C.c_int32)), _dp(out))
    return out


HALO_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_int)
_halo_cb_keep = None


def set_halo_callback(fn):
    """Test hook: fn(array_pointer, field_loc, field_type) replaces every halo update of the oracle (None: back to the
    built-in one).  The pointer addresses the whole (nblocks, ny_block, nx_block) array being updated."""
    global _halo_cb_keep
    lib().evp_oracle_set_halo_callback.restype = None
    _halo_cb_keep = HALO_CB(fn) if fn is not None else None
    lib().evp_oracle_set_halo_callback(_halo_cb_keep if _halo_cb_keep is not None else C.cast(None, HALO_CB))
