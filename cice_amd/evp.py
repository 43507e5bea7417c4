"""Host-side front end of the MI355X-native EVP core.

This mirrors, in Python, the interface CICE's evp() uses for an alternative EVP
core (module ice_dyn_evp1d: dyn_evp1d_init / dyn_evp1d_run / dyn_evp1d_finalize,
cicecore/cicedyn/dynamics/ice_dyn_evp1d.F90:25,73,121) on top of the C ABI in
include/cice_evp_hip.h -- same field names, same argument meaning, fail-stop
error behaviour (abort_ice -> EvpHipError).  The Fortran binding a CICE
maintainer would use is cice_amd/fortran/ice_dyn_evp_hip.F90; both call exactly
the same shared library.

There is no CPU fallback: if libcice_evp_hip.so is missing or the HIP runtime
finds no device, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["CICE_EVP_HIP_LIBRARY"]) if os.environ.get("CICE_EVP_HIP_LIBRARY") else HERE / "libcice_evp_hip.so"   # (override: experiments)
LIB_TESTING_PATH = HERE / "libcice_evp_hip_testing.so"

BND = {"closed": 0, "open": 1, "cyclic": 2, "tripole": 3, "tripoleT": 4}

# order of the 32-entry field table == argument order of cice_evp_hip_run
FIELDS = [
    "stressp_1", "stressp_2", "stressp_3", "stressp_4",
    "stressm_1", "stressm_2", "stressm_3", "stressm_4",
    "stress12_1", "stress12_2", "stress12_3", "stress12_4",
    "strength", "cdn_ocnU", "aiU", "uocnU", "vocnU", "waterxU", "wateryU",
    "forcexU", "forceyU", "umassdti", "fmU", "strintxU", "strintyU", "TbU",
    "taubxU", "taubyU", "uvel", "vvel", "uvel_init", "vvel_init",
]
OUTPUTS = FIELDS[:12] + ["strintxU", "strintyU", "taubxU", "taubyU", "uvel", "vvel"]
EXPORTS = [
    "cice_evp_hip_abi_version", "cice_evp_hip_last_error", "cice_evp_hip_init",
    "cice_evp_hip_set_metrics", "cice_evp_hip_run", "cice_evp_hip_finalize",
    "cice_evp_hip_upload", "cice_evp_hip_subcycle", "cice_evp_hip_download", "cice_evp_hip_sync",
    "cice_evp_hip_comm_unique_id", "cice_evp_hip_comm_init", "cice_evp_hip_comm_info", "cice_evp_hip_get_timings",
    "cice_evp_hip_time_kernels", "cice_evp_hip_mark", "cice_evp_hip_seam_fin_plan",
    "cice_evp_hip_cgrid_set_prep_geometry", "cice_evp_hip_cgrid_prep", "cice_evp_hip_cgrid_seabed_lkd", "cice_evp_hip_cgrid_seabed_prob",
    "cice_evp_hip_cgrid_prep_finish", "cice_evp_hip_cgrid_fetch", "cice_evp_hip_cgrid_set_tb", "cice_evp_hip_describe_path",
    "cice_evp_hip_pin_host", "cice_evp_hip_set_post_geometry", "cice_evp_hip_deformations", "cice_evp_hip_dyn_finish",
    "cice_evp_hip_halo_export", "cice_evp_hip_halo_import", "cice_evp_hip_stress_halo", "cice_evp_hip_stress_halo_available", 
    "cice_evp_hip_set_prep_geometry", "cice_evp_hip_prep", "cice_evp_hip_set_strength", "cice_evp_hip_set_tbu", "cice_evp_hip_seabed_lkd", "cice_evp_hip_seabed_prob", "cice_evp_hip_halo_mask", "cice_evp_hip_march_info", "cice_evp_hip_prep_fetch",
    "cice_evp_hip_addr", "cice_evp_hip_set_option", "cice_evp_hip_fetch_stresses", "cice_evp_hip_invalidate_stresses",
    "cice_evp_hip_cgrid_set_geometry", "cice_evp_hip_cgrid_run", "cice_evp_hip_cgrid_upload", "cice_evp_hip_cgrid_subcycle",
    "cice_evp_hip_cgrid_download", "cice_evp_hip_cgrid_sync", "cice_evp_hip_cgrid_deformations", "cice_evp_hip_cgrid_dyn_finish", "cice_evp_hip_cgrid_timings", "cice_evp_hip_stream_probe", 
]
# libcice_evp_hip_testing.so only (include/cice_evp_hip_testing.h): plan introspection of the CPU tests, read-outs of the tools,
# the test transport
TEST_EXPORTS = [
    "cice_evp_hip_cgrid_fold_plan", "cice_evp_hip_cgrid_window_plan", "cice_evp_hip_cgrid_window_plan_ext", "cice_evp_hip_cgrid_window_deps", "cice_evp_hip_cgrid_strip_plan", "cice_evp_hip_set_test_transport", "cice_evp_hip_march_plan",
    "cice_evp_hip_debug_cuload", "cice_evp_hip_debug_prof", "cice_evp_hip_debug_cgrid_prof", "cice_evp_hip_debug_cgres_prof", "cice_evp_hip_plan_build", "cice_evp_hip_halo_plan", "cice_evp_hip_seam_plan",
    "cice_evp_hip_peer_plan", "cice_evp_hip_peer_signs", "cice_evp_hip_center_plan", "cice_evp_hip_stress_plan",
    "cice_evp_hip_fold_split_plan", "cice_evp_hip_plan_flags", "cice_evp_hip_fold_images_plan",
]
# environment switches only the test build reads (cice_amd/csrc/evp_host.h: env_test): experiments, fault injection, routing
# of on-device copies through the remote transports.  An EvpHip made while one of them is set uses the test build.
TEST_ENV = [
    "CICE_EVP_HIP_RES_REMOTE", "CICE_EVP_HIP_RES_REMOTE_BREAK", "CICE_EVP_HIP_RES_ORDER", "CICE_EVP_HIP_RES_PROF", "CICE_EVP_HIP_RES_DEBUG",
    "CICE_EVP_HIP_MARCH_OWN", "CICE_EVP_HIP_MARCH_SELFX", "CICE_EVP_HIP_MARCH_SEG", "CICE_EVP_HIP_MARCH_ORDER",
    "CICE_EVP_HIP_MARCH_LEAN", "CICE_EVP_HIP_MARCH_K", "CICE_EVP_HIP_CGRID_SPLIT", "CICE_EVP_HIP_CGRID_XCD", "CICE_EVP_HIP_CGRID_ONE_XCD", "CICE_EVP_HIP_CGRID_ONE_SHAPE",
    "CICE_EVP_HIP_CGRID_ONE_STRIP", "CICE_EVP_HIP_CGRID_FAST", "CICE_EVP_HIP_HALO_DEBUG", "CICE_EVP_HIP_SEAM_FIN", "CICE_EVP_HIP_OVERLAP",
    "CICE_EVP_HIP_HALO_RIDE", "CICE_EVP_HIP_GATHER", "CICE_EVP_HIP_SIMPLE", "CICE_EVP_HIP_SELF_EXCHANGE", "CICE_EVP_HIP_FLAGS", "CICE_EVP_HIP_LEAN",
    "CICE_EVP_HIP_PREFETCH", "CICE_EVP_HIP_FAULT_REPLAY", "CICE_EVP_HIP_MARCH_BANDSEG", "CICE_EVP_HIP_CGRID_PROF",
    # A/B switches of kernels and transports (forced tile shapes, schedules the default never picks, the ring exchange's other forms)
    "CICE_EVP_HIP_NO_OVERLAP", "CICE_EVP_HIP_TYB", "CICE_EVP_HIP_NOGRAPH", "CICE_EVP_HIP_GRAPH_RCCL", "CICE_EVP_HIP_RES_LOGW", "CICE_EVP_HIP_RES_COOP",
    "CICE_EVP_HIP_MARCH_EXT", "CICE_EVP_HIP_MARCH_DIRECT", "CICE_EVP_HIP_CGRID_FUSED", "CICE_EVP_HIP_CGRID_GEO",
    "CICE_EVP_HIP_CGRID_RES_SLEEP", "CICE_EVP_HIP_CGRID_RES_CULL", "CICE_EVP_HIP_CGRID_RES_DEBUG",
    "CICE_EVP_HIP_CGRID_STRIP", "CICE_EVP_HIP_CGRID_STRIP_SEG", "CICE_EVP_HIP_CGRID_STRIP_EDGE", "CICE_EVP_HIP_CGRID_STRIP_RIDE", "CICE_EVP_HIP_CGRID_STRIP_LEN", "CICE_EVP_HIP_CGRID_STRIP_ITEMS", "CICE_EVP_HIP_CGRID_STRIP_LAST",
]
# C-grid subcycle (cice_evp_hip_cgrid_*): order of the pointer tables, see include/cice_evp_hip.h
CGRID_FIELDS = ["uvelE", "vvelE", "uvelN", "vvelN", "uvel", "vvel", "stresspT", "stressmT", "stress12T", "stress12U",
                "strintxE", "strintyN", "taubxE", "taubyN", "zetax2T", "etax2T", "etax2U", "shearU", "deltaU"]
CGRID_INPUTS = ["strength", "cdn_ocnE", "aiE", "uocnE", "vocnE", "waterxE", "forcexE", "emassdti", "fmE", "uvelE_init",
                "TbE", "rheofactE", "cdn_ocnN", "aiN", "uocnN", "vocnN", "wateryN", "forceyN", "nmassdti", "fmN",
                "vvelN_init", "TbN", "rheofactN"]
CGRID_STATIC = ["dxT", "dyT", "dxU", "dyU", "dxE", "dyE", "dxN", "dyN", "uarea", "tarea", "earea", "narea", "earear",
                "narear", "epm", "npm", "uvm", "hm", "DminTarea", "ratiodxN", "ratiodxNr", "ratiodyE", "ratiodyEr"]
CGRID_MASKS = ["iceTmask", "iceUmask", "iceEmask", "iceNmask"]
VISC_METHOD = {"avg_zeta": 0, "avg_strength": 1}
HALO_BLOB = 1024   # CICE_EVP_HIP_HALO_BLOB
OPT_STRESS_RESIDENT = 1   # CICE_EVP_HIP_OPT_STRESS_RESIDENT
# T-grid inputs of the preparation phase (order of cice_evp_hip_prep's tfields11) and the
# products cice_evp_hip_prep_fetch serves (index = `which`)
PREP_T = ["aice", "vice", "vsno", "aice_init", "cdn_ocn", "uocn", "vocn", "ss_tltx", "ss_tlty",
          "strairxT", "strairyT"]
PREP_FETCH = ["aiU", "cdn_ocnU", "uocnU", "vocnU", "umassdti", "fmU", "waterxU", "wateryU", "forcexU",
              "forceyU", "uvel_init", "vvel_init", "strtltxU", "strtltyU", "strairxU", "strairyU",
              "tmass", "umass", "uvel", "vvel"]
PREP_FETCH_MORE = {"TbU": 20}      # further products cice_evp_hip_prep_fetch serves (not made by cice_evp_hip_prep itself)


class PrepParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("rhoi", C.c_double), ("rhos", C.c_double), ("gravit", C.c_double),
                ("dyn_area_min", C.c_double), ("dyn_mass_min", C.c_double), ("ssh_stress_coupled", C.c_int32)]

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class EvpHipError(RuntimeError):
    """Raised where the Fortran wrapper would call abort_ice(...)."""


class Dims(C.Structure):
    _fields_ = [("nx_block", C.c_int32), ("ny_block", C.c_int32), ("nblocks", C.c_int32),
                ("max_blocks", C.c_int32), ("nghost", C.c_int32), ("nx_global", C.c_int32),
                ("ny_global", C.c_int32), ("ew_boundary_type", C.c_int32),
                ("ns_boundary_type", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32),
                ("ilo", _i32p), ("ihi", _i32p), ("jlo", _i32p), ("jhi", _i32p),
                ("iglob0", _i32p), ("jglob0", _i32p), ("nblocks_tot", C.c_int32),
                ("gi0", _i32p), ("gj0", _i32p), ("gnx", _i32p), ("gny", _i32p),
                ("gowner", _i32p), ("glocal", _i32p)]


class Params(C.Structure):
    _fields_ = [("ndte", C.c_int32), ("strict", C.c_int32)] + [(n, C.c_double) for n in (
        "arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping", "Ktens",
        "deltaminEVP", "u0", "cosw", "sinw", "rhow")]


_lib = None
_lib_testing = None


def testing_wanted() -> bool:
    """True while the process environment holds a switch only the test build reads."""
    return any(k in os.environ for k in TEST_ENV)


def load_library(path: os.PathLike | None = None, testing: bool = False) -> C.CDLL:
    """dlopen the product library (testing=True: the test build, which adds include/cice_evp_hip_testing.h); raises if it has
    not been built.  Both may live in one process: each was linked -Bsymbolic and keeps its own device state."""
    global _lib, _lib_testing
    if path is None:
        if testing and _lib_testing is not None:
            return _lib_testing
        if not testing and _lib is not None:
            return _lib
    p = Path(path) if path else (LIB_TESTING_PATH if testing else LIB_PATH)
    if not p.exists():
        raise EvpHipError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(str(p), mode=C.RTLD_LOCAL)
    for name in EXPORTS + (TEST_EXPORTS if testing else []):
        getattr(lib, name).restype = C.c_int
    lib.cice_evp_hip_addr.restype = C.c_void_p        # the one entry point that does not return a status
    if path is None:
        if testing:
            _lib_testing = lib
        else:
            _lib = lib
    return lib


def _ip(a):
    return a.ctypes.data_as(_i32p)


def _dp(a):
    return a.ctypes.data_as(_f64p)


def _check(lib, rc, what):
    if rc != 0:
        buf = C.create_string_buffer(1024)
        lib.cice_evp_hip_last_error(buf, 1024)
        raise EvpHipError(f"{what} ERROR: rc={rc}: {buf.value.decode(errors='replace')}")


def make_dims(decomp, rank: int = 0):
    """(Dims, keepalive) for `rank` of a cice_amd.decomp.Decomp."""
    blks = decomp.local_blocks(rank)
    loc = [np.array(v, dtype=np.int32) for v in (
        [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks], [b.jhi for b in blks],
        [b.gi0 for b in blks], [b.gj0 for b in blks])]
    allb = decomp.blocks
    tab = [np.array(v, dtype=np.int32) for v in (
        [b.gi0 for b in allb], [b.gj0 for b in allb], [b.gnx for b in allb], [b.gny for b in allb],
        [b.owner for b in allb], [b.local for b in allb])]
    d = Dims(decomp.nx_block, decomp.ny_block, len(blks), len(blks), 1, decomp.nx_global,
             decomp.ny_global, BND[decomp.ew], BND[decomp.ns], rank, decomp.nranks,
             *[_ip(a) for a in loc], len(allb), *[_ip(a) for a in tab])
    return d, (loc, tab)


def make_params(scal: dict, strict: bool = False) -> Params:
    p = Params()
    p.ndte = int(scal.get("ndte", 120))
    p.strict = 1 if strict else 0
    for k in ("arlx1i", "denom1", "brlx", "revp", "e_factor", "epp2i", "capping", "Ktens",
              "deltaminEVP", "u0", "cosw", "sinw", "rhow"):
        setattr(p, k, float(scal[k]))
    return p


def fold_split_plan() -> dict:
    """Lists of the shifted-copy exchange of the plan built last (halo_plan(): cice_evp_hip_plan_build)."""
    lib = load_library(testing=True)
    out = {}
    for which, name in enumerate(("shift_cells", "center_dst", "stress_dst", "seam_dst", "seam_slot", "stress_own_dst", "stress_own_src",
                                  "stress_corner_dst", "stress_corner_src")):
        n = C.c_int32(0)
        split = lib.cice_evp_hip_fold_split_plan(C.c_int32(which), C.byref(n), None)
        a = np.zeros(max(n.value, 1), dtype=np.int32)
        lib.cice_evp_hip_fold_split_plan(C.c_int32(which), C.byref(n), _ip(a))
        out[name] = a[:n.value]
        out["fold_split"] = bool(split)
    return out


def cgrid_fold_plan(dims: "Dims", loc: str) -> dict:
    """Host only: the C-grid fold step of one field location on a tripole grid (see the header)."""
    lib = load_library(testing=True)
    code = {"center": 0, "NEcorner": 1, "Eface": 2, "Nface": 3}[loc]
    n = C.c_int32(0)
    _check(lib, lib.cice_evp_hip_cgrid_fold_plan(C.byref(dims), C.c_int32(code), C.byref(n), None, None, None, None), "(cgrid_fold_plan)")
    out = {k: np.zeros(n.value, dtype=np.int32) for k in ("dst", "a", "b", "flip")}
    _check(lib, lib.cice_evp_hip_cgrid_fold_plan(C.byref(dims), C.c_int32(code), C.byref(n), *[_ip(out[k]) for k in ("dst", "a", "b", "flip")]),
           "(cgrid_fold_plan)")
    return out


def cgrid_window_plan(dims: "Dims", ox: int, oy: int, extra: int = 0) -> dict:
    """Host only: the window table of the C grid's one-launch kernel (extra = 1: of the on-chip resident one; extra = 2: of the
    resident one on a tripole grid, 17 x 17 positions with a mirrored mini-tile in the windows at the fold; see the header)."""
    lib = load_library(testing=True)
    n = C.c_int32(0)
    a = (C.byref(dims), C.c_int32(ox), C.c_int32(oy), C.c_int32(extra), C.byref(n))
    _check(lib, lib.cice_evp_hip_cgrid_window_plan_ext(*a, None, None), "(cgrid_window_plan)")
    tiles = np.zeros((n.value, 4), dtype=np.int32)
    tab = np.zeros((n.value, 17, 17) if extra == 2 else (n.value, oy + extra, ox + extra), dtype=np.int32)
    _check(lib, lib.cice_evp_hip_cgrid_window_plan_ext(*a, _ip(tiles), _ip(tab)), "(cgrid_window_plan)")
    return dict(tiles=tiles, tab=tab)


def cgrid_window_deps(dims: "Dims") -> dict:
    """Host only: the hand-off graph of the on-chip resident C-grid kernel's windows (see the testing header)."""
    lib = load_library(testing=True)
    nw, ne, n1, nu = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _check(lib, lib.cice_evp_hip_cgrid_window_deps(C.byref(dims), C.byref(nw), C.byref(ne), C.byref(n1), C.byref(nu)), "(cgrid_window_deps)")
    return dict(windows=nw.value, edges=ne.value, oneway=n1.value, unsafe=nu.value)


def cgrid_strip_plan(dims: "Dims", ex=32, ey=8, lo0=3, slots=2048, seg_min=8, seg=0) -> dict:
    """Host only: the marched C-grid kernel's work items and the windows the windowed kernel keeps (see the testing header)."""
    lib = load_library(testing=True)
    ni, nw, sr = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    a = (C.byref(dims), C.c_int32(ex), C.c_int32(ey), C.c_int32(lo0), C.c_int32(slots), C.c_int32(seg_min), C.c_int32(seg))
    _check(lib, lib.cice_evp_hip_cgrid_strip_plan(*a, C.byref(ni), None, C.c_int32(0), C.byref(nw), None, None, C.c_int32(0), C.byref(sr)), "(cgrid_strip_plan)")
    items = np.zeros((max(ni.value, 1), 6), dtype=np.int32)
    tiles = np.zeros((max(nw.value, 1), 4), dtype=np.int32)
    inz = np.zeros(max(nw.value, 1), dtype=np.uint8)
    _check(lib, lib.cice_evp_hip_cgrid_strip_plan(*a, C.byref(ni), _ip(items), C.c_int32(len(items)), C.byref(nw), _ip(tiles),
                                                  inz.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int32(len(tiles)), C.byref(sr)), "(cgrid_strip_plan)")
    return dict(items=items[:ni.value], tiles=tiles[:nw.value], in_zone=inz[:nw.value].astype(bool), segment_rows=sr.value)


def stream_probe(ncells: int) -> float:
    """Bytes/s of a plain 30-in / 16-out streaming kernel over `ncells` elements per array (see the header)."""
    lib = load_library()
    out = C.c_double(0.0)
    _check(lib, lib.cice_evp_hip_stream_probe(C.c_int64(ncells), C.byref(out)), "(dyn_evp_hip_stream_probe)")
    return out.value


class EvpHip:
    """One EVP core per process (the library keeps a single device state, like the
    module-level state of ice_dyn_evp1d)."""

    def __init__(self, dims: Dims, params: Params, HTE, HTN, dxT, dyT, uarear, tarea, keepalive=None, testing: bool | None = None):
        # the product library unless the caller asks for the test build or the environment holds one of its switches
        self.testing = testing_wanted() if testing is None else bool(testing)
        self.lib = load_library(testing=self.testing)
        self._keep = keepalive
        self.shape = (dims.nblocks, dims.ny_block, dims.nx_block)
        arrs = [self._c(a) for a in (HTE, HTN, dxT, dyT, uarear, tarea)]
        rc = self.lib.cice_evp_hip_init(C.byref(dims), C.byref(params), *[_dp(a) for a in arrs])
        _check(self.lib, rc, "(dyn_evp_hip_init)")
        self.ndte = params.ndte
        self._open = True

    # -- helpers ---------------------------------------------------------------
    def _c(self, a, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.shape != self.shape:
            raise EvpHipError(f"array shape {a.shape} != (nblocks, ny_block, nx_block) {self.shape}")
        return a

    def set_metrics(self, **m):
        order = ("cxp", "cyp", "cxm", "cym", "dxhy", "dyhx", "DminTarea")
        arrs = [self._c(m[k]) if k in m and m[k] is not None else None for k in order]
        rc = self.lib.cice_evp_hip_set_metrics(*[(_dp(a) if a is not None else None) for a in arrs])
        _check(self.lib, rc, "(dyn_evp_hip_set_metrics)")

    def pin_host(self, *arrays):
        """Page-lock caller-owned arrays that stay alive until finalize() (see the header)."""
        for a in arrays:
            _check(self.lib, self.lib.cice_evp_hip_pin_host(C.c_void_p(a.ctypes.data), C.c_int64(a.nbytes)),
                   "(dyn_evp_hip_pin_host)")

    def set_option(self, key: int, value: int):
        _check(self.lib, self.lib.cice_evp_hip_set_option(C.c_int32(key), C.c_int32(value)), "(dyn_evp_hip_set_option)")

    def fetch_stresses(self, out: dict | None = None) -> dict:
        out = out if out is not None else {k: np.zeros(self.shape) for k in FIELDS[:12]}
        tab = (_f64p * 12)(*[_dp(out[k]) for k in FIELDS[:12]])
        _check(self.lib, self.lib.cice_evp_hip_fetch_stresses(tab), "(dyn_evp_hip_fetch_stresses)")
        return out

    def invalidate_stresses(self):
        _check(self.lib, self.lib.cice_evp_hip_invalidate_stresses(), "(dyn_evp_hip_invalidate_stresses)")

    def run_inplace(self, work: dict, tm, um, ndte: int | None = None):
        """Like run() but directly on the caller's arrays (no copies): the Fortran call shape."""
        args = [(_dp(work[k]) if k in work and work[k] is not None else None) for k in FIELDS]
        rc = self.lib.cice_evp_hip_run(*args, _ip(tm), _ip(um), C.c_int32(self.ndte if ndte is None else ndte))
        _check(self.lib, rc, "(dyn_evp_hip_run)")

    # -- dyn_evp1d_run equivalent ----------------------------------------------
    def run(self, fields: dict, iceTmask, iceUmask, ndte: int | None = None) -> dict:
        """H2D, ndte subcycles, D2H.  `fields` maps FIELDS -> arrays; the inout/out
        ones (OUTPUTS) are updated in place in copies that are returned."""
        work = {k: np.array(self._c(fields[k]), copy=True) for k in FIELDS
                if k in fields and fields[k] is not None}
        tm = self._c(iceTmask, np.int32)
        um = self._c(iceUmask, np.int32)
        args = [(_dp(work[k]) if k in work else None) for k in FIELDS]
        rc = self.lib.cice_evp_hip_run(*args, _ip(tm), _ip(um), C.c_int32(self.ndte if ndte is None else ndte))
        _check(self.lib, rc, "(dyn_evp_hip_run)")
        return {k: work[k] for k in OUTPUTS}

    # -- resident-state entry points ----------------------------------------------
    def upload(self, fields: dict, iceTmask, iceUmask):
        self._up = [self._c(fields[k]) if k in fields and fields[k] is not None else None for k in FIELDS]
        tab = (_f64p * len(FIELDS))(*[(_dp(a) if a is not None else None) for a in self._up])
        tm = self._c(iceTmask, np.int32)
        um = self._c(iceUmask, np.int32)
        rc = self.lib.cice_evp_hip_upload(tab, _ip(tm), _ip(um))
        _check(self.lib, rc, "(dyn_evp_hip_upload)")

    # -- next tier f-2: evp()'s preparation phase on the device ---------------------------
    def set_prep_geometry(self, tmask, umask, hm, tarea, uarea, fcor_blk):
        tm, um = self._c(tmask, np.int32), self._c(umask, np.int32)
        a = [self._c(x) for x in (hm, tarea, uarea, fcor_blk)]
        _check(self.lib, self.lib.cice_evp_hip_set_prep_geometry(_ip(tm), _ip(um), *[_dp(x) for x in a]),
               "(dyn_evp_hip_set_prep_geometry)")

    def prep(self, pp: "PrepParams", tfields: dict, state: dict):
        """state: the 12 stresses, uvel, vvel, iceUmask (previous call), optional TbU and
        strintxU/strintyU/strocnxU/strocnyU.  Returns (iceTmask, iceUmask, zeroed dict)."""
        t = [self._c(tfields[k]) for k in PREP_T]
        ttab = (_f64p * 11)(*[_dp(a) for a in t])
        f = [self._c(state[k]) if k in state and state[k] is not None and k in FIELDS[:12] + ["uvel", "vvel", "TbU"]
             else None for k in FIELDS]
        ftab = (_f64p * len(FIELDS))(*[(_dp(a) if a is not None else None) for a in f])
        tm = np.zeros(self.shape, dtype=np.int32)
        um = np.array(state["iceUmask"], dtype=np.int32, order="C", copy=True).reshape(self.shape)
        z = {k: (np.array(state[k], dtype=np.float64, order="C", copy=True) if state.get(k) is not None else None)
             for k in ("strintxU", "strintyU", "strocnxU", "strocnyU")}
        rc = self.lib.cice_evp_hip_prep(C.byref(pp), ttab, ftab, _ip(tm), _ip(um),
                                        *[(_dp(z[k]) if z[k] is not None else None)
                                          for k in ("strintxU", "strintyU", "strocnxU", "strocnyU")])
        _check(self.lib, rc, "(dyn_evp_hip_prep)")
        return tm, um, z

    def set_strength(self, strength):
        a = self._c(strength)
        _check(self.lib, self.lib.cice_evp_hip_set_strength(_dp(a)), "(dyn_evp_hip_set_strength)")

    def set_tbu(self, TbU):
        a = self._c(TbU)
        _check(self.lib, self.lib.cice_evp_hip_set_tbu(_dp(a)), "(dyn_evp_hip_set_tbu)")

    def seabed_lkd(self, hwater, k1, k2, alphab, threshold_hw):
        a = self._c(hwater) if hwater is not None else None
        _check(self.lib, self.lib.cice_evp_hip_seabed_lkd(_dp(a) if a is not None else None, C.c_double(k1), C.c_double(k2),
                                                           C.c_double(alphab), C.c_double(threshold_hw)), "(dyn_evp_hip_seabed_lkd)")

    def seabed_prob(self, hwater, aicen, vicen, alphab, rhoi, gravit, pi, puny):
        """aicen / vicen: [nblocks][ncat][ny][nx] (the memory image of ice_state's (nx, ny, ncat, blocks))."""
        h = self._c(hwater) if hwater is not None else None
        a = np.ascontiguousarray(aicen, dtype=np.float64)
        v = np.ascontiguousarray(vicen, dtype=np.float64)
        _check(self.lib, self.lib.cice_evp_hip_seabed_prob(_dp(h) if h is not None else None, _dp(a), _dp(v), C.c_int32(a.shape[1]),
                                                            C.c_double(alphab), C.c_double(rhoi), C.c_double(gravit), C.c_double(pi),
                                                            C.c_double(puny)), "(dyn_evp_hip_seabed_prob)")

    # -- C-grid subcycle -------------------------------------------
    def cgrid_set_geometry(self, static: dict):
        arrs = [self._c(static[k]) for k in CGRID_STATIC]
        tab = (_f64p * len(arrs))(*[_dp(a) for a in arrs])
        _check(self.lib, self.lib.cice_evp_hip_cgrid_set_geometry(tab), "(dyn_evp_hip_cgrid_set_geometry)")

    def cgrid_upload(self, state: dict, inputs: dict, masks: dict, visc_method: str = "avg_zeta"):
        st = [self._c(state[k]) for k in CGRID_FIELDS[:14]]
        inp = [self._c(inputs[k]) for k in CGRID_INPUTS]
        mk = [self._c(masks[k], np.int32) for k in CGRID_MASKS]
        rc = self.lib.cice_evp_hip_cgrid_upload((_f64p * 14)(*[_dp(a) for a in st]), (_f64p * len(inp))(*[_dp(a) for a in inp]),
                                                *[_ip(m) for m in mk], C.c_int32(VISC_METHOD[visc_method]))
        _check(self.lib, rc, "(dyn_evp_hip_cgrid_upload)")

    # -- the preparation phase on the C grid (cice_evp_hip_cgrid_prep) ---------------
    def cgrid_set_prep_geometry(self, static: dict):
        m = [self._c(static[k], np.int32) for k in ("tmask", "umaskCD", "emask", "nmask")]
        f = [self._c(static[k]) for k in ("fcor_blk", "fcorE_blk", "fcorN_blk")]
        _check(self.lib, self.lib.cice_evp_hip_cgrid_set_prep_geometry(*[_ip(a) for a in m], *[_dp(a) for a in f]),
               "(dyn_evp_hip_cgrid_set_prep_geometry)")

    def cgrid_prep(self, pp: "PrepParams", tfields: dict, state: dict | None, masks_prev: dict) -> dict:
        """state: the first 12 of CGRID_FIELDS as evp() is entered with them (None: keep what the device holds);
        masks_prev: iceUmask, iceEmask, iceNmask of the previous call.  Returns the four new masks."""
        t = [self._c(tfields[k]) for k in PREP_T]
        ttab = (_f64p * 11)(*[_dp(a) for a in t])
        stab = None
        if state is not None:
            st = [self._c(state[k]) for k in CGRID_FIELDS[:12]]
            stab = (_f64p * 12)(*[_dp(a) for a in st])
        out = {k: np.array(masks_prev[k], dtype=np.int32, order="C", copy=True).reshape(self.shape)
               for k in ("iceUmask", "iceEmask", "iceNmask")}
        out["iceTmask"] = np.zeros(self.shape, dtype=np.int32)
        rc = self.lib.cice_evp_hip_cgrid_prep(C.byref(pp), ttab, stab, *[_ip(out[k]) for k in CGRID_MASKS])
        _check(self.lib, rc, "(dyn_evp_hip_cgrid_prep)")
        return out

    def cgrid_seabed_lkd(self, hwater, k1, k2, alphab, threshold_hw):
        a = self._c(hwater) if hwater is not None else None
        _check(self.lib, self.lib.cice_evp_hip_cgrid_seabed_lkd(_dp(a) if a is not None else None, C.c_double(k1), C.c_double(k2),
                                                                 C.c_double(alphab), C.c_double(threshold_hw)),
               "(dyn_evp_hip_cgrid_seabed_lkd)")

    def cgrid_seabed_prob(self, hwater, aicen, vicen, alphab, rhoi, gravit, pi, puny):
        h = self._c(hwater) if hwater is not None else None
        a = np.ascontiguousarray(aicen, dtype=np.float64)
        v = np.ascontiguousarray(vicen, dtype=np.float64)
        _check(self.lib, self.lib.cice_evp_hip_cgrid_seabed_prob(_dp(h) if h is not None else None, _dp(a), _dp(v),
                                                                  C.c_int32(a.shape[1]), C.c_double(alphab), C.c_double(rhoi),
                                                                  C.c_double(gravit), C.c_double(pi), C.c_double(puny)),
               "(dyn_evp_hip_cgrid_seabed_prob)")

    def cgrid_prep_finish(self, strength, visc_method: str = "avg_zeta"):
        a = self._c(strength)
        _check(self.lib, self.lib.cice_evp_hip_cgrid_prep_finish(_dp(a), C.c_int32(VISC_METHOD[visc_method])),
               "(dyn_evp_hip_cgrid_prep_finish)")

    def cgrid_fetch(self, name: str):
        """One of CGRID_FIELDS / CGRID_INPUTS as it is on the device."""
        table, index = (0, CGRID_FIELDS.index(name)) if name in CGRID_FIELDS else (1, CGRID_INPUTS.index(name))
        out = np.zeros(self.shape)
        _check(self.lib, self.lib.cice_evp_hip_cgrid_fetch(C.c_int32(table), C.c_int32(index), _dp(out)), "(dyn_evp_hip_cgrid_fetch)")
        return out

    def cgrid_subcycle(self, ndte: int):
        _check(self.lib, self.lib.cice_evp_hip_cgrid_subcycle(C.c_int32(ndte)), "(dyn_evp_hip_cgrid_subcycle)")

    def cgrid_dyn_finish(self, prev: dict | None = None) -> dict:
        """dyn_finish at N and E points from the resident state; prev: the four inout arrays (default zeros)."""
        keys = ["strocnxN", "strocnyN", "strocnxE", "strocnyE"]
        out = {k: (np.array(self._c(prev[k]), copy=True) if prev and k in prev else np.zeros(self.shape)) for k in keys}
        _check(self.lib, self.lib.cice_evp_hip_cgrid_dyn_finish(*[_dp(out[k]) for k in keys]), "(dyn_evp_hip_cgrid_dyn_finish)")
        return out

    def cgrid_deformations(self, tarear, prev: dict | None = None) -> dict:
        """deformationsC_T on the resident final state of the C-grid loop; `prev`: the five inout arrays (zeros if absent)."""
        keys = ("divu", "shear", "vort", "rdg_conv", "rdg_shear")
        out = {k: (np.array(prev[k], dtype=np.float64, order="C", copy=True) if prev else np.zeros(self.shape)) for k in keys}
        t = self._c(tarear) if tarear is not None else None
        _check(self.lib, self.lib.cice_evp_hip_cgrid_deformations(_dp(t) if t is not None else None, *[_dp(out[k]) for k in keys]),
               "(dyn_evp_hip_cgrid_deformations)")
        return out

    def cgrid_download(self) -> dict:
        out = {k: np.zeros(self.shape) for k in CGRID_FIELDS}
        tab = (_f64p * len(CGRID_FIELDS))(*[_dp(out[k]) for k in CGRID_FIELDS])
        _check(self.lib, self.lib.cice_evp_hip_cgrid_download(tab), "(dyn_evp_hip_cgrid_download)")
        return out

    def cgrid_run(self, ndte: int, state: dict, inputs: dict, masks: dict, visc_method: str = "avg_zeta") -> dict:
        """cice_evp_hip_cgrid_run on copies of `state` (the Fortran call shape: arrays updated in place)."""
        work = {k: (np.array(state[k], dtype=np.float64, order="C", copy=True) if k in state else np.zeros(self.shape))
                for k in CGRID_FIELDS}
        inp = [self._c(inputs[k]) for k in CGRID_INPUTS]
        mk = [self._c(masks[k], np.int32) for k in CGRID_MASKS]
        rc = self.lib.cice_evp_hip_cgrid_run(C.c_int32(ndte), C.c_int32(VISC_METHOD[visc_method]),
                                             (_f64p * len(CGRID_FIELDS))(*[_dp(work[k]) for k in CGRID_FIELDS]),
                                             (_f64p * len(inp))(*[_dp(a) for a in inp]), *[_ip(m) for m in mk])
        _check(self.lib, rc, "(dyn_evp_hip_cgrid_run)")
        return work

    def cgrid_sync(self):
        _check(self.lib, self.lib.cice_evp_hip_cgrid_sync(), "(dyn_evp_hip_cgrid_sync)")

    def cgrid_timings(self):
        out = np.zeros(15)
        _check(self.lib, self.lib.cice_evp_hip_cgrid_timings(_dp(out), C.c_int32(15)), "(dyn_evp_hip_cgrid_timings)")
        return dict(loop_ms=float(out[0]), nsub=int(out[1]), prep_ms=float(out[2]), one_launch_subcycles=int(out[3]),
                    geometry_derived=bool(out[4]), resident_subcycles=int(out[5]), resident_probe_ms=float(out[6]),
                    resident_fallbacks=int(out[7]), resident_windows_with_ice=int(out[8]), resident_windows=int(out[9]),
                    marched_items=int(out[10]), marched_cells=int(out[11]), marched_edge_windows=int(out[12]),
                    marched_segment_rows=int(out[13]), marched_lengths_derived=bool(out[14]))

    def debug_cgres_prof(self):
        self._need_testing("debug_cgres_prof")
        out = np.zeros((4096, 4, 8), dtype=np.uint64)
        n = self.lib.cice_evp_hip_debug_cgres_prof(out.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(4096))
        if n < 0:
            _check(self.lib, n, "(debug_cgres_prof)")
        return out[:n]

    def prep_fetch(self, name: str):
        out = np.zeros(self.shape)
        which = PREP_FETCH_MORE[name] if name in PREP_FETCH_MORE else PREP_FETCH.index(name)
        _check(self.lib, self.lib.cice_evp_hip_prep_fetch(C.c_int32(which), _dp(out)),
               "(dyn_evp_hip_prep_fetch)")
        return out

    def subcycle(self, ndte: int | None = None):
        rc = self.lib.cice_evp_hip_subcycle(C.c_int32(self.ndte if ndte is None else ndte))
        _check(self.lib, rc, "(dyn_evp_hip_subcycle)")

    def sync(self):
        _check(self.lib, self.lib.cice_evp_hip_sync(), "(dyn_evp_hip_sync)")

    def download_into(self, out: dict):
        """D2H into the caller's (ideally page-locked) arrays; fields absent from `out` stay on the device."""
        tab = (_f64p * len(FIELDS))(*[(_dp(out[k]) if k in out else None) for k in FIELDS])
        _check(self.lib, self.lib.cice_evp_hip_download(tab), "(dyn_evp_hip_download)")

    def download(self, skip_stresses: bool = False) -> dict:
        out = {k: np.zeros(self.shape) for k in OUTPUTS if not (skip_stresses and k in FIELDS[:12])}
        tab = (_f64p * len(FIELDS))(*[(_dp(out[k]) if k in out else None) for k in FIELDS])
        rc = self.lib.cice_evp_hip_download(tab)
        _check(self.lib, rc, "(dyn_evp_hip_download)")
        return out

    def timings(self) -> dict:
        t = np.zeros(16)
        self.lib.cice_evp_hip_get_timings(_dp(t), 16)
        return dict(loop_ms=t[0], h2d_ms=t[1], d2h_ms=t[2], nsub=int(t[3]), launches_per_subcycle=t[4],
                    tile_variant=int(t[5]), marks_ms=t[6], stream_probe_ms=t[7], resident_probe_ms=t[8],
                    halo_transport={0: "none", 1: "rccl", 2: "mailbox"}[int(t[9])], prep_ms=t[10], resident_fallbacks=int(t[11]),
                    halo_send_cells=int(t[12]), halo_recv_cells=int(t[13]), resident_tiles_run=int(t[14]), resident_tiles=int(t[15]))

    def _need_testing(self, what):
        if not self.testing:
            raise EvpHipError(f"{what} exists in the test build only: EvpHip(..., testing=True)")

    def set_test_transport(self, xchg, reduce):
        """Test hook (see the header): xchg(peer_ranks, send_counts, recv_counts, send ndarray, recv ndarray) and
        reduce(op, value) -> value, called on the host by the marching path instead of RCCL."""
        self._need_testing("set_test_transport")
        XF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                         C.POINTER(C.c_double), C.POINTER(C.c_double))
        RF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p)

        def x_c(user, npeers, pr, sc, rc, send, recv):
            try:
                ranks = [pr[q] for q in range(npeers)]
                ns = [sc[q] for q in range(npeers)]
                nr = [rc[q] for q in range(npeers)]
                sa = np.ctypeslib.as_array(send, shape=(max(sum(ns), 1),))
                ra = np.ctypeslib.as_array(recv, shape=(max(sum(nr), 1),))
                xchg(ranks, ns, nr, sa, ra)
                return 0
            except Exception as e:  # noqa: BLE001
                print("test transport xchg failed:", e, flush=True)
                return 1

        def r_c(user, op, ptr):
            try:
                if op == 0:
                    v = C.cast(ptr, C.POINTER(C.c_int32))
                else:
                    v = C.cast(ptr, C.POINTER(C.c_uint32))
                v[0] = int(reduce(int(op), int(v[0])))
                return 0
            except Exception as e:  # noqa: BLE001
                print("test transport reduce failed:", e, flush=True)
                return 1

        self._test_cb = (XF(x_c), RF(r_c))
        _check(self.lib, self.lib.cice_evp_hip_set_test_transport(self._test_cb[0], self._test_cb[1], None), "(set_test_transport)")

    def comm_info(self) -> dict:
        """What RCCL reports about this rank's communicator, and the PCI bus id of its device."""
        v = np.full(5, -1, dtype=np.int32)
        buf = C.create_string_buffer(64)
        _check(self.lib, self.lib.cice_evp_hip_comm_info(_ip(v), 5, buf, C.c_int32(64)), "(comm_info)")
        return dict(have_comm=bool(v[0] == 1), rccl_nranks=int(v[1]), rccl_rank=int(v[2]), rccl_device=int(v[3]),
                    hip_device=int(v[4]), device_bus_id=buf.value.decode(errors="replace"))

    def describe_path(self) -> str:
        buf = C.create_string_buffer(600)
        _check(self.lib, self.lib.cice_evp_hip_describe_path(buf, C.c_int32(600)), "(describe_path)")
        return buf.value.decode(errors="replace")

    def march_info(self) -> dict:
        """The marching path (evp_march.hip): did it run, how is the domain cut."""
        v = np.zeros(10, dtype=np.int32)
        v[7] = -1
        self.lib.cice_evp_hip_march_info(_ip(v), 10)
        return dict(mode=int(v[0]), passes=int(v[1]), declined=int(v[2]), strips=int(v[3]), segments=int(v[4]),
                    seglen=int(v[5]), last_call=bool(v[6]), kpass=int(v[8]), subcycles=int(v[9]),
                    ring={-1: "not set up", 0: "rccl", 1: "direct stores (HIP IPC)", 2: "direct, on trial"}.get(int(v[7]), "?"))

    def halo_mask(self, halomask):
        """ice_HaloMask for the in-loop velocity exchange; halomask None = full halo."""
        a = self._c(halomask, np.int32) if halomask is not None else None
        _check(self.lib, self.lib.cice_evp_hip_halo_mask(_ip(a) if a is not None else None), "(dyn_evp_hip_halo_mask)")

    def stress_halo(self):
        """Tripole: 12 x ice_HaloUpdate_stress on the resident stresses (ice_dyn_evp.F90:1321-1389)."""
        _check(self.lib, self.lib.cice_evp_hip_stress_halo(), "(dyn_evp_hip_stress_halo)")

    def mark(self, which: int):
        _check(self.lib, self.lib.cice_evp_hip_mark(C.c_int32(which)), "(dyn_evp_hip_mark)")

    # -- next tier: deformations / dyn_finish on the resident final state -----------------
    def set_post_geometry(self, dxU, dyU, tarear):
        a = [self._c(x) for x in (dxU, dyU, tarear)]
        _check(self.lib, self.lib.cice_evp_hip_set_post_geometry(*[_dp(x) for x in a]), "(set_post_geometry)")

    def deformations(self) -> dict:
        names = ["divu", "shear", "vort", "rdg_conv", "rdg_shear"]
        out = {n: np.zeros(self.shape) for n in names}
        _check(self.lib, self.lib.cice_evp_hip_deformations(*[_dp(out[n]) for n in names]), "(deformations)")
        return out

    def dyn_finish(self, strocnxU, strocnyU) -> dict:
        sx = np.array(self._c(strocnxU), copy=True)
        sy = np.array(self._c(strocnyU), copy=True)
        _check(self.lib, self.lib.cice_evp_hip_dyn_finish(_dp(sx), _dp(sy)), "(dyn_finish)")
        return dict(strocnxU=sx, strocnyU=sy)

    def debug_cuload(self):
        self._need_testing("debug_cuload")
        a = np.zeros((2048, 8), dtype=np.int32)
        _check(self.lib, self.lib.cice_evp_hip_debug_cuload(_ip(a), C.c_int32(a.size)), "(debug_cuload)")
        return a

    def debug_prof(self, ntiles_max: int = 4096):
        self._need_testing("debug_prof")
        a = np.zeros((ntiles_max, 4, 8), dtype=np.uint64)
        _check(self.lib, self.lib.cice_evp_hip_debug_prof(a.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(ntiles_max)), "(debug_prof)")
        return a

    def debug_cgrid_prof(self, ntiles_max: int = 1 << 16):
        self._need_testing("debug_cgrid_prof")
        a = np.zeros((ntiles_max, 8), dtype=np.uint64)
        n = self.lib.cice_evp_hip_debug_cgrid_prof(a.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(ntiles_max))
        if n < 0:
            _check(self.lib, n, "(debug_cgrid_prof)")
        return a[:n]

    def time_kernels(self, nrep: int = 50) -> dict:
        t = np.zeros(3)
        _check(self.lib, self.lib.cice_evp_hip_time_kernels(C.c_int32(nrep), _dp(t)), "(time_kernels)")
        return dict(stencil_ms=t[0], halo_ms=t[1], stencil_period_ms=t[2])

    # -- RCCL ------------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _check(self.lib, self.lib.cice_evp_hip_comm_unique_id(buf), "(dyn_evp_hip_comm_unique_id)")
        return buf.raw

    def comm_init(self, uid: bytes):
        assert len(uid) == 128
        _check(self.lib, self.lib.cice_evp_hip_comm_init(C.c_char_p(uid)), "(dyn_evp_hip_comm_init)")

    # mailbox halo without RCCL: export -> host all-gather (rank order) -> import (collective)
    def halo_export(self) -> bytes:
        buf = C.create_string_buffer(HALO_BLOB)
        _check(self.lib, self.lib.cice_evp_hip_halo_export(buf), "(dyn_evp_hip_halo_export)")
        return buf.raw

    def halo_import(self, blobs):
        raw = b"".join(blobs)
        assert len(raw) == HALO_BLOB * len(blobs)
        _check(self.lib, self.lib.cice_evp_hip_halo_import(C.c_char_p(raw), C.c_int32(len(blobs))),
               "(dyn_evp_hip_halo_import)")

    # -- dyn_evp1d_finalize equivalent ---------------------------------------------------
    def finalize(self):
        if getattr(self, "_open", False):
            self.lib.cice_evp_hip_finalize()
            self._open = False

    def __del__(self):
        try:
            self.finalize()
        except Exception:
            pass


def march_plan(dims: Dims, own_max: int = 0, wrap_inside: bool = True, ext: int = 0) -> dict:
    """Host-only geometry + exchange lists of the marching path for `dims.rank` (CPU tests)."""
    lib = load_library(testing=True)
    geo = np.zeros(14, dtype=np.int32)
    _check(lib, lib.cice_evp_hip_march_plan(C.byref(dims), own_max, int(wrap_inside), ext, _ip(geo), None, None, None, None, None,
                                            None), "(march_plan)")
    npeer, ns, nr = int(geo[6]), int(geo[7]), int(geo[8])
    pr, pns, pnr = [np.zeros(max(npeer, 1), dtype=np.int32) for _ in range(3)]
    sp = np.zeros(max(ns, 1), dtype=np.int32)
    r1, r2 = [np.zeros(max(nr, 1), dtype=np.int32) for _ in range(2)]
    _check(lib, lib.cice_evp_hip_march_plan(C.byref(dims), own_max, int(wrap_inside), ext, _ip(geo), _ip(pr), _ip(pns), _ip(pnr),
                                            _ip(sp), _ip(r1), _ip(r2)), "(march_plan)")
    return dict(gx0=int(geo[0]), gy0=int(geo[1]), nxr=int(geo[2]), nyr=int(geo[3]), own=int(geo[4]), nstrips=int(geo[5]),
                wrapx=bool(geo[9]), ext=tuple(int(v) for v in geo[10:14]), peer_rank=pr[:npeer], peer_nsend=pns[:npeer],
                peer_nrecv=pnr[:npeer], send_pos=sp[:ns], recv_pos1=r1[:nr], recv_pos2=r2[:nr])


def halo_plan(dims: Dims) -> dict:
    """Host-only halo plan of `dims.rank` (no device needed): for CPU tests."""
    lib = load_library(testing=True)
    _check(lib, lib.cice_evp_hip_plan_build(C.byref(dims)), "(plan_build)")
    cnt = np.zeros(4, dtype=np.int32)
    lib.cice_evp_hip_halo_plan(_ip(cnt), None, None, None, None, None, None, None, None)
    nl, npeer, ns, nr = [int(v) for v in cnt]
    ld, ls, lg = [np.zeros(max(nl, 1), dtype=np.int32) for _ in range(3)]
    pr, pns, pnr = [np.zeros(max(npeer, 1), dtype=np.int32) for _ in range(3)]
    ss = np.zeros(max(ns, 1), dtype=np.int32)
    rd = np.zeros(max(nr, 1), dtype=np.int32)
    lib.cice_evp_hip_halo_plan(_ip(cnt), _ip(ld), _ip(ls), _ip(lg), _ip(pr), _ip(pns), _ip(pnr), _ip(ss), _ip(rd))
    c3 = np.zeros(3, dtype=np.int32)
    lib.cice_evp_hip_seam_plan(_ip(c3), None, None, None, None, None, None)
    npair, npole, nlate = [int(v) for v in c3]
    sa, sb = [np.zeros(max(npair, 1), dtype=np.int32) for _ in range(2)]
    sp = np.zeros(max(npole, 1), dtype=np.int32)
    td, ts, tg = [np.zeros(max(nlate, 1), dtype=np.int32) for _ in range(3)]
    lib.cice_evp_hip_seam_plan(_ip(c3), _ip(sa), _ip(sb), _ip(sp), _ip(td), _ip(ts), _ip(tg))
    c1 = np.zeros(1, dtype=np.int32)
    lib.cice_evp_hip_stress_plan(_ip(c1), None, None)
    nst = int(c1[0])
    std, sts = [np.zeros(max(nst, 1), dtype=np.int32) for _ in range(2)]
    lib.cice_evp_hip_stress_plan(_ip(c1), _ip(std), _ip(sts))
    lib.cice_evp_hip_center_plan(_ip(c1), None, None, None)
    ncen = int(c1[0])
    cd, cs, cv = [np.zeros(max(ncen, 1), dtype=np.int32) for _ in range(3)]
    center_remote = lib.cice_evp_hip_center_plan(_ip(c1), _ip(cd), _ip(cs), _ip(cv)) == 1
    sd = np.zeros(max(ns, 1), dtype=np.int32)
    rg = np.zeros(max(nr, 1), dtype=np.int32)
    lib.cice_evp_hip_peer_plan(_ip(sd), _ip(rg))
    rsg = np.zeros(max(nr, 1), dtype=np.int32)
    lib.cice_evp_hip_peer_signs(_ip(rsg))
    c2 = np.zeros(2, dtype=np.int32)
    stress_remote = lib.cice_evp_hip_seam_fin_plan(_ip(c2), None, None, None, None) == 1
    nfin, tail = int(c2[0]), int(c2[1])
    fd, fa, fb, fc = [np.zeros(max(nfin, 1), dtype=np.int32) for _ in range(4)]
    lib.cice_evp_hip_seam_fin_plan(_ip(c2), _ip(fd), _ip(fa), _ip(fb), _ip(fc))
    c4 = np.zeros((max(npeer, 1), 4), dtype=np.int32)
    lib.cice_evp_hip_fold_images_plan(_ip(c4), None, None, None)
    ssg = np.zeros(max(ns, 1), dtype=np.int32)
    fo = np.zeros((max(int(c4[:, 2].sum()), 1), 3), dtype=np.int32)
    fi = np.zeros((max(int(c4[:, 3].sum()), 1), 3), dtype=np.int32)
    lib.cice_evp_hip_fold_images_plan(_ip(c4), _ip(ssg), _ip(fo), _ip(fi))
    fl = np.zeros(5, dtype=np.int32)
    lib.cice_evp_hip_plan_flags(_ip(fl), 5)
    return dict(peer_counts4=c4[:npeer], send_sign=ssg[:ns], fimg_out=fo[:int(c4[:npeer, 2].sum())], fimg_in=fi[:int(c4[:npeer, 3].sum())],
                any_fold_exchange=bool(fl[0]), fold_rows=int(fl[1]), fin_dst=fd[:nfin], fin_a=fa[:nfin], fin_b=fb[:nfin], fin_coef=fc[:nfin], tail=tail, stress_remote=stress_remote, recv_sign=rsg[:nr],
                stress_dst=std[:nst], stress_src=sts[:nst], send_dst=sd[:ns], recv_gid=rg[:nr],
                center_dst=cd[:ncen], center_src=cs[:ncen], center_vsign=cv[:ncen], center_remote=center_remote,
                local_dst=ld[:nl], local_src=ls[:nl], local_sign=lg[:nl], peer_rank=pr[:npeer],
                peer_nsend=pns[:npeer], peer_nrecv=pnr[:npeer], send_src=ss[:ns], recv_dst=rd[:nr],
                seam_a=sa[:npair], seam_b=sb[:npair], seam_pole=sp[:npole],
                late_dst=td[:nlate], late_src=ts[:nlate], late_sign=tg[:nlate])
