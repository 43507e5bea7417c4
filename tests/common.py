"""Shared helpers of the test-suite (test infrastructure; may use the oracle)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

import oracle  # oracle/oracle.py  (CPU restatement -- the checker)

GOLDEN = Path(__file__).resolve().parent / "golden"
GOLDEN_CASES = sorted(p.stem for p in GOLDEN.glob("*.npz") if not p.stem.startswith(("cgrid_", "tript_", "cgtript_")))
TFOLD_CASES = sorted(p.stem for p in GOLDEN.glob("tript_*.npz"))         # ns_boundary_type = 'tripoleT' (B grid, the loop only)
CGRID_CASES = sorted(p.stem for p in GOLDEN.glob("cgrid_*.npz"))      # C-grid subcycle
CGRID_TFOLD_CASES = sorted(p.stem for p in GOLDEN.glob("cgtript_*.npz"))   # ... on ns_boundary_type = 'tripoleT'


class GoldenCase:
    """A fixture made by tests/golden/make_golden.py from the compiled reference."""

    def __init__(self, name):
        self.name = name
        z = np.load(GOLDEN / f"{name}.npz")
        self.d = {k: z[k] for k in z.files}
        self.ew, self.ns = str(self.d["ew"]), str(self.d["ns"])
        dims = [int(v) for v in self.d["dims"]]
        (self.nx_block, self.ny_block, self.nblocks, self.nghost, self.nx_global, self.ny_global,
         self.ndte, self.ncalls, _) = dims
        self.nsub_list = [int(v) for v in self.d["nsub_list"]]
        self.scal = self.d["scalars"]
        self.blk = np.asarray(self.d["blkinfo"]).reshape(self.nblocks, 8)

    # --- oracle-side objects ---
    def oracle_domain(self):
        return oracle.OracleDomain.from_dump(self.d, self.ew, self.ns)

    def oracle_params(self):
        return oracle.params_from_scalars(self.scal)

    def static(self):
        return {k: self.d[k] for k in oracle.STATIC_FIELDS}

    def inputs(self, icall=1):
        dyn = {k: self.d[f"in{icall:02d}_{k}"] for k in oracle.DYN_FIELDS}
        return dyn, self.d[f"in{icall:02d}_iceTmask"], self.d[f"in{icall:02d}_iceUmask"]

    # --- C-grid fixtures (cgrid_*): loop inputs captured after a preparation-only evp(), reference outputs ---
    def cgrid_inputs(self, icall=1):
        """(state, inputs, masks) of the C-grid subcycle loop; evp()'s work arrays start at zero."""
        pre = f"in{icall:02d}_"
        state = {k: self.d[pre + k] for k in oracle.C_FIELDS if k not in oracle.C_WORK}
        return state, {k: self.d[pre + k] for k in oracle.C_INPUTS}, {k: self.d[pre + k] for k in oracle.C_MASKS}

    def cgrid_static(self):
        return {k: self.d[k] for k in oracle.C_STATIC}

    def cgrid_expected(self, icall, nsub):
        return {k: self.d[f"o{icall:02d}n{nsub:04d}_{k}"] for k in oracle.C_FIELDS}

    # --- preparation phase on the C grid: what it reads (cp*), with the loop inputs (in*) as its expected products ---
    def cgrid_prep_static(self):
        st = self.cgrid_static()
        st.update({k: self.d[k] for k in oracle.C_PREP_MASKS + oracle.C_PREP_FCOR})
        return st

    def cgrid_prep_inputs(self, icall=1):
        """(T-grid fields, state + previous masks evp() is entered with, the loop inputs the previous call left)."""
        pre = f"cp{icall:02d}_"
        t = {k: self.d[pre + k] for k in oracle.PREP_T}
        state = {k: self.d[pre + k] for k in oracle.C_FIELDS[:12]}
        z = np.zeros_like(self.d["tarea"])
        last = f"o{icall - 1:02d}n{self.nsub_list[-1]:04d}_"
        for k in ("taubxE", "taubyN"):
            state[k] = z if icall == 1 else self.d[last + k]
        for k in ("iceUmask", "iceEmask", "iceNmask"):
            state[k] = self.d[pre + k]
        prev = None if icall == 1 else {k: self.d[f"in{icall - 1:02d}_{k}"] for k in oracle.C_INPUTS}
        return t, state, prev

    # --- preparation phase of evp() (f-2): its inputs, parameters and captured products ---
    def prep_static(self):
        return {k: self.d[k] for k in ("tmask", "umask", "hm", "tarea", "uarea", "fcor_blk")}

    def prep_inputs(self, icall=1):
        """(T-grid fields, state evp() is entered with)."""
        t = {k: self.d[f"pr{icall:02d}_{k}"] for k in oracle.PREP_T}
        state = {k: self.d[f"pr{icall:02d}_{k}"] for k in oracle.DYN_FIELDS[:12] + ["uvel", "vvel", "iceUmask"]}
        z = np.zeros_like(self.d["tarea"])
        # strintxU/strocnxU enter evp() as left by the previous call; zero before the first
        for k in ("strintxU", "strintyU"):
            state[k] = z if icall == 1 else self.d[f"o{icall - 1:02d}n{self.nsub_list[-1]:04d}_{k}"]
        for k in ("strocnxU", "strocnyU"):
            state[k] = z if icall == 1 else self.d[f"o{icall - 1:02d}n{self.nsub_list[-1]:04d}_{k}"]
        return t, state

    def prep_scal_dict(self):
        s = self.scal
        return dict(dt=s[28], rhoi=s[17], rhos=s[18], gravit=s[19], dyn_area_min=s[20], dyn_mass_min=s[21],
                    cosw=s[10], sinw=s[11], ssh_coupled=int(s[22]))

    def expected(self, icall, nsub):
        return {k: self.d[f"o{icall:02d}n{nsub:04d}_{k}"] for k in oracle.OUT_FIELDS}

    # --- product-side objects (C ABI structs) ---
    def scal_dict(self):
        s = self.scal
        return dict(ndte=self.ndte, arlx1i=s[0], denom1=s[1], brlx=s[2], revp=s[3], e_factor=s[4],
                    epp2i=s[5], capping=s[6], Ktens=s[7], deltaminEVP=s[8], u0=s[9], cosw=s[10],
                    sinw=s[11], rhow=s[12])

    def hip_dims(self):
        from cice_amd import evp
        b = self.blk
        loc = [np.ascontiguousarray(b[:, k], dtype=np.int32) for k in (0, 1, 2, 3, 6, 7)]
        d = evp.Dims(self.nx_block, self.ny_block, self.nblocks, self.nblocks, self.nghost,
                     self.nx_global, self.ny_global, evp.BND[self.ew], evp.BND[self.ns], 0, 1,
                     *[a.ctypes.data_as(evp._i32p) for a in loc], 0, None, None, None, None, None, None)
        return d, loc


def tfold_untouched(c: "GoldenCase"):
    """Cells of a tripoleT fixture that evp()'s ice_HaloUpdate_stress calls after the loop do not rewrite: everything but
    the top physical row and the ghost row above it (ice_boundary.F90:7700-7790 with the T-fold offsets) -- the stress
    arrays of whole-evp() fixtures are compared there; the velocities everywhere."""
    keep = np.ones((c.nblocks, c.ny_block, c.nx_block), dtype=bool)
    for b in range(c.nblocks):
        jlo, jhi, jg0 = int(c.blk[b, 2]), int(c.blk[b, 3]), int(c.blk[b, 7])
        if jg0 + (jhi - jlo) == c.ny_global:
            keep[b, jhi - 1:, :] = False
    return keep


def bits_equal(a, b) -> bool:
    """Bit-for-bit equality: fp64 arrays are compared as their 64-bit patterns, so +0.0 and -0.0 differ and a NaN equals
    only the same NaN (np.array_equal compares values: it lets a sign-of-zero difference through).  Other dtypes: values."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype == np.float64 and b.dtype == np.float64:
        return bool(np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64)))
    return bool(np.array_equal(a, b))


def assert_bitwise(got: dict, want: dict, what=""):
    bad = []
    for k, w in want.items():
        g = got[k]
        if not bits_equal(g, w):
            gb, wb = (np.ascontiguousarray(x, dtype=np.float64).view(np.uint64) for x in (g, w))
            ne = np.argwhere(gb != wb)
            nz = int(((g == w) & (gb != wb)).sum())
            bad.append(f"{k}: {len(ne)} cells differ ({nz} of them only in the sign of zero), max|d|={np.nanmax(np.abs(g - w)):.3e}, "
                       f"first at {ne[0].tolist()}")
    assert not bad, f"{what} not bit-identical:\n  " + "\n  ".join(bad)


def max_rel_err(got: dict, want: dict, keys):
    """max over fields of max|got-want| / max|want| (field-wise scale)."""
    worst = 0.0
    for k in keys:
        scale = max(np.abs(want[k]).max(), 1e-300)
        worst = max(worst, np.abs(got[k] - want[k]).max() / scale)
    return worst


# The default GPU suite has to fit the driver's time limit with room to spare (round-5 review: keep it under ~8 minutes).
# Cases that repeat a path another case already pins (another cut of the same grid, one more seed) run when
# EVP_GPU_SUITE=full -- as the wide sweeps do through their *_SWEEP_SEEDS switches; their logs go under profiles/.
FULL_SUITE = __import__("os").environ.get("EVP_GPU_SUITE", "") == "full"


def wide(cases):
    """`cases` in the full run, nothing in the default one (use inside a parametrize list: [...] + wide([...]))."""
    return list(cases) if FULL_SUITE else []
