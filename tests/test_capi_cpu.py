"""CPU: the C-ABI library loads and exports every symbol include/*.h declares, and
its host-side halo plan reproduces the ghost-cell semantics of the oracle.  No
compute entry point is called (no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import oracle
from cice_amd import decomp, evp
from common import bits_equal

ROOT = Path(__file__).resolve().parents[1]


def declared_functions(header="cice_evp_hip.h"):
    txt = (ROOT / "include" / header).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|void\s*\*)\s*(cice_evp_hip_\w+)\s*\(", txt)))


def exported(lib_path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib_path)], capture_output=True, text=True, check=True).stdout
    return sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln)


def test_library_exports_every_declared_symbol():
    """The product library exports exactly what include/cice_evp_hip.h declares -- no test hook, no introspection entry
    point, no C++ symbol; the test build adds exactly what include/cice_evp_hip_testing.h declares."""
    lib = evp.load_library()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cice_evp_hip.h but not exported"
    assert sorted(evp.EXPORTS) == names
    assert lib.cice_evp_hip_abi_version() == 1
    assert exported(evp.LIB_PATH) == names
    extra = declared_functions("cice_evp_hip_testing.h")
    assert sorted(evp.TEST_EXPORTS) == extra and not set(extra) & set(names)
    tlib = evp.load_library(testing=True)
    for n in names + extra:
        assert hasattr(tlib, n), f"{n} missing from the test build"
    assert exported(evp.LIB_TESTING_PATH) == sorted(names + extra)
    for n in extra:
        assert not hasattr(lib, n), f"{n} is a test-build entry point but the product library exports it"


def test_product_library_ignores_the_test_builds_switches():
    """Strings of the experiment / fault-injection switches are read through env_test(), which is a constant NULL in the
    product library: no getenv of theirs can be reached (the switch names may still appear as literals)."""
    import subprocess
    src = "".join(p.read_text() for p in (ROOT / "cice_amd" / "csrc").glob("*.cpp")) + \
          "".join(p.read_text() for p in (ROOT / "cice_amd" / "csrc").glob("*.hip"))
    for k in evp.TEST_ENV:
        assert f'env("{k}")' not in src and f'getenv("{k}")' not in src, k
        assert f'env_test("{k}")' in src or f'fault_hook("{k}")' in src, f"{k} listed in evp.TEST_ENV but not read by the test build"
    kept = set(re.findall(r'(?<![_a-z])env\("(CICE_EVP_HIP_\w+)"\)', src))
    assert not kept & set(evp.TEST_ENV)
    assert kept <= {"CICE_EVP_HIP_" + k for k in ("DEVICE", "VERBOSE", "HALO", "HALO_TIMEOUT_MS", "RESIDENT", "MARCH", "MARCH_OVERLAP",
                                                  "CGRID_ONE", "CGRID_RESIDENT")}, kept
    # ... and the product library does not even carry the other names (env_test is a macro that drops the literal there)
    import subprocess
    names = set(re.findall(r"CICE_EVP_HIP_[A-Z0-9_]+", subprocess.run(["strings", str(evp.LIB_PATH)], capture_output=True, text=True).stdout))
    assert len(names) <= 15, sorted(names)
    assert kept <= names | {"CICE_EVP_HIP_CGRID_RESIDENT"}


def test_struct_layout_matches_header():
    # sizes implied by the header on LP64: dims = 11 int32 (+pad) + 6 ptr + int32 (+pad) + 6 ptr
    assert C.sizeof(evp.Dims) == 48 + 6 * 8 + 8 + 6 * 8
    assert C.sizeof(evp.Params) == 8 + 13 * 8


def test_compute_entry_points_fail_loudly_without_init():
    lib = evp.load_library()
    assert lib.cice_evp_hip_subcycle(C.c_int32(1)) != 0
    buf = C.create_string_buffer(256)
    lib.cice_evp_hip_last_error(buf, 256)
    assert b"upload" in buf.value or b"initialised" in buf.value


def apply_plan_single_rank(plan, a):
    """What halo_uv() does on one GPU: local copies, tripole seam (pairs, poles), late copies."""
    flat = a.reshape(-1)
    src = plan["local_src"]
    val = np.where(src >= 0, plan["local_sign"] * flat[np.maximum(src, 0)], 0.0)
    flat[plan["local_dst"]] = val
    if len(plan["seam_a"]):
        xa, xb = flat[plan["seam_a"]].copy(), flat[plan["seam_b"]].copy()
        xavg = 0.5 * (xa + (-1.0) * xb)
        flat[plan["seam_a"]] = xavg
        flat[plan["seam_b"]] = -1.0 * xavg
    if len(plan["seam_pole"]):
        flat[plan["seam_pole"]] = -1.0 * flat[plan["seam_pole"]]
    if len(plan["late_dst"]):
        flat[plan["late_dst"]] = plan["late_sign"] * flat[plan["late_src"]]
    return a


@pytest.mark.parametrize("ew,ns,bx,by", [("cyclic", "closed", 7, 5), ("closed", "closed", 20, 6),
                                         ("cyclic", "cyclic", 8, 9), ("cyclic", "closed", 20, 18),
                                         ("cyclic", "tripole", 20, 18), ("cyclic", "tripole", 5, 6),
                                         ("cyclic", "tripole", 7, 18),
                                         # T-fold: the top physical row is a destination too (image of row NY-1)
                                         ("cyclic", "tripoleT", 20, 18), ("cyclic", "tripoleT", 5, 6),
                                         ("cyclic", "tripoleT", 7, 18)])
def test_halo_plan_matches_oracle_semantics(ew, ns, bx, by):
    dc = decomp.Decomp(20, 18, bx, by, ew, ns, 1)
    d, keep = evp.make_dims(dc, 0)
    plan = evp.halo_plan(d)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), 20, 18, ew, ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    rng = np.random.default_rng(7)
    a = rng.standard_normal(dc.shape(0))
    want = oracle.halo_update(dom, a.copy(), "NEcorner", "vector")
    got = apply_plan_single_rank(plan, a.copy())
    assert bits_equal(got, want)
    # every ghost cell that has a source appears exactly once
    assert len(set(plan["local_dst"].tolist())) == len(plan["local_dst"])


@pytest.mark.parametrize("bx,by", [(20, 18), (5, 6), (7, 18), (10, 9)])
def test_stress_symmetrisation_lists_match_oracle(bx, by):
    """The lists behind cice_evp_hip_stress_halo (ice_HaloUpdate_stress, f-3) applied with numpy
    == the oracle's restatement (itself pinned by the tripole fixtures of the whole evp())."""
    dc = decomp.Decomp(20, 18, bx, by, "cyclic", "tripole", 1)
    d, keep = evp.make_dims(dc, 0)
    plan = evp.halo_plan(d)
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), 20, 18, "cyclic", "tripole",
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    rng = np.random.default_rng(11)
    out = {k: rng.standard_normal(dc.shape(0)) for k in evp.FIELDS[:12]}
    want = oracle.tripole_stress_sym(dom, {k: v.copy() for k, v in out.items()})
    names = evp.FIELDS[:12]
    got = {k: v.copy() for k, v in out.items()}
    dst, src = plan["stress_dst"], plan["stress_src"]
    assert len(dst) > 0 and (src >= 0).all()
    for k, name in enumerate(names):
        got[name].reshape(-1)[dst] = out[names[k ^ 2]].reshape(-1)[src]     # partner: 1<->3, 2<->4
    for name in names:
        assert bits_equal(got[name], want[name]), name
    # no list on a grid without a tripole seam
    d2, keep2 = evp.make_dims(decomp.Decomp(20, 18, bx, by, "cyclic", "closed", 1), 0)
    assert len(evp.halo_plan(d2)["stress_dst"]) == 0


@pytest.mark.parametrize("ew,ns,bx,by", [("cyclic", "closed", 7, 5), ("closed", "closed", 20, 6), ("cyclic", "cyclic", 8, 9),
                                         ("cyclic", "tripole", 20, 18), ("cyclic", "tripole", 5, 6)])
@pytest.mark.parametrize("vector", [False, True])
def test_center_field_halo_lists_match_oracle(ew, ns, bx, by, vector):
    """Ghost-cell lists of cell-centre fields (the T-grid halos of evp()'s preparation phase)
    applied with numpy == the oracle's halo update for field_loc_center, scalar and vector kinds
    (the oracle's is pinned through the prep fixtures, tripole included)."""
    dc = decomp.Decomp(20, 18, bx, by, ew, ns, 1)
    d, keep = evp.make_dims(dc, 0)
    plan = evp.halo_plan(d)
    assert not plan["center_remote"]
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), 20, 18, ew, ns,
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    a = np.random.default_rng(3).standard_normal(dc.shape(0))
    want = oracle.halo_update(dom, a.copy(), "center", "vector" if vector else "scalar")
    got = a.copy()
    flat = got.reshape(-1)
    src = plan["center_src"]
    sgn = plan["center_vsign"] if vector else np.ones_like(plan["center_vsign"])
    flat[plan["center_dst"]] = np.where(src >= 0, sgn * a.reshape(-1)[np.maximum(src, 0)], 0.0)
    assert bits_equal(got, want)


def apply_fin(plan, flat):
    """The general seam step (any rank layout): every entry from RAW values, then all stores."""
    a, b, c = plan["fin_a"], plan["fin_b"], plan["fin_coef"].astype(np.float64)
    pair = b >= 0
    res = np.where(pair, c * (0.5 * (flat[a] + (-1.0) * flat[np.maximum(b, 0)])), c * flat[a])
    flat[plan["fin_dst"]] = res


@pytest.mark.parametrize("bx,by", [(20, 18), (5, 6), (7, 18), (10, 9)])
def test_general_seam_lists_equal_single_rank_form(bx, by):
    """fin lists (the seam step for any rank layout) on one rank == local copies + pairs + poles + late copies
    == the oracle's tripole halo update."""
    dc = decomp.Decomp(20, 18, bx, by, "cyclic", "tripole", 1)
    d, keep = evp.make_dims(dc, 0)
    plan = evp.halo_plan(d)
    assert plan["tail"] == 0 and not plan["stress_remote"] and len(plan["fin_dst"]) >= 20
    blks = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(blks), 20, 18, "cyclic", "tripole",
                              [b.ilo for b in blks], [b.ihi for b in blks], [b.jlo for b in blks],
                              [b.jhi for b in blks], [b.gi0 for b in blks], [b.gj0 for b in blks])
    a = np.random.default_rng(5).standard_normal(dc.shape(0))
    want = oracle.halo_update(dom, a.copy(), "NEcorner", "vector")
    got = a.copy()
    flat = got.reshape(-1)
    src = plan["local_src"]
    flat[plan["local_dst"]] = np.where(src >= 0, plan["local_sign"] * flat[np.maximum(src, 0)], 0.0)
    apply_fin(plan, flat)
    assert bits_equal(got, want)


def test_tripole_plan_with_the_seam_split_across_ranks():
    """px = 2: seam pairs and the east-west neighbours of the seam row live on different ranks -- the plan now
    carries staging slots for the raw partner values instead of refusing; y slabs keep everything local."""
    dc = decomp.Decomp(20, 18, 10, 9, "cyclic", "tripole", 2, (2, 1))   # seam row cut in x
    for r in range(2):
        d, keep = evp.make_dims(dc, r)
        plan = evp.halo_plan(d)
        assert plan["tail"] > 0 and plan["stress_remote"] and len(plan["seam_a"]) == 0
        n_local = dc.nx_block * dc.ny_block * len(dc.local_blocks(r))
        assert (plan["recv_dst"] >= n_local).sum() == plan["tail"]
    dc = decomp.Decomp(20, 18, 20, 9, "cyclic", "tripole", 2, (1, 2))    # y slabs: seam on one rank
    for r in range(2):
        d, keep = evp.make_dims(dc, r)
        plan = evp.halo_plan(d)
        assert (len(plan["seam_a"]) > 0) == (r == 1) and plan["tail"] == 0 and not plan["stress_remote"]


@pytest.mark.parametrize("nranks,shape", [(4, (2, 2)), (2, (2, 1)), (2, (1, 2)), (8, (4, 2))])
def test_fold_exchange_flag_is_the_same_on_every_rank(nranks, shape):
    """cice_evp_hip_halo_mask is collective: whether the in-loop exchange may be masked must come out alike on all
    ranks.  On a tripole grid split in x AND y only the fold-row ranks have fold-crossing / staging entries of their
    own; the flag is a property of the whole layout (ice_boundary.F90:979,1022: fold messages are never masked)."""
    dc = decomp.Decomp(24, 20, 24 // shape[0], 20 // shape[1], "cyclic", "tripole", nranks, shape)
    flags, own = [], []
    for r in range(nranks):
        d, keep = evp.make_dims(dc, r)
        plan = evp.halo_plan(d)
        flags.append(plan["any_fold_exchange"])
        own.append(plan["tail"] > 0 or bool((plan["recv_sign"] < 0).any()))
    assert len(set(flags)) == 1
    assert flags[0] == any(own)
    if shape == (2, 2):
        assert not all(own) and flags[0]          # the case the per-rank decision got wrong
    dc = decomp.Decomp(24, 20, 12, 10, "cyclic", "closed", 4, (2, 2))
    for r in range(4):
        d, keep = evp.make_dims(dc, r)
        assert not evp.halo_plan(d)["any_fold_exchange"]


def test_halo_plan_rejects_bad_geometry():
    dc = decomp.Decomp(20, 18, 10, 9, "cyclic", "closed", 1)
    d, keep = evp.make_dims(dc, 0)
    d.nghost = 2
    with pytest.raises(evp.EvpHipError):
        evp.halo_plan(d)


def test_decomp_fold_scatter_matches_the_halo_rules():
    """decomp.scatter(fold=...) -- used to lay out synthetic tripole workloads -- fills the ghost row beyond the fold
    like ice_HaloUpdate does for locations whose points are not ON the fold (centre, E face), scalars and vectors
    (checked against the oracle's halo update, itself pinned to the reference's tripole fixtures)."""
    from cice_amd import decomp
    nx, ny = 24, 18
    dc = decomp.Decomp(nx, ny, 8, 6, "cyclic", "tripole", 1)
    ob = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(ob), nx, ny, "cyclic", "tripole", [b.ilo for b in ob],
                              [b.ihi for b in ob], [b.jlo for b in ob], [b.jhi for b in ob], [b.gi0 for b in ob],
                              [b.gj0 for b in ob])
    g = np.random.default_rng(1).standard_normal((ny, nx))
    for loc in ("center", "Eface"):
        for kind, sg in (("scalar", 1.0), ("vector", -1.0)):
            a = dc.scatter(g, 0, fill=0.0, fold=(loc, sg))
            w = oracle.halo_update(dom, np.ascontiguousarray(dc.scatter(g, 0, fill=0.0)), loc, kind)
            assert bits_equal(a, w), (loc, kind)


@pytest.mark.parametrize("ns", ["tripole", "tripoleT"])
@pytest.mark.parametrize("nx,ny,bx,by", [(24, 18, 24, 18), (28, 20, 14, 10), (32, 24, 8, 8), (26, 14, 10, 5)])
def test_cgrid_fold_lists_equal_the_halo_update_at_the_fold(nx, ny, bx, by, ns):
    """C grid on tripole grids: the host-built fold lists (cice_evp_hip_cgrid_fold_plan) applied to arbitrary block
    arrays give, on every cell of the fold row and of the ghost row beyond it, exactly what ice_HaloUpdate gives
    there -- all four field locations, scalar and vector kinds, 1 ... 12 blocks incl. padded ones (against the
    oracle's halo update, pinned to the reference's tripole fixtures for every location).  tripoleT (end of round 4): the
    T-fold lists -- every location rewrites the top physical row there, centre and E-face fields are the ones ON the fold --
    against the oracle's T-fold update, pinned to the reference's tripoleT fixtures (B-grid loop and preparation, C-grid loop)."""
    from cice_amd import decomp
    dc = decomp.Decomp(nx, ny, bx, by, "cyclic", ns, 1)
    d, keep = evp.make_dims(dc, 0)
    ob = dc.local_blocks(0)
    dom = oracle.OracleDomain(dc.nx_block, dc.ny_block, len(ob), nx, ny, "cyclic", ns, [b.ilo for b in ob],
                              [b.ihi for b in ob], [b.jlo for b in ob], [b.jhi for b in ob], [b.gi0 for b in ob],
                              [b.gj0 for b in ob])
    rng = np.random.default_rng(nx * 100 + ny)
    for loc in ("center", "NEcorner", "Eface", "Nface"):
        L = evp.cgrid_fold_plan(d, loc)
        assert len(L["dst"]) > 0 and len(set(L["dst"].tolist())) == len(L["dst"])        # every cell once
        on_fold = (L["b"] >= 0).sum()
        assert (on_fold > 0) == (loc in (("NEcorner", "Nface") if ns == "tripole" else ("center", "Eface")))
        for kind, isign in (("scalar", 1.0), ("vector", -1.0)):
            x = rng.standard_normal((len(ob), dc.ny_block, dc.nx_block))
            want = oracle.halo_update(dom, x.copy(), loc, kind).reshape(-1)
            flat = x.reshape(-1)
            s = np.where(L["flip"] != 0, isign, 1.0)
            xa = np.where(L["a"] >= 0, flat[np.maximum(L["a"], 0)], 0.0)
            xb = np.where(L["b"] >= 0, flat[np.maximum(L["b"], 0)], 0.0)
            val = np.where(L["b"] != -1, s * (0.5 * (xa + isign * xb)), s * xa)
            assert bits_equal(val, want[L["dst"]]), (loc, kind, int((val != want[L["dst"]]).sum()))


@pytest.mark.parametrize("nx,ny,bx,by", [(24, 18, 24, 18), (28, 20, 14, 10), (40, 30, 20, 5), (72, 48, 36, 48), (60, 31, 20, 31),
                                         (30, 40, 30, 14)])
def test_cgrid_fold_window_table_mirrors_in_source_orientation(nx, ny, bx, by):
    """The resident C-grid kernel's window table on a tripole (u-fold) grid (halo_plan.cpp: build_fold_window_table): the rows up
    to the fold row name the global cell of the position, every interior cell is owned by exactly one window (owned rows end at
    the limit in tiles[3]); in the windows at the fold the tile rows tf+1 .. tf+3 name the global rows NY-2 .. NY at columns that
    INCREASE with tx, placed so that the normal fold-row position tx faces the source of an E-face / corner field at column
    15 - tx (global column NX - ig) and of a centre / N-face field at 16 - tx (NX - ig + 1) -- ice_boundary.F90:1626-1722."""
    from cice_amd import decomp
    dc = decomp.Decomp(nx, ny, bx, by, "cyclic", "tripole", 1)
    d, keep = evp.make_dims(dc, 0)
    ob = dc.local_blocks(0)
    nxb, nyb = dc.nx_block, dc.ny_block
    plane = nxb * nyb
    home = {}
    for k, b in enumerate(ob):
        for j in range(b.jlo, b.jhi + 1):
            for i in range(b.ilo, b.ihi + 1):
                home[(b.gi0 + i - b.ilo, b.gj0 + j - b.jlo)] = k * plane + (j - 1) * nxb + (i - 1)
    P = evp.cgrid_window_plan(d, 16, 16, 2)
    owned = np.zeros(len(ob) * plane, dtype=int)
    nfold = 0
    wrap = lambda g: (g - 1) % nx + 1
    for (k, i0, j0, flags), tab in zip(P["tiles"], P["tab"]):
        b = ob[k]
        fw, tf, jmax = flags & 1, (flags >> 8) & 255, flags >> 16
        nfold += fw
        assert jmax <= b.jhi and (not fw or (jmax == b.jhi and b.gj0 + b.jhi - b.jlo == ny and tf == 2 + b.jhi - j0 and tf <= 12))
        for ty in range(17):
            for tx in range(17):
                i, j = i0 - 2 + tx, j0 - 2 + ty
                gi, gj = wrap(b.gi0 + i - b.ilo), b.gj0 + j - b.jlo
                e = int(tab[ty, tx])
                if not fw or ty <= tf:
                    if 1 <= gj <= ny:
                        assert e == home[(gi, gj)], (k, i0, j0, tx, ty)
                    if 2 <= tx <= 14 and 2 <= ty <= 14 and i <= b.ihi and j <= jmax:
                        owned[e] += 1
                elif ty <= tf + 3:
                    gjm = ny - (tf + 3 - ty)
                    kk, r = divmod(e, plane)
                    bb = ob[kk]
                    jj, ii = r // nxb + 1, r % nxb + 1
                    assert bb.ilo <= ii <= bb.ihi and bb.jlo <= jj <= bb.jhi and bb.gj0 + jj - bb.jlo == gjm
                    gim = bb.gi0 + ii - bb.ilo
                    if tx <= 15:        # the normal position 15 - tx: E-face / corner partner NX - ig (NX for ig = NX)
                        g = wrap(b.gi0 + (i0 - 2 + 15 - tx) - b.ilo)
                        assert gim == wrap(nx - g), (k, i0, j0, tx, ty, gim, g)
                    g = wrap(b.gi0 + (i0 - 2 + 16 - tx) - b.ilo)   # the normal position 16 - tx: centre / N-face partner NX - ig + 1
                    assert gim == wrap(nx - g + 1), (k, i0, j0, tx, ty, gim, g)
                else:
                    assert e < 0
    interior = np.zeros(len(ob) * plane, dtype=int)
    interior[list(home.values())] = 1
    assert bits_equal(owned, interior)
    assert nfold == sum(-(-(b.ihi - b.ilo + 1) // 13) for b in ob if b.gj0 + b.jhi - b.jlo == ny)


@pytest.mark.parametrize("nx,ny,bx,by,ew,ns", [(24, 18, 24, 18, "cyclic", "closed"), (28, 20, 14, 10, "cyclic", "closed"),
                                               (26, 22, 10, 12, "cyclic", "cyclic"), (40, 30, 20, 10, "closed", "closed"),
                                               (70, 37, 24, 13, "cyclic", "closed"), (9, 7, 4, 3, "cyclic", "cyclic")])
@pytest.mark.parametrize("ox,oy,extra", [(32, 8, 0), (64, 16, 0), (8, 6, 0), (16, 16, 1)])
def test_cgrid_window_table_names_the_source_cell_of_every_position(nx, ny, bx, by, ew, ns, ox, oy, extra):
    """C grid, one launch per subcycle: the host-built window table (cice_evp_hip_cgrid_window_plan; what the kernel reads)
    against the decomposition's global numbering.  A position (tx, ty) of a window is the cell (i0-2+tx, j0-2+ty) of its
    block's numbering -- possibly a ghost cell, possibly beyond the block's array; the table must name the interior cell
    that holds that GLOBAL cell (through periodic boundaries and into other blocks, any number of steps away), or a ghost
    cell outside the domain where the boundary is closed.  Every interior cell is owned by exactly one window, and the
    'regular' mark (the kernel then computes the index instead of loading it) is set exactly where the table is the identity."""
    from cice_amd import decomp
    dc = decomp.Decomp(nx, ny, bx, by, ew, ns, 1)
    d, keep = evp.make_dims(dc, 0)
    ob = dc.local_blocks(0)
    nxb, nyb = dc.nx_block, dc.ny_block
    plane = nxb * nyb
    home = {}                                   # global (gi, gj) -> flat index of the interior cell that holds it
    for k, b in enumerate(ob):
        for j in range(b.jlo, b.jhi + 1):
            for i in range(b.ilo, b.ihi + 1):
                home[(b.gi0 + i - b.ilo, b.gj0 + j - b.jlo)] = k * plane + (j - 1) * nxb + (i - 1)
    assert len(home) == nx * ny
    # (extra = 1: the on-chip resident kernel's table, one more row and column of positions per window, same owned range)
    P = evp.cgrid_window_plan(d, ox, oy, extra)
    owned = np.zeros(len(ob) * plane, dtype=int)
    for (k, i0, j0, regular), tab in zip(P["tiles"], P["tab"]):
        b = ob[k]
        ident = True
        for ty in range(oy + extra):
            for tx in range(ox + extra):
                i, j = i0 - 2 + tx, j0 - 2 + ty
                gi, gj = b.gi0 + i - b.ilo, b.gj0 + j - b.jlo
                if ew == "cyclic":
                    gi = (gi - 1) % nx + 1
                if ns == "cyclic":
                    gj = (gj - 1) % ny + 1
                e = int(tab[ty, tx])
                inside = 1 <= gi <= nx and 1 <= gj <= ny
                direct = k * plane + (j - 1) * nxb + (i - 1) if (1 <= i <= nxb and 1 <= j <= nyb) else None
                ident = ident and direct is not None and e == direct
                if inside:
                    assert e == home[(gi, gj)], (k, i0, j0, tx, ty, e, home[(gi, gj)])
                else:
                    assert e < 0, (k, i0, j0, tx, ty, e)             # beyond a closed boundary: a ghost cell, read as it is
                    c = -1 - e
                    kb, r = divmod(c, plane)
                    jc, ic = r // nxb + 1, r % nxb + 1
                    bb = ob[kb]
                    assert not (bb.ilo <= ic <= bb.ihi and bb.jlo <= jc <= bb.jhi)
                    gic, gjc = bb.gi0 + ic - bb.ilo, bb.gj0 + jc - bb.jlo
                    assert (ew != "cyclic" and not 1 <= gic <= nx) or (ns != "cyclic" and not 1 <= gjc <= ny), (c, gic, gjc)
                if 2 <= tx <= ox - 2 and 2 <= ty <= oy - 2 and i <= b.ihi and j <= b.jhi:
                    owned[e] += 1
        assert bool(regular) == ident, (k, i0, j0, regular, ident)
    interior = np.zeros(len(ob) * plane, dtype=int)
    interior[list(home.values())] = 1
    assert bits_equal(owned, interior)      # every interior cell in exactly one window, nothing else owned


def test_tracked_pmc_summary_feeds_the_bench_line():
    """bench.py divides counters of the newest profiles/r*_pmc_summary.json by the live kernel time: the summary must
    hold the entry of every kernel the roofline block quotes (a summary reduced from a partial profile directory
    silently turned `roofline.frac` into null once)."""
    import json
    from pathlib import Path
    files = sorted((Path(__file__).resolve().parents[1] / "profiles").glob("r*_pmc_summary.json"))
    assert files, "no PMC summary tracked"
    k = json.loads(files[-1].read_text())["kernels"]
    for key in ("gx1res", "gx1str", "s01str", "s01march"):
        assert key in k, f"{files[-1].name}: no entry for {key}"
        assert k[key].get("hbm_bytes_per_launch") and k[key].get("valu_busy_simd_cycles_per_launch"), key
        assert k[key]["kernel_trace"]["avg_us"] > 0
    # C grid: the kernels the bench line's cgrid block runs by default (round 5: the on-chip resident kernel on gx1, one launch
    # per subcycle on 3600 x 2400)
    for key in ("cgx1res", "cgs01one"):
        assert k[key]["per_subcycle"]["hbm_bytes"], key


@pytest.mark.parametrize("nx,ny,bx,by,ew,ns", [
    (54, 26, 27, 26, "cyclic", "closed"), (40, 27, 40, 27, "cyclic", "closed"),      # the round-5 advisor's two: 8 and 7 one-way before
    (320, 384, 320, 384, "cyclic", "closed"), (100, 116, 50, 58, "cyclic", "closed"), (72, 40, 36, 20, "cyclic", "tripole"),
    (320, 384, 320, 384, "cyclic", "tripole"), (320, 384, 80, 96, "cyclic", "tripole")])
def test_cgrid_resident_window_handoffs_are_safe(nx, ny, bx, by, ew, ns):
    """Round-5 advice: the resident C-grid kernel's exact-tag record protocol with two slots per cell assumed a window is never more
    than one subcycle ahead of a window that reads it -- true only if every hand-off is mutual.  A narrow window at a block's edge
    (one or two owned columns) used to poll every non-owned position of its 17 x 17 tile, far into windows that do not poll it back;
    fold windows whose mirror images do not line up read one another one-way too.  Now (a) the ring is bounded to what the owned
    cells can reach (halo_plan.h: cgres_in_reach), (b) there are FOUR slots per cell, and (c) the library works the graph out from
    the kernel's own rule and uses the kernel only if every window that is read is held back by its reader through a chain of at
    most three hand-offs (cice_evp_hip_cgrid_window_deps: unsafe == 0)."""
    from cice_amd import decomp
    dc = decomp.Decomp(nx, ny, bx, by, ew, ns, 1)
    d, keep = evp.make_dims(dc, 0)
    g = evp.cgrid_window_deps(d)
    assert g["windows"] > 0 and g["edges"] > 0 and g["unsafe"] == 0, g
    if ns != "tripole":
        assert g["oneway"] == 0, g


@pytest.mark.parametrize("nx,ny,bx,by,ew,lo0,seg", [(3600, 2400, 3600, 2400, "cyclic", 3, 0), (3600, 2400, 1800, 1200, "cyclic", 2, 0),
                                                    (720, 540, 720, 540, "cyclic", 3, 0), (400, 64, 200, 32, "cyclic", 3, 5),
                                                    (190, 50, 190, 50, "closed", 2, 3), (331, 47, 331, 47, "cyclic", 3, 1),
                                                    (140, 90, 70, 90, "cyclic", 3, 0), (1000, 37, 250, 37, "closed", 2, 64)])
def test_cgrid_marched_kernel_plan_covers_every_cell_once(nx, ny, bx, by, ew, lo0, seg):
    """How the one-launch C-grid schedule of large domains shares a rank between the marched kernel (cg_strip) and the windowed one
    (halo_plan.h: strip_zones / strip_items / strip_windows): every interior cell of every block is owned exactly once -- by one lane
    of one work item, or by one window the windowed kernel keeps --, the items own exactly the cells of the windows they replace, every
    position an item computes is an interior cell of its block and every row / column it loads lies inside the block's array."""
    from cice_amd import decomp
    dc = decomp.Decomp(nx, ny, bx, by, ew, "closed", 1)
    d, keep = evp.make_dims(dc, 0)
    ex, ey = 32, 8
    pl = evp.cgrid_strip_plan(d, ex=ex, ey=ey, lo0=lo0, slots=2048, seg_min=8, seg=seg)
    blks = dc.local_blocks(0)
    own = np.zeros((len(blks), dc.ny_block + 1, dc.nx_block + 1), dtype=np.int32)      # 1-based
    for b, c, ja, jb, lo, hi in pl["items"]:
        B = blks[b]
        assert lo0 <= lo <= hi <= 61 and ja <= jb
        own[b, ja:jb + 1, c - 2 + lo:c - 2 + hi + 1] += 1
        # computed positions: lanes 0 .. 62, rows ja - 2 .. jb + 1; loaded: lane 63 too, rows ja - 6 .. jb + 2
        assert B.ilo <= c - 2 and c + 60 <= B.ihi and B.jlo <= ja - 2 and jb + 1 <= B.jhi, (b, c, ja, jb)
        assert c + 61 <= dc.nx_block and ja - 6 >= 1 and jb + 2 <= dc.ny_block
    marched = own.copy()
    win = np.zeros_like(own)
    for (b, i0, j0, reg), inz in zip(pl["tiles"], pl["in_zone"]):
        B = blks[b]
        i1, j1 = min(i0 + ex - 4, B.ihi), min(j0 + ey - 4, B.jhi)
        (win if inz else own)[b, j0:j1 + 1, i0:i1 + 1] += 1
        assert reg or not inz
    for b, B in enumerate(blks):
        inner = own[b, B.jlo:B.jhi + 1, B.ilo:B.ihi + 1]
        assert (inner == 1).all(), f"block {b}: {int((inner != 1).sum())} interior cells not owned exactly once"
        own[b, B.jlo:B.jhi + 1, B.ilo:B.ihi + 1] = 0
    assert not own.any(), "a cell outside the interior is owned"
    assert (marched == win).all(), "the work items do not own exactly the cells of the windows they replace"
    if len(pl["items"]):
        rows = pl["items"][:, 3] - pl["items"][:, 2] + 1
        assert rows.max() <= pl["segment_rows"] and (seg or len(pl["items"]) <= 2048 or pl["segment_rows"] == 8)
    if (nx, ny) == (3600, 2400) and bx == 3600:
        assert len(pl["items"]) == 2013 and pl["segment_rows"] == 73 and int(marched.sum()) == 8525130


def test_cgrid_resident_window_handoff_graph_against_a_python_restatement():
    """The library's hand-off graph against a restatement of the rule in Python from the window table itself, over random cuts --
    including the few that keep one-way hand-offs: counted alike, and every one of them with a chain back of at most three."""
    from cice_amd import decomp
    rng = np.random.default_rng(7)
    seen_oneway = 0
    for _ in range(40):
        nbx, nby = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        bx, by = int(rng.integers(3, 45)), int(rng.integers(3, 45))
        nx, ny = bx * nbx - int(rng.integers(0, min(bx, 3))), by * nby - int(rng.integers(0, min(by, 3)))
        ew, ns = str(rng.choice(["cyclic", "closed"])), str(rng.choice(["closed", "cyclic"]))
        dc = decomp.Decomp(nx, ny, bx, by, ew, ns, 1)
        d, keep = evp.make_dims(dc, 0)
        ob = dc.local_blocks(0)
        P = evp.cgrid_window_plan(d, 16, 16, 1)
        owner = {}
        for w, ((k, i0, j0, _), tab) in enumerate(zip(P["tiles"], P["tab"])):
            b = ob[k]
            for ty in range(2, 15):
                for tx in range(2, 15):
                    if i0 - 2 + tx <= b.ihi and j0 - 2 + ty <= b.jhi:
                        owner[int(tab[ty, tx])] = w
        reads = {w: set() for w in range(len(P["tiles"]))}
        for w, ((k, i0, j0, _), tab) in enumerate(zip(P["tiles"], P["tab"])):
            b = ob[k]
            lx, ly = min(14, 2 + b.ihi - i0), min(14, 2 + b.jhi - j0)
            for ty in range(17):
                for tx in range(17):
                    e = int(tab[ty, tx])
                    mine = 2 <= tx <= 14 and 2 <= ty <= 14 and i0 - 2 + tx <= b.ihi and j0 - 2 + ty <= b.jhi
                    if (tx, ty) == (16, 16) or mine or e < 0 or tx > lx + 3 or ty > ly + 3:
                        continue
                    if owner[e] != w:
                        reads[w].add(owner[e])
        edges = [(w, p) for w in reads for p in reads[w]]
        oneway = unsafe = 0
        for w, p in edges:
            if w in reads[p]:
                continue
            oneway += 1
            front, found = {p}, False
            for _len in range(3):                    # a chain p reads ... reads w of at most three hand-offs
                front = set().union(*[reads[x] for x in front]) if front else set()
                if w in front:
                    found = True
                    break
            unsafe += not found
        g = evp.cgrid_window_deps(d)
        assert (g["windows"], g["edges"], g["oneway"], g["unsafe"]) == (len(P["tiles"]), len(edges), oneway, unsafe), (nx, ny, bx, by, ew, ns, g, len(edges), oneway, unsafe)
        seen_oneway += oneway > 0
    assert seen_oneway >= 1          # (the draw holds cuts with one-way hand-offs)
