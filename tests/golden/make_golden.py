#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs oracle/_ref/evp_ref_harness_strict -- the reference's own evp() compiled
unmodified from /root/reference by oracle/ref/build_ref.sh (amdflang -O2
-ffp-contract=off) -- on small self-contained cases and freezes, per case:
  * static grid + metric arrays and the EVP scalars,
  * the model state evp() is entered with (pr*: inputs of its preparation phase, f-2),
  * every input of the EVP subcycle captured at the drop-in boundary (in*) and the other
    products of the preparation phase (pq*),
  * the reference's outputs after nsub subcycles for several nsub.
A fixture is data only (inputs + expected outputs).  Re-run in the development
container (needs /root/reference for build_ref.sh):  python tests/golden/make_golden.py
"""
from __future__ import annotations

import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle" / "ref"))

import run_ref  # noqa: E402
from cice_amd import synth  # noqa: E402

OUT = Path(__file__).resolve().parent

STATIC = ["HTE", "HTN", "dxT", "dyT", "tarea", "uarear", "cxp", "cyp", "cxm", "cym", "dxhy", "dyhx",
          "DminTarea", "dxU", "dyU", "tarear",   # the last three: deformations (next tier f-1)
          "uarea", "fcor_blk", "hwater", "hm", "tmask", "umask"]   # preparation phase (next tier f-2)

CASES = {
    # name: (nx, ny, bx, by, ew, ns, harness kwargs)
    "rect_cyc_2x2_full": (24, 20, 12, 10, "cyclic", "closed",
                          dict(grid_kind="rect", icecase="full", nsub_list=[1, 10, 120], ncalls=2)),
    "pop_cyc_1blk_patchy": (24, 20, 24, 20, "cyclic", "closed",
                            dict(grid_kind="popfile", icecase="patchy", nsub_list=[1, 2, 120], ncalls=1)),
    "pop_cyc_3x2pad_caps": (26, 22, 10, 12, "cyclic", "closed",
                            dict(grid_kind="popfile", icecase="caps", nsub_list=[1, 120], ncalls=2)),
    "pop_closed_2x2_revp": (24, 20, 12, 10, "closed", "closed",
                            dict(grid_kind="popfile", icecase="full", nsub_list=[1, 120], ncalls=2,
                                 h_revised=True, h_arlx=300.0, h_brlx=300.0)),
    "pop_cyc_2x1_cap0_ktens": (24, 20, 12, 20, "cyclic", "closed",
                               dict(grid_kind="popfile", icecase="patchy", nsub_list=[1, 120], ncalls=1,
                                    h_capping=0.0, h_Ktens=0.2, h_e_yield=1.5, h_e_plast=2.5)),
    # fractional capping: the general branch of visc_replpress (both quotients, ice_dyn_shared.F90:2469-2470)
    "pop_cyc_2x2_cap05": (24, 20, 12, 10, "cyclic", "closed",
                          dict(grid_kind="popfile", icecase="caps", nsub_list=[1, 120], ncalls=1,
                               h_capping=0.5, h_Ktens=0.1)),
    "pop_cyc_2x2_seabed": (24, 20, 12, 10, "cyclic", "closed",
                           dict(grid_kind="popfile", icecase="full", nsub_list=[1, 120], ncalls=2,
                                h_seabed=True)),
    # seabed stress, probabilistic method (seabed_stress_factor_prob, ice_dyn_shared.F90:1475-1683)
    "pop_cyc_2x2_seabedprob": (24, 20, 12, 10, "cyclic", "closed",
                               dict(grid_kind="popfile", icecase="patchy", nsub_list=[1, 120], ncalls=2,
                                    h_seabed=True, h_seabed_method="probabilistic")),
    # the ice cover changes between the two calls (cells gain and lose ice: dyn_prep2's new-ice / no-ice branches, :747-764)
    # and the sea surface slopes (ssh_stress = 'coupled'): the preparation phase's inputs pr02 differ from pr01
    "pop_cyc_2x2_evolve_coupled": (24, 20, 12, 10, "cyclic", "closed",
                                   dict(grid_kind="popfile", icecase="patchy", nsub_list=[1, 120], ncalls=2, h_evolve=True,
                                        h_ssh="coupled")),
    # tripole (u-fold) north boundary: seam-row averaging, mirrored ghost row, and -- in the
    # expected outputs only -- evp()'s ice_HaloUpdate_stress symmetrisation after the loop
    "trip_cyc_2x2_full": (28, 20, 14, 10, "cyclic", "tripole",
                          dict(grid_kind="tripolefile", icecase="full", nsub_list=[1, 2, 120], ncalls=2)),
    "trip_cyc_1blk_patchy": (24, 18, 24, 18, "cyclic", "tripole",
                             dict(grid_kind="tripolefile", icecase="patchy", nsub_list=[1, 120], ncalls=1)),
    "trip_cyc_4x3_caps": (32, 24, 8, 8, "cyclic", "tripole",
                          dict(grid_kind="tripolefile", icecase="caps", nsub_list=[1, 120], ncalls=2)),
    # tripoleT (T-fold, ice_domain.F90:260): the fold runs through the T-cell centres of the top row; the U points of that
    # row are images of row NY-1 (ice_boundary.F90:1563-1622: NE-corner offsets (0, 1), no pair averaging)
    "tript_cyc_2x2_full": (28, 20, 14, 10, "cyclic", "tripoleT",
                           dict(grid_kind="tripolefile", icecase="full", nsub_list=[1, 2, 120], ncalls=2)),
    "tript_cyc_1blk_patchy": (24, 18, 24, 18, "cyclic", "tripoleT",
                              dict(grid_kind="tripolefile", icecase="patchy", nsub_list=[1, 120], ncalls=1)),
}


# C-grid subcycle: the harness captures the loop's inputs after a preparation-only evp() call
# (module-private ones through oracle/ref/evp_peek.c) and the reference's outputs of evp() with grid_ice = 'C'
CGRID_STATIC = ["dxT", "dyT", "dxU", "dyU", "dxE", "dyE", "dxN", "dyN", "uarea", "tarea", "earea", "narea", "earear",
                "narear", "epm", "npm", "uvm", "hm", "DminTarea", "ratiodxN", "ratiodxNr", "ratiodyE", "ratiodyEr"]
CGRID_PREP_STATIC = ["tmask", "umaskCD", "emask", "nmask", "fcor_blk", "fcorE_blk", "fcorN_blk", "hwater"]
CGRID_CASES = {
    "cgrid_cyc_2x2_patchy": (24, 20, 12, 10, "cyclic", "closed",
                             dict(icecase="patchy", nsub_list=[1, 2, 120], ncalls=2, h_evolve=True)),
    "cgrid_closed_2x2_revp": (24, 20, 12, 10, "closed", "closed",
                              dict(icecase="full", nsub_list=[1, 120], ncalls=2, h_revised=True, h_arlx=300.0,
                                   h_brlx=300.0, h_evolve=True, h_ssh="coupled")),
    "cgrid_cyc_3x2pad_cap05_avgstrength": (26, 22, 10, 12, "cyclic", "closed",
                                           dict(icecase="caps", nsub_list=[1, 120], ncalls=1, h_capping=0.5,
                                                h_Ktens=0.1, h_visc_method="avg_strength")),
    "cgrid_cyc_1blk_seabed": (24, 20, 24, 20, "cyclic", "closed",
                              dict(icecase="full", nsub_list=[1, 120], ncalls=1, h_seabed=True)),
    # seabed stress, probabilistic method at E / N points (seabed_stress_factor_prob with TbE / TbN, ice_dyn_shared.F90:1656-1676)
    "cgrid_cyc_2x2_seabedprob": (24, 20, 12, 10, "cyclic", "closed",
                                 dict(icecase="patchy", nsub_list=[1, 120], ncalls=1, h_seabed=True,
                                      h_seabed_method="probabilistic")),
    "cgrid_cyccyc_2x2_cap0_ktens": (24, 20, 12, 10, "cyclic", "cyclic",
                                    dict(icecase="patchy", nsub_list=[1, 120], ncalls=1, h_capping=0.0, h_Ktens=0.2,
                                         h_e_yield=1.5, h_e_plast=2.5, h_ssh="coupled")),
    # tripole (u-fold): N-face fields lie ON the fold (top row averaged pairwise), E-face / centre fields mirror across it
    "cgrid_trip_2x2_full": (28, 20, 14, 10, "cyclic", "tripole",
                            dict(icecase="patchy", nsub_list=[1, 2, 120], ncalls=2, h_evolve=True)),
    "cgrid_trip_1blk_patchy_avgstrength": (24, 18, 24, 18, "cyclic", "tripole",
                                           dict(icecase="patchy", nsub_list=[1, 120], ncalls=1, h_capping=0.5,
                                                h_visc_method="avg_strength")),
    "cgrid_trip_4x3_caps_seabed": (32, 24, 8, 8, "cyclic", "tripole",
                                   dict(icecase="caps", nsub_list=[1, 120], ncalls=1, h_seabed=True)),
    # tripoleT (T-fold): centre and E-face fields lie ON the fold (top row made symmetric pairwise and rewritten from its
    # mirror), NE-corner and N-face fields have their top row as the image of row NY-1 (ice_boundary.F90:1563-1622)
    "cgtript_cyc_2x2_patchy": (28, 20, 14, 10, "cyclic", "tripoleT",
                               dict(icecase="patchy", nsub_list=[1, 2, 120], ncalls=2, h_evolve=True)),
    "cgtript_cyc_1blk_full_avgstrength": (24, 18, 24, 18, "cyclic", "tripoleT",
                                          dict(icecase="full", nsub_list=[1, 120], ncalls=1, h_visc_method="avg_strength")),
}


def make_cgrid_case(name, spec):
    nx, ny, bx, by, ew, ns, kw = spec
    kw = dict(kw)
    td = tempfile.mkdtemp(prefix="golden_")
    g = synth.make_grid(nx, ny, dx0=1.1e5, ns=("tripole" if ns == "tripoleT" else ns))
    run_ref.write_pop_grid(td + "/grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
    run_ref.write_kmt(td + "/kmt.bin", g["kmt"])
    d, txt = run_ref.run_harness(nx, ny, bx, by, ew=ew, ns=ns, variant="strict", h_ndte=120,
                                 grid_kind=("tripolefile" if ns in ("tripole", "tripoleT") else "popfile"),
                                 grid_files=(td + "/grid.bin", td + "/kmt.bin"), h_grid_ice="C", **kw)
    keep = {"dims": d["dims"], "blkinfo": d["blkinfo"], "scalars": d["scalars"], "nsub_list": d["nsub_list"],
            "ew": np.array(ew), "ns": np.array(ns), "visc_method": np.array(kw.get("h_visc_method", "avg_zeta"))}
    for k in CGRID_STATIC:
        keep[k] = d[k]
    keep["tarear"] = d["tarear"]                 # deformationsC_T (the o*_divu / shear / vort / rdg_* arrays)
    for k in CGRID_PREP_STATIC:                  # the preparation phase on the C grid (cp*: what it reads per call)
        keep[k] = d[k]
    for k, v in d.items():
        if k[:2] in ("in", "cp") or (k.startswith("o") and k[1:3].isdigit()):
            keep[k] = v
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **keep)
    print(f"{name}: {path.stat().st_size/1024:.0f} KiB, active T/U/E/N cells "
          f"{[int(d['in01_ice%smask' % c].sum()) for c in 'TUEN']}, max|uE| {np.abs(d['o01n0120_uvelE']).max():.4f}")


def make_case(name, spec):
    nx, ny, bx, by, ew, ns, kw = spec
    kw = dict(kw)
    grid_files = None
    td = tempfile.mkdtemp(prefix="golden_")
    if kw["grid_kind"] != "rect":
        g = synth.make_grid(nx, ny, dx0=1.1e5, ns=("tripole" if ns == "tripoleT" else ns))
        run_ref.write_pop_grid(td + "/grid.bin", g["ULAT"], g["ULON"], g["HTN"] * 100.0, g["HTE"] * 100.0)
        run_ref.write_kmt(td + "/kmt.bin", g["kmt"])
        grid_files = (td + "/grid.bin", td + "/kmt.bin")
    d, txt = run_ref.run_harness(nx, ny, bx, by, ew=ew, ns=ns, variant="strict", h_ndte=120,
                                 grid_files=grid_files, **kw)
    keep = {"dims": d["dims"], "blkinfo": d["blkinfo"], "scalars": d["scalars"],
            "nsub_list": d["nsub_list"], "ew": np.array(ew), "ns": np.array(ns)}
    for k in STATIC:
        keep[k] = d[k]
    for k, v in d.items():
        if k[:2] in ("in", "pr", "pq") or (k.startswith("o") and k[1:3].isdigit()):
            # next-tier diagnostics (deformations, dyn_finish) are kept for SURVEY §8 f-1
            keep[k] = v
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **keep)
    nact = int(d["in01_iceTmask"].sum()), int(d["in01_iceUmask"].sum())
    print(f"{name}: {path.stat().st_size/1024:.0f} KiB, active T/U cells {nact}, "
          f"max|u| {np.abs(d['o01n0120_uvel']).max():.4f}")


if __name__ == "__main__":
    if not run_ref.have_ref("strict"):
        raise SystemExit("build the reference first: oracle/ref/build_ref.sh strict")
    only = sys.argv[1:]
    for name, spec in CASES.items():
        if only and name not in only:
            continue
        make_case(name, spec)
    for name, spec in CGRID_CASES.items():
        if only and name not in only:
            continue
        make_cgrid_case(name, spec)
