"""Synthetic, self-contained workloads for the EVP subcycle.

The real gx3/gx1/tx1 grid and forcing files live outside the reference
repository (configuration/scripts/options/set_nml.gx1:6-12), so every size in
BASELINE.json is generated analytically here: a curvilinear-metric global grid
(non-zero dxhy/dyhx, cxp != cyp), a land mask with closed N/S rows and two
"continents", and the fields that `evp()` hands to its EVP core
(dyn_evp1d_run's argument list, ice_dyn_evp1d.F90:121-153).

Global arrays are numpy [ny_global][nx_global] (i fastest), i.e. the memory
image of Fortran (nx_global, ny_global).
"""
from __future__ import annotations

import numpy as np

OMEGA = 7.292e-5          # shared/ice_constants.F90:23
RHOW = 1026.0             # Icepack default, consumed at ice_dyn_shared.F90:920
RHOI, RHOS = 917.0, 330.0

# named sizes of BASELINE.json `configs`
GRIDS = {
    "gx3": dict(nx=100, ny=116, dx0=3.3e5, ns="closed"),
    "gx1": dict(nx=320, ny=384, dx0=1.1e5, ns="closed"),
    "tx1": dict(nx=360, ny=240, dx0=1.1e5, ns="tripole"),
    "s01": dict(nx=3600, ny=2400, dx0=1.1e4, ns="closed"),   # synthetic 0.1-degree class
    # per-rank pieces of a 0.25-degree class grid (1440x1080) on 8 and 4 GPUs: sizes just beyond what
    # the on-chip resident kernel holds (tools/selfx_timing.py: where does the riding exchange pay?)
    "q8": dict(nx=720, ny=270, dx0=2.8e4, ns="closed"),
    "q4": dict(nx=720, ny=540, dx0=2.8e4, ns="closed"),
    "q1": dict(nx=1440, ny=1080, dx0=2.8e4, ns="closed"),    # the whole 0.25-degree class grid
    # 320 tiles of 16x16: every CU holds one or two tiles of the on-chip resident kernel (pace of a SIMD with two waves)
    "p2": dict(nx=300, ny=240, dx0=1.1e5, ns="closed"),
    # a small tripole grid (tests: every variant of the resident C-grid kernel's FOLD form fits with two workgroups per CU)
    "tx3": dict(nx=100, ny=116, dx0=3.3e5, ns="tripole"),
}


def make_grid(nx: int, ny: int, dx0: float = 1.1e5, dy0: float | None = None, ns: str = "closed",
              continents: bool = True) -> dict:
    """Global geometry: HTN/HTE [m] (N and E edge lengths of T-cells,
    ice_grid.F90:1000-1061), ULAT/ULON [rad], kmt (1 = ocean)."""
    dy0 = dx0 if dy0 is None else dy0
    i = np.arange(1, nx + 1, dtype=np.float64)[None, :]
    j = np.arange(1, ny + 1, dtype=np.float64)[:, None]
    HTN = dx0 * (1.0 + 0.3 * np.cos(2 * np.pi * (j - 0.5) / ny)) * (1.0 + 0.05 * np.sin(2 * np.pi * i / nx))
    HTE = dy0 * (1.0 + 0.2 * np.cos(2 * np.pi * (j - 0.5) / ny)) * (1.0 + 0.04 * np.cos(2 * np.pi * i / nx))
    ULAT = np.deg2rad(-78.0 + 165.0 * j / ny) * np.ones((1, nx))
    ULON = np.deg2rad(360.0 * i / nx) * np.ones((ny, 1))
    kmt = np.ones((ny, nx), dtype=np.int32)
    if ns == "closed":
        kmt[:2, :] = 0
        kmt[-2:, :] = 0          # cf. rectgrid, ice_grid.F90:2752-2755
    else:
        kmt[:2, :] = 0           # tripole: closed in the south only
        # the two northern poles of a tripole grid sit on land (U points i = nx/2 and i = nx of
        # the seam row): mask the T-cells around them, as on real tx1-type grids
        for ic in (nx // 2, nx):
            for di in (-1, 0, 1, 2):
                kmt[-2:, (ic - 1 + di) % nx] = 0
    if continents and nx >= 20 and ny >= 20:
        kmt[int(0.30 * ny):int(0.55 * ny), int(0.10 * nx):int(0.30 * nx)] = 0
        kmt[int(0.60 * ny):int(0.80 * ny), int(0.55 * nx):int(0.80 * nx)] = 0
    return dict(nx=nx, ny=ny, ns=ns, ew="cyclic", HTN=HTN, HTE=HTE, ULAT=ULAT, ULON=ULON, kmt=kmt)


def derive_geometry(g: dict) -> dict:
    """dxT,dyT,dxU,dyU,tarea,uarea,uarear and masks on the global grid, by the
    formulas of primary_grid_lengths_HTN/HTE (ice_grid.F90:3063-3280),
    init_grid2 (:676-714) and makemask (:3382-3383); E-W cyclic."""
    HTN, HTE = g["HTN"], g["HTE"]
    dxU = 0.5 * (HTN + np.roll(HTN, -1, axis=1))
    dxT = np.empty_like(HTN)
    dxT[1:] = 0.5 * (HTN[1:] + HTN[:-1])
    dxT[0] = 2.0 * HTN[1] - HTN[2]
    dyT = 0.5 * (HTE + np.roll(HTE, 1, axis=1))
    dyU = np.empty_like(HTE)
    dyU[:-1] = 0.5 * (HTE[:-1] + HTE[1:])
    dyU[-1] = 2.0 * HTE[-2] - HTE[-3]
    tarea = dxT * dyT
    uarea = dxU * dyU
    hm = g["kmt"].astype(np.float64)
    hm_n = np.vstack([hm[1:], np.zeros((1, hm.shape[1]))])     # (i, j+1); closed/seam row: land
    uvm = np.minimum(np.minimum(hm, np.roll(hm, -1, axis=1)),
                     np.minimum(hm_n, np.roll(hm_n, -1, axis=1)))
    out = dict(g)
    out.update(dxT=dxT, dyT=dyT, dxU=dxU, dyU=dyU, tarea=tarea, uarea=uarea,
               uarear=np.where(uarea > 0, 1.0 / uarea, 0.0),
               tarear=np.where(tarea > 0, 1.0 / tarea, 0.0),
               hm=hm, uvm=uvm, tmask=hm > 0.5, umask=uvm > 0.5)
    return out


def make_state(g: dict, case: str = "full", dt: float = 3600.0, seed: int | None = None,
               warm: bool = False) -> dict:
    """Inputs of the EVP subcycle on the global grid (U-grid fields at NE corners).

    case 'full' : ice on every ocean cell (headline case, every ocean cell active)
    case 'caps' : ice only in the top/bottom 25 % of rows, tapered (~35 % active)
    seed        : PCG64 +-1 % multiplicative perturbation of ice thickness
    warm        : non-zero initial stresses/velocities (otherwise cold start)
    """
    nx, ny = g["nx"], g["ny"]
    x = (np.arange(1, nx + 1) - 0.5)[None, :] / nx * np.ones((ny, 1))
    y = (np.arange(1, ny + 1) - 0.5)[:, None] / ny * np.ones((1, nx))
    tmask, umask = g["tmask"], g["umask"]

    aice = np.where(tmask, 0.95 * (1.0 - 0.04 * np.sin(2 * np.pi * x) * np.cos(4 * np.pi * y)), 0.0)
    hi = 2.0 * (1.0 + 0.1 * np.sin(4 * np.pi * y) * np.cos(2 * np.pi * x))
    if case == "caps":
        taper = np.clip((np.abs(y - 0.5) - 0.25) / 0.05, 0.0, 1.0)
        aice = aice * taper
    elif case != "full":
        raise ValueError(case)
    if seed is not None:
        rng = np.random.Generator(np.random.PCG64(seed))
        hi = hi * (1.0 + 0.01 * (2.0 * rng.random((ny, nx)) - 1.0))
    vice = hi * aice
    iceT = tmask & (aice > 1e-11)
    # Hibler-form strength (a fixture INPUT, like in the reference harness)
    strength = np.where(iceT, 2.75e4 * vice * np.exp(-20.0 * (1.0 - aice)), 0.0)

    def t2u(a):   # unweighted 4-point T -> U average (stand-in for grid_average_X2Y 'S')
        an = np.vstack([a[1:], a[-1:]])
        return 0.25 * (a + np.roll(a, -1, axis=1) + an + np.roll(an, -1, axis=1))

    aiU = t2u(aice)
    umass = t2u(RHOI * vice + RHOS * 0.2 * aice)
    iceU = umask & (aiU > 1e-11) & (umass > 1e-10)
    z = np.zeros((ny, nx))
    uocn = 0.2 * y - 0.1                      # box2001-style currents, ice_forcing.F90:5242-5245
    vocn = -0.2 * x + 0.1
    fcor = 2.0 * OMEGA * np.sin(g["ULAT"])
    fm = np.where(iceU, fcor * umass, 0.0)
    umassdti = np.where(iceU, umass / dt, 0.0)
    strairx = aiU * 0.1 * np.sin(2 * np.pi * x) * np.sin(np.pi * y)
    strairy = aiU * 0.1 * np.sin(np.pi * x) * np.sin(2 * np.pi * y)
    # geostrophic tilt, ice_dyn_shared.F90:823-826 ; waterx with cosw=1, sinw=0 (:819-820)
    forcex = np.where(iceU, strairx - fm * vocn, 0.0)
    forcey = np.where(iceU, strairy + fm * uocn, 0.0)
    st = dict(
        strength=strength, cdn_ocnU=np.full((ny, nx), 0.00536), aiU=aiU, uocnU=uocn, vocnU=vocn,
        waterxU=np.where(iceU, uocn, 0.0), wateryU=np.where(iceU, vocn, 0.0),
        forcexU=forcex, forceyU=forcey, umassdti=umassdti, fmU=fm, TbU=z.copy(),
        strintxU=z.copy(), strintyU=z.copy(), taubxU=z.copy(), taubyU=z.copy(),
        iceTmask=iceT.astype(np.int32), iceUmask=iceU.astype(np.int32),
    )
    if warm:
        u0 = np.where(iceU, 0.05 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), 0.0)
        v0 = np.where(iceU, 0.05 * np.cos(2 * np.pi * x) * np.sin(4 * np.pi * y), 0.0)
        s0 = np.where(iceT, -0.1 * strength * (1.0 + 0.3 * np.sin(6 * np.pi * x)), 0.0)
    else:
        u0, v0, s0 = z, z, z
    st["uvel"], st["vvel"] = u0.copy(), v0.copy()
    st["uvel_init"], st["vvel_init"] = u0.copy(), v0.copy()
    for k, fac in (("stressp", 1.0), ("stressm", 0.1), ("stress12", 0.05)):
        for c in range(1, 5):
            st[f"{k}_{c}"] = s0 * fac * (1.0 + 0.01 * c)
    return st


def evp_scalars(ndte: int, dt: float = 3600.0, revised_evp: bool = False, elasticDamp: float = 0.36,
                arlx: float = 300.0, brlx: float = 300.0, e_yieldcurve: float = 2.0,
                e_plasticpot: float = 2.0, capping: float = 1.0, Ktens: float = 0.0,
                deltaminEVP: float = 1e-11) -> dict:
    """set_evp_parameters (ice_dyn_shared.F90:453-486) as plain host arithmetic."""
    epp2i = 1.0 / e_plasticpot ** 2
    e_factor = e_yieldcurve ** 2 / e_plasticpot ** 4
    if revised_evp:
        revp, denom1, arlx1i = 1.0, 1.0, 1.0 / arlx
    else:
        revp = 0.0
        arlx = 2.0 * elasticDamp * float(ndte)
        arlx1i = 1.0 / arlx
        brlx = float(ndte)
        denom1 = 1.0 / (1.0 + arlx1i)
    return dict(ndte=ndte, arlx1i=arlx1i, denom1=denom1, brlx=brlx, revp=revp, e_factor=e_factor,
                epp2i=epp2i, capping=capping, Ktens=Ktens, deltaminEVP=deltaminEVP,
                u0=5e-5, cosw=1.0, sinw=0.0, rhow=RHOW)


def make_primary(g: dict, case: str = "full", seed: int = 1) -> dict:
    """The model state evp() is entered with (inputs of its preparation phase), on
    the global grid: T-grid fields aice..strairyT, the previous velocities / stresses / U mask
    (with cells that gain and lose ice), and the static fields dyn_prep1/2 and the T->U averages
    read.  Analytic + PCG64 noise; for parity and timing runs of cice_evp_hip_prep."""
    nx, ny = g["nx"], g["ny"]
    x = (np.arange(1, nx + 1) - 0.5)[None, :] / nx * np.ones((ny, 1))
    y = (np.arange(1, ny + 1) - 0.5)[:, None] / ny * np.ones((1, nx))
    rng = np.random.Generator(np.random.PCG64(seed))
    tmask, umask = g["tmask"], g["umask"]
    aice = np.where(tmask, 0.9 + 0.05 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), 0.0)
    if case == "caps":
        aice = aice * np.clip((np.abs(y - 0.5) - 0.25) / 0.05, 0.0, 1.0)
    holes = np.sin(11 * np.pi * x) * np.sin(7 * np.pi * y) > 0.8          # open water patches
    aice = np.where(holes, 0.0, aice)
    hi = 2.0 * (1.0 + 0.1 * np.sin(4 * np.pi * y) * np.cos(2 * np.pi * x)) * (1.0 + 0.01 * (2 * rng.random((ny, nx)) - 1))
    vice = hi * aice
    t = dict(aice=aice, vice=vice, vsno=0.2 * aice, aice_init=aice * (1.0 - 0.01 * rng.random((ny, nx))),
             cdn_ocn=0.00536 * (1.0 + 0.1 * rng.random((ny, nx))), uocn=0.2 * y - 0.1, vocn=-0.2 * x + 0.1,
             ss_tltx=1e-6 * np.sin(2 * np.pi * y), ss_tlty=1e-6 * np.cos(2 * np.pi * x),
             strairxT=aice * 0.1 * np.sin(2 * np.pi * x) * np.sin(np.pi * y),
             strairyT=aice * 0.1 * np.sin(np.pi * x) * np.sin(2 * np.pi * y))
    old = umask & (rng.random((ny, nx)) > 0.1)               # previous U mask: ~10 % "new ice" cells
    state = dict(uvel=np.where(old, 0.05 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), 0.0),
                 vvel=np.where(old, 0.05 * np.cos(2 * np.pi * x) * np.sin(4 * np.pi * y), 0.0),
                 iceUmask=old.astype(np.int32))
    for k, name in enumerate(["stressp_1", "stressp_2", "stressp_3", "stressp_4", "stressm_1", "stressm_2",
                              "stressm_3", "stressm_4", "stress12_1", "stress12_2", "stress12_3", "stress12_4"]):
        state[name] = np.where(tmask, -1000.0 * (1.0 + 0.1 * k) * (1.0 + 0.3 * np.sin(6 * np.pi * x)), 0.0)
    static = dict(tmask=tmask.astype(np.int32), umask=umask.astype(np.int32), hm=g["hm"], tarea=g["tarea"],
                  uarea=g["uarea"], fcor_blk=2.0 * OMEGA * np.sin(g["ULAT"]))
    return dict(t=t, state=state, static=static)


# ---- C-grid workloads: what evp()'s loop for grid_ice = 'C' reads, on the global grid ----------
def cgrid_geometry(g: dict, deltaminEVP: float = 1e-11) -> dict:
    """`g` from derive_geometry -> the 23 static arrays of cice_evp_hip_cgrid_set_geometry (E-W cyclic, closed in
    y).  Lengths at the faces as in ice_grid.F90 (dyE = HTE, dxN = HTN; dxE, dyN the four-point means of HTN, HTE around the face on closed
    grids -- the reference's start-up formulas, which the marched C-grid kernel can then form itself --, through the cell centres on the
    synthetic tripole grids, whose fold rows are built to mirror), face
    masks epm / npm = both neighbouring T-cells ocean (makemask), boundary-condition ratios as init_evp builds them
    (ice_dyn_evp.F90:232-239)."""
    dxT, dyT, hm = g["dxT"], g["dyT"], g["hm"]
    trip = g.get("ns") == "tripole"
    # (i, j+1): beyond the top row a closed grid repeats it; a tripole grid mirrors across the fold -- cell-centre
    # fields column NX-i+1, E-face fields column NX-i (u-fold, ice_boundary.F90:1626-1683)
    n = lambda a: np.vstack([a[1:], a[-1:, ::-1] if trip else a[-1:]])
    n_e = lambda a: np.vstack([a[1:], np.roll(a[-1:, ::-1], -1, axis=1) if trip else a[-1:]])
    e = lambda a: np.roll(a, -1, axis=1)                     # (i+1, j), cyclic
    dxN, dyE = g["HTN"], g["HTE"]
    if trip:
        dxE = 0.5 * (dxT + e(dxT))
        dyN = 0.5 * (dyT + n(dyT))
    else:
        # the reference's four-point means (primary_grid_lengths_HTN / _HTE, ice_grid.F90:3139-3146, 3245-3253), in its order of
        # summation; the rows it extrapolates (j = 1, j = ny_global: land in these grids) repeat their neighbour here
        s_ = lambda a: np.vstack([a[:1], a[:-1]])             # (i, j-1)
        w_ = lambda a: np.roll(a, 1, axis=1)                  # (i-1, j)
        dxE = 0.25 * (dxN + e(dxN) + s_(dxN) + e(s_(dxN)))
        dyN = 0.25 * (dyE + w_(dyE) + n(dyE) + w_(n(dyE)))
    earea, narea = dxE * dyE, dxN * dyN
    hm_n = np.vstack([hm[1:], hm[-1:, ::-1] if trip else np.zeros((1, hm.shape[1]))])
    epm = np.minimum(hm, e(hm))
    npm = np.minimum(hm, hm_n)
    rxN = -e(dxN) / dxN
    ryE = -n_e(dyE) / dyE
    return dict(dxT=dxT, dyT=dyT, dxU=g["dxU"], dyU=g["dyU"], dxE=dxE, dyE=dyE, dxN=dxN, dyN=dyN, uarea=g["uarea"],
                tarea=g["tarea"], earea=earea, narea=narea, earear=1.0 / earea, narear=1.0 / narea, epm=epm, npm=npm,
                uvm=g["uvm"], hm=hm, DminTarea=deltaminEVP * g["tarea"], ratiodxN=rxN, ratiodxNr=1.0 / rxN,
                ratiodyE=ryE, ratiodyEr=1.0 / ryE, _deltaminEVP=np.float64(deltaminEVP))


def cgrid_state(g: dict, cg: dict, case: str = "full", dt: float = 3600.0, seed: int | None = None,
                warm: bool = True, seabed: bool = False) -> dict:
    """(state14, inputs23, masks4) of the C-grid loop: the same ice cover, forcing and currents as make_state, with
    the momentum terms at the E and N faces (two-point averages of the T-cell fields, as dyn_prep2 leaves them)."""
    nx, ny = g["nx"], g["ny"]
    x = (np.arange(1, nx + 1) - 0.5)[None, :] / nx * np.ones((ny, 1))
    y = (np.arange(1, ny + 1) - 0.5)[:, None] / ny * np.ones((1, nx))
    tmask = g["tmask"]
    aice = np.where(tmask, 0.95 * (1.0 - 0.04 * np.sin(2 * np.pi * x) * np.cos(4 * np.pi * y)), 0.0)
    hi = 2.0 * (1.0 + 0.1 * np.sin(4 * np.pi * y) * np.cos(2 * np.pi * x))
    if case == "caps":
        aice = aice * np.clip((np.abs(y - 0.5) - 0.25) / 0.05, 0.0, 1.0)
    elif case != "full":
        raise ValueError(case)
    if seed is not None:
        rng = np.random.Generator(np.random.PCG64(seed))
        hi = hi * (1.0 + 0.01 * (2.0 * rng.random((ny, nx)) - 1.0))
    vice = hi * aice
    iceT = tmask & (aice > 1e-11)
    strength = np.where(iceT, 2.75e4 * vice * np.exp(-20.0 * (1.0 - aice)), 0.0)
    mass = RHOI * vice + RHOS * 0.2 * aice
    n = lambda a: np.vstack([a[1:], a[-1:]])
    e = lambda a: np.roll(a, -1, axis=1)
    fcor = 2.0 * OMEGA * np.sin(g["ULAT"])
    uocn = 0.2 * y - 0.1
    vocn = -0.2 * x + 0.1
    z = np.zeros((ny, nx))
    out_in, masks = {"strength": strength}, {"iceTmask": iceT.astype(np.int32)}
    vel = {}
    for tag, sh, pm in (("E", e, cg["epm"]), ("N", n, cg["npm"])):
        ai = 0.5 * (aice + sh(aice))
        ms = 0.5 * (mass + sh(mass))
        ice = (pm > 0.5) & (ai > 1e-11) & (ms > 1e-10)
        fm = np.where(ice, fcor * ms, 0.0)
        strair = ai * 0.1 * (np.sin(2 * np.pi * x) * np.sin(np.pi * y) if tag == "E" else np.sin(np.pi * x) * np.sin(2 * np.pi * y))
        masks[f"ice{tag}mask"] = ice.astype(np.int32)
        out_in[f"cdn_ocn{tag}"] = np.full((ny, nx), 0.00536)
        out_in[f"ai{tag}"] = ai
        out_in[f"uocn{tag}"] = uocn
        out_in[f"vocn{tag}"] = vocn
        out_in[("waterxE" if tag == "E" else "wateryN")] = np.where(ice, uocn if tag == "E" else vocn, 0.0)
        out_in[("forcexE" if tag == "E" else "forceyN")] = np.where(ice, strair + (-fm * vocn if tag == "E" else fm * uocn), 0.0)
        out_in[("emassdti" if tag == "E" else "nmassdti")] = np.where(ice, ms / dt, 0.0)
        out_in[f"fm{tag}"] = fm
        out_in[f"Tb{tag}"] = np.where(ice & (y < 0.2), 0.5 * ai, 0.0) if seabed else z.copy()
        out_in[f"rheofact{tag}"] = np.where(ice, 1.0, 0.0)
        vel[tag] = ice
    iceU = g["umask"] & (0.25 * (aice + e(aice) + n(aice) + e(n(aice))) > 1e-11)
    masks["iceUmask"] = iceU.astype(np.int32)
    if warm:
        uE = np.where(vel["E"], 0.05 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), 0.0)
        vN = np.where(vel["N"], 0.05 * np.cos(2 * np.pi * x) * np.sin(4 * np.pi * y), 0.0)
        s0 = np.where(iceT, -0.1 * strength * (1.0 + 0.3 * np.sin(6 * np.pi * x)), 0.0)
    else:
        uE, vN, s0 = z.copy(), z.copy(), z.copy()
    w = lambda a: np.roll(a, 1, axis=1)
    s = lambda a: np.vstack([a[:1], a[:-1]])
    ea, na = cg["earea"], cg["narea"]
    # the other component at each face and the corner velocities, by the loop's own averages
    uN = (w(uE) * w(ea) + uE * ea + n(w(uE)) * n(w(ea)) + n(uE) * n(ea)) / (w(ea) + ea + n(w(ea)) + n(ea)) * cg["npm"]
    vE = (s(vN) * s(na) + s(e(vN)) * s(e(na)) + vN * na + e(vN) * e(na)) / (s(na) + s(e(na)) + na + e(na)) * cg["epm"]
    uU = (uE * ea + n(uE) * n(ea)) / (ea + n(ea)) * g["uvm"]
    vU = (vN * na + e(vN) * e(na)) / (na + e(na)) * g["uvm"]
    state = dict(uvelE=uE, vvelE=vE, uvelN=uN, vvelN=vN, uvel=uU, vvel=vU, stresspT=s0, stressmT=0.1 * s0,
                 stress12T=0.05 * s0, stress12U=np.where(iceU, 0.05 * 0.25 * (s0 + e(s0) + n(s0) + e(n(s0))), 0.0),
                 strintxE=z.copy(), strintyN=z.copy(), taubxE=z.copy(), taubyN=z.copy())
    out_in["uvelE_init"], out_in["vvelN_init"] = uE.copy(), vN.copy()
    return state, out_in, masks


CGRID_FILL_ONE = ("dxT", "dyT", "dxU", "dyU", "dxE", "dyE", "dxN", "dyN", "uarea", "tarea", "earea", "narea", "earear",
                  "narear", "ratiodxN", "ratiodxNr", "ratiodyE", "ratiodyEr")


CGRID_LOC = {   # where each array lives (tripole grids: which mirror rule fills its ghost row beyond the fold)
    "NEcorner": ("dxU", "dyU", "uarea", "uvm", "uvel", "vvel", "stress12U", "iceUmask"),
    "Eface": ("dxE", "dyE", "earea", "earear", "epm", "ratiodyE", "ratiodyEr", "uvelE", "vvelE", "strintxE", "taubxE",
              "cdn_ocnE", "aiE", "uocnE", "vocnE", "waterxE", "forcexE", "emassdti", "fmE", "uvelE_init", "TbE", "rheofactE",
              "iceEmask"),
    "Nface": ("dxN", "dyN", "narea", "narear", "npm", "ratiodxN", "ratiodxNr", "uvelN", "vvelN", "strintyN", "taubyN",
              "cdn_ocnN", "aiN", "uocnN", "vocnN", "wateryN", "forceyN", "nmassdti", "fmN", "vvelN_init", "TbN", "rheofactN",
              "iceNmask"),
}
CGRID_VECTOR = ("uvel", "vvel", "uvelE", "vvelE", "uvelN", "vvelN", "strintxE", "strintyN", "taubxE", "taubyN", "uocnE", "vocnE",
                "uocnN", "vocnN", "waterxE", "wateryN", "forcexE", "forceyN", "uvelE_init", "vvelN_init")


def cgrid_scatter(dc, rank: int, cg: dict, state: dict, inputs: dict, masks: dict):
    """Global C-grid workload -> the block arrays of `rank` (lengths and areas 1 where a ghost cell has no source; on a
    tripole grid the ghost row beyond the fold mirrored by the rule of each array's location)."""
    def fold(k):
        loc = next((l for l, names in CGRID_LOC.items() if k in names), "center")
        return (loc, -1.0 if k in CGRID_VECTOR else 1.0)
    sc = lambda k, v, fill: dc.scatter(v, rank, fill=fill, fold=fold(k))
    # (DminTarea = deltaminEVP * tarea on EVERY cell of the block arrays, ghost cells without a source -- tarea 1 -- included:
    # ice_dyn_shared.F90:385)
    dmin = float(cg.get("_deltaminEVP", 1e-11))
    static = {k: sc(k, v, dmin if k == "DminTarea" else (1.0 if k in CGRID_FILL_ONE else 0.0)) for k, v in cg.items()
              if not k.startswith("_")}
    # the boundary-condition ratios as init_evp builds them (ice_dyn_evp.F90:232-239): on the interior cells of each block,
    # from the BLOCK arrays (ghost cells as the halo left them) -- next to a closed boundary that is the fill value 1
    for b, blk in enumerate(dc.local_blocks(rank)):
        js, je, is_, ie = blk.jlo - 1, blk.jhi, blk.ilo - 1, blk.ihi
        dxN, dyE = static["dxN"][b], static["dyE"][b]
        rx = -dxN[js:je, is_ + 1:ie + 1] / dxN[js:je, is_:ie]
        ry = -dyE[js + 1:je + 1, is_:ie] / dyE[js:je, is_:ie]
        static["ratiodxN"][b][js:je, is_:ie], static["ratiodxNr"][b][js:je, is_:ie] = rx, 1.0 / rx
        static["ratiodyE"][b][js:je, is_:ie], static["ratiodyEr"][b][js:je, is_:ie] = ry, 1.0 / ry
    return (static, {k: sc(k, v, 0.0) for k, v in state.items()}, {k: sc(k, v, 0.0) for k, v in inputs.items()},
            {k: sc(k, v, 0) for k, v in masks.items()})


def bgrid_fold_metrics(dc, rank: int, g: dict):
    """dxhy, dyhx as block arrays with the ghost cells CICE gives them (ice_dyn_shared.F90:401-424: computed on the
    physical cells, then a halo update as cell-centre VECTOR fields with fill value 1): what the Fortran shim hands to
    cice_evp_hip_set_metrics on tripole grids, where the north ghost row is a sign-flipped mirror image."""
    HTE, HTN = g["HTE"], g["HTN"]
    dxhy = 0.5 * (HTE - np.roll(HTE, 1, axis=1))
    dyhx = np.empty_like(HTN)
    dyhx[1:] = 0.5 * (HTN[1:] - HTN[:-1])
    dyhx[0] = 0.5 * (HTN[0] - 1.0)          # (the ghost row south of a closed boundary holds the fill value 1)
    return (dc.scatter(dxhy, rank, fill=1.0, fold=("center", -1.0)), dc.scatter(dyhx, rank, fill=1.0, fold=("center", -1.0)))


def cgrid_prep_inputs(g: dict, cg: dict, case: str = "full", seed: int = 5, coupled: bool = False):
    """What evp()'s preparation phase reads on the C grid, as global [ny][nx] arrays: (tfields11, static7, masks_prev3)
    -- the T-grid state and forcing of cgrid_state's ice cover (perturbed by `seed`), the land masks / Coriolis arrays
    of the three velocity locations, and a previous call's ice masks that differ from the ones the preparation will
    find (cells gain and lose ice)."""
    nx, ny = g["nx"], g["ny"]
    x = (np.arange(1, nx + 1) - 0.5)[None, :] / nx * np.ones((ny, 1))
    y = (np.arange(1, ny + 1) - 0.5)[:, None] / ny * np.ones((1, nx))
    rng = np.random.Generator(np.random.PCG64(seed))
    tmask = g["tmask"]
    aice = np.where(tmask, 0.95 * (1.0 - 0.04 * np.sin(2 * np.pi * x) * np.cos(4 * np.pi * y)), 0.0)
    if case == "caps":
        aice = aice * np.clip((np.abs(y - 0.5) - 0.25) / 0.05, 0.0, 1.0)
    aice = np.where(rng.random((ny, nx)) < 0.03, 0.0, aice)                     # scattered open water
    hi = 2.0 * (1.0 + 0.1 * np.sin(4 * np.pi * y) * np.cos(2 * np.pi * x)) * (1.0 + 0.01 * (2.0 * rng.random((ny, nx)) - 1.0))
    vice = hi * aice
    t = dict(aice=aice, vice=vice, vsno=0.2 * aice, aice_init=aice * (1.0 - 0.02 * rng.random((ny, nx))),
             cdn_ocn=0.00536 * (1.0 + 0.1 * rng.random((ny, nx))), uocn=0.2 * y - 0.1 + 0.01 * np.sin(6 * np.pi * x),
             vocn=-0.2 * x + 0.1 + 0.01 * np.cos(4 * np.pi * y),
             ss_tltx=(2e-6 * np.sin(2 * np.pi * x) * np.cos(4 * np.pi * y)) if coupled else np.zeros((ny, nx)),
             ss_tlty=(-1.5e-6 * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y)) if coupled else np.zeros((ny, nx)),
             strairxT=aice * 0.1 * np.sin(2 * np.pi * x) * np.sin(np.pi * y),
             strairyT=aice * 0.1 * np.sin(np.pi * x) * np.sin(2 * np.pi * y))
    fcor = 2.0 * OMEGA * np.sin(g["ULAT"])
    static = dict(tmask=tmask.astype(np.int32), umaskCD=g["umask"].astype(np.int32), emask=(cg["epm"] > 0.5).astype(np.int32),
                  nmask=(cg["npm"] > 0.5).astype(np.int32), fcor_blk=fcor, fcorE_blk=0.999 * fcor, fcorN_blk=1.001 * fcor)
    prev = {k: ((rng.random((ny, nx)) < 0.8) & (static[m] != 0)).astype(np.int32)
            for k, m in (("iceUmask", "umaskCD"), ("iceEmask", "emask"), ("iceNmask", "nmask"))}
    return t, static, prev
