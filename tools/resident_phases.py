"""Where a subcycle of the on-chip resident kernel goes, per wave: shader-cycle stamps collected under
CICE_EVP_HIP_RES_PROF=1 (16 x 16 tiles).  Usage: python tools/resident_phases.py [gx1|gx3|p2] [ndte]"""
import os, sys, pathlib
R = str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0] = [R, R + "/tests", R + "/oracle"]
os.environ["CICE_EVP_HIP_RES_PROF"] = "1"
os.environ.setdefault("CICE_EVP_HIP_RESIDENT", "1")
os.environ.setdefault("CICE_EVP_HIP_RES_LOGW", "4"); os.environ.setdefault("CICE_EVP_HIP_TYB", "4")
import numpy as np
from cice_amd import evp, synth
from test_gpu_parity import synth_case
wl = sys.argv[1] if len(sys.argv) > 1 else "gx1"
ndte = int(sys.argv[2]) if len(sys.argv) > 2 else 120
spec = synth.GRIDS[wl]
scal = synth.evp_scalars(120)
dc, geo, fields, tm, um = synth_case(wl, "full", seed=1, warm=True)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep, testing=True)
core.upload(fields, tm, um)
for _ in range(3): core.subcycle(ndte)
core.sync()
p = core.debug_prof().astype(np.float64)
tt = core.timings()
core.finalize()
used = p[:, :, :5].sum(axis=(1, 2)) > 0
p = p[used]
ph = p[:, :, :5] / ndte                     # cycles per subcycle
names = ["poll", "stress", "wait B1", "stepu+publish", "wait B2"]
nact = p[:, 0, 7].astype(int)
word5 = p[:, 0, 5].astype(np.int64)
rank = (word5 & 255).astype(int)
cu = ((word5 >> 8) & 0xffff).astype(int)
simd = ((p[:, :, 5].astype(np.int64) >> 24) & 3).astype(int)        # [tile][chunk]: SIMD of the wave that took the chunk
print("RESULT", wl, "tiles", len(p), "us/subcycle (event)", 1e3 * tt["loop_ms"] / ndte)
def show(sel, what):
    if not sel.any(): return
    late, early = ph[sel][:, 0, :], ph[sel][:, 1:, :].reshape(-1, 5)
    print(f"  {what}: {int(sel.sum())} tiles; total cycles/subcycle rim wave {late.sum(axis=1).mean():.0f}")
    print("     rim wave      " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, late.mean(axis=0))))
    print("     other waves   " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, early.mean(axis=0))))
show(nact == 4, "full tiles")
for r in range(3):
    show((nact == 4) & (rank == r), f"full tiles, arrival rank {r} on their CU")
show((nact > 0) & (nact < 4), "partial tiles")
show(nact == 0, "ice-free tiles")

# per CU: how many tiles it holds, and how long a subcycle of its tiles takes (cycles; all phases of the rim wave)
tot = ph[:, 0, :].sum(axis=1)
work = ph[:, :, 1].sum(axis=1) + ph[:, :, 3].sum(axis=1)        # stress + step cycles over the four waves of a tile
ncu = {c: int((cu == c).sum()) for c in np.unique(cu)}
per_cu = np.array([ncu[c] for c in cu])
for n in sorted(set(per_cu)):
    sel = per_cu == n
    print(f"  CUs holding {n} tiles: {len(set(cu[sel]))} CUs, {int(sel.sum())} tiles; rim-wave cycles/subcycle mean {tot[sel].mean():.0f} max {tot[sel].max():.0f}; "
          f"poll mean {ph[sel][:, 0, 0].mean():.0f} min {ph[sel][:, 0, 0].min():.0f}; active waves per CU mean {np.mean([nact[cu == c].sum() for c in set(cu[sel])]):.1f} "
          f"max {max(nact[cu == c].sum() for c in set(cu[sel]))}")
order = np.argsort(ph[:, 0, 0])[:8]
print("  tiles with the shortest poll (the ones everybody waits for):")
for t in order:
    c = cu[t]
    print(f"    tile cu {c:4d} rank {rank[t]} nact {nact[t]} tiles on CU {ncu[c]} (nact {list(nact[cu == c])}): " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, ph[t, 0, :])))

# do rim waves (chunk 0) of the tiles of one CU share a SIMD, and does it matter?
share = np.zeros(len(p), bool)
for c in np.unique(cu):
    idx = np.nonzero(cu == c)[0]
    act = [t for t in idx if nact[t] > 0]
    for t in act:
        share[t] = any(simd[o, 0] == simd[t, 0] for o in act if o != t)
full = nact == 4
for lab, sel in (("rim wave alone on its SIMD among rim waves", full & ~share), ("rim wave shares its SIMD with another tile's rim wave", full & share)):
    if sel.any():
        print(f"  {lab}: {int(sel.sum())} tiles: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, ph[sel][:, 0, :].mean(axis=0))))
# how many ice-holding chunks sit on the rim wave's SIMD (all tiles of the CU)
for nshare in range(1, 5):
    sel = np.zeros(len(p), bool)
    for t in np.nonzero(full)[0]:
        idx = np.nonzero(cu == cu[t])[0]
        n = sum(int(simd[o, ch] == simd[t, 0]) for o in idx for ch in range(int(nact[o])))
        sel[t] = n == nshare
    if sel.any():
        print(f"  {nshare} ice-holding waves on the rim wave's SIMD: {int(sel.sum())} tiles: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, ph[sel][:, 0, :].mean(axis=0))))
