"""Timing of the remote-halo transports on ONE GPU (CICE_EVP_HIP_SELF_EXCHANGE routes every
inter-block ghost copy through the exchange with the rank itself).  Usage:
  CICE_EVP_HIP_SELF_EXCHANGE=1 CICE_EVP_HIP_HALO=direct|rccl python tools/selfx_timing.py [gx1|s01] [bx by]"""
import sys, os, time
import pathlib; R=str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0]=[R, R+'/tests', R+'/oracle']
import numpy as np
from cice_amd import evp, synth, decomp
from test_gpu_parity import synth_case
wl = sys.argv[1] if len(sys.argv) > 1 else "gx1"
bs = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else {"gx3": (100, 116), "gx1": (320, 192), "s01": (1800, 1200), "q8": (720, 270), "q4": (720, 540), "p2": (300, 240)}[wl]
ndte = {"gx3": 120, "gx1": 120, "s01": 48, "q8": 120, "q4": 120, "p2": 120}[wl]
scal = synth.evp_scalars(120)
dc, geo, fields, tm, um = synth_case(wl, "full", seed=1, warm=True, bs=bs)
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep, testing=True)
if os.environ.get("CICE_EVP_HIP_SELF_EXCHANGE"):
    core.comm_init(core.comm_unique_id())
core.upload(fields, tm, um)
for _ in range(2): core.subcycle(ndte)
core.sync()
t0=time.perf_counter()
for _ in range(10): core.subcycle(ndte)
core.sync()
t=time.perf_counter()-t0
out=core.download()
tt=core.timings()
print("RESULT", wl, bs, 'us/subcycle %.2f' % (1e6*t/(10*ndte)), 'checksum', float(np.abs(out['uvel']).sum()), tt['halo_transport'], tt['launches_per_subcycle'], tt['tile_variant'])
if os.environ.get("EVP_SHOW_CULOAD"):
    a = core.debug_cuload()
    used = a[a[:, 1] != 0][:, 2:6]
    print("CULOAD cus", len(used), "waves/SIMD histogram", np.bincount(used.ravel(), minlength=5).tolist(),
          "CUs by max SIMD load", np.bincount(used.max(axis=1), minlength=5).tolist())
core.finalize()
