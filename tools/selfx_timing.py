import sys, os, time
import pathlib; R=str(pathlib.Path(__file__).resolve().parents[1]); sys.path[:0]=[R, R+'/tests', R+'/oracle']
import numpy as np
from cice_amd import evp, synth, decomp
from test_gpu_parity import synth_case
scal = synth.evp_scalars(120)
dc, geo, fields, tm, um = synth_case("gx1", "full", seed=1, warm=True, bs=(320,192))
d, keep = evp.make_dims(dc, 0)
core = evp.EvpHip(d, evp.make_params(scal, strict=True), geo["HTE"], geo["HTN"], geo["dxT"], geo["dyT"], geo["uarear"], geo["tarea"], keepalive=keep)
if os.environ.get("CICE_EVP_HIP_SELF_EXCHANGE"):
    core.comm_init(core.comm_unique_id())
core.upload(fields, tm, um)
for _ in range(2): core.subcycle(120)
core.sync()
t0=time.perf_counter()
for _ in range(10): core.subcycle(120)
core.sync()
t=time.perf_counter()-t0
out=core.download()
print("RESULT", os.environ.get("CICE_EVP_HIP_SELF_EXCHANGE"), os.environ.get("CICE_EVP_HIP_GRAPH_RCCL"), 'us/subcycle', 1e6*t/1200, 'checksum', float(np.abs(out['uvel']).sum()), core.timings())
core.finalize()
