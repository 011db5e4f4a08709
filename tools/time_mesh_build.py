"""per-stage timing of the device mesh build (ICON_AMD_DEBUG_SYNC=1 prints every stage) and the end-to-end figure
   ICON_AMD_DEBUG_SYNC=1 python tools/time_mesh_build.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import MeshHandle
a = synth.make_assets("body")
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(x).to(dev)
args = (T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
torch.cuda.synchronize()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print(f"--- build {k}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    h = MeshHandle(*args, validate=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"build {k}: host {1e3 * (t1 - t0):.3f} ms, until done {1e3 * (t2 - t0):.3f} ms, stats {h.stats()}", flush=True)
