"""repro of test_lattice_slab_split_gathered_messages[5] with a synchronisation after every call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_parity import make_engine, T, dev
from common import assets
world = int(sys.argv[1]) if len(sys.argv) > 1 else 5
body = assets("body")
res = 33
feat = T(body.features)
full = make_engine(body).eval_slab(feat, res, 0, res); torch.cuda.synchronize(); print("full ok", flush=True)
per = -(-res // world)
bounds = [(min(r * per, res), min(r * per + per, res)) for r in range(world)]
stride = 8 + (per * res * res + 7) // 8 * 8
engines = [make_engine(body) for _ in range(world)]
msgs = []
for r, e in enumerate(engines):
    z0, z1 = bounds[r]
    msg = torch.full((stride,), 77, dtype=torch.int8, device=dev()); msg[:8] = 0
    n = (z1 - z0) * res * res
    e.slab_features(feat, res, z0, z1, signs=msg[8:8 + n], count=msg[:8].view(torch.int64))
    torch.cuda.synchronize(); print("features", r, z0, z1, int(msg[:8].view(torch.int64)[0]), flush=True)
    msgs.append(msg)
gathered = torch.cat(msgs).contiguous()
parts = []
for r, e in enumerate(engines):
    z0, z1 = bounds[r]
    parts.append(e.slab_finish_gathered(res, z0, z1, gathered, stride, world, r))
    torch.cuda.synchronize(); print("finish", r, flush=True)
print("equal", torch.equal(torch.cat(parts), full))
