#!/bin/bash
# Sample board power and clocks (rocm-smi) while the dense 257^3 step runs back to back; evidence for DESIGN.md section 4.4.
#   usage (GPU box): bash tools/power_trace.sh [precision]
P=${1:-f16x3}
python - <<PY &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
dev = torch.device("cuda:0")
a = synth.make_assets("body")
T = lambda x: torch.from_numpy(x).to(dev)
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip, precision="$P")
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features)
out = torch.empty((257, 257, 257), device=dev)
t0 = time.time()
n = 0
while time.time() - t0 < 12.0:
    for _ in range(20): eng.eval_slab(feat, 257, 0, 257, out=out)
    torch.cuda.synchronize(); n += 20
print(f"$P: {n} volumes in {time.time() - t0:.1f} s = {(time.time() - t0) / n * 1e3:.2f} ms / volume")
PY
PID=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | tr -s ' ' | head -8
  echo "--"
  sleep 1
done
wait $PID
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
