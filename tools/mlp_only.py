"""Launch only the MLP kernel on 257^3 worth of realistic rows (for rocprofv3 --pmc / timing on the GPU box).
usage: mlp_only.py <precision> [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icon_amd import synth
from icon_amd.engine import MlpHandle

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
a = synth.make_assets("body")
mlp = MlpHandle({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
N = 257 ** 3
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.zeros((N, 16), device=dev)
x[:, :13] = torch.randn((N, 13), device=dev, generator=g)
y = mlp.forward(x, precision=prec); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n_launch): y = mlp.forward(x, precision=prec)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n_launch
print(f"{prec}: {ms:.3f} ms per launch, {344602 * N / ms / 1e9:.1f} TFLOP/s algorithmic")
