"""Is k_mlp_f16x3 clock/power bound?  Same kernel, same launch, three data sets: the synthetic checkpoint on N(0,1)
rows, the same checkpoint on all-zero rows, and an all-zero checkpoint on all-zero rows (no operand bit toggles
in the matrix pipe).  MI355X_MICROARCH.md (DVFS give-back): identical instruction streams run faster on quiet data
because the chip clocks to its power budget.  usage: mlp_power_probe.py [precision] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icon_amd import synth
from icon_amd.engine import MlpHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
sd = synth.make_mlp_state_dict()
zero = {k: (np.ones_like(v) if k.endswith("running_var") else np.zeros_like(v)) for k, v in sd.items()}
N = 257 ** 3
g = torch.Generator(device=dev); g.manual_seed(1)
xr = torch.zeros((N, 16), device=dev); xr[:, :13] = torch.randn((N, 13), device=dev, generator=g)
xz = torch.zeros((N, 16), device=dev)
def run(tag, state, x):
    mlp = MlpHandle({k: torch.from_numpy(v) for k, v in state.items()})
    mlp.forward(x, precision=prec); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n_launch): mlp.forward(x, precision=prec)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n_launch
    print(f"{prec} {tag:34s} {ms:7.3f} ms  {344602 * N / ms / 1e9:7.1f} TFLOP/s algorithmic")
run("random weights, N(0,1) rows", sd, xr)
run("random weights, zero rows", sd, xz)
run("zero weights, zero rows", zero, xz)
run("random weights, N(0,1) rows (again)", sd, xr)
