#!/usr/bin/env python3
"""Stress of the fused MLP kernel's tile partition (XCD-local pools): N dense launches with random (pool share, group size,
reserved CUs, resolution) - every volume must equal the all-static one bit for bit, and the nine counters must be clean after
every launch (the next launch's correctness depends on it).    N=300 python tools/stress_steal.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icon_amd import synth  # noqa: E402
from icon_amd.engine import IconQueryEngine  # noqa: E402


def main():
    n = int(os.environ.get("N", "300"))
    dev = torch.device("cuda:0")
    a = synth.make_assets("body")
    T = lambda x: torch.from_numpy(x).to(dev)
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
    eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
    feat = T(a.features)
    w = eng._work()
    rng = np.random.RandomState(int(os.environ.get("SEED", "1")))
    want = {}
    for res in (129, 161, 257):
        w.set_steal(0, 1)
        want[res] = eng.eval_slab(feat, res, 0, res).clone()
    bad = 0
    for k in range(n):
        res = int(rng.choice([129, 161, 257], p=[0.4, 0.3, 0.3]))
        permille = int(rng.choice([1, 30, 100, 150, 250, 500, 900, 1000]))
        group = int(rng.choice([1, 2, 3, 5, 8, 32, 127]))
        reserve = int(rng.choice([0, 0, 0, 8, 16, 100, 250]))
        w.set_steal(permille, group)
        w.set_reserve_cus(reserve)
        out = torch.full((res, res, res), float("nan"), device=dev)
        eng.eval_slab(feat, res, 0, res, out=out)
        if not torch.equal(out.view(torch.int32), want[res].view(torch.int32)):
            bad += 1
            print(f"MISMATCH at launch {k}: res {res} permille {permille} group {group} reserve {reserve}: "
                  f"{int((out != want[res]).sum())} voxels differ, {int(torch.isnan(out).sum())} never written", flush=True)
    w.set_reserve_cus(0); w.set_steal(150, 2)
    print(f"stress_steal {'ok' if bad == 0 else 'FAILED'}: {n} launches, {bad} mismatches; box {open('/proc/sys/kernel/random/boot_id').read().strip()}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
