#!/usr/bin/env python3
"""The fused MLP kernel's tile partition on the headline workload (257^3, icon prior, reference cmap mode): for every
(pool permille, group) the MLP stage time (HIP events, median of --reps launches, interleaved round robin so that clock drift
hits every setting alike), the spans of the 256 workgroups, the clock of every XCD and the tail (kernel - median span).
    python tools/steal_probe.py [--res 257] [--reps 12] > profiles/r06_steal_probe.txt"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icon_amd import synth  # noqa: E402
from icon_amd.engine import IconQueryEngine  # noqa: E402


def wg_summary(rec, kernel_ms):
    span = rec[:, 2]
    xcd = rec[:, 0].astype(int)
    mhz = rec[:, 3] / rec[:, 2] * 1e-3
    end = rec[:, 1] + rec[:, 2]
    return {"wg_span_ms": {"min": float(span.min()), "median": float(np.median(span)), "max": float(span.max())},
            "last_end_ms": float(end.max()), "first_end_ms": float(end.min()),
            "per_xcd_clock_mhz": [float(mhz[xcd == x].mean()) if (xcd == x).any() else None for x in range(8)],
            "per_xcd_span_ms": [float(span[xcd == x].mean()) if (xcd == x).any() else None for x in range(8)],
            "tiles": {"min": int(rec[:, 4].min()), "max": int(rec[:, 4].max())},
            "tail_ms": float(kernel_ms - np.median(span))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=257)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--settings", default="0:1,50:2,100:1,100:2,100:4,100:8,200:2,200:8,400:4,1000:4,1000:16,1000:64")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = synth.make_assets("body")
    T = lambda x: torch.from_numpy(x).to(dev)
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
    eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
    feat = T(a.features)
    res = args.res
    out = torch.empty((res, res, res), device=dev)
    settings = [tuple(int(v) for v in s.split(":")) for s in args.settings.split(",")]
    w = eng._work()
    for _ in range(5):
        eng.eval_slab(feat, res, 0, res, out=out)
    w.profile(True)
    ms = {s: [] for s in settings}
    step = {s: [] for s in settings}
    last = {}
    for rep in range(args.reps):
        for s in settings:
            w.set_steal(*s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.eval_slab(feat, res, 0, res, out=out)
            e1.record()
            st = w.stage_ms()
            torch.cuda.synchronize()
            ms[s].append(st[2])
            step[s].append(e0.elapsed_time(e1))
            last[s] = wg_summary(w.profile_workgroups(), st[2])
    for s in settings:
        line = {"permille": s[0], "group": s[1], "mlp_stage_ms": {"median": float(np.median(ms[s])), "min": float(np.min(ms[s])), "max": float(np.max(ms[s]))},
                "step_ms_median": float(np.median(step[s])), **last[s]}
        print(json.dumps(line))


if __name__ == "__main__":
    main()
