#!/usr/bin/env python3
"""Latency DISTRIBUTION of the reference's schedule as one native call (icon_adaptive_eval), every call on its own with the
stream drained in between: p50 / p99 / max over --calls calls for [33..257] and [33..513], with the Python garbage collector
frozen (gc.freeze(), gc.disable() - INTEGRATION.md) and without.    python tools/schedule_latency.py > profiles/r06_schedule_latency.txt"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icon_amd import synth  # noqa: E402
from icon_amd.engine import IconQueryEngine  # noqa: E402


def distribution(eng, feat, sched, calls):
    for _ in range(10):
        eng.adaptive_eval(feat, sched)
    torch.cuda.synchronize()
    ts = np.empty(calls)
    for k in range(calls):
        t0 = time.perf_counter()
        eng.adaptive_eval(feat, sched)          # synchronises for its counts (the reference's own None test does)
        ts[k] = (time.perf_counter() - t0) * 1e3
    return {"p50": float(np.percentile(ts, 50)), "p90": float(np.percentile(ts, 90)), "p99": float(np.percentile(ts, 99)), "max": float(ts.max()),
            "min": float(ts.min()), "calls": int(calls)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=400)
    args = ap.parse_args()
    a = synth.make_assets("body")
    T = lambda x: torch.from_numpy(x).cuda()
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
    eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
    feat = T(a.features)
    for frozen in (False, True):
        if frozen:
            gc.collect(); gc.freeze(); gc.disable()
        for sched in ([33, 65, 129, 257], [33, 65, 129, 257, 513]):
            d = distribution(eng, feat, sched, args.calls)
            print(json.dumps({"schedule": sched, "gc": "frozen + disabled" if frozen else "default", **d}), flush=True)


if __name__ == "__main__":
    main()
