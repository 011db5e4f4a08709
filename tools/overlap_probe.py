#!/usr/bin/env python3
"""Can the nearest-triangle search of one volume run BESIDE the fused MLP kernel of another on the CUs the MLP kernel
leaves free (icon_work_set_reserve_cus)?  The MLP kernel takes a CU whole (132 KiB LDS, every vector register), so a search
wave can only land on a reserved CU; the MLP kernel is power-bound (its clock rises when CUs are taken away), the search is
VALU-issue-bound.  For R reserved CUs: the MLP kernel alone, the geometry pass (search + sign codes + outlier scan) alone on the
whole chip, and both at once on two streams (MLP launched first) - combined vs summed time.
    python tools/overlap_probe.py > profiles/r06_overlap_probe.txt"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=257)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--reserve", default="0,16,32,48,64,96")
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
    n = res ** 3
    msg = [torch.empty((8 + (n + 3) // 4 + 7) // 8 * 8, dtype=torch.uint8, device=dev) for _ in range(2)]
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def geometry(work):
        eng.slab_features(feat, res, 0, res, msg=msg[work], work=work)

    def mlp(work):
        eng.slab_finish_gathered(res, 0, res, msg[work], msg[work].numel(), 1, 0, out=out, work=work)

    def timed(fa, fb):
        """fa on stream A, then fb on stream B, both after a common start event; ms until A, B and both are done"""
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        eA, eB = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        sA.wait_event(e0); sB.wait_event(e0)
        if fa:
            with torch.cuda.stream(sA):
                fa()
        eA.record(sA)
        if fb:
            with torch.cuda.stream(sB):
                fb()
        eB.record(sB)
        torch.cuda.synchronize()
        ta, tb = e0.elapsed_time(eA), e0.elapsed_time(eB)
        return ta, tb, max(ta, tb)

    with torch.cuda.stream(sA):
        geometry(0); mlp(0)
    with torch.cuda.stream(sB):
        geometry(1); mlp(1)
    torch.cuda.synchronize()
    med = lambda xs: float(np.median(xs))
    for R in [int(v) for v in args.reserve.split(",")]:
        eng._work(0).set_reserve_cus(R)
        with torch.cuda.stream(sA):
            geometry(0)
        torch.cuda.synchronize()
        alone_mlp, alone_geo, both, both_a, both_b = [], [], [], [], []
        for _ in range(args.reps):
            alone_mlp.append(timed(lambda: mlp(0), None)[0])
            alone_geo.append(timed(None, lambda: geometry(1))[1])
            ta, tb, tt = timed(lambda: mlp(0), lambda: geometry(1))
            both.append(tt); both_a.append(ta); both_b.append(tb)
        line = {"reserve_cus": R, "mlp_alone_ms": med(alone_mlp), "geometry_alone_ms": med(alone_geo), "sum_ms": med(alone_mlp) + med(alone_geo),
                "combined_ms": med(both), "combined_mlp_done_ms": med(both_a), "combined_geometry_done_ms": med(both_b),
                "gain_ms": med(alone_mlp) + med(alone_geo) - med(both)}
        print(json.dumps(line), flush=True)
    eng._work(0).set_reserve_cus(0)


if __name__ == "__main__":
    main()
