#!/usr/bin/env python
"""bench.py - BASELINE.json's headline metric on MI355X: query-points/sec of one dense 257^3
("256^3") occupancy-lattice evaluation for the icon-filter configuration.

A "step" is one full reconEngine forward for one image: lattice generation, nearest-triangle /
inside queries against the SMPL-size body (V=6,890 / F=13,776), barycentric attributes, outlier
clipping (reference cmap semantics), bilinear feature gather + front/back select, the fused
13->512->256->128->1 MLP, the in_cube mask and the [D,H,W] volume write - plus, for N > 1, the
RCCL exchange of the outlier sign lists and the all_gather of the Z-slabs.  Per-image constants
(feature planes, packed mesh + BVH, folded weights) are resident in HBM before the timed region;
their one-off preparation time is reported separately in config.prep_ms.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--res 257] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement; roofline + cpu_baseline objects
described in DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

# two OpenMP runtimes live in this process during the cpu_baseline leg (torch's and the oracle's):
# make idle workers sleep instead of spinning against each other
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MLP_FLOP_PER_POINT = 344_602          # 2 * (13*512 + 512*256 + 269*128 + 141), SURVEY.md §8(d)
# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks for the instruction each path issues
PEAK_TFLOPS = {"f32": 157.3,          # v_mfma_f32_32x32x2_f32
               "f16x3": 2500.0,       # v_mfma_f32_32x32x16_f16; 3 MFMA products per algorithmic MAC
               "mx6": 2500.0}         # same f16 peak; 4 f16 + 2 fp6 MFMAs per K=64 (1.5 issue slots per K=16)
KERNEL = {"f32": "k_mlp_f32", "f16x3": "k_mlp_f16x3", "mx6": "k_mlp_mx6"}
DTYPE = {"f32": "f32", "f16x3": "f32 via 3x f16 split MFMA (f32 accumulate)",
         "mx6": "f32 via f16 MFMA + block-scaled fp6 cross terms (f32 accumulate)"}


def cpu_baseline(assets, res, budget_s=14.0):
    """Reference query path on the host cores: oracle/query_torch.py (the reference's torch
    operators, leaves from oracle/icon_oracle.c with OpenMP), timed on whole z-planes of the same
    lattice.  Sample size is calibrated to ~budget_s seconds of CPU work."""
    import numpy as np
    import torch
    from icon_amd import synth
    from oracle import oracle as orc, query_torch as qt

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    mlp = qt.build_mlp(assets.state_dict)
    mid = res // 2
    probe = synth.lattice_points(res, mid, mid + 1)[: 16384]
    # the port runs on every host core by default; on a many-core box the brute-force leaves can
    # be faster with fewer OpenMP threads: probe a few counts and keep the best (reported as "cores")
    torch.set_num_threads(min(cores, 64))
    v, f = assets.smpl_verts[0], assets.smpl_faces[0]
    best_rate, threads = 0.0, cores
    for n in sorted({c for c in (cores, cores // 2, cores // 4, 64, 32, 16) if 1 <= c <= cores}, reverse=True):
        orc.set_num_threads(n)
        t0 = time.perf_counter()
        orc.nearest_brute(v, f, probe)
        r = len(probe) / (time.perf_counter() - t0)
        if r > best_rate:
            best_rate, threads = r, n
    orc.set_num_threads(threads)
    t0 = time.perf_counter()
    qt.query(assets, mlp, probe, assets.sdf_clip)
    best_rate = len(probe) / (time.perf_counter() - t0)
    rate = best_rate
    planes = int(max(1, min(res, (rate * budget_s) // (res * res))))
    zs = np.unique(np.linspace(res // 8, res - 1 - res // 8, planes).round().astype(int))
    pts = np.concatenate([synth.lattice_points(res, int(z), int(z) + 1) for z in zs])
    qt.TIMES["leaves"] = 0.0
    t0 = time.perf_counter()
    qt.query(assets, mlp, pts, assets.sdf_clip)
    dt = time.perf_counter() - t0
    return {"value": len(pts) / dt, "unit": "points/s", "cores": threads, "host_cores": cores, "kind": "port",
            "seconds": {"total": dt, "leaves_c_openmp": qt.TIMES["leaves"], "torch_ops": dt - qt.TIMES["leaves"]},
            "sample": f"{len(zs)} whole z-planes of the {res}^3 lattice ({len(pts)} points, {dt:.1f} s), "
                      f"oracle/query_torch.py: torch-CPU operators of the reference ({torch.get_num_threads()} threads) + "
                      f"brute-force C leaves (OpenMP, {orc.num_threads()} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=257)
    ap.add_argument("--cmap-mode", default="reference", choices=["reference", "local"])
    ap.add_argument("--search", default="bvh", choices=["bvh", "brute"])
    ap.add_argument("--precision", default="mx6", choices=["f32", "f16x3", "mx6"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-adaptive", action="store_true",
                    help="also time the reference's coarse-to-fine schedule (extra small launches: keep it off when profiling)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from icon_amd import synth, _lib
    from icon_amd.engine import IconQueryEngine, query_func
    from icon_amd.recon import DenseReconEngine
    from types import SimpleNamespace

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    _lib.require_device()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    res = args.res
    a = synth.make_assets("body")
    T = lambda x: torch.from_numpy(x).to(dev)
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip, cmap_mode=args.cmap_mode, search=args.search,
                          precision=args.precision)
    eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
    feats = [T(a.features)]
    recon = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                             resolutions=[33, 65, 129, res] if res == 257 else [res], align_corners=True,
                             balance_value=0.5, faster=True, engine=eng).to(dev)
    opt = SimpleNamespace(num_views=1)

    def step():
        return recon(opt=opt, netG=eng, features=feats, proj_matrix=None)

    # one-off per-image preparation (BVH build, plane repack, BatchNorm fold + operand packing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng._mesh_handle(); eng._feat_handle(feats[0]); eng._mlp_handle()
    torch.cuda.synchronize()
    prep_ms = (time.perf_counter() - t0) * 1e3

    for _ in range(args.warmup):
        step()
    eng._work().profile(True)
    stage = np.zeros(3)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        occ = step()
        # stage_ms waits on this step's last event (the MLP) - the same point the reference's
        # `(occupancys > 0.5).sum() == 0` check already synchronises on
        stage += np.array(eng._work().stage_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage /= max(args.steps, 1)
    assert occ is not None and occ.shape == (res, res, res)

    n_points = res ** 3
    my_points = n_points if world == 1 else (lambda z: (z[1] - z[0]) * res * res)(
        __import__("icon_amd.recon", fromlist=["slab_bounds"]).slab_bounds(res, world, rank))
    value = n_points * args.steps / elapsed
    mlp_s = stage[2] * 1e-3
    achieved = (MLP_FLOP_PER_POINT * my_points / mlp_s) / 1e12 if mlp_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(KERNEL[args.precision] + "_bytes_per_launch")
        except Exception:
            traffic = None

    # informational: the reference's own coarse-to-fine schedule (Seg3dLossless._forward_faster, ~1 % of the
    # lattice queried) on the same engine - not the metric, which is the dense grid
    adaptive_ms = None
    if args.with_adaptive and world == 1 and res == 257:
        from icon_amd.recon import AdaptiveReconEngine
        ad = AdaptiveReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                 resolutions=[33, 65, 129, res], align_corners=True).to(dev)
        eng._work().profile(False)
        for _ in range(2):
            ad(opt=opt, netG=eng, features=feats, proj_matrix=None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            ad(opt=opt, netG=eng, features=feats, proj_matrix=None)
        torch.cuda.synchronize()
        adaptive_ms = (time.perf_counter() - t1) / 5 * 1e3

    if rank == 0:
        out = {
            "metric": "query-points/sec at 256^3 grid", "value": value, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic",
            "config": {
                "workload": f"icon-filter.yaml, {res}^3 lattice (mcube_res={res - 1}), 1 image: SMPL-size body "
                            f"V=6890/F=13776, planes [1,12,128,128], MLP 13-512-256-128-1, cmap_mode={args.cmap_mode}, mlp={args.precision}",
                "parallelism": f"zslab{world}", "points_per_step": n_points, "prep_ms": prep_ms,
                "stage_ms": {"features": stage[0], "cmap_patch": stage[1], "mlp": stage[2]},
                "reference_schedule_ms_per_volume": adaptive_ms,
            },
            "roofline": {"bound": "mfma", "kernel": KERNEL[args.precision], "achieved": achieved,
                         "peak": PEAK_TFLOPS[args.precision],
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": traffic,
                         "flop_per_launch": MLP_FLOP_PER_POINT * my_points, "avg_launch_ms": stage[2]},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a, res)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
