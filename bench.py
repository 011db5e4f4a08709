#!/usr/bin/env python
"""bench.py - BASELINE.json's headline metric on MI355X: query-points/sec of one dense 257^3
("256^3") occupancy-lattice evaluation for the icon-filter configuration, plus the second half of
the metric (mesh Chamfer vs the reference's schedule) and a live parity sample against the checker.

A "step" is one full reconEngine forward for one image: lattice generation, nearest-triangle /
inside queries against the SMPL-size body (V=6,890 / F=13,776), barycentric attributes, outlier
clipping (reference cmap semantics), bilinear feature gather + front/back select, the fused
13->512->256->128->1 MLP (f32-class arithmetic by default), the in_cube mask and the [D,H,W] volume
write - plus, for N > 1, the RCCL exchange of the outlier sign lists and the all_gather of the
Z-slabs (--replicas: one image per GPU instead, whole volumes, no data-path collective, "weak" scaling -
BASELINE.json configs[4] with --res 513).  Per-image constants (feature planes, packed mesh + BVH, folded weights) are resident in
HBM before the timed region; their one-off preparation time is reported in config.prep_ms.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--res 257] [--prior icon|pamir]
                    [--precision f16x3|f32] [--replicas] [--no-cpu-baseline] [--no-extras]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement; roofline + cpu_baseline objects and
the extra config entries are described in DESIGN.md "Measurement").  --no-extras skips everything
after the timed region (use it under rocprofv3 so the trace holds only the dense step).
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
ALGO_BYTES_PER_POINT = 4              # one fp32 occupancy write, lattice generated in-kernel
# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks for the instruction each path issues
PEAK_TFLOPS = {"f32": 157.3,          # v_mfma_f32_32x32x2_f32
               "f16x3": 2500.0}       # v_mfma_f32_32x32x16_f16; 3 MFMA products per algorithmic MAC
KERNEL = {"f32": "k_mlp_f32", "f16x3": "k_fused_f16x3"}   # f16x3: features + MLP in one kernel
DTYPE = {"f32": "f32", "f16x3": "f32 via 3x f16 split MFMA (22-bit operands, f32 accumulate)"}


def host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(assets, res, budget_s=14.0):
    """The reference's query path on the host cores, same lattice, bounded sample.

    kind "reference": the reference's OWN query_func -> HGPIFuNet.query -> cal_sdf_batch -> MLP, imported
    verbatim from /root/reference (oracle/ref_loader.py) - only where that tree exists (the build
    container); kind "port": oracle/query_torch.py, the same torch-CPU operators restated (the GPU box).
    Either way the two O(N*F) third-party leaves (kaolin nearest triangle / check_sign, absent upstream on
    CPU) are the checker's EXACT BVH / ray-bin versions (oracle/icon_accel.c), as SURVEY.md §8(d) asks.
    Thread policy is fixed: every core of the process's affinity mask for torch and for OpenMP; no probing."""
    import numpy as np
    import torch
    from icon_amd import synth
    from oracle import oracle as orc, query_torch as qt, ref_loader

    cores = host_cores()
    torch.set_num_threads(cores)
    orc.set_num_threads(cores)
    mid = res // 2
    kind = "port"
    if ref_loader.available():
        try:
            ref = ref_loader.load()
            netG, cfg = ref_loader.build_netG(assets)
            feats = [torch.from_numpy(assets.features)]

            def run(pts):
                with torch.no_grad():
                    return ref.query_func(cfg, netG, feats, torch.from_numpy(pts)[None])[0, 0].numpy()
            kind = "reference"
        except Exception as e:                       # keep the bench line alive; say what happened
            print(f"cpu_baseline: reference import failed ({e!r}); timing the port", file=sys.stderr)
    if kind == "port":
        mlp = qt.build_mlp(assets.state_dict)

        def run(pts):
            return qt.query(assets, mlp, pts, assets.sdf_clip)
    probe = synth.lattice_points(res, mid, mid + 1)
    run(probe[:4096])                                 # first-touch / thread-pool start-up
    t0 = time.perf_counter()
    run(probe)
    rate = len(probe) / (time.perf_counter() - t0)
    planes = int(max(1, min(res, (rate * budget_s) // (res * res))))
    zs = np.unique(np.linspace(res // 8, res - 1 - res // 8, planes).round().astype(int))
    pts = np.concatenate([synth.lattice_points(res, int(z), int(z) + 1) for z in zs])
    qt.TIMES["leaves"] = 0.0
    t0 = time.perf_counter()
    run(pts)
    dt = time.perf_counter() - t0
    out = {"value": len(pts) / dt, "unit": "points/s", "cores": cores, "kind": kind,
           "sample": f"{len(zs)} whole z-planes of the {res}^3 lattice ({len(pts)} points, {dt:.1f} s): "
                     + ("the reference's own query_func/HGPIFuNet.query/MLP run verbatim from /root/reference"
                        if kind == "reference" else "oracle/query_torch.py (the reference's torch-CPU operators restated)")
                     + f", torch {torch.__version__} on {torch.get_num_threads()} threads; kaolin leaves = exact BVH nearest + "
                       f"binned ray parity in C/OpenMP ({orc.num_threads()} threads)"}
    if kind == "port":
        out["seconds"] = {"total": dt, "leaves_c_openmp": qt.TIMES["leaves"], "torch_ops": dt - qt.TIMES["leaves"]}
    return out


def parity_sample(assets, res, occ, cmap_mode, eng=None, n_each=12288, seed=1993):
    """max / p99.9 |occ - checker| on a stratified sample of the lattice: uniform points, points around the 0.5
    level set (where the mesh comes from) and shell points.  The checker evaluates the geometry half on the
    WHOLE lattice (the tiled outlier-cmap rule needs every sign) and the MLP in float64 on the sample.
    With `eng`: the errors split by TIE stratum - a point is "tied" when the runner-up triangle's d^2 lies within
    one float32 ulp of the winner's (icon_sdf_query_ties), i.e. its norm / cmap / vis hang on the last bit of the
    unpinned kaolin leaf (lib/dataset/mesh_util.py:374-390).  Within ONE definition of the leaf (checker == HIP)
    both strata meet the tolerance; only the untied stratum is comparable point by point with a kaolin run."""
    import numpy as np
    import torch
    from icon_amd import synth
    from oracle import oracle as orc
    rng = np.random.RandomState(seed)
    flat = occ.reshape(-1)
    n = flat.numel()
    uni = rng.randint(0, n, n_each)
    band = torch.nonzero((flat > 0.2) & (flat < 0.8)).reshape(-1).cpu().numpy()
    lvl = band[rng.randint(0, len(band), min(n_each, len(band)))] if len(band) else uni[:0]
    k = rng.randint(0, res, (n_each // 4, 3))
    k[np.arange(len(k)), rng.randint(0, 3, len(k))] = rng.choice([0, res - 1], len(k))
    shell = (k[:, 2] * res + k[:, 1]) * res + k[:, 0]
    idx = np.unique(np.concatenate([uni, lvl, shell])).astype(np.int64)
    t0 = time.perf_counter()
    pts = synth.lattice_points(res)
    ref, _ = orc.query_icon_subset(assets.smpl_verts[0], assets.smpl_faces[0], assets.smpl_cmap[0], assets.smpl_vis[0],
                                   assets.features, orc.Mlp(assets.state_dict), pts, idx, sdf_clip=assets.sdf_clip,
                                   f64=True, cmap_local=(cmap_mode == "local"))
    got = flat[torch.from_numpy(idx).to(flat.device)].cpu().numpy()
    err = np.abs(got - ref)
    out = {"n": int(len(idx)), "max_abs": float(err.max()), "p999_abs": float(np.quantile(err, 0.999)),
           "mean_abs": float(err.mean()), "tolerance": 1e-4, "within": bool(err.max() <= 1e-4),
           "strata": {"uniform": int(len(uni)), "level_set_band": int(len(lvl)), "shell": int(len(shell))},
           "checker": "oracle/icon_oracle.c, geometry on all %d lattice points, float64 MLP on the sample" % n}
    if eng is not None:
        mesh = eng._mesh_handle()
        dev = flat.device
        ulps = mesh.sdf_query_ties(torch.from_numpy(pts[idx]).to(dev))["ulps"].cpu().numpy()
        tied = ulps <= 1
        out["max_abs_untied"] = float(err[~tied].max()) if (~tied).any() else None
        out["max_abs_tied"] = float(err[tied].max()) if tied.any() else None
        out["frac_tied"] = float(tied.mean())
        band_s = np.isin(idx, lvl)
        out["frac_tied_level_set_band"] = float(tied[band_s].mean()) if band_s.any() else None
        # the whole lattice: how much of it is tie-sensitive at all (histogram of the ulp gap to the runner-up)
        ul = mesh.sdf_query_ties(torch.from_numpy(pts).to(dev))["ulps"]
        edges = [0, 1, 2, 5, 17, 255, 256]
        hist = torch.histc(ul.float(), bins=256, min=0, max=256).cpu().numpy()
        out["tie_ulps_histogram_lattice"] = {f"{a}" if b == a + 1 else f"{a}-{b - 1}": int(hist[a:b].sum()) for a, b in zip(edges[:-1], edges[1:])}
        out["frac_tied_lattice"] = float((ul <= 1).float().mean().item())
        out["tie_definition"] = "runner-up face's d^2 within 1 float32 ulp of the winner's (255 = none within the search bound)"
    out["seconds"] = time.perf_counter() - t0
    return out


def mesh_vs_oracle(assets, res, occ, cmap_mode, band=0.05):
    """SURVEY.md section 8(d) metric (i): the mesh of the GPU volume against the mesh of the ORACLE's volume (float64 MLP),
    Chamfer / P2S as lib/dataset/Evaluator.py:200-230 defines them (x100, [-1,1]-cube units).  Marching cubes reads a lattice
    value only where it (a) ends a lattice edge whose end points lie on different sides of the level or (b) could change side:
    the oracle is evaluated on exactly those voxels - every end point of a crossing edge of the GPU volume plus every voxel
    within `band` of the level (500 x the parity tolerance) - and spliced into a copy of the GPU volume; every other value of
    the two volumes is ASSUMED to be the same number - an assumption the GPU volume itself cannot vouch for (a surface the
    GPU misses altogether has no crossing edge in ITS volume), so the oracle is also evaluated OUTSIDE that set: on every
    voxel of the stride-4 sub-lattice (any region the GPU gets wrong that is 4 voxels across in each axis holds one) and on
    100,000 random voxels; `within` requires that none of them lies on the other side of the level or differs by more
    than the parity tolerance.  They are spliced in as well; both volumes then go through the same marching cubes."""
    import numpy as np
    import torch
    from icon_amd import metrics, synth
    from icon_amd.recon import export_mesh_device
    from oracle import oracle as orc
    t0 = time.perf_counter()
    ins = occ > 0.5
    need = (occ - 0.5).abs() < band
    for ax in range(3):
        a, b = [slice(None)] * 3, [slice(None)] * 3
        a[ax], b[ax] = slice(0, -1), slice(1, None)
        cross = ins[tuple(a)] != ins[tuple(b)]
        need[tuple(a)] |= cross
        need[tuple(b)] |= cross
    n_need = int(need.sum().item())
    # the independent part: voxels chosen WITHOUT looking at the GPU volume
    probe = torch.zeros_like(need)
    probe[::4, ::4, ::4] = True
    rng = np.random.RandomState(7)
    probe.view(-1)[torch.from_numpy(rng.randint(0, res ** 3, 100_000)).to(occ.device)] = True
    probe &= ~need
    idx = torch.nonzero((need | probe).reshape(-1)).reshape(-1)
    idx_h = idx.cpu().numpy().astype(np.int64)
    ref, _ = orc.query_icon_subset(assets.smpl_verts[0], assets.smpl_faces[0], assets.smpl_cmap[0], assets.smpl_vis[0],
                                   assets.features, orc.Mlp(assets.state_dict), synth.lattice_points(res), idx_h,
                                   sdf_clip=assets.sdf_clip, f64=True, cmap_local=(cmap_mode == "local"))
    t_oracle = time.perf_counter() - t0
    occ_o = occ.clone()
    occ_o.view(-1)[idx] = torch.from_numpy(ref).to(occ.device)
    err_all = (occ_o.view(-1)[idx] - occ.reshape(-1)[idx]).abs()
    is_probe = probe.reshape(-1)[idx]
    err = err_all[~is_probe]
    err_probe = err_all[is_probe]
    probe_side = int((((occ_o > 0.5) != ins) & probe).sum().item())
    probe_max = float(err_probe.max().item()) if err_probe.numel() else 0.0
    vg, fg = export_mesh_device(occ, 0.5)
    vo, fo = export_mesh_device(occ_o, 0.5)
    ch, p2s = metrics.chamfer_p2s(metrics.to_unit_cube(vg, res), fg, metrics.to_unit_cube(vo, res), fo, n=100_000)
    same_topology = bool(fg.shape == fo.shape and torch.equal(fg, fo))
    vmax = float((vg - vo).abs().max().item()) if vg.shape == vo.shape else None
    return {"chamfer_x100_dense_vs_oracle": ch, "p2s_x100_dense_vs_oracle": p2s, "tolerance_x100": 0.01,
            "within": bool(ch <= 0.01 and probe_side == 0 and probe_max <= 1e-4),
            "oracle_voxels": n_need, "max_abs_on_them": float(err.max().item()) if err.numel() else 0.0,
            "independent_voxels": int(is_probe.sum().item()), "independent_voxels_changing_side": probe_side, "independent_max_abs": probe_max,
            "independent_note": "voxels chosen without looking at the GPU volume (stride-4 sub-lattice + 100,000 random, minus the set above): "
                                "the oracle must agree with the GPU volume there for the splice to be the oracle's mesh",
            "voxels_changing_side": int(((occ_o > 0.5) != ins).sum().item()), "faces_gpu": int(fg.shape[0]), "faces_oracle": int(fo.shape[0]),
            "same_faces": same_topology, "max_vertex_move_voxels": vmax, "samples_per_mesh": 100_000,
            "dense_vs_oracle_seconds": time.perf_counter() - t0, "oracle_seconds": t_oracle,
            "dense_vs_oracle_note": "(i) oracle (oracle/icon_oracle.c, geometry on the whole lattice, float64 MLP) on every voxel marching cubes can see a "
                    f"difference through: end points of crossing edges + |occ - 0.5| < {band}; spliced into the GPU volume; same marching cubes on both"}


def mlp_zero_data(dev, n_points, launches=3):
    """The instruction-stream invariant of the MLP chain: the standalone k_mlp_f16x3 kernel (the fused kernel's chunk bodies)
    on all-zero rows with an all-zero checkpoint - no operand bit toggles in the matrix pipe, so the chip holds its clock
    (9.7 ms where real data takes 13.8).  Moves only when the CODE or the box's maximum clock does."""
    import numpy as np
    import torch
    from icon_amd import synth
    from icon_amd.engine import MlpHandle
    sd = synth.make_mlp_state_dict()
    zero = {k: (np.ones_like(v) if k.endswith("running_var") else np.zeros_like(v)) for k, v in sd.items()}
    mlp = MlpHandle({k: torch.from_numpy(v) for k, v in zero.items()})
    x = torch.zeros((n_points, 16), device=dev)
    mlp.forward(x, precision="f16x3")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        mlp.forward(x, precision="f16x3")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches


def tie_sensitivity(assets, res, occ, make_engine, step_with, ulps=1):
    """The same volume under the ALTERNATIVE tie rule (among the faces within `ulps` ulps of the minimum d^2 the
    highest index instead of the exact minimum / lowest index): how many lattice values, level-set voxels and how much
    of the mesh hang on the unpinned tie behaviour of kaolin's point_to_mesh_distance."""
    import torch
    from icon_amd import metrics
    from icon_amd.recon import export_mesh_device
    t0 = time.perf_counter()
    e2 = make_engine("f16x3")
    e2.tie_rule = ("highest", ulps)
    occ2 = step_with(e2)
    d = (occ2 - occ).abs()
    band = (occ - 0.5).abs() < 0.2
    out = {"rule": f"highest face index among the faces within {ulps} ulp of the minimum d^2 (default: lowest index, exact minimum)",
           "values_moved_gt_1e-4": int((d > 1e-4).sum().item()), "frac_moved_gt_1e-4": float((d > 1e-4).float().mean().item()),
           "max_move": float(d.max().item()),
           "level_set_band_voxels": int(band.sum().item()), "level_set_band_moved_gt_1e-4": int(((d > 1e-4) & band).sum().item()),
           "level_set_band_max_move": float(d[band].max().item()) if band.any() else 0.0,
           "voxels_changing_side_of_0.5": int(((occ > 0.5) != (occ2 > 0.5)).sum().item())}
    va, fa = export_mesh_device(occ, 0.5)
    vb, fb = export_mesh_device(occ2.contiguous(), 0.5)
    ch, p2s = metrics.chamfer_p2s(metrics.to_unit_cube(va, res), fa, metrics.to_unit_cube(vb, res), fb, n=100_000)
    out["mesh_chamfer_x100"] = ch
    out["mesh_p2s_x100"] = p2s
    out["voxel_x100"] = 2.0 / (res - 1) * 100.0
    out["seconds"] = time.perf_counter() - t0
    return out


def library_gemm_ceiling(dev, m=2_097_152, iters=8):
    """What the vendor library sustains on THIS box for the MLP's dominant GEMM shape, random N(0,1) operands (the matrix
    pipe's clock depends on the operand bits): layer 1 of the regressor as a plain f16 GEMM with f32
    accumulation - [points, 512] x [512, 256] through hipBLASLt / rocBLAS (torch.matmul) - and an 8192^3 GEMM as the
    library's best case.  An independent calibration of the ceiling k_fused_f16x3 is priced against: the kernel issues three
    f16 MFMA products per algorithmic MAC, so its ISSUED rate (3 x achieved) is what compares with these numbers."""
    import torch
    out = {}
    g = torch.Generator(device=dev).manual_seed(1993)
    for name, (mm, kk, nn) in {"layer1_points_x512_x256": (m, 512, 256), "square_8192": (8192, 8192, 8192)}.items():
        a = torch.randn((mm, kk), device=dev, dtype=torch.float16, generator=g)
        b = torch.randn((kk, nn), device=dev, dtype=torch.float16, generator=g)
        for _ in range(3):
            c = a @ b
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out[name] = {"tflops": 2.0 * mm * kk * nn / (ms * 1e-3) / 1e12, "ms": ms, "shape": [mm, kk, nn]}
        del a, b, c
    return out


def self_launch(n: int) -> int:
    """Run this very command line as n ranks of one node (torch.distributed.run, 127.0.0.1 rendezvous on a free
    port); the ranks' stdout / stderr pass through, so rank 0's JSON line is this process's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=257)
    ap.add_argument("--prior", default="icon", choices=["icon", "pamir"])
    ap.add_argument("--cmap-mode", default="reference", choices=["reference", "local"])
    ap.add_argument("--search", default="bvh", choices=["bvh", "brute"])
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"])
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: one image per GPU, every rank evaluates a whole volume, no data-path collective (BASELINE.json "
                         "configs[4]: 513^3 x 8 images); default: ONE image, Z-slabs sharded over the ranks (configs[2])")
    ap.add_argument("--mesh-exchange", action="store_true",
                    help="N > 1, sharded: the step ends in the MESH (DenseReconEngine.forward_mesh: every rank triangulates its own "
                         "Z-slab, keyed meshes are exchanged instead of the 68 MB volume) - same points evaluated, other collective; "
                         "the default (volume all-gather, what BASELINE.json's north_star names) is what the driver measures")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="N > 1: CUs the persistent MLP kernel leaves to the RCCL kernels of the overlapped all_gather (DenseReconEngine "
                         "reserve_cus); -1 = the engine's default (16 over RCCL with the overlapped gather, else 0)")
    ap.add_argument("--no-overlap-gather", action="store_true",
                    help="N > 1, sharded: one blocking sign exchange and ONE volume all_gather after the whole slab instead of the "
                         "split / overlapped protocol (DenseReconEngine overlap_gather=False) - the A/B leg of tools/scale_round.sh")
    ap.add_argument("--slab-layout", default="ab", choices=["ab", "contiguous"],
                    help="sharded only: 'ab' = two Z-slabs per rank, each volume gather lands in one contiguous block of the result (no "
                         "assembly copies); 'contiguous' = the round-5 layout (one cost-weighted slab per rank, gathered in two halves)")
    ap.add_argument("--gather-to", type=int, default=-1,
                    help="sharded, 'ab' layout: only this rank receives the volume (gather instead of all_gather); -1 = every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-extras", action="store_true",
                    help="also run the whole-lattice CPU checker legs (parity sample, mesh vs oracle) above 257^3 (513^3: about a minute)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the post-timing legs (reference schedule, parity sample, mesh Chamfer): "
                         "use it under rocprofv3 so the trace holds only the dense step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves - re-exec under torch.distributed.run, one process per
        # GPU, rendezvous on 127.0.0.1 (the driver's own `python -m torch.distributed.run ... bench.py` form sets WORLD_SIZE
        # and never comes through here)
        sys.exit(self_launch(args.gpus))

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
    if args.gpus != world and world > 1:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    _lib.require_device()
    # ICON_AMD_DIST_BACKEND=gloo: debugging aid - several ranks on the GPUs that are there (RCCL refuses two ranks on
    # one device); collectives are staged through the host, the numbers mean nothing, the control flow is the real one
    backend = os.environ.get("ICON_AMD_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        # rank setup, before any rendezvous: one process per GPU needs one device per rank
        raise SystemExit(f"bench.py rank {rank}: need {world} HIP devices for --gpus {world} (one process per GPU over RCCL), "
                         f"this node shows {torch.cuda.device_count()}; ICON_AMD_DIST_BACKEND=gloo runs the ranks on the "
                         "devices that are there (control flow only, the numbers mean nothing)")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where the bookkeeping collectives live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    res = args.res
    a = synth.make_assets("body", prior_type=args.prior)
    T = lambda x: torch.from_numpy(x).to(dev)

    def make_engine(precision):
        e = IconQueryEngine(prior_type=args.prior, sdf_clip=a.sdf_clip, cmap_mode=args.cmap_mode, search=args.search,
                            precision=precision)
        if args.prior == "icon":
            e.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
        else:
            e.set_volume_features(T(a.vol_feat))
        e.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
        return e

    eng = make_engine(args.precision)
    feats = [T(a.features)]
    recon = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                             resolutions={257: [33, 65, 129, 257], 513: [33, 65, 129, 257, 513]}.get(res, [res]), align_corners=True,
                             balance_value=0.5, faster=True, engine=eng, shard=not args.replicas, overlap_gather=not args.no_overlap_gather,
                             reserve_cus=None if args.reserve_cus < 0 else args.reserve_cus, slab_layout=args.slab_layout,
                             gather_to=None if (args.gather_to < 0 or world == 1 or args.replicas) else args.gather_to).to(dev)
    opt = SimpleNamespace(num_views=1)

    mesh_exchange = bool(args.mesh_exchange and world > 1 and not args.replicas)

    def step(r=recon, e=eng):
        if mesh_exchange:
            return r.forward_mesh(opt=opt, netG=e, features=feats, proj_matrix=None)
        return r(opt=opt, netG=e, features=feats, proj_matrix=None)

    # one-off per-image preparation (BVH build, plane repack, BatchNorm fold + operand packing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng._mesh_handle(); eng._feat_handle(feats[0]); eng._mlp_handle()
    torch.cuda.synchronize()
    prep_ms = (time.perf_counter() - t0) * 1e3

    for _ in range(args.warmup):
        step()
    eng._work().profile(True)
    two_works = world > 1 and not args.replicas and not args.no_overlap_gather and not mesh_exchange
    if two_works:
        eng._work(1).profile(True)                      # split phase 1: the second half-slab runs on its own workspace
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
        last_stage = np.array(eng._work().stage_ms())
        stage += last_stage
        if two_works and recon.last_stats.get("split_features"):
            try:
                stage += np.array(eng._work(1).stage_ms())     # the rank's stage times = both half-slabs
            except Exception:
                pass                                    # (a rank whose slab fits the first half: no call on workspace 1)
    barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage /= max(args.steps, 1)
    try:
        detail = eng._work().profile_detail()          # of the LAST timed step: search kernel alone, cycles / effective clock of the MLP kernel
    except Exception as ex:
        detail = {"error": repr(ex)}
    try:
        wg = eng._work().profile_workgroups()           # of the same launch: one record per workgroup of the persistent grid
    except Exception as ex:
        wg = repr(ex)
    eng._work().profile(False)
    if two_works:
        eng._work(1).profile(False)
    if mesh_exchange:                                   # the step returned (verts, faces): what marching cubes on the volume gives
        assert occ is not None and occ[0].shape[1] == 3 and occ[1].shape[1] == 3 and occ[1].shape[0] > 0
    elif getattr(recon, "gather_to", None) is not None and recon.gather_to != rank:
        assert occ is None                              # gather_to: only the destination rank holds the volume
    else:
        assert occ is not None and occ.shape == (res, res, res)

    n_points = res ** 3
    # the cut the engine actually used: one Z-slab per rank, or - 'ab' layout - two (DenseReconEngine.ab_pieces)
    if world > 1 and not args.replicas:
        my_slabs = list(recon.last_stats["pieces"][rank]) if recon.last_stats.get("pieces") else [tuple(recon.last_stats["slabs"][rank])]
    else:
        my_slabs = [(0, res)]
    z0, z1 = my_slabs[0][0], my_slabs[-1][1]
    my_points = sum(b - a for a, b in my_slabs) * res * res
    images = world if args.replicas else 1
    value = images * n_points * args.steps / elapsed
    mlp_s = stage[2] * 1e-3
    # the points the MLP kernel actually multiplies: with the shell skip (fused f16x3 path, BVH search) the strict in_cube
    # shell of the lattice is written as 0 without being evaluated - 393 k of the 16.97 M points of a 257^3 volume
    shell_skipped = args.precision == "f16x3" and args.search == "bvh" and os.environ.get("ICON_AMD_SHELL_SKIP", "1") != "0" \
        and os.environ.get("ICON_AMD_UNFUSED", "0") == "0"
    exec_points = sum(max(min(b, res - 1) - max(a, 1), 0) for a, b in my_slabs) * (res - 2) ** 2 if shell_skipped else my_points
    achieved = (MLP_FLOP_PER_POINT * exec_points / mlp_s) / 1e12 if mlp_s > 0 else 0.0
    # HBM bytes of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE need their own rocprofv3 runs,
    # tools/gpu_round.sh): quoted only while profiles/traffic.json was taken on exactly these kernel sources
    traffic, traffic_note = None, "profiles/traffic.json absent"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from rocprof_summary import kernel_sources_sha
            tj = json.load(open(tpath))
            if tj.get("kernel_sources_sha") != kernel_sources_sha():
                traffic_note = ("profiles/traffic.json is stale: taken on kernel sources " + str(tj.get("kernel_sources_sha"))
                                + ", this tree is " + kernel_sources_sha() + " - regenerate with tools/gpu_round.sh")
            elif res != 257 or args.prior != "icon":
                traffic_note = "profiles/traffic.json holds the 257^3 icon step only"
            else:
                traffic = tj.get(KERNEL[args.precision] + "_bytes_per_launch")
                traffic_note = ("PMC FETCH_SIZE (x 1.0: calibrated on the known slot / code arrays, tools/rocprof_summary.py) + WRITE_SIZE "
                                "per launch (profiles/traffic.json, same kernel sources)")
                if traffic is not None and my_points != n_points:
                    traffic = traffic * my_points / n_points          # the PMC passes were taken on whole-volume launches
        except Exception as ex:
            traffic, traffic_note = None, f"profiles/traffic.json unreadable: {ex!r}"

    # per-rank stage times (every rank's slab differs in traversal cost): makes a SCALE run diagnosable
    rank_stage = None
    if world > 1:
        mine = torch.tensor([z0, z1, stage[0], stage[1], stage[2], my_elapsed / args.steps * 1e3], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_stage = [{"rank": r, "planes": [int(v[0]), int(v[1])], "features_ms": float(v[2]), "cmap_patch_ms": float(v[3]),
                       "mlp_ms": float(v[4]), "step_ms": float(v[5])} for r, v in enumerate(allr)]

    # what the process group actually is (the driver's "did RCCL see N ranks" check reads this): backend, the world size the
    # group reports, every rank's device as IT sees it
    dist_info = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": local_dev, "device": props.name,
                "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            rccl = None
        dist_info = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "rccl_version": rccl,
                     "distinct_devices": len({(d["device_index"], d["uuid"]) for d in seen}), "ranks": seen}
    extras = {}
    sustained = None
    zero_ms = None
    if not args.no_extras and world == 1 and rank == 0 and args.precision == "f16x3":
        try:
            sustained = library_gemm_ceiling(dev)
        except Exception as ex:
            sustained = {"error": repr(ex)}
        try:
            zero_ms = mlp_zero_data(dev, 257 ** 3)
        except Exception as ex:
            zero_ms = repr(ex)
    if not args.no_extras and world == 1 and rank == 0 and args.prior == "icon":
        from icon_amd.recon import AdaptiveReconEngine, export_mesh_device
        from icon_amd import metrics
        # (2) the reference's own coarse-to-fine schedule (Seg3dLossless._forward_faster, ~1 % of the lattice
        #     queried, last level interpolated) on the same engine: second baseline line + the "reference mesh"
        sched = {257: [33, 65, 129, 257], 513: [33, 65, 129, 257, 513]}.get(res)      # apps/ICON.py:62-72 for mcube_res 256 / 512
        if sched:
            ad = AdaptiveReconEngine(faster=True, query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                     resolutions=sched, align_corners=True).to(dev)
            import gc
            for _ in range(5):
                vol_ad = ad(opt=opt, netG=eng, features=feats, proj_matrix=None)
            # every call on its own (the stream drained in between), 200 of them, the garbage collector frozen as INTEGRATION.md
            # recommends for a serving loop: the distribution, not one median
            gc.collect(); gc.freeze(); gc_was = gc.isenabled(); gc.disable()
            ts = []
            try:
                for _ in range(200):
                    t1 = time.perf_counter()
                    vol_ad = ad(opt=opt, netG=eng, features=feats, proj_matrix=None)     # (returns after its own synchronisation: the counts)
                    ts.append((time.perf_counter() - t1) * 1e3)
            finally:
                if gc_was:
                    gc.enable()
                gc.unfreeze()
            extras["reference_schedule_ms_per_volume"] = float(np.median(ts))
            extras["reference_schedule_ms"] = {"p50": float(np.percentile(ts, 50)), "p99": float(np.percentile(ts, 99)), "max": float(np.max(ts)),
                                               "calls": len(ts), "gc": "frozen + disabled",
                                               "launches": "21 kernel launches per [33,65,129,257] schedule, no fill, no copy (profiles/r06_adaptive_timeline.csv)"}
            extras["reference_schedule_native"] = bool(ad.last_stats.get("native", False))
            extras["reference_schedule_points"] = int(sum(ad.last_stats.get("queries", [])))
            # (3) mesh Chamfer / P2S, lib/dataset/Evaluator.py:200-230, in [-1,1]-cube units x100 (apps/ICON.py:758-759)
            try:
                t1 = time.perf_counter()
                vd, fd = export_mesh_device(occ, 0.5)
                va, fa = export_mesh_device(vol_ad.contiguous(), 0.5)
                ch, p2s = metrics.chamfer_p2s(metrics.to_unit_cube(vd, res), fd, metrics.to_unit_cube(va, res), fa, n=100_000)
                torch.cuda.synchronize()
                extras["mesh"] = {"chamfer_x100_dense_vs_reference_schedule": ch, "p2s_x100": p2s,
                                  "voxel_x100": 2.0 / (res - 1) * 100.0, "faces_dense": int(fd.shape[0]),
                                  "faces_reference_schedule": int(fa.shape[0]), "samples_per_mesh": 100_000,
                                  "seconds": time.perf_counter() - t1,
                                  "note": f"(ii) dense {res}^3 field vs the reference's adaptive schedule (interpolated last level): "
                                          "a sub-voxel non-zero value is expected (SURVEY.md finding 1)"}
                del vd, fd, va, fa
            except Exception as ex:
                extras["mesh"] = {"error": repr(ex)}
            # (3b) metric (i): the same dense mesh against the ORACLE's (CPU leg: 257^3 only unless --full-extras)
            if res <= 257 or args.full_extras:
                try:
                    extras.setdefault("mesh", {}).update(mesh_vs_oracle(a, res, occ, args.cmap_mode))
                except Exception as ex:
                    extras.setdefault("mesh", {})["dense_vs_oracle_error"] = repr(ex)
        # (2b) COLD per-image figures: fresh SMPL tensors every step (what apps/infer.py does: filter() hands over new tensors
        #      per image, lib/net/HGPIFuNet.py:236-240) - the mesh preparation (normals, BVH, records, ray bins: kernels on the
        #      stream, icon_mesh_create_arena) is inside the timed region, unlike `value`
        try:
            base = [T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis)]

            def cold(fn, n=6):
                ts, host = [], []
                for _ in range(n):
                    fresh = [t.clone() for t in base]            # stands for filter(): new tensors, new addresses
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    eng.set_mesh(*fresh)
                    eng._mesh_handle()
                    host.append(time.perf_counter() - t1)
                    fn()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t1)
                return float(np.median(ts[1:]) * 1e3), float(np.median(host[1:]) * 1e3)
            dense_ms, create_host_ms = cold(step)
            cold_cfg = {"dense": dense_ms, "mesh_create_host_ms": create_host_ms,
                        "note": "median of 5 after one warm-up; new SMPL tensors every step, mesh built on the device inside the timed region; "
                                "mesh_create_host_ms = host time of icon_mesh_create_arena (enqueue only, it never waits); reference_mode_image = "
                                "mesh build + the reference's schedule + marching cubes + clean_mesh (apps/ICON.py:729-761), device tensors throughout"}
            if sched:
                cold_cfg["reference_schedule"] = cold(lambda: ad(opt=opt, netG=eng, features=feats, proj_matrix=None))[0]
                # the whole image in the reference's own mode (apps/ICON.py:729-761): schedule -> export_mesh -> clean_mesh
                from icon_amd.recon import clean_mesh

                def image():
                    vol = ad(opt=opt, netG=eng, features=feats, proj_matrix=None)
                    clean_mesh(*export_mesh_device(vol.contiguous(), 0.5))
                cold_cfg["reference_mode_image"] = cold(image)[0]
                img_stage = []
                for _ in range(5):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    vol = ad(opt=opt, netG=eng, features=feats, proj_matrix=None); torch.cuda.synchronize(); t2 = time.perf_counter()
                    vm, fm = export_mesh_device(vol.contiguous(), 0.5); torch.cuda.synchronize(); t3 = time.perf_counter()
                    clean_mesh(vm, fm); torch.cuda.synchronize(); t4 = time.perf_counter()
                    img_stage.append(((t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
                med = np.median(np.array(img_stage[1:]), 0)
                cold_cfg["reference_mode_stages_warm"] = {"schedule": float(med[0]), "marching_cubes": float(med[1]), "clean_mesh": float(med[2])}
                cold_cfg["reference_mode_mesh"] = {"faces_marching_cubes": int(fm.shape[0]), "schedule": sched,
                                                   "schedule_points": int(sum(ad.last_stats.get("queries", []))),
                                                   "native": bool(ad.last_stats.get("native", False))}
                free_b, total_b = torch.cuda.mem_get_info(dev)
                cold_cfg["hbm_gib"] = {"in_use_after_the_image": (total_b - free_b) / 2 ** 30, "torch_peak_allocated": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
                                       "note": "device memory in use by this process after the legs above (dense step + schedule + mesh extraction; the "
                                               "library's workspaces only grow) and the peak of torch's own tensors"}
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fresh = [t.clone() for t in base]
            eng.set_mesh(*fresh)
            torch.cuda.synchronize()
            ev0.record(); eng._mesh_handle(); ev1.record()
            torch.cuda.synchronize()
            cold_cfg["mesh_build_device_ms"] = float(ev0.elapsed_time(ev1))
            extras["cold_image_ms"] = cold_cfg
        except Exception as ex:
            extras["cold_image_ms"] = {"error": repr(ex)}
        # (2c) what leaving CUs to a collective would cost the dense step (multi-GPU knob reserve_cus, measured on ONE GPU: the
        #      persistent MLP kernel's grid shrinks by k workgroups)
        try:
            cost = {}
            for k in (0, 8, 16, 32):
                eng._work().set_reserve_cus(k)
                step(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                cost[str(k)] = (time.perf_counter() - t1) / 5 * 1e3
            eng._work().set_reserve_cus(0)
            extras["reserve_cus_cost"] = {"ms_per_step": cost, "note": "dense 257^3 step on one GPU with the MLP kernel's persistent grid "
                                          "k workgroups smaller (k CUs left to RCCL when sharded; DenseReconEngine(reserve_cus=k))"}
        except Exception as ex:
            extras["reserve_cus_cost"] = {"error": repr(ex)}
        # (4) live parity sample against the checker (CPU leg over the whole lattice: 257^3 only unless --full-extras)
        if res <= 257 or args.full_extras:
            try:
                extras["parity"] = parity_sample(a, res, occ, args.cmap_mode, eng=eng)
            except Exception as ex:
                extras["parity"] = {"error": repr(ex)}
        # (5) how much of the volume depends on the tie rule of the unpinned nearest-triangle leaf
        try:
            def step_with(e):
                r = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[res],
                                     align_corners=True, engine=e).to(dev)
                return step(r, e)
            extras["tie_sensitivity"] = tie_sensitivity(a, res, occ, make_engine, step_with)
        except Exception as ex:
            extras["tie_sensitivity"] = {"error": repr(ex)}

    if rank == 0:
        cfg_name = "icon-filter.yaml" if args.prior == "icon" else "pamir.yaml (hoisted VolumeEncoder output [1,7,32^3])"
        out = {
            "metric": "query-points/sec at 256^3 grid", "value": value, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.replicas else "strong", "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic",
            "config": {
                "workload": f"{cfg_name}, {res}^3 lattice (mcube_res={res - 1}), {images} image{'s, one per GPU' if images > 1 else ''}: SMPL-size body "
                            f"V=6890/F=13776, planes [1,{a.features.shape[1]},128,128], MLP 13-512-256-128-1, "
                            f"cmap_mode={args.cmap_mode}, mlp={args.precision}",
                "parallelism": f"replicas{world}" if args.replicas else f"zslab{world}", "points_per_step": images * n_points, "prep_ms": prep_ms,
                "stage_ms": {"features": stage[0], "cmap_patch": stage[1], "mlp": stage[2]},
                **extras,
            },
            "roofline": {"bound": "mfma", "kernel": KERNEL[args.precision], "achieved": achieved,
                         "peak": PEAK_TFLOPS[args.precision],
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": traffic, "traffic_note": traffic_note,
                         "flop_per_launch": MLP_FLOP_PER_POINT * exec_points, "points_multiplied_per_launch": exec_points,
                         "avg_launch_ms": stage[2],
                         "algorithmic_hbm_bytes_per_launch": ALGO_BYTES_PER_POINT * my_points},
        }
        # box or build?  effective clock = shader cycles / wall time of the MLP kernel's workgroup 0 (s_memtime / s_memrealtime
        # stamps inside the launch, last timed step); cycles_per_launch is the CODE's figure, the clock is the BOX's
        if "error" not in detail:
            out["roofline"].update({"effective_clock_mhz": detail["effective_clock_mhz"], "cycles_per_launch": detail["fused_cycles"],
                                    "clock_note": "s_memtime / s_memrealtime stamps of the kernel's workgroup 0 in the last timed step: "
                                                  "cycles_per_launch moves with the code, effective_clock_mhz with the box (nominal 2400)"})
            out["config"]["stage_ms"]["nearest_kernel"] = detail["nearest_ms"]
        else:
            out["roofline"]["clock_note"] = detail["error"]
        # where the launch ends: the kernel is a persistent grid (one workgroup per CU) and finishes with its slowest workgroup.
        # Every workgroup stamps its start / end (s_memrealtime) and its cycles (s_memtime) and names its XCD (HW_REG_XCC_ID)
        if isinstance(wg, np.ndarray) and len(wg):
            span, xcd, mhz = wg[:, 2], wg[:, 0].astype(int), wg[:, 3] / wg[:, 2] * 1e-3
            out["roofline"].update({
                "wg_span_ms": {"min": float(span.min()), "median": float(np.median(span)), "max": float(span.max())},
                "per_xcd_clock_mhz": [float(mhz[xcd == x].mean()) if (xcd == x).any() else None for x in range(8)],
                "per_xcd_tiles": [int(wg[xcd == x, 4].sum()) for x in range(8)],
                "tail_ms": float(last_stage[2] - np.median(span)),
                "tail_inside_kernel_ms": float((wg[:, 1] + span).max() - np.median(span)),
                "workgroups": int(len(wg)), "tiles_per_workgroup": {"min": int(wg[:, 4].min()), "max": int(wg[:, 4].max())},
                "partition": "static runs + a pool (15 % of the tiles by default) drawn in groups of 2 by the workgroups that finish first "
                             "(icon_work_set_steal); the XCDs hold different clocks under the power limit - per_xcd_clock_mhz - and take "
                             "tiles in that proportion - per_xcd_tiles",
                "wg_note": "last timed step.  tail_ms = that step's launch duration (HIP events around the kernel and its two 5-us companions) - "
                           "the median workgroup span; tail_inside_kernel_ms = last workgroup's end - first workgroup's start - median span"})
        elif not isinstance(wg, np.ndarray):
            out["roofline"]["wg_note"] = wg
        if isinstance(zero_ms, float):
            out["roofline"].update({"mlp_zero_data_ms": zero_ms, "mlp_zero_data_points": 257 ** 3,
                                    "zero_data_note": "standalone k_mlp_f16x3 (the fused kernel's chunk bodies) on all-zero rows and weights, 257^3 "
                                                      "points, same run: the instruction stream's own time at the clock the box can hold"})
        elif zero_ms is not None:
            out["roofline"]["zero_data_note"] = zero_ms
        if sustained is not None and "error" not in sustained:
            # the library's f16 GEMM rate on this box, random operands, vs what the kernel ISSUES (3 f16 MFMA products per MAC)
            lib_peak = max(v["tflops"] for v in sustained.values())
            out["roofline"].update({"sustained_peak": lib_peak, "sustained_peak_layer1_shape": sustained["layer1_points_x512_x256"]["tflops"],
                                    "issued_tflops": 3.0 * achieved, "frac_of_sustained": 3.0 * achieved / lib_peak,
                                    "sustained_note": "hipBLASLt / rocBLAS f16 GEMM (f32 accumulate, N(0,1) operands) timed on this box in this run: "
                                                      "the best of [2M x 512] x [512 x 256] (layer 1's shape) and 8192^3; frac_of_sustained = "
                                                      "3 x achieved / that (the kernel issues three f16 MFMA products per algorithmic MAC)",
                                    "library_gemm": sustained})
        elif sustained is not None:
            out["roofline"]["sustained_peak"] = None
            out["roofline"]["sustained_note"] = sustained["error"]
        if rank_stage is not None:
            out["config"]["rank_stage_ms"] = rank_stage
        if dist_info is not None:
            out["config"]["dist"] = dist_info
        if world > 1 and not args.replicas:
            out["config"]["reserve_cus"] = getattr(recon, "reserve_cus_effective", None)     # CUs the MLP grid left to the collective
            out["config"]["gather"] = "mesh" if mesh_exchange else "volume"
            out["config"]["overlap_gather"] = not args.no_overlap_gather
            out["config"]["split_features"] = bool(recon.last_stats.get("split_features", False))
            out["config"]["slab_layout"] = recon.last_stats.get("layout", "contiguous")
            out["config"]["gather_to"] = recon.last_stats.get("gather_to")
            out["config"]["assembly_copies"] = recon.last_stats.get("assembly_copies")
            if recon.last_stats.get("pieces"):
                out["config"]["pieces"] = [[list(a), list(b)] for a, b in recon.last_stats["pieces"]]
            if mesh_exchange:
                out["config"]["exchanged_bytes_per_step"] = recon.last_stats.get("exchanged_bytes")
        if not args.no_cpu_baseline and world == 1 and args.prior == "icon":
            out["cpu_baseline"] = cpu_baseline(a, res)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
