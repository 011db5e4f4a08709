"""Reference-verbatim vs port: the two CPU baselines bench.py can report, timed side by side on the same lattice planes
(build container only - /root/reference does not travel to the GPU box, where bench.py times the port).
    python tools/cpu_baseline_compare.py [--res 257] [--planes 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=257)
    ap.add_argument("--planes", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch
    from icon_amd import synth
    from oracle import oracle as orc, query_torch as qt, ref_loader
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores); orc.set_num_threads(cores)
    a = synth.make_assets("body")
    res = args.res
    zs = np.unique(np.linspace(res // 8, res - 1 - res // 8, args.planes).round().astype(int))
    pts = np.concatenate([synth.lattice_points(res, int(z), int(z) + 1) for z in zs])
    mlp = qt.build_mlp(a.state_dict)
    out = {}

    def port(p):
        return qt.query(a, mlp, p, a.sdf_clip)
    runs = {"port": port}
    if ref_loader.available():
        ref = ref_loader.load()
        netG, cfg = ref_loader.build_netG(a)
        feats = [torch.from_numpy(a.features)]

        def reference(p):
            with torch.no_grad():
                return ref.query_func(cfg, netG, feats, torch.from_numpy(p)[None])[0, 0].numpy()
        runs["reference"] = reference
    vals, best = {}, {k: 1e30 for k in runs}
    for fn in runs.values():
        fn(pts[:4096])
    for _ in range(args.repeat):                     # interleaved, best of N: the build container shares its cores
        for name, fn in runs.items():
            t0 = time.perf_counter(); vals[name] = fn(pts); best[name] = min(best[name], time.perf_counter() - t0)
    for name in runs:
        out[name] = len(pts) / best[name]
        print(f"{name:10s} {len(pts)} points ({len(zs)} planes of {res}^3) in {best[name]:.2f} s -> {out[name] / 1e3:.1f} k points/s on {cores} cores "
              f"(torch {torch.__version__})")
    if "reference" in vals:
        print(f"max |reference - port| = {np.abs(vals['reference'] - vals['port']).max():.3e}; port / reference throughput = {out['port'] / out['reference']:.3f}")


if __name__ == "__main__" and "--adaptive" not in sys.argv:
    main()


def reference_adaptive(repeat=3):
    """Second baseline line of BASELINE.md section 3: wall clock of the reference's OWN schedule - Seg3dLossless [33,65,129,257],
    faster=True (apps/ICON.py:62-90), run verbatim on the CPU over the exact accelerated leaves - for one image."""
    import numpy as np
    import torch
    from icon_amd import synth
    from oracle import oracle as orc, ref_loader
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores); orc.set_num_threads(cores)
    ref = ref_loader.load()
    a = synth.make_assets("body")
    netG, cfg = ref_loader.build_netG(a)
    counts = []

    def qf(opt, netG, features, points, proj_matrix=None):
        counts.append(int(points.shape[1]))
        return ref.query_func(opt, netG, features, points, proj_matrix)
    best = 1e30
    with torch.no_grad():
        eng = ref.Seg3dLossless(query_func=qf, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[33, 65, 129, 257],
                                align_corners=True, balance_value=0.5, faster=True)
        for _ in range(repeat):
            counts.clear()
            t0 = time.perf_counter()
            vol = eng(opt=cfg, netG=netG, features=[torch.from_numpy(a.features)], proj_matrix=None)
            best = min(best, time.perf_counter() - t0)
    print(f"reference Seg3dLossless [33,65,129,257] faster=True, verbatim, {cores} cores: {best * 1e3:.0f} ms per image "
          f"({sum(counts)} points queried in {len(counts)} calls: {counts}); volume {tuple(vol.shape)}")


if __name__ == "__main__" and "--adaptive" in sys.argv:
    reference_adaptive()
