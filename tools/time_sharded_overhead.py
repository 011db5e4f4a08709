"""Overhead of the sharded (multi-GPU) code path itself, measured with a ONE-rank NCCL group:
_forward_sharded (messages + 2 collectives + slab buffers) vs the plain single-GPU call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from types import SimpleNamespace
from icon_amd import synth
from icon_amd.engine import IconQueryEngine, query_func
from icon_amd.recon import DenseReconEngine
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feats = [T(a.features)]; opt = SimpleNamespace(num_views=1)
for res in (257, 129):
    rec = DenseReconEngine(query_func=query_func, resolutions=[33, res], align_corners=True, engine=eng).cuda()
    def timed(f, n=10):
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    plain = timed(lambda: rec(opt=opt, netG=eng, features=feats, proj_matrix=None))
    shard = timed(lambda: rec._none_if_empty(rec._forward_sharded(eng, feats[0], res, dist, 1, 0)))
    print(f"res {res}: plain forward {plain:.3f} ms, sharded path with world=1 {shard:.3f} ms (overhead {shard - plain:.3f} ms)")
dist.destroy_process_group()
