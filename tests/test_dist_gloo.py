"""CPU, world_size 2 over gloo: the Z-slab sharding of DenseReconEngine - slab bounds, the
outlier sign-list exchange (counts, padding, concatenation order, rank offsets) and the final
all_gather - with the compute backend replaced by a CPU checker built on the oracle.

The product has no CPU compute path; the `backend=` injection point exists for exactly this test
(the default backend is the HIP engine)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT, assets, orc
from icon_amd import synth
from icon_amd.recon import DenseReconEngine, slab_bounds

RES = 17


class OracleBackend:
    """eval_slab / slab_features / slab_finish with the oracle; asserts that what the distributed
    driver hands to slab_finish is exactly the global outlier list of the whole lattice."""
    prior_type = "icon"

    def __init__(self, a, cmap_mode):
        self.a, self.cmap_mode = a, cmap_mode
        pts = synth.lattice_points(RES)
        self.full, _ = orc.query_icon(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], a.features,
                                      orc.Mlp(a.state_dict), pts, sdf_clip=a.sdf_clip,
                                      cmap_local=(cmap_mode == "local"))
        self.full = self.full.reshape(RES, RES, RES)
        sdf = orc.cal_sdf(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], pts)["sdf"]
        self.out_mask = (np.abs(sdf) >= np.float32(a.sdf_clip)).reshape(RES, RES * RES)
        self.signs = np.sign(sdf).astype(np.int8).reshape(RES, RES * RES)
        self.calls = []

    def _local_list(self, z0, z1):
        return self.signs[z0:z1][self.out_mask[z0:z1]]

    def eval_slab(self, im_feat, res, z0, z1, out=None):
        self.calls.append(("eval", z0, z1))
        t = torch.from_numpy(self.full[z0:z1].copy())
        if out is not None:
            out.copy_(t)
            return out
        return t

    def slab_features(self, im_feat, res, z0, z1, signs=None, count=None):
        self.calls.append(("features", z0, z1))
        lst = self._local_list(z0, z1)
        if signs is None:
            signs = torch.zeros((z1 - z0) * res * res, dtype=torch.int8)
        if count is None:
            count = torch.zeros(1, dtype=torch.int64)
        assert signs.numel() == (z1 - z0) * res * res and signs.dtype == torch.int8 and count.dtype == torch.int64
        signs[: len(lst)] = torch.from_numpy(lst.copy())
        signs[len(lst):] = 99                       # garbage past the count must never be used
        count[0] = len(lst)
        return signs, count

    def slab_finish_gathered(self, res, z0, z1, gathered, stride, world, rank, out=None):
        """the single-collective protocol: message r = [int64 count_r][int8 signs_r ...] at r * stride"""
        self.calls.append(("finish", z0, z1))
        assert gathered.dtype == torch.int8 and gathered.numel() == world * stride and stride % 8 == 0
        g = gathered.numpy()
        counts = [int(g[r * stride: r * stride + 8].view(np.int64)[0]) for r in range(world)]
        lst = np.concatenate([g[r * stride + 8: r * stride + 8 + c] for r, c in enumerate(counts)])
        exp = self._local_list(0, res)
        assert sum(counts) == len(exp)
        assert np.array_equal(lst, exp), "global sign list differs from lattice order"
        assert sum(counts[:rank]) == int(self.out_mask[:z0].sum())
        return self.eval_slab(None, res, z0, z1, out=out)

    def slab_finish(self, res, z0, z1, signs_global, k_total, rank_offset, out=None, device=None):
        self.calls.append(("finish", z0, z1))
        exp = self._local_list(0, res)
        assert k_total == len(exp)
        assert np.array_equal(signs_global.numpy(), exp), "global sign list differs from lattice order"
        assert rank_offset == int(self.out_mask[:z0].sum())
        return self.eval_slab(None, res, z0, z1, out=out)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cmap_mode, q, legacy=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = assets("ico")
        be = OracleBackend(a, cmap_mode)
        be.cmap_mode = cmap_mode
        if legacy:
            class _NoGathered:                   # proxy without slab_finish_gathered -> legacy two-step exchange
                def __init__(self, inner): self._i = inner
                def __getattr__(self, name):
                    if name == "slab_finish_gathered":
                        raise AttributeError(name)
                    return getattr(self._i, name)
            be_used = _NoGathered(be)
        else:
            be_used = be
        recon = DenseReconEngine(query_func=None, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                 resolutions=[9, RES], align_corners=True, backend=be_used)
        occ = recon(opt=None, netG=None, features=[torch.from_numpy(a.features)], proj_matrix=None)
        ok = occ is not None and occ.shape == (RES, RES, RES) and np.array_equal(occ.numpy(), be.full)
        z0, z1, _ = slab_bounds(RES, world, rank)
        kinds = [c[0] for c in be.calls]
        ok = ok and all(c[1:] == (z0, z1) for c in be.calls)
        ok = ok and (kinds == (["features", "finish", "eval"] if cmap_mode == "reference" else ["eval"]))
        q.put((rank, bool(ok), kinds))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cmap_mode,legacy", [("reference", False), ("reference", True), ("local", False)])
@pytest.mark.parametrize("world", [2, 3])
def test_zslab_sharding_gloo(cmap_mode, legacy, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cmap_mode, q, legacy)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res


def test_single_process_path_uses_one_slab():
    a = assets("ico")
    be = OracleBackend(a, "reference")
    recon = DenseReconEngine(resolutions=[RES], align_corners=True, backend=be)
    occ = recon(opt=None, netG=None, features=[torch.from_numpy(a.features)], proj_matrix=None)
    assert np.array_equal(occ.numpy(), be.full) and be.calls == [("eval", 0, RES)]
