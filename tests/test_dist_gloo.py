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
    split_features = False     # True: the driver may run phase 1 per half-slab on workspaces 0 / 1 (the HIP engine's protocol since round 5)

    def __init__(self, a, cmap_mode, split=False):
        self.a, self.cmap_mode = a, cmap_mode
        self.split_features = split
        self.phase1 = {}           # workspace -> the slab its phase 1 ran on
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

    def slab_features(self, im_feat, res, z0, z1, signs=None, count=None, msg=None, work=0):
        self.calls.append(("features", z0, z1))
        self.phase1[work] = (z0, z1)
        lst = self._local_list(z0, z1)
        if msg is not None:
            # the exchange message of the single-collective protocol: [int64 K][2 bits per sign (sign + 1), 4 per byte]
            n = (z1 - z0) * res * res
            assert msg.element_size() == 1 and msg.numel() >= 8 + (n + 3) // 4
            m = msg.numpy().view(np.uint8)
            m[:] = 0xAB                               # garbage past the packed signs must never be used
            m[:8] = np.array([len(lst)], np.int64).view(np.uint8)
            pad = np.zeros((len(lst) + 3) // 4 * 4, np.uint8)
            pad[: len(lst)] = (lst + 1).astype(np.uint8)
            m[8:8 + len(pad) // 4] = (pad.reshape(-1, 4) << (2 * np.arange(4, dtype=np.uint8))).sum(1).astype(np.uint8)
            return msg
        if signs is None:
            signs = torch.zeros((z1 - z0) * res * res, dtype=torch.int8)
        if count is None:
            count = torch.zeros(1, dtype=torch.int64)
        assert signs.numel() == (z1 - z0) * res * res and signs.dtype == torch.int8 and count.dtype == torch.int64
        signs[: len(lst)] = torch.from_numpy(lst.copy())
        signs[len(lst):] = 99                       # garbage past the count must never be used
        count[0] = len(lst)
        return signs, count

    def slab_finish_gathered(self, res, z0, z1, gathered, stride, world, rank, out=None, za=None, zb=None, device=None, work=0):
        """the single-collective protocol: message r = [int64 count_r][2-bit packed signs_r ...] at r * stride;
        evaluates the planes [za, zb) of the slab into out (the slab's buffer).  (Split phase 1: `world` / `rank` count HALF-slabs.)"""
        za = z0 if za is None else za
        zb = z1 if zb is None else zb
        assert z0 <= za < zb <= z1
        assert self.phase1.get(work) == (z0, z1), "phase 2 on a workspace whose phase 1 ran on another slab"
        self.calls.append(("finish", z0, z1, za, zb))
        if gathered is not None:
            assert gathered.element_size() == 1 and gathered.numel() == world * stride and stride % 8 == 0
            g = gathered.numpy().view(np.uint8)
            counts = [int(g[r * stride: r * stride + 8].view(np.int64)[0]) for r in range(world)]
            lst = []
            for r, c in enumerate(counts):
                b = g[r * stride + 8: r * stride + 8 + (c + 3) // 4]
                lst.append((((b[:, None] >> (2 * np.arange(4, dtype=np.uint8))) & 3).reshape(-1)[:c]).astype(np.int8) - 1)
            lst = np.concatenate(lst)
            exp = self._local_list(0, res)
            assert sum(counts) == len(exp)
            assert np.array_equal(lst, exp), "global sign list differs from lattice order"
            assert sum(counts[:rank]) == int(self.out_mask[:z0].sum())
        else:
            assert self.cmap_mode == "local"
        out[za - z0: zb - z0].copy_(torch.from_numpy(self.full[za:zb].copy()))
        return out

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


def _worker(rank, world, port, cmap_mode, q, legacy=False, overlap=True, skew=False, split=False, layout="contiguous", gather_to=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = assets("ico")
        be = OracleBackend(a, cmap_mode, split=split)
        be.cmap_mode = cmap_mode
        if skew:
            # replicas of the SMPL tensors that differ between ranks (here grossly): every rank would derive another
            # cost-weighted cut on its own - rank 0's cut must be the one everybody uses
            be._mesh_key = ("image-1",)
            be.mesh_z_range = lambda: (-0.9 + 0.5 * rank, -0.6 + 0.5 * rank)
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
                                 resolutions=[9, RES], align_corners=True, backend=be_used, overlap_gather=overlap, slab_layout=layout,
                                 gather_to=gather_to)
        occ = recon(opt=None, netG=None, features=[torch.from_numpy(a.features)], proj_matrix=None)
        if layout == "ab":
            # two slabs per rank; each volume gather lands in one contiguous block of ONE buffer: the result is a view of it
            pa, pb, pieces = DenseReconEngine.ab_pieces(RES, world)
            st = recon.last_stats
            ok = st.get("layout") == "ab" and st["assembly_copies"] == 0 and st["pieces"] == pieces and st["gather_to"] == gather_to
            if gather_to is None or gather_to == rank:
                ok = ok and occ is not None and occ.shape == (RES, RES, RES) and np.array_equal(occ.numpy(), be.full)
                ok = ok and occ.is_contiguous() and occ.storage_offset() == 0 and occ.untyped_storage().nbytes() == (pa + pb) * world * RES * RES * 4
            else:
                ok = ok and occ is None
            want = ["features_a", "gather_signs_a", "features_b", "gather_signs_b", "wait_signs", "finish_a", "gather_volume_a", "finish_b", "gather_volume_b"] \
                if cmap_mode == "reference" else ["features_a", "finish_a", "gather_volume_a", "features_b", "finish_b", "gather_volume_b"]
            ok = ok and [o.replace("_async", "") for o in st["order"]] == want
            ok = ok and all(o.endswith("_async") for o in st["order"] if o.startswith("gather_"))
            mine = [p for p in pieces[rank] if p[1] > p[0]]
            feats = [c[1:3] for c in be.calls if c[0] == "features"]
            fins = [c[1:5] for c in be.calls if c[0] == "finish"]
            ok = ok and feats == mine and fins == [p + p for p in mine]
            # a second image on the same engine: fresh result buffer (the first volume is not overwritten), same protocol
            occ2 = recon(opt=None, netG=None, features=[torch.from_numpy(a.features)], proj_matrix=None)
            if occ is not None:
                ok = ok and occ2 is not None and occ2.data_ptr() != occ.data_ptr() and np.array_equal(occ.numpy(), be.full) and np.array_equal(occ2.numpy(), be.full)
            q.put((rank, bool(ok), [c[0] for c in be.calls] + ["ab"]))
            return
        ok = occ is not None and occ.shape == (RES, RES, RES) and np.array_equal(occ.numpy(), be.full)
        if skew:
            from icon_amd.recon import slab_partition, plane_weights
            z0, z1 = slab_partition(RES, world, plane_weights(RES, -0.9, -0.6))[rank]       # rank 0's view of the body
            ok = ok and recon.last_stats["slabs"][rank] == (z0, z1)
            occ2 = recon(opt=None, netG=None, features=[torch.from_numpy(a.features)], proj_matrix=None)   # cached cut: no broadcast
            ok = ok and np.array_equal(occ2.numpy(), be.full)
            be.calls = be.calls[: len(be.calls) // 2]
        else:
            z0, z1, _ = slab_bounds(RES, world, rank)
        kinds = [c[0] for c in be.calls]
        did_split = bool(recon.last_stats.get("split_features"))
        if did_split:
            # phase 1 per half-slab, each half's sign exchange asynchronous and enqueued BEFORE the next half's search; one wait
            # for both; then the halves are finished and their volume gathers started in turn
            order = recon.last_stats["order"]
            ok = ok and [o.replace("_async", "") for o in order] == ["features_a", "gather_signs_a", "features_b", "gather_signs_b", "wait_signs",
                                                                    "finish_a", "gather_volume_a", "finish_b", "gather_volume_b"]
            ok = ok and all(o.endswith("_async") for o in order if o.startswith("gather_"))     # CPU tensors under gloo: really asynchronous
            halves = sorted({c[1:3] for c in be.calls})
            ok = ok and halves[0][0] == z0 and halves[-1][1] == z1 and all(a[1] == b[0] for a, b in zip(halves[:-1], halves[1:]))
            ok = ok and kinds == ["features"] * kinds.count("features") + ["finish"] * kinds.count("finish") and 1 <= kinds.count("features") <= 2
            q.put((rank, bool(ok), kinds + ["split"]))
            return
        ok = ok and all(c[1:3] == (z0, z1) for c in be.calls)
        if legacy:
            ok = ok and (kinds == (["features", "finish", "eval"] if cmap_mode == "reference" else ["eval"]))
        else:
            # phase 1, then the slab finished in one piece (overlap off) or two halves of the padded slab (overlap on)
            pieces = [c[3:] for c in be.calls if c[0] == "finish"]
            ok = ok and kinds[0] == "features" and set(kinds[1:]) <= {"finish"}
            ok = ok and len(pieces) in ((1,) if not overlap else (1, 2))
            ok = ok and pieces[0][0] == z0 and pieces[-1][1] == z1 and all(a1 == b0 for (_, a1), (b0, _) in zip(pieces[:-1], pieces[1:]))
        q.put((rank, bool(ok), kinds))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cmap_mode,legacy,overlap,skew,split", [
    (2, "reference", False, True, False, False), (2, "reference", False, False, False, False), (2, "reference", True, True, False, False),
    (2, "local", False, True, False, False), (2, "local", True, True, False, False), (2, "reference", False, True, True, False),
    (2, "reference", False, True, False, True), (2, "reference", False, True, True, True), (2, "local", False, True, False, True),
    (2, "reference", False, False, False, True),
    # three ranks (uneven slabs) for the protocols with the most bookkeeping; the others differ from world 2 only in the numbers
    (3, "reference", False, True, False, False), (3, "reference", True, True, False, False), (3, "reference", False, True, True, True),
    (3, "local", False, True, False, True)])
def test_zslab_sharding_gloo(world, cmap_mode, legacy, overlap, skew, split):
    """packed sign messages, the two-half volume gather, the legacy host-side exchange, ranks whose replicas of
    the body disagree (rank 0's cut is broadcast), and - split - phase 1 per half-slab with ASYNCHRONOUS sign exchanges
    (the first half's all_gather in flight while the second half is searched): 2 x world messages interleaved into the
    very list of the unsplit protocol (the backend asserts the global list, the offsets and the workspace pairing)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cmap_mode, q, legacy, overlap, skew, split)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    if split and cmap_mode == "reference" and overlap:
        assert all(r[2][-1] == "split" for r in res), res          # the split protocol really ran on every rank


@pytest.mark.parametrize("world,cmap_mode,gather_to", [(2, "reference", None), (2, "local", None), (2, "reference", 1), (2, "local", 0),
                                                       (3, "reference", None), (3, "reference", 1), (4, "local", 0), (8, "reference", 5)])
def test_ab_layout_gloo(world, cmap_mode, gather_to):
    """the 'ab' slab layout (round 6, the default with the overlapped gather): two Z-slabs per rank - A_r, B_r - each a complete
    pipeline on its own workspace; lattice order of the pieces = A_0..A_{w-1}, B_0..B_{w-1} = message order of the two sign
    gathers in ONE buffer (the backend asserts the global outlier list, every piece's rank offset and the workspace pairing);
    each volume gather lands in one contiguous block of one fresh buffer whose first `res` planes ARE the result (no assembly
    copy: storage size, offset and contiguity asserted); gather_to: only the destination rank receives (dist.gather straight
    into its place) and forward() is None elsewhere.  World sizes with short and EMPTY last pieces (17 planes over 8 ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cmap_mode, q, False, True, False, True, "ab", gather_to)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert all(r[2][-1] == "ab" for r in res), res


def test_ab_pieces_partition():
    """every plane belongs to exactly one piece; no rank carries more than ceil(res / world) planes (what the contiguous cut gives
    its largest slab); the pieces in (A_0..A_{w-1}, B_0..B_{w-1}) order are the lattice order"""
    for res in (9, 17, 33, 65, 129, 257, 513):
        for world in (2, 3, 4, 5, 8, 16):
            if -(-res // world) < 2:
                continue
            pa, pb, pieces = DenseReconEngine.ab_pieces(res, world)
            flat = [p[0] for p in pieces] + [p[1] for p in pieces]
            planes = [z for a, b in flat for z in range(a, b)]
            assert planes == list(range(res)), (res, world)
            assert max(sum(b - a for a, b in p) for p in pieces) <= -(-res // world)
            assert pa + pb == -(-res // world) and 0 <= pa - pb <= 1


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("split", [True, False])
def test_zslab_sharding_gloo_many_ranks(world, split):
    """the rank counts of the scaling run (N = 4, 8; the driver measures 1 / 2 / 4 / 8): slabs of two or three planes at this
    resolution - half-slabs of one plane, second halves that are EMPTY on the ranks with the shorter slab - through the split
    and the unsplit protocol; every rank assembles the oracle's volume bit for bit and the backend asserts the global sign list"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "reference", q, False, True, False, split)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    if split:
        assert all(r[2][-1] == "split" for r in res), res


def standin_slab_mesh(buf, z0, res, zc0, zc1, halo, level):
    """A stand-in for the slab triangulation with marching cubes' OWNERSHIP structure (what the exchange protocol depends on): a cell
    owns the crossings of its +x / +y / +z edges (keys 3 * cell + direction, the halo layer only its x / y edges), and its faces -
    every run of three in [its own crossings, the x / y crossings of the cell ABOVE] - refer to vertices of the next layer.  Plain numpy loops; -> (keys, verts, faces) torch tensors, faces index the call's own vertices."""
    vol = buf.numpy() if hasattr(buf, "numpy") else np.asarray(buf)
    n = res - 1
    inside = lambda z: vol[z + 1 - z0][1:, 1:] > level                           # corner 0 of the cells of layer z: plane z + 1
    keys, verts, idx = [], [], {}
    for z in list(range(zc0, zc1)) + ([zc1] if halo else []):
        own = z < zc1
        P = inside(z)
        Pz = inside(z + 1) if (own and z + 1 < n) else None
        for y in range(n):
            for x in range(n):
                cell = (z * n + y) * n + x
                cand = [(0, x + 1 < n and P[y, x] != P[y, min(x + 1, n - 1)], (x + 0.5, y, z)),
                        (1, y + 1 < n and P[y, x] != P[min(y + 1, n - 1), x], (x, y + 0.5, z)),
                        (2, Pz is not None and P[y, x] != Pz[y, x], (x, y, z + 0.5))]
                for d, hit, pos in cand:
                    if hit:
                        idx[(cell, d)] = len(keys); keys.append(3 * cell + d); verts.append(pos)
    faces = []
    for z in range(zc0, zc1):
        if z + 1 >= n:
            continue
        for y in range(n):
            for x in range(n):
                here, above = (z * n + y) * n + x, ((z + 1) * n + y) * n + x
                lst = [idx[(here, d)] for d in (0, 1, 2) if (here, d) in idx] + [idx[(above, d)] for d in (0, 1) if (above, d) in idx]
                faces += [tuple(lst[k:k + 3]) for k in range(len(lst) - 2)]          # every run of three: many refer to the layer above
    return (torch.tensor(keys, dtype=torch.int64), torch.tensor(verts, dtype=torch.float32).reshape(-1, 3),
            torch.tensor(faces, dtype=torch.int64).reshape(-1, 3))


def _mesh_worker(rank, world, port, cmap_mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = assets("ico")
        be = OracleBackend(a, cmap_mode)
        be.cmap_mode = cmap_mode
        recon = DenseReconEngine(query_func=None, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                 resolutions=[9, RES], align_corners=True, backend=be)
        recon.slab_mesher = standin_slab_mesh
        out = recon.forward_mesh(opt=None, netG=None, features=[torch.from_numpy(a.features)], proj_matrix=None)
        wk, wv, wf = standin_slab_mesh(torch.from_numpy(be.full), 0, RES, 0, RES - 1, False, 0.5)      # the whole volume at once
        ok = out is not None and recon.last_stats.get("gather") == "mesh" and recon.last_stats["collectives"] == 3
        ok = ok and len(wf) > 30 and torch.equal(out[0], wv) and torch.equal(out[1], wf)
        q.put((rank, bool(ok), [tuple(out[0].shape), tuple(out[1].shape), tuple(wv.shape), tuple(wf.shape)] if out is not None else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cmap_mode", ["reference", "local"])
@pytest.mark.parametrize("world", [2, 3])
def test_mesh_exchange_protocol_gloo(cmap_mode, world):
    """DenseReconEngine.forward_mesh under gloo ranks on the CPU: slabs from the checker backend, the slab triangulation replaced
    by a stand-in with marching cubes' ownership structure (standin_slab_mesh) - what is under test is the PROTOCOL: the halo
    plane from the next rank, the cell-layer ranges, the sizes, the one packed message per rank, the merge by key.  The merged mesh
    equals the stand-in applied to the whole volume, vertex for vertex and face for face.  (The real triangulation behind the
    same protocol: tests/test_gpu_ties_shell.py::test_real_ranks_exchange_meshes_instead_of_the_volume.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mesh_worker, args=(r, world, port, cmap_mode, q)) for r in range(world)]
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
