"""GPU tests of round 6: the static-plus-stealing tile partition of the fused MLP kernel (any split of the tiles into a
static part and a pool drawn in groups gives the same volume bit for bit, the ticket pair cleans itself), the per-workgroup
profile records (icon_work_profile_workgroups)."""
import numpy as np
import pytest
import torch

from common import assets, vol_assets
from icon_amd import synth
from test_gpu_parity import T, dev, make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def body():
    return assets("body")


def bits(t):
    return t.contiguous().view(torch.int32)


# ---------------------------------------------------------------------------------------------
# k_fused_f16x3: static runs + a pool of tiles drawn by whoever finishes first
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cmap_mode", ["reference", "local"])
def test_steal_partition_is_free(body, cmap_mode):
    """a tile's result does not depend on which workgroup evaluates it or when: every (pool share, group size) reproduces the
    all-static volume of rounds 1-5 bit for bit - incl. everything pooled (no static run: the first group is drawn before
    the loop), one-tile groups (a draw per tile), groups larger than a span's pool part (one draw takes it whole), a pool share
    that rounds to no tile at all (static), and a slab that is not the whole lattice.  NaN-prefilled outputs: a tile nobody evaluated would show."""
    feat = T(body.features)
    eng = make_engine(body, cmap_mode=cmap_mode)
    res = 129                                     # 127^3 interior points = 8,002 tiles of 256 > 256 workgroups
    w = eng._work()
    w.set_steal(0, 1)
    want = eng.eval_slab(feat, res, 0, res).clone()
    want_part = eng.eval_slab(feat, res, 40, 97).clone()
    assert float(want.max()) > 0.5 and torch.isfinite(want).all()
    for permille, group in [(100, 2), (100, 1), (37, 3), (500, 8), (1000, 1), (1000, 7), (1000, 127), (999, 2), (1, 1), (400, 127)]:
        w.set_steal(permille, group)
        for rep in range(2):                      # twice: the second launch finds the ticket pair as the first left it (clean)
            out = torch.full((res, res, res), float("nan"), device=dev())
            eng.eval_slab(feat, res, 0, res, out=out)
            assert torch.equal(bits(out), bits(want)), (permille, group, rep)
        out = torch.full((57, res, res), float("nan"), device=dev())
        eng.eval_slab(feat, res, 40, 97, out=out)
        assert torch.equal(bits(out), bits(want_part)), (permille, group, "part")
    w.set_steal(150, 2)


@pytest.mark.parametrize("prior", ["pamir", "pifu"])
def test_steal_partition_is_free_other_priors(prior):
    """the pamir / pifu instantiations of the kernel share the loop"""
    from icon_amd.engine import IconQueryEngine
    feat, vol, sd = vol_assets(prior)
    eng = IconQueryEngine(prior_type=prior)
    if vol is not None:
        eng.set_volume_features(T(vol))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in sd.items()})
    f = T(feat)
    res = 129
    w = eng._work()
    w.set_steal(0, 1)
    want = eng.eval_slab(f, res, 0, res).clone()
    assert torch.isfinite(want).all() and float(want.abs().max()) > 0
    for permille, group in [(100, 2), (1000, 1), (333, 5)]:
        w.set_steal(permille, group)
        out = torch.full((res, res, res), float("nan"), device=dev())
        eng.eval_slab(f, res, 0, res, out=out)
        assert torch.equal(bits(out), bits(want)), (permille, group)


def test_steal_with_reserved_cus_and_split_slab(body):
    """a smaller grid (icon_work_set_reserve_cus) and the split slab protocol's piecewise launches draw from the same pair"""
    feat = T(body.features)
    eng = make_engine(body)
    res = 129
    w = eng._work()
    w.set_steal(0, 1)
    want = eng.eval_slab(feat, res, 0, res).clone()
    w.set_steal(250, 3)
    for reserve in (16, 200, 255):
        w.set_reserve_cus(reserve)
        out = torch.full((res, res, res), float("nan"), device=dev())
        eng.eval_slab(feat, res, 0, res, out=out)
        assert torch.equal(bits(out), bits(want)), reserve
    w.set_reserve_cus(0)
    eng.slab_features(feat, res, 0, res)
    out = torch.full((res, res, res), float("nan"), device=dev())
    for za, zb in [(0, 50), (50, 51), (51, 129)]:
        eng.slab_finish_gathered(res, 0, res, None, 0, 1, 0, out=out, za=za, zb=zb)
    assert torch.equal(bits(out), bits(want))
    w.set_steal(150, 2)


def test_workgroup_profile_records(body):
    """icon_work_profile_workgroups: one record per workgroup of the persistent grid - its XCD, when it started, how long it
    ran, its shader cycles, the tiles it evaluated.  The tiles add up to the launch's; with a pool the counts differ between
    workgroups, all static they differ by at most one; block b runs on XCD b mod 8 (observed placement, not a contract:
    only that all 8 XCDs appear is asserted)."""
    feat = T(body.features)
    eng = make_engine(body)
    res = 161
    ntiles = -(-(res - 2) ** 3 // 256)
    w = eng._work()
    w.profile(True)
    try:
        for permille in (0, 100):
            w.set_steal(permille, 2)
            eng.eval_slab(feat, res, 0, res)
            rec = w.profile_workgroups()
            d = w.profile_detail()
            assert rec.shape[1] == 5 and rec.shape[0] >= 64
            assert int(rec[:, 4].sum()) == ntiles, (permille, rec[:, 4].sum(), ntiles)
            assert set(rec[:, 0].astype(int)) == set(range(8))
            assert (rec[:, 2] > 0).all() and (rec[:, 3] > 0).all() and (rec[:, 1] >= 0).all() and rec[:, 1].min() == 0
            mhz = rec[:, 3] / rec[:, 2] * 1e-3
            assert (mhz > 300).all() and (mhz < 3000).all(), (mhz.min(), mhz.max())
            # workgroup 0's record is the one profile_detail reports
            assert abs(rec[0, 3] - d["fused_cycles"]) <= 1e-3 * d["fused_cycles"] + 2000
            if permille == 0:
                assert rec[:, 4].max() - rec[:, 4].min() <= 1
            else:
                per = ntiles // rec.shape[0]                     # every workgroup evaluates the static part of its own span itself
                assert rec[:, 4].min() >= per - per * permille // 1000
    finally:
        w.profile(False)
        w.set_steal(150, 2)


# ---------------------------------------------------------------------------------------------
# a shared-walk search that gave up is reported by the forward() that used it
# ---------------------------------------------------------------------------------------------
def test_dense_forward_reports_a_lost_hand_over_with_its_own_image(body):
    """reconEngine.forward synchronises for its None test and then polls the mesh AND the workspaces: with the injected lost
    push (tests/test_gpu_round5.py) the dense 33^3 forward of THIS image raises - before, the suspect volume was returned and
    the NEXT image's call was refused instead - and the engine works again once the fault is off."""
    from types import SimpleNamespace
    from icon_amd.engine import IconAmdError, query_func
    from icon_amd.recon import DenseReconEngine
    from test_gpu_round5 import set_option
    feat = [T(body.features)]
    eng = make_engine(body)
    recon = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[33],
                             align_corners=True, engine=eng).to(dev())
    opt = SimpleNamespace(num_views=1)
    want = recon(opt=opt, netG=eng, features=feat, proj_matrix=None).clone()
    set_option("share_spin_log2", 10)
    set_option("share_lose_push", 3)
    try:
        with pytest.raises(IconAmdError, match="shared-walk search"):
            recon(opt=opt, netG=eng, features=feat, proj_matrix=None)
    finally:
        set_option("share_lose_push", 0)
        set_option("share_spin_log2", 0)
    got = recon(opt=opt, netG=eng, features=feat, proj_matrix=None)          # the report cleared the record: no refusal one image late
    assert torch.equal(bits(got), bits(want))


# ---------------------------------------------------------------------------------------------
# the one-command real-package harness, with the repo's CPU checkers standing in for the packages
# ---------------------------------------------------------------------------------------------
def test_real_package_harness_self_test():
    """tools/parity_real_packages.py --stand-ins: every section of the harness (distance / sign, vertex normals, marching cubes,
    clean_mesh, visibility, voxeliser) runs the product leaf through the C ABI and compares it with the oracle's restatement in
    the package's place - all PASS, so a DIFF printed where the real packages exist is the package's, not the harness's"""
    import os
    import subprocess
    import sys
    from common import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_real_packages.py"), "--stand-ins", "--res", "49", "--strict"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " 0 DIFF, 0 ABSENT" in p.stdout and p.stdout.count("[PASS") >= 8, p.stdout[-3000:]


# ---------------------------------------------------------------------------------------------
# device marching cubes at the SHIPPED size against the INDEPENDENT classic implementation
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["schedule", "dense"])
def test_device_marching_cubes_513_vs_classic_marching_cubes(body, which):
    """mcube_res = 512 (configs/icon-filter.yaml:23): lib/common/seg3d_lossless.py:587-596 hands occupancys[1:,1:,1:] to PyMCubes.
    Until round 5 the device mesh at 513^3 was compared with the repo's OWN host implementation + table-free invariants only; here
    the published classic algorithm (oracle/mc_classic.py: Lorensen-Cline / Bourke table, read as PyMCubes reads the array, run
    block-wise over the blocks that hold a crossing) triangulates the same 513^3 volumes - the reference schedule's and the dense
    one - and the device mesh must be the same vertex set and the same triangle set, winding included, after the reference's
    own [:, [2,1,0]] / [:, [0,2,1]] conventions."""
    import os
    import sys
    from types import SimpleNamespace
    from common import ROOT
    from icon_amd.recon import DenseReconEngine
    from oracle import mc_classic
    from test_gpu_round5 import recon513
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from parity_real_packages import compare_meshes
    eng = make_engine(body)
    feat = T(body.features)
    if which == "schedule":
        occ = recon513(eng)(opt=SimpleNamespace(num_views=1), netG=eng, features=[feat], proj_matrix=None)
    else:
        occ = eng.eval_slab(feat, 513, 0, 513)
    vo, fo = DenseReconEngine(resolutions=[513], align_corners=True).export_mesh(occ)       # the product, reference conventions
    final = occ[1:, 1:, 1:].contiguous().cpu().numpy()
    cv, cf = mc_classic.marching_cubes_blocks(final, 0.5, set_below=True, block=64)
    cv, cf = cv[:, [2, 1, 0]], cf[:, [0, 2, 1]]                                               # seg3d_lossless.py:594-596
    assert len(fo) > 400000
    # (coordinates run to 512: a float32 ulp there is 6e-5 - the device interpolates in float32, the classic code in float64)
    c = compare_meshes(vo.numpy(), fo.numpy(), cv, cf, tol=1e-4, lattice=True)
    print(c)
    assert c["verts"][0] == c["verts"][1] and c["faces"][0] == c["faces"][1], c
    assert c["same_vertex_set"] and c["offset"] is None, c
    assert c["only_ours"] == 0 and c["only_theirs"] == 0 and c["flipped"] == 0, c
    assert c["max_vertex_diff"] <= 4e-4, c


# ---------------------------------------------------------------------------------------------
# the reference schedule as one native call: launch diet (26 launches), deferred range rescue
# ---------------------------------------------------------------------------------------------
def test_schedule_defers_the_range_rescue_and_reruns_when_it_fires(body):
    """icon_adaptive_eval with counts synchronises at its end: the fused kernels of its levels raise ONE sticky word and no
    k_rescue_fused is launched behind them; a raised word (operands beyond the f16 range - here a body squashed flat, whose
    sliver triangles extrapolate |norm| to 1e5, tests/test_gpu_parity.py) runs the schedule again the per-launch way.  The
    volume equals the host-driven schedule's (per-call rescue) and is finite; an ordinary subject never reruns."""
    import warnings
    from types import SimpleNamespace
    from icon_amd.engine import IconQueryEngine, query_func
    from icon_amd.recon import AdaptiveReconEngine
    feat = T(body.features)
    eng = make_engine(body)
    vol, counts, pos = eng.adaptive_eval(feat, [17, 33, 65])
    assert eng._work().adaptive_reruns() == 0 and pos and torch.isfinite(vol).all()
    v = (body.smpl_verts * np.asarray((1.0, 1.0, 1e-4), np.float32)).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flat = IconQueryEngine(prior_type="icon", sdf_clip=body.sdf_clip)
        flat.set_mesh(T(v), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
        flat.set_regressor({k: torch.from_numpy(w) for k, w in body.state_dict.items()})
        got, counts, pos = flat.adaptive_eval(feat, [9, 17, 33])
        reruns = flat._work().adaptive_reruns()
        kw = dict(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[9, 17, 33], align_corners=True, faster=True)
        host = AdaptiveReconEngine(**kw).to(dev())
        host.native = False
        want = host(opt=SimpleNamespace(num_views=1), netG=flat, features=[feat], proj_matrix=None)
    assert torch.isfinite(got).all()
    assert reruns == 1, reruns                                  # the case really exercises the range path
    assert host.last_stats["queries"][0] == 9 ** 3 == int(counts[0])
    if want is not None:
        assert (got - want).abs().max().item() <= 1e-6 * max(1.0, float(want.abs().max()))
    # asynchronous form (no counts): the per-launch rescue stays - same volume
    got2, _, _ = flat.adaptive_eval(feat, [9, 17, 33], counts=False)
    torch.cuda.synchronize()
    assert torch.equal(bits(got2), bits(got)) and flat._work().adaptive_reruns() == 1


# ---------------------------------------------------------------------------------------------
# the None rule of reconEngine.forward as one native kernel
# ---------------------------------------------------------------------------------------------
def test_none_rule_kernel_equals_the_torch_expression():
    """Seg3dLossless._forward_faster returns None when nothing exceeds 0.5 on its COARSEST lattice (seg3d_lossless.py:173-177) -
    for a dense volume the sub-lattice of stride (res_last - 1) / (res_first - 1).  icon_volume_any_above against
    `(occ[::s, ::s, ::s] > 0.5).sum() == 0`: volumes whose only voxel above the level sits ON the sub-lattice (first, last, inner
    point) and one step OFF it (the rule must not see it), an all-below volume, exactly 0.5 (not above)."""
    from icon_amd.recon import DenseReconEngine
    for res_list in ([33, 65, 129], [17, 129], [65]):
        rec = DenseReconEngine(resolutions=res_list, align_corners=True).to(dev())
        r, s = res_list[-1], (res_list[-1] - 1) // max(res_list[0] - 1, 1)
        base = torch.full((r, r, r), 0.25, device=dev())
        cases = [((0, 0, 0), True), ((r - 1, r - 1, r - 1), True), ((s, 2 * s % r, 0), True), ((r - 1, 0, s), True)]
        if s > 1:
            cases += [((1, 0, 0), False), ((s, s, s + 1), False), ((s - 1, s, s), False)]
        for (z, y, x), seen in cases:
            vol = base.clone()
            vol[z, y, x] = 0.75
            want = not bool((vol[::s, ::s, ::s] > 0.5).sum() == 0)
            assert want == seen, (res_list, z, y, x)
            got = rec._none_if_empty(vol)
            assert (got is not None) == seen, (res_list, z, y, x)
            assert got is None or got.data_ptr() == vol.data_ptr()
        assert rec._none_if_empty(base) is None
        edge = base.clone(); edge[0, 0, 0] = 0.5
        assert rec._none_if_empty(edge) is None
