"""numpy model of the mixed f16 + MX-fp6 MLP arithmetic (mlp_mx6.hip) - predicts its error against
the float64 MLP before/independently of the kernel.  Test infrastructure (uses oracle/)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icon_amd import synth
from oracle import oracle as orc

RHO = [np.array([(t & 3) + 8 * (t >> 2) + 4 * h for t in range(16)]) for h in range(2)]


def fold(sd):
    Ws, Bs = [], []
    for l in range(4):
        W = sd[f"filters.{l}.weight"][:, :, 0].astype(np.float64)
        b = sd[f"filters.{l}.bias"].astype(np.float64)
        if l != 3:
            g = sd[f"norms.{l}.weight"].astype(np.float64); be = sd[f"norms.{l}.bias"].astype(np.float64)
            m = sd[f"norms.{l}.running_mean"].astype(np.float64); v = sd[f"norms.{l}.running_var"].astype(np.float64)
            s = g / np.sqrt(v + 1e-5)
            W = W * s[:, None]; b = (b - m) * s + be
        Ws.append(W.astype(np.float32)); Bs.append(b.astype(np.float32))
    return Ws, Bs


def q6(v, e):
    """e2m3 quantisation of v / 2^e (round-to-nearest-even, saturate 7.5), returned in v's units"""
    s = np.ldexp(1.0, np.asarray(e).astype(np.int64))
    a = np.abs(v) / s
    ee = np.clip(np.floor(np.log2(np.maximum(a, 1.0))), 0, 2)
    step = np.ldexp(1.0, (ee - 3).astype(np.int64))
    qa = np.minimum(np.round(a / step) * step, 7.5)
    return np.sign(v) * qa * s


def block_exp(v, axis):
    """floor(log2(max|v|)) per block (MX: shared exponent), -127 for all-zero blocks"""
    m = np.max(np.abs(v), axis=axis, keepdims=True)
    return np.where(m > 0, np.floor(np.log2(np.maximum(m, 1e-300))), -127.0)


def kblocks(cin_hidden):
    """K index sets: one 64-channel chunk = two accumulator tiles; lane half hh of both tiles = one 32-element block"""
    out = []
    for c in range(0, cin_hidden, 64):
        for hh in range(2):
            out.append(np.concatenate([c + RHO[hh], c + 32 + RHO[hh]]))
    return out


def f16(x, rtz=False):
    if not rtz:
        return x.astype(np.float16).astype(np.float64)
    h = x.astype(np.float16).astype(np.float64)
    over = np.abs(h) > np.abs(x)
    hb = x.astype(np.float16).view(np.uint16).copy()
    hb[over] -= 1
    return hb.view(np.float16).astype(np.float64)


def layer(W, b, h, x16, wscale, mode, rtz, lo_shift, true_lo_max):
    """W [M, Kh (+c0)], h [Kh, N] f32 activations, x16 raw inputs [c0, N] or None"""
    Kh = h.shape[0]
    Ws = W.astype(np.float64) * wscale
    Whi = f16(Ws); Wlo = Ws - Whi
    acc = (b.astype(np.float64) * wscale)[:, None] + np.zeros((W.shape[0], h.shape[1]))
    if x16 is not None:
        acc += Ws[:, Kh:] @ x16.astype(np.float64)            # f16x3 K-step: ~exact
    hh = h.astype(np.float64)
    hhi = f16(hh, rtz); hlo = hh - hhi
    acc += Whi[:, :Kh] @ hhi
    if mode == "f16x3":
        acc += Whi[:, :Kh] @ f16(hlo) + f16(Wlo[:, :Kh]) @ hhi
        return acc / wscale
    for kb in kblocks(Kh):
        Wb, Wlb, hb, hlb = Ws[:, kb], Wlo[:, kb], hh[kb], hlo[kb]
        eW, eWl = block_exp(Wb, 1) - 2, block_exp(Wlb, 1) - 2
        eh = block_exp(hb, 0) - 2
        ehl = (block_exp(hlb, 0) - 2) if true_lo_max else eh - lo_shift
        acc += q6(Wb, eW) @ q6(hlb, ehl) + q6(Wlb, eWl) @ q6(hb, eh)
    return acc / wscale


def pick_scale(W):
    mx = float(np.abs(W).max())
    return np.ldexp(1.0, int(np.clip(np.floor(np.log2(8192.0 / mx)), -12, 24)))


def forward(Ws, Bs, X, mode="mx6", rtz=False, lo_shift=11, true_lo_max=False):
    x = X.T.astype(np.float32)                                 # [c0, N]
    leaky = lambda y: np.where(y > 0, y, 0.01 * y).astype(np.float32)
    h0 = leaky((Ws[0].astype(np.float64) @ x + Bs[0][:, None]))            # layer 0: f16x3, ~exact
    h1 = leaky(layer(Ws[1], Bs[1], h0, None, pick_scale(Ws[1]), mode, rtz, lo_shift, true_lo_max))
    h2 = leaky(layer(Ws[2], Bs[2], h1, x, pick_scale(Ws[2]), mode, rtz, lo_shift, true_lo_max))
    y = Ws[3].astype(np.float64) @ np.concatenate([h2, x], 0) + Bs[3][:, None]
    return y[0]


def main():
    a = synth.make_assets("body")
    n = int(os.environ.get("N", 60000))
    pts = np.concatenate([synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], n // 2), synth.lattice_points(33)[: n // 2]])
    _, X = orc.query_icon(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], a.features, orc.Mlp(a.state_dict), pts,
                          sdf_clip=a.sdf_clip)
    ref = synth.mlp_forward_f64(a.state_dict, X.T)[0]
    Ws, Bs = fold(a.state_dict)
    for name, kw in [("f16x3 rtz", dict(mode="f16x3", rtz=True)),
                     ("mx6 rtn lo=E-11 (bound 4)", dict(lo_shift=11)),
                     ("mx6 rtn lo=E-12 (bound 8, clips)", dict(lo_shift=12)),
                     ("mx6 rtn true lo max", dict(true_lo_max=True)),
                     ("mx6 rtz lo=E-10", dict(rtz=True, lo_shift=10))]:
        y = forward(Ws, Bs, X, **kw)
        e = np.abs(y - ref)
        print(f"{name:36s} max {e.max():.3e}  mean {e.mean():.3e}  p99.9 {np.quantile(e, 0.999):.3e}")


if __name__ == "__main__":
    main()
