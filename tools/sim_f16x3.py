"""numpy model of the 3x f16 split-precision MLP (icon_amd/csrc/mlp_f16x3.hip): weights as hi + lo f16 pieces of w * scale,
hidden activations as hi = f16_rtz(v), lo = f16_rne(v - hi), every product hi*hi + hi*lo + lo*hi accumulated in f32 -
with and without the packer's per-layer ACTIVATION scale.  Run on a checkpoint whose hidden layer is scaled by g (and the
next layer by 1/g: the same function): shows what tests/test_gpu_parity.py::test_hidden_activations_below_the_f16_normal_range
checks on the GPU, without one.   python tools/sim_f16x3.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icon_amd import synth          # noqa: E402
from oracle import oracle as orc    # noqa: E402


def f16_rtz(v):
    h = v.astype(np.float16)
    hf = h.astype(np.float32)
    over = np.abs(hf) > np.abs(v)
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float16)


def split_act(v):
    hi = f16_rtz(v)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def split_w(w):
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def pick_scale(W):
    mx = np.abs(W).max()
    return np.float32(2.0 ** int(np.clip(np.floor(np.log2(8192.0 / mx)), -60, 60)))


def act_scale(W, b, in_ms):
    ms = (W.astype(np.float64) ** 2 * in_ms[None, :]).sum(1) + b.astype(np.float64) ** 2
    t = np.sqrt(ms.mean())
    return np.float32(2.0 ** int(np.clip(np.rint(np.log2(16.0 / t)), -60, 60))), ms


def mm3(Wh, Wl, xh, xl):
    return (Wh @ xh + Wh @ xl + Wl @ xh).astype(np.float32)


def leaky(a, inv):
    return (np.float32(0.495) * inv * np.abs(a) + (np.float32(0.505) * inv) * a).astype(np.float32)


def forward(sd, x, use_act_scale):
    W, B = [], []                                     # BatchNorm folded in float64, as icon_mlp_create does
    for l in range(4):
        w = np.asarray(sd[f"filters.{l}.weight"], np.float64).reshape(sd[f"filters.{l}.weight"].shape[0], -1)
        b = np.asarray(sd[f"filters.{l}.bias"], np.float64)
        if l < 3:
            k = np.asarray(sd[f"norms.{l}.weight"], np.float64) / np.sqrt(np.asarray(sd[f"norms.{l}.running_var"], np.float64) + 1e-5)
            w = w * k[:, None]
            b = (b - np.asarray(sd[f"norms.{l}.running_mean"], np.float64)) * k + np.asarray(sd[f"norms.{l}.bias"], np.float64)
        W.append(w.astype(np.float32)); B.append(b.astype(np.float32))
    c0 = W[0].shape[1]
    A0, ms0 = act_scale(W[0], B[0], np.ones(c0))
    A1, _ = act_scale(W[1], B[1], ms0)
    if not use_act_scale:
        A0 = A1 = np.float32(1.0)
    W1 = W[1] / A0
    W2 = W[2].copy(); W2[:, :256] /= A1
    s0, s1, s2 = pick_scale(W[0]), pick_scale(W1), pick_scale(W2)
    xT = x.T.astype(np.float32)
    xh, xl = split_act(xT)
    h, l = split_w(W[0] * s0)
    a0 = mm3(h, l, xh, xl) + (B[0] * s0)[:, None]
    v0 = leaky(a0, A0 / s0)
    bh, bl = split_act(v0)
    h, l = split_w(W1 * s1)
    a1 = mm3(h, l, bh, bl) + (B[1] * s1)[:, None]
    v1 = leaky(a1, A1 / s1)
    bh, bl = split_act(v1)
    h, l = split_w(W2 * s2)
    a2 = mm3(h[:, :256], l[:, :256], bh, bl) + mm3(h[:, 256:], l[:, 256:], xh, xl) + (B[2] * s2)[:, None]
    v2 = leaky(a2, np.float32(1.0) / s2)
    y = W[3][:, :128] @ v2 + W[3][:, 128:] @ xT + B[3][:, None]
    return y[0], (np.abs(v0).mean(), np.abs(v1).mean())


if __name__ == "__main__":
    x = synth.representative_rows(8192, 13, seed=11)
    for layer in (0, 1):
        for g in (1.0, 1e-3, 1e-4, 1e-5, 1e3):
            sd = {k: v.copy() for k, v in synth.make_mlp_state_dict(seed=synth.SEED + 7, sdf_gain=8.0, learned_std=0.5).items()}
            sd[f"norms.{layer}.weight"] = (sd[f"norms.{layer}.weight"] * g).astype(np.float32)
            sd[f"norms.{layer}.bias"] = (sd[f"norms.{layer}.bias"] * g).astype(np.float32)
            w = sd[f"filters.{layer + 1}.weight"].copy()
            nh = 512 if layer == 0 else 256
            w[:, :nh] = (w[:, :nh] / g).astype(np.float32)
            sd[f"filters.{layer + 1}.weight"] = w
            ref = orc.Mlp(sd).forward(x, f64=True)[:, 0]
            out = []
            for use in (False, True):
                y, mags = forward(sd, x, use)
                out.append(f"{'with' if use else 'without'} activation scale: max err {np.abs(y - ref).max():.2e} (split operands ~{mags[layer]:.1e})")
            print(f"layer {layer} g {g:g}: " + "; ".join(out))
