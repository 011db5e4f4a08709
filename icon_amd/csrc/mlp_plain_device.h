// The MLP in plain f32 for ONE point, one wavefront: the range safety net of the split-precision kernels.
//
// The f16x3 path carries every operand as f16 pieces: an input or a hidden activation beyond the f16 range (65504 -
// three orders of magnitude above anything a body mesh produces, but reachable with a degenerate one: the unclamped
// barycentric extrapolation of mesh_util.py:319-354 on a sliver triangle) becomes inf and the point's occupancy NaN, where
// the reference's f32 MLP returns a number.  Those kernels raise a flag when an in-cube result is not finite; k_rescue_*
// then recompute exactly those points here: folded weights in [Cout][Cin] f32, k-ordered fma chains (lib/net/MLP.py:49-72
// with BatchNorm folded, LeakyReLU 0.01, res layers [2,3]).  Costs one 3-microsecond launch that reads one word when the
// flag is down.
#pragma once
#include "common.h"

namespace icon {

struct MlpPlain {
    const float *w0, *b0;     // [512][16] (input slots >= c0 are zero), [512]
    const float *w1, *b1;     // [256][512], [256]
    const float *w2, *b2;     // [128][256 + 16], [128]
    const float *w3;          // [128 + 16]
    float b3;
    int last_op;
    int c0;                   // row slots >= c0 are not inputs (they may hold anything)
};

MlpPlain mlp_plain_of(const icon_mlp *m);      // mlp_kernels.hip

constexpr int kPlainFloats = 512 * 16 + 512 + 256 * 512 + 256 + 128 * 272 + 128 + 144;
constexpr int kPlainLds = 16 + 512 + 256 + 128;      // floats of LDS scratch per wave: x, h0, h1, h2

// all 64 lanes of one wave; x[16] (slots >= c0 zero) already in LDS `s`; returns the pre-mask network output in every lane
__device__ __forceinline__ float mlp_plain_wave(const MlpPlain &P, float *s, int lane)
{
    float *x = s, *h0 = s + 16, *h1 = h0 + 512, *h2 = h1 + 256;
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 8; ++r) {
        const int o = lane + 64 * r;
        const float *w = P.w0 + o * 16;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(w[k], x[k], acc);
        acc += P.b0[o];
        h0[o] = acc < 0.0f ? 0.01f * acc : acc;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 4; ++r) {
        const int o = lane + 64 * r;
        const float *w = P.w1 + (size_t)o * 512;
        float acc = 0.0f;
        for (int k = 0; k < 512; ++k) acc = fmaf(w[k], h0[k], acc);
        acc += P.b1[o];
        h1[o] = acc < 0.0f ? 0.01f * acc : acc;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 2; ++r) {
        const int o = lane + 64 * r;
        const float *w = P.w2 + (size_t)o * 272;
        float acc = 0.0f;
        for (int k = 0; k < 256; ++k) acc = fmaf(w[k], h1[k], acc);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(w[256 + k], x[k], acc);
        acc += P.b2[o];
        h2[o] = acc < 0.0f ? 0.01f * acc : acc;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float part = fmaf(P.w3[lane], h2[lane], P.w3[lane + 64] * h2[lane + 64]);
    if (lane < 16) part = fmaf(P.w3[128 + lane], x[lane], part);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    return apply_last_op(part + P.b3, P.last_op);
}

}  // namespace icon
