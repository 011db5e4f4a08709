// mlp_kernels.hip - the per-point occupancy MLP of ICON as one fused, register-resident MFMA chain.
//
// Replaces lib/net/MLP.py:49-72 as instantiated by lib/net/HGPIFuNet.py:128-133
// (13 -> 512 -> 256 -> (256+13) -> 128 -> (128+13) -> 1, BatchNorm1d(eval) + LeakyReLU(0.01) after
// the first three Conv1d(k=1), raw input re-concatenated before layers 2 and 3, no last_op in
// test mode) and the in_cube mask of lib/net/HGPIFuNet.py:363.
//
// Design (gfx950, wave64):
//  * out[channels x points] = W[channels x K] * act[K x points], v_mfma_f32_32x32x2_f32
//    (f32 in, f32 accumulate: bit-for-bit a k-ordered fmaf chain, 157 TFLOP/s peak).
//  * ONE WAVEFRONT OWNS 32 POINTS AND CARRIES THEM THROUGH ALL FOUR LAYERS IN REGISTERS.
//    The C/D fragment of a 32x32 MFMA leaves lane (j = lane&31, h = lane>>5) holding, for point j,
//    rows rho(t) + 4h, rho(t) = (t&3) + 8(t>>2), t = 0..15.  The B operand of the next layer's
//    k-step needs lane (j, h) to supply activation k = 2*step + h of point j.  The k order of a
//    GEMM is free, so the host packs every weight matrix with its K axis PERMUTED such that
//    k-step (32-tile m, register t) multiplies rows {32m + rho(t), 32m + rho(t) + 4}: the
//    accumulator register t of tile m IS the B operand of that k-step.  No LDS, no shuffles, no
//    barriers between layers; bias is the accumulator's initial value; LeakyReLU is one v_mul +
//    v_max per register.
//  * The 512-wide hidden layer is never materialised: layer 0 is produced 32 channels at a time
//    and consumed immediately as 16 k-steps of layer 1 (76 % of the FLOPs).
//  * Weights (689 KB, BatchNorm folded on the host in float64) stream from L2 as 16-byte loads
//    laid out [..][lane][4] so each wave-load is one contiguous KiB.
//  * Layer 3 (141 -> 1) is 72 VALU fmas per lane plus one cross-half exchange.
#include "common.h"
#include "mlp_plain_device.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace icon {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMlpBlock = 256;             // 4 waves x 32 points
constexpr int kPtsPerWave = 32;
constexpr int kPtsPerBlock = (kMlpBlock / 64) * kPtsPerWave;

// blob sizes (floats)
constexpr size_t kW0 = 16 * 2 * 64 * 4;        // [c16][sq2][lane][4]
constexpr size_t kB0 = 16 * 2 * 16;            // [c16][h][16]
constexpr size_t kW1 = 16 * 4 * 8 * 64 * 4;    // [c16][tq4][m8][lane][4]
constexpr size_t kB1 = 8 * 2 * 16;             // [m8][h][16]
constexpr size_t kW2 = 8 * 4 * 4 * 64 * 4;     // [m8][tq4][m2_4][lane][4]
constexpr size_t kW2x = 2 * 4 * 64 * 4;        // [sq2][m2_4][lane][4]
constexpr size_t kB2 = 4 * 2 * 16;             // [m2_4][h][16]
constexpr size_t kW3 = 2 * 72;                 // [h][64 + 8]

struct MlpDev {
    const float *w0, *b0, *w1, *b1, *w2, *w2x, *b2, *w3;
    float b3;
    int c0;
    int last_op;
};

__device__ __forceinline__ f32x16 leaky(f32x16 v)
{
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = fmaxf(v[t], 0.01f * v[t]);
    return v;
}

__device__ __forceinline__ f32x16 load16(const float *p)
{
    const float4 *q = reinterpret_cast<const float4 *>(p);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    f32x16 v;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
    return v;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// MASK: multiply by the in_cube bit of the row's code word (query path); otherwise plain MLP.forward
template <bool MASK>
__global__ __launch_bounds__(kMlpBlock, 2) void k_mlp_f32(const float *__restrict__ X, int64_t N, float *__restrict__ out, MlpDev w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t base = ((int64_t)blockIdx.x * (kMlpBlock / 64) + wave) * kPtsPerWave;
    if (base >= N) return;                       // wave-uniform; the kernel has no barriers
    const int64_t pi = min(base + j, N - 1);

    // B operand of layer 0 and of the skip connections: lane (j,h) holds slots 8h..8h+7 of point j
    float xr[8];
    {
        const float4 *q = reinterpret_cast<const float4 *>(X + pi * kXRow + 8 * h);
        const float4 a = q[0], b = q[1];
        xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = b.x; xr[5] = b.y; xr[6] = b.z; xr[7] = b.w;
    }
    float maskf = 1.0f;
    if (MASK) {
        const uint32_t code = (uint32_t)__float_as_int(X[pi * kXRow + kCodeSlot]);
        maskf = (code & kCodeInCube) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) xr[s] = (s + 8 * h < w.c0) ? xr[s] : 0.0f;   // pad slots / code word never reach the GEMM

    // ---- layers 0 + 1 fused: 16 chunks of 32 hidden channels -------------------------------
    f32x16 acc1[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc1[m] = load16(w.b1 + (m * 2 + h) * 16);

    const float4 *w0 = reinterpret_cast<const float4 *>(w.w0) + lane;
    const float4 *w1 = reinterpret_cast<const float4 *>(w.w1) + lane;
    for (int c = 0; c < 16; ++c) {
        f32x16 h0 = load16(w.b0 + (c * 2 + h) * 16);
        const float4 a0 = w0[(c * 2 + 0) * 64], a1 = w0[(c * 2 + 1) * 64];
        h0 = MFMA(a0.x, xr[0], h0); h0 = MFMA(a0.y, xr[1], h0); h0 = MFMA(a0.z, xr[2], h0); h0 = MFMA(a0.w, xr[3], h0);
        h0 = MFMA(a1.x, xr[4], h0); h0 = MFMA(a1.y, xr[5], h0); h0 = MFMA(a1.z, xr[6], h0); h0 = MFMA(a1.w, xr[7], h0);
        h0 = leaky(h0);
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            float4 a[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = w1[((c * 4 + tq) * 8 + m) * 64];
#pragma unroll
            for (int m = 0; m < 8; ++m) acc1[m] = MFMA(a[m].x, h0[4 * tq + 0], acc1[m]);
#pragma unroll
            for (int m = 0; m < 8; ++m) acc1[m] = MFMA(a[m].y, h0[4 * tq + 1], acc1[m]);
#pragma unroll
            for (int m = 0; m < 8; ++m) acc1[m] = MFMA(a[m].z, h0[4 * tq + 2], acc1[m]);
#pragma unroll
            for (int m = 0; m < 8; ++m) acc1[m] = MFMA(a[m].w, h0[4 * tq + 3], acc1[m]);
        }
    }

    // ---- layer 2: K = 256 (layer-1 output, in registers) + 16 (raw input) ---------------------
    f32x16 acc2[4];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = load16(w.b2 + (m2 * 2 + h) * 16);
    const float4 *w2 = reinterpret_cast<const float4 *>(w.w2) + lane;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const f32x16 h1 = leaky(acc1[m]);
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            float4 a[4];
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) a[m2] = w2[((m * 4 + tq) * 4 + m2) * 64];
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].x, h1[4 * tq + 0], acc2[m2]);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].y, h1[4 * tq + 1], acc2[m2]);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].z, h1[4 * tq + 2], acc2[m2]);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].w, h1[4 * tq + 3], acc2[m2]);
        }
    }
    {
        const float4 *w2x = reinterpret_cast<const float4 *>(w.w2x) + lane;
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
            float4 a[4];
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) a[m2] = w2x[(sq * 4 + m2) * 64];
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].x, xr[4 * sq + 0], acc2[m2]);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].y, xr[4 * sq + 1], acc2[m2]);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].z, xr[4 * sq + 2], acc2[m2]);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = MFMA(a[m2].w, xr[4 * sq + 3], acc2[m2]);
        }
    }

    // ---- layer 3: (128 + 16) -> 1 on the VALU; lane (j,h) holds half of point j's channels ----
    const float *w3 = w.w3 + h * 72;
    float part = 0.0f;
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const f32x16 h2 = leaky(acc2[m2]);
        const f32x16 wv = load16(w3 + m2 * 16);
#pragma unroll
        for (int t = 0; t < 16; ++t) part = fmaf(wv[t], h2[t], part);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) part = fmaf(w3[64 + s], xr[s], part);
    const float other = __shfl_xor(part, 32);
    const float y = apply_last_op((part + other) + w.b3, w.last_op);
    if (h == 0 && base + j < N) out[base + j] = (MASK ? maskf != 0.0f : true) ? y : 0.0f;     // select: a masked point is 0 whatever the network said
}

int mlp_launch(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, int precision, hipStream_t st)
{
    if (precision == ICON_PRECISION_F32) return mlp_launch_ex(mlp, d_x, N, d_out, true, st);
    if (precision == ICON_PRECISION_F16X3) return mlp_launch_f16x3(mlp, d_x, N, d_out, true, st);
    return fail(ICON_ERR_ARG, "mlp: unknown precision");
}

int mlp_launch_ex(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, bool mask, hipStream_t st)
{
    if (N <= 0) return ICON_OK;
    const float *b = mlp->d_blob;
    MlpDev w;
    w.w0 = b + mlp->off_w0; w.b0 = b + mlp->off_b0; w.w1 = b + mlp->off_w1; w.b1 = b + mlp->off_b1;
    w.w2 = b + mlp->off_w2; w.w2x = b + mlp->off_w2x; w.b2 = b + mlp->off_b2; w.w3 = b + mlp->off_w3;
    w.b3 = mlp->b3; w.c0 = mlp->c0; w.last_op = mlp->last_op;
    const int64_t nb = (N + kPtsPerBlock - 1) / kPtsPerBlock;
    ICON_ARG(nb < (1ll << 31), "mlp: N too large for one launch");
    if (mask) hipLaunchKernelGGL(k_mlp_f32<true>, dim3((unsigned)nb), dim3(kMlpBlock), 0, st, d_x, N, d_out, w);
    else      hipLaunchKernelGGL(k_mlp_f32<false>, dim3((unsigned)nb), dim3(kMlpBlock), 0, st, d_x, N, d_out, w);
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

}  // namespace icon

using namespace icon;

extern "C" int icon_mlp_create(int n_layers, const int *cin, const int *cout, const int *is_res,
                               const float *const *h_W, const float *const *h_b,
                               const float *const *h_bn_gamma, const float *const *h_bn_beta,
                               const float *const *h_bn_mean, const float *const *h_bn_var,
                               float bn_eps, void *stream, icon_mlp_t **out)
{
    ICON_ARG(out != nullptr, "icon_mlp_create: out is null");
    *out = nullptr;
    ICON_ARG(cin && cout && is_res && h_W && h_b, "icon_mlp_create: null argument");
    if (n_layers != 4 || cout[0] != 512 || cout[1] != 256 || cout[2] != 128 || cout[3] != 1 ||
        is_res[0] || is_res[1] || !is_res[2] || !is_res[3])
        return fail(ICON_ERR_UNSUPPORTED, "icon_mlp_create: built for c0->512->256->(256+c0)->128->(128+c0)->1 "
                                          "(mlp_dim [*,512,256,128,1], res_layers [2,3,4])");
    const int c0 = cin[0];
    if (c0 < 1 || c0 > kCodeSlot) return fail(ICON_ERR_UNSUPPORTED, "icon_mlp_create: input width must be 1..15");
    ICON_ARG(cin[1] == 512 && cin[2] == 256 + c0 && cin[3] == 128 + c0, "icon_mlp_create: inconsistent Cin");
    const bool has_bn = h_bn_gamma && h_bn_beta && h_bn_mean && h_bn_var;

    // fold BatchNorm1d(eval): W' = W * g/sqrt(var+eps), b' = (b - mean) * g/sqrt(var+eps) + beta
    std::vector<std::vector<float>> W(4), B(4);
    for (int l = 0; l < 4; ++l) {
        const int co = cout[l], ci = cin[l];
        W[l].resize((size_t)co * ci); B[l].resize(co);
        for (int o = 0; o < co; ++o) {
            double scale = 1.0, shift = 0.0;
            if (l < 3 && has_bn) {
                scale = (double)h_bn_gamma[l][o] / std::sqrt((double)h_bn_var[l][o] + (double)bn_eps);
                shift = (double)h_bn_beta[l][o] - (double)h_bn_mean[l][o] * scale;
            }
            for (int k = 0; k < ci; ++k) W[l][(size_t)o * ci + k] = (float)((double)h_W[l][(size_t)o * ci + k] * scale);
            B[l][o] = (float)((double)h_b[l][o] * scale + shift);
        }
    }

    icon_mlp *m = new icon_mlp();
    m->c0 = c0;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 3) & ~(size_t)3; return o; };
    m->off_w0 = take(kW0); m->off_b0 = take(kB0); m->off_w1 = take(kW1); m->off_b1 = take(kB1);
    m->off_w2 = take(kW2); m->off_w2x = take(kW2x); m->off_b2 = take(kB2); m->off_w3 = take(kW3);
    m->off_plain = take(kPlainFloats); m->off_flag = take(4);
    std::vector<float> blob(off, 0.0f);
    auto rho = [](int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; };

    // layer 0: A operand of k-step s for M-tile c, lane (i,h): W0[32c+i][slot s+8h]
    for (int c = 0; c < 16; ++c)
        for (int s = 0; s < 8; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5, slot = s + 8 * h;
                const float v = (slot < c0) ? W[0][(size_t)(32 * c + i) * c0 + slot] : 0.0f;
                blob[m->off_w0 + (((size_t)(c * 2 + (s >> 2)) * 64 + lane) * 4) + (s & 3)] = v;
            }
    for (int c = 0; c < 16; ++c)
        for (int h = 0; h < 2; ++h)
            for (int t = 0; t < 16; ++t) blob[m->off_b0 + (size_t)(c * 2 + h) * 16 + t] = B[0][32 * c + rho(t, h)];
    // layer 1: k-step (c,t) for M-tile m, lane (i,h): W1[32m+i][32c + rho(t,h)]
    for (int c = 0; c < 16; ++c)
        for (int t = 0; t < 16; ++t)
            for (int mm = 0; mm < 8; ++mm)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    blob[m->off_w1 + ((((size_t)(c * 4 + (t >> 2)) * 8 + mm) * 64 + lane) * 4) + (t & 3)] =
                        W[1][(size_t)(32 * mm + i) * 512 + 32 * c + rho(t, h)];
                }
    for (int mm = 0; mm < 8; ++mm)
        for (int h = 0; h < 2; ++h)
            for (int t = 0; t < 16; ++t) blob[m->off_b1 + (size_t)(mm * 2 + h) * 16 + t] = B[1][32 * mm + rho(t, h)];
    // layer 2: hidden part, k-step (m,t) for M-tile m2: W2[32m2+i][32m + rho(t,h)]; input part: W2[..][256 + slot]
    const int ci2 = 256 + c0;
    for (int mm = 0; mm < 8; ++mm)
        for (int t = 0; t < 16; ++t)
            for (int m2 = 0; m2 < 4; ++m2)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    blob[m->off_w2 + ((((size_t)(mm * 4 + (t >> 2)) * 4 + m2) * 64 + lane) * 4) + (t & 3)] =
                        W[2][(size_t)(32 * m2 + i) * ci2 + 32 * mm + rho(t, h)];
                }
    for (int s = 0; s < 8; ++s)
        for (int m2 = 0; m2 < 4; ++m2)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5, slot = s + 8 * h;
                const float v = (slot < c0) ? W[2][(size_t)(32 * m2 + i) * ci2 + 256 + slot] : 0.0f;
                blob[m->off_w2x + ((((size_t)(s >> 2) * 4 + m2) * 64 + lane) * 4) + (s & 3)] = v;
            }
    for (int m2 = 0; m2 < 4; ++m2)
        for (int h = 0; h < 2; ++h)
            for (int t = 0; t < 16; ++t) blob[m->off_b2 + (size_t)(m2 * 2 + h) * 16 + t] = B[2][32 * m2 + rho(t, h)];
    // layer 3
    for (int h = 0; h < 2; ++h) {
        for (int m2 = 0; m2 < 4; ++m2)
            for (int t = 0; t < 16; ++t) blob[m->off_w3 + (size_t)h * 72 + m2 * 16 + t] = W[3][32 * m2 + rho(t, h)];
        for (int s = 0; s < 8; ++s) {
            const int slot = s + 8 * h;
            blob[m->off_w3 + (size_t)h * 72 + 64 + s] = (slot < c0) ? W[3][128 + slot] : 0.0f;
        }
    }
    m->b3 = B[3][0];
    {   // plain [Cout][Cin] copies for the f32 safety net (mlp_plain_device.h): w0 [512][16] | b0 | w1 [256][512] | b1 | w2 [128][272] | b2 | w3 [144]
        float *q = blob.data() + m->off_plain;
        for (int o = 0; o < 512; ++o)
            for (int k = 0; k < c0; ++k) q[o * 16 + k] = W[0][(size_t)o * c0 + k];
        q += 512 * 16;
        for (int o = 0; o < 512; ++o) q[o] = B[0][o];
        q += 512;
        for (size_t i = 0; i < (size_t)256 * 512; ++i) q[i] = W[1][i];
        q += 256 * 512;
        for (int o = 0; o < 256; ++o) q[o] = B[1][o];
        q += 256;
        for (int o = 0; o < 128; ++o) {
            for (int k = 0; k < 256; ++k) q[o * 272 + k] = W[2][(size_t)o * ci2 + k];
            for (int k = 0; k < c0; ++k) q[o * 272 + 256 + k] = W[2][(size_t)o * ci2 + 256 + k];
        }
        q += 128 * 272;
        for (int o = 0; o < 128; ++o) q[o] = B[2][o];
        q += 128;
        for (int k = 0; k < 128; ++k) q[k] = W[3][k];
        for (int k = 0; k < c0; ++k) q[128 + k] = W[3][128 + k];
    }
    m->blob_bytes = blob.size() * sizeof(float);
    hipError_t e = hipMalloc((void **)&m->d_blob, m->blob_bytes);
    if (e != hipSuccess) { delete m; return fail(ICON_ERR_HIP, std::string("hipMalloc mlp: ") + hipGetErrorString(e)); }
    hipStream_t st = (hipStream_t)stream;
    e = hipMemcpyAsync(m->d_blob, blob.data(), m->blob_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { icon_mlp_destroy(m); return fail(ICON_ERR_HIP, std::string("upload mlp: ") + hipGetErrorString(e)); }
    int rc = mlp_pack_f16x3(m, W, B, st);
    if (rc) { icon_mlp_destroy(m); return rc; }
    *out = m;
    return ICON_OK;
}

// ---- the f32 safety net of the split-precision kernels over materialised rows (mlp_plain_device.h) ----------------
namespace icon {

MlpPlain mlp_plain_of(const icon_mlp *m)
{
    const float *q = m->d_blob + m->off_plain;
    MlpPlain P;
    P.w0 = q; q += 512 * 16; P.b0 = q; q += 512; P.w1 = q; q += 256 * 512; P.b1 = q; q += 256;
    P.w2 = q; q += 128 * 272; P.b2 = q; q += 128; P.w3 = q;
    P.b3 = m->b3; P.last_op = m->last_op; P.c0 = m->c0;
    return P;
}

__global__ __launch_bounds__(64) void k_rescue_rows(const float *__restrict__ X, int64_t N, float *__restrict__ out, MlpPlain P, const int *flag, int always)
{
    __shared__ float s[kPlainLds];
    if (!always && *flag == 0) return;                        // the usual case: one word read per workgroup of a small grid
    const int lane = threadIdx.x;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < N; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + lane;
        const float v = i < N ? out[i] : 0.0f;
        unsigned long long todo = __ballot(not_finite(v));
        while (todo) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t ib = base + b;
            if (lane < 16) s[lane] = lane < P.c0 ? X[ib * kXRow + lane] : 0.0f;
            const float y = mlp_plain_wave(P, s, lane);
            if (lane == 0) out[ib] = y;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ICON_AMD_RESCUE_ALWAYS=1 (diagnostics): scan the results even when no kernel raised the flag
int rescue_always()
{
    static const int v = (getenv("ICON_AMD_RESCUE_ALWAYS") && atoi(getenv("ICON_AMD_RESCUE_ALWAYS")) != 0) ? 1 : 0;
    return v;
}

int mlp_flag_reset(const icon_mlp *mlp, hipStream_t st)
{
    ICON_HIP(hipMemsetAsync(mlp->d_blob + mlp->off_flag, 0, sizeof(int), st));
    return ICON_OK;
}

int mlp_rescue_rows(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, hipStream_t st)
{
    const int64_t nb = (N + 63) / 64;
    hipLaunchKernelGGL(k_rescue_rows, dim3((unsigned)std::min<int64_t>(nb, 2048)), dim3(64), 0, st, d_x, N, d_out, mlp_plain_of(mlp),
                       reinterpret_cast<const int *>(mlp->d_blob + mlp->off_flag), rescue_always());
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

}  // namespace icon

extern "C" int icon_mlp_destroy(icon_mlp_t *m)
{
    if (!m) return ICON_OK;
    (void)hipFree(m->d_blob); (void)hipFree(m->d_f16);
    delete m;
    return ICON_OK;
}

extern "C" int icon_mlp_set_last_op(icon_mlp_t *m, int last_op)
{
    ICON_ARG(m != nullptr, "icon_mlp_set_last_op: mlp is null");
    ICON_ARG(last_op == ICON_LASTOP_NONE || last_op == ICON_LASTOP_SIGMOID, "icon_mlp_set_last_op: unknown last_op");
    m->last_op = last_op;
    return ICON_OK;
}

extern "C" int icon_mlp_forward(const icon_mlp_t *mlp, const float *d_x, int64_t N, float *d_out,
                                int precision, void *stream)
{
    ICON_ARG(mlp && d_x && d_out, "icon_mlp_forward: null argument");
    ICON_ARG(N >= 0, "icon_mlp_forward: negative N");
    if (precision == ICON_PRECISION_F32) return mlp_launch_ex(mlp, d_x, N, d_out, false, (hipStream_t)stream);
    if (precision == ICON_PRECISION_F16X3) return mlp_launch_f16x3(mlp, d_x, N, d_out, false, (hipStream_t)stream);
    return fail(ICON_ERR_ARG, "icon_mlp_forward: unknown precision");
}
