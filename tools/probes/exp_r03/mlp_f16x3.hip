// mlp_f16x3.hip - the occupancy MLP on the f16 matrix cores with float32-class accuracy.
//
// Same contract and the same register-chained layer structure as mlp_kernels.hip (see there for
// the reference citations: lib/net/MLP.py:49-72, lib/net/HGPIFuNet.py:128-133,363), but every
// float32 product a*b is evaluated as three f16 MFMA products with f32 accumulation,
//       a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,   x_hi = f16(x), x_lo = f16(x - x_hi),
// so both operands carry 22 significant bits and the dropped a_lo*b_lo term is 2^-22 relative:
// the result differs from the exact-f32 path by ~1e-6 of the activations' magnitude (measured in
// tests/test_gpu_parity.py against the float64 oracle; the north-star tolerance is 1e-4).
// v_mfma_f32_32x32x16_f16 retires K=16 in 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64, so
// three of them are 5.3x faster than the exact-f32 MFMA chain.
//
// What changes structurally at that rate:
//  * weights can no longer stream from L2 per wavefront (85 B/clk/CU): a 512-thread workgroup
//    (8 waves x 32 points) shares them through LDS.  The packed hi/lo operand image (680 KB) is cut
//    into 20 chunks of 32-40 KB that are DMA'd global->LDS with global_load_lds_dwordx4 into a
//    double buffer; chunk k+1 lands while chunk k is being multiplied, one barrier per chunk.
//  * A operands are read with ds_read_b128 in [..][lane][16 B] order (conflict-free);
//    B operands are the previous layer's accumulators, LeakyReLU'd and split to hi/lo in
//    registers (v_cvt_pkrtz_f16_f32) - still no activation ever touches LDS or HBM.
//  * per-layer power-of-two weight scales keep the lo parts out of the f16 subnormal range; they are
//    divided out (exactly) in the activation step.
#include "mlp_f16x3_device.h"

#include <cmath>
#include <cstring>

namespace icon {

template <bool MASK>
__global__ __launch_bounds__(kF16Block, 2) void k_mlp_f16x3(const float *__restrict__ X, int64_t N, float *__restrict__ out, MlpF16Dev w)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
#if defined(ICON_EXP_PRIO)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);      // the second-dispatched half loses the issue arbitration otherwise
#endif
    const int64_t base = ((int64_t)blockIdx.x * (kF16Block / 64) + wave) * 32;
    const int64_t pi = min(base + j, N - 1);      // waves past the end still help with the DMA + barriers

    // side arrays -> LDS once: no ordinary global load may sit between an LDS-DMA and its consumer
    // (vmcnt retires in order, so waiting for such a load would drain the DMA queue as well)
    issue_units(w.image, smem + kW0Off, kW0Bytes / 1024, wave, lane);     // resident layer-0 operands
    float *side = reinterpret_cast<float *>(smem + kSideOff);
    for (int i = threadIdx.x; i < kSideFloats; i += kF16Block) side[i] = w.side[i];
    const float *sb0 = side, *sb1 = side + 512, *sb2 = side + 768, *sw3 = side + 896;

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
    for (int s = 0; s < 8; ++s) xr[s] = (s + 8 * h < w.c0) ? xr[s] : 0.0f;
    half8 xhi, xlo;
    split8(xr, xhi, xlo);
    issue_chunk(w.image, smem, 0, wave, lane);
    __syncthreads();   // side arrays visible, chunk 0 landed (the barrier's release waits for the LDS-DMA)

    f32x16 acc1[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc1[m] = ld16(sb1 + (m * 2 + h) * 16);

    // ---- layers 0 + 1: one chunk = 32 hidden channels ------------------------------------------
    half8 bh[2], bl[2];
    activate_split(l0_tile(smem + kW0Off, sb0, 0, xhi, xlo, h, lane), LeakyK{w.inv0, w.p0, w.q0}, bh, bl);
    for (int c = 0; c < 16; ++c) {
        l01_chunk(smem + (c & 1) * kBufBytes, smem + ((c + 1) & 1) * kBufBytes, smem + kW0Off, sb0, w.image, c, acc1, xhi, xlo,
                  LeakyK{w.inv0, w.p0, w.q0}, h, lane, wave, bh, bl);
        ICON_CHUNK_BARRIER();   // all waves done with this buffer AND the next chunk has landed
    }

    // ---- layer 2: K = 256 (registers) + 16 (raw input) ---------------------------------------------
    f32x16 acc2[4];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = ld16(sb2 + (m2 * 2 + h) * 16);
    activate_split(acc1[0], LeakyK{w.inv1, w.p1, w.q1}, bh, bl);
    l2_chunk<0>(smem, smem + kBufBytes, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl);
    ICON_CHUNK_BARRIER();
    l2_chunk<1>(smem + kBufBytes, smem, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl);
    ICON_CHUNK_BARRIER();
    l2_chunk<2>(smem, smem + kBufBytes, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl);
    ICON_CHUNK_BARRIER();
    l2_chunk<3>(smem + kBufBytes, smem, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl);

    // ---- layer 3 on the VALU (f32) ----------------------------------------------------------------
    const float *w3 = sw3 + h * 72;
    float part = 0.0f;
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const f32x16 wv = ld16(w3 + m2 * 16);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            part = fmaf(wv[t], leaky_scaled(acc2[m2][t], LeakyK{w.inv2, w.p2, w.q2}), part);
        }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) part = fmaf(w3[64 + s], xr[s], part);
    const float other = __shfl_xor(part, 32);
    const float y = apply_last_op((part + other) + w.b3, w.last_op);
    if (h == 0 && base + j < N) out[base + j] = masked_result(y, MASK ? maskf != 0.0f : true, w.flag);
}

// ---------------------------------------------------------------------------------------------
// host side: operand image
// ---------------------------------------------------------------------------------------------
uint16_t f32_to_f16_rtn(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                  // rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                               // rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    if (e < -14) { shift = 13 + (-14 - e); e = -15; } else shift = 13;
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) ++r;
    uint32_t out = (e == -15) ? r : (((uint32_t)(e + 15) << 10) + (r - 0x400u));  // carry propagates into exponent
    return (uint16_t)(sign | out);
}

float f16_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 0x400u), e - 25);
    return sign ? -v : v;
}

float pick_scale(const std::vector<float> &W)
{
    float mx = 0.f;
    for (float v : W) mx = std::max(mx, std::fabs(v));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    int e = (int)std::floor(std::log2(8192.0 / (double)mx));
    e = std::min(std::max(e, -12), 24);
    return std::ldexp(1.0f, e);
}

// W: folded float32 weights of the four layers (row-major [cout][cin]), B: folded biases
int mlp_pack_f16x3(icon_mlp *m, const std::vector<std::vector<float>> &W, const std::vector<std::vector<float>> &B,
                   hipStream_t st)
{
    const int c0 = m->c0;
    const float s0 = pick_scale(W[0]), s1 = pick_scale(W[1]), s2 = pick_scale(W[2]);
    std::vector<uint16_t> img(kImageBytes / 2, 0);
    auto rho = [](int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; };
    // one 1-KiB slot = [lane 64][8 halves]; write hi into `slot`, lo into `slot + 1`
    auto put = [&](int chunk, int slot, int lane, int e, float wv, float scale) {
        const float ws = wv * scale;
        const uint16_t hi = f32_to_f16_rtn(ws);
        const uint16_t lo = f32_to_f16_rtn(ws - f16_to_f32(hi));
        const size_t base = ((size_t)(chunk < 0 ? 0 : chunk_offset(chunk)) + slot) * 512;   // in halves; chunk -1 = W0 region
        img[base + (size_t)lane * 8 + e] = hi;
        img[base + 512 + (size_t)lane * 8 + e] = lo;
    };
    const int ci2 = 256 + c0;
    parallel_for(64, [&](int lane) {       // every lane owns its own 16-byte column of each slot
        const int i = lane & 31, g = lane >> 5;
        for (int e = 0; e < 8; ++e) {
            // layers 0+1, chunk c
            for (int c = 0; c < 16; ++c) {
                const int slot0 = 8 * g + e;
                put(-1, 2 * c, lane, e, slot0 < c0 ? W[0][(size_t)(32 * c + i) * c0 + slot0] : 0.f, s0);
                for (int u = 0; u < 2; ++u)
                    for (int mm = 0; mm < 8; ++mm)
                        put(c, (u * 8 + mm) * 2, lane, e, W[1][(size_t)(32 * mm + i) * 512 + 32 * c + rho(8 * u + e, g)], s1);
            }
            // layer 2, chunk 16+q covers hidden tiles 2q, 2q+1
            for (int q = 0; q < 4; ++q)
                for (int mm = 0; mm < 2; ++mm)
                    for (int u = 0; u < 2; ++u)
                        for (int m2 = 0; m2 < 4; ++m2)
                            put(16 + q, ((mm * 2 + u) * 4 + m2) * 2, lane, e,
                                W[2][(size_t)(32 * m2 + i) * ci2 + 32 * (2 * q + mm) + rho(8 * u + e, g)], s2);
            for (int m2 = 0; m2 < 4; ++m2) {
                const int slot0 = 8 * g + e;
                put(19, 32 + m2 * 2, lane, e, slot0 < c0 ? W[2][(size_t)(32 * m2 + i) * ci2 + 256 + slot0] : 0.f, s2);
            }
        }
    });
    // f32 side arrays: scaled biases (accumulator initial values) and the last layer
    std::vector<float> side(16 * 2 * 16 + 8 * 2 * 16 + 4 * 2 * 16 + 2 * 72, 0.f);
    float *b0 = side.data(), *b1 = b0 + 512, *b2 = b1 + 256, *w3 = b2 + 128;
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 16; ++t) {
            for (int c = 0; c < 16; ++c) b0[(c * 2 + h) * 16 + t] = B[0][32 * c + rho(t, h)] * s0;
            for (int mm = 0; mm < 8; ++mm) b1[(mm * 2 + h) * 16 + t] = B[1][32 * mm + rho(t, h)] * s1;
            for (int m2 = 0; m2 < 4; ++m2) b2[(m2 * 2 + h) * 16 + t] = B[2][32 * m2 + rho(t, h)] * s2;
        }
    for (int h = 0; h < 2; ++h) {
        for (int m2 = 0; m2 < 4; ++m2)
            for (int t = 0; t < 16; ++t) w3[h * 72 + m2 * 16 + t] = W[3][32 * m2 + rho(t, h)];
        for (int s = 0; s < 8; ++s) w3[h * 72 + 64 + s] = (s + 8 * h < c0) ? W[3][128 + s + 8 * h] : 0.f;
    }
    const size_t side_bytes = side.size() * sizeof(float);
    ICON_HIP(hipMalloc((void **)&m->d_f16, kImageBytes + side_bytes));
    ICON_HIP(hipMemcpyAsync(m->d_f16, img.data(), kImageBytes, hipMemcpyHostToDevice, st));
    ICON_HIP(hipMemcpyAsync(m->d_f16 + kImageBytes, side.data(), side_bytes, hipMemcpyHostToDevice, st));
    ICON_HIP(hipStreamSynchronize(st));
    m->f16_inv[0] = 1.0f / s0; m->f16_inv[1] = 1.0f / s1; m->f16_inv[2] = 1.0f / s2;
    return ICON_OK;
}

int mlp_launch_f16x3(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, bool mask, hipStream_t st)
{
    if (N <= 0) return ICON_OK;
    MlpF16Dev w;
    w.image = mlp->d_f16;
    w.side = reinterpret_cast<const float *>(mlp->d_f16 + kImageBytes);
    w.b3 = mlp->b3; w.inv0 = mlp->f16_inv[0]; w.inv1 = mlp->f16_inv[1]; w.inv2 = mlp->f16_inv[2]; w.c0 = mlp->c0; w.last_op = mlp->last_op;
    w.flag = reinterpret_cast<int *>(mlp->d_blob + mlp->off_flag);
    w.p0 = 0.505f * w.inv0; w.q0 = 0.495f * w.inv0; w.p1 = 0.505f * w.inv1; w.q1 = 0.495f * w.inv1; w.p2 = 0.505f * w.inv2; w.q2 = 0.495f * w.inv2;
    const int64_t nb = (N + kF16Pts - 1) / kF16Pts;
    ICON_ARG(nb < (1ll << 31), "mlp: N too large for one launch");
    int rc = once_per_device(6, [] {       // per device: a process may drive several (common.h)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_f16x3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_f16x3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    });
    if (rc) return rc;
    rc = mlp_flag_reset(mlp, st);
    if (rc) return rc;
    if (mask) hipLaunchKernelGGL(k_mlp_f16x3<true>, dim3((unsigned)nb), dim3(kF16Block), kLdsBytes, st, d_x, N, d_out, w);
    else      hipLaunchKernelGGL(k_mlp_f16x3<false>, dim3((unsigned)nb), dim3(kF16Block), kLdsBytes, st, d_x, N, d_out, w);
    ICON_HIP(hipGetLastError());
    return mlp_rescue_rows(mlp, d_x, N, d_out, st);
}

}  // namespace icon
