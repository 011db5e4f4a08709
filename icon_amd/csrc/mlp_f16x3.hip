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
#include "common.h"

#include <cmath>
#include <cstring>

namespace icon {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int kF16Block = 512;                 // 8 waves x 32 points
constexpr int kF16Pts = (kF16Block / 64) * 32; // 256 points per workgroup
constexpr int kBufBytes = 40 * 1024;
constexpr int kSideFloats = 512 + 256 + 128 + 144;   // b0 | b1 | b2 | w3, staged once per workgroup
constexpr int kW0Off = 2 * kBufBytes;                // layer-0 operands, resident for the whole workgroup
constexpr int kW0Bytes = 32 * 1024;
constexpr int kSideOff = kW0Off + kW0Bytes;
constexpr int kLdsBytes = kSideOff + 4352;           // 80 KiB double buffer + 32 KiB W0 + 4.25 KiB side arrays

// packed image: [W0: 32 KiB][layer-1 chunks 0..15: 32 KiB each][layer-2 chunks 16..18: 32 KiB, 19: 40 KiB]
// chunk k: size in KiB and offset in KiB inside the image
__host__ __device__ constexpr int chunk_units(int k) { return k < 19 ? 32 : 40; }
__host__ __device__ constexpr int chunk_offset(int k) { return 32 + 32 * k; }
constexpr size_t kImageBytes = (size_t)(32 + 19 * 32 + 40) * 1024;   // 680 KiB

struct MlpF16Dev {
    const char *image;          // packed f16 hi/lo A operands, chunked
    const float *side;          // f32: b0 [16][2][16] | b1 [8][2][16] | b2 [4][2][16] (bias * weight scale) | w3 [2][72]
    float b3;
    float inv0, inv1, inv2;     // 1 / weight scale of layers 0..2
    int c0;
};

__device__ __forceinline__ f32x16 ld16(const float *p)
{
    const float4 *q = reinterpret_cast<const float4 *>(p);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    f32x16 v;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
    return v;
}

// x -> (hi, lo) with hi = f16_rtz(x), lo = f16_rtz(x - hi); 8 values -> one MFMA operand each
__device__ __forceinline__ void split8(const float *v, half8 &hi, half8 &lo)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]);
        const float r0 = v[2 * q] - (float)h[0], r1 = v[2 * q + 1] - (float)h[1];
        const fp16x2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
        hi[2 * q] = (_Float16)h[0]; hi[2 * q + 1] = (_Float16)h[1];
        lo[2 * q] = (_Float16)l[0]; lo[2 * q + 1] = (_Float16)l[1];
    }
}

// activation step of a finished tile: undo the weight scale, LeakyReLU(0.01), split for the next GEMM
__device__ __forceinline__ void activate_split(const f32x16 &acc, float inv, half8 hi[2], half8 lo[2])
{
    float v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { const float x = acc[t] * inv; v[t] = fmaxf(x, 0.01f * x); }
    split8(v, hi[0], lo[0]);
    split8(v + 8, hi[1], lo[1]);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ half8 lds_op(const char *buf, int slot, int lane)
{
    return *reinterpret_cast<const half8 *>(buf + slot * 1024 + lane * 16);
}

// every wave DMAs 1 KiB pieces round-robin: global [piece][lane][16 B] -> LDS, same order
__device__ __forceinline__ void issue_units(const char *src, char *buf, int units, int wave, int lane)
{
    for (int u = wave; u < units; u += kF16Block / 64)
        __builtin_amdgcn_global_load_lds((gvoid_t *)(src + u * 1024 + lane * 16), (lvoid_t *)(buf + u * 1024), 16, 0, 0);
}
__device__ __forceinline__ void issue_chunk(const char *image, char *buf, int k, int wave, int lane)
{
    issue_units(image + (size_t)chunk_offset(k) * 1024, buf, chunk_units(k), wave, lane);
}

// 3-term product group for 4 output tiles sharing one B operand pair
#define TRIPLE4(ACC, M0, AH, AL, BH, BL)                                     \
    _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4) ACC[M0 + i4] = MFMA16(AH[i4], BH, ACC[M0 + i4]); \
    _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4) ACC[M0 + i4] = MFMA16(AH[i4], BL, ACC[M0 + i4]); \
    _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4) ACC[M0 + i4] = MFMA16(AL[i4], BH, ACC[M0 + i4]);

// The per-chunk bodies take the buffer being READ, the buffer being FILLED and the side arrays as
// __restrict__ parameters of a force-inlined function: after inlining, the ds_reads carry alias
// scopes that prove they cannot touch the LDS-DMA destination, so the compiler's waitcnt pass does
// not drain the DMA queue (s_waitcnt vmcnt(0)) in front of every LDS read - the DMA for chunk k+1
// stays in flight for the whole multiplication of chunk k and is only waited for at the barrier.

// 3-term product group for 2 output tiles sharing one B operand pair
#define TRIPLE2(ACC, M0, AH, AL, BH, BL)                                     \
    ACC[M0] = MFMA16(AH[0], BH, ACC[M0]); ACC[M0 + 1] = MFMA16(AH[1], BH, ACC[M0 + 1]); \
    ACC[M0] = MFMA16(AH[0], BL, ACC[M0]); ACC[M0 + 1] = MFMA16(AH[1], BL, ACC[M0 + 1]); \
    ACC[M0] = MFMA16(AL[0], BH, ACC[M0]); ACC[M0 + 1] = MFMA16(AL[1], BH, ACC[M0 + 1]);

// layer 0, hidden tile c (32 channels): 3 MFMAs from the resident W0 region
__device__ __forceinline__ f32x16 l0_tile(const char *__restrict__ W0, const float *__restrict__ sb0, int c, half8 xhi, half8 xlo,
                                          int h, int lane)
{
    f32x16 h0 = ld16(sb0 + (c * 2 + h) * 16);
    const half8 a_hi = lds_op(W0, 2 * c, lane), a_lo = lds_op(W0, 2 * c + 1, lane);
    h0 = MFMA16(a_hi, xhi, h0); h0 = MFMA16(a_hi, xlo, h0); h0 = MFMA16(a_lo, xhi, h0);
    return h0;
}

// one eighth of the activation step of a finished layer-0 tile: values 2k, 2k+1 -> LeakyReLU ->
// hi/lo halves -> pair (k&3) of the next B operand (u = k>>2)
__device__ __forceinline__ void act_part(const f32x16 &acc, int k, float inv, half8 (&nh)[2], half8 (&nl)[2])
{
    const float x0 = acc[2 * k] * inv, x1 = acc[2 * k + 1] * inv;
    const float v0 = fmaxf(x0, 0.01f * x0), v1 = fmaxf(x1, 0.01f * x1);
    fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    fp16x2 ll = __builtin_amdgcn_cvt_pkrtz(v0 - (float)hh[0], v1 - (float)hh[1]);
    // the (empty) volatile asm is ordered against the surrounding sched_barriers, which keeps this
    // VALU work in the MFMA group it was written next to instead of being sunk to the end of the chunk
    int hb = __builtin_bit_cast(int, hh), lb = __builtin_bit_cast(int, ll);
    asm volatile("" : "+v"(hb), "+v"(lb));
    hh = __builtin_bit_cast(fp16x2, hb); ll = __builtin_bit_cast(fp16x2, lb);
    const int u = k >> 2, q = k & 3;
    nh[u][2 * q] = (_Float16)hh[0]; nh[u][2 * q + 1] = (_Float16)hh[1];
    nl[u][2 * q] = (_Float16)ll[0]; nl[u][2 * q + 1] = (_Float16)ll[1];
}

// A operands of MFMA group g of a layer-1 chunk: output tiles 2*(g&3), +1 for k-step u = g>>2
__device__ __forceinline__ void load_group(const char *__restrict__ L, int g, int lane, half8 (&a)[4])
{
    const int slot = ((g >> 2) * 8 + (g & 3) * 2) * 2;
    a[0] = lds_op(L, slot, lane); a[1] = lds_op(L, slot + 2, lane);        // hi of tile 0, 1
    a[2] = lds_op(L, slot + 1, lane); a[3] = lds_op(L, slot + 3, lane);    // lo of tile 0, 1
}

// layer 1, chunk c (K = hidden channels 32c..32c+31, B operands bh/bl prepared one iteration
// earlier) SOFTWARE-PIPELINED with layer 0 of chunk c+1: its three MFMAs are issued first; its
// LeakyReLU + hi/lo split (pure VALU, 8 parts) is slotted between the eight 6-MFMA groups of this
// chunk, and the LDS reads of group g+1 are issued ahead of the MFMAs of group g.  sched_barrier
// pins that interleave so the matrix pipe and the VALU run side by side instead of alternating.
__device__ __forceinline__ void l01_chunk(const char *__restrict__ L, char *__restrict__ nxt, const char *__restrict__ W0,
                                          const float *__restrict__ sb0, const char *image, int c, f32x16 (&acc1)[8],
                                          half8 xhi, half8 xlo, float inv0, int h, int lane, int wave,
                                          half8 (&bh)[2], half8 (&bl)[2])
{
    issue_chunk(image, nxt, c + 1, wave, lane);
    const f32x16 h0n = l0_tile(W0, sb0, min(c + 1, 15), xhi, xlo, h, lane);   // c == 15: harmless repeat
    half8 nh[2], nl[2];
    half8 a[2][4];
    load_group(L, 0, lane, a[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) load_group(L, g + 1, lane, a[(g + 1) & 1]);
        const int u = g >> 2, m0 = (g & 3) * 2;
        const half8 ah[2] = {a[g & 1][0], a[g & 1][1]}, al[2] = {a[g & 1][2], a[g & 1][3]};
        TRIPLE2(acc1, m0, ah, al, bh[u], bl[u])
        if (g >= 1) act_part(h0n, g - 1, inv0, nh, nl);
        __builtin_amdgcn_sched_barrier(0);
    }
    act_part(h0n, 7, inv0, nh, nl);
    bh[0] = nh[0]; bh[1] = nh[1]; bl[0] = nl[0]; bl[1] = nl[1];
}

// layer 2, chunk 16+Q: hidden tiles 2Q, 2Q+1 (+ the raw-input k-step in the last chunk).
// Same pipelining as layer 1: while the 24 MFMAs of hidden tile m run, the activation + split of
// tile m+1 (the next B operand) is slotted between the four 6-MFMA groups, two parts per group.
template <int Q>
__device__ __forceinline__ void l2_chunk(const char *__restrict__ L, char *__restrict__ nxt, const char *image,
                                         f32x16 (&acc1)[8], f32x16 (&acc2)[4], half8 xhi, half8 xlo, float inv1,
                                         int lane, int wave, half8 (&bh)[2], half8 (&bl)[2])
{
    if (Q < 3) issue_chunk(image, nxt, 17 + Q, wave, lane);
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
        constexpr int kLast = 7;
        const int m = 2 * Q + mm;
        half8 nh[2], nl[2];
        half8 a[2][4];
        auto load2 = [&](int g, half8 (&dst)[4]) {          // group g: k-step u = g>>1, output tiles 2*(g&1), +1
            const int slot = (((mm * 2 + (g >> 1)) * 4 + (g & 1) * 2) * 2);
            dst[0] = lds_op(L, slot, lane); dst[1] = lds_op(L, slot + 2, lane);
            dst[2] = lds_op(L, slot + 1, lane); dst[3] = lds_op(L, slot + 3, lane);
        };
        load2(0, a[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) load2(g + 1, a[(g + 1) & 1]);
            const int u = g >> 1, m0 = (g & 1) * 2;
            const half8 ah[2] = {a[g & 1][0], a[g & 1][1]}, al[2] = {a[g & 1][2], a[g & 1][3]};
            TRIPLE2(acc2, m0, ah, al, bh[u], bl[u])
            if (m < kLast) { act_part(acc1[m < kLast ? m + 1 : m], 2 * g, inv1, nh, nl); act_part(acc1[m < kLast ? m + 1 : m], 2 * g + 1, inv1, nh, nl); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (m < kLast) { bh[0] = nh[0]; bh[1] = nh[1]; bl[0] = nl[0]; bl[1] = nl[1]; }
    }
    if (Q == 3) {
        half8 ah[4], al[4];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) { ah[i4] = lds_op(L, 32 + i4 * 2, lane); al[i4] = lds_op(L, 33 + i4 * 2, lane); }
        TRIPLE4(acc2, 0, ah, al, xhi, xlo)
    }
}

template <bool MASK>
__global__ __launch_bounds__(kF16Block, 2) void k_mlp_f16x3(const float *__restrict__ X, int64_t N, float *__restrict__ out, MlpF16Dev w)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
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
    activate_split(l0_tile(smem + kW0Off, sb0, 0, xhi, xlo, h, lane), w.inv0, bh, bl);
    for (int c = 0; c < 16; ++c) {
        l01_chunk(smem + (c & 1) * kBufBytes, smem + ((c + 1) & 1) * kBufBytes, smem + kW0Off, sb0, w.image, c, acc1, xhi, xlo,
                  w.inv0, h, lane, wave, bh, bl);
        __syncthreads();   // all waves done with this buffer AND the next chunk has landed
    }

    // ---- layer 2: K = 256 (registers) + 16 (raw input) ---------------------------------------------
    f32x16 acc2[4];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = ld16(sb2 + (m2 * 2 + h) * 16);
    activate_split(acc1[0], w.inv1, bh, bl);
    l2_chunk<0>(smem, smem + kBufBytes, w.image, acc1, acc2, xhi, xlo, w.inv1, lane, wave, bh, bl);
    __syncthreads();
    l2_chunk<1>(smem + kBufBytes, smem, w.image, acc1, acc2, xhi, xlo, w.inv1, lane, wave, bh, bl);
    __syncthreads();
    l2_chunk<2>(smem, smem + kBufBytes, w.image, acc1, acc2, xhi, xlo, w.inv1, lane, wave, bh, bl);
    __syncthreads();
    l2_chunk<3>(smem + kBufBytes, smem, w.image, acc1, acc2, xhi, xlo, w.inv1, lane, wave, bh, bl);

    // ---- layer 3 on the VALU (f32) ----------------------------------------------------------------
    const float *w3 = sw3 + h * 72;
    float part = 0.0f;
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const f32x16 wv = ld16(w3 + m2 * 16);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float x = acc2[m2][t] * w.inv2;
            part = fmaf(wv[t], fmaxf(x, 0.01f * x), part);
        }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) part = fmaf(w3[64 + s], xr[s], part);
    const float other = __shfl_xor(part, 32);
    const float y = (part + other) + w.b3;
    if (h == 0 && base + j < N) out[base + j] = MASK ? maskf * y : y;
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
    w.b3 = mlp->b3; w.inv0 = mlp->f16_inv[0]; w.inv1 = mlp->f16_inv[1]; w.inv2 = mlp->f16_inv[2]; w.c0 = mlp->c0;
    const int64_t nb = (N + kF16Pts - 1) / kF16Pts;
    ICON_ARG(nb < (1ll << 31), "mlp: N too large for one launch");
    static bool attr_set = false;
    if (!attr_set) {
        ICON_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_f16x3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        ICON_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_f16x3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        attr_set = true;
    }
    if (mask) hipLaunchKernelGGL(k_mlp_f16x3<true>, dim3((unsigned)nb), dim3(kF16Block), kLdsBytes, st, d_x, N, d_out, w);
    else      hipLaunchKernelGGL(k_mlp_f16x3<false>, dim3((unsigned)nb), dim3(kF16Block), kLdsBytes, st, d_x, N, d_out, w);
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

}  // namespace icon
