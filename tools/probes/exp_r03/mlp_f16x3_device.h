// mlp_f16x3_device.h - device side of the 3x f16 split-precision MLP (see mlp_f16x3.hip for the design notes):
// operand image layout, LDS-DMA helpers and the per-chunk MFMA bodies, shared by k_mlp_f16x3 (MLP on
// materialised input rows) and k_fused_f16x3 (fused_f16x3.hip: features computed in the same kernel).
#pragma once
#include "common.h"

namespace icon {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

#if defined(ICON_EXP_HALFWG)
// experiment (with NO_DMA + SLOTWRAP): workgroups of 4 waves / 128 points and 52 KiB of LDS, TWO per CU - do the serial
// phases of one tile (prologue, barriers, layer 3) hide behind the MFMA phases of the other workgroup's tile?
constexpr int kF16Block = 256;
#else
constexpr int kF16Block = 512;                 // 8 waves x 32 points
#endif
constexpr int kF16Pts = (kF16Block / 64) * 32; // 256 points per workgroup
#if defined(ICON_EXP_SLOTWRAP)
constexpr int kBufBytes = 8 * 1024;            // experiment: every A-operand read wraps into the first 8 slots
#else
constexpr int kBufBytes = 40 * 1024;
#endif
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
    // LeakyReLU constants 0.505 / scale and 0.495 / scale per layer, from the host: wave-uniform kernel arguments stay in SGPRs
    // (computed in the kernel they took five vector registers of a body that has none to spare: 20 B/lane of scratch)
    float p0, q0, p1, q1, p2, q2;
    int c0;
    int last_op;                // ICON_LASTOP_*
    int *flag;                  // raised when an in-cube result is not finite (operand beyond the f16 range): k_rescue_* redo the point in f32
};

// the in_cube mask as a SELECT (a masked point is 0 whatever the network said - 0 * NaN would not be), and the range flag
__device__ __forceinline__ float masked_result(float y, bool in_cube, int *flag)
{
    if (in_cube && not_finite(y)) *flag = 1;
    return in_cube ? y : 0.0f;
}

__device__ __forceinline__ f32x16 ld16(const float *p)
{
    const float4 *q = reinterpret_cast<const float4 *>(p);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    f32x16 v;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
    return v;
}

// lo halves of a pair: f16(v - hi), the subtraction exact in f32 (hi is a truncation of v), ONE rounding
// (nearest-even) into f16 - v_fma_mixlo/hi_f16 computes fma(f16 hi, -1.0, f32 v) straight into the low / high
// half of the packed result: 2 issue slots where `v - (float)hi` + cvt_pkrtz takes 5 (2 x v_cvt_f32_f16,
// 2 x v_sub_f32, v_cvt_pkrtz).  The activation VALU work is dynamic power the matrix pipe cannot use: the
// kernel is power-bound (tools/mlp_power_probe.py), so every removed VALU instruction is time.
__device__ __forceinline__ fp16x2 residual_pair(fp16x2 hh, float v0, float v1)
{
    const int hb = __builtin_bit_cast(int, hh);
    int lb;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(v1));
    return __builtin_bit_cast(fp16x2, lb);
}

// x -> (hi, lo) with hi = f16_rtz(x), lo = f16_rne(x - hi); 8 values -> one MFMA operand each
__device__ __forceinline__ void split8(const float *v, half8 &hi, half8 &lo)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]);
        const fp16x2 l = residual_pair(h, v[2 * q], v[2 * q + 1]);
        hi[2 * q] = (_Float16)h[0]; hi[2 * q + 1] = (_Float16)h[1];
        lo[2 * q] = (_Float16)l[0]; lo[2 * q + 1] = (_Float16)l[1];
    }
}

// LeakyReLU(0.01) of an accumulator value with the weight scale divided out, in TWO VALU operations:
//   max(x, 0.01 x) = 0.505 x + 0.495 |x|,  x = a * inv   ->   fma(|a|, 0.495 inv, (0.505 inv) * a)
// instead of three (a * inv, 0.01 * x, max).  0.505f + 0.495f == 1 exactly; the negative slope comes out as 0.00999999 instead of
// 0.00999999978 (1e-8 of |x| - a twentieth of the 2^-22 the split product carries) and the positive branch takes one more
// rounding.  Round 3: the activation work hides in the MFMA shadow but not its energy - 14.04-14.18 vs 14.14-14.26 ms standalone.
// ICON_ACT_EXACT=1 builds the three-operation form.
#ifndef ICON_ACT_EXACT
#define ICON_ACT_EXACT 0
#endif
struct LeakyK { float inv, p, q; };          // 1 / scale, 0.505 / scale, 0.495 / scale
__device__ __forceinline__ float leaky_scaled(float a, const LeakyK &k)
{
#if ICON_ACT_EXACT
    const float x = a * k.inv;
    return fmaxf(x, 0.01f * x);
#else
    return fmaf(fabsf(a), k.q, k.p * a);
#endif
}

// activation step of a finished tile: undo the weight scale, LeakyReLU(0.01), split for the next GEMM
__device__ __forceinline__ void activate_split(const f32x16 &acc, const LeakyK &inv, half8 hi[2], half8 lo[2])
{
    float v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = leaky_scaled(acc[t], inv);
    split8(v, hi[0], lo[0]);
    split8(v + 8, hi[1], lo[1]);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ half8 lds_op(const char *buf, int slot, int lane)
{
#if defined(ICON_EXP_SLOTWRAP)
    if (slot >= 8) slot &= 7;                  // (the resident W0 region is addressed with slots < 32 through the same helper: keep those)
#endif
    return *reinterpret_cast<const half8 *>(buf + slot * 1024 + lane * 16);
}

// every wave DMAs 1 KiB pieces round-robin: global [piece][lane][16 B] -> LDS, same order
// Timing-only experiment switches (tools/exp_fused.sh builds variants with -DICON_EXP_*; every one of them gives WRONG
// results - they price one component of the kernel by removing it): NO_AREADS - one A-operand group per chunk, reused;
// NO_DMA - no global->LDS weight stream; NO_ACT - no LeakyReLU / hi-lo split VALU work; NO_BAR - no chunk barriers.
#if defined(ICON_EXP_NO_BAR)
#define ICON_CHUNK_BARRIER() do { } while (0)
#else
#define ICON_CHUNK_BARRIER() __syncthreads()
#endif

__device__ __forceinline__ void issue_units(const char *src, char *buf, int units, int wave, int lane)
{
#if defined(ICON_EXP_NO_DMA)
    if (units != kW0Bytes / 1024) return;
#endif
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
#if defined(ICON_EXP_GRAY)
// experiment: every MFMA shares one operand with its predecessor (does operand-latch toggling cost energy?)
#define TRIPLE2(ACC, M0, AH, AL, BH, BL)                                     \
    ACC[M0] = MFMA16(AH[0], BH, ACC[M0]); ACC[M0] = MFMA16(AH[0], BL, ACC[M0]); \
    ACC[M0 + 1] = MFMA16(AH[1], BL, ACC[M0 + 1]); ACC[M0 + 1] = MFMA16(AH[1], BH, ACC[M0 + 1]); \
    ACC[M0 + 1] = MFMA16(AL[1], BH, ACC[M0 + 1]); ACC[M0] = MFMA16(AL[0], BH, ACC[M0]);
#else
#define TRIPLE2(ACC, M0, AH, AL, BH, BL)                                     \
    ACC[M0] = MFMA16(AH[0], BH, ACC[M0]); ACC[M0 + 1] = MFMA16(AH[1], BH, ACC[M0 + 1]); \
    ACC[M0] = MFMA16(AH[0], BL, ACC[M0]); ACC[M0 + 1] = MFMA16(AH[1], BL, ACC[M0 + 1]); \
    ACC[M0] = MFMA16(AL[0], BH, ACC[M0]); ACC[M0 + 1] = MFMA16(AL[1], BH, ACC[M0 + 1]);
#endif

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
__device__ __forceinline__ void act_part(const f32x16 &acc, int k, const LeakyK &inv, half8 (&nh)[2], half8 (&nl)[2])
{
#if defined(ICON_EXP_NO_ACT)
    {   // keep the data dependence on the accumulator (one v_mov-class op per pair), drop the arithmetic
        const int u = k >> 2, q = k & 3;
        const fp16x2 raw = __builtin_bit_cast(fp16x2, __float_as_int(acc[2 * k]) ^ __float_as_int(inv.inv));
        nh[u][2 * q] = (_Float16)raw[0]; nh[u][2 * q + 1] = (_Float16)raw[1];
        nl[u][2 * q] = (_Float16)raw[1]; nl[u][2 * q + 1] = (_Float16)raw[0];
        return;
    }
#endif

    const float v0 = leaky_scaled(acc[2 * k], inv), v1 = leaky_scaled(acc[2 * k + 1], inv);      // 7 VALU per pair with the split below
    fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    fp16x2 ll = residual_pair(hh, v0, v1);
#if defined(ICON_EXP_ACT2X)
    {   // the same 9 VALU once more on the same data, results discarded: what does a VALU instruction cost on real data?
        float y0 = acc[2 * k], y1 = acc[2 * k + 1];
        asm volatile("" : "+v"(y0), "+v"(y1));
        const float z0 = y0 * inv.inv, z1 = y1 * inv.inv;
        const float u0 = fmaxf(z0, 0.01f * z0), u1 = fmaxf(z1, 0.01f * z1);
        fp16x2 h2 = __builtin_amdgcn_cvt_pkrtz(u0, u1);
        fp16x2 l2 = residual_pair(h2, u0, u1);
        int a2 = __builtin_bit_cast(int, h2), b2 = __builtin_bit_cast(int, l2);
        asm volatile("" :: "v"(a2), "v"(b2));
    }
#endif
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
                                          half8 xhi, half8 xlo, const LeakyK &inv0, int h, int lane, int wave,
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
#if defined(ICON_EXP_NO_AREADS)
        if (g == 0) load_group(L, 1, lane, a[1]);
#else
        if (g + 1 < 8) load_group(L, g + 1, lane, a[(g + 1) & 1]);
#endif
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
                                         f32x16 (&acc1)[8], f32x16 (&acc2)[4], half8 xhi, half8 xlo, const LeakyK &inv1,
                                         int lane, int wave, half8 (&bh)[2], half8 (&bl)[2], int next_chunk = (Q < 3) ? 17 + Q : -1)
{
    // next_chunk: the chunk DMA'd into `nxt` while this one is multiplied; the persistent kernel passes 0 for
    // Q == 3 (chunk 0 of its NEXT tile) - issued from inside this body so that it shares the alias scopes
    if (next_chunk >= 0) issue_chunk(image, nxt, next_chunk, wave, lane);
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
#if defined(ICON_EXP_NO_AREADS)
            if (g == 0) load2(1, a[1]);
#else
            if (g + 1 < 4) load2(g + 1, a[(g + 1) & 1]);
#endif
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


}  // namespace icon
