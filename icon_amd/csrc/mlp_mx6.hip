// mlp_mx6.hip - the occupancy MLP with the split-precision cross terms on the MX (block-scaled)
// fp6 matrix instructions of gfx950.
//
// Same contract and register-chained layer structure as mlp_f16x3.hip (reference: lib/net/MLP.py:49-72,
// lib/net/HGPIFuNet.py:128-133,363).  mlp_f16x3 evaluates every product as
//       W*h ~= W_hi*h_hi + W_hi*h_lo + W_lo*h_hi          (three f16 MFMAs per K=16)
// The two cross terms are 2^-11 of the main term, so 4 significant bits are enough for them:
//       W*h ~= W_hi*h_hi  (v_mfma_f32_32x32x16_f16, 4 per K=64)
//            + fp6(W)*fp6(h_lo) + fp6(W_lo)*fp6(h)        (v_mfma_scale_f32_32x32x64_f8f6f4, e2m3, 2 per K=64)
// with one shared power-of-two exponent per 32-element block (per weight row / per point) in the
// MX scale operands.  6 instead of 12 MFMA issue slots per K=64; the error against the float64 MLP
// is ~2e-5 worst case / 1e-6 mean on the occupancy (tools/sim_mx6.py models it, tests bound it;
// north-star tolerance 1e-4).  Layer 0 (K=13) and the raw-input k-step of layer 2 stay on the
// 3 x f16 scheme, layer 3 on the VALU in f32.
//
// Structure (different from mlp_f16x3, because the conversion work per chunk is larger):
//  * a chunk is K=64 hidden channels: 32 KiB f16 hi operands + 2 x 12 KiB fp6 operands + 1 KiB of
//    e8m0 scales, DMA'd global->LDS into a double buffer (2 x 57 KiB) while the previous chunk is
//    being multiplied.
//  * no software pipelining inside a wavefront.  Each wave alternates a VALU phase V(k) (layer-0
//    tiles of chunk k -> LeakyReLU -> f16 hi / residual -> block exponent -> fp6) and an MFMA phase
//    M(k) (48 MFMAs).  The two waves that share a SIMD (wave w and w+4 of the 512-thread workgroup)
//    run the SAME instruction stream with the per-chunk barrier at different points of it - waves
//    0-3 do V(k) M(k) | barrier, waves 4-7 do M(k) V(k+1) | barrier - so in every barrier interval
//    one wave of the SIMD is in its VALU phase while the other feeds the matrix pipe.
#include "common.h"

#include <cmath>
#include <cstring>

namespace icon {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int kMxBlock = 512;                  // 8 waves x 32 points
constexpr int kMxPts = (kMxBlock / 64) * 32;
constexpr int kMxBuf = 57 * 1024;
constexpr int kMxW0Off = 2 * kMxBuf;
constexpr int kMxW0Bytes = 32 * 1024;
constexpr int kMxSideOff = kMxW0Off + kMxW0Bytes;
constexpr int kMxSideFloats = 512 + 256 + 128 + 144;          // b0 | b1 | b2 | w3 (same as mlp_f16x3)
constexpr int kMxLds = kMxSideOff + 4352;                     // 150.25 KiB

// image: [W0 32 KiB][layer-1 chunks 0..7: 57 KiB][layer-2 chunks 8..10: 29 KiB, 11: 37 KiB]
__host__ __device__ constexpr int mx_chunk_units(int k) { return k < 8 ? 57 : (k < 11 ? 29 : 37); }
__host__ __device__ constexpr int mx_chunk_offset(int k) { return k < 8 ? 32 + 57 * k : 488 + 29 * (k - 8); }
constexpr size_t kMxImageBytes = (size_t)(488 + 3 * 29 + 37) * 1024;   // 612 KiB

// byte offsets inside a chunk with NT output tiles (8: layer 1, 4: layer 2)
template <int NT> struct MxLay {
    static constexpr int HI = 0;                       // [k-step u 0..3][tile] x 1 KiB   f16 hi of W
    static constexpr int W6A = NT * 4096;              // [tile] x 1 KiB    fp6(W), first 16 B per lane
    static constexpr int W6B = W6A + NT * 1024;        // [tile] x 512 B    fp6(W), last 8 B per lane
    static constexpr int L6A = W6B + NT * 512;         // fp6(W - W_hi), same split
    static constexpr int L6B = L6A + NT * 1024;
    static constexpr int SC = L6B + NT * 512;          // e8m0 bytes: [lane][W tiles..][W_lo tiles..]
    static constexpr int RAW = 29 * 1024;              // (last layer-2 chunk) raw-input k-step, hi/lo f16
};

struct MlpMx6Dev {
    const char *image;
    const float *side;
    float b3;
    float inv0, inv1, inv2;
    int c0;
};

__device__ __forceinline__ f32x16 mx_ld16(const float *p)
{
    const float4 *q = reinterpret_cast<const float4 *>(p);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    f32x16 v;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
    return v;
}

#define MX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
// fp6 (e2m3) x fp6, byte OPA of the A scale register, byte 0 of the B scale register
#define MX_MFMA6(a, b, c, OPA, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 2, 2, (OPA), (sa), 0, (sb))

__device__ __forceinline__ half8 mx_op(const char *buf, int slot, int lane)
{
    return *reinterpret_cast<const half8 *>(buf + slot * 1024 + lane * 16);
}
__device__ __forceinline__ i32x8 mx_op6(const char *buf, int offA, int offB, int tile, int lane)
{
    const uint4 p = *reinterpret_cast<const uint4 *>(buf + offA + tile * 1024 + lane * 16);
    const uint2 r = *reinterpret_cast<const uint2 *>(buf + offB + tile * 512 + lane * 8);
    i32x8 v = {(int)p.x, (int)p.y, (int)p.z, (int)p.w, (int)r.x, (int)r.y, 0, 0};
    return v;
}

// LDS-DMA of 1 KiB pieces (global [piece][lane][16 B] -> LDS, same order), every wave takes pieces
// round-robin.  Issued through inline asm on purpose: the compiler's wait-count pass treats a
// pending global_load_lds as a "flat" access and from then on turns EVERY LDS wait into
// lgkmcnt(0) / vmcnt(0), which serialises the operand prefetch of the MFMA phase.  Hidden from that
// pass, the ds_read waits are counted ones; mx_dma_wait() supplies the vmcnt(0) before the barrier.
__device__ __forceinline__ void mx_issue_units(const char *src, char *buf, int units, int wave, int lane)
{
    for (int u = wave; u < units; u += kMxBlock / 64) {
        const char *g = src + u * 1024 + lane * 16;
        const uint32_t l = (uint32_t)(uintptr_t)(lvoid_t *)(buf + u * 1024);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(l), "v"(g) : "memory");
    }
}
__device__ __forceinline__ void mx_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
__device__ __forceinline__ void mx_issue_chunk(const char *image, char *buf, int k, int wave, int lane)
{
    mx_issue_units(image + (size_t)mx_chunk_offset(k) * 1024, buf, mx_chunk_units(k), wave, lane);
}

// x -> (hi, lo) f16 pairs for the 3 x f16 k-steps (raw input), as in mlp_f16x3
__device__ __forceinline__ void mx_split8(const float *v, half8 &hi, half8 &lo)
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

// B operands of one K=64 chunk for this lane's point: 32 channels (the lane's 16 rows of two
// accumulator tiles), element p = 16*T + t  <->  channel 32*(tile T) + rho(t, lane>>5)
struct MxB {
    half32 hi;        // f16(h), round to nearest: four MFMA operands (elements 8u..8u+7)
    i32x8 h6, l6;     // fp6(h / 2^(sh-127)), fp6((h - hi) / 2^(sl-127))
    int sh, sl;       // e8m0 block scales
};

// VALU phase: two finished accumulator tiles -> LeakyReLU -> MxB
__device__ __forceinline__ void mx_make_b(const f32x16 &ta, const f32x16 &tb, float inv, MxB &b)
{
    half32 hi, lo;
    float mx = 0.0f;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float a0 = T ? tb[2 * q] : ta[2 * q], a1 = T ? tb[2 * q + 1] : ta[2 * q + 1];
            const float x0 = a0 * inv, x1 = a1 * inv;
            const float v0 = fmaxf(x0, 0.01f * x0), v1 = fmaxf(x1, 0.01f * x1);
            const f32x2 vv = {v0, v1};
            const half2v hh = __builtin_convertvector(vv, half2v);          // v_cvt_pk_f16_f32 (RTN)
            const f32x2 rr = {v0 - (float)hh[0], v1 - (float)hh[1]};        // exact
            const half2v ll = __builtin_convertvector(rr, half2v);
            mx = fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1)));
            hi[16 * T + 2 * q] = hh[0]; hi[16 * T + 2 * q + 1] = hh[1];
            lo[16 * T + 2 * q] = ll[0]; lo[16 * T + 2 * q + 1] = ll[1];
        }
    }
    // shared exponent: 2^(E-2) with E = floor(log2(max|h|)) puts the block maximum in [4, 8) of the
    // e2m3 range (7.5 saturates); |h - hi| <= 2^(E-11), so its block uses 2^(E-13)
    const int eb = (int)(__float_as_uint(mx) >> 23);
    const int sh = max(eb, 14) - 2, sl = sh - 11;
    const u32x6 ph = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hi, __uint_as_float((unsigned)sh << 23));
    const u32x6 pl = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(lo, __uint_as_float((unsigned)sl << 23));
    b.hi = hi;
    b.h6 = i32x8{(int)ph[0], (int)ph[1], (int)ph[2], (int)ph[3], (int)ph[4], (int)ph[5], 0, 0};
    b.l6 = i32x8{(int)pl[0], (int)pl[1], (int)pl[2], (int)pl[3], (int)pl[4], (int)pl[5], 0, 0};
    b.sh = sh; b.sl = sl;
}

#define MX_BHI(b, u) __builtin_shufflevector((b).hi, (b).hi, 8 * (u), 8 * (u) + 1, 8 * (u) + 2, 8 * (u) + 3, 8 * (u) + 4, 8 * (u) + 5, 8 * (u) + 6, 8 * (u) + 7)

// fp6 operand j of a chunk: j < NT -> fp6(W) of tile j, else fp6(W - W_hi) of tile j - NT
template <int NT>
__device__ __forceinline__ i32x8 mx_ld6(const char *__restrict__ L, int j, int lane)
{
    using Y = MxLay<NT>;
    return j < NT ? mx_op6(L, Y::W6A, Y::W6B, j, lane) : mx_op6(L, Y::L6A, Y::L6B, j - NT, lane);
}

// fp6 MFMA J of the chunk (J literal: the scale byte select must be an integer constant)
#define MX_OP6(NT, J)                                                                                               \
    if constexpr ((J) < 2 * NT) {                                                                                   \
        constexpr int t6 = (J) % NT;                                                                                \
        const i32x8 cur = w6[(J) & 3];                                                                              \
        if ((J) + 4 < 2 * NT) w6[(J) & 3] = mx_ld6<NT>(L, (J) + 4, lane);                                           \
        if ((J) < NT) acc[t6] = MX_MFMA6(cur, b.l6, acc[t6], t6 & 3, scw[t6 >> 2], b.sl);                            \
        else          acc[t6] = MX_MFMA6(cur, b.h6, acc[t6], t6 & 3, scl[t6 >> 2], b.sh);                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }

// MFMA phase of a chunk with NT output tiles, K-major: for each of the four f16 k-steps all tiles,
// then fp6(W) x fp6(h_lo) for all tiles, then fp6(W_lo) x fp6(h).  Consecutive MFMAs hit different
// accumulators; every A operand is requested 8 (f16) / 4 (fp6) MFMAs ahead of its use through a
// rotating register window, so only ~32 operand registers are live beside the accumulators.
template <int NT>
__device__ __forceinline__ void mx_m_phase(const char *__restrict__ L, int lane, f32x16 (&acc)[NT], const MxB &b)
{
    int scw[2], scl[2];
    if (NT == 8) {
        const int4 s = *reinterpret_cast<const int4 *>(L + MxLay<8>::SC + lane * 16);
        scw[0] = s.x; scw[1] = s.y; scl[0] = s.z; scl[1] = s.w;
    } else {
        const int2 s = *reinterpret_cast<const int2 *>(L + MxLay<4>::SC + lane * 8);
        scw[0] = s.x; scl[0] = s.y; scw[1] = 0; scl[1] = 0;
    }
    half8 a[8];
    i32x8 w6[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = mx_op(L, i, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) {          // f16 op i: k-step i / NT, tile i % NT; its LDS slot is i
        const half8 cur = a[i & 7];
        if (i + 8 < 4 * NT) a[i & 7] = mx_op(L, i + 8, lane);
        if (i >= 4 * NT - 4) w6[i - (4 * NT - 4)] = mx_ld6<NT>(L, i - (4 * NT - 4), lane);
        const int u = i / NT;
        const half8 bu = u == 0 ? MX_BHI(b, 0) : u == 1 ? MX_BHI(b, 1) : u == 2 ? MX_BHI(b, 2) : MX_BHI(b, 3);
        acc[i % NT] = MX_MFMA16(cur, bu, acc[i % NT]);
        __builtin_amdgcn_sched_barrier(0);
    }
    MX_OP6(NT, 0) MX_OP6(NT, 1) MX_OP6(NT, 2) MX_OP6(NT, 3) MX_OP6(NT, 4) MX_OP6(NT, 5) MX_OP6(NT, 6) MX_OP6(NT, 7)
    MX_OP6(NT, 8) MX_OP6(NT, 9) MX_OP6(NT, 10) MX_OP6(NT, 11) MX_OP6(NT, 12) MX_OP6(NT, 13) MX_OP6(NT, 14) MX_OP6(NT, 15)
    // keep the phase together: without a use here the compiler sinks the tail of the MFMA chain below
    // the VALU phase that follows (and spills the operands it had already loaded)
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
}

// layer 0, hidden tile c (32 channels): 3 MFMAs (3 x f16) from the resident W0 region
__device__ __forceinline__ f32x16 mx_l0_tile(const char *__restrict__ W0, const float *__restrict__ sb0, int c, half8 xhi, half8 xlo,
                                             int h, int lane)
{
    f32x16 h0 = mx_ld16(sb0 + (c * 2 + h) * 16);
    const half8 a_hi = mx_op(W0, 2 * c, lane), a_lo = mx_op(W0, 2 * c + 1, lane);
    h0 = MX_MFMA16(a_hi, xhi, h0); h0 = MX_MFMA16(a_hi, xlo, h0); h0 = MX_MFMA16(a_lo, xhi, h0);
    return h0;
}
__device__ __forceinline__ void mx_v_l1(const char *__restrict__ W0, const float *__restrict__ sb0, int k, half8 xhi, half8 xlo,
                                        float inv0, int h, int lane, MxB &b)
{
    const f32x16 ta = mx_l0_tile(W0, sb0, 2 * k, xhi, xlo, h, lane);
    const f32x16 tb = mx_l0_tile(W0, sb0, 2 * k + 1, xhi, xlo, h, lane);
    mx_make_b(ta, tb, inv0, b);
}

// one barrier interval of layer 1: chunk k is multiplied while chunk k+1 lands in the other buffer.
// `late` waves (4-7) arrive with b = B(k) already built and leave with B(k+1) (or the first layer-2
// operands when k == 7).
__device__ __forceinline__ void mx_l1_step(const char *__restrict__ cur, char *__restrict__ nxt, const char *__restrict__ W0,
                                           const float *__restrict__ sb0, const char *image, int k, f32x16 (&acc1)[8],
                                           half8 xhi, half8 xlo, float inv0, float inv1, int h, int lane, int wave, bool late,
                                           MxB &b)
{
    mx_issue_chunk(image, nxt, k + 1, wave, lane);
    if (!late) mx_v_l1(W0, sb0, k, xhi, xlo, inv0, h, lane, b);
    mx_m_phase<8>(cur, lane, acc1, b);
    if (late) {
        if (k < 7) mx_v_l1(W0, sb0, k + 1, xhi, xlo, inv0, h, lane, b);
        else mx_make_b(acc1[0], acc1[1], inv1, b);
    }
}

template <int Q>
__device__ __forceinline__ void mx_l2_step(const char *__restrict__ cur, char *__restrict__ nxt, const char *image,
                                           f32x16 (&acc1)[8], f32x16 (&acc2)[4], float inv1, int lane, int wave, bool late, MxB &b)
{
    if (Q < 3) mx_issue_chunk(image, nxt, 9 + Q, wave, lane);
    if (!late) mx_make_b(acc1[2 * Q], acc1[2 * Q + 1], inv1, b);
    mx_m_phase<4>(cur, lane, acc2, b);
    if (late && Q < 3) mx_make_b(acc1[Q < 3 ? 2 * Q + 2 : 0], acc1[Q < 3 ? 2 * Q + 3 : 1], inv1, b);
}

template <bool MASK>
__global__ __launch_bounds__(kMxBlock, 2) void k_mlp_mx6(const float *__restrict__ X, int64_t N, float *__restrict__ out, MlpMx6Dev w)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const bool late = wave >= 4;                  // waves w and w+4 share a SIMD
    const int64_t base = ((int64_t)blockIdx.x * (kMxBlock / 64) + wave) * 32;
    const int64_t pi = min(base + j, N - 1);      // waves past the end still help with the DMA + barriers

    mx_issue_units(w.image, smem + kMxW0Off, kMxW0Bytes / 1024, wave, lane);     // resident layer-0 operands
    float *side = reinterpret_cast<float *>(smem + kMxSideOff);
    for (int i = threadIdx.x; i < kMxSideFloats; i += kMxBlock) side[i] = w.side[i];
    const float *sb0 = side, *sb1 = side + 512, *sb2 = side + 768, *sw3 = side + 896;

    half8 xhi, xlo;
    {
        float xr[8];
        const float4 *q = reinterpret_cast<const float4 *>(X + pi * kXRow + 8 * h);
        const float4 a = q[0], b = q[1];
        xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = b.x; xr[5] = b.y; xr[6] = b.z; xr[7] = b.w;
#pragma unroll
        for (int s = 0; s < 8; ++s) xr[s] = (s + 8 * h < w.c0) ? xr[s] : 0.0f;
        mx_split8(xr, xhi, xlo);
    }
    mx_issue_chunk(w.image, smem, 0, wave, lane);
    mx_dma_wait();
    __syncthreads();   // side arrays visible, W0 + chunk 0 landed (the barrier's release waits for the LDS-DMA)

    f32x16 acc1[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc1[m] = mx_ld16(sb1 + (m * 2 + h) * 16);

    // ---- layers 0 + 1: one chunk = 64 hidden channels ---------------------------------------------
    MxB b;
    if (late) mx_v_l1(smem + kMxW0Off, sb0, 0, xhi, xlo, w.inv0, h, lane, b);
    else { b.hi = (half32)(_Float16)0; b.h6 = i32x8{0, 0, 0, 0, 0, 0, 0, 0}; b.l6 = b.h6; b.sh = 127; b.sl = 127; }
    for (int k = 0; k < 8; ++k) {
        mx_l1_step(smem + (k & 1) * kMxBuf, smem + ((k + 1) & 1) * kMxBuf, smem + kMxW0Off, sb0, w.image, k, acc1, xhi, xlo,
                   w.inv0, w.inv1, h, lane, wave, late, b);
        mx_dma_wait();
    __syncthreads();   // all waves done with this buffer AND the next chunk has landed
    }

    // ---- layer 2: K = 256 (registers) + 16 (raw input) ---------------------------------------------
    f32x16 acc2[4];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = mx_ld16(sb2 + (m2 * 2 + h) * 16);
    mx_l2_step<0>(smem, smem + kMxBuf, w.image, acc1, acc2, w.inv1, lane, wave, late, b);
    mx_dma_wait();
    __syncthreads();
    mx_l2_step<1>(smem + kMxBuf, smem, w.image, acc1, acc2, w.inv1, lane, wave, late, b);
    mx_dma_wait();
    __syncthreads();
    mx_l2_step<2>(smem, smem + kMxBuf, w.image, acc1, acc2, w.inv1, lane, wave, late, b);
    mx_dma_wait();
    __syncthreads();
    mx_l2_step<3>(smem + kMxBuf, smem, w.image, acc1, acc2, w.inv1, lane, wave, late, b);

    // raw-input k-step of layer 2 (3 x f16) and layer 3 on the VALU; the input row is re-read here
    // (no LDS-DMA is in flight any more) instead of being kept in registers through layers 1-2
    float xr[8];
    {
        const float *Xp = X;
        asm volatile("" : "+s"(Xp));      // a fresh load, not the prologue's value kept alive across layers 1-2
        const float4 *q = reinterpret_cast<const float4 *>(Xp + pi * kXRow + 8 * h);
        const float4 a = q[0], c = q[1];
        xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = c.x; xr[5] = c.y; xr[6] = c.z; xr[7] = c.w;
#pragma unroll
        for (int s = 0; s < 8; ++s) xr[s] = (s + 8 * h < w.c0) ? xr[s] : 0.0f;
        mx_split8(xr, xhi, xlo);
    }
    {
        const char *L = smem + kMxBuf + MxLay<4>::RAW;
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) {
            const half8 a_hi = mx_op(L, 2 * m2, lane), a_lo = mx_op(L, 2 * m2 + 1, lane);
            acc2[m2] = MX_MFMA16(a_hi, xhi, acc2[m2]);
            acc2[m2] = MX_MFMA16(a_hi, xlo, acc2[m2]);
            acc2[m2] = MX_MFMA16(a_lo, xhi, acc2[m2]);
        }
    }
    float maskf = 1.0f;
    if (MASK) {
        const uint32_t code = (uint32_t)__float_as_int(X[pi * kXRow + kCodeSlot]);
        maskf = (code & kCodeInCube) ? 1.0f : 0.0f;
    }
    const float *w3 = sw3 + h * 72;
    float part = 0.0f;
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const f32x16 wv = mx_ld16(w3 + m2 * 16);
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
// e2m3 code of a (already divided by the block scale), round to nearest even, saturating at 7.5;
// decode (verified on gfx950, tools/probes/mx_decode.hip): e == 0 ? m/8 : (1 + m/8) * 2^(e-1)
static uint8_t fp6_e2m3(float v)
{
    const uint8_t sign = std::signbit(v) ? 0x20 : 0;
    const float a = std::fabs(v);
    if (!(a < 7.75f)) return sign | 0x1f;
    if (a < 1.0f) {
        const int m = (int)std::nearbyint(a * 8.0f);           // 0..8; 8 -> 1.0 = code 0x08
        return sign | (uint8_t)m;
    }
    int e = a < 2.0f ? 1 : (a < 4.0f ? 2 : 3);
    int m = (int)std::nearbyint(std::ldexp(a, 4 - e));        // 8..16
    if (m == 16) { m = 8; ++e; }
    if (e > 3) return sign | 0x1f;
    return sign | (uint8_t)((e << 3) | (m - 8));
}

// 32 values -> e8m0 block scale + 32 six-bit codes packed LSB first into 6 dwords
static uint8_t fp6_block(const float *v, uint32_t *out6)
{
    float mx = 0.f;
    for (int p = 0; p < 32; ++p) mx = std::max(mx, std::fabs(v[p]));
    int e8 = 127;
    if (mx > 0.f && std::isfinite(mx)) {
        int ex;
        (void)std::frexp(mx, &ex);                 // mx = f * 2^ex, f in [0.5, 1)  ->  floor(log2 mx) = ex - 1
        e8 = std::min(std::max(ex - 1 - 2 + 127, 1), 254);
    }
    const float inv = std::ldexp(1.0f, 127 - e8);
    uint64_t acc[3] = {0, 0, 0};
    for (int p = 0; p < 32; ++p) {
        const uint64_t code = fp6_e2m3(v[p] * inv);
        const int bit = 6 * p;
        acc[bit / 64] |= code << (bit % 64);
        if (bit % 64 > 58) acc[bit / 64 + 1] |= code >> (64 - bit % 64);
    }
    for (int i = 0; i < 3; ++i) { out6[2 * i] = (uint32_t)acc[i]; out6[2 * i + 1] = (uint32_t)(acc[i] >> 32); }
    return (uint8_t)e8;
}

int mlp_pack_mx6(icon_mlp *m, const std::vector<std::vector<float>> &W, const std::vector<std::vector<float>> &B, hipStream_t st)
{
    const int c0 = m->c0;
    const float s0 = pick_scale(W[0]), s1 = pick_scale(W[1]), s2 = pick_scale(W[2]);
    std::vector<uint8_t> img(kMxImageBytes, 0);
    auto rho = [](int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; };
    auto put16 = [&](size_t byte_off, int lane, int e, uint16_t v) { memcpy(&img[byte_off + (size_t)lane * 16 + 2 * e], &v, 2); };
    // 3 x f16 operand pair (hi in `slot`, lo in `slot + 1`) of a K=16 step over the raw input
    auto put_raw = [&](size_t base, int slot, int lane, int e, float wv, float scale) {
        const float ws = wv * scale;
        const uint16_t hi = f32_to_f16_rtn(ws);
        put16(base + (size_t)slot * 1024, lane, e, hi);
        put16(base + (size_t)(slot + 1) * 1024, lane, e, f32_to_f16_rtn(ws - f16_to_f32(hi)));
    };
    const int ci2 = 256 + c0;
    for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 31, g = lane >> 5;
        for (int e = 0; e < 8; ++e) {
            const int slot0 = 8 * g + e;
            for (int c = 0; c < 16; ++c) put_raw(0, 2 * c, lane, e, slot0 < c0 ? W[0][(size_t)(32 * c + i) * c0 + slot0] : 0.f, s0);
            for (int m2 = 0; m2 < 4; ++m2)
                put_raw((size_t)mx_chunk_offset(11) * 1024 + MxLay<4>::RAW, 2 * m2, lane, e,
                        slot0 < c0 ? W[2][(size_t)(32 * m2 + i) * ci2 + 256 + slot0] : 0.f, s2);
        }
    }
    // K = 64 chunks of layers 1 and 2: element p of the lane's block <-> hidden channel 64*c + 32*(p>>4) + rho(p&15, g)
    auto pack_chunk = [&](int chunk, int nt, const std::vector<float> &Wl, int cin, int kbase, float scale) {
        const size_t cb = (size_t)mx_chunk_offset(chunk) * 1024;
        const int w6a = nt * 4096, w6b = w6a + nt * 1024, l6a = w6b + nt * 512, l6b = l6a + nt * 1024, sc = l6b + nt * 512;
        for (int tile = 0; tile < nt; ++tile)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, g = lane >> 5;
                float ws[32], wl[32];
                for (int p = 0; p < 32; ++p) {
                    const float v = Wl[(size_t)(32 * tile + i) * cin + kbase + 32 * (p >> 4) + rho(p & 15, g)] * scale;
                    const uint16_t hi = f32_to_f16_rtn(v);
                    ws[p] = v; wl[p] = v - f16_to_f32(hi);
                    put16(cb + (size_t)((p >> 3) * nt + tile) * 1024, lane, p & 7, hi);
                }
                uint32_t c6[6];
                const uint8_t ew = fp6_block(ws, c6);
                memcpy(&img[cb + w6a + (size_t)tile * 1024 + lane * 16], c6, 16);
                memcpy(&img[cb + w6b + (size_t)tile * 512 + lane * 8], c6 + 4, 8);
                const uint8_t el = fp6_block(wl, c6);
                memcpy(&img[cb + l6a + (size_t)tile * 1024 + lane * 16], c6, 16);
                memcpy(&img[cb + l6b + (size_t)tile * 512 + lane * 8], c6 + 4, 8);
                img[cb + sc + (size_t)lane * (2 * nt) + tile] = ew;
                img[cb + sc + (size_t)lane * (2 * nt) + nt + tile] = el;
            }
    };
    for (int c = 0; c < 8; ++c) pack_chunk(c, 8, W[1], 512, 64 * c, s1);
    for (int q = 0; q < 4; ++q) pack_chunk(8 + q, 4, W[2], ci2, 64 * q, s2);

    std::vector<float> side(kMxSideFloats, 0.f);
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
    ICON_HIP(hipMalloc((void **)&m->d_mx6, kMxImageBytes + side_bytes));
    ICON_HIP(hipMemcpyAsync(m->d_mx6, img.data(), kMxImageBytes, hipMemcpyHostToDevice, st));
    ICON_HIP(hipMemcpyAsync(m->d_mx6 + kMxImageBytes, side.data(), side_bytes, hipMemcpyHostToDevice, st));
    ICON_HIP(hipStreamSynchronize(st));
    return ICON_OK;
}

int mlp_launch_mx6(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, bool mask, hipStream_t st)
{
    if (N <= 0) return ICON_OK;
    MlpMx6Dev w;
    w.image = mlp->d_mx6;
    w.side = reinterpret_cast<const float *>(mlp->d_mx6 + kMxImageBytes);
    w.b3 = mlp->b3; w.inv0 = mlp->f16_inv[0]; w.inv1 = mlp->f16_inv[1]; w.inv2 = mlp->f16_inv[2]; w.c0 = mlp->c0;
    const int64_t nb = (N + kMxPts - 1) / kMxPts;
    ICON_ARG(nb < (1ll << 31), "mlp: N too large for one launch");
    static bool attr_set = false;
    if (!attr_set) {
        ICON_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_mx6<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMxLds));
        ICON_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_mx6<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMxLds));
        attr_set = true;
    }
    if (mask) hipLaunchKernelGGL(k_mlp_mx6<true>, dim3((unsigned)nb), dim3(kMxBlock), kMxLds, st, d_x, N, d_out, w);
    else      hipLaunchKernelGGL(k_mlp_mx6<false>, dim3((unsigned)nb), dim3(kMxBlock), kMxLds, st, d_x, N, d_out, w);
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

}  // namespace icon
