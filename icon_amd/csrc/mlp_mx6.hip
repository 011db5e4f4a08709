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
//  * phases instead of fine-grained software pipelining: per chunk a wave runs an MFMA phase M(k)
//    (48 MFMAs, operands prefetched through a rotating register window) and then a VALU phase V(k+1)
//    (LeakyReLU -> f16 hi / residual -> block exponent -> fp6 of the NEXT chunk's B operands).  The
//    six layer-0 MFMAs that produce V(k+1)'s input are issued inside the tail of M(k), where their
//    dependency latency is covered by the fp6 MFMAs around them.  256 registers per wave (2 waves per
//    SIMD) leave no room for a second B operand set, which is what interleaving V into M would need.
//  * persistent workgroups: one per CU, looping over point tiles.  Layer-0 weights and the side
//    arrays are staged once; chunk 0 and the input rows of the next tile are requested during the
//    last layer-2 step of the current one, so a tile starts without a cold prologue.
#include "common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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
constexpr int kDmaFirst = 20;                                 // of 57: pieces issued by the V-first waves
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
    int c0;
    int last_op;
    unsigned long long *trace;     // debug (ICON_AMD_MX6_TRACE): s_memtime stamps of workgroup 0, second tile
    int *flag;                     // raised when an in-cube result is not finite (operand beyond the f16 range): k_rescue_rows redoes the point in f32
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

#define MX_STAMP(slot) if (tr) { tr[slot] = __builtin_amdgcn_s_memtime(); }
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
// one piece: 64 lanes x 16 B from g (wave-uniform) + lane * 16 -> LDS at lds (wave-uniform) + lane * 16
__device__ __forceinline__ void mx_issue_piece(const char *g, char *lds, int lane16)
{
    const uint32_t l = (uint32_t)(uintptr_t)(lvoid_t *)lds;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(l), "v"(lane16), "s"(g) : "memory");
}
__device__ __forceinline__ void mx_issue_units(const char *src, char *buf, int units, int wave, int lane)
{
    for (int u = wave; u < units; u += kMxBlock / 64) mx_issue_piece(src + u * 1024, buf + u * 1024, lane * 16);
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

// VALU phase: two finished accumulator tiles -> LeakyReLU -> MxB.  (The accumulators carry the
// layer's power-of-two weight scale; LeakyReLU commutes with it and the next layer's packed weights
// divide it out, so there is no rescaling multiply here.)
// DMA: the LDS-DMA of the next chunk is issued from here, one 1 KiB piece per value pair - a piece
// costs its wave 50-100+ issue cycles, which inside the MFMA stream were matrix-pipe bubbles.  The
// waves that run this phase first in a barrier interval take the first kDmaFirst pieces, their SIMD
// partners (whose VALU phase comes after their MFMA phase and is off the interval's critical path)
// the rest.
struct MxDma { const char *src; char *dst; int first, last, widx, lane16; };    // pieces [first, last) of the chunk, 4 issuing waves

template <bool DMA>
__device__ __forceinline__ void mx_make_b(const f32x16 &ta, const f32x16 &tb, MxB &b, const MxDma &dma)
{
    half32 hi, lo;
    float mx = 0.0f;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float x0 = T ? tb[2 * q] : ta[2 * q], x1 = T ? tb[2 * q + 1] : ta[2 * q + 1];
            const float v0 = fmaxf(x0, 0.01f * x0), v1 = fmaxf(x1, 0.01f * x1);
            const f32x2 vv = {v0, v1};
            const half2v hh = __builtin_convertvector(vv, half2v);          // v_cvt_pk_f16_f32 (RTN)
            // residual v - hi (exact in f32) rounded once to f16, written straight into the packed pair:
            // fma(hi.f16, -1.0, v.f32) -> f16 low / high half.  The compiler's own selection for
            // `v - (float)hi` is 2 x v_cvt_f32_f16 + v_pk_add_f32 + v_cvt_pk_f16_f32 (5 issue slots).
            const int hb = __builtin_bit_cast(int, hh);
            int lb;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(v0));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(v1));
            const half2v ll = __builtin_bit_cast(half2v, lb);
            const _Float16 l0 = ll[0], l1 = ll[1];
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(v0)), __builtin_fabsf(v1));   // v_max3_f32 |.|
            hi[16 * T + 2 * q] = hh[0]; hi[16 * T + 2 * q + 1] = hh[1];
            lo[16 * T + 2 * q] = l0; lo[16 * T + 2 * q + 1] = l1;
            if (DMA) {
                const int p = 8 * T + q;                                          // slot 0..15 of this wave
                const int u = min(dma.first + dma.widx + 4 * p, dma.last - 1);     // past the end: repeat the final piece
                if (dma.first + 4 * p < dma.last + 3) mx_issue_piece(dma.src + u * 1024, dma.dst + u * 1024, dma.lane16);
            }
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
// layer-0 work slotted into the fp6 stage of an M phase: hidden tiles 2*cn, 2*cn+1 from l0.xhi/xlo into
// l0.h0n.  Layer-1 chunks (NT == 8): tile 0 before fp6 ops 2/4/6, tile 1 before ops 10/12/14 (a
// layer-2 chunk with L0H: ops 1/2/3 and 5/6/7 - measured slower, the extra live registers spill).  Operands are requested 1-2 ops earlier.
#define MX_L0_HOOK(NT, J)                                                                                           \
    if constexpr (NT == 8 || L0H) {                                                                                 \
        constexpr int half_ops = NT, step = NT / 4;          /* 8 ops per tile, MFMAs every 2nd op; or 4 and every op */ \
        constexpr int tl = (J) / half_ops, r = (J) % half_ops;                                                       \
        if (r == 0) {                                                                                               \
            const int c0t = 2 * l0.cn + tl;                                                                         \
            l0.h0n[tl] = mx_ld16(l0.sb0 + (c0t * 2 + l0.h) * 16);                                                   \
            l0a_hi = mx_op(l0.W0, 2 * c0t, lane); l0a_lo = mx_op(l0.W0, 2 * c0t + 1, lane);                         \
        }                                                                                                           \
        if (r == 1 * step) l0.h0n[tl] = MX_MFMA16(l0a_hi, l0.xhi, l0.h0n[tl]);                                       \
        if (r == 2 * step) l0.h0n[tl] = MX_MFMA16(l0a_hi, l0.xlo, l0.h0n[tl]);                                       \
        if (r == 3 * step) l0.h0n[tl] = MX_MFMA16(l0a_lo, l0.xhi, l0.h0n[tl]);                                       \
    }

#define MX_OP6(NT, J)                                                                                               \
    if constexpr ((J) < 2 * NT) {                                                                                   \
        constexpr int t6 = (J) % NT;                                                                                \
        const i32x8 cur = w6[(J) & 3];                                                                              \
        if ((J) + 4 < 2 * NT) w6[(J) & 3] = mx_ld6<NT>(L, (J) + 4, lane);                                           \
        MX_L0_HOOK(NT, J)                                                                                           \
        if ((J) < NT) acc[t6] = MX_MFMA6(cur, b.l6, acc[t6], t6 & 3, scw[t6 >> 2], b.sl);                            \
        else          acc[t6] = MX_MFMA6(cur, b.h6, acc[t6], t6 & 3, scl[t6 >> 2], b.sh);                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }

// what the layer-0 hook needs: resident W0 operands, biases, the raw-input operands, which pair of
// hidden tiles to produce, and where to leave them
struct MxL0 {
    const char *W0;
    const float *sb0;
    half8 xhi, xlo;
    int cn, h;
    f32x16 h0n[2];
};

// MFMA phase of a chunk with NT output tiles, K-major: for each of the four f16 k-steps all tiles,
// then fp6(W) x fp6(h_lo) for all tiles, then fp6(W_lo) x fp6(h).  Consecutive MFMAs hit different
// accumulators; every A operand is requested 8 (f16) / 4 (fp6) MFMAs ahead of its use through a
// rotating register window, so only ~32 operand registers are live beside the accumulators.
template <int NT, bool L0H>
__device__ __forceinline__ void mx_m_phase(const char *__restrict__ L, int lane, f32x16 (&acc)[NT], const MxB &b, MxL0 &l0)
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
    half8 l0a_hi, l0a_lo;
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
// one barrier interval of layer 1: chunk k is multiplied while chunk k+1 lands in the other buffer.
// The two waves that share a SIMD (w and w+4) run the same stream with the barrier at different
// points of it: "early" waves (0-3) build B(k) from l0.h0n and then multiply, "late" waves (4-7)
// arrive with B(k) built and build B(k+1) after multiplying - so while one of them is in its VALU
// phase the other one has the matrix pipe to itself.
__device__ __forceinline__ void mx_l1_step(const char *__restrict__ cur, char *__restrict__ nxt, const char *image, int k,
                                           f32x16 (&acc1)[8], MxL0 &l0, int lane, int wave, bool late, MxB &b, unsigned long long *tr)
{
    const char *src = image + (size_t)mx_chunk_offset(k + 1) * 1024;
    const int units = mx_chunk_units(k + 1), split = min(kDmaFirst, units);
    if (!late) mx_make_b<true>(l0.h0n[0], l0.h0n[1], b, MxDma{src, nxt, 0, split, wave & 3, lane * 16});
    l0.cn = min(k + 1, 7);                        // k == 7: harmless repeat of the last pair
    MX_STAMP(0)
    mx_m_phase<8, true>(cur, lane, acc1, b, l0);
    MX_STAMP(1)
    if (late) {
        const MxDma dma = {src, nxt, split, units, wave & 3, lane * 16};
        if (k < 7) mx_make_b<true>(l0.h0n[0], l0.h0n[1], b, dma);
        else mx_make_b<true>(acc1[0], acc1[1], b, dma);
    }
}

template <int Q>
__device__ __forceinline__ void mx_l2_step(const char *__restrict__ cur, char *__restrict__ nxt, const char *image, f32x16 (&acc1)[8],
                                           f32x16 (&acc2)[4], MxL0 &l0, int lane, int wave, bool late, MxB &b)
{
    // next chunk: 9 + Q of this tile, or chunk 0 of the next tile
    constexpr int kNext = Q < 3 ? 9 + Q : 0;
    const char *src = image + (size_t)mx_chunk_offset(kNext) * 1024;
    constexpr int units = mx_chunk_units(kNext);
    // the M-first waves have no VALU phase after the last chunk: there the V-first waves issue everything
    constexpr int split = Q < 3 ? (kDmaFirst < units ? kDmaFirst : units) : units;
    if (!late) mx_make_b<true>(acc1[2 * Q], acc1[2 * Q + 1], b, MxDma{src, nxt, 0, split, wave & 3, lane * 16});
    mx_m_phase<4, false>(cur, lane, acc2, b, l0);
    if (late && Q < 3) mx_make_b<true>(acc1[Q < 3 ? 2 * Q + 2 : 0], acc1[Q < 3 ? 2 * Q + 3 : 1], b, MxDma{src, nxt, split, units, wave & 3, lane * 16});
}

__device__ __forceinline__ void mx_load_row(const float *X, int64_t pi, int h, int c0, float (&xr)[8])
{
    const float4 *q = reinterpret_cast<const float4 *>(X + pi * kXRow + 8 * h);
    const float4 a = q[0], c = q[1];
    xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = c.x; xr[5] = c.y; xr[6] = c.z; xr[7] = c.w;
#pragma unroll
    for (int s = 0; s < 8; ++s) xr[s] = (s + 8 * h < c0) ? xr[s] : 0.0f;
}

template <bool MASK>
__global__ __launch_bounds__(kMxBlock, 2) void k_mlp_mx6(const float *__restrict__ X, int64_t N, float *__restrict__ out, MlpMx6Dev w,
                                                         int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const bool late = wave < 4;                   // waves w and w+4 share a SIMD; the older one (arbitration winner) multiplies first

    mx_issue_units(w.image, smem + kMxW0Off, kMxW0Bytes / 1024, wave, lane);     // resident layer-0 operands
    float *side = reinterpret_cast<float *>(smem + kMxSideOff);
    for (int i = threadIdx.x; i < kMxSideFloats; i += kMxBlock) side[i] = w.side[i];
    const float *sb0 = side, *sb1 = side + 512, *sb2 = side + 768, *sw3 = side + 896;

    MxL0 l0;
    l0.W0 = smem + kMxW0Off; l0.sb0 = sb0; l0.h = h; l0.cn = 0;
    int tile = blockIdx.x;
    {
        const int64_t base = ((int64_t)tile * (kMxBlock / 64) + wave) * 32;
        float xr[8];
        mx_load_row(X, min(base + j, N - 1), h, w.c0, xr);
        mx_split8(xr, l0.xhi, l0.xlo);
    }
    mx_issue_chunk(w.image, smem, 0, wave, lane);
    mx_dma_wait();
    __syncthreads();   // side arrays visible, W0 + chunk 0 landed

    int iter = 0;
    for (; tile < ntiles; tile += gridDim.x, ++iter) {
        unsigned long long *tr = (w.trace && blockIdx.x == 0 && iter == 1 && lane == 0) ? w.trace + wave * 64 : nullptr;
        MX_STAMP(0)
        const int64_t base = ((int64_t)tile * (kMxBlock / 64) + wave) * 32;
        const int64_t pi = min(base + j, N - 1);      // waves past the end still help with the DMA + barriers
        f32x16 acc1[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc1[m] = mx_ld16(sb1 + (m * 2 + h) * 16);

        // ---- layers 0 + 1: one chunk = 64 hidden channels ---------------------------------------------
        MxB b;
        l0.h0n[0] = mx_l0_tile(l0.W0, sb0, 0, l0.xhi, l0.xlo, h, lane);
        l0.h0n[1] = mx_l0_tile(l0.W0, sb0, 1, l0.xhi, l0.xlo, h, lane);
        if (late) mx_make_b<false>(l0.h0n[0], l0.h0n[1], b, MxDma{nullptr, nullptr, 0, 0, 0, 0});
        else { b.hi = (half32)(_Float16)0; b.h6 = i32x8{0, 0, 0, 0, 0, 0, 0, 0}; b.l6 = b.h6; b.sh = 127; b.sl = 127; }
        MX_STAMP(1)
        for (int k = 0; k < 8; ++k) {
            mx_l1_step(smem + (k & 1) * kMxBuf, smem + ((k + 1) & 1) * kMxBuf, w.image, k, acc1, l0, lane, wave, late, b, tr ? tr + 2 + 4 * k : nullptr);
            mx_dma_wait();
            MX_STAMP(2 + 4 * k + 2)
            __syncthreads();   // all waves done with this buffer AND the next chunk has landed
            MX_STAMP(2 + 4 * k + 3)
        }

        // ---- layer 2: K = 256 (registers) + 16 (raw input) ---------------------------------------------
        f32x16 acc2[4];
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = mx_ld16(sb2 + (m2 * 2 + h) * 16);
        mx_l2_step<0>(smem, smem + kMxBuf, w.image, acc1, acc2, l0, lane, wave, late, b);
        mx_dma_wait();
        __syncthreads();
        MX_STAMP(34)
        mx_l2_step<1>(smem + kMxBuf, smem, w.image, acc1, acc2, l0, lane, wave, late, b);
        mx_dma_wait();
        __syncthreads();
        MX_STAMP(35)
        mx_l2_step<2>(smem, smem + kMxBuf, w.image, acc1, acc2, l0, lane, wave, late, b);
        mx_dma_wait();
        __syncthreads();
        MX_STAMP(36)
        // last step: this tile's input row again (raw-input k-step, layer 3) and the next tile's
        // (chunk 0 of the next tile goes into the free buffer from inside the last MFMA phase)
        float xr[8], xn[8];
        float maskf = 1.0f;
        {
            const float *Xp = X;
            asm volatile("" : "+s"(Xp));      // fresh loads, not values kept alive across the layers
            mx_load_row(Xp, pi, h, w.c0, xr);
            const int64_t nbase = ((int64_t)(tile + gridDim.x) * (kMxBlock / 64) + wave) * 32;
            mx_load_row(Xp, min(nbase + j, N - 1), h, w.c0, xn);
            if (MASK) {
                const uint32_t code = (uint32_t)__float_as_int(Xp[pi * kXRow + kCodeSlot]);
                maskf = (code & kCodeInCube) ? 1.0f : 0.0f;
            }
        }
        mx_l2_step<3>(smem + kMxBuf, smem, w.image, acc1, acc2, l0, lane, wave, late, b);
        MX_STAMP(37)

        // raw-input k-step of layer 2 (3 x f16) and layer 3 on the VALU
        {
            const char *L = smem + kMxBuf + MxLay<4>::RAW;
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) {
                const half8 a_hi = mx_op(L, 2 * m2, lane), a_lo = mx_op(L, 2 * m2 + 1, lane);
                acc2[m2] = MX_MFMA16(a_hi, l0.xhi, acc2[m2]);
                acc2[m2] = MX_MFMA16(a_hi, l0.xlo, acc2[m2]);
                acc2[m2] = MX_MFMA16(a_lo, l0.xhi, acc2[m2]);
            }
        }
        const float *w3 = sw3 + h * 72;
        float part = 0.0f;
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) {
            const f32x16 wv = mx_ld16(w3 + m2 * 16);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float x = acc2[m2][t];
                part = fmaf(wv[t], fmaxf(x, 0.01f * x), part);
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) part = fmaf(w3[64 + s], xr[s], part);
        const float other = __shfl_xor(part, 32);
        const float y = apply_last_op((part + other) + w.b3, w.last_op);
        if (h == 0 && base + j < N) {
            const bool in_cube = MASK ? maskf != 0.0f : true;          // select, not multiply: a masked point is 0 whatever the network said
            if (in_cube && not_finite(y)) *w.flag = 1;
            out[base + j] = in_cube ? y : 0.0f;
        }

        mx_split8(xn, l0.xhi, l0.xlo);
        MX_STAMP(38)
        mx_dma_wait();
        MX_STAMP(39)
        __syncthreads();   // chunk 0 of the next tile landed; everyone is done with the raw-input operands
        MX_STAMP(40)
    }
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
    // power-of-two weight scales: 1 unless a layer's weights are tiny (f16 would lose their low bits).
    // Accumulators of layer l carry s_l; LeakyReLU is positively homogeneous, so the next layer's
    // weights are packed as W * s_{l+1} / s_l and nothing is rescaled at run time.
    auto lift = [](const std::vector<float> &Wl) {
        float mx = 0.f;
        for (float v : Wl) mx = std::max(mx, std::fabs(v));
        if (!(mx > 0.f) || !std::isfinite(mx) || mx >= 0.015625f) return 1.0f;
        int ex; (void)std::frexp(mx, &ex);
        return std::ldexp(1.0f, std::min(1 - ex, 24));        // brings max|W| into [1, 2)
    };
    const float s0 = lift(W[0]), s1 = lift(W[1]), s2 = lift(W[2]);
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
    parallel_for(12, [&](int k) {          // chunks write disjoint parts of the image
        if (k < 8) pack_chunk(k, 8, W[1], 512, 64 * k, s1 / s0);
        else pack_chunk(k, 4, W[2], ci2, 64 * (k - 8), s2 / s1);
    });

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
            for (int t = 0; t < 16; ++t) w3[h * 72 + m2 * 16 + t] = W[3][32 * m2 + rho(t, h)] / s2;
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
    w.b3 = mlp->b3; w.c0 = mlp->c0; w.last_op = mlp->last_op; w.trace = nullptr;
    w.flag = reinterpret_cast<int *>(mlp->d_blob + mlp->off_flag);
    { const int rc = mlp_flag_reset(mlp, st); if (rc) return rc; }
    static const bool want_trace = getenv("ICON_AMD_MX6_TRACE") != nullptr;
    if (want_trace) { ICON_HIP(hipMalloc((void **)&w.trace, 8 * 64 * 8)); ICON_HIP(hipMemsetAsync(w.trace, 0, 8 * 64 * 8, st)); }
    const int64_t nt = (N + kMxPts - 1) / kMxPts;
    ICON_ARG(nt < (1ll << 31), "mlp: N too large for one launch");
    int n_cu = 0;
    { const int rc = device_cu_count(&n_cu); if (rc) return rc; }
    {
        const int rc = once_per_device(7, [] {   // per device: a process may drive several (common.h)
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_mx6<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMxLds);
            return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_mx6<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMxLds);
        });
        if (rc) return rc;
    }
    const unsigned nb = (unsigned)std::min<int64_t>(nt, n_cu);      // one persistent workgroup per CU (LDS-limited)
    if (mask) hipLaunchKernelGGL(k_mlp_mx6<true>, dim3(nb), dim3(kMxBlock), kMxLds, st, d_x, N, d_out, w, (int)nt);
    else      hipLaunchKernelGGL(k_mlp_mx6<false>, dim3(nb), dim3(kMxBlock), kMxLds, st, d_x, N, d_out, w, (int)nt);
    if (want_trace) {
        unsigned long long t[8 * 64];
        ICON_HIP(hipStreamSynchronize(st));
        ICON_HIP(hipMemcpy(t, w.trace, sizeof(t), hipMemcpyDeviceToHost));
        (void)hipFree(w.trace);
        for (int wv = 0; wv < 8; wv += 4) {
            const unsigned long long *r = t + wv * 64;
            fprintf(stderr, "mx6 trace wave %d (cycles): V0 %llu |", wv, r[1] - r[0]);
            for (int k = 0; k < 8; ++k)
                fprintf(stderr, " k%d issue %llu M %llu V %llu wait %llu bar %llu |", k, r[2 + 4 * k] - (k ? r[2 + 4 * k - 1] : r[1]),
                        r[3 + 4 * k] - r[2 + 4 * k], r[4 + 4 * k] - r[3 + 4 * k], 0ull, r[5 + 4 * k] - r[4 + 4 * k]);
            fprintf(stderr, " L2 steps %llu %llu %llu %llu | epilogue %llu dma-wait %llu barrier %llu | tile total %llu\n", r[34] - r[33], r[35] - r[34],
                    r[36] - r[35], r[37] - r[36], r[38] - r[37], r[39] - r[38], r[40] - r[39], r[40] - r[0]);
        }
    }
    ICON_HIP(hipGetLastError());
    return mlp_rescue_rows(mlp, d_x, N, d_out, st);
}

}  // namespace icon
