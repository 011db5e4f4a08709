// query_kernels.hip - gfx950 kernels for the per-point geometry/feature half of HGPIFuNet.query:
//   nearest triangle (BVH2, per-lane LDS stack, or LDS-tiled brute force), +x ray parity over
//   (y,z) bins, Heidrich barycentric interpolation, outlier clipping, bilinear / trilinear
//   feature gather + front/back select, and assembly of the 16-float MLP input rows.
//
// Reference being replaced (paths relative to the reference root):
//   lib/dataset/mesh_util.py:357-396 (cal_sdf_batch), :319-354 (barycentrics), :266-277 (feat_select)
//   lib/net/HGPIFuNet.py:268-365 (query), lib/net/geometry.py:21-61 (index, orthogonal)
//
// Arithmetic spec: float32, the only fused operations are the explicit fmaf() calls,
// IEEE division / sqrt (hipcc default: correctly rounded).  This file is compiled with
// -ffp-contract=off; the pragma below is a second line of defence.
#pragma clang fp contract(off)

#include "common.h"

#include <cstdlib>
#include <cstring>

namespace icon {

// ---------------------------------------------------------------------------------------------
// small vector helpers
// ---------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// S2: exact point-triangle squared distance, "face or nearest edge" form on per-triangle constants
// prepared by the host (TriPre).  Same operations on the same operands as the checker's
// orc_tri_dist2, hence bit-identical results; no division, no branches:
//   (s,t) = barycentrics of the plane projection; inside -> |p - (a + s ab + t ac)|^2,
//   else min over the three segments of |p - (origin + clamp(t,0,1) * edge)|^2.
struct TriC {   // TriPre fields as values (SGPRs when read through the constant address space)
    f3 a, b, ab, ac, bc;
    float i00, i11, ibc, a00, a01, a11, inn;
};

__device__ __forceinline__ float seg_dist2(f3 p, f3 o, f3 e, float dot_e_po, float inv_len2)
{
    const float t = fminf(fmaxf(dot_e_po * inv_len2, 0.0f), 1.0f);
    const f3 q = mk3(fmaf(e.x, t, o.x), fmaf(e.y, t, o.y), fmaf(e.z, t, o.z));
    const f3 d = sub3(p, q);
    return dot3(d, d);
}

__device__ __forceinline__ float tri_dist2(f3 p, const TriC &t)
{
    const f3 ap = sub3(p, t.a), bp = sub3(p, t.b);
    const float d1 = dot3(t.ab, ap), d2 = dot3(t.ac, ap), d3 = dot3(t.bc, bp);
    const float s = fmaf(t.a11, d1, -(t.a01 * d2)) * t.inn;
    const float u = fmaf(t.a00, d2, -(t.a01 * d1)) * t.inn;
    const bool inside = (s >= 0.0f) & (u >= 0.0f) & (s + u <= 1.0f);
    const f3 q = mk3(fmaf(t.ac.x, u, fmaf(t.ab.x, s, t.a.x)), fmaf(t.ac.y, u, fmaf(t.ab.y, s, t.a.y)),
                     fmaf(t.ac.z, u, fmaf(t.ab.z, s, t.a.z)));
    const f3 df = sub3(p, q);
    float d_face = dot3(df, df);
    const float e0 = seg_dist2(p, t.a, t.ab, d1, t.i00);
    const float e1 = seg_dist2(p, t.a, t.ac, d2, t.i11);
    const float e2 = seg_dist2(p, t.b, t.bc, d3, t.ibc);
    float d_edge = fminf(fminf(e0, e1), e2);
    asm volatile("" : "+v"(d_face), "+v"(d_edge));    // keep the final choice a v_cndmask
    return inside ? d_face : d_edge;
}

template <class Ptr>
__device__ __forceinline__ TriC load_tric(Ptr q)   // q: 24 dwords of a TriPre
{
    TriC t;
    t.a = mk3(q[0], q[1], q[2]); t.b = mk3(q[3], q[4], q[5]); t.ab = mk3(q[6], q[7], q[8]);
    t.ac = mk3(q[9], q[10], q[11]); t.bc = mk3(q[12], q[13], q[14]);
    t.i00 = q[15]; t.i11 = q[16]; t.ibc = q[17]; t.a00 = q[18]; t.a01 = q[19]; t.a11 = q[20]; t.inn = q[21];
    return t;
}

// ---- two triangles per instruction: the same test on packed f32 (v_pk_add / v_pk_mul / v_pk_fma) ----
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const f2 cf2;
struct f3x2 { f2 x, y, z; };
__device__ __forceinline__ f2 bc2(float v) { f2 r; r.x = v; r.y = v; return r; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f3x2 sub3x2(f3x2 a, f3x2 b) { f3x2 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
__device__ __forceinline__ f2 dot3x2(f3x2 a, f3x2 b) { return fma2(a.z, b.z, fma2(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f2 clamp01x2(f2 v)
{
    f2 r; r.x = fminf(fmaxf(v.x, 0.0f), 1.0f); r.y = fminf(fmaxf(v.y, 0.0f), 1.0f); return r;
}
__device__ __forceinline__ f2 seg_dist2x2(f3x2 p, f3x2 o, f3x2 e, f2 dot_e_po, f2 inv_len2)
{
    const f2 t = clamp01x2(dot_e_po * inv_len2);
    f3x2 q; q.x = fma2(e.x, t, o.x); q.y = fma2(e.y, t, o.y); q.z = fma2(e.z, t, o.z);
    const f3x2 d = sub3x2(p, q);
    return dot3x2(d, d);
}

// q: 24 f2 fields of one leaf pair (constant address space -> SGPR pairs)
__device__ __forceinline__ f2 tri_dist2_pair(f3 p1, cf2 *q)
{
    f3x2 p; p.x = bc2(p1.x); p.y = bc2(p1.y); p.z = bc2(p1.z);
    f3x2 a, b, ab, ac, bc;
    a.x = q[0]; a.y = q[1]; a.z = q[2]; b.x = q[3]; b.y = q[4]; b.z = q[5];
    ab.x = q[6]; ab.y = q[7]; ab.z = q[8]; ac.x = q[9]; ac.y = q[10]; ac.z = q[11]; bc.x = q[12]; bc.y = q[13]; bc.z = q[14];
    const f2 i00 = q[15], i11 = q[16], ibc = q[17], a00 = q[18], a01 = q[19], a11 = q[20], inn = q[21];
    const f3x2 ap = sub3x2(p, a), bp = sub3x2(p, b);
    const f2 d1 = dot3x2(ab, ap), d2 = dot3x2(ac, ap), d3 = dot3x2(bc, bp);
    const f2 s = fma2(a11, d1, -(a01 * d2)) * inn;
    const f2 u = fma2(a00, d2, -(a01 * d1)) * inn;
    const f2 su = s + u;
    f3x2 qf; qf.x = fma2(ac.x, u, fma2(ab.x, s, a.x)); qf.y = fma2(ac.y, u, fma2(ab.y, s, a.y)); qf.z = fma2(ac.z, u, fma2(ab.z, s, a.z));
    const f3x2 df = sub3x2(p, qf);
    f2 d_face = dot3x2(df, df);
    const f2 e0 = seg_dist2x2(p, a, ab, d1, i00);
    const f2 e1 = seg_dist2x2(p, a, ac, d2, i11);
    const f2 e2 = seg_dist2x2(p, b, bc, d3, ibc);
    f2 d_edge; d_edge.x = fminf(fminf(e0.x, e1.x), e2.x); d_edge.y = fminf(fminf(e0.y, e1.y), e2.y);
    const bool in0 = (s.x >= 0.0f) & (u.x >= 0.0f) & (su.x <= 1.0f);
    const bool in1 = (s.y >= 0.0f) & (u.y >= 0.0f) & (su.y <= 1.0f);
    float f0 = d_face.x, f1 = d_face.y, g0 = d_edge.x, g1 = d_edge.y;
    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(g0), "+v"(g1));
    f2 r; r.x = in0 ? f0 : g0; r.y = in1 ? f1 : g1;
    return r;
}

// S4: +x ray / triangle crossing with the canonical (lower vertex id first) edge rule.
__device__ __forceinline__ float edge_fn(float yi, float zi, float yj, float zj, float qy, float qz)
{
    // twice the signed area of (q, vi, vj), relative to q: exactly 0 when q projects onto an end point
    const float t1 = (yi - qy) * (zj - qz);
    return fmaf(-(zi - qz), (yj - qy), t1);
}
__device__ __forceinline__ bool edge_side(float yi, float zi, float yj, float zj, float e)
{
    // exact zeros: symbolic perturbation q -> q + (eps, eps^2), identical for every edge
    if (e > 0.0f) return true;
    if (e < 0.0f) return false;
    const float dz = zj - zi, dy = yj - yi;
    if (dz != 0.0f) return dz < 0.0f;
    return dy > 0.0f;
}
__device__ __forceinline__ void oriented_edge(int ia, f3 a, int ib, f3 b, float qy, float qz, float &val, bool &pos)
{
    if (ia < ib) { const float e = edge_fn(a.y, a.z, b.y, b.z, qy, qz); val = e; pos = edge_side(a.y, a.z, b.y, b.z, e); }
    else         { const float e = edge_fn(b.y, b.z, a.y, a.z, qy, qz); val = -e; pos = !edge_side(b.y, b.z, a.y, a.z, e); }
}
__device__ __forceinline__ int ray_hit(f3 p, f3 a, f3 b, f3 c, int ia, int ib, int ic)
{
    float e_ab, e_bc, e_ca; bool s_ab, s_bc, s_ca;
    oriented_edge(ia, a, ib, b, p.y, p.z, e_ab, s_ab);
    oriented_edge(ib, b, ic, c, p.y, p.z, e_bc, s_bc);
    oriented_edge(ic, c, ia, a, p.y, p.z, e_ca, s_ca);
    if (!(s_ab == s_bc && s_bc == s_ca)) return 0;
    const float num = fmaf(e_ab, c.x - p.x, fmaf(e_ca, b.x - p.x, e_bc * (a.x - p.x)));
    return s_ab ? (num > 0.0f) : (num < 0.0f);
}

// Wave-uniform reads of the (read-only) BVH and triangle arrays go through the CONSTANT address
// space so the compiler emits scalar loads (s_load_dwordx8/x16 into SGPRs, served by the scalar
// cache) instead of 64 identical vector loads.
typedef __attribute__((address_space(4))) const float cfloat;
__device__ __forceinline__ cfloat *as_const(const void *p) { return (cfloat *)(uintptr_t)p; }

__device__ __forceinline__ void load_tri_pos_uniform(const TriRec *t, f3 &a, f3 &b, f3 &c)
{
    cfloat *q = as_const(t);
    a = mk3(q[0], q[1], q[2]); b = mk3(q[3], q[4], q[5]); c = mk3(q[6], q[7], q[8]);
}

__device__ __forceinline__ void load_tri_pos(const TriRec *t, f3 &a, f3 &b, f3 &c)
{
    const float4 *q = reinterpret_cast<const float4 *>(t);
    const float4 q0 = q[0], q1 = q[1];
    const float q2 = reinterpret_cast<const float *>(t)[8];
    a = mk3(q0.x, q0.y, q0.z); b = mk3(q0.w, q1.x, q1.y); c = mk3(q1.z, q1.w, q2);
}
__device__ __forceinline__ void load_tri_full(const TriRec *t, f3 &a, f3 &b, f3 &c, int &ia, int &ib, int &ic)
{
    const float4 *q = reinterpret_cast<const float4 *>(t);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2];
    a = mk3(q0.x, q0.y, q0.z); b = mk3(q0.w, q1.x, q1.y); c = mk3(q1.z, q1.w, q2.x);
    ia = __float_as_int(q2.y); ib = __float_as_int(q2.z); ic = __float_as_int(q2.w);
}

__device__ __forceinline__ float box_dist2(float lx, float ly, float lz, float hx, float hy, float hz, f3 p)
{
    const float dx = fmaxf(fmaxf(lx - p.x, p.x - hx), 0.0f);
    const float dy = fmaxf(fmaxf(ly - p.y, p.y - hy), 0.0f);
    const float dz = fmaxf(fmaxf(lz - p.z, p.z - hz), 0.0f);
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// both children of a node at once (packed f32): squared distance from p to each child's box
__device__ __forceinline__ f2 box_dist2_pair(cf2 *q, f3 p)
{
    const f2 px = bc2(p.x), py = bc2(p.y), pz = bc2(p.z);
    const f2 ax = q[0] - px, bx = px - q[3];
    const f2 ay = q[1] - py, by = py - q[4];
    const f2 az = q[2] - pz, bz = pz - q[5];
    f2 dx, dy, dz;
    dx.x = fmaxf(fmaxf(ax.x, bx.x), 0.0f); dx.y = fmaxf(fmaxf(ax.y, bx.y), 0.0f);
    dy.x = fmaxf(fmaxf(ay.x, by.x), 0.0f); dy.y = fmaxf(fmaxf(ay.y, by.y), 0.0f);
    dz.x = fmaxf(fmaxf(az.x, bz.x), 0.0f); dz.y = fmaxf(fmaxf(az.y, bz.y), 0.0f);
    return fma2(dz, dz, fma2(dy, dy, dx * dx));
}

struct Nearest { float d2; int slot; int face; };

// Pruning bound: a subtree may be skipped only if no triangle in it can tie or beat `best`.
// Computed distances carry < 1e-6 absolute error (coordinates are O(1)), so the bound is
// (sqrt(best) + 4e-6)^2 with a relative cushion; see DESIGN.md "BVH conservativeness".
__device__ __forceinline__ float prune_threshold(float best)
{
    const float s = sqrtf(best) + 4e-6f;
    return s * s * 1.000001f;
}

// BVH2 PACKET traversal: the 64 lanes of a wavefront descend the tree TOGETHER.  Control flow, the
// node / leaf addresses and the stack are wave-uniform (scalar registers, scalar loads, one LDS
// word per stack entry per wave); every lane tests its own point against the shared node boxes
// and triangles, and a subtree is entered when ANY lane still needs it.  In lattice mode a
// wavefront owns a 4x4x4 block of lattice points, so the lanes' candidate sets nearly coincide.
// A leaf holds up to 4 TriPre records (96 B each, scalar loads); the distance test is branch-free.  Near child first, ordered by the block's centre lane.
// `live` = false parks a padding lane: it never votes and its result is discarded.
template <bool STATS = false>
__device__ __forceinline__ Nearest nearest_packet(const MeshDev &m, f3 p, bool live, int *wstack /* LDS, kStackDepth ints of this wave */,
                                                  int *n_nodes = nullptr, int *n_tris = nullptr, float thr0 = INFINITY)
{
    Nearest nr; nr.d2 = INFINITY; nr.slot = 0; nr.face = 0x7fffffff;
    unsigned long long key = 0x7f8000007fffffffull;   // (+inf, INT_MAX)
    int slot = 0;
    float thr = live ? thr0 : -INFINITY;
    int sp = 0;
    int cur = 0;
    while (true) {
        if (cur < 0) {
            const int code = ~cur;
            const int leaf = code >> 2, cnt = (code & 3) + 1;
            if (STATS) *n_tris += cnt;
            const unsigned long long before = key;
            for (int pr = 0; pr * 2 < cnt; ++pr) {
                cf2 *q = reinterpret_cast<cf2 *>(as_const(&m.leaves[leaf].pair[pr]));
                const f2 d2 = tri_dist2_pair(p, q);
                const f2 fc = q[22];
                // S3 as ONE unsigned 64-bit minimum: key = (bits of d^2) << 32 | face.  d^2 >= +0, so
                // its bit pattern orders like the value; equal d^2 -> lower face id wins; NaN (bits
                // above +inf) never wins; a padding copy has the same key as its original.
                const unsigned long long k0 = ((unsigned long long)(unsigned)__float_as_int(d2.x) << 32) | (unsigned)__float_as_int(fc.x);
                const unsigned long long k1 = ((unsigned long long)(unsigned)__float_as_int(d2.y) << 32) | (unsigned)__float_as_int(fc.y);
                const bool u0 = live & (k0 < key);
                key = u0 ? k0 : key; slot = u0 ? leaf * kLeafMax + 2 * pr : slot;
                const bool u1 = live & (k1 < key);
                key = u1 ? k1 : key; slot = u1 ? leaf * kLeafMax + 2 * pr + 1 : slot;
            }
            const bool improved = key != before;
            nr.d2 = __int_as_float((int)(key >> 32));
            if (__any(improved)) thr = live ? prune_threshold(nr.d2) : thr;
            if (sp == 0) break;
            cur = __builtin_amdgcn_readfirstlane(wstack[--sp]);
        } else {
            if (STATS) ++*n_nodes;
            cf2 *q = reinterpret_cast<cf2 *>(as_const(m.nodes + cur));   // lo.x lo.y lo.z hi.x hi.y hi.z (children 0,1), ids
            const f2 dd = box_dist2_pair(q, p);
            const float d0 = dd.x, d1 = dd.y;
            const f2 ids = q[6];
            const int c0 = __float_as_int(ids.x), c1 = __float_as_int(ids.y);
            const bool v0 = __any(d0 <= thr), v1 = __any(d1 <= thr);
            if (v0 && v1) {
                // order by the block's centre lane; non-negative floats order like their bit patterns
                const int e0 = __builtin_amdgcn_readlane(__float_as_int(d0), 21);
                const int e1 = __builtin_amdgcn_readlane(__float_as_int(d1), 21);
                const bool first0 = e0 <= e1;
                wstack[sp++] = first0 ? c1 : c0;
                cur = first0 ? c0 : c1;
            } else if (v0) cur = c0;
            else if (v1) cur = c1;
            else {
                if (sp == 0) break;
                cur = __builtin_amdgcn_readfirstlane(wstack[--sp]);
            }
        }
    }
    nr.d2 = __int_as_float((int)(key >> 32)); nr.slot = slot; nr.face = (int)(key & 0xffffffffu);
    return nr;
}

// BVH2 traversal, ONE WAVEFRONT PER POINT: the 64 lanes expand 64 tree nodes / test 64 triangles of the
// SAME query point per round.  A point far from the surface has hundreds of leaves whose boxes are
// closer than its nearest triangle; walked one after the other by a single lane that is a chain of
// ~500 dependent loads (0.5 ms for a single wave), here it is ~10-20 rounds: 33 us for 64 points.
// Throughput is ~9 ns per point (the lattice packets amortise to 0.25 ns), so this is the search for
// small batches only - see kPacketMinPoints.
//   1. greedy descent (near child first) to one leaf -> initial bound
//   2. LIFO frontier of node references in LDS: a round pops up to 64 entries, every lane tests the
//      two child boxes of its node against the current bound and pushes the survivors (inner nodes
//      back on the frontier, leaves with their box distance on a leaf list); whenever the leaf
//      list holds 16 leaves (64 triangle slots), or the frontier is empty, a leaf round runs: one
//      lane per triangle, the wave minimum of the (d^2, face) keys goes through one LDS atomic.
// Same S2 distance, same key, same pruning bound as the other traversals -> same results.
// point mode: batches below this size go one wavefront per point, larger ones through Morton-ordered packets
// (measured crossover on MI355X, tools/time_query_points.py: 60k points 0.42 vs 0.67 ms, 200k ~1.4 vs 0.70 ms)
constexpr int64_t kPacketMinPoints = 98304;
constexpr int kCoopWaves = 4;                  // wavefronts (= points) per workgroup
constexpr int kCoopLeaves = 256;               // leaf list (ref, box distance)
// per-wave LDS: [frontier: cap ints][leaf refs][leaf box distances][best key][best slot].  cap >= 64 * (tree
// depth + 2) is the LIFO bound of a 64-wide expansion (a round pops the 64 deepest entries and pushes at
// most 128 one level deeper); the SMPL tree (depth ~20) needs 6 KiB per wave, so ~5 waves per SIMD fit.
struct CoopLds {
    int *frontier; int cap;
    int *leaf_ref; float *leaf_d;
    unsigned long long *best; int *best_slot;
};
__host__ __device__ inline int coop_cap(int depth) { return 64 * (depth + 3); }
__host__ __device__ inline size_t coop_wave_bytes(int cap) { return (size_t)cap * 4 + kCoopLeaves * 8 + 16; }
__device__ __forceinline__ CoopLds coop_lds(char *smem, int wave, int cap)
{
    char *b = smem + (size_t)wave * coop_wave_bytes(cap);
    CoopLds S;
    S.best = reinterpret_cast<unsigned long long *>(b);
    S.best_slot = reinterpret_cast<int *>(b + 8);
    S.leaf_ref = reinterpret_cast<int *>(b + 16);
    S.leaf_d = reinterpret_cast<float *>(b + 16 + kCoopLeaves * 4);
    S.frontier = reinterpret_cast<int *>(b + 16 + kCoopLeaves * 8);
    S.cap = cap;
    return S;
}

__device__ __forceinline__ Nearest nearest_coop(const MeshDev &m, f3 p, const CoopLds &S)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned long long key = 0x7f8000007fffffffull;
    int slot = 0;
    // test triangle (leaf, t) if t < cnt; returns key (or +inf key)
    auto tri_key = [&](int leaf, int t, int cnt, int &out_slot) -> unsigned long long {
        if (t >= cnt) return 0x7f8000007fffffffull;
        const float *q = reinterpret_cast<const float *>(m.leaves + leaf) + (t >> 1) * 48 + (t & 1);
        TriC tc;
        tc.a = mk3(q[0], q[2], q[4]); tc.b = mk3(q[6], q[8], q[10]); tc.ab = mk3(q[12], q[14], q[16]);
        tc.ac = mk3(q[18], q[20], q[22]); tc.bc = mk3(q[24], q[26], q[28]);
        tc.i00 = q[30]; tc.i11 = q[32]; tc.ibc = q[34]; tc.a00 = q[36]; tc.a01 = q[38]; tc.a11 = q[40]; tc.inn = q[42];
        const float d2 = tri_dist2(p, tc);
        out_slot = leaf * kLeafMax + t;
        return ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(q[44]);
    };
    // wave minimum of (key, slot) -> uniform
    auto wave_min = [&](unsigned long long k, int sl) {
        if (lane == 0) { *reinterpret_cast<volatile unsigned long long *>(S.best) = key; *reinterpret_cast<volatile int *>(S.best_slot) = slot; }
        __builtin_amdgcn_wave_barrier();
        if (k < key) atomicMin(S.best, k);
        __builtin_amdgcn_wave_barrier();
        const unsigned long long b = *reinterpret_cast<volatile unsigned long long *>(S.best);
        if (k == b && b != key) *reinterpret_cast<volatile int *>(S.best_slot) = sl;        // any lane holding the minimum (same face -> same slot, or a padding copy)
        __builtin_amdgcn_wave_barrier();
        if (b != key) { key = b; slot = *reinterpret_cast<volatile int *>(S.best_slot); }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- 1. greedy descent ---------------------------------------------------------------------------
    int cur = 0;
    while (cur >= 0) {
        const float4 *q4 = reinterpret_cast<const float4 *>(m.nodes + cur);
        const float4 n0 = q4[0], n1 = q4[1], n2 = q4[2];
        const float2 ids = *reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(m.nodes + cur) + 12);
        const float d0 = box_dist2(n0.x, n0.z, n1.x, n1.z, n2.x, n2.z, p);
        const float d1 = box_dist2(n0.y, n0.w, n1.y, n1.w, n2.y, n2.w, p);
        cur = __builtin_amdgcn_readfirstlane(d0 <= d1 ? __float_as_int(ids.x) : __float_as_int(ids.y));
    }
    {
        const int code = ~cur, leaf = code >> 2, cnt = (code & 3) + 1;
        int sl = 0;
        const unsigned long long k = tri_key(leaf, lane & 3, cnt, sl);
        wave_min(k, sl);
    }
    float thr = prune_threshold(__int_as_float((int)(key >> 32)));

    // ---- 2. frontier -----------------------------------------------------------------------------------
    int nf = 1, nl = 0;                           // uniform counters
    if (lane == 0) S.frontier[0] = 0;
    __builtin_amdgcn_wave_barrier();
    while (nf > 0 || nl > 0) {
        if (nl >= 16 || nf == 0) {
            // leaf round: up to 16 leaves from the end of the list, one lane per triangle slot
            const int take = min(nl, 16);
            const int e = nl - take + (lane >> 2);
            unsigned long long k = 0x7f8000007fffffffull;
            int sl = 0;
            if ((lane >> 2) < take) {
                const int code = ~reinterpret_cast<volatile int *>(S.leaf_ref)[e];
                if (reinterpret_cast<volatile float *>(S.leaf_d)[e] <= thr) k = tri_key(code >> 2, lane & 3, (code & 3) + 1, sl);
            }
            nl -= take;
            const unsigned long long before = key;
            wave_min(k, sl);
            if (key != before) thr = prune_threshold(__int_as_float((int)(key >> 32)));
            continue;
        }
        // inner round: pop up to 64 nodes (fewer if their children might not fit)
        // popping `take` nodes frees `take` entries and pushes at most 2 * take: net growth <= take
        const int take = min(min(nf, 64), min(S.cap - nf, (kCoopLeaves - nl) / 2));
        const int e = nf - take + lane;
        bool v0 = false, v1 = false;
        int c0 = 0, c1 = 0;
        float d0 = 0.f, d1 = 0.f;
        if (lane < take) {
            const int node = reinterpret_cast<volatile int *>(S.frontier)[e];
            const float4 *q4 = reinterpret_cast<const float4 *>(m.nodes + node);
            const float4 n0 = q4[0], n1 = q4[1], n2 = q4[2];
            const float2 ids = *reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(m.nodes + node) + 12);
            d0 = box_dist2(n0.x, n0.z, n1.x, n1.z, n2.x, n2.z, p);
            d1 = box_dist2(n0.y, n0.w, n1.y, n1.w, n2.y, n2.w, p);
            c0 = __float_as_int(ids.x); c1 = __float_as_int(ids.y);
            v0 = d0 <= thr; v1 = d1 <= thr;
        }
        nf -= take;
        __builtin_amdgcn_wave_barrier();
        const unsigned long long bi0 = __ballot(v0 && c0 >= 0), bi1 = __ballot(v1 && c1 >= 0);
        const unsigned long long bl0 = __ballot(v0 && c0 < 0), bl1 = __ballot(v1 && c1 < 0);
        if (v0 && c0 >= 0) S.frontier[nf + __popcll(bi0 & lt_mask)] = c0;
        if (v1 && c1 >= 0) S.frontier[nf + __popcll(bi0) + __popcll(bi1 & lt_mask)] = c1;
        if (v0 && c0 < 0) { const int o = nl + __popcll(bl0 & lt_mask); S.leaf_ref[o] = c0; S.leaf_d[o] = d0; }
        if (v1 && c1 < 0) { const int o = nl + __popcll(bl0) + __popcll(bl1 & lt_mask); S.leaf_ref[o] = c1; S.leaf_d[o] = d1; }
        nf += __popcll(bi0) + __popcll(bi1);
        nl += __popcll(bl0) + __popcll(bl1);
        __builtin_amdgcn_wave_barrier();
    }
    Nearest nr;
    nr.d2 = __int_as_float((int)(key >> 32)); nr.slot = slot; nr.face = (int)(key & 0xffffffffu);
    return nr;
}

// Brute force over all triangle slots, staged through LDS in tiles (validation path).
// Reads the same TriPre constants as the packet traversal (slot s = record s&3 of leaf s>>2).
constexpr int kBruteTile = 128;   // 128 x 96 B = 12 KiB of LDS
template <int BLOCK>
__device__ __forceinline__ Nearest nearest_brute(const MeshDev &m, f3 p, float *tile /* LDS, kBruteTile*24 floats */)
{
    Nearest nr; nr.d2 = INFINITY; nr.slot = 0; nr.face = 0x7fffffff;
    const float *src = reinterpret_cast<const float *>(m.leaves);
    for (int base = 0; base < m.n_tris; base += kBruteTile) {
        __syncthreads();
        const int n = min(kBruteTile, m.n_tris - base);
        for (int k = threadIdx.x; k < n * 24; k += BLOCK) {
            const int s = base + k / 24, fld = k % 24;        // slot s = pair (s>>1)&1, lane s&1 of leaf s>>2
            tile[k] = src[((size_t)(s >> 2) * 2 + ((s >> 1) & 1)) * 48 + fld * 2 + (s & 1)];
        }
        __syncthreads();
        for (int t = 0; t < n; ++t) {
            const float *r = tile + t * 24;
            const float d2 = tri_dist2(p, load_tric(r));
            const int face = __float_as_int(r[22]);
            if (d2 < nr.d2 || (d2 == nr.d2 && face < nr.face)) { nr.d2 = d2; nr.slot = base + t; nr.face = face; }
        }
    }
    return nr;
}

__device__ __forceinline__ int bin_cell(float v, float v0, float inv, int g)
{
    const int c = (int)floorf((v - v0) * inv);
    return min(max(c, 0), g - 1);
}

__device__ __forceinline__ bool inside_bins(const MeshDev &m, f3 p)
{
    if (!(p.y >= m.bin_y0 && p.y <= m.bin_y1 && p.z >= m.bin_z0 && p.z <= m.bin_z1)) return false;
    const int cy = bin_cell(p.y, m.bin_y0, m.bin_inv_y, m.gy);
    const int cz = bin_cell(p.z, m.bin_z0, m.bin_inv_z, m.gz);
    const int cell = cz * m.gy + cy;
    const int beg = m.bin_start[cell], end = m.bin_start[cell + 1];
    int cnt = 0;
    for (int k = beg; k < end; ++k) {
        f3 a, b, c; int ia, ib, ic;
        load_tri_full(m.tris + m.bin_slots[k], a, b, c, ia, ib, ic);
        cnt += ray_hit(p, a, b, c, ia, ib, ic);
    }
    return (cnt & 1) != 0;
}

__device__ __forceinline__ bool inside_brute(const MeshDev &m, f3 p)
{
    int cnt = 0;
    for (int s = 0; s < m.n_tris; ++s) {
        f3 a, b, c; int ia, ib, ic;
        load_tri_full(m.tris + s, a, b, c, ia, ib, ic);
        if (ia >= 0) cnt += ray_hit(p, a, b, c, ia, ib, ic);     // ia < 0: padding copy of a short leaf
    }
    return (cnt & 1) != 0;
}

// Lattice mode: every point of an x-row shares (y,z), hence the same set of triangles whose (y,z)
// projection contains it - the 2-D half of the ray test does not depend on x.  k_row_crossings
// finds that set once per row (<= kRowCap slots, ascending); the per-point test then only
// re-evaluates ray_hit() on those few triangles instead of scanning the whole bin.
constexpr int kRowCap = 16;

__device__ __forceinline__ bool ray_covers(f3 p, f3 a, f3 b, f3 c, int ia, int ib, int ic)
{
    float e_ab, e_bc, e_ca; bool s_ab, s_bc, s_ca;
    oriented_edge(ia, a, ib, b, p.y, p.z, e_ab, s_ab);
    oriented_edge(ib, b, ic, c, p.y, p.z, e_bc, s_bc);
    oriented_edge(ic, c, ia, a, p.y, p.z, e_ca, s_ca);
    return s_ab == s_bc && s_bc == s_ca;
}

__device__ __forceinline__ bool inside_row(const MeshDev &m, f3 p, const int32_t *row_count, const int32_t *row_slots, int64_t row)
{
    const int n = row_count[row];
    if (n < 0) return inside_bins(m, p);          // list overflowed: fall back to the bin scan
    int cnt = 0;
    for (int k = 0; k < n; ++k) {
        f3 a, b, c; int ia, ib, ic;
        load_tri_full(m.tris + row_slots[row * kRowCap + k], a, b, c, ia, ib, ic);
        cnt += ray_hit(p, a, b, c, ia, ib, ic);
    }
    return (cnt & 1) != 0;
}

// S5: per-point tail of cal_sdf_batch (mesh_util.py:375-394)
struct SdfOut { float sdf; f3 nrm; f3 cm; float vis; };

__device__ __forceinline__ SdfOut sdf_attrs(const MeshDev &m, f3 p, const Nearest &nr, bool inside)
{
    f3 v0, v1, v2;
    load_tri_pos(m.tris + nr.slot, v0, v1, v2);
    const float4 *q = reinterpret_cast<const float4 *>(m.attr + nr.slot);
    const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5];
    // n[3][3] = a0.xyzw a1.xyzw a2.x ; cm[3][3] = a2.yzw a3.xyzw a4.xy ; vis[3] = a4.zw a5.x
    const float n0[3] = {a0.x, a0.y, a0.z}, n1[3] = {a0.w, a1.x, a1.y}, n2[3] = {a1.z, a1.w, a2.x};
    const float c0[3] = {a2.y, a2.z, a2.w}, c1[3] = {a3.x, a3.y, a3.z}, c2[3] = {a3.w, a4.x, a4.y};
    const float s0 = a4.z, s1 = a4.w, s2 = a5.x;
    // barycentric_coordinates_of_projection (unclamped)
    const f3 u = sub3(v1, v0), v = sub3(v2, v0);
    const f3 n = cross3(u, v);
    float s = dot3(n, n);
    if (s == 0.0f) s = 1e-6f;
    const float inv = 1.0f / s;
    const f3 ww = sub3(p, v0);
    const float b2 = dot3(cross3(u, ww), n) * inv;
    const float b1 = dot3(cross3(ww, v), n) * inv;
    const float w0 = (1.0f - b1) - b2, w1 = b1, w2 = b2;
    SdfOut o;
    o.cm = mk3(fmaf(c2[0], w2, fmaf(c1[0], w1, c0[0] * w0)), fmaf(c2[1], w2, fmaf(c1[1], w1, c0[1] * w0)),
               fmaf(c2[2], w2, fmaf(c1[2], w1, c0[2] * w0)));
    const float nx = fmaf(n2[0], w2, fmaf(n1[0], w1, n0[0] * w0));
    const float ny = fmaf(n2[1], w2, fmaf(n1[1], w1, n0[1] * w0));
    const float nz = fmaf(n2[2], w2, fmaf(n1[2], w1, n0[2] * w0));
    o.nrm = mk3(-nx, ny, -nz);
    const float vsum = fmaf(s2, w2, fmaf(s1, w1, s0 * w0));
    o.vis = (vsum >= 0.1f) ? 1.0f : 0.0f;
    const float dist = sqrtf(nr.d2) / sqrtf(3.0f);
    o.sdf = inside ? dist : -dist;
    return o;
}

// ---------------------------------------------------------------------------------------------
// feature gather (grid_sample, bilinear / trilinear, zeros padding, align_corners=True)
// planes: [n_select][H][W][cpad] channel-last, one tap = cpad/4 float4 loads
// ---------------------------------------------------------------------------------------------
template <int C4>
__device__ __forceinline__ void gather_planes(const FeatDev &f, int sel, float x, float y, float *out /* C4*4 */)
{
    const int H = f.H, W = f.W;
    const float ix = ((x + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((y + 1.0f) / 2.0f) * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = ((float)x1 - ix) * ((float)y1 - iy);
    const float ne = (ix - (float)x0) * ((float)y1 - iy);
    const float sw = ((float)x1 - ix) * (iy - (float)y0);
    const float se = (ix - (float)x0) * (iy - (float)y0);
    const float4 *base = reinterpret_cast<const float4 *>(f.planes) + (size_t)sel * H * W * C4;
    const bool bx0 = x0 >= 0 && x0 < W, bx1 = x1 >= 0 && x1 < W, by0 = y0 >= 0 && y0 < H, by1 = y1 >= 0 && y1 < H;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < C4; ++c) {
        const float4 t00 = (bx0 && by0) ? base[((size_t)y0 * W + x0) * C4 + c] : zero;
        const float4 t01 = (bx1 && by0) ? base[((size_t)y0 * W + x1) * C4 + c] : zero;
        const float4 t10 = (bx0 && by1) ? base[((size_t)y1 * W + x0) * C4 + c] : zero;
        const float4 t11 = (bx1 && by1) ? base[((size_t)y1 * W + x1) * C4 + c] : zero;
        float acc;
        acc = t00.x * nw; acc += t01.x * ne; acc += t10.x * sw; acc += t11.x * se; out[4 * c + 0] = acc;
        acc = t00.y * nw; acc += t01.y * ne; acc += t10.y * sw; acc += t11.y * se; out[4 * c + 1] = acc;
        acc = t00.z * nw; acc += t01.z * ne; acc += t10.z * sw; acc += t11.z * se; out[4 * c + 2] = acc;
        acc = t00.w * nw; acc += t01.w * ne; acc += t10.w * sw; acc += t11.w * se; out[4 * c + 3] = acc;
    }
}

template <int C4>
__device__ __forceinline__ void gather_volume(const FeatDev &f, float x, float y, float z, float *out /* C4*4 */)
{
    const int D = f.Dv, H = f.Hv, W = f.Wv;
    const float ix = ((x + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((y + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((z + 1.0f) / 2.0f) * (float)(D - 1);
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
    const float tx = ix - (float)x0, ty = iy - (float)y0, tz = iz - (float)z0;
    const float4 *base = reinterpret_cast<const float4 *>(f.vol);
#pragma unroll
    for (int c = 0; c < C4 * 4; ++c) out[c] = 0.0f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
                const float wgt = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) * (dz ? tz : 1.0f - tz);
                if (xi >= 0 && xi < W && yi >= 0 && yi < H && zi >= 0 && zi < D) {
#pragma unroll
                    for (int c = 0; c < C4; ++c) {
                        const float4 t = base[(((size_t)zi * H + yi) * W + xi) * C4 + c];
                        out[4 * c + 0] += t.x * wgt; out[4 * c + 1] += t.y * wgt;
                        out[4 * c + 2] += t.z * wgt; out[4 * c + 3] += t.w * wgt;
                    }
                }
            }
}

// dispatch on the padded channel count (cpad in {4,8,12,16})
__device__ __forceinline__ void gather_planes_dyn(const FeatDev &f, int sel, float x, float y, float *out)
{
    switch (f.cpad >> 2) {
        case 1: gather_planes<1>(f, sel, x, y, out); break;
        case 2: gather_planes<2>(f, sel, x, y, out); break;
        case 3: gather_planes<3>(f, sel, x, y, out); break;
        default: gather_planes<4>(f, sel, x, y, out); break;
    }
}

// ---------------------------------------------------------------------------------------------
// point sources
// ---------------------------------------------------------------------------------------------
// Lattice tiling: a wavefront owns a 4x4x4 block of lattice points (spatially compact, so its
// 64 BVH traversals follow nearly the same path); a 256-thread workgroup owns 16x4x4.
// Workgroups are numbered y-slowest; consecutive workgroups go to different XCDs (hardware
// round-robin), which balances the strongly position-dependent traversal cost.  The optional
// contiguous-band-per-XCD remap (ICON_AMD_XCD_REMAP=1) keeps each L2 on one band of the body but
// was measured 1.6x slower: the mesh fits every L2 anyway and the bands are unequal work.
struct LatticeMap {
    int res, z0, nz;           // evaluated planes [z0, z0+nz)
    int tx, ty, tz;            // tile counts
    int remap;                 // 1: contiguous run of tiles per XCD, 0: tiles interleaved over XCDs
};

__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    // bijective "contiguous chunk per XCD" remap (blocks are dispatched round-robin to 8 XCDs)
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

__device__ __forceinline__ bool lattice_point(const LatticeMap &L, int &ix, int &iy, int &iz)
{
    const int nb = L.tx * L.ty * L.tz;
    const int t = L.remap ? xcd_remap(blockIdx.x, nb) : (int)blockIdx.x;
    const int bty = t / (L.tz * L.tx);
    const int rem = t - bty * (L.tz * L.tx);
    const int btz = rem / L.tx, btx = rem - btz * L.tx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ix = btx * 16 + wave * 4 + (lane & 3);
    iy = bty * 4 + ((lane >> 2) & 3);
    iz = btz * 4 + (lane >> 4);
    return ix < L.res && iy < L.res && iz < L.nz;
}

__device__ __forceinline__ f3 lattice_world(int res, int ix, int iy, int iz)
{
    // batch_eval, seg3d_lossless.py:132-137 with align_corners=True:
    //   coords.float() / (R-1) * (b_max - b_min) + b_min, b_min=[-1,1,-1], b_max=[1,-1,1]
    const float rm1 = (float)(res - 1);
    const float fx = (float)ix / rm1, fy = (float)iy / rm1, fz = (float)iz / rm1;
    return mk3(fx * 2.0f + (-1.0f), fy * (-2.0f) + 1.0f, fz * 2.0f + (-1.0f));
}

__device__ __forceinline__ f3 project(const Calib &c, f3 p)
{
    // orthogonal(): baddbmm(trans, rot, points), geometry.py:54-56
    // k-ordered fma chain, then + trans (what ATen's CPU baddbmm computes for K = 3); exact for identity
    f3 r;
    r.x = fmaf(c.m[2], p.z, fmaf(c.m[1], p.y, c.m[0] * p.x)) + c.m[3];
    r.y = fmaf(c.m[6], p.z, fmaf(c.m[5], p.y, c.m[4] * p.x)) + c.m[7];
    r.z = fmaf(c.m[10], p.z, fmaf(c.m[9], p.y, c.m[8] * p.x)) + c.m[11];
    return r;
}

__device__ __forceinline__ uint32_t in_cube_bit(f3 p)
{
    const bool in = p.x > -1.0f && p.x < 1.0f && p.y > -1.0f && p.y < 1.0f && p.z > -1.0f && p.z < 1.0f;
    return in ? kCodeInCube : 0u;
}

__device__ __forceinline__ void store_row(float *X, int64_t i, const float *row)
{
    float4 *dst = reinterpret_cast<float4 *>(X + i * kXRow);
    dst[0] = make_float4(row[0], row[1], row[2], row[3]);
    dst[1] = make_float4(row[4], row[5], row[6], row[7]);
    dst[2] = make_float4(row[8], row[9], row[10], row[11]);
    dst[3] = make_float4(row[12], row[13], row[14], row[15]);
}

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// cal_sdf_batch for explicit points (API parity with the reference function; also the unit-test
// surface for the geometry kernels)
template <bool BRUTE>
__global__ __launch_bounds__(kBlock) void k_sdf_query(MeshDev m, const float *__restrict__ pts, int64_t N,
                                                      float *sdf, float *nrm, float *cm, float *vis,
                                                      int64_t *face, uint8_t *inside_out, const int32_t *__restrict__ perm)
{
    __shared__ int lds[kBruteTile * 24];   // brute: TriPre tile; packet: 4 wave stacks
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < N;
    if (live && perm) i = perm[i];           // Morton order (sort_points.hip): a wave's 64 points are neighbours
    const int64_t ic = live ? i : (N - 1);
    const f3 p = mk3(pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2]);
    Nearest nr;
    bool ins;
    if (BRUTE) { nr = nearest_brute<kBlock>(m, p, reinterpret_cast<float *>(lds)); ins = inside_brute(m, p); }
    else       { nr = nearest_packet(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth); ins = inside_bins(m, p); }
    if (!live) return;
    const SdfOut o = sdf_attrs(m, p, nr, ins);
    sdf[i] = o.sdf;
    nrm[3 * i] = o.nrm.x; nrm[3 * i + 1] = o.nrm.y; nrm[3 * i + 2] = o.nrm.z;
    cm[3 * i] = o.cm.x; cm[3 * i + 1] = o.cm.y; cm[3 * i + 2] = o.cm.z;
    vis[i] = o.vis;
    if (face) face[i] = nr.face;
    if (inside_out) inside_out[i] = ins ? 1 : 0;
}

// point mode, one wavefront per point (see nearest_coop)
__global__ __launch_bounds__(kCoopWaves * 64) void k_nearest_coop(MeshDev m, Calib cal, const float *__restrict__ pts, int64_t N, int2 *__restrict__ near,
                                                                 int cap)
{
    extern __shared__ __attribute__((aligned(16))) char coop_smem[];
    const int64_t i = (int64_t)blockIdx.x * kCoopWaves + (threadIdx.x >> 6);
    if (i >= N) return;
    const f3 p = project(cal, mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    const Nearest nr = nearest_coop(m, p, coop_lds(coop_smem, threadIdx.x >> 6, cap));
    if ((threadIdx.x & 63) == 0) near[i] = make_int2(nr.slot, __float_as_int(nr.d2));
}

__global__ __launch_bounds__(kCoopWaves * 64) void k_sdf_query_coop(MeshDev m, const float *__restrict__ pts, int64_t N,
                                                                   float *sdf, float *nrm, float *cm, float *vis,
                                                                   int64_t *face, uint8_t *inside_out, int cap)
{
    extern __shared__ __attribute__((aligned(16))) char coop_smem[];
    const int64_t i = (int64_t)blockIdx.x * kCoopWaves + (threadIdx.x >> 6);
    if (i >= N) return;
    const f3 p = mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    const Nearest nr = nearest_coop(m, p, coop_lds(coop_smem, threadIdx.x >> 6, cap));
    if ((threadIdx.x & 63) != 0) return;
    const bool ins = inside_bins(m, p);
    const SdfOut o = sdf_attrs(m, p, nr, ins);
    sdf[i] = o.sdf;
    nrm[3 * i] = o.nrm.x; nrm[3 * i + 1] = o.nrm.y; nrm[3 * i + 2] = o.nrm.z;
    cm[3 * i] = o.cm.x; cm[3 * i + 1] = o.cm.y; cm[3 * i + 2] = o.cm.z;
    vis[i] = o.vis;
    if (face) face[i] = nr.face;
    if (inside_out) inside_out[i] = ins ? 1 : 0;
}

// Nearest-triangle search as its own launch: the packet traversal needs ~36 VGPRs, so it runs at
// full occupancy (8 waves / SIMD hide the dependent scalar-load chain), which the register-heavier
// attribute / gather code below would cap at 5.  Output: (slot, bits of d^2) per point, 8 B.
template <bool LATTICE>
__global__ __launch_bounds__(kBlock) void k_nearest(MeshDev m, Calib cal, LatticeMap L, const float *__restrict__ pts, int64_t N,
                                                    int2 *__restrict__ near, const int32_t *__restrict__ perm)
{
    __shared__ int lds[(kBlock / 64) * kStackDepth];
    int64_t i; bool live; f3 p;
    if (LATTICE) {
        int ix, iy, iz;
        live = lattice_point(L, ix, iy, iz);
        const int cx = min(ix, L.res - 1), cy = min(iy, L.res - 1), cz = min(iz, L.nz - 1);
        p = lattice_world(L.res, cx, cy, cz + L.z0);
        i = ((int64_t)cz * L.res + cy) * L.res + cx;
    } else {
        i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        live = i < N;
        if (!live) i = N - 1;
        if (perm) i = perm[i];          // Morton order: the wave's 64 points are neighbours (sort_points.hip)
        p = project(cal, mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    }
    const Nearest nr = nearest_packet(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth);
    if (live) near[i] = make_int2(nr.slot, __float_as_int(nr.d2));
}

// Feature assembly: one 16-float row per point,
//   icon : [img(csel) | sdf | cmap r g b | norm x y z | 0.. | code]
//   pamir: [img(C) | vol(Cv) | 0.. | code]      pifu: [img(C) | z | 0.. | code]
// Rows are indexed by the point's linear index (lattice: (z*R + y)*R + x relative to plane z0).
template <int PRIOR, bool LATTICE, bool BRUTE>
__global__ __launch_bounds__(kBlock) void k_features(MeshDev m, FeatDev f, Calib cal, LatticeMap L,
                                                     const float *__restrict__ pts, int64_t N,
                                                     float sdf_clip, int cmap_local,
                                                     const int32_t *__restrict__ row_count, const int32_t *__restrict__ row_slots,
                                                     const int2 *__restrict__ near, float *__restrict__ X,
                                                     uint8_t *__restrict__ code8)
{
    __shared__ int lds[(PRIOR == ICON_PRIOR_ICON && BRUTE) ? kBruteTile * 24 : 1];
    int64_t i; bool live; f3 p;
    int64_t yz_row = 0;
    if (LATTICE) {
        int ix, iy, iz;
        live = lattice_point(L, ix, iy, iz);
        const int cx = min(ix, L.res - 1), cy = min(iy, L.res - 1), cz = min(iz, L.nz - 1);
        p = lattice_world(L.res, cx, cy, cz + L.z0);
        i = ((int64_t)cz * L.res + cy) * L.res + cx;
        yz_row = (int64_t)cz * L.res + cy;
    } else {
        i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        live = i < N;
        if (!live) i = N - 1;
        p = project(cal, mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    }
    float row[kXRow];
#pragma unroll
    for (int k = 0; k < kXRow; ++k) row[k] = 0.0f;
    uint32_t code = in_cube_bit(p);
    if (PRIOR == ICON_PRIOR_ICON) {
        Nearest nr;
        bool ins;
        if (BRUTE) { nr = nearest_brute<kBlock>(m, p, reinterpret_cast<float *>(lds)); ins = inside_brute(m, p); }
        else {
            const int2 nn = near[i];                       // k_nearest ran on the same stream just before
            nr.slot = nn.x; nr.d2 = __int_as_float(nn.y); nr.face = 0;
            ins = LATTICE ? inside_row(m, p, row_count, row_slots, yz_row) : inside_bins(m, p);
        }
        const SdfOut o = sdf_attrs(m, p, nr, ins);
        float s = o.sdf;
        f3 cmv = o.cm;
        if (fabsf(s) >= sdf_clip) {            // HGPIFuNet.py:298-305
            const int sg = (s > 0.0f) ? 1 : ((s < 0.0f) ? -1 : 0);
            s = (float)sg;
            code |= kCodeOutlier | ((uint32_t)(sg + 1) << kCodeSignShift);
            if (cmap_local) cmv = mk3(s, s, s);   // reference mode: patched later from the sign list
        }
        float g[16];
        gather_planes_dyn(f, (o.vis != 0.0f) ? 0 : 1, p.x, p.y, g);   // feat_select: vis==1 -> front half
        const int h = f.csel;
        for (int k = 0; k < h; ++k) row[k] = g[k];
        row[h] = s;
        row[h + 1] = cmv.x; row[h + 2] = cmv.y; row[h + 3] = cmv.z;
        row[h + 4] = o.nrm.x; row[h + 5] = o.nrm.y; row[h + 6] = o.nrm.z;
    } else {
        float g[16];
        gather_planes_dyn(f, 0, p.x, p.y, g);
        const int h = f.csel;
        for (int k = 0; k < h; ++k) row[k] = g[k];
        if (PRIOR == ICON_PRIOR_PAMIR) {
            float v[8];
            if (f.vpad == 8) gather_volume<2>(f, p.x, p.y, p.z, v); else gather_volume<1>(f, p.x, p.y, p.z, v);
            for (int k = 0; k < f.Cv; ++k) row[h + k] = v[k];
        } else {
            row[h] = p.z;
        }
    }
    row[kCodeSlot] = __int_as_float((int)code);
    if (live) { store_row(X, i, row); code8[i] = (uint8_t)code; }   // byte copy of the code word: the outlier passes stream 1 B/pt
}

// diagnostics: per-wavefront BVH work of the lattice traversal (DESIGN.md reports visited nodes / point)
__global__ __launch_bounds__(kBlock) void k_traversal_stats(MeshDev m, LatticeMap L, unsigned long long *out /* [4] */, int seeded)
{
    __shared__ int lds[(kBlock / 64) * kStackDepth];
    int ix, iy, iz;
    const bool live = lattice_point(L, ix, iy, iz);
    const int cx = min(ix, L.res - 1), cy = min(iy, L.res - 1), cz = min(iz, L.nz - 1);
    const f3 p = lattice_world(L.res, cx, cy, cz + L.z0);
    int nn = 0, nt = 0;
    Nearest nr = nearest_packet<true>(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, &nn, &nt);
    if (seeded) {       // experiment: the work of a traversal that starts from the final bound (seeded == 1) or from the
                        // bound a coarse pass would give: best distance at the tile's centre lane + tile radius (seeded == 2)
        nn = 0; nt = 0;
        float thr0 = prune_threshold(nr.d2);
        if (seeded == 2) {
            const float dc = sqrtf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(nr.d2), 21)));
            const float h = 2.0f / (float)(L.res - 1);
            const float rad = 2.6f * h;                       // > half diagonal of a 4x4x4 block (sqrt(3) * 1.5 h)
            thr0 = prune_threshold((dc + rad) * (dc + rad));
        }
        nr = nearest_packet<true>(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, &nn, &nt, thr0);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], 1ull); atomicAdd(&out[1], (unsigned long long)nn); atomicAdd(&out[2], (unsigned long long)nt);
    }
    if (live && nr.slot < 0) atomicAdd(&out[3], 1ull);
}

// one thread per (y, z) row of the slab: triangles whose (y,z) projection covers the row
__global__ __launch_bounds__(kBlock) void k_row_crossings(MeshDev m, LatticeMap L, int32_t *row_count, int32_t *row_slots)
{
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= (int64_t)L.nz * L.res) return;
    const int iy = (int)(row % L.res), iz = (int)(row / L.res);
    const f3 p = lattice_world(L.res, 0, iy, iz + L.z0);
    int n = 0;
    if (p.y >= m.bin_y0 && p.y <= m.bin_y1 && p.z >= m.bin_z0 && p.z <= m.bin_z1) {
        const int cy = bin_cell(p.y, m.bin_y0, m.bin_inv_y, m.gy);
        const int cz = bin_cell(p.z, m.bin_z0, m.bin_inv_z, m.gz);
        const int cell = cz * m.gy + cy;
        const int beg = m.bin_start[cell], end = m.bin_start[cell + 1];
        for (int k = beg; k < end; ++k) {
            const int slot = m.bin_slots[k];
            f3 a, b, c; int ia, ib, ic;
            load_tri_full(m.tris + slot, a, b, c, ia, ib, ic);
            if (ray_covers(p, a, b, c, ia, ib, ic)) {
                if (n < kRowCap) row_slots[row * kRowCap + n] = slot;
                ++n;
            }
        }
    }
    row_count[row] = (n <= kRowCap) ? n : -1;
}

// ---------------------------------------------------------------------------------------------
// outlier sign list (reference cmap mode): count -> scan -> compact -> patch, all in the linear
// point order of the call.
// ---------------------------------------------------------------------------------------------
constexpr int kScanBlock = 1024;

__device__ __forceinline__ uint32_t row_code(const float *X, int64_t i)
{
    return (uint32_t)__float_as_int(X[i * kXRow + kCodeSlot]);
}

__global__ __launch_bounds__(kScanBlock) void k_outlier_count(const uint8_t *__restrict__ code8, int64_t N, int32_t *block_counts)
{
    __shared__ int wsum[kScanBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const bool o = (i < N) && (code8[i] & kCodeOutlier);
    const unsigned long long b = __ballot(o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) s += wsum[w];
        block_counts[blockIdx.x] = s;
    }
}

// single-workgroup exclusive scan of the per-block counts (<= a few 100k entries);
// offsets are int64 (a 513^3 lattice has 1.35e8 points)
__global__ __launch_bounds__(1024) void k_scan_blocks(const int32_t *counts, int64_t nblocks, int64_t *offsets, int64_t *total)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t per = (nblocks + 1023) / 1024;
    const int64_t beg = min((int64_t)t * per, nblocks), end = min(beg + per, nblocks);
    int64_t s = 0;
    for (int64_t k = beg; k < end; ++k) s += counts[k];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int64_t run = 0;
        for (int k = 0; k < 1024; ++k) { const int64_t v = part[k]; part[k] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    int64_t run = part[t];
    for (int64_t k = beg; k < end; ++k) { offsets[k] = run; run += counts[k]; }
}

__device__ __forceinline__ int64_t outlier_rank(bool o, const int64_t *block_offsets, int *wsum)
{
    // exclusive rank of this thread's outlier among the outliers of the call (linear order)
    const unsigned long long b = __ballot(o);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int k = 0; k < w; ++k) before += wsum[k];
    before += __popcll(b & ((1ull << lane) - 1ull));
    return block_offsets[blockIdx.x] + before;
}

__global__ __launch_bounds__(kScanBlock) void k_outlier_compact(const uint8_t *__restrict__ code8, int64_t N,
                                                                const int64_t *block_offsets, int8_t *signs)
{
    __shared__ int wsum[kScanBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t r = outlier_rank(o, block_offsets, wsum);
    if (o) signs[r] = (int8_t)((int)((code >> kCodeSignShift) & 3u) - 1);
}

// cmap[j][k] = s[(3j + k) mod K]  with j = global outlier rank = rank_offset + local rank
__global__ __launch_bounds__(kScanBlock) void k_outlier_patch(float *__restrict__ X, const uint8_t *__restrict__ code8, int64_t N, int cmap_slot,
                                                              const int64_t *block_offsets, const int8_t *signs_global,
                                                              int64_t k_total, int64_t rank_offset)
{
    __shared__ int wsum[kScanBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t j = rank_offset + outlier_rank(o, block_offsets, wsum);
    if (o) {
        float *row = X + i * kXRow + cmap_slot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t mm = 3 * j + k;                 // j < K  =>  mm < 3K
            if (mm >= k_total) mm -= k_total;
            if (mm >= k_total) mm -= k_total;
            row[k] = (float)signs_global[mm];
        }
    }
}

// The same patch with the global sign list left where the all_gather put it: rank r's message is
// gathered + r * stride = [int64 count_r][int8 signs_r ...] (multi-GPU path, recon.py).  K, this
// rank's offset and the segment of every index are derived on the device from the headers, so the
// host never has to read the counts (no synchronisation between the exchange and the MLP launch).
constexpr int kMaxWorld = 64;
__global__ __launch_bounds__(kScanBlock) void k_outlier_patch_seg(float *__restrict__ X, const uint8_t *__restrict__ code8, int64_t N, int cmap_slot,
                                                                  const int64_t *block_offsets, const int8_t *__restrict__ gathered,
                                                                  int64_t stride, int world, int rank)
{
    __shared__ int wsum[kScanBlock / 64];
    __shared__ int64_t off[kMaxWorld + 1];
    if (threadIdx.x == 0) {
        int64_t a = 0;
        for (int r = 0; r < world; ++r) { off[r] = a; a += *reinterpret_cast<const int64_t *>(gathered + (int64_t)r * stride); }
        off[world] = a;
    }
    __syncthreads();
    const int64_t K = off[world];
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t j = off[rank] + outlier_rank(o, block_offsets, wsum);
    if (o && K > 0) {
        float *row = X + i * kXRow + cmap_slot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t mm = 3 * j + k;                 // j < K  =>  mm < 3K
            if (mm >= K) mm -= K;
            if (mm >= K) mm -= K;
            int r = 0;
            while (mm >= off[r + 1]) ++r;
            row[k] = (float)gathered[(int64_t)r * stride + 8 + (mm - off[r])];
        }
    }
}

// repack feature planes [C][H][W] -> [n_select][H][W][cpad] (channel-last, zero padded)
__global__ void k_pack_planes(const float *__restrict__ src, int C, int H, int W, int n_select, int csel, int cpad, float *dst)
{
    const int64_t n = (int64_t)n_select * H * W * cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpad);
        const int64_t pix = (i / cpad) % ((int64_t)H * W);
        const int sel = (int)(i / ((int64_t)cpad * H * W));
        dst[i] = (c < csel) ? src[((int64_t)(sel * csel + c)) * H * W + pix] : 0.0f;
    }
}

}  // namespace icon

// =============================================================================================
// C ABI
// =============================================================================================
using namespace icon;


extern "C" int icon_sdf_query(const icon_mesh_t *mesh, const float *d_points, int64_t N,
                              float *d_sdf, float *d_norm, float *d_cmap, float *d_vis,
                              int64_t *d_face, uint8_t *d_inside, int search, void *stream)
{
    ICON_ARG(mesh && d_points && d_sdf && d_norm && d_cmap && d_vis, "icon_sdf_query: null argument");
    ICON_ARG(N >= 0, "icon_sdf_query: negative N");
    if (N == 0) return ICON_OK;
    const int64_t nb = (N + kBlock - 1) / kBlock;
    ICON_ARG(nb < (1ll << 31), "icon_sdf_query: N too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (search == ICON_SEARCH_BRUTE) {
        hipLaunchKernelGGL(k_sdf_query<true>, dim3((unsigned)nb), dim3(kBlock), 0, st, mesh->dev, d_points, N, d_sdf, d_norm,
                           d_cmap, d_vis, d_face, d_inside, nullptr);
    } else if (N < kPacketMinPoints) {    // sparse / unordered points: one wavefront per point (see nearest_coop)
        const int cap = coop_cap((int)mesh->stats[1]);
        hipLaunchKernelGGL(k_sdf_query_coop, dim3((unsigned)((N + kCoopWaves - 1) / kCoopWaves)), dim3(kCoopWaves * 64), kCoopWaves * coop_wave_bytes(cap), st,
                           mesh->dev, d_points, N, d_sdf, d_norm, d_cmap, d_vis, d_face, d_inside, cap);
    } else {                              // large batches: packets over the Morton order (scratch freed after a stream sync)
        icon_work tmp;
        static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        const int32_t *perm = nullptr;
        int rc = morton_order(&tmp, d_points, ident, N, st, &perm);
        if (!rc) {
            hipLaunchKernelGGL(k_sdf_query<false>, dim3((unsigned)nb), dim3(kBlock), 0, st, mesh->dev, d_points, N, d_sdf, d_norm,
                               d_cmap, d_vis, d_face, d_inside, perm);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(ICON_ERR_HIP, "icon_sdf_query: launch failed");
        }
        (void)hipFree(tmp.d_sort_keys); (void)hipFree(tmp.d_sort_idx); (void)hipFree(tmp.d_sort_tmp);
        return rc;
    }
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

extern "C" int icon_feat_create(const float *d_planes, int C, int H, int W, int n_select,
                                const float *d_vol, int Cv, int Dv, int Hv, int Wv,
                                void *stream, icon_feat_t **out)
{
    ICON_ARG(out != nullptr, "icon_feat_create: out is null");
    *out = nullptr;
    ICON_ARG(d_planes && C > 0 && H > 1 && W > 1, "icon_feat_create: bad planes");
    ICON_ARG(n_select == 1 || n_select == 2, "icon_feat_create: n_select must be 1 or 2");
    ICON_ARG(C % n_select == 0, "icon_feat_create: C not divisible by n_select");
    const int csel = C / n_select;
    const int cpad = (csel + 3) & ~3;
    if (cpad > 16) return fail(ICON_ERR_UNSUPPORTED, "icon_feat_create: more than 16 channels per tap");
    if (d_vol) {
        ICON_ARG(Cv > 0 && Dv > 1 && Hv > 1 && Wv > 1, "icon_feat_create: bad volume");
        if (Cv > 8) return fail(ICON_ERR_UNSUPPORTED, "icon_feat_create: more than 8 volume channels");
    }
    hipStream_t st = (hipStream_t)stream;
    icon_feat *f = new icon_feat();
    const size_t n = (size_t)n_select * H * W * cpad;
    hipError_t e = hipMalloc((void **)&f->d_planes, n * sizeof(float));
    if (e != hipSuccess) { delete f; return fail(ICON_ERR_HIP, std::string("hipMalloc planes: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(k_pack_planes, dim3(1024), dim3(256), 0, st, d_planes, C, H, W, n_select, csel, cpad, f->d_planes);
    FeatDev &d = f->dev;
    d.planes = f->d_planes; d.C = C; d.H = H; d.W = W; d.n_select = n_select; d.csel = csel; d.cpad = cpad;
    d.vol = nullptr; d.Cv = 0; d.Dv = d.Hv = d.Wv = 0; d.vpad = 0;
    if (d_vol) {
        const int vpad = (Cv + 3) & ~3;
        const size_t nv = (size_t)Dv * Hv * Wv * vpad;
        e = hipMalloc((void **)&f->d_vol, nv * sizeof(float));
        if (e != hipSuccess) { icon_feat_destroy(f); return fail(ICON_ERR_HIP, std::string("hipMalloc vol: ") + hipGetErrorString(e)); }
        // a volume is a "plane" of H' = D*H rows: same channel-last repack
        hipLaunchKernelGGL(k_pack_planes, dim3(1024), dim3(256), 0, st, d_vol, Cv, Dv * Hv, Wv, 1, Cv, vpad, f->d_vol);
        d.vol = f->d_vol; d.Cv = Cv; d.Dv = Dv; d.Hv = Hv; d.Wv = Wv; d.vpad = vpad;
    }
    e = hipGetLastError();
    if (e != hipSuccess) { icon_feat_destroy(f); return fail(ICON_ERR_HIP, std::string("pack planes: ") + hipGetErrorString(e)); }
    *out = f;
    return ICON_OK;
}

extern "C" int icon_feat_destroy(icon_feat_t *f)
{
    if (!f) return ICON_OK;
    (void)hipFree(f->d_planes); (void)hipFree(f->d_vol);
    delete f;
    return ICON_OK;
}

extern "C" int icon_work_create(icon_work_t **out)
{
    ICON_ARG(out != nullptr, "icon_work_create: out is null");
    *out = new icon_work();
    return ICON_OK;
}

extern "C" int icon_work_destroy(icon_work_t *w)
{
    if (!w) return ICON_OK;
    (void)hipFree(w->d_x); (void)hipFree(w->d_block_counts); (void)hipFree(w->d_block_offsets);
    (void)hipFree(w->d_signs); (void)hipFree(w->d_total); (void)hipFree(w->d_row_count); (void)hipFree(w->d_row_slots); (void)hipFree(w->d_near); (void)hipFree(w->d_code8);
    (void)hipFree(w->d_sort_keys); (void)hipFree(w->d_sort_idx); (void)hipFree(w->d_sort_tmp);
    for (int k = 0; k < 4; ++k) if (w->ev[k]) (void)hipEventDestroy(w->ev[k]);
    icon::mc_destroy(w->mc);
    delete w;
    return ICON_OK;
}

extern "C" int icon_work_profile(icon_work_t *w, int enable)
{
    ICON_ARG(w != nullptr, "icon_work_profile: work is null");
    if (enable && !w->ev[0])
        for (int k = 0; k < 4; ++k) ICON_HIP(hipEventCreate(&w->ev[k]));
    w->prof = enable != 0;
    w->ev_valid = false;
    return ICON_OK;
}

extern "C" int icon_work_stage_ms(icon_work_t *w, float out_ms[3])
{
    ICON_ARG(w && out_ms, "icon_work_stage_ms: null argument");
    if (!w->prof || !w->ev_valid) return fail(ICON_ERR_STATE, "icon_work_stage_ms: no profiled call on this workspace");
    ICON_HIP(hipEventSynchronize(w->ev[3]));
    for (int k = 0; k < 3; ++k) ICON_HIP(hipEventElapsedTime(&out_ms[k], w->ev[k], w->ev[k + 1]));
    return ICON_OK;
}

namespace {

inline void mark(icon_work *w, int k, hipStream_t st)
{
    if (w->prof) { (void)hipEventRecord(w->ev[k], st); if (k == 3) w->ev_valid = true; }
}

int ensure_work(icon_work *w, int64_t n_points)
{
    if (n_points > w->cap_points) {
        (void)hipFree(w->d_x); w->d_x = nullptr; w->cap_points = 0;
        ICON_HIP(hipMalloc((void **)&w->d_x, (size_t)n_points * kXRow * sizeof(float)));
        (void)hipFree(w->d_near); w->d_near = nullptr;
        ICON_HIP(hipMalloc((void **)&w->d_near, (size_t)n_points * 8));
        (void)hipFree(w->d_code8); w->d_code8 = nullptr;
        ICON_HIP(hipMalloc((void **)&w->d_code8, (size_t)n_points));
        w->cap_points = n_points;
    }
    const int64_t nblk = (n_points + kScanBlock - 1) / kScanBlock;
    if (nblk > w->cap_blocks) {
        (void)hipFree(w->d_block_counts); (void)hipFree(w->d_block_offsets);
        w->d_block_counts = nullptr; w->d_block_offsets = nullptr; w->cap_blocks = 0;
        ICON_HIP(hipMalloc((void **)&w->d_block_counts, (size_t)nblk * sizeof(int32_t)));
        ICON_HIP(hipMalloc((void **)&w->d_block_offsets, (size_t)nblk * sizeof(int64_t)));
        w->cap_blocks = nblk;
    }
    if (n_points > w->cap_signs) {
        (void)hipFree(w->d_signs); w->d_signs = nullptr; w->cap_signs = 0;
        ICON_HIP(hipMalloc((void **)&w->d_signs, (size_t)n_points));
        w->cap_signs = n_points;
    }
    if (!w->d_total) ICON_HIP(hipMalloc((void **)&w->d_total, sizeof(int64_t)));
    return ICON_OK;
}

int check_prior(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior, int *c0)
{
    ICON_ARG(feat != nullptr, "feature handle is null");
    const FeatDev &f = feat->dev;
    if (prior == ICON_PRIOR_ICON) {
        ICON_ARG(mesh != nullptr, "icon prior needs a mesh handle");
        ICON_ARG(f.n_select == 2, "icon prior needs feature planes created with n_select = 2");
        *c0 = f.csel + 7;
    } else if (prior == ICON_PRIOR_PAMIR) {
        ICON_ARG(f.n_select == 1 && f.vol != nullptr, "pamir prior needs n_select = 1 and a volume");
        *c0 = f.csel + f.Cv;
    } else if (prior == ICON_PRIOR_PIFU) {
        ICON_ARG(f.n_select == 1, "pifu prior needs n_select = 1");
        *c0 = f.csel + 1;
    } else {
        return fail(ICON_ERR_ARG, "unknown prior_type");
    }
    if (*c0 > kCodeSlot) return fail(ICON_ERR_UNSUPPORTED, "more than 15 MLP input channels");
    return ICON_OK;
}

template <bool LATTICE>
int launch_features(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior, float sdf_clip, int cmap_mode,
                    const Calib &cal, const LatticeMap &L, const float *d_points, int64_t N, int search,
                    icon_work *work, hipStream_t st)
{
    float *d_x = work->d_x;
    int32_t *row_count = nullptr, *row_slots = nullptr;
    if (LATTICE && prior == ICON_PRIOR_ICON && search != ICON_SEARCH_BRUTE) {
        const int64_t rows = (int64_t)L.nz * L.res;
        if (rows > work->cap_rows) {
            (void)hipFree(work->d_row_count); (void)hipFree(work->d_row_slots);
            work->d_row_count = nullptr; work->d_row_slots = nullptr; work->cap_rows = 0;
            ICON_HIP(hipMalloc((void **)&work->d_row_count, (size_t)rows * sizeof(int32_t)));
            ICON_HIP(hipMalloc((void **)&work->d_row_slots, (size_t)rows * kRowCap * sizeof(int32_t)));
            work->cap_rows = rows;
        }
        row_count = work->d_row_count; row_slots = work->d_row_slots;
        hipLaunchKernelGGL(k_row_crossings, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, mesh->dev, L,
                           row_count, row_slots);
    }
    int64_t nb;
    if (LATTICE) nb = (int64_t)L.tx * L.ty * L.tz; else nb = (N + kBlock - 1) / kBlock;
    ICON_ARG(nb > 0 && nb < (1ll << 31), "too many workgroups for one launch");
    const dim3 grid((unsigned)nb), block(kBlock);
    const MeshDev md = mesh ? mesh->dev : MeshDev{};
    const int local = (cmap_mode == ICON_CMAP_LOCAL) ? 1 : 0;
    const bool brute = (search == ICON_SEARCH_BRUTE);
    int2 *near = nullptr;
    if (prior == ICON_PRIOR_ICON && !brute) {
        near = reinterpret_cast<int2 *>(work->d_near);
        // point mode: sparse batches walk the tree one lane per point; a batch dense enough for a wave's 64
        // Morton neighbours to be close together (>= ~2M points in the cube) goes through the packet kernel
        const int32_t *perm = nullptr;
        static const int mode = getenv("ICON_AMD_POINT_SEARCH") ? atoi(getenv("ICON_AMD_POINT_SEARCH")) : 0;   // 0 auto, 2 coop, 3 packets
        if (!LATTICE && mode != 3 && (N < kPacketMinPoints || mode == 2)) {
            { const int cap = coop_cap((int)mesh->stats[1]);
              hipLaunchKernelGGL(k_nearest_coop, dim3((unsigned)((N + kCoopWaves - 1) / kCoopWaves)), dim3(kCoopWaves * 64), kCoopWaves * coop_wave_bytes(cap), st, md, cal, d_points, N, near, cap); }
        } else {
            if (!LATTICE) {
                const int rc = morton_order(work, d_points, cal.m, N, st, &perm);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((k_nearest<LATTICE>), grid, block, 0, st, md, cal, L, d_points, N, near, perm);
        }
    }
#define ICON_LAUNCH(P, B) hipLaunchKernelGGL((k_features<P, LATTICE, B>), grid, block, 0, st, md, feat->dev, cal, L, d_points, N, sdf_clip, local, row_count, row_slots, near, d_x, work->d_code8)
    if (prior == ICON_PRIOR_ICON) { if (brute) ICON_LAUNCH(ICON_PRIOR_ICON, true); else ICON_LAUNCH(ICON_PRIOR_ICON, false); }
    else if (prior == ICON_PRIOR_PAMIR) ICON_LAUNCH(ICON_PRIOR_PAMIR, false);
    else ICON_LAUNCH(ICON_PRIOR_PIFU, false);
#undef ICON_LAUNCH
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

// count + scan + compact over rows [0, N): fills w->d_block_offsets, w->d_total, and `signs`
int outlier_list(icon_work *w, int64_t N, int8_t *signs, hipStream_t st)
{
    const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(k_outlier_count, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, w->d_code8, N, w->d_block_counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, w->d_block_counts, nblk, w->d_block_offsets, w->d_total);
    hipLaunchKernelGGL(k_outlier_compact, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, w->d_code8, N, w->d_block_offsets, signs);
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

}  // namespace

namespace icon {
// device-side K: same as k_outlier_patch but K and the list come from this call's own scan
__global__ __launch_bounds__(kScanBlock) void k_outlier_patch_self(float *__restrict__ X, const uint8_t *__restrict__ code8, int64_t N, int cmap_slot,
                                                                   const int64_t *block_offsets, const int8_t *signs,
                                                                   const int64_t *k_total_dev)
{
    __shared__ int wsum[kScanBlock / 64];
    const int64_t K = *k_total_dev;
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t j = outlier_rank(o, block_offsets, wsum);
    if (o) {
        float *row = X + i * kXRow + cmap_slot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t mm = 3 * j + k;
            if (mm >= K) mm -= K;
            if (mm >= K) mm -= K;
            row[k] = (float)signs[mm];
        }
    }
}
}  // namespace icon

namespace {
int patch_only(icon_work *w, int64_t N, int cmap_slot, hipStream_t st)
{
    const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(icon::k_outlier_patch_self, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, w->d_x, w->d_code8, N, cmap_slot,
                       w->d_block_offsets, w->d_signs, w->d_total);
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}
}  // namespace

extern "C" int icon_query_points(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                                 int prior_type, float sdf_clip, int cmap_mode, const float *h_calib,
                                 const float *d_points, int64_t N, float *d_occ,
                                 int search, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_points && d_occ, "icon_query_points: null argument");
    ICON_ARG(N >= 0, "icon_query_points: negative N");
    int c0 = 0;
    int rc = check_prior(mesh, feat, prior_type, &c0);
    if (rc) return rc;
    ICON_ARG(c0 == mlp->c0, "icon_query_points: MLP input width does not match the feature layout");
    if (N == 0) return ICON_OK;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_work(work, N))) return rc;
    Calib cal;
    static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(cal.m, h_calib ? h_calib : ident, sizeof(cal.m));
    LatticeMap L{};
    mark(work, 0, st);
    if ((rc = launch_features<false>(mesh, feat, prior_type, sdf_clip, cmap_mode, cal, L, d_points, N, search, work, st))) return rc;
    const bool patch = (prior_type == ICON_PRIOR_ICON && cmap_mode == ICON_CMAP_REFERENCE);
    if (patch && (rc = outlier_list(work, N, work->d_signs, st))) return rc;
    mark(work, 1, st);
    if (patch && (rc = patch_only(work, N, feat->dev.csel + 1, st))) return rc;
    mark(work, 2, st);
    rc = mlp_launch(mlp, work->d_x, N, d_occ, precision, st);
    mark(work, 3, st);
    return rc;
}

static int lattice_map(int res, int z0, int z1, LatticeMap *L)
{
    ICON_ARG(res >= 3 && (res & 1) == 1, "lattice resolution must be odd and >= 3 (seg3d_lossless.py:84-86)");
    ICON_ARG(z0 >= 0 && z1 > z0 && z1 <= res, "bad z range");
    L->res = res; L->z0 = z0; L->nz = z1 - z0;
    L->tx = (res + 15) / 16; L->ty = (res + 3) / 4; L->tz = (L->nz + 3) / 4;
    const char *e = getenv("ICON_AMD_XCD_REMAP");
    L->remap = e ? atoi(e) : 0;   // interleaved is balanced; contiguous XCD bands measured 1.6x slower
    return ICON_OK;
}

extern "C" int icon_grid_slab_features(const icon_mesh_t *mesh, const icon_feat_t *feat,
                                       int prior_type, float sdf_clip, int cmap_mode,
                                       int res, int z0, int z1, int8_t *d_signs_local, int64_t *d_count_local,
                                       int search, icon_work_t *work, void *stream)
{
    ICON_ARG(work != nullptr, "icon_grid_slab_features: work is null");
    int c0 = 0;
    int rc = check_prior(mesh, feat, prior_type, &c0);
    if (rc) return rc;
    LatticeMap L;
    if ((rc = lattice_map(res, z0, z1, &L))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)L.nz * res * res;
    if ((rc = ensure_work(work, N))) return rc;
    Calib cal{};
    work->slab_ready = false;
    mark(work, 0, st);
    if ((rc = launch_features<true>(mesh, feat, prior_type, sdf_clip, cmap_mode, cal, L, nullptr, N, search, work, st))) return rc;
    work->slab_needs_patch = (prior_type == ICON_PRIOR_ICON && cmap_mode == ICON_CMAP_REFERENCE);
    work->slab_cmap_slot = feat->dev.csel + 1;
    work->slab_c0 = c0;
    if (work->slab_needs_patch) {
        int8_t *signs = d_signs_local ? d_signs_local : work->d_signs;
        if ((rc = outlier_list(work, N, signs, st))) return rc;
        if (d_count_local)
            ICON_HIP(hipMemcpyAsync(d_count_local, work->d_total, sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    } else if (d_count_local) {
        ICON_HIP(hipMemsetAsync(d_count_local, 0, sizeof(int64_t), st));
    }
    mark(work, 1, st);
    work->slab_res = res; work->slab_z0 = z0; work->slab_z1 = z1; work->slab_ready = true;
    return ICON_OK;
}

extern "C" int icon_grid_slab_finish(const icon_mlp_t *mlp, int res, int z0, int z1,
                                     const int8_t *d_signs_global, int64_t k_total, int64_t rank_offset,
                                     float *d_occ, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_occ, "icon_grid_slab_finish: null argument");
    if (!work->slab_ready || work->slab_res != res || work->slab_z0 != z0 || work->slab_z1 != z1)
        return fail(ICON_ERR_STATE, "icon_grid_slab_finish: no matching icon_grid_slab_features call on this workspace");
    ICON_ARG(work->slab_c0 == mlp->c0, "icon_grid_slab_finish: MLP input width does not match the feature layout");
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)(z1 - z0) * res * res;
    if (work->slab_needs_patch && k_total > 0) {
        ICON_ARG(d_signs_global != nullptr, "icon_grid_slab_finish: sign list is null");
        const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
        hipLaunchKernelGGL(k_outlier_patch, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, work->d_x, work->d_code8, N, work->slab_cmap_slot,
                           work->d_block_offsets, d_signs_global, k_total, rank_offset);
        ICON_HIP(hipGetLastError());
    }
    mark(work, 2, st);
    work->slab_ready = false;
    const int rc = mlp_launch(mlp, work->d_x, N, d_occ, precision, st);
    mark(work, 3, st);
    return rc;
}

extern "C" int icon_grid_slab_finish_gathered(const icon_mlp_t *mlp, int res, int z0, int z1,
                                              const int8_t *d_gathered, int64_t stride, int world, int rank,
                                              float *d_occ, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_occ, "icon_grid_slab_finish_gathered: null argument");
    ICON_ARG(world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world, "icon_grid_slab_finish_gathered: bad world / rank");
    ICON_ARG(stride >= 8 && (stride & 7) == 0, "icon_grid_slab_finish_gathered: stride must be a multiple of 8 (int64 header)");
    if (!work->slab_ready || work->slab_res != res || work->slab_z0 != z0 || work->slab_z1 != z1)
        return fail(ICON_ERR_STATE, "icon_grid_slab_finish_gathered: no matching icon_grid_slab_features call on this workspace");
    ICON_ARG(work->slab_c0 == mlp->c0, "icon_grid_slab_finish_gathered: MLP input width does not match the feature layout");
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)(z1 - z0) * res * res;
    if (work->slab_needs_patch) {
        ICON_ARG(d_gathered != nullptr, "icon_grid_slab_finish_gathered: gathered messages are null");
        const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
        hipLaunchKernelGGL(k_outlier_patch_seg, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, work->d_x, work->d_code8, N, work->slab_cmap_slot,
                           work->d_block_offsets, d_gathered, stride, world, rank);
        ICON_HIP(hipGetLastError());
    }
    mark(work, 2, st);
    work->slab_ready = false;
    const int rc = mlp_launch(mlp, work->d_x, N, d_occ, precision, st);
    mark(work, 3, st);
    return rc;
}

extern "C" int icon_debug_traversal_stats(const icon_mesh_t *mesh, int res, int z0, int z1, uint64_t out[3])
{
    ICON_ARG(mesh && out, "icon_debug_traversal_stats: null argument");
    LatticeMap L;
    int rc = lattice_map(res, z0, z1, &L);
    if (rc) return rc;
    unsigned long long *d = nullptr;
    ICON_HIP(hipMalloc((void **)&d, 4 * sizeof(unsigned long long)));
    ICON_HIP(hipMemset(d, 0, 4 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_traversal_stats, dim3((unsigned)(L.tx * L.ty * L.tz)), dim3(kBlock), 0, 0, mesh->dev, L, d,
                       getenv("ICON_AMD_STATS_SEEDED") ? atoi(getenv("ICON_AMD_STATS_SEEDED")) : 0);
    unsigned long long h[4];
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ICON_ERR_HIP, std::string("traversal stats: ") + hipGetErrorString(e));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    return ICON_OK;
}

extern "C" int icon_grid_eval_slab(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                                   int prior_type, float sdf_clip, int cmap_mode,
                                   int res, int z0, int z1, float *d_occ,
                                   int search, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_occ, "icon_grid_eval_slab: null argument");
    int rc = icon_grid_slab_features(mesh, feat, prior_type, sdf_clip, cmap_mode, res, z0, z1, nullptr, nullptr, search, work, stream);
    if (rc) return rc;
    ICON_ARG(work->slab_c0 == mlp->c0, "icon_grid_eval_slab: MLP input width does not match the feature layout");
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)(z1 - z0) * res * res;
    // the slab's own sign list is the whole list (single call == whole lattice or caller's choice)
    if (work->slab_needs_patch && (rc = patch_only(work, N, work->slab_cmap_slot, st))) return rc;
    mark(work, 2, st);
    work->slab_ready = false;
    rc = mlp_launch(mlp, work->d_x, N, d_occ, precision, st);
    mark(work, 3, st);
    return rc;
}
