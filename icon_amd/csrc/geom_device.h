// geom_device.h - device functions of the per-point geometry / feature half of HGPIFuNet.query, shared by
// query_kernels.hip (the stand-alone kernels) and fused_f16x3.hip (features computed inside the MLP kernel).
//
// Reference being replaced (paths relative to the reference root):
//   lib/dataset/mesh_util.py:357-396 (cal_sdf_batch), :319-354 (barycentrics), :266-277 (feat_select)
//   lib/net/HGPIFuNet.py:268-365 (query), lib/net/geometry.py:21-61 (index, orthogonal)
//
// Arithmetic spec: float32, the only fused operations are the explicit fmaf() calls, IEEE
// division / sqrt.  Every translation unit that includes this header MUST be compiled with
// -ffp-contract=off (the Makefile's EXACT flags); the pragma below is a second line of defence.
#pragma once
#pragma clang fp contract(off)

#include "common.h"
#include "mesh_rules.h"

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
//   (s,t) = barycentrics of the plane projection; inside -> |(p - a) - s ab - t ac|^2,
//   else min over the three segments of |(p - origin) - clamp(t,0,1) * edge|^2  (fma chains on the rounded
//   p - origin: one fma per axis and term, and the rounding happens at the triangle's scale).
struct TriC {   // TriPre fields as values (SGPRs when read through the constant address space)
    f3 a, b, ab, ac, bc;
    float i00, i11, ibc, a00, a01, a11, inn;
};

__device__ __forceinline__ float seg_dist2(f3 po, f3 e, float dot_e_po, float inv_len2)
{
    // po = p - origin; residual (p - o) - t e as one fma per axis (spec S2)
    const float t = fminf(fmaxf(dot_e_po * inv_len2, 0.0f), 1.0f);
    const f3 d = mk3(fmaf(-t, e.x, po.x), fmaf(-t, e.y, po.y), fmaf(-t, e.z, po.z));
    return dot3(d, d);
}

__device__ __forceinline__ float tri_dist2(f3 p, const TriC &t)
{
    const f3 ap = sub3(p, t.a), bp = sub3(p, t.b);
    const float d1 = dot3(t.ab, ap), d2 = dot3(t.ac, ap), d3 = dot3(t.bc, bp);
    const float s = fmaf(t.a11, d1, -(t.a01 * d2)) * t.inn;
    const float u = fmaf(t.a00, d2, -(t.a01 * d1)) * t.inn;
    const bool inside = (s >= 0.0f) & (u >= 0.0f) & (s + u <= 1.0f);
    const f3 df = mk3(fmaf(-u, t.ac.x, fmaf(-s, t.ab.x, ap.x)), fmaf(-u, t.ac.y, fmaf(-s, t.ab.y, ap.y)),
                      fmaf(-u, t.ac.z, fmaf(-s, t.ab.z, ap.z)));
    float d_face = dot3(df, df);
    const float e0 = seg_dist2(ap, t.ab, d1, t.i00);
    const float e1 = seg_dist2(ap, t.ac, d2, t.i11);
    const float e2 = seg_dist2(bp, t.bc, d3, t.ibc);
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
__device__ __forceinline__ f2 seg_dist2x2(f3x2 po, f3x2 e, f2 dot_e_po, f2 inv_len2)
{
    // t = clamp(dot * inv, 0, 1): the packed multiply carries the clamp itself (the compiler emits a v_max ... clamp
    // per component after it); NaN -> 0 like fminf(fmaxf(x, 0), 1)
    f2 t;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(t) : "v"(dot_e_po), "s"(inv_len2));
    const f2 nt = -t;
    f3x2 d; d.x = fma2(nt, e.x, po.x); d.y = fma2(nt, e.y, po.y); d.z = fma2(nt, e.z, po.z);
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
    const bool in0 = (s.x >= 0.0f) & (u.x >= 0.0f) & (su.x <= 1.0f);
    const bool in1 = (s.y >= 0.0f) & (u.y >= 0.0f) & (su.y <= 1.0f);
    const f2 e0 = seg_dist2x2(ap, ab, d1, i00);
    const f2 e1 = seg_dist2x2(ap, ac, d2, i11);
    const f2 e2 = seg_dist2x2(bp, bc, d3, ibc);
    f2 r; r.x = fminf(fminf(e0.x, e1.x), e2.x); r.y = fminf(fminf(e0.y, e1.y), e2.y);
    // the face distance is selected only where the projection falls inside the triangle: for most of a packet's
    // candidates no lane's does, and the wave skips it and the selection (same selected bits either way)
    if (__any(in0 | in1)) {
        const f2 ns = -s, nu = -u;
        f3x2 df; df.x = fma2(nu, ac.x, fma2(ns, ab.x, ap.x)); df.y = fma2(nu, ac.y, fma2(ns, ab.y, ap.y)); df.z = fma2(nu, ac.z, fma2(ns, ab.z, ap.z));
        const f2 d_face = dot3x2(df, df);
        float f0 = d_face.x, f1 = d_face.y, g0 = r.x, g1 = r.y;
        asm volatile("" : "+v"(f0), "+v"(f1), "+v"(g0), "+v"(g1));     // keep the choice a v_cndmask
        r.x = in0 ? f0 : g0; r.y = in1 ? f1 : g1;
    }
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

// box distance^2 beyond which |sdf| = d / sqrt(3) >= sdf_clip for certain (d >= box distance; the computed d^2 is
// within 1e-6 relative of the true one): (clip sqrt(3))^2 with a cushion.  Negative clips: every point off the box.
__host__ __device__ inline float far_box_dist2(float sdf_clip)
{
    const float cb = (sdf_clip > 0.0f ? sdf_clip : 0.0f) * 1.7320508f;
    return cb * cb * 1.0001f + 1e-6f;
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

// the BVH root (a node id, or a leaf code for meshes of at most kLeafMax triangles): decided by the device build, read
// through a wave-uniform scalar load
__device__ __forceinline__ int mesh_root(const MeshDev &m)
{
    return __builtin_amdgcn_readfirstlane(*reinterpret_cast<__attribute__((address_space(4))) const int *>((uintptr_t)&m.dyn->root));
}

// Pruning bound: a subtree may be skipped only if no triangle in it can tie or beat `best`.
// Computed distances carry < 1e-6 absolute error (coordinates are O(1)), so the bound must be at least
// (sqrt(best) + 4e-6)^2 (1 + 1e-6); see DESIGN.md "BVH conservativeness".  Evaluated without the square
// root (an IEEE sqrtf is ~25 instructions, once per visited leaf): 2 sqrt(b) <= b / c + c for any c > 0, with
// c = 0.05: (sqrt(b) + e)^2 <= b (1 + 20 e) + 0.05 e + e^2 - looser by < 1e-4 relative, which changes nothing
// measurable in the candidate sets and nothing at all in the result.
__device__ __forceinline__ float prune_threshold(float best)
{
    return fmaf(best, 1.000083f, 2.1e-7f);
}

// BVH2 PACKET traversal: the 64 lanes of a wavefront descend the tree TOGETHER.  Control flow, the
// node / leaf addresses and the stack are wave-uniform (scalar registers, scalar loads, one LDS
// word per stack entry per wave); every lane tests its own point against the shared node boxes
// and triangles, and a subtree is entered when ANY lane still needs it.  In lattice mode a
// wavefront owns a 4x4x4 block of lattice points, so the lanes' candidate sets nearly coincide.
// A leaf holds up to 4 TriPre records (96 B each, scalar loads); the distance test has no per-lane branch (one
// wave-uniform skip of the face term).  Near child first, ordered by the block's centre lane.
// `live` = false parks a padding lane: it never votes and its result is discarded.
// TIES: also track the runner-up - the smallest key that differs from the winner's (a padding copy of a short leaf
// carries the key of its original and is therefore never its own runner-up).  Every face whose d^2 lies within
// the pruning bound of the final winner is visited (the bound only ever shrinks towards its final value), so the
// runner-up is exact whenever it is within ~8e-5 relative of the winner - far beyond the 255 ulps reported.
template <bool STATS = false, bool TIES = false>
__device__ __forceinline__ Nearest nearest_packet(const MeshDev &m, f3 p, bool live, int *wstack /* LDS, kStackDepth ints of this wave */,
                                                  int *n_nodes = nullptr, int *n_tris = nullptr, float thr0 = INFINITY,
                                                  unsigned long long *runner_up = nullptr, int center_lane = 21,
                                                  bool have_root = false, int root = 0)
{
    Nearest nr; nr.d2 = INFINITY; nr.slot = 0; nr.face = 0x7fffffff;
    unsigned long long key = 0x7f8000007fffffffull;   // (+inf, INT_MAX)
    unsigned long long key2 = 0x7f8000007fffffffull;
    float thr = live ? thr0 : -INFINITY;
    int sp = 0;
    int cur = have_root ? root : mesh_root(m);        // (wave-uniform choice; the caller that holds the root in a register passes it)
    while (true) {
        if (cur < 0) {
            const int code = ~cur;
            const int leaf = code >> 2, cnt = (code & 3) + 1;
            if (STATS) *n_tris += cnt;
            const int npairs = __builtin_amdgcn_readfirstlane((cnt + 1) >> 1);      // 1 or 2, wave-uniform (scalar loop counter)
            for (int pr = 0; pr < npairs; ++pr) {
                cf2 *q = reinterpret_cast<cf2 *>(as_const(&m.leaves[leaf].pair[pr]));
                const f2 d2 = tri_dist2_pair(p, q);
                const f2 fc = q[22];
                // S3 as ONE unsigned 64-bit minimum: key = (bits of d^2) << 32 | face.  d^2 >= +0, so
                // its bit pattern orders like the value; equal d^2 -> lower face id wins; NaN (bits
                // above +inf) never wins; a padding copy has the same key as its original.
                const unsigned long long k0 = ((unsigned long long)(unsigned)__float_as_int(d2.x) << 32) | (unsigned)__float_as_int(fc.x);
                const unsigned long long k1 = ((unsigned long long)(unsigned)__float_as_int(d2.y) << 32) | (unsigned)__float_as_int(fc.y);
                // (the winner's slot is looked up from its face id at the end: nothing else to carry per test; a parked
                //  lane computes on a clamped copy of a real point and may update its key freely - it never votes
                //  (thr = -inf) and its result is not stored - so the update needs no `live` mask)
                if (TIES) {
                    if (k0 < key) { key2 = key; key = k0; } else if (k0 != key && k0 < key2) key2 = k0;
                    if (k1 < key) { key2 = key; key = k1; } else if (k1 != key && k1 < key2) key2 = k1;
                } else {
                    key = (k0 < key) ? k0 : key;
                    key = (k1 < key) ? k1 : key;
                }
            }
            nr.d2 = __int_as_float((int)(key >> 32));
            thr = live ? prune_threshold(nr.d2) : thr;           // one fma + select: cheaper than finding out whether the key moved
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
                const int e0 = __builtin_amdgcn_readlane(__float_as_int(d0), center_lane);
                const int e1 = __builtin_amdgcn_readlane(__float_as_int(d1), center_lane);
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
    nr.d2 = __int_as_float((int)(key >> 32)); nr.face = (int)(key & 0xffffffffu);
    nr.slot = (nr.face != 0x7fffffff) ? m.face2slot[nr.face] : 0;
    if (TIES) *runner_up = key2;
    return nr;
}

// ONE packet searched by the NW wavefronts of a workgroup.  The coarse lattices of the reference's schedule hold fewer packets
// than the GPU has wave slots, so a launch lasts as long as its LONGEST walk - ~1,100 - 2,500 dependent steps, five times the
// mean, and a perfect starting bound shortens it by 10 % only: the walk is verification, not search.  Every wave holds the
// same 64 points and the waves share the walk NODE BY NODE (handing each wave one of the 32 subtrees five levels down was
// measured first: no gain - a packet inside the body finds nearly all of its walk in one or two of them).
// Every wave runs the depth-first walk on its own LDS stack; when it has to postpone
// a far child and the workgroup's shared queue is short (fewer entries than waves), the child goes to the queue instead, and
// a wave whose stack runs empty takes its next subtree from there.  Queue: a ring with tickets (tail / head by LDS atomics),
// a count of completed pushes (avail) and a ready value per slot; `active` counts the waves that may still push - a wave
// leaves when it reads avail == 0 and THEN active == 0 (a pop attempt counts as active, so a transiently negative `avail`
// cannot hide an entry from every wave at once).  The waves share the pruning bound per lane (an LDS minimum of the d^2 bit
// patterns after every leaf): a subtree is skipped when it cannot tie or beat the best distance ANY wave has found - the
// triangle that set it stays with that wave, and the keys are merged at the end, so the result is the minimum of (d^2, face)
// over a superset of the faces within the final bound: the same key as nearest_packet, bit for bit.
// All NW waves must call this together (workgroup barriers).  The result is valid in wave 0 only.
constexpr int kRing = 64, kRingEmpty = 0x7fffffff;
// EVERY wait of the hand-over is bounded (round 5).  The two waits below - a popper for the push that owns its ticket, a pusher
// for the popper a full turn of the ring behind to clear its slot - are waits for a wave that is itself never blocked, so they
// last nanoseconds; but an unbounded spin turns any bug or lost store into a GPU hang that ends in SIGKILL with no message (the
// one unexplained fatal signal of round 4).  On expiry the wave records WHAT it was waiting for in the workspace's error word
// (host-mapped memory, written on errors only: code, workgroup, ticket, head / tail at that moment), gives the wait up and the
// walk still terminates - its result is then suspect, and the host reports ICON_ERR_STATE at its next status read
// (icon_work_status; icon_adaptive_eval / icon_grid_eval_slab / icon_query_points check the word without synchronising).
constexpr int kShareSpinLog2 = 18;            // 2^18 polls of ~0.2 us: ~50 ms, five orders of magnitude above a real hand-over
constexpr int kShareErrPop = 1;               // a popper's ticket was never filled (lost / abandoned push)
constexpr int kShareErrPush = 2;              // a pusher's slot was never cleared (ring a full turn behind)
constexpr int kShareErrIdle = 3;              // the idle loop of a wave expired while other waves were still active
// test-only behaviour, all zero in production (icon_debug_set_option "share_ring" / "share_lose_push" / "share_spin_log2"):
//   ring     - forced ring size (power of two >= 2): a ring smaller than 2 NW makes pushers really wait for poppers
//   lose     - the push with ticket `lose` (>= 1) of every packet claims its ticket and announces it, but never stores the node
//   spin_log2 - wait bound 2^spin_log2 polls
struct ShareLds {
    unsigned thr[64];
    int head, tail, avail, active;
    int mask, lose, spins, pad1;
    int pad[8];
    int ring[kRing];
    unsigned long long keys[1];           // [NW][64]
};
__host__ __device__ constexpr size_t share_lds_bytes(int nw) { return sizeof(ShareLds) + (size_t)nw * 64 * 8; }

// first error wins (system scope: the record lives in host memory); lane 0 of the reporting wave only.  The host reads the
// record without synchronising (work_check_err): the winner first CLAIMS word 0 with kShareErrClaimed - which the host treats
// as "nothing reported yet" -, writes the details, fences, and only then stores the code, so that a code is never seen
// ahead of its details
constexpr int kShareErrClaimed = -1;
__device__ __forceinline__ void share_report(const ShareDbg &dbg, int code, int ticket, const ShareLds &S)
{
    if (!dbg.err) return;
    int expected = 0;
    if (__hip_atomic_compare_exchange_strong(dbg.err, &expected, kShareErrClaimed, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
        dbg.err[1] = (int)blockIdx.x; dbg.err[2] = ticket;
        dbg.err[3] = *(volatile const int *)&S.head; dbg.err[4] = *(volatile const int *)&S.tail;
        dbg.err[5] = *(volatile const int *)&S.avail; dbg.err[6] = *(volatile const int *)&S.active;
        dbg.err[7] = (int)(threadIdx.x >> 6);
        __threadfence_system();
        __hip_atomic_store(dbg.err, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ int share_pop(ShareLds &S, int lane, const ShareDbg &dbg)
{
    int t = 0;
    if (lane == 0) t = atomicSub(&S.avail, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t <= 0) { if (lane == 0) atomicAdd(&S.avail, 1); return kRingEmpty; }
    int i = 0;
    if (lane == 0) i = atomicAdd(&S.head, 1);
    i = __builtin_amdgcn_readfirstlane(i);
    volatile int *slot = &S.ring[i & S.mask];
    int v = __builtin_amdgcn_readfirstlane(*slot);
    for (int spins = 0; v == kRingEmpty; ++spins) {               // the push that owns this ticket is completing
        if (spins >= S.spins) {                                  // ... or was lost: give the ticket up (the walk misses a subtree: reported)
            if (lane == 0) share_report(dbg, kShareErrPop, i, S);
            return kRingEmpty;
        }
        __builtin_amdgcn_s_sleep(1);
        v = __builtin_amdgcn_readfirstlane(*slot);
    }
    if (lane == 0) *slot = kRingEmpty;
    return v;
}
// true: the node is in the queue; false: it is not (the caller keeps it on its own stack - nothing is lost)
__device__ __forceinline__ bool share_push(ShareLds &S, int lane, int node, const ShareDbg &dbg)
{
    int ok = 1;
    if (lane == 0) {
        const int i = atomicAdd(&S.tail, 1);
        if (i == S.lose) {                                       // test only: the hand-over a popper will wait for in vain
            atomicAdd(&S.avail, 1);
        } else {
            int spins = 0;
            while (atomicCAS(&S.ring[i & S.mask], kRingEmpty, node) != kRingEmpty) {      // the popper a full turn behind has not cleared its slot yet
                if (++spins >= S.spins) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (ok) atomicAdd(&S.avail, 1);
            else share_report(dbg, kShareErrPush, i, S);         // the ticket stays unfilled: its popper reports and gives up as well
        }
    }
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

template <int NW>
__device__ __forceinline__ Nearest nearest_shared(const MeshDev &m, f3 p, bool live, int *wstack, char *smem /* share_lds_bytes(NW) */, int center_lane,
                                                  const ShareDbg &dbg)
{
    ShareLds &S = *reinterpret_cast<ShareLds *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                             // (the previous packet's keys have been read)
    if (threadIdx.x < 64) { S.thr[lane] = 0x7f800000u; S.ring[lane] = lane == 0 ? mesh_root(m) : kRingEmpty; }
    if (threadIdx.x == 0) {
        S.head = 0; S.tail = 1; S.avail = 1; S.active = NW;
        S.mask = (dbg.ring >= 2 && dbg.ring <= kRing ? dbg.ring : kRing) - 1;
        S.lose = dbg.lose > 0 ? dbg.lose : -1;
        S.spins = 1 << (dbg.spin_log2 > 0 && dbg.spin_log2 < 30 ? dbg.spin_log2 : kShareSpinLog2);
    }
    __syncthreads();
    unsigned long long key = 0x7f8000007fffffffull;
    float thr = live ? INFINITY : -INFINITY;
    int sp = 0;
    while (true) {
        int cur;
        if (sp > 0) cur = __builtin_amdgcn_readfirstlane(wstack[--sp]);
        else {
            cur = share_pop(S, lane, dbg);
            if (cur == kRingEmpty) {                             // nothing to take: idle until an entry appears or every wave is idle
                if (lane == 0) atomicSub(&S.active, 1);
                int spins = 0;
                while (true) {
                    const int av = __builtin_amdgcn_readfirstlane(*(volatile int *)&S.avail);
                    const int ac = __builtin_amdgcn_readfirstlane(*(volatile int *)&S.active);
                    if (av > 0) {
                        if (lane == 0) atomicAdd(&S.active, 1);
                        cur = share_pop(S, lane, dbg);
                        if (cur != kRingEmpty) break;
                        if (lane == 0) atomicSub(&S.active, 1);
                    } else if (ac <= 0) break;
                    // leaving early costs parallelism, not correctness: an idle wave holds no work.  A single walk is ~1 ms at most,
                    // so an idle wave that outlasts the bound has seen a workgroup that stopped making progress: reported
                    // (not scaled by the test switch: with a wait bound of a few hundred polls an idle wave legitimately outlasts it)
                    if (++spins > (4 << kShareSpinLog2)) { if (lane == 0) share_report(dbg, kShareErrIdle, spins, S); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (cur == kRingEmpty) break;
            }
            thr = live ? prune_threshold(__uint_as_float(S.thr[lane])) : thr;
        }
        while (true) {
            if (cur < 0) {
                const int code = ~cur;
                const int leaf = code >> 2, cnt = (code & 3) + 1;
                const int npairs = __builtin_amdgcn_readfirstlane((cnt + 1) >> 1);
                for (int pr = 0; pr < npairs; ++pr) {
                    cf2 *q = reinterpret_cast<cf2 *>(as_const(&m.leaves[leaf].pair[pr]));
                    const f2 d2 = tri_dist2_pair(p, q);
                    const f2 fc = q[22];
                    const unsigned long long k0 = ((unsigned long long)(unsigned)__float_as_int(d2.x) << 32) | (unsigned)__float_as_int(fc.x);
                    const unsigned long long k1 = ((unsigned long long)(unsigned)__float_as_int(d2.y) << 32) | (unsigned)__float_as_int(fc.y);
                    key = (k0 < key) ? k0 : key;
                    key = (k1 < key) ? k1 : key;
                }
                const unsigned mine = (unsigned)(key >> 32);
                const unsigned old = atomicMin(&S.thr[lane], mine);            // d^2 >= +0: the bit patterns order like the values
                thr = live ? prune_threshold(__uint_as_float(min(old, mine))) : thr;
                break;
            }
            cf2 *q = reinterpret_cast<cf2 *>(as_const(m.nodes + cur));
            const f2 dd = box_dist2_pair(q, p);
            const float d0 = dd.x, d1 = dd.y;
            const f2 ids = q[6];
            const int c0 = __float_as_int(ids.x), c1 = __float_as_int(ids.y);
            const bool v0 = __any(d0 <= thr), v1 = __any(d1 <= thr);
            if (v0 && v1) {
                const int e0 = __builtin_amdgcn_readlane(__float_as_int(d0), center_lane);
                const int e1 = __builtin_amdgcn_readlane(__float_as_int(d1), center_lane);
                const bool first0 = e0 <= e1;
                const int far = first0 ? c1 : c0;
                const int av = __builtin_amdgcn_readfirstlane(*(volatile int *)&S.avail);
                if (av >= NW || !share_push(S, lane, far, dbg)) wstack[sp++] = far;
                cur = first0 ? c0 : c1;
            } else if (v0) cur = c0;
            else if (v1) cur = c1;
            else break;
        }
    }
    S.keys[wave * 64 + lane] = key;
    __syncthreads();
    Nearest nr; nr.d2 = INFINITY; nr.slot = 0; nr.face = 0x7fffffff;
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < NW; ++w) { const unsigned long long k = S.keys[w * 64 + lane]; key = (k < key) ? k : key; }
        nr.d2 = __int_as_float((int)(key >> 32)); nr.face = (int)(key & 0xffffffffu);
        nr.slot = (nr.face != 0x7fffffff) ? m.face2slot[nr.face] : 0;
    }
    return nr;
}

// Diagnostics - the ALTERNATIVE tie rule: among the faces whose d^2 lies within `ulps` float32 ulps of the minimum
// `best_bits` (found by a first nearest_packet pass), the one with the HIGHEST face index, together with its own
// d^2.  ulps = 0 is "highest index on exact ties" - the mirror image of S3; ulps >= 1 stands for any other
// correctly-rounding implementation of the same point-triangle distance (kaolin's), whose last bits decide between
// the mathematically tied triangles around a vertex or an edge differently.  Same boxes, same S2 distance, same
// pruning bound (now constant) as nearest_packet.
__device__ __forceinline__ Nearest nearest_packet_alt(const MeshDev &m, f3 p, bool live, int *wstack, uint32_t best_bits, uint32_t ulps)
{
    const float thr = live ? prune_threshold(__int_as_float((int)best_bits)) : -INFINITY;
    const uint32_t lim = best_bits + ulps;                 // d^2 >= 0: bit patterns order like the values
    int face = -1;
    uint32_t fbits = best_bits;
    int sp = 0, cur = mesh_root(m);
    while (true) {
        if (cur < 0) {
            const int code = ~cur;
            const int leaf = code >> 2, cnt = (code & 3) + 1;
            const int npairs = __builtin_amdgcn_readfirstlane((cnt + 1) >> 1);
            for (int pr = 0; pr < npairs; ++pr) {
                cf2 *q = reinterpret_cast<cf2 *>(as_const(&m.leaves[leaf].pair[pr]));
                const f2 d2 = tri_dist2_pair(p, q);
                const f2 fc = q[22];
                const uint32_t b0 = (uint32_t)__float_as_int(d2.x), b1 = (uint32_t)__float_as_int(d2.y);
                const int f0 = __float_as_int(fc.x), f1 = __float_as_int(fc.y);
                if (b0 <= lim && f0 > face) { face = f0; fbits = b0; }
                if (b1 <= lim && f1 > face) { face = f1; fbits = b1; }
            }
            if (sp == 0) break;
            cur = __builtin_amdgcn_readfirstlane(wstack[--sp]);
        } else {
            cf2 *q = reinterpret_cast<cf2 *>(as_const(m.nodes + cur));
            const f2 dd = box_dist2_pair(q, p);
            const f2 ids = q[6];
            const int c0 = __float_as_int(ids.x), c1 = __float_as_int(ids.y);
            const bool v0 = __any(dd.x <= thr), v1 = __any(dd.y <= thr);
            if (v0 && v1) { wstack[sp++] = c1; cur = c0; }
            else if (v0) cur = c0;
            else if (v1) cur = c1;
            else {
                if (sp == 0) break;
                cur = __builtin_amdgcn_readfirstlane(wstack[--sp]);
            }
        }
    }
    Nearest nr;
    nr.face = face >= 0 ? face : 0x7fffffff;
    nr.d2 = __int_as_float((int)fbits);
    nr.slot = face >= 0 ? m.face2slot[face] : 0;
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
// (measured crossover on MI355X in round 2: 60k points 0.42 vs 0.67 ms, 200k ~1.4 vs 0.70 ms)
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
        out_slot = leaf + t;                                  // slots are positions: a leaf's id is its first slot
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
    const int root = mesh_root(m);
    int cur = root;
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
    int nf = root >= 0 ? 1 : 0, nl = 0;           // uniform counters (a root that is a leaf has been tested already)
    if (lane == 0) S.frontier[0] = root;
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

// Brute force over all triangle slots, staged through LDS in tiles (validation path).  The S2 constants of a slot are
// formed here from its corners by the builder's own tri_setup (mesh_rules.h): the same operations, the same bits as the
// LeafRec the packet traversal reads.
constexpr int kBruteTile = 128;   // 128 x 96 B = 12 KiB of LDS
template <int BLOCK>
__device__ __forceinline__ Nearest nearest_brute(const MeshDev &m, f3 p, float *tile /* LDS, kBruteTile*24 floats */)
{
    Nearest nr; nr.d2 = INFINITY; nr.slot = 0; nr.face = 0x7fffffff;
    for (int base = 0; base < m.n_tris; base += kBruteTile) {
        __syncthreads();
        const int n = min(kBruteTile, m.n_tris - base);
        for (int k = threadIdx.x; k < n; k += BLOCK) {
            const TriRec &tr = m.tris[base + k];
            TriPre pre;
            tri_setup(tr.a, tr.b, tr.c, m.slot2face[base + k], pre);
            const float *src = reinterpret_cast<const float *>(&pre);
            for (int fld = 0; fld < 24; ++fld) tile[k * 24 + fld] = src[fld];
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

__device__ __forceinline__ bool inside_brute(const MeshDev &m, f3 p);

__device__ __forceinline__ bool inside_bins(const MeshDev &m, f3 p)
{
    const MeshDyn &d = *m.dyn;
    if (d.gy == 0) return inside_brute(m, p);      // the bin lists did not fit their buffer (kMeshBinOverflow)
    if (!(p.y >= d.bin_y0 && p.y <= d.bin_y1 && p.z >= d.bin_z0 && p.z <= d.bin_z1)) return false;
    const int cy = bin_cell(p.y, d.bin_y0, d.bin_inv_y, d.gy);
    const int cz = bin_cell(p.z, d.bin_z0, d.bin_inv_z, d.gz);
    const int cell = cz * d.gy + cy;
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
        cnt += ray_hit(p, a, b, c, ia, ib, ic);
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

__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    // bijective "contiguous chunk per XCD" remap (blocks are dispatched round-robin to 8 XCDs)
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// L.trim: the host tiled the whole slab; leave out every face of the cube that is farther from the body's bounding box
// than the clip band is wide - those points are "outside" outliers without looking at the mesh (k_sign's own far
// test) - and re-tile.  Wave-uniform scalar arithmetic on the box in MeshDyn.
__device__ __forceinline__ LatticeMap lattice_trim(LatticeMap L, const MeshDev &m)
{
    if (!L.trim) return L;
    const MeshDyn &d = *m.dyn;
    const int res = L.res, z0 = L.z0, z1 = L.z0 + L.nz;
    int lo[3] = {0, 0, 0}, hi[3] = {res, res, res};
    const float need = L.trim_need;
    // lattice_world: x = -1 at ix = 0, +1 at ix = res-1; y = +1 at iy = 0, -1 at iy = res-1; z like x
    if (d.box_lo[0] + 1.0f > need) lo[0] = 1;
    if (1.0f - d.box_hi[0] > need) hi[0] = res - 1;
    if (1.0f - d.box_hi[1] > need) lo[1] = 1;
    if (d.box_lo[1] + 1.0f > need) hi[1] = res - 1;
    if (d.box_lo[2] + 1.0f > need) lo[2] = 1;
    if (1.0f - d.box_hi[2] > need) hi[2] = res - 1;
    L.sx0 = lo[0]; L.sx1 = hi[0]; L.sy0 = lo[1]; L.sy1 = hi[1];
    L.sz0 = max(z0, lo[2]) - z0; L.sz1 = min(z1, hi[2]) - z0;
    if (L.sz1 < L.sz0) L.sz1 = L.sz0;
    const int P = L.pk ? L.pk : 4;
    L.tx = (L.sx1 - L.sx0 + 4 * P - 1) / (4 * P); L.ty = (L.sy1 - L.sy0 + P - 1) / P; L.tz = (L.sz1 - L.sz0 + P - 1) / P;
    L.trim = 0;
    return L;
}

// tile t of the (trimmed) tiling, packet `wave` (0..3) of the tile, lane of the packet
__device__ __forceinline__ bool lattice_point_at(const LatticeMap &L, int t, int wave, int lane, int &ix, int &iy, int &iz)
{
    const int nb = L.tx * L.ty * L.tz;
    if (t >= nb) { ix = iy = iz = 0x3fffffff; return false; }     // a workgroup beyond the (trimmed) tiling
    if (L.remap == 1) t = xcd_remap(t, nb);
    else if (L.remap == 2) {
        // the tx workgroups of one (y,z) block row on ONE XCD (block b runs on XCD b % 8): the row's results are
        // contiguous in memory, and the 64-byte pieces of a 128-byte line written by x-neighbours then meet in the
        // same L2 instead of leaving two XCDs as partial lines.  Rows still alternate over the XCDs (balance).
        const int full = (nb / (8 * L.tx)) * (8 * L.tx);
        if (t < full) {
            const int r8 = t & 7, k = t >> 3;
            t = ((k / L.tx) * 8 + r8) * L.tx + (k % L.tx);
        }
    }
    const int bty = t / (L.tz * L.tx);
    const int rem = t - bty * (L.tz * L.tx);
    const int btz = rem / L.tx;
    int btx = rem - btz * L.tx;
    // Block b runs on XCD b % 8.  When the number of blocks per x-row is a multiple of 8 (16 for the 255-point interior
    // rows of a 257^3 lattice) every XCD would own fixed x-columns of the volume - the far-field columns cost more than
    // the ones through the body, and the launch waits for the slowest XCD (measured 3.57 vs 3.00 ms).  Rotating the
    // x-position by the row number keeps the mapping a bijection and hands every XCD every column in turn.
    if (L.remap == 0) btx = (btx + btz + bty * L.tz) % L.tx;
    // tiles cover the search region [sx0, sx1) x [sy0, sy1) x planes [sz0, sz1) (iz is relative to the slab).  A wavefront owns
    // a P x P x P block of points, P = L.pk: 4 on the fine lattices; on a COARSE lattice (the first levels of the reference's
    // schedule: spacing 2-8x the triangle size) the 64 points of a 4^3 block have little of their search in common and the
    // packet walks the union - 2^3 or single points per wavefront there (the lanes beyond P^3 are parked on point 0 of the block)
    const int P = L.pk ? L.pk : 4;
    const bool used = lane < P * P * P;
    const int l = used ? lane : 0;
    ix = L.sx0 + (btx * 4 + wave) * P + l % P;
    iy = L.sy0 + bty * P + (l / P) % P;
    iz = L.sz0 + btz * P + l / (P * P);
    return used && ix < L.sx1 && iy < L.sy1 && iz < L.sz1;
}
__device__ __forceinline__ bool lattice_point(const LatticeMap &L, int &ix, int &iy, int &iz)
{
    return lattice_point_at(L, (int)blockIdx.x, threadIdx.x >> 6, threadIdx.x & 63, ix, iy, iz);
}
// The per-packet set-up of the throughput launch (k_nearest<LATTICE>, 4^3 packets, default block order), precomputed ONCE per call.
// Round 4 moved the trim into the kernel (the body's box is known on the device only): every one of the 262,144 packets of a
// 257^3 call then re-derived the trimmed tiling from MeshDyn and decoded its tile with six integer divisions by run-time values
// - the ISA has no integer division, each is a ~25-instruction float-reciprocal sequence on the VECTOR unit even for wave-uniform
// operands - plus three per-lane ones by the run-time packet size: ~350 instructions per packet, 6 % of a kernel that is VALU-
// issue-bound, and 82 more than round 3 (profiles/r04_pmc_lds.txt: +1.5 % SQ_INSTS_VALU, +4.1 % cycles).  Now the first thread
// of the row-crossings kernel - which runs before the search on the same stream anyway - writes this record (the trimmed region,
// the tile counts and multiply-high reciprocals of the two divisors), and a packet's set-up is one s_load_dwordx16, three
// s_mul_hi_u32 and shifts.
struct LatticeFast {
    int32_t sx0, sy0, sz0, sx1, sy1, sz1;     // the trimmed search region (lattice_trim), z relative to the slab
    int32_t tx, tz, txtz, nb;                 // tiles per x-row, tile planes, tx * tz, tiles of the region
    uint32_t m_tx, m_txtz;                    // udiv_magic of tx and tx * tz
    int32_t root;                             // MeshDyn::root (saves the packet a dependent scalar load through m.dyn)
    int32_t pad[3];
    // followed by float coord[res]: lattice_world's x / z coordinate of index i (y = -coord[i], bit for bit: 2f - 1 and
    // -2f + 1 round alike) - three IEEE divisions by res - 1 per lane (~45 instructions) become three cached loads
};
static_assert(sizeof(LatticeFast) == 64, "LatticeFast layout");
constexpr int kLatticeFastMaxRes = 2048;      // (res^3 < 2^31 keeps res <= 1290)
constexpr size_t kLatticeFastBytes = sizeof(LatticeFast) + kLatticeFastMaxRes * sizeof(float);
// floor(n / d) for 32-bit n as one multiply-high: m = floor(2^32 / d) undershoots n / d by less than n / 2^32 < 1, i.e. the
// quotient is right or one too small - one compare-and-increment repairs it for EVERY n (no range assumption, no branch);
// d == 1: m = 2^32 - 1 gives n - 1, repaired the same way.
__host__ __device__ inline uint32_t udiv_magic(uint32_t d) { return d > 1 ? (uint32_t)(0x100000000ull / d) : 0xffffffffu; }
__device__ __forceinline__ uint32_t udiv_fast(uint32_t n, uint32_t d, uint32_t m)
{
    const uint32_t q = __umulhi(n, m);
    return q + ((n - q * d >= d) ? 1u : 0u);
}
__device__ __forceinline__ void lattice_fast_write(LatticeFast *out, LatticeMap L, const MeshDev &m)
{
    L = lattice_trim(L, m);
    LatticeFast F;
    F.sx0 = L.sx0; F.sy0 = L.sy0; F.sz0 = L.sz0; F.sx1 = L.sx1; F.sy1 = L.sy1; F.sz1 = L.sz1;
    F.tx = L.tx; F.tz = L.tz; F.txtz = L.tx * L.tz; F.nb = L.tx * L.ty * L.tz;
    F.m_tx = udiv_magic((uint32_t)max(L.tx, 1)); F.m_txtz = udiv_magic((uint32_t)max(L.tx * L.tz, 1));
    F.root = m.dyn->root;
    F.pad[0] = F.pad[1] = F.pad[2] = 0;
    *out = F;
}
// the coordinate table behind the record: thread i of the launch writes entry i (the launch has at least res threads)
__device__ __forceinline__ void lattice_fast_coords(LatticeFast *out, int res, int64_t i)
{
    if (i < res) {
        const float f = (float)(int)i / (float)(res - 1);
        reinterpret_cast<float *>(out + 1)[i] = f * 2.0f + (-1.0f);       // lattice_world's expression for x and z
    }
}
__device__ __forceinline__ f3 lattice_world_fast(const LatticeFast *lf, int ix, int iy, int iz)
{
    const float *c = reinterpret_cast<const float *>(lf + 1);
    return mk3(c[ix], -c[iy], c[iz]);
}
// the packet of workgroup t / wave / lane under the record: the same point lattice_point_at(lattice_trim(L), ...) names for
// pk = 4, remap = 0 (test_lattice_fast_setup_names_the_same_points runs both over whole tilings)
__device__ __forceinline__ LatticeFast lattice_fast_load(const LatticeFast *lf)
{
    typedef __attribute__((address_space(4))) const int32_t *cint;
    const cint q = (cint)(uintptr_t)lf;                          // wave-uniform constant-address loads: merged into wide s_loads
    LatticeFast F;
    F.sx0 = q[0]; F.sy0 = q[1]; F.sz0 = q[2]; F.sx1 = q[3]; F.sy1 = q[4]; F.sz1 = q[5];
    F.tx = q[6]; F.tz = q[7]; F.txtz = q[8]; F.nb = q[9];
    F.m_tx = (uint32_t)q[10]; F.m_txtz = (uint32_t)q[11]; F.root = q[12];
    return F;
}
__device__ __forceinline__ bool lattice_point_fast(const LatticeFast &F, uint32_t t, int wave, int lane, int &cx, int &cy, int &cz)
{
    const uint32_t tx = (uint32_t)F.tx, tz = (uint32_t)F.tz, txtz = (uint32_t)F.txtz;      // scalar registers, scalar arithmetic
    const uint32_t bty = udiv_fast(t, txtz, F.m_txtz);
    const uint32_t rem = t - bty * txtz;
    const uint32_t btz = udiv_fast(rem, tx, F.m_tx);
    uint32_t btx = rem - btz * tx;
    const uint32_t rot = btz + bty * tz;                          // x-position rotated by the row number (lattice_point_at)
    btx += rot - udiv_fast(rot, tx, F.m_tx) * tx;
    btx = min(btx, btx - tx);                                     // (unsigned: btx - tx wraps unless btx >= tx)
    const int ix = F.sx0 + (int)(btx * 16u) + wave * 4 + (lane & 3);
    const int iy = F.sy0 + (int)(bty * 4u) + ((lane >> 2) & 3);
    const int iz = F.sz0 + (int)(btz * 4u) + (lane >> 4);
    cx = min(ix, F.sx1 - 1); cy = min(iy, F.sy1 - 1); cz = min(iz, F.sz1 - 1);      // (lattice_clamp)
    return ix < F.sx1 && iy < F.sy1 && iz < F.sz1;
}

// the lane whose box distances order the two children of a node: the block's centre (4^3), its first point otherwise
__device__ __forceinline__ int packet_center_lane(const LatticeMap &L) { return (L.pk == 0 || L.pk == 4) ? 21 : 0; }

// a padding lane of a boundary tile works on a clamped copy of a real point of the region
__device__ __forceinline__ void lattice_clamp(const LatticeMap &L, int ix, int iy, int iz, int &cx, int &cy, int &cz)
{
    cx = min(ix, L.sx1 - 1); cy = min(iy, L.sy1 - 1); cz = min(iz, L.sz1 - 1);
}

__device__ __forceinline__ f3 lattice_world(int res, int ix, int iy, int iz)
{
    // batch_eval, seg3d_lossless.py:132-137 with align_corners=True:
    //   coords.float() / (R-1) * (b_max - b_min) + b_min, b_min=[-1,1,-1], b_max=[1,-1,1]
    const float rm1 = (float)(res - 1);
    const float fx = (float)ix / rm1, fy = (float)iy / rm1, fz = (float)iz / rm1;
    return mk3(fx * 2.0f + (-1.0f), fy * (-2.0f) + 1.0f, fz * 2.0f + (-1.0f));
}

// NaN / Inf / astronomically large coordinates (bad caller data) are evaluated at +-kFarCoord: far outside the cube, so the
// occupancy is 0 and the point is an "outside" entry of the call's outlier list like any other far point - instead of a
// search that finds no triangle and then indexes with it (v_max/v_min return the finite operand for a NaN).  No-op for every
// coordinate of magnitude <= 64: the body lives in [-1,1]^3.
constexpr float kFarCoord = 64.0f;
__device__ __forceinline__ f3 clamp_far(f3 r)
{
    r.x = fminf(fmaxf(r.x, -kFarCoord), kFarCoord);
    r.y = fminf(fmaxf(r.y, -kFarCoord), kFarCoord);
    r.z = fminf(fmaxf(r.z, -kFarCoord), kFarCoord);
    return r;
}

// The raw cal_sdf_batch leaf (icon_sdf_query) takes points in whatever units the caller's mesh is in (voxel units of
// export_mesh, centimetres of a scan): only what cannot be searched is changed - NaN / Inf / beyond +-1e18 (where d^2
// leaves float32) - never an ordinary far point.
__device__ __forceinline__ f3 clamp_raw(f3 r)
{
    constexpr float kBig = 1e18f;
    r.x = fminf(fmaxf(r.x, -kBig), kBig);
    r.y = fminf(fmaxf(r.y, -kBig), kBig);
    r.z = fminf(fmaxf(r.z, -kBig), kBig);
    return r;
}

__device__ __forceinline__ f3 project(const Calib &c, f3 p)
{
    // orthogonal(): baddbmm(trans, rot, points), geometry.py:54-56
    // k-ordered fma chain, then + trans (what ATen's CPU baddbmm computes for K = 3); exact for identity
    f3 r;
    r.x = fmaf(c.m[2], p.z, fmaf(c.m[1], p.y, c.m[0] * p.x)) + c.m[3];
    r.y = fmaf(c.m[6], p.z, fmaf(c.m[5], p.y, c.m[4] * p.x)) + c.m[7];
    r.z = fmaf(c.m[10], p.z, fmaf(c.m[9], p.y, c.m[8] * p.x)) + c.m[11];
    return clamp_far(r);
}

__device__ __forceinline__ Calib resolve_calib(Calib c)
{
    if (c.d) {
        cfloat *q = as_const(c.d);
#pragma unroll
        for (int k = 0; k < 12; ++k) c.m[k] = q[k];
    }
    return c;
}

__device__ __forceinline__ uint32_t in_cube_bit(f3 p)
{
    const bool in = p.x > -1.0f && p.x < 1.0f && p.y > -1.0f && p.y < 1.0f && p.z > -1.0f && p.z < 1.0f;
    return in ? kCodeInCube : 0u;
}

// the 1-byte code of a point (icon prior): in_cube, inside (check_sign), and for |sdf| >= sdf_clip the outlier
// flag with sign(sdf) + 1 (lib/net/HGPIFuNet.py:298-299); sdf = +-sqrt(d^2)/sqrt(3) exactly as sdf_attrs forms it
__device__ __forceinline__ uint32_t sign_code(f3 p, float d2, bool ins, float sdf_clip)
{
    const float dist = sqrtf(d2) / sqrtf(3.0f);      // mesh_util.py:391
    const float s = ins ? dist : -dist;              // :393-394
    uint32_t code = in_cube_bit(p) | (ins ? kCodeInside : 0u);
    if (fabsf(s) >= sdf_clip) {
        const int sg = (s > 0.0f) ? 1 : ((s < 0.0f) ? -1 : 0);
        code |= kCodeOutlier | ((uint32_t)(sg + 1) << kCodeSignShift);
    }
    return code;
}

// Hand-over of the nearest-triangle result to k_sign / the fused kernel.  |sdf| = sqrt(d^2)/sqrt(3) does not depend on
// the inside test, so the kernel that holds d^2 in a register decides "outside the clip band" (exactly the comparison
// sign_code makes) and flags it in the slot word; d^2 itself is only stored for the ~6 % of points inside the band,
// the only ones whose consumers read it: ~90 MB less HBM traffic per 257^3 step.
constexpr uint32_t kNearFar = 0x8000u;
__device__ __forceinline__ bool near_far_exact(float d2, float sdf_clip)
{
    const float dist = sqrtf(d2) / sqrtf(3.0f);
    return dist >= sdf_clip && dist > 0.0f;                 // dist == 0 (sign 0) stays on the general path
}
__device__ __forceinline__ void store_near(const NearRef &r, int64_t i, const Nearest &nr, float sdf_clip)
{
    // `far` must be the very comparison sign_code makes on sqrt(d^2) / sqrt(3) (IEEE sqrt + division: ~45 instructions per
    // lane).  d^2 beyond 3 clip^2 (1 +- 1e-4) decides it without them - the two roundings move the quotient by 2e-7 relative -
    // and only a wave that holds a point INSIDE that sliver (or a clip that is not an ordinary positive number) evaluates the
    // exact form: a wave-uniform branch that is almost never taken.
    const float c2 = sdf_clip * sdf_clip * 3.0f;
    const bool ordinary = sdf_clip > 1e-12f && sdf_clip < 1e12f;
    const bool surely_far = nr.d2 > c2 * 1.0001f, surely_near = nr.d2 < c2 * 0.9999f;
    bool far = surely_far;
    if (__any(!ordinary || !(surely_far || surely_near))) far = near_far_exact(nr.d2, sdf_clip);
    r.lo[i] = (uint16_t)(((uint32_t)nr.slot & 0x7fffu) | (far ? kNearFar : 0u));
    if (r.hi) r.hi[i] = (uint8_t)((uint32_t)nr.slot >> 15);   // wave-uniform: meshes with more than 32,768 slots only
    if (!far) r.d2[i] = nr.d2;
}
// readers take the index of the point IN ITS CALL; a lattice-subset call (NearRef::map) finds its entry through the map
__device__ __forceinline__ int64_t near_index(const NearRef &r, int64_t i) { return r.map ? (int64_t)r.map[i] : i; }
__device__ __forceinline__ bool near_is_far(const NearRef &r, int64_t i) { return (r.lo[near_index(r, i)] & kNearFar) != 0; }
__device__ __forceinline__ float near_d2(const NearRef &r, int64_t i) { return r.d2[near_index(r, i)]; }
__device__ __forceinline__ int near_slot_of(const NearRef &r, int64_t i)
{
    const int64_t e = near_index(r, i);
    uint32_t s = (uint32_t)r.lo[e] & 0x7fffu;
    if (r.hi) s |= (uint32_t)r.hi[e] << 15;
    return (int)s;
}
// code byte of a point flagged kNearFar: what sign_code returns for |s| >= sdf_clip, s = +-dist, dist > 0
__device__ __forceinline__ uint32_t sign_code_far(f3 p, bool ins)
{
    return in_cube_bit(p) | (ins ? kCodeInside : 0u) | kCodeOutlier | ((ins ? 2u : 0u) << kCodeSignShift);
}

__device__ __forceinline__ void store_row(float *X, int64_t i, const float *row)
{
    float4 *dst = reinterpret_cast<float4 *>(X + i * kXRow);
    dst[0] = make_float4(row[0], row[1], row[2], row[3]);
    dst[1] = make_float4(row[4], row[5], row[6], row[7]);
    dst[2] = make_float4(row[8], row[9], row[10], row[11]);
    dst[3] = make_float4(row[12], row[13], row[14], row[15]);
}

}  // namespace icon
