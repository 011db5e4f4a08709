/*
 * oracle/icon_oracle.c - CPU restatement of ICON's per-point occupancy query path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under icon_amd/ links, imports or executes this file;
 * it is the checker used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * What it restates (all citations relative to /root/reference):
 *   orc_vertex_normals   pytorch3d Meshes.verts_normals_padded(), call site
 *                        lib/dataset/mesh_util.py:367               (third-party leaf)
 *   orc_nearest_brute    kaolin.metrics.trianglemesh.point_to_mesh_distance, call site
 *                        lib/dataset/mesh_util.py:374               (third-party leaf)
 *   orc_check_sign       kaolin.ops.mesh.check_sign, call site
 *                        lib/dataset/mesh_util.py:393               (third-party leaf)
 *   orc_cal_sdf          cal_sdf_batch                 lib/dataset/mesh_util.py:357-396
 *                        + barycentric_coordinates_of_projection   :319-354
 *                        + face_vertices               lib/common/render_utils.py:149-163
 *   orc_query_icon       HGPIFuNet.query (icon branch) lib/net/HGPIFuNet.py:268-367
 *                        + index / grid_sample         lib/net/geometry.py:21-43
 *                        + feat_select                 lib/dataset/mesh_util.py:266-277
 *                        + MLP.forward                 lib/net/MLP.py:49-72
 *
 * PARITY STATUS.  Everything above the three third-party leaves is pinned by running the
 * reference's own Python verbatim (oracle/ref_loader.py) against this file - see
 * tests/test_oracle_vs_reference.py and tests/golden/.  The three leaves live in kaolin
 * 0.11.0 / pytorch3d (requirements_colab.txt:36, requirements.txt:33,35), whose sources are
 * not under /root/reference and which publish no golden vectors: PARITY UNPINNED for the
 * leaves.  They are pinned mathematically instead (exact closest-triangle distance; inside /
 * outside of a watertight mesh), cross-checked by an independent float64 brute force in
 * tests/test_oracle_leaves.py, and the tie rule - lowest face index among float32-equal
 * squared distances - is DEFINED here.
 *
 * ARITHMETIC SPEC.  float32 throughout, no contraction except the explicit fmaf() calls
 * written below, IEEE division and sqrt.  The HIP kernels implement the same operation
 * sequence for the nearest-triangle and ray-parity predicates so that the integer outputs
 * (face index, inside flag, visibility flag) are bit-exact, not merely close.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 v3_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline float v3_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* cross product, each component = fma(a1, b2, -(a2*b1)) */
static inline v3 v3_cross(v3 a, v3 b)
{
    v3 r;
    r.x = fmaf(a.y, b.z, -(a.z * b.y));
    r.y = fmaf(a.z, b.x, -(a.x * b.z));
    r.z = fmaf(a.x, b.y, -(a.y * b.x));
    return r;
}
static inline v3 ld3(const float *p, int64_t i) { v3 r = { p[3 * i], p[3 * i + 1], p[3 * i + 2] }; return r; }

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * S1  vertex normals: sum over incident faces, in ascending face order, of the un-normalised
 *     face cross product (v1-v0) x (v2-v0); then v / max(|v|, 1e-6)
 *     (pytorch3d Meshes._compute_vertex_normals + F.normalize(eps=1e-6); pytorch3d accumulates
 *     with a non-deterministic index_add, we define the order).
 * ---------------------------------------------------------------------------------------- */
void orc_vertex_normals(const float *verts, int64_t V, const int64_t *faces, int64_t F, float *out)
{
    memset(out, 0, sizeof(float) * 3 * (size_t)V);
    for (int64_t f = 0; f < F; ++f) {
        const int64_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const v3 a = ld3(verts, i0), b = ld3(verts, i1), c = ld3(verts, i2);
        const v3 n = v3_cross(v3_sub(b, a), v3_sub(c, a));
        const int64_t ids[3] = { i0, i1, i2 };
        for (int k = 0; k < 3; ++k) {
            out[3 * ids[k] + 0] += n.x;
            out[3 * ids[k] + 1] += n.y;
            out[3 * ids[k] + 2] += n.z;
        }
    }
    for (int64_t v = 0; v < V; ++v) {
        const float x = out[3 * v], y = out[3 * v + 1], z = out[3 * v + 2];
        float len = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
        if (len < 1e-6f) len = 1e-6f;
        out[3 * v] = x / len; out[3 * v + 1] = y / len; out[3 * v + 2] = z / len;
    }
}

/* ------------------------------------------------------------------------------------------
 * S2  exact point-triangle squared distance, "face or nearest edge" form.
 *     Per-triangle constants (orc_tri_setup, evaluated once per triangle on the CPU in both the
 *     checker and the product's mesh preparation): edge vectors, reciprocal squared edge lengths,
 *     the Gram matrix of (ab, ac) and the reciprocal of its determinant; a zero-length edge gets a
 *     reciprocal of 0 (its segment collapses to the vertex), a zero-area triangle a NaN determinant
 *     reciprocal (never "inside": the minimum over its edge segments is its exact distance).
 *     Per point: barycentrics of the plane projection (s, t); if 0 <= s, 0 <= t, s + t <= 1 the
 *     result is |(p - a) - s ab - t ac|^2, otherwise the minimum over the three edge segments
 *     of |(p - origin) - clamp(t) * edge|^2 (fma chains starting from the rounded p - origin).
 *     No division, no data-dependent branches: on the GPU this is ~65 VALU operations per
 *     triangle (sub / mul / fma / clamp / min / select), all exactly rounded.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    v3 a, b, ab, ac, bc;
    float i00, i11, ibc;      /* 1/|ab|^2, 1/|ac|^2, 1/|bc|^2 (0 if the edge has zero length) */
    float a00, a01, a11, inn; /* ab.ab, ab.ac, ac.ac, 1/(a00*a11 - a01^2) (NaN if not positive) */
} orc_tri;

void orc_tri_setup(const float *pa, const float *pb, const float *pc, orc_tri *t)
{
    t->a = ld3(pa, 0); t->b = ld3(pb, 0);
    const v3 c = ld3(pc, 0);
    t->ab = v3_sub(t->b, t->a); t->ac = v3_sub(c, t->a); t->bc = v3_sub(c, t->b);
    t->a00 = v3_dot(t->ab, t->ab); t->a01 = v3_dot(t->ab, t->ac); t->a11 = v3_dot(t->ac, t->ac);
    const float b11 = v3_dot(t->bc, t->bc);
    t->i00 = (t->a00 > 0.0f) ? 1.0f / t->a00 : 0.0f;
    t->i11 = (t->a11 > 0.0f) ? 1.0f / t->a11 : 0.0f;
    t->ibc = (b11 > 0.0f) ? 1.0f / b11 : 0.0f;
    const float nn = fmaf(t->a00, t->a11, -(t->a01 * t->a01));
    /* zero area (or a sliver whose determinant rounds to <= 0): NaN -> both barycentrics NaN -> never "inside" ->
     * the distance is the minimum over the three edge segments, which is exact for a degenerate triangle */
    t->inn = (nn > 0.0f) ? 1.0f / nn : NAN;
}

/* max / min as plain comparisons (libm's fmaxf/fminf are out-of-line calls without fast-math).
 * For the finite operands that occur here they return exactly what fmaxf/fminf and the GPU's
 * v_max_f32 / v_min_f32 return. */
static inline float max_f(float a, float b) { return (a > b) ? a : b; }
static inline float min_f(float a, float b) { return (b < a) ? b : a; }

static inline float seg_dist2(v3 po, v3 e, float dot_e_po, float inv_len2)
{
    /* po = p - origin; the residual is formed from it, (p - o) - t e, not as p - (o + t e): one rounding at the
     * scale of the triangle instead of one at the scale of the coordinates */
    const float t = min_f(max_f(dot_e_po * inv_len2, 0.0f), 1.0f);
    v3 d; d.x = fmaf(-t, e.x, po.x); d.y = fmaf(-t, e.y, po.y); d.z = fmaf(-t, e.z, po.z);
    return v3_dot(d, d);
}

float orc_tri_dist2(const float *pp, const orc_tri *t)
{
    const v3 p = ld3(pp, 0);
    const v3 ap = v3_sub(p, t->a), bp = v3_sub(p, t->b);
    const float d1 = v3_dot(t->ab, ap), d2 = v3_dot(t->ac, ap), d3 = v3_dot(t->bc, bp);
    const float s = fmaf(t->a11, d1, -(t->a01 * d2)) * t->inn;
    const float u = fmaf(t->a00, d2, -(t->a01 * d1)) * t->inn;
    const int inside = (s >= 0.0f) & (u >= 0.0f) & (s + u <= 1.0f);
    v3 df;
    df.x = fmaf(-u, t->ac.x, fmaf(-s, t->ab.x, ap.x));
    df.y = fmaf(-u, t->ac.y, fmaf(-s, t->ab.y, ap.y));
    df.z = fmaf(-u, t->ac.z, fmaf(-s, t->ab.z, ap.z));
    const float d_face = v3_dot(df, df);
    const float e0 = seg_dist2(ap, t->ab, d1, t->i00);
    const float e1 = seg_dist2(ap, t->ac, d2, t->i11);
    const float e2 = seg_dist2(bp, t->bc, d3, t->ibc);
    const float d_edge = min_f(min_f(e0, e1), e2);
    return inside ? d_face : d_edge;
}

float orc_point_tri_dist2(const float *pp, const float *pa, const float *pb, const float *pc)
{
    orc_tri t;
    orc_tri_setup(pa, pb, pc, &t);
    return orc_tri_dist2(pp, &t);
}

/* S3  argmin over faces, linear scan, strict '<' => lowest index wins exact ties; NaN never wins */
void orc_nearest_brute(const float *verts, const int64_t *faces, int64_t F,
                       const float *pts, int64_t N, float *out_d2, int64_t *out_idx)
{
    orc_tri *tri = (orc_tri *)malloc(sizeof(orc_tri) * (size_t)F);
    for (int64_t f = 0; f < F; ++f)
        orc_tri_setup(verts + 3 * faces[3 * f], verts + 3 * faces[3 * f + 1], verts + 3 * faces[3 * f + 2], tri + f);
    /* blocked for cache reuse: 64 points x tiles of 128 triangles (11 KiB); every point still sees
     * the faces in ascending order, so the strict '<' keeps the lowest index on exact ties */
    const int64_t PB = 64, TB = 128;
    const int64_t nblocks = (N + PB - 1) / PB;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < nblocks; ++blk) {
        const int64_t i0 = blk * PB, i1 = (i0 + PB < N) ? i0 + PB : N;
        float best[64]; int64_t bi[64];
        for (int64_t i = i0; i < i1; ++i) { best[i - i0] = INFINITY; bi[i - i0] = 0; }
        for (int64_t f0 = 0; f0 < F; f0 += TB) {
            const int64_t f1 = (f0 + TB < F) ? f0 + TB : F;
            for (int64_t i = i0; i < i1; ++i) {
                float b = best[i - i0]; int64_t k = bi[i - i0];
                for (int64_t f = f0; f < f1; ++f) {
                    const float d = orc_tri_dist2(pts + 3 * i, tri + f);
                    if (d < b) { b = d; k = f; }
                }
                best[i - i0] = b; bi[i - i0] = k;
            }
        }
        for (int64_t i = i0; i < i1; ++i) { out_d2[i] = best[i - i0]; out_idx[i] = bi[i - i0]; }
    }
    free(tri);
}

/* ------------------------------------------------------------------------------------------
 * S4  inside test: parity of +x ray crossings.  The ray from p along +x hits triangle (a,b,c)
 *     iff p's (y,z) projection lies in the projected triangle and the hit is in front of p.
 *     Every mesh edge is evaluated once in a canonical direction (lower vertex id first), so
 *     the two triangles sharing an edge take opposite sides of it for every query point and a
 *     ray through an edge is counted exactly once; exact zeros (ray through a vertex) are
 *     resolved by one global symbolic perturbation of the query point (edge_side).
 * ---------------------------------------------------------------------------------------- */
/* twice the signed area of (q, vi, vj) in the (y,z) plane, vi -> vj in canonical (lower id first)
 * direction, written relative to q so that it is EXACTLY zero when q coincides with the
 * projection of either end point */
static inline float edge_fn(float yi, float zi, float yj, float zj, float qy, float qz)
{
    const float t1 = (yi - qy) * (zj - qz);
    return fmaf(-(zi - qz), (yj - qy), t1);
}

/* side of q w.r.t. the canonical edge; an exact zero is resolved by the symbolic perturbation
 * q -> q + (eps, eps^2): sign of dE/dqy = -(zj - zi), then of dE/dqz = (yj - yi).  The
 * perturbation is the same for every edge, so a ray through a mesh vertex or edge is assigned to
 * exactly the triangles a slightly shifted ray would cross (simulation of simplicity). */
static inline int edge_side(float yi, float zi, float yj, float zj, float e)
{
    if (e > 0.0f) return 1;
    if (e < 0.0f) return 0;
    const float dz = zj - zi, dy = yj - yi;
    if (dz != 0.0f) return dz < 0.0f;
    return dy > 0.0f;
}

/* oriented edge value and side for the edge from vertex (id ia) to vertex (id ib) */
static inline void oriented_edge(int64_t ia, v3 a, int64_t ib, v3 b, float qy, float qz,
                                 float *val, int *pos)
{
    if (ia < ib) { const float e = edge_fn(a.y, a.z, b.y, b.z, qy, qz); *val = e; *pos = edge_side(a.y, a.z, b.y, b.z, e); }
    else         { const float e = edge_fn(b.y, b.z, a.y, a.z, qy, qz); *val = -e; *pos = !edge_side(b.y, b.z, a.y, a.z, e); }
}

int orc_ray_hit(const float *pp, int64_t ia, const float *pa, int64_t ib, const float *pb,
                int64_t ic, const float *pc)
{
    const v3 p = ld3(pp, 0), a = ld3(pa, 0), b = ld3(pb, 0), c = ld3(pc, 0);
    float e_ab, e_bc, e_ca; int s_ab, s_bc, s_ca;
    oriented_edge(ia, a, ib, b, p.y, p.z, &e_ab, &s_ab);
    oriented_edge(ib, b, ic, c, p.y, p.z, &e_bc, &s_bc);
    oriented_edge(ic, c, ia, a, p.y, p.z, &e_ca, &s_ca);
    if (!(s_ab == s_bc && s_bc == s_ca)) return 0;
    /* barycentric weights: a <- e_bc, b <- e_ca, c <- e_ab; depth test without division */
    const float num = fmaf(e_ab, c.x - p.x, fmaf(e_ca, b.x - p.x, e_bc * (a.x - p.x)));
    return s_ab ? (num > 0.0f) : (num < 0.0f);
}

void orc_check_sign(const float *verts, const int64_t *faces, int64_t F,
                    const float *pts, int64_t N, uint8_t *inside)
{
    const int64_t PB = 64, TB = 256;
    const int64_t nblocks = (N + PB - 1) / PB;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < nblocks; ++blk) {
        const int64_t i0 = blk * PB, i1 = (i0 + PB < N) ? i0 + PB : N;
        int cnt[64];
        for (int64_t i = i0; i < i1; ++i) cnt[i - i0] = 0;
        for (int64_t f0 = 0; f0 < F; f0 += TB) {
            const int64_t f1 = (f0 + TB < F) ? f0 + TB : F;
            for (int64_t i = i0; i < i1; ++i) {
                int c = 0;
                for (int64_t f = f0; f < f1; ++f) {
                    const int64_t ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
                    c += orc_ray_hit(pts + 3 * i, ia, verts + 3 * ia, ib, verts + 3 * ib, ic, verts + 3 * ic);
                }
                cnt[i - i0] += c;
            }
        }
        for (int64_t i = i0; i < i1; ++i) inside[i] = (uint8_t)(cnt[i - i0] & 1);
    }
}

/* ------------------------------------------------------------------------------------------
 * S5  cal_sdf_batch (lib/dataset/mesh_util.py:357-396), batch 1.
 *     out_sdf [N], out_norm [N,3], out_cmap [N,3], out_vis [N] (0/1), optional out_idx [N].
 * ---------------------------------------------------------------------------------------- */
static void bary_heidrich(v3 p, v3 v0, v3 v1, v3 v2, float w[3])
{
    /* barycentric_coordinates_of_projection, mesh_util.py:337-353: unclamped */
    const v3 u = v3_sub(v1, v0), v = v3_sub(v2, v0);
    const v3 n = v3_cross(u, v);
    float s = v3_dot(n, n);
    if (s == 0.0f) s = 1e-6f;
    const float inv = 1.0f / s;
    const v3 ww = v3_sub(p, v0);
    const float b2 = v3_dot(v3_cross(u, ww), n) * inv;
    const float b1 = v3_dot(v3_cross(ww, v), n) * inv;
    w[0] = (1.0f - b1) - b2; w[1] = b1; w[2] = b2;
}

/* icon_accel.c: the same two leaves through a BVH / ray bins, bit-identical results */
typedef struct orc_accel orc_accel;
orc_accel *orc_accel_build(const float *verts, int64_t V, const int64_t *faces, int64_t F);
void orc_accel_free(orc_accel *A);
void orc_accel_nearest(const orc_accel *A, const float *pts, int64_t N, float *out_d2, int64_t *out_idx);
void orc_accel_check_sign(const orc_accel *A, const float *pts, int64_t N, uint8_t *inside);
void orc_accel_apply_tie_rule(const orc_accel *A, const float *pts, int64_t N, float *io_d2, int64_t *io_idx);

/* 1 (default): cal_sdf / query_icon answer the two O(N*F) leaves through icon_accel.c;
 * 0: through the linear scans above (the definition).  tests/test_oracle_leaves.py holds both equal. */
static int g_accel = 1;
void orc_set_accel(int on) { g_accel = on ? 1 : 0; }
int orc_get_accel(void) { return g_accel; }

void orc_cal_sdf(const float *verts, int64_t V, const int64_t *faces, int64_t F,
                 const float *cmaps, const float *vis, const float *pts, int64_t N,
                 float *out_sdf, float *out_norm, float *out_cmap, float *out_vis,
                 int64_t *out_idx, uint8_t *out_inside)
{
    float *vn = (float *)malloc(sizeof(float) * 3 * (size_t)V);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)N);
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    uint8_t *ins = (uint8_t *)malloc((size_t)N);
    orc_vertex_normals(verts, V, faces, F, vn);
    if (g_accel) {
        orc_accel *A = orc_accel_build(verts, V, faces, F);
        orc_accel_nearest(A, pts, N, d2, idx);
        orc_accel_apply_tie_rule(A, pts, N, d2, idx);          /* diagnostics: no-op unless orc_set_tie_rule(1, u) */
        orc_accel_check_sign(A, pts, N, ins);
        orc_accel_free(A);
    } else {
        orc_nearest_brute(verts, faces, F, pts, N, d2, idx);
        orc_check_sign(verts, faces, F, pts, N, ins);
    }
    const float sqrt3 = sqrtf(3.0f);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const int64_t f = idx[i];
        const int64_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const int64_t id[3] = { i0, i1, i2 };
        float w[3];
        bary_heidrich(ld3(pts, i), ld3(verts, i0), ld3(verts, i1), ld3(verts, i2), w);
        for (int k = 0; k < 3; ++k) {
            /* (attr * w[:, :, None]).sum(1): ((a0*w0 + a1*w1) + a2*w2) */
            out_cmap[3 * i + k] = fmaf(cmaps[3 * id[2] + k], w[2],
                                       fmaf(cmaps[3 * id[1] + k], w[1], cmaps[3 * id[0] + k] * w[0]));
            const float nk = fmaf(vn[3 * id[2] + k], w[2],
                                  fmaf(vn[3 * id[1] + k], w[1], vn[3 * id[0] + k] * w[0]));
            out_norm[3 * i + k] = (k == 1) ? nk : -nk;           /* * [-1, 1, -1], :389-390 */
        }
        const float vsum = fmaf(vis[id[2]], w[2], fmaf(vis[id[1]], w[1], vis[id[0]] * w[0]));
        out_vis[i] = (vsum >= 0.1f) ? 1.0f : 0.0f;               /* .ge(1e-1), :387-388 */
        const float dist = sqrtf(d2[i]) / sqrt3;                 /* :391 */
        out_sdf[i] = ins[i] ? dist : -dist;                      /* :393-394 */
        if (out_idx) out_idx[i] = f;
        if (out_inside) out_inside[i] = ins[i];
    }
    free(vn); free(d2); free(idx); free(ins);
}

/* ------------------------------------------------------------------------------------------
 * S7  grid_sample(feat[1,C,H,W], (x,y)), bilinear, zeros padding, align_corners=True
 *     (lib/net/geometry.py:41-42; ATen GridSampler arithmetic)
 * ---------------------------------------------------------------------------------------- */
static inline float plane_at(const float *pl, int H, int W, int iy, int ix)
{
    return (ix >= 0 && ix < W && iy >= 0 && iy < H) ? pl[(size_t)iy * W + ix] : 0.0f;
}

void orc_bilinear(const float *feat, int C, int H, int W, float x, float y, float *out)
{
    const float ix = ((x + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((y + 1.0f) / 2.0f) * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = ((float)x1 - ix) * ((float)y1 - iy);
    const float ne = (ix - (float)x0) * ((float)y1 - iy);
    const float sw = ((float)x1 - ix) * (iy - (float)y0);
    const float se = (ix - (float)x0) * (iy - (float)y0);
    for (int c = 0; c < C; ++c) {
        const float *pl = feat + (size_t)c * H * W;
        float acc = plane_at(pl, H, W, y0, x0) * nw;
        acc += plane_at(pl, H, W, y0, x1) * ne;
        acc += plane_at(pl, H, W, y1, x0) * sw;
        acc += plane_at(pl, H, W, y1, x1) * se;
        out[c] = acc;
    }
}

/* S7b 3-D grid_sample(vol[1,C,D,H,W], (x,y,z)) trilinear, zeros padding, align_corners=True
 *     (lib/net/geometry.py:32-35,41-42; PaMIR branch HGPIFuNet.py:351-354) */
void orc_trilinear(const float *vol, int C, int D, int H, int W, float x, float y, float z, float *out)
{
    const float ix = ((x + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((y + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((z + 1.0f) / 2.0f) * (float)(D - 1);
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
    const float tx = ix - (float)x0, ty = iy - (float)y0, tz = iz - (float)z0;
    for (int c = 0; c < C; ++c) {
        const float *v = vol + (size_t)c * D * H * W;
        float acc = 0.0f;
        for (int dz = 0; dz < 2; ++dz)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
                    const float wgt = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) * (dz ? tz : 1.0f - tz);
                    if (xi >= 0 && xi < W && yi >= 0 && yi < H && zi >= 0 && zi < D)
                        acc += v[((size_t)zi * H + yi) * W + xi] * wgt;
                }
        out[c] = acc;
    }
}

/* ------------------------------------------------------------------------------------------
 * S8  MLP.forward (lib/net/MLP.py:49-72): Conv1d(k=1) -> BatchNorm1d(eval, eps 1e-5; none for norm_mlp = 'weight') ->
 *     LeakyReLU(0.01); raw input re-concatenated (after the activations) before res layers;
 *     last_op: none in test mode, Sigmoid otherwise (HGPIFuNet.py:133).  Weights arrive un-folded, exactly as the
 *     reference state_dict holds them.  accumulate_f64 != 0 gives the high-precision variant.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int n_layers;              /* number of Conv1d layers                     */
    const int *cin;            /* [n_layers] input channels incl. skip concat */
    const int *cout;           /* [n_layers]                                  */
    const int *is_res;         /* [n_layers] 1 if cat([y, x]) precedes it     */
    const float *const *W;     /* [n_layers] -> [cout, cin]                   */
    const float *const *b;     /* [n_layers] -> [cout]                        */
    const float *const *bn_g;  /* [n_layers-1] gamma                          */
    const float *const *bn_b;  /* beta                                        */
    const float *const *bn_m;  /* running_mean                                */
    const float *const *bn_v;  /* running_var                                 */
    int last_op;               /* 0: none (cfg.test_mode); 1: Sigmoid (MLP.py:68-70, HGPIFuNet.py:133) */
} orc_mlp;

#define ORC_MAXC 1024

static void mlp_point(const orc_mlp *m, const float *x, int c0, float *out, int accumulate_f64)
{
    float cur[ORC_MAXC], nxt[ORC_MAXC];
    int n_cur = c0;
    memcpy(cur, x, sizeof(float) * (size_t)c0);
    for (int l = 0; l < m->n_layers; ++l) {
        if (m->is_res[l]) { memcpy(cur + n_cur, x, sizeof(float) * (size_t)c0); n_cur += c0; }
        const int ci = m->cin[l], co = m->cout[l];
        (void)ci;
        for (int o = 0; o < co; ++o) {
            const float *w = m->W[l] + (size_t)o * n_cur;
            float y;
            if (accumulate_f64) {
                double acc = 0.0;
                for (int k = 0; k < n_cur; ++k) acc += (double)w[k] * (double)cur[k];
                acc += (double)m->b[l][o];
                if (l != m->n_layers - 1) {
                    if (m->bn_m)   /* NULL: norm_mlp = 'weight' or none - activation only, MLP.py:64-65 */
                        acc = (acc - (double)m->bn_m[l][o]) / sqrt((double)m->bn_v[l][o] + 1e-5) * (double)m->bn_g[l][o]
                              + (double)m->bn_b[l][o];
                    if (acc < 0.0) acc *= 0.01;
                }
                y = (float)acc;
            } else {
                float acc = 0.0f;
                for (int k = 0; k < n_cur; ++k) acc += w[k] * cur[k];
                acc += m->b[l][o];
                if (l != m->n_layers - 1) {
                    if (m->bn_m) acc = (acc - m->bn_m[l][o]) / sqrtf(m->bn_v[l][o] + 1e-5f) * m->bn_g[l][o] + m->bn_b[l][o];
                    if (acc < 0.0f) acc *= 0.01f;
                }
                y = acc;
            }
            nxt[o] = y;
        }
        memcpy(cur, nxt, sizeof(float) * (size_t)co);
        n_cur = co;
    }
    if (m->last_op == 1) {                                       /* self.last_op(y), MLP.py:68-70 */
        for (int o = 0; o < n_cur; ++o)
            cur[o] = accumulate_f64 ? (float)(1.0 / (1.0 + exp(-(double)cur[o]))) : 1.0f / (1.0f + expf(-cur[o]));
    }
    memcpy(out, cur, sizeof(float) * (size_t)n_cur);
}

/* x: [N, c0] point-major -> out [N, c_last] */
void orc_mlp_forward(const orc_mlp *m, const float *x, int64_t N, int c0, float *out, int accumulate_f64)
{
    const int c_last = m->cout[m->n_layers - 1];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        mlp_point(m, x + (size_t)i * c0, c0, out + (size_t)i * c_last, accumulate_f64);
}

/* orthogonal(), lib/net/geometry.py:54-56: baddbmm(trans, rot, points); calib = 12 floats [3,4] */
void orc_project(const float *calib, const float *pts, int64_t N, float *xyz)
{
    for (int64_t i = 0; i < N; ++i)
        for (int r = 0; r < 3; ++r) {
            /* (rot @ p) as a k-ordered fma chain, then + trans: bit-identical to ATen's CPU
             * baddbmm for K = 3 (probe in tests/test_oracle_vs_reference.py); exact for identity */
            float acc = calib[4 * r + 0] * pts[3 * i];
            acc = fmaf(calib[4 * r + 1], pts[3 * i + 1], acc);
            acc = fmaf(calib[4 * r + 2], pts[3 * i + 2], acc);
            xyz[3 * i + r] = acc + calib[4 * r + 3];
        }
}

/* in_cube = all(-1 < xyz < 1) strict; preds = in_cube * preds (HGPIFuNet.py:274-275,363) */
void orc_mask_in_cube(const float *xyz, int64_t N, float *occ)
{
    for (int64_t i = 0; i < N; ++i) {
        const int in = xyz[3 * i] > -1.0f && xyz[3 * i] < 1.0f && xyz[3 * i + 1] > -1.0f && xyz[3 * i + 1] < 1.0f
                       && xyz[3 * i + 2] > -1.0f && xyz[3 * i + 2] < 1.0f;
        occ[i] = (in ? 1.0f : 0.0f) * occ[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * S6+S7+S8  HGPIFuNet.query, icon branch with smpl_feats = [sdf, norm, vis, cmap]
 *     (lib/net/HGPIFuNet.py:268-312,329-365), identity or affine calibration.
 *     calib: 12 floats, row-major [3,4] (rot | trans) as used by orthogonal(), geometry.py:54-56.
 *     out_x (optional): [N,C/2+7] MLP input in reference channel order
 *                       [img0..5, sdf, cmap r g b, norm x y z] (HGPIFuNet.py:301-311,343,359).
 * ---------------------------------------------------------------------------------------- */
/* cfg.net.smpl_feats (lib/net/HGPIFuNet.py:301-309): bit 0 = 'cmap', bit 1 = 'norm' follow the sdf ('sdf' is always there);
 * bit 2 = 'vis': smpl_vis selects the feature half (:334-336) - without it every feature channel is an input (:345-346,
 * configs/train/icon-mvp.yaml:40).  Default: all, as every configs/ *.yaml. */
static int g_smpl_mask = 7;
void orc_set_smpl_feats(int has_cmap, int has_norm, int has_vis) { g_smpl_mask = (has_cmap ? 1 : 0) | (has_norm ? 2 : 0) | (has_vis ? 4 : 0); }
static int icon_img_width(int C) { return (g_smpl_mask & 4) ? C / 2 : C; }
int orc_icon_c0(int C) { return icon_img_width(C) + 1 + ((g_smpl_mask & 1) ? 3 : 0) + ((g_smpl_mask & 2) ? 3 : 0); }

void orc_query_icon(const float *verts, int64_t V, const int64_t *faces, int64_t F,
                    const float *cmaps, const float *vis,
                    const float *feat, int C, int H, int W,
                    const orc_mlp *mlp, float sdf_clip, const float *calib,
                    const float *pts, int64_t N, float *out_occ, float *out_x, int accumulate_f64,
                    int cmap_local)
{
    const int half = icon_img_width(C);
    const int c0 = orc_icon_c0(C);
    float *xyz = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    orc_project(calib, pts, N, xyz);
    float *sdf = (float *)malloc(sizeof(float) * (size_t)N);
    float *nrm = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    float *cm = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    float *vs = (float *)malloc(sizeof(float) * (size_t)N);
    orc_cal_sdf(verts, V, faces, F, cmaps, vis, xyz, N, sdf, nrm, cm, vs, NULL, NULL);
    float *X = out_x ? out_x : (float *)malloc(sizeof(float) * (size_t)c0 * (size_t)N);
    /* outlier handling, HGPIFuNet.py:298-305.
     *   smpl_sdf[outlier] = sign(smpl_sdf[outlier])
     *   smpl_cmap[outlier.repeat(1,1,3)] = smpl_sdf[outlier].repeat(1,1,3)
     * smpl_sdf[outlier] is the 1-D list s[0..K) of outlier signs in point order; .repeat(1,1,3)
     * TILES that list three times ([s0..sK-1, s0..sK-1, s0..sK-1]) and the masked assignment
     * consumes it in row-major order, so the j-th outlier's channel k receives s[(3j+k) mod K] -
     * the signs of OTHER points of the same call.  That is what the reference computes, so it is
     * the default here (cmap_local == 0); cmap_local == 1 is the evidently intended per-point
     * rule cmap := own sign, kept as an opt-in variant. */
    float *olist = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    int64_t *orank = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    int64_t K = 0;
    for (int64_t i = 0; i < N; ++i) {
        orank[i] = -1;
        if (fabsf(sdf[i]) >= sdf_clip) {
            orank[i] = K;
            olist[K++] = (sdf[i] > 0.0f) ? 1.0f : ((sdf[i] < 0.0f) ? -1.0f : 0.0f);
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float s = sdf[i];
        float c3[3] = { cm[3 * i], cm[3 * i + 1], cm[3 * i + 2] };
        if (orank[i] >= 0) {
            const int64_t j = orank[i];
            s = olist[j];
            for (int k = 0; k < 3; ++k) c3[k] = cmap_local ? s : olist[(3 * j + k) % K];
        }
        float fall[64];
        orc_bilinear(feat, C, H, W, xyz[3 * i], xyz[3 * i + 1], fall);
        const int off = ((g_smpl_mask & 4) && vs[i] == 0.0f) ? half : 0;   /* feat_select, mesh_util.py:272-275 */
        float *x = X + (size_t)c0 * i;
        for (int k = 0; k < half; ++k) x[k] = fall[off + k];
        int o = half;
        x[o++] = s;
        if (g_smpl_mask & 1) { x[o] = c3[0]; x[o + 1] = c3[1]; x[o + 2] = c3[2]; o += 3; }
        if (g_smpl_mask & 2) { x[o] = nrm[3 * i]; x[o + 1] = nrm[3 * i + 1]; x[o + 2] = nrm[3 * i + 2]; }
    }
    orc_mlp_forward(mlp, X, N, c0, out_occ, accumulate_f64);
    orc_mask_in_cube(xyz, N, out_occ);
    if (!out_x) free(X);
    free(xyz); free(sdf); free(nrm); free(cm); free(vs); free(olist); free(orank);
}

/* The same call, but the MLP (and the output) only for the points listed in `subset` (M indices into
 * the call, ascending or not): the geometry half runs on ALL N points because the tiled outlier-cmap
 * assignment makes every point depend on the outlier signs of the whole call.  This is what lets
 * bench.py check a stratified sample of the 257^3 lattice against the checker in seconds.
 * out_occ [M], out_x [M, C/2+7] (optional). */
void orc_query_icon_subset(const float *verts, int64_t V, const int64_t *faces, int64_t F,
                           const float *cmaps, const float *vis,
                           const float *feat, int C, int H, int W,
                           const orc_mlp *mlp, float sdf_clip, const float *calib,
                           const float *pts, int64_t N, const int64_t *subset, int64_t M,
                           float *out_occ, float *out_x, int accumulate_f64, int cmap_local)
{
    const int half = icon_img_width(C);
    const int c0 = orc_icon_c0(C);
    float *xyz = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    orc_project(calib, pts, N, xyz);
    float *sdf = (float *)malloc(sizeof(float) * (size_t)N);
    float *nrm = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    float *cm = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    float *vs = (float *)malloc(sizeof(float) * (size_t)N);
    orc_cal_sdf(verts, V, faces, F, cmaps, vis, xyz, N, sdf, nrm, cm, vs, NULL, NULL);
    int8_t *olist = (int8_t *)malloc((size_t)(N > 0 ? N : 1));
    int64_t *orank = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    int64_t K = 0;
    for (int64_t i = 0; i < N; ++i) {
        orank[i] = -1;
        if (fabsf(sdf[i]) >= sdf_clip) {
            orank[i] = K;
            olist[K++] = (int8_t)((sdf[i] > 0.0f) ? 1 : ((sdf[i] < 0.0f) ? -1 : 0));
        }
    }
    float *X = out_x ? out_x : (float *)malloc(sizeof(float) * (size_t)c0 * (size_t)(M > 0 ? M : 1));
    float *sxyz = (float *)malloc(sizeof(float) * 3 * (size_t)(M > 0 ? M : 1));
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const int64_t i = subset[m];
        float s = sdf[i];
        float c3[3] = { cm[3 * i], cm[3 * i + 1], cm[3 * i + 2] };
        if (orank[i] >= 0) {
            const int64_t j = orank[i];
            s = (float)olist[j];
            for (int k = 0; k < 3; ++k) c3[k] = cmap_local ? s : (float)olist[(3 * j + k) % K];
        }
        float fall[64];
        orc_bilinear(feat, C, H, W, xyz[3 * i], xyz[3 * i + 1], fall);
        const int off = ((g_smpl_mask & 4) && vs[i] == 0.0f) ? half : 0;
        float *x = X + (size_t)c0 * m;
        for (int k = 0; k < half; ++k) x[k] = fall[off + k];
        int o = half;
        x[o++] = s;
        if (g_smpl_mask & 1) { x[o] = c3[0]; x[o + 1] = c3[1]; x[o + 2] = c3[2]; o += 3; }
        if (g_smpl_mask & 2) { x[o] = nrm[3 * i]; x[o + 1] = nrm[3 * i + 1]; x[o + 2] = nrm[3 * i + 2]; }
        sxyz[3 * m] = xyz[3 * i]; sxyz[3 * m + 1] = xyz[3 * i + 1]; sxyz[3 * m + 2] = xyz[3 * i + 2];
    }
    orc_mlp_forward(mlp, X, M, c0, out_occ, accumulate_f64);
    orc_mask_in_cube(sxyz, M, out_occ);
    if (!out_x) free(X);
    free(xyz); free(sdf); free(nrm); free(cm); free(vs); free(olist); free(orank); free(sxyz);
}

/* PaMIR branch (HGPIFuNet.py:348-354): point_feat = [index(im_feat, xy) (C), index(vol_feat, xyz) (Cv)]
 * PIFu  branch (HGPIFuNet.py:356-357): point_feat = [index(im_feat, xy) (C), z]      (vol == NULL) */
void orc_query_vol(const float *feat, int C, int H, int W,
                   const float *vol, int Cv, int Dv, int Hv, int Wv,
                   const orc_mlp *mlp, const float *calib,
                   const float *pts, int64_t N, float *out_occ, float *out_x, int accumulate_f64)
{
    const int c0 = C + (vol ? Cv : 1);
    float *xyz = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    orc_project(calib, pts, N, xyz);
    float *X = out_x ? out_x : (float *)malloc(sizeof(float) * (size_t)c0 * (size_t)N);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float *x = X + (size_t)c0 * i;
        orc_bilinear(feat, C, H, W, xyz[3 * i], xyz[3 * i + 1], x);
        if (vol) orc_trilinear(vol, Cv, Dv, Hv, Wv, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], x + C);
        else x[C] = xyz[3 * i + 2];
    }
    orc_mlp_forward(mlp, X, N, c0, out_occ, accumulate_f64);
    orc_mask_in_cube(xyz, N, out_occ);
    if (!out_x) free(X);
    free(xyz);
}

/* ---------------------------------------------------------------------------------------------
 * get_visibility (lib/dataset/mesh_util.py:280-316; call sites TestDataset.py:134-137,
 * PIFuDataset.py:436, lib/common/render.py:76): which SMPL vertices belong to a face that wins the
 * depth test at >= 1 pixel centre of a 4096^2 orthographic rasterisation.
 *
 * The rasteriser itself is pytorch3d's rasterize_meshes (requirements.txt:33, unpinned, absent from
 * /root/reference): PARITY UNPINNED for this leaf.  Restated from its published algorithm
 * (pytorch3d/csrc/rasterize_meshes + rasterization_utils, blur_radius 0, faces_per_pixel 1,
 * perspective_correct, cull_backfaces) with the reference's own pre/post-processing:
 *   xyz = (cat(xy, -z) + 1) / 2                      -> the mesh sits in the [0,1]^2 quadrant of NDC
 *   pixel centres at NDC -1 + (2i+1)/S
 *   per face: area = EF(v2; v0,v1); skip if area < 0 (back face) or |area| <= 1e-8
 *             w_k = EF(p; ...)/(area + 1e-8); perspective correction t0 = w0 z1 z2, t1 = z0 w1 z2,
 *             t2 = z0 z1 w2, w'_k = t_k / max(t0+t1+t2, 1e-8); pz = sum w'_k z_k; skip if pz < 0,
 *             if the centre is outside the face's xy bounding box, or unless w'_0, w'_1, w'_2 > 0
 *   the face with the smallest pz wins (lowest index on exact ties)
 *   vis[faces[unique(pix_to_face)]] = 1, where pix_to_face contains -1 (background: three quadrants
 *   of the image are always empty) and faces[-1] is the LAST face - its vertices are always marked
 *   (kept for bug compatibility).
 * All arithmetic float32, no contraction; the HIP kernel (icon_amd/csrc/vis_kernels.hip) evaluates the
 * same expressions, so the visible-vertex set is compared for equality.
 * ------------------------------------------------------------------------------------------- */
static inline float vis_ef(float px, float py, float ax, float ay, float bx, float by)
{
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

/* depth of face (X0..Z2) at pixel centre (px,py), or -1 if it does not cover it */
float orc_vis_face_depth(float px, float py, const float *X, const float *Y, const float *Z)
{
    const float eps = 1e-8f;
    const float area = vis_ef(X[2], Y[2], X[0], Y[0], X[1], Y[1]);
    if (area < 0.0f || fabsf(area) <= eps) return -1.0f;
    const float xmin = min_f(X[0], min_f(X[1], X[2])), xmax = max_f(X[0], max_f(X[1], X[2]));
    const float ymin = min_f(Y[0], min_f(Y[1], Y[2])), ymax = max_f(Y[0], max_f(Y[1], Y[2]));
    if (px < xmin || px > xmax || py < ymin || py > ymax) return -1.0f;
    const float den = area + eps;
    const float w0 = vis_ef(px, py, X[1], Y[1], X[2], Y[2]) / den;
    const float w1 = vis_ef(px, py, X[2], Y[2], X[0], Y[0]) / den;
    const float w2 = vis_ef(px, py, X[0], Y[0], X[1], Y[1]) / den;
    const float t0 = w0 * Z[1] * Z[2], t1 = Z[0] * w1 * Z[2], t2 = Z[0] * Z[1] * w2;
    const float dn = max_f(t0 + t1 + t2, eps);
    const float b0 = t0 / dn, b1 = t1 / dn, b2 = t2 / dn;
    const float pz = b0 * Z[0] + b1 * Z[1] + b2 * Z[2];
    if (pz < 0.0f) return -1.0f;
    if (!(b0 > 0.0f && b1 > 0.0f && b2 > 0.0f)) return -1.0f;
    return pz;
}

/* xy [V,2], z [V] exactly as passed to get_visibility(xy, z, faces); out_vis [V] in {0,1};
 * out_face (optional) [S/2 * S/2] winning face per pixel of the [0,1]^2 quadrant, -1 = background */
void orc_visibility(const float *xy, const float *z, int64_t V, const int64_t *faces, int64_t F, int S,
                    float *out_vis, int64_t *out_face)
{
    const int H = S / 2;                            /* pixels i >= S/2 have centres in (0, 1) */
    const size_t npx = (size_t)H * H;
    uint64_t *zb = (uint64_t *)malloc(npx * sizeof(uint64_t));
    for (size_t i = 0; i < npx; ++i) zb[i] = UINT64_MAX;
    for (int64_t f = 0; f < F; ++f) {
        float X[3], Y[3], Z[3];
        for (int k = 0; k < 3; ++k) {
            const int64_t v = faces[3 * f + k];
            X[k] = (xy[2 * v] + 1.0f) / 2.0f; Y[k] = (xy[2 * v + 1] + 1.0f) / 2.0f; Z[k] = (-z[v] + 1.0f) / 2.0f;
        }
        const float xmin = min_f(X[0], min_f(X[1], X[2])), xmax = max_f(X[0], max_f(X[1], X[2]));
        const float ymin = min_f(Y[0], min_f(Y[1], Y[2])), ymax = max_f(Y[0], max_f(Y[1], Y[2]));
        /* conservative pixel range (the exact tests are in orc_vis_face_depth) */
        int i0 = (int)floorf((xmin + 1.0f) * 0.5f * (float)S) - 1, i1 = (int)ceilf((xmax + 1.0f) * 0.5f * (float)S) + 1;
        int j0 = (int)floorf((ymin + 1.0f) * 0.5f * (float)S) - 1, j1 = (int)ceilf((ymax + 1.0f) * 0.5f * (float)S) + 1;
        if (i0 < H) i0 = H;
        if (j0 < H) j0 = H;
        if (i1 > S - 1) i1 = S - 1;
        if (j1 > S - 1) j1 = S - 1;
        for (int j = j0; j <= j1; ++j)
            for (int i = i0; i <= i1; ++i) {
                const float px = -1.0f + (float)(2 * i + 1) / (float)S, py = -1.0f + (float)(2 * j + 1) / (float)S;
                const float pz = orc_vis_face_depth(px, py, X, Y, Z);
                if (pz < 0.0f) continue;
                uint32_t bits; memcpy(&bits, &pz, 4);
                const uint64_t key = ((uint64_t)bits << 32) | (uint64_t)(uint32_t)f;
                uint64_t *slot = &zb[(size_t)(j - H) * H + (i - H)];
                if (key < *slot) *slot = key;
            }
    }
    for (int64_t v = 0; v < V; ++v) out_vis[v] = 0.0f;
    for (size_t i = 0; i < npx; ++i) {
        const int64_t f = (zb[i] == UINT64_MAX) ? -1 : (int64_t)(zb[i] & 0xffffffffu);
        if (out_face) out_face[i] = f;
        if (f >= 0) for (int k = 0; k < 3; ++k) out_vis[faces[3 * f + k]] = 1.0f;
    }
    if (F > 0 && S >= 2) for (int k = 0; k < 3; ++k) out_vis[faces[3 * (F - 1) + k]] = 1.0f;   /* faces[-1], see header */
    free(zb);
}
