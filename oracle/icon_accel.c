/*
 * oracle/icon_accel.c - accelerated, BIT-IDENTICAL versions of the two O(N*F) leaves of the checker.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as icon_oracle.c: nothing under icon_amd/ links or calls it).
 *
 * The definitions stay the linear scans of icon_oracle.c:
 *   orc_nearest_brute  - kaolin point_to_mesh_distance, call site lib/dataset/mesh_util.py:374
 *   orc_check_sign     - kaolin check_sign,             call site lib/dataset/mesh_util.py:393
 * This file answers the same two questions through a bounding-volume hierarchy / (y,z) ray bins so
 * that (i) bench.py's cpu_baseline times an exact ACCELERATED leaf, as SURVEY.md section 8(d) asks
 * ("use the KD-tree-accelerated exact leaf for timing fairness"), and (ii) the checker can afford the
 * whole 257^3 lattice (the reference's tiled outlier-cmap assignment, lib/net/HGPIFuNet.py:303-305,
 * couples every point of a call to the outlier signs of the others).
 *
 * Equality with the linear scans is not an approximation argument:
 *   nearest: the same orc_tri_dist2 on the same per-triangle constants; a subtree is skipped only if
 *            its box is farther than (sqrt(best) + 4e-6)^2 * (1 + 1e-6) - computed distances carry
 *            < 1e-6 absolute error for O(1) coordinates - so no triangle that could tie or beat the
 *            result of the scan is ever skipped; among float32-equal distances the lowest face index
 *            wins, as in the scan.  tests/test_oracle_leaves.py compares both bit for bit.
 *   sign   : a triangle contributes a crossing only if the query's (y,z) lies in its projected
 *            (closed) bounding box; the bins enumerate exactly a superset of those triangles and the
 *            same orc_ray_hit decides.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;
typedef struct {
    v3 a, b, ab, ac, bc;
    float i00, i11, ibc;
    float a00, a01, a11, inn;
} orc_tri;

/* icon_oracle.c */
void orc_tri_setup(const float *pa, const float *pb, const float *pc, orc_tri *t);
float orc_tri_dist2(const float *pp, const orc_tri *t);
int orc_ray_hit(const float *pp, int64_t ia, const float *pa, int64_t ib, const float *pb, int64_t ic, const float *pc);

typedef struct {
    float lo[3], hi[3];
    int32_t left, right;     /* inner: children; leaf: left = -1 - first, right = count */
} anode;

typedef struct orc_accel {
    int64_t V, F;
    float *verts;            /* copy [V,3] */
    int64_t *faces;          /* copy [F,3] */
    orc_tri *tri;            /* per face */
    int32_t *order;          /* leaf order -> face id */
    anode *nodes; int32_t n_nodes;
    /* ray bins over (y, z) */
    int gy, gz;
    float y0, z0, y1, z1, inv_y, inv_z;
    int32_t *bin_start, *bin_face;
} orc_accel;

static inline float minf(float a, float b) { return (b < a) ? b : a; }
static inline float maxf(float a, float b) { return (a > b) ? a : b; }

/* ---- BVH build: median split of the centroids along the widest axis, <= 4 triangles per leaf ---- */
typedef struct { float c[3]; float lo[3], hi[3]; int32_t f; } prim;
static int g_axis;
static int cmp_prim(const void *pa, const void *pb)
{
    const prim *a = (const prim *)pa, *b = (const prim *)pb;
    if (a->c[g_axis] < b->c[g_axis]) return -1;
    if (a->c[g_axis] > b->c[g_axis]) return 1;
    return (a->f > b->f) - (a->f < b->f);
}

static int32_t build_rec(orc_accel *A, prim *P, int32_t first, int32_t count)
{
    const int32_t id = A->n_nodes++;
    anode *n = &A->nodes[id];
    float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int k = 0; k < 3; ++k) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
    for (int32_t i = first; i < first + count; ++i)
        for (int k = 0; k < 3; ++k) {
            n->lo[k] = minf(n->lo[k], P[i].lo[k]); n->hi[k] = maxf(n->hi[k], P[i].hi[k]);
            clo[k] = minf(clo[k], P[i].c[k]); chi[k] = maxf(chi[k], P[i].c[k]);
        }
    if (count <= 4) { n->left = -1 - first; n->right = count; return id; }
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    g_axis = ax;
    qsort(P + first, (size_t)count, sizeof(prim), cmp_prim);
    const int32_t half = count / 2;
    const int32_t l = build_rec(A, P, first, half);
    const int32_t r = build_rec(A, P, first + half, count - half);
    A->nodes[id].left = l; A->nodes[id].right = r;       /* A->nodes does not move: allocated up front */
    return id;
}

static inline int cell_of(float v, float v0, float inv, int g)
{
    int c = (int)floorf((v - v0) * inv);
    if (c < 0) c = 0;
    if (c > g - 1) c = g - 1;
    return c;
}

orc_accel *orc_accel_build(const float *verts, int64_t V, const int64_t *faces, int64_t F)
{
    orc_accel *A = (orc_accel *)calloc(1, sizeof(orc_accel));
    A->V = V; A->F = F;
    A->verts = (float *)malloc(sizeof(float) * 3 * (size_t)V);
    memcpy(A->verts, verts, sizeof(float) * 3 * (size_t)V);
    A->faces = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)(F > 0 ? F : 1));
    memcpy(A->faces, faces, sizeof(int64_t) * 3 * (size_t)F);
    A->tri = (orc_tri *)malloc(sizeof(orc_tri) * (size_t)(F > 0 ? F : 1));
    prim *P = (prim *)malloc(sizeof(prim) * (size_t)(F > 0 ? F : 1));
    float ylo = INFINITY, yhi = -INFINITY, zlo = INFINITY, zhi = -INFINITY;
    for (int64_t f = 0; f < F; ++f) {
        const float *a = verts + 3 * faces[3 * f], *b = verts + 3 * faces[3 * f + 1], *c = verts + 3 * faces[3 * f + 2];
        orc_tri_setup(a, b, c, A->tri + f);
        for (int k = 0; k < 3; ++k) {
            P[f].lo[k] = minf(minf(a[k], b[k]), c[k]);
            P[f].hi[k] = maxf(maxf(a[k], b[k]), c[k]);
            P[f].c[k] = 0.5f * (P[f].lo[k] + P[f].hi[k]);
        }
        P[f].f = (int32_t)f;
        ylo = minf(ylo, P[f].lo[1]); yhi = maxf(yhi, P[f].hi[1]);
        zlo = minf(zlo, P[f].lo[2]); zhi = maxf(zhi, P[f].hi[2]);
    }
    A->nodes = (anode *)malloc(sizeof(anode) * (size_t)(2 * F + 2));
    A->n_nodes = 0;
    if (F > 0) build_rec(A, P, 0, (int32_t)F);
    A->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F > 0 ? F : 1));
    for (int64_t i = 0; i < F; ++i) A->order[i] = P[i].f;

    /* ray bins: ~2 cells per triangle */
    int g = (int)ceil(sqrt((double)(F > 0 ? F : 1) * 2.0));
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;
    A->gy = A->gz = g;
    A->y0 = ylo; A->y1 = yhi; A->z0 = zlo; A->z1 = zhi;
    A->inv_y = (yhi > ylo) ? (float)g / (yhi - ylo) : 0.0f;
    A->inv_z = (zhi > zlo) ? (float)g / (zhi - zlo) : 0.0f;
    const int64_t ncell = (int64_t)g * g;
    A->bin_start = (int32_t *)calloc((size_t)ncell + 1, sizeof(int32_t));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            int32_t run = 0;
            for (int64_t c = 0; c < ncell; ++c) { const int32_t n = A->bin_start[c]; A->bin_start[c] = run; run += n; }
            A->bin_start[ncell] = run;
            A->bin_face = (int32_t *)malloc(sizeof(int32_t) * (size_t)(run > 0 ? run : 1));
        }
        int32_t *fill = NULL;
        if (pass == 1) { fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncell); memcpy(fill, A->bin_start, sizeof(int32_t) * (size_t)ncell); }
        /* P was reordered by the BVH build: recompute the projected boxes from the faces themselves
         * (ascending face id inside every cell) */
        for (int64_t f = 0; f < F; ++f) {
            const float *a = verts + 3 * faces[3 * f], *b = verts + 3 * faces[3 * f + 1], *c = verts + 3 * faces[3 * f + 2];
            const float fy0 = minf(minf(a[1], b[1]), c[1]), fy1 = maxf(maxf(a[1], b[1]), c[1]);
            const float fz0 = minf(minf(a[2], b[2]), c[2]), fz1 = maxf(maxf(a[2], b[2]), c[2]);
            const int cy0 = cell_of(fy0, A->y0, A->inv_y, g), cy1 = cell_of(fy1, A->y0, A->inv_y, g);
            const int cz0 = cell_of(fz0, A->z0, A->inv_z, g), cz1 = cell_of(fz1, A->z0, A->inv_z, g);
            for (int cz = cz0; cz <= cz1; ++cz)
                for (int cy = cy0; cy <= cy1; ++cy) {
                    const int64_t cell = (int64_t)cz * g + cy;
                    if (pass == 0) A->bin_start[cell]++;
                    else A->bin_face[fill[cell]++] = (int32_t)f;
                }
        }
        free(fill);
    }
    free(P);
    return A;
}

void orc_accel_free(orc_accel *A)
{
    if (!A) return;
    free(A->verts); free(A->faces); free(A->tri); free(A->order); free(A->nodes); free(A->bin_start); free(A->bin_face);
    free(A);
}

static inline float box_d2(const anode *n, const float *p)
{
    const float dx = maxf(maxf(n->lo[0] - p[0], p[0] - n->hi[0]), 0.0f);
    const float dy = maxf(maxf(n->lo[1] - p[1], p[1] - n->hi[1]), 0.0f);
    const float dz = maxf(maxf(n->lo[2] - p[2], p[2] - n->hi[2]), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

static inline float prune_thr(float best)
{
    const float s = sqrtf(best) + 4e-6f;
    return s * s * 1.000001f;
}

void orc_accel_nearest(const orc_accel *A, const float *pts, int64_t N, float *out_d2, int64_t *out_idx)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
        const float *p = pts + 3 * i;
        float best = INFINITY, thr = INFINITY;
        int64_t bf = 0;
        int32_t stack[256];      /* depth of a median-split tree over F <= 2^60 triangles: < 64; two pushes per pop */
        int sp = 0;
        if (A->F > 0) stack[sp++] = 0;
        while (sp > 0) {
            const anode *n = &A->nodes[stack[--sp]];
            if (box_d2(n, p) > thr) continue;
            if (n->left < 0) {
                const int32_t first = -1 - n->left;
                for (int32_t k = first; k < first + n->right; ++k) {
                    const int64_t f = A->order[k];
                    const float d = orc_tri_dist2(p, A->tri + f);
                    if (d < best || (d == best && f < bf)) { best = d; bf = f; thr = prune_thr(best); }
                }
            } else {
                const anode *l = &A->nodes[n->left], *r = &A->nodes[n->right];
                const float dl = box_d2(l, p), dr = box_d2(r, p);
                /* far child first on the stack, near child popped next */
                if (dl <= dr) { if (dr <= thr) stack[sp++] = n->right; if (dl <= thr) stack[sp++] = n->left; }
                else          { if (dl <= thr) stack[sp++] = n->left; if (dr <= thr) stack[sp++] = n->right; }
            }
        }
        out_d2[i] = best; out_idx[i] = bf;
    }
}

/* ------------------------------------------------------------------------------------------
 * Tie diagnostics (test infrastructure for icon_sdf_query_ties / icon_work_set_tie_rule).
 * lib/dataset/mesh_util.py:374-390: the winner of point_to_mesh_distance decides the triangle
 * whose unclamped barycentric extrapolation gives norm / cmap / vis; the triangles around a
 * vertex or an edge are mathematically equidistant and the last bits of d^2 decide.
 *   orc_accel_nearest_ties: winner (S3), runner-up = next-smallest (d^2, face) key, and the
 *     number of float32 ulps between the two squared distances (clipped to 255; 255 and
 *     face2 = -1 when there is no second face).
 *   orc_set_tie_rule(1, u): orc_accel_nearest (and so orc_cal_sdf / orc_query_icon) returns,
 *     among the faces within u ulps of the minimum d^2, the one with the HIGHEST index and its
 *     own d^2 - the alternative resolution of the unpinned kaolin ties; (0, 0) = the definition.
 * ---------------------------------------------------------------------------------------- */
static int g_tie_rule = 0;
static uint32_t g_tie_ulps = 0;
void orc_set_tie_rule(int rule, int ulps) { g_tie_rule = rule; g_tie_ulps = (uint32_t)(ulps < 0 ? 0 : ulps); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* every face with d^2 <= lim (bit pattern order == value order for d^2 >= 0): visit(face, d2) */
static void accel_within(const orc_accel *A, const float *p, float best, uint32_t lim_bits, int64_t *io_face, float *io_d2)
{
    const float thr = prune_thr(best);
    int32_t stack[256];
    int sp = 0;
    if (A->F > 0) stack[sp++] = 0;
    while (sp > 0) {
        const anode *n = &A->nodes[stack[--sp]];
        if (box_d2(n, p) > thr) continue;
        if (n->left < 0) {
            const int32_t first = -1 - n->left;
            for (int32_t k = first; k < first + n->right; ++k) {
                const int64_t f = A->order[k];
                const float d = orc_tri_dist2(p, A->tri + f);
                if (f2u(d) <= lim_bits && f > *io_face) { *io_face = f; *io_d2 = d; }
            }
        } else {
            stack[sp++] = n->right; stack[sp++] = n->left;
        }
    }
}

void orc_accel_nearest_ties(const orc_accel *A, const float *pts, int64_t N, float *out_d2, int64_t *out_idx,
                            int64_t *out_idx2, uint8_t *out_ulps)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
        const float *p = pts + 3 * i;
        float best = INFINITY, second = INFINITY, thr = INFINITY;
        int64_t bf = -1, sf = -1;
        int32_t stack[256];
        int sp = 0;
        if (A->F > 0) stack[sp++] = 0;
        while (sp > 0) {
            const anode *n = &A->nodes[stack[--sp]];
            if (box_d2(n, p) > thr) continue;
            if (n->left < 0) {
                const int32_t first = -1 - n->left;
                for (int32_t k = first; k < first + n->right; ++k) {
                    const int64_t f = A->order[k];
                    const float d = orc_tri_dist2(p, A->tri + f);
                    if (!(d == d)) continue;                                       /* NaN never wins, never runs up */
                    if (bf < 0 || d < best || (d == best && f < bf)) {
                        second = best; sf = bf; best = d; bf = f; thr = prune_thr(best);
                    } else if (sf < 0 || d < second || (d == second && f < sf)) {
                        second = d; sf = f;
                    }
                }
            } else {
                const anode *l = &A->nodes[n->left], *r = &A->nodes[n->right];
                const float dl = box_d2(l, p), dr = box_d2(r, p);
                if (dl <= dr) { if (dr <= thr) stack[sp++] = n->right; if (dl <= thr) stack[sp++] = n->left; }
                else          { if (dl <= thr) stack[sp++] = n->left; if (dr <= thr) stack[sp++] = n->right; }
            }
        }
        out_d2[i] = best; out_idx[i] = bf < 0 ? 0 : bf;
        /* the runner-up is exact when it lies within the pruning bound of the winner - always the case for the
         * <= 255 ulps that are reported; farther away it is reported as 255 whatever was seen */
        uint32_t u = 255;
        if (sf >= 0 && second >= best) { const uint32_t g = f2u(second) - f2u(best); u = g < 255u ? g : 255u; }
        out_idx2[i] = sf;                                   /* -1: the mesh has a single face */
        out_ulps[i] = (uint8_t)u;
    }
}

void orc_accel_apply_tie_rule(const orc_accel *A, const float *pts, int64_t N, float *io_d2, int64_t *io_idx)
{
    if (!g_tie_rule) return;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
        int64_t f = -1; float d = io_d2[i];
        accel_within(A, pts + 3 * i, io_d2[i], f2u(io_d2[i]) + g_tie_ulps, &f, &d);
        if (f >= 0) { io_idx[i] = f; io_d2[i] = d; }
    }
}

void orc_accel_check_sign(const orc_accel *A, const float *pts, int64_t N, uint8_t *inside)
{
    const float *verts = A->verts;
    const int64_t *faces = A->faces;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < N; ++i) {
        const float *p = pts + 3 * i;
        int cnt = 0;
        if (p[1] >= A->y0 && p[1] <= A->y1 && p[2] >= A->z0 && p[2] <= A->z1) {
            const int cy = cell_of(p[1], A->y0, A->inv_y, A->gy), cz = cell_of(p[2], A->z0, A->inv_z, A->gz);
            const int64_t cell = (int64_t)cz * A->gy + cy;
            for (int32_t k = A->bin_start[cell]; k < A->bin_start[cell + 1]; ++k) {
                const int64_t f = A->bin_face[k];
                const int64_t ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
                cnt += orc_ray_hit(p, ia, verts + 3 * ia, ib, verts + 3 * ib, ic, verts + 3 * ic);
            }
        }
        inside[i] = (uint8_t)(cnt & 1);
    }
}

/* ---------------------------------------------------------------------------------------------
 * PaMIR semantic voxelisation - restatement of voxelize_cuda.forward_semantic_voxelization
 * (YuliangXiu/neural_voxelization_layer, unpinned in requirements.txt:34; NOT under /root/reference),
 * call site lib/net/voxelize.py:57-59, arguments built by Voxelization.forward (:119-137), parameters
 * volume_res = 128, sigma = 0.05 (lib/net/HGPIFuNet.py:109-118).
 *
 * PARITY UNPINNED: the CUDA source is absent and the reference holds no test vector for it.  The
 * semantics are therefore DEFINED here, from the call site's contract and the published description of the
 * layer (PaMIR, "semantic volume": tetrahedralised SMPL -> occupancy; every inside voxel gets the
 * Gaussian-weighted average of the semantic codes of the SURFACE vertices, weight_sum starts at 1e-3 as
 * lib/net/voxelize.py:49-51 allocates it; outside voxels stay 0 as :44-48 initialises them):
 *   voxel (z,y,x) has centre p = ((x,y,z) + 0.5) / R - 0.5        (the volume spans [-0.5, 0.5]^3: the caller
 *                                                                 scales the body by 0.5, TestDataset.py:172-179)
 *   occ  = 1 iff p lies in (or on the boundary of) at least one tetrahedron
 *   sem  = occ * sum_v w_v code_v / (1e-3 + sum_v w_v),  w_v = exp(-|p - v|^2 / (2 sigma^2)),  v over the surface vertices
 * Output layout [z][y][x][3] (the reference's "(batch_size, z_dims, y_dims, x_dims, channel_num)").
 *
 * Inside test (float32, no contraction; the HIP kernel evaluates the same expressions): for the faces
 * (b,c,d), (a,d,c), (a,b,d), (a,c,b) of the tetrahedron, s_k = dot(cross(e1, e2), p - origin); p is inside iff
 * all four s_k have the sign of the tetrahedron's own orientation or are zero (degenerate: volume 0 -> never).
 * ------------------------------------------------------------------------------------------- */
static inline float det3(const float *o, const float *u, const float *v, const float *p)
{
    const float ux = u[0] - o[0], uy = u[1] - o[1], uz = u[2] - o[2];
    const float vx = v[0] - o[0], vy = v[1] - o[1], vz = v[2] - o[2];
    const float px = p[0] - o[0], py = p[1] - o[1], pz = p[2] - o[2];
    const float cx = fmaf(uy, vz, -(uz * vy)), cy = fmaf(uz, vx, -(ux * vz)), cz = fmaf(ux, vy, -(uy * vx));
    return fmaf(cz, pz, fmaf(cy, py, cx * px));
}

void orc_semantic_voxelize(const float *verts, int64_t V, int64_t V_surf, const float *code,
                           const int64_t *tets, int64_t T, int res, float sigma, float *out /* [res][res][res][3] */,
                           uint8_t *occ_out /* optional [res][res][res] */)
{
    const int64_t n = (int64_t)res * res * res;
    uint8_t *occ = (uint8_t *)calloc((size_t)n, 1);
    const float inv = 1.0f / (float)res;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t ia = tets[4 * t], ib = tets[4 * t + 1], ic = tets[4 * t + 2], id = tets[4 * t + 3];
        if (ia < 0 || ib < 0 || ic < 0 || id < 0 || ia >= V || ib >= V || ic >= V || id >= V) continue;
        const float *a = verts + 3 * ia, *b = verts + 3 * ib, *c = verts + 3 * ic, *d = verts + 3 * id;
        const float vol = det3(a, b, c, d);
        if (vol == 0.0f) continue;
        int lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            const float mn = minf(minf(a[k], b[k]), minf(c[k], d[k])), mx = maxf(maxf(a[k], b[k]), maxf(c[k], d[k]));
            lo[k] = (int)floorf((mn + 0.5f) * (float)res - 0.5f);
            hi[k] = (int)ceilf((mx + 0.5f) * (float)res - 0.5f);
            if (lo[k] < 0) lo[k] = 0;
            if (hi[k] > res - 1) hi[k] = res - 1;
        }
        for (int z = lo[2]; z <= hi[2]; ++z)
            for (int y = lo[1]; y <= hi[1]; ++y)
                for (int x = lo[0]; x <= hi[0]; ++x) {
                    const float p[3] = { ((float)x + 0.5f) * inv - 0.5f, ((float)y + 0.5f) * inv - 0.5f, ((float)z + 0.5f) * inv - 0.5f };
                    const float s0 = det3(b, c, d, p), s1 = det3(a, d, c, p), s2 = det3(a, b, d, p), s3 = det3(a, c, b, p);
                    /* orientation of (b,c,d | a) is -vol: inside <=> every s_k has the sign of -vol (or is zero) */
                    const int in = (vol > 0.0f) ? (s0 <= 0.0f && s1 <= 0.0f && s2 <= 0.0f && s3 <= 0.0f)
                                                : (s0 >= 0.0f && s1 >= 0.0f && s2 >= 0.0f && s3 >= 0.0f);
                    if (in) occ[((int64_t)z * res + y) * res + x] = 1;
                }
    }
    const float k2 = 1.0f / (2.0f * sigma * sigma);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        float *o = out + 3 * i;
        o[0] = o[1] = o[2] = 0.0f;
        if (!occ[i]) continue;
        const int x = (int)(i % res), y = (int)((i / res) % res), z = (int)(i / ((int64_t)res * res));
        const float px = ((float)x + 0.5f) * inv - 0.5f, py = ((float)y + 0.5f) * inv - 0.5f, pz = ((float)z + 0.5f) * inv - 0.5f;
        double ws = 1e-3, s0 = 0.0, s1 = 0.0, s2 = 0.0;          /* float64 accumulation: the checker's reference value */
        for (int64_t v = 0; v < V_surf; ++v) {
            const float dx = px - verts[3 * v], dy = py - verts[3 * v + 1], dz = pz - verts[3 * v + 2];
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            const double wgt = exp(-(double)d2 * (double)k2);
            ws += wgt; s0 += wgt * code[3 * v]; s1 += wgt * code[3 * v + 1]; s2 += wgt * code[3 * v + 2];
        }
        o[0] = (float)(s0 / ws); o[1] = (float)(s1 / ws); o[2] = (float)(s2 / ws);
    }
    if (occ_out) memcpy(occ_out, occ, (size_t)n);
    free(occ);
}
