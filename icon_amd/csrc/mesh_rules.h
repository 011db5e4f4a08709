// mesh_rules.h - the decisions of the per-image BVH / ray-bin build that the device builder (mesh_device.hip)
// and the host builder (mesh_build.cpp: the checker of the device build, and the path for ICON_AMD_MESH_BUILD=host)
// must make IDENTICALLY, so that both emit the same arrays bit for bit (tests/test_gpu_mesh_build.py compares them).
//
// float32 / float64 arithmetic without contraction (both translation units are compiled with -ffp-contract=off).
#pragma once
#include "common.h"

namespace icon {

// ---- depth bound ------------------------------------------------------------------------------------------
// The traversal stacks (one LDS word per level per wave; 64 frontier entries per level in the one-wave-per-point
// search) are sized on the host BEFORE the tree exists, so the builder guarantees a bound that depends on F only:
// a node of n triangles at depth d may take a SAH split only if, with median splits from there on, no leaf would
// end up deeper than depth_bound(F); otherwise it is halved by position.
__host__ __device__ inline int ilog2_ceil(int64_t n) { int l = 0; while (((int64_t)1 << l) < n) ++l; return l; }
__host__ __device__ inline int median_levels(int n) { int l = 0; while (n > kLeafMax) { n = (n + 1) / 2; ++l; } return l; }
__host__ __device__ inline int depth_bound(int64_t F)
{
    const int b = ilog2_ceil(F < 2 ? 2 : F) + 10;
    return b < kStackDepth - 2 ? b : kStackDepth - 2;
}
__host__ __device__ inline bool force_median(int depth, int n, int bound) { return depth + 1 + median_levels(n - 1) > bound; }

// ---- input hygiene ------------------------------------------------------------------------------------------
// a coordinate the builder refuses (host build: error; device build: status bit, the coordinate counts as 0)
__host__ __device__ inline bool bad_coord(float v) { return !(v >= -1e6f && v <= 1e6f); }
// -0 -> +0: min / max then give the same bits whatever the order of the operands
__host__ __device__ inline float canon(float v) { return v + 0.0f; }

// ---- binned SAH ------------------------------------------------------------------------------------------
constexpr int kSahBins = 16;
__host__ __device__ inline int sah_bin(float c, float lo, float ext)
{
    int b = (int)((c - lo) / ext * (float)kSahBins);
    return b < 0 ? 0 : (b > kSahBins - 1 ? kSahBins - 1 : b);
}
struct BoxD { float lo[3], hi[3]; };
__host__ __device__ inline double box_area(const float lo[3], const float hi[3])
{
    const double dx0 = (double)hi[0] - lo[0], dy0 = (double)hi[1] - lo[1], dz0 = (double)hi[2] - lo[2];
    const double dx = dx0 > 0.0 ? dx0 : 0.0, dy = dy0 > 0.0 ? dy0 : 0.0, dz = dz0 > 0.0 ? dz0 : 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}

// ---- (y,z) ray bins: square cells, about two per triangle, float32 throughout (IEEE sqrt / divide on both sides) ----
struct BinGrid { float y0, z0, y1, z1, inv_y, inv_z; int gy, gz; };
constexpr float kBinEps = 1e-5f;
__host__ __device__ inline BinGrid bin_grid(const float box_lo[3], const float box_hi[3], int64_t F)
{
    BinGrid g;
    g.y0 = box_lo[1] - 4 * kBinEps; g.y1 = box_hi[1] + 4 * kBinEps;
    g.z0 = box_lo[2] - 4 * kBinEps; g.z1 = box_hi[2] + 4 * kBinEps;
    const float ey = g.y1 - g.y0, ez = g.z1 - g.z0;
    float a = ey * ez;
    if (!(a > 1e-12f)) a = 1e-12f;
    const float cell = sqrtf(a / (2.0f * (float)F));
    int gy = (int)ceilf(ey / cell), gz = (int)ceilf(ez / cell);
    g.gy = gy < 1 ? 1 : (gy > 2048 ? 2048 : gy);
    g.gz = gz < 1 ? 1 : (gz > 2048 ? 2048 : gz);
    g.inv_y = (float)g.gy / ey; g.inv_z = (float)g.gz / ez;
    return g;
}
__host__ __device__ inline int bin_cell_of(float v, float v0, float inv, int g)
{
    const int c = (int)floorf((v - v0) * inv);
    return c < 0 ? 0 : (c > g - 1 ? g - 1 : c);
}
// cells and list entries the arena reserves (gy * gz <= 2 F + gy + gz + 1; entries: 48 per triangle on average
// is ~3x what a body mesh needs - beyond it the inside tests fall back to the brute-force parity count)
__host__ __device__ inline int64_t bin_cells_cap(int64_t F) { return 2 * F + 4100; }
__host__ __device__ inline int64_t bin_entries_cap(int64_t F) { return 48 * F + 2 * bin_cells_cap(F); }

// S2 per-triangle constants (same float32 operation sequence as the checker's orc_tri_setup)
__host__ __device__ inline float dot3r(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
__host__ __device__ inline void tri_setup(const float *a, const float *b, const float *c, int32_t face, TriPre &t)
{
    for (int k = 0; k < 3; ++k) { t.a[k] = a[k]; t.b[k] = b[k]; t.ab[k] = b[k] - a[k]; t.ac[k] = c[k] - a[k]; t.bc[k] = c[k] - b[k]; }
    t.a00 = dot3r(t.ab, t.ab); t.a01 = dot3r(t.ab, t.ac); t.a11 = dot3r(t.ac, t.ac);
    const float b11 = dot3r(t.bc, t.bc);
    t.i00 = (t.a00 > 0.0f) ? 1.0f / t.a00 : 0.0f;
    t.i11 = (t.a11 > 0.0f) ? 1.0f / t.a11 : 0.0f;
    t.ibc = (b11 > 0.0f) ? 1.0f / b11 : 0.0f;
    const float nn = fmaf(t.a00, t.a11, -(t.a01 * t.a01));
    // zero area (or a sliver whose Gram determinant rounds to <= 0): NaN makes both barycentrics NaN, every
    // comparison of the inside test false, and the distance the minimum over the three edge segments - exact
    t.inn = (nn > 0.0f) ? 1.0f / nn : __builtin_nanf("");
    t.face = face; t.pad = 0;
}

// ---- arena layout (one device allocation per mesh) ------------------------------------------------------------
constexpr int kTopLevels = 7;          // BVH levels 0..6 are split by multi-workgroup kernels (k_bvh_bin / k_bvh_part)
constexpr int kSubMax = 256;           // subtrees of at most this many triangles are finished by ONE workgroup in LDS (k_bvh_sub)
// (measured on the SMPL-size body, MI355X: 5 levels / 1024 -> k_bvh_sub 257 us, the top levels 75 us; 7 / 256 -> 84 + 122 us:
//  the subtree kernel pays a few microseconds per node PER WAVEFRONT, more, smaller subtrees spread the nodes over more CUs)
constexpr int kChunk = 256;            // triangles per workgroup of the top-level kernels
constexpr int kAdjCap = 16;            // incident (face, corner) entries kept per vertex (more: the vertex scans all faces)
constexpr int kHistWords = 3 * kSahBins * 13;   // per node: [axis][bin][count, lo xyz, hi xyz (triangle boxes), lo xyz, hi xyz (centroids)] as ordered-uint codes
constexpr int kTaskSlots = (1 << (kTopLevels + 1)) - 1;   // task records of levels 0..kTopLevels

struct BTask {                         // one node of the top of the tree while it is being built
    int32_t begin, end, depth, parent, side, kind, buf, from_atomics;
    float box[6], cb[6];               // lo xyz, hi xyz of the triangle boxes / of the centroids
    uint32_t ubox[12];                 // the same as ordered-uint maxima (positional splits: accumulated by every chunk)
};
static_assert(sizeof(BTask) == 128, "BTask layout");

struct BuildHdr {                      // zeroed before every build
    int32_t n_sub;                     // entries of the subtree queue
    int32_t pad[3];
};

struct MeshLayout {
    size_t dyn, hdr, valence, leaf_cnt, tasks, hist, cell_count, cell_cursor, zero_end;   // [dyn, zero_end) is zeroed per build
    size_t vnormals, nodes, leaves, tris, attr, slot2face, face2slot, bin_start, bin_slots;
    size_t tbox, cen, order0, order1, adj, chunkcnt, subq, sublist, bounds_part, total;
    int64_t nck;                       // chunk records per top level
};
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline MeshLayout mesh_layout(int64_t V, int64_t F)
{
    MeshLayout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes); return at; };
    const int64_t cells = bin_cells_cap(F);
    L.nck = F / kChunk + (1 << kTopLevels) + 2;
    L.dyn = take(sizeof(MeshDyn));
    L.hdr = take(sizeof(BuildHdr));
    L.valence = take(sizeof(int32_t) * V);
    L.leaf_cnt = take((size_t)F);
    L.tasks = take(sizeof(BTask) * kTaskSlots);
    L.hist = take(sizeof(uint32_t) * kHistWords * ((1 << kTopLevels) - 1));
    L.cell_count = take(sizeof(int32_t) * (cells + 1));
    L.cell_cursor = take(sizeof(int32_t) * cells);
    L.zero_end = o;
    L.vnormals = take(sizeof(float) * 3 * V);
    L.nodes = take(sizeof(BvhNode) * F);
    L.leaves = take(sizeof(LeafRec) * F);
    L.tris = take(sizeof(TriRec) * F);
    L.attr = take(sizeof(TriAttr) * F);
    L.slot2face = take(sizeof(int32_t) * F);
    L.face2slot = take(sizeof(int32_t) * F);
    L.bin_start = take(sizeof(int32_t) * (cells + 1));
    L.bin_slots = take(sizeof(int32_t) * bin_entries_cap(F));
    L.tbox = take(sizeof(float) * 6 * F);
    L.cen = take(sizeof(float) * 3 * F);
    L.order0 = take(sizeof(int32_t) * F);
    L.order1 = take(sizeof(int32_t) * F);
    L.adj = take(sizeof(int32_t) * kAdjCap * V);
    L.chunkcnt = take(sizeof(int32_t) * 3 * kSahBins * L.nck * kTopLevels);
    L.subq = take(sizeof(int32_t) * (kTaskSlots + 1));
    L.sublist = take((size_t)80 * 2 * (F / 5 + 4));            // task lists of subtrees too large for LDS (mesh_device.hip: STask, 80 B)
    L.bounds_part = take(sizeof(uint32_t) * 12 * ((size_t)F / 256 + 1));   // per-workgroup mesh bounds of k_face_prep
    L.total = o;
    return L;
}

}  // namespace icon
