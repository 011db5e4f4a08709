// mcubes.cpp - host marching cubes for Seg3dLossless.export_mesh
// (reference lib/common/seg3d_lossless.py:583-604: marching cubes at 0.5 on occupancys[1:,1:,1:],
//  verts[:, [2,1,0]] -> (x,y,z), faces[:, [0,2,1]]).
//
// The reference delegates to kaolin.ops.conversions.voxelgrids_to_trianglemeshes (<= 256^3) or
// PyMCubes (larger), neither of which is in the reference tree.  This is a from-scratch
// implementation: the 256-case triangle table is GENERATED at start-up from the cube topology
// (per-face marching squares with the "separate the inside corners" rule on ambiguous faces, which
// depends only on the face's own corner states and therefore agrees between the two cubes
// sharing the face => the output is watertight), loops are fan-triangulated, vertices are placed
// by linear interpolation and shared between cubes through rolling per-plane edge maps.
// Orientation: triangle normals point from inside (occ > level) to outside.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

// corner id = x + 2y + 4z.  Faces with corners in CCW order seen from outside the cube.
const int kFaces[6][4] = {
    {4, 5, 7, 6},  // z = 1
    {0, 2, 3, 1},  // z = 0
    {1, 3, 7, 5},  // x = 1
    {0, 4, 6, 2},  // x = 0
    {2, 6, 7, 3},  // y = 1
    {0, 1, 5, 4},  // y = 0
};

// 12 cube edges: (corner a, corner b), a < b, differing in exactly one bit.
struct Edge { int a, b; };
Edge g_edges[12];
int g_edge_of[8][8];
// per case: triangles as triples of cube-edge ids
std::once_flag g_once;
std::vector<std::array<int, 3>> g_case_tris[256];

bool edges_share_face(int e1, int e2)
{
    for (int f = 0; f < 6; ++f) {
        int hit = 0;
        for (int k = 0; k < 4; ++k) {
            const int c = kFaces[f][k];
            hit += (c == g_edges[e1].a) + (c == g_edges[e1].b) + (c == g_edges[e2].a) + (c == g_edges[e2].b);
        }
        if (hit == 4) return true;
    }
    return false;
}

// Ear clipping of one loop.  A diagonal between two loop vertices that lie on a common cube face
// would sit IN that face, where the neighbouring cube can produce the same diagonal (ambiguous
// faces carry two segments): four triangles on one edge.  So ears are only cut along diagonals
// that go through the cube's interior; backtracking finds such an order whenever one exists.
bool clip(std::vector<int> &poly, std::vector<std::array<int, 3>> &out, bool strict)
{
    const size_t n = poly.size();
    if (n < 3) return true;
    if (n == 3) { out.push_back({poly[0], poly[2], poly[1]}); return true; }   // reversed: outward normals
    for (size_t i = 0; i < n; ++i) {
        const int a = poly[i], b = poly[(i + 1) % n], c2 = poly[(i + 2) % n];
        if (strict && edges_share_face(a, c2)) continue;
        std::vector<int> rest;
        for (size_t k = 0; k < n; ++k) if (k != (i + 1) % n) rest.push_back(poly[k]);
        const size_t mark = out.size();
        out.push_back({a, c2, b});
        if (clip(rest, out, strict)) return true;
        out.resize(mark);
    }
    return false;
}

void triangulate(const std::vector<int> &loop, std::vector<std::array<int, 3>> &out)
{
    std::vector<int> poly = loop;
    if (!clip(poly, out, true)) { poly = loop; clip(poly, out, false); }
}

void build_tables()
{
    int ne = 0;
    memset(g_edge_of, -1, sizeof(g_edge_of));
    for (int a = 0; a < 8; ++a)
        for (int bit = 1; bit < 8; bit <<= 1) {
            const int b = a | bit;
            if (b != a) { g_edges[ne] = {a, b}; g_edge_of[a][b] = g_edge_of[b][a] = ne; ++ne; }
        }
    for (int c = 0; c < 256; ++c) {
        // directed segments (from edge -> to edge), inside region on the left seen from outside
        int next[12];
        for (int e = 0; e < 12; ++e) next[e] = -1;
        for (int f = 0; f < 6; ++f) {
            bool in[4];
            int nin = 0;
            for (int k = 0; k < 4; ++k) { in[k] = (c >> kFaces[f][k]) & 1; nin += in[k]; }
            if (nin == 0 || nin == 4) continue;
            // every maximal cyclic run of inside corners: segment from its exit edge to its entry edge
            for (int k = 0; k < 4; ++k) {
                if (in[k] && !in[(k + 1) & 3]) {             // run ends at corner k: exit edge (k, k+1)
                    int s = k;
                    while (in[(s + 3) & 3]) s = (s + 3) & 3;  // run starts at corner s
                    const int exit_e = g_edge_of[kFaces[f][k]][kFaces[f][(k + 1) & 3]];
                    const int entry_e = g_edge_of[kFaces[f][(s + 3) & 3]][kFaces[f][s]];
                    next[exit_e] = entry_e;
                }
            }
        }
        // follow the loops, fan-triangulate, reverse so that normals point outside
        bool used[12] = {false};
        for (int e0 = 0; e0 < 12; ++e0) {
            if (next[e0] < 0 || used[e0]) continue;
            std::vector<int> loop;
            int e = e0;
            while (!used[e]) { used[e] = true; loop.push_back(e); e = next[e]; }
            triangulate(loop, g_case_tris[c]);
        }
    }
}

}  // namespace

namespace icon {
// flattened case table for mc_device.hip: [256][16] = triangle count, then up to 5 x 3 cube-edge ids;
// edges[12][2] = (corner a, corner b), corner id = x + 2y + 4z
void mc_tables(int8_t table[256][16], int8_t edges[12][2])
{
    std::call_once(g_once, build_tables);
    for (int c = 0; c < 256; ++c) {
        const auto &t = g_case_tris[c];
        table[c][0] = (int8_t)t.size();
        for (size_t k = 0; k < 5; ++k)
            for (int q = 0; q < 3; ++q) table[c][1 + 3 * k + q] = (k < t.size()) ? (int8_t)t[k][q] : 0;
    }
    for (int e = 0; e < 12; ++e) { edges[e][0] = (int8_t)g_edges[e].a; edges[e][1] = (int8_t)g_edges[e].b; }
}
int mc_max_tris()
{
    std::call_once(g_once, build_tables);
    size_t m = 0;
    for (int c = 0; c < 256; ++c) m = std::max(m, g_case_tris[c].size());
    return (int)m;
}
}  // namespace icon

namespace {

struct McResult {
    const float *occ = nullptr; int res = 0; float level = 0.f;
    std::vector<float> verts; std::vector<int64_t> faces;
    bool valid = false;
};
thread_local McResult g_last;

void run_mc(const float *occ, int res, float level, McResult &out)
{
    std::call_once(g_once, build_tables);
    out.verts.clear(); out.faces.clear();
    const int n = res - 1;                       // cropped grid occ[1:,1:,1:] has n^3 samples
    if (n < 2) return;
    auto at = [&](int z, int y, int x) -> float { return occ[((size_t)(z + 1) * res + (y + 1)) * res + (x + 1)]; };
    const size_t plane = (size_t)n * n;
    std::vector<int64_t> xe[2], ye[2], ze;
    for (int k = 0; k < 2; ++k) { xe[k].assign(plane, -1); ye[k].assign(plane, -1); }
    ze.assign(plane, -1);
    auto add_vertex = [&](float x, float y, float z) -> int64_t {
        out.verts.push_back(x); out.verts.push_back(y); out.verts.push_back(z);
        return (int64_t)(out.verts.size() / 3 - 1);
    };
    auto lerp = [&](float v0, float v1) -> float {
        const float d = v1 - v0;
        float t = (d != 0.f) ? (level - v0) / d : 0.5f;
        return std::fmin(std::fmax(t, 0.f), 1.f);
    };
    // vertices on x- and y-edges of plane z
    auto plane_vertices = [&](int z, std::vector<int64_t> &xev, std::vector<int64_t> &yev) {
        std::fill(xev.begin(), xev.end(), -1); std::fill(yev.begin(), yev.end(), -1);
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const float v = at(z, y, x);
                const bool in = v > level;
                if (x + 1 < n) { const float w = at(z, y, x + 1); if ((w > level) != in) xev[(size_t)y * n + x] = add_vertex(x + lerp(v, w), (float)y, (float)z); }
                if (y + 1 < n) { const float w = at(z, y + 1, x); if ((w > level) != in) yev[(size_t)y * n + x] = add_vertex((float)x, y + lerp(v, w), (float)z); }
            }
    };
    plane_vertices(0, xe[0], ye[0]);
    for (int z = 0; z + 1 < n; ++z) {
        const int cur = z & 1, nxt = cur ^ 1;
        plane_vertices(z + 1, xe[nxt], ye[nxt]);
        std::fill(ze.begin(), ze.end(), -1);
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const float v = at(z, y, x), w = at(z + 1, y, x);
                if ((v > level) != (w > level)) ze[(size_t)y * n + x] = add_vertex((float)x, (float)y, z + lerp(v, w));
            }
        for (int y = 0; y + 1 < n; ++y)
            for (int x = 0; x + 1 < n; ++x) {
                int c = 0;
                for (int k = 0; k < 8; ++k)
                    if (at(z + ((k >> 2) & 1), y + ((k >> 1) & 1), x + (k & 1)) > level) c |= 1 << k;
                if (c == 0 || c == 255) continue;
                for (const auto &t : g_case_tris[c]) {
                    int64_t id[3];
                    for (int q = 0; q < 3; ++q) {
                        const Edge &e = g_edges[t[q]];
                        const int ax = e.a & 1, ay = (e.a >> 1) & 1, az = (e.a >> 2) & 1;
                        const int dir = e.a ^ e.b;   // 1: x-edge, 2: y-edge, 4: z-edge
                        const size_t cell = (size_t)(y + ay) * n + (x + ax);
                        if (dir == 1) id[q] = xe[az ? nxt : cur][cell];
                        else if (dir == 2) id[q] = ye[az ? nxt : cur][cell];
                        else id[q] = ze[cell];
                    }
                    if (id[0] < 0 || id[1] < 0 || id[2] < 0) continue;   // cannot happen for consistent data
                    if (id[0] == id[1] || id[1] == id[2] || id[0] == id[2]) continue;
                    out.faces.push_back(id[0]); out.faces.push_back(id[1]); out.faces.push_back(id[2]);
                }
            }
    }
}

}  // namespace

extern "C" int icon_export_mesh(const float *h_occ, int res, float level,
                                float *h_verts, int64_t *n_verts, int64_t *h_faces, int64_t *n_faces)
{
    ICON_ARG(h_occ && n_verts && n_faces, "icon_export_mesh: null argument");
    ICON_ARG(res >= 3, "icon_export_mesh: res must be >= 3");
    McResult &r = g_last;
    const bool fill = (h_verts != nullptr && h_faces != nullptr);
    if (!(fill && r.valid && r.occ == h_occ && r.res == res && r.level == level)) {
        run_mc(h_occ, res, level, r);
        r.occ = h_occ; r.res = res; r.level = level; r.valid = true;
    }
    const int64_t nv = (int64_t)(r.verts.size() / 3), nf = (int64_t)(r.faces.size() / 3);
    if (fill) {
        ICON_ARG(*n_verts >= nv && *n_faces >= nf, "icon_export_mesh: output buffers too small");
        if (nv) memcpy(h_verts, r.verts.data(), sizeof(float) * 3 * (size_t)nv);
        if (nf) memcpy(h_faces, r.faces.data(), sizeof(int64_t) * 3 * (size_t)nf);
        r.valid = false; r.verts.clear(); r.verts.shrink_to_fit(); r.faces.clear(); r.faces.shrink_to_fit();
    }
    *n_verts = nv; *n_faces = nf;
    return ICON_OK;
}
