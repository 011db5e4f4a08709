// mesh_build.cpp - per-image body-mesh preparation: the C entry points (icon_mesh_create / _create_arena / _status /
// _stats) and the HOST builder.
//
// Replaces the per-call prologue of cal_sdf_batch (reference lib/dataset/mesh_util.py:367-372):
// vertex normals (pytorch3d Meshes.verts_normals_padded) and the four face_vertices() gathers
// (lib/common/render_utils.py:149-163), and builds what the query kernels traverse: a BVH2 over
// the triangles (binned SAH) and a (y,z) bin grid for the +x ray-parity inside test.
//
// Since round 4 the build runs ON THE DEVICE (mesh_device.hip: no copy of the mesh to the host, no allocation, no
// synchronisation); the host builder below emits the same arrays bit for bit and stays as its checker
// (tests/test_gpu_mesh_build.py) and as the ICON_AMD_MESH_BUILD=host path.
//
// float32 arithmetic here follows the spec in DESIGN.md §"Arithmetic spec" (explicit fmaf, no
// other contraction) so that the normals are bit-identical to the checker's.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <numeric>
#include <thread>

#include "common.h"
#include "mesh_rules.h"

namespace icon {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }
const char *last_error_cstr() { return g_err.c_str(); }

void debug_sync(const char *what, hipStream_t st)
{
    static const int on = getenv("ICON_AMD_DEBUG_SYNC") ? atoi(getenv("ICON_AMD_DEBUG_SYNC")) : 0;
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const double since = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - last).count();
    fprintf(stderr, "[icon_amd] (+%.3f ms on the host) %s ...", since, what); fflush(stderr);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipStreamSynchronize(st);
    last = std::chrono::steady_clock::now();
    fprintf(stderr, " %s after %.3f ms\n", e == hipSuccess ? "done" : hipGetErrorString(e), std::chrono::duration<double, std::milli>(last - t0).count());
    fflush(stderr);
}

// ---- host thread pool ----------------------------------------------------------------------------------
// The per-image preparation is a handful of sub-millisecond parallel sections (BVH subtrees, slot records, ray bins,
// operand packing); starting fifteen threads for each of them cost more than the work.  The workers are started once,
// detached, and never torn down (a static destructor would race with them at exit).  One job at a time: a caller that
// finds the pool busy (another host thread preparing another image) runs its section on its own thread.
namespace {
struct Pool {
    std::mutex job_mu;                       // held for the duration of one run()
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    const std::function<void(int)> *fn = nullptr;
    int n = 0, active = 0, n_workers = 0;
    uint64_t epoch = 0;
    std::atomic<int> next{0};

    explicit Pool(int workers) : n_workers(workers)
    {
        for (int t = 0; t < workers; ++t) std::thread([this] { worker(); }).detach();
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_start.wait(lk, [&] { return epoch != seen; });
            seen = epoch;
            const std::function<void(int)> *f = fn;
            const int nn = n;
            lk.unlock();
            for (int i = next.fetch_add(1); i < nn; i = next.fetch_add(1)) (*f)(i);
            lk.lock();
            if (--active == 0) cv_done.notify_one();
        }
    }
    void run(int nn, const std::function<void(int)> &f)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = &f; n = nn; next.store(0); active = n_workers; ++epoch;
        }
        cv_start.notify_all();
        for (int i = next.fetch_add(1); i < nn; i = next.fetch_add(1)) f(i);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return active == 0; });      // every worker has left the loop: `f` may go out of scope
    }
};

Pool *pool_instance()
{
    static Pool *p = [] {
        const int hw = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        return hw > 1 ? new Pool(hw - 1) : nullptr;          // never deleted, see above
    }();
    return p;
}
}  // namespace

// ICON_AMD_BUILD_THREADS=1: everything on the calling thread (the same results - the parallel sections write disjoint data)
void parallel_for(int n, const std::function<void(int)> &fn)
{
    int nt = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("ICON_AMD_BUILD_THREADS")) nt = std::max(atoi(e), 1);
    static thread_local bool in_section = false;             // a section that opens another one runs it inline
    Pool *p = (nt > 1 && n > 1 && !in_section) ? pool_instance() : nullptr;
    if (p && p->job_mu.try_lock()) {
        in_section = true;
        p->run(n, [&](int i) { in_section = true; fn(i); });
        in_section = false;
        p->job_mu.unlock();
        return;
    }
    for (int i = 0; i < n; ++i) fn(i);
}

namespace {

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b)
{
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}

// ---------------------------------------------------------------------------------------------------------
// The HOST builder: the checker of the device build (mesh_device.hip) and the path of ICON_AMD_MESH_BUILD=host /
// icon_debug_set_mesh_build(1).  Sequential, synchronous, and - by construction - the SAME arrays as the device
// build, bit for bit: every decision comes from mesh_rules.h (tests/test_gpu_mesh_build.py compares the arenas).
// ---------------------------------------------------------------------------------------------------------
struct HostBuild {
    int64_t V, F;
    int bound;
    std::vector<float> tbox, cen;
    std::vector<int32_t> order, tmp;
    std::vector<BvhNode> nodes;
    std::vector<uint8_t> leaf_cnt;
    MeshDyn dyn{};

    void link(int parent, int side, int ref, const float box[6])
    {
        if (parent < 0) { dyn.root = ref; return; }
        BvhNode &nd = nodes[parent];
        for (int a = 0; a < 3; ++a) { nd.lo[a][side] = box[a]; nd.hi[a][side] = box[3 + a]; }
        if (side) nd.child1 = ref; else nd.child0 = ref;
    }

    void bounds(int a, int b, float box[6], float cb[6]) const
    {
        for (int k = 0; k < 3; ++k) { box[k] = INFINITY; box[3 + k] = -INFINITY; cb[k] = INFINITY; cb[3 + k] = -INFINITY; }
        for (int i = a; i < b; ++i) {
            const int e = order[i];
            for (int k = 0; k < 3; ++k) {
                box[k] = fminf(box[k], tbox[6 * (size_t)e + k]); box[3 + k] = fmaxf(box[3 + k], tbox[6 * (size_t)e + 3 + k]);
                cb[k] = fminf(cb[k], cen[3 * (size_t)e + k]); cb[3 + k] = fmaxf(cb[3 + k], cen[3 * (size_t)e + k]);
            }
        }
    }

    void build(int begin, int end, int depth, int parent, int side, const float box[6], const float cb[6])
    {
        const int n = end - begin;
        if (n <= kLeafMax) {
            leaf_cnt[begin] = (uint8_t)n;
            link(parent, side, ~((begin << 2) | (n - 1)), box);
            dyn.n_leaves++; dyn.depth = std::max(dyn.depth, depth);
            return;
        }
        float lo[3], ext[3];
        for (int a = 0; a < 3; ++a) { lo[a] = cb[a]; ext[a] = cb[3 + a] - cb[a]; }
        bool valid = false;
        int best_axis = 0, best_bin = 0, nleft = 0;
        float cbox[2][6], ccb[2][6];
        if (!force_median(depth, n, bound)) {
            double best = std::numeric_limits<double>::infinity();
            for (int ax = 0; ax < 3; ++ax) {
                if (!(ext[ax] > 0.0f)) continue;
                int cnt[kSahBins] = {0};
                float bb[kSahBins][6], bc[kSahBins][6];
                for (int b = 0; b < kSahBins; ++b)
                    for (int k = 0; k < 3; ++k) { bb[b][k] = INFINITY; bb[b][3 + k] = -INFINITY; bc[b][k] = INFINITY; bc[b][3 + k] = -INFINITY; }
                for (int i = begin; i < end; ++i) {
                    const int e = order[i];
                    const int b = sah_bin(cen[3 * (size_t)e + ax], lo[ax], ext[ax]);
                    cnt[b]++;
                    for (int k = 0; k < 3; ++k) {
                        bb[b][k] = fminf(bb[b][k], tbox[6 * (size_t)e + k]); bb[b][3 + k] = fmaxf(bb[b][3 + k], tbox[6 * (size_t)e + 3 + k]);
                        bc[b][k] = fminf(bc[b][k], cen[3 * (size_t)e + k]); bc[b][3 + k] = fmaxf(bc[b][3 + k], cen[3 * (size_t)e + k]);
                    }
                }
                for (int b = 0; b < kSahBins - 1; ++b) {
                    float sb[2][6], sc[2][6];
                    int c2[2] = {0, 0};
                    for (int s = 0; s < 2; ++s)
                        for (int k = 0; k < 3; ++k) { sb[s][k] = INFINITY; sb[s][3 + k] = -INFINITY; sc[s][k] = INFINITY; sc[s][3 + k] = -INFINITY; }
                    for (int q = 0; q < kSahBins; ++q) {
                        if (!cnt[q]) continue;
                        const int s = q <= b ? 0 : 1;
                        c2[s] += cnt[q];
                        for (int k = 0; k < 3; ++k) {
                            sb[s][k] = fminf(sb[s][k], bb[q][k]); sb[s][3 + k] = fmaxf(sb[s][3 + k], bb[q][3 + k]);
                            sc[s][k] = fminf(sc[s][k], bc[q][k]); sc[s][3 + k] = fmaxf(sc[s][3 + k], bc[q][3 + k]);
                        }
                    }
                    if (!c2[0] || !c2[1]) continue;
                    const double cost = box_area(sb[0], sb[0] + 3) * c2[0] + box_area(sb[1], sb[1] + 3) * c2[1];
                    if (cost < best) {
                        best = cost; valid = true; best_axis = ax; best_bin = b; nleft = c2[0];
                        memcpy(cbox, sb, sizeof(sb)); memcpy(ccb, sc, sizeof(sc));
                    }
                }
            }
        }
        if (valid) {
            // stable partition: lefts keep their order, rights keep theirs
            int l = begin, r = 0;
            for (int i = begin; i < end; ++i) {
                const int e = order[i];
                if (sah_bin(cen[3 * (size_t)e + best_axis], lo[best_axis], ext[best_axis]) <= best_bin) order[l++] = e; else tmp[r++] = e;
            }
            for (int i = 0; i < r; ++i) order[l + i] = tmp[i];
        } else {
            nleft = n / 2;                                       // positional halving
            bounds(begin, begin + nleft, cbox[0], ccb[0]);
            bounds(begin + nleft, end, cbox[1], ccb[1]);
        }
        const int mid = begin + nleft, id = mid - 1;
        link(parent, side, id, box);
        dyn.n_nodes++;
        build(begin, mid, depth + 1, id, 0, cbox[0], ccb[0]);
        build(mid, end, depth + 1, id, 1, cbox[1], ccb[1]);
    }
};

// everything the build emits, as host arrays in arena order
struct HostArrays {
    MeshDyn dyn{};
    std::vector<float> vn;
    std::vector<BvhNode> nodes;
    std::vector<LeafRec> leafrec;
    std::vector<TriRec> tris;
    std::vector<TriAttr> attr;
    std::vector<int32_t> order, face2slot, bin_start, bin_slots;
};

// pure host code: no HIP call (icon_debug_host_mesh_build runs it without a device)
int host_build(const std::vector<float> &verts, const std::vector<int64_t> &faces, const std::vector<float> &cmap, const std::vector<float> &vis,
               int64_t V, int64_t F, int bound, HostArrays &out)
{
    for (int64_t i = 0; i < 3 * F; ++i)
        ICON_ARG(faces[i] >= 0 && faces[i] < V, "icon_mesh_create: face index out of range");
    for (int64_t i = 0; i < 3 * V; ++i)
        ICON_ARG(!bad_coord(verts[i]), "icon_mesh_create: non-finite (or absurdly large) vertex coordinate");
    HostBuild hb;
    hb.V = V; hb.F = F; hb.bound = bound;

    // S1: vertex normals = sum over incident faces (ascending face index) of (v1-v0)x(v2-v0),
    // then v / max(|v|, 1e-6)  [pytorch3d verts_normals_padded + F.normalize(eps=1e-6)]
    std::vector<float> &vn = out.vn;
    vn.assign(3 * V, 0.f);
    for (int64_t f = 0; f < F; ++f) {
        const int64_t id[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
        const V3 a{verts[3 * id[0]], verts[3 * id[0] + 1], verts[3 * id[0] + 2]};
        const V3 b{verts[3 * id[1]], verts[3 * id[1] + 1], verts[3 * id[1] + 2]};
        const V3 c{verts[3 * id[2]], verts[3 * id[2] + 1], verts[3 * id[2] + 2]};
        const V3 n = cross(sub(b, a), sub(c, a));
        for (int k = 0; k < 3; ++k) { vn[3 * id[k]] += n.x; vn[3 * id[k] + 1] += n.y; vn[3 * id[k] + 2] += n.z; }
    }
    for (int64_t v = 0; v < V; ++v) {
        const float x = vn[3 * v], y = vn[3 * v + 1], z = vn[3 * v + 2];
        float len = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
        if (len < 1e-6f) len = 1e-6f;
        vn[3 * v] = x / len; vn[3 * v + 1] = y / len; vn[3 * v + 2] = z / len;
    }

    // triangle boxes, centroids, bounds
    hb.tbox.resize(6 * F); hb.cen.resize(3 * F); hb.order.resize(F); hb.tmp.resize(F);
    hb.nodes.assign(F, BvhNode{}); hb.leaf_cnt.assign(F, 0);
    std::iota(hb.order.begin(), hb.order.end(), 0);
    float box[6], cb[6];
    for (int64_t f = 0; f < F; ++f)
        for (int a = 0; a < 3; ++a) {
            const float p0 = verts[3 * faces[3 * f] + a], p1 = verts[3 * faces[3 * f + 1] + a], p2 = verts[3 * faces[3 * f + 2] + a];
            const float lo = canon(fminf(fminf(p0, p1), p2)), hi = canon(fmaxf(fmaxf(p0, p1), p2));
            hb.tbox[6 * f + a] = lo; hb.tbox[6 * f + 3 + a] = hi; hb.cen[3 * f + a] = canon(0.5f * (lo + hi));
        }
    hb.bounds(0, (int)F, box, cb);
    hb.build(0, (int)F, 0, -1, 0, box, cb);
    MeshDyn &dyn = hb.dyn;
    for (int a = 0; a < 3; ++a) { dyn.box_lo[a] = box[a]; dyn.box_hi[a] = box[3 + a]; }
    const BinGrid g = bin_grid(dyn.box_lo, dyn.box_hi, F);
    dyn.gy = g.gy; dyn.gz = g.gz; dyn.bin_y0 = g.y0; dyn.bin_z0 = g.z0; dyn.bin_y1 = g.y1; dyn.bin_z1 = g.z1; dyn.bin_inv_y = g.inv_y; dyn.bin_inv_z = g.inv_z;

    // slot-ordered records
    const std::vector<int32_t> &order = hb.order;
    std::vector<TriRec> &tris = out.tris;
    std::vector<TriAttr> &attr = out.attr;
    std::vector<int32_t> &face2slot = out.face2slot;
    std::vector<LeafRec> &leafrec = out.leafrec;
    tris.resize(F); attr.resize(F); face2slot.resize(F); leafrec.resize(F);
    memset((void *)leafrec.data(), 0, sizeof(LeafRec) * F);
    for (int64_t p = 0; p < F; ++p) {
        const int64_t f = order[p];
        const int64_t id[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
        TriRec &tr = tris[p];
        for (int k = 0; k < 3; ++k) { tr.a[k] = verts[3 * id[0] + k]; tr.b[k] = verts[3 * id[1] + k]; tr.c[k] = verts[3 * id[2] + k]; }
        tr.ia = (int32_t)id[0]; tr.ib = (int32_t)id[1]; tr.ic = (int32_t)id[2];
        TriAttr &a = attr[p];
        for (int c = 0; c < 3; ++c) {
            for (int k = 0; k < 3; ++k) { a.n[c][k] = vn[3 * id[c] + k]; a.cm[c][k] = cmap[3 * id[c] + k]; }
            a.vis[c] = vis[id[c]];
        }
        a.face = (int32_t)f; a.pad[0] = a.pad[1] = 0;
        face2slot[f] = (int32_t)p;
    }
    for (int64_t L = 0; L < F; ++L) {
        const int cnt = hb.leaf_cnt[L];
        for (int t = 0; t < (cnt ? kLeafMax : 0); ++t) {
            const int64_t p = L + std::min(t, cnt - 1);
            TriPre pre;
            tri_setup(tris[p].a, tris[p].b, tris[p].c, order[p], pre);
            const float *src = reinterpret_cast<const float *>(&pre);
            for (int fld = 0; fld < 24; ++fld) leafrec[L].pair[t >> 1][fld][t & 1] = src[fld];
        }
    }

    // (y,z) ray bins: every triangle is listed in all cells its (y,z) bounding box, grown by eps, overlaps.
    // bin_cell_of() is monotone, so a query point inside the grown box lands in one of those cells; eps covers the
    // rounding of the float32 edge functions.  Filled in slot order: ascending lists.
    const int64_t n_cells = (int64_t)g.gy * g.gz;
    std::vector<int32_t> &bin_start = out.bin_start;
    std::vector<int32_t> &bin_slots = out.bin_slots;
    bin_start.assign(n_cells + 1, 0);
    auto range = [&](int64_t p, int &cy0, int &cy1, int &cz0, int &cz1) {
        const float *tb = &hb.tbox[6 * (size_t)order[p]];
        cy0 = bin_cell_of(tb[1] - kBinEps, g.y0, g.inv_y, g.gy); cy1 = bin_cell_of(tb[4] + kBinEps, g.y0, g.inv_y, g.gy);
        cz0 = bin_cell_of(tb[2] - kBinEps, g.z0, g.inv_z, g.gz); cz1 = bin_cell_of(tb[5] + kBinEps, g.z0, g.inv_z, g.gz);
    };
    for (int64_t p = 0; p < F; ++p) {
        int cy0, cy1, cz0, cz1; range(p, cy0, cy1, cz0, cz1);
        for (int cz = cz0; cz <= cz1; ++cz)
            for (int cy = cy0; cy <= cy1; ++cy) bin_start[(size_t)cz * g.gy + cy + 1]++;
    }
    int64_t max_bin = 0, total = 0;
    for (int64_t i = 1; i <= n_cells; ++i) { max_bin = std::max<int64_t>(max_bin, bin_start[i]); total += bin_start[i]; }
    bin_slots.clear();
    if (total <= bin_entries_cap(F)) {
        for (int64_t i = 1; i <= n_cells; ++i) bin_start[i] += bin_start[i - 1];
        bin_slots.resize(total);
        std::vector<int32_t> fill(bin_start.begin(), bin_start.end() - 1);
        for (int64_t p = 0; p < F; ++p) {
            int cy0, cy1, cz0, cz1; range(p, cy0, cy1, cz0, cz1);
            for (int cz = cz0; cz <= cz1; ++cz)
                for (int cy = cy0; cy <= cy1; ++cy) bin_slots[fill[(size_t)cz * g.gy + cy]++] = (int32_t)p;
        }
        dyn.bin_entries = (int32_t)total; dyn.max_bin = (int32_t)max_bin;
    } else {                                                     // as the device build: no lists, brute-force parity
        std::fill(bin_start.begin(), bin_start.end(), 0);
        dyn.gy = dyn.gz = 0; dyn.status |= kMeshBinOverflow;
    }
    dyn.status |= kMeshBuilt;
    out.dyn = dyn;
    out.nodes.swap(hb.nodes);
    out.order.swap(hb.order);
    return ICON_OK;
}

template <class Put>
int emit_arena(const HostArrays &h, const MeshLayout &Ly, Put put_fn)
{
    int rc;
    std::vector<MeshDyn> dynv(1, h.dyn);
    if ((rc = put_fn(Ly.dyn, dynv.data(), sizeof(MeshDyn))) || (rc = put_fn(Ly.vnormals, h.vn.data(), h.vn.size() * 4)) ||
        (rc = put_fn(Ly.nodes, h.nodes.data(), h.nodes.size() * sizeof(BvhNode))) ||
        (rc = put_fn(Ly.leaves, h.leafrec.data(), h.leafrec.size() * sizeof(LeafRec))) ||
        (rc = put_fn(Ly.tris, h.tris.data(), h.tris.size() * sizeof(TriRec))) || (rc = put_fn(Ly.attr, h.attr.data(), h.attr.size() * sizeof(TriAttr))) ||
        (rc = put_fn(Ly.slot2face, h.order.data(), h.order.size() * 4)) || (rc = put_fn(Ly.face2slot, h.face2slot.data(), h.face2slot.size() * 4)) ||
        (rc = put_fn(Ly.bin_start, h.bin_start.data(), h.bin_start.size() * 4)) || (rc = put_fn(Ly.bin_slots, h.bin_slots.data(), h.bin_slots.size() * 4)))
        return rc;
    return ICON_OK;
}

int mesh_build_host(icon_mesh *m, const float *d_verts, const int64_t *d_faces, const float *d_cmap, const float *d_vis, hipStream_t st)
{
    const int64_t V = m->V, F = m->F;
    std::vector<float> verts(3 * V), cmap(3 * V), vis(V);
    std::vector<int64_t> faces(3 * F);
    ICON_HIP(hipMemcpyAsync(verts.data(), d_verts, sizeof(float) * 3 * V, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipMemcpyAsync(faces.data(), d_faces, sizeof(int64_t) * 3 * F, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipMemcpyAsync(cmap.data(), d_cmap, sizeof(float) * 3 * V, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipMemcpyAsync(vis.data(), d_vis, sizeof(float) * V, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipStreamSynchronize(st));
    HostArrays h;
    int rc = host_build(verts, faces, cmap, vis, V, F, m->depth_bound, h);
    if (rc) return rc;
    char *ar = m->arena;
    rc = emit_arena(h, mesh_layout(V, F), [&](size_t off, const void *src, size_t bytes) -> int {
        if (bytes) ICON_HIP(hipMemcpyAsync(ar + off, src, bytes, hipMemcpyHostToDevice, st));
        return ICON_OK;
    });
    if (rc) return rc;
    ICON_HIP(hipStreamSynchronize(st));   // host vectors go out of scope
    *m->h_dyn = h.dyn;
    ICON_HIP(hipEventRecord(m->built, st));
    return ICON_OK;
}

int g_mesh_build_host = -1;      // -1: read ICON_AMD_MESH_BUILD once ("host" selects the host builder); 0 / 1: icon_debug_set_mesh_build

int mesh_create_common(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F, const float *d_cmap, const float *d_vis,
                       void *d_arena, size_t arena_bytes, bool owns, hipStream_t st, icon_mesh_t **out)
{
    icon_mesh *m = new icon_mesh();
    m->V = V; m->F = F; m->arena = (char *)d_arena; m->arena_bytes = arena_bytes; m->owns_arena = owns;
    m->depth_bound = depth_bound(F);
    int rc = mesh_host_state_get(&m->h_dyn, &m->built);
    if (rc) { icon_mesh_destroy(m); return rc; }
    mesh_bind_arena(m, mesh_layout(V, F));
    if (g_mesh_build_host < 0) { const char *e = getenv("ICON_AMD_MESH_BUILD"); g_mesh_build_host = (e && !strcmp(e, "host")) ? 1 : 0; }
    rc = g_mesh_build_host ? mesh_build_host(m, d_verts, d_faces, d_cmap, d_vis, st) : mesh_build_device(m, d_verts, d_faces, d_cmap, d_vis, st);
    if (rc) { icon_mesh_destroy(m); return rc; }
    *out = m;
    return ICON_OK;
}

int mesh_check_args(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F, const float *d_cmap, const float *d_vis, icon_mesh_t **out)
{
    ICON_ARG(out != nullptr, "icon_mesh_create: out is null");
    *out = nullptr;
    ICON_ARG(d_verts && d_faces && d_cmap && d_vis, "icon_mesh_create: null input pointer");
    ICON_ARG(V >= 3 && F >= 1, "icon_mesh_create: need V >= 3 and F >= 1");
    // slots are positions: 23 bits travel from the search to the feature phase (NearRef), 27 fit a leaf code
    ICON_ARG(F <= (1 << 23) && V < (1ll << 31), "icon_mesh_create: mesh too large (more than 2^23 faces)");
    return ICON_OK;
}

}  // namespace
}  // namespace icon

using namespace icon;

extern "C" const char *icon_last_error(void) { return icon::last_error_cstr(); }
extern "C" int icon_version(void) { return ICON_AMD_VERSION; }
extern "C" int icon_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int icon_mesh_arena_bytes(int64_t V, int64_t F, int64_t *bytes)
{
    ICON_ARG(bytes != nullptr && V >= 3 && F >= 1 && F <= (1 << 23), "icon_mesh_arena_bytes: bad argument");
    *bytes = (int64_t)mesh_layout(V, F).total;
    return ICON_OK;
}

extern "C" int icon_mesh_create_arena(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F,
                                      const float *d_cmap, const float *d_vis, void *d_arena, int64_t arena_bytes,
                                      void *stream, icon_mesh_t **out)
{
    int rc = mesh_check_args(d_verts, V, d_faces, F, d_cmap, d_vis, out);
    if (rc) return rc;
    ICON_ARG(d_arena != nullptr && ((uintptr_t)d_arena & 255) == 0, "icon_mesh_create_arena: the arena must be 256-byte aligned device memory");
    ICON_ARG(arena_bytes >= (int64_t)mesh_layout(V, F).total, "icon_mesh_create_arena: arena smaller than icon_mesh_arena_bytes(V, F)");
    return mesh_create_common(d_verts, V, d_faces, F, d_cmap, d_vis, d_arena, (size_t)arena_bytes, false, (hipStream_t)stream, out);
}

extern "C" int icon_mesh_create(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F,
                                const float *d_cmap, const float *d_vis, void *stream, icon_mesh_t **out)
{
    int rc = mesh_check_args(d_verts, V, d_faces, F, d_cmap, d_vis, out);
    if (rc) return rc;
    const size_t bytes = mesh_layout(V, F).total;
    void *arena = nullptr;
    ICON_HIP(hipMalloc(&arena, bytes));
    rc = mesh_create_common(d_verts, V, d_faces, F, d_cmap, d_vis, arena, bytes, true, (hipStream_t)stream, out);
    if (rc && !*out) (void)hipFree(arena);         // (mesh_create_common destroys the handle on failure; it never owned the arena then)
    return rc;
}

extern "C" int icon_mesh_destroy(icon_mesh_t *m)
{
    if (!m) return ICON_OK;
    if (m->owns_arena && m->arena) (void)hipFree(m->arena);     // (hipFree waits for the device: the caller-owned form never does)
    mesh_host_state_put(m->h_dyn, m->built);
    delete m;
    return ICON_OK;
}

extern "C" int icon_debug_set_mesh_build(int host)
{
    icon::g_mesh_build_host = host ? 1 : 0;
    return ICON_OK;
}

extern "C" int icon_mesh_vertex_normals(const icon_mesh_t *m, float *d_out, void *stream)
{
    ICON_ARG(m && d_out, "icon_mesh_vertex_normals: null argument");
    ICON_HIP(hipMemcpyAsync(d_out, m->d_vnormals, sizeof(float) * 3 * m->V, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ICON_OK;
}

// what the build found wrong with its input - known only once the build has RUN: wait != 0 blocks until then,
// wait == 0 answers from the pinned mirror as it is (bits == -1: still building)
extern "C" int icon_mesh_status(const icon_mesh_t *m, int wait, int *bits)
{
    ICON_ARG(m != nullptr, "icon_mesh_status: mesh is null");
    if (wait) ICON_HIP(hipEventSynchronize(m->built));
    const int st = *reinterpret_cast<const volatile int32_t *>(&m->h_dyn->status);
    if (!(st & kMeshBuilt)) { if (bits) *bits = -1; return ICON_OK; }
    if (bits) *bits = st & ~kMeshBuilt;
    if (st & kMeshBadFace) return fail(ICON_ERR_ARG, "icon_mesh_create: face index out of range");
    if (st & kMeshBadVertex) return fail(ICON_ERR_ARG, "icon_mesh_create: non-finite (or absurdly large) vertex coordinate");
    if (st & kMeshInternal) return fail(ICON_ERR_STATE, "icon_mesh_create: the device BVH build failed an internal check");
    return ICON_OK;
}

extern "C" int icon_mesh_stats(const icon_mesh_t *m, int64_t out[6])
{
    ICON_ARG(m && out, "icon_mesh_stats: null argument");
    ICON_HIP(hipEventSynchronize(m->built));
    const MeshDyn &d = *m->h_dyn;
    out[0] = d.n_nodes; out[1] = d.depth; out[2] = d.bin_entries; out[3] = d.max_bin; out[4] = d.n_leaves; out[5] = m->F;
    return ICON_OK;
}

// The host builder on HOST buffers (no device needed): fills h_arena (>= icon_mesh_arena_bytes, the caller zeroes it) in
// the arena layout.  The CPU test-suite checks the tree's invariants through it; the GPU suite compares the device
// build's arena with it byte for byte.
extern "C" int icon_debug_host_mesh_build(const float *h_verts, int64_t V, const int64_t *h_faces, int64_t F,
                                          const float *h_cmap, const float *h_vis, void *h_arena, int64_t arena_bytes)
{
    ICON_ARG(h_verts && h_faces && h_cmap && h_vis && h_arena, "icon_debug_host_mesh_build: null argument");
    ICON_ARG(V >= 3 && F >= 1 && F <= (1 << 23), "icon_debug_host_mesh_build: bad sizes");
    const MeshLayout Ly = mesh_layout(V, F);
    ICON_ARG(arena_bytes >= (int64_t)Ly.total, "icon_debug_host_mesh_build: arena too small");
    std::vector<float> verts(h_verts, h_verts + 3 * V), cmap(h_cmap, h_cmap + 3 * V), vis(h_vis, h_vis + V);
    std::vector<int64_t> faces(h_faces, h_faces + 3 * F);
    HostArrays h;
    int rc = host_build(verts, faces, cmap, vis, V, F, depth_bound(F), h);
    if (rc) return rc;
    char *ar = (char *)h_arena;
    return emit_arena(h, Ly, [&](size_t off, const void *src, size_t bytes) -> int { if (bytes) memcpy(ar + off, src, bytes); return ICON_OK; });
}

// byte offsets of the arena sections (tests compare the arenas of the host and the device build):
// out = [dyn, vnormals, nodes, leaves, tris, attr, slot2face, face2slot, bin_start, bin_slots, end of bin_slots, total]
extern "C" int icon_debug_mesh_layout(int64_t V, int64_t F, int64_t out[12])
{
    ICON_ARG(out && V >= 3 && F >= 1 && F <= (1 << 23), "icon_debug_mesh_layout: bad argument");
    const MeshLayout L = mesh_layout(V, F);
    const size_t v[12] = {L.dyn, L.vnormals, L.nodes, L.leaves, L.tris, L.attr, L.slot2face, L.face2slot, L.bin_start, L.bin_slots, L.tbox, L.total};
    for (int k = 0; k < 12; ++k) out[k] = (int64_t)v[k];
    return ICON_OK;
}
