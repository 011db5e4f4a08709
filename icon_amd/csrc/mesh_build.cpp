// mesh_build.cpp - per-image body-mesh preparation (host side of icon_mesh_create).
//
// Replaces the per-call prologue of cal_sdf_batch (reference lib/dataset/mesh_util.py:367-372):
// vertex normals (pytorch3d Meshes.verts_normals_padded) and the four face_vertices() gathers
// (lib/common/render_utils.py:149-163), and builds what the query kernels traverse: a BVH2 over
// the triangles (binned SAH) and a (y,z) bin grid for the +x ray-parity inside test.
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

namespace icon {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }
const char *last_error_cstr() { return g_err.c_str(); }

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

struct Box {
    float lo[3] = {INFINITY, INFINITY, INFINITY};
    float hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    double area() const
    {
        const double dx = std::max(0.0, (double)hi[0] - lo[0]), dy = std::max(0.0, (double)hi[1] - lo[1]),
                     dz = std::max(0.0, (double)hi[2] - lo[2]);
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

struct Builder {
    const float *verts;
    const int64_t *faces;
    const Box *tbox;               // [F]
    const float *cen;              // [F][3]
    int32_t *order;                // permutation of faces (shared; every (sub)build owns a disjoint range); leaves index into it
    std::vector<BvhNode> nodes;
    std::vector<std::pair<int, int>> leaves;   // (first index into `order`, count 1..kLeafMax)
    int max_depth = 0;
    int leaf_cap = kLeafMax;                   // triangles per leaf (tuning knob ICON_AMD_LEAF, 1..kLeafMax)

    // partitions order[begin, end) (binned SAH, median fallback) and returns the split position;
    // fills the range's box.  Returns -1 when the range becomes a leaf.
    int split(int begin, int end, int depth, Box &box)
    {
        box = Box();
        Box cb;
        for (int i = begin; i < end; ++i) { box.grow(tbox[order[i]]); cb.grow(&cen[3 * order[i]]); }
        const int n = end - begin;
        if (n <= leaf_cap) return -1;

        int axis = 0, mid = -1;
        const bool force_median = depth >= kStackDepth - 6;
        if (!force_median) {
            constexpr int NB = 16;
            double best = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.f)) continue;
                Box bb[NB]; int cnt[NB] = {0};
                for (int i = begin; i < end; ++i) {
                    int b = (int)((cen[3 * order[i] + ax] - lo) / ext * NB);
                    b = std::min(std::max(b, 0), NB - 1);
                    bb[b].grow(tbox[order[i]]); cnt[b]++;
                }
                double ra[NB]; int rc[NB]; Box acc; int c = 0;
                for (int b = NB - 1; b >= 1; --b) { acc.grow(bb[b]); c += cnt[b]; ra[b] = acc.area(); rc[b] = c; }
                acc = Box(); c = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    acc.grow(bb[b]); c += cnt[b];
                    if (c == 0 || rc[b + 1] == 0) continue;
                    const double cost = acc.area() * c + ra[b + 1] * rc[b + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                axis = best_axis;
                const float lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
                auto it = std::partition(order + begin, order + end, [&](int32_t f) {
                    int b = (int)((cen[3 * f + axis] - lo) / ext * 16);
                    b = std::min(std::max(b, 0), 15);
                    return b <= best_bin;
                });
                mid = (int)(it - order);
            }
        }
        if (mid <= begin || mid >= end) {   // median split on the widest centroid axis
            axis = 0;
            for (int ax = 1; ax < 3; ++ax)
                if (cb.hi[ax] - cb.lo[ax] > cb.hi[axis] - cb.lo[axis]) axis = ax;
            mid = begin + n / 2;
            std::nth_element(order + begin, order + mid, order + end,
                             [&](int32_t a, int32_t b) {
                                 const float ca = cen[3 * a + axis], cb2 = cen[3 * b + axis];
                                 return ca < cb2 || (ca == cb2 && a < b);
                             });
        }
        return mid;
    }

    // returns child reference (>=0 node, <0 leaf code); fills `box`.  Nodes and leaves are numbered in
    // depth-first pre-order.
    int32_t build(int begin, int end, int depth, Box &box)
    {
        max_depth = std::max(max_depth, depth);
        const int mid = split(begin, end, depth, box);
        if (mid < 0) { leaves.emplace_back(begin, end - begin); return ~(int32_t)(((leaves.size() - 1) << 2) | (size_t)(end - begin - 1)); }
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box b0, b1;
        const int32_t c0 = build(begin, mid, depth + 1, b0);
        const int32_t c1 = build(mid, end, depth + 1, b1);
        set_node(nodes[me], c0, c1, b0, b1);
        return me;
    }

    static void set_node(BvhNode &nd, int32_t c0, int32_t c1, const Box &b0, const Box &b1)
    {
        for (int k = 0; k < 3; ++k) { nd.lo[k][0] = b0.lo[k]; nd.hi[k][0] = b0.hi[k]; nd.lo[k][1] = b1.lo[k]; nd.hi[k][1] = b1.hi[k]; }
        nd.child0 = c0; nd.child1 = c1; nd.pad[0] = nd.pad[1] = 0;
    }

    // ---- the same tree, built by several threads ---------------------------------------------------
    // The top of the tree is split level by level down to kParDepth; every subtree below is an independent
    // job on a disjoint range of `order` (own node / leaf vectors).  A final pass emits top nodes and
    // job results in depth-first pre-order with rebased indices, so the arrays are IDENTICAL to what
    // build() produces (same node numbering, same leaf numbering, same `order`).
    static constexpr int kParDepth = 5;
    struct Job { int begin, end, depth; Builder *sub; int32_t ref; Box box; };
    struct Plan { int mid; int left, right; bool is_job; int job; Box box; };     // index into plans

    int32_t emit(int pi, const std::vector<Plan> &plans, std::vector<Job> &jobs, Box &box)
    {
        const Plan &p = plans[pi];
        if (p.is_job) {
            Job &j = jobs[p.job];
            const int32_t nbase = (int32_t)nodes.size(), lbase = (int32_t)leaves.size();
            auto rebase = [&](int32_t c) {
                if (c >= 0) return c + nbase;
                const int32_t code = ~c;
                return ~(int32_t)((((code >> 2) + lbase) << 2) | (code & 3));
            };
            for (const BvhNode &n : j.sub->nodes) { BvhNode m = n; m.child0 = rebase(n.child0); m.child1 = rebase(n.child1); nodes.push_back(m); }
            leaves.insert(leaves.end(), j.sub->leaves.begin(), j.sub->leaves.end());
            max_depth = std::max(max_depth, j.sub->max_depth);
            box = j.box;
            return rebase(j.ref);
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box b0, b1;
        const int32_t c0 = emit(p.left, plans, jobs, b0);
        const int32_t c1 = emit(p.right, plans, jobs, b1);
        set_node(nodes[me], c0, c1, b0, b1);
        box = p.box;
        return me;
    }

    // The top of the tree level by level: the nodes of one level own disjoint ranges of `order`, so their splits run side
    // by side (the sequential recursion spent 5 x F triangle visits here, more than the 32 subtree jobs together); the
    // plan tree that comes out is the one a depth-first recursion builds - emit() numbers nodes by ITS walk, not by creation order.
    int plan_levels(int F, std::vector<Plan> &plans, std::vector<Job> &jobs)
    {
        struct Item { int begin, end, plan; };
        plans.emplace_back();
        std::vector<Item> level{{0, F, 0}};
        for (int depth = 0; !level.empty(); ++depth) {
            std::vector<int> mids(level.size(), -1);
            std::vector<Box> boxes(level.size());
            std::vector<char> is_job(level.size(), 0);
            for (size_t i = 0; i < level.size(); ++i) is_job[i] = depth >= kParDepth || level[i].end - level[i].begin <= 64;
            parallel_for((int)level.size(), [&](int i) {
                if (!is_job[i]) mids[i] = split(level[i].begin, level[i].end, depth, boxes[i]);
            });
            std::vector<Item> next;
            for (size_t i = 0; i < level.size(); ++i) {
                const Item it = level[i];
                if (is_job[i] || mids[i] < 0) {          // mids < 0 cannot happen for n > 64 >= leaf_cap, kept for safety: a leaf-sized job
                    plans[it.plan].is_job = true; plans[it.plan].job = -1;
                    plans[it.plan].mid = it.begin; plans[it.plan].left = it.end; plans[it.plan].right = depth;     // parked until numbered below
                    continue;
                }
                max_depth = std::max(max_depth, depth);
                const int l = (int)plans.size(), r = l + 1;
                plans.emplace_back(); plans.emplace_back();
                plans[it.plan].is_job = false; plans[it.plan].mid = mids[i]; plans[it.plan].box = boxes[i];
                plans[it.plan].left = l; plans[it.plan].right = r;
                next.push_back({it.begin, mids[i], l});
                next.push_back({mids[i], it.end, r});
            }
            level.swap(next);
        }
        // jobs numbered depth-first, left before right (the order is not visible in the output)
        std::vector<int> stack{0};
        while (!stack.empty()) {
            const int pi = stack.back(); stack.pop_back();
            Plan &p = plans[pi];
            if (p.is_job) {
                p.job = (int)jobs.size();
                jobs.push_back(Job{p.mid, p.left, p.right, nullptr, 0, Box()});
            } else {
                stack.push_back(p.right); stack.push_back(p.left);
            }
        }
        return 0;
    }

    int32_t build_parallel(int F, Box &box, int n_threads)
    {
        std::vector<Plan> plans;
        std::vector<Job> jobs;
        const int root = plan_levels(F, plans, jobs);
        std::vector<Builder> subs(jobs.size(), *this);
        for (size_t i = 0; i < jobs.size(); ++i) { subs[i].nodes.clear(); subs[i].leaves.clear(); subs[i].max_depth = 0; jobs[i].sub = &subs[i]; }
        (void)n_threads;
        parallel_for((int)jobs.size(), [&](int i) {
            Job &j = jobs[i];
            j.ref = j.sub->build(j.begin, j.end, j.depth, j.box);
        });
        return emit(root, plans, jobs, box);
    }
};

// S2 per-triangle constants (same float32 operation sequence as the checker's orc_tri_setup)
inline float dot3(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
inline void tri_setup(const float *a, const float *b, const float *c, int32_t face, TriPre &t)
{
    for (int k = 0; k < 3; ++k) { t.a[k] = a[k]; t.b[k] = b[k]; t.ab[k] = b[k] - a[k]; t.ac[k] = c[k] - a[k]; t.bc[k] = c[k] - b[k]; }
    t.a00 = dot3(t.ab, t.ab); t.a01 = dot3(t.ab, t.ac); t.a11 = dot3(t.ac, t.ac);
    const float b11 = dot3(t.bc, t.bc);
    t.i00 = (t.a00 > 0.0f) ? 1.0f / t.a00 : 0.0f;
    t.i11 = (t.a11 > 0.0f) ? 1.0f / t.a11 : 0.0f;
    t.ibc = (b11 > 0.0f) ? 1.0f / b11 : 0.0f;
    const float nn = fmaf(t.a00, t.a11, -(t.a01 * t.a01));
    // zero area (or a sliver whose Gram determinant rounds to <= 0): NaN makes both barycentrics NaN, every
    // comparison of the inside test false, and the distance the minimum over the three edge segments - exact
    t.inn = (nn > 0.0f) ? 1.0f / nn : std::numeric_limits<float>::quiet_NaN();
    t.face = face; t.pad = 0;
}

inline int cell_of(float v, float v0, float inv, int g)
{
    int c = (int)floorf((v - v0) * inv);
    return std::min(std::max(c, 0), g - 1);
}

template <class T>
int upload(T **dst, const std::vector<T> &src, hipStream_t st)
{
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    ICON_HIP(hipMalloc((void **)dst, bytes));
    if (!src.empty()) ICON_HIP(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, st));
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

extern "C" int icon_mesh_create(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F,
                                const float *d_cmap, const float *d_vis, void *stream, icon_mesh_t **out)
{
    ICON_ARG(out != nullptr, "icon_mesh_create: out is null");
    *out = nullptr;
    ICON_ARG(d_verts && d_faces && d_cmap && d_vis, "icon_mesh_create: null input pointer");
    ICON_ARG(V >= 3 && F >= 1, "icon_mesh_create: need V >= 3 and F >= 1");
    ICON_ARG(F < (1 << 27) && V < (1ll << 31), "icon_mesh_create: mesh too large");
    hipStream_t st = (hipStream_t)stream;
    const bool verbose = getenv("ICON_AMD_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();

    std::vector<float> verts(3 * V), cmap(3 * V), vis(V);
    std::vector<int64_t> faces(3 * F);
    ICON_HIP(hipMemcpyAsync(verts.data(), d_verts, sizeof(float) * 3 * V, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipMemcpyAsync(faces.data(), d_faces, sizeof(int64_t) * 3 * F, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipMemcpyAsync(cmap.data(), d_cmap, sizeof(float) * 3 * V, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipMemcpyAsync(vis.data(), d_vis, sizeof(float) * V, hipMemcpyDeviceToHost, st));
    ICON_HIP(hipStreamSynchronize(st));
    for (int64_t i = 0; i < 3 * F; ++i)
        ICON_ARG(faces[i] >= 0 && faces[i] < V, "icon_mesh_create: face index out of range");
    for (int64_t i = 0; i < 3 * V; ++i)     // a NaN breaks the strict weak ordering of the builder's partitions
        ICON_ARG(std::isfinite(verts[i]) && std::fabs(verts[i]) <= 1e6f, "icon_mesh_create: non-finite (or absurdly large) vertex coordinate");

    const auto t1 = now();
    // S1: vertex normals = sum over incident faces (ascending face index) of (v1-v0)x(v2-v0),
    // then v / max(|v|, 1e-6)  [pytorch3d verts_normals_padded + F.normalize(eps=1e-6)]
    std::vector<float> vn(3 * V, 0.f);
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

    const auto t2 = now();
    // BVH
    Builder bd;
    bd.verts = verts.data(); bd.faces = faces.data();
    std::vector<Box> tbox(F);
    std::vector<float> cen(3 * F);
    std::vector<int32_t> order(F);
    std::iota(order.begin(), order.end(), 0);
    Box mesh_box;
    for (int64_t f = 0; f < F; ++f) {
        for (int k = 0; k < 3; ++k) tbox[f].grow(&verts[3 * faces[3 * f + k]]);
        for (int k = 0; k < 3; ++k) cen[3 * f + k] = 0.5f * (tbox[f].lo[k] + tbox[f].hi[k]);
        mesh_box.grow(tbox[f]);
    }
    bd.tbox = tbox.data(); bd.cen = cen.data(); bd.order = order.data();
    bd.nodes.reserve(F);
    if (const char *e = getenv("ICON_AMD_LEAF")) bd.leaf_cap = std::min(std::max(atoi(e), 1), kLeafMax);
    Box root_box;
    // ICON_AMD_BUILD_THREADS (default: up to 16 hardware threads; 1 = the sequential builder, same tree)
    int n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("ICON_AMD_BUILD_THREADS")) n_threads = std::max(atoi(e), 1);
    int32_t root = (n_threads > 1 && F >= 2048) ? bd.build_parallel((int)F, root_box, n_threads) : bd.build(0, (int)F, 0, root_box);
    if (getenv("ICON_AMD_BUILD_CHECK")) {      // self-test: the threaded builder must reproduce the sequential arrays exactly
        Builder sq = bd;
        std::vector<int32_t> order2(F);
        std::iota(order2.begin(), order2.end(), 0);
        sq.order = order2.data(); sq.nodes.clear(); sq.leaves.clear(); sq.max_depth = 0;
        Box rb;
        const int32_t r2 = sq.build(0, (int)F, 0, rb);
        const bool same = r2 == root && sq.nodes.size() == bd.nodes.size() && sq.leaves == bd.leaves && order2 == order &&
                          sq.max_depth == bd.max_depth &&
                          (sq.nodes.empty() || memcmp(sq.nodes.data(), bd.nodes.data(), sq.nodes.size() * sizeof(BvhNode)) == 0) &&
                          memcmp(&rb, &root_box, sizeof(Box)) == 0;
        if (!same) return fail(ICON_ERR_STATE, "icon_mesh_create: threaded BVH build differs from the sequential build");
    }
    int32_t root_is_leaf = 0;
    if (root < 0) {   // tiny mesh: wrap the single leaf in a node with an empty second child
        BvhNode nd{};
        for (int k = 0; k < 3; ++k) { nd.lo[k][0] = root_box.lo[k]; nd.hi[k][0] = root_box.hi[k]; nd.lo[k][1] = INFINITY; nd.hi[k][1] = -INFINITY; }
        nd.child0 = root; nd.child1 = root;  // second child is never visited: its box distance is +inf
        bd.nodes.push_back(nd);
        root_is_leaf = 1;
    } else if (root != 0) {
        return fail(ICON_ERR_STATE, "icon_mesh_create: BVH root is not node 0");
    }
    if (bd.max_depth + 2 > kStackDepth) return fail(ICON_ERR_UNSUPPORTED, "icon_mesh_create: BVH too deep");

    const auto t3 = now();
    // slot-ordered triangle records / attributes.  Every leaf owns exactly kLeafMax consecutive
    // slots; short leaves are padded with copies of their last triangle (same face id, so a copy can
    // never change the arg-min) whose vertex ids are -1 so the ray-parity scans skip them.
    const int64_t n_leaves = (int64_t)bd.leaves.size();
    const int64_t S = n_leaves * kLeafMax;
    std::vector<TriRec> tris(S);
    std::vector<TriAttr> attr(S);
    std::vector<int32_t> slot2face(S);
    std::vector<int32_t> face2slot(std::max<int64_t>(F, 1), 0);
    std::vector<LeafRec> leafrec(n_leaves);
    std::vector<int64_t> slot_src(S, -1);          // real slots: index into bd.order; padding: -1
    const int n_rec_chunks = (int)std::min<int64_t>(64, std::max<int64_t>(n_leaves, 1));
    parallel_for(n_rec_chunks, [&](int chunk) {
    for (int64_t L = n_leaves * chunk / n_rec_chunks; L < n_leaves * (chunk + 1) / n_rec_chunks; ++L) {
        const int begin = bd.leaves[L].first, cnt = bd.leaves[L].second;
        for (int t = 0; t < kLeafMax; ++t) {
            const int64_t s = L * kLeafMax + t;
            const bool real = t < cnt;
            const int64_t f = order[begin + std::min(t, cnt - 1)];
            const int64_t id[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
            TriRec &tr = tris[s];
            for (int k = 0; k < 3; ++k) { tr.a[k] = verts[3 * id[0] + k]; tr.b[k] = verts[3 * id[1] + k]; tr.c[k] = verts[3 * id[2] + k]; }
            tr.ia = real ? (int32_t)id[0] : -1; tr.ib = real ? (int32_t)id[1] : -1; tr.ic = real ? (int32_t)id[2] : -1;
            TriAttr &a = attr[s];
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 3; ++k) { a.n[c][k] = vn[3 * id[c] + k]; a.cm[c][k] = cmap[3 * id[c] + k]; }
            for (int c = 0; c < 3; ++c) a.vis[c] = vis[id[c]];
            a.face = (int32_t)f; a.pad[0] = a.pad[1] = 0;
            slot2face[s] = (int32_t)f;
            if (real) { slot_src[s] = begin + t; face2slot[f] = (int32_t)s; }
            TriPre pre;
            tri_setup(tr.a, tr.b, tr.c, (int32_t)f, pre);
            const float *src = reinterpret_cast<const float *>(&pre);
            for (int fld = 0; fld < 24; ++fld) leafrec[L].pair[t >> 1][fld][t & 1] = src[fld];
        }
    }
    });

    const auto t4 = now();
    // (y,z) ray bins: every triangle is listed in all cells its (y,z) bounding box, grown by
    // eps, overlaps.  cell_of() is monotone, so a query point inside the grown box lands in one
    // of those cells; eps covers the rounding of the float32 edge functions.
    const float eps = 1e-5f;
    const float y0 = mesh_box.lo[1] - 4 * eps, y1 = mesh_box.hi[1] + 4 * eps;
    const float z0 = mesh_box.lo[2] - 4 * eps, z1 = mesh_box.hi[2] + 4 * eps;
    // square cells, about two per triangle
    const double cell = std::sqrt(std::max((double)(y1 - y0) * (double)(z1 - z0), 1e-12) / (2.0 * (double)F));
    const int gy = std::min(std::max((int)std::ceil((y1 - y0) / cell), 1), 2048);
    const int gz = std::min(std::max((int)std::ceil((z1 - z0) / cell), 1), 2048);
    const float inv_y = (float)gy / (y1 - y0), inv_z = (float)gz / (z1 - z0);
    std::vector<int32_t> bin_start((size_t)gy * gz + 1, 0);
    auto range = [&](int64_t s, int &cy0, int &cy1, int &cz0, int &cz1) {
        const Box &b = tbox[order[slot_src[s]]];
        cy0 = cell_of(b.lo[1] - eps, y0, inv_y, gy); cy1 = cell_of(b.hi[1] + eps, y0, inv_y, gy);
        cz0 = cell_of(b.lo[2] - eps, z0, inv_z, gz); cz1 = cell_of(b.hi[2] + eps, z0, inv_z, gz);
    };
    // three parallel passes over the slots (count, fill, sort) around a sequential prefix sum; the atomics only decide
    // the order INSIDE a bin, which the last pass makes ascending again - the arrays are those of a sequential fill
    const int n_bin_chunks = (int)std::min<int64_t>(64, std::max<int64_t>(S / 256, 1));
    parallel_for(n_bin_chunks, [&](int chunk) {
        for (int64_t s = S * chunk / n_bin_chunks; s < S * (chunk + 1) / n_bin_chunks; ++s) {
            if (slot_src[s] < 0) continue;
            int cy0, cy1, cz0, cz1; range(s, cy0, cy1, cz0, cz1);
            for (int cz = cz0; cz <= cz1; ++cz)
                for (int cy = cy0; cy <= cy1; ++cy) __atomic_fetch_add(&bin_start[(size_t)cz * gy + cy + 1], 1, __ATOMIC_RELAXED);
        }
    });
    int64_t max_bin = 0;
    for (size_t i = 1; i < bin_start.size(); ++i) { max_bin = std::max<int64_t>(max_bin, bin_start[i]); bin_start[i] += bin_start[i - 1]; }
    std::vector<int32_t> bin_slots(bin_start.back());
    {
        std::vector<int32_t> fill(bin_start.begin(), bin_start.end() - 1);
        parallel_for(n_bin_chunks, [&](int chunk) {
            for (int64_t s = S * chunk / n_bin_chunks; s < S * (chunk + 1) / n_bin_chunks; ++s) {
                if (slot_src[s] < 0) continue;
                int cy0, cy1, cz0, cz1; range(s, cy0, cy1, cz0, cz1);
                for (int cz = cz0; cz <= cz1; ++cz)
                    for (int cy = cy0; cy <= cy1; ++cy)
                        bin_slots[__atomic_fetch_add(&fill[(size_t)cz * gy + cy], 1, __ATOMIC_RELAXED)] = (int32_t)s;
            }
        });
        const int64_t n_cells = (int64_t)gy * gz;
        const int n_sort_chunks = (int)std::min<int64_t>(64, std::max<int64_t>(n_cells / 256, 1));
        parallel_for(n_sort_chunks, [&](int chunk) {
            for (int64_t c = n_cells * chunk / n_sort_chunks; c < n_cells * (chunk + 1) / n_sort_chunks; ++c)
                std::sort(bin_slots.begin() + bin_start[c], bin_slots.begin() + bin_start[c + 1]);      // ascending slot order inside every bin
        });
    }

    if (getenv("ICON_AMD_BUILD_CHECK")) {      // self-test: the parallel passes must reproduce the sequential fill exactly
        std::vector<int32_t> fill2(bin_start.begin(), bin_start.end() - 1), slots2(bin_slots.size());
        for (int64_t s = 0; s < S; ++s) {
            if (slot_src[s] < 0) continue;
            int cy0, cy1, cz0, cz1; range(s, cy0, cy1, cz0, cz1);
            for (int cz = cz0; cz <= cz1; ++cz)
                for (int cy = cy0; cy <= cy1; ++cy) slots2[fill2[(size_t)cz * gy + cy]++] = (int32_t)s;
        }
        if (slots2 != bin_slots) return fail(ICON_ERR_STATE, "icon_mesh_create: parallel ray-bin fill differs from the sequential fill");
    }

    const auto t5 = now();
    icon_mesh *m = new icon_mesh();
    m->V = V; m->F = F;
    int rc;
    if ((rc = upload(&m->d_vnormals, vn, st)) || (rc = upload(&m->d_nodes, bd.nodes, st)) ||
        (rc = upload(&m->d_tris, tris, st)) || (rc = upload(&m->d_attr, attr, st)) ||
        (rc = upload(&m->d_slot2face, slot2face, st)) || (rc = upload(&m->d_face2slot, face2slot, st)) || (rc = upload(&m->d_leaves, leafrec, st)) ||
        (rc = upload(&m->d_bin_start, bin_start, st)) ||
        (rc = upload(&m->d_bin_slots, bin_slots, st))) {
        icon_mesh_destroy(m);
        return rc;
    }
    ICON_HIP(hipStreamSynchronize(st));   // host vectors go out of scope
    MeshDev &d = m->dev;
    d.nodes = m->d_nodes; d.tris = m->d_tris; d.attr = m->d_attr; d.slot2face = m->d_slot2face; d.face2slot = m->d_face2slot;
    d.leaves = m->d_leaves;
    d.n_tris = (int32_t)S; d.root_is_leaf = root_is_leaf;
    d.bin_start = m->d_bin_start; d.bin_slots = m->d_bin_slots;
    d.bin_y0 = y0; d.bin_z0 = z0; d.bin_y1 = y1; d.bin_z1 = z1; d.bin_inv_y = inv_y; d.bin_inv_z = inv_z;
    d.gy = gy; d.gz = gz;
    for (int k = 0; k < 3; ++k) { d.box_lo[k] = mesh_box.lo[k]; d.box_hi[k] = mesh_box.hi[k]; }
    m->stats[0] = (int64_t)bd.nodes.size(); m->stats[1] = bd.max_depth;
    m->stats[2] = (int64_t)bin_slots.size(); m->stats[3] = max_bin;
    m->stats[4] = n_leaves; m->stats[5] = S;
    if (verbose)
        fprintf(stderr, "[icon_amd] mesh_create: d2h %.2f ms, normals %.2f, bvh %.2f, records %.2f, bins %.2f, upload %.2f\n",
                ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, now()));
    *out = m;
    return ICON_OK;
}

extern "C" int icon_mesh_destroy(icon_mesh_t *m)
{
    if (!m) return ICON_OK;
    (void)hipFree(m->d_vnormals); (void)hipFree(m->d_nodes); (void)hipFree(m->d_tris); (void)hipFree(m->d_attr);
    (void)hipFree(m->d_slot2face); (void)hipFree(m->d_face2slot); (void)hipFree(m->d_bin_start); (void)hipFree(m->d_bin_slots); (void)hipFree(m->d_leaves);
    delete m;
    return ICON_OK;
}

extern "C" int icon_mesh_vertex_normals(const icon_mesh_t *m, float *d_out, void *stream)
{
    ICON_ARG(m && d_out, "icon_mesh_vertex_normals: null argument");
    ICON_HIP(hipMemcpyAsync(d_out, m->d_vnormals, sizeof(float) * 3 * m->V, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ICON_OK;
}

extern "C" int icon_mesh_stats(const icon_mesh_t *m, int64_t out[6])
{
    ICON_ARG(m && out, "icon_mesh_stats: null argument");
    for (int k = 0; k < 6; ++k) out[k] = m->stats[k];
    return ICON_OK;
}
