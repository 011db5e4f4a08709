// clean_mesh.hip - lib/dataset/mesh_util.py:778-791 (clean_mesh: trimesh's split(only_watertight=False), the component with the
// most vertices kept), called at apps/ICON.py:755-756 on the marching-cubes output - as ONE native call on the stream.
//
// trimesh's connectivity (graph.split -> face_adjacency): two faces are adjacent when they share an EDGE that exactly two
// faces use; faces that merely touch in a vertex (marching-cubes pinch points) are not connected.  Until round 4 the edge
// grouping, the per-component vertex counts and the compaction were torch operators around the union-find kernel (argsort of
// 3 F edge keys, unique, a bincount over F bins, nonzero, boolean indexing: 2.0 ms for the 117,000-face body surface, three
// times the whole coarse-to-fine evaluation that produces the volume).  Here:
//   edges      an open-addressing hash table keyed by (min, max) vertex: per slot the number of uses and the first two faces
//   components lock-free union-find over the FACES (one union per slot with exactly two uses); label = smallest face index
//   sizes      distinct (label, vertex) incidences through the same table (a pinch vertex counts for both sides, as in
//              trimesh's submeshes); one atomic per wave and label
//   winner     max of (vertices << 32 | ~label): most vertices, ties to the component holding the lowest-index face
//   output     order-preserving compaction of the winner's faces and vertices (flags, block counts, scan, scatter), vertex
//              indices renumbered; (vertices, faces) counts through pinned memory - the only synchronisation
#include "common.h"

namespace icon {

namespace {

constexpr unsigned long long kEmptyKey = ~0ull;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}

// slot of `key` in the table (inserted if absent): linear probing, the table is at most half full
__device__ __forceinline__ unsigned hash_slot(unsigned long long *keys, unsigned mask, unsigned long long key, bool *fresh)
{
    unsigned s = (unsigned)mix64(key) & mask;
    while (true) {
        // (an ordinary load may see the slot as it WAS - the eight XCDs have an L2 each - but a slot only ever goes from empty to
        //  ONE key: a stale "empty" is settled by the CAS, anything else read here is final)
        const unsigned long long k = keys[s];
        if (k == key) { *fresh = false; return s; }
        if (k == kEmptyKey) {
            const unsigned long long old = atomicCAS(&keys[s], kEmptyKey, key);
            if (old == kEmptyKey) { *fresh = true; return s; }
            if (old == key) { *fresh = false; return s; }
        }
        s = (s + 1) & mask;
    }
}

// FRESH: agent-scope loads (the eight XCDs have an L2 each).  The first walk uses ordinary loads - a stale parent is still an
// ancestor-or-self of what it was, the walk ends at a node that WAS a root, and the CAS below is the judge; after a lost CAS the
// walk is repeated on fresh values.
template <bool FRESH>
__device__ __forceinline__ int uf_find(int *parent, int x)
{
    while (true) {
        const int p = FRESH ? __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : parent[x];
        if (p == x) return x;
        const int gp = FRESH ? __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : parent[p];
        if (gp != p) parent[x] = gp;       // path halving: benign race, always an ancestor
        x = p;
    }
}
__device__ __forceinline__ void uf_unite(int *parent, int u, int v)
{
    u = uf_find<false>(parent, u); v = uf_find<false>(parent, v);
    while (u != v) {
        const int hi = max(u, v), lo = min(u, v);
        if (atomicCAS(&parent[hi], hi, lo) == hi) return;       // hooks the larger root under the smaller: a root is the smallest index of its tree
        u = uf_find<true>(parent, u); v = uf_find<true>(parent, v);
    }
}

struct CleanCtx {
    const float *verts; const int64_t *faces; int64_t V, F;
    unsigned long long *keys; unsigned mask;      // hash table (edges, then (label, vertex) incidences)
    int *cnt, *f0, *f1;                          // per slot: uses of the edge, its first two faces
    int *slot_of;                                // [3 F] slot of every edge instance
    int *parent, *label;                         // [F]
    int *comp_verts;                             // [F] vertices of the component whose label is the index
    unsigned long long *best;                    // (vertices << 32) | ~label of the winner
    uint8_t *keep_f, *keep_v;                    // [F], [V]
    int *blk_f, *blk_v;                          // counts / offsets per 256 entries
    int *remap;                                  // [V] new index of a kept vertex
    int *totals;                                 // [0] vertices kept, [1] faces kept, [2] bad-input flag
};

__global__ __launch_bounds__(256) void k_cm_init(CleanCtx c)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < c.F) { c.parent[i] = (int)i; c.comp_verts[i] = 0; }
    if (i < c.V) { c.keep_v[i] = 0; c.remap[i] = 0x7fffffff; }    // remap doubles as "smallest label among the vertex's faces" until the emit passes
    if (i == 0) { *c.best = 0ull; c.totals[0] = c.totals[1] = c.totals[2] = 0; }
}

__global__ __launch_bounds__(256) void k_cm_edges(CleanCtx c)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * c.F) return;
    const int64_t f = t / 3; const int k = (int)(t - 3 * f);
    const int64_t a = c.faces[3 * f + k], b = c.faces[3 * f + (k + 1) % 3];
    if (a < 0 || b < 0 || a >= c.V || b >= c.V) { c.totals[2] = 1; return; }
    const unsigned long long key = (unsigned long long)min(a, b) * (unsigned long long)c.V + (unsigned long long)max(a, b);
    bool fresh;
    const unsigned s = hash_slot(c.keys, c.mask, key, &fresh);
    const int n = atomicAdd(&c.cnt[s], 1);
    if (n == 0) c.f0[s] = (int)f; else if (n == 1) c.f1[s] = (int)f;
    c.slot_of[t] = (int)s;
}

// one thread per edge instance, in FACE order (marching-cubes faces come out cell by cell: neighbours in the list are neighbours
// in space, the trees stay shallow - the same unions in hash-slot order took 450-920 us, these 150): the face unites with the
// other user of an edge that exactly two faces use, once per pair
__global__ __launch_bounds__(256) void k_cm_unite(CleanCtx c)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * c.F || c.totals[2]) return;
    const int s = c.slot_of[t];
    if (c.cnt[s] != 2) return;
    const int f = (int)(t / 3), a = c.f0[s], b = c.f1[s];
    const int g = a == f ? b : a;
    if (g < f) uf_unite(c.parent, f, g);
}

// read-only walk to the root, result out of place (a flatten that compresses in place can have its final store overtaken by
// another thread's path-halving store)
__global__ __launch_bounds__(256) void k_cm_flatten(CleanCtx c)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= c.F) return;
    int x = (int)i;
    while (true) { const int p = c.parent[x]; if (p == x) break; x = p; }
    c.label[i] = x;
}

// vertices per component = distinct (label, vertex) incidences.  Round 4 pushed all 3 F incidences through the hash table (287 us
// for the 551,000 faces of a 513^3 surface); but a vertex of a marching-cubes surface lies in ONE component unless it is a pinch
// point, so (round 5): the smallest label among a vertex's faces (k_cm_vmin: one atomicMin per incidence on a [V] array) counts
// the vertex once for that component, and only the incidences whose face carries ANOTHER label - the pinch vertices, a handful - go
// through the table to be counted once per (label, vertex).  The same numbers, whatever the mesh (triangle soups included).
__global__ __launch_bounds__(256) void k_cm_vmin(CleanCtx c)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * c.F || c.totals[2]) return;
    atomicMin(&c.remap[c.faces[t]], c.label[t / 3]);
}

// one atomic per wave and label (a body surface is one component: 58,000 increments of ONE counter otherwise)
__device__ __forceinline__ void cm_count_labels(const CleanCtx &c, bool todo, int L)
{
    while (__any(todo)) {
        const unsigned long long m = __ballot(todo);
        const int leader = __ffsll((long long)m) - 1;
        const int Lr = __shfl(L, leader);
        const unsigned long long same = __ballot(todo && L == Lr);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&c.comp_verts[Lr], __popcll(same));
        todo = todo && L != Lr;
    }
}

__global__ __launch_bounds__(256) void k_cm_sizes(CleanCtx c)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = c.totals[2] == 0;
    // (a) every referenced vertex, for the smallest label among its faces
    int L = (ok && t < c.V) ? c.remap[t] : 0x7fffffff;
    cm_count_labels(c, L != 0x7fffffff, L);
    // (b) the incidences under any OTHER label: once per (label, vertex)
    bool fresh = false;
    L = -1;
    if (ok && t < 3 * c.F) {
        L = c.label[t / 3];
        const int64_t v = c.faces[t];
        if (L != c.remap[v])
            (void)hash_slot(c.keys, c.mask, (unsigned long long)L * (unsigned long long)c.V + (unsigned long long)v, &fresh);
    }
    cm_count_labels(c, fresh, L);
}

__global__ __launch_bounds__(256) void k_cm_best(CleanCtx c)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= c.F || c.label[f] != (int)f) return;                // component roots only
    atomicMax(c.best, ((unsigned long long)(unsigned)c.comp_verts[f] << 32) | (unsigned long long)(0xffffffffu - (unsigned)f));
}

__global__ __launch_bounds__(256) void k_cm_flags(CleanCtx c)
{
    __shared__ int ws[4];
    if (c.totals[2]) return;                                     // a face names a missing vertex (uniform): nothing is written
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int win = (int)(0xffffffffu - (unsigned)(*c.best & 0xffffffffull));
    const bool keep = f < c.F && c.label[f] == win;
    if (f < c.F) c.keep_f[f] = keep ? 1 : 0;
    if (keep) { c.keep_v[c.faces[3 * f]] = 1; c.keep_v[c.faces[3 * f + 1]] = 1; c.keep_v[c.faces[3 * f + 2]] = 1; }
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) c.blk_f[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_cm_count_v(CleanCtx c)
{
    __shared__ int ws[4];
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(v < c.V && c.keep_v[v]);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) c.blk_v[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of the block counts, in place (one workgroup; eight counts per thread and round); block 0: vertices, 1: faces
__global__ __launch_bounds__(1024) void k_cm_scan(CleanCtx c)
{
    __shared__ int wtot[16];
    int *cnt = blockIdx.x == 0 ? c.blk_v : c.blk_f;
    const int nb = (int)(((blockIdx.x == 0 ? c.V : c.F) + 255) / 256);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < nb; base += 8192) {
        const int i = base + (int)threadIdx.x * 8;
        int v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = i + k < nb ? cnt[i + k] : 0; sum += v[k]; }
        int incl = sum;
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
        __syncthreads();
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int t = wtot[q]; before += q < w ? t : 0; all += t; }
        int run = carry + before + incl - sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (i + k < nb) cnt[i + k] = run; run += v[k]; }
        carry += all;
    }
    if (threadIdx.x == 0) c.totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void k_cm_emit_v(CleanCtx c, float *__restrict__ out_verts)
{
    __shared__ int ws[4];
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = v < c.V && c.keep_v[v];
    const unsigned long long b = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) ws[w] = __popcll(b);
    __syncthreads();
    if (!keep) return;
    int k = c.blk_v[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) k += ws[q];
    c.remap[v] = k;
    out_verts[3 * (int64_t)k] = c.verts[3 * v]; out_verts[3 * (int64_t)k + 1] = c.verts[3 * v + 1]; out_verts[3 * (int64_t)k + 2] = c.verts[3 * v + 2];
}

__global__ __launch_bounds__(256) void k_cm_emit_f(CleanCtx c, int32_t *__restrict__ out_faces)
{
    __shared__ int ws[4];
    if (c.totals[2]) return;
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = f < c.F && c.keep_f[f];
    const unsigned long long b = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) ws[w] = __popcll(b);
    __syncthreads();
    if (!keep) return;
    int k = c.blk_f[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) k += ws[q];
    out_faces[3 * (int64_t)k] = c.remap[c.faces[3 * f]]; out_faces[3 * (int64_t)k + 1] = c.remap[c.faces[3 * f + 1]];
    out_faces[3 * (int64_t)k + 2] = c.remap[c.faces[3 * f + 2]];
}

}  // namespace

// scratch of icon_clean_mesh, owned by the workspace: grown on demand, never shrunk
struct CleanState {
    char *buf = nullptr; size_t bytes = 0;
    int *h_totals = nullptr;                     // pinned
};

void clean_destroy(CleanState *s)
{
    if (!s) return;
    (void)hipFree(s->buf); (void)hipHostFree(s->h_totals);
    delete s;
}

}  // namespace icon

using namespace icon;

// d_verts [V,3] f32, d_faces [F,3] i64 (device) -> the largest component: d_out_verts [>= V,3] f32, d_out_faces [>= F,3] i32
// (device, caller-allocated at the input sizes; the first h_counts[0] vertices / h_counts[1] faces are valid).  Synchronises
// the stream once, for the two counts.  A face naming a vertex that does not exist: ICON_ERR_ARG, nothing written.
extern "C" int icon_clean_mesh(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F, float *d_out_verts, int32_t *d_out_faces,
                               int64_t *h_counts, icon_work_t *work, void *stream)
{
    ICON_ARG(work && h_counts && d_verts && d_faces && d_out_verts && d_out_faces, "icon_clean_mesh: null argument");
    ICON_ARG(V > 0 && F > 0 && V < (1ll << 31) && F < (1ll << 28), "icon_clean_mesh: 0 < V < 2^31, 0 < F < 2^28");      // (the table: <= 2^31 slots)
    hipStream_t st = (hipStream_t)stream;
    if (!work->clean) work->clean = new CleanState();
    CleanState *s = work->clean;
    if (!s->h_totals) ICON_HIP(hipHostMalloc((void **)&s->h_totals, 4 * sizeof(int), hipHostMallocDefault));
    unsigned slots = 1024;
    while ((int64_t)slots < 6 * F) slots <<= 1;                 // 3 F keys at most: the table stays at most half full
    const int64_t nbf = (F + 255) / 256, nbv = (V + 255) / 256;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) / 256 * 256; return at; };
    const size_t o_keys = take((size_t)slots * 8), o_cnt = take((size_t)slots * 4), o_f0 = take((size_t)slots * 4), o_f1 = take((size_t)slots * 4);
    const size_t o_slot = take((size_t)F * 12);
    const size_t o_parent = take((size_t)F * 4), o_label = take((size_t)F * 4), o_cv = take((size_t)F * 4), o_best = take(8);
    const size_t o_kf = take((size_t)F), o_kv = take((size_t)V), o_bf = take((size_t)nbf * 4), o_bv = take((size_t)nbv * 4);
    const size_t o_remap = take((size_t)V * 4), o_tot = take(16);
    if (o > s->bytes) {
        (void)hipFree(s->buf); s->buf = nullptr; s->bytes = 0;
        ICON_HIP(hipMalloc((void **)&s->buf, o));
        s->bytes = o;
    }
    CleanCtx c{};
    c.verts = d_verts; c.faces = d_faces; c.V = V; c.F = F;
    c.keys = reinterpret_cast<unsigned long long *>(s->buf + o_keys); c.mask = slots - 1;
    c.cnt = reinterpret_cast<int *>(s->buf + o_cnt); c.f0 = reinterpret_cast<int *>(s->buf + o_f0); c.f1 = reinterpret_cast<int *>(s->buf + o_f1);
    c.slot_of = reinterpret_cast<int *>(s->buf + o_slot);
    c.parent = reinterpret_cast<int *>(s->buf + o_parent); c.label = reinterpret_cast<int *>(s->buf + o_label);
    c.comp_verts = reinterpret_cast<int *>(s->buf + o_cv); c.best = reinterpret_cast<unsigned long long *>(s->buf + o_best);
    c.keep_f = reinterpret_cast<uint8_t *>(s->buf + o_kf); c.keep_v = reinterpret_cast<uint8_t *>(s->buf + o_kv);
    c.blk_f = reinterpret_cast<int *>(s->buf + o_bf); c.blk_v = reinterpret_cast<int *>(s->buf + o_bv);
    c.remap = reinterpret_cast<int *>(s->buf + o_remap); c.totals = reinterpret_cast<int *>(s->buf + o_tot);

    ICON_HIP(hipMemsetAsync(c.keys, 0xff, (size_t)slots * 8, st));
    ICON_HIP(hipMemsetAsync(c.cnt, 0, (size_t)slots * 4, st));
    const unsigned gF = (unsigned)nbf, gV = (unsigned)nbv, g3F = (unsigned)((3 * F + 255) / 256);
    hipLaunchKernelGGL(k_cm_init, dim3(std::max(gF, gV)), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_edges, dim3(g3F), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_unite, dim3(g3F), dim3(256), 0, st, c);
    ICON_HIP(hipMemsetAsync(c.keys, 0xff, (size_t)slots * 8, st));             // the table serves the incidence count next
    hipLaunchKernelGGL(k_cm_flatten, dim3(gF), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_vmin, dim3(g3F), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_sizes, dim3(std::max(g3F, gV)), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_best, dim3(gF), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_flags, dim3(gF), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_count_v, dim3(gV), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cm_scan, dim3(2), dim3(1024), 0, st, c);
    hipLaunchKernelGGL(k_cm_emit_v, dim3(gV), dim3(256), 0, st, c, d_out_verts);
    hipLaunchKernelGGL(k_cm_emit_f, dim3(gF), dim3(256), 0, st, c, d_out_faces);
    ICON_HIP(hipGetLastError());
    ICON_HIP(hipMemcpyAsync(s->h_totals, c.totals, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
    ICON_HIP(hipStreamSynchronize(st));
    if (s->h_totals[2]) return fail(ICON_ERR_ARG, "icon_clean_mesh: face index out of range");
    h_counts[0] = s->h_totals[0]; h_counts[1] = s->h_totals[1];
    return ICON_OK;
}
