// mesh_device.hip - per-image body-mesh preparation ON THE DEVICE (round 4): vertex normals, a binned-SAH BVH2,
// the slot-ordered triangle records and the (y,z) ray bins are built by kernels enqueued on the caller's stream
// from the device tensors of smpl_feat_dict; icon_mesh_create_arena never copies the mesh to the host, never
// allocates and never synchronises.
//
// Reference being replaced: the per-call prologue of cal_sdf_batch (lib/dataset/mesh_util.py:367-372:
// Meshes(verts, faces).verts_normals_padded() and four face_vertices() gathers, on the device, on every query()
// call) plus what the kaolin leaves build internally.  Here: once per image.
//
// The tree is the one the host builder (mesh_build.cpp) emits - same splits, same arrays, bit for bit
// (mesh_rules.h holds every shared decision; tests/test_gpu_mesh_build.py compares the two):
//   * top-down binned SAH (16 bins x 3 axes on the centroids of the triangle boxes, float64 cost), STABLE
//     partitions, positional halving where SAH has no candidate or the depth bound says so;
//   * IDs are positions (common.h): node id = split position - 1, leaf id = first slot, slot = position in
//     `order` - nothing has to be counted or compacted before it can be addressed.
// Kernels (one stream, 19 operations):
//   k_face_prep       per face: triangle box + centroid, mesh bounds (ordered-uint atomic maxima), vertex adjacency
//   k_vertex_normals  per vertex: incident faces in ascending order (S1 is order-sensitive), normalise; + root task
//   k_bvh_bin / k_bvh_part  x kTopLevels: a node too large for one workgroup - every 256-triangle chunk bins into
//                     the node's histogram (LDS, then global atomics) and records its own per-bin counts; the
//                     partition kernel re-derives the split in every chunk from the finished histogram, places its
//                     chunk from the counts of the chunks before it (stable, no scan pass) and chunk 0 writes the
//                     children's task records
//   k_bvh_sub         one workgroup per subtree of <= 1024 triangles, data in LDS: wavefronts take nodes from a
//                     shared pool, one wave splits one node (histogram by LDS atomics, 45 SAH candidates on 45 lanes,
//                     ballot-ranked stable partition), pushes one child and continues with the other
//   k_tri_records     per slot: TriRec / TriAttr / LeafRec (S2 constants), inverse permutation, ray-bin cell counts
//   k_scan_cells, k_bin_fill, k_bin_sort   CSR ray bins (lists ascending, as a sequential fill would leave them)
#pragma clang fp contract(off)

#include "common.h"
#include "mesh_rules.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace icon {

namespace {

struct BuildCtx {
    const float *verts; const int64_t *faces; const float *cmap; const float *vis;
    int32_t V, F, bound;
    MeshDyn *dyn; BuildHdr *hdr;
    int32_t *valence; uint8_t *leaf_cnt; BTask *tasks; uint32_t *hist; int32_t *cell_count, *cell_cursor;
    float *vnormals; BvhNode *nodes; LeafRec *leaves; TriRec *tris; TriAttr *attr; int32_t *slot2face, *face2slot;
    int32_t *bin_start, *bin_slots;
    float *tbox, *cen; int32_t *order[2]; int32_t *adj; int32_t *chunkcnt; int32_t *subq; struct STask *sublist;
    int32_t nck; int64_t cells_cap, entries_cap;
};

// ---- ordered-uint encoding of floats: max-atomics with a zero identity ---------------------------------------
__device__ __forceinline__ uint32_t enc(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float dec(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
// MINIMA are kept as maxima of the COMPLEMENTED code (~enc is order-reversing; 0 is still the identity).  No float negation
// anywhere: with minima stored as max(enc(-x)) and decoded as -dec(u) - or with the sign bit flipped by an integer xor
// next to the bitcast - hipcc (ROCm 7.2) lost the sign flip of ONE element of the unrolled decode loops (the select had
// gone to the scalar ALU and the fneg was dropped: lo.x came out as +|lo.x|; found by the byte comparison with the host build).
__device__ __forceinline__ uint32_t enc_min(float f) { return ~enc(f); }
__device__ __forceinline__ float dec_min(uint32_t u) { return dec(~u); }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float sane(float v, int *bad) { if (bad_coord(v)) { *bad = 1; return 0.0f; } return v; }

__device__ __forceinline__ void load_face(const BuildCtx &c, int f, int id[3], float p[3][3], int *bad_face, int *bad_vert)
{
    for (int k = 0; k < 3; ++k) {
        int64_t v = c.faces[3 * (int64_t)f + k];
        if (v < 0 || v >= c.V) { *bad_face = 1; v = 0; }
        id[k] = (int)v;
        for (int a = 0; a < 3; ++a) p[k][a] = sane(c.verts[3 * v + a], bad_vert);
    }
}

// ---------------------------------------------------------------------------------------------
// 1. faces
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_face_prep(BuildCtx c)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool live = f < c.F;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, ce[3] = {0.f, 0.f, 0.f};
    int bf = 0, bv = 0;
    if (live) {
        int id[3]; float p[3][3];
        load_face(c, f, id, p, &bf, &bv);
        for (int a = 0; a < 3; ++a) {
            lo[a] = canon(fminf(fminf(p[0][a], p[1][a]), p[2][a]));
            hi[a] = canon(fmaxf(fmaxf(p[0][a], p[1][a]), p[2][a]));
            ce[a] = canon(0.5f * (lo[a] + hi[a]));
            c.tbox[6 * f + a] = lo[a]; c.tbox[6 * f + 3 + a] = hi[a]; c.cen[3 * f + a] = ce[a];
        }
        c.order[0][f] = f;
        if (!bf)                                           // a face naming a missing vertex contributes to no normal
            for (int k = 0; k < 3; ++k) {
                const int pos = atomicAdd(&c.valence[id[k]], 1);
                if (pos < kAdjCap) c.adj[(size_t)id[k] * kAdjCap + pos] = 3 * f + k;
            }
        if (bf | bv) atomicOr(&c.dyn->status, (bf ? kMeshBadFace : 0) | (bv ? kMeshBadVertex : 0));
    }
    // mesh bounds: wave reduction, one atomic per value and wave
    uint32_t u[12];
    for (int a = 0; a < 3; ++a) {
        u[a] = live ? enc_min(lo[a]) : 0u; u[3 + a] = live ? enc(hi[a]) : 0u;
        u[6 + a] = live ? enc_min(ce[a]) : 0u; u[9 + a] = live ? enc(ce[a]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        uint32_t v = u[k];
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = v > o ? v : o; }
        if ((threadIdx.x & 63) == 0 && v) atomicMax(&c.hdr->mesh_ubox[k], v);
    }
}

// ---------------------------------------------------------------------------------------------
// 2. vertex normals (S1): sum over incident faces, ascending face id (then corner), of (v1-v0)x(v2-v0);
//    v / max(|v|, 1e-6)  [pytorch3d verts_normals_padded + F.normalize(eps=1e-6)]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void face_normal(const BuildCtx &c, int f, float n[3])
{
    int id[3]; float p[3][3]; int bf = 0, bv = 0;
    load_face(c, f, id, p, &bf, &bv);
    const float ux = p[1][0] - p[0][0], uy = p[1][1] - p[0][1], uz = p[1][2] - p[0][2];
    const float vx = p[2][0] - p[0][0], vy = p[2][1] - p[0][1], vz = p[2][2] - p[0][2];
    n[0] = fmaf(uy, vz, -(uz * vy)); n[1] = fmaf(uz, vx, -(ux * vz)); n[2] = fmaf(ux, vy, -(uy * vx));
}

__global__ __launch_bounds__(256) void k_vertex_normals(BuildCtx c)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v == 0) {
        // the root task and what the query kernels need of the bounding box
        float box[6], cb[6];
        for (int a = 0; a < 3; ++a) {
            box[a] = dec_min(c.hdr->mesh_ubox[a]); box[3 + a] = dec(c.hdr->mesh_ubox[3 + a]);
            cb[a] = dec_min(c.hdr->mesh_ubox[6 + a]); cb[3 + a] = dec(c.hdr->mesh_ubox[9 + a]);
        }
        BTask &r = c.tasks[0];
        r.begin = 0; r.end = c.F; r.depth = 0; r.parent = -1; r.side = 0; r.buf = 0; r.from_atomics = 0;
        for (int k = 0; k < 6; ++k) { r.box[k] = box[k]; r.cb[k] = cb[k]; }
        if (c.F > kSubMax && kTopLevels > 0) r.kind = 1;
        else { r.kind = 2; c.subq[atomicAdd(&c.hdr->n_sub, 1)] = 0; }
        MeshDyn &d = *c.dyn;
        for (int a = 0; a < 3; ++a) { d.box_lo[a] = box[a]; d.box_hi[a] = box[3 + a]; }
        const BinGrid g = bin_grid(d.box_lo, d.box_hi, c.F);
        d.gy = g.gy; d.gz = g.gz; d.bin_y0 = g.y0; d.bin_z0 = g.z0; d.bin_y1 = g.y1; d.bin_z1 = g.z1; d.bin_inv_y = g.inv_y; d.bin_inv_z = g.inv_z;
    }
    if (v >= c.V) return;
    int bv = 0;
    for (int a = 0; a < 3; ++a) (void)sane(c.verts[3 * (int64_t)v + a], &bv);
    if (bv) atomicOr(&c.dyn->status, kMeshBadVertex);
    float x = 0.f, y = 0.f, z = 0.f;
    const int cnt = c.valence[v];
    if (cnt <= kAdjCap) {
        int e[kAdjCap];
        for (int k = 0; k < kAdjCap; ++k) e[k] = k < cnt ? c.adj[(size_t)v * kAdjCap + k] : 0x7fffffff;
        for (int i = 1; i < kAdjCap; ++i) {                  // insertion sort (the fill order is whatever the atomics made it)
            const int key = e[i];
            int j = i - 1;
            while (j >= 0 && e[j] > key) { e[j + 1] = e[j]; --j; }
            e[j + 1] = key;
        }
        for (int k = 0; k < cnt; ++k) {
            float n[3];
            face_normal(c, e[k] / 3, n);
            x += n[0]; y += n[1]; z += n[2];
        }
    } else {                                                  // a vertex of more than kAdjCap corners: scan
        for (int f = 0; f < c.F; ++f) {
            int hit = 0; bool ok = true;
            for (int k = 0; k < 3; ++k) {
                const int64_t q = c.faces[3 * (int64_t)f + k];
                if (q < 0 || q >= c.V) ok = false;
                if (q == v) ++hit;
            }
            if (!ok || !hit) continue;
            float n[3];
            face_normal(c, f, n);
            for (int h = 0; h < hit; ++h) { x += n[0]; y += n[1]; z += n[2]; }
        }
    }
    float len = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
    if (len < 1e-6f) len = 1e-6f;
    c.vnormals[3 * (int64_t)v] = x / len; c.vnormals[3 * (int64_t)v + 1] = y / len; c.vnormals[3 * (int64_t)v + 2] = z / len;
}

// ---------------------------------------------------------------------------------------------
// 3. BVH
// ---------------------------------------------------------------------------------------------
struct Decision { int valid, axis, bin, nleft; float cbox[2][6], ccb[2][6]; };

__device__ __forceinline__ void task_bounds(const BTask &t, float box[6], float cb[6])
{
    if (t.from_atomics) {
        for (int a = 0; a < 3; ++a) {
            box[a] = dec_min(t.ubox[a]); box[3 + a] = dec(t.ubox[3 + a]);
            cb[a] = dec_min(t.ubox[6 + a]); cb[3 + a] = dec(t.ubox[9 + a]);
        }
    } else {
        for (int k = 0; k < 6; ++k) { box[k] = t.box[k]; cb[k] = t.cb[k]; }
    }
}

// one triangle into a node's histogram (LDS): per axis with a positive centroid extent, 13 atomics
template <class GetC, class GetB>
__device__ __forceinline__ void hist_add(uint32_t *h, const float lo[3], const float ext[3], GetC cen, GetB tb)
{
    const float cx = cen(0), cy = cen(1), cz = cen(2);
    uint32_t u[12];
    u[0] = enc_min(tb(0)); u[1] = enc_min(tb(1)); u[2] = enc_min(tb(2)); u[3] = enc(tb(3)); u[4] = enc(tb(4)); u[5] = enc(tb(5));
    u[6] = enc_min(cx); u[7] = enc_min(cy); u[8] = enc_min(cz); u[9] = enc(cx); u[10] = enc(cy); u[11] = enc(cz);
    const float cc[3] = {cx, cy, cz};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (!(ext[ax] > 0.0f)) continue;
        uint32_t *q = h + (ax * kSahBins + sah_bin(cc[ax], lo[ax], ext[ax])) * 13;
        atomicAdd(q, 1u);
#pragma unroll
        for (int k = 0; k < 12; ++k) atomicMax(q + 1 + k, u[k]);
    }
}

// The split of a node from its finished histogram, by ONE wavefront: lane (axis * 15 + b) prices the candidate
// "bins 0..b | bins b+1..15" of its axis exactly as the host's sweep does (unions of min / max are order-free, the
// cost is the same float64 expression); the winner is the first minimum in (axis, bin) order.  The winning lane
// writes the decision and the children's bounds to *D.
__device__ __forceinline__ void sah_choose(const uint32_t *h, const float ext[3], int lane, Decision *D)
{
    const int ax = lane / 15, b = lane - ax * 15;
    double cost = INFINITY;
    float bl[2][6], cl[2][6];
    int cnt[2] = {0, 0};
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < 3; ++k) { bl[s][k] = INFINITY; bl[s][3 + k] = -INFINITY; cl[s][k] = INFINITY; cl[s][3 + k] = -INFINITY; }
    if (lane < 45 && ext[ax] > 0.0f) {
        for (int k = 0; k < kSahBins; ++k) {
            const uint32_t *q = h + (ax * kSahBins + k) * 13;
            const int n = (int)q[0];
            if (!n) continue;
            const int s = k <= b ? 0 : 1;
            cnt[s] += n;
            for (int a = 0; a < 3; ++a) {
                bl[s][a] = fminf(bl[s][a], dec_min(q[1 + a])); bl[s][3 + a] = fmaxf(bl[s][3 + a], dec(q[4 + a]));
                cl[s][a] = fminf(cl[s][a], dec_min(q[7 + a])); cl[s][3 + a] = fmaxf(cl[s][3 + a], dec(q[10 + a]));
            }
        }
        if (cnt[0] && cnt[1]) cost = box_area(bl[0], bl[0] + 3) * cnt[0] + box_area(bl[1], bl[1] + 3) * cnt[1];
    }
    double best = cost;
    for (int d = 1; d < 64; d <<= 1) { const double o = __shfl_xor(best, d); best = o < best ? o : best; }
    const unsigned long long win = __ballot(cost == best && cost < (double)INFINITY);
    if (!win) { if (lane == 0) D->valid = 0; return; }
    const int w = __ffsll((long long)win) - 1;
    if (lane == w) {
        D->valid = 1; D->axis = ax; D->bin = b; D->nleft = cnt[0];
        for (int s = 0; s < 2; ++s)
            for (int k = 0; k < 6; ++k) { D->cbox[s][k] = bl[s][k]; D->ccb[s][k] = cl[s][k]; }
    }
}

// a processed node (or leaf) reports itself to its parent: its reference and its own box go into the parent's record
__device__ __forceinline__ void link_node(const BuildCtx &c, int parent, int side, int ref, const float box[6])
{
    if (parent < 0) { c.dyn->root = ref; return; }
    BvhNode &nd = c.nodes[parent];
    for (int a = 0; a < 3; ++a) { nd.lo[a][side] = box[a]; nd.hi[a][side] = box[3 + a]; }
    if (side) nd.child1 = ref; else nd.child0 = ref;
}

// which (task, chunk) of level L workgroup w works on: task s owns workgroups [begin / kChunk + s, ... + chunks)
// (disjoint for the tasks of one level: they are in position order and every task adds one to the base)
__device__ __forceinline__ bool find_chunk(const BTask *T, int nslots, int w, int &slot, int &k)
{
    for (int s = 0; s < nslots; ++s) {
        if (T[s].kind != 1) continue;
        const int base = T[s].begin / kChunk + s, nc = (T[s].end - T[s].begin + kChunk - 1) / kChunk;
        if (w >= base && w < base + nc) { slot = s; k = w - base; return true; }
    }
    return false;
}

__global__ __launch_bounds__(kChunk) void k_bvh_bin(BuildCtx c, int L)
{
    __shared__ uint32_t h[kHistWords];
    const BTask *T = c.tasks + ((1 << L) - 1);
    int s = 0, k = 0;
    if (!find_chunk(T, 1 << L, (int)blockIdx.x, s, k)) return;
    const BTask &P = T[s];
    const int n = P.end - P.begin;
    if (force_median(P.depth, n, c.bound)) return;            // positional split: no histogram
    float box[6], cb[6], lo[3], ext[3];
    task_bounds(P, box, cb);
    for (int a = 0; a < 3; ++a) { lo[a] = cb[a]; ext[a] = cb[3 + a] - cb[a]; }
    for (int i = threadIdx.x; i < kHistWords; i += kChunk) h[i] = 0;
    __syncthreads();
    const int i = k * kChunk + (int)threadIdx.x;
    if (i < n) {
        const int e = c.order[L & 1][P.begin + i];
        const float *ce = c.cen + 3 * (size_t)e, *tb = c.tbox + 6 * (size_t)e;
        hist_add(h, lo, ext, [&](int a) { return ce[a]; }, [&](int q) { return tb[q]; });
    }
    __syncthreads();
    uint32_t *g = c.hist + (size_t)((1 << L) - 1 + s) * kHistWords;
    for (int q = threadIdx.x; q < kHistWords; q += kChunk) {
        const uint32_t v = h[q];
        if (!v) continue;
        if (q % 13 == 0) atomicAdd(g + q, v); else atomicMax(g + q, v);
    }
    int32_t *cc = c.chunkcnt + ((size_t)L * c.nck + blockIdx.x) * (3 * kSahBins);
    if (threadIdx.x < 3 * kSahBins) cc[threadIdx.x] = (int32_t)h[threadIdx.x * 13];
}

__global__ __launch_bounds__(kChunk) void k_bvh_part(BuildCtx c, int L)
{
    __shared__ Decision D;
    __shared__ int s_wcnt[kChunk / 64];
    __shared__ int s_loff;
    BTask *T = c.tasks + ((1 << L) - 1);
    int s = 0, k = 0;
    if (!find_chunk(T, 1 << L, (int)blockIdx.x, s, k)) return;
    const BTask P = T[s];
    const int n = P.end - P.begin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float box[6], cb[6], lo[3], ext[3];
    task_bounds(P, box, cb);
    for (int a = 0; a < 3; ++a) { lo[a] = cb[a]; ext[a] = cb[3 + a] - cb[a]; }
    const bool forced = force_median(P.depth, n, c.bound);
    if (wave == 0) {
        if (forced) { if (lane == 0) D.valid = 0; }
        else sah_choose(c.hist + (size_t)((1 << L) - 1 + s) * kHistWords, ext, lane, &D);
    }
    __syncthreads();
    const bool valid = D.valid != 0;
    const int nleft = valid ? D.nleft : n / 2;
    const int axis = valid ? D.axis : 0, bin = valid ? D.bin : 0;
    // lefts in the chunks before this one: from their recorded bin counts
    if (wave == 0) {
        int sum = 0;
        if (valid) {
            const int32_t *cc = c.chunkcnt + ((size_t)L * c.nck + (blockIdx.x - k)) * (3 * kSahBins) + axis * kSahBins;
            for (int q = lane; q < k; q += 64)
                for (int b = 0; b <= bin; ++b) sum += cc[(size_t)q * (3 * kSahBins) + b];
            for (int d = 1; d < 64; d <<= 1) sum += __shfl_xor(sum, d);
        } else {
            sum = min(k * kChunk, nleft);
        }
        if (lane == 0) s_loff = sum;
    }
    const int i = k * kChunk + (int)threadIdx.x;
    const bool act = i < n;
    const int e = act ? c.order[L & 1][P.begin + i] : 0;
    const bool isl = act && (valid ? sah_bin(c.cen[3 * (size_t)e + axis], lo[axis], ext[axis]) <= bin : i < nleft);
    const unsigned long long bl = __ballot(isl);
    if (lane == 0) s_wcnt[wave] = __popcll(bl);
    __syncthreads();
    int lrank = __popcll(bl & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) lrank += s_wcnt[w];
    const int loff = s_loff;
    if (act) {
        const int dest = isl ? P.begin + loff + lrank : P.begin + nleft + (k * kChunk - loff) + ((int)threadIdx.x - lrank);
        c.order[(L + 1) & 1][dest] = e;
    }
    BTask *C = c.tasks + ((1 << (L + 1)) - 1) + 2 * s;
    if (!valid && act) {                                      // positional split: the children's bounds are not in a histogram
        uint32_t *u = C[isl ? 0 : 1].ubox;
        const float *ce = c.cen + 3 * (size_t)e, *tb = c.tbox + 6 * (size_t)e;
        for (int a = 0; a < 3; ++a) {
            atomicMax(u + a, enc_min(tb[a])); atomicMax(u + 3 + a, enc(tb[3 + a]));
            atomicMax(u + 6 + a, enc_min(ce[a])); atomicMax(u + 9 + a, enc(ce[a]));
        }
    }
    if (k == 0 && threadIdx.x == 0) {
        const int mid = P.begin + nleft, id = mid - 1;
        link_node(c, P.parent, P.side, id, box);
        atomicAdd(&c.dyn->n_nodes, 1);
        for (int sd = 0; sd < 2; ++sd) {
            BTask &t = C[sd];
            t.begin = sd ? mid : P.begin; t.end = sd ? P.end : mid; t.depth = P.depth + 1; t.parent = id; t.side = sd;
            t.buf = (L + 1) & 1; t.from_atomics = valid ? 0 : 1;
            if (valid) for (int q = 0; q < 6; ++q) { t.box[q] = D.cbox[sd][q]; t.cb[q] = D.ccb[sd][q]; }
            const int nc = t.end - t.begin;
            if (nc > kSubMax && L + 1 < kTopLevels) t.kind = 1;
            else { t.kind = 2; c.subq[atomicAdd(&c.hdr->n_sub, 1)] = (1 << (L + 1)) - 1 + 2 * s + sd; }
        }
    }
}

// ---- subtrees: one workgroup per subtree, level by level, one wavefront per node ------------------------------------
// The nodes of one level of the subtree sit in a list; wave w takes nodes w, w + 8, ...: histogram by LDS atomics,
// 45 SAH candidates on 45 lanes, ballot-ranked stable partition; children that are not leaves are appended to the next
// level's list (one LDS atomic per child), a barrier, the lists swap.  (A first version handed nodes from wave to wave
// through a lock-protected pool with sleeping pollers: correct, and 16-43 SECONDS per mesh - seven idle waves per
// workgroup kept the lock busy around the clock.  No locks, no polling now.)
constexpr int kSubWaves = 8;
constexpr int kListCap = 256;          // nodes of one level: every pending node holds >= 5 of the subtree's <= kSubMax triangles
struct STask { int32_t begin, end, depth, parent, side, buf; float box[6], cb[6]; int32_t pad[2]; };
static_assert(sizeof(STask) == 80, "STask layout");

struct SubLds {
    int gid[kSubMax];
    float cen[kSubMax * 3];
    float tbox[kSubMax * 6];
    uint16_t ord[2][kSubMax];
    uint32_t hist[kSubWaves][kHistWords];
    Decision dec[kSubWaves];
    STask list[2][kListCap];
    int count[2];
};

// element access of a subtree: LDS-resident (local ids) or in global memory (face ids; a subtree too large for LDS)
struct LdsAcc {
    SubLds *S; int sb;
    __device__ __forceinline__ int get(int buf, int pos) const { return S->ord[buf][pos - sb]; }
    __device__ __forceinline__ void put(int buf, int pos, int j) const { S->ord[buf][pos - sb] = (uint16_t)j; }
    __device__ __forceinline__ float cen(int j, int a) const { return S->cen[j * 3 + a]; }
    __device__ __forceinline__ float tb(int j, int q) const { return S->tbox[j * 6 + q]; }
    __device__ __forceinline__ int face(int j) const { return S->gid[j]; }
};
struct GlbAcc {
    int32_t *ord[2]; const float *c3; const float *b6;
    __device__ __forceinline__ int get(int buf, int pos) const { return ord[buf][pos]; }
    __device__ __forceinline__ void put(int buf, int pos, int j) const { ord[buf][pos] = j; }
    __device__ __forceinline__ float cen(int j, int a) const { return c3[3 * (size_t)j + a]; }
    __device__ __forceinline__ float tb(int j, int q) const { return b6[6 * (size_t)j + q]; }
    __device__ __forceinline__ int face(int j) const { return j; }
};

template <class A>
__device__ __forceinline__ void finalize_leaf(const BuildCtx &c, const A &acc, const STask &t, int lane)
{
    const int n = t.end - t.begin;
    if (lane < n) c.slot2face[t.begin + lane] = acc.face(acc.get(t.buf, t.begin + lane));
    if (lane == 0) {
        c.leaf_cnt[t.begin] = (uint8_t)n;
        link_node(c, t.parent, t.side, ~((t.begin << 2) | (n - 1)), t.box);
        atomicAdd(&c.dyn->n_leaves, 1);
        atomicMax(&c.dyn->depth, t.depth);
    }
}

// bounds of the elements [a, b) of a node (positional splits only)
template <class A>
__device__ __forceinline__ void range_bounds(const A &acc, int buf, int a, int b, int lane, float box[6], float cb[6])
{
    float v[12];
    for (int k = 0; k < 3; ++k) { v[k] = INFINITY; v[3 + k] = -INFINITY; v[6 + k] = INFINITY; v[9 + k] = -INFINITY; }
    for (int i = a + lane; i < b; i += 64) {
        const int j = acc.get(buf, i);
        for (int k = 0; k < 3; ++k) {
            v[k] = fminf(v[k], acc.tb(j, k)); v[3 + k] = fmaxf(v[3 + k], acc.tb(j, 3 + k));
            v[6 + k] = fminf(v[6 + k], acc.cen(j, k)); v[9 + k] = fmaxf(v[9 + k], acc.cen(j, k));
        }
    }
    for (int k = 0; k < 12; ++k) {
        const bool mn = (k % 6) < 3;
        for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_xor(v[k], d); v[k] = mn ? fminf(v[k], o) : fmaxf(v[k], o); }
    }
    for (int k = 0; k < 6; ++k) { box[k] = v[k]; cb[k] = v[6 + k]; }
}

// one node by one wavefront: a leaf is finished, anything else is split and its children that are not leaves are handed
// to `emit` (all lanes call it with the same task)
template <class A, class Emit>
__device__ __forceinline__ void process_node(const BuildCtx &c, const A &acc, uint32_t *h /* LDS, this wave's */, Decision *D /* LDS, this wave's */,
                                             const STask &t, int lane, Emit emit)
{
    const int n = t.end - t.begin;
    if (n <= kLeafMax) { finalize_leaf(c, acc, t, lane); return; }
    float lo[3], ext[3];
    for (int a = 0; a < 3; ++a) { lo[a] = t.cb[a]; ext[a] = t.cb[3 + a] - t.cb[a]; }
    bool valid = false;
    if (!force_median(t.depth, n, c.bound) && (ext[0] > 0.0f || ext[1] > 0.0f || ext[2] > 0.0f)) {
        for (int i = lane; i < kHistWords; i += 64) h[i] = 0;
        wave_sync();
        for (int i = lane; i < n; i += 64) {
            const int j = acc.get(t.buf, t.begin + i);
            hist_add(h, lo, ext, [&](int a) { return acc.cen(j, a); }, [&](int q) { return acc.tb(j, q); });
        }
        wave_sync();
        sah_choose(h, ext, lane, D);
        wave_sync();
        valid = D->valid != 0;
    }
    int nleft, cbuf;
    float cbox[2][6], ccb[2][6];
    if (valid) {
        const int axis = D->axis, bin = D->bin;
        nleft = D->nleft;
        for (int s = 0; s < 2; ++s)
            for (int k = 0; k < 6; ++k) { cbox[s][k] = D->cbox[s][k]; ccb[s][k] = D->ccb[s][k]; }
        int lb = 0, rb = nleft;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < n;
            const int j = act ? acc.get(t.buf, t.begin + i) : 0;
            const bool isl = act && sah_bin(acc.cen(j, axis), lo[axis], ext[axis]) <= bin;
            const unsigned long long bl = __ballot(isl), br = __ballot(act && !isl);
            if (isl) acc.put(t.buf ^ 1, t.begin + lb + __popcll(bl & lt), j);
            else if (act) acc.put(t.buf ^ 1, t.begin + rb + __popcll(br & lt), j);
            lb += __popcll(bl); rb += __popcll(br);
        }
        if (lb != nleft && lane == 0) atomicOr(&c.dyn->status, kMeshInternal);
        cbuf = t.buf ^ 1;
        wave_sync();
    } else {
        nleft = n / 2; cbuf = t.buf;
        range_bounds(acc, t.buf, t.begin, t.begin + nleft, lane, cbox[0], ccb[0]);
        range_bounds(acc, t.buf, t.begin + nleft, t.end, lane, cbox[1], ccb[1]);
    }
    const int mid = t.begin + nleft, id = mid - 1;
    if (lane == 0) { link_node(c, t.parent, t.side, id, t.box); atomicAdd(&c.dyn->n_nodes, 1); }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        STask ch;
        ch.begin = s ? mid : t.begin; ch.end = s ? t.end : mid; ch.depth = t.depth + 1; ch.parent = id; ch.side = s; ch.buf = cbuf;
        for (int k = 0; k < 6; ++k) { ch.box[k] = cbox[s][k]; ch.cb[k] = ccb[s][k]; }
        ch.pad[0] = ch.pad[1] = 0;
        if (ch.end - ch.begin <= kLeafMax) finalize_leaf(c, acc, ch, lane);
        else emit(ch);
    }
}

// the levels of one subtree.  `lists`: [2][cap] task lists (LDS, or - for a subtree too large for LDS - this subtree's
// slice of the arena), counts[2] in LDS; all `nwaves` waves of the workgroup call this together.
template <class A>
__device__ __forceinline__ void sub_levels(const BuildCtx &c, const A &acc, SubLds *S, STask *list0, STask *list1, int cap,
                                           int wave, int lane, int nwaves)
{
    STask *lists[2] = {list0, list1};
    for (int lv = 0;; ++lv) {
        __syncthreads();                                       // the level's list is complete, the other count is 0
        const int cur = lv & 1, nc = S->count[cur];
        if (nc == 0) break;
        for (int i = wave; i < nc; i += nwaves) {
            const STask t = lists[cur][i];                     // (a plain struct copy: punning it through int* broke under strict aliasing)
            process_node(c, acc, S->hist[wave], &S->dec[wave], t, lane, [&](const STask &ch) {
                int idx = 0;
                if (lane == 0) idx = atomicAdd(&S->count[cur ^ 1], 1);
                idx = __builtin_amdgcn_readfirstlane(idx);
                if (idx < cap) {
                    if (lane == 0) lists[cur ^ 1][idx] = ch;
                } else if (lane == 0) {
                    atomicOr(&c.dyn->status, kMeshInternal);
                }
            });
        }
        __syncthreads();                                       // every wave is done with this level
        if (threadIdx.x == 0) S->count[cur] = 0;
    }
}

__global__ __launch_bounds__(kSubWaves * 64) void k_bvh_sub(BuildCtx c)
{
    extern __shared__ __attribute__((aligned(16))) char sub_smem[];
    SubLds *S = reinterpret_cast<SubLds *>(sub_smem);
    if ((int)blockIdx.x >= c.hdr->n_sub) return;
    const BTask &T = c.tasks[c.subq[blockIdx.x]];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = T.end - T.begin;
    const bool in_lds = n <= kSubMax;
    if (threadIdx.x == 0) {
        STask r;
        r.begin = T.begin; r.end = T.end; r.depth = T.depth; r.parent = T.parent; r.side = T.side;
        r.buf = in_lds ? 0 : T.buf;
        task_bounds(T, r.box, r.cb);
        r.pad[0] = r.pad[1] = 0;
        (in_lds ? S->list[0] : c.sublist + (size_t)(T.begin / 5) * 2)[0] = r;
        S->count[0] = 1; S->count[1] = 0;
    }
    if (in_lds) {
        for (int i = threadIdx.x; i < n; i += kSubWaves * 64) {
            const int e = c.order[T.buf][T.begin + i];
            S->gid[i] = e; S->ord[0][i] = (uint16_t)i;
            for (int a = 0; a < 3; ++a) S->cen[i * 3 + a] = c.cen[3 * (size_t)e + a];
            for (int q = 0; q < 6; ++q) S->tbox[i * 6 + q] = c.tbox[6 * (size_t)e + q];
        }
        LdsAcc acc{S, T.begin};
        sub_levels(c, acc, S, S->list[0], S->list[1], kListCap, wave, lane, kSubWaves);
    } else {
        // an oversize subtree (pathologically unbalanced top levels): elements and task lists in global memory; the lists
        // are this subtree's slice of c.sublist - [begin / 5, begin / 5 + n / 5) entries of each list, disjoint between subtrees
        GlbAcc acc{{c.order[0], c.order[1]}, c.cen, c.tbox};
        STask *base = c.sublist + (size_t)(T.begin / 5) * 2;
        const int cap = n / 5;                                  // pending nodes hold >= 5 triangles each; 2 * cap entries fit the slice
        sub_levels(c, acc, S, base, base + cap, cap, wave, lane, kSubWaves);
    }
}

// ---------------------------------------------------------------------------------------------
// 4. slot-ordered records + ray bins
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ BinGrid grid_of(const BuildCtx &c) { return bin_grid(c.dyn->box_lo, c.dyn->box_hi, c.F); }
__device__ __forceinline__ void cell_range(const BinGrid &g, const float *tb, int &cy0, int &cy1, int &cz0, int &cz1)
{
    cy0 = bin_cell_of(tb[1] - kBinEps, g.y0, g.inv_y, g.gy); cy1 = bin_cell_of(tb[4] + kBinEps, g.y0, g.inv_y, g.gy);
    cz0 = bin_cell_of(tb[2] - kBinEps, g.z0, g.inv_z, g.gz); cz1 = bin_cell_of(tb[5] + kBinEps, g.z0, g.inv_z, g.gz);
}

__global__ __launch_bounds__(256) void k_tri_records(BuildCtx c)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= c.F) return;
    const int f = c.slot2face[p];
    int id[3]; float v[3][3]; int bf = 0, bv = 0;
    load_face(c, f, id, v, &bf, &bv);
    c.face2slot[f] = p;
    TriRec &tr = c.tris[p];
    for (int k = 0; k < 3; ++k) { tr.a[k] = v[0][k]; tr.b[k] = v[1][k]; tr.c[k] = v[2][k]; }
    tr.ia = id[0]; tr.ib = id[1]; tr.ic = id[2];
    TriAttr &at = c.attr[p];
    for (int q = 0; q < 3; ++q) {
        for (int k = 0; k < 3; ++k) { at.n[q][k] = c.vnormals[3 * (size_t)id[q] + k]; at.cm[q][k] = c.cmap[3 * (size_t)id[q] + k]; }
        at.vis[q] = c.vis[id[q]];
    }
    at.face = f; at.pad[0] = at.pad[1] = 0;
    // the leaf this slot belongs to: the leaf start among the four positions ending here
    int leaf = -1, myt = 0, cnt = 0;
    for (int t = 0; t < kLeafMax; ++t) {
        const int q = p - t;
        if (q >= 0 && c.leaf_cnt[q] > t) { leaf = q; myt = t; cnt = c.leaf_cnt[q]; break; }
    }
    if (leaf < 0) { atomicOr(&c.dyn->status, kMeshInternal); return; }
    TriPre pre;
    tri_setup(v[0], v[1], v[2], f, pre);
    const float *src = reinterpret_cast<const float *>(&pre);
    LeafRec &lr = c.leaves[leaf];
    const int tend = (myt == cnt - 1) ? kLeafMax : myt + 1;    // short leaves repeat their last triangle
    for (int t = myt; t < tend; ++t)
        for (int fld = 0; fld < 24; ++fld) lr.pair[t >> 1][fld][t & 1] = src[fld];
    // ray-bin cells covered by the (y,z) box of the triangle, grown by eps
    const BinGrid g = grid_of(c);
    int cy0, cy1, cz0, cz1;
    cell_range(g, c.tbox + 6 * (size_t)f, cy0, cy1, cz0, cz1);
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy) atomicAdd(&c.cell_count[(size_t)cz * g.gy + cy], 1);
}

// exclusive scan of the cell counts by one workgroup (every thread a contiguous segment)
__global__ __launch_bounds__(1024) void k_scan_cells(BuildCtx c)
{
    __shared__ int64_t wtot[16];
    __shared__ int64_t s_base[1024];
    const BinGrid g = grid_of(c);
    const int n = g.gy * g.gz;
    const int seg = (n + 1023) / 1024;
    const int a = min((int)threadIdx.x * seg, n), b = min(a + seg, n);
    int64_t sum = 0;
    for (int i = a; i < b; ++i) sum += c.cell_count[i];
    // block exclusive scan of the 1024 partial sums
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const int64_t up = __shfl_up(incl, d); if (lane >= d) incl += up; }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int64_t base = 0, all = 0;
    for (int q = 0; q < 16; ++q) { const int64_t t = wtot[q]; if (q < w) base += t; all += t; }
    s_base[threadIdx.x] = base + incl - sum;
    __syncthreads();
    const bool fits = all <= c.entries_cap;
    int64_t run = s_base[threadIdx.x];
    for (int i = a; i < b; ++i) { c.bin_start[i] = fits ? (int32_t)run : 0; run += c.cell_count[i]; }
    if (threadIdx.x == 0) {
        c.bin_start[n] = fits ? (int32_t)all : 0;
        c.dyn->bin_entries = fits ? (int32_t)all : 0;
        if (!fits) { c.dyn->gy = 0; c.dyn->gz = 0; atomicOr(&c.dyn->status, kMeshBinOverflow); }
    }
}

__global__ __launch_bounds__(256) void k_bin_fill(BuildCtx c)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= c.F || (c.dyn->status & kMeshBinOverflow)) return;
    const BinGrid g = grid_of(c);
    int cy0, cy1, cz0, cz1;
    cell_range(g, c.tbox + 6 * (size_t)c.slot2face[p], cy0, cy1, cz0, cz1);
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy) {
            const size_t cell = (size_t)cz * g.gy + cy;
            c.bin_slots[c.bin_start[cell] + atomicAdd(&c.cell_cursor[cell], 1)] = p;
        }
}

__global__ __launch_bounds__(256) void k_bin_sort(BuildCtx c)
{
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell == 0) atomicOr(&c.dyn->status, kMeshBuilt);
    if (c.dyn->status & kMeshBinOverflow) return;
    const BinGrid g = grid_of(c);
    if (cell >= (int64_t)g.gy * g.gz) return;
    const int a = c.bin_start[cell], b = c.bin_start[cell + 1];
    int32_t *q = c.bin_slots;
    for (int i = a + 1; i < b; ++i) {                          // ascending slots inside every bin
        const int key = q[i];
        int j = i - 1;
        while (j >= a && q[j] > key) { q[j + 1] = q[j]; --j; }
        q[j + 1] = key;
    }
    if (b - a > 0) atomicMax(&c.dyn->max_bin, b - a);
}

// ---- pinned host mirrors of MeshDyn + events, pooled: nothing is allocated per image in the steady state --------------
std::mutex g_pool_mu;
std::vector<MeshDyn *> g_free_dyn;
std::vector<hipEvent_t> g_free_ev;
// a handle destroyed while its build was still in flight: its mirror must not be handed out before the build's last copy has
// landed in it (it would report the OLD mesh's status for the new one)
std::vector<std::pair<MeshDyn *, hipEvent_t>> g_in_flight;

}  // namespace

int mesh_host_state_get(MeshDyn **h, hipEvent_t *ev)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_in_flight.size();) {
        if (hipEventQuery(g_in_flight[i].second) != hipErrorNotReady) {
            g_free_dyn.push_back(g_in_flight[i].first); g_free_ev.push_back(g_in_flight[i].second);
            g_in_flight[i] = g_in_flight.back(); g_in_flight.pop_back();
        } else ++i;
    }
    (void)hipGetLastError();
    if (g_free_dyn.empty()) {
        constexpr int kBatch = 32;
        MeshDyn *blk = nullptr;
        ICON_HIP(hipHostMalloc((void **)&blk, sizeof(MeshDyn) * kBatch, hipHostMallocDefault));    // never freed: lives with the process
        for (int i = 0; i < kBatch; ++i) g_free_dyn.push_back(blk + i);
    }
    if (g_free_ev.empty()) {
        hipEvent_t e;
        ICON_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g_free_ev.push_back(e);
    }
    *h = g_free_dyn.back(); g_free_dyn.pop_back();
    *ev = g_free_ev.back(); g_free_ev.pop_back();
    memset(*h, 0, sizeof(MeshDyn));
    return ICON_OK;
}

void mesh_host_state_put(MeshDyn *h, hipEvent_t ev)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (h && ev && hipEventQuery(ev) == hipErrorNotReady) { g_in_flight.emplace_back(h, ev); (void)hipGetLastError(); return; }
    (void)hipGetLastError();
    if (h) g_free_dyn.push_back(h);
    if (ev) g_free_ev.push_back(ev);
}

void mesh_bind_arena(icon_mesh *m, const MeshLayout &L)
{
    char *b = m->arena;
    m->d_dyn = reinterpret_cast<MeshDyn *>(b + L.dyn);
    m->d_vnormals = reinterpret_cast<float *>(b + L.vnormals);
    MeshDev &d = m->dev;
    d.nodes = reinterpret_cast<const BvhNode *>(b + L.nodes);
    d.tris = reinterpret_cast<const TriRec *>(b + L.tris);
    d.attr = reinterpret_cast<const TriAttr *>(b + L.attr);
    d.slot2face = reinterpret_cast<const int32_t *>(b + L.slot2face);
    d.face2slot = reinterpret_cast<const int32_t *>(b + L.face2slot);
    d.leaves = reinterpret_cast<const LeafRec *>(b + L.leaves);
    d.dyn = m->d_dyn;
    d.n_tris = (int32_t)m->F;
    d.bin_start = reinterpret_cast<const int32_t *>(b + L.bin_start);
    d.bin_slots = reinterpret_cast<const int32_t *>(b + L.bin_slots);
}

// enqueue the whole build on `st`; the mesh handle is usable by later work on the same stream as soon as this returns
int mesh_build_device(icon_mesh *m, const float *d_verts, const int64_t *d_faces, const float *d_cmap, const float *d_vis, hipStream_t st)
{
    const MeshLayout L = mesh_layout(m->V, m->F);
    char *b = m->arena;
    BuildCtx c{};
    c.verts = d_verts; c.faces = d_faces; c.cmap = d_cmap; c.vis = d_vis;
    c.V = (int32_t)m->V; c.F = (int32_t)m->F; c.bound = m->depth_bound;
    c.dyn = reinterpret_cast<MeshDyn *>(b + L.dyn); c.hdr = reinterpret_cast<BuildHdr *>(b + L.hdr);
    c.valence = reinterpret_cast<int32_t *>(b + L.valence); c.leaf_cnt = reinterpret_cast<uint8_t *>(b + L.leaf_cnt);
    c.tasks = reinterpret_cast<BTask *>(b + L.tasks); c.hist = reinterpret_cast<uint32_t *>(b + L.hist);
    c.cell_count = reinterpret_cast<int32_t *>(b + L.cell_count); c.cell_cursor = reinterpret_cast<int32_t *>(b + L.cell_cursor);
    c.vnormals = reinterpret_cast<float *>(b + L.vnormals); c.nodes = reinterpret_cast<BvhNode *>(b + L.nodes);
    c.leaves = reinterpret_cast<LeafRec *>(b + L.leaves); c.tris = reinterpret_cast<TriRec *>(b + L.tris);
    c.attr = reinterpret_cast<TriAttr *>(b + L.attr); c.slot2face = reinterpret_cast<int32_t *>(b + L.slot2face);
    c.face2slot = reinterpret_cast<int32_t *>(b + L.face2slot); c.bin_start = reinterpret_cast<int32_t *>(b + L.bin_start);
    c.bin_slots = reinterpret_cast<int32_t *>(b + L.bin_slots); c.tbox = reinterpret_cast<float *>(b + L.tbox);
    c.cen = reinterpret_cast<float *>(b + L.cen); c.order[0] = reinterpret_cast<int32_t *>(b + L.order0);
    c.order[1] = reinterpret_cast<int32_t *>(b + L.order1); c.adj = reinterpret_cast<int32_t *>(b + L.adj);
    c.chunkcnt = reinterpret_cast<int32_t *>(b + L.chunkcnt); c.subq = reinterpret_cast<int32_t *>(b + L.subq);
    c.sublist = reinterpret_cast<STask *>(b + L.sublist);
    c.nck = (int32_t)L.nck; c.cells_cap = bin_cells_cap(m->F); c.entries_cap = bin_entries_cap(m->F);

    ICON_HIP(hipMemsetAsync(b + L.dyn, 0, L.zero_end - L.dyn, st));
    const unsigned nbF = (unsigned)((m->F + 255) / 256), nbV = (unsigned)((m->V + 255) / 256);
    hipLaunchKernelGGL(k_face_prep, dim3(nbF), dim3(256), 0, st, c);
    debug_sync("k_face_prep", st);
    hipLaunchKernelGGL(k_vertex_normals, dim3(nbV), dim3(256), 0, st, c);
    debug_sync("k_vertex_normals", st);
    if (m->F > kSubMax) {
        for (int lv = 0; lv < kTopLevels; ++lv) {
            const unsigned nb = (unsigned)(m->F / kChunk + (1 << lv) + 1);
            hipLaunchKernelGGL(k_bvh_bin, dim3(nb), dim3(kChunk), 0, st, c, lv);
            hipLaunchKernelGGL(k_bvh_part, dim3(nb), dim3(kChunk), 0, st, c, lv);
            debug_sync("k_bvh_bin + k_bvh_part", st);
        }
    }
    static_assert(sizeof(SubLds) <= 160 * 1024 - 2048, "k_bvh_sub: LDS");
    {
        const int rc = once_per_device(8, [] { return hipFuncSetAttribute(reinterpret_cast<const void *>(k_bvh_sub), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SubLds)); });
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_bvh_sub, dim3((unsigned)(m->F > kSubMax ? kTaskSlots + 1 : 1)), dim3(kSubWaves * 64), sizeof(SubLds), st, c);
    debug_sync("k_bvh_sub", st);
    hipLaunchKernelGGL(k_tri_records, dim3(nbF), dim3(256), 0, st, c);
    debug_sync("k_tri_records", st);
    hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, st, c);
    hipLaunchKernelGGL(k_bin_fill, dim3(nbF), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_bin_sort, dim3((unsigned)((c.cells_cap + 255) / 256)), dim3(256), 0, st, c);
    debug_sync("ray bins", st);
    ICON_HIP(hipGetLastError());
    ICON_HIP(hipMemcpyAsync(m->h_dyn, c.dyn, sizeof(MeshDyn), hipMemcpyDeviceToHost, st));
    ICON_HIP(hipEventRecord(m->built, st));
    return ICON_OK;
}

}  // namespace icon
