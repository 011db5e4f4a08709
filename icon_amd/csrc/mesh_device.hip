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
    float *tbox, *cen; int32_t *order[2]; int32_t *adj; int32_t *chunkcnt; int32_t *subq; struct STask *sublist; uint32_t *bounds_part;
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

// lanes of ONE wavefront hand data to each other through LDS: the LDS operations of a wave complete in order, so all that is
// needed is their completion and a compiler barrier.  (A workgroup-scope fence also waits for the wave's outstanding GLOBAL
// stores - the node links and leaf records of the node before - 1-2 us each time: ~190 us of k_bvh_sub.)  GLOBAL = true: the
// data itself lives in global memory (subtrees too large for LDS): wait for those too.
template <bool GLOBAL = false>
__device__ __forceinline__ void wave_sync()
{
    if (GLOBAL) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    // mesh bounds: wave reduction -> workgroup (LDS) -> this workgroup's row of the partial array, plain stores (216 waves x 12
    // atomics on ONE cache line took 30 us: same-line device atomics serialise at ~12 ns each); k_vertex_normals reduces the rows
    __shared__ uint32_t s_part[4][12];
    uint32_t u[12];
    for (int a = 0; a < 3; ++a) {
        u[a] = live ? enc_min(lo[a]) : 0u; u[3 + a] = live ? enc(hi[a]) : 0u;
        u[6 + a] = live ? enc_min(ce[a]) : 0u; u[9 + a] = live ? enc(ce[a]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        uint32_t v = u[k];
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = v > o ? v : o; }
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const uint32_t a0 = s_part[0][threadIdx.x], a1 = s_part[1][threadIdx.x], a2 = s_part[2][threadIdx.x], a3 = s_part[3][threadIdx.x];
        const uint32_t m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3;
        c.bounds_part[(size_t)blockIdx.x * 12 + threadIdx.x] = m01 > m23 ? m01 : m23;
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
    __shared__ uint32_t s_red[256];
    if (blockIdx.x == 0) {
        // workgroup 0 first reduces the per-workgroup bounds of k_face_prep (12 maxima over nbF rows)
        const int nbF = (c.F + 255) / 256;
        const int k = threadIdx.x % 12, r0 = threadIdx.x / 12;             // 21 row groups x 12 values (threads 252..255 idle)
        uint32_t m = 0;
        if (threadIdx.x < 252)
            for (int r = r0; r < nbF; r += 21) { const uint32_t x = c.bounds_part[(size_t)r * 12 + k]; m = x > m ? x : m; }
        s_red[threadIdx.x] = m;
        __syncthreads();
        uint32_t mm = 0;
        if (threadIdx.x < 12)
            for (int g = 0; g < 21; ++g) { const uint32_t x = s_red[g * 12 + threadIdx.x]; mm = x > mm ? x : mm; }
        __syncthreads();
        if (threadIdx.x < 12) s_red[threadIdx.x] = mm;
        __syncthreads();
    }
    if (v == 0) {
        // the root task and what the query kernels need of the bounding box
        float box[6], cb[6];
        for (int a = 0; a < 3; ++a) {
            box[a] = dec_min(s_red[a]); box[3 + a] = dec(s_red[3 + a]);
            cb[a] = dec_min(s_red[6 + a]); cb[3 + a] = dec(s_red[9 + a]);
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
    // lane = axis * 16 + bin (48 lanes): every lane loads ITS bin, inclusive prefix (bins 0..b) and suffix (bins b..15) unions by
    // log-step scans inside the 16-lane rows; candidate b of an axis = prefix[b] | suffix[b + 1].  (A lane per candidate that
    // walked all 16 bins cost ~4 us per node - most of k_bvh_sub, whose nodes are mostly small.)
    const int ax = lane >> 4, b = lane & 15;
    const bool cell = lane < 48;
    const uint32_t *q = h + (cell ? lane : 0) * 13;
    int pc = cell ? (int)q[0] : 0;
    float pb[6], pcb[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pb[a] = (cell && pc) ? dec_min(q[1 + a]) : INFINITY; pb[3 + a] = (cell && pc) ? dec(q[4 + a]) : -INFINITY;
        pcb[a] = (cell && pc) ? dec_min(q[7 + a]) : INFINITY; pcb[3 + a] = (cell && pc) ? dec(q[10 + a]) : -INFINITY;
    }
    int sc = pc;
    float sb[6], scb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { sb[k] = pb[k]; scb[k] = pcb[k]; }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const bool up = b >= d, dn = b + d < 16;
        const int uc = __shfl_up(pc, d, 16), dc = __shfl_down(sc, d, 16);
        pc += up ? uc : 0; sc += dn ? dc : 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float ub = __shfl_up(pb[k], d, 16), ucb = __shfl_up(pcb[k], d, 16);
            const float db = __shfl_down(sb[k], d, 16), dcb = __shfl_down(scb[k], d, 16);
            if (k < 3) {
                pb[k] = up ? fminf(pb[k], ub) : pb[k]; pcb[k] = up ? fminf(pcb[k], ucb) : pcb[k];
                sb[k] = dn ? fminf(sb[k], db) : sb[k]; scb[k] = dn ? fminf(scb[k], dcb) : scb[k];
            } else {
                pb[k] = up ? fmaxf(pb[k], ub) : pb[k]; pcb[k] = up ? fmaxf(pcb[k], ucb) : pcb[k];
                sb[k] = dn ? fmaxf(sb[k], db) : sb[k]; scb[k] = dn ? fmaxf(scb[k], dcb) : scb[k];
            }
        }
    }
    // right side of candidate b: the suffix of bin b + 1
    const int cr = __shfl_down(sc, 1, 16);
    float Rb[6], Rc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { Rb[k] = __shfl_down(sb[k], 1, 16); Rc[k] = __shfl_down(scb[k], 1, 16); }
    double cost = INFINITY;
    const float ext_ax = ax == 0 ? ext[0] : (ax == 1 ? ext[1] : ext[2]);
    if (cell && b < 15 && ext_ax > 0.0f && pc && cr) cost = box_area(pb, pb + 3) * pc + box_area(Rb, Rb + 3) * cr;
    double best = cost;
    for (int d = 1; d < 64; d <<= 1) { const double o = __shfl_xor(best, d); best = o < best ? o : best; }
    const unsigned long long win = __ballot(cost == best && cost < (double)INFINITY);
    if (!win) { if (lane == 0) D->valid = 0; return; }
    const int w = __ffsll((long long)win) - 1;                 // the first minimum in (axis, bin) order, as the host's sweep finds it
    if (lane == w) {
        D->valid = 1; D->axis = ax; D->bin = b; D->nleft = pc;
#pragma unroll
        for (int k = 0; k < 6; ++k) { D->cbox[0][k] = pb[k]; D->cbox[1][k] = Rb[k]; D->ccb[0][k] = pcb[k]; D->ccb[1][k] = Rc[k]; }
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
    const int lane = threadIdx.x & 63;
    int found = -1, fk = 0;
    for (int s = lane; s < nslots; s += 64) {                  // every wave scans the (<= 2^kTopLevels) slots with its 64 lanes
        if (T[s].kind != 1) continue;
        const int base = T[s].begin / kChunk + s, nc = (T[s].end - T[s].begin + kChunk - 1) / kChunk;
        if (w >= base && w < base + nc) { found = s; fk = w - base; }
    }
    const unsigned long long hit = __ballot(found >= 0);
    if (!hit) return false;
    const int src = __ffsll((long long)hit) - 1;              // at most one slot owns workgroup w
    slot = __shfl(found, src); k = __shfl(fk, src);
    return true;
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
        const int e = ((L & 1) ? c.order[1] : c.order[0])[P.begin + i];
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
    __shared__ uint32_t s_hist[kHistWords];
    if (!forced) {                                             // the node's finished histogram: one coalesced read into LDS
        const uint32_t *g = c.hist + (size_t)((1 << L) - 1 + s) * kHistWords;
        for (int q = threadIdx.x; q < kHistWords; q += kChunk) s_hist[q] = g[q];
    }
    __syncthreads();
    if (wave == 0) {
        if (forced) { if (lane == 0) D.valid = 0; }
        else sah_choose(s_hist, ext, lane, &D);
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
    const int e = act ? ((L & 1) ? c.order[1] : c.order[0])[P.begin + i] : 0;
    const bool isl = act && (valid ? sah_bin(c.cen[3 * (size_t)e + axis], lo[axis], ext[axis]) <= bin : i < nleft);
    const unsigned long long bl = __ballot(isl);
    if (lane == 0) s_wcnt[wave] = __popcll(bl);
    __syncthreads();
    int lrank = __popcll(bl & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) lrank += s_wcnt[w];
    const int loff = s_loff;
    if (act) {
        const int dest = isl ? P.begin + loff + lrank : P.begin + nleft + (k * kChunk - loff) + ((int)threadIdx.x - lrank);
        (((L + 1) & 1) ? c.order[1] : c.order[0])[dest] = e;
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
constexpr int kSubWaves = 16;
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
    int n_nodes, n_leaves, depth;      // this subtree's share of the statistics: ONE global atomic each at the end (13,500
                                       // fire-and-forget atomics on one cache line of MeshDyn were ~160 us of the kernel)
};

// element access of a subtree: LDS-resident (local ids) or in global memory (face ids; a subtree too large for LDS)
struct LdsAcc {
    static constexpr bool kGlobal = false;
    SubLds *S; int sb;
    __device__ __forceinline__ int get(int buf, int pos) const { return S->ord[buf][pos - sb]; }
    __device__ __forceinline__ void put(int buf, int pos, int j) const { S->ord[buf][pos - sb] = (uint16_t)j; }
    __device__ __forceinline__ float cen(int j, int a) const { return S->cen[j * 3 + a]; }
    __device__ __forceinline__ float tb(int j, int q) const { return S->tbox[j * 6 + q]; }
    __device__ __forceinline__ int face(int j) const { return S->gid[j]; }
};
struct GlbAcc {
    static constexpr bool kGlobal = true;
    int32_t *ord[2]; const float *c3; const float *b6;
    __device__ __forceinline__ int get(int buf, int pos) const { return (buf ? ord[1] : ord[0])[pos]; }
    __device__ __forceinline__ void put(int buf, int pos, int j) const { (buf ? ord[1] : ord[0])[pos] = j; }
    __device__ __forceinline__ float cen(int j, int a) const { return c3[3 * (size_t)j + a]; }
    __device__ __forceinline__ float tb(int j, int q) const { return b6[6 * (size_t)j + q]; }
    __device__ __forceinline__ int face(int j) const { return j; }
};

template <class A>
__device__ __forceinline__ void finalize_leaf(const BuildCtx &c, const A &acc, SubLds *S, const STask &t, int lane)
{
    const int n = t.end - t.begin;
    if (lane < n) c.slot2face[t.begin + lane] = acc.face(acc.get(t.buf, t.begin + lane));
    if (lane == 0) {
        c.leaf_cnt[t.begin] = (uint8_t)n;
        link_node(c, t.parent, t.side, ~((t.begin << 2) | (n - 1)), t.box);
        atomicAdd(&S->n_leaves, 1);
        atomicMax(&S->depth, t.depth);
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
__device__ __forceinline__ void process_node(const BuildCtx &c, const A &acc, SubLds *S, uint32_t *h /* LDS, this wave's */, Decision *D /* LDS, this wave's */,
                                             const STask &t, int lane, Emit emit)
{
    const int n = t.end - t.begin;
    if (n <= kLeafMax) { finalize_leaf(c, acc, S, t, lane); return; }
    float lo[3], ext[3];
    for (int a = 0; a < 3; ++a) { lo[a] = t.cb[a]; ext[a] = t.cb[3 + a] - t.cb[a]; }
    bool valid = false;
    if (!force_median(t.depth, n, c.bound) && (ext[0] > 0.0f || ext[1] > 0.0f || ext[2] > 0.0f)) {
        for (int i = lane; i < kHistWords; i += 64) h[i] = 0;
        wave_sync<A::kGlobal>();
        for (int i = lane; i < n; i += 64) {
            const int j = acc.get(t.buf, t.begin + i);
            hist_add(h, lo, ext, [&](int a) { return acc.cen(j, a); }, [&](int q) { return acc.tb(j, q); });
        }
        wave_sync<A::kGlobal>();
        sah_choose(h, ext, lane, D);
        wave_sync<A::kGlobal>();
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
        const float lo_ax = axis == 0 ? lo[0] : (axis == 1 ? lo[1] : lo[2]), ext_ax = axis == 0 ? ext[0] : (axis == 1 ? ext[1] : ext[2]);
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < n;
            const int j = act ? acc.get(t.buf, t.begin + i) : 0;
            const bool isl = act && sah_bin(acc.cen(j, axis), lo_ax, ext_ax) <= bin;
            const unsigned long long bl = __ballot(isl), br = __ballot(act && !isl);
            if (isl) acc.put(t.buf ^ 1, t.begin + lb + __popcll(bl & lt), j);
            else if (act) acc.put(t.buf ^ 1, t.begin + rb + __popcll(br & lt), j);
            lb += __popcll(bl); rb += __popcll(br);
        }
        if (lb != nleft && lane == 0) atomicOr(&c.dyn->status, kMeshInternal);
        cbuf = t.buf ^ 1;
        wave_sync<A::kGlobal>();
    } else {
        nleft = n / 2; cbuf = t.buf;
        range_bounds(acc, t.buf, t.begin, t.begin + nleft, lane, cbox[0], ccb[0]);
        range_bounds(acc, t.buf, t.begin + nleft, t.end, lane, cbox[1], ccb[1]);
    }
    const int mid = t.begin + nleft, id = mid - 1;
    if (lane == 0) { link_node(c, t.parent, t.side, id, t.box); atomicAdd(&S->n_nodes, 1); }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        STask ch;
        ch.begin = s ? mid : t.begin; ch.end = s ? t.end : mid; ch.depth = t.depth + 1; ch.parent = id; ch.side = s; ch.buf = cbuf;
        for (int k = 0; k < 6; ++k) { ch.box[k] = cbox[s][k]; ch.cb[k] = ccb[s][k]; }
        ch.pad[0] = ch.pad[1] = 0;
        if (ch.end - ch.begin <= kLeafMax) finalize_leaf(c, acc, S, ch, lane);
        else emit(ch);
    }
}

// the levels of one subtree.  `lists`: [2][cap] task lists (LDS, or - for a subtree too large for LDS - this subtree's
// slice of the arena), counts[2] in LDS; all `nwaves` waves of the workgroup call this together.
template <class A>
__device__ __forceinline__ void sub_levels(const BuildCtx &c, const A &acc, SubLds *S, STask *list0, STask *list1, int cap,
                                           int wave, int lane, int nwaves)
{
    for (int lv = 0;; ++lv) {
        __syncthreads();                                       // the level's list is complete, the other count is 0
        const int cur = lv & 1, nc = S->count[cur];
        if (nc == 0) break;
        // (selected, not indexed: a two-entry pointer array indexed by `cur` lives in scratch memory - a global-memory
        //  round trip per access, several per node)
        STask *const lcur = cur ? list1 : list0, *const lnext = cur ? list0 : list1;
        for (int i = wave; i < nc; i += nwaves) {
            const STask t = lcur[i];                     // (a plain struct copy: punning it through int* broke under strict aliasing)
            process_node(c, acc, S, S->hist[wave], &S->dec[wave], t, lane, [&](const STask &ch) {
                int idx = 0;
                if (lane == 0) idx = atomicAdd(&S->count[cur ^ 1], 1);
                idx = __builtin_amdgcn_readfirstlane(idx);
                if (idx < cap) {
                    if (lane == 0) lnext[idx] = ch;
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
        S->n_nodes = 0; S->n_leaves = 0; S->depth = 0;
    }
    if (in_lds) {
        for (int i = threadIdx.x; i < n; i += kSubWaves * 64) {
            const int e = (T.buf ? c.order[1] : c.order[0])[T.begin + i];
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
    if (threadIdx.x == 0) {                                     // (sub_levels ends behind a barrier: the counts are final)
        if (S->n_nodes) atomicAdd(&c.dyn->n_nodes, S->n_nodes);
        atomicAdd(&c.dyn->n_leaves, S->n_leaves);
        atomicMax(&c.dyn->depth, S->depth);
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

// exclusive scan of the cell counts by one workgroup: tiles of 4,096 cells (four per thread, coalesced), a running carry
__global__ __launch_bounds__(1024) void k_scan_cells(BuildCtx c)
{
    __shared__ int wtot[16];
    const BinGrid g = grid_of(c);
    const int n = g.gy * g.gz;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int i = base + (int)threadIdx.x * 4;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i + k < n) ? c.cell_count[i + k] : 0;
        const int mine = v[0] + v[1] + v[2] + v[3];
        int incl = mine;
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
        __syncthreads();                                       // (the previous tile's wtot has been read)
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int t = wtot[q]; before += q < w ? t : 0; all += t; }
        int64_t run = carry + before + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (i + k < n) c.bin_start[i + k] = (int32_t)run; run += v[k]; }
        carry += all;
    }
    const bool fits = carry <= c.entries_cap;
    __syncthreads();                                           // (every thread has written its last tile)
    if (!fits)                                                  // as the host build: no lists at all (brute-force parity count)
        for (int i = threadIdx.x; i <= n; i += 1024) c.bin_start[i] = 0;
    if (threadIdx.x == 0) {
        if (fits) c.bin_start[n] = (int32_t)carry;
        c.dyn->bin_entries = fits ? (int32_t)carry : 0;
        if (!fits) { c.dyn->gy = 0; c.dyn->gz = 0; atomicOr(&c.dyn->status, kMeshBinOverflow); }
    }
}

__global__ __launch_bounds__(256) void k_bin_fill(BuildCtx c)
{
    // eight threads per triangle, the cells of its range dealt round-robin: ~3 returning atomics in a row per thread, not ~19
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = t >> 3, sub = t & 7;
    if (p >= c.F || (c.dyn->status & kMeshBinOverflow)) return;
    const BinGrid g = grid_of(c);
    int cy0, cy1, cz0, cz1;
    cell_range(g, c.tbox + 6 * (size_t)c.slot2face[p], cy0, cy1, cz0, cz1);
    int q = 0;
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy, ++q) {
            if ((q & 7) != sub) continue;
            const size_t cell = (size_t)cz * g.gy + cy;
            c.bin_slots[c.bin_start[cell] + atomicAdd(&c.cell_cursor[cell], 1)] = p;
        }
}

template <int N>
__device__ __forceinline__ void sort_regs(int32_t *q, int len)
{
    int v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = k < len ? q[k] : 0x7fffffff;
    // Batcher's odd-even mergesort: every index below is a compile-time constant after unrolling
#pragma unroll
    for (int p = 1; p < N; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j + k < N; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < N && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const int lo = min(v[i + j], v[i + j + k]), hi = max(v[i + j], v[i + j + k]);
                        v[i + j] = lo; v[i + j + k] = hi;
                    }
#pragma unroll
    for (int k = 0; k < N; ++k) if (k < len) q[k] = v[k];
}

__global__ __launch_bounds__(256) void k_bin_sort(BuildCtx c)
{
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell == 0) atomicOr(&c.dyn->status, kMeshBuilt);
    if (c.dyn->status & kMeshBinOverflow) return;
    const BinGrid g = grid_of(c);
    const bool live = cell < (int64_t)g.gy * g.gz;
    const int a = live ? c.bin_start[cell] : 0, b = live ? c.bin_start[cell + 1] : 0, len = b - a;
    {   // longest list: one atomic per wavefront (27,648 atomics on one address were ~165 us of this kernel)
        int m = len;
        for (int d = 1; d < 64; d <<= 1) m = max(m, __shfl_xor(m, d));
        if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&c.dyn->max_bin, m);
    }
    int32_t *q = c.bin_slots + a;
    // ascending slots inside every bin, as a sequential fill leaves them: the list in registers - ONE round of loads, Batcher's
    // odd-even merge network, one round of stores.  (Sorting in place in global memory is a chain of dependent round trips:
    // the few lists of 33..49 entries that took that path cost 170 us - the whole kernel.)
    if (len > 1 && len <= 32) sort_regs<32>(q, len);
    else if (len > 32 && len <= 64) sort_regs<64>(q, len);
    else if (len > 64) {
        for (int i = 1; i < len; ++i) {
            const int key = q[i];
            int j = i - 1;
            while (j >= 0 && q[j] > key) { q[j + 1] = q[j]; --j; }
            q[j + 1] = key;
        }
    }
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
    c.sublist = reinterpret_cast<STask *>(b + L.sublist); c.bounds_part = reinterpret_cast<uint32_t *>(b + L.bounds_part);
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
    hipLaunchKernelGGL(k_bin_fill, dim3((unsigned)((m->F * 8 + 255) / 256)), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_bin_sort, dim3((unsigned)((c.cells_cap + 255) / 256)), dim3(256), 0, st, c);
    debug_sync("ray bins", st);
    ICON_HIP(hipGetLastError());
    ICON_HIP(hipMemcpyAsync(m->h_dyn, c.dyn, sizeof(MeshDyn), hipMemcpyDeviceToHost, st));
    ICON_HIP(hipEventRecord(m->built, st));
    return ICON_OK;
}

}  // namespace icon
