// mc_device.hip - marching cubes on the GPU for Seg3dLossless.export_mesh
// (reference lib/common/seg3d_lossless.py:583-604; the reference itself uses kaolin's CUDA marching
// cubes for grids <= 256^3).  SURVEY.md §8f row 1: once the field takes ~20 ms, a host marching
// cubes (0.2 s + a 68 MB device->host copy) dominates the image; this keeps the volume on the
// device and ships only the mesh (~2 MB).
//
// Same case table, same vertex placement arithmetic and the same output conventions as the host
// implementation in mcubes.cpp (tests compare the two as sets): samples are occ[1:,1:,1:]; vertices
// (x,y,z) in cropped-grid index units; triangle normals point from inside (occ > level) to outside.
//
//   k_mc_classify : per sample point - crossing flags of its +x/+y/+z edges and, as the cube's
//                   low corner, the cube's triangle count               -> 1 byte + 1 byte / point
//   k_mc_blocksum / k_mc_scanblocks / k_mc_vertices : exclusive scan of (vertex, triangle) counts packed
//                   in one 64-bit word (1024 points per block); k_mc_vertices also writes the
//                   interpolated vertices and each point's first vertex index
//   k_mc_faces    : per cube - triangles with vertex ids looked up through the first-index array
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace icon {

void mc_tables(int8_t table[256][16], int8_t edges[12][2]);   // mcubes.cpp

namespace {

constexpr int kMcBlock = 1024;

struct McTab {
    int8_t tri[256][16];
    int8_t edge[12][2];
};
__constant__ McTab c_mc;
std::once_flag g_mc_once;
hipError_t g_mc_err = hipSuccess;

__device__ __forceinline__ float mc_at(const float *occ, int res, int z, int y, int x)
{
    return occ[((size_t)(z + 1) * res + (y + 1)) * res + (x + 1)];
}

// cell index -> (x, y, z), x fastest.  32-bit arithmetic whenever the index allows it (every res <= 1291 does: 1290^3 < 2^31.1 -
// checked per call): a 64-bit division by a run-time n is ~40 instructions, three of them per cell were most of the
// classification pass (97 us for 16.8 M cells)
__device__ __forceinline__ void mc_cell(int64_t i, int n, int &x, int &y, int &z)
{
    if (i < (1ll << 31)) {
        const uint32_t ii = (uint32_t)i, un = (uint32_t)n;
        const uint32_t r = ii / un;
        x = (int)(ii - r * un);
        const uint32_t zz = r / un;
        y = (int)(r - zz * un); z = (int)zz;
    } else {
        const int64_t r = i / n;
        x = (int)(i - r * n); z = (int)(r / n); y = (int)(r - (int64_t)z * n);
    }
}

__device__ __forceinline__ float mc_lerp(float level, float v0, float v1)
{
    const float d = v1 - v0;
    const float t = (d != 0.0f) ? (level - v0) / d : 0.5f;
    return fminf(fmaxf(t, 0.0f), 1.0f);
}

// flags: bit0/1/2 = +x/+y/+z edge crossed, bit3 = inside
__device__ __forceinline__ unsigned long long mc_block_exscan(unsigned long long v, unsigned long long *wsum, unsigned long long *total);

// also: block_tot[block] = (triangles << 32 | vertices) of the block's cells - the blocks away from the surface (99 % of them)
// are skipped by the emit passes on that one word
// [zc0, zc1): the cell layers this call triangulates (the whole cube: 0, res - 1).  A Z-slab rank of the sharded driver owns a
// range of layers and holds the planes they read only (`occ` is then a VIRTUAL base pointer: plane z of the full volume at
// occ + z res^2, dereferenced for its own planes alone); with `halo` the layer zc1 - whose corner plane is the neighbour's first
// plane - is classified for what the layer below needs from it: the inside bit and the crossings of its x / y edges (their
// vertices are emitted a second time here, keyed, and merged after the exchange), no z edges, no triangles.  Everything else: 0.
__global__ __launch_bounds__(kMcBlock) void k_mc_classify(const float *__restrict__ occ, int res, float level, int64_t npts,
                                                          uint8_t *__restrict__ flags, uint8_t *__restrict__ ntri,
                                                          unsigned long long *__restrict__ block_tot, int zc0, int zc1, int halo)
{
    __shared__ unsigned wsum[kMcBlock / 64];
    const int n = res - 1;
    const int64_t i = (int64_t)blockIdx.x * kMcBlock + threadIdx.x;
    unsigned mine = 0;                                           // vertices | triangles << 16 of this cell (<= 3, <= 5)
    if (i < npts) {
    int x, y, z; mc_cell(i, n, x, y, z);
    const bool own = z >= zc0 && z < zc1, hal = halo && z == zc1;
    if (!own && !hal) { flags[i] = 0; ntri[i] = 0; }
    else {
    // the eight corners from ONE base address and three clamped strides (a corner beyond the last cell repeats the last one):
    // per-corner min() / multiply chains were most of this kernel's instructions
    const float *p0 = occ + ((size_t)(z + 1) * res + (y + 1)) * res + (x + 1);
    const size_t dx = x + 1 < n ? 1 : 0, dy = y + 1 < n ? (size_t)res : 0, dz = (z + 1 < n && own) ? (size_t)res * res : 0;
    bool in[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = p0[((k & 1) ? dx : 0) + ((k & 2) ? dy : 0) + ((k & 4) ? dz : 0)] > level;
    uint8_t f = in[0] ? 8 : 0;
    if (x + 1 < n && in[1] != in[0]) f |= 1;
    if (y + 1 < n && in[2] != in[0]) f |= 2;
    if (own && z + 1 < n && in[4] != in[0]) f |= 4;
    flags[i] = f;
    uint8_t t = 0;
    if (own && x + 1 < n && y + 1 < n && z + 1 < n) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c |= in[k] ? (1 << k) : 0;
        t = (uint8_t)c_mc.tri[c][0];
    }
    ntri[i] = t;
    mine = (unsigned)__popc(f & 7) | ((unsigned)t << 16);
    }
    }
    // block total: a 32-bit wave reduction of the packed pair (a wave's sums are <= 192 and <= 320), 16 partial sums through LDS
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long nv = 0, nt = 0;
        for (int k = 0; k < kMcBlock / 64; ++k) { nv += wsum[k] & 0xffffu; nt += wsum[k] >> 16; }
        block_tot[blockIdx.x] = nv | (nt << 32);
    }
}

// The same pass with FOUR consecutive cells of an x-row per thread (n % 4 == 0: every standard size, n = 256 / 512): the 4 cells'
// 32 corners are 20 distinct values (5 along x in each of the 4 rows around the cells), the row addresses and the cell index are
// computed once, flags / triangle counts leave as one aligned 32-bit word each.  A 513^3 volume is 134 M cells: the one-cell
// form took 0.67 ms there (8 loads + a full index decode per cell; 0.8 TB/s of the 540 MB it has to read), the reference's
// default mcube_res = 512 makes that the common case.  256 threads = the same 1,024-cell blocks as above (block_tot, the scan
// and the emit passes are unchanged); bit-for-bit the same flags / counts (tests compare device and host marching cubes as sets).
__global__ __launch_bounds__(kMcBlock / 4) void k_mc_classify4(const float *__restrict__ occ, int res, float level, int64_t npts,
                                                              uint8_t *__restrict__ flags, uint8_t *__restrict__ ntri,
                                                              unsigned long long *__restrict__ block_tot, int zc0, int zc1, int halo)
{
    __shared__ unsigned wsum[kMcBlock / 4 / 64];
    const int n = res - 1;
    const int64_t i = ((int64_t)blockIdx.x * (kMcBlock / 4) + threadIdx.x) * 4;      // first of this thread's four cells (same row: n % 4 == 0)
    unsigned mine = 0;
    if (i < npts) {
        int x, y, z; mc_cell(i, n, x, y, z);
        const bool own = z >= zc0 && z < zc1, hal = halo && z == zc1;
        uint32_t f4 = 0, t4 = 0;
        if (own || hal) {
            const float *p0 = occ + ((size_t)(z + 1) * res + (y + 1)) * res + (x + 1);
            const size_t dy = y + 1 < n ? (size_t)res : 0, dz = (z + 1 < n && own) ? (size_t)res * res : 0;
            const int last = (x + 4 < n) ? 4 : 3;                 // the corner beyond the row's last cell repeats the last one (dx = 0 there)
            bool in[4][5];                                        // [row: (y, z), (y+1, z), (y, z+1), (y+1, z+1)][x .. x+4]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *q = p0 + ((r & 1) ? dy : 0) + ((r & 2) ? dz : 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) in[r][k] = q[k] > level;
                in[r][4] = q[last] > level;
            }
            const bool cube_row = own && y + 1 < n && z + 1 < n;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool has_x = x + k + 1 < n;
                uint32_t f = in[0][k] ? 8u : 0u;
                if (has_x && in[0][k + 1] != in[0][k]) f |= 1u;
                if (y + 1 < n && in[1][k] != in[0][k]) f |= 2u;
                if (own && z + 1 < n && in[2][k] != in[0][k]) f |= 4u;
                uint32_t t = 0;
                if (cube_row && has_x) {
                    const int c = (in[0][k] ? 1 : 0) | (in[0][k + 1] ? 2 : 0) | (in[1][k] ? 4 : 0) | (in[1][k + 1] ? 8 : 0) |
                                  (in[2][k] ? 16 : 0) | (in[2][k + 1] ? 32 : 0) | (in[3][k] ? 64 : 0) | (in[3][k + 1] ? 128 : 0);
                    t = (uint32_t)(uint8_t)c_mc.tri[c][0];
                }
                f4 |= f << (8 * k); t4 |= t << (8 * k);
                mine += (unsigned)__popc(f & 7u) | (t << 16);
            }
        }
        *reinterpret_cast<uint32_t *>(flags + i) = f4;            // i % 4 == 0 and the arrays come from hipMalloc: aligned
        *reinterpret_cast<uint32_t *>(ntri + i) = t4;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long nv = 0, nt = 0;
        for (int k = 0; k < kMcBlock / 4 / 64; ++k) { nv += wsum[k] & 0xffffu; nt += wsum[k] >> 16; }
        block_tot[blockIdx.x] = nv | (nt << 32);
    }
}

__device__ __forceinline__ unsigned long long mc_count(const uint8_t *flags, const uint8_t *ntri, int64_t i, int64_t npts)
{
    if (i >= npts) return 0ull;
    return (unsigned long long)__popc(flags[i] & 7) | ((unsigned long long)ntri[i] << 32);
}

// block-wide exclusive scan of one 64-bit value per thread (two independent 32-bit sums packed)
__device__ __forceinline__ unsigned long long mc_block_exscan(unsigned long long v, unsigned long long *wsum /* LDS [16] */,
                                                              unsigned long long *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (int k = 0; k < kMcBlock / 64; ++k) { const unsigned long long s = wsum[k]; if (k < w) before += s; all += s; }
    if (total) *total = all;
    return before + inc - v;
}

// exclusive scan of the block totals (66,000 for a 257^3 volume) into offsets: one workgroup, eight consecutive entries per
// thread and round - coalesced (a contiguous range per thread was a 41 us pass of strided loads)
// ... and the list of the blocks that hold anything (any order): the emit passes are launched over those only (600 of 66,000
// for the body at 257^3: 66,000 workgroups that read one word and leave were 30 us per pass)
__global__ __launch_bounds__(1024) void k_mc_scanblocks(const unsigned long long *__restrict__ block_tot, unsigned long long *__restrict__ block_off,
                                                        int64_t nblocks, unsigned long long *totals, int32_t *__restrict__ active)
{
    __shared__ unsigned long long wtot[16];
    __shared__ int n_active;
    if (threadIdx.x == 0) n_active = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long carry = 0;
    for (int64_t base = 0; base < nblocks; base += 8192) {
        const int64_t i = base + (int64_t)threadIdx.x * 8;
        unsigned long long v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = i + k < nblocks ? block_tot[i + k] : 0ull; sum += v[k]; }
        unsigned long long incl = sum;
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(incl, d); if (lane >= d) incl += up; }
        __syncthreads();
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        unsigned long long before = 0, all = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const unsigned long long t = wtot[q]; before += q < w ? t : 0ull; all += t; }
        unsigned long long run = carry + before + incl - sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i + k < nblocks) block_off[i + k] = run;
            if (v[k]) active[atomicAdd(&n_active, 1)] = (int32_t)(i + k);     // (LDS counter; the first __syncthreads of the round ordered its reset)
            run += v[k];
        }
        carry += all;
    }
    __syncthreads();
    if (threadIdx.x == 0) { totals[0] = carry & 0xffffffffull; totals[1] = carry >> 32; totals[2] = (unsigned long long)n_active; }
}

// The same scan in two parallel passes (round 5): a 513^3 volume has 131,072 block totals and the one-workgroup form above walks them
// in 16 rounds - 124 us, as much as the face pass.  k_mc_scan_local: one workgroup per 8,192 totals, prefix inside the chunk +
// the chunk's total; k_mc_scan_apply: every workgroup adds the totals of the chunks before it (<= 64 of them: summed by one wave),
// appends its non-empty blocks to the active list (any order) and the last chunk writes the grand totals.
constexpr int kMcChunk = 8192, kMcMaxChunks = 4096;
__global__ __launch_bounds__(1024) void k_mc_scan_local(const unsigned long long *__restrict__ block_tot, unsigned long long *__restrict__ block_off,
                                                        int64_t nblocks, unsigned long long *__restrict__ chunk_tot, unsigned long long *totals)
{
    __shared__ unsigned long long wtot[16];
    if (blockIdx.x == 0 && threadIdx.x == 0) totals[2] = 0ull;   // the active-list counter of k_mc_scan_apply
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kMcChunk + (int64_t)threadIdx.x * 8;
    unsigned long long v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = i + k < nblocks ? block_tot[i + k] : 0ull; sum += v[k]; }
    unsigned long long incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(incl, d); if (lane >= d) incl += up; }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const unsigned long long t = wtot[q]; before += q < w ? t : 0ull; all += t; }
    unsigned long long run = before + incl - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (i + k < nblocks) block_off[i + k] = run; run += v[k]; }
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = all;
}
__global__ __launch_bounds__(1024) void k_mc_scan_apply(const unsigned long long *__restrict__ block_tot, unsigned long long *__restrict__ block_off,
                                                        int64_t nblocks, const unsigned long long *__restrict__ chunk_tot, int nchunks,
                                                        unsigned long long *totals, int32_t *__restrict__ active)
{
    __shared__ unsigned long long base_s;
    if (threadIdx.x < 64) {                                      // the chunks before this one (and, for the last chunk, all of them)
        unsigned long long b = 0, a = 0;
        for (int c = threadIdx.x; c < nchunks; c += 64) { const unsigned long long t = chunk_tot[c]; a += t; b += c < (int)blockIdx.x ? t : 0ull; }
        for (int d = 32; d >= 1; d >>= 1) { b += __shfl_xor(b, d); a += __shfl_xor(a, d); }
        if (threadIdx.x == 0) {
            base_s = b;
            if ((int)blockIdx.x == nchunks - 1) { totals[0] = a & 0xffffffffull; totals[1] = a >> 32; }
        }
    }
    __syncthreads();
    const unsigned long long base = base_s;
    const int64_t i0 = (int64_t)blockIdx.x * kMcChunk + (int64_t)threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t i = i0 + k;
        if (i >= nblocks) break;
        block_off[i] += base;
        if (block_tot[i]) active[atomicAdd(&totals[2], 1ull)] = (int32_t)i;
    }
}

__global__ __launch_bounds__(kMcBlock) void k_mc_vertices(const float *__restrict__ occ, int res, float level, int64_t npts,
                                                          const uint8_t *__restrict__ flags, const uint8_t *__restrict__ ntri,
                                                          const int32_t *__restrict__ active,
                                                          const unsigned long long *__restrict__ block_offsets,
                                                          int32_t *__restrict__ first_vertex, int32_t *__restrict__ first_tri,
                                                          float *__restrict__ verts, int64_t *__restrict__ keys)
{
    __shared__ unsigned long long wsum[kMcBlock / 64];
    const int64_t blk = active[blockIdx.x];                      // a block with a vertex or a triangle (nobody reads the first_* entries of the others)
    const int n = res - 1;
    const int64_t i = blk * kMcBlock + threadIdx.x;
    const unsigned long long own = mc_count(flags, ntri, i, npts);
    const unsigned long long ex = mc_block_exscan(own, wsum, nullptr) + block_offsets[blk];
    if (i >= npts || own == 0ull) return;                        // first_vertex is read for vertex owners, first_tri for cells with triangles
    const int32_t v0 = (int32_t)(ex & 0xffffffffull);
    first_vertex[i] = v0;
    first_tri[i] = (int32_t)(ex >> 32);
    const uint8_t f = flags[i];
    if (!(f & 7)) return;
    int x, y, z; mc_cell(i, n, x, y, z);
    const float v = mc_at(occ, res, z, y, x);
    int k = v0;
    // keys (optional): 3 * cell + edge direction - the vertex's place in the order of the whole-volume call
    if (f & 1) { const float t = mc_lerp(level, v, mc_at(occ, res, z, y, x + 1)); verts[3 * k] = x + t; verts[3 * k + 1] = (float)y; verts[3 * k + 2] = (float)z; if (keys) keys[k] = 3 * i; ++k; }
    if (f & 2) { const float t = mc_lerp(level, v, mc_at(occ, res, z, y + 1, x)); verts[3 * k] = (float)x; verts[3 * k + 1] = y + t; verts[3 * k + 2] = (float)z; if (keys) keys[k] = 3 * i + 1; ++k; }
    if (f & 4) { const float t = mc_lerp(level, v, mc_at(occ, res, z + 1, y, x)); verts[3 * k] = (float)x; verts[3 * k + 1] = (float)y; verts[3 * k + 2] = z + t; if (keys) keys[k] = 3 * i + 2; }
}

__global__ __launch_bounds__(kMcBlock) void k_mc_faces(int res, int64_t npts, const uint8_t *__restrict__ flags,
                                                       const uint8_t *__restrict__ ntri, const int32_t *__restrict__ active,
                                                       const int32_t *__restrict__ first_vertex,
                                                       const int32_t *__restrict__ first_tri, int64_t *__restrict__ faces)
{
    const int n = res - 1;
    const int64_t i = (int64_t)active[blockIdx.x] * kMcBlock + threadIdx.x;
    if (i >= npts) return;
    const int nt = ntri[i];
    if (nt == 0) return;
    int x, y, z; mc_cell(i, n, x, y, z);
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t j = ((int64_t)(z + ((k >> 2) & 1)) * n + (y + ((k >> 1) & 1))) * n + (x + (k & 1));
        c |= (flags[j] & 8) ? (1 << k) : 0;
    }
    int64_t *out = faces + 3 * (int64_t)first_tri[i];
    for (int t = 0; t < nt; ++t)
        for (int q = 0; q < 3; ++q) {
            const int e = c_mc.tri[c][1 + 3 * t + q];
            const int a = c_mc.edge[e][0], dir = a ^ c_mc.edge[e][1];       // 1: x edge, 2: y edge, 4: z edge
            const int64_t j = ((int64_t)(z + ((a >> 2) & 1)) * n + (y + ((a >> 1) & 1))) * n + (x + (a & 1));
            const int fl = flags[j];
            const int rank = (dir == 1) ? 0 : (dir == 2) ? (fl & 1) : ((fl & 1) + ((fl >> 1) & 1));
            out[3 * t + q] = (int64_t)first_vertex[j] + rank;
        }
}

}  // namespace

struct McDevState {
    uint8_t *flags = nullptr, *ntri = nullptr;
    unsigned long long *block_sums = nullptr, *block_tot = nullptr, *totals = nullptr, *chunk_tot = nullptr;
    int32_t *active = nullptr; int64_t n_active = 0;      // blocks holding a vertex or a triangle
    int32_t *first_vertex = nullptr, *first_tri = nullptr;
    int64_t cap = 0;
    const float *occ = nullptr; int res = 0; float level = 0.f; bool counted = false;
};

static void mc_free(McDevState *s)
{
    (void)hipFree(s->flags); (void)hipFree(s->ntri); (void)hipFree(s->block_sums); (void)hipFree(s->block_tot); (void)hipFree(s->totals); (void)hipFree(s->chunk_tot); (void)hipFree(s->active);
    (void)hipFree(s->first_vertex); (void)hipFree(s->first_tri);
    *s = McDevState();
}

}  // namespace icon

using namespace icon;

extern "C" int icon_mc_count(const float *d_occ, int res, float level, icon_work_t *work, void *stream,
                             int64_t *n_verts, int64_t *n_faces)
{
    return icon_mc_count_range(d_occ, res, level, 0, res - 1, 0, work, stream, n_verts, n_faces);
}

extern "C" int icon_mc_count_range(const float *d_occ, int res, float level, int zc0, int zc1, int halo, icon_work_t *work, void *stream,
                                   int64_t *n_verts, int64_t *n_faces)
{
    ICON_ARG(d_occ && work && n_verts && n_faces, "icon_mc_count: null argument");
    ICON_ARG(res >= 3 && res <= 1291, "icon_mc_count: res out of range");
    ICON_ARG(zc0 >= 0 && zc0 <= zc1 && zc1 <= res - 1 && (!halo || zc1 < res - 1), "icon_mc_count_range: bad layer range");
    std::call_once(g_mc_once, [] {
        McTab h;
        mc_tables(h.tri, h.edge);
        g_mc_err = hipMemcpyToSymbol(HIP_SYMBOL(c_mc), &h, sizeof(h));
    });
    if (g_mc_err != hipSuccess) return fail(ICON_ERR_HIP, std::string("marching-cubes tables: ") + hipGetErrorString(g_mc_err));
    hipStream_t st = (hipStream_t)stream;
    if (!work->mc) work->mc = new McDevState();
    McDevState *s = work->mc;
    const int n = res - 1;
    const int64_t npts = (int64_t)n * n * n;
    const int64_t nblk = (npts + kMcBlock - 1) / kMcBlock;
    if (npts > s->cap) {
        mc_free(s);
        ICON_HIP(hipMalloc((void **)&s->flags, (size_t)npts));
        ICON_HIP(hipMalloc((void **)&s->ntri, (size_t)npts));
        ICON_HIP(hipMalloc((void **)&s->block_sums, (size_t)nblk * sizeof(unsigned long long)));
        ICON_HIP(hipMalloc((void **)&s->block_tot, (size_t)nblk * sizeof(unsigned long long)));
        ICON_HIP(hipMalloc((void **)&s->totals, 4 * sizeof(unsigned long long)));
        ICON_HIP(hipMalloc((void **)&s->chunk_tot, (size_t)kMcMaxChunks * sizeof(unsigned long long)));
        ICON_HIP(hipMalloc((void **)&s->active, (size_t)nblk * sizeof(int32_t)));
        ICON_HIP(hipMalloc((void **)&s->first_vertex, (size_t)npts * sizeof(int32_t)));
        ICON_HIP(hipMalloc((void **)&s->first_tri, (size_t)npts * sizeof(int32_t)));
        s->cap = npts;
    }
    static const bool one_cell = getenv("ICON_AMD_MC_CLASSIFY1") != nullptr;         // A/B: the one-cell-per-thread form
    if ((res - 1) % 4 == 0 && !one_cell)
        hipLaunchKernelGGL(k_mc_classify4, dim3((unsigned)nblk), dim3(kMcBlock / 4), 0, st, d_occ, res, level, npts, s->flags, s->ntri, s->block_tot, zc0, zc1, halo);
    else
        hipLaunchKernelGGL(k_mc_classify, dim3((unsigned)nblk), dim3(kMcBlock), 0, st, d_occ, res, level, npts, s->flags, s->ntri, s->block_tot, zc0, zc1, halo);
    const int64_t nchunks = (nblk + kMcChunk - 1) / kMcChunk;
    if (nchunks >= 2 && nchunks <= kMcMaxChunks) {
        hipLaunchKernelGGL(k_mc_scan_local, dim3((unsigned)nchunks), dim3(1024), 0, st, s->block_tot, s->block_sums, nblk, s->chunk_tot, s->totals);
        hipLaunchKernelGGL(k_mc_scan_apply, dim3((unsigned)nchunks), dim3(1024), 0, st, s->block_tot, s->block_sums, nblk, s->chunk_tot, (int)nchunks, s->totals, s->active);
    } else {
        hipLaunchKernelGGL(k_mc_scanblocks, dim3(1), dim3(1024), 0, st, s->block_tot, s->block_sums, nblk, s->totals, s->active);
    }
    ICON_HIP(hipGetLastError());
    unsigned long long h[3];
    ICON_HIP(hipMemcpyAsync(h, s->totals, sizeof(h), hipMemcpyDeviceToHost, st));
    ICON_HIP(hipStreamSynchronize(st));
    *n_verts = (int64_t)h[0]; *n_faces = (int64_t)h[1]; s->n_active = (int64_t)h[2];
    s->occ = d_occ; s->res = res; s->level = level; s->counted = true;
    return ICON_OK;
}

extern "C" int icon_mc_emit(float *d_verts, int64_t *d_faces, icon_work_t *work, void *stream)
{
    return icon_mc_emit_keyed(d_verts, d_faces, nullptr, work, stream);
}

extern "C" int icon_mc_emit_keyed(float *d_verts, int64_t *d_faces, int64_t *d_keys, icon_work_t *work, void *stream)
{
    ICON_ARG(work && work->mc && work->mc->counted, "icon_mc_emit: call icon_mc_count first");
    ICON_ARG(d_verts && d_faces, "icon_mc_emit: null output");
    McDevState *s = work->mc;
    hipStream_t st = (hipStream_t)stream;
    const int n = s->res - 1;
    const int64_t npts = (int64_t)n * n * n;
    if (s->n_active > 0) {
        hipLaunchKernelGGL(k_mc_vertices, dim3((unsigned)s->n_active), dim3(kMcBlock), 0, st, s->occ, s->res, s->level, npts, s->flags, s->ntri,
                           s->active, s->block_sums, s->first_vertex, s->first_tri, d_verts, d_keys);
        hipLaunchKernelGGL(k_mc_faces, dim3((unsigned)s->n_active), dim3(kMcBlock), 0, st, s->res, npts, s->flags, s->ntri, s->active, s->first_vertex,
                           s->first_tri, d_faces);
    }
    ICON_HIP(hipGetLastError());
    s->counted = false;
    return ICON_OK;
}

namespace icon {
void mc_destroy(McDevState *s) { if (s) { mc_free(s); delete s; } }
}
