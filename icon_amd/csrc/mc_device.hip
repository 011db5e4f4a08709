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

__device__ __forceinline__ float mc_lerp(float level, float v0, float v1)
{
    const float d = v1 - v0;
    const float t = (d != 0.0f) ? (level - v0) / d : 0.5f;
    return fminf(fmaxf(t, 0.0f), 1.0f);
}

// flags: bit0/1/2 = +x/+y/+z edge crossed, bit3 = inside
__global__ __launch_bounds__(kMcBlock) void k_mc_classify(const float *__restrict__ occ, int res, float level, int64_t npts,
                                                          uint8_t *__restrict__ flags, uint8_t *__restrict__ ntri)
{
    const int n = res - 1;
    const int64_t i = (int64_t)blockIdx.x * kMcBlock + threadIdx.x;
    if (i >= npts) return;
    const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((int64_t)n * n));
    bool in[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int xx = min(x + (k & 1), n - 1), yy = min(y + ((k >> 1) & 1), n - 1), zz = min(z + ((k >> 2) & 1), n - 1);
        in[k] = mc_at(occ, res, zz, yy, xx) > level;
    }
    uint8_t f = in[0] ? 8 : 0;
    if (x + 1 < n && in[1] != in[0]) f |= 1;
    if (y + 1 < n && in[2] != in[0]) f |= 2;
    if (z + 1 < n && in[4] != in[0]) f |= 4;
    flags[i] = f;
    uint8_t t = 0;
    if (x + 1 < n && y + 1 < n && z + 1 < n) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c |= in[k] ? (1 << k) : 0;
        t = (uint8_t)c_mc.tri[c][0];
    }
    ntri[i] = t;
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

__global__ __launch_bounds__(kMcBlock) void k_mc_blocksum(const uint8_t *__restrict__ flags, const uint8_t *__restrict__ ntri,
                                                          int64_t npts, unsigned long long *block_sums)
{
    __shared__ unsigned long long wsum[kMcBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kMcBlock + threadIdx.x;
    unsigned long long total;
    (void)mc_block_exscan(mc_count(flags, ntri, i, npts), wsum, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_mc_scanblocks(unsigned long long *block_sums, int64_t nblocks, unsigned long long *totals)
{
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int64_t per = (nblocks + 1023) / 1024;
    const int64_t beg = min((int64_t)t * per, nblocks), end = min(beg + per, nblocks);
    unsigned long long s = 0;
    for (int64_t k = beg; k < end; ++k) s += block_sums[k];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int k = 0; k < 1024; ++k) { const unsigned long long v = part[k]; part[k] = run; run += v; }
        totals[0] = run & 0xffffffffull; totals[1] = run >> 32;
    }
    __syncthreads();
    unsigned long long run = part[t];
    for (int64_t k = beg; k < end; ++k) { const unsigned long long v = block_sums[k]; block_sums[k] = run; run += v; }
}

__global__ __launch_bounds__(kMcBlock) void k_mc_vertices(const float *__restrict__ occ, int res, float level, int64_t npts,
                                                          const uint8_t *__restrict__ flags, const uint8_t *__restrict__ ntri,
                                                          const unsigned long long *__restrict__ block_offsets,
                                                          int32_t *__restrict__ first_vertex, int32_t *__restrict__ first_tri,
                                                          float *__restrict__ verts)
{
    __shared__ unsigned long long wsum[kMcBlock / 64];
    const int n = res - 1;
    const int64_t i = (int64_t)blockIdx.x * kMcBlock + threadIdx.x;
    const unsigned long long ex = mc_block_exscan(mc_count(flags, ntri, i, npts), wsum, nullptr) + block_offsets[blockIdx.x];
    if (i >= npts) return;
    const int32_t v0 = (int32_t)(ex & 0xffffffffull);
    first_vertex[i] = v0;
    first_tri[i] = (int32_t)(ex >> 32);
    const uint8_t f = flags[i];
    if (!(f & 7)) return;
    const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((int64_t)n * n));
    const float v = mc_at(occ, res, z, y, x);
    int k = v0;
    if (f & 1) { const float t = mc_lerp(level, v, mc_at(occ, res, z, y, x + 1)); verts[3 * k] = x + t; verts[3 * k + 1] = (float)y; verts[3 * k + 2] = (float)z; ++k; }
    if (f & 2) { const float t = mc_lerp(level, v, mc_at(occ, res, z, y + 1, x)); verts[3 * k] = (float)x; verts[3 * k + 1] = y + t; verts[3 * k + 2] = (float)z; ++k; }
    if (f & 4) { const float t = mc_lerp(level, v, mc_at(occ, res, z + 1, y, x)); verts[3 * k] = (float)x; verts[3 * k + 1] = (float)y; verts[3 * k + 2] = z + t; }
}

__global__ __launch_bounds__(kMcBlock) void k_mc_faces(int res, int64_t npts, const uint8_t *__restrict__ flags,
                                                       const uint8_t *__restrict__ ntri, const int32_t *__restrict__ first_vertex,
                                                       const int32_t *__restrict__ first_tri, int64_t *__restrict__ faces)
{
    const int n = res - 1;
    const int64_t i = (int64_t)blockIdx.x * kMcBlock + threadIdx.x;
    if (i >= npts) return;
    const int nt = ntri[i];
    if (nt == 0) return;
    const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((int64_t)n * n));
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
    unsigned long long *block_sums = nullptr, *totals = nullptr;
    int32_t *first_vertex = nullptr, *first_tri = nullptr;
    int64_t cap = 0;
    const float *occ = nullptr; int res = 0; float level = 0.f; bool counted = false;
};

static void mc_free(McDevState *s)
{
    (void)hipFree(s->flags); (void)hipFree(s->ntri); (void)hipFree(s->block_sums); (void)hipFree(s->totals);
    (void)hipFree(s->first_vertex); (void)hipFree(s->first_tri);
    *s = McDevState();
}

}  // namespace icon

using namespace icon;

extern "C" int icon_mc_count(const float *d_occ, int res, float level, icon_work_t *work, void *stream,
                             int64_t *n_verts, int64_t *n_faces)
{
    ICON_ARG(d_occ && work && n_verts && n_faces, "icon_mc_count: null argument");
    ICON_ARG(res >= 3 && res <= 1291, "icon_mc_count: res out of range");
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
        ICON_HIP(hipMalloc((void **)&s->totals, 2 * sizeof(unsigned long long)));
        ICON_HIP(hipMalloc((void **)&s->first_vertex, (size_t)npts * sizeof(int32_t)));
        ICON_HIP(hipMalloc((void **)&s->first_tri, (size_t)npts * sizeof(int32_t)));
        s->cap = npts;
    }
    hipLaunchKernelGGL(k_mc_classify, dim3((unsigned)nblk), dim3(kMcBlock), 0, st, d_occ, res, level, npts, s->flags, s->ntri);
    hipLaunchKernelGGL(k_mc_blocksum, dim3((unsigned)nblk), dim3(kMcBlock), 0, st, s->flags, s->ntri, npts, s->block_sums);
    hipLaunchKernelGGL(k_mc_scanblocks, dim3(1), dim3(1024), 0, st, s->block_sums, nblk, s->totals);
    ICON_HIP(hipGetLastError());
    unsigned long long h[2];
    ICON_HIP(hipMemcpyAsync(h, s->totals, sizeof(h), hipMemcpyDeviceToHost, st));
    ICON_HIP(hipStreamSynchronize(st));
    *n_verts = (int64_t)h[0]; *n_faces = (int64_t)h[1];
    s->occ = d_occ; s->res = res; s->level = level; s->counted = true;
    return ICON_OK;
}

extern "C" int icon_mc_emit(float *d_verts, int64_t *d_faces, icon_work_t *work, void *stream)
{
    ICON_ARG(work && work->mc && work->mc->counted, "icon_mc_emit: call icon_mc_count first");
    ICON_ARG(d_verts && d_faces, "icon_mc_emit: null output");
    McDevState *s = work->mc;
    hipStream_t st = (hipStream_t)stream;
    const int n = s->res - 1;
    const int64_t npts = (int64_t)n * n * n;
    const int64_t nblk = (npts + kMcBlock - 1) / kMcBlock;
    hipLaunchKernelGGL(k_mc_vertices, dim3((unsigned)nblk), dim3(kMcBlock), 0, st, s->occ, s->res, s->level, npts, s->flags, s->ntri,
                       s->block_sums, s->first_vertex, s->first_tri, d_verts);
    hipLaunchKernelGGL(k_mc_faces, dim3((unsigned)nblk), dim3(kMcBlock), 0, st, s->res, npts, s->flags, s->ntri, s->first_vertex,
                       s->first_tri, d_faces);
    ICON_HIP(hipGetLastError());
    s->counted = false;
    return ICON_OK;
}

namespace icon {
void mc_destroy(McDevState *s) { if (s) { mc_free(s); delete s; } }
}
