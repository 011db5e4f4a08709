// sort_points.hip - Morton order of arbitrary query points (point mode of HGPIFuNet.query).
//
// The packet traversal of k_nearest lets the 64 lanes of a wavefront walk the BVH together, which
// pays when the 64 points are neighbours (lattice mode hands every wave a 4x4x4 block).  query()
// gets points in any order - Seg3dLossless's coarse-to-fine batches are x-fastest lists spread over
// the whole cube - so the search runs over a Morton-sorted permutation and writes its result back
// to the point's own slot; everything order-dependent (the reference's outlier list, the output
// layout) still sees the caller's order.  30-bit keys, rocPRIM radix sort of (key, index) pairs.
#include <cstring>

#include "common.h"

#include <rocprim/rocprim.hpp>

namespace icon {

__device__ __forceinline__ uint32_t spread10(uint32_t v)      // 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

struct Calib12 { float m[12]; const float *d; };

__global__ void k_morton_keys(const float *__restrict__ pts, Calib12 c, int64_t N, uint32_t *__restrict__ keys, int32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (c.d)
        for (int k = 0; k < 12; ++k) c.m[k] = c.d[k];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float v = c.m[4 * r] * x + c.m[4 * r + 1] * y + c.m[4 * r + 2] * z + c.m[4 * r + 3];
        q[r] = fminf(fmaxf((v + 1.25f) * (1024.0f / 2.5f), 0.0f), 1023.0f);     // ordering only: any monotone map will do
    }
    keys[i] = spread10((uint32_t)q[0]) | (spread10((uint32_t)q[1]) << 1) | (spread10((uint32_t)q[2]) << 2);
    idx[i] = (int32_t)i;
}

int morton_order(icon_work *w, const float *d_points, const float *calib12, const float *d_calib12, int64_t N, hipStream_t st, const int32_t **perm)
{
    ICON_ARG(N > 0 && N < (1ll << 31), "morton_order: bad point count");
    size_t tmp = 0;
    ICON_HIP(rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                                       (size_t)N, 0, 30, st));
    if (N > w->cap_sort) {
        (void)hipFree(w->d_sort_keys); (void)hipFree(w->d_sort_idx);
        w->d_sort_keys = nullptr; w->d_sort_idx = nullptr; w->cap_sort = 0;
        ICON_HIP(hipMalloc((void **)&w->d_sort_keys, 2 * (size_t)N * sizeof(uint32_t)));
        ICON_HIP(hipMalloc((void **)&w->d_sort_idx, 2 * (size_t)N * sizeof(int32_t)));
        w->cap_sort = N;
    }
    if (tmp > w->sort_tmp_bytes) {
        (void)hipFree(w->d_sort_tmp); w->d_sort_tmp = nullptr; w->sort_tmp_bytes = 0;
        ICON_HIP(hipMalloc(&w->d_sort_tmp, tmp));
        w->sort_tmp_bytes = tmp;
    }
    Calib12 c;
    memcpy(c.m, calib12, sizeof(c.m));
    c.d = d_calib12;
    uint32_t *k0 = w->d_sort_keys, *k1 = w->d_sort_keys + w->cap_sort;
    int32_t *i0 = w->d_sort_idx, *i1 = w->d_sort_idx + w->cap_sort;
    hipLaunchKernelGGL(k_morton_keys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_points, c, N, k0, i0);
    ICON_HIP(hipGetLastError());
    size_t bytes = w->sort_tmp_bytes;
    ICON_HIP(rocprim::radix_sort_pairs(w->d_sort_tmp, bytes, k0, k1, i0, i1, (size_t)N, 0, 30, st));
    *perm = i1;
    return ICON_OK;
}

}  // namespace icon
