// vis_kernels.hip - SMPL vertex visibility (lib/dataset/mesh_util.py:280-316 get_visibility; call
// sites lib/dataset/TestDataset.py:134-137, lib/dataset/PIFuDataset.py:436, lib/common/render.py:76).
//
// The reference rasterises the 13,776 SMPL triangles with pytorch3d at 4096^2 (orthographic, 1 face
// per pixel, back faces culled, "perspective-correct" depth) and marks the vertices of every face
// that owns at least one pixel.  pytorch3d is not vendored (requirements.txt:33): this is a
// restatement of its published per-pixel rule (spelled out next to the CPU checker's orc_visibility;
// DESIGN.md section 4.6), float32 expressions in a fixed order (this file is compiled with
// -ffp-contract=off) so the visible set is compared for equality with the checker.  PARITY UNPINNED
// against pytorch3d itself.
//
// One wavefront per face: its lanes sweep the pixel centres of the face's bounding box and
// atomicMin a (depth bits, face index) key into a 64-bit z-buffer that covers the [0,1]^2 quadrant
// of NDC the reference's (xyz+1)/2 mapping puts the mesh in (S/2 x S/2 pixels, 33.5 MB at 4096).
// A second pass marks the vertices of the winners.  faces[-1] (background pixels exist: three
// quadrants of the image are empty) is always marked, as in the reference.
#include "common.h"

namespace icon {

__device__ __forceinline__ float vis_ef(float px, float py, float ax, float ay, float bx, float by)
{
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
__device__ __forceinline__ float vmax(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float vmin(float a, float b) { return (b < a) ? b : a; }

__global__ __launch_bounds__(64) void k_vis_raster(const float *__restrict__ xy, const float *__restrict__ z,
                                                   const int64_t *__restrict__ faces, int64_t F, int64_t V, int S,
                                                   unsigned long long *__restrict__ zb)
{
    const int64_t f = blockIdx.x;
    if (f >= F) return;
    const int H = S / 2;
    float X[3], Y[3], Z[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t v = faces[3 * f + k];
        if (v < 0 || v >= V) return;                                       // bad caller data: the face does not exist
        X[k] = (xy[2 * v] + 1.0f) / 2.0f; Y[k] = (xy[2 * v + 1] + 1.0f) / 2.0f; Z[k] = (-z[v] + 1.0f) / 2.0f;
    }
    const float eps = 1e-8f;
    const float area = vis_ef(X[2], Y[2], X[0], Y[0], X[1], Y[1]);
    if (area < 0.0f || fabsf(area) <= eps) return;                       // back face / degenerate
    const float xmin = vmin(X[0], vmin(X[1], X[2])), xmax = vmax(X[0], vmax(X[1], X[2]));
    const float ymin = vmin(Y[0], vmin(Y[1], Y[2])), ymax = vmax(Y[0], vmax(Y[1], Y[2]));
    int i0 = (int)floorf((xmin + 1.0f) * 0.5f * (float)S) - 1, i1 = (int)ceilf((xmax + 1.0f) * 0.5f * (float)S) + 1;
    int j0 = (int)floorf((ymin + 1.0f) * 0.5f * (float)S) - 1, j1 = (int)ceilf((ymax + 1.0f) * 0.5f * (float)S) + 1;
    i0 = max(i0, H); j0 = max(j0, H); i1 = min(i1, S - 1); j1 = min(j1, S - 1);
    if (i0 > i1 || j0 > j1) return;
    const int w = i1 - i0 + 1;
    const int64_t n = (int64_t)w * (j1 - j0 + 1);
    const float den = area + eps;
    for (int64_t t = threadIdx.x; t < n; t += 64) {
        const int j = j0 + (int)(t / w), i = i0 + (int)(t % w);
        const float px = -1.0f + (float)(2 * i + 1) / (float)S, py = -1.0f + (float)(2 * j + 1) / (float)S;
        if (px < xmin || px > xmax || py < ymin || py > ymax) continue;
        const float w0 = vis_ef(px, py, X[1], Y[1], X[2], Y[2]) / den;
        const float w1 = vis_ef(px, py, X[2], Y[2], X[0], Y[0]) / den;
        const float w2 = vis_ef(px, py, X[0], Y[0], X[1], Y[1]) / den;
        const float t0 = w0 * Z[1] * Z[2], t1 = Z[0] * w1 * Z[2], t2 = Z[0] * Z[1] * w2;
        const float dn = vmax(t0 + t1 + t2, eps);
        const float b0 = t0 / dn, b1 = t1 / dn, b2 = t2 / dn;
        const float pz = b0 * Z[0] + b1 * Z[1] + b2 * Z[2];
        if (pz < 0.0f) continue;
        if (!(b0 > 0.0f && b1 > 0.0f && b2 > 0.0f)) continue;
        const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned long long)(uint32_t)f;
        atomicMin(&zb[(size_t)(j - H) * H + (i - H)], key);
    }
}

__global__ void k_vis_resolve(const unsigned long long *__restrict__ zb, int64_t npx, const int64_t *__restrict__ faces, int64_t F, int64_t V,
                              float *__restrict__ vis)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && F > 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                        // faces[-1]: the background index
            const int64_t v = faces[3 * (F - 1) + k];
            if (v >= 0 && v < V) vis[v] = 1.0f;
        }
    }
    if (i >= npx) return;
    const unsigned long long key = zb[i];
    if (key == ~0ull) return;
    const int64_t f = (int64_t)(key & 0xffffffffull);
#pragma unroll
    for (int k = 0; k < 3; ++k) vis[faces[3 * f + k]] = 1.0f;             // a face in the z-buffer passed k_vis_raster's index check
}

}  // namespace icon

using namespace icon;

extern "C" int icon_visibility(const float *d_xy, const float *d_z, int64_t V, const int64_t *d_faces, int64_t F, int image_size,
                               float *d_vis, void *stream)
{
    ICON_ARG(d_xy && d_z && d_faces && d_vis, "icon_visibility: null argument");
    ICON_ARG(V > 0 && F > 0 && F < (1ll << 31), "icon_visibility: bad mesh size");
    ICON_ARG(image_size >= 2 && image_size <= 16384 && (image_size & 1) == 0, "icon_visibility: image_size must be even, 2..16384");
    hipStream_t st = (hipStream_t)stream;
    const int H = image_size / 2;
    const size_t npx = (size_t)H * H;
    unsigned long long *zb = nullptr;
    ICON_HIP(hipMalloc((void **)&zb, npx * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(zb, 0xff, npx * sizeof(unsigned long long), st);
    if (e == hipSuccess) e = hipMemsetAsync(d_vis, 0, (size_t)V * sizeof(float), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_vis_raster, dim3((unsigned)F), dim3(64), 0, st, d_xy, d_z, d_faces, F, V, image_size, zb);
        hipLaunchKernelGGL(k_vis_resolve, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, zb, (int64_t)npx, d_faces, F, V, d_vis);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);       // the z-buffer is freed below
    (void)hipFree(zb);
    if (e != hipSuccess) return fail(ICON_ERR_HIP, std::string("icon_visibility: ") + hipGetErrorString(e));
    return ICON_OK;
}
