// mesh_cc.hip - connected components of a triangle mesh on the device: the engine behind clean_mesh
// (lib/dataset/mesh_util.py:778-791: trimesh split -> keep the component with the most vertices,
// called at apps/ICON.py:755-756 on the marching-cubes output).
//
// Lock-free union-find over the vertices: every face unites its three corners (atomicCAS hooks the larger
// root under the smaller one, path halving on the way), a second pass flattens; the label of a vertex is
// the smallest vertex index of its component.  A 257^3 body surface (~70k vertices / 140k faces) takes a
// few tens of microseconds; the selection / compaction of the winning component is plumbing on the host
// layer (icon_amd/recon.py: clean_mesh).
#include "common.h"

namespace icon {

__device__ __forceinline__ int cc_find(int *parent, int x)
{
    while (true) {
        const int p = parent[x];
        if (p == x) return x;
        const int gp = parent[p];
        if (gp != p) parent[x] = gp;       // path halving: benign race, always an ancestor
        x = p;
    }
}

__device__ __forceinline__ void cc_unite(int *parent, int u, int v)
{
    while (true) {
        u = cc_find(parent, u); v = cc_find(parent, v);
        if (u == v) return;
        const int hi = max(u, v), lo = min(u, v);
        if (atomicCAS(&parent[hi], hi, lo) == hi) return;
    }
}

__global__ void k_cc_init(int *parent, int64_t V)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) parent[i] = (int)i;
}

__global__ void k_cc_union(const int64_t *__restrict__ faces, int64_t F, int64_t V, int *parent, int *bad)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int64_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
    if (a < 0 || b < 0 || c < 0 || a >= V || b >= V || c >= V) { *bad = 1; return; }
    cc_unite(parent, (int)a, (int)b);
    cc_unite(parent, (int)a, (int)c);
}

// read-only walk to the root, result out of place: a flatten that compresses in place can have its final
// store parent[i] = root overtaken by another thread's path-halving store parent[i] = grandparent
__global__ void k_cc_flatten(const int *__restrict__ parent, int *__restrict__ labels, int64_t V)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    int x = (int)i;
    while (true) { const int p = parent[x]; if (p == x) break; x = p; }
    labels[i] = x;
}

}  // namespace icon

using namespace icon;

extern "C" int icon_mesh_components(const int64_t *d_faces, int64_t F, int64_t V, int32_t *d_labels, void *stream)
{
    ICON_ARG(d_labels != nullptr && V >= 0 && F >= 0, "icon_mesh_components: bad argument");
    ICON_ARG(V < (1ll << 31), "icon_mesh_components: more than 2^31 vertices");
    ICON_ARG(F == 0 || d_faces != nullptr, "icon_mesh_components: faces are null");
    if (V == 0) return ICON_OK;
    hipStream_t st = (hipStream_t)stream;
    int *d_bad = nullptr;
    ICON_HIP(hipMalloc((void **)&d_bad, sizeof(int)));
    ICON_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), st));
    int *d_parent = nullptr;
    hipError_t ea = hipMalloc((void **)&d_parent, (size_t)V * sizeof(int));
    if (ea != hipSuccess) { (void)hipFree(d_bad); return fail(ICON_ERR_HIP, std::string("icon_mesh_components: ") + hipGetErrorString(ea)); }
    hipLaunchKernelGGL(k_cc_init, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, d_parent, V);
    if (F > 0) hipLaunchKernelGGL(k_cc_union, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, d_faces, F, V, d_parent, d_bad);
    hipLaunchKernelGGL(k_cc_flatten, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, d_parent, d_labels, V);
    int bad = 0;
    hipError_t e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_bad); (void)hipFree(d_parent);
    if (e != hipSuccess) return fail(ICON_ERR_HIP, std::string("icon_mesh_components: ") + hipGetErrorString(e));
    if (bad) return fail(ICON_ERR_ARG, "icon_mesh_components: face index out of range");
    return ICON_OK;
}
