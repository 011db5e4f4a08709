// probe: an MFMA stream whose A operands come from LDS through a rotating 8-deep register window
// (one ds_read_b128 per MFMA, counted lgkmcnt waits) - cycles per MFMA for 1 / 2 waves per SIMD, with
// f16 K=16 MFMAs only or with the fp6 K=64 (b128 + b64 per operand) mix of the mx6 kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: registers only; 1: LDS window, f16 ops; 2: 32 f16 ops + 16 fp6 ops per 48; 3: mode 2 + LDS-DMA of 57 KiB per iteration
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed, const char *img)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 57 * 1024 / 4; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = seed * (i & 7);
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = seed * (t + e);
    half8 b; for (int e = 0; e < 8; ++e) b[e] = (_Float16)(seed - e);
    i32x8 b6 = {1, 2, 3, 4, 5, 6, 0, 0};
    const char *L = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int it = 0; it < iters; ++it) {
        half8 a[8];
        i32x8 w6[4];
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const half8 *>(L + i * 1024 + lane * 16);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(seed + e + i);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const half8 cur = a[i & 7];
            if (MODE >= 1 && i + 8 < 32) a[i & 7] = *reinterpret_cast<const half8 *>(L + (i + 8) * 1024 + lane * 16);
            if (MODE >= 2 && i >= 28) {
                const uint4 p = *reinterpret_cast<const uint4 *>(L + 32768 + (i - 28) * 1024 + lane * 16);
                const uint2 r = *reinterpret_cast<const uint2 *>(L + 40960 + (i - 28) * 512 + lane * 8);
                w6[i - 28] = i32x8{(int)p.x, (int)p.y, (int)p.z, (int)p.w, (int)r.x, (int)r.y, 0, 0};
            }
            acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, b, acc[i & 7], 0, 0, 0);
            if (MODE == 3 && i % 4 == 3) {
                const int u = wave + (blockDim.x >> 6) * (i / 4);
                if (u < 57) {
                    const unsigned l = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(smem + 58368 + u * 1024);
                    const char *g = img + (size_t)((it & 7) * 57 + u) * 1024;
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(l), "v"(lane * 16), "s"(g) : "memory");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE >= 2) {
                const i32x8 cur = w6[j & 3];
                if (j + 4 < 16) {
                    const uint4 p = *reinterpret_cast<const uint4 *>(L + 32768 + ((j + 4) & 7) * 1024 + lane * 16);
                    const uint2 r = *reinterpret_cast<const uint2 *>(L + 40960 + ((j + 4) & 7) * 512 + lane * 8);
                    w6[j & 3] = i32x8{(int)p.x, (int)p.y, (int)p.z, (int)p.w, (int)r.x, (int)r.y, 0, 0};
                }
                acc[j & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cur, b6, acc[j & 7], 2, 2, 0, 127, 0, 127);
            } else {
                acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 7], b, acc[j & 7], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(float *d, int threads)
{
    static char *img = nullptr; if (!img) { hipMalloc(&img, 8 * 57 * 1024); hipMemset(img, 1, 8 * 57 * 1024); }
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
    k<MODE><<<256, threads, 150 * 1024>>>(d, 50, 1.0f, img);
    hipEventRecord(e0);
    k<MODE><<<256, threads, 150 * 1024>>>(d, iters, 1.0f, img);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  waves/SIMD %d: %.2f ns per MFMA slot\n", MODE, threads / 256, ms * 1e6 / iters / (48.0 * threads / 256));
}
int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    run<0>(d, 256); run<0>(d, 512); run<1>(d, 256); run<1>(d, 512); run<2>(d, 256); run<2>(d, 512); run<3>(d, 256); run<3>(d, 512);
    return 0;
}
