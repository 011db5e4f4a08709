// probe: two waves per SIMD alternating an MFMA phase (54 MFMAs) and a VALU phase (NV plain VALU ops),
// skewed (waves 0-3 MFMA first, waves 4-7 VALU first) or not, with a workgroup barrier per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NV, int MODE>   // MODE 0: all waves M then V; 1: waves>=4 V first; 2: interleaved (8 VALU after every 3rd MFMA... NV/18 per MFMA)
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed)
{
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = seed * (t + e);
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + e); b[e] = (_Float16)(seed - e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed + threadIdx.x + e;
    const bool vfirst = MODE == 1 && (threadIdx.x >> 8);
    auto mphase = [&]() {
#pragma unroll
        for (int i = 0; i < 54; ++i) {
            acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 7], 0, 0, 0);
            if (MODE == 2) {
#pragma unroll
                for (int e = 0; e < (NV + 53) / 54; ++e) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[e & 7]) : "v"(seed));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto vphase = [&]() {
        if (MODE == 2) return;
#pragma unroll
        for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[e & 7]) : "v"(seed));
    };
    for (int it = 0; it < iters; ++it) {
        if (vfirst) { vphase(); mphase(); } else { mphase(); vphase(); }
        __syncthreads();
    }
    float s = 0;
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int MODE> void run(float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    k<NV, MODE><<<256, 512>>>(d, 50, 1.0f);
    hipEventRecord(e0);
    k<NV, MODE><<<256, 512>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  VALU per phase %3d: %.0f ns per iteration (108 MFMAs per SIMD = %.0f ns at 16 ns each)\n", MODE, NV, ms * 1e6 / iters, 108 * 16.0);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 0>(d); run<136, 0>(d); run<136, 1>(d); run<136, 2>(d); run<272, 0>(d); run<272, 1>(d); run<272, 2>(d);
    return 0;
}
