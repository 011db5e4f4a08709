// probe: do VALU instructions hide behind a back-to-back MFMA stream of the same wave (1 or 2 waves per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NV, int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = seed * (t + e);
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + e); b[e] = (_Float16)(seed - e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed + threadIdx.x + e;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[e]) : "v"(seed));
                if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[e]) : "v"(seed));
                if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[e]) : "v"(seed));
                if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(v[e]) : "v"(seed));
                if (KIND == 4 && e == 0) {
                    typedef _Float16 h32 __attribute__((ext_vector_type(32)));
                    typedef unsigned u6 __attribute__((ext_vector_type(6)));
                    h32 hv; for (int q = 0; q < 32; ++q) hv[q] = (_Float16)v[q & 7];
                    asm volatile("" : "+v"(hv));
                    u6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, seed);
                    asm volatile("" : "+v"(r));
                    v[0] += __uint_as_float(r[0] & 1);
                }
                if (KIND == 5 && (e & 3) == 0) {      // one value pair of the mx6 kernel: 8 VALU, dependent
                    float x0 = v[e], x1 = v[e + 1];
                    asm volatile("v_mul_f32 %0, 0x3c23d70a, %2\n\tv_mul_f32 %1, 0x3c23d70a, %3\n\tv_max_f32 %2, %2, %0\n\tv_max_f32 %3, %3, %1\n\t"
                                 "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_max3_f32 %2, %2, |%3|, |%3|"
                                 : "=&v"(v[e + 2]), "=&v"(v[e + 3]), "+v"(x0), "+v"(x1));
                    v[e] = x0; v[e + 1] = x1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int KIND> void run(float *d, int threads)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    k<NV, KIND><<<blocks, threads>>>(d, 100, 1.0f);
    hipEventRecord(e0);
    k<NV, KIND><<<blocks, threads>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: (threads/256) waves, each 4*iters MFMAs
    const double mfma_per_simd = (threads / 256.0) * 4.0 * iters;
    printf("waves/SIMD %d  kind %d  VALU per MFMA %d: %.1f ns per MFMA slot (ideal 32 cycles = %.1f ns at 2.0 GHz)\n", threads / 256, KIND, NV,
           ms * 1e6 / mfma_per_simd, 16.0);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 0>(d, 256); run<2, 0>(d, 256); run<4, 0>(d, 256); run<6, 0>(d, 256); run<8, 0>(d, 256);
    run<0, 0>(d, 512); run<2, 0>(d, 512); run<4, 0>(d, 512); run<6, 0>(d, 512); run<8, 0>(d, 512);
    run<4, 1>(d, 512); run<4, 2>(d, 512); run<4, 3>(d, 512); run<8, 3>(d, 512); run<1, 4>(d, 512); run<4, 5>(d, 512); run<8, 5>(d, 512); run<4, 5>(d, 256);
    return 0;
}
