// micro-probe: issue rate of the gfx950 MFMA flavours a split-precision MLP could use
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int seed)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int t = 0; t < 16; ++t) acc[i][t] = 0.f;
    half8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * ((threadIdx.x * 7 + i + seed) % 97)); b[i] = (_Float16)(0.002f * ((threadIdx.x * 3 + i) % 89)); }
    i32x8 ia, ib; for (int i = 0; i < 8; ++i) { ia[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i + seed); ib[i] = 0x34343434 ^ (threadIdx.x * 0x00010203 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, acc[i], 0, 0, 0, 127, 0, 127);   // fp8 e4m3
            if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, acc[i], 2, 2, 0, 127, 0, 127);   // fp6 e2m3
            if (MODE == 3) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, acc[i], 4, 4, 0, 127, 0, 127);   // fp4
        }
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int t = 0; t < 16; ++t) s += acc[i][t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, double flop_per_inst)
{
    float *d; hipMalloc(&d, 2048 * 256 * 4);
    const int iters = 4000;
    k<MODE><<<2048, 256>>>(d, 100, 1); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<2048, 256>>>(d, iters, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 2048.0 * 4 /*waves*/ * iters * 4;
    printf("%-10s %8.3f ms  %8.1f TFLOP/s  (%.1f Ginst/s)\n", name, ms, insts * flop_per_inst / ms / 1e9, insts / ms / 1e6);
    hipFree(d);
}
int main()
{
    run<0>("f16 K16", 2.0 * 32 * 32 * 16);
    run<1>("fp8 K64", 2.0 * 32 * 32 * 64);
    run<2>("fp6 K64", 2.0 * 32 * 32 * 64);
    run<3>("fp4 K64", 2.0 * 32 * 32 * 64);
    return 0;
}
