// micro-probe: SUSTAINED rate of f16 vs i8 MFMA under the chip's power management, with operands that toggle
// (eight rotating operand sets of random bits) or do not (zeros).  Question behind it: would an
// integer-sliced (Ozaki) product - six i8 MFMAs instead of three f16 ones - run cooler?  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

typedef short bf8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: f16 random, 1: f16 zeros, 2: i8 random, 3: i8 zeros, 4: bf16 random, 5: fp8 (f8f6f4 32x32x64) random,
                      // 6: f16 random with the low 5 mantissa bits of ONE operand cleared (cheaper partial products?)
__global__ __launch_bounds__(512) void k(float *out, int iters, int seed)
{
    unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + seed;
    i32x4 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int t = 0; t < 4; ++t) {
            unsigned ra = rnd(s), rb = rnd(s);
            if (MODE < 2 || MODE == 6) {  // two halves per dword: keep exponents in a sane range (|x| in [0.5, 2)), random sign + mantissa
                ra = (ra & 0x83ff83ffu) | 0x38003800u; rb = (rb & 0x83ff83ffu) | 0x38003800u;
                if (MODE == 6) rb &= 0xffe0ffe0u;
            }
            if (MODE == 4) {              // bf16: sign + 7 mantissa bits random, exponent 126/127
                ra = (ra & 0x807f807fu) | 0x3f003f00u; rb = (rb & 0x807f807fu) | 0x3f003f00u;
            }
            if (MODE == 5) {              // fp8 e4m3 bytes: sign + 3 mantissa bits random, exponent 7
                ra = (ra & 0x87878787u) | 0x38383838u; rb = (rb & 0x87878787u) | 0x38383838u;
            }
            if (MODE == 1 || MODE == 3) { ra = 0; rb = 0; }
            a[i][t] = (int)ra; b[i][t] = (int)rb;
        }
    f32x16 facc[8]; i32x16 iacc[8];
    for (int i = 0; i < 8; ++i) for (int t = 0; t < 16; ++t) { facc[i][t] = 0.f; iacc[i][t] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE < 2 || MODE == 6) facc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[i]), __builtin_bit_cast(half8, b[(i + 3) & 7]), facc[i], 0, 0, 0);
            else if (MODE == 4) facc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[i]), __builtin_bit_cast(bf8, b[(i + 3) & 7]), facc[i], 0, 0, 0);
            else if (MODE == 5) {
                i32x8 av, bv;
                for (int t = 0; t < 4; ++t) { av[t] = a[i][t]; av[t + 4] = a[(i + 1) & 7][t]; bv[t] = b[(i + 3) & 7][t]; bv[t + 4] = b[(i + 4) & 7][t]; }
                facc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, facc[i], 0 /* fp8 */, 0 /* fp8 */, 0, 127, 0, 127);
            }
            else iacc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[(i + 3) & 7], iacc[i], 0, 0, 0);
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) for (int t = 0; t < 16; ++t) r += facc[i][t] + (float)iacc[i][t];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, double mac_per_inst)
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    k<MODE><<<256, 512>>>(d, 2000, 1); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 600000;             // 130-200 ms: long enough for the power management to settle
    hipEventRecord(e0); k<MODE><<<256, 512>>>(d, iters, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 256.0 * 8 /*waves*/ * iters * 8;
    // issue peak: 1024 SIMDs x 2.4 GHz / 32 cycles per 32x32 MFMA = 76.8 Ginst/s
    printf("%-12s %8.2f ms  %7.1f Ginst/s  %7.1f TMAC/s  (%.0f %% of the 2.4 GHz issue peak)\n", name, ms, insts / ms / 1e6,
           insts * mac_per_inst / ms / 1e9, 100.0 * (insts / ms / 1e6) / 76.8);
    hipFree(d);
}
int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<1>("f16 zeros", 32.0 * 32 * 16);
        run<0>("f16 random", 32.0 * 32 * 16);
        run<3>("i8 zeros", 32.0 * 32 * 32);
        run<2>("i8 random", 32.0 * 32 * 32);
        run<4>("bf16 random", 32.0 * 32 * 16);
        run<6>("f16 rnd, b 6b", 32.0 * 32 * 16);
        run<5>("fp8 random", 32.0 * 32 * 64);
    }
    return 0;
}
