// probe: byte select (opsel) of the MX scale operands of mfma_scale_f32_32x32x64_f8f6f4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int OA, int OB>
__global__ void k(float *out, float zero, int sa, int sb, const int *pat)
{
    i32x8 ones;
    for (int i = 0; i < 8; ++i) ones[i] = pat[i];
    f32x16 c;
    for (int t = 0; t < 16; ++t) c[t] = zero;
    asm volatile("" : "+v"(c));
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, ones, c, 2, 2, OA, sa, OB, sb);
    for (int t = 0; t < 16; ++t) out[threadIdx.x * 16 + t] = c[t];
}
template <int OA, int OB> void run(float *d, int sa, int sb, const int *pat)
{
    float h[1024];
    k<OA, OB><<<1, 64>>>(d, 0.f, sa, sb, pat); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    float mn = h[0], mx = h[0]; for (int i = 0; i < 1024; ++i) { mn = fminf(mn, h[i]); mx = fmaxf(mx, h[i]); }
    printf("opsel (%d,%d) scales (%#010x, %#010x): D min %g max %g\n", OA, OB, sa, sb, mn, mx);
}
int main()
{
    unsigned long long acc[4] = {0, 0, 0, 0};
    for (int f = 0; f < 32; ++f) { const int bit = 6 * f + 3; acc[bit / 64] |= 1ull << (bit % 64); }
    int hp[8]; for (int i = 0; i < 4; ++i) { hp[2 * i] = (int)(unsigned)acc[i]; hp[2 * i + 1] = (int)(unsigned)(acc[i] >> 32); }
    int *pat; hipMalloc(&pat, 32); hipMemcpy(pat, hp, 32, hipMemcpyHostToDevice);
    float *d; hipMalloc(&d, 4096);
    run<0, 0>(d, 0x7f, 0x7f, pat);
    run<0, 0>(d, 0x7f7f7f7f, 0x7f7f7f7f, pat);
    run<1, 0>(d, 0x7f7f7f7f, 0x7f7f7f7f, pat);
    run<2, 0>(d, 0x7f7f7f7f, 0x7f7f7f7f, pat);
    run<3, 3>(d, 0x7f7f7f7f, 0x7f7f7f7f, pat);
    run<0, 0>(d, 0x8281807f, 0x7f, pat);
    run<1, 0>(d, 0x8281807f, 0x7f, pat);
    run<2, 0>(d, 0x8281807f, 0x7f, pat);
    run<3, 0>(d, 0x8281807f, 0x7f, pat);
    run<0, 1>(d, 0x7f, 0x8281807f, pat);
    run<0, 2>(d, 0x7f, 0x8281807f, pat);
    run<0, 3>(d, 0x7f, 0x8281807f, pat);
    run<3, 3>(d, 0x8281807f, 0x8281807f, pat);
    run<0, 0>(d, 0x8281807f, 0x8281807f, pat);
    return 0;
}
