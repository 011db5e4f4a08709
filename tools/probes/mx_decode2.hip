// probe: A = one code at (g, field), B = all 1.0 -> D[0][0] = hardware's reading of A
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ i32x8 one(int field, unsigned code, bool on)
{
    i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!on) return v;
    const unsigned long long bits = code;
    const int bit = 6 * field, w = bit / 32, sh = bit % 32;
    v[w] = (int)(unsigned)(bits << sh); if (sh > 26 && w + 1 < 6) v[w + 1] = (int)(unsigned)(bits >> (32 - sh));
    return v;
}
__global__ void k(float *out)
{
    const int lane = threadIdx.x, g = lane >> 5;
    i32x8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
    // 0x08 repeated every 6 bits: 192 bits
    unsigned long long acc[3] = {0, 0, 0};
    for (int f = 0; f < 32; ++f) { const int bit = 6 * f; acc[bit / 64] |= 0x08ull << (bit % 64); if (bit % 64 > 58) acc[bit / 64 + 1] |= 0x08ull >> (64 - bit % 64); }
    for (int i = 0; i < 3; ++i) { ones[2 * i] = (int)(unsigned)acc[i]; ones[2 * i + 1] = (int)(unsigned)(acc[i] >> 32); }
    for (int gg = 0; gg < 2; ++gg)
        for (int f = 0; f < 32; ++f)
            for (int ci = 0; ci < 3; ++ci) {
                const unsigned code = ci == 0 ? 0x08 : ci == 1 ? 0x01 : 0x2c;
                f32x16 c; for (int t = 0; t < 16; ++t) c[t] = 0.f;
                c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(one(f, code, g == gg), ones, c, 2, 2, 0, 127, 0, 127);
                if (lane == 0) out[(gg * 32 + f) * 3 + ci] = c[0];
            }
    f32x16 c; for (int t = 0; t < 16; ++t) c[t] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, ones, c, 2, 2, 0, 127, 0, 127);
    if (lane == 0) out[192] = c[0];
}
int main()
{
    float *d, h[193]; hipMalloc(&d, sizeof(h)); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("g%d f%2d: 1.0->%g  0.125->%g  -1.5->%g\n", i >> 5, i & 31, h[3 * i], h[3 * i + 1], h[3 * i + 2]);
    printf("ones.ones = %g (expect 64)\n", h[192]);
    return 0;
}
