// probe: value decode of each 6-bit code by the fp6 MFMA (A code X at field 3, B code 0x08 at field 3)
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
    for (int x = 0; x < 64; ++x) {
        f32x16 c; for (int t = 0; t < 16; ++t) c[t] = 0.f;
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(one(3, x, g == 0), one(3, 0x08, g == 0), c, 2, 2, 0, 127, 0, 127);
        if (lane == 0) out[x] = c[0];
        for (int t = 0; t < 16; ++t) c[t] = 0.f;
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(one(3, 0x08, g == 0), one(3, x, g == 0), c, 2, 2, 0, 127, 0, 127);
        if (lane == 0) out[64 + x] = c[0];
        for (int t = 0; t < 16; ++t) c[t] = 0.f;
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(one(5, x, g == 1), one(5, x, g == 1), c, 2, 2, 0, 127, 0, 127);
        if (lane == 0) out[128 + x] = c[0];
    }
}
int main()
{
    float *d, h[192]; hipMalloc(&d, sizeof(h)); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int x = 0; x < 64; ++x) printf("code %02x: A-decode %g  B-decode %g  x*x %g\n", x, h[x], h[64 + x], h[128 + x]);
    return 0;
}
