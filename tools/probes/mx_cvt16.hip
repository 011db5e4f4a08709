// probe: v_cvt_scalef32_pk32_fp6_f16 - element order, scale, f16 subnormal inputs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
__global__ void k(const float *in, float scale, unsigned *out)
{
    h32 v;
    for (int i = 0; i < 32; ++i) v[i] = (_Float16)in[threadIdx.x * 32 + i];
    u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
int main()
{
    float h[64 * 32];
    for (int i = 0; i < 64 * 32; ++i) h[i] = 0.f;
    for (int i = 0; i < 32; ++i) h[i] = 0.125f * (i % 16) * (i < 16 ? 1 : -1);      // lane 0: order
    for (int i = 0; i < 32; ++i) h[32 + i] = (0.25f + 0.125f * i) * 1e-6f;            // lane 1: f16 subnormals (scale 2^-22)
    for (int i = 0; i < 32; ++i) h[64 + i] = 0.3f + 0.01f * i;                        // lane 2: rounding
    float *din; unsigned *dout; hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 64 * 24); hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    const float scs[3] = {1.0f, 1.0f / 4194304.0f, 0.125f};
    for (int s = 0; s < 3; ++s) {
        k<<<1, 64>>>(din, scs[s], dout);
        unsigned o[64 * 6]; hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        const int l = s;   // lane s under scale s
        printf("scale %g lane %d fields:", scs[s], l);
        for (int n = 0; n < 32; ++n) {
            const unsigned long long bit = 6ull * n; const unsigned w = bit / 32, sh = bit % 32;
            const unsigned long long two = o[l * 6 + w] | ((unsigned long long)(w + 1 < 6 ? o[l * 6 + w + 1] : 0) << 32);
            printf(" %02llx", (two >> sh) & 63);
        }
        printf("\n");
    }
    return 0;
}
