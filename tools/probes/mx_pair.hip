// probe: which (lane group, 6-bit field) of A pairs with which (lane group, field) of B in
// mfma_scale_f32_32x32x64_f8f6f4 with fp6 operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ i32x8 onehot(int field, bool on)
{
    i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!on) return v;
    const unsigned long long bits = 0x08ull;           // 1.0 in e2m3
    const int bit = 6 * field, w = bit / 32, sh = bit % 32;
    unsigned lo = (unsigned)(bits << sh), hi = (sh > 26) ? (unsigned)(bits >> (32 - sh)) : 0u;
    v[w] = (int)lo; if (w + 1 < 6) v[w + 1] = (int)hi;
    return v;
}

__global__ void k(unsigned char *M /*[64][64]*/)
{
    const int lane = threadIdx.x, g = lane >> 5;
    for (int a = 0; a < 64; ++a)
        for (int b = 0; b < 64; ++b) {
            const i32x8 ia = onehot(a & 31, g == (a >> 5));
            const i32x8 ib = onehot(b & 31, g == (b >> 5));
            f32x16 c; for (int t = 0; t < 16; ++t) c[t] = 0.f;
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c, 2, 2, 0, 127, 0, 127);
            if (lane == 0) M[a * 64 + b] = (c[0] != 0.f) ? 1 : 0;
        }
}
int main()
{
    unsigned char *d, h[4096]; hipMalloc(&d, 4096); k<<<1, 64>>>(d); hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    for (int a = 0; a < 64; ++a) { printf("A(g%d,f%2d) pairs with B:", a >> 5, a & 31); for (int b = 0; b < 64; ++b) if (h[a * 64 + b]) printf(" (g%d,f%d)", b >> 5, b & 31); printf("\n"); }
    return 0;
}
