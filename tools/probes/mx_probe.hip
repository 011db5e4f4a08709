// probe: semantics of cvt_scalef32_pk32_fp6_f32 and of the scale operands of
// mfma_scale_f32_32x32x64_f8f6f4 (fp6 e2m3) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

__global__ void k_cvt(const float *in, float scale, unsigned *out)
{
    f32x16 v0, v1;
    for (int i = 0; i < 16; ++i) { v0[i] = in[threadIdx.x * 32 + i]; v1[i] = in[threadIdx.x * 32 + 16 + i]; }
    u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(v0, v1, scale);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}

// D = A*B, operands converted per lane slice by the same instruction
__global__ void k_mfma(const float *A /*[32][64]*/, const float *B /*[64][32]*/, float sa_cvt, float sb_cvt, int sa, int sb,
                       float *D /*[32][32]*/, float zero)
{
    const int lane = threadIdx.x, r = lane & 31, g = lane >> 5;
    f32x16 va0, va1, vb0, vb1;
    for (int e = 0; e < 16; ++e) {
        va0[e] = A[r * 64 + 32 * g + e]; va1[e] = A[r * 64 + 32 * g + 16 + e];
        vb0[e] = B[(32 * g + e) * 32 + r]; vb1[e] = B[(32 * g + 16 + e) * 32 + r];
    }
    u32x6 pa = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(va0, va1, sa_cvt);
    u32x6 pb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(vb0, vb1, sb_cvt);
    i32x8 ia = {(int)pa[0], (int)pa[1], (int)pa[2], (int)pa[3], (int)pa[4], (int)pa[5], 0, 0};
    i32x8 ib = {(int)pb[0], (int)pb[1], (int)pb[2], (int)pb[3], (int)pb[4], (int)pb[5], 0, 0};
    f32x16 c;
    for (int t = 0; t < 16; ++t) c[t] = zero;   // a literal 0 lets the compiler overlap vdst with srcB (garbage)
    asm volatile("" : "+v"(c));
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c, 2, 2, 0, sa, 0, sb);
    for (int t = 0; t < 16; ++t) { const int row = (t & 3) + 8 * (t >> 2) + 4 * g; D[row * 32 + r] = c[t]; }
}

int main()
{
    float h[64 * 32];
    for (int i = 0; i < 64 * 32; ++i) h[i] = 0.f;
    const float vals[16] = {0.f, 0.125f, 0.25f, 0.5f, 0.875f, 1.f, 1.125f, 1.5f, 2.f, 3.f, 3.75f, 4.f, 6.f, 7.5f, 8.f, 100.f};
    for (int i = 0; i < 16; ++i) { h[i] = vals[i]; h[16 + i] = -vals[i]; }
    h[32 + 0] = 1.f; h[64 + 5] = 1.f; h[96 + 31] = 1.f;
    for (int i = 0; i < 32; ++i) h[128 + i] = 3.0f;
    for (int i = 0; i < 32; ++i) h[160 + i] = 0.3f + 0.01f * i;      // rounding mode probe
    float *din; unsigned *dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 64 * 6 * 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    const float scs[3] = {1.0f, 2.0f, 0.5f};
    for (float sc : scs) {
        k_cvt<<<1, 64>>>(din, sc, dout);
        unsigned o[64 * 6];
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        printf("scale %.2f\n", sc);
        for (int l = 0; l < 6; ++l) {
            printf(" lane %d:", l);
            for (int i = 0; i < 6; ++i) printf(" %08x", o[l * 6 + i]);
            printf("\n  fields:");
            for (int n = 0; n < 32; ++n) {
                const unsigned long long bit = 6ull * n; const unsigned w = bit / 32, sh = bit % 32;
                const unsigned long long two = o[l * 6 + w] | ((unsigned long long)(w + 1 < 6 ? o[l * 6 + w + 1] : 0) << 32);
                printf(" %02llx", (two >> sh) & 63);
            }
            printf("\n");
        }
    }
    float A[32 * 64], B[64 * 32], Dh[32 * 32], Dr[32 * 32];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = (float)(((i * 3 + k * 5) % 7) - 3) * 0.5f;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)(((k * 2 + j * 7) % 5) - 2) * 0.25f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int k = 0; k < 64; ++k) s += (double)A[i * 64 + k] * B[k * 32 + j];
        Dr[i * 32 + j] = (float)s;
    }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dD, sizeof(Dh));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    struct Case { float sa_cvt, sb_cvt; int sa, sb; };
    const Case cases[] = {{1, 1, 127, 127}, {1, 1, 128, 127}, {1, 1, 127, 126}, {0.5f, 1, 126, 127}, {1, 0.25f, 127, 125}, {1, 1, 0, 0},
                          {1, 1, 0x7f7f7f7f, 0x7f7f7f7f}, {1, 1, 0x80 << 8 | 0x7f, 127}};
    for (const Case &c : cases) {
        k_mfma<<<1, 64>>>(dA, dB, c.sa_cvt, c.sb_cvt, c.sa, c.sb, dD, 0.f);
        hipMemcpy(Dh, dD, sizeof(Dh), hipMemcpyDeviceToHost);
        double maxerr = 0, ratio = 0; int cnt = 0;
        for (int i = 0; i < 1024; ++i) { maxerr = fmax(maxerr, fabs(Dh[i] - Dr[i])); if (fabs(Dr[i]) > 0.5) { ratio += Dh[i] / Dr[i]; ++cnt; } }
        printf("cvt-scales (%.2f, %.2f) mfma-scales (%#x, %#x): max|D-ref| %.4g  mean D/ref %.4g  D[0][0..3] %g %g %g %g (ref %g %g %g %g)\n",
               c.sa_cvt, c.sb_cvt, c.sa, c.sb, maxerr, ratio / cnt, Dh[0], Dh[1], Dh[2], Dh[3], Dr[0], Dr[1], Dr[2], Dr[3]);
    }
    return 0;
}
