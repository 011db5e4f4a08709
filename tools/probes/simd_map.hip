// probe: which SIMD each wave of a 512-thread workgroup lands on (HW_ID bits [5:4])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out)
{
    unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned *d, h[8 * 8]; hipMalloc(&d, sizeof(h)); k<<<8, 512, 150 * 1024>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; ++b) { printf("block %d simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u\n", (h[b * 8] >> 8) & 15); }
    return 0;
}
