// Where do the two wavefronts of a 128-thread workgroup land?  (walk blocks: one walking wave + one helper wave, 132 KB of LDS
// so that a CU holds one block.)  Prints how many blocks have both waves on the same SIMD.
// build: hipcc --offload-arch=gfx950 -O2 -o build/wave_placement tools/ubench/wave_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(128) void k(unsigned *out)
{
    extern __shared__ int lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 2 + (threadIdx.x >> 6)] = id;
    lds[threadIdx.x] = id;
}
int main()
{
    const int B = 256;
    unsigned *d;
    hipMalloc(&d, B * 2 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 135424);
    k<<<B, 128, 135424>>>(d);
    std::vector<unsigned> h(B * 2);
    hipMemcpy(h.data(), d, B * 2 * 4, hipMemcpyDeviceToHost);
    int same = 0;
    for (int b = 0; b < B; b++) {
        unsigned a = h[b * 2], c = h[b * 2 + 1];
        unsigned sa = (a >> 4) & 3, sc = (c >> 4) & 3;
        if (sa == sc) same++;
        if (b < 8) printf("block %d: wave0 hw_id %08x (wave %u simd %u cu %u sh %u se %u)  wave1 %08x (wave %u simd %u cu %u)\n", b, a, a & 15, sa, (a >> 8) & 15, (a >> 12) & 1, (a >> 13) & 7, c, c & 15, sc, (c >> 8) & 15);
    }
    printf("%d of %d blocks have both wavefronts on the same SIMD\n", same, B);
    return 0;
}
