// Does the voxel ORDER of a search's field matter for a saturated walk launch (profiles/HISTORY.md)?  2 048 wavefronts, each taking a
// synthetic random lattice walk in its own 256^3 field of 24-byte records (as many fields as fit: 224 x 402 MB), every step loading
// the six neighbours' records of the voxel it stands on (36 lanes x 4 B, like the walk loop) and waiting for them.  Address of
// voxel (x, y, z): row-major (the product's layout: a step in z is 1.5 MB away, i.e. always another 2-MB page) or brick-major (16^3
// bricks of 96 KB, bricks in row-major order: a walk stays inside a few bricks for many steps).
// build: hipcc --offload-arch=gfx950 -O2 -o build/page_locality tools/ubench/page_locality.hip ; run: build/page_locality [fields]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <bool BRICK>
__device__ __forceinline__ int64_t vox(int x, int y, int z, int n)
{
    if (!BRICK) return ((int64_t)z * n + y) * n + x;
    const int nb = n >> 4;
    return ((((int64_t)(z >> 4) * nb + (y >> 4)) * nb + (x >> 4)) << 12) + (((z & 15) << 8) | ((y & 15) << 4) | (x & 15));
}
template <bool BRICK>
__global__ __launch_bounds__(64) void k(const float *fields, int64_t stride, int n, int steps, int nfields, float *sink)
{
    const int lane = threadIdx.x, w = blockIdx.x;
    const float *f = fields + (int64_t)(w % nfields) * stride;
    int x = 8 + (w * 37) % (n - 16), y = 8 + (w * 101) % (n - 16), z = 8 + (w * 53) % (n - 16);
    const int j = lane / 6, k6 = lane % 6;                      // lanes 0..35: neighbour j, edge k6
    const int dx = j == 0 ? -1 : j == 1 ? 1 : 0, dy = j == 2 ? -1 : j == 3 ? 1 : 0, dz = j == 4 ? -1 : j == 5 ? 1 : 0;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        if (lane < 36) {
            int X = min(max(x + dx, 0), n - 1), Y = min(max(y + dy, 0), n - 1), Z = min(max(z + dz, 0), n - 1);
            acc += f[vox<BRICK>(X, Y, Z, n) * 6 + k6];
        }
        // a persistent random walk (keeps its direction 3 times out of 4), the same for both layouts
        const uint32_t r = mix((uint32_t)w * 2654435761u + (uint32_t)s);
        const int dir = (r & 3) ? (int)((mix((uint32_t)w + (uint32_t)(s >> 3) * 40503u) % 6)) : (int)((r >> 2) % 6);
        x = min(max(x + (dir == 0 ? -1 : dir == 1 ? 1 : 0), 1), n - 2);
        y = min(max(y + (dir == 2 ? -1 : dir == 3 ? 1 : 0), 1), n - 2);
        z = min(max(z + (dir == 4 ? -1 : dir == 5 ? 1 : 0), 1), n - 2);
        acc = __shfl(acc, 0) * 0.f + acc;                       // the loads are waited for every step, like records are
    }
    if (acc == 123.456f) sink[w] = acc;
}
int main(int argc, char **argv)
{
    const int n = 256, waves = 2048, steps = 2000;
    int nfields = argc > 1 ? atoi(argv[1]) : 224;
    const int64_t stride = (int64_t)n * n * n * 6;
    float *fields, *sink;
    if (hipMalloc(&fields, sizeof(float) * stride * nfields) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipMalloc(&sink, 4 * waves);
    hipMemset(fields, 0, sizeof(float) * stride * nfields);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int layout = 0; layout < 2; layout++)
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (layout == 0) k<false><<<waves, 64>>>(fields, stride, n, steps, nfields, sink);
            else k<true><<<waves, 64>>>(fields, stride, n, steps, nfields, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%s, %d fields of 402 MB, %d waves x %d steps: %.3f ms = %.1f ns per step and wave\n", layout ? "brick-major (16^3)" : "row-major", nfields, waves, steps, ms, ms * 1e6 / steps);
        }
    return 0;
}
