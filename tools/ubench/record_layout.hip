// What would 32-byte records buy a saturated walk launch (profiles/HISTORY.md)?  Waves take synthetic lattice walks, each in its own field set
// (as in tools/ubench/page_locality.hip), and every step load what the lazy walk loop loads for the six neighbours of the voxel:
//   layout A (the product's): pheromone [N][6] floats, heuristic [N][6] floats (shared by 64 waves), stamp [N] u32  -- three arrays, 24-B records
//   layout B: pheromone + stamp [N][8] (six values, the stamp, one pad), heuristic [N][8]                           -- two arrays, 32-B records
// and wait for them (the real loop requests one step ahead; here the step count per second is what is compared).
// build: hipcc --offload-arch=gfx950 -O2 -o build/record_layout tools/ubench/record_layout.hip ; run: build/record_layout [fields] [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int RS>
__global__ __launch_bounds__(64) void k(const float *pher, const float *heur, const uint32_t *stamp, int64_t n3, int n, int steps, int nfields, float *sink)
{
    const int lane = threadIdx.x, w = blockIdx.x;
    const float *p = pher + (int64_t)(w % nfields) * n3 * RS;
    const float *h = heur + (int64_t)((w % nfields) / 64) * n3 * RS;   // ~64 searches share a heuristic field
    const uint32_t *st = stamp + (int64_t)(w % nfields) * n3;
    int x = 8 + (w * 37) % (n - 16), y = 8 + (w * 101) % (n - 16), z = 8 + (w * 53) % (n - 16);
    const int j = lane / 6, k6 = lane % 6;
    const int dx = j == 0 ? -1 : j == 1 ? 1 : 0, dy = j == 2 ? -1 : j == 3 ? 1 : 0, dz = j == 4 ? -1 : j == 5 ? 1 : 0;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        if (lane < 42) {
            const int jj = lane < 36 ? j : lane - 36;             // lanes 36..41: the six stamps
            const int ddx = jj == 0 ? -1 : jj == 1 ? 1 : 0, ddy = jj == 2 ? -1 : jj == 3 ? 1 : 0, ddz = jj == 4 ? -1 : jj == 5 ? 1 : 0;
            const int64_t v = ((int64_t)min(max(z + (lane < 36 ? dz : ddz), 0), n - 1) * n + min(max(y + (lane < 36 ? dy : ddy), 0), n - 1)) * n + min(max(x + (lane < 36 ? dx : ddx), 0), n - 1);
            if (lane < 36) acc += p[v * RS + k6] + h[v * RS + k6];
            else acc += RS == 8 ? p[v * 8 + 6] : __uint_as_float(st[v]);
        }
        const uint32_t r = mix((uint32_t)w * 2654435761u + (uint32_t)s);
        const int dir = (r & 3) ? (int)((mix((uint32_t)w + (uint32_t)(s >> 3) * 40503u) % 6)) : (int)((r >> 2) % 6);
        x = min(max(x + (dir == 0 ? -1 : dir == 1 ? 1 : 0), 1), n - 2);
        y = min(max(y + (dir == 2 ? -1 : dir == 3 ? 1 : 0), 1), n - 2);
        z = min(max(z + (dir == 4 ? -1 : dir == 5 ? 1 : 0), 1), n - 2);
        acc = __shfl(acc, 0) * 0.f + acc;
    }
    if (acc == 123.456f) sink[w] = acc;
}
int main(int argc, char **argv)
{
    const int n = 256, steps = 2000;
    const int nfields = argc > 1 ? atoi(argv[1]) : 128, waves = argc > 2 ? atoi(argv[2]) : 2048;
    const int64_t n3 = (int64_t)n * n * n;
    float *pher, *heur, *sink;
    uint32_t *stamp;
    if (hipMalloc(&pher, sizeof(float) * n3 * 8 * nfields) != hipSuccess || hipMalloc(&heur, sizeof(float) * n3 * 8 * (nfields / 64 + 1)) != hipSuccess ||
        hipMalloc(&stamp, 4 * n3 * nfields) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipMalloc(&sink, 4 * waves);
    hipMemset(pher, 0, sizeof(float) * n3 * 8 * nfields);
    hipMemset(heur, 0, sizeof(float) * n3 * 8 * (nfields / 64 + 1));
    hipMemset(stamp, 0, 4 * n3 * nfields);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int layout = 0; layout < 2; layout++)
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (layout == 0) k<6><<<waves, 64>>>(pher, heur, stamp, n3, n, steps, nfields, sink);
            else k<8><<<waves, 64>>>(pher, heur, stamp, n3, n, steps, nfields, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%s, %d field sets, %d waves x %d steps: %.3f ms = %.1f ns per step and wave, %.2f steps/ns in all\n",
                            layout ? "32-byte records, stamp inside (2 arrays)" : "24-byte records + stamps (3 arrays) ", nfields, waves, steps, ms, ms * 1e6 / steps, (double)waves * steps / (ms * 1e6));
        }
    return 0;
}
