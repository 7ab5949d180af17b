// sweep_variants.hip -- dst = src * rho over a 128^3 x 6 fp32 field (48 MiB in, 48 MiB out, ping-pong like the generation loop):
// how fast can the evaporation sweep (ACSRank_3D.hpp:268-272) go while both buffers sit in the 256 MiB Infinity Cache, and with
// which shape?  (diagnostic, not product)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sweep_variants.hip -o build/sweep_variants && build/sweep_variants [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT_LD, bool NT_ST>
__global__ void k_sweep(const v4f *__restrict__ s4, v4f *__restrict__ d4, long n4, float rho)
{
    const long gsz = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * gsz < n4; i += U * gsz) {
        v4f a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = NT_LD ? __builtin_nontemporal_load(s4 + i + u * gsz) : s4[i + u * gsz];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] *= rho;
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT_ST) __builtin_nontemporal_store(a[u], d4 + i + u * gsz); else d4[i + u * gsz] = a[u]; }
    }
    for (; i < n4; i += gsz) { v4f a = s4[i]; a *= rho; d4[i] = a; }
}

// contiguous chunk per block instead of grid-stride (each block streams one region)
template <int U>
__global__ void k_sweep_chunk(const v4f *__restrict__ s4, v4f *__restrict__ d4, long n4, float rho)
{
    const long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    long i = lo + threadIdx.x;
    for (; i + (U - 1) * (long)blockDim.x < hi; i += U * (long)blockDim.x) {
        v4f a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = s4[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] *= rho;
#pragma unroll
        for (int u = 0; u < U; u++) d4[i + u * blockDim.x] = a[u];
    }
    for (; i < hi; i += blockDim.x) { v4f a = s4[i]; a *= rho; d4[i] = a; }
}

template <class F>
static double time_it(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 6; i++) launch(i);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch(i);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 128;
    const long nf = 6L * n * n * n, n4 = nf / 4;
    float *A, *B;
    hipMalloc(&A, nf * 4); hipMalloc(&B, nf * 4);
    hipMemset(A, 0, nf * 4); hipMemset(B, 0, nf * 4);
    const double bytes = 8.0 * nf;
    const int reps = 64;
#define RUN(name, expr)                                                                                   \
    {                                                                                                     \
        double us = time_it([&](int it) { const v4f *s = (const v4f *)((it & 1) ? B : A); v4f *d = (v4f *)((it & 1) ? A : B); expr; }, reps); \
        printf("%-58s %8.2f us  %6.2f TB/s\n", name, us, bytes / us / 1e6);                               \
    }
    for (int blocks : {2048, 4096, 8192, 16384}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride U=4, 256 thr, %d blocks (the product's shape at 4096)", blocks);
        RUN(nm, (k_sweep<4, false, false><<<blocks, 256>>>(s, d, n4, 0.999f)));
    }
    for (int blocks : {1024, 2048, 4096}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride U=8, 256 thr, %d blocks", blocks);
        RUN(nm, (k_sweep<8, false, false><<<blocks, 256>>>(s, d, n4, 0.999f)));
    }
    for (int blocks : {2048, 4096, 8192}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride U=2, 256 thr, %d blocks", blocks);
        RUN(nm, (k_sweep<2, false, false><<<blocks, 256>>>(s, d, n4, 0.999f)));
    }
    for (int thr : {512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride U=4, %d thr, %d blocks", thr, 4096 * 256 / thr);
        RUN(nm, (k_sweep<4, false, false><<<4096 * 256 / thr, thr>>>(s, d, n4, 0.999f)));
    }
    RUN("grid-stride U=4, 4096 blocks, nontemporal loads", (k_sweep<4, true, false><<<4096, 256>>>(s, d, n4, 0.999f)));
    RUN("grid-stride U=4, 4096 blocks, nontemporal stores", (k_sweep<4, false, true><<<4096, 256>>>(s, d, n4, 0.999f)));
    RUN("grid-stride U=4, 4096 blocks, nontemporal both", (k_sweep<4, true, true><<<4096, 256>>>(s, d, n4, 0.999f)));
    for (int blocks : {1024, 2048, 4096, 8192}) {
        char nm[128];
        snprintf(nm, sizeof nm, "contiguous chunk per block U=4, 256 thr, %d blocks", blocks);
        RUN(nm, (k_sweep_chunk<4><<<blocks, 256>>>(s, d, n4, 0.999f)));
    }
    for (int blocks : {2048, 4096}) {
        char nm[128];
        snprintf(nm, sizeof nm, "contiguous chunk per block U=8, 256 thr, %d blocks", blocks);
        RUN(nm, (k_sweep_chunk<8><<<blocks, 256>>>(s, d, n4, 0.999f)));
    }
    return 0;
}
