// sweep_gap_pmc.hip -- why does the evaporation sweep (dst = src * rho, 128^3 x 6 fp32, ping-pong; ACSRank_3D.hpp:268-272) last
// 14.4 us when it follows itself and 16.2-16.5 us when ANY other launch sits in between (profiles/r03/fused_launch_anatomy.txt)?
// One mode per process, so that rocprofv3 --pmc passes give per-dispatch counters of exactly that sequence (diagnostic, not product):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sweep_gap_pmc.hip -o build/sweep_gap_pmc
//   build/sweep_gap_pmc <mode> [reps]        mode: b2b | idle | valu | tiny | touch | twin | idle_b2b2
//     b2b        sweep, sweep, sweep, ...                                    (its own predecessor)
//     idle       (one wavefront spinning ~20 us, no memory) , sweep, ...
//     valu       (every SIMD issuing VALU for ~20 us, no memory), sweep, ...
//     tiny       (a 1-block kernel that returns at once), sweep, ...
//     touch      (one wavefront spinning ~20 us, then 1 dword per 4 KiB page of BOTH buffers read by 256 blocks), sweep, ...
//     twin       sweep, sweep_twin (same code, another kernel symbol), sweep, ...  (is it the kernel OBJECT?)
//     idle_b2b2  idle, sweep, sweep, idle, sweep, sweep ...                  (first vs second sweep behind a gap)
//     stream     (the same sweep kernel on two OTHER 48 MiB buffers), sweep, ...  (the memory system stays loaded, the buffers differ)
// Prints, per group of sweeps: per-dispatch duration from start/stop events (hipExtLaunchKernelGGL) and the shader clock the sweep's
// block 0 saw (delta s_memtime / delta s_memrealtime x 100 MHz).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
struct Clk { unsigned long long c0, r0, c1, r1; };
struct Blk { unsigned int r0, r1; };   // every block's first and last s_memrealtime (100 MHz), low words

#define SWEEP_BODY                                                                                               \
    const unsigned int br0 = (unsigned int)__builtin_amdgcn_s_memrealtime();                                    \
    unsigned long long c0 = 0, r0 = 0;                                                                          \
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); } \
    const long gsz = (long)gridDim.x * blockDim.x;                                                              \
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;                                                       \
    for (; i + 3 * gsz < n4; i += 4 * gsz) {                                                                     \
        v4f a[4];                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 4; u++) a[u] = s4[i + u * gsz];                                   \
        _Pragma("unroll") for (int u = 0; u < 4; u++) a[u] *= rho;                                              \
        _Pragma("unroll") for (int u = 0; u < 4; u++) d4[i + u * gsz] = a[u];                                   \
    }                                                                                                           \
    for (; i < n4; i += gsz) { v4f a = s4[i]; a *= rho; d4[i] = a; }                                            \
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk->c0 = c0; clk->r0 = r0; clk->c1 = __builtin_amdgcn_s_memtime(); clk->r1 = __builtin_amdgcn_s_memrealtime(); } \
    if (blk && threadIdx.x == 0) { blk[blockIdx.x].r0 = br0; blk[blockIdx.x].r1 = (unsigned int)__builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(256) void k_sweep(const v4f *__restrict__ s4, v4f *__restrict__ d4, long n4, float rho, Clk *clk, Blk *blk) { SWEEP_BODY }
__global__ __launch_bounds__(256) void k_sweep_twin(const v4f *__restrict__ s4, v4f *__restrict__ d4, long n4, float rho, Clk *clk, Blk *blk) { SWEEP_BODY }

// spins until `ticks` of the 100 MHz clock have passed; lanes issue VALU adds meanwhile when `busy`
__global__ __launch_bounds__(256) void k_spin(unsigned long long ticks, int busy, float *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x, b = 1.f;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (busy) {
#pragma unroll
            for (int u = 0; u < 64; u++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (a == 12345.678f) sink[0] = a;
}
__global__ void k_tiny(float *sink) { if (threadIdx.x == 1234567) sink[0] = 1.f; }
// one dword per 4 KiB of both buffers (every translation the sweep will need, none of its data to speak of)
__global__ __launch_bounds__(256) void k_touch(const float *A, const float *B, long nf, float *sink)
{
    float acc = 0.f;
    for (long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 1024; p < nf; p += (long)gridDim.x * blockDim.x * 1024) acc += A[p] + B[p];
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "b2b";
    const int reps = argc > 2 ? atoi(argv[2]) : 60;
    const int n = 128;
    const long nf = 6L * n * n * n, n4 = nf / 4;
    float *A, *B, *sink;
    Clk *clk;
    Blk *blk;
    float *Cb = nullptr, *Db = nullptr;
    const int NB = 4096;
    hipMalloc(&A, nf * 4); hipMalloc(&B, nf * 4); hipMalloc(&sink, 256); hipMalloc(&clk, sizeof(Clk) * (2 * reps + 16));
    hipMalloc(&blk, sizeof(Blk) * NB * (2 * reps + 16));
    hipMemset(A, 0, nf * 4); hipMemset(B, 0, nf * 4);
    if (!strcmp(mode, "stream") || !strcmp(mode, "chain")) { hipMalloc(&Cb, nf * 4); hipMalloc(&Db, nf * 4); hipMemset(Cb, 0, nf * 4); hipMemset(Db, 0, nf * 4); }
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<hipEvent_t> e0, e1;
    std::vector<int> kind;   // 0: first sweep behind whatever precedes, 1: second sweep (idle_b2b2)
    int it = 0, nsw = 0;
    auto sweep = [&](bool twin, int k, bool ext = true) {
        const v4f *s = (const v4f *)((it & 1) ? B : A);
        v4f *d = (v4f *)((it & 1) ? A : B);
        it++;
        hipEvent_t a = nullptr, b = nullptr;
        if (!ext) {   // a plain launch: no events, no completion signal of its own -- only the in-kernel stamps say how long it took
            k_sweep<<<4096, 256, 0, st>>>(s, d, n4, 0.999f, clk + nsw, blk + (size_t)nsw * NB);
            e0.push_back(a); e1.push_back(b); kind.push_back(k); nsw++;
            return;
        }
        hipEventCreate(&a); hipEventCreate(&b);
        if (twin) hipExtLaunchKernelGGL(k_sweep_twin, dim3(4096), dim3(256), 0, st, a, b, 0, s, d, n4, 0.999f, clk + nsw, blk + (size_t)nsw * NB);
        else hipExtLaunchKernelGGL(k_sweep, dim3(4096), dim3(256), 0, st, a, b, 0, s, d, n4, 0.999f, clk + nsw, blk + (size_t)nsw * NB);
        e0.push_back(a); e1.push_back(b); kind.push_back(k);
        nsw++;
    };
    const unsigned long long T20 = 2000;   // 20 us of the 100 MHz clock
    for (int w = 0; w < 8; w++) sweep(false, -1);   // warm-up
    hipStreamSynchronize(st);
    hipEvent_t w0, w1;
    hipEventCreate(&w0); hipEventCreate(&w1);
    hipEventRecord(w0, st);
    for (int r = 0; r < reps; r++) {
        if (!strcmp(mode, "b2b")) sweep(false, 0);
        else if (!strcmp(mode, "idle")) { k_spin<<<1, 64, 0, st>>>(T20, 0, sink); sweep(false, 0); }
        else if (!strcmp(mode, "valu")) { k_spin<<<ncu * 4, 256, 0, st>>>(T20, 1, sink); sweep(false, 0); }
        else if (!strcmp(mode, "tiny")) { k_tiny<<<1, 64, 0, st>>>(sink); sweep(false, 0); }
        else if (!strcmp(mode, "touch")) { k_spin<<<1, 64, 0, st>>>(T20, 0, sink); k_touch<<<256, 256, 0, st>>>(A, B, nf, sink); sweep(false, 0); }
        else if (!strcmp(mode, "twin")) { sweep((r & 1) != 0, r & 1); }
        else if (!strcmp(mode, "plain_b2b")) sweep(false, 0, false);
        else if (!strcmp(mode, "tiny_plain")) { k_tiny<<<1, 64, 0, st>>>(sink); sweep(false, 0, false); }
        else if (!strcmp(mode, "idle_plain")) { k_spin<<<1, 64, 0, st>>>(T20, 0, sink); sweep(false, 0, false); }
        else if (!strcmp(mode, "idle_ext") || !strcmp(mode, "tiny_ext") || !strcmp(mode, "ext_then_plain")) {   // the kernel in between launched WITH events
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            if (!strcmp(mode, "tiny_ext")) hipExtLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, a, b, 0, sink);
            else hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, a, b, 0, T20, 0, sink);
            sweep(false, 0, strcmp(mode, "ext_then_plain") != 0);
        }
        else if (!strcmp(mode, "chain")) {      // A->B, B->C, C->D, D->A ...: every sweep reads what its predecessor wrote, on rotating buffers
            float *bufs[4] = {A, B, Cb, Db};
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipExtLaunchKernelGGL(k_sweep, dim3(4096), dim3(256), 0, st, a, b, 0, (const v4f *)bufs[r & 3], (v4f *)bufs[(r + 1) & 3], n4, 0.999f, clk + nsw, blk + (size_t)nsw * NB);
            e0.push_back(a); e1.push_back(b); kind.push_back(0); nsw++;
        }
        else if (!strcmp(mode, "samedir")) {    // A->B, A->B, A->B ...: same buffers, nothing read that the predecessor wrote
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipExtLaunchKernelGGL(k_sweep, dim3(4096), dim3(256), 0, st, a, b, 0, (const v4f *)A, (v4f *)B, n4, 0.999f, clk + nsw, blk + (size_t)nsw * NB);
            e0.push_back(a); e1.push_back(b); kind.push_back(0); nsw++;
        }
        else if (!strcmp(mode, "half")) {       // the same sweep with a quarter of the blocks (half the CUs' wave slots idle) on the same buffers in between
            const v4f *s_ = (const v4f *)((it & 1) ? B : A); v4f *d_ = (v4f *)((it & 1) ? A : B); it++;
            k_sweep<<<1024, 256, 0, st>>>(s_, d_, n4, 0.999f, nullptr, nullptr); sweep(false, 0);
        }
        else if (!strcmp(mode, "full_spin")) {  // every wave slot of the chip taken by a spinning kernel of the sweep's block shape for ~20 us
            k_spin<<<ncu * 8, 256, 0, st>>>(T20, 1, sink); sweep(false, 0);
        }
        else if (!strcmp(mode, "stream")) { k_sweep<<<4096, 256, 0, st>>>((const v4f *)((r & 1) ? Db : Cb), (v4f *)((r & 1) ? Cb : Db), n4, 0.999f, nullptr, nullptr); sweep(false, 0); }
        else if (!strcmp(mode, "idle_b2b2")) { k_spin<<<1, 64, 0, st>>>(T20, 0, sink); sweep(false, 0); sweep(false, 1); }
        else { fprintf(stderr, "unknown mode %s\n", mode); return 2; }
    }
    hipEventRecord(w1, st);
    hipStreamSynchronize(st);
    { float wms = 0; hipEventElapsedTime(&wms, w0, w1); printf("mode %-10s whole sequence: %.2f us per repetition (stream time between two events around all %d repetitions)\n", mode, wms * 1e3 / reps, reps); }
    std::vector<Clk> h(nsw);
    hipMemcpy(h.data(), clk, sizeof(Clk) * nsw, hipMemcpyDeviceToHost);
    double us[2] = {0, 0}, ghz[2] = {0, 0};
    int cnt[2] = {0, 0};
    // in-kernel view of every sweep dispatch: span = last block's end - first block's start (100 MHz ticks), and how the 4096 blocks'
    // starts and ends are spread over it (the device holds 2048 of these blocks at once: the second half starts as the first retires)
    std::vector<Blk> hb((size_t)nsw * NB);
    hipMemcpy(hb.data(), blk, sizeof(Blk) * hb.size(), hipMemcpyDeviceToHost);
    double span[2] = {0, 0}, s50[2] = {0, 0}, s100[2] = {0, 0}, e50[2] = {0, 0}, blk_first[2] = {0, 0}, blk_second[2] = {0, 0};
    for (int i = 0; i < nsw; i++) {
        if (kind[i] < 0) continue;
        float ms = 0;
        if (e0[i]) hipEventElapsedTime(&ms, e0[i], e1[i]);
        us[kind[i]] += ms * 1e3;
        ghz[kind[i]] += (double)(h[i].c1 - h[i].c0) / (double)(h[i].r1 - h[i].r0) * 0.1;
        cnt[kind[i]]++;
        {
            const Blk *q = hb.data() + (size_t)i * NB;
            std::vector<unsigned int> st_(NB), en_(NB);
            unsigned int t0 = q[0].r0;
            for (int j = 0; j < NB; j++) if ((int)(q[j].r0 - t0) < 0) t0 = q[j].r0;
            double d1 = 0, d2 = 0; int n1 = 0, n2 = 0;
            for (int j = 0; j < NB; j++) { st_[j] = q[j].r0 - t0; en_[j] = q[j].r1 - t0; }
            std::vector<unsigned int> ss = st_, ee = en_;
            std::sort(ss.begin(), ss.end()); std::sort(ee.begin(), ee.end());
            const unsigned int mid = ss[NB / 2];
            for (int j = 0; j < NB; j++) { if (st_[j] < mid) { d1 += en_[j] - st_[j]; n1++; } else { d2 += en_[j] - st_[j]; n2++; } }
            span[kind[i]] += ee[NB - 1] * 0.01; s50[kind[i]] += ss[NB / 2 - 1] * 0.01; s100[kind[i]] += ss[NB - 1] * 0.01; e50[kind[i]] += ee[NB / 2] * 0.01;
            blk_first[kind[i]] += d1 / (n1 ? n1 : 1) * 0.01; blk_second[kind[i]] += d2 / (n2 ? n2 : 1) * 0.01;
        }
    }
    for (int k = 0; k < 2; k++)
        if (cnt[k])
            printf("mode %-10s sweep kind %d: %6.2f us per dispatch (events), shader clock seen by block 0 %.3f GHz, %d dispatches, %.2f TB/s\n", mode, k,
                   us[k] / cnt[k], ghz[k] / cnt[k], cnt[k], 8.0 * nf / (us[k] / cnt[k]) / 1e6),
            printf("     in-kernel (s_memrealtime of every block): first start -> last end %6.2f us; block 2048 of 4096 started at %5.2f us, the last at %5.2f us; half of the blocks done at %5.2f us;"
                   " a block of the first half lasts %5.2f us, of the second half %5.2f us\n", span[k] / cnt[k], s50[k] / cnt[k], s100[k] / cnt[k], e50[k] / cnt[k], blk_first[k] / cnt[k], blk_second[k] / cnt[k]);
    return 0;
}
