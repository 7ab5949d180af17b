// vmm_threads.hip -- does the driver's zero-fill of fresh device memory go faster from several host threads?  (diagnostic, not product;
// round 5: the first C5-sized solver of a process spends ~5 s in it, profiles/r05/vmm_probe.txt)
//   vmm_threads <threads> <GiB> [mode]     mode 0: hipMemCreate of 512 MiB chunks (what the context's arena does), 1: hipMalloc of 512 MiB blocks
// One configuration per process: memory a process has freed comes back to the same process without the wipe.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/vmm_threads.hip -o build/vmm_threads -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static const size_t CH = (size_t)512 << 20;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 1;
    const size_t gib = argc > 2 ? (size_t)atoll(argv[2]) : 64;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const size_t N = gib * 2;   // chunks
    if (hipSetDevice(0) != hipSuccess) return 1;
    hipFree(nullptr);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    std::vector<void *> blocks(N, nullptr);
    std::vector<double> t_thread(T, 0.0);
    std::vector<int> bad(T, 0);
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            hipSetDevice(0);
            const double a = now();
            for (size_t i = t; i < N; i += T) {
                hipError_t e = mode == 0 ? hipMemCreate(&h[i], CH, &prop, 0) : hipMalloc(&blocks[i], CH);
                if (e != hipSuccess) bad[t]++;
            }
            t_thread[t] = now() - a;
        });
    for (auto &x : th) x.join();
    const double t_create = now() - t0;
    int nbad = 0;
    for (int b : bad) nbad += b;
    double t_map = 0, t_touch = 0;
    if (mode == 0 && !nbad) {
        void *va = nullptr;
        const double a = now();
        if (hipMemAddressReserve(&va, N * CH, 0, nullptr, 0) != hipSuccess) return 2;
        for (size_t i = 0; i < N; i++)
            if (hipMemMap((char *)va + i * CH, CH, 0, h[i], 0) != hipSuccess) return 3;
        if (hipMemSetAccess(va, N * CH, &acc, 1) != hipSuccess) return 4;
        t_map = now() - a;
        const double b = now();
        hipMemset(va, 1, N * CH);
        hipDeviceSynchronize();
        t_touch = now() - b;
    } else if (!nbad) {
        const double b = now();
        for (size_t i = 0; i < N; i++) hipMemsetAsync(blocks[i], 1, CH, 0);
        hipDeviceSynchronize();
        t_touch = now() - b;
    }
    double slowest = 0;
    for (double x : t_thread) slowest = x > slowest ? x : slowest;
    printf("%s, %2d thread%s, %3zu GiB fresh: allocation %.3f s (%.1f ms per GiB; slowest thread %.3f s), map + access %.3f s, first touch %.3f s%s\n",
           mode == 0 ? "hipMemCreate 512 MiB chunks" : "hipMalloc 512 MiB blocks  ", T, T == 1 ? " " : "s", gib, t_create, t_create / gib * 1e3, slowest, t_map, t_touch,
           nbad ? "  (FAILED allocations)" : "");
    return 0;
}
