// vmm_probe.hip -- can the context's block cache be an ARENA of physical chunks mapped into virtual ranges (hipMemCreate / hipMemMap)
// instead of whole hipMalloc blocks?  (diagnostic, not product; round 5)  Measures on the box it runs on:
//   * is virtual memory management supported, what is the allocation granularity;
//   * hipMemCreate / hipMemMap / hipMemSetAccess / hipMemUnmap / hipMemRelease per chunk of 32 MiB and 512 MiB;
//   * a streaming read+write kernel (the evaporation sweep's access pattern) on a mapped 2 GiB block against the same on hipMalloc memory;
//   * what re-mapping kept chunks into a NEW virtual range costs against free -> allocate of the same bytes (the wipe the driver does).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/vmm_probe.hip -o build/vmm_probe && build/vmm_probe [GiB of the free/alloc test]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void k_scale(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4, float r)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = src[i];
        v.x *= r; v.y *= r; v.z *= r; v.w *= r;
        dst[i] = v;
    }
}

static int time_kernel(const char *what, float *a, float *b, size_t bytes)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n4 = bytes / 16;
    for (int w = 0; w < 2; w++) k_scale<<<8192, 256>>>((const float4 *)a, (float4 *)b, n4, 0.8f);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 5; w++) k_scale<<<8192, 256>>>((const float4 *)a, (float4 *)b, n4, 0.8f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.1f us per pass of %zu MiB read + write = %.2f TB/s\n", what, ms * 200.0, bytes >> 20, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t big_gib = argc > 1 ? (size_t)atoi(argv[1]) : 64;
    int dev = 0, vmm = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", vmm);
    if (!vmm) return 0;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu B, recommended %zu B\n", gmin, grec);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    for (size_t chunk : {(size_t)32 << 20, (size_t)512 << 20}) {
        const size_t total = (size_t)4 << 30, n = total / chunk;
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        double t0 = now();
        for (size_t i = 0; i < n; i++) CK(hipMemCreate(&h[i], chunk, &prop, 0));
        const double t_create = now() - t0;
        void *va = nullptr;
        t0 = now();
        CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
        const double t_res = now() - t0;
        t0 = now();
        for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
        const double t_map = now() - t0;
        t0 = now();
        CK(hipMemSetAccess(va, total, &acc, 1));
        const double t_acc = now() - t0;
        CK(hipMemset(va, 0, total));
        CK(hipDeviceSynchronize());
        if (time_kernel(chunk == ((size_t)32 << 20) ? "mapped, 32 MiB chunks" : "mapped, 512 MiB chunks", (float *)va, (float *)va + total / 8, total / 2)) return 1;
        t0 = now();
        CK(hipMemUnmap(va, total));
        const double t_unmap = now() - t0;
        // the same chunks into a new range, in another order: what a re-created solver pays
        void *vb = nullptr;
        t0 = now();
        CK(hipMemAddressReserve(&vb, total, 0, nullptr, 0));
        for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)vb + i * chunk, chunk, 0, h[n - 1 - i], 0));
        CK(hipMemSetAccess(vb, total, &acc, 1));
        const double t_remap = now() - t0;
        if (time_kernel("re-mapped in reverse order", (float *)vb, (float *)vb + total / 8, total / 2)) return 1;
        CK(hipMemUnmap(vb, total));
        t0 = now();
        for (size_t i = 0; i < n; i++) CK(hipMemRelease(h[i]));
        const double t_rel = now() - t0;
        CK(hipMemAddressFree(va, total));
        CK(hipMemAddressFree(vb, total));
        printf("chunk %4zu MiB x %4zu: create %.3f ms each, reserve %.3f ms, map %.3f ms each, set access %.3f ms (whole range), unmap %.3f ms (whole), "
               "re-reserve + re-map + access %.3f ms (whole 4 GiB), release %.3f ms each\n",
               chunk >> 20, n, 1e3 * t_create / n, 1e3 * t_res, 1e3 * t_map / n, 1e3 * t_acc, 1e3 * t_unmap, 1e3 * t_remap, 1e3 * t_rel / n);
    }
    {   // hipMalloc memory for comparison
        float *m = nullptr;
        CK(hipMalloc((void **)&m, (size_t)4 << 30));
        CK(hipMemset(m, 0, (size_t)4 << 30));
        CK(hipDeviceSynchronize());
        if (time_kernel("hipMalloc", m, m + ((size_t)4 << 30) / 8, (size_t)2 << 30)) return 1;
        CK(hipFree(m));
    }
    {   // free -> allocate of big_gib GiB (the wipe) against unmap -> re-map of kept chunks
        const size_t total = big_gib << 30, chunk = (size_t)512 << 20, n = total / chunk;
        void *m = nullptr;
        double t0 = now();
        CK(hipMalloc(&m, total));
        CK(hipMemset(m, 1, total));
        CK(hipDeviceSynchronize());
        printf("hipMalloc + memset of %zu GiB: %.3f s\n", big_gib, now() - t0);
        t0 = now();
        CK(hipFree(m));
        const double t_free = now() - t0;
        t0 = now();
        CK(hipMalloc(&m, total));
        printf("hipFree %.3f s, then hipMalloc of the same %zu GiB: %.3f s\n", t_free, big_gib, now() - t0);
        CK(hipFree(m));
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        t0 = now();
        for (size_t i = 0; i < n; i++) CK(hipMemCreate(&h[i], chunk, &prop, 0));
        printf("hipMemCreate of %zu x 512 MiB right after that free: %.3f s\n", n, now() - t0);
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
        for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess(va, total, &acc, 1));
        CK(hipMemset(va, 1, total));
        CK(hipDeviceSynchronize());
        t0 = now();
        CK(hipMemUnmap(va, total));
        CK(hipMemAddressFree(va, total));
        void *vb = nullptr;
        CK(hipMemAddressReserve(&vb, total, 0, nullptr, 0));
        for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)vb + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess(vb, total, &acc, 1));
        const double t_cycle = now() - t0;
        void *small = nullptr;
        t0 = now();
        CK(hipMalloc(&small, 1 << 20));
        printf("unmap + free range + reserve + map + access of %zu GiB of kept chunks: %.3f s; a 1 MiB hipMalloc behind it: %.4f s\n", big_gib, t_cycle, now() - t0);
        CK(hipFree(small));
        CK(hipMemUnmap(vb, total));
        CK(hipMemAddressFree(vb, total));
        t0 = now();
        for (size_t i = 0; i < n; i++) CK(hipMemRelease(h[i]));
        const double t_rel = now() - t0;
        t0 = now();
        CK(hipMalloc(&small, 1 << 20));
        printf("release of the %zu chunks: %.3f s; a 1 MiB hipMalloc behind it: %.3f s\n", n, t_rel, now() - t0);
        CK(hipFree(small));
    }
    {   // first mapping of FRESH chunks (is the physical memory committed by hipMemCreate or by hipMemMap?), copies out of a mapped range
        const size_t chunk = (size_t)512 << 20, n = 64, total = n * chunk;
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        double t0 = now();
        for (size_t i = 0; i < n; i++) CK(hipMemCreate(&h[i], chunk, &prop, 0));
        const double t_create = now() - t0;
        size_t f1 = 0, tt = 0;
        CK(hipMemGetInfo(&f1, &tt));
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
        t0 = now();
        for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
        const double t_map = now() - t0;
        t0 = now();
        CK(hipMemSetAccess(va, total, &acc, 1));
        const double t_acc = now() - t0;
        size_t f2 = 0;
        CK(hipMemGetInfo(&f2, &tt));
        t0 = now();
        CK(hipMemset(va, 7, total));
        CK(hipDeviceSynchronize());
        const double t_set = now() - t0;
        printf("fresh 32 GiB in 512 MiB chunks: create %.4f s (free memory after it %.1f GiB of %.1f), map %.4f s, set access %.4f s (free after %.1f GiB), first memset %.4f s\n",
               t_create, f1 / 1073741824.0, tt / 1073741824.0, t_map, t_acc, f2 / 1073741824.0, t_set);
        std::vector<unsigned char> host((size_t)64 << 20);
        t0 = now();
        CK(hipMemcpy(host.data(), (char *)va + chunk - ((size_t)32 << 20), host.size(), hipMemcpyDeviceToHost));   // across a chunk boundary
        printf("hipMemcpy D2H of 64 MiB across a chunk boundary: %.4f s, bytes ok: %d\n", now() - t0, host[0] == 7 && host[host.size() - 1] == 7);
        CK(hipMemcpy2D(host.data(), 4096, (char *)va + chunk - 8192, 1 << 20, 4096, 64, hipMemcpyDeviceToHost));
        printf("hipMemcpy2D out of the mapped range: ok (%d)\n", host[4095] == 7);
        CK(hipMemUnmap(va, total));
        CK(hipMemAddressFree(va, total));
        for (size_t i = 0; i < n; i++) CK(hipMemRelease(h[i]));
    }
    {   // out of memory: which call reports it?
        const size_t chunk = (size_t)512 << 20;
        std::vector<hipMemGenericAllocationHandle_t> h;
        std::vector<void *> vas;
        hipError_t e = hipSuccess;
        const char *where = "none";
        for (size_t i = 0; i < 700; i++) {
            hipMemGenericAllocationHandle_t hh;
            e = hipMemCreate(&hh, chunk, &prop, 0);
            if (e != hipSuccess) { where = "hipMemCreate"; break; }
            h.push_back(hh);
            void *va = nullptr;
            e = hipMemAddressReserve(&va, chunk, 0, nullptr, 0);
            if (e != hipSuccess) { where = "hipMemAddressReserve"; break; }
            e = hipMemMap(va, chunk, 0, hh, 0);
            if (e != hipSuccess) { where = "hipMemMap"; hipMemAddressFree(va, chunk); break; }
            vas.push_back(va);
            e = hipMemSetAccess(va, chunk, &acc, 1);
            if (e != hipSuccess) { where = "hipMemSetAccess"; break; }
        }
        (void)hipGetLastError();
        printf("out of memory after %zu chunks of 512 MiB (%.1f GiB): %s says %s\n", vas.size(), vas.size() * 0.5, where, hipGetErrorString(e));
        for (void *va : vas) { hipMemUnmap(va, chunk); hipMemAddressFree(va, chunk); }
        for (auto hh : h) hipMemRelease(hh);
        size_t f = 0, tt = 0;
        CK(hipMemGetInfo(&f, &tt));
        printf("free memory after releasing them: %.1f GiB\n", f / 1073741824.0);
    }
    return 0;
}
