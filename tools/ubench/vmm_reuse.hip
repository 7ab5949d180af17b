// vmm_reuse.hip -- what happens to translations and to physical memory when an address range is unmapped and used again?  (diagnostic,
// not product; round 5: the context's arena in csrc/weldacs.hip is built on the answers)
//   T1  when does physical memory go back to the driver: after hipMemUnmap, after hipMemRelease, after hipMemAddressFree?
//   T2  a range is unmapped and mapped onto OTHER chunks: does a kernel see the new bytes or the old ones?  Variants: address range
//       freed and reserved again / kept reserved; old chunks released or kept; device synchronised; reading kernel on a new stream;
//       a kernel that sweeps 8 GiB of other memory in between
//   T3  does hipMemAddressReserve honour an address hint?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/vmm_reuse.hip -o build/vmm_reuse && build/vmm_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static const size_t CH = (size_t)512 << 20;

__global__ void k_fill(unsigned *p, size_t n, unsigned v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// how many of n words (one per 4 KiB page) equal v
__global__ void k_count(const unsigned *p, size_t pages, unsigned v, unsigned long long *out)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pages; i += (size_t)gridDim.x * blockDim.x) c += p[i * 1024] == v;
    atomicAdd(out, c);
}
__global__ void k_touch(const unsigned *p, size_t n, unsigned long long *out)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i];
    if (c == 0x123456789ULL) atomicAdd(out, 1ULL);
}

static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static unsigned long long *d_cnt;

static double free_gib()
{
    size_t f = 0, t = 0;
    hipMemGetInfo(&f, &t);
    return f / 1073741824.0;
}
static int map_chunks(void *va, const std::vector<hipMemGenericAllocationHandle_t> &h)
{
    for (size_t i = 0; i < h.size(); i++) CK(hipMemMap((char *)va + i * CH, CH, 0, h[i], 0));
    CK(hipMemSetAccess(va, h.size() * CH, &acc, 1));
    return 0;
}
static long long count_eq(void *va, size_t bytes, unsigned v, hipStream_t st)
{
    hipMemsetAsync(d_cnt, 0, 8, st);
    k_count<<<1024, 256, 0, st>>>((const unsigned *)va, bytes / 4096, v, d_cnt);
    unsigned long long c = 0;
    hipStreamSynchronize(st);
    if (hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)c;
}

int main()
{
    CK(hipSetDevice(0));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMalloc((void **)&d_cnt, 8));
    const size_t N = 8;   // chunks per set: 4 GiB
    printf("T1: when does physical memory return?  free now %.2f GiB\n", free_gib());
    for (int order = 0; order < 2; order++) {
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        for (auto &x : h) CK(hipMemCreate(&x, CH, &prop, 0));
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, N * CH, 0, nullptr, 0));
        if (map_chunks(va, h)) return 1;
        k_fill<<<4096, 256>>>((unsigned *)va, N * CH / 4, 1u);
        CK(hipDeviceSynchronize());
        printf("  order %d: mapped + touched %.2f", order, free_gib());
        CK(hipMemUnmap(va, N * CH));
        printf(" | unmapped %.2f", free_gib());
        if (order == 0) {
            for (auto &x : h) CK(hipMemRelease(x));
            printf(" | handles released %.2f", free_gib());
            CK(hipMemAddressFree(va, N * CH));
            printf(" | range freed %.2f GiB\n", free_gib());
        } else {
            CK(hipMemAddressFree(va, N * CH));
            printf(" | range freed %.2f", free_gib());
            for (auto &x : h) CK(hipMemRelease(x));
            printf(" | handles released %.2f GiB\n", free_gib());
        }
    }
    printf("T2: unmap, then map OTHER chunks at the same address: new bytes (ok) or old bytes (STALE)?\n");
    hipStream_t s2;
    for (int variant = 0; variant < 8; variant++) {
        // bit 0: the range is freed and reserved again (else kept reserved); bit 1: the old chunks' handles are released before the new map;
        // bit 2: hipDeviceSynchronize between; variants 6 / 7: as 0 / 1 with the reading kernel on a NEW stream and an 8 GiB sweep of other memory in between
        const bool refree = variant & 1, release_old = variant & 2, devsync = (variant & 4) && variant < 6, extra = variant >= 6;
        std::vector<hipMemGenericAllocationHandle_t> A(N), B(N);
        for (auto &x : A) CK(hipMemCreate(&x, CH, &prop, 0));
        for (auto &x : B) CK(hipMemCreate(&x, CH, &prop, 0));
        void *va = nullptr, *vb = nullptr;
        CK(hipMemAddressReserve(&va, N * CH, 0, nullptr, 0));
        CK(hipMemAddressReserve(&vb, N * CH, 0, nullptr, 0));
        if (map_chunks(va, A) || map_chunks(vb, B)) return 1;
        k_fill<<<4096, 256>>>((unsigned *)va, N * CH / 4, 0xAAAAAAAAu);
        k_fill<<<4096, 256>>>((unsigned *)vb, N * CH / 4, 0xBBBBBBBBu);
        CK(hipDeviceSynchronize());
        const long long a0 = count_eq(va, N * CH, 0xAAAAAAAAu, 0);      // (the translations of va are cached now)
        CK(hipMemUnmap(va, N * CH));
        CK(hipMemUnmap(vb, N * CH));
        if (release_old) for (auto &x : A) CK(hipMemRelease(x));
        void *va2 = va;
        if (refree) {
            CK(hipMemAddressFree(va, N * CH));
            CK(hipMemAddressReserve(&va2, N * CH, 0, va, 0));
        }
        if (devsync) CK(hipDeviceSynchronize());
        if (map_chunks(va2, B)) return 1;       // B's bytes at (hopefully) the address A had
        hipStream_t st = 0;
        if (extra) {
            CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            st = s2;
            unsigned *other = nullptr;
            CK(hipMalloc((void **)&other, (size_t)8 << 30));
            k_touch<<<8192, 256, 0, st>>>(other, ((size_t)8 << 30) / 4, d_cnt);
            CK(hipStreamSynchronize(st));
            CK(hipFree(other));
        }
        const long long nb = count_eq(va2, N * CH, 0xBBBBBBBBu, st), na = count_eq(va2, N * CH, 0xAAAAAAAAu, st);
        printf("  variant %d (%s, old chunks %s%s%s): same address %s; of %zu pages %lld read the NEW bytes, %lld the OLD ones%s (before: %lld old)\n", variant,
               refree ? "range freed + reserved again" : "range kept reserved", release_old ? "released" : "kept", devsync ? ", device synchronised" : "",
               extra ? ", new stream + 8 GiB sweep" : "", va2 == va ? "yes" : "NO", N * CH / 4096, nb, na, na ? "  <-- STALE" : "", a0);
        CK(hipMemUnmap(va2, N * CH));
        CK(hipMemAddressFree(va2, N * CH));
        CK(hipMemAddressFree(vb, N * CH));
        if (!release_old) for (auto &x : A) CK(hipMemRelease(x));
        for (auto &x : B) CK(hipMemRelease(x));
        if (extra) CK(hipStreamDestroy(s2));
    }
    printf("T3: address hints\n");
    {
        void *a = nullptr, *b = nullptr;
        CK(hipMemAddressReserve(&a, CH, 0, nullptr, 0));
        void *hint = (char *)a - ((size_t)1 << 40);   // 1 TiB below whatever the allocator hands out
        hipError_t e = hipMemAddressReserve(&b, CH, 0, hint, 0);
        printf("  default %p; hint %p -> %s %p (%s)\n", a, hint, hipGetErrorString(e), b, b == hint ? "honoured" : "not honoured");
        if (e == hipSuccess) hipMemAddressFree(b, CH);
        hipMemAddressFree(a, CH);
    }
    return 0;
}
