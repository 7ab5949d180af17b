// weldacs.hip -- C ABI of libweldacs.so (include/weldacs.h) over the gfx950 kernels.
// Single translation unit: hipcc --offload-arch=gfx950 -ffp-contract=off (see build.py).
// There is no CPU compute path in this library: without a HIP device wa_ctx_create fails.
#include "../../include/weldacs.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <float.h>
#include <locale.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <string>
#include <vector>

#include "acs_kernels.hpp"
#include "grid_kernels.hpp"
#include "gtsp_kernels.hpp"
#include "traj_kernels.hpp"
#include "stl_text.hpp"

// ------------------------------------------------------------------ handles
struct WaDevBlock { void *p; size_t bytes; unsigned long long stamp; };
// a block handed out by the arena: a reserved virtual range with physical chunks mapped into it (see "the contexts' memory" below)
struct WaArenaBlock { void *va; size_t va_bytes; std::vector<hipMemGenericAllocationHandle_t> chunks[3]; unsigned long long stamp; };
struct wa_ctx {
    int device;
    hipStream_t stream;    // every kernel of this context
    std::string err;
    hipDeviceProp_t prop;
    bool lds_attr_set = false;   // dynamic-LDS limit of the walk kernels raised on this device (wa_acs_run)
    // Device blocks of destroyed solvers, kept for the next solver of this context (see ctx_alloc below): the driver wipes device memory
    // that is given back, in the background at ~36 GB/s (MI355X, ROCm 7.2), and the next allocation of any size waits until ALL of it
    // is clean -- 5.3 s after the 190 GB of a C5-sized solver were freed, against 0.1 s to clear and initialise them
    // (profiles/r04/create_time.txt, alloc_after_free.txt).
    std::vector<WaDevBlock> cache;             // whole hipMalloc blocks below the arena's threshold (or all, without the arena): exact-fit reuse, bounded
    std::unordered_map<void *, size_t> live;   // blocks handed out by ctx_alloc from hipMalloc and not yet returned
    size_t cache_bytes = 0;                    // kept bytes: cached whole blocks + pooled chunks
    bool cache_on = true, poison = false;      // WA_DEV_CACHE=0 / WA_DEV_POISON=1, read at wa_ctx_create
    // the arena (round 5): physical chunks of two sizes, created once and mapped into whatever virtual range the next block needs
    bool arena_on = false;                     // the device supports virtual memory management and WA_DEV_ARENA != 0
    bool arena_strict = false;       // kept blocks are only reused for requests of exactly their size (the retry of a solver that did not fit)
    std::vector<hipMemGenericAllocationHandle_t> pool[3];   // kept chunks per size class (WA_ARENA_SZ), unmapped
    std::unordered_map<void *, WaArenaBlock> arena_live;    // blocks handed out
    std::vector<WaArenaBlock> arena_kept;                   // blocks given back, STILL MAPPED: a request of exactly that size takes one as it is

    size_t keep_limit = 0;                     // bytes the context may keep (WA_DEV_KEEP_FRAC of the device's memory)
    size_t small_cache_limit = 0;              // ... of which in whole cached blocks (WA_DEV_SMALL_CACHE_MB)
    unsigned long long clock = 0;              // LRU stamps of the cached whole blocks
    int64_t stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // wa_ctx_cache_stats
};
struct wa_grid {
    wa_ctx *ctx;
    WaDims d;
    float precision;
    int32_t wall;
    int64_t n_free;
    float *cx, *cy, *cz;   // device
    uint8_t *occ;          // device
    // host mirror of the axis tables and the occupancy for calls that resolve a few points (wa_grid_resolve_points: ACS_Rank::setPoints
    // resolves TWO): fetched on the first such call -- a grid does not change after it has been built
    mutable std::vector<float> h_cx, h_cy, h_cz;
    mutable std::vector<uint8_t> h_occ;
    mutable bool h_valid = false;
    mutable std::mutex h_mu;   // guards the mirror's first fill (concurrent wa_grid_resolve_points calls on one grid)
};
struct wa_traj {
    wa_ctx *ctx;
    int64_t n;
    float *xyz;            // device, n x 3
};
struct wa_bspline {
    wa_ctx *ctx;
    WaSpline S;            // knots / cps on the device
    float *d_ends;         // init rows + fin rows staged for k_bspline_setup
    bool set;
    // host mirror of knots / control points for single-point calls (wa_bspline_eval_host): fetched on the first such call after SetParam
    std::vector<float> h_knots, h_cps;
    bool h_valid = false;
};
struct EvPair { hipEvent_t a, b; int cls; };
struct wa_acs {
    wa_ctx *ctx;
    const wa_grid *grid;
    int32_t n_slots, max_colony, n_active, nb;
    int64_t path_cap;
    WaAcsDev D;            // D.pher always points at the CURRENT pheromone buffer
    int32_t *d_stage = nullptr;      // staging block of acs_fetch_results (the best paths of all slots, packed), grows on demand
    size_t stage_words = 0;
    // host copy of every slot's control block and best path behind the last run (acs_fetch_results): wa_acs_result / _batch read it
    std::vector<float> ltab_host;    // host copy of D.ltab (uploaded asynchronously at creation)
    std::vector<WaSlotCtl> res_ctl;
    std::vector<int32_t> res_words;
    int64_t res_longest = 0;
    bool res_valid = false;
    int32_t *paths_arr[2] = {nullptr, nullptr};   // the ants' paths: one array ([1] == [0]), or two alternating by generation when stragglers are handed over
    float *pher_buf[2];    // double buffer: evaporation writes the other one (dst = src * rho)
    float *pher_alloc[2], *heur_alloc;   // the allocations behind pher_buf[] / D.heur (fields + guard bands)
    uint32_t *stamp_alloc;               // ... and D.stamp (lazy solvers)
    int cur_buf;
    std::vector<int> slot_buf;           // which of the two buffers holds slot q's current field (inactive slots do not follow the flips)
    bool walk_asm;         // hand-scheduled walk loop (default); WA_WALK_ASM=0 keeps the compiler-scheduled one
    int walk_warm;         // touch loads in the hand-scheduled loop: -1 by launch size (wa_acs_run), 0 / 1 forced (WA_WALK_WARM)
    int32_t last_walk[4];  // wa_acs_walk_info
#ifdef WA_STATE_HASH
    unsigned long long *d_hashlog = nullptr;   // [WA_HASH_GENS][3][n_slots][8], diagnostic build only (k_state_hash)
#endif
    int tab16_env, id_bits; // WA_TAB16 (-1 by rule, 0 never, 1 wherever possible); bits of the grid's voxel ids
    int lds_pad;           // WA_WALK_LDS_PAD: experiment knob, extra dynamic LDS per walk block (occupancy at a constant table)
    int walk_direct;       // the loop WITHOUT look-ahead for saturated launches: -1 by rule (walk_direct_rule), 0 / 1 forced (WA_WALK_DIRECT)
    int walk_flags;        // k_walk_dev's switches: hand-scheduled loop, re-entry onto the replay track (WA_REENTRY=0: off), see acs_create
    WaRun R;
    bool begun;
    bool lazy;                          // lazy evaporation (wa_acs_create_lazy): never-deposited voxels are not swept
    std::vector<int> lazy_mode;         // per slot: init mode of the stored records (-1 unknown)
    std::vector<float> lazy_p0;
    int32_t gens_enqueued, colony_bound, hash_log2, evap_blocks, lazy_blocks_env, straggler_gens = 64;
    long long *d_starts, *d_ends;
    uint32_t *d_streams;
    int32_t *d_hslot, *d_hlist, *d_hends;   // per search: the heuristic field it reads / the fields wa_acs_begin computes and their end points
    int32_t heur_fields;                // fields in the pool (heur_alloc): grows on demand, at most one per slot
    size_t heur_guard;                  // floats of guard band around the pool
    std::vector<long long> heur_end;    // per field: the end point it holds (-1: none), the beta it was computed with,
    std::vector<float> heur_beta;       // ... and the last wa_acs_begin that used it (oldest goes first)
    std::vector<long long> heur_used;
    long long heur_batch;
    // profiling
    bool prof, prof_sweep_all;   // prof_sweep_all: the sweep-carrying launch of EVERY generation carries its own start/stop events
    bool prof_paired;            // a stamped no-op dispatch in front of every timed sweep-carrying launch (prof_pair_marker)
    int32_t prof_every;
    std::vector<EvPair> ev;
    double prof_ms[WA_K_COUNT];
    int64_t prof_n[WA_K_COUNT];
    // pipelined groups (wa_acs_run): the active slots split into groups, each with a stream of its own, so that one group's HBM-bound
    // sweep runs under another group's latency-bound walk.  Forked from / joined into the context's stream inside every wa_acs_run call.
    int32_t pipe_groups_env = 0;         // WA_PIPE_GROUPS, read at creation (0: by rule)
    std::vector<hipStream_t> gstream;
    std::vector<hipEvent_t> gjoin;
    int32_t sweep_nt_env = -1;           // WA_SWEEP_NT: cache policy of the sweep (-1: by rule)
    hipEvent_t gfork = nullptr;
    int32_t last_groups = 1;             // groups the last wa_acs_run call used (wa_acs_pipeline_info)
    bool ref_spec = true;                // WA_REF_SPEC: REF mode speculates converged generations (k_ref_draws / k_walk_ref_spec)
    bool drain_ok = true;                // WA_STRAGGLER_DRAIN: the last generation of a call may hand over too; whoever reads results first drains
    bool pending_resume = false;         // ... its stragglers sit in their pool: the next walk launch (or a drain launch) finishes them
    // The last generation of a call only hands over when the caller has shown that calls follow each other without a read in between
    // (chunked runs, generation-by-generation loops): behind a lone call the drain launch would only add to what the caller waits for
    // (measured on bench.py --steps 20: -1 %).  ran_before / read_since_run track that pattern.
    bool ran_before = false, read_since_run = false, chained = false;
};

static int fail(wa_ctx *c, int code, const char *fmt, const char *a = "")
{
    if (c) {
        char buf[512];
        snprintf(buf, sizeof buf, fmt, a);
        c->err = buf;
    }
    return code;
}
#define HIPC(ctx, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail((ctx), WA_ERR_DEVICE, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

// Every entry point runs on the device of its context, whatever device the calling thread had current (another
// context on another GPU, torch.cuda.set_device ...), and leaves the caller's current device as it found it.
struct WaDevGuard {
    int prev = -1;
    bool switched = false, ok = true;
    explicit WaDevGuard(const wa_ctx *c)
    {
        if (!c) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) {
            ok = hipSetDevice(c->device) == hipSuccess;
            switched = ok && prev >= 0;
        }
    }
    ~WaDevGuard() { if (switched) hipSetDevice(prev); }
    WaDevGuard(const WaDevGuard &) = delete;
    WaDevGuard &operator=(const WaDevGuard &) = delete;
};

static int env_int(const char *name, int def)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : def;
}

// ------------------------------------------------------------------ the contexts' memory: an arena of physical chunks + a small block cache
// Solvers take their device memory through ctx_alloc / ctx_free.  What a solver gives back STAYS with the context: the driver wipes
// device memory a process has used -- when it is released, and before the pages are handed out again -- at ~25-40 ms per GB (MI355X,
// ROCm 7.2; untouched memory of a fresh box comes at once: profiles/r05/vmm_threads.txt), and the next allocation of
// any size -- in this process or the next -- waits until all of it is clean: 3.9 s behind 64 GiB, 5.3 s behind the 190 GB of a C5-sized
// solver, against the 0.1 s it takes to clear and initialise them (profiles/r04/alloc_after_free.txt, profiles/r05/vmm_probe.txt).
//
// Round 4 kept whole hipMalloc blocks and reused one only for a request of (nearly) its size: solvers of DIFFERENT shapes in sequence --
// the 1-, 2-, ... 32-slot solvers of a benchmark, the drop-in's 1-slot then N-slot solver -- piled up blocks nobody could use, and a big
// solver behind them ran out of memory beside ~100 GB of them.  A virtual address range must be contiguous; physical memory
// need not be.  So, where the device supports virtual memory management (hipMemCreate / hipMemAddressReserve / hipMemMap: it does on
// MI355X), blocks of WA_ARENA_MIN bytes and up are built from CHUNKS: physical allocations of 512 MiB, 32 MiB and 2 MiB (a block = as
// many big ones as fit, then its tail in the smaller classes: at most 15 + 15 of them, rounded up to 2 MiB like the driver's own granule)
// mapped into a reserved address range.
//   * A block that is given back stays MAPPED (arena_kept): a request of its size, or smaller by up to a third, takes it as it stands --
//     the same or a similar solver shape created again maps nothing (arena_alloc_locked says why the surplus is worth it).
//   * A request no kept block fits takes chunks from the pools; when those run short, kept blocks are HARVESTED, least recently used
//     first: unmapped, their chunks to the pools -- so a solver of ANY shape is built from what solvers of other shapes gave back, and
//     only the difference is created fresh.
//   * An address range that has been unmapped is NEVER MAPPED AGAIN.  Measured in round 5 (tools/ubench/vmm_reuse.hip,
//     profiles/r05/vmm_reuse.txt): a kernel that reads through a range which was unmapped and then mapped onto OTHER chunks still gets
//     the OLD chunks' bytes on this stack (MI355X, ROCm 7.2) -- whether the range was kept reserved or freed and reserved again, whether
//     the old chunks were released or kept, with hipDeviceSynchronize in between or not; every page of 4 GiB, not a few -- while a range
//     nobody was mapped at before is fine.  (Found the hard way: the second solver of a context walked on the first solver's bytes, left
//     its field and faulted.)  Every new block therefore gets an address range from a cursor that only moves up (va_reserve_fresh_locked:
//     hipMemAddressReserve honours address hints), and harvested ranges are given back to the address-space allocator at once -- which
//     is also what lets physical memory return to the driver: released chunks only do once the ranges they were mapped at are freed.
// Measured (profiles/r05/vmm_probe.txt): mapping ~5 us per chunk, unmap + reserve + map + set-access of 64 GiB 3 ms, the evaporation
// sweep's access pattern at the same rate on mapped memory as on hipMalloc memory, 1-D host copies across chunk boundaries fine (the
// runtime's 2-D copy is not: wa_acs_result_batch gathers on the device instead), out of memory reported by hipMemCreate.  Chunks are
// fungible: under memory pressure (a chunk cannot be created, or a plain hipMalloc of this library fails) kept memory goes back to the
// driver -- pooled chunks of the size class NOT being asked for first, then kept blocks, least recently used first -- and only as much
// as is needed, not all.  Blocks below WA_ARENA_MIN (and every block where the arena is off: WA_DEV_ARENA=0 or no VMM support) keep
// round 4's exact-fit cache of whole hipMalloc blocks, now bounded: least recently used blocks are released beyond
// WA_DEV_SMALL_CACHE_MB (2 048; without the arena WA_DEV_KEEP_PCT).  Nothing is assumed about a block's contents, fresh or reused
// (WA_DEV_POISON=1 fills every block with 0xff bytes before it is handed out: the GPU suite passes that way).  Kept bytes count as free
// in wa_ctx_memory_info; wa_ctx_trim releases them, WA_DEV_CACHE=0 switches everything off.  wa_ctx_cache_stats reports what was served
// from kept memory and what had to be created.
static std::mutex g_cache_mu;
static std::vector<wa_ctx *> g_cache_ctxs;
static const size_t WA_CACHE_MIN_BYTES = (size_t)1 << 20;
static const size_t WA_ARENA_SZ[3] = {(size_t)512 << 20, (size_t)32 << 20, (size_t)2 << 20}, WA_ARENA_MIN = (size_t)2 << 20;
enum { WA_ST_ALLOCS = 0, WA_ST_HIT_BYTES, WA_ST_MISS_BYTES, WA_ST_RELEASED_BYTES, WA_ST_FULL_HITS, WA_ST_OOM_EVENTS, WA_ST_WAIT_MS };

static hipMemAllocationProp arena_prop(int device)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    return prop;
}

// A fresh address range for a block: one that no block of this process has ever been mapped at (see above).  hipMemAddressReserve
// honours an address hint (measured), so the ranges come from a cursor that only moves up through a region of the address space nothing
// else uses (32 TiB .. 96 TiB: the host heap sits near 0x55.., mmap and the runtime's own allocations near 0x7f..); a range that was
// given back to the address-space allocator can then never be handed out again.  When a hint is not honoured the cursor skips ahead and
// tries again; when the region is used up (64 TiB of blocks re-mapped: hundreds of C5-sized reshapes) the arena stops building blocks
// and the caller falls back to whole hipMalloc blocks.
static const uintptr_t WA_VA_FIRST = (uintptr_t)32 << 40, WA_VA_END = (uintptr_t)96 << 40;
static uintptr_t g_va_cursor = WA_VA_FIRST, g_va_end = WA_VA_END;
static bool g_va_exhausted = false;   // the window is used up, or the platform does not honour address hints: no new arena blocks in this process
static bool g_va_nohint = false;      // WA_DEV_ARENA_NOHINT=1 (test knob): reserve without a hint, i.e. what a platform that ignores hints would do
// The cursor advances by what a block takes (rounded to the 2 MiB granule by the caller) plus a 2 MiB gap, aligned like the largest chunk
// class the block holds -- 1 GiB for blocks of 512 MiB and up, 32 MiB, 2 MiB -- so that the translation of a chunk is not split by its
// address (round 5 advanced by whole GiB: a small solver's forty 2-50 MiB blocks used up the window as fast as a C5-sized solver's).
static hipError_t va_reserve_fresh_locked(size_t bytes, void **out)
{
    *out = nullptr;
    if (g_va_exhausted) return hipErrorOutOfMemory;
    const uintptr_t align = bytes >= WA_ARENA_SZ[0] ? (uintptr_t)1 << 30 : bytes >= WA_ARENA_SZ[1] ? (uintptr_t)WA_ARENA_SZ[1] : (uintptr_t)WA_ARENA_SZ[2];
    for (int tries = 0; tries < 8; tries++) {
        const uintptr_t start = (g_va_cursor + align - 1) & ~(align - 1);
        if (start + bytes > g_va_end || start + bytes < start) { g_va_exhausted = true; return hipErrorOutOfMemory; }
        void *hint = (void *)start, *va = nullptr;
        const hipError_t e = hipMemAddressReserve(&va, bytes, 0, g_va_nohint ? nullptr : hint, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); g_va_cursor = start + ((uintptr_t)1 << 40); continue; }
        if (va == hint) {
            g_va_cursor = start + bytes + (uintptr_t)WA_ARENA_SZ[2];   // (a 2 MiB gap: ranges never touch)
            *out = va;
            return hipSuccess;
        }
        hipMemAddressFree(va, bytes);              // somewhere else: not known to be fresh
        g_va_cursor = start + ((uintptr_t)1 << 40);
    }
    g_va_exhausted = true;   // eight hints in a row refused or answered elsewhere: this platform does not place ranges where they are asked for
    return hipErrorOutOfMemory;
}
// a kept (mapped, idle) block gives its chunks to the pools; its address range is given back and never handed out again
static void arena_harvest_locked(wa_ctx *c, size_t idx)
{
    WaArenaBlock b = std::move(c->arena_kept[idx]);
    c->arena_kept.erase(c->arena_kept.begin() + (long)idx);
    hipMemUnmap(b.va, b.va_bytes);
    hipMemAddressFree(b.va, b.va_bytes);   // (physical memory only returns to the driver once the ranges it was mapped at are freed: measured)
    for (int k = 0; k < 3; k++)
        for (auto h : b.chunks[k]) c->pool[k].push_back(h);      // (cache_bytes unchanged: kept blocks and pooled chunks both count)
}
static size_t arena_oldest_kept(const wa_ctx *c)
{
    size_t best = 0;
    for (size_t i = 1; i < c->arena_kept.size(); i++)
        if (c->arena_kept[i].stamp < c->arena_kept[best].stamp) best = i;
    return best;
}

// kept memory of one context -> the driver, at least `need` bytes of it if there is that much (need = 0: everything).  Pooled chunks of
// the size classes that are NOT wanted go first (the wanted class's pool is empty on the asking context), then kept blocks, least
// recently used first, then cached whole blocks, least recently used first.
static size_t cache_release_locked(wa_ctx *c, size_t need = 0, int spare_class = -1)
{
    size_t freed = 0;
    auto enough = [&]() { return need != 0 && freed >= need; };
    auto drain_pools = [&]() {
        for (int pass = 0; pass < 4 && !enough(); pass++) {
            const int k = pass < 3 ? pass : spare_class;
            if (k < 0 || (pass < 3 && k == spare_class)) continue;
            while (!c->pool[k].empty() && !enough()) {
                hipMemRelease(c->pool[k].back());
                c->pool[k].pop_back();
                freed += WA_ARENA_SZ[k];
            }
        }
    };
    drain_pools();
    while (!c->arena_kept.empty() && !enough()) {
        arena_harvest_locked(c, arena_oldest_kept(c));
        drain_pools();
    }
    std::sort(c->cache.begin(), c->cache.end(), [](const WaDevBlock &a, const WaDevBlock &b) { return a.stamp > b.stamp; });   // oldest last
    while (!c->cache.empty() && !enough()) {
        hipFree(c->cache.back().p);
        freed += c->cache.back().bytes;
        c->cache.pop_back();
    }
    c->cache_bytes -= freed < c->cache_bytes ? freed : c->cache_bytes;
    c->stat[WA_ST_RELEASED_BYTES] += (int64_t)freed;
    return freed;
}
static size_t cache_release_device_locked(int device, size_t need = 0, int spare_class = -1)
{
    size_t freed = 0;
    for (wa_ctx *o : g_cache_ctxs)
        if (o->device == device && (need == 0 || freed < need)) freed += cache_release_locked(o, need ? need - freed : 0, spare_class);
    return freed;
}
static size_t cache_release_device(int device, size_t need = 0)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    return cache_release_device_locked(device, need);
}

// plain hipMalloc for everything that does not go through a context's cache (grids, traces, staging, the communicator's scratch): when
// the device is out of memory the kept memory of the contexts on it is released -- as much as is needed -- and the call retried while
// the driver's wipe proceeds, for a few seconds at most
static hipError_t dev_malloc(void **out, size_t bytes)
{
    if (bytes < 16) bytes = 16;
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return e;
    // rounds: what is needed (+ slack), twice that, four times, everything.  After each release the call is retried while the driver
    // wipes what it got back (memory in that state is neither free nor allocatable: hipMalloc fails rather than waits), 2 s at most
    // per round; a device with nothing kept on it gives up after 0.3 s
    size_t need = bytes + ((size_t)64 << 20);
    for (int round = 0; round < 4; round++) {
        const size_t freed = cache_release_device(dev, round < 3 ? need : 0);
        const auto t0 = std::chrono::steady_clock::now();
        const double limit = freed ? 2.0 : 0.3;
        for (;;) {
            e = hipMalloc(out, bytes);
            if (e != hipErrorOutOfMemory) break;
            (void)hipGetLastError();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (e != hipErrorOutOfMemory || freed == 0) break;
        need *= 2;
    }
    if (e != hipSuccess) *out = nullptr;
    return e;
}
template <class T>
static hipError_t dalloc(T **p, size_t count)
{
    return dev_malloc((void **)p, count * sizeof(T));
}

// one chunk of class k for a block under construction: from the pool; else from a kept block that nobody asked for (harvested, least
// recently used first); else created -- pooled chunks of the other classes and cached whole blocks make room when the device is full
static hipError_t arena_chunk(wa_ctx *c, int k, hipMemGenericAllocationHandle_t *h, bool *kept)
{
    const size_t sz = WA_ARENA_SZ[k];
    while (c->pool[k].empty() && !c->arena_kept.empty()) {
        // (a kept block may hold no chunk of this class: harvesting goes on until one turns up or nothing is kept any more)
        arena_harvest_locked(c, arena_oldest_kept(c));
    }
    if (!c->pool[k].empty()) {
        *h = c->pool[k].back();
        c->pool[k].pop_back();
        c->cache_bytes -= sz;
        *kept = true;
        return hipSuccess;
    }
    *kept = false;
    const hipMemAllocationProp prop = arena_prop(c->device);
    hipError_t e = hipMemCreate(h, sz, &prop, 0);
    for (int tries = 0; e == hipErrorOutOfMemory && tries < 3; tries++) {
        (void)hipGetLastError();
        c->stat[WA_ST_OOM_EVENTS]++;
        // tries 0: enough of the classes that are not wanted (and of the whole blocks); 1: twice that; 2: everything on the device
        if (cache_release_device_locked(c->device, tries < 2 ? (sz << tries) + ((size_t)64 << 20) : 0, k) == 0 && tries > 0) break;
        e = hipMemCreate(h, sz, &prop, 0);   // (waits for the driver's wipe of what was just released: ~30 ms per GB)
    }
    if (e == hipErrorOutOfMemory) {
        // memory that somebody released a moment ago -- another context of this process, the process before this one -- is neither free
        // nor allocatable while the driver wipes it: retry while the free figure still moves, a few seconds at most
        const auto t0 = std::chrono::steady_clock::now();
        size_t last_free = 0;
        double t_change = 0;
        for (;;) {
            (void)hipGetLastError();
            const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            size_t f = 0, tot = 0;
            if (hipMemGetInfo(&f, &tot) != hipSuccess) break;
            if (f != last_free) { last_free = f; t_change = t; }
            if (t > 8.0 || t - t_change > 1.0) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            e = hipMemCreate(h, sz, &prop, 0);
            if (e != hipErrorOutOfMemory) break;
        }
    }
    return e;
}
static void arena_return_chunks(wa_ctx *c, WaArenaBlock &b)
{
    for (int k = 0; k < 3; k++) {
        for (auto h : b.chunks[k]) { c->pool[k].push_back(h); c->cache_bytes += WA_ARENA_SZ[k]; }
        b.chunks[k].clear();
    }
}
static hipError_t arena_alloc_locked(wa_ctx *c, void **out, size_t bytes)
{
    const size_t rounded = (bytes + WA_ARENA_SZ[2] - 1) / WA_ARENA_SZ[2] * WA_ARENA_SZ[2];
    {   // a kept block of this size, or the smallest one that is larger by up to half: taken as it stands, surplus included (of equals the
        // most recently used first: its bytes are the likeliest to be in a cache).  The surplus is what keeps a process that builds solver
        // after solver of SIMILAR shapes (224, 190, 207 ... slots) from re-mapping 190 GB into fresh address ranges every time: ranges
        // are never used twice (above), so tools/arena_stress.py ran out of them after ~380 such solvers and every later one was built from
        // plain hipMalloc blocks in 4.9 s instead of 0.09 s.  All arrays of a solver shrink and grow together, so what the larger blocks
        // hold back is memory the context was keeping anyway; a request that cannot be met because of it is retried with exact sizes
        // (arena_strict: acs_create).
        long best = -1;
        for (size_t i = 0; i < c->arena_kept.size(); i++) {
            const size_t vb = c->arena_kept[i].va_bytes;
            if (vb < rounded || (vb != rounded && (c->arena_strict || vb - rounded > rounded / 2))) continue;
            if (best < 0 || vb < c->arena_kept[(size_t)best].va_bytes ||
                (vb == c->arena_kept[(size_t)best].va_bytes && c->arena_kept[i].stamp > c->arena_kept[(size_t)best].stamp)) best = (long)i;
        }
        if (best >= 0) {
            WaArenaBlock b = std::move(c->arena_kept[(size_t)best]);
            c->arena_kept.erase(c->arena_kept.begin() + best);
            c->cache_bytes -= b.va_bytes;
            c->stat[WA_ST_HIT_BYTES] += (int64_t)rounded;
            c->stat[WA_ST_FULL_HITS]++;
            *out = b.va;
            c->arena_live.emplace(b.va, std::move(b));
            return hipSuccess;
        }
    }
    // (the address range first: when the window is used up no chunk is taken from the pools or created only to be put back -- ADVICE r05)
    void *va = nullptr;
    hipError_t e = c->arena_on ? va_reserve_fresh_locked(rounded, &va) : hipErrorOutOfMemory;   // a range nobody has been mapped at
    if (e != hipSuccess) { *out = nullptr; return e; }
    size_t cnt[3], rest = rounded;
    for (int k = 0; k < 3; k++) { cnt[k] = rest / WA_ARENA_SZ[k]; rest %= WA_ARENA_SZ[k]; }
    WaArenaBlock b;
    b.va = nullptr;
    b.va_bytes = rounded;
    b.stamp = 0;
    size_t from_kept = 0;
    for (int k = 0; k < 3; k++)
        for (size_t i = 0; i < cnt[k] && e == hipSuccess; i++) {
            hipMemGenericAllocationHandle_t h;
            bool kept = false;
            e = arena_chunk(c, k, &h, &kept);
            if (e == hipSuccess) {
                b.chunks[k].push_back(h);
                if (kept) from_kept += WA_ARENA_SZ[k];
            }
        }
    size_t mapped = 0;
    if (e == hipSuccess) {
        for (int k = 0; k < 3; k++)
            for (size_t i = 0; i < b.chunks[k].size() && e == hipSuccess; i++) {
                e = hipMemMap((char *)va + mapped, WA_ARENA_SZ[k], 0, b.chunks[k][i], 0);
                if (e == hipSuccess) mapped += WA_ARENA_SZ[k];
            }
        if (e == hipSuccess) {
            hipMemAccessDesc acc = {};
            acc.location.type = hipMemLocationTypeDevice;
            acc.location.id = c->device;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(va, rounded, &acc, 1);
        }
    }
    if (e != hipSuccess) {   // nothing of a half-built block survives; its chunks stay with the context, its range is not used again
        (void)hipGetLastError();
        if (va) {
            if (mapped) hipMemUnmap(va, mapped);
            hipMemAddressFree(va, rounded);
        }
        arena_return_chunks(c, b);
        *out = nullptr;
        return e;
    }
    c->stat[WA_ST_HIT_BYTES] += (int64_t)from_kept;
    c->stat[WA_ST_MISS_BYTES] += (int64_t)(rounded - from_kept);
    if (from_kept == rounded) c->stat[WA_ST_FULL_HITS]++;
    b.va = va;
    c->arena_live.emplace(va, std::move(b));
    *out = va;
    return hipSuccess;
}
// the context may keep keep_limit bytes: beyond it kept memory goes back to the driver, whole cached blocks first
static void cache_enforce_limits_locked(wa_ctx *c)
{
    size_t whole = 0;
    for (auto &b : c->cache) whole += b.bytes;
    // (once the arena has stopped building blocks -- address window used up -- whole blocks are what is kept, up to the context's limit)
    const size_t small_limit = (c->arena_on && !g_va_exhausted) ? c->small_cache_limit : c->keep_limit;
    if (whole > small_limit) {
        std::sort(c->cache.begin(), c->cache.end(), [](const WaDevBlock &a, const WaDevBlock &b) { return a.stamp > b.stamp; });
        while (!c->cache.empty() && whole > small_limit) {
            hipFree(c->cache.back().p);
            whole -= c->cache.back().bytes;
            c->cache_bytes -= c->cache.back().bytes;
            c->stat[WA_ST_RELEASED_BYTES] += (int64_t)c->cache.back().bytes;
            c->cache.pop_back();
        }
    }
    if (c->cache_bytes > c->keep_limit) cache_release_locked(c, c->cache_bytes - c->keep_limit);
}
static hipError_t ctx_alloc_bytes(wa_ctx *c, void **out, size_t bytes)
{
    if (bytes < 16) bytes = 16;
    *out = nullptr;
    const bool big = c->cache_on && bytes >= WA_CACHE_MIN_BYTES;
    if (big) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        c->stat[WA_ST_ALLOCS]++;
        if ((c->arena_on || !c->arena_kept.empty()) && bytes >= WA_ARENA_MIN) {   // (kept blocks go on serving when no new ones are built)
            const auto t0 = std::chrono::steady_clock::now();
            const hipError_t e = arena_alloc_locked(c, out, bytes);
            c->stat[WA_ST_WAIT_MS] += (int64_t)(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            if (e == hipSuccess) {
                if (c->poison) return hipMemsetAsync(*out, 0xff, bytes, c->stream);
                return hipSuccess;
            }
            *out = nullptr;   // (no chunk, or no fresh address range left: a whole hipMalloc block below, if the device has one)
        }
        int best = -1;
        for (int i = 0; i < (int)c->cache.size(); i++) {
            const size_t b = c->cache[i].bytes;
            if (b >= bytes && b - bytes <= bytes / 8 && (best < 0 || b < c->cache[best].bytes)) best = i;
        }
        if (best >= 0) {
            *out = c->cache[best].p;
            c->live[*out] = c->cache[best].bytes;
            c->cache_bytes -= c->cache[best].bytes;
            c->stat[WA_ST_HIT_BYTES] += (int64_t)c->cache[best].bytes;
            c->stat[WA_ST_FULL_HITS]++;
            c->cache.erase(c->cache.begin() + best);
        }
    }
    if (!*out) {
        const hipError_t e = dev_malloc(out, bytes);   // (releases kept memory and retries when the device is full)
        if (e != hipSuccess) { *out = nullptr; return e; }
        if (big) {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            c->live[*out] = bytes;
            c->stat[WA_ST_MISS_BYTES] += (int64_t)bytes;
        }
    }
    if (c->poison) return hipMemsetAsync(*out, 0xff, bytes, c->stream);
    return hipSuccess;
}
template <class T>
static hipError_t ctx_alloc(wa_ctx *c, T **p, size_t count)
{
    return ctx_alloc_bytes(c, (void **)p, count * sizeof(T));
}
// (the caller has made sure that nothing in flight still uses the block: wa_acs_destroy waits for the context's streams first)
static void ctx_free(wa_ctx *c, const void *cp)
{
    void *p = const_cast<void *>(cp);
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto ia = c->arena_live.find(p);
        if (ia != c->arena_live.end()) {
            WaArenaBlock b = std::move(ia->second);
            c->arena_live.erase(ia);
            if (c->cache_on) {   // stays mapped: the next request of exactly this size takes it as it is
                b.stamp = ++c->clock;
                c->cache_bytes += b.va_bytes;
                c->arena_kept.push_back(std::move(b));
                cache_enforce_limits_locked(c);
            } else {
                hipMemUnmap(p, b.va_bytes);
                hipMemAddressFree(p, b.va_bytes);
                for (int k = 0; k < 3; k++)
                    for (auto h : b.chunks[k]) hipMemRelease(h);
            }
            return;
        }
        auto it = c->live.find(p);
        if (it != c->live.end()) {
            const size_t bytes = it->second;
            c->live.erase(it);
            if (c->cache_on) {
                c->cache.push_back({p, bytes, ++c->clock});
                c->cache_bytes += bytes;
                cache_enforce_limits_locked(c);
                return;
            }
        }
    }
    hipFree(p);
}

extern "C" {

#ifdef WA_TEST_KNOBS
const char *wa_version(void) { return "weldacs 0.3 (gfx950, test knobs)"; }
#else
const char *wa_version(void) { return "weldacs 0.3 (gfx950)"; }
#endif

int wa_ctx_create(int device_ordinal, wa_ctx **out)
{
    if (!out) return WA_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return WA_ERR_DEVICE;
    wa_ctx *c = new wa_ctx();
    c->device = device_ordinal;
    WaDevGuard dev_guard_(c);   // streams and events are created on the context's device; the caller's current device is restored
    if (!dev_guard_.ok ||
        hipGetDeviceProperties(&c->prop, device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return WA_ERR_DEVICE;
    }
    c->cache_on = env_int("WA_DEV_CACHE", 1) != 0;
    c->poison = env_int("WA_DEV_POISON", 0) != 0;
    {   // the arena needs virtual memory management; how much the context may keep is a share of the device's memory
        int vmm = 0;
        if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, device_ordinal) != hipSuccess) { vmm = 0; (void)hipGetLastError(); }
        c->arena_on = c->cache_on && vmm != 0 && env_int("WA_DEV_ARENA", 1) != 0;
        int pct = env_int("WA_DEV_KEEP_PCT", 95);
        pct = pct < 0 ? 0 : pct > 100 ? 100 : pct;
        c->keep_limit = (size_t)((double)c->prop.totalGlobalMem * pct / 100.0);
        c->small_cache_limit = (size_t)env_int("WA_DEV_SMALL_CACHE_MB", 2048) << 20;
    }
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        // test knobs of the arena's fall-backs (tests/test_gpu_arena.py), process-wide like the window itself: WA_DEV_ARENA_VA_MB = size of
        // the address window in MiB (0: nothing -- every block a whole allocation although the arena is on), WA_DEV_ARENA_NOHINT=1 = reserve
        // without address hints, i.e. a platform that does not place ranges where they are asked for
        const int va_mb = env_int("WA_DEV_ARENA_VA_MB", -1);
        if (va_mb >= 0 && g_va_end == WA_VA_END) g_va_end = WA_VA_FIRST + ((uintptr_t)va_mb << 20);
        if (env_int("WA_DEV_ARENA_NOHINT", 0)) g_va_nohint = true;
        g_cache_ctxs.push_back(c);
    }
    *out = c;
    return WA_OK;
}
int wa_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
int wa_ctx_memory_info(wa_ctx *c, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!c) return WA_ERR_ARG;
    WaDevGuard dev_guard_(c);
    if (!dev_guard_.ok) return WA_ERR_DEVICE;   // the context's device could not be made current
    size_t f = 0, t = 0;
    HIPC(c, hipMemGetInfo(&f, &t));
    {   // blocks the contexts on this device keep for their next solver are there for the asking
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (const wa_ctx *o : g_cache_ctxs)
            if (o->device == c->device) f += o->cache_bytes;
    }
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return WA_OK;
}
int wa_ctx_cached_bytes(wa_ctx *c, int64_t *bytes)
{
    if (!c || !bytes) return WA_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    *bytes = (int64_t)c->cache_bytes;
    return WA_OK;
}
int wa_ctx_cache_stats(wa_ctx *c, int64_t out[8])
{
    if (!c || !out) return WA_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (int i = 0; i < 7; i++) out[i] = c->stat[i];
    out[7] = c->arena_on ? (g_va_exhausted ? 2 : 1) : 0;   // 2: in use, but its address window is used up -- new shapes come as whole blocks
    return WA_OK;
}
int wa_ctx_trim(wa_ctx *c)
{
    if (!c) return WA_ERR_ARG;
    WaDevGuard dev_guard_(c);
    if (!dev_guard_.ok) return WA_ERR_DEVICE;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    cache_release_locked(c);
    return WA_OK;
}
void wa_ctx_destroy(wa_ctx *c)
{
    if (!c) return;
    WaDevGuard dev_guard_(c);
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        cache_release_locked(c);
        g_cache_ctxs.erase(std::remove(g_cache_ctxs.begin(), g_cache_ctxs.end(), c), g_cache_ctxs.end());
    }
    hipStreamDestroy(c->stream);
    delete c;
}
const char *wa_last_error(const wa_ctx *c) { return c ? c->err.c_str() : "no context"; }
int wa_ctx_device_name(const wa_ctx *c, char *buf, size_t cap)
{
    if (!c || !buf || !cap) return WA_ERR_ARG;
    snprintf(buf, cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    return WA_OK;
}
int wa_ctx_sync(wa_ctx *c)
{
    WaDevGuard dev_guard_(c);
    if (!dev_guard_.ok) return WA_ERR_DEVICE;   // the context's device could not be made current
    if (!c) return WA_ERR_ARG;
    HIPC(c, hipStreamSynchronize(c->stream));
    return WA_OK;
}
void *wa_ctx_stream(wa_ctx *c) { return c ? (void *)c->stream : nullptr; }

#include "host_grid.inc"
#include "host_acs.inc"
#include "host_gtsp.inc"
#include "host_traj.inc"
#include "host_comm.inc"

}  // extern "C"
