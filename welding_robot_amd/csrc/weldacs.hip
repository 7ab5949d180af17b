// weldacs.hip -- C ABI of libweldacs.so (include/weldacs.h) over the gfx950 kernels.
// Single translation unit: hipcc --offload-arch=gfx950 -ffp-contract=off (see build.py).
// There is no CPU compute path in this library: without a HIP device wa_ctx_create fails.
#include "../../include/weldacs.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

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

// ------------------------------------------------------------------ handles
struct WaDevBlock { void *p; size_t bytes; };
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
    std::vector<WaDevBlock> cache;
    std::unordered_map<void *, size_t> live;   // big blocks handed out by ctx_alloc and not yet returned
    size_t cache_bytes = 0;
    bool cache_on = true, poison = false;      // WA_DEV_CACHE=0 / WA_DEV_POISON=1, read at wa_ctx_create
};
struct wa_grid {
    wa_ctx *ctx;
    WaDims d;
    float precision;
    int32_t wall;
    int64_t n_free;
    float *cx, *cy, *cz;   // device
    uint8_t *occ;          // device
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
    int32_t *paths_base = nullptr;   // the ants' paths: one array, or two (paths_half elements apart) alternating by generation when stragglers are handed over
    size_t paths_half = 0;
    float *pher_buf[2];    // double buffer: evaporation writes the other one (dst = src * rho)
    float *pher_alloc[2], *heur_alloc;   // the allocations behind pher_buf[] / D.heur (fields + guard bands)
    uint32_t *stamp_alloc;               // ... and D.stamp (lazy solvers)
    int cur_buf;
    std::vector<int> slot_buf;           // which of the two buffers holds slot q's current field (inactive slots do not follow the flips)
    bool walk_asm;         // hand-scheduled walk loop (default); WA_WALK_ASM=0 keeps the compiler-scheduled one
    int walk_warm;         // touch loads in the hand-scheduled loop: -1 by launch size (wa_acs_run), 0 / 1 forced (WA_WALK_WARM)
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

template <class T>
static hipError_t dalloc(T **p, size_t count)
{
    return hipMalloc((void **)p, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
}

static int env_int(const char *name, int def)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : def;
}

// ------------------------------------------------------------------ the contexts' block caches
// Solvers take their device memory through ctx_alloc / ctx_free: a block of 1 MiB or more that a solver gives back stays with the
// context, and the next solver of a similar shape (the drop-in's pair loop creates one per searchBestPathOfPoints call, a planning
// service one per job) gets it back without the driver's free -> wipe -> allocate round trip.  Nothing is assumed about a block's
// contents, fresh or reused (WA_DEV_POISON=1 fills every block with 0xff bytes before it is handed out: the GPU suite passes that
// way).  Cached bytes count as free in wa_ctx_memory_info; when the device runs out, every context's cache on that device is
// released and the allocation retried, waiting for the wipe.  wa_ctx_trim releases a context's cache, WA_DEV_CACHE=0 switches it off.
static std::mutex g_cache_mu;
static std::vector<wa_ctx *> g_cache_ctxs;
static const size_t WA_CACHE_MIN_BYTES = (size_t)1 << 20;

static size_t cache_release_locked(wa_ctx *c)
{
    size_t freed = 0;
    for (auto &b : c->cache) { hipFree(b.p); freed += b.bytes; }
    c->cache.clear();
    c->cache_bytes = 0;
    return freed;
}
static size_t cache_release_device(int device)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t freed = 0;
    for (wa_ctx *o : g_cache_ctxs)
        if (o->device == device) freed += cache_release_locked(o);
    return freed;
}
static hipError_t ctx_alloc_bytes(wa_ctx *c, void **out, size_t bytes)
{
    if (bytes < 16) bytes = 16;
    *out = nullptr;
    const bool big = c->cache_on && bytes >= WA_CACHE_MIN_BYTES;
    if (big) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        int best = -1;
        for (int i = 0; i < (int)c->cache.size(); i++) {
            const size_t b = c->cache[i].bytes;
            if (b >= bytes && b - bytes <= bytes / 8 && (best < 0 || b < c->cache[best].bytes)) best = i;
        }
        if (best >= 0) {
            *out = c->cache[best].p;
            c->live[*out] = c->cache[best].bytes;
            c->cache_bytes -= c->cache[best].bytes;
            c->cache.erase(c->cache.begin() + best);
        }
    }
    if (!*out) {
        hipError_t e = hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory) {
            // every cache on this device goes back to the driver; memory that is being wiped is neither free nor allocatable for a
            // while (hipMalloc fails rather than waits when most of the device is in that state): retry while the free figure moves
            (void)hipGetLastError();
            cache_release_device(c->device);
            size_t last_free = 0;
            int stable = 0;
            for (int tries = 0; tries < 1200; tries++) {
                e = hipMalloc(out, bytes);
                if (e != hipErrorOutOfMemory) break;
                (void)hipGetLastError();
                size_t f = 0, t = 0;
                if (hipMemGetInfo(&f, &t) != hipSuccess) break;
                stable = f == last_free ? stable + 1 : 0;
                last_free = f;
                if (stable >= 20) break;   // a second without change: it really does not fit
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        }
        if (e != hipSuccess) { *out = nullptr; return e; }
        if (big) {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            c->live[*out] = bytes;
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
        auto it = c->live.find(p);
        if (it != c->live.end()) {
            const size_t bytes = it->second;
            c->live.erase(it);
            if (c->cache_on) {
                c->cache.push_back({p, bytes});
                c->cache_bytes += bytes;
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
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
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
