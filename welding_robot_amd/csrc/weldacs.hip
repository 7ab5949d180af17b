// weldacs.hip -- C ABI of libweldacs.so (include/weldacs.h) over the gfx950 kernels.
// Single translation unit: hipcc --offload-arch=gfx950 -ffp-contract=off (see build.py).
// There is no CPU compute path in this library: without a HIP device wa_ctx_create fails.
#include "../../include/weldacs.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "acs_kernels.hpp"
#include "grid_kernels.hpp"
#include "gtsp_kernels.hpp"
#include "traj_kernels.hpp"

// ------------------------------------------------------------------ handles
struct wa_ctx {
    int device;
    hipStream_t stream;    // main stream: every kernel except the overlapped evaporation sweep
    hipStream_t stream2;   // evaporation sweep of generation g runs here, concurrently with walk g
    hipEvent_t ev_fork, ev_join;
    std::string err;
    hipDeviceProp_t prop;
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
};
struct EvPair { hipEvent_t a, b; int cls; };
struct wa_acs {
    wa_ctx *ctx;
    const wa_grid *grid;
    int32_t n_slots, max_colony, n_active, nb;
    int64_t path_cap;
    WaAcsDev D;            // D.pher always points at the CURRENT pheromone buffer
    float *pher_buf[2];    // double buffer: evaporation writes the other one (dst = src * rho)
    int cur_buf;
    WaRun R;
    bool begun, overlap_walk, overlap_rank, fuse, inplace, fuse_table;
    bool lazy;                          // lazy evaporation (wa_acs_create_lazy): never-deposited voxels are not swept
    std::vector<int> lazy_mode;         // per slot: init mode of the stored records (-1 unknown)
    std::vector<float> lazy_p0;
    int32_t gens_enqueued, colony_bound, hash_log2, evap_blocks;
    // hipGraph of `graph_len` generations of the fused DEV loop (0 = off), valid for the run begun last
    int32_t graph_len, graph_buf0, genbase_host;
    hipGraph_t graph;
    hipGraphExec_t graph_exec;
    long long *d_starts, *d_ends;
    uint32_t *d_streams;
    // profiling
    bool prof;
    int32_t prof_every;
    std::vector<EvPair> ev;
    double prof_ms[WA_K_COUNT];
    int64_t prof_n[WA_K_COUNT];
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

extern "C" {

const char *wa_version(void) { return "weldacs 0.1 (gfx950)"; }

int wa_ctx_create(int device_ordinal, wa_ctx **out)
{
    if (!out) return WA_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return WA_ERR_DEVICE;
    if (hipSetDevice(device_ordinal) != hipSuccess) return WA_ERR_DEVICE;
    wa_ctx *c = new wa_ctx();
    c->device = device_ordinal;
    if (hipGetDeviceProperties(&c->prop, device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return WA_ERR_DEVICE;
    }
    *out = c;
    return WA_OK;
}
void wa_ctx_destroy(wa_ctx *c)
{
    if (!c) return;
    hipStreamDestroy(c->stream);
    hipStreamDestroy(c->stream2);
    hipEventDestroy(c->ev_fork);
    hipEventDestroy(c->ev_join);
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
    if (!c) return WA_ERR_ARG;
    HIPC(c, hipStreamSynchronize(c->stream2));
    HIPC(c, hipStreamSynchronize(c->stream));
    return WA_OK;
}
void *wa_ctx_stream(wa_ctx *c) { return c ? (void *)c->stream : nullptr; }

// ------------------------------------------------------------------ STL (read_STL.hpp)
int64_t wa_stl_parse(const void *buf, size_t len, float *tris, int64_t cap_tris)
{
    const uint8_t *b = (const uint8_t *)buf;
    if (!b) return -WA_ERR_ARG;
    if (len < 84) return -WA_ERR_FILE;            // shorter than header + count
    if (b[79] != 0) return -WA_ERR_FORMAT;        // ASCII sniff (:65); ASCII carries no usable normals (Q11)
    int32_t n;
    memcpy(&n, b + 80, 4);                         // cpyint :158
    if (n < 0 || (size_t)n * 50 + 84 > len) return -WA_ERR_FILE;
    if (!tris) return n;
    if (cap_tris < n) return -WA_ERR_CAPACITY;
    const uint8_t *p = b + 84;
    for (int64_t i = 0; i < n; i++, p += 50) memcpy(tris + i * 12, p, 48);  // :142-151
    return n;
}
int64_t wa_stl_read_file(const char *path, float *tris, int64_t cap_tris)
{
    if (!path) return -WA_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) return -WA_ERR_FILE;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    rewind(f);
    std::vector<uint8_t> buf(sz > 0 ? (size_t)sz : 0);
    size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    if ((long)got != sz) return -WA_ERR_FILE;
    return wa_stl_parse(buf.data(), buf.size(), tris, cap_tris);
}

// ------------------------------------------------------------------ grid
int wa_axis_coords(float lo, float hi, float precision, int32_t wall, int32_t n, float *out)
{
    if (!out || n < 0) return WA_ERR_ARG;
    for (int32_t i = 0; i < n; i++)  // model_grid_map.hpp:204-211
        out[i] = i < wall ? lo - (float)(wall - i) * precision
                          : (i >= (n - wall) ? hi + (float)(i - n + wall) * precision : lo + (float)(i - wall) * precision);
    return WA_OK;
}

static int grid_alloc(wa_ctx *ctx, int32_t nx, int32_t ny, int32_t nz, const float *cx, const float *cy,
                      const float *cz, float precision, int32_t wall, wa_grid **out)
{
    if (nx < 1 || ny < 1 || nz < 1) return fail(ctx, WA_ERR_ARG, "grid dimensions must be >= 1");
    int64_t n = (int64_t)nx * ny * nz;
    if (n > (int64_t)WA_ID_MASK) return fail(ctx, WA_ERR_ARG, "grid larger than 2^29 voxels");
    wa_grid *g = new wa_grid();
    g->ctx = ctx;
    g->d.nx = nx; g->d.ny = ny; g->d.nz = nz; g->d.nxy = nx * ny; g->d.n = n;
    g->precision = precision;
    g->wall = wall;
    g->n_free = -1;
    g->cx = g->cy = g->cz = nullptr;
    g->occ = nullptr;
    if (dalloc(&g->cx, nx) || dalloc(&g->cy, ny) || dalloc(&g->cz, nz) || dalloc(&g->occ, n)) {
        wa_grid_destroy(g);
        return fail(ctx, WA_ERR_ALLOC, "grid device allocation failed");
    }
    HIPC(ctx, hipMemcpyAsync(g->cx, cx, sizeof(float) * nx, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(g->cy, cy, sizeof(float) * ny, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(g->cz, cz, sizeof(float) * nz, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    *out = g;
    return WA_OK;
}

static int grid_count_free(wa_grid *g)
{
    wa_ctx *ctx = g->ctx;
    unsigned long long *d_cnt = nullptr, h = 0;
    hipError_t e = hipMalloc((void **)&d_cnt, 8);
    e = e ? e : hipMemsetAsync(d_cnt, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        k_count_free<<<1024, 256, 0, ctx->stream>>>(g->occ, g->d.n, d_cnt);
        e = hipGetLastError();
    }
    e = e ? e : hipMemcpyAsync(&h, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream);
    e = e ? e : hipStreamSynchronize(ctx->stream);
    hipFree(d_cnt);
    if (e != hipSuccess) return fail(ctx, WA_ERR_DEVICE, "free-voxel count: %s", hipGetErrorString(e));
    g->n_free = (int64_t)h;
    return WA_OK;
}

int wa_grid_from_mesh(wa_ctx *ctx, const float *tris, int64_t n_tris, float precision, int32_t wall,
                      wa_grid **out, float *bbox6_out)
{
    if (!ctx || !tris || !out || n_tris <= 0 || !(precision > 0) || wall < 0) return fail(ctx, WA_ERR_ARG, "wa_grid_from_mesh: bad argument");
    *out = nullptr;
    // bbox over all vertices (model_grid_map.hpp:165-181) and the ranges (:198-200)
    float mn[3] = {tris[3], tris[4], tris[5]}, mx[3] = {tris[3], tris[4], tris[5]};
    for (int64_t t = 0; t < n_tris; t++)
        for (int v = 0; v < 3; v++)
            for (int c = 0; c < 3; c++) {
                float q = tris[t * 12 + 3 + v * 3 + c];
                mx[c] = q > mx[c] ? q : mx[c];
                mn[c] = q < mn[c] ? q : mn[c];
            }
    int32_t dims[3];
    for (int c = 0; c < 3; c++) dims[c] = (int)((mx[c] - mn[c]) / precision) + 1 + 2 * wall;
    if (bbox6_out) for (int c = 0; c < 3; c++) { bbox6_out[c] = mn[c]; bbox6_out[3 + c] = mx[c]; }
    std::vector<float> ax[3];
    for (int c = 0; c < 3; c++) {
        if (dims[c] < 1) return fail(ctx, WA_ERR_ARG, "degenerate mesh extent");
        ax[c].resize(dims[c]);
        wa_axis_coords(mn[c], mx[c], precision, wall, dims[c], ax[c].data());
    }
    wa_grid *g = nullptr;
    int rc = grid_alloc(ctx, dims[0], dims[1], dims[2], ax[0].data(), ax[1].data(), ax[2].data(), precision, wall, &g);
    if (rc) return rc;
    float *d_tris = nullptr;
    if (dalloc(&d_tris, (size_t)n_tris * 12)) { wa_grid_destroy(g); return fail(ctx, WA_ERR_ALLOC, "triangle buffer"); }
    hipError_t ve = hipMemcpyAsync(d_tris, tris, sizeof(float) * 12 * n_tris, hipMemcpyHostToDevice, ctx->stream);
    if (ve == hipSuccess) {
        if (env_int("WA_VOXELIZE_DENSE", 0)) {   // the O(T*N^3) form, kept for comparison
            unsigned blocks = (unsigned)((g->d.n + 255) / 256);
            k_voxelize<<<blocks, 256, 0, ctx->stream>>>(d_tris, n_tris, precision, g->d, g->cx, g->cy, g->cz, g->occ);
        } else {
            ve = hipMemsetAsync(g->occ, 1, (size_t)g->d.n, ctx->stream);
            for (int64_t t0 = 0; ve == hipSuccess && t0 < n_tris; t0 += 1 << 20) {
                const int64_t cnt = n_tris - t0 < (1 << 20) ? n_tris - t0 : (1 << 20);
                k_voxelize_clip<<<dim3((unsigned)cnt, 4), 256, 0, ctx->stream>>>(d_tris + t0 * 12, cnt, precision, g->d, g->cx, g->cy, g->cz, g->occ);
            }
        }
        ve = ve ? ve : hipGetLastError();
    }
    if (ve == hipSuccess) ve = hipStreamSynchronize(ctx->stream);
    hipFree(d_tris);
    if (ve != hipSuccess) { wa_grid_destroy(g); return fail(ctx, WA_ERR_DEVICE, "voxelise: %s", hipGetErrorString(ve)); }
    rc = grid_count_free(g);
    if (rc) { wa_grid_destroy(g); return rc; }
    *out = g;
    return WA_OK;
}

int wa_grid_from_occupancy(wa_ctx *ctx, const uint8_t *free_, int32_t nx, int32_t ny, int32_t nz,
                           const float *cx, const float *cy, const float *cz, float precision,
                           int32_t wall, wa_grid **out)
{
    if (!ctx || !free_ || !cx || !cy || !cz || !out || !(precision > 0)) return fail(ctx, WA_ERR_ARG, "wa_grid_from_occupancy: bad argument");
    *out = nullptr;
    wa_grid *g = nullptr;
    int rc = grid_alloc(ctx, nx, ny, nz, cx, cy, cz, precision, wall, &g);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(g->occ, free_, (size_t)g->d.n, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    rc = grid_count_free(g);
    if (rc) { wa_grid_destroy(g); return rc; }
    *out = g;
    return WA_OK;
}

void wa_grid_destroy(wa_grid *g)
{
    if (!g) return;
    hipFree(g->cx); hipFree(g->cy); hipFree(g->cz); hipFree(g->occ);
    delete g;
}
int wa_grid_info(const wa_grid *g, int32_t dims3[3], float *precision, int32_t *wall, int64_t *n_free)
{
    if (!g) return WA_ERR_ARG;
    if (dims3) { dims3[0] = g->d.nx; dims3[1] = g->d.ny; dims3[2] = g->d.nz; }
    if (precision) *precision = g->precision;
    if (wall) *wall = g->wall;
    if (n_free) *n_free = g->n_free;
    return WA_OK;
}
int wa_grid_read_occupancy(const wa_grid *g, uint8_t *free_out)
{
    if (!g || !free_out) return WA_ERR_ARG;
    HIPC(g->ctx, hipMemcpy(free_out, g->occ, (size_t)g->d.n, hipMemcpyDeviceToHost));
    return WA_OK;
}
int wa_grid_read_coords(const wa_grid *g, float *cx, float *cy, float *cz)
{
    if (!g) return WA_ERR_ARG;
    if (cx) HIPC(g->ctx, hipMemcpy(cx, g->cx, sizeof(float) * g->d.nx, hipMemcpyDeviceToHost));
    if (cy) HIPC(g->ctx, hipMemcpy(cy, g->cy, sizeof(float) * g->d.ny, hipMemcpyDeviceToHost));
    if (cz) HIPC(g->ctx, hipMemcpy(cz, g->cz, sizeof(float) * g->d.nz, hipMemcpyDeviceToHost));
    return WA_OK;
}
int wa_grid_resolve_points(const wa_grid *g, const float *pts_xyz, int32_t n_pts, int64_t *ids_out)
{
    if (!g || !pts_xyz || !ids_out || n_pts < 0) return WA_ERR_ARG;
    if (n_pts == 0) return WA_OK;
    wa_ctx *ctx = g->ctx;
    float *d_pts = nullptr;
    long long *d_ids = nullptr;
    std::vector<long long> h(n_pts);
    hipError_t e = dalloc(&d_pts, (size_t)n_pts * 3);
    e = e ? e : dalloc(&d_ids, (size_t)n_pts);
    if (e != hipSuccess) { hipFree(d_pts); hipFree(d_ids); return fail(ctx, WA_ERR_ALLOC, "resolve buffers"); }
    e = hipMemcpyAsync(d_pts, pts_xyz, sizeof(float) * 3 * n_pts, hipMemcpyHostToDevice, ctx->stream);
    e = e ? e : hipMemsetAsync(d_ids, 0xff, sizeof(long long) * n_pts, ctx->stream);  // -1
    if (e == hipSuccess) {
        if (env_int("WA_RESOLVE_DENSE", 0)) {   // thread per voxel x every point, kept for comparison
            unsigned blocks = (unsigned)((g->d.n + 255) / 256);
            k_resolve_points<<<blocks, 256, 0, ctx->stream>>>(g->d, g->cx, g->cy, g->cz, g->occ, g->precision, d_pts, n_pts, d_ids);
        } else {
            k_resolve_points_clip<<<(unsigned)n_pts, 256, 0, ctx->stream>>>(g->d, g->cx, g->cy, g->cz, g->occ, g->precision, d_pts, n_pts, d_ids);
        }
        e = hipGetLastError();
    }
    e = e ? e : hipMemcpyAsync(h.data(), d_ids, sizeof(long long) * n_pts, hipMemcpyDeviceToHost, ctx->stream);
    e = e ? e : hipStreamSynchronize(ctx->stream);
    hipFree(d_pts);
    hipFree(d_ids);
    if (e != hipSuccess) return fail(ctx, WA_ERR_DEVICE, "wa_grid_resolve_points: %s", hipGetErrorString(e));
    for (int32_t i = 0; i < n_pts; i++) ids_out[i] = h[i];
    return WA_OK;
}

// ------------------------------------------------------------------ ACS
void wa_acs_default_params(wa_acs_params *p)
{
    if (!p) return;
    p->alpha = 1;            // ACSRank_3D.hpp:319
    p->beta = 0.6f;          // :320
    p->rho = 0.8f;           // :321
    p->pheromone_0 = 1.f;    // :324
    p->max_iteration = 150;  // :322
    p->predict = 10.f;       // default argument of searchBestPathOfPoints :427
    p->fixed_colony = 0;
    p->rng_mode = WA_RNG_DEV;
    p->seed = 1;
}


static int acs_create(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony,
                      int64_t path_capacity, int32_t nb, bool lazy, wa_acs **out)
{
    if (!ctx || !grid || !out || n_slots < 1 || max_colony < 1) return fail(ctx, WA_ERR_ARG, "wa_acs_create: bad argument");
    *out = nullptr;
    wa_acs *s = new wa_acs();
    memset(&s->D, 0, sizeof s->D);
    s->ctx = ctx;
    s->grid = grid;
    s->n_slots = n_slots;
    s->max_colony = max_colony;
    s->n_active = 0;
    s->begun = false;
    s->gens_enqueued = 0;
    s->prof = false;
    s->prof_every = 1;
    s->d_starts = s->d_ends = nullptr;
    s->d_streams = nullptr;
    s->pher_buf[0] = s->pher_buf[1] = nullptr;
    s->cur_buf = 0;
    for (int i = 0; i < WA_K_COUNT; i++) { s->prof_ms[i] = 0; s->prof_n[i] = 0; }
    const int64_t n = grid->d.n;
    if (nb != 6 && nb != 26) { delete s; return fail(ctx, WA_ERR_ARG, "wa_acs_create: neighbourhood must be 6 or 26"); }
    s->nb = nb;
    s->lazy = lazy;
    s->lazy_mode.assign(n_slots, -1);
    s->lazy_p0.assign(n_slots, 0.f);
    if (nb == 6 && 24 * n >= (int64_t)1 << 31) {  // the walk addresses a slot's pheromone field with signed 32-bit byte offsets
        delete s;
        return fail(ctx, WA_ERR_ARG, "wa_acs_create: grids above 89,478,485 voxels (~447^3) are not supported");
    }
    if (nb == 26 && n > (int64_t)WaNbT<26>::IDM) {  // path word = 27-bit voxel id + 5-bit edge index
        delete s;
        return fail(ctx, WA_ERR_ARG, "wa_acs_create: 26-neighbour grids above 2^27 voxels (512^3) are not supported");
    }
    if (path_capacity <= 0) path_capacity = n < (1 << 18) ? n : (1 << 18);
    if (path_capacity > n) path_capacity = n;
    if (path_capacity < 2) path_capacity = 2;
    s->path_cap = path_capacity;
    WaAcsDev &D = s->D;
    D.d = grid->d;
    D.cx = grid->cx; D.cy = grid->cy; D.cz = grid->cz; D.occ = grid->occ;
    D.nb = nb;
    D.pher_stride = (((int64_t)nb * n + 63) / 64) * 64;
    D.path_cap = path_capacity;
    D.vbits_words = (n + 31) / 32;
    D.max_colony = max_colony;
    D.trace_cap = 0;
    int lg = 11;
    while ((1 << lg) < 16 * (grid->d.nx + grid->d.ny + grid->d.nz) && lg < 13) lg++;
    s->hash_log2 = env_int("WA_HASH_LOG2", lg);
    if (s->hash_log2 < 6) s->hash_log2 = 6;
    if (s->hash_log2 > 14) s->hash_log2 = 14;
    {   // sweep grid: measured on MI355X -- 100 MB fields (128^3 x 6) peak at 4096 blocks (6.0 TB/s; 2048: 5.5, 8192: 5.8),
        // 436 MB fields (128^3 x 26) want 2-3 float4 per thread (49152 blocks: 6.1 TB/s; 32768: 5.9; 4096: 4.6)
        const int64_t n4 = (int64_t)nb * n / 4;
        int64_t blocks = n4 <= ((int64_t)8 << 20) ? 4096 : n4 / 555;
        if (blocks > 65536) blocks = 65536;
        s->evap_blocks = env_int("WA_EVAP_BLOCKS", (int)blocks);
    }
    s->overlap_walk = env_int("WA_OVERLAP_WALK", 0) != 0;
    s->overlap_rank = env_int("WA_OVERLAP_RANK", 0) != 0;
    s->fuse = env_int("WA_FUSE", 1) != 0;
    s->fuse_table = env_int("WA_FUSE_TABLE", 1) != 0;
    s->inplace = env_int("WA_EVAP_INPLACE", 0) != 0 && !s->overlap_walk;
    if (lazy) { s->fuse = s->fuse_table = true; s->overlap_walk = s->overlap_rank = s->inplace = false; }   // one loop shape only
    s->graph_len = lazy ? 0 : env_int("WA_GRAPH", 0) & ~1;   // even: the pheromone double buffer is back where it started
    s->graph = nullptr;
    s->graph_exec = nullptr;
    s->genbase_host = 0;
    const size_t S = (size_t)n_slots, C = (size_t)max_colony;
    hipError_t e = hipSuccess;
    e = e ? e : dalloc(&s->pher_buf[0], S * D.pher_stride);
    if (!lazy) e = e ? e : dalloc(&s->pher_buf[1], S * D.pher_stride);   // the lazy sweep is in place: one field
    if (lazy) {
        e = e ? e : dalloc(&D.stamp, S * n);
        e = e ? e : dalloc(&D.dirty_list, S * n);
        e = e ? e : dalloc(&D.dcount, S * 2);
    }
    e = e ? e : dalloc(&D.heur, S * D.pher_stride);
    e = e ? e : dalloc(&D.mask, S * D.pher_stride);
    e = e ? e : dalloc(&D.bestmark, S * n);
    e = e ? e : dalloc(&D.bestpath, S * path_capacity);
    e = e ? e : dalloc(&D.bestpos, S * n);
    e = e ? e : dalloc(&D.besttabu, S * path_capacity);
    if (lazy || env_int("WA_REPLAY", 1) != 0)   // replay table: 8 floats per best-path node (6 neighbours) / 32 (26 neighbours)
        e = e ? e : dalloc(&D.rtab, S * path_capacity * (nb == 6 ? 8 : WA_ROW26) + 256);
    e = e ? e : dalloc(&D.paths, S * C * path_capacity);
    e = e ? e : dalloc(&D.antL, S * C);
    e = e ? e : dalloc(&D.antLen, S * C);
    e = e ? e : dalloc(&D.perm, S * C);
    e = e ? e : dalloc(&D.depA, S * C);
    e = e ? e : dalloc(&D.sortk, S * C * 2);
    e = e ? e : dalloc(&D.vbits, S * C * D.vbits_words);
    e = e ? e : dalloc(&D.ctl, S);
    e = e ? e : dalloc(&D.rng, 1);
    e = e ? e : dalloc(&D.dbg, 16);
    e = e ? e : dalloc(&D.genbase, 1);
    e = e ? e : dalloc(&s->d_starts, S);
    e = e ? e : dalloc(&s->d_ends, S);
    e = e ? e : dalloc(&s->d_streams, S);
    if (e != hipSuccess) {
        wa_acs_destroy(s);
        return fail(ctx, WA_ERR_ALLOC, "wa_acs_create: device allocation failed: %s", hipGetErrorString(e));
    }
    HIPC(ctx, hipMemsetAsync(D.mask, 0, sizeof(unsigned long long) * S * D.pher_stride, ctx->stream));
    HIPC(ctx, hipMemsetAsync(D.heur, 0, sizeof(float) * S * D.pher_stride, ctx->stream));
    D.pher = s->pher_buf[0];
    s->cur_buf = 0;
    HIPC(ctx, hipMemsetAsync(s->pher_buf[0], 0, sizeof(float) * S * D.pher_stride, ctx->stream));
    if (s->pher_buf[1]) HIPC(ctx, hipMemsetAsync(s->pher_buf[1], 0, sizeof(float) * S * D.pher_stride, ctx->stream));
    if (lazy) {
        HIPC(ctx, hipMemsetAsync(D.stamp, 0, sizeof(uint32_t) * S * n, ctx->stream));
        HIPC(ctx, hipMemsetAsync(D.dcount, 0, sizeof(int32_t) * S * 2, ctx->stream));
    }
    HIPC(ctx, hipMemsetAsync(D.bestmark, 0, sizeof(uint32_t) * S * n, ctx->stream));
    HIPC(ctx, hipMemsetAsync(D.vbits, 0, sizeof(uint32_t) * S * C * D.vbits_words, ctx->stream));
    HIPC(ctx, hipMemsetAsync(D.ctl, 0, sizeof(WaSlotCtl) * S, ctx->stream));
    HIPC(ctx, hipMemsetAsync(D.dbg, 0, sizeof(unsigned long long) * 16, ctx->stream));
    HIPC(ctx, hipMemsetAsync(D.genbase, 0, sizeof(int32_t), ctx->stream));
    WaGlibcRand r0;
    wa_glibc_seed(&r0, 1);  // a process that never calls srand() behaves as srand(1)
    HIPC(ctx, hipMemcpyAsync(D.rng, &r0, sizeof r0, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    int rc = wa_acs_init_pheromone(s, -1, 1.0f);
    if (rc) { wa_acs_destroy(s); *out = nullptr; return rc; }
    return WA_OK;
}

int wa_acs_create(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony, int64_t path_capacity,
                  wa_acs **out)
{
    return acs_create(ctx, grid, n_slots, max_colony, path_capacity, 6, false, out);
}
int wa_acs_create_nb(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony, int64_t path_capacity,
                     int32_t neighbourhood, wa_acs **out)
{
    return acs_create(ctx, grid, n_slots, max_colony, path_capacity, neighbourhood, false, out);
}
int wa_acs_create_lazy(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony, int64_t path_capacity,
                       wa_acs **out)
{
    return acs_create(ctx, grid, n_slots, max_colony, path_capacity, 6, true, out);
}

static void free_trace(wa_acs *s)
{
    hipFree(s->D.trBest); hipFree(s->D.trIter); hipFree(s->D.trColony); hipFree(s->D.trFinite); hipFree(s->D.trSteps);
    s->D.trBest = s->D.trIter = nullptr;
    s->D.trColony = s->D.trFinite = nullptr;
    s->D.trSteps = nullptr;
    s->D.trace_cap = 0;
}

void wa_acs_destroy(wa_acs *s)
{
    if (!s) return;
    hipStreamSynchronize(s->ctx->stream2);
    hipStreamSynchronize(s->ctx->stream);
    for (auto &p : s->ev) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    WaAcsDev &D = s->D;
    hipFree(s->pher_buf[0]); hipFree(s->pher_buf[1]); hipFree(D.heur); hipFree(D.mask); hipFree(D.bestmark); hipFree(D.bestpath); hipFree(D.bestpos); hipFree(D.besttabu); hipFree(D.rtab);
    hipFree(D.paths); hipFree(D.antL); hipFree(D.antLen); hipFree(D.perm); hipFree(D.depA);
    hipFree(D.sortk); hipFree(D.vbits); hipFree(D.ctl); hipFree(D.rng); hipFree(D.dbg); hipFree(D.genbase);
    hipFree(D.stamp); hipFree(D.dirty_list); hipFree(D.dcount);
    if (s->graph_exec) hipGraphExecDestroy(s->graph_exec);
    if (s->graph) hipGraphDestroy(s->graph);
    hipFree(s->d_starts); hipFree(s->d_ends); hipFree(s->d_streams);
    free_trace(s);
    delete s;
}

static int init_pher(wa_acs *s, int32_t slot, float p0, int mode)
{
    if (!s || slot >= s->n_slots || !(p0 >= 0)) return fail(s ? s->ctx : nullptr, WA_ERR_ARG, "pheromone init: bad argument");
    int32_t slot0 = slot < 0 ? 0 : slot, cnt = slot < 0 ? s->n_slots : 1;
    if (s->nb == 26) {
        dim3 grid26((unsigned)((s->D.d.n * 26 + 255) / 256), (unsigned)cnt);
        k_init_pheromone26<<<grid26, 256, 0, s->ctx->stream>>>(s->D, slot0, p0, mode);
        HIPC(s->ctx, hipGetLastError());
        return WA_OK;
    }
    if (s->lazy) {
        // same init mode and p0 as the records already hold: rewrite the dirty records only (reset() between pair
        // searches costs O(deposited voxels) instead of a 24-B/voxel pass); otherwise the full pass
        bool same = true;
        for (int32_t q = slot0; q < slot0 + cnt; q++) same = same && s->lazy_mode[q] == mode && s->lazy_p0[q] == p0;
        if (same) {
            k_lazy_restore<<<dim3(64, (unsigned)cnt), 256, 0, s->ctx->stream>>>(s->D, slot0, p0, mode);
        } else {
            dim3 gridf((unsigned)((s->D.d.n + 255) / 256), (unsigned)cnt);
            k_init_pheromone<<<gridf, 256, 0, s->ctx->stream>>>(s->D, slot0, p0, mode);
            HIPC(s->ctx, hipMemsetAsync(s->D.stamp + (int64_t)slot0 * s->D.d.n, 0, sizeof(uint32_t) * cnt * s->D.d.n, s->ctx->stream));
        }
        k_lazy_clear<<<(cnt + 63) / 64, 64, 0, s->ctx->stream>>>(s->D, slot0, cnt, p0);
        HIPC(s->ctx, hipGetLastError());
        for (int32_t q = slot0; q < slot0 + cnt; q++) { s->lazy_mode[q] = mode; s->lazy_p0[q] = p0; }
        return WA_OK;
    }
    dim3 grid((unsigned)((s->D.d.n + 255) / 256), (unsigned)cnt);
    k_init_pheromone<<<grid, 256, 0, s->ctx->stream>>>(s->D, slot0, p0, mode);
    HIPC(s->ctx, hipGetLastError());
    return WA_OK;
}
int wa_acs_init_pheromone(wa_acs *s, int32_t slot, float p0) { return init_pher(s, slot, p0, 0); }
int wa_acs_reset_pheromone(wa_acs *s, int32_t slot, float p0) { return init_pher(s, slot, p0, 1); }

int wa_acs_srand(wa_acs *s, uint32_t seed)
{
    if (!s) return WA_ERR_ARG;
    WaGlibcRand r;
    wa_glibc_seed(&r, seed);
    HIPC(s->ctx, hipMemcpyAsync(s->D.rng, &r, sizeof r, hipMemcpyHostToDevice, s->ctx->stream));
    HIPC(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return WA_OK;
}
int wa_acs_rand_state(wa_acs *s, int32_t st[36], int32_t set)
{
    if (!s || !st) return WA_ERR_ARG;
    WaGlibcRand r;
    if (set) {
        memcpy(r.r, st, sizeof(int32_t) * 34);
        r.f = st[34];
        r.b = st[35];
        HIPC(s->ctx, hipMemcpyAsync(s->D.rng, &r, sizeof r, hipMemcpyHostToDevice, s->ctx->stream));
        HIPC(s->ctx, hipStreamSynchronize(s->ctx->stream));
    } else {
        HIPC(s->ctx, hipStreamSynchronize(s->ctx->stream));
        HIPC(s->ctx, hipMemcpy(&r, s->D.rng, sizeof r, hipMemcpyDeviceToHost));
        memcpy(st, r.r, sizeof(int32_t) * 34);
        st[34] = r.f;
        st[35] = r.b;
    }
    return WA_OK;
}

int wa_acs_begin(wa_acs *s, const wa_acs_params *p, int32_t n_problems, const int64_t *start_ids,
                 const int64_t *end_ids, const uint32_t *streams)
{
    if (!s || !p || !start_ids || !end_ids) return fail(s ? s->ctx : nullptr, WA_ERR_ARG, "wa_acs_begin: null argument");
    wa_ctx *ctx = s->ctx;
    if (n_problems < 1 || n_problems > s->n_slots) return fail(ctx, WA_ERR_ARG, "wa_acs_begin: n_problems exceeds the solver's slots");
    if (p->rng_mode != WA_RNG_REF && p->rng_mode != WA_RNG_DEV) return fail(ctx, WA_ERR_ARG, "wa_acs_begin: rng_mode");
    if (p->rng_mode == WA_RNG_REF && n_problems != 1)
        return fail(ctx, WA_ERR_ARG, "wa_acs_begin: REF mode shares one libc stream, so problems run one at a time");
    if (p->max_iteration < 0 || !(p->rho > 0) || !(p->pheromone_0 >= 0) || p->alpha < 0)
        return fail(ctx, WA_ERR_ARG, "wa_acs_begin: parameter out of range");
    for (int32_t i = 0; i < n_problems; i++) {
        if (start_ids[i] < 0 || end_ids[i] < 0) return fail(ctx, WA_ERR_POINT, "wa_acs_begin: unresolved route point");
        if (start_ids[i] >= s->D.d.n || end_ids[i] >= s->D.d.n) return fail(ctx, WA_ERR_ARG, "wa_acs_begin: voxel id out of range");
    }
    WaRun &R = s->R;
    if (s->lazy && s->begun && R.rho != p->rho)   // pending evaporations of deposited records belong to the previous rho
        k_lazy_flush<<<dim3(64, (unsigned)s->n_slots), 256, 0, ctx->stream>>>(s->D, R.rho);
    R.alpha = p->alpha; R.beta = p->beta; R.rho = p->rho; R.pheromone_0 = p->pheromone_0;
    R.predict = p->predict; R.precision = s->grid->precision; R.fixed_colony = p->fixed_colony;
    R.rng_mode = p->rng_mode; R.seed = p->seed;
    // largest colony this run can reach: min(best.L, predict) <= predict (:247)
    int32_t bound = p->fixed_colony > 0 ? p->fixed_colony : (int32_t)(0.35 * (double)p->predict / (double)R.precision);
    if (bound > s->max_colony) return fail(ctx, WA_ERR_CAPACITY, "wa_acs_begin: colony exceeds max_colony of the solver");
    s->colony_bound = bound < 0 ? 0 : bound;
    if (s->lazy && (p->rng_mode != WA_RNG_DEV || s->colony_bound > WA_RANK_LDS || (int32_t)(0.2 * s->colony_bound) + 1 > 64 || !s->D.rtab))
        return fail(ctx, WA_ERR_ARG, "wa_acs_begin: a lazily evaporating solver runs the fused DEV loop only (DEV mode, <= 2048 ants, <= 64 depositing ranks)");
    if (p->max_iteration > s->D.trace_cap) {
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        free_trace(s);
        size_t T = (size_t)s->n_slots * p->max_iteration;
        if (dalloc(&s->D.trBest, T) || dalloc(&s->D.trIter, T) || dalloc(&s->D.trColony, T) ||
            dalloc(&s->D.trFinite, T) || dalloc(&s->D.trSteps, T))
            return fail(ctx, WA_ERR_ALLOC, "trace buffers");
        s->D.trace_cap = p->max_iteration;
    }
    std::vector<long long> hs(n_problems), he(n_problems);
    std::vector<uint32_t> hst(n_problems);
    for (int32_t i = 0; i < n_problems; i++) { hs[i] = start_ids[i]; he[i] = end_ids[i]; hst[i] = streams ? streams[i] : (uint32_t)i; }
    HIPC(ctx, hipMemcpyAsync(s->d_starts, hs.data(), sizeof(long long) * n_problems, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(s->d_ends, he.data(), sizeof(long long) * n_problems, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(s->d_streams, hst.data(), sizeof(uint32_t) * n_problems, hipMemcpyHostToDevice, ctx->stream));
    k_begin<<<(n_problems + 63) / 64, 64, 0, ctx->stream>>>(s->D, R, n_problems, s->d_starts, s->d_ends, s->d_streams);
    k_set_genbase<<<1, 1, 0, ctx->stream>>>(s->D, 0);
    s->genbase_host = 0;
    if (s->graph_exec) { hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }   // parameters are baked into the nodes
    if (s->graph) { hipGraphDestroy(s->graph); s->graph = nullptr; }
    if (s->nb == 26) {
        dim3 hg26((unsigned)((s->D.d.n * 26 + 255) / 256), (unsigned)n_problems);
        k_heuristic26<<<hg26, 256, 0, ctx->stream>>>(s->D, R.beta);
    } else {
        dim3 hg((unsigned)((s->D.d.n + 255) / 256), (unsigned)n_problems);
        k_heuristic<<<hg, 256, 0, ctx->stream>>>(s->D, R.beta);
    }
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hipStreamSynchronize(ctx->stream));  // host staging vectors go out of scope
    s->n_active = n_problems;
    s->begun = true;
    s->gens_enqueued = 0;
    return WA_OK;
}

static EvPair *prof_open(wa_acs *s, int cls, bool sampled)
{
    if (!sampled) return nullptr;
    EvPair p;
    p.cls = cls;
    if (hipEventCreate(&p.a) != hipSuccess) return nullptr;
    if (hipEventCreate(&p.b) != hipSuccess) { hipEventDestroy(p.a); return nullptr; }
    hipEventRecord(p.a, s->ctx->stream);
    s->ev.push_back(p);
    return &s->ev.back();
}
static void prof_close(wa_acs *s, EvPair *p)
{
    if (p) hipEventRecord(p->b, s->ctx->stream);
}

// dst = src * rho for `cnt` slots starting at slot0.  When `timed` the dispatch carries its own
// start/stop events (hipExtLaunchKernelGGL): they stamp the kernel itself, not the stream gaps.
static void launch_evaporate(wa_acs *s, hipStream_t st, const float *src, float *dst, int32_t slot0, int32_t cnt, float rho, bool timed)
{
    dim3 grid((unsigned)s->evap_blocks, (unsigned)cnt);
    const float *sp = src + (int64_t)slot0 * s->D.pher_stride;
    float *dp = dst + (int64_t)slot0 * s->D.pher_stride;
    if (timed) {
        EvPair p;
        p.cls = WA_K_EVAPORATE;
        if (hipEventCreate(&p.a) == hipSuccess) {
            if (hipEventCreate(&p.b) == hipSuccess) {
                hipExtLaunchKernelGGL(k_evaporate, grid, dim3(256), 0, st, p.a, p.b, 0, sp, dp, s->D.pher_stride, (int64_t)s->nb * s->D.d.n, rho);
                s->ev.push_back(p);
                return;
            }
            hipEventDestroy(p.a);
        }
    }
    k_evaporate<<<grid, 256, 0, st>>>(sp, dp, s->D.pher_stride, (int64_t)s->nb * s->D.d.n, rho);
}

// the fused post-walk launch (sweep + rank + mark), timed per dispatch when sampled
static void launch_fused(wa_acs *s, const float *src, float *dst, int32_t P, int32_t gen_off, bool timed)
{
    if (s->lazy) {   // sparse sweep of the deposited voxels + rank + mark (which also enrols newly deposited voxels)
        const int32_t E = env_int("WA_LAZY_BLOCKS", 2048);   // grid-stride over the dirty list; surplus blocks exit at once
        dim3 lgrid((unsigned)(E + 512), (unsigned)P);
        if (timed) {
            EvPair p;
            p.cls = WA_K_EVAPORATE;
            if (hipEventCreate(&p.a) == hipSuccess) {
                if (hipEventCreate(&p.b) == hipSuccess) {
                    hipExtLaunchKernelGGL(k_evap_rank_mark<true>, lgrid, dim3(256), 0, s->ctx->stream, p.a, p.b, 0, s->D, s->R, src, dst, E, gen_off);
                    s->ev.push_back(p);
                    return;
                }
                hipEventDestroy(p.a);
            }
        }
        k_evap_rank_mark<true><<<lgrid, 256, 0, s->ctx->stream>>>(s->D, s->R, src, dst, E, gen_off);
        return;
    }
    dim3 grid((unsigned)(s->evap_blocks + 512), (unsigned)P);
    if (timed) {
        EvPair p;
        p.cls = WA_K_EVAPORATE;
        if (hipEventCreate(&p.a) == hipSuccess) {
            if (hipEventCreate(&p.b) == hipSuccess) {
                hipExtLaunchKernelGGL(k_evap_rank_mark<false>, grid, dim3(256), 0, s->ctx->stream, p.a, p.b, 0, s->D, s->R, src, dst,
                                      s->evap_blocks, gen_off);
                s->ev.push_back(p);
                return;
            }
            hipEventDestroy(p.a);
        }
    }
    k_evap_rank_mark<false><<<grid, 256, 0, s->ctx->stream>>>(s->D, s->R, src, dst, s->evap_blocks, gen_off);
}

// One generation = walk -> rank -> evaporate -> deposit (ACSRank_3D.hpp:252-280), enqueued on one
// stream without host synchronisation.  The sweep is out-of-place (dst = src*rho into the other
// pheromone buffer, which then becomes current), which makes two overlaps legal; both were
// measured on MI355X (128^3, 256 ants) and both LOSE, so they are off by default and kept as knobs:
//   WA_OVERLAP_WALK=1  sweep on stream2 alongside the walk (it reads the field the walk reads):
//                      5561 vs 5764 gen/s -- its traffic lengthens the latency-bound walk by what it saves
//   WA_OVERLAP_RANK=1  rank on stream2 alongside the sweep: 5322 gen/s -- the event fork/join
//                      costs more than the 10 us ranking kernel it hides
int wa_acs_run(wa_acs *s, int32_t n_generations)
{
    if (!s || n_generations < 0) return WA_ERR_ARG;
    wa_ctx *ctx = s->ctx;
    if (!s->begun) return fail(ctx, WA_ERR_STATE, "wa_acs_run before wa_acs_begin");
    const int32_t P = s->n_active;
    const size_t shmem = sizeof(int32_t) << s->hash_log2;
    const int32_t dep_bound = (int32_t)(0.2 * s->colony_bound) + 1;
    const int32_t chunks = (dep_bound + 63) / 64;
    const bool fused = s->lazy || s->nb == 6 && s->fuse && s->R.rng_mode == WA_RNG_DEV && s->colony_bound <= WA_RANK_LDS && dep_bound <= 64 &&
                       !s->overlap_walk && !s->overlap_rank;
    // One generation of the fused DEV loop: walk -> {sweep + rank + mark} -> {apply + replay table}.  gen_off is
    // relative to the device generation counter; advance != 0 on the last generation of a captured graph.
    auto enqueue_fused = [&](int32_t gen_off, int32_t advance, bool sampled) {
        const bool one_buf = s->inplace || s->lazy;
        float *src = s->pher_buf[s->cur_buf], *dst = s->pher_buf[s->cur_buf ^ (one_buf ? 0 : 1)];
        EvPair *e = prof_open(s, WA_K_WALK, sampled);
        if (s->colony_bound > 0) {
            dim3 wg((unsigned)s->colony_bound, (unsigned)P);
            if (s->lazy) {
                if (s->R.alpha == 1) k_walk_dev<true, true><<<wg, 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen_off);
                else k_walk_dev<false, true><<<wg, 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen_off);
            } else {
                if (s->R.alpha == 1) k_walk_dev<true, false><<<wg, 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen_off);
                else k_walk_dev<false, false><<<wg, 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen_off);
            }
        }
        prof_close(s, e);
        launch_fused(s, src, dst, P, gen_off, sampled);
        if (!one_buf) s->cur_buf ^= 1;
        s->D.pher = dst;
        e = prof_open(s, WA_K_DEPOSIT, sampled);
        if (s->D.rtab && s->fuse_table) {
            k_apply_table<<<dim3(WA_TABLE_BLOCKS + 512, (unsigned)P), 256, 0, ctx->stream>>>(s->D, s->R, advance);
        } else {
            k_deposit_apply<6><<<dim3(8, 64, (unsigned)P), 256, 0, ctx->stream>>>(s->D, 0);
            if (s->D.rtab) k_replay_table<<<dim3(32, (unsigned)P), 256, 0, ctx->stream>>>(s->D, s->R);
        }
        prof_close(s, e);
    };
    int32_t g0 = 0;
    // hipGraph replay (WA_GRAPH=G): G generations captured once per run and launched as one graph; the kernels
    // take their generation number from the device counter, the graph's last kernel advances it by G
    const int32_t G = s->graph_len;
    if (fused && G > 0 && !s->prof && s->D.rtab && s->fuse_table && !s->inplace && n_generations >= G) {
        if (s->graph_exec && s->cur_buf != s->graph_buf0 && n_generations > G) {  // an odd number of plain generations ran since
            enqueue_fused(s->gens_enqueued - s->genbase_host, 0, false);
            s->gens_enqueued++;
            g0++;
        }
        if (!s->graph_exec || s->cur_buf == s->graph_buf0) {
            if (s->genbase_host != s->gens_enqueued) {
                k_set_genbase<<<1, 1, 0, ctx->stream>>>(s->D, s->gens_enqueued);
                s->genbase_host = s->gens_enqueued;
            }
            if (!s->graph_exec) {
                s->graph_buf0 = s->cur_buf;
                HIPC(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
                for (int32_t q = 0; q < G; q++) enqueue_fused(q, q == G - 1 ? G : 0, false);
                HIPC(ctx, hipStreamEndCapture(ctx->stream, &s->graph));
                HIPC(ctx, hipGraphInstantiate(&s->graph_exec, s->graph, nullptr, nullptr, 0));
            }
            while (n_generations - g0 >= G) {
                HIPC(ctx, hipGraphLaunch(s->graph_exec, ctx->stream));
                g0 += G;
                s->gens_enqueued += G;
                s->genbase_host += G;
            }
        }
    }
    for (int32_t g = g0; g < n_generations; g++) {
        const bool sampled = s->prof && ((s->gens_enqueued % s->prof_every) == 0);
        const int32_t gen = s->gens_enqueued;  // == the device-side generation counter since wa_acs_begin
        if (fused) {  // DEV fast path: 3 launches
            enqueue_fused(gen - s->genbase_host, 0, sampled);
            s->gens_enqueued++;
            continue;
        }
        float *src = s->pher_buf[s->cur_buf], *dst = s->pher_buf[s->cur_buf ^ (s->inplace ? 0 : 1)];
        if (s->overlap_walk && s->nb == 6) {  // fork the sweep before the walk
            HIPC(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPC(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            launch_evaporate(s, ctx->stream2, src, dst, 0, P, s->R.rho, sampled);
        }
        EvPair *e = prof_open(s, WA_K_WALK, sampled);
        if (s->nb == 26) {
            if (s->R.rng_mode == WA_RNG_DEV) {
                if (s->colony_bound > 0) k_walk_dev26<<<dim3((unsigned)s->colony_bound, (unsigned)P), 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen);
            } else {
                k_walk_ref26<<<dim3(1, 1), 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen);
            }
        } else if (s->R.rng_mode == WA_RNG_DEV) {
            if (s->colony_bound > 0) {
                dim3 wg((unsigned)s->colony_bound, (unsigned)P);
                if (s->R.alpha == 1) k_walk_dev<true, false><<<wg, 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen - s->genbase_host);
                else k_walk_dev<false, false><<<wg, 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen - s->genbase_host);
            }
        } else {
            k_walk_ref<<<dim3(1, 1), 64, shmem, ctx->stream>>>(s->D, s->R, s->hash_log2, gen);
        }
        prof_close(s, e);
        if (s->overlap_walk && s->nb == 6) {  // rank on the main stream, join the early sweep
            e = prof_open(s, WA_K_RANK, sampled);
            k_rank<6><<<P, 256, 0, ctx->stream>>>(s->D, s->R, gen);
            prof_close(s, e);
            HIPC(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            HIPC(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        } else if (s->overlap_rank && s->nb == 6) {  // rank on stream2 || sweep on the main stream
            HIPC(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPC(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            k_rank<6><<<P, 256, 0, ctx->stream2>>>(s->D, s->R, gen);
            HIPC(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            launch_evaporate(s, ctx->stream, src, dst, 0, P, s->R.rho, sampled);
            HIPC(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        } else {  // fully serial
            e = prof_open(s, WA_K_RANK, sampled);
            if (s->nb == 26) k_rank<26><<<P, 256, 0, ctx->stream>>>(s->D, s->R, gen);
            else k_rank<6><<<P, 256, 0, ctx->stream>>>(s->D, s->R, gen);
            prof_close(s, e);
            launch_evaporate(s, ctx->stream, src, dst, 0, P, s->R.rho, sampled);
        }
        if (!s->inplace) s->cur_buf ^= 1;
        s->D.pher = dst;
        e = prof_open(s, WA_K_DEPOSIT, sampled);
        for (int32_t c = 0; c < chunks; c++) {
            dim3 dg(8, 64, (unsigned)P);
            if (s->nb == 26) {
                k_deposit_mark<26><<<dg, 256, 0, ctx->stream>>>(s->D, c * 64);
                k_deposit_apply<26><<<dg, 256, 0, ctx->stream>>>(s->D, c * 64);
            } else {
                k_deposit_mark<6><<<dg, 256, 0, ctx->stream>>>(s->D, c * 64);
                k_deposit_apply<6><<<dg, 256, 0, ctx->stream>>>(s->D, c * 64);
            }
        }
        if (s->D.rtab && s->R.rng_mode == WA_RNG_DEV) {
            if (s->nb == 26) k_replay_table26<<<dim3(257, (unsigned)P), 64, 0, ctx->stream>>>(s->D, s->R);
            else k_replay_table<<<dim3(32, (unsigned)P), 256, 0, ctx->stream>>>(s->D, s->R);
        }
        prof_close(s, e);
        s->gens_enqueued++;
    }
    HIPC(ctx, hipGetLastError());
    return WA_OK;
}

int wa_acs_sync(wa_acs *s)
{
    if (!s) return WA_ERR_ARG;
    wa_ctx *ctx = s->ctx;
    HIPC(ctx, hipStreamSynchronize(ctx->stream2));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    // fold finished event pairs into the accumulators
    for (auto &p : s->ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { s->prof_ms[p.cls] += ms; s->prof_n[p.cls]++; }
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    s->ev.clear();
    if (s->begun && s->n_active > 0) {
        std::vector<WaSlotCtl> c(s->n_active);
        HIPC(ctx, hipMemcpy(c.data(), s->D.ctl, sizeof(WaSlotCtl) * s->n_active, hipMemcpyDeviceToHost));
        for (auto &x : c) {
            if (x.flags & WA_FLAG_PATH_OVERFLOW) return fail(ctx, WA_ERR_CAPACITY, "a walk outgrew path_capacity; results are not reference-exact");
            if (x.flags & WA_FLAG_COLONY_OVERFLOW) return fail(ctx, WA_ERR_CAPACITY, "colony exceeded max_colony");
        }
    }
    return WA_OK;
}

int wa_acs_solve(wa_acs *s, const wa_acs_params *p, int32_t n_problems, const int64_t *start_ids,
                 const int64_t *end_ids, const uint32_t *streams)
{
    int rc = wa_acs_begin(s, p, n_problems, start_ids, end_ids, streams);
    if (rc) return rc;
    rc = wa_acs_run(s, p->max_iteration);
    if (rc) return rc;
    return wa_acs_sync(s);
}

int wa_acs_result(wa_acs *s, int32_t slot, float *cost, int64_t *len, int32_t *path_ids, int8_t *choices, int64_t cap)
{
    if (!s || slot < 0 || slot >= s->n_slots) return WA_ERR_ARG;
    wa_ctx *ctx = s->ctx;
    if (!s->begun) return fail(ctx, WA_ERR_STATE, "no solve has run");
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    WaSlotCtl c;
    HIPC(ctx, hipMemcpy(&c, s->D.ctl + slot, sizeof c, hipMemcpyDeviceToHost));
    if (cost) *cost = c.bestL;
    int64_t n = isinf(c.bestL) ? 0 : c.best_len;
    if (len) *len = n;
    if ((path_ids || choices) && n > 0) {
        if (cap < n) return fail(ctx, WA_ERR_CAPACITY, "wa_acs_result: output capacity too small");
        std::vector<int32_t> w(n);
        HIPC(ctx, hipMemcpy(w.data(), s->D.bestpath + (int64_t)slot * s->D.path_cap, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) {
            const int32_t idm = s->nb == 26 ? WaNbT<26>::IDM : WaNbT<6>::IDM;
            const int sh = s->nb == 26 ? WaNbT<26>::SHIFT : WaNbT<6>::SHIFT;
            if (path_ids) path_ids[i] = w[i] & idm;
            if (choices && i > 0) choices[i - 1] = (int8_t)((uint32_t)w[i] >> sh);
        }
    }
    return WA_OK;
}

int wa_acs_trace(wa_acs *s, int32_t slot, int32_t *generations_done, float *best_L, float *iter_best_L,
                 int32_t *colony, int32_t *finite, int64_t *steps)
{
    if (!s || slot < 0 || slot >= s->n_slots) return WA_ERR_ARG;
    wa_ctx *ctx = s->ctx;
    if (!s->begun) return fail(ctx, WA_ERR_STATE, "no solve has run");
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    int32_t g = s->gens_enqueued < s->D.trace_cap ? s->gens_enqueued : s->D.trace_cap;
    if (generations_done) *generations_done = g;
    int64_t off = (int64_t)slot * s->D.trace_cap;
    if (g > 0) {
        if (best_L) HIPC(ctx, hipMemcpy(best_L, s->D.trBest + off, sizeof(float) * g, hipMemcpyDeviceToHost));
        if (iter_best_L) HIPC(ctx, hipMemcpy(iter_best_L, s->D.trIter + off, sizeof(float) * g, hipMemcpyDeviceToHost));
        if (colony) HIPC(ctx, hipMemcpy(colony, s->D.trColony + off, sizeof(int32_t) * g, hipMemcpyDeviceToHost));
        if (finite) HIPC(ctx, hipMemcpy(finite, s->D.trFinite + off, sizeof(int32_t) * g, hipMemcpyDeviceToHost));
        if (steps) HIPC(ctx, hipMemcpy(steps, s->D.trSteps + off, sizeof(int64_t) * g, hipMemcpyDeviceToHost));
    }
    return WA_OK;
}

int wa_acs_export_trace(wa_acs *s, void *dst_device, int32_t gen0, int32_t count)
{
    if (!s || !dst_device || gen0 < 0 || count < 1) return WA_ERR_ARG;
    wa_ctx *ctx = s->ctx;
    if (!s->begun || gen0 + count > s->D.trace_cap) return fail(ctx, WA_ERR_STATE, "wa_acs_export_trace: range not recorded");
    HIPC(ctx, hipMemcpy2DAsync(dst_device, sizeof(float) * count, s->D.trBest + gen0, sizeof(float) * s->D.trace_cap,
                               sizeof(float) * count, (size_t)s->n_active, hipMemcpyDeviceToDevice, ctx->stream));
    return WA_OK;
}

int wa_acs_read_pheromone(wa_acs *s, int32_t slot, float *out)
{
    if (!s || !out || slot < 0 || slot >= s->n_slots) return WA_ERR_ARG;
    wa_ctx *ctx = s->ctx;
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t m = (int64_t)s->nb * s->D.d.n;
    if (s->lazy) {   // the field as the dense sweep would have left it
        float *d_out = nullptr;
        if (dalloc(&d_out, (size_t)m)) return fail(ctx, WA_ERR_ALLOC, "wa_acs_read_pheromone: staging");
        k_lazy_materialise<<<(unsigned)((m + 255) / 256), 256, 0, ctx->stream>>>(s->D, s->R, slot, d_out);
        hipError_t h = hipGetLastError();
        h = h ? h : hipMemcpy(out, d_out, sizeof(float) * m, hipMemcpyDeviceToHost);
        hipFree(d_out);
        if (h != hipSuccess) return fail(ctx, WA_ERR_DEVICE, "wa_acs_read_pheromone: %s", hipGetErrorString(h));
        return WA_OK;
    }
    HIPC(ctx, hipMemcpy(out, s->D.pher + (int64_t)slot * s->D.pher_stride, sizeof(float) * m, hipMemcpyDeviceToHost));
    uint32_t *u = (uint32_t *)out;
    for (int64_t i = 0; i < m; i++) u[i] &= 0x7fffffffu;  // drop the admissibility bit
    return WA_OK;
}

int wa_acs_last_params(wa_acs *s, int32_t slot, int32_t *colony, float *lambda, float *Q)
{
    if (!s || slot < 0 || slot >= s->n_slots) return WA_ERR_ARG;
    wa_ctx *ctx = s->ctx;
    if (!s->begun || s->gens_enqueued < 1) return fail(ctx, WA_ERR_STATE, "no generation has run");
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    WaSlotCtl c;
    HIPC(ctx, hipMemcpy(&c, s->D.ctl + slot, sizeof c, hipMemcpyDeviceToHost));
    int32_t g = s->gens_enqueued - 1;
    if (colony && g < s->D.trace_cap) HIPC(ctx, hipMemcpy(colony, s->D.trColony + (int64_t)slot * s->D.trace_cap + g, 4, hipMemcpyDeviceToHost));
    if (lambda) *lambda = c.dep_lambda;
    if (Q) *Q = c.dep_Q;
    return WA_OK;
}

int wa_acs_profile(wa_acs *s, int32_t enable, int32_t sample_every)
{
    if (!s) return WA_ERR_ARG;
    s->prof = enable != 0;
    s->prof_every = sample_every > 0 ? sample_every : 1;
    for (int i = 0; i < WA_K_COUNT; i++) { s->prof_ms[i] = 0; s->prof_n[i] = 0; }
    return WA_OK;
}
int wa_acs_profile_read(wa_acs *s, double ms[WA_K_COUNT], int64_t launches[WA_K_COUNT])
{
    if (!s) return WA_ERR_ARG;
    int rc = wa_acs_sync(s);
    for (int i = 0; i < WA_K_COUNT; i++) {
        if (ms) ms[i] = s->prof_ms[i];
        if (launches) launches[i] = s->prof_n[i];
    }
    return rc;
}
int wa_acs_debug_counters(wa_acs *s, uint64_t out16[16], int32_t reset)
{
    if (!s || !out16) return WA_ERR_ARG;
    HIPC(s->ctx, hipStreamSynchronize(s->ctx->stream));
    HIPC(s->ctx, hipMemcpy(out16, s->D.dbg, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost));
    if (reset) HIPC(s->ctx, hipMemset(s->D.dbg, 0, sizeof(uint64_t) * 16));
    return WA_OK;
}
int wa_acs_evaporate(wa_acs *s, int32_t slot, float rho, int32_t repeats)
{
    if (!s || slot < 0 || slot >= s->n_slots || repeats < 1) return WA_ERR_ARG;
    if (s->lazy) return fail(s->ctx, WA_ERR_STATE, "wa_acs_evaporate: this solver evaporates lazily (no dense sweep to run)");
    // same out-of-place sweep as the generation loop; all slots flip together, so the other
    // slots are carried across with rho = 1 (exact copy)
    for (int32_t r = 0; r < repeats; r++) {
        float *src = s->pher_buf[s->cur_buf], *dst = s->pher_buf[s->cur_buf ^ 1];
        for (int32_t q = 0; q < s->n_slots; q++)
            launch_evaporate(s, s->ctx->stream, src, dst, q, 1, q == slot ? rho : 1.0f, s->prof && q == slot);
        s->cur_buf ^= 1;
        s->D.pher = dst;
    }
    HIPC(s->ctx, hipGetLastError());
    return WA_OK;
}

// ------------------------------------------------------------------ GTSP
int wa_gtsp_solve(wa_ctx *ctx, const double *dist, int32_t n, int32_t cnt, int32_t n_instances,
                  const wa_gtsp_params *p, int32_t *rand_state36, int32_t *tour_edges,
                  double *tour_cost, int32_t *iterations, double *pheromone_out)
{
    if (!ctx || !dist || !p || !tour_edges || n < 2 || n_instances < 1) return fail(ctx, WA_ERR_ARG, "wa_gtsp_solve: bad argument");
    if (p->rng_mode == WA_RNG_REF && n_instances != 1) return fail(ctx, WA_ERR_ARG, "wa_gtsp_solve: REF mode runs one instance");
    const size_t I = (size_t)n_instances, nn = (size_t)n * n;
    WaGtspDev G;
    memset(&G, 0, sizeof G);
    double *d_dist = nullptr;
    hipError_t e = hipSuccess;
    e = e ? e : dalloc(&d_dist, I * nn);
    e = e ? e : dalloc(&G.pher, I * nn);
    e = e ? e : dalloc(&G.h6, I * nn);
    e = e ? e : dalloc(&G.info, I * (size_t)257 * n + I * nn);
    e = e ? e : dalloc(&G.antL, I * n);
    e = e ? e : dalloc(&G.tours, I * nn * 2);
    e = e ? e : dalloc(&G.best, I * n * 2);
    e = e ? e : dalloc(&G.inJ, I * nn);
    e = e ? e : dalloc(&G.rbuf, nn);
    e = e ? e : dalloc(&G.rng, 1);
    e = e ? e : dalloc(&G.out_cost, I);
    e = e ? e : dalloc(&G.out_iters, I);
    auto cleanup = [&]() {
        hipFree(d_dist); hipFree(G.pher); hipFree(G.h6); hipFree(G.info); hipFree(G.antL); hipFree(G.tours);
        hipFree(G.best); hipFree(G.inJ); hipFree(G.rbuf); hipFree(G.rng); hipFree(G.out_cost); hipFree(G.out_iters);
    };
    if (e != hipSuccess) { cleanup(); return fail(ctx, WA_ERR_ALLOC, "wa_gtsp_solve: %s", hipGetErrorString(e)); }
    G.dist = d_dist;
    G.n = n; G.cnt = cnt; G.max_iterations = p->max_iterations; G.rng_mode = p->rng_mode;
    G.seed = p->seed; G.stream0 = p->stream;
    WaGlibcRand r;
    if (rand_state36) { memcpy(r.r, rand_state36, sizeof(int32_t) * 34); r.f = rand_state36[34]; r.b = rand_state36[35]; }
    else wa_glibc_seed(&r, 1);
    hipError_t h = hipMemcpyAsync(d_dist, dist, sizeof(double) * I * nn, hipMemcpyHostToDevice, ctx->stream);
    h = h ? h : hipMemcpyAsync(G.rng, &r, sizeof r, hipMemcpyHostToDevice, ctx->stream);
    h = h ? h : hipMemsetAsync(G.best, 0, sizeof(int32_t) * I * n * 2, ctx->stream);
    if (h == hipSuccess) {
        // fast path: lanes = ants, info matrix in LDS (rows padded to the mask width, odd stride) when
        // it fits; for n <= 64 the per-ant prefix sums live there too (binary-search second pass)
        const int nw = n <= 64 ? 1 : (n <= 128 ? 2 : 4);
        const size_t ld = (size_t)(64 * nw + 1);
        const size_t info_bytes = sizeof(double) * ld * n;
        const bool in_lds = info_bytes <= 140 * 1024;
        const unsigned threads = (unsigned)(((n + 63) / 64) * 64);
        const bool prefix = nw == 1;
        // one instance (or a few): spread each ant over a wavefront; many instances already fill the GPU with the
        // lanes-as-ants kernel (64 instances x 64 cities: 48 ms vs 82 ms), so the choice goes by the wave count
        const bool wave_path = env_int("WA_GTSP_WAVE", (int64_t)n_instances * n <= 1024 ? 1 : 0) != 0;
        if (n <= 256 && wave_path && env_int("WA_GTSP_GENERIC", 0) == 0) {
            // wave-per-ant path: lanes = cities, grid = (ants, instances); construct -> update per iteration
            WaGtspWave W;
            W.G = G;
            W.state = nullptr;
            W.valid = nullptr;
            h = dalloc(&W.state, I);
            h = h ? h : dalloc(&W.valid, I * n);
            // cities per lane: the n ordered additions are sequential whatever the layout, but a lane-to-lane hop
            // (two DPP moves) costs several additions, so few lanes with many cities each win (n/NC hops per step)
            const int nc = n <= 128 ? 8 : 16;
            const size_t info_bytes_w = sizeof(double) * nn;
            const bool stage = info_bytes_w <= 128 * 1024;
            const size_t shm_w = stage ? info_bytes_w : 0;
            const int32_t max_it = p->max_iterations > 0 ? p->max_iterations : n * n;
            std::vector<WaGtspState> hst(I);
#define WA_GTSPW_CONSTRUCT(NC, ST)                                                                                            \
    do {                                                                                                                      \
        if (!attr_set) h = hipFuncSetAttribute((const void *)k_gtspw_construct<NC, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_w); \
        if (h == hipSuccess) k_gtspw_construct<NC, ST><<<dim3((unsigned)n, (unsigned)n_instances), 64, shm_w, ctx->stream>>>(W, it);           \
    } while (0)
            if (h == hipSuccess) {
                k_gtspw_init<<<(unsigned)n_instances, 256, 0, ctx->stream>>>(W);
                h = hipGetLastError();
            }
            bool attr_set = false;
            for (int32_t it = 0; h == hipSuccess && it < max_it; it++) {
                if (nc == 8) { if (stage) WA_GTSPW_CONSTRUCT(8, true); else WA_GTSPW_CONSTRUCT(8, false); }
                else { if (stage) WA_GTSPW_CONSTRUCT(16, true); else WA_GTSPW_CONSTRUCT(16, false); }
                attr_set = true;
                if (h == hipSuccess) k_gtspw_update<<<(unsigned)n_instances, 256, 0, ctx->stream>>>(W, it);
                if ((it & 7) == 7 || it == max_it - 1) {   // poll the stagnation stop (:263)
                    h = h ? h : hipMemcpyAsync(hst.data(), W.state, sizeof(WaGtspState) * I, hipMemcpyDeviceToHost, ctx->stream);
                    h = h ? h : hipStreamSynchronize(ctx->stream);
                    bool all = true;
                    for (auto &x : hst) all = all && x.stop != 0;
                    if (all) break;
                }
            }
#undef WA_GTSPW_CONSTRUCT
            if (h == hipSuccess) h = hipGetLastError();
            h = h ? h : hipStreamSynchronize(ctx->stream);
            hipFree(W.state);
            hipFree(W.valid);
        } else if (n <= 256 && env_int("WA_GTSP_GENERIC", 0) == 0) {
            size_t shm = (in_lds ? info_bytes : 0) + (prefix ? sizeof(double) * ld * threads : 0);
#define WA_GTSP_LAUNCH(NW, LDS, PFX)                                                                                     \
    do {                                                                                                                 \
        h = hipFuncSetAttribute((const void *)k_gtsp_fast<NW, LDS, PFX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
        if (h == hipSuccess) k_gtsp_fast<NW, LDS, PFX><<<(unsigned)n_instances, threads, shm, ctx->stream>>>(G);         \
    } while (0)
            if (nw == 1) WA_GTSP_LAUNCH(1, true, true);
            else if (nw == 2 && in_lds) WA_GTSP_LAUNCH(2, true, false);
            else if (nw == 2) WA_GTSP_LAUNCH(2, false, false);
            else WA_GTSP_LAUNCH(4, false, false);
#undef WA_GTSP_LAUNCH
        } else {
            k_gtsp<<<(unsigned)n_instances, 256, 0, ctx->stream>>>(G);
        }
        if (h == hipSuccess) h = hipGetLastError();
    }
    std::vector<double> cost(I);
    std::vector<int32_t> its(I);
    h = h ? h : hipMemcpyAsync(tour_edges, G.best, sizeof(int32_t) * I * n * 2, hipMemcpyDeviceToHost, ctx->stream);
    h = h ? h : hipMemcpyAsync(cost.data(), G.out_cost, sizeof(double) * I, hipMemcpyDeviceToHost, ctx->stream);
    h = h ? h : hipMemcpyAsync(its.data(), G.out_iters, sizeof(int32_t) * I, hipMemcpyDeviceToHost, ctx->stream);
    h = h ? h : hipMemcpyAsync(&r, G.rng, sizeof r, hipMemcpyDeviceToHost, ctx->stream);
    if (pheromone_out) h = h ? h : hipMemcpyAsync(pheromone_out, G.pher, sizeof(double) * I * nn, hipMemcpyDeviceToHost, ctx->stream);
    h = h ? h : hipStreamSynchronize(ctx->stream);
    cleanup();
    if (h != hipSuccess) return fail(ctx, WA_ERR_DEVICE, "wa_gtsp_solve: %s", hipGetErrorString(h));
    for (size_t i = 0; i < I; i++) {
        if (tour_cost) tour_cost[i] = cost[i];
        if (iterations) iterations[i] = its[i];
    }
    if (rand_state36 && p->rng_mode == WA_RNG_REF) { memcpy(rand_state36, r.r, sizeof(int32_t) * 34); rand_state36[34] = r.f; rand_state36[35] = r.b; }
    return WA_OK;
}

// ------------------------------------------------------------------ path post-processing
static int traj_alloc(wa_ctx *ctx, int64_t n, wa_traj **out)
{
    wa_traj *t = new wa_traj();
    t->ctx = ctx;
    t->n = n;
    t->xyz = nullptr;
    if (dalloc(&t->xyz, (size_t)n * 3)) {
        delete t;
        return fail(ctx, WA_ERR_ALLOC, "trajectory device allocation failed");
    }
    *out = t;
    return WA_OK;
}

int wa_traj_from_points(wa_ctx *ctx, const float *xyz, int64_t n, wa_traj **out)
{
    if (!ctx || !out || n < 0 || (n > 0 && !xyz)) return fail(ctx, WA_ERR_ARG, "wa_traj_from_points: bad argument");
    wa_traj *t = nullptr;
    int rc = traj_alloc(ctx, n, &t);
    if (rc) return rc;
    if (n) {
        hipError_t h = hipMemcpyAsync(t->xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, ctx->stream);
        h = h ? h : hipStreamSynchronize(ctx->stream);
        if (h != hipSuccess) { wa_traj_destroy(t); return fail(ctx, WA_ERR_DEVICE, "wa_traj_from_points: %s", hipGetErrorString(h)); }
    }
    *out = t;
    return WA_OK;
}

int wa_traj_stitch(const wa_grid *g, const int64_t *seg_ids, const int64_t *seg_off, int32_t n_seg,
                   const uint8_t *reverse, wa_traj **out)
{
    if (!g) return WA_ERR_ARG;
    wa_ctx *ctx = g->ctx;
    if (!out || n_seg < 0 || !seg_off) return fail(ctx, WA_ERR_ARG, "wa_traj_stitch: bad argument");
    if (seg_off[0] != 0) return fail(ctx, WA_ERR_ARG, "wa_traj_stitch: seg_off[0] must be 0");
    for (int32_t s = 0; s < n_seg; s++)
        if (seg_off[s + 1] < seg_off[s]) return fail(ctx, WA_ERR_ARG, "wa_traj_stitch: seg_off must be non-decreasing");
    int64_t n = seg_off[n_seg];
    if (n > 0 && !seg_ids) return fail(ctx, WA_ERR_ARG, "wa_traj_stitch: seg_ids is NULL");
    for (int64_t i = 0; i < n; i++)
        if (seg_ids[i] < 0 || seg_ids[i] >= g->d.n) return fail(ctx, WA_ERR_ARG, "wa_traj_stitch: node id outside the grid");
    wa_traj *t = nullptr;
    int rc = traj_alloc(ctx, n, &t);
    if (rc) return rc;
    if (n) {
        long long *d_ids = nullptr, *d_off = nullptr;
        uint8_t *d_rev = nullptr;
        hipError_t h = dalloc(&d_ids, (size_t)n);
        h = h ? h : dalloc(&d_off, (size_t)n_seg + 1);
        if (reverse) h = h ? h : dalloc(&d_rev, (size_t)n_seg);
        h = h ? h : hipMemcpyAsync(d_ids, seg_ids, sizeof(long long) * n, hipMemcpyHostToDevice, ctx->stream);
        h = h ? h : hipMemcpyAsync(d_off, seg_off, sizeof(long long) * (n_seg + 1), hipMemcpyHostToDevice, ctx->stream);
        if (reverse) h = h ? h : hipMemcpyAsync(d_rev, reverse, (size_t)n_seg, hipMemcpyHostToDevice, ctx->stream);
        if (h == hipSuccess) {
            k_stitch<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_ids, d_off, n_seg, d_rev, g->d, g->cx, g->cy, g->cz, t->xyz, n);
            h = hipGetLastError();
        }
        h = h ? h : hipStreamSynchronize(ctx->stream);
        hipFree(d_ids); hipFree(d_off); hipFree(d_rev);
        if (h != hipSuccess) { wa_traj_destroy(t); return fail(ctx, WA_ERR_DEVICE, "wa_traj_stitch: %s", hipGetErrorString(h)); }
    }
    *out = t;
    return WA_OK;
}

int64_t wa_traj_size(const wa_traj *t) { return t ? t->n : -1; }
int wa_traj_read(const wa_traj *t, float *xyz)
{
    if (!t || (t->n > 0 && !xyz)) return WA_ERR_ARG;
    if (t->n) HIPC(t->ctx, hipMemcpy(xyz, t->xyz, sizeof(float) * 3 * t->n, hipMemcpyDeviceToHost));
    return WA_OK;
}
void wa_traj_destroy(wa_traj *t)
{
    if (!t) return;
    hipFree(t->xyz);
    delete t;
}

int wa_bspline_create(wa_ctx *ctx, int32_t dim, int32_t degree, int32_t level_ini, int32_t level_fin,
                      int64_t n_middle, wa_bspline **out)
{
    if (!ctx || !out) return fail(ctx, WA_ERR_ARG, "wa_bspline_create: null argument");
    if (dim < 1 || dim > WA_BS_MAX_DIM) return fail(ctx, WA_ERR_ARG, "wa_bspline_create: dim must be 1..16");
    if (degree < 0 || degree > WA_BS_MAX_DEGREE) return fail(ctx, WA_ERR_ARG, "wa_bspline_create: degree must be 0..7");
    if (level_ini < 0 || level_fin < 0 || level_ini > degree || level_fin > degree)
        return fail(ctx, WA_ERR_ARG, "wa_bspline_create: constraint levels must be 0..degree");
    if (n_middle < 0) return fail(ctx, WA_ERR_ARG, "wa_bspline_create: n_middle < 0");
    int64_t nk = degree + n_middle + 2 + level_ini + level_fin + 1;   // BSplineBasic.h:38-39
    int64_t nc = n_middle + 2 + level_ini + level_fin;                // :40
    if (nk < 2 * (degree + 1)) return fail(ctx, WA_ERR_ARG, "wa_bspline_create: invalid setup (num_knots < 2*(degree+1))");
    wa_bspline *b = new wa_bspline();
    b->ctx = ctx;
    b->S.dim = dim; b->S.degree = degree; b->S.ci = level_ini; b->S.cf = level_fin;
    b->S.n_middle = n_middle; b->S.n_knots = nk; b->S.n_cps = nc;
    b->S.knots = nullptr; b->S.cps = nullptr; b->S.uninit = 0.0f;
    b->d_ends = nullptr;
    b->set = false;
    hipError_t h = dalloc(&b->S.knots, (size_t)nk);
    h = h ? h : dalloc(&b->S.cps, (size_t)nc * dim);
    h = h ? h : dalloc(&b->d_ends, (size_t)(level_ini + level_fin + 2) * dim);
    h = h ? h : hipMemsetAsync(b->S.knots, 0, sizeof(float) * nk, ctx->stream);         // constructor zero-fills (:44-51)
    h = h ? h : hipMemsetAsync(b->S.cps, 0, sizeof(float) * nc * dim, ctx->stream);
    h = h ? h : hipStreamSynchronize(ctx->stream);
    if (h != hipSuccess) { wa_bspline_destroy(b); return fail(ctx, WA_ERR_ALLOC, "wa_bspline_create: %s", hipGetErrorString(h)); }
    *out = b;
    return WA_OK;
}

void wa_bspline_destroy(wa_bspline *b)
{
    if (!b) return;
    hipFree(b->S.knots); hipFree(b->S.cps); hipFree(b->d_ends);
    delete b;
}

int wa_bspline_set_uninit(wa_bspline *b, uint32_t float_bits)
{
    if (!b) return WA_ERR_ARG;
    memcpy(&b->S.uninit, &float_bits, 4);
    return WA_OK;
}

static int bspline_setup(wa_bspline *b, const float *init, const float *fin, const float *d_middle, int64_t stride,
                         float fin_time)
{
    wa_ctx *ctx = b->ctx;
    const WaSpline &S = b->S;
    size_t ni = (size_t)(S.ci + 1) * S.dim, nf = (size_t)(S.cf + 1) * S.dim;
    std::vector<float> ends(ni + nf);
    memcpy(ends.data(), init, sizeof(float) * ni);
    memcpy(ends.data() + ni, fin, sizeof(float) * nf);
    HIPC(ctx, hipMemcpyAsync(b->d_ends, ends.data(), sizeof(float) * (ni + nf), hipMemcpyHostToDevice, ctx->stream));
    k_bspline_setup<<<1, 64, 0, ctx->stream>>>(S, b->d_ends, fin_time);
    HIPC(ctx, hipGetLastError());
    if (S.n_middle) {
        long long total = S.n_middle * S.dim;
        k_bspline_middle<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(S, d_middle, stride);
        HIPC(ctx, hipGetLastError());
    }
    HIPC(ctx, hipStreamSynchronize(ctx->stream));   // `ends` is host-stack staging
    b->set = true;
    return WA_OK;
}

int wa_bspline_set_param(wa_bspline *b, const float *init, const float *fin, const float *middle, int64_t stride,
                         float fin_time)
{
    if (!b) return WA_ERR_ARG;
    wa_ctx *ctx = b->ctx;
    if (!init || !fin || (b->S.n_middle > 0 && !middle)) return fail(ctx, WA_ERR_ARG, "wa_bspline_set_param: null argument");
    if (stride < b->S.dim) return fail(ctx, WA_ERR_ARG, "wa_bspline_set_param: stride < dim");
    if (!(fin_time > 0.0f) || !isfinite(fin_time)) return fail(ctx, WA_ERR_ARG, "wa_bspline_set_param: fin_time must be finite and > 0");
    float *d_mid = nullptr;
    if (b->S.n_middle) {
        if (dalloc(&d_mid, (size_t)b->S.n_middle * stride)) return fail(ctx, WA_ERR_ALLOC, "wa_bspline_set_param: staging");
        hipError_t h = hipMemcpyAsync(d_mid, middle, sizeof(float) * b->S.n_middle * stride, hipMemcpyHostToDevice, ctx->stream);
        if (h != hipSuccess) { hipFree(d_mid); return fail(ctx, WA_ERR_DEVICE, "wa_bspline_set_param: %s", hipGetErrorString(h)); }
    }
    int rc = bspline_setup(b, init, fin, d_mid, stride, fin_time);
    hipFree(d_mid);
    return rc;
}

int wa_bspline_set_param_traj(wa_bspline *b, const float *init, const float *fin, const wa_traj *middle, float fin_time)
{
    if (!b) return WA_ERR_ARG;
    wa_ctx *ctx = b->ctx;
    if (!init || !fin || !middle) return fail(ctx, WA_ERR_ARG, "wa_bspline_set_param_traj: null argument");
    if (b->S.dim != 3) return fail(ctx, WA_ERR_ARG, "wa_bspline_set_param_traj: dim must be 3");
    if (middle->n != b->S.n_middle) return fail(ctx, WA_ERR_ARG, "wa_bspline_set_param_traj: trajectory size != n_middle");
    if (!(fin_time > 0.0f) || !isfinite(fin_time)) return fail(ctx, WA_ERR_ARG, "wa_bspline_set_param_traj: fin_time must be finite and > 0");
    return bspline_setup(b, init, fin, middle->xyz, 3, fin_time);
}

int wa_bspline_info(const wa_bspline *b, int64_t *n_knots, int64_t *n_cps)
{
    if (!b) return WA_ERR_ARG;
    if (n_knots) *n_knots = b->S.n_knots;
    if (n_cps) *n_cps = b->S.n_cps;
    return WA_OK;
}

int wa_bspline_read(const wa_bspline *b, float *knots, float *cps)
{
    if (!b) return WA_ERR_ARG;
    if (knots) HIPC(b->ctx, hipMemcpy(knots, b->S.knots, sizeof(float) * b->S.n_knots, hipMemcpyDeviceToHost));
    if (cps) HIPC(b->ctx, hipMemcpy(cps, b->S.cps, sizeof(float) * b->S.n_cps * b->S.dim, hipMemcpyDeviceToHost));
    return WA_OK;
}

static hipError_t bspline_launch(wa_bspline *b, const float *d_u, float t0, float dt, int64_t count, int32_t der,
                                 float *d_out, uint8_t *d_ok)
{
    hipStream_t st = b->ctx->stream;
    unsigned blocks = (unsigned)((count + 255) / 256);
    switch (b->S.degree) {
#define WA_BS_CASE(D) case D: k_bspline_eval<D><<<blocks, 256, 0, st>>>(b->S, d_u, t0, dt, count, der, d_out, d_ok); break;
        WA_BS_CASE(0) WA_BS_CASE(1) WA_BS_CASE(2) WA_BS_CASE(3) WA_BS_CASE(4) WA_BS_CASE(5) WA_BS_CASE(6) WA_BS_CASE(7)
#undef WA_BS_CASE
    }
    return hipGetLastError();
}

static int bspline_run(wa_bspline *b, const float *u, float t0, float dt, int64_t count, int32_t der, float *out,
                       uint8_t *ok, wa_traj **out_traj, const char *who)
{
    wa_ctx *ctx = b->ctx;
    if (!b->set) return fail(ctx, WA_ERR_STATE, "%s: SetParam has not run", who);
    if (count < 0 || der < 0) return fail(ctx, WA_ERR_ARG, "%s: bad count / derivative level", who);
    if (out_traj && b->S.dim != 3) return fail(ctx, WA_ERR_ARG, "%s: a trajectory output needs dim == 3", who);
    if (out_traj) *out_traj = nullptr;
    if (count == 0) return out_traj ? traj_alloc(ctx, 0, out_traj) : WA_OK;
    const int dim = b->S.dim;
    float *d_u = nullptr, *d_out = nullptr;
    uint8_t *d_ok = nullptr;
    wa_traj *t = nullptr;
    hipError_t h = hipSuccess;
    if (out_traj) {
        int rc = traj_alloc(ctx, count, &t);
        if (rc) return rc;
        d_out = t->xyz;
    } else {
        h = dalloc(&d_out, (size_t)count * dim);
    }
    if (u) {
        h = h ? h : dalloc(&d_u, (size_t)count);
        h = h ? h : hipMemcpyAsync(d_u, u, sizeof(float) * count, hipMemcpyHostToDevice, ctx->stream);
    }
    if (ok) h = h ? h : dalloc(&d_ok, (size_t)count);
    h = h ? h : bspline_launch(b, d_u, t0, dt, count, der, d_out, d_ok);
    if (out) h = h ? h : hipMemcpyAsync(out, d_out, sizeof(float) * count * dim, hipMemcpyDeviceToHost, ctx->stream);
    if (ok) h = h ? h : hipMemcpyAsync(ok, d_ok, (size_t)count, hipMemcpyDeviceToHost, ctx->stream);
    h = h ? h : hipStreamSynchronize(ctx->stream);
    hipFree(d_u); hipFree(d_ok);
    if (!out_traj) hipFree(d_out);
    if (h != hipSuccess) {
        wa_traj_destroy(t);
        return fail(ctx, WA_ERR_DEVICE, "B-spline evaluation: %s", hipGetErrorString(h));
    }
    if (out_traj) *out_traj = t;
    return WA_OK;
}

int wa_bspline_eval(wa_bspline *b, const float *u, int64_t count, int32_t der, float *out, uint8_t *ok)
{
    if (!b) return WA_ERR_ARG;
    if (count > 0 && (!u || !out)) return fail(b->ctx, WA_ERR_ARG, "wa_bspline_eval: null argument");
    return bspline_run(b, u, 0.0f, 0.0f, count, der, out, ok, nullptr, "wa_bspline_eval");
}

int wa_bspline_sample(wa_bspline *b, float t0, float dt, int64_t count, int32_t der, float *out, uint8_t *ok,
                      wa_traj **out_traj)
{
    if (!b) return WA_ERR_ARG;
    return bspline_run(b, nullptr, t0, dt, count, der, out, ok, out_traj, "wa_bspline_sample");
}

}  // extern "C"
