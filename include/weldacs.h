/* include/weldacs.h -- C ABI of libweldacs.so (MI355X / gfx950 HIP backend).
 *
 * The reference (mhsitu/welding_robot) has no FFI: its planning path lives behind the public
 * members of header-only C++ classes.  This header is the boundary a maintainer
 * binds instead; each entry point names the reference interface it replaces.  The drop-in C++
 * headers under welding_robot_amd/include/core/ (same class names as the reference) are written
 * purely on top of this ABI, and tests/bench reach it through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, caller-allocated outputs, opaque handles.
 *   - every function returns a wa_status (0 = ok) unless it returns a count; wa_last_error()
 *     gives the text.  The library never exit()s and never prints (the reference does both:
 *     read_STL.hpp:34-59, ACSRank_3D.hpp:239).
 *   - there is NO CPU fallback: with no HIP device wa_ctx_create fails with WA_ERR_DEVICE.
 *   - lattice: voxel id = (z*ny + y)*nx + x  (model_grid_map.hpp:203-216);
 *     edge k of a voxel: 0:z-1 1:y-1 2:x-1 3:x+1 4:y+1 5:z+1  (ACSRank_3D.hpp:355-365).
 *   - a wa_ctx and everything created from it is used from one host thread at a time.
 */
#ifndef WELDACS_H
#define WELDACS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    WA_OK = 0,
    WA_ERR_ARG = 1,       /* bad argument / null pointer / size                          */
    WA_ERR_DEVICE = 2,    /* no HIP device, HIP runtime error                            */
    WA_ERR_ALLOC = 3,     /* host or device allocation failed                            */
    WA_ERR_FILE = 4,      /* cannot open / short read (read_STL.hpp:34-59 exit(1..3))    */
    WA_ERR_FORMAT = 5,    /* an ASCII STL the reference's reader never returns from     */
    WA_ERR_POINT = 6,     /* a route point resolves to no free voxel (ACSRank_3D.hpp:491) */
    WA_ERR_CAPACITY = 7,  /* a walk outgrew path_capacity / more ants than max_colony     */
    WA_ERR_STATE = 8      /* call order (e.g. results before a solve)                     */
} wa_status;

typedef struct wa_ctx wa_ctx;
typedef struct wa_grid wa_grid;
typedef struct wa_acs wa_acs;
typedef struct wa_traj wa_traj;       /* device-resident polyline, n x 3 floats */
typedef struct wa_bspline wa_bspline;

/* ---- context --------------------------------------------------------------------------- */
const char *wa_version(void);
/* visible HIP devices (0 when there is none or no runtime): the drop-in ACS_Rank shards its pair searches over all of
 * them, one wa_ctx per device (ACSRank_3D.hpp:472-499 is a loop over independent searches) */
int wa_device_count(void);
/* free / total memory of the context's device in bytes (either pointer may be NULL): what the slot count of a solver for many
 * pair searches is sized by, and what a caller can poll after another process has just released the GPU */
int wa_ctx_memory_info(wa_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);
/* A context keeps the device memory (blocks of 1 MiB and up) of the solvers destroyed on it and builds the next solver from it: the
 * reference never frees anything (SURVEY 8(b) ownership; ACSRank_3D.hpp:456-460 allocates once) and creates its search once per
 * process; a host that runs searchBestPathOfPoints per job would otherwise pay the driver's wipe of the freed memory before every
 * re-allocation (seconds for a C5-sized solver).  Where the device supports virtual memory management the memory is kept as an ARENA
 * of physical chunks (512 MiB and 32 MiB) that are mapped into a fresh address range for every block of 32 MiB and up, so a solver of
 * ANY shape is served from what solvers of other shapes gave back; smaller blocks, and every block without that support
 * (WA_DEV_ARENA=0), are whole allocations reused for requests of nearly their size.  Kept bytes count as free in wa_ctx_memory_info;
 * at most WA_DEV_KEEP_PCT (95) percent of the device's memory is kept; when the device runs out, kept memory goes back to the driver
 * -- as much as is needed --, as it does on wa_ctx_trim and with the context.  Other PROCESSES on the same GPU cannot have those bytes
 * until then: a host that shares its GPU trims after its last job.  WA_DEV_CACHE=0 in the environment switches the mechanism off. */
int wa_ctx_cached_bytes(wa_ctx *ctx, int64_t *bytes);
int wa_ctx_trim(wa_ctx *ctx);
/* counters since the context was created: [0] blocks asked for (1 MiB and up), [1] bytes served from kept memory, [2] bytes newly
 * obtained from the driver, [3] bytes given back to the driver, [4] blocks served ENTIRELY from kept memory, [5] out-of-memory events
 * handled by releasing kept memory, [6] milliseconds spent building arena blocks, [7] 1 when the arena is in use, 2 when it is but its
 * address window is used up (blocks kept so far go on serving, new shapes come as whole allocations) */
int wa_ctx_cache_stats(wa_ctx *ctx, int64_t out[8]);
/* Every call on a context (and on anything created from it) runs on that context's device regardless of the calling
 * thread's current HIP device, and restores the caller's current device before returning. */
int wa_ctx_create(int device_ordinal, wa_ctx **out);
void wa_ctx_destroy(wa_ctx *ctx);
const char *wa_last_error(const wa_ctx *ctx);
int wa_ctx_device_name(const wa_ctx *ctx, char *buf, size_t cap);
int wa_ctx_sync(wa_ctx *ctx);
/* the HIP stream every kernel of this context is launched on (hipStream_t as void*) */
void *wa_ctx_stream(wa_ctx *ctx);

/* ---- mesh input: replaces STLReader::readFile + TriangleList (read_STL.hpp:26-77,:88,:131-156)
 * tris = n x 12 floats (normal, v0, v1, v2).  Returns the triangle count (>= 0) or -wa_status.
 * Pass tris = NULL to query the count.  Byte 79 != 0 selects the ASCII branch (:65-68, :99-129), read with the reference's stream
 * semantics: normals stay (0, 0, 0), an unreadable vertex keeps the previous triangle's value, a text without a
 * "facet" word where the reader expects one gives 0 triangles.  WA_ERR_FORMAT: the text ends directly behind a "facet" word (the
 * reference's loop does not end on it); WA_ERR_FILE: shorter than the 80 bytes the sniff reads / a truncated binary file. */
int64_t wa_stl_parse(const void *buf, size_t len, float *tris, int64_t cap_tris);
int64_t wa_stl_read_file(const char *path, float *tris, int64_t cap_tris);

/* ---- grid map: replaces GridMap<float>::creatGridMap / readGridMap / ptr_grid_map
 *      (model_grid_map.hpp:151-298, :300-356, :358) -------------------------------------- */
/* voxelise a mesh on the device (bbox :165-181, ranges :198-200, coords :204-211, occupancy
 * :223-268).  bbox6_out (optional) = min xyz, max xyz of the mesh. */
int wa_grid_from_mesh(wa_ctx *ctx, const float *tris, int64_t n_tris, float precision, int32_t wall,
                      wa_grid **out, float *bbox6_out);
/* adopt an existing occupancy (free_[id] != 0 <=> isFree) with per-axis node coordinates --
 * the readGridMap path and the synthetic benchmark grids */
int wa_grid_from_occupancy(wa_ctx *ctx, const uint8_t *free_, int32_t nx, int32_t ny, int32_t nz,
                           const float *cx, const float *cy, const float *cz, float precision,
                           int32_t wall, wa_grid **out);
/* model_grid_map.hpp:204-211 / :321-328 piecewise axis coordinates (host helper, n values) */
int wa_axis_coords(float lo, float hi, float precision, int32_t wall, int32_t n, float *out);
void wa_grid_destroy(wa_grid *g);
int wa_grid_info(const wa_grid *g, int32_t dims3[3], float *precision, int32_t *wall, int64_t *n_free);
int wa_grid_read_occupancy(const wa_grid *g, uint8_t *free_out /* nx*ny*nz */);
int wa_grid_read_coords(const wa_grid *g, float *cx, float *cy, float *cz);
/* ACS_Rank::setPoints / checkRoutePoints (ACSRank_3D.hpp:537-565, :511-535): for each point the
 * LAST free voxel in raster order within +-(float)(1.2*precision) on every axis; -1 if none.  Calls with up to 16 points (setPoints
 * resolves two) are answered from a host mirror of the grid's axis tables and occupancy, fetched once per grid (0.2 us per call instead
 * of a launch + synchronise + copy); larger batches by the kernel; WA_RESOLVE_HOST=0: always the kernel.  Same ids either way. */
int wa_grid_resolve_points(const wa_grid *g, const float *pts_xyz, int32_t n_pts, int64_t *ids_out);

/* ---- rank-based ACS: replaces ACS_Rank::initFromGridMap / computeSolution / reset /
 *      searchBestPathOfPoints' pair loop / getSolution (ACSRank_3D.hpp:220-504) ----------- */
enum { WA_RNG_REF = 0, WA_RNG_DEV = 1 };
/* WA_RNG_REF: glibc rand() stream shared sequentially by all ants and all problems, libstdc++
 *             std::sort tie order -- bit-identical to the reference, ants walk one after another
 *             (a parity mode, one wavefront per problem).
 * WA_RNG_DEV: counter-based draw keyed (seed, stream, generation, ant, step), ranks ties by ant
 *             index -- the parallel production mode, one wavefront per ant. */
typedef struct {
    int32_t alpha;          /* pheromone exponent, reference 1       (ACSRank_3D.hpp:319) */
    float beta;             /* heuristic weight, reference 0.6       (:320)               */
    float rho;              /* evaporation factor, reference 0.8     (:321)               */
    float pheromone_0;      /* reference 1                           (:324)               */
    int32_t max_iteration;  /* generations, reference 150            (:322)               */
    float predict;          /* computeSolution(predict_path_len)     (:220)               */
    int32_t fixed_colony;   /* 0: reference-adaptive colony (:247); >0: pinned ant count   */
    int32_t rng_mode;       /* WA_RNG_REF / WA_RNG_DEV                                     */
    uint64_t seed;          /* DEV counter key (REF: use wa_acs_srand)                     */
} wa_acs_params;
void wa_acs_default_params(wa_acs_params *p);

/* n_slots independent problems can be in flight.  Each owns a pheromone field (24 B/voxel, two of them with the dense
 * sweep: it is out of place), best-path stamps (8 B/voxel), deposit rank masks (6 B/voxel when max_colony <= 35 --
 * at most 8 ranks deposit, ACSRank_3D.hpp:200 -- 48 B/voxel otherwise) and its ants' paths (4 B * max_colony *
 * path_capacity); the heuristic fields (24 B/voxel) are a pool shared by the slots: one per distinct END point in use,
 * computed by wa_acs_begin when the pool does not hold it yet.  max_colony bounds the ants per generation,
 * path_capacity the nodes per walk (<= number of voxels). */
int wa_acs_create(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony,
                  int64_t path_capacity, wa_acs **out);
/* Same with an explicit neighbourhood: 6 = face neighbours (what wa_acs_create builds, the reference as
 * shipped), 26 = faces + edges + corners -- the variant the reference's initFromGridMap walks but stubs
 * out by setting the two extra distances to 0 (ACSRank_3D.hpp:361-388); here they carry the values the
 * reference keeps in comments (precision*1.414f, precision*1.732f) and evaporation / reset cover all 26
 * edges.  Edge index k of a voxel follows the reference's cube loop: offsets z, y, x in {-1,0,1}, z outermost,
 * centre skipped (k = 0 is (-1,-1,-1), k = 25 is (+1,+1,+1)).  Fields are [N][26]; grids up to 2^27 voxels.
 * Every other wa_acs_* call is unchanged; wa_acs_read_pheromone then fills nvox*26 floats and
 * wa_acs_result's `choices` are 0..25. */
int wa_acs_create_nb(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony,
                     int64_t path_capacity, int32_t neighbourhood, wa_acs **out);
/* Same results as wa_acs_create, evaporation evaluated LAZILY: a voxel none of whose outgoing edges ever received a
 * deposit is never swept -- after g evaporations each of its in-bounds edges holds pheromone_0*rho*...*rho in the
 * reference's own fp32 rounding, which the solver carries as one scalar -- and only the voxels on deposited paths
 * are multiplied by rho every generation (ACSRank_3D.hpp:268-272 restricted to where it can differ).  Every value
 * any kernel reads, and wa_acs_read_pheromone's output, is bit-identical to the dense sweep.  A generation then
 * costs O(deposited voxels) instead of 48 B/voxel and reset_pheromone O(deposited voxels) instead of 24 B/voxel:
 * meant for many pair searches on large grids (BASELINE config C5).  DEV mode, <= 2048 ants and
 * <= 64 depositing ranks only (WA_ERR_ARG at wa_acs_begin otherwise); wa_acs_evaporate is not available. */
int wa_acs_create_lazy(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony,
                       int64_t path_capacity, wa_acs **out);
/* ... with 6 or 26 neighbours (round 5): the 26-neighbour variant the reference stubs out (ACSRank_3D.hpp:361-388) swept 26 x 8 B x N
 * per search and generation in its dense form -- 3.5 GB at 256^3 -- which kept it a small-grid feature; lazily evaporated it runs at
 * pair-planning scale.  Same restrictions as wa_acs_create_lazy. */
int wa_acs_create_lazy_nb(wa_ctx *ctx, const wa_grid *grid, int32_t n_slots, int32_t max_colony,
                          int64_t path_capacity, int32_t neighbourhood, wa_acs **out);
void wa_acs_destroy(wa_acs *s);
/* device bytes a solver of this shape takes: per slot, per heuristic field (the pool holds one per distinct END point of a
 * batch, at least min(n_slots, 4) -- from 32 slots on n_slots / 8, between 8 and 24) and once per solver; the per-generation trace (20 B per slot and generation) comes on top.
 * Nothing is allocated.  The drop-in ACS_Rank sizes the concurrent pair searches of a device from this and
 * wa_ctx_memory_info (ACSRank_3D.hpp:472-499 runs them one after another).  The straggler pools of small dense solvers come on top:
 * wa_acs_straggler_pool_bytes. */
int wa_acs_memory_estimate(const wa_grid *grid, int32_t max_colony, int64_t path_capacity, int32_t neighbourhood, int32_t lazy,
                           int64_t *bytes_per_slot, int64_t *bytes_per_heuristic_field, int64_t *bytes_fixed);
/* A dense solver (6 or 26 neighbours) of at most 256 ants and at most 16 slots (WA_STRAGGLER_SLOTS) additionally holds, PER SLOT, the
 * arrival list and what the straggler hand-over needs (see wa_acs_debug_counters): a second array of ants' paths (max_colony x
 * path_capacity words: stragglers walk on in place while the next generation writes the other array) + 256 spill-bitmap rows
 * (0.33 GB per slot at 128^3 with 256 ants and the default path capacity).  *bytes = what a solver of this shape holds in total, 0 when it
 * gets none (lazy, larger colonies, more slots, WA_STRAGGLERS=0). */
int wa_acs_straggler_pool_bytes(const wa_grid *grid, int32_t n_slots, int32_t max_colony, int64_t path_capacity, int32_t neighbourhood,
                                int32_t lazy, int64_t *bytes);
/* initFromGridMap :343-408: in-bounds edges pheromone_0, out-of-bounds edges 0. slot<0: all */
int wa_acs_init_pheromone(wa_acs *s, int32_t slot, float pheromone_0);
/* reset() :307-315: every edge pheromone_0 */
int wa_acs_reset_pheromone(wa_acs *s, int32_t slot, float pheromone_0);
/* REF mode libc state: srand(seed) (:327); get/set = 34 words + 2 indices, to hand the stream
 * on to wa_gtsp_solve exactly as the reference's process-global rand() does */
int wa_acs_srand(wa_acs *s, uint32_t seed);
int wa_acs_rand_state(wa_acs *s, int32_t state36_inout[36], int32_t set);

/* start n_problems searches on slots 0..n-1 (setPoints already resolved to voxel ids; streams[i]
 * is the DEV-mode stream key of problem i, NULL = 0..n-1), then advance all of them by
 * n_generations (enqueued on the context stream, asynchronous), then wait. */
int wa_acs_begin(wa_acs *s, const wa_acs_params *p, int32_t n_problems, const int64_t *start_ids,
                 const int64_t *end_ids, const uint32_t *streams);
int wa_acs_run(wa_acs *s, int32_t n_generations);
int wa_acs_sync(wa_acs *s);
/* Pipelined groups.  The searches of a batch are independent (ACSRank_3D.hpp:472-499 is a loop over searches with a pheromone
 * reset in between, :481): wa_acs_run splits the active slots into `groups` contiguous groups that advance on HIP streams of
 * their own, so that one group's HBM-bound evaporation sweep (:268-272) runs under another group's latency-bound walk
 * (:252-261).  The groups are forked from the context's stream when wa_acs_run starts and joined into it before it returns,
 * so everything enqueued on the context afterwards is ordered behind all of them.  Every slot's own launch sequence is the
 * single-stream one: results do not depend on the split.  groups = 0: by rule (WA_PIPE_GROUPS, read at creation, overrides
 * the rule); 1: everything on the context's stream.  wa_acs_pipeline_info: the groups the last wa_acs_run used. */
int wa_acs_set_pipeline(wa_acs *s, int32_t groups);
int wa_acs_pipeline_info(const wa_acs *s, int32_t *groups_last_run);
/* begin + run(max_iteration) + sync */
int wa_acs_solve(wa_acs *s, const wa_acs_params *p, int32_t n_problems, const int64_t *start_ids,
                 const int64_t *end_ids, const uint32_t *streams);

/* best-so-far of a slot (Agent<float> best: L, path ids, edge choices = nodeIndex()).  When no
 * ant has arrived cost = +inf (a valid result) and len = 0.  The first read behind a run (wa_acs_run / wa_acs_solve)
 * fetches every slot's result to the host in one go; wa_acs_result / wa_acs_result_batch* are served from that copy until the next
 * wa_acs_begin / wa_acs_run (a per-slot read in a host loop, getSolution :506-509, costs a host array access). */
int wa_acs_result(wa_acs *s, int32_t slot, float *cost, int64_t *len, int32_t *path_ids,
                  int8_t *choices, int64_t cap);
/* the same for slots 0 .. n_slots-1 in one round trip (batched pair searches): costs[q], lens[q] (0 when the cost is
 * +inf), and -- if path_ids is not NULL -- slot q's path ids at path_ids[q*path_stride ...]; WA_ERR_CAPACITY when a
 * path is longer than path_stride (lens is filled in either way: call again with a larger buffer). */
int wa_acs_result_batch(wa_acs *s, int32_t n_slots, float *costs, int64_t *lens, int32_t *path_ids, int64_t path_stride);
/* ... and the edge choices (Agent::nodeIndex(), as wa_acs_result's `choices`): slot q's lens[q] - 1 edge indices at choices[q*path_stride ...] */
int wa_acs_result_batch_choices(wa_acs *s, int32_t n_slots, float *costs, int64_t *lens, int32_t *path_ids, int8_t *choices, int64_t path_stride);
/* per-generation history the reference computes and discards (:295-296).  Arrays of
 * generations_done entries; any pointer may be NULL. */
int wa_acs_trace(wa_acs *s, int32_t slot, int32_t *generations_done, float *best_L, float *iter_best_L,
                 int32_t *colony, int32_t *finite, int64_t *steps);
/* device-to-device copy (on the context stream) of best_L[gen0 .. gen0+count) of every active
 * slot into dst_device[slot*count + g] -- feeds the RCCL MIN all-reduce of the global best */
int wa_acs_export_trace(wa_acs *s, void *dst_device, int32_t gen0, int32_t count);
int wa_acs_read_pheromone(wa_acs *s, int32_t slot, float *out /* nvox*6 */);
/* the agents[] of the generation walked last (ACSRank_3D.hpp:251-261): per ant L (+inf = dead end, :88-91) and node
 * count (Agent::getPath()->size()); *colony = ants of that generation, of which min(colony, cap) are written. */
int wa_acs_read_ants(wa_acs *s, int32_t slot, int32_t *colony, float *L, int32_t *len, int32_t cap);
/* the node ids one ant of that generation visited, start first (Agent::getPath(), ACSRank_3D.hpp:34,75-77): *len = node
 * count, of which min(*len, cap) ids are written. */
int wa_acs_read_ant_path(wa_acs *s, int32_t slot, int32_t ant, int32_t *ids, int32_t cap, int32_t *len);
int wa_acs_last_params(wa_acs *s, int32_t slot, int32_t *colony, float *lambda, float *Q);

/* kernel timing with HIP events on the context stream.  Enable before wa_acs_run; afterwards
 * ms[i]/launches[i] hold the summed event time and launch count of kernel class i. */
enum { WA_K_WALK = 0, WA_K_RANK = 1, WA_K_EVAPORATE = 2, WA_K_DEPOSIT = 3, WA_K_COUNT = 4 };
/* enable: 0 off; 1 every sample_every-th generation has all its launches stamped; 3 = the same, and the launch that carries the
 * evaporation sweep is stamped in EVERY generation (per-dispatch start/stop events of hipExtLaunchKernelGGL: no extra stream
 * operation, so the timed loop is not perturbed -- what bench.py's roofline figure uses); bit 2 (value 4) added to either: every
 * stamped sweep-carrying launch is preceded by a no-op dispatch with events of its own -- a dispatch with events directly behind one
 * without (the walk) reports 1.6-2 us that belong to that transition, not to the kernel (profiles/r06/sweep_gap.txt) */
int wa_acs_profile(wa_acs *s, int32_t enable, int32_t sample_every);
/* what the last DEV walk launch of a 6-neighbour solver ran with (the visited set of ACSRank_3D.hpp:70, :144-146 lives in LDS, one table per
 * walking ant): [0] log2 of the table's slots, [1] 1 if it kept 16-bit entries (saturated launches of grids whose voxel ids an entry can name;
 * WA_TAB16=0 in the environment switches them off, =1 forces them), [2] bytes of LDS per walk block -- min(163840 / [2], 16) blocks are resident
 * per CU --, [3] 1 if the loop carried touch loads (a lone search) */
int wa_acs_walk_info(const wa_acs *s, int32_t out[4]);
int wa_acs_profile_read(wa_acs *s, double ms[WA_K_COUNT], int64_t launches[WA_K_COUNT]);
/* diagnostic counters.  Product build: out16[9] = ants handed over as stragglers, out16[7] = stragglers finished by a resume block (the two
 * are equal after wa_acs_run returns), out16[6] = REF-mode ants confirmed by the converged-colony speculation (below), everything else zero; the cycle counters of the walk's inner loop only exist in the diagnostic
 * builds (-DWA_STAMPS / -DWA_ANT_TIME, tools/).
 * Stragglers (dense searches of at most 256 ants in solvers of at most 16 slots, 6 or 26 neighbours, DEV mode, alpha == 1, the first 64
 * generations of a search): only the ranks o <= lambda - 1 deposit (ACSRank_3D.hpp:200) and only the shortest ant can become the best path
 * (:263-264), so an ant that is already longer than floor(lambda - 1) + 1 arrivals of its generation (26 neighbours: whose L so far
 * already exceeds theirs) can change neither (about 140 of 256 ants per exploratory generation); at one of the loop's checks (every 64
 * nodes, every 16 once shorter ants have arrived) it leaves the walk launch -- which lasts as long as its longest ant -- and a resume
 * block of the NEXT generation's walk launch finishes the same walk on the previous generation's field (intact until the next sweep),
 * adding its arrival and its steps to its own generation's trace entry.  Lists and pools are per slot: every search of a batch hands
 * its own stragglers over.  The last generation of a wa_acs_run call hands over too once calls have been seen to follow each other
 * without a read in between (chunked runs, generation-by-generation loops; behind a lone call it would only add a launch to what the
 * caller waits for); its stragglers are finished by the next call's first walk launch or -- when results are read first
 * (wa_acs_sync, wa_acs_result, wa_acs_trace, wa_acs_read_ants ...) -- by a launch of resume blocks only, which also puts the finished
 * walks back into agents[] (WA_STRAGGLER_DRAIN=0: the last generation of a call never hands over, as in round 3).  agents[] and the
 * trace are complete whenever they are read.  Results are bit-identical with the mechanism on or off (WA_STRAGGLERS=0, read at
 * wa_acs_create; wa_acs_set_stragglers(s, 0) at run time).
 * REF mode once the colony has converged (6 neighbours, alpha == 1; WA_REF_SPEC=0 switches it off): the shared libc stream forces the ants
 * to walk one after another because an ant's first draw is the previous ants' total step count -- but when every ant re-walks the best path
 * that count is known.  The stream of the whole generation is generated ahead, every ant checks IN PARALLEL (replay table) that it follows
 * the whole best path with the draws it would be dealt, and the sequential walk starts at the first ant that does not, with the stream taken
 * to exactly that ant's first draw.  Same draws, same results, same stream position as the reference (BASELINE config 3, 500 REF generations: 8.7 -> 1.25 s). */
int wa_acs_debug_counters(wa_acs *s, uint64_t out16[16], int32_t reset);
/* the same two counts per slot (ants handed over / stragglers finished by a resume block; equal whenever they are read) */
int wa_acs_straggler_counters(wa_acs *s, int32_t slot, uint64_t *handed_over, uint64_t *resumed, int32_t reset);
/* generations of a search during which its ants may be handed over (default 64, WA_STRAGGLER_GENS; 0: off; < 0: back to the default) */
int wa_acs_set_stragglers(wa_acs *s, int32_t generations);
/* evaporation sweep alone (ACSRank_3D.hpp:268-272) over `slot` -- for roofline measurements */
int wa_acs_evaporate(wa_acs *s, int32_t slot, float rho, int32_t repeats);

/* ---- multi-GPU: problems shard one set per GPU (main.cpp:268-283 is a loop over independent searches, and so is a
 *      multi-start batch); the only exchange is the global-best path cost per generation -- what each rank's
 *      ACSRank_3D.hpp:263-264 would publish -- as an RCCL all-reduce (ncclMin) over xGMI.  One wa_comm per wa_ctx
 *      (= per device, per process or host thread); libweldacs.so links librccl itself, no torch / MPI needed. ---- */
#define WA_COMM_ID_BYTES 128
typedef struct wa_comm wa_comm;
/* rank 0 creates the id (ncclGetUniqueId) and ships the bytes to the other ranks by any means it has
 * (a file, a socket, MPI_Bcast, a torch.distributed store ...) */
int wa_comm_unique_id(uint8_t id_out[WA_COMM_ID_BYTES]);
/* collective over all `world` ranks (ncclCommInitRank); the communicator runs its exchanges on its own stream */
int wa_comm_create(wa_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[WA_COMM_ID_BYTES], wa_comm **out);
void wa_comm_destroy(wa_comm *c);
int wa_comm_info(const wa_comm *c, int32_t *rank, int32_t *world);
/* Every wait of the exchanges below is bounded and watches the communicator (ncclCommGetAsyncError): a peer that died, or that does not
 * answer within WA_COMM_TIMEOUT_S seconds (environment, read by wa_comm_create; default 600, 0 = wait for ever), makes the call give the
 * communicator up (ncclCommAbort) and return WA_ERR_DEVICE instead of blocking; every later call on it returns WA_ERR_STATE.
 * wa_comm_abort does the same on request (a host that has learnt by other means that a peer is gone); wa_comm_destroy is still due.
 * The reference has no counterpart (its pair loop is one process: ACSRank_3D.hpp:472-499); the convention is SURVEY 8(b)'s: status
 * codes, never a hang.
 * wa_comm_stats: [0] ranks as RCCL counts them (ncclCommCount), [1] RCCL's version code (ncclGetVersion), [2] all-reduce calls issued on
 * this communicator, [3] other collectives / sends / receives issued, [4] 1 once the communicator has been aborted. */
int wa_comm_abort(wa_comm *c);
int wa_comm_stats(wa_comm *c, int64_t out[5]);
/* global_best[g] = MIN over all ranks and all active slots of best_L[g] for g in [gen0, gen0 + count), together with WHO holds it:
 * one ncclAllReduce(ncclUint64, ncclMin) of the packed key (float bits of the cost << 32 | rank << 16 | slot; costs are non-negative
 * or +inf, so the bits order like the values; ties go to the lowest rank, then the lowest slot).  Asynchronous -- waits (event) for
 * the generations enqueued so far, runs on the communicator's stream beside the generations enqueued afterwards.  Every rank must
 * call it with the same (gen0, count) sequence.  The global best is published, never fed back into a problem's own colony / Q, so
 * per-problem results equal the single-GPU run.  At most 65 536 active slots per rank. */
int wa_acs_allreduce_best(wa_acs *s, wa_comm *c, int32_t gen0, int32_t count);
/* wait for the exchanges enqueued so far, then copy global_best[gen0 .. gen0 + count) to the host; WA_ERR_STATE for a generation that
 * has not been through wa_acs_allreduce_best.  ..._owner: also the rank and the slot whose search holds that cost (its path:
 * wa_acs_result on that rank; wa_comm_gather_paths brings it to one rank); any output may be NULL. */
int wa_comm_read_best(wa_comm *c, int32_t gen0, int32_t count, float *out);
int wa_comm_read_best_owner(wa_comm *c, int32_t gen0, int32_t count, float *cost, int32_t *owner_rank, int32_t *owner_slot);
/* the packed key itself, for a host that runs the reduction through another transport (host-side helpers, no device work) */
int wa_comm_pack_best_key(float cost, int32_t rank, int32_t slot, uint64_t *key);
int wa_comm_unpack_best_key(uint64_t key, float *cost, int32_t *rank, int32_t *slot);
/* End of a sharded pair-planning run (every rank ran its share of ACSRank_3D.hpp:472-499; ACS_GTSP.hpp:224-253 wants all costs,
 * :286-298 all paths on one rank).  Both are blocking and size-prefixed (an all-gather of the counts, then the padded records).
 * wa_comm_allgather_costs: all[index_mine[i]] = cost_mine[i] for every rank's pairs, on every rank (entries nobody owns keep what
 * the caller put there).
 * wa_comm_gather_paths: rank `root` receives every rank's paths (path i: global index index_mine[i], len_mine[i] node ids, all ids
 * back to back in ids_mine) by ncclSend / ncclRecv; on the root the totals come back and wa_comm_gathered_paths_read copies them out
 * in rank order (index_out / len_out: n_paths_total entries, ids_out: n_ids_total); on the other ranks the totals are 0. */
int wa_comm_allgather_costs(wa_comm *c, int32_t n_mine, const int32_t *index_mine, const float *cost_mine, int32_t n_total, float *all);
int wa_comm_gather_paths(wa_comm *c, int32_t root, int32_t n_mine, const int32_t *index_mine, const int64_t *len_mine, const int32_t *ids_mine,
                         int64_t *n_paths_total, int64_t *n_ids_total);
int wa_comm_gathered_paths_read(wa_comm *c, int32_t *index_out, int64_t *len_out, int32_t *ids_out);
/* paths / node ids the last wa_comm_gather_paths left on this rank (the root: every rank's; elsewhere 0): what the three buffers of
 * wa_comm_gathered_paths_read must hold.  A gather that failed leaves nothing readable (0, 0). */
int wa_comm_gathered_paths_counts(const wa_comm *c, int64_t *n_paths, int64_t *n_ids);
/* The occupancy grid to every rank, once: rank `root` built its grid -- wa_grid_from_mesh is the reference's O(triangles x
 * voxels) creatGridMap, model_grid_map.hpp:223-268 -- and every other rank receives a replica (occupancy + the three axis tables:
 * ncclBroadcast over xGMI; 16 MiB at 256^3) instead of repeating that step.  `grid`: the root's grid (ignored elsewhere).  *out: a new
 * grid owned by the caller on every rank but the root, NULL on the root (which keeps using its own).  Collective: every rank calls it;
 * a rank whose arguments are bad still takes part in the header exchange, so that all ranks return an error together. */
int wa_comm_broadcast_grid(wa_comm *c, int32_t root, const wa_grid *grid, wa_grid **out);
/* bookkeeping helpers for a C++ launcher (timing max over ranks, totals): blocking all-reduce of host doubles */
enum { WA_COMM_MIN = 0, WA_COMM_MAX = 1, WA_COMM_SUM = 2 };
int wa_comm_allreduce_f64(wa_comm *c, double *inout, int32_t count, int32_t op);
int wa_comm_barrier(wa_comm *c);

/* ---- weld-seam ordering: replaces ACS_GTSP::readFromGraphFile's init + computeSolution
 *      (ACS_GTSP.hpp:187-218, :224-253, :255-284) ----------------------------------------- */
typedef struct {
    int32_t rng_mode;        /* WA_RNG_REF / WA_RNG_DEV */
    uint64_t seed;           /* DEV */
    uint32_t stream;         /* DEV */
    int32_t max_iterations;  /* <= 0: city_num^2 (:216) */
} wa_gtsp_params;
/* dist = n x n row-major symmetric doubles (diagonal ignored), cnt = distance count of the graph
 * header (:229).  n_instances problems laid out back to back (dist, tour_edges 2*n each, cost,
 * iters).  REF mode: rand_state36 (inout, may be NULL -> srand(1)) continues the libc stream
 * and n_instances must be 1. */
int wa_gtsp_solve(wa_ctx *ctx, const double *dist, int32_t n, int32_t cnt, int32_t n_instances,
                  const wa_gtsp_params *p, int32_t *rand_state36, int32_t *tour_edges,
                  double *tour_cost, int32_t *iterations, double *pheromone_out);

/* ---- path post-processing: what main.cpp:283-352 does with the planned path ----
 * wa_traj_stitch replaces ACS_GTSP::read_all_segments / read_segment (ACS_GTSP.hpp:286-312): segment s
 * is the node ids seg_ids[seg_off[s] .. seg_off[s+1]) (n_seg+1 offsets); coordinates come from the
 * grid's axis tables on the device.  reverse (may be NULL) flips individual segments -- the reference
 * appends best_matrix[i][j] as stored even when the tour runs j -> i. */
int wa_traj_stitch(const wa_grid *g, const int64_t *seg_ids, const int64_t *seg_off, int32_t n_seg,
                   const uint8_t *reverse, wa_traj **out);
int wa_traj_from_points(wa_ctx *ctx, const float *xyz, int64_t n, wa_traj **out);
int64_t wa_traj_size(const wa_traj *t);
int wa_traj_read(const wa_traj *t, float *xyz /* n x 3 */);
void wa_traj_destroy(wa_traj *t);

/* BS_Basic<float, DIM, DEGREE, CONST_LEVEL_INI, CONST_LEVEL_FIN> (BSplineBasic.h:33-56); the template
 * arguments are run-time values.  WA_ERR_ARG where the reference indexes out of bounds: dim outside
 * 1..16, degree outside 0..7, a constraint level above the degree, or NumKnots < 2*(DEGREE+1) (:53-55). */
int wa_bspline_create(wa_ctx *ctx, int32_t dim, int32_t degree, int32_t level_ini, int32_t level_fin,
                      int64_t n_middle, wa_bspline **out);
void wa_bspline_destroy(wa_bspline *b);
/* The reference reads heap cells it never wrote when level_fin + 1 > degree (c_mat[idx][CL+1],
 * BSplineBasic.h:414-431 -- BS_Basic<float,3,2,2,2> at main.cpp:337 does): their value is an input
 * here, as a 32-bit float pattern.  Default 0 (a fresh zeroed heap). */
int wa_bspline_set_uninit(wa_bspline *b, uint32_t float_bits);
/* SetParam (:72-78).  init / fin: (level+1) x dim floats (position, velocity, acceleration ...);
 * middle: n_middle rows of `stride` floats, the first dim of each are used (:458-464).  fin_time > 0. */
int wa_bspline_set_param(wa_bspline *b, const float *init, const float *fin, const float *middle,
                         int64_t stride, float fin_time);
/* same with the middle points already on the device (dim must be 3, wa_traj_size == n_middle) */
int wa_bspline_set_param_traj(wa_bspline *b, const float *init, const float *fin, const wa_traj *middle,
                              float fin_time);
int wa_bspline_info(const wa_bspline *b, int64_t *n_knots, int64_t *n_cps);
int wa_bspline_read(const wa_bspline *b, float *knots, float *cps /* n_cps x dim */);
/* getCurvePoint (:87-112, der = 0) / getCurveDerPoint (:122-146, der >= 1) for `count` times at once.
 * ok[i] (may be NULL) = the reference's bool result; rows with ok = 0 are zero-filled (the reference
 * leaves `ret` untouched). */
int wa_bspline_eval(wa_bspline *b, const float *u, int64_t count, int32_t der, float *out /* count x dim */,
                    uint8_t *ok);
/* ONE time, evaluated on the host from a mirror of the knots and control points (fetched once per SetParam) by the same fp32
 * operations in the same order as the kernel: bit-identical to wa_bspline_eval, ~0.1-0.2 us per call instead of a launch +
 * synchronise + copy.  What the drop-in BS_Basic::getCurvePoint / getCurveDerPoint use: main.cpp:302-316 / :341-351 call them once per
 * sample inside clock()-paced loops whose sample count depends on how long a call takes.  out: dim floats (zeros when *ok = 0). */
int wa_bspline_eval_host(wa_bspline *b, float u, int32_t der, float *out /* dim */, uint8_t *ok);
/* fixed-rate sampling u_i = t0 + (float)i * dt (fp32), replacing main.cpp's clock()-paced loops
 * (:302-316, :341-351).  out / ok may be NULL; out_traj (may be NULL, needs dim == 3 and der == 0 or
 * any der) receives the samples as a device-resident polyline, e.g. as the next spline's middle points. */
int wa_bspline_sample(wa_bspline *b, float t0, float dt, int64_t count, int32_t der, float *out,
                      uint8_t *ok, wa_traj **out_traj);

#ifdef __cplusplus
}
#endif
#endif /* WELDACS_H */
