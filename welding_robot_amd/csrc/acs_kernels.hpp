// acs_kernels.hpp -- device side of ACS_Rank (ACSRank_3D.hpp), hand-written for gfx950 wave64.
//
//   k_init_pheromone  initFromGridMap :343-408 / reset :307-315         HBM write, 24 B/voxel
//   k_heuristic       the (1 + beta*cos) factor of selectNext :151-154  once per problem
//   k_begin           computeSolution prologue :229-233 + first :247-249
//   k_walk<DEV>       one WAVEFRONT per ant: lanes 0-5 own the six neighbours   latency-bound
//   k_walk<REF>       one wavefront per problem walks the ants in order on the libc stream
//   k_rank            best update :263-264, rank :273-275, deposit coefficients, next :247-249
//   k_evaporate       :268-272, float4 stream over 6N floats                     HBM-bound
//   k_deposit_mark/_apply   update_pheromone :198-215, rank-ordered adds without float atomics
//
// fp32 semantics are the reference's: IEEE div/sqrt, denormals kept, no FMA
// contraction (the TU is built with -ffp-contract=off), NaN propagation as written (Q3).
#pragma once
#include "wa_device.h"

struct WaAcsDev {
    WaDims d;
    const float *cx, *cy, *cz;
    const uint8_t *occ;            // free_[id]
    float *pher, *heur;            // [slot][pher_stride]
    unsigned long long *mask;      // [slot][pher_stride]  deposit rank masks, one bit per depositing rank of the pass (<= 64) ...
    uint8_t *mask8;                // ... or, when at most 8 ranks can ever deposit (max_colony <= 35), one BYTE per edge (mask == null)
    uint32_t *bestmark;            // [slot][n]
    int32_t *bestpos;              // [slot][n]  index of a marked voxel on the best path
    uint8_t *besttabu;             // [slot][path_cap] bit k: neighbour k of best[i] lies on the prefix best[0..i]
    int32_t *bestpath;             // [slot][path_cap]
    float *rtab;                   // [slot][path_cap][8] replay table of the best path (see k_replay_table); may be null
    int32_t *paths;                // [slot][max_colony][path_cap]
    float *antL;                   // [slot][max_colony]
    int32_t *antLen;               // [slot][max_colony]
    int32_t *perm;                 // [slot][max_colony]   rank o-1 -> ant
    float *depA;                   // [slot][max_colony]   (lambda-o)*Q/L of rank o
    float *sortk;                  // [slot][2*max_colony] REF introsort scratch (key, tag records)
    uint32_t *vbits;               // [slot][vbits_rows][vbits_words] spill tabu bitmap (all zero at rest)
    WaSlotCtl *ctl;                // [slot]
    WaGlibcRand *rng;              // REF stream (one per solver, like the process-global rand())
    unsigned long long *dbg;       // [16] diagnostic cycle counters (only written by -DWA_STAMPS builds)
    float *trBest, *trIter;        // [slot][trace_cap]
    int32_t *trColony, *trFinite;
    long long *trSteps;
    int64_t pher_stride;           // floats per slot (6N rounded up to 64)
    int64_t path_cap;
    int64_t vbits_words;
    int32_t max_colony;
    int32_t trace_cap;
    int32_t nb;                    // edges per voxel: 6 (face neighbours) or 26 (faces + edges + corners)
    // lazy evaporation: a voxel whose six outgoing edges never received a deposit ("clean", stamp 0) is never swept; its
    // edges are worth ctl.clean (or 0 where the stored value is 0).  A deposited ("dirty") voxel carries
    // stamp = 1 + the evaporation count its stored record is current for; whoever needs the record later applies the
    // missing multiplications by rho one by one (same fp32 roundings as the sweep).  Records are brought current when
    // they receive a deposit, and every `period` generations (16 or 64, see k_evap_rank_mark) by a background pass over 1/period of the
    // dirty list, so about that many multiplications at most are ever pending.  dcount[slot][2] = {list entries the
    // background pass may touch, append cursor}.  All null in the (default) dense mode.
    uint32_t *stamp;               // [slot][n]
    int32_t *dirty_list;           // [slot][n]
    int32_t *dcount;               // [slot][2]
    // stragglers (single-search dense solvers, colony <= 256; all null otherwise): an ant that can no longer be among the depositing ranks
    // nor become the best path leaves its launch at one of the loop's checks (every 64 nodes; every 16 once it has seen shorter arrivals) and is finished by a resume block of the NEXT generation's
    // walk launch, on the previous generation's field (see k_walk_dev)
    uint32_t *arr_len;             // [slot][256] node counts of the running generation's arrivals (26 neighbours: the bits of their L; 0xffffffff = none yet)
    uint32_t *arr_n;               // [slot]
    int32_t *pool_n;               // [slot][2]   stragglers of generation g in pool [g & 1]
    int32_t *pool_rec;             // [slot][2][WA_RESUME_MAX][WA_POOL_REC]  (ant, node count at the hand-over, 26 neighbours: bits of L so far)
    int32_t *pool_path;            // [slot][2][WA_RESUME_MAX][path_cap]  the straggler's path so far (its own slot belongs to the next generation's ant)
    unsigned long long *strag_cnt; // [slot][2]  ants handed over / stragglers finished by a resume block, per slot (wa_acs_straggler_counters)
    const float *prev_pher;        // the field of the previous generation (intact until the next sweep): what a resume block walks on
    float *ltab;                   // [path_cap + 1] L after i steps = precision added i times in fp32 (:78), one table per solver
    int32_t guard_bytes;           // guard band in front of / behind the pheromone and heuristic allocations (6-neighbour solvers)
    int32_t stamp_guard_bytes;     // ... and the stamp allocation of a lazily evaporating solver
    int32_t vbits_rows;            // bitmap rows per slot: max_colony (+ WA_RESUME_MAX rows of the resume blocks when the solver has straggler pools)
};

#define WA_RESUME_MAX 256
#define WA_POOL_REC 4

// the arrival list and the straggler pools of ONE slot (every search of a launch hands its own stragglers over)
struct WaStrag {
    uint32_t *arr_len, *arr_n;
    int32_t *pool_n, *pool_rec, *pool_path;
};
__device__ __forceinline__ WaStrag wa_strag_of(const WaAcsDev &D, int32_t slot)
{
    WaStrag g;
    g.arr_len = D.arr_len + (int64_t)slot * 256;
    g.arr_n = D.arr_n + slot;
    g.pool_n = D.pool_n + (int64_t)slot * 2;
    g.pool_rec = D.pool_rec + (int64_t)slot * 2 * WA_RESUME_MAX * WA_POOL_REC;
    g.pool_path = D.pool_path + (int64_t)slot * 2 * WA_RESUME_MAX * D.path_cap;
    return g;
}

// rank masks of one slot: u64 per edge, or one byte per edge for small colonies (8x less memory: 805 -> 101 MB per slot at 256^3)
struct WaMaskRef {
    unsigned long long *w;
    uint8_t *b;
};
__device__ __forceinline__ WaMaskRef wa_mask_of(const WaAcsDev &D, int32_t slot)
{
    WaMaskRef m;
    m.w = D.mask ? D.mask + (int64_t)slot * D.pher_stride : nullptr;
    m.b = D.mask8 ? D.mask8 + (int64_t)slot * D.pher_stride : nullptr;
    return m;
}
__device__ __forceinline__ void wa_mask_or(const WaMaskRef &m, int64_t e, int bit)
{
    if (m.w) atomicOr(&m.w[e], 1ULL << bit);
    else atomicOr(reinterpret_cast<unsigned int *>(m.b + (e & ~(int64_t)3)), (1u << bit) << (8 * (int)(e & 3)));
}
__device__ __forceinline__ unsigned long long wa_mask_get(const WaMaskRef &m, int64_t e) { return m.w ? m.w[e] : (unsigned long long)m.b[e]; }
__device__ __forceinline__ void wa_mask_clear(const WaMaskRef &m, int64_t e)
{
    if (m.w) m.w[e] = 0;
    else m.b[e] = 0;
}


// path word = voxel id | (edge index taken to arrive << SHIFT)
template <int NB> struct WaNbT;
template <> struct WaNbT<6> { static constexpr int SHIFT = WA_K_SHIFT; static constexpr int32_t IDM = (int32_t)WA_ID_MASK; };
template <> struct WaNbT<26> { static constexpr int SHIFT = 27; static constexpr int32_t IDM = (1 << 27) - 1; };

__device__ __forceinline__ int32_t wa_delta(int k, int32_t nx, int32_t nxy)
{
    // edge order of ACSRank_3D.hpp:355-365: z-1, y-1, x-1, x+1, y+1, z+1
    return k == 0 ? -nxy : k == 1 ? -nx : k == 2 ? -1 : k == 3 ? 1 : k == 4 ? nx : nxy;
}

// offsets (dx, dy, dz) of edge k.  6 neighbours: the push order of ACSRank_3D.hpp:355-365 (z-1, y-1, x-1, x+1, y+1, z+1);
// 26 neighbours: the reference's cube loop (:352-388) -- z offset outermost, then y, then x, centre skipped
__device__ __forceinline__ void wa_off26(int k, int &dx, int &dy, int &dz)
{
    const int q = k < 13 ? k : k + 1;
    dz = q / 9 - 1;
    dy = (q / 3) % 3 - 1;
    dx = q % 3 - 1;
}
template <int NB>
__device__ __forceinline__ void wa_edge_offset(int k, int &dx, int &dy, int &dz)
{
    if (NB == 26) { wa_off26(k, dx, dy, dz); return; }
    dx = k == 2 ? -1 : k == 3 ? 1 : 0;
    dy = k == 1 ? -1 : k == 4 ? 1 : 0;
    dz = k == 0 ? -1 : k == 5 ? 1 : 0;
}

// ------------------------------------------------------------------ pheromone init / reset
// mode 0: initFromGridMap (out-of-bounds edges 0), mode 1: reset() (every edge pheromone_0).
// The sign bit is set on edges whose neighbour is out of bounds or occupied.  Thread per (voxel, edge): coalesced 4-byte stores
// over the [N][NB] field, one definition for both neighbourhoods.
template <int NB>
__global__ __launch_bounds__(256) void k_init_pheromone(WaAcsDev D, int32_t slot0, float p0, int32_t mode)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= D.d.n * NB) return;
    const int32_t slot = slot0 + blockIdx.y;
    const int64_t id = t / NB;
    const int k = (int)(t - id * NB);
    const int32_t x = (int32_t)(id % D.d.nx), y = (int32_t)((id / D.d.nx) % D.d.ny), z = (int32_t)(id / D.d.nxy);
    int dx, dy, dz;
    wa_edge_offset<NB>(k, dx, dy, dz);
    const int32_t X = x + dx, Y = y + dy, Z = z + dz;
    const bool inb = X >= 0 && X < D.d.nx && Y >= 0 && Y < D.d.ny && Z >= 0 && Z < D.d.nz;
    const bool adm = inb && D.occ[id + dz * D.d.nxy + dy * D.d.nx + dx] != 0;
    const float v = (inb || mode == 1) ? p0 : 0.f;
    D.pher[(int64_t)slot * D.pher_stride + t] = adm ? v : -v;
}

// ------------------------------------------------------------------ heuristic field
// (1 + beta*cos) of :151-154 is a function of the voxel, the edge and the END point only: the fields live in a pool,
// wa_acs_begin computes one per distinct end point of its batch that the pool does not hold yet (`fields` / `ends` = pool
// index and end point of each field to compute) and every search reads the field ctl.heur_slot names
template <int NB>
__global__ __launch_bounds__(256) void k_heuristic(WaAcsDev D, float beta, const int32_t *fields, const int32_t *ends)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= D.d.n * NB) return;
    const int32_t slot = fields[blockIdx.y];
    const int64_t id = t / NB;
    const int k = (int)(t - id * NB);
    const int32_t end = ends[blockIdx.y];
    const int32_t x = (int32_t)(id % D.d.nx), y = (int32_t)((id / D.d.nx) % D.d.ny), z = (int32_t)(id / D.d.nxy);
    const int32_t ex = end % D.d.nx, ey = (end / D.d.nx) % D.d.ny, ez = end / D.d.nxy;
    int dx, dy, dz;
    wa_edge_offset<NB>(k, dx, dy, dz);
    const int32_t X = x + dx, Y = y + dy, Z = z + dz;
    float out = 0.f;
    if (X >= 0 && X < D.d.nx && Y >= 0 && Y < D.d.ny && Z >= 0 && Z < D.d.nz) {
        const float ax = D.cx[ex] - D.cx[x], ay = D.cy[ey] - D.cy[y], az = D.cz[ez] - D.cz[z];  // :137
        const float bx = D.cx[X] - D.cx[x], by = D.cy[Y] - D.cy[y], bz = D.cz[Z] - D.cz[z];     // :151
        const float dot = ax * bx + ay * by + az * bz;
        const float na = sqrtf(ax * ax + ay * ay + az * az);
        const float nb = sqrtf(bx * bx + by * by + bz * bz);
        out = 1 + beta * (dot / (na * nb));   // :152-154 (0/0 = NaN on a duplicated seam coordinate, Q3)
    }
    D.heur[(int64_t)slot * D.pher_stride + t] = out;
}

// :247-249 -- colony in double then truncated, lambda double -> float, Q float
__device__ __forceinline__ void wa_next_params(WaSlotCtl &c, const WaRun &R, int which)
{
    int32_t colony;
    if (R.fixed_colony > 0) colony = R.fixed_colony;
    else colony = (int32_t)(0.35 * (double)(c.bestL < R.predict ? c.bestL : R.predict) / (double)R.precision);
    c.colony[which] = colony;
    c.lambda[which] = (float)(0.2 * (double)colony);
    c.Q[which] = R.pheromone_0 / c.lambda[which] * (c.bestL == INFINITY ? R.predict : c.bestL);
}

__global__ void k_begin(WaAcsDev D, WaRun R, int32_t n_problems, const long long *starts,
                        const long long *ends, const uint32_t *streams, const int32_t *heur_slots)
{
    int32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_problems) return;
    WaSlotCtl c = D.ctl[slot];
    c.start = (int32_t)starts[slot];
    c.end = (int32_t)ends[slot];
    c.stream = streams ? streams[slot] : (uint32_t)slot;
    c.heur_slot = heur_slots[slot];
    c.clean[0] = c.clean[c.gen & 1];   // lazy evaporation: the field's clean value carries over; generation parity restarts
    c.evap_base += (uint32_t)c.gen;    // ... and so does the count of evaporations applied so far
    c.gen = 0;
    c.tabu_gen = -2;
    c.bestL = INFINITY;  // :232; the best PATH is kept (Q9) but unreachable while bestL is inf
    c.best_len = 0;
    c.n_dep = 0;
    c.flags = 0;
    wa_next_params(c, R, 0);
    D.ctl[slot] = c;
    if (D.pool_n) {
        const WaStrag sg = wa_strag_of(D, slot);
        sg.pool_n[0] = sg.pool_n[1] = 0;
        *sg.arr_n = 0;
        for (int i = 0; i < 256; i++) sg.arr_len[i] = 0xffffffffu;
    }
}


// stored value -> value after `lag` more evaporations (:270, one rounding per multiplication like the sweep)
__device__ __forceinline__ float wa_catch_up(float v, uint32_t lag, float rho)
{
    for (uint32_t i = 0; i < lag; i++) v *= rho;
    return v;
}

// ------------------------------------------------------------------ the walk
// One wavefront = one ant.  Lanes 0..5 own the six neighbours (edge order of :355-365); the
// wave is alone on its SIMD most of the time, so the inner loop is written for instruction
// count, not occupancy: no divergent branches on the fast path, the two ORDERED float sums of
// selectNext (forward `total` :155, reverse `prob_sum` :177) are 5-step DPP row scans, the
// roulette pick is one compare + ballot + find-last-bit, and the tabu probe's terminating
// empty slot doubles as the insertion slot of the chosen neighbour.
//
// tabu set = open-addressing hash of voxel ids in LDS (the reference's std::set, :70,:145);
// when a walk outgrows 3/4 of the table the wave spills to its private global bitmap and
// continues in the generic (slow) loop.
struct WaTabu {
    int32_t *tab;
    uint32_t mask, shift;
    uint32_t *bits;
    bool spilled;
};
__device__ __forceinline__ bool tabu_has(const WaTabu &t, int32_t key)
{
    if (t.spilled) {
        uint32_t w = __hip_atomic_load(&t.bits[(uint32_t)key >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (w >> (key & 31)) & 1u;
    }
    uint32_t h = ((uint32_t)key * 2654435761u) >> t.shift;
    for (;;) {
        int32_t v = t.tab[h];
        if (v == key) return true;
        if (v == WA_HASH_EMPTY) return false;
        h = (h + 1) & t.mask;
    }
}
__device__ __forceinline__ void tabu_insert(const WaTabu &t, int32_t key)
{
    if (t.spilled) {
        uint32_t old = atomicOr(&t.bits[(uint32_t)key >> 5], 1u << (key & 31));
        asm volatile("" ::"v"(old));  // returning atomic: completed before the next lookup
        return;
    }
    uint32_t h = ((uint32_t)key * 2654435761u) >> t.shift;
    while (t.tab[h] != WA_HASH_EMPTY) h = (h + 1) & t.mask;
    t.tab[h] = key;
}

// lane `lane_uniform` of v := val_uniform (both wave-uniform).  The s_nop covers the wait states the assembler cannot see through the
// inline statement (an SGPR written by a VALU instruction -- v_readlane -- read as data / lane select by the next VALU instruction)
__device__ __forceinline__ int32_t wa_writelane(int32_t v, int32_t val_uniform, int32_t lane_uniform)
{
    asm volatile("s_mov_b32 m0, %2\n s_nop 3\n v_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(val_uniform), "s"(lane_uniform) : "m0");   // (one SGPR + m0: the constant bus takes no two SGPRs)
    return v;
}

// lane i <- lane i-1 (row_shr:1) / lane i <- lane i+1 (row_shl:1); lanes shifted in read 0
__device__ __forceinline__ float dpp_from_below(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_above(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x101, 0xf, 0xf, true));
}

// whole-wave versions (wave_shr:1 / wave_shl:1, GFX9 DPP): lane i <- lane i-1 / lane i+1 across all 64 lanes
__device__ __forceinline__ float dpp_wave_from_below(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_wave_from_above(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true));
}

// In-kernel stamps (diagnostic builds only, -DWA_STAMPS): s_memtime at section boundaries of the
// walk's inner loop, differences summed per section; ant 0 of slot 0 writes the sums to D.dbg.
// Never enabled in the product build (cdna_hip_programming.md 7, "In-kernel stamps").
#ifdef WA_STAMPS
#define WA_STAMP(i)                                                                               \
    do {                                                                                          \
        unsigned long long t_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        stamp_acc[i] += t_ - stamp_prev;                                                          \
        stamp_prev = t_;                                                                          \
    } while (0)
#else
#define WA_STAMP(i) do { } while (0)
#endif

// The two ORDERED fp32 sums of selectNext over the (zero-padded) candidate values `a` of one 8-lane
// group: t -> role 5 holds total = ((((0+a0)+a1)+...)+a5) (:155); c -> role i holds prob_sum after adding
// candidates 5..i (:172-177).  One definition for the walk step and for the replay table, so the bits agree.
__device__ __forceinline__ void wa_ordered_sums(float a, float &t, float &c)
{
    t = 0.f + a;
#pragma unroll
    for (int i = 0; i < 5; i++) t = dpp_from_below(t) + a;
    c = 0.f + a;
#pragma unroll
    for (int i = 0; i < 5; i++) c = dpp_from_above(c) + a;
}

struct WaWalkState {
    int32_t cur, len;
    uint32_t step;
    float L;
    bool done;
    int reason;   // why an unfinished walk came back from the fast loop: 0 = its limits (table load, capacity), 4 = rejoin watch
    int32_t pbuf; // ... and the words of its incomplete 64-word block (lane i = word i of the block), for the re-entry
    bool pbuf_valid;
};

// fast path: hash tabu only.  Returns with st.done set, or with st.done clear when the table
// reached its spill threshold (the caller continues in wa_walk_slow).
//
// Path words are not stored one per step: a global store per step would put the store's
// round trip on the critical path (CDNA4 counts stores in vmcnt and the data VGPR cannot be
// reused before the store retires).  Instead lane (len & 63) captures the word in a VGPR and
// the wave flushes 64 consecutive path entries with ONE coalesced 256-byte store.
template <int MODE, bool ALPHA1, bool SPARSE>
__device__ __forceinline__ void wa_walk_fast(const WaRun &R, const float *__restrict__ pher,
                                             const float *__restrict__ heur, const uint32_t *__restrict__ stamp, float clean_info,
                                             uint32_t evap_now,
                                             int32_t *__restrict__ path,
                                             int32_t *tab, int hash_log2, int32_t nx, int32_t nxy, int32_t n_vox,
                                             int32_t path_cap, int32_t end, uint64_t antkey, int32_t &rng_rs, int32_t &rng_f,
                                             int32_t &rng_b, int32_t spill_at, WaWalkState &st, int32_t *flags_out,
                                             unsigned long long *dbg, const int32_t *prefix_words)
{
    // Lane layout: group j = lane >> 3 (j < 6), role k2 = lane & 7 (k2 < 6).  Every step, group j
    // PREFETCHES the pheromone/heuristic record of neighbour j of the current voxel (36 lanes x 2
    // dwords); the group of the neighbour that gets picked then simply becomes the active group
    // of the next step, so the record is already in the right lanes and the HBM / Infinity-Cache
    // latency overlaps with this step's decision instead of following it.
    const int lane = threadIdx.x;
    const int j = lane >> 3, k2 = lane & 7;
    const bool lane_ok = j < 6 && k2 < 6;
    const int32_t dk = wa_delta(k2, nx, nxy);   // edge this lane evaluates when its group is active
    const int32_t dj = wa_delta(j, nx, nxy);    // neighbour of `cur` this lane's group prefetches
    const int32_t last_id = n_vox - 1;
    const int32_t limit = path_cap < spill_at + 1 ? path_cap : spill_at + 1;  // leave the loop when len reaches it
    const uint32_t hmask = (1u << hash_log2) - 1u, hshift = 32 - hash_log2;
    const char *pher_b = reinterpret_cast<const char *>(pher);
    const char *heur_b = reinterpret_cast<const char *>(heur);
    // per-lane constants so that the per-step address math is one scalar multiply + one VALU add:
    //   byte offset of (neighbour j of cur, edge k2) = cur*24 + (dj*24 + k2*4), clamped into the field
    //   hash of (cur + dk)                           = (cur*K + dk*K) >> shift      (mod 2^32)
    const int32_t pf_const = dj * 24 + k2 * 4;
    const int32_t kc = k2 < 6 ? k2 : 5;       // idle lanes (roles 6,7 / groups 6,7) load too, harmlessly in range
    const int32_t pf_lo = kc * 4, pf_hi = last_id * 24 + kc * 4;
    const uint32_t hk_const = (uint32_t)dk * 2654435761u;
    int32_t cur = st.cur, len = st.len;      // the step about to be taken is step number len - 1
    float L = st.L;
    // lane (i & 63) holds path word i of the current 64-entry block; when the walk resumes after a
    // replayed prefix the already-written part of that block comes from the prefix
    int32_t pbuf = st.cur;
    if (prefix_words) pbuf = lane < (st.len & 63) ? prefix_words[(st.len & ~63) + lane] : 0;
    int grp = 0;             // group holding the record of `cur`
    float ublock = 0.f;      // DEV: lane i holds the uniform draw of the step with (len & 63) == i
    float pp = -0.f, ph = 0.f;
    uint32_t pd = 1;         // SPARSE: stamp of the voxel whose record pp/ph belong to (0 = clean => edges are worth clean_info)
    // software pipeline: the record of `cur` (pp/ph) and the tabu probe of its neighbours (tv/hs)
    // are issued one step early, right after `cur` became known, and consumed at the loop top
    int32_t nb = cur + dk;
    uint32_t hs = ((uint32_t)cur * 2654435761u + hk_const) >> hshift;
    int32_t tv = WA_HASH_EMPTY;
    if (lane_ok && j == 0) {
        const uint32_t boff = ((uint32_t)cur * 6u + (uint32_t)k2) * 4u;
        pp = *reinterpret_cast<const float *>(pher_b + boff);
        ph = *reinterpret_cast<const float *>(heur_b + boff);
    }
    if (SPARSE) pd = stamp[cur];
    tv = tab[hs];            // every lane probes (unmasked): only the active group's result is used
    if (MODE == 1) ublock = (float)wa_ctr_draw(antkey, (uint32_t)((len & ~63) + lane - 1)) / 2147483648.0f;
    bool dead = false;
#ifdef WA_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    for (;;) {
        WA_STAMP(0);                             // loop back-edge + wait for the prefetched record
        const float p = pp, h = ph;              // record of `cur`, valid in group `grp`
        const uint32_t pstamp = pd;
        // lane predicates are kept as 64-bit SCALAR masks (one v_cmp each, combined with s_and): a ballot of a
        // compound lane condition would round-trip through a VGPR (v_cndmask + v_cmp) every time it is tested
        const unsigned long long actm = 0x3fULL << (grp * 8);   // roles 0..5 of the active group
        {                                        // prefetch the six neighbours' records; every lane loads
            int32_t cur24 = cur * 24;            // (no exec masking): addresses are clamped into the field,
            asm volatile("" : "+s"(cur24));      // an out-of-bounds neighbour is never walked to
            int32_t boff = cur24 + pf_const;     // < 2 GiB (checked at create)
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(boff) : "v"(boff), "v"(pf_lo), "v"(pf_hi));
            pp = *reinterpret_cast<const float *>(pher_b + (uint32_t)boff);
            ph = *reinterpret_cast<const float *>(heur_b + (uint32_t)boff);
            if (SPARSE) {                        // the neighbour's stamp travels with its record
                int32_t vj = cur + dj;
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(vj) : "v"(vj), "v"(0), "v"(last_id));
                pd = stamp[vj];
            }
        }
        WA_STAMP(1);                             // prefetch issue
        // ---- tabu probe results of the active lanes; collisions (rare) walk the chain here
        unsigned long long un = actm & __ballot(tv != nb) & __ballot(tv != WA_HASH_EMPTY);
        while (__builtin_expect(un != 0, 0)) {
            if ((un >> lane) & 1ULL) {
                hs = (hs + 1) & hmask;
                tv = tab[hs];
            }
            un = actm & __ballot(tv != nb) & __ballot(tv != WA_HASH_EMPTY);
        }
        WA_STAMP(2);                             // probe wait + collision check
        // in bounds and free (sign bit clear), not visited (:145-148)
        const unsigned long long admm = actm & __ballot((int32_t)__float_as_uint(p) >= 0) & __ballot(tv != nb);
        float mag = fabsf(p);
        uint32_t stv = 1;
        if (SPARSE) {   // the six lanes of the active group hold the same voxel's stamp: uniform, so scalar control flow
            stv = (uint32_t)__builtin_amdgcn_readlane((int)pstamp, grp * 8);
            if (stv != 0) mag = wa_catch_up(mag, evap_now + 1u - stv, R.rho);     // pending evaporations of a deposited voxel
        }
        float pa = ALPHA1 ? mag : wa_powi(mag, R.alpha);
        if (SPARSE && stv == 0) pa = clean_info;                                  // never-deposited voxel: every admissible edge holds the clean value
        const float info = pa * h;                                                // :154
        float a;  // adm ? info : 0 -- x + 0.0f == x: padding keeps both sums exact
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(a) : "v"(info), "s"(admm));
        float t, c;  // total -> role 5 of the active group; prob_sum after candidate i -> role i
        wa_ordered_sums(a, t, c);
        const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), grp * 8 + 5));
        WA_STAMP(3);                                   // admissibility + both ordered scans
        float rnd;                                     // (float)rand() / (float)RAND_MAX, RAND_MAX -> 2^31 (:169)
        if (MODE == 1) {                               // DEV draws are pure functions of (ant, step): 64 at a
            if (__builtin_expect((len & 63) == 0, 0))  // time, one per lane, lane i = the step with len & 63 == i
                ublock = (float)wa_ctr_draw(antkey, (uint32_t)(len + lane - 1)) / 2147483648.0f;
            rnd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ublock), len & 63));
        } else {
            // no candidate (:162-166) returns before rand() is called: only draw when one exists
            if (admm == 0) { dead = true; break; }
            rnd = (float)wa_glibc_next_lanes(rng_rs, rng_f, rng_b) / 2147483648.0f;  // lockstep private copies
        }
        rnd *= total;                                  // :170
        const unsigned long long m2 = admm & __ballot(c >= rnd);  // :178
        if (__builtin_expect(m2 == 0, 0)) { dead = true; break; }  // no candidate (:162-166) or fall-through (:191-192)
        const int pick_lane = 63 - __clzll((long long)m2);        // first hit when scanning i = 5..0
        const int pick = pick_lane - grp * 8;
        WA_STAMP(4);                                   // draw, compare, ballot, pick
        if (lane == pick_lane) tab[hs] = nb;           // addNextNode :75 -- the probe ended on the free slot
        // ---- issue the next step's probe (after the insert: LDS is in order) for the new active group
        grp = pick;
        cur += __builtin_amdgcn_readlane(dk, pick);    // lane k (< 6) holds delta_k
        nb = cur + dk;
        uint32_t curK = (uint32_t)cur * 2654435761u;
        asm volatile("" : "+s"(curK));                 // scalar multiply; the per-lane part is hk_const
        hs = (curK + hk_const) >> hshift;
        tv = tab[hs];
        WA_STAMP(5);                                   // insert + next probe issue
        // ---- bookkeeping
        {   // lane (len & 63) of pbuf <- path word (:76-77); one v_writelane instead of mov+cmp+cndmask.
            // s_nop covers the "VALU-written SGPR as lane select" hazard the compiler cannot see in asm.
            const int32_t word = cur | (pick << WA_K_SHIFT), sel = len & 63;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(pbuf) : "s"(word), "s"(sel) : "m0");
        }
        if (__builtin_expect((len & 63) == 63, 0))     // block full: one coalesced store
            path[(len & ~63) + lane] = pbuf;
        len++;
        L += R.precision;                              // :78, distance == precision (:378)
        WA_STAMP(6);                                   // path capture, counters
        // arrived (:182-186), or the table is 3/4 full / the path buffer is full: one test, sorted out below
        if (__builtin_expect((cur == end) | (len >= limit), 0)) break;
    }
    const int exit_code = dead ? 1 : (cur == end ? 2 : 3);  // 1 dead end, 2 arrived, 3 limit (spill / capacity)
#ifdef WA_STAMPS
    if (dbg && lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&dbg[i], stamp_acc[i]);
        atomicAdd(&dbg[8], (unsigned long long)(len - 1));
    }
#endif
    if (exit_code == 1) L = INFINITY;
    st.done = exit_code != 3;
    if (!st.done && len >= path_cap) {                 // the next step would not fit path[]
        if (lane == 0) atomicOr(flags_out, WA_FLAG_PATH_OVERFLOW);
        L = INFINITY;
        st.done = true;
    }
    if (len & 63) {  // partial last block (entries [len & ~63, len))
        if (lane < (len & 63)) path[(len & ~63) + lane] = pbuf;
    }
    st.cur = cur; st.len = len; st.step = (uint32_t)(len - 1); st.L = L;
}

#ifdef WA_STRAG_TIME
__device__ unsigned long long wa_strag_t[128 * 8];
__device__ __forceinline__ uint32_t wa_strag_arr(uint32_t *arr_n, int32_t cut_n, int32_t gen) {
    const uint32_t i = atomicAdd(arr_n, 1u);
    if ((int32_t)i == cut_n - 1 && gen < 128) wa_strag_t[gen * 8 + 1] = wall_clock64();
    return i;
}
#define WA_ARR_IDX wa_strag_arr(sg.arr_n, cut_n, gen)
#else
#define WA_ARR_IDX atomicAdd(sg.arr_n, 1u)
#endif
#include "walk_loop_gfx950.hpp"   // wa_walk_fast_asm<LAZY>: the hand-scheduled general step

// generic path: handles the spilled (global bitmap) tabu; same arithmetic, written plainly
template <int MODE, bool SPARSE>
__device__ __forceinline__ void wa_walk_slow(const WaAcsDev &D, const WaRun &R, const float *pher, const float *heur,
                                          const uint32_t *stamp, float clean_info, uint32_t evap_now,
                                          int32_t *path, WaTabu T, int32_t end, uint64_t antkey, int32_t &rng_rs,
                                          int32_t &rng_f, int32_t &rng_b, int32_t spill_at, WaWalkState &st,
                                          int32_t *flags_out)
{
    const int lane = threadIdx.x;
    const int32_t nx = D.d.nx, nxy = D.d.nxy;
    int32_t cur = st.cur, len = st.len;
    uint32_t step = st.step;
    float L = st.L;
    const int k = lane;
    const int32_t dk = wa_delta(k, nx, nxy);
    for (;;) {
        if (!T.spilled && len > spill_at) {  // hash nearly full: move the set to the bitmap
            __threadfence();
            for (int i = lane; i < len; i += 64) {
                int32_t id = __hip_atomic_load(&path[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & WA_ID_MASK;
                uint32_t old = atomicOr(&T.bits[(uint32_t)id >> 5], 1u << (id & 31));
                asm volatile("" ::"v"(old));
            }
            __threadfence();
            T.spilled = true;
            if (lane == 0) atomicOr(flags_out, WA_FLAG_BITMAP_USED);
        }
        float p = -0.f, h = 0.f;
        bool adm = false;
        if (k < 6) {
            p = pher[(int64_t)cur * 6 + k];
            h = heur[(int64_t)cur * 6 + k];
            if ((__float_as_uint(p) >> 31) == 0) adm = !tabu_has(T, cur + dk);
        }
        float mag = fabsf(p);
        uint32_t stv = 1;
        if (SPARSE) {
            stv = stamp[cur];
            if (stv != 0) mag = wa_catch_up(mag, evap_now + 1u - stv, R.rho);
        }
        float pa = wa_powi(mag, R.alpha);
        if (SPARSE && stv == 0) pa = clean_info;
        float info = pa * h;
        uint32_t m = (uint32_t)__ballot(adm) & 0x3fu;
        if (m == 0) { L = INFINITY; break; }
        float v[6];
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(info), i));
            if ((m >> i) & 1u) total += v[i];
        }
        int32_t r;
        if (MODE == 1) r = (int32_t)wa_ctr_draw(antkey, step);
        else r = wa_glibc_next_lanes(rng_rs, rng_f, rng_b);
        float rnd = (float)r / 2147483648.0f;
        rnd *= total;
        float prob = 0.f;
        int pick = -1;
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            if (pick < 0 && ((m >> i) & 1u)) {
                prob += v[i];
                if (prob >= rnd) pick = i;
            }
        }
        if (pick < 0) { L = INFINITY; break; }
        int32_t next = cur + wa_delta(pick, nx, nxy);
        if (len >= D.path_cap) {
            if (lane == 0) atomicOr(flags_out, WA_FLAG_PATH_OVERFLOW);
            L = INFINITY;
            break;
        }
        if (lane == 0) {
            path[len] = next | (pick << WA_K_SHIFT);
            tabu_insert(T, next);
        }
        __builtin_amdgcn_wave_barrier();
        len++;
        L += R.precision;
        step++;
        if (next == end) break;
        cur = next;
    }
    if (T.spilled) {  // leave the bitmap all-zero for the next walk
        __threadfence();
        for (int i = lane; i < len; i += 64) {
            int32_t id = __hip_atomic_load(&path[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & WA_ID_MASK;
            __hip_atomic_store(&T.bits[(uint32_t)id >> 5], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
    }
    st.cur = cur; st.len = len; st.step = step; st.L = L;
    st.done = true;
}

// ------------------------------------------------------------------ replay of the best path
// While an ant has followed the global-best path from the start, its visited set is exactly the
// path prefix, so admissibility, info, `total` and the cumulative thresholds at node i are the
// same for every such ant: k_replay_table computes them once per generation, and the ant only has
// to check that its own draw picks the path's edge:  rnd = u * total;  first i (from 5 down) with
// thr[i] >= rnd  ==  next_k ?   Bit-identical to taking the full step (same operands, same order,
// same draw), about a fifth of the instructions.  At the first node where the draw picks another
// edge the ant rebuilds its tabu hash from the prefix and continues in the general loop, which
// recomputes that step in full.  After convergence nearly every step of every ant is a replay step.
// Returns 1 dead end at node i, 2 arrived, 3 deviates at node i (i in `node`).
__device__ __forceinline__ int wa_walk_replay(const float *__restrict__ T, int32_t rlen, uint64_t antkey, int32_t &node)
{
    // Replay steps do not depend on each other while the ant stays on the path, so 64 consecutive nodes
    // are checked at once, ONE LANE PER NODE: the lane reads its node's 32-byte row (two coalesced 16-B
    // loads straight from the table, the next 64 rows already in flight), forms its own draw (a pure
    // function of the step number = node index), rnd = u * total, and finds the edge the roulette would
    // take: scanning i = 5..0 the first thr[i] >= rnd is the highest set bit of the 6 comparisons.
    // The first lane whose edge is not the path's edge is the first node where the ant leaves the path
    // (some edge taken) or dies (none).
    const int lane = threadIdx.x;
    const float4 *__restrict__ T4 = reinterpret_cast<const float4 *>(T);
    const int32_t last = rlen - 1;                      // decisions exist at nodes 0 .. rlen-2
    int32_t nv = lane < last ? lane : last - 1;         // (rlen >= 2; masked lanes re-read a valid row)
    float4 a = T4[2 * nv], b = T4[2 * nv + 1];
    for (int32_t i0 = 0;; i0 += 64) {
        const int32_t nodev = i0 + lane;
        const bool valid = nodev < last;
        const float4 ca = a, cb = b;
        if (i0 + 64 < last) {                           // rows of the next 64 nodes
            nv = nodev + 64 < last ? nodev + 64 : last - 1;
            a = T4[2 * nv];
            b = T4[2 * nv + 1];
        }
        float rnd = (float)wa_ctr_draw(antkey, (uint32_t)nodev) / 2147483648.0f;  // (float)rand()/(float)RAND_MAX (:169)
        rnd *= cb.z;                                                               // :170, total
        const int nk = __float_as_int(cb.w);
        // thr = admissible ? prob_sum : -inf   (:178)
        const uint32_t h = (ca.x >= rnd ? 1u : 0u) | (ca.y >= rnd ? 2u : 0u) | (ca.z >= rnd ? 4u : 0u) | (ca.w >= rnd ? 8u : 0u) |
                           (cb.x >= rnd ? 16u : 0u) | (cb.y >= rnd ? 32u : 0u);
        const int pick = h ? 31 - __clz((int)h) : -1;
        const unsigned long long fm = __ballot(valid && pick != nk);
        if (__builtin_expect(fm != 0, 0)) {
            const int g = __ffsll((long long)fm) - 1;
            node = i0 + g;
            return __builtin_amdgcn_readlane((int)h, g) ? 3 : 1;
        }
        if (i0 + 64 >= last) { node = last; return 2; }  // every decision up to the last node followed the path
    }
}

// ------------------------------------------------------------------ back onto the replay track after a detour
// An ant that left the best path and came back to it stands on best[q] with its own visited set V (its tabu hash).  Row j
// of the replay table was built for the visited set best[0..j]; it says what THIS ant would do at best[j] iff the two sets
// agree on the six neighbours of best[j]:  a neighbour the row treats as admissible must not be in V (a detour node next
// to the path), and a neighbour the row treats as visited-because-on-the-prefix must be in V or be one of best[q..j-1],
// which the ant visits on the way (a path node the detour skipped is not).  64 rows are checked at once, one lane per
// row: six LDS probes of V, the position of a skipped-looking neighbour from bestpos[], then the usual draw-against-
// thresholds test.  Returns 1 dead end at best[stop], 2 arrived (stop = last node), 3 the ant has to take a general step
// at best[stop] (its draw leaves the path there, or the row does not apply to it); rows q .. stop-1 were followed.
__device__ __forceinline__ int wa_replay_from(const float *__restrict__ T, const int32_t *__restrict__ bpath, const uint8_t *__restrict__ btabu,
                                              const int32_t *__restrict__ pos, int32_t blen, int32_t q, uint32_t step_q, uint64_t antkey,
                                              const WaTabu &V, int32_t nx, int32_t nxy, int32_t max_rows, int32_t &stop)
{
    const int lane = threadIdx.x;
    const float4 *__restrict__ T4 = reinterpret_cast<const float4 *>(T);
    const int32_t last = blen - 1;                      // decisions exist at nodes 0 .. last-1
    const int32_t lim = q + max_rows < last ? q + max_rows : last;
    for (int32_t j0 = q;; j0 += 64) {
        const int32_t j = j0 + lane;
        const bool live = j < lim;
        const int32_t jj = live ? j : (q < last ? q : last - 1);
        const float4 a = T4[2 * jj], b = T4[2 * jj + 1];
        const int32_t v = bpath[jj] & (int32_t)WA_ID_MASK;
        const uint32_t bt = btabu[jj];
        const float thr[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
        bool applies = true;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const bool adm = thr[k] != -INFINITY;       // (a NaN threshold -- seam, Q3 -- is an admissible edge)
            const bool pre = (bt >> k) & 1u;
            if (live && (adm || pre)) {                 // either way the neighbour is in bounds
                const int32_t nb = v + wa_delta(k, nx, nxy);
                const bool inV = tabu_has(V, nb);
                if (adm) applies = applies && !inV;
                else if (!inV) { const int32_t ps = pos[nb]; applies = applies && ps >= q && ps <= j; }
            }
        }
        float rnd = (float)wa_ctr_draw(antkey, step_q + (uint32_t)(j - q)) / 2147483648.0f;   // (float)rand()/(float)RAND_MAX (:169)
        rnd *= b.z;                                                                             // :170, total
        const int nk = __float_as_int(b.w);
        const uint32_t h = (a.x >= rnd ? 1u : 0u) | (a.y >= rnd ? 2u : 0u) | (a.z >= rnd ? 4u : 0u) | (a.w >= rnd ? 8u : 0u) |
                           (b.x >= rnd ? 16u : 0u) | (b.y >= rnd ? 32u : 0u);
        const int pick = h ? 31 - __clz((int)h) : -1;
        const unsigned long long fm = __ballot(live && (!applies || pick != nk));
        if (fm != 0) {
            const int g = __ffsll((long long)fm) - 1;
            stop = j0 + g;
            const bool ok_row = (__ballot(applies) >> g) & 1ULL;
            return (ok_row && __builtin_amdgcn_readlane((int)h, g) == 0) ? 1 : 3;
        }
        if (j0 + 64 >= lim) { stop = lim; return lim == last ? 2 : 3; }
    }
}

#ifdef WA_ANT_TIME
#define WA_PHASE(i) do { if (slot == 0 && ant == 0 && threadIdx.x == 0 && D.dbg) atomicAdd(&D.dbg[i], (unsigned long long)__builtin_readcyclecounter()); } while (0)
#else
#define WA_PHASE(i) do { } while (0)
#endif
// every slot of the tabu hash := empty.  Eight 1-KB wave stores per trip (immediate offsets, no address arithmetic between them):
// the 128 KB table of a lone search takes ~0.5 us instead of the 4.6 us of a store-per-trip loop (measured, tools/ant_time.py)
__device__ __forceinline__ void wa_tabu_clear(int4 *tab4, int hash_log2)
{
    const int n16 = (1 << hash_log2) / 4, lane = threadIdx.x;
    const int4 e = make_int4(-1, -1, -1, -1);
    int i = lane;
    for (; i + 7 * 64 < n16; i += 8 * 64) {
#pragma unroll
        for (int u = 0; u < 8; u++) tab4[i + u * 64] = e;
    }
    for (; i < n16; i += 64) tab4[i] = e;
    // the entry behind the table is a sentinel: the hand-scheduled loop reads every probed slot together with its successor, and the
    // successor of the LAST slot is this one -- neither empty nor any key, so that lane takes the slow path to slot 0
    if (lane == 0) reinterpret_cast<int32_t *>(tab4)[1 << hash_log2] = WA_HASH_SENTINEL;
}

template <int MODE, bool ALPHA1, bool SPARSE, bool WARM = true, bool REJ = true>
__device__ __forceinline__ void wa_walk_one(const WaAcsDev &D, const WaRun &R, int32_t slot, int32_t ant,
                                            int32_t start, int32_t end, uint64_t antkey, int32_t *tab,
                                            int hash_log2, int32_t &rng_rs, int32_t &rng_f, int32_t &rng_b,
                                            int32_t *flags_out, int32_t rlen, float bestL, float clean, uint32_t evap_now, int32_t walk_flags,
                                            uint32_t best_ver, int32_t heur_slot, int32_t cut_n = 0x7fffffff, const int32_t *res_words = nullptr,
                                            int32_t res_len = 0, int32_t gen = 0, int32_t bits_row = -1)
{
    // cut_n: straggler check (0x7fffffff = off).  res_words / res_len: this block RESUMES a straggler of the previous generation -- the
    // walk continues behind its res_len nodes (D.pher is then that generation's field, rlen 0, no rejoin watch) and only its statistics
    // are delivered (the ant's slot in agents[] belongs to the running generation's ant by now)
    const int lane = threadIdx.x;
    const WaStrag sg = wa_strag_of(D, slot);   // (only dereferenced where D.pool_n is set: cut_n / res_words say so)
    const float *pher = D.pher + (int64_t)slot * D.pher_stride;
    const float *heur = D.heur + (int64_t)heur_slot * D.pher_stride;   // (the caller read it with the rest of the control block)
    const uint32_t *stamp = SPARSE ? D.stamp + (int64_t)slot * D.d.n : nullptr;
    const float clean_info = SPARSE ? wa_powi(clean, R.alpha) : 0.f;   // power() of the clean value, once per walk
    int32_t *path = res_words ? const_cast<int32_t *>(res_words) : D.paths + ((int64_t)slot * D.max_colony + ant) * D.path_cap;
    const int32_t *bpath = D.bestpath + (int64_t)slot * D.path_cap;
    WaWalkState st;
    st.cur = start; st.len = 1; st.step = 0; st.L = 0.f; st.done = false; st.pbuf = 0; st.pbuf_valid = false;
    const int32_t *prefix_words = nullptr;
    if (res_words) {
        st.len = res_len;
        st.cur = __builtin_amdgcn_readfirstlane(res_words[res_len - 1] & (int32_t)WA_ID_MASK);
        st.step = (uint32_t)(res_len - 1);
        for (int32_t q = 0; q < res_len - 1; q++) st.L += R.precision;   // :78, one add per step taken
        prefix_words = res_words;
    } else if (MODE == 1 && rlen > 1) {
        int32_t node = 0;
        // the first 512 words of the best path are requested BEFORE the replay decides how many of them the ant walks: their round trip
        // runs beside the table rows' (once converged every ant copies all of them)
        int32_t w0[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int32_t q = u * 64 + lane;
            w0[u] = q < rlen ? bpath[q] : 0;
        }
        const int what = wa_walk_replay(D.rtab + (int64_t)slot * D.path_cap * 8, rlen, antkey, node);
#ifdef WA_STAMPS
        if (lane == 0 && D.dbg) {   // diagnostic: how far do ants follow the best path?  [10] += nodes replayed, [11] += ants,
            atomicAdd(&D.dbg[10], (unsigned long long)node);          // [12] += ants that arrived on the replay track
            atomicAdd(&D.dbg[11], 1ULL);
            if (what == 2) atomicAdd(&D.dbg[12], 1ULL);
        }
#endif
        st.len = node + 1;
        // the walked prefix IS the best path's.  512 words per round: eight independent loads per lane, then eight stores
        // (a load-store pair per round would put one memory round trip per 64 words on every converged walk)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int32_t q = u * 64 + lane;
            if (q < st.len) path[q] = w0[u];
        }
        for (int32_t q0 = 512; q0 < st.len; q0 += 512) {
            int32_t w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int32_t q = q0 + u * 64 + lane;
                w[u] = q < st.len ? bpath[q] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int32_t q = q0 + u * 64 + lane;
                if (q < st.len) path[q] = w[u];
            }
        }
        if (what != 3) {  // finished on the replay track
            // arriving over the whole best path accumulates exactly the steps that produced best.L
            const float L = what == 2 ? bestL : INFINITY;
            if (lane == 0) {
                D.antL[(int64_t)slot * D.max_colony + ant] = L;
                D.antLen[(int64_t)slot * D.max_colony + ant] = st.len;
                if (what == 2 && cut_n != 0x7fffffff) __hip_atomic_store(&sg.arr_len[WA_ARR_IDX & 255u], (uint32_t)st.len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an arrival, for the straggler check (write-through: the checking ants sit on other XCDs)
            }
            return;
        }
        st.cur = bpath[node] & (int32_t)WA_ID_MASK;
        st.step = (uint32_t)node;                                // steps taken so far = draws consumed
        for (int32_t q = 0; q < node; q++) st.L += R.precision;  // :78, one add per step taken
        prefix_words = bpath;
    }
    WaTabu T;
    T.tab = tab;
    T.mask = (1u << hash_log2) - 1u;
    T.shift = 32 - hash_log2;
    // (a resume block spills into a bitmap row of its own, behind the ants' rows: the ant's row belongs to the running generation's ant)
    T.bits = D.vbits + ((int64_t)slot * D.vbits_rows + (bits_row >= 0 ? bits_row : ant)) * D.vbits_words;
    T.spilled = false;
    const int32_t spill_at = (int32_t)((3u << hash_log2) >> 2);

    WA_PHASE(6);
    int4 *tab4 = reinterpret_cast<int4 *>(tab);
    wa_tabu_clear(tab4, hash_log2);
    __builtin_amdgcn_wave_barrier();
    WA_PHASE(7);
    if (prefix_words && st.len <= spill_at) {  // (a longer prefix goes straight to the spilled slow loop)
        // tabu set := the replayed prefix.  Distinct keys, no deletions: any insertion order gives a valid
        // open-addressing table, so the lanes insert concurrently with compare-and-swap on the slot.
        for (int32_t q = lane; q < st.len; q += 64) {
            const int32_t key = prefix_words[q] & (int32_t)WA_ID_MASK;
            uint32_t h = ((uint32_t)key * 2654435761u) >> T.shift;
            while (atomicCAS(&tab[h], WA_HASH_EMPTY, key) != WA_HASH_EMPTY) h = (h + 1) & T.mask;
        }
    } else if (!prefix_words && lane == 0) {
        tabu_insert(T, start);  // addStartNode :81-86 (path[0] is buffered by the fast loop)
    }
    __builtin_amdgcn_wave_barrier();
    const int32_t fast_limit = (int32_t)D.path_cap < spill_at + 1 ? (int32_t)D.path_cap : spill_at + 1;
    bool use_asm = false;
#ifndef WA_STAMPS
    use_asm = ALPHA1 && (walk_flags & 1) && (MODE == 1 || !SPARSE);   // (REF mode: the same loop, draws from the libc stream)
#endif
    // ---- a straggler (the loop left through its check, st.reason == 5): its path so far goes to a pool entry of its generation; agents[]
    // says "not arrived, st.len nodes" (what the ranking sees); a resume block of the next walk launch finishes it and adds the rest to
    // the generation's statistics.  False when the pool is full: the ant walks on without the check.
    auto hand_over = [&]() -> bool {
        int32_t r = 0;
        if (lane == 0) r = atomicAdd(&sg.pool_n[gen & 1], 1);
        r = __builtin_amdgcn_readfirstlane(r);
        if (r >= WA_RESUME_MAX) {
            if (lane == 0) atomicSub(&sg.pool_n[gen & 1], 1);
            return false;
        }
        int32_t *pp = sg.pool_path + ((int64_t)(gen & 1) * WA_RESUME_MAX + r) * D.path_cap;
        // (through L2: the last, incomplete block was stored by this very wavefront a moment ago)
        // 512 words per round: eight independent loads per lane, then eight stores (one memory round trip per round, not per 64 words)
        for (int32_t q0 = 0; q0 < st.len; q0 += 512) {
            int32_t w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int32_t q = q0 + u * 64 + lane;
                w[u] = q < st.len ? __hip_atomic_load(&path[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int32_t q = q0 + u * 64 + lane;
                if (q < st.len) pp[q] = w[u];
            }
        }
        if (lane == 0) {
            sg.pool_rec[((gen & 1) * WA_RESUME_MAX + r) * WA_POOL_REC] = ant;
            sg.pool_rec[((gen & 1) * WA_RESUME_MAX + r) * WA_POOL_REC + 1] = st.len;
            D.antL[(int64_t)slot * D.max_colony + ant] = INFINITY;
            D.antLen[(int64_t)slot * D.max_colony + ant] = st.len;
#ifndef WA_ANT_TIME
            if (D.dbg) atomicAdd(&D.dbg[9], 1ULL);   // ants handed over since the counters were last reset (wa_acs_debug_counters)
            atomicAdd(&D.strag_cnt[slot * 2], 1ULL);
#endif
        }
        return true;
    };
    if (REJ && st.len < fast_limit && use_asm && prefix_words && (walk_flags & 2)) {
        // The ant replayed a prefix of the best path and left it.  Measured (profiles/HISTORY.md): such an ant is back on the path after a
        // median of 3-4 steps and 82-92 % of its remaining nodes lie on it, so the general loop runs with a rejoin watch and every
        // time the ant is found on the path again it goes back onto the replay track for as long as the table applies to it.
        const uint32_t *mark = D.bestmark + (int64_t)slot * D.d.n;
        const int32_t *bpos = D.bestpos + (int64_t)slot * D.d.n;
        const uint8_t *btabu = D.besttabu + (int64_t)slot * D.path_cap;
        const float *RT = D.rtab + (int64_t)slot * D.path_cap * 8;
        // forced hand-backs are test knobs (tests/test_gpu_reentry.py): compiled only into the -DWA_TEST_KNOBS build of the library
#ifdef WA_TEST_KNOBS
        const bool knob_never = walk_flags & 4, knob_anywhere = walk_flags & 8;
        int32_t hold = ((walk_flags >> 8) & 0xffff) ? ((walk_flags >> 8) & 0xffff) : 1, backoff = 1;
#else
        constexpr bool knob_never = false, knob_anywhere = false;
        int32_t hold = 1, backoff = 1;
#endif
#ifdef WA_ANT_TIME
        unsigned long long dbg_hand = 0, dbg_gain = 0, dbg_t_hand = 0;
        const int32_t dbg_prefix = st.len;
#endif
        for (;;) {
            wa_walk_fast_asm<SPARSE ? 3 : 2, WARM>(R, pher, heur, stamp, clean_info, evap_now, path, tab, hash_log2, D.d.nx, D.d.nxy, (int32_t)D.path_cap, end, antkey, spill_at,
                                             D.guard_bytes, D.stamp_guard_bytes, D.ltab, st, flags_out, prefix_words, nullptr, mark, knob_anywhere ? 0u : best_ver, hold,
                                             SPARSE ? nullptr : sg.arr_len, cut_n);
            prefix_words = path;                                  // from now on the ant's own words (its partial block is in memory)
#ifdef WA_ANT_TIME
            if (dbg_t_hand) { dbg_t_hand = 0; }
#endif
            if (st.done || st.reason != 4) break;   // (5: a straggler, handed over below)
#ifdef WA_ANT_TIME
            dbg_hand++;
            const unsigned long long dbg_t0 = __builtin_readcyclecounter();
#endif
            int32_t gained = 0;
            const uint32_t mk = mark[st.cur];
            const int32_t ps = bpos[st.cur];                      // (fetched beside the stamp, meaningful only under it)
            const int32_t q = mk == best_ver ? ps : -1;
            if (q >= 0 && q < rlen - 1 && !knob_never) {
                int32_t room = spill_at - st.len;                 // nodes the tabu hash / the path may still take
                if ((int32_t)D.path_cap - st.len < room) room = (int32_t)D.path_cap - st.len;
                int32_t stop = q;
                const int kind = room > 0 ? wa_replay_from(RT, bpath, btabu, bpos, rlen, q, (uint32_t)(st.len - 1), antkey, T, D.d.nx, D.d.nxy, room, stop) : 3;
                gained = stop - q;
                for (int32_t t = lane; t < gained; t += 64) {     // the ant walked best[q+1 .. stop]: path words (:76-77) and tabu set (:75)
                    const int32_t w = bpath[q + 1 + t];
                    path[st.len + t] = w;
                    const int32_t key = w & (int32_t)WA_ID_MASK;
                    uint32_t hh = ((uint32_t)key * 2654435761u) >> T.shift;
                    while (atomicCAS(&tab[hh], WA_HASH_EMPTY, key) != WA_HASH_EMPTY) hh = (hh + 1) & T.mask;
                }
                __builtin_amdgcn_wave_barrier();
                if (gained > 0) {
                    // the words of the ant's incomplete 64-word block stay in a register across the re-entry (lane i = word i of the
                    // block): what was there before the commit, then the committed words -- all of it when the commit crossed a boundary
                    const int32_t old_len = st.len, new_len = st.len + gained;
                    const int32_t wi = (new_len & ~63) + lane;
                    int32_t pb = 0;
                    if (lane < (new_len & 63)) pb = wi >= old_len ? bpath[q + 1 + (wi - old_len)] : st.pbuf;
                    st.pbuf = pb;
                }
                st.len += gained;
                st.step = (uint32_t)(st.len - 1);
                st.cur = bpath[stop] & (int32_t)WA_ID_MASK;
                if (kind == 2) { st.L = D.ltab[st.len - 1]; st.done = true; break; }   // arrived over the rest of the best path (:78)
                if (kind == 1) { st.L = INFINITY; st.done = true; break; }      // no candidate at best[stop] (:162-166, :191-192)
            }
            if (gained > 0) { backoff = 1; hold = 1; }
            else { hold = backoff; backoff = backoff < 32 ? backoff * 2 : 32; }   // the table does not apply here: walk on before asking again
#ifdef WA_ANT_TIME
            dbg_gain += (unsigned long long)gained;
            dbg_t_hand = 1;
            if (lane == 0 && D.dbg) atomicAdd(&D.dbg[13], (unsigned long long)__builtin_readcyclecounter() - dbg_t0);   // ticks between leaving the loop and re-entering it (re-entry prologue not included)
#endif
            if (st.len >= fast_limit) break;
        }
#ifdef WA_ANT_TIME
        if (lane == 0 && D.dbg) {   // [12] the ant with the most hand-backs: (hand-backs, nodes gained on the replay track, general steps, replayed prefix); [14] += hand-backs, [15] += ants in this loop
            atomicMax(&D.dbg[12], (dbg_hand << 48) | (dbg_gain << 32) | ((unsigned long long)(st.len - dbg_prefix - (int32_t)dbg_gain) << 16) | (unsigned long long)dbg_prefix);
            atomicAdd(&D.dbg[14], dbg_hand);
            atomicAdd(&D.dbg[15], 1ULL);
        }
#endif
        if (!st.done && st.reason == 5 && hand_over()) return;
        if (!st.done) st.L = D.ltab[st.len - 1];                  // the generic loop goes on adding to it (also behind a full pool)
    } else if (st.len < fast_limit && use_asm && MODE == 0) {
        // REF mode on the hand-scheduled loop: draws from the shared libc stream, 64 at a time (wa_walk_fast_asm<..., REFDRAW>); whatever
        // it leaves undone -- a dead end to be decided, a walk past the table's load limit -- the generic loop below finishes
        wa_walk_fast_asm<0, WARM, true>(R, pher, heur, stamp, clean_info, evap_now, path, tab, hash_log2, D.d.nx, D.d.nxy, (int32_t)D.path_cap, end, antkey, spill_at,
                                        D.guard_bytes, D.stamp_guard_bytes, D.ltab, st, flags_out, prefix_words, nullptr, nullptr, 0, 0, nullptr, 0x7fffffff,
                                        &rng_rs, &rng_f, &rng_b);
    } else if (st.len < fast_limit && use_asm) {
        WA_PHASE(8);
        wa_walk_fast_asm<SPARSE ? 1 : 0, WARM>(R, pher, heur, stamp, clean_info, evap_now, path, tab, hash_log2, D.d.nx, D.d.nxy, (int32_t)D.path_cap, end, antkey, spill_at,
                                 D.guard_bytes, D.stamp_guard_bytes, D.ltab, st, flags_out, prefix_words, (slot == 0 && ant == 0) ? D.dbg : nullptr,
                                 nullptr, 0, 0, SPARSE ? nullptr : sg.arr_len, cut_n);
        if (!st.done && st.reason == 5) {
            if (hand_over()) return;
            st.L = D.ltab[st.len - 1];              // the pool is full: the generic loop finishes this ant
        }
    }
    else if (st.len < fast_limit)
        wa_walk_fast<MODE, ALPHA1, SPARSE>(R, pher, heur, stamp, clean_info, evap_now, path, tab, hash_log2, D.d.nx, D.d.nxy, (int32_t)D.d.n, (int32_t)D.path_cap, end, antkey,
                                   rng_rs, rng_f, rng_b, spill_at, st, flags_out, (slot == 0 && ant == 0) ? D.dbg : nullptr, prefix_words);
    else if (st.len >= (int32_t)D.path_cap) {  // cannot happen after a replay (the best path fits), kept for symmetry
        if (lane == 0) atomicOr(flags_out, WA_FLAG_PATH_OVERFLOW);
        st.L = INFINITY;
        st.done = true;
    } else if (!prefix_words && lane == 0) {
        path[0] = start;  // the slow loop reads the path back from memory
    }
    WA_PHASE(9);
    if (!st.done) wa_walk_slow<MODE, SPARSE>(D, R, pher, heur, stamp, clean_info, evap_now, path, T, end, antkey, rng_rs, rng_f, rng_b, spill_at, st, flags_out);
    if (res_words) {   // a resumed straggler: the rest of its walk belongs to generation `gen`'s statistics
#ifndef WA_ANT_TIME
        if (lane == 0 && D.dbg) atomicAdd(&D.dbg[7], 1ULL);   // ... and stragglers finished by a resume block
        if (lane == 0) atomicAdd(&D.strag_cnt[slot * 2 + 1], 1ULL);
#endif
        if (lane == 0 && gen < D.trace_cap) {
            const int64_t t = (int64_t)slot * D.trace_cap + gen;
            if (st.L != INFINITY) atomicAdd(&D.trFinite[t], 1);
            atomicAdd(reinterpret_cast<unsigned long long *>(&D.trSteps[t]), (unsigned long long)(st.len - res_len));
        }
        if (walk_flags & 64) {   // drain launch (no newer generation's ant owns the slot): the finished walk goes back to agents[]
            int32_t *own = D.paths + ((int64_t)slot * D.max_colony + ant) * D.path_cap;
            __threadfence();     // (the last words were stored by this wavefront; read them back through L2)
            for (int32_t q = lane; q < st.len; q += 64) own[q] = __hip_atomic_load(&path[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) {
                D.antL[(int64_t)slot * D.max_colony + ant] = st.L;
                D.antLen[(int64_t)slot * D.max_colony + ant] = st.len;
            }
        }
        return;
    }
    if (lane == 0) {
        D.antL[(int64_t)slot * D.max_colony + ant] = st.L;
        D.antLen[(int64_t)slot * D.max_colony + ant] = st.len;
        if (st.L != INFINITY && cut_n != 0x7fffffff) __hip_atomic_store(&sg.arr_len[WA_ARR_IDX & 255u], (uint32_t)st.len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an arrival, for the straggler check (write-through: the checking ants sit on other XCDs)
    }
}

// ------------------------------------------------------------------ replay table of the best path
// One 16-lane row per best-path node i (roles 0..5 = the six edges): the walk's own step evaluation with
// visited set = {best[0..i]}, i.e. neighbour nb is tabu iff it is marked and bestpos[nb] <= i (bits
// precomputed by wa_best_prefix_tabu whenever the best path changes).
// Output per node: thr[k] = admissible ? prob_sum_k : -inf (k = 0..5), total, edge taken to best[i+1].
template <int NB>
__device__ __forceinline__ void wa_apply_body(const WaAcsDev &D, int32_t slot, int32_t base, int32_t bit, int32_t bx, int32_t nbx,
                                              bool skip_best_src, float *s_dep);

// apply_here: the row also APPLIES the pending ranked deposits (mask != 0) of its six edges -- same adds, same
// ascending rank order as wa_apply_body -- writes them back, clears the masks, and evaluates on the new values.
__device__ __forceinline__ void wa_table_rows(const WaAcsDev &D, const WaRun &R, int32_t slot, int32_t row0, int32_t rows, bool apply_here,
                                              const float *s_dep, int32_t w_first, int32_t w_first_next)
{
    // w_first / w_first_next = bestpath[row0], bestpath[row0 + 1], loaded by the caller before the best length was
    // known (speculatively, inside the allocation) so that the row's record loads start one round trip earlier
    const WaSlotCtl *ctl = &D.ctl[slot];
    if (ctl->bestL == INFINITY) return;
    const int32_t blen = ctl->best_len;
    const uint32_t ver = ctl->best_ver;
    const float lambda = ctl->dep_lambda, Q = ctl->dep_Q, bestL = ctl->dep_bestL;
    const int32_t k2 = threadIdx.x & 15;
    const int32_t kk = k2 < 6 ? k2 : 5;
    const int32_t *bpath = D.bestpath + (int64_t)slot * D.path_cap;
    uint8_t *btabu = D.besttabu + (int64_t)slot * D.path_cap;
    const uint32_t *mark = D.bestmark + (int64_t)slot * D.d.n;
    const int32_t *pos = D.bestpos + (int64_t)slot * D.d.n;
    // the best path changed in the generation just ranked: its prefix-tabu bits (which neighbours of best[i] lie on
    // best[0..i]) are rebuilt here, one row per node and all rows at once, instead of by the single block that ranks --
    // that block's dependent gathers used to outlast the whole evaporation sweep in exploratory generations
    const bool rebuild = ctl->tabu_gen + 1 == ctl->gen;
    float *pher = D.pher + (int64_t)slot * D.pher_stride;
    const WaMaskRef mask = wa_mask_of(D, slot);
    const float *heur = D.heur + (int64_t)D.ctl[slot].heur_slot * D.pher_stride;
    float *T = D.rtab + (int64_t)slot * D.path_cap * 8;
    const int32_t dk = wa_delta(kk, D.d.nx, D.d.nxy);
    const int32_t last_id = (int32_t)D.d.n - 1;
    // lazy evaporation: a best-path node that never received a deposit (possible when no rank deposits at all)
    // holds the clean value of the field as it stands now, i.e. after this generation's evaporation
    // (and a deposited one that received nothing this generation may have evaporations pending: read-side catch-up)
    const uint32_t *stamp = D.stamp ? D.stamp + (int64_t)slot * D.d.n : nullptr;
    const float clean_now = ctl->clean[ctl->gen & 1];
    const uint32_t evap_tab = ctl->evap_base + (uint32_t)ctl->gen;   // the fused launch already counted this generation
    for (int32_t i = row0; i < blen; i += rows) {
        const int32_t wv = i == row0 ? w_first : bpath[i];
        const int32_t wn = i + 1 < blen ? (i == row0 ? w_first_next : bpath[i + 1]) : 0;
        const int32_t v = wv & (int32_t)WA_ID_MASK;
        // all record loads of the row are independent of each other
        const int64_t e = (int64_t)v * 6 + kk;
        float p = pher[e];
        const float h = heur[e];
        if (stamp) {
            const uint32_t stv = stamp[v];
            p = stv == 0 ? copysignf(clean_now, p) : copysignf(wa_catch_up(fabsf(p), evap_tab + 1u - stv, R.rho), p);
        }
        unsigned long long m = apply_here ? wa_mask_get(mask, e) : 0ULL;
        int32_t nbid = v + dk;
        nbid = nbid < 0 ? 0 : nbid > last_id ? last_id : nbid;       // (an out-of-bounds edge is inadmissible by its sign bit whatever is found here)
        const uint32_t mk = (apply_here || rebuild) ? mark[nbid] : 0u;
        uint32_t bt;
        if (rebuild) {   // neighbour k2 is tabu for an ant standing on best[i] that came along the path iff it lies on best[0..i]
            // (only for a neighbour id inside the field: wa_replay_from looks such a neighbour up by its id)
            const bool on = k2 < 6 && nbid == v + dk && mk == ver && pos[nbid] <= i;
            bt = (uint32_t)(__ballot(on) >> (threadIdx.x & 48)) & 0x3fu;   // the six lanes of this 16-lane row
            if (k2 == 0) btabu[i] = (uint8_t)bt;
        } else {
            bt = btabu[i];
        }
        bool adm = false;
        if (k2 < 6) {
            if (m) {  // somebody walked (v, k2): apply the ranked deposits in ascending rank order (:210-211)
                const bool onbest = mk == ver;  // v itself is on the best path (:209)
                const float bonus = (float)onbest * lambda * Q / bestL;
                while (m) {
                    int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    p += s_dep[b] + bonus;
                }
                pher[e] = p;
                wa_mask_clear(mask, e);
            }
            // in bounds and free (:148), and not on the prefix best[0..i] (:145-146)
            adm = (__float_as_uint(p) >> 31) == 0 && !((bt >> k2) & 1u);
        } else {
            p = -0.f;
        }
        const float info = (R.alpha == 1 ? fabsf(p) : wa_powi(fabsf(p), R.alpha)) * (k2 < 6 ? h : 0.f);  // :154
        const float a = adm ? info : 0.f;
        float t, c;
        wa_ordered_sums(a, t, c);
        if (k2 < 6) T[(int64_t)i * 8 + k2] = adm ? c : -INFINITY;
        if (k2 == 5) T[(int64_t)i * 8 + 6] = t;
        if (k2 == 0) T[(int64_t)i * 8 + 7] = __int_as_float(i + 1 < blen ? (int32_t)((uint32_t)wn >> WA_K_SHIFT) : -1);
    }
}

__global__ __launch_bounds__(256) void k_replay_table(WaAcsDev D, WaRun R)
{
    const int32_t row0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int32_t *bpath = D.bestpath + (int64_t)blockIdx.y * D.path_cap;
    const int32_t w0 = row0 < D.path_cap ? bpath[row0] : 0, w1 = row0 + 1 < D.path_cap ? bpath[row0 + 1] : 0;
    wa_table_rows(D, R, blockIdx.y, row0, (gridDim.x * blockDim.x) >> 4, false, nullptr, w0, w1);
}

// Deposit apply + replay table in ONE launch (DEV fast path, <= 64 depositing ranks): blocks [0, TB) are
// table rows that also apply the deposits on every edge leaving a best-path node -- the only values
// the table depends on -- and blocks [TB, TB + 8*64) are the ordinary apply pass, which skips exactly
// those edges.  The two roles touch disjoint edges, so no ordering between them is needed.
// table blocks: the host passes 64 for one or a few searches (1024 rows: the 800-1 300-node best paths of the exploratory generations get a
// row each; measured on the driver's command: 32 blocks 5 358 gen/s, 64 5 432, 96 5 351; no difference once converged) and 32 for launches
// that carry 32 searches or more (C5 with 224: 0.535 s against 0.540)
#define WA_TABLE_BLOCKS_MAX 64
// split_log2: apply blocks per depositing rank = 1 << this (the host passes 2 for one or a few searches -- 8 blocks per rank are no
// faster --, 1 for launches that carry 32 searches or more)
__global__ __launch_bounds__(256) void k_apply_table(WaAcsDev D, WaRun R, int32_t split_log2, int32_t table_blocks)
{
    __shared__ float s_dep[64];
    const int32_t slot = blockIdx.y;
    // lazy evaporation: voxels that became dirty in this generation join the swept set from the next sweep on
    if (D.dcount && blockIdx.x == 0 && threadIdx.x == 0) D.dcount[slot * 2] = D.dcount[slot * 2 + 1];
    if (D.pool_n && blockIdx.x == 0) {   // stragglers: the next generation starts with no arrivals and an empty pool of its own
        const WaStrag sg = wa_strag_of(D, slot);
        sg.arr_len[threadIdx.x] = 0xffffffffu;
        if (threadIdx.x == 0) { *sg.arr_n = 0; sg.pool_n[D.ctl[slot].gen & 1] = 0; }   // (ctl.gen is already the next generation's number)
    }
    if ((int32_t)blockIdx.x < table_blocks) {
        // independent loads first: deposit coefficients, control block, this row's path words
        const int32_t tid = threadIdx.x, row0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
        const float dep_mine = (tid < 64 && tid < D.max_colony) ? D.depA[(int64_t)slot * D.max_colony + tid] : 0.f;
        const int32_t *bpath = D.bestpath + (int64_t)slot * D.path_cap;
        const int32_t w0 = row0 < D.path_cap ? bpath[row0] : 0, w1 = row0 + 1 < D.path_cap ? bpath[row0 + 1] : 0;
        const int32_t n_dep = D.ctl[slot].n_dep;
        if (tid < 64) s_dep[tid] = tid < n_dep ? dep_mine : 0.f;
        __syncthreads();
        wa_table_rows(D, R, slot, row0, (table_blocks * blockDim.x) >> 4, true, s_dep, w0, w1);
        return;
    }
    const int32_t ab = (int32_t)blockIdx.x - table_blocks;  // 0..(ranks << split_log2)-1: (bx, rank bit)
    wa_apply_body<6>(D, slot, 0, ab >> split_log2, ab & ((1 << split_log2) - 1), 1 << split_log2, true, s_dep);
}

// DEV: grid = (max_colony, n_problems), block = one wavefront
// WARM: the hand-scheduled loop touches the records two hops ahead of the ant (pays while a search has the GPU to itself, costs
// when many searches saturate it: see walk_loop_gfx950.hpp)
// REJ: the kernel carries the rejoin watch + re-entry onto the replay track.  The host launches the instantiation without it for
// the first generations of a search, in which the watch cannot be armed yet (it waits for a best path that has been stable for
// WA_REENTRY_STABLE generations): the mere presence of that code costs the exploratory walk 1.5 % (187 vs 190 us per launch).
template <bool ALPHA1, bool SPARSE, bool WARM = true, bool REJ = true>
__global__ __launch_bounds__(64) void k_walk_dev(WaAcsDev D, WaRun R, int hash_log2, int32_t gen, int32_t walk_flags)
{
    extern __shared__ int32_t lds[];
    const int32_t slot = blockIdx.y, ant = blockIdx.x;
    const WaSlotCtl *c = &D.ctl[slot];
    const int32_t colony = c->colony[gen & 1];
#ifdef WA_STRAG_TIME
    if (threadIdx.x == 0 && gen < 128) atomicMax(&wa_strag_t[gen * 8 + 0], ~(unsigned long long)wall_clock64());
#endif
    if (!SPARSE && ALPHA1 && D.pool_n && (int32_t)blockIdx.x >= D.max_colony) {
        // ---- resume block: a straggler of generation gen - 1 (walk_flags bit 5 allowed it to leave that launch) finishes its walk here,
        // on that generation's field, beside this generation's ants; only that generation's statistics hear of it
        const int32_t r = (int32_t)blockIdx.x - D.max_colony, pg = (gen - 1) & 1;
        const WaStrag sg = wa_strag_of(D, slot);
        if (gen < 1 || r >= sg.pool_n[pg] || r >= WA_RESUME_MAX) return;
        const int32_t a = sg.pool_rec[(pg * WA_RESUME_MAX + r) * WA_POOL_REC], n0 = sg.pool_rec[(pg * WA_RESUME_MAX + r) * WA_POOL_REC + 1];
        WaAcsDev Dp = D;
        Dp.pher = const_cast<float *>(D.prev_pher);
        const uint64_t key = wa_ctr_antkey(wa_ctr_key(R.seed, c->stream, (uint32_t)(gen - 1)), (uint32_t)a);
        int32_t f0 = 0, b0 = 0, rs0 = 0;
        wa_walk_one<1, true, false, WARM, false>(Dp, R, slot, a, c->start, c->end, key, lds, hash_log2, rs0, f0, b0, &D.ctl[slot].flags, 0, INFINITY, 0.f, 0u,
                                                 walk_flags & (1 | 64), 0u, c->heur_slot, 0x7fffffff, sg.pool_path + ((int64_t)pg * WA_RESUME_MAX + r) * D.path_cap, n0, gen - 1,
                                                 D.max_colony + r);
#ifdef WA_STRAG_TIME
        if (threadIdx.x == 0 && gen < 128) atomicMax(&wa_strag_t[gen * 8 + 3], (unsigned long long)wall_clock64());
#endif
        return;
    }
    if (walk_flags & 64) return;   // a drain launch (wa_acs_sync and friends behind a call whose last generation handed over) only resumes
    if (ant >= colony || colony > D.max_colony) return;  // overflow is flagged by the rank step
    const uint64_t antkey = wa_ctr_antkey(wa_ctr_key(R.seed, c->stream, (uint32_t)gen), (uint32_t)ant);
    int32_t f = 0, b = 0, rs_unused = 0;
    const float bestL = c->bestL;
    const int32_t rlen = (D.rtab && bestL != INFINITY) ? c->best_len : 0;
    // the rejoin watch pays once the colony has settled on the best path (it costs a failed attempt every few steps while the
    // ants still explore): it is switched on when that path has not changed for a number of generations
    if (gen - c->tabu_gen < ((walk_flags >> 24) & 127)) walk_flags &= ~2;
#ifdef WA_ANT_TIME   // diagnostic build (tools/ant_time.py): shader-clock ticks of every ant's block against its step count
    const unsigned long long t0_ = __builtin_readcyclecounter();
    if (slot == 0 && ant == 0 && threadIdx.x == 0 && D.dbg) atomicAdd(&D.dbg[5], t0_);
#endif
    // the straggler check (walk_flags bit 5; never in the last generation of a wa_acs_run call): an ant longer than floor(lambda - 1) + 1
    // arrivals cannot be among the depositing ranks (:200) nor be the iteration's best
    int32_t cut_n = 0x7fffffff;
    if (!SPARSE && ALPHA1 && (walk_flags & 32) && D.pool_n) cut_n = (int32_t)(c->lambda[gen & 1] - 1.f) + 1;
    if (cut_n < 1) cut_n = 1;
    wa_walk_one<1, ALPHA1, SPARSE, WARM, REJ>(D, R, slot, ant, c->start, c->end, antkey, lds, hash_log2, rs_unused, f, b, &D.ctl[slot].flags, rlen, bestL,
                                   c->clean[gen & 1], c->evap_base + (uint32_t)gen, walk_flags, c->best_ver, c->heur_slot, cut_n, nullptr, 0, gen);
#ifdef WA_STRAG_TIME
    if (threadIdx.x == 0 && gen < 128) {
        const bool arrived = D.antL[(int64_t)slot * D.max_colony + ant] != INFINITY;
        atomicMax(&wa_strag_t[gen * 8 + (arrived ? 2 : 4)], (unsigned long long)wall_clock64());
        atomicMax(&wa_strag_t[gen * 8 + (arrived ? 6 : 7)], ((unsigned long long)wall_clock64() << 16) | (unsigned long long)(D.antLen[(int64_t)slot * D.max_colony + ant] & 0xffff));
        if (arrived) atomicAdd(&wa_strag_t[gen * 8 + 5], 1ULL);
    }
#endif
#ifdef WA_ANT_TIME
    if (threadIdx.x == 0 && D.dbg) {
        if (slot == 0 && ant == 0) atomicAdd(&D.dbg[10], (unsigned long long)__builtin_readcyclecounter());
        const unsigned long long t = __builtin_readcyclecounter() - t0_;
        const unsigned long long n = (unsigned long long)(D.antLen[(int64_t)slot * D.max_colony + ant] - 1);
        atomicMax(&D.dbg[1], (t << 24) | n);        // the slowest ant: (ticks, steps)
        atomicMax(&D.dbg[4], (n << 32) | t);        // the ant with the most steps: (steps, ticks)
        atomicAdd(&D.dbg[2], t);
        atomicAdd(&D.dbg[3], n);
    }
#endif
}

// REF: grid = (1, 1): the ants of the single in-flight problem walk one after another and draw
// from the shared glibc stream in exactly the reference's order (:252-261)
// walk_flags bit 0 (and alpha == 1): the hand-scheduled loop with draws from the libc stream, 64 at a time (walk_loop_gfx950.hpp, REFDRAW)
__global__ __launch_bounds__(64) void k_walk_ref(WaAcsDev D, WaRun R, int hash_log2, int32_t gen, int32_t walk_flags)
{
    extern __shared__ int32_t lds[];
    const int32_t slot = blockIdx.y;
    const WaSlotCtl *c = &D.ctl[slot];
    int32_t colony = c->colony[gen & 1];
    if (colony > D.max_colony) return;
    // the 31-word libc state lives in ONE register, word j in lane j; the two indices are wave-uniform (a per-lane copy of the
    // array indexed by them compiles to a 31-way select chain per access: ~90 instructions per draw)
    int32_t r = threadIdx.x < 31 ? D.rng->r[threadIdx.x] : 0;   // lane j holds word j of the state (see wa_glibc_next_lanes)
    int32_t f = D.rng->f, b = D.rng->b;
    const int32_t start = c->start, end = c->end, heur_slot = c->heur_slot;
    if (R.alpha == 1 && (walk_flags & 1)) {
        for (int32_t ant = 0; ant < colony; ant++)
            wa_walk_one<0, true, false, true, false>(D, R, slot, ant, start, end, 0, lds, hash_log2, r, f, b, &D.ctl[slot].flags, 0, INFINITY, 0.f, 0u, 1, 0u, heur_slot);
    } else {
        for (int32_t ant = 0; ant < colony; ant++)
            wa_walk_one<0, false, false>(D, R, slot, ant, start, end, 0, lds, hash_log2, r, f, b, &D.ctl[slot].flags, 0, INFINITY, 0.f, 0u, 0, 0u, heur_slot);
    }
    if (threadIdx.x < 31) D.rng->r[threadIdx.x] = r;
    if (threadIdx.x == 0) {
        D.rng->f = f;
        D.rng->b = b;
    }
}

// ------------------------------------------------------------------ libstdc++ std::sort order
// (bits/stl_algo.h introsort + bits/stl_heap.h, GCC 11) restated for one thread on (key, tag)
// records; reproduces the permutation the reference gets from std::sort at :273.
struct WaRec { float k; int32_t t; };
__device__ inline void ss_push_heap(WaRec *first, long hole, long top, WaRec value)
{
    long parent = (hole - 1) / 2;
    while (hole > top && first[parent].k < value.k) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
__device__ inline void ss_adjust_heap(WaRec *first, long hole, long len, WaRec value)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (first[child].k < first[child - 1].k) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    ss_push_heap(first, hole, top, value);
}
__device__ inline void ss_heap_sort(WaRec *first, WaRec *last)
{
    long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            WaRec v = first[parent];
            ss_adjust_heap(first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        WaRec v = *last;
        *last = *first;
        ss_adjust_heap(first, 0, last - first, v);
    }
}
__device__ inline void ss_swap(WaRec *a, WaRec *b) { WaRec t = *a; *a = *b; *b = t; }
__device__ inline void ss_unguarded_linear_insert(WaRec *last)
{
    WaRec v = *last;
    WaRec *next = last - 1;
    while (v.k < next->k) { *last = *next; last = next; --next; }
    *last = v;
}
__device__ inline void ss_insertion_sort(WaRec *first, WaRec *last)
{
    if (first == last) return;
    for (WaRec *i = first + 1; i != last; ++i) {
        if (i->k < first->k) {
            WaRec v = *i;
            for (WaRec *j = i; j != first; --j) *j = *(j - 1);
            *first = v;
        } else ss_unguarded_linear_insert(i);
    }
}
__device__ inline void wa_std_sort(WaRec *v, int32_t n)
{
    if (n <= 0) return;
    long lg = 0;
    for (unsigned long m = (unsigned long)n; m > 1; m >>= 1) lg++;
    // __introsort_loop with its tail recursion turned into an explicit stack
    struct Frame { WaRec *first, *last; long depth; };
    Frame stack[72];
    int sp = 0;
    stack[sp++] = {v, v + n, 2 * lg};
    while (sp > 0) {
        Frame fr = stack[--sp];
        WaRec *first = fr.first, *last = fr.last;
        long depth = fr.depth;
        while (last - first > 16) {
            if (depth == 0) { ss_heap_sort(first, last); break; }
            --depth;
            WaRec *mid = first + (last - first) / 2;
            WaRec *a = first + 1, *b = mid, *c = last - 1;  // __move_median_to_first
            if (a->k < b->k) {
                if (b->k < c->k) ss_swap(first, b);
                else if (a->k < c->k) ss_swap(first, c);
                else ss_swap(first, a);
            } else if (a->k < c->k) ss_swap(first, a);
            else if (b->k < c->k) ss_swap(first, c);
            else ss_swap(first, b);
            WaRec *lo = first + 1, *hi = last;  // __unguarded_partition, pivot = *first
            for (;;) {
                while (lo->k < first->k) ++lo;
                --hi;
                while (first->k < hi->k) --hi;
                if (!(lo < hi)) break;
                ss_swap(lo, hi);
                ++lo;
            }
            // the reference recurses on [cut,last) FIRST, then loops on [first,cut).  The two
            // ranges are disjoint, so the order of processing does not change the result.
            stack[sp++] = {lo, last, depth};
            last = lo;
        }
    }
    if (n > 16) {
        ss_insertion_sort(v, v + 16);
        for (WaRec *i = v + 16; i != v + n; ++i) ss_unguarded_linear_insert(i);
    } else ss_insertion_sort(v, v + n);
}

// ------------------------------------------------------------------ rank
// one workgroup per problem: iteration best -> global best (strict <, first ant wins :263-264),
// ranking (:273-275), per-rank deposit coefficient, trace, next generation's parameters.
// The (L, ant) sort keys are staged in LDS (up to WA_RANK_LDS ants) so the counting rank reads
// broadcast LDS words instead of a dependent chain of global loads.
#define WA_RANK_LDS 2048
template <int NB>
__global__ __launch_bounds__(256) void k_rank(WaAcsDev D, WaRun R, int32_t gen)
{
    const int32_t slot = blockIdx.x, tid = threadIdx.x;
    WaSlotCtl *ctl = &D.ctl[slot];
    const int32_t colony = ctl->colony[gen & 1];
    const float *antL = D.antL + (int64_t)slot * D.max_colony;
    const int32_t *antLen = D.antLen + (int64_t)slot * D.max_colony;
    int32_t *perm = D.perm + (int64_t)slot * D.max_colony;
    float *depA = D.depA + (int64_t)slot * D.max_colony;
    __shared__ unsigned long long s_keys[WA_RANK_LDS];
    __shared__ unsigned long long s_min;
    __shared__ int32_t s_fin, s_ndep;
    __shared__ unsigned long long s_steps;
    if (tid == 0) { s_min = ~0ULL; s_fin = 0; s_ndep = 0; s_steps = 0; }
    __syncthreads();
    if (colony > D.max_colony) {
        if (tid == 0) { atomicOr(&ctl->flags, WA_FLAG_COLONY_OVERFLOW); ctl->gen = gen + 1; }
        return;
    }
    const bool in_lds = colony <= WA_RANK_LDS;
    // L >= 0 or +inf, so the uint32 order of the bit pattern is the float order
    unsigned long long mykey = ~0ULL;
    int32_t myfin = 0;
    unsigned long long mysteps = 0;
    for (int32_t a = tid; a < colony; a += blockDim.x) {
        float La = antL[a];
        unsigned long long key = ((unsigned long long)__float_as_uint(La) << 32) | (uint32_t)a;
        if (in_lds) s_keys[a] = key;
        mykey = key < mykey ? key : mykey;
        myfin += (La != INFINITY) ? 1 : 0;
        mysteps += (unsigned long long)(antLen[a] - 1);
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long ok = __shfl_down(mykey, o, 64);
        mykey = ok < mykey ? ok : mykey;
        myfin += __shfl_down(myfin, o, 64);
        mysteps += __shfl_down(mysteps, o, 64);
    }
    if ((tid & 63) == 0) {
        atomicMin(&s_min, mykey);
        atomicAdd(&s_fin, myfin);
        atomicAdd(&s_steps, mysteps);
    }
    __syncthreads();
    float iterL = INFINITY;
    int32_t iterAnt = -1;
    if (colony > 0) { iterL = __uint_as_float((uint32_t)(s_min >> 32)); iterAnt = (int32_t)(s_min & 0xffffffffu); }
    float bestL = ctl->bestL;
    uint32_t ver = ctl->best_ver;
    const float lambda = ctl->lambda[gen & 1], Q = ctl->Q[gen & 1];
    __syncthreads();
    if (iterAnt >= 0 && iterL < bestL) {  // best = agentK (:264): copy the path, re-stamp membership
        const int32_t blen = antLen[iterAnt];
        const int32_t *src = D.paths + ((int64_t)slot * D.max_colony + iterAnt) * D.path_cap;
        int32_t *dst = D.bestpath + (int64_t)slot * D.path_cap;
        uint32_t *mark = D.bestmark + (int64_t)slot * D.d.n;
        int32_t *pos = D.bestpos + (int64_t)slot * D.d.n;
        ver = ver + 1;
        for (int32_t i = tid; i < blen; i += blockDim.x) {
            int32_t w = src[i];
            dst[i] = w;
            mark[w & WaNbT<NB>::IDM] = ver;
            pos[w & WaNbT<NB>::IDM] = i;
        }
        bestL = iterL;
        if (tid == 0) { ctl->bestL = bestL; ctl->best_len = blen; ctl->best_ver = ver; ctl->tabu_gen = gen; }   // the replay-table rows rebuild the prefix-tabu bits
    }
    // ---- ranking
    if (R.rng_mode == 1) {  // DEV: ascending (L, ant) by counting
        for (int32_t a = tid; a < colony; a += blockDim.x) {
            int32_t r = 0;
            if (in_lds) {
                const unsigned long long ka = s_keys[a];
#pragma unroll 8
                for (int32_t b = 0; b < colony; b++) r += s_keys[b] < ka ? 1 : 0;
            } else {
                const unsigned long long ka = ((unsigned long long)__float_as_uint(antL[a]) << 32) | (uint32_t)a;
                for (int32_t b = 0; b < colony; b++) {
                    unsigned long long kb = ((unsigned long long)__float_as_uint(antL[b]) << 32) | (uint32_t)b;
                    r += kb < ka ? 1 : 0;
                }
            }
            perm[r] = a;
            // deposit coefficient of update_pheromone (:200,:211) for rank o = r + 1
            const int32_t o = r + 1;
            const float La = antL[a];
            const bool ok = !(La == INFINITY || (float)o > lambda - 1);
            depA[r] = ok ? (lambda - (float)o) * Q / La : 0.f;
            if (ok) atomicMax(&s_ndep, o);
        }
    } else {
        if (tid == 0) {  // REF: libstdc++'s permutation
            WaRec *rec = (WaRec *)(D.sortk + (int64_t)slot * D.max_colony * 2);
            for (int32_t a = 0; a < colony; a++) { rec[a].k = antL[a]; rec[a].t = a; }
            wa_std_sort(rec, colony);
            for (int32_t a = 0; a < colony; a++) perm[a] = rec[a].t;
        }
        __syncthreads();
        for (int32_t o = 1 + tid; o <= colony; o += blockDim.x) {
            float La = antL[perm[o - 1]];
            bool ok = !(La == INFINITY || (float)o > lambda - 1);
            depA[o - 1] = ok ? (lambda - (float)o) * Q / La : 0.f;
            if (ok) atomicMax(&s_ndep, o);
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (gen < D.trace_cap) {
            int64_t t = (int64_t)slot * D.trace_cap + gen;
            D.trBest[t] = bestL;
            D.trIter[t] = iterL;
            D.trColony[t] = colony;
            D.trFinite[t] = s_fin;
            D.trSteps[t] = (long long)s_steps;
        }
        WaSlotCtl c = *ctl;
        c.bestL = bestL;
        c.dep_lambda = lambda;
        c.dep_Q = Q;
        c.dep_bestL = bestL;
        c.n_dep = s_ndep;
        c.gen = gen + 1;
        wa_next_params(c, R, (gen + 1) & 1);
        *ctl = c;
    }
}

// ------------------------------------------------------------------ the evaporation sweep body
// :268-272 -- dst = src * rho over n_floats values; float4 per lane, 4 independent float4 in flight per
// thread, grid-stride over E blocks.  One definition for k_evaporate and the fused k_evap_rank_mark.
typedef float wa_v4f __attribute__((ext_vector_type(4)));
// NT bit 0 / bit 1: non-temporal loads / stores (the `nt` bit of global_load / global_store: the lines stream through the caches instead of
// displacing what is there).  Wrong for a lone search at 128^3 -- the next walk finds the swept field in the Infinity Cache -- and
// right when several searches share the GPU (their fields are past every cache anyway and the walking groups' records stay in L2)
template <int NT>
__device__ __forceinline__ wa_v4f wa_sweep_ld(const wa_v4f *p) { return (NT & 1) ? __builtin_nontemporal_load(p) : *p; }
template <int NT>
__device__ __forceinline__ void wa_sweep_st(wa_v4f *p, wa_v4f v)
{
    if (NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int NT>
__device__ __forceinline__ void wa_sweep_body_nt(const float *src, float *dst, int64_t n_floats, float rho, int32_t ebx, int32_t E)
{
    const wa_v4f *s4 = reinterpret_cast<const wa_v4f *>(src);
    wa_v4f *d4 = reinterpret_cast<wa_v4f *>(dst);
    const int64_t n4 = n_floats >> 2;
    const int64_t gsz = (int64_t)E * blockDim.x;
    int64_t i = (int64_t)ebx * blockDim.x + threadIdx.x;
    for (; i + 3 * gsz < n4; i += 4 * gsz) {
        wa_v4f a = wa_sweep_ld<NT>(s4 + i), b = wa_sweep_ld<NT>(s4 + i + gsz), c = wa_sweep_ld<NT>(s4 + i + 2 * gsz), d = wa_sweep_ld<NT>(s4 + i + 3 * gsz);
        a *= rho; b *= rho; c *= rho; d *= rho;
        wa_sweep_st<NT>(d4 + i, a); wa_sweep_st<NT>(d4 + i + gsz, b); wa_sweep_st<NT>(d4 + i + 2 * gsz, c); wa_sweep_st<NT>(d4 + i + 3 * gsz, d);
    }
    for (; i < n4; i += gsz) {
        wa_v4f a = wa_sweep_ld<NT>(s4 + i);
        a *= rho;
        wa_sweep_st<NT>(d4 + i, a);
    }
    // tail (n_floats is even; at most 2 floats)
    const int64_t t = (n4 << 2) + (int64_t)ebx * blockDim.x + threadIdx.x;
    if (t < n_floats) dst[t] = src[t] * rho;
}
__device__ __forceinline__ void wa_sweep_body(const float *src, float *dst, int64_t n_floats, float rho, int32_t ebx, int32_t E, int32_t nt = 0)
{
    switch (nt & 3) {   // (uniform over the launch)
    case 0: wa_sweep_body_nt<0>(src, dst, n_floats, rho, ebx, E); break;
    case 1: wa_sweep_body_nt<1>(src, dst, n_floats, rho, ebx, E); break;
    case 2: wa_sweep_body_nt<2>(src, dst, n_floats, rho, ebx, E); break;
    default: wa_sweep_body_nt<3>(src, dst, n_floats, rho, ebx, E); break;
    }
}

// ------------------------------------------------------------------ fused post-walk launch (DEV mode)
// One launch = ranking and deposit marking (blocks [0, MB), MB = 8 x the most ranks that can deposit) + the
// evaporation sweep (blocks [MB, MB+E)):
// the sweep only touches the pheromone buffers, rank/mark only the ants' results and the rank
// masks, so they share a launch instead of three dependent kernel boundaries.  Every mark block
// re-derives the (L, ant) ranking in LDS (256 broadcast reads per thread); block 0 additionally
// PUBLISHES what k_rank publishes (global best, perm/depA for the apply pass, trace, the next
// generation's parameters -- into slot [(gen+1)&1], which nobody reads during this launch).
// Preconditions (checked by the host): DEV mode, colony <= WA_RANK_LDS, at most 64 depositing ranks.
// split_log2: mark blocks per depositing rank = 1 << this (C3, 500 generations: 8 blocks per rank 21.6 k gen/s, 4 22.1 k, 2 21.9 k;
// C5 with 224 searches per launch: 4 blocks 0.636 s, 2 0.622 s, 1 0.623 s) -- the host passes 2 or 1
template <bool SPARSE, int NB>
__global__ __launch_bounds__(256) void k_evap_rank_mark(WaAcsDev D, WaRun R, const float *src_base,
                                                        float *dst_base, int32_t E, int32_t gen, int32_t MB, int32_t split_log2, int32_t lazy_period,
                                                        int32_t sweep_nt)
{
    const int32_t slot = blockIdx.y, tid = threadIdx.x;
    // the MB rank/mark blocks come FIRST in the grid so that they are dispatched immediately and
    // their latency-bound work hides under the sweep blocks that follow
    if ((int32_t)blockIdx.x >= MB) {  // ---- sweep: dst = src * rho (same body as k_evaporate)
        if (!SPARSE) {
            wa_sweep_body(src_base + (int64_t)slot * D.pher_stride, dst_base + (int64_t)slot * D.pher_stride, (int64_t)NB * D.d.n, R.rho,
                          (int32_t)blockIdx.x - MB, E, sweep_nt);
        } else {
            // lazy evaporation, background pass: every lazy_period-th entry of the dirty list (phase = generation) is brought
            // current in place, so no record has more than ~lazy_period multiplications pending (whoever reads a record applies
            // the pending ones exactly, one rounding each: the period only trades this pass against those catch-ups; the host
            // passes 16 for a few searches per launch and 64 for 64 and more -- C5, 224 searches: 16 0.618 s, 32 0.583, 64 0.570,
            // 256 0.563; the 32-search pair planning of bench.py: 325 k / 320 k / 303 k pair-generations/s with 16 / 32 / 64)
            // A record is claimed by exchanging its stamp (the mark blocks of this launch claim the same way when a
            // voxel receives a deposit), so exactly one thread applies the pending multiplications.
            float *ph = dst_base + (int64_t)slot * D.pher_stride;
            const int32_t *list = D.dirty_list + (int64_t)slot * D.d.n;
            uint32_t *stamp = D.stamp + (int64_t)slot * D.d.n;
            const int32_t n0 = D.dcount[slot * 2];
            const uint32_t evap_now = D.ctl[slot].evap_base + (uint32_t)gen;
            const uint32_t target = evap_now + 2u;   // stamp of "current after this generation's evaporation"
            const float rho = R.rho;
            const int64_t first = (int64_t)(evap_now % (uint32_t)lazy_period);
            for (int64_t q = first + (int64_t)lazy_period * ((int64_t)((int32_t)blockIdx.x - MB) * blockDim.x + tid); q < n0;
                 q += (int64_t)lazy_period * E * blockDim.x) {
                const int32_t v = list[q];
                const uint32_t old = atomicExch(&stamp[v], target);
                if (old == target) continue;
#pragma unroll
                for (int k = 0; k < 6; k++) ph[(int64_t)v * 6 + k] = wa_catch_up(ph[(int64_t)v * 6 + k], target - old, rho);
            }
        }
        return;
    }
    // ---- rank + mark
    const int32_t mb = (int32_t)blockIdx.x;  // 0..511: (bx = mb & 7, rank bit = mb >> 3)
    WaSlotCtl *ctl = &D.ctl[slot];
    const int32_t colony = ctl->colony[gen & 1];
    const float lambda = ctl->lambda[gen & 1], Q = ctl->Q[gen & 1];
    const float *antL = D.antL + (int64_t)slot * D.max_colony;
    const int32_t *antLen = D.antLen + (int64_t)slot * D.max_colony;
    __shared__ unsigned long long s_keys[WA_RANK_LDS];
    __shared__ int32_t s_perm[WA_RANK_LDS], s_len[WA_RANK_LDS];
    __shared__ int32_t s_ndep, s_fin;
    __shared__ unsigned long long s_steps;
    if (tid == 0) { s_ndep = 0; s_fin = 0; s_steps = 0; }
    // This block is a chain of dependent global loads beside a sweep that saturates the memory system (every level costs 2-3 us there):
    // the ants' results are requested for ALL max_colony ants before the control block says how many there are (inside the allocation;
    // entries beyond the colony are never looked at), and every ant's length goes to LDS with its key, so that the ranked ant's length is
    // an LDS read: control block + results -> path words -> marks, three levels instead of five.
    const int32_t cmax = D.max_colony < WA_RANK_LDS ? D.max_colony : WA_RANK_LDS;
    for (int32_t a = tid; a < cmax; a += blockDim.x) {
        const float La = antL[a];
        const int32_t na = antLen[a];
        s_keys[a] = ((unsigned long long)__float_as_uint(La) << 32) | (uint32_t)a;
        s_len[a] = na;
    }
    if (colony > D.max_colony || colony > WA_RANK_LDS) {
        if (mb == 0 && tid == 0) atomicOr(&ctl->flags, WA_FLAG_COLONY_OVERFLOW);
        return;
    }
    __syncthreads();
    int32_t myfin = 0;
    unsigned long long mysteps = 0;
    if (mb == 0)
        for (int32_t a = tid; a < colony; a += blockDim.x) {
            myfin += (__uint_as_float((uint32_t)(s_keys[a] >> 32)) != INFINITY) ? 1 : 0;
            mysteps += (unsigned long long)(s_len[a] - 1);
        }
    for (int32_t a = tid; a < colony; a += blockDim.x) {  // ascending (L, ant) by counting (:273-275, DEV tie rule)
        const unsigned long long ka = s_keys[a];
        int32_t r = 0;
#pragma unroll 8
        for (int32_t b = 0; b < colony; b++) r += s_keys[b] < ka ? 1 : 0;
        s_perm[r] = a;
        const int32_t o = r + 1;
        const float La = __uint_as_float((uint32_t)(ka >> 32));
        const bool ok = !(La == INFINITY || (float)o > lambda - 1);  // :200
        if (ok) atomicMax(&s_ndep, o);
        if (mb == 0) {  // publish for the apply pass
            D.perm[(int64_t)slot * D.max_colony + r] = a;
            D.depA[(int64_t)slot * D.max_colony + r] = ok ? (lambda - (float)o) * Q / La : 0.f;  // :211
        }
    }
    if (mb == 0) {
        for (int o = 32; o > 0; o >>= 1) { myfin += __shfl_down(myfin, o, 64); mysteps += __shfl_down(mysteps, o, 64); }
        if ((tid & 63) == 0) { atomicAdd(&s_fin, myfin); atomicAdd(&s_steps, mysteps); }
    }
    __syncthreads();
    const int32_t n_dep = s_ndep;
    if (mb == 0) {  // ---- publish: iteration best -> global best (:263-264), trace, next parameters (:247-249)
        float iterL = INFINITY;
        int32_t iterAnt = -1;
        if (colony > 0) { iterAnt = s_perm[0]; iterL = __uint_as_float((uint32_t)(s_keys[iterAnt] >> 32)); }  // rank 1 = first ant with the minimal L
        float bestL = ctl->bestL;
        uint32_t ver = ctl->best_ver;
        int32_t blen = ctl->best_len;
        bool changed = false;
        if (iterAnt >= 0 && iterL < bestL) {
            blen = s_len[iterAnt];
            const int32_t *srcp = D.paths + ((int64_t)slot * D.max_colony + iterAnt) * D.path_cap;
            int32_t *dstp = D.bestpath + (int64_t)slot * D.path_cap;
            uint32_t *mark = D.bestmark + (int64_t)slot * D.d.n;
            int32_t *pos = D.bestpos + (int64_t)slot * D.d.n;
            ver = ver + 1;
            for (int32_t i = tid; i < blen; i += blockDim.x) {
                int32_t w = srcp[i];
                dstp[i] = w;
                mark[w & WaNbT<NB>::IDM] = ver;
                pos[w & WaNbT<NB>::IDM] = i;
            }
            bestL = iterL;
            changed = true;
        }
        if (tid == 0) {
            if (gen < D.trace_cap) {
                int64_t t = (int64_t)slot * D.trace_cap + gen;
                D.trBest[t] = bestL;
                D.trIter[t] = iterL;
                D.trColony[t] = colony;
                D.trFinite[t] = s_fin;
                D.trSteps[t] = (long long)s_steps;
            }
            // written in place, field by field (a local copy indexed by the generation's parity lives in scratch memory: the launch
            // then needs a scratch set-up); nothing a sibling block reads during this launch changes: [gen&1] slots, start/end/stream
            ctl->bestL = bestL;
            ctl->best_len = blen;
            ctl->best_ver = ver;
            ctl->dep_lambda = lambda;
            ctl->dep_Q = Q;
            ctl->dep_bestL = bestL;
            ctl->n_dep = n_dep;
            ctl->gen = gen + 1;
            if (changed) ctl->tabu_gen = gen;   // the replay-table rows of this generation rebuild the prefix-tabu bits
            ctl->clean[(gen + 1) & 1] = ctl->clean[gen & 1] * R.rho;   // what one more evaporation makes of a never-deposited edge
            wa_next_params(*ctl, R, (gen + 1) & 1);
        }
    }
    // ---- mark: OR bit (o-1) into the rank mask of every directed edge of ranked ant o
    const int32_t bit = mb >> split_log2, bx = mb & ((1 << split_log2) - 1), o = bit + 1;
    if (o > n_dep) return;
    const int32_t a = s_perm[o - 1];
    const int32_t len = s_len[a];
    const int32_t *path = D.paths + ((int64_t)slot * D.max_colony + a) * D.path_cap;
    const WaMaskRef mask = wa_mask_of(D, slot);
    float *ph = dst_base + (int64_t)slot * D.pher_stride;
    const float clean_next = ctl->clean[gen & 1] * R.rho;   // == what block 0 publishes into clean[(gen+1)&1]
    for (int32_t i = 1 + bx * blockDim.x + tid; i < len; i += (blockDim.x << split_log2)) {
        int32_t w = path[i];
        int32_t v = path[i - 1] & WaNbT<NB>::IDM;
        int64_t e = (int64_t)v * NB + ((uint32_t)w >> WaNbT<NB>::SHIFT);
        wa_mask_or(mask, e, bit);
        if (SPARSE) {   // v receives a deposit: its record must be current (after this generation's evaporation) for the apply pass
            uint32_t *stamp = D.stamp + (int64_t)slot * D.d.n;
            const uint32_t target = ctl->evap_base + (uint32_t)gen + 2u;
            const uint32_t old = atomicExch(&stamp[v], target);
            if (old == 0) {            // first deposit ever: v joins the dirty list, its record is written at the clean value
                const int32_t idx = atomicAdd(&D.dcount[slot * 2 + 1], 1);
                D.dirty_list[(int64_t)slot * D.d.n + idx] = v;
#pragma unroll
                for (int k = 0; k < 6; k++) {   // stored = the init value: 0 stays 0 (out-of-bounds edge of initFromGridMap), p0 became clean_next
                    const float st0 = ph[(int64_t)v * 6 + k];
                    ph[(int64_t)v * 6 + k] = copysignf(fabsf(st0) == 0.f ? 0.f : clean_next, st0);
                }
            } else if (old != target) {   // deposited before: apply the evaporations it has missed since
#pragma unroll
                for (int k = 0; k < 6; k++) ph[(int64_t)v * 6 + k] = wa_catch_up(ph[(int64_t)v * 6 + k], target - old, R.rho);
            }
        }
    }
}

// ------------------------------------------------------------------ evaporation (the HBM sweep)
// :268-272 -- every edge of every voxel, occupied voxels and out-of-bounds edges included:
// dst = src * rho over 6N floats, 48 B of traffic per voxel (24 read + 24 written).  The
// pheromone field is double-buffered: the sweep is out of place (the buffer it reads stays intact until the next
// sweep -- what a resumed straggler walks on); src == dst is allowed (in-place).  float4 per lane, 4 independent
// float4 in flight per thread, grid-stride.  sweep_nt: see wa_sweep_body.
__global__ __launch_bounds__(256) void k_evaporate(const float *src_base, float *dst_base,
                                                   int64_t stride, int64_t n_floats, float rho, int32_t sweep_nt)
{
    wa_sweep_body(src_base + (int64_t)blockIdx.y * stride, dst_base + (int64_t)blockIdx.y * stride, n_floats, rho,
                  (int32_t)blockIdx.x, (int32_t)gridDim.x, sweep_nt);
}

// ------------------------------------------------------------------ ranked deposit
// update_pheromone (:198-215) adds, per ranked ant in rank order, a float to every directed edge
// of its path.  Float adds do not commute, so instead of atomics: pass 1 ORs bit (o-1-base) into
// a per-edge rank mask; pass 2 lets the LOWEST rank present on an edge own it and apply all
// present ranks in ascending order (= the reference's order), then clear the mask.
// grid = (blocks, 64 ranks, n_problems)
template <int NB>
__global__ __launch_bounds__(256) void k_deposit_mark(WaAcsDev D, int32_t base)
{
    const int32_t slot = blockIdx.z, bit = blockIdx.y, o = base + bit + 1;
    const WaSlotCtl *c = &D.ctl[slot];
    if (o > c->n_dep) return;
    const int32_t a = D.perm[(int64_t)slot * D.max_colony + o - 1];
    const int32_t len = D.antLen[(int64_t)slot * D.max_colony + a];
    const int32_t *path = D.paths + ((int64_t)slot * D.max_colony + a) * D.path_cap;
    const WaMaskRef mask = wa_mask_of(D, slot);
    for (int32_t i = 1 + blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
        int32_t w = path[i];
        int32_t v = path[i - 1] & WaNbT<NB>::IDM;
        int64_t e = (int64_t)v * NB + ((uint32_t)w >> WaNbT<NB>::SHIFT);
        wa_mask_or(mask, e, bit);
    }
}
// Body of the apply pass for rank bit `bit` of chunk `base`, x-block `bx` of `nbx`.  skip_best_src: edges that
// leave a best-path node belong to the replay-table rows of the same launch (k_apply_table).
template <int NB>
__device__ __forceinline__ void wa_apply_body(const WaAcsDev &D, int32_t slot, int32_t base, int32_t bit, int32_t bx, int32_t nbx,
                                              bool skip_best_src, float *s_dep)
{
    // The kernel is a chain of dependent global loads (control block -> rank -> ant -> path word -> edge record),
    // so loads are issued as early as their addresses are known, speculatively where a bound is not yet known
    // (always inside the allocation): three dependent levels instead of eight.
    const int32_t o = base + bit + 1;
    const int32_t tid = threadIdx.x, C = D.max_colony;
    const WaSlotCtl *c = &D.ctl[slot];
    // level 1: addresses that depend only on the launch geometry
    const float dep_mine = (tid < 64 && base + tid < C) ? D.depA[(int64_t)slot * C + base + tid] : 0.f;
    const int32_t a = o - 1 < C ? D.perm[(int64_t)slot * C + o - 1] : 0;
    const int32_t n_dep = c->n_dep;
    const uint32_t ver = c->best_ver;
    const float lambda = c->dep_lambda, Q = c->dep_Q, bestL = c->dep_bestL;
    if (o > n_dep) return;
    if (tid < 64) s_dep[tid] = base + tid < n_dep ? dep_mine : 0.f;
    // level 2: the ranked ant's length and this thread's first path words
    const int32_t *path = D.paths + ((int64_t)slot * C + a) * D.path_cap;
    const int32_t i0 = 1 + bx * (int32_t)blockDim.x + tid;
    const int32_t len = D.antLen[(int64_t)slot * C + a];
    int32_t w = i0 < D.path_cap ? path[i0] : 0, pv = i0 < D.path_cap ? path[i0 - 1] : 0;
    __syncthreads();
    const WaMaskRef mask = wa_mask_of(D, slot);
    float *pher = D.pher + (int64_t)slot * D.pher_stride;
    const uint32_t *mark = D.bestmark + (int64_t)slot * D.d.n;
    for (int32_t i = i0; i < len; i += nbx * (int32_t)blockDim.x) {
        if (i != i0) { w = path[i]; pv = path[i - 1]; }
        const int32_t v = pv & WaNbT<NB>::IDM;
        const int64_t e = (int64_t)v * NB + ((uint32_t)w >> WaNbT<NB>::SHIFT);
        // level 3: four independent loads
        const uint32_t mv = mark[v], mw = mark[w & WaNbT<NB>::IDM];
        unsigned long long m = wa_mask_get(mask, e);
        float p = pher[e];
        const bool v_best = mv == ver;
        if (skip_best_src && v_best) continue;
        if (m == 0 || (__ffsll((long long)m) - 1) != bit) continue;  // not the owner
        const bool onbest = v_best && mw == ver;                      // :209
        const float bonus = (float)onbest * lambda * Q / bestL;       // second term of :211, the same for every rank
        while (m) {
            int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            p += s_dep[b] + bonus;  // :210-211
        }
        pher[e] = p;
        wa_mask_clear(mask, e);
    }
}

template <int NB>
__global__ __launch_bounds__(256) void k_deposit_apply(WaAcsDev D, int32_t base)
{
    __shared__ float s_dep_[64];
    wa_apply_body<NB>(D, blockIdx.z, base, blockIdx.y, blockIdx.x, gridDim.x, false, s_dep_);
}

// ------------------------------------------------------------------ lazy evaporation: reset / read-back helpers
// after k_init_pheromone (every record holds its init value): nothing is dirty, the clean value is p0
__global__ void k_lazy_clear(WaAcsDev D, int32_t slot0, int32_t cnt, float p0)
{
    const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cnt) return;
    const int32_t slot = slot0 + q;
    D.dcount[slot * 2] = 0;
    D.dcount[slot * 2 + 1] = 0;
    D.ctl[slot].clean[0] = p0;
    D.ctl[slot].clean[1] = p0;
    D.ctl[slot].evap_base = 0;
    D.ctl[slot].gen = 0;
}
// reset() of a lazy slot whose init mode and p0 are unchanged: only the dirty records are rewritten
// (same values as k_init_pheromone) and their flags cleared.  grid.y = slots.
__global__ __launch_bounds__(256) void k_lazy_restore(WaAcsDev D, int32_t slot0, float p0, int32_t mode)
{
    const int32_t slot = slot0 + blockIdx.y;
    const int32_t n = D.dcount[slot * 2 + 1];
    const int32_t *list = D.dirty_list + (int64_t)slot * D.d.n;
    uint32_t *stamp = D.stamp + (int64_t)slot * D.d.n;
    float *ph = D.pher + (int64_t)slot * D.pher_stride;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const int32_t id = list[q];
        const int32_t x = id % D.d.nx, y = (id / D.d.nx) % D.d.ny, z = id / D.d.nxy;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int32_t X = x + (k == 2 ? -1 : k == 3 ? 1 : 0), Y = y + (k == 1 ? -1 : k == 4 ? 1 : 0),
                          Z = z + (k == 0 ? -1 : k == 5 ? 1 : 0);
            const bool inb = X >= 0 && X < D.d.nx && Y >= 0 && Y < D.d.ny && Z >= 0 && Z < D.d.nz;
            const bool adm = inb && D.occ[id + wa_delta(k, D.d.nx, D.d.nxy)] != 0;
            const float v = (inb || mode == 1) ? p0 : 0.f;
            ph[(int64_t)id * 6 + k] = adm ? v : -v;
        }
        stamp[id] = 0;
    }
}
// bring every deposited record current (before a solve that evaporates with a different rho: the pending
// multiplications belong to the old one).  grid.y = slots.
__global__ __launch_bounds__(256) void k_lazy_flush(WaAcsDev D, float rho_old)
{
    const int32_t slot = blockIdx.y;
    const int32_t n = D.dcount[slot * 2 + 1];
    const int32_t *list = D.dirty_list + (int64_t)slot * D.d.n;
    uint32_t *stamp = D.stamp + (int64_t)slot * D.d.n;
    float *ph = D.pher + (int64_t)slot * D.pher_stride;
    const WaSlotCtl *c = &D.ctl[slot];
    const uint32_t target = c->evap_base + (uint32_t)c->gen + 1u;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const int32_t v = list[q];
        const uint32_t old = stamp[v];
        if (old == target) continue;
#pragma unroll
        for (int k = 0; k < 6; k++) ph[(int64_t)v * 6 + k] = wa_catch_up(ph[(int64_t)v * 6 + k], target - old, rho_old);
        stamp[v] = target;
    }
}

// the field as the dense sweep would have left it: deposited records with their pending evaporations applied,
// clean records at the clean value
__global__ __launch_bounds__(256) void k_lazy_materialise(WaAcsDev D, WaRun R, int32_t slot, float *out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= D.d.n * 6) return;
    const int64_t v = e / 6;
    const float st0 = D.pher[(int64_t)slot * D.pher_stride + e];
    const WaSlotCtl *c = &D.ctl[slot];
    const float clean = c->clean[c->gen & 1];
    const uint32_t evap_now = c->evap_base + (uint32_t)c->gen;
    const uint32_t stv = D.stamp[(int64_t)slot * D.d.n + v];
    out[e] = stv != 0 ? wa_catch_up(fabsf(st0), evap_now + 1u - stv, R.rho) : (fabsf(st0) == 0.f ? 0.f : clean);
}

// =====================================================================================================
// 26-neighbour variant (ACSRank_3D.hpp:352-388 with the two distances the reference keeps
// in comments restored: edge neighbours precision*1.414f, corner neighbours precision*1.732f).
// Edge order = the reference's cube loop: z offset outermost, then y, then x, centre skipped.
// Same selectNext, same ranking, same deposit; pheromone / heuristic / rank-mask fields are [N][26].
// One wavefront per ant, lane k < 26 owns neighbour k: LDS hash tabu with bitmap spill, ordered sums as
// whole-wave DPP chains, cache-warming loads for the next step's records, best-path replay (k_replay_table26).
// =====================================================================================================
// ---- best-path replay for the 26-neighbour walk: same idea as wa_walk_replay / k_replay_table.
// Row of best-path node i = 32 floats: thr[26] (admissible ? prob_sum : -inf), total, edge taken to best[i+1],
// L accumulated on arrival at node i (the in-order sum of the step lengths, which differ per move type here), pad.
#define WA_ROW26 32
// arrival lengths of the best path: one sequential fp32 chain in walk order (:78).  The step lengths are fetched and
// classified by the whole block (tiles of 1024 through LDS); thread 0 only adds.
__device__ __forceinline__ void wa_table26_lengths(const WaAcsDev &D, const WaRun &R, int32_t slot, float *s_d)
{
    const WaSlotCtl *ctl = &D.ctl[slot];
    if (ctl->bestL == INFINITY) return;
    const int32_t blen = ctl->best_len;
    const int32_t *bpath = D.bestpath + (int64_t)slot * D.path_cap;
    float *T = D.rtab + (int64_t)slot * D.path_cap * WA_ROW26;
    const float d1 = R.precision, d2 = R.precision * 1.414f, d3 = R.precision * 1.732f;
    float L = 0.f;
    if (threadIdx.x == 0) T[28] = L;
    for (int32_t base = 1; base < blen; base += 1024) {
        const int32_t cnt = blen - base < 1024 ? blen - base : 1024;
        for (int32_t q = threadIdx.x; q < cnt; q += blockDim.x) {
            int px, py, pz;
            wa_off26((int)((uint32_t)bpath[base + q] >> WaNbT<26>::SHIFT), px, py, pz);
            const int type = (px != 0) + (py != 0) + (pz != 0);
            s_d[q] = type == 1 ? d1 : type == 2 ? d2 : d3;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int32_t q = 0; q < cnt; q++) {
                L += s_d[q];
                T[(int64_t)(base + q) * WA_ROW26 + 28] = L;
            }
        __syncthreads();
    }
}
// One wavefront per best-path node (wave w of n_waves takes nodes w, w + n_waves, ...): the walk's own step evaluation with
// visited set = best[0..i].  apply_here: the row first applies the pending ranked deposits (mask != 0) of its 26 edges -- same
// adds, same ascending rank order as wa_apply_body, which skips edges leaving a best-path node when it shares the launch.
__device__ __forceinline__ void wa_table26_rows(const WaAcsDev &D, const WaRun &R, int32_t slot, int32_t w, int32_t n_waves, bool apply_here,
                                                const float *s_dep)
{
    const int lane = threadIdx.x & 63;
    const WaSlotCtl *ctl = &D.ctl[slot];
    if (ctl->bestL == INFINITY) return;
    const int32_t blen = ctl->best_len;
    const uint32_t ver = ctl->best_ver;
    const float lambda = ctl->dep_lambda, Q = ctl->dep_Q, bestL = ctl->dep_bestL;
    const int32_t *bpath = D.bestpath + (int64_t)slot * D.path_cap;
    const uint32_t *mark = D.bestmark + (int64_t)slot * D.d.n;
    const int32_t *pos = D.bestpos + (int64_t)slot * D.d.n;
    float *pher = D.pher + (int64_t)slot * D.pher_stride;
    const WaMaskRef mask = wa_mask_of(D, slot);
    const float *heur = D.heur + (int64_t)D.ctl[slot].heur_slot * D.pher_stride;
    float *T = D.rtab + (int64_t)slot * D.path_cap * WA_ROW26;
    const int k = lane < 26 ? lane : 25;
    int dx, dy, dz;
    wa_off26(k, dx, dy, dz);
    const int32_t dk = dz * D.d.nxy + dy * D.d.nx + dx;
    const int32_t last_id = (int32_t)D.d.n - 1;
    for (int32_t i = w; i < blen - 1; i += n_waves) {   // decisions exist at nodes 0 .. blen-2
        const int32_t v = bpath[i] & WaNbT<26>::IDM;
        float p = -0.f, h = 0.f;
        bool adm = false;
        if (lane < 26) {
            const int64_t e = (int64_t)v * 26 + lane;
            p = pher[e];
            h = heur[e];
            int32_t nb = v + dk;
            nb = nb < 0 ? 0 : nb > last_id ? last_id : nb;    // (an out-of-bounds edge is inadmissible by its sign bit whatever is found here)
            const uint32_t mk = mark[nb];
            unsigned long long m = apply_here ? wa_mask_get(mask, e) : 0ULL;
            if (m) {  // somebody walked (v, lane): the ranked deposits in ascending rank order (:210-211); v is on the best path (:209)
                const float bonus = (float)(mk == ver) * lambda * Q / bestL;
                while (m) {
                    const int bq = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    p += s_dep[bq] + bonus;
                }
                pher[e] = p;
                wa_mask_clear(mask, e);
            }
            if ((__float_as_uint(p) >> 31) == 0)              // in bounds and free (:148)
                adm = !(mk == ver && pos[nb] <= i);           // not on the prefix best[0..i] (:145-146)
        }
        const float info = wa_powi(fabsf(p), R.alpha) * h;    // :154
        const float a = adm ? info : 0.f;
        float t = 0.f + a, c = 0.f + a;
#pragma unroll
        for (int q = 0; q < 25; q++) {
            t = dpp_wave_from_below(t) + a;
            c = dpp_wave_from_above(c) + a;
        }
        float *row = T + (int64_t)i * WA_ROW26;
        if (lane < 26) row[lane] = adm ? c : -INFINITY;
        if (lane == 25) row[26] = t;
        if (lane == 0) row[27] = __int_as_float((int32_t)((uint32_t)bpath[i + 1] >> WaNbT<26>::SHIFT));
    }
}
__global__ __launch_bounds__(64) void k_replay_table26(WaAcsDev D, WaRun R)
{
    __shared__ float s_d[1024];
    if (blockIdx.x == 0) wa_table26_lengths(D, R, blockIdx.y, s_d);
    else wa_table26_rows(D, R, blockIdx.y, (int32_t)blockIdx.x - 1, (int32_t)gridDim.x - 1, false, nullptr);
}
// Deposit apply + replay table of the 26-neighbour search in ONE launch, like k_apply_table: block 0 = arrival lengths,
// blocks [1, 1 + WA_TABLE26_BLOCKS) = table rows (four wavefronts each) that also apply the deposits on edges leaving a
// best-path node, the rest = the ordinary apply pass, which skips exactly those edges.
#define WA_TABLE26_BLOCKS 64
__global__ __launch_bounds__(256) void k_apply_table26(WaAcsDev D, WaRun R)
{
    __shared__ float s_d[1024];
    __shared__ float s_dep[64];
    const int32_t slot = blockIdx.y, tid = threadIdx.x;
    if (D.pool_n && blockIdx.x == 0) {   // stragglers: the next generation starts with no arrivals and an empty pool of its own (see k_apply_table)
        const WaStrag sg = wa_strag_of(D, slot);
        sg.arr_len[tid] = 0xffffffffu;
        if (tid == 0) { *sg.arr_n = 0; sg.pool_n[D.ctl[slot].gen & 1] = 0; }
    }
    if (blockIdx.x == 0) { wa_table26_lengths(D, R, slot, s_d); return; }
    if ((int32_t)blockIdx.x <= WA_TABLE26_BLOCKS) {
        const float dep_mine = (tid < 64 && tid < D.max_colony) ? D.depA[(int64_t)slot * D.max_colony + tid] : 0.f;
        const int32_t n_dep = D.ctl[slot].n_dep;
        if (tid < 64) s_dep[tid] = tid < n_dep ? dep_mine : 0.f;
        __syncthreads();
        wa_table26_rows(D, R, slot, ((int32_t)blockIdx.x - 1) * 4 + (tid >> 6), WA_TABLE26_BLOCKS * 4, true, s_dep);
        return;
    }
    const int32_t ab = (int32_t)blockIdx.x - 1 - WA_TABLE26_BLOCKS;  // (bx = ab & 7, rank bit = ab >> 3)
    wa_apply_body<26>(D, slot, 0, ab >> 3, ab & 7, 8, true, s_dep);
}

// one lane per node, 64 nodes per ballot.  Returns 1 dead end at node i, 2 arrived, 3 deviates at node i (i in `node`).
__device__ __forceinline__ int wa_walk_replay26(const float *__restrict__ T, int32_t rlen, uint64_t antkey, int32_t &node)
{
    const int lane = threadIdx.x;
    const float4 *__restrict__ T4 = reinterpret_cast<const float4 *>(T);
    const int32_t last = rlen - 1;
    for (int32_t i0 = 0;; i0 += 64) {
        const int32_t nodev = i0 + lane;
        const bool valid = nodev < last;
        const int32_t nv = valid ? nodev : last - 1;
        float4 r[7];
#pragma unroll
        for (int q = 0; q < 7; q++) r[q] = T4[(int64_t)nv * (WA_ROW26 / 4) + q];
        float rnd = (float)wa_ctr_draw(antkey, (uint32_t)nodev) / 2147483648.0f;   // :169
        rnd *= r[6].z;                                                              // total (:170)
        const int nk = __float_as_int(r[6].w);
        uint32_t h = 0;
#pragma unroll
        for (int q = 0; q < 7; q++) {
            h |= (r[q].x >= rnd ? 1u : 0u) << (4 * q);
            h |= (r[q].y >= rnd ? 1u : 0u) << (4 * q + 1);
            if (q < 6) {
                h |= (r[q].z >= rnd ? 1u : 0u) << (4 * q + 2);
                h |= (r[q].w >= rnd ? 1u : 0u) << (4 * q + 3);
            }
        }
        const int pick = h ? 31 - __clz((int)h) : -1;     // first hit scanning 25..0 (:172-189)
        const unsigned long long fm = __ballot(valid && pick != nk);
        if (__builtin_expect(fm != 0, 0)) {
            const int g = __ffsll((long long)fm) - 1;
            node = i0 + g;
            return __builtin_amdgcn_readlane((int)h, g) ? 3 : 1;
        }
        if (i0 + 64 >= last) { node = last; return 2; }
    }
}

template <int MODE>
__device__ __forceinline__ void wa_walk_one26(const WaAcsDev &D, const WaRun &R, int32_t slot, int32_t ant, int32_t start,
                                              int32_t end, uint64_t antkey, int32_t *tab, int hash_log2, int32_t &rng_rs,
                                              int32_t &rng_f, int32_t &rng_b, int32_t *flags_out, int32_t rlen,
                                              int32_t cut_n = 0x7fffffff, int32_t *res_words = nullptr, int32_t res_len = 0, float res_L = 0.f,
                                              int32_t gen = 0, int32_t bits_row = -1, bool drain = false)
{
    // Stragglers (see wa_walk_one / k_walk_dev): step lengths differ per move type here, so the arrivals publish the bits of
    // their L (positive floats order like unsigned integers) and an ant compares the L it has accumulated so far -- a lower bound of
    // its final L, every step adds a positive length -- against them.  res_words != nullptr: a resume block, which finishes the
    // straggler whose path so far (res_len nodes, length res_L) stands in res_words and goes on writing there.
    const int lane = threadIdx.x;
    const WaStrag sg = wa_strag_of(D, slot);   // (only dereferenced where D.pool_n is set)
    const float *pher = D.pher + (int64_t)slot * D.pher_stride;
    const float *heur = D.heur + (int64_t)D.ctl[slot].heur_slot * D.pher_stride;
    int32_t *path = res_words ? res_words : D.paths + ((int64_t)slot * D.max_colony + ant) * D.path_cap;
    const int32_t *pfx = res_words ? res_words : D.bestpath + (int64_t)slot * D.path_cap;   // where the walked prefix stands
    const bool cutting = MODE == 1 && cut_n != 0x7fffffff;
    auto publish = [&](float Larr) {   // an arrival, for the straggler check (write-through: the checking ants sit on other XCDs)
        if (cutting && lane == 0 && Larr != INFINITY)
            __hip_atomic_store(&sg.arr_len[atomicAdd(sg.arr_n, 1u) & 255u], __float_as_uint(Larr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto finish = [&](float Lf, int32_t lenf) {   // agents[] of an ant -- or, for a resumed straggler, the rest of its generation's statistics
        if (res_words) {
            if (lane == 0) {
                if (D.dbg) atomicAdd(&D.dbg[7], 1ULL);
                atomicAdd(&D.strag_cnt[slot * 2 + 1], 1ULL);
                if (gen < D.trace_cap) {
                    const int64_t t = (int64_t)slot * D.trace_cap + gen;
                    if (Lf != INFINITY) atomicAdd(&D.trFinite[t], 1);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&D.trSteps[t]), (unsigned long long)(lenf - res_len));
                }
            }
            if (drain) {   // drain launch (see k_walk_dev): the finished walk goes back to agents[]
                int32_t *own = D.paths + ((int64_t)slot * D.max_colony + ant) * D.path_cap;
                __threadfence();
                for (int32_t q = lane; q < lenf; q += 64) own[q] = __hip_atomic_load(&res_words[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) {
                    D.antL[(int64_t)slot * D.max_colony + ant] = Lf;
                    D.antLen[(int64_t)slot * D.max_colony + ant] = lenf;
                }
            }
            return;
        }
        if (lane == 0) {
            D.antL[(int64_t)slot * D.max_colony + ant] = Lf;
            D.antLen[(int64_t)slot * D.max_colony + ant] = lenf;
        }
        publish(Lf);
    };
    int32_t r_node = 0;
    float r_L = 0.f;
    if (res_words) { r_node = res_len - 1; r_L = res_L; }
    else if (MODE == 1 && rlen > 1) {   // follow the best path while the ant's own draws take its edges
        const float *RT = D.rtab + (int64_t)slot * D.path_cap * WA_ROW26;
        const int32_t *bpath = D.bestpath + (int64_t)slot * D.path_cap;
        const int what = wa_walk_replay26(RT, rlen, antkey, r_node);
        for (int32_t q = lane; q <= r_node; q += 64) path[q] = bpath[q];   // the walked prefix IS the best path's
        r_L = RT[(int64_t)r_node * WA_ROW26 + 28];                          // L on arrival at that node
        if (what != 3) {
            finish(what == 2 ? r_L : INFINITY, r_node + 1);
            return;
        }
    }
    WaTabu T;
    T.tab = tab;
    T.mask = (1u << hash_log2) - 1u;
    T.shift = 32 - hash_log2;
    // (a resume block spills into a bitmap row of its own, behind the ants' rows)
    T.bits = D.vbits + ((int64_t)slot * D.vbits_rows + (bits_row >= 0 ? bits_row : ant)) * D.vbits_words;
    T.spilled = false;
    const int32_t spill_at = (int32_t)((3u << hash_log2) >> 2);
    int4 *tab4 = reinterpret_cast<int4 *>(tab);
    wa_tabu_clear(tab4, hash_log2);
    __builtin_amdgcn_wave_barrier();
    if (r_node > 0) {   // deviated at best[r_node] (or resumed): tabu set := the walked prefix (distinct keys: concurrent CAS inserts)
        if (r_node + 1 <= spill_at) {
            for (int32_t q = lane; q <= r_node; q += 64) {
                const int32_t key = pfx[q] & WaNbT<26>::IDM;
                uint32_t h = ((uint32_t)key * 2654435761u) >> T.shift;
                while (atomicCAS(&tab[h], WA_HASH_EMPTY, key) != WA_HASH_EMPTY) h = (h + 1) & T.mask;
            }
        }   // (a longer prefix goes straight to the bitmap: the loop below spills from path[] when len > spill_at)
    } else if (lane == 0) {
        tabu_insert(T, start);  // addStartNode :81-86
        path[0] = start;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    // lane constants: neighbour offset, in-bounds test inputs, step length by move type (:369-385)
    const int k = lane < 26 ? lane : 25;
    int dx, dy, dz;
    wa_off26(k, dx, dy, dz);
    const int32_t dk = dz * D.d.nxy + dy * D.d.nx + dx;
    const float d1 = R.precision, d2 = R.precision * 1.414f, d3 = R.precision * 1.732f;
    int32_t cur = start, len = 1;
    uint32_t step = 0;
    float L = 0.f;
    if (r_node > 0) {
        cur = __builtin_amdgcn_readfirstlane(pfx[r_node] & WaNbT<26>::IDM);
        len = r_node + 1;
        step = (uint32_t)r_node;   // steps taken so far = draws consumed
        L = r_L;
    }
    float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
    const int64_t last_rec = (D.d.n - 1) * 26;
    if (MODE == 1 && R.alpha == 1 && len <= spill_at && len < (int32_t)D.path_cap && 104LL * D.d.n < (1LL << 31)) {
        // ---- the general step while the tabu set lives in the LDS hash (DEV mode, alpha == 1, fields below 2 GB so that byte offsets fit
        // 32 bits): the same arithmetic as the loop below, with what a LONE wavefront pays for taken out of the step (it issues one
        // instruction per ~4 cycles whatever the type, see walk_loop_gfx950.hpp): the 64 draws of a block of steps are formed at once
        // (lane i = step base + i) and picked with a readlane; offset, path word and step length of the pick come out of lane constants
        // with one readlane each instead of the cube arithmetic; path words collect in a register and leave as one 256-byte store per
        // 64 steps; the probe's terminating empty slot of the picked lane is the insertion slot (one ds_write, no second chain walk);
        // the records of the NEXT voxel and its tabu probe are requested right behind the pick, before the bookkeeping, and the touch
        // loads for the step after that follow them; addresses are 32-bit byte offsets from scalar bases.
        const uint32_t dkw = (uint32_t)dk + ((uint32_t)k << WaNbT<26>::SHIFT);   // cur + this = path word of the move along edge k
        const int typek = (dx != 0) + (dy != 0) + (dz != 0);
        const float dlen = typek == 1 ? d1 : typek == 2 ? d2 : d3;                 // :369-385
        const uint32_t hk = (uint32_t)dk * 2654435761u;                            // hash(cur + dk) = cur * K + dk * K
        const char *pher_b = reinterpret_cast<const char *>(pher), *heur_b = reinterpret_cast<const char *>(heur);
        const uint32_t lane_off = (uint32_t)k * 4u;                                // this lane's edge inside a 104-byte record
        const int32_t last_vox = (int32_t)D.d.n - 1;
        int32_t pbuf = 0;                                                          // lane i = path word (len & ~63) + i
        if (r_node > 0) { if (lane < (len & 63)) pbuf = pfx[(len & ~63) + lane]; }
        else pbuf = start;                                                         // (lane 0 is the only one that counts: len == 1)
        asm volatile("" : "+v"(pbuf));   // the load above is waited for HERE: left pending, the compiler's waitcnt pass puts a vmcnt(0) in front of the
                                         // loop's v_writelane into this register -- i.e. waits for the touch loads in every step
        float ublock = (float)wa_ctr_draw(antkey, (step & ~63u) + (uint32_t)lane) / 2147483648.0f;   // (float)rand()/(float)RAND_MAX (:169)
        // Vector memory returns in order and the compiler's waitcnt pass would wait for the youngest load it knows: the loop's six loads
        // per step are therefore inline statements with an exact wait -- the two record loads (needed at the top of the next step) are
        // issued FIRST, the four touch loads behind them land in registers nobody reads (v250..v253, never allocated otherwise: the
        // kernel needs ~30) and stay in flight across the `s_waitcnt vmcnt(4)`.  (Loads the pass does not see only make its own waits
        // stricter than it thinks, never weaker.)
        float p = -0.f, h = 0.f;
        {
            const uint32_t off = (uint32_t)cur * 104u + lane_off;
            asm volatile("global_load_dword %0, %2, %3\n global_load_dword %1, %2, %4\n s_waitcnt vmcnt(0)"
                         : "=&v"(p), "=&v"(h) : "v"(off), "s"(pher_b), "s"(heur_b) : "memory");
        }
        // tabu probe of neighbour k (:145): ends on the key (visited) or on an empty slot (not visited; where the key would go)
        uint32_t hs = ((uint32_t)cur * 2654435761u + hk) >> T.shift;
        int32_t tv = tab[hs];
        bool alive = true, cut = false;
        int32_t em = 63;   // the straggler check runs when (node count & em) == 0: at block boundaries, every 16 nodes once shorter ants have arrived
        while (len <= spill_at && len < (int32_t)D.path_cap) {
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(p), "+v"(h));                 // this step's records; the touch loads stay in flight
            const int32_t key = cur + dk;
            while (tv != key && tv != WA_HASH_EMPTY) { hs = (hs + 1) & T.mask; tv = tab[hs]; }   // (rare: the slot held another key)
            const bool adm = lane < 26 && (__float_as_uint(p) >> 31) == 0 && tv != key;   // sign bit: out of bounds or occupied (:148)
            const float a = adm ? fabsf(p) * h : 0.f;                              // :154 (alpha == 1)
            const unsigned long long mb = __ballot(adm);
            if (mb == 0) { L = INFINITY; alive = false; break; }                   // :162-166
            float t = 0.f + a, c = 0.f + a;
#pragma unroll
            for (int i = 0; i < 25; i++) {
                t = dpp_wave_from_below(t) + a;
                c = dpp_wave_from_above(c) + a;
            }
            const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 25));
            float rnd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ublock), (int)(step & 63u)));
            rnd *= total;                                                          // :170
            const unsigned long long hit = __ballot(adm && c >= rnd);             // first hit in descending edge order (:178)
            if (hit == 0) { L = INFINITY; alive = false; break; }                  // :191-192
            const int pick = 63 - __clzll((long long)hit);
            const int32_t word = (int32_t)((uint32_t)cur + (uint32_t)__builtin_amdgcn_readlane((int)dkw, pick));
            const int32_t next = word & WaNbT<26>::IDM;
            const int32_t slot_pick = __builtin_amdgcn_readlane((int)hs, pick);   // where the picked neighbour's probe ended: empty
            // the NEXT step's records and tabu probe first: their latency runs under the bookkeeping below
            {
                const uint32_t off = (uint32_t)next * 104u + lane_off;
                asm volatile("global_load_dword %0, %2, %3\n global_load_dword %1, %2, %4" : "=&v"(p), "=&v"(h) : "v"(off), "s"(pher_b), "s"(heur_b) : "memory");
            }
            if (lane == 0) tab[slot_pick] = next;                                  // addNextNode :75 (before the probe below: LDS is in order)
            hs = ((uint32_t)next * 2654435761u + hk) >> T.shift;
            tv = tab[hs];
            {   // ... then the touches for the step after that (both ends of every neighbour's two records)
                int32_t v2 = next + dk;
                v2 = v2 < 0 ? 0 : v2 > last_vox ? last_vox : v2;
                const uint32_t off = (uint32_t)v2 * 104u;
                asm volatile("global_load_dword v250, %0, %1\n global_load_dword v251, %0, %1 offset:100\n"
                             "global_load_dword v252, %0, %2\n global_load_dword v253, %0, %2 offset:100"
                             : : "v"(off), "s"(pher_b), "s"(heur_b) : "memory", "v250", "v251", "v252", "v253");
            }
            pbuf = wa_writelane(pbuf, word, len & 63);                             // :76-77
            len++;
            if ((len & 63) == 0) path[len - 64 + lane] = pbuf;
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dlen), pick));   // :78
            step++;
            if ((step & 63u) == 0) ublock = (float)wa_ctr_draw(antkey, step + (uint32_t)lane) / 2147483648.0f;
            cur = next;
            if (next == end) { alive = false; break; }
            if (cutting && (len & em) == 0) {   // arrivals of this generation with a smaller L than this ant has already
                const uint32_t mine = __float_as_uint(L);
                uint32_t e0 = __hip_atomic_load(&sg.arr_len[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t e1 = __hip_atomic_load(&sg.arr_len[lane + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t e2 = __hip_atomic_load(&sg.arr_len[lane + 128], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t e3 = __hip_atomic_load(&sg.arr_len[lane + 192], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int32_t shorter = __popcll(__ballot(e0 < mine)) + __popcll(__ballot(e1 < mine)) + __popcll(__ballot(e2 < mine)) + __popcll(__ballot(e3 < mine));
                if (shorter > 0) em = 15;
                if (shorter >= cut_n) { cut = true; break; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory", "v250", "v251", "v252", "v253");   // the last touches land before anything else runs
        if (lane < (len & 63)) path[(len & ~63) + lane] = pbuf;                    // the partial last block
        if (!alive) {
            finish(L, len);
            return;
        }
        if (cut) {
            // a straggler: its path so far goes to a pool entry of its generation; agents[] says "not arrived, len nodes" (what the ranking
            // sees); a resume block of the next walk launch finishes it.  Pool full: the ant walks on in the loop below, without the check
            int32_t r = 0;
            if (lane == 0) r = atomicAdd(&sg.pool_n[gen & 1], 1);
            r = __builtin_amdgcn_readfirstlane(r);
            if (r < WA_RESUME_MAX) {
                int32_t *pp = sg.pool_path + ((int64_t)(gen & 1) * WA_RESUME_MAX + r) * D.path_cap;
                for (int32_t q0 = 0; q0 < len; q0 += 512) {   // (through L2: the last block was stored by this very wavefront a moment ago)
                    int32_t w[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int32_t q = q0 + u * 64 + lane;
                        w[u] = q < len ? __hip_atomic_load(&path[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int32_t q = q0 + u * 64 + lane;
                        if (q < len) pp[q] = w[u];
                    }
                }
                if (lane == 0) {
                    int32_t *rec = sg.pool_rec + ((gen & 1) * WA_RESUME_MAX + r) * WA_POOL_REC;
                    rec[0] = ant; rec[1] = len; rec[2] = __float_as_int(L);
                    D.antL[(int64_t)slot * D.max_colony + ant] = INFINITY;
                    D.antLen[(int64_t)slot * D.max_colony + ant] = len;
                    if (D.dbg) atomicAdd(&D.dbg[9], 1ULL);
                    atomicAdd(&D.strag_cnt[slot * 2], 1ULL);
                }
                return;
            }
            if (lane == 0) atomicSub(&sg.pool_n[gen & 1], 1);
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();   // the hash is nearly full (the loop below moves the set to the bitmap: it reads path[] back) or path[]
    }                                      // is: the generic loop goes on from here and decides exactly as it always did
    for (;;) {
        if (!T.spilled && len > spill_at) {  // hash nearly full: move the set to the bitmap
            __threadfence();
            for (int i = lane; i < len; i += 64) {
                int32_t id = __hip_atomic_load(&path[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & WaNbT<26>::IDM;
                uint32_t old = atomicOr(&T.bits[(uint32_t)id >> 5], 1u << (id & 31));
                asm volatile("" ::"v"(old));
            }
            __threadfence();
            T.spilled = true;
            if (lane == 0) atomicOr(flags_out, WA_FLAG_BITMAP_USED);
        }
        float p = -0.f, h = 0.f;
        bool adm = false;
        if (lane < 26) {
            p = pher[(int64_t)cur * 26 + lane];
            h = heur[(int64_t)cur * 26 + lane];
        }
        asm volatile("" ::"v"(w0), "v"(w1), "v"(w2), "v"(w3));   // last step's cache-warming loads retire before these
        if (lane < 26 && (__float_as_uint(p) >> 31) == 0) adm = !tabu_has(T, cur + dk);   // sign bit: out of bounds or occupied
        const float info = wa_powi(fabsf(p), R.alpha) * h;                        // :154
        const unsigned long long mb = __ballot(adm);
        if (mb == 0) { L = INFINITY; break; }                                     // :162-166
        // the two ORDERED sums of selectNext as whole-wave DPP chains over the zero-padded candidates:
        // t: lane i <- lane i-1, after 25 steps lane 25 holds (((0+a0)+a1)+...)+a25            (:155)
        // c: lane i <- lane i+1, after 25 steps lane i holds prob_sum once candidates 25..i are in (:172-177)
        const float a = adm ? info : 0.f;
        float t = 0.f + a, c = 0.f + a;
#pragma unroll
        for (int i = 0; i < 25; i++) {
            t = dpp_wave_from_below(t) + a;
            c = dpp_wave_from_above(c) + a;
        }
        const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 25));
        int32_t r;
        if (MODE == 1) r = (int32_t)wa_ctr_draw(antkey, step);
        else r = wa_glibc_next_lanes(rng_rs, rng_f, rng_b);
        float rnd = (float)r / 2147483648.0f;                                     // :169
        rnd *= total;
        const unsigned long long hit = __ballot(adm && c >= rnd);                 // first hit in descending edge order
        const int pick = hit ? 63 - __clzll((long long)hit) : -1;
        if (pick < 0) { L = INFINITY; break; }                                    // :191-192
        int px, py, pz;
        wa_off26(pick, px, py, pz);
        const int32_t next = cur + pz * D.d.nxy + py * D.d.nx + px;
        if (len >= D.path_cap) {
            if (lane == 0) atomicOr(flags_out, WA_FLAG_PATH_OVERFLOW);
            L = INFINITY;
            break;
        }
        if (lane == 0) {
            path[len] = next | (pick << WaNbT<26>::SHIFT);
            tabu_insert(T, next);
        }
        __builtin_amdgcn_wave_barrier();
        len++;
        const int type = (px != 0) + (py != 0) + (pz != 0);
        L += type == 1 ? d1 : type == 2 ? d2 : d3;                                // :78
        step++;
        if (next == end) break;
        cur = next;
        {   // the records the NEXT step may need are those of cur's 26 neighbours: lane k touches both ends of
            // neighbour k's 104-byte pheromone and heuristic records so that step's loads hit in cache
            int64_t rec = ((int64_t)cur + dk) * 26;
            rec = rec < 0 ? 0 : rec > last_rec ? last_rec : rec;
            w0 = pher[rec]; w1 = pher[rec + 25];
            w2 = heur[rec]; w3 = heur[rec + 25];
        }
    }
    asm volatile("" ::"v"(w0), "v"(w1), "v"(w2), "v"(w3));
    if (T.spilled) {  // leave the bitmap all-zero for the next walk
        __threadfence();
        for (int i = lane; i < len; i += 64) {
            int32_t id = __hip_atomic_load(&path[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & WaNbT<26>::IDM;
            __hip_atomic_store(&T.bits[(uint32_t)id >> 5], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
    }
    finish(L, len);
}

// walk_flags bit 5: this generation may hand its stragglers over (the next launch of the call carries resume blocks, see k_walk_dev)
__global__ __launch_bounds__(64) void k_walk_dev26(WaAcsDev D, WaRun R, int hash_log2, int32_t gen, int32_t walk_flags)
{
    extern __shared__ int32_t lds[];
    const int32_t slot = blockIdx.y, ant = blockIdx.x;
    const WaSlotCtl *c = &D.ctl[slot];
    const int32_t colony = c->colony[gen & 1];
    int32_t f = 0, b = 0, rs_unused = 0;
    if (D.pool_n && (int32_t)blockIdx.x >= D.max_colony) {
        // ---- resume block: a straggler of generation gen - 1 finishes its walk here, on that generation's field
        const int32_t r = (int32_t)blockIdx.x - D.max_colony, pg = (gen - 1) & 1;
        const WaStrag sg = wa_strag_of(D, slot);
        if (gen < 1 || r >= sg.pool_n[pg] || r >= WA_RESUME_MAX) return;
        const int32_t *rec = sg.pool_rec + (pg * WA_RESUME_MAX + r) * WA_POOL_REC;
        const int32_t a = rec[0], n0 = rec[1];
        const float L0 = __int_as_float(rec[2]);
        WaAcsDev Dp = D;
        Dp.pher = const_cast<float *>(D.prev_pher);
        const uint64_t key = wa_ctr_antkey(wa_ctr_key(R.seed, c->stream, (uint32_t)(gen - 1)), (uint32_t)a);
        wa_walk_one26<1>(Dp, R, slot, a, c->start, c->end, key, lds, hash_log2, rs_unused, f, b, &D.ctl[slot].flags, 0, 0x7fffffff,
                         sg.pool_path + ((int64_t)pg * WA_RESUME_MAX + r) * D.path_cap, n0, L0, gen - 1, D.max_colony + r, (walk_flags & 64) != 0);
        return;
    }
    if (walk_flags & 64) return;   // drain launch: resume blocks only
    if (ant >= colony || colony > D.max_colony) return;
    const uint64_t antkey = wa_ctr_antkey(wa_ctr_key(R.seed, c->stream, (uint32_t)gen), (uint32_t)ant);
    const int32_t rlen = (D.rtab && c->bestL != INFINITY) ? c->best_len : 0;
    // an ant with a larger L than floor(lambda - 1) + 1 arrivals cannot be among the depositing ranks (:200) nor be the iteration's best
    int32_t cut_n = 0x7fffffff;
    if ((walk_flags & 32) && D.pool_n && R.alpha == 1) cut_n = (int32_t)(c->lambda[gen & 1] - 1.f) + 1;
    if (cut_n < 1) cut_n = 1;
    wa_walk_one26<1>(D, R, slot, ant, c->start, c->end, antkey, lds, hash_log2, rs_unused, f, b, &D.ctl[slot].flags, rlen, cut_n, nullptr, 0, 0.f, gen);
}

__global__ __launch_bounds__(64) void k_walk_ref26(WaAcsDev D, WaRun R, int hash_log2, int32_t gen)
{
    extern __shared__ int32_t lds[];
    const int32_t slot = blockIdx.y;
    const WaSlotCtl *c = &D.ctl[slot];
    const int32_t colony = c->colony[gen & 1];
    if (colony > D.max_colony) return;
    int32_t r = threadIdx.x < 31 ? D.rng->r[threadIdx.x] : 0;   // lane j holds word j of the state (see wa_glibc_next_lanes)
    int32_t f = D.rng->f, b = D.rng->b;
    const int32_t start = c->start, end = c->end;
    for (int32_t ant = 0; ant < colony; ant++)
        wa_walk_one26<0>(D, R, slot, ant, start, end, 0, lds, hash_log2, r, f, b, &D.ctl[slot].flags, 0);
    if (threadIdx.x < 31) D.rng->r[threadIdx.x] = r;
    if (threadIdx.x == 0) {
        D.rng->f = f;
        D.rng->b = b;
    }
}
