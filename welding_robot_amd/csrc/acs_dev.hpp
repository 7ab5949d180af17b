// acs_dev.hpp -- the solver's device block (WaAcsDev), rank masks, straggler views, edge offsets and the per-problem kernels
// (k_init_pheromone<NB>, k_heuristic<NB>, k_begin).  Part of acs_kernels.hpp (included from there, in this order).
#pragma once
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
    int32_t *paths;                // [slot][max_colony][path_cap]  (solvers that hand stragglers over keep TWO such arrays and alternate by generation: see prev_paths)
    float *antL;                   // [slot][max_colony]
    int32_t *antLen;               // [slot][max_colony]
    int32_t *antRep;               // [slot][max_colony]   1: the ant of the generation walked last ARRIVED ON THE REPLAY TRACK, i.e. its path is the best path it replayed, word for word (k_walk_dev; 0 from every other walk kernel): ranked ants of that kind are marked together (k_evap_rank_mark)
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
    int32_t *prev_paths;           // [slot][max_colony][path_cap]  the paths of the PREVIOUS generation's ants: a straggler's walk so far stays where it is (no copy) and its resume block walks on in place, while the running generation's ants write `paths`
    // REF mode, converged colonies (k_ref_draws / k_walk_ref_spec / k_walk_ref): the libc stream generated ahead for a whole generation under
    // the assumption that every ant follows the best path, so that the ants can check that assumption IN PARALLEL
    int32_t *ref_draws;            // [max_colony * WA_REF_SPEC_LEN] the next colony * (best_len - 1) outputs of the stream, in order
    int32_t *ref_state;            // [blocks + 1][32] rotated state (lane j = r[(f + j) % 31]) in front of stream output 64 * block
    const uint32_t *ref_jump;      // [31][31] the stream's state advanced by WA_REF_SUPER outputs as a matrix over Z/2^32 (companion matrix ^ WA_REF_SUPER, host_acs.inc)
    int32_t *ref_ok;               // [max_colony + 2] per ant: followed the whole best path with the draws it was dealt; [max_colony] = speculation active, [max_colony + 1] = steps per ant
    unsigned long long *strag_cnt; // [slot][2]  ants handed over / stragglers finished by a resume block, per slot (wa_acs_straggler_counters)
    const float *prev_pher;        // the field of the previous generation (intact until the next sweep): what a resume block walks on
    float *ltab;                   // [path_cap + 1] L after i steps = precision added i times in fp32 (:78), one table per solver
    uint32_t tab16_kmul;           // 16-bit tabu entries (WaTabu): K_B << (32 - B) for this grid's B-bit voxel ids, 0 where an entry cannot name them (6 neighbours only)
    int32_t guard_bytes;           // guard band in front of / behind the pheromone and heuristic allocations (6-neighbour solvers)
    int32_t stamp_guard_bytes;     // ... and the stamp allocation of a lazily evaporating solver
    int32_t vbits_rows;            // bitmap rows per slot: max_colony (+ WA_RESUME_MAX rows of the resume blocks when the solver has straggler pools)
};

#define WA_REF_SPEC_LEN 4096       // longest best path (in steps) for which a REF generation is speculated
#define WA_REF_SUPER 1024          // outputs per wavefront of k_ref_draws_fill (a multiple of 64)
#define WA_RESUME_MAX 256
#define WA_POOL_REC 4

// the arrival list and the straggler pools of ONE slot (every search of a launch hands its own stragglers over)
struct WaStrag {
    uint32_t *arr_len, *arr_n;
    int32_t *pool_n, *pool_rec;
};
__device__ __forceinline__ WaStrag wa_strag_of(const WaAcsDev &D, int32_t slot)
{
    WaStrag g;
    g.arr_len = D.arr_len + (int64_t)slot * 256;
    g.arr_n = D.arr_n + slot;
    g.pool_n = D.pool_n + (int64_t)slot * 2;
    g.pool_rec = D.pool_rec + (int64_t)slot * 2 * WA_RESUME_MAX * WA_POOL_REC;
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
// several rank bits at once (ranks whose ants walked the same path: k_evap_rank_mark)
__device__ __forceinline__ void wa_mask_or_bits(const WaMaskRef &m, int64_t e, unsigned long long bits)
{
    if (m.w) atomicOr(&m.w[e], bits);
    else atomicOr(reinterpret_cast<unsigned int *>(m.b + (e & ~(int64_t)3)), ((unsigned int)bits & 0xffu) << (8 * (int)(e & 3)));
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

// ------------------------------------------------------------------ per-voxel field writers (pheromone init / reset, heuristic)
// One thread per VOXEL: its coordinates cost two 32-bit divisions ONCE, what the NB edges share (the vector to the end point and its norm) is computed
// once, and the NB values of a block's 256 voxels go through LDS so that the block writes 256 x NB contiguous floats.  Until round 6 these were
// thread-per-EDGE kernels that spent their time on 64-bit index divisions (k_heuristic: 0.52 ms per 256^3 field = 0.78 TB/s, 63 fields = 33 ms of
// BASELINE config C5's 0.40 s; k_init_pheromone: 1.3 TB/s).  Same arithmetic per value, in the same order: the fields are bit-identical.
template <int NB>
__device__ __forceinline__ void wa_write_voxel_block(float *field, int64_t n_vox, const float (&v)[NB], bool live)
{
    __shared__ float s_v[256 * NB];
    const int tid = threadIdx.x;
    if (live) {
#pragma unroll
        for (int k = 0; k < NB; k++) s_v[tid * NB + k] = v[k];
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * 256 * NB, end = n_vox * NB;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const int64_t t = base + j * 256 + tid;
        if (t < end) field[t] = s_v[j * 256 + tid];
    }
}

// mode 0: initFromGridMap (out-of-bounds edges 0), mode 1: reset() (every edge pheromone_0).
// The sign bit is set on edges whose neighbour is out of bounds or occupied.  grid = (ceil(N / 256), slots), one definition for both neighbourhoods.
template <int NB>
__global__ __launch_bounds__(256) void k_init_pheromone(WaAcsDev D, int32_t slot0, float p0, int32_t mode)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const bool live = (int64_t)id < D.d.n;
    const int32_t slot = slot0 + blockIdx.y;
    float v[NB];
    if (live) {
        const uint32_t nx = (uint32_t)D.d.nx, ny = (uint32_t)D.d.ny, row = id / nx;
        const int32_t x = (int32_t)(id - row * nx), z = (int32_t)(row / ny), y = (int32_t)(row - (uint32_t)z * ny);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            int dx, dy, dz;
            wa_edge_offset<NB>(k, dx, dy, dz);
            const int32_t X = x + dx, Y = y + dy, Z = z + dz;
            const bool inb = X >= 0 && X < D.d.nx && Y >= 0 && Y < D.d.ny && Z >= 0 && Z < D.d.nz;
            const bool adm = inb && D.occ[(int64_t)id + dz * D.d.nxy + dy * D.d.nx + dx] != 0;
            const float val = (inb || mode == 1) ? p0 : 0.f;
            v[k] = adm ? val : -val;
        }
    }
    wa_write_voxel_block<NB>(D.pher + (int64_t)slot * D.pher_stride, D.d.n, v, live);
}

// ------------------------------------------------------------------ heuristic field
// (1 + beta*cos) of :151-154 is a function of the voxel, the edge and the END point only: the fields live in a pool,
// wa_acs_begin computes one per distinct end point of its batch that the pool does not hold yet (`fields` / `ends` = pool
// index and end point of each field to compute) and every search reads the field ctl.heur_slot names.  grid = (ceil(N / 256), fields)
template <int NB>
__global__ __launch_bounds__(256) void k_heuristic(WaAcsDev D, float beta, const int32_t *fields, const int32_t *ends)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const bool live = (int64_t)id < D.d.n;
    const int32_t slot = fields[blockIdx.y];
    const int32_t end = ends[blockIdx.y];
    float v[NB];
    if (live) {
        const uint32_t nx = (uint32_t)D.d.nx, ny = (uint32_t)D.d.ny, row = id / nx;
        const int32_t x = (int32_t)(id - row * nx), z = (int32_t)(row / ny), y = (int32_t)(row - (uint32_t)z * ny);
        const int32_t ex = end % D.d.nx, ey = (end / D.d.nx) % D.d.ny, ez = end / D.d.nxy;
        const float ax = D.cx[ex] - D.cx[x], ay = D.cy[ey] - D.cy[y], az = D.cz[ez] - D.cz[z];  // :137
        const float na = sqrtf(ax * ax + ay * ay + az * az);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            int dx, dy, dz;
            wa_edge_offset<NB>(k, dx, dy, dz);
            const int32_t X = x + dx, Y = y + dy, Z = z + dz;
            float out = 0.f;
            if (X >= 0 && X < D.d.nx && Y >= 0 && Y < D.d.ny && Z >= 0 && Z < D.d.nz) {
                const float bx = D.cx[X] - D.cx[x], by = D.cy[Y] - D.cy[y], bz = D.cz[Z] - D.cz[z];     // :151
                const float dot = ax * bx + ay * by + az * bz;
                const float nb = sqrtf(bx * bx + by * by + bz * bz);
                out = 1 + beta * (dot / (na * nb));   // :152-154 (0/0 = NaN on a duplicated seam coordinate, Q3)
            }
            v[k] = out;
        }
    }
    wa_write_voxel_block<NB>(D.heur + (int64_t)slot * D.pher_stride, D.d.n, v, live);
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
