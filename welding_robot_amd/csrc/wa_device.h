// wa_device.h -- shared host/device definitions of libweldacs (gfx950).
//
// Data layout in HBM (per solver, per problem slot) -- see DESIGN.md:
//   pher  float[6N]  pheromone of directed edge (voxel, k); the SIGN BIT carries the static
//                    admissibility of the edge (set = neighbour out of bounds or occupied), so
//                    one 24-byte load per step answers both ACSRank_3D.hpp:148 and :154.
//                    |value| is bit-identical to the reference's adjacency_infos[k].pheromone.
//   heur  float[6N]  (1 + beta*cos) of ACSRank_3D.hpp:151-154 for the slot's end node
//   mask  u64[6N]    deposit rank masks (zero between generations)
//   paths int32[max_colony][path_cap]  node id | (edge k taken to arrive << 29)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WA_K_SHIFT 29
#define WA_ID_MASK 0x1FFFFFFFu
#define WA_HASH_EMPTY (-1)
#define WA_HASH_SENTINEL (-2)   // the entry behind the table (see wa_tabu_clear): never empty, never a voxel id
#define WA_MAX_TRACKED_ERR 1

enum { WA_FLAG_PATH_OVERFLOW = 1, WA_FLAG_COLONY_OVERFLOW = 2, WA_FLAG_BITMAP_USED = 4 };

struct WaDims {
    int32_t nx, ny, nz;
    int32_t nxy;      // nx*ny
    int64_t n;        // nx*ny*nz
};

// run-constant parameters (ACSRank_3D.hpp:319-326 + the call-site predict)
struct WaRun {
    int32_t alpha;
    float beta, rho, pheromone_0, predict, precision;
    int32_t fixed_colony;
    int32_t rng_mode;
    uint64_t seed;
};

// per-slot control block, lives in device memory, updated by the kernels only
struct WaSlotCtl {
    int32_t start, end;
    uint32_t stream;
    int32_t gen;            // generations completed (informational: kernels get `gen` as an argument)
    // parameters of generation g live in slot [g & 1]: the block that publishes generation g+1's
    // values writes the other slot while its sibling blocks still read generation g's
    int32_t colony[2];      // ants                               (:247)
    float lambda[2], Q[2];  //                                    (:248-249)
    float bestL;            // best.L                            (:232,:263)
    int32_t best_len;
    uint32_t best_ver;      // bestmark[v] == best_ver <=> v on best path
    // frozen for the deposit of the generation just walked
    float dep_lambda, dep_Q, dep_bestL;
    int32_t n_dep;          // ranks 1..n_dep deposit             (:200)
    int32_t flags;
    // lazy evaporation (wa_acs_create_lazy): the value every never-deposited in-bounds edge holds after the
    // evaporations so far, pheromone_0 * rho * rho * ... in the reference's own fp32 rounding; slot [g & 1] like the others
    float clean[2];
    uint32_t evap_base;     // evaporations applied to the field before generation 0 of the current solve (carried across solves)
    int32_t tabu_gen;       // generation in which the best path last changed: the replay-table rows of that generation rebuild besttabu[]
    int32_t heur_slot;      // whose heuristic field this search reads: searches of one wa_acs_begin with the same end point share one
    int32_t pad_;
    // bit o-1: rank o of the generation just ranked belongs to an ant that arrived on the replay track (its path is the best path, word for word).
    // The post-walk launch marks all of them through the lowest one; the apply pass skips the others (never an edge's lowest rank).  0 from k_rank
    unsigned long long rep_mask;
};

// glibc TYPE_3 state as the kernels keep it: r[0..30], f index, b index
struct WaGlibcRand {
    int32_t r[34];
    int32_t f, b;
};

__host__ __device__ inline uint64_t wa_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// DEV-mode draw: a pure function of (seed, stream, generation, ant, step) so that every ant of
// every problem walks in parallel.  64-bit mixing once per generation and once per ant; the
// per-step draw is a 32-bit avalanche hash (two multiplies) because it sits in the walk's
// issue-bound inner loop.  Identical to wo_ctr_* in oracle/weld_oracle.c.
__host__ __device__ inline uint64_t wa_ctr_key(uint64_t seed, uint32_t stream, uint32_t gen)
{
    return wa_mix64(seed + 0x9E3779B97F4A7C15ULL * (((uint64_t)stream << 32) | gen));
}
__host__ __device__ inline uint64_t wa_ctr_antkey(uint64_t key, uint32_t ant)
{
    return wa_mix64(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)ant + 1));
}
__host__ __device__ inline uint32_t wa_ctr_draw(uint64_t antkey, uint32_t step)
{
    uint32_t x = ((uint32_t)antkey + step * 0x9E3779B9u) ^ (uint32_t)(antkey >> 32);
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x >> 1;  // 31 bits, like rand()
}

// glibc random_r TYPE_3 (stdlib/random_r.c): the stream behind the reference's rand()
__host__ __device__ inline int32_t wa_glibc_next(int32_t *r, int32_t &f, int32_t &b)
{
    uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
    r[f] = (int32_t)v;
    if (++f >= 31) { f = 0; ++b; }
    else if (++b >= 31) b = 0;
    return (int32_t)((v >> 1) & 0x7fffffffu);
}
#if defined(__HIPCC__)
// the same generator with the state spread over the lanes of a wavefront: lane j < 31 holds r[j], f and b are wave-uniform.
// r[f] += r[b]; result = r[f] >> 1 (random_r.c) as two v_readlane, one add and one v_writelane
__device__ __forceinline__ int32_t wa_glibc_next_lanes(int32_t &rs, int32_t &f, int32_t &b)
{
    f = __builtin_amdgcn_readfirstlane(f);   // (uniform by construction; this tells the compiler)
    b = __builtin_amdgcn_readfirstlane(b);
    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(rs, f) + (uint32_t)__builtin_amdgcn_readlane(rs, b);
    asm volatile("s_mov_b32 m0, %2\n s_nop 3\n v_writelane_b32 %0, %1, m0" : "+v"(rs) : "s"((int32_t)v), "s"(f) : "m0");
    if (++f >= 31) { f = 0; ++b; }
    else if (++b >= 31) b = 0;
    return (int32_t)((v >> 1) & 0x7fffffffu);
}
#endif
__host__ inline void wa_glibc_seed(WaGlibcRand *s, uint32_t seed)
{
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    s->r[0] = word;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        s->r[i] = word;
    }
    s->f = 3;
    s->b = 0;
    for (int i = 0; i < 310; i++) (void)wa_glibc_next(s->r, s->f, s->b);
}

// power() of ACSRank_3D.hpp:48-60
template <class T>
__host__ __device__ inline T wa_powi(T x, int y)
{
    T ans = 1;
    while (y) {
        if (y & 1) ans *= x;
        x *= x;
        y >>= 1;
    }
    return ans;
}
