// acs_update.hpp -- everything behind the walk (ACSRank_3D.hpp:263-280): libstdc++ sort order, k_rank, the evaporation sweep, the fused
// post-walk launch (sweep + rank + mark), ranked deposit, lazy-evaporation helpers.  Part of acs_kernels.hpp (included from there, in this order).
#pragma once
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
#ifndef WA_RANK_LDS
#define WA_RANK_LDS 2048
#endif
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
            // (one thread restates std::sort: its records live in LDS -- the DEV branch's key array, unused here -- when the colony fits:
            //  a record access is then ~100 cycles instead of a global round trip; 263 -> ~60 us per generation at 256 ants)
            WaRec *rec = in_lds ? reinterpret_cast<WaRec *>(s_keys) : (WaRec *)(D.sortk + (int64_t)slot * D.max_colony * 2);
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
        c.rep_mask = 0;   // (the plain loop marks and applies rank by rank)
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
#define WA_LEN_OF(x) ((x) & 0x3fffffff)
template <bool SPARSE, int NB>
__global__ __launch_bounds__(256) void k_evap_rank_mark(WaAcsDev D, WaRun R, const float *src_base,
                                                        float *dst_base, int32_t E, int32_t gen, int32_t MB, int32_t split_log2, int32_t lazy_period,
                                                        int32_t sweep_nt, int32_t rank_cap)
{
    // rank_cap: ants the rank arrays in (dynamic) LDS hold, 16 bytes each: the host passes WA_RANK_LDS for dense solvers -- the launch as it always was --
    // and the solver's max_colony (rounded up to 64) for lazily evaporating ones, whose post-walk launch is all latency chains: a 33-KB block of a 256-ant
    // colony kept four blocks per CU resident, a 4-KB one keeps eight (32 lazy 256-ant searches: 253 -> 283 k problem-generations/s, round 6)
    extern __shared__ unsigned long long wa_rank_lds[];
    const int32_t slot = blockIdx.y, tid = threadIdx.x;
    // the MB rank/mark blocks come FIRST in the grid so that they are dispatched immediately and
    // their latency-bound work hides under the sweep blocks that follow
    if ((int32_t)blockIdx.x >= MB) {  // ---- sweep: dst = src * rho (same body as k_evaporate)
#ifdef WA_TEST_KNOBS
        if (sweep_nt & 0x100) return;   // timing aid (tools/fused_parts.py): the launch without its sweep
#endif
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
                for (int k = 0; k < NB; k++) ph[(int64_t)v * NB + k] = wa_catch_up(ph[(int64_t)v * NB + k], target - old, rho);
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
    unsigned long long *s_keys = wa_rank_lds;
    int32_t *s_perm = reinterpret_cast<int32_t *>(wa_rank_lds + rank_cap), *s_len = s_perm + rank_cap;
    __shared__ int32_t s_ndep, s_fin;
    __shared__ unsigned long long s_steps;
    if (tid == 0) { s_ndep = 0; s_fin = 0; s_steps = 0; }
    // Block 0 compares the iteration's best with the global best and then REPLACES it (its thread 0, behind the second barrier below).  Every thread
    // takes its copy of the old best HERE, in front of the barriers: read behind them, a wavefront that is scheduled late found the value thread 0
    // had just published, took the new best for no improvement and skipped its quarter of the path copy -- 64 stale words in bestpath[], once in
    // ~3 000 searches of a saturated batch when the launch keeps eight blocks per CU resident (round 6: tools/state_hash.py, profiles/r06/best_copy_race.txt)
    float bestL = INFINITY;
    uint32_t ver = 0;
    int32_t blen = 0;
    if (mb == 0) { bestL = ctl->bestL; ver = ctl->best_ver; blen = ctl->best_len; }
    // This block is a chain of dependent global loads beside a sweep that saturates the memory system (every level costs 2-3 us there):
    // the ants' results are requested for ALL max_colony ants before the control block says how many there are (inside the allocation;
    // entries beyond the colony are never looked at), and every ant's length goes to LDS with its key, so that the ranked ant's length is
    // an LDS read: control block + results -> path words -> marks, three levels instead of five.
    const int32_t cmax = D.max_colony < rank_cap ? D.max_colony : rank_cap;
    for (int32_t a = tid; a < cmax; a += blockDim.x) {
        const float La = antL[a];
        const int32_t na = antLen[a];
        const int32_t rep = D.antRep[(int64_t)slot * D.max_colony + a];
        s_keys[a] = ((unsigned long long)__float_as_uint(La) << 32) | (uint32_t)a;
        s_len[a] = na | ((rep & 1) << 30);        // (bit 30: the ant arrived on the replay track -- its path is the best path it replayed; WA_LEN_OF strips it)
    }
    if (colony > D.max_colony || colony > rank_cap) {
        if (mb == 0 && tid == 0) atomicOr(&ctl->flags, WA_FLAG_COLONY_OVERFLOW);
        return;
    }
    __syncthreads();
    int32_t myfin = 0;
    unsigned long long mysteps = 0;
    if (mb == 0)
        for (int32_t a = tid; a < colony; a += blockDim.x) {
            myfin += (__uint_as_float((uint32_t)(s_keys[a] >> 32)) != INFINITY) ? 1 : 0;
            mysteps += (unsigned long long)(WA_LEN_OF(s_len[a]) - 1);
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
#ifdef WA_TEST_KNOBS
        // test knob (tests/test_gpu_late_waves.py): wavefronts 1..3 of the publishing block fall ~20 us behind wavefront 0 right here
        if ((sweep_nt & 0x400) && (tid >> 6) != 0) { for (int z = 0; z < 8; z++) __builtin_amdgcn_s_sleep(127); asm volatile("" ::: "memory"); }   // (8 x 127 x 64 clocks; no memory access moves across)
#endif
#ifdef WA_BEST_READ_LATE   // (diagnostic build: the reads where they were before the fix, for the negative half of tests/test_gpu_late_waves.py)
        // (read past the CU's vector cache, where a quiet GPU would still hold the old line: in a saturated launch that line has long been evicted)
        bestL = __hip_atomic_load(&ctl->bestL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ver = __hip_atomic_load(&ctl->best_ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        blen = __hip_atomic_load(&ctl->best_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        bool changed = false;
        if (iterAnt >= 0 && iterL < bestL) {
            blen = WA_LEN_OF(s_len[iterAnt]);
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
    // ---- mark: OR bit (o-1) into the rank mask of every directed edge of ranked ant o.
    // Ranked ants that ARRIVED ON THE REPLAY TRACK walked the same path word for word (the best path as it stood during the walk): once a colony has
    // converged that is nearly all of them, and marking each on its own puts n_dep atomics -- and, lazily evaporated, n_dep stamp exchanges -- on every
    // edge of that one path (round 6: the lazy post-walk launch of a 32-search batch of 256-ant colonies was 46 % of the batch's kernel time).  The
    // blocks of the LOWEST such rank mark for all of them (one OR of all their bits, one claim per voxel); the blocks of the others have nothing to do.
    // Same masks, same stamps: the apply pass cannot tell.
    __shared__ unsigned long long s_rep;
    if (tid < 64) {
        const bool r = tid < n_dep && ((s_len[s_perm[tid]] >> 30) & 1);
        const unsigned long long G_ = __ballot(r);
        if (tid == 0) s_rep = G_;
    }
    __syncthreads();
    const unsigned long long G = s_rep;
    if (mb == 0 && tid == 0) ctl->rep_mask = G;   // (for the apply pass: the blocks of the other replaying ranks have nothing to do there either)
    const int32_t bit = mb >> split_log2, bx = mb & ((1 << split_log2) - 1), o = bit + 1;
    if (o > n_dep) return;
    unsigned long long bits = 1ULL << bit;
    if ((G >> bit) & 1ULL) {
        if (bit != __ffsll((long long)G) - 1) return;
        bits = G;
    }
#ifdef WA_TEST_KNOBS
    if (sweep_nt & 0x200) return;       // timing aid: the launch without the marks (nothing is deposited: the colony keeps exploring)
#endif
    const int32_t a = s_perm[o - 1];
    const int32_t len = WA_LEN_OF(s_len[a]);
    const int32_t *path = D.paths + ((int64_t)slot * D.max_colony + a) * D.path_cap;
    const WaMaskRef mask = wa_mask_of(D, slot);
    float *ph = dst_base + (int64_t)slot * D.pher_stride;
    const float clean_next = ctl->clean[gen & 1] * R.rho;   // == what block 0 publishes into clean[(gen+1)&1]
    for (int32_t i = 1 + bx * blockDim.x + tid; i < len; i += (blockDim.x << split_log2)) {
        int32_t w = path[i];
        int32_t v = path[i - 1] & WaNbT<NB>::IDM;
        int64_t e = (int64_t)v * NB + ((uint32_t)w >> WaNbT<NB>::SHIFT);
        wa_mask_or_bits(mask, e, bits);
        if (SPARSE) {   // v receives a deposit: its record must be current (after this generation's evaporation) for the apply pass
            uint32_t *stamp = D.stamp + (int64_t)slot * D.d.n;
            const uint32_t target = ctl->evap_base + (uint32_t)gen + 2u;
            const uint32_t old = atomicExch(&stamp[v], target);
            if (old == 0) {            // first deposit ever: v joins the dirty list, its record is written at the clean value
                // (one atomic per WAVEFRONT on the list's cursor, not one per voxel: in a search's first generations nearly every marked voxel is new,
                //  tens of thousands of increments of one address per search and launch)
                const unsigned long long newm = __ballot(1);
                const int lane_ = tid & 63, lead_ = __ffsll((long long)newm) - 1;
                int32_t base_ = 0;
                if (lane_ == lead_) base_ = atomicAdd(&D.dcount[slot * 2 + 1], (int32_t)__popcll(newm));
                base_ = __shfl(base_, lead_, 64);
                const int32_t idx = base_ + (int32_t)__popcll(newm & ((1ULL << lane_) - 1ULL));
                D.dirty_list[(int64_t)slot * D.d.n + idx] = v;
#pragma unroll
                for (int k = 0; k < NB; k++) {   // stored = the init value: 0 stays 0 (out-of-bounds edge of initFromGridMap), p0 became clean_next
                    const float st0 = ph[(int64_t)v * NB + k];
                    ph[(int64_t)v * NB + k] = copysignf(fabsf(st0) == 0.f ? 0.f : clean_next, st0);
                }
            } else if (old != target) {   // deposited before: apply the evaporations it has missed since
#pragma unroll
                for (int k = 0; k < NB; k++) ph[(int64_t)v * NB + k] = wa_catch_up(ph[(int64_t)v * NB + k], target - old, R.rho);
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
    const unsigned long long rep = c->rep_mask;
    if (o > n_dep) return;
    // a rank whose ant replayed the best path, but not the lowest such rank: every edge of its path carries that lower rank's bit too, so it owns none
    if (o <= 64 && ((rep >> (o - 1)) & 1ULL) && (o - 1) != __ffsll((long long)rep) - 1) return;
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
template <int NB>
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
        for (int k = 0; k < NB; k++) {
            int dx, dy, dz;
            wa_edge_offset<NB>(k, dx, dy, dz);
            const int32_t X = x + dx, Y = y + dy, Z = z + dz;
            const bool inb = X >= 0 && X < D.d.nx && Y >= 0 && Y < D.d.ny && Z >= 0 && Z < D.d.nz;
            const bool adm = inb && D.occ[id + dz * D.d.nxy + dy * D.d.nx + dx] != 0;
            const float v = (inb || mode == 1) ? p0 : 0.f;
            ph[(int64_t)id * NB + k] = adm ? v : -v;
        }
        stamp[id] = 0;
    }
}
// rows of `width` words, `pitch` words apart -> packed (wa_acs_result_batch: the best paths of all slots in one host copy).  grid.y = rows.
__global__ __launch_bounds__(256) void k_gather_rows(int32_t *__restrict__ dst, const int32_t *__restrict__ src, int64_t pitch, int64_t width)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) dst[(int64_t)blockIdx.y * width + i] = src[(int64_t)blockIdx.y * pitch + i];
}
// The two init modes (initFromGridMap: out-of-bounds edges 0, ACSRank_3D.hpp:389-403; reset(): every edge pheromone_0, :307-315) differ
// ONLY in the out-of-bounds edges, and those exist on the six faces of the lattice only: a lazy slot that changes mode with p0 unchanged
// (the first reset() behind initFromGridMap -- every pair-planning run) rewrites 2(nx ny + nx nz + ny nz) floats instead of 6 N
// (256^3: 0.4 M instead of 100 M per slot; 224 slots: 70 ms of the first batch's read-back).  Thread per (face voxel): its one edge that
// leaves the lattice through that face, the value k_init_pheromone would store (sign set: never admissible).  grid.y = slots.
__global__ __launch_bounds__(256) void k_lazy_faces(WaAcsDev D, int32_t slot0, float p0, int32_t mode)
{
    const int32_t slot = slot0 + blockIdx.y;
    const int64_t nx = D.d.nx, ny = D.d.ny, nz = D.d.nz, nxy = D.d.nxy;
    const int64_t fz = nx * ny, fy = nx * nz, fx = ny * nz;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * (fz + fy + fx)) return;
    int64_t id;
    int k;
    if (t < 2 * fz) {            // z = 0 (edge 0: z-1) and z = nz-1 (edge 5: z+1)
        const bool hi = t >= fz;
        id = (hi ? t - fz : t) + (hi ? (nz - 1) * nxy : 0);
        k = hi ? 5 : 0;
    } else if ((t -= 2 * fz) < 2 * fy) {   // y = 0 (edge 1) and y = ny-1 (edge 4)
        const bool hi = t >= fy;
        const int64_t q = hi ? t - fy : t;
        id = (q / nx) * nxy + (hi ? (ny - 1) * nx : 0) + q % nx;
        k = hi ? 4 : 1;
    } else {                     // x = 0 (edge 2) and x = nx-1 (edge 3)
        t -= 2 * fy;
        const bool hi = t >= fx;
        const int64_t q = hi ? t - fx : t;
        id = (q / ny) * nxy + (q % ny) * nx + (hi ? nx - 1 : 0);
        k = hi ? 3 : 2;
    }
    const float v = mode == 1 ? p0 : 0.f;
    D.pher[(int64_t)slot * D.pher_stride + id * 6 + k] = -v;
}
// bring every deposited record current (before a solve that evaporates with a different rho: the pending
// multiplications belong to the old one).  grid.y = slots.
template <int NB>
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
        for (int k = 0; k < NB; k++) ph[(int64_t)v * NB + k] = wa_catch_up(ph[(int64_t)v * NB + k], target - old, rho_old);
        stamp[v] = target;
    }
}

// the field as the dense sweep would have left it: deposited records with their pending evaporations applied,
// clean records at the clean value
template <int NB>
__global__ __launch_bounds__(256) void k_lazy_materialise(WaAcsDev D, WaRun R, int32_t slot, float *out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= D.d.n * NB) return;
    const int64_t v = e / NB;
    const float st0 = D.pher[(int64_t)slot * D.pher_stride + e];
    const WaSlotCtl *c = &D.ctl[slot];
    const float clean = c->clean[c->gen & 1];
    const uint32_t evap_now = c->evap_base + (uint32_t)c->gen;
    const uint32_t stv = D.stamp[(int64_t)slot * D.d.n + v];
    out[e] = stv != 0 ? wa_catch_up(fabsf(st0), evap_now + 1u - stv, R.rho) : (fabsf(st0) == 0.f ? 0.f : clean);
}

#ifdef WA_STATE_HASH
// ------------------------------------------------------------------ diagnostic build (-DWA_STATE_HASH, tools/state_hash.py): an order-free digest
// of everything a search's next kernel depends on, taken behind every launch of the generation loop.  Two runs of the same batch must produce the
// same digests whatever the timing; the first (generation, launch, slot, part) that differs names the kernel whose output depends on scheduling.
// Lazily evaporated fields are digested as the VALUES they stand for (pending evaporations applied): which records the background pass
// refreshed in which generation depends on the order the voxels joined the dirty list, the values do not.
// parts: [0] field values, [1] set of dirty voxels, [2] rank masks, [3] the ants' results and paths, [4] best path, [5] control block + dirty count,
//        [6] prefix-tabu bits of the best path, [7] replay table (both rebuilt by the apply + table launch: stale behind the launch in front of it)
__device__ __forceinline__ unsigned long long wa_h2(unsigned long long a, unsigned long long b)
{
    return wa_mix64(a * 0x9E3779B97F4A7C15ULL + wa_mix64(b + 0x632BE59BD9B4E019ULL));
}
__global__ __launch_bounds__(256) void k_state_hash(WaAcsDev D, WaRun R, unsigned long long *out, int32_t phase)
{
    const int32_t slot = blockIdx.y, tid = threadIdx.x;
    const WaSlotCtl *c = &D.ctl[slot];
    const int32_t g = c->gen;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float *ph = D.pher + (int64_t)slot * D.pher_stride;
    const uint32_t *stamp = D.stamp ? D.stamp + (int64_t)slot * D.d.n : nullptr;
    const WaMaskRef mask = wa_mask_of(D, slot);
    const float clean = c->clean[g & 1];
    const uint32_t evap_now = c->evap_base + (uint32_t)g;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + tid; v < D.d.n; v += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t stv = stamp ? stamp[v] : 1u;
        for (int k = 0; k < 6; k++) {
            const int64_t e = v * 6 + k;
            const float st0 = ph[e];
            float val = fabsf(st0);
            if (stamp) val = stv != 0 ? wa_catch_up(val, evap_now + 1u - stv, R.rho) : (val == 0.f ? 0.f : clean);
            h[0] += wa_h2((unsigned long long)e, (unsigned long long)(__float_as_uint(val) | (__float_as_uint(st0) & 0x80000000u)));
            const unsigned long long m = wa_mask_get(mask, e);
            if (m) h[2] += wa_h2((unsigned long long)e, m);
        }
        if (stamp && stv != 0) h[1] += wa_h2((unsigned long long)v, 1ULL);
    }
    if (blockIdx.x == 0) {
        for (int32_t a = 0; a < D.max_colony; a++) {
            const float La = D.antL[(int64_t)slot * D.max_colony + a];
            const int32_t na = D.antLen[(int64_t)slot * D.max_colony + a];
            if (tid == 0) h[3] += wa_h2((unsigned long long)a, ((unsigned long long)__float_as_uint(La) << 32) | (uint32_t)na);
            const int32_t *path = D.paths + ((int64_t)slot * D.max_colony + a) * D.path_cap;
            if (a < c->colony[(g + 1) & 1] || a < c->colony[g & 1])
                for (int32_t j = tid; j < na && j < D.path_cap; j += blockDim.x) h[3] += wa_h2((unsigned long long)a * D.path_cap + j, (unsigned long long)(uint32_t)path[j]);
        }
    }
    if (blockIdx.x == 1 && c->bestL != INFINITY) {
        const int32_t blen = c->best_len;
        for (int32_t i = tid; i < blen; i += blockDim.x) {
            h[4] += wa_h2((unsigned long long)i, (unsigned long long)(uint32_t)D.bestpath[(int64_t)slot * D.path_cap + i]);
            h[6] += wa_h2((unsigned long long)i + (1ULL << 32), (unsigned long long)D.besttabu[(int64_t)slot * D.path_cap + i]);
            if (D.rtab) for (int k = 0; k < 8; k++) h[7] += wa_h2((unsigned long long)i * 8 + k + (2ULL << 32), (unsigned long long)__float_as_uint(D.rtab[((int64_t)slot * D.path_cap + i) * 8 + k]));
        }
    }
    if (blockIdx.x == 2 && tid == 0) {
        unsigned long long x = 0;
        // (best_ver, tabu_gen and evap_base carry over from solve to solve; n_dep / dep_* / perm / depA are the previous solve's until the first ranking)
        x = wa_h2(x, __float_as_uint(c->bestL)); x = wa_h2(x, (uint32_t)c->best_len);
        x = wa_h2(x, (uint32_t)c->colony[0]); x = wa_h2(x, (uint32_t)c->colony[1]); x = wa_h2(x, __float_as_uint(c->lambda[0])); x = wa_h2(x, __float_as_uint(c->lambda[1]));
        x = wa_h2(x, __float_as_uint(c->Q[0])); x = wa_h2(x, __float_as_uint(c->Q[1])); x = wa_h2(x, __float_as_uint(c->clean[g & 1])); x = wa_h2(x, (uint32_t)g);
        if (phase > 0) {
            x = wa_h2(x, (uint32_t)c->n_dep); x = wa_h2(x, (uint32_t)(c->tabu_gen == g - 1));
            x = wa_h2(x, __float_as_uint(c->dep_lambda)); x = wa_h2(x, __float_as_uint(c->dep_Q)); x = wa_h2(x, __float_as_uint(c->dep_bestL));
            for (int32_t o = 0; o < c->n_dep && o < D.max_colony; o++) {
                x = wa_h2(x, (uint32_t)D.perm[(int64_t)slot * D.max_colony + o]); x = wa_h2(x, __float_as_uint(D.depA[(int64_t)slot * D.max_colony + o]));
            }
        }
        if (D.dcount) { x = wa_h2(x, (uint32_t)D.dcount[slot * 2]); x = wa_h2(x, (uint32_t)D.dcount[slot * 2 + 1]); }
        h[5] = x;
    }
    __shared__ unsigned long long s_h[8];
    if (tid < 8) s_h[tid] = 0;
    __syncthreads();
    for (int i = 0; i < 8; i++) {
        unsigned long long x = h[i];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if ((tid & 63) == 0 && x) atomicAdd(&s_h[i], x);
    }
    __syncthreads();
    if (tid < 8 && s_h[tid]) atomicAdd(&out[(int64_t)slot * 8 + tid], s_h[tid]);
}
#endif
