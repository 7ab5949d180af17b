// acs_walk.hpp -- the ant walk (ACSRank_3D.hpp:134-193, :252-261): LDS tabu hash with bitmap spill, the compiler-scheduled step, the
// hand-scheduled gfx950 loop (walk_loop_gfx950.hpp), best-path replay and re-entry, the replay table, k_walk_dev / k_walk_ref.
// Part of acs_kernels.hpp (included from there, in this order).
#pragma once
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
    // 16-bit entries (round 6, VERDICT r05 task 1): half the LDS per walk block, so twice the resident blocks where a launch has more blocks
    // than fit (160 KB per CU: nine blocks of a 2^12 table with 32-bit keys, sixteen -- the register file's limit -- with these).  An entry
    // names its key EXACTLY: h = id * K_B mod 2^B is a bijection on the grid's B-bit voxel ids (K_B odd: the golden multiplier of width B,
    // which spreads the near-sequential ids of a lattice walk like the 32-bit one does); the slot index is h's top hl bits, the entry the
    // next 12 bits (hl + 12 >= B: nothing of h is left out) and, in its low nibble, how far behind its home slot it sits (linear probing).
    // 0xFFFF = empty, 0xFFFE = the sentinel behind the table; displacements 0..13 can be said.  Measured on the oracle's own ant paths
    // (tests/tools/tabu16_model.py: 92 first-generation walks of up to 1 772 nodes on 256^3, 2^12 slots): largest displacement 13, none
    // beyond.  A probe or an insert that WOULD run past 13 sets `ovf` and the walk spills to its bitmap like one that outgrows the table.
    bool t16 = false;
    uint32_t kmul = 0;       // K_B << (32 - B): (id * kmul) holds h in its top B bits
    int32_t hl = 0;          // log2 of the slots
    int32_t ovf_dw = 0;      // dword index (from tab) of the overflow flag: the LDS word behind the sentinel (every table has it; only 16-bit tables set it)
};
#define WA_T16_EMPTY 0xFFFFu
#define WA_T16_SENTINEL 0xFFFEu
#define WA_T16_MAXD 13u
__device__ __forceinline__ uint32_t wa_t16_idx(const WaTabu &t, uint32_t tt) { return tt >> (32 - t.hl); }
__device__ __forceinline__ uint32_t wa_t16_cmp(const WaTabu &t, uint32_t tt) { return (tt >> (16 - t.hl)) & 0xFFF0u; }
__device__ __forceinline__ bool tabu_has(const WaTabu &t, int32_t key)
{
    if (t.spilled) {
        uint32_t w = __hip_atomic_load(&t.bits[(uint32_t)key >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (w >> (key & 31)) & 1u;
    }
    if (t.t16) {
        const volatile uint16_t *tb = reinterpret_cast<const volatile uint16_t *>(t.tab);
        const uint32_t tt = (uint32_t)key * t.kmul, c = wa_t16_cmp(t, tt);
        uint32_t h = wa_t16_idx(t, tt);
        for (uint32_t d = 0; d <= WA_T16_MAXD; d++) {
            const uint32_t v = tb[h];
            if (v == c + d) return true;
            if (v == WA_T16_EMPTY) return false;
            h = (h + 1) & t.mask;
        }
        return false;   // fourteen slots without the key: it is not in the table (no insert ever lands further from home; one that would, sets ovf and the walk spills)
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
    if (t.t16) {
        volatile uint16_t *tb = reinterpret_cast<volatile uint16_t *>(t.tab);
        const uint32_t tt = (uint32_t)key * t.kmul, c = wa_t16_cmp(t, tt);
        uint32_t h = wa_t16_idx(t, tt);
        for (uint32_t d = 0; d <= WA_T16_MAXD; d++) {
            if (tb[h] == WA_T16_EMPTY) { tb[h] = (uint16_t)(c + d); return; }
            h = (h + 1) & t.mask;
        }
        reinterpret_cast<volatile int32_t *>(t.tab)[t.ovf_dw] = 1;
        return;
    }
    uint32_t h = ((uint32_t)key * 2654435761u) >> t.shift;
    while (t.tab[h] != WA_HASH_EMPTY) h = (h + 1) & t.mask;
    t.tab[h] = key;
}
// the same by many lanes at once (distinct keys, no deletions: any insertion order gives a valid open-addressing table): compare-and-swap on
// the slot -- for 16-bit entries on the word that holds it
__device__ __forceinline__ void tabu_insert_atomic(const WaTabu &t, int32_t key)
{
    if (t.t16) {
        uint32_t *tw = reinterpret_cast<uint32_t *>(t.tab);
        const uint32_t tt = (uint32_t)key * t.kmul, c = wa_t16_cmp(t, tt);
        uint32_t h = wa_t16_idx(t, tt), d = 0;
        while (d <= WA_T16_MAXD) {
            const uint32_t w = __hip_atomic_load(&tw[h >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), sh = (h & 1u) * 16u;
            if (((w >> sh) & 0xFFFFu) == WA_T16_EMPTY) {
                if (atomicCAS(&tw[h >> 1], w, (w & ~(0xFFFFu << sh)) | ((c + d) << sh)) == w) return;
                continue;   // the word changed under us (its other half, or this slot): look again
            }
            h = (h + 1) & t.mask;
            d++;
        }
        reinterpret_cast<volatile int32_t *>(t.tab)[t.ovf_dw] = 1;
        return;
    }
    uint32_t h = ((uint32_t)key * 2654435761u) >> t.shift;
    while (atomicCAS(&t.tab[h], WA_HASH_EMPTY, key) != WA_HASH_EMPTY) h = (h + 1) & t.mask;
}
// did a probe or an insert of this wavefront run past what a 16-bit entry can say?  (wave-uniform; every lane's LDS accesses are done)
__device__ __forceinline__ bool tabu_overflowed(const WaTabu &t)
{
    if (!t.t16 || t.spilled) return false;
    __builtin_amdgcn_wave_barrier();
    return __builtin_amdgcn_readfirstlane(reinterpret_cast<volatile int32_t *>(t.tab)[t.ovf_dw]) != 0;
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
                                          int32_t *flags_out, int32_t max_steps = 0x7fffffff)
{
    // max_steps = 1: ONE step by the book and back to the caller with the walk unfinished (the hand-scheduled loop with 16-bit tabu entries hands a step
    // over when a probe chain outruns what an entry can say: the lookups here are exact -- fourteen slots decide -- and the loop is re-entered)
    const int lane = threadIdx.x;
    const int32_t nx = D.d.nx, nxy = D.d.nxy;
    int32_t cur = st.cur, len = st.len;
    uint32_t step = st.step;
    float L = st.L;
    const int k = lane;
    const int32_t dk = wa_delta(k, nx, nxy);
    bool finished = true;
    for (;;) {
        if (!T.spilled && (len > spill_at || tabu_overflowed(T))) {  // hash nearly full (or a 16-bit entry could not be said): move the set to the bitmap
#if !defined(WA_ANT_TIME) && !defined(WA_STAMPS)
            if (lane == 0 && D.dbg) {   // wa_acs_debug_counters: [13] walks that spilled, [14] of them because a 16-bit entry could not be said
                atomicAdd(&D.dbg[13], 1ULL);
                if (len <= spill_at) atomicAdd(&D.dbg[14], 1ULL);
            }
#endif
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
        if (--max_steps == 0) { finished = false; break; }
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
    st.done = finished;
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
// draws != nullptr (REF mode, k_walk_ref_spec): the draw of step i is draws[i] -- the libc stream's output the ant would be dealt if every
// ant in front of it followed the whole best path -- instead of the counter hash
__device__ __forceinline__ int wa_walk_replay(const float *__restrict__ T, int32_t rlen, uint64_t antkey, int32_t &node, const int32_t *__restrict__ draws = nullptr)
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
        float rnd = (float)(draws ? (valid ? __hip_atomic_load(&draws[nodev], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0) : (int32_t)wa_ctr_draw(antkey, (uint32_t)nodev)) / 2147483648.0f;  // (float)rand()/(float)RAND_MAX (:169)
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
        if (tabu_overflowed(V)) { stop = q; return 3; }   // a lookup of V was left undecided (16-bit entries): no row can be trusted -- the general step decides (and spills)
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
__device__ __forceinline__ void wa_tabu_clear(int4 *tab4, int hash_log2, bool t16 = false)
{
    const int n16 = ((t16 ? 2 : 4) << hash_log2) / 16, lane = threadIdx.x;   // 16-byte units of the table (0xFFFF = the empty 16-bit entry: the same bytes)
    const int4 e = make_int4(-1, -1, -1, -1);
    int i = lane;
    for (; i + 7 * 64 < n16; i += 8 * 64) {
#pragma unroll
        for (int u = 0; u < 8; u++) tab4[i + u * 64] = e;
    }
    for (; i < n16; i += 64) tab4[i] = e;
    // the entry behind the table is a sentinel: the hand-scheduled loop reads every probed slot together with its successor, and the
    // successor of the LAST slot is this one -- neither empty nor any key, so that lane takes the slow path to slot 0.  The word behind it is
    // the overflow flag of the 16-bit tables (WaTabu::ovf)
    if (lane == 0) {
        int32_t *tab = reinterpret_cast<int32_t *>(tab4);
        const int dw = t16 ? (1 << hash_log2) / 2 : (1 << hash_log2);
        tab[dw] = t16 ? (int32_t)(0xFFFF0000u | WA_T16_SENTINEL) : WA_HASH_SENTINEL;
        tab[dw + 1] = 0;
    }
}

template <int MODE, bool ALPHA1, bool SPARSE, bool WARM = true, bool REJ = true, bool DIRECT = false, bool T16 = false>
__device__ __forceinline__ void wa_walk_one(const WaAcsDev &D, const WaRun &R, int32_t slot, int32_t ant,
                                            int32_t start, int32_t end, uint64_t antkey, int32_t *tab,
                                            int hash_log2, int32_t &rng_rs, int32_t &rng_f, int32_t &rng_b,
                                            int32_t *flags_out, int32_t rlen, float bestL, float clean, uint32_t evap_now, int32_t walk_flags,
                                            uint32_t best_ver, int32_t heur_slot, int32_t cut_n = 0x7fffffff, const int32_t *res_words = nullptr,
                                            int32_t res_len = 0, int32_t gen = 0, int32_t bits_row = -1,
                                            const int32_t *ref_draws = nullptr, const int32_t *ref_snap = nullptr, int32_t *replayed_out = nullptr)
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
    } else if ((MODE == 1 || ref_draws) && rlen > 1) {
        // (REF mode, ref_draws: the next rlen - 1 outputs of the libc stream, dealt ahead WITHOUT committing them; ref_snap: the rotated
        //  state in front of every 64th of them -- the stream is then taken to the draws the ant really consumed, see k_walk_ref)
        int32_t node = 0;
        // the first 512 words of the best path are requested BEFORE the replay decides how many of them the ant walks: their round trip
        // runs beside the table rows' (once converged every ant copies all of them)
        int32_t w0[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int32_t q = u * 64 + lane;
            w0[u] = q < rlen ? bpath[q] : 0;
        }
        const int what = wa_walk_replay(D.rtab + (int64_t)slot * D.path_cap * 8, rlen, antkey, node, ref_draws);
        if (MODE == 0) {   // one draw per step taken; a dead end on the path (what == 1: candidates, none picked) consumed its draw too (:169 before :178)
            wa_glibc_seek(ref_snap, what == 2 ? rlen - 1 : what == 1 ? node + 1 : node, rng_rs, rng_f, rng_b);
            if (replayed_out) *replayed_out = node;
        }
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
                if (MODE == 1 && what == 2) D.antRep[(int64_t)slot * D.max_colony + ant] = 1;   // its path IS the best path it replayed (cleared at the top of this walk)
                if (what == 2 && cut_n != 0x7fffffff) __hip_atomic_store(&sg.arr_len[WA_ARR_IDX & 255u], (uint32_t)st.len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an arrival, for the straggler check (write-through: the checking ants sit on other XCDs)
            }
            return;
        }
        st.cur = bpath[node] & (int32_t)WA_ID_MASK;
        st.step = (uint32_t)node;                                // steps taken so far = draws consumed
        for (int32_t q = 0; q < node; q++) st.L += R.precision;  // :78, one add per step taken
        prefix_words = bpath;
    }
    // T16: 16-bit tabu entries (the host launches those instantiations for saturated launches of grids whose ids an entry can name: WaAcsDev::tab16_kmul)
    static_assert(!T16 || (!WARM && !DIRECT && MODE == 1 && ALPHA1), "16-bit tabu entries: DEV mode, alpha 1, the loops without touch loads");
    constexpr bool t16 = T16;
    WaTabu T;
    T.tab = tab;
    T.mask = (1u << hash_log2) - 1u;
    T.shift = 32 - hash_log2;
    T.t16 = t16; T.kmul = D.tab16_kmul; T.hl = hash_log2;
    T.ovf_dw = (t16 ? (1 << hash_log2) / 2 : (1 << hash_log2)) + 1;
    // (a resume block spills into a bitmap row of its own, behind the ants' rows: the ant's row belongs to the running generation's ant)
    T.bits = D.vbits + ((int64_t)slot * D.vbits_rows + (bits_row >= 0 ? bits_row : ant)) * D.vbits_words;
    T.spilled = false;
    const int32_t spill_at = (int32_t)((3u << hash_log2) >> 2);

    WA_PHASE(6);
    int4 *tab4 = reinterpret_cast<int4 *>(tab);
    wa_tabu_clear(tab4, hash_log2, t16);
    __builtin_amdgcn_wave_barrier();
    WA_PHASE(7);
    if (prefix_words && st.len <= spill_at) {  // (a longer prefix goes straight to the spilled slow loop)
        // tabu set := the replayed prefix.  Distinct keys, no deletions: any insertion order gives a valid
        // open-addressing table, so the lanes insert concurrently with compare-and-swap on the slot.
        for (int32_t q = lane; q < st.len; q += 64) tabu_insert_atomic(T, prefix_words[q] & (int32_t)WA_ID_MASK);
    } else if (!prefix_words && lane == 0) {
        tabu_insert(T, start);  // addStartNode :81-86 (path[0] is buffered by the fast loop)
    }
    __builtin_amdgcn_wave_barrier();
    // (16-bit entries: a prefix that could not be filed -- or a build / parameter set without the hand-scheduled loop, which alone knows them -- goes
    //  straight to the slow loop, which spills)
    int32_t fast_limit = (int32_t)D.path_cap < spill_at + 1 ? (int32_t)D.path_cap : spill_at + 1;
    bool use_asm = false;
#ifndef WA_STAMPS
    use_asm = ALPHA1 && (walk_flags & 1) && (MODE == 1 || !SPARSE);   // (REF mode: the same loop, draws from the libc stream)
#endif
    if (t16 && (!use_asm || tabu_overflowed(T))) {
        if (lane == 0) tab[T.ovf_dw] = 1;
        __builtin_amdgcn_wave_barrier();
        fast_limit = 0;
    }
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
        // (the path so far stays where it is: the next generation's ants write the OTHER paths array, the resume block walks on in this one)
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
            wa_walk_fast_asm<SPARSE ? 3 : 2, WARM, false, DIRECT, T16>(R, pher, heur, stamp, clean_info, evap_now, path, tab, hash_log2, D.d.nx, D.d.nxy, (int32_t)D.path_cap, end, antkey, spill_at,
                                             D.guard_bytes, D.stamp_guard_bytes, D.ltab, st, flags_out, prefix_words, nullptr, mark, knob_anywhere ? 0u : best_ver, hold,
                                             SPARSE ? nullptr : sg.arr_len, cut_n, nullptr, nullptr, nullptr, t16 ? T.kmul : 0u);
            prefix_words = path;                                  // from now on the ant's own words (its partial block is in memory)
#ifdef WA_ANT_TIME
            if (dbg_t_hand) { dbg_t_hand = 0; }
#endif
            if (T16 && !st.done && st.reason == 6) {   // 16-bit entries: a probe chain outran what an entry can say -- this ONE step by the book, then back into the loop
                wa_walk_slow<MODE, SPARSE>(D, R, pher, heur, stamp, clean_info, evap_now, path, T, end, antkey, rng_rs, rng_f, rng_b, spill_at, st, flags_out, 1);
                st.pbuf_valid = false;
                if (st.done || tabu_overflowed(T) || st.len >= fast_limit) break;
                continue;
            }
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
                    tabu_insert_atomic(T, w & (int32_t)WA_ID_MASK);
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
            if (tabu_overflowed(T)) break;   // (16-bit entries: a lookup or a commit ran past what an entry can say -- the slow loop spills and goes on)
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
        for (;;) {
            wa_walk_fast_asm<SPARSE ? 1 : 0, WARM, false, DIRECT, T16>(R, pher, heur, stamp, clean_info, evap_now, path, tab, hash_log2, D.d.nx, D.d.nxy, (int32_t)D.path_cap, end, antkey, spill_at,
                                     D.guard_bytes, D.stamp_guard_bytes, D.ltab, st, flags_out, prefix_words, (slot == 0 && ant == 0) ? D.dbg : nullptr,
                                     nullptr, 0, 0, SPARSE ? nullptr : sg.arr_len, cut_n, nullptr, nullptr, nullptr, t16 ? T.kmul : 0u);
            if (!T16 || st.done || st.reason != 6) break;
            // 16-bit entries: a probe chain outran what an entry can say -- this ONE step by the book (the lookups there are exact), then back into the loop
            wa_walk_slow<MODE, SPARSE>(D, R, pher, heur, stamp, clean_info, evap_now, path, T, end, antkey, rng_rs, rng_f, rng_b, spill_at, st, flags_out, 1);
            prefix_words = path;
            st.pbuf_valid = false;
            if (st.done || tabu_overflowed(T) || st.len >= fast_limit) break;
        }
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
        if (walk_flags & 64) {   // drain launch (no newer generation's ant owns the agents[] entry): the finished walk's result goes there
            if (lane == 0) {     // (its path is where it always was: the previous generation's paths array, which is what wa_acs_read_ant_path reads)
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
// DIRECT: the hand-scheduled loop without look-ahead (saturated launches; see walk_loop_gfx950.hpp, W = DIRECT)
template <bool ALPHA1, bool SPARSE, bool WARM = true, bool REJ = true, bool DIRECT = false, bool T16 = false>
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
        wa_walk_one<1, true, false, WARM, false, DIRECT, T16>(Dp, R, slot, a, c->start, c->end, key, lds, hash_log2, rs0, f0, b0, &D.ctl[slot].flags, 0, INFINITY, 0.f, 0u,
                                                 walk_flags & (1 | 64), 0u, c->heur_slot, 0x7fffffff, D.prev_paths + ((int64_t)slot * D.max_colony + a) * D.path_cap, n0, gen - 1,
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
    // the "arrived on the replay track" flag of this generation's ant: cleared here, set by wa_walk_one if it does (a resume block, above, finishes a
    // PREVIOUS generation's ant and leaves the flag alone)
    if (threadIdx.x == 0) D.antRep[(int64_t)slot * D.max_colony + ant] = 0;
    wa_walk_one<1, ALPHA1, SPARSE, WARM, REJ, DIRECT, T16>(D, R, slot, ant, c->start, c->end, antkey, lds, hash_log2, rs_unused, f, b, &D.ctl[slot].flags, rlen, bestL,
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

// ---- REF mode once the colony has converged: every ant re-walks the best path, L = best_len - 1 steps and as many draws each, so ant a
// is dealt stream outputs [a * L, (a + 1) * L) -- IF every ant in front of it does the same.  Three launches per generation:
//   k_ref_draws      one wavefront generates colony * L outputs of the libc stream ahead (64 per pass, wa_glibc_block) and keeps the
//                    rotated state in front of every 64th output;
//   k_walk_ref_spec  one wavefront per ant checks with the replay table whether the ant, dealt those draws, follows the whole best path
//                    (the same test the DEV-mode replay makes, one lane per node) and, if so, delivers its result;
//   k_walk_ref       takes the stream to the first ant A that did NOT (all ants in front of it are confirmed, so ITS offset is right),
//                    and walks ants A .. colony-1 one after another as ever.  A = colony: nothing left to walk.
// Every ant's draws are the reference's.
// The generator in two steps, because one wavefront producing 64 outputs per pass needs 1 ms for the ~97 000 outputs of a generation:
//   k_ref_draws        one wavefront takes the state from superblock to superblock (WA_REF_SUPER = 1024 outputs each) by a matrix-vector
//                      product over Z/2^32 -- the recurrence is linear: state(n + 1024) = J * state(n), J = companion matrix ^ 1024, computed
//                      once on the host -- lane i = row i, 31 broadcast multiply-adds per superblock (~95 superblocks: ~20 us);
//   k_ref_draws_fill   one wavefront per superblock generates its 1024 outputs (16 passes of wa_glibc_block) and keeps the rotated state in
//                      front of every 64th.
__global__ __launch_bounds__(64) void k_ref_draws(WaAcsDev D, WaRun R, int32_t gen)
{
    const int lane = threadIdx.x;
    const WaSlotCtl *c = &D.ctl[0];
    const int32_t colony = c->colony[gen & 1];
    const int32_t L = (D.rtab && c->bestL != INFINITY) ? c->best_len - 1 : 0;
    // only once the best path has been stable for a few generations: while the colony explores the first ant already leaves it
    const bool on = R.alpha == 1 && L >= 1 && L <= WA_REF_SPEC_LEN && colony >= 2 && colony <= D.max_colony && gen - c->tabu_gen >= 3;
    if (lane == 0) { D.ref_ok[D.max_colony] = on ? 1 : 0; D.ref_ok[D.max_colony + 1] = L; }
    if (!on) return;
    uint32_t row[31];                                    // row `lane` of the jump matrix
#pragma unroll
    for (int j = 0; j < 31; j++) row[j] = lane < 31 ? D.ref_jump[lane * 31 + j] : 0u;
    int32_t rot = wa_glibc_rotate(lane < 31 ? D.rng->r[lane] : 0, D.rng->f);
    const int64_t total = (int64_t)colony * L;
    const int64_t supers = (total + WA_REF_SUPER - 1) / WA_REF_SUPER;
    for (int64_t sb = 0; sb <= supers; sb++) {           // (one more than needed: the state behind the last output of a full last superblock)
        if (lane < 32) D.ref_state[sb * (WA_REF_SUPER / 64) * 32 + lane] = rot;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 31; j++) acc += row[j] * (uint32_t)__builtin_amdgcn_readlane(rot, j);
        rot = (int32_t)acc;
    }
}

__global__ __launch_bounds__(64) void k_ref_draws_fill(WaAcsDev D, WaRun R, int32_t gen)
{
    const int lane = threadIdx.x;
    if (!D.ref_ok[D.max_colony]) return;
    const int64_t total = (int64_t)D.ctl[0].colony[gen & 1] * D.ref_ok[D.max_colony + 1];
    const int64_t t0 = (int64_t)blockIdx.x * WA_REF_SUPER;
    if (t0 >= total) return;
    int32_t rot = lane < 32 ? D.ref_state[(t0 >> 6) * 32 + lane] : 0;
    for (int64_t t = t0; t < t0 + WA_REF_SUPER; t += 64) {
        if (t != t0 && lane < 32) D.ref_state[(t >> 6) * 32 + lane] = rot;   // (also at t == total: the state behind the last output)
        if (t >= total) break;
        int32_t raw;
        wa_glibc_block_raw<64, 0>(rot, raw);
        if (t + lane < total) D.ref_draws[t + lane] = raw;
    }
}

__global__ __launch_bounds__(64) void k_walk_ref_spec(WaAcsDev D, WaRun R, int32_t gen)
{
    const int lane = threadIdx.x;
    const int32_t ant = blockIdx.x;
    if (!D.ref_ok[D.max_colony]) return;
    const WaSlotCtl *c = &D.ctl[0];
    const int32_t colony = c->colony[gen & 1];
    if (ant >= colony) return;
    const int32_t L = D.ref_ok[D.max_colony + 1], rlen = L + 1;
    int32_t node = 0;
    const int what = wa_walk_replay(D.rtab, rlen, 0, node, D.ref_draws + (int64_t)ant * L);
    if (what != 2) {
        if (lane == 0) D.ref_ok[ant] = 0;
        return;
    }
    // the ant re-walked the best path: its agents[] entry is that path (:76-78), arriving accumulates exactly the steps that produced best.L
    int32_t *path = D.paths + (int64_t)ant * D.path_cap;
    for (int32_t q = lane; q < rlen; q += 64) path[q] = D.bestpath[q];
    if (lane == 0) {
        D.antL[ant] = c->bestL;
        D.antLen[ant] = rlen;
        D.ref_ok[ant] = 1;
    }
}

// REF: grid = (1, 1): the ants of the single in-flight problem walk one after another and draw
// from the shared glibc stream in exactly the reference's order (:252-261)
// walk_flags bit 0 (and alpha == 1): the hand-scheduled loop with draws from the libc stream, 64 at a time (walk_loop_gfx950.hpp, REFDRAW)
// walk_flags bit 1: k_ref_draws / k_walk_ref_spec ran in front of this launch (see above)
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
    int32_t first = 0;
    if ((walk_flags & 2) && D.ref_ok[D.max_colony]) {
        // the first ant that did not follow the whole best path with the draws it was dealt; everything in front of it stands
        first = colony;
        for (int32_t a0 = 0; a0 < colony; a0 += 64) {
            const int32_t a = a0 + (int32_t)threadIdx.x;
            const unsigned long long bad = __ballot(a < colony && D.ref_ok[a] == 0);
            if (bad) { first = a0 + __ffsll((long long)bad) - 1; break; }
        }
        // the stream in front of that ant's first draw = output index first * L: the kept state in front of the 64-block it lies in,
        // then the remaining outputs one by one
        const int64_t t = (int64_t)first * D.ref_ok[D.max_colony + 1];
        const int32_t rot = threadIdx.x < 32 ? D.ref_state[(t >> 6) * 32 + threadIdx.x] : 0;
        f = (int32_t)((f + ((t >> 6) << 6)) % 31);
        r = wa_glibc_unrotate(rot, f);
        b = f + 28;
        b = b >= 31 ? b - 31 : b;
        for (int32_t q = 0; q < (int32_t)(t & 63); q++) (void)wa_glibc_next_lanes(r, f, b);
#if !defined(WA_STAMPS) && !defined(WA_ANT_TIME)
        if (threadIdx.x == 0 && D.dbg) atomicAdd(&D.dbg[6], (unsigned long long)first);   // ants confirmed by the speculation (wa_acs_debug_counters)
#endif
    }
    if (R.alpha == 1 && (walk_flags & 1)) {
        // The ants that are left walk one after another.  While the replay table is good (same condition as the speculation) each of them
        // is first dealt the next L outputs of the stream WITHOUT committing them and follows the best path for as long as its own draws
        // take the path's edges (wa_walk_replay, one lane per node); the stream is then taken to the draws it really consumed and the
        // general loop goes on from where it left the path.  Given up for the rest of the generation when the ants leave the path early.
        const bool prefix_ok = (walk_flags & 2) && D.ref_ok[D.max_colony];
        const int32_t L = prefix_ok ? D.ref_ok[D.max_colony + 1] : 0;
        int32_t tried = 0, followed = 0;
        for (int32_t ant = first; ant < colony; ant++) {
            const bool prefix = prefix_ok && !(tried >= 16 && followed < 48 * tried);
            if (prefix) {
                int32_t rot = wa_glibc_rotate(r, f);
                for (int32_t t = 0; t < L; t += 64) {
                    if (threadIdx.x < 32) D.ref_state[(t >> 6) * 32 + threadIdx.x] = rot;
                    int32_t raw;
                    wa_glibc_block_raw<64, 0>(rot, raw);
                    if (t + (int32_t)threadIdx.x < L) D.ref_draws[t + threadIdx.x] = raw;
                }
                if (threadIdx.x < 32) D.ref_state[((L + 63) >> 6) * 32 + threadIdx.x] = rot;
                __threadfence();   // the replay reads them back through L2
                int32_t replayed = 0;
                wa_walk_one<0, true, false, true, false>(D, R, slot, ant, start, end, 0, lds, hash_log2, r, f, b, &D.ctl[slot].flags, L + 1, c->bestL, 0.f, 0u, 1, 0u, heur_slot,
                                                         0x7fffffff, nullptr, 0, 0, -1, D.ref_draws, D.ref_state, &replayed);
                tried++;
                followed += replayed;
            } else {
                wa_walk_one<0, true, false, true, false>(D, R, slot, ant, start, end, 0, lds, hash_log2, r, f, b, &D.ctl[slot].flags, 0, INFINITY, 0.f, 0u, 1, 0u, heur_slot);
            }
        }
    } else {
        for (int32_t ant = first; ant < colony; ant++)
            wa_walk_one<0, false, false>(D, R, slot, ant, start, end, 0, lds, hash_log2, r, f, b, &D.ctl[slot].flags, 0, INFINITY, 0.f, 0u, 0, 0u, heur_slot);
    }
    if (threadIdx.x < 31) D.rng->r[threadIdx.x] = r;
    if (threadIdx.x == 0) {
        D.rng->f = f;
        D.rng->b = b;
    }
}
