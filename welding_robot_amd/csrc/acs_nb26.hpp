// acs_nb26.hpp -- the 26-neighbour variant.  Part of acs_kernels.hpp (included from there, in this order).
#pragma once
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
    // lazy evaporation (see wa_table_rows): a best-path node that never received a deposit holds the clean value of the field as it stands
    // now, a deposited one that received nothing this generation may have evaporations pending (read-side catch-up)
    const uint32_t *stamp = D.stamp ? D.stamp + (int64_t)slot * D.d.n : nullptr;
    const float clean_now = ctl->clean[ctl->gen & 1];
    const uint32_t evap_tab = ctl->evap_base + (uint32_t)ctl->gen;   // the fused launch already counted this generation
    for (int32_t i = w; i < blen - 1; i += n_waves) {   // decisions exist at nodes 0 .. blen-2
        const int32_t v = bpath[i] & WaNbT<26>::IDM;
        float p = -0.f, h = 0.f;
        bool adm = false;
        if (lane < 26) {
            const int64_t e = (int64_t)v * 26 + lane;
            p = pher[e];
            h = heur[e];
            if (stamp) {
                const uint32_t stv = stamp[v];
                p = stv == 0 ? copysignf(clean_now, p) : copysignf(wa_catch_up(fabsf(p), evap_tab + 1u - stv, R.rho), p);
            }
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
    // lazy evaporation: voxels that became dirty in this generation join the swept set from the next sweep on (see k_apply_table)
    if (D.dcount && blockIdx.x == 0 && tid == 0) D.dcount[slot * 2] = D.dcount[slot * 2 + 1];
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

// SPARSE: the field of a lazily evaporating solver (round 5; see WaAcsDev and wa_walk_fast): the stamp of the voxel the ant stands on
// decides -- 0: never deposited, every admissible edge is worth the clean value; else the stored value with the evaporations it has
// missed applied one by one.  The stamp travels with the voxel's record (one more load behind the two record loads).
template <int MODE, bool SPARSE = false>
__device__ __forceinline__ void wa_walk_one26(const WaAcsDev &D, const WaRun &R, int32_t slot, int32_t ant, int32_t start,
                                              int32_t end, uint64_t antkey, int32_t *tab, int hash_log2, int32_t &rng_rs,
                                              int32_t &rng_f, int32_t &rng_b, int32_t *flags_out, int32_t rlen,
                                              int32_t cut_n = 0x7fffffff, int32_t *res_words = nullptr, int32_t res_len = 0, float res_L = 0.f,
                                              int32_t gen = 0, int32_t bits_row = -1, bool drain = false, float clean = 0.f, uint32_t evap_now = 0u)
{
    const uint32_t *stamp = SPARSE ? D.stamp + (int64_t)slot * D.d.n : nullptr;
    const float clean_info = SPARSE ? wa_powi(clean, R.alpha) : 0.f;   // power() of the clean value, once per walk
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
            if (drain && lane == 0) {   // drain launch (see k_walk_dev): the finished walk's result goes to agents[]
                D.antL[(int64_t)slot * D.max_colony + ant] = Lf;
                D.antLen[(int64_t)slot * D.max_colony + ant] = lenf;
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
    if (MODE == 1 && !res_words && lane == 0) D.antRep[(int64_t)slot * D.max_colony + ant] = 0;   // (set again below if this ant arrives on the replay track)
    if (res_words) { r_node = res_len - 1; r_L = res_L; }
    else if (MODE == 1 && rlen > 1) {   // follow the best path while the ant's own draws take its edges
        const float *RT = D.rtab + (int64_t)slot * D.path_cap * WA_ROW26;
        const int32_t *bpath = D.bestpath + (int64_t)slot * D.path_cap;
        const int what = wa_walk_replay26(RT, rlen, antkey, r_node);
        for (int32_t q = lane; q <= r_node; q += 64) path[q] = bpath[q];   // the walked prefix IS the best path's
        r_L = RT[(int64_t)r_node * WA_ROW26 + 28];                          // L on arrival at that node
        if (what != 3) {
            if (what == 2 && lane == 0) D.antRep[(int64_t)slot * D.max_colony + ant] = 1;   // arrived on the replay track: its path is the best path (see k_evap_rank_mark)
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
        uint32_t sv = 1u;                                                          // SPARSE: the stamp of the voxel whose record p / h are
        {
            const uint32_t off = (uint32_t)cur * 104u + lane_off;
            asm volatile("global_load_dword %0, %2, %3\n global_load_dword %1, %2, %4\n s_waitcnt vmcnt(0)"
                         : "=&v"(p), "=&v"(h) : "v"(off), "s"(pher_b), "s"(heur_b) : "memory");
            if (SPARSE) sv = stamp[cur];
        }
        const char *stamp_b = reinterpret_cast<const char *>(stamp);
        // tabu probe of neighbour k (:145): ends on the key (visited) or on an empty slot (not visited; where the key would go)
        uint32_t hs = ((uint32_t)cur * 2654435761u + hk) >> T.shift;
        int32_t tv = tab[hs];
        bool alive = true, cut = false;
        int32_t em = 63;   // the straggler check runs when (node count & em) == 0: at block boundaries, every 16 nodes once shorter ants have arrived
        while (len <= spill_at && len < (int32_t)D.path_cap) {
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(p), "+v"(h), "+v"(sv));       // this step's records; the touch loads stay in flight
            const int32_t key = cur + dk;
            while (tv != key && tv != WA_HASH_EMPTY) { hs = (hs + 1) & T.mask; tv = tab[hs]; }   // (rare: the slot held another key)
            const bool adm = lane < 26 && (__float_as_uint(p) >> 31) == 0 && tv != key;   // sign bit: out of bounds or occupied (:148)
            float mag = fabsf(p);
            if (SPARSE) {   // every lane holds the same voxel's stamp: uniform, so scalar control flow
                const uint32_t stv = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv);
                mag = stv == 0 ? clean_info : wa_catch_up(mag, evap_now + 1u - stv, R.rho);
            }
            const float a = adm ? mag * h : 0.f;                                   // :154 (alpha == 1)
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
                if (SPARSE) {   // ... and its stamp, behind them and in front of the touches (vector memory returns in order: vmcnt(4) covers it)
                    const uint32_t soff = (uint32_t)next * 4u;
                    asm volatile("global_load_dword %0, %1, %2" : "=&v"(sv) : "v"(soff), "s"(stamp_b) : "memory");
                }
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
                // (the path so far stays where it is: see WaAcsDev::prev_paths)
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
        float pa = wa_powi(fabsf(p), R.alpha);
        if (SPARSE) {
            const uint32_t stv = stamp[cur];
            pa = stv == 0 ? clean_info : wa_powi(wa_catch_up(fabsf(p), evap_now + 1u - stv, R.rho), R.alpha);
        }
        const float info = pa * h;                                                // :154
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
template <bool SPARSE>
__global__ __launch_bounds__(64) void k_walk_dev26(WaAcsDev D, WaRun R, int hash_log2, int32_t gen, int32_t walk_flags)
{
    extern __shared__ int32_t lds[];
    const int32_t slot = blockIdx.y, ant = blockIdx.x;
    const WaSlotCtl *c = &D.ctl[slot];
    const int32_t colony = c->colony[gen & 1];
    int32_t f = 0, b = 0, rs_unused = 0;
    if (!SPARSE && D.pool_n && (int32_t)blockIdx.x >= D.max_colony) {
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
                         D.prev_paths + ((int64_t)slot * D.max_colony + a) * D.path_cap, n0, L0, gen - 1, D.max_colony + r, (walk_flags & 64) != 0);
        return;
    }
    if (walk_flags & 64) return;   // drain launch: resume blocks only
    if (ant >= colony || colony > D.max_colony) return;
    const uint64_t antkey = wa_ctr_antkey(wa_ctr_key(R.seed, c->stream, (uint32_t)gen), (uint32_t)ant);
    const int32_t rlen = (D.rtab && c->bestL != INFINITY) ? c->best_len : 0;
    // an ant with a larger L than floor(lambda - 1) + 1 arrivals cannot be among the depositing ranks (:200) nor be the iteration's best
    int32_t cut_n = 0x7fffffff;
    if (!SPARSE && (walk_flags & 32) && D.pool_n && R.alpha == 1) cut_n = (int32_t)(c->lambda[gen & 1] - 1.f) + 1;
    if (cut_n < 1) cut_n = 1;
    wa_walk_one26<1, SPARSE>(D, R, slot, ant, c->start, c->end, antkey, lds, hash_log2, rs_unused, f, b, &D.ctl[slot].flags, rlen, cut_n, nullptr, 0, 0.f, gen,
                             -1, false, c->clean[gen & 1], c->evap_base + (uint32_t)gen);
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
