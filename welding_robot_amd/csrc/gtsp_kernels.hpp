// gtsp_kernels.hpp -- device side of ACS_GTSP (ACS_GTSP.hpp): the weld-seam ordering loop.
//
// One workgroup per TSP instance runs the whole computeSolution loop (:255-284) without host
// round trips; one thread per ant (colony = city_num, :192).  fp64 throughout, operations in the
// reference's order: info = pheromone^1 * (h^2 * h^4) (:117-118 through power()), roulette over
// the unvisited set in ascending city order with >= (:127-142), tour length without the closing
// edge (:36-44), evaporation by (1 - 0.1) then iteration-best deposit on all N edges with the
// symmetric copy (:175-184), stop after city_num+1 non-improving iterations (:263,:269-275).
#pragma once
#include "wa_device.h"

struct WaGtspDev {
    const double *dist;   // [inst][n*n]
    double *pher, *h6, *info;  // [inst][n*n]
    double *antL;         // [inst][n]
    int32_t *tours;       // [inst][n ants][n steps][2]
    int32_t *best;        // [inst][n][2]
    uint8_t *inJ;         // [inst][n ants][n]
    int32_t *rbuf;        // REF: n*(n-1) draws of the current iteration
    WaGlibcRand *rng;     // REF
    double *out_cost;     // [inst]
    int32_t *out_iters;   // [inst]
    int32_t n, cnt, max_iterations, rng_mode;
    uint64_t seed;
    uint32_t stream0;
};

__global__ __launch_bounds__(256) void k_gtsp(WaGtspDev G)
{
    const int32_t inst = blockIdx.x, tid = threadIdx.x, n = G.n;
    const int64_t nn = (int64_t)n * n;
    const double *dist = G.dist + inst * nn;
    double *pher = G.pher + inst * nn, *h6 = G.h6 + inst * nn, *info = G.info + inst * nn;
    double *antL = G.antL + (int64_t)inst * n;
    int32_t *tours = G.tours + inst * nn * 2;
    int32_t *best = G.best + (int64_t)inst * n * 2;
    uint8_t *inJ = G.inJ + inst * nn;
    const double INF = (double)0x3f3f3f3f;  // ACS_GTSP.hpp:19
    const double alpha = 0.1;               // :189
    __shared__ double s_pher0, s_bestL, s_last, s_nowL;
    __shared__ int32_t s_bad, s_nowk, s_stop, s_it;

    // readFromGraphFile :239-249 + init_param :201-213
    if (tid == 0) {
        double tmp = 0;
        for (int32_t i = 0; i < n; i++)
            for (int32_t j = i + 1; j < n; j++) tmp += dist[(int64_t)i * n + j];
        s_pher0 = (double)G.cnt / (tmp * n);
        s_bestL = INF;
        s_last = INF;
        s_bad = 0;
        s_stop = 0;
        s_it = 0;
    }
    __syncthreads();
    for (int64_t e = tid; e < nn; e += blockDim.x) {
        int32_t i = (int32_t)(e / n), j = (int32_t)(e % n);
        pher[e] = s_pher0;
        double h = 1 / ((i == j ? 0.0 : dist[e]) + 1e-8);  // :211
        h6[e] = wa_powi(h, 6);                               // power(herustic, beta = 6) :118
    }
    __syncthreads();
    const int32_t max_it = G.max_iterations > 0 ? G.max_iterations : n * n;  // :216
    int32_t rf = 0, rb = 0;
    int32_t rr[31];
    if (G.rng_mode == 0 && tid == 0) {
        for (int i = 0; i < 31; i++) rr[i] = G.rng->r[i];
        rf = G.rng->f;
        rb = G.rng->b;
    }
    for (int32_t it = 0; it < max_it; it++) {
        if (tid == 0) s_stop = s_bad > n ? 1 : 0;  // :263
        __syncthreads();
        if (s_stop) break;
        // reset :103-120
        for (int64_t e = tid; e < nn; e += blockDim.x) {
            info[e] = wa_powi(pher[e], 1) * h6[e];
            inJ[e] = (e / n) == (e % n) ? 0 : 1;
        }
        if (G.rng_mode == 0 && tid == 0)  // the libc draws of this iteration in (step, ant) order
            for (int32_t q = 0; q < n * (n - 1); q++) G.rbuf[q] = wa_glibc_next(rr, rf, rb);
        __syncthreads();
        // construct_solution :146-159 -- ants are independent once the draws are fixed
        const uint64_t key = wa_ctr_key(G.seed, G.stream0 + (uint32_t)inst, (uint32_t)it);
        for (int32_t k = tid; k < n; k += blockDim.x) {
            uint8_t *J = inJ + (int64_t)k * n;
            const uint64_t antkey = wa_ctr_antkey(key, (uint32_t)k);
            int32_t r = k, left = n - 1;
            for (int32_t step = 0; step < n; step++) {
                int32_t next = k;  // r1[k]
                if (left > 0) {    // select_next :122-144
                    int32_t rv = G.rng_mode == 0 ? G.rbuf[step * n + k]
                                                 : (int32_t)wa_ctr_draw(antkey, (uint32_t)step);
                    double rnd = (double)rv / (double)2147483647;
                    const double *row = info + (int64_t)r * n;
                    double sum = 0, sp = 0;
                    for (int32_t c = 0; c < n; c++)
                        if (J[c]) sum += row[c];
                    rnd *= sum;
                    for (int32_t c = 0; c < n; c++)
                        if (J[c]) {
                            sp += row[c];
                            if (sp >= rnd) { next = c; break; }
                        }
                }
                if (J[next]) { J[next] = 0; left--; }
                tours[((int64_t)k * n + step) * 2] = r;
                tours[((int64_t)k * n + step) * 2 + 1] = next;
                r = next;
            }
            double L = 0;  // ACS_Tour::calc :36-44
            for (int32_t e = 0; e < n - 1; e++) {
                int32_t a = tours[((int64_t)k * n + e) * 2], b = tours[((int64_t)k * n + e) * 2 + 1];
                L += a == b ? 0.0 : dist[(int64_t)a * n + b];
            }
            antL[k] = L;
        }
        __syncthreads();
        // update_pheromone :161-185
        if (tid == 0) {
            double nowL = INF;
            int32_t nowk = -1;
            for (int32_t k = 0; k < n; k++)
                if (antL[k] < nowL) { nowL = antL[k]; nowk = k; }
            s_nowL = nowL;
            s_nowk = nowk;
        }
        __syncthreads();
        const int32_t nowk = s_nowk;
        if (nowk >= 0 && s_nowL < s_bestL)
            for (int32_t e = tid; e < 2 * n; e += blockDim.x) best[e] = tours[(int64_t)nowk * n * 2 + e];
        for (int64_t e = tid; e < nn; e += blockDim.x) pher[e] *= (1 - alpha);
        __syncthreads();
        if (tid == 0) {
            if (nowk >= 0) {
                if (s_nowL < s_bestL) s_bestL = s_nowL;
                for (int32_t e = 0; e < n; e++) {
                    int32_t a = tours[((int64_t)nowk * n + e) * 2], b = tours[((int64_t)nowk * n + e) * 2 + 1];
                    pher[(int64_t)a * n + b] += 1. / (double)s_nowL;
                    pher[(int64_t)b * n + a] = pher[(int64_t)a * n + b];
                }
            }
            if (s_last > s_bestL) { s_last = s_bestL; s_bad = 0; }
            else s_bad++;
            s_it = it + 1;
        }
        __syncthreads();
    }
    if (tid == 0) {
        G.out_cost[inst] = s_bestL;
        G.out_iters[inst] = s_it;
        if (G.rng_mode == 0) {
            for (int i = 0; i < 31; i++) G.rng->r[i] = rr[i];
            G.rng->f = rf;
            G.rng->b = rb;
        }
    }
}

// ------------------------------------------------------------------ fast path, n <= 256 cities
// lanes = ants (ant k = thread k; ceil(n/64) wavefronts per instance).  What makes it fast:
//   * `info` (pheromone^1 * heuristic^6, rebuilt every iteration :114-119) lives in LDS with an ODD
//     row stride (n | 1 doubles), so 64 ants standing on 64 different cities read 64 different
//     banks; pheromone / heuristic stay in global memory (L2-resident, touched n^2 per iteration);
//   * the unvisited set J[k] (a std::set in the reference, :98) is a register bitmask;
//   * the two ordered passes of select_next (:127-142) are plain predicated loops over the
//     cities: adding +0.0 for a visited city leaves the fp64 partial sums bit-identical;
//   * the tour length is accumulated while the tour is built -- same additions, same order as
//     ACS_Tour::calc (:36-44), closing edge excluded.
// NW = number of 64-bit words of the unvisited mask (n <= 64*NW).  INFO_LDS: info fits in LDS.
// 1.0 or 0.0 from bit c of a 32-bit mask word, built with two integer ops (no compare / select):
// the visited mask then enters the ordered sum as fma(x, m, sum) -- exact, because x * 1.0 and
// x * 0.0 are exact for finite x, so the fma rounds once exactly like `sum + x` / leaves sum alone.
__device__ __forceinline__ double wa_bit_as_double(uint32_t word, int c)
{
    const uint32_t hi = ((word >> c) & 1u) * 0x3FF00000u;
    return __hiloint2double((int)hi, 0);
}
template <int NW, bool INFO_LDS, bool PREFIX>
__global__ __launch_bounds__(256) void k_gtsp_fast(WaGtspDev G)
{
    extern __shared__ double lds_info[];
    const int32_t inst = blockIdx.x, tid = threadIdx.x, n = G.n;
    // rows are padded to the full mask width (+1: odd stride) so the city loops have a fixed trip
    // count of 64 per mask word and unroll into batches of independent LDS reads feeding the
    // (inherently serial) fp64 add chain; columns >= n are never selected (their mask bits are 0)
    const int32_t ld = 64 * NW + 1;
    const int64_t nn = (int64_t)n * n;
    const double *dist = G.dist + inst * nn;
    double *pher = G.pher + inst * nn, *h6 = G.h6 + inst * nn;
    double *info = INFO_LDS ? lds_info : G.info + inst * (int64_t)ld * n;
    double *prefix = lds_info + (int64_t)ld * n + (int64_t)tid * ld;  // PREFIX: this ant's running sums (odd stride)
    int32_t *tours = G.tours + inst * nn * 2;  // [ant][step] -> next city (n*n ints used)
    int32_t *best = G.best + (int64_t)inst * n * 2;
    const double INF = (double)0x3f3f3f3f;  // ACS_GTSP.hpp:19
    const double alpha = 0.1;               // :189
    __shared__ double s_L[256];
    __shared__ int32_t s_tour[256];    // the iteration-best tour, staged for the deposit
    __shared__ uint8_t s_valid[256];   // ant k's tour visits every city exactly once
    __shared__ double s_pher0, s_bestL, s_last, s_nowL;
    __shared__ int32_t s_bad, s_nowk, s_stop, s_it;
    if (tid == 0) {
        double tmp = 0;
        for (int32_t i = 0; i < n; i++)
            for (int32_t j = i + 1; j < n; j++) tmp += dist[(int64_t)i * n + j];  // :239-249
        s_pher0 = (double)G.cnt / (tmp * n);
        s_bestL = INF; s_last = INF; s_bad = 0; s_stop = 0; s_it = 0;
    }
    __syncthreads();
    for (int64_t e = tid; e < nn; e += blockDim.x) {
        int32_t i = (int32_t)(e / n), j = (int32_t)(e % n);
        pher[e] = s_pher0;
        double h = 1 / ((i == j ? 0.0 : dist[e]) + 1e-8);  // :211
        h6[e] = wa_powi(h, 6);                               // :118
    }
    __syncthreads();
    for (int64_t e = tid; e < (int64_t)ld * n; e += blockDim.x) info[e] = 0.0;  // padding columns stay 0
    __syncthreads();
    const int32_t max_it = G.max_iterations > 0 ? G.max_iterations : n * n;  // :216
    int32_t rf = 0, rb = 0;
    int32_t rr[31];
    if (G.rng_mode == 0 && tid == 0) {
        for (int i = 0; i < 31; i++) rr[i] = G.rng->r[i];
        rf = G.rng->f;
        rb = G.rng->b;
    }
    const int32_t ncol = (n + 15) & ~15;  // city loops run over whole 16-column batches (padding is masked out)
    const int32_t k = tid;  // lanes = ants.  (Fewer ants per wavefront does not help: every ant's n x n
                            // ordered fp64 work is serial inside its lane whatever the other lanes do.)
    for (int32_t it = 0; it < max_it; it++) {
        if (tid == 0) s_stop = s_bad > n ? 1 : 0;  // :263
        __syncthreads();
        if (s_stop) break;
        for (int64_t e = tid; e < nn; e += blockDim.x) {  // reset :114-119
            int32_t i = (int32_t)(e / n), j = (int32_t)(e % n);
            info[(int64_t)i * ld + j] = wa_powi(pher[e], 1) * h6[e];
        }
        if (G.rng_mode == 0 && tid == 0)  // the libc draws of this iteration in (step, ant) order
            for (int32_t q = 0; q < n * (n - 1); q++) G.rbuf[q] = wa_glibc_next(rr, rf, rb);
        __syncthreads();
        if (k < n) {  // construct_solution :146-159 for ant k
            const uint64_t antkey = wa_ctr_antkey(wa_ctr_key(G.seed, G.stream0 + (uint32_t)inst, (uint32_t)it), (uint32_t)k);
            unsigned long long J[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const int32_t cnt = n - w * 64;
                J[w] = cnt >= 64 ? ~0ULL : (cnt > 0 ? ((1ULL << cnt) - 1ULL) : 0ULL);
            }
#pragma unroll
            for (int w = 0; w < NW; w++)
                if ((k >> 6) == w) J[w] &= ~(1ULL << (k & 63));  // J[i].erase(r1[i]) :111
            int32_t r = k, left = n - 1;
            for (int32_t step = 0; step < n; step++) {
                int32_t next = k;  // r1[k]
                if (left > 0) {    // select_next :122-144
                    const int32_t rv = G.rng_mode == 0 ? G.rbuf[step * n + k] : (int32_t)wa_ctr_draw(antkey, (uint32_t)step);
                    double rnd = (double)rv / (double)2147483647;
                    const double *row = info + (int64_t)r * ld;
                    double sum = 0;
                    for (int32_t c0 = 0; c0 < ncol; c0 += 16) {  // 16 cities per batch: 16 independent LDS reads
                        const uint32_t m16 = (uint32_t)(J[NW == 1 ? 0 : (c0 >> 6)] >> (c0 & 63));
#pragma unroll
                        for (int32_t i = 0; i < 16; i++) {
                            sum = __builtin_fma(row[c0 + i], wa_bit_as_double(m16, i), sum);
                            if (PREFIX) prefix[c0 + i] = sum;
                        }
                    }
                    rnd *= sum;
                    if (PREFIX) {
                        // sum_prob of the second pass (:135-141) takes exactly the values prefix[c]: find the
                        // first c with prefix[c] >= rnd (non-decreasing: terms are >= 0), then the first
                        // unvisited city at or after it (prefix is flat over visited cities)
                        int32_t idx = 0;  // columns >= ncol were never written this step: treat them as +inf
#pragma unroll
                        for (int32_t half = 32 * NW; half > 0; half >>= 1)
                            if (idx + half - 1 < ncol && prefix[idx + half - 1] < rnd) idx += half;
                        if (idx < ncol && prefix[idx] >= rnd) {
#pragma unroll
                            for (int w = NW - 1; w >= 0; w--) {
                                const int32_t lo = idx - w * 64;
                                const unsigned long long m = lo >= 64 ? 0ULL : (lo <= 0 ? J[w] : (J[w] & (~0ULL << lo)));
                                if (m) next = w * 64 + (__ffsll((long long)m) - 1);
                            }
                        }
                    } else {
                        double sp = 0;
                        bool found = false;
                        for (int32_t c0 = 0; c0 < ncol; c0 += 16) {
                            const uint32_t m16 = (uint32_t)(J[NW == 1 ? 0 : (c0 >> 6)] >> (c0 & 63));
#pragma unroll
                            for (int32_t i = 0; i < 16; i++) {
                                sp = __builtin_fma(row[c0 + i], wa_bit_as_double(m16, i), sp);
                                if (!found && ((m16 >> i) & 1u) && sp >= rnd) { next = c0 + i; found = true; }
                            }
                        }
                    }
                }
#pragma unroll
                for (int w = 0; w < NW; w++)
                    if ((next >> 6) == w) {
                        const unsigned long long bit = 1ULL << (next & 63);
                        if (J[w] & bit) { J[w] &= ~bit; left--; }  // J[k].erase(next) :153
                    }
                tours[(int64_t)k * n + step] = next;
                r = next;
            }
            // tour length (calc :36-44) after the walk: the same in-order fp64 sum, but its distance loads no longer
            // sit one global round trip deep inside every construction step
            double L = 0;
            r = k;
            for (int32_t step = 0; step < n - 1; step++) {
                const int32_t nx2 = tours[(int64_t)k * n + step];
                L += r == nx2 ? 0.0 : dist[(int64_t)r * n + nx2];
                r = nx2;
            }
            s_L[k] = L;
            s_valid[k] = left == 0 ? 1 : 0;
        }
        __syncthreads();
        if (tid == 0) {  // update_pheromone :163-174: first strictly smallest tour
            double nowL = INF;
            int32_t nowk = -1;
            for (int32_t a = 0; a < n; a++)
                if (s_L[a] < nowL) { nowL = s_L[a]; nowk = a; }
            s_nowL = nowL;
            s_nowk = nowk;
        }
        __syncthreads();
        const int32_t nowk = s_nowk;
        if (nowk >= 0 && s_nowL < s_bestL)  // best = now_best :171-174, stored as (r, s) edges
            for (int32_t e = tid; e < n; e += blockDim.x) {
                best[2 * e] = e == 0 ? nowk : tours[(int64_t)nowk * n + e - 1];
                best[2 * e + 1] = tours[(int64_t)nowk * n + e];
            }
        for (int64_t e = tid; e < nn; e += blockDim.x) pher[e] *= (1 - alpha);  // :175-177
        if (nowk >= 0)
            for (int32_t e = tid; e < n; e += blockDim.x) s_tour[e] = tours[(int64_t)nowk * n + e];
        __syncthreads();
        // deposit :179-184 on all n edges incl. the closing one.  A tour that visits every city once has n distinct
        // undirected edges (n >= 3), each receiving exactly one add: order-free, so one lane per edge.  A degenerate
        // tour (a step found no candidate and fell back to the start city) may repeat an edge: sequential as written.
        const bool par_deposit = nowk >= 0 && n >= 3 && s_valid[nowk];
        if (par_deposit) {
            for (int32_t e = tid; e < n; e += blockDim.x) {
                const int32_t a = e == 0 ? nowk : s_tour[e - 1], b = s_tour[e];
                const double pv = pher[(int64_t)a * n + b] + 1. / (double)s_nowL;
                pher[(int64_t)a * n + b] = pv;
                pher[(int64_t)b * n + a] = pv;
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (nowk >= 0) {
                if (s_nowL < s_bestL) s_bestL = s_nowL;
                if (!par_deposit) {
                    int32_t a = nowk;
                    for (int32_t e = 0; e < n; e++) {
                        const int32_t b = s_tour[e];
                        pher[(int64_t)a * n + b] += 1. / (double)s_nowL;
                        pher[(int64_t)b * n + a] = pher[(int64_t)a * n + b];
                        a = b;
                    }
                }
            }
            if (s_last > s_bestL) { s_last = s_bestL; s_bad = 0; }
            else s_bad++;
            s_it = it + 1;
        }
        __syncthreads();
    }
    if (tid == 0) {
        G.out_cost[inst] = s_bestL;
        G.out_iters[inst] = s_it;
        if (G.rng_mode == 0) {
            for (int i = 0; i < 31; i++) G.rng->r[i] = rr[i];
            G.rng->f = rf;
            G.rng->b = rb;
        }
    }
}

// ------------------------------------------------------------------ wave-per-ant path (default for n <= 256)
// The lanes-as-ants kernel above leaves every ant's n^2 ordered fp64 additions in one lane.  Here one WAVEFRONT
// builds one ant's tour with the lanes spread over the CITIES (NC = 8 or 16 consecutive cities per lane), so
// the roulette of a step is one ordered prefix chain across the wave (the sum must be taken in ascending city
// order: whole-wave DPP, lane l continues from lane l-1's partial sum) followed by a ballot for the first city
// whose prefix reaches the draw.  Ants of an iteration are independent => grid = (ants, instances); the
// iteration barrier is the kernel boundary: construct -> update, two launches per iteration, the host polls the
// stagnation stop (:263) every few iterations.
struct WaGtspState {       // per instance, device memory
    double pher0, bestL, last;
    int32_t bad, it, stop, pad_;
};
struct WaGtspWave {
    WaGtspDev G;
    WaGtspState *state;    // [inst]
    uint8_t *valid;        // [inst][n]  ant's tour visits every city exactly once
};

__device__ __forceinline__ double wa_dpp_below_f64(double x)   // lane l <- lane l-1, lane 0 <- +0.0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// ordered sum over city index: a[q] = term of city lane*NC+q (0.0 where masked: x + 0.0 == x keeps the partial sums
// exact); returns in t[q] the running sum up to and including that city.  After round i lanes 0..i hold their final
// values (lane l restarts from lane l-1's last partial sum each round), so `rounds` = number of lanes that own a city.
template <int NC>
__device__ __forceinline__ void wa_city_prefix(const double (&a)[NC], double (&t)[NC], int rounds)
{
    double carry = 0.0;
#pragma unroll 2
    for (int round = 0; round < rounds; round++) {
        double s = carry;
#pragma unroll
        for (int q = 0; q < NC; q++) { s = s + a[q]; t[q] = s; }
        carry = wa_dpp_below_f64(s);
    }
}

__global__ __launch_bounds__(256) void k_gtspw_init(WaGtspWave W)
{
    const WaGtspDev &G = W.G;
    const int32_t inst = blockIdx.x, tid = threadIdx.x, n = G.n;
    const int64_t nn = (int64_t)n * n;
    const double *dist = G.dist + inst * nn;
    double *pher = G.pher + inst * nn, *h6 = G.h6 + inst * nn, *info = G.info + inst * nn;
    __shared__ double s_pher0;
    if (tid == 0) {
        double tmp = 0;
        for (int32_t i = 0; i < n; i++)
            for (int32_t j = i + 1; j < n; j++) tmp += dist[(int64_t)i * n + j];  // :239-249
        s_pher0 = (double)G.cnt / (tmp * n);
        WaGtspState st;
        st.pher0 = s_pher0; st.bestL = (double)0x3f3f3f3f; st.last = (double)0x3f3f3f3f;
        st.bad = 0; st.it = 0; st.stop = 0; st.pad_ = 0;
        W.state[inst] = st;
        if (G.rng_mode == 0) {   // the libc draws of iteration 0 in (step, ant) order
            int32_t rr[31];
            for (int i = 0; i < 31; i++) rr[i] = G.rng->r[i];
            int32_t rf = G.rng->f, rb = G.rng->b;
            for (int32_t q = 0; q < n * (n - 1); q++) G.rbuf[q] = wa_glibc_next(rr, rf, rb);
            for (int i = 0; i < 31; i++) G.rng->r[i] = rr[i];
            G.rng->f = rf;
            G.rng->b = rb;
        }
    }
    __syncthreads();
    for (int64_t e = tid; e < nn; e += blockDim.x) {
        const int32_t i = (int32_t)(e / n), j = (int32_t)(e % n);
        pher[e] = s_pher0;
        const double h = 1 / ((i == j ? 0.0 : dist[e]) + 1e-8);  // :211
        const double hh = wa_powi(h, 6);                           // :118
        h6[e] = hh;
        info[e] = wa_powi(s_pher0, 1) * hh;                        // reset :114-119 for iteration 0
    }
}

// one wavefront = ant k of instance inst, iteration `it` (construct_solution :146-159 + ACS_Tour::calc :36-44)
template <int NC, bool STAGE>
__global__ __launch_bounds__(64) void k_gtspw_construct(WaGtspWave W, int32_t it)
{
    extern __shared__ double lds_info[];   // STAGE: the whole info matrix (n*n doubles)
    __shared__ int32_t s_tour[256];
    const WaGtspDev &G = W.G;
    const int32_t k = blockIdx.x, inst = blockIdx.y, lane = threadIdx.x, n = G.n;
    if (W.state[inst].stop) return;
    const int64_t nn = (int64_t)n * n;
    const double *dist = G.dist + inst * nn;
    const double *info = G.info + inst * nn;
    if (STAGE) {
        for (int64_t e = lane; e < nn; e += 64) lds_info[e] = info[e];
        __builtin_amdgcn_wave_barrier();
    }
    const double *src = STAGE ? lds_info : info;
    const uint64_t antkey = wa_ctr_antkey(wa_ctr_key(G.seed, G.stream0 + (uint32_t)inst, (uint32_t)it), (uint32_t)k);
    // unvisited set J[k] (:98,:111): every lane keeps the bits of its own NC cities
    const int32_t c0 = lane * NC;
    const int rounds = (n + NC - 1) / NC, last_lane = rounds - 1;
    uint32_t um = 0;   // bit q: city c0+q unvisited
#pragma unroll
    for (int q = 0; q < NC; q++)
        if (c0 + q < n && c0 + q != k) um |= 1u << q;
    int32_t r = k, left = n - 1;
    for (int32_t step = 0; step < n; step++) {
        int32_t next = k;  // r1[k]
        bool picked = false;
        if (left > 0) {    // select_next :122-144
            const int32_t rv = G.rng_mode == 0 ? G.rbuf[step * n + k] : (int32_t)wa_ctr_draw(antkey, (uint32_t)step);
            double rnd = (double)rv / (double)2147483647;
            double a[NC], t[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) a[q] = ((um >> q) & 1u) ? src[(int64_t)r * n + c0 + q] : 0.0;
            wa_city_prefix<NC>(a, t, rounds);
            const double total = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t[NC - 1]), last_lane),
                                                  __builtin_amdgcn_readlane(__double2loint(t[NC - 1]), last_lane));
            rnd *= total;
            int myq = -1;      // first unvisited city of this lane whose running sum reaches the draw (:135-141)
#pragma unroll
            for (int q = NC - 1; q >= 0; q--)
                if (((um >> q) & 1u) && t[q] >= rnd) myq = q;
            const unsigned long long hit = __ballot(myq >= 0);
            if (hit) {
                const int hl = __ffsll((long long)hit) - 1;
                next = hl * NC + __builtin_amdgcn_readlane(myq, hl);
                picked = true;
            }
        }
        if (picked) {      // J[k].erase(next) :153 -- the fallback r1[k] is not in J
            if (lane == next / NC) um &= ~(1u << (next % NC));
            left--;
        }
        if (lane == 0) {
            s_tour[step] = next;
            G.tours[inst * nn * 2 + (int64_t)k * n + step] = next;
        }
        r = next;
    }
    __builtin_amdgcn_wave_barrier();
    // tour length: the same in-order fp64 sum, closing edge excluded (:36-44)
    double a[NC], t[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) {
        const int32_t e = c0 + q;
        double term = 0.0;
        if (e < n - 1) {
            const int32_t ca = e == 0 ? k : s_tour[e - 1], cb = s_tour[e];
            term = ca == cb ? 0.0 : dist[(int64_t)ca * n + cb];
        }
        a[q] = term;
    }
    wa_city_prefix<NC>(a, t, rounds);
    if (lane == last_lane) {
        G.antL[(int64_t)inst * n + k] = t[NC - 1];
        W.valid[(int64_t)inst * n + k] = left == 0 ? 1 : 0;
    }
}

// update_pheromone :161-185, the stagnation bookkeeping :269-275, and the next iteration's reset :114-119
__global__ __launch_bounds__(256) void k_gtspw_update(WaGtspWave W, int32_t it)
{
    const WaGtspDev &G = W.G;
    const int32_t inst = blockIdx.x, tid = threadIdx.x, n = G.n;
    WaGtspState *S = &W.state[inst];
    if (S->stop) return;
    const int64_t nn = (int64_t)n * n;
    double *pher = G.pher + inst * nn, *h6 = G.h6 + inst * nn, *info = G.info + inst * nn;
    const int32_t *tours = G.tours + inst * nn * 2;
    const double *antL = G.antL + (int64_t)inst * n;
    int32_t *best = G.best + (int64_t)inst * n * 2;
    const double INF = (double)0x3f3f3f3f;
    const double alpha = 0.1;
    __shared__ double s_L[256];
    __shared__ int32_t s_tour[256];
    __shared__ double s_nowL, s_bestL;
    __shared__ int32_t s_nowk;
    for (int32_t a = tid; a < n; a += blockDim.x) s_L[a] = antL[a];
    __syncthreads();
    if (tid == 0) {  // first strictly smallest tour
        double nowL = INF;
        int32_t nowk = -1;
        for (int32_t a = 0; a < n; a++)
            if (s_L[a] < nowL) { nowL = s_L[a]; nowk = a; }
        s_nowL = nowL;
        s_nowk = nowk;
        s_bestL = S->bestL;
    }
    __syncthreads();
    const int32_t nowk = s_nowk;
    if (nowk >= 0)
        for (int32_t e = tid; e < n; e += blockDim.x) s_tour[e] = tours[(int64_t)nowk * n + e];
    for (int64_t e = tid; e < nn; e += blockDim.x) pher[e] *= (1 - alpha);  // :175-177
    __syncthreads();
    if (nowk >= 0 && s_nowL < s_bestL)  // best = now_best :171-174, stored as (r, s) edges
        for (int32_t e = tid; e < n; e += blockDim.x) {
            best[2 * e] = e == 0 ? nowk : s_tour[e - 1];
            best[2 * e + 1] = s_tour[e];
        }
    // deposit :179-184 (see k_gtsp_fast: one lane per edge when the tour is a permutation, else as written)
    const bool par_deposit = nowk >= 0 && n >= 3 && W.valid[(int64_t)inst * n + (nowk >= 0 ? nowk : 0)];
    if (par_deposit) {
        for (int32_t e = tid; e < n; e += blockDim.x) {
            const int32_t a = e == 0 ? nowk : s_tour[e - 1], b = s_tour[e];
            const double pv = pher[(int64_t)a * n + b] + 1. / (double)s_nowL;
            pher[(int64_t)a * n + b] = pv;
            pher[(int64_t)b * n + a] = pv;
        }
    } else if (nowk >= 0 && tid == 0) {
        int32_t a = nowk;
        for (int32_t e = 0; e < n; e++) {
            const int32_t b = s_tour[e];
            pher[(int64_t)a * n + b] += 1. / (double)s_nowL;
            pher[(int64_t)b * n + a] = pher[(int64_t)a * n + b];
            a = b;
        }
    }
    __syncthreads();
    for (int64_t e = tid; e < nn; e += blockDim.x) info[e] = wa_powi(pher[e], 1) * h6[e];  // next iteration's reset
    if (tid == 0) {
        WaGtspState st = *S;
        if (nowk >= 0 && s_nowL < st.bestL) st.bestL = s_nowL;
        if (st.last > st.bestL) { st.last = st.bestL; st.bad = 0; }
        else st.bad++;
        st.it = it + 1;
        const int32_t max_it = G.max_iterations > 0 ? G.max_iterations : n * n;  // :216
        st.stop = (st.bad > n || st.it >= max_it) ? 1 : 0;                        // :263
        *S = st;
        G.out_cost[inst] = st.bestL;
        G.out_iters[inst] = st.it;
        if (G.rng_mode == 0 && !st.stop) {   // the libc draws of the next iteration
            int32_t rr[31];
            for (int i = 0; i < 31; i++) rr[i] = G.rng->r[i];
            int32_t rf = G.rng->f, rb = G.rng->b;
            for (int32_t q = 0; q < n * (n - 1); q++) G.rbuf[q] = wa_glibc_next(rr, rf, rb);
            for (int i = 0; i < 31; i++) G.rng->r[i] = rr[i];
            G.rng->f = rf;
            G.rng->b = rb;
        }
    }
}
