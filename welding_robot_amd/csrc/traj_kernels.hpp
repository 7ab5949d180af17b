// traj_kernels.hpp -- what happens to a path right after the search:
//   k_stitch          ACS_GTSP::read_all_segments (ACS_GTSP.hpp:286-298) on node ids, with the per-segment
//                     reversal the reference lacks as an option
//   k_bspline_setup   BS_Basic::SetParam (BSplineBasic.h:72-78): knot chain, constrained control points
//   k_bspline_middle  _CalcCPoints (:458-464)
//   k_bspline_eval    getCurvePoint / getCurveDerPoint (:87-146) for a batch of times: one lane per sample
// The arithmetic is fp32 in the reference's operation order (no contraction, correctly rounded division),
// so results are bit-identical to BS_Basic<float, DIM, DEGREE, CI, CF> on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wa_device.h"

#define WA_BS_MAX_DEGREE 7
#define WA_BS_W (WA_BS_MAX_DEGREE + 1)
#define WA_BS_MAX_DIM 16

struct WaSpline {
    int32_t dim, degree, ci, cf;
    long long n_middle, n_knots, n_cps;
    float *knots;   // n_knots
    float *cps;     // n_cps x dim
    float uninit;   // value of the heap cells the reference reads without writing (c_mat[idx][CL+1], :414-431)
};

// ---------------------------------------------------------------- stitching
__global__ void k_stitch(const long long *__restrict__ ids, const long long *__restrict__ off, int32_t n_seg,
                         const uint8_t *__restrict__ rev, WaDims d, const float *__restrict__ cx,
                         const float *__restrict__ cy, const float *__restrict__ cz, float *__restrict__ out,
                         long long n_total)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    int lo = 0, hi = n_seg;   // segment s with off[s] <= i < off[s+1]; empty segments are skipped by construction
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    long long a = off[lo], e = off[lo + 1];
    long long src = (rev && rev[lo]) ? e - 1 - (i - a) : i;
    long long id = ids[src];
    long long z = id / d.nxy, r = id - z * d.nxy;
    long long y = r / d.nx, x = r - y * d.nx;
    out[i * 3 + 0] = cx[x];
    out[i * 3 + 1] = cy[y];
    out[i * 3 + 2] = cz[z];
}

// ---------------------------------------------------------------- BS_Basic pieces
// _findSpan (:358-385)
__host__ __device__ inline bool wa_bs_find_span(const float *K, long long nk, float u, long long *ret)
{
    float last = K[nk - 1];
    if (u < K[0] || last < u) return false;
    float dd = u - last;
    if ((double)(dd * dd) < 1.e-10) {   // SP_IS_EQUAL (:8): fp32 product against a double literal
        for (long long i = nk - 2; i > -1; --i)
            if (K[i] < u && u <= K[i + 1]) { *ret = i; return true; }
        return false;
    }
    long long low = 0, high = nk - 1, mid = (low + high) >> 1;
    int guard = 0;
    while (u < K[mid] || u >= K[mid + 1]) {
        if (u < K[mid]) high = mid; else low = mid;
        mid = (low + high) >> 1;
        if (++guard > 200) return false;   // non-monotone knots: the reference would not terminate
    }
    *ret = mid;
    return true;
}

// _BasisFuns (:330-353)
template <int DEG>
__host__ __device__ __forceinline__ void wa_bs_basis_funs(const float *K, float *N, long long span, float u)
{
    float left = 0.0f, right = 0.0f, saved = 0.0f, temp = 0.0f;
    N[0] = 1.0f;
#pragma unroll
    for (int j = 1; j <= DEG; ++j) {
        saved = 0.0f;
#pragma unroll
        for (int r = 0; r < j; ++r) {
            left = u - K[span + 1 - (j - r)];
            right = K[span + (r + 1)] - u;
            if ((right + left) != 0) temp = N[r] / (right + left);
            N[r] = saved + right * temp;
            saved = left * temp;
        }
        N[j] = saved;
    }
}

// _BasisFunsDers(ders, span, u, n) (:237-323); D is a run-time degree here (setup and derivative paths)
__host__ __device__ inline void wa_bs_basis_ders(const float *K, int D, float uninit, float ders[][WA_BS_W + 2],
                                 long long span, float u, int n)
{
    float ndu[WA_BS_W][WA_BS_W], a[2][WA_BS_W];
    float saved = 0.0f, left = 0.0f, right = 0.0f, temp = 0.0f, d = 0.0f;
    for (int i = 0; i < WA_BS_W; i++)
        for (int j = 0; j < WA_BS_W; j++) ndu[i][j] = uninit;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < WA_BS_W; j++) a[i][j] = uninit;
    ndu[0][0] = 1.0f;
    for (int j = 1; j <= D; ++j) {
        saved = 0.0f;
        for (int r = 0; r < j; ++r) {
            left = u - K[span + 1 - (j - r)];
            right = K[span + (r + 1)] - u;
            ndu[j][r] = right + left;
            temp = ndu[r][j - 1] / ndu[j][r];
            ndu[r][j] = saved + right * temp;
            saved = left * temp;
        }
        ndu[j][j] = saved;
    }
    for (int j = 0; j <= D; ++j) ders[0][j] = ndu[j][D];
    for (int r = 0; r <= D; ++r) {
        int s1 = 0, s2 = 1;
        a[0][0] = 1.0f;
        for (int k = 1; k <= n; ++k) {
            d = 0.0f;
            int rk = r - k, pk = D - k, j1, j2;
            if (r >= k) {
                a[s2][0] = a[s1][0] / ndu[pk + 1][rk];
                d = a[s2][0] * ndu[rk][pk];
            }
            j1 = (rk >= -1) ? 1 : -rk;
            j2 = (r - 1 <= pk) ? k - 1 : D - r;
            for (int j = j1; j <= j2; ++j) {
                a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pk + 1][rk + j];
                d += a[s2][j] * ndu[rk + j][pk];
            }
            if (r <= pk) {
                a[s2][k] = -a[s1][k - 1] / ndu[pk + 1][r];
                d += a[s2][k] * ndu[r][pk];
            }
            ders[k][r] = d;
            int t = s1; s1 = s2; s2 = t;
        }
    }
    int r = D;
    for (int k = 1; k <= n; ++k) {
        for (int j = 0; j <= D; ++j) ders[k][j] *= (float)r;
        r *= (D - k);
    }
}

// SetParam (:72-78) = _CalcKnot (:150-171) + _CalcConstrainedCPoints (:387-447), one wavefront.  The knot chain
// K[i] = K[i-1] + step is a sequential fp32 recurrence (its rounding is part of the result: it drifts from i*step).
// ends = init rows then fin rows, (level+1) x dim each.
__global__ void __launch_bounds__(64) k_bspline_setup(WaSpline S, const float *__restrict__ ends, float fin_time)
{
    const int D = S.degree, dim = S.dim;
    float *K = S.knots, *C = S.cps;
    const float *init = ends, *fin = ends + (size_t)(S.ci + 1) * dim;
    {
        // every lane runs the same recurrence; lane l keeps the value of step 64*c + l, so the chain costs
        // three VALU operations per knot and the stores are one coalesced 256-B row per 64 knots
        const int lane = threadIdx.x;
        const long long nmid = S.n_knots - 2 * D - 2;
        const float step = fin_time / (float)(nmid + 1);
        float k = 0.0f;
        for (long long base = 0; base < nmid; base += 64) {
            float mine = 0.0f;
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                k = k + step;
                mine = (lane == q) ? k : mine;
            }
            if (base + lane < nmid) K[D + 1 + base + lane] = mine;
        }
        if (lane < D + 1) {
            K[lane] = 0.0f;
            K[D + 1 + nmid + lane] = fin_time;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence();
    for (int m = 0; m < dim; ++m) {
        C[m] = init[m];
        C[(S.n_cps - 1) * dim + m] = fin[m];
    }
    float mat[WA_BS_W][WA_BS_W + 2];
    long long span;
    for (int i = 0; i < WA_BS_W; i++)
        for (int j = 0; j < WA_BS_W + 2; j++) mat[i][j] = S.uninit;
    if (wa_bs_find_span(K, S.n_knots, 0.0f, &span)) wa_bs_basis_ders(K, D, S.uninit, mat, span, 0.0f, S.ci);
    for (int j = 1; j < S.ci + 1; ++j)
        for (int k = 0; k < dim; ++k) {
            float v = init[j * dim + k];
            for (int h = j; h > 0; --h) v -= mat[j][h - 1] * C[(h - 1) * dim + k];
            C[j * dim + k] = v / mat[j][j];
        }
    for (int i = 0; i < WA_BS_W; i++)
        for (int j = 0; j < WA_BS_W + 2; j++) mat[i][j] = S.uninit;
    if (wa_bs_find_span(K, S.n_knots, fin_time, &span)) wa_bs_basis_ders(K, D, S.uninit, mat, span, fin_time, S.cf);
    int idx = 1;
    for (long long j = S.n_cps - 2; j > S.n_cps - 2 - S.cf; --j) {
        for (int k = 0; k < dim; ++k) {
            float v = fin[idx * dim + k];
            for (int h = idx; h > 0; --h) v -= mat[idx][S.cf + 2 - h] * C[(S.n_cps - h) * dim + k];
            C[j * dim + k] = v / mat[idx][S.cf + 1 - idx];
        }
        ++idx;
    }
}

// _CalcCPoints (:458-464): control point ci+1+i <- first dim floats of middle row i
__global__ void k_bspline_middle(WaSpline S, const float *__restrict__ middle, long long stride)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = S.n_middle * S.dim;
    if (t >= total) return;
    long long i = t / S.dim;
    int m = (int)(t - i * S.dim);
    S.cps[(S.ci + 1 + i) * S.dim + m] = middle[i * stride + m];
}

// getCurvePoint (:87-112) / getCurveDerPoint (:122-146) for ONE time u: dim values to `o`; false (and zeros) where the reference would
// refuse (span search fails, d > DEGREE).  One definition for the kernel (one lane per time) and for the host path of single-point
// calls (wa_bspline_eval_host: the same source compiled for the host with -ffp-contract=off, so the same fp32 operations in the same order).
template <int DEG>
__host__ __device__ __forceinline__ bool wa_bs_eval_one(const WaSpline &S, const float *__restrict__ K, const float *__restrict__ C,
                                                        float u, int32_t der, float *__restrict__ o)
{
    const int dim = S.dim;
    float lastk = K[S.n_knots - 1];
    if (u < K[0]) u = K[0];
    else if (u > lastk) u = lastk;
    long long span = 0;
    bool good = der <= DEG && wa_bs_find_span(K, S.n_knots, u, &span);
    good = good && span - DEG >= 0 && span < S.n_cps && span + DEG < S.n_knots;
    if (!good) {
        for (int m = 0; m < dim; ++m) o[m] = 0.0f;
        return false;
    }
    if (der == 0) {
        float N[DEG + 1];
        wa_bs_basis_funs<DEG>(K, N, span, u);
        const float *c0 = C + (span - DEG) * dim;
        for (int j = 0; j < dim; ++j) {
            float c = 0.0f;
#pragma unroll
            for (int q = 0; q <= DEG; ++q) c += N[q] * c0[q * dim + j];
            o[j] = c;
        }
    } else {
        float nd[WA_BS_W][WA_BS_W + 2];
        wa_bs_basis_ders(K, DEG, S.uninit, nd, span, u, der);
        const float *c0 = C + (span - DEG) * dim;
        for (int m = 0; m < dim; ++m) {
            float c = 0.0f;
            for (int j = 0; j <= DEG; ++j) c += nd[der][j] * c0[j * dim + m];
            o[m] = c;
        }
    }
    return true;
}

// one lane per time.  us == nullptr: u = t0 + i*dt.
template <int DEG>
__global__ void k_bspline_eval(WaSpline S, const float *__restrict__ us, float t0, float dt, long long count,
                               int32_t der, float *__restrict__ out, uint8_t *__restrict__ ok)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float u = us ? us[i] : t0 + (float)i * dt;
    const bool good = wa_bs_eval_one<DEG>(S, S.knots, S.cps, u, der, out + i * S.dim);
    if (ok) ok[i] = good ? 1 : 0;
}
