/* oracle/weld_oracle.c -- TEST INFRASTRUCTURE (see weld_oracle.h for the parity statement).
 *
 * Plain-C restatement of the reference's planning path on flat arrays.  All float
 * arithmetic is written operation-for-operation as the reference evaluates it (fp32,
 * left-to-right, no contraction: built with -ffp-contract=off, no -ffast-math), because
 * the parity bar is bit-exact pheromone fields and voxel-id paths.
 *
 * Lattice conventions shared with the HIP side:
 *   voxel id = (z*ny + y)*nx + x            (model_grid_map.hpp:203-216 raster counter)
 *   edge k of a voxel: 0:z-1 1:y-1 2:x-1 3:x+1 4:y+1 5:z+1   (ACSRank_3D.hpp:355-365 push order)
 *   pher[id*6+k]; out-of-bounds edges start at 0 (:396) and become pheromone_0 at reset (:313)
 */
#define _POSIX_C_SOURCE 200809L
#include "weld_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ================================================================== RNG ============ */
/* glibc stdlib/random_r.c, TYPE_3 (x^31 + x^3 + 1), the generator behind rand()/srand()
 * that ACSRank_3D.hpp:169,327 and ACS_GTSP.hpp:126 use.  Not in /root/reference: libc
 * 2.35 (Ubuntu 22.04); pinned by SURVEY KA5 and against the live libc in the tests. */
int32_t wo_rand(wo_glibc_rand *s)
{
    uint32_t v = (uint32_t)s->r[s->f] + (uint32_t)s->r[s->b];
    s->r[s->f] = (int32_t)v;
    if (++s->f >= 31) { s->f = 0; ++s->b; }
    else if (++s->b >= 31) s->b = 0;
    s->calls++;
    return (int32_t)((v >> 1) & 0x7fffffffu);
}

void wo_srand(wo_glibc_rand *s, uint32_t seed)
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
    for (int i = 0; i < 310; i++) (void)wo_rand(s);
    s->calls = 0;
}

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t wo_splitmix64(uint64_t *state)
{
    *state += 0x9E3779B97F4A7C15ULL;
    return mix64(*state);
}

/* DEV-mode random integer: a pure function of (seed, stream, generation, ant, step), so
 * every ant of every problem can walk in parallel.  Same formula as csrc/wa_device.h. */
static uint64_t ctr_antkey(uint64_t seed, uint32_t stream, uint32_t gen, uint32_t ant)
{
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * (((uint64_t)stream << 32) | gen));
    return mix64(k + 0x9E3779B97F4A7C15ULL * ((uint64_t)ant + 1));
}
static uint32_t ctr_draw(uint64_t antkey, uint32_t step)
{
    uint32_t x = ((uint32_t)antkey + step * 0x9E3779B9u) ^ (uint32_t)(antkey >> 32);
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x >> 1;
}
uint32_t wo_ctr_rand31(uint64_t seed, uint32_t stream, uint32_t gen, uint32_t ant, uint32_t step)
{
    return ctr_draw(ctr_antkey(seed, stream, gen, ant), step);
}

/* ================================================================== std::sort ====== */
/* libstdc++ (GCC 11) bits/stl_algo.h introsort + bits/stl_heap.h, restated on (key, tag)
 * records compared by key only -- reproduces the permutation std::sort gives the reference
 * at ACSRank_3D.hpp:273 when path lengths tie.  Pinned against the real
 * std::sort through `ref_harness sort`. */
typedef struct { float k; int32_t t; } srec;
#define LT(a, b) ((a).k < (b).k)

static void ss_push_heap(srec *first, long hole, long top, srec value)
{
    long parent = (hole - 1) / 2;
    while (hole > top && LT(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void ss_adjust_heap(srec *first, long hole, long len, srec value)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (LT(first[child], first[child - 1])) child--;
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
static void ss_heap_sort(srec *first, srec *last)
{ /* __partial_sort(first, last, last) = make_heap + sort_heap */
    long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            srec v = first[parent];
            ss_adjust_heap(first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        srec v = *last;
        *last = *first;
        ss_adjust_heap(first, 0, last - first, v);
    }
}
static void ss_swap(srec *a, srec *b) { srec t = *a; *a = *b; *b = t; }
static void ss_median_to_first(srec *result, srec *a, srec *b, srec *c)
{
    if (LT(*a, *b)) {
        if (LT(*b, *c)) ss_swap(result, b);
        else if (LT(*a, *c)) ss_swap(result, c);
        else ss_swap(result, a);
    } else if (LT(*a, *c)) ss_swap(result, a);
    else if (LT(*b, *c)) ss_swap(result, c);
    else ss_swap(result, b);
}
static srec *ss_partition(srec *first, srec *last, srec *pivot)
{
    for (;;) {
        while (LT(*first, *pivot)) ++first;
        --last;
        while (LT(*pivot, *last)) --last;
        if (!(first < last)) return first;
        ss_swap(first, last);
        ++first;
    }
}
static void ss_introsort_loop(srec *first, srec *last, long depth)
{
    while (last - first > 16) {
        if (depth == 0) { ss_heap_sort(first, last); return; }
        --depth;
        srec *mid = first + (last - first) / 2;
        ss_median_to_first(first, first + 1, mid, last - 1);
        srec *cut = ss_partition(first + 1, last, first);
        ss_introsort_loop(cut, last, depth);
        last = cut;
    }
}
static void ss_unguarded_linear_insert(srec *last)
{
    srec v = *last;
    srec *next = last - 1;
    while (LT(v, *next)) { *last = *next; last = next; --next; }
    *last = v;
}
static void ss_insertion_sort(srec *first, srec *last)
{
    if (first == last) return;
    for (srec *i = first + 1; i != last; ++i) {
        if (LT(*i, *first)) {
            srec v = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(srec));
            *first = v;
        } else ss_unguarded_linear_insert(i);
    }
}
void wo_std_sort_perm(const float *keys, int32_t n, int32_t *perm)
{
    if (n <= 0) return;
    srec *v = (srec *)malloc((size_t)n * sizeof(srec));
    for (int32_t i = 0; i < n; i++) { v[i].k = keys[i]; v[i].t = i; }
    long lg = 0;
    for (unsigned long m = (unsigned long)n; m > 1; m >>= 1) lg++;
    ss_introsort_loop(v, v + n, 2 * lg);
    if (n > 16) {
        ss_insertion_sort(v, v + 16);
        for (srec *i = v + 16; i != v + n; ++i) ss_unguarded_linear_insert(i);
    } else ss_insertion_sort(v, v + n);
    for (int32_t i = 0; i < n; i++) perm[i] = v[i].t;
    free(v);
}

/* DEV rank rule: ascending L, ties by ant index (what csrc/acs_rank.hip computes by counting) */
void wo_stable_rank_perm(const float *keys, int32_t n, int32_t *perm)
{
    for (int32_t a = 0; a < n; a++) {
        int32_t r = 0;
        for (int32_t b = 0; b < n; b++)
            if (keys[b] < keys[a] || (keys[b] == keys[a] && b < a)) r++;
        perm[r] = a;
    }
}

/* ================================================================== STL ============ */
/* read_STL.hpp:65-72 format sniff on byte 79, :131-156 binary layout, :99-129 the ASCII branch.  The reference
 * exit()s on I/O errors (:34-59); the restatement returns negative codes instead.
 *
 * ASCII (:99-129): the reference wraps the file in a std::stringstream (the text up to the first NUL byte) and reads
 *   ss >> name >> name; ss.get();
 *   loop { ss >> w; if (w != "facet") break; getline x 2; 3 x (ss >> w >> x >> y >> z); push; getline x 3; }
 * with ONE Triangles object declared outside the loop: the "facet normal ..." line is skipped, so every normal stays (0, 0, 0)
 * (plane distance 0 => every voxel of a triangle's bounding box +- p is occupied), and a read that fails leaves the
 * previous triangle's values in place.  Restated with the stream semantics of libstdc++ (GCC 11): sentry / skipws, eofbit and
 * failbit, operator>>(string) leaving its target alone when the sentry fails, num_get's character filter (bits/locale_facets.tcc
 * _M_extract_float, "C" locale) in front of strtof, value 0 + failbit on a conversion error, +-FLT_MAX + failbit on overflow.
 * One input the reference never returns from -- the text ends directly behind a "facet" token, so that every later read fails
 * with "facet" still in the string: an endless push_back -- is reported as -10. */
typedef struct { const uint8_t *s; size_t n, pos; int eof, fail; } wo_is;
static int wo_is_space(int c) { return c == ' ' || (c >= 9 && c <= 13); }
static int wo_is_sentry(wo_is *t, int noskipws)
{
    int good = !t->eof && !t->fail;
    if (good && !noskipws) {
        while (t->pos < t->n && wo_is_space(t->s[t->pos])) t->pos++;
        if (t->pos >= t->n) { t->eof = 1; good = 0; }
    }
    if (good) return 1;
    t->fail = 1;
    return 0;
}
/* operator>>(istream&, string&): 1 = the string was assigned [*tok, *tok + *len) */
static int wo_is_token(wo_is *t, const uint8_t **tok, size_t *len)
{
    if (!wo_is_sentry(t, 0)) return 0;
    const size_t a = t->pos;
    while (t->pos < t->n && !wo_is_space(t->s[t->pos])) t->pos++;
    if (t->pos >= t->n) t->eof = 1;
    *tok = t->s + a;
    *len = t->pos - a;
    if (*len == 0) t->fail = 1;
    return 1;
}
/* std::getline(istream&, string&): 1 = the string was assigned */
static int wo_is_getline(wo_is *t, const uint8_t **line, size_t *len)
{
    if (!wo_is_sentry(t, 1)) return 0;
    const size_t a = t->pos;
    size_t extracted = 0;
    while (t->pos < t->n && t->s[t->pos] != '\n') { t->pos++; extracted++; }
    *line = t->s + a;
    *len = t->pos - a;
    if (t->pos >= t->n) t->eof = 1;
    else { t->pos++; extracted++; }
    if (!extracted) t->fail = 1;
    return 1;
}
static void wo_is_get(wo_is *t)
{
    if (!wo_is_sentry(t, 1)) return;
    if (t->pos < t->n) t->pos++;
    else { t->eof = 1; t->fail = 1; }
}
/* operator>>(istream&, float&) */
static void wo_is_float(wo_is *t, float *v)
{
    if (!wo_is_sentry(t, 0)) return;
    const size_t cap = t->n - t->pos + 4;            /* the filter passes at most what is left of the text (+ sign, '0', terminator) */
    char *x = (char *)malloc(cap);
    if (!x) { t->fail = 1; return; }
    size_t k = 0;
    int eof = t->pos >= t->n, mant = 0, dec = 0, sci = 0;
    int c = eof ? 0 : t->s[t->pos];
#define WO_NEXT() do { if (++t->pos < t->n) c = t->s[t->pos]; else eof = 1; } while (0)
#define WO_PUT(ch) do { if (k + 1 < cap) x[k++] = (char)(ch); } while (0)
    if (!eof && (c == '+' || c == '-')) { WO_PUT(c); WO_NEXT(); }
    while (!eof && c == '0') {              /* leading zeros collapse into one */
        if (!mant) { WO_PUT('0'); mant = 1; }
        WO_NEXT();
    }
    while (!eof) {
        if (c >= '0' && c <= '9') { WO_PUT(c); mant = 1; }
        else if (c == '.' && !dec && !sci) { WO_PUT('.'); dec = 1; }
        else if ((c == 'e' || c == 'E') && !sci && mant) {
            WO_PUT('e');
            sci = 1;
            if (++t->pos < t->n) {
                c = t->s[t->pos];
                if (c == '+' || c == '-') WO_PUT(c);
                else continue;
            } else { eof = 1; break; }
        } else break;
        WO_NEXT();
    }
#undef WO_NEXT
#undef WO_PUT
    x[k] = 0;
    char *end = x;
    float r = strtof(x, &end);
    if (end == x || *end != 0) { r = 0.f; t->fail = 1; }
    else if (r == INFINITY) { r = FLT_MAX; t->fail = 1; }
    else if (r == -INFINITY) { r = -FLT_MAX; t->fail = 1; }
    *v = r;
    free(x);
    if (eof) t->eof = 1;
}
static int wo_tok_is(const uint8_t *p, size_t len, const char *w) { return len == strlen(w) && memcmp(p, w, len) == 0; }
/* tris == NULL: count only */
static int64_t wo_stl_ascii(const uint8_t *buf, size_t len, float *tris)
{
    wo_is t = {buf, 0, 0, 0, 0};
    while (t.n < len && buf[t.n] != 0) t.n++;      /* std::stringstream ss(buffer): a C string */
    const uint8_t *w = (const uint8_t *)"";
    size_t wl = 0;
    const uint8_t *q;
    size_t ql;
    if (wo_is_token(&t, &q, &ql)) { w = q; wl = ql; }
    if (wo_is_token(&t, &q, &ql)) { w = q; wl = ql; }   /* (into `name`, not `useless`: kept apart below) */
    wo_is_get(&t);
    w = (const uint8_t *)"";
    wl = 0;
    float tri[12];
    memset(tri, 0, sizeof tri);
    int64_t n = 0;
    for (;;) {
        const int got = wo_is_token(&t, &q, &ql);
        if (got) { w = q; wl = ql; }
        if (!wo_tok_is(w, wl, "facet")) break;
        if (!got) return -10;                              /* nothing will ever change `useless` again: the reference loops forever */
        if (wo_is_getline(&t, &q, &ql)) { w = q; wl = ql; }
        if (wo_is_getline(&t, &q, &ql)) { w = q; wl = ql; }
        for (int i = 0; i < 3; i++) {
            if (wo_is_token(&t, &q, &ql)) { w = q; wl = ql; }
            wo_is_float(&t, &tri[3 + 3 * i]);
            wo_is_float(&t, &tri[4 + 3 * i]);
            wo_is_float(&t, &tri[5 + 3 * i]);
        }
        if (tris) memcpy(tris + n * 12, tri, sizeof tri);
        n++;
        if (wo_is_getline(&t, &q, &ql)) { w = q; wl = ql; }
        if (wo_is_getline(&t, &q, &ql)) { w = q; wl = ql; }
        if (wo_is_getline(&t, &q, &ql)) { w = q; wl = ql; }
    }
    return n;
}
int64_t wo_stl_count(const uint8_t *buf, size_t len)
{
    if (len < 80) return -3;
    if (buf[79] != 0) return wo_stl_ascii(buf, len, NULL);   /* :65-68 */
    if (len < 84) return -3;
    int32_t n;
    memcpy(&n, buf + 80, 4);
    if (n < 0 || (size_t)n * 50 + 84 > len) return -3;
    return n;
}
int64_t wo_stl_parse(const uint8_t *buf, size_t len, float *tris)
{
    int64_t n = wo_stl_count(buf, len);
    if (n < 0) return n;
    if (buf[79] != 0) return wo_stl_ascii(buf, len, tris);
    const uint8_t *p = buf + 84;
    for (int64_t i = 0; i < n; i++) {
        memcpy(tris + i * 12, p, 48); /* normal, v0, v1, v2 (:142-150) */
        p += 50;                      /* + 2-byte attribute (:151) */
    }
    return n;
}

/* ================================================================== grid =========== */
/* model_grid_map.hpp:165-181 bbox, :198-200 ranges */
void wo_grid_dims(const float *tris, int64_t ntris, float precision, int32_t wall,
                  int32_t dims[3], float bbox[6])
{
    float mn[3] = {tris[3], tris[4], tris[5]}, mx[3] = {tris[3], tris[4], tris[5]};
    for (int64_t t = 0; t < ntris; t++)
        for (int v = 0; v < 3; v++)
            for (int c = 0; c < 3; c++) {
                float q = tris[t * 12 + 3 + v * 3 + c];
                mx[c] = q > mx[c] ? q : mx[c];
                mn[c] = q < mn[c] ? q : mn[c];
            }
    for (int c = 0; c < 3; c++) {
        dims[c] = (int)((mx[c] - mn[c]) / precision) + 1 + 2 * wall;
        bbox[c] = mn[c];
        bbox[3 + c] = mx[c];
    }
}

/* model_grid_map.hpp:204-211 (and :321-328): piecewise node coordinate along one axis */
void wo_axis_coords(float lo, float hi, float precision, int32_t wall, int32_t n, float *out)
{
    for (int32_t i = 0; i < n; i++)
        out[i] = i < wall ? lo - (float)(wall - i) * precision
                          : (i >= (n - wall) ? hi + (float)(i - n + wall) * precision
                                             : lo + (float)(i - wall) * precision);
}

/* model_grid_map.hpp:223-268 -- every triangle x every voxel */
void wo_voxelize(const float *tris, int64_t ntris, float precision, int32_t nx, int32_t ny,
                 int32_t nz, const float *cx, const float *cy, const float *cz, uint8_t *free_out)
{
    size_t n = (size_t)nx * ny * nz;
    memset(free_out, 1, n);
    const double thr = 1.2 * precision; /* :256 double compare */
    for (int64_t t = 0; t < ntris; t++) {
        const float *T = tris + t * 12;
        const float nxn = T[0], nyn = T[1], nzn = T[2];
        float D = -(T[3] * nxn + T[4] * nyn + T[5] * nzn); /* :224-226 */
        float mn[3] = {T[3], T[4], T[5]}, mx[3] = {T[3], T[4], T[5]};
        for (int v = 0; v < 3; v++)
            for (int c = 0; c < 3; c++) {
                float q = T[3 + v * 3 + c];
                mx[c] = q > mx[c] ? q : mx[c];
                mn[c] = q < mn[c] ? q : mn[c];
            }
        for (int c = 0; c < 3; c++) { mn[c] -= precision; mx[c] += precision; } /* :243-248 */
        for (int32_t z = 0; z < nz; z++)
            for (int32_t y = 0; y < ny; y++)
                for (int32_t x = 0; x < nx; x++) {
                    float px = cx[x], py = cy[y], pz = cz[z];
                    float dist = px * nxn + py * nyn + pz * nzn + D; /* :252-254 */
                    float ad = dist > 0 ? dist : -dist;              /* my_abs :23 */
                    if ((double)ad < thr)
                        if (mn[0] <= px && px <= mx[0] && mn[1] <= py && py <= mx[1] &&
                            mn[2] <= pz && pz <= mx[2])
                            free_out[((size_t)z * ny + y) * nx + x] = 0;
                }
    }
}

/* ACSRank_3D.hpp:537-565: no early exit => the LAST free voxel in raster order that lies
 * within t = (float)(1.2*precision) of the point on every axis (Q4) */
int64_t wo_resolve_point(int32_t nx, int32_t ny, int32_t nz, const float *cx, const float *cy,
                         const float *cz, const uint8_t *free_, float precision, const float pt[3])
{
    float t = 1.2 * precision;
    int64_t found = -1;
    for (int32_t z = 0; z < nz; z++) {
        float dz = pt[2] - cz[z];
        if (!((dz > 0 ? dz : -dz) < t)) continue;
        for (int32_t y = 0; y < ny; y++) {
            float dy = pt[1] - cy[y];
            if (!((dy > 0 ? dy : -dy) < t)) continue;
            for (int32_t x = 0; x < nx; x++) {
                float dx = pt[0] - cx[x];
                if (!((dx > 0 ? dx : -dx) < t)) continue;
                int64_t id = ((int64_t)z * ny + y) * nx + x;
                if (free_[id]) found = id;
            }
        }
    }
    return found;
}

int64_t wo_synth_grid(int32_t n, uint64_t seed, double occ_prob, uint8_t *free_out)
{
    uint64_t st = seed;
    size_t tot = (size_t)n * n * n;
    for (size_t i = 0; i < tot; i++) {
        double u = (double)(wo_splitmix64(&st) >> 11) * (1.0 / 9007199254740992.0);
        free_out[i] = u < occ_prob ? 0 : 1;
    }
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < 2; y++)
            for (int x = 0; x < 2; x++) {
                free_out[((size_t)z * n + y) * n + x] = 1;
                free_out[((size_t)(n - 1 - z) * n + (n - 1 - y)) * n + (n - 1 - x)] = 1;
            }
    int64_t fr = 0;
    for (size_t i = 0; i < tot; i++) fr += free_out[i];
    return fr;
}

/* ================================================================== ACS_Rank ======= */
typedef struct {
    int32_t *ids;    /* path voxel ids (Agent::path)        */
    int8_t *choice;  /* edge index taken (Agent::node_index) */
    int64_t len, cap;
    float L;
} wo_ant;

struct wo_acs {
    int32_t nx, ny, nz;
    int64_t n;
    float precision;
    float *cx, *cy, *cz;
    uint8_t *free_;
    int nb;               /* neighbourhood: 6 (the reference as shipped) or 26 (its stubbed variant, :361-388) */
    const int *dx, *dy, *dz;
    float dist[26];       /* step length per edge (:369-385) */
    float *pher;          /* [n][nb] */
    uint32_t *visit;      /* tabu stamps, one array reused by all ants */
    uint32_t visit_stamp;
    uint32_t *bestmark;   /* best-path membership stamps */
    uint32_t best_ver;
    wo_ant best;          /* persists across solves (Q9) */
    int32_t last_colony;
    float last_lambda, last_Q;
    int32_t *last_len;    /* per-ant node count / L of the last generation walked (agents[] of :251) */
    float *last_L;
    int32_t last_cap;
    int32_t *last_ids;    /* their paths, back to back (diagnostics) */
    int64_t last_ids_cap;
};

static void ant_push(wo_ant *a, int32_t id, int8_t ch)
{
    if (a->len == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 256;
        a->ids = (int32_t *)realloc(a->ids, (size_t)a->cap * sizeof(int32_t));
        a->choice = (int8_t *)realloc(a->choice, (size_t)a->cap);
    }
    a->ids[a->len] = id;
    a->choice[a->len] = ch; /* choice[i] = edge taken to ARRIVE at ids[i]; choice[0] unused */
    a->len++;
}

static const int DX[6] = {0, 0, -1, 1, 0, 0}, DY[6] = {0, -1, 0, 0, 1, 0}, DZ[6] = {-1, 0, 0, 0, 0, 1};
/* the 3x3x3 cube around a node in the reference's loop order z (outer), y, x (inner), centre skipped (:352-357) */
static const int DX26[26] = {-1, 0, 1, -1, 0, 1, -1, 0, 1, -1, 0, 1, -1, 1, -1, 0, 1, -1, 0, 1, -1, 0, 1, -1, 0, 1};
static const int DY26[26] = {-1, -1, -1, 0, 0, 0, 1, 1, 1, -1, -1, -1, 0, 0, 1, 1, 1, -1, -1, -1, 0, 0, 0, 1, 1, 1};
static const int DZ26[26] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1};

wo_acs *wo_acs_create(int32_t nx, int32_t ny, int32_t nz, const float *cx, const float *cy,
                      const float *cz, const uint8_t *free_, float precision, float pheromone_0)
{
    return wo_acs_create_nb(nx, ny, nz, cx, cy, cz, free_, precision, pheromone_0, 6);
}

/* nb = 26: initFromGridMap with the two distances the author commented out restored
 * (`precision * 1.414f` for edge neighbours :380, `precision * 1.732f` for corner neighbours :383) and
 * the evaporation / reset loops (:270, :312, hard-wired to 6) covering every adjacency entry. */
wo_acs *wo_acs_create_nb(int32_t nx, int32_t ny, int32_t nz, const float *cx, const float *cy,
                         const float *cz, const uint8_t *free_, float precision, float pheromone_0, int32_t nb)
{
    if (nb != 6 && nb != 26) return NULL;
    wo_acs *s = (wo_acs *)calloc(1, sizeof(wo_acs));
    s->nb = nb;
    s->dx = nb == 6 ? DX : DX26; s->dy = nb == 6 ? DY : DY26; s->dz = nb == 6 ? DZ : DZ26;
    for (int k = 0; k < nb; k++) {
        int type = (s->dx[k] != 0) + (s->dy[k] != 0) + (s->dz[k] != 0);
        s->dist[k] = type == 1 ? precision : type == 2 ? precision * 1.414f : precision * 1.732f;
    }
    s->nx = nx; s->ny = ny; s->nz = nz;
    s->n = (int64_t)nx * ny * nz;
    s->precision = precision;
    s->cx = (float *)malloc(sizeof(float) * nx); memcpy(s->cx, cx, sizeof(float) * nx);
    s->cy = (float *)malloc(sizeof(float) * ny); memcpy(s->cy, cy, sizeof(float) * ny);
    s->cz = (float *)malloc(sizeof(float) * nz); memcpy(s->cz, cz, sizeof(float) * nz);
    s->free_ = (uint8_t *)malloc((size_t)s->n); memcpy(s->free_, free_, (size_t)s->n);
    s->pher = (float *)malloc(sizeof(float) * (size_t)nb * (size_t)s->n);
    s->visit = (uint32_t *)calloc((size_t)s->n, sizeof(uint32_t));
    s->bestmark = (uint32_t *)calloc((size_t)s->n, sizeof(uint32_t));
    s->best.L = INFINITY;
    /* initFromGridMap :343-408: in-bounds edges pheromone_0, out-of-bounds edges 0 */
    for (int32_t z = 0; z < nz; z++)
        for (int32_t y = 0; y < ny; y++)
            for (int32_t x = 0; x < nx; x++) {
                float *p = s->pher + (size_t)nb * (((size_t)z * ny + y) * nx + x);
                for (int k = 0; k < nb; k++) {
                    int32_t X = x + s->dx[k], Y = y + s->dy[k], Z = z + s->dz[k];
                    p[k] = (X < 0 || X >= nx || Y < 0 || Y >= ny || Z < 0 || Z >= nz) ? 0.f : pheromone_0;
                }
            }
    return s;
}

void wo_acs_destroy(wo_acs *s)
{
    if (!s) return;
    free(s->cx); free(s->cy); free(s->cz); free(s->free_); free(s->pher); free(s->visit);
    free(s->bestmark); free(s->best.ids); free(s->best.choice); free(s->last_len); free(s->last_L); free(s->last_ids); free(s);
}

void wo_acs_reset(wo_acs *s, float pheromone_0)
{ /* :307-315 -- every edge, the out-of-bounds ones included */
    for (int64_t i = 0; i < s->nb * s->n; i++) s->pher[i] = pheromone_0;
}

/* power() :48-60, T = float */
static float powi_f(float x, int y)
{
    float ans = 1;
    while (y) {
        if (y & 1) ans *= x;
        x *= x;
        y >>= 1;
    }
    return ans;
}

/* (1 + beta*cos) for neighbour k of voxel (x,y,z) towards end voxel -- :137,:151-154 */
static float heuristic_term(const wo_acs *s, int32_t x, int32_t y, int32_t z, int32_t nbx,
                            int32_t nby, int32_t nbz, int32_t ex, int32_t ey, int32_t ez, float beta)
{
    float ax = s->cx[ex] - s->cx[x], ay = s->cy[ey] - s->cy[y], az = s->cz[ez] - s->cz[z];
    float bx = s->cx[nbx] - s->cx[x], by = s->cy[nby] - s->cy[y], bz = s->cz[nbz] - s->cz[z];
    float dot = ax * bx + ay * by + az * bz;
    float na = sqrtf(ax * ax + ay * ay + az * az);
    float nb = sqrtf(bx * bx + by * by + bz * bz);
    float c = dot / (na * nb);
    return 1 + beta * c;
}

void wo_acs_heuristic(const wo_acs *s, int64_t end_id, float beta, float *out)
{
    int32_t ex = (int32_t)(end_id % s->nx), ey = (int32_t)((end_id / s->nx) % s->ny),
            ez = (int32_t)(end_id / ((int64_t)s->nx * s->ny));
    for (int32_t z = 0; z < s->nz; z++)
        for (int32_t y = 0; y < s->ny; y++)
            for (int32_t x = 0; x < s->nx; x++)
                for (int k = 0; k < s->nb; k++) {
                    int32_t X = x + s->dx[k], Y = y + s->dy[k], Z = z + s->dz[k];
                    size_t o = (size_t)s->nb * (((size_t)z * s->ny + y) * s->nx + x) + k;
                    if (X < 0 || X >= s->nx || Y < 0 || Y >= s->ny || Z < 0 || Z >= s->nz) out[o] = 0.f;
                    else out[o] = heuristic_term(s, x, y, z, X, Y, Z, ex, ey, ez, beta);
                }
}

/* One ant: addStartNode :81-86 then while(selectNext) :134-193 */
static void walk_ant(wo_acs *s, const wo_acs_params *p, wo_ant *a, int64_t start, int64_t end,
                     wo_glibc_rand *rng, uint32_t gen, uint32_t ant_idx)
{
    const int32_t nx = s->nx, ny = s->ny, nz = s->nz;
    const int32_t ex = (int32_t)(end % nx), ey = (int32_t)((end / nx) % ny), ez = (int32_t)(end / ((int64_t)nx * ny));
    if (++s->visit_stamp == 0) { memset(s->visit, 0, sizeof(uint32_t) * (size_t)s->n); s->visit_stamp = 1; }
    const uint32_t stamp = s->visit_stamp;
    a->len = 0;
    a->L = 0;
    ant_push(a, (int32_t)start, -1);
    s->visit[start] = stamp;
    int64_t cur = start;
    uint32_t step = 0;
    for (;;) {
        int32_t x = (int32_t)(cur % nx), y = (int32_t)((cur / nx) % ny), z = (int32_t)(cur / ((int64_t)nx * ny));
        const int NB = s->nb;
        const int *DX = s->dx, *DY = s->dy, *DZ = s->dz;
        float info[26];
        int adm[26], nadm = 0;
        float total = 0;
        for (int k = 0; k < NB; k++) {
            adm[k] = 0;
            int32_t X = x + DX[k], Y = y + DY[k], Z = z + DZ[k];
            if (X < 0 || X >= nx || Y < 0 || Y >= ny || Z < 0 || Z >= nz) continue; /* self-pointer :148,:395 */
            int64_t nb = ((int64_t)Z * ny + Y) * nx + X;
            if (s->visit[nb] == stamp) continue; /* tabu :145-146 */
            if (!s->free_[nb]) continue;        /* :148 */
            info[k] = powi_f(s->pher[cur * NB + k], p->alpha) *
                      heuristic_term(s, x, y, z, X, Y, Z, ex, ey, ez, p->beta); /* :154 */
            total += info[k];                                                   /* :155 */
            adm[k] = 1;
            nadm++;
        }
        if (nadm == 0) { a->L = INFINITY; return; } /* :162-166 */
        int32_t r = p->rng_mode == WO_RNG_REF ? wo_rand(rng)
                                              : (int32_t)wo_ctr_rand31(p->seed, p->stream, gen, ant_idx, step);
        float rnd = (float)r / (float)2147483647; /* (float)RAND_MAX == 2^31 :169 */
        rnd *= total;
        float prob = 0;
        int pick = -1;
        for (int k = NB - 1; k >= 0; k--) { /* reverse cumulative order :172-189 */
            if (!adm[k]) continue;
            prob += info[k];
            if (prob >= rnd) { pick = k; break; }
        }
        if (pick < 0) { a->L = INFINITY; return; } /* :191-192 (see header note on the UB case) */
        int64_t nb = ((int64_t)(z + DZ[pick]) * ny + (y + DY[pick])) * nx + (x + DX[pick]);
        s->visit[nb] = stamp;
        ant_push(a, (int32_t)nb, (int8_t)pick);
        a->L += s->dist[pick]; /* :78; distance == precision for the six face neighbours :378 */
        step++;
        if (nb == end) return; /* :182-186 */
        cur = nb;
    }
}

static void ant_copy(wo_ant *dst, const wo_ant *src)
{
    if (dst->cap < src->len) {
        dst->cap = src->len;
        dst->ids = (int32_t *)realloc(dst->ids, (size_t)dst->cap * sizeof(int32_t));
        dst->choice = (int8_t *)realloc(dst->choice, (size_t)dst->cap);
    }
    memcpy(dst->ids, src->ids, (size_t)src->len * sizeof(int32_t));
    memcpy(dst->choice, src->choice, (size_t)src->len);
    dst->len = src->len;
    dst->L = src->L;
}

int32_t wo_acs_solve(wo_acs *s, const wo_acs_params *p, int64_t start_id, int64_t end_id,
                     wo_glibc_rand *rng, float *trace_bestL, float *trace_iterbestL,
                     int32_t *trace_colony, int32_t *trace_finite, int64_t *trace_steps, double *timing)
{
    if (p->rng_mode == WO_RNG_REF && !rng) return -1;
    wo_ant *ants = NULL;
    int32_t ants_cap = 0;
    float *keys = NULL;
    int32_t *perm = NULL;
    double tw = 0, te = 0, ts = 0, td = 0;
    s->best.L = INFINITY; /* :232 -- best.path is NOT cleared (Q9) */
    for (int32_t g = 0; g < p->max_iteration; g++) {
        /* :247-249 */
        int32_t colony;
        if (p->fixed_colony > 0) colony = p->fixed_colony;
        else colony = (int32_t)(0.35 * (s->best.L < p->predict ? s->best.L : p->predict) / s->precision);
        float lambda = 0.2 * colony;
        float Q = p->pheromone_0 / lambda * (s->best.L == INFINITY ? p->predict : s->best.L);
        if (colony > ants_cap) {
            ants = (wo_ant *)realloc(ants, sizeof(wo_ant) * (size_t)colony);
            memset(ants + ants_cap, 0, sizeof(wo_ant) * (size_t)(colony - ants_cap));
            keys = (float *)realloc(keys, sizeof(float) * (size_t)colony);
            perm = (int32_t *)realloc(perm, sizeof(int32_t) * (size_t)colony);
            ants_cap = colony;
        }
        double t0 = now_s();
        float iterbest = INFINITY;
        int32_t finite = 0;
        int64_t steps = 0;
        for (int32_t a = 0; a < colony; a++) { /* :252-265 */
            walk_ant(s, p, &ants[a], start_id, end_id, rng, (uint32_t)g, (uint32_t)a);
            if (ants[a].L < s->best.L) {
                ant_copy(&s->best, &ants[a]);
                s->best_ver++;
                for (int64_t i = 0; i < s->best.len; i++) s->bestmark[s->best.ids[i]] = s->best_ver;
            }
            if (ants[a].L < iterbest) iterbest = ants[a].L;
            if (ants[a].L != INFINITY) finite++;
            steps += ants[a].len - 1;
        }
        double t1 = now_s();
        for (int64_t i = 0; i < s->nb * s->n; i++) s->pher[i] *= p->rho; /* :268-272 */
        double t2 = now_s();
        for (int32_t a = 0; a < colony; a++) keys[a] = ants[a].L;
        if (p->rng_mode == WO_RNG_REF) wo_std_sort_perm(keys, colony, perm); /* :273-274 */
        else wo_stable_rank_perm(keys, colony, perm);
        double t3 = now_s();
        for (int32_t o = 1; o <= colony; o++) { /* :275-280 -> update_pheromone :198-215 */
            const wo_ant *a = &ants[perm[o - 1]];
            if (a->L == INFINITY || (float)o > lambda - 1) continue;
            for (int64_t i = 0; i + 1 < a->len; i++) {
                int32_t v = a->ids[i], w = a->ids[i + 1];
                int k = a->choice[i + 1];
                int onbest = s->bestmark[v] == s->best_ver && s->bestmark[w] == s->best_ver; /* :209 */
                s->pher[(size_t)v * s->nb + k] += (lambda - (float)o) * Q / a->L + (float)onbest * lambda * Q / s->best.L;
            }
        }
        double t4 = now_s();
        tw += t1 - t0; te += t2 - t1; ts += t3 - t2; td += t4 - t3;
        if (trace_bestL) trace_bestL[g] = s->best.L;
        if (trace_iterbestL) trace_iterbestL[g] = iterbest;
        if (trace_colony) trace_colony[g] = colony;
        if (trace_finite) trace_finite[g] = finite;
        if (trace_steps) trace_steps[g] = steps;
        s->last_colony = colony; s->last_lambda = lambda; s->last_Q = Q;
        if (colony > s->last_cap) {
            s->last_len = (int32_t *)realloc(s->last_len, sizeof(int32_t) * (size_t)colony);
            s->last_L = (float *)realloc(s->last_L, sizeof(float) * (size_t)colony);
            s->last_cap = colony;
        }
        int64_t tot = 0;
        for (int32_t a = 0; a < colony; a++) { s->last_len[a] = (int32_t)ants[a].len; s->last_L[a] = ants[a].L; tot += ants[a].len; }
        if (g == p->max_iteration - 1) {
            if (tot > s->last_ids_cap) { s->last_ids = (int32_t *)realloc(s->last_ids, sizeof(int32_t) * (size_t)tot); s->last_ids_cap = tot; }
            int64_t o = 0;
            for (int32_t a = 0; a < colony; a++) { memcpy(s->last_ids + o, ants[a].ids, sizeof(int32_t) * (size_t)ants[a].len); o += ants[a].len; }
        }
    }
    for (int32_t a = 0; a < ants_cap; a++) { free(ants[a].ids); free(ants[a].choice); }
    free(ants); free(keys); free(perm);
    if (timing) { timing[0] = tw; timing[1] = te; timing[2] = ts; timing[3] = td; }
    return 0;
}

float wo_acs_best_L(const wo_acs *s) { return s->best.L; }
int64_t wo_acs_best_len(const wo_acs *s) { return s->best.len; }
void wo_acs_best_path(const wo_acs *s, int32_t *ids, int32_t *choice)
{
    for (int64_t i = 0; i < s->best.len; i++) ids[i] = s->best.ids[i];
    if (choice) for (int64_t i = 1; i < s->best.len; i++) choice[i - 1] = s->best.choice[i];
}
const float *wo_acs_pheromone(const wo_acs *s) { return s->pher; }
void wo_acs_last_params(const wo_acs *s, int32_t *colony, float *lambda, float *Q)
{
    *colony = s->last_colony; *lambda = s->last_lambda; *Q = s->last_Q;
}
void wo_acs_last_paths(const wo_acs *s, int32_t *ids)
{
    int64_t tot = 0;
    for (int32_t a = 0; a < s->last_colony; a++) tot += s->last_len[a];
    memcpy(ids, s->last_ids, sizeof(int32_t) * (size_t)tot);
}
int32_t wo_acs_last_ants(const wo_acs *s, int32_t *lens, float *L)
{
    for (int32_t a = 0; a < s->last_colony; a++) { if (lens) lens[a] = s->last_len[a]; if (L) L[a] = s->last_L[a]; }
    return s->last_colony;
}

/* ================================================================== ACS_GTSP ======= */
static double powi_d(double x, int y)
{ /* power() ACSRank_3D.hpp:48-60, T = double */
    double ans = 1;
    while (y) {
        if (y & 1) ans *= x;
        x *= x;
        y >>= 1;
    }
    return ans;
}

int32_t wo_gtsp_solve(const double *dist, int32_t n, int32_t cnt, const wo_gtsp_params *p,
                      wo_glibc_rand *rng, int32_t *tour_edges, double *tour_L, double *pher_out)
{
    const double INF = (double)0x3f3f3f3f; /* ACS_GTSP.hpp:19 */
    const double alpha = 0.1;              /* :189 */
    const int delta = 1, beta = 6;         /* :190-191 */
    const int32_t colony = n;              /* :192 */
    size_t nn = (size_t)n * n;
    double *pher = (double *)malloc(sizeof(double) * nn), *heur = (double *)malloc(sizeof(double) * nn),
           *info = (double *)malloc(sizeof(double) * nn);
    /* readFromGraphFile :239-249: tmp accumulates in upper-triangle raster order */
    double tmp = 0;
    for (int32_t i = 0; i < n; i++)
        for (int32_t j = i + 1; j < n; j++) tmp += dist[(size_t)i * n + j];
    double pher0 = (double)cnt / (tmp * n);
    for (int32_t i = 0; i < n; i++)
        for (int32_t j = 0; j < n; j++) { /* :207-212 */
            pher[(size_t)i * n + j] = pher0;
            heur[(size_t)i * n + j] = 1 / ((i == j ? 0.0 : dist[(size_t)i * n + j]) + 1e-8);
        }
    uint8_t *inJ = (uint8_t *)malloc(nn); /* J[k] as membership flags, iterated ascending like std::set */
    int32_t *r = (int32_t *)malloc(sizeof(int32_t) * n), *cntJ = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *tours = (int32_t *)malloc(sizeof(int32_t) * 2 * nn); /* [ant][step][2] */
    int32_t *best = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    double bestL = INF; /* best.clean() :214 */
    int32_t best_sz = 0;
    int32_t max_it = p->max_iterations > 0 ? p->max_iterations : n * n; /* :216 */
    double last = INF;
    int32_t bad = 0, it = 0;
    for (; it < max_it; it++) { /* :261-276 */
        if (bad > n) break;
        /* reset :103-120 */
        for (int32_t k = 0; k < n; k++) {
            memset(inJ + (size_t)k * n, 1, (size_t)n);
            inJ[(size_t)k * n + k] = 0;
            cntJ[k] = n - 1;
            r[k] = k;
        }
        for (size_t e = 0; e < nn; e++) info[e] = powi_d(pher[e], delta) * powi_d(heur[e], beta);
        /* construct_solution :146-159 */
        for (int32_t step = 0; step < n; step++)
            for (int32_t k = 0; k < colony; k++) {
                int32_t next = k; /* r1[k] */
                uint8_t *J = inJ + (size_t)k * n;
                if (cntJ[k] > 0) { /* select_next :122-144 */
                    int32_t rv = p->rng_mode == WO_RNG_REF
                                     ? wo_rand(rng)
                                     : (int32_t)wo_ctr_rand31(p->seed, p->stream, (uint32_t)it, (uint32_t)k, (uint32_t)step);
                    double rnd = (double)rv / (double)2147483647;
                    const double *row = info + (size_t)r[k] * n;
                    double sum = 0, sp = 0;
                    for (int32_t c = 0; c < n; c++) if (J[c]) sum += row[c];
                    rnd *= sum;
                    for (int32_t c = 0; c < n; c++)
                        if (J[c]) {
                            sp += row[c];
                            if (sp >= rnd) { next = c; break; }
                        }
                }
                if (J[next]) { J[next] = 0; cntJ[k]--; }
                tours[((size_t)k * n + step) * 2] = r[k];
                tours[((size_t)k * n + step) * 2 + 1] = next;
                r[k] = next;
            }
        /* update_pheromone :161-185 */
        double nowL = INF;
        int32_t nowk = -1;
        for (int32_t k = 0; k < colony; k++) {
            double L = 0; /* calc :36-44: closing edge excluded */
            for (int32_t e = 0; e < n - 1; e++) {
                int32_t a = tours[((size_t)k * n + e) * 2], b = tours[((size_t)k * n + e) * 2 + 1];
                L += a == b ? 0.0 : dist[(size_t)a * n + b];
            }
            if (L < nowL) { nowL = L; nowk = k; }
        }
        if (nowk >= 0 && nowL < bestL) {
            bestL = nowL;
            best_sz = n;
            memcpy(best, tours + (size_t)nowk * n * 2, sizeof(int32_t) * 2 * (size_t)n);
        }
        for (size_t e = 0; e < nn; e++) pher[e] *= (1 - alpha);
        if (nowk >= 0)
            for (int32_t e = 0; e < n; e++) {
                int32_t a = tours[((size_t)nowk * n + e) * 2], b = tours[((size_t)nowk * n + e) * 2 + 1];
                pher[(size_t)a * n + b] += 1. / (double)nowL;
                pher[(size_t)b * n + a] = pher[(size_t)a * n + b];
            }
        if (last > bestL) { last = bestL; bad = 0; }
        else bad++;
    }
    for (int32_t e = 0; e < 2 * best_sz; e++) tour_edges[e] = best[e];
    *tour_L = bestL;
    if (pher_out) memcpy(pher_out, pher, sizeof(double) * nn);
    free(pher); free(heur); free(info); free(inJ); free(r); free(cntJ); free(tours); free(best);
    return it;
}

/* =====================================================================================
 * BS_Basic<float, DIM, DEGREE, CI, CF>  (core/BSplineBasic.h) -- trajectory smoothing that
 * main.cpp:287-352 applies to the stitched path.  Template arguments are run-time here.
 * ===================================================================================== */
#define BS_W (WO_BS_MAX_DEGREE + 1)

int wo_bspline_init(wo_bspline *b, int32_t dim, int32_t degree, int32_t ci, int32_t cf,
                    int64_t n_middle, float uninit)
{
    memset(b, 0, sizeof *b);
    /* ci/cf > DEGREE would index _ndu[DEGREE-k+1] with a negative row (:279-281): not restated */
    if (dim < 1 || dim > WO_BS_MAX_DIM || degree < 0 || degree > WO_BS_MAX_DEGREE || ci < 0 ||
        cf < 0 || ci > degree || cf > degree || n_middle < 0)
        return -1;
    b->dim = dim; b->degree = degree; b->ci = ci; b->cf = cf; b->n_middle = n_middle;
    b->n_knots = degree + n_middle + 2 + ci + cf + 1;                         /* :38-39 */
    b->n_cps = n_middle + 2 + ci + cf;                                        /* :40    */
    if (b->n_knots < 2 * (degree + 1)) return -1;                             /* :53-55 "Invalid setup" */
    b->knots = (float *)calloc((size_t)b->n_knots, sizeof(float));            /* :44 zeroed */
    b->cps = (float *)calloc((size_t)(b->n_cps * dim), sizeof(float));        /* :47-51 zeroed */
    b->uninit = uninit;
    return (b->knots && b->cps) ? 0 : -1;
}

void wo_bspline_free(wo_bspline *b)
{
    free(b->knots); free(b->cps);
    b->knots = NULL; b->cps = NULL;
}

/* _findSpan :358-385 */
static int bs_find_span(const wo_bspline *b, float u, int64_t *ret)
{
    const float *K = b->knots;
    int64_t nk = b->n_knots;
    if (u < K[0] || K[nk - 1] < u) return 0;
    float d = u - K[nk - 1];
    if ((double)(d * d) < 1.e-10) {            /* SP_IS_EQUAL :8: fp32 product against a double literal */
        for (int64_t i = nk - 2; i > -1; --i)
            if (K[i] < u && u <= K[i + 1]) { *ret = i; return 1; }
        return 0;
    }
    int64_t low = 0, high = nk - 1, mid = (low + high) >> 1;
    int guard = 0;
    while (u < K[mid] || u >= K[mid + 1]) {
        if (u < K[mid]) high = mid; else low = mid;
        mid = (low + high) >> 1;
        if (++guard > 200) return 0;           /* the reference would spin forever (non-monotone knots) */
    }
    *ret = mid;
    return 1;
}

/* _BasisFuns :330-353, with _Left/_Right :354-356 */
static void bs_basis_funs(const wo_bspline *b, float *N, int64_t span, float u)
{
    const float *K = b->knots;
    float left = 0.0f, right = 0.0f, saved = 0.0f, temp = 0.0f;
    N[0] = 1.0f;
    for (int j = 1; j <= b->degree; ++j) {
        saved = 0.0f;
        for (int r = 0; r < j; ++r) {
            left = u - K[span + 1 - (j - r)];
            right = K[span + (r + 1)] - u;
            if ((right + left) != 0) temp = N[r] / (right + left);   /* temp survives a zero sum */
            N[r] = saved + right * temp;
            saved = left * temp;
        }
        N[j] = saved;
    }
}

/* _BasisFunsDers(ders, span, u, n) :237-323 */
static void bs_basis_ders(const wo_bspline *b, float ders[][BS_W + 2], int64_t span, float u, int n)
{
    const float *K = b->knots;
    const int D = b->degree;
    float ndu[BS_W][BS_W], a[2][BS_W];
    float saved = 0.0f, left = 0.0f, right = 0.0f, temp = 0.0f, d = 0.0f;
    for (int i = 0; i < BS_W; i++)
        for (int j = 0; j < BS_W; j++) ndu[i][j] = b->uninit;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < BS_W; j++) a[i][j] = b->uninit;
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

/* _BasisFunsDers(ders, u, n) :227-235: nothing is written when the span search fails */
static void bs_basis_ders_at(const wo_bspline *b, float ders[][BS_W + 2], float u, int n)
{
    int64_t span;
    if (!bs_find_span(b, u, &span)) return;
    bs_basis_ders(b, ders, span, u, n);
}

void wo_bspline_set_param(wo_bspline *b, const float *init, const float *fin, const float *middle,
                          int64_t stride, float fin_time)
{
    const int D = b->degree, dim = b->dim;
    float *K = b->knots, *C = b->cps;
    /* _CalcKnot :150-171 */
    {
        int64_t i = 0;
        int64_t nmid = b->n_knots - 2 * D - 2;
        float step = fin_time / (float)(nmid + 1);
        for (int j = 0; j < D + 1; ++j) K[i++] = 0.0f;
        for (int64_t j = 0; j < nmid; ++j) { K[i] = K[i - 1] + step; ++i; }
        for (int j = 0; j < D + 1; ++j) K[i++] = fin_time;
    }
    /* _CalcConstrainedCPoints :387-447 */
    for (int m = 0; m < dim; ++m) {
        C[m] = init[m];
        C[(b->n_cps - 1) * dim + m] = fin[m];
    }
    float mat[BS_W][BS_W + 2];
    for (int i = 0; i < BS_W; i++)
        for (int j = 0; j < BS_W + 2; j++) mat[i][j] = b->uninit;
    bs_basis_ders_at(b, mat, 0.0f, b->ci);
    for (int j = 1; j < b->ci + 1; ++j)
        for (int k = 0; k < dim; ++k) {
            float v = init[j * dim + k];
            for (int h = j; h > 0; --h) v -= mat[j][h - 1] * C[(h - 1) * dim + k];
            C[j * dim + k] = v / mat[j][j];
        }
    for (int i = 0; i < BS_W; i++)
        for (int j = 0; j < BS_W + 2; j++) mat[i][j] = b->uninit;
    bs_basis_ders_at(b, mat, fin_time, b->cf);
    {
        int idx = 1;
        for (int64_t j = b->n_cps - 2; j > b->n_cps - 2 - b->cf; --j) {
            for (int k = 0; k < dim; ++k) {
                float v = fin[idx * dim + k];
                for (int h = idx; h > 0; --h)
                    v -= mat[idx][b->cf + 2 - h] * C[(b->n_cps - h) * dim + k];   /* column cf+1 is never written when cf+1 > DEGREE */
                C[j * dim + k] = v / mat[idx][b->cf + 1 - idx];
            }
            ++idx;
        }
    }
    /* _CalcCPoints :458-464 */
    for (int64_t i = 0; i < b->n_middle; ++i)
        for (int m = 0; m < dim; ++m) C[(b->ci + 1 + i) * dim + m] = middle[i * stride + m];
}

static int bs_span_in_range(const wo_bspline *b, int64_t span)
{
    return span - b->degree >= 0 && span < b->n_cps && span + b->degree < b->n_knots;
}

int wo_bspline_point(const wo_bspline *b, float u, float *ret)
{
    const float *K = b->knots;
    int64_t span;
    if (u < K[0]) u = K[0];
    else if (u > K[b->n_knots - 1]) u = K[b->n_knots - 1];
    if (!bs_find_span(b, u, &span)) return 0;
    if (!bs_span_in_range(b, span)) return 0;      /* reference: out-of-bounds read */
    float N[BS_W];
    bs_basis_funs(b, N, span, u);
    for (int j = 0; j < b->dim; ++j) {
        float c = 0.0f;
        for (int i = 0; i <= b->degree; ++i) c += N[i] * b->cps[(span - b->degree + i) * b->dim + j];
        ret[j] = c;
    }
    return 1;
}

int wo_bspline_der(const wo_bspline *b, float u, int32_t d, float *ret)
{
    const float *K = b->knots;
    int64_t span;
    if (d > b->degree || d < 0) return 0;          /* :123 `return 0.0` == false */
    if (u < K[0]) u = K[0];
    else if (u > K[b->n_knots - 1]) u = K[b->n_knots - 1];
    if (!bs_find_span(b, u, &span)) return 0;      /* _CurveDerivsAlg1V :173-203 */
    if (!bs_span_in_range(b, span)) return 0;
    float nders[BS_W][BS_W + 2];
    bs_basis_ders(b, nders, span, u, d);
    for (int m = 0; m < b->dim; ++m) {
        float c = 0.0f;
        for (int j = 0; j <= b->degree; ++j) c += nders[d][j] * b->cps[(span - b->degree + j) * b->dim + m];
        ret[m] = c;
    }
    return 1;
}

void wo_bspline_sample(const wo_bspline *b, float t0, float dt, int64_t count, int32_t der, float *out,
                       uint8_t *ok)
{
    for (int64_t i = 0; i < count; ++i) {
        float u = t0 + (float)i * dt;
        int r = der == 0 ? wo_bspline_point(b, u, out + i * b->dim) : wo_bspline_der(b, u, der, out + i * b->dim);
        if (ok) ok[i] = (uint8_t)r;
    }
}

/* ACS_GTSP::read_all_segments (ACS_GTSP.hpp:286-298): segment after segment, node after node;
 * node id -> (x, y, z) index -> axis-table coordinates (model_grid_map.hpp:204-215) */
void wo_stitch_segments(const int64_t *seg_ids, const int64_t *seg_off, int32_t n_seg, const uint8_t *reverse,
                        int32_t nx, int32_t ny, const float *cx, const float *cy, const float *cz,
                        float *out_xyz)
{
    int64_t o = 0;
    for (int32_t s = 0; s < n_seg; ++s) {
        int64_t a = seg_off[s], e = seg_off[s + 1];
        for (int64_t j = 0; j < e - a; ++j) {
            int64_t id = (reverse && reverse[s]) ? seg_ids[e - 1 - j] : seg_ids[a + j];
            int64_t x = id % nx, y = (id / nx) % ny, z = id / ((int64_t)nx * ny);
            out_xyz[o * 3 + 0] = cx[x];
            out_xyz[o * 3 + 1] = cy[y];
            out_xyz[o * 3 + 2] = cz[z];
            ++o;
        }
    }
}
