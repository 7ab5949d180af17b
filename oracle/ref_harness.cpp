// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Drives the *unmodified* reference headers where they lie under /root/reference
// (core/ACSRank_3D.hpp, core/read_STL.hpp, core/ACS_GTSP.hpp, core/model_grid_map.hpp)
// and dumps what they compute into a small tagged binary file ("WAF", see
// tests/waf.py) so that oracle/weld_oracle.c -- the plain-C restatement -- can be
// pinned bit-for-bit against the real thing, and so that the reference's own CPU
// loop can be timed on the GPU box's host cores (bench.py cpu_baseline.kind =
// "reference").
//
// Built by oracle/Makefile into oracle/_ref/ref_harness (git-ignored; travels to the
// GPU box as a binary).  No reference source text is copied here: everything the
// reference computes is reached through #include + -fno-access-control, i.e. this
// file only *calls* private members (initFromGridMap, setPoints, computeSolution,
// selectNext, update_pheromone, ...).  The process-global `time()` is interposed so
// that the `srand(time(0))` inside ACS_Rank::initFromGridMap (ACSRank_3D.hpp:327)
// becomes a seedable call, and `rand()` is wrapped (dlsym RTLD_NEXT) only to COUNT
// calls -- the numbers still come from glibc.
//
// Sub-commands (key=value arguments):
//   rand      seed= n=                         glibc rand() known answers
//   sort      n= seed= levels=                 std::sort permutation on tied float keys
//   voxelize  stl= p= wall= [gridout=]         STLReader + GridMap::creatGridMap
//   acs       (stl= p= wall= | gridin=) (snode=z,y,x enode=z,y,x | spt=x,y,z ept=x,y,z)
//             seed= iters= predict= [driven=0|1] [fixed=N] [dumppher=1]
//   pairs     (stl= p= wall= | gridin=) pts=FILE predict= seed= graph=FILE [gtsp=1]
//   gtsp      graph=FILE seed=
// every command takes out=FILE (WAF) and prints one JSON line with timings on stderr.
#include <dlfcn.h>
#include <sys/time.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

// ---- interposers (must precede the reference headers only in link order, not text)
static long g_fake_time = 0;
static unsigned long long g_rand_calls = 0;
extern "C" time_t time(time_t *t)
{
    if (t) *t = (time_t)g_fake_time;
    return (time_t)g_fake_time;
}
extern "C" int rand(void)
{
    static int (*real)(void) = nullptr;
    if (!real) real = (int (*)(void))dlsym(RTLD_NEXT, "rand");
    ++g_rand_calls;
    return real();
}

#include "core/ACSRank_3D.hpp"
#include "core/read_STL.hpp"
#include "core/ACS_GTSP.hpp"

static double now_s()
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

// ---------------------------------------------------------------- WAF writer
struct Waf {
    FILE *fp;
    explicit Waf(const std::string &path) : fp(fopen(path.c_str(), "wb"))
    {
        if (!fp) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(4); }
        fwrite("WAF1", 1, 4, fp);
    }
    ~Waf() { if (fp) fclose(fp); }
    void put(const char *name, char dtype, const void *data, uint64_t count, size_t esz)
    {
        uint32_t nl = (uint32_t)strlen(name);
        fwrite(&nl, 4, 1, fp);
        fwrite(name, 1, nl, fp);
        fwrite(&dtype, 1, 1, fp);
        fwrite(&count, 8, 1, fp);
        if (count) fwrite(data, esz, count, fp);
    }
    void i32(const char *n, const std::vector<int32_t> &v) { put(n, 'i', v.data(), v.size(), 4); }
    void i64(const char *n, const std::vector<int64_t> &v) { put(n, 'q', v.data(), v.size(), 8); }
    void f32(const char *n, const std::vector<float> &v) { put(n, 'f', v.data(), v.size(), 4); }
    void f64(const char *n, const std::vector<double> &v) { put(n, 'd', v.data(), v.size(), 8); }
    void u8(const char *n, const std::vector<uint8_t> &v) { put(n, 'B', v.data(), v.size(), 1); }
    void one_i64(const char *n, int64_t v) { put(n, 'q', &v, 1, 8); }
    void one_f32(const char *n, float v) { put(n, 'f', &v, 1, 4); }
    void one_f64(const char *n, double v) { put(n, 'd', &v, 1, 8); }
    void str(const char *n, const std::string &s) { put(n, 'B', s.data(), s.size(), 1); }
};

typedef std::map<std::string, std::string> Args;
static Args parse(int argc, char **argv)
{
    Args a;
    for (int i = 2; i < argc; i++) {
        std::string s(argv[i]);
        size_t eq = s.find('=');
        if (eq == std::string::npos) a[s] = "1";
        else a[s.substr(0, eq)] = s.substr(eq + 1);
    }
    return a;
}
static bool has(const Args &a, const char *k) { return a.find(k) != a.end(); }
static std::string gets(const Args &a, const char *k, const char *def = "")
{
    auto it = a.find(k);
    return it == a.end() ? std::string(def) : it->second;
}
static long getl(const Args &a, const char *k, long def) { return has(a, k) ? atol(gets(a, k).c_str()) : def; }
static float getf(const Args &a, const char *k, float def) { return has(a, k) ? strtof(gets(a, k).c_str(), nullptr) : def; }
static std::vector<float> get3f(const Args &a, const char *k)
{
    std::vector<float> v;
    std::string s = gets(a, k);
    char *p = &s[0];
    while (*p) { v.push_back(strtof(p, &p)); if (*p == ',') ++p; }
    return v;
}

// ---------------------------------------------------------------- helpers on reference objects
static void dump_grid(Waf &w, ACS_Rank &s)
{
    Vertex3<float> ***g = s.ptr_grid_map();
    int nx = s.rangeX, ny = s.rangeY, nz = s.rangeZ;
    std::vector<int32_t> dims = {nx, ny, nz, s.wall};
    w.i32("dims", dims);
    w.one_f32("precision", s.precision);
    std::vector<float> cx(nx), cy(ny), cz(nz);
    for (int x = 0; x < nx; x++) cx[x] = g[0][0][x].pt.x;
    for (int y = 0; y < ny; y++) cy[y] = g[0][y][0].pt.y;
    for (int z = 0; z < nz; z++) cz[z] = g[z][0][0].pt.z;
    w.f32("cx", cx);
    w.f32("cy", cy);
    w.f32("cz", cz);
    std::vector<uint8_t> fr((size_t)nx * ny * nz);
    size_t n = 0;
    int64_t sep_bad = 0;  // sanity: coordinates separable per axis and ids raster z,y,x
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++) {
                const Vertex3<float> &v = g[z][y][x];
                fr[n] = v.isFree ? 1 : 0;
                if (v.pt.x != cx[x] || v.pt.y != cy[y] || v.pt.z != cz[z] || v.id != n) sep_bad++;
                n++;
            }
    w.u8("free", fr);
    w.one_i64("separable_violations", sep_bad);
}

static bool make_grid(const Args &a, STLReader &model, ACS_Rank &s, double *t_vox)
{
    double t0 = now_s();
    if (has(a, "gridin")) {
        s.readGridMap(gets(a, "gridin"));
    } else {
        model.readFile(gets(a, "stl"));
        const std::vector<Triangles<float>> meshes = model.TriangleList();
        s.creatGridMap(meshes, getf(a, "p", 0.005f), (int)getl(a, "wall", 10), gets(a, "gridout"));
    }
    if (t_vox) *t_vox = now_s() - t0;
    return s.ptr_grid_map() != NULL;
}

static std::vector<int32_t> path_ids(const Agent<float> &ag)
{
    std::vector<int32_t> ids;
    const std::vector<ACS_Node<float> *> *p = ag.getPath();
    for (auto n : *p) ids.push_back((int32_t)n->id);
    return ids;
}

static void dump_pheromone(Waf &w, ACS_Rank &s, bool full)
{
    int nx = s.rangeX, ny = s.rangeY, nz = s.rangeZ;
    std::vector<float> ph;
    if (full) ph.reserve((size_t)nx * ny * nz * 6);
    double sum = 0;
    uint64_t hx = 0;
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++)
                for (int k = 0; k < 6; k++) {
                    float v = s.nodes[z][y][x].adjacency_infos[k].pheromone;
                    uint32_t bits;
                    memcpy(&bits, &v, 4);
                    hx = (hx ^ bits) * 1099511628211ULL;
                    sum += v;
                    if (full) ph.push_back(v);
                }
    w.one_f64("pher_sum", sum);
    w.one_i64("pher_hash", (int64_t)hx);
    if (full) w.f32("pher", ph);
}

// Re-drive of the generation loop through the reference's own members, so that a per
// generation trace can be recorded and the ant count can be pinned (BASELINE's "256
// ants" -- the reference recomputes colony_num every generation, ACSRank_3D.hpp:247).
// With fixed == 0 this must reproduce ACS_Rank::computeSolution bit for bit; `acs
// driven=0` vs `driven=1` is the check of that.
struct Trace {
    std::vector<int32_t> colony, finite, steps;
    std::vector<float> bestL, iterbestL, lambda, Q;
    double t_walk = 0, t_evap = 0, t_sort = 0, t_dep = 0;
};
static void drive(ACS_Rank &s, float predict, int iters, int fixed, Trace &tr)
{
    s.path_x.clear(); s.path_y.clear(); s.path_z.clear();
    s.best.L = INF_FLOAT;
    for (int g = 0; g < iters; g++) {
        if (fixed > 0) s.colony_num = fixed;
        else s.colony_num = 0.35 * (s.best.L < predict ? s.best.L : predict) / s.precision;
        s.lambda = 0.2 * s.colony_num;
        s.Q = s.pheromone_0 / s.lambda * (s.best.L == INF_FLOAT ? predict : s.best.L);
        s.agents.resize(s.colony_num);
        double t0 = now_s();
        float ib = INF_FLOAT;
        int fin = 0, st = 0;
        for (auto &ant : s.agents) {
            ACS_Node<float> *cur = s.start_node, *next = nullptr;
            ant.addStartNode(cur);
            while (s.selectNext(ant, cur, next)) cur = next;
            if (ant.L < s.best.L) s.best = ant;
            if (ant.L < ib) ib = ant.L;
            if (ant.L != INF_FLOAT) fin++;
            st += (int)ant.getPath()->size() - 1;
        }
        double t1 = now_s();
        for_each_nodes(s.nodes, s.rangeX, s.rangeY, s.rangeZ, [&](int z, int y, int x) {
            for (int k = 0; k < 6; k++) s.nodes[z][y][x].adjacency_infos[k].pheromone *= s.rho;
        });
        double t2 = now_s();
        std::sort(s.agents.begin(), s.agents.end(),
                  [](Agent<float> &a, Agent<float> &b) -> bool { return a.L < b.L; });
        double t3 = now_s();
        int order = 1;
        for (auto &ant : s.agents) s.update_pheromone(ant, order++);
        double t4 = now_s();
        s.agents.clear();
        tr.t_walk += t1 - t0; tr.t_evap += t2 - t1; tr.t_sort += t3 - t2; tr.t_dep += t4 - t3;
        tr.colony.push_back(s.colony_num); tr.finite.push_back(fin); tr.steps.push_back(st);
        tr.bestL.push_back(s.best.L); tr.iterbestL.push_back(ib);
        tr.lambda.push_back(s.lambda); tr.Q.push_back(s.Q);
    }
}

// ---------------------------------------------------------------- commands
static int cmd_rand(const Args &a, Waf &w)
{
    srand((unsigned)getl(a, "seed", 1));
    int n = (int)getl(a, "n", 16);
    std::vector<int32_t> v(n);
    for (int i = 0; i < n; i++) v[i] = rand();
    w.i32("rand", v);
    w.one_i64("RAND_MAX", RAND_MAX);
    return 0;
}

struct Keyed { float L; int32_t tag; };
static int cmd_sort(const Args &a, Waf &w)
{
    // std::sort on (L, tag) records compared by L only -> which permutation libstdc++'s
    // introsort produces among ties.  Keys come from a tiny LCG so the
    // python side can regenerate them.
    int n = (int)getl(a, "n", 64), levels = (int)getl(a, "levels", 8);
    uint32_t st = (uint32_t)getl(a, "seed", 1);
    std::vector<Keyed> v(n);
    std::vector<float> keys(n);
    for (int i = 0; i < n; i++) {
        st = st * 1664525u + 1013904223u;
        v[i].L = keys[i] = (float)((st >> 8) % (uint32_t)levels);
        v[i].tag = i;
    }
    std::sort(v.begin(), v.end(), [](Keyed &x, Keyed &y) -> bool { return x.L < y.L; });
    std::vector<int32_t> perm(n);
    for (int i = 0; i < n; i++) perm[i] = v[i].tag;
    w.f32("keys", keys);
    w.i32("perm", perm);
    return 0;
}

static int cmd_voxelize(const Args &a, Waf &w)
{
    STLReader model;
    ACS_Rank s;
    double tv = 0;
    make_grid(a, model, s, &tv);
    dump_grid(w, s);
    w.one_i64("n_tris", (int64_t)model.TriangleList().size());
    std::vector<float> tris;
    for (auto &t : model.TriangleList()) {
        tris.push_back(t.nor_vec.x); tris.push_back(t.nor_vec.y); tris.push_back(t.nor_vec.z);
        for (int j = 0; j < 3; j++) { tris.push_back(t.vertex[j].x); tris.push_back(t.vertex[j].y); tris.push_back(t.vertex[j].z); }
    }
    w.f32("tris", tris);
    w.one_f64("t_voxelize", tv);
    fprintf(stderr, "{\"cmd\":\"voxelize\",\"t_voxelize\":%.6f}\n", tv);
    return 0;
}

static bool resolve_points(const Args &a, ACS_Rank &s, Point3<float> &sp, Point3<float> &ep)
{
    if (has(a, "snode")) {
        std::vector<float> u = get3f(a, "snode"), v = get3f(a, "enode");
        sp = s.nodes[(int)u[0]][(int)u[1]][(int)u[2]].pt;
        ep = s.nodes[(int)v[0]][(int)v[1]][(int)v[2]].pt;
    } else {
        std::vector<float> u = get3f(a, "spt"), v = get3f(a, "ept");
        sp = Point3<float>(u[0], u[1], u[2]);
        ep = Point3<float>(v[0], v[1], v[2]);
    }
    return s.setPoints(sp, ep);
}

static int cmd_acs(const Args &a, Waf &w)
{
    STLReader model;
    ACS_Rank s;
    double tv = 0;
    make_grid(a, model, s, &tv);
    long seed = getl(a, "seed", 12345);
    int iters = (int)getl(a, "iters", 150);
    float predict = getf(a, "predict", 10.f);
    int fixed = (int)getl(a, "fixed", 0);
    bool driven = getl(a, "driven", 0) != 0 || fixed > 0;
    double t0 = now_s();
    s.initFromGridMap();
    double t_init = now_s() - t0;
    s.max_iteration = iters;
    srand((unsigned)seed);
    Point3<float> sp, ep;
    bool ok = resolve_points(a, s, sp, ep);
    w.one_i64("points_ok", ok);
    if (!ok) return 0;
    w.one_i64("start_id", (int64_t)s.start_node->id);
    w.one_i64("end_id", (int64_t)s.end_node->id);
    g_rand_calls = 0;
    Trace tr;
    t0 = now_s();
    if (driven) drive(s, predict, iters, fixed, tr);
    else s.computeSolution(predict);
    double t_solve = now_s() - t0;
    std::vector<int32_t> dims = {s.rangeX, s.rangeY, s.rangeZ, s.wall};
    w.i32("dims", dims);
    w.one_f32("precision", s.precision);
    w.one_f32("best_L", s.best.L);
    w.i32("best_path", path_ids(s.best));
    w.i32("best_choice", std::vector<int32_t>(s.best.nodeIndex()->begin(), s.best.nodeIndex()->end()));
    w.one_i64("colony_last", s.colony_num);
    w.one_f32("lambda_last", s.lambda);
    w.one_f32("Q_last", s.Q);
    w.one_i64("rand_calls", (int64_t)g_rand_calls);
    w.one_i64("next_rand", rand());
    dump_pheromone(w, s, getl(a, "dumppher", 0) != 0);
    if (driven) {
        w.i32("tr_colony", tr.colony); w.i32("tr_finite", tr.finite); w.i32("tr_steps", tr.steps);
        w.f32("tr_bestL", tr.bestL); w.f32("tr_iterbestL", tr.iterbestL);
        w.f32("tr_lambda", tr.lambda); w.f32("tr_Q", tr.Q);
    }
    w.one_f64("t_init", t_init);
    w.one_f64("t_solve", t_solve);
    fprintf(stderr,
            "{\"cmd\":\"acs\",\"iters\":%d,\"t_solve\":%.6f,\"t_init\":%.6f,\"t_grid\":%.6f,\"t_walk\":%.6f,"
            "\"t_evap\":%.6f,\"t_sort\":%.6f,\"t_dep\":%.6f,\"rand_calls\":%llu,\"best_L\":%.9g}\n",
            iters, t_solve, t_init, tv, tr.t_walk, tr.t_evap, tr.t_sort, tr.t_dep, g_rand_calls, (double)s.best.L);
    return 0;
}

static std::string slurp(const std::string &p)
{
    std::string s;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return s;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

static void dump_gtsp(Waf &w, ACS_GTSP &g)
{
    std::vector<int32_t> tour;
    for (auto &e : g.best.path) { tour.push_back(e.first); tour.push_back(e.second); }
    w.i32("tour_edges", tour);
    w.one_f64("tour_L", g.best.L);
    w.one_i64("gtsp_iters", g.index_itera);
    w.one_f64("gtsp_pher0", g.pheromone_0);
}

static int cmd_pairs(const Args &a, Waf &w)
{
    // The unmodified main.cpp:279-283 sequence.  srand(time(0)) inside initFromGridMap
    // picks the seed up from the interposed time().
    STLReader model;
    ACS_Rank s;
    make_grid(a, model, s, nullptr);
    g_fake_time = getl(a, "seed", 12345);
    std::string graph = gets(a, "graph");
    g_rand_calls = 0;
    double t0 = now_s();
    s.searchBestPathOfPoints(getf(a, "predict", 0.5f), gets(a, "pts"), graph);
    double t_pairs = now_s() - t0;
    int P = (int)s.route_points.size();
    w.one_i64("n_points", P);
    w.one_i64("rand_calls_pairs", (int64_t)g_rand_calls);
    std::vector<float> costs;
    std::vector<int32_t> lens, allids;
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) {
            if (i == j) { costs.push_back(0); lens.push_back(0); continue; }
            std::vector<int32_t> ids = path_ids(s.best_matrix[i][j]);
            costs.push_back(s.best_matrix[i][j].L);
            lens.push_back((int32_t)ids.size());
            if (i < j) allids.insert(allids.end(), ids.begin(), ids.end());
        }
    w.f32("pair_cost", costs);
    w.i32("pair_len", lens);
    w.i32("pair_paths_upper", allids);
    w.str("graph_text", slurp(graph));
    dump_pheromone(w, s, false);
    double t_gtsp = 0;
    if (getl(a, "gtsp", 0)) {
        ACS_GTSP g;
        std::string gfile = gets(a, "graphfixed", graph.c_str());
        t0 = now_s();
        g.readFromGraphFile(gfile);
        g.computeSolution();
        t_gtsp = now_s() - t0;
        g.read_all_segments(s.best_matrix);
        dump_gtsp(w, g);
        w.f32("g_path_x", g.g_path_x);
        w.f32("g_path_y", g.g_path_y);
        w.f32("g_path_z", g.g_path_z);
        w.one_i64("segments", g.path_segment_nums());
    }
    w.one_i64("next_rand", rand());
    fprintf(stderr, "{\"cmd\":\"pairs\",\"t_pairs\":%.6f,\"t_gtsp\":%.6f}\n", t_pairs, t_gtsp);
    return 0;
}

static int cmd_gtsp(const Args &a, Waf &w)
{
    ACS_GTSP g;
    g.readFromGraphFile(gets(a, "graph"));
    srand((unsigned)getl(a, "seed", 1));
    g_rand_calls = 0;
    double t0 = now_s();
    g.computeSolution();
    double t = now_s() - t0;
    dump_gtsp(w, g);
    int N = g.city_num;
    std::vector<double> ph((size_t)N * N), dis((size_t)N * N);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            ph[(size_t)i * N + j] = g.pheromone[i][j];
            dis[(size_t)i * N + j] = i == j ? 0.0 : g.dis[i][j];
        }
    w.f64("gtsp_pher", ph);
    w.f64("gtsp_dis", dis);
    w.one_i64("rand_calls", (int64_t)g_rand_calls);
    w.one_i64("next_rand", rand());
    w.one_f64("t_gtsp", t);
    fprintf(stderr, "{\"cmd\":\"gtsp\",\"cities\":%d,\"iters\":%d,\"t_gtsp\":%.6f,\"best\":%.9g}\n", N, g.index_itera, t, g.best.L);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_harness <rand|sort|voxelize|acs|pairs|gtsp> key=value...\n"); return 2; }
    Args a = parse(argc, argv);
    if (!has(a, "out")) { fprintf(stderr, "out=FILE required\n"); return 2; }
    // the reference narrates every generation / distance on stdout: silence it
    if (!freopen("/dev/null", "w", stdout)) return 3;
    Waf w(gets(a, "out"));
    std::string c(argv[1]);
    if (c == "rand") return cmd_rand(a, w);
    if (c == "sort") return cmd_sort(a, w);
    if (c == "voxelize") return cmd_voxelize(a, w);
    if (c == "acs") return cmd_acs(a, w);
    if (c == "pairs") return cmd_pairs(a, w);
    if (c == "gtsp") return cmd_gtsp(a, w);
    fprintf(stderr, "unknown command %s\n", argv[1]);
    return 2;
}
