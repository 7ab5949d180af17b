// ACS_GTSP.hpp -- drop-in for the reference's core/ACS_GTSP.hpp on the C ABI (wa_gtsp_solve).
// Same class names and public members (ACS_Tour; ACS_GTSP::readFromGraphFile / computeSolution /
// read_all_segments / read_segment / path_segment_nums / plot_route_path / g_path_x,y,z).  Include it
// after ACSRank_3D.hpp, as main.cpp:9-11 does (the reference header silently relies on that order).
// Extensions: setDistanceMatrix (skip the graph.in round trip), setRngMode, setSeed, bestTour.
#ifndef _ACS_HPP
#define _ACS_HPP
#include "ACSRank_3D.hpp"

#define INF 0x3f3f3f3f

typedef std::pair<int, int> pair_int;

typedef struct ACS_Tour {
    std::vector<pair_int> path;
    double L;
    void clean() { L = INF; path.clear(); path.shrink_to_fit(); }
    void calc(double **_dis)
    {
        L = 0;
        int sz = path.size();
        for (int i = 0; i < sz - 1; i++) L += _dis[path[i].first][path[i].second];
    }
    void push_back(int x, int y) { path.push_back(std::make_pair(x, y)); }
    int size() { return (int)path.size(); }
    int r(int i) { return path[i].first; }
    int s(int i) { return path[i].second; }
    void print()
    {
        int sz = path.size();
        for (int i = 0; i < sz; i++) printf("%d->", path[i].first + 1);
        if (sz) printf("%d\n", path[sz - 1].second + 1);
    }
    bool operator<(const ACS_Tour &a) const { return L < a.L; }
} ACS_Tour;

class ACS_GTSP {
public:
    std::vector<float> g_path_x, g_path_y, g_path_z;

    ACS_GTSP() { best.clean(); }

    // reference ACS_GTSP.hpp:224-253
    bool readFromGraphFile(std::string filename)
    {
        FILE *fp = fopen(filename.c_str(), "r");
        if (!fp) return false;  // the reference dereferences NULL here
        int cnt = 0, n = 0;
        if (fscanf(fp, "%d %d", &n, &cnt) != 2 || n < 2) { fclose(fp); return false; }
        std::vector<double> d((size_t)n * n, 0.0);
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                double v = 0;
                if (fscanf(fp, "%lf", &v) != 1) v = 0;
                d[(size_t)i * n + j] = d[(size_t)j * n + i] = v;
                printf("distance: %lf \r\n", v);
            }
        fclose(fp);
        return setDistanceMatrix(d.data(), n, cnt);
    }
    // extension: in-memory hand-off of the pair-cost matrix (n x n row-major, symmetric)
    bool setDistanceMatrix(const double *d, int n, int cnt = -1)
    {
        city_num = n;
        dist_cnt = cnt < 0 ? n * (n - 1) / 2 : cnt;
        dis.assign(d, d + (size_t)n * n);
        best.clean();
        init_flag = true;
        return true;
    }
    void setRngMode(int m) { rng_mode = m; }
    void setSeed(uint64_t s) { seed = s; }
    const ACS_Tour &bestTour() const { return best; }
    int iterations() const { return index_itera; }

    // reference :255-284
    bool computeSolution()
    {
        if (!init_flag) return false;
        wa_ctx *ctx = weldacs_dropin::context();
        if (!ctx) return false;
        wa_gtsp_params p;
        p.rng_mode = rng_mode;
        p.seed = seed;
        p.stream = 0;
        p.max_iterations = 0;
        std::vector<int32_t> edges((size_t)city_num * 2);
        double cost = 0;
        int32_t iters = 0;
        int32_t *st = (rng_mode == WA_RNG_REF && weldacs_dropin::rand_state_valid()) ? weldacs_dropin::rand_state() : NULL;
        int rc = wa_gtsp_solve(ctx, dis.data(), city_num, dist_cnt, 1, &p, st, edges.data(), &cost, &iters, NULL);
        if (rc != WA_OK) { printf("[ACS GTSP] %s\n", wa_last_error(ctx)); return false; }
        index_itera = iters;
        best.clean();
        best.L = cost;
        for (int i = 0; i < city_num; i++) best.push_back(edges[2 * i], edges[2 * i + 1]);
        printf("Best in all = %.2lf\n", best.L);
        best.print();
        return true;
    }

    void read_all_segments(Agent<float> **&best_matrix)  // :286-298
    {
        for (int i = 0; i + 1 < (int)best.path.size(); i++) append(best_matrix[best.path[i].first][best.path[i].second]);
    }
    void read_segment(Agent<float> **best_matrix, int i)  // :303-312, i starts from 1
    {
        append(best_matrix[best.path[i - 1].first][best.path[i - 1].second]);
    }
    int path_segment_nums() { return (int)best.path.size() - 1; }
    void plot_route_path(int figureNumber)
    {
#ifdef WELDACS_WITH_MATPLOTLIB
        std::map<std::string, std::string> keywords;
        keywords.insert(std::pair<std::string, std::string>("c", "gray"));
        keywords.insert(std::pair<std::string, std::string>("marker", "o"));
        plt::scatter(g_path_x, g_path_y, g_path_z, 1, keywords, figureNumber);
#else
        (void)figureNumber;
#endif
    }
    ~ACS_GTSP() {}

private:
    int city_num = 0, dist_cnt = 0, index_itera = 0;
    int rng_mode = WA_RNG_DEV;
    uint64_t seed = 1;
    bool init_flag = false;
    std::vector<double> dis;
    ACS_Tour best;
    void append(const Agent<float> &a)
    {
        const std::vector<ACS_Node<float> *> *segment = a.getPath();
        for (size_t j = 0; j < segment->size(); j++) {
            g_path_x.push_back((*segment)[j]->pt.x);
            g_path_y.push_back((*segment)[j]->pt.y);
            g_path_z.push_back((*segment)[j]->pt.z);
        }
    }
};

#endif
