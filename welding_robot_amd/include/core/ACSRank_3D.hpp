// ACSRank_3D.hpp -- drop-in for the reference's core/ACSRank_3D.hpp on the C ABI of libweldacs.so.
// Same class names and public members (power, _Inf_of_Points_t, ACS_Node<T>, Agent<T>, ACS_Rank with
// searchBestPathOfPoints / getSolution / checkRoutePoints / setPoints / best_matrix / route_points)
// so that main.cpp:280 and ACS_GTSP::read_all_segments compile unchanged; the search itself
// (reference ACSRank_3D.hpp:134-305) runs in the HIP kernels of welding_robot_amd/csrc/.
//
// Extensions (non-breaking; the reference hard-wires all of these):
//   setRngMode(WA_RNG_REF | WA_RNG_DEV)  REF = glibc rand() stream + std::sort tie order, bit-identical
//                                        to the reference, pairs solved one after another;
//                                        DEV (default) = counter RNG, all pairs solved concurrently
//   setSeed(s)            REF: srand(s) instead of srand(time(0)) (:327); DEV: counter key
//   setMaxIteration(n)    default 150 (:322)      setFixedColony(n)  0 = reference-adaptive (:247)
//   setNeighbourhood(n)   6 (default, the reference as shipped) or 26: the reference's own stubbed variant with
//                         edge/corner moves of length precision*1.414f / precision*1.732f (:361-388)
//   setLazyEvaporation(b) default true: DEV-mode pair searches use wa_acs_create_lazy (never-deposited voxels are not
//                         swept; identical results, a generation costs O(deposited voxels) instead of 48 B/voxel)
//   setGraphFileCompat(b) true = reproduce the reference's damaged graph.in header (Q6); default writes
//                         a correct file.  The cost matrix is always also kept in memory: cost_matrix().
#ifndef _ACS_3D_HPP
#define _ACS_3D_HPP
#include <assert.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <deque>
#include <thread>
#include <set>
#include <unordered_map>

#include "model_grid_map.hpp"

#define INF_FLOAT (1.0 / 0.0)
#define INF_INT 0x3f3f3f3f
#define eps 1e-8

typedef Point3<int> Point3i;

template <class T>
struct _Inf_of_Points_t {
    _Inf_of_Points_t() {}
    _Inf_of_Points_t(T a, T b, T c) : distance(a), pheromone(b), info(c) {}
    T distance, pheromone, info;
};

// The device keeps the lattice implicit, so adjacency_* stay empty; nodes exist only for voxels that
// appear on a returned path (what main.cpp and read_all_segments dereference: ->pt, ->id, ->isFree).
template <class T>
class ACS_Node : public Vertex3<T> {
public:
    std::vector<ACS_Node<T> *> adjacency_nodes;
    std::vector<_Inf_of_Points_t<T>> adjacency_infos;
};

template <class T>
T power(T x, int y)
{
    T ans = 1;
    while (y) {
        if (y & 1) ans *= x;
        x *= x;
        y >>= 1;
    }
    return ans;
}

template <class T>
class Agent {
private:
    std::vector<ACS_Node<T> *> path;
    std::vector<int> node_index;

public:
    std::set<unsigned long int> tabu_list;
    T L;
    void addNextNode(ACS_Node<T> *node, int index, T _dis)
    {
        tabu_list.insert(node->id);
        path.push_back(node);
        node_index.push_back(index);
        L += _dis;
    }
    void addStartNode(ACS_Node<T> *_start)
    {
        tabu_list.insert(_start->id);
        path.push_back(_start);
        L = 0;
    }
    void setDeadEnd() { L = INF_FLOAT; }
    const std::vector<ACS_Node<T> *> *getPath() const { return &path; }
    const std::vector<int> *nodeIndex() const { return &node_index; }
    bool findPathNode(ACS_Node<T> *target)
    {
        for (auto node : path)
            if (target == node) return true;
        return false;
    }
    // extension: rebuild from a device result
    void assign(const std::vector<ACS_Node<T> *> &p, const std::vector<int> &idx, T cost)
    {
        path = p;
        node_index = idx;
        tabu_list.clear();
        for (auto n : p) tabu_list.insert(n->id);
        L = cost;
    }
};

class ACS_Rank : public GridMap<float> {
public:
    Agent<float> **best_matrix;
    std::vector<Point3<float>> route_points;

    ACS_Rank() : best_matrix(NULL) {}
    ~ACS_Rank() { if (solver) wa_acs_destroy(solver); }

    // ---- extensions
    void setRngMode(int mode) { rng_mode = mode; }
    void setSeed(uint64_t s) { seed = s; seeded = true; }
    void setMaxIteration(int n) { max_iteration = n; }
    void setFixedColony(int n) { fixed_colony = n; }
    void setNeighbourhood(int n) { neighbourhood = n; }
    void setLazyEvaporation(bool b) { lazy = b; }
    void setGraphFileCompat(bool b) { graph_compat = b; }
    // pair searches in flight per device.  Default 0 = sized per shard: as many as the device's free memory (wa_ctx_memory_info,
    // wa_acs_memory_estimate) and the measured footprint limit allow, then evened out over whole batches
    void setConcurrentPairs(int n) { concurrent_pairs = n; }
    // what the last searchBestPathOfPoints did per shard (device, pairs, concurrent slots, batches, summed Manhattan length)
    struct ShardReport { int device; int pairs; int slots; int batches; long long weight; };
    const std::vector<ShardReport> &lastShards() const { return shard_report; }
    // The pair loop (:472-499) is a loop over independent searches: in DEV mode it is sharded over these devices, one host
    // thread + one wa_ctx (+ a replica of the grid) per entry; the pairs are dealt longest-first (see searchBestPathOfPoints).
    // Every pair keeps its GLOBAL index as stream key, so the cost matrix and the paths do not depend on the number of shards.
    // Default: the primary context's device, then every other visible device (wa_device_count()).  An ordinal may be listed
    // twice (two contexts on one GPU).
    void setDevices(const std::vector<int> &ordinals) { devices = ordinals; devices_set = true; }
    // The contexts keep the device blocks of the solvers of earlier calls for the next call (wa_ctx_cached_bytes); this hands them back.
    void trimDeviceMemory() { weldacs_dropin::trim_device_memory(); }
    int lastStatus() const { return last_status; }
    const std::vector<float> &cost_matrix() const { return costs; }

    // reference ACSRank_3D.hpp:427-504
    void searchBestPathOfPoints(float predict_path_len = 10, std::string read_file = "", std::string output_file = "")
    {
        int point_num = 0;
        if (read_file == "") {
            std::cout << "[ACS 3D] Please enter passing point number: ";
            std::cin >> point_num;
            route_points.resize(point_num);
            std::cout << "[ACS 3D] Please enter passing point in order: " << std::endl;
            for (int i = 0; i < point_num; i++) std::cin >> route_points[i].x >> route_points[i].y >> route_points[i].z;
        } else {
            FILE *fp = fopen(read_file.c_str(), "r");
            if (fp == NULL) {
                std::cout << "[ACS 3D] Failed to read file, reject to init." << std::endl;
                last_status = WA_ERR_FILE;
                return;
            }
            if (fscanf(fp, "%d", &point_num) != 1 || point_num < 0) point_num = 0;
            route_points.resize(point_num);
            for (int i = 0; i < point_num; i++)
                if (fscanf(fp, "%f %f %f", &route_points[i].x, &route_points[i].y, &route_points[i].z) != 3) break;
            fclose(fp);
        }
        // the reference only allocates best_matrix on the file branch (:456-460) and crashes on the
        // interactive one; here both get it
        best_matrix = new Agent<float> *[point_num];
        for (int i = 0; i < point_num; i++) best_matrix[i] = new Agent<float>[point_num];
        costs.assign((size_t)point_num * point_num, 0.f);
        if (!initFromGridMap(predict_path_len)) return;
        checkRoutePoints();
        std::vector<int64_t> ids(point_num);
        std::vector<float> xyz((size_t)point_num * 3);
        for (int i = 0; i < point_num; i++) { xyz[3 * i] = route_points[i].x; xyz[3 * i + 1] = route_points[i].y; xyz[3 * i + 2] = route_points[i].z; }
        if (point_num) wa_grid_resolve_points(device_grid(), xyz.data(), point_num, ids.data());
        // all pairs i < j (:472-499); a pair with an unresolved point ends the loop like the reference
        std::vector<std::pair<int, int>> pairs;
        bool wrong = false;
        for (int i = 0; i < point_num && !wrong; i++)
            for (int j = i + 1; j < point_num; j++) {
                if (ids[i] < 0 || ids[j] < 0) {
                    printf("[ACS 3D] Wrong point : (%.3f, %.3f, %.3f) or (%.3f, %.3f, %.3f), program will exit immediately \r\n",
                           route_points[i].x, route_points[i].y, route_points[i].z, route_points[j].x, route_points[j].y, route_points[j].z);
                    wrong = true;
                    last_status = WA_ERR_POINT;
                    break;
                }
                pairs.push_back(std::make_pair(i, j));
            }
        wa_acs_params p;
        wa_acs_default_params(&p);
        p.max_iteration = max_iteration;
        p.predict = predict_path_len;
        p.fixed_colony = fixed_colony;
        p.rng_mode = rng_mode;
        p.seed = seed;
        std::vector<float> order_costs;
        // ---- the searches: one shard per device (DEV mode).  Shard 0 is the primary context (weldacs_dropin::context()), so the
        // primary's ordinal leads the list and the remaining visible devices follow
        std::vector<int> devs;
        if (rng_mode == WA_RNG_DEV) {
            if (devices_set) devs = devices;
            else {
                const int prim = weldacs_dropin::device_ordinal();
                devs.push_back(prim);
                for (int d = 0, n = wa_device_count(); d < n; d++) if (d != prim) devs.push_back(d);
            }
        }
        if (devs.size() > pairs.size()) devs.resize(pairs.size());
        if (devs.empty()) devs.push_back(weldacs_dropin::device_ordinal());
        const int D = (int)devs.size();
        std::vector<PairResult> res(pairs.size());
        std::vector<int> shard_rc(D, WA_OK);
        std::vector<std::string> shard_err(D);
        // Dealing: a search costs about as much as its walks are long, i.e. grows with the Manhattan distance of
        // its two points (in voxel steps).  Longest-processing-time-first: units sorted by weight, each to the least loaded
        // device.  A unit is an END-POINT GROUP (all pairs (i, j) with the same j: they share one heuristic field on the device,
        // ACSRank_3D.hpp:151-154, and its cache lines) while there are enough groups to balance (>= 4 per device), a single
        // pair otherwise.  No result depends on the dealing: every search draws from the stream of its GLOBAL pair index.
        std::vector<long long> wgt(pairs.size());
        for (size_t k = 0; k < pairs.size(); k++) {
            const int64_t a = ids[pairs[k].first], b = ids[pairs[k].second];
            const int64_t nxy = (int64_t)rangeX * rangeY;
            wgt[k] = 1 + llabs(a / nxy - b / nxy) + llabs((a / rangeX) % rangeY - (b / rangeX) % rangeY) + llabs(a % rangeX - b % rangeX);
        }
        std::vector<std::vector<size_t>> shard_pairs(D);
        std::vector<long long> load(D, 0);
        {
            std::vector<std::vector<size_t>> units;
            if (D > 1 && point_num - 1 >= 4 * D) {
                units.resize(point_num);
                for (size_t k = 0; k < pairs.size(); k++) units[pairs[k].second].push_back(k);
            } else {
                for (size_t k = 0; k < pairs.size(); k++) units.push_back(std::vector<size_t>(1, k));
            }
            std::vector<std::pair<long long, size_t>> order;   // (-weight, unit): heaviest first, ties by index
            for (size_t u = 0; u < units.size(); u++) {
                long long w = 0;
                for (size_t k : units[u]) w += wgt[k];
                if (!units[u].empty()) order.push_back(std::make_pair(-w, u));
            }
            std::sort(order.begin(), order.end());
            for (auto &o : order) {
                int best_d = 0;
                for (int d = 1; d < D; d++) if (load[d] < load[best_d]) best_d = d;
                for (size_t k : units[o.second]) shard_pairs[best_d].push_back(k);
                load[best_d] -= o.first;
            }
        }
        // Within a shard: end-point groups side by side (one heuristic field each), the group with the longest search first and
        // the longest search first inside a group, so that the partly filled last batch holds the shortest searches
        for (int d = 0; d < D; d++) {
            std::vector<size_t> &mine = shard_pairs[d];
            if (rng_mode == WA_RNG_REF) { std::sort(mine.begin(), mine.end()); continue; }   // the reference's order (one libc stream)
            std::vector<long long> gmax(point_num, 0);
            for (size_t k : mine) gmax[pairs[k].second] = std::max(gmax[pairs[k].second], wgt[k]);
            std::sort(mine.begin(), mine.end(), [&](size_t a, size_t b) {
                const int ea = pairs[a].second, eb = pairs[b].second;
                if (ea != eb) return gmax[ea] != gmax[eb] ? gmax[ea] > gmax[eb] : ea < eb;
                return wgt[a] != wgt[b] ? wgt[a] > wgt[b] : a < b;
            });
        }
        shard_report.assign(D, ShardReport());
        std::vector<int> shard_slots(D, 1);
        auto run_shard = [&](int d, wa_ctx *ctx, wa_acs *sv) {
            const int batch = rng_mode == WA_RNG_REF ? 1 : shard_slots[d];
            const std::vector<size_t> &mine = shard_pairs[d];
            ShardReport &rep = shard_report[d];
            rep.device = devs[d]; rep.pairs = (int)mine.size(); rep.slots = batch; rep.weight = load[d];
            rep.batches = (int)((mine.size() + batch - 1) / batch);
            for (size_t b0 = 0; b0 < mine.size(); b0 += batch) {
                int nb = (int)std::min<size_t>(batch, mine.size() - b0);
                std::vector<int64_t> s0(nb), e0(nb);
                std::vector<uint32_t> st(nb);
                // slot order inside the batch: the blocks of a walk launch start in slot order and the launch lasts as long as its longest ant, so
                // the longest searches go first; the library runs the two halves of the slots as two pipelined groups, so the sorted list is dealt
                // alternately to the halves (BASELINE config C5: 45.2 -> 43.4 ms per batch).  Results do not depend on the order (stream = pair index)
                std::vector<size_t> bord(mine.begin() + (std::ptrdiff_t)b0, mine.begin() + (std::ptrdiff_t)b0 + nb);
                if (rng_mode != WA_RNG_REF) {
                    std::sort(bord.begin(), bord.end(), [&](size_t a, size_t b) { return wgt[a] != wgt[b] ? wgt[a] > wgt[b] : a < b; });
                    std::vector<size_t> dealt;
                    for (size_t q = 1; q < bord.size(); q += 2) dealt.push_back(bord[q]);
                    for (size_t q = 0; q < bord.size(); q += 2) dealt.push_back(bord[q]);
                    bord.swap(dealt);
                }
                for (int q = 0; q < nb; q++) { size_t k = bord[(size_t)q]; s0[q] = ids[pairs[k].first]; e0[q] = ids[pairs[k].second]; st[q] = (uint32_t)k; }
                int rc = wa_acs_solve(sv, &p, nb, s0.data(), e0.data(), st.data());  // computeSolution :480
                if (rc == WA_OK) rc = wa_acs_reset_pheromone(sv, -1, p.pheromone_0);  // reset() :481
                if (rc == WA_OK) {
                    // the whole batch in one round trip (costs and lengths, then one strided copy of the paths): two calls per pair
                    // would be ~60 us of synchronise + copy each, 0.12 s for the 2 016 pairs of BASELINE config C5
                    std::vector<float> bc((size_t)nb);
                    std::vector<int64_t> bl((size_t)nb);
                    rc = wa_acs_result_batch_choices(sv, nb, bc.data(), bl.data(), NULL, NULL, 0);
                    int64_t stride = 0;
                    for (int q = 0; q < nb; q++) stride = std::max<int64_t>(stride, bl[(size_t)q]);
                    std::vector<int32_t> bi((size_t)nb * (size_t)stride);
                    std::vector<int8_t> bk((size_t)nb * (size_t)stride);
                    if (rc == WA_OK && stride > 0) rc = wa_acs_result_batch_choices(sv, nb, bc.data(), bl.data(), bi.data(), bk.data(), stride);
                    for (int q = 0; q < nb && rc == WA_OK; q++) {
                        PairResult &r = res[bord[(size_t)q]];
                        const size_t len = (size_t)bl[(size_t)q];
                        r.cost = bc[(size_t)q];
                        r.ids.assign(bi.begin() + (size_t)q * (size_t)stride, bi.begin() + (size_t)q * (size_t)stride + len);
                        r.ch.assign(bk.begin() + (size_t)q * (size_t)stride, bk.begin() + (size_t)q * (size_t)stride + len);
                    }
                }
                if (rc != WA_OK) { shard_rc[d] = rc; shard_err[d] = wa_last_error(ctx); return; }
            }
        };
        // slots of shard 0 (the others are sized on their own device when their context exists); the primary's solver is rebuilt
        // if its size changed
        if (rng_mode != WA_RNG_REF && concurrent_pairs <= 0 && solver) {
            // sized by rule from the device's FREE memory: the one-slot solver initFromGridMap built must not count against it
            wa_acs_destroy(solver); solver = NULL; slots = 0;
        }
        shard_slots[0] = rng_mode == WA_RNG_REF ? 1 : slots_for(weldacs_dropin::context(), predict_path_len, (int)shard_pairs[0].size(), shard_pairs[0], pairs);
        if (rng_mode != WA_RNG_REF && (!solver || shard_slots[0] != slots)) {
            if (solver) wa_acs_destroy(solver);
            solver = NULL;
            slots_override = shard_slots[0];
            int rc = make_solver(weldacs_dropin::context(), device_grid(), predict_path_len, &solver);
            slots_override = 0;
            if (rc != WA_OK) { printf("[ACS 3D] %s\n", wa_last_error(weldacs_dropin::context())); last_status = rc; return; }
        }
        if (D == 1) {
            run_shard(0, weldacs_dropin::context(), solver);
        } else {
            // replicas: occupancy and axis coordinates from the host mirror, one context + grid + solver per extra shard
            std::vector<uint8_t> fr((size_t)size_of_map());
            std::vector<float> ax((size_t)rangeX), ay((size_t)rangeY), az((size_t)rangeZ);
            Vertex3<float> ***m = ptr_grid_map();
            for (int z = 0, id = 0; z < rangeZ; z++)
                for (int y = 0; y < rangeY; y++)
                    for (int x = 0; x < rangeX; x++, id++) fr[(size_t)id] = m[z][y][x].isFree ? 1 : 0;
            for (int x = 0; x < rangeX; x++) ax[x] = m[0][0][x].pt.x;
            for (int y = 0; y < rangeY; y++) ay[y] = m[0][y][0].pt.y;
            for (int z = 0; z < rangeZ; z++) az[z] = m[z][0][0].pt.z;
            std::vector<wa_ctx *> cs(D, (wa_ctx *)NULL);
            std::vector<wa_grid *> gs(D, (wa_grid *)NULL);
            std::vector<wa_acs *> ss(D, (wa_acs *)NULL);
            cs[0] = weldacs_dropin::context(); ss[0] = solver;
            for (int d = 1; d < D; d++) {
                int rc = WA_OK;
                cs[d] = weldacs_dropin::shard_context(d, devs[d], &rc);   // kept for the process, with the blocks its solvers give back
                if (rc == WA_OK) rc = wa_grid_from_occupancy(cs[d], fr.data(), rangeX, rangeY, rangeZ, ax.data(), ay.data(), az.data(), precision, wall, &gs[d]);
                if (rc == WA_OK) {   // sized on the shard's own device, which may be shared with another shard
                    shard_slots[d] = slots_for(cs[d], predict_path_len, (int)shard_pairs[d].size(), shard_pairs[d], pairs);
                    slots_override = shard_slots[d];
                    rc = make_solver(cs[d], gs[d], predict_path_len, &ss[d]);
                    slots_override = 0;
                }
                if (rc != WA_OK) { shard_rc[d] = rc; shard_err[d] = cs[d] ? wa_last_error(cs[d]) : "wa_ctx_create failed"; }
            }
            std::vector<std::thread> th;
            for (int d = 1; d < D; d++)
                if (shard_rc[d] == WA_OK) th.emplace_back(run_shard, d, cs[d], ss[d]);
            run_shard(0, cs[0], ss[0]);
            for (auto &t : th) t.join();
            for (int d = 1; d < D; d++) {
                if (ss[d]) wa_acs_destroy(ss[d]);
                if (gs[d]) wa_grid_destroy(gs[d]);
            }
        }
        for (int d = 0; d < D; d++)
            if (shard_rc[d] != WA_OK) { printf("[ACS 3D] shard %d (device %d): %s\n", d, devs[d], shard_err[d].c_str()); last_status = shard_rc[d]; return; }
        for (int d = 0; d < D && rng_mode == WA_RNG_DEV; d++)
            printf("[ACS 3D] shard %d: device %d, %d pair searches in %d batch(es) of up to %d, summed Manhattan length %lld\n", d,
                   shard_report[d].device, shard_report[d].pairs, shard_report[d].batches, shard_report[d].slots, shard_report[d].weight);
        // ---- gather in the reference's pair order (`best` keeps its previous path when no ant arrived, Q9)
        for (size_t k = 0; k < pairs.size(); k++) {
            const int i = pairs[k].first, j = pairs[k].second;
            adopt_best(res[k]);
            best_matrix[i][j] = best;
            best_matrix[j][i] = best;
            costs[(size_t)i * point_num + j] = costs[(size_t)j * point_num + i] = best.L;
            order_costs.push_back(best.L);
            printf("[ACS 3D] <Point (%.3f, %.3f, %.3f) : Point (%.3f, %.3f, %.3f)> Path length: %.3f\r\n", route_points[i].x,
                   route_points[i].y, route_points[i].z, route_points[j].x, route_points[j].y, route_points[j].z, best.L);
        }
        if (rng_mode == WA_RNG_REF) {  // hand the libc stream on to ACS_GTSP, as the process-global rand() does
            wa_acs_rand_state(solver, weldacs_dropin::rand_state(), 0);
            weldacs_dropin::rand_state_valid() = true;
        }
        if (output_file != "") write_graph(output_file, point_num, order_costs);
        if (!wrong) last_status = WA_OK;
        printf("[ACS 3D] %d Result has been written to \"%s\" \r\n", (int)order_costs.size(), output_file.c_str());
    }

    const Agent<float> *getSolution() const { return &best; }

    void checkRoutePoints()  // :511-535 (diagnostic only)
    {
        int n = (int)route_points.size();
        std::vector<float> xyz((size_t)n * 3);
        std::vector<int64_t> ids(n);
        for (int i = 0; i < n; i++) { xyz[3 * i] = route_points[i].x; xyz[3 * i + 1] = route_points[i].y; xyz[3 * i + 2] = route_points[i].z; }
        if (n && device_grid()) wa_grid_resolve_points(device_grid(), xyz.data(), n, ids.data());
        for (int i = 0; i < n; i++)
            if (!device_grid() || ids[i] < 0)
                printf("[ACS 3D] Invalid route point, please reset point(%.3f, %.3f, %.3f) \n", route_points[i].x, route_points[i].y, route_points[i].z);
        printf("[ACS 3D] %d route points have been checked. \n", n);
    }

    bool setPoints(Point3<float> &start, Point3<float> &end)  // :537-565
    {
        if (!device_grid()) return false;
        float xyz[6] = {start.x, start.y, start.z, end.x, end.y, end.z};
        int64_t ids[2] = {-1, -1};
        wa_grid_resolve_points(device_grid(), xyz, 2, ids);
        start_id = ids[0];
        end_id = ids[1];
        return ids[0] >= 0 && ids[1] >= 0;
    }

    // extension: one computeSolution between the points given to setPoints (private in the reference)
    bool solveCurrent(float predict_path_len)
    {
        if (start_id < 0 || end_id < 0 || !initFromGridMap(predict_path_len)) return false;
        wa_acs_params p;
        wa_acs_default_params(&p);
        p.max_iteration = max_iteration; p.predict = predict_path_len; p.fixed_colony = fixed_colony;
        p.rng_mode = rng_mode; p.seed = seed;
        last_status = wa_acs_solve(solver, &p, 1, &start_id, &end_id, NULL);
        if (last_status != WA_OK) return false;
        fetch_best(0);
        return true;
    }

    void plot_path(Agent<float> &agentK, int figureNumber)
    {
#ifdef WELDACS_WITH_MATPLOTLIB
        const std::vector<ACS_Node<float> *> *path = agentK.getPath();
        for (auto it : *path) { path_x.push_back(it->pt.x); path_y.push_back(it->pt.y); path_z.push_back(it->pt.z); }
        std::map<std::string, std::string> keywords;
        keywords.insert(std::pair<std::string, std::string>("c", "red"));
        keywords.insert(std::pair<std::string, std::string>("linewidth", "2"));
        plt::plot3(path_x, path_y, path_z, keywords, figureNumber);
#else
        (void)agentK; (void)figureNumber;
#endif
    }
    void plot_route_point(int figureNumber)
    {
#ifdef WELDACS_WITH_MATPLOTLIB
        std::map<std::string, std::string> keywords;
        keywords.insert(std::pair<std::string, std::string>("c", "red"));
        keywords.insert(std::pair<std::string, std::string>("marker", "o"));
        for (auto &it : route_points) { path_x.push_back(it.x); path_y.push_back(it.y); path_z.push_back(it.z); }
        plt::scatter(path_x, path_y, path_z, 3, keywords, figureNumber);
#else
        (void)figureNumber;
#endif
    }

private:
    struct PairResult { float cost = 0; std::vector<int32_t> ids; std::vector<int8_t> ch; };
    std::vector<int> devices;
    bool devices_set = false;
    wa_acs *solver = NULL;
    int slots = 1;
    int rng_mode = WA_RNG_DEV, max_iteration = 150, fixed_colony = 0, concurrent_pairs = 0, neighbourhood = 6, slots_override = 0;
    std::vector<ShardReport> shard_report;
    bool lazy = true;
    uint64_t seed = 1;
    bool seeded = false, graph_compat = false;
    int last_status = WA_OK;
    int64_t start_id = -1, end_id = -1;
    Agent<float> best;
    std::vector<float> costs;
    std::vector<float> path_x, path_y, path_z;
    std::deque<ACS_Node<float>> node_pool;
    std::unordered_map<int32_t, ACS_Node<float> *> node_of;

    // initFromGridMap :317-410 -- parameters, srand(time(0)), solver for this grid
    bool initFromGridMap(float predict)
    {
        wa_ctx *ctx = weldacs_dropin::context();
        if (!ctx || !device_grid()) { last_status = WA_ERR_STATE; return false; }
        if (solver) { wa_acs_destroy(solver); solver = NULL; }
        if (!seeded) seed = (uint64_t)time(0);  // :327
        int rc = make_solver(ctx, device_grid(), predict, &solver);
        if (rc != WA_OK) { printf("[ACS 3D] %s\n", wa_last_error(ctx)); last_status = rc; return false; }
        wa_acs_init_pheromone(solver, -1, 1.0f);
        if (rng_mode == WA_RNG_REF) wa_acs_srand(solver, (uint32_t)seed);
        printf("[ACS 3D] Created %d nodes, node cubiod [x: %d, y: %d, z: %d]\r\n", size_of_map(), rangeX, rangeY, rangeZ);
        return true;
    }
    int colony_of(float predict) const
    {
        int colony = fixed_colony > 0 ? fixed_colony : (int)(0.35 * (double)predict / (double)precision);
        return colony < 1 ? 1 : colony;
    }
    bool lazy_ok(int colony) const { return lazy && rng_mode == WA_RNG_DEV && colony <= 2048 && (int)(0.2 * colony) + 1 <= 64; }   // (6 or 26 neighbours: wa_acs_create_lazy_nb)
    // ... and where it pays.  Measured (round 6, tools/lazy_crossover.py -> profiles/r06/lazy_crossover.txt: 24-ant pair planning, 150 generations,
    // 32^3 .. 256^3, 1 .. 21 concurrent searches, seconds per plan dense / lazy): the lazy field wins from three searches on at every size (0.44-0.58 x at
    // three, 0.07 x for 21 searches at 256^3) and for a lone search from 128^3 on (0.96 x; 192^3 0.67, 256^3 0.42); a LONE search on a grid of up to
    // 64^3 is where the dense sweep is cheaper than the lazy walk's stamp loads (1.22 x).  (Rounds 2-5 took the lazy field whenever it was allowed.)
    bool lazy_for(int colony, int64_t n_searches)
    {
        if (!lazy_ok(colony)) return false;
        int32_t dims[3] = {0, 0, 0};
        if (n_searches <= 1 && wa_grid_info(device_grid(), dims, NULL, NULL, NULL) == WA_OK && (int64_t)dims[0] * dims[1] * dims[2] <= ((int64_t)1 << 19)) return false;
        return true;
    }
    // Concurrent pair searches of a shard with `n_pairs` searches.  Upper bound from memory: 3/4 of what the device has free
    // (ctx == NULL: the primary context), and never more than ~200 GB of fields -- past that footprint the walk's random record
    // loads slow down (measured on BASELINE config C5: 224 slots of 0.85 GB run 2 016 searches in 0.57 s, 252 in 1.0 s).  The
    // heuristic pool holds one field per distinct end point of a batch (at most 8 are assumed here: the shard's order keeps
    // end-point groups together).  Then whole batches: ceil(n / cap) batches of equal size instead of full ones + a remainder.
    int slots_for(wa_ctx *ctx, float predict, int n_pairs, const std::vector<size_t> &mine, const std::vector<std::pair<int, int>> &pairs)
    {
        if (n_pairs < 1) return 1;
        if (concurrent_pairs > 0) return std::min(concurrent_pairs, n_pairs);
        if (!ctx) ctx = weldacs_dropin::context();
        const int colony = colony_of(predict);
        int64_t per_slot = 0, per_field = 0, fixed = 0, free_b = 0, total_b = 0;
        const bool lz = lazy_for(colony, n_pairs);
        if (wa_acs_memory_estimate(device_grid(), colony, 0, neighbourhood, lz ? 1 : 0, &per_slot, &per_field, &fixed) != WA_OK ||
            wa_ctx_memory_info(ctx, &free_b, &total_b) != WA_OK || per_slot <= 0) return std::min(16, n_pairs);
        per_slot += 20 * (int64_t)max_iteration;                      // the per-generation trace
        std::vector<int> ends;
        for (size_t k : mine) if (std::find(ends.begin(), ends.end(), pairs[k].second) == ends.end()) ends.push_back(pairs[k].second);
        const int64_t fields = std::min<int64_t>(std::max<int64_t>(4, (int64_t)ends.size()), 8);
        int64_t budget = std::min<int64_t>(free_b / 4 * 3, (int64_t)200e9) - fixed - fields * per_field;
        int64_t cap = budget / per_slot;
        // a launch of more walk blocks than three rounds of resident wavefronts (one per SIMD x 256 CUs x 4, two with a 2^12 hash)
        // only queues: bound the slots by that too
        cap = std::min<int64_t>(cap, std::max<int64_t>(1, (int64_t)3 * 2048 / colony));
        if (cap < 1) cap = 1;
        // small dense solvers also hold straggler pools per slot (wa_acs_straggler_pool_bytes; none for lazy or > 16 slots)
        while (cap > 1) {
            int64_t pools = 0;
            if (wa_acs_straggler_pool_bytes(device_grid(), (int32_t)std::min<int64_t>(cap, n_pairs), colony, 0, neighbourhood, lz ? 1 : 0, &pools) != WA_OK ||
                pools <= budget - cap * per_slot) break;
            cap--;
        }
        const int64_t batches = (n_pairs + cap - 1) / cap;
        return (int)((n_pairs + batches - 1) / batches);
    }
    // one solver for this grid on `ctx` (the primary context or a shard's)
    int make_solver(wa_ctx *ctx, wa_grid *g, float predict, wa_acs **out)
    {
        const int colony = colony_of(predict);
        // (initFromGridMap builds the primary's solver before the pairs are known: one slot; the pair loop re-creates it at the
        // shard's size)
        const int want = rng_mode == WA_RNG_REF ? 1 : (slots_override > 0 ? slots_override : std::max(1, concurrent_pairs));
        if (ctx == weldacs_dropin::context()) slots = want;
        int rc = lazy_for(colony, want) ? wa_acs_create_lazy_nb(ctx, g, want, colony, 0, neighbourhood, out) : wa_acs_create_nb(ctx, g, want, colony, 0, neighbourhood, out);
        if (rc == WA_OK) rc = wa_acs_init_pheromone(*out, -1, 1.0f);
        return rc;
    }
    void adopt_best(const PairResult &r)
    {
        const size_t len = r.ids.size();
        if (len > 0) {
            std::vector<ACS_Node<float> *> p(len);
            std::vector<int> idx(len - 1);
            for (size_t i = 0; i < len; i++) p[i] = node(r.ids[i]);
            for (size_t i = 0; i + 1 < len; i++) idx[i] = r.ch[i];
            best.assign(p, idx, r.cost);
        } else {
            best.L = r.cost;  // +inf: path left as it was (Q9)
        }
    }
    ACS_Node<float> *node(int32_t id)
    {
        auto it = node_of.find(id);
        if (it != node_of.end()) return it->second;
        node_pool.emplace_back();
        ACS_Node<float> *n = &node_pool.back();
        const Vertex3<float> &v = ptr_grid_map()[id / (rangeX * rangeY)][(id / rangeX) % rangeY][id % rangeX];
        n->pt = v.pt; n->isFree = v.isFree; n->id = v.id;
        node_of[id] = n;
        return n;
    }
    void fetch_best(int slot)
    {
        float cost = 0;
        int64_t len = 0;
        wa_acs_result(solver, slot, &cost, &len, NULL, NULL, 0);
        if (len > 0) {
            std::vector<int32_t> ids((size_t)len);
            std::vector<int8_t> ch((size_t)len);
            wa_acs_result(solver, slot, &cost, &len, ids.data(), ch.data(), len);
            std::vector<ACS_Node<float> *> p((size_t)len);
            std::vector<int> idx((size_t)len - 1);
            for (int64_t i = 0; i < len; i++) p[i] = node(ids[i]);
            for (int64_t i = 0; i + 1 < len; i++) idx[i] = ch[i];
            best.assign(p, idx, cost);
        } else {
            best.L = cost;  // +inf: path left as it was (Q9)
        }
    }
    void write_graph(const std::string &file, int point_num, const std::vector<float> &c)
    {
        FILE *fp = fopen(file.c_str(), "w");
        if (!fp) return;
        if (graph_compat) {  // :470,:488,:500-501 byte for byte
            fprintf(fp, "%d %d\n", 0, 0);
            for (float v : c) fprintf(fp, "%.3f\n", v);
            rewind(fp);
            fprintf(fp, "%d %d\r", point_num, (int)c.size());
        } else {
            fprintf(fp, "%d %d\n", point_num, (int)c.size());
            for (float v : c) fprintf(fp, "%.9g\n", (double)v);
        }
        fclose(fp);
    }
};

#endif
