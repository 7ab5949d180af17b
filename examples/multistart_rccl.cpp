// multistart_rccl.cpp -- BASELINE config C4 from a C++ host, no Python, no torch, no MPI:
// one independent 128^3 search per GPU (grid seed 2024 + rank, colony seed 12345 + rank), one host thread + one
// wa_ctx + one wa_comm per device, and the global-best path cost of every generation (what each rank's
// ACSRank_3D.hpp:263-264 publishes) MIN-all-reduced over RCCL in chunks that overlap with the next generations.
// This is the shape a multi-start main.cpp:268-283 takes on an 8 x MI355X node.
//
//   multistart_rccl <grid n> <ants> <generations> <devices: "all" | "0" | "0,1,..."> <dump.txt>
// dump: "local r g BESTBITS" per rank and generation, "global g BESTBITS OWNER_RANK OWNER_SLOT" per generation, "gens_per_s X",
// then the path that achieved the final global best, fetched from its owner: "owner_path RANK N" + N node ids (wa_comm_gather_paths).
// g++ -std=c++14 -I include examples/multistart_rccl.cpp -L welding_robot_amd/lib -lweldacs -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "weldacs.h"

// welding_robot_amd/synth.py: iid Bernoulli(0.10) occupancy from a splitmix64 stream in raster order, corner blocks free
static void synth_grid(int n, uint64_t seed, std::vector<uint8_t> &free_)
{
    const size_t tot = (size_t)n * n * n;
    free_.resize(tot);
    for (size_t i = 0; i < tot; i++) {
        uint64_t z = seed + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
        free_[i] = u >= 0.10 ? 1 : 0;
    }
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < 2; y++)
            for (int x = 0; x < 2; x++) {
                free_[((size_t)z * n + y) * n + x] = 1;
                free_[((size_t)(n - 2 + z) * n + (n - 2 + y)) * n + (n - 2 + x)] = 1;
            }
}

struct Rank {
    int rc = WA_OK;
    std::string err;
    std::vector<float> local, global;
    std::vector<int32_t> owner_rank, owner_slot;
    std::vector<int32_t> gathered_index, gathered_ids;   // rank 0: every rank's best path (index = rank)
    std::vector<int64_t> gathered_len;
    double seconds = 0;
};

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: multistart_rccl n ants generations devices dump\n"); return 1; }
    const int n = atoi(argv[1]), ants = atoi(argv[2]), K = atoi(argv[3]), chunk = 50;
    std::vector<int> devs;
    if (!strcmp(argv[4], "all")) for (int d = 0; d < wa_device_count(); d++) devs.push_back(d);
    else for (char *t = strtok(argv[4], ","); t; t = strtok(NULL, ",")) devs.push_back(atoi(t));
    const int W = (int)devs.size();
    if (W < 1) { printf("no HIP device: libweldacs has no CPU fallback\n"); return 2; }
    uint8_t id[WA_COMM_ID_BYTES];
    if (wa_comm_unique_id(id) != WA_OK) { printf("wa_comm_unique_id failed\n"); return 2; }
    std::vector<Rank> R(W);
    auto run = [&](int r) {
        Rank &me = R[r];
        wa_ctx *ctx = NULL; wa_grid *g = NULL; wa_acs *s = NULL; wa_comm *c = NULL;
        auto bail = [&](int rc) { me.rc = rc; me.err = ctx ? wa_last_error(ctx) : "wa_ctx_create failed"; };
        int rc = wa_ctx_create(devs[r], &ctx);
        if (rc) return bail(rc);
        // every rank enters the collective wa_comm_create even if something below fails later: create it first
        rc = wa_comm_create(ctx, r, W, id, &c);
        if (rc) return bail(rc);
        std::vector<uint8_t> fr;
        synth_grid(n, 2024 + (uint64_t)r, fr);
        std::vector<float> ax((size_t)n);
        for (int i = 0; i < n; i++) ax[i] = (float)i;
        rc = wa_grid_from_occupancy(ctx, fr.data(), n, n, n, ax.data(), ax.data(), ax.data(), 1.0f, 0, &g);
        if (!rc) rc = wa_acs_create(ctx, g, 1, ants, 0, &s);
        if (!rc) rc = wa_acs_init_pheromone(s, -1, 1.0f);
        int64_t ids[2] = {-1, -1};
        const float pts[6] = {0, 0, 0, (float)(n - 1), (float)(n - 1), (float)(n - 1)};
        if (!rc) rc = wa_grid_resolve_points(g, pts, 2, ids);
        wa_acs_params p;
        wa_acs_default_params(&p);
        p.max_iteration = K; p.fixed_colony = ants; p.predict = (float)(ants / 0.35); p.rng_mode = WA_RNG_DEV; p.seed = 12345 + (uint64_t)r;
        const uint32_t stream = (uint32_t)r;
        if (!rc) rc = wa_acs_begin(s, &p, 1, &ids[0], &ids[1], &stream);
        if (!rc) rc = wa_comm_barrier(c);
        const auto t0 = std::chrono::steady_clock::now();
        for (int done = 0; done < K && !rc; done += chunk) {
            const int cnt = K - done < chunk ? K - done : chunk;
            rc = wa_acs_run(s, cnt);                                   // computeSolution's generations (:237-299), asynchronous
            if (!rc) rc = wa_acs_allreduce_best(s, c, done, cnt);      // overlaps with the next chunk
        }
        if (!rc) rc = wa_acs_sync(s);
        me.global.resize((size_t)K);
        me.local.resize((size_t)K);
        me.owner_rank.resize((size_t)K);
        me.owner_slot.resize((size_t)K);
        if (!rc) rc = wa_comm_read_best_owner(c, 0, K, me.global.data(), me.owner_rank.data(), me.owner_slot.data());
        if (!rc) rc = wa_comm_barrier(c);
        me.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        int32_t gd = 0;
        if (!rc) rc = wa_acs_trace(s, 0, &gd, me.local.data(), NULL, NULL, NULL, NULL);
        if (!rc) rc = wa_comm_allreduce_f64(c, &me.seconds, 1, WA_COMM_MAX);   // the slowest rank's time
        // the path behind the global best lives on its owner: every rank hands its best path (index = its rank) to rank 0
        float cost = 0;
        int64_t len = 0;
        std::vector<int32_t> path;
        if (!rc) rc = wa_acs_result(s, 0, &cost, &len, NULL, NULL, 0);
        path.resize((size_t)len);
        if (!rc && len > 0) rc = wa_acs_result(s, 0, &cost, &len, path.data(), NULL, len);
        int64_t np = 0, ni = 0;
        const int32_t my_index = r;
        if (!rc) rc = wa_comm_gather_paths(c, 0, len > 0 ? 1 : 0, &my_index, &len, path.data(), &np, &ni);
        if (!rc && r == 0) {
            me.gathered_index.resize((size_t)np); me.gathered_len.resize((size_t)np); me.gathered_ids.resize((size_t)ni);
            rc = wa_comm_gathered_paths_read(c, me.gathered_index.data(), me.gathered_len.data(), me.gathered_ids.data());
        }
        if (rc) bail(rc);
        if (s) wa_acs_destroy(s);
        if (g) wa_grid_destroy(g);
        if (c) wa_comm_destroy(c);
        wa_ctx_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int r = 1; r < W; r++) th.emplace_back(run, r);
    run(0);
    for (auto &t : th) t.join();
    for (int r = 0; r < W; r++)
        if (R[r].rc) { printf("rank %d (device %d): status %d: %s\n", r, devs[r], R[r].rc, R[r].err.c_str()); return 3; }
    FILE *fp = fopen(argv[5], "w");
    if (!fp) return 4;
    auto bitsof = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
    for (int r = 0; r < W; r++)
        for (int g = 0; g < K; g++) fprintf(fp, "local %d %d %08x\n", r, g, bitsof(R[r].local[g]));
    for (int g = 0; g < K; g++) fprintf(fp, "global %d %08x %d %d\n", g, bitsof(R[0].global[g]), R[0].owner_rank[g], R[0].owner_slot[g]);
    fprintf(fp, "gens_per_s %.3f\n", (double)K * W / R[0].seconds);
    {   // the final global best's path, as rank 0 received it from its owner
        const int owner = R[0].owner_rank[K - 1];
        size_t off = 0;
        for (size_t i = 0; i < R[0].gathered_index.size(); i++) {
            if (R[0].gathered_index[i] == owner) {
                fprintf(fp, "owner_path %d %lld", owner, (long long)R[0].gathered_len[i]);
                for (int64_t q = 0; q < R[0].gathered_len[i]; q++) fprintf(fp, " %d", R[0].gathered_ids[off + (size_t)q]);
                fprintf(fp, "\n");
            }
            off += (size_t)R[0].gathered_len[i];
        }
    }
    fclose(fp);
    printf("[multistart] %d rank(s), %d generations each, %.1f generations/s whole job, global best %.3f\n", W, K, (double)K * W / R[0].seconds, R[0].global[K - 1]);
    return 0;
}
