// shard_check.cpp -- test helper: ACS_Rank::searchBestPathOfPoints (ACSRank_3D.hpp:427-504) sharded over a device list.
//   shard_check <stl> <precision> <wall> <points.in> <predict> <seed> <devices: "all" | "0" | "0,0" ...> <dump.txt> [slots: 0 = sized by rule] [calls]
// dump: one line per ordered pair i<j: "pair i j COSTBITS len id id id ..." + the cost matrix; <dump>.shards: one line per shard
// "shard d device pairs slots batches weight" (ACS_Rank::lastShards()); <dump>.cache: one line per call "call k bytes bytes ..." = what the
// primary and the further shards' contexts keep after the call (wa_ctx_cached_bytes), then "trimmed bytes ..." after trimDeviceMemory()
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "core/ACSRank_3D.hpp"
#include "core/read_STL.hpp"

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: shard_check stl p wall points predict seed devices dump\n"); return 1; }
    STLReader model;
    if (!model.readFile(argv[1])) return 2;
    ACS_Rank sp;
    if (!sp.creatGridMap(model.TriangleList(), (float)atof(argv[2]), atoi(argv[3]), "")) return 3;
    sp.setRngMode(WA_RNG_DEV);
    sp.setSeed((uint64_t)atoll(argv[6]));
    sp.setMaxIteration(60);
    sp.setConcurrentPairs(argc > 9 ? atoi(argv[9]) : 3);
    if (strcmp(argv[7], "all")) {
        std::vector<int> d;
        for (char *t = strtok(argv[7], ","); t; t = strtok(NULL, ",")) d.push_back(atoi(t));
        sp.setDevices(d);
    }
    const int calls = argc > 10 ? atoi(argv[10]) : 1;
    FILE *fc = fopen((std::string(argv[8]) + ".cache").c_str(), "w");
    if (!fc) return 5;
    auto kept = [&](const char *tag, int k) {
        fprintf(fc, "%s %d", tag, k);
        int64_t b = 0;
        if (weldacs_dropin::ctx_slot() && wa_ctx_cached_bytes(weldacs_dropin::ctx_slot(), &b) == WA_OK) fprintf(fc, " %lld", (long long)b);
        for (auto &sc : weldacs_dropin::shard_slots())
            if (sc.ctx && wa_ctx_cached_bytes(sc.ctx, &b) == WA_OK) fprintf(fc, " %lld", (long long)b);
        fprintf(fc, "\n");
    };
    for (int k = 0; k < calls; k++) {
        sp.searchBestPathOfPoints((float)atof(argv[5]), argv[4], "");
        if (sp.lastStatus() != WA_OK) return 4;
        kept("call", k);
    }
    sp.trimDeviceMemory();
    kept("trimmed", calls);
    fclose(fc);
    FILE *fp = fopen(argv[8], "w");
    if (!fp) return 5;
    const int P = (int)sp.route_points.size();
    for (int i = 0; i < P; i++)
        for (int j = i + 1; j < P; j++) {
            const Agent<float> &a = sp.best_matrix[i][j];
            unsigned u;
            memcpy(&u, &a.L, 4);
            fprintf(fp, "pair %d %d %08x %d", i, j, u, (int)a.getPath()->size());
            for (auto n : *a.getPath()) fprintf(fp, " %lu", n->id);
            fprintf(fp, "\n");
        }
    for (float c : sp.cost_matrix()) { unsigned u; memcpy(&u, &c, 4); fprintf(fp, "%08x ", u); }
    fprintf(fp, "\n");
    fclose(fp);
    fp = fopen((std::string(argv[8]) + ".shards").c_str(), "w");
    if (!fp) return 5;
    int d = 0;
    for (const ACS_Rank::ShardReport &r : sp.lastShards()) fprintf(fp, "shard %d %d %d %d %d %lld\n", d++, r.device, r.pairs, r.slots, r.batches, r.weight);
    fclose(fp);
    return 0;
}
