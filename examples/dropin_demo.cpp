// examples/dropin_demo.cpp -- the planning sequence of the reference's main.cpp:273-283, compiled
// against the drop-in headers (welding_robot_amd/include/core) and libweldacs.so instead of the
// reference's header-only classes.  The five calls in the middle are verbatim main.cpp.
//
//   g++ -std=c++14 -Iinclude -Iwelding_robot_amd/include examples/dropin_demo.cpp
//       -Lwelding_robot_amd/lib -lweldacs -Wl,-rpath,$PWD/welding_robot_amd/lib -o dropin_demo
//   ./dropin_demo cubic.stl 0.0219 8 points.in 0.5 graph.in [ref SEED | dev SEED] [out.txt]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "core/ACSRank_3D.hpp"
#include "core/read_STL.hpp"
#include "core/ACS_GTSP.hpp"

STLReader model;
ACS_Rank SearchPath;
ACS_GTSP GlobalRoute;

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s model.stl precision wall points.in predict graph.in [ref|dev seed] [out.txt]\n", argv[0]);
        return 2;
    }
    const char *mode = argc > 7 ? argv[7] : "dev";
    unsigned long seed = argc > 8 ? strtoul(argv[8], NULL, 10) : 1;
    if (!strcmp(mode, "ref")) {  // reproduce the reference bit for bit, including its graph.in damage (Q6)
        SearchPath.setRngMode(WA_RNG_REF);
        SearchPath.setGraphFileCompat(true);
        GlobalRoute.setRngMode(WA_RNG_REF);
    }
    SearchPath.setSeed(seed);
    GlobalRoute.setSeed(seed);

    if (!model.readFile(argv[1])) return 1;                                   // main.cpp:273
    const std::vector<Triangles<float>> meshes = model.TriangleList();         // :274
    SearchPath.creatGridMap(meshes, strtof(argv[2], NULL), atoi(argv[3]), ""); // :279
    SearchPath.searchBestPathOfPoints(strtof(argv[5], NULL), argv[4], argv[6]);// :280
    if (SearchPath.lastStatus() != WA_OK) return 3;
    GlobalRoute.readFromGraphFile(argv[6]);                                    // :281
    GlobalRoute.computeSolution();                                             // :282
    GlobalRoute.read_all_segments(SearchPath.best_matrix);                     // :283

    FILE *out = argc > 9 ? fopen(argv[9], "w") : stdout;
    int P = (int)SearchPath.route_points.size();
    fprintf(out, "points %d\n", P);
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) fprintf(out, "cost %d %d %.9g\n", i, j, i == j ? 0.0 : (double)SearchPath.best_matrix[i][j].L);
    fprintf(out, "tour_L %.17g iters %d\n", GlobalRoute.bestTour().L, GlobalRoute.iterations());
    for (auto &e : GlobalRoute.bestTour().path) fprintf(out, "edge %d %d\n", e.first, e.second);
    fprintf(out, "gpath %d\n", (int)GlobalRoute.g_path_x.size());
    for (size_t i = 0; i < GlobalRoute.g_path_x.size(); i++)
        fprintf(out, "%.9g %.9g %.9g\n", (double)GlobalRoute.g_path_x[i], (double)GlobalRoute.g_path_y[i], (double)GlobalRoute.g_path_z[i]);
    if (out != stdout) fclose(out);
    return 0;
}
