// examples/scalar_calls.cpp -- what the SCALAR calls of the drop-in cost, and what main.cpp's clock()-paced sampling loops
// (:302-316, :341-351) collect with them.  The reference calls getCurvePoint once per sample (~100 ns of scalar work,
// BSplineBasic.h:85-112); the drop-in evaluates such single points on the host from a mirror of the spline (wa_bspline_eval_host,
// bit-identical to the kernel) instead of a launch + synchronise + copy per call.  Reported per call: getCurvePoint on the host path
// and through the device, getCurveDerPoint, ACS_Rank::setPoints (a point-resolve kernel), wa_acs_result (a synchronise + two copies).
//
//   g++ -std=c++14 -O1 -Iinclude -Iwelding_robot_amd/include examples/scalar_calls.cpp -Lwelding_robot_amd/lib -lweldacs
//       -Wl,-rpath,$PWD/welding_robot_amd/lib -o scalar_calls
//   ./scalar_calls model.stl precision wall [out.txt]
// out: "key value" lines (us per call, sample counts of the two loop shapes on either path, mismatches between the paths).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <chrono>
#include <vector>

#include "core/ACSRank_3D.hpp"
#include "core/read_STL.hpp"
#include "core/BSplineBasic.h"

STLReader model;
ACS_Rank SearchPath;

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the shape of main.cpp:302-316: a sample whenever `pace` ticks of clock() have passed, until `until` ticks
template <class Curve>
static int paced_loop(Curve &curve, long pace, long until, std::vector<float> &xyz)
{
    clock_t base_t = clock();
    clock_t now_t = clock() - base_t;
    float res[3] = {0, 0, 0};
    int n = 0;
    do {
        if (clock() - base_t - now_t >= pace) {
            now_t = clock() - base_t;
            curve.getCurvePoint((float)now_t, res);
            xyz.push_back(res[0]); xyz.push_back(res[1]); xyz.push_back(res[2]);
            n++;
        }
    } while (now_t <= until);
    return n;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s model.stl precision wall [out.txt]\n", argv[0]); return 2; }
    FILE *out = argc > 4 ? fopen(argv[4], "w") : stdout;
    if (!out) return 2;
    if (!model.readFile(argv[1])) return 1;
    const std::vector<Triangles<float>> meshes = model.TriangleList();
    SearchPath.creatGridMap(meshes, strtof(argv[2], NULL), atoi(argv[3]), "");
    // (gridStatus() is GridMap's verdict on creatGridMap; lastStatus() is ACS_Rank's own, about the searches -- found by the sanitizer
    //  leg, tests/test_sanitizers.py: without a device the wrong one let the program run on into a null grid)
    if (SearchPath.gridStatus() != WA_OK || !SearchPath.ptr_grid_map()) { printf("grid map: status %d\n", SearchPath.gridStatus()); return 3; }

    // ---- a polyline of 400 control points through the grid's bounding box, as main.cpp hands the stitched path to BS_Basic<float,3,0,0,0>
    const int pt_num = 400;
    float **ctrl = new float *[pt_num];
    for (int i = 0; i < pt_num; i++) {
        ctrl[i] = new float[3];
        ctrl[i][0] = 0.01f * (float)i; ctrl[i][1] = 0.5f + 0.003f * (float)((i * 37) % 101); ctrl[i][2] = 1.0f - 0.002f * (float)i;
    }
    BS_Basic<float, 3, 0, 0, 0> first(pt_num);
    first.SetParam(ctrl[0], ctrl[pt_num - 1], ctrl, 150);
    float second_start[9] = {ctrl[0][0], ctrl[0][1], ctrl[0][2], 0, 0, 0, 0, 0, 0};
    float second_end[9] = {ctrl[pt_num - 1][0], ctrl[pt_num - 1][1], ctrl[pt_num - 1][2], 0, 0, 0, 0, 0, 0};
    BS_Basic<float, 3, 2, 2, 2> second(pt_num);
    second.SetParam(second_start, second_end, ctrl, 6000);

    // ---- per-call cost; host path against device path, bit for bit
    float res[3], ref[3];
    int mismatches = 0;
    const int NH = 200000, ND = 300;
    first.getCurvePoint(1.0f, res);            // (the first host call fetches the mirror)
    second.getCurvePoint(1.0f, res);
    double t0 = now_us();
    float acc = 0;
    for (int i = 0; i < NH; i++) { second.getCurvePoint(6000.0f * (float)i / (float)NH, res); acc += res[0]; }
    const double us_host = (now_us() - t0) / NH;
    t0 = now_us();
    for (int i = 0; i < NH; i++) { second.getCurveDerPoint(6000.0f * (float)i / (float)NH, 1, res); acc += res[0]; }
    const double us_host_der = (now_us() - t0) / NH;
    second.setHostEvaluation(false);
    second.getCurvePoint(1.0f, res);
    t0 = now_us();
    for (int i = 0; i < ND; i++) { second.getCurvePoint(6000.0f * (float)i / (float)ND, res); acc += res[0]; }
    const double us_dev = (now_us() - t0) / ND;
    for (int i = 0; i <= 2000; i++) {          // the two paths agree on every bit, end points and out-of-range times included
        const float u = -5.0f + 6010.0f * (float)i / 2000.0f;
        for (int d = 0; d <= 2; d++) {
            second.setHostEvaluation(false);
            bool okd = d ? second.getCurveDerPoint(u, d, ref) : second.getCurvePoint(u, ref);
            second.setHostEvaluation(true);
            bool okh = d ? second.getCurveDerPoint(u, d, res) : second.getCurvePoint(u, res);
            if (okd != okh || (okd && memcmp(ref, res, sizeof ref) != 0)) mismatches++;
        }
    }
    fprintf(out, "getCurvePoint_host_us %.4f\ngetCurveDerPoint_host_us %.4f\ngetCurvePoint_device_us %.3f\nhost_device_mismatches %d\n", us_host, us_host_der, us_dev, mismatches);

    // ---- the two sampling loops of main.cpp, host path and device path
    std::vector<float> xyz;
    first.setHostEvaluation(true); second.setHostEvaluation(true);
    const int n1h = paced_loop(first, 10, 150, xyz), n2h = paced_loop(second, 50, 6000, xyz);
    first.setHostEvaluation(false); second.setHostEvaluation(false);
    const int n1d = paced_loop(first, 10, 150, xyz), n2d = paced_loop(second, 50, 6000, xyz);
    fprintf(out, "loop1_samples_host %d\nloop2_samples_host %d\nloop1_samples_device %d\nloop2_samples_device %d\n", n1h, n2h, n1d, n2d);
    fprintf(out, "loop1_samples_ideal %d\nloop2_samples_ideal %d\n", 150 / 10 + 1, 6000 / 50 + 1);

    // ---- setPoints (ACSRank_3D.hpp:537-565) and the best-path read-back
    Vertex3<float> ***gm = SearchPath.ptr_grid_map();
    const int rx = SearchPath.rangeX, ry = SearchPath.rangeY, rz = SearchPath.rangeZ;
    Point3<float> a = gm[1][1][1].pt, b = gm[rz - 2][ry - 2][rx - 2].pt;
    SearchPath.setPoints(a, b);
    const int NS = 200;
    t0 = now_us();
    for (int i = 0; i < NS; i++) SearchPath.setPoints(a, b);
    const double us_set = (now_us() - t0) / NS;
    fprintf(out, "setPoints_us %.3f\n", us_set);
    fprintf(out, "checksum %g\n", (double)acc);
    if (out != stdout) fclose(out);
    return 0;
}
