// model_grid_map.hpp -- drop-in for the reference's core/model_grid_map.hpp, written on the C ABI
// of libweldacs.so (include/weldacs.h).  Same type names and public members (Point3, Triangles,
// Vertex3, GridMap<T>::creatGridMap / readGridMap / ptr_grid_map / size_of_map, the public fields
// precision, wall, rangeX/Y/Z) so that main.cpp:279 compiles unchanged; the voxelisation itself
// (reference model_grid_map.hpp:165-268) runs in the k_voxelize_clip HIP kernel.
//
// Differences a maintainer should know (all opt-outs of reference quirks):
//  * no #include "matplotlibcpp.h": plot_grid_map/show_plot are no-ops unless the translation unit
//    defines WELDACS_WITH_MATPLOTLIB before including this header (then the reference's calls run);
//  * the grid file written by creatGridMap(..., file) carries the TRUE mesh bbox and %.9g precision,
//    so readGridMap round-trips (the reference writes the last triangle's bbox, Q5);
//  * objects are re-usable (the reference asserts on a second creatGridMap, Q10).
#ifndef _MODEL_GRID_MAP_HPP
#define _MODEL_GRID_MAP_HPP
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "weldacs_dropin.h"
#ifdef WELDACS_WITH_MATPLOTLIB
#include "matplotlibcpp.h"
namespace plt = matplotlibcpp;
#endif

#define sqr(x) ((x) * (x))
#define my_abs(x) ((x) > 0 ? (x) : -(x))

enum class GRID_PROGRESS { READ_MESHES, READ_GRID_MAP, CREATED_NODES, OPERATING_MESHES, WRITING_FILE };

template <class T>
class Point3 {
public:
    Point3() : x(0), y(0), z(0) {}
    Point3(T _x, T _y, T _z) : x(_x), y(_y), z(_z) {}
    T x, y, z;
    Point3<T> operator+(const Point3<T> &b) { return Point3<T>(x + b.x, y + b.y, z + b.z); }
    Point3<T> operator-(const Point3<T> &b) { return Point3<T>(x - b.x, y - b.y, z - b.z); }
    T norm() { return sqrt(sqr(x) + sqr(y) + sqr(z)); }
    static T manhattan_distance(const Point3<T> &a, const Point3<T> &b) { return (a.x - b.x) + (a.y - b.y) + (a.z - b.z); }
    static T euler_distance(const Point3<T> &a, const Point3<T> &b) { return sqrt(sqr(a.x - b.x) + sqr(a.y - b.y) + sqr(a.z - b.z)); }
    static T dot(const Point3<T> &a, const Point3<T> &b) { return (a.x * b.x + a.y * b.y + a.z * b.z); }
};

template <class T>
struct Triangles {
public:
    Point3<T> nor_vec;
    Point3<T> vertex[3];
    int trait;
};

template <class T>
class Vertex3 {
public:
    Point3<T> pt;
    bool isFree;
    unsigned long int id;
};


template <class T>
class GridMap {
public:
    Vertex3<T> ***creatGridMap(const std::vector<Triangles<T>> &mesh, T _precision, int _wall, std::string file_name = "")
    {
        release();
        precision = _precision;
        wall = _wall;
        wa_ctx *ctx = weldacs_dropin::context();
        if (!ctx || mesh.empty()) return NULL;
        std::vector<float> tris(mesh.size() * 12);
        for (size_t i = 0; i < mesh.size(); i++) {
            float *t = &tris[i * 12];
            t[0] = mesh[i].nor_vec.x; t[1] = mesh[i].nor_vec.y; t[2] = mesh[i].nor_vec.z;
            for (int v = 0; v < 3; v++) { t[3 + 3 * v] = mesh[i].vertex[v].x; t[4 + 3 * v] = mesh[i].vertex[v].y; t[5 + 3 * v] = mesh[i].vertex[v].z; }
        }
        float bbox[6];
        int rc = wa_grid_from_mesh(ctx, tris.data(), (int64_t)mesh.size(), precision, wall, &grid_h, bbox);
        last_status = rc;
        if (rc != WA_OK) { printf("[Grid Map] %s\n", wa_last_error(ctx)); return NULL; }
        min_x = bbox[0]; min_y = bbox[1]; min_z = bbox[2]; max_x = bbox[3]; max_y = bbox[4]; max_z = bbox[5];
        {   // what the reference's members hold when it writes its file: the LAST triangle's bounding box
            // +- precision (model_grid_map.hpp:228-248 reuse min_*/max_* as scratch inside the triangle loop)
            const Triangles<T> &t = mesh.back();
            T lo[3] = {t.vertex[0].x, t.vertex[0].y, t.vertex[0].z}, hi[3] = {lo[0], lo[1], lo[2]};
            for (int i = 0; i < 3; i++) {
                const T c[3] = {t.vertex[i].x, t.vertex[i].y, t.vertex[i].z};
                for (int a = 0; a < 3; a++) { hi[a] = c[a] > hi[a] ? c[a] : hi[a]; lo[a] = c[a] < lo[a] ? c[a] : lo[a]; }
            }
            for (int a = 0; a < 3; a++) { q5_lo[a] = lo[a] - precision; q5_hi[a] = hi[a] + precision; }
        }
        printf("[Grid Map]max(%.2f, %.2f, %.2f), min(%.2f, %.2f, %.2f) \n", max_x, max_y, max_z, min_x, min_y, min_z);
        printf("[Grid Map] %d triangles is scanned... \n", (int)mesh.size());
        materialise();
        printf("[Grid Map] %d nodes is created... \n", map_size);
        if (file_name != "") write_file(file_name);
        printf("[Grid Map] Done! \r\n");
        return grid_map;
    }

    // Reads the reference's own files and this header's.  A file whose voxel list is shorter than its header announces
    // is an error (the reference keeps reading with a failing fscanf and leaves the remaining voxels free).
    void readGridMap(std::string file_name)
    {
        last_status = WA_OK;
        FILE *fp = fopen(file_name.c_str(), "r");
        if (fp == NULL) {
            last_status = WA_ERR_FILE;
            std::cout << "[Grid Map] Failed to read file, skipping..." << std::endl;
            return;
        }
        release();
        int rx, ry, rz, ms;
        if (fscanf(fp, "%d %d %d %d %f %d", &ms, &rx, &ry, &rz, &precision, &wall) != 6 ||
            fscanf(fp, "%f %f %f %f %f %f", &min_x, &min_y, &min_z, &max_x, &max_y, &max_z) != 6 || rx < 1 || ry < 1 || rz < 1 ||
            !(precision > 0) || wall < 0) {
            fclose(fp);
            last_status = WA_ERR_FORMAT;
            std::cout << "[Grid Map] Malformed grid file, skipping..." << std::endl;
            return;
        }
        std::vector<float> cx(rx), cy(ry), cz(rz);
        wa_axis_coords(min_x, max_x, precision, wall, rx, cx.data());  // model_grid_map.hpp:321-328
        wa_axis_coords(min_y, max_y, precision, wall, ry, cy.data());
        wa_axis_coords(min_z, max_z, precision, wall, rz, cz.data());
        std::vector<uint8_t> fr((size_t)rx * ry * rz, 1);
        size_t got = 0;
        for (; got < fr.size(); got++) {
            int v = 1;
            if (fscanf(fp, "%d", &v) != 1) break;
            fr[got] = v ? 1 : 0;
        }
        fclose(fp);
        if (got != fr.size()) {
            last_status = WA_ERR_FORMAT;
            std::cout << "[Grid Map] Truncated grid file: " << got << " of " << fr.size() << " voxels, skipping..." << std::endl;
            return;
        }
        wa_ctx *ctx = weldacs_dropin::context();
        if (!ctx) { last_status = WA_ERR_DEVICE; return; }
        last_status = wa_grid_from_occupancy(ctx, fr.data(), rx, ry, rz, cx.data(), cy.data(), cz.data(), precision, wall, &grid_h);
        if (last_status != WA_OK) { printf("[Grid Map] %s\n", wa_last_error(ctx)); return; }
        materialise();
        printf("\n[Grid Map] Successfully read grid map from %s \r\n", file_name.c_str());
    }

    Vertex3<T> ***ptr_grid_map() const { return grid_map; }
    int size_of_map() const { return map_size; }

    void plot_grid_map(int figureNumber)
    {
#ifdef WELDACS_WITH_MATPLOTLIB
        std::map<std::string, std::string> keywords;
        keywords.insert(std::pair<std::string, std::string>("marker", "o"));
        plt::scatter(x_list, y_list, z_list, 1, keywords, figureNumber);
#else
        (void)figureNumber;
#endif
    }
    void show_plot()
    {
#ifdef WELDACS_WITH_MATPLOTLIB
        plt::show();
        plt::cla();
#endif
    }

    // extension: the opaque device grid behind this map
    wa_grid *device_grid() const { return grid_h; }
    // extension: status of the last creatGridMap / readGridMap (wa_status)
    int gridStatus() const { return last_status; }
    // extension: grid-file format.  Default: the TRUE bounding box with 9 significant digits, so that readGridMap
    // rebuilds exactly the grid that was written.  Compat: byte for byte the reference's file (model_grid_map.hpp:275-294)
    // -- "%f" and the last triangle's box instead of the mesh's, which does not round-trip.
    void setGridFileCompat(bool on) { file_compat = on; }

    T precision;
    int wall;
    int rangeX, rangeY, rangeZ;

    GridMap() : precision(0), wall(0), rangeX(0), rangeY(0), rangeZ(0) {}
    ~GridMap() { release(); }

protected:
    std::vector<T> x_list, y_list, z_list;  // coordinates of occupied voxels (plot only)

private:
    Vertex3<float> ***grid_map = NULL;
    std::vector<Vertex3<float>> flat;
    std::vector<Vertex3<float> *> rows;
    std::vector<Vertex3<float> **> planes;
    wa_grid *grid_h = NULL;
    T min_x = 0, min_y = 0, min_z = 0, max_x = 0, max_y = 0, max_z = 0;
    T q5_lo[3] = {0, 0, 0}, q5_hi[3] = {0, 0, 0};
    int map_size = 0;
    int last_status = WA_OK;
    bool file_compat = false;

    void release()
    {
        if (grid_h) wa_grid_destroy(grid_h);
        grid_h = NULL;
        grid_map = NULL;
        flat.clear(); rows.clear(); planes.clear();
        x_list.clear(); y_list.clear(); z_list.clear();
        map_size = 0;
    }
    // host mirror Vertex3[z][y][x] that ptr_grid_map() hands out (model_grid_map.hpp:203-216)
    void materialise()
    {
        int32_t dims[3];
        wa_grid_info(grid_h, dims, NULL, NULL, NULL);
        rangeX = dims[0]; rangeY = dims[1]; rangeZ = dims[2];
        map_size = rangeX * rangeY * rangeZ;
        std::vector<float> cx(rangeX), cy(rangeY), cz(rangeZ);
        std::vector<uint8_t> fr((size_t)map_size);
        wa_grid_read_coords(grid_h, cx.data(), cy.data(), cz.data());
        wa_grid_read_occupancy(grid_h, fr.data());
        flat.resize((size_t)map_size);
        rows.resize((size_t)rangeZ * rangeY);
        planes.resize((size_t)rangeZ);
        size_t id = 0;
        for (int z = 0; z < rangeZ; z++) {
            planes[z] = &rows[(size_t)z * rangeY];
            for (int y = 0; y < rangeY; y++) {
                rows[(size_t)z * rangeY + y] = &flat[id];
                for (int x = 0; x < rangeX; x++, id++) {
                    Vertex3<float> &v = flat[id];
                    v.pt.x = cx[x]; v.pt.y = cy[y]; v.pt.z = cz[z];
                    v.isFree = fr[id] != 0;
                    v.id = id;
                    if (!v.isFree) { x_list.push_back(v.pt.x); y_list.push_back(v.pt.y); z_list.push_back(v.pt.z); }
                }
            }
        }
        grid_map = planes.data();
    }
    void write_file(const std::string &file_name)
    {
        FILE *fp = fopen(file_name.c_str(), "w");
        if (!fp) { last_status = WA_ERR_FILE; return; }
        if (file_compat) {
            fprintf(fp, "%d %d %d %d %f %d\n", map_size, rangeX, rangeY, rangeZ, precision, wall);
            fprintf(fp, "%f %f %f %f %f %f\n", q5_lo[0], q5_lo[1], q5_lo[2], q5_hi[0], q5_hi[1], q5_hi[2]);
        } else {
            fprintf(fp, "%d %d %d %d %.9g %d\n", map_size, rangeX, rangeY, rangeZ, (double)precision, wall);
            fprintf(fp, "%.9g %.9g %.9g %.9g %.9g %.9g\n", (double)min_x, (double)min_y, (double)min_z, (double)max_x, (double)max_y, (double)max_z);
        }
        for (int i = 0; i < rangeZ; i++)
            for (int j = 0; j < rangeY; j++) {
                for (int k = 0; k < rangeX; k++) fprintf(fp, "%d ", (int)grid_map[i][j][k].isFree);
                fprintf(fp, "\n");
            }
        fclose(fp);
        printf("\n[Grid Map] Successfully write to %s \r\n", file_name.c_str());
    }
};

#endif
