// read_STL.hpp -- drop-in for the reference's core/read_STL.hpp on the C ABI (wa_stl_read_file).
// Same class and members (STLReader::readFile / NumTri / PointList / TriangleList).  The reference
// exit(1..3)s on I/O errors (read_STL.hpp:34-59); this header prints the same message and returns
// false instead -- a library must not end the process.  ASCII STL is read as the reference reads it
// (:99-129: the normals stay 0, which marks every voxel in each triangle's bbox occupied).
#ifndef _READ_STL_HPP
#define _READ_STL_HPP
#include <stdio.h>

#include <string>
#include <vector>

#include "model_grid_map.hpp"

typedef Point3<float> Point3f;

class STLReader {
public:
    bool readFile(std::string file_name)
    {
        int64_t n = wa_stl_read_file(file_name.c_str(), NULL, 0);
        if (n < 0) {
            fputs(n == -WA_ERR_FORMAT ? "Format error" : (n == -WA_ERR_FILE ? "File error" : "Reading error"), stderr);
            return false;
        }
        std::vector<float> t((size_t)n * 12);
        wa_stl_read_file(file_name.c_str(), t.data(), n);
        Triangles<float> tri;
        tri.trait = 0;
        for (int64_t i = 0; i < n; i++) {  // appends across calls like the reference (no clear)
            const float *p = &t[(size_t)i * 12];
            tri.nor_vec = Point3f(p[0], p[1], p[2]);
            for (int j = 0; j < 3; j++) tri.vertex[j] = Point3f(p[3 + 3 * j], p[4 + 3 * j], p[5 + 3 * j]);
            triangleMesh.push_back(tri);
        }
        unTriangles = (unsigned int)n;
        return true;
    }
    int NumTri() { return unTriangles; }
    std::vector<Point3f> &PointList() { return pointList; }
    const std::vector<Triangles<float>> &TriangleList() { return triangleMesh; }

private:
    std::vector<Point3f> pointList;
    std::vector<Triangles<float>> triangleMesh;
    unsigned int unTriangles = 0;
};

#endif
