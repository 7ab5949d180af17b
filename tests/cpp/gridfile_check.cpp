// gridfile_check.cpp -- test helper: GridMap<float>'s file hand-offs through the drop-in header
// (model_grid_map.hpp:275-294 writer, :300-356 reader).  Built by tests/test_gpu_gridfile.py with the host g++.
//   gridfile_check write <stl> <precision> <wall> <out.in> <compat 0|1> <dump.txt>
//   gridfile_check read  <file.in> <dump.txt>
// dump: "status S" / "dims nx ny nz wall" / "precision HEX" / "cx HEX..." / "cy ..." / "cz ..." / "free 0101..."
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "core/read_STL.hpp"
#include "core/model_grid_map.hpp"

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

static int dump(GridMap<float> &g, const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return 3;
    fprintf(fp, "status %d\n", g.gridStatus());
    Vertex3<float> ***m = g.ptr_grid_map();
    if (m) {
        fprintf(fp, "dims %d %d %d %d\n", g.rangeX, g.rangeY, g.rangeZ, g.wall);
        fprintf(fp, "precision %08x\n", bits(g.precision));
        fprintf(fp, "cx");
        for (int x = 0; x < g.rangeX; x++) fprintf(fp, " %08x", bits(m[0][0][x].pt.x));
        fprintf(fp, "\ncy");
        for (int y = 0; y < g.rangeY; y++) fprintf(fp, " %08x", bits(m[0][y][0].pt.y));
        fprintf(fp, "\ncz");
        for (int z = 0; z < g.rangeZ; z++) fprintf(fp, " %08x", bits(m[z][0][0].pt.z));
        fprintf(fp, "\nfree ");
        for (int z = 0; z < g.rangeZ; z++)
            for (int y = 0; y < g.rangeY; y++)
                for (int x = 0; x < g.rangeX; x++) fputc(m[z][y][x].isFree ? '1' : '0', fp);
        fprintf(fp, "\n");
    }
    fclose(fp);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 8 && !strcmp(argv[1], "write")) {
        STLReader model;
        if (!model.readFile(argv[2])) return 2;
        GridMap<float> g;
        g.setGridFileCompat(atoi(argv[6]) != 0);
        g.creatGridMap(model.TriangleList(), (float)atof(argv[3]), atoi(argv[4]), argv[5]);
        return dump(g, argv[7]);
    }
    if (argc >= 4 && !strcmp(argv[1], "read")) {
        GridMap<float> g;
        g.readGridMap(argv[2]);
        return dump(g, argv[3]);
    }
    fprintf(stderr, "usage: gridfile_check write|read ...\n");
    return 1;
}
