// stl_text_check.cpp -- the library's ASCII STL reader (welding_robot_amd/csrc/stl_text.hpp = read_STL.hpp:99-129 with the stream rules
// of the reference's run-time) compiled on its own by the host compiler, so that it can run under AddressSanitizer + UBSan
// (tests/test_sanitizers.py; libweldacs.so itself is built by hipcc and cannot).  For every file named on the command line: the triangle
// count (or -status) of a count-only pass, of a pass into an exactly sized buffer and of a pass into a buffer one triangle short, and an
// FNV-1a hash of the triangles -- which the test holds against what libweldacs.so returns for the same bytes.
//   g++ -std=c++14 -fsanitize=address,undefined -Iinclude tests/cpp/stl_text_check.cpp -o stl_text_check && ./stl_text_check files...
#include <stdio.h>

#include <vector>

#include "../../welding_robot_amd/csrc/stl_text.hpp"

int main(int argc, char **argv)
{
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { printf("%s: cannot open\n", argv[a]); return 2; }
        std::vector<uint8_t> buf;
        uint8_t chunk[4096];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
        fclose(f);
        // (exactly sized heap copy without a terminator: reading one byte past the text is an error ASan reports)
        uint8_t *exact = (uint8_t *)malloc(buf.size() ? buf.size() : 1);
        memcpy(exact, buf.data(), buf.size());
        const int64_t n = stl_parse_text(exact, buf.size(), nullptr, 0);
        unsigned long long h = 1469598103934665603ULL;
        int64_t n2 = n, n3 = n;
        if (n >= 0) {
            std::vector<float> tris((size_t)n * 12 + 1, -7.f);
            n2 = stl_parse_text(exact, buf.size(), tris.data(), n);
            if (tris[(size_t)n * 12] != -7.f) { printf("%s: wrote past the buffer\n", argv[a]); return 3; }
            for (size_t i = 0; i < (size_t)n * 12 * 4; i++) h = (h ^ ((const uint8_t *)tris.data())[i]) * 1099511628211ULL;
            if (n > 0) {
                std::vector<float> small((size_t)(n - 1) * 12 + 1, -7.f);
                n3 = stl_parse_text(exact, buf.size(), small.data(), n - 1);
                if (small[(size_t)(n - 1) * 12] != -7.f) { printf("%s: wrote past the short buffer\n", argv[a]); return 3; }
            }
        }
        free(exact);
        printf("%lld %lld %lld %016llx\n", (long long)n, (long long)n2, (long long)n3, h);
    }
    return 0;
}
