// stl_text.hpp -- the ASCII branch of STLReader (read_STL.hpp:99-129) as the reference's run-time executes it.  Plain host C++ (no HIP):
// included by host_grid.inc (wa_stl_parse) and, on its own, by tests/cpp/stl_text_check.cpp, which runs it under ASan + UBSan.
#pragma once
#include <float.h>
#include <locale.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/weldacs.h"

// The ASCII branch (read_STL.hpp:99-129).  The reference reads the text through a std::stringstream:
//     ss >> name >> name; ss.get();
//     do { ss >> word; if (word != "facet") break;  getline, getline;  3 x (ss >> word >> x >> y >> z);  push_back(tri);  getline x 3; } while (1);
// `tri` lives outside the loop and its normal is never assigned: every triangle arrives with the normal (0, 0, 0) -- creatGridMap's plane
// distance is then 0 and every voxel of the triangle's bounding box +- precision is occupied -- and a vertex the stream could not
// read keeps the previous triangle's value.  Same results here, so the stream is modelled with the rules the reference's run-time applies
// (libstdc++): formatted reads skip white space and fail at the end of the text; a read on a stream that is not good() sets the fail flag and
// leaves its target alone; reaching the end inside a word or a number raises the end flag only; numbers pass num_get's filter (sign, leading
// zeros folded into one, digits, one point, one exponent with its sign) and then strtof in the "C" locale -- text strtof does not take
// whole is a failure with the value 0, an overflow a failure with +-FLT_MAX.
// One text has no result in the reference: when it ends directly behind a "facet" word, every later read fails, `word` stays "facet" and
// the loop pushes triangles until memory runs out.  That text is refused (WA_ERR_FORMAT).
namespace {
struct WaStlText {
    const uint8_t *s;
    size_t n, at = 0;
    bool at_end = false, failed = false;
    WaStlText(const uint8_t *b, size_t len) : s(b), n(0) { while (n < len && b[n] != 0) n++; }   // a C string: up to the first NUL
    static bool blank(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
    bool ready(bool formatted)
    {
        bool ok = !at_end && !failed;
        if (ok && formatted) {
            while (at < n && blank(s[at])) at++;
            if (at == n) { at_end = true; ok = false; }
        }
        if (!ok) failed = true;
        return ok;
    }
    // operator>>(std::string&): false = the target keeps its old value
    bool word(std::string &w)
    {
        if (!ready(true)) return false;
        size_t a = at;
        while (at < n && !blank(s[at])) at++;
        if (at == n) at_end = true;
        w.assign((const char *)s + a, at - a);
        return true;
    }
    bool line(std::string &w)
    {
        if (!ready(false)) return false;
        size_t a = at;
        while (at < n && s[at] != '\n') at++;
        w.assign((const char *)s + a, at - a);
        if (at == n) { at_end = true; if (at == a) failed = true; }   // nothing extracted at all
        else at++;                                                    // the delimiter is extracted, not stored
        return true;
    }
    void skip_char()
    {
        if (!ready(false)) return;
        if (at < n) at++;
        else { at_end = true; failed = true; }
    }
    void number(float &v, locale_t c_locale)
    {
        if (!ready(true)) return;
        std::string x;
        bool mantissa = false, point = false, exponent = false;
        if (at < n && (s[at] == '+' || s[at] == '-')) x += (char)s[at++];
        while (at < n && s[at] == '0') { if (!mantissa) x += '0'; mantissa = true; at++; }
        while (at < n) {
            const int c = s[at];
            if (c >= '0' && c <= '9') { x += (char)c; mantissa = true; at++; }
            else if (c == '.' && !point && !exponent) { x += '.'; point = true; at++; }
            else if ((c == 'e' || c == 'E') && !exponent && mantissa) {
                x += 'e';
                exponent = true;
                at++;
                if (at < n && (s[at] == '+' || s[at] == '-')) x += (char)s[at++];
            } else break;
        }
        if (at == n) at_end = true;
        char *rest = nullptr;
        float r = strtof_l(x.c_str(), &rest, c_locale);
        if (rest == x.c_str() || *rest != 0) { r = 0.f; failed = true; }
        else if (r == INFINITY) { r = FLT_MAX; failed = true; }
        else if (r == -INFINITY) { r = -FLT_MAX; failed = true; }
        v = r;
    }
};
}  // namespace

static int64_t stl_parse_text(const uint8_t *b, size_t len, float *tris, int64_t cap_tris)
{
    locale_t c_locale = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    if (!c_locale) return -WA_ERR_ALLOC;
    WaStlText in(b, len);
    std::string name, word;
    in.word(name);
    in.word(name);
    in.skip_char();
    float tri[12] = {0.f};      // normal (never read: stays 0), v0, v1, v2
    int64_t count = 0, rc = 0;
    for (;;) {
        const bool fresh = in.word(word);
        if (word != "facet") break;
        if (!fresh) { rc = -WA_ERR_FORMAT; break; }       // see above: the reference never comes back from this text
        in.line(word);                                    // the rest of "facet normal nx ny nz"
        in.line(word);                                    // "outer loop"
        for (int v = 0; v < 3; v++) {
            in.word(word);                                // "vertex"
            for (int c = 0; c < 3; c++) in.number(tri[3 + 3 * v + c], c_locale);
        }
        if (tris) {
            if (count >= cap_tris) { rc = -WA_ERR_CAPACITY; break; }
            memcpy(tris + count * 12, tri, sizeof tri);
        }
        count++;
        in.line(word);                                    // the rest of the last vertex line, "endloop", "endfacet"
        in.line(word);
        in.line(word);
    }
    freelocale(c_locale);
    return rc ? rc : count;
}
