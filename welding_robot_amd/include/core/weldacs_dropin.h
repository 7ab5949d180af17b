// weldacs_dropin.h -- state shared by the drop-in headers (model_grid_map.hpp, ACSRank_3D.hpp, ACS_GTSP.hpp,
// BSplineBasic.h): the process-wide wa_ctx and the carried libc rand() stream.
#ifndef WELDACS_DROPIN_H
#define WELDACS_DROPIN_H
#include <stdint.h>

#include <iostream>

#include "weldacs.h"

namespace weldacs_dropin {
// one context per process, like the reference's file-scope globals (main.cpp:33-35)
inline wa_ctx *&ctx_slot() { static wa_ctx *c = nullptr; return c; }
inline int &device_ordinal() { static int d = 0; return d; }
inline wa_ctx *context()
{
    wa_ctx *&c = ctx_slot();
    if (!c) {
        int rc = wa_ctx_create(device_ordinal(), &c);
        if (rc != WA_OK) {
            std::cout << "[weldacs] no usable MI355X/HIP device (wa_ctx_create -> " << rc << "); there is no CPU fallback." << std::endl;
            c = nullptr;
        }
    }
    return c;
}
// the reference's process-global rand() stream (Q2), carried between ACS_Rank and ACS_GTSP
inline int32_t *rand_state() { static int32_t st[36]; return st; }
inline bool &rand_state_valid() { static bool v = false; return v; }
}  // namespace weldacs_dropin
#endif
