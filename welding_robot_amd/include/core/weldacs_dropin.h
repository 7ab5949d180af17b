// weldacs_dropin.h -- state shared by the drop-in headers (model_grid_map.hpp, ACSRank_3D.hpp, ACS_GTSP.hpp,
// BSplineBasic.h): the process-wide wa_ctx and the carried libc rand() stream.
#ifndef WELDACS_DROPIN_H
#define WELDACS_DROPIN_H
#include <stdint.h>

#include <iostream>
#include <vector>

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
// Contexts of the further shards of ACS_Rank::searchBestPathOfPoints (shard 0 runs on the primary context): one per shard index, kept
// for the process like the primary, so that the device blocks a shard's solver gives back serve the same shard of the next call
// (wa_ctx_cached_bytes).  A shard that moves to another device gets a new context.
struct ShardCtx { wa_ctx *ctx; int device; };
inline std::vector<ShardCtx> &shard_slots() { static std::vector<ShardCtx> v; return v; }
inline wa_ctx *shard_context(int shard, int device, int *rc_out)
{
    std::vector<ShardCtx> &v = shard_slots();
    if ((int)v.size() <= shard) v.resize((size_t)shard + 1, ShardCtx{nullptr, -1});
    ShardCtx &s = v[(size_t)shard];
    if (s.ctx && s.device != device) { wa_ctx_destroy(s.ctx); s.ctx = nullptr; }
    int rc = WA_OK;
    if (!s.ctx) { rc = wa_ctx_create(device, &s.ctx); s.device = device; if (rc != WA_OK) s.ctx = nullptr; }
    if (rc_out) *rc_out = rc;
    return s.ctx;
}
// hand the device blocks every context of the drop-in keeps back to the driver (before another library needs the memory)
inline void trim_device_memory()
{
    if (ctx_slot()) wa_ctx_trim(ctx_slot());
    for (ShardCtx &s : shard_slots()) if (s.ctx) wa_ctx_trim(s.ctx);
}
// the reference's process-global rand() stream (Q2), carried between ACS_Rank and ACS_GTSP
inline int32_t *rand_state() { static int32_t st[36]; return st; }
inline bool &rand_state_valid() { static bool v = false; return v; }
}  // namespace weldacs_dropin
#endif
