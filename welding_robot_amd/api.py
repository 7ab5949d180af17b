"""Thin numpy-facing wrappers over the C ABI (include/weldacs.h) for tests and bench.py.
All compute happens in libweldacs.so's HIP kernels; this file only marshals pointers."""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib as L
from ._lib import AcsParams, GtspParams, RNG_DEV, RNG_REF, WeldacsError  # noqa: F401


def _ptr(a):
    return a.ctypes.data if a is not None else None


# Contexts still open when the interpreter exits are closed HERE, children first, while the HIP / RCCL runtimes are still up: left to
# __del__ during interpreter teardown the destroy calls can arrive after those runtimes' own exit handlers have run (seen on ROCm 7.2 as
# "terminate called after throwing std::bad_variant_access" from a communicator destroyed that late).
_live_contexts = weakref.WeakSet()


@atexit.register
def _close_live_contexts():
    for c in list(_live_contexts):
        try:
            c.close()
        except Exception:   # noqa: BLE001 -- exit path: nothing useful can be done with an error here
            pass


class Context:
    def __init__(self, device=0, lib_path=None):
        self.lib = L.load(lib_path)
        h = C.c_void_p()
        rc = self.lib.wa_ctx_create(device, C.byref(h))
        if rc:
            raise WeldacsError(rc, "wa_ctx_create(%d) failed: no usable HIP device" % device)
        self.h = h
        self._children = weakref.WeakSet()  # grids / solvers must be destroyed before the context
        _live_contexts.add(self)

    def check(self, rc):
        if rc:
            raise WeldacsError(rc, self.lib.wa_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            for ch in sorted(list(self._children), key=lambda o: 0 if isinstance(o, (AcsSolver, Comm)) else 1):
                ch.close()
            self.lib.wa_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        self.lib.wa_ctx_device_name(self.h, buf, 256)
        return buf.value.decode()

    @property
    def stream(self):
        return self.lib.wa_ctx_stream(self.h)

    def memory_info(self):
        """(free, total) bytes of device memory"""
        f, t = C.c_int64(), C.c_int64()
        self.check(self.lib.wa_ctx_memory_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def cached_bytes(self):
        """device bytes of destroyed solvers this context keeps for its next solver (wa_ctx_cached_bytes); counted as free by memory_info"""
        b = C.c_int64()
        self.check(self.lib.wa_ctx_cached_bytes(self.h, C.byref(b)))
        return b.value

    def trim(self):
        """give the kept blocks back to the driver (wa_ctx_trim)"""
        self.check(self.lib.wa_ctx_trim(self.h))

    def cache_stats(self):
        """wa_ctx_cache_stats as a dict: what the context's kept memory served and what had to come from the driver"""
        v = (C.c_int64 * 8)()
        self.check(self.lib.wa_ctx_cache_stats(self.h, v))
        keys = ("blocks", "hit_bytes", "miss_bytes", "released_bytes", "blocks_all_from_kept", "oom_events", "arena_build_ms", "arena")
        d = {k: int(x) for k, x in zip(keys, v)}
        d["kept_bytes"] = self.cached_bytes()
        return d

    def sync(self):
        self.check(self.lib.wa_ctx_sync(self.h))


def stl_parse(data):
    lib = L.load()
    buf = np.frombuffer(data, np.uint8)
    n = lib.wa_stl_parse(_ptr(buf), len(buf), None, 0)
    if n < 0:
        raise WeldacsError(-n, "wa_stl_parse")
    tris = np.empty((n, 12), np.float32)
    lib.wa_stl_parse(_ptr(buf), len(buf), _ptr(tris), n)
    return tris


def stl_read_file(path):
    lib = L.load()
    n = lib.wa_stl_read_file(path.encode(), None, 0)
    if n < 0:
        raise WeldacsError(-n, "wa_stl_read_file(%s)" % path)
    tris = np.empty((n, 12), np.float32)
    lib.wa_stl_read_file(path.encode(), _ptr(tris), n)
    return tris


def axis_coords(lo, hi, precision, wall, n):
    out = np.empty(n, np.float32)
    L.load().wa_axis_coords(C.c_float(lo), C.c_float(hi), C.c_float(precision), wall, n, _ptr(out))
    return out


class Grid:
    def __init__(self, ctx, handle, bbox=None):
        self.ctx, self.h, self.bbox = ctx, handle, bbox
        ctx._children.add(self)
        dims = np.zeros(3, np.int32)
        p, w, nf = C.c_float(), C.c_int32(), C.c_int64()
        ctx.check(ctx.lib.wa_grid_info(self.h, _ptr(dims), C.byref(p), C.byref(w), C.byref(nf)))
        self.nx, self.ny, self.nz = (int(v) for v in dims)
        self.precision, self.wall, self.n_free = np.float32(p.value), w.value, nf.value

    @classmethod
    def from_mesh(cls, ctx, tris, precision, wall):
        tris = np.ascontiguousarray(tris, np.float32)
        h = C.c_void_p()
        bbox = np.zeros(6, np.float32)
        ctx.check(ctx.lib.wa_grid_from_mesh(ctx.h, _ptr(tris), len(tris), C.c_float(precision), wall, C.byref(h), _ptr(bbox)))
        return cls(ctx, h, bbox)

    @classmethod
    def from_occupancy(cls, ctx, free, cx, cy, cz, precision, wall=0):
        free = np.ascontiguousarray(free, np.uint8).reshape(-1)
        cx, cy, cz = (np.ascontiguousarray(a, np.float32) for a in (cx, cy, cz))
        assert free.size == len(cx) * len(cy) * len(cz)
        h = C.c_void_p()
        ctx.check(ctx.lib.wa_grid_from_occupancy(ctx.h, _ptr(free), len(cx), len(cy), len(cz), _ptr(cx), _ptr(cy),
                                                 _ptr(cz), C.c_float(precision), wall, C.byref(h)))
        return cls(ctx, h)

    @property
    def n(self):
        return self.nx * self.ny * self.nz

    def occupancy(self):
        out = np.empty(self.n, np.uint8)
        self.ctx.check(self.ctx.lib.wa_grid_read_occupancy(self.h, _ptr(out)))
        return out

    def coords(self):
        cx, cy, cz = np.empty(self.nx, np.float32), np.empty(self.ny, np.float32), np.empty(self.nz, np.float32)
        self.ctx.check(self.ctx.lib.wa_grid_read_coords(self.h, _ptr(cx), _ptr(cy), _ptr(cz)))
        return cx, cy, cz

    def resolve(self, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        ids = np.empty(len(pts), np.int64)
        self.ctx.check(self.ctx.lib.wa_grid_resolve_points(self.h, _ptr(pts), len(pts), _ptr(ids)))
        return ids

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wa_grid_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def default_params(**kw):
    p = AcsParams()
    L.load().wa_acs_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class AcsSolver:
    def __init__(self, ctx, grid, n_slots=1, max_colony=256, path_capacity=0, neighbourhood=6, lazy=False):
        self.ctx, self.grid, self.n_slots, self.max_colony, self.nb = ctx, grid, n_slots, max_colony, neighbourhood
        h = C.c_void_p()
        if lazy:
            ctx.check(ctx.lib.wa_acs_create_lazy_nb(ctx.h, grid.h, n_slots, max_colony, path_capacity, neighbourhood, C.byref(h)))
        else:
            ctx.check(ctx.lib.wa_acs_create_nb(ctx.h, grid.h, n_slots, max_colony, path_capacity, neighbourhood, C.byref(h)))
        self.h = h
        self.iters = 0
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wa_acs_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def init_pheromone(self, p0=1.0, slot=-1):
        self.ctx.check(self.ctx.lib.wa_acs_init_pheromone(self.h, slot, C.c_float(p0)))

    def reset_pheromone(self, p0=1.0, slot=-1):
        self.ctx.check(self.ctx.lib.wa_acs_reset_pheromone(self.h, slot, C.c_float(p0)))

    def srand(self, seed):
        self.ctx.check(self.ctx.lib.wa_acs_srand(self.h, seed & 0xFFFFFFFF))

    def rand_state(self, state=None):
        st = np.zeros(36, np.int32) if state is None else np.ascontiguousarray(state, np.int32)
        self.ctx.check(self.ctx.lib.wa_acs_rand_state(self.h, _ptr(st), 0 if state is None else 1))
        return st

    def _arrs(self, starts, ends, streams):
        starts = np.ascontiguousarray(np.atleast_1d(starts), np.int64)
        ends = np.ascontiguousarray(np.atleast_1d(ends), np.int64)
        streams = None if streams is None else np.ascontiguousarray(np.atleast_1d(streams), np.uint32)
        return starts, ends, streams

    def begin(self, params, starts, ends, streams=None):
        starts, ends, streams = self._arrs(starts, ends, streams)
        self.n_active = len(starts)
        self.iters = params.max_iteration
        self.ctx.check(self.ctx.lib.wa_acs_begin(self.h, C.byref(params), len(starts), _ptr(starts), _ptr(ends), _ptr(streams)))

    def run(self, n_generations):
        self.ctx.check(self.ctx.lib.wa_acs_run(self.h, n_generations))

    def sync(self):
        self.ctx.check(self.ctx.lib.wa_acs_sync(self.h))

    def straggler_counters(self, slot=0, reset=False):
        """(ants handed over, stragglers finished by a resume block) of one slot"""
        a, b = C.c_uint64(), C.c_uint64()
        self.ctx.check(self.ctx.lib.wa_acs_straggler_counters(self.h, slot, C.byref(a), C.byref(b), 1 if reset else 0))
        return a.value, b.value

    def set_stragglers(self, generations):
        """generations of a search during which ants may be handed over (0: off, < 0: default)"""
        self.ctx.check(self.ctx.lib.wa_acs_set_stragglers(self.h, generations))

    def set_pipeline(self, groups):
        """groups of slots that advance on streams of their own inside run() (0: by rule, 1: one stream)"""
        self.ctx.check(self.ctx.lib.wa_acs_set_pipeline(self.h, groups))

    def pipeline_groups(self):
        g = C.c_int32()
        self.ctx.check(self.ctx.lib.wa_acs_pipeline_info(self.h, C.byref(g)))
        return g.value

    def solve(self, params, starts, ends, streams=None):
        starts, ends, streams = self._arrs(starts, ends, streams)
        self.n_active = len(starts)
        self.iters = params.max_iteration
        self.ctx.check(self.ctx.lib.wa_acs_solve(self.h, C.byref(params), len(starts), _ptr(starts), _ptr(ends), _ptr(streams)))

    def result(self, slot=0):
        cost, n = C.c_float(), C.c_int64()
        self.ctx.check(self.ctx.lib.wa_acs_result(self.h, slot, C.byref(cost), C.byref(n), None, None, 0))
        ids = np.empty(n.value, np.int32)
        ch = np.empty(max(n.value - 1, 0), np.int8)
        if n.value:
            self.ctx.check(self.ctx.lib.wa_acs_result(self.h, slot, C.byref(cost), C.byref(n), _ptr(ids), _ptr(ch), n.value))
        return np.float32(cost.value), ids, ch

    def results(self, n_slots=None):
        """(costs, [path ids per slot]) of slots 0..n_slots-1 in one round trip"""
        n = self.n_slots if n_slots is None else n_slots
        costs, lens = np.empty(n, np.float32), np.empty(n, np.int64)
        self.ctx.check(self.ctx.lib.wa_acs_result_batch(self.h, n, _ptr(costs), _ptr(lens), None, 0))
        stride = int(lens.max()) if n else 0
        buf = np.empty((n, max(stride, 1)), np.int32)
        if stride:
            self.ctx.check(self.ctx.lib.wa_acs_result_batch(self.h, n, _ptr(costs), _ptr(lens), _ptr(buf), stride))
        return costs, [buf[q, :lens[q]].copy() for q in range(n)]

    def trace(self, slot=0):
        g = C.c_int32()
        self.ctx.check(self.ctx.lib.wa_acs_trace(self.h, slot, C.byref(g), None, None, None, None, None))
        n = g.value
        t = dict(bestL=np.zeros(n, np.float32), iterbestL=np.zeros(n, np.float32), colony=np.zeros(n, np.int32),
                 finite=np.zeros(n, np.int32), steps=np.zeros(n, np.int64))
        self.ctx.check(self.ctx.lib.wa_acs_trace(self.h, slot, C.byref(g), _ptr(t["bestL"]), _ptr(t["iterbestL"]),
                                                 _ptr(t["colony"]), _ptr(t["finite"]), _ptr(t["steps"])))
        return t

    def export_trace(self, dst_device_ptr, gen0, count):
        self.ctx.check(self.ctx.lib.wa_acs_export_trace(self.h, dst_device_ptr, gen0, count))

    def pheromone(self, slot=0):
        out = np.empty(self.grid.n * self.nb, np.float32)
        self.ctx.check(self.ctx.lib.wa_acs_read_pheromone(self.h, slot, _ptr(out)))
        return out

    def ants(self, slot=0):
        """(L, node count) per ant of the generation walked last"""
        c = C.c_int32()
        self.ctx.check(self.ctx.lib.wa_acs_read_ants(self.h, slot, C.byref(c), None, None, 0))
        Ls, lens = np.empty(c.value, np.float32), np.empty(c.value, np.int32)
        self.ctx.check(self.ctx.lib.wa_acs_read_ants(self.h, slot, C.byref(c), _ptr(Ls), _ptr(lens), c.value))
        return Ls, lens

    def ant_path(self, ant, slot=0):
        """node ids visited by one ant of the generation walked last (Agent::getPath())"""
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.wa_acs_read_ant_path(self.h, slot, ant, None, 0, C.byref(n)))
        ids = np.empty(n.value, np.int32)
        self.ctx.check(self.ctx.lib.wa_acs_read_ant_path(self.h, slot, ant, _ptr(ids), n.value, C.byref(n)))
        return ids

    def last_params(self, slot=0):
        c, l, q = C.c_int32(), C.c_float(), C.c_float()
        self.ctx.check(self.ctx.lib.wa_acs_last_params(self.h, slot, C.byref(c), C.byref(l), C.byref(q)))
        return c.value, np.float32(l.value), np.float32(q.value)

    def walk_info(self):
        """what the last DEV walk launch ran with (wa_acs_walk_info): table size, 16-bit entries or not, LDS per walk block, resident blocks per CU"""
        v = (C.c_int32 * 4)()
        self.ctx.check(self.ctx.lib.wa_acs_walk_info(self.h, v))
        lds = int(v[2])
        return dict(hash_log2=int(v[0]), entries16=bool(v[1]), lds_bytes_per_block=lds, resident_blocks_per_cu=min(163840 // lds, 16) if lds else 0, touch_loads=bool(v[3]))

    def profile(self, enable=True, sample_every=1, sweep_every_generation=False, paired=False):
        """paired: every timed sweep-carrying launch is preceded by a stamped no-op dispatch (wa_acs_profile bit 2; why: profiles/r06/sweep_gap.txt)"""
        self.ctx.check(self.ctx.lib.wa_acs_profile(self.h, ((3 if sweep_every_generation else 1) | (4 if paired else 0)) if enable else 0, sample_every))

    def profile_read(self):
        ms = np.zeros(L.K_COUNT, np.float64)
        n = np.zeros(L.K_COUNT, np.int64)
        self.ctx.check(self.ctx.lib.wa_acs_profile_read(self.h, _ptr(ms), _ptr(n)))
        names = ["walk", "rank", "evaporate", "deposit"]
        return {k: dict(ms=float(ms[i]), launches=int(n[i])) for i, k in enumerate(names)}

    def evaporate(self, slot=0, rho=0.8, repeats=1):
        self.ctx.check(self.ctx.lib.wa_acs_evaporate(self.h, slot, C.c_float(rho), repeats))


def memory_estimate(grid, max_colony, path_capacity=0, neighbourhood=6, lazy=False):
    """(bytes per slot, bytes per heuristic field, fixed bytes) of a solver of this shape (wa_acs_memory_estimate)"""
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    grid.ctx.check(grid.ctx.lib.wa_acs_memory_estimate(grid.h, max_colony, path_capacity, neighbourhood, 1 if lazy else 0,
                                                       C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def straggler_pool_bytes(grid, n_slots, max_colony, path_capacity=0, neighbourhood=6, lazy=False):
    """bytes of arrival lists + straggler pools a solver of this shape holds (0 when it gets none)"""
    b = C.c_int64()
    grid.ctx.check(grid.ctx.lib.wa_acs_straggler_pool_bytes(grid.h, n_slots, max_colony, path_capacity, neighbourhood, 1 if lazy else 0, C.byref(b)))
    return b.value


def pack_best_key(cost, rank, slot, lib_path=None):
    """the 64-bit key of the global-best exchange (host-side helper of the library: cost bits << 32 | rank << 16 | slot)"""
    k = C.c_uint64()
    rc = L.load(lib_path).wa_comm_pack_best_key(C.c_float(cost), rank, slot, C.byref(k))
    if rc:
        raise WeldacsError(rc, "wa_comm_pack_best_key")
    return k.value


def unpack_best_key(key, lib_path=None):
    c, r, s_ = C.c_float(), C.c_int32(), C.c_int32()
    L.load(lib_path).wa_comm_unpack_best_key(C.c_uint64(int(key)), C.byref(c), C.byref(r), C.byref(s_))
    return np.float32(c.value), r.value, s_.value


def pair_slots_by_rule(ctx, grid, colony, n_pairs, n_ends, max_iteration, lazy=True, neighbourhood=6, all_fields=False):
    """Concurrent pair searches for `n_pairs` searches on this device -- the rule of the drop-in ACS_Rank::slots_for
    (welding_robot_amd/include/core/ACSRank_3D.hpp): 3/4 of the free memory but at most ~200 GB of fields, at most three
    rounds of resident walk blocks, then whole batches of equal size.  Returns (slots, batches)."""
    per_slot, per_field, fixed = memory_estimate(grid, colony, 0, neighbourhood, lazy)
    per_slot += 20 * max_iteration
    free, _ = ctx.memory_info()
    fields = max(4, n_ends) if all_fields else min(max(4, n_ends), 8)   # (all_fields: every end point's heuristic field stays resident)
    cap = (min(free // 4 * 3, int(200e9)) - fixed - fields * per_field) // per_slot
    cap = max(1, min(cap, max(1, 3 * 2048 // colony)))
    batches = -(-n_pairs // cap)
    return -(-n_pairs // batches), batches


class Comm:
    """wa_comm: RCCL communicator behind the C ABI (csrc/host_comm.inc) -- the global-best exchange of the multi-GPU path.
    Rank 0 makes the id (Comm.unique_id()), the caller ships the 128 bytes to the other ranks."""

    @staticmethod
    def unique_id(lib_path=None):
        buf = np.zeros(128, np.uint8)
        rc = L.load(lib_path).wa_comm_unique_id(_ptr(buf))
        if rc:
            raise WeldacsError(rc, "wa_comm_unique_id failed")
        return buf

    def __init__(self, ctx, rank, world, uid):
        self.ctx = ctx
        uid = np.ascontiguousarray(uid, np.uint8)
        assert uid.size == 128
        h = C.c_void_p()
        ctx.check(ctx.lib.wa_comm_create(ctx.h, rank, world, _ptr(uid), C.byref(h)))
        self.h, self.rank, self.world = h, rank, world
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wa_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def abort(self):
        """give the communicator up now (wa_comm_abort -> ncclCommAbort): every later call fails with WA_ERR_STATE"""
        self.ctx.check(self.ctx.lib.wa_comm_abort(self.h))

    def stats(self):
        """what RCCL itself says about this communicator, and what was issued on it (wa_comm_stats)"""
        v = (C.c_int64 * 5)()
        self.ctx.check(self.ctx.lib.wa_comm_stats(self.h, v))
        return dict(ranks=int(v[0]), version=int(v[1]), allreduce_calls=int(v[2]), other_calls=int(v[3]), aborted=bool(v[4]))

    def allreduce_best(self, solver, gen0, count):
        """asynchronous: MIN over ranks (and active slots) of best_L[gen0 .. gen0+count)"""
        self.ctx.check(self.ctx.lib.wa_acs_allreduce_best(solver.h, self.h, gen0, count))

    def read_best(self, gen0, count):
        out = np.empty(count, np.float32)
        self.ctx.check(self.ctx.lib.wa_comm_read_best(self.h, gen0, count, _ptr(out)))
        return out

    def read_best_owner(self, gen0, count):
        """(global best cost, owner rank, owner slot) per generation: whose search holds the path that achieved it"""
        cost, rk, sl = np.empty(count, np.float32), np.empty(count, np.int32), np.empty(count, np.int32)
        self.ctx.check(self.ctx.lib.wa_comm_read_best_owner(self.h, gen0, count, _ptr(cost), _ptr(rk), _ptr(sl)))
        return cost, rk, sl

    def allgather_costs(self, index, cost, n_total, fill=np.nan):
        """every rank's (pair index, cost) records to every rank: a vector of n_total costs (entries nobody owns = fill)"""
        index = np.ascontiguousarray(index, np.int32)
        cost = np.ascontiguousarray(cost, np.float32)
        assert index.shape == cost.shape
        out = np.full(n_total, fill, np.float32)
        self.ctx.check(self.ctx.lib.wa_comm_allgather_costs(self.h, index.size, _ptr(index), _ptr(cost), n_total, _ptr(out)))
        return out

    def gather_paths(self, paths, root=0):
        """paths: {global index: node ids} of this rank.  On `root`: {index: ids} of ALL ranks; elsewhere {}."""
        keys = list(paths)
        index = np.ascontiguousarray(keys, np.int32)
        lens = np.ascontiguousarray([len(paths[k]) for k in keys], np.int64)
        ids = np.ascontiguousarray(np.concatenate([np.asarray(paths[k], np.int32) for k in keys]) if keys else np.zeros(0), np.int32)
        npaths, nids = C.c_int64(), C.c_int64()
        self.ctx.check(self.ctx.lib.wa_comm_gather_paths(self.h, root, index.size, _ptr(index), _ptr(lens), _ptr(ids), C.byref(npaths), C.byref(nids)))
        if self.rank != root:
            return {}
        self.ctx.check(self.ctx.lib.wa_comm_gathered_paths_counts(self.h, C.byref(npaths), C.byref(nids)))   # what the read below copies out
        gi, gl, gd = np.empty(npaths.value, np.int32), np.empty(npaths.value, np.int64), np.empty(nids.value, np.int32)
        self.ctx.check(self.ctx.lib.wa_comm_gathered_paths_read(self.h, _ptr(gi), _ptr(gl), _ptr(gd)))
        off = np.concatenate([[0], np.cumsum(gl)])
        return {int(gi[i]): gd[off[i]:off[i + 1]].copy() for i in range(npaths.value)}

    def broadcast_grid(self, grid, root=0):
        """wa_comm_broadcast_grid: rank `root` passes its Grid, every other rank None; everybody gets a Grid back (the root its own)"""
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.wa_comm_broadcast_grid(self.h, root, grid.h if grid is not None else None, C.byref(h)))
        if self.rank == root:
            return grid
        return Grid(self.ctx, h)

    def allreduce(self, values, op="max"):
        v = np.ascontiguousarray(np.atleast_1d(values), np.float64).copy()
        self.ctx.check(self.ctx.lib.wa_comm_allreduce_f64(self.h, _ptr(v), v.size, {"min": 0, "max": 1, "sum": 2}[op]))
        return v

    def barrier(self):
        self.ctx.check(self.ctx.lib.wa_comm_barrier(self.h))


class Trajectory:
    """Device-resident polyline (n x 3 floats): a stitched path or a sampled spline."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        ctx._children.add(self)

    @classmethod
    def from_points(cls, ctx, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        h = C.c_void_p()
        ctx.check(ctx.lib.wa_traj_from_points(ctx.h, _ptr(xyz), len(xyz), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def stitch(cls, grid, segments, reverse=None):
        """ACS_GTSP::read_all_segments: `segments` = list of node-id arrays in tour order."""
        segs = [np.ascontiguousarray(s, np.int64).reshape(-1) for s in segments]
        ids = np.concatenate(segs) if segs else np.zeros(0, np.int64)
        off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
        rev = np.ascontiguousarray(reverse, np.uint8) if reverse is not None else None
        h = C.c_void_p()
        ctx = grid.ctx
        ctx.check(ctx.lib.wa_traj_stitch(grid.h, _ptr(ids), _ptr(off), len(segs), _ptr(rev), C.byref(h)))
        return cls(ctx, h)

    def __len__(self):
        return int(self.ctx.lib.wa_traj_size(self.h))

    def points(self):
        out = np.empty((len(self), 3), np.float32)
        self.ctx.check(self.ctx.lib.wa_traj_read(self.h, _ptr(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wa_traj_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Bspline:
    """BS_Basic<float, dim, degree, level_ini, level_fin> (core/BSplineBasic.h) on the device."""

    def __init__(self, ctx, dim, degree, level_ini, level_fin, n_middle, uninit_bits=0):
        self.ctx, self.dim, self.degree, self.n_middle = ctx, dim, degree, n_middle
        h = C.c_void_p()
        ctx.check(ctx.lib.wa_bspline_create(ctx.h, dim, degree, level_ini, level_fin, n_middle, C.byref(h)))
        self.h = h
        ctx._children.add(self)
        if uninit_bits:
            ctx.check(ctx.lib.wa_bspline_set_uninit(self.h, uninit_bits))

    def set_param(self, init, fin, middle, fin_time):
        """SetParam; `middle` is an (n_middle, stride >= dim) array or a Trajectory."""
        init = np.ascontiguousarray(init, np.float32)
        fin = np.ascontiguousarray(fin, np.float32)
        if isinstance(middle, Trajectory):
            rc = self.ctx.lib.wa_bspline_set_param_traj(self.h, _ptr(init), _ptr(fin), middle.h, C.c_float(fin_time))
        else:
            middle = np.ascontiguousarray(middle, np.float32)
            middle = middle.reshape(self.n_middle, -1) if middle.size else np.zeros((0, self.dim), np.float32)
            rc = self.ctx.lib.wa_bspline_set_param(self.h, _ptr(init), _ptr(fin), _ptr(middle), middle.shape[1],
                                                   C.c_float(fin_time))
        self.ctx.check(rc)

    def arrays(self):
        nk, nc = C.c_int64(), C.c_int64()
        self.ctx.check(self.ctx.lib.wa_bspline_info(self.h, C.byref(nk), C.byref(nc)))
        knots, cps = np.empty(nk.value, np.float32), np.empty((nc.value, self.dim), np.float32)
        self.ctx.check(self.ctx.lib.wa_bspline_read(self.h, _ptr(knots), _ptr(cps)))
        return knots, cps

    def eval(self, us, der=0):
        us = np.ascontiguousarray(us, np.float32).reshape(-1)
        out = np.empty((len(us), self.dim), np.float32)
        ok = np.empty(len(us), np.uint8)
        self.ctx.check(self.ctx.lib.wa_bspline_eval(self.h, _ptr(us), len(us), der, _ptr(out), _ptr(ok)))
        return out, ok

    def eval_host(self, u, der=0):
        """one time on the host (wa_bspline_eval_host): (dim values, ok)"""
        out = np.zeros(self.dim, np.float32)
        ok = C.c_uint8()
        self.ctx.check(self.ctx.lib.wa_bspline_eval_host(self.h, C.c_float(u), der, _ptr(out), C.byref(ok)))
        return out, bool(ok.value)

    def sample(self, t0, dt, count, der=0, host=True, device=False):
        out = np.empty((count, self.dim), np.float32) if host else None
        ok = np.empty(count, np.uint8) if host else None
        th = C.c_void_p()
        self.ctx.check(self.ctx.lib.wa_bspline_sample(self.h, C.c_float(t0), C.c_float(dt), count, der, _ptr(out),
                                                      _ptr(ok), C.byref(th) if device else None))
        traj = Trajectory(self.ctx, th) if device else None
        return (out, ok, traj) if device else (out, ok)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wa_bspline_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def gtsp_solve(ctx, dist, cnt=None, mode=RNG_DEV, seed=1, stream=0, max_iterations=0, rand_state=None, want_pher=False):
    dist = np.ascontiguousarray(dist, np.float64)
    if dist.ndim == 2:
        dist = dist[None]
    inst, n = dist.shape[0], dist.shape[1]
    cnt = n * (n - 1) // 2 if cnt is None else cnt
    p = GtspParams(mode, seed, stream, max_iterations)
    edges = np.zeros((inst, n, 2), np.int32)
    cost = np.zeros(inst, np.float64)
    iters = np.zeros(inst, np.int32)
    pher = np.zeros((inst, n, n), np.float64) if want_pher else None
    st = None if rand_state is None else np.ascontiguousarray(rand_state, np.int32)
    ctx.check(ctx.lib.wa_gtsp_solve(ctx.h, _ptr(dist), n, cnt, inst, C.byref(p), _ptr(st), _ptr(edges), _ptr(cost),
                                    _ptr(iters), _ptr(pher)))
    return dict(edges=edges, L=cost, iters=iters, pher=pher, rand_state=st)
