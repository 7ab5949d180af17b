"""ctypes binding of oracle/libweld_oracle.so -- the CHECKER.  Test infrastructure only:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from
the welding_robot_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# WELD_ORACLE_LIB: another build of the same source (oracle/Makefile `asan`: tests/test_sanitizers.py)
LIB_PATH = os.environ.get("WELD_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libweld_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "ref_harness")

REF, DEV = 0, 1


class GlibcRand(C.Structure):
    _fields_ = [("r", C.c_int32 * 34), ("f", C.c_int32), ("b", C.c_int32), ("calls", C.c_uint64)]


class AcsParams(C.Structure):
    _fields_ = [("alpha", C.c_int32), ("beta", C.c_float), ("rho", C.c_float),
                ("pheromone_0", C.c_float), ("max_iteration", C.c_int32), ("predict", C.c_float),
                ("fixed_colony", C.c_int32), ("rng_mode", C.c_int32), ("seed", C.c_uint64),
                ("stream", C.c_uint32)]


class BsplineStruct(C.Structure):
    _fields_ = [("dim", C.c_int32), ("degree", C.c_int32), ("ci", C.c_int32), ("cf", C.c_int32),
                ("n_middle", C.c_int64), ("n_knots", C.c_int64), ("n_cps", C.c_int64),
                ("knots", C.POINTER(C.c_float)), ("cps", C.POINTER(C.c_float)), ("uninit", C.c_float)]


class GtspParams(C.Structure):
    _fields_ = [("rng_mode", C.c_int32), ("seed", C.c_uint64), ("stream", C.c_uint32),
                ("max_iterations", C.c_int32)]


def build():
    if os.environ.get("WELD_ORACLE_LIB"):
        return
    src = [os.path.join(ORACLE_DIR, f) for f in ("weld_oracle.c", "weld_oracle.h")]
    if (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libweld_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.wo_rand.restype = C.c_int32
        L.wo_ctr_rand31.restype = C.c_uint32
        L.wo_ctr_rand31.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.wo_stl_count.restype = C.c_int64
        L.wo_stl_count.argtypes = [C.c_void_p, C.c_size_t]
        L.wo_stl_parse.restype = C.c_int64
        L.wo_stl_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.wo_grid_dims.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
        L.wo_axis_coords.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p]
        L.wo_voxelize.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_resolve_point.restype = C.c_int64
        L.wo_resolve_point.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_float, C.c_void_p]
        L.wo_synth_grid.restype = C.c_int64
        L.wo_synth_grid.argtypes = [C.c_int32, C.c_uint64, C.c_double, C.c_void_p]
        L.wo_acs_create.restype = C.c_void_p
        L.wo_acs_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_float, C.c_float]
        L.wo_acs_create_nb.restype = C.c_void_p
        L.wo_acs_create_nb.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_float, C.c_float, C.c_int32]
        L.wo_acs_destroy.argtypes = [C.c_void_p]
        L.wo_acs_reset.argtypes = [C.c_void_p, C.c_float]
        L.wo_acs_solve.restype = C.c_int32
        L.wo_acs_solve.argtypes = [C.c_void_p, C.POINTER(AcsParams), C.c_int64, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_acs_best_L.restype = C.c_float
        L.wo_acs_best_L.argtypes = [C.c_void_p]
        L.wo_acs_best_len.restype = C.c_int64
        L.wo_acs_best_len.argtypes = [C.c_void_p]
        L.wo_acs_best_path.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_acs_pheromone.restype = C.POINTER(C.c_float)
        L.wo_acs_pheromone.argtypes = [C.c_void_p]
        L.wo_acs_last_params.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_acs_last_paths.restype = None
        L.wo_acs_last_paths.argtypes = [C.c_void_p, C.c_void_p]
        L.wo_acs_last_ants.restype = C.c_int32
        L.wo_acs_last_ants.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_acs_heuristic.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
        L.wo_gtsp_solve.restype = C.c_int32
        L.wo_gtsp_solve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(GtspParams), C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_std_sort_perm.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        BP = C.POINTER(BsplineStruct)
        L.wo_bspline_init.argtypes = [BP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_float]
        L.wo_bspline_free.argtypes = [BP]
        L.wo_bspline_set_param.argtypes = [BP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
        L.wo_bspline_point.argtypes = [BP, C.c_float, C.c_void_p]
        L.wo_bspline_der.argtypes = [BP, C.c_float, C.c_int32, C.c_void_p]
        L.wo_bspline_sample.argtypes = [BP, C.c_float, C.c_float, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.wo_stitch_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_stable_rank_perm.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


# ------------------------------------------------------------------ thin pythonic wrappers
def srand(seed):
    s = GlibcRand()
    lib().wo_srand(C.byref(s), C.c_uint32(seed))
    return s


def rand(s):
    return lib().wo_rand(C.byref(s))


def std_sort_perm(keys):
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.empty(len(keys), np.int32)
    lib().wo_std_sort_perm(keys.ctypes.data, len(keys), perm.ctypes.data)
    return perm


def stable_rank_perm(keys):
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.empty(len(keys), np.int32)
    lib().wo_stable_rank_perm(keys.ctypes.data, len(keys), perm.ctypes.data)
    return perm


def stl_parse(data):
    buf = np.frombuffer(data, np.uint8)
    n = lib().wo_stl_count(buf.ctypes.data, len(buf))
    if n < 0:
        raise ValueError("stl error %d" % n)
    tris = np.empty((n, 12), np.float32)
    lib().wo_stl_parse(buf.ctypes.data, len(buf), tris.ctypes.data)
    return tris


class Grid:
    """nx,ny,nz, cx,cy,cz (float32), free (uint8 raster z,y,x), precision, wall"""

    def __init__(self, cx, cy, cz, free, precision, wall=0):
        self.cx = np.ascontiguousarray(cx, np.float32)
        self.cy = np.ascontiguousarray(cy, np.float32)
        self.cz = np.ascontiguousarray(cz, np.float32)
        self.nx, self.ny, self.nz = len(self.cx), len(self.cy), len(self.cz)
        self.free = np.ascontiguousarray(free, np.uint8).reshape(-1)
        assert self.free.size == self.nx * self.ny * self.nz
        self.precision = np.float32(precision)
        self.wall = wall

    @property
    def n(self):
        return self.nx * self.ny * self.nz

    def resolve(self, pt):
        pt = np.ascontiguousarray(pt, np.float32)
        return lib().wo_resolve_point(self.nx, self.ny, self.nz, self.cx.ctypes.data, self.cy.ctypes.data,
                                      self.cz.ctypes.data, self.free.ctypes.data, C.c_float(self.precision),
                                      pt.ctypes.data)

    def node_pt(self, z, y, x):
        return np.array([self.cx[x], self.cy[y], self.cz[z]], np.float32)


def grid_from_mesh(tris, precision, wall):
    tris = np.ascontiguousarray(tris, np.float32)
    dims = np.zeros(3, np.int32)
    bbox = np.zeros(6, np.float32)
    L = lib()
    L.wo_grid_dims(tris.ctypes.data, len(tris), C.c_float(precision), wall, dims.ctypes.data, bbox.ctypes.data)
    axes = []
    for c in range(3):
        out = np.empty(dims[c], np.float32)
        L.wo_axis_coords(C.c_float(bbox[c]), C.c_float(bbox[3 + c]), C.c_float(precision), wall, int(dims[c]),
                         out.ctypes.data)
        axes.append(out)
    free = np.empty(int(dims[0]) * int(dims[1]) * int(dims[2]), np.uint8)
    L.wo_voxelize(tris.ctypes.data, len(tris), C.c_float(precision), int(dims[0]), int(dims[1]), int(dims[2]),
                  axes[0].ctypes.data, axes[1].ctypes.data, axes[2].ctypes.data, free.ctypes.data)
    return Grid(axes[0], axes[1], axes[2], free, precision, wall)


def synth_grid(n, seed=2024, occ_prob=0.10):
    """SURVEY 8(d) synthetic benchmark grid: wall 0, precision 1, coords = index."""
    free = np.empty(n * n * n, np.uint8)
    lib().wo_synth_grid(n, C.c_uint64(seed), C.c_double(occ_prob), free.ctypes.data)
    ax = np.arange(n, dtype=np.float32)
    return Grid(ax, ax.copy(), ax.copy(), free, 1.0, 0)


def write_grid_in(grid, path, lo=None, hi=None):
    """The reference's grid-map text format (model_grid_map.hpp:275-294) so that a synthetic
    grid can enter the real reference through GridMap::readGridMap (:300-356).  Only exact for
    wall == 0 grids whose coordinates are lo + i*p."""
    lo = lo if lo is not None else (grid.cx[0], grid.cy[0], grid.cz[0])
    hi = hi if hi is not None else (grid.cx[-1], grid.cy[-1], grid.cz[-1])
    with open(path, "w") as f:
        f.write("%d %d %d %d %f %d\n" % (grid.n, grid.nx, grid.ny, grid.nz, grid.precision, grid.wall))
        f.write("%f %f %f %f %f %f\n" % (lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]))
        fr = grid.free.reshape(grid.nz * grid.ny, grid.nx)
        for row in fr:
            f.write(" ".join("1" if v else "0" for v in row) + " \n")


def q5_bbox(tris, precision):
    """What GridMap's min_*/max_* members hold when creatGridMap writes its file: the LAST triangle's
    bounding box -+ precision, in float arithmetic (model_grid_map.hpp:228-248)."""
    v = np.ascontiguousarray(tris, np.float32)[-1, 3:12].reshape(3, 3)
    p = np.float32(precision)
    return (v.min(axis=0) - p).astype(np.float32), (v.max(axis=0) + p).astype(np.float32)


def read_grid_in(path):
    """GridMap::readGridMap restated (model_grid_map.hpp:300-356): header, per-axis coordinates from the header's
    box (:321-328 == wo_axis_coords), then one int per voxel in raster z,y,x order; a voxel whose token is missing
    stays free (the reference ignores fscanf's result).  Returns (Grid, voxel tokens read)."""
    tok = open(path, "rb").read().split()
    ms, rx, ry, rz = (int(t) for t in tok[:4])
    prec, wall = np.float32(tok[4].decode()), int(tok[5])
    box = [np.float32(t.decode()) for t in tok[6:12]]
    ax = []
    for c, n in enumerate((rx, ry, rz)):
        out = np.empty(n, np.float32)
        lib().wo_axis_coords(C.c_float(box[c]), C.c_float(box[3 + c]), C.c_float(prec), wall, n, out.ctypes.data)
        ax.append(out)
    vals = tok[12:12 + rx * ry * rz]
    free = np.ones(rx * ry * rz, np.uint8)
    free[:len(vals)] = [1 if int(t) else 0 for t in vals]
    return Grid(ax[0], ax[1], ax[2], free, prec, wall), len(vals)


class Acs:
    def __init__(self, grid, pheromone_0=1.0, nb=6):
        self.grid, self.nb = grid, nb
        self.h = lib().wo_acs_create_nb(grid.nx, grid.ny, grid.nz, grid.cx.ctypes.data, grid.cy.ctypes.data,
                                        grid.cz.ctypes.data, grid.free.ctypes.data, C.c_float(grid.precision),
                                        C.c_float(pheromone_0), nb)
        assert self.h, "nb must be 6 or 26"

    def __del__(self):
        if getattr(self, "h", None):
            lib().wo_acs_destroy(self.h)
            self.h = None

    def reset(self, pheromone_0=1.0):
        lib().wo_acs_reset(self.h, C.c_float(pheromone_0))

    def solve(self, start_id, end_id, iters, predict, fixed_colony=0, mode=REF, rng=None, seed=0, stream=0,
              alpha=1, beta=0.6, rho=0.8, pheromone_0=1.0):
        p = AcsParams(alpha, beta, rho, pheromone_0, iters, predict, fixed_colony, mode, seed, stream)
        tr = dict(bestL=np.zeros(iters, np.float32), iterbestL=np.zeros(iters, np.float32),
                  colony=np.zeros(iters, np.int32), finite=np.zeros(iters, np.int32),
                  steps=np.zeros(iters, np.int64))
        timing = np.zeros(4, np.float64)
        rc = lib().wo_acs_solve(self.h, C.byref(p), start_id, end_id, C.byref(rng) if rng is not None else None,
                                tr["bestL"].ctypes.data, tr["iterbestL"].ctypes.data, tr["colony"].ctypes.data,
                                tr["finite"].ctypes.data, tr["steps"].ctypes.data, timing.ctypes.data)
        assert rc == 0
        tr["timing"] = timing
        return tr

    @property
    def best_L(self):
        return np.float32(lib().wo_acs_best_L(self.h))

    def best_path(self):
        n = lib().wo_acs_best_len(self.h)
        ids = np.empty(n, np.int32)
        ch = np.empty(max(n - 1, 0), np.int32)
        lib().wo_acs_best_path(self.h, ids.ctypes.data, ch.ctypes.data)
        return ids, ch

    def pheromone(self):
        ptr = lib().wo_acs_pheromone(self.h)
        return np.ctypeslib.as_array(ptr, shape=(self.grid.n * self.nb,)).copy()

    def last_params(self):
        c, l, q = C.c_int32(), C.c_float(), C.c_float()
        lib().wo_acs_last_params(self.h, C.byref(c), C.byref(l), C.byref(q))
        return c.value, np.float32(l.value), np.float32(q.value)

    def last_ants(self):
        n = lib().wo_acs_last_ants(self.h, None, None)
        lens, L = np.empty(n, np.int32), np.empty(n, np.float32)
        lib().wo_acs_last_ants(self.h, lens.ctypes.data, L.ctypes.data)
        return lens, L

    def last_paths(self):
        """node ids of every ant of the last generation, one array per ant"""
        lens, _ = self.last_ants()
        ids = np.empty(int(lens.sum()), np.int32)
        lib().wo_acs_last_paths(self.h, ids.ctypes.data)
        return np.split(ids, np.cumsum(lens)[:-1])

    def heuristic(self, end_id, beta=0.6):
        out = np.empty(self.grid.n * self.nb, np.float32)
        lib().wo_acs_heuristic(self.h, end_id, C.c_float(beta), out.ctypes.data)
        return out


def gtsp_solve(dist, cnt=None, mode=REF, rng=None, seed=0, stream=0, max_iterations=0, want_pher=False):
    dist = np.ascontiguousarray(dist, np.float64)
    n = dist.shape[0]
    cnt = n * (n - 1) // 2 if cnt is None else cnt
    p = GtspParams(mode, seed, stream, max_iterations)
    edges = np.zeros(2 * n, np.int32)
    L = C.c_double()
    pher = np.zeros((n, n), np.float64) if want_pher else None
    it = lib().wo_gtsp_solve(dist.ctypes.data, n, cnt, C.byref(p), C.byref(rng) if rng is not None else None,
                             edges.ctypes.data, C.byref(L), pher.ctypes.data if want_pher else None)
    return dict(iters=it, edges=edges.reshape(n, 2), L=L.value, pher=pher)


class Bspline:
    """BS_Basic<float, dim, degree, ci, cf> restated (BSplineBasic.h); `uninit_bits` is the 32-bit
    pattern of the heap cells the reference reads without writing."""

    def __init__(self, dim, degree, ci, cf, n_middle, uninit_bits=0):
        self.s = BsplineStruct()
        un = np.array([uninit_bits], np.uint32).view(np.float32)[0]
        if lib().wo_bspline_init(C.byref(self.s), dim, degree, ci, cf, n_middle, C.c_float(un)) != 0:
            raise ValueError("invalid BS_Basic arguments")
        self.dim, self.degree = dim, degree

    def __del__(self):
        if getattr(self, "s", None) is not None:
            lib().wo_bspline_free(C.byref(self.s))
            self.s = None

    def set_param(self, init, fin, middle, fin_time):
        init = np.ascontiguousarray(init, np.float32)
        fin = np.ascontiguousarray(fin, np.float32)
        middle = np.ascontiguousarray(middle, np.float32)
        middle = middle.reshape(self.s.n_middle, -1) if middle.size else np.zeros((0, self.dim), np.float32)
        lib().wo_bspline_set_param(C.byref(self.s), init.ctypes.data, fin.ctypes.data, middle.ctypes.data,
                                   middle.shape[1], C.c_float(fin_time))

    @property
    def knots(self):
        return np.ctypeslib.as_array(self.s.knots, (self.s.n_knots,)).copy()

    @property
    def cps(self):
        return np.ctypeslib.as_array(self.s.cps, (self.s.n_cps * self.dim,)).copy().reshape(-1, self.dim)

    def eval(self, us, der=0, prefill=-777.0):
        us = np.ascontiguousarray(us, np.float32)
        out = np.full((len(us), self.dim), prefill, np.float32)
        ok = np.zeros(len(us), np.uint8)
        for i, u in enumerate(us):
            row = out[i]
            ok[i] = (lib().wo_bspline_point(C.byref(self.s), C.c_float(u), row.ctypes.data) if der == 0 else
                     lib().wo_bspline_der(C.byref(self.s), C.c_float(u), der, row.ctypes.data))
        return out, ok

    def sample(self, t0, dt, count, der=0, prefill=-777.0):
        out = np.full((count, self.dim), prefill, np.float32)
        ok = np.zeros(count, np.uint8)
        lib().wo_bspline_sample(C.byref(self.s), C.c_float(t0), C.c_float(dt), count, der, out.ctypes.data,
                                ok.ctypes.data)
        return out, ok


def stitch_segments(seg_ids, seg_off, reverse, nx, ny, cx, cy, cz):
    seg_ids = np.ascontiguousarray(seg_ids, np.int64)
    seg_off = np.ascontiguousarray(seg_off, np.int64)
    rev = np.ascontiguousarray(reverse, np.uint8) if reverse is not None else None
    out = np.zeros((len(seg_ids), 3), np.float32)
    lib().wo_stitch_segments(seg_ids.ctypes.data, seg_off.ctypes.data, len(seg_off) - 1,
                             rev.ctypes.data if rev is not None else None, nx, ny,
                             np.ascontiguousarray(cx, np.float32).ctypes.data,
                             np.ascontiguousarray(cy, np.float32).ctypes.data,
                             np.ascontiguousarray(cz, np.float32).ctypes.data, out.ctypes.data)
    return out


def pher_hash(pher):
    """hx = (hx ^ bits) * 1099511628211 from 0 over float32 bit patterns."""
    bits = np.ascontiguousarray(pher, np.float32).view(np.uint32)
    h = 0
    M = (1 << 64) - 1
    for b in bits.tolist():
        h = ((h ^ b) * 1099511628211) & M
    return h


def fnv1a_bytes(b):
    h = 1469598103934665603
    M = (1 << 64) - 1
    for v in bytes(b):
        h = ((h ^ v) * 1099511628211) & M
    return h


def run_ref(cmd, out, **kw):
    """Run the real reference (oracle/_ref/ref_harness).  Returns the WAF dict."""
    import waf
    args = [REF_BIN, cmd, "out=%s" % out] + ["%s=%s" % (k, v) for k, v in kw.items()]
    subprocess.check_call(args, stderr=subprocess.DEVNULL)
    return waf.load(out)


def have_ref():
    return os.path.exists(REF_BIN)
