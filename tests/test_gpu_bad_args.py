"""include/weldacs.h with VALID handles and nothing else: every entry point that takes a context, grid, solver, trajectory, spline or
communicator is called with that handle (and, where it takes a second handle, that too) and NULL / zero for all other arguments.  The library
must never crash or exit() (errors are status codes, the headers translate them): the child process that makes the calls has to
come back with code 0 and a line per call.  Calls for which zeros are a valid request simply succeed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r'''
import sys
sys.path.insert(0, %r)
import ctypes as C
import numpy as np
from welding_robot_amd import _lib as L, api
lib = L.load()
ctx = api.Context(0)
free = np.ones(12 * 10 * 8, np.uint8)
ax = [np.arange(k, dtype=np.float32) for k in (12, 10, 8)]
grid = api.Grid.from_occupancy(ctx, free, ax[0], ax[1], ax[2], np.float32(1.0), 0)
solvers = {"dense": api.AcsSolver(ctx, grid, 2, 8), "lazy": api.AcsSolver(ctx, grid, 2, 8, lazy=True), "nb26": api.AcsSolver(ctx, grid, 1, 8, neighbourhood=26)}
begun = api.AcsSolver(ctx, grid, 1, 8)
begun.solve(api.default_params(max_iteration=3, predict=8 / 0.35, fixed_colony=8, rng_mode=api.RNG_DEV, seed=1), 0, 12 * 10 * 8 - 1)
solvers["begun"] = begun
traj = api.Trajectory.from_points(ctx, np.zeros((5, 3), np.float32))
spl = api.Bspline(ctx, 3, 0, 0, 0, 5)
comm = api.Comm(ctx, 0, 1, api.Comm.unique_id())
first = {"wa_ctx_": ctx.h, "wa_grid_from": ctx.h, "wa_gtsp": ctx.h, "wa_comm_create": ctx.h, "wa_traj_from": ctx.h, "wa_bspline_create": ctx.h, "wa_last_error": ctx.h,
         "wa_grid_": grid.h, "wa_traj_stitch": grid.h, "wa_acs_memory": grid.h, "wa_acs_straggler_pool": grid.h,
         "wa_traj_": traj.h, "wa_bspline_": spl.h, "wa_comm_": comm.h}
skip = {"wa_ctx_create", "wa_ctx_destroy", "wa_grid_destroy", "wa_acs_destroy", "wa_traj_destroy", "wa_bspline_destroy", "wa_comm_destroy", "wa_version", "wa_device_count",
        "wa_stl_parse", "wa_stl_read_file", "wa_axis_coords", "wa_acs_default_params", "wa_comm_unique_id", "wa_comm_pack_best_key", "wa_comm_unpack_best_key"}


def handle_for(name):
    for pre in sorted(first, key=len, reverse=True):
        if name.startswith(pre):
            return first[pre]
    return None


def zeros(args):
    return [a(0.0) if a in (C.c_float, C.c_double) else None if (a is C.c_void_p or a is C.c_char_p or hasattr(a, "contents")) else a(0) for a in args]


for name in sorted(L.SYMBOLS):
    if name in skip:
        continue
    res, args = L.SYMBOLS[name]
    if name.startswith("wa_acs_") and handle_for(name) is None:
        targets = []
        for tag, s in solvers.items():
            v = zeros(args)
            if name.startswith("wa_acs_create"):
                v[0], v[1] = ctx.h, grid.h          # context + grid, zero slots / colony
            else:
                v[0] = s.h
                if name == "wa_acs_allreduce_best":
                    v[1] = comm.h
            targets.append((tag, v))
    else:
        v = zeros(args)
        v[0] = handle_for(name)
        assert v[0] is not None, name
        if name in ("wa_bspline_set_param_traj",):
            targets = [("", v), ("traj", v[:1] + [traj.h] + v[2:])]
        else:
            targets = [("", v)]
    for tag, v in targets:
        r = getattr(lib, name)(*v)
        print(name, tag or "-", r if res is C.c_int else "-", flush=True)
ctx.sync()
print("done", flush=True)
# (nothing is closed here on purpose: the wrappers close what is still open when the interpreter exits, children first, while the
#  runtimes are still up -- api._close_live_contexts)
''' % ROOT


def test_valid_handles_and_nothing_else_never_crash():
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("wa_") or l == "done"]
    assert r.returncode == 0 and lines and lines[-1] == "done", (lines[-3:], r.stderr[-2500:])
    assert len(lines) > 100
    # creation with zero slots / zero colony, a run of zero generations on a solver that has not begun, reading results nobody asked for:
    # errors, not successes
    status = {}
    for l in lines[:-1]:
        n, tag, v = l.split()
        status[(n, tag)] = v
    for n in ("wa_acs_create", "wa_acs_create_lazy", "wa_acs_create_nb", "wa_grid_from_mesh", "wa_grid_from_occupancy", "wa_traj_from_points", "wa_bspline_create", "wa_comm_create", "wa_gtsp_solve"):
        for (name, tag), v in status.items():
            if name == n:
                assert v != "0", (name, tag)
    assert status[("wa_acs_run", "dense")] != "0" and status[("wa_acs_result", "dense")] != "0"
