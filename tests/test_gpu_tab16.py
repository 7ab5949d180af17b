"""16-bit tabu entries (csrc/acs_walk.hpp WaTabu, csrc/walk_loop_gfx950.hpp T16; VERDICT r05 task 1): the visited set of a walking ant
(ACSRank_3D.hpp:70 `tabu_list`, lookup :144-146, insert :73-79) held in half the LDS -- the 12-bit quotient of a bijective hash + a 4-bit probe
displacement per slot -- for launches with more walk blocks than fit with 32-bit keys.  An entry names its key exactly, so everything
observable must stay bit-identical to the oracle: traces, every ant, best path, whole field; with tables small enough that probe chains run
past what an entry can say (the walk then spills to its bitmap) and with tables the walk outgrows; dense and lazy fields; the rejoin
watch; several searches per launch; and against the same launches with 32-bit keys.  WA_TAB16=1 forces the entries wherever they can
name the grid's ids (they are chosen by rule only when a launch does not fit)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_parity import _dev_vs_oracle, bits, dgrid_from, ogrid
from welding_robot_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


class env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("log2", [6, 8, 10, 12])
def test_entries16_equal_the_oracle_at_every_table_size(ctx, lazy, log2):
    """cubic.stl (25 x 32 x 25 = 20 000 voxels: 15-bit ids, so even a 2^6 table's entries can name them): 64 slots -> nearly every walk spills
    (outgrown or a chain past displacement 13), 2^12 -> none does"""
    og = ogrid("cubic.stl", "0.0219", 8)
    sid, eid = og.resolve(og.node_pt(4, 4, 4)), og.resolve(og.node_pt(20, 27, 20))
    with env(WA_TAB16=1, WA_HASH_LOG2=log2):
        dg = dgrid_from(ctx, og)
        s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=16, lazy=lazy)
        p = api.default_params(max_iteration=3, predict=1.03, fixed_colony=16, rng_mode=api.RNG_DEV, seed=5)
        s.solve(p, sid, eid)
        info = s.walk_info()
        s.close()
        assert info["entries16"] and info["hash_log2"] == log2 and not info["touch_loads"], info
        assert info["lds_bytes_per_block"] == (2 << log2) + 832
        _dev_vs_oracle(ctx, og, sid, eid, 60, 1.03, 16, seed=12345, lazy=lazy)
        _dev_vs_oracle(ctx, og, sid, eid, 150, 5.0, 0, seed=7, stream=3, lazy=lazy)      # adaptive colony, converges: replay + rejoin watch


def test_entries16_on_the_piece_and_on_128_cubed(ctx):
    with env(WA_TAB16=1):
        og = ogrid("simplified_piece.stl", "0.0148", 4)
        _dev_vs_oracle(ctx, og, 2177, 48575, 200, 5.4126, 128, seed=12345)                 # BASELINE config C2
        _dev_vs_oracle(ctx, og, 2177, 48575, 200, 5.4126, 128, seed=12345, lazy=True)
        og = O.synth_grid(128, seed=2024, occ_prob=0.10)
        _dev_vs_oracle(ctx, og, 16513, 2097151, 6, 731.43, 256, seed=12345)                # C3's first generations: 800-1 300-node walks
    with env(WA_TAB16=1, WA_HASH_LOG2=10):                                                 # ... which a 2^10 table cannot hold: spills
        _dev_vs_oracle(ctx, og, 16513, 2097151, 3, 731.43, 256, seed=12345, lazy=True)


def test_an_entry_that_cannot_name_the_ids_is_not_used(ctx):
    """128^3 = 21-bit ids: a 2^8 table would need 13 quotient bits -- the launch keeps 32-bit keys although WA_TAB16=1 asks"""
    og = O.synth_grid(128, seed=2024, occ_prob=0.10)
    with env(WA_TAB16=1, WA_HASH_LOG2=8):
        dg = dgrid_from(ctx, og)
        s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=8)
        s.solve(api.default_params(max_iteration=2, predict=731.43, fixed_colony=8, rng_mode=api.RNG_DEV, seed=5), 16513, 2097151)
        assert not s.walk_info()["entries16"]
        s.close()
    with env(WA_TAB16=1, WA_HASH_LOG2=9):
        s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=8)
        s.solve(api.default_params(max_iteration=2, predict=731.43, fixed_colony=8, rng_mode=api.RNG_DEV, seed=5), 16513, 2097151)
        assert s.walk_info()["entries16"]
        s.close()
    dg.close()


@pytest.mark.parametrize("lazy", [False, True])
def test_a_saturated_batch_takes_entries16_by_rule_and_equals_the_32_bit_run(ctx, lazy):
    """more walk blocks than fit with 32-bit keys -> the rule picks the 16-bit entries; same histories, ants, paths and fields as WA_TAB16=0"""
    from welding_robot_amd import synth
    n, P, ants, gens = 64, 48, 128, 40
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=11, occ_prob=0.12)
    dg = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    pts = synth.synth_weld_points(free, n, 12, seed=3)
    starts, ends = [int(pts[i % 12]) for i in range(P)], [int(pts[(i * 5 + 1) % 12]) for i in range(P)]
    ends = [e if e != s_ else int(pts[(k + 2) % 12]) for k, (s_, e) in enumerate(zip(starts, ends))]
    p = api.default_params(max_iteration=gens, predict=float(ants / 0.35), fixed_colony=ants, rng_mode=api.RNG_DEV, seed=21)
    res = {}
    for mode in (0, -1):
        with env(WA_TAB16=mode):
            s = api.AcsSolver(ctx, dg, n_slots=P, max_colony=ants, lazy=lazy)
            s.solve(p, starts, ends, streams=list(range(P)))
            info = s.walk_info()
            assert info["entries16"] == (mode == -1), (mode, info)
            res[mode] = [(s.trace(q)["bestL"].tobytes(), s.trace(q)["steps"].tobytes(), s.result(q)[1].tobytes(), s.ants(q)[1].tobytes()) for q in range(P)]
            res[(mode, "pher")] = [s.pheromone(q).tobytes() for q in (0, 7, P - 1)]
            res[(mode, "info")] = info
            s.close()
    assert res[0] == res[-1] and res[(0, "pher")] == res[(-1, "pher")]
    # the same LDS holds twice the slots (64^3: the 32-bit rule stops at 2^11 = 8 KB, sixteen resident blocks per CU -- what four wavefronts per SIMD allow --,
    # the 16-bit entries give the same sixteen a 2^12 table); on 256^3, where the 32-bit floor is 2^12 = 16 KB, it is nine blocks against sixteen
    a, b = res[(0, "info")], res[(-1, "info")]
    assert b["resident_blocks_per_cu"] >= a["resident_blocks_per_cu"] and (b["hash_log2"], b["resident_blocks_per_cu"]) > (a["hash_log2"], 0) and \
        b["lds_bytes_per_block"] <= a["lds_bytes_per_block"] and b["hash_log2"] >= a["hash_log2"], (a, b)
    # ... and one of the searches against the oracle
    og = O.Grid(cx, cy, cz, free, prec, wall)
    a = O.Acs(og)
    tr = a.solve(starts[7], ends[7], gens, float(ants / 0.35), fixed_colony=ants, mode=O.DEV, seed=21, stream=7)
    assert res[-1][7][0] == np.ascontiguousarray(tr["bestL"], np.float32).tobytes()
    dg.close()
