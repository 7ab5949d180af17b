"""bench.py end to end on the GPU box at a small step count: the JSON contract the driver parses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "4", "--cpu-gens", "100"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "acs_generations_per_sec" and d["unit"] == "generations/s" and d["value"] > 100
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 4 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.3 < rf["frac"] < 1.0
    assert rf["algorithmic_bytes_per_launch"] == 48.0 * 128 ** 3
    # the roofline figure is the launch of the TIMED loop that carries the sweep (k_evap_rank_mark), stamped per dispatch over
    # the timed region; the sweep kernel launched alone (>= 64 stamped launches, at the BASELINE size and past the Infinity
    # Cache) stands beside it; measured traffic (committed PMC passes) is within 2 % of the algorithmic bytes where it is quoted
    assert rf["kernel"].startswith("k_evap_rank_mark") and rf["sampled_launches"] >= 32
    assert rf["frac_hbm"] == rf["sweep_alone"]["frac_256"] and rf["frac_hbm"] > 0.7   # the HBM-resident sweep (non-temporal by rule past the Infinity Cache)
    sa = rf["sweep_alone"]
    assert sa["kernel"].startswith("k_evaporate") and 0.3 < sa["frac_256"] < 1.0 and sa["frac_128"] >= rf["frac"] * 0.95
    assert sa["sweep_256"]["algorithmic_bytes_per_launch"] == 48.0 * 256 ** 3
    assert abs(sa["traffic_128"] / rf["algorithmic_bytes_per_launch"] - 1.0) < 0.02 and abs(sa["traffic_256"] / (48.0 * 256 ** 3) - 1.0) < 0.02
    assert abs(rf["traffic"] / rf["algorithmic_bytes_per_launch"] - 1.0) < 0.05 and rf["traffic_source"].startswith(("live:", "committed:"))
    ws = d["walk_step"]
    assert ws["instructions_per_step"] > 40 and ws["isa_file"].endswith("walk_loop_isa.txt") and 0.3 < ws["frac_of_issue_floor"] < 1.0
    ms = d["multi_start"]   # eight searches together on one GPU: more problem-generations/s than one alone, an HBM-resident sweep, all converge
    assert ms["problem_generations_per_s"] > 15000 and 0.4 < ms["in_loop_frac"] < 1.0
    assert ms["bytes_per_launch"] == 8 * 48.0 * 128 ** 3 and all(c == 378.0 for c in ms["best_costs"])
    assert ms["lazy_identical_histories"] is True and ms["lazy_problem_generations_per_s"] > ms["problem_generations_per_s"]
    # the batch as the library pipelines it by rule (two groups of slots on streams of their own) against one stream: same histories, not slower
    assert ms["pipelined"]["groups"] == 2 and ms["one_stream"]["groups"] == 1 and ms["pipelined_identical_histories"] is True
    assert ms["pipelined"]["problem_generations_per_s"] > 0.97 * ms["one_stream"]["problem_generations_per_s"]
    mc = d["multi_start_curve"]["rows"]   # problems-per-GPU curve: P = 1..32, dense and lazy, one stream and pipelined
    assert {(r["kind"], r["problems"]) for r in mc} == {(k, P) for k in ("dense", "lazy") for P in (1, 2, 4, 8, 16, 32)}
    assert all(r["identical_to_one_stream"] for r in mc if r["groups"] > 1) and any(r["groups"] == 2 for r in mc)
    ks = d["kernel_ms_sampled"]
    assert ks["generations"] == [0, 10, 20, 30, 40, 50] and ks["walk"] > 0
    rm = d["ref_mode"]    # the reference's own rand() stream on the same grid: converged by generation 200, far faster than a walk per ant
    assert rm["best_cost"] == 378.0 and rm["generations_per_s"] > 40
    fr, c5 = d["full_run"], d["c5_full"]
    assert fr["generations"] == 500 and fr["best_cost"] == 378.0 and fr["generations_per_s"] > d["value"]
    assert c5["all_reached"] is True and c5["slots_by_rule"] * c5["batches"] >= 2016 and c5["t_pairs_s"] < 10
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert cb["generations"] == [0, 59]                      # the CPU baseline is timed on the timed region's own generations
    if cb["kind"] == "reference":
        assert cb["window_100"]["generations"] == [0, 99] and cb["window_100"]["value"] > 0
    cc = d["cost_check"]   # all 60 timed generations replayed by the CPU port: history and final best path equal
    assert cc["bit_equal_trace"] is True and cc["generations"] == 60 and cc["best_path_equal"] is True
    sg = d["stragglers"]   # the hand-over engaged in the timed region and every straggler was finished
    assert sg["handed_over"] > 0 and sg["handed_over"] == sg["finished_by_resume_blocks"]
    assert abs(d["ms_per_step"] * d["value"] / 1e3 - 1.0) < 1e-6


def test_a_failing_extra_does_not_cost_the_line():
    """the extras stand outside `value`: when one of them throws (device memory, say), the driver must still get its JSON line, with the
    failure reported under the extra's key"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu", "--no-roofline-256"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, WA_BENCH_FAIL="all"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 100 and d["steps"] == 10 and d["roofline"]["frac"] > 0.3
    for k in ("walk_step", "full_run", "ref_mode", "multi_start", "multi_start_curve", "c5_pair_planning", "c5_full"):
        assert "WA_BENCH_FAIL" in d[k]["error"], k
    assert "walk_step_extra failed" in r.stderr
