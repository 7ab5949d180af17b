"""bench.py end to end on the GPU box at a small step count: the JSON contract the driver parses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "4", "--cpu-gens", "3"],
                       capture_output=True, text=True, timeout=600)
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
    # the roofline figure comes from the sweep kernel launched alone (>= 64 stamped launches whatever --steps is), at the
    # BASELINE size and past the Infinity Cache; measured traffic (committed PMC passes) is within 2 % of the algorithmic bytes
    assert rf["sampled_launches"] >= 64 and rf["kernel"].startswith("k_evaporate")
    assert 0.3 < rf["frac_256"] < 1.0 and rf["sweep_256"]["algorithmic_bytes_per_launch"] == 48.0 * 256 ** 3
    assert abs(rf["traffic"] / rf["algorithmic_bytes_per_launch"] - 1.0) < 0.02 and abs(rf["traffic_256"] / (48.0 * 256 ** 3) - 1.0) < 0.02
    assert d["fused_launch"]["sampled_launches"] >= 32 and d["fused_launch"]["avg_launch_ms"] > rf["avg_launch_ms"] * 0.9
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert d["cost_check"]["bit_equal_trace"] is True  # the GPU's best-cost history equals the CPU port's
    assert abs(d["ms_per_step"] * d["value"] / 1e3 - 1.0) < 1e-6
