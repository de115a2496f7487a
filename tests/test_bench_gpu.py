"""The bench line's contract on one GPU (bench.py): one JSON line with the keys the driver parses, the roofline object and
the numbers that round 6 moved to the top level.  A bounded run (MNRF_BENCH_LEGS) of one timed frame."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MNRF_BENCH_LEGS"] = "headline,other,train"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "one JSON line at N = 1"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e5 and abs(d["value"] - d["config"]["rays_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9 and 0.05 < rf["frac"] < 0.34
    assert rf["traffic"] is None or rf["traffic"] > 2e8      # (bytes per launch from profiles/traffic.json; >= the 226 MB algorithmic)
    # round 6: what a harness that keeps only top-level values must still see
    assert d["fp32_rays_per_s"] > 1e5 and 0.5 < d["fp32_frac"] < 1.0
    assert len(d["train_ms"]) == 5 and d["train_ms"][0] > 0 and d["train_ms"][0] == d["train_step"]["ms_per_step"]
    assert d["train_route"] == d["train_step"]["route"] == "graph" and 0.05 < d["train_frac"] < 0.3
    assert d["legs_requested"] == ["headline", "other", "train"]
