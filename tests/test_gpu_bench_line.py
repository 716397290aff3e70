"""The driver's contract for bench.py, checked on the real thing: one JSON line on stdout with the agreed keys, `roofline` and (when not switched off) `cpu_baseline`,
measured by the product path.  A short run (3 steps) with the side legs off; the full line is profiles/rNN/bench_b16.json."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_line_with_the_contract_keys(gpu_required):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--sustain-seconds", "0", "--no-reference-precision",
                        "--no-fp16-leg", "--no-trained-leg", "--no-train-leg", "--no-single-tile-leg", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"{len(lines)} lines on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "tiles_per_sec_512x512x3" and d["unit"] == "tiles/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 100 and abs(d["value"] - 16 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["lane_streams"] == "independent"
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0.05 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # two lanes on independent streams: per-kernel numbers from the one-lane pass, the overlapped ones beside them
    assert "ONE lane" in rf["measured_on"] and rf["in_timed_region"]["avg_launch_ms"] > rf["avg_launch_ms"]
