"""GPU: bench.py prints ONE JSON line that honours the driver contract (fields, types, roofline and cpu_baseline objects)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    assert d["metric"].startswith("reference-views/sec (fwd+bwd), DTU 1600") and d["unit"] == "ref-views/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 30 * d["cpu_baseline"]["value"]                      # north_star: >= 30x the CPU baseline
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6  # value = units / time, B = 4 per step
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic"] > rf["algorithmic_bytes_per_launch"] * 0.5
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "ref-views/s" and 1 <= cb["cores"] <= 16 and "sample" in cb
