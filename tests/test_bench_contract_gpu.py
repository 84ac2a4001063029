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
    assert (rf["traffic"] is None and rf["traffic_null_reason"]) or rf["traffic"] > rf["algorithmic_bytes_per_launch"] * 0.5
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "ref-views/s" and 1 <= cb["cores"] <= 16 and "sample" in cb
    assert "3 timed passes" in cb["sample"] and cb["cpu"]
    assert d["config"]["exchange"] == "none"


def _json_line(r):
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_one_rank_process_group_over_rccl():
    """World size 1 under a launcher environment: bench.py joins an RCCL ("nccl") process group and runs the per-stage
    asynchronous gradient all-reduces from the autograd hooks -- the exchange code path of the multi-GPU runs, on one GPU."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    d = _json_line(r)
    assert d["n_gpus"] == 1 and d["config"]["ranks_seen_by_collective"] == 1
    assert "1009119 fp32" in d["config"]["exchange"] and "nccl" in d["config"]["exchange"]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no external launcher (what the driver runs): the parent spawns the ranks itself.
    On the one-GPU test box both ranks share cuda:0 and talk over gloo (--rehearse-on-one-gpu): control path, not a
    measurement."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "1", "--rehearse-on-one-gpu"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    d = _json_line(r)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["config"]["ranks_seen_by_collective"] == 2
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
