"""CPU, gloo, world_size 2: the N>1 control path of bench.py (barrier-bracketed timing with MAX over ranks, the
per-step gradient all-reduce, per-rank sharding of the units) without touching a GPU."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    import bench
    dev = torch.device("cpu")
    grad = torch.full((bench.DDP_GRAD_ELEMS,), float(rank + 1))
    calls = {"n": 0}

    def step():
        calls["n"] += 1
        time.sleep(0.01 * (rank + 1))          # rank 1 is the slow one
        grad.fill_(float(rank + 1))
        dist.all_reduce(grad)                  # the only exchange of the path: DDP's gradient bucket

    elapsed = bench.timed_region(step, steps=3, warmup=1, device=dev, world=world)
    out[rank] = dict(elapsed=elapsed, calls=calls["n"], grad=float(grad[0]), grad_last=float(grad[-1]),
                     seed=bench.rank_seed(rank), rate=bench.whole_job_rate(world, 4, 3, elapsed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_and_exchange():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    # every rank reports the SAME elapsed = max over ranks, and it covers the slow rank's 3 timed steps
    assert res[0]["elapsed"] == res[1]["elapsed"]
    assert res[0]["elapsed"] >= 3 * 0.02
    # exactly warmup + steps calls per rank
    assert res[0]["calls"] == res[1]["calls"] == 4
    # all-reduce summed the two ranks' gradient vectors (1 + 2), over the whole 1,009,119-element bucket
    assert res[0]["grad"] == res[1]["grad"] == 3.0 and res[0]["grad_last"] == 3.0
    # units are sharded: each rank works on its own reference views; whole-job rate counts both ranks
    assert res[0]["seed"] != res[1]["seed"]
    assert abs(res[0]["rate"] - 2 * 4 * 3 / res[0]["elapsed"]) < 1e-9


def test_algorithmic_bytes_match_survey_totals():
    """DTU config, S=4: forward 925.9 MB, backward 1496.6 MB per reference view."""
    import bench
    fwd = bwd = 0
    for (H, W, C, G, D, scale, itv) in bench.DTU_STAGES:
        px = H * W
        fwd += bench.algorithmic_bytes("mvster_cost_volume_fwd", 1, C, G, D, px, 4)
        fwd += bench.algorithmic_bytes("mvster_depth_regress_fwd", 1, 0, 0, D, px, 0)
        bwd += bench.algorithmic_bytes("mvster_cost_volume_bwd", 1, C, G, D, px, 4)
        bwd += bench.algorithmic_bytes("mvster_softmax_bwd", 1, 0, 0, D, px, 0)
    # SURVEY's forward total is the eval variant (+ px for the confidence map); bench prices training mode
    fwd_eval = fwd + sum(4 * H * W for (H, W, *_rest) in bench.DTU_STAGES)
    assert abs(fwd_eval / 1e6 - 925.9) < 0.2
    assert abs(bwd / 1e6 - 1496.6) < 0.2


RANK_PROGRAM = """
import json, os, sys
import torch
import torch.distributed as dist
dist.init_process_group(backend="gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
if "--fail-on-rank-1" in sys.argv and rank == 1:
    print("rank 1 is about to fail", flush=True)
    sys.exit(3)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"world": world, "sum": float(t), "local_rank": os.environ["LOCAL_RANK"], "argv": sys.argv[1:]}), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_self_launcher_starts_ranks_and_relays_rank0(tmp_path, capsys):
    """`python bench.py --gpus N` without torch.distributed.run: the parent spawns N ranks with the rendezvous
    environment, relays rank 0's JSON line and returns 0 (control path on CPU/gloo with a stand-in rank program)."""
    import json
    import bench
    prog = tmp_path / "rank_program.py"
    prog.write_text(RANK_PROGRAM)
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "1"], timeout_s=120, program=str(prog))
    out = capsys.readouterr().out
    assert rc == 0
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"world": 2, "sum": 3.0, "local_rank": "0", "argv": ["--gpus", "2", "--steps", "1"]}


def test_self_launcher_propagates_a_failing_rank(tmp_path, capsys):
    import bench
    prog = tmp_path / "rank_program.py"
    prog.write_text(RANK_PROGRAM)
    t0 = time.time()
    rc = bench.launch_ranks(2, ["--fail-on-rank-1"], timeout_s=120, program=str(prog))
    assert rc != 0                       # rank 1 died: the surviving rank is terminated, the parent reports failure
    assert time.time() - t0 < 100
    err = capsys.readouterr().err        # ... and what rank 1 printed before it died is replayed, prefixed by its rank
    assert "rank 1 exited" in err and "[rank 1]" in err


def test_exchange_buckets_add_up_to_the_reference_parameter_count():
    import bench
    assert sum(bench.EXCHANGE_BUCKETS) == bench.DDP_GRAD_ELEMS == 1009119


def test_pmc_traffic_goes_null_when_the_kernel_sources_changed(tmp_path):
    """roofline.traffic is only reported while the committed PMC summary was measured on the kernel sources in the tree."""
    import json
    import bench
    fp = bench.source_fingerprint()
    doc = {"some_kernel<1>": {"hbm_bytes_per_launch": 123.0, "TCC_EA0_ATOMIC_sum_avg": 5.0},
           "helper_kernel<2>": {"hbm_bytes_per_launch": 77.0, "TCC_EA0_ATOMIC_sum_avg": 1.0}, "_meta": {"csrc_sha": fp}}
    f = tmp_path / "r9_pmc.json"
    f.write_text(json.dumps(doc))
    traffic, src, entry, why = bench.pmc_traffic("some_kernel<1>", path=str(f))
    assert traffic == 123.0 and why is None and entry["TCC_EA0_ATOMIC_sum_avg"] == 5.0
    # an entry point that launches several kernels: traffic and atomics are summed over them (dominant kernel first)
    traffic, src, entry, why = bench.pmc_traffic("some_kernel<1>;helper_kernel<2>;absent_kernel", path=str(f))
    assert traffic == 200.0 and entry["TCC_EA0_ATOMIC_sum_avg"] == 6.0 and entry["kernels_summed"] == ["some_kernel<1>", "helper_kernel<2>"]
    doc["_meta"]["csrc_sha"] = "0" * 16                       # the sources moved on
    f.write_text(json.dumps(doc))
    traffic, src, entry, why = bench.pmc_traffic("some_kernel<1>", path=str(f))
    assert traffic is None and "stale" in why
    del doc["_meta"]                                          # a summary without provenance is not trusted either
    f.write_text(json.dumps(doc))
    assert bench.pmc_traffic("some_kernel<1>", path=str(f))[0] is None
    assert bench.pmc_traffic("other_kernel", path=str(f))[0] is None


def test_conv_flops_know_every_convolution_entry_point():
    """The e2e object's convolution TFLOP/s counts the inference entry points (weights pre-packed, BatchNorm in the epilogue)
    like the training ones: same layer, same flops."""
    import bench
    N, Cin, Cout, H, W, K = 5, 32, 64, 288, 400, 3
    want = 2.0 * N * Cin * Cout * K * K * H * W
    p = 0                                                      # pointers are opaque to the formula
    assert bench.conv_flops("mvster_conv_mfma_fwd", (p, p, p, p, p, 0, N, Cin, Cout, H, W, K, 1, 0, p)) == want
    assert bench.conv_flops("mvster_conv_mfma_fwd_ep", (p, p, p, p, N, Cin, Cout, H, W, K, 1, 0, 0, p, p, 1, p)) == want
    assert bench.conv_flops("mvster_conv_direct_fwd", (p, p, p, p, N, Cin, Cout, H, W, K, p)) == want
    assert bench.conv_flops("mvster_conv_direct_fwd_ep", (p, p, p, p, N, Cin, Cout, H, W, K, 0, p, p, 1, p)) == want
    assert bench.conv_flops("mvster_conv_direct_fwd_layout", (p, p, p, p, N, Cin, Cout, H, W, K, 1, p)) == want
    tr = 2.0 * N * Cin * Cout * K * K * H * W                  # transposed: per INPUT pixel
    assert bench.conv_flops("mvster_conv_mfma_transposed", (p, p, p, p, 0, N, Cin, Cout, H, W, 2 * H, 2 * W, K, p)) == tr
    assert bench.conv_flops("mvster_conv_mfma_transposed_ep", (p, p, p, N, Cin, Cout, H, W, 2 * H, 2 * W, K, p, p, 1, p)) == tr
    assert bench.conv_flops("mvster_bn_eval_prepare", (p,) * 8) == 0.0
