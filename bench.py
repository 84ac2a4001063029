#!/usr/bin/env python3
"""bench.py -- ref-views/sec (fwd+bwd) of the MVSTER cost-volume hot path on synthetic DTU-shaped tensors.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

One process per GPU.  `python bench.py --gpus N` with N > 1 starts its own N ranks (fresh child processes, spawned
before the parent makes any GPU call) unless a launcher (torch.distributed.run) already did; ranks talk over
RCCL (torch.distributed backend "nccl").  A "step" is one
forward + backward pass of the hot path over one batch of reference views at the four literal DTU stage
shapes (1600x1184, 1 ref + 4 src views, D = 8,8,4,4, G = 8,8,4,4):

  per stage: hypothesis scheduling (A5/A6) -> poses -> fused cost volume (A2) -> stub regnet (vol.sum(1), the
  callback the reference's reg2d occupies; out of scope) -> depth regression (A4); loss = sum(attn * w);
  backward: softmax backward -> stub regnet backward -> fused cost-volume backward (A3: coefficient kernel + counting sort
  by destination tile + owner kernels, no float atomics).

Every rank works on its own batch (units = reference views, no data-path collective).  With more than one rank
the step carries the only exchange the reference has in training, DDP's gradient all-reduce (train_mvs4.py:562-570):
1,009,119 fp32 per step, here as four asynchronous all-reduces issued from autograd hooks the moment a stage's
backward has produced its reference-feature gradient (payload: the leading elements of that gradient), so that the
collectives overlap the rest of the backward like DDP's buckets do; the step waits for them at its end.

Prints ONE JSON line on rank 0 (see README / the driver contract) with two extra objects:
  roofline     -- dominant kernel's algorithmic HBM bytes per launch / its average duration (HIP events on the
                  launch stream, collected over the timed steps) against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (same torch operator sequence as the reference's CPU path, kind "port")
                  timed on this host's cores on a bounded sample (one reference view; 1 warm-up + 3 timed passes, median).
  e2e          -- after the timed region: the drop-in MVS4net shell's training step (forward, MVS4net_loss ot_iter = 10,
                  backward, Adam) and inference forward at 1600x1152, N = 5, eager and as one HIP graph (--no-e2e skips it).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from collections import defaultdict

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import deep_reconstruction_with_epipolar_lines_mvster_amd as mv           # noqa: E402
from deep_reconstruction_with_epipolar_lines_mvster_amd import _lib      # noqa: E402

# (H, W, C, G, D, intrinsics scale, split_itv)  -- DTU 1600x1184, stage_splits 8,8,4,4, group_cor_dim 8,8,4,4,
# depth_interals_ratio 0.5,0.5,0.5,1 (scripts/train_dtu.sh:20-45 of the reference)
DTU_STAGES = [
    (148, 200, 64, 8, 8, 0.125, 0.5),
    (296, 400, 32, 8, 8, 0.25, 0.5),
    (592, 800, 16, 4, 4, 0.5, 0.5),
    (1184, 1600, 8, 4, 4, 1.0, 1.0),
]
N_SRC = 4
# the other BASELINE.json configurations' stage shapes (parity-test cases; `--config` times them for DESIGN.md, the
# headline line stays the DTU one): (H, W, C, G, D, scale, split_itv), source views, reference views per step
OTHER_CONFIGS = {
    "hires": ([(192, 256, 64, 8, 16, 0.125, 0.5), (384, 512, 32, 8, 8, 0.25, 0.5), (768, 1024, 16, 4, 8, 0.5, 0.5),
               (1536, 2048, 8, 4, 4, 1.0, 1.0)], 9, 1, "High-res 2048x1536, 9 src views, D=(16,8,8,4)"),
    "plumbing": ([(64, 80, 64, 8, 4, 0.125, 0.5), (128, 160, 32, 8, 4, 0.25, 0.5), (256, 320, 16, 4, 4, 0.5, 0.5),
                  (512, 640, 8, 4, 4, 1.0, 1.0)], 3, 1, "640x512, 3 src views, D=(4,4,4,4)"),
}
DEPTH_RANGE = (425.0, 935.0)
ATTN_TEMP = 2.0
DDP_GRAD_ELEMS = 1009119           # parameters of the reference model
HBM_PEAK_GBS = 8000.0              # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_views(B, N, H, W):
    """pinhole f = 2890*(W/1600) px, principal point at the centre, identity rotation,
    translation (30 v, -10 v, 0) mm for view v."""
    mats = torch.zeros(B, N, 2, 4, 4)
    for v in range(N):
        K = torch.eye(4)
        K[0, 0] = K[1, 1] = 2890.0 * W / 1600.0
        K[0, 2], K[1, 2] = W / 2.0, H / 2.0
        E = torch.eye(4)
        E[0, 3], E[1, 3] = 30.0 * v, -10.0 * v
        mats[:, v, 0], mats[:, v, 1] = E, K
    return mats


def scene_features(B, C, H, W, n_views, gen, device):
    """Multi-view-consistent synthetic features: a smooth textured surface Z(X,Y) (mm, inside the DTU depth range)
    seen by the SURVEY 8(d) cameras; channel c of every view samples the same band-limited world texture
    T_c(X,Y), so the group correlation peaks at the true depth and the cascade produces piecewise-smooth depth
    maps like real data does (noise features make them spatially random)."""
    f = 2890.0 * W / 1600.0
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                            torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    dx, dy = (xs - W / 2.0) / f, (ys - H / 2.0) / f

    def surface(X, Y):
        return 650.0 + 110.0 * torch.sin(X / 170.0) * torch.cos(Y / 130.0) + 35.0 * torch.sin(X / 41.0 + 1.0) * torch.sin(Y / 53.0)

    mm_per_px = 650.0 / f
    k = 3
    period = (5.0 + 20.0 * torch.rand(C, k, generator=gen)) * mm_per_px          # 5..25 px at this stage
    angle = 6.2832 * torch.rand(C, k, generator=gen)
    wx = (6.2832 / period * torch.cos(angle)).to(device)
    wy = (6.2832 / period * torch.sin(angle)).to(device)
    ph = (6.2832 * torch.rand(C, k, generator=gen)).to(device)
    feats = []
    for v in range(n_views):
        tx, ty = 30.0 * v, -10.0 * v
        z = torch.full_like(dx, 650.0)
        for _ in range(8):
            z = surface(z * dx - tx, z * dy - ty)
        X, Y = z * dx - tx, z * dy - ty
        fv = torch.sin(wx.view(C, k, 1, 1) * X + wy.view(C, k, 1, 1) * Y + ph.view(C, k, 1, 1)).sum(1)      # [C,H,W]
        feats.append(fv.unsqueeze(0).repeat(B, 1, 1, 1).contiguous().requires_grad_(True))
    return feats


def make_inputs(B, seed, device, data="noise", shapes=None, n_src=None):
    g = torch.Generator().manual_seed(seed)
    stages = []
    n_src = n_src or N_SRC
    for (H, W, C, G, D, scale, itv) in (shapes or DTU_STAGES):
        if data == "scene":
            feats = scene_features(B, C, H, W, n_src + 1, g, device)
        else:
            feats = [torch.randn(B, C, H, W, generator=g).to(device).requires_grad_(True) for _ in range(n_src + 1)]
        mats = synth_views(B, n_src + 1, H, W).to(device)
        if n_src > 4:
            mats[:, :, 0, :2, 3] *= 4.0 / n_src                      # keep the outer views' baselines inside the DTU rig's span
        lw = torch.randn(B, D, H, W, generator=g).to(device)
        stages.append(dict(feats=feats, mats=mats, lw=lw, H=H, W=W, C=C, G=G, D=D, itv=itv))
    dv = torch.tensor([list(DEPTH_RANGE)] * B).to(device)
    return stages, dv


def stub_regnet(vol):
    return vol.sum(1)


def hot_path_step(stages, dv, net, device):
    """Forward + backward of the 4-stage cascade; returns the stage outputs."""
    loss = None
    prev = None
    for k, st in enumerate(stages):
        if k == 0:
            hypo = mv.init_inverse_range(dv, st["D"], device, torch.float32, st["H"], st["W"])
        else:
            hypo = mv.schedule_inverse_range(prev["inverse_min_depth"].detach(), prev["inverse_max_depth"].detach(),
                                             st["D"], st["H"], st["W"])
        prev = net(st["feats"], st["mats"], hypo, stub_regnet, k, group_cor=True, group_cor_dim=st["G"],
                   split_itv=st["itv"])
        term = (prev["attn_weight"] * st["lw"]).sum()
        loss = term if loss is None else loss + term
    loss.backward()
    for st in stages:
        for f in st["feats"]:
            f.grad = None
    return prev


# ---------------------------------------------------------------------------------- algorithmic bytes
def algorithmic_bytes(entry, B, C, G, D, HW, S):
    px = HW * B
    if entry == "mvster_cost_volume_fwd":
        return 4 * (C * px + S * C * px + D * px) + 4 * G * D * px
    if entry == "mvster_cost_volume_bwd":
        return 4 * (G * D * px + C * px + S * C * px + D * px) + 4 * (C * px + S * C * px)
    if entry == "mvster_depth_regress_fwd":
        return 4 * (D * px + D * px + px + 2 * px)
    if entry == "mvster_softmax_bwd":
        return 4 * 3 * D * px
    if entry in ("mvster_init_range", "mvster_schedule_inverse_range"):
        return 4 * D * px
    return 0


def kernel_symbol(entry, B, C, G, D, H, W, S):
    """Name of the dominant gfx950 kernel an ABI entry point launches for this shape, as rocprofv3 prints it.  The
    backward's route comes from the library itself (mvster_cost_volume_bwd_route), not from a mirror of its rules."""
    if entry == "mvster_cost_volume_bwd":
        need = _lib.bwd_workspace(S, B, C, G, D, H, W, H, W, _lib.LAYOUT_NHWC)
        return _lib.bwd_route(S, B, C, G, D, H, W, H, W, _lib.LAYOUT_NHWC, need)[2]
    if entry == "mvster_cost_volume_fwd":
        return "cost_volume_fwd_nhwc_kernel<%d, %d, %d, 1>" % (C, G, D)
    return entry.replace("mvster_", "") + "_kernel"


HOT_PATH_SOURCES = ("cost_volume_*.hip", "cost_volume_common.h", "mvster_device.h", "mvster_host.h", "layout.hip", "pose.hip",
                    "ranges.hip", "regress.hip", "warp.hip")


def source_fingerprint():
    """sha256 over the sources of the hot-path kernels (everything a bench.py step launches): ties a committed PMC summary
    to the code it was taken from.  The F2/F3 translation units (convolutions, BatchNorm, FPN merge, Sinkhorn) and the ABI
    header's declarations are not part of it: they do not change what the summarised kernels do."""
    import glob
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "deep_reconstruction_with_epipolar_lines_mvster_amd", "csrc")
    for f in sorted(f for pat in HOT_PATH_SOURCES for f in glob.glob(os.path.join(csrc, pat))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(symbols, path=None, fingerprint=None):
    """HBM bytes per launch of an ABI entry = the sum over its kernels `symbols` (';'-separated, dominant first) from the
    newest committed PMC summary (tools/pmc_summary.py: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE of
    this same bench command, gfx950 corrections applied).
    Returns (bytes, file, entry of the dominant kernel + summed counters, reason): bytes is None -- with the reason -- when
    no summary covers the dominant kernel or when the summary was taken from other kernel sources than the ones in this
    tree (its "_meta.csrc_sha" differs)."""
    import glob
    names = [s for s in symbols.split(";") if s]
    files = [path] if path else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    fingerprint = fingerprint or source_fingerprint()
    reason = "no PMC summary under profiles/ lists %s" % names[0]
    for f in reversed(files):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        entry = doc.get(names[0])
        if not (entry and "hbm_bytes_per_launch" in entry):
            continue
        rel = os.path.relpath(f, ROOT)
        have = (doc.get("_meta") or {}).get("csrc_sha")
        if have != fingerprint:
            reason = "%s was collected from kernel sources %s, this tree is %s: stale, re-run tools/pmc_pass.sh" % (rel, have, fingerprint)
            continue
        found = [doc[n] for n in names if n in doc and "hbm_bytes_per_launch" in doc[n]]
        total = dict(entry)
        for key in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "TCC_EA0_ATOMIC_sum_avg"):
            total[key] = sum(e.get(key, 0.0) for e in found)
        total["kernels_summed"] = [n for n in names if n in doc]
        return total["hbm_bytes_per_launch"], rel, total, None
    return None, None, None, reason


class LaunchTimer:
    """Brackets every C-ABI call with HIP events on torch's current stream (the stream the kernels launch on)."""

    def __init__(self):
        self.records = defaultdict(list)
        self.enabled = False
        self._orig = _lib.call

    def install(self):
        orig = self._orig
        timer = self

        def timed_call(name, *args):
            if not timer.enabled:
                return orig(name, *args)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            orig(name, *args)
            b.record()
            # mvster_cost_volume_bwd_ws is mvster_cost_volume_bwd plus trailing workspace arguments; _fwd_counted is
            # mvster_cost_volume_fwd plus the tile counters: same leading arguments
            name = {"mvster_cost_volume_bwd_ws": "mvster_cost_volume_bwd",
                    "mvster_cost_volume_fwd_counted": "mvster_cost_volume_fwd"}.get(name, name)
            timer.records[(name, timer.shape_key(name, args))].append((a, b))

        _lib.call = timed_call
        mv.hotpath._lib.call = timed_call

    @staticmethod
    def shape_key(name, args):
        if name == "mvster_cost_volume_fwd":
            B, C, G, D, H, W = args[7:13]
            return (B, C, G, D, H * W, args[2], H, W)
        if name == "mvster_cost_volume_bwd":
            B, C, G, D, H, W = args[10:16]
            return (B, C, G, D, H * W, args[5], H, W)
        if name == "mvster_depth_regress_fwd":
            B, D, H, W = args[8:12]
            return (B, 0, 0, D, H * W, 0)
        if name == "mvster_softmax_bwd":
            B, D, H, W = args[3:7]
            return (B, 0, 0, D, H * W, 0)
        if name == "mvster_init_range":
            B, D, H, W = args[3:7]
            return (B, 0, 0, D, H * W, 0)
        if name == "mvster_schedule_inverse_range":
            B, D, _, _, H, W = args[3:9]
            return (B, 0, 0, D, H * W, 0)
        return (0, 0, 0, 0, 0, 0)

    def summary(self):
        rows = []
        for (name, key), evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            B, C, G, D, HW, S = key[:6]
            H, W = key[6:8] if len(key) >= 8 else (0, 0)
            nbytes = algorithmic_bytes(name, B, C, G, D, HW, S)
            rows.append(dict(entry=name, B=B, C=C, G=G, D=D, HW=HW, S=S, H=H, W=W, launches=len(ms), avg_ms=sum(ms) / len(ms),
                             total_ms=sum(ms), algorithmic_bytes=nbytes))
        rows.sort(key=lambda r: -r["total_ms"])
        return rows


# ---------------------------------------------------------------------------------- CPU baseline (oracle, kind "port")
def host_cores():
    """CPU cores this process may actually use: cgroup quota, then affinity, then the box's share per GPU (16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline():
    from oracle import mvster_oracle as orc
    torch.set_num_threads(host_cores())
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(1)
    dv = torch.tensor([list(DEPTH_RANGE)])
    data = []
    for (H, W, C, G, D, scale, itv) in DTU_STAGES:
        feats = [torch.randn(1, C, H, W, generator=g).requires_grad_(True) for _ in range(N_SRC + 1)]
        data.append((feats, synth_views(1, N_SRC + 1, H, W), torch.randn(1, D, H, W, generator=g)))

    def one_pass(stage_ids):
        prev, loss = None, 0
        for k in stage_ids:
            H, W, C, G, D, scale, itv = DTU_STAGES[k]
            feats, mats, lw = data[k]
            if prev is None:
                hypo = orc.init_inverse_range(dv, D, dv.device, dv.dtype, H, W)
            else:
                hypo = orc.schedule_inverse_range(prev["inverse_min_depth"].detach(), prev["inverse_max_depth"].detach(), D, H, W)
            prev = orc.stage_forward(feats, mats, hypo, stub_regnet, True, group_cor=True, group_cor_dim=G, split_itv=itv,
                                     inverse_depth=True, mono=False, attn_fuse_d=True, attn_temp=ATTN_TEMP)
            loss = loss + (prev["attn_weight"] * lw).sum()
        loss.backward()

    one_pass([0, 1, 2, 3])             # one full untimed warm-up pass
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        one_pass([0, 1, 2, 3])
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    return {"value": 1.0 / dt, "unit": "ref-views/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "1 reference view (B=1), DTU 1600x1184 S=4 D=(8,8,4,4), all 4 stages fwd+bwd of oracle/mvster_oracle.py "
                      "with %d torch threads: 1 warm-up + 3 timed passes (%s s), median %.1f s"
                      % (cores, " / ".join("%.1f" % t for t in times), dt)}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def device_sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_region(step, steps, warmup, device, world):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides;
    returns the elapsed seconds, MAX over ranks -- the driver's timing contract."""
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    device_sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if world > 1:
        dist.barrier()
    device_sync(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def whole_job_rate(world, per_rank_units, steps, elapsed):
    """ref-views/s of the whole job: every rank processed per_rank_units per step (weak scaling)."""
    return world * per_rank_units * steps / elapsed


def rank_seed(rank):
    return 1234 + rank


# ---------------------------------------------------------------------------------- multi-rank launch and exchange
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, timeout_s=1500.0, program=None):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters, so none inherits GPU
    state -- this parent has made no HIP call), wait for all of them, relay rank 0's JSON line.  Any rank failing
    terminates the others and makes the parent exit non-zero; the output of every rank > 0 is kept in a temporary file
    and replayed on stderr, prefixed by its rank, when anything failed (a traceback on rank 3 is not lost).
    Returns the exit code."""
    import tempfile
    port = free_port()
    procs, logs = [], []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL exchanges device-memory handles between the ranks' processes; the hosts this
        # runs on support only the dmabuf IPC path, and with the legacy mode the first collective fails with
        # "hipIpcGetMemHandle: invalid argument".  Inherited if already set; forced here so a bare shell works too.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = None if r == 0 else tempfile.TemporaryFile(mode="w+b")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, program or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else log, stderr=None if r == 0 else subprocess.STDOUT))
    deadline = time.time() + timeout_s
    rc = 0
    pending = set(range(n))
    out0 = b""
    try:
        while pending and rc == 0:
            for r in sorted(pending):
                if r == 0:
                    try:
                        out0 += procs[0].communicate(timeout=0.2)[0] or b""
                    except subprocess.TimeoutExpired:
                        continue
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    print("bench.py: rank %d exited with %d" % (r, code), file=sys.stderr, flush=True)
                    rc = code if code > 0 else 1
            if time.time() > deadline:
                print("bench.py: ranks still running after %.0f s" % timeout_s, file=sys.stderr, flush=True)
                rc = 124
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
        for r, log in enumerate(logs):
            if log is None:
                continue
            if rc != 0:
                log.seek(0)
                for line in log.read().decode(errors="replace").splitlines()[-60:]:
                    print("[rank %d] %s" % (r, line), file=sys.stderr)
            log.close()
        sys.stderr.flush()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


# DDP's gradient traffic per step in the reference: 1,009,119 fp32 parameters.  Split by the stage whose
# backward produces them last-to-first (stage 4 first): reg.3 / reg.2 / reg.1 / reg.0 + feature + mono head.
EXCHANGE_BUCKETS = [195_345, 195_345, 195_345, DDP_GRAD_ELEMS - 3 * 195_345]     # issued after stage 4, 3, 2, 1 backward


class GradientExchange:
    """Stand-in for DistributedDataParallel on the hot-path benchmark (which has no parameters: the regnet is a stub).
    One asynchronous all-reduce per stage, issued from an autograd hook as soon as that stage's backward has written its
    reference-feature gradient; the payload is the leading elements of that gradient (real gradient data, the byte count
    of the reference's DDP traffic).  The collectives run on RCCL's stream and overlap the backward kernels of the
    remaining stages; wait() joins them at the end of the step."""

    def __init__(self, stages):
        self.work = []
        self.elems = 0
        self.buffers = []
        order = list(range(len(stages) - 1, -1, -1))                 # backward visits stage 4 first
        for n, k in zip(EXCHANGE_BUCKETS, order):
            leaf = stages[k]["feats"][0]
            n = min(n, leaf.numel())
            buf = torch.empty(n, dtype=torch.float32, device=leaf.device)
            self.buffers.append(buf)
            self.elems += n
            leaf.register_hook(self._hook(buf, n))

    def _hook(self, buf, n):
        def on_grad(grad):
            buf.copy_(grad.reshape(-1)[:n])
            self.work.append(dist.all_reduce(buf, async_op=True))
            return None
        return on_grad

    def wait(self):
        for w in self.work:
            w.wait()
        self.work.clear()


# ---------------------------------------------------------------------------------- end-to-end step (SURVEY 8d ii)
FP32_MATRIX_PEAK_TFLOPS = 157.3      # v_mfma_f32_32x32x2_f32 / vector fp32, MI355X (MI355X_MICROARCH.md)


def conv_flops(name, a):
    """FLOPs of one convolution C-ABI call from its arguments (include/mvster_hip.h); 0 for every other entry."""
    if name == "mvster_conv_mfma_fwd":
        N, Cin, Cout, H, W, K, st = a[6:13]
        return 2.0 * N * Cin * Cout * K * K * (H // st) * (W // st)
    if name == "mvster_conv_mfma_wgrad":
        N, Cin, Cout, H, W, K, st = a[5:12]
        return 2.0 * N * Cin * Cout * K * K * (H // st) * (W // st)
    if name == "mvster_conv_mfma_depth3_fwd":
        N, C, Cout, H, W, K = a[6:12]
        return 2.0 * N * 3 * C * Cout * K * K * H * W
    if name == "mvster_conv_mfma_depth3_wgrad":
        N, C, Cout, H, W, K = a[5:11]
        return 2.0 * N * 3 * C * Cout * K * K * H * W
    if name == "mvster_conv_mfma_transposed":
        N, Cin, Cout, H, W = a[5:10]
        return 2.0 * N * Cin * Cout * a[12] * a[12] * H * W
    if name in ("mvster_conv_direct_fwd", "mvster_conv_direct_fwd_layout", "mvster_conv_direct_fwd_ep"):
        N, Cin, Cout, H, W, K = a[4:10]
        return 2.0 * N * Cin * Cout * K * K * H * W
    if name == "mvster_conv_mfma_fwd_ep":          # (x, packed, bias, y, N, Cin, Cout, H, W, K, stride, ...): Cin counts the virtual channels
        N, Cin, Cout, H, W, K, st = a[4:11]
        return 2.0 * N * Cin * Cout * K * K * (H // st) * (W // st)
    if name == "mvster_conv_mfma_transposed_ep":   # (x, packed, y, N, Cin, Cout, H, W, Ho, Wo, K, ...)
        N, Cin, Cout, H, W = a[3:8]
        return 2.0 * N * Cin * Cout * a[10] * a[10] * H * W
    if name == "mvster_conv_direct_wgrad":
        N, Cin, Cout, H, W, K = a[6:12]
        return 2.0 * N * Cin * Cout * K * K * H * W
    return 0.0


def e2e_measure(device, steps=5, height=1152, width=1600, views=5, batches=(1, 4)):
    """The reference's training step (train_mvs4.py:314,335,344: forward, MVS4net_loss with ot_iter = 10, backward, Adam) and
    its inference forward (test_mvs4.py:408-415) on the drop-in MVS4net shell at 1600x1152 (1184 is illegal for reg2d,
    SURVEY 8d), random-init weights, synthetic images: eager and as one HIP graph.  Informational: the headline metric is
    the hot path above.  Returns the `e2e` object of the JSON line."""
    from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net as _m
    torch.backends.cudnn.benchmark = False
    H, W, N = height, width, views
    out = {"shape": "%dx%d, N=%d views (1 ref + %d src), D=(8,8,4,4), G=(8,8,4,4), inverse depth, mono head, Adam, fp32" % (W, H, N, N - 1),
           "note": "end-to-end MVS4net shell; eager = ~1400 kernel launches issued from Python, graph = the same step as one HIP graph"}
    orig_call = _lib.call
    for B in batches:
        torch.manual_seed(0)
        model = mv.MVS4net(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
                           stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True,
                           group_cor_dim=[8, 8, 4, 4], inverse_depth=True, mono=True, attn_temp=2).to(device)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        imgs = [torch.rand(B, 3, H, W, device=device) for _ in range(N)]
        proj = {"stage%d" % (k + 1): synth_views(B, N, H // sc, W // sc).to(device) for k, sc in enumerate((8, 4, 2, 1))}
        dv = torch.tensor([list(DEPTH_RANGE)] * B, device=device)
        gts = {"stage%d" % (k + 1): 480 + 400 * torch.rand(B, H // sc, W // sc, device=device) for k, sc in enumerate((8, 4, 2, 1))}
        masks = {k: (torch.rand_like(v) < 0.7).float() for k, v in gts.items()}
        kw = dict(stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)

        def train_step():
            opt.zero_grad(set_to_none=True)
            loss = mv.MVS4net_loss(model(imgs, proj, dv), gts, masks, **kw)[0]
            loss.backward()
            opt.step()

        def eval_step():
            with torch.no_grad():
                return model(imgs, proj, dv)["stage4"]["depth"]

        def timed(step):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(device)
            return 1e3 * (time.perf_counter() - t0) / steps

        def as_graph(step, train):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                    # allocator warm-up on the capture side stream
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            if train:
                opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                step()
            graph.replay()
            return graph.replay

        for mode, step in (("train", train_step), ("eval", eval_step)):
            if mode == "eval" and B != batches[0]:
                continue                                     # inference runs one reference view at a time (test_mvs4.py:947)
            model.train(mode == "train")
            torch.cuda.reset_peak_memory_stats(device)
            step()
            step()
            # one instrumented eager step: C-ABI launches and the convolutions' share (HIP events around each call)
            calls, conv = [0], []

            def counted(name, *a):
                calls[0] += 1
                fl = conv_flops(name, a)
                if fl == 0.0:
                    return orig_call(name, *a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig_call(name, *a)
                e1.record()
                conv.append((fl, e0, e1))
                return r

            _lib.call = _m._lib.call = mv.hotpath._lib.call = counted
            try:
                step()
                torch.cuda.synchronize(device)
            finally:
                _lib.call = _m._lib.call = mv.hotpath._lib.call = orig_call
            conv_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in conv)
            conv_tf = sum(fl for fl, _, _ in conv) / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None
            eager = timed(step)
            rec = {"eager_ms": round(eager, 2)}
            try:
                replay = as_graph(step, mode == "train")
                rec["graph_ms"] = round(timed(replay), 2)
            except Exception as exc:                           # noqa: BLE001 -- report, keep the bench line alive
                rec["graph_ms"] = None
                rec["graph_error"] = str(exc)[:200]
            best = min(v for v in (rec["eager_ms"], rec.get("graph_ms")) if v)
            rec.update({"ref_views_per_s": round(B / (best * 1e-3), 2), "peak_mem_GB": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
                        "hip_abi_calls_per_step": calls[0], "conv_ms_eager": round(conv_ms, 2),
                        "conv_TFLOPs": round(conv_tf, 1) if conv_tf else None,
                        "conv_frac_of_fp32_matrix_peak": round(conv_tf / FP32_MATRIX_PEAK_TFLOPS, 3) if conv_tf else None})
            out["%s_B%d" % (mode, B)] = rec
        del model, opt, imgs, gts, masks
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="reference views per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scene", action="store_true", help="skip the informational run of the same step on scene features")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end MVS4net step timed after the headline region")
    ap.add_argument("--data", default="noise", choices=["noise", "scene"],
                    help="noise: N(0,1) features (SURVEY 8d; spatially random depth, worst case for gather/scatter); "
                         "scene: multi-view-consistent textured surface (smooth depth like real data)")
    ap.add_argument("--config", default="dtu", choices=["dtu"] + sorted(OTHER_CONFIGS),
                    help="dtu: the headline workload (BASELINE.json metric); hires / plumbing: the other BASELINE configurations' "
                         "stage shapes, timed for DESIGN.md (their JSON line says so in config.workload; not the headline)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank REHEARSAL on a single GPU: every rank uses cuda:0 and the exchange runs over gloo "
                         "(RCCL refuses two ranks on one device); exercises the N>1 control path, not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing above has touched the GPU (imports only).
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    distributed = "WORLD_SIZE" in os.environ        # a launcher started us: join its process group, even when it has one rank
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = "gloo" if args.rehearse_on_one_gpu else "nccl"
        if backend == "gloo":
            dist.init_process_group(backend="gloo", init_method="env://")
        else:
            dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
    _lib.load()

    B = args.batch
    other = OTHER_CONFIGS.get(args.config)
    if other is not None:
        B = other[2] if args.batch == 4 else args.batch
    stages, dv = make_inputs(B, seed=rank_seed(rank), device=device, data=args.data, shapes=other and other[0], n_src=other and other[1])
    net = mv.stagenet(inverse_depth=True, mono=True, attn_fuse_d=True, attn_temp=ATTN_TEMP).train()
    exchange = GradientExchange(stages) if distributed else None
    timer = LaunchTimer()
    timer.install()

    def step():
        hot_path_step(stages, dv, net, device)
        if exchange is not None:
            exchange.wait()

    timer.enabled = False
    for _ in range(args.warmup):
        step()
    timer.enabled = True
    elapsed = timed_region(step, args.steps, 0, device, world)
    timer.enabled = False

    ranks_seen = world
    if distributed:
        t = torch.ones(1, device=device)
        dist.all_reduce(t)
        ranks_seen = int(t.item())

    if rank == 0:
        rows = timer.summary()
        dom = rows[0]
        achieved = dom["algorithmic_bytes"] / (dom["avg_ms"] * 1e-3) / 1e9
        kernels = [{"entry": r["entry"], "shape": "B%d C%d G%d D%d px%d S%d" % (r["B"], r["C"], r["G"], r["D"], r["HW"], r["S"]),
                    "avg_ms": round(r["avg_ms"], 4), "launches": r["launches"],
                    "GBps": round(r["algorithmic_bytes"] / (r["avg_ms"] * 1e-3) / 1e9, 1)} for r in rows[:8]]
        symbol = kernel_symbol(dom["entry"], dom["B"], dom["C"], dom["G"], dom["D"], dom["H"], dom["W"], dom["S"])
        if B == 4 and args.data == "noise" and other is None:
            traffic, traffic_src, pmc, why = pmc_traffic(symbol)
        else:
            traffic, traffic_src, pmc, why = None, None, None, "PMC summaries are collected for --batch 4 --data noise only"
        out = {
            "metric": "reference-views/sec (fwd+bwd), DTU 1600x1184 N=5 D=(8,8,4,4)" if other is None else
                      "reference-views/sec (fwd+bwd), " + other[3] + " [not the headline metric]",
            "value": whole_job_rate(world, B, args.steps, elapsed),
            "unit": "ref-views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if args.data == "noise" else "synthetic (multi-view-consistent scene features)",
            "scope": "hot path only: the regnet callback is a stub (reg2d is illegal at H=1184, SURVEY 8d); the end-to-end "
                     "training / inference step at 1600x1152 is in the `e2e` object (measured after the timed region)",
            "config": {"workload": ("MVSTER cost-volume hot path fwd+bwd (stub regnet), DTU 1600x1184 stage shapes "
                                    "148x200/296x400/592x800/1184x1600, 1 ref + 4 src views, D=(8,8,4,4), G=(8,8,4,4), "
                                    "C=(64,32,16,8), inverse depth, attn_temp 2") if other is None else
                       ("NOT the headline workload -- MVSTER cost-volume hot path fwd+bwd (stub regnet), " + other[3]),
                       "ref_views_per_gpu_per_step": B, "global_batch": B * world,
                       "exchange": "none" if exchange is None else
                       "%d fp32 per step (the reference's DDP gradient volume) as %d async all-reduces over %s issued from "
                       "autograd hooks as each stage's backward completes (payload: leading elements of that stage's "
                       "reference-feature gradient); stand-in for DDP, the hot path has no parameters"
                       % (exchange.elems, len(exchange.buffers), backend),
                       "ranks_seen_by_collective": ranks_seen},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": (traffic_src + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of "
                                            "this command (B=4), bytes = 2*FETCH_KiB*1024 + WRITE_KiB*1024 (gfx950 FETCH_SIZE "
                                            "halving verified on kernels with known byte counts); source fingerprint matches")
                         if traffic_src else None,
                         "traffic_null_reason": why,
                         # integer slot claims of the counting sort (one per occupied LDS table entry per workgroup); the
                         # backward issues no float atomics
                         "int_atomic_requests_per_launch": pmc and pmc.get("TCC_EA0_ATOMIC_sum_avg"),
                         "kernel": "%s = %s [B%d C%d G%d D%d px%d S%d]" % (dom["entry"], symbol.replace(";", " + "), dom["B"],
                                                                          dom["C"], dom["G"], dom["D"], dom["HW"], dom["S"]),
                         "note": "achieved / avg_launch_ms are the whole entry (all kernels of the route, HIP events around the "
                                 "C-ABI call); traffic and atomic requests are summed over the same kernels",
                         "avg_launch_ms": round(dom["avg_ms"], 4), "launches_timed": dom["launches"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes"]},
            "kernels": kernels,
        }
        # the forward kernels' own ceiling is not HBM but the L1 / texture path: every (sample, view) pulls four taps of C
        # floats into registers (tools/probes/gather_probe.hip measures what that path delivers for this access pattern)
        fwd_rows = [r for r in rows if r["entry"] == "mvster_cost_volume_fwd"]
        if fwd_rows:
            f0 = max(fwd_rows, key=lambda r: r["total_ms"])
            gathered = 4.0 * f0["C"] * 4 * f0["B"] * f0["HW"] * f0["D"] * f0["S"]
            out["gather_path"] = {
                "kernel": "mvster_cost_volume_fwd [B%d C%d G%d D%d px%d S%d]" % (f0["B"], f0["C"], f0["G"], f0["D"], f0["HW"], f0["S"]),
                "bytes_into_registers_per_launch": gathered, "avg_launch_ms": round(f0["avg_ms"], 4),
                "achieved_TBps": round(gathered / (f0["avg_ms"] * 1e-3) / 1e12, 2),
                "measured_ceiling_TBps": "10.5-16.2 (profiles/r3_gather_probe.txt: the same 128 B per sample gathered with no arithmetic, "
                                         "by address pattern)",
                "note": "informational: roofline above prices the dominant entry against HBM as the contract asks"}
        if world == 1 and not args.no_scene and other is None and args.data == "noise" and B == 4:
            # the same step on multi-view-consistent features (smooth depth, like real scenes: the hypotheses of neighbouring
            # pixels land next to each other in the source views); after and outside the timed region, informational
            stages.clear()
            torch.cuda.empty_cache()
            try:
                stages2, dv2 = make_inputs(B, seed=rank_seed(rank), device=device, data="scene")
                step2 = lambda: hot_path_step(stages2, dv2, net, device)          # noqa: E731
                el2 = timed_region(step2, 10, 3, device, 1)
                out["scene_data"] = {"value": round(B * 10 / el2, 1), "unit": "ref-views/s", "ms_per_step": round(1e3 * el2 / 10, 3),
                                     "steps": 10, "warmup": 3,
                                     "note": "same workload and code on synthetic multi-view-consistent scene features (--data scene)"}
                stages2.clear()
            except Exception as exc:                               # noqa: BLE001 -- the headline line must survive
                out["scene_data"] = {"error": str(exc)[:300]}
        if world == 1 and not args.no_e2e and other is None:
            # after and outside the timed region; frees the hot-path inputs first
            stages.clear()
            torch.cuda.empty_cache()
            try:
                out["e2e"] = e2e_measure(device)
            except Exception as exc:                               # noqa: BLE001 -- the headline line must survive
                out["e2e"] = {"error": str(exc)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
