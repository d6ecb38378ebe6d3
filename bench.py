#!/usr/bin/env python3
"""Headline benchmark: PhysicsAttention forward throughput (points/s) on a 1M-point
irregular mesh, M = 64 slices (BASELINE.json metric / config C3), on N GPUs of one node.

A step = one forward of Physics_Attention_Irregular_Mesh(dim=256, heads=8, dim_head=32,
slice_num=64, rotate=45) over one synthetic cloud of `--points` points resident in HBM
(the module-level workload of SURVEY.md 8d).  With N > 1 the SAME cloud is point-sharded
over the ranks (strong scaling, config C4): one all-gather of the slice records per forward.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     -- the dominant hand-written kernel (largest share of the step): algorithmic bytes
                  / measured duration (HIP events on the launch stream) against the 8 TB/s HBM
                  peak; `all_kernels_frac` lists the same fraction for every streaming kernel
  dtype        -- "bf16x3": every product is three bf16 MFMAs (hi*hi + hi*lo + lo*hi) with fp32
                  accumulation and fp32 I/O, i.e. fp32-class results on the bf16 matrix cores
  cpu_baseline -- the CPU oracle (op-for-op port of the reference's forward) timed on the
                  host cores of this box on the same 1M-point input (rank 0, N = 1 only)
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
STAGES = ("project", "slice", "merge", "erwin", "deslice", "to_out")  # erwin = positions + tree + Erwin kernel
# algorithmic HBM bytes per point of the hand-written streaming kernels (DESIGN.md s.4 / SURVEY.md 8d),
# fp32 I/O, D = dim = H*C: projection GEMMs read D and write D floats, the slice pass reads D, the
# de-slice pass reads and writes D.
BYTES_PER_POINT = {"project": lambda D: 8 * D, "slice": lambda D: 4 * D, "deslice": lambda D: 8 * D,
                   "to_out": lambda D: 8 * D}  # (the projection's 4*H B/point of s and the passes' reads of it: not credited)
KERNEL_OF_STAGE = {"project": "gemm_ws_split", "slice": "slice", "deslice": "deslice", "to_out": "gemm_ws"}


def build_module(dev):
    from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh
    import torch.nn as nn

    torch.manual_seed(0)
    mod = Physics_Attention_Irregular_Mesh(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45)
    # the weight-init rule the reference's Model applies (HAETransolver_Irregular_Mesh.py:279-295)
    for m in mod.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.constant_(m.bias, 0)
    return mod.eval().to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the informational training-step leg (profiling runs: "
                    "forward kernels only in the summary)")
    ap.add_argument("--sweep", action="store_true",
                    help="the reference's own harness instead of the headline bench: forward time and peak memory of "
                         "HAETransolver_Irregular_Mesh at 1e3 .. 3e6 points (benchmarks/00-VelocityMemMocker/"
                         "velocity-mem-checker.py), one JSON line")
    ap.add_argument("--sweep-fp32", action="store_true", help="--sweep without torch.autocast (the reference runs under it)")
    args = ap.parse_args()
    if args.sweep:
        raise SystemExit(sweep_main(args))

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python3 bench.py --gpus N`: become the launcher -- N children under torch.distributed.run, started
        # BEFORE anything here touches the GPU (no exec from a process that holds the device), exit with their code
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    rehearsal = world > n_dev  # several ranks per GPU (HAET_BENCH_REHEARSE, see self_launch): gloo moves the records
    dev = torch.device("cuda", local % max(n_dev, 1))
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from haet_amd import _lib
    from haet_amd.sharding import shard_bounds, shard_points

    _lib.load()
    mod = build_module(dev)
    if world > 1:
        shard_points(mod)

    N = args.points
    a, b = shard_bounds(N, rank, world)
    g = torch.Generator().manual_seed(1234)
    # the full synthetic cloud is generated identically on every rank, each keeps its shard
    x_full = torch.randn(1, N, 256, generator=g)
    x = x_full[:, a:b].contiguous().to(dev)
    if not (rank == 0 and world == 1 and not args.no_cpu_baseline):
        del x_full

    # A FULL collection of this process's heap (torch and what it imports lazily: ~10^6 tracked objects) takes ~75 ms
    # (tools/stall_probe.py) -- 50 steps of the timed region if one lands there.  Everything alive now is moved out of the
    # collector's reach; the collector stays ON and walks what the loop itself allocates.
    gc.collect()
    gc.freeze()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def event_pool(n_forwards):
        # Created AND recorded once up front: torch creates the hipEvent lazily at the first record(),
        # ~60 us of host time each; 14 per forward would make a small forward launch-bound.
        pool = [torch.cuda.Event(enable_timing=True) for _ in range((len(STAGES) + 1) * n_forwards)]
        for e in pool:
            e.record()
        return pool

    with torch.no_grad():
        # Untimed priming before the W warm-up steps, under the conditions of the timed loop (stage
        # events on, host running ahead of the GPU): the first time the launch queue gets deep the HIP
        # runtime grows its pools, one stall of ~40 ms somewhere in the first ~15 forwards -- which is
        # 0.8 ms per step on a 50-step run of a 125k-point shard (0.5 ms of GPU work per forward).
        prime = max(0, 64 - args.warmup)
        mod.stage_events, mod.event_pool = [], event_pool(prime)
        for _ in range(prime):
            out = mod(x)
        mod.stage_events, mod.event_pool = None, None
        for _ in range(args.warmup):
            out = mod(x)
        mod.stage_events = []  # per-stage HIP events on the launch stream, read after the loop
        mod.event_pool = event_pool(args.steps)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = mod(x)
        t_launch = time.perf_counter() - t0
        sync()
        dt = time.perf_counter() - t0
        if os.environ.get("HAET_BENCH_DEBUG"):
            print(f"[debug] rank {rank}: host launch path {1e3 * t_launch / args.steps:.3f} ms, "
                  f"drained {1e3 * dt / args.steps:.3f} ms per forward", file=sys.stderr)
    events = mod.stage_events
    mod.stage_events = None
    mod.event_pool = None
    if world > 1:
        t = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stage_ms = {s: 0.0 for s in STAGES}
    for name, e0, e1 in events:
        stage_ms[name] += e0.elapsed_time(e1)
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}

    if rank == 0:
        ms = dt / args.steps * 1e3
        # dominant kernel = the hand-written streaming kernel with the largest share of the step.  When the de-slice
        # pass ran fused with the to_out GEMM (haet_deslice_out_fwd_split: out_x never written) the "deslice" stage IS
        # that kernel (1 KB/point of cells in, 1 KB/point of output out) and the "to_out" stage is the dropout only.
        fused_out = stage_ms["to_out"] < 0.1 * stage_ms["deslice"]
        kernel_of = dict(KERNEL_OF_STAGE)
        candidates = dict(BYTES_PER_POINT)
        if fused_out:
            kernel_of["deslice"] = "deslice_out"
            candidates.pop("to_out")
        dom = max(candidates, key=lambda k: stage_ms[k])
        n_local = b - a
        gbs = n_local * BYTES_PER_POINT[dom](256) / (stage_ms[dom] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic_bytes(kernel_of[dom], N) if world == 1 else (None, None)
        res = {
            "metric": "forward points/sec + L2 vs ref, 1M-pt irregular mesh, M=64 slices",
            "value": N * args.steps / dt, "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16x3", "data": "synthetic",
            "config": {"workload": "configs[2]: Physics_Attention_Irregular_Mesh forward, "
                       f"N={N} points, dim=256, heads=8, dim_head=32, M=64, Erwin defaults, B=1",
                       "points": N, "slices": 64, "sharding": f"points/{world}"},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "stage": dom, "achieved": gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         # the whole forward against SURVEY.md 8d's fused floor of 3*D*4 B/point (x read once per
                         # pass, the output written once) at the HBM peak
                         "path_frac": N * 3 * 256 * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "all_kernels_frac": {k: round(n_local * BYTES_PER_POINT[k](256) / (stage_ms[k] * 1e-3)
                                                       / 1e9 / HBM_PEAK_GBS, 4) for k in candidates}},
        }
        if world == 1:
            # the vendor peak AND a streaming-copy ceiling measured on this box (a 1 GB fp32 torch copy_:
            # read + write), outside the timed region
            ceil = copy_ceiling_gbs(dev)
            res["roofline"]["copy_ceiling"] = round(ceil, 1)
            res["roofline"]["frac_of_copy_ceiling"] = gbs / ceil
        if rehearsal:
            res["rehearsal"] = f"{world} ranks on {n_dev} GPU(s), records over gloo: functional run, not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            res.update(cpu_leg(mod, x_full, out))
        if world == 1 and not args.no_train_leg:
            res.update(train_leg(mod, x))
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def copy_ceiling_gbs(dev, mb: int = 1024, reps: int = 10) -> float:
    """GB/s of a device-to-device copy of `mb` MB (bytes read + bytes written over the time of `reps` copies)."""
    a = torch.empty(mb * 2 ** 20 // 4, dtype=torch.float32, device=dev).normal_()
    b_ = torch.empty_like(a)
    for _ in range(3):
        b_.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b_.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * a.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def sweep_main(args):
    from haet_amd import _lib, harness

    _lib.load()
    torch.cuda.set_device(0)
    rows = harness.sweep(autocast=not args.sweep_fp32, timed=max(1, min(args.steps, 10)), log=lambda m: print(m, file=sys.stderr))
    print(json.dumps({"metric": "forward seconds and peak memory vs points, HAETransolver_Irregular_Mesh "
                                "(velocity-mem-checker.py sweep)", "unit": "s", "n_gpus": 1, "data": "synthetic",
                      "config": {"workload": "n_hidden=256, n_layers=1, n_head=8, mlp_ratio=2, out_dim=4, slice_num=32, "
                                             "space_dim=3, fun_dim=1, batch 1, torch.rand inputs",
                                 "autocast": not args.sweep_fp32}, "rows": rows}), flush=True)
    return 0


def visible_gpu_count() -> int:
    """GPUs this process may use, WITHOUT touching the HIP runtime (the launcher must stay strictly pre-GPU: it
    starts the ranks as children): the KFD topology in sysfs lists one node per agent, GPU nodes are those with
    simd_count > 0; *_VISIBLE_DEVICES narrows the set like the runtime would."""
    import glob

    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:  # no KFD topology visible (some containers): NVML-style count, which does not create a HIP context
        return torch.cuda.device_count()
    n = 0
    for props in nodes:
        try:
            with open(props) as fh:
                fields = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(fields.get("simd_count", "0")) > 0:
                n += 1
        except OSError:
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def self_launch(args):
    """Run this script as N ranks on this node and return their exit code."""
    import socket
    import subprocess

    n_dev = visible_gpu_count()
    env = dict(os.environ)
    if n_dev < args.gpus:
        if not os.environ.get("HAET_BENCH_REHEARSE") or n_dev < 1 or args.gpus > 6:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node shows {n_dev} "
                  "(HAET_BENCH_REHEARSE=1 runs up to 6 ranks on the GPUs there are, over gloo: a functional "
                  "rehearsal, not a measurement)", file=sys.stderr)
            return 3
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def train_leg(mod, x, steps=5):
    """Extra, not the headline metric: one training step of the same module on the same cloud (train mode,
    forward + backward through the hand-written backward kernels).  Never fails the bench."""
    try:
        state = {k: v.clone() for k, v in mod.state_dict().items()}
        mod.train()
        xg = x.detach().requires_grad_(True)
        gout = torch.randn_like(x)

        def step():
            mod.zero_grad(set_to_none=True)
            xg.grad = None
            mod(xg).backward(gout)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        peak = torch.cuda.max_memory_allocated() / 1e9
        mod.eval()
        mod.zero_grad(set_to_none=True)
        mod.load_state_dict(state)
        return {"train_step": {"ms_per_step": 1e3 * dt, "points_per_s": x.shape[1] / dt, "steps": steps,
                               "what": "forward + backward of the module, train mode, fp32 gradients",
                               "peak_memory_gb": round(peak, 2)}}
    except Exception as exc:  # noqa: BLE001 -- informational leg only
        return {"train_step": {"error": repr(exc)}}


def kernel_source_id():
    """Identifies the build the numbers belong to: a hash over the kernel sources (haet_amd/csrc)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "haet_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "haet_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_bytes(kernel, n_points):
    """(HBM bytes per launch of `kernel`, source file) from the committed PMC pass of THIS build
    (profiles/*pmc_traffic.json whose `source_id` matches the kernel sources; FETCH_SIZE corrected x2 for
    gfx950 + WRITE_SIZE, MI355X_MICROARCH.md HBM section) -- (None, None) when no pass of this build and this
    size is committed: a byte count of another build is not this run's traffic."""
    import glob

    sid = kernel_source_id()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), key=os.path.getmtime, reverse=True):
        d = json.load(open(f))
        k = d.get("kernels", {}).get(kernel)
        if k and d.get("points") == n_points and d.get("source_id") == sid:
            return (2.0 * k["fetch_kb_raw"] + k["write_kb"]) * 1024.0, os.path.basename(f)
    return None, None


def host_cores():
    """Threads for the CPU leg: the cores this process may really use (affinity mask, cgroup
    quota); a GPU box exposes all 256 host threads but grants one GPU's share of them (16),
    and 256 oversubscribed threads run the CPU path ~4x slower than 16."""
    env = os.environ.get("HAET_CPU_THREADS")
    if env:
        return int(env)
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n if n <= 32 else 16


def cpu_leg(mod, x_full, gpu_out):
    """CPU oracle on the same input: the reported CPU baseline (kind 'port': the reference
    itself cannot travel to this box) and the full-size rel-L2 of the GPU result."""
    from oracle import haet_oracle as O

    n_thr = host_cores()
    torch.set_num_threads(n_thr)
    cfg = O.AttnCfg(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45)
    sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
    times = []
    with torch.no_grad():
        for i in range(3):  # 1 warm-up + 2 timed forwards of the full cloud (~6 s each on 16 cores)
            t0 = time.perf_counter()
            ref = O.physics_attention_forward(x_full, sd, cfg)
            times.append(time.perf_counter() - t0)
    best = min(times[1:])
    N = x_full.shape[1]
    return {
        "cpu_baseline": {"value": N / best, "unit": "points/s", "cores": torch.get_num_threads(),
                         "kind": "port",
                         "sample": f"full workload: {N}-point forward, fp32, best of 2 after 1 warm-up "
                                   f"({best:.2f} s each)"},
        "rel_l2_vs_cpu_oracle": O.rel_l2(gpu_out.cpu(), ref),
    }


if __name__ == "__main__":
    main()
