#!/usr/bin/env python3
"""bench.py — image-pairs/s of RegressionMatcher.match() at 560->864 on N MI355X (BASELINE.json metric), plus the
HBM roofline of the local_correlation kernel and a CPU baseline timed on the same box.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

step     = one pass of the hot path (symmetric coarse 560 pass + 864 upsample pass + post-processing) over
           --pairs synthetic image pairs per GPU, inputs already resident in HBM, outputs left on the device;
           for N>1 the per-rank results are gathered to rank 0 (the path's only exchange step) inside the step.
workload = BASELINE.json configs[1]: roma_outdoor 560->864 full coarse-to-fine, batch=1 per GPU, random-init weights of the
           shipped architecture (no checkpoints offline), fp16 autocast semantics of the reference's GPU path.
roofline = local_correlation: algorithmic bytes (f0+f1+flow+out) of the 5 launches per step / their
           HIP-event durations on the launch stream, against 8 TB/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(device, dtype):
    from roma_amd.model_zoo import build_roma
    from roma_amd.synthetic import load_synthetic_weights
    model = build_roma((560, 560), upsample_preds=True, amp_dtype=dtype)
    load_synthetic_weights(model, seed=0)
    model.upsample_res = (864, 864)
    return model.to(device).eval()


def set_dtype(model, dt):
    model.encoder.amp_dtype = dt
    model.decoder.amp_dtype = dt
    for r in model.decoder.conv_refiner.values():
        r.amp_dtype = dt


def host_cores():
    """CPU threads this process may really use: cgroup quota / affinity mask, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("ROMA_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(gpu_fp32_result, budget_s=240):
    """The oracle (CPU port of the reference path, oracle/roma_oracle.py) on ONE full 560->864 pair on the box's host
    cores, in a fresh CPU-only child process (oracle/cpu_baseline.py) with a hard time budget; also the parity gate."""
    import subprocess
    import tempfile
    import numpy as np
    cores = host_cores()
    out = tempfile.mkdtemp(prefix="roma_cpu_")
    log(f"[bench] cpu_baseline: oracle 560->864, {cores} threads, budget {budget_s}s ...")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--out", out, "--threads", str(cores)], cwd=ROOT, env=env,
                       check=True, timeout=budget_s)
    except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
        log(f"[bench] cpu_baseline did not finish: {e!r}")
        return None, None
    meta = json.load(open(os.path.join(out, "cpu_baseline.json")))
    res = {"value": 1.0 / meta["seconds"], "unit": "image-pairs/s", "cores": meta["threads"], "kind": "port",
           "sample": f"1 synthetic pair, full 560->864 symmetric match, oracle/roma_oracle.py fp32, {meta['seconds']:.1f} s"}
    parity = None
    if gpu_fp32_result is not None:
        gw, gc = gpu_fp32_result
        dw = (gw.cpu()[0] - torch.from_numpy(np.load(os.path.join(out, "warp.npy")))[0]).abs()
        dc = (gc.cpu()[0] - torch.from_numpy(np.load(os.path.join(out, "certainty.npy")))[0]).abs()
        parity = {"mode": "fp32", "warp_max_abs": float(dw.max()), "certainty_max_abs": float(dc.max()),
                  "warp_frac_gt_1e-3": float((dw > 1e-3).float().mean()), "tolerance": 1e-3}
    return res, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1, help="image pairs per GPU per step (weak scaling)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no ROCm device is visible (there is no CPU fallback path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    if args.gpus != world and rank == 0:
        log(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}")

    from roma_amd import ops
    from roma_amd.dist import gather_results
    from roma_amd.synthetic import synthetic_pair
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    do_cpu = (rank == 0 and world == 1 and not args.no_cpu)
    torch.set_grad_enabled(False)
    t0 = time.time()
    torch.set_num_threads(host_cores())
    model = build_model(device, dtype)
    log(f"[bench] rank {rank}: model built in {time.time()-t0:.1f}s")

    P = args.pairs
    first = rank * P
    pairs = [synthetic_pair(first + i) for i in range(P)]
    A_lo, B_lo, A_hi, B_hi = (torch.cat([p[j] for p in pairs]).to(device) for j in range(4))

    def step():
        warp, cert = model.match_tensors(A_lo, B_lo, A_hi, B_hi)
        if world > 1:
            warp, cert = gather_results(warp, cert, world * P, dst=0)
        return warp, cert

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.TIMER.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.TIMER.stop()
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        lc = ops.TIMER.summary().get("local_corr")
        roof = None
        if lc:
            a = lc["bytes"] / lc["seconds"]
            roof = {"bound": "hbm", "kernel": "local_corr_kernel", "achieved": a / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": a / HBM_PEAK, "traffic": None, "launches": lc["launches"],
                    "avg_launch_us": lc["seconds"] / lc["launches"] * 1e6, "algorithmic_bytes_per_launch": lc["bytes"] / lc["launches"],
                    "per_shape": {k: {"us": v[2] / v[0] * 1e6, "GB/s": v[1] / v[2] / 1e9} for k, v in sorted(lc["by_tag"].items())}}
            pmc = os.path.join(ROOT, "profiles", "local_corr_traffic.json")
            if os.path.exists(pmc):          # HBM bytes per launch from a separate rocprofv3 --pmc pass (see profiles/README.md)
                roof["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
        cpu = parity = None
        if do_cpu:
            set_dtype(model, torch.float32)
            g32 = model.match_tensors(A_lo[:1], B_lo[:1], A_hi[:1], B_hi[:1])
            torch.cuda.synchronize()
            set_dtype(model, dtype)
            cpu, parity = cpu_baseline(g32)
        total_pairs = world * P * args.steps
        line = {
            "metric": "image-pairs/sec at 560->864", "value": total_pairs / elapsed, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "roma_outdoor 560->864 full coarse-to-fine symmetric match, random-init weights",
                       "pairs_per_gpu_per_step": P, "global_pairs_per_step": world * P, "coarse_res": 560, "upsample_res": 864,
                       "parallelism": f"pair-sharded x{world}, gather of (warp, certainty) to rank 0" if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
