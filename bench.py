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


PARITY_IMAGES = [os.path.join(ROOT, "tests", "golden", "assets", f"sacre_coeur_{n}.jpg") for n in "AB"]


def cpu_baseline(gpu_fp32_result, budget_s=240, images=None):
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
        cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--out", out, "--threads", str(cores)]
        if images:
            cmd += ["--images"] + list(images)
        subprocess.run(cmd, cwd=ROOT, env=env, check=True, timeout=budget_s)
    except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
        log(f"[bench] cpu_baseline did not finish: {e!r}")
        return None, None
    meta = json.load(open(os.path.join(out, "cpu_baseline.json")))
    what = "the sacre_coeur photograph pair (tests/golden/assets)" if images else "1 synthetic pair"
    res = {"value": 1.0 / meta["seconds"], "unit": "image-pairs/s", "cores": meta["threads"], "kind": "port",
           "sample": f"{what}, full 560->864 symmetric match, oracle/roma_oracle.py fp32, {meta['seconds']:.1f} s"}
    parity = None
    if gpu_fp32_result is not None:
        gw, gc = gpu_fp32_result
        dw = (gw.cpu()[0] - torch.from_numpy(np.load(os.path.join(out, "warp.npy")))[0]).abs()
        dc = (gc.cpu()[0] - torch.from_numpy(np.load(os.path.join(out, "certainty.npy")))[0]).abs()
        parity = {"mode": "fp32", "inputs": "sacre_coeur_A/B.jpg" if images else "synthetic pair 0", "warp_max_abs": float(dw.max()), "certainty_max_abs": float(dc.max()),
                  "warp_frac_gt_1e-3": float((dw > 1e-3).float().mean()), "tolerance": 1e-3}
    return res, parity


LC_SHAPES = [("L16", 512, 40, 7), ("L8", 512, 70, 3), ("L4", 256, 140, 2), ("U8", 512, 108, 3), ("U4", 256, 216, 2)]


def local_corr_microbench(device, dtype, pairs, iters=20):
    """SURVEY §8(d) Metric 2: the same kernel on the five 560->864 call shapes with the prescribed COHERENT flow
    (identity grid o affine(rot 10 deg, scale 1.1, shift 0.05) + N(0, (0.5 px)^2)) and with U(-1.2,1.2) (adversarial
    floor).  A trained matcher produces coherent flow; random-init weights (what the timed pipeline runs) do not."""
    import math
    from roma_amd import ops
    es = 4 if dtype == torch.float32 else 2
    out = {}
    g = torch.Generator().manual_seed(0)
    gd = torch.Generator(device=device).manual_seed(0)
    # "coherent_16pairs": §8(d) asks for >= 16 pairs per launch so that the small maps are not launch / occupancy bound
    for kind in ("coherent", "adversarial", "coherent_16pairs"):
        tot_b = tot_t = 0.0
        per = {}
        B = 32 if kind == "coherent_16pairs" else 2 * pairs
        for name, C, h, r in LC_SHAPES:
            K = (2 * r + 1) ** 2
            if kind == "coherent_16pairs":          # features drawn on the device (1.5 GB per tensor at the largest shape)
                f0 = torch.randn(B, h, h, C, generator=gd, device=device, dtype=dtype).permute(0, 3, 1, 2)
                f1 = torch.randn(B, h, h, C, generator=gd, device=device, dtype=dtype).permute(0, 3, 1, 2)
            else:
                f0 = torch.randn(B, C, h, h, generator=g).to(device=device, dtype=dtype).contiguous(memory_format=torch.channels_last)
                f1 = torch.randn(B, C, h, h, generator=g).to(device=device, dtype=dtype).contiguous(memory_format=torch.channels_last)
            if kind.startswith("coherent"):
                ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h)
                gx, gy = ys[None, :].expand(h, h), ys[:, None].expand(h, h)
                c, s_ = math.cos(math.radians(10)) * 1.1, math.sin(math.radians(10)) * 1.1
                flow = torch.stack((c * gx - s_ * gy + 0.05, s_ * gx + c * gy - 0.05))[None].expand(B, 2, h, h)
                flow = flow + torch.randn(B, 2, h, h, generator=g) * (0.5 * 2 / h)
            else:
                flow = torch.rand(B, 2, h, h, generator=g) * 2.4 - 1.2
            flow = flow.contiguous().to(device)
            o = ops.nhwc_empty(B, K, h, h, dtype, device)
            for _ in range(3):
                ops.local_correlation(f0, f1, r, flow=flow, out=o)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                ops.local_correlation(f0, f1, r, flow=flow, out=o)
            e.record()
            torch.cuda.synchronize()
            t = s.elapsed_time(e) / iters * 1e-3
            nb = 2 * B * C * h * h * es + B * 2 * h * h * 4 + B * K * h * h * es
            per[name] = {"us": t * 1e6, "GB/s": nb / t / 1e9}
            tot_b += nb
            tot_t += t
        out[kind] = {"achieved": tot_b / tot_t / 1e9, "frac": tot_b / tot_t / HBM_PEAK, "unit": "GB/s", "B": B, "per_shape": per}
        del f0, f1, o, flow
    out["note"] = "standalone launches of local_corr_kernel on the 5 call shapes (B = 2 x pairs per launch), same process, after the timed region"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1, help="image pairs per GPU per step (weak scaling)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity leg")
    ap.add_argument("--no-microbench", action="store_true", help="skip the standalone local_corr launches (for --pmc passes over the pipeline only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no ROCm device is visible (there is no CPU fallback path)")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    device = torch.device("cuda", local_rank % ndev)
    import torch.distributed as dist
    backend = os.environ.get("ROMA_BENCH_BACKEND", "nccl")        # "gloo" only for rehearsing N>1 on a one-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
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
            if backend != "nccl":                       # rehearsal path: gloo gathers host tensors
                warp, cert = warp.cpu(), cert.cpu()
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
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        lc = ops.TIMER.summary().get("local_corr")
        roof = None
        if lc:
            a = lc["bytes"] / lc["seconds"]
            roof = {"bound": "hbm", "kernel": "local_corr_kernel", "achieved": a / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": a / HBM_PEAK, "traffic": None, "launches": lc["launches"],
                    "event_bracket_overhead_us": ops.TIMER.bracket_overhead_s * 1e6,
                    "avg_launch_us": lc["seconds"] / lc["launches"] * 1e6, "algorithmic_bytes_per_launch": lc["bytes"] / lc["launches"],
                    "per_shape": {k: {"us": v[2] / v[0] * 1e6, "GB/s": v[1] / v[2] / 1e9} for k, v in sorted(lc["by_tag"].items())}}
            pmc = os.path.join(ROOT, "profiles", "local_corr_traffic.json")
            if os.path.exists(pmc):          # HBM bytes per launch from a separate rocprofv3 --pmc pass (see profiles/README.md)
                roof["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
        micro = None if args.no_microbench else local_corr_microbench(device, dtype, P)
        cpu = parity = None
        if do_cpu:
            # parity gate + CPU baseline on a real photograph pair when the fixture images are present (the decoder's
            # arg-max makes near-tied synthetic textures flip coarse pixels between ANY two fp32 implementations)
            images = PARITY_IMAGES if all(os.path.exists(f) for f in PARITY_IMAGES) else None
            if images:
                from PIL import Image
                from roma_amd.matcher import preprocess
                ims = [Image.open(f).convert("RGB") for f in images]
                pin = [preprocess(im, (560, 560))[None].to(device) for im in ims] + [preprocess(im, (864, 864))[None].to(device) for im in ims]
            else:
                pin = [A_lo[:1], B_lo[:1], A_hi[:1], B_hi[:1]]
            del model                                    # its ViT weights were cast to the amp dtype in place
            torch.cuda.empty_cache()
            model32 = build_model(device, torch.float32)
            g32 = model32.match_tensors(*pin)
            torch.cuda.synchronize()
            del model32
            cpu, parity = cpu_baseline(g32, images=images)
        total_pairs = world * P * args.steps
        line = {
            "metric": "image-pairs/sec at 560->864", "value": total_pairs / elapsed, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "roma_outdoor 560->864 full coarse-to-fine symmetric match, random-init weights",
                       "pairs_per_gpu_per_step": P, "global_pairs_per_step": world * P, "coarse_res": 560, "upsample_res": 864,
                       "parallelism": f"pair-sharded x{world}, gather of (warp, certainty) to rank 0" if world > 1 else "single GPU"},
            "roofline": roof, "roofline_microbench": micro, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
