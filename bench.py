#!/usr/bin/env python3
"""bench.py — image-pairs/s of RegressionMatcher.match() at 560->864 on N MI355X (BASELINE.json metric), plus the
HBM roofline of the local_correlation kernel, a CPU baseline timed on the same box and the parity of what is timed.

    python bench.py --gpus N --steps K --warmup W [--workload outdoor|coarse|indoor_sample|tiny] [--pairs P]

N > 1: when WORLD_SIZE is not set, bench.py launches its own ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py ...`, one rank per GPU, as a child process) and relays rank 0's JSON line; under an external torchrun it is a
rank.  N ranks need N devices; ROMA_BENCH_REHEARSAL=1 (with ROMA_BENCH_BACKEND=gloo) allows ranks to share a device for a
dry run and tags the line `"rehearsal": true` with n_gpus = the number of DISTINCT devices.

step     = one pass of the hot path over --pairs synthetic image pairs per GPU, inputs already resident in HBM, outputs
           left on the device; for N>1 the per-rank results are gathered to rank 0 (the path's only exchange step) inside
           the step.
workloads (BASELINE.json configs):
  outdoor       configs[1] (default; configs[2] = the same with --pairs 8 --gpus 8): roma_outdoor 560->864 full
                coarse-to-fine symmetric match, fp16 autocast semantics of the reference's GPU path
  coarse        configs[0] on the GPU: the 560 x 560 coarse pass alone (upsample_preds = False; the reference's CPU-runnable case)
  indoor_sample configs[3]: roma_indoor (same architecture) 560->864, 8 pairs per step, + sample(num=10000) per pair
  tiny          configs[4]: tiny_roma_v1 (XFeat-topology backbone) on 256 pairs of 480x640 per step
roofline = local_correlation: algorithmic bytes (f0+f1+flow+out) of its launches in the timed region / their
           HIP-event durations on the launch stream, against 8 TB/s (tiny: the fused corr+soft-argmax MFMA kernel, fp32 MFMA peak).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK = 157.3e12  # same guide: fp32 MFMA = fp32 vector peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="outdoor", choices=["outdoor", "coarse", "indoor_sample", "tiny"])
    ap.add_argument("--pairs", type=int, default=0, help="image pairs per GPU per step (weak scaling); 0 = the workload's default (1 / 8 / 256)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--graph", action="store_true", help="replay ONE captured hipGraph per step instead of ~900 eager launches (measured: "
                    "+0.4 %% — the step is GPU-bound, not launch-bound; the roofline events then come from an eager pass after the timed region)")
    ap.add_argument("--wire", default="fp32", choices=["fp32", "fp16"], help="dtype of (warp, certainty) on the links of the N>1 gather")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity legs")
    ap.add_argument("--no-microbench", action="store_true", help="skip the standalone local_corr launches (for --pmc passes over the pipeline only)")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """--gpus N without a torchrun environment: start the N ranks as a CHILD process (never exec: this process may not
    touch the GPU before, and the parent stays a plain relay), pass rank 0's stdout through, return the child's exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


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
WEIGHTS_NOTE = ("roma_amd.synthetic.load_synthetic_weights(seed=0): N(0, 1/fan_in) weights, non-trivial BatchNorm statistics, to_out x6, "
                "pos_conv x4 (no checkpoints offline; SURVEY §8(d) suggests torch default init — either way random weights give "
                "spatially INCOHERENT flow, which is the slow regime of local_corr)")
INPUTS_NOTE = ("roma_amd.synthetic.synthetic_pair(i): smooth random texture + known similarity warp, generator seed 1234+i, "
               "ImageNet-normalised statistics (SURVEY §8(d) prescribes N(0,1) tensors with the same seeds)")


def build_model(device, dtype):
    import torch  # noqa: F401
    from roma_amd.model_zoo import build_roma
    from roma_amd.synthetic import load_synthetic_weights
    model = build_roma((560, 560), upsample_preds=True, amp_dtype=dtype)
    load_synthetic_weights(model, seed=0)
    model.upsample_res = (864, 864)
    return model.to(device).eval()


def run_oracle(images, budget_s=300):
    """The oracle (CPU port of the reference path, oracle/roma_oracle.py) on ONE full 560->864 pair on the box's host cores,
    in a fresh CPU-only child process (oracle/cpu_baseline.py) with a hard time budget.  Returns (cpu_baseline, warp, cert)."""
    import tempfile
    import numpy as np
    import torch
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
        return None, None, None
    meta = json.load(open(os.path.join(out, "cpu_baseline.json")))
    what = "the sacre_coeur photograph pair (tests/golden/assets)" if images else "1 synthetic pair"
    res = {"value": 1.0 / meta["seconds"], "unit": "image-pairs/s", "cores": meta["threads"], "kind": "port",
           "sample": f"{what}, full 560->864 symmetric match, oracle/roma_oracle.py fp32, {meta['seconds']:.1f} s"}
    return res, torch.from_numpy(np.load(os.path.join(out, "warp.npy"))), torch.from_numpy(np.load(os.path.join(out, "certainty.npy")))


def diff_stats(gw, gc, rw, rc):
    dw = (gw.float().cpu()[0] - rw[0]).abs()
    dc = (gc.float().cpu()[0] - rc[0]).abs()
    return {"warp_max_abs": float(dw.max()), "warp_median_abs": float(dw.median()), "warp_frac_gt_1e-3": float((dw > 1e-3).float().mean()),
            "certainty_max_abs": float(dc.max()), "certainty_median_abs": float(dc.median()),
            "certainty_frac_gt_1e-3": float((dc > 1e-3).float().mean())}


def parity_legs(device, dtype, pin, images):
    """Parity of the product path against the CPU oracle on the same inputs / weights (bar: 1e-3 max-abs, BASELINE.json):
      parity      — fp32 mode through the PRODUCT GP kernels (fp32-MFMA CosKernel + blocked Cholesky) and, beside it, with the
                    GP in torch fp64 (which removes this side's share of the ill-conditioned solve's rounding);
      parity_fp16 — the mode that is TIMED (fp16: the reference's GPU autocast semantics) against the same oracle run, plus
                    the number of scale-16 arg-max flips against the fp32 mode (cls_to_flow_refine's 4096-way arg-max is the
                    path's one discontinuity)."""
    import torch
    cpu, rw, rc = run_oracle(images)
    if cpu is None:
        return None, None, None
    res = {}
    arg = {}
    for name, dt, gp in (("fp32", torch.float32, "fp32"), ("fp32_gp64", torch.float32, "fp64"), ("timed", dtype, "fp32")):
        m = build_model(device, dt)
        m.decoder.gp_precision = gp
        m.decoder.record = {}
        w, c = m.match_tensors(*pin)
        torch.cuda.synchronize()
        res[name] = diff_stats(w, c, rw, rc)
        arg[name] = m.decoder.record.get("argmax16")
        del m
        torch.cuda.empty_cache()
    tol = 1e-3
    parity = dict(res["fp32"], mode="fp32", gp="fp32-MFMA CosKernel (fp64 chunk sums / epilogue) + fused blocked Cholesky solve (product kernels)", tolerance=tol,
                  inputs="sacre_coeur_A/B.jpg" if images else "synthetic pair 0", weights="synthetic seed 0",
                  with_fp64_gp=res["fp32_gp64"])
    flips = int((arg["timed"] != arg["fp32"]).sum()) if arg["timed"] is not None and arg["fp32"] is not None else None
    p16 = dict(res["timed"], mode={torch.float16: "fp16", torch.bfloat16: "bf16", torch.float32: "fp32"}[dtype], vs="CPU oracle fp32 (the cpu_baseline run)",
               argmax16_flips_vs_fp32_mode=flips, argmax16_total=int(arg["fp32"].numel()) if arg["fp32"] is not None else None, tolerance=tol,
               note="decoder transformer: fp32 token/residual stream, 16-bit GEMMs/attention/logits (transformer/__init__.py:30-46)")
    return cpu, parity, p16


LC_SHAPES = [("L16", 512, 40, 7), ("L8", 512, 70, 3), ("L4", 256, 140, 2), ("U8", 512, 108, 3), ("U4", 256, 216, 2)]
# what roma_local_corr's AUTO policy launches for 16-bit channels-last inputs (roma_amd/csrc/local_corr.hip, launch_any): names as
# rocprofv3 shows them
LC_KERNELS = ("local_corr_rows_kernel (row-streaming; all five levels of the 560 -> 864 match: r <= 3 with C = 256 / 512, and the r = 7 "
              "scale-16 level, whose 40 x 40 map fits the wide-box instantiation)")


def known_warp_flow(scale_unused, b, h, w, device, upsample_unused, rot_deg=8.0, scale=1.08, shift=0.04, jitter_px=0.5, gen=None):
    """The flow a converged matcher would predict on roma_amd.synthetic.synthetic_pair: B(p) = A(theta p), so the B -> A half of the
    symmetric batch is theta applied to the pixel grid and the A -> B half its inverse; + N(0, (0.5 px)^2) as in SURVEY §8(d)."""
    import math
    import torch
    c, s_ = math.cos(math.radians(rot_deg)) * scale, math.sin(math.radians(rot_deg)) * scale
    th = torch.tensor([[c, -s_, shift], [s_, c, -shift], [0.0, 0.0, 1.0]], dtype=torch.float64)
    inv = torch.linalg.inv(th)
    xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w, dtype=torch.float64)
    ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h, dtype=torch.float64)
    gx, gy = xs[None, :].expand(h, w), ys[:, None].expand(h, w)

    def apply(m):
        return torch.stack((m[0, 0] * gx + m[0, 1] * gy + m[0, 2], m[1, 0] * gx + m[1, 1] * gy + m[1, 2]))
    half = b // 2
    flow = torch.cat((apply(inv)[None].expand(half, 2, h, w), apply(th)[None].expand(b - half, 2, h, w))).float()
    noise = torch.randn(flow.shape, generator=gen)
    noise[:, 0] *= jitter_px * 2 / w
    noise[:, 1] *= jitter_px * 2 / h
    return (flow + noise).contiguous().to(device)


def lc_roofline(lc, traffic=None, tnote=None, flow_note=""):
    from roma_amd import ops
    a = lc["bytes"] / lc["seconds"]
    return {"bound": "hbm", "kernel": LC_KERNELS, "achieved": a / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": a / HBM_PEAK, "traffic": traffic, "traffic_note": tnote, "launches": lc["launches"],
            "event_bracket_overhead_us": ops.TIMER.bracket_overhead_s * 1e6,
            "avg_launch_us": lc["seconds"] / lc["launches"] * 1e6, "algorithmic_bytes_per_launch": lc["bytes"] / lc["launches"],
            "flow": flow_note,
            "per_shape": {k: {"us": v[2] / v[0] * 1e6, "GB/s": v[1] / v[2] / 1e9} for k, v in sorted(lc["by_tag"].items())}}


def local_corr_microbench(device, dtype, pairs, iters=20):
    """SURVEY §8(d) Metric 2: the same kernel on the five 560->864 call shapes with the prescribed COHERENT flow
    (identity grid o affine(rot 10 deg, scale 1.1, shift 0.05) + N(0, (0.5 px)^2)) and with U(-1.2,1.2) (adversarial
    floor).  A trained matcher produces coherent flow; random-init weights (what the timed pipeline runs) do not."""
    import math
    import torch
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
    out["note"] = ("standalone launches of roma_local_corr on the 5 call shapes (B = 2 x pairs per launch), same process, after the timed "
                   "region; kernel variant AUTO: " + LC_KERNELS)
    return out


LOCAL_CORR_SOURCES = ("local_corr.hip", "local_corr_t8.hip", "local_corr_rows.hip", "lc_variants.h", "lc_device.h", "common.h")


def kernel_source_sha1():
    """One digest over every source file of roma_local_corr (dispatch + the kernels + their shared device headers, common.h included)."""
    h = hashlib.sha1()
    for f in LOCAL_CORR_SOURCES:
        h.update(open(os.path.join(ROOT, "roma_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def measured_traffic():
    """HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/local_corr_traffic.json, tools/pmc_traffic.py).
    Only trusted when it was collected for the local_corr sources this run executes (sha1 stamp); else null + a note."""
    pmc = os.path.join(ROOT, "profiles", "local_corr_traffic.json")
    if not os.path.exists(pmc):
        return None, "no profiles/local_corr_traffic.json"
    d = json.load(open(pmc))
    if d.get("kernel_source_sha1") != kernel_source_sha1():
        return None, "profiles/local_corr_traffic.json was collected for different local_corr sources (sha1 mismatch): stale, not reported"
    return d.get("hbm_bytes_per_launch"), f"rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, {d.get('dispatches')} dispatches (profiles/local_corr_traffic.json)"


def make_workload(args, device, dtype, rank, P):
    """Returns (step_fn, config dict, model-ish handle, pinned parity inputs builder)."""
    import torch
    from roma_amd.synthetic import synthetic_pair
    if args.workload in ("outdoor", "coarse", "indoor_sample"):
        model = build_model(device, dtype)
        first = rank * P
        pairs = [synthetic_pair(first + i) for i in range(P)]
        A_lo, B_lo, A_hi, B_hi = (torch.cat([p[j] for p in pairs]).to(device) for j in range(4))
        if args.workload == "outdoor":
            graphed = None
            if args.graph:
                try:                                              # one hipGraph per step: ~900 launches replayed by one host call
                    graphed = model.make_graphed(A_lo, B_lo, A_hi, B_hi)
                except Exception as e:                            # noqa: BLE001 — capture is an optimisation, the eager path is the same kernels
                    log(f"[bench] hipGraph capture failed ({e!r}); running eagerly")
                    torch.cuda.synchronize()

            def step():
                if graphed is not None:
                    return graphed(A_lo, B_lo, A_hi, B_hi)
                return model.match_tensors(A_lo, B_lo, A_hi, B_hi)
            step.eager = lambda: model.match_tensors(A_lo, B_lo, A_hi, B_hi)
            cfg = {"workload": "roma_outdoor 560->864 full coarse-to-fine symmetric match (BASELINE configs[1]; configs[2] = --pairs 8 --gpus 8)",
                   "launch": "one captured hipGraph per step (both HIP streams of the step inside it)" if graphed is not None else "eager launches"}
        elif args.workload == "coarse":
            model.upsample_preds = False                           # BASELINE configs[0]: the 560 x 560 coarse pass alone

            def step():
                return model.match_tensors(A_lo, B_lo)
            cfg = {"workload": "roma_outdoor 560x560 coarse-only symmetric match (BASELINE configs[0], the reference's CPU-runnable case, on the GPU)"}
        else:
            it = [rank]

            def step():
                it[0] += 1
                warp, cert = model.match_tensors(A_lo, B_lo, A_hi, B_hi)
                samples = [model.sample(warp[i], cert[i], num=10000, seed=1234 + 1000 * it[0] + i) for i in range(P)]
                return warp, cert, samples
            cfg = {"workload": "roma_indoor 560->864 symmetric match + sample(num=10000) per pair (BASELINE configs[3]); roma_indoor is the "
                               "roma_outdoor architecture with other weights (model_zoo/__init__.py:54-73)", "sample_num": 10000,
                   "sample_mode": model.sample_mode}
        cfg.update(coarse_res=560, upsample_res=None if args.workload == "coarse" else 864, weights=WEIGHTS_NOTE, inputs=INPUTS_NOTE)
        return step, cfg, model, (A_lo, B_lo, A_hi, B_hi)
    # tiny
    from roma_amd.tiny import TinyRoMa, XFeatBackbone
    from roma_amd.synthetic import synthetic_state_dict
    xf = XFeatBackbone()
    xf.load_state_dict(synthetic_state_dict({k: v.shape for k, v in xf.state_dict().items()}, seed=0), strict=True)
    model = TinyRoMa(xf.eval(), freeze_xfeat=True, exact_softmax=False)
    model.load_state_dict(synthetic_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed=0))
    model = model.to(device).eval()
    xf.to(device)
    g = torch.Generator().manual_seed(1234 + rank)
    im0 = torch.rand(P, 3, 480, 640, generator=g).to(device)
    im1 = torch.rand(P, 3, 480, 640, generator=g).to(device)

    def step():
        return model.match(im0, im1, batched=True)
    cfg = {"workload": "tiny_roma_v1 (XFeat-topology backbone, synthetic weights) match() on 480x640 pairs (BASELINE configs[4])",
           "resolution": [480, 640], "inputs": "U(0,1) images, generator seed 1234+rank", "weights": "synthetic seed 0; XFeat backbone = the "
           "published topology restated in roma_amd/tiny.py (hub repository absent offline; backbone parity unpinned)"}
    return step, cfg, model, None


def stub_main(args, world, rank):
    """ROMA_BENCH_STUB=1: the launcher / rendezvous / barrier / max-over-ranks / gather / JSON plumbing of this file with a
    stand-in step on the CPU over gloo — what tests/test_dist_gloo.py runs where there is no GPU.  Never a measurement."""
    import torch
    import torch.distributed as dist
    from roma_amd.dist import gather_results
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    P = args.pairs or 1

    def step():
        warp = torch.full((P, 4, 8, 4), float(rank))
        cert = torch.full((P, 4, 8), float(rank))
        return gather_results(warp, cert, world * P, dst=0) if world > 1 else (warp, cert)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        order_ok = all(float(out[0][i * P, 0, 0, 0]) == float(i) for i in range(world))
        print(json.dumps({"metric": "image-pairs/sec at 560->864", "value": world * P * args.steps / elapsed, "unit": "image-pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "STUB (no GPU work): launcher / collective plumbing only", "gather_order_ok": order_ok},
                          "stub": True}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))          # before anything in this process touches the GPU

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ROMA_BENCH_STUB", "0") == "1":
        return stub_main(args, world, rank)
    ndev = torch.cuda.device_count()                  # does not initialise the GPU
    rehearsal = os.environ.get("ROMA_BENCH_REHEARSAL", "0") == "1"
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: no ROCm device is visible (there is no CPU fallback path)")
    if world > ndev and not rehearsal:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} device(s) visible — one rank per GPU (ROMA_BENCH_REHEARSAL=1 "
                         f"with ROMA_BENCH_BACKEND=gloo allows a dry run on shared devices)")
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no ROCm device is usable")
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

    from roma_amd import ops
    from roma_amd.dist import gather_results
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    if args.workload == "tiny":
        dtype = torch.float32                               # TinyRoMa runs fp32 (no autocast in tiny.py:330-367)
    do_cpu = (rank == 0 and world == 1 and not args.no_cpu and args.workload == "outdoor")
    torch.set_grad_enabled(False)
    t0 = time.time()
    torch.set_num_threads(host_cores())
    P = args.pairs or {"outdoor": 1, "coarse": 1, "indoor_sample": 8, "tiny": 256}[args.workload]
    run, cfg, model, resident = make_workload(args, device, dtype, rank, P)
    log(f"[bench] rank {rank}: workload {args.workload} built in {time.time()-t0:.1f}s")

    def step():
        out = run()
        if world > 1:
            warp, cert = out[0], out[1]
            if backend != "nccl":                       # rehearsal path: gloo gathers host tensors
                warp, cert = warp.cpu(), cert.cpu()
            gather_results(warp, cert, world * P, dst=0, wire_dtype=torch.float16 if args.wire == "fp16" else None)
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.TIMER.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
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
        devs = [None] * world
        dist.all_gather_object(devs, f"{os.uname().nodename}:{local_rank % ndev}")
        n_distinct = len(set(devs))
    else:
        n_distinct = 1

    eager = getattr(run, "eager", None)
    if rank == 0 and eager is not None and "hipGraph" in cfg.get("launch", ""):
        # A graph replay offers no per-kernel events: the local_corr launches are timed with HIP events over an EAGER run of the
        # same steps (same kernels, same inputs, same streams) right after the timed region.
        n_ev = max(3, min(args.steps, 10))
        eager()
        torch.cuda.synchronize()
        ops.TIMER.start()
        for _ in range(n_ev):
            eager()
        torch.cuda.synchronize()
        ops.TIMER.stop()
        cfg["roofline_events"] = f"HIP events on the launch stream over {n_ev} eager steps run right after the timed graph replays"
    if rank == 0:
        summ = ops.TIMER.summary()
        roof = None
        if args.workload == "tiny":
            tc = summ.get("tiny_corr")
            if tc:
                a = tc["bytes"] / tc["seconds"]              # "bytes" carries FLOPs for this MFMA-bound kernel
                roof = {"bound": "mfma", "kernel": "tiny_corr_kernel", "achieved": a / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                        "frac": a / MFMA_F32_PEAK, "traffic": None, "launches": tc["launches"],
                        "avg_launch_us": tc["seconds"] / tc["launches"] * 1e6, "algorithmic_flops_per_launch": tc["bytes"] / tc["launches"]}
        else:
            lc = summ.get("local_corr")
            if lc:
                traffic, tnote = measured_traffic() if args.workload == "outdoor" and P == 1 else (None, "collected for the default workload only")
                roof = lc_roofline(lc, traffic, tnote, "as produced by the timed pipeline (random-init weights => spatially incoherent flow: every "
                                   "pixel gathers its own (2r+2)^2 patch; see roofline_pipeline_coherent / roofline_microbench for coherent flow)")
        coherent = fp32_mode = None
        if args.workload == "outdoor" and world == 1 and not args.no_microbench and eager is not None:
            # (1) the SAME pipeline with the flow entering every refiner replaced by the known warp of the synthetic pair + 0.5 px noise
            # (Decoder.flow_override, diagnostic): the local_corr regime of a trained matcher, inside the pipeline.  Not part of `value`.
            gen = torch.Generator().manual_seed(0)
            cache = {}

            def override(ins, b, hs, ws, dev, upsample):
                key = (ins, b, hs, ws, upsample)
                if key not in cache:
                    cache[key] = known_warp_flow(ins, b, hs, ws, dev, upsample, gen=gen)
                return cache[key]
            model.decoder.flow_override = override
            try:
                eager()
                torch.cuda.synchronize()
                n_co = max(2, min(args.steps, 5))
                ops.TIMER.start()
                for _ in range(n_co):
                    eager()
                torch.cuda.synchronize()
                ops.TIMER.stop()
                lcc = ops.TIMER.summary().get("local_corr")
                if lcc:
                    coherent = lc_roofline(lcc, None, "not collected", "DIAGNOSTIC: same pipeline, same weights and inputs, but the flow entering "
                                           "every refiner replaced by the synthetic pair's known similarity warp + N(0, (0.5 px)^2) "
                                           f"(Decoder.flow_override); {n_co} eager steps after the timed region")
            finally:
                model.decoder.flow_override = None
            # (2) throughput of the mode that meets the 1e-3 parity bar (fp32 storage and arithmetic, product GP kernels): same step
            m32 = build_model(device, torch.float32)
            A_lo, B_lo, A_hi, B_hi = resident
            m32.match_tensors(A_lo, B_lo, A_hi, B_hi)
            torch.cuda.synchronize()
            n32 = max(2, min(args.steps, 5))
            t1 = time.perf_counter()
            for _ in range(n32):
                m32.match_tensors(A_lo, B_lo, A_hi, B_hi)
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t1) / n32
            fp32_mode = {"value": P / dt32, "unit": "image-pairs/s", "ms_per_step": dt32 * 1e3, "steps": n32,
                         "note": "fp32 mode (the one whose (warp, certainty) meet the 1e-3 max-abs bar: see `parity`), same workload and inputs"}
            del m32
            torch.cuda.empty_cache()
        micro = None
        if not args.no_microbench and args.workload == "outdoor":
            micro = local_corr_microbench(device, dtype, P)
        extra = {}
        if args.workload == "indoor_sample":
            # the same pairs without the sampling stage, for the "with and without sampling" report of SURVEY §8(d)
            A_lo, B_lo, A_hi, B_hi = resident
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_ns = max(2, args.steps // 4)
            for _ in range(n_ns):
                model.match_tensors(A_lo, B_lo, A_hi, B_hi)
            torch.cuda.synchronize()
            extra["ms_per_step_without_sample"] = (time.perf_counter() - t1) / n_ns * 1e3
        cpu = parity = parity16 = None
        if do_cpu:
            # parity + CPU baseline on a real photograph pair when the fixture images are present (the decoder's arg-max makes
            # near-tied synthetic textures flip coarse pixels between ANY two fp32 implementations)
            images = PARITY_IMAGES if all(os.path.exists(f) for f in PARITY_IMAGES) else None
            if images:
                from PIL import Image
                from roma_amd.matcher import preprocess
                ims = [Image.open(f).convert("RGB") for f in images]
                pin = [preprocess(im, (560, 560))[None].to(device) for im in ims] + [preprocess(im, (864, 864))[None].to(device) for im in ims]
            else:
                pin = [t[:1] for t in resident]
            del model, run                               # free the timed model before the three parity models are built
            torch.cuda.empty_cache()
            cpu, parity, parity16 = parity_legs(device, dtype, pin, images)
        total_pairs = world * P * args.steps
        cfg.update(pairs_per_gpu_per_step=P, global_pairs_per_step=world * P,
                   parallelism=f"pair-sharded x{world}, one ordered gather of (warp, certainty) to rank 0 per step ({args.wire} on the links)" if world > 1 else "single GPU")
        cfg.update(extra)
        line = {
            "metric": {"tiny": "image-pairs/sec at 480x640 (tiny_roma_v1)", "coarse": "image-pairs/sec at 560 (coarse only)"}.get(args.workload, "image-pairs/sec at 560->864"),
            "value": total_pairs / elapsed, "unit": "image-pairs/s",
            "n_gpus": n_distinct if rehearsal else world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {torch.float16: "fp16", torch.bfloat16: "bf16", torch.float32: "fp32"}[dtype],
            "data": "synthetic", "config": cfg,
            "roofline": roof, "roofline_pipeline_coherent": coherent, "roofline_microbench": micro, "value_fp32_mode": fp32_mode,
            "cpu_baseline": cpu, "parity": parity, "parity_fp16": parity16,
        }
        if rehearsal:
            line["rehearsal"] = True
            line["ranks"] = world
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
