"""The N>1 path on CPU: world_size-2 gloo processes shard a batch of pairs, run a (fake, deterministic) per-pair
matcher on their contiguous shard and gather the results to rank 0 in pair order.  The real matcher needs a GPU; the
sharding / padding / ordering logic under test is exactly what bench.py runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_match(pairs):
    """(pair index i) -> warp filled with i, certainty filled with i/100: order and padding errors are visible."""
    H, W2 = 3, 4
    w = torch.stack([torch.full((H, W2, 4), float(i)) for i in pairs]) if pairs else torch.zeros(0, H, W2, 4)
    c = torch.stack([torch.full((H, W2), i / 100.0) for i in pairs]) if pairs else torch.zeros(0, H, W2)
    return w, c


def _worker(rank, world, port, npairs, q, wire=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from roma_amd.dist import match_sharded, shard_range
        lo, hi = shard_range(npairs, world, rank)
        warp, cert = match_sharded(_fake_match, list(range(npairs)), wire_dtype=wire)
        if rank == 0:
            q.put((warp.clone(), cert.clone(), (lo, hi)))
        else:
            assert warp is None and cert is None
            q.put((None, None, (lo, hi)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("npairs", [2, 5, 8])
def test_two_rank_shard_and_gather(npairs):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, npairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    warp, cert, _ = next(r for r in results if r[0] is not None)
    assert warp.shape == (npairs, 3, 4, 4) and cert.shape == (npairs, 3, 4)
    for i in range(npairs):
        assert float(warp[i].min()) == float(warp[i].max()) == float(i)
        assert abs(float(cert[i, 0, 0]) - i / 100.0) < 1e-7
    spans = sorted(r[2] for r in results)
    assert spans[0][0] == 0 and spans[-1][1] == npairs and spans[0][1] == spans[1][0]


def test_two_rank_gather_with_fp16_wire_format():
    """wire_dtype=float16: half the bytes on the links, results back in fp32 on rank 0 and exact for fp16-representable values
    (pair indices, i/100 rounded to fp16)."""
    world, npairs = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, npairs, q, torch.float16)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    warp, cert, _ = next(r for r in results if r[0] is not None)
    assert warp.dtype == torch.float32 and cert.dtype == torch.float32 and warp.shape == (npairs, 3, 4, 4)
    for i in range(npairs):
        assert float(warp[i].min()) == float(warp[i].max()) == float(i)
        assert abs(float(cert[i, 0, 0]) - i / 100.0) < 5e-4


def test_single_process_is_identity():
    from roma_amd.dist import match_sharded
    w, c = match_sharded(_fake_match, [0, 1, 2])
    assert w.shape[0] == 3 and float(w[2, 0, 0, 0]) == 2.0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start 2 ranks itself (as a child process), run the
    barrier / max-over-ranks / ordered-gather plumbing and print ONE JSON line with n_gpus = 2.  ROMA_BENCH_STUB=1 replaces
    the GPU step by a stand-in so that this runs on the CPU box over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ROMA_BENCH_STUB="1", ROMA_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["stub"] is True
    assert d["config"]["gather_order_ok"] is True and d["value"] > 0


def test_bench_refuses_mismatched_world():
    """Under an external launcher bench.py is a rank: --gpus must equal WORLD_SIZE (a silent N=1 measurement is the failure
    mode this guards against)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ROMA_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
