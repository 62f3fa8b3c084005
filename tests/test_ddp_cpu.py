"""CPU, world_size 2 over gloo: the bucketed mean all-reduce of the flat gradient buffer
(uvc_amd.ddp.FlatGradReducer / bucket_plan) -- every element reduced exactly once, result = mean over
ranks, dual scalar slot carried in the tail bucket."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Off:
    def __init__(self, depth, per_block, embed, head, small):
        self.blk = [[embed + l * per_block] * 12 for l in range(depth)]
        self.n_main = embed + depth * per_block + head
        self.n_total = self.n_main + small


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, depth, nb, out, front=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uvc_amd.ddp import FlatGradReducer, bucket_plan
    off = _Off(depth, per_block=1000, embed=300, head=500, small=40)
    n_extra = 4
    # front > 0: a model whose flat buffer carries extra parameters behind the engine's layout (T2T-ViT's tokens-to-token
    # module, uvc_amd/t2t_vit.py); they and the dual-scalar slot behind them ride in the tail bucket
    n_flat = off.n_total + front
    plan = bucket_plan(off, depth, n_extra, nb, n_flat)
    off.n_total = n_flat                          # the rest of the check addresses the whole buffer
    # coverage: every index of [0, n_flat + n_extra) in exactly one bucket
    cover = torch.zeros(off.n_total + n_extra, dtype=torch.int32)
    for _, ranges in plan:
        for o, n in ranges:
            cover[o:o + n] += 1
    assert bool((cover == 1).all()), "bucket plan must tile the flat buffer"
    ends = [p[0] for p in plan]
    assert ends == sorted(ends) and ends[-1] == depth + 3
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(off.n_total + n_extra, generator=g)
    flat[off.n_total] = 3.25                      # the dual scalar: identical on every rank
    mine = flat.clone()
    red = FlatGradReducer(flat, [p[1] for p in plan])
    for i in range(len(plan)):
        red.launch(i)
    red.finish()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    mean = torch.stack(gathered).mean(0)
    ok = torch.allclose(flat, mean, rtol=1e-6, atol=1e-7) and float(flat[off.n_total]) == 3.25
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_bucketed_mean_allreduce_gloo_world2():
    for depth, nb in ((12, 4), (12, 1), (2, 4), (14, 3)):
        port = _free_port()
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, depth, nb, out), nprocs=2, join=True)
        assert out[0] and out[1], (depth, nb)


def test_bucketed_allreduce_with_front_end_segment_gloo_world2():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, 14, 4, out, 7777), nprocs=2, join=True)
    assert out[0] and out[1]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py started under a launcher whose WORLD_SIZE differs from --gpus exits non-zero before touching a device (it would otherwise
    print a line whose n_gpus is not what the caller asked for); and with no launcher and more ranks asked for than devices visible."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and not r.stdout.strip()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UVC_BENCH_SHARE_DEVICE")}
    if torch.cuda.device_count() < 64:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "device(s) visible" in r.stderr and not r.stdout.strip()
