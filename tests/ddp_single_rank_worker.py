"""Worker for tests/test_ddp_gpu.py: one process, a real RCCL communicator of size 1 (torch.distributed backend "nccl"),
and the product's bucketed backward with the collectives actually issued on it (the reducer is told world = 2 so it does
not short-circuit; AVG over one rank is the identity, so the result must equal the single-process gradients)."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import scenarios as SC  # noqa: E402
from helpers import load_golden, split_draws  # noqa: E402
from stage1_driver import Stage1Run  # noqa: E402


def one_step(distributed):
    from uvc_amd.ddp import DistributedDataParallel
    run = Stage1Run("micro_pruned", precision="fp32")
    gold = load_golden("micro_pruned")
    if distributed:
        ddp = DistributedDataParallel(run.model, num_buckets=2, dual_scalar=run.minimax.z)
        assert ddp.reducer.avg, "nccl backend must use ReduceOp.AVG"
        ddp.world = 2
        ddp.reducer.world = 2
        run.trainer.ddp = ddp
    r = run.r
    x_all, y_all = SC.make_inputs(r)
    md, e1, e2 = split_draws(r, gold, 0, run.cfg.depth)
    run.inject_draws(md, e1, e2)
    out = run.step(torch.from_numpy(x_all[0]).cuda(), torch.from_numpy(y_all[0]).cuda())
    torch.cuda.synchronize()
    return run.model._flat_grad[:run.model._off.n_total].clone(), float(out["loss"]), float(run.minimax.z), run.model._flat.clone()


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    # the collectives the DDP path uses, on their own
    t = torch.arange(1024, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        w = dist.all_reduce(t[128:512], op=dist.ReduceOp.AVG, async_op=True)
    w.wait()
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(t, torch.arange(1024, device="cuda", dtype=torch.float32))
    dist.broadcast(t, src=0)
    dist.barrier()
    g0, l0, z0, p0 = one_step(False)
    g1, l1, z1, p1 = one_step(True)
    assert l0 == l1 and z0 == z1, (l0, l1, z0, z1)
    assert torch.equal(g0, g1), float((g0 - g1).abs().max())
    assert torch.equal(p0, p1)
    dist.barrier()
    dist.destroy_process_group()
    print("DDP_SINGLE_RANK_OK")


if __name__ == "__main__":
    main()
