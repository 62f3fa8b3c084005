"""Worker for tests/test_ddp_gpu.py::test_two_ranks_match_single_process: TWO processes on the one GPU of the box, a gloo
communicator (RCCL refuses two ranks on one device; gloo moves CUDA tensors through the host), each rank running the product's
Stage-1 step with the bucketed, overlapped all-reduce on its half of the batch.  Rank 0 also runs the same step alone on
the full batch first: the post-all-reduce gradients, the updated weights and the dual scalar must agree to float32
summation-order tolerance (test pyramid iv)."""
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

NAME = "micro_pruned"


def step(rank, world, distributed):
    from uvc_amd.ddp import DistributedDataParallel
    run = Stage1Run(NAME, precision="fp32")
    gold = load_golden(NAME)
    if distributed:
        ddp = DistributedDataParallel(run.model, num_buckets=2, dual_scalar=run.minimax.z)
        assert ddp.world == world and ddp.reducer.avg == (dist.get_backend() == "nccl")
        run.trainer.ddp = ddp
    r = run.r
    x_all, y_all = SC.make_inputs(r)
    md, e1, e2 = split_draws(r, gold, 0, run.cfg.depth)
    run.inject_draws(md, e1, e2)                       # identical noise on every rank, as in the reference (same seed)
    x, y = torch.from_numpy(x_all[0]).cuda(), torch.from_numpy(y_all[0]).cuda()
    if distributed:
        n = x.shape[0] // world
        x, y = x[rank * n:(rank + 1) * n].contiguous(), y[rank * n:(rank + 1) * n].contiguous()
    out = run.step(x, y)
    torch.cuda.synchronize()
    n_tot = run.model._off.n_total
    return run.model._flat_grad[:n_tot].clone(), float(run.minimax.z), run.model._flat.clone(), float(out["gnorm"])


def step_t2t(rank, world, distributed):
    """The same on T2T-ViT (micro): the tokens-to-token gradients sit behind the engine's layout and ride in the tail bucket."""
    import t2t_scenarios as TS
    from test_t2t_stage1_gpu import build
    from uvc_amd.ddp import DistributedDataParallel
    r, cfg, S, tr = build("t2t_micro_train", "fp32")
    if distributed:
        tr.ddp = DistributedDataParallel(tr.model, num_buckets=2, dual_scalar=tr.minimax.z)
        assert tr.ddp.world == world
    x_all, y_all = TS.stage1_inputs(r)
    md, e1, e2 = TS.stage1_draws(r, cfg.depth)[0]
    gd = torch.stack([torch.from_numpy(d) for d in md]).cuda()
    tr.model.exp_source = lambda shape, t=gd: t
    q = [torch.from_numpy(e1).cuda(), torch.from_numpy(e2).cuda()]
    tr.minimax.exp_source = lambda shape, q=q: q.pop(0)
    x, y = torch.from_numpy(x_all[0]).cuda(), torch.from_numpy(y_all[0]).cuda()
    if distributed:
        n = x.shape[0] // world
        x, y = x[rank * n:(rank + 1) * n].contiguous(), y[rank * n:(rank + 1) * n].contiguous()
    out = tr.step(x, y, zero_grad=False)
    torch.cuda.synchronize()
    return tr.model._flat_grad[:tr.model.n_flat].clone(), float(tr.minimax.z), tr.model._flat.clone(), float(out["gnorm"])


def main():
    global step
    if os.environ.get("UVC_DDP_MODEL") == "t2t":
        step = step_t2t
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("UVC_DDP_BACKEND", "gloo")           # "nccl" = RCCL, one rank per GPU (needs >= world devices)
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    ref = step(0, 1, False) if rank == 0 else None
    dist.barrier()
    g, z, p, gn = step(rank, world, True)
    # every rank holds the same reduced gradients and weights
    gs = [torch.empty_like(g) for _ in range(world)]
    dist.all_gather(gs, g)
    assert all(torch.equal(gs[0], t) for t in gs), "ranks disagree after the all-reduce"
    if rank == 0:
        g0, z0, p0, gn0 = ref
        sc = float(g0.abs().max())
        err = float((g - g0).abs().max()) / sc
        assert err < 2e-5, f"gradients: 2-rank mean vs single process rel err {err:.2e}"
        assert abs(gn - gn0) <= 1e-5 * gn0, (gn, gn0)
        assert abs(z - z0) <= 1e-6 * max(1.0, abs(z0)), (z, z0)
        perr = float((p - p0).abs().max())
        assert perr < 5e-6, f"weights after the step differ by {perr:.2e}"
        print("DDP_TWO_RANK_OK", err, perr, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
