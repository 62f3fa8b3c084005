"""Stage-1 driver pieces (mirror of ``UVC/joint_train.py``): layer wiring, mask buffers,
checkpoint writer.  The argparse surface and the training loop live in uvc_amd/cli.py."""
from __future__ import annotations

import os

import torch


def get_uvc_layers(model, args=None):
    """joint_train.py:530-564: W1 = attn.proj, W2 = mlp.fc1, W3 = mlp.fc2 per block, matched by
    module name in registration order; s_dict[module] = [layer, column], r_dict[module] = layer."""
    layer_names = {None: None}
    uvc_layers = {"W1": [], "W2": [], "W3": []}
    for name, m in model.named_modules():
        if not hasattr(m, "in_features"):
            continue
        if "attn.proj" in name:
            uvc_layers["W1"].append(m)
        elif "mlp.fc2" in name:
            uvc_layers["W3"].append(m)
        elif "mlp.fc1" in name:
            uvc_layers["W2"].append(m)
        else:
            continue
        layer_names[m] = name
        m.uvc_s = 0
    uvc_layers_dict = {"s_dict": {}, "r_dict": {}}
    for i, m in enumerate(uvc_layers["W1"]):
        uvc_layers_dict["s_dict"][m] = [i, 0]
        uvc_layers_dict["r_dict"][m] = i
    for i, m in enumerate(uvc_layers["W3"]):
        uvc_layers_dict["s_dict"][m] = [i, 1]
    return layer_names, uvc_layers, uvc_layers_dict


def register_masks(model):
    """joint_train.py:169-171: every module with a ``weight`` gets a ``mask`` buffer of ones."""
    for _, m in model.named_modules():
        if hasattr(m, "weight") and not hasattr(m, "mask"):
            m.register_buffer("mask", torch.ones_like(m.weight))


def count_mask(model):
    """joint_train.py:182-188."""
    total = 0
    for _, m in model.named_modules():
        if hasattr(m, "mask"):
            total = total + m.mask.sum()
    return total / 1e6


def save_model(args, model, minimax_model, global_step, barrier=True):
    """joint_train.py:107-119: the checkpoint is the bare state_dict (incl. mask buffers); s, r, y, p, z are not persisted
    (SURVEY.md Q10).  The reference lets EVERY rank torch.save the same path, which can interleave writes under torchrun;
    here rank 0 writes to a temporary file and renames it (atomic), and the other ranks wait at a barrier so the file is
    complete when any rank returns -- same file, same content (replicas hold identical weights).  ``barrier=False`` for
    callers that save from rank 0 only (Stage-2's save-best, post_train.py:393-397)."""
    model_to_save = model.module if hasattr(model, "module") else model
    path = os.path.join(args.output_dir, args.name, f"{args.model_type}_{global_step}.pth.tar")
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank() if dist_on else 0
    if rank == 0 or not barrier:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp"
        torch.save(model_to_save.state_dict(), tmp)
        os.replace(tmp, path)
    if dist_on and barrier:
        torch.distributed.barrier()
    if args.local_rank in [-1, 0]:
        print("Saved model checkpoint to [DIR: %s]", args.output_dir)
    return path
