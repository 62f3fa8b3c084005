"""A minimal module tree with the reference's UVC layer names (blocks.i.attn.proj / mlp.fc1 /
mlp.fc2, joint_train.py:530-564) for testing the UVC engine without the transformer."""
import torch
import torch.nn as nn


class _Attn(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.proj = nn.Linear(D, D)


class _Mlp(nn.Module):
    def __init__(self, D, F):
        super().__init__()
        self.fc1 = nn.Linear(D, F)
        self.fc2 = nn.Linear(F, D)


class _Block(nn.Module):
    def __init__(self, D, F):
        super().__init__()
        self.attn = _Attn(D)
        self.mlp = _Mlp(D, F)


class _PE:
    grid_size = (4, 4)


class StubModel(nn.Module):
    def __init__(self, L, D, F):
        super().__init__()
        self.blocks = nn.Sequential(*[_Block(D, F) for _ in range(L)])
        self.block_skip_gating = nn.Parameter(torch.tensor([-1.0, 1.0]).expand(L, 2).contiguous())
        self.patch_embed = _PE()
        self.eps = 0.1
        self.enable_warmup = 0
        for _, m in self.named_modules():
            if hasattr(m, "weight"):
                m.register_buffer("mask", torch.ones_like(m.weight))
