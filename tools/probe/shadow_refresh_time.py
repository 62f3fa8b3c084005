import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, bench
from uvc_amd import _lib as L
from uvc_amd.stage1 import Stage1Trainer, default_args
from uvc_amd import model_distilled as MD
a = default_args(model_type="deit_tiny_patch16_224", precision="bf16", train_batch_size=8, local_rank=0)
tr = Stage1Trainer(a, device="cuda:0", distributed=False)
m = tr.model
lib = MD._bind()
def f():
    L.check(lib.uvc_vit_update_shadows(C.byref(m._cfg), L.ptr(m._flat), L.ptr(m._shadow), L.cur_stream()), "x")
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("uvc_vit_update_shadows: %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
