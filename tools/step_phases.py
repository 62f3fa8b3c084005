"""Where one UVC-train step spends its time on the MAIN stream, without a profiler: events between the phases of Stage1Trainer.step
(same calls, same order), N steps enqueued without synchronisation, mean GPU time between consecutive events.
    python tools/step_phases.py            (STEP_MODEL / STEP_BATCH as tools/host_ahead.py; TEACHER_FIRST=0: student forward enqueued before the teacher's)"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uvc_amd.stage1 import Stage1Trainer, default_args  # noqa: E402
from uvc_amd.optim import clip_grad_norm_  # noqa: E402
from uvc_amd.losses import unit_gradient  # noqa: E402
from uvc_amd.uvc_optimizer import uvc_optimizer  # noqa: E402

model = os.environ.get("STEP_MODEL", "deit_tiny_patch16_224")
batch = int(os.environ.get("STEP_BATCH", "512"))
a = default_args(model_type=model, precision="bf16", train_batch_size=batch, local_rank=0)
tr = Stage1Trainer(a, device="cuda:0", distributed=False)
bench.pruned_state(tr)
tr.begin_epoch(a.warmup_epochs + 1)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(batch, 3, a.img_size, a.img_size, device="cuda", generator=g)
y = torch.softmax(torch.randn(batch, a.num_classes, device="cuda", generator=g), -1)
NAMES = ["prefetch enqueued -> student forward done", "loss", "backward", "clip + AdamW + schedule", "uvc_optimizer + zero_grad", "(next step's start)"]


HOST = []


def step(ev):
    self = tr
    cur = torch.cuda.current_stream()
    self.noise.begin_step(self.global_step, resume_window=False)
    ev[0].record(cur)
    h0 = time.perf_counter()
    if os.environ.get("TEACHER_FIRST", "1") != "0":
        self.criterion.prefetch(x)
    h1 = time.perf_counter()
    outputs, _ = self.model(x, self.get_tau(), a.patch_ratio)
    h2 = time.perf_counter()
    if os.environ.get("TEACHER_FIRST", "1") == "0":
        self.criterion.prefetch(x)
    HOST.append((h0, h1, h2))
    ev[1].record(cur)
    loss = self.criterion(x, outputs, y)
    ev[2].record(cur)
    loss.backward(unit_gradient(loss.device))
    ev[3].record(cur)
    clip_grad_norm_(self.model, a.max_grad_norm)
    self.optimizer.step()
    self.scheduler.step()
    self.global_step += 1
    if not self.minimax.model.enable_warmup:
        self.zlr_scheduler(self.dual_opt, self.epoch, "zlr")
    ev[4].record(cur)
    self.minimax.update_gating()
    _, _, _, _, self.gating_grad_list = uvc_optimizer(
        self.optimizer, self.minimax, self.s_opt, self.r_opt, self.g_opt, self.dual_opt, a, {"global_step": self.global_step},
        [], self.flops_list, a.z_grad_clip, self.global_step, a.gating_interval, self.gating_grad_list)
    self.optimizer.zero_grad()
    ev[5].record(cur)


for _ in range(10):
    tr.step(x, y)
torch.cuda.synchronize()
N = int(os.environ.get("STEPS", "40"))
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(N)]
base = torch.cuda.Event(enable_timing=True)
base.record(torch.cuda.current_stream())
torch.cuda.synchronize()
hbase = time.perf_counter()
for i in range(N):
    step(evs[i])
hend = time.perf_counter()
torch.cuda.synchronize()
print("host: enqueue loop %.2f ms per step" % ((hend - hbase) / N * 1e3))
for i in (0, 1, 2, 5, 10, 20, N - 1):
    g0 = base.elapsed_time(evs[i][0])
    h0, h1, h2 = [(t - hbase) * 1e3 for t in HOST[i]]
    print("  step %2d: GPU reaches the step's first event at %8.2f ms; host enqueued it at %8.2f, teacher forward enqueued by %8.2f (+%.2f), student forward by %8.2f (+%.2f)" % (
        i, g0, h0, h1, h1 - h0, h2, h2 - h1))
tot = evs[5][0].elapsed_time(evs[N - 1][0]) / (N - 6)
print("%s batch %d: %.3f ms per step (events on the main stream, steps 5..%d)" % (model, batch, tot, N - 1))
for k in range(5):
    d = sum(evs[i][k].elapsed_time(evs[i][k + 1]) for i in range(5, N - 1)) / (N - 6)
    print("  %-45s %.3f ms" % (NAMES[k] if k else NAMES[0], d))
d = sum(evs[i][5].elapsed_time(evs[i + 1][0]) for i in range(5, N - 1)) / (N - 6)
print("  %-45s %.3f ms" % ("step end -> next step's first event", d))
