"""Host-side schedules of the Stage-1 driver (mirror of ``UVC/utils/scheduler.py``).

WarmupCosineSchedule / WarmupLinearSchedule (utils/scheduler.py:29-63) are LambdaLR subclasses
with the same constructor; PresetLRScheduler (utils/scheduler.py:69-93) keeps the reference's
behaviour of looking for a ``"zlr"`` key in the param groups -- which the dual optimiser does not
have, so z's learning rate stays at zlr_schedule_list[0].
"""
import math

from torch.optim.lr_scheduler import LambdaLR


def warmup_cosine(step, warmup_steps, t_total, cycles=0.5):
    if step < warmup_steps:
        return float(step) / float(max(1.0, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, t_total - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * progress)))


def warmup_linear(step, warmup_steps, t_total):
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


class WarmupCosineSchedule(LambdaLR):
    def __init__(self, optimizer, warmup_steps, t_total, cycles=.5, last_epoch=-1):
        self.warmup_steps, self.t_total, self.cycles = warmup_steps, t_total, cycles
        super().__init__(optimizer, lambda s: warmup_cosine(s, warmup_steps, t_total, cycles), last_epoch=last_epoch)


class WarmupLinearSchedule(LambdaLR):
    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        self.warmup_steps, self.t_total = warmup_steps, t_total
        super().__init__(optimizer, lambda s: warmup_linear(s, warmup_steps, t_total), last_epoch=last_epoch)


class PresetLRScheduler:
    """iteration -> value table applied to param_group[lr_name] (only where that key exists)."""

    def __init__(self, decay_schedule):
        self.decay_schedule = dict(decay_schedule)
        print(f"=> Using a preset learning rate schedule: {self.decay_schedule}")

    def __call__(self, optimizer, iteration, lr_name="zlr"):
        for group in optimizer.param_groups:
            if lr_name not in group:
                continue
            new = self.decay_schedule.get(iteration, group[lr_name])
            if new != group[lr_name]:
                print(f"==== [Schduler Working]{lr_name} is changed from {group[lr_name]} to {new}")
                group[lr_name] = new

    @staticmethod
    def get_lr(optimizer):
        return optimizer.param_groups[0]["lr"]
