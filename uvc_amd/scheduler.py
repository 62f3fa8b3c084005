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


# ----------------------------------------------------------------------------------------------------
# Stage-2 (post_train.py:38,301): timm.scheduler.create_scheduler -> CosineLRScheduler, stepped once per epoch
# (`scheduler.step(epoch)` :339).  timm (0.3.2, Baseline_pruning/requirements.txt:3) is not in the image: this is a
# restatement of its published cosine schedule for the arguments the reference passes (no LR noise, one cycle,
# warm-up counted in epochs, t_mul = 1).
def cosine_epoch_lr(t, base_lr, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0, decay_rate=1.0, cycle_limit=1):
    if t < warmup_t:
        return warmup_lr_init + t * (base_lr - warmup_lr_init) / warmup_t
    i = t // t_initial
    t_curr = t - t_initial * i
    gamma = decay_rate ** i
    lr_min_i, lr_max_i = lr_min * gamma, base_lr * gamma
    if cycle_limit == 0 or i < cycle_limit:
        return lr_min_i + 0.5 * (lr_max_i - lr_min_i) * (1 + math.cos(math.pi * t_curr / t_initial))
    return lr_min


class CosineLRScheduler:
    """timm.scheduler.CosineLRScheduler(optimizer, t_initial, t_mul=1., lr_min, decay_rate, warmup_lr_init, warmup_t,
    cycle_limit=1, t_in_epochs=True): `step(epoch)` writes the epoch's value into every param group; constructing it
    with a warm-up sets the groups to warmup_lr_init."""

    def __init__(self, optimizer, t_initial, t_mul=1.0, lr_min=0.0, decay_rate=1.0, warmup_t=0, warmup_lr_init=0.0,
                 warmup_prefix=False, cycle_limit=0, t_in_epochs=True, noise_range_t=None, **_):
        if t_mul != 1.0 or warmup_prefix or noise_range_t is not None or not t_in_epochs:
            raise NotImplementedError("CosineLRScheduler: only the configuration post_train.py builds (t_mul=1, no noise, epochs)")
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.decay_rate = t_initial, lr_min, decay_rate
        self.warmup_t, self.warmup_lr_init, self.cycle_limit = warmup_t, warmup_lr_init, cycle_limit
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        if warmup_t:
            self._set(self.warmup_lr_init for _ in self.base_values)

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def get_epoch_values(self, epoch):
        return [cosine_epoch_lr(epoch, b, self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init, self.decay_rate,
                                self.cycle_limit) for b in self.base_values]

    def get_cycle_length(self, cycles=0):
        cycles = cycles or self.cycle_limit
        return self.t_initial * max(1, cycles)

    def step(self, epoch, metric=None):
        self._set(self.get_epoch_values(epoch))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


def create_scheduler(args, optimizer):
    """timm.scheduler.create_scheduler for --sched cosine (post_train.py:469-493 defaults): returns
    (scheduler, num_epochs incl. cool-down)."""
    if getattr(args, "sched", "cosine") != "cosine":
        raise NotImplementedError(f"--sched {args.sched}: Stage-2 of the reference runs the cosine schedule")
    if getattr(args, "lr_noise", None) is not None:
        raise NotImplementedError("--lr-noise")
    sch = CosineLRScheduler(optimizer, t_initial=args.epochs, t_mul=1.0, lr_min=args.min_lr, decay_rate=args.decay_rate,
                            warmup_lr_init=args.warmup_lr, warmup_t=args.warmup_epochs, cycle_limit=1, t_in_epochs=True)
    return sch, sch.get_cycle_length() + args.cooldown_epochs
