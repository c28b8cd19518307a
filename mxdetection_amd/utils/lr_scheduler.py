"""Learning-rate schedule (MXNet-lineage role: `WarmupMultiFactorScheduler` handed to
mx.optimizer SGD). A pure function of the 0-based iteration index, so every rank computes the same value without
communication; the value reaches the captured update kernels through a device scalar (Detector.replay(lr=...))."""


class WarmupMultiFactorScheduler:
    def __init__(self, base_lr, steps=(), factor=0.1, warmup_steps=0, warmup_lr=0.0, warmup_mode="linear"):
        steps = [int(s) for s in steps]
        if any(b <= a for a, b in zip(steps, steps[1:])):
            raise ValueError("lr steps must be strictly increasing, got %r" % (steps,))
        if steps and steps[0] < 1:
            raise ValueError("lr steps must be >= 1, got %r" % (steps,))
        if warmup_mode not in ("linear", "constant"):
            raise ValueError("warmup_mode must be 'linear' or 'constant', got %r" % (warmup_mode,))
        self.base_lr, self.steps, self.factor = float(base_lr), steps, float(factor)
        self.warmup_steps, self.warmup_lr, self.warmup_mode = int(warmup_steps), float(warmup_lr), warmup_mode

    def __call__(self, it):
        """Learning rate of iteration `it` (0-based)."""
        if it < self.warmup_steps:
            if self.warmup_mode == "constant":
                return self.warmup_lr
            return self.warmup_lr + (self.base_lr - self.warmup_lr) * (float(it) / float(self.warmup_steps))
        lr = self.base_lr
        for s in self.steps:
            if it >= s:
                lr *= self.factor
        return lr


def scaled_lr(lr_per_16, global_batch):
    """Linear scaling rule: the lineage quotes lr 0.02 for 8 GPUs x 2 images."""
    return lr_per_16 * global_batch / 16.0


def epoch_steps(lr_step_epochs, iters_per_epoch, begin_epoch=0):
    """Epoch boundaries -> iteration indices, dropping boundaries already behind a resumed run."""
    return [int((e - begin_epoch) * iters_per_epoch) for e in lr_step_epochs if e > begin_epoch]
