"""Learning-rate control for the MI355X PPO agent.

The agent keeps the learning rate on the device: the KL-band rule below is evaluated inside the
Adam kernel (csrc/optim.hip) from the all-reduced minibatch KL, in the same double-precision
arithmetic as the host functions here, so the trajectory is bit-identical to a host-side schedule
without the reference's per-minibatch `.item()` (rl_games/common/a2c_common.py:1557-1563).

Host API: scheduler objects with the constructor arguments and the
`update(current_lr, entropy_coef, epoch, frames, kl_dist) -> (lr, entropy_coef)` call of
rl_games/common/schedulers.py:1-58, so configs (`lr_schedule: adaptive | linear | None`) and user code
that drives a scheduler by hand keep working.  The rules themselves are two pure functions;
`device_rule()` is what the optimiser kernel consumes.
"""


def kl_band_step(lr, kl, threshold, factor, floor, ceiling):
    """Shrink lr by `factor` when kl is above twice the threshold, grow it when below half
    (the shrink is tested first and the grow second, on the ORIGINAL lr, like schedulers.py:27-33)."""
    too_far = kl > 2.0 * threshold
    too_close = kl < 0.5 * threshold
    out = max(lr / factor, floor) if too_far else lr
    return min(lr * factor, ceiling) if too_close else out


def linear_decay(start, end, done, total):
    """Value of a linear ramp from `start` (done = 0) to `end` (done >= total)."""
    remaining = max(0, total - done) / total
    return end + (start - end) * remaining


class RLScheduler:
    """Common interface; `device_rule()` is None for schedules that never change lr per minibatch."""

    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist=None, **kwargs):
        raise NotImplementedError

    def device_rule(self):
        return None


class IdentityScheduler(RLScheduler):
    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist=None, **kwargs):
        return current_lr, entropy_coef


class AdaptiveScheduler(RLScheduler):
    def __init__(self, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2, lr_multiplier=1.5):
        self.kl_threshold, self.lr_multiplier = kl_threshold, lr_multiplier
        self.min_lr, self.max_lr = min_lr, max_lr

    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        return kl_band_step(current_lr, kl_dist, self.kl_threshold, self.lr_multiplier, self.min_lr,
                            self.max_lr), entropy_coef

    def device_rule(self):
        return dict(kl_threshold=self.kl_threshold, min_lr=self.min_lr, max_lr=self.max_lr,
                    lr_multiplier=self.lr_multiplier)


class LinearScheduler(RLScheduler):
    def __init__(self, start_lr, min_lr=1e-6, max_steps=1000000, use_epochs=True, apply_to_entropy=False,
                 **kwargs):
        self.start_lr, self.min_lr, self.max_steps = start_lr, min_lr, max_steps
        self.use_epochs, self.apply_to_entropy = use_epochs, apply_to_entropy
        if apply_to_entropy:
            self.start_entropy_coef = kwargs.pop('start_entropy_coef', 0.01)
            self.min_entropy_coef = kwargs.pop('min_entropy_coef', 0.0001)

    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist=None, **kwargs):
        done = epoch if self.use_epochs else frames
        lr = linear_decay(self.start_lr, self.min_lr, done, self.max_steps)
        if self.apply_to_entropy:
            entropy_coef = linear_decay(self.start_entropy_coef, self.min_entropy_coef, done, self.max_steps)
        return lr, entropy_coef
