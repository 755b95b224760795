"""Running normalisers backed by the HIP kernels in csrc/running_stats.hip.

`RunningMeanStd` keeps the constructor, buffers and call signature of
`rl_games.algos_torch.running_mean_std.RunningMeanStd` (rl_games/algos_torch/
running_mean_std.py:19-114): buffers `running_mean` / `running_var` (float64) and `count`
(int64, 0-dim) live in `state_dict()` under the same names, so checkpoints written by either
implementation load into the other (normaliser state is part of
`model.state_dict()`).  `forward(input, denorm=False, mask=None)` updates the statistics when
the module is in training mode (population variance, `count += input.size(0)` even when
masked, :83) and returns the clamped normalised input.

`GeneralizedMovingStats` covers the one implementation PPO can reach ('mean_std', used by
`normalize_rms_advantage`, rl_games/common/a2c_common.py:473-475).
"""
import torch
from torch import nn

from . import ops


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        super().__init__()
        self.insize = insize
        self.epsilon = epsilon
        self.norm_only = norm_only
        self.per_channel = per_channel
        if per_channel:
            raise NotImplementedError(
                'per_channel RunningMeanStd (image observations) is outside the MI355X hot path; '
                'use flat observations')
        self.axis = [0]
        in_size = insize
        self.register_buffer('running_mean', torch.zeros(in_size, dtype=torch.float64))
        self.register_buffer('running_var', torch.ones(in_size, dtype=torch.float64))
        self.register_buffer('count', torch.ones((), dtype=torch.int64))
        self._partials = None
        # fused training forward (csrc/mlp_chain.hip): per-minibatch moments of the epoch's dataset and
        # the second state buffer set the in-kernel fold publishes into (plain attributes - checkpoints
        # only ever see running_mean / running_var / count)
        self._mb_table = self._mb_scratch = None
        self._shadow = None

    # ---- in-kernel fold (rlg_mlp_chain_forward's rms_batch arguments) -----------------------------
    def precompute_minibatch_moments(self, obs, minibatch_rows):
        """Column moments of every minibatch of the epoch in two launches.  The dataset's minibatches
        are fixed row slices (rl_games/common/datasets.py:57-75) that all mini-epochs revisit, so the
        batch statistics RunningMeanStd.forward recomputes on every visit are the same numbers."""
        self._mb_table, self._mb_scratch = ops.column_moments_segments(
            obs, minibatch_rows, self._mb_table, self._mb_scratch)
        return self._mb_table

    def fold_buffers(self, mb_index):
        """(rms, rms_fold) for the fused forward of minibatch `mb_index`: the state goes back and forth
        between the registered buffers and a second set - even minibatches read the registered ones
        and publish into the other set, odd ones the reverse (a pure function of the index: captured
        graphs and the eager path agree without any host-side mirror).  `fold_sync(nmb)` follows the
        last minibatch of a mini-epoch."""
        if self._shadow is None or self._shadow[0].device != self.running_mean.device:
            self._shadow = (torch.empty_like(self.running_mean), torch.empty_like(self.running_var),
                            torch.empty_like(self.count))
        official = (self.running_mean, self.running_var, self.count)
        cur, nxt = (official, self._shadow) if mb_index % 2 == 0 else (self._shadow, official)
        return (cur[0], cur[1]), (self._mb_table[mb_index], cur[2], nxt[0], nxt[1], nxt[2])

    def fold_sync(self, num_minibatches):
        """After the last minibatch of a mini-epoch: an odd number of folds leaves the state in the
        second set - three tiny copies bring it back to the registered buffers."""
        if num_minibatches % 2:
            self.running_mean.copy_(self._shadow[0])
            self.running_var.copy_(self._shadow[1])
            self.count.copy_(self._shadow[2])

    def _cols(self):
        return self.running_mean.numel()

    def update(self, input, mask=None, selected_only=False):
        """Fold a batch into the running statistics (no output).  `selected_only` reproduces
        `self(x[valid])`: moments and count of the masked-in rows only."""
        x = input.reshape(input.shape[0], -1)
        if x.shape[1] != self._cols():
            raise ValueError(f'expected {self._cols()} features, got {x.shape[1]}')
        m = None
        if mask is not None:
            m = mask.reshape(-1).to(torch.float32)
            if not m.is_contiguous():
                m = m.contiguous()
        if not x.is_contiguous():
            x = x.contiguous()
        nb = ops.column_moments_blocks(x.shape[0], x.shape[1])
        need = nb * (2 * x.shape[1] + 1)
        if self._partials is None or self._partials.numel() < need or self._partials.device != x.device:
            self._partials = torch.empty(need, dtype=torch.float64, device=x.device)
        part = self._partials[:need]
        ops.column_moments(x, m, part)
        mode = 0 if m is None else (2 if selected_only else 1)
        ops.rms_update(part, nb, x.shape[1], x.shape[0], mode, self.running_mean, self.running_var,
                       self.count)

    def forward(self, input, denorm: bool = False, mask=None, out=None):
        if input.dtype != torch.float32:
            input = input.float()
        if not input.is_contiguous():
            input = input.contiguous()
        if self.training:
            self.update(input, mask)
        mode = 1 if denorm else (2 if self.norm_only else 0)
        return ops.rms_apply(input, self.running_mean, self.running_var, self.epsilon, mode, out=out)


class GeneralizedMovingStats(nn.Module):
    """EMA statistics, `impl='mean_std'` only (moving_mean_std.py:24-28,:57-61,:119-122)."""

    def __init__(self, insize, impl='mean_std', decay=0.99, max=1e5, eps=0.0, perclo=0.05, perchi=0.95):
        super().__init__()
        if impl != 'mean_std':
            raise NotImplementedError(f"GeneralizedMovingStats impl '{impl}' is not on the PPO hot path")
        self.impl = impl
        self.decay = decay
        self.max = max
        self.eps = eps
        self.register_buffer('step', torch.ones(1, dtype=torch.int32))
        self.register_buffer('mean', torch.zeros(insize, dtype=torch.float32))
        self.register_buffer('sqrs', torch.zeros(insize, dtype=torch.float32))

    def kernel_state(self):
        return {'mean': self.mean, 'sqrs': self.sqrs, 'step': self.step, 'decay': self.decay,
                'max': self.max, 'eps': self.eps}

    def get_mean_std(self):
        var = self.sqrs - self.mean.pow(2)
        return self.mean, torch.sqrt(torch.clamp_min(var, 1 / self.max ** 2) + self.eps)

    def forward(self, input, mask=None, denorm=False):
        """The module call of the reference (moving_mean_std.py:102-150, 'mean_std'), as torch ops: what the agent uses
        where the fused prepare kernels do not apply (value_size > 1, rl_games_amd/torch_fallback.py); the BASELINE
        configurations update and apply these statistics inside rlg_prepare_finalize / rlg_prepare_apply
        (kernel_state()).  Training mode: one EMA update from the valid rows (none valid: no update), every update
        weighs the same whatever its row count; output clamp((x - mean) / std, -5, 5)."""
        if self.training:
            x = input
            if mask is not None:
                valid = mask.reshape(-1) > 0
                x = input[valid] if bool(valid.any()) else None
            if x is not None:
                self.step += 1
                keep = self.decay
                self.mean.mul_(keep).add_((1 - keep) * x.mean(dim=0))
                self.sqrs.mul_(keep).add_((1 - keep) * (x * x).mean(dim=0))
        mean, std = self.get_mean_std()
        if denorm:
            return input * std + mean
        return ((input - mean) / std).clamp(-5.0, 5.0)

