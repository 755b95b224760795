"""Flat parameter arena + fused clip/Adam step (csrc/optim.hip).

Replaces the optimiser side of A2CBase.trancate_gradients_and_step
(rl_games/common/a2c_common.py:493-514): the reference concatenates every `.grad` into a
fresh buffer for the all-reduce and scatters it back, then runs clip_grad_norm_ and a foreach
Adam.  Here parameters, gradients and both Adam moments are *views* of four contiguous fp32
arenas created once, so
  * the multi-GPU all-reduce runs in place on `flat_grads` (one collective, no cat/copy; one
    extra tail slot carries the minibatch KL so the separate scalar all-reduce of
    a2c_common.py:1560 and the lr broadcast of :569 are folded into it),
  * clipping + Adam + the KL-adaptive learning-rate update are two launches.

`state_dict()` / `load_state_dict()` use torch.optim.Adam's format, so the 'optimizer' entry
of a checkpoint (a2c_common.py:829, :862) is interchangeable with the reference's.
"""
import torch

from . import ops

_TAIL = 4   # extra fp32 slots at the end of the gradient arena: [0] = minibatch KL


class FlatArena:
    """Rebinds every parameter's `.data` and `.grad` to views of two contiguous fp32 buffers.
    Pure tensor plumbing (works on any device - the multi-process gloo tests use it on CPU)."""

    def __init__(self, params, layout=None):
        """`params`: logical order (= model.parameters(), the order torch.optim indexes its
        state by).  `layout`: optional physical order of the same tensors inside the arena, used
        to make e.g. the value and mu head weights adjacent so that they form one GEMM operand."""
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('no parameters')
        # bumped by everything in this package that changes parameter VALUES (optimiser steps, restores, broadcasts):
        # derived copies of the weights (ops.MlpChain's bf16 planes) are valid for one value of it
        self.weights_version = 0
        physical = list(layout) if layout is not None else self.params
        if sorted(id(p) for p in physical) != sorted(id(p) for p in self.params):
            raise ValueError('layout must be a permutation of params')
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat_params = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(self.numel + _TAIL, dtype=torch.float32, device=dev)
        off = 0
        where = {}
        for p in physical:
            n = p.numel()
            self.flat_params[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_params[off:off + n].view(p.shape)
            p.grad = self.flat_grads[off:off + n].view(p.shape)
            where[id(p)] = (off, n)
            off += n
        self.offsets = [where[id(p)] for p in self.params]
        self.grads = self.flat_grads[:self.numel]
        self.kl_slot = self.flat_grads[self.numel:self.numel + 1]

    def zero_grad(self, set_to_none=False):
        self.flat_grads.zero_()

    def weights_changed(self):
        """Call after writing parameter values by any other way than FlatAdam.step."""
        self.weights_version += 1

    def weights_token(self):
        """What derived copies of the weights (ops.MlpChain's planes / fragments) are stamped with: the counter above -
        bumped by this package's own writers, whose kernels write through raw pointers - together with the autograd
        version counters of the parameters and of the arena, which every in-place torch write bumps
        (`model.load_state_dict`, `p.copy_()`, `p.mul_()` under no_grad, writes to `flat_params`): a write from outside
        the package invalidates the copies without having to call weights_changed().  Two kinds of writes have no
        version counter to bump and need the explicit call: `p.data.copy_()`, and c10d's in-place collectives
        (`dist.broadcast` / `all_reduce` on `flat_params` leave `_version` untouched - `A2CAgent.broadcast_parameters`
        calls weights_changed() behind both of its broadcasts)."""
        return (self.weights_version, self.flat_params._version, sum(p._version for p in self.params))

    def span(self, first, last):
        """(params view, grads view) of the contiguous arena range covering parameters `first`
        .. `last` (which must be physically adjacent, in that order)."""
        return self.span_of([first, last])

    def span_of(self, params):
        """(params view, grads view) of the contiguous arena range covering `params`, which must follow one another
        physically in that order."""
        ids = {id(p): i for i, p in enumerate(self.params)}
        spans = [self.offsets[ids[id(p)]] for p in params]
        for (o0, n0), (o1, _) in zip(spans, spans[1:]):
            if o1 != o0 + n0:
                raise ValueError('parameters are not adjacent in the arena')
        start, end = spans[0][0], spans[-1][0] + spans[-1][1]
        return self.flat_params[start:end], self.flat_grads[start:end]


class FlatAdam(FlatArena):
    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, layout=None):
        super().__init__(params, layout)
        dev = self.flat_params.device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on the MI355X only (no CPU fallback)')
        self.exp_avg = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.step_count = 0                     # host mirror of the device counter below
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        # two fp64 lr slots, ping-pong by step parity (see csrc/optim.hip)
        self.lr_slots = torch.tensor([float(lr), float(lr)], dtype=torch.float64, device=dev)
        self.norm_partials = torch.zeros(ops.grad_norm_blocks(self.numel), dtype=torch.float64, device=dev)
        self.stats = torch.zeros(4, dtype=torch.float32, device=dev)
        self.param_groups = [{'params': self.params, 'lr': float(lr), 'betas': betas, 'eps': eps,
                              'weight_decay': weight_decay}]

    # ------------------------------------------------------------------ learning rate
    @property
    def cur(self):
        """Slot the NEXT step reads: step s (1-based) reads slot (s-1)&1."""
        return self.step_count & 1

    def set_lr(self, lr):
        """Host-driven lr (linear / identity schedules, restore): overwrites the live slot."""
        self.lr_slots[self.cur] = float(lr)
        self.param_groups[0]['lr'] = float(lr)

    def current_lr(self):
        """Reads the live lr back (one device->host sync; call once per epoch, not per step)."""
        lr = float(self.lr_slots[self.cur].item())
        self.param_groups[0]['lr'] = lr
        return lr

    def last_and_next_lr(self):
        """(lr used by the most recent step, lr the next step will use) - one host sync."""
        both = self.lr_slots.tolist()
        nxt, used = both[self.cur], both[self.cur ^ 1]
        if self.step_count == 0:
            used = nxt
        self.param_groups[0]['lr'] = nxt
        return used, nxt

    # ------------------------------------------------------------------ step
    def step(self, grad_scale=1.0, max_norm=None, schedule=None, kl_scale=1.0, norm_ready=None, skip_flag=None,
             pack=None):
        """grad_scale: 1/world_size after a SUM all-reduce.  max_norm: clip threshold or None.
        schedule: None or dict(kl_threshold, min_lr, max_lr, lr_multiplier) -> KL-adaptive lr
        driven by the KL in `kl_slot` (times kl_scale).  norm_ready = (partials fp64, count): the
        launch that wrote the gradients already left the per-block sums of (g * grad_scale)^2 AND advanced
        the device step counter (ops.MlpDwPlan.launch(norm=...)) - no grad_sumsq launch.  skip_flag: device
        address of the in-graph all-reduce's error word (IpcAllReduce.error_word): a step behind a failed
        collective leaves parameters, moments and learning rate untouched.  pack: an ops.MlpChain whose weights live
        in this arena - the launch also writes the chain's bf16 weight planes for the new weights (one launch instead of
        Adam + pack; csrc/mlp_chain_bx.hip adam_pack_kernel)."""
        self.step_count += 1
        self.weights_version += 1
        if norm_ready is None:
            # it also advances the device step counter the Adam kernel reads
            ops.grad_sumsq(self.grads, grad_scale, self.norm_partials, self.step_counter)
            partials = self.norm_partials if max_norm is not None else None
        else:
            partials = norm_ready[0][:norm_ready[1]] if max_norm is not None else None
        kw = {}
        kind = 0
        if schedule is not None:
            kind = 1
            kw = dict(kl_threshold=schedule['kl_threshold'], min_lr=schedule['min_lr'],
                      max_lr=schedule['max_lr'], lr_multiplier=schedule['lr_multiplier'])
        ops.adam_step(self.flat_params, self.grads, self.exp_avg, self.exp_avg_sq, partials,
                      grad_scale, 0.0 if max_norm is None else max_norm, self.lr_slots,
                      self.step_counter, betas=self.betas, eps=self.eps,
                      weight_decay=self.weight_decay, schedule_kind=kind,
                      kl=self.kl_slot if kind else None, kl_scale=kl_scale, stats_out=self.stats,
                      skip_flag=skip_flag, pack=None if pack is None else pack.adam_pack_target(), **kw)
        if pack is not None:
            pack.mark_planes(self.weights_token())

    def step_done(self):
        """Host mirrors after a replayed graph performed the step (the captured launches advance the device counter)."""
        self.step_count += 1
        self.weights_version += 1

    # ------------------------------------------------------------------ checkpoint format
    def state_dict(self):
        state = {}
        # like torch.optim.Adam: no per-parameter state before the first step, and the reference's
        # defaults (`fused: None`, `foreach: None`) so that a reference agent restoring this
        # checkpoint keeps its own default (foreach) implementation
        for i, (off, n) in enumerate(self.offsets if self.step_count > 0 else ()):
            shape = self.params[i].shape
            state[i] = {'step': torch.tensor(float(self.step_count)),
                        'exp_avg': self.exp_avg[off:off + n].view(shape).clone(),
                        'exp_avg_sq': self.exp_avg_sq[off:off + n].view(shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group.update(lr=self.current_lr(), amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, decoupled_weight_decay=False,
                     params=list(range(len(self.params))))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        st = sd.get('state', {})
        steps = []
        for i, (off, n) in enumerate(self.offsets):
            if i not in st:
                continue
            self.exp_avg[off:off + n].copy_(st[i]['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st[i]['exp_avg_sq'].reshape(-1))
            steps.append(int(float(st[i]['step'])))
        if steps:
            self.step_count = max(steps)
            self.step_counter.fill_(self.step_count)
            both = self.lr_slots.tolist()
            self.lr_slots.fill_(both[0])
        groups = sd.get('param_groups') or []
        if groups:
            self.set_lr(groups[0].get('lr', self.param_groups[0]['lr']))
