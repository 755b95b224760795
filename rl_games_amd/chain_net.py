"""A plain MLP + linear head(s) trained on the fused chain kernels, without autograd.

What autograd does between a loss kernel and the optimiser for the networks of the reference's builder
(rl_games/algos_torch/network_builder.py:224-311: `actor_mlp` / `critic_mlp` = Linear + activation pairs, then `value` /
`logits` Linear heads) - forward with the activations kept, dZ of every hidden layer with the bias-gradient column sums,
the weight gradients - as the launches of the continuous actor's hot path: the input normaliser + hidden layers + heads
as ONE forward launch, the dX / activation-backward / bias-sum chain as ONE backward launch (csrc/mlp_chain*.hip,
`ops.MlpChain`), every weight gradient as the MFMA launch of csrc/mlp_dw.hip (`ops.MlpDwPlan`).

Users: the central value network (central_value.py, one value column) and the discrete-action agent (discrete_agent.py:
[value | logits] behind a shared trunk, or one chain per trunk with `separate: True` as in ppo_cartpole.yaml).  The
continuous actor has its own engine (mlp_engine.ManualMLP: loss inside the backward launch, LSTM, HIP graphs).
"""
import torch
from torch import nn

from . import ops

_ACT_NAMES = {nn.ELU: 'elu', nn.ReLU: 'relu', nn.Tanh: 'tanh', nn.Identity: 'None'}


def arena_layout(params, head_groups):
    """Physical arena order for `FlatArena(layout=...)`: every weight matrix first (parameters() order), then each
    group of head weights (adjacent, in the given order - one GEMM operand per group), then the vectors, then each
    group's biases.  Matrix sizes of the supported shapes are multiples of 4 floats, so the matrices stay 16-byte
    aligned; the logical (optimiser-state) order is untouched.  head_groups: lists of nn.Linear."""
    head_w = [m.weight for g in head_groups for m in g]
    head_b = [m.bias for g in head_groups for m in g]
    skip = {id(p) for p in head_w + head_b}
    rest = [p for p in params if id(p) not in skip]
    return [p for p in rest if p.dim() >= 2] + head_w + [p for p in rest if p.dim() < 2] + head_b


class ChainNet:
    """trunk: nn.Sequential of Linear + activation pairs; heads: list of nn.Linear over the trunk's output whose
    weights (and biases) are adjacent in `arena`, in that order (arena_layout) - their columns side by side are the
    chain's last layer.  Raises NotImplementedError for networks outside the kernels' envelope."""

    def __init__(self, trunk, heads, arena, max_rows):
        self.linears = [m for m in trunk if isinstance(m, nn.Linear)]
        acts = [m for m in trunk if not isinstance(m, nn.Linear)]
        if not self.linears or len(acts) != len(self.linears):
            raise NotImplementedError('unexpected MLP structure')
        name = _ACT_NAMES.get(type(acts[0]))
        if name is None or any(type(a) is not type(acts[0]) for a in acts):
            raise NotImplementedError('elu / relu / tanh / identity trunks only')
        if isinstance(acts[0], nn.ELU) and acts[0].alpha != 1.0:
            raise NotImplementedError('elu alpha != 1')
        if any(l.out_features % 4 for l in self.linears):
            raise NotImplementedError('hidden widths must be multiples of 4')
        heads = list(heads)
        K = self.linears[-1].out_features
        if any(h.in_features != K for h in heads):
            raise NotImplementedError('heads must read the last hidden layer')
        self.head_cols = sum(h.out_features for h in heads)
        if len(heads) == 1:
            h = heads[0]
            self.head_w, self.head_w_grad, self.head_b, self.head_b_grad = h.weight, h.weight.grad, h.bias, h.bias.grad
        else:
            wp, wg = arena.span_of([h.weight for h in heads])
            self.head_b, self.head_b_grad = arena.span_of([h.bias for h in heads])
            self.head_w, self.head_w_grad = wp.view(self.head_cols, K), wg.view(self.head_cols, K)
        dev = self.head_w.device
        widths = [l.out_features for l in self.linears]
        layers = [(l.weight, l.bias, name) for l in self.linears] + [(self.head_w, self.head_b, 'None')]
        self.chain = ops.MlpChain(layers, dev, weights_version=arena.weights_token)
        self.Hs = [torch.empty(max_rows, w, device=dev) for w in widths]
        self.dA = [torch.empty(max_rows, w, device=dev) for w in widths]
        self.heads = torch.empty(max_rows, self.head_cols, device=dev)
        self.d_heads = torch.empty(max_rows, self.head_cols, device=dev)
        self.xn = torch.empty(max_rows, self.linears[0].in_features, device=dev)
        nb = (max_rows + 15) // 16                          # one partial row per 16-row group at most
        self.partials = [torch.empty(nb * w, dtype=torch.float64, device=dev) for w in widths]
        self._plans = {}
        self._rows = 0
        self._x = None
        self.last_dw_path = None

    @torch.no_grad()
    def forward(self, x, rms, eps):
        """x [rows, in] RAW inputs; rms = (running_mean, running_var) or None.  Returns the heads [rows, head_cols]
        and keeps what backward() reads."""
        rows = x.shape[0]
        if not x.is_contiguous():
            x = x.contiguous()
        heads = self.heads[:rows]
        xn = self.xn[:rows] if rms is not None else None
        self.chain.forward(x, heads, act_out=[h[:rows] for h in self.Hs], rms=rms, eps=eps, xn_out=xn)
        self._rows, self._x = rows, (xn if rms is not None else x)
        return heads

    @torch.no_grad()
    def backward(self):
        """d loss / d heads in self.d_heads[:rows] -> every gradient of the network in the arena."""
        rows, L = self._rows, len(self.linears)
        d_heads = self.d_heads[:rows]
        acts = [h[:rows] for h in self.Hs]
        dzs = [d[:rows] for d in self.dA]
        nblk = self.chain.num_blocks(rows, 1)
        parts = [p[:nblk * l.out_features] for p, l in zip(self.partials, self.linears)]
        self.chain.backward(d_heads, acts, dzs, parts)
        jobs, colsums = [], []
        if self.head_cols == 1:
            # one output column - a weighted column sum of the last activations, and the sum of d heads
            torch.mv(acts[-1].t(), d_heads.view(-1), out=self.head_w_grad.view(-1))
        else:
            jobs.append((d_heads, acts[-1], self.head_w_grad))
        torch.sum(d_heads, dim=0, out=self.head_b_grad)
        for l in range(L - 1, -1, -1):
            lin = self.linears[l]
            jobs.append((dzs[l], acts[l - 1] if l > 0 else self._x, lin.weight.grad))
            colsums.append((parts[l], nblk, lin.out_features, lin.bias.grad))
        fast = [j for j in jobs if j[2].shape[1] % 4 == 0 and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in j)]
        slow = [j for j in jobs if not any(j is f for f in fast)]
        plan = None
        if fast:
            key = (rows,) + tuple(tuple(g.shape) for _, _, g in fast)
            plan = self._plans.get(key)
            if plan is None:
                try:
                    plan = ops.MlpDwPlan([tuple(g.shape) for _, _, g in fast], rows, fast[0][2].device)
                except NotImplementedError:
                    plan = False
                self._plans[key] = plan
        if plan:
            plan.launch(fast, colsums)                      # (bias gradients finished by the same finalise launch)
        else:
            slow = jobs
            for part, nb, cols, out in colsums:
                ops.colsum_finalize(part, nb, cols, out)
        self.last_dw_path = 'mfma' if plan else 'library'
        for dz, x, g in slow:                               # e.g. a first layer over 9 state features: not a multiple of 4
            torch.mm(dz.t(), x, out=g)
