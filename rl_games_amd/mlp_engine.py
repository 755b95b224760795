"""Manual forward/backward of the actor-critic MLP (no autograd graph).

For the plain-MLP policies of the BASELINE configs the training step is a fixed chain
    obs -> [Linear -> act] x L -> fused (value | mu) head -> fused PPO loss kernel
so the backward pass is written out explicitly instead of being recorded by autograd:
  * the GEMMs are the same rocBLAS/hipBLASLt calls nn.Linear's autograd issues
    (forward addmm; dX = dZ W; dW = dZ^T A), written straight into preallocated buffers and
    into the gradient arena (no AccumulateGrad adds, no per-step allocations);
  * activation backward + bias-gradient column sum is ONE pass (csrc/mlp_fused.hip) instead of
    aten's elu_backward followed by a separate sum(0) reduction;
  * the value and mu heads (`a2c_network.value`, `a2c_network.mu`,
    rl_games/algos_torch/network_builder.py:295-311) are adjacent in the parameter arena and
    run as one [1+A, K] GEMM operand; their bias gradients come out of the loss kernel.
  * an optional single-layer LSTM between the trunk and the heads (BASELINE config #5,
    `rnn: {name: lstm, layers: 1}`, network_builder.py:447-512) runs as ONE sequence-persistent
    kernel per direction (csrc/lstm.hip) instead of a Python loop of per-timestep MIOpen calls
    with done resets (rl_games/common/layers/recurrent.py:26-58) and autograd BPTT through it;
    its input projection and all four weight/bias gradients are whole-sequence GEMMs.
Numerics: aten's formulas - by default (in-place activations) act' from the layer output like
elu_backward(is_result=True): h > 0 ? 1 : h + 1; with `inplace_act=False` from the pre-activation
(exp(z)); GEMM
results are the library's, as before.  Parameters keep their reference names and shapes.
"""
import torch
from torch import nn

from . import ops


class ManualMLP:
    def __init__(self, net, arena, max_rows, mfma_dw=True, inplace_act=True, fused_chain=True):
        """net: policy.ActorCriticNetwork (no RNN); arena: FlatArena laid out by `layout(net)`.
        mfma_dw: weight gradients through the one-launch f32-MFMA kernel (csrc/mlp_dw.hip).
        fused_chain: the whole MLP (observation normaliser, hidden layers, fused value|mu head) as
        ONE forward launch and ONE backward launch with the activations of a row tile resident in
        LDS (csrc/mlp_chain.hip); plain MLP policies only (the LSTM path keeps the per-layer GEMMs).
        (Measured and rejected: forking the library dW GEMMs onto a second stream - multi-branch
        hipGraphs cost more in cross-branch dependencies than the overlap gains.)"""
        self.net = net
        self.mfma_dw = mfma_dw
        self._dw_plans = {}
        self.last_dw_path = None
        self.last_dw_jobs = None
        self._x_normalised = False
        self.arena = arena
        self.linears = [m for m in net.actor_mlp if isinstance(m, nn.Linear)]
        acts = [m for m in net.actor_mlp if not isinstance(m, nn.Linear)]
        if len(acts) != len(self.linears):
            raise NotImplementedError('unexpected MLP structure')
        name = {nn.ELU: 'elu', nn.ReLU: 'relu', nn.Tanh: 'tanh', nn.Identity: 'None'}.get(type(acts[0]))
        if name is None or any(type(a) is not type(acts[0]) for a in acts):
            raise NotImplementedError('manual MLP engine supports elu / relu / tanh / identity trunks')
        if isinstance(acts[0], nn.ELU) and acts[0].alpha != 1.0:
            raise NotImplementedError('elu alpha != 1')
        self.act_name = name
        self.act_kind = ops.ACT_KINDS[name]
        self.act_module = acts[0]
        if not isinstance(net.value_act, nn.Identity) or not isinstance(net.mu_act, nn.Identity):
            raise NotImplementedError('head activations are not supported by the manual MLP engine')
        if any(l.out_features % 4 for l in self.linears):
            raise NotImplementedError('hidden widths must be multiples of 4')
        self.A = net.mu.out_features
        self.V = net.value.out_features
        K = net.mu.in_features
        self.lstm = None
        if getattr(net, 'has_rnn', False):
            if net.rnn_name != 'lstm' or net.rnn_layers != 1 or not ops.lstm_supported(net.rnn_units):
                raise NotImplementedError('manual engine: only a single-layer LSTM with 16/32/64 units')
            self.lstm = net.rnn.rnn
            self.Hr = net.rnn_units
        wp, wg = arena.span(net.value.weight, net.mu.weight)
        bp, bg = arena.span(net.value.bias, net.mu.bias)
        self.head_w, self.head_w_grad = wp.view(self.V + self.A, K), wg.view(self.V + self.A, K)
        self.head_b, self.head_b_grad = bp, bg
        self._head_mfma = (mfma_dw and self.V + self.A <= 64 and ops.mlp_rowgemm_supported(K, K)
                           and self.head_w.data_ptr() % 16 == 0)
        dev = wp.device
        self.max_rows = max_rows
        widths = [l.out_features for l in self.linears]
        # inplace_act: the activation overwrites the pre-activation (one [rows, width] buffer per
        # layer instead of two) and backward takes act' from the output, like torch's in-place ELU
        # (elu_backward(is_result=True): h > 0 ? 1 : h + 1).  Halves the activation footprint of a
        # minibatch (276 -> 184 MB at 32,768 rows), which then fits the 256 MB Infinity Cache together
        # with the gradients.
        self.inplace_act = bool(inplace_act) and self.act_kind in (1, 2, 3)
        self.Hs = [torch.empty(max_rows, w, device=dev) for w in widths]      # activations
        self.Z = self.Hs if (self.inplace_act or self.act_kind == 0) else \
            [torch.empty(max_rows, w, device=dev) for w in widths]            # pre-activations
        self.dA = [torch.empty(max_rows, w, device=dev) for w in widths]      # d activation / d pre-act
        self.heads = torch.empty(max_rows, self.V + self.A, device=dev)
        self.d_heads = torch.empty(max_rows, self.V + self.A, device=dev)
        self.nb = [ops.act_bwd_blocks(max_rows, w) for w in widths]
        self.chain = None
        self._deferred = None            # arguments of a training forward that backward() will launch (forward_obs(defer=True))
        self.last_step_fused = False     # the last backward() ran forward + loss + backward as one launch
        self._pending_backward = False
        self._fused_trunk = False
        if fused_chain and self.lstm is None:
            try:
                layers = [(l.weight, l.bias, self.act_name) for l in self.linears]
                layers.append((self.head_w, self.head_b, 'None'))
                self.chain = ops.MlpChain(layers, dev, weights_version=arena.weights_token)
            except NotImplementedError:
                self.chain = None
        # Recurrent policies (round 3): the trunk IN FRONT of the LSTM - observation normaliser, hidden layers and
        # the gate-input product W_ih x + (b_ih + b_hh) as the chain's last (linear) layer - is the same fused
        # forward / backward launch pair; the sequence-persistent LSTM kernels and the head product follow.
        self.chain_rnn = None
        if fused_chain and self.lstm is not None:
            try:
                self.bias_sum = torch.empty(4 * self.Hr, device=dev)
                layers = [(l.weight, l.bias, self.act_name) for l in self.linears]
                layers.append((self.lstm.weight_ih_l0, self.bias_sum, 'None'))
                self.chain_rnn = ops.MlpChain(layers, dev, weights_version=arena.weights_token)
            except NotImplementedError:
                self.chain_rnn = None
        if self.chain is not None or self.chain_rnn is not None:
            self.nb = [(max_rows + 15) // 16 for _ in widths]      # one partial row per 16-row group at most
            self.xn = torch.empty(max_rows, self.linears[0].in_features, device=dev)
        self.partials = [torch.empty(nb * w, dtype=torch.float64, device=dev) for nb, w in zip(self.nb, widths)]
        if self.lstm is not None:
            Hr = self.Hr
            self.gates = torch.empty(max_rows, 4 * Hr, device=dev)
            self.d_gates = torch.empty(max_rows, 4 * Hr, device=dev)
            self.rnn_out = torch.empty(max_rows, Hr, device=dev)
            self.d_rnn_out = torch.empty(max_rows, Hr, device=dev)
            self.c_all = torch.empty(max_rows, Hr, device=dev)
            self.hprev = torch.empty(max_rows, Hr, device=dev)
            if self.chain_rnn is None:
                self.bias_sum = torch.empty(4 * Hr, device=dev)
            self.gate_partials = torch.empty(ops.act_bwd_blocks(max_rows, 4 * Hr) * 4 * Hr,
                                             dtype=torch.float64, device=dev)
            # final states of the last forward, ping-pong so that a caller may feed them back in
            self._state_buf = [[torch.empty(1, max_rows, Hr, device=dev) for _ in range(2)] for _ in range(2)]
            self._state_flip = 0
            self.last_states = None

    @staticmethod
    def layout(net):
        """Physical arena order: every weight MATRIX first (parameters() order, the value and mu
        head weights last and adjacent), then the vectors (biases, sigma; value and mu head biases
        last and adjacent).  Matrix sizes are multiples of 4 floats for every supported shape, so
        all weight / weight-gradient views stay 16-byte aligned (vector loads/stores of the MFMA
        kernels); the logical (optimizer-state) order is untouched."""
        head_w = [net.value.weight, net.mu.weight]
        head_b = [net.value.bias, net.mu.bias]
        skip = {id(p) for p in head_w + head_b}
        rest = [p for p in net.parameters() if id(p) not in skip]
        mats = [p for p in rest if p.dim() >= 2]
        vecs = [p for p in rest if p.dim() < 2]
        # the layers' bias vectors directly behind the last matrix: the pipelined forward kernels read up to 48 bytes
        # past a matrix whose width is no multiple of 16 and accept that only when the bytes belong to another array
        # of the SAME network (csrc/mlp_chain.hip, chain_pipe_fill) - sigma and the like come after them
        linear_biases = {id(m.bias) for m in net.modules() if isinstance(m, torch.nn.Linear) and m.bias is not None}
        vecs.sort(key=lambda p: id(p) not in linear_biases)
        return mats + head_w + vecs + head_b

    # ------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, keep=True, rnn_states=None, dones=None, seq_length=1, raw_rms=None, eps=1e-5):
        """x: [rows, in] normalised observations.  Returns heads [rows, V+A] (col 0..V-1 value,
        then mu).  `keep` retains what backward() needs (activations, LSTM cell states).  LSTM policies: rows are
        ordered (sequence, t) with `seq_length` steps each, rnn_states = (h0, c0) of shape
        [1, rows/seq_length, H], dones [rows] u8 resets the state entering a step (or None);
        the final states are left in `self.last_states`.
        raw_rms (recurrent policies on the fused trunk, `chain_rnn`): x holds RAW observations and raw_rms =
        (running_mean, running_var) or () for normalize_input off - the launch normalises on the way in."""
        rows = x.shape[0]
        if self._pending_backward and not keep:
            # in-place activations: an inference forward reuses the buffers backward() reads
            raise RuntimeError('ManualMLP.forward(keep=False) would overwrite the activations a pending '
                               'backward() still needs')
        self._pending_backward = bool(keep)
        a = x
        fused_trunk = raw_rms is not None
        if fused_trunk and (self.chain_rnn is None or self.lstm is None):
            raise ValueError('raw_rms: the fused recurrent trunk is not available for this policy')
        for l, lin in enumerate(() if fused_trunk else self.linears):
            z = self.Z[l][:rows]
            torch.addmm(lin.bias, a, lin.weight.t(), out=z)
            h = self.Hs[l][:rows]                      # same storage as z when inplace_act
            if self.act_kind == 1:
                torch.ops.aten.elu.out(z, out=h)
            elif self.act_kind == 2:
                torch.clamp_min(z, 0.0, out=h)
            elif self.act_kind == 3:
                torch.tanh(z, out=h)
            else:
                h = z
            a = h
        if self.lstm is not None:
            rnn = self.lstm
            S = rows // seq_length
            if S * seq_length != rows:
                raise ValueError(f'rows ({rows}) must be a multiple of seq_length ({seq_length})')
            h0, c0 = rnn_states[0][0], rnn_states[1][0]
            if h0.shape[0] != S or not h0.is_contiguous() or not c0.is_contiguous():
                raise ValueError(f'rnn_states must be contiguous [1, {S}, {self.Hr}] tensors')
            torch.add(rnn.bias_ih_l0, rnn.bias_hh_l0, out=self.bias_sum)
            gates = self.gates[:rows]
            if fused_trunk:
                rms = raw_rms if len(raw_rms) else None
                if keep:
                    acts = [h[:rows] for h in self.Hs]
                    xn = self.xn[:rows] if rms is not None else None
                    self.chain_rnn.forward(x, gates, act_out=acts, rms=rms, eps=eps, xn_out=xn)
                    x = xn if rms is not None else x          # what the first layer's weight gradient reads
                    a = acts[-1]
                else:
                    self.chain_rnn.forward(x, gates, rms=rms, eps=eps)
                    a = None
            else:
                torch.addmm(self.bias_sum, a, rnn.weight_ih_l0.t(), out=gates)
            out = self.rnn_out[:rows]
            # Final states are produced for inference calls only (keep=False: rollout / get_values),
            # into whichever buffer pair the inputs do NOT live in.  A training forward must not touch
            # these buffers: they hold the live rollout state that is carried into the next epoch.
            hT = cT = None
            if not keep:
                self._state_flip = 1 if h0.data_ptr() == self._state_buf[0][0].data_ptr() else 0
                hT = self._state_buf[self._state_flip][0][:, :S]
                cT = self._state_buf[self._state_flip][1][:, :S]
            ops.lstm_seq_forward(gates, rnn.weight_hh_l0, h0, c0, dones, out,
                                 self.c_all[:rows] if keep else None, self.hprev[:rows] if keep else None,
                                 None if hT is None else hT[0], None if cT is None else cT[0],
                                 seq_len=seq_length)
            self.last_states = None if hT is None else (hT, cT)
            self._rnn_in, self._c0, self._dones, self._T = a, c0, dones, seq_length
            a = out
        heads = self.heads[:rows]
        if self._head_mfma and a.is_contiguous() and a.data_ptr() % 16 == 0:
            # skinny output (1 + A columns): the LDS-free MFMA kernel beats the library GEMM 2.5x here
            # (8 vs 20 us at 32,768 x 100 -> 22); for the wide hidden layers the library stays ahead.
            ops.mlp_linear_act_forward(a, self.head_w, self.head_b, heads, act_kind=0)
        else:
            torch.addmm(self.head_b, a, self.head_w.t(), out=heads)
        self._x, self._rows, self._last = x, rows, a
        self._fused_trunk = fused_trunk and keep
        return heads

    @torch.no_grad()
    def forward_obs(self, obs, rms=None, eps=1e-5, keep=True, rms_fold=None, defer=False):
        """Fused chain: obs [rows, in] RAW observations; rms = (running_mean, running_var) fp64 or None
        (normalize_input off).  One launch: normalise -> hidden layers -> heads.  keep=True retains
        the activations and the normalised observations for backward(); keep=False (rollout,
        get_values) writes nothing but the heads, so it cannot disturb a pending backward.
        rms_fold: RunningMeanStd.fold_buffers() - the statistics update of a training forward happens
        in the launch's prologue.  defer=True (training forward whose backward() follows with a ppo_loss descriptor):
        nothing is launched yet - backward() issues forward + loss + backward as ONE launch where the chain has that
        form (ops.MlpChain.step: minibatches < 16,384 rows), else the forward now-deferred and then the backward; the
        returned heads view is where the values will be."""
        rows = obs.shape[0]
        heads = self.heads[:rows]
        self._deferred = None
        if keep:
            acts = [h[:rows] for h in self.Hs]
            xn = self.xn[:rows] if rms is not None else None
            if defer:
                self._deferred = dict(x=obs, heads=heads, act_out=acts, rms=rms, eps=eps, xn_out=xn, rms_fold=rms_fold)
            else:
                self.chain.forward(obs, heads, act_out=acts, rms=rms, eps=eps, xn_out=xn, rms_fold=rms_fold)
            self._x = xn if rms is not None else obs
            self._x_normalised = rms is not None          # (went through the normaliser's clamp to [-5, 5])
            self._rows, self._last = rows, acts[-1]
            self._pending_backward = True
        else:
            self.chain.forward(obs, heads, rms=rms, eps=eps)
        return heads

    def _one_launch_step(self, fwd, d_heads, dzs, parts, ppo_loss):
        ok = self.chain.step(fwd['x'], fwd['heads'], fwd['act_out'], d_heads, dzs, parts, ppo_loss, rms=fwd['rms'],
                             eps=fwd['eps'], xn_out=fwd['xn_out'], rms_fold=fwd['rms_fold'])
        self.last_step_fused = bool(ok)
        return ok

    def gradient_elements(self):
        """Number of arena gradient elements that backward() + the loss finalise produce on the fused
        path: trunk weights and biases, the fused head matrix, the head biases and logstd."""
        n = sum(l.weight.numel() + l.bias.numel() for l in self.linears)
        return n + self.head_w_grad.numel() + (self.V + self.A) + self.A

    def values_view(self, heads):
        return heads[:, :self.V]

    def mu_view(self, heads):
        return heads[:, self.V:]

    @torch.no_grad()
    def backward(self, d_heads, loss_finalize=None, norm=None, ppo_loss=None):
        """d_heads [rows, V+A] = d loss / d heads.  Writes every weight/bias gradient of the trunk
        and the head WEIGHT gradient into the arena (head bias gradients are written by the loss
        finalise kernel).  The dX chain runs first; the weight gradients - which nothing in that
        chain waits for - are then ONE f32-MFMA launch for all layers (csrc/mlp_dw.hip), or the
        library GEMMs when a shape is outside that kernel's envelope / `mfma_dw` is off.
        loss_finalize: ops.loss_finalize_desc(...) - folded in the weight-gradient finalise launch when
        there is one, launched on its own otherwise.  norm = (partials, grad_scale, step_counter): see
        ops.MlpDwPlan.launch; returns the number of valid norm partials, or None when the gradient norm
        was not produced (a gradient of this step did not come out of that launch).  ppo_loss
        (ops.ppo_loss_desc, fused chain only): the backward launch evaluates the PPO loss first and so
        produces d_heads itself."""
        rows = self._rows
        L = len(self.linears)
        self._pending_backward = False
        if self.chain is not None:
            # one launch: dZ of every hidden layer + per-workgroup bias-gradient column sums
            nblk = self.chain.num_blocks(rows, 1)
            acts = [h[:rows] for h in self.Hs]
            dzs = [d[:rows] for d in self.dA]
            parts = [p[:nblk * w.out_features] for p, w in zip(self.partials, self.linears)]
            fwd, self._deferred = self._deferred, None
            self.last_step_fused = False
            if fwd is None or not (ppo_loss is not None and self._one_launch_step(fwd, d_heads, dzs, parts, ppo_loss)):
                if fwd is not None:                     # (no one-launch form for this shape: the two launches)
                    self.chain.forward(fwd['x'], fwd['heads'], act_out=fwd['act_out'], rms=fwd['rms'], eps=fwd['eps'],
                                       xn_out=fwd['xn_out'], rms_fold=fwd['rms_fold'])
                self.chain.backward(d_heads, acts, dzs, parts, ppo_loss=ppo_loss)
            # (dZ, X, grad, layer): the layer index names dZ's row of the backward's gradient maxima (fp16 form)
            jobs = [(d_heads, acts[-1], self.head_w_grad, L)]
            colsums = []
            for l in range(L - 1, -1, -1):
                lin = self.linears[l]
                jobs.append((dzs[l], acts[l - 1] if l > 0 else self._x, lin.weight.grad, l))
                colsums.append((parts[l], nblk, lin.out_features, lin.bias.grad))
            # the fp16 form of the weight-gradient launch splits X under the forward's FIXED scales: hidden activations, and
            # observations that went through the normaliser's clamp (raw observations have no bound: the bf16 form then)
            maxima = self.chain.gradient_maxima(rows) if self._x_normalised else None
            return self._weight_grads(jobs, rows, colsums, loss_finalize, norm, maxima=maxima)
        jobs = [(d_heads, self._last, self.head_w_grad)]           # (dZ, X, grad) per weight matrix
        colsums = []                                               # (partials, blocks, cols, bias.grad)
        if self.lstm is not None:
            rnn = self.lstm
            d_out = self.d_rnn_out[:rows]
            if self.V + self.A <= ops.NARROW_MAX:
                ops.narrow_dx(d_heads, self.head_w, d_out)
            else:
                torch.mm(d_heads, self.head_w, out=d_out)
            gates, dg = self.gates[:rows], self.d_gates[:rows]
            ops.lstm_seq_backward(gates, self.c_all[:rows], self._c0, self._dones, rnn.weight_hh_l0, d_out, dg,
                                  self._T)
            G = 4 * self.Hr
            nbg = ops.act_bwd_blocks(rows, G)
            gpart = self.gate_partials[:nbg * G]
            ops.act_bwd_colsum(dg, None, dg, 0, gpart, nbg)          # identity: column sums only
            colsums.append((gpart, nbg, G, rnn.bias_ih_l0.grad))
            colsums.append((gpart, nbg, G, rnn.bias_hh_l0.grad))     # d b_hh = d b_ih
            jobs.append((dg, self.hprev[:rows], rnn.weight_hh_l0.grad))
            jobs.append((dg, self._rnn_in, rnn.weight_ih_l0.grad))
            if self._fused_trunk:
                # dX chain of the trunk in ONE launch, from d gates down: dZ of every hidden layer + the
                # per-workgroup bias-gradient column sums (the chain's "head" layer is W_ih)
                nblk = self.chain_rnn.num_blocks(rows, 1)
                acts = [h[:rows] for h in self.Hs]
                dzs = [t[:rows] for t in self.dA]
                parts = [p[:nblk * w.out_features] for p, w in zip(self.partials, self.linears)]
                self.chain_rnn.backward(dg, acts, dzs, parts)
                for l in range(L - 1, -1, -1):
                    lin = self.linears[l]
                    jobs.append((dzs[l], acts[l - 1] if l > 0 else self._x, lin.weight.grad))
                    colsums.append((parts[l], nblk, lin.out_features, lin.bias.grad))
                return self._weight_grads(jobs, rows, colsums, loss_finalize)
            d = self.dA[L - 1][:rows]
            torch.mm(dg, rnn.weight_ih_l0, out=d)
        else:
            d = self.dA[L - 1][:rows]
            torch.mm(d_heads, self.head_w, out=d)
        for l in range(L - 1, -1, -1):
            lin = self.linears[l]
            w = lin.out_features
            nb = ops.act_bwd_blocks(rows, w)
            part = self.partials[l][:nb * w]
            ops.act_bwd_colsum(d, self.Z[l][:rows], d, self.act_kind + (16 if self.inplace_act else 0), part, nb)
            colsums.append((part, nb, w, lin.bias.grad))
            a_prev = self.Hs[l - 1][:rows] if l > 0 else self._x
            jobs.append((d, a_prev, lin.weight.grad))
            if l > 0:
                d_prev = self.dA[l - 1][:rows]
                torch.mm(d, lin.weight, out=d_prev)
                d = d_prev
        return self._weight_grads(jobs, rows, colsums, loss_finalize)

    def _weight_grads(self, jobs, rows, colsums=(), loss_finalize=None, norm=None, maxima=None):
        """jobs: (dZ [rows, No], X [rows, Mi], grad [No, Mi][, layer index]).  maxima: ops.MlpChain.gradient_maxima() of the
        step, or None (then, or when a job carries no layer index, the launch runs its bf16 form).  Everything inside the MFMA kernel's
        envelope (Mi % 4 == 0, 16-byte aligned contiguous operands) goes into one launch; the rest
        (e.g. a first layer over 3 observations) uses the library GEMM."""
        fast, slow = [], []
        norm_blocks = None
        if self.mfma_dw:
            for job in jobs:
                g = job[2]
                ok = (g.shape[1] % 4 == 0 and g.shape[1] >= 4
                      and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in job[:3]))
                (fast if ok else slow).append(job)
        else:
            slow = list(jobs)
        if fast:
            fast.sort(key=lambda job: -job[2].numel())          # heaviest layer's blocks first in the launch
            key = (rows,) + tuple(tuple(j[2].shape) for j in fast)
            plan = self._dw_plans.get(key)
            if plan is None:
                try:
                    plan = ops.MlpDwPlan([tuple(j[2].shape) for j in fast], rows, fast[0][2].device)
                except NotImplementedError:
                    plan = False
                self._dw_plans[key] = plan
            if plan:
                # bias gradients (and the loss partials) finished in the same finalise launch; the gradient
                # norm as well when every gradient of the step is written there
                # (a norm buffer that is too small for this plan's finalise grid: the caller's grad_sumsq
                #  launch takes over instead of an error in the middle of an epoch)
                whole = (norm is not None and not slow and loss_finalize is not None
                         and norm[0].numel() >= plan.finalize_blocks(colsums, loss_finalize))
                mx = None
                if maxima is not None and all(len(j) == 4 for j in fast):
                    mx = (maxima, [j[3] for j in fast],
                          [ops.SPLIT_SCALE_OBS_NORM if j[3] == 0 else ops.SPLIT_SCALE_HIDDEN for j in fast],
                          self.chain.maxima_rows_per_entry)
                triples = [j[:3] for j in fast]
                norm_blocks = plan.launch(triples, colsums, loss_finalize, norm if whole else None, maxima=mx)
                if not whole:
                    norm_blocks = None
                colsums = ()
                loss_finalize = None
                self.last_dw_jobs = (triples, plan, mx)        # bench.py times this launch after the run
            else:
                slow = slow + fast
                fast = []
        if loss_finalize is not None:
            ops.ppo_loss_finalize_from(loss_finalize, jobs[0][2].device)
        for part, nb, cols, out in colsums:
            ops.colsum_finalize(part, nb, cols, out)
        self.last_dw_path = 'mfma' if fast else 'library'
        self.last_dw_library_jobs = 0
        for dz, x, g in (j[:3] for j in slow):
            if (g.shape[1] <= ops.NARROW_MAX and g.shape[0] <= 256 and dz.stride(1) == 1 and x.stride(1) == 1
                    and g.is_contiguous()):
                ops.narrow_dw(dz, x, g)            # a first layer over a few observations (config #5: obs 3)
            else:
                torch.mm(dz.t(), x, out=g)
                self.last_dw_library_jobs += 1
        return norm_blocks
