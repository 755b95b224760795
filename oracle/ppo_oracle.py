"""TEST INFRASTRUCTURE ONLY - CPU restatement of the rl_games PPO hot path.

This module is the *checker* for the HIP kernels in rl_games_amd/csrc.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the
product package `rl_games_amd` never does (tests/test_boundary.py greps for that).

What it restates.  The reference (Denys88/rl_games v2.0.0) computes this path with eager
fp32 PyTorch CPU ops.  PyTorch is the third-party dependency that owns the arithmetic
(pyproject.toml:12 `torch>=2.7.0`; this image has torch 2.10.0), so the restatement below
uses the *same* torch CPU primitives in the *same* order - element-wise results are then
bit-identical to the reference and reductions use the identical ATen kernels.  Every
function cites the reference lines it follows (paths relative to the reference checkout).

Pinning.  tests/golden/make_golden.py imports the real reference from /root/reference (with
test-only stubs for the absent `gymnasium`/`tensorboardX` packages), runs its leaf functions
and a full `A2CAgent.train_epoch` on seeded inputs, and stores inputs+outputs as fixtures;
tests/test_oracle_epoch.py, tests/test_oracle_gae.py and tests/test_vs_reference_cpu.py check every
function here against those fixtures, against the imported reference functions and against
the reference's own known-answer tests (tests/test_triton_gae.py:20-61 fp64 recursion,
tests/test_multigpu_stats_sync.py pooled moments, tests/test_rms_advantage.py EMA stats,
tests/test_ppo_masking.py masked means).  Rows with no known-answer test in the reference
(clipped surrogate / value loss / bound loss / policy_kl values) are pinned to outputs of
the reference functions themselves run here.
"""
import math

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# a4/a5/a6 - GAE scan, returns, env-major flattening
# --------------------------------------------------------------------------------------


def gae_scan(rewards, values, dones_f, last_values, last_dones_f, gamma, tau):
    """[H,N,V] advantages.  rl_games/triton_kernels/gae_kernel.py:63-80 (_pytorch_gae).

    dones_f / last_dones_f are float tensors (`dones.float()`, a2c_common.py:1055-1056);
    dones_f[0] is never read (gae_kernel.py:69-75)."""
    horizon = rewards.shape[0]
    out = torch.zeros_like(rewards)
    carry = 0
    for t in range(horizon - 1, -1, -1):
        if t == horizon - 1:
            alive_next = (1.0 - last_dones_f).unsqueeze(1)
            v_next = last_values
        else:
            alive_next = (1.0 - dones_f[t + 1]).unsqueeze(1)
            v_next = values[t + 1]
        td = rewards[t] + gamma * v_next * alive_next - values[t]       # :78
        carry = td + gamma * tau * alive_next * carry                    # :79
        out[t] = carry
    return out


def gae_scalar_f64(rewards, values, dones_f, last_values, last_dones_f, gamma, tau):
    """Independent per-(env, value) fp64 recursion - the reference test's ground truth,
    tests/test_triton_gae.py:20-42.  Pure-Python loops: small shapes only."""
    horizon, num_envs, vsize = rewards.shape
    r, v, d = rewards.double(), values.double(), dones_f.double()
    lv, ld = last_values.double(), last_dones_f.double()
    out = torch.zeros(horizon, num_envs, vsize, dtype=torch.float64)
    for e in range(num_envs):
        for k in range(vsize):
            acc = 0.0
            for t in range(horizon - 1, -1, -1):
                if t == horizon - 1:
                    nxt, alive = lv[e, k], 1.0 - ld[e]
                else:
                    nxt, alive = v[t + 1, e, k], 1.0 - d[t + 1, e]
                td = r[t, e, k] + gamma * nxt * alive - v[t, e, k]
                acc = td + gamma * tau * alive * acc
                out[t, e, k] = acc
    return out.to(rewards.dtype)


def flatten_env_major(x):
    """[H,N,...] -> [N*H,...] with flat index env*H + t.
    rl_games/common/a2c_common.py:33-40 (swap_and_flatten01)."""
    if x is None:
        return None
    shape = x.size()
    return x.transpose(0, 1).reshape(shape[0] * shape[1], *shape[2:])


def returns_from_advantages(advs, values):
    """rl_games/common/a2c_common.py:1060."""
    return advs + values


# --------------------------------------------------------------------------------------
# a8 - RunningMeanStd
# --------------------------------------------------------------------------------------


def masked_mean_var(x, mask):
    """rl_games/algos_torch/torch_ext.py:182-191 (get_mean_var_with_masks, unbiased)."""
    n_valid = mask.sum().clamp(min=1.0)
    xm = x * mask
    mean = xm.sum() / n_valid
    spread = (((xm) ** 2) / n_valid).sum() - ((xm / n_valid).sum()) ** 2
    var = spread * n_valid / (n_valid - 1).clamp(min=1.0)
    return mean, var


def new_running_stats(size):
    """Initial state: mean 0, var 1 (fp64), count 1 (int64).
    rl_games/algos_torch/running_mean_std.py:46-53."""
    return {'running_mean': torch.zeros(size, dtype=torch.float64),
            'running_var': torch.ones(size, dtype=torch.float64),
            'count': torch.ones((), dtype=torch.int64)}


def running_stats_merge(state, batch_mean, batch_var, batch_count):
    """Chan parallel-variance merge in the dtype of the state.
    rl_games/algos_torch/running_mean_std.py:55-67."""
    mean, var, count = state['running_mean'], state['running_var'], state['count']
    n_old = count.to(mean.dtype)
    n_tot = n_old + batch_count
    gap = batch_mean - mean
    mean_new = mean + gap * batch_count / n_tot
    m2 = var * n_old + batch_var * batch_count + gap ** 2 * n_old * batch_count / n_tot
    return {'running_mean': mean_new, 'running_var': m2 / n_tot, 'count': count + batch_count}


def running_stats_forward(state, x, training, denorm=False, mask=None, epsilon=1e-5,
                          norm_only=False):
    """Returns (y, new_state).  rl_games/algos_torch/running_mean_std.py:69-114 for the
    non-per-channel case (axis 0).  Note batch_count = x.size(0) even when masked (:83)."""
    if training:
        if mask is not None:
            b_mean, b_var = masked_mean_var(x, mask)
        else:
            b_mean = x.mean([0])
            b_var = x.var([0], unbiased=False)
        state = running_stats_merge(state, b_mean, b_var, x.size(0))
    mean32 = state['running_mean'].float()
    var32 = state['running_var'].float()
    if denorm:
        y = torch.clamp(x, min=-5.0, max=5.0)
        y = torch.sqrt(var32 + epsilon) * y + mean32
    elif norm_only:
        y = x / torch.sqrt(var32 + epsilon)
    else:
        y = (x - mean32) / torch.sqrt(var32 + epsilon)
        y = torch.clamp(y, min=-5.0, max=5.0)
    return y, state


# --------------------------------------------------------------------------------------
# a9 - EMA advantage statistics (GeneralizedMovingStats, impl='mean_std')
# --------------------------------------------------------------------------------------


def new_moving_stats(size=1):
    """rl_games/algos_torch/moving_mean_std.py:24-28."""
    return {'step': torch.ones(1, dtype=torch.int32),
            'mean': torch.zeros(size, dtype=torch.float32),
            'sqrs': torch.zeros(size, dtype=torch.float32)}


def moving_stats_forward(state, x, training, decay, mask=None, max_val=1e5, eps=0.0):
    """Returns (y, new_state).  rl_games/algos_torch/moving_mean_std.py:136-150 with
    _update_stats :102-134 and _get_stats :52-61 ('mean_std')."""
    state = {k: v.clone() for k, v in state.items()}
    if training:
        sample = x
        skip = False
        if mask is not None:
            keep = mask.reshape(-1) > 0
            if not bool(keep.any()):
                skip = True
            else:
                sample = x[keep]
        if not skip:
            state['step'] += 1
            s_mean = torch.mean(sample, dim=0)
            s_sqr = torch.mean(sample * sample, dim=0)
            state['mean'].mul_(decay).add_((1 - decay) * s_mean)
            state['sqrs'].mul_(decay).add_((1 - decay) * s_sqr)
    mean = state['mean']
    var = state['sqrs'] - mean.pow(2)
    std = torch.sqrt(torch.clamp_min(var, 1 / max_val ** 2) + eps)
    y = torch.empty_like(x)
    y.copy_(x)
    y.sub_(mean).div_(std)
    y.clamp_(-5.0, 5.0)
    return y, state


# --------------------------------------------------------------------------------------
# a7 - prepare_dataset
# --------------------------------------------------------------------------------------


def normalize_advantages(adv, mask=None):
    """Batch normalisation of advantages with the UNBIASED std.
    rl_games/common/a2c_common.py:1634 and torch_ext.py:172-180 (masked)."""
    if mask is None:
        return (adv - adv.mean()) / (adv.std() + 1e-8)
    mean, var = masked_mean_var(adv, mask)
    return (adv - mean) / (torch.sqrt(var) + 1e-8)


def prepare_dataset(returns, values, value_stats, normalize_value=True, normalize_advantage=True,
                    mask=None, adv_ema_state=None, adv_ema_decay=0.5, freeze_critic=False):
    """rl_games/common/a2c_common.py:1586-1660 (ContinuousA2CBase.prepare_dataset), value
    and advantage part.  Returns dict(old_values, returns, advantages, value_stats[, ema])."""
    adv = returns - values                                                    # :1598
    if normalize_value:
        if freeze_critic:                                                     # :1601-1604
            values, _ = running_stats_forward(value_stats, values, training=False)
            returns, _ = running_stats_forward(value_stats, returns, training=False)
        elif mask is not None:                                                # :1605-1615
            keep = mask.bool()
            _, value_stats = running_stats_forward(value_stats, values[keep], training=True)
            _, value_stats = running_stats_forward(value_stats, returns[keep], training=True)
            values, _ = running_stats_forward(value_stats, values, training=False)
            returns, _ = running_stats_forward(value_stats, returns, training=False)
        else:                                                                 # :1616-1620
            values, value_stats = running_stats_forward(value_stats, values, training=True)
            returns, value_stats = running_stats_forward(value_stats, returns, training=True)
    adv = torch.sum(adv, axis=1)                                              # :1622
    out = {}
    if normalize_advantage:
        if adv_ema_state is not None:                                         # :1626-1632
            adv, adv_ema_state = moving_stats_forward(adv_ema_state, adv, True, adv_ema_decay,
                                                      mask=mask)
            out['adv_ema_state'] = adv_ema_state
        else:
            adv = normalize_advantages(adv, mask)
    out.update(old_values=values, returns=returns, advantages=adv, value_stats=value_stats)
    return out


# --------------------------------------------------------------------------------------
# a12' - Normal-distribution epilogue; a12 - losses; a13 - KL
# --------------------------------------------------------------------------------------


def neglogp(actions, mu, sigma, logstd):
    """rl_games/algos_torch/models.py:361-364."""
    return 0.5 * (((actions - mu) / sigma) ** 2).sum(dim=-1) \
        + 0.5 * np.log(2.0 * np.pi) * actions.size(-1) \
        + logstd.sum(dim=-1)


def normal_entropy(mu, sigma):
    """torch.distributions.Normal(mu, sigma).entropy().sum(-1), models.py:335-337."""
    return torch.distributions.Normal(mu, sigma, validate_args=False).entropy().sum(dim=-1)


def smooth_clamp(x, lo, hi):
    """rl_games/common/common_losses.py:32-36."""
    return 1 / (1 + torch.exp((-(x - lo) / (hi - lo) + 0.5) * 4)) * (hi - lo) + lo


def actor_loss(old_neglogp, new_neglogp, advantage, e_clip, ppo=True, smooth=False):
    """rl_games/common/common_losses.py:64-82 (actor_loss) / :39-61 (smoothed)."""
    if not ppo:
        return new_neglogp * advantage
    ratio = torch.exp(old_neglogp - new_neglogp)
    surr1 = advantage * ratio
    if smooth:
        surr2 = advantage * smooth_clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)
    else:
        surr2 = advantage * torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)
    return torch.max(-surr1, -surr2)


def critic_loss(old_values, values, e_clip, returns, clip_value):
    """rl_games/common/common_losses.py:16-29 (default_critic_loss)."""
    if clip_value:
        clipped = old_values + (values - old_values).clamp(-e_clip, e_clip)
        return torch.max((values - returns) ** 2, (clipped - returns) ** 2)
    return (returns - values) ** 2


def bound_loss(mu, kind='bound'):
    """rl_games/algos_torch/a2c_continuous.py:241-257."""
    if kind == 'regularisation':
        return (mu * mu).sum(axis=-1)
    soft_bound = 1.1
    high = torch.clamp_min(mu - soft_bound, 0.0) ** 2
    low = torch.clamp_max(mu + soft_bound, 0.0) ** 2
    return (low + high).sum(axis=-1)


def masked_means(losses, mask=None):
    """rl_games/algos_torch/torch_ext.py:157-170 (apply_masks)."""
    if mask is None:
        return [torch.mean(l) for l in losses], None
    m = mask.unsqueeze(1)
    denom = m.sum().clamp(min=1.0)
    return [(l * m).sum() / denom for l in losses], denom


def ppo_losses(old_neglogp, new_neglogp, advantage, old_values, values, returns, mu, entropy,
               e_clip, critic_coef, entropy_coef, bounds_coef, clip_value=True, mask=None,
               smooth=False, bound_kind='bound', ppo=True):
    """rl_games/algos_torch/a2c_continuous.py:97-134 (calc_losses); ppo = the agent's `self.ppo` (:111).
    Returns (loss, a_loss, c_loss, entropy, b_loss)."""
    a = actor_loss(old_neglogp, new_neglogp, advantage, e_clip, ppo, smooth)
    c = critic_loss(old_values, values, e_clip, returns, clip_value)
    if bounds_coef is None:
        b = torch.zeros(mu.shape[0]) if bound_kind == 'bound' else torch.zeros(mu.shape[0])
    else:
        b = bound_loss(mu, bound_kind)
    (a_m, c_m, e_m, b_m), _ = masked_means(
        [a.unsqueeze(1), c, entropy.unsqueeze(1), b.unsqueeze(1)], mask)
    coef_b = bounds_coef if bounds_coef is not None else 0.0
    loss = a_m + 0.5 * c_m * critic_coef - e_m * entropy_coef + b_m * coef_b
    return loss, a_m, c_m, e_m, b_m


def policy_kl(new_mu, new_sigma, old_mu, old_sigma, mask=None):
    """rl_games/algos_torch/torch_ext.py:27-36 with p0 = new policy, p1 = old policy
    (a2c_continuous.py:215-221; masked mean over valid rows :218-221)."""
    c1 = torch.log(old_sigma / new_sigma + 1e-5)
    c2 = (new_sigma ** 2 + (old_mu - new_mu) ** 2) / (2.0 * (old_sigma ** 2 + 1e-5))
    kl = (c1 + c2 + (-1.0 / 2.0)).sum(dim=-1)
    if mask is None:
        return kl.mean()
    return (kl * mask).sum() / mask.sum().clamp(min=1.0)


def distribution_loss_and_grads(mu, logstd, values, batch, hp, mask=None):
    """Forward + autograd backward of the model epilogue + calc_losses, from the network
    outputs (mu [mb,A], logstd [A] fixed-sigma parameter, values [mb,1]) down to scalars.
    models.py:329-347 + a2c_continuous.py:97-134,215-221.  Returns dict of scalars and
    d loss / d{mu, logstd, values}."""
    mu = mu.detach().clone().requires_grad_(True)
    logstd = logstd.detach().clone().requires_grad_(True)
    values = values.detach().clone().requires_grad_(True)
    sigma_row = torch.exp(logstd)                                # models.py:296 ('exp')
    sigma = mu * 0 + sigma_row                                   # network_builder.py:512
    ent = normal_entropy(mu, sigma)
    nlp = torch.squeeze(neglogp(batch['actions'], mu, sigma, mu * 0 + logstd))
    loss, a, c, e, b = ppo_losses(
        batch['old_logp_actions'], nlp, batch['advantages'], batch['old_values'], values,
        batch['returns'], mu, ent, hp['e_clip'], hp['critic_coef'], hp['entropy_coef'],
        hp.get('bounds_loss_coef'), hp.get('clip_value', True), mask,
        hp.get('use_smooth_clamp', False), hp.get('bound_loss_type', 'bound'), hp.get('ppo', True))
    loss.backward()
    with torch.no_grad():
        kl = policy_kl(mu.detach(), sigma.detach(), batch['mu'], batch['sigma'], mask)
    return {'loss': loss.detach(), 'a_loss': a.detach(), 'c_loss': c.detach(),
            'entropy': e.detach(), 'b_loss': b.detach(), 'kl': kl, 'neglogp': nlp.detach(),
            'sigma': sigma.detach(), 'd_mu': mu.grad, 'd_logstd': logstd.grad,
            'd_values': values.grad}


def masked_categorical(logits, masks=None):
    """CategoricalMasked (rl_games/common/extensions/distributions.py:24-47): returns
    (distribution, entropy) with disallowed logits replaced by -1e8 and dropped from the entropy."""
    if masks is None:
        cat = torch.distributions.Categorical(logits=logits)
        return cat, cat.entropy()
    cat = torch.distributions.Categorical(logits=torch.where(masks, logits, torch.tensor(-1e+8, dtype=logits.dtype)))
    p_log_p = torch.where(masks, cat.logits * cat.probs, torch.tensor(0.0, dtype=logits.dtype))
    return cat, -p_log_p.sum(-1)


def categorical_loss_and_grads(logits, values, batch, hp, mask=None, branch_sizes=None, action_masks=None):
    """Discrete agent: ModelA2C / ModelA2CMultiDiscrete epilogues (rl_games/algos_torch/models.py:95-111,
    :153-179: per-head Categorical(Masked), neglogp and entropy summed over heads) +
    DiscreteA2CAgent.calc_gradients losses and KL (rl_games/algos_torch/a2c_discrete.py:166-198),
    backward by autograd.  logits [mb, sum(branch_sizes)] (heads concatenated)."""
    logits = logits.detach().clone().requires_grad_(True)
    values = values.detach().clone().requires_grad_(True)
    sizes = [logits.shape[1]] if branch_sizes is None else list(branch_sizes)
    heads = torch.split(logits, sizes, dim=1)
    head_masks = [None] * len(sizes) if action_masks is None else torch.split(action_masks.bool(), sizes, dim=1)
    acts = batch['actions'].reshape(logits.shape[0], len(sizes))
    nlp = 0
    ent = 0
    for b, (lg, am) in enumerate(zip(heads, head_masks)):
        cat, h = masked_categorical(lg, am)
        nlp = nlp + (-cat.log_prob(acts[:, b]))
        ent = ent + h
    a = actor_loss(batch['old_logp_actions'], nlp, batch['advantages'], hp['e_clip'], hp.get('ppo', True),
                   hp.get('use_smooth_clamp', False))
    c = critic_loss(batch['old_values'], values, hp['e_clip'], batch['returns'], hp.get('clip_value', True))
    (a_m, c_m, e_m), _ = masked_means([a.unsqueeze(1), c, ent.unsqueeze(1)], mask)
    loss = a_m + 0.5 * c_m * hp['critic_coef'] - e_m * hp['entropy_coef']
    loss.backward()
    with torch.no_grad():
        kl = 0.5 * ((batch['old_logp_actions'] - nlp) ** 2)
        kl = kl.mean() if mask is None else (kl * mask).sum() / mask.sum().clamp(min=1.0)
    return {'loss': loss.detach(), 'a_loss': a_m.detach(), 'c_loss': c_m.detach(), 'entropy': e_m.detach(),
            'kl': kl, 'neglogp': nlp.detach(), 'd_logits': logits.grad, 'd_values': values.grad}


# --------------------------------------------------------------------------------------
# a15 - learning-rate control
# --------------------------------------------------------------------------------------


def adaptive_lr(current_lr, kl, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2, lr_multiplier=1.5):
    """rl_games/common/schedulers.py:27-33 (AdaptiveScheduler.update), python floats."""
    lr = current_lr
    if kl > 2.0 * kl_threshold:
        lr = max(current_lr / lr_multiplier, min_lr)
    if kl < 0.5 * kl_threshold:
        lr = min(current_lr * lr_multiplier, max_lr)
    return lr


def linear_lr(start_lr, steps, max_steps, min_lr=1e-6):
    """rl_games/common/schedulers.py:50-58 (LinearScheduler.update)."""
    mul = max(0, max_steps - steps) / max_steps
    return min_lr + (start_lr - min_lr) * mul


# --------------------------------------------------------------------------------------
# a14 - gradient truncation + Adam
# --------------------------------------------------------------------------------------


def clip_and_adam_reference(params, grads, exp_avg, exp_avg_sq, step, lr, grad_norm=1.0,
                            truncate=True, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One optimiser step through the very torch objects the reference uses:
    clip_grad_norm_ (a2c_common.py:510-511) then optim.Adam(eps=1e-8).step()
    (a2c_continuous.py:44-48, a2c_common.py:513).  Tensors are updated out of place and
    returned as (params, exp_avg, exp_avg_sq, total_norm)."""
    ps = [torch.nn.Parameter(p.detach().clone()) for p in params]
    for p, g in zip(ps, grads):
        p.grad = g.detach().clone()
    total_norm = torch.zeros(())
    if truncate:
        total_norm = torch.nn.utils.clip_grad_norm_(ps, grad_norm)
    opt = torch.optim.Adam(ps, lr, betas=betas, eps=eps, weight_decay=weight_decay)
    for p, m, v in zip(ps, exp_avg, exp_avg_sq):
        opt.state[p] = {'step': torch.tensor(float(step)), 'exp_avg': m.detach().clone(),
                        'exp_avg_sq': v.detach().clone()}
    opt.step()
    return ([p.detach() for p in ps], [opt.state[p]['exp_avg'] for p in ps],
            [opt.state[p]['exp_avg_sq'] for p in ps], total_norm)


# --------------------------------------------------------------------------------------
# a19 - cross-rank pooled merge of RunningMeanStd
# --------------------------------------------------------------------------------------


def stats_totals(state):
    """rl_games/common/a2c_common.py:43-47 (_running_stats_totals)."""
    n = state['count']
    return (n.clone(), state['running_mean'] * n,
            (state['running_var'] + state['running_mean'] ** 2) * n)


def pooled_merge(state, snapshot, all_reduce_sum):
    """rl_games/common/a2c_common.py:61-93 (merge_rank_stats).  `all_reduce_sum(t)` sums t
    in place across ranks.  Returns (new_state, new_snapshot)."""
    cur = stats_totals(state)
    if snapshot is None:
        deltas = [c.clone() for c in cur]
        base = [torch.zeros_like(c) for c in cur]
    else:
        deltas = [c - p for c, p in zip(cur, snapshot)]
        base = snapshot
    for t in deltas:
        all_reduce_sum(t)
    n = base[0] + deltas[0]
    s1 = base[1] + deltas[1]
    s2 = base[2] + deltas[2]
    mean = s1 / n
    var = (s2 / n - mean ** 2).clamp_(min=1e-8)
    new_state = {'count': n.clone(), 'running_mean': mean, 'running_var': var}
    return new_state, (n.clone(), s1.clone(), s2.clone())


class StatsSyncOracle:
    """CPU restatement of the flat-buffer form of the pooled merge that the product runs as two
    kernels around one collective (rl_games_amd/distributed.py:StatsSync, csrc/running_stats.hip):
    same interface as rl_games_amd.ops.StatsSyncKernels, arithmetic = stats_totals / pooled_merge above
    (rl_games/common/a2c_common.py:43-93, :124-141) segment by segment.  Test infrastructure."""

    PACK_DELTAS, PACK_SEED, PACK_STATE = 0, 1, 2
    APPLY_MERGE, APPLY_STATE = 0, 2

    def __init__(self, modules):
        self.modules = list(modules)
        self.dims = [int(m.running_mean.numel()) for m in self.modules]
        self.offsets, off = [], 0
        for d in self.dims:
            self.offsets.append(off)
            off += 1 + 2 * d
        self.flat_size = off

    def _state(self, m):
        return {'count': m.count.detach().cpu(), 'running_mean': m.running_mean.detach().cpu(),
                'running_var': m.running_var.detach().cpu()}

    def pack(self, has_snapshot, snapshot, out, mode):
        for m, d, o, has in zip(self.modules, self.dims, self.offsets, has_snapshot):
            st = self._state(m)
            if mode == self.PACK_STATE:
                vals = (st['count'].double().reshape(1), st['running_mean'], st['running_var'])
            else:
                n, s1, s2 = stats_totals(st)
                vals = (n.double().reshape(1), s1, s2)
            flat = torch.cat([v.reshape(-1) for v in vals])
            if mode == self.PACK_SEED:
                snapshot[o:o + 1 + 2 * d] = flat.to(snapshot.device)
            elif mode == self.PACK_DELTAS and has:
                out[o:o + 1 + 2 * d] = (flat - snapshot[o:o + 1 + 2 * d].cpu()).to(out.device)
            else:
                out[o:o + 1 + 2 * d] = flat.to(out.device)

    def apply(self, has_snapshot, snapshot, reduced, mode):
        for m, d, o, has in zip(self.modules, self.dims, self.offsets, has_snapshot):
            seg = reduced[o:o + 1 + 2 * d].cpu()
            if mode == self.APPLY_STATE:
                m.count.copy_(seg[0].to(torch.int64))
                m.running_mean.copy_(seg[1:1 + d])
                m.running_var.copy_(seg[1 + d:])
                continue
            base = snapshot[o:o + 1 + 2 * d].cpu() if has else torch.zeros(1 + 2 * d, dtype=torch.float64)
            n = base[0] + seg[0]
            s1 = base[1:1 + d] + seg[1:1 + d]
            s2 = base[1 + d:] + seg[1 + d:]
            mean = s1 / n
            var = (s2 / n - mean ** 2).clamp_(min=1e-8)
            m.count.copy_(n.to(torch.int64))
            m.running_mean.copy_(mean)
            m.running_var.copy_(var)
            snapshot[o:o + 1 + 2 * d] = torch.cat([n.reshape(1), s1, s2]).to(snapshot.device)


# --------------------------------------------------------------------------------------
# a3 - per-step rollout glue
# --------------------------------------------------------------------------------------


def shape_rewards(rewards, scale=1.0, shift=0.0, min_val=-math.inf, max_val=math.inf, log_val=False):
    """rl_games/common/tr_helpers.py:33-42 (DefaultRewardsShaper.__call__)."""
    r = rewards + shift
    r = r * scale
    r = torch.clamp(r, min_val, max_val)
    if log_val:
        r = torch.log(r)
    return r


def bootstrap_timeouts(shaped, values, time_outs, gamma):
    """rl_games/common/a2c_common.py:1022-1023."""
    return shaped + gamma * values * time_outs.unsqueeze(1).float()


def average_meter_update(mean, current_size, values, max_size):
    """rl_games/algos_torch/torch_ext.py:333-342 (AverageMeter.update).
    Returns (mean, current_size)."""
    size = values.size()[0]
    if size == 0:
        return mean, current_size
    new_mean = torch.mean(values.float(), dim=0)
    size = int(np.clip(size, 0, max_size))
    old_size = min(max_size - size, current_size)
    size_sum = old_size + size
    return (mean * old_size + new_mean * size) / size_sum, size_sum


def episode_bookkeeping(cur_rewards, cur_shaped, cur_lengths, rewards, shaped, dones,
                        live_rows=None, num_agents=1):
    """rl_games/common/a2c_common.py:1027-1051.  Returns the updated accumulators plus the
    finished-episode rows that feed the AverageMeters (before zeroing)."""
    if live_rows is not None:
        cur_rewards = cur_rewards + rewards * live_rows.unsqueeze(1)
        cur_shaped = cur_shaped + shaped * live_rows.unsqueeze(1)
        cur_lengths = cur_lengths + live_rows
    else:
        cur_rewards = cur_rewards + rewards
        cur_shaped = cur_shaped + shaped
        cur_lengths = cur_lengths + 1
    done_idx = dones.nonzero(as_tuple=False)[::num_agents]
    finished = (cur_rewards[done_idx], cur_shaped[done_idx], cur_lengths[done_idx], done_idx)
    alive = (1.0 - dones.float()).unsqueeze(1)
    return cur_rewards * alive, cur_shaped * alive, cur_lengths * alive.squeeze(1), finished
