"""Torch-op forms of the update's arithmetic for shapes OUTSIDE the HIP kernels' envelope.

SURVEY 8a'-13: the reference's API promises generality that none of the BASELINE configurations use; the plan was
"native kernels for the configs' fast path, reference-equivalent torch ops elsewhere".  This module is the "elsewhere"
for `value_size > 1` (rewards / values / returns of shape [B, V], rl_games/common/a2c_common.py:1622 sums the advantages
over V): the fused loss kernels (csrc/ppo_loss.hip, ppo_loss_tile.hpp) carry ONE value column, so an agent with V > 1
takes autograd through these functions instead - on the device, with torch's own kernels; nothing here runs on the hot
path of a BASELINE configuration.  Pure functions of tensors (any device): tests/test_vs_reference_cpu.py holds them to
the reference's own functions.
"""
import math

import torch


def _clip_surrogate(ratio, lo, hi, smooth):
    if not smooth:
        return torch.clamp(ratio, lo, hi)
    # common_losses.py:32-36 (smooth_clamp): a sigmoid ramp between the two bounds
    t = (0.5 - (ratio - lo) / (hi - lo)) * 4
    return (hi - lo) / (1 + torch.exp(t)) + lo


def ppo_loss(mu, logstd, values, actions, old_neglogp, advantages, old_values, returns, *, e_clip, critic_coef,
             entropy_coef, bounds_coef, bound_kind, clip_value, smooth, mask=None, sigma_fn=None):
    """The model epilogue + calc_losses of the continuous agent (models.py:329-364, a2c_continuous.py:97-134,
    common_losses.py:16-82) for a fixed-sigma policy: mu [mb, A], logstd [A] (the parameter), values / old_values /
    returns [mb, V].  bound_kind: 0 none, 1 'bound', 2 'regularisation' (ops.BOUND_KINDS); smooth: ops.SURROGATE_* (False /
    True / 2 = ppo: False).
    sigma_fn: None (sigma = exp(logstd)) or the network's raw -> (sigma, log sigma) map (`apply_sigma_parametrization`,
    models.py:272-301: bounds, a floor, the softplus / linear forms).
    Returns (loss, dict of the detached scalars a_loss / c_loss / entropy / b_loss, sigma [A])."""
    mb, A = mu.shape
    if sigma_fn is None:
        sigma = torch.exp(logstd)
    else:
        sigma, logstd = sigma_fn(logstd)
    z = (actions - mu) / sigma
    neglogp = 0.5 * (z * z).sum(dim=-1) + 0.5 * math.log(2.0 * math.pi) * A + logstd.sum(dim=-1)
    entropy = (0.5 + 0.5 * math.log(2.0 * math.pi) + torch.log(sigma)).sum(dim=-1).expand(mb)
    if int(smooth) == 2:                                    # ppo: False   common_losses.py:59, 80
        a_rows = neglogp * advantages
    else:
        ratio = torch.exp(old_neglogp - neglogp)
        a_rows = torch.max(-advantages * ratio, -advantages * _clip_surrogate(ratio, 1.0 - e_clip, 1.0 + e_clip, int(smooth) == 1))
    if clip_value:
        clipped = old_values + (values - old_values).clamp(-e_clip, e_clip)
        c_rows = torch.max((values - returns) ** 2, (clipped - returns) ** 2)
    else:
        c_rows = (returns - values) ** 2
    if bound_kind == 1:
        b_rows = (torch.clamp_max(mu + 1.1, 0.0) ** 2 + torch.clamp_min(mu - 1.1, 0.0) ** 2).sum(dim=-1)
    elif bound_kind == 2:
        b_rows = (mu * mu).sum(dim=-1)
    else:
        b_rows = torch.zeros_like(neglogp)

    def mean(x):
        # torch_ext.apply_masks (torch_ext.py:157-170): x [mb, k].  Masked: the sum over the valid rows (all k columns)
        # over the NUMBER OF VALID ROWS (>= 1) - for the value loss of V columns that is V times the row mean, as in the
        # reference; unmasked: the mean over all mb * k entries
        if mask is None:
            return x.mean()
        m = mask.reshape(-1, 1)
        return (x * m).sum() / m.sum().clamp(min=1.0)
    a_loss, c_loss = mean(a_rows.unsqueeze(1)), mean(c_rows)
    ent, b_loss = mean(entropy.unsqueeze(1)), mean(b_rows.unsqueeze(1))
    loss = a_loss + 0.5 * c_loss * critic_coef - ent * entropy_coef + b_loss * bounds_coef
    scalars = {'a_loss': a_loss.detach(), 'c_loss': c_loss.detach(), 'entropy': ent.detach(), 'b_loss': b_loss.detach()}
    return loss, scalars, sigma


def policy_kl(mu, sigma, old_mu, old_sigma, mask=None):
    """torch_ext.policy_kl(p0 = new, p1 = old) and its masked mean (torch_ext.py:27-36, a2c_continuous.py:215-221)."""
    rows = (torch.log(old_sigma / sigma + 1e-5) + (sigma ** 2 + (old_mu - mu) ** 2) / (2.0 * (old_sigma ** 2 + 1e-5))
            - 0.5).sum(dim=-1)
    if mask is None:
        return rows.mean()
    m = mask.reshape(-1)
    return (rows * m).sum() / m.sum().clamp(min=1.0)


def normalize_advantages(adv, mask=None):
    """a2c_common.py:1634 (torch.std: unbiased) / torch_ext.normalization_with_masks + get_mean_var_with_masks
    (torch_ext.py:172-191: unbiased as well, denominators clamped to >= 1)."""
    if mask is None:
        return (adv - adv.mean()) / (adv.std() + 1e-8)
    m = mask.reshape(-1).to(adv.dtype)
    n = m.sum().clamp(min=1.0)
    am = adv * m
    mean = am.sum() / n
    second = ((am ** 2) / n).sum() - ((am / n).sum()) ** 2
    var = second * n / (n - 1).clamp(min=1.0)
    return (adv - mean) / (torch.sqrt(var) + 1e-8)
