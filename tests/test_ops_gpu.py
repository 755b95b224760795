"""GPU parity of the rollout-buffer, running-statistics, PPO-loss and optimiser kernels
(through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances: integer / index / copy semantics bit-exact; element-wise fp32 bit-exact where the
kernel performs the same op chain; results that pass through a reduction (means, variances,
norms) within rtol 1e-5 (north_star) - the kernels reduce in fp64, torch in fp32."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
RTOL = 1e-5


def g(seed):
    return torch.Generator().manual_seed(seed)


# ----------------------------------------------------------------------------- rollout buffer

@pytest.mark.parametrize('N,H,O_,A', [(300, 8, 108, 21), (64, 4, 3, 1), (1000, 16, 60, 8)])
def test_store_step_matches_indexed_assignment(N, H, O_, A):
    from rl_games_amd import ops
    gen = g(0)
    fields = {'obses': (O_,), 'actions': (A,), 'mus': (A,), 'sigmas': (A,), 'neglogpacs': (),
              'values': (1,)}
    phys = {k: torch.zeros((N, H) + s, device=DEV) for k, s in fields.items()}
    phys['dones'] = torch.zeros((N, H), dtype=torch.uint8, device=DEV)
    ref = {k: torch.zeros((H, N) + s) for k, s in fields.items()}
    ref['dones'] = torch.zeros((H, N), dtype=torch.uint8)
    for step in (0, 3, H - 1):
        vals = {k: torch.randn((N,) + s, generator=gen) for k, s in fields.items()}
        vals['dones'] = (torch.rand(N, generator=gen) < 0.3).to(torch.uint8)
        for k, v in vals.items():
            ref[k][step, :] = v                                   # experience.py:456
        ops.rollout_store_step([(v.to(DEV), phys[k]) for k, v in vals.items()], N, H, step)
    for k in ref:
        view = phys[k].transpose(0, 1)                            # the [H, N, ...] API view
        assert torch.equal(view.cpu(), ref[k]), k
        # swap_and_flatten01 of the view is the physical storage itself (no copy)
        flat = O.flatten_env_major(view)
        assert flat.data_ptr() == phys[k].data_ptr()
        assert torch.equal(flat.cpu(), O.flatten_env_major(ref[k]))


@pytest.mark.parametrize('V,masked,to_dtype,agents', [(1, False, torch.bool, 1), (2, False, torch.float32, 1),
                                                       (1, True, torch.uint8, 1), (1, False, None, 1),
                                                       (1, False, torch.bool, 2), (2, False, None, 3)])
def test_post_step_matches_oracle(V, masked, to_dtype, agents):
    """agents > 1: rows are (env, agent); the agents of an env finish together and the meters
    take all_done_indices[::num_agents] (a2c_common.py:1040-1044), i.e. one row per env."""
    from rl_games_amd import ops
    N, H, gamma = (777, 6, 0.99) if agents == 1 else (258 * agents, 6, 0.99)
    gen = g(1)
    shaper = (0.5, 2.0, -3.0, 3.0)
    nb = ops.post_step_num_blocks(N)
    rewards_buf = torch.zeros(N, H, V, device=DEV)
    cur_r = torch.randn(N, V, generator=gen)
    cur_s = torch.randn(N, V, generator=gen)
    cur_l = torch.randint(0, 50, (N,), generator=gen).float()
    d_cur = [t.clone().to(DEV) for t in (cur_r, cur_s, cur_l)]
    ep_partials = torch.zeros(H, nb, 2 * V + 2, dtype=torch.float64, device=DEV)
    ref_buf = torch.zeros(H, N, V)
    meters = {'m': [torch.zeros(V), torch.zeros(V), torch.zeros(1)], 'n': [0, 0, 0]}
    max_size = 100
    for step in range(H):
        rewards = torch.randn(N, V, generator=gen)
        values = torch.randn(N, V, generator=gen)
        dones = (torch.rand(N // agents, generator=gen) < 0.2).to(torch.uint8).repeat_interleave(agents)
        time_outs = (torch.rand(N, generator=gen) < 0.5) & dones.bool()
        live = (torch.rand(N, generator=gen) < 0.8).float() if masked else None
        shaped = O.shape_rewards(rewards, scale=shaper[1], shift=shaper[0], min_val=shaper[2],
                                 max_val=shaper[3])
        if to_dtype is not None:
            shaped = O.bootstrap_timeouts(shaped, values, time_outs, gamma)
        ref_buf[step, :] = shaped
        cur_r, cur_s, cur_l, fin = O.episode_bookkeeping(cur_r, cur_s, cur_l, rewards, shaped, dones,
                                                         live_rows=live, num_agents=agents)
        for j, vals in enumerate(fin[:3]):
            vals = vals.reshape(vals.shape[0], -1) if j < 2 else vals.reshape(-1, 1)
            meters['m'][j], meters['n'][j] = O.average_meter_update(meters['m'][j], meters['n'][j],
                                                                    vals, max_size)
        to = None if to_dtype is None else time_outs.to(to_dtype).to(DEV)
        ops.rollout_post_step(rewards.to(DEV), dones.to(DEV), to, values.to(DEV),
                              None if live is None else live.to(DEV), rewards_buf, d_cur[0], d_cur[1],
                              d_cur[2], ep_partials, shaper, to_dtype is not None, gamma, H, step,
                              num_agents=agents)
        assert torch.equal(d_cur[0].cpu(), cur_r) and torch.equal(d_cur[1].cpu(), cur_s)
        assert torch.equal(d_cur[2].cpu(), cur_l)
    assert torch.equal(rewards_buf.transpose(0, 1).cpu(), ref_buf)
    mr, ms, ml = torch.zeros(V, device=DEV), torch.zeros(V, device=DEV), torch.zeros(1, device=DEV)
    sizes = torch.zeros(3, dtype=torch.int32, device=DEV)
    fin_total = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.episode_meters_update(ep_partials, H, nb, V, max_size, mr, ms, ml, sizes, fin_total)
    assert sizes.tolist() == meters['n']
    for dev_t, ref_t in zip((mr, ms, ml), meters['m']):
        assert torch.allclose(dev_t.cpu(), ref_t.reshape(-1), rtol=RTOL, atol=1e-6)


def test_post_step_log_val_shaper():
    """reward_shaper log_val (tr_helpers.py:40-41): log of the shifted, scaled, clamped reward - inside the post-step
    kernel (round 4; NotImplementedError before).  logf against torch.log: 1e-6 relative; a clamp floor > 0 keeps the
    argument positive like every configuration that uses the option."""
    from rl_games_amd import ops
    N, H, V = 500, 3, 1
    gen = g(4)
    shaper = (2.0, 0.5, 0.05, 10.0, True)
    nb = ops.post_step_num_blocks(N)
    rewards_buf = torch.zeros(N, H, V, device=DEV)
    cur = [torch.zeros(N, V, device=DEV), torch.zeros(N, V, device=DEV), torch.zeros(N, device=DEV)]
    ep_partials = torch.zeros(H, nb, 2 * V + 2, dtype=torch.float64, device=DEV)
    for step in range(H):
        rewards = torch.randn(N, V, generator=gen) * 3
        dones = (torch.rand(N, generator=gen) < 0.2).to(torch.uint8)
        want = O.shape_rewards(rewards, scale=shaper[1], shift=shaper[0], min_val=shaper[2], max_val=shaper[3], log_val=True)
        ops.rollout_post_step(rewards.to(DEV), dones.to(DEV), None, torch.zeros(N, V, device=DEV), None, rewards_buf,
                              cur[0], cur[1], cur[2], ep_partials, shaper, False, 0.99, H, step)
        got = rewards_buf[:, step].cpu()
        assert torch.isfinite(got).all()
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)


# ----------------------------------------------------------------------------- RunningMeanStd

@pytest.mark.parametrize('rows,C', [(4096, 108), (1000, 60), (777, 3), (5000, 1), (300, 260), (64, 7)])
def test_running_mean_std_update_and_normalise(rows, C):
    from rl_games_amd import ops
    gen = g(2)
    state = O.new_running_stats(C)
    dmean = state['running_mean'].clone().to(DEV)
    dvar = state['running_var'].clone().to(DEV)
    dcount = state['count'].clone().reshape(1).to(DEV)
    for it in range(3):
        x = torch.randn(rows, C, generator=gen) * (1 + it) + 3.0 * it - 1.0
        y_ref, state = O.running_stats_forward(state, x, training=True)
        xd = x.to(DEV)
        part, nb = ops.column_moments(xd)
        ops.rms_update(part, nb, C, rows, 0, dmean, dvar, dcount)
        y = ops.rms_apply(xd, dmean, dvar, 1e-5, 0)
        assert dcount.item() == state['count'].item()
        assert torch.allclose(dmean.cpu(), state['running_mean'], rtol=1e-6, atol=1e-7)
        assert torch.allclose(dvar.cpu(), state['running_var'], rtol=2e-6, atol=1e-9)
        assert torch.allclose(y.cpu(), y_ref, rtol=RTOL, atol=2e-6)
    # eval-mode normalise and de-normalise use the state only.  The kernel's divide and sqrt are
    # correctly rounded; torch's CPU sqrt is not on every host (AVX-512 boxes differ by 1 ulp),
    # so compare to 2 ulp instead of bitwise, and bitwise against an fp64-evaluated reference.
    state_dev = {'running_mean': dmean.cpu(), 'running_var': dvar.cpu(), 'count': dcount.cpu()[0]}
    x = torch.randn(rows, C, generator=gen) * 4
    y_ref, _ = O.running_stats_forward(state_dev, x, training=False)
    y = ops.rms_apply(x.to(DEV), dmean, dvar, 1e-5, 0).cpu()
    assert torch.allclose(y, y_ref, rtol=3e-7, atol=1e-7)
    m32, v32 = state_dev['running_mean'].float(), state_dev['running_var'].float()
    den = torch.sqrt((v32 + 1e-5).double()).float()              # correctly rounded sqrt
    exact = torch.clamp(((x - m32).double() / den.double()).float(), -5.0, 5.0)
    assert torch.equal(y, exact)
    y_ref, _ = O.running_stats_forward(state_dev, x, training=False, denorm=True)
    yd = ops.rms_apply(x.to(DEV), dmean, dvar, 1e-5, 1).cpu()
    assert torch.allclose(yd, y_ref, rtol=1e-6, atol=2e-6)
    assert torch.equal(yd, den * torch.clamp(x, -5.0, 5.0) + m32)


def test_running_mean_std_masked_modes():
    """mode 1 = forward(x, mask=...) (unbiased masked moments, count += rows); mode 2 = the
    value normaliser's `x[valid]` path (a2c_common.py:1609-1611)."""
    from rl_games_amd import ops
    gen = g(3)
    rows = 3000
    x = torch.randn(rows, 1, generator=gen) * 2 + 1
    mask = (torch.rand(rows, generator=gen) < 0.7).float()
    for mode in (1, 2):
        state = O.new_running_stats(1)
        if mode == 1:
            _, ref = O.running_stats_forward(state, x, training=True, mask=mask.unsqueeze(1))
        else:
            _, ref = O.running_stats_forward(state, x[mask.bool()], training=True)
        dmean = state['running_mean'].clone().to(DEV)
        dvar = state['running_var'].clone().to(DEV)
        dcount = state['count'].clone().reshape(1).to(DEV)
        part, nb = ops.column_moments(x.to(DEV), mask.to(DEV))
        ops.rms_update(part, nb, 1, rows, mode, dmean, dvar, dcount)
        assert dcount.item() == ref['count'].item()
        assert torch.allclose(dmean.cpu(), ref['running_mean'], rtol=1e-6, atol=1e-7)
        assert torch.allclose(dvar.cpu(), ref['running_var'], rtol=2e-6)


@pytest.mark.parametrize('N,H', [(1000, 32), (130, 8), (4096, 16)])
@pytest.mark.parametrize('ema', [False, True])
def test_prepare_dataset_from_gae_moments(N, H, ema):
    """GAE kernel -> prepare_finalize -> prepare_apply == oracle prepare_dataset
    (a2c_common.py:1586-1660) on the same rollout tensors."""
    from oracle.seeded_inputs import gae_inputs
    from rl_games_amd import ops
    from rl_games_amd.gae import gae_returns_advantages
    r, v, d, lv, ld = gae_inputs(H, N, 1, seed=4, p_done=0.05)
    v = v * 2 + 5
    advs = O.gae_scan(r, v, d, lv, ld, 0.99, 0.95)
    returns = O.flatten_env_major(O.returns_from_advantages(advs, v))
    values = O.flatten_env_major(v)
    vstate = O.new_running_stats(1)
    ema_state = O.new_moving_stats(1) if ema else None
    ref = O.prepare_dataset(returns, values, vstate, adv_ema_state=ema_state, adv_ema_decay=0.5)

    rp, vp = r[..., 0].t().contiguous().to(DEV), v[..., 0].t().contiguous().to(DEV)
    dp = d.t().contiguous().to(torch.uint8).to(DEV)
    ret, adv, part = gae_returns_advantages(rp, vp, dp, lv[:, 0].contiguous().to(DEV),
                                            ld.to(torch.uint8).to(DEV), 0.99, 0.95)
    vs = (vstate['running_mean'].clone().to(DEV), vstate['running_var'].clone().to(DEV),
          vstate['count'].clone().reshape(1).to(DEV))
    stats = ops.prepare_stats_buffer(DEV)
    flags = ops.PREP_NORM_VALUE | (ops.PREP_EMA_ADV if ema else ops.PREP_NORM_ADV)
    ema_dev = None
    if ema:
        ema_dev = {'mean': torch.zeros(1, device=DEV), 'sqrs': torch.zeros(1, device=DEV),
                   'step': torch.ones(1, dtype=torch.int32, device=DEV), 'decay': 0.5, 'max': 1e5, 'eps': 0.0}
    ops.prepare_finalize(part, N * H, flags, vs, 1e-5, ema_dev, stats)
    vals_n = vp.clone()
    ops.prepare_apply(vals_n, ret, adv, flags, stats)
    assert vs[2].item() == ref['value_stats']['count'].item()
    assert torch.allclose(vs[0].cpu(), ref['value_stats']['running_mean'], rtol=1e-6)
    assert torch.allclose(vs[1].cpu(), ref['value_stats']['running_var'], rtol=2e-6)
    assert torch.allclose(vals_n.reshape(-1, 1).cpu(), ref['old_values'], rtol=RTOL, atol=2e-6)
    assert torch.allclose(ret.reshape(-1, 1).cpu(), ref['returns'], rtol=RTOL, atol=2e-6)
    assert torch.allclose(adv.reshape(-1).cpu(), ref['advantages'], rtol=RTOL, atol=2e-6)
    if ema:
        assert ema_dev['step'].item() == ref['adv_ema_state']['step'].item()
        assert torch.allclose(ema_dev['mean'].cpu(), ref['adv_ema_state']['mean'], rtol=RTOL, atol=1e-7)
        assert torch.allclose(ema_dev['sqrs'].cpu(), ref['adv_ema_state']['sqrs'], rtol=RTOL)


# ----------------------------------------------------------------------------- PPO loss

def _loss_inputs(mb, A, seed):
    """SURVEY 8d kernel-level inputs: both clip sides fire, bound loss active."""
    gen = g(seed)
    old_nlp = torch.randn(mb, generator=gen)
    batch = {
        'old_logp_actions': old_nlp,
        'advantages': torch.randn(mb, generator=gen),
        'old_values': torch.randn(mb, 1, generator=gen),
        'returns': torch.randn(mb, 1, generator=gen),
        'actions': torch.randn(mb, A, generator=gen),
        'mu': 1.2 * torch.randn(mb, A, generator=gen),
        'sigma': torch.exp(0.1 * torch.randn(mb, A, generator=gen)),
    }
    mu = batch['mu'] + 0.1 * torch.randn(mb, A, generator=gen)
    logstd = 0.1 * torch.randn(A, generator=gen)
    values = batch['old_values'] + 0.3 * torch.randn(mb, 1, generator=gen)
    # make neglogp land near old_neglogp so that ratios straddle the clip range
    with torch.no_grad():
        sigma = torch.exp(logstd)
        nlp = O.neglogp(batch['actions'], mu, mu * 0 + sigma, mu * 0 + logstd)
        batch['old_logp_actions'] = nlp + 0.2 * torch.randn(mb, generator=gen)
    return mu, logstd, values, batch


@pytest.mark.parametrize('mb,A', [(4096, 21), (1000, 8), (300, 1), (257, 33), (32768, 21)])
@pytest.mark.parametrize('variant', ['bound', 'smooth_reg', 'noclip_nobound', 'masked', 'ppo_false'])
def test_ppo_loss_forward_backward_kl(mb, A, variant):
    from rl_games_amd import ops
    mu, logstd, values, batch = _loss_inputs(mb, A, seed=mb + A)
    hp = {'e_clip': 0.2, 'critic_coef': 2.0, 'entropy_coef': 0.01, 'bounds_loss_coef': 1e-4,
          'clip_value': True, 'use_smooth_clamp': False, 'bound_loss_type': 'bound'}
    mask = None
    if variant == 'smooth_reg':
        hp.update(use_smooth_clamp=True, bound_loss_type='regularisation', bounds_loss_coef=0.01)
    elif variant == 'noclip_nobound':
        hp.update(clip_value=False, bounds_loss_coef=None)
    elif variant == 'masked':
        mask = (torch.rand(mb, generator=g(7)) < 0.8).float()
    elif variant == 'ppo_false':
        hp.update(ppo=False)                # a_loss = neglogp * advantage (common_losses.py:59, 80): surrogate kind 2
    ref = O.distribution_loss_and_grads(mu, logstd, values, batch, hp, mask)

    d = lambda t: t.contiguous().to(DEV)
    old_mu, old_sigma = d(batch['mu']), d(batch['sigma'])
    d_mu = torch.empty(mb, A, device=DEV)
    d_val = torch.empty(mb, device=DEV)
    nb = ops.ppo_loss_blocks(mb)
    partials = torch.empty(nb, ops.ppo_loss_partials_per_block(A), dtype=torch.float64, device=DEV)
    scalars = torch.zeros(8, device=DEV)
    d_logstd = torch.zeros(A, device=DEV)
    kl_slot = torch.zeros(1, device=DEV)
    coef_b = hp['bounds_loss_coef'] if hp['bounds_loss_coef'] is not None else 0.0
    kind = 0 if hp['bounds_loss_coef'] is None else ops.BOUND_KINDS[hp['bound_loss_type']]
    mask_d = None if mask is None else d(mask)
    mask_sum = None if mask is None else mask_d.sum().reshape(1)
    ops.ppo_loss_fused(d(mu), d(logstd), d(values.reshape(-1)), d(batch['actions']),
                       d(batch['old_logp_actions']), d(batch['advantages']),
                       d(batch['old_values'].reshape(-1)), d(batch['returns'].reshape(-1)), old_mu,
                       old_sigma, d_mu, d_val, partials, hp['e_clip'], hp['critic_coef'], coef_b,
                       hp['clip_value'], ops.SURROGATE_NONE if not hp.get('ppo', True) else hp['use_smooth_clamp'], kind, True,
                       mask_d, mask_sum)
    ops.ppo_loss_finalize(partials, nb, A, mb, mask is not None, hp['critic_coef'], hp['entropy_coef'],
                          coef_b, scalars, d_logstd, kl_slot)
    s = scalars.cpu()
    # Scalar losses are means of O(1) terms of both signs; the reference sums them in fp32, the
    # kernel in fp64.  Ground truth = the oracle evaluated in fp64: the kernel must agree with it
    # to rtol 1e-5 (atol 1e-7), and with the fp32 oracle to within the oracle's own rounding.
    to64 = lambda t: t.double() if torch.is_floating_point(t) else t
    truth = O.distribution_loss_and_grads(to64(mu), to64(logstd), to64(values),
                                          {k: to64(v) for k, v in batch.items()}, hp,
                                          None if mask is None else mask.double())
    for k, name in enumerate(('a_loss', 'c_loss', 'entropy', 'b_loss', 'kl', 'loss')):
        t64 = truth[name].item()
        own_err = abs(ref[name].item() - t64)
        assert abs(s[k].item() - t64) <= RTOL * abs(t64) + 2e-7, (name, s[k].item(), t64)
        assert abs(s[k].item() - ref[name].item()) <= RTOL * abs(t64) + 2e-7 + 2 * own_err, \
            (name, s[k].item(), ref[name].item(), t64)
    assert kl_slot.item() == s[4].item()
    # Gradients.  ratio = exp(old_nlp - nlp) with |nlp| ~ 0.5*A*z^2: the fp32 rounding of the
    # exponent (~1e-6 absolute) is a ~1e-5 RELATIVE perturbation of a few rows' gradients in any
    # fp32 implementation, the reference included (x(1 + 10*ratio) more under smooth_clamp,
    # whose derivative contains exp(-10*ratio)).  Criterion: as close to the fp64 truth as the
    # fp32 oracle is (max error within 8x of the oracle's own max error), and element-wise within 5e-5 of the fp32 oracle for >= 99.5 % of the entries (2e-3 for all).
    scale = 1.0 / mb
    for got, key in ((d_mu.cpu(), 'd_mu'), (d_val.cpu().reshape(-1, 1), 'd_values'),
                     (d_logstd.cpu(), 'd_logstd')):
        r32, r64 = ref[key], truth[key]
        atol = 1e-6 * r32.abs().max().item()     # sums of cancelling terms: floor relative to the tensor's scale
        assert torch.allclose(got, r32, rtol=2e-3, atol=atol), key
        outliers = ((got - r32).abs() > 5e-5 * r32.abs() + atol).float().mean().item()
        assert key == 'd_logstd' or outliers <= 5e-3, (key, outliers)   # d_logstd: A sums, atol only
        err_kernel = (got.double() - r64).abs().max().item()
        err_oracle = (r32.double() - r64).abs().max().item()
        # (+ 1e-6 of the tensor's scale: with ppo False the loss is linear in neglogp, the fp32 oracle lands within 1e-10
        #  of the fp64 value and "8 x the oracle's error" would ask the kernel for more than fp32 products can give)
        assert err_kernel <= 8 * err_oracle + 1e-9 * scale + 1e-6 * r64.abs().max().item(), (key, err_kernel, err_oracle)
    # update_mu_sigma write-back (datasets.py:42-43): bit-exact copies of the new policy
    assert torch.equal(old_mu.cpu(), mu)
    # sigma = exp(logstd): device expf and the host's exp may differ in the last bit
    assert torch.allclose(old_sigma.cpu(), ref['sigma'], rtol=3e-7, atol=0)


def test_ppo_loss_max_tie_and_clip_edges():
    """Rows constructed to sit exactly on the torch.max tie / clamp boundary."""
    from rl_games_amd import ops
    mb, A = 512, 4
    mu, logstd, values, batch = _loss_inputs(mb, A, seed=99)
    with torch.no_grad():
        sigma = torch.exp(logstd)
        nlp = O.neglogp(batch['actions'], mu, mu * 0 + sigma, mu * 0 + logstd)
    batch['old_logp_actions'] = nlp.clone()            # ratio == 1 exactly: both branches tie
    batch['advantages'][::7] = 0.0                     # zero advantage: both branches 0
    values = batch['old_values'].clone()               # delta == 0: value branches tie
    values[::5] += 0.2                                 # delta == e_clip (inclusive edge, in fp32 ~)
    hp = {'e_clip': 0.2, 'critic_coef': 1.0, 'entropy_coef': 0.0, 'bounds_loss_coef': 1e-4,
          'clip_value': True, 'use_smooth_clamp': False, 'bound_loss_type': 'bound'}
    ref = O.distribution_loss_and_grads(mu, logstd, values, batch, hp, None)
    d = lambda t: t.contiguous().to(DEV)
    d_mu, d_val = torch.empty(mb, A, device=DEV), torch.empty(mb, device=DEV)
    nb = ops.ppo_loss_blocks(mb)
    partials = torch.empty(nb, ops.ppo_loss_partials_per_block(A), dtype=torch.float64, device=DEV)
    ops.ppo_loss_fused(d(mu), d(logstd), d(values.reshape(-1)), d(batch['actions']),
                       d(batch['old_logp_actions']), d(batch['advantages']),
                       d(batch['old_values'].reshape(-1)), d(batch['returns'].reshape(-1)),
                       d(batch['mu']), d(batch['sigma']), d_mu, d_val, partials, 0.2, 1.0, 1e-4, True,
                       False, 1, False)
    assert torch.allclose(d_mu.cpu(), ref['d_mu'], rtol=5e-5, atol=1e-10)
    assert torch.allclose(d_val.cpu(), ref['d_values'].reshape(-1), rtol=RTOL, atol=1e-10)


# ----------------------------------------------------------------------------- optimiser

@pytest.mark.parametrize('truncate', [True, False])
def test_adam_step_matches_torch_adam(truncate):
    from rl_games_amd import ops
    gen = g(5)
    shapes = [(400, 108), (400,), (200, 400), (200,), (21,), (1, 100)]
    params = [torch.randn(s, generator=gen) * 0.1 for s in shapes]
    exp_avg = [torch.randn(s, generator=gen) * 0.01 for s in shapes]
    exp_avg_sq = [torch.rand(s, generator=gen) * 1e-3 for s in shapes]
    n = sum(p.numel() for p in params)
    flat = lambda ts: torch.cat([t.reshape(-1) for t in ts]).to(DEV)
    fp, fm, fv = flat(params), flat(exp_avg), flat(exp_avg_sq)
    lr_slots = torch.tensor([3e-4, 0.0], dtype=torch.float64, device=DEV)
    stats = torch.zeros(4, device=DEV)
    step_counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    cur = 0
    lr = 3e-4
    for step in range(1, 4):
        grads = [torch.randn(s, generator=gen) * (3.0 if step == 1 else 0.01) for s in shapes]
        kl = 0.02 if step == 1 else (0.001 if step == 2 else 0.008)
        p_ref, m_ref, v_ref, norm_ref = O.clip_and_adam_reference(params, grads, exp_avg, exp_avg_sq,
                                                                 step - 1, lr, 1.0, truncate)
        fg = flat(grads)
        partials = torch.empty(ops.grad_norm_blocks(n), dtype=torch.float64, device=DEV)
        ops.grad_sumsq(fg, 1.0, partials, step_counter)          # advances the device step counter
        assert step_counter.item() == step
        ops.adam_step(fp, fg, fm, fv, partials if truncate else None, 1.0, 1.0, lr_slots, step_counter,
                      schedule_kind=1, kl=torch.tensor([kl], device=DEV), stats_out=stats)
        params, exp_avg, exp_avg_sq = p_ref, m_ref, v_ref
        if truncate:
            assert np.isclose(stats[0].item(), norm_ref.item(), rtol=1e-6)
        assert np.isclose(stats[2].item(), lr, rtol=1e-7)
        lr = O.adaptive_lr(lr, kl)
        cur ^= 1
        assert lr_slots[cur].item() == lr               # python-double arithmetic, bit-exact
        assert torch.allclose(fp.cpu(), flat(params).cpu(), rtol=1e-6, atol=1e-8)
        assert torch.allclose(fm.cpu(), flat(exp_avg).cpu(), rtol=1e-6, atol=1e-9)
        assert torch.allclose(fv.cpu(), flat(exp_avg_sq).cpu(), rtol=1e-6, atol=1e-12)


def test_adam_multi_gpu_average_and_kl_scale():
    """grad_scale = 1/world (a2c_common.py:505-507) and kl_scale for a cross-rank KL sum."""
    from rl_games_amd import ops
    gen = g(6)
    n = 5000
    p0 = torch.randn(n, generator=gen)
    g_sum = torch.randn(n, generator=gen)          # what all_reduce(SUM) over 4 ranks leaves
    ref_p, _, _, _ = O.clip_and_adam_reference([p0], [g_sum / 4], [torch.zeros(n)], [torch.zeros(n)],
                                               0, 1e-3, 1.0, True)
    fp, fg = p0.clone().to(DEV), g_sum.clone().to(DEV)
    fm, fv = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    partials = torch.empty(ops.grad_norm_blocks(n), dtype=torch.float64, device=DEV)
    step_counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.grad_sumsq(fg, 0.25, partials, step_counter)
    lr_slots = torch.tensor([1e-3, 0.0], dtype=torch.float64, device=DEV)
    ops.adam_step(fp, fg, fm, fv, partials, 0.25, 1.0, lr_slots, step_counter, schedule_kind=1,
                  kl=torch.tensor([4 * 0.02], device=DEV), kl_scale=0.25)
    assert torch.allclose(fp.cpu(), ref_p[0], rtol=1e-6, atol=1e-8)
    assert lr_slots[1].item() == O.adaptive_lr(1e-3, float(np.float32(0.02 * 4) * np.float32(0.25)))


# ----------------------------------------------------------------------------- manual MLP backward

@pytest.mark.parametrize('rows,C', [(32768, 400), (1000, 200), (777, 100), (64, 8), (5000, 68)])
@pytest.mark.parametrize('act', ['elu', 'relu', 'tanh', 'None'])
def test_act_bwd_colsum_matches_autograd(rows, C, act):
    """d_pre = d_out * act'(z), db = sum_rows d_pre  ==  aten's activation backward followed by the
    bias-gradient sum(0) of nn.Linear's backward."""
    from rl_games_amd import ops
    gen = g(rows + C)
    z = torch.randn(rows, C, generator=gen) * 2
    d_out = torch.randn(rows, C, generator=gen)
    fn = {'elu': torch.nn.functional.elu, 'relu': torch.relu, 'tanh': torch.tanh, 'None': lambda t: t}[act]
    zz = z.clone().requires_grad_(True)
    fn(zz).backward(d_out)
    ref = zz.grad
    dz = d_out.clone().to(DEV)
    nb = ops.act_bwd_blocks(rows, C)
    part = torch.empty(nb * C, dtype=torch.float64, device=DEV)
    ops.act_bwd_colsum(dz, z.to(DEV), dz, ops.ACT_KINDS[act], part, nb)
    db = torch.zeros(C, device=DEV)
    ops.colsum_finalize(part, nb, C, db)
    # exp / tanh differ in the last bits between the device and the host libm
    assert torch.allclose(dz.cpu(), ref, rtol=2e-5 if act == 'tanh' else 3e-7, atol=2e-6 if act == 'tanh' else 1e-7)
    truth = ref.double().sum(0)
    scale = ref.abs().double().sum(0)
    assert ((db.cpu().double() - truth).abs() <= 1e-6 * scale + 1e-9).all()


def test_ppo_loss_on_strided_head_views_and_bias_grads():
    """mu / values / d_mu / d_values as column views of one fused [mb, 1+A] buffer give the same
    results as the contiguous call; the extra partial columns are the head-bias gradients."""
    from rl_games_amd import ops
    mb, A = 3000, 21
    mu, logstd, values, batch = _loss_inputs(mb, A, seed=5)
    d = lambda t: t.contiguous().to(DEV)
    args = (d(batch['actions']), d(batch['old_logp_actions']), d(batch['advantages']),
            d(batch['old_values'].reshape(-1)), d(batch['returns'].reshape(-1)))
    nb, W = ops.ppo_loss_blocks(mb), ops.ppo_loss_partials_per_block(A)
    # contiguous
    d_mu, d_val = torch.empty(mb, A, device=DEV), torch.empty(mb, device=DEV)
    p1 = torch.empty(nb, W, dtype=torch.float64, device=DEV)
    ops.ppo_loss_fused(d(mu), d(logstd), d(values.reshape(-1)), *args, d(batch['mu']), d(batch['sigma']),
                       d_mu, d_val, p1, 0.2, 2.0, 1e-4, True, False, 1, False)
    # fused head buffer
    heads = torch.cat([values, mu], dim=1).to(DEV)
    d_heads = torch.zeros(mb, 1 + A, device=DEV)
    p2 = torch.empty(nb, W, dtype=torch.float64, device=DEV)
    ops.ppo_loss_fused(heads[:, 1:], d(logstd), heads[:, 0], *args, d(batch['mu']), d(batch['sigma']),
                       d_heads[:, 1:], d_heads[:, 0], p2, 0.2, 2.0, 1e-4, True, False, 1, False)
    assert torch.equal(d_heads[:, 1:], d_mu) and torch.equal(d_heads[:, 0], d_val)
    assert torch.equal(p1, p2)
    scal, dls = torch.zeros(8, device=DEV), torch.zeros(A, device=DEV)
    bmu, bval = torch.zeros(A, device=DEV), torch.zeros(1, device=DEV)
    ops.ppo_loss_finalize(p2, nb, A, mb, False, 2.0, 0.0, 1e-4, scal, dls, None, bmu, bval)
    assert torch.allclose(bmu.cpu(), d_mu.cpu().double().sum(0).float(), rtol=1e-5, atol=1e-9)
    assert torch.allclose(bval.cpu(), d_val.cpu().double().sum().float().reshape(1), rtol=1e-5, atol=1e-9)


# ----------------------------------------------------------------------------- fp32-MFMA MLP kernels

@pytest.mark.parametrize('rows', [32768, 4096, 1000, 37, 2])
def test_mlp_dw_mfma_matches_library_gemm(rows):
    """All four weight-gradient GEMMs of the BASELINE MLP in one MFMA launch vs torch.mm and an
    fp64 evaluation: the kernel must be as accurate as the library's fp32 result."""
    from rl_games_amd import ops
    g = torch.Generator().manual_seed(rows)
    shapes = [(400, 108), (200, 400), (100, 200), (22, 100)]
    layers, refs = [], []
    for No, Mi in shapes:
        dz = torch.randn(rows, No, generator=g).to(DEV)
        x = torch.randn(rows, Mi, generator=g).to(DEV)
        grad = torch.full((No, Mi), float('nan'), device=DEV)
        layers.append((dz, x, grad))
        refs.append((dz.t() @ x, dz.double().t() @ x.double()))
    plan = ops.MlpDwPlan(shapes, rows, DEV)
    plan.launch(layers)
    for (dz, x, grad), (lib32, t64) in zip(layers, refs):
        assert torch.isfinite(grad).all()
        err = (grad.double() - t64).abs().max().item()
        err_lib = (lib32.double() - t64).abs().max().item()
        scale = t64.abs().max().item()
        assert err <= max(4 * err_lib, 1e-6 * scale), (tuple(grad.shape), err, err_lib, scale)
    # bias gradients ride along: column sums of per-block fp64 partials
    cs_items, want = [], []
    for cols, nb in ((400, 37), (100, 256), (33, 1)):
        part = torch.randn(nb * cols, generator=g, dtype=torch.float64).to(DEV)
        out = torch.full((cols,), float('nan'), device=DEV)
        cs_items.append((part, nb, cols, out))
        want.append(part.view(nb, cols).sum(0).float())
    plan.launch(layers, cs_items)
    for (part, nb, cols, out), w in zip(cs_items, want):
        assert torch.allclose(out, w, rtol=1e-6, atol=1e-6), cols
    plan.launch(layers)    # deterministic: same bits on a second launch
    again = [l[2].clone() for l in layers]
    plan.launch(layers)
    assert all(torch.equal(a, l[2]) for a, l in zip(again, layers))


def test_mlp_dw_plan_rejects_unsupported_shapes():
    from rl_games_amd import ops
    with pytest.raises(NotImplementedError):
        ops.MlpDwPlan([(10, 7)], 64, DEV)
    plan = ops.MlpDwPlan([(8, 12)], 64, DEV)
    with pytest.raises(ValueError):
        plan.launch([(torch.zeros(64, 8, device=DEV), torch.zeros(64, 16, device=DEV), torch.zeros(8, 12, device=DEV))])


@pytest.mark.parametrize('rows,N,K,act', [(32768, 400, 108, 'elu'), (4096, 200, 400, 'elu'), (1000, 100, 200, 'tanh'),
                                          (333, 22, 100, 'None'), (65, 64, 4, 'relu'), (130, 400, 108, 'elu')])
def test_mlp_rowgemm_forward_and_backward_match_torch(rows, N, K, act):
    """LDS-free f32-MFMA Linear+activation forward and (dZ W)*act'(Z) backward vs torch and fp64."""
    from rl_games_amd import ops
    g = torch.Generator().manual_seed(rows + N + K)
    x = torch.randn(rows, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    kind = ops.ACT_KINDS[act]
    fn = {'elu': torch.nn.functional.elu, 'tanh': torch.tanh, 'relu': torch.relu, 'None': lambda t: t}[act]
    z = torch.full((rows, N), float('nan'), device=DEV)
    hh = torch.full((rows, N), float('nan'), device=DEV)
    ops.mlp_linear_act_forward(x, w, b, hh, pre_act=z, act_kind=kind)
    z64 = torch.addmm(b.double(), x.double(), w.double().t())
    z32 = torch.addmm(b, x, w.t())
    err, err_lib = (z.double() - z64).abs().max().item(), (z32.double() - z64).abs().max().item()
    assert err <= max(4 * err_lib, 1e-6 * z64.abs().max().item()), (err, err_lib)
    assert torch.allclose(hh, fn(z), rtol=1e-6, atol=1e-7)          # activation of the kernel's own Z
    h2 = torch.full((rows, N), float('nan'), device=DEV)
    ops.mlp_linear_act_forward(x, w, b, h2, act_kind=kind)           # inference form (no Z)
    assert torch.equal(h2, hh)
    # backward: d_prev = (dz @ w) * act'(z_prev) with dz [rows, N], w [N, K], z_prev [rows, K]
    if N % 4 == 0:
        dz = torch.randn(rows, N, generator=g).to(DEV)
        zp = torch.randn(rows, K, generator=g).to(DEV)
        out = torch.full((rows, K), float('nan'), device=DEV)
        ops.mlp_linear_act_backward(dz, w, zp, out, act_kind=kind)
        zp_ = zp.double().requires_grad_(True)
        fn(zp_).backward(dz.double() @ w.double())
        t64 = zp_.grad
        lib = (dz @ w) * torch.autograd.grad(fn(zp.requires_grad_(True)).sum(), zp)[0] if act != 'None' else dz @ w
        e, e_lib = (out.double() - t64).abs().max().item(), (lib.double() - t64).abs().max().item()
        assert e <= max(4 * e_lib, 1e-6 * t64.abs().max().item()), (e, e_lib)
        out2 = torch.empty(rows, K, device=DEV)
        ops.mlp_linear_act_backward(dz, w, None, out2)
        e2 = (out2.double() - dz.double() @ w.double()).abs().max().item()
        assert e2 <= max(4 * ((dz @ w).double() - dz.double() @ w.double()).abs().max().item(), 1e-6)


# ----------------------------------------------------------------------------- cross-rank statistics sync (a19)

def _gpu_stats(dims, seed):
    from rl_games_amd.normalizers import RunningMeanStd
    gen = g(seed)
    mods = []
    for d in dims:
        m = RunningMeanStd((d,)).to(DEV)
        m.running_mean.copy_(torch.randn(d, generator=gen, dtype=torch.float64) * 3)
        m.running_var.copy_(torch.rand(d, generator=gen, dtype=torch.float64) * 5 + 0.01)
        m.count.fill_(int(torch.randint(1, 10 ** 7, (1,), generator=gen)))
        mods.append(m)
    return mods


class _CpuStats:
    def __init__(self, m):
        self.running_mean, self.running_var, self.count = m.running_mean.cpu().clone(), m.running_var.cpu().clone(), m.count.cpu().clone()


@pytest.mark.parametrize('dims', [(108, 1), (3,), (60, 1, 17, 1), (1000,)])
def test_stats_sync_kernels_bit_exact_vs_oracle(dims):
    """rlg_stats_sync_pack / _apply (all normalisers in one flat fp64 buffer) against the oracle's restatement of
    a2c_common.py:43-93, :124-141: three epochs of pooled merges with an emulated 4-rank collective, a seed in
    between, then the broadcast path - statistics, snapshot and counts bit for bit (same fp64 ops, same order)."""
    from rl_games_amd import distributed as rdist
    mods = _gpu_stats(dims, seed=len(dims))
    refs = [_CpuStats(m) for m in mods]
    ours = rdist.StatsSync(mods)
    theirs = rdist.StatsSync(refs, kernels=O.StatsSyncOracle(refs))
    gen = g(11)

    def fake_all_reduce(t):             # 4 ranks: the other three contribute 3.5x of this rank's deltas
        t.mul_(4.5)

    for epoch in range(3):
        for m, r in zip(mods, refs):
            dm = torch.randn(r.running_mean.shape, generator=gen, dtype=torch.float64) * 0.1
            dv = torch.rand(r.running_var.shape, generator=gen, dtype=torch.float64)
            for t in (m, r):
                t.running_mean.add_(dm.to(t.running_mean.device))
                t.running_var.add_(dv.to(t.running_var.device))
                t.count.add_(4096 * (epoch + 1))
        if epoch == 1:
            ours.seed()
            theirs.seed()
        ours.merge(fake_all_reduce)
        theirs.merge(fake_all_reduce)
        assert torch.equal(ours.deltas.cpu(), theirs.deltas) and torch.equal(ours.snapshot.cpu(), theirs.snapshot)
        for m, r in zip(mods, refs):
            assert m.count.item() == r.count.item()
            assert torch.equal(m.running_mean.cpu(), r.running_mean) and torch.equal(m.running_var.cpu(), r.running_var)
    # broadcast mode: "rank 0" hands over a different state
    other = [(torch.randn_like(r.running_mean), torch.rand_like(r.running_var) + 1, 12345 + i) for i, r in enumerate(refs)]

    def fake_broadcast(t):
        off = 0
        for mean, var, n in other:
            d = mean.numel()
            t[off] = float(n)
            t[off + 1:off + 1 + d] = mean.to(t.device)
            t[off + 1 + d:off + 1 + 2 * d] = var.to(t.device)
            off += 1 + 2 * d
    ours.adopt_rank0(fake_broadcast)
    for m, (mean, var, n) in zip(mods, other):
        assert m.count.item() == n and torch.equal(m.running_mean.cpu(), mean) and torch.equal(m.running_var.cpu(), var)


def test_stats_sync_function_seams_and_clamp():
    """The reference's per-module function names stay callable (merge_rank_stats / seed_stats_sync_snapshot /
    broadcast_rank_stats); a variance that cancels below 1e-8 is clamped like `.clamp_(min=1e-8)`."""
    from rl_games_amd import distributed as rdist
    (m,) = _gpu_stats((4,), seed=5)
    m.running_mean.fill_(1e6)
    m.running_var.fill_(1e-12)
    m.count.fill_(1000)
    rdist.merge_rank_stats(m, lambda t: t.mul_(2))
    assert m.count.item() == 2000
    assert torch.all(m.running_var >= 1e-8) and torch.allclose(m.running_mean, torch.full_like(m.running_mean, 1e6))
    before = (m.running_mean.clone(), m.running_var.clone(), m.count.item())
    rdist.seed_stats_sync_snapshot(m)
    rdist.merge_rank_stats(m, lambda t: t.mul_(2))          # nothing new since the seed: deltas are zero
    assert m.count.item() == before[2] and torch.equal(m.running_mean, before[0])
    rdist.broadcast_rank_stats(m, lambda t: t)
    assert m.count.item() == before[2] and torch.equal(m.running_mean, before[0]) and torch.equal(m.running_var, m.running_var)
    with pytest.raises(ValueError):
        rdist.StatsSync([])


# ----------------------------------------------------------------------------- narrow products (config #5 shapes)

@pytest.mark.parametrize('rows,No,Mi', [(16384, 2, 64), (4096, 2, 64), (37, 5, 13), (1, 1, 4), (1000, 8, 100)])
def test_narrow_dx_matches_fp64(rows, No, Mi):
    """rlg_narrow_dx (dX = dZ W for heads with <= 8 outputs: the (value | mu) head of a Pendulum-shaped policy)
    against an fp64 evaluation; at most K fp32 roundings per element."""
    from rl_games_amd import ops
    gen = g(rows + No)
    dz = torch.randn(rows, No, generator=gen)
    w = torch.randn(No, Mi, generator=gen)
    dx = torch.full((rows, Mi), float('nan'), device=DEV)
    ops.narrow_dx(dz.to(DEV), w.to(DEV), dx)
    ref = dz.double() @ w.double()
    scale = (dz.abs().double() @ w.abs().double())
    assert ((dx.cpu().double() - ref).abs() <= (No + 1) * 6e-8 * scale + 1e-30).all()
    with pytest.raises(RuntimeError):
        ops.narrow_dx(torch.zeros(4, 9, device=DEV), torch.zeros(9, 4, device=DEV), torch.zeros(4, 4, device=DEV))


@pytest.mark.parametrize('rows,No,Mi', [(16384, 64, 3), (4096, 64, 3), (37, 20, 7), (1, 4, 1), (70000, 256, 8), (513, 100, 2)])
def test_narrow_dw_matches_fp64_and_is_deterministic(rows, No, Mi):
    """rlg_narrow_dw (grad = dZ^T X for a first layer over <= 8 observations) against fp64: fp32 products, fp32
    sums over a thread's run of rows, fp64 across - far inside rtol 1e-5 of the |dZ|^T |X| scale; two launches agree
    bit for bit (fixed combination order)."""
    from rl_games_amd import ops
    gen = g(rows + Mi)
    dz = torch.randn(rows, No, generator=gen)
    x = torch.randn(rows, Mi, generator=gen) * 2 + 0.3
    g1 = torch.full((No, Mi), float('nan'), device=DEV)
    g2 = torch.full((No, Mi), float('nan'), device=DEV)
    ops.narrow_dw(dz.to(DEV), x.to(DEV), g1)
    ops.narrow_dw(dz.to(DEV), x.to(DEV), g2)
    assert torch.equal(g1, g2)
    ref = dz.double().t() @ x.double()
    scale = dz.abs().double().t() @ x.abs().double()
    assert ((g1.cpu().double() - ref).abs() <= 1e-5 * scale + 1e-30).all()


_HEAD_SCRIPT = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from rl_games_amd import ops
out = {}
for N, A, ld, H, vstats, envact in ((1000, 21, 43, 4, True, True), (64, 3, 4, 2, False, False), (4097, 8, 9, 3, True, False)):
    g = torch.Generator().manual_seed(N + A)
    dev = 'cuda:0'
    heads = torch.randn(N, ld, generator=g).to(dev)
    logstd = (0.3 * torch.randn(A, generator=g)).to(dev)
    noise = torch.randn(N, A, generator=g).to(dev)
    vs = (torch.tensor([0.3], dtype=torch.float64, device=dev), torch.tensor([2.5], dtype=torch.float64, device=dev)) if vstats else None
    storage = {k: torch.zeros(N, H, A, device=dev) for k in ('actions', 'mus', 'sigmas')}
    storage.update({k: torch.zeros(N, H, device=dev) for k in ('neglogpacs', 'values')})
    actions, values = torch.empty(N, A, device=dev), torch.empty(N, device=dev)
    env = None
    if envact:
        env = (torch.empty(N, A, device=dev), -torch.rand(A, generator=g).to(dev) - 0.5, torch.rand(A, generator=g).to(dev) + 0.5)
    for t in range(H):
        ops.rollout_policy_head(heads, logstd, noise, vs, 1e-5, actions, values, storage, H, t, env_actions=env)
    torch.cuda.synchronize()
    out[N] = {k: v.cpu() for k, v in storage.items()}
    out[N].update(actions=actions.cpu(), values=values.cpu())
    if envact:
        out[N]['env'] = env[0].cpu()
torch.save(out, sys.argv[2])
'''


def test_rollout_policy_head_staged_in_lds_gives_the_bits_of_the_direct_form(tmp_path):
    """rlg_rollout_policy_head stages a block's heads / noise tiles in LDS (one coalesced read of each); RLG_ROLLOUT_HEAD_LDS=0
    keeps the direct form (one thread walking an env's row, a second pass over both arrays).  Same expressions in the same
    order: every output bit for bit - ragged row counts, wide and narrow heads, with and without value statistics /
    rescaled env actions."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'head.py'
    script.write_text(_HEAD_SCRIPT)
    outs = {}
    for mode in ('0', '1'):
        path = str(tmp_path / f'out{mode}.pt')
        env = dict(os.environ, RLG_ROLLOUT_HEAD_LDS=mode)
        subprocess.run([sys.executable, str(script), root, path], check=True, env=env, timeout=600)
        outs[mode] = torch.load(path)
    for N, fields in outs['0'].items():
        for k, v in fields.items():
            assert torch.equal(v, outs['1'][N][k]), (N, k)
        assert fields['neglogpacs'].abs().sum() > 0 and torch.isfinite(fields['actions']).all()
