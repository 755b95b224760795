"""TEST INFRASTRUCTURE ONLY - seeded synthetic inputs shared by tests/, the golden-fixture
generator and bench.py's parity checks."""
import torch


def gae_inputs(horizon, num_envs, value_size, seed=0, p_done=0.15):
    """Seeded GAE inputs, generated exactly like the reference's own test factory
    (tests/test_triton_gae.py:45-52): one CPU generator, draw order rewards, values, dones,
    last_values, last_dones."""
    g = torch.Generator().manual_seed(seed)
    rewards = torch.randn(horizon, num_envs, value_size, generator=g)
    values = torch.randn(horizon, num_envs, value_size, generator=g)
    dones = (torch.rand(horizon, num_envs, generator=g) < p_done).float()
    last_values = torch.randn(num_envs, value_size, generator=g)
    last_dones = (torch.rand(num_envs, generator=g) < p_done).float()
    return rewards, values, dones, last_values, last_dones
