"""rl_games_amd - the rl_games PPO hot path (rollout buffer, GAE, running statistics,
clipped-PPO loss, optimiser step) as hand-written gfx950 HIP kernels behind the rl_games
A2CAgent / Runner API.  MI355X only; there is no CPU fallback."""

__version__ = '0.1.0'
