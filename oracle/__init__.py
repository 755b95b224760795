"""TEST INFRASTRUCTURE ONLY: CPU oracle for the rl_games PPO hot path (see ppo_oracle.py)."""
