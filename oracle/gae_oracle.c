/*
 * gae_oracle.c - TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the reference's GAE
 * backward scan, used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker for the HIP kernels in rl_games_amd/csrc/gae.hip.  Nothing in the product
 * path (rl_games_amd/) may link or call this file.
 *
 * Follows rl_games/triton_kernels/gae_kernel.py:63-80 (_pytorch_gae), one scalar fp32
 * operation per PyTorch op, in the same order:
 *     nextnonterminal = 1.0 - dones[t+1]            (last_dones at t = H-1)   :69-75
 *     delta = rewards[t] + gamma*nextvalues*nextnonterminal - values[t]        :78
 *     lastgaelam = delta + gamma*tau*nextnonterminal*lastgaelam                :79
 * Python evaluates gamma*tau in double and PyTorch rounds the product to fp32 once; the
 * caller passes that value as `gamma_tau`.  Build with -ffp-contract=off (oracle/Makefile)
 * so no multiply-add is fused.
 *
 * gae_f64_reference follows the independent fp64 scalar recursion the reference's own test
 * uses as ground truth: tests/test_triton_gae.py:20-42 (reference_gae).
 *
 * Layout: time-major contiguous, rewards/values/advs [H][N][V], dones [H][N] (as float),
 * last_values [N][V], last_dones [N] (as float).
 */
#include <stddef.h>
#include <stdint.h>

void gae_f32_scan(const float* rewards, const float* values, const float* dones,
                  const float* last_values, const float* last_dones, float* advs,
                  int horizon, int num_envs, int value_size, float gamma, float gamma_tau) {
  const size_t nv = (size_t)num_envs * (size_t)value_size;
  for (int e = 0; e < num_envs; ++e) {
    for (int k = 0; k < value_size; ++k) {
      const size_t col = (size_t)e * (size_t)value_size + (size_t)k;
      float lastgaelam = 0.0f;
      for (int t = horizon - 1; t >= 0; --t) {
        float nextnonterminal, nextvalues;
        if (t == horizon - 1) {
          nextnonterminal = 1.0f - last_dones[e];
          nextvalues = last_values[col];
        } else {
          nextnonterminal = 1.0f - dones[(size_t)(t + 1) * (size_t)num_envs + (size_t)e];
          nextvalues = values[(size_t)(t + 1) * nv + col];
        }
        const float gv = gamma * nextvalues;
        const float gvn = gv * nextnonterminal;
        const float rg = rewards[(size_t)t * nv + col] + gvn;
        const float delta = rg - values[(size_t)t * nv + col];
        const float c = gamma_tau * nextnonterminal;
        const float cl = c * lastgaelam;
        lastgaelam = delta + cl;
        advs[(size_t)t * nv + col] = lastgaelam;
      }
    }
  }
}

/* returns = advs + values (a2c_common.py:1060); advantages = returns - values
 * (a2c_common.py:1598) - element-wise, both individually rounded. */
void returns_and_advantages_f32(const float* advs, const float* values, float* returns,
                                float* advantages, size_t count) {
  for (size_t i = 0; i < count; ++i) {
    const float ret = advs[i] + values[i];
    returns[i] = ret;
    advantages[i] = ret - values[i];
  }
}

void gae_f64_reference(const float* rewards, const float* values, const float* dones,
                       const float* last_values, const float* last_dones, double* advs,
                       int horizon, int num_envs, int value_size, double gamma, double tau) {
  const size_t nv = (size_t)num_envs * (size_t)value_size;
  for (int e = 0; e < num_envs; ++e) {
    for (int k = 0; k < value_size; ++k) {
      const size_t col = (size_t)e * (size_t)value_size + (size_t)k;
      double lastgaelam = 0.0;
      for (int t = horizon - 1; t >= 0; --t) {
        double nextvalue, nextnonterminal;
        if (t == horizon - 1) {
          nextvalue = (double)last_values[col];
          nextnonterminal = 1.0 - (double)last_dones[e];
        } else {
          nextvalue = (double)values[(size_t)(t + 1) * nv + col];
          nextnonterminal = 1.0 - (double)dones[(size_t)(t + 1) * (size_t)num_envs + (size_t)e];
        }
        const double delta = (double)rewards[(size_t)t * nv + col] +
                             gamma * nextvalue * nextnonterminal - (double)values[(size_t)t * nv + col];
        lastgaelam = delta + gamma * tau * nextnonterminal * lastgaelam;
        advs[(size_t)t * nv + col] = lastgaelam;
      }
    }
  }
}
