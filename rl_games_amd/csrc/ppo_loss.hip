// Fused clipped-PPO loss: distribution epilogue + losses + KL + analytic backward, gfx950.
//
// Replaces, per minibatch (continuous actions, fixed-sigma 'exp' parametrisation, value_size 1):
//   * the Normal-distribution epilogue of ModelA2CContinuousLogStd.forward
//     (rl_games/algos_torch/models.py:329-347: sigma = exp(logstd) :296, entropy :337,
//     neglogp :361-364);
//   * A2CAgent.calc_losses (rl_games/algos_torch/a2c_continuous.py:97-134) with
//     common_losses.actor_loss / smoothed_actor_loss / default_critic_loss
//     (rl_games/common/common_losses.py:64-82, :39-61, :16-29), bound_loss / reg_loss
//     (a2c_continuous.py:241-257) and torch_ext.apply_masks (torch_ext.py:157-170);
//   * loss.backward() down to the network outputs (a2c_continuous.py:211): d loss/d mu,
//     d loss/d logstd, d loss/d value, reproducing torch.max's tie rule (equal branches split
//     the gradient 1/2 + 1/2) and clamp's inclusive pass-through range;
//   * torch_ext.policy_kl (torch_ext.py:27-36; masked mean a2c_continuous.py:215-221);
//   * PPODataset.update_mu_sigma (rl_games/common/datasets.py:33-43): the new mu/sigma are
//     written over the old ones in the same pass (the old values are read first for the KL).
//
// One block = 64 rows, 256 threads.  Phase 1 walks the block's [256, A] tile element-wise with coalesced
// loads (mu, actions, old mu, old sigma), writes the per-element terms to LDS; phase 2 has one
// thread per row reduce over A (odd LDS row stride -> conflict free), evaluate the scalar
// losses and the row's gradient coefficient; phase 3 walks the tile again (mu/actions come
// back from L2) to emit d mu coalesced and the per-column logstd gradient terms; phase 4
// reduces those over rows.  Per-block fp64 partial sums go to global memory (no atomics);
// ppo_loss_finalize_kernel folds them into the scalars and d logstd.
//
// Algorithmic HBM traffic per row: reads 4*A*4 + 5*4 (+4 mask), writes 3*A*4 + 4 bytes
// (d mu, new mu, new sigma, d value) = 28*A + 24 bytes.  mu / values / d mu / d values may be
// strided views (columns of a fused [mb, 1+A] head buffer); the column sums of d mu and the sum
// of d value (= the head biases' gradients) ride along in the block partials.

#include "ppo_loss_tile.hpp"

#include <cstdlib>

namespace rlg {

static int loss_tile_rows(int minibatch) {
  static const int forced = [] {
    const char* e = std::getenv("RLG_LOSS_ROWS");       // tools: A/B measurements (16 / 32 / 64)
    return e ? std::atoi(e) : 0;
  }();
  if (forced == 16 || forced == 32 || forced == 64) return forced;
  return minibatch <= kLossSmallBatch ? kLossRowsSmall : kLossRows;
}

template <int kRows>
__global__ __launch_bounds__(kLossThreads) void ppo_loss_kernel(LossArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ppo_loss_tile<kRows>(p, lds, blockIdx.x);
}

// Scalars written by the finalise kernel (fp32[8], read lazily by the host):
//   [0] a_loss [1] c_loss [2] entropy [3] b_loss [4] kl [5] total loss [6] sum(mask) [7] unused

__global__ __launch_bounds__(1024) void ppo_loss_finalize_kernel(
    const double* __restrict__ partials, int nblocks, int A, int mb, int masked,
    float critic_coef, float entropy_coef, float bounds_coef, float* __restrict__ scalars,
    float* __restrict__ d_logstd, float* __restrict__ kl_slot, float* __restrict__ d_mu_bias,
    float* __restrict__ d_value_bias) {
  // 32 block-slices x 32 columns per pass; slices are folded through LDS in a fixed order.
  __shared__ double part[32][33];
  __shared__ double sh[kLossScalars];
  const int W = kLossScalars + 2 * A;
  const int col_in_pass = threadIdx.x & 31;
  const int slice = threadIdx.x >> 5;
  for (int c0 = 0; c0 < W; c0 += 32) {
    const int c = c0 + col_in_pass;
    double s = 0.0;
    if (c < W) {
      // four independent loads in flight per thread (the launch is a chain of L2 round trips otherwise);
      // the summation order is fixed, so the result does not depend on timing
      int b = slice;
      for (; b + 96 < nblocks; b += 128) {
        const double v0 = partials[static_cast<long long>(b) * W + c];
        const double v1 = partials[static_cast<long long>(b + 32) * W + c];
        const double v2 = partials[static_cast<long long>(b + 64) * W + c];
        const double v3 = partials[static_cast<long long>(b + 96) * W + c];
        s += v0;
        s += v1;
        s += v2;
        s += v3;
      }
      for (; b < nblocks; b += 32) s += partials[static_cast<long long>(b) * W + c];
    }
    part[slice][col_in_pass] = s;
    __syncthreads();
    if (slice == 0 && c < W) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += part[k][col_in_pass];
      if (c < kLossScalars) {
        sh[c] = t;
      } else {
        part[0][col_in_pass] = t;   // keep the column total for the d_logstd pass below
      }
    }
    __syncthreads();
    // d loss / d logstd_a = sum_i g_i (1 - z^2)  -  entropy_coef * sum_i w_i   (d ent/d logstd = 1)
    if (slice == 0 && c < W && c >= kLossScalars) {
      const double msum0 = sh[5];
      const double denom0 = masked ? fmax(msum0, 1.0) : static_cast<double>(mb);
      const float w_total = static_cast<float>(msum0 / denom0);
      const int a = c - kLossScalars;
      if (a < A) {
        d_logstd[a] = static_cast<float>(part[0][col_in_pass]) - entropy_coef * w_total;
      } else if (d_mu_bias) {
        d_mu_bias[a - A] = static_cast<float>(part[0][col_in_pass]);   // bias grad of the mu head
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double msum = sh[5];
    const double denom = masked ? fmax(msum, 1.0) : static_cast<double>(mb);
    const float a_loss = static_cast<float>(sh[0] / denom);
    const float c_loss = static_cast<float>(sh[1] / denom);
    const float ent = static_cast<float>(sh[2] / denom);
    const float b_loss = static_cast<float>(sh[3] / denom);
    const float kl = static_cast<float>(sh[4] / denom);
    // loss = a + 0.5*c*critic_coef - entropy*entropy_coef + b*bounds_coef   a2c_continuous.py:133
    const float loss = ((a_loss + (0.5f * c_loss) * critic_coef) - ent * entropy_coef) + b_loss * bounds_coef;
    scalars[0] = a_loss;
    scalars[1] = c_loss;
    scalars[2] = ent;
    scalars[3] = b_loss;
    scalars[4] = kl;
    scalars[5] = loss;
    scalars[6] = static_cast<float>(msum);
    scalars[7] = 0.0f;
    if (kl_slot) *kl_slot = kl;
    if (d_value_bias) *d_value_bias = static_cast<float>(sh[6]);       // bias grad of the value head
  }
}


// ---------------------------------------------------------------------------------
// Discrete (Categorical) PPO loss - DiscreteA2CAgent.calc_gradients
// (rl_games/algos_torch/a2c_discrete.py:121-209) with the ModelA2C epilogue
// (rl_games/algos_torch/models.py:95-111: Categorical(logits) -> neglogp, entropy):
//   nlp = logsumexp(z) - z[a] ;  p = softmax(z) ;  H = -sum p log p
//   a_loss / c_loss as in the continuous case; loss = a + 0.5 c critic_coef - H entropy_coef
//   kl  = 0.5 (old_nlp - nlp)^2                                   (:192-198)
// backward: d nlp/d z_j = p_j - [j == a] ;  d H/d z_j = -p_j (log p_j + H).
// One thread per row (the number of actions is small: 2 for config #1's CartPole).
// ---------------------------------------------------------------------------------
constexpr int kMaxBranches = 16;

struct DiscreteLossArgs {
  const float* logits;       // [mb, n] row stride ld; n = sum of branch widths
  long long ld;
  const float* values;       // [mb], element stride ld_values
  long long ld_values;
  const long long* actions;  // [mb, nb] int64 (nb = 1: [mb])
  const unsigned char* action_masks;   // [mb, n] bool (1 = allowed) or nullptr
  const float* old_neglogp;
  const float* advantages;
  const float* old_values;
  const float* returns;
  const float* mask;         // or nullptr
  const float* mask_sum;
  float* d_logits;           // [mb, n], row stride ld_d_logits
  float* d_values;           // [mb], element stride ld_d_values
  long long ld_d_logits, ld_d_values;
  double* partials;          // [gridDim.x][kLossScalars]
  int mb, n, nb;
  int off[kMaxBranches + 1]; // branch b covers logits [off[b], off[b+1])
  float e_clip, critic_coef, entropy_coef;
  int clip_value, smooth;
};

// CategoricalMasked (rl_games/common/extensions/distributions.py:24-47): disallowed logits are
// replaced by -1e8 before the softmax, contribute 0 to the entropy and receive no gradient.
__device__ __forceinline__ float masked_logit(const DiscreteLossArgs& p, const float* z, const unsigned char* am, int j) {
  return (am == nullptr || am[j]) ? z[j] : -1e8f;
}

__global__ __launch_bounds__(256) void ppo_loss_discrete_kernel(DiscreteLossArgs p) {
  __shared__ double red[kLossScalars * 4];
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  double acc[kLossScalars] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (i < p.mb) {
    const float* z = p.logits + i * p.ld;
    const unsigned char* am = p.action_masks ? p.action_masks + i * p.n : nullptr;
    // ---- pass 1: per-branch log-sum-exp, neglogp and entropy (ModelA2CMultiDiscrete sums them,
    //      models.py:168-173; a single branch is ModelA2C, :95-111)
    float nlp = 0.0f, H = 0.0f;
    float lse_b[kMaxBranches], H_b[kMaxBranches];
    for (int b = 0; b < p.nb; ++b) {
      const int j0 = p.off[b], j1 = p.off[b + 1];
      float zmax = masked_logit(p, z, am, j0);
      for (int j = j0 + 1; j < j1; ++j) zmax = fmaxf(zmax, masked_logit(p, z, am, j));
      float se = 0.0f;
      for (int j = j0; j < j1; ++j) se += expf(masked_logit(p, z, am, j) - zmax);
      const float lse = zmax + logf(se);
      const int a = j0 + static_cast<int>(p.actions[i * p.nb + b]);
      nlp += -(masked_logit(p, z, am, a) - lse);
      float Hb = 0.0f;
      for (int j = j0; j < j1; ++j) {
        if (am && !am[j]) continue;
        const float lp = z[j] - lse;
        Hb -= expf(lp) * lp;
      }
      lse_b[b] = lse;
      H_b[b] = Hb;
      H += Hb;
    }
    const float lo = 1.0f - p.e_clip, hi = 1.0f + p.e_clip;
    const float adv = p.advantages[i];
    const float old_nlp = p.old_neglogp[i];
    const float ratio = expf(old_nlp - nlp);
    float l2, dl2;
    if (p.smooth == kSurrogateSmooth) {
      l2 = adv * smooth_clamp_f(ratio, lo, hi);
      dl2 = smooth_clamp_grad(ratio, lo, hi);
    } else {
      l2 = adv * clamp_nan(ratio, lo, hi);
      dl2 = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    }
    const float n1 = -(adv * ratio), n2 = -l2;
    float a_loss = fmaxf(n1, n2);
    float w1, w2;
    if (n1 > n2) { w1 = 1.0f; w2 = 0.0f; } else if (n2 > n1) { w1 = 0.0f; w2 = 1.0f; } else { w1 = w2 = 0.5f; }
    float g_nlp = adv * (w1 + w2 * dl2) * ratio;
    if (p.smooth == kSurrogateNone) {                                       // ppo: False   common_losses.py:59, 80
      a_loss = nlp * adv;
      g_nlp = adv;
    }
    const float v = p.values[i * p.ld_values], vo = p.old_values[i], R = p.returns[i];
    float c_loss, g_v;
    if (p.clip_value) {
      const float delta = v - vo;
      const float vclip = vo + clamp_nan(delta, -p.e_clip, p.e_clip);
      const float d1 = v - R, d2 = vclip - R;
      const float c1 = d1 * d1, c2 = d2 * d2;
      c_loss = fmaxf(c1, c2);
      const float in = (delta >= -p.e_clip && delta <= p.e_clip) ? 1.0f : 0.0f;
      if (c1 > c2) g_v = 2.0f * d1; else if (c2 > c1) g_v = 2.0f * d2 * in;
      else g_v = 0.5f * (2.0f * d1) + 0.5f * (2.0f * d2 * in);
    } else {
      const float d = R - v;
      c_loss = d * d;
      g_v = -2.0f * d;
    }
    const float m = p.mask ? p.mask[i] : 1.0f;
    const float denom = p.mask ? fmaxf(*p.mask_sum, 1.0f) : static_cast<float>(p.mb);
    const float w = m / denom;
    // ---- pass 2: d loss / d logits.  d nlp/d z_j = p_j - [j == a];  d H_b/d z_j = -p_j (log p_j + H_b)
    for (int b = 0; b < p.nb; ++b) {
      const int j0 = p.off[b], j1 = p.off[b + 1];
      const int a = j0 + static_cast<int>(p.actions[i * p.nb + b]);
      for (int j = j0; j < j1; ++j) {
        float g = 0.0f;
        if (!am || am[j]) {
          const float lp = z[j] - lse_b[b];
          const float pj = expf(lp);
          const float dnlp = pj - (j == a ? 1.0f : 0.0f);
          const float dH = -pj * (lp + H_b[b]);
          g = w * (g_nlp * dnlp - p.entropy_coef * dH);
        }
        p.d_logits[i * p.ld_d_logits + j] = g;
      }
    }
    p.d_values[i * p.ld_d_values] = (0.5f * p.critic_coef) * g_v * w;
    const float dk = old_nlp - nlp;
    acc[0] = static_cast<double>(a_loss) * m;
    acc[1] = static_cast<double>(c_loss) * m;
    acc[2] = static_cast<double>(H) * m;
    acc[4] = static_cast<double>(0.5f * (dk * dk)) * m;
    acc[5] = m;
  }
  block_sum<kLossScalars, 256>(acc, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kLossScalars; ++k) p.partials[static_cast<long long>(blockIdx.x) * kLossScalars + k] = acc[k];
  }
}

// ---------------------------------------------------------------------------------
// Value-only loss of the central value network - CentralValueTrain.calc_loss
// (rl_games/algos_torch/central_value.py:262-276): common_losses.critic_loss (clipped or plain,
// common_losses.py:16-29) + torch_ext.apply_masks mean.  Emits d loss / d value per row and the
// block partials {0, sum c, 0, 0, 0, sum mask, 0} that rlg_ppo_loss_finalize (actions_num 0)
// reduces; with critic_coef = 2 there its total "loss" slot is exactly mean(c).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void value_loss_kernel(
    const float* __restrict__ values, const float* __restrict__ old_values, const float* __restrict__ returns,
    const float* __restrict__ mask, const float* __restrict__ mask_sum, float* __restrict__ d_values,
    double* __restrict__ partials, int mb, float e_clip, int clip_value) {
  __shared__ double red[kLossScalars * 4];
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  double acc[kLossScalars] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (i < mb) {
    const float v = values[i], vo = old_values[i], R = returns[i];
    float c_loss, g_v;
    if (clip_value) {
      const float delta = v - vo;
      const float vclip = vo + clamp_nan(delta, -e_clip, e_clip);
      const float d1 = v - R, d2 = vclip - R;
      const float c1 = d1 * d1, c2 = d2 * d2;
      c_loss = fmaxf(c1, c2);
      const float in = (delta >= -e_clip && delta <= e_clip) ? 1.0f : 0.0f;
      if (c1 > c2) g_v = 2.0f * d1; else if (c2 > c1) g_v = 2.0f * d2 * in;
      else g_v = 0.5f * (2.0f * d1) + 0.5f * (2.0f * d2 * in);
    } else {
      const float d = R - v;
      c_loss = d * d;
      g_v = -2.0f * d;
    }
    const float m = mask ? mask[i] : 1.0f;
    const float denom = mask ? fmaxf(*mask_sum, 1.0f) : static_cast<float>(mb);
    d_values[i] = g_v * (m / denom);
    acc[1] = static_cast<double>(c_loss) * m;
    acc[5] = m;
  }
  block_sum<kLossScalars, 256>(acc, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kLossScalars; ++k) partials[static_cast<long long>(blockIdx.x) * kLossScalars + k] = acc[k];
  }
}

}  // namespace rlg

extern "C" {

int rlg_ppo_loss_num_blocks(int minibatch) {
  const int rows = rlg::loss_tile_rows(minibatch);
  return (minibatch + rows - 1) / rows;
}


int rlg_ppo_loss_partials_per_block(int actions) { return rlg::kLossScalars + 2 * actions; }

int rlg_ppo_loss_fused(const float* mu, const float* logstd, const float* values,
                       const float* actions, const float* old_neglogp, const float* advantages,
                       const float* old_values, const float* returns, float* old_mu,
                       float* old_sigma, const float* mask_or_null, const float* mask_sum_or_null,
                       float* d_mu, float* d_values, double* partials, int minibatch, int actions_num,
                       int ld_mu, int ld_values, int ld_d_mu, int ld_d_values, float e_clip,
                       float critic_coef, float bounds_coef, int clip_value, int use_smooth_clamp,
                       int bound_kind, int write_back, void* stream) {
  using namespace rlg;
  if (minibatch <= 0 || actions_num <= 0) return static_cast<int>(hipErrorInvalidValue);
  if (mask_or_null && !mask_sum_or_null) return static_cast<int>(hipErrorInvalidValue);
  LossArgs p;
  p.mu = mu;
  p.logstd = logstd;
  p.values = values;
  p.actions = actions;
  p.old_neglogp = old_neglogp;
  p.advantages = advantages;
  p.old_values = old_values;
  p.returns = returns;
  p.old_mu = old_mu;
  p.old_sigma = old_sigma;
  p.mask = mask_or_null;
  p.mask_sum = mask_sum_or_null;
  p.d_mu = d_mu;
  p.d_values = d_values;
  p.partials = partials;
  p.mb = minibatch;
  p.A = actions_num;
  p.ld_mu = ld_mu;
  p.ld_val = ld_values;
  p.ld_dmu = ld_d_mu;
  p.ld_dval = ld_d_values;
  p.e_clip = e_clip;
  p.critic_coef = critic_coef;
  p.bounds_coef = bounds_coef;
  p.clip_value = clip_value;
  p.smooth = use_smooth_clamp;
  p.bound_kind = bound_kind;
  p.write_back = write_back;
  const int tile_rows = loss_tile_rows(minibatch);
  const size_t shm = ppo_loss_lds_bytes(tile_rows, actions_num);
  if (shm > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
  const int grid = rlg_ppo_loss_num_blocks(minibatch);
  if (tile_rows == kLossRows) {
    hipLaunchKernelGGL(ppo_loss_kernel<kLossRows>, dim3(grid), dim3(kLossThreads), shm,
                       static_cast<hipStream_t>(stream), p);
  } else if (tile_rows == 32) {
    hipLaunchKernelGGL(ppo_loss_kernel<32>, dim3(grid), dim3(kLossThreads), shm,
                       static_cast<hipStream_t>(stream), p);
  } else {
    hipLaunchKernelGGL(ppo_loss_kernel<kLossRowsSmall>, dim3(grid), dim3(kLossThreads), shm,
                       static_cast<hipStream_t>(stream), p);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_value_loss(const float* values, const float* old_values, const float* returns,
                   const float* mask_or_null, const float* mask_sum_or_null, float* d_values, double* partials,
                   int minibatch, float e_clip, int clip_value, void* stream) {
  if (minibatch <= 0) return static_cast<int>(hipErrorInvalidValue);
  if (mask_or_null && !mask_sum_or_null) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::value_loss_kernel, dim3((minibatch + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), values, old_values, returns, mask_or_null,
                     mask_sum_or_null, d_values, partials, minibatch, e_clip, clip_value);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_ppo_loss_discrete_num_blocks(int minibatch) { return (minibatch + 255) / 256; }

int rlg_ppo_loss_discrete_strided(const float* logits, long long ld_logits, const float* values, long long ld_values,
                                  const long long* actions, const unsigned char* action_masks_or_null,
                                  const int* branch_sizes, int num_branches, const float* old_neglogp,
                                  const float* advantages, const float* old_values, const float* returns,
                                  const float* mask_or_null, const float* mask_sum_or_null, float* d_logits,
                                  long long ld_d_logits, float* d_values, long long ld_d_values, double* partials,
                                  int minibatch, float e_clip, float critic_coef, float entropy_coef, int clip_value,
                                  int use_smooth_clamp, void* stream) {
  using namespace rlg;
  if (minibatch <= 0 || num_branches <= 0 || num_branches > kMaxBranches)
    return static_cast<int>(hipErrorInvalidValue);
  if (mask_or_null && !mask_sum_or_null) return static_cast<int>(hipErrorInvalidValue);
  if (ld_values <= 0 || ld_d_values <= 0) return static_cast<int>(hipErrorInvalidValue);
  DiscreteLossArgs p;
  p.off[0] = 0;
  for (int b = 0; b < num_branches; ++b) {
    if (branch_sizes[b] <= 0) return static_cast<int>(hipErrorInvalidValue);
    p.off[b + 1] = p.off[b] + branch_sizes[b];
  }
  p.logits = logits;
  p.ld = ld_logits;
  p.values = values;
  p.ld_values = ld_values;
  p.actions = actions;
  p.action_masks = action_masks_or_null;
  p.old_neglogp = old_neglogp;
  p.advantages = advantages;
  p.old_values = old_values;
  p.returns = returns;
  p.mask = mask_or_null;
  p.mask_sum = mask_sum_or_null;
  p.d_logits = d_logits;
  p.d_values = d_values;
  p.ld_d_logits = ld_d_logits;
  p.ld_d_values = ld_d_values;
  p.partials = partials;
  p.mb = minibatch;
  p.nb = num_branches;
  p.n = p.off[num_branches];
  if (ld_logits < p.n || ld_d_logits < p.n) return static_cast<int>(hipErrorInvalidValue);
  p.e_clip = e_clip;
  p.critic_coef = critic_coef;
  p.entropy_coef = entropy_coef;
  p.clip_value = clip_value;
  p.smooth = use_smooth_clamp;
  hipLaunchKernelGGL(ppo_loss_discrete_kernel, dim3(rlg_ppo_loss_discrete_num_blocks(minibatch)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_ppo_loss_discrete(const float* logits, long long ld_logits, const float* values,
                          const long long* actions, const unsigned char* action_masks_or_null,
                          const int* branch_sizes, int num_branches, const float* old_neglogp,
                          const float* advantages, const float* old_values, const float* returns,
                          const float* mask_or_null, const float* mask_sum_or_null, float* d_logits,
                          float* d_values, double* partials, int minibatch, float e_clip, float critic_coef,
                          float entropy_coef, int clip_value, int use_smooth_clamp, void* stream) {
  long long n = 0;
  for (int b = 0; b < num_branches && b < rlg::kMaxBranches; ++b) n += branch_sizes[b];
  return rlg_ppo_loss_discrete_strided(logits, ld_logits, values, 1, actions, action_masks_or_null, branch_sizes,
                                       num_branches, old_neglogp, advantages, old_values, returns, mask_or_null,
                                       mask_sum_or_null, d_logits, n, d_values, 1, partials, minibatch, e_clip,
                                       critic_coef, entropy_coef, clip_value, use_smooth_clamp, stream);
}

int rlg_ppo_loss_finalize(const double* partials, int num_blocks, int actions_num, int minibatch,
                          int masked, float critic_coef, float entropy_coef, float bounds_coef,
                          float* scalars8, float* d_logstd, float* kl_slot_or_null,
                          float* d_mu_bias_or_null, float* d_value_bias_or_null, void* stream) {
  hipLaunchKernelGGL(rlg::ppo_loss_finalize_kernel, dim3(1), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), partials, num_blocks, actions_num, minibatch,
                     masked, critic_coef, entropy_coef, bounds_coef, scalars8, d_logstd,
                     kl_slot_or_null, d_mu_bias_or_null, d_value_bias_or_null);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
