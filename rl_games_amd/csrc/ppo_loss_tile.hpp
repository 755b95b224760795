// The clipped-PPO loss of one tile of minibatch rows: device code shared by the stand-alone kernel of
// ppo_loss.hip (see there for what it replaces) and the fused backward kernel of mlp_chain.hip.
#pragma once

#include "rlg_device.hpp"

namespace rlg {

constexpr int kLossRows = 64;      // rows per block (one LDS tile set) ...
constexpr int kLossRowsSmall = 16; // ... and for minibatches of at most kLossSmallBatch rows: a tile is a chain of
constexpr int kLossSmallBatch = 8192;   // short phases, so a small minibatch wants more, smaller tiles
constexpr int kLossThreads = 256;  // threads per block: all walk the tile, the first kLossRows own a row
constexpr int kLossScalars = 7;  // a_loss, c_loss, entropy, b_loss, kl, mask sum, sum d_value
// the `use_smooth_clamp` argument of the C entry points: which actor loss (rl_games/common/common_losses.py:39-82)
constexpr int kSurrogateClip = 0, kSurrogateSmooth = 1, kSurrogateNone = 2;

// The five sums of a row over its A actions (z^2, KL terms, bound terms, log sigma, entropy terms: `.sum(dim=-1)` in the
// reference, models.py:361-364, torch_ext.py:31, a2c_continuous.py:241-253) are accumulated in fp64 and rounded to fp32 ONCE:
// the correctly rounded value of the exact sum of the fp32 terms.  There is no "reference order" to reproduce - ATen's CPU sum
// over a contiguous last dimension is a vectorised cascade whose grouping depends on the host's vector width, a GPU build of
// the reference reduces in yet another order; measured on 200,000 rows of A = 21 (profiles/r6_row_sum_order.txt): the order
// a = 0 .. A-1 reproduces the bits of torch.sum on the AVX-512 host in 38 % of the rows, four partial sums + butterfly
// (rounds 3 - 5) in 51 %, the correctly rounded sum in 56 % - and it is the only one of them that is within half an ulp
// of EVERY fp32 summation order's target, on any machine.  RLG_LOSS_ROWSUM_F64=0: fp32 partial sums (the round-5 bits).
#ifndef RLG_LOSS_ROWSUM_F64
#define RLG_LOSS_ROWSUM_F64 1
#endif
#if RLG_LOSS_ROWSUM_F64
typedef double row_acc_t;
#else
typedef float row_acc_t;
#endif

struct LossArgs {
  // network outputs
  const float* mu;         // [mb, A]
  const float* logstd;     // [A]
  const float* values;     // [mb]
  // minibatch slices of the dataset
  const float* actions;    // [mb, A]
  const float* old_neglogp;  // [mb]
  const float* advantages;   // [mb]
  const float* old_values;   // [mb]
  const float* returns;      // [mb]
  float* old_mu;           // [mb, A]  read, then overwritten with mu     (update_mu_sigma)
  float* old_sigma;        // [mb, A]  read, then overwritten with sigma
  const float* mask;       // [mb] or nullptr (rnn_masks)
  const float* mask_sum;   // device scalar sum(mask) for this minibatch, or nullptr
  // outputs
  float* d_mu;             // [mb, A]
  float* d_values;         // [mb]
  double* partials;        // [gridDim.x][kLossScalars + 2A]: scalars | d logstd terms | sum_rows d_mu
  int mb, A;
  int ld_mu, ld_val, ld_dmu, ld_dval;   // row strides (elements) of mu / values / d_mu / d_values
  float e_clip, critic_coef, bounds_coef;
  int clip_value;          // default_critic_loss clip flag
  int smooth;              // surrogate kind: 0 clipped PPO, 1 smooth clamp (use_smooth_clamp), 2 none (ppo: False)
  int bound_kind;          // 0 none (coef None), 1 'bound', 2 'regularisation'
  int write_back;          // overwrite old_mu/old_sigma with the new policy's
};

__device__ __forceinline__ float smooth_clamp_f(float x, float mi, float mx) {
  // 1/(1 + exp((-(x-mi)/(mx-mi)+0.5)*4)) * (mx-mi) + mi          common_losses.py:32-36
  const float t = ((-(x - mi) / (mx - mi)) + 0.5f) * 4.0f;
  return (1.0f / (1.0f + expf(t))) * (mx - mi) + mi;
}

__device__ __forceinline__ float smooth_clamp_grad(float x, float mi, float mx) {
  // d/dx of the above: s = 1/(1+e^t), ds/dt = -e^t s^2, dt/dx = -4/(mx-mi)  ->  4 e^t s^2.
  // (4 s (1-s) is the same number but cancels catastrophically once s -> 1.)
  const float t = ((-(x - mi) / (mx - mi)) + 0.5f) * 4.0f;
  const float e = expf(t);
  const float s = 1.0f / (1.0f + e);
  return (4.0f * e) * (s * s);
}

// The row-level part of the loss: from a row's five sums over the actions (z^2, KL terms, bound terms, log sigma, entropy
// terms) and its scalars to  g = d loss / d neglogp x weight,  w = mask / count,  dv = d loss / d value,  and the row's
// seven contributions to the tile's partial sums.  One definition for both tile forms below.
struct LossRow {
  float g, w, dv;
};
__device__ __forceinline__ LossRow ppo_loss_row(const LossArgs& p, int A, float s_z2, float s_kl, float s_b, float s_ls, float s_ent,
                                                float r_adv, float r_onlp, float r_v, float r_vo, float r_ret, float r_mask, float lo,
                                                float hi, float denom_count, double (&acc)[kLossScalars]) {
  // neglogp                                                                models.py:361-364
  const float nlp = (0.5f * s_z2 + static_cast<float>(0.9189385332046727 * A)) + s_ls;
  const float adv = r_adv;
  const float ratio = expf(r_onlp - nlp);                                   // common_losses.py:75
  const float surr1 = adv * ratio;
  float l2, dl2_dratio;  // second branch and its derivative w.r.t. ratio (without the -adv)
  if (p.smooth == kSurrogateSmooth) {
    l2 = adv * smooth_clamp_f(ratio, lo, hi);
    dl2_dratio = smooth_clamp_grad(ratio, lo, hi);
  } else {
    l2 = adv * clamp_nan(ratio, lo, hi);
    dl2_dratio = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
  }
  const float n1 = -surr1, n2 = -l2;
  float a_loss = fmaxf(n1, n2);                                             // :78
  // torch.max backward: the larger branch takes the gradient, equal branches split it
  float w1, w2;
  if (n1 > n2) {
    w1 = 1.0f;
    w2 = 0.0f;
  } else if (n2 > n1) {
    w1 = 0.0f;
    w2 = 1.0f;
  } else {
    w1 = 0.5f;
    w2 = 0.5f;
  }
  // d a_loss / d ratio = -adv*(w1 + w2*dl2) ; d ratio / d nlp = -ratio
  float g_nlp = adv * (w1 + w2 * dl2_dratio) * ratio;
  if (p.smooth == kSurrogateNone) {                                         // ppo: False - plain A2C   common_losses.py:59, 80
    a_loss = nlp * adv;
    g_nlp = adv;
  }

  // critic                                                                 common_losses.py:20-27
  const float v = r_v, vo = r_vo, R = r_ret;
  float c_loss, g_v;
  if (p.clip_value) {
    const float delta = v - vo;
    const float vclip = vo + clamp_nan(delta, -p.e_clip, p.e_clip);
    const float d1 = v - R, d2 = vclip - R;
    const float c1 = d1 * d1, c2 = d2 * d2;
    c_loss = fmaxf(c1, c2);
    const float in = (delta >= -p.e_clip && delta <= p.e_clip) ? 1.0f : 0.0f;
    if (c1 > c2) {
      g_v = 2.0f * d1;
    } else if (c2 > c1) {
      g_v = 2.0f * d2 * in;
    } else {
      g_v = 0.5f * (2.0f * d1) + 0.5f * (2.0f * d2 * in);
    }
  } else {
    const float d = R - v;
    c_loss = d * d;
    g_v = -2.0f * d;
  }

  const float m = r_mask;
  const float w = m / denom_count;          // d(mean)/d(element)
  const float dv = (0.5f * p.critic_coef) * g_v * w;                        // a2c_continuous.py:133
  acc[6] = static_cast<double>(dv);
  acc[0] = static_cast<double>(a_loss) * m;
  acc[1] = static_cast<double>(c_loss) * m;
  acc[2] = static_cast<double>(s_ent) * m;
  acc[3] = static_cast<double>(s_b) * m;
  acc[4] = static_cast<double>(s_kl) * m;
  acc[5] = m;
  return LossRow{g_nlp * w, w, dv};
}

// One tile of kRows rows (the tile_index-th of the minibatch) by the kThreads threads of a workgroup; `lds`:
// ppo_loss_lds_bytes(kRows, A) bytes, 16-byte aligned.  Called by ppo_loss_kernel and by the fused backward
// kernel in front of its own prologue (csrc/mlp_chain.hip: d heads of the tile are then already there).
template <int kRows, int kThreads = kLossThreads>
__device__ __forceinline__ void ppo_loss_tile_rows(const LossArgs& p, float* lds, int tile_index) {
  static_assert(kRows <= kThreads, "one thread per row in phase 2");
  const int A = p.A;
  const int AP = A | 1;                       // odd row stride: conflict-free row walks
  float* t_z2 = lds;                          // [256][AP]  z^2, later g*(1-z^2)
  float* t_kl = t_z2 + kRows * AP;        // [256][AP]
  float* t_b = t_kl + kRows * AP;         // [256][AP]
  float* row_g = t_b + kRows * AP;        // [256]  d loss / d neglogp of the row
  float* row_w = row_g + kRows;           // [256]  inv_count * mask of the row
  float* col_sigma = row_w + kRows;       // [A]
  float* col_logstd = col_sigma + A;          // [A]
  float* col_ent = col_logstd + A;            // [A]  entropy term of the column (the same for every row)
  double* red = reinterpret_cast<double*>(
      (reinterpret_cast<uintptr_t>(col_ent + A) + 7) & ~static_cast<uintptr_t>(7));

  const int tid = threadIdx.x;
  const long long row0 = static_cast<long long>(tile_index) * kRows;
  const int rows = static_cast<int>(min(static_cast<long long>(kRows), p.mb - row0));
  const long long e0 = row0 * A;              // first element of the tile
  const int tile_elems = rows * A;

  // The per-row inputs of phase 2 and the first kBatch element rounds of phase 1 are requested before anything else
  // is computed: as written phase by phase, a tile is ~14 dependent memory round trips (the element loop cannot be
  // pipelined by the compiler: old_mu / old_sigma are read AND written), which a backward workgroup that owns its CU
  // alone (csrc/mlp_chain_bx.hip) pays in full.
  // phase 2 works on a row with kPart threads (a quad: three of them take one of the row's three sums over the
  // actions each, in the order a = 0 .. A-1 of a single thread, the fourth the two sums of column constants) when the
  // workgroup has four threads per row, else with one
  constexpr int kPart = (kThreads >= 4 * kRows) ? 4 : 1;
  const int prow = tid / kPart, ppart = tid % kPart;
  float r_adv = 0.0f, r_onlp = 0.0f, r_v = 0.0f, r_vo = 0.0f, r_ret = 0.0f, r_mask = 1.0f;
  if (prow < rows) {
    const long long i = row0 + prow;
    r_adv = p.advantages[i];
    r_onlp = p.old_neglogp[i];
    r_v = p.values[i * p.ld_val];
    r_vo = p.old_values[i];
    r_ret = p.returns[i];
    if (p.mask) r_mask = p.mask[i];
  }
  constexpr int kBatch = 8;                   // element rounds held in registers (A <= 8 * kThreads / kRows)
  float e_mu[kBatch], e_x[kBatch], e_omu[kBatch], e_osg[kBatch], e_ls[kBatch];
  {
    int r = tid / A, a = tid - r * A;
    const int dr = kThreads / A, da = kThreads - dr * A;
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int e = tid + k * kThreads;
      e_mu[k] = e_x[k] = e_omu[k] = e_ls[k] = 0.0f;
      e_osg[k] = 1.0f;
      if (e < tile_elems) {
        e_ls[k] = p.logstd[a];
        e_mu[k] = p.mu[(row0 + r) * p.ld_mu + a];
        e_x[k] = p.actions[e0 + e];
        e_omu[k] = p.old_mu[e0 + e];
        e_osg[k] = p.old_sigma[e0 + e];
      }
      a += da;
      r += dr;
      if (a >= A) {
        a -= A;
        r += 1;
      }
    }
  }
  for (int a = tid; a < A; a += kThreads) {
    const float ls = p.logstd[a];
    col_logstd[a] = ls;
    const float sg = expf(ls);                                                // models.py:296
    col_sigma[a] = sg;
    // Normal.entropy(): 0.5 + 0.5*log(2*pi) + log(scale) - once per column, not once per row and column
    col_ent[a] = 1.4189385332046727f + logf(sg);
  }
  // (no barrier here: phase 1 forms sigma = expf(logstd[a]) per element - the same bits; the column arrays are read
  //  behind the barrier that ends phase 1)

  const float lo = 1.0f - p.e_clip, hi = 1.0f + p.e_clip;
  float denom_count = static_cast<float>(p.mb);
  if (p.mask) denom_count = fmaxf(*p.mask_sum, 1.0f);                         // torch_ext.py:165

  // ------------------------------ phase 1: element-wise ------------------------------
  {
    int r = tid / A, a = tid - r * A;
    const int dr = kThreads / A, da = kThreads - dr * A;
    auto element = [&](int e, int r, int a, float mu, float x, float omu, float osg, float ls) {
      const float sg = expf(ls);                                              // models.py:296
      const float z = (x - mu) / sg;                                          // models.py:362
      t_z2[r * AP + a] = z * z;
      // policy_kl(p0 = new, p1 = old)                                        torch_ext.py:28-31
      const float c1 = logf(osg / sg + 1e-5f);
      const float dm = omu - mu;
      const float c2 = (sg * sg + dm * dm) / (2.0f * (osg * osg + 1e-5f));
      t_kl[r * AP + a] = (c1 + c2) + (-0.5f);
      float b = 0.0f;
      if (p.bound_kind == 1) {                                                // a2c_continuous.py:248-253
        const float hi_t = fmaxf(mu - 1.1f, 0.0f);
        const float lo_t = fminf(mu + 1.1f, 0.0f);
        b = lo_t * lo_t + hi_t * hi_t;
      } else if (p.bound_kind == 2) {                                         // :241-246
        b = mu * mu;
      }
      t_b[r * AP + a] = b;
      if (p.write_back) {                                                     // datasets.py:42-43
        p.old_mu[e0 + e] = mu;
        p.old_sigma[e0 + e] = sg;
      }
    };
    auto advance = [&]() {
      a += da;
      r += dr;
      if (a >= A) {
        a -= A;
        r += 1;
      }
    };
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int e = tid + k * kThreads;
      if (e < tile_elems) element(e, r, a, e_mu[k], e_x[k], e_omu[k], e_osg[k], e_ls[k]);
      advance();
    }
    for (int e = tid + kBatch * kThreads; e < tile_elems; e += kThreads) {
      element(e, r, a, p.mu[(row0 + r) * p.ld_mu + a], p.actions[e0 + e], p.old_mu[e0 + e], p.old_sigma[e0 + e], p.logstd[a]);
      advance();
    }
  }
  __syncthreads();

  // ------------------------------ phase 2: one thread per row ------------------------
  double acc[kLossScalars] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  float s_z2 = 0.0f, s_kl = 0.0f, s_b = 0.0f, s_ls = 0.0f, s_ent = 0.0f;
  if constexpr (kPart == 4) {
    // every sum by ONE thread (row_acc_t: see below), the five of a row by the four threads of its quad at the same time;
    // then each thread of the quad collects all five
    const int rr = prow < rows ? prow : 0;
    row_acc_t mine = 0, mine2 = 0;
    if (ppart == 3) {
      for (int a = 0; a < A; ++a) {
        mine += static_cast<row_acc_t>(col_logstd[a]);
        mine2 += static_cast<row_acc_t>(col_ent[a]);
      }
    } else {
      const float* tile = ppart == 0 ? t_z2 : (ppart == 1 ? t_kl : t_b);
      for (int a = 0; a < A; ++a) mine += static_cast<row_acc_t>(tile[rr * AP + a]);
    }
    const float mine_f = static_cast<float>(mine), mine2_f = static_cast<float>(mine2);
    const int quad = static_cast<int>(threadIdx.x & 63) & ~3;
    s_z2 = __shfl(mine_f, quad + 0, kWave);
    s_kl = __shfl(mine_f, quad + 1, kWave);
    s_b = __shfl(mine_f, quad + 2, kWave);
    s_ls = __shfl(mine_f, quad + 3, kWave);
    s_ent = __shfl(mine2_f, quad + 3, kWave);
  } else if (prow < rows) {
    row_acc_t d_z2 = 0, d_kl = 0, d_b = 0, d_ls = 0, d_ent = 0;
    for (int a = 0; a < A; ++a) {
      d_z2 += static_cast<row_acc_t>(t_z2[prow * AP + a]);
      d_kl += static_cast<row_acc_t>(t_kl[prow * AP + a]);
      d_b += static_cast<row_acc_t>(t_b[prow * AP + a]);
      d_ls += static_cast<row_acc_t>(col_logstd[a]);
      d_ent += static_cast<row_acc_t>(col_ent[a]);
    }
    s_z2 = static_cast<float>(d_z2);
    s_kl = static_cast<float>(d_kl);
    s_b = static_cast<float>(d_b);
    s_ls = static_cast<float>(d_ls);
    s_ent = static_cast<float>(d_ent);
  }
  if (prow < rows && ppart == 0) {
    const long long i = row0 + prow;
    const LossRow R = ppo_loss_row(p, A, s_z2, s_kl, s_b, s_ls, s_ent, r_adv, r_onlp, r_v, r_vo, r_ret, r_mask, lo, hi, denom_count, acc);
    row_g[prow] = R.g;
    row_w[prow] = R.w;
    p.d_values[i * p.ld_dval] = R.dv;
  }
  block_sum<kLossScalars, kThreads>(acc, red);
  double* out = p.partials + static_cast<long long>(tile_index) * (kLossScalars + 2 * A);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < kLossScalars; ++k) out[k] = acc[k];
  }
  __syncthreads();   // row_g / row_w visible; also fences the reuse of `red`

  // ------------------------------ phase 3: d mu, logstd terms -------------------------
  {
    int r = tid / A, a = tid - r * A;
    const int dr = kThreads / A, da = kThreads - dr * A;
    auto element = [&](int r, int a, float mu, float x) {
      const float sg = col_sigma[a];
      const float z = (x - mu) / sg;
      float db = 0.0f;
      if (p.bound_kind == 1) {
        db = 2.0f * fminf(mu + 1.1f, 0.0f) + 2.0f * fmaxf(mu - 1.1f, 0.0f);
      } else if (p.bound_kind == 2) {
        db = 2.0f * mu;
      }
      // d nlp / d mu = -z / sigma
      const float dmu = row_g[r] * (-(z / sg)) + (row_w[r] * p.bounds_coef) * db;
      p.d_mu[(row0 + r) * p.ld_dmu + a] = dmu;
      t_kl[r * AP + a] = dmu;                     // column sums -> bias gradient of the mu head
      // d nlp / d logstd = 1 - z^2
      t_z2[r * AP + a] = row_g[r] * (1.0f - z * z);
    };
    auto advance = [&]() {
      a += da;
      r += dr;
      if (a >= A) {
        a -= A;
        r += 1;
      }
    };
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      if (tid + k * kThreads < tile_elems) element(r, a, e_mu[k], e_x[k]);     // (mu, x: still in registers)
      advance();
    }
    for (int e = tid + kBatch * kThreads; e < tile_elems; e += kThreads) {
      element(r, a, p.mu[(row0 + r) * p.ld_mu + a], p.actions[e0 + e]);
      advance();
    }
  }
  __syncthreads();

  // ------------------------------ phase 4: column sums over the block's rows ----------
  // two column sets (d logstd terms in t_z2, d mu in t_kl); 8 row groups x A columns each - both sets in one sweep - then
  // 2 A threads fold the 8 partials of their column (fixed order).
  {
    constexpr int groups = 8;
    const int per = (rows + groups - 1) / groups;
    for (int j = tid; j < 2 * groups * A; j += kThreads) {
      const int set = j / (groups * A);
      const int jj = j - set * groups * A;
      const int g = jj / A, a = jj - g * A;
      const float* tile = set == 0 ? t_z2 : t_kl;
      double s = 0.0;
      const int r_end = min(rows, (g + 1) * per);
      for (int r = g * per; r < r_end; ++r) s += static_cast<double>(tile[r * AP + a]);
      red[j] = s;
    }
    __syncthreads();
    for (int j = tid; j < 2 * A; j += kThreads) {
      const int set = j / A, a = j - set * A;
      double s = 0.0;
      for (int g = 0; g < groups; ++g) s += red[set * groups * A + g * A + a];
      out[kLossScalars + set * A + a] = s;
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------
// Quad form (A <= 4 * kQuadK, four threads per row available - every shape the agent runs): thread (row r = tid >> 2,
// q = tid & 3) owns the actions a = q, q + 4, ... of its row IN REGISTERS from the loads to the stores.  The tile form
// above walks the tile four times through LDS behind seven barriers - ~14 dependent memory round trips and 21 k cycles
// for 64 rows when nothing else runs on the CU (the split-bf16 backward, csrc/mlp_chain_bx.hip).  Here: every load up
// front, the row's five sums by two butterfly steps inside the quad, the row-level maths redundantly in its four
// threads, d mu / write-back straight from the registers, the column sums over the rows by butterflies over the
// lanes that share q (fp64) and one exchange between the waves: two barriers.
// Same formulas (ppo_loss_row, the element expressions) and - since round 6, with the row sums accumulated in fp64 and rounded
// once (row_acc_t above) - the same bits as the tile form.
// ------------------------------------------------------------------------------------------------
constexpr int kQuadK = 8;

// What a thread of the quad form needs of its row BEFORE the forward's outputs (mu, value) exist: requested first by the
// fused forward + loss + backward launch (csrc/mlp_chain.hip), kept in registers through the forward.
struct LossQuadInputs {
  float r_adv, r_onlp, r_vo, r_ret, r_mask;
  float e_x[kQuadK], e_omu[kQuadK], e_osg[kQuadK], e_ls[kQuadK];
};

template <int kRows, int kThreads>
__device__ __forceinline__ void ppo_loss_quad_load(const LossArgs& p, int tile_index, LossQuadInputs& in) {
  static_assert(kThreads >= 4 * kRows, "four threads per row");
  const int A = p.A;
  const int tid = threadIdx.x;
  const int r = tid >> 2, q = tid & 3;
  const long long row0 = static_cast<long long>(tile_index) * kRows;
  const int rows = static_cast<int>(min(static_cast<long long>(kRows), p.mb - row0));
  const bool row_ok = r < rows;
  const long long i = row0 + (row_ok ? r : 0);
  in.r_adv = in.r_onlp = in.r_vo = in.r_ret = 0.0f;
  in.r_mask = 1.0f;
  if (row_ok) {
    in.r_adv = p.advantages[i];
    in.r_onlp = p.old_neglogp[i];
    in.r_vo = p.old_values[i];
    in.r_ret = p.returns[i];
    if (p.mask) in.r_mask = p.mask[i];
  }
#pragma unroll
  for (int k = 0; k < kQuadK; ++k) {
    const int a = q + 4 * k;
    in.e_x[k] = in.e_omu[k] = in.e_ls[k] = 0.0f;
    in.e_osg[k] = 1.0f;
    if (row_ok && a < A) {
      in.e_ls[k] = p.logstd[a];
      in.e_x[k] = p.actions[i * A + a];
      in.e_omu[k] = p.old_mu[i * A + a];
      in.e_osg[k] = p.old_sigma[i * A + a];
    }
  }
}

// handoff (optional, LDS, [kRows][handoff_ld] floats): the tile's d heads - column 0 d value, columns 1 .. A d mu - for a
// caller that consumes them in the same workgroup (the fused backward: no store fence + re-load from global memory).
template <int kRows, int kThreads>
__device__ __forceinline__ void ppo_loss_quad_run(const LossArgs& p, float* lds, int tile_index, const LossQuadInputs& in,
                                                  float* handoff = nullptr, int handoff_ld = 0) {
  static_assert(kThreads >= 4 * kRows, "four threads per row");
  constexpr int kWaves = kThreads / kWave;
  const int A = p.A;
  const int tid = threadIdx.x;
  const int r = tid >> 2, q = tid & 3;
  const long long row0 = static_cast<long long>(tile_index) * kRows;
  const int rows = static_cast<int>(min(static_cast<long long>(kRows), p.mb - row0));
  const bool row_ok = r < rows;                       // (threads >= 4 * kRows and rows past the end: no row)
  const long long i = row0 + (row_ok ? r : 0);
  double* red = reinterpret_cast<double*>(lds);       // [kWaves][2][A] column partials, then the block sums' scratch
  double* red_cols = red;
  double* red_scal = red + kWaves * 2 * A;

  // ---- the forward's outputs for this tile; everything else came with `in`
  const float r_adv = in.r_adv, r_onlp = in.r_onlp, r_vo = in.r_vo, r_ret = in.r_ret, r_mask = in.r_mask;
  float r_v = 0.0f;
  float e_mu[kQuadK], e_x[kQuadK], e_omu[kQuadK], e_osg[kQuadK], e_ls[kQuadK];
  if (row_ok) r_v = p.values[i * p.ld_val];
#pragma unroll
  for (int k = 0; k < kQuadK; ++k) {
    const int a = q + 4 * k;
    e_mu[k] = 0.0f;
    e_x[k] = in.e_x[k];
    e_omu[k] = in.e_omu[k];
    e_osg[k] = in.e_osg[k];
    e_ls[k] = in.e_ls[k];
    if (row_ok && a < A) e_mu[k] = p.mu[i * p.ld_mu + a];
  }
  const float lo = 1.0f - p.e_clip, hi = 1.0f + p.e_clip;
  float denom_count = static_cast<float>(p.mb);
  if (p.mask) denom_count = fmaxf(*p.mask_sum, 1.0f);                         // torch_ext.py:165

  // ---- element-wise: the terms of the row's five sums
  float e_z[kQuadK], e_sg[kQuadK];
  row_acc_t s_z2 = 0, s_kl = 0, s_b = 0, s_ls = 0, s_ent = 0;
#pragma unroll
  for (int k = 0; k < kQuadK; ++k) {
    const int a = q + 4 * k;
    e_z[k] = 0.0f;
    e_sg[k] = 1.0f;
    if (a < A) {                                       // (also for threads without a row: keeps s_ls / s_ent uniform)
      const float ls = row_ok ? e_ls[k] : p.logstd[a];
      const float sg = expf(ls);                                              // models.py:296
      e_sg[k] = sg;
      s_ls += static_cast<row_acc_t>(ls);
      s_ent += static_cast<row_acc_t>(1.4189385332046727f + logf(sg));      // Normal.entropy(): 0.5 + 0.5*log(2*pi) + log(scale)
      if (row_ok) {
        const float mu = e_mu[k], x = e_x[k], omu = e_omu[k], osg = e_osg[k];
        const float z = (x - mu) / sg;                                        // models.py:362
        e_z[k] = z;
        s_z2 += static_cast<row_acc_t>(z * z);
        // policy_kl(p0 = new, p1 = old)                                      torch_ext.py:28-31
        const float c1 = logf(osg / sg + 1e-5f);
        const float dm = omu - mu;
        const float c2 = (sg * sg + dm * dm) / (2.0f * (osg * osg + 1e-5f));
        s_kl += static_cast<row_acc_t>((c1 + c2) + (-0.5f));
        if (p.bound_kind == 1) {                                              // a2c_continuous.py:248-253
          const float hi_t = fmaxf(mu - 1.1f, 0.0f);
          const float lo_t = fminf(mu + 1.1f, 0.0f);
          s_b += static_cast<row_acc_t>(lo_t * lo_t + hi_t * hi_t);
        } else if (p.bound_kind == 2) {                                       // :241-246
          s_b += static_cast<row_acc_t>(mu * mu);
        }
        if (p.write_back) {                                                   // datasets.py:42-43
          p.old_mu[i * A + a] = mu;
          p.old_sigma[i * A + a] = sg;
        }
      }
    }
  }
  // the quad's four partial sums -> the row's sums, the same bits in all four threads: (s0 + s1) + (s2 + s3); in fp64 the
  // grouping does not matter (A <= 32 fp32 terms: every partial sum is exact or within 2^-53), the ONE rounding to fp32
  // below makes each row sum the correctly rounded value of its terms' exact sum
  auto quad_sum = [&](row_acc_t v) -> float {
    v += __shfl_xor(v, 1, kWave);
    v += __shfl_xor(v, 2, kWave);
    return static_cast<float>(v);
  };
  const float r_z2 = quad_sum(s_z2);
  const float r_kl = quad_sum(s_kl);
  const float r_b = quad_sum(s_b);
  const float r_ls = quad_sum(s_ls);
  const float r_ent = quad_sum(s_ent);

  // ---- the row
  double acc[kLossScalars] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double none[kLossScalars];
  const LossRow R = ppo_loss_row(p, A, r_z2, r_kl, r_b, r_ls, r_ent, r_adv, r_onlp, r_v, r_vo, r_ret, r_mask, lo, hi, denom_count,
                                 (row_ok && q == 0) ? acc : none);
  if (row_ok && q == 0) {
    p.d_values[i * p.ld_dval] = R.dv;
    if (handoff != nullptr) handoff[r * handoff_ld] = R.dv;
  }

  // ---- d mu and the d logstd terms of the thread's elements; their sums over the rows of this wave (fp64 butterflies
  //      over the lanes that share q: xor 4 .. 32), lanes 0..3 of every wave leave them in LDS
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < kQuadK; ++k) {
    const int a = q + 4 * k;
    double c_dls = 0.0, c_dmu = 0.0;
    if (row_ok && a < A) {
      const float mu = e_mu[k], z = e_z[k], sg = e_sg[k];
      float db = 0.0f;
      if (p.bound_kind == 1) {
        db = 2.0f * fminf(mu + 1.1f, 0.0f) + 2.0f * fmaxf(mu - 1.1f, 0.0f);
      } else if (p.bound_kind == 2) {
        db = 2.0f * mu;
      }
      // d nlp / d mu = -z / sigma
      const float dmu = R.g * (-(z / sg)) + (R.w * p.bounds_coef) * db;
      p.d_mu[i * p.ld_dmu + a] = dmu;
      if (handoff != nullptr) handoff[r * handoff_ld + 1 + a] = dmu;
      c_dmu = static_cast<double>(dmu);                // column sums -> bias gradient of the mu head
      c_dls = static_cast<double>(R.g * (1.0f - z * z));     // d nlp / d logstd = 1 - z^2
    }
    if (4 * k < A) {                                   // (uniform: some thread of the tile owns column q + 4 k)
#pragma unroll
      for (int o = 4; o < kWave; o <<= 1) {
        c_dls += __shfl_xor(c_dls, o, kWave);
        c_dmu += __shfl_xor(c_dmu, o, kWave);
      }
      if (lane < 4 && a < A) {
        red_cols[(wave * 2 + 0) * A + a] = c_dls;
        red_cols[(wave * 2 + 1) * A + a] = c_dmu;
      }
    }
  }
  block_sum<kLossScalars, kThreads>(acc, red_scal);      // (its barrier also publishes red_cols)
  double* out = p.partials + static_cast<long long>(tile_index) * (kLossScalars + 2 * A);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < kLossScalars; ++k) out[k] = acc[k];
  }
  if constexpr (kWaves == 1) __syncthreads();            // (block_sum of a single wave has no barrier)
  for (int j = tid; j < 2 * A; j += kThreads) {
    const int set = j / A, a = j - set * A;
    double s2 = 0.0;
    for (int w = 0; w < kWaves; ++w) s2 += red_cols[(w * 2 + set) * A + a];
    out[kLossScalars + set * A + a] = s2;
  }
  __syncthreads();                                       // the LDS is the caller's again
}

// load + run: the tile as one call (every load up front, as before the split)
template <int kRows, int kThreads>
__device__ __forceinline__ void ppo_loss_tile_quad(const LossArgs& p, float* lds, int tile_index, float* handoff = nullptr,
                                                   int handoff_ld = 0) {
  LossQuadInputs in;
  ppo_loss_quad_load<kRows, kThreads>(p, tile_index, in);
  ppo_loss_quad_run<kRows, kThreads>(p, lds, tile_index, in, handoff, handoff_ld);
}

// The tile by whichever form fits.  Returns true when the d heads were left in `handoff` (quad form only).
template <int kRows, int kThreads = kLossThreads>
__device__ __forceinline__ bool ppo_loss_tile(const LossArgs& p, float* lds, int tile_index, float* handoff = nullptr,
                                              int handoff_ld = 0) {
  if constexpr (kThreads >= 4 * kRows) {
    if (p.A <= 4 * kQuadK) {
      ppo_loss_tile_quad<kRows, kThreads>(p, lds, tile_index, handoff, handoff_ld);
      return handoff != nullptr;
    }
  }
  ppo_loss_tile_rows<kRows, kThreads>(p, lds, tile_index);
  return false;
}


// LDS bytes of one ppo_loss_tile<rows>: 3 tiles [rows][A|1] + 2 row vectors + 3 column vectors (floats),
// then the fp64 reduction scratch (2 column sets x 8 row groups x A, or the block sums' per-wave values).
inline size_t ppo_loss_lds_bytes(int rows, int A, int threads = kLossThreads) {
  const int AP = A | 1;
  size_t shm = (static_cast<size_t>(3) * rows * AP + 2 * rows + 3 * A + 4) * sizeof(float);
  shm = (shm + 7) & ~static_cast<size_t>(7);
  const size_t red_doubles = static_cast<size_t>(16) * A > static_cast<size_t>(kLossScalars) * (threads / kWave)
                                 ? static_cast<size_t>(16) * A
                                 : static_cast<size_t>(kLossScalars) * (threads / kWave);
  const size_t tile_form = shm + red_doubles * sizeof(double);
  // quad form: [waves][2][A] column partials + the block sums' per-wave values, all fp64
  const size_t quad_form = (static_cast<size_t>(threads / kWave) * (2 * A + kLossScalars)) * sizeof(double);
  return tile_form > quad_form ? tile_form : quad_form;
}

}  // namespace rlg
