// Sequence-persistent LSTM layer for the recurrent PPO policy (BASELINE config #5), gfx950.
//
// Replaces, for the `rnn: {name: lstm, layers: 1}` policies of A2CBuilder
// (rl_games/algos_torch/network_builder.py:447-512 with rl_games/common/layers/recurrent.py:26-58
// `LSTMWithDones`), the per-timestep torch.nn.LSTM (MIOpen) calls, the done-state resets between
// them and - in backward - autograd's BPTT through that Python loop.
//
// One launch runs ALL timesteps of a tile of sequences:
//   * the recurrent weights W_hh [4H, H] (64 KB for H = 64) live in LDS for the whole kernel;
//   * thread (j, group) owns hidden unit j of R sequences: their cell state c stays in registers,
//     the hidden state goes through a double-buffered LDS tile (one barrier per timestep);
//   * the input projection  x_t W_ih^T + b_ih + b_hh  for every timestep is ONE library GEMM
//     before the kernel (rows ordered seq*T + t, the dataset's order); the kernel overwrites it
//     with the activated gates (i, f, g, o - torch.nn.LSTM's gate order), which backward reuses;
//   * done handling: where dones[seq, t] is set the state ENTERING step t is zeroed
//     (recurrent.py:45-55 semantics; the mirror is policy.RnnWithDones).
// Backward walks the same tile in reverse, emitting d(gates pre-activation) [B, 4H]; the weight
// gradients are then plain GEMMs over all timesteps at once (dW_ih = dG^T X, dW_hh = dG^T Hprev).
//
// The matvec per step is H*4H MACs per sequence (16 K at H = 64): far below MFMA tile sizes per
// block and latency-bound by the timestep chain, so it runs on VALU FMAs out of LDS.

#include "rlg_device.hpp"

namespace rlg {

constexpr int kLstmMaxSeqPerBlock = 16;
constexpr int kLstmThreads = 256;

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int H, int kLstmSeqPerBlock>
__global__ __launch_bounds__(kLstmThreads) void lstm_seq_fwd_kernel(
    float* __restrict__ gates,           // [S*T, 4H]  in: x-part + biases, out: activated gates
    const float* __restrict__ w_hh,      // [4H, H]
    const float* __restrict__ h0,        // [S, H]
    const float* __restrict__ c0,        // [S, H]
    const uint8_t* __restrict__ dones,   // [S*T] or nullptr
    float* __restrict__ out,             // [S*T, H]  h_t
    float* __restrict__ c_all,           // [S*T, H]  c_t             (nullptr: not kept)
    float* __restrict__ hprev,           // [S*T, H]  state entering step t, after the reset (nullptr)
    float* __restrict__ hT,              // [S, H] final h (nullptr)
    float* __restrict__ cT,              // [S, H] final c (nullptr)
    int S, int T) {
  constexpr int G = 4 * H;
  constexpr int kGroups = kLstmThreads / H;            // sequence groups per block
  constexpr int R = kLstmSeqPerBlock / kGroups;        // sequences per thread
  static_assert(kLstmThreads % H == 0 && kLstmSeqPerBlock % kGroups == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wT = smem;                                     // [H][4H]: wT[k][r] = w_hh[r][k]
  float* hbuf = smem + H * G;                           // [2][SB][H]
  const int tid = threadIdx.x;
  const int j = tid % H;
  const int grp = tid / H;
  for (int idx = tid; idx < G * H; idx += kLstmThreads) {
    const int k = idx / G, r = idx - k * G;             // LDS write contiguous, global read strided (L2)
    wT[idx] = w_hh[r * H + k];
  }
  const int seq0 = blockIdx.x * kLstmSeqPerBlock;
  int seq[R];
  bool live[R];
  float c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int s = seq0 + grp * R + r;
    live[r] = s < S;
    seq[r] = live[r] ? s : S - 1;
    c[r] = c0[static_cast<long long>(seq[r]) * H + j];
    hbuf[(grp * R + r) * H + j] = h0[static_cast<long long>(seq[r]) * H + j];
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const float* hcur = hbuf + (t & 1) * kLstmSeqPerBlock * H;
    float* hnext = hbuf + ((t + 1) & 1) * kLstmSeqPerBlock * H;
    float acc[4][R];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[g][r] = 0.0f;
    }
#pragma unroll 4
    for (int k = 0; k < H; ++k) {
      const float w0 = wT[k * G + 0 * H + j];
      const float w1 = wT[k * G + 1 * H + j];
      const float w2 = wT[k * G + 2 * H + j];
      const float w3 = wT[k * G + 3 * H + j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float hv = hcur[(grp * R + r) * H + k];   // wave-uniform address: LDS broadcast
        acc[0][r] = __builtin_fmaf(w0, hv, acc[0][r]);
        acc[1][r] = __builtin_fmaf(w1, hv, acc[1][r]);
        acc[2][r] = __builtin_fmaf(w2, hv, acc[2][r]);
        acc[3][r] = __builtin_fmaf(w3, hv, acc[3][r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = static_cast<long long>(seq[r]) * T + t;
      const float keep = (dones && dones[row]) ? 0.0f : 1.0f;
      float* grow = gates + row * G;
      const float gi = sigmoid_f(grow[0 * H + j] + keep * acc[0][r]);
      const float gf = sigmoid_f(grow[1 * H + j] + keep * acc[1][r]);
      const float gg = tanhf(grow[2 * H + j] + keep * acc[2][r]);
      const float go = sigmoid_f(grow[3 * H + j] + keep * acc[3][r]);
      const float cn = gf * (c[r] * keep) + gi * gg;
      const float hn = go * tanhf(cn);
      const float hp = hcur[(grp * R + r) * H + j] * keep;
      c[r] = cn;
      hnext[(grp * R + r) * H + j] = hn;
      if (live[r]) {
        grow[0 * H + j] = gi;
        grow[1 * H + j] = gf;
        grow[2 * H + j] = gg;
        grow[3 * H + j] = go;
        out[row * H + j] = hn;
        if (c_all) c_all[row * H + j] = cn;
        if (hprev) hprev[row * H + j] = hp;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!live[r]) continue;
    if (hT) hT[static_cast<long long>(seq[r]) * H + j] = hbuf[(T & 1) * kLstmSeqPerBlock * H + (grp * R + r) * H + j];
    if (cT) cT[static_cast<long long>(seq[r]) * H + j] = c[r];
  }
}

template <int H, int kLstmSeqPerBlock>
__global__ __launch_bounds__(kLstmThreads) void lstm_seq_bwd_kernel(
    const float* __restrict__ gates,     // [S*T, 4H] activated gates of the forward pass
    const float* __restrict__ c_all,     // [S*T, H]
    const float* __restrict__ c0,        // [S, H]
    const uint8_t* __restrict__ dones,   // [S*T] or nullptr
    const float* __restrict__ w_hh,      // [4H, H]
    const float* __restrict__ d_out,     // [S*T, H]  d loss / d h_t (from the layers above)
    float* __restrict__ d_gates,         // [S*T, 4H] d loss / d gate pre-activations
    int S, int T) {
  constexpr int G = 4 * H;
  constexpr int kGroups = kLstmThreads / H;
  constexpr int R = kLstmSeqPerBlock / kGroups;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w = smem;                                      // [4H][H] as stored
  float* dgb = smem + G * H;                            // [SB][4H]
  const int tid = threadIdx.x;
  const int j = tid % H;
  const int grp = tid / H;
  for (int idx = tid; idx < G * H; idx += kLstmThreads) w[idx] = w_hh[idx];
  const int seq0 = blockIdx.x * kLstmSeqPerBlock;
  int seq[R];
  bool live[R];
  float dh_next[R], dc_next[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int s = seq0 + grp * R + r;
    live[r] = s < S;
    seq[r] = live[r] ? s : S - 1;
    dh_next[r] = 0.0f;
    dc_next[r] = 0.0f;
  }
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    float keep[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = static_cast<long long>(seq[r]) * T + t;
      keep[r] = (dones && dones[row]) ? 0.0f : 1.0f;
      const float* grow = gates + row * G;
      const float gi = grow[0 * H + j], gf = grow[1 * H + j], gg = grow[2 * H + j], go = grow[3 * H + j];
      const float ct = c_all[row * H + j];
      const float c_in = (t > 0 ? c_all[(row - 1) * H + j] : c0[static_cast<long long>(seq[r]) * H + j]) * keep[r];
      const float dh = d_out[row * H + j] + dh_next[r];
      const float tc = tanhf(ct);
      const float d_o = dh * tc;
      const float dc = dc_next[r] + (dh * go) * (1.0f - tc * tc);
      const float dgi = (dc * gg) * (gi * (1.0f - gi));
      const float dgf = (dc * c_in) * (gf * (1.0f - gf));
      const float dgg = (dc * gi) * (1.0f - gg * gg);
      const float dgo = d_o * (go * (1.0f - go));
      dc_next[r] = (dc * gf) * keep[r];
      float* db = dgb + (grp * R + r) * G;
      db[0 * H + j] = dgi;
      db[1 * H + j] = dgf;
      db[2 * H + j] = dgg;
      db[3 * H + j] = dgo;
      if (live[r]) {
        float* drow = d_gates + row * G;
        drow[0 * H + j] = dgi;
        drow[1 * H + j] = dgf;
        drow[2 * H + j] = dgg;
        drow[3 * H + j] = dgo;
      }
    }
    __syncthreads();
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
#pragma unroll 4
    for (int row = 0; row < G; ++row) {
      const float wv = w[row * H + j];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(dgb[(grp * R + r) * G + row], wv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) dh_next[r] = acc[r] * keep[r];
    __syncthreads();
  }
}

// Sequences per block: 16 (4 per thread: every W_hh value read from LDS feeds 4 FMAs) when that still
// gives >= 256 blocks, otherwise fewer, so that a 1,024-sequence minibatch uses the whole chip instead
// of 64 CUs.
static int lstm_seq_per_block(int S, int H) {
  const int min_sb = kLstmThreads / H;                 // one sequence per thread group at least
  int sb = kLstmMaxSeqPerBlock;
  while (sb > min_sb && (S + sb - 1) / sb < 256) sb >>= 1;
  return sb;
}

template <int H, int SB>
static int launch_lstm_fwd_sb(float* gates, const float* w_hh, const float* h0, const float* c0,
                              const uint8_t* dones, float* out, float* c_all, float* hprev, float* hT,
                              float* cT, int S, int T, hipStream_t st) {
  if constexpr (SB < kLstmThreads / H) {
    return static_cast<int>(hipErrorInvalidValue);
  } else {
    const size_t shm = (static_cast<size_t>(4) * H * H + 2 * SB * H) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_fwd_kernel<H, SB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shm));
      if (e != hipSuccess) return static_cast<int>(e);
      attr_set = true;
    }
    const int grid = (S + SB - 1) / SB;
    hipLaunchKernelGGL((lstm_seq_fwd_kernel<H, SB>), dim3(grid), dim3(kLstmThreads), shm, st, gates, w_hh, h0,
                       c0, dones, out, c_all, hprev, hT, cT, S, T);
    RLG_RETURN_LAUNCH_STATUS();
  }
}

template <int H>
static int launch_lstm_fwd(float* gates, const float* w_hh, const float* h0, const float* c0,
                           const uint8_t* dones, float* out, float* c_all, float* hprev, float* hT,
                           float* cT, int S, int T, hipStream_t st) {
  switch (lstm_seq_per_block(S, H)) {
    case 16: return launch_lstm_fwd_sb<H, 16>(gates, w_hh, h0, c0, dones, out, c_all, hprev, hT, cT, S, T, st);
    case 8: return launch_lstm_fwd_sb<H, 8>(gates, w_hh, h0, c0, dones, out, c_all, hprev, hT, cT, S, T, st);
    default: return launch_lstm_fwd_sb<H, 4>(gates, w_hh, h0, c0, dones, out, c_all, hprev, hT, cT, S, T, st);
  }
}

template <int H, int SB>
static int launch_lstm_bwd_sb(const float* gates, const float* c_all, const float* c0, const uint8_t* dones,
                              const float* w_hh, const float* d_out, float* d_gates, int S, int T,
                              hipStream_t st) {
  if constexpr (SB < kLstmThreads / H) {
    return static_cast<int>(hipErrorInvalidValue);
  } else {
    const size_t shm = (static_cast<size_t>(4) * H * H + SB * 4 * H) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_bwd_kernel<H, SB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shm));
      if (e != hipSuccess) return static_cast<int>(e);
      attr_set = true;
    }
    const int grid = (S + SB - 1) / SB;
    hipLaunchKernelGGL((lstm_seq_bwd_kernel<H, SB>), dim3(grid), dim3(kLstmThreads), shm, st, gates, c_all, c0,
                       dones, w_hh, d_out, d_gates, S, T);
    RLG_RETURN_LAUNCH_STATUS();
  }
}

template <int H>
static int launch_lstm_bwd(const float* gates, const float* c_all, const float* c0, const uint8_t* dones,
                           const float* w_hh, const float* d_out, float* d_gates, int S, int T,
                           hipStream_t st) {
  switch (lstm_seq_per_block(S, H)) {
    case 16: return launch_lstm_bwd_sb<H, 16>(gates, c_all, c0, dones, w_hh, d_out, d_gates, S, T, st);
    case 8: return launch_lstm_bwd_sb<H, 8>(gates, c_all, c0, dones, w_hh, d_out, d_gates, S, T, st);
    default: return launch_lstm_bwd_sb<H, 4>(gates, c_all, c0, dones, w_hh, d_out, d_gates, S, T, st);
  }
}

}  // namespace rlg

extern "C" {

int rlg_lstm_supported(int hidden) { return (hidden == 16 || hidden == 32 || hidden == 64) ? 1 : 0; }

int rlg_lstm_seq_forward(float* gates, const float* w_hh, const float* h0, const float* c0,
                         const unsigned char* dones_or_null, float* out, float* c_all_or_null,
                         float* hprev_or_null, float* h_final_or_null, float* c_final_or_null,
                         int num_seqs, int seq_len, int hidden, void* stream) {
  using namespace rlg;
  if (num_seqs <= 0 || seq_len <= 0) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (hidden) {
    case 16: return launch_lstm_fwd<16>(gates, w_hh, h0, c0, dones_or_null, out, c_all_or_null, hprev_or_null,
                                        h_final_or_null, c_final_or_null, num_seqs, seq_len, st);
    case 32: return launch_lstm_fwd<32>(gates, w_hh, h0, c0, dones_or_null, out, c_all_or_null, hprev_or_null,
                                        h_final_or_null, c_final_or_null, num_seqs, seq_len, st);
    case 64: return launch_lstm_fwd<64>(gates, w_hh, h0, c0, dones_or_null, out, c_all_or_null, hprev_or_null,
                                        h_final_or_null, c_final_or_null, num_seqs, seq_len, st);
    default: return static_cast<int>(hipErrorInvalidValue);
  }
}

int rlg_lstm_seq_backward(const float* gates, const float* c_all, const float* c0,
                          const unsigned char* dones_or_null, const float* w_hh, const float* d_out,
                          float* d_gates, int num_seqs, int seq_len, int hidden, void* stream) {
  using namespace rlg;
  if (num_seqs <= 0 || seq_len <= 0) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (hidden) {
    case 16: return launch_lstm_bwd<16>(gates, c_all, c0, dones_or_null, w_hh, d_out, d_gates, num_seqs, seq_len, st);
    case 32: return launch_lstm_bwd<32>(gates, c_all, c0, dones_or_null, w_hh, d_out, d_gates, num_seqs, seq_len, st);
    case 64: return launch_lstm_bwd<64>(gates, c_all, c0, dones_or_null, w_hh, d_out, d_gates, num_seqs, seq_len, st);
    default: return static_cast<int>(hipErrorInvalidValue);
  }
}

}  // extern "C"
