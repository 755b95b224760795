/*
 * rlg_hip.h - C ABI of librlg_hip.so: the MI355X (gfx950) PPO hot path behind the
 * rl_games A2CAgent / torch_runner.Runner API.
 *
 * The reference (Denys88/rl_games v2.0.0) is 100 % Python on PyTorch and has no FFI of
 * its own; every entry point below therefore replaces a *Python* function or method of
 * the reference, cited as `file:line` relative to the reference checkout.  The binding a
 * reference maintainer would add is a ctypes stub; the in-tree host
 * code (the .py files of rl_games_amd/) is that binding.
 *
 * Conventions (all entry points):
 *   - plain device pointers + explicit sizes/strides, scalars by value, no torch types;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - launch only: never allocates, never synchronises, safe inside hipGraph capture;
 *   - returns 0 (hipSuccess) or the hipError_t of the failed launch / hipErrorInvalidValue
 *     for an unsupported shape; the Python binding raises RuntimeError on non-zero;
 *   - fp32 arithmetic is evaluated op by op (library built with -ffp-contract=off) in the
 *     order of the reference's eager PyTorch ops, so integer/mask/index results are
 *     bit-exact and fp32 results agree to rounding of reductions.
 *   - strides are in ELEMENTS, not bytes.
 */
#ifndef RLG_HIP_H_
#define RLG_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * GAE backward scan
 *   replaces rl_games/triton_kernels/gae_kernel.py: _gae_kernel :17-60, _pytorch_gae
 *   :63-80, _triton_gae :83-119, compute_gae :125-147; called from
 *   rl_games/common/a2c_common.py: A2CBase.discount_values :729-734.
 * ---------------------------------------------------------------------------------- */

/* General layout: rewards/values/advs [H, N, V] fp32 with arbitrary element strides,
 * dones [H, N] and last_dones [N] either fp32 (dones_are_float=1, what the reference
 * passes after `.float()`, a2c_common.py:1055-1056) or uint8/bool (dones_are_float=0).
 * strides17 (host array of 17 int64): r_t,r_e,r_v, v_t,v_e,v_v, d_t,d_e, lv_e,lv_v,
 * ld_e, a_t,a_e,a_v, q_t,q_e,q_v.  returns_or_null, when non-NULL, additionally
 * receives advs + values (a2c_common.py:1060) with strides q_*. */
int rlg_gae_strided(const float* rewards, const float* values, const void* dones,
                    const float* last_values, const void* last_dones, float* advs,
                    float* returns_or_null, int horizon, int num_envs, int value_size,
                    const long long* strides17, int dones_are_float, float gamma,
                    float gamma_tau, void* stream);

/* Env-major fast path = the ExperienceBuffer's native storage (flat index env*H + t,
 * the order swap_and_flatten01 produces: a2c_common.py:33-40): rewards/values [N, H] fp32
 * contiguous, dones [N, H] uint8 contiguous, value_size 1, 16-byte aligned bases,
 * horizon % 4 == 0 and 4 <= horizon <= 64 (rlg_gae_envmajor_supported). */
int rlg_gae_envmajor_supported(int horizon);

/* Number of wave tiles (= rows of the [*, 6] fp64 moment_partials array). */
int rlg_gae_envmajor_num_partials(int num_envs);

/* One pass: returns = A + v (a2c_common.py:1060), advantages = returns - v
 * (a2c_common.py:1598, both individually rounded) and, if moment_partials != NULL, per
 * tile fp64 {sum adv, sum adv^2, sum v, sum v^2, sum ret, sum ret^2} for the advantage
 * normaliser (a2c_common.py:1634) and value RunningMeanStd (a2c_common.py:1616-1620). */
int rlg_gae_envmajor_fused(const float* rewards, const float* values, const uint8_t* dones,
                           const float* last_values, const uint8_t* last_dones, float* returns,
                           float* advantages, double* moment_partials, int num_envs, int horizon,
                           float gamma, float gamma_tau, void* stream);

/* Profiling hooks used by bench.py (not part of the reference surface): hipEvent handles as
 * void*, and the fused GAE launch with start/stop events attached to that dispatch on the same
 * stream (hipExtLaunchKernelGGL) - elapsed time = the kernel's own duration, as rocprofv3 reports it. */
int rlg_event_create(void** event_out);
int rlg_event_destroy(void* event);
int rlg_event_elapsed_us(void* start, void* stop, float* us_out);   /* synchronises on `stop` */
int rlg_gae_envmajor_fused_timed(const float* rewards, const float* values, const uint8_t* dones,
                                 const float* last_values, const uint8_t* last_dones, float* returns,
                                 float* advantages, double* moment_partials, int num_envs,
                                 int horizon, float gamma, float gamma_tau, void* stream,
                                 void* ev_start, void* ev_stop);

/* Raw A_t only (the compute_gae return value) on the env-major layout. */
int rlg_gae_envmajor_raw(const float* rewards, const float* values, const uint8_t* dones,
                         const float* last_values, const uint8_t* last_dones, float* gae_out,
                         int num_envs, int horizon, float gamma, float gamma_tau, void* stream);

/* ------------------------------------------------------------------------------------
 * Rollout buffer (ExperienceBuffer) writes and per-step glue
 *   replaces rl_games/common/experience.py: ExperienceBuffer.update_data :433-456 and the
 *   per-step part of rl_games/common/a2c_common.py: A2CBase.play_steps :994-1051.
 *   Storage is env-major: field[env][t][row], flat index env*H + t (swap_and_flatten01
 *   order, a2c_common.py:33-40), so get_transformed_list (experience.py:508-522) is a view.
 * ---------------------------------------------------------------------------------- */

/* One launch for all `update_data(name, step, val)` calls of a step (a2c_common.py:1000-1011):
 * srcs[k] is [num_envs][row_bytes[k]] contiguous, dsts[k] the env-major field storage.
 * count <= 12.  srcs/dsts/row_bytes are HOST arrays of device pointers / sizes. */
int rlg_rollout_store_step(int count, const void* const* srcs, void* const* dsts,
                           const int* row_bytes, int num_envs, int horizon, int step,
                           void* stream);

/* Non-temporal stores for buffer rows of >= 128 bytes (observations); returns the previous setting,
 * enable < 0 only queries.  Default on. */
int rlg_rollout_store_streaming(int enable);

int rlg_rollout_post_step_num_blocks(int num_envs);

/* After vec_env.step: DefaultRewardsShaper (rl_games/common/tr_helpers.py:33-42; clamp_rewards bit 0 =
 * clamp to [rmin, rmax], bit 1 = log_val: log of the shaped reward behind the clamp), time-out bootstrap (a2c_common.py:1021-1023), update_data('rewards') (:1025),
 * current_rewards/current_shaped_rewards/current_lengths accumulate + zero-on-done
 * (:1027-1051).  Finished-episode sums go to ep_partials[step][block][2V+2] (fp64) and are
 * folded into the meters by rlg_episode_meters_update.  time_outs_kind: 0 none, 1 u8/bool,
 * 2 fp32.  live_rows (next_step autoreset mask, :1002-1006) may be NULL. value_size <= 8.
 * num_envs counts rows (envs x agents); with num_agents > 1 only the first agent row of an env
 * feeds the meters (all_done_indices[::num_agents], :1040-1044). */
int rlg_rollout_post_step(const float* rewards, const uint8_t* dones, const void* time_outs,
                          int time_outs_kind, const float* values, const float* live_rows,
                          float* rewards_buf, float* cur_rewards, float* cur_shaped,
                          float* cur_lengths, double* ep_partials, float shift, float scale,
                          float rmin, float rmax, int clamp_rewards, int bootstrap, float gamma,
                          int num_envs, int horizon, int value_size, int step, int num_agents,
                          void* stream);

/* Replays AverageMeter.update (rl_games/algos_torch/torch_ext.py:333-342) for the three
 * episode meters (game_rewards, game_shaped_rewards, game_lengths; a2c_common.py:1042-1044)
 * over steps 0..horizon-1.  current_sizes is int[3], finished_total an int64 counter. */
int rlg_episode_meters_update(const double* ep_partials, int horizon, int num_blocks,
                              int value_size, int max_size, float* mean_rewards,
                              float* mean_shaped, float* mean_lengths, int* current_sizes,
                              long long* finished_total, void* stream);

/* Rollout policy head (rl_games/algos_torch/models.py:348-364, denorm_value :58-60) fused with
 * the update_data writes of its outputs (a2c_common.py:1008-1009).  heads [N, ld]: column 0 the
 * critic value, columns 1..A the action means; noise [N, A] ~ N(0,1).  Writes actions/mus/
 * sigmas/neglogpacs/values of step `step` into the env-major buffer fields and the contiguous
 * actions [N, A] / de-normalised values [N] the env step and the post-step kernel consume. */
int rlg_rollout_policy_head(const float* heads, int ld_heads, const float* logstd, const float* noise,
                            const double* value_mean_or_null, const double* value_var_or_null,
                            float eps, float* actions_out, float* values_out, float* buf_actions,
                            float* buf_mus, float* buf_sigmas, float* buf_neglogp, float* buf_values,
                            /* optional [N, A]: rescale_actions(low, high, clamp(actions, -1, 1)) - what
                             * preprocess_actions hands to the env when clip_actions is set
                             * (a2c_common.py:725-733); act_low / act_high [A] */
                            float* env_actions_out_or_null, const float* act_low, const float* act_high,
                            int num_envs, int horizon, int actions_num, int step, void* stream);

/* play_steps_rnn zero-on-done: s[:, done_envs, :] = 0 (a2c_common.py:1150-1153).
 * states [layers][num_envs][units] contiguous. */
int rlg_rnn_zero_done_states(float* states, const uint8_t* dones, int layers, int num_envs,
                             int units, void* stream);

/* ------------------------------------------------------------------------------------
 * RunningMeanStd / advantage normalisation
 *   replaces rl_games/algos_torch/running_mean_std.py: RunningMeanStd.forward :69-114,
 *   _update_mean_var_count_from_moments :55-67; rl_games/algos_torch/torch_ext.py:
 *   get_mean_var_with_masks :182-191; ContinuousA2CBase.prepare_dataset value/advantage part
 *   (rl_games/common/a2c_common.py:1598-1634); GeneralizedMovingStats 'mean_std'
 *   (rl_games/algos_torch/moving_mean_std.py:52-61,:102-134,:136-150).
 * ---------------------------------------------------------------------------------- */

int rlg_column_moments_num_blocks(long long rows, int cols);

/* Per-block fp64 partial sums of a row-major fp32 [rows, cols] matrix:
 * partials[block][2*cols+1] = {sum x*m [cols], sum x*x*m [cols], sum m}; m = row mask or 1. */
int rlg_column_moments(const float* x, const float* row_mask_or_null, long long rows, int cols,
                       double* partials, int num_blocks, void* stream);

/* The same moments for num_segments consecutive bands of rows_per_segment rows in two launches -
 * the minibatches of one epoch, which every mini-epoch revisits unchanged (rl_games/common/
 * datasets.py:57-75: fixed slices): table[seg][2*cols+1] = {sum[cols], sumsq[cols], rows}.
 * partials: scratch of num_segments * num_blocks * (2*cols+1) doubles.  rlg_mlp_chain_forward folds
 * a table row into the running state in its prologue. */
int rlg_column_moments_segments(const float* x, long long rows_per_segment, int cols, int num_segments,
                                double* partials, int num_blocks, double* table, void* stream);

/* Chan merge of the batch moments into the fp64/int64 running state (running_mean_std.py:
 * 55-67).  mode 0: population variance, count += rows (:74-75,:83); mode 1: masked moments
 * of get_mean_var_with_masks, count += rows (:72,:83); mode 2: moments of the selected rows
 * only, count += #selected (values[valid], a2c_common.py:1609-1611).  `ticket` is a zero-initialised
 * device word owned by the caller (self-resetting arrival counter of the per-column blocks). */
int rlg_rms_update(const double* partials, int num_blocks, int cols, long long total_rows,
                   int mode, double* running_mean, double* running_var, long long* count,
                   unsigned int* ticket, void* stream);

/* y = clamp((x-mean)/sqrt(var+eps),-5,5) (mode 0, :112-113); denorm (mode 1, :106-107);
 * norm_only (mode 2, :110).  mean/var are the fp64 buffers, cast to fp32 like `.float()`. */
int rlg_rms_apply(const float* x, float* y, long long rows, int cols, const double* running_mean,
                  const double* running_var, float eps, int mode, void* stream);

/* Cross-rank synchronisation of ALL running normalisers of an agent, once per epoch (multi-GPU).
 *   replaces rl_games/common/a2c_common.py: _running_stats_totals :43-47, seed_stats_sync_snapshot
 *   :50-58, merge_rank_stats :61-93, broadcast_rank_stats :124-141; called from
 *   A2CBase.sync_running_stats :782-808 and _seed_stats_sync_snapshots :766-780.
 * The reference runs ~12 fp64 torch ops and THREE collectives per normaliser; here normaliser s
 * (state: means[s] / vars[s] fp64 [dims[s]], counts[s] int64 scalar) is the segment
 *   [count | first moments [dims[s]] | second moments [dims[s]]]
 * of one flat fp64 buffer (segments back to back, rlg_stats_sync_flat_size doubles), so an epoch's
 * exchange is pack -> ONE collective over the flat buffer -> apply.  num_segments <= 8.  Same fp64
 * operations in the same order as the reference (bit-identical statistics for identical reduced
 * sums); counts travel as doubles (exact below 2^53).  has_snapshot[s] == 0: the segment's whole
 * history is rank-local (no merge yet, :72-74) - its base is zero.
 * pack  mode 0: out = totals(state) - snapshot (this epoch's deltas);  mode 1: snapshot = totals(state)
 *       (after loading a checkpoint; out unused);  mode 2: out = raw state [count, mean, var] (broadcast mode).
 * apply mode 0: totals = snapshot + reduced;  mean = s1/n, var = max(s2/n - mean^2, 1e-8);  snapshot = totals;
 *       mode 2: state = reduced (rank 0's raw state after a broadcast). */
long long rlg_stats_sync_flat_size(int num_segments, const int* dims);
int rlg_stats_sync_pack(int num_segments, double* const* means, double* const* vars, long long* const* counts,
                        const int* dims, const int* has_snapshot, double* snapshot, double* out, int mode,
                        void* stream);
int rlg_stats_sync_apply(int num_segments, double* const* means, double* const* vars, long long* const* counts,
                         const int* dims, const int* has_snapshot, double* snapshot, const double* reduced, int mode,
                         void* stream);

int rlg_prepare_stats_bytes(void);

/* value_size==1 prepare_dataset statistics from the GAE kernel's partial moments: updates the
 * value RunningMeanStd with `values` then `returns` (a2c_common.py:1616-1619), computes the
 * advantage mean / unbiased std + 1e-8 (:1634) or the EMA statistics (moving_mean_std.py).
 * flags: 1 normalize_value, 2 normalize_advantage, 4 freeze_critic, 8 normalize_rms_advantage. */
/* Same 6 sums as the GAE kernel's partials (7 with a mask: + sum mask, every term weighted
 * by the mask) for batches that did not come out of rlg_gae_envmajor_fused. */
int rlg_triple_moments_num_blocks(long long batch);
int rlg_triple_moments(const float* advantages, const float* values, const float* returns,
                       const float* mask_or_null, long long batch, double* partials, int num_blocks,
                       void* stream);

/* stride 6: unmasked partials; stride 7: masked (valid-row statistics, a2c_common.py:1605-1615,
 * torch_ext.py:172-191). */
int rlg_prepare_finalize(const double* gae_partials, int num_tiles, int stride, long long batch,
                         int flags,
                         double* value_running_mean, double* value_running_var,
                         long long* value_count, float eps, float* ema_mean, float* ema_sqrs,
                         int* ema_step, float ema_decay, float ema_factor, float ema_max,
                         float ema_eps, void* stats_out, void* stream);

/* values/returns normalised with their respective statistics, advantages normalised
 * (a2c_common.py:1618-1619,:1634).  Outputs may alias the inputs. */
int rlg_prepare_apply(const float* values, const float* returns, const float* advantages,
                      float* values_out, float* returns_out, float* advantages_out, long long batch,
                      int flags, const void* stats, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused clipped-PPO loss forward + backward + KL
 *   replaces rl_games/algos_torch/models.py: ModelA2CContinuousLogStd epilogue :333-347,
 *   neglogp :361-364; rl_games/algos_torch/a2c_continuous.py: calc_losses :97-134,
 *   bound_loss/reg_loss :241-257, the KL block of calc_gradients :215-221;
 *   rl_games/common/common_losses.py: actor_loss :64-82, smoothed_actor_loss :39-61,
 *   default_critic_loss :16-29; rl_games/algos_torch/torch_ext.py: apply_masks :157-170,
 *   policy_kl :27-36; rl_games/common/datasets.py: PPODataset.update_mu_sigma :33-43.
 * ---------------------------------------------------------------------------------- */

int rlg_ppo_loss_num_blocks(int minibatch);
int rlg_ppo_loss_partials_per_block(int actions);

/* mu [mb,A], logstd [A] (fixed sigma, 'exp'), values [mb]; dataset slices actions [mb,A],
 * old_neglogp/advantages/old_values/returns [mb], old_mu/old_sigma [mb,A] (overwritten with
 * the new mu/sigma when write_back).  Emits d_mu [mb,A], d_values [mb] (already scaled by
 * 1/mb or mask/sum(mask)) and fp64 partials [blocks][rlg_ppo_loss_partials_per_block(A)].
 * mu/values/d_mu/d_values take row strides ld_* (elements) so they can be column views of a fused
 * [mb, 1+A] head buffer.  bound_kind 0 none / 1 'bound' / 2 'regularisation'.
 * use_smooth_clamp selects the actor loss of rl_games/common/common_losses.py:39-82: 0 = actor_loss (clipped PPO),
 * 1 = smoothed_actor_loss (`use_smooth_clamp: True`), 2 = `ppo: False` (a_loss = neglogp * advantage, either function's
 * else branch; round 6) - here and in every entry point / descriptor that carries the flag. */
int rlg_ppo_loss_fused(const float* mu, const float* logstd, const float* values,
                       const float* actions, const float* old_neglogp, const float* advantages,
                       const float* old_values, const float* returns, float* old_mu,
                       float* old_sigma, const float* mask_or_null, const float* mask_sum_or_null,
                       float* d_mu, float* d_values, double* partials, int minibatch, int actions_num,
                       int ld_mu, int ld_values, int ld_d_mu, int ld_d_values, float e_clip,
                       float critic_coef, float bounds_coef, int clip_value, int use_smooth_clamp,
                       int bound_kind, int write_back, void* stream);

/* Value-only loss of the central value network - replaces CentralValueTrain.calc_loss
 * (rl_games/algos_torch/central_value.py:262-276: common_losses.critic_loss + apply_masks).
 * partials [ceil(mb/256)][7] feed rlg_ppo_loss_finalize (actions_num 0, critic_coef 2 -> total =
 * mean c_loss); d_values [mb] = d mean(c) / d value. */
int rlg_value_loss(const float* values, const float* old_values, const float* returns,
                   const float* mask_or_null, const float* mask_sum_or_null, float* d_values, double* partials,
                   int minibatch, float e_clip, int clip_value, void* stream);

/* Discrete (Categorical) variant - replaces rl_games/algos_torch/a2c_discrete.py:
 * DiscreteA2CAgent.calc_gradients :121-209 with the model epilogues ModelA2C (models.py:95-111) and
 * ModelA2CMultiDiscrete (:153-179) incl. CategoricalMasked (common/extensions/distributions.py:24-47):
 * logits [mb, n] (row stride ld), n = sum(branch_sizes); actions int64 [mb, num_branches];
 * action_masks bool [mb, n] or NULL.  Emits d_logits [mb, n] (actor + entropy terms, scaled) and
 * d_values [mb]; partials [blocks][7] feed rlg_ppo_loss_finalize with actions_num = 0
 * (kl = 0.5 (old_nlp - nlp)^2, :192-198). */
int rlg_ppo_loss_discrete_num_blocks(int minibatch);
int rlg_ppo_loss_discrete(const float* logits, long long ld_logits, const float* values,
                          const long long* actions, const unsigned char* action_masks_or_null,
                          const int* branch_sizes, int num_branches, const float* old_neglogp,
                          const float* advantages, const float* old_values, const float* returns,
                          const float* mask_or_null, const float* mask_sum_or_null, float* d_logits,
                          float* d_values, double* partials, int minibatch, float e_clip, float critic_coef,
                          float entropy_coef, int clip_value, int use_smooth_clamp, void* stream);
/* The same launch over heads that live inside wider rows (the fused chain's [value | logits] head matrix and its
 * gradient): values / d_values with an element stride, d_logits with a row stride.  rlg_ppo_loss_discrete is this with
 * ld_values = ld_d_values = 1, ld_d_logits = n. */
int rlg_ppo_loss_discrete_strided(const float* logits, long long ld_logits, const float* values, long long ld_values,
                                  const long long* actions, const unsigned char* action_masks_or_null,
                                  const int* branch_sizes, int num_branches, const float* old_neglogp,
                                  const float* advantages, const float* old_values, const float* returns,
                                  const float* mask_or_null, const float* mask_sum_or_null, float* d_logits,
                                  long long ld_d_logits, float* d_values, long long ld_d_values, double* partials,
                                  int minibatch, float e_clip, float critic_coef, float entropy_coef, int clip_value,
                                  int use_smooth_clamp, void* stream);

/* scalars8 = {a_loss, c_loss, entropy, b_loss, kl, loss, sum(mask), 0}; d_logstd [A];
 * kl_slot_or_null receives the KL as well (e.g. the tail slot of the flat gradient arena);
 * d_mu_bias_or_null [A] / d_value_bias_or_null [1] receive sum_rows d_mu / d_values, i.e. the
 * bias gradients of the mu and value heads (what nn.Linear's backward computes with sum(0)). */
int rlg_ppo_loss_finalize(const double* partials, int num_blocks, int actions_num, int minibatch,
                          int masked, float critic_coef, float entropy_coef, float bounds_coef,
                          float* scalars8, float* d_logstd, float* kl_slot_or_null,
                          float* d_mu_bias_or_null, float* d_value_bias_or_null, void* stream);

/* ------------------------------------------------------------------------------------
 * Gradient truncation + Adam + adaptive learning rate on a flat arena
 *   replaces rl_games/common/a2c_common.py: trancate_gradients_and_step :493-514, update_lr
 *   :564-576, the per-minibatch schedule :1557-1563; rl_games/common/schedulers.py:
 *   AdaptiveScheduler.update :27-33; optimiser built at a2c_continuous.py:44-48.
 * ---------------------------------------------------------------------------------- */

int rlg_grad_norm_num_blocks(long long n);
/* Also increments *step_counter_or_null (the device-resident Adam step count) by one. */
int rlg_grad_sumsq(const float* grads, long long n, float grad_scale, double* partials,
                   int num_blocks, long long* step_counter_or_null, void* stream);

/* One optimiser step over n contiguous parameters.  grads are first scaled by grad_scale
 * (1/world_size) and by the clip coefficient min(1, max_norm/(norm+1e-6)) when
 * norm_partials is given.  step = *step_counter is the 1-based Adam step (device word, advanced
 * by rlg_grad_sumsq); lr_slots[(step-1)&1] is the lr of this step, the other slot receives the next
 * lr (schedule_kind 1: KL-adaptive on *kl * kl_scale).  stats_out[4] = {total_norm, clip_coef,
 * lr_used, lr_next}.  No per-call host scalars: the pair replays from a captured HIP graph.
 * skip_flag_or_null: device word (rlg_ipc_comm_error_word); non-zero = the gradients are invalid
 * (a failed in-graph all-reduce): parameters and moments stay untouched, the lr is carried over. */
int rlg_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                  const double* norm_partials_or_null, int norm_blocks, float grad_scale,
                  float max_norm, double* lr_slots, const long long* step_counter, double beta1,
                  double beta2, double eps, double weight_decay, int schedule_kind,
                  const float* kl_or_null, float kl_scale, double kl_threshold, double min_lr,
                  double max_lr, double lr_multiplier, float* stats_out_or_null,
                  const unsigned* skip_flag_or_null, void* stream);

/* rlg_adam_step that also leaves the fused chain's weight planes (both directions, the layout and buffer of
 * rlg_mlp_chain_pack_planes direction 2) for the NEW weights: the thread that updates a 4 x 4 block of a weight
 * matrix writes its rows as forward fragments and its columns as backward fragments (csrc/mlp_chain_bx.hip,
 * adam_pack_kernel) - one launch instead of rlg_adam_step + rlg_mlp_chain_pack_planes per optimiser step, and no pack
 * launch in front of the rollout forwards.  Same Adam arithmetic, same plane bytes.  hipErrorInvalidValue (use the
 * two launches) unless every weights[l] lies inside [params, params + n) at a multiple of 4 floats with
 * in_features[l] % 4 == 0; `planes` must have been packed in full once (the zero padding of the fragments). */
int rlg_adam_step_pack(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                       const double* norm_partials_or_null, int norm_blocks, float grad_scale, float max_norm,
                       double* lr_slots, const long long* step_counter, double beta1, double beta2, double eps,
                       double weight_decay, int schedule_kind, const float* kl_or_null, float kl_scale,
                       double kl_threshold, double min_lr, double max_lr, double lr_multiplier, float* stats_out_or_null,
                       const unsigned* skip_flag_or_null, int num_layers, const float* const* weights,
                       const int* in_features, const int* out_features, void* planes, void* stream);

/* ------------------------------------------------------------------------------------
 * Manual MLP backward helpers
 *   replace, per hidden layer of A2CBuilder's MLP (rl_games/algos_torch/network_builder.py:
 *   118-147, forward :498), aten's activation backward + the bias-gradient sum(0) of
 *   nn.Linear's backward that loss.backward() (a2c_continuous.py:211) runs.
 * ---------------------------------------------------------------------------------- */

int rlg_act_bwd_num_blocks(long long rows, int cols);

/* d_pre = d_out * act'(pre_act) (may alias d_out), partials[block][cols] = column sums of d_pre.
 * act_kind 0 identity, 1 elu(alpha 1), 2 relu, 3 tanh; 17 / 18 / 19: the same derivatives evaluated
 * from the layer OUTPUT act(z) passed as `pre_act` (in-place activations; aten's elu_backward with
 * is_result = true: h > 0 ? 1 : h + 1).  cols % 4 == 0, ld = row stride. */
int rlg_act_bwd_colsum(const float* d_out, const float* pre_act, float* d_pre, long long rows,
                       int cols, long long ld, int act_kind, double* partials, int num_blocks,
                       void* stream);

/* out[c] (+)= sum_b partials[b][c]: the bias gradient. */
int rlg_colsum_finalize(const double* partials, int num_blocks, int cols, float* out, int accumulate,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * fp32-MFMA MLP layer kernels (fused epilogues)
 *   replace torch.addmm + activation of every nn.Linear/activation pair built by
 *   A2CBuilder._build_sequential_mlp (rl_games/algos_torch/network_builder.py:118-147, forward
 *   :498) and the mu/value heads (:295-311).
 * ---------------------------------------------------------------------------------- */

/* ---- MLP forward / input-gradient GEMMs on f32 MFMA with fused epilogues -------------------
 * H = act(X W^T + b) (+ Z) replaces nn.Linear + activation of A2CBuilder._build_sequential_mlp and
 * the heads (rl_games/algos_torch/network_builder.py:118-147, :295-311, forward :498-512);
 * dZ_prev = (dZ W) * act'(Z_prev) replaces their autograd (grad_output.mm(weight) + elu_backward).
 * in_features (forward) / out_features (backward) must be a multiple of 4 (rlg_mlp_rowgemm_supported). */
int rlg_mlp_rowgemm_supported(int reduction_dim, long long leading_dim);
int rlg_mlp_linear_act_forward(const float* x, long long ldx, const float* w, const float* bias_or_null,
                               float* pre_act_or_null, float* out, long long ldo, int rows, int out_features,
                               int in_features, int act_kind, void* stream);
int rlg_mlp_linear_act_backward(const float* dz, long long lddz, const float* w, const float* z_prev_or_null,
                                float* dz_prev, long long ldo, int rows, int out_features, int in_features,
                                int act_kind, void* stream);

/* ---- MLP weight gradients on the matrix cores -----------------------------------------------
 * G_l [No, Mi] = dZ_l^T X_l for every nn.Linear of the policy MLP in ONE launch (+ one finalise
 * launch): replaces autograd's `grad_output.t().mm(input)` of A2CBuilder's layers
 * (rl_games/algos_torch/network_builder.py:118-147, heads :295-311; torch.nn.Linear backward).
 * fp32 in, fp32 out, fp32 accumulation.  Products: split-bf16 (each fp32 product as six exact bf16 plane
 * products on v_mfma_f32_16x16x32_bf16, truncation <= 3 * 2^-24 |x||y| - as accurate against fp64 as
 * exact products) by default; RLG_DW_BF16=0 in the environment selects exact f32 products
 * (v_mfma_f32_16x16x4_f32, 1.4x slower).
 * rlg_mlp_dw_plan fills plan4 = {bo, tiles_o, tiles_i, ksplit} for one layer and returns the
 * workspace size in floats (-1: shape unsupported, use the library GEMM); target_blocks <= 0: the
 * default workgroup count. */
long long rlg_mlp_dw_plan(int rows, int out_features, int in_features, int target_blocks, int* plan4);
/* colsum_* (num_colsums may be 0): bias gradients `grad_output.sum(0)` finished in the same finalise
 * launch from the per-block fp64 column sums of rlg_act_bwd_colsum (partials [blocks][cols]). */
/* The arguments of rlg_ppo_loss_finalize as a struct: rlg_mlp_dw_launch can fold the loss partials in
 * its finalise launch (a few extra workgroups next to the weight-gradient ones) instead of a launch of
 * its own - the head-bias / logstd gradients, the loss scalars and the KL are then ready when the
 * weight gradients are. */
typedef struct rlg_loss_finalize_desc {
  const double* partials;
  int num_blocks, actions_num, minibatch, masked;
  float critic_coef, entropy_coef, bounds_coef;
  float* scalars8;
  float* d_logstd;
  float* kl_slot_or_null;
  float* d_mu_bias_or_null;
  float* d_value_bias_or_null;
} rlg_loss_finalize_desc;

int rlg_mlp_dw_launch(int num_layers, const float* const* dz, const float* const* x, float* const* partial,
                      float* const* grad, const int* out_features, const int* in_features,
                      const int* plans4, int rows, int num_colsums, const double* const* colsum_partials,
                      const int* colsum_blocks, const int* colsum_cols, float* const* colsum_out,
                      const rlg_loss_finalize_desc* loss_finalize_or_null,
                      /* optional by-product: norm_partials[b] = sum (g * grad_scale)^2 over the gradient
                       * elements finalise block b wrote (*finalize_blocks_out blocks, host int) - the input of
                       * rlg_adam_step's clipping when THIS launch produces every gradient of the arena and
                       * nothing touches them before Adam (single GPU); it then also advances step_counter
                       * like rlg_grad_sumsq does. */
                      double* norm_partials_or_null, float grad_scale, long long* step_counter_or_null,
                      int* finalize_blocks_out_or_null, void* stream);

/* ---- the MLP as one vertically fused chain on f32 MFMA (csrc/mlp_chain.hip) ------------------
 * forward : heads = head(act(... act(norm(x) W_0^T + b_0) ...)) for a row tile, every layer in ONE
 *           launch with the activations resident in LDS; replaces A2CBuilder.Network.forward
 *           (rl_games/algos_torch/network_builder.py:447-512: actor_mlp + value/mu heads, arranged
 *           as one [1+A, K] last layer) and norm_obs in front of it (rl_games/algos_torch/models.py:
 *           54-56; RunningMeanStd eval formula rl_games/algos_torch/running_mean_std.py:112-113).
 *           act_out[l] (row stride act_ld[l]) receives layer l's output: required for the last
 *           layer (the heads), optional (training: what backward needs) for the hidden ones;
 *           xn_out (optional) receives the normalised observations [rows, in_0].
 * backward: dZ_{l-1} = (dZ_l W_l) * act'_{l-1}(H_{l-1}) down the chain from d_out = d loss / d heads
 *           (autograd's grad_output.mm(weight) + activation backward + bias sum(0) per nn.Linear);
 *           dz_out[l] (l < last) receives dZ_l, bias_partials[l] (optional) [num_blocks, out_l]
 *           fp64 per-workgroup column sums of dZ_l (finished by rlg_mlp_dw_launch's colsum items).
 * acts: 0 identity, 1 elu, 2 relu, 3 tanh (backward evaluates act' from the layer OUTPUT).
 * groups: 16-row groups per workgroup, 1 / 2 / 4 (0 = chosen from rows and direction:
 * rlg_mlp_chain_groups; rlg_mlp_chain_num_blocks takes the resolved value).  direction of rlg_mlp_chain_groups and
 * rlg_mlp_chain_bx_supported names the launch kind: 0 = inference forward, 1 = backward, 2 = training forward (activations
 * kept) - the 64-row split-product kernels take training launches from 8,192 rows, inference forwards from 16,384.
 * rlg_mlp_chain_lds_bytes: LDS of one workgroup (direction 0 forward, 1 backward), -1 if the
 * network does not fit the 160 KiB LDS with that many groups. */
int rlg_mlp_chain_prepare(void);   /* once per process, outside stream capture: raises the kernels' LDS limit */
int rlg_mlp_chain_groups(long long rows, int requested, int direction);
int rlg_mlp_chain_num_blocks(long long rows, int groups);
int rlg_mlp_chain_lds_bytes(int num_layers, const int* in_features, const int* out_features, int groups,
                            int direction);
/* tools only: the following forward launches record shader-clock stamps per phase into
 * buffer [blocks][4 waves][32] (int64); NULL switches it off. */
int rlg_mlp_chain_debug_stamps(long long* buffer);
/* HIP events (rlg_event_create) bound to the NEXT chain dispatch (forward or backward), one-shot; not for
 * launches inside a graph capture.  bench.py's roofline_fwd / roofline_bwd. */
/* Split-fp16 launches (round 6).  The weight-gradient launch sums over the batch rows, so it scales an operand by ONE power
 * of two per K-slice of rows.  rlg_mlp_chain_gradient_maxima: the next rlg_mlp_chain_backward launch, if it runs the
 * split-fp16 kernel, leaves per 64-row workgroup the largest magnitude of dZ of layer l (the last layer: of the d heads it
 * read) in entries[l * stride + workgroup] (plain stores; stride >= ceil(rows / 64)); one-shot.  The lean 16-row launches
 * (rlg_mlp_chain_backward_lean / _step_lean) take the same setter and leave one entry per 16-row workgroup (stride >=
 * ceil(rows / 16)); rows_per_entry of rlg_mlp_dw_gradient_maxima says which (64 or 16).
 * rlg_mlp_dw_gradient_maxima hands them to the NEXT rlg_mlp_dw_launch: dz_slot[k] = the layer of job k's dz, x_scale[k] = the
 * fixed power of two job k's x is split under (the forward's: 16 for hidden activations, 4096 for normalised
 * observations); every wave takes the largest entry over its own rows.  Without them the launch runs the bf16 form. */
int rlg_mlp_chain_gradient_maxima(float* entries, int stride);
int rlg_mlp_dw_gradient_maxima(const float* entries, int stride, int rows_per_entry, const int* dz_slot, const float* x_scale,
                               int num_layers);
/* plane products per fp32 product of the split-product chain kernels of this build: 3 (fp16 planes) or 6 (bf16 planes) */
int rlg_mlp_chain_split_products(void);
int rlg_mlp_chain_time_next(void* ev_start, void* ev_stop);

int rlg_mlp_chain_forward(int num_layers, const float* const* weights, const float* const* biases,
                          const int* in_features, const int* out_features, const int* acts,
                          float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                          const double* rms_mean_or_null, const double* rms_var, float rms_eps,
                          float* xn_out_or_null,
                          /* training-mode RunningMeanStd.forward (rl_games/algos_torch/running_mean_std.py:
                           * 69-84, called from models.py:54-56): fold this minibatch's column moments
                           * rms_batch = {sum[in], sumsq[in], rows} (rlg_column_moments_segments) into the
                           * state BEFORE normalising; the new state goes to the *_out buffers, which must
                           * differ from the inputs (every workgroup folds from the old state).  NULL
                           * rms_batch: normalise with the state as it is. */
                          const double* rms_batch_or_null, const long long* rms_count,
                          double* rms_mean_out, double* rms_var_out, long long* rms_count_out,
                          long long rows, int groups, void* pack_backward_planes_or_null,
                          const void* weight_planes_or_null, void* stream);
/* Arguments of the clipped-PPO loss (the parameter list of rlg_ppo_loss_fused as a struct): with a
 * non-NULL descriptor the BACKWARD launch evaluates the loss of its own row tile in front of its
 * prologue - no separate loss launch - i.e. it first writes d mu / d values (which must be views of the
 * d_out buffer this launch then reads), the mu/sigma write-back and one row of partials per workgroup
 * (rlg_mlp_chain_num_blocks(rows, resolved backward groups) rows of rlg_ppo_loss_partials_per_block
 * doubles).  minibatch must equal rows. */
typedef struct rlg_ppo_loss_desc {
  const float* mu;            /* [rows, A] view, row stride ld_mu */
  const float* logstd;        /* [A] */
  const float* values;        /* [rows] view, stride ld_values */
  const float* actions;       /* [rows, A] */
  const float* old_neglogp;   /* [rows] */
  const float* advantages;
  const float* old_values;
  const float* returns;
  float* old_mu;              /* [rows, A] read, then overwritten when write_back */
  float* old_sigma;
  const float* mask_or_null;
  const float* mask_sum_or_null;
  float* d_mu;                /* [rows, A] view, row stride ld_d_mu */
  float* d_values;            /* [rows] view, stride ld_d_values */
  double* partials;
  int minibatch, actions_num;
  int ld_mu, ld_values, ld_d_mu, ld_d_values;
  float e_clip, critic_coef, bounds_coef;
  int clip_value, use_smooth_clamp, bound_kind, write_back;
} rlg_ppo_loss_desc;

int rlg_mlp_chain_backward(int num_layers, const float* const* weights, const int* in_features,
                           const int* out_features, const int* acts, const float* const* act_in,
                           const long long* act_ld, const float* d_out, long long ld_dout,
                           float* const* dz_out, const long long* dz_ld, double* const* bias_partials_or_null,
                           const rlg_ppo_loss_desc* ppo_loss_or_null, long long rows, int groups,
                           const void* weight_planes_or_null, void* stream);

/* Forward + PPO loss + backward of one minibatch as ONE launch (round 4; csrc/mlp_chain.hip,
 * mlp_chain_step_pipe_kernel): the arguments of rlg_mlp_chain_forward in its training form (every act_out given) and of
 * rlg_mlp_chain_backward with a loss descriptor whose mu / values are the forward's heads and whose d_mu / d_values are
 * `d_out`.  For minibatches below 16,384 rows (a data-parallel rank's 4,096 - 8,192): the loss tile's inputs are
 * requested before the forward and arrive during it, and the launch boundary between the two halves is gone
 * (a2c_continuous.py:136-234: model forward, calc_losses, loss.backward() up to the weight-gradient GEMMs).  Same device
 * code as the two launches, same results.  Returns hipErrorNotSupported (801) when the shape is outside the kernel's
 * envelope - the caller then issues rlg_mlp_chain_forward and rlg_mlp_chain_backward. */
int rlg_mlp_chain_step(int num_layers, const float* const* weights, const float* const* biases,
                       const int* in_features, const int* out_features, const int* acts,
                       float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                       const double* rms_mean, const double* rms_var, float rms_eps, float* xn_out,
                       const double* rms_batch, const long long* rms_count, double* rms_mean_out,
                       double* rms_var_out, long long* rms_count_out, float* d_out, long long ld_dout,
                       float* const* dz_out, const long long* dz_ld, double* const* bias_partials_or_null,
                       const rlg_ppo_loss_desc* ppo_loss, long long rows, void* stream);

/* The 16-row launches of a data-parallel rank's minibatches (< 16,384 rows; rollouts of that size) in their LEAN form
 * (round 4; csrc/mlp_chain_lean.hip, mlp_chain_fwd_lean_kernel / mlp_chain_bwd_lean_kernel): the weights as FRAGMENTS (round 6: two fp16 planes of 64 w per group of 32 k values where rounds 4 - 5 had two fp32 chunks - the same bytes; csrc/split_f16.hpp) in
 * the order each of the 8 waves of a workgroup consumes them - one linear stream per wave across blocks and layers, zero
 * padding instead of out-of-range selects; the backward's fragments are the transposed matrices (one 16-byte load per lane and chunk where the row-major matrix
 * needs four strided dword loads).  Same maths as rlg_mlp_chain_forward / rlg_mlp_chain_backward in their 16-row form
 * (network_builder.py:447-512 and autograd's backward of it); round 6: three fp16 plane products per fp32 product - results
 * within the split kernels' tolerance of the pipelined exact-product kernels (bit-identical to them in a -DRLG_LEAN_F16=0 build).
 *   rlg_mlp_chain_frags_bytes: size of one direction's fragment buffer (0 forward, 1 backward), < 0: shape
 *     outside the format.
 *   rlg_mlp_chain_pack_frags / _both: weights -> fragments, one launch (the bias pointers are not read); to be repeated behind every change of
 *     the weights (an agent issues it behind its optimiser step).
 *   rlg_mlp_chain_forward_lean / rlg_mlp_chain_backward_lean: the launches; arguments as their namesakes without the
 *     weight pointers.  hipErrorNotSupported (801): shape outside the kernels' envelope (LDS, unaligned H / dZ
 *     rows) - the caller then uses rlg_mlp_chain_forward / rlg_mlp_chain_backward. */
long long rlg_mlp_chain_frags_bytes(int num_layers, const int* in_features, const int* out_features, int direction);
int rlg_mlp_chain_pack_frags(int num_layers, const float* const* weights, const float* const* biases_or_null,
                             const int* in_features, const int* out_features, int direction, void* frags, void* stream);
int rlg_mlp_chain_pack_frags_both(int num_layers, const float* const* weights, const float* const* biases,
                                  const int* in_features, const int* out_features, void* frags_fwd, void* frags_bwd_or_null,
                                  void* stream);
int rlg_mlp_chain_forward_lean(int num_layers, const float* const* biases, const int* in_features, const int* out_features,
                               const int* acts, float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                               const double* rms_mean, const double* rms_var, float rms_eps, float* xn_out,
                               const double* rms_batch, const long long* rms_count, double* rms_mean_out,
                               double* rms_var_out, long long* rms_count_out, long long rows, const void* frags,
                               void* stream);
/* forward + PPO loss + backward in ONE lean launch (the arguments of rlg_mlp_chain_step without the weight pointers;
 * minibatches of at most 16 rows x CUs; 801 otherwise) */
int rlg_mlp_chain_step_lean(int num_layers, const float* const* biases, const int* in_features, const int* out_features,
                            const int* acts, float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                            const double* rms_mean, const double* rms_var, float rms_eps, float* xn_out,
                            const double* rms_batch, const long long* rms_count, double* rms_mean_out,
                            double* rms_var_out, long long* rms_count_out, float* d_out, long long ld_dout,
                            float* const* dz_out, const long long* dz_ld, double* const* bias_partials,
                            const rlg_ppo_loss_desc* ppo_loss, long long rows, const void* frags_fwd, const void* frags_bwd,
                            void* stream);
int rlg_mlp_chain_backward_lean(int num_layers, const int* in_features, const int* out_features, const int* acts,
                                const float* const* act_in, const long long* act_ld, const float* d_out, long long ld_dout,
                                float* const* dz_out, const long long* dz_ld, double* const* bias_partials,
                                const rlg_ppo_loss_desc* ppo_loss, long long rows, const void* frags, void* stream);

/* Split-product form of the chain (csrc/mlp_chain_bx.hip): every fp32 product as three exact fp16 plane products on
 * v_mfma_f32_16x16x32_f16 (round 6, csrc/split_f16.hpp; six bf16 plane products in a -DRLG_BX_F16=0 build: results within
 * 3 * 2^-24 |x||w| per product of the exact-product kernels).  The weights
 * are split ONCE per optimizer step into plane fragments; the launch that is given them (weight_planes_or_null of
 * rlg_mlp_chain_backward, direction 1) uses the split kernel when rlg_mlp_chain_bx_supported says so and the
 * activation arrays are 16-byte aligned, else the exact-product kernel.  Same autograd nodes as above
 * (rl_games/algos_torch/network_builder.py:447-512).
 * pack_backward_planes_or_null of rlg_mlp_chain_forward: the forward launch of a training step splits the weights
 * for the backward launch that follows it (same weights) in extra workgroups at the end of its grid - no launch of its own.
 *   rlg_mlp_chain_planes_bytes : size of the fragment buffer for one direction (0 forward products, 1 backward) or,
 *                                direction 2, of ONE buffer with both (the backward fragments at
 *                                rlg_mlp_chain_planes_offset(..., 1))
 *   rlg_mlp_chain_pack_planes  : weights [out, in] fp32 -> fragments (one launch, all layers; direction 2: both
 *                                directions into the combined buffer - networks of up to 4 layers) */
long long rlg_mlp_chain_planes_bytes(int num_layers, const int* in_features, const int* out_features, int direction);
long long rlg_mlp_chain_planes_offset(int num_layers, const int* in_features, const int* out_features, int direction);
int rlg_mlp_chain_pack_planes(int num_layers, const float* const* weights, const int* in_features,
                              const int* out_features, int direction, void* planes, void* stream);
int rlg_mlp_chain_bx_supported(int num_layers, const int* in_features, const int* out_features, long long rows,
                               int groups, int direction);

/* ---- recurrent policy (BASELINE config #5) -------------------------------------------------
 * Sequence-persistent LSTM layer: replaces the per-timestep torch.nn.LSTM calls + done-state
 * resets of rl_games/common/layers/recurrent.py:26-58 (LSTMWithDones) as used by A2CBuilder
 * (rl_games/algos_torch/network_builder.py:447-512, rnn after the MLP) and autograd's BPTT.
 * gates [S*T, 4H] (row = seq*T + t) holds x_t W_ih^T + b_ih + b_hh on entry and the activated
 * gates (i, f, g, o) on exit; the state entering step t is zeroed where dones[seq*T+t] != 0.
 * hidden must satisfy rlg_lstm_supported (W_hh stays in LDS for the whole sequence). */
int rlg_lstm_supported(int hidden);
int rlg_lstm_seq_forward(float* gates, const float* w_hh, const float* h0, const float* c0,
                         const unsigned char* dones_or_null, float* out, float* c_all_or_null,
                         float* hprev_or_null, float* h_final_or_null, float* c_final_or_null,
                         int num_seqs, int seq_len, int hidden, void* stream);
/* d_gates [S*T, 4H] = d loss / d gate pre-activations given d_out = d loss / d h_t. */
int rlg_lstm_seq_backward(const float* gates, const float* c_all, const float* c0,
                          const unsigned char* dones_or_null, const float* w_hh, const float* d_out,
                          float* d_gates, int num_seqs, int seq_len, int hidden, void* stream);

/* ---- products too narrow for the MFMA kernels (csrc/mlp_narrow.hip; BASELINE config #5: obs 3, act 1) ----------
 * rlg_narrow_dx: dX [rows, in] = dZ [rows, out] W [out, in] for out <= 8 - autograd's grad_output.mm(weight) of the fused
 *   (value | mu) head (rl_games/algos_torch/network_builder.py:295-311, :506-512).
 * rlg_narrow_dw: grad [out, in] = dZ^T X for in <= 8, out <= 256 - grad_output.t().mm(input) of actor_mlp's first
 *   nn.Linear over a few observations (network_builder.py:118-147); partials: rlg_narrow_dw_blocks(rows) * out * in
 *   doubles of scratch (fp64 partial sums per workgroup, combined in a fixed order). */
int rlg_narrow_dx(const float* dz, long long lddz, const float* w, float* dx, long long lddx, long long rows,
                  int out_features, int in_features, void* stream);
int rlg_narrow_dw_blocks(long long rows);
int rlg_narrow_dw(const float* dz, long long lddz, const float* x, long long ldx, float* grad, double* partials,
                  long long rows, int out_features, int in_features, void* stream);

/* ------------------------------------------------------------------------------------
 * In-graph gradient all-reduce over peer-mapped device memory (csrc/ipc_allreduce.hip)
 *   replaces dist.all_reduce(SUM) of the flattened gradients in A2CBase.trancate_gradients_and_step
 *   (rl_games/common/a2c_common.py:493-509) and of the minibatch KL (:1559-1560, carried in a tail
 *   slot of the same arena).  One process per GPU: rlg_ipc_comm_create allocates the rank's staging
 *   memory and returns its hipIpcMemHandle_t (rlg_ipc_handle_bytes() bytes); the host exchanges the
 *   handles (any channel), rlg_ipc_comm_connect maps the peers; rlg_ipc_allreduce_sum is then ONE
 *   kernel launch (no host sync, capturable in a HIP graph) that leaves the identical rank-ordered
 *   sum in `data` on every rank.  The wait for the peers is bounded by wall time (default 600 s,
 *   RLG_IPC_TIMEOUT_S / rlg_ipc_comm_set_timeout; <= 0 = unbounded).  A launch that gives up, and every
 *   launch after it, is fail-safe: zeros instead of a sum of stale data in `data`, the sticky error word
 *   (rlg_ipc_comm_error_word) set - rlg_adam_step takes that word as its skip flag, so no parameter is
 *   touched by invalid gradients - and rlg_ipc_comm_status reports the launch ordinal.
 *   rlg_ipc_comm_create fails (the caller then uses RCCL) when the runtime has no fine-grained device
 *   memory: the protocol needs mid-kernel visibility across devices.
 *   Variant 1 (rlg_ipc_comm_set_variant / RLG_IPC_TWO_PHASE): reduce-scatter + all-gather - 2/P of the
 *   bytes per xGMI link, two synchronisations; same bits as the one-shot variant.
 * ---------------------------------------------------------------------------------- */
int rlg_ipc_handle_bytes(void);
int rlg_ipc_comm_create(int rank, int world, long long max_floats, void** comm_out, void* handle_out);
int rlg_ipc_comm_connect(void* comm, const void* all_handles /* world x handle bytes, rank-major */);
int rlg_ipc_comm_fine_grained(void* comm);
int rlg_ipc_comm_set_timeout(void* comm, double seconds);
int rlg_ipc_comm_set_variant(void* comm, int two_phase);
/* the settings in effect, environment defaults (RLG_IPC_TWO_PHASE, RLG_IPC_TIMEOUT_S) included; timeout 0 = unbounded.
 * Every rank must run the same variant: the host compares these across ranks before the first launch. */
int rlg_ipc_comm_get_config(void* comm, int* two_phase_out, double* timeout_s_out);
int rlg_ipc_comm_error_word(void* comm, unsigned** word_out);
int rlg_ipc_allreduce_sum(void* comm, float* data, long long n, void* stream);
/* The same launch with a by-product: norm_partials[rlg_ipc_allreduce_norm_blocks()] = per-workgroup sums of
 * (reduced x * grad_scale)^2 over the first norm_n elements (the gradients; the arena's tail slots are not),
 * and *step_counter += 1 - what rlg_grad_sumsq would compute in a launch of its own after the collective
 * (clip_grad_norm_ of the averaged gradients, a2c_common.py:498-512). */
int rlg_ipc_allreduce_norm_blocks(void);
int rlg_ipc_allreduce_sum_norm(void* comm, float* data, long long n, double* norm_partials, long long norm_n,
                               float grad_scale, long long* step_counter_or_null, void* stream);
int rlg_ipc_comm_status(void* comm, unsigned* launches_out, unsigned* timed_out_launch_out);
int rlg_ipc_comm_destroy(void* comm);

/* ---- RCCL behind the C ABI (csrc/rccl_wrap.hip) ---------------------------------------------------
 * "plus an RCCL wrapper taking ncclComm_t".  The gradient all-reduce of
 * A2CBase.trancate_gradients_and_step (rl_games/common/a2c_common.py:493-509) as a launch on the caller's stream
 * - capturable into the mini-epoch HIP graph, unlike a collective issued through torch.distributed - for nodes on
 * which the hipIpc kernel (rlg_ipc_*) is not available.  `comm` is an ncclComm_t; the unique id (rlg_rccl_unique_id_bytes()
 * bytes, from rank 0's rlg_rccl_get_unique_id) travels through whatever channel the ranks share.  librccl is resolved
 * with dlopen at first use: rlg_rccl_available() == 0 and hipErrorNotSupported from the others when there is none.
 * Every rank receives the same bits; in place. */
int rlg_rccl_available(void);
int rlg_rccl_unique_id_bytes(void);
int rlg_rccl_get_unique_id(void* id_out);
int rlg_rccl_comm_create(const void* unique_id, int rank, int world, void** comm_out /* ncclComm_t */);
int rlg_rccl_allreduce_sum(void* comm /* ncclComm_t */, float* data, long long n, void* stream);
int rlg_rccl_allreduce_sum_f64(void* comm /* ncclComm_t */, double* data, long long n, void* stream);
int rlg_rccl_comm_destroy(void* comm /* ncclComm_t */);

#ifdef __cplusplus
}
#endif

#endif /* RLG_HIP_H_ */
