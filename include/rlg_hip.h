/*
 * rlg_hip.h - C ABI of librlg_hip.so: the MI355X (gfx950) PPO hot path behind the
 * rl_games A2CAgent / torch_runner.Runner API.
 *
 * The reference (Denys88/rl_games v2.0.0) is 100 % Python on PyTorch and has no FFI of
 * its own; every entry point below therefore replaces a *Python* function or method of
 * the reference, cited as `file:line` relative to the reference checkout.  The binding a
 * reference maintainer would add is a ctypes stub; the in-tree host
 * code (rl_games_amd/*.py) is that binding.
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

/* Raw A_t only (the compute_gae return value) on the env-major layout. */
int rlg_gae_envmajor_raw(const float* rewards, const float* values, const uint8_t* dones,
                         const float* last_values, const uint8_t* last_dones, float* gae_out,
                         int num_envs, int horizon, float gamma, float gamma_tau, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* RLG_HIP_H_ */
