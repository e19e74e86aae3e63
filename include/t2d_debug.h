/* t2d_debug.h -- test and measurement hooks of the MI355X-native batched tactics2d step.
 *
 * NOT part of the product ABI.  libt2d_hip.so (include/t2d.h) exports none of these; they exist only in
 * libt2d_hip_debug.so = the same sources built with -DT2D_DEBUG_HOOKS (python -m tactics2d_amd.build --debug-lib), which
 * exports everything t2d.h declares plus what is declared here.  Loaded by tests/, bench.py's closed_loop leg and scripts/
 * through tactics2d_amd/debug.py; no product module imports that.  The reference's operator API has no counterpart of any of
 * this (fault injection, a delayed collective, a stand-in policy, placement maps): they are how the build is TESTED.
 *
 * The -DT2D_DEBUG_HOOKS build differs from the product in exactly: these entry points, tactics2d_amd/csrc/t2d_loop.hip, and two
 * reads inside the step kernel (the placement map of a single launch, the fault word of a chained one).
 */
#ifndef T2D_DEBUG_H_
#define T2D_DEBUG_H_
#include "t2d.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Placement of the step launch: entry b = logical workgroup (the envs [g * epb, (g + 1) * epb), epb = 256 / padded
 * max_agents unless the geometry budget narrowed it) | wave rotation 0..3 << 16 that physical workgroup b steps.  Must be a
 * permutation of the launch's workgroups; NULL / 0 restores the identity.  Results never depend on it -- the hardware
 * places workgroup b on XCD b mod 8 and its waves on fixed SIMDs, so the map only decides which envs share a SIMD and
 * which XCD gets the expensive ones.  Reset by every t2d_set_*_geometry.  Host memory.                                  */
int t2d_debug_set_step_placement(t2d_pool* pool, const uint32_t* map_host, int32_t n_workgroups);

/* Test hook: the CHAIN launches of t2d_step_n enqueued from now on break ONE hand-off on purpose -- workgroup 1 posts its step
 * 1 with a foreign XCC id (kind 1: what a consumer on another XCD would see) or not at all (kind 2: its consumer's bounded
 * wait runs out after ~0.2 s); kind 3: it posts its step 0 with a foreign XCC id, so that the failure is on record while the
 * fragment's first step is still being dispatched (a grid larger than the device holds: the checkpoint must still be complete);
 * 0 = off.  Needs a pool of >= 2 step workgroups and fragments of >= 3 steps to have any effect. */
int t2d_debug_chain_fault(t2d_pool* pool, int32_t kind);

/* Test hook: occupies the pool's gather stream for `microseconds` (one idle wave), so that the gathers enqueued after it
 * start late -- what a slow peer does to the collective.  tests/test_gpu_dist.py uses it to check that a step about to
 * overwrite a record slot really waits for the gather that still has to read it.                                       */
int t2d_debug_delay_gather(t2d_pool* pool, int32_t microseconds);

/* Which instantiation of the step / event kernel the LAST launch of this process took: its template arguments as written at the
 * launch site in tactics2d_amd/csrc/t2d_collide.hip, e.g. "(true, 1, false, true)" = the chained fused step, fast integrator.
 * tests/test_gpu_forms.py drives one recipe per launch site and holds the set it sees against the list in the source file
 * (DESIGN.md 4.2c).  Not thread-safe; a static string.                                                                    */
const char* t2d_debug_last_step_kernel(void);

/* ---- the closed loop (measurement / test helpers; tactics2d_amd/csrc/t2d_loop.hip) -----------------------------------------
 * The reference's callers run  action = policy(obs); obs, reward, ... = env.step(action)  (envs/parking.py:219-256 inside the
 * tutorial's training loop).  On the device that is: a policy kernel that reads the state the previous step left behind and
 * writes an [N, 2] (steering, accel) tensor -> t2d_step reading it in place (t2d_bind_actions_strided) -> the policy again,
 * with no host synchronisation; the envs cut into groups -- one pool and one stream each -- so that one group's policy,
 * start-up and tail overlap the other groups' busy middle (env groups: tactics2d_amd/pipeline.py, t2d_step_groups).
 *   t2d_debug_feedback_policy   a STAND-IN policy, one launch: per participant accel = clip(k_speed (v_target - speed), -3, 2),
 *                               steering = k_steer sin(0.05 x + 0.08 y + heading); act_out_dev = f32 [N][2] (steering, accel).
 *   t2d_debug_closed_loop_*     n iterations of (that policy, t2d_step) per group enqueued by ONE host call -- launcher 0: from
 *                               the calling thread, round the groups step by step; 1: one host thread per group; 2: one captured
 *                               hipGraph per group holding graph_steps iterations, replayed (a replay rewrites the record-ring
 *                               slots of the capture: pick graph_steps = a multiple of T2D_RECORD_RING, or read T2D_F_STATUS /
 *                               T2D_F_REWARD).  create binds each pool's actions to its act_out_dev[g]; run returns when
 *                               everything is enqueued (t2d_sync / stream synchronisation waits for it); results are those of
 *                               the same policy and t2d_step calls on one pool holding all the envs. */
typedef struct t2d_closed_loop t2d_closed_loop;
int t2d_debug_feedback_policy(t2d_pool* pool, float* act_out_dev, float v_target, float k_speed, float k_steer, void* hip_stream);
int t2d_debug_closed_loop_create(t2d_pool* const* pools, void* const* hip_streams, float* const* act_out_dev, int32_t n_groups,
                                 int32_t interval_ms, int32_t launcher, int32_t graph_steps, t2d_closed_loop** out);
int t2d_debug_closed_loop_run(t2d_closed_loop* loop, int32_t n_steps);
int t2d_debug_closed_loop_destroy(t2d_closed_loop* loop);

#ifdef __cplusplus
}
#endif
#endif /* T2D_DEBUG_H_ */
