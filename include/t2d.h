/*
 * t2d.h -- C ABI of libt2d_hip.so: the MI355X-native batched env.step() hot path
 * for tactics2d (physics integrators + collision / out-of-bound / off-lane events).
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain
 * pointers / ints / sizes, returns an int status (0 = T2D_OK) and never throws.
 * No torch types appear here; PyTorch only ever sees the raw device pointers
 * returned by t2d_get_field().
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * the tactics2d repository root, v0.1.9rc3):
 *
 *   t2d_set_param_table   <- SingleTrackKinematics.__init__  physics/single_track_kinematics.py:62-124
 *                            SingleTrackDynamics.__init__    physics/single_track_dynamics.py:58-138
 *                            PointMass.__init__              physics/point_mass.py:33-81
 *                            Vehicle/Cyclist/Pedestrian templates participant/element/participant_template.py:42-257
 *   t2d_reset             <- ParticipantBase.reset / Trajectory.add_state   participant/trajectory/trajectory.py:115-149
 *                            _ParkingScenarioManager.reset   envs/parking.py:397-441
 *   t2d_integrate         <- PhysicsModelBase.step           physics/physics_model_base.py:28
 *                            SingleTrackKinematics.step/_step physics/single_track_kinematics.py:126-198
 *                            SingleTrackDynamics.step/_step  physics/single_track_dynamics.py:140-251
 *                            PointMass.step/_step_newton     physics/point_mass.py:83-175,209-232
 *   t2d_set_static_geometry <- StaticCollision.reset         traffic/event_detection/collision.py:45-46
 *                            OutBound.reset                  traffic/event_detection/out_bound.py:50-65
 *   t2d_set_lane_geometry <- OffLane.reset                   traffic/event_detection/off_lane.py:19-20
 *   t2d_collide           <- Vehicle.get_pose                participant/element/vehicle.py:263-281
 *                            StaticCollision.update          traffic/event_detection/collision.py:37-43
 *                            DynamicCollision.update         traffic/event_detection/collision.py:18-25 (intended semantics)
 *                            OutBound.update                 traffic/event_detection/out_bound.py:37-48
 *                            OffLane.update                  traffic/event_detection/off_lane.py:16-17 (stub; build-defined:
 *                                                            not union(lane polygons).contains(pose), the predicate of
 *                                                            out_bound.py:37-48 applied to the lanes)
 *   t2d_snapshot/restore  <- ParkingEnv.reset / _ParkingScenarioManager.reset       envs/parking.py:262-298,397-441
 *   t2d_set_target_areas  <- Arrival.reset                traffic/event_detection/arrival.py:49-51
 *                            Arrival.update / NoAction.update (IoU)   arrival.py:32-47, no_action.py:32-53
 *   t2d_lidar_config/scan <- SingleLineLidar.__init__ / _scan_obstacles    sensor/lidar.py:33-57,128-221
 *   t2d_check_status      <- _ParkingScenarioManager.check_status           envs/parking.py:361-392
 *   t2d_step              <- _ParkingScenarioManager.update + check_status  envs/parking.py:352-392
 *                            ParkingEnv.step terminated/truncated/reward    envs/parking.py:219-256,148-161
 *                            TimeExceed.update               traffic/event_detection/time_exceed.py:26-33
 *   t2d_step_host         <- ParkingEnv.step as its caller sees it: host action in, host 5-tuple out
 *                            envs/parking.py:219-256, _get_infos / _get_relative_pose :190-217
 *
 * Threading: one host thread per pool.  t2d_integrate / t2d_collide / t2d_step are
 * asynchronous on the supplied hipStream_t (passed as void*; NULL = the null stream).
 * Nothing synchronises implicitly except t2d_download / t2d_upload / t2d_reset /
 * t2d_sync / the t2d_set_* calls (which copy from host memory), and those wait for THIS
 * pool's work only -- the streams it was launched on since the last such call plus its own
 * internal streams -- never for the whole device: other pools (env groups) and a policy
 * running on other streams keep going.
 *
 * Non-finite values.  The integrators follow the reference: np.clip(nan, lo, hi) is nan (single_track_kinematics.py:192-193),
 * np.clip(+-inf) the bound, np.mod(+-inf, 2 pi) nan -- a NaN action or state makes the participant's state NaN exactly where
 * numpy would (tests/golden/nonfinite.npz, made by importing the reference).  What the reference leaves to GEOS is BUILD-DEFINED
 * here: a participant whose pose (x, y or heading) is not finite takes no part in event detection -- its flags are 0 and nobody
 * collides with it --, its IoU events are not evaluated (T2D_F_IOU = NaN), as a lidar ego it sees nothing (+inf on every beam)
 * and as a lidar obstacle it is skipped; an IDM participant with a non-finite pose leads nobody (comparisons with NaN are
 * false).  Nothing hangs and no other participant's result changes (tests/test_gpu_nonfinite.py).  t2d_step_host rejects
 * non-finite actions when given an action box (T2D_ERR_ACTION), as `action_space.contains` does.
 *
 * Ownership: the pool owns every device buffer.  Host pointers passed in are read
 * during the call and never retained.  Device pointers handed out by t2d_get_field
 * stay valid until t2d_destroy.
 */
#ifndef T2D_H_
#define T2D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2D_ABI_VERSION 12

/* ---- status codes --------------------------------------------------------------- */
#define T2D_OK            0
#define T2D_ERR_INVALID   1   /* bad argument (null, out of range, inconsistent sizes)   */
#define T2D_ERR_HIP       2   /* a HIP runtime call failed; see t2d_last_error           */
#define T2D_ERR_NOMEM     3
#define T2D_ERR_STATE     4   /* call order violated (e.g. step before param table)      */
#define T2D_ERR_GEOMETRY  5   /* polygon not convex / degenerate / too many vertices     */
#define T2D_ERR_ACTION    6   /* t2d_step_host: an action outside the caller's action box (InvalidAction, envs/parking.py:235-236); nothing was stepped */

/* ---- physics model ids (column T2D_P_MODEL of a parameter row) --------------------- */
#define T2D_MODEL_KINEMATICS 0   /* SingleTrackKinematics */
#define T2D_MODEL_DYNAMICS   1   /* SingleTrackDynamics   */
#define T2D_MODEL_POINTMASS  2   /* PointMass, newton back-end */
#define T2D_MODEL_POINTMASS_EULER 4   /* PointMass, euler back-end (point_mass.py:177-207, backend="euler"): vx, vy AND heading are state
                                       * (the clipped speed is re-projected onto the previous sub-step's heading).  A rare,
                                       * selectable second back-end: integrated by the side kernel that also takes
                                       * T2D_MODEL_DRIFT (one launch ahead of the step launch; such pools are not chained) */
#define T2D_MODEL_DRIFT      3   /* SingleTrackDrift (Pacejka tyres, default Tire constants); extra state
                                  * T2D_F_OMEGA_F / T2D_F_OMEGA_R; integrated by its own kernel */

/* ---- shape kinds (column T2D_P_SHAPE) ---------------------------------------------- */
#define T2D_SHAPE_OBB    0   /* Vehicle / Cyclist / Other: length x width box             */
#define T2D_SHAPE_CIRCLE 1   /* Pedestrian: (centre, radius = width / 2)                  */

/* ---- parameter-table row: T2D_PARAM_COLS doubles per participant type ---------------
 * Ranges are ALREADY normalised by the host exactly like the reference constructors
 * (A.1 in SURVEY.md); an unbounded range has its bit cleared in T2D_P_RANGE_FLAGS.  */
enum {
    T2D_P_MODEL = 0,       /* T2D_MODEL_*                                   */
    T2D_P_LF = 1,
    T2D_P_LR = 2,
    T2D_P_WB = 3,          /* lf + lr, summed on the host like the reference */
    T2D_P_STEER_LO = 4,
    T2D_P_STEER_HI = 5,
    T2D_P_SPEED_LO = 6,
    T2D_P_SPEED_HI = 7,
    T2D_P_ACCEL_LO = 8,
    T2D_P_ACCEL_HI = 9,
    T2D_P_RANGE_FLAGS = 10, /* bit0 steer bounded, bit1 speed bounded, bit2 accel bounded */
    T2D_P_MASS = 11,
    T2D_P_MASS_HEIGHT = 12,
    T2D_P_MU = 13,
    T2D_P_IZ = 14,
    T2D_P_CF = 15,
    T2D_P_CR = 16,
    T2D_P_DELTA_T_MS = 17,  /* integer-valued: sub-step in ms (reference _DELTA_T = 5)  */
    T2D_P_SHAPE = 18,       /* T2D_SHAPE_*                                              */
    T2D_P_LENGTH = 19,      /* OBB length (m)                                           */
    T2D_P_WIDTH = 20,       /* OBB width (m); circle radius = width / 2                 */
    T2D_P_RESERVED0 = 21,
    T2D_P_RESERVED1 = 22,
    T2D_P_RESERVED2 = 23,
    T2D_PARAM_COLS = 24,
    /* T2D_MODEL_DRIFT rows reuse four columns the other models leave alone: */
    T2D_P_DRIFT_TSB = 15,    /* T_sb, brake-torque split   (single_track_drift.py:102) */
    T2D_P_DRIFT_TSE = 16,    /* T_se, engine-torque split  (:103)                      */
    T2D_P_DRIFT_RADIUS = 22, /* effective wheel radius (m) (:101)                      */
    T2D_P_DRIFT_IYW = 23,    /* wheel inertia I_yw         (:106)                      */
    /* ... and rows of the other models carry two DERIVED values there, written by the library (whatever the caller put
     * into the reserved columns is ignored): */
    T2D_P_DT_S = 22,         /* (double)delta_t_ms / 1000: the Euler sub-step in seconds (_step's `dt`)            */
    T2D_P_SUBSTEPS = 23      /* (interval_ms / delta_t_ms) * 65536 + interval_ms % delta_t_ms for the interval of the
                              * launch in flight (a 32-thread kernel refreshes it when the interval changes)        */
};
#define T2D_RANGE_STEER 1
#define T2D_RANGE_SPEED 2
#define T2D_RANGE_ACCEL 4
#define T2D_MAX_TYPES 32

/* ---- pool fields (t2d_get_field / t2d_download / t2d_upload) ------------------------
 * Per-participant fields hold N = n_env * max_agents elements, env-major
 * (index = env * max_agents + agent).  Per-env fields hold n_env elements.         */
enum {
    T2D_F_X = 0,        /* f32[N]  env-local x (m)                                   */
    T2D_F_Y = 1,        /* f32[N]                                                    */
    T2D_F_HEADING = 2,  /* f32[N]  stored heading, np.mod(phi, 2*pi)                 */
    T2D_F_SPEED = 3,    /* f32[N]  signed scalar speed                               */
    T2D_F_VX = 4,       /* f32[N]  written by kinematics / point-mass only           */
    T2D_F_VY = 5,       /* f32[N]                                                    */
    T2D_F_ACT0 = 6,     /* f32[N]  accel (vehicles) | ax (point mass)                */
    T2D_F_ACT1 = 7,     /* f32[N]  steer (vehicles) | ay (point mass)                */
    T2D_F_IDS = 8,      /* u32[N]  model_id | type_id<<8 | active<<16                */
    T2D_F_FLAGS = 9,    /* u32[N]  event bits, see T2D_FLAG_*                        */
    T2D_F_APPLIED0 = 10,/* f32[N]  clipped accel actually applied (State.accel)      */
    T2D_F_APPLIED1 = 11,/* f32[N]  clipped steer actually applied                    */
    T2D_F_ENV_FLAGS = 12,  /* u32[E]  OR of the participants' flags                  */
    T2D_F_CNT_STEP = 13,   /* i32[E]  ScenarioManager.cnt_step                       */
    T2D_F_FRAME_MS = 14,   /* i32[E]  State.frame of the env (ms)                    */
    T2D_F_STATUS = 15,     /* u8[E*4] scenario_status, traffic_status, terminated, truncated */
    T2D_F_REWARD = 16,     /* f32[E]                                                  */
    T2D_F_RECORD = 17,     /* u32[T2D_RECORD_RING][E][2] packed per-env result records {reward bits, status
                              word}, a ring: t2d_step number k (0-based since create) writes slot
                              k % T2D_RECORD_RING, so a collective can ship the records of the last K
                              steps in one message while the following steps already run          */
    T2D_F_IOU = 18,        /* f32[E]  IoU(ego pose, target) of the last step; NaN = not evaluated (None) */
    T2D_F_CNT_NO_ACTION = 19, /* i32[E] NoAction.cnt_no_action                          */
    T2D_F_LIDAR = 20,      /* f32[E][n_beams] last t2d_lidar_scan into the pool's own buffer (size set by
                              t2d_lidar_config; +inf = no return)                                 */
    T2D_F_LEADER = 21,     /* i32[N]  agent index (inside the env) of the IDM leader chosen by the last
                            *         t2d_idm_actions, -1 = none / participant not IDM-controlled   */
    T2D_F_OMEGA_F = 22,    /* f32[N]  SingleTrackDrift front wheel angular speed (rad/s); 0 after t2d_reset */
    T2D_F_OMEGA_R = 23,    /* f32[N]  rear wheel                                                        */
    T2D_F_COUNT = 24
};

/* ---- per-participant / per-env event bits ------------------------------------------- */
#define T2D_FLAG_COLLISION_DYNAMIC 1u   /* OBB/circle intersects another active participant */
#define T2D_FLAG_COLLISION_STATIC  2u   /* intersects a static polygon (StaticCollision)    */
#define T2D_FLAG_OUT_BOUND         4u   /* not boundary.contains(pose)  (OutBound)          */
#define T2D_FLAG_OFF_LANE          8u   /* not union(lanes).contains(pose); build-defined, DESIGN.md */

/* ---- ScenarioStatus / TrafficStatus values: traffic/status.py:10-61 ----------------- */
#define T2D_TRAFFIC_NO_ACTION_QUIRK 5  /* parking.py:373 stores ScenarioStatus.NO_ACTION (5) in traffic_status */
#define T2D_SCENARIO_NORMAL        1
#define T2D_SCENARIO_COMPLETED     2
#define T2D_SCENARIO_TIME_EXCEEDED 3
#define T2D_SCENARIO_OUT_BOUND     4
#define T2D_SCENARIO_NO_ACTION     5
#define T2D_SCENARIO_FAILED        6
#define T2D_TRAFFIC_NORMAL            1
#define T2D_TRAFFIC_COLLISION_STATIC  3
#define T2D_TRAFFIC_COLLISION_DYNAMIC 4
#define T2D_TRAFFIC_OFF_LANE          6

/* ---- geometry limits ----------------------------------------------------------------- */
#define T2D_MAX_POLY_VERTS 8      /* static / lane polygons: convex, 3..8 vertices (5..8: evaluated as a fan of quads) */
#define T2D_RECORD_RING 64      /* slots of the per-env result-record ring (T2D_F_RECORD): two gather fragments of 32 steps */
#define T2D_MAX_AGENTS 256        /* participants per env                                 */

/* ---- status / reward configuration (t2d_set_status_config) --------------------------- */
typedef struct t2d_status_config {
    int32_t max_step;            /* TimeExceed.max_step; <= 0 disables (time_exceed.py:20) */
    int32_t ego_index;           /* agent whose flags drive the env status (0)            */
    int32_t check_dynamic;       /* include participant-participant collision in status   */
    int32_t check_off_lane;      /* include build-defined off-lane in status              */
    float reward_collision;      /* -5  envs/parking.py:151-152                           */
    float reward_time_exceed;    /* -1  envs/parking.py:153-157                           */
    float reward_out_bound;      /* -5  envs/parking.py:158-159                           */
    float reward_completed;      /* +5  envs/parking.py:160-161                           */
    float time_penalty_scale;    /* 0.001: -tanh(cnt_step / max_step) * scale  :163       */
    /* ---- IoU events of the ego (scope rows f1), all off by default ------------------------- */
    int32_t check_arrival;       /* Arrival.update arrival.py:32-47: IoU(pose, target) >= threshold
                                    -> COMPLETED (needs t2d_set_target_areas)               */
    int32_t check_no_action;     /* NoAction.update no_action.py:32-53: IoU(pose, last pose) >
                                    no_action_iou on more than no_action_max_step checks    */
    int32_t no_action_max_step;  /* 100  envs/parking.py:344                                   */
    int32_t shaped_reward;       /* IoU gain + distance-to-target gain of _get_reward :163-188 */
    float arrival_threshold;     /* 0.95  arrival.py:22                                        */
    float no_action_iou;         /* 0.999 no_action.py:47                                      */
    float dist_reward_scale;     /* 0.1   envs/parking.py:187                                  */
} t2d_status_config;

typedef struct t2d_pool t2d_pool;

/* Error text of the most recent failing call on this pool (owned by the pool), or of the
 * most recent failing t2d_create when pool == NULL (thread-local).                      */
const char* t2d_last_error(const t2d_pool* pool);
int t2d_abi_version(void);

/* Create a pool of n_env environments x max_agents participants on HIP device device_id.
 * All state starts zeroed and inactive.                                                  */
int t2d_create(int32_t n_env, int32_t max_agents, int32_t device_id, t2d_pool** out_pool);
int t2d_destroy(t2d_pool* pool);

/* rows: n_types rows of row_stride doubles (row_stride >= T2D_PARAM_COLS), host memory.  */
int t2d_set_param_table(t2d_pool* pool, const double* rows, int32_t n_types, int32_t row_stride);

/* Static obstacle polygons + map boundary, CSR, host memory, env-local fp32 coordinates.
 *   env_poly_offsets  [n_env + 1]  polygons of env e are [off[e], off[e+1])
 *   poly_vert_offsets [n_poly + 1] vertices of polygon p are [off[p], off[p+1])
 *   verts_xy          [2 * n_vert] interleaved x,y
 *   boundary          [4 * n_env]  xmin, xmax, ymin, ymax (OutBound tuple order) or NULL
 *   boundary_valid    [n_env]      0 = "boundary is None" -> never out of bound; NULL = all valid
 * Polygons must be convex with 3..T2D_MAX_POLY_VERTS vertices; either winding accepted.  A polygon of
 * 5..8 vertices is evaluated as its fan of quads (v0 v1 v2 v3), (v0 v3 v4 v5), (v0 v5 v6 v7): the union is the
 * polygon, `intersects` is the OR over the parts; the lidar scans the undivided ring.
 * Size: the step kernels keep the static + lane parts of one workgroup's envs in ONE packed LDS record of at most 32 KiB
 * (t2d_geometry_budget says what a scene needs).  A scene beyond that -- a reference map of hundreds of lanelets,
 * map/element/map.py:242-329 answers its proximity queries with an STRtree -- is NOT refused: t2d_set_static_geometry /
 * t2d_set_lane_geometry keep its parts in global memory behind one uniform grid per env (the HBM grid tier,
 * tactics2d_amd/csrc/t2d_mapgrid.hip), and t2d_step / t2d_step_n run as t2d_integrate -> a map-events launch -> the event + status
 * launch (t2d_step_form: T2D_FORM_UNFUSED): the same flags, statuses and rewards at any map size, about three launches per step
 * instead of one.  (Not for generated parking scenes: their capacity layout is fixed.)              */
int t2d_set_static_geometry(t2d_pool* pool, const int32_t* env_poly_offsets,
                            const int32_t* poly_vert_offsets, const float* verts_xy,
                            const float* boundary, const uint8_t* boundary_valid);

/* Lane polygons for the build-defined off-lane flag; same CSR convention.  Envs with no
 * lane polygons never raise T2D_FLAG_OFF_LANE (== the reference stub).  The flag is
 * `not union(lanes).contains(pose)`: the boundary of the union of each env's lanes is extracted
 * here, once (lanes that abut must share their vertices exactly, or overlap: a sliver between two
 * almost-collinear edges is a real gap of the union).                                      */
int t2d_set_lane_geometry(t2d_pool* pool, const int32_t* env_lane_offsets,
                          const int32_t* lane_vert_offsets, const float* verts_xy);

int t2d_set_status_config(t2d_pool* pool, const t2d_status_config* cfg);

/* Target parking areas for Arrival and the reward shaping (Arrival.reset arrival.py:49-51,
 * _ParkingScenarioManager.target_area envs/parking.py:399-401): one quadrilateral per env,
 * target_xy[n_env][4][2] fp32 (either winding, convex); centroid[n_env][2] = its area centroid
 * (`target_area.geometry.centroid`) or NULL to have it computed.  NULL target_xy removes them.
 * Host memory.  (Re)initialises the per-env min-distance-to-target from the current state.    */
int t2d_set_target_areas(t2d_pool* pool, const float* target_xy, const float* centroid);

/* (Re)initialise participants.  All arrays are host memory with N = n_env*max_agents
 * elements; env_mask (n_env bytes, NULL = every env) selects which envs are written.
 * vx / vy may be NULL (then vx = speed*cos(heading), vy = speed*sin(heading) in fp64,
 * rounded to fp32 -- State.velocity, participant/trajectory/state.py:152-169).
 * Resets cnt_step, frame, status, reward and flags of the selected envs.                */
int t2d_reset(t2d_pool* pool, const uint8_t* env_mask, const float* x, const float* y,
              const float* heading, const float* speed, const float* vx, const float* vy,
              const uint8_t* type_id, const uint8_t* active);

/* Zero-copy actions: make the integrator read ACT0/ACT1 from caller-owned DEVICE memory (e.g. a
 * policy's output tensor, N floats each) instead of the pool's own buffers.  NULL, NULL rebinds
 * the pool's buffers.  The caller keeps the memory alive and orders its writes on the stream;
 * the library only ever READS it (IDM-controlled participants take their action from the pool's
 * own ACT0/ACT1 fields, where t2d_idm_actions writes).  t2d_upload of ACT0 / ACT1 ends a binding:
 * uploaded actions are the actions from then on.                                                */
int t2d_bind_actions(t2d_pool* pool, const float* act0_dev, const float* act1_dev);
/* The same with a stride: participant i's actions are act0_dev[i * stride] and act1_dev[i * stride] (stride in
 * elements, >= 1).  A policy's [N, 2] output in the reference's (steering, accel) layout (envs/parking.py:130-139)
 * binds as act0 = out + 1, act1 = out, stride = 2 -- no copy kernels between the policy and the step.              */
int t2d_bind_actions_strided(t2d_pool* pool, const float* act0_dev, const float* act1_dev, int32_t stride);
/* Optional guard for bound memory: n_elements = how many float elements may be read from act0_dev and from act1_dev (0 = not
 * declared, the default after every t2d_bind_actions[_strided]; the library cannot see the size of caller-owned memory).  With a
 * declared extent, t2d_step_n refuses (T2D_ERR_INVALID) a fragment whose last step would read beyond it -- participant N - 1 of
 * step n_steps - 1 reads element (N - 1) * stride + (n_steps - 1) * act_step_stride -- instead of faulting on the device.
 * T2D_ERR_STATE while the pool reads its own ACT0 / ACT1 fields; T2D_ERR_INVALID if one action set does not fit.
 * (The reference has no counterpart: its actions are Python tuples, envs/parking.py:239.)                                      */
int t2d_set_action_extent(t2d_pool* pool, int64_t n_elements);

/* Physics only: one PhysicsModelBase.step(interval_ms) for every active participant,
 * actions taken from fields ACT0/ACT1.                                                  */
int t2d_integrate(t2d_pool* pool, int32_t interval_ms, void* hip_stream);
#define T2D_MAX_INTERVAL_MS 32767   /* interval_ms of every stepping call: 1 .. 32767 (the sub-step count travels in 15 bits) */
/* Events only: recompute FLAGS / ENV_FLAGS from the current poses.                      */
int t2d_collide(t2d_pool* pool, void* hip_stream);
/* ScenarioManager.check_status alone (envs/parking.py:361-392): events + the status / reward /
 * counter epilogue on the current poses (what t2d_step runs after the integrator).         */
int t2d_check_status(t2d_pool* pool, int32_t interval_ms, void* hip_stream);
/* ScenarioManager.update + check_status: integrate, collide, status/reward epilogue.  By default
 * ONE fused launch (the participant is integrated in registers and its new pose feeds the event
 * phases directly); t2d_set_fused_step(pool, 0) selects the two-kernel form, whose results are
 * bit-identical.  kernel_id 2 of t2d_profile_read times the fused launch.                   */
int t2d_step(t2d_pool* pool, int32_t interval_ms, void* hip_stream);
int t2d_set_fused_step(t2d_pool* pool, int32_t on);
/* Pools with ONE participant per env (ParkingEnv, BASELINE config 2) whose parameter table holds box-shaped types only
 * and that have no lane geometry take their fused step with one WAVE per env instead of one lane per participant (the
 * quads of the lot, the constraints of the two IoUs spread over the wave's lanes: ~2.5x faster at 4096 envs).  Same
 * arithmetic, same results bit for bit -- with ONE exception: the single-ego kernel always ITERATES a kinematic step, so on
 * pools large enough for the general kernel to resum it (integrator variant 1 at >= 131072 participants on an MI355X, or
 * variant 3: t2d_set_integrator_variant) the two agree to the series' truncation (< 1e-9 m; the last bit of the fp32 state may
 * differ), and bit for bit again under variants 0 and 2.  On by default, t2d_set_ego_kernel(pool, 0) keeps the pool on the
 * general kernel. */
int t2d_set_ego_kernel(t2d_pool* pool, int32_t on);
/* One step of n pools in a single call -- env groups on separate streams (independent environments cut into
 * groups whose launches overlap: one group's start-up latency and tail hide behind the others' busy middle,
 * DESIGN.md "Env groups").  For i in [0, n): if act0 / act1 are non-NULL, t2d_bind_actions(pools[i],
 * act0[i], act1[i]); then t2d_step(pools[i], interval_ms, hip_streams[i]).  Exists because at ~6 us of GPU
 * time per group and step, one host call per pool and per step is what limits the rate.  With n > 1 the step
 * kernels are also told that launches overlap, which turns their wave priorities round (a launch that has the GPU
 * to itself serves the waves that are behind first, overlapping launches the waves that are about to retire:
 * DESIGN.md 4.2); pools stepped on different streams through separate t2d_step calls keep the single-launch rule.
 * Returns the first error (its message is on that pool).                                                 */
int t2d_step_groups(t2d_pool* const* pools, const float* const* act0_dev, const float* const* act1_dev,
                    void* const* hip_streams, int32_t n, int32_t interval_ms);

/* n_steps consecutive t2d_step's enqueued by ONE call -- a rollout fragment on resident actions (the loop of
 * envs/parking.py:219-256 without a host round trip per step).  Step k reads participant i's action at
 * act[i * stride + k * act_step_stride] of the bound (or the pool's own) action arrays: act_step_stride = 0 repeats one
 * action set (frame skip), n_env * max_agents * stride walks an action ring laid out [n_steps][N].  Results are exactly
 * those of n_steps t2d_step calls; per-step rewards / statuses are in the record ring (T2D_F_RECORD), the participant
 * fields hold the last step's values.
 * How: pools whose step is the fused kernel alone (no drift / regenerated scenes; installed IDM controllers are run by the
 * step launch itself, here as in t2d_step, for envs of 2..64 participants) get ONE launch
 * holding all the steps.  Large pools: workgroup (g, k) takes step k of the envs of workgroup g and is ordered after workgroup
 * (g, k - 1) by a per-workgroup word in device memory (kept inside one XCD's L2, the placement checked: DESIGN.md 4.10) -- no
 * launch boundary between steps, so the start-up of step k + 1 overlaps the tail of step k.  Small pools: every workgroup
 * loops over the steps itself, and where a step's workgroups number at most the device's CUs a second set of waves per
 * workgroup integrates step k + 1 while the first checks the events of step k (t2d_step_form tells which).
 * Failure (never observed; forced by tests with t2d_debug_chain_fault): every wait is bounded, and a chained hand-off checks
 * that producer and consumer share an XCD.  A workgroup whose hand-off fails records the failure and goes on (never a hang):
 * what it and everything enqueued behind it on the device computes from then on -- state, flags, records, a t2d_gather of those
 * records, a lidar scan -- is INVALID until the host has noticed: the first t2d_sync / t2d_download / t2d_step_n after it
 * returns T2D_ERR_STATE -- once -- and the pool goes on with ordinary launches (t2d_set_step_chaining re-enables chaining).
 * CHAIN forms: the pool has then been rolled back to the state and step count (t2d_step_count) it had when the failed fragment
 * began -- every chained fragment checkpoints what it starts from -- so the caller re-issues its steps from there; the pure
 * outputs (flags, status, reward, records, vx / vy of the single-track models, the applied action) are rewritten by the next step.  LOOP forms (a wait inside a workgroup ran out): the state is
 * undefined, t2d_reset / t2d_restore(mode 0) / uploads make the pool usable again.
 * act_step_stride > 0 needs an action ring bound with t2d_bind_actions[_strided] (n_steps * act_step_stride elements beyond
 * the last participant's first action); the pool's own ACT0 / ACT1 fields hold one set: T2D_ERR_INVALID otherwise.
 * Other pools, and every pool after
 * t2d_set_step_chaining(pool, 0, *), take n_steps ordinary launches -- and t2d_step its plainest form: installed IDM
 * controllers as a launch of their own (on = 1: automatic, the default; 2 / 3 / 4 pin the chained
 * form / the plain loop / the loop with integrator but without lane waves -- measurements and tests).  priority_rule: wave priorities inside a chained launch
 * (1, default: the rule for overlapping work; 0: the single-launch rule).  kernel_id 7 in t2d_profile_read
 * (one "launch" = one chained launch of up to T2D_RECORD_RING steps).                                                    */
int t2d_step_n(t2d_pool* pool, int32_t interval_ms, int32_t n_steps, int64_t act_step_stride, void* hip_stream);
int t2d_set_step_chaining(t2d_pool* pool, int32_t on, int32_t priority_rule);
/* Small pools of envs with 33..64 participants (at most 4 x the device's CUs envs, no IoU events): the fused step gives every
 * env a workgroup of its own and runs its event stages on four waves side by side (pairs / static polygons / two halves of the
 * lane polygons) instead of one wave walking through all of them -- same arithmetic, same results, a shorter dependent
 * chain per env.  Only pools whose envs carry static obstacles AND lanes take it (a pool with nothing to run side by side is
 * slower that way).  On by default (t2d_step and t2d_step_n alike); 0 keeps one wave per env.                              */
int t2d_set_split_step(t2d_pool* pool, int32_t on);
/* Which form of the step kernel a call with n_steps steps takes on this pool now (diagnostics, tests, bench lines):        */
enum {
    T2D_FORM_UNFUSED = 0,     /* separate integrate + check_status launches, or helper kernels around the fused step */
    T2D_FORM_STEP = 1,        /* one fused launch per step, one wave per env (or per 2..64 small envs)                 */
    T2D_FORM_STEP_SPLIT = 2,  /* one fused launch per step, one workgroup per env                                      */
    T2D_FORM_EGO = 3,         /* single-ego kernel, one launch per step                                                */
    T2D_FORM_EGO_LOOP = 4,    /* single-ego kernel, the lanes loop over the steps                                      */
    T2D_FORM_CHAIN = 5,       /* one launch of (workgroup, step) workgroups ordered by counters                        */
    T2D_FORM_CHAIN_SPLIT = 6, /* the same with one workgroup per env                                                   */
    T2D_FORM_LOOP = 7,        /* resident workgroups loop over the steps                                               */
    T2D_FORM_LOOP_PIPE = 8,   /* ... with integrator waves running one step ahead of the event waves                   */
    T2D_FORM_EGO_LOOP_PIPE = 9 /* the single-ego kernel's loop with integrator waves                                  */
};
int t2d_step_form(t2d_pool* pool, int32_t n_steps);   /* a T2D_FORM_* value; -1: null pool */

/* Zero-copy device pointer of a field (for wrapping as a torch tensor).                 */
int t2d_get_field(t2d_pool* pool, int32_t field_id, void** dev_ptr, size_t* nbytes);
/* Synchronous host<->device copies of a whole field (parity tests, small envs).         */
int t2d_download(t2d_pool* pool, int32_t field_id, void* host_dst, size_t nbytes);
int t2d_upload(t2d_pool* pool, int32_t field_id, const void* host_src, size_t nbytes);
/* Waits for THIS pool's work only: the streams it was launched on since the last such call (up to four are tracked; more
 * distinct streams, or a tracked stream that has been destroyed in the meantime, make the call fall back to a device-wide
 * synchronise) plus the pool's own internal streams.  The set-up calls and the two copies above wait the same way.       */
int t2d_sync(t2d_pool* pool);

/* ---- the Gym-API host path: host actions in, ONE packed host frame out ------------------------------------------------
 * What the reference's caller gets from ParkingEnv.step (envs/parking.py:219-256) is host values: the observation, the reward,
 * terminated / truncated and the info dict of _get_infos (:203-217: lidar, state, target area / heading, the two statuses and
 * the relative pose of _get_relative_pose :190-201).  t2d_step_host is that call for every env of the pool: it takes the
 * actions from HOST memory, runs t2d_step (and, when the frame carries a lidar section, t2d_lidar_scan writing straight into
 * the frame), packs everything the 5-tuple needs of the ego of every env into one contiguous FRAME and brings it to pinned host
 * memory with one asynchronous copy and one stream synchronisation -- instead of one blocking copy per field.
 *   t2d_frame_config   chooses the sections (T2D_FRAME_*), allocates the device frame and n_host_frames (1..T2D_MAX_HOST_FRAMES)
 *                      pinned host frames, and fills
 *                      *layout with the byte offsets of the sections inside a frame (-1 = section absent).  Asking for the
 *                      configuration already in place keeps the frames (and what a caller still holds of them); a DIFFERENT one
 *                      frees them: views of earlier frames are invalid from then on, as after t2d_destroy, and a pool whose
 *                      actions are the ones t2d_step_host last staged reads its OWN action fields (T2D_F_ACT0 / ACT1) again
 *                      until the next t2d_step_host / t2d_bind_actions / upload.  Call it again after
 *                      t2d_lidar_config changed the beam count.  T2D_FRAME_ZEROCOPY: no copy commands at all -- the step kernel
 *                      reads the actions from mapped host memory and the pack / lidar kernels write the mapped host frame
 *                      (lowest latency for small pools; a large pool's 8 B per participant would cross PCIe inside the step).
 *   t2d_set_target_headings  target_heading of every env (envs/parking.py:401), host fp64 [n_env]; generated scenes
 *                      (t2d_parking_scenes) bring their own.  Without it diff_heading is NaN.
 *   t2d_step_host      actions_host = f32 [n_env * max_agents][2] in the reference's action layout (steering, accel)
 *                      (envs/parking.py:239; a point mass: (ay, ax)), or NULL = the actions already in / bound to the pool.
 *                      Ends any t2d_bind_actions binding (the pool reads a buffer of its own from then on).  Returns with
 *                      *frame_host pointing at the filled host frame: frame_index (0 .. n_host_frames - 1) names the pinned
 *                      frame to fill -- a caller that hands the frame's memory on (numpy views) picks one nobody holds any
 *                      more, and no copy is needed -- or -1 = the frames in turn.  Reports a failed scene regeneration like
 *                      t2d_sync does.
 *                      action_box = {steering lo, steering hi, accel lo, accel hi} or NULL: `action_space.contains(action)`
 *                      of envs/parking.py:235-236 for every row, checked while the actions are staged (closed bounds, a NaN
 *                      is outside) -- T2D_ERR_ACTION, nothing stepped AND nothing staged (the verdict precedes the copy: the
 *                      actions of the last accepted call stay in place), the message names the first offending row.
 *   t2d_host_action_buffer  the pool's own pinned staging buffer, f32 [n_env * max_agents][2] (valid until the next
 *                      t2d_frame_config): a caller that writes its actions THERE and passes that pointer to t2d_step_host saves the
 *                      staging copy (2 MB per step at 4096 x 64: ~ 70 us of host memcpy); the box check still reads every row.
 *   t2d_frame_fetch    the frame of the CURRENT state without stepping (what reset() returns; the lidar section is scanned
 *                      from the current poses).
 * Frame sections (E = n_env; every offset a multiple of 256 B):
 *   header     u32 [16]: {step count (low half), scene-regeneration error word, 0 ...}
 *   obs        f32 [E][6]   x, y, heading, speed, vx, vy of the ego (vx / vy as stored: see T2D_F_VX)
 *   rel        f64 [E][3]   diff_position, diff_angle, diff_heading of _get_relative_pose: |centroid - (x, y)|,
 *                           atan2(cy - y, cx - x) - heading, target_heading - heading, in fp64 from the stored fp32 state
 *                           (NaN without target areas / headings)
 *   reward     f32 [E]      status  u8 [E][4]   iou f32 [E]   frame_ms i32 [E]   cnt_step i32 [E]
 *   episode    i32 [E]      generated scenes: episode number of the env (0 otherwise)
 *   target     f32 [E][8] + f64 [E] target_heading   (T2D_FRAME_TARGET: the target area the env is in NOW; generated scenes)
 *   lidar      f32 [E][n_beams]                      (T2D_FRAME_LIDAR)                                                      */
#define T2D_FRAME_LIDAR    1u
#define T2D_FRAME_TARGET   2u
#define T2D_FRAME_ZEROCOPY 4u
#define T2D_MAX_HOST_FRAMES 16
typedef struct t2d_frame_layout {
    int64_t bytes;          /* size of one frame */
    int64_t off_obs, off_rel, off_reward, off_status, off_iou, off_frame_ms, off_cnt_step, off_episode;
    int64_t off_target, off_target_heading, off_lidar;   /* -1 = absent */
    int32_t n_env, n_beams;
} t2d_frame_layout;
int t2d_frame_config(t2d_pool* pool, uint32_t sections, int32_t n_host_frames, t2d_frame_layout* layout);
int t2d_set_target_headings(t2d_pool* pool, const double* heading_host);
int t2d_step_host(t2d_pool* pool, const float* actions_host, const float* action_box, int32_t interval_ms, void* hip_stream,
                  int32_t frame_index, const void** frame_host);
int t2d_frame_fetch(t2d_pool* pool, void* hip_stream, int32_t frame_index, const void** frame_host);
int t2d_host_action_buffer(t2d_pool* pool, float** actions_host);

/* Episode-start snapshot for device-side (auto-)reset -- the vector-env counterpart of
 * ParkingEnv.reset (envs/parking.py:262-298) without a host round trip.
 * t2d_snapshot copies the current participant state (x, y, heading, speed, vx, vy, ids) into a
 * pool-owned device snapshot.  t2d_restore (asynchronous on the stream) writes it back and
 * clears cnt_step / frame / status / reward / flags:
 *   mode 0: every env;  mode 1: only envs whose status says terminated or truncated.       */
int t2d_snapshot(t2d_pool* pool);
int t2d_restore(t2d_pool* pool, int32_t mode, void* hip_stream);
/* Vector-env auto-reset fused into t2d_step: an env whose status comes out terminated or truncated
 * is put back to the snapshot at the end of the same launch (state, ids, cnt_step, frame).  Its
 * status / reward / flags keep the terminal step's values until the next step overwrites them, as
 * Gym vector envs report them.  Needs a snapshot.                                                */
int t2d_set_auto_reset(t2d_pool* pool, int32_t on);

/* Single-line lidar of the ego of every env (SingleLineLidar, sensor/lidar.py:33-221, as configured by
 * ParkingEnv envs/parking.py:303-304,422-431: 360 beams, 20 m): distance to the nearest obstacle edge
 * along n_beams rays at angles linspace(0, 2 pi, n_beams, endpoint=False) in the ego frame; +inf = no
 * return.  Obstacles = the static polygons of t2d_set_static_geometry and, when include_participants,
 * the boxes of the other active participants.  beam_sin / beam_cos: host arrays [n_beams] with the sin /
 * cos of those angles (numpy), or NULL to have the library compute them with libm.
 * t2d_lidar_scan writes fp32 [n_env][n_beams] to out_dev (caller-owned device memory, e.g. the policy's
 * observation tensor) or, when NULL, to the pool's T2D_F_LIDAR buffer.  kernel_id 3 in t2d_profile_read. */
int t2d_lidar_config(t2d_pool* pool, int32_t n_beams, float max_range, int32_t include_participants,
                     const double* beam_sin, const double* beam_cos);
int t2d_lidar_scan(t2d_pool* pool, float* out_dev, void* hip_stream);

/* On-device scripted agents: IDM car following (IDMController, controller/idm_controller.py:33-157).
 * ctrl_rows: host array [n_ctrl][row_stride >= T2D_IDM_COLS] of fp64 parameter sets -- the constructor
 * arguments of idm_controller.py:33-57 (`configure` :143-157 = calling t2d_set_idm again) plus the two
 * columns of the build-defined leader rule; ctrl_id: host array [n_env * max_agents], index of the
 * participant's parameter set or T2D_IDM_NONE (its action stays whatever the caller supplied).
 * n_ctrl = 0 uninstalls.  t2d_idm_actions = IDMController.step :59-93 for every controlled participant:
 * acceleration (np.clip-ed to [-comfortable_deceleration, max_acceleration]) -> the pool's T2D_F_ACT0, steering
 * 0.0 -> T2D_F_ACT1 (always the pool's own fields: memory bound with t2d_bind_actions is never written; controlled
 * participants are integrated from the pool's fields, the others from the bound memory), leader index ->
 * T2D_F_LEADER.  The reference takes `leading_state` from its caller; here the leader is the nearest active
 * participant ahead (0 < longitudinal offset <= horizon along the own heading) inside the own corridor
 * (|lateral offset| <= lane_half_width), lowest index on ties; none -> free-flow branch.  Distance is
 * centre to centre (np.hypot), as in :111-113.  While installed, t2d_step, t2d_step_n and t2d_integrate run it
 * first: t2d_integrate as a launch of its own on the same stream (kernel_id 4 in t2d_profile_read); the step launches
 * -- for envs of 2..64 participants without IoU events -- in their own front, same leaders and accelerations
 * (t2d_set_step_chaining(pool, 0, *) keeps the separate launch there as well).                */
enum t2d_idm_col {
    T2D_IDM_DESIRED_SPEED = 0,
    T2D_IDM_TIME_HEADWAY = 1,
    T2D_IDM_MIN_SPACING = 2,
    T2D_IDM_MAX_ACCEL = 3,
    T2D_IDM_COMF_DECEL = 4,
    T2D_IDM_DELTA = 5,
    T2D_IDM_LANE_HALF_WIDTH = 6, /* build-defined leader rule */
    T2D_IDM_HORIZON = 7,         /* build-defined: look-ahead (m), may be +inf */
    T2D_IDM_COLS = 8
};
#define T2D_IDM_NONE 255
/* values of forced_leader_dev[i] (device array [n_env * max_agents], or NULL = search everywhere): the
 * reference's calling convention `step(ego_state, leading_state)` -- an agent index inside the env (an
 * inactive or out-of-range index counts as no leader), T2D_IDM_LEADER_FREE = `leading_state=None`,
 * T2D_IDM_LEADER_SEARCH = apply the leader rule above.                                               */
#define T2D_IDM_LEADER_FREE (-1)
#define T2D_IDM_LEADER_SEARCH (-2)
int t2d_set_idm(t2d_pool* pool, const double* ctrl_rows, int32_t n_ctrl, int32_t row_stride,
                const uint8_t* ctrl_id);
int t2d_idm_actions(t2d_pool* pool, const int32_t* forced_leader_dev, void* hip_stream);

/* verify_state -- the reference's "very rough check" of a state transition (SingleTrackKinematics.verify_state
 * physics/single_track_kinematics.py:200-250, SingleTrackDynamics.verify_state single_track_dynamics.py:253-306,
 * PointMass.verify_state point_mass.py:234-259; called by ParticipantBase._verify_state participant_base.py:107-118).
 * last_state = the pool's current state, candidate = four device arrays [n_env * max_agents] (x, y, heading, speed);
 * valid_dev[i] = 1 where the reference returns True (also for inactive participants and interval_ms = 0).
 * Quirks kept: any unbounded range -> True; x / y are tested with strict inequalities against an unsorted range. */
int t2d_verify_state(t2d_pool* pool, const float* x_dev, const float* y_dev, const float* heading_dev,
                     const float* speed_dev, int32_t interval_ms, uint8_t* valid_dev, void* hip_stream);

/* Reset-time scene synthesis: ParkingLotGenerator.generate
 * (map/generator/generate_parking_lot.py:239-444) for n_env independent scenes, one lane per scene, on `device_id`.
 * PARITY UNPINNED against the reference: it draws from numpy's global MT19937 stream and evaluates its predicates in
 * shapely (neither available to this build), so the kernel follows the reference's distributions, draw order, control
 * flow and predicates (closed `intersects`, `distance`, `contains`) on a counter-based stream of its own
 * (oracle/t2d_oracle.c: t2do_generate_parking states it; the two agree bit for bit).  Scene e uses stream
 * (seed, first_env + e): a sharded job generates the same scenes whatever the number of ranks.
 * Outputs are HOST arrays (the call synchronises): quads [n_env][T2D_GEN_MAX_QUADS][4][2] fp32 obstacle quads in
 * Map.areas order (same id replaces: map.py:444-453), quad_id their reference ids (-1 = unused slot), n_quads,
 * start [n_env][3] = x, y, heading (fp64, heading not wrapped: +pi when flipped, :409-419), target [n_env][4][2],
 * target_heading (fp64), boundary [n_env][4] = xmin, xmax, ymin, ymax (:436-440), info = T2D_GEN_* bits |
 * obstacle attempts << 8 | start attempts << 16.  The reference's rejection loops are unbounded; here an env whose
 * loop hits the cap is returned with T2D_GEN_UNVERIFIED / T2D_GEN_START_UNVERIFIED set, never silently.        */
#define T2D_GEN_MAX_QUADS 12
#define T2D_GEN_MAX_ATTEMPTS 8
#define T2D_GEN_MAX_START_ATTEMPTS 64
#define T2D_GEN_BAY 0x01u              /* mode == "bay" (else "parallel") */
#define T2D_GEN_UNVERIFIED 0x02u       /* _verify_obstacles still False after T2D_GEN_MAX_ATTEMPTS */
#define T2D_GEN_START_UNVERIFIED 0x04u /* _verify_start_state still False after T2D_GEN_MAX_START_ATTEMPTS */
#define T2D_GEN_NONCONVEX 0x08u        /* an obstacle quad is not convex (the event kernels need convex polygons) */
#define T2D_GEN_OVERFLOW 0x10u         /* more than T2D_GEN_MAX_QUADS distinct ids / obstacle list full */
#define T2D_GEN_START_FLIPPED 0x20u
#define T2D_GEN_TARGET_FLIPPED 0x40u
int t2d_generate_parking(int32_t device_id, uint64_t seed, int64_t first_env, int32_t n_env, double type_proportion,
                         double vehicle_length, double vehicle_width, float* quads, int32_t* quad_id,
                         int32_t* n_quads, double* start, float* target, double* target_heading, float* boundary,
                         uint32_t* info);

/* The same generator writing straight into a pool (one participant per env, SingleTrackKinematics/Dynamics/PointMass
 * agent of type 0): every env's obstacles, boundary, target area (+ area centroid), start pose, episode snapshot and
 * IoU / shaping state are produced and installed by one launch -- what t2d_set_static_geometry + t2d_set_target_areas
 * + t2d_reset + t2d_snapshot do for host-described scenes, with nothing crossing PCIe (envs/parking.py:397-441).
 * Call after t2d_set_param_table / t2d_set_status_config.  Scene of (env e, episode k) = stream
 * first_env + e + k * env_stride (env_stride = total envs of the job).  regenerate != 0: every later t2d_step /
 * t2d_check_status is followed, on the same stream, by a launch that gives each env whose episode just ended
 * (terminated | truncated) the scene of its next episode -- the reference's reset() per episode -- while the terminal
 * status / reward stay readable until the next step; kernel_id 6 in t2d_profile_read.  regenerate = 1: the scenes of
 * the next 16 episodes of every env are kept staged in HBM and topped up every 8 steps on a stream owned by the pool
 * (a scene is a ~46 us single-lane chain), so the step's stream only copies: sixteen lanes per finished env, in the
 * epilogue of the ego step kernel itself when that is the pool's step (no launch behind it, no kernel_id 6 then),
 * else in a launch of their own.  The shortest episode the status rules allow is two steps, the ring holds sixteen and
 * the step stream waits for the refill before last: an env cannot outrun its ring.  Should one ever find its slot
 * unstaged it keeps its lot, and the next t2d_sync / t2d_download / t2d_step_n returns T2D_ERR_STATE (once; call
 * t2d_parking_scenes again) -- results never depend on timing.  regenerate = 2: no staging,
 * scenes are generated on the step's stream.  Geometry lives in a
 * fixed-capacity layout (T2D_GEN_MAX_QUADS polygon slots per env); t2d_set_static_geometry leaves this mode.
 * t2d_get_parking_scenes copies the current scenes (arrays as in t2d_generate_parking, any pointer may be NULL) and
 * the per-env episode numbers to the host.                                                                         */
int t2d_parking_scenes(t2d_pool* pool, uint64_t seed, int64_t first_env, int64_t env_stride, double type_proportion,
                       double vehicle_length, double vehicle_width, int32_t regenerate);
int t2d_get_parking_scenes(t2d_pool* pool, float* quads, int32_t* quad_id, int32_t* n_quads, double* start,
                           float* target, double* target_heading, float* boundary, uint32_t* info, int32_t* episode);

/* Kernel variants: 0 = exact (library-grade fp64 trig every sub-step), 1 = fast (default: SingleTrackKinematics steps whose
 * speed stays inside its bounds are RESUMMED -- the Euler sum of the step's sub-steps evaluated as a series instead of
 * iterated, truncation < 1e-9 m --, everything else advances cos / sin by a rotation recurrence), 2 = fast with the kinematic
 * steps iterated as well (rounds 1-4's form; A/B measurements and tests).  Pools of fewer than two waves per SIMD of the device
 * (< 131072 participants on an MI355X) iterate under variant 1 too -- a lone wave is a latency chain the series' table fetch
 * lengthens --; 3 = variant 1 with the series whatever the pool size (tests).  The single-ego kernel (t2d_set_ego_kernel) and
 * the looping forms of t2d_step_n (pools of <= 2 workgroups per CU) iterate under every variant.  All satisfy the 1e-5
 * contract; see DESIGN.md. */
int t2d_set_integrator_variant(t2d_pool* pool, int32_t variant);

/* Which pure OUTPUT columns the integrators store.  The reference's step() returns a State carrying vx / vy and the
 * applied (clipped) action beside the pose (single_track_kinematics.py:169-198); on the step path nothing reads them back:
 * vx / vy of the single-track models are v cos / v sin of the new heading and T2D_F_APPLIED0/1 the clipped action.  A
 * rollout that does not consume them leaves 16 B per participant and step unwritten (a quarter of the step's HBM writes).
 * Default T2D_OUT_ALL (the reference's State).  A point mass's vx / vy are state and are stored whatever the mask says;
 * with a bit cleared the column keeps its last stored values.                                                          */
#define T2D_OUT_VELOCITY 1u   /* T2D_F_VX / T2D_F_VY of SingleTrackKinematics participants */
#define T2D_OUT_APPLIED 2u    /* T2D_F_APPLIED0 / T2D_F_APPLIED1 */
#define T2D_OUT_ALL 3u
int t2d_set_outputs(t2d_pool* pool, uint32_t mask);

/* Per-kernel timing with HIP events recorded on the launch stream around each kernel.
 * kernel_id: 0 = integrate, 1 = collide(+status), 2 = fused step, 3 = lidar, 4 = idm, 5 = drift, 6 = scene regeneration.                                   */
int t2d_profile_enable(t2d_pool* pool, int32_t on);
int t2d_profile_read(t2d_pool* pool, int32_t kernel_id, double* total_ms, int64_t* launches);

/* ---- multi-GPU: environments shard across ranks, one process per GPU, no data-path collective.  The only
 * exchange is an all-gather of the 8-byte per-env result records {reward bits, status word} -- RCCL over xGMI, issued
 * from here, reading the record ring (T2D_F_RECORD) in place.
 *   t2d_comm_unique_id  ncclGetUniqueId: rank 0 calls it and ships the 128 bytes to the other ranks by whatever the
 *                       launcher offers (a torch.distributed store, MPI, a file): that bootstrap is the host's business.
 *   t2d_comm_init       ncclCommInitRank on the pool's device (collective over the ranks).  world = 1 with id = NULL
 *                       needs no RCCL at all.  RCCL is opened with dlopen here, not linked.
 *   t2d_gather          all-gather of the records of the LAST n_steps steps (n_steps divides T2D_RECORD_RING and the
 *                       number of steps taken so far) into out_dev = u32 [world][n_steps][n_env][2], caller-owned device
 *                       memory.  nccl_comm = an ncclComm_t of the caller's, or NULL = the pool's own.  Ordered after
 *                       what is enqueued on hip_stream (the steps' stream) and run on a stream of the pool's own, so the
 *                       following steps do not wait for it; a later step that is about to overwrite a slot a gather
 *                       still reads waits for that gather first (event wait on the step's stream, inside t2d_step).
 *   t2d_gather_wait     makes hip_stream wait for every gather issued so far (block_host = 0), or blocks the host until
 *                       they are done (block_host != 0); out_dev may be read after that.                              */
#define T2D_COMM_ID_BYTES 128
int t2d_comm_unique_id(uint8_t* id_out);
int t2d_comm_init(t2d_pool* pool, const uint8_t* id, int32_t rank, int32_t world);
int t2d_gather(t2d_pool* pool, void* nccl_comm, int32_t n_steps, void* out_dev, void* hip_stream);
/* What the pool's communicator is: native_rccl = 1 when t2d_comm_init created an RCCL communicator (0: none, or the
 * RCCL-free world of one), world / rank read back FROM that communicator (ncclCommCount / ncclCommUserRank) -- the proof
 * a multi-GPU run can print that RCCL saw N ranks.  Any pointer may be NULL.                                        */
int t2d_comm_info(t2d_pool* pool, int32_t* native_rccl, int32_t* world, int32_t* rank);
/* t2d_step / t2d_check_status calls so far (t2d_step_n counts n): the step count t2d_gather's n_steps must divide, and the
 * record-ring slot of the next step (count % T2D_RECORD_RING).  -1 for a null pool.                                   */
int64_t t2d_step_count(const t2d_pool* pool);
int t2d_gather_wait(t2d_pool* pool, void* hip_stream, int32_t block_host);

/* Capacity planning: resident workgroups per CU of the fused step kernel with this pool's geometry, its LDS bytes per
 * workgroup (static tables + the workgroup's geometry record), and (may be NULL) the bytes of packed geometry records
 * the workgroups of one step launch stage into LDS.  The 4096 x 64 metric launch is one wave-round of 1024 workgroups
 * on 256 CUs and needs 4; a scene whose record grows past the LDS budget halves the rate.                          */
int t2d_step_occupancy(t2d_pool* pool, int32_t* blocks_per_cu, int64_t* lds_bytes,
                       int64_t* geometry_bytes_per_launch);

/* A hipStreamNonBlocking stream created by the library (priority as hipStreamCreateWithPriority: 0 default, negative
 * higher) -- for hosts that step env groups on streams of their own (t2d_step_groups) without another HIP binding.      */
int t2d_stream_create(int32_t device_id, int32_t priority, void** out_stream);
int t2d_stream_destroy(void* stream);

/* Host-only (no device is touched): the rectangles t2d_set_lane_geometry finds inside the union of each env's lane polygons
 * -- the certificate behind the step kernel's off-lane short cut (a pose whose box lies in one of them is contained in the
 * union).  out = f32 [n_env][T2D_SAFE_RECTS][4] (xmin, xmax, ymin, ymax); unused slots hold (+inf, -inf, +inf, -inf).
 * Same CSR arguments as t2d_set_lane_geometry.  Lets a caller (and the CPU tests) hold the certificate against its own
 * predicate (map/element/map.py:242-329 answers the same "what is near this pose" question with an STRtree).            */
#define T2D_SAFE_RECTS 4
int t2d_lane_safe_rects(int32_t n_env, const int32_t* env_lane_offsets, const int32_t* lane_vert_offsets,
                        const float* verts_xy, float* out);

/* Host-only (no device is touched): the LDS budget of a scene before it is installed.  The step kernels keep the static and
 * lane geometry of one workgroup's envs in ONE packed record of at most 32 KiB; t2d_set_static_geometry / t2d_set_lane_geometry
 * narrow the workgroup down to one wave (64 / padded max_agents envs) before they move the scene to the HBM grid tier.  This call runs
 * the same preparation (convexity checks, fans of quads for 5..8-gons, the boundary pieces of each env's lane union) on host
 * CSR arrays (as in t2d_set_*_geometry; either pair may be NULL) and reports the dwords the fullest such workgroup needs and
 * the budget (8192): what tactics2d_amd/mapgeom.py uses to say how many lane / obstacle polygons of a reference map
 * (map/element/lane.py:125-130, map/element/area.py) an env can carry.                                                      */
int t2d_geometry_budget(int32_t n_env, int32_t max_agents, const int32_t* env_poly_offsets, const int32_t* poly_vert_offsets,
                        const float* poly_xy, const int32_t* env_lane_offsets, const int32_t* lane_vert_offsets,
                        const float* lane_xy, int32_t* dwords_needed, int32_t* dwords_budget, int32_t* envs_per_workgroup);

/* Test and measurement hooks (fault injection, a gather delay, a stand-in policy and closed-loop runner, placement maps) are
 * NOT part of this library: include/t2d_debug.h, exported only by libt2d_hip_debug.so (the same sources built with
 * -DT2D_DEBUG_HOOKS), which tests/, bench.py's closed_loop leg and scripts/ load.                                          */

#ifdef __cplusplus
}
#endif
#endif /* T2D_H_ */
