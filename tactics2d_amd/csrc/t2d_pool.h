// t2d_pool.h -- internal layout of a participant pool (host + device views).
//
// Data layout in HBM (all allocations hipMalloc'ed once in t2d_create, 256-B aligned):
//   per participant, N = n_env * max_agents, env-major (idx = env * A + agent), one
//   contiguous array per field (Struct-of-Arrays) so a wave64 touching 64 consecutive
//   participants issues one 256-B coalesced transaction per field:
//     x, y, heading, speed, vx, vy, act0, act1, applied0, applied1 : f32[N]
//     ids, flags                                                   : u32[N]
//   per env: env_flags u32[E], cnt_step i32[E], frame_ms i32[E], status u8[4E], reward f32[E]
//   parameter table: fp64, TRANSPOSED [T2D_PARAM_COLS][T2D_MAX_TYPES] so that lanes reading
//     column c for different type ids hit consecutive LDS banks after staging
//   static / lane geometry: CSR offsets + fp64 CCW vertices + per-polygon AABB (fp64)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/t2d.h"

namespace t2d {

// Packed geometry record of one collide workgroup (EPB consecutive envs), all dwords, kind k =
// 0 static obstacles / 1 lanes:
//   pstart[k][EPB+1]   first polygon of each env inside the record (int)
//   vstart[k][MP_k+1]  first vertex of each polygon inside the record (int)
//   aabb[k][MP_k]      xmin, xmax, ymin, ymax (fp32, exact: vertices are fp32)
//   xy[k][MV_k]        CCW vertices x, y (fp32)
//   bstart[MP_1+1]     lanes only: first boundary piece of each lane polygon inside the record (int)
//   bnd[MB][4]         lanes only: boundary pieces of the union of the env's lanes, fp64 Ax, Ay, Bx, By (16-B aligned),
//                      grouped by the lane polygon whose edge they are part of (off-lane = not union.contains(pose))
//   safe[EPB][kSafeRects]  lanes only: axis-aligned rectangles (xmin, xmax, ymin, ymax, fp32) known to lie inside the
//                      union of the env's lanes (a rectangular lane, or several that abut exactly, shrunk by 0.1 mm): a
//                      pose whose box lies in one of them is contained in the union -- certified without the polygon tests.
//                      Unused slots hold the empty box (+inf, -inf, +inf, -inf)
// MP_k / MV_k / MB = largest polygon / vertex / piece count of any workgroup; stride is a multiple of 4
// dwords so records are copied to LDS with 16-B loads.
struct GeoLayout {
    int32_t stride;          // dwords per record (0 when there is no geometry)
    int32_t epb;             // environments per collide workgroup
    int32_t has[2];
    int32_t off_pstart[2], off_vstart[2], off_aabb[2], off_xy[2];  // dword offsets
    int32_t off_bstart, off_bnd;                                   // lanes: boundary pieces of the union (dword offsets)
    int32_t off_safe;                                              // lanes: rectangles inside the union (dword offset)
};
constexpr int kSafeRects = 4;   // per env
constexpr int kKinDegree = 4;   // resummed kinematics: polynomials of degree 4 in a^2 (truncation < 2e-10 m at |a| <= 0.5)
enum { KIN_Q0 = 0, KIN_Q4, KIN_Q2, KIN_Q6, KIN_R4, KIN_R8, KIN_R2, KIN_R6 };

// Maps beyond the step kernel's LDS record: one uniform grid per env in HBM (t2d_mapgrid.hip)
constexpr float kGridMargin = 1e-2f;   // parts are registered, and poses looked up, with boxes widened by this much (m)
struct MapGridEnv {
    float x0, y0, inv_cell;   // cell (ix, iy) covers [x0 + ix / inv_cell, ...) x [y0 + iy / inv_cell, ...)
    int32_t nx, ny, cell_off; // the env's cells are cell_start[cell_off + iy * nx + ix]
    int32_t has_lanes, pad;
};
// One registration of a part in a cell: the part's vertices INLINE (padded to four the way load_quad_f32 pads: a triangle repeats its
// first vertex), its boundary pieces' range and the first cell of the part's own cell range -- so a pose that meets the part through
// several cells evaluates it once (in the first cell both ranges share) and the walk is cell -> record, one dependent load.
struct MapItem {
    float xy[8];
    int32_t bnd0, bnd1;       // lane parts: pieces [bnd0, bnd1) of MapGridView::bnd (static parts: 0, 0)
    uint32_t cell_lo;         // ix0 | iy0 << 16 of the cells the part is registered in
    uint32_t kind;            // 0 static, 1 lane
};
static_assert(sizeof(MapItem) == 48, "three 16-byte loads per record");
struct MapGridView {
    const MapGridEnv* env;        // [E]
    const int32_t* cell_start;    // CSR over all cells of all envs (+ 1)
    const MapItem* items;         // per cell: the parts whose (widened) box overlaps it
    const double* bnd;            // boundary pieces of the envs' lane unions, 4 doubles each
};

// What kernels receive by value.
struct LidarView;
struct SceneView;
struct PoolView {
    int32_t n_env, A, N;
    // (sits in the padding behind N, in the cache line every kernel reads when it starts: asked for in the middle of a lone
    // wave's step it is a scalar-cache hit, not a miss of its own -- 512 x 32 envs one launch per step: 12.2 -> 12.7 us
    // with the word further down)  the sub-step count the resummation table below was built for; 0: none
    int32_t kin_n;
    float *x, *y, *heading, *speed, *vx, *vy, *applied0, *applied1;
    const float *act0, *act1;   // the pool's own action fields, or caller-owned device memory (t2d_bind_actions): read only
    int32_t act_stride;         // participant i's actions are act0[i * act_stride], act1[i * act_stride] (1 unless bound strided)
    // IDM agents while caller actions are bound: controlled lanes (idm_ctrl[i] != T2D_IDM_NONE) take their action from the
    // pool's own fields, where the idm kernel writes -- never into the caller's memory.  Null otherwise.
    const uint8_t* idm_ctrl;
    const float *own_act0, *own_act1;
    float *omega_f, *omega_r;  // SingleTrackDrift wheel speeds
    uint32_t *ids, *flags, *env_flags;
    int32_t *cnt_step, *frame_ms;
    uint8_t* status;
    float* reward;
    uint2* record;  // this step's slot of the ring of {reward bits, status word} records
    const double* params;  // [T2D_PARAM_COLS][T2D_MAX_TYPES]
    int32_t n_types;
    // static + lane geometry: one fixed-stride packed record per collide workgroup (see GeoLayout)
    const uint32_t* geo;           // [n_blocks * stride] dwords or null
    GeoLayout geo_layout;
    const float* boundary;         // [4*E] or null
    const uint8_t* boundary_valid; // [E] or null
    // IoU events of the ego (Arrival / NoAction) and reward shaping state, per env
    const double* target_xy;   // [E][8] CCW target quad, or null
    const double* target_c;    // [E][2] its area centroid
    double* last_pose;         // [E][8] pose seen by the previous NoAction.update
    uint8_t* last_valid;       // [E]
    int32_t* cnt_na;           // [E] NoAction.cnt_no_action
    double* max_iou;           // [E] ParkingEnv._max_iou
    double* min_dist;          // [E] ParkingEnv._min_dist_to_target
    const double* snap_min_dist;  // [E] value at the snapshot (episode start)
    const double* time_penalty;   // [max_step + 1] -tanh(cnt / max_step) * time_penalty_scale (host libm), or null
    float* iou;                // [E] last IoU(pose, target), NaN = None
    const float* snap[6];      // episode-start snapshot (x, y, heading, speed, vx, vy) or null
    const uint32_t* snap_ids;
    const float* snap_omega[2];  // wheel speeds at the snapshot; null unless a drift type is present
    int32_t auto_reset;        // t2d_step restores finished envs in its epilogue
    // T2D_OUT_* : which pure OUTPUT columns the integrators store (t2d_set_outputs).  vx / vy of the single-track models and
    // the applied (clipped) action are derived from the state and the action -- nothing on the step path reads them back;
    // a point mass's vx / vy are state and always stored.
    int32_t out_mask;
    // 0: the step launch has the GPU to itself (one wave-round): waves that are BEHIND go first, so that a SIMD's four
    // waves finish together.  1: launches of several pools overlap (t2d_step_groups): waves past the integrator go first,
    // so that workgroups retire and the next launch's can start.  See the priority note in t2d_collide.hip.
    int32_t overlapped;
    // step launch: physical workgroup -> (logical workgroup | wave rotation << 16), or null = identity (t2d_set_step_placement)
    const uint32_t* wgmap;
    // Chained multi-step launch (t2d_step_n; collide_kernel<..., CHAIN>): ONE launch of n_steps x chain_wgs workgroups,
    // workgroup (x = g, y = k) takes step k of the envs of workgroup g and first waits until chain_done[g] == chain_base + k
    // (set by the workgroup that took step k - 1 of the same envs).  Null for the plain one-step launch.
    unsigned long long* chain_done;   // [chain_wgs] {steps of its envs completed so far (a count that only ever grows), XCC id}
    uint32_t* chain_err;        // 1: a workgroup's wait ran out, 2: its predecessor ran on another XCD (never expected; the host checks)
    uint32_t chain_base;        // what chain_done[] holds when this launch starts
    int32_t chain_real_wgs;     // workgroups that own envs; the grid's x extent is rounded up to a multiple of 8
    int32_t split_step;         // the step launch gives every env a workgroup of its own (SPLIT form, small pools of 64-agent envs)
    int32_t loop_steps;         // > 0: the LOOP form -- every workgroup walks through this many steps itself (small pools)
    int32_t pipe_step;          // LOOP form with integrator waves running a step ahead of the event waves (PIPE)
    int32_t chain_k;            // > 1: the chained form with workgroups that take chain_k (= loop_steps) consecutive steps each
    int64_t chain_act_step;     // elements between the action sets of consecutive steps (0: the same actions every step)
    uint2* record_ring;         // the whole ring of per-env result records; step k writes slot (record_slot0 + k) % ring
    int32_t record_slot0;
    // CHAIN launches: what the state was when the fragment started, written by the workgroups of its step 0 from the values
    // they load anyway -- [x | y | heading | speed | vx | vy | ids] of N words each, then cnt_step / frame_ms of n_env words
    // each.  A fragment whose hand-off fails (chain_err) is rolled back to it by the host: t2d_step_n never leaves a pool
    // on stale results (see t2d_api.hip quiesce).  ckpt_tag travels into chain_err[1] with the error: which fragment.
    uint32_t* ckpt;
    uint32_t ckpt_tag;
    uint32_t chain_fault;       // test hook (t2d_debug_chain_fault): 1 = a producer posts a foreign XCC id, 2 = one never posts
    // t2d_step_n on a pool with installed IDM controllers (PIPE form): the integrator waves run the controller ahead of every
    // step themselves (t2d_idm_dev.h) -- its rows, every participant's controller id, and the places t2d_idm_actions writes
    const double* idm_rows;
    const uint8_t* idm_ctrl_all;
    int32_t* idm_leader;
    float *idm_act0_own, *idm_act1_own;
    int32_t idm_n_ctrl;
    const SceneView* regen;   // ego step kernel, regenerating pool: the device copy of the pool's SceneView, else null
    double interval_s;        // (double)interval_ms / 1000 of this launch (PointMass's dt), divided once by the host
    // SingleTrackKinematics, fast variant: the Euler sum of a step RESUMMED instead of iterated (t2d_integrate_dev.h
    // resum_kinematics).  kin_n = the sub-step count the table below was built for (0: none; lanes whose type has another
    // count keep the recurrence loop); kin_coef[i][p] = coefficient of a^(2 i) of polynomial p (KIN_Q0 ...), each the moment
    // of the centred sub-step index it multiplies, divided by its factorial, times the sign / the 1/2, 1/6 of its b-term;
    // kin_geo = {m, M, m - 1/2, m (m - 1) / 2, M^2 / 2, m + 1, (m + 1)^2 / 2, n} with m = (n - 1) / 2, M = n / 2.  Computed by
    // the host (long double) whenever the interval changes; read by the kernels as scalar loads from their argument block.
    double kin_coef[kKinDegree + 1][8];
    double kin_geo[8];
    // pools whose static + lane geometry lives in the HBM grid tier (t2d_mapgrid.hip): the per-participant verdicts
    // (T2D_FLAG_COLLISION_STATIC | T2D_FLAG_OFF_LANE) of t2d_mapgrid.hip's two launches, OR-ed into the flags by the event kernel; else null
    const uint32_t* map_flags;
    unsigned long long* dbg;  // phase cycle accumulators (profiling builds with -DT2D_TIMING only)
    double cell;      // spatial-hash cell edge (m) >= max circum-diameter * 1.001
    double inv_cell;
};

// Lidar inputs (t2d_lidar.hip): plain per-env CSR of the static obstacle rings + beam tables.
struct LidarView {
    const int32_t* env_vert_off;  // [E+1] vertex range of env e, or null (no static obstacles)
    const int32_t* env_vert_cnt;  // null, or [E] vertices in use when envs own fixed-capacity ranges (generated scenes)
    const float* xy;              // [V][4] one record per polygon EDGE: x1, y1, x2, y2 (vertex v -> next vertex of its ring)
    // [V] per edge: index (0..15) of its ring among its env's rings when the ring may take part in the occlusion culling of
    // the scan (CCW, convex, every interior angle with sin >= 0.05, <= 16 rings and <= 48 edges in the env), else 0xff; or null
    const uint8_t* edge_meta;
    const double* beam_pre;       // [n_beams][6] per beam: a = sin, b = -cos of linspace(0, 2pi, n, endpoint=False)[k]
                                  // (lidar.py:161-162) and the four slack-widened bounds of its end point (x hi / lo, y hi / lo)
    double max_range;
    int32_t n_beams, include_participants, ego_index, max_static_verts;
    int32_t max_slots;  // LDS edge slots per env: max_static_verts + 4 * max_agents (when participants are scanned)
    int32_t queue_len;  // (set by launch_lidar) candidates per wave and compaction round
};

// Device-side ParkingLotGenerator (t2d_generate.hip): stream parameters, the per-scene output arrays and -- when the
// scenes are installed in a pool -- the places t2d_set_static_geometry / t2d_set_target_areas / t2d_reset /
// t2d_snapshot would have written: every env owns T2D_GEN_MAX_QUADS polygon slots of 4 vertices.
struct SceneArrays {  // one record per scene, the layout of t2d_generate_parking's outputs
    float* quads; int32_t* quad_id; int32_t* n_quads; double* start; float* target; double* target_heading;
    float* boundary; uint32_t* info;
};
struct SceneView {
    uint64_t seed;
    int64_t first_env, env_stride;  // scene of (env e, episode k) = stream first_env + e + k * env_stride
    double type_proportion, len, wid;
    SceneArrays live;     // [n_env] the scenes the envs are in
    SceneArrays staged;   // [n_env * ring] scenes of coming episodes (episode k of env e in slot k % ring), or nulls
    int32_t ring;         // staged scenes per env (0 = none: regeneration generates on the step's stream)
    int32_t* staged_ep;   // [n_env * ring] episode number held by each staged slot (-1 = empty)
    uint2* refill_list;   // [n_env * ring] {staged slot, episode to generate for it} found by a refill's scan; refill_count[0] of them
    uint32_t* refill_count;
    int32_t* episode;     // [n_env]
    uint32_t* geo;        // the pool's geometry records (capacity layout)
    GeoLayout gl;
    float* lidar_xy; int32_t* lidar_cnt;
    uint8_t* lidar_meta;  // [n_env][4 * T2D_GEN_MAX_QUADS] per edge: its ring's slot when the ring may take part in the scan's occlusion culling, else 0xff
    uint32_t* commit_err; // sticky: an env whose episode ended found no staged scene (scene_commit_kernel)
    float* boundary; double* target_xy; double* target_c;
    float* snap[6]; uint32_t* snap_ids; uint32_t ids_word;
    double* snap_min_dist;
};

// IDM controllers (t2d_idm.hip): parameter sets + one controller id per participant.
struct IdmView {
    const double* rows;      // [n_ctrl][T2D_IDM_COLS]
    const uint8_t* ctrl_id;  // [N] index into rows, T2D_IDM_NONE = action left to the caller
    int32_t* leader;         // [N] out: agent index of the chosen leader inside the env, -1 = none
    int32_t n_ctrl;
};

// The Gym-API host frame (t2d_step_host): where the pack kernel writes what ParkingEnv.step hands its caller, one frame per
// step -- `out` is the device frame (then copied to pinned host memory) or the mapped host frame itself (T2D_FRAME_ZEROCOPY).
struct FrameView {
    char* out;
    t2d_frame_layout lay;
    const double* target_heading;   // [E] or null (diff_heading = NaN)
    const float* target_quads;      // [E][8] scene mode: the live target areas; else null
    const int32_t* episode;         // [E] scene mode, else null
    const uint32_t* commit_err;     // scene regeneration's sticky error word, or null
    uint32_t step_count;
    int32_t ego_index;
};

constexpr int kIdsModelShift = 0;
constexpr int kIdsTypeShift = 8;
constexpr int kIdsActiveShift = 16;

}  // namespace t2d

struct t2d_pool {
    t2d::PoolView v{};
    int device = 0;
    bool have_params = false;
    bool have_reset = false;
    int integrator_variant = 1;
    bool kin_resum = true;   // fast variant: resummed kinematic steps (t2d_set_integrator_variant(pool, 2) keeps the recurrence loop)
    bool kin_resum_forced = false;   // ... also on pools too small to fill the GPU (variant 3: tests)
    bool fused_step = true;  // t2d_step = one launch (integrate + events + status)
    bool ego_kernel = true;  // single-ego pools step with one wave per env (t2d_ego.hip) when they qualify
    bool all_boxes = false;  // every row of the parameter table is T2D_SHAPE_OBB
    t2d_status_config status_cfg{};
    std::string err;
    double host_params[T2D_MAX_TYPES][T2D_PARAM_COLS]{};
    // owned device buffers (freed in t2d_destroy)
    void* field_ptr[T2D_F_COUNT]{};
    size_t field_bytes[T2D_F_COUNT]{};
    double* d_params = nullptr;
    uint32_t* d_geo = nullptr;
    // the HBM grid tier of maps too large for the LDS record (rebuild_geo): device copies of the parts + the grid
    bool grid_tier = false;
    t2d::MapGridView mapgrid{};
    t2d::MapGridEnv* d_grid_env = nullptr;
    int32_t* d_grid_cell_start = nullptr;
    t2d::MapItem* d_grid_items = nullptr;
    uint32_t* d_map_flags = nullptr;
    void* d_grid_seg = nullptr;   // t2d_mapgrid.hip: one segment per sixteen participants (poses, verdict bits, the queue of pairs to decide)
    double* d_grid_bnd = nullptr;
    float* d_boundary = nullptr;
    uint8_t* d_boundary_valid = nullptr;
    // host copies of the CCW-normalised CSR geometry, kind 0 static / 1 lanes
    struct HostGeo {
        bool present = false;
        // polygons as the EVENT kernels see them: CCW, and every polygon with 5..8 vertices replaced by its fan of
        // quads (v0 v1 v2 v3), (v0 v3 v4 v5), (v0 v5 v6 v7) (last part a triangle for odd counts): the union is the
        // polygon, so `intersects` / point-in tests are the OR over the parts, and the kernels only ever meet 3- and
        // 4-vertex polygons (fan_parts in oracle/t2d_oracle.c states the same)
        std::vector<int32_t> env_off, vert_off;
        std::vector<float> xy, aabb;
        // the caller's rings, CCW, undivided: what the lidar scans (a fan's inner diagonals are not walls)
        std::vector<int32_t> ring_env_off, ring_vert_off;
        std::vector<float> ring_xy;
        // lanes only: boundary pieces of the union of each env's lanes, CSR per lane polygon (build_lane_boundary)
        std::vector<int32_t> bnd_off;
        std::vector<double> bnd;
        // lanes only: per env kSafeRects rectangles inside the union of its lanes (build_safe_rects), 4 floats each
        std::vector<float> safe;
    } hgeo[2];
    double *d_target_xy = nullptr, *d_target_c = nullptr, *d_last_pose = nullptr, *d_max_iou = nullptr,
           *d_min_dist = nullptr, *d_snap_min_dist = nullptr;
    uint8_t* d_last_valid = nullptr;
    bool have_target = false;
    // lidar (row f2)
    bool lidar_on = false;
    t2d::LidarView lidar{};
    int32_t *d_lidar_env_off = nullptr, *d_lidar_next = nullptr;
    uint8_t* d_lidar_meta = nullptr;
    float* d_lidar_xy = nullptr;
    double *d_beam_sin = nullptr, *d_beam_cos = nullptr;
    double* d_time_penalty = nullptr;
    bool has_drift = false;   // a T2D_MODEL_DRIFT row is in the parameter table
    float* d_snap_omega[2]{};
    // IDM agents (row f3)
    bool idm_on = false;
    t2d::IdmView idm{};
    double* d_idm_rows = nullptr;
    uint8_t* d_idm_ctrl = nullptr;
    float* d_snap[6]{};      // x, y, heading, speed, vx, vy at episode start
    uint32_t* d_snap_ids = nullptr;
    uint32_t* d_wgmap = nullptr;   // step-launch placement (t2d_debug_set_step_placement)
    bool have_snapshot = false;
    bool auto_reset = false;
    // generated parking scenes (row f4): capacity-layout geometry owned by the device
    bool scene_mode = false, scene_regen = false;
    t2d::SceneView scene{};
    void* d_scene_arrays = nullptr;   // one allocation behind scene.live / scene.staged / scene.episode
    hipStream_t scene_stream = nullptr;   // staged scenes are refilled here, off the step's stream
    hipEvent_t ev_scene_commit = nullptr, ev_scene_refill = nullptr;
    bool scene_refill_pending = false;
    bool scene_commit_failed = false;
    bool scene_committed_in_step = false;   // the last step launch was the ego kernel with the commit in its epilogue
    t2d::SceneView* d_scene_view = nullptr;
    bool scene_commit_used = false;   // a commit launch since the last host synchronisation (its sticky error word is read then)
    int32_t* d_lidar_cnt = nullptr;
    long long step_count = 0;  // t2d_step calls so far (selects the record ring slot)
    int derived_interval = -1; // the interval_ms column T2D_P_SUBSTEPS of the device table was derived for (-1: none yet)
    // chained multi-step launches (t2d_step_n): per-workgroup step counters + one error word behind them
    unsigned long long* d_chain = nullptr;   // chain_slots words, then the error word
    int chain_slots = 0;
    uint32_t chain_count = 0;      // what every counter holds once the launches enqueued so far have run
    bool chain_steps = true;       // t2d_set_step_chaining(pool, 0, *): t2d_step_n falls back to one launch per step
    bool chain_loop = true;        // small pools take the LOOP form (t2d_set_step_chaining(pool, 2, *): never)
    bool chain_pipe2 = true;       // ... and lane waves (PIPE = 2) where the pool has lane polygons
    bool chain_pipe = true;        // small pools: the LOOP form carries integrator waves that run a step ahead (PIPE)
    bool split_steps = true;       // small pools of 64-agent envs step with one env per workgroup (t2d_set_split_step)
    bool chain_priority = true;    // wave priorities of a chained launch: 1 = the rule for overlapping work (PoolView::overlapped)
    bool chain_used = false, chain_failed = false;
    // a failed chained launch (chain_err: 1 = a bounded wait ran out, 2 = producer and consumer of a hand-off on different
    // XCDs): reported ONCE by the next t2d_sync / t2d_download / t2d_step_n; chaining stays off for the pool afterwards.
    // CHAIN-form fragments carry a checkpoint of the state they started from (PoolView::ckpt): the pool is rolled back to it,
    // step count included, and goes on from there with plain launches.
    int chain_err_code = 0;
    bool chain_rolled_back = false;
    long long chain_rollback_step = 0;
    uint32_t* d_ckpt = nullptr;
    bool ckpt_armed = false;       // every multi-step launch since the last quiesce was a CHAIN launch with a checkpoint
    uint32_t chain_sig = 0;        // shape (workgroups, split) of the last CHAIN launch whose counters d_chain holds; 0 = none
    uint32_t chain_fault = 0;      // t2d_debug_chain_fault
    uint32_t types_used = 0;       // bit t: some active participant has type t (t2d_reset)
    int chain_depth = 1;           // steps per workgroup of the chained form of large pools (T2D_CHAIN_DEPTH in the environment)
    int device_cus = 0;            // compute units of the pool's device (read once)
    // result gather (the one collective of the path): RCCL communicator + a stream of its own, so that the steps that
    // follow a fragment do not wait for its all-gather; slot_event[k] != null = a gather that reads record slot k was
    // enqueued and the step about to overwrite that slot must wait for it first
    void* comm = nullptr;          // ncclComm_t created by t2d_comm_init (null: none / world of one)
    int comm_rank = 0, comm_world = 1;
    hipStream_t gather_stream = nullptr;
    hipEvent_t ev_frag_ready = nullptr;
    hipEvent_t ev_gather[T2D_RECORD_RING]{};
    hipEvent_t slot_event[T2D_RECORD_RING]{};
    hipEvent_t last_gather = nullptr;
    // streams with work of this pool possibly in flight (what the set-up calls / t2d_sync wait for)
    static constexpr int kMaxLiveStreams = 4;
    hipStream_t live_streams[kMaxLiveStreams]{};
    int n_live_streams = 0;
    bool live_overflow = false;
    // the Gym-API host frame (t2d_frame_config / t2d_step_host)
    uint32_t frame_sections = 0;
    t2d_frame_layout frame_layout{};
    char* d_frame = nullptr;          // device frame (copy mode)
    char* h_frame[T2D_MAX_HOST_FRAMES]{};   // pinned (and mapped) host frames
    int n_host_frames = 0, frame_turn = 0;
    float* h_actions = nullptr;       // pinned (and mapped) staging of the host actions, [N][2]
    float* d_actions = nullptr;       // device copy of them (copy mode)
    bool act_in_frame = false;        // v.act0 / v.act1 point into d_actions / the mapped h_actions (set by t2d_step_host)
    int64_t act_extent = 0;       // elements readable behind each bound action pointer (t2d_set_action_extent; 0 = not declared)
    double* d_target_heading = nullptr;
    // profiling
    bool profiling = false;
    static constexpr int kMaxProfSteps = 4096;
    hipEvent_t* prof_events = nullptr;  // 2 events per recorded launch
    int prof_kernel[2 * kMaxProfSteps]{};
    int prof_count = 0;
};

// kernel launchers (defined in t2d_integrate.hip / t2d_collide.hip)
namespace t2d {
hipError_t launch_integrate(const PoolView& v, int interval_ms, int variant, bool allow_wide, int only_model, hipStream_t s);
// column T2D_P_SUBSTEPS of the device table for `interval_ms` (rows of the drift model keep their own column 23)
hipError_t launch_derive(double* params, int n_types, int interval_ms, hipStream_t s);
hipError_t launch_collide(const PoolView& v, const t2d_status_config& cfg, bool with_status,
                          int interval_ms, int fuse_variant, hipStream_t s);
// n_steps fused steps in one launch (v.chain_* set by the caller); see PoolView::chain_done
hipError_t launch_step_chain(const PoolView& v, const t2d_status_config& cfg, int interval_ms, int variant, int n_steps,
                             hipStream_t s);
hipError_t launch_ego_step(const PoolView& v, const t2d_status_config& cfg, int interval_ms, int variant, hipStream_t s);
hipError_t step_occupancy(const PoolView& v, int* blocks_per_cu, size_t* lds_bytes);
hipError_t launch_lidar(const PoolView& v, const LidarView& lv, float* out, hipStream_t s);
hipError_t launch_drift(const PoolView& v, int interval_ms, hipStream_t s);
hipError_t launch_verify(const PoolView& v, const float* x, const float* y, const float* heading, const float* speed,
                         int interval_ms, uint8_t* valid, hipStream_t s);
hipError_t launch_idm(const PoolView& v, const IdmView& iv, const int32_t* forced_leader, float* act0_own, float* act1_own,
                      hipStream_t s);
hipError_t launch_restore(const PoolView& v, float* const* snap, const uint32_t* snap_ids, int mode,
                          hipStream_t s);
hipError_t launch_spin(long long ticks_100mhz, hipStream_t s);
size_t map_segment_bytes(int n_participants);   // the walk -> decisions hand-over buffer of launch_map_events
hipError_t launch_map_events(const PoolView& v, const MapGridView& mg, void* segments, uint32_t* out, hipStream_t s);
#ifdef T2D_DEBUG_HOOKS
const char* last_collide_form();   // template arguments of the collide_kernel instantiation the last launch took (t2d_collide.hip)
#endif
hipError_t launch_frame_pack(const PoolView& v, const FrameView& fv, hipStream_t s);
// note that work of this pool was enqueued on `s` by code outside t2d_api.hip (a replayed graph): t2d_sync waits for it
void pool_touch(t2d_pool* p, hipStream_t s);
bool split_eligible(const PoolView& v, const t2d_status_config& cfg, int log2A, int device_cus);
hipError_t launch_parking_scenes(const PoolView& v, const SceneView& sv, int n_env, int mode, hipStream_t s);
hipError_t launch_scene_refill(const SceneView& sv, int n_env, hipStream_t s);
// regenerate = 1: envs whose episode just ended take the scene staged for their next one, sixteen lanes per env
hipError_t launch_scene_commit(const PoolView& v, const SceneView& sv, int n_env, hipStream_t s);
}  // namespace t2d
