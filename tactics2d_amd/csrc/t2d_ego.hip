// t2d_ego.hip -- the ScenarioManager step of SINGLE-EGO environments (max_agents = 1: ParkingEnv, BASELINE.json
// config 2: 4096 vectorised parking envs x 1 ego), SIXTEEN LANES per environment.
//
// Replaces (reference, tactics2d v0.1.9rc3) -- the same functions as the fused step kernel of t2d_collide.hip:
//   SingleTrackKinematics.step / _step   physics/single_track_kinematics.py:126-198 (any of the three models)
//   Vehicle.get_pose                     participant/element/vehicle.py:263-281
//   StaticCollision.update               traffic/event_detection/collision.py:37-43
//   OutBound.update                      traffic/event_detection/out_bound.py:37-48
//   Arrival.update / NoAction.update     arrival.py:32-47, no_action.py:32-53 (IoU of two quads)
//   _ParkingScenarioManager.update + check_status, ParkingEnv.step / _get_reward   envs/parking.py:352-392, 219-256, 148-190
//
// Why a kernel of its own: with one participant per environment the general kernel's mapping (one lane per
// participant) has every lane walk the whole chain alone -- 20 sub-steps, a sweep over the parking lot's quads, two quad
// IoUs of 32 IEEE divisions each, the status epilogue: 17.5 us of dependent latency at 4096 envs (64 waves on 1024
// SIMDs).  Here SIXTEEN lanes share one environment (four envs per wave; 4096 envs = 1024 waves = one per SIMD):
//   * the integrator and the pose are evaluated by all 16 lanes alike (the instruction stream one lane would run);
//   * static collision: one lane per obstacle quad (16 at a time), the verdicts meet in a ballot;
//   * the two IoUs: 2 IoUs x 8 clipped-edge terms = 16 lanes, each term the oracle's clipped_edge_term (4 IEEE
//     divisions in flight), the eight terms of an IoU summed in the oracle's tree ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7))
//     -- a butterfly over neighbouring lanes, fp addition being commutative;
//   * the epilogue runs on the group's first lane.
// (A whole wave per env was measured first: 64 x the integrator's instructions, 25 us -- issue-bound on redundancy.)
// Every value is produced by the arithmetic of the general kernel (same device functions, same operation order), so
// with the exact integrator the two agree bit for bit (tests/test_gpu_ego.py), and with the oracle.
#include "t2d_geom_dev.h"
#include "t2d_integrate_dev.h"
#include "t2d_scene_dev.h"

// s_sleep between two polls of a PIPE progress word, in units of 64 cycles.  A waiting wave shares its SIMD with the wave it
// waits for: measured on cfg3 / cfg4 / cfg5 / cfg2 (us per step) 0: 8.9 / 6.4 / 9.4 / -; 1: 8.8 / 6.4 / 9.2 / 4.9; 4: 8.6 / 6.5 /
// 8.7 / 4.9; 8: 8.6 / 6.6 / 8.6 / 4.9; 16: 8.5 / 6.9 / 8.7; 32: 8.6 / 7.6 / 9.1.
#ifndef T2D_POLL_SLEEP
#define T2D_POLL_SLEEP 4
#endif
namespace t2d {

namespace {

using namespace geom;

constexpr int kEgoBlock = 256;
constexpr int kEgoLanes = 16;                        // lanes per environment
constexpr int kEgoPerBlock = kEgoBlock / kEgoLanes;  // environments per workgroup

T2D_DEV void ego_wave_sync() {   // LDS writes of this wave -> visible to its other lanes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// LOOP = true (t2d_step_n): the launch holds pv.loop_steps steps and every group of lanes walks through them itself --
// step k reads action set k of the bound ring and writes record slot (record_slot0 + k) of the ring; what one trip stored
// (state, counters, detector history) the next reads back with sc1 loads, behind an s_waitcnt vmcnt(0).  An env's chain of
// steps then pays no launch boundary and no start-up per step: with one wave per SIMD they are a quarter of a step.  The
// argument structs are read through a kernarg pointer laundered at the top of every trip (see collide_kernel's LOOP form).
// PIPE = true (a LOOP launch of at most one workgroup per CU): the workgroup carries a second set of four waves, and wave
// 4 + w integrates step k + 1 of the envs of wave w while that wave checks the events of step k -- collide_kernel's PIPE
// form (t2d_collide.hip): speculation on "the episode goes on", integrated again from the snapshot when it did not, nothing
// visible before the commit; the hand-over (x, y, heading, ids per env) and the verdicts go through LDS.  The detector
// history (previous pose, NoAction counter, _max_iou, _min_dist) is the event wave's alone and stays where it was.
constexpr int kEgoSpinLimit = 1 << 17;
T2D_DEV void ego_pipe_wait(uint32_t* word, uint32_t want, uint32_t* err) {   // bounded: a lost wait raises chain_err, never hangs
    int spins = 0;
    while ((int32_t)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - want) < 0) {
        __builtin_amdgcn_s_sleep(T2D_POLL_SLEEP);
        if (++spins > kEgoSpinLimit) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    asm volatile("" ::: "memory");
}
T2D_DEV void ego_pipe_post(uint32_t* word, uint32_t value) {   // (after the wave's plain LDS writes: they complete in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int VARIANT, bool LOOP = false, bool PIPE = false>
__global__ __launch_bounds__(PIPE ? 2 * kEgoBlock : kEgoBlock, 2) void ego_step_kernel(PoolView pv_arg, t2d_status_config cfg_arg, int interval_ms) {
    static_assert(!PIPE || LOOP, "PIPE = a LOOP launch with integrator waves");
    auto pvp = [&]() { if constexpr (LOOP) return late_args(); else return &pv_arg; }();
    auto cfgp = [&]() { if constexpr (LOOP) return late_cfg(); else return &cfg_arg; }();
#define pv (*pvp)
#define cfg (*cfgp)
    // the two quads of each IoU, [env of the workgroup][iou][A | B][x0 y0 ... x3 y3]
    __shared__ double s_quad[kEgoPerBlock][2][2][8];
    // PIPE: the hand-over by step parity, the verdict per env, the progress words of every wave pair
    __shared__ float s_hand[PIPE ? 2 : 1][3][PIPE ? kEgoPerBlock : 1];
    __shared__ uint32_t s_hand_ids[PIPE ? 2 : 1][PIPE ? kEgoPerBlock : 1];
    __shared__ uint32_t s_dec[PIPE ? kEgoPerBlock : 1], s_seq_i[PIPE ? 4 : 1], s_seq_e[PIPE ? 4 : 1];
    const int etid = (int)(threadIdx.x & (kEgoBlock - 1));   // (PIPE: thread 256 + t serves the env of thread t)
    const int lane = threadIdx.x & 63;
    const int l = etid & (kEgoLanes - 1);               // lane inside the env's group
    const int grp = etid / kEgoLanes;                   // env inside the workgroup
    const int gbase = lane & ~(kEgoLanes - 1);          // first lane of the group inside the wave
    const unsigned long long gmask = ((1ull << kEgoLanes) - 1ull) << gbase;
    const int env_raw = blockIdx.x * kEgoPerBlock + grp;
    const bool live = env_raw < pv.n_env;               // (padding groups of the last workgroup run along on env 0, write nothing)
    const int env = live ? env_raw : 0;
    const int idx = env;                                // max_agents == 1
    if constexpr (PIPE) {
        if (threadIdx.x < 4u) s_seq_i[threadIdx.x] = s_seq_e[threadIdx.x] = 0u;
        if (threadIdx.x < (unsigned)kEgoPerBlock) s_dec[threadIdx.x] = 0u;
        __syncthreads();
        if (threadIdx.x >= (unsigned)kEgoBlock) {
            // ======== integrator waves ========
            const int w = etid >> 6;
            auto G = [](auto* q) { return as_global(q); };
            uint32_t ids = ld_state<true>(G(pv.ids) + idx);
            float x = ld_state<true>(G(pv.x) + idx), y = ld_state<true>(G(pv.y) + idx), h = ld_state<true>(G(pv.heading) + idx);
            float v = ld_state<true>(G(pv.speed) + idx), vx = 0.f, vy = 0.f;
            if (((ids >> kIdsModelShift) & 0xff) == T2D_MODEL_POINTMASS) {
                vx = ld_state<true>(G(pv.vx) + idx);
                vy = ld_state<true>(G(pv.vy) + idx);
            }
            const int n_steps = pv.loop_steps;
            const size_t act_step = (size_t)pv.chain_act_step, ai0 = (size_t)idx * pv.act_stride;
            float a0 = pv.act0[ai0], a1 = pv.act1[ai0];
            for (int k = 0; k <= n_steps; ++k) {
                const KernargView ia = late_args();
                float na0 = 0.f, na1 = 0.f;   // the next step's actions: their latency overlaps this integration
                if (k + 1 < n_steps) {
                    na0 = ia->act0[ai0 + (size_t)(k + 1) * act_step];
                    na1 = ia->act1[ai0 + (size_t)(k + 1) * act_step];
                }
                float nx = x, ny = y, nh = h, nv = v, nvx = vx, nvy = vy, app0 = 0.f, app1 = 0.f;
                bool moved = false, has_vel = false;
                bool todo = k < n_steps, decided = k == 0;
                for (;;) {   // at most two rounds: the speculative one, and one for envs whose episode had ended
                    const int model = (ids >> kIdsModelShift) & 0xff, type = (ids >> kIdsTypeShift) & 0xff;
                    if (todo) {
                        nx = x; ny = y; nh = h; nv = v; nvx = vx; nvy = vy;
                        moved = false; has_vel = false;
                    }
                    if (todo && live && ((ids >> kIdsActiveShift) & 0xffu) && model < T2D_MODEL_DRIFT) {
                        auto P = [&](int col) -> double { return ia->params[col * T2D_MAX_TYPES + type]; };
                        const bool pm = model == T2D_MODEL_POINTMASS;
                        const integ::StepOut o = integ::step_participant<VARIANT>(model, P, (double)x, (double)y, (double)h, (double)v,
                                                                                  pm ? (double)vx : 0.0, pm ? (double)vy : 0.0,
                                                                                  (double)a0, (double)a1, interval_ms, ia->interval_s);
                        nx = (float)o.x; ny = (float)o.y; nh = (float)o.heading; nv = (float)o.speed;
                        moved = true;
                        has_vel = o.has_velocity;
                        if (o.has_velocity) {
                            nvx = (float)o.vx;
                            nvy = (float)o.vy;
                        }
                        app0 = (float)o.app0;
                        app1 = (float)o.app1;
                    }
                    if (k < n_steps && l == 0) {   // the hand-over, written while the verdict is still out
                        s_hand[k & 1][0][grp] = nx;
                        s_hand[k & 1][1][grp] = ny;
                        s_hand[k & 1][2][grp] = nh;
                        s_hand_ids[k & 1][grp] = live ? ids : 0u;
                    }
                    if (decided) break;
                    ego_pipe_wait(&s_seq_e[w], (uint32_t)k, ia->chain_err);
                    decided = true;
                    const bool done = live && s_dec[grp] != 0u;
                    if (__ballot(done) == 0ull) break;
                    if (done) {   // back to the snapshot (the state arrays are the integrator's to write)
                        const float r0 = ia->snap[0][idx], r1 = ia->snap[1][idx], r2 = ia->snap[2][idx], r3 = ia->snap[3][idx];
                        const float r4 = ia->snap[4][idx], r5 = ia->snap[5][idx];
                        const uint32_t rid = ia->snap_ids[idx];
                        if (l == 0) {
                            ia->x[idx] = r0; ia->y[idx] = r1; ia->heading[idx] = r2; ia->speed[idx] = r3;
                            ia->vx[idx] = r4; ia->vy[idx] = r5; ia->ids[idx] = rid;
                        }
                        x = r0; y = r1; h = r2; v = r3; vx = r4; vy = r5; ids = rid;
                    }
                    todo = done && k < n_steps;
                    if (k == n_steps) break;
                }
                if (k == n_steps) break;
                ego_pipe_post(&s_seq_i[w], (uint32_t)k + 1u);   // commit: the progress word, then the state arrays
                if (moved && l == 0) {
                    ia->x[idx] = nx; ia->y[idx] = ny; ia->heading[idx] = nh; ia->speed[idx] = nv;
                    if (has_vel && (((ids >> kIdsModelShift) & 0xff) == T2D_MODEL_POINTMASS || (ia->out_mask & T2D_OUT_VELOCITY))) {
                        ia->vx[idx] = nvx;
                        ia->vy[idx] = nvy;
                    }
                    if (ia->out_mask & T2D_OUT_APPLIED) {
                        ia->applied0[idx] = app0;
                        ia->applied1[idx] = app1;
                    }
                }
                x = nx; y = ny; h = nh; v = nv;
                if (has_vel) {
                    vx = nvx;
                    vy = nvy;
                }
                a0 = na0;
                a1 = na1;
            }
            return;
        }
    }
    int step_k = 0;
    for (;;) {   // (one trip unless LOOP)
    if constexpr (LOOP) {
        pvp = late_args();
        cfgp = late_cfg();
    }
    auto G = [](auto* q) { return as_global(q); };

    // ---------------- loads (group-uniform addresses) -------------------------------------------------------------
    uint32_t ids = 0;
    float fx = 0.f, fy = 0.f, fh = 0.f, fv = 0.f, fa0 = 0.f, fa1 = 0.f;
    if constexpr (!PIPE) {
        ids = ld_state<LOOP>(G(pv.ids) + idx);
        fx = ld_state<LOOP>(G(pv.x) + idx); fy = ld_state<LOOP>(G(pv.y) + idx); fh = ld_state<LOOP>(G(pv.heading) + idx);
        fv = ld_state<LOOP>(G(pv.speed) + idx);
        const size_t ai = (size_t)idx * pv.act_stride + (LOOP ? (size_t)step_k * (size_t)pv.chain_act_step : 0);
        fa0 = pv.act0[ai]; fa1 = pv.act1[ai];
        if (pv.idm_ctrl && pv.idm_ctrl[idx] != T2D_IDM_NONE) {
            fa0 = pv.own_act0[idx];
            fa1 = pv.own_act1[idx];
        }
    }
    const int pre_cnt = ld_state<LOOP>(G(pv.cnt_step) + env), pre_frame = ld_state<LOOP>(G(pv.frame_ms) + env);
    float bxmin = 0, bxmax = 0, bymin = 0, bymax = 0;
    bool has_boundary = false;
    if (pv.boundary) {
        const float4 b = reinterpret_cast<const float4*>(pv.boundary)[env];
        bxmin = b.x; bxmax = b.y; bymin = b.z; bymax = b.w;
        has_boundary = pv.boundary_valid ? pv.boundary_valid[env] != 0 : true;
    }
    // this env's obstacle quads in the packed geometry record of its group of envs (t2d_pool.h GeoLayout)
    const auto& gl = pv.geo_layout;
    int p0 = 0, p1 = 0;
    const uint32_t* rec = nullptr;
    if (pv.geo && gl.has[0]) {
        rec = pv.geo + (size_t)(env / gl.epb) * gl.stride;
        const int* pstart = reinterpret_cast<const int*>(rec) + gl.off_pstart[0];
        const int el = env % gl.epb;
        p0 = pstart[el];
        p1 = pstart[el + 1];
    }

    // The inputs of the IoU phase and of the epilogue that do not depend on the new pose, requested now: one wave per SIMD
    // hides nothing, so fetched where they are used each of them is a memory round trip of its own at the end of the chain
    // (integrator -> pose -> obstacles -> IoUs -> status); requested here they arrive while the integrator runs.
    const bool iou_on = cfg.check_no_action || cfg.check_arrival;
    const int iou_k = l >> 3, iou_t = l & 7;   // lane's role in the IoU phase: which IoU, which coordinate / term
    uint8_t pre_last_valid = 0;
    double pre_other = 0.0;
    int pre_cna = 0;
    double pre_max_iou = 0.0, pre_min_dist = 0.0, pre_tcx = 0.0, pre_tcy = 0.0;
    if (iou_on) {
        pre_last_valid = ld_state<LOOP>(G(pv.last_valid) + env);
        const double* other = iou_k == 0 ? pv.last_pose + 8 * (size_t)env : (pv.target_xy ? pv.target_xy + 8 * (size_t)env : nullptr);
        if (other) pre_other = ld_state<LOOP>(G(other) + iou_t);   // (the previous pose is written by every trip)
        pre_cna = ld_state<LOOP>(G(pv.cnt_na) + env);
    }
    if (cfg.shaped_reward) {
        pre_max_iou = ld_state<LOOP>(G(pv.max_iou) + env);
        if (pv.target_c) {
            pre_tcx = pv.target_c[2 * (size_t)env];
            pre_tcy = pv.target_c[2 * (size_t)env + 1];
            pre_min_dist = ld_state<LOOP>(G(pv.min_dist) + env);
        }
    }
    [[maybe_unused]] scene::StagedPart staged{};   // regenerating pool: the lane's share of the env's next lot (t2d_scene_dev.h)
    if constexpr (!LOOP) {
        if (pv.regen) staged = scene::fetch_staged(*pv.regen, env, l);
    }
    if constexpr (PIPE) {   // this step's state, committed by the pair's integrator wave (the loads above are in flight meanwhile)
        ego_pipe_wait(&s_seq_i[etid >> 6], (uint32_t)step_k + 1u, pv.chain_err);
        ids = s_hand_ids[step_k & 1][grp];
        fx = s_hand[step_k & 1][0][grp];
        fy = s_hand[step_k & 1][1][grp];
        fh = s_hand[step_k & 1][2][grp];
    }
    const bool active = live && ((ids >> kIdsActiveShift) & 0xffu);
    const int type = (ids >> kIdsTypeShift) & 0xff;
    const int model = (ids >> kIdsModelShift) & 0xff;
    auto P = [&](int col) -> double { return pv.params[col * T2D_MAX_TYPES + type]; };

    // ---------------- physics: one PhysicsModelBase.step, the group's lanes all the same --------------------------
    if (!PIPE && active && model < T2D_MODEL_DRIFT) {   // (SingleTrackDrift participants are integrated by drift_kernel)
        double pvx = 0.0, pvy = 0.0;
        if (model == T2D_MODEL_POINTMASS) {
            pvx = (double)ld_state<LOOP>(G(pv.vx) + idx);
            pvy = (double)ld_state<LOOP>(G(pv.vy) + idx);
        }
        const integ::StepOut o = integ::step_participant<VARIANT>(model, P, (double)fx, (double)fy, (double)fh, (double)fv, pvx,
                                                                  pvy, (double)fa0, (double)fa1, interval_ms, pv.interval_s);
        fx = (float)o.x;
        fy = (float)o.y;
        fh = (float)o.heading;
        if (l == 0) {
            pv.x[idx] = fx;
            pv.y[idx] = fy;
            pv.heading[idx] = fh;
            pv.speed[idx] = (float)o.speed;
            if (o.has_velocity && (model == T2D_MODEL_POINTMASS || (pv.out_mask & T2D_OUT_VELOCITY))) {
                pv.vx[idx] = (float)o.vx;
                pv.vy[idx] = (float)o.vy;
            }
            if (pv.out_mask & T2D_OUT_APPLIED) {
                pv.applied0[idx] = (float)o.app0;
                pv.applied1[idx] = (float)o.app1;
            }
        }
    }
    double pre_tp = 0.0;
    if (pv.time_penalty && cfg.max_step > 0) {
        const int c = pre_cnt + 1;
        pre_tp = pv.time_penalty[c < cfg.max_step ? c : cfg.max_step];
    }

    // ---------------- pose, out-of-bound, conservative fp32 box (as in collide_kernel phase 1) ---------------------
    Quad A{};
    bool ego_obb = false;
    uint32_t f = 0;
    float box_lo_x = 0, box_hi_x = 0, box_lo_y = 0, box_hi_y = 0;
    const double cx = (double)fx, cy = (double)fy;
    // (an ego whose pose is not finite takes no part in event detection -- no flag, no IoU -- like the general kernel's)
    if (active && __builtin_isfinite(fx) && __builtin_isfinite(fy) && __builtin_isfinite(fh)) {
        const int kind = (int)P(T2D_P_SHAPE);
        const double L = P(T2D_P_LENGTH), W = P(T2D_P_WIDTH);
        double lo_x, hi_x, lo_y, hi_y;
        bool out = false;
        if (kind == T2D_SHAPE_OBB) {
            ego_obb = true;
            double s, c;
            sincos_det((double)fh, s, c);
            const double hl = 0.5 * L, hw = 0.5 * W;
            const double lx[4] = {hl, hl, -hl, -hl};
            const double ly[4] = {-hw, hw, hw, -hw};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                A.x[k] = c * lx[k] - s * ly[k] + cx;
                A.y[k] = s * lx[k] + c * ly[k] + cy;
            }
            lo_x = hi_x = A.x[0];
            lo_y = hi_y = A.y[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                lo_x = A.x[k] < lo_x ? A.x[k] : lo_x; hi_x = A.x[k] > hi_x ? A.x[k] : hi_x;
                lo_y = A.y[k] < lo_y ? A.y[k] : lo_y; hi_y = A.y[k] > hi_y ? A.y[k] : hi_y;
            }
            if (has_boundary)
                out = lo_x < (double)bxmin || hi_x > (double)bxmax || lo_y < (double)bymin || hi_y > (double)bymax;
        } else {   // (a circular ego never reaches this kernel: the host keeps such pools on the general one)
            const double rad = 0.5 * W;
            lo_x = cx - rad; hi_x = cx + rad; lo_y = cy - rad; hi_y = cy + rad;
            if (has_boundary)
                out = cx - rad < (double)bxmin || cx + rad > (double)bxmax || cy - rad < (double)bymin ||
                      cy + rad > (double)bymax;
        }
        if (out) f |= T2D_FLAG_OUT_BOUND;
        box_lo_x = (float)lo_x; box_lo_x -= __builtin_fabsf(box_lo_x) * 1.2e-7f + 1e-6f;
        box_hi_x = (float)hi_x; box_hi_x += __builtin_fabsf(box_hi_x) * 1.2e-7f + 1e-6f;
        box_lo_y = (float)lo_y; box_lo_y -= __builtin_fabsf(box_lo_y) * 1.2e-7f + 1e-6f;
        box_hi_y = (float)hi_y; box_hi_y += __builtin_fabsf(box_hi_y) * 1.2e-7f + 1e-6f;
    }

    // ---------------- static collision: one obstacle quad per lane, 16 at a time ------------------------------------
    {
        bool hit = false;
        for (int c0 = 0;; c0 += kEgoLanes) {
            if (__ballot(ego_obb && p0 + c0 < p1) == 0ull) break;
            const int p = p0 + c0 + l;
            bool near = false;
            int v0 = 0, n = 0;
            if (ego_obb && p < p1) {
                const float4 bb = reinterpret_cast<const float4*>(rec + gl.off_aabb[0])[p];
                const int* vstart = reinterpret_cast<const int*>(rec) + gl.off_vstart[0];
                v0 = vstart[p];
                n = vstart[p + 1] - v0;
                // boxes that do not meet cannot intersect (the pose box is rounded outwards: strictly conservative)
                near = !(bb.x > box_hi_x || bb.y < box_lo_x || bb.z > box_hi_y || bb.w < box_lo_y);
            }
            if (__ballot(near) != 0ull) {
                if (near) {
                    const float* xy = reinterpret_cast<const float*>(rec + gl.off_xy[0]);
                    hit |= sat_quads(A, load_quad_f32(xy + 2 * v0, n));
                }
            }
        }
        if ((__ballot(hit) & gmask) != 0ull) f |= T2D_FLAG_COLLISION_STATIC;
    }
    if (live && l == 0) {
        pv.flags[idx] = f;
        pv.env_flags[env] = f;
    }

    // ---------------- the two quad IoUs of the ego: 16 lanes = 2 IoUs x 8 clipped-edge terms -----------------------
    double iou_na = 0.0, iou_ar = 0.0;
    if (__ballot(ego_obb && (cfg.check_no_action || cfg.check_arrival)) != 0ull) {
        const bool mine = ego_obb && (cfg.check_no_action || cfg.check_arrival);
        const int k = l >> 3;      // 0: NoAction (pose vs the previous pose), 1: Arrival (pose vs the target bay)
        const int t = l & 7;       // term: 0-3 = edges of the pose clipped to the other quad, 4-7 = the other way round
        const bool want = mine && (k == 0 ? (cfg.check_no_action && pre_last_valid)
                                          : (cfg.check_arrival && pv.target_xy != nullptr));
        {   // lane t of each IoU writes coordinate t of A (the pose, from registers) and of B
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v = t == 2 * q ? A.x[q] : v;
                v = t == 2 * q + 1 ? A.y[q] : v;
            }
            s_quad[grp][k][0][t] = v;
            s_quad[grp][k][1][t] = want ? pre_other : 0.0;
        }
        ego_wave_sync();
        double value = 0.0;
        if (__ballot(want) != 0ull) {
            Quad QA, QB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                QA.x[q] = s_quad[grp][k][0][2 * q]; QA.y[q] = s_quad[grp][k][0][2 * q + 1];
                QB.x[q] = s_quad[grp][k][1][2 * q]; QB.y[q] = s_quad[grp][k][1][2 * q + 1];
            }
            const bool strict = t >= 4;   // edges of B are clipped to A with coincident boundary pieces dropped
            const double* S = s_quad[grp][k][strict ? 1 : 0];
            const int i0 = t & 3, i1 = (t + 1) & 3;
            const double p0x = S[2 * i0], p0y = S[2 * i0 + 1], p1x = S[2 * i1], p1y = S[2 * i1 + 1];
            double s = clipped_edge_term(p0x, p0y, p1x, p1y, strict ? QA : QB, strict, QA.x[0], QA.y[0]);
            // ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)): the terms sit on neighbouring lanes
            s = s + __shfl_xor(s, 1);
            s = s + __shfl_xor(s, 2);
            s = s + __shfl_xor(s, 4);
            double inter = s;
            if (inter < 0.0) inter = 0.0;
            const double uni = quad_area2(QA) + quad_area2(QB) - inter;
            value = want ? inter / uni : 0.0;
        }
        iou_na = __shfl(value, gbase);
        iou_ar = __shfl(value, gbase + 8);
    }

    // ---------------- status / reward epilogue (collide_kernel phase 3), the group's first lane --------------------
    bool ended = false;   // (first lane: this step ended the env's episode)
    if (live && l == 0) {
    const int cnt = pre_cnt + 1;  // parking.py:353
    pv.cnt_step[env] = cnt;
    pv.frame_ms[env] = pre_frame + interval_ms;
    int scen = T2D_SCENARIO_NORMAL, traf = T2D_TRAFFIC_NORMAL;
    double iou = 0.0;
    bool has_iou = false;
    if (cfg.max_step > 0 && cnt > cfg.max_step) {
        scen = T2D_SCENARIO_TIME_EXCEEDED;  // later detectors are not updated (parking.py:366-369)
    } else {
        bool na = false;
        if (cfg.check_no_action && ego_obb) {  // NoAction.update (no_action.py:41-53)
            double* last = pv.last_pose + 8 * (size_t)env;
            int cna = pre_cna;
            if (!pre_last_valid) {
                pv.last_valid[env] = 1;
            } else {
                cna = iou_na > (double)cfg.no_action_iou ? cna + 1 : 0;
                pv.cnt_na[env] = cna;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                last[2 * k] = A.x[k];
                last[2 * k + 1] = A.y[k];
            }
            na = cna > cfg.no_action_max_step;
        }
        if (na) {
            traf = T2D_TRAFFIC_NO_ACTION_QUIRK;  // parking.py:373 writes ScenarioStatus.NO_ACTION here
        } else if (f & T2D_FLAG_OUT_BOUND) {
            scen = T2D_SCENARIO_OUT_BOUND;
        } else if (f & T2D_FLAG_COLLISION_STATIC) {
            scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_STATIC;
        } else if (cfg.check_arrival && pv.target_xy && ego_obb) {  // Arrival.update (arrival.py:42-47)
            iou = iou_ar;
            has_iou = true;
            if (iou >= (double)cfg.arrival_threshold) scen = T2D_SCENARIO_COMPLETED;
        }
    }
    double rd;  // ParkingEnv._get_reward (envs/parking.py:148-190)
    if (traf == T2D_TRAFFIC_COLLISION_STATIC) rd = cfg.reward_collision;
    else if (scen == T2D_SCENARIO_TIME_EXCEEDED || scen == T2D_SCENARIO_NO_ACTION) rd = cfg.reward_time_exceed;
    else if (scen == T2D_SCENARIO_OUT_BOUND) rd = cfg.reward_out_bound;
    else if (scen == T2D_SCENARIO_COMPLETED) rd = cfg.reward_completed;
    else {
        rd = cfg.max_step > 0 ? (pv.time_penalty ? pre_tp : -tanh((double)cnt / (double)cfg.max_step) * (double)cfg.time_penalty_scale) : 0.0;
        if (cfg.shaped_reward) {
            double mi = pre_max_iou;
            double iou_reward = 0.0;
            if (has_iou) iou_reward = mi == -INFINITY ? iou : iou - mi;
            rd = rd + iou_reward;
            if (has_iou) pv.max_iou[env] = mi > iou ? mi : iou;
            if (pv.target_c) {
                const double dx = cx - pre_tcx;
                const double dy = cy - pre_tcy;
                const double d = __builtin_sqrt(dx * dx + dy * dy);
                const double md = pre_min_dist;
                if (d < md) {
                    rd += (md - d) * (double)cfg.dist_reward_scale;
                    pv.min_dist[env] = d;
                }
            }
        }
    }
    const float r = (float)rd;
    pv.iou[env] = has_iou ? (float)iou : __builtin_nanf("");
    const bool terminated = scen == T2D_SCENARIO_COMPLETED;
    const bool truncated = !terminated && (scen != T2D_SCENARIO_NORMAL || traf != T2D_TRAFFIC_NORMAL);
    uchar4 st;
    st.x = (unsigned char)scen; st.y = (unsigned char)traf;
    st.z = terminated; st.w = truncated;
    reinterpret_cast<uchar4*>(pv.status)[env] = st;
    pv.reward[env] = r;
    const uint2 recv = make_uint2(__float_as_uint(r), (uint32_t)scen | (uint32_t)traf << 8 |
                                                        (uint32_t)terminated << 16 | (uint32_t)truncated << 24);
    if (LOOP) pv.record_ring[(size_t)((pv.record_slot0 + step_k) & (T2D_RECORD_RING - 1)) * (size_t)pv.n_env + env] = recv;
    else pv.record[env] = recv;
    if (PIPE && pv.auto_reset) s_dec[grp] = (terminated || truncated) ? 1u : 0u;   // (read by the integrator wave before it commits)
    ended = terminated || truncated;
    // (a regenerating pool: the episode continues in another lot -- below, by all the group's lanes -- not at this lot's start)
    if (pv.auto_reset && ended && (LOOP || !pv.regen)) {  // ParkingEnv.reset: state, counters, detector state back to the start
        // (every snapshot value first, then the stores: as load / store pairs each pair waits for its own memory round trip)
        const double smd = pv.snap_min_dist[env];
        const float r0 = pv.snap[0][idx], r1 = pv.snap[1][idx], r2 = pv.snap[2][idx], r3 = pv.snap[3][idx];
        const float r4 = pv.snap[4][idx], r5 = pv.snap[5][idx];
        const uint32_t rid = pv.snap_ids[idx];
        const bool drift = pv.snap_omega[0] != nullptr;
        float w0 = 0.f, w1 = 0.f;
        if (drift) {
            w0 = pv.snap_omega[0][idx];
            w1 = pv.snap_omega[1][idx];
        }
        pv.cnt_step[env] = 0;
        pv.frame_ms[env] = 0;
        pv.last_valid[env] = 0;
        pv.cnt_na[env] = 0;
        pv.max_iou[env] = -INFINITY;
        pv.min_dist[env] = smd;
        if (!PIPE) {   // (PIPE: the state arrays are the integrator wave's to write)
            pv.x[idx] = r0;
            pv.y[idx] = r1;
            pv.heading[idx] = r2;
            pv.speed[idx] = r3;
            pv.vx[idx] = r4;
            pv.vy[idx] = r5;
            pv.ids[idx] = rid;
            if (drift) {
                pv.omega_f[idx] = w0;
                pv.omega_r[idx] = w1;
            }
        }
    }
    }   // (the group's first lane)
    if constexpr (!LOOP) {
        // t2d_parking_scenes(regenerate = 1): an env whose episode ended moves into the lot staged for its next episode, here
        // instead of in a launch of its own behind the step -- the group's sixteen lanes are the sixteen of
        // scene::commit_staged (t2d_scene_dev.h); its fence between the loads and the stores also puts the first lane's
        // epilogue stores (counters, detector state) ahead of the ones that start the new episode
        if (pv.regen) {   // (uniform)
            if (__shfl((int)ended, gbase)) scene::commit_staged(pv, *pv.regen, env, l, staged);
        }
    }
    if (!LOOP) break;
    if constexpr (PIPE) ego_pipe_post(&s_seq_e[etid >> 6], (uint32_t)step_k + 1u);   // the verdicts first: nobody waits for the stores but this wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this trip's stores are in the L2 before the next trip's sc1 loads
    ego_wave_sync();
    if (++step_k >= pv.loop_steps) break;
    }
#undef pv
#undef cfg
}

}  // namespace

hipError_t launch_ego_step(const PoolView& v, const t2d_status_config& cfg, int interval_ms, int variant, hipStream_t s) {
    const dim3 grid((v.n_env + kEgoPerBlock - 1) / kEgoPerBlock), block(kEgoBlock);
    if (v.loop_steps > 0 && v.pipe_step) {   // ... with integrator waves a step ahead
        const dim3 block2(2 * kEgoBlock);
        if (variant == 0) hipLaunchKernelGGL((ego_step_kernel<0, true, true>), grid, block2, 0, s, v, cfg, interval_ms);
        else hipLaunchKernelGGL((ego_step_kernel<1, true, true>), grid, block2, 0, s, v, cfg, interval_ms);
        return hipGetLastError();
    }
    if (v.loop_steps > 0) {   // t2d_step_n: the groups walk through the steps themselves
        if (variant == 0) hipLaunchKernelGGL((ego_step_kernel<0, true>), grid, block, 0, s, v, cfg, interval_ms);
        else hipLaunchKernelGGL((ego_step_kernel<1, true>), grid, block, 0, s, v, cfg, interval_ms);
        return hipGetLastError();
    }
    if (variant == 0) hipLaunchKernelGGL(ego_step_kernel<0>, grid, block, 0, s, v, cfg, interval_ms);
    else hipLaunchKernelGGL(ego_step_kernel<1>, grid, block, 0, s, v, cfg, interval_ms);
    return hipGetLastError();
}

}  // namespace t2d
