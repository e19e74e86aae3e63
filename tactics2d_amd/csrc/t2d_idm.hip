// t2d_idm.hip -- on-device scripted agents: IDM car-following (scope row f3).
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   IDMController.step                controller/idm_controller.py:59-93   (free flow | car following, np.clip)
//   IDMController._idm_acceleration   controller/idm_controller.py:95-141  (s*, (v/v_des)^delta, (s*/s)^2)
// The reference leaves the choice of `leading_state` to its caller and ships no caller; the leader rule
// here is BUILD-DEFINED (oracle t2do_idm is the definition): among the other active participants of
// the env, those ahead (longitudinal offset 0 < lon <= horizon along the own heading) inside the own
// corridor (|lateral offset| <= lane_half_width), the one with the smallest lon, lowest index on ties.
// A caller that knows the leader (the reference's calling convention) passes it per participant instead.
//
// One lane per participant, EPB = 256 / A_pad whole envs per workgroup.  The env's (x, y, speed, active)
// go through LDS once; every controlled lane sweeps its env's list with wave-uniform LDS reads (broadcast,
// conflict free), ~12 fp64 operations per candidate, then evaluates the IDM law once.  Output: accel ->
// the pool's act0 field, steer 0 -> its act1 field (the reference returns (steering, acceleration); the physics
// models take (accel, steer)), leader index -> T2D_F_LEADER.  HBM: 17 B read + 12 B written per participant.
#include <type_traits>

#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

constexpr int kIdmBlock = 256;

// IDMController.step + _idm_acceleration for one participant (oracle t2do_idm_accel)
struct IdmRow {
    double des, T, s0, amax, b, delta, hw, horizon;
};

T2D_DEV double idm_law(const IdmRow& c, double v, bool has_lead, double dx, double dy, double v_lead) {
    const double des = c.des, T = c.T, s0 = c.s0, amax = c.amax, b = c.b, delta = c.delta;
    // (v / v_des)^delta once, ahead of the regimes: a wave with leaders for some lanes and none for others runs both
    // branches, and the power (deterministic log + exp) is the bulk of either
    const double pw = des > 0.0 ? pow_det(v / des, delta) : 0.0;
    double a;
    if (!has_lead) {  // :75-85
        if (des > 0.0) a = amax * (1.0 - pw);
        else a = v > 0.0 ? -b : 0.0;
    } else {  // :106-141
        const double dist = __builtin_sqrt(dx * dx + dy * dy);  // np.hypot
        const double dv = v_lead - v;
        double s_star = s0 + v * T + (v * dv) / (2.0 * __builtin_sqrt(amax * b));
        if (s0 > s_star) s_star = s0;  // max(s_star, min_spacing)
        if (dist > 0.0) {
            const double term = des > 0.0 ? pw : (v > 0.0 ? 1.0 : 0.0);
            const double q = s_star / dist;
            a = amax * (1.0 - term - q * q);
        } else {
            a = -b;
        }
    }
    return clipd(a, -b, amax);  // np.clip :90
}

// the smallest double above h for h >= 0 (h itself when it is +inf or NaN: `lon < h` then equals `lon <= h` for every
// finite lon); 0 for h < 0, where no offset is both > 0 and <= h
T2D_DEV double just_above(double h) {
    if (!(h >= 0.0)) return h != h ? h : 0.0;
    if (h == __builtin_inf()) return h;
    return __longlong_as_double(__double_as_longlong(h + 0.0) + 1);   // h + 0.0: -0.0 -> +0.0
}

// act0_own / act1_own: the POOL's action fields (T2D_F_ACT0 / ACT1) -- never caller-owned memory bound with
// t2d_bind_actions; while a binding is in effect the integrators take the controlled lanes' actions from there
__global__ __launch_bounds__(kIdmBlock) void idm_kernel(PoolView pv, IdmView iv, const int32_t* forced, float* act0_own,
                                                        float* act1_own, int log2A) {
    // (x, y) as fp64 pairs, NaN for inactive slots: every comparison of the sweep is then false for
    // them, and a participant never selects itself (its own offset is exactly 0, not > 0)
    __shared__ double2 s_xy[kIdmBlock];
    __shared__ float s_v[kIdmBlock];
    const int tid = threadIdx.x;
    // every kernel argument of the load phase in one scalar round trip (see collide_kernel)
    // (global address space kept through the asm: plain pointers come out of it generic, i.e. as flat loads, which
    // count against the LDS wait counter as well)
    const T2D_GLOBAL uint32_t* a_ids = (const T2D_GLOBAL uint32_t*)pv.ids;
    const T2D_GLOBAL float *a_x = (const T2D_GLOBAL float*)pv.x, *a_y = (const T2D_GLOBAL float*)pv.y,
                           *a_h = (const T2D_GLOBAL float*)pv.heading, *a_v = (const T2D_GLOBAL float*)pv.speed;
    const T2D_GLOBAL uint8_t* a_ctrl = (const T2D_GLOBAL uint8_t*)iv.ctrl_id;
    const T2D_GLOBAL double* a_rows = (const T2D_GLOBAL double*)iv.rows;
    int a_n_env = pv.n_env, a_A = pv.A;
    asm volatile("" : "+s"(a_ids), "+s"(a_x), "+s"(a_y), "+s"(a_h), "+s"(a_v), "+s"(a_ctrl), "+s"(a_rows), "+s"(a_n_env), "+s"(a_A));
    const int A_pad = 1 << log2A;
    const int epb = kIdmBlock >> log2A;
    const int env_local = tid >> log2A;
    const int agent = tid & (A_pad - 1);
    const int env = blockIdx.x * epb + env_local;
    const bool valid = env < a_n_env && agent < a_A;
    const int idx = valid ? env * a_A + agent : 0;
    float fx = 0, fy = 0, fh = 0, fv = 0;
    uint32_t ids = 0;
    int ctrl = T2D_IDM_NONE;
    if (valid) {
        ids = a_ids[idx];
        fx = a_x[idx];
        fy = a_y[idx];
        fh = a_h[idx];
        fv = a_v[idx];
        ctrl = a_ctrl[idx];
    }
    const bool active = valid && ((ids >> kIdsActiveShift) & 0xffu);
    // the controller's row and the heading's sine / cosine before the barrier: the row is a second dependent round trip
    // to memory (after ctrl_id) and every wave of the launch reaches it at the same time -- it overlaps the LDS fill
    const bool controlled = active && ctrl != T2D_IDM_NONE && ctrl < iv.n_ctrl;
    IdmRow c{};
    double sn = 0.0, cs = 1.0;
    if (controlled) {
        const T2D_GLOBAL double* r = a_rows + (size_t)ctrl * T2D_IDM_COLS;
        c.des = r[T2D_IDM_DESIRED_SPEED]; c.T = r[T2D_IDM_TIME_HEADWAY]; c.s0 = r[T2D_IDM_MIN_SPACING];
        c.amax = r[T2D_IDM_MAX_ACCEL]; c.b = r[T2D_IDM_COMF_DECEL]; c.delta = r[T2D_IDM_DELTA];
        c.hw = r[T2D_IDM_LANE_HALF_WIDTH]; c.horizon = r[T2D_IDM_HORIZON];
    }
    const double qnan = __builtin_nan("");
    s_xy[tid] = active ? make_double2((double)fx, (double)fy) : make_double2(qnan, qnan);
    s_v[tid] = fv;
    if (controlled) sincos_det((double)fh, sn, cs);
    __syncthreads();
    if (!valid) return;
    int lead = -1;
    if (controlled) {
        const double hw = c.hw, horizon = c.horizon;
        const int base = env_local << log2A;
        // `lon <= horizon` rides on the running minimum: it starts at the first double above the horizon and a candidate
        // must be strictly below it -- one compare and two selects fewer per candidate than testing the horizon apart
        double best = just_above(horizon);
        const double x0 = (double)fx, y0 = (double)fy;
        const int want = forced ? forced[idx] : T2D_IDM_LEADER_SEARCH;
        if (want >= 0 && want < pv.A && want != agent && s_xy[base + want].x == s_xy[base + want].x)
            lead = want;  // the caller's leading_state
        if (want == T2D_IDM_LEADER_SEARCH) {
            auto sweep = [&](int j0, int j1) {
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {
                    const double2 q = s_xy[base + j];
                    const double dx = q.x - x0, dy = q.y - y0;
                    const double lon = __builtin_fma(dx, cs, dy * sn);
                    const double lat = __builtin_fma(dy, cs, -(dx * sn));
                    const bool take = lon > 0.0 && lon < best && __builtin_fabs(lat) <= hw;   // strict: lowest index on ties
                    best = take ? lon : best;
                    lead = take ? j : lead;
                }
            };
            // the same sweep with constant bounds: fully unrolled, the candidate's index is an inline constant of its
            // select and its LDS address an immediate offset (no loop counter, no index register: ~2.5 of ~16 issued
            // instructions per candidate)
            auto sweep_const = [&](auto j0c, auto j1c) {
#pragma unroll
                for (int j = decltype(j0c)::value; j < decltype(j1c)::value; ++j) {
                    const double2 q = s_xy[base + j];
                    const double dx = q.x - x0, dy = q.y - y0;
                    const double lon = __builtin_fma(dx, cs, dy * sn);
                    const double lat = __builtin_fma(dy, cs, -(dx * sn));
                    const bool take = lon > 0.0 && lon < best && __builtin_fabs(lat) <= hw;
                    best = take ? lon : best;
                    lead = take ? j : lead;
                }
            };
            // wave priority by progress (see the step kernel): the launch is one wave-round, a SIMD's waves should finish
            // together.  The sweep in quarters, the quarter's number is the priority.
            if (pv.A == 64) {
                using I0 = std::integral_constant<int, 0>; using I16 = std::integral_constant<int, 16>;
                using I32 = std::integral_constant<int, 32>; using I48 = std::integral_constant<int, 48>;
                using I64 = std::integral_constant<int, 64>;
                __builtin_amdgcn_s_setprio(3); sweep_const(I0{}, I16{});
                __builtin_amdgcn_s_setprio(2); sweep_const(I16{}, I32{});
                __builtin_amdgcn_s_setprio(1); sweep_const(I32{}, I48{});
                __builtin_amdgcn_s_setprio(0); sweep_const(I48{}, I64{});
            } else {
                const int q1 = pv.A >> 2, q2 = pv.A >> 1, q3 = q1 + q2;
                __builtin_amdgcn_s_setprio(3); sweep(0, q1);
                __builtin_amdgcn_s_setprio(2); sweep(q1, q2);
                __builtin_amdgcn_s_setprio(1); sweep(q2, q3);
                __builtin_amdgcn_s_setprio(0); sweep(q3, pv.A);
            }
        }
        double dx = 0.0, dy = 0.0, vl = 0.0;
        if (lead >= 0) {
            dx = s_xy[base + lead].x - x0;
            dy = s_xy[base + lead].y - y0;
            vl = (double)s_v[base + lead];
        }
        act0_own[idx] = (float)idm_law(c, (double)fv, lead >= 0, dx, dy, vl);
        act1_own[idx] = 0.0f;
    }
    iv.leader[idx] = lead;
}

}  // namespace

hipError_t launch_idm(const PoolView& v, const IdmView& iv, const int32_t* forced_leader, float* act0_own, float* act1_own,
                      hipStream_t s) {
    int log2A = 0;
    while ((1 << log2A) < v.A) ++log2A;
    const int epb = kIdmBlock >> log2A;
    hipLaunchKernelGGL(idm_kernel, dim3((v.n_env + epb - 1) / epb), dim3(kIdmBlock), 0, s, v, iv, forced_leader, act0_own, act1_own, log2A);
    return hipGetLastError();
}

}  // namespace t2d
