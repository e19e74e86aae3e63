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
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

constexpr int kIdmBlock = 256;

// IDMController.step + _idm_acceleration for one participant (oracle t2do_idm_accel)
T2D_DEV double idm_law(const double* c, double v, bool has_lead, double dx, double dy, double v_lead) {
    const double des = c[T2D_IDM_DESIRED_SPEED], T = c[T2D_IDM_TIME_HEADWAY], s0 = c[T2D_IDM_MIN_SPACING];
    const double amax = c[T2D_IDM_MAX_ACCEL], b = c[T2D_IDM_COMF_DECEL], delta = c[T2D_IDM_DELTA];
    double a;
    if (!has_lead) {  // :75-85
        if (des > 0.0) a = amax * (1.0 - pow_det(v / des, delta));
        else a = v > 0.0 ? -b : 0.0;
    } else {  // :106-141
        const double dist = __builtin_sqrt(dx * dx + dy * dy);  // np.hypot
        const double dv = v_lead - v;
        double s_star = s0 + v * T + (v * dv) / (2.0 * __builtin_sqrt(amax * b));
        if (s0 > s_star) s_star = s0;  // max(s_star, min_spacing)
        if (dist > 0.0) {
            const double term = des > 0.0 ? pow_det(v / des, delta) : (v > 0.0 ? 1.0 : 0.0);
            const double q = s_star / dist;
            a = amax * (1.0 - term - q * q);
        } else {
            a = -b;
        }
    }
    return clipd(a, -b, amax);  // np.clip :90
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
    const uint32_t* a_ids = pv.ids;
    const float *a_x = pv.x, *a_y = pv.y, *a_h = pv.heading, *a_v = pv.speed;
    const uint8_t* a_ctrl = iv.ctrl_id;
    const double* a_rows = iv.rows;
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
    const double qnan = __builtin_nan("");
    s_xy[tid] = active ? make_double2((double)fx, (double)fy) : make_double2(qnan, qnan);
    s_v[tid] = fv;
    __syncthreads();
    if (!valid) return;
    int lead = -1;
    if (active && ctrl != T2D_IDM_NONE && ctrl < iv.n_ctrl) {
        const double* c = a_rows + (size_t)ctrl * T2D_IDM_COLS;
        const double hw = c[T2D_IDM_LANE_HALF_WIDTH], horizon = c[T2D_IDM_HORIZON];
        double sn, cs;
        sincos_det((double)fh, sn, cs);
        const int base = env_local << log2A;
        double best = __builtin_inf();
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
                    const bool ok = lon > 0.0 && lon <= horizon && __builtin_fabs(lat) <= hw;
                    const double key = ok ? lon : __builtin_inf();
                    if (key < best) {
                        best = key;
                        lead = j;
                    }
                }
            };
            // wave priority by progress (see the step kernel): the launch is one wave-round, a SIMD's waves should finish
            // together.  The sweep in quarters, the quarter's number is the priority.
            const int q1 = pv.A >> 2, q2 = pv.A >> 1, q3 = q1 + q2;
            __builtin_amdgcn_s_setprio(3); sweep(0, q1);
            __builtin_amdgcn_s_setprio(2); sweep(q1, q2);
            __builtin_amdgcn_s_setprio(1); sweep(q2, q3);
            __builtin_amdgcn_s_setprio(0); sweep(q3, pv.A);
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
