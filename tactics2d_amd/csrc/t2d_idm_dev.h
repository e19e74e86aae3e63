// t2d_idm_dev.h -- the per-participant part of the IDM agents (t2d_idm.hip): the controller's row, the build-defined leader
// rule over an env's participants staged in LDS, and the IDM law.  Device functions shared by idm_kernel and by the
// integrator waves of the fused step kernel's PIPE form (t2d_collide.hip), which run the controller ahead of each step
// themselves when a pool with installed controllers is stepped through t2d_step_n.
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   IDMController.step                controller/idm_controller.py:59-93
//   IDMController._idm_acceleration   controller/idm_controller.py:95-141
#pragma once
#include <type_traits>

#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {
namespace idm {

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


T2D_DEV IdmRow load_row(const T2D_GLOBAL double* r) {
    IdmRow c;
    c.des = r[T2D_IDM_DESIRED_SPEED]; c.T = r[T2D_IDM_TIME_HEADWAY]; c.s0 = r[T2D_IDM_MIN_SPACING];
    c.amax = r[T2D_IDM_MAX_ACCEL]; c.b = r[T2D_IDM_COMF_DECEL]; c.delta = r[T2D_IDM_DELTA];
    c.hw = r[T2D_IDM_LANE_HALF_WIDTH]; c.horizon = r[T2D_IDM_HORIZON];
    return c;
}

// The leader rule for one controlled participant: among slots [0, A) of its env -- get(j) = the (x, y) of slot j, NaN for
// inactive slots -- those ahead (0 < lon <= horizon along the own heading, sn / cs its sine / cosine) inside the own corridor
// (|lat| <= hw), the one with the smallest lon, lowest index on ties.  PRIO: the sweep in quarters with the wave priority
// falling 3 -> 0 (idm_kernel: the launch is one wave-round, a SIMD's waves should finish together).
template <bool PRIO, class Get>
T2D_DEV int find_leader(Get get, int A, const IdmRow& c, double x0, double y0, double sn, double cs) {
    const double hw = c.hw;
    // `lon <= horizon` rides on the running minimum: it starts at the first double above the horizon and a candidate
    // must be strictly below it -- one compare and two selects fewer per candidate than testing the horizon apart
    double best = just_above(c.horizon);
    int lead = -1;
    auto sweep = [&](int j0, int j1) {
#pragma unroll 4
        for (int j = j0; j < j1; ++j) {
            const double2 q = get(j);
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
            const double2 q = get(j);
            const double dx = q.x - x0, dy = q.y - y0;
            const double lon = __builtin_fma(dx, cs, dy * sn);
            const double lat = __builtin_fma(dy, cs, -(dx * sn));
            const bool take = lon > 0.0 && lon < best && __builtin_fabs(lat) <= hw;
            best = take ? lon : best;
            lead = take ? j : lead;
        }
    };
    if (A == 64) {
        using I0 = std::integral_constant<int, 0>; using I16 = std::integral_constant<int, 16>;
        using I32 = std::integral_constant<int, 32>; using I48 = std::integral_constant<int, 48>;
        using I64 = std::integral_constant<int, 64>;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        sweep_const(I0{}, I16{});
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        sweep_const(I16{}, I32{});
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        sweep_const(I32{}, I48{});
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        sweep_const(I48{}, I64{});
    } else {
        const int q1 = A >> 2, q2 = A >> 1, q3 = q1 + q2;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        sweep(0, q1);
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        sweep(q1, q2);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        sweep(q2, q3);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        sweep(q3, A);
    }
    return lead;
}

}  // namespace idm
}  // namespace t2d
