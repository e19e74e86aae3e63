// t2d_drift.hip -- SingleTrackDrift: dynamic single-track model with Pacejka tyres (scope row f4).
//
// Replaces (reference, tactics2d v0.1.9rc3), physics/single_track_drift.py:
//   Tire constants :16-49, _pure_slip_longitudinal_tire_forces :183-201, _pure_slip_lateral_tire_forces
//   :203-222, _combined_slip_longitudinal_tire_forces :224-250, _combined_slip_lateral_tire_forces :252-289,
//   _tire_forces :291-344, _step :346-465, step :467-503.
//
// Its own kernel: ~34 arctan + ~30 sin/cos per sub-step make this model ~50x the work of the others
// (~3.4 k fp64 VALU instructions per sub-step, 195 VGPRs with everything inlined so that the independent
// front / rear tyre chains interleave), so it stays out of the fused step kernel's register and code budget.  t2d_step / t2d_integrate launch it first when the parameter table holds a drift type; the
// other kernels pass drift lanes through.  One lane per participant, non-drift lanes exit at once.  All
// arithmetic is the oracle's (t2do_drift), operation by operation, deterministic trig, no contraction:
// state bit-identical to the oracle after the fp32 store.  The camber argument is the literal 0 at every
// call site of the reference, folded here exactly as in the oracle.  Extra state: wheel speeds
// T2D_F_OMEGA_F / T2D_F_OMEGA_R.  Only the reference's built-in Tire constants are supported.
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

constexpr int kDriftBlock = 64;

constexpr double TP_cx1 = 1.6411, TP_dx1 = 1.1739, TP_ex1 = 0.4640, TP_kx1 = 22.303, TP_hx1 = 1.2297e-3,
                 TP_vx1 = -8.8098e-6, TR_bx1 = 13.276, TR_bx2 = -13.778, TR_ex1 = 1.2568, TR_cx1 = 0.6522,
                 TR_hx1 = 5.0722e-3, TP_cy1 = 1.3507, TP_dy1 = 1.0489, TP_ey1 = -7.4722e-3, TP_ky1 = -21.920,
                 TR_by1 = 7.1433, TR_by2 = 9.1917, TR_by3 = -2.7856e-2, TR_cy1 = 1.0719, TR_ey1 = -0.2757,
                 TR_hy1 = 5.7448e-6, TR_vy1 = -2.7825e-2, TR_vy4 = 12.120, TR_vy5 = 1.9, TR_vy6 = -10.704;

T2D_DEV double sin_d(double x) {
    double s, c;
    sincos_det(x, s, c);
    return s;
}
T2D_DEV double cos_d(double x) {
    double s, c;
    sincos_det(x, s, c);
    return c;
}
T2D_DEV double atan_d(double x) { return atan_det(x); }

T2D_DEV double safe_den(double u) { return __builtin_fabs(u) > 1e-6 ? u : (u >= 0 ? 1e-6 : -1e-6); }

// C * atan(B*s - E*(B*s - atan(B*s))): the magic-formula angle
T2D_DEV double mf_angle(double B, double C, double E, double s) {
    const double bs = B * s;
    return C * atan_d(bs - E * (bs - atan_d(bs)));
}
T2D_DEV double pure_long(double kappa, double F_z) {
    const double S_vx = TP_vx1 * F_z;
    const double kappa_x = -kappa + TP_hx1;
    const double D_x = TP_dx1 * F_z;
    const double B_x = (TP_kx1 * F_z) / (TP_cx1 * D_x + 1e-6);
    return D_x * sin_d(mf_angle(B_x, TP_cx1, TP_ex1, kappa_x) + S_vx);
}
T2D_DEV double pure_lat(double alpha, double F_z) {
    const double alpha_y = alpha + 0.0;
    const double D_y = TP_dy1 * F_z;
    const double B_y = (TP_ky1 * F_z) / (TP_cy1 * D_y + 1e-6);
    return D_y * sin_d(mf_angle(B_y, TP_cy1, TP_ey1, alpha_y) + 0.0);
}
T2D_DEV double comb_long(double kappa, double alpha, double F0_x) {
    const double alpha_s = alpha + TR_hx1;
    const double B = TR_bx1 * cos_d(atan_d(TR_bx2 * kappa));
    const double D = F0_x / cos_d(mf_angle(B, TR_cx1, TR_ex1, TR_hx1));
    return D * cos_d(mf_angle(B, TR_cx1, TR_ex1, alpha_s));
}
T2D_DEV double comb_lat(double kappa, double alpha, double F_z, double F0_y) {
    const double kappa_s = kappa + TR_hy1;
    const double B = TR_by1 * cos_d(atan_d(TR_by2 * (alpha - TR_by3)));
    const double D = F0_y / cos_d(mf_angle(B, TR_cy1, TR_ey1, TR_hy1));
    const double D_vy = TP_dy1 * F_z * TR_vy1 * cos_d(atan_d(TR_vy4 * alpha));
    const double S_vy = D_vy * sin_d(TR_vy5 * atan_d(TR_vy6 * kappa));
    return D * cos_d(mf_angle(B, TR_cy1, TR_ey1, kappa_s)) + S_vy;
}

__global__ __launch_bounds__(kDriftBlock) void drift_kernel(PoolView pv, int interval_ms) {
    const int i = blockIdx.x * kDriftBlock + threadIdx.x;
    if (i >= pv.N) return;
    const uint32_t ids = pv.ids[i];
    const int model = (ids >> kIdsModelShift) & 0xff;
    if (!((ids >> kIdsActiveShift) & 0xffu) || model < T2D_MODEL_DRIFT) return;
    const int type = (ids >> kIdsTypeShift) & 0xff;
    auto P = [&](int col) -> double { return pv.params[col * T2D_MAX_TYPES + type]; };
    if (model == T2D_MODEL_POINTMASS_EULER) {
        // PointMass._step_euler (physics/point_mass.py:177-207): per sub-step v += a h; speed clipped -- and, when the clip
        // moved it by more than 1e-12, the velocity re-projected onto the heading of the PREVIOUS sub-step (:195-197) --;
        // position += v h; heading = atan2(vy, vx).  interval // delta_t sub-steps of delta_t and one of the remainder.
        const bool idm = pv.idm_ctrl && pv.idm_ctrl[i] != T2D_IDM_NONE;
        const double ax = (double)(idm ? pv.own_act0[i] : pv.act0[(size_t)i * pv.act_stride]),
                     ay = (double)(idm ? pv.own_act1[i] : pv.act1[(size_t)i * pv.act_stride]);
        double x = (double)pv.x[i], y = (double)pv.y[i], heading = (double)pv.heading[i];
        double vx = (double)pv.vx[i], vy = (double)pv.vy[i];
        const bool clip_s = (int)P(T2D_P_RANGE_FLAGS) & T2D_RANGE_SPEED;
        const double lo = P(T2D_P_SPEED_LO), hi = P(T2D_P_SPEED_HI);
        const int delta_t = (int)P(T2D_P_DELTA_T_MS);
        const int n_sub = interval_ms / delta_t, rem = interval_ms % delta_t;
        for (int k = 0; k <= n_sub; ++k) {
            double h = (double)delta_t / 1000;
            if (k == n_sub) {
                if (rem <= 0) break;
                h = (double)rem / 1000;
            }
            vx += ax * h;
            vy += ay * h;
            const double speed = __builtin_sqrt(vx * vx + vy * vy);   // np.linalg.norm([vx, vy])
            const double sc = clip_s ? clipd(speed, lo, hi) : speed;
            if (__builtin_fabs(speed - sc) > 1e-12) {
                double sn, cs;
                sincos_det(heading, sn, cs);
                vx = sc * cs;
                vy = sc * sn;
            }
            x += vx * h;
            y += vy * h;
            heading = atan2_det(vy, vx);
        }
        pv.x[i] = (float)x;
        pv.y[i] = (float)y;
        pv.heading[i] = (float)heading;
        pv.speed[i] = (float)__builtin_sqrt(vx * vx + vy * vy);   // State.speed, lazily ||(vx, vy)|| (state.py:135-150)
        pv.vx[i] = (float)vx;
        pv.vy[i] = (float)vy;
        if (pv.out_mask & T2D_OUT_APPLIED) {
            pv.applied0[i] = (float)ax;
            pv.applied1[i] = (float)ay;
        }
        return;
    }
    double x = (double)pv.x[i], y = (double)pv.y[i], phi = (double)pv.heading[i], v = (double)pv.speed[i];
    double omega_wf = (double)pv.omega_f[i], omega_wr = (double)pv.omega_r[i];
    const bool idm_lane = pv.idm_ctrl && pv.idm_ctrl[i] != T2D_IDM_NONE;   // IDM lane while caller actions are bound
    double accel = (double)(idm_lane ? pv.own_act0[i] : pv.act0[(size_t)i * pv.act_stride]),
           delta = (double)(idm_lane ? pv.own_act1[i] : pv.act1[(size_t)i * pv.act_stride]);
    const int flags = (int)P(T2D_P_RANGE_FLAGS);
    if (flags & T2D_RANGE_ACCEL) accel = clipd(accel, P(T2D_P_ACCEL_LO), P(T2D_P_ACCEL_HI));
    if (flags & T2D_RANGE_STEER) delta = clipd(delta, P(T2D_P_STEER_LO), P(T2D_P_STEER_HI));
    const bool clip_v = flags & T2D_RANGE_SPEED;
    const double vlo = P(T2D_P_SPEED_LO), vhi = P(T2D_P_SPEED_HI);
    const double lf = P(T2D_P_LF), lr = P(T2D_P_LR), wb = P(T2D_P_WB), mass = P(T2D_P_MASS), Iz = P(T2D_P_IZ);
    const double radius = P(T2D_P_DRIFT_RADIUS), Tsb = P(T2D_P_DRIFT_TSB), Tse = P(T2D_P_DRIFT_TSE),
                 Iyw = P(T2D_P_DRIFT_IYW);
    const int delta_t = (int)P(T2D_P_DELTA_T_MS);
    const int n_steps = interval_ms / delta_t, rem = interval_ms % delta_t;
    double sin_dl, cos_dl;
    sincos_det(delta, sin_dl, cos_dl);
    const double tan_dl = sin_dl / cos_dl;
    double d_phi = v / wb * tan_dl;
    double beta = atan_d(lr / lf * tan_dl);
    double T_B, T_E;
    if (accel > 0) {
        T_B = 0;
        T_E = mass * radius * accel;
    } else {
        T_B = mass * radius * accel;
        T_E = 0;
    }
    const double F_zf = (mass * 9.81 * lr) / wb, F_zr = (mass * 9.81 * lf) / wb;
    const int total = n_steps + (rem > 0 ? 1 : 0);
    for (int k = 0; k < total; ++k) {
        const double dt = k < n_steps ? (double)delta_t / 1000 : (double)rem / 1000;
        const double v_safe = safe_den(v);
        double sin_b, cos_b;
        sincos_det(beta, sin_b, cos_b);
        const double cos_b_safe = safe_den(cos_b);
        const double alpha_f = atan_d((v_safe * sin_b + d_phi * lf) / (v_safe * cos_b_safe)) - delta;
        const double alpha_r = atan_d((v_safe * sin_b - d_phi * lr) / (v_safe * cos_b_safe));
        const double u_wf = v_safe * cos_b_safe * cos_dl + (v_safe * sin_b + lf * d_phi) * sin_dl;
        const double u_wr = v_safe * cos_b_safe;
        const double s_f = 1 - radius * omega_wf / safe_den(u_wf);
        const double s_r = 1 - radius * omega_wr / safe_den(u_wr);
        const double F0_xf = pure_long(s_f, F_zf), F0_xr = pure_long(s_r, F_zr);
        const double F0_yf = pure_lat(alpha_f, F_zf), F0_yr = pure_lat(alpha_r, F_zr);
        const double F_lf = comb_long(s_f, alpha_f, F0_xf), F_lr = comb_long(s_r, alpha_r, F0_xr);
        const double F_sf = comb_lat(s_f, alpha_f, F_zf, F0_yf), F_sr = comb_lat(s_r, alpha_r, F_zr, F0_yr);
        double sn, cs;
        sincos_det(phi + beta, sn, cs);
        const double dx = v * cs, dy = v * sn;
        double dv, d_beta, d_owf, d_owr;
        if (__builtin_fabs(v) >= 0.1) {
            double sdb, cdb;
            sincos_det(delta - beta, sdb, cdb);
            dv = 1 / mass * (-F_sf * sdb + F_sr * sin_b + F_lr * cos_b + F_lf * cdb);
            d_beta = -d_phi + 1 / (mass * v_safe) * (F_sf * cdb + F_sr * cos_b - F_lr * sin_b + F_lf * sdb);
            const double dd_phi = 1 / Iz * (F_sf * cos_dl * lf - F_sr * lr + F_lf * sin_dl * lf);
            d_phi += dd_phi * dt;
            d_owf = 1 / Iyw * (-radius * F_lf + Tsb * T_B + Tse * T_E);
            d_owr = 1 / Iyw * (-radius * F_lr + (1 - Tsb) * T_B + (1 - Tse) * T_E);
        } else {
            const double tb = 1 + tan_dl * lr / wb;
            dv = accel;
            d_beta = lr / (tb * tb) / wb / (cos_dl * cos_dl) * delta;
            d_phi += v * cos_b / wb * tan_dl * dt;
            d_owf = 1 / (cos_dl * radius) * (accel * cos_b - v * sin_b * d_beta + v * cos_b * tan_dl * delta);
            d_owr = 1 / radius * (accel * cos_b - v * sin_b * d_beta);
        }
        x += dx * dt;
        y += dy * dt;
        v += dv * dt;
        phi += d_phi * dt;
        beta += d_beta * dt;
        omega_wf += d_owf * dt;
        omega_wr += d_owr * dt;
        if (clip_v) v = clipd(v, vlo, vhi);
    }
    pv.x[i] = (float)x;
    pv.y[i] = (float)y;
    pv.heading[i] = (float)mod_two_pi(phi);
    pv.speed[i] = (float)v;
    pv.omega_f[i] = (float)omega_wf;
    pv.omega_r[i] = (float)omega_wr;
    if (pv.out_mask & T2D_OUT_APPLIED) {
        pv.applied0[i] = (float)accel;
        pv.applied1[i] = (float)delta;
    }
}

}  // namespace

hipError_t launch_drift(const PoolView& v, int interval_ms, hipStream_t s) {
    hipLaunchKernelGGL(drift_kernel, dim3((v.N + kDriftBlock - 1) / kDriftBlock), dim3(kDriftBlock), 0, s, v, interval_ms);
    return hipGetLastError();
}

}  // namespace t2d
