// t2d_integrate_dev.h -- per-participant integrators (device functions) shared by the stand-alone
// integrate kernel (t2d_integrate.hip) and the fused step kernel (t2d_collide.hip).
#pragma once
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

// The kernel's own argument block (PoolView is the first parameter of the step kernels, t2d_status_config the second),
// through an empty asm: loads of its fields through the returned pointer cannot be moved above this point (see
// collide_kernel: hoisted argument loads cost it 108 spilled scalars).
typedef const __attribute__((address_space(4))) PoolView* KernargView;
T2D_DEV KernargView late_args() {
    auto kp = (KernargView)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}
// byte offset of the second kernel argument (t2d_status_config) in the argument block: behind the PoolView
constexpr size_t kCfgArgOffset = (sizeof(PoolView) + alignof(t2d_status_config) - 1) / alignof(t2d_status_config) * alignof(t2d_status_config);
typedef const __attribute__((address_space(4))) t2d_status_config* KernargCfg;
T2D_DEV KernargCfg late_cfg() { return (KernargCfg)((const __attribute__((address_space(4))) char*)late_args() + kCfgArgOffset); }

namespace integ {

constexpr double kG = 9.81;  // PhysicsModelBase._G

struct StepOut {
    double x, y, heading, speed, vx, vy, app0, app1;
    bool has_velocity;
};

constexpr double kEpsMax = 0.1;    // largest sub-step angle the Taylor rotation accepts
constexpr double kEpsTiny = 0.01;  // below this, sin to e^3 / cos to e^4 are exact to < 1e-13

// rotate (c, s) by angle eps, |eps| <= kEpsMax: sin to e^7, cos to e^8 (truncation < 3e-15)
T2D_DEV void rotate_small(double eps, double& c, double& s) {
    double e2 = eps * eps;
    double ps = __builtin_fma(e2, -1.0 / 5040.0, 1.0 / 120.0);
    ps = __builtin_fma(e2, ps, -1.0 / 6.0);
    ps = __builtin_fma(e2, ps, 1.0);
    double se = eps * ps;
    double pc = __builtin_fma(e2, 1.0 / 40320.0, -1.0 / 720.0);
    pc = __builtin_fma(e2, pc, 1.0 / 24.0);
    pc = __builtin_fma(e2, pc, -0.5);
    double ce = __builtin_fma(e2, pc, 1.0);
    double cn = __builtin_fma(c, ce, -(s * se));
    double sn = __builtin_fma(s, ce, c * se);
    c = cn;
    s = sn;
}

// rotate (c, s) by a tiny angle |eps| <= kEpsTiny: sin = e - e^3/6 (err e^5/120 < 1e-12 relative to
// e), cos = 1 - e^2/2 + e^4/24 (err e^6/720 < 2e-15)
T2D_DEV void rotate_tiny(double eps, double& c, double& s) {
    double e2 = eps * eps;
    double se = eps * __builtin_fma(e2, -1.0 / 6.0, 1.0);
    double ce = __builtin_fma(e2, __builtin_fma(e2, 1.0 / 24.0, -0.5), 1.0);
    double cn = __builtin_fma(c, ce, -(s * se));
    double sn = __builtin_fma(s, ce, c * se);
    c = cn;
    s = sn;
}

// rotate (c, s) by |eps| <= 0.6: sin to e^13, cos to e^14 (truncation < 4e-16) -- the two rotations of a resummed step
T2D_DEV void rotate_mid(double eps, double& c, double& s) {
    const double e2 = eps * eps;
    double ps = __builtin_fma(e2, 1.0 / 6227020800.0, -1.0 / 39916800.0);
    ps = __builtin_fma(e2, ps, 1.0 / 362880.0);
    ps = __builtin_fma(e2, ps, -1.0 / 5040.0);
    ps = __builtin_fma(e2, ps, 1.0 / 120.0);
    ps = __builtin_fma(e2, ps, -1.0 / 6.0);
    ps = __builtin_fma(e2, ps, 1.0);
    const double se = eps * ps;
    double pc = __builtin_fma(e2, -1.0 / 87178291200.0, 1.0 / 479001600.0);
    pc = __builtin_fma(e2, pc, -1.0 / 3628800.0);
    pc = __builtin_fma(e2, pc, 1.0 / 40320.0);
    pc = __builtin_fma(e2, pc, -1.0 / 720.0);
    pc = __builtin_fma(e2, pc, 1.0 / 24.0);
    pc = __builtin_fma(e2, pc, -0.5);
    const double ce = __builtin_fma(e2, pc, 1.0);
    const double cn = __builtin_fma(c, ce, -(s * se));
    const double sn = __builtin_fma(s, ce, c * se);
    c = cn;
    s = sn;
}

// The resummed kinematic step.  With constant (clipped) acceleration and steering angle and the speed inside its bounds for
// the whole step -- the linear case below -- the reference's n Euler sub-steps (single_track_kinematics.py:149-160) are
//     x_n = x_0 + dt sum_k v_k cos(theta_k),   v_k = v_0 + k ah,   theta_k = theta_0 + k eps_0 + dlt k (k - 1) / 2
// (theta = phi + beta, eps_0 = v_0 kh, dlt = ah kh).  Centred on u = k - m, m = (n - 1) / 2, and scaled by M = n / 2:
//     theta_k = Theta + a w + b w^2,  v_k = V + (ah M) w,  w = u / M in (-1, 1),  a = (eps_0 + dlt (m - 1/2)) M,  b = dlt M^2 / 2
// so the sum is  n [ (cos, sin)(Theta) rotated by (P, Q) ],  P = V E[cos f] + ah M E[w cos f],  Q = V E[sin f] + ah M E[w sin f],
// f = a w + b w^2, E = the mean over the n sub-steps.  Expanding in a and b and dropping the odd moments of w (zero by
// symmetry) leaves eight polynomials in a^2 whose coefficients depend on n alone (PoolView::kin_coef, built by the host):
// one centre rotation + ~45 fma replace n x 11 operations.  |a| <= kResumAmax (half the heading change of the step),
// |b| <= kResumBmax: truncation a^10 / 10!, b^4 / 24 of the path -- < 2e-10 m; outside, the recurrence loop runs.
constexpr double kResumAmax = 0.5;
constexpr double kResumBmax = 5e-3;
struct ResumIn { double a, b, ahM, V; };
// fp64 operations with ONE operand in scalar registers, spelled out: the table's coefficients arrive by scalar loads from the
// argument block, and left to itself the compiler copies every one of them into vector registers first (two v_mov_b32 per
// coefficient: 64 moves around 32 fma -- the resummed step then issued as many instructions as the loop it replaces)
T2D_DEV double fma_vvs(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
T2D_DEV double mul_vs(double a, double c) {
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}
T2D_DEV double add_vs(double a, double c) {
    double r;
    asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}
T2D_DEV void resum_sums(const KernargView ka, const ResumIn& r, double& P, double& Q) {
    const double a2 = r.a * r.a, b2 = r.b * r.b;
    double q[8];
    // (the first Horner step has two scalar operands -- one more than an instruction may read: multiply, then add)
#pragma unroll
    for (int p = 0; p < 8; ++p) q[p] = add_vs(mul_vs(a2, ka->kin_coef[kKinDegree][p]), ka->kin_coef[kKinDegree - 1][p]);
#pragma unroll
    for (int i = kKinDegree - 2; i >= 0; --i) {
#pragma unroll
        for (int p = 0; p < 8; ++p) q[p] = fma_vvs(q[p], a2, ka->kin_coef[i][p]);
    }
    const double ec = __builtin_fma(b2, q[KIN_Q4], q[KIN_Q0]);
    const double es = r.b * __builtin_fma(b2, q[KIN_Q6], q[KIN_Q2]);
    const double ewc = (r.a * r.b) * __builtin_fma(b2, q[KIN_R8], q[KIN_R4]);
    const double ews = r.a * __builtin_fma(b2, q[KIN_R6], q[KIN_R2]);
    P = __builtin_fma(r.ahM, ewc, r.V * ec);
    Q = __builtin_fma(r.ahM, ews, r.V * es);
}

// 1/x to ~1 ulp: hardware estimate + two Newton steps (fast variant only; the exact variant
// uses IEEE division).  x must be normal and finite.
T2D_DEV double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
// 1/x with ONE Newton step (~2^-50): enough as the seed of div_r, whose residual correction
// squares the remaining error
T2D_DEV double rcp_nr1(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
// 1/sqrt(x) to ~1 ulp, x > 0 normal
T2D_DEV double rsq_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = y * __builtin_fma(-h, y * y, 1.5);
    return y * __builtin_fma(-h, y * y, 1.5);
}
// a / b given r ~ 1/b: one residual correction makes the quotient correctly rounded except in
// vanishingly rare cases (Markstein); b normal, quotient finite
T2D_DEV double div_r(double a, double b, double r) {
    double q = a * r;
    double rem = __builtin_fma(-q, b, a);
    return __builtin_fma(rem, r, q);
}

// RESUM: compile the resummed step in (the stand-alone integrator and the fused step of pools that fill the GPU; the looping
// forms of small pools and the single-ego kernel keep the recurrence loop alone: their lone waves are latency chains that a
// wave-uniform table fetch lengthens, and their register allocation is at its limit as it is -- measured, DESIGN.md 8.20;
// touching the table's cache lines early, in the start-up phase, cost every launch ~1 us: six more scalar-cache misses while
// all waves start together)
template <int VARIANT, bool RESUM = false, typename PF>
T2D_DEV StepOut step_kinematics(PF P, double x, double y, double phi, double v, double accel,
                                double delta, int interval) {
    int flags = (int)P(T2D_P_RANGE_FLAGS);
    if (flags & T2D_RANGE_ACCEL) accel = clipd(accel, P(T2D_P_ACCEL_LO), P(T2D_P_ACCEL_HI));
    if (flags & T2D_RANGE_STEER) delta = clipd(delta, P(T2D_P_STEER_LO), P(T2D_P_STEER_HI));
    const bool clip_v = flags & T2D_RANGE_SPEED;
    const double vlo = P(T2D_P_SPEED_LO), vhi = P(T2D_P_SPEED_HI);
    const double lr = P(T2D_P_LR), wb = P(T2D_P_WB);
    // the sub-step (double)delta_t / 1000 and the counts interval // delta_t, interval % delta_t depend on the TYPE and the
    // launch only: the host keeps them in two columns of the table (t2d_api.hip derive_kernel) -- per lane they were a
    // 32-bit integer division and an IEEE fp64 division, ~50 instructions per wave and step
    const double dt = P(T2D_P_DT_S);
    const int sub = (int)P(T2D_P_SUBSTEPS);
    const int n_steps = sub >> 16;
    const int rem = sub & 0xffff;
    const int total = n_steps + (rem > 0 ? 1 : 0);
    double ovx, ovy;  // (all StepOut fields are assembled in ONE place per model: stores into a
                      // struct from several branches get sunk into pointer-phis and cost scratch)
    if (VARIANT == 0) {
        const double tand = tan_det(delta);
        const double t = lr / wb * tand;
        const double beta = atan_det(t);
        double sb, cb;
        sincos_det(beta, sb, cb);
        for (int k = 0; k < total; ++k) {
            double h = k < n_steps ? dt : (double)rem / 1000;
            double sn, cs;
            sincos_det(phi + beta, sn, cs);
            double dx = v * cs;
            double dy = v * sn;
            double dphi = v / wb * tand * cb;
            x += dx * h;
            y += dy * h;
            phi += dphi * h;
            v += accel * h;
            if (clip_v) v = clipd(v, vlo, vhi);
        }
        double sp, cp;
        sincos_det(phi, sp, cp);
        ovx = v * cp;
        ovy = v * sp;
    } else {
        double sd, cd, sp, cp;
        sincos_det_steer_and(delta, phi, sd, cd, sp, cp);
        const double tw = sd * rcp_nr(cd * wb);  // tan(delta) / wb
        const double t = lr * tw;                // tan(beta)
        // cos(beta) = 1/sqrt(1+t^2), sin(beta) = t*cos(beta): no atan needed
        const double cb = rsq_nr(__builtin_fma(t, t, 1.0));
        const double sb = t * cb;
        double c = cp * cb - sp * sb;  // cos(phi + beta)
        double s = sp * cb + cp * sb;
        const double kk = tw * cb;  // d(phi)/dt = v * kk
        // (np.clip keeps a NaN speed NaN, min / max would hand back the bound: a clipped speed is NaN only when the step's
        // first sum v + accel h is -- a NaN state or action, inf - inf -- so the loop adds this: 0, or NaN for such a lane)
        const double kv_first = v + accel * dt;
        const double kv_poison = kv_first != kv_first ? kv_first : 0.0;
        auto sub_step = [&](double h, double ah, double kh) {
            const double eps = v * kh;  // d(phi) of this sub-step
            const double vh = v * h;
            x = __builtin_fma(vh, c, x);
            y = __builtin_fma(vh, s, y);
            phi += eps;
            v += ah;
            if (clip_v) v = __builtin_fmin(__builtin_fmax(v, vlo), vhi) + kv_poison;   // np.clip, NaN kept (kv_poison below)
            if (__builtin_fabs(eps) <= kEpsMax) {
                rotate_small(eps, c, s);
            } else {  // absurd yaw rates (unbounded speed): re-seed from phi
                double s2, c2;
                sincos_det(phi, s2, c2);
                c = c2 * cb - s2 * sb;
                s = s2 * cb + c2 * sb;
            }
        };
        const double ah = accel * dt, kh = kk * dt;
        // Main loop.  While the speed is not clipped it is linear in the sub-step index, so the
        // sub-step angle eps_k = v_k * kh grows by the constant dlt = ah * kh and (cos eps, sin eps)
        // is itself advanced by a fixed rotation (cD, sD) instead of being re-evaluated: 11 fp64
        // operations per sub-step instead of 21.  Decided once per step and per wave: if any lane
        // would clip its speed or leave the small-angle range during this step, the whole wave
        // takes the generic loop.
        // a lane sitting exactly on a speed bound with the acceleration pushing outwards stays there for the whole
        // step (clip(bound + ah) = bound): linear with zero effective acceleration.  (Cyclists braked to their lower
        // bound 0 would otherwise send their whole wave through the plain loop on every step.)
        const bool pinned = clip_v && ((v == vhi && ah >= 0.0) || (v == vlo && ah <= 0.0));
        const double ah_l = pinned ? 0.0 : ah;
        const double v_end = v + (double)n_steps * ah_l;
        const double eps0 = v * kh, eps_end = v_end * kh, dlt = ah_l * kh;
        const bool lane_linear = (!clip_v || (v >= vlo && v <= vhi && v_end >= vlo && v_end <= vhi)) &&
                                 __builtin_fabs(eps0) <= kEpsMax && __builtin_fabs(eps_end) <= kEpsMax;
        // the resummed step (see resum_sums): every lane of the wave linear, of the sub-step count the table was built for,
        // and inside the series' range
        bool wave_resum = false;
        KernargView ka = nullptr;
        double ra = 0.0, rb = 0.0;
        if constexpr (RESUM) {
            ka = late_args();
            const int kn = ka->kin_n;   // (0: no table -- pools too small for it, variant 2 -- and nothing else of it is fetched)
            if (kn > 0) {
                ra = __builtin_fma(dlt, ka->kin_geo[2], eps0) * ka->kin_geo[1];   // a = (eps_0 + dlt (m - 1/2)) M
                rb = dlt * ka->kin_geo[4];                                        // b = dlt M^2 / 2
                const bool lane_resum = lane_linear && n_steps == kn && __builtin_fabs(ra) <= kResumAmax && __builtin_fabs(rb) <= kResumBmax;
                wave_resum = __ballot(!lane_resum) == 0ull;
            }
        }
        if (RESUM && wave_resum) {
            const double gm = ka->kin_geo[0], gM = ka->kin_geo[1];
            double Ps, Qs;
            resum_sums(ka, ResumIn{ra, rb, ah_l * gM, __builtin_fma(gm, ah_l, v)}, Ps, Qs);
            rotate_mid(__builtin_fma(dlt, ka->kin_geo[3], gm * eps0), c, s);   // theta_0 -> Theta: m eps_0 + dlt m (m - 1) / 2
            const double fn = ka->kin_geo[7], ndt = fn * dt;
            x = __builtin_fma(ndt, __builtin_fma(c, Ps, -(s * Qs)), x);
            y = __builtin_fma(ndt, __builtin_fma(s, Ps, c * Qs), y);
            // Theta -> theta_n: A (m + 1) + B (m + 1)^2 with A = a / M, B = dlt / 2
            rotate_mid(__builtin_fma(dlt, ka->kin_geo[6], (__builtin_fma(dlt, ka->kin_geo[2], eps0)) * ka->kin_geo[5]), c, s);
            phi += __builtin_fma(fn, eps0, dlt * (0.5 * (fn * (fn - 1.0))));
            v = v_end;
        } else if (__ballot(!lane_linear) == 0ull) {
            double ce = 1.0, se = 0.0, cD = 1.0, sD = 0.0;
            const double amax = __builtin_fmax(__builtin_fabs(eps0), __builtin_fabs(eps_end));
            if (__ballot(amax > kEpsTiny) == 0ull) {  // the usual case: |sub-step angle| <= 0.01 rad in the whole wave
                rotate_tiny(eps0, ce, se);
                rotate_tiny(dlt, cD, sD);
            } else {
                rotate_small(eps0, ce, se);  // (cos eps_0, sin eps_0)
                rotate_small(dlt, cD, sD);   // |dlt| <= |eps_end - eps0| / n <= 2 kEpsMax / n
            }
            // v_k * dt advances by a constant, phi and v are the closed forms of their sums: 11 operations per sub-step
            double vh = v * dt;
            const double dvh = ah_l * dt;
#pragma unroll 4
            for (int k = 0; k < n_steps; ++k) {
                x = __builtin_fma(vh, c, x);
                y = __builtin_fma(vh, s, y);
                const double cn = __builtin_fma(c, ce, -(s * se));
                const double sn = __builtin_fma(s, ce, c * se);
                c = cn;
                s = sn;
                const double cen = __builtin_fma(ce, cD, -(se * sD));
                const double sen = __builtin_fma(se, cD, ce * sD);
                ce = cen;
                se = sen;
                vh += dvh;
            }
            const double fn = (double)n_steps;
            phi += __builtin_fma(fn, eps0, dlt * (0.5 * (fn * (fn - 1.0))));
            v = v_end;
        } else {
            // Lanes whose speed reaches a bound INSIDE the step (v + k ah crosses it at sub-step k*) are piecewise linear:
            // before k* as above, from k* on pinned at the bound (clip(bound + ah) = bound for the rest of the step).  The
            // recurrence loop handles them by switching the lane's increments at k* -- a wave-uniform test per sub-step,
            // the switch itself only in the few sub-steps in which some lane crosses.  (With one parking ego per env, speed
            // range +- 0.5 m/s and random accelerations, 7 % of the egos cross in a step and sent a quarter of the
            // single-ego kernel's waves -- and with them every launch -- through the plain loop.)
            const bool inside = clip_v && !pinned && v >= vlo && v <= vhi;
            const double v_unc = v + (double)n_steps * ah;
            const bool crossing = inside && (v_unc > vhi || v_unc < vlo);
            const double vB = v_unc > vhi ? vhi : vlo;
            int kstar = 0x7fffffff;
            if (crossing) {
                const double t = (vB - v) / ah;                      // > 0: updates until the bound is reached
                const double tc = __builtin_ceil(t);
                kstar = tc < 1.0 ? 1 : (tc > 1e6 ? 1000000 : (int)tc);
            }
            const double epsB = vB * kh;
            const bool lane_piecewise = (lane_linear || (crossing && __builtin_fabs(eps0) <= kEpsMax && __builtin_fabs(epsB) <= kEpsMax));
            if (__ballot(!lane_piecewise) == 0ull) {
                double ce = 1.0, se = 0.0, cD = 1.0, sD = 0.0, ceB = 1.0, seB = 0.0;
                const double dlt_p = crossing ? ah * kh : dlt;
                rotate_small(eps0, ce, se);
                rotate_small(dlt_p, cD, sD);
                rotate_small(epsB, ceB, seB);
                double vh = v * dt, dvh = (crossing ? ah : ah_l) * dt;
                const double vhB = vB * dt;
                for (int k = 0; k < n_steps; ++k) {
                    if (__ballot(k == kstar) != 0ull) {
                        const bool sw = k == kstar;
                        vh = sw ? vhB : vh;
                        dvh = sw ? 0.0 : dvh;
                        ce = sw ? ceB : ce;
                        se = sw ? seB : se;
                        cD = sw ? 1.0 : cD;
                        sD = sw ? 0.0 : sD;
                    }
                    x = __builtin_fma(vh, c, x);
                    y = __builtin_fma(vh, s, y);
                    const double cn = __builtin_fma(c, ce, -(s * se));
                    const double sn = __builtin_fma(s, ce, c * se);
                    c = cn;
                    s = sn;
                    const double cen = __builtin_fma(ce, cD, -(se * sD));
                    const double sen = __builtin_fma(se, cD, ce * sD);
                    ce = cen;
                    se = sen;
                    vh += dvh;
                }
                if (crossing) {
                    const int ks = kstar < n_steps ? kstar : n_steps;   // sub-steps taken with the unclipped, linear speed
                    const double fk = (double)ks;
                    phi += __builtin_fma(fk, eps0, dlt_p * (0.5 * (fk * (fk - 1.0)))) + (double)(n_steps - ks) * epsB;
                    v = vB;   // (k* <= n: the n-th update reaches or passes the bound and is clipped to it)
                } else {
                    const double fn = (double)n_steps;
                    phi += __builtin_fma(fn, eps0, dlt * (0.5 * (fn * (fn - 1.0))));
                    v = v_end;
                }
            } else {
                for (int k = 0; k < n_steps; ++k) sub_step(dt, ah, kh);
            }
        }
        if (rem > 0) {
            const double hr = (double)rem / 1000;
            sub_step(hr, accel * hr, kk * hr);
        }
        // cos(phi) = cos((phi+beta) - beta)
        ovx = v * (c * cb + s * sb);
        ovy = v * (s * cb - c * sb);
    }
    return StepOut{x, y, mod_two_pi(phi), v, ovx, ovy, accel, delta, true};
}

template <int VARIANT, typename PF>
T2D_DEV StepOut step_dynamics(PF P, double x, double y, double phi, double v, double accel,
                              double delta, int interval) {
    int flags = (int)P(T2D_P_RANGE_FLAGS);
    if (flags & T2D_RANGE_ACCEL) accel = clipd(accel, P(T2D_P_ACCEL_LO), P(T2D_P_ACCEL_HI));
    if (flags & T2D_RANGE_STEER) delta = clipd(delta, P(T2D_P_STEER_LO), P(T2D_P_STEER_HI));
    const bool clip_v = flags & T2D_RANGE_SPEED;
    const double vlo = P(T2D_P_SPEED_LO), vhi = P(T2D_P_SPEED_HI);
    const double lf = P(T2D_P_LF), lr = P(T2D_P_LR), wb = P(T2D_P_WB);
    const double mass = P(T2D_P_MASS), hcg = P(T2D_P_MASS_HEIGHT), mu = P(T2D_P_MU);
    const double Iz = P(T2D_P_IZ), cf = P(T2D_P_CF), cr = P(T2D_P_CR);
    const double dt = P(T2D_P_DT_S);                      // (double)delta_t / 1000, per type (see step_kinematics)
    const int n_steps = (int)P(T2D_P_SUBSTEPS) >> 16;     // interval // delta_t; the remainder is never integrated (:143)

    // the exact variant keeps the reference's IEEE divisions; the fast one multiplies by Newton reciprocals of the
    // per-type constants (wb, lf, Iz) -- ~1e-16 relative, far inside the contract wherever the model is well conditioned
    double factor_f, factor_r, mmi, tand, d_phi, beta;
    if (VARIANT == 0) {
        factor_f = (kG * lr - accel * hcg) / wb;
        factor_r = (kG * lf + accel * hcg) / wb;
        mmi = mu * mass / Iz;
        tand = tan_det(delta);
        d_phi = v / wb * tand;
        beta = atan_det(lr / lf * tand);
    } else {
        const double inv_wb = rcp_nr(wb);
        factor_f = (kG * lr - accel * hcg) * inv_wb;
        factor_r = (kG * lf + accel * hcg) * inv_wb;
        mmi = mu * mass * rcp_nr(Iz);
        double sd, cd;
        sincos_det_steer(delta, sd, cd);
        tand = sd * rcp_nr(cd);
        d_phi = v * inv_wb * tand;
        beta = atan_det(lr * rcp_nr(lf) * tand);
    }
    const double lf_cf_ff = lf * cf * factor_f;
    const double lr_cr_fr = lr * cr * factor_r;
    const double lf2_cf_ff = lf * lf * cf * factor_f;
    const double lr2_cr_fr = lr * lr * cr * factor_r;
    const double cf_ff = cf * factor_f;
    const double cr_fr = cr * factor_r;
    const double k21 = lr_cr_fr - lf_cf_ff;
    const double k34 = lf2_cf_ff + lr2_cr_fr;
    const double k65 = cr_fr + cf_ff;

    if (VARIANT == 0) {
        for (int k = 0; k < n_steps; ++k) {
            double s, c;
            sincos_det(phi + beta, s, c);
            double dx = v * c;
            double dy = v * s;
            double av = __builtin_fabs(v);
            double v_safe = av > 1e-6 ? v : (v >= 0 ? 1e-6 : -1e-6);
            double d_beta;
            if (av >= 0.1) {
                double dd_phi = mmi * (lf_cf_ff * delta + k21 * beta - k34 * d_phi / v_safe);
                d_beta = mu / v_safe * (cf_ff * delta - k65 * beta + k21 * d_phi / v_safe) - d_phi;
                d_phi += dd_phi * dt;
            } else {
                double tb = 1 + tand * lr / wb;
                double sd, cd;
                sincos_det(delta, sd, cd);
                d_beta = lr / (tb * tb) / wb / (cd * cd) * delta;
                double sbt, cbt;
                sincos_det(beta, sbt, cbt);
                d_phi += v * cbt / wb * tand * dt;
            }
            x += dx * dt;
            y += dy * dt;
            v += accel * dt;
            phi += d_phi * dt;
            beta += d_beta * dt;
            if (clip_v) v = clipd(v, vlo, vhi);
        }
    } else {
        // Same recurrences; the three divisions by v_safe share one Newton reciprocal with a
        // residual-corrected quotient, x / y use fma, cos/sin(phi + beta) a rotation recurrence.
        double s, c;
        sincos_det(phi + beta, s, c);
        const double c1 = lf_cf_ff * delta, c2 = cf_ff * delta, ah = accel * dt;
        // The speed moves monotonically between v0 and clamp(v0 + n*ah): when both ends are at least 0.1 m/s
        // (plus slack for the accumulated rounding) and of one sign for every lane of the wave, the |v| >= 0.1
        // branch is taken in every sub-step and the loop below runs without the test, the v_safe guard and the
        // low-speed code; constants are folded (mmi*dt) and phi / beta advance by fma.
        const double v_lin = __builtin_fma((double)n_steps, ah, v);
        const double v_fin = clip_v ? __builtin_fmin(__builtin_fmax(v_lin, vlo), vhi) : v_lin;
        const bool always_fast = (v >= 0.1000001 && v_fin >= 0.1000001) || (v <= -0.1000001 && v_fin <= -0.1000001);
        int k_done = 0;
        if (__ballot(!always_fast) == 0ull) {
            const double mmidt = mmi * dt;
            // unbounded speed = bounds at -inf / +inf: one min / max pair per sub-step, no per-lane select
            const double vlo_e = clip_v ? vlo : -__builtin_inf(), vhi_e = clip_v ? vhi : __builtin_inf();
            auto sub_step = [&]() {
                const double vh = v * dt;
                x = __builtin_fma(vh, c, x);
                y = __builtin_fma(vh, s, y);
                const double r = rcp_nr1(v);
                const double w = d_phi * r;
                const double in1 = __builtin_fma(-k34, w, __builtin_fma(k21, beta, c1));
                const double d_beta = __builtin_fma(mu * r, __builtin_fma(k21, w, __builtin_fma(-k65, beta, c2)), -d_phi);
                d_phi = __builtin_fma(mmidt, in1, d_phi);
                v = __builtin_fmin(__builtin_fmax(v + ah, vlo_e), vhi_e);
                phi = __builtin_fma(d_phi, dt, phi);
                beta = __builtin_fma(d_beta, dt, beta);
                const double eps = (d_phi + d_beta) * dt;
                const double aeps = __builtin_fabs(eps);
                // (one wave-uniform test per sub-step.  Round 4 tried the tiny-angle formula without asking, the largest sub-step
                // angle tested once per step and the step redone when it was too large: the eight start values kept alive for
                // the redo made the loop slower -- cfg3 fragments 8.4 -> 10.1 us per step, scripts/ab_step.py)
                if (__ballot(aeps > kEpsTiny) == 0ull) rotate_tiny(eps, c, s);
                else if (aeps <= kEpsMax) rotate_small(eps, c, s);
                else sincos_det(phi + beta, s, c);
            };
            // two sub-steps per trip: half the loop bookkeeping (the trip count is per lane: delta_t is a type parameter)
            int k = 0;
            for (; k + 2 <= n_steps; k += 2) {
                sub_step();
                sub_step();
            }
            if (k < n_steps) sub_step();
            k_done = n_steps;
        }
        // np.clip keeps a NaN speed NaN where min / max would hand back the bound.  A clipped speed can only be NaN when the
        // step's first sum v + ah is (a NaN state or action, inf - inf): then every sub-step's is -- so the loop clips with
        // min / max and adds this (0, or NaN for such a lane) instead of two compares and four selects per sub-step
        // (np.clip spelled out cost the highway pool's fragments 2 %: 8.44 -> 8.64 us per step, same-box A/B)
        const double v_first = v + ah;
        const double v_poison = v_first != v_first ? v_first : 0.0;
        for (int k = k_done; k < n_steps; ++k) {
            const double vh = v * dt;
            x = __builtin_fma(vh, c, x);
            y = __builtin_fma(vh, s, y);
            const double av = __builtin_fabs(v);
            const double v_safe = av > 1e-6 ? v : (v >= 0 ? 1e-6 : -1e-6);
            double d_beta;
            if (av >= 0.1) {
                // one Newton reciprocal (~1e-15) serves all three divisions by v_safe; fma freely:
                // ~1e-16 relative changes of the feedback terms are far inside the 1e-5 contract
                // wherever the reference itself is well conditioned (dynamics conditioning)
                const double r = rcp_nr1(v_safe);
                const double w = d_phi * r;
                const double dd_phi = mmi * __builtin_fma(-k34, w, __builtin_fma(k21, beta, c1));
                d_beta = __builtin_fma(mu * r, __builtin_fma(k21, w, __builtin_fma(-k65, beta, c2)), -d_phi);
                d_phi = __builtin_fma(dd_phi, dt, d_phi);
            } else {
                double tb = 1 + tand * lr / wb;
                double sd, cd;
                sincos_det(delta, sd, cd);
                d_beta = lr / (tb * tb) / wb / (cd * cd) * delta;
                double sbt, cbt;
                sincos_det(beta, sbt, cbt);
                d_phi += v * cbt / wb * tand * dt;
            }
            v += ah;
            const double e1 = d_phi * dt;
            const double e2 = d_beta * dt;
            phi += e1;
            beta += e2;
            if (clip_v) v = __builtin_fmin(__builtin_fmax(v, vlo), vhi) + v_poison;   // np.clip, NaN kept: see v_poison
            const double eps = e1 + e2;
            const double aeps = __builtin_fabs(eps);
            if (__ballot(aeps > kEpsTiny) == 0ull) rotate_tiny(eps, c, s);   // wave-uniform: usual case
            else if (aeps <= kEpsMax) rotate_small(eps, c, s);
            else sincos_det(phi + beta, s, c);
        }
    }
    // reference State has vx = vy = None (:220-227)
    return StepOut{x, y, mod_two_pi(phi), v, 0.0, 0.0, accel, delta, false};
}

template <typename PF>
T2D_DEV StepOut step_pointmass(PF P, double x, double y, double vx, double vy, double ax,
                               double ay, double dt /* (double)interval_ms / 1000, divided once by the host */) {
    int flags = (int)P(T2D_P_RANGE_FLAGS);
    const double lo = P(T2D_P_SPEED_LO), hi = P(T2D_P_SPEED_HI);
    double nvx = vx + ax * dt;
    double nvy = vy + ay * dt;
    double ns = __builtin_sqrt(nvx * nvx + nvy * nvy);
    double ox, oy, ovx, ovy;
    if (!(flags & T2D_RANGE_SPEED) || (lo <= ns && ns <= hi)) {
        ox = x + vx * dt + 0.5 * ax * (dt * dt);
        oy = y + vy * dt + 0.5 * ay * (dt * dt);
        ovx = nvx;
        ovy = nvy;
    } else {
        bool lower = ns < lo;
        double bound = lower ? lo : hi;
        double a_ = ax * ax + ay * ay;
        double b_ = 2 * (ax * vx + ay * vy);
        double c_ = vx * vx + vy * vy - bound * bound;
        double t1;
        if (__builtin_fabs(a_) < 1e-12) {
            t1 = __builtin_fabs(b_) < 1e-12 ? 0.0 : -c_ / b_;
        } else {
            double disc = b_ * b_ - 4 * a_ * c_;
            if (!(disc > 0.0)) disc = 0.0;
            double sq = __builtin_sqrt(disc);
            t1 = lower ? (-b_ - sq) / (2 * a_) : (-b_ + sq) / (2 * a_);
        }
        t1 = clipd(t1, 0.0, dt);
        double t2 = dt - t1;
        ovx = vx + ax * t1;
        ovy = vy + ay * t1;
        ox = x + vx * t1 + 0.5 * ax * (t1 * t1) + ovx * t2;
        oy = y + vy * t1 + 0.5 * ay * (t1 * t1) + ovy * t2;
    }
    return StepOut{ox, oy, atan2_det(ovy, ovx), __builtin_sqrt(ovx * ovx + ovy * ovy), ovx, ovy, ax, ay, true};
}


// one PhysicsModelBase.step for the participant held in registers
template <int VARIANT, bool RESUM = false, typename PF>
T2D_DEV StepOut step_participant(int model, PF P, double x, double y, double heading, double speed, double vx,
                                 double vy, double a0, double a1, int interval_ms, double interval_s) {
    if (model == T2D_MODEL_KINEMATICS) return step_kinematics<VARIANT, RESUM>(P, x, y, heading, speed, a0, a1, interval_ms);
    if (model == T2D_MODEL_DYNAMICS) return step_dynamics<VARIANT>(P, x, y, heading, speed, a0, a1, interval_ms);
    return step_pointmass(P, x, y, vx, vy, a0, a1, interval_s);
}

}  // namespace integ
}  // namespace t2d
