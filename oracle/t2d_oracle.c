/*
 * t2d_oracle.c -- CPU restatement (fp64, scalar) of the tactics2d hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (tactics2d_amd/, libt2d_hip.so)
 * may import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity status
 *   physics  : PINNED.  Checked against golden vectors produced by importing the
 *              reference (`oracle/gen_golden.py` -> tests/golden/ npz + physics_kats.json);
 *              see tests/test_oracle_physics.py.
 *   geometry : PARITY UNPINNED against the reference's engine.  The reference delegates
 *              collision / boundary predicates to shapely>=2.0.7,<2.1.0 (GEOS), which is not
 *              in /root/reference, not installed, and whose results no reference test pins.
 *              The predicates below restate shapely's documented *semantics* (closed-set
 *              `intersects`, `contains`) on the reference's own vertex construction, and are
 *              pinned by hand-built KATs (tests/golden/geometry_kats.json), an independent
 *              cross-check against matplotlib.path and, for `intersects` and the IoU, exact rational
 *              arithmetic on the same binary64 inputs (tests/test_oracle_geometry.py,
 *              tests/test_iou_events.py) -- the mathematical values GEOS evaluates robustly.
 *   IDM controller, verify_state, SingleTrackDrift : PINNED by golden vectors produced by running the
 *              reference (oracle/gen_golden_idm.py, gen_golden_verify.py, gen_golden_drift.py).
 *   ParkingLotGenerator : the restatement is PINNED by replay (round 6).  t2do_generate_parking restates distributions, draw
 *              order, control flow and predicate semantics on a counter stream of its own, so its scenes are not numpy's;
 *              oracle/gen_golden_generator.py executes the reference's own class on 240 seeds of numpy's stream with every
 *              draw recorded (exact stand-ins for its shapely calls), and t2do_generate_parking_replay, fed the same draws,
 *              consumes them in the same order and yields the same scenes (tests/test_generator.py).  Property tests beside it.
 *   lidar    : PINNED (round 6).  The reference module imports shapely (cannot be imported here), but its numeric code needs
 *              none of it: oracle/gen_golden_lidar.py executes _rotate_and_filter_obstacles (lidar.py:97-126) and the
 *              statements of _scan_obstacles from the beam table on (:160-221) where they lie -> tests/golden/lidar.npz;
 *              oracle/lidar_ref.py equals them bit for bit, t2do_lidar to fp32 rounding with the same hits (tests/test_lidar.py).
 *   status / reward epilogue : PINNED (round 6).  oracle/gen_golden_status.py executes ParkingEnv.step, _get_reward,
 *              _get_relative_pose, _ParkingScenarioManager.check_status and the TimeExceed / NoAction / Arrival classes
 *              where they lie on 48 scripted episodes (the IoUs their detectors see are t2do_quad_iou's for the same
 *              quads: geometry stays unpinned) -> tests/golden/status_epilogue.npz; t2do_status_ex reproduces every
 *              status, flag and reward (tests/test_iou_events.py).
 *   parameter tables, Vehicle.load_from_template / get_pose, Map.boundary : PINNED (round 6) by loading / executing the
 *              reference's definitions (oracle/gen_golden_tables.py; tests/test_host.py, tests/test_mapgeom.py).
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * tactics2d repo root).  Arithmetic is IEEE fp64, evaluated left-to-right exactly as the
 * cited Python expression associates; compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/t2d.h"

#define TWO_PI (2.0 * 3.141592653589793)
#define G_ACC 9.81 /* PhysicsModelBase._G, physics/physics_model_base.py */

/* ====================================================================================
 * Deterministic sin/cos ("t2d_sincos", DESIGN.md section "Deterministic trig").
 * Used for pose construction so that event flags are reproducible bit-for-bit on any
 * IEEE-754 machine; agrees with libm to <= 2 ulp (tests/test_oracle_geometry.py).
 * Independent restatement of the spec: Cody-Waite 3-term reduction by pi/2 with fused
 * multiply-adds, then the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4].
 * ================================================================================== */
static const double PIO2_HI = 1.5707963267948966;      /* fl(pi/2)                 */
static const double PIO2_MID = 6.123233995736766e-17;  /* fl(pi/2 - PIO2_HI)       */
static const double PIO2_LO = -1.4973849048591698e-33; /* fl(pi/2 - hi - mid)      */
static const double TWO_OVER_PI = 0.6366197723675814;

void t2do_sincos(double x, double* s_out, double* c_out) {
    double k = rint(x * TWO_OVER_PI);
    double r = fma(-k, PIO2_HI, x);
    r = fma(-k, PIO2_MID, r);
    r = fma(-k, PIO2_LO, r);
    double z = r * r;
    /* sin kernel */
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    double sr = fma(r * z, ps, r);
    /* cos kernel */
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    long long q = (long long)k;
    switch (q & 3) {
        case 0: *s_out = sr; *c_out = cr; break;
        case 1: *s_out = cr; *c_out = -sr; break;
        case 2: *s_out = -sr; *c_out = -cr; break;
        default: *s_out = -cr; *c_out = sr; break;
    }
}


/* Deterministic atan ("t2d_atan"): classic 4-breakpoint argument reduction
 * (7/16, 11/16, 19/16, 39/16) + odd degree-23 polynomial split into even / odd halves.
 * IEEE +,-,*,/ only (no fma), so any conforming machine reproduces it bit-for-bit.      */
double t2do_atan(double x) {
    static const double hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    static const double lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    static const double aT[11] = {3.33333333333329318027e-01, -1.99999999998764832476e-01,
                                  1.42857142725034663711e-01, -1.11111104054623557880e-01,
                                  9.09088713343650656196e-02, -7.69187620504482999495e-02,
                                  6.66107313738753120669e-02, -5.83357013379057348645e-02,
                                  4.97687799461593236017e-02, -3.65315727442169155270e-02,
                                  1.62858201153657823623e-02};
    int neg = x < 0.0;
    double ax = fabs(x);
    int id;
    if (ax != ax) return x;
    if (ax >= 1.8014398509481984e16) return neg ? -(hi[3] + lo[3]) : (hi[3] + lo[3]);
    if (ax < 0.4375) {
        if (ax < 7.450580596923828e-09) return x; /* 2^-27 */
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else { id = 1; ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else { id = 3; ax = -1.0 / ax; }
    }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    double r;
    if (id < 0) r = ax - ax * (s1 + s2);
    else r = hi[id] - ((ax * (s1 + s2) - lo[id]) - ax);
    return neg ? -r : r;
}


/* atan2 on top of t2do_atan (deterministic spec; atan2(0, 0) = 0 like numpy) */
double t2do_atan2(double y, double x) {
    const double pi = 3.141592653589793, pio2 = 1.5707963267948966;
    if (x != x || y != y) return x + y;
    if (y == 0.0) {
        if (x > 0.0 || (x == 0.0 && !signbit(x))) return y;
        return signbit(y) ? -pi : pi;
    }
    if (x == 0.0) return y > 0.0 ? pio2 : -pio2;
    double a = t2do_atan(y / x);
    if (x > 0.0) return a;
    return y > 0.0 ? a + pi : a - pi;
}

/* trig dispatch: mode 0 = libm (what the reference's numpy calls resolve to, up to the
 * last ulp), mode 1 = deterministic (bit-reproducible; the GPU "exact" variant uses it). */
static int g_trig = 0;
void t2do_set_trig(int mode) { g_trig = mode; }

/* Batch entry points below loop over independent participants / envs; the all-cores CPU baseline
 * of bench.py spreads those loops over host threads (OpenMP, when built with -fopenmp).  Results
 * do not depend on the thread count.  Default 1 thread. */
static int g_threads = 1;
void t2do_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int t2do_get_threads(void) { return g_threads; }
int t2do_has_openmp(void) {
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}
static double T_sin(double x) { if (!g_trig) return sin(x); double s, c; t2do_sincos(x, &s, &c); return s; }
static double T_cos(double x) { if (!g_trig) return cos(x); double s, c; t2do_sincos(x, &s, &c); return c; }
static double T_tan(double x) { if (!g_trig) return tan(x); double s, c; t2do_sincos(x, &s, &c); return s / c; }
static double T_atan(double x) { return g_trig ? t2do_atan(x) : atan(x); }
static double T_atan2(double y, double x) { return g_trig ? t2do_atan2(y, x) : atan2(y, x); }

/* ====================================================================================
 * Physics
 * ================================================================================== */
static double clip(double v, double lo, double hi) { /* np.clip = min(max(v, lo), hi) */
    double t = v < lo ? lo : v;
    return t > hi ? hi : t;
}

/* np.mod(a, b) for b > 0: fmod, then shift negatives up (numpy npy_divmod semantics).
 * np.mod(-1e-17, 2*pi) == 2*pi exactly (SURVEY.md finding 10).                         */
static double np_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* SingleTrackKinematics.step + _step: physics/single_track_kinematics.py:126-198.
 * out: x, y, heading, speed, vx, vy, applied_accel, applied_steer                      */
void t2do_kinematics(const double* p, double x, double y, double phi, double v, double accel,
                     double delta, int interval, double* out) {
    int flags = (int)p[T2D_P_RANGE_FLAGS];
    if (flags & T2D_RANGE_ACCEL) accel = clip(accel, p[T2D_P_ACCEL_LO], p[T2D_P_ACCEL_HI]); /* :192 */
    if (flags & T2D_RANGE_STEER) delta = clip(delta, p[T2D_P_STEER_LO], p[T2D_P_STEER_HI]); /* :193 */
    double lr = p[T2D_P_LR], wb = p[T2D_P_WB];
    int delta_t = (int)p[T2D_P_DELTA_T_MS];
    double beta = T_atan(lr / wb * T_tan(delta)); /* :127 */
    double dt = (double)delta_t / 1000;       /* :128 */
    int n_steps = interval / delta_t;         /* :129 */
    int remainder = interval % delta_t;       /* :130 */
    for (int i = 0; i <= n_steps; ++i) {
        double h = dt;
        if (i == n_steps) { /* :151-163 remainder step */
            if (remainder <= 0) break;
            h = (double)remainder / 1000;
        }
        double dx = v * T_cos(phi + beta);                  /* :138 */
        double dy = v * T_sin(phi + beta);                  /* :139 */
        double dphi = v / wb * T_tan(delta) * T_cos(beta);    /* :141 */
        x += dx * h;                                      /* :143 */
        y += dy * h;
        phi += dphi * h;
        v += accel * h;
        if (flags & T2D_RANGE_SPEED) v = clip(v, p[T2D_P_SPEED_LO], p[T2D_P_SPEED_HI]); /* :148 */
    }
    out[0] = x;
    out[1] = y;
    out[2] = np_mod(phi, TWO_PI); /* :169 */
    out[3] = v;
    out[4] = v * T_cos(phi);        /* :170  un-wrapped phi */
    out[5] = v * T_sin(phi);
    out[6] = accel;
    out[7] = delta;
}

/* SingleTrackDynamics.step + _step: physics/single_track_dynamics.py:140-251.
 * vx, vy are not produced by the reference (State(..., speed=v), :220-227) -> NaN.      */
void t2do_dynamics(const double* p, double x, double y, double phi, double v, double accel,
                   double delta, int interval, double* out) {
    int flags = (int)p[T2D_P_RANGE_FLAGS];
    if (flags & T2D_RANGE_ACCEL) accel = clip(accel, p[T2D_P_ACCEL_LO], p[T2D_P_ACCEL_HI]); /* :245 */
    if (flags & T2D_RANGE_STEER) delta = clip(delta, p[T2D_P_STEER_LO], p[T2D_P_STEER_HI]); /* :246 */
    double lf = p[T2D_P_LF], lr = p[T2D_P_LR], wb = p[T2D_P_WB];
    double mass = p[T2D_P_MASS], hcg = p[T2D_P_MASS_HEIGHT], mu = p[T2D_P_MU];
    double Iz = p[T2D_P_IZ], cf = p[T2D_P_CF], cr = p[T2D_P_CR];
    int delta_t = (int)p[T2D_P_DELTA_T_MS];
    double dt = (double)delta_t / 1000; /* :141 */
    int n_steps = interval / delta_t;   /* :142 ; remainder (:143) is never integrated */

    double factor_f = (G_ACC * lr - accel * hcg) / wb; /* :145 */
    double factor_r = (G_ACC * lf + accel * hcg) / wb; /* :146 */
    double lf_cf_ff = lf * cf * factor_f;              /* :149 */
    double lr_cr_fr = lr * cr * factor_r;
    double lf2_cf_ff = lf * lf * cf * factor_f;        /* lf**2 * cf * factor_f */
    double lr2_cr_fr = lr * lr * cr * factor_r;
    double cf_ff = cf * factor_f;
    double cr_fr = cr * factor_r;

    double d_phi = v / wb * T_tan(delta);        /* :159 */
    double beta = T_atan(lr / lf * T_tan(delta));  /* :160 */

    for (int i = 0; i < n_steps; ++i) {
        double dx = v * T_cos(phi + beta); /* :164 */
        double dy = v * T_sin(phi + beta);
        double v_safe = fabs(v) > 1e-6 ? v : (v >= 0 ? 1e-6 : -1e-6); /* :169 */
        double d_beta;
        if (fabs(v) >= 0.1) { /* :171 */
            double dd_phi = mu * mass / Iz *
                            (lf_cf_ff * delta + (lr_cr_fr - lf_cf_ff) * beta -
                             (lf2_cf_ff + lr2_cr_fr) * d_phi / v_safe); /* :172-181 */
            d_beta = mu / v_safe *
                         (cf_ff * delta - (cr_fr + cf_ff) * beta +
                          (lr_cr_fr - lf_cf_ff) * d_phi / v_safe) -
                     d_phi; /* :182-191 */
            d_phi += dd_phi * dt; /* :192 */
        } else {
            double tb = 1 + T_tan(delta) * lr / wb;
            double cd = T_cos(delta);
            d_beta = lr / (tb * tb) / wb / (cd * cd) * delta;            /* :194-200 */
            d_phi += v * T_cos(beta) / wb * T_tan(delta) * dt;               /* :210 */
        }
        x += dx * dt; /* :212-216 */
        y += dy * dt;
        v += accel * dt;
        phi += d_phi * dt;
        beta += d_beta * dt;
        if (flags & T2D_RANGE_SPEED) v = clip(v, p[T2D_P_SPEED_LO], p[T2D_P_SPEED_HI]); /* :218 */
    }
    out[0] = x;
    out[1] = y;
    out[2] = np_mod(phi, TWO_PI);
    out[3] = v;
    out[4] = NAN;
    out[5] = NAN;
    out[6] = accel;
    out[7] = delta;
}

/* PointMass.step + _step_newton: physics/point_mass.py:83-175, 209-232.
 * The accel clip at :222-225 is dead code (its result is unused) and is not restated.
 * out: x, y, heading, speed (= ||(vx,vy)||, State.speed lazily), vx, vy, ax, ay         */
void t2do_pointmass(const double* p, double x, double y, double vx, double vy, double ax,
                    double ay, int interval, double* out) {
    int flags = (int)p[T2D_P_RANGE_FLAGS];
    double lo = p[T2D_P_SPEED_LO], hi = p[T2D_P_SPEED_HI];
    double dt = (double)interval / 1000; /* :86 */
    double nvx = vx + ax * dt;           /* :88 */
    double nvy = vy + ay * dt;
    double ns = sqrt(nvx * nvx + nvy * nvy); /* np.linalg.norm :90 */
    double ox, oy, ovx, ovy;
    if (!(flags & T2D_RANGE_SPEED) || (lo <= ns && ns <= hi)) { /* :93 */
        ox = x + vx * dt + 0.5 * ax * (dt * dt);                 /* :96  dt**2 */
        oy = y + vy * dt + 0.5 * ay * (dt * dt);
        ovx = nvx;
        ovy = nvy;
    } else {
        int lower = ns < lo;                         /* :105 vs :139 */
        double bound = lower ? lo : hi;
        double a_ = ax * ax + ay * ay;               /* :106  ax**2 + ay**2 */
        double b_ = 2 * (ax * vx + ay * vy);         /* :107 */
        double c_ = vx * vx + vy * vy - bound * bound; /* :108 */
        double t1;
        if (fabs(a_) < 1e-12) {                      /* :111 */
            if (fabs(b_) < 1e-12) t1 = 0.0;
            else t1 = -c_ / b_;                      /* :118 */
        } else {
            double disc = b_ * b_ - 4 * a_ * c_;     /* :121 */
            if (!(disc > 0.0)) disc = 0.0;           /* max(0.0, disc) :123 */
            t1 = lower ? (-b_ - sqrt(disc)) / (2 * a_)   /* :124 */
                       : (-b_ + sqrt(disc)) / (2 * a_);  /* :158 */
        }
        t1 = clip(t1, 0.0, dt); /* :127 */
        double t2 = dt - t1;
        ovx = vx + ax * t1; /* :129 */
        ovy = vy + ay * t1;
        ox = x + vx * t1 + 0.5 * ax * (t1 * t1) + ovx * t2; /* :134 */
        oy = y + vy * t1 + 0.5 * ay * (t1 * t1) + ovy * t2;
    }
    out[0] = ox;
    out[1] = oy;
    out[2] = T_atan2(ovy, ovx); /* :98 / :136 / :170 */
    out[3] = sqrt(ovx * ovx + ovy * ovy);
    out[4] = ovx;
    out[5] = ovy;
    out[6] = ax;
    out[7] = ay;
}

/* PointMass._step_euler: physics/point_mass.py:177-207 (cross-check only; the reference
 * test asserts Hausdorff(newton, euler) < 0.01, tests/test_physics.py:248-249).
 * io: x, y, heading, vx, vy updated in place.                                           */
void t2do_pointmass_euler(const double* p, double* x, double* y, double* heading, double* vx,
                          double* vy, double ax, double ay, int interval) {
    int flags = (int)p[T2D_P_RANGE_FLAGS];
    int delta_t = (int)p[T2D_P_DELTA_T_MS];
    int n = interval / delta_t, rem = interval % delta_t;
    for (int i = 0; i <= n; ++i) {
        double dt = (double)delta_t / 1000;
        if (i == n) {
            if (rem <= 0) break;
            dt = (double)rem / 1000;
        }
        *vx += ax * dt; /* :188 */
        *vy += ay * dt;
        double speed = sqrt(*vx * *vx + *vy * *vy);
        double sc = (flags & T2D_RANGE_SPEED) ? clip(speed, p[T2D_P_SPEED_LO], p[T2D_P_SPEED_HI]) : speed;
        if (fabs(speed - sc) > 1e-12) { /* :195 */
            *vx = sc * T_cos(*heading);
            *vy = sc * T_sin(*heading);
        }
        *x += *vx * dt; /* :199 */
        *y += *vy * dt;
        *heading = T_atan2(*vy, *vx);
    }
}

/* Batched driver over an SoA pool slice (what t2d_integrate does on the GPU).
 * Inputs are the pool's fp32 columns; out is n x 8 doubles (un-rounded fp64 results):
 * x, y, heading, speed, vx, vy, applied0, applied1.  Inactive participants copy their
 * state through unchanged (applied = NaN).                                              */
void t2do_integrate(const double* rows, int row_stride, int n, const float* x, const float* y,
                    const float* heading, const float* speed, const float* vx, const float* vy,
                    const float* act0, const float* act1, const uint8_t* type_id,
                    const uint8_t* active, int interval_ms, double* out) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int i = 0; i < n; ++i) {
        double* o = out + 8 * (size_t)i;
        if (active && !active[i]) {
            o[0] = x[i]; o[1] = y[i]; o[2] = heading[i]; o[3] = speed[i];
            o[4] = vx ? vx[i] : NAN; o[5] = vy ? vy[i] : NAN; o[6] = NAN; o[7] = NAN;
            continue;
        }
        const double* p = rows + (size_t)type_id[i] * row_stride;
        int model = (int)p[T2D_P_MODEL];
        if (model == T2D_MODEL_KINEMATICS)
            t2do_kinematics(p, x[i], y[i], heading[i], speed[i], act0[i], act1[i], interval_ms, o);
        else if (model == T2D_MODEL_DYNAMICS)
            t2do_dynamics(p, x[i], y[i], heading[i], speed[i], act0[i], act1[i], interval_ms, o);
        else if (model == T2D_MODEL_POINTMASS_EULER) { /* PointMass(backend="euler").step: point_mass.py:228-229 */
            double ex = x[i], ey = y[i], eh = heading[i], evx = vx[i], evy = vy[i];
            t2do_pointmass_euler(p, &ex, &ey, &eh, &evx, &evy, act0[i], act1[i], interval_ms);
            o[0] = ex; o[1] = ey; o[2] = eh; o[3] = sqrt(evx * evx + evy * evy);
            o[4] = evx; o[5] = evy; o[6] = act0[i]; o[7] = act1[i];
        } else
            t2do_pointmass(p, x[i], y[i], vx[i], vy[i], act0[i], act1[i], interval_ms, o);
    }
}

/* ====================================================================================
 * Geometry / events
 * ================================================================================== */

/* Vehicle.get_pose: participant/element/vehicle.py:263-281 with the bbox vertex order of
 * vehicle.py:132-142: (+L/2,-W/2), (+L/2,+W/2), (-L/2,+W/2), (-L/2,-W/2)  (counter-clockwise).
 * shapely affine_transform matrix [cos h, -sin h, sin h, cos h, x, y]:
 *     X = cos*lx - sin*ly + x ;  Y = sin*lx + cos*ly + y        (left-to-right)
 * trig: 0 = deterministic t2do_sincos (the definition used for flags), 1 = libm.        */
void t2do_pose_obb(double x, double y, double h, double L, double W, int trig, double* v8) {
    double s, c;
    if (trig == 0) t2do_sincos(h, &s, &c);
    else { s = sin(h); c = cos(h); }
    const double lx[4] = {0.5 * L, 0.5 * L, -0.5 * L, -0.5 * L};
    const double ly[4] = {-0.5 * W, 0.5 * W, 0.5 * W, -0.5 * W};
    for (int k = 0; k < 4; ++k) {
        v8[2 * k] = c * lx[k] - s * ly[k] + x;
        v8[2 * k + 1] = s * lx[k] + c * ly[k] + y;
    }
}

/* orientation of r relative to the directed line p->q: > 0 left, < 0 right, 0 on it */
/* pose of the ego of every env (input of t2do_status_ex), batched so that the CPU baseline does
 * not loop in Python: pose8[e] = t2do_pose_obb(ego) for box-shaped egos, is_obb[e] = 0 otherwise */
void t2do_ego_poses(const double* rows, int row_stride, int n_env, int A, int ego_index,
                    const float* x, const float* y, const float* heading, const uint8_t* type_id,
                    int trig, double* pose8, double* xy, uint8_t* is_obb) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int e = 0; e < n_env; ++e) {
        size_t i = (size_t)e * A + ego_index;
        const double* p = rows + (size_t)type_id[i] * row_stride;
        xy[2 * e] = x[i];
        xy[2 * e + 1] = y[i];
        is_obb[e] = 0;
        for (int k = 0; k < 8; ++k) pose8[8 * (size_t)e + k] = 0.0;
        /* (an ego whose pose is not finite takes no part in the IoU events either: t2do_collide) */
        if ((int)p[T2D_P_SHAPE] == T2D_SHAPE_OBB && isfinite(x[i]) && isfinite(y[i]) && isfinite(heading[i])) {
            t2do_pose_obb(x[i], y[i], heading[i], p[T2D_P_LENGTH], p[T2D_P_WIDTH], trig, pose8 + 8 * (size_t)e);
            is_obb[e] = 1;
        }
    }
}

static double orient(const double* p, const double* q, const double* r) {
    double a = q[0] - p[0], b = r[1] - p[1];
    double c = q[1] - p[1], d = r[0] - p[0];
    return a * b - c * d;
}

/* an edge of the CCW polygon A separates when every vertex of B is strictly to its right */
static int has_separating_edge(const double* A, int nA, const double* B, int nB) {
    for (int i = 0; i < nA; ++i) {
        const double* p = A + 2 * i;
        const double* q = A + 2 * ((i + 1) % nA);
        int all_out = 1;
        for (int j = 0; j < nB; ++j)
            if (!(orient(p, q, B + 2 * j) < 0.0)) { all_out = 0; break; }
        if (all_out) return 1;
    }
    return 0;
}

/* shapely `A.intersects(B)` for convex polygons (collision.py:22,40): closed sets share at
 * least one point -- touching edges / corners and containment all count.  Separating-axis
 * theorem with strict separation.  Both polygons counter-clockwise.                      */
int t2do_convex_intersects(const double* A, int nA, const double* B, int nB) {
    if (has_separating_edge(A, nA, B, nB)) return 0;
    if (has_separating_edge(B, nB, A, nA)) return 0;
    return 1;
}

/* closed point-in-convex-polygon (CCW) */
int t2do_point_in_convex(const double* P, int n, const double* pt) {
    for (int i = 0; i < n; ++i)
        if (orient(P + 2 * i, P + 2 * ((i + 1) % n), pt) < 0.0) return 0;
    return 1;
}

static double seg_dist2(const double* p, const double* q, const double* c) {
    double dx = q[0] - p[0], dy = q[1] - p[1];
    double wx = c[0] - p[0], wy = c[1] - p[1];
    double dd = dx * dx + dy * dy;
    double t = 0.0;
    if (dd > 0.0) {
        t = (wx * dx + wy * dy) / dd;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    }
    double ex = wx - t * dx, ey = wy - t * dy;
    return ex * ex + ey * ey;
}

/* Pedestrian pose = (centre, radius) (pedestrian.py:138-149) vs convex polygon, closed */
int t2do_circle_convex_intersects(const double* c, double R, const double* P, int n) {
    if (t2do_point_in_convex(P, n, c)) return 1;
    double R2 = R * R;
    for (int i = 0; i < n; ++i)
        if (seg_dist2(P + 2 * i, P + 2 * ((i + 1) % n), c) <= R2) return 1;
    return 0;
}

int t2do_circle_circle_intersects(const double* c1, double R1, const double* c2, double R2) {
    double dx = c1[0] - c2[0], dy = c1[1] - c2[1];
    double rr = R1 + R2;
    return dx * dx + dy * dy <= rr * rr;
}

/* signed area * 2 (shoelace) -- used to normalise winding to CCW */
static double area2(const double* P, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const double* p = P + 2 * i;
        const double* q = P + 2 * ((i + 1) % n);
        a += p[0] * q[1] - q[0] * p[1];
    }
    return a;
}

/* load fp32 polygon -> fp64 CCW; returns n */
static int load_poly(const float* verts_xy, int v0, int v1, double* P) {
    int n = v1 - v0;
    for (int k = 0; k < n; ++k) {
        P[2 * k] = verts_xy[2 * (v0 + k)];
        P[2 * k + 1] = verts_xy[2 * (v0 + k) + 1];
    }
    if (area2(P, n) < 0.0) {
        for (int a = 0, b = n - 1; a < b; ++a, --b) {
            double tx = P[2 * a], ty = P[2 * a + 1];
            P[2 * a] = P[2 * b]; P[2 * a + 1] = P[2 * b + 1];
            P[2 * b] = tx; P[2 * b + 1] = ty;
        }
    }
    return n;
}

/* 1 if convex (after CCW normalisation, collinear runs allowed), 0 otherwise */
int t2do_polygon_is_convex(const float* verts_xy, int n) {
    double P[2 * 64];
    if (n < 3 || n > 64) return 0;
    load_poly(verts_xy, 0, n, P);
    if (!(area2(P, n) > 0.0)) return 0;
    for (int i = 0; i < n; ++i)
        if (orient(P + 2 * i, P + 2 * ((i + 1) % n), P + 2 * ((i + 2) % n)) < 0.0) return 0;
    return 1;
}

/* Polygons as the event kernels see them (t2d_api.hip prepare_polys does the same at t2d_set_*_geometry): a convex
 * polygon of 5..8 vertices is replaced by its fan of quads (v0 v1 v2 v3), (v0 v3 v4 v5), (v0 v5 v6 v7) -- the last
 * part a triangle for odd counts, parts without area dropped.  The union of the parts IS the polygon, so closed
 * `intersects` and point-in tests are the OR over the parts; evaluated on the parts, the fp64 orientation signs are
 * the ones the GPU evaluates.  P: CCW fp64 vertices.  Returns the number of parts (<= 3). */
static int fan_parts(const double* P, int n, double parts[3][8], int pn[3]) {
    int m = 0;
    if (n <= 4) {
        for (int k = 0; k < 2 * n; ++k) parts[0][k] = P[k];
        pn[0] = n;
        return 1;
    }
    for (int k = 1; k < n - 1; k += 2) {
        const int cnt = k + 2 <= n - 1 ? 4 : 3;
        const int idx[4] = {0, k, k + 1, k + 2};
        for (int j = 0; j < cnt; ++j) { parts[m][2 * j] = P[2 * idx[j]]; parts[m][2 * j + 1] = P[2 * idx[j] + 1]; }
        if (area2(parts[m], cnt) > 0.0) pn[m++] = cnt;
    }
    return m;
}

/* the lane polygons [l0, l1) as their parts: a CSR of its own (fp32 vertices, CCW); caller frees *vo and *xy */
static int decompose_lanes(const int32_t* lane_vert_off, const float* lane_xy, int l0, int l1, int32_t** vo, float** xy) {
    const int nl = l1 - l0 > 0 ? l1 - l0 : 0;
    *vo = (int32_t*)malloc(sizeof(int32_t) * (3 * (size_t)nl + 1));
    *xy = (float*)malloc(sizeof(float) * 2 * 4 * 3 * (size_t)(nl + 1));
    int np = 0;
    (*vo)[0] = 0;
    for (int li = l0; li < l1; ++li) {
        double P[2 * T2D_MAX_POLY_VERTS], parts[3][8];
        int pn[3];
        const int n = load_poly(lane_xy, lane_vert_off[li], lane_vert_off[li + 1], P);
        const int m = fan_parts(P, n, parts, pn);
        for (int k = 0; k < m; ++k) {
            for (int j = 0; j < 2 * pn[k]; ++j) (*xy)[2 * (*vo)[np] + j] = (float)parts[k][j];
            (*vo)[np + 1] = (*vo)[np] + pn[k];
            ++np;
        }
    }
    return np;
}

/* ------------------------------------------------------------------------------------------
 * Off-lane = `not union(lanes).contains(pose)`.  The reference's OffLane.update
 * (off_lane.py:16-17) is a stub; the predicate mirrored is OutBound.update out_bound.py:37-48:
 * `not region.contains(pose)`, shapely `contains` = no point of the pose in the exterior of the
 * region (touching the region's boundary from inside is still contained).  BUILD-DEFINED, parity
 * unpinned (no shapely here, and the reference never evaluates it).
 *
 * region U = closed union of the env's convex lane polygons.  For a convex pose P:
 *     P in U   <=>   centre(P) in U   and   no piece of the boundary of U meets the interior of P
 * (the interior of P is connected: if it does not meet the boundary of U it lies wholly inside or
 * wholly outside U, and the centre decides which).  Two exact short cuts come first, in this order:
 * a pose with all four vertices in ONE lane polygon is contained (convexity); a pose with a vertex
 * in NO lane polygon is not.  Only bodies that straddle lanes reach the boundary test.  This catches bodies that cut a corner of the
 * union with all four vertices in lanes, holes of the union inside a body, and a body that exactly
 * fills a gap between lanes.
 *
 * Boundary of U (computed once per env when the lanes are installed -- t2do_lane_boundary here,
 * build_lane_boundary in t2d_api.hip): every edge q0 -> q1 of every lane polygon L (CCW) minus the
 * parts whose right-hand (outer) side is covered by another lane M.  "Covered by M" = the closed
 * parametric clip of the edge against M's half-planes (num + t den >= 0, tc = -num / den, IEEE
 * division) with positive length, where an edge of M that is collinear with q0 -> q1 AND points
 * the same way rejects M (M then lies on L's own side of the line).  The covered intervals are
 * merged in ascending order; two that meet within T2D_LANE_TAU (in t) count as joined -- adjacent
 * lanes share a vertex, and the two clip parameters of that vertex agree only up to rounding.
 * What is left over becomes boundary pieces [A, B] in fp64 (A = q0 when the piece starts at t = 0,
 * else q0 + t d; likewise B).  Lane polygons that are meant to abut must share their edge LINE
 * exactly (same fp32 vertices, or axis-parallel): a sliver between two almost-collinear edges is a
 * real gap of U.
 *
 * Tests on the pose (orientation signs only, like t2do_convex_intersects):
 *   box:    piece [A, B] misses the open quad P  <=>  some edge of P has A and B on its outer side
 *           or on it (orient <= 0), or all four vertices of P lie on one closed side of line AB
 *   circle: piece within the open disc  <=>  squared distance(centre, [A, B]) < R^2
 * ---------------------------------------------------------------------------------------- */
#define T2D_LANE_TAU 1e-9

/* part of q0 -> q1 whose right-hand side is covered by the CCW convex polygon M: 1 and (*a, *b) */
static int edge_covered_by(const double* q0, const double* q1, const double* M, int n, double* a, double* b) {
    const double dx = q1[0] - q0[0], dy = q1[1] - q0[1];
    double t0 = 0.0, t1 = 1.0;
    for (int j = 0; j < n; ++j) {
        const double* f0 = M + 2 * j;
        const double* f1 = M + 2 * ((j + 1) % n);
        const double ex = f1[0] - f0[0], ey = f1[1] - f0[1];
        const double num = ex * (q0[1] - f0[1]) - ey * (q0[0] - f0[0]); /* inside <=> num + t*den >= 0 */
        const double den = ex * dy - ey * dx;
        if (den == 0.0) {
            if (num < 0.0) return 0;
            if (num == 0.0 && ex * dx + ey * dy > 0.0) return 0;
        } else {
            const double tc = -num / den;
            if (den > 0.0) t0 = tc > t0 ? tc : t0;
            else t1 = tc < t1 ? tc : t1;
        }
    }
    if (!(t0 < t1)) return 0;
    *a = t0; *b = t1;
    return 1;
}

/* Boundary pieces of the union of the lane polygons [l0, l1) (CSR as in t2d_set_lane_geometry).
 * pieces[4 * k] = Ax, Ay, Bx, By; owner[k] = lane polygon (absolute index) the piece is an edge part of.
 * Pieces are emitted lane by lane, edge by edge, in ascending t.  Returns the number of pieces; nothing is
 * written beyond cap (call with cap = 0 to size the arrays). */
static int lane_boundary_parts(const int32_t* lane_vert_off, const float* lane_xy, int l0, int l1, double* pieces,
                               int32_t* owner, int cap) {
    const int nl = l1 - l0;
    int count = 0;
    if (nl <= 0) return 0;
    double* polys = (double*)malloc(sizeof(double) * 2 * T2D_MAX_POLY_VERTS * (size_t)nl);
    int* pn = (int*)malloc(sizeof(int) * (size_t)nl);
    double* ia = (double*)malloc(sizeof(double) * 2 * (size_t)nl);
    double* ib = ia + nl;
    for (int i = 0; i < nl; ++i)
        pn[i] = load_poly(lane_xy, lane_vert_off[l0 + i], lane_vert_off[l0 + i + 1], polys + 2 * T2D_MAX_POLY_VERTS * (size_t)i);
    for (int i = 0; i < nl; ++i) {
        const double* L = polys + 2 * T2D_MAX_POLY_VERTS * (size_t)i;
        for (int j = 0; j < pn[i]; ++j) {
            const double* q0 = L + 2 * j;
            const double* q1 = L + 2 * ((j + 1) % pn[i]);
            const double dx = q1[0] - q0[0], dy = q1[1] - q0[1];
            if (dx == 0.0 && dy == 0.0) continue;
            int m = 0;
            for (int k = 0; k < nl; ++k) {
                if (k == i) continue;
                double a, b;
                if (!edge_covered_by(q0, q1, polys + 2 * T2D_MAX_POLY_VERTS * (size_t)k, pn[k], &a, &b)) continue;
                int pos = m++;   /* insertion sort by start */
                while (pos > 0 && ia[pos - 1] > a) { ia[pos] = ia[pos - 1]; ib[pos] = ib[pos - 1]; --pos; }
                ia[pos] = a; ib[pos] = b;
            }
            double r = 0.0;
            for (int k = 0; k <= m; ++k) {
                const double a = k < m ? ia[k] : 1.0;
                const int gap = k < m ? a > r + T2D_LANE_TAU : r < 1.0 - T2D_LANE_TAU;
                if (gap) {
                    if (count < cap) {
                        double* P = pieces + 4 * (size_t)count;
                        P[0] = r == 0.0 ? q0[0] : q0[0] + r * dx;
                        P[1] = r == 0.0 ? q0[1] : q0[1] + r * dy;
                        P[2] = a == 1.0 ? q1[0] : q0[0] + a * dx;
                        P[3] = a == 1.0 ? q1[1] : q0[1] + a * dy;
                        owner[count] = l0 + i;
                    }
                    ++count;
                }
                if (k < m && ib[k] > r) r = ib[k];
            }
        }
    }
    free(polys); free(pn); free(ia);
    return count;
}

/* public form: lane polygons as given (3..8 vertices); owner[k] indexes their PARTS (fan_parts), in order */
int t2do_lane_boundary(const int32_t* lane_vert_off, const float* lane_xy, int l0, int l1, double* pieces,
                       int32_t* owner, int cap) {
    int32_t* vo; float* xy;
    const int np = decompose_lanes(lane_vert_off, lane_xy, l0, l1, &vo, &xy);
    const int n = lane_boundary_parts(vo, xy, 0, np, pieces, owner, cap);
    free(vo); free(xy);
    return n;
}

/* 1 when the boundary piece A -> B meets the interior of the CCW convex quad pose8 */
int t2do_piece_meets_quad_interior(const double* piece, const double* pose8) {
    const double* A = piece;
    const double* B = piece + 2;
    for (int i = 0; i < 4; ++i) {
        const double* p = pose8 + 2 * i;
        const double* q = pose8 + 2 * ((i + 1) & 3);
        if (orient(p, q, A) <= 0.0 && orient(p, q, B) <= 0.0) return 0;
    }
    int all_ge = 1, all_le = 1;
    for (int k = 0; k < 4; ++k) {
        const double o = orient(A, B, pose8 + 2 * k);
        if (!(o >= 0.0)) all_ge = 0;
        if (!(o <= 0.0)) all_le = 0;
    }
    return !(all_ge || all_le);
}

static int point_in_lanes(const double* pt, const int32_t* lane_vert_off, const float* lane_xy, int l0, int l1) {
    for (int li = l0; li < l1; ++li) {
        double P[2 * T2D_MAX_POLY_VERTS];
        int n = load_poly(lane_xy, lane_vert_off[li], lane_vert_off[li + 1], P);
        if (t2do_point_in_convex(P, n, pt)) return 1;
    }
    return 0;
}

/* `union(lanes).contains(pose)`: box pose (8 doubles, vertex order of t2do_pose_obb, centre cxy) */
static int box_in_lane_union(const double* pose8, const double* cxy, const int32_t* lane_vert_off,
                             const float* lane_xy, int l0, int l1, const double* pieces, int n_pieces) {
#ifdef T2DO_LANE_SHORTCUTS   /* rounds 1-2: two short cuts ahead of the definition (equal verdicts; kept for comparison) */
    for (int li = l0; li < l1; ++li) {   /* all four vertices in one convex lane polygon: contained */
        double P[2 * T2D_MAX_POLY_VERTS];
        int n = load_poly(lane_xy, lane_vert_off[li], lane_vert_off[li + 1], P), k = 0;
        while (k < 4 && t2do_point_in_convex(P, n, pose8 + 2 * k)) ++k;
        if (k == 4) return 1;
    }
    for (int k = 0; k < 4; ++k)          /* a vertex in no lane polygon: not contained */
        if (!point_in_lanes(pose8 + 2 * k, lane_vert_off, lane_xy, l0, l1)) return 0;
#endif
    if (!point_in_lanes(cxy, lane_vert_off, lane_xy, l0, l1)) return 0;
    for (int k = 0; k < n_pieces; ++k)
        if (t2do_piece_meets_quad_interior(pieces + 4 * (size_t)k, pose8)) return 0;
    return 1;
}

/* circle pose (pedestrian.py:138-149): centre in U and no boundary piece inside the open disc */
static int circle_in_lane_union(const double* c, double R, const int32_t* lane_vert_off, const float* lane_xy,
                                int l0, int l1, const double* pieces, int n_pieces) {
    if (!point_in_lanes(c, lane_vert_off, lane_xy, l0, l1)) return 0;
    const double R2 = R * R;
    for (int k = 0; k < n_pieces; ++k)
        if (seg_dist2(pieces + 4 * (size_t)k, pieces + 4 * (size_t)k + 2, c) < R2) return 0;
    return 1;
}

/* stand-alone forms for the known-answer tests (parts and boundary are rebuilt on every call) */
int t2do_pose_in_lane_union(const double* pose8, const double* cxy, const int32_t* lane_vert_off,
                            const float* lane_xy, int l0, int l1) {
    int32_t* vo; float* xy;
    const int np = decompose_lanes(lane_vert_off, lane_xy, l0, l1, &vo, &xy);
    const int n = lane_boundary_parts(vo, xy, 0, np, NULL, NULL, 0);
    double* pieces = (double*)malloc(sizeof(double) * 4 * (size_t)(n + 1));
    int32_t* owner = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    lane_boundary_parts(vo, xy, 0, np, pieces, owner, n);
    const int r = box_in_lane_union(pose8, cxy, vo, xy, 0, np, pieces, n);
    free(pieces); free(owner); free(vo); free(xy);
    return r;
}

int t2do_circle_in_lane_union(const double* c, double R, const int32_t* lane_vert_off, const float* lane_xy,
                              int l0, int l1) {
    int32_t* vo; float* xy;
    const int np = decompose_lanes(lane_vert_off, lane_xy, l0, l1, &vo, &xy);
    const int n = lane_boundary_parts(vo, xy, 0, np, NULL, NULL, 0);
    double* pieces = (double*)malloc(sizeof(double) * 4 * (size_t)(n + 1));
    int32_t* owner = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    lane_boundary_parts(vo, xy, 0, np, pieces, owner, n);
    const int r = circle_in_lane_union(c, R, vo, xy, 0, np, pieces, n);
    free(pieces); free(owner); free(vo); free(xy);
    return r;
}

/* Event flags of every participant (what t2d_collide does on the GPU), brute force:
 *   COLLISION_DYNAMIC  intended DynamicCollision (collision.py:18-25): pose intersects the
 *                      pose of any other ACTIVE participant of the same env, all pairs
 *   COLLISION_STATIC   StaticCollision.update (collision.py:37-43): any static polygon
 *   OUT_BOUND          OutBound.update (out_bound.py:37-48): not boundary.contains(pose);
 *                      touching the boundary from inside is still contained -> strict tests
 *   OFF_LANE           build-defined (reference stub off_lane.py:16-17 returns False):
 *                      not union(lane polygons).contains(pose), t2do_pose_in_lane_union above
 *                      (circle: its centre lies in no lane polygon); never raised for an env
 *                      without lane polygons
 * CSR arrays as in t2d_set_static_geometry / t2d_set_lane_geometry (may be NULL = none).   */
void t2do_collide(const double* rows, int row_stride, int n_env, int A, const float* x,
                  const float* y, const float* heading, const uint8_t* type_id,
                  const uint8_t* active, const int32_t* env_poly_off, const int32_t* poly_vert_off,
                  const float* poly_xy, const float* boundary, const uint8_t* boundary_valid,
                  const int32_t* env_lane_off, const int32_t* lane_vert_off, const float* lane_xy,
                  int trig, uint32_t* flags, uint32_t* env_flags) {
#pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
    double* V = (double*)malloc(sizeof(double) * 8 * (size_t)A);
    double* C = (double*)malloc(sizeof(double) * 3 * (size_t)A);
    int* kind = (int*)malloc(sizeof(int) * (size_t)A);
    int* present = (int*)malloc(sizeof(int) * (size_t)A);
#pragma omp for schedule(dynamic, 4)
    for (int e = 0; e < n_env; ++e) {
        size_t base = (size_t)e * A;
        const int n_lanes_e = env_lane_off ? env_lane_off[e + 1] - env_lane_off[e] : 0;
        int n_pieces = 0, n_lane_parts = 0;
        double* pieces = NULL;
        int32_t* lvo = NULL;
        float* lxy = NULL;
        if (n_lanes_e > 0) {   /* parts + boundary of the env's lane union (the GPU path builds both once, at t2d_set_lane_geometry) */
            n_lane_parts = decompose_lanes(lane_vert_off, lane_xy, env_lane_off[e], env_lane_off[e + 1], &lvo, &lxy);
            n_pieces = lane_boundary_parts(lvo, lxy, 0, n_lane_parts, NULL, NULL, 0);
            pieces = (double*)malloc((sizeof(double) * 4 + sizeof(int32_t)) * (size_t)(n_pieces + 1));
            lane_boundary_parts(lvo, lxy, 0, n_lane_parts, pieces, (int32_t*)(pieces + 4 * (size_t)(n_pieces + 1)), n_pieces);
        }
        for (int i = 0; i < A; ++i) {
            flags[base + i] = 0;
            /* build-defined: a participant whose pose is not finite (a NaN action went through np.clip, an overflow) takes no
             * part in event detection -- it raises no flag and nobody collides with it -- exactly like an inactive one (the
             * reference hands such a pose to GEOS, whose answer is not defined) */
            present[i] = active[base + i] && isfinite(x[base + i]) && isfinite(y[base + i]) && isfinite(heading[base + i]);
            if (!present[i]) continue;
            const double* p = rows + (size_t)type_id[base + i] * row_stride;
            kind[i] = (int)p[T2D_P_SHAPE];
            C[3 * i] = x[base + i];
            C[3 * i + 1] = y[base + i];
            C[3 * i + 2] = 0.5 * p[T2D_P_WIDTH]; /* pedestrian radius = width / 2 */
            if (kind[i] == T2D_SHAPE_OBB)
                t2do_pose_obb(x[base + i], y[base + i], heading[base + i], p[T2D_P_LENGTH],
                              p[T2D_P_WIDTH], trig, V + 8 * i);
        }
        for (int i = 0; i < A; ++i) {
            if (!present[i]) continue;
            uint32_t f = 0;
            /* participant vs participant */
            for (int j = 0; j < A && !(f & T2D_FLAG_COLLISION_DYNAMIC); ++j) {
                if (j == i || !present[j]) continue;
                int hit;
                if (kind[i] == T2D_SHAPE_OBB && kind[j] == T2D_SHAPE_OBB)
                    hit = t2do_convex_intersects(V + 8 * i, 4, V + 8 * j, 4);
                else if (kind[i] == T2D_SHAPE_OBB)
                    hit = t2do_circle_convex_intersects(C + 3 * j, C[3 * j + 2], V + 8 * i, 4);
                else if (kind[j] == T2D_SHAPE_OBB)
                    hit = t2do_circle_convex_intersects(C + 3 * i, C[3 * i + 2], V + 8 * j, 4);
                else
                    hit = t2do_circle_circle_intersects(C + 3 * i, C[3 * i + 2], C + 3 * j, C[3 * j + 2]);
                if (hit) f |= T2D_FLAG_COLLISION_DYNAMIC;
            }
            /* participant vs static polygons */
            if (env_poly_off) {
                for (int pi = env_poly_off[e]; pi < env_poly_off[e + 1] && !(f & T2D_FLAG_COLLISION_STATIC); ++pi) {
                    double P[2 * T2D_MAX_POLY_VERTS], parts[3][8];
                    int pn[3];
                    int n = load_poly(poly_xy, poly_vert_off[pi], poly_vert_off[pi + 1], P);
                    const int m = fan_parts(P, n, parts, pn);   /* the polygon as its quads: OR over the parts */
                    for (int k = 0; k < m; ++k) {
                        int hit = kind[i] == T2D_SHAPE_OBB
                                      ? t2do_convex_intersects(V + 8 * i, 4, parts[k], pn[k])
                                      : t2do_circle_convex_intersects(C + 3 * i, C[3 * i + 2], parts[k], pn[k]);
                        if (hit) { f |= T2D_FLAG_COLLISION_STATIC; break; }
                    }
                }
            }
            /* map boundary */
            if (boundary && (!boundary_valid || boundary_valid[e])) {
                double xmin = boundary[4 * e], xmax = boundary[4 * e + 1];
                double ymin = boundary[4 * e + 2], ymax = boundary[4 * e + 3];
                int out = 0;
                if (kind[i] == T2D_SHAPE_OBB) {
                    for (int k = 0; k < 4; ++k) {
                        double vx_ = V[8 * i + 2 * k], vy_ = V[8 * i + 2 * k + 1];
                        if (vx_ < xmin || vx_ > xmax || vy_ < ymin || vy_ > ymax) out = 1;
                    }
                } else {
                    double cx = C[3 * i], cy = C[3 * i + 1], R = C[3 * i + 2];
                    if (cx - R < xmin || cx + R > xmax || cy - R < ymin || cy + R > ymax) out = 1;
                }
                if (out) f |= T2D_FLAG_OUT_BOUND;
            }
            /* lanes (build-defined): not union(lanes).contains(pose) */
            if (n_lanes_e > 0) {
                const int in = kind[i] == T2D_SHAPE_OBB
                                   ? box_in_lane_union(V + 8 * i, C + 3 * i, lvo, lxy, 0, n_lane_parts, pieces, n_pieces)
                                   : circle_in_lane_union(C + 3 * i, C[3 * i + 2], lvo, lxy, 0, n_lane_parts, pieces, n_pieces);
                if (!in) f |= T2D_FLAG_OFF_LANE;
            }
            flags[base + i] = f;
        }
        uint32_t ef = 0;
        for (int i = 0; i < A; ++i) ef |= flags[base + i];
        env_flags[e] = ef;
        free(pieces); free(lvo); free(lxy);
    }
    free(V); free(C); free(kind); free(present);
    }
}

/* ------------------------------------------------------------------------------------------
 * IoU of two convex quadrilaterals (Arrival.update arrival.py:42-44, NoAction.update
 * no_action.py:44-46: iou = intersection.area / union.area).  The reference delegates to GEOS
 * overlay (parity unpinned); this restatement integrates the boundary of A n B directly:
 * every edge of A is clipped to the closed polygon B, every edge of B to the OPEN side of A's
 * edge lines where they are parallel (so coincident boundary pieces count once), and
 * 2*area = sum of cross(a - O, b - O) over the kept oriented pieces (Green's theorem), O = A[0].
 * Partial sums are combined in the fixed tree order the GPU uses.  Both quads CCW.
 * ---------------------------------------------------------------------------------------- */
static double clipped_edge_term(const double* p0, const double* p1, const double* Q, int strict,
                                const double* O) {
    const double dx = p1[0] - p0[0], dy = p1[1] - p0[1];
    double t0 = 0.0, t1 = 1.0;
    int ok = 1;
    for (int j = 0; j < 4; ++j) {
        const double* q0 = Q + 2 * j;
        const double* q1 = Q + 2 * ((j + 1) & 3);
        const double ex = q1[0] - q0[0], ey = q1[1] - q0[1];
        const double num = ex * (p0[1] - q0[1]) - ey * (p0[0] - q0[0]); /* inside <=> num + t*den >= 0 */
        const double den = ex * dy - ey * dx;
        if (den == 0.0) {
            if (num < 0.0 || (strict && num == 0.0)) ok = 0;
        } else {
            const double tc = -num / den;
            if (den > 0.0) t0 = tc > t0 ? tc : t0;
            else t1 = tc < t1 ? tc : t1;
        }
    }
    if (!ok || !(t0 < t1)) return 0.0;
    const double ax = p0[0] + t0 * dx - O[0], ay = p0[1] + t0 * dy - O[1];
    const double bx = p0[0] + t1 * dx - O[0], by = p0[1] + t1 * dy - O[1];
    return ax * by - bx * ay;
}

static double quad_area2(const double* P) {
    double a = 0.0;
    for (int i = 0; i < 4; ++i) {
        const double* p = P + 2 * i;
        const double* q = P + 2 * ((i + 1) & 3);
        a += (p[0] - P[0]) * (q[1] - P[1]) - (q[0] - P[0]) * (p[1] - P[1]);
    }
    return a;
}

/* twice the area of A n B */
double t2do_quad_intersection_area2(const double* A, const double* B) {
    double s[8];
    for (int i = 0; i < 4; ++i) s[i] = clipped_edge_term(A + 2 * i, A + 2 * ((i + 1) & 3), B, 0, A);
    for (int i = 0; i < 4; ++i) s[4 + i] = clipped_edge_term(B + 2 * i, B + 2 * ((i + 1) & 3), A, 1, A);
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

double t2do_quad_iou(const double* A, const double* B) {
    double inter = t2do_quad_intersection_area2(A, B);
    if (inter < 0.0) inter = 0.0;
    const double uni = quad_area2(A) + quad_area2(B) - inter;
    return inter / uni;
}

/* _ParkingScenarioManager.update/check_status (envs/parking.py:352-392) and
 * ParkingEnv.step/_get_reward (envs/parking.py:219-256,148-190), in the reference's early-return
 * order, for the ego of every env:
 *   time exceed (cnt_step > max_step, time_exceed.py:32-33)   -> TIME_EXCEEDED; later detectors are
 *                                                                NOT updated this step
 *   no action   (NoAction.update no_action.py:32-53)          -> traffic_status = 5 (the reference
 *                 stores ScenarioStatus.NO_ACTION in traffic_status, parking.py:373), scenario NORMAL
 *   out of bound                                              -> OUT_BOUND
 *   static collision                                          -> FAILED + COLLISION_STATIC
 *   [build-defined: dynamic collision / off-lane when enabled -> FAILED + COLLISION_DYNAMIC / OFF_LANE]
 *   arrival     (Arrival.update arrival.py:32-47)             -> COMPLETED when IoU >= threshold
 * reward (_get_reward :148-190): -5 static collision, -1 time exceeded, -5 out of bound, +5 completed,
 * else time penalty -tanh(cnt/max_step)*0.001 + (shaped_reward: IoU gain over the best IoU so far +
 * 0.1 * improvement of the best distance to the target centroid).  The no-action case falls into
 * the shaped branch with iou = None exactly like the reference.
 * Per-env IoU state (NULL when the IoU features are off): ego_pose (8 doubles, CCW, from
 * t2do_pose_obb), ego_xy, ego_is_obb, target (8 doubles CCW) / target_c (centroid) / has_target,
 * last_pose / last_valid / cnt_na (NoAction), max_iou / min_dist (reward shaping), iou_out (NaN = None).
 * status: 4 bytes per env = scenario, traffic, terminated, truncated.                    */
void t2do_status_ex(const t2d_status_config* cfg, int n_env, int A, const uint32_t* flags,
                    int interval_ms, int32_t* cnt_step, int32_t* frame_ms, uint8_t* status, float* reward,
                    const double* ego_pose, const double* ego_xy, const uint8_t* ego_is_obb,
                    const double* target, const double* target_c, int has_target, double* last_pose,
                    uint8_t* last_valid, int32_t* cnt_na, double* max_iou, double* min_dist, float* iou_out) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int e = 0; e < n_env; ++e) {
        cnt_step[e] += 1; /* parking.py:353 */
        frame_ms[e] += interval_ms;
        uint32_t f = flags[(size_t)e * A + cfg->ego_index];
        int scen = T2D_SCENARIO_NORMAL, traf = T2D_TRAFFIC_NORMAL;
        double iou = 0.0;
        int has_iou = 0;
        const int obb = ego_is_obb ? ego_is_obb[e] : 0;
        if (cfg->max_step > 0 && cnt_step[e] > cfg->max_step) {
            scen = T2D_SCENARIO_TIME_EXCEEDED; /* parking.py:366-369 */
        } else {
            int na = 0;
            if (cfg->check_no_action && obb && last_pose) { /* parking.py:371-374, no_action.py:41-53 */
                const double* pose = ego_pose + 8 * (size_t)e;
                double* last = last_pose + 8 * (size_t)e;
                if (!last_valid[e]) {
                    last_valid[e] = 1;
                } else {
                    const double i2 = t2do_quad_iou(pose, last);
                    cnt_na[e] = i2 > (double)cfg->no_action_iou ? cnt_na[e] + 1 : 0;
                }
                memcpy(last, pose, 8 * sizeof(double));
                na = cnt_na[e] > cfg->no_action_max_step;
            }
            if (na) {
                traf = T2D_TRAFFIC_NO_ACTION_QUIRK;
            } else if (f & T2D_FLAG_OUT_BOUND) {
                scen = T2D_SCENARIO_OUT_BOUND;
            } else if (f & T2D_FLAG_COLLISION_STATIC) {
                scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_STATIC;
            } else if (cfg->check_dynamic && (f & T2D_FLAG_COLLISION_DYNAMIC)) {
                scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_DYNAMIC;
            } else if (cfg->check_off_lane && (f & T2D_FLAG_OFF_LANE)) {
                scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_OFF_LANE;
            } else if (cfg->check_arrival && has_target && obb) { /* parking.py:387-390 */
                iou = t2do_quad_iou(ego_pose + 8 * (size_t)e, target + 8 * (size_t)e);
                has_iou = 1;
                if (iou >= (double)cfg->arrival_threshold) scen = T2D_SCENARIO_COMPLETED;
            }
        }
        double r;
        if (traf == T2D_TRAFFIC_COLLISION_STATIC) r = cfg->reward_collision;           /* :151-152 */
        else if (scen == T2D_SCENARIO_TIME_EXCEEDED || scen == T2D_SCENARIO_NO_ACTION) r = cfg->reward_time_exceed;
        else if (scen == T2D_SCENARIO_OUT_BOUND) r = cfg->reward_out_bound;
        else if (scen == T2D_SCENARIO_COMPLETED) r = cfg->reward_completed;
        else if (traf == T2D_TRAFFIC_COLLISION_DYNAMIC || traf == T2D_TRAFFIC_OFF_LANE) r = cfg->reward_collision;
        else {
            r = cfg->max_step > 0 ? -tanh((double)cnt_step[e] / (double)cfg->max_step) * (double)cfg->time_penalty_scale
                                  : 0.0; /* :163 */
            if (cfg->shaped_reward && max_iou) {
                double iou_reward = 0.0;                                               /* :164-167 */
                if (has_iou) iou_reward = max_iou[e] == -INFINITY ? iou : iou - max_iou[e];
                r = r + iou_reward;                                                    /* :169 */
                if (has_iou) max_iou[e] = max_iou[e] > iou ? max_iou[e] : iou;         /* :170 */
                if (has_target) {
                    const double dx = ego_xy[2 * e] - target_c[2 * e], dy = ego_xy[2 * e + 1] - target_c[2 * e + 1];
                    const double d = sqrt(dx * dx + dy * dy);                          /* :172-185 */
                    if (d < min_dist[e]) {
                        r += (min_dist[e] - d) * (double)cfg->dist_reward_scale;       /* :186-188 */
                        min_dist[e] = d;
                    }
                }
            }
        }
        status[4 * e] = (uint8_t)scen;
        status[4 * e + 1] = (uint8_t)traf;
        status[4 * e + 2] = scen == T2D_SCENARIO_COMPLETED;                       /* :245-246 */
        status[4 * e + 3] = scen != T2D_SCENARIO_COMPLETED &&
                            (scen != T2D_SCENARIO_NORMAL || traf != T2D_TRAFFIC_NORMAL); /* :247-248 */
        reward[e] = (float)r;
        if (iou_out) iou_out[e] = has_iou ? (float)iou : NAN;
    }
}

void t2do_status(const t2d_status_config* cfg, int n_env, int A, const uint32_t* flags,
                 int interval_ms, int32_t* cnt_step, int32_t* frame_ms, uint8_t* status,
                 float* reward) {
    t2do_status_ex(cfg, n_env, A, flags, interval_ms, cnt_step, frame_ms, status, reward, NULL, NULL, NULL,
                   NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL);
}

/* ==========================================================================================
 * Single-line lidar of the ego (scope row f2): SingleLineLidar._scan_obstacles,
 * sensor/lidar.py:128-221, with the ring transform of _rotate_and_filter_obstacles :98-126.
 * Obstacles = the env's static polygons (areas of type "obstacle", :137-143) and, when
 * include_participants, the poses of the other ACTIVE box-shaped participants (:146-153;
 * pedestrians return (location, radius) from get_pose and are skipped by the isinstance test).
 * beam_sin / beam_cos: sin / cos of linspace(0, 2 pi, n_beams, endpoint=False) computed by the host
 * (numpy), so every implementation shares them bit for bit.  The sensor's own rotation uses the
 * deterministic sincos (trig = 0) or libm (trig = 1).  out: n_env x n_beams fp32, +inf = no return.
 * ======================================================================================== */
static double lidar_edge(double a, double b, double lx, double ly, double R, double x1, double y1,
                         double x2, double y2) {
    const double tz = 1e-8, tinf = R * 10;            /* :198-199 */
    const double d = y2 - y1, e = x1 - x2, f = y1 * x2 - x1 * y2;   /* :183-185 */
    double det = a * e - b * d;                       /* :188 */
    const int parallel = det == 0.0;
    if (parallel) det = 1.0;
    double rx = (b * f) / det;                        /* c = 0: b*f - c*e = b*f, c*d - a*f = -(a*f)  :191-192 */
    double ry = (-(a * f)) / det;
    const double mx = tz > lx ? tz : lx, nx = -tz < lx ? -tz : lx;
    const double my = tz > ly ? tz : ly, ny = -tz < ly ? -tz : ly;
    if (rx > mx + tz) rx = tinf;                      /* :205-208 */
    if (rx < nx - tz) rx = tinf;
    if (ry > my + tz) ry = tinf;
    if (ry < ny - tz) ry = tinf;
    if (rx > (x1 > x2 ? x1 : x2) + tz) rx = tinf;     /* :210-213 */
    if (rx < (x1 < x2 ? x1 : x2) - tz) rx = tinf;
    if (ry > (y1 > y2 ? y1 : y2) + tz) ry = tinf;
    if (ry < (y1 < y2 ? y1 : y2) - tz) ry = tinf;
    if (parallel) rx = tinf;                          /* :215 */
    return sqrt(rx * rx + ry * ry);                   /* :218 */
}

void t2do_lidar(const double* rows, int row_stride, int n_env, int A, int ego_index, const float* x,
                const float* y, const float* heading, const uint8_t* type_id, const uint8_t* active,
                const int32_t* env_poly_off, const int32_t* poly_vert_off, const float* poly_xy,
                int include_participants, int n_beams, double max_range, const double* beam_sin,
                const double* beam_cos, int trig, float* out) {
    double* ex = (double*)malloc(sizeof(double) * 4 * (size_t)(A * 4 + 16 * T2D_MAX_POLY_VERTS * 64));
    for (int env = 0; env < n_env; ++env) {
        const size_t base = (size_t)env * A;
        const size_t ie = base + ego_index;
        float* o = out + (size_t)env * n_beams;
        /* (build-defined: an ego whose pose is not finite scans nothing, a participant whose pose is not finite is no obstacle) */
        if (!active[ie] || !isfinite(x[ie]) || !isfinite(y[ie]) || !isfinite(heading[ie])) {
            for (int k = 0; k < n_beams; ++k) o[k] = INFINITY;
            continue;
        }
        double sn, cs;
        if (trig == 0) t2do_sincos((double)heading[ie], &sn, &cs);
        else { sn = sin((double)heading[ie]); cs = cos((double)heading[ie]); }
        const double px = x[ie], py = y[ie];
        const double x_off = -px * cs - py * sn;      /* :112-113 */
        const double y_off = px * sn - py * cs;
        int ne = 0;
        double ring[2 * T2D_MAX_POLY_VERTS];
        /* gather rings -> edges in the sensor frame */
        #define T2DO_ADD_RING(n_)                                                            \
            for (int k = 0; k < (n_); ++k) {                                                 \
                const int k2 = (k + 1) % (n_);                                               \
                ex[4 * ne + 0] = cs * ring[2 * k] + sn * ring[2 * k + 1] + x_off;            \
                ex[4 * ne + 1] = -sn * ring[2 * k] + cs * ring[2 * k + 1] + y_off;           \
                ex[4 * ne + 2] = cs * ring[2 * k2] + sn * ring[2 * k2 + 1] + x_off;          \
                ex[4 * ne + 3] = -sn * ring[2 * k2] + cs * ring[2 * k2 + 1] + y_off;         \
                ++ne;                                                                        \
            }
        if (env_poly_off) {
            for (int p = env_poly_off[env]; p < env_poly_off[env + 1]; ++p) {
                const int v0 = poly_vert_off[p], n = poly_vert_off[p + 1] - v0;
                for (int k = 0; k < n; ++k) { ring[2 * k] = poly_xy[2 * (v0 + k)]; ring[2 * k + 1] = poly_xy[2 * (v0 + k) + 1]; }
                T2DO_ADD_RING(n)
            }
        }
        if (include_participants) {
            for (int j = 0; j < A; ++j) {
                if (j == ego_index || !active[base + j]) continue;
                if (!isfinite(x[base + j]) || !isfinite(y[base + j]) || !isfinite(heading[base + j])) continue;
                const double* p = rows + (size_t)type_id[base + j] * row_stride;
                if ((int)p[T2D_P_SHAPE] != T2D_SHAPE_OBB) continue;
                t2do_pose_obb(x[base + j], y[base + j], heading[base + j], p[T2D_P_LENGTH], p[T2D_P_WIDTH], trig, ring);
                T2DO_ADD_RING(4)
            }
        }
        #undef T2DO_ADD_RING
        for (int k = 0; k < n_beams; ++k) {
            if (ne == 0) { o[k] = INFINITY; continue; }       /* :173-175 */
            const double a = beam_sin[k], b = -beam_cos[k];   /* :161-162 */
            const double lx = beam_cos[k] * max_range, ly = beam_sin[k] * max_range;   /* :201-204 */
            double best = INFINITY;
            for (int q = 0; q < ne; ++q) {
                const double dd = lidar_edge(a, b, lx, ly, max_range, ex[4 * q], ex[4 * q + 1], ex[4 * q + 2], ex[4 * q + 3]);
                best = dd < best ? dd : best;
            }
            best = best < 0 ? 0 : (best > max_range ? max_range : best);   /* np.clip :219 */
            o[k] = best == max_range ? INFINITY : (float)best;              /* :220 */
        }
    }
    free(ex);
}

/* ------------------------------------------------------------------------------------------
 * IDM car-following controller (scope row f3).
 *   IDMController.step               controller/idm_controller.py:59-93
 *   IDMController._idm_acceleration  controller/idm_controller.py:95-141
 * Python's float ** float is C pow(); np.hypot is C hypot().  g_trig = 0 uses those (pins the
 * restatement against tests/golden/idm.npz, produced by running the reference); g_trig = 1 uses
 * the deterministic exp/log/pow spec shared with the GPU (tactics2d_amd/csrc/t2d_math.h), which
 * the golden test bounds against libm.  The leader rule of t2do_idm is BUILD-DEFINED: the
 * reference receives `leading_state` from its caller.
 * ---------------------------------------------------------------------------------------- */
double t2do_log(double x) { /* fdlibm e_log.c formulation, no fma */
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440) {
        m = m * 2.0;
        e -= 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

double t2do_exp(double x) { /* fdlibm e_exp.c formulation, no fma */
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.1332191019411) return 0.0;
    const double k = rint(x * invln2);
    const double hi = x - k * ln2HI;
    const double lo = k * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return ldexp(y, (int)k);
}

double t2do_pow(double x, double y) {
    if (y == 0.0) return 1.0;
    if (x != x || y != y) return x + y;
    const double yi = rint(y);
    if (yi == y && fabs(y) <= 64.0) {
        int n = (int)fabs(yi);
        double r = 1.0, b = x;
        while (n) {
            if (n & 1) r = r * b;
            b = b * b;
            n >>= 1;
        }
        return y < 0.0 ? 1.0 / r : r;
    }
    if (x < 0.0) return NAN; /* Python: complex result, outside the contract */
    if (x == 0.0) return y > 0.0 ? 0.0 : INFINITY;
    return t2do_exp(y * t2do_log(x));
}
static double T_pow(double x, double y) { return g_trig ? t2do_pow(x, y) : pow(x, y); }
static double T_hypot(double x, double y) { return g_trig ? sqrt(x * x + y * y) : hypot(x, y); }

/* one IDMController.step: c = {desired_speed, time_headway, min_spacing, max_acceleration,
 * comfortable_deceleration, delta, ...}; returns the clipped acceleration (steering is 0.0) */
double t2do_idm_accel(const double* c, double v, int has_lead, double dx, double dy, double v_lead) {
    const double des = c[T2D_IDM_DESIRED_SPEED], T = c[T2D_IDM_TIME_HEADWAY], s0 = c[T2D_IDM_MIN_SPACING];
    const double amax = c[T2D_IDM_MAX_ACCEL], b = c[T2D_IDM_COMF_DECEL], delta = c[T2D_IDM_DELTA];
    double a;
    if (!has_lead) { /* :75-85 */
        if (des > 0.0) a = amax * (1.0 - T_pow(v / des, delta));
        else a = v > 0.0 ? -b : 0.0;
    } else {
        const double dist = T_hypot(dx, dy);                                      /* :111-113 */
        const double dv = v_lead - v;                                             /* :116 */
        double s_star = s0 + v * T + (v * dv) / (2.0 * sqrt(amax * b));           /* :120-124 */
        if (s0 > s_star) s_star = s0;                                             /* :125 max() */
        if (dist > 0.0) {                                                         /* :129 */
            const double term = des > 0.0 ? T_pow(v / des, delta) : (v > 0.0 ? 1.0 : 0.0);
            const double q = s_star / dist;
            a = amax * (1.0 - term - q * q);                                      /* :137-139 */
        } else {
            a = -b;                                                               /* :141 */
        }
    }
    return clip(a, -b, amax);                                                     /* :90 */
}

/* Batched: every controlled participant of every env.  forced_leader: NULL or [n_env*A] with an agent
 * index, T2D_IDM_LEADER_FREE or T2D_IDM_LEADER_SEARCH.  act0/act1 are updated in place for controlled
 * participants only (fp32, as the pool stores them); leader_out[i] = chosen leader or -1. */
void t2do_idm(const double* ctrl_rows, int row_stride, int n_ctrl, const uint8_t* ctrl_id, int n_env, int A,
              const float* x, const float* y, const float* heading, const float* speed, const uint8_t* active,
              const int32_t* forced_leader, float* act0, float* act1, int32_t* leader_out) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int e = 0; e < n_env; ++e) {
        const size_t base = (size_t)e * A;
        for (int i = 0; i < A; ++i) {
            const size_t idx = base + i;
            int lead = -1;
            const int ctrl = ctrl_id[idx];
            if (active[idx] && ctrl != T2D_IDM_NONE && ctrl < n_ctrl) {
                const double* c = ctrl_rows + (size_t)ctrl * row_stride;
                const double hw = c[T2D_IDM_LANE_HALF_WIDTH], horizon = c[T2D_IDM_HORIZON];
                double sn, cs;
                t2do_sincos((double)heading[idx], &sn, &cs); /* the rule is build-defined: always the det spec */
                const int want = forced_leader ? forced_leader[idx] : T2D_IDM_LEADER_SEARCH;
                if (want >= 0 && want < A && want != i && active[base + want]) lead = want;
                double best = INFINITY;
                for (int j = 0; want == T2D_IDM_LEADER_SEARCH && j < A; ++j) {
                    if (j == i || !active[base + j]) continue;
                    const double dx = (double)x[base + j] - (double)x[idx];
                    const double dy = (double)y[base + j] - (double)y[idx];
                    const double lon = fma(dx, cs, dy * sn); /* one rounding each, as on the GPU */
                    const double lat = fma(dy, cs, -(dx * sn));
                    if (lon > 0.0 && lon <= horizon && fabs(lat) <= hw && lon < best) {
                        best = lon;
                        lead = j;
                    }
                }
                double dx = 0.0, dy = 0.0, vl = 0.0;
                if (lead >= 0) {
                    dx = (double)x[base + lead] - (double)x[idx];
                    dy = (double)y[base + lead] - (double)y[idx];
                    vl = (double)speed[base + lead];
                }
                act0[idx] = (float)t2do_idm_accel(c, (double)speed[idx], lead >= 0, dx, dy, vl);
                act1[idx] = 0.0f;
            }
            leader_out[idx] = lead;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * verify_state: "very rough check" of a candidate state against the last one.
 *   SingleTrackKinematics.verify_state  physics/single_track_kinematics.py:200-250
 *   SingleTrackDynamics.verify_state    physics/single_track_dynamics.py:253-306 (same check)
 *   PointMass.verify_state              physics/point_mass.py:234-259
 * Quirks kept: any unbounded range -> True; x/y use STRICT inequalities against a range whose ends
 * are not sorted (cos < 0 makes it empty); the heading window wraps.
 * ---------------------------------------------------------------------------------------- */
int t2do_verify_state(const double* p, double lx, double ly, double lh, double lv, double lvx, double lvy,
                      double x, double y, double h, double v, int interval_ms) {
    if (interval_ms == 0) return 1;
    const int model = (int)p[T2D_P_MODEL];
    const int flags = (int)p[T2D_P_RANGE_FLAGS];
    if (model == T2D_MODEL_POINTMASS) { /* point_mass.py:249-259 */
        const double dt = (double)interval_ms / 1000;
        const double den = 2 / (dt * dt);
        const double ax = (x - lx - lvx * dt) * den;
        const double ay = (y - ly - lvy * dt) * den;
        if (flags & T2D_RANGE_ACCEL) {
            const double a = sqrt(ax * ax + ay * ay);
            if (!(p[T2D_P_ACCEL_LO] <= a && a <= p[T2D_P_ACCEL_HI])) return 0;
        }
        return 1;
    }
    const double dt = (double)interval_ms / 1000;
    if ((flags & 7) != 7) return 1; /* None in [steer_range, speed_range, accel_range] */
    const double wb = p[T2D_P_WB], k = p[T2D_P_LR] / wb;
    const double st[2] = {p[T2D_P_STEER_LO], p[T2D_P_STEER_HI]};
    const double ac[2] = {p[T2D_P_ACCEL_LO], p[T2D_P_ACCEL_HI]};
    double beta[2], hr[2], sr[2], xr[2], yr[2];
    for (int i = 0; i < 2; ++i) {
        beta[i] = T_atan(k * st[i]);
        hr[i] = np_mod(lh + lv / wb * T_sin(beta[i]) * dt, TWO_PI);
        sr[i] = clip(lv + ac[i] * dt, p[T2D_P_SPEED_LO], p[T2D_P_SPEED_HI]);
        xr[i] = lx + sr[i] * T_cos(lh + beta[i]) * dt;
        yr[i] = ly + sr[i] * T_sin(lh + beta[i]) * dt;
    }
    if (hr[0] < hr[1] && !(hr[0] <= h && h <= hr[1])) return 0;
    if (hr[0] > hr[1] && !(hr[0] <= h || h <= hr[1])) return 0;
    if (!(sr[0] <= v && v <= sr[1])) return 0;
    if (!(xr[0] < x && x < xr[1]) || !(yr[0] < y && y < yr[1])) return 0;
    return 1;
}

/* batched: last state = (lx, ly, lh, lv, lvx, lvy)[n] fp32 as stored in the pool, candidate = (x, y, h, v)[n] */
void t2do_verify_batch(const double* rows, int row_stride, int n, const uint8_t* type_id, const float* lx,
                       const float* ly, const float* lh, const float* lv, const float* lvx, const float* lvy,
                       const float* x, const float* y, const float* h, const float* v, int interval_ms,
                       uint8_t* valid) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int i = 0; i < n; ++i)
        valid[i] = (uint8_t)t2do_verify_state(rows + (size_t)type_id[i] * row_stride, lx[i], ly[i], lh[i], lv[i],
                                              lvx[i], lvy[i], x[i], y[i], h[i], v[i], interval_ms);
}

/* ------------------------------------------------------------------------------------------
 * SingleTrackDrift (scope row f4): dynamic single-track model with Pacejka tyres.
 *   Tire constants                      physics/single_track_drift.py:16-49
 *   _pure_slip_longitudinal_tire_forces :183-201      _pure_slip_lateral_tire_forces :203-222
 *   _combined_slip_longitudinal_...     :224-250      _combined_slip_lateral_...     :252-289
 *   _tire_forces                        :291-344      _step :346-465      step :467-503
 * gamma (camber) is the literal 0 at every call site (:326-339), which the restatement folds:
 * mu_x = p_dx1, mu_y = p_dy1, S_hy = S_vy = 0, r_vy3 * gamma = 0.  Values that the reference
 * evaluates several times from the same argument (sin(beta), B*kappa_x ...) are computed once.
 * Quirks kept: d_phi / beta re-initialised per call (:364-365); the remainder sub-step IS
 * integrated (:356-359, unlike SingleTrackDynamics); dd_phi of the low-speed branch is dead.
 * ---------------------------------------------------------------------------------------- */
static const double TP_cx1 = 1.6411, TP_dx1 = 1.1739, TP_ex1 = 0.4640, TP_kx1 = 22.303, TP_hx1 = 1.2297e-3,
                    TP_vx1 = -8.8098e-6, TR_bx1 = 13.276, TR_bx2 = -13.778, TR_ex1 = 1.2568, TR_cx1 = 0.6522,
                    TR_hx1 = 5.0722e-3, TP_cy1 = 1.3507, TP_dy1 = 1.0489, TP_ey1 = -7.4722e-3, TP_ky1 = -21.920,
                    TR_by1 = 7.1433, TR_by2 = 9.1917, TR_by3 = -2.7856e-2, TR_cy1 = 1.0719, TR_ey1 = -0.2757,
                    TR_hy1 = 5.7448e-6, TR_vy1 = -2.7825e-2, TR_vy4 = 12.120, TR_vy5 = 1.9, TR_vy6 = -10.704;

static double safe_den(double u) { return fabs(u) > 1e-6 ? u : (u >= 0 ? 1e-6 : -1e-6); }

/* D * sin|cos( C * atan(B*s - E*(B*s - atan(B*s))) ): the magic-formula core shared by all four */
static double mf_angle(double B, double C, double E, double s) {
    const double bs = B * s;
    return C * T_atan(bs - E * (bs - T_atan(bs)));
}

static double pure_long(double kappa, double F_z) { /* :183-201, gamma = 0 */
    const double S_vx = TP_vx1 * F_z;
    const double kappa_x = -kappa + TP_hx1;
    const double D_x = TP_dx1 * F_z;
    const double B_x = (TP_kx1 * F_z) / (TP_cx1 * D_x + 1e-6);
    return D_x * T_sin(mf_angle(B_x, TP_cx1, TP_ex1, kappa_x) + S_vx);
}

static double pure_lat(double alpha, double F_z) { /* :203-222, gamma = 0: S_hy = S_vy = 0, mu_y = p_dy1 */
    const double alpha_y = alpha + 0.0;
    const double D_y = TP_dy1 * F_z;
    const double B_y = (TP_ky1 * F_z) / (TP_cy1 * D_y + 1e-6);
    return D_y * T_sin(mf_angle(B_y, TP_cy1, TP_ey1, alpha_y) + 0.0);
}

static double comb_long(double kappa, double alpha, double F0_x) { /* :224-250 */
    const double alpha_s = alpha + TR_hx1;
    const double B = TR_bx1 * T_cos(T_atan(TR_bx2 * kappa));
    const double D = F0_x / T_cos(mf_angle(B, TR_cx1, TR_ex1, TR_hx1));
    return D * T_cos(mf_angle(B, TR_cx1, TR_ex1, alpha_s));
}

static double comb_lat(double kappa, double alpha, double F_z, double F0_y) { /* :252-289, gamma = 0 */
    const double kappa_s = kappa + TR_hy1;
    const double B = TR_by1 * T_cos(T_atan(TR_by2 * (alpha - TR_by3)));
    const double D = F0_y / T_cos(mf_angle(B, TR_cy1, TR_ey1, TR_hy1));
    const double D_vy = TP_dy1 * F_z * TR_vy1 * T_cos(T_atan(TR_vy4 * alpha));
    const double S_vy = D_vy * T_sin(TR_vy5 * T_atan(TR_vy6 * kappa));
    return D * T_cos(mf_angle(B, TR_cy1, TR_ey1, kappa_s)) + S_vy;
}

/* out[8] = x, y, heading, speed, omega_wf, omega_wr, applied accel, applied steer */
void t2do_drift(const double* p, double x, double y, double phi, double v, double omega_wf, double omega_wr,
                double accel, double delta, int interval_ms, double* out) {
    const int flags = (int)p[T2D_P_RANGE_FLAGS];
    if (flags & T2D_RANGE_ACCEL) accel = clip(accel, p[T2D_P_ACCEL_LO], p[T2D_P_ACCEL_HI]); /* :495 */
    if (flags & T2D_RANGE_STEER) delta = clip(delta, p[T2D_P_STEER_LO], p[T2D_P_STEER_HI]); /* :496 */
    const double lf = p[T2D_P_LF], lr = p[T2D_P_LR], wb = p[T2D_P_WB], mass = p[T2D_P_MASS], Iz = p[T2D_P_IZ];
    const double radius = p[T2D_P_DRIFT_RADIUS], Tsb = p[T2D_P_DRIFT_TSB], Tse = p[T2D_P_DRIFT_TSE],
                 Iyw = p[T2D_P_DRIFT_IYW];
    const int delta_t = (int)p[T2D_P_DELTA_T_MS];
    const int n_steps = interval_ms / delta_t, rem = interval_ms % delta_t;
    const double tan_d = T_tan(delta), sin_d = T_sin(delta), cos_d = T_cos(delta);
    double d_phi = v / wb * tan_d;              /* :364 */
    double beta = T_atan(lr / lf * tan_d);      /* :365 */
    double T_B, T_E;
    if (accel > 0) { T_B = 0; T_E = mass * radius * accel; } else { T_B = mass * radius * accel; T_E = 0; }
    const double F_zf = (mass * 9.81 * lr) / wb, F_zr = (mass * 9.81 * lf) / wb; /* :308-309 */
    for (int k = 0; k < n_steps + (rem > 0 ? 1 : 0); ++k) {
        const double dt = k < n_steps ? (double)delta_t / 1000 : (double)rem / 1000;
        const double v_safe = safe_den(v); /* :376; _tire_forces applies the same guard again: idempotent */
        const double sin_b = T_sin(beta), cos_b = T_cos(beta);
        /* ---- _tire_forces :291-344 ---- */
        const double cos_b_safe = safe_den(cos_b);
        const double alpha_f = T_atan((v_safe * sin_b + d_phi * lf) / (v_safe * cos_b_safe)) - delta;
        const double alpha_r = T_atan((v_safe * sin_b - d_phi * lr) / (v_safe * cos_b_safe));
        const double u_wf = v_safe * cos_b_safe * cos_d + (v_safe * sin_b + lf * d_phi) * sin_d;
        const double u_wr = v_safe * cos_b_safe;
        const double s_f = 1 - radius * omega_wf / safe_den(u_wf);
        const double s_r = 1 - radius * omega_wr / safe_den(u_wr);
        const double F0_xf = pure_long(s_f, F_zf), F0_xr = pure_long(s_r, F_zr);
        const double F0_yf = pure_lat(alpha_f, F_zf), F0_yr = pure_lat(alpha_r, F_zr);
        const double F_lf = comb_long(s_f, alpha_f, F0_xf), F_lr = comb_long(s_r, alpha_r, F0_xr);
        const double F_sf = comb_lat(s_f, alpha_f, F_zf, F0_yf), F_sr = comb_lat(s_r, alpha_r, F_zr, F0_yr);
        /* ---- _step body ---- */
        const double dx = v * T_cos(phi + beta), dy = v * T_sin(phi + beta);
        double dv, d_beta, d_owf, d_owr;
        if (fabs(v) >= 0.1) { /* :384-419 */
            const double sdb = T_sin(delta - beta), cdb = T_cos(delta - beta);
            dv = 1 / mass * (-F_sf * sdb + F_sr * sin_b + F_lr * cos_b + F_lf * cdb);
            d_beta = -d_phi + 1 / (mass * v_safe) * (F_sf * cdb + F_sr * cos_b - F_lr * sin_b + F_lf * sdb);
            const double dd_phi = 1 / Iz * (F_sf * cos_d * lf - F_sr * lr + F_lf * sin_d * lf);
            d_phi += dd_phi * dt;
            d_owf = 1 / Iyw * (-radius * F_lf + Tsb * T_B + Tse * T_E);
            d_owr = 1 / Iyw * (-radius * F_lr + (1 - Tsb) * T_B + (1 - Tse) * T_E);
        } else { /* :420-451 */
            const double tb = 1 + tan_d * lr / wb;
            dv = accel;
            d_beta = lr / (tb * tb) / wb / (cos_d * cos_d) * delta;
            d_phi += v * cos_b / wb * tan_d * dt;
            d_owf = 1 / (cos_d * radius) * (accel * cos_b - v * sin_b * d_beta + v * cos_b * tan_d * delta);
            d_owr = 1 / radius * (accel * cos_b - v * sin_b * d_beta);
        }
        x += dx * dt;
        y += dy * dt;
        v += dv * dt;
        phi += d_phi * dt;
        beta += d_beta * dt;
        omega_wf += d_owf * dt;
        omega_wr += d_owr * dt;
        if (flags & T2D_RANGE_SPEED) v = clip(v, p[T2D_P_SPEED_LO], p[T2D_P_SPEED_HI]); /* :461 */
    }
    out[0] = x; out[1] = y; out[2] = np_mod(phi, TWO_PI); out[3] = v;
    out[4] = omega_wf; out[5] = omega_wr; out[6] = accel; out[7] = delta;
}

/* batched over the T2D_MODEL_DRIFT participants (others: state passed through, omegas untouched) */
void t2do_drift_batch(const double* rows, int row_stride, int n, const float* x, const float* y, const float* heading,
                      const float* speed, const float* omega_f, const float* omega_r, const float* act0,
                      const float* act1, const uint8_t* type_id, const uint8_t* active, int interval_ms, double* out) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int i = 0; i < n; ++i) {
        double* o = out + 8 * (size_t)i;
        const double* p = rows + (size_t)type_id[i] * row_stride;
        if ((active && !active[i]) || (int)p[T2D_P_MODEL] != T2D_MODEL_DRIFT) {
            o[0] = x[i]; o[1] = y[i]; o[2] = heading[i]; o[3] = speed[i];
            o[4] = omega_f[i]; o[5] = omega_r[i]; o[6] = 0.0; o[7] = 0.0;
            continue;
        }
        t2do_drift(p, x[i], y[i], heading[i], speed[i], omega_f[i], omega_r[i], act0[i], act1[i], interval_ms, o);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Row f4: ParkingLotGenerator.generate (map/generator/generate_parking_lot.py:239-444) restated per
 * env.  The reference draws from numpy's global MT19937 stream and evaluates its predicates in shapely/GEOS
 * (PINNED by replay on recorded draws: see the header), so this follows the reference's sampling
 * distributions, draw ORDER, geometric predicates and control flow -- including the two behaviours
 * that follow from its bookkeeping: side vehicles appended during rejected attempts stay in the
 * `obstacles` list (:283-285, :320-322), and Map.add_area keys areas by id so a later obstacle with
 * the same id replaces the earlier one (map.py:444-453; far wall "0003" vs left vehicle "0003") --
 * on a stream of its own:
 *     state0 = seed + (env + 1) * 0xD1B54A32D192ED03 ; every draw u = (splitmix64(&state) >> 11) * 2^-53
 *     uniform(a, b) = a + (b - a) * u ;  normal(m, s) = m + s * sqrt(-2 log(1 - u1)) * cos(2 pi u2)
 * The kernel in t2d_generate.hip implements the same specification and must agree bit for bit in
 * deterministic-trig mode.  Rejection loops are capped (T2D_GEN_MAX_ATTEMPTS / _START_ATTEMPTS); a
 * capped env is reported in `info`, never silently accepted.                                       */
#define PI_D 3.141592653589793
#define M_PI_2_D (PI_D / 2)
typedef struct { uint64_t s; } gen_rng;
/* REPLAY (tests only: t2do_generate_parking_replay): the draws come from a tape of the VALUES numpy handed the reference when
 * its own generate() ran (oracle/gen_golden_generator.py records every np.random call): kind 0 = a uniform in [0, 1)
 * (np.random.rand() / uniform()), 1 = np.random.uniform(a, b), 2 = np.random.normal(mean, std).  A draw of another kind than the
 * tape's next entry, or a draw past its end, is a desynchronisation: the restatement asked for its random numbers in another
 * ORDER than the reference.  Not thread safe: one env at a time. */
static const double* g_tape_val = NULL;
static const int32_t* g_tape_kind = NULL;
static int g_tape_n = 0, g_tape_pos = 0, g_tape_bad = -1;
static int tape_take(int kind, double* out) {
    if (!g_tape_val) return 0;
    if (g_tape_pos >= g_tape_n || g_tape_kind[g_tape_pos] != kind) {
        if (g_tape_bad < 0) g_tape_bad = g_tape_pos;
        *out = kind == 2 ? 0.0 : 0.5;
        if (g_tape_pos < g_tape_n) ++g_tape_pos;
        return 1;
    }
    *out = g_tape_val[g_tape_pos++];
    return 1;
}
static double gen_u(gen_rng* r) {
    double t;
    if (tape_take(0, &t)) return t;
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static double gen_uniform(gen_rng* r, double a, double b) {
    double t;
    if (tape_take(1, &t)) return t;
    return a + (b - a) * gen_u(r);
}
static double gen_normal(gen_rng* r, double mean, double std) {
    double t;
    if (tape_take(2, &t)) return t;
    double u1 = 1.0 - gen_u(r), u2 = gen_u(r);
    double rad = sqrt(-2.0 * (g_trig ? t2do_log(u1) : log(u1)));
    return mean + std * (rad * T_cos(TWO_PI * u2));
}
/* _truncate_gaussian :60-62 */
static double gen_tg(gen_rng* r, double mean, double std, double lo, double hi) {
    return clip(gen_normal(r, mean, std), lo, hi);
}
/* _get_bbox :64-87 (vertex order and the affine matrix [cos, -sin, sin, cos, cx, cy]) */
static void gen_bbox(double cx, double cy, double h, double len, double wid, double* q) {
    const double c = T_cos(h), s = T_sin(h);
    const double lx[4] = {0.5 * len, 0.5 * len, -0.5 * len, -0.5 * len};
    const double ly[4] = {-0.5 * wid, 0.5 * wid, 0.5 * wid, -0.5 * wid};
    for (int k = 0; k < 4; ++k) {
        q[2 * k] = (c * lx[k] + (-s) * ly[k]) + cx;
        q[2 * k + 1] = (s * lx[k] + c * ly[k]) + cy;
    }
}
/* _get_random_position :89-99; np.mean / np.std of the 2-tuples */
static void gen_random_position(gen_rng* r, const double* origin, double a0, double a1, double r0, double r1,
                                double* out) {
    double am = (a0 + a1) / 2.0, rm = (r0 + r1) / 2.0;
    double as = sqrt(((a0 - am) * (a0 - am) + (a1 - am) * (a1 - am)) / 2.0);
    double rs = sqrt(((r0 - rm) * (r0 - rm) + (r1 - rm) * (r1 - rm)) / 2.0);
    double angle = gen_tg(r, am, as, a0, a1);
    double radius = gen_tg(r, rm, rs, r0, r1);
    out[0] = origin[0] + radius * T_cos(angle);
    out[1] = origin[1] + radius * T_sin(angle);
}
static void gen_ccw(const double* q, double* o) { /* winding normalised for the SAT predicates */
    if (area2(q, 4) < 0.0) for (int k = 0; k < 4; ++k) { o[2 * k] = q[2 * (3 - k)]; o[2 * k + 1] = q[2 * (3 - k) + 1]; }
    else memcpy(o, q, 8 * sizeof(double));
}
static int gen_convex(const double* q) { /* q counter-clockwise: no right turn */
    for (int k = 0; k < 4; ++k)
        if (orient(q + 2 * k, q + 2 * ((k + 1) & 3), q + 2 * ((k + 2) & 3)) < 0.0) return 0;
    return 1;
}
static int gen_intersects(const double* a, const double* b) {
    double A[8], B[8];
    gen_ccw(a, A); gen_ccw(b, B);
    return t2do_convex_intersects(A, 4, B, 4);
}
/* shapely distance between two convex quads: 0 when they share a point, else the closest vertex-edge pair */
static double gen_distance(const double* a, const double* b) {
    if (gen_intersects(a, b)) return 0.0;
    double best = INFINITY;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double d1 = seg_dist2(b + 2 * j, b + 2 * ((j + 1) & 3), a + 2 * i);
            double d2 = seg_dist2(a + 2 * j, a + 2 * ((j + 1) & 3), b + 2 * i);
            best = fmin(best, fmin(d1, d2));
        }
    return sqrt(best);
}
#define GEN_LIST_CAP 24
typedef struct { double q[GEN_LIST_CAP][8]; int id[GEN_LIST_CAP]; int n; int overflow; } gen_list;
static void gen_append(gen_list* l, int id, const double* q) {
    if (l->n >= GEN_LIST_CAP) { l->overflow = 1; return; }
    memcpy(l->q[l->n], q, 8 * sizeof(double));
    l->id[l->n++] = id;
}
/* _get_side_vehicle :175-205 */
static void gen_side_vehicle(gen_rng* r, int bay, double len, double wid, double d0, double d1, int left, double* q) {
    double heading = bay ? gen_tg(r, M_PI_2_D, PI_D / 54, PI_D * 4 / 9, PI_D * 5 / 9)
                         : gen_tg(r, 0.0, PI_D / 54, -PI_D / 18, PI_D / 18);
    double side = left ? -1.0 : 1.0;
    double x = 0.0 + side * ((bay ? wid : len) + gen_uniform(r, d0, d1));
    double t[8];
    gen_bbox(x, 0.0, heading, len, wid, t);
    double m = bay ? fmin(t[7], t[5]) : fmin(t[7], t[1]); /* bottom_right, bottom_left | top_right */
    double min_y = -m + 0.8;
    double y = gen_tg(r, min_y + 0.4, 0.2, min_y, min_y + 0.8);
    gen_bbox(x, y, heading, len, wid, q);
}

void t2do_generate_parking(uint64_t seed, int64_t env0, int n_env, double type_proportion, double len, double wid,
                           float* quads /* [n_env][12][8] */, int32_t* quad_id /* [n_env][12] */, int32_t* n_quads,
                           double* start /* [n_env][3] */, float* target /* [n_env][8] */, double* target_heading,
                           float* boundary /* [n_env][4] */, uint32_t* info) {
    const double SIZE = 30.0, MARGIN = 13.0, D0 = 0.8, D1 = 1.6;
    if (!(type_proportion >= 0.0)) type_proportion = 0.0; /* np.clip(type_proportion, 0, 1) :57 */
    if (type_proportion > 1.0) type_proportion = 1.0;
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
    for (int e = 0; e < n_env; ++e) {
        gen_rng rng = {seed + (uint64_t)(env0 + e + 1) * 0xD1B54A32D192ED03ull};
        uint32_t flags = 0;
        const int bay = gen_u(&rng) < type_proportion; /* :256 */
        const double slot_len = bay ? 7.0 : 4.5;
        const double next = bay ? wid : len;
        const int n_more = bay ? (9 - 3) / 2 : (7 - 3) / 2;
        const double thr = bay ? 0.85 : 0.25 * len;
        gen_list L; L.n = 0; L.overflow = 0;
        double tq[8], th = 0.0, back[8], lo_[8], ro_[8];
        int attempts = 0, valid = 0;
        while (!valid) { /* :260-331 */
            ++attempts;
            /* _get_target_area :101-117 */
            th = bay ? gen_tg(&rng, M_PI_2_D, PI_D / 54, PI_D * 4 / 9, PI_D * 5 / 9)
                     : gen_tg(&rng, 0.0, PI_D / 54, -PI_D / 18, PI_D / 18);
            gen_bbox(0.0, 0.0, th, len, wid, tq);
            double y_min = -(bay ? fmin(tq[7], tq[5]) : fmin(tq[7], tq[1])) + D0;
            double cy = gen_tg(&rng, y_min + 0.4, 0.2, y_min, y_min + 0.8);
            gen_bbox(0.0, cy, th, len, wid, tq);
            /* _get_back_wall :119-125 */
            double ww = gen_uniform(&rng, 0.5, 1.5);
            gen_bbox(0.0, 0.0 - ww / 2, 0.0, SIZE, ww, back);
            double d0 = D0 + 0.1, d1 = D1;
            if (gen_u(&rng) < 0.2) { /* _get_left_wall :127-149 */
                double a[2], b[2];
                gen_random_position(&rng, bay ? tq + 2 : tq + 4, PI_D * 11 / 12, PI_D * 13 / 12, d0, d1, a);
                gen_random_position(&rng, bay ? tq + 4 : tq + 6, PI_D * 11 / 12, PI_D * 13 / 12, d0, d1, b);
                lo_[0] = a[0]; lo_[1] = a[1]; lo_[2] = b[0]; lo_[3] = b[1];
                lo_[4] = 0.0 - SIZE / 2; lo_[5] = 0.0; lo_[6] = 0.0 - SIZE / 2; lo_[7] = a[1];
            } else {
                gen_side_vehicle(&rng, bay, len, wid, d0, d1, 1, lo_);
                for (int i = 0; i < n_more; ++i) {
                    d0 += next + D0; d1 += next + D0;
                    double q[8];
                    gen_side_vehicle(&rng, bay, len, wid, d0, d1, 1, q);
                    gen_append(&L, 2 * i + 3, q);
                }
            }
            double dl = gen_distance(tq, lo_);
            d0 = fmax(thr - dl, 0.0) + D0; d1 = D1;
            if (gen_u(&rng) < 0.2) { /* _get_right_wall :151-173 */
                double a[2], b[2];
                gen_random_position(&rng, bay ? tq + 6 : tq + 0, -PI_D * 1 / 12, PI_D * 1 / 12, d0, d1, a); /* bottom left */
                gen_random_position(&rng, bay ? tq + 0 : tq + 2, -PI_D * 1 / 12, PI_D * 1 / 12, d0, d1, b); /* top left */
                ro_[0] = 0.0 + SIZE / 2; ro_[1] = tq[3]; ro_[2] = 0.0 + SIZE / 2; ro_[3] = 0.0;
                ro_[4] = a[0]; ro_[5] = a[1]; ro_[6] = b[0]; ro_[7] = b[1];
            } else {
                gen_side_vehicle(&rng, bay, len, wid, d0, d1, 0, ro_);
                for (int i = 0; i < n_more; ++i) {
                    d0 += next + D0; d1 += next + D0;
                    double q[8];
                    gen_side_vehicle(&rng, bay, len, wid, d0, d1, 0, q);
                    gen_append(&L, 2 * i + 4, q);
                }
            }
            double dr = gen_distance(tq, ro_);
            /* _verify_obstacles :207-223 (`any(dists) < 0.8` compares a bool: it only rejects dl == dr == 0) */
            valid = !(gen_intersects(tq, back) || gen_intersects(tq, lo_) || gen_intersects(tq, ro_));
            if (valid && !(dl != 0.0 || dr != 0.0)) valid = 0;
            if (valid && dl + dr < thr) valid = 0;
            if (!valid && attempts >= T2D_GEN_MAX_ATTEMPTS) { flags |= T2D_GEN_UNVERIFIED; break; }
        }
        gen_append(&L, 0, back); gen_append(&L, 1, lo_); gen_append(&L, 2, ro_);
        double y_max = -INFINITY; /* :338-346 */
        for (int i = 0; i < L.n; ++i)
            for (int k = 0; k < 4; ++k) y_max = fmax(y_max, L.q[i][2 * k + 1]);
        y_max += D0;
        if (gen_u(&rng) < 0.2) { /* far wall :347-356 */
            double w = gen_uniform(&rng, 0.0, 0.2), q[8];
            gen_bbox(0.0, y_max + slot_len, 0.0, SIZE, w, q);
            gen_append(&L, 3, q);
        } else { /* three perturbed vehicles beyond the start range :357-387 */
            const double bx0 = 0.0 - SIZE / 2, bx1 = 0.0 + SIZE / 2;
            const double yc = y_max + slot_len + 4;
            double bb[8];
            gen_bbox(0.0, yc, 0.0, SIZE, 8.0, bb);
            const double y0 = y_max + slot_len + 2, y1 = y_max + slot_len + 6;
            int id = L.n + 1;
            for (int t = 0; t < 3; ++t) {
                double x = gen_uniform(&rng, bx0, bx1);
                double y = gen_uniform(&rng, y0, y1);
                double h = gen_u(&rng) * 2 * PI_D;
                double q[8];
                gen_bbox(x, y, h, len, wid, q);
                int inside = 1;
                for (int k = 0; k < 8; ++k) q[k] = q[k] + 0.5 * gen_u(&rng);
                for (int k = 0; k < 4; ++k) /* Polygon(bbox).contains(shape): closed rectangle test on the vertices */
                    if (!(q[2 * k] >= bb[4] && q[2 * k] <= bb[0] && q[2 * k + 1] >= bb[1] && q[2 * k + 1] <= bb[3])) inside = 0;
                if (inside) { gen_append(&L, id, q); ++id; }
            }
        }
        { /* random drop :389-390 */
            int m = 0;
            for (int i = 0; i < L.n; ++i)
                if (gen_u(&rng) >= 0.05) { if (m != i) { memcpy(L.q[m], L.q[i], sizeof L.q[0]); L.id[m] = L.id[i]; } ++m; }
            L.n = m;
        }
        for (int i = 0; i < L.n; ++i) { /* every obstacle must be usable by the convex predicates */
            double c[8];
            gen_ccw(L.q[i], c);
            if (!gen_convex(c)) flags |= T2D_GEN_NONCONVEX;
        }
        /* start state :396-407, _get_start_state :225-229, _verify_start_state :231-237 */
        double sx = 0.0, sy = 0.0, sh = 0.0;
        int s_attempts = 0;
        for (;;) {
            ++s_attempts;
            sx = gen_uniform(&rng, -SIZE / 4, SIZE / 4);
            sy = gen_uniform(&rng, y_max + D0 + 1, y_max + slot_len - 1);
            sh = gen_tg(&rng, 0.0, PI_D / 54, -PI_D / 18, PI_D / 18);
            double sq[8];
            gen_bbox(sx, sy, sh, len, wid, sq);
            int ok = 1;
            for (int i = 0; i < L.n && ok; ++i) if (gen_intersects(sq, L.q[i])) ok = 0;
            if (ok && gen_intersects(sq, tq)) ok = 0;
            if (ok) break;
            if (s_attempts >= T2D_GEN_MAX_START_ATTEMPTS) { flags |= T2D_GEN_START_UNVERIFIED; break; }
        }
        /* flip :409-434 */
        double tx = (((tq[0] + tq[2]) + tq[4]) + tq[6]) / 4.0, ty = (((tq[1] + tq[3]) + tq[5]) + tq[7]) / 4.0;
        if (gen_u(&rng) > 0.5) {
            double sq[8];
            gen_bbox(sx, sy, sh, len, wid, sq);
            double cx = (((sq[0] + sq[2]) + sq[4]) + sq[6]) / 4.0, cyy = (((sq[1] + sq[3]) + sq[5]) + sq[7]) / 4.0;
            sx = 2 * cx - sx; sy = 2 * cyy - sy; sh += PI_D;
            if (!bay) { th += PI_D; gen_bbox(tx, ty, th, len, wid, tq); flags |= T2D_GEN_TARGET_FLIPPED; }
            flags |= T2D_GEN_START_FLIPPED;
        }
        /* Map.add_area in list order: same id replaces, position of the first insertion (dict semantics) */
        int n_out = 0; int ids[T2D_GEN_MAX_QUADS]; double outq[T2D_GEN_MAX_QUADS][8];
        for (int i = 0; i < L.n; ++i) {
            int at = -1;
            for (int k = 0; k < n_out; ++k) if (ids[k] == L.id[i]) at = k;
            if (at < 0) { if (n_out >= T2D_GEN_MAX_QUADS) { flags |= T2D_GEN_OVERFLOW; continue; } at = n_out++; ids[at] = L.id[i]; }
            memcpy(outq[at], L.q[i], sizeof outq[0]);
        }
        if (L.overflow) flags |= T2D_GEN_OVERFLOW;
        for (int k = 0; k < T2D_GEN_MAX_QUADS; ++k)
            for (int c = 0; c < 8; ++c) quads[((size_t)e * T2D_GEN_MAX_QUADS + k) * 8 + c] = k < n_out ? (float)outq[k][c] : 0.0f;
        for (int k = 0; k < T2D_GEN_MAX_QUADS; ++k) quad_id[(size_t)e * T2D_GEN_MAX_QUADS + k] = k < n_out ? ids[k] : -1;
        n_quads[e] = n_out;
        start[3 * (size_t)e] = sx; start[3 * (size_t)e + 1] = sy; start[3 * (size_t)e + 2] = sh;
        for (int c = 0; c < 8; ++c) target[8 * (size_t)e + c] = (float)tq[c];
        target_heading[e] = th;
        /* :436-440 */
        boundary[4 * (size_t)e] = (float)floor(fmin(sx, tx) - MARGIN);
        boundary[4 * (size_t)e + 1] = (float)ceil(fmax(sx, tx) + MARGIN);
        boundary[4 * (size_t)e + 2] = (float)floor(fmin(sy, ty) - MARGIN);
        boundary[4 * (size_t)e + 3] = (float)ceil(fmax(sy, ty) + MARGIN);
        info[e] = flags | (bay ? T2D_GEN_BAY : 0u) | ((uint32_t)(attempts > 255 ? 255 : attempts) << 8) |
                  ((uint32_t)(s_attempts > 255 ? 255 : s_attempts) << 16);
    }
}

/* One scene from a tape of recorded draws (see g_tape_val above).  Returns the number of draws consumed; *desync = index of the
 * first draw that did not match the tape's kind or ran past its end, -1 = none. */
int t2do_generate_parking_replay(const double* tape_val, const int32_t* tape_kind, int n_tape, double type_proportion, double len,
                                 double wid, float* quads, int32_t* quad_id, int32_t* n_quads, double* start, float* target,
                                 double* target_heading, float* boundary, uint32_t* info, int32_t* desync) {
    const int threads = g_threads;
    g_threads = 1;
    g_tape_val = tape_val; g_tape_kind = tape_kind; g_tape_n = n_tape; g_tape_pos = 0; g_tape_bad = -1;
    t2do_generate_parking(0, 0, 1, type_proportion, len, wid, quads, quad_id, n_quads, start, target, target_heading, boundary, info);
    const int used = g_tape_pos;
    if (desync) *desync = g_tape_bad;
    g_tape_val = NULL; g_tape_kind = NULL; g_tape_n = 0;
    g_threads = threads;
    return used;
}

int t2do_abi_version(void) { return T2D_ABI_VERSION; }
