// t2d_math.h -- device-side fp64 helpers for the gfx950 kernels.
//
// Everything here is built from IEEE-754 +,-,*,/ , sqrt, rint and explicit fma only, and the
// translation unit is compiled with -ffp-contract=off, so results are bit-reproducible and
// equal to the C restatement of the same spec in oracle/t2d_oracle.c (DESIGN.md,
// "Deterministic trig").  No ocml transcendental is used on any path that feeds a flag.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define T2D_DEV __device__ __forceinline__
// pointer into device memory proper (global_load / global_store, not the generic flat forms)
#define T2D_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T2D_GLOBAL T* as_global(T* p) { return (T2D_GLOBAL T*)p; }

// a load of state that an earlier step of the SAME launch stored (t2d_step_n): relaxed, agent scope = `sc1`, served by
// the L2 and never by this CU's L1; a plain load otherwise
template <bool MULTI, class T>
__device__ __forceinline__ T ld_state(const T2D_GLOBAL T* p) {
    if (MULTI) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

namespace t2d {

constexpr double kTwoPi = 2.0 * 3.141592653589793;
constexpr double kPio2Hi = 1.5707963267948966;
constexpr double kPio2Mid = 6.123233995736766e-17;
constexpr double kPio2Lo = -1.4973849048591698e-33;
constexpr double kTwoOverPi = 0.6366197723675814;

#ifdef T2D_TRIG_TABLE
// (defined by a translation unit before it includes this header: t2d_collide.hip)
// The fp64 constants of sincos_det / atan_det as tables in constant memory: as literals each of them costs two move
// instructions every time the function runs (an fp64 literal cannot be an instruction operand), a table row comes in with
// one scalar load for eight of them.  Same values, same operations, same bits.  (External linkage on purpose: a static
// table that nothing writes is folded back into literals by the compiler.)
__constant__ double kSinCosTab[16] = {
    0.6366197723675814, 1.5707963267948966, 6.123233995736766e-17, -1.4973849048591698e-33,
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,
    8.33333333332248946124e-03, -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05,
    -1.38888888888741095749e-03, 4.16666666666666019037e-02};
__constant__ double kAtanTab[12] = {
    3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02, 0.0};
#endif
// k mod 4 of an integral-valued double (the quadrant of the Cody-Waite quotient), through the 64-bit conversion (trunc, ldexp,
// floor, fma, cvt: five instructions).  Round 4 tried the one-instruction 32-bit conversion behind a wave-uniform test of
// |k| < 2^31: four instructions fewer, and 6 % SLOWER on the pools that run one or two waves per SIMD (cfg3 as t2d_step_n
// fragments 8.4 -> 9.2 us per step, scripts/ab_step.py) -- a ballot and a branch on a lone wave's dependent chain cost more
// than the instructions they skip.
T2D_DEV int quadrant_of(double k) { return (int)((long long)k & 3); }

// sin/cos of x: 3-term Cody-Waite reduction by pi/2 (fma), minimax kernels on [-pi/4, pi/4].
// <= 1 ulp against libm for |x| < ~1e5 (tests/test_oracle_geometry.py, tests/test_gpu_math.py).
T2D_DEV void sincos_det(double x, double& s_out, double& c_out) {
#ifdef T2D_TRIG_TABLE
    const double* T = kSinCosTab;
    double k = __builtin_rint(x * T[0]);
    double r = __builtin_fma(-k, T[1], x);
    r = __builtin_fma(-k, T[2], r);
    r = __builtin_fma(-k, T[3], r);
    double z = r * r;
    double ps = T[4];
    ps = __builtin_fma(ps, z, T[5]);
    ps = __builtin_fma(ps, z, T[6]);
    ps = __builtin_fma(ps, z, T[7]);
    ps = __builtin_fma(ps, z, T[8]);
    ps = __builtin_fma(ps, z, T[9]);
    double sr = __builtin_fma(r * z, ps, r);
    double pc = T[10];
    pc = __builtin_fma(pc, z, T[11]);
    pc = __builtin_fma(pc, z, T[12]);
    pc = __builtin_fma(pc, z, T[13]);
    pc = __builtin_fma(pc, z, T[14]);
    pc = __builtin_fma(pc, z, T[15]);
#else
    double k = __builtin_rint(x * kTwoOverPi);
    double r = __builtin_fma(-k, kPio2Hi, x);
    r = __builtin_fma(-k, kPio2Mid, r);
    r = __builtin_fma(-k, kPio2Lo, r);
    double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    double sr = __builtin_fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
#endif
    double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    // the quadrant = k mod 4: through the one-instruction 32-bit conversion when every lane's quotient fits (|x| < 3e9:
    // always, short of an integrator that has blown up), the 64-bit one -- four instructions -- otherwise; same bits
    const int quad = quadrant_of(k);
    double s = (quad & 1) ? cr : sr;
    double c = (quad & 1) ? sr : cr;
    s_out = (quad & 2) ? -s : s;
    c_out = ((quad + 1) & 2) ? -c : c;
}

// sincos_det for |x| <= pi/4 in EVERY lane of the wave (caller checks with a ballot): the reduction quotient is 0,
// r = x exactly, so the kernels alone give the very same bits without the rint / fma / quadrant selects.
T2D_DEV void sincos_det_small(double r, double& s_out, double& c_out) {
    double z = r * r;
#ifdef T2D_TRIG_TABLE
    const double* T = kSinCosTab;
    double ps = T[4];
    ps = __builtin_fma(ps, z, T[5]);
    ps = __builtin_fma(ps, z, T[6]);
    ps = __builtin_fma(ps, z, T[7]);
    ps = __builtin_fma(ps, z, T[8]);
    ps = __builtin_fma(ps, z, T[9]);
    s_out = __builtin_fma(r * z, ps, r);
    double pc = T[10];
    pc = __builtin_fma(pc, z, T[11]);
    pc = __builtin_fma(pc, z, T[12]);
    pc = __builtin_fma(pc, z, T[13]);
    pc = __builtin_fma(pc, z, T[14]);
    pc = __builtin_fma(pc, z, T[15]);
#else
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    s_out = __builtin_fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
#endif
    c_out = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
}
// steering angles: |delta| <= 0.7 rad practically always
T2D_DEV void sincos_det_steer(double x, double& s_out, double& c_out) {
    if (__ballot(!(__builtin_fabs(x) <= 0.78)) == 0ull) sincos_det_small(x, s_out, c_out);
    else sincos_det(x, s_out, c_out);
}

// sincos_det of TWO angles in one pass -- the steering angle (small in every lane: see sincos_det_steer; else the general
// reduction) and the heading: the two polynomial evaluations interleaved, so that each table constant is an operand of both
// while it sits in its scalar register.  Evaluated one after the other the compiler parks the sixteen constants in vector
// registers between the two calls: sixteen 64-bit moves per wave and call site.  Same operations per angle, same bits.
T2D_DEV void sincos_det_steer_and(double xa, double xb, double& sa_out, double& ca_out, double& sb_out, double& cb_out) {
#ifdef T2D_TRIG_TABLE
    const double* T = kSinCosTab;
    const double t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3], p0 = T[4], p1 = T[5], p2 = T[6], p3 = T[7], p4 = T[8], p5 = T[9],
                 q0 = T[10], q1 = T[11], q2 = T[12], q3 = T[13], q4 = T[14], q5 = T[15];
#else
    const double t0 = kTwoOverPi, t1 = kPio2Hi, t2 = kPio2Mid, t3 = kPio2Lo;
    const double p0 = 1.58969099521155010221e-10, p1 = -2.50507602534068634195e-08, p2 = 2.75573137070700676789e-06,
                 p3 = -1.98412698298579493134e-04, p4 = 8.33333333332248946124e-03, p5 = -1.66666666666666324348e-01;
    const double q0 = -1.13596475577881948265e-11, q1 = 2.08757232129817482790e-09, q2 = -2.75573143513906633035e-07,
                 q3 = 2.48015872894767294178e-05, q4 = -1.38888888888741095749e-03, q5 = 4.16666666666666019037e-02;
#endif
    const bool small_a = __ballot(!(__builtin_fabs(xa) <= 0.78)) == 0ull;   // wave-uniform
    double ka = 0.0, ra = xa;
    if (__builtin_expect(!small_a, 0)) {
        asm volatile("" : "+v"(xa));   // (keeps the rare side a real branch: left alone the compiler evaluates the reduction and selects)
        ka = __builtin_rint(xa * t0);
        ra = __builtin_fma(-ka, t1, xa);
        ra = __builtin_fma(-ka, t2, ra);
        ra = __builtin_fma(-ka, t3, ra);
    }
    const double kb = __builtin_rint(xb * t0);
    double rb = __builtin_fma(-kb, t1, xb);
    rb = __builtin_fma(-kb, t2, rb);
    rb = __builtin_fma(-kb, t3, rb);
    const double za = ra * ra, zb = rb * rb;
    double psa = p0, psb = p0;
    psa = __builtin_fma(psa, za, p1); psb = __builtin_fma(psb, zb, p1);
    psa = __builtin_fma(psa, za, p2); psb = __builtin_fma(psb, zb, p2);
    psa = __builtin_fma(psa, za, p3); psb = __builtin_fma(psb, zb, p3);
    psa = __builtin_fma(psa, za, p4); psb = __builtin_fma(psb, zb, p4);
    psa = __builtin_fma(psa, za, p5); psb = __builtin_fma(psb, zb, p5);
    const double sra = __builtin_fma(ra * za, psa, ra), srb = __builtin_fma(rb * zb, psb, rb);
    double pca = q0, pcb = q0;
    pca = __builtin_fma(pca, za, q1); pcb = __builtin_fma(pcb, zb, q1);
    pca = __builtin_fma(pca, za, q2); pcb = __builtin_fma(pcb, zb, q2);
    pca = __builtin_fma(pca, za, q3); pcb = __builtin_fma(pcb, zb, q3);
    pca = __builtin_fma(pca, za, q4); pcb = __builtin_fma(pcb, zb, q4);
    pca = __builtin_fma(pca, za, q5); pcb = __builtin_fma(pcb, zb, q5);
    const double cra = __builtin_fma(za * za, pca, __builtin_fma(-0.5, za, 1.0));
    const double crb = __builtin_fma(zb * zb, pcb, __builtin_fma(-0.5, zb, 1.0));
    auto finish = [](double k, double sr, double cr, double& s_out, double& c_out) {
        const int quad = quadrant_of(k);
        const double s = (quad & 1) ? cr : sr;
        const double c = (quad & 1) ? sr : cr;
        s_out = (quad & 2) ? -s : s;
        c_out = ((quad + 1) & 2) ? -c : c;
    };
    if (small_a) {   // (quotient 0: the kernels alone are sincos_det's result)
        sa_out = sra;
        ca_out = cra;
    } else {
        finish(ka, sra, cra, sa_out, ca_out);
    }
    finish(kb, srb, crb, sb_out, cb_out);
}

T2D_DEV double tan_det(double x) {
    double s, c;
    sincos_det(x, s, c);
    return s / c;
}

// atan: 4-breakpoint reduction + odd polynomial (no fma; see oracle t2do_atan).
T2D_DEV double atan_det(double x) {
#ifdef T2D_TRIG_TABLE
    const double* AT = kAtanTab;
    const double aT0 = AT[0], aT1 = AT[1], aT2 = AT[2], aT3 = AT[3], aT4 = AT[4], aT5 = AT[5], aT6 = AT[6], aT7 = AT[7],
                 aT8 = AT[8], aT9 = AT[9], aT10 = AT[10];
#else
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
#endif
    bool neg = x < 0.0;
    double ax = __builtin_fabs(x);
    if (ax != ax) return x;
    if (ax >= 1.8014398509481984e16) {
        double r = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
        return neg ? -r : r;
    }
    int id;
    double hi = 0.0, lo = 0.0;
    if (ax < 0.4375) {
        if (ax < 7.450580596923828e-09) return x;
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) {
            id = 0; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17;
            ax = (2.0 * ax - 1.0) / (2.0 + ax);
        } else {
            id = 1; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17;
            ax = (ax - 1.0) / (ax + 1.0);
        }
    } else {
        if (ax < 2.4375) {
            id = 2; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17;
            ax = (ax - 1.5) / (1.0 + 1.5 * ax);
        } else {
            id = 3; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17;
            ax = -1.0 / ax;
        }
    }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    double r;
    if (id < 0) r = ax - ax * (s1 + s2);
    else r = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -r : r;
}

// atan2 on top of atan_det (oracle t2do_atan2).  atan2(0, 0) = 0 like numpy.
T2D_DEV double atan2_det(double y, double x) {
    const double pi = 3.141592653589793;
    const double pio2 = 1.5707963267948966;
    if (x != x || y != y) return x + y;
    if (y == 0.0) {
        if (x > 0.0 || (x == 0.0 && !__builtin_signbit(x))) return y;  // +-0
        return __builtin_signbit(y) ? -pi : pi;
    }
    if (x == 0.0) return y > 0.0 ? pio2 : -pio2;
    double a = atan_det(y / x);
    if (x > 0.0) return a;
    return y > 0.0 ? a + pi : a - pi;
}

T2D_DEV double clipd(double v, double lo, double hi) {  // np.clip
    double t = v < lo ? lo : v;
    return t > hi ? hi : t;
}

// np.mod(phi, 2*pi): exact remainder, equal to numpy's fmod-then-shift (one rounding only when
// the shifted value is not representable, exactly as in numpy).  Fast path: quotient estimate
// + fma (signs of fma results are exact, so an off-by-one estimate is repaired); beyond 1e9 rad
// (only reachable when the reference's own integrator has blown up) the estimate can be off by
// more than one and the exact library fmod takes over.
T2D_DEV double mod_two_pi(double phi) {
    if (!(__builtin_fabs(phi) < 1e9)) {
        double m = fmod(phi, kTwoPi);
        if (m != 0.0) {
            if (m < 0.0) m += kTwoPi;
        } else {
            m = 0.0;
        }
        return m;
    }
    // k_true = the largest integer k with phi - k*2pi >= 0.  The estimate only has to be within one
    // of k_true (the fma residuals below are exact in sign and repair it), so a multiplication by
    // 1/(2 pi) replaces the IEEE division: |phi| < 1e9 keeps its error below 1e-6.
    double k = __builtin_floor(phi * 0.15915494309189535);
    double r = __builtin_fma(-k, kTwoPi, phi);
    if (r < 0.0) {
        r = __builtin_fma(-(k - 1.0), kTwoPi, phi);
    } else {
        double r1 = __builtin_fma(-(k + 1.0), kTwoPi, phi);
        if (r1 >= 0.0) r = r1;
    }
    return r;
}

// ---- exp / log / pow for the IDM controller's (v / v_des) ** delta (oracle: t2do_exp/log/pow) ----
// Public fdlibm formulations (e_log.c / e_exp.c), written without fma so that the C oracle computes the
// same bits with -ffp-contract=off.  < 1 ulp each; pow_det = exp(y log x) is good to ~|y log x| ulp,
// integer-valued exponents up to 64 use a fixed square-and-multiply chain instead.
T2D_DEV double log_det(double x) {  // x > 0, finite
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int e;
    double m = __builtin_frexp(x, &e);  // [0.5, 1)
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

T2D_DEV double exp_det(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893384) return __builtin_inf();
    if (x < -745.1332191019411) return 0.0;
    const double k = __builtin_rint(x * invln2);
    const double hi = x - k * ln2HI;
    const double lo = k * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return __builtin_ldexp(y, (int)k);
}

T2D_DEV double pow_det(double x, double y) {  // float.__pow__ on the IDM path (idm_controller.py:79,129)
    if (y == 0.0) return 1.0;
    if (x != x || y != y) return x + y;
    const double yi = __builtin_rint(y);
    if (yi == y && __builtin_fabs(y) <= 64.0) {
        int n = (int)__builtin_fabs(yi);
        double r = 1.0, b = x;
        while (n) {
            if (n & 1) r = r * b;
            b = b * b;
            n >>= 1;
        }
        return y < 0.0 ? 1.0 / r : r;
    }
    if (x < 0.0) return __builtin_nan("");  // Python yields a complex number: outside the contract
    if (x == 0.0) return y > 0.0 ? 0.0 : __builtin_inf();
    return exp_det(y * log_det(x));
}

}  // namespace t2d
