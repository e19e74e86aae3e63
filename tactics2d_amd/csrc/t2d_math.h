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

namespace t2d {

constexpr double kTwoPi = 2.0 * 3.141592653589793;
constexpr double kPio2Hi = 1.5707963267948966;
constexpr double kPio2Mid = 6.123233995736766e-17;
constexpr double kPio2Lo = -1.4973849048591698e-33;
constexpr double kTwoOverPi = 0.6366197723675814;

// sin/cos of x: 3-term Cody-Waite reduction by pi/2 (fma), minimax kernels on [-pi/4, pi/4].
// <= 1 ulp against libm for |x| < ~1e5 (tests/test_oracle_geometry.py, tests/test_gpu_math.py).
T2D_DEV void sincos_det(double x, double& s_out, double& c_out) {
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
    double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    long long q = (long long)k;
    int quad = (int)(q & 3);
    double s = (quad & 1) ? cr : sr;
    double c = (quad & 1) ? sr : cr;
    s_out = (quad & 2) ? -s : s;
    c_out = ((quad + 1) & 2) ? -c : c;
}

T2D_DEV double tan_det(double x) {
    double s, c;
    sincos_det(x, s, c);
    return s / c;
}

// atan: 4-breakpoint reduction + odd polynomial (no fma; see oracle t2do_atan).
T2D_DEV double atan_det(double x) {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
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

}  // namespace t2d
