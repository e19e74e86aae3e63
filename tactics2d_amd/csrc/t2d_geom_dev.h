// t2d_geom_dev.h -- convex-quad predicates shared by the event kernels (t2d_collide.hip: one lane per participant;
// t2d_ego.hip: one wave per single-ego environment).  Every function is the arithmetic of oracle/t2d_oracle.c, operation
// by operation (-ffp-contract=off).
#pragma once
#include "t2d_math.h"

namespace t2d {
namespace geom {

T2D_DEV double orient(double px, double py, double qx, double qy, double rx, double ry) {
    double a = qx - px, b = ry - py;
    double c = qy - py, d = rx - px;
    return a * b - c * d;
}

// A convex quadrilateral (or triangle padded by repeating vertex 0) in registers.  Padding never
// changes a predicate: duplicate vertices repeat an existing test, padded edges are zero-length
// (orientation exactly 0: never separating, never "outside").
struct Quad {
    double x[4], y[4];
};

T2D_DEV Quad load_quad_f32(const float* p, int n) {  // interleaved x,y fp32 in LDS, n = 3 or 4
    Quad r;
    const float2* q = reinterpret_cast<const float2*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 v = q[j < n ? j : 0];
        r.x[j] = (double)v.x;
        r.y[j] = (double)v.y;
    }
    return r;
}

// closed-set convex `intersects` of two quads: the orientation evaluations of oracle
// t2do_convex_intersects(A, 4, B, n) (plus harmless padded ones).
T2D_DEV bool sat_quads(const Quad& A, const Quad& B) {
    // straight-line: the lanes of a wave hold different candidate pairs, so an early return on the first separating
    // edge only saves work when all 64 agree; without the exits the 32 orientation signs are independent
    bool separated = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (i + 1) & 3;
        bool all_out = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) all_out &= orient(A.x[i], A.y[i], A.x[k], A.y[k], B.x[j], B.y[j]) < 0.0;
        separated |= all_out;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (j + 1) & 3;
        bool all_out = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) all_out &= orient(B.x[j], B.y[j], B.x[k], B.y[k], A.x[i], A.y[i]) < 0.0;
        separated |= all_out;
    }
    return !separated;
}

T2D_DEV bool point_in_quad(const Quad& B, double x, double y) {
    bool in = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (j + 1) & 3;
        in &= !(orient(B.x[j], B.y[j], B.x[k], B.y[k], x, y) < 0.0);
    }
    return in;
}

// ---- IoU of two convex quads (Arrival / NoAction), oracle t2do_quad_iou: the boundary of A n B is
// integrated directly -- every edge of A clipped to closed B, every edge of B clipped to A with
// coincident (parallel, on-the-line) pieces dropped -- and the 8 partial sums are combined in a
// fixed tree order.  Out of line: only the ego lane of an env runs it.
// Branch-free on purpose: the four clip parameters of a term -- and the 8 terms of an IoU -- are independent IEEE
// divisions; written with if / else every one of them sat in its own basic block and the single ego lane walked
// 33 divisions one after the other.  A division by den == 0 yields inf / nan that is never selected.
T2D_DEV double clipped_edge_term(double p0x, double p0y, double p1x, double p1y, const Quad& Q, bool strict,
                                 double Ox, double Oy) {
    const double dx = p1x - p0x, dy = p1y - p0y;
    double t0 = 0.0, t1 = 1.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (j + 1) & 3;
        const double ex = Q.x[k] - Q.x[j], ey = Q.y[k] - Q.y[j];
        const double num = ex * (p0y - Q.y[j]) - ey * (p0x - Q.x[j]);
        const double den = ex * dy - ey * dx;
        const double tc = -num / den;
        const bool par = den == 0.0;   // (bitwise, not short-circuit: no control flow)
        ok = ok & !(par & ((num < 0.0) | (strict & (num == 0.0))));
        t0 = (!par & (den > 0.0) & (tc > t0)) ? tc : t0;
        t1 = (!par & (den < 0.0) & (tc < t1)) ? tc : t1;
    }
    const double ax = p0x + t0 * dx - Ox, ay = p0y + t0 * dy - Oy;
    const double bx = p0x + t1 * dx - Ox, by = p0y + t1 * dy - Oy;
    const double term = ax * by - bx * ay;
    return (ok & (t0 < t1)) ? term : 0.0;
}

T2D_DEV double quad_area2(const Quad& P) {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (i + 1) & 3;
        a += (P.x[i] - P.x[0]) * (P.y[k] - P.y[0]) - (P.x[k] - P.x[0]) * (P.y[i] - P.y[0]);
    }
    return a;
}

}  // namespace geom
}  // namespace t2d
