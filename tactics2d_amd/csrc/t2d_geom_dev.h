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

// Certifying filter in front of sat_quads for two RECTANGLES (participant boxes: vertex order front-right, front-left,
// rear-left, rear-right, participant/element/vehicle.py:132-142).  Two rectangles are disjoint iff one of their four edge
// directions separates them, so four projections decide what sat_quads decides with 32 orientations -- where the answer is
// not a matter of rounding.  With the full edge vectors P = v0 - v3 (length), Q = v1 - v0 (width) and D = twice the centre
// offset, the gap along n is  |n.D| - (|n.PA| + |n.QA| + |n.PB| + |n.QB|) = 2 |n| x (distance between the projections).
// Returns 0: separated by more than kRectMargin / (2 |n|) >= 2.5e-8 m along some edge direction (the edge of that box
// facing the other one then has all four vertices of the other strictly outside by that much: sat_quads finds it);
// 1: the projections overlap by more than that on all four (every edge of either box then has a vertex of the other box
// inside by that much: sat_quads finds no separating edge); 2: closer to touching than that -- undecided, the caller runs
// sat_quads.  The rounding of either evaluation is ~1e-11 m, three orders below the margin, so 0 / 1 are sat_quads' answers.
constexpr double kRectMargin = 1e-6;
T2D_DEV int rect_pair_filter(const Quad& A, const Quad& B) {
    const double pax = A.x[0] - A.x[3], pay = A.y[0] - A.y[3], qax = A.x[1] - A.x[0], qay = A.y[1] - A.y[0];
    const double pbx = B.x[0] - B.x[3], pby = B.y[0] - B.y[3], qbx = B.x[1] - B.x[0], qby = B.y[1] - B.y[0];
    const double dx = (B.x[0] + B.x[2]) - (A.x[0] + A.x[2]), dy = (B.y[0] + B.y[2]) - (A.y[0] + A.y[2]);
    auto dot = [](double ax, double ay, double bx, double by) { return __builtin_fma(ax, bx, ay * by); };
    const double papb = dot(pax, pay, pbx, pby), paqb = dot(pax, pay, qbx, qby);
    const double qapb = dot(qax, qay, pbx, pby), qaqb = dot(qax, qay, qbx, qby);
    // (P.Q of one box is zero up to rounding, ~1e-15 of the margin: left out)
    const double g0 = __builtin_fabs(dot(pax, pay, dx, dy)) - (dot(pax, pay, pax, pay) + __builtin_fabs(papb) + __builtin_fabs(paqb));
    const double g1 = __builtin_fabs(dot(qax, qay, dx, dy)) - (dot(qax, qay, qax, qay) + __builtin_fabs(qapb) + __builtin_fabs(qaqb));
    const double g2 = __builtin_fabs(dot(pbx, pby, dx, dy)) - (dot(pbx, pby, pbx, pby) + __builtin_fabs(papb) + __builtin_fabs(qapb));
    const double g3 = __builtin_fabs(dot(qbx, qby, dx, dy)) - (dot(qbx, qby, qbx, qby) + __builtin_fabs(paqb) + __builtin_fabs(qaqb));
    const double g = __builtin_fmax(__builtin_fmax(g0, g1), __builtin_fmax(g2, g3));
    return g > kRectMargin ? 0 : (g < -kRectMargin ? 1 : 2);
}

// Certificate of separation in front of sat_quads(A, B) for a RECTANGLE A (participant box) and a convex CCW polygon B:
// true when some edge of B has the whole of A strictly on its outer side -- A's support along the edge's outward normal
// n = (ey, -ex) does not reach the edge's line: n.(c - B_j) - (|n.P| + |n.Q|) / 2 > margin (P, Q the box's edge vectors, c its
// centre; everything below carries the factor 2).  Then orient(B_j, B_k, A_i) < 0 for all four vertices by more than the margin
// (1e-9 |n|, a nanometre: a thousand times the rounding of either evaluation), which is the oracle's separating-edge rule.
// Padded vertices (triangles repeat vertex 0) give a zero normal: never certifies.
// Returns 0: separated (certified), 1: intersecting (certified: the box's centre lies strictly inside B -- inside every edge
// by the same margin -- so no edge of either polygon can have all of the other's vertices outside), 2: undecided.
T2D_DEV int rect_vs_convex_filter(const Quad& A, const Quad& B) {
    const double px = A.x[0] - A.x[3], py = A.y[0] - A.y[3], qx = A.x[1] - A.x[0], qy = A.y[1] - A.y[0];
    const double c2x = A.x[0] + A.x[2], c2y = A.y[0] + A.y[2];
    bool out = false, in = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (j + 1) & 3;
        const double nx = B.y[k] - B.y[j], ny = B.x[j] - B.x[k];
        const double s2 = __builtin_fma(nx, c2x - 2.0 * B.x[j], ny * (c2y - 2.0 * B.y[j]));
        const double e2 = __builtin_fabs(__builtin_fma(nx, px, ny * py)) + __builtin_fabs(__builtin_fma(nx, qx, ny * qy));
        const double m = 2e-9 * (__builtin_fabs(nx) + __builtin_fabs(ny));
        out |= s2 - e2 > m;
        in &= (s2 < -m) | ((nx == 0.0) & (ny == 0.0));   // (the zero-length edge of a padded triangle says nothing)
    }
    return out ? 0 : (in ? 1 : 2);
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

// squared distance of the point (cx, cy) from the segment P -> Q (oracle t2do_seg_dist2)
T2D_DEV double seg_dist2(double px, double py, double qx, double qy, double cx, double cy) {
    double dx = qx - px, dy = qy - py;
    double wx = cx - px, wy = cy - py;
    double dd = dx * dx + dy * dy;
    double t = 0.0;
    if (dd > 0.0) {
        t = (wx * dx + wy * dy) / dd;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    }
    double ex = wx - t * dx, ey = wy - t * dy;
    return ex * ex + ey * ey;
}


// ---- off-lane = not union(lanes).contains(pose): boundary pieces of the union vs the pose --------------------------
// oracle t2do_piece_meets_quad_interior: the piece A -> B misses the open CCW quad P when some edge of P has A and B
// on its outer side or on it, or all four vertices of P lie on one closed side of the line AB
T2D_DEV bool piece_meets_quad_interior(const Quad& P, double ax, double ay, double bx, double by) {
    bool sep = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (i + 1) & 3;
        sep |= (int)(orient(P.x[i], P.y[i], P.x[k], P.y[k], ax, ay) <= 0.0) & (int)(orient(P.x[i], P.y[i], P.x[k], P.y[k], bx, by) <= 0.0);
    }
    bool all_ge = true, all_le = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double o = orient(ax, ay, bx, by, P.x[k], P.y[k]);
        all_ge &= o >= 0.0;
        all_le &= o <= 0.0;
    }
    return !(sep | all_ge | all_le);
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
