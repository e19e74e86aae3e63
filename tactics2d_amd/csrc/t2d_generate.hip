// t2d_generate.hip -- reset-time scene synthesis on the device (scope row f4).
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   ParkingLotGenerator.generate            map/generator/generate_parking_lot.py:239-444
//   ._get_target_area / _get_back_wall      :101-125
//   ._get_left_wall / _get_right_wall       :127-173
//   ._get_side_vehicle                      :175-205
//   ._verify_obstacles / _verify_start_state :207-223, :231-237
//   Map.add_area (same id replaces)         map/element/map.py:444-453
// PARITY UNPINNED against the reference (numpy's global MT19937 stream + shapely predicates cannot run
// in this build): distributions, draw order, control flow and predicate semantics follow the reference;
// the random stream is the counter-based one specified in include/t2d.h / oracle t2do_generate_parking,
// and this kernel agrees with that oracle bit for bit.
//
// One lane per scene.  A scene is a sequential rejection sampler (every draw depends on the outcome of
// the previous tests), so the parallelism is across scenes only; the obstacle list lives in LDS
// (lane-interleaved, 1.7 KB per lane).  Reset-time work: one launch per batch of scenes, not per step.
#include <hip/hip_runtime.h>

#include <cmath>

#include "t2d_math.h"
#include "t2d_pool.h"
#include "t2d_scene_dev.h"

namespace t2d {

namespace {

constexpr int kGenBlock = 64;
constexpr int kListCap = 24;  // obstacle list entries per scene (3 rejected attempts' worth; more is flagged)
constexpr double kPi = 3.141592653589793;
constexpr double kTwoPi = 2.0 * 3.141592653589793;

struct Quad {
    double v[8];
};

struct Stream {  // splitmix64 counter stream, one per scene
    uint64_t s;
    T2D_DEV double u() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
    T2D_DEV double uniform(double a, double b) { return a + (b - a) * u(); }
    T2D_DEV double normal(double mean, double std) {  // Box-Muller, cosine branch
        const double u1 = 1.0 - u(), u2 = u();
        const double rad = __builtin_sqrt(-2.0 * log_det(u1));
        double sn, cs;
        sincos_det(kTwoPi * u2, sn, cs);
        return mean + std * (rad * cs);
    }
    T2D_DEV double trunc_gauss(double mean, double std, double lo, double hi) {  // :60-62
        return clipd(normal(mean, std), lo, hi);
    }
};

// _get_bbox: body-frame ring (+L/2,-W/2), (+L/2,+W/2), (-L/2,+W/2), (-L/2,-W/2) through [cos, -sin, sin, cos, cx, cy]
T2D_DEV Quad make_box(double cx, double cy, double h, double len, double wid) {
    double sn, cs;
    sincos_det(h, sn, cs);
    Quad q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double lx = (k < 2 ? 0.5 : -0.5) * len;
        const double ly = (k == 1 || k == 2 ? 0.5 : -0.5) * wid;
        q.v[2 * k] = (cs * lx + (-sn) * ly) + cx;
        q.v[2 * k + 1] = (sn * lx + cs * ly) + cy;
    }
    return q;
}

T2D_DEV double turn(const double* p, const double* q, const double* r) {
    const double a = q[0] - p[0], b = r[1] - p[1];
    const double c = q[1] - p[1], d = r[0] - p[0];
    return a * b - c * d;
}

T2D_DEV double shoelace2(const Quad& q) {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        a += q.v[2 * i] * q.v[2 * j + 1] - q.v[2 * j] * q.v[2 * i + 1];
    }
    return a;
}

T2D_DEV Quad counter_clockwise(const Quad& q) {
    if (!(shoelace2(q) < 0.0)) return q;
    Quad o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o.v[2 * k] = q.v[2 * (3 - k)];
        o.v[2 * k + 1] = q.v[2 * (3 - k) + 1];
    }
    return o;
}

T2D_DEV bool is_convex_ccw(const Quad& q) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (turn(&q.v[2 * k], &q.v[2 * ((k + 1) & 3)], &q.v[2 * ((k + 2) & 3)]) < 0.0) ok = false;
    return ok;
}

// an edge of the counter-clockwise quad A has every vertex of B strictly on its right
T2D_DEV bool edge_separates(const Quad& A, const Quad& B) {
    bool found = false;
    for (int i = 0; i < 4; ++i) {
        const double* p = &A.v[2 * i];
        const double* q = &A.v[2 * ((i + 1) & 3)];
        bool all_out = true;
        for (int j = 0; j < 4; ++j)
            if (!(turn(p, q, &B.v[2 * j]) < 0.0)) all_out = false;
        found = found || all_out;
    }
    return found;
}

// shapely intersects for convex quads: closed sets share a point (touching counts)
T2D_DEV bool touches_or_overlaps(const Quad& a, const Quad& b) {
    const Quad A = counter_clockwise(a), B = counter_clockwise(b);
    return !(edge_separates(A, B) || edge_separates(B, A));
}

T2D_DEV double point_segment_d2(const double* p, const double* q, const double* c) {
    const double dx = q[0] - p[0], dy = q[1] - p[1];
    const double wx = c[0] - p[0], wy = c[1] - p[1];
    const double dd = dx * dx + dy * dy;
    double t = 0.0;
    if (dd > 0.0) {
        t = (wx * dx + wy * dy) / dd;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    }
    const double ex = wx - t * dx, ey = wy - t * dy;
    return ex * ex + ey * ey;
}

// shapely distance between convex quads
T2D_DEV double gap(const Quad& a, const Quad& b) {
    if (touches_or_overlaps(a, b)) return 0.0;
    double best = INFINITY;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const double d1 = point_segment_d2(&b.v[2 * j], &b.v[2 * ((j + 1) & 3)], &a.v[2 * i]);
            const double d2 = point_segment_d2(&a.v[2 * j], &a.v[2 * ((j + 1) & 3)], &b.v[2 * i]);
            best = __builtin_fmin(best, __builtin_fmin(d1, d2));
        }
    return __builtin_sqrt(best);
}

// The obstacle list of a lane and the Map.areas slot tables live in LDS, lane-interleaved (word w of lane t at
// base[w * kGenBlock + t]: conflict-free): in private memory every list access is a scratch round trip, and a scene is
// one long dependent chain, so those latencies add up to most of the launch.
struct LaneMem {
    double* q;       // [kListCap * 8] list polygons
    int* id;         // [kListCap] their reference ids
    int* slot_id;    // [T2D_GEN_MAX_QUADS] ids in Map.areas order
    int* slot_src;   // [T2D_GEN_MAX_QUADS] list index that holds the area's final polygon
    T2D_DEV Quad quad(int i) const {
        Quad r;
#pragma unroll
        for (int c = 0; c < 8; ++c) r.v[c] = q[(i * 8 + c) * kGenBlock];
        return r;
    }
    T2D_DEV void put(int i, const Quad& v) const {
#pragma unroll
        for (int c = 0; c < 8; ++c) q[(i * 8 + c) * kGenBlock] = v.v[c];
    }
};

// What `generate` hands back besides the areas (registers)
struct SceneHead {
    int n;  // areas
    double sx, sy, sh;
    Quad target;
    double target_h;
    float bound[4];
    uint32_t info;
};

// ParkingLotGenerator.generate for the scene that owns counter stream `stream_index`
T2D_DEV void make_scene(uint64_t seed, int64_t stream_index, double type_proportion, double len, double wid,
                        const LaneMem& m, SceneHead& out) {
    Stream rng{seed + (uint64_t)(stream_index + 1) * 0xD1B54A32D192ED03ull};
    constexpr double kSize = 30.0, kMargin = 13.0, kD0 = 0.8, kD1 = 1.6;
    uint32_t flags = 0;

    int n_list = 0;
    bool list_full = false;
    auto append = [&](int id, const Quad& q) {
        if (n_list >= kListCap) {
            list_full = true;
            return;
        }
        m.put(n_list, q);
        m.id[n_list * kGenBlock] = id;
        ++n_list;
    };

    const bool bay = rng.u() < type_proportion;  // :256
    const double slot_len = bay ? 7.0 : 4.5;
    const double next = bay ? wid : len;
    const int n_more = bay ? 3 : 2;  // (n_parking_lots - 3) // 2
    const double thr = bay ? 0.85 : 0.25 * len;

    auto mode_heading = [&]() {
        return bay ? rng.trunc_gauss(kPi / 2, kPi / 54, kPi * 4 / 9, kPi * 5 / 9)
                   : rng.trunc_gauss(0.0, kPi / 54, -kPi / 18, kPi / 18);
    };
    // the lower edge of a box placed at y = 0 decides how far it sits from the back wall
    auto lift = [&](const Quad& at_zero) {
        const double m = bay ? __builtin_fmin(at_zero.v[7], at_zero.v[5]) : __builtin_fmin(at_zero.v[7], at_zero.v[1]);
        const double y_min = -m + kD0;
        return rng.trunc_gauss(y_min + 0.4, 0.2, y_min, y_min + 0.8);
    };
    auto side_vehicle = [&](double d0, double d1, bool left) {  // :175-205
        const double heading = mode_heading();
        const double x = 0.0 + (left ? -1.0 : 1.0) * ((bay ? wid : len) + rng.uniform(d0, d1));
        const double y = lift(make_box(x, 0.0, heading, len, wid));
        return make_box(x, y, heading, len, wid);
    };
    auto random_position = [&](const double* origin, double a0, double a1, double r0, double r1, double* out) {  // :89-99
        const double am = (a0 + a1) / 2.0, rm = (r0 + r1) / 2.0;
        const double as = __builtin_sqrt(((a0 - am) * (a0 - am) + (a1 - am) * (a1 - am)) / 2.0);
        const double rs = __builtin_sqrt(((r0 - rm) * (r0 - rm) + (r1 - rm) * (r1 - rm)) / 2.0);
        const double angle = rng.trunc_gauss(am, as, a0, a1);
        const double radius = rng.trunc_gauss(rm, rs, r0, r1);
        double sn, cs;
        sincos_det(angle, sn, cs);
        out[0] = origin[0] + radius * cs;
        out[1] = origin[1] + radius * sn;
    };

    Quad target, back, left_ob, right_ob;
    double target_h = 0.0;
    int attempts = 0;
    for (;;) {  // :260-331
        ++attempts;
        target_h = mode_heading();
        const double cy = lift(make_box(0.0, 0.0, target_h, len, wid));
        target = make_box(0.0, cy, target_h, len, wid);
        const double wall_w = rng.uniform(0.5, 1.5);
        back = make_box(0.0, 0.0 - wall_w / 2, 0.0, kSize, wall_w);

        double d0 = kD0 + 0.1, d1 = kD1;
        if (rng.u() < 0.2) {  // wall on the left :127-149
            double a[2], b[2];
            random_position(bay ? &target.v[2] : &target.v[4], kPi * 11 / 12, kPi * 13 / 12, d0, d1, a);
            random_position(bay ? &target.v[4] : &target.v[6], kPi * 11 / 12, kPi * 13 / 12, d0, d1, b);
            left_ob = Quad{{a[0], a[1], b[0], b[1], 0.0 - kSize / 2, 0.0, 0.0 - kSize / 2, a[1]}};
        } else {
            left_ob = side_vehicle(d0, d1, true);
            for (int i = 0; i < n_more; ++i) {
                d0 += next + kD0;
                d1 += next + kD0;
                append(2 * i + 3, side_vehicle(d0, d1, true));
            }
        }
        const double dl = gap(target, left_ob);
        d0 = __builtin_fmax(thr - dl, 0.0) + kD0;
        d1 = kD1;
        if (rng.u() < 0.2) {  // wall on the right :151-173
            double a[2], b[2];
            random_position(bay ? &target.v[6] : &target.v[0], -kPi * 1 / 12, kPi * 1 / 12, d0, d1, a);
            random_position(bay ? &target.v[0] : &target.v[2], -kPi * 1 / 12, kPi * 1 / 12, d0, d1, b);
            right_ob = Quad{{0.0 + kSize / 2, target.v[3], 0.0 + kSize / 2, 0.0, a[0], a[1], b[0], b[1]}};
        } else {
            right_ob = side_vehicle(d0, d1, false);
            for (int i = 0; i < n_more; ++i) {
                d0 += next + kD0;
                d1 += next + kD0;
                append(2 * i + 4, side_vehicle(d0, d1, false));
            }
        }
        const double dr = gap(target, right_ob);
        // _verify_obstacles :207-223 (`any(dists) < 0.8` compares a bool: rejects only dl == dr == 0)
        bool valid = !(touches_or_overlaps(target, back) || touches_or_overlaps(target, left_ob) ||
                       touches_or_overlaps(target, right_ob));
        if (valid && !(dl != 0.0 || dr != 0.0)) valid = false;
        if (valid && dl + dr < thr) valid = false;
        if (valid) break;
        if (attempts >= T2D_GEN_MAX_ATTEMPTS) {
            flags |= T2D_GEN_UNVERIFIED;
            break;
        }
    }
    append(0, back);
    append(1, left_ob);
    append(2, right_ob);

    double y_max = -INFINITY;  // :338-346
    for (int i = 0; i < n_list; ++i)
        for (int k = 0; k < 4; ++k) y_max = __builtin_fmax(y_max, m.q[(i * 8 + 2 * k + 1) * kGenBlock]);
    y_max += kD0;
    if (rng.u() < 0.2) {  // far wall :347-356
        const double w = rng.uniform(0.0, 0.2);
        append(3, make_box(0.0, y_max + slot_len, 0.0, kSize, w));
    } else {  // three perturbed vehicles behind the start range :357-387
        const Quad bb = make_box(0.0, y_max + slot_len + 4, 0.0, kSize, 8.0);
        const double y0 = y_max + slot_len + 2, y1 = y_max + slot_len + 6;
        int id = n_list + 1;
        for (int t = 0; t < 3; ++t) {
            const double x = rng.uniform(0.0 - kSize / 2, 0.0 + kSize / 2);
            const double y = rng.uniform(y0, y1);
            const double h = rng.u() * 2 * kPi;
            Quad q = make_box(x, y, h, len, wid);
            for (int k = 0; k < 8; ++k) q.v[k] = q.v[k] + 0.5 * rng.u();
            bool inside = true;  // Polygon(bbox).contains(shape): closed rectangle test on the vertices
            for (int k = 0; k < 4; ++k)
                if (!(q.v[2 * k] >= bb.v[4] && q.v[2 * k] <= bb.v[0] && q.v[2 * k + 1] >= bb.v[1] &&
                      q.v[2 * k + 1] <= bb.v[3]))
                    inside = false;
            if (inside) {
                append(id, q);
                ++id;
            }
        }
    }
    {  // random drop :389-390
        int kept = 0;
        for (int i = 0; i < n_list; ++i)
            if (rng.u() >= 0.05) {
                if (kept != i) {
                    m.put(kept, m.quad(i));
                    m.id[kept * kGenBlock] = m.id[i * kGenBlock];
                }
                ++kept;
            }
        n_list = kept;
    }
    for (int i = 0; i < n_list; ++i)
        if (!is_convex_ccw(counter_clockwise(m.quad(i)))) flags |= T2D_GEN_NONCONVEX;

    // start state :396-407
    double sx = 0.0, sy = 0.0, sh = 0.0;
    int s_attempts = 0;
    for (;;) {
        ++s_attempts;
        sx = rng.uniform(-kSize / 4, kSize / 4);
        sy = rng.uniform(y_max + kD0 + 1, y_max + slot_len - 1);
        sh = rng.trunc_gauss(0.0, kPi / 54, -kPi / 18, kPi / 18);
        const Quad body = make_box(sx, sy, sh, len, wid);
        bool ok = true;
        for (int i = 0; i < n_list && ok; ++i)
            if (touches_or_overlaps(body, m.quad(i))) ok = false;
        if (ok && touches_or_overlaps(body, target)) ok = false;
        if (ok) break;
        if (s_attempts >= T2D_GEN_MAX_START_ATTEMPTS) {
            flags |= T2D_GEN_START_UNVERIFIED;
            break;
        }
    }
    // flip :409-434
    const double tx = (((target.v[0] + target.v[2]) + target.v[4]) + target.v[6]) / 4.0;
    const double ty = (((target.v[1] + target.v[3]) + target.v[5]) + target.v[7]) / 4.0;
    if (rng.u() > 0.5) {
        const Quad body = make_box(sx, sy, sh, len, wid);
        const double cx = (((body.v[0] + body.v[2]) + body.v[4]) + body.v[6]) / 4.0;
        const double cy = (((body.v[1] + body.v[3]) + body.v[5]) + body.v[7]) / 4.0;
        sx = 2 * cx - sx;
        sy = 2 * cy - sy;
        sh += kPi;
        flags |= T2D_GEN_START_FLIPPED;
        if (!bay) {
            target_h += kPi;
            target = make_box(tx, ty, target_h, len, wid);
            flags |= T2D_GEN_TARGET_FLIPPED;
        }
    }

    // Map.add_area in list order: an id already present keeps its slot and takes the new polygon
    int n_out = 0;
    for (int i = 0; i < n_list; ++i) {
        const int id_i = m.id[i * kGenBlock];
        int at = -1;
        for (int k = 0; k < n_out; ++k)
            if (m.slot_id[k * kGenBlock] == id_i) at = k;
        if (at < 0) {
            if (n_out >= T2D_GEN_MAX_QUADS) {
                flags |= T2D_GEN_OVERFLOW;
                continue;
            }
            at = n_out++;
            m.slot_id[at * kGenBlock] = id_i;
        }
        m.slot_src[at * kGenBlock] = i;
    }
    if (list_full) flags |= T2D_GEN_OVERFLOW;

    out.n = n_out;
    out.sx = sx;
    out.sy = sy;
    out.sh = sh;
    out.target = target;
    out.target_h = target_h;
    out.bound[0] = (float)__builtin_floor(__builtin_fmin(sx, tx) - kMargin);  // :436-440
    out.bound[1] = (float)__builtin_ceil(__builtin_fmax(sx, tx) + kMargin);
    out.bound[2] = (float)__builtin_floor(__builtin_fmin(sy, ty) - kMargin);
    out.bound[3] = (float)__builtin_ceil(__builtin_fmax(sy, ty) + kMargin);
    out.info = flags | (bay ? T2D_GEN_BAY : 0u) | ((uint32_t)(attempts > 255 ? 255 : attempts) << 8) |
               ((uint32_t)(s_attempts > 255 ? 255 : s_attempts) << 16);
}

// one record of the per-scene arrays (t2d_generate_parking / t2d_get_parking_scenes / the staging ring)
T2D_DEV void store_scene(const SceneArrays& A, size_t at, const LaneMem& m, const SceneHead& sc) {
    float* oq = A.quads + at * T2D_GEN_MAX_QUADS * 8;
    for (int k = 0; k < T2D_GEN_MAX_QUADS; ++k) {
        Quad q{{0, 0, 0, 0, 0, 0, 0, 0}};
        if (k < sc.n) q = m.quad(m.slot_src[k * kGenBlock]);
        for (int c = 0; c < 8; ++c) oq[8 * k + c] = (float)q.v[c];
        A.quad_id[at * T2D_GEN_MAX_QUADS + k] = k < sc.n ? m.slot_id[k * kGenBlock] : -1;
    }
    A.n_quads[at] = sc.n;
    A.start[3 * at] = sc.sx;
    A.start[3 * at + 1] = sc.sy;
    A.start[3 * at + 2] = sc.sh;
    for (int c = 0; c < 8; ++c) A.target[8 * at + c] = (float)sc.target.v[c];
    A.target_heading[at] = sc.target_h;
    for (int c = 0; c < 4; ++c) A.boundary[4 * at + c] = sc.bound[c];
    A.info[at] = sc.info;
}

// One record of the scene arrays in registers.  All its loads are issued back to back (16-B vectors, no stores in
// between), so a lane that takes a staged scene pays one memory latency, not one per word.
struct SceneRec {
    float4 quad[2 * T2D_GEN_MAX_QUADS];
    int32_t id[T2D_GEN_MAX_QUADS];
    int32_t n;
    double start[3];
    float4 target[2];
    double target_h;
    float4 bound;
    uint32_t info;
};
T2D_DEV SceneRec load_rec(const SceneArrays& A, size_t at) {
    SceneRec r;
    const float4* q = reinterpret_cast<const float4*>(A.quads + at * T2D_GEN_MAX_QUADS * 8);
    const int4* ids = reinterpret_cast<const int4*>(A.quad_id + at * T2D_GEN_MAX_QUADS);
    const float4* t = reinterpret_cast<const float4*>(A.target + at * 8);
#pragma unroll
    for (int k = 0; k < 2 * T2D_GEN_MAX_QUADS; ++k) r.quad[k] = q[k];
#pragma unroll
    for (int k = 0; k < T2D_GEN_MAX_QUADS / 4; ++k) {
        const int4 v = ids[k];
        r.id[4 * k] = v.x; r.id[4 * k + 1] = v.y; r.id[4 * k + 2] = v.z; r.id[4 * k + 3] = v.w;
    }
    r.n = A.n_quads[at];
    r.start[0] = A.start[3 * at]; r.start[1] = A.start[3 * at + 1]; r.start[2] = A.start[3 * at + 2];
    r.target[0] = t[0]; r.target[1] = t[1];
    r.target_h = A.target_heading[at];
    r.bound = reinterpret_cast<const float4*>(A.boundary)[at];
    r.info = A.info[at];
    return r;
}
T2D_DEV void store_rec(const SceneArrays& A, size_t at, const SceneRec& r) {
    float4* q = reinterpret_cast<float4*>(A.quads + at * T2D_GEN_MAX_QUADS * 8);
#pragma unroll
    for (int k = 0; k < 2 * T2D_GEN_MAX_QUADS; ++k) q[k] = r.quad[k];
#pragma unroll
    for (int k = 0; k < T2D_GEN_MAX_QUADS; ++k) A.quad_id[at * T2D_GEN_MAX_QUADS + k] = r.id[k];
    A.n_quads[at] = r.n;
    A.start[3 * at] = r.start[0]; A.start[3 * at + 1] = r.start[1]; A.start[3 * at + 2] = r.start[2];
    reinterpret_cast<float4*>(A.target + at * 8)[0] = r.target[0];
    reinterpret_cast<float4*>(A.target + at * 8)[1] = r.target[1];
    A.target_heading[at] = r.target_h;
    reinterpret_cast<float4*>(A.boundary)[at] = r.bound;
    A.info[at] = r.info;
}

// What t2d_set_static_geometry / t2d_set_target_areas / t2d_reset / t2d_snapshot would do for env e, written in place:
// the env's K polygon slots of the workgroup geometry record (dead slots get a box nothing can meet), its lidar ring
// slots, boundary, target area + area centroid, the ego's state and episode snapshot, the IoU / shaping state.
T2D_DEV void install_scene(const PoolView& pv, const SceneView& sv, int e, const SceneRec& R, bool first) {
    const int n_areas = R.n;
    constexpr int K = T2D_GEN_MAX_QUADS;
    const GeoLayout& gl = sv.gl;
    const int blk = e / gl.epb, el = e - blk * gl.epb;
    uint32_t* rec = sv.geo + (size_t)blk * gl.stride;
    float4* bb = reinterpret_cast<float4*>(rec + gl.off_aabb[0]) + K * el;
    float* xy = reinterpret_cast<float*>(rec + gl.off_xy[0]) + 8 * K * el;
    float4* ledge = reinterpret_cast<float4*>(sv.lidar_xy) + (size_t)e * 4 * K;   // one record per edge
    for (int k = 0; k < K; ++k) scene::install_quad_slot(bb, xy, ledge, sv.lidar_meta ? sv.lidar_meta + (size_t)e * 4 * K : nullptr, k, k < n_areas, R.quad[2 * k], R.quad[2 * k + 1]);
    sv.lidar_cnt[e] = 4 * n_areas;
    reinterpret_cast<float4*>(sv.boundary)[e] = R.bound;
    double cx, cy;
    scene::install_target(sv, e, R.target[0], R.target[1], cx, cy);
    // ego state + episode snapshot (t2d_reset with speed 0, then t2d_snapshot); one participant per env
    const float fx = (float)R.start[0], fy = (float)R.start[1], fh = (float)R.start[2];
    const float st[6] = {fx, fy, fh, 0.f, 0.f, 0.f};
    float* cur[6] = {pv.x, pv.y, pv.heading, pv.speed, pv.vx, pv.vy};
    for (int k = 0; k < 6; ++k) {
        cur[k][e] = st[k];
        sv.snap[k][e] = st[k];
    }
    pv.ids[e] = sv.ids_word;
    sv.snap_ids[e] = sv.ids_word;
    const double dx = (double)fx - cx, dy = (double)fy - cy;
    const double dist = __builtin_sqrt(dx * dx + dy * dy);
    pv.min_dist[e] = dist;
    sv.snap_min_dist[e] = dist;
    pv.max_iou[e] = -INFINITY;
    pv.last_valid[e] = 0;
    pv.cnt_na[e] = 0;
    pv.iou[e] = NAN;
    pv.env_flags[e] = 0;
    pv.cnt_step[e] = 0;
    pv.frame_ms[e] = 0;
    if (first) {  // a finished episode keeps its terminal status / reward / flags visible until the next step
        pv.flags[e] = 0;
        pv.reward[e] = 0.f;
        uchar4 s4;
        s4.x = T2D_SCENARIO_NORMAL; s4.y = T2D_TRAFFIC_NORMAL; s4.z = 0; s4.w = 0;
        reinterpret_cast<uchar4*>(pv.status)[e] = s4;
    }
}

// mode 0: generate into the live arrays; 1: generate + install every env (episode 0); 2: envs whose episode just ended
// move on to their next episode -- taken from the staging ring when there is one, generated here otherwise
__global__ __launch_bounds__(kGenBlock) void parking_scene_kernel(PoolView pv, SceneView sv, int n_env, int mode) {
    const int e = blockIdx.x * kGenBlock + threadIdx.x;
    if (e >= n_env) return;
    int episode = 0;
    if (mode == 2) {
        const uchar4 st = reinterpret_cast<const uchar4*>(pv.status)[e];
        if (!(st.z | st.w)) return;
        episode = sv.episode[e] + 1;
        if (sv.ring > 0) {
            const size_t slot = (size_t)e * sv.ring + (size_t)(episode % sv.ring);
            if (sv.staged_ep[slot] == episode) {  // the usual case: prepared ahead by the refill launch
                const SceneRec R = load_rec(sv.staged, slot);
                store_rec(sv.live, e, R);
                install_scene(pv, sv, e, R, false);
                __threadfence();           // the slot is free for the refill stream only once everything was read
                sv.episode[e] = episode;
                return;
            }
        }
    }
    __shared__ double s_q[kListCap * 8 * kGenBlock];
    __shared__ int s_i[(kListCap + 2 * T2D_GEN_MAX_QUADS) * kGenBlock];
    const LaneMem m{s_q + threadIdx.x, s_i + threadIdx.x, s_i + kListCap * kGenBlock + threadIdx.x,
                    s_i + (kListCap + T2D_GEN_MAX_QUADS) * kGenBlock + threadIdx.x};
    SceneHead sc;
    make_scene(sv.seed, sv.first_env + e + (int64_t)episode * sv.env_stride, sv.type_proportion, sv.len, sv.wid, m, sc);
    store_scene(sv.live, e, m, sc);
    if (mode != 0) {
        install_scene(pv, sv, e, load_rec(sv.live, e), mode == 1);   // reads back what this lane just stored
        sv.episode[e] = episode;
    }
}

// regenerate = 1 on a pool whose step is not the ego kernel (which commits in its own epilogue): a launch of its own after
// the step, sixteen lanes per env (scene::commit_staged, t2d_scene_dev.h)
__global__ __launch_bounds__(256) void scene_commit_kernel(PoolView pv, SceneView sv, int n_env) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int e = gid / scene::kCommitLanes, lane = gid & (scene::kCommitLanes - 1);
    if (e >= n_env) return;
    const uchar4 st = reinterpret_cast<const uchar4*>(pv.status)[e];
    if (!(st.z | st.w)) return;
    scene::commit_staged(pv, sv, e, lane, scene::fetch_staged(sv, e, lane));
}

// Topping up the staging ring, on the pool's own stream while the env keeps stepping.  Slot j of env e must hold the
// episode in (k, k + ring] that is congruent to j, k = the env's current episode; a slot holding anything else was consumed
// and is generated anew.  `episode` may advance meanwhile: a stale (smaller) k only postpones a slot to the next refill, and
// a slot rewritten here is `ring` episodes away from the one the step stream reads next.
// Two launches: the SCAN (one lane per slot, a few loads) appends the slots that need a scene to a list; the GENERATOR walks
// the list, one lane per scene.  In the steady state a few per cent of the envs finish an episode between two refills: one
// lane per ENV kept all n_env / 64 workgroups busy for a whole scene (46 us, 110 KB of LDS each: a quarter of the CUs with
// room for 4 instead of 16 lidar workgroups while the steps went on beside it) for two or three useful lanes per wave; the
// list fills whole waves instead, and the workgroups without work leave at once.
__global__ __launch_bounds__(256) void scene_refill_scan_kernel(SceneView sv, int n_env) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_env * sv.ring) return;
    const int e = gid / sv.ring, j = gid - e * sv.ring;
    const int k = sv.episode[e];
    const int want = k + 1 + ((j - (k + 1)) % sv.ring + sv.ring) % sv.ring;
    if (sv.staged_ep[gid] == want) return;
    sv.refill_list[atomicAdd(sv.refill_count, 1u)] = make_uint2((uint32_t)gid, (uint32_t)want);
}

__global__ __launch_bounds__(kGenBlock) void scene_refill_kernel(SceneView sv) {
    const uint32_t n = *sv.refill_count;
    if (blockIdx.x * kGenBlock >= n) return;
    __shared__ double s_q[kListCap * 8 * kGenBlock];
    __shared__ int s_i[(kListCap + 2 * T2D_GEN_MAX_QUADS) * kGenBlock];
    const LaneMem m{s_q + threadIdx.x, s_i + threadIdx.x, s_i + kListCap * kGenBlock + threadIdx.x,
                    s_i + (kListCap + T2D_GEN_MAX_QUADS) * kGenBlock + threadIdx.x};
    for (uint32_t i = blockIdx.x * kGenBlock + threadIdx.x; i < n; i += gridDim.x * kGenBlock) {
        const uint2 item = sv.refill_list[i];
        const int e = (int)item.x / sv.ring, want = (int)item.y;
        SceneHead sc;
        make_scene(sv.seed, sv.first_env + e + (int64_t)want * sv.env_stride, sv.type_proportion, sv.len, sv.wid, m, sc);
        store_scene(sv.staged, item.x, m, sc);
        __threadfence();
        sv.staged_ep[item.x] = want;
    }
}

}  // namespace

hipError_t launch_scene_refill(const SceneView& sv, int n_env, hipStream_t s) {
    if (n_env <= 0 || sv.ring <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(sv.refill_count, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const long long slots = (long long)n_env * sv.ring;
    hipLaunchKernelGGL(scene_refill_scan_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, sv, n_env);
    // (as many workgroups as one scene per env needs: more work than that -- the first fill -- goes round in the kernel's loop)
    hipLaunchKernelGGL(scene_refill_kernel, dim3((n_env + kGenBlock - 1) / kGenBlock), dim3(kGenBlock), 0, s, sv);
    return hipGetLastError();
}

hipError_t launch_scene_commit(const PoolView& v, const SceneView& sv, int n_env, hipStream_t s) {
    if (n_env <= 0 || sv.ring <= 0) return hipSuccess;
    const long long threads = (long long)n_env * scene::kCommitLanes;
    hipLaunchKernelGGL(scene_commit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, v, sv, n_env);
    return hipGetLastError();
}

hipError_t launch_parking_scenes(const PoolView& v, const SceneView& sv, int n_env, int mode, hipStream_t s) {
    if (n_env <= 0) return hipSuccess;
    hipLaunchKernelGGL(parking_scene_kernel, dim3((n_env + kGenBlock - 1) / kGenBlock), dim3(kGenBlock), 0, s, v, sv,
                       n_env, mode);
    return hipGetLastError();
}

}  // namespace t2d

extern "C" int t2d_generate_parking(int32_t device_id, uint64_t seed, int64_t first_env, int32_t n_env,
                                    double type_proportion, double vehicle_length, double vehicle_width,
                                    float* quads, int32_t* quad_id, int32_t* n_quads, double* start, float* target,
                                    double* target_heading, float* boundary, uint32_t* info) {
    using namespace t2d;
    if (n_env < 0 || !quads || !quad_id || !n_quads || !start || !target || !target_heading || !boundary || !info)
        return T2D_ERR_INVALID;
    // ParkingLotGenerator.__init__ :45-57: an invalid vehicle size falls back to the default, the proportion is clipped
    if (vehicle_length < vehicle_width || !(vehicle_length > 0.0) || !(vehicle_width > 0.0)) {
        vehicle_length = 5.3;
        vehicle_width = 2.5;
    }
    if (!(type_proportion >= 0.0)) type_proportion = 0.0;
    if (type_proportion > 1.0) type_proportion = 1.0;
    if (n_env == 0) return T2D_OK;
    if (hipSetDevice(device_id) != hipSuccess) return T2D_ERR_HIP;
    const size_t E = (size_t)n_env;
    const size_t sizes[8] = {E * T2D_GEN_MAX_QUADS * 8 * sizeof(float), E * T2D_GEN_MAX_QUADS * sizeof(int32_t),
                             E * sizeof(int32_t), E * 3 * sizeof(double), E * 8 * sizeof(float), E * sizeof(double),
                             E * 4 * sizeof(float), E * sizeof(uint32_t)};
    void* host[8] = {quads, quad_id, n_quads, start, target, target_heading, boundary, info};
    size_t total = 0, off[8];
    for (int k = 0; k < 8; ++k) {
        off[k] = total;
        total += (sizes[k] + 255) & ~(size_t)255;
    }
    char* dev = nullptr;
    if (hipMalloc(&dev, total) != hipSuccess) return T2D_ERR_HIP;
    SceneView sv{};
    sv.seed = seed; sv.first_env = first_env; sv.env_stride = 0;
    sv.type_proportion = type_proportion; sv.len = vehicle_length; sv.wid = vehicle_width;
    sv.live = SceneArrays{(float*)(dev + off[0]), (int32_t*)(dev + off[1]), (int32_t*)(dev + off[2]), (double*)(dev + off[3]),
                          (float*)(dev + off[4]), (double*)(dev + off[5]), (float*)(dev + off[6]), (uint32_t*)(dev + off[7])};
    int rc = T2D_OK;
    if (launch_parking_scenes(PoolView{}, sv, n_env, 0, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
        rc = T2D_ERR_HIP;
    for (int k = 0; k < 8 && rc == T2D_OK; ++k)
        if (hipMemcpy(host[k], dev + off[k], sizes[k], hipMemcpyDeviceToHost) != hipSuccess) rc = T2D_ERR_HIP;
    (void)hipFree(dev);
    return rc;
}
