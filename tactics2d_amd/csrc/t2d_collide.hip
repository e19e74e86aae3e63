// t2d_collide.hip -- event detection for gfx950 (MI355X): participant-vs-participant and
// participant-vs-static-polygon closed-set `intersects`, map-boundary containment, the
// build-defined off-lane test, and the ScenarioManager status / reward epilogue.
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   Vehicle.get_pose          participant/element/vehicle.py:263-281  (bbox order :132-142)
//   Pedestrian.get_pose       participant/element/pedestrian.py:138-149 (centre, radius)
//   StaticCollision.update    traffic/event_detection/collision.py:37-43
//   DynamicCollision.update   traffic/event_detection/collision.py:18-25 (intended semantics)
//   OutBound.update           traffic/event_detection/out_bound.py:37-48
//   OffLane.update            traffic/event_detection/off_lane.py:16-17 (stub; build-defined)
//   _ParkingScenarioManager.check_status + ParkingEnv.step/_get_reward
//                             envs/parking.py:361-392, 243-250, 148-166
//
// Mapping: 256-thread workgroup = EPB = 256 / A_pad whole environments (A_pad = max_agents
// rounded up to a power of two), one lane per participant.  Phases (LDS only, 3 barriers):
//   1. pose: deterministic fp64 sin/cos of the stored heading -> 4 OBB vertices (or circle),
//      written to LDS as coordinate planes s_v[k][lane] (SoA: conflict-free gathers); each
//      lane links itself into its env's uniform spatial-hash grid (cell >= largest
//      circum-diameter, per-env bucket heads in LDS, atomicExch-built linked lists); static and
//      lane polygons of the workgroup's envs are staged to LDS when they fit.
//   2. each lane walks the 3x3 neighbouring cells, circle-rejects candidates with a 1e-6 m
//      safety margin (never changes a result: intersecting shapes always pass) and runs the
//      separating-axis test in fp64 -- strict separation, so touching counts, like shapely.
//   3. wave ballot -> LDS OR -> per-env flags; one lane per env runs the status epilogue.
//
// Every predicate is the exact arithmetic of oracle/t2d_oracle.c (same operation order,
// -ffp-contract=off, deterministic trig), so flags are bit-exact against the oracle.
// Bound: LDS/latency + fp64 VALU (about 20 B of HBM per participant); see DESIGN.md.
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

constexpr int kBlock = 256;
constexpr int kStageEnvs = 8;      // stage static geometry when EPB <= 8 (A_pad >= 32)
constexpr int kStageVerts = 128;   // per env, per kind (static / lane)
constexpr int kStagePolys = 16;    // per env, per kind
constexpr int kMaxHeads = 512;     // EPB * H for A_pad >= 8
constexpr double kRejectMargin = 1e-6;

T2D_DEV double orient(double px, double py, double qx, double qy, double rx, double ry) {
    double a = qx - px, b = ry - py;
    double c = qy - py, d = rx - px;
    return a * b - c * d;
}

// Accessor over a polygon stored as interleaved x,y doubles (static / lane polygons; flat
// pointer: LDS when staged, global otherwise).
struct PolyAoS {
    const double* p;
    int n;
    T2D_DEV void get(int j, double& x, double& y) const {
        x = p[2 * j];
        y = p[2 * j + 1];
    }
};
// Accessor over another participant's OBB in the LDS coordinate planes.
struct ObbLds {
    const double* base;  // &s_v[0][lane_j]
    static constexpr int n = 4;
    T2D_DEV void get(int j, double& x, double& y) const {
        x = base[(2 * j) * kBlock];
        y = base[(2 * j + 1) * kBlock];
    }
};

// closed-set convex `intersects`: A = own OBB (registers), B via accessor.  Same orientation
// evaluations as oracle t2do_convex_intersects(A, 4, B, n).
template <class PB>
T2D_DEV bool sat_obb(const double (&ax)[4], const double (&ay)[4], const PB& B) {
    const int nB = B.n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double px = ax[i], py = ay[i], qx = ax[(i + 1) & 3], qy = ay[(i + 1) & 3];
        bool all_out = true;
        for (int j = 0; j < nB; ++j) {
            double rx, ry;
            B.get(j, rx, ry);
            if (!(orient(px, py, qx, qy, rx, ry) < 0.0)) { all_out = false; break; }
        }
        if (all_out) return false;
    }
    for (int j = 0; j < nB; ++j) {
        double px, py, qx, qy;
        B.get(j, px, py);
        B.get(j + 1 == nB ? 0 : j + 1, qx, qy);
        bool all_out = true;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!(orient(px, py, qx, qy, ax[i], ay[i]) < 0.0)) all_out = false;
        if (all_out) return false;
    }
    return true;
}

template <class PB>
T2D_DEV bool point_in_convex(const PB& B, double x, double y) {
    const int n = B.n;
    for (int j = 0; j < n; ++j) {
        double px, py, qx, qy;
        B.get(j, px, py);
        B.get(j + 1 == n ? 0 : j + 1, qx, qy);
        if (orient(px, py, qx, qy, x, y) < 0.0) return false;
    }
    return true;
}

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

// oracle t2do_circle_convex_intersects
template <class PB>
T2D_DEV bool circle_vs_convex(double cx, double cy, double R, const PB& B) {
    if (point_in_convex(B, cx, cy)) return true;
    const double R2 = R * R;
    const int n = B.n;
    for (int j = 0; j < n; ++j) {
        double px, py, qx, qy;
        B.get(j, px, py);
        B.get(j + 1 == n ? 0 : j + 1, qx, qy);
        if (seg_dist2(px, py, qx, qy, cx, cy) <= R2) return true;
    }
    return false;
}

T2D_DEV uint32_t cell_hash(int cx, int cy) {
    return ((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u);
}

struct StageDesc {  // per staged env, per kind
    int p0, np;     // first polygon, polygon count
    int v0;         // first vertex (global index)
};

template <bool WITH_STATUS>
__global__ __launch_bounds__(kBlock) void collide_kernel(PoolView pv, t2d_status_config cfg,
                                                         int interval_ms, int log2A, int stage) {
    __shared__ double s_v[8][kBlock];      // OBB vertex coordinate planes x0,y0,...,x3,y3
    __shared__ double s_c[3][kBlock];      // centre x, centre y, bounding radius
    __shared__ int s_kind[kBlock];         // T2D_SHAPE_* or -1 = inactive
    __shared__ int s_head[kMaxHeads];
    __shared__ int s_next[kBlock];
    __shared__ uint32_t s_flags[kBlock];
    __shared__ uint32_t s_env_or[kBlock];
    __shared__ double s_poly[2][kStageEnvs][2 * kStageVerts];
    __shared__ double s_aabb[2][kStageEnvs][4 * kStagePolys];
    __shared__ int s_voff[2][kStageEnvs][kStagePolys + 1];
    __shared__ StageDesc s_desc[2][kStageEnvs];

    const int tid = threadIdx.x;
    const int A_pad = 1 << log2A;
    const int EPB = kBlock >> log2A;
    const int env_local = tid >> log2A;
    const int agent = tid & (A_pad - 1);
    const int env = blockIdx.x * EPB + env_local;
    const bool valid = env < pv.n_env && agent < pv.A;
    const int idx = valid ? env * pv.A + agent : 0;
    const bool use_grid = log2A >= 3;
    const int H = 2 * A_pad;  // buckets per env (power of two)

    // ---------------- phase 1: pose, grid insert, geometry staging -----------------------
    uint32_t ids = 0;
    float fx = 0, fy = 0, fh = 0;
    if (valid) {
        ids = pv.ids[idx];
        fx = pv.x[idx];
        fy = pv.y[idx];
        fh = pv.heading[idx];
    }
    const bool active = valid && ((ids >> kIdsActiveShift) & 0xffu);
    const int type = (ids >> kIdsTypeShift) & 0xff;

    if (use_grid)
        for (int k = tid; k < EPB * H; k += kBlock) s_head[k] = -1;
    s_env_or[tid] = 0;

    if (stage) {  // cooperative copy of this workgroup's static + lane polygons into LDS
        if (tid < 2 * EPB) {
            const int kind = tid / EPB, el = tid % EPB;
            const int e = blockIdx.x * EPB + el;
            const int32_t* eoff = kind == 0 ? pv.env_poly_off : pv.env_lane_off;
            const int32_t* voff = kind == 0 ? pv.poly_vert_off : pv.lane_vert_off;
            StageDesc d{0, 0, 0};
            if (eoff && e < pv.n_env) {
                d.p0 = eoff[e];
                d.np = eoff[e + 1] - d.p0;
                d.v0 = voff[d.p0];
            }
            s_desc[kind][el] = d;
        }
    }
    __syncthreads();  // (a) heads cleared, descriptors visible

    double ax[4], ay[4];
    double cx = (double)fx, cy = (double)fy, R = 0.0, rad = 0.0;
    int kind = -1;
    int gcx = 0, gcy = 0;
    if (active) {
        const double L = pv.params[T2D_P_LENGTH * T2D_MAX_TYPES + type];
        const double W = pv.params[T2D_P_WIDTH * T2D_MAX_TYPES + type];
        kind = (int)pv.params[T2D_P_SHAPE * T2D_MAX_TYPES + type];
        R = pv.params[T2D_P_RESERVED0 * T2D_MAX_TYPES + type];  // bounding radius (host-computed)
        rad = 0.5 * W;
        if (kind == T2D_SHAPE_OBB) {
            double s, c;
            sincos_det((double)fh, s, c);
            const double hl = 0.5 * L, hw = 0.5 * W;
            const double lx[4] = {hl, hl, -hl, -hl};
            const double ly[4] = {-hw, hw, hw, -hw};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ax[k] = c * lx[k] - s * ly[k] + cx;
                ay[k] = s * lx[k] + c * ly[k] + cy;
                s_v[2 * k][tid] = ax[k];
                s_v[2 * k + 1][tid] = ay[k];
            }
        }
        s_c[0][tid] = cx;
        s_c[1][tid] = cy;
        s_c[2][tid] = R;
        if (use_grid) {
            gcx = (int)__builtin_floor(cx * pv.inv_cell);
            gcy = (int)__builtin_floor(cy * pv.inv_cell);
            const int b = env_local * H + (int)(cell_hash(gcx, gcy) & (uint32_t)(H - 1));
            s_next[tid] = atomicExch(&s_head[b], tid);
        }
    }
    s_kind[tid] = kind;

    if (stage) {
        for (int kd = 0; kd < 2; ++kd) {
            const int32_t* voff = kd == 0 ? pv.poly_vert_off : pv.lane_vert_off;
            const double* xy = kd == 0 ? pv.poly_xy : pv.lane_xy;
            const double* bb = kd == 0 ? pv.poly_aabb : pv.lane_aabb;
            for (int el = 0; el < EPB; ++el) {
                const StageDesc d = s_desc[kd][el];
                if (d.np == 0) continue;
                const int nv = voff[d.p0 + d.np] - d.v0;
                for (int k = tid; k < 2 * nv; k += kBlock) s_poly[kd][el][k] = xy[2 * d.v0 + k];
                for (int k = tid; k < 4 * d.np; k += kBlock) s_aabb[kd][el][k] = bb[4 * d.p0 + k];
                for (int k = tid; k <= d.np; k += kBlock) s_voff[kd][el][k] = voff[d.p0 + k] - d.v0;
            }
        }
    }
    __syncthreads();  // (b) poses, grid lists and staged geometry visible

    // ---------------- phase 2: tests --------------------------------------------------------
    uint32_t f = 0;
    if (active) {
        // ---- participant vs participant --------------------------------------------------
        auto test_pair = [&](int j) -> bool {  // j = workgroup-local lane of the other participant
            const int kj = s_kind[j];
            if (kj < 0) return false;
            const double ox = s_c[0][j], oy = s_c[1][j], oR = s_c[2][j];
            const double dx = cx - ox, dy = cy - oy;
            const double rr = R + oR + kRejectMargin;
            if (dx * dx + dy * dy > rr * rr) return false;  // cannot touch
            if (kind == T2D_SHAPE_OBB && kj == T2D_SHAPE_OBB) return sat_obb(ax, ay, ObbLds{&s_v[0][j]});
            if (kind == T2D_SHAPE_OBB) return circle_vs_convex(ox, oy, oR, ObbLds{&s_v[0][tid]});
            if (kj == T2D_SHAPE_OBB) return circle_vs_convex(cx, cy, rad, ObbLds{&s_v[0][j]});
            const double r2 = rad + oR;  // circle-circle: bounding radius == radius
            return dx * dx + dy * dy <= r2 * r2;
        };
        bool hit = false;
        if (use_grid) {
            for (int oy_ = -1; oy_ <= 1 && !hit; ++oy_)
                for (int ox_ = -1; ox_ <= 1 && !hit; ++ox_) {
                    const int b = env_local * H +
                                  (int)(cell_hash(gcx + ox_, gcy + oy_) & (uint32_t)(H - 1));
                    for (int j = s_head[b]; j >= 0 && !hit; j = s_next[j])
                        if (j != tid) hit = test_pair(j);
                }
        } else {
            const int j0 = env_local << log2A;
            for (int a = 0; a < A_pad && !hit; ++a)
                if (a != agent) hit = test_pair(j0 + a);
        }
        if (hit) f |= T2D_FLAG_COLLISION_DYNAMIC;

        // ---- participant vs static polygons ----------------------------------------------
        if (pv.env_poly_off) {
            int p0, np, vbase;
            const double* xy;
            const double* bb;
            const int* voff;
            if (stage) {
                p0 = 0; np = s_desc[0][env_local].np; vbase = 0;
                xy = s_poly[0][env_local]; bb = s_aabb[0][env_local]; voff = s_voff[0][env_local];
            } else {
                p0 = pv.env_poly_off[env];
                np = pv.env_poly_off[env + 1] - p0;
                vbase = 0;
                xy = pv.poly_xy; bb = pv.poly_aabb; voff = pv.poly_vert_off;
            }
            (void)vbase;
            for (int p = p0; p < p0 + np; ++p) {
                const double m = R + kRejectMargin;
                if (cx + m < bb[4 * p] || cx - m > bb[4 * p + 1] || cy + m < bb[4 * p + 2] ||
                    cy - m > bb[4 * p + 3])
                    continue;
                const int v0 = voff[p];
                const PolyAoS B{xy + 2 * v0, voff[p + 1] - v0};
                const bool h2 = kind == T2D_SHAPE_OBB ? sat_obb(ax, ay, B)
                                                      : circle_vs_convex(cx, cy, rad, B);
                if (h2) { f |= T2D_FLAG_COLLISION_STATIC; break; }
            }
        }

        // ---- map boundary: not boundary.contains(pose) ---------------------------------
        if (pv.boundary && (!pv.boundary_valid || pv.boundary_valid[env])) {
            const double xmin = pv.boundary[4 * env], xmax = pv.boundary[4 * env + 1];
            const double ymin = pv.boundary[4 * env + 2], ymax = pv.boundary[4 * env + 3];
            bool out = false;
            if (kind == T2D_SHAPE_OBB) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ax[k] < xmin || ax[k] > xmax || ay[k] < ymin || ay[k] > ymax) out = true;
            } else {
                if (cx - rad < xmin || cx + rad > xmax || cy - rad < ymin || cy + rad > ymax) out = true;
            }
            if (out) f |= T2D_FLAG_OUT_BOUND;
        }

        // ---- lanes (build-defined): some pose vertex lies in no lane polygon -------------
        if (pv.env_lane_off) {
            int p0, np;
            const double* xy;
            const double* bb;
            const int* voff;
            if (stage) {
                p0 = 0; np = s_desc[1][env_local].np;
                xy = s_poly[1][env_local]; bb = s_aabb[1][env_local]; voff = s_voff[1][env_local];
            } else {
                p0 = pv.env_lane_off[env];
                np = pv.env_lane_off[env + 1] - p0;
                xy = pv.lane_xy; bb = pv.lane_aabb; voff = pv.lane_vert_off;
            }
            if (np > 0) {
                const int nv = kind == T2D_SHAPE_OBB ? 4 : 1;
                bool off = false;
                for (int k = 0; k < nv && !off; ++k) {
                    double qx, qy;
                    if (kind == T2D_SHAPE_OBB) {
                        qx = s_v[2 * k][tid];
                        qy = s_v[2 * k + 1][tid];
                    } else {
                        qx = cx; qy = cy;
                    }
                    bool inside = false;
                    for (int p = p0; p < p0 + np && !inside; ++p) {
                        if (qx + kRejectMargin < bb[4 * p] || qx - kRejectMargin > bb[4 * p + 1] ||
                            qy + kRejectMargin < bb[4 * p + 2] || qy - kRejectMargin > bb[4 * p + 3])
                            continue;  // > 1e-6 m outside the polygon's box: certainly outside
                        const int v0 = voff[p];
                        inside = point_in_convex(PolyAoS{xy + 2 * v0, voff[p + 1] - v0}, qx, qy);
                    }
                    if (!inside) off = true;
                }
                if (off) f |= T2D_FLAG_OFF_LANE;
            }
        }
    }

    // ---------------- phase 3: reduce + status epilogue ------------------------------------
    if (valid) pv.flags[idx] = f;
    s_flags[tid] = f;
    if (__ballot(f != 0) != 0ull && f != 0) atomicOr(&s_env_or[env_local], f);
    __syncthreads();  // (c)

    if (valid && agent == 0) {
        pv.env_flags[env] = s_env_or[env_local];
        if (WITH_STATUS) {
            const int cnt = pv.cnt_step[env] + 1;  // parking.py:353
            pv.cnt_step[env] = cnt;
            pv.frame_ms[env] += interval_ms;
            const uint32_t ef = s_flags[(env_local << log2A) + cfg.ego_index];
            int scen = T2D_SCENARIO_NORMAL, traf = T2D_TRAFFIC_NORMAL;
            float r;
            if (cfg.max_step > 0 && cnt > cfg.max_step) {
                scen = T2D_SCENARIO_TIME_EXCEEDED; r = cfg.reward_time_exceed;
            } else if (ef & T2D_FLAG_OUT_BOUND) {
                scen = T2D_SCENARIO_OUT_BOUND; r = cfg.reward_out_bound;
            } else if (ef & T2D_FLAG_COLLISION_STATIC) {
                scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_STATIC; r = cfg.reward_collision;
            } else if (cfg.check_dynamic && (ef & T2D_FLAG_COLLISION_DYNAMIC)) {
                scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_DYNAMIC; r = cfg.reward_collision;
            } else if (cfg.check_off_lane && (ef & T2D_FLAG_OFF_LANE)) {
                scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_OFF_LANE; r = cfg.reward_collision;
            } else {
                const double tp = cfg.max_step > 0
                                      ? -tanh((double)cnt / (double)cfg.max_step) * (double)cfg.time_penalty_scale
                                      : 0.0;
                r = (float)tp;
            }
            const bool terminated = scen == T2D_SCENARIO_COMPLETED;
            const bool truncated = !terminated && (scen != T2D_SCENARIO_NORMAL || traf != T2D_TRAFFIC_NORMAL);
            uchar4 st;
            st.x = (unsigned char)scen; st.y = (unsigned char)traf;
            st.z = terminated; st.w = truncated;
            reinterpret_cast<uchar4*>(pv.status)[env] = st;
            pv.reward[env] = r;
        }
    }
}

}  // namespace

hipError_t launch_collide(const PoolView& v, const t2d_status_config& cfg, bool with_status,
                          int interval_ms, const int* geo_max, hipStream_t s) {
    // geo_max: {max polys/env, max poly verts/env, max lanes/env, max lane verts/env}
    int log2A = 0;
    while ((1 << log2A) < v.A) ++log2A;
    const int EPB = kBlock >> log2A;
    const int grid = (v.n_env + EPB - 1) / EPB;
    const int stage = (EPB <= kStageEnvs && geo_max[0] <= kStagePolys && geo_max[1] <= kStageVerts &&
                       geo_max[2] <= kStagePolys && geo_max[3] <= kStageVerts &&
                       (v.env_poly_off || v.env_lane_off))
                          ? 1
                          : 0;
    if (with_status)
        hipLaunchKernelGGL(collide_kernel<true>, dim3(grid), dim3(kBlock), 0, s, v, cfg, interval_ms, log2A, stage);
    else
        hipLaunchKernelGGL(collide_kernel<false>, dim3(grid), dim3(kBlock), 0, s, v, cfg, interval_ms, log2A, stage);
    return hipGetLastError();
}

}  // namespace t2d
