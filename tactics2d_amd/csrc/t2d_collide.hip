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
// Mapping: one workgroup = EPB whole environments (EPB * A_pad <= 256 threads, A_pad =
// max_agents rounded up to a power of two), one lane per participant.  Phases:
//   0. every global load of the workgroup is issued up front so only ONE memory latency is
//      exposed: participant state, map boundary, the 4 shape columns of the type table, and the
//      workgroup's packed static+lane geometry record (fixed stride, built once on the host at
//      t2d_set_*_geometry: per-env polygon ranges, per-polygon vertex ranges, fp32 AABBs and
//      CCW vertices) copied with 16-B loads straight into dynamic LDS.
//   1. pose: deterministic fp64 sin/cos of the stored heading -> 4 OBB vertices (or circle),
//      written to LDS as coordinate planes s_v[k][lane] (SoA: conflict-free gathers); each
//      lane links itself into its env's uniform spatial-hash grid (cell >= largest
//      circum-diameter, per-env bucket heads in LDS, atomicExch-built linked lists).
//   2. each lane walks the 3x3 neighbouring cells, circle-rejects candidates with a 1e-6 m
//      safety margin (never changes a result: intersecting shapes always pass) and runs the
//      separating-axis test in fp64 on register-resident vertices -- strict separation, so
//      touching counts, like shapely.  Static / lane polygons come from the LDS record.
//   3. wave ballot -> LDS OR -> per-env flags; one lane per env runs the status epilogue.
//
// Every predicate is the exact arithmetic of oracle/t2d_oracle.c (same operation order,
// -ffp-contract=off, deterministic trig), so flags are bit-exact against the oracle.
// Bound: LDS/latency + fp64 VALU (about 20 B of HBM per participant); see DESIGN.md.
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

#ifndef T2D_ABLATE
#define T2D_ABLATE 0  // profiling builds: 1 no lanes, 2 no narrow phase, 4 no broad phase, 8 no static
#endif
#ifndef T2D_COLLIDE_WAVES
#define T2D_COLLIDE_WAVES 4  // min waves / SIMD the register allocator must allow
#endif
constexpr int kBlock = 256;
constexpr int kMaxHeads = 512;  // EPB * H for A_pad >= 8
constexpr double kRejectMargin = 1e-6;

T2D_DEV double orient(double px, double py, double qx, double qy, double rx, double ry) {
    double a = qx - px, b = ry - py;
    double c = qy - py, d = rx - px;
    return a * b - c * d;
}

// A convex polygon held in registers, padded to MAXN vertices by repeating vertex 0.  Padding
// never changes a predicate: duplicate vertices repeat an existing test, and the padded edges
// are zero-length (orientation 0: never separating, never "outside").
template <int MAXN>
struct RegPoly {
    double x[MAXN], y[MAXN];
};

template <int MAXN>
T2D_DEV RegPoly<MAXN> load_poly_f32(const float* p, int n) {  // interleaved x,y fp32 (LDS)
    RegPoly<MAXN> r;
    const float2* q = reinterpret_cast<const float2*>(p);
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
        const float2 v = q[j < n ? j : 0];
        r.x[j] = (double)v.x;
        r.y[j] = (double)v.y;
    }
    return r;
}

T2D_DEV RegPoly<4> load_obb_lds(const double* base) {  // &s_v[0][lane_j], planes kBlock apart
    RegPoly<4> r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.x[j] = base[(2 * j) * kBlock];
        r.y[j] = base[(2 * j + 1) * kBlock];
    }
    return r;
}

// closed-set convex `intersects`, A = own OBB.  Same orientation evaluations (plus harmless
// padded ones) as oracle t2do_convex_intersects(A, 4, B, n).
template <int MAXN>
T2D_DEV bool sat_obb(const double (&ax)[4], const double (&ay)[4], const RegPoly<MAXN>& B) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double px = ax[i], py = ay[i], qx = ax[(i + 1) & 3], qy = ay[(i + 1) & 3];
        bool all_out = true;
#pragma unroll
        for (int j = 0; j < MAXN; ++j) all_out &= orient(px, py, qx, qy, B.x[j], B.y[j]) < 0.0;
        if (all_out) return false;
    }
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
        const double px = B.x[j], py = B.y[j];
        const double qx = B.x[j + 1 < MAXN ? j + 1 : 0], qy = B.y[j + 1 < MAXN ? j + 1 : 0];
        bool all_out = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) all_out &= orient(px, py, qx, qy, ax[i], ay[i]) < 0.0;
        if (all_out) return false;
    }
    return true;
}

template <int MAXN>
T2D_DEV bool point_in_convex(const RegPoly<MAXN>& B, double x, double y) {
    bool in = true;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
        const int k = j + 1 < MAXN ? j + 1 : 0;
        in &= !(orient(B.x[j], B.y[j], B.x[k], B.y[k], x, y) < 0.0);
    }
    return in;
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
template <int MAXN>
T2D_DEV bool circle_vs_convex(double cx, double cy, double R, const RegPoly<MAXN>& B, int n) {
    if (point_in_convex(B, cx, cy)) return true;
    const double R2 = R * R;
    bool hit = false;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {  // real edges only (j < n): padded ones are skipped
        const int k = j + 1 < MAXN ? j + 1 : 0;
        hit |= j < n && seg_dist2(B.x[j], B.y[j], B.x[k], B.y[k], cx, cy) <= R2;
    }
    return hit;
}

// ---- streaming variants for the rare 5..8-vertex polygons: vertices are re-read from LDS instead
// of being held in 32 VGPRs, which keeps the kernel at 4 waves / SIMD.  Same predicates.
struct PolyLds {
    const float2* q;
    int n;
    T2D_DEV void get(int j, double& x, double& y) const {
        const float2 v = q[j];
        x = (double)v.x;
        y = (double)v.y;
    }
};

T2D_DEV bool sat_obb_stream(const double (&ax)[4], const double (&ay)[4], const PolyLds& B) {
    // pass 1: every B vertex against the 4 edges of A (one LDS read per vertex, branch-free)
    bool out0 = true, out1 = true, out2 = true, out3 = true;
    for (int j = 0; j < B.n; ++j) {
        double rx, ry;
        B.get(j, rx, ry);
        out0 &= orient(ax[0], ay[0], ax[1], ay[1], rx, ry) < 0.0;
        out1 &= orient(ax[1], ay[1], ax[2], ay[2], rx, ry) < 0.0;
        out2 &= orient(ax[2], ay[2], ax[3], ay[3], rx, ry) < 0.0;
        out3 &= orient(ax[3], ay[3], ax[0], ay[0], rx, ry) < 0.0;
    }
    if (out0 | out1 | out2 | out3) return false;
    // pass 2: every edge of B against the 4 vertices of A (vertex carried between iterations)
    double px, py;
    B.get(B.n - 1, px, py);
    bool sep = false;
    for (int j = 0; j < B.n; ++j) {
        double qx, qy;
        B.get(j, qx, qy);
        bool all_out = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) all_out &= orient(px, py, qx, qy, ax[i], ay[i]) < 0.0;
        sep |= all_out;
        px = qx;
        py = qy;
    }
    return !sep;
}

T2D_DEV bool point_in_convex_stream(const PolyLds& B, double x, double y) {
    bool in = true;
    double px, py;
    B.get(B.n - 1, px, py);
    for (int j = 0; j < B.n; ++j) {
        double qx, qy;
        B.get(j, qx, qy);
        in &= !(orient(px, py, qx, qy, x, y) < 0.0);
        px = qx;
        py = qy;
    }
    return in;
}

T2D_DEV bool circle_vs_convex_stream(double cx, double cy, double R, const PolyLds& B) {
    if (point_in_convex_stream(B, cx, cy)) return true;
    const double R2 = R * R;
    bool hit = false;
    for (int j = 0; j < B.n; ++j) {
        double px, py, qx, qy;
        B.get(j, px, py);
        B.get(j + 1 == B.n ? 0 : j + 1, qx, qy);
        hit |= seg_dist2(px, py, qx, qy, cx, cy) <= R2;
    }
    return hit;
}

T2D_DEV uint32_t cell_hash(int cx, int cy) {
    return ((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u);
}

template <bool WITH_STATUS>
__global__ __launch_bounds__(kBlock, T2D_COLLIDE_WAVES) void collide_kernel(PoolView pv, t2d_status_config cfg,
                                                         int interval_ms, int log2A) {
    __shared__ double s_v[8][kBlock];   // OBB vertex coordinate planes x0,y0,...,x3,y3
    __shared__ double s_c[3][kBlock];   // centre x, centre y, bounding radius
    __shared__ double s_par[4][T2D_MAX_TYPES];  // length, width, shape, bounding radius per type
    __shared__ int s_kind[kBlock];      // T2D_SHAPE_* or -1 = inactive
    __shared__ int s_head[kMaxHeads];
    __shared__ int s_next[kBlock];
    __shared__ uint32_t s_flags[kBlock];
    __shared__ uint32_t s_env_or[kBlock];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_geo[];  // packed geometry record

    const GeoLayout& gl = pv.geo_layout;
    const int tid = threadIdx.x;
    const int A_pad = 1 << log2A;
    const int EPB = gl.epb;
    const int nthreads = EPB << log2A;
    const int env_local = tid >> log2A;
    const int agent = tid & (A_pad - 1);
    const int env = blockIdx.x * EPB + env_local;
    const bool valid = env < pv.n_env && agent < pv.A;
    const int idx = valid ? env * pv.A + agent : 0;
    const bool use_hash_grid = log2A > 6;  // envs larger than a wave use the LDS spatial hash
    const bool use_grid = use_hash_grid;
    const int H = 2 * A_pad;  // buckets per env (power of two)

    // ---------------- phase 0: issue every global load, clear LDS tables ------------------
    uint32_t ids = 0;
    float fx = 0, fy = 0, fh = 0;
    float bxmin = 0, bxmax = 0, bymin = 0, bymax = 0;
    bool has_boundary = false;
    if (valid) {
        ids = pv.ids[idx];
        fx = pv.x[idx];
        fy = pv.y[idx];
        fh = pv.heading[idx];
        if (pv.boundary) {
            const float4 b = reinterpret_cast<const float4*>(pv.boundary)[env];
            bxmin = b.x; bxmax = b.y; bymin = b.z; bymax = b.w;
            has_boundary = pv.boundary_valid ? pv.boundary_valid[env] != 0 : true;
        }
    }
    double par_stage[2] = {0.0, 0.0};  // 4 shape columns x 32 types = 128 doubles, <= 2 per thread
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = tid + k * nthreads;
        if (q < 4 * T2D_MAX_TYPES && (k == 0 || nthreads < 4 * T2D_MAX_TYPES)) {
            const int col = q / T2D_MAX_TYPES, ty = q % T2D_MAX_TYPES;
            const int src = col == 0 ? T2D_P_LENGTH : col == 1 ? T2D_P_WIDTH : col == 2 ? T2D_P_SHAPE : T2D_P_RESERVED0;
            par_stage[k] = pv.params[src * T2D_MAX_TYPES + ty];
        }
    }
    // geometry record -> LDS in batches of kBatch 16-B loads per thread (one latency per batch;
    // one batch covers 8 x nthreads x 16 B, i.e. the whole record whenever A >= 16)
    const int n_vec = pv.geo ? gl.stride >> 2 : 0;
    constexpr int kBatch = 8;
    const uint4* gsrc = reinterpret_cast<const uint4*>(pv.geo + (size_t)blockIdx.x * gl.stride);
    uint4 geo_stage[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        geo_stage[k] = make_uint4(0, 0, 0, 0);
        if (tid + k * nthreads < n_vec) geo_stage[k] = gsrc[tid + k * nthreads];
    }
    if (use_grid)
        for (int k = tid; k < EPB * H; k += nthreads) s_head[k] = -1;
    s_env_or[tid] = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = tid + k * nthreads;
        if (q < 4 * T2D_MAX_TYPES && (k == 0 || nthreads < 4 * T2D_MAX_TYPES))
            s_par[q / T2D_MAX_TYPES][q % T2D_MAX_TYPES] = par_stage[k];
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k)
        if (tid + k * nthreads < n_vec) reinterpret_cast<uint4*>(s_geo)[tid + k * nthreads] = geo_stage[k];
    for (int base = kBatch * nthreads; base < n_vec; base += kBatch * nthreads) {  // big records only
        uint4 g2[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            g2[k] = make_uint4(0, 0, 0, 0);
            if (base + tid + k * nthreads < n_vec) g2[k] = gsrc[base + tid + k * nthreads];
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
            if (base + tid + k * nthreads < n_vec)
                reinterpret_cast<uint4*>(s_geo)[base + tid + k * nthreads] = g2[k];
    }
    __syncthreads();  // (a) tables cleared, type columns + geometry record staged

    // ---------------- phase 1: pose + grid insert -----------------------------------------
    const bool active = valid && ((ids >> kIdsActiveShift) & 0xffu);
    const int type = (ids >> kIdsTypeShift) & 0xff;
    double ax[4], ay[4];
    double cx = (double)fx, cy = (double)fy, R = 0.0, rad = 0.0;
    int kind = -1;
    int gcx = 0, gcy = 0;
    if (active) {
        const double L = s_par[0][type];
        const double W = s_par[1][type];
        kind = (int)s_par[2][type];
        R = s_par[3][type];  // bounding radius (host-computed)
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
    __syncthreads();  // (b) poses and grid lists visible

    // ---------------- phase 2: tests --------------------------------------------------------
    // Broad phase for envs that fit in one wave (A_pad <= 64): every lane compares its bounding
    // circle against all agents of its env with cross-lane shuffles -- no LDS traffic, no
    // dependent chain, fp32 with a 1 cm safety margin (strictly conservative for |x|,|y| < 4 km:
    // fp32 rounding moves the test by < 1e-4 m there) -- and keeps a 64-bit candidate mask.
    // Larger envs (A_pad > 64) walk the LDS spatial-hash grid instead.  Executed by ALL lanes
    // (inactive ones publish a negative radius) because shuffles read from executing lanes only.
    unsigned long long cand = 0ull;
    const float R32 = active ? (float)R + 5e-3f : -1.0f;
    if (!use_hash_grid && !(T2D_ABLATE & 4)) {
        const int seg0 = (tid & 63) & ~(A_pad - 1);  // first lane of my env inside the wave
        for (int a = 0; a < A_pad; ++a) {
            const float ox = __shfl(fx, seg0 + a), oy = __shfl(fy, seg0 + a), oR = __shfl(R32, seg0 + a);
            const float dx = fx - ox, dy = fy - oy, rr = R32 + oR;
            const bool near = oR >= 0.0f && dx * dx + dy * dy <= rr * rr;
            cand |= (unsigned long long)near << a;
        }
        cand &= ~(1ull << agent);
    }

    uint32_t f = 0;
    if (active) {
        // ---- participant vs participant (narrow phase, fp64, oracle arithmetic) ------------
        auto test_shapes = [&](int j) -> bool {  // j = workgroup-local lane of the other participant
            const int kj = s_kind[j];
            if (kind == T2D_SHAPE_OBB && kj == T2D_SHAPE_OBB) return sat_obb(ax, ay, load_obb_lds(&s_v[0][j]));
            const double ox = s_c[0][j], oy = s_c[1][j], oR = s_c[2][j];
            if (kind == T2D_SHAPE_OBB) return circle_vs_convex(ox, oy, oR, load_obb_lds(&s_v[0][tid]), 4);
            if (kj == T2D_SHAPE_OBB) return circle_vs_convex(cx, cy, rad, load_obb_lds(&s_v[0][j]), 4);
            const double dx = cx - ox, dy = cy - oy;
            const double r2 = rad + oR;  // circle-circle: bounding radius == radius
            return dx * dx + dy * dy <= r2 * r2;
        };
        bool hit = false;
        if (!use_hash_grid) {
            const int j0 = env_local << log2A;
            while (cand != 0ull && !hit) {
                const int a = __ffsll((long long)cand) - 1;
                cand &= cand - 1ull;
                hit = (T2D_ABLATE & 2) ? false : test_shapes(j0 + a);
            }
        } else {
            for (int oy_ = -1; oy_ <= 1 && !hit; ++oy_)
                for (int ox_ = -1; ox_ <= 1 && !hit; ++ox_) {
                    const int b = env_local * H +
                                  (int)(cell_hash(gcx + ox_, gcy + oy_) & (uint32_t)(H - 1));
                    for (int j = s_head[b]; j >= 0 && !hit; j = s_next[j]) {
                        if (j == tid || s_kind[j] < 0) continue;
                        const double dx = cx - s_c[0][j], dy = cy - s_c[1][j];
                        const double rr = R + s_c[2][j] + kRejectMargin;
                        if (dx * dx + dy * dy > rr * rr) continue;  // cannot touch
                        hit = test_shapes(j);
                    }
                }
        }
        if (hit) f |= T2D_FLAG_COLLISION_DYNAMIC;

        // ---- geometry record accessors -------------------------------------------------------
        const int* pstart0 = reinterpret_cast<const int*>(s_geo) + gl.off_pstart[0];
        const int* pstart1 = reinterpret_cast<const int*>(s_geo) + gl.off_pstart[1];

        // ---- participant vs static polygons ----------------------------------------------
        // pass 1 (branch-free, loads pipeline): bit mask of polygons whose box is within reach
        // of the bounding circle; pass 2: exact test on the survivors only.
        if (gl.has[0] && !(T2D_ABLATE & 8)) {
            const int* vstart = reinterpret_cast<const int*>(s_geo) + gl.off_vstart[0];
            const float4* bb = reinterpret_cast<const float4*>(s_geo + gl.off_aabb[0]);
            const float* xy = reinterpret_cast<const float*>(s_geo + gl.off_xy[0]);
            const int pend = pstart0[env_local + 1];
            bool shit = false;
            for (int c0 = pstart0[env_local]; c0 < pend && !shit; c0 += 32) {
                const int cn = pend - c0 < 32 ? pend - c0 : 32;
                uint32_t m = 0;
                for (int q = 0; q < cn; ++q) {
                    const float4 b = bb[c0 + q];  // xmin, xmax, ymin, ymax
                    const bool ov = !(fx + R32 < b.x || fx - R32 > b.y || fy + R32 < b.z || fy - R32 > b.w);
                    m |= (uint32_t)ov << q;
                }
                while (m != 0u && !shit) {
                    const int p = c0 + __ffs((int)m) - 1;
                    m &= m - 1u;
                    const int v0 = vstart[p], n = vstart[p + 1] - v0;
                    if (n <= 4) {
                        const RegPoly<4> B = load_poly_f32<4>(xy + 2 * v0, n);
                        shit = kind == T2D_SHAPE_OBB ? sat_obb(ax, ay, B) : circle_vs_convex(cx, cy, rad, B, n);
                    } else {
                        const PolyLds B{reinterpret_cast<const float2*>(xy + 2 * v0), n};
                        shit = kind == T2D_SHAPE_OBB ? sat_obb_stream(ax, ay, B) : circle_vs_convex_stream(cx, cy, rad, B);
                    }
                }
            }
            if (shit) f |= T2D_FLAG_COLLISION_STATIC;
        }

        // ---- map boundary: not boundary.contains(pose) ---------------------------------
        if (has_boundary) {
            const double xmin = bxmin, xmax = bxmax, ymin = bymin, ymax = bymax;
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
        if (gl.has[1] && !(T2D_ABLATE & 1)) {
            const int p0 = pstart1[env_local], p1 = pstart1[env_local + 1];
            if (p1 > p0) {
                const int* vstart = reinterpret_cast<const int*>(s_geo) + gl.off_vstart[1];
                const float4* bb = reinterpret_cast<const float4*>(s_geo + gl.off_aabb[1]);
                const float* xy = reinterpret_cast<const float*>(s_geo + gl.off_xy[1]);
                const int nv = kind == T2D_SHAPE_OBB ? 4 : 1;
                double qx[4], qy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    qx[k] = kind == T2D_SHAPE_OBB ? ax[k] : cx;
                    qy[k] = kind == T2D_SHAPE_OBB ? ay[k] : cy;
                }
                // box of the pose vertices: lanes farther than the margin from it contain none
                double lo_x = qx[0], hi_x = qx[0], lo_y = qy[0], hi_y = qy[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    lo_x = qx[k] < lo_x ? qx[k] : lo_x; hi_x = qx[k] > hi_x ? qx[k] : hi_x;
                    lo_y = qy[k] < lo_y ? qy[k] : lo_y; hi_y = qy[k] > hi_y ? qy[k] : hi_y;
                }
                unsigned inside = 0;  // bit k: vertex k lies in some lane polygon
                const unsigned all = (1u << nv) - 1u;
                for (int c0 = p0; c0 < p1 && inside != all; c0 += 32) {
                    const int cn = p1 - c0 < 32 ? p1 - c0 : 32;
                    uint32_t m = 0;  // pass 1: lanes whose box is within 1e-6 m of the pose's box
                    for (int q = 0; q < cn; ++q) {
                        const float4 b = bb[c0 + q];
                        const bool ov = !(hi_x + kRejectMargin < (double)b.x || lo_x - kRejectMargin > (double)b.y ||
                                          hi_y + kRejectMargin < (double)b.z || lo_y - kRejectMargin > (double)b.w);
                        m |= (uint32_t)ov << q;
                    }
                    while (m != 0u && inside != all) {  // pass 2: exact containment per vertex
                        const int p = c0 + __ffs((int)m) - 1;
                        m &= m - 1u;
                        const int v0 = vstart[p], n = vstart[p + 1] - v0;
                        if (n <= 4) {
                            const RegPoly<4> B = load_poly_f32<4>(xy + 2 * v0, n);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (k < nv && !(inside >> k & 1u) && point_in_convex(B, qx[k], qy[k])) inside |= 1u << k;
                        } else {
                            const PolyLds B{reinterpret_cast<const float2*>(xy + 2 * v0), n};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (k < nv && !(inside >> k & 1u) && point_in_convex_stream(B, qx[k], qy[k])) inside |= 1u << k;
                        }
                    }
                }
                if (inside != all) f |= T2D_FLAG_OFF_LANE;
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
                          int interval_ms, hipStream_t s) {
    int log2A = 0;
    while ((1 << log2A) < v.A) ++log2A;
    const int EPB = v.geo_layout.epb;
    const int grid = (v.n_env + EPB - 1) / EPB;
    const int threads = EPB << log2A;
    const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
    if (with_status)
        hipLaunchKernelGGL(collide_kernel<true>, dim3(grid), dim3(threads), dyn, s, v, cfg, interval_ms, log2A);
    else
        hipLaunchKernelGGL(collide_kernel<false>, dim3(grid), dim3(threads), dyn, s, v, cfg, interval_ms, log2A);
    return hipGetLastError();
}

}  // namespace t2d
