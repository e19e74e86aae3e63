// t2d_lidar.hip -- single-line lidar observation of the ego of every environment (scope row f2).
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   SingleLineLidar._scan_obstacles            sensor/lidar.py:128-221  (rays x edges determinant solve)
//   SingleLineLidar._rotate_and_filter_obstacles sensor/lidar.py:98-126 (rings into the sensor frame)
//   as used by ParkingEnv: 360 beams, 20 m      envs/parking.py:303-304, 422-431
//
// One workgroup per environment.  Phase 1: the env's obstacle rings -- static polygons and, optionally,
// the poses of the other active box participants -- are turned into edges in the sensor frame (fp64,
// deterministic sincos of the stored headings) and staged in LDS.  Phase 2: one lane per beam sweeps the
// edge list from LDS (wave-uniform reads).  A conservative side-of-line pre-test (both end points more
// than 1e-5 m on the same side of the beam's line) skips the exact intersection arithmetic for the edges
// a beam cannot reach; it never changes a result because such an edge can only produce intersections the
// reference's own 1e-8 segment filter rejects.  The accepted (beam, edge) pairs run the reference's
// arithmetic operation by operation (IEEE division, no contraction), so the output equals the oracle bit
// for bit.  Output: fp32 [n_env][n_beams], +inf = no return: written once, coalesced -- the one genuinely
// HBM-streaming product of the step (5.9 MB at 4096 envs x 360 beams).
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

constexpr int kLidarBlock = 128;

T2D_DEV double lidar_edge(double a, double b, double lx, double ly, double R, double x1, double y1, double x2,
                          double y2) {
    const double tz = 1e-8, tinf = R * 10;
    const double d = y2 - y1, e = x1 - x2, f = y1 * x2 - x1 * y2;
    double det = a * e - b * d;
    const bool parallel = det == 0.0;
    if (parallel) det = 1.0;
    double rx = (b * f) / det;
    double ry = (-(a * f)) / det;
    const double mx = tz > lx ? tz : lx, nx = -tz < lx ? -tz : lx;
    const double my = tz > ly ? tz : ly, ny = -tz < ly ? -tz : ly;
    if (rx > mx + tz) rx = tinf;
    if (rx < nx - tz) rx = tinf;
    if (ry > my + tz) ry = tinf;
    if (ry < ny - tz) ry = tinf;
    if (rx > (x1 > x2 ? x1 : x2) + tz) rx = tinf;
    if (rx < (x1 < x2 ? x1 : x2) - tz) rx = tinf;
    if (ry > (y1 > y2 ? y1 : y2) + tz) ry = tinf;
    if (ry < (y1 < y2 ? y1 : y2) - tz) ry = tinf;
    if (parallel) rx = tinf;
    return __builtin_sqrt(rx * rx + ry * ry);
}

__global__ __launch_bounds__(kLidarBlock) void lidar_kernel(PoolView pv, LidarView lv, float* out) {
    extern __shared__ __attribute__((aligned(16))) double s_edge[];  // [slots][4] = x1, y1, x2, y2 (sensor frame)
    __shared__ double s_ego[4];                                       // cos, sin, x_off, y_off
    __shared__ int s_ego_active;
    const int env = blockIdx.x;
    const int tid = threadIdx.x;
    const int A = pv.A;
    const size_t base = (size_t)env * A;
    const double kFar = 1e30;  // an edge nobody can see: yields >= 10 R for every beam, like no edge at all

    if (tid == 0) {
        const size_t ie = base + lv.ego_index;
        const uint32_t ids = pv.ids[ie];
        s_ego_active = (ids >> kIdsActiveShift) & 0xff;
        double sn, cs;
        sincos_det((double)pv.heading[ie], sn, cs);
        const double px = pv.x[ie], py = pv.y[ie];
        s_ego[0] = cs;
        s_ego[1] = sn;
        s_ego[2] = -px * cs - py * sn;   // lidar.py:112-113
        s_ego[3] = px * sn - py * cs;
    }
    __syncthreads();
    const double cs = s_ego[0], sn = s_ego[1], x_off = s_ego[2], y_off = s_ego[3];

    // ---- phase 1a: static polygon edges (vertex v -> next vertex of its ring) ------------------------
    int n_static = 0;
    if (lv.env_vert_off) {
        const int v0 = lv.env_vert_off[env];
        n_static = lv.env_vert_off[env + 1] - v0;
        for (int q = tid; q < n_static; q += kLidarBlock) {
            const float2 p = reinterpret_cast<const float2*>(lv.xy)[v0 + q];
            const float2 r = reinterpret_cast<const float2*>(lv.xy)[lv.next_vert[v0 + q]];
            s_edge[4 * q + 0] = cs * (double)p.x + sn * (double)p.y + x_off;
            s_edge[4 * q + 1] = -sn * (double)p.x + cs * (double)p.y + y_off;
            s_edge[4 * q + 2] = cs * (double)r.x + sn * (double)r.y + x_off;
            s_edge[4 * q + 3] = -sn * (double)r.x + cs * (double)r.y + y_off;
        }
    }
    // ---- phase 1b: the other participants' boxes (4 edges each; skipped slots get the far edge) -------
    int n_slots = n_static;
    if (lv.include_participants) {
        n_slots += 4 * A;
        for (int j = tid; j < A; j += kLidarBlock) {
            const uint32_t ids = pv.ids[base + j];
            const int type = (ids >> kIdsTypeShift) & 0xff;
            const bool use = j != lv.ego_index && ((ids >> kIdsActiveShift) & 0xff) &&
                             (int)pv.params[T2D_P_SHAPE * T2D_MAX_TYPES + type] == T2D_SHAPE_OBB;
            double vx[4], vy[4];
            if (use) {
                const double L = pv.params[T2D_P_LENGTH * T2D_MAX_TYPES + type];
                const double W = pv.params[T2D_P_WIDTH * T2D_MAX_TYPES + type];
                double s2, c2;
                sincos_det((double)pv.heading[base + j], s2, c2);
                const double cx = pv.x[base + j], cy = pv.y[base + j];
                const double hl = 0.5 * L, hw = 0.5 * W;
                const double lx[4] = {hl, hl, -hl, -hl};
                const double ly[4] = {-hw, hw, hw, -hw};
#pragma unroll
                for (int k = 0; k < 4; ++k) {  // Vehicle.get_pose, then the sensor-frame transform
                    const double X = c2 * lx[k] - s2 * ly[k] + cx;
                    const double Y = s2 * lx[k] + c2 * ly[k] + cy;
                    vx[k] = cs * X + sn * Y + x_off;
                    vy[k] = -sn * X + cs * Y + y_off;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double* e = s_edge + 4 * (size_t)(n_static + 4 * j + k);
                e[0] = use ? vx[k] : kFar;
                e[1] = use ? vy[k] : kFar;
                e[2] = use ? vx[(k + 1) & 3] : kFar;
                e[3] = use ? vy[(k + 1) & 3] : kFar + 1.0;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: one lane per beam ---------------------------------------------------------------------
    const double R = lv.max_range;
    float* o = out + (size_t)env * lv.n_beams;
    for (int k = tid; k < lv.n_beams; k += kLidarBlock) {
        float res = __builtin_inff();
        if (s_ego_active && n_slots > 0) {
            const double bs = lv.beam_sin[k], bc = lv.beam_cos[k];
            const double a = bs, b = -bc;                    // lidar.py:161-162
            const double lx = bc * R, ly = bs * R;           // :201-204
            double best = __builtin_inf();
            for (int q = 0; q < n_slots; ++q) {
                const double x1 = s_edge[4 * q], y1 = s_edge[4 * q + 1], x2 = s_edge[4 * q + 2], y2 = s_edge[4 * q + 3];
                const double s1 = a * x1 + b * y1, s2 = a * x2 + b * y2;   // signed distances to the beam's line
                if ((s1 > 1e-5 && s2 > 1e-5) || (s1 < -1e-5 && s2 < -1e-5)) continue;   // cannot intersect
                const double dd = lidar_edge(a, b, lx, ly, R, x1, y1, x2, y2);
                best = dd < best ? dd : best;
            }
            best = best < 0.0 ? 0.0 : (best > R ? R : best);   // np.clip(0, R)
            res = best == R ? __builtin_inff() : (float)best;
        }
        o[k] = res;
    }
}

}  // namespace

hipError_t launch_lidar(const PoolView& v, const LidarView& lv, float* out, hipStream_t s) {
    const size_t dyn = sizeof(double) * 4 * (size_t)(lv.max_static_verts + (lv.include_participants ? 4 * v.A : 0));
    hipLaunchKernelGGL(lidar_kernel, dim3(v.n_env), dim3(kLidarBlock), dyn, s, v, lv, out);
    return hipGetLastError();
}

}  // namespace t2d
