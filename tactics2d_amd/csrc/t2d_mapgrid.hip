// t2d_mapgrid.hip -- static-collision and off-lane events against maps that do NOT fit the step kernel's LDS record.
//
// The step kernel keeps the static + lane polygons of a workgroup's envs in one packed LDS record of at most 32 KiB
// (t2d_collide.hip); a reference map (map/element/map.py: hundreds of lanelets, each a ring of tens of points -> thousands of
// convex pieces, tactics2d_amd/mapgeom.py) does not fit.  The reference answers "which polygons are near this pose" with an
// STRtree over the whole map (map/element/map.py:242-329: Map.query_point / query_bbox).  Here: a uniform grid per env in HBM
// -- cell -> the convex parts whose box overlaps it -- built once by t2d_set_static_geometry / t2d_set_lane_geometry when the
// record would overflow; one lane per participant derives its pose, walks the cells its box touches and runs the SAME
// predicates as the step kernel (t2d_geom_dev.h: the oracle's arithmetic, operation by operation), reading the parts from
// global memory (L2-resident: a part is 32 bytes).  Its verdicts go to a per-participant word that the event kernel ORs into the
// flags before its reduce / status epilogue (PoolView::map_flags), so everything downstream -- env flags, check_status order,
// rewards, auto-reset, records -- is the ordinary path.
//
//   StaticCollision.update   traffic/event_detection/collision.py:37-43   any(pose.intersects(obstacle))
//   OffLane (build-defined)  traffic/event_detection/off_lane.py:16-17    not union(lanes).contains(pose): DESIGN.md 4.3a
//
// Completeness of the candidates: a part is registered in every cell its box (widened by kGridMargin) overlaps, a pose visits every
// cell its box (widened alike) overlaps; a point common to pose and part -- a touching corner included -- lies in both boxes,
// hence in a cell both know.  A part met through several cells is evaluated more than once: the verdicts are ORs.
#include "t2d_geom_dev.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

using namespace geom;
constexpr int kMapBlock = 256;

// oracle t2do_circle_convex_intersects on a (padded) quad: the centre inside, or an edge within R
T2D_DEV bool circle_vs_quad(double cx, double cy, double R, const Quad& B) {
    if (point_in_quad(B, cx, cy)) return true;
    const double R2 = R * R;
    bool hit = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (j + 3) & 3;   // edge k -> j, the order of circle_vs_generic (previous vertex first)
        hit |= seg_dist2(B.x[k], B.y[k], B.x[j], B.y[j], cx, cy) <= R2;
    }
    return hit;
}

__global__ __launch_bounds__(kMapBlock) void map_events_kernel(PoolView pv, MapGridView mg, uint32_t* out) {
    const int i = blockIdx.x * kMapBlock + threadIdx.x;
    if (i >= pv.N) return;
    const uint32_t ids = pv.ids[i];
    const float fx = pv.x[i], fy = pv.y[i], fh = pv.heading[i];
    uint32_t f = 0u;
    // (a participant whose pose is not finite takes no part in event detection: t2d_collide.hip, oracle t2do_collide)
    if (((ids >> kIdsActiveShift) & 0xffu) && __builtin_isfinite(fx) && __builtin_isfinite(fy) && __builtin_isfinite(fh)) {
        const int env = i / pv.A;
        const int type = (ids >> kIdsTypeShift) & 0xff;
        const int kind = (int)pv.params[T2D_P_SHAPE * T2D_MAX_TYPES + type];
        const double L = pv.params[T2D_P_LENGTH * T2D_MAX_TYPES + type];
        const double W = pv.params[T2D_P_WIDTH * T2D_MAX_TYPES + type];
        const double cx = (double)fx, cy = (double)fy;
        Quad A;
        double lo_x, hi_x, lo_y, hi_y;
        const double rad = 0.5 * W;
        if (kind == T2D_SHAPE_OBB) {   // Vehicle.get_pose: the step kernel's pose phase, expression by expression
            double s, c;
            sincos_det((double)fh, s, c);
            const double hl = 0.5 * L, hw = 0.5 * W;
            const double chl = c * hl, shw = s * hw, shl = s * hl, chw = c * hw;
            const double u = chl + shw, w = chl - shw;
            const double pp = shl - chw, qq = shl + chw;
            A.x[0] = u + cx; A.x[1] = w + cx; A.x[2] = cx - u; A.x[3] = cx - w;
            A.y[0] = pp + cy; A.y[1] = qq + cy; A.y[2] = cy - pp; A.y[3] = cy - qq;
            const double mx = __builtin_fmax(__builtin_fabs(u), __builtin_fabs(w));
            const double my = __builtin_fmax(__builtin_fabs(pp), __builtin_fabs(qq));
            lo_x = cx - mx; hi_x = cx + mx; lo_y = cy - my; hi_y = cy + my;
        } else {
            lo_x = cx - rad; hi_x = cx + rad; lo_y = cy - rad; hi_y = cy + rad;
#pragma unroll
            for (int k = 0; k < 4; ++k) { A.x[k] = cx; A.y[k] = cy; }
        }
        const MapGridEnv g = mg.env[env];
        // cells the widened box touches (empty when the pose lies wholly outside the grid's extent)
        const double m = (double)kGridMargin;
        int ix0 = (int)__builtin_floor((lo_x - m - (double)g.x0) * (double)g.inv_cell);
        int ix1 = (int)__builtin_floor((hi_x + m - (double)g.x0) * (double)g.inv_cell);
        int iy0 = (int)__builtin_floor((lo_y - m - (double)g.y0) * (double)g.inv_cell);
        int iy1 = (int)__builtin_floor((hi_y + m - (double)g.y0) * (double)g.inv_cell);
        ix0 = ix0 < 0 ? 0 : ix0;
        iy0 = iy0 < 0 ? 0 : iy0;
        ix1 = ix1 >= g.nx ? g.nx - 1 : ix1;
        iy1 = iy1 >= g.ny ? g.ny - 1 : iy1;
        bool hit_static = false, in_lane = false, cut = false;
        const double px = A.x[0] - A.x[3], py = A.y[0] - A.y[3], qx = A.x[1] - A.x[0], qy = A.y[1] - A.y[0];
        const double c2x = A.x[0] + A.x[2], c2y = A.y[0] + A.y[2];
        for (int iy = iy0; iy <= iy1; ++iy) {
            for (int ix = ix0; ix <= ix1; ++ix) {
                const int c = g.cell_off + iy * g.nx + ix;
                const int it1 = mg.cell_start[c + 1];
                for (int it = mg.cell_start[c]; it < it1; ++it) {
                    const uint32_t item = mg.cell_items[it];
                    const int k = (int)(item >> 31), part = (int)(item & 0x7fffffffu);
                    const int v0 = mg.vert_off[k][part], n = mg.vert_off[k][part + 1] - v0;
                    const Quad B = load_quad_f32(mg.xy[k] + 2 * (size_t)v0, n);
                    if (k == 0) {
                        if (!hit_static)
                            hit_static = kind == T2D_SHAPE_OBB ? sat_quads(A, B) : circle_vs_quad(cx, cy, rad, B);
                        continue;
                    }
                    in_lane |= point_in_quad(B, cx, cy);
                    const int b1 = mg.bnd_off[part + 1];
                    for (int b = mg.bnd_off[part]; b < b1 && !cut; ++b) {
                        const double* P = mg.bnd + 4 * (size_t)b;
                        const double ax = P[0], ay = P[1], bx = P[2], by = P[3];
                        if (kind == T2D_SHAPE_OBB) {
                            // (the support test of the step kernel's process_lane first: a piece clear of the pose's extent
                            // along its own normal cannot meet it -- same margin, same verdicts)
                            const double nx = ay - by, ny = bx - ax;
                            const double s2 = __builtin_fma(nx, c2x - 2.0 * ax, ny * (c2y - 2.0 * ay));
                            const double e2 = __builtin_fabs(__builtin_fma(nx, px, ny * py)) + __builtin_fabs(__builtin_fma(nx, qx, ny * qy));
                            const bool clear = __builtin_fabs(s2) > e2 + 2e-9 * (__builtin_fabs(nx) + __builtin_fabs(ny));
                            if (!clear) cut = piece_meets_quad_interior(A, ax, ay, bx, by);
                        } else {
                            cut = seg_dist2(ax, ay, bx, by, cx, cy) < rad * rad;
                        }
                    }
                }
            }
        }
        if (hit_static) f |= T2D_FLAG_COLLISION_STATIC;
        if (g.has_lanes && !(in_lane && !cut)) f |= T2D_FLAG_OFF_LANE;
    }
    out[i] = f;
}

}  // namespace

hipError_t launch_map_events(const PoolView& v, const MapGridView& mg, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(map_events_kernel, dim3((v.N + kMapBlock - 1) / kMapBlock), dim3(kMapBlock), 0, s, v, mg, out);
    return hipGetLastError();
}

}  // namespace t2d
