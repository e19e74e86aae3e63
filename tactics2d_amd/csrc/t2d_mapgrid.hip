// t2d_mapgrid.hip -- static-collision and off-lane events against maps that do NOT fit the step kernel's LDS record.
//
// The step kernel keeps the static + lane polygons of a workgroup's envs in one packed LDS record of at most 32 KiB
// (t2d_collide.hip); a reference map (map/element/map.py: hundreds of lanelets, each a ring of tens of points -> thousands of
// convex pieces, tactics2d_amd/mapgeom.py) does not fit.  The reference answers "which polygons are near this pose" with an
// STRtree over the whole map (map/element/map.py:242-329: Map.query_point / query_bbox).  Here: a uniform grid per env in HBM
// -- cell -> the convex parts whose box overlaps it -- built once by t2d_set_static_geometry / t2d_set_lane_geometry when the
// record would overflow; SIXTEEN lanes per participant derive its pose, walk the cells its box touches -- one registration record
// per lane and trip (MapItem: the part's vertices inline, 48 bytes, L2-resident) -- and sort the candidates whose box comes within
// the margin of the pose's into a workgroup queue; the queue's entries -- (participant, static part) and (participant, boundary
// piece) pairs -- are then dealt over the workgroup's lanes and decided by the SAME predicates as the step kernel
// (t2d_geom_dev.h: the oracle's arithmetic, operation by operation).  The verdicts go to a per-participant word that the event
// kernel ORs into the flags before its reduce / status epilogue (PoolView::map_flags), so everything downstream -- env flags,
// check_status order, rewards, auto-reset, records -- is the ordinary path.
// (Round 6's first form -- one lane per participant walking its candidates one after the other, the parts behind two index
// arrays -- was a chain of dependent L2 round trips on one wave per SIMD: 100 us for 65 536 participants on 876 lane pieces per
// env, scripts/mapgrid_timing.py.)
//
//   StaticCollision.update   traffic/event_detection/collision.py:37-43   any(pose.intersects(obstacle))
//   OffLane (build-defined)  traffic/event_detection/off_lane.py:16-17    not union(lanes).contains(pose): DESIGN.md 4.3a
//
// Completeness of the candidates: a part is registered in every cell its box (widened by kGridMargin) overlaps, a pose visits every
// cell its box (widened alike) overlaps; a point common to pose and part -- a touching corner included -- lies in both boxes,
// hence in a cell both know.  A part met through several cells is taken in the FIRST cell the two cell ranges share (its own
// range's first cell travels in the record); the verdicts are ORs, so neither order nor a repeat could change them.
// The box test in front of the exact predicates rejects a candidate only when the two boxes are more than 2 kGridMargin apart:
// five orders of magnitude more than the rounding of any predicate behind it.
#include "t2d_geom_dev.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

using namespace geom;
constexpr int kMapBlock = 256;
constexpr int kMapLanes = 16;                         // lanes per participant
constexpr int kMapPerBlock = kMapBlock / kMapLanes;   // participants per workgroup
constexpr int kMapQueue = 1024;                       // (participant, part / piece) pairs a workgroup queues; more are decided in place

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

struct MapPose {   // what the exact predicates need of a participant (LDS, one per participant of the workgroup)
    double ax[4], ay[4];
    double cx, cy, rad;
    int32_t obb, pad;
};

T2D_DEV Quad quad_of(const float4& r0, const float4& r1) {   // (no private array: one would live in scratch memory)
    Quad B;
    B.x[0] = (double)r0.x; B.y[0] = (double)r0.y; B.x[1] = (double)r0.z; B.y[1] = (double)r0.w;
    B.x[2] = (double)r1.x; B.y[2] = (double)r1.y; B.x[3] = (double)r1.z; B.y[3] = (double)r1.w;
    return B;
}

// static part against pose: StaticCollision.update (collision.py:37-43)
T2D_DEV bool static_hit(const MapPose& P, const Quad& B) {
    if (P.obb) {
        Quad A;
#pragma unroll
        for (int j = 0; j < 4; ++j) { A.x[j] = P.ax[j]; A.y[j] = P.ay[j]; }
        return sat_quads(A, B);
    }
    return circle_vs_quad(P.cx, P.cy, P.rad, B);
}

// one boundary piece of the lane union against the open pose (off-lane)
T2D_DEV bool piece_cuts(const MapPose& P, const double* piece) {
    const double ax = piece[0], ay = piece[1], bx = piece[2], by = piece[3];
    if (P.obb) {
        Quad A;
#pragma unroll
        for (int j = 0; j < 4; ++j) { A.x[j] = P.ax[j]; A.y[j] = P.ay[j]; }
        // (the support test of the step kernel's process_lane first: a piece clear of the pose's extent along its own normal
        // cannot meet it -- same margin, same verdicts)
        const double px = A.x[0] - A.x[3], py = A.y[0] - A.y[3], qx = A.x[1] - A.x[0], qy = A.y[1] - A.y[0];
        const double c2x = A.x[0] + A.x[2], c2y = A.y[0] + A.y[2];
        const double nx = ay - by, ny = bx - ax;
        const double s2 = __builtin_fma(nx, c2x - 2.0 * ax, ny * (c2y - 2.0 * ay));
        const double e2 = __builtin_fabs(__builtin_fma(nx, px, ny * py)) + __builtin_fabs(__builtin_fma(nx, qx, ny * qy));
        const bool clear = __builtin_fabs(s2) > e2 + 2e-9 * (__builtin_fabs(nx) + __builtin_fabs(ny));
        return !clear && piece_meets_quad_interior(A, ax, ay, bx, by);
    }
    return seg_dist2(ax, ay, bx, by, P.cx, P.cy) < P.rad * P.rad;
}

constexpr uint32_t kHitStatic = 1u, kInLane = 2u, kCut = 4u;

// Measured on the way (65 536 participants, 876 lane pieces per env; scripts/mapgrid_timing.py, rocprofv3): round 6's first form
// (one lane per participant, the parts behind two index arrays) 100 us; this form 31-38 us.  The kernel is a chain of latencies
// per workgroup (state -> type row -> sincos -> cell ranges -> records -> barrier -> exact tests -> barrier) at three or four
// workgroups per CU, not of instructions: the pose derived by ONE lane per participant and read from LDS behind a barrier issues
// 29 % fewer VALU instructions and takes 52 us; 5 / 6 / 8 waves per SIMD (96 / 80 / 64 registers, 50-70 of them spilled) 65 / 60 /
// 100 us; the full queue's in-place decisions out of line 49 us; 4 waves per SIMD (128 registers, 18 spilled) 1.05 x the 3 waves
// (148 registers, none spilled) below.  What would halve it again is the exact tests in a launch of their own, so that the walk
// fits 64 registers: not built.
#ifndef T2D_MAP_WAVES
#define T2D_MAP_WAVES 3
#endif
__global__ __launch_bounds__(kMapBlock, T2D_MAP_WAVES) void map_events_kernel(PoolView pv, MapGridView mg, uint32_t* out) {
    __shared__ MapPose s_pose[kMapPerBlock];
    __shared__ uint32_t s_verdict[kMapPerBlock];
    __shared__ uint32_t s_queue[kMapQueue];   // slot << 28 | is_piece << 27 | index (item or piece; the host keeps both below 2^27)
    __shared__ int s_qn;
    __shared__ int s_cell_it0[kMapPerBlock][kMapLanes], s_cell_excl[kMapPerBlock][kMapLanes], s_cell_xy[kMapPerBlock][kMapLanes];   // phase 1: a chunk of cells per participant
    const int tid = threadIdx.x;
    const int l = tid & (kMapLanes - 1);
    const int slot = tid / kMapLanes;
    const int i_raw = blockIdx.x * kMapPerBlock + slot;
    const bool live = i_raw < pv.N;
    const int i = live ? i_raw : 0;
    if (tid == 0) s_qn = 0;
    if (tid < kMapPerBlock) s_verdict[tid] = 0u;
    __syncthreads();

    // ---- phase 1: the pose (all sixteen lanes alike), the cells, the candidates -------------------------------------------------
    const uint32_t ids = pv.ids[i];
    const float fx = pv.x[i], fy = pv.y[i], fh = pv.heading[i];
    // (a participant whose pose is not finite takes no part in event detection: t2d_collide.hip, oracle t2do_collide)
    const bool active = live && ((ids >> kIdsActiveShift) & 0xffu) && __builtin_isfinite(fx) && __builtin_isfinite(fy) && __builtin_isfinite(fh);
    const int env = i / pv.A;
    const MapGridEnv g = mg.env[env];
    uint32_t mine = 0u;   // this lane's share of the participant's verdict bits
    auto push = [&](uint32_t entry, auto&& decide_here) {
        const int pos = atomicAdd(&s_qn, 1);
        if (pos < kMapQueue) s_queue[pos] = entry;
        else decide_here();   // a full queue (hundreds of parts around sixteen poses): decided by the lane that found it
    };
    if (active) {
        const int type = (ids >> kIdsTypeShift) & 0xff;
        const int kind = (int)pv.params[T2D_P_SHAPE * T2D_MAX_TYPES + type];
        const double L = pv.params[T2D_P_LENGTH * T2D_MAX_TYPES + type];
        const double W = pv.params[T2D_P_WIDTH * T2D_MAX_TYPES + type];
        const double cx = (double)fx, cy = (double)fy;
        MapPose P;
        P.cx = cx; P.cy = cy; P.rad = 0.5 * W; P.obb = kind == T2D_SHAPE_OBB; P.pad = 0;
        double lo_x, hi_x, lo_y, hi_y;
        if (P.obb) {   // Vehicle.get_pose: the step kernel's pose phase, expression by expression
            double s, c;
            sincos_det((double)fh, s, c);
            const double hl = 0.5 * L, hw = 0.5 * W;
            const double chl = c * hl, shw = s * hw, shl = s * hl, chw = c * hw;
            const double u = chl + shw, w = chl - shw;
            const double pp = shl - chw, qq = shl + chw;
            P.ax[0] = u + cx; P.ax[1] = w + cx; P.ax[2] = cx - u; P.ax[3] = cx - w;
            P.ay[0] = pp + cy; P.ay[1] = qq + cy; P.ay[2] = cy - pp; P.ay[3] = cy - qq;
            const double mx = __builtin_fmax(__builtin_fabs(u), __builtin_fabs(w));
            const double my = __builtin_fmax(__builtin_fabs(pp), __builtin_fabs(qq));
            lo_x = cx - mx; hi_x = cx + mx; lo_y = cy - my; hi_y = cy + my;
        } else {
            lo_x = cx - P.rad; hi_x = cx + P.rad; lo_y = cy - P.rad; hi_y = cy + P.rad;
#pragma unroll
            for (int k = 0; k < 4; ++k) { P.ax[k] = cx; P.ay[k] = cy; }
        }
        if (l == 0) s_pose[slot] = P;
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
        // the box test in front of the queue: the pose's box widened by 2 margins (fp64: exact conversions of the parts' fp32 boxes)
        const double bx0 = lo_x - 2.0 * m, bx1 = hi_x + 2.0 * m, by0 = lo_y - 2.0 * m, by1 = hi_y + 2.0 * m;
        const uint32_t tag = (uint32_t)slot << 28;
        // The cells' item ranges are fetched sixteen at a time, one per lane (ONE round trip for the pose's whole neighbourhood
        // instead of one per cell), and their items dealt over the lanes as one flat list: a prefix sum over the lanes' counts,
        // kept in LDS, says which cell a flat index belongs to.
        const int ncx = ix1 - ix0 + 1, ncy = iy1 - iy0 + 1;
        const int nc = ncx > 0 && ncy > 0 ? ncx * ncy : 0;
        const float inv_ncx = 1.0f / (float)(ncx > 0 ? ncx : 1);
        for (int c0 = 0; c0 < nc; c0 += kMapLanes) {
            const int j = c0 + l;
            int it0 = 0, cnt = 0, cell_xy = 0;
            if (j < nc) {
                // j = jy * ncx + jx without an integer division (j < 2^20: the float quotient is off by at most one)
                int jy = (int)((float)j * inv_ncx);
                int jx = j - jy * ncx;
                if (jx < 0) { jx += ncx; --jy; }
                if (jx >= ncx) { jx -= ncx; ++jy; }
                const int c = g.cell_off + (iy0 + jy) * g.nx + ix0 + jx;
                it0 = mg.cell_start[c];
                cnt = mg.cell_start[c + 1] - it0;
                cell_xy = (ix0 + jx) | ((iy0 + jy) << 16);
            }
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < kMapLanes; d <<= 1) {
                const int up = __shfl_up(incl, d, kMapLanes);
                if (l >= d) incl += up;
            }
            const int total = __shfl(incl, kMapLanes - 1, kMapLanes);
            s_cell_it0[slot][l] = it0;
            s_cell_excl[slot][l] = incl - cnt;
            s_cell_xy[slot][l] = cell_xy;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's LDS operations complete in order)
            __builtin_amdgcn_wave_barrier();
            int jj = 0;
            for (int fidx = l; fidx < total; fidx += kMapLanes) {
                while (jj + 1 < kMapLanes && s_cell_excl[slot][jj + 1] <= fidx) ++jj;   // (monotone: fidx only grows)
                const int it = s_cell_it0[slot][jj] + (fidx - s_cell_excl[slot][jj]);
                const int cxy = s_cell_xy[slot][jj];
                const int ix = cxy & 0xffff, iy = cxy >> 16;
                const float4* rp = reinterpret_cast<const float4*>(mg.items + it);
                const float4 r0 = rp[0], r1 = rp[1];
                const uint4 r2 = reinterpret_cast<const uint4*>(rp)[2];
                // once per part: in the first cell its range and the pose's share
                const int pix0 = (int)(r2.z & 0xffffu), piy0 = (int)(r2.z >> 16);
                if (ix != (pix0 > ix0 ? pix0 : ix0) || iy != (piy0 > iy0 ? piy0 : iy0)) continue;
                const float pxmin = fminf(fminf(r0.x, r0.z), fminf(r1.x, r1.z)), pxmax = fmaxf(fmaxf(r0.x, r0.z), fmaxf(r1.x, r1.z));
                const float pymin = fminf(fminf(r0.y, r0.w), fminf(r1.y, r1.w)), pymax = fmaxf(fmaxf(r0.y, r0.w), fmaxf(r1.y, r1.w));
                if ((double)pxmax < bx0 || (double)pxmin > bx1 || (double)pymax < by0 || (double)pymin > by1) continue;
                if (r2.w == 0u) {
                    push(tag | (uint32_t)it, [&] { if (static_hit(P, quad_of(r0, r1))) mine |= kHitStatic; });
                } else {
                    if (point_in_quad(quad_of(r0, r1), cx, cy)) mine |= kInLane;
                    for (int b = (int)r2.x; b < (int)r2.y; ++b)
                        push(tag | (1u << 27) | (uint32_t)b, [&] { if (piece_cuts(P, mg.bnd + 4 * (size_t)b)) mine |= kCut; });
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();   // (the next chunk overwrites the tables)
        }
    }
    if (mine) atomicOr(&s_verdict[slot], mine);
    __syncthreads();

    // ---- phase 2: the queued pairs, one per lane -------------------------------------------------------------------------------
    const int qn = s_qn < kMapQueue ? s_qn : kMapQueue;
    for (int e = tid; e < qn; e += kMapBlock) {
        const uint32_t en = s_queue[e];
        const int sl = (int)(en >> 28);
        const uint32_t idx = en & ((1u << 27) - 1u);
        const MapPose P = s_pose[sl];
        if (en & (1u << 27)) {
            if (piece_cuts(P, mg.bnd + 4 * (size_t)idx)) atomicOr(&s_verdict[sl], kCut);
        } else {
            const float4* rp = reinterpret_cast<const float4*>(mg.items + idx);
            const float4 r0 = rp[0], r1 = rp[1];
            if (static_hit(P, quad_of(r0, r1))) atomicOr(&s_verdict[sl], kHitStatic);
        }
    }
    __syncthreads();
    if (live && l == 0) {
        const uint32_t v = s_verdict[slot];
        uint32_t f = 0u;
        if (active) {
            if (v & kHitStatic) f |= T2D_FLAG_COLLISION_STATIC;
            if (g.has_lanes && !((v & kInLane) && !(v & kCut))) f |= T2D_FLAG_OFF_LANE;
        }
        out[i_raw] = f;
    }
}

}  // namespace

hipError_t launch_map_events(const PoolView& v, const MapGridView& mg, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(map_events_kernel, dim3((v.N + kMapPerBlock - 1) / kMapPerBlock), dim3(kMapBlock), 0, s, v, mg, out);
    return hipGetLastError();
}

}  // namespace t2d
