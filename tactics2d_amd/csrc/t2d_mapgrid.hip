// t2d_mapgrid.hip -- static-collision and off-lane events against maps that do NOT fit the step kernel's LDS record.
//
// The step kernel keeps the static + lane polygons of a workgroup's envs in one packed LDS record of at most 32 KiB
// (t2d_collide.hip); a reference map (map/element/map.py: hundreds of lanelets, each a ring of tens of points -> thousands of
// convex pieces, tactics2d_amd/mapgeom.py) does not fit.  The reference answers "which polygons are near this pose" with an
// STRtree over the whole map (map/element/map.py:242-329: Map.query_point / query_bbox).  Here: a uniform grid per env in HBM
// -- cell -> the convex parts whose box overlaps it -- built once by t2d_set_static_geometry / t2d_set_lane_geometry when the
// record would overflow.  Two launches per step:
//   map_walk_kernel    FOUR lanes per participant derive its pose, walk the cells its box touches -- one registration record per
//                      lane and trip (MapItem: the part's vertices inline, 48 bytes, L2-resident) -- settle "centre in this lane
//                      part" on the spot and queue the static parts and boundary pieces whose box comes within the margin of the
//                      pose's as (participant, part) / (participant, piece) pairs; queue, poses and verdict bits go to the
//                      workgroup's segment in global memory (64 participants without a single pair are finished here);
//   map_decide_kernel  one wave per segment decides a pair per lane with the SAME predicates as the step kernel (t2d_geom_dev.h:
//                      the oracle's arithmetic, operation by operation) and writes the 64 participants' flags.
// The verdicts go to a per-participant word that the event kernel ORs into the flags before its reduce / status epilogue
// (PoolView::map_flags), so everything downstream -- env flags, check_status order, rewards, auto-reset, records -- is the
// ordinary path.
// History of the form (65 536 participants on 876 lane pieces per env, scripts/mapgrid_timing.py / mapgrid_prof.sh): one lane per
// participant walking its candidates one after the other, the parts behind two index arrays -- a chain of dependent L2 round
// trips on one wave per SIMD: 100 us; sixteen lanes per participant, inline records, the decisions behind a workgroup barrier in
// the same launch (148 registers: three workgroups per CU): 31-38 us; the decisions in a launch of their own (the walk: 92
// registers): 24 + 8 us; four lanes per participant (the pose derived a quarter as often, a wave's trips as full): 19 + 7 us.
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
#ifndef T2D_MAP_LANES
#define T2D_MAP_LANES 4
#endif
constexpr int kMapLanes = T2D_MAP_LANES;              // lanes per participant
constexpr int kMapPerBlock = kMapBlock / kMapLanes;   // participants per workgroup
constexpr int kMapQueue = 32 * kMapPerBlock;          // (participant, part / piece) pairs a workgroup hands to the second launch; more: decided from scratch there
constexpr int kMapSlotShift = 26, kMapPieceBit = 25;  // a pair: slot << 26 | is_piece << 25 | index (registration or boundary piece: the host keeps both below 2^25)
static_assert(kMapPerBlock <= 64, "six bits of slot");

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

constexpr uint32_t kHitStatic = 1u, kInLane = 2u, kCut = 4u, kActive = 8u, kHasLanes = 16u;
constexpr uint32_t kMapOverflow = 0xffffffffu;

// What the walk hands to the decisions, one segment per workgroup of sixteen participants (global memory, written and read once
// per step): the poses, the verdict bits the walk settled itself and the queue of (participant, part / piece) pairs.
struct MapSegment {
    uint32_t count, pad[3];             // 0: nothing to decide (the walk wrote the flags); kMapOverflow: more pairs than the queue holds
    uint32_t verdict[kMapPerBlock];
    MapPose pose[kMapPerBlock];
    uint32_t queue[kMapQueue];          // (kMapSlotShift, kMapPieceBit)
};

T2D_DEV uint32_t flags_of(uint32_t v) {
    uint32_t f = 0u;
    if (v & kActive) {
        if (v & kHitStatic) f |= T2D_FLAG_COLLISION_STATIC;
        if ((v & kHasLanes) && !((v & kInLane) && !(v & kCut))) f |= T2D_FLAG_OFF_LANE;
    }
    return f;
}

// The walk needs none of the fp64 predicates but "centre in this lane part": in ONE kernel with the decisions it waited for
// registers it never used.  Variants of the one-kernel form measured on the way: the pose derived by ONE lane per participant and
// read from LDS behind a barrier -- 29 % fewer VALU instructions, 52 us (a chain of latencies per workgroup, not of instructions);
// 5 / 6 / 8 waves per SIMD by spilling 50-70 registers: 65 / 60 / 100 us; the in-place decisions of a full queue out of line: 49 us.
// Of this kernel, at sixteen lanes per participant: 6 / 8 waves per SIMD (21 / 32 registers spilled) 54 / 65 us against 37 at 5;
// 16 / 8 / 4 lanes per participant 37 / 31 / 30 us (walk + decisions + the event launch behind them).  At four lanes the launch
// is four waves per SIMD, and the allocation is held to that (104 registers, none spilled).
#ifndef T2D_MAP_WALK_WAVES
#define T2D_MAP_WALK_WAVES 4
#endif
__global__ __launch_bounds__(kMapBlock, T2D_MAP_WALK_WAVES) void map_walk_kernel(PoolView pv, MapGridView mg, MapSegment* seg, uint32_t* out) {
    __shared__ MapPose s_pose[kMapPerBlock];
    __shared__ uint32_t s_verdict[kMapPerBlock];
    __shared__ uint32_t s_queue[kMapQueue];
    __shared__ int s_qn;
    __shared__ int s_cell_it0[kMapPerBlock][kMapLanes], s_cell_excl[kMapPerBlock][kMapLanes], s_cell_xy[kMapPerBlock][kMapLanes];   // a chunk of cells per participant
    const int tid = threadIdx.x;
    const int l = tid & (kMapLanes - 1);
    const int slot = tid / kMapLanes;
    const int i_raw = blockIdx.x * kMapPerBlock + slot;
    const bool live = i_raw < pv.N;
    const int i = live ? i_raw : 0;
    if (tid == 0) s_qn = 0;
    if (tid < kMapPerBlock) s_verdict[tid] = 0u;
    __syncthreads();

    // ---- the pose (all sixteen lanes alike), the cells, the candidates -----------------------------------------------------------
    const uint32_t ids = pv.ids[i];
    const float fx = pv.x[i], fy = pv.y[i], fh = pv.heading[i];
    // (a participant whose pose is not finite takes no part in event detection: t2d_collide.hip, oracle t2do_collide)
    const bool active = live && ((ids >> kIdsActiveShift) & 0xffu) && __builtin_isfinite(fx) && __builtin_isfinite(fy) && __builtin_isfinite(fh);
    const MapGridEnv g = mg.env[i / pv.A];
    uint32_t mine = 0u;   // this lane's share of the participant's verdict bits
    auto push = [&](uint32_t entry) {   // (a full queue keeps counting: the second launch then decides these sixteen poses from scratch)
        const int pos = atomicAdd(&s_qn, 1);
        if (pos < kMapQueue) s_queue[pos] = entry;
    };
    if (active) {
        const int type = (ids >> kIdsTypeShift) & 0xff;
        const int kind = (int)pv.params[T2D_P_SHAPE * T2D_MAX_TYPES + type];
        const double L = pv.params[T2D_P_LENGTH * T2D_MAX_TYPES + type];
        const double W = pv.params[T2D_P_WIDTH * T2D_MAX_TYPES + type];
        const double cx = (double)fx, cy = (double)fy;
        double lo_x, hi_x, lo_y, hi_y;
        {
            MapPose P;
            P.cx = cx; P.cy = cy; P.rad = 0.5 * W; P.obb = kind == T2D_SHAPE_OBB; P.pad = 0;
            if (P.obb) {   // Vehicle.get_pose: the step kernel's pose phase, expression by expression
                double sn, cs;
                sincos_det((double)fh, sn, cs);
                const double hl = 0.5 * L, hw = 0.5 * W;
                const double chl = cs * hl, shw = sn * hw, shl = sn * hl, chw = cs * hw;
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
            if (l == 0) {
                s_pose[slot] = P;
                mine |= kActive | (g.has_lanes ? kHasLanes : 0u);
            }
        }
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
        const uint32_t tag = (uint32_t)slot << kMapSlotShift;
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
            // (the records of trip t + 1 are requested before those of trip t are looked at: a trip's latency hides behind the
            // previous trip's arithmetic instead of following it -- 44.2 -> 43.2 us per step; fetching sixteen cells' ranges per
            // chunk, four per lane, instead of four: 44.5)
            int jj = 0;
            float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;
            uint4 n2 = make_uint4(0u, 0u, 0u, 0u);
            int n_it = 0, n_cxy = 0;
            auto fetch = [&](int fidx) {
                while (jj + 1 < kMapLanes && s_cell_excl[slot][jj + 1] <= fidx) ++jj;   // (monotone: fidx only grows)
                n_it = s_cell_it0[slot][jj] + (fidx - s_cell_excl[slot][jj]);
                n_cxy = s_cell_xy[slot][jj];
                const float4* rp = reinterpret_cast<const float4*>(mg.items + n_it);
                n0 = rp[0];
                n1 = rp[1];
                n2 = reinterpret_cast<const uint4*>(rp)[2];
            };
            int fidx = l;
            bool have = fidx < total;
            if (have) fetch(fidx);
            while (have) {
                const float4 r0 = n0, r1 = n1;
                const uint4 r2 = n2;
                const int it = n_it, cxy = n_cxy;
                fidx += kMapLanes;
                have = fidx < total;
                if (have) fetch(fidx);
                const int ix = cxy & 0xffff, iy = cxy >> 16;
                // once per part: in the first cell its range and the pose's share
                const int pix0 = (int)(r2.z & 0xffffu), piy0 = (int)(r2.z >> 16);
                if (ix != (pix0 > ix0 ? pix0 : ix0) || iy != (piy0 > iy0 ? piy0 : iy0)) continue;
                const float pxmin = fminf(fminf(r0.x, r0.z), fminf(r1.x, r1.z)), pxmax = fmaxf(fmaxf(r0.x, r0.z), fmaxf(r1.x, r1.z));
                const float pymin = fminf(fminf(r0.y, r0.w), fminf(r1.y, r1.w)), pymax = fmaxf(fmaxf(r0.y, r0.w), fmaxf(r1.y, r1.w));
                if ((double)pxmax < bx0 || (double)pxmin > bx1 || (double)pymax < by0 || (double)pymin > by1) continue;
                if (r2.w == 0u) {
                    push(tag | (uint32_t)it);
                } else {
                    if (point_in_quad(quad_of(r0, r1), cx, cy)) mine |= kInLane;
                    for (int b = (int)r2.x; b < (int)r2.y; ++b) push(tag | (1u << kMapPieceBit) | (uint32_t)b);
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();   // (the next chunk overwrites the tables)
        }
    }
    if (mine) atomicOr(&s_verdict[slot], mine);
    __syncthreads();

    // ---- hand-over: sixteen poses without a single pair to decide are finished here; otherwise the poses, the verdicts so far
    // and the queue go to the workgroup's segment -----------------------------------------------------------------------------------
    const int qn = s_qn;
    MapSegment& sg = seg[blockIdx.x];
    if (qn == 0) {
        if (live && l == 0) out[i_raw] = flags_of(s_verdict[slot]);
        if (tid == 0) sg.count = 0u;
        return;
    }
    if (tid == 0) sg.count = qn > kMapQueue ? kMapOverflow : (uint32_t)qn;
    if (tid < kMapPerBlock) sg.verdict[tid] = s_verdict[tid];
    {   // (poses of participants that take no part were never written: what travels is never read)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(s_pose);
        uint32_t* dst = reinterpret_cast<uint32_t*>(sg.pose);
        for (int k = tid; k < (int)(sizeof(MapPose) * kMapPerBlock / 4); k += kMapBlock) dst[k] = src[k];
    }
    const int n_copy = qn < kMapQueue ? qn : kMapQueue;
    for (int k = tid; k < n_copy; k += kMapBlock) sg.queue[k] = s_queue[k];
}

// The decisions: one wave per segment, a pair per lane -- StaticCollision.update / the off-lane boundary test with the step kernel's
// predicates -- then the sixteen participants' flags.
__global__ __launch_bounds__(64) void map_decide_kernel(PoolView pv, MapGridView mg, const MapSegment* seg, uint32_t* out) {
    __shared__ uint32_t s_v[kMapPerBlock];
    const MapSegment& sg = seg[blockIdx.x];
    const uint32_t count = sg.count;
    if (count == 0u) return;   // (the walk wrote these participants' flags)
    const int lane = threadIdx.x;
    if (lane < kMapPerBlock) s_v[lane] = sg.verdict[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (count != kMapOverflow) {
        for (uint32_t e = lane; e < count; e += 64u) {
            const uint32_t en = sg.queue[e];
            const int sl = (int)(en >> kMapSlotShift);
            const uint32_t idx = en & ((1u << kMapPieceBit) - 1u);
            const MapPose P = sg.pose[sl];
            if (en & (1u << kMapPieceBit)) {
                if (piece_cuts(P, mg.bnd + 4 * (size_t)idx)) atomicOr(&s_v[sl], kCut);
            } else {
                const float4* rp = reinterpret_cast<const float4*>(mg.items + idx);
                const float4 r0 = rp[0], r1 = rp[1];
                if (static_hit(P, quad_of(r0, r1))) atomicOr(&s_v[sl], kHitStatic);
            }
        }
    } else if (lane < kMapPerBlock && (s_v[lane] & kActive)) {
        // More pairs than the queue holds (hundreds of parts around sixteen poses): every registration of every cell the pose
        // touches, decided in place by one lane per participant -- a registration met twice is decided twice, the verdicts are ORs.
        // (kInLane is the walk's: it settles every "centre inside" itself.)
        const MapPose P = sg.pose[lane];
        const MapGridEnv g = mg.env[(blockIdx.x * kMapPerBlock + lane) / pv.A];
        double lo_x, hi_x, lo_y, hi_y;
        if (P.obb) {   // (the box of the rounded vertices IS cx -+ max(|u|, |w|): rounding is monotone)
            lo_x = __builtin_fmin(__builtin_fmin(P.ax[0], P.ax[1]), __builtin_fmin(P.ax[2], P.ax[3]));
            hi_x = __builtin_fmax(__builtin_fmax(P.ax[0], P.ax[1]), __builtin_fmax(P.ax[2], P.ax[3]));
            lo_y = __builtin_fmin(__builtin_fmin(P.ay[0], P.ay[1]), __builtin_fmin(P.ay[2], P.ay[3]));
            hi_y = __builtin_fmax(__builtin_fmax(P.ay[0], P.ay[1]), __builtin_fmax(P.ay[2], P.ay[3]));
        } else {
            lo_x = P.cx - P.rad; hi_x = P.cx + P.rad; lo_y = P.cy - P.rad; hi_y = P.cy + P.rad;
        }
        const double m = (double)kGridMargin;
        int ix0 = (int)__builtin_floor((lo_x - m - (double)g.x0) * (double)g.inv_cell);
        int ix1 = (int)__builtin_floor((hi_x + m - (double)g.x0) * (double)g.inv_cell);
        int iy0 = (int)__builtin_floor((lo_y - m - (double)g.y0) * (double)g.inv_cell);
        int iy1 = (int)__builtin_floor((hi_y + m - (double)g.y0) * (double)g.inv_cell);
        ix0 = ix0 < 0 ? 0 : ix0;
        iy0 = iy0 < 0 ? 0 : iy0;
        ix1 = ix1 >= g.nx ? g.nx - 1 : ix1;
        iy1 = iy1 >= g.ny ? g.ny - 1 : iy1;
        uint32_t v = 0u;
        for (int iy = iy0; iy <= iy1; ++iy)
            for (int ix = ix0; ix <= ix1; ++ix) {
                const int c = g.cell_off + iy * g.nx + ix;
                const int it1 = mg.cell_start[c + 1];
                for (int it = mg.cell_start[c]; it < it1; ++it) {
                    const float4* rp = reinterpret_cast<const float4*>(mg.items + it);
                    const uint4 r2 = reinterpret_cast<const uint4*>(rp)[2];
                    if (r2.w == 0u) {
                        if (!(v & kHitStatic) && static_hit(P, quad_of(rp[0], rp[1]))) v |= kHitStatic;
                    } else {
                        for (int b = (int)r2.x; b < (int)r2.y && !(v & kCut); ++b)
                            if (piece_cuts(P, mg.bnd + 4 * (size_t)b)) v |= kCut;
                    }
                }
            }
        if (v) atomicOr(&s_v[lane], v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int ip = blockIdx.x * kMapPerBlock + lane;
    if (lane < kMapPerBlock && ip < pv.N) out[ip] = flags_of(s_v[lane]);
}

}  // namespace

size_t map_segment_bytes(int n_participants) { return sizeof(MapSegment) * (size_t)((n_participants + kMapPerBlock - 1) / kMapPerBlock); }

hipError_t launch_map_events(const PoolView& v, const MapGridView& mg, void* segments, uint32_t* out, hipStream_t s) {
    const int n_seg = (v.N + kMapPerBlock - 1) / kMapPerBlock;
    MapSegment* seg = static_cast<MapSegment*>(segments);
    hipLaunchKernelGGL(map_walk_kernel, dim3(n_seg), dim3(kMapBlock), 0, s, v, mg, seg, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(map_decide_kernel, dim3(n_seg), dim3(64), 0, s, v, mg, (const MapSegment*)seg, out);
    return hipGetLastError();
}

}  // namespace t2d
