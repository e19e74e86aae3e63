// Installing a generated parking lot in a pool (device side), shared by the generator's kernels (t2d_generate.hip) and the
// ego step kernel (t2d_ego.hip), whose epilogue moves an env whose episode ended into the lot staged for its next one.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {
namespace scene {

// fp32 ring -> counter-clockwise fp32 ring, decided like prepare_polys (t2d_api.hip): shoelace of the fp32 values in fp64
T2D_DEV void ring_ccw_f32(const float* q, float* o) {
    double a = 0.0;   // (shoelace2 of t2d_generate.hip, term for term)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        a += (double)q[2 * i] * (double)q[2 * j + 1] - (double)q[2 * j] * (double)q[2 * i + 1];
    }
    const bool flip = a < 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = flip ? q[2 * (3 - k)] : q[2 * k];
        o[2 * k + 1] = flip ? q[2 * (3 - k) + 1] : q[2 * k + 1];
    }
}

// Polygon slot k of an env: the obstacle quad as the event kernels and the lidar want it -- counter-clockwise fp32 ring, its
// box, one record per edge -- or, for a slot the scene does not use, a box nothing can meet.  `meta`: per edge the slot of
// its ring when the ring may take part in the scan's occlusion culling (t2d_lidar.hip: counter-clockwise, convex, sin of
// every interior angle >= 0.05 -- the rule rebuild_lidar_geo applies to host-described rings), else 0xff.
T2D_DEV void install_quad_slot(float4* bb, float* xy, float4* ledge, uint8_t* meta, int k, bool used, float4 q_lo, float4 q_hi) {
    float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 box = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    uint8_t ring_meta = 0xff;
    if (used) {
        const float raw[8] = {q_lo.x, q_lo.y, q_lo.z, q_lo.w, q_hi.x, q_hi.y, q_hi.z, q_hi.w};
        ring_ccw_f32(raw, r);
        box = make_float4(r[0], r[0], r[1], r[1]);
        for (int v = 1; v < 4; ++v) {
            box.x = __builtin_fminf(box.x, r[2 * v]);
            box.y = __builtin_fmaxf(box.y, r[2 * v]);
            box.z = __builtin_fminf(box.z, r[2 * v + 1]);
            box.w = __builtin_fmaxf(box.w, r[2 * v + 1]);
        }
        bool ok = true;
        for (int i = 0; i < 4; ++i) {
            const int pi = (i + 3) & 3, ni = (i + 1) & 3;
            const double ux = (double)r[2 * pi] - (double)r[2 * i], uy = (double)r[2 * pi + 1] - (double)r[2 * i + 1];
            const double wx = (double)r[2 * ni] - (double)r[2 * i], wy = (double)r[2 * ni + 1] - (double)r[2 * i + 1];
            const double cr = wx * uy - wy * ux;   // > 0 at a convex vertex of a counter-clockwise ring
            const double den = __builtin_sqrt((ux * ux + uy * uy) * (wx * wx + wy * wy));
            ok = ok && den > 0.0 && cr >= 0.05 * den;
        }
        if (ok) ring_meta = (uint8_t)k;
    }
    bb[k] = box;
    const float4 lo = make_float4(r[0], r[1], r[2], r[3]), hi = make_float4(r[4], r[5], r[6], r[7]);
    reinterpret_cast<float4*>(xy)[2 * k] = lo;
    reinterpret_cast<float4*>(xy)[2 * k + 1] = hi;
    ledge[4 * k] = lo;                                        // (v0, v1)
    ledge[4 * k + 1] = make_float4(r[2], r[3], r[4], r[5]);   // (v1, v2)
    ledge[4 * k + 2] = hi;                                    // (v2, v3)
    ledge[4 * k + 3] = make_float4(r[6], r[7], r[0], r[1]);   // (v3, v0)
    if (meta) *reinterpret_cast<uint32_t*>(meta + 4 * k) = 0x01010101u * ring_meta;
}

// target area (t2d_set_target_areas): fp32 ring as doubles, counter-clockwise, area centroid
T2D_DEV void install_target(const SceneView& sv, int e, float4 t_lo, float4 t_hi, double& cx, double& cy) {
    float tr[8];
    const float traw[8] = {t_lo.x, t_lo.y, t_lo.z, t_lo.w, t_hi.x, t_hi.y, t_hi.z, t_hi.w};
    ring_ccw_f32(traw, tr);
    double tq[8], area = 0.0;
    cx = 0.0; cy = 0.0;
    for (int c = 0; c < 8; ++c) tq[c] = (double)tr[c];
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        area += tq[2 * i] * tq[2 * j + 1] - tq[2 * j] * tq[2 * i + 1];
    }
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        const double w = tq[2 * i] * tq[2 * j + 1] - tq[2 * j] * tq[2 * i + 1];
        cx += (tq[2 * i] + tq[2 * j]) * w;
        cy += (tq[2 * i + 1] + tq[2 * j + 1]) * w;
    }
    cx = cx / (3.0 * area);
    cy = cy / (3.0 * area);
    for (int c = 0; c < 8; ++c) sv.target_xy[8 * (size_t)e + c] = tq[c];
    sv.target_c[2 * (size_t)e] = cx;
    sv.target_c[2 * (size_t)e + 1] = cy;
}

// regenerate = 1, every step: an env whose episode just ended moves into the scene the refill launch staged for its next
// episode -- SIXTEEN LANES PER ENV: lane k < 12 copies and installs polygon slot k (ring, box, four lidar edges, culling
// byte), lane 12 the boundary, the target area and its centroid, the scene's head, lane 13 the ego's state and episode
// snapshot, lane 14 the counters and detector state.  The work of install_scene, element for element (same device functions,
// same values), as two memory round trips of plain copies instead of one lane's chain of four dependent round trips and
// twelve serial quads -- and without the in-place generator next to it: that path (make_scene: 402 registers, 110 KB of LDS
// per 64 lanes) held parking_scene_kernel to one block per CU.  An env that finds no staged scene for its episode -- by
// construction of the ring impossible (16 slots, topped up every 8 steps, at most one episode per two steps: t2d_api.hip
// regenerate_done_scenes) -- raises a sticky error word the host reports at its next synchronisation, and keeps its scene.
constexpr int kCommitLanes = 16;
// What a lane of env e's group fetches from the staging ring for its part of the commit: the env's next episode, the tag of
// the slot that episode lives in, and the lane's share of the staged record.  The ego step kernel calls this at its very start
// -- for every env, whether or not its episode will end: 600 B per env and step, and the three dependent round trips (episode
// -> tag + record) hide behind the integrator instead of standing, cold, between the step's verdict and the copy.  A staged
// slot stays as it is until its episode has been consumed (scene_refill_scan_kernel), so the values are good whenever they
// were read since the env's previous commit.
struct StagedPart {
    float4 a, b;          // lanes 0..11: the quad of polygon slot `lane`; lane 12: the target quad
    float4 bound;         // lane 12
    double s0, s1, s2;    // lanes 12, 13: start x, y, heading
    double th;            // lane 12: target heading
    int32_t episode, tag, qid, n_areas;
    uint32_t info;
};
T2D_DEV StagedPart fetch_staged(const SceneView& sv, int e, int lane) {
    constexpr int K = T2D_GEN_MAX_QUADS;
    static_assert(K <= 12, "lanes 0..11 take the polygon slots");
    const SceneArrays& S = sv.staged;
    StagedPart r;
    r.a = r.b = r.bound = make_float4(0.f, 0.f, 0.f, 0.f);
    r.s0 = r.s1 = r.s2 = r.th = 0.0;
    r.qid = -1; r.info = 0u;
    r.episode = sv.episode[e] + 1;
    const size_t slot = (size_t)e * sv.ring + (size_t)(r.episode % sv.ring);
    r.tag = sv.staged_ep[slot];
    r.n_areas = S.n_quads[slot];
    if (lane < K) {
        r.a = reinterpret_cast<const float4*>(S.quads + slot * K * 8)[2 * lane];
        r.b = reinterpret_cast<const float4*>(S.quads + slot * K * 8)[2 * lane + 1];
        r.qid = S.quad_id[slot * K + lane];
    }
    if (lane == 12 || lane == 13) {
        r.s0 = S.start[3 * slot]; r.s1 = S.start[3 * slot + 1]; r.s2 = S.start[3 * slot + 2];
    }
    if (lane == 12) {
        r.a = reinterpret_cast<const float4*>(S.target + slot * 8)[0];
        r.b = reinterpret_cast<const float4*>(S.target + slot * 8)[1];
        r.th = S.target_heading[slot];
        r.bound = reinterpret_cast<const float4*>(S.boundary)[slot];
        r.info = S.info[slot];
    }
    return r;
}

// lane = 0 .. 15 of env e's group, called by all sixteen (the caller has established that e's episode just ended);
// P = fetch_staged(sv, e, lane)
T2D_DEV void commit_staged(const PoolView& pv, const SceneView& sv, int e, int lane, const StagedPart& P) {
    constexpr int K = T2D_GEN_MAX_QUADS;
    const int episode = P.episode;
    if (P.tag != episode) {
        if (lane == 0) atomicOr(sv.commit_err, 1u);
        return;
    }
    const SceneArrays& Lv = sv.live;
    const float4 q_lo = P.a, q_hi = P.b, t_lo = P.a, t_hi = P.b, bound = P.bound;
    const int32_t qid = P.qid, n_areas = P.n_areas;
    const double sx = P.s0, sy = P.s1, sh = P.s2, th = P.th;
    const uint32_t info = P.info;
    // every earlier store of this wave -- the step epilogue's, to words the lanes below write again -- is in the L2 before the
    // stores below are issued (and so is every load of the staged slot, whose refill may start once `episode` has moved)
    __threadfence();
    if (lane < K) {   // polygon slot `lane`: the live copy of the record, then what install_scene writes for it
        reinterpret_cast<float4*>(Lv.quads + (size_t)e * K * 8)[2 * lane] = q_lo;
        reinterpret_cast<float4*>(Lv.quads + (size_t)e * K * 8)[2 * lane + 1] = q_hi;
        Lv.quad_id[(size_t)e * K + lane] = qid;
        const GeoLayout& gl = sv.gl;
        const int blk = e / gl.epb, el = e - blk * gl.epb;
        uint32_t* rec = sv.geo + (size_t)blk * gl.stride;
        install_quad_slot(reinterpret_cast<float4*>(rec + gl.off_aabb[0]) + K * el, reinterpret_cast<float*>(rec + gl.off_xy[0]) + 8 * K * el,
                          reinterpret_cast<float4*>(sv.lidar_xy) + (size_t)e * 4 * K, sv.lidar_meta ? sv.lidar_meta + (size_t)e * 4 * K : nullptr,
                          lane, lane < n_areas, q_lo, q_hi);
    } else if (lane == 12) {   // the scene's head, boundary, target area + centroid, the distance the shaping starts from
        Lv.n_quads[e] = n_areas;
        Lv.start[3 * (size_t)e] = sx; Lv.start[3 * (size_t)e + 1] = sy; Lv.start[3 * (size_t)e + 2] = sh;
        reinterpret_cast<float4*>(Lv.target + (size_t)e * 8)[0] = t_lo;
        reinterpret_cast<float4*>(Lv.target + (size_t)e * 8)[1] = t_hi;
        Lv.target_heading[e] = th;
        reinterpret_cast<float4*>(Lv.boundary)[e] = bound;
        Lv.info[e] = info;
        sv.lidar_cnt[e] = 4 * n_areas;
        reinterpret_cast<float4*>(sv.boundary)[e] = bound;
        double cx, cy;
        install_target(sv, e, t_lo, t_hi, cx, cy);
        const double dx = (double)(float)sx - cx, dy = (double)(float)sy - cy;
        const double dist = __builtin_sqrt(dx * dx + dy * dy);
        pv.min_dist[e] = dist;
        sv.snap_min_dist[e] = dist;
    } else if (lane == 13) {   // ego state + episode snapshot (t2d_reset with speed 0, then t2d_snapshot)
        const float stv[6] = {(float)sx, (float)sy, (float)sh, 0.f, 0.f, 0.f};
        float* cur[6] = {pv.x, pv.y, pv.heading, pv.speed, pv.vx, pv.vy};
        for (int k = 0; k < 6; ++k) {
            cur[k][e] = stv[k];
            sv.snap[k][e] = stv[k];
        }
        pv.ids[e] = sv.ids_word;
        sv.snap_ids[e] = sv.ids_word;
    } else if (lane == 14) {   // counters and detector state (the terminal status / reward / flags stay visible until the next step)
        pv.max_iou[e] = -INFINITY;
        pv.last_valid[e] = 0;
        pv.cnt_na[e] = 0;
        pv.iou[e] = NAN;
        pv.env_flags[e] = 0;
        pv.cnt_step[e] = 0;
        pv.frame_ms[e] = 0;
        sv.episode[e] = episode;
    }
}


}  // namespace scene
}  // namespace t2d
