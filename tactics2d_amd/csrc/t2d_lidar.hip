// t2d_lidar.hip -- single-line lidar observation of the ego of every environment (scope row f2).
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   SingleLineLidar._scan_obstacles            sensor/lidar.py:128-221  (rays x edges determinant solve)
//   SingleLineLidar._rotate_and_filter_obstacles sensor/lidar.py:98-126 (rings into the sensor frame)
//   as used by ParkingEnv: 360 beams, 20 m      envs/parking.py:303-304, 422-431
//
// One workgroup per environment.  Phase 1: the env's obstacle rings -- static polygons and, optionally,
// the poses of the other active box participants -- are turned into edges in the sensor frame (fp64,
// deterministic sincos of the stored headings) and staged in LDS together with the range of beam indices
// each edge can possibly be seen by (the arc between its end points, widened by a beam on each side; all
// beams if the edge's line passes within 1 mm of the sensor; none if the edge lies beyond the range).
// Phase 2: one lane per beam sweeps the edge list with an integer span test (wave-uniform LDS reads),
// collects its candidates in a bit mask and runs the exact arithmetic only on those.  The span is strictly
// conservative, so it never changes a result: an edge outside it can only produce intersections the
// reference's own 1e-8 segment / ray filters reject.  The accepted (beam, edge) pairs run the reference's
// arithmetic operation by operation (IEEE division, no contraction), so the output equals the oracle bit
// for bit.  Output: fp32 [n_env][n_beams], +inf = no return: written once, coalesced -- the one genuinely
// HBM-streaming product of the step (5.9 MB at 4096 envs x 360 beams).
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {

namespace {

constexpr int kLidarBlock = 128;
#ifndef T2D_LIDAR_ONE_WAVE
#define T2D_LIDAR_ONE_WAVE 0   // (round 6, measured and left off: one wave per env scans 4096 lots in 21.7 us against 20.0 with two --
                               // HIP events around each launch, same box, ABAB; the vector step 29.7 against 27.4 -- bit-identical)
#endif
constexpr int kOneWaveMinEnvs = 2048;   // one wave per env must still put >= 2 waves on every SIMD
constexpr int kLidarQueue = 256;  // (beam, edge) candidates per wave per round (LidarView::queue_len: 128 where that buys a workgroup per CU)
// An edge's beam span in LDS, one word: first beam (12 bits, < n_beams <= 4096) | length + 1 (13 bits, -1 .. n_beams) |
// "16 + ring" of a front edge with a core (5 bits, else 0: see the occlusion culling below).  0 = nothing to scatter.
T2D_DEV uint32_t pack_span(int2 sp, int core = 0) { return (uint32_t)sp.x | (uint32_t)(sp.y + 1) << 12 | (uint32_t)core << 25; }
constexpr int kLidarStaticLds = 104;      // the kernel's __shared__ arrays (culling tables + queue counters)
constexpr int kLidarLdsPerCu16 = 10240;   // 160 KB / 16 workgroups (8 waves per SIMD of two-wave workgroups)

// What the determinant solve needs of an edge, computed once per edge instead of once per (beam, edge) candidate:
// d, e, f of the edge's line and the segment's coordinate bounds with the reference's 1e-8 slack already applied --
// the same expressions, so the same bits.
struct EdgePre {
    double d, e, f, x_hi, x_lo, y_hi, y_lo, pad;
};
T2D_DEV EdgePre edge_pre(double x1, double y1, double x2, double y2) {
    const double tz = 1e-8;
    EdgePre p;
    p.d = y2 - y1;
    p.e = x1 - x2;
    p.f = y1 * x2 - x1 * y2;
    p.x_hi = (x1 > x2 ? x1 : x2) + tz;
    p.x_lo = (x1 < x2 ? x1 : x2) - tz;
    p.y_hi = (y1 > y2 ? y1 : y2) + tz;
    p.y_lo = (y1 < y2 ? y1 : y2) - tz;
    p.pad = 0.0;
    return p;
}

// n1 / d and n2 / d, each correctly rounded exactly as the `/` operator rounds it: the compiler's own fp64 division sequence
// (v_div_scale, v_rcp + two Newton steps on the scaled denominator, quotient, residual, v_div_fmas, v_div_fixup), written
// out so that the two quotients share the refined reciprocal -- it depends on the scaled denominator alone.  The scaling
// looks at the numerator's exponent too, so the sharing is conditional on both divisions scaling the denominator alike
// (always, for coordinates in metres); a lane where they differ divides the second quotient on its own.
T2D_DEV void div_pair(double n1, double n2, double d, double& q1, double& q2) {
    bool f1, f2, unused;
    const double ds1 = __builtin_amdgcn_div_scale(n1, d, false, &unused);   // scaled denominator
    const double ds2 = __builtin_amdgcn_div_scale(n2, d, false, &unused);
    const double ns1 = __builtin_amdgcn_div_scale(n1, d, true, &f1);        // scaled numerators
    const double ns2 = __builtin_amdgcn_div_scale(n2, d, true, &f2);
    double r = __builtin_amdgcn_rcp(ds1);
    double e = __builtin_fma(-ds1, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-ds1, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double qa = ns1 * r;
    const double ra = __builtin_fma(-ds1, qa, ns1);
    q1 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(ra, r, qa, f1), d, n1);
    if (__builtin_expect(__double_as_longlong(ds1) == __double_as_longlong(ds2), 1)) {
        const double qb = ns2 * r;
        const double rb = __builtin_fma(-ds1, qb, ns2);
        q2 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(rb, r, qb, f2), d, n2);
    } else {
        q2 = n2 / d;
    }
}

// (returns the squared distance; see the note at its end)
T2D_DEV double lidar_edge(double a, double b, double bx_hi, double bx_lo, double by_hi, double by_lo, double R,
                          const EdgePre& E) {
    const double det = a * E.e - b * E.d;
    double rx, ry;
    div_pair(b * E.f, -(a * E.f), det, rx, ry);
    // The reference replaces a coordinate that fails one of its eight bounds by 10 R, which makes the candidate's distance
    // >= 10 R: clipped to R at the end, i.e. "no return".  Any value >= R does the same, so a rejected candidate
    // contributes +inf, and the two bounds on each side of a coordinate -- the beam's and the segment's -- are one
    // (rx > hi1 or rx > hi2  <=>  rx > min(hi1, hi2)).  A NaN coordinate fails no test and makes the distance NaN, which
    // the caller skips -- as in the reference's sequence of replacements.  Parallel beam and edge (det = 0; the reference
    // divides by 1 instead and rejects the candidate): the quotients are +-inf, which fail a bound, or NaN.
    const bool bad = (rx > __builtin_fmin(bx_hi, E.x_hi)) | (rx < __builtin_fmax(bx_lo, E.x_lo)) |
                     (ry > __builtin_fmin(by_hi, E.y_hi)) | (ry < __builtin_fmax(by_lo, E.y_lo));
    const double dd = rx * rx + ry * ry;   // SQUARED distance: sqrt is monotone, so the per-beam minimum takes one sqrt at the end
    return bad ? __builtin_inf() : dd;
}

// Beam-index span of an edge given in the sensor frame: {first beam, number of further beams} or
// {0, -1} = invisible.  Beams are at angles k * dbeam.  fp32 is enough: the span is widened by the error
// margin derived below (>= 0.1 beam).
// facing (may be null): +1 = the sensor is on the OUTER side of this edge of a CCW ring (the edge faces the sensor), -1 =
// on the inner side (a back edge as long as the sensor is outside the ring), 0 = too close to call (within 1 mm of the
// edge's line, an edge shorter than 10 cm, an end point within 1 cm of the sensor): such an edge is neither.
T2D_DEV int2 edge_span(double x1, double y1, double x2, double y2, double R, int n_beams, int* facing = nullptr) {
    const float fx1 = (float)x1, fy1 = (float)y1, fx2 = (float)x2, fy2 = (float)y2;
    const float ex = fx2 - fx1, ey = fy2 - fy1;
    const float len2 = ex * ex + ey * ey;
    const float d1 = fx1 * fx1 + fy1 * fy1, d2 = fx2 * fx2 + fy2 * fy2;
    const float cross = fx1 * fy2 - fx2 * fy1;   // = orient(P1, P2, sensor): > 0 with the sensor on the left of P1 -> P2
    if (facing) {
        const bool sure = len2 >= 1e-2f && cross * cross > 1e-6f * len2 && d1 >= 1e-4f && d2 >= 1e-4f && d1 < 1e30f && d2 < 1e30f;
        *facing = !sure ? 0 : (cross < 0.0f ? 1 : -1);
    }
    if (!(len2 > 0.0f) || !(d1 < 1e30f) || !(d2 < 1e30f)) return make_int2(0, -1);  // degenerate / placeholder
    const float dline2 = cross * cross / len2;                 // squared distance sensor -> edge line
    const float t = -(fx1 * ex + fy1 * ey) / len2;             // foot point parameter
    const float dseg2 = t <= 0.0f ? d1 : (t >= 1.0f ? d2 : dline2);
    const float Rm = (float)R * 1.0001f + 1e-3f;
    if (dseg2 > Rm * Rm) return make_int2(0, -1);              // entirely beyond the range
    if (dline2 < 1e-6f) return make_int2(0, n_beams);           // line through the sensor (within 1 mm): every beam
    const float twopi = 6.2831853f;
    float a1 = atan2f(fy1, fx1), a2 = atan2f(fy2, fx2);
    float da = a2 - a1;
    if (da > 3.14159265f) da -= twopi;
    if (da < -3.14159265f) da += twopi;
    float start = da >= 0.0f ? a1 : a2;
    const float sweep = fabsf(da);
    if (start < 0.0f) start += twopi;
    const float inv = (float)n_beams / twopi;
    // beams k with start - m <= k * dbeam <= start + sweep + m.  m covers the fp32 error of the end-point angles:
    // coordinates good to ~2e-6 m seen from the nearer end point's distance (>= 1 mm here), atan2f's ~1e-6 rad, the
    // reference's own 1e-8 m segment slack, plus 0.1 beam of plain margin
    const float rmin2 = d1 < d2 ? d1 : d2;
    const float mb = 0.1f + inv * (1e-5f + 4e-6f * __builtin_amdgcn_rsqf(rmin2 > 1e-6f ? rmin2 : 1e-6f));
    int k0 = (int)ceilf(start * inv - mb);
    const int k1 = (int)floorf((start + sweep) * inv + mb);
    const int len = k1 - k0;                                    // number of FURTHER beams after k0
    if (len < 0) return make_int2(0, -1);                       // the arc falls between two beams
    if (len + 1 >= n_beams) return make_int2(0, n_beams);
    k0 %= n_beams;
    if (k0 < 0) k0 += n_beams;
    return make_int2(k0, len);
}

// WAVES = waves per SIMD the register allocation is held to: short edge lists (ParkingEnv: ~32 edges) are bound by
// the chain of dependent latencies per workgroup, so twice the resident workgroups beat the 22 spilled registers
// (39 vs 47 us at 4096 envs); long lists (participants scanned: 250+ edges) are issue-bound and keep all 98 registers.
// BLOCK = threads per env: 128 (two waves).  BLOCK = 64 -- one wave per env, six beams per lane, no workgroup barrier that means
// anything, half the waves -- was the review's proposal for the ParkingEnv scan (32 static edges need 32 lanes in phase 1 and 64 in
// the scatter: the second wave only helps in the evaluation and pays its own ego transform for it).  Built in round 6
// (-DT2D_LIDAR_ONE_WAVE=1), bit-identical, SLOWER: the evaluation is 57 % of a wave's cycles (scripts/lidar_phases.py) and its
// compaction rounds are chains of LDS round trips that eight waves per SIMD hide and four do not.
template <int WAVES, bool PARTS, int BLOCK = kLidarBlock, bool PRE = (WAVES == 8)>
__global__ __launch_bounds__(BLOCK, WAVES) void lidar_kernel(PoolView pv, LidarView lv, float* out) {
    constexpr int kLidarBlock = BLOCK;          // (shadows the namespace constant: everything below is per instantiation)
    constexpr int kChunk = BLOCK / 2;           // edges per scatter chunk: two lanes per edge
    // short edge lists (the 8-waves-per-SIMD instantiation, ParkingEnv) keep the precomputed EdgePre per edge (64 B);
    // long lists keep the four end-point coordinates (32 B) and derive it per candidate: twice the LDS per slot would
    // cost them resident workgroups (47 -> 61 us on the 252-edge scene)
    constexpr bool kPre = PRE;
    constexpr int kSlotDoubles = kPre ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) double s_edge_raw[];  // [slots][kSlotDoubles] (sensor frame)
    auto put_edge = [&](int slot, double x1, double y1, double x2, double y2) {
        if (kPre) {
            reinterpret_cast<EdgePre*>(s_edge_raw)[slot] = edge_pre(x1, y1, x2, y2);
        } else {
            double* e = s_edge_raw + 4 * (size_t)slot;
            e[0] = x1; e[1] = y1; e[2] = x2; e[3] = y2;
        }
    };
    auto get_edge = [&](int slot) -> EdgePre {
        if (kPre) return reinterpret_cast<const EdgePre*>(s_edge_raw)[slot];
        const double* e = s_edge_raw + 4 * (size_t)slot;
        return edge_pre(e[0], e[1], e[2], e[3]);
    };
    uint32_t* const s_span = reinterpret_cast<uint32_t*>(s_edge_raw + (size_t)kSlotDoubles * lv.max_slots);  // [slots] beam span per edge (pack_span)
    // [n_beams] running minimum per beam as the bit pattern of a non-negative double (monotone), then the wave queues
    unsigned long long* const s_best = reinterpret_cast<unsigned long long*>(s_span + ((lv.max_slots + 1) & ~1));
    unsigned long long* const s_mask = s_best + lv.n_beams;   // [n_beams] candidate edges (bit q) of the current 64-edge chunk
    uint32_t* const s_queue = reinterpret_cast<uint32_t*>(s_mask + lv.n_beams);       // [kLidarBlock / 64][queue_len]
    __shared__ int s_qcount[kLidarBlock / 64];
    // Occlusion culling (short static lists, rings described by lv.edge_meta): per edge q its ring | facing, per ring the
    // bits of its back edges.  A beam that passes through the CORE of a front edge of a ring -- its span less one beam at
    // either end: the span is the edge's arc widened by < 0.15 beam, so a core beam lies >= 0.85 beam inside the arc --
    // crosses that edge in its interior, enters the (convex) ring there and leaves it through a back edge at
    // least r_V sin(0.85 beam) sin(gamma) >= 2e-6 m further out (1024 beams) (r_V >= 1 cm the nearer end point's distance, gamma the
    // ring's interior angle there, sin >= 0.05 by the host's choice of rings): the front edge's hit is accepted whenever the
    // back edge's would be, and it is strictly the smaller one -- the back edge's candidate cannot be the beam's minimum
    // and is dropped before the exact arithmetic.  Lists of <= 48 edges (a generated parking lot's 12 quads): bits 0..47 of a
    // beam's candidate word are the edges, bits 48..63 name the rings whose core covers it.
    constexpr bool kCull = kPre && !PARTS;
    constexpr int kCullEdges = 48;
    // (96 bytes: with 32 slots and 360 beams a workgroup's LDS is within 100 bytes of 160 KB / 16 -- 4096 envs in one round
    // of workgroups; 80 bytes more here made it 14 per CU and 17.3 -> 22.4 us.)  A front edge's "16 + ring" travels in its
    // span word instead of a table of its own.
    __shared__ uint32_t s_back[kCull ? 16 : 1];      // per ring: its back edges among edges 0..31
    __shared__ uint32_t s_back_hi[kCull ? 8 : 1];    // ... among edges 32..47, two rings per word
    const int env = blockIdx.x;
    const int tid = threadIdx.x;
    const int A = pv.A;
    const size_t base = (size_t)env * A;
    const double kFar = 1e30;  // an edge nobody can see: yields >= 10 R for every beam, like no edge at all
#ifdef T2D_TIMING
    unsigned long long lt_prev_ = __builtin_readcyclecounter();
    const size_t lt_slot_ = ((size_t)env * (kLidarBlock / 64) + (tid >> 6)) * 16;
#define T2D_LMARK(k)                                                                  \
    do {                                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter();                 \
        if ((tid & 63) == 0) pv.dbg[lt_slot_ + k] = now_ - lt_prev_;                  \
        lt_prev_ = now_;                                                              \
    } while (0)
#else
#define T2D_LMARK(k)
#endif

    // wave priority by progress, as in the step kernel (3 while the edges are staged, 2 in the scatter, 1 in the
    // evaluation, 0 for the output): every workgroup of the launch is resident at once (8 waves per SIMD), and a SIMD is
    // busiest while all of them are alive.  25.2 -> 23.6 us at 4096 x 32 edges, 43.1 -> 40.4 at 252 edges.
    __builtin_amdgcn_s_setprio(3);
    // the first kLidarBlock static edges are fetched before the ego transform is known: their latency overlaps the
    // ego's loads + sincos instead of following them (one record per edge: no vertex -> next-vertex indirection)
    int n_static = 0, v0 = 0;
    float4 first_edge = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cull_maybe = kPre && !PARTS && lv.edge_meta != nullptr;
    uint8_t meta_first = 0xff;
    if (lv.env_vert_cnt) {
        // capacity layout (generated parking lots): env e owns slots [e * max_static_verts, ...), unused ones hold zeros --
        // the edge and its culling byte are fetched beside the count, not behind it (one dependent load less at the front
        // of a chain every workgroup of the launch waits through at the same time)
        v0 = env * lv.max_static_verts;
        if (tid < lv.max_static_verts) {
            first_edge = reinterpret_cast<const float4*>(lv.xy)[v0 + tid];
            if (cull_maybe) meta_first = lv.edge_meta[v0 + tid];
        }
        n_static = lv.env_vert_cnt[env];
    } else if (lv.env_vert_off) {
        v0 = lv.env_vert_off[env];
        n_static = lv.env_vert_off[env + 1] - v0;
        if (tid < n_static) {
            first_edge = reinterpret_cast<const float4*>(lv.xy)[v0 + tid];
            if (cull_maybe) meta_first = lv.edge_meta[v0 + tid];
        }
    }
    // the ego transform, by every lane for itself: the same instructions whether one lane or sixty-four execute them, and
    // no LDS hand-over + workgroup barrier before the first edge can be transformed
    const size_t ie = base + lv.ego_index;
    // (an ego whose pose is not finite scans nothing -- every beam +inf -- and a participant whose pose is not finite is no
    // obstacle: build-defined, like the event kernels and the oracle)
    const float ego_hf = pv.heading[ie], ego_xf = pv.x[ie], ego_yf = pv.y[ie];
    const bool ego_active = ((pv.ids[ie] >> kIdsActiveShift) & 0xff) != 0 && __builtin_isfinite(ego_hf) &&
                            __builtin_isfinite(ego_xf) && __builtin_isfinite(ego_yf);
    double sn, cs;
    sincos_det((double)ego_hf, sn, cs);
    const double ego_x = ego_xf, ego_y = ego_yf;
    const double x_off = -ego_x * cs - ego_y * sn;   // lidar.py:112-113
    const double y_off = ego_x * sn - ego_y * cs;
    T2D_LMARK(0);

    // ---- phase 1a: static polygon edges (vertex v -> next vertex of its ring) ------------------------
    // (the culling's bit layout wants every edge in ONE scatter chunk)
    const bool cull_on = kCull && lv.edge_meta != nullptr && n_static <= (kCullEdges < kChunk ? kCullEdges : kChunk);   // (workgroup-uniform)
    if (kCull && tid < 16) s_back[tid] = 0u;
    if (kCull && tid < 8) s_back_hi[tid] = 0u;
    for (int q = tid; q < n_static; q += kLidarBlock) {
        const float4 ed = q == tid ? first_edge : reinterpret_cast<const float4*>(lv.xy)[v0 + q];
        const double x1 = cs * (double)ed.x + sn * (double)ed.y + x_off;
        const double y1 = -sn * (double)ed.x + cs * (double)ed.y + y_off;
        const double x2 = cs * (double)ed.z + sn * (double)ed.w + x_off;
        const double y2 = -sn * (double)ed.z + cs * (double)ed.w + y_off;
        put_edge(q, x1, y1, x2, y2);
        if (cull_on) {   // (n_static <= 48 < kLidarBlock: q == tid)
            int facing = 0;
            const int2 sp = edge_span(x1, y1, x2, y2, lv.max_range, lv.n_beams, &facing);
            const int ring = meta_first;
            int core = 0;
            if (ring != 0xff) {
                // a front edge has a core when its span is a proper arc of at least three beams
                if (facing > 0 && sp.y >= 2 && sp.y < lv.n_beams) core = 16 + ring;
                if (facing < 0) {
                    if (q < 32) atomicOr(&s_back[ring], 1u << q);
                    else atomicOr(&s_back_hi[ring >> 1], 1u << ((q - 32) + 16 * (ring & 1)));
                }
            }
            s_span[q] = pack_span(sp, core);
        } else {
            s_span[q] = pack_span(edge_span(x1, y1, x2, y2, lv.max_range, lv.n_beams));
        }
    }
    // ---- phase 1b: the other participants' boxes (4 edges each; skipped slots get the far edge) -------
    int n_slots = n_static;
    if (PARTS && lv.include_participants) {
        n_slots += 4 * A;
        for (int j = tid; j < A; j += kLidarBlock) {
            const uint32_t ids = pv.ids[base + j];
            const int type = (ids >> kIdsTypeShift) & 0xff;
            const bool use = j != lv.ego_index && ((ids >> kIdsActiveShift) & 0xff) &&
                             (int)pv.params[T2D_P_SHAPE * T2D_MAX_TYPES + type] == T2D_SHAPE_OBB &&
                             __builtin_isfinite(pv.heading[base + j]) && __builtin_isfinite(pv.x[base + j]) &&
                             __builtin_isfinite(pv.y[base + j]);
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
                const double e0 = use ? vx[k] : kFar, e1 = use ? vy[k] : kFar;
                const double e2 = use ? vx[(k + 1) & 3] : kFar, e3 = use ? vy[(k + 1) & 3] : kFar + 1.0;
                put_edge(n_static + 4 * j + k, e0, e1, e2, e3);
                s_span[n_static + 4 * j + k] = pack_span(use ? edge_span(e0, e1, e2, e3, lv.max_range, lv.n_beams)
                                                             : make_int2(0, -1));
            }
        }
    }
    // (the beam tables of phase 2 are initialised on this side of the barrier: the first chunk then needs none of its own)
    for (int k = tid; k < lv.n_beams; k += kLidarBlock) {
        s_best[k] = 0x7ff0000000000000ull;
        s_mask[k] = 0ull;
    }
    __syncthreads();

    T2D_LMARK(1);
    __builtin_amdgcn_s_setprio(2);
    // ---- phase 2: beams x candidate edges ------------------------------------------------------------------
    // Pass 1 (edge-major scatter of the spans into per-beam candidate masks, see below).  Pass 2: the candidates of the
    // wave's 64 beams are compacted into an LDS queue and evaluated one per lane (dense lanes: the rounds needed
    // are total / 64 instead of the largest per-beam count), each result folded into its beam's minimum by a
    // 64-bit LDS atomic min on the bit pattern (distances are >= 0, so the order of bit patterns is the order of
    // values).  min over the same set of values: the result does not depend on the evaluation order.
    const double R = lv.max_range;
    const int lane = tid & 63, wave = tid >> 6;
    const int queue_len = lv.queue_len;
    uint32_t* const queue = s_queue + wave * queue_len;
    int* const qcount = &s_qcount[wave];
    auto wave_sync = [] {   // LDS traffic of this wave only: its LDS operations complete in order (see t2d_collide.hip)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    const bool scan = ego_active && n_slots > 0;
    const int n_iter = (lv.n_beams + kLidarBlock - 1) / kLidarBlock;
    for (int c0 = 0; c0 < n_slots && scan; c0 += kChunk) {
        // pass 1, edge-major: every edge of the chunk ORs its bit into the masks of the beams of its span (two
        // lanes per edge, alternate beams) -- sum of span lengths instead of beams x edges comparisons
        if (c0 > 0) {
            for (int k = tid; k < lv.n_beams; k += kLidarBlock) s_mask[k] = 0ull;
            __syncthreads();
        }
        {
            // short spans: the edge's two lanes take alternate beams (<= kLongSpan / 2 rounds); a long span (a wall a
            // few metres away covers 100+ beams) would keep the other 126 lanes at the barrier for one pair's 50+
            // rounds, so the wave sweeps each of those together, 64 consecutive beams per round.  Same bits either
            // way.  (Dealing all pairs out evenly through a prefix sum + search was measured too: no faster -- the
            // phase is bound by its longest dependent chain, not by the number of atomics.)
            constexpr int kLongSpan = 24;
            const int q = tid >> 1;
            uint32_t spw = 0u;
            if (c0 + q < n_slots) spw = s_span[c0 + q];
            const int2 sp = make_int2((int)(spw & 0xfffu), (int)((spw >> 12) & 0x1fffu) - 1);
            const int cr = cull_on ? (int)(spw >> 25) : 0;   // (see s_back: a front edge's ring rides in its span word)
            const int last = sp.y < lv.n_beams ? sp.y : lv.n_beams - 1;   // sp.y = -1: invisible
            const bool is_long = last >= kLongSpan;
            // beams i = 1 .. last - 1 of a front edge's span are its core: they also get the bit of its ring
            unsigned long long core_bit = 0ull;
            if (cr) core_bit = 1ull << (kCullEdges + (cr & 15));
            if (!is_long) {
                for (int i = tid & 1; i <= last; i += 2) {
                    int kb = sp.x + i;
                    kb -= kb >= lv.n_beams ? lv.n_beams : 0;
                    atomicOr(&s_mask[kb], (1ull << q) | ((i >= 1 && i <= last - 1) ? core_bit : 0ull));
                }
            }
            unsigned long long todo = __ballot(is_long && !(tid & 1));   // the even lane of a pair speaks for its edge
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const int first = __builtin_amdgcn_readlane(sp.x, src), n_last = __builtin_amdgcn_readlane(last, src);
                const unsigned long long bit = 1ull << ((wave * 64 + src) >> 1);
                const unsigned long long cbit = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(core_bit >> 32), src)) << 32;
                for (int i = lane; i <= n_last; i += 64) {
                    int kb = first + i;
                    kb -= kb >= lv.n_beams ? lv.n_beams : 0;
                    atomicOr(&s_mask[kb], bit | ((i >= 1 && i <= n_last - 1) ? cbit : 0ull));
                }
            }
        }
        __syncthreads();
        T2D_LMARK(2);
        __builtin_amdgcn_s_setprio(1);
        for (int it = 0; it < n_iter; ++it) {
            const int k = tid + it * kLidarBlock;
            unsigned long long m = k < lv.n_beams ? s_mask[k] : 0ull;
            if (cull_on) {   // drop the back edges of every ring whose core covers this beam (see s_back above)
                uint32_t cov = (uint32_t)(m >> kCullEdges), cull_lo = 0u, cull_hi = 0xffff0000u;   // (the ring bits go in any case)
                while (cov) {
                    const int r = __ffs((int)cov) - 1;
                    cov &= cov - 1u;
                    cull_lo |= s_back[r];
                    if (n_static > 32) cull_hi |= (s_back_hi[r >> 1] >> (16 * (r & 1))) & 0xffffu;   // (workgroup-uniform)
                }
                m &= ~((unsigned long long)cull_hi << 32 | cull_lo);
            }
            for (;;) {  // compaction rounds (all 64 lanes take part)
                const int cnt = __popcll(m);
                if (__ballot(cnt > 0) == 0ull) break;
                if (lane == 0) *qcount = 0;
                wave_sync();
                const int off = cnt > 0 ? atomicAdd(qcount, cnt) : queue_len;
                const int room = queue_len - off;
                const int n_emit = room <= 0 ? 0 : (cnt < room ? cnt : room);
                for (int e = 0; e < n_emit; ++e) {
                    const int q = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    queue[off + e] = (uint32_t)k | ((uint32_t)(c0 + q) << 16);
                }
                wave_sync();
                const int total = *qcount;
                const int n_round = total < queue_len ? total : queue_len;
                for (int j = lane; j < n_round; j += 64) {
                    const uint32_t en = queue[j];
                    const int kb = (int)(en & 0xffffu), q = (int)(en >> 16);
                    const double2* bp = reinterpret_cast<const double2*>(lv.beam_pre + 6 * (size_t)kb);
                    const double2 ab = bp[0], bx = bp[1], by = bp[2];
                    const double dd = lidar_edge(ab.x, ab.y, bx.x, bx.y, by.x, by.y, R, get_edge(q));
                    if (dd == dd) atomicMin(&s_best[kb], (unsigned long long)__double_as_longlong(dd));
                }
                wave_sync();
            }
        }
        T2D_LMARK(3);
        if (c0 + kChunk < n_slots) __syncthreads();  // the next chunk clears s_mask
    }
    wave_sync();
    __builtin_amdgcn_s_setprio(0);
    float* o = out + (size_t)env * lv.n_beams;
    for (int k = tid; k < lv.n_beams; k += kLidarBlock) {
        float res = __builtin_inff();
        if (scan) {
            double best = __builtin_sqrt(__longlong_as_double((long long)s_best[k]));   // min of sqrt = sqrt of min, bit for bit
            best = best < 0.0 ? 0.0 : (best > R ? R : best);   // np.clip(0, R)
            res = best == R ? __builtin_inff() : (float)best;
        }
        o[k] = res;
    }
    T2D_LMARK(4);
}

}  // namespace

hipError_t launch_lidar(const PoolView& v, const LidarView& lv_in, float* out, hipStream_t s) {
    LidarView lv = lv_in;
    const bool short_list = lv.max_slots <= 64;   // EdgePre records (64 B) for short lists, end points (32 B) otherwise
    auto lds = [&](int queue_len) {
        return (short_list ? sizeof(EdgePre) : 4 * sizeof(double)) * (size_t)lv.max_slots + 4 * (size_t)((lv.max_slots + 1) & ~1) +
               16 * (size_t)lv.n_beams + 4 * (size_t)queue_len * (kLidarBlock / 64);
    };
    // a generated parking lot's 48 edge slots: half the queue (one more compaction round now and then) keeps the workgroup
    // within a sixteenth of the CU's LDS -- 4096 envs resident at once instead of 3584 and a second round
    lv.queue_len = kLidarQueue;
    if (short_list && lds(kLidarQueue) + kLidarStaticLds > kLidarLdsPerCu16 && lds(kLidarQueue / 2) + kLidarStaticLds <= kLidarLdsPerCu16)
        lv.queue_len = kLidarQueue / 2;
    const size_t dyn = lds(lv.queue_len);
    // (the scan of static obstacles only -- ParkingEnv -- is compiled without the participants' phase: at the 64
    // registers of 8 waves / SIMD that code cost the whole kernel 27 spilled registers, reloaded in the evaluation loop)
    // ParkingEnv's laid-out lots (<= 32 static edges) in pools that fill the GPU with one wave per env: see BLOCK above
    if (short_list && !lv.include_participants && lv.max_slots <= 32 && v.n_env >= kOneWaveMinEnvs && T2D_LIDAR_ONE_WAVE) {
        lv.queue_len = kLidarQueue;
        const size_t dyn1 = (sizeof(EdgePre)) * (size_t)lv.max_slots + 4 * (size_t)((lv.max_slots + 1) & ~1) +
                            16 * (size_t)lv.n_beams + 4 * (size_t)lv.queue_len;
        hipLaunchKernelGGL((lidar_kernel<4, false, 64, true>), dim3(v.n_env), dim3(64), dyn1, s, v, lv, out);
        return hipGetLastError();
    }
    if (short_list && !lv.include_participants)
        hipLaunchKernelGGL((lidar_kernel<8, false>), dim3(v.n_env), dim3(kLidarBlock), dyn, s, v, lv, out);
    else if (short_list)
        hipLaunchKernelGGL((lidar_kernel<8, true>), dim3(v.n_env), dim3(kLidarBlock), dyn, s, v, lv, out);
    else
        hipLaunchKernelGGL((lidar_kernel<4, true>), dim3(v.n_env), dim3(kLidarBlock), dyn, s, v, lv, out);
    return hipGetLastError();
}

}  // namespace t2d
