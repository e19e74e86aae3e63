// t2d_integrate.hip -- wavefront-parallel physics integrator for gfx950 (MI355X).
//
// One lane = one participant; a wave64 = 64 consecutive pool slots (one env's 64 agents,
// 2 x 32-agent envs, or 64 single-agent envs), so every SoA field load/store is one
// coalesced 256-B transaction.  The per-type parameter table is staged once per workgroup
// into LDS in the transposed [column][type] layout (conflict-free: lanes read one column at
// consecutive type slots; equal types broadcast).
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   SingleTrackKinematics.step/_step   physics/single_track_kinematics.py:126-198
//   SingleTrackDynamics.step/_step     physics/single_track_dynamics.py:140-251
//   PointMass.step/_step_newton        physics/point_mass.py:83-175,209-232
//
// Arithmetic contract (DESIGN.md "Precision"): state is stored fp32, every accumulator is
// fp64 in registers, one rounding on store.  All recurrences that feed back (phi, v, beta,
// d_phi) are evaluated with the reference's association and no contraction
// (-ffp-contract=off), so they equal the CPU oracle bit for bit given equal trig inputs.
//   VARIANT 0 "exact": deterministic sin/cos of (phi + beta) every sub-step -- bit-identical
//                      to oracle/t2d_oracle.c in deterministic-trig mode.
//   VARIANT 1 "fast" : cos/sin(phi + beta) advanced by an fp64 rotation recurrence (9th/10th
//                      order Taylor in the sub-step angle, |eps| <= 0.25, else re-seeded);
//                      only x, y, vx, vy (pure outputs) see it; error << 1e-9.
//
// Roofline: nominally HBM-streaming (44-60 B per participant-step), but 20 explicit-Euler
// sub-steps of fp64 math per participant make it VALU-bound; see DESIGN.md.
#include "t2d_integrate_dev.h"

namespace t2d {

namespace {

#ifndef T2D_WIDE_ONLY_KIN
#define T2D_WIDE_ONLY_KIN 1
#endif
constexpr int kBlock = 256;
constexpr int kWideMinParticipants = 8 * 64 * 4 * 256 * 4;   // four per lane and still eight waves' worth of lanes per SIMD
using namespace integ;

template <int VARIANT>
__global__ __launch_bounds__(kBlock) void integrate_kernel(PoolView pv, int interval_ms) {
    __shared__ double s_par[T2D_PARAM_COLS * T2D_MAX_TYPES];  // 6 KiB
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    const bool in_range = i < pv.N;

    // issue the coalesced state loads before the table staging so their latency overlaps it
    uint32_t ids = 0;
    float fx = 0, fy = 0, fh = 0, fv = 0, fa0 = 0, fa1 = 0;
    if (in_range) {
        ids = pv.ids[i];
        fx = pv.x[i];
        fy = pv.y[i];
        fh = pv.heading[i];
        fv = pv.speed[i];
        fa0 = pv.act0[(size_t)i * pv.act_stride];
        fa1 = pv.act1[(size_t)i * pv.act_stride];
        if (pv.idm_ctrl && pv.idm_ctrl[i] != T2D_IDM_NONE) {  // IDM lane while caller actions are bound
            fa0 = pv.own_act0[i];
            fa1 = pv.own_act1[i];
        }
    }
    // table staging: all three 8-B loads per thread in flight together (one exposed latency)
    static_assert(T2D_PARAM_COLS * T2D_MAX_TYPES == 3 * kBlock, "staging assumes 3 loads per thread");
    const double t0 = pv.params[tid], t1 = pv.params[tid + kBlock], t2 = pv.params[tid + 2 * kBlock];
    s_par[tid] = t0;
    s_par[tid + kBlock] = t1;
    s_par[tid + 2 * kBlock] = t2;
    __syncthreads();

    const bool active = in_range && ((ids >> kIdsActiveShift) & 0xffu);
    if (!active) return;
    const int type = (ids >> kIdsTypeShift) & 0xff;
    const int model = (ids >> kIdsModelShift) & 0xff;
    if (model >= T2D_MODEL_DRIFT) return;  // SingleTrackDrift, PointMass(backend="euler"): integrated by drift_kernel (t2d_drift.hip)
    auto P = [&](int col) -> double { return s_par[col * T2D_MAX_TYPES + type]; };

    StepOut o;
    if (model == T2D_MODEL_KINEMATICS) {
        o = step_kinematics<VARIANT, true>(P, (double)fx, (double)fy, (double)fh, (double)fv, (double)fa0,
                                     (double)fa1, interval_ms);
    } else if (model == T2D_MODEL_DYNAMICS) {
        o = step_dynamics<VARIANT>(P, (double)fx, (double)fy, (double)fh, (double)fv, (double)fa0,
                                   (double)fa1, interval_ms);
    } else {
        o = step_pointmass(P, (double)fx, (double)fy, (double)pv.vx[i], (double)pv.vy[i],
                           (double)fa0, (double)fa1, pv.interval_s);
    }
    pv.x[i] = (float)o.x;
    pv.y[i] = (float)o.y;
    pv.heading[i] = (float)o.heading;
    pv.speed[i] = (float)o.speed;
    if (o.has_velocity && (model == T2D_MODEL_POINTMASS || (pv.out_mask & T2D_OUT_VELOCITY))) {
        pv.vx[i] = (float)o.vx;
        pv.vy[i] = (float)o.vy;
    }
    if (pv.out_mask & T2D_OUT_APPLIED) {
        pv.applied0[i] = (float)o.app0;
        pv.applied1[i] = (float)o.app1;
    }
}

// The same step, FOUR consecutive participants per lane: pools large enough to stay wide afterwards (launch_integrate) take
// their state through 16-byte loads and stores -- a quarter of the memory instructions, four times the bytes in flight per
// wave.  At 4 M point masses the one-per-lane kernel moved 3.5 TB/s of its ~60 B per participant (72 us, 0.32 of the 8 TB/s
// peak on the 44-B figure); the model's arithmetic is the same per participant, the results are the same bits
// (tests/test_gpu_physics.py holds this kernel against the fused step's one-per-lane integrator).
// ONLY = T2D_MODEL_POINTMASS: every active participant of the pool is a (newton) point mass (the host knows: t2d_reset keeps the set of types in
// use) -- the instantiation carries that model alone: ~50 registers instead of the 158 of the three bodies unrolled four times,
// i.e. eight waves per SIMD instead of three, and every load is issued before the first one returns (nothing waits for the ids
// word to learn the model).  The point mass is the one model whose step is cheap enough to be memory-bound.
template <int VARIANT, int ONLY = -1>   // ONLY: the one model every active participant has (T2D_MODEL_*), or -1 = any
__global__ __launch_bounds__(kBlock) void integrate_wide_kernel(PoolView pv, int interval_ms) {
    constexpr bool ONLY_PM = ONLY == T2D_MODEL_POINTMASS, ONLY_KIN = ONLY == T2D_MODEL_KINEMATICS;
    __shared__ double s_par[T2D_PARAM_COLS * T2D_MAX_TYPES];
    const int tid = threadIdx.x;
    const int i4 = blockIdx.x * kBlock + tid;          // index of the lane's group of four
    const bool in_range = 4 * (size_t)i4 < (size_t)pv.N;   // (N is a multiple of 4: launch_integrate)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    u32x4 ids = {0u, 0u, 0u, 0u};
    f32x4 fx = {0, 0, 0, 0}, fy = fx, fh = fx, fv = fx, fa0 = fx, fa1 = fx, fvx = fx, fvy = fx;
    if (in_range) {   // what every model reads; heading / speed (single-track models) and vx / vy (point mass) follow the ids word
        ids = reinterpret_cast<const u32x4*>(pv.ids)[i4];
        fx = reinterpret_cast<const f32x4*>(pv.x)[i4];
        fy = reinterpret_cast<const f32x4*>(pv.y)[i4];
        fa0 = reinterpret_cast<const f32x4*>(pv.act0)[i4];
        fa1 = reinterpret_cast<const f32x4*>(pv.act1)[i4];
        if constexpr (ONLY_PM) {
            fvx = reinterpret_cast<const f32x4*>(pv.vx)[i4];
            fvy = reinterpret_cast<const f32x4*>(pv.vy)[i4];
        }
        if constexpr (ONLY_KIN) {
            fh = reinterpret_cast<const f32x4*>(pv.heading)[i4];
            fv = reinterpret_cast<const f32x4*>(pv.speed)[i4];
        }
    }
    static_assert(T2D_PARAM_COLS * T2D_MAX_TYPES == 3 * kBlock, "staging assumes 3 loads per thread");
    const double t0 = pv.params[tid], t1 = pv.params[tid + kBlock], t2 = pv.params[tid + 2 * kBlock];
    s_par[tid] = t0;
    s_par[tid + kBlock] = t1;
    s_par[tid + 2 * kBlock] = t2;
    __syncthreads();
    if (!in_range) return;
    // a point mass's state is (x, y, vx, vy) -- it reads neither heading nor speed --, a single-track model's (x, y, heading,
    // speed): each pair is fetched only where one of the lane's four needs it (wave-uniform in pools sorted by kind).  For a
    // point-mass pool that is 8 of 36 bytes read per participant not read: 61 -> 5x us at 4 M.
    bool any_pm = false, any_st = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int model = (ids[k] >> kIdsModelShift) & 0xff;
        const bool act = (ids[k] >> kIdsActiveShift) & 0xffu;
        any_pm |= act && model == T2D_MODEL_POINTMASS;
        any_st |= act && model < T2D_MODEL_POINTMASS;
    }
    if constexpr (ONLY < 0) {
        if (any_st) {
            fh = reinterpret_cast<const f32x4*>(pv.heading)[i4];
            fv = reinterpret_cast<const f32x4*>(pv.speed)[i4];
        }
        if (any_pm) {
            fvx = reinterpret_cast<const f32x4*>(pv.vx)[i4];
            fvy = reinterpret_cast<const f32x4*>(pv.vy)[i4];
        }
    }
    f32x4 ovx = fvx, ovy = fvy, oa0 = {0, 0, 0, 0}, oa1 = oa0;
    uint32_t done = 0u, has_vel = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool active = (ids[k] >> kIdsActiveShift) & 0xffu;
        const int type = (ids[k] >> kIdsTypeShift) & 0xff;
        const int model = (ids[k] >> kIdsModelShift) & 0xff;
        if (!active || model >= T2D_MODEL_DRIFT) continue;
        auto P = [&](int col) -> double { return s_par[col * T2D_MAX_TYPES + type]; };
        StepOut o;
        if (ONLY_KIN || (ONLY < 0 && model == T2D_MODEL_KINEMATICS)) {
            o = step_kinematics<VARIANT, true>(P, (double)fx[k], (double)fy[k], (double)fh[k], (double)fv[k], (double)fa0[k],
                                               (double)fa1[k], interval_ms);
        } else if (ONLY < 0 && model == T2D_MODEL_DYNAMICS) {
            o = step_dynamics<VARIANT>(P, (double)fx[k], (double)fy[k], (double)fh[k], (double)fv[k], (double)fa0[k],
                                       (double)fa1[k], interval_ms);
        } else {
            o = step_pointmass(P, (double)fx[k], (double)fy[k], (double)fvx[k], (double)fvy[k], (double)fa0[k], (double)fa1[k],
                               pv.interval_s);
        }
        fx[k] = (float)o.x;
        fy[k] = (float)o.y;
        fh[k] = (float)o.heading;
        fv[k] = (float)o.speed;
        oa0[k] = (float)o.app0;
        oa1[k] = (float)o.app1;
        done |= 1u << k;
        if (o.has_velocity && (model == T2D_MODEL_POINTMASS || (pv.out_mask & T2D_OUT_VELOCITY))) {
            ovx[k] = (float)o.vx;
            ovy[k] = (float)o.vy;
            has_vel |= 1u << k;
        }
    }
    if (done == 0u) return;
    // whole-vector stores where all four were stepped (the usual case); element stores otherwise -- a slot that was not
    // stepped (inactive, drift model) keeps every byte it had
    auto put = [&](float* col, const f32x4& v, uint32_t mask) {
        if (mask == 15u) {
            reinterpret_cast<f32x4*>(col)[i4] = v;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (mask >> k & 1u) col[4 * (size_t)i4 + k] = v[k];
        }
    };
    put(pv.x, fx, done);
    put(pv.y, fy, done);
    put(pv.heading, fh, done);
    put(pv.speed, fv, done);
    if (has_vel) {
        put(pv.vx, ovx, has_vel);
        put(pv.vy, ovy, has_vel);
    }
    if (pv.out_mask & T2D_OUT_APPLIED) {
        put(pv.applied0, oa0, done);
        put(pv.applied1, oa1, done);
    }
}

// verify_state: the "very rough check" of a candidate state against the pool's current (= last) state.
//   SingleTrackKinematics.verify_state physics/single_track_kinematics.py:200-250, SingleTrackDynamics
//   :253-306 (same check), PointMass.verify_state physics/point_mass.py:234-259.  Oracle: t2do_verify_state.
struct VerifyArgs {
    const float *x, *y, *heading, *speed;  // candidate state [N], device
    uint8_t* valid;                        // out [N]: 1 = plausible (inactive participants: 1)
    int interval_ms;
};

__global__ __launch_bounds__(kBlock) void verify_kernel(PoolView pv, VerifyArgs a) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= pv.N) return;
    const uint32_t ids = pv.ids[i];
    uint8_t ok = 1;
    if (((ids >> kIdsActiveShift) & 0xffu) && a.interval_ms != 0) {
        const int type = (ids >> kIdsTypeShift) & 0xff;
        const int model = (ids >> kIdsModelShift) & 0xff;
        auto P = [&](int col) -> double { return pv.params[col * T2D_MAX_TYPES + type]; };
        const int flags = (int)P(T2D_P_RANGE_FLAGS);
        const double dt = (double)a.interval_ms / 1000;
        const double lx = (double)pv.x[i], ly = (double)pv.y[i];
        const double x = (double)a.x[i], y = (double)a.y[i];
        if (model == T2D_MODEL_POINTMASS) {
            const double den = 2 / (dt * dt);
            const double ax = (x - lx - (double)pv.vx[i] * dt) * den;
            const double ay = (y - ly - (double)pv.vy[i] * dt) * den;
            if (flags & T2D_RANGE_ACCEL) {
                const double acc = __builtin_sqrt(ax * ax + ay * ay);
                if (!(P(T2D_P_ACCEL_LO) <= acc && acc <= P(T2D_P_ACCEL_HI))) ok = 0;
            }
        } else if ((flags & 7) == 7) {
            const double lh = (double)pv.heading[i], lv = (double)pv.speed[i];
            const double h = (double)a.heading[i], v = (double)a.speed[i];
            const double wb = P(T2D_P_WB), k = P(T2D_P_LR) / wb;
            const double st[2] = {P(T2D_P_STEER_LO), P(T2D_P_STEER_HI)};
            const double ac[2] = {P(T2D_P_ACCEL_LO), P(T2D_P_ACCEL_HI)};
            const double vlo = P(T2D_P_SPEED_LO), vhi = P(T2D_P_SPEED_HI);
            double hr[2], sr[2], xr[2], yr[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const double beta = atan_det(k * st[e]);
                double sb, cb, sh, ch;
                sincos_det(beta, sb, cb);
                hr[e] = mod_two_pi(lh + lv / wb * sb * dt);
                sr[e] = clipd(lv + ac[e] * dt, vlo, vhi);
                sincos_det(lh + beta, sh, ch);
                xr[e] = lx + sr[e] * ch * dt;
                yr[e] = ly + sr[e] * sh * dt;
            }
            if (hr[0] < hr[1] && !(hr[0] <= h && h <= hr[1])) ok = 0;
            if (hr[0] > hr[1] && !(hr[0] <= h || h <= hr[1])) ok = 0;
            if (!(sr[0] <= v && v <= sr[1])) ok = 0;
            if (!(xr[0] < x && x < xr[1]) || !(yr[0] < y && y < yr[1])) ok = 0;
        }
    }
    a.valid[i] = ok;
}

}  // namespace

hipError_t launch_verify(const PoolView& v, const float* x, const float* y, const float* heading, const float* speed,
                         int interval_ms, uint8_t* valid, hipStream_t s) {
    VerifyArgs a{x, y, heading, speed, valid, interval_ms};
    hipLaunchKernelGGL(verify_kernel, dim3((v.N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, v, a);
    return hipGetLastError();
}

// Column T2D_P_SUBSTEPS of the device table: (interval // delta_t) << 16 | interval % delta_t per type, for the interval of the
// launches that follow on this stream (t2d_api.hip launches it when the interval changes; rows of the drift model own the
// column).  The integrators read it instead of dividing per lane.
namespace {
__global__ void derive_kernel(double* params, int n_types, int interval_ms) {
    const int t = threadIdx.x;
    if (t >= n_types || (int)params[T2D_P_MODEL * T2D_MAX_TYPES + t] == T2D_MODEL_DRIFT) return;
    const int delta_t = (int)params[T2D_P_DELTA_T_MS * T2D_MAX_TYPES + t];
    const int n = interval_ms / delta_t, rem = interval_ms - n * delta_t;
    params[T2D_P_SUBSTEPS * T2D_MAX_TYPES + t] = (double)(n * 65536 + rem);
}
}  // namespace
hipError_t launch_derive(double* params, int n_types, int interval_ms, hipStream_t s) {
    hipLaunchKernelGGL(derive_kernel, dim3(1), dim3(T2D_MAX_TYPES), 0, s, params, n_types, interval_ms);
    return hipGetLastError();
}

hipError_t launch_integrate(const PoolView& v, int interval_ms, int variant, bool allow_wide, int only_model, hipStream_t s) {
    // pools that still put eight waves' worth of lanes on every SIMD with four participants per lane (>= 2 M on an MI355X:
    // at 1 M the wide kernel's 158 registers -- three waves per SIMD -- made every model slower, 21.8 -> 24.0 us), contiguous
    // actions, no IDM lanes reading the pool's own fields beside caller-owned ones, every column 16-byte aligned
    auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (allow_wide && v.N >= kWideMinParticipants && (v.N & 3) == 0 && v.act_stride == 1 && !v.idm_ctrl && aligned(v.act0) && aligned(v.act1) &&
        aligned(v.x) && aligned(v.y) && aligned(v.heading) && aligned(v.speed) && aligned(v.vx) && aligned(v.vy) &&
        aligned(v.applied0) && aligned(v.applied1) && aligned(v.ids)) {
        const int grid4 = (v.N / 4 + kBlock - 1) / kBlock;
        if (only_model == T2D_MODEL_POINTMASS) {   // (one arithmetic for both variants: the point mass has no trig recurrence to approximate)
            hipLaunchKernelGGL((integrate_wide_kernel<1, T2D_MODEL_POINTMASS>), dim3(grid4), dim3(kBlock), 0, s, v, interval_ms);
            return hipGetLastError();
        }
        if (only_model == T2D_MODEL_KINEMATICS && T2D_WIDE_ONLY_KIN) {
            if (variant == 0) hipLaunchKernelGGL((integrate_wide_kernel<0, T2D_MODEL_KINEMATICS>), dim3(grid4), dim3(kBlock), 0, s, v, interval_ms);
            else hipLaunchKernelGGL((integrate_wide_kernel<1, T2D_MODEL_KINEMATICS>), dim3(grid4), dim3(kBlock), 0, s, v, interval_ms);
            return hipGetLastError();
        }
        if (variant == 0)
            hipLaunchKernelGGL(integrate_wide_kernel<0>, dim3(grid4), dim3(kBlock), 0, s, v, interval_ms);
        else
            hipLaunchKernelGGL(integrate_wide_kernel<1>, dim3(grid4), dim3(kBlock), 0, s, v, interval_ms);
        return hipGetLastError();
    }
    const int grid = (v.N + kBlock - 1) / kBlock;
    if (variant == 0)
        hipLaunchKernelGGL(integrate_kernel<0>, dim3(grid), dim3(kBlock), 0, s, v, interval_ms);
    else
        hipLaunchKernelGGL(integrate_kernel<1>, dim3(grid), dim3(kBlock), 0, s, v, interval_ms);
    return hipGetLastError();
}

}  // namespace t2d
