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

constexpr int kBlock = 256;
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
        fa0 = pv.act0[i];
        fa1 = pv.act1[i];
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
    auto P = [&](int col) -> double { return s_par[col * T2D_MAX_TYPES + type]; };

    StepOut o;
    if (model == T2D_MODEL_KINEMATICS) {
        o = step_kinematics<VARIANT>(P, (double)fx, (double)fy, (double)fh, (double)fv, (double)fa0,
                                     (double)fa1, interval_ms);
    } else if (model == T2D_MODEL_DYNAMICS) {
        o = step_dynamics<VARIANT>(P, (double)fx, (double)fy, (double)fh, (double)fv, (double)fa0,
                                   (double)fa1, interval_ms);
    } else {
        o = step_pointmass(P, (double)fx, (double)fy, (double)pv.vx[i], (double)pv.vy[i],
                           (double)fa0, (double)fa1, interval_ms);
    }
    pv.x[i] = (float)o.x;
    pv.y[i] = (float)o.y;
    pv.heading[i] = (float)o.heading;
    pv.speed[i] = (float)o.speed;
    if (o.has_velocity) {
        pv.vx[i] = (float)o.vx;
        pv.vy[i] = (float)o.vy;
    }
    pv.applied0[i] = (float)o.app0;
    pv.applied1[i] = (float)o.app1;
}

}  // namespace

hipError_t launch_integrate(const PoolView& v, int interval_ms, int variant, hipStream_t s) {
    const int grid = (v.N + kBlock - 1) / kBlock;
    if (variant == 0)
        hipLaunchKernelGGL(integrate_kernel<0>, dim3(grid), dim3(kBlock), 0, s, v, interval_ms);
    else
        hipLaunchKernelGGL(integrate_kernel<1>, dim3(grid), dim3(kBlock), 0, s, v, interval_ms);
    return hipGetLastError();
}

}  // namespace t2d
