// t2d_idm.hip -- on-device scripted agents: IDM car-following (scope row f3).
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   IDMController.step                controller/idm_controller.py:59-93   (free flow | car following, np.clip)
//   IDMController._idm_acceleration   controller/idm_controller.py:95-141  (s*, (v/v_des)^delta, (s*/s)^2)
// The reference leaves the choice of `leading_state` to its caller and ships no caller; the leader rule
// here is BUILD-DEFINED (oracle t2do_idm is the definition): among the other active participants of
// the env, those ahead (longitudinal offset 0 < lon <= horizon along the own heading) inside the own
// corridor (|lateral offset| <= lane_half_width), the one with the smallest lon, lowest index on ties.
// A caller that knows the leader (the reference's calling convention) passes it per participant instead.
//
// One lane per participant, EPB = 256 / A_pad whole envs per workgroup.  The env's (x, y, speed, active)
// go through LDS once; every controlled lane sweeps its env's list with wave-uniform LDS reads (broadcast,
// conflict free), ~12 fp64 operations per candidate, then evaluates the IDM law once.  Output: accel ->
// the pool's act0 field, steer 0 -> its act1 field (the reference returns (steering, acceleration); the physics
// models take (accel, steer)), leader index -> T2D_F_LEADER.  HBM: 17 B read + 12 B written per participant.
#include "t2d_idm_dev.h"

namespace t2d {

namespace {

using namespace idm;

constexpr int kIdmBlock = 256;

// act0_own / act1_own: the POOL's action fields (T2D_F_ACT0 / ACT1) -- never caller-owned memory bound with
// t2d_bind_actions; while a binding is in effect the integrators take the controlled lanes' actions from there
__global__ __launch_bounds__(kIdmBlock) void idm_kernel(PoolView pv, IdmView iv, const int32_t* forced, float* act0_own,
                                                        float* act1_own, int log2A) {
    // (x, y) as fp64 pairs, NaN for inactive slots: every comparison of the sweep is then false for
    // them, and a participant never selects itself (its own offset is exactly 0, not > 0)
    __shared__ double2 s_xy[kIdmBlock];
    __shared__ float s_v[kIdmBlock];
    const int tid = threadIdx.x;
    // every kernel argument of the load phase in one scalar round trip (see collide_kernel)
    // (global address space kept through the asm: plain pointers come out of it generic, i.e. as flat loads, which
    // count against the LDS wait counter as well)
    const T2D_GLOBAL uint32_t* a_ids = (const T2D_GLOBAL uint32_t*)pv.ids;
    const T2D_GLOBAL float *a_x = (const T2D_GLOBAL float*)pv.x, *a_y = (const T2D_GLOBAL float*)pv.y,
                           *a_h = (const T2D_GLOBAL float*)pv.heading, *a_v = (const T2D_GLOBAL float*)pv.speed;
    const T2D_GLOBAL uint8_t* a_ctrl = (const T2D_GLOBAL uint8_t*)iv.ctrl_id;
    const T2D_GLOBAL double* a_rows = (const T2D_GLOBAL double*)iv.rows;
    int a_n_env = pv.n_env, a_A = pv.A;
    asm volatile("" : "+s"(a_ids), "+s"(a_x), "+s"(a_y), "+s"(a_h), "+s"(a_v), "+s"(a_ctrl), "+s"(a_rows), "+s"(a_n_env), "+s"(a_A));
    const int A_pad = 1 << log2A;
    const int epb = kIdmBlock >> log2A;
    const int env_local = tid >> log2A;
    const int agent = tid & (A_pad - 1);
    const int env = blockIdx.x * epb + env_local;
    const bool valid = env < a_n_env && agent < a_A;
    const int idx = valid ? env * a_A + agent : 0;
    float fx = 0, fy = 0, fh = 0, fv = 0;
    uint32_t ids = 0;
    int ctrl = T2D_IDM_NONE;
    if (valid) {
        ids = a_ids[idx];
        fx = a_x[idx];
        fy = a_y[idx];
        fh = a_h[idx];
        fv = a_v[idx];
        ctrl = a_ctrl[idx];
    }
    const bool active = valid && ((ids >> kIdsActiveShift) & 0xffu);
    // the controller's row and the heading's sine / cosine before the barrier: the row is a second dependent round trip
    // to memory (after ctrl_id) and every wave of the launch reaches it at the same time -- it overlaps the LDS fill
    const bool controlled = active && ctrl != T2D_IDM_NONE && ctrl < iv.n_ctrl;
    IdmRow c{};
    double sn = 0.0, cs = 1.0;
    if (controlled) c = load_row(a_rows + (size_t)ctrl * T2D_IDM_COLS);
    const double qnan = __builtin_nan("");
    s_xy[tid] = active ? make_double2((double)fx, (double)fy) : make_double2(qnan, qnan);
    s_v[tid] = fv;
    if (controlled) sincos_det((double)fh, sn, cs);
    __syncthreads();
    if (!valid) return;
    int lead = -1;
    if (controlled) {
        const int base = env_local << log2A;
        const double x0 = (double)fx, y0 = (double)fy;
        const int want = forced ? forced[idx] : T2D_IDM_LEADER_SEARCH;
        if (want >= 0 && want < pv.A && want != agent && s_xy[base + want].x == s_xy[base + want].x)
            lead = want;  // the caller's leading_state
        if (want == T2D_IDM_LEADER_SEARCH) lead = find_leader<true>([&](int j) { return s_xy[base + j]; }, pv.A, c, x0, y0, sn, cs);
        double dx = 0.0, dy = 0.0, vl = 0.0;
        if (lead >= 0) {
            dx = s_xy[base + lead].x - x0;
            dy = s_xy[base + lead].y - y0;
            vl = (double)s_v[base + lead];
        }
        act0_own[idx] = (float)idm_law(c, (double)fv, lead >= 0, dx, dy, vl);
        act1_own[idx] = 0.0f;
    }
    iv.leader[idx] = lead;
}

}  // namespace

hipError_t launch_idm(const PoolView& v, const IdmView& iv, const int32_t* forced_leader, float* act0_own, float* act1_own,
                      hipStream_t s) {
    int log2A = 0;
    while ((1 << log2A) < v.A) ++log2A;
    const int epb = kIdmBlock >> log2A;
    hipLaunchKernelGGL(idm_kernel, dim3((v.n_env + epb - 1) / epb), dim3(kIdmBlock), 0, s, v, iv, forced_leader, act0_own, act1_own, log2A);
    return hipGetLastError();
}

}  // namespace t2d
