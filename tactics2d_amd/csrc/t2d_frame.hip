// t2d_frame.hip -- the pack kernel of the Gym-API host path (t2d_step_host, include/t2d.h): one lane per env gathers what
// ParkingEnv.step returns to its caller (envs/parking.py:219-256: observation, reward, terminated / truncated, and the info
// dict of _get_infos :203-217 with the relative pose of _get_relative_pose :190-201) from the pool's columns into ONE
// contiguous frame, so that the host fetches a step's results with one copy instead of one per field.
// HBM-bound by construction and tiny: 4096 envs read ~50 B and write ~90 B each.
#include "t2d_math.h"
#include "t2d_pool.h"

namespace t2d {
namespace {

__global__ __launch_bounds__(64) void frame_pack_kernel(PoolView pv, FrameView fv) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const t2d_frame_layout& L = fv.lay;
    if (e == 0) {
        uint32_t* hdr = reinterpret_cast<uint32_t*>(fv.out);
        hdr[0] = fv.step_count;
        hdr[1] = fv.commit_err ? *fv.commit_err : 0u;
    }
    if (e >= pv.n_env) return;
    const int i = e * pv.A + fv.ego_index;
    const float x = pv.x[i], y = pv.y[i], h = pv.heading[i];
    float* obs = reinterpret_cast<float*>(fv.out + L.off_obs) + 6 * (size_t)e;
    obs[0] = x; obs[1] = y; obs[2] = h; obs[3] = pv.speed[i]; obs[4] = pv.vx[i]; obs[5] = pv.vy[i];
    reinterpret_cast<float*>(fv.out + L.off_reward)[e] = pv.reward[e];
    reinterpret_cast<uint32_t*>(fv.out + L.off_status)[e] = reinterpret_cast<const uint32_t*>(pv.status)[e];
    reinterpret_cast<float*>(fv.out + L.off_iou)[e] = pv.iou[e];
    reinterpret_cast<int32_t*>(fv.out + L.off_frame_ms)[e] = pv.frame_ms[e];
    reinterpret_cast<int32_t*>(fv.out + L.off_cnt_step)[e] = pv.cnt_step[e];
    reinterpret_cast<int32_t*>(fv.out + L.off_episode)[e] = fv.episode ? fv.episode[e] : 0;
    // _get_relative_pose (envs/parking.py:190-201), fp64 like the reference, from the state as stored:
    //   diff_position = np.linalg.norm(target_pose - state.location)        (sqrt(dx*dx + dy*dy) for a 2-vector)
    //   diff_angle    = np.arctan2(target_pose[1] - state.y, target_pose[0] - state.x) - state.heading
    //   diff_heading  = target_heading - state.heading
    double* rel = reinterpret_cast<double*>(fv.out + L.off_rel) + 3 * (size_t)e;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    double d0 = nan, d1 = nan, d2 = nan;
    if (pv.target_c) {
        const double dx = pv.target_c[2 * (size_t)e] - (double)x, dy = pv.target_c[2 * (size_t)e + 1] - (double)y;
        d0 = sqrt(dx * dx + dy * dy);
        d1 = atan2_det(dy, dx) - (double)h;
    }
    const double th = fv.target_heading ? fv.target_heading[e] : nan;
    if (fv.target_heading) d2 = th - (double)h;
    rel[0] = d0; rel[1] = d1; rel[2] = d2;
    if (L.off_target >= 0) {
        float4* t = reinterpret_cast<float4*>(fv.out + L.off_target) + 2 * (size_t)e;
        if (fv.target_quads) {
            const float4* q = reinterpret_cast<const float4*>(fv.target_quads) + 2 * (size_t)e;
            t[0] = q[0]; t[1] = q[1];
        } else if (pv.target_xy) {   // the CCW-normalised quad the IoU kernels see
            const double* q = pv.target_xy + 8 * (size_t)e;
            t[0] = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
            t[1] = make_float4((float)q[4], (float)q[5], (float)q[6], (float)q[7]);
        } else {
            t[0] = t[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        reinterpret_cast<double*>(fv.out + L.off_target_heading)[e] = th;
    }
}

}  // namespace

hipError_t launch_frame_pack(const PoolView& v, const FrameView& fv, hipStream_t s) {
    hipLaunchKernelGGL(frame_pack_kernel, dim3((v.n_env + 63) / 64), dim3(64), 0, s, v, fv);
    return hipGetLastError();
}

}  // namespace t2d
