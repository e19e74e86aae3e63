// t2d_api.hip -- host side of libt2d_hip.so: pool lifetime, uploads, launches (C ABI of
// include/t2d.h).  No CPU compute fallback exists: every entry point that needs the GPU
// returns T2D_ERR_HIP with the HIP error text when the device / runtime is unavailable.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

// RCCL: types and enums only -- the library is opened with dlopen when a communicator is asked for.  A single-GPU
// install without the rccl development headers still builds: the handful of ABI types used here is declared locally then
// (public NCCL/RCCL ABI: the id is 128 opaque bytes, results and data types are plain enums, ncclUint32 = 3).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3 } ncclDataType_t;
}
#endif

#include <algorithm>
#include <limits>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "t2d_pool.h"
#include "t2d_host.h"
#ifndef T2D_LOOP_MAX_WGS_PER_CU
#define T2D_LOOP_MAX_WGS_PER_CU 2
#endif
#ifdef T2D_DEBUG_HOOKS
#include "../../include/t2d_debug.h"
#endif

namespace t2d {
namespace host {
thread_local std::string g_create_err;
int fail(t2d_pool* p, int code, const std::string& msg) {
    if (p) p->err = msg;
    else g_create_err = msg;
    return code;
}
const std::string& create_error() { return g_create_err; }
}  // namespace host
}  // namespace t2d
using namespace t2d::host;
namespace {

// RCCL, opened on demand (t2d_comm_unique_id / t2d_comm_init): libt2d_hip.so itself does not link it, so a single-GPU
// user never loads it, and a process that already holds a copy (torch ships one) shares that copy by soname.
struct Rccl {
    bool tried = false, ok = false;
    std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;       // optional: t2d_comm_info reads the world back
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
Rccl& rccl() {
    static Rccl r;
    if (r.tried) return r;
    r.tried = true;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        r.err = std::string("cannot open librccl: ") + dlerror();
        return r;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(h, "ncclCommUserRank");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    if (!r.ok) r.err = "librccl lacks an expected symbol";
    return r;
}

// Streams the pool has launched on since it was last quiesced.  Set-up calls, up/downloads and t2d_sync wait for
// THESE (plus the pool's own internal streams), not for the device: another pool's env group or a policy running on
// other streams is not stalled by them (no hidden device-wide syncs).
void touch(t2d_pool* p, hipStream_t s) {
    for (int k = 0; k < p->n_live_streams; ++k)
        if (p->live_streams[k] == s) return;
    if (p->n_live_streams < t2d_pool::kMaxLiveStreams) p->live_streams[p->n_live_streams++] = s;
    else p->live_overflow = true;   // more distinct streams than tracked: the next quiesce falls back to the device
}

}  // namespace
namespace t2d { hipError_t launch_chain_rollback(const PoolView& v, const uint32_t* ckpt, hipStream_t s); }
namespace {

hipError_t quiesce(t2d_pool* p) {
    hipError_t e = hipSuccess;
    bool whole_device = p->live_overflow;
    for (int k = 0; k < p->n_live_streams && !whole_device; ++k)
        if (hipStreamSynchronize(p->live_streams[k]) != hipSuccess) {
            // a caller's stream that no longer exists (stepped on a temporary stream and destroyed it): its work cannot be
            // waited for by handle any more -- wait for the device instead of failing or skipping the rest
            (void)hipGetLastError();
            whole_device = true;
        }
    if (whole_device) {
        e = hipDeviceSynchronize();
    } else {   // the pool's own streams, whatever happened above
        if (p->scene_stream) e = hipStreamSynchronize(p->scene_stream);
        if (p->gather_stream) {
            const hipError_t e2 = hipStreamSynchronize(p->gather_stream);
            if (e == hipSuccess) e = e2;
        }
    }
    p->n_live_streams = 0;
    p->live_overflow = false;
    if (e == hipSuccess && p->scene_commit_used && p->scene.commit_err) {
        uint32_t err = 0;
        e = hipMemcpy(&err, p->scene.commit_err, sizeof(err), hipMemcpyDeviceToHost);
        p->scene_commit_used = false;
        if (e == hipSuccess && err) {
            (void)hipMemset(p->scene.commit_err, 0, sizeof(err));
            p->scene_commit_failed = true;
        }
    }
    if (e == hipSuccess && p->chain_used) {   // a chained launch reports a broken hand-off through two words in device memory
        uint32_t err[2] = {0, 0};   // {1: a bounded wait ran out | 2: producer and consumer on different XCDs, ckpt_tag of the fragment}
        e = hipMemcpy(err, p->d_chain + p->chain_slots, sizeof(err), hipMemcpyDeviceToHost);
        p->chain_used = false;
        if (e == hipSuccess && err[0]) {
            (void)hipMemset(p->d_chain + p->chain_slots, 0, sizeof(err));
            p->chain_failed = true;
            p->chain_err_code = (int)err[0];
            p->chain_steps = false;   // whatever broke the hand-off (a wait that ran out, workgroups of one env set on two XCDs) may do so again
            p->chain_sig = 0;         // (the counters are in no known state: a later chained launch starts them afresh)
            p->chain_rolled_back = false;
            if (p->ckpt_armed && p->d_ckpt) {
                // CHAIN form: whatever the failed fragment and the fragments enqueued behind it computed is discarded -- its
                // checkpoint holds what it started from (complete: its own failure never stops its step-0 stores, a later
                // fragment never overwrites it): put the pool back there.  The tag is the low half of the step count at
                // the fragment's start.
                const uint32_t back = (uint32_t)p->step_count - err[1];
                e = t2d::launch_chain_rollback(p->v, p->d_ckpt, nullptr);
                if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
                if (e == hipSuccess) {
                    p->step_count -= (long long)back;
                    p->chain_rolled_back = true;
                    p->chain_rollback_step = p->step_count;
                    p->v.record = (uint2*)p->field_ptr[T2D_F_RECORD] +
                                  (size_t)((p->step_count + T2D_RECORD_RING - 1) % T2D_RECORD_RING) * p->v.n_env;
                }
            }
        }
    }
    return e;
}

// A failed chained launch is reported ONCE, by the first t2d_sync / t2d_download / t2d_step_n after it; the pool then goes on
// with plain launches (t2d_set_step_chaining turns chaining back on).
int report_chain_failure(t2d_pool* p) {
    if (p->scene_commit_failed) {   // (scene_commit_kernel: cannot happen while the refill cadence below holds)
        p->scene_commit_failed = false;
        return fail(p, T2D_ERR_STATE, "scene regeneration: an env whose episode ended found no staged scene for its next episode and "
                                      "kept its old one -- call t2d_parking_scenes again before stepping on");
    }
    if (!p->chain_failed) return T2D_OK;
    p->chain_failed = false;
    const std::string why = p->chain_err_code == 2 ? "two workgroups of one env set ran on different XCDs"
                                                   : "a workgroup's bounded wait for its previous step ran out";
    if (p->chain_rolled_back)
        return fail(p, T2D_ERR_STATE, "a t2d_step_n launch failed (" + why + "): the pool was rolled back to step " +
                                          std::to_string(p->chain_rollback_step) + ", the start of the failed fragment -- flags, status, "
                                          "reward and records hold nothing valid until the next step; re-issue the steps from there "
                                          "(they take plain launches now)");
    return fail(p, T2D_ERR_STATE, "a t2d_step_n launch failed (" + why + "): its results are invalid -- t2d_reset / t2d_restore / "
                                      "upload the state before stepping on (plain launches from now on)");
}

size_t field_elem_bytes(int f) {
    switch (f) {
        case T2D_F_RECORD: return 8 * T2D_RECORD_RING;  // ring slots x {u32 reward bits, u32 status word} per env
        case T2D_F_STATUS: return 4;  // 4 x u8 per env
        default: return 4;
    }
}
bool field_per_env(int f) { return f >= T2D_F_ENV_FLAGS && f < T2D_F_LEADER; }

// (Re)initialise the per-env IoU / shaping state: NoAction forgets its pose, _max_iou = -inf,
// _min_dist_to_target = ||start - target centroid|| (envs/parking.py:280-296), inf without targets.
int init_iou_state(t2d_pool* p, const uint8_t* env_mask, const float* hx, const float* hy) {
    const int E = p->v.n_env, A = p->v.A, ego = p->status_cfg.ego_index;
    std::vector<double> maxi(E), mind(E), tc;
    std::vector<uint8_t> lv(E);
    std::vector<int32_t> cna(E);
    std::vector<float> iou(E);
    const bool partial = env_mask != nullptr;
    if (partial) {
        T2D_HIP(p, hipMemcpy(maxi.data(), p->d_max_iou, sizeof(double) * E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(mind.data(), p->d_min_dist, sizeof(double) * E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(lv.data(), p->d_last_valid, E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(cna.data(), p->v.cnt_na, 4 * (size_t)E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(iou.data(), p->v.iou, 4 * (size_t)E, hipMemcpyDeviceToHost));
    }
    if (p->have_target) {
        tc.resize(2 * (size_t)E);
        T2D_HIP(p, hipMemcpy(tc.data(), p->d_target_c, sizeof(double) * 2 * E, hipMemcpyDeviceToHost));
    }
    for (int e = 0; e < E; ++e) {
        if (partial && !env_mask[e]) continue;
        maxi[e] = -INFINITY;
        lv[e] = 0;
        cna[e] = 0;
        iou[e] = NAN;
        if (p->have_target) {
            const double dx = (double)hx[(size_t)e * A + ego] - tc[2 * (size_t)e];
            const double dy = (double)hy[(size_t)e * A + ego] - tc[2 * (size_t)e + 1];
            mind[e] = sqrt(dx * dx + dy * dy);
        } else {
            mind[e] = INFINITY;
        }
    }
    T2D_HIP(p, hipMemcpy(p->d_max_iou, maxi.data(), sizeof(double) * E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->d_min_dist, mind.data(), sizeof(double) * E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->d_last_valid, lv.data(), E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.cnt_na, cna.data(), 4 * (size_t)E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.iou, iou.data(), 4 * (size_t)E, hipMemcpyHostToDevice));
    return T2D_OK;
}

// plain per-env CSR of the static obstacle rings for the lidar kernel (from the host copy of the geometry)
int rebuild_lidar_geo(t2d_pool* p) {
    if (!p->lidar_on) return T2D_OK;
    const int E = p->v.n_env;
    const auto& g = p->hgeo[0];
    int rc;
    p->lidar.max_static_verts = 0;
    p->lidar.env_vert_cnt = nullptr;
    p->lidar.edge_meta = nullptr;
    if (p->scene_mode) {  // generated scenes: every env owns 4 * T2D_GEN_MAX_QUADS edge slots of d_lidar_xy, the
        constexpr int VS = 4 * T2D_GEN_MAX_QUADS;  // scene kernel maintains the edges and the per-env count
        std::vector<int32_t> evo(E + 1);
        for (int e = 0; e <= E; ++e) evo[e] = VS * e;
        if ((rc = dev_replace(p, &p->d_lidar_env_off, evo.data(), evo.size()))) return rc;
        p->lidar.max_static_verts = VS;
        p->lidar.env_vert_cnt = p->d_lidar_cnt;
        p->lidar.edge_meta = p->d_lidar_meta;   // (ring slots < 16 and <= 48 edges per env by construction)
    } else if (!g.present || g.ring_env_off[E] == 0) {
        if ((rc = dev_replace<int32_t>(p, &p->d_lidar_env_off, nullptr, 0))) return rc;
        if ((rc = dev_replace<float>(p, &p->d_lidar_xy, nullptr, 0))) return rc;
    } else {   // the caller's rings, not the event kernels' fans
        const int P = g.ring_env_off[E], V = g.ring_vert_off[P];
        std::vector<int32_t> evo(E + 1);
        std::vector<float> edges(4 * (size_t)V);   // one record per edge: vertex v and the next vertex of its ring
        for (int e = 0; e <= E; ++e) evo[e] = g.ring_vert_off[g.ring_env_off[e]];
        for (int q = 0; q < P; ++q)
            for (int v = g.ring_vert_off[q]; v < g.ring_vert_off[q + 1]; ++v) {
                const int nx = v + 1 < g.ring_vert_off[q + 1] ? v + 1 : g.ring_vert_off[q];
                edges[4 * (size_t)v] = g.ring_xy[2 * (size_t)v]; edges[4 * (size_t)v + 1] = g.ring_xy[2 * (size_t)v + 1];
                edges[4 * (size_t)v + 2] = g.ring_xy[2 * (size_t)nx]; edges[4 * (size_t)v + 3] = g.ring_xy[2 * (size_t)nx + 1];
            }
        for (int e = 0; e < E; ++e) p->lidar.max_static_verts = std::max(p->lidar.max_static_verts, evo[e + 1] - evo[e]);
        // which rings may take part in the scan's occlusion culling (t2d_lidar.hip): well-shaped rings (the chord bound of
        // the culling argument needs sin(interior angle) >= 0.05 at every vertex) of envs with <= 16 rings and <= 48 edges
        std::vector<uint8_t> meta((size_t)V, 0xff);
        for (int e = 0; e < E; ++e) {
            const int q0 = g.ring_env_off[e], q1 = g.ring_env_off[e + 1];
            if (q1 - q0 > 16 || evo[e + 1] - evo[e] > 48) continue;
            for (int q = q0; q < q1; ++q) {
                const int a = g.ring_vert_off[q], n = g.ring_vert_off[q + 1] - a;
                bool ok = true;
                for (int i = 0; i < n && ok; ++i) {
                    const float* v = &g.ring_xy[2 * (size_t)(a + i)];
                    const float* pr = &g.ring_xy[2 * (size_t)(a + (i + n - 1) % n)];
                    const float* nx = &g.ring_xy[2 * (size_t)(a + (i + 1) % n)];
                    const double ux = (double)pr[0] - v[0], uy = (double)pr[1] - v[1], wx = (double)nx[0] - v[0], wy = (double)nx[1] - v[1];
                    const double cr = std::fabs(ux * wy - uy * wx), den = std::sqrt((ux * ux + uy * uy) * (wx * wx + wy * wy));
                    ok = den > 0.0 && cr >= 0.05 * den;
                }
                if (ok)
                    for (int i = 0; i < n; ++i) meta[(size_t)(a + i)] = (uint8_t)(q - q0);
            }
        }
        if ((rc = dev_replace(p, &p->d_lidar_env_off, evo.data(), evo.size()))) return rc;
        if ((rc = dev_replace(p, &p->d_lidar_xy, edges.data(), edges.size()))) return rc;
        if ((rc = dev_replace(p, &p->d_lidar_meta, meta.data(), meta.size()))) return rc;
        p->lidar.edge_meta = p->d_lidar_meta;
    }
    p->lidar.env_vert_off = p->d_lidar_env_off;
    p->lidar.xy = p->d_lidar_xy;
    p->lidar.max_slots = p->lidar.max_static_verts + (p->lidar.include_participants ? 4 * p->v.A : 0);
    if ((sizeof(double) * (p->lidar.max_slots <= 64 ? 8 : 4) + 8) * (size_t)p->lidar.max_slots + 16 * (size_t)p->lidar.n_beams + 2048 > 60 * 1024)
        return fail(p, T2D_ERR_GEOMETRY, "too many obstacle edges per env for the lidar's LDS edge list");
    return T2D_OK;
}

int record_event(t2d_pool* p, int kernel_id, hipStream_t s, bool begin) {
    if (!p->profiling) return T2D_OK;
    if (begin) {
        if (p->prof_count + 2 > 2 * t2d_pool::kMaxProfSteps) return T2D_OK;  // buffer full
    } else if (!(p->prof_count & 1)) {
        return T2D_OK;  // the matching begin was skipped
    }
    T2D_HIP(p, hipEventRecord(p->prof_events[p->prof_count], s));
    p->prof_kernel[p->prof_count] = kernel_id;
    p->prof_count++;
    return T2D_OK;
}

}  // namespace


namespace t2d {
namespace {
struct SnapPtrs { const float* f[6]; const uint32_t* ids; };
__global__ __launch_bounds__(256) void restore_kernel(PoolView pv, SnapPtrs sp, int mode) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= pv.N) return;
    const int env = i / pv.A;
    if (mode == 1) {
        const uchar4 st = reinterpret_cast<const uchar4*>(pv.status)[env];
        if (!(st.z | st.w)) return;
    }
    pv.x[i] = sp.f[0][i]; pv.y[i] = sp.f[1][i]; pv.heading[i] = sp.f[2][i];
    pv.speed[i] = sp.f[3][i]; pv.vx[i] = sp.f[4][i]; pv.vy[i] = sp.f[5][i];
    pv.ids[i] = sp.ids[i];
    pv.flags[i] = 0;
    if (pv.snap_omega[0]) {
        pv.omega_f[i] = pv.snap_omega[0][i];
        pv.omega_r[i] = pv.snap_omega[1][i];
    }
    // the env record (status, counters) is cleared by restore_env_kernel, launched after this
    // kernel on the same stream, so every participant has read `status` before it changes
}
// one wave that does nothing for `ticks` of the 100 MHz real-time counter (t2d_debug_delay_gather)
__global__ __launch_bounds__(64) void spin_kernel(long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(256) void restore_env_kernel(PoolView pv, int mode) {
    const int env = blockIdx.x * 256 + threadIdx.x;
    if (env >= pv.n_env) return;
    if (mode == 1) {
        const uchar4 st = reinterpret_cast<const uchar4*>(pv.status)[env];
        if (!(st.z | st.w)) return;
    }
    pv.env_flags[env] = 0; pv.cnt_step[env] = 0; pv.frame_ms[env] = 0; pv.reward[env] = 0.f;
    pv.last_valid[env] = 0; pv.cnt_na[env] = 0; pv.max_iou[env] = -INFINITY;
    pv.min_dist[env] = pv.snap_min_dist[env]; pv.iou[env] = NAN;
    uchar4 st; st.x = T2D_SCENARIO_NORMAL; st.y = T2D_TRAFFIC_NORMAL; st.z = 0; st.w = 0;
    reinterpret_cast<uchar4*>(pv.status)[env] = st;
}
// back to the checkpoint of a failed CHAIN fragment (PoolView::ckpt): the state arrays and the envs' counters
__global__ __launch_bounds__(256) void chain_rollback_kernel(PoolView pv, const uint32_t* ck) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)pv.N;
    if (i < pv.N) {
        // (every slot of the pool was loaded and checkpointed by the fragment's first step, active or not; a point mass's
        // velocity is the only velocity that is state)
        const uint32_t ids = ck[6 * n + i];
        pv.x[i] = __uint_as_float(ck[i]);
        pv.y[i] = __uint_as_float(ck[n + i]);
        pv.heading[i] = __uint_as_float(ck[2 * n + i]);
        pv.speed[i] = __uint_as_float(ck[3 * n + i]);
        if (((ids >> kIdsActiveShift) & 0xffu) && ((ids >> kIdsModelShift) & 0xffu) == T2D_MODEL_POINTMASS) {
            pv.vx[i] = __uint_as_float(ck[4 * n + i]);
            pv.vy[i] = __uint_as_float(ck[5 * n + i]);
        }
        pv.ids[i] = ids;
    }
    if (i < pv.n_env) {
        pv.cnt_step[i] = (int32_t)ck[7 * n + i];
        pv.frame_ms[i] = (int32_t)ck[7 * n + pv.n_env + i];
    }
}
}  // namespace
void pool_touch(t2d_pool* p, hipStream_t s) { touch(p, s); }
hipError_t launch_chain_rollback(const PoolView& v, const uint32_t* ckpt, hipStream_t s) {
    hipLaunchKernelGGL(chain_rollback_kernel, dim3((v.N + 255) / 256), dim3(256), 0, s, v, ckpt);
    return hipGetLastError();
}
hipError_t launch_spin(long long ticks, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks);
    return hipGetLastError();
}
hipError_t launch_restore(const PoolView& v, float* const* snap, const uint32_t* snap_ids, int mode,
                          hipStream_t s) {
    SnapPtrs sp;
    for (int k = 0; k < 6; ++k) sp.f[k] = snap[k];
    sp.ids = snap_ids;
    hipLaunchKernelGGL(restore_kernel, dim3((v.N + 255) / 256), dim3(256), 0, s, v, sp, mode);
    hipLaunchKernelGGL(restore_env_kernel, dim3((v.n_env + 255) / 256), dim3(256), 0, s, v, mode);
    return hipGetLastError();
}
}  // namespace t2d

extern "C" {

static void frame_release(t2d_pool* p);
static void refresh_idm_view(t2d_pool* p);

const char* t2d_last_error(const t2d_pool* pool) {
    return pool ? pool->err.c_str() : create_error().c_str();
}
int t2d_abi_version(void) { return T2D_ABI_VERSION; }

int t2d_create(int32_t n_env, int32_t max_agents, int32_t device_id, t2d_pool** out_pool) {
    if (!out_pool) return fail(nullptr, T2D_ERR_INVALID, "out_pool is null");
    *out_pool = nullptr;
    if (n_env <= 0 || max_agents <= 0 || max_agents > T2D_MAX_AGENTS)
        return fail(nullptr, T2D_ERR_INVALID, "n_env must be > 0 and 1 <= max_agents <= 256");
    if ((int64_t)n_env * max_agents > (int64_t)1 << 30)
        return fail(nullptr, T2D_ERR_INVALID, "pool too large");
    int n_dev = 0;
    T2D_HIP(nullptr, hipGetDeviceCount(&n_dev));
    if (device_id < 0 || device_id >= n_dev)
        return fail(nullptr, T2D_ERR_HIP, "HIP device " + std::to_string(device_id) + " not available (" +
                                              std::to_string(n_dev) + " devices visible)");
    T2D_HIP(nullptr, hipSetDevice(device_id));
    t2d_pool* p = new (std::nothrow) t2d_pool();
    if (!p) return fail(nullptr, T2D_ERR_NOMEM, "host allocation failed");
    p->device = device_id;
    p->v.n_env = n_env;
    p->v.A = max_agents;
    p->v.N = n_env * max_agents;
    for (int f = 0; f < T2D_F_COUNT; ++f) {
        const size_t n = field_per_env(f) ? (size_t)n_env : (size_t)p->v.N;
        p->field_bytes[f] = n * field_elem_bytes(f);
        if (f == T2D_F_LIDAR) {  // sized by t2d_lidar_config
            p->field_bytes[f] = 0;
            continue;
        }
        hipError_t e = hipMalloc(&p->field_ptr[f], p->field_bytes[f]);
        if (e == hipSuccess) e = hipMemset(p->field_ptr[f], 0, p->field_bytes[f]);
        if (e != hipSuccess) {
            std::string msg = std::string("hipMalloc/hipMemset: ") + hipGetErrorString(e);
            t2d_destroy(p);
            return fail(nullptr, e == hipErrorOutOfMemory ? T2D_ERR_NOMEM : T2D_ERR_HIP, msg);
        }
    }
    hipError_t e = hipMalloc((void**)&p->d_params, sizeof(double) * T2D_PARAM_COLS * T2D_MAX_TYPES);
    if (e != hipSuccess) {
        t2d_destroy(p);
        return fail(nullptr, T2D_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    {
        const size_t E = (size_t)n_env;
        hipError_t e2 = hipSuccess;
        auto alloc = [&](void** ptr, size_t bytes) {
            if (e2 == hipSuccess) e2 = hipMalloc(ptr, bytes);
            if (e2 == hipSuccess) e2 = hipMemset(*ptr, 0, bytes);
        };
        alloc((void**)&p->d_last_pose, E * 8 * sizeof(double));
        alloc((void**)&p->d_max_iou, E * sizeof(double));
        alloc((void**)&p->d_min_dist, E * sizeof(double));
        alloc((void**)&p->d_snap_min_dist, E * sizeof(double));
        alloc((void**)&p->d_last_valid, E);
        p->chain_slots = (n_env + 7 + 8) & ~7;   // one counter per step workgroup (at most one per env) + padding; then the error word
        alloc((void**)&p->d_chain, sizeof(unsigned long long) * ((size_t)p->chain_slots + 1));
        if (e2 != hipSuccess) {
            t2d_destroy(p);
            return fail(nullptr, T2D_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e2));
        }
    }
    t2d::PoolView& v = p->v;
    v.x = (float*)p->field_ptr[T2D_F_X];
    v.y = (float*)p->field_ptr[T2D_F_Y];
    v.heading = (float*)p->field_ptr[T2D_F_HEADING];
    v.speed = (float*)p->field_ptr[T2D_F_SPEED];
    v.vx = (float*)p->field_ptr[T2D_F_VX];
    v.vy = (float*)p->field_ptr[T2D_F_VY];
    v.act0 = (float*)p->field_ptr[T2D_F_ACT0];
    v.act1 = (float*)p->field_ptr[T2D_F_ACT1];
    v.act_stride = 1;
    v.applied0 = (float*)p->field_ptr[T2D_F_APPLIED0];
    v.applied1 = (float*)p->field_ptr[T2D_F_APPLIED1];
    v.out_mask = (int32_t)T2D_OUT_ALL;
    v.omega_f = (float*)p->field_ptr[T2D_F_OMEGA_F];
    v.omega_r = (float*)p->field_ptr[T2D_F_OMEGA_R];
    v.ids = (uint32_t*)p->field_ptr[T2D_F_IDS];
    v.flags = (uint32_t*)p->field_ptr[T2D_F_FLAGS];
    v.env_flags = (uint32_t*)p->field_ptr[T2D_F_ENV_FLAGS];
    v.cnt_step = (int32_t*)p->field_ptr[T2D_F_CNT_STEP];
    v.frame_ms = (int32_t*)p->field_ptr[T2D_F_FRAME_MS];
    v.status = (uint8_t*)p->field_ptr[T2D_F_STATUS];
    v.reward = (float*)p->field_ptr[T2D_F_REWARD];
    v.record = (uint2*)p->field_ptr[T2D_F_RECORD];
    v.iou = (float*)p->field_ptr[T2D_F_IOU];
    v.cnt_na = (int32_t*)p->field_ptr[T2D_F_CNT_NO_ACTION];
    v.target_xy = nullptr;
    v.target_c = nullptr;
    v.last_pose = p->d_last_pose;
    v.last_valid = p->d_last_valid;
    v.max_iou = p->d_max_iou;
    v.min_dist = p->d_min_dist;
    v.snap_min_dist = p->d_snap_min_dist;
    v.params = p->d_params;
    v.cell = 1.0;
    v.inv_cell = 1.0;
    v.dbg = nullptr;
    v.map_flags = nullptr;
    for (int k = 0; k < 6; ++k) v.snap[k] = nullptr;
    v.snap_ids = nullptr;
    v.auto_reset = 0;
    v.overlapped = 0;
    v.wgmap = nullptr;
#ifdef T2D_TIMING
    {   // (T2D_TIMING_WORDS: room for the chained form's record per wave AND step, scripts/chain_timing.py)
        size_t words = (size_t)(n_env + 64) * 16 * 4;
        if (const char* e = getenv("T2D_TIMING_WORDS")) words = std::max(words, (size_t)strtoull(e, nullptr, 10));
        (void)hipMalloc((void**)&v.dbg, words * sizeof(unsigned long long));
        (void)hipMemset(v.dbg, 0, words * sizeof(unsigned long long));
    }
#endif
    v.geo = nullptr;
    v.geo_layout = t2d::GeoLayout{};
    p->v.n_env = n_env;   // (envs_per_workgroup reads it)
    v.geo_layout.epb = envs_per_workgroup(p, log2_pad(max_agents));
    // ParkingEnv defaults: envs/parking.py:106 (max_step 2e4), :151-163 (reward table)
    p->status_cfg = t2d_status_config{20000, 0, 0, 0, -5.0f, -1.0f, -5.0f, 5.0f, 0.001f,
                                      0, 0, 100, 0, 0.95f, 0.999f, 0.1f};
#ifndef T2D_CHAIN_DEPTH_DEFAULT
#define T2D_CHAIN_DEPTH_DEFAULT 1
#endif
    p->chain_depth = T2D_CHAIN_DEPTH_DEFAULT;
#ifdef T2D_EXPERIMENTS   // (builds with -DT2D_EXPERIMENTS carry the instantiation; the product's depth is 1)
    if (const char* e = getenv("T2D_CHAIN_DEPTH")) p->chain_depth = std::max(1, std::min(16, atoi(e)));
#endif
    *out_pool = p;
    return T2D_OK;
}

int t2d_destroy(t2d_pool* p) {
    if (!p) return T2D_OK;
    (void)hipSetDevice(p->device);
    (void)quiesce(p);  // nothing of this pool may still be running (incl. a scene refill on its own stream)
    for (int f = 0; f < T2D_F_COUNT; ++f)
        if (p->field_ptr[f]) (void)hipFree(p->field_ptr[f]);
    void* bufs[] = {p->d_params, p->d_geo, p->d_boundary, p->d_boundary_valid, p->d_target_xy, p->d_target_c,
                    p->d_last_pose, p->d_max_iou, p->d_min_dist, p->d_snap_min_dist, p->d_last_valid,
                    p->d_lidar_env_off, p->d_lidar_next, p->d_lidar_meta, p->d_lidar_xy, p->d_beam_sin, p->d_beam_cos,
                    p->d_snap[0], p->d_snap[1], p->d_snap[2],
                    p->d_snap[3], p->d_snap[4], p->d_snap[5], p->d_snap_ids, p->d_wgmap, p->d_idm_rows, p->d_idm_ctrl, p->d_snap_omega[0], p->d_snap_omega[1], p->d_time_penalty,
                    p->d_scene_arrays, p->d_lidar_cnt, p->d_chain, p->d_ckpt, p->d_scene_view,
                    p->d_grid_env, p->d_grid_cell_start, p->d_grid_items, p->d_map_flags, p->d_grid_bnd, p->d_grid_seg};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    frame_release(p);
    if (p->d_target_heading) (void)hipFree(p->d_target_heading);
    if (p->comm && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)p->comm);
    if (p->gather_stream) (void)hipStreamDestroy(p->gather_stream);
    if (p->ev_frag_ready) (void)hipEventDestroy(p->ev_frag_ready);
    for (hipEvent_t e : p->ev_gather)
        if (e) (void)hipEventDestroy(e);
    if (p->scene_stream) (void)hipStreamDestroy(p->scene_stream);
    if (p->ev_scene_commit) (void)hipEventDestroy(p->ev_scene_commit);
    if (p->ev_scene_refill) (void)hipEventDestroy(p->ev_scene_refill);
    if (p->prof_events) {
        for (int i = 0; i < 2 * t2d_pool::kMaxProfSteps; ++i) (void)hipEventDestroy(p->prof_events[i]);
        delete[] p->prof_events;
    }
    delete p;
    return T2D_OK;
}

int t2d_set_param_table(t2d_pool* p, const double* rows, int32_t n_types, int32_t row_stride) {
    if (!p) return T2D_ERR_INVALID;
    if (!rows || n_types <= 0 || n_types > T2D_MAX_TYPES || row_stride < T2D_PARAM_COLS)
        return fail(p, T2D_ERR_INVALID, "need 1..32 types and row_stride >= 24");
    T2D_HIP(p, hipSetDevice(p->device));
    std::vector<double> t(T2D_PARAM_COLS * T2D_MAX_TYPES, 0.0);
    double dmax = 0.0;
    bool has_drift = false;
    for (int ty = 0; ty < n_types; ++ty) {
        const double* r = rows + (size_t)ty * row_stride;
        const int model = (int)r[T2D_P_MODEL];
        if (model < 0 || model > T2D_MODEL_POINTMASS_EULER)
            return fail(p, T2D_ERR_INVALID, "row " + std::to_string(ty) + ": unknown model id");
        if (model == T2D_MODEL_POINTMASS_EULER) has_drift = true;   // (integrated by the side kernel, like the drift model)
        if (model == T2D_MODEL_DRIFT) {
            has_drift = true;
            if (!(r[T2D_P_MASS] > 0.0) || !(r[T2D_P_IZ] > 0.0) || !(r[T2D_P_DRIFT_RADIUS] > 0.0) ||
                !(r[T2D_P_DRIFT_IYW] > 0.0) || !(r[T2D_P_LF] != 0.0))
                return fail(p, T2D_ERR_INVALID, "row " + std::to_string(ty) +
                                                    ": SingleTrackDrift needs mass, I_z, radius, I_yw > 0 and lf != 0");
        }
        const int dt = (int)r[T2D_P_DELTA_T_MS];
        if (dt < 1) return fail(p, T2D_ERR_INVALID, "row " + std::to_string(ty) + ": delta_t must be >= 1 ms");
        if (model != T2D_MODEL_POINTMASS && model != T2D_MODEL_POINTMASS_EULER && !(r[T2D_P_WB] != 0.0))
            return fail(p, T2D_ERR_INVALID, "row " + std::to_string(ty) + ": zero wheel base");
        for (int c = 0; c < T2D_PARAM_COLS; ++c) {
            p->host_params[ty][c] = r[c];
            t[(size_t)c * T2D_MAX_TYPES + ty] = r[c];
        }
        const double L = r[T2D_P_LENGTH], W = r[T2D_P_WIDTH];
        const double br = (int)r[T2D_P_SHAPE] == T2D_SHAPE_CIRCLE ? 0.5 * W : 0.5 * sqrt(L * L + W * W);
        t[(size_t)T2D_P_RESERVED0 * T2D_MAX_TYPES + ty] = br;  // bounding radius for the reject test
        if (model != T2D_MODEL_DRIFT) {   // the two derived columns (include/t2d.h): the sub-step in seconds now, its counts per launch interval
            t[(size_t)T2D_P_DT_S * T2D_MAX_TYPES + ty] = (double)dt / 1000;
            t[(size_t)T2D_P_SUBSTEPS * T2D_MAX_TYPES + ty] = 0.0;
        }
        dmax = std::max(dmax, 2.0 * br);
    }
    p->v.n_types = n_types;
    p->has_drift = has_drift;
    p->all_boxes = true;
    for (int ty = 0; ty < n_types; ++ty)
        if ((int)p->host_params[ty][T2D_P_SHAPE] != T2D_SHAPE_OBB) p->all_boxes = false;
    p->v.cell = dmax * 1.001 + 1e-3;  // 3x3 cell neighbourhood is then provably sufficient
    p->v.inv_cell = 1.0 / p->v.cell;
    T2D_HIP(p, quiesce(p));
    T2D_HIP(p, hipMemcpy(p->d_params, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
    // the derived column (T2D_P_SUBSTEPS) of the new table, for the interval the pool was last stepped with, at once: a
    // captured graph of step launches does not hold the derive launch and would otherwise replay on a column of zeros
    // (no sub-steps at all, silently) after a new table
    const int prev_interval = p->derived_interval;
    p->derived_interval = -1;   // (the next stepping call derives again: the host-side values travel with it)
    if (prev_interval > 0) {
        T2D_HIP(p, t2d::launch_derive(p->d_params, n_types, prev_interval, nullptr));
        T2D_HIP(p, hipStreamSynchronize(nullptr));
    }
    p->have_params = true;
    return T2D_OK;
}

int t2d_set_static_geometry(t2d_pool* p, const int32_t* env_poly_offsets,
                            const int32_t* poly_vert_offsets, const float* verts_xy,
                            const float* boundary, const uint8_t* boundary_valid) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    const int E = p->v.n_env;
    int rc;
    if (p->scene_mode) {  // host-described geometry replaces the generated scenes
        p->scene_mode = p->scene_regen = false;
        if ((rc = dev_replace<float>(p, &p->d_lidar_xy, nullptr, 0))) return rc;
    }
    if (env_poly_offsets) {
        if (!poly_vert_offsets || (!verts_xy && env_poly_offsets[E] > 0))
            return fail(p, T2D_ERR_INVALID, "polygon CSR arrays missing");
        if ((rc = prepare_polys(p, env_poly_offsets, poly_vert_offsets, verts_xy, p->hgeo[0])) != T2D_OK)
            return rc;
    } else {
        p->hgeo[0] = t2d_pool::HostGeo{};
    }
    if ((rc = rebuild_geo(p)) != T2D_OK) return rc;
    if ((rc = rebuild_lidar_geo(p)) != T2D_OK) return rc;
    if (boundary) {
        if ((rc = dev_replace(p, &p->d_boundary, boundary, (size_t)4 * E))) return rc;
        if ((rc = dev_replace(p, &p->d_boundary_valid, boundary_valid, boundary_valid ? (size_t)E : 0)))
            return rc;
    } else {
        if ((rc = dev_replace<float>(p, &p->d_boundary, nullptr, 0))) return rc;
        if ((rc = dev_replace<uint8_t>(p, &p->d_boundary_valid, nullptr, 0))) return rc;
    }
    p->v.boundary = p->d_boundary;
    p->v.boundary_valid = p->d_boundary_valid;
    return T2D_OK;
}

int t2d_set_lane_geometry(t2d_pool* p, const int32_t* env_lane_offsets,
                          const int32_t* lane_vert_offsets, const float* verts_xy) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    const int E = p->v.n_env;
    int rc;
    if (p->scene_mode) return fail(p, T2D_ERR_STATE, "lane geometry cannot be combined with generated parking scenes");
    if (env_lane_offsets) {
        if (!lane_vert_offsets || (!verts_xy && env_lane_offsets[E] > 0))
            return fail(p, T2D_ERR_INVALID, "lane CSR arrays missing");
        if ((rc = prepare_polys(p, env_lane_offsets, lane_vert_offsets, verts_xy, p->hgeo[1])) != T2D_OK)
            return rc;
        build_lane_boundary(E, p->hgeo[1]);
        build_safe_rects(E, p->hgeo[1]);
    } else {
        p->hgeo[1] = t2d_pool::HostGeo{};
    }
    return rebuild_geo(p);
}

int t2d_set_target_areas(t2d_pool* p, const float* target_xy, const float* centroid) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    const int E = p->v.n_env;
    int rc;
    if (p->scene_mode) return fail(p, T2D_ERR_STATE, "target areas belong to the generated parking scenes; "
                                                      "call t2d_set_static_geometry first to leave that mode");
    if (!target_xy) {
        if ((rc = dev_replace<double>(p, &p->d_target_xy, nullptr, 0))) return rc;
        if ((rc = dev_replace<double>(p, &p->d_target_c, nullptr, 0))) return rc;
        p->have_target = false;
    } else {
        std::vector<double> xy(8 * (size_t)E), c(2 * (size_t)E);
        for (int e = 0; e < E; ++e) {
            std::vector<double> q(8);
            for (int k = 0; k < 8; ++k) q[k] = (double)target_xy[8 * (size_t)e + k];
            double a = area2(q);
            if (a < 0.0) {  // clockwise -> reverse
                for (int i = 0, j = 3; i < j; ++i, --j) {
                    std::swap(q[2 * i], q[2 * j]);
                    std::swap(q[2 * i + 1], q[2 * j + 1]);
                }
                a = -a;
            }
            if (!(a > 0.0)) return fail(p, T2D_ERR_GEOMETRY, "target area " + std::to_string(e) + " is degenerate");
            for (int i = 0; i < 4; ++i)
                if (orient_h(&q[2 * i], &q[2 * ((i + 1) & 3)], &q[2 * ((i + 2) & 3)]) < 0.0)
                    return fail(p, T2D_ERR_GEOMETRY, "target area " + std::to_string(e) + " is not convex");
            memcpy(&xy[8 * (size_t)e], q.data(), 8 * sizeof(double));
            if (centroid) {
                c[2 * (size_t)e] = centroid[2 * (size_t)e];
                c[2 * (size_t)e + 1] = centroid[2 * (size_t)e + 1];
            } else {  // area centroid of the polygon (shapely Polygon.centroid)
                double cx = 0.0, cy = 0.0;
                for (int i = 0; i < 4; ++i) {
                    const int j = (i + 1) & 3;
                    const double w = q[2 * i] * q[2 * j + 1] - q[2 * j] * q[2 * i + 1];
                    cx += (q[2 * i] + q[2 * j]) * w;
                    cy += (q[2 * i + 1] + q[2 * j + 1]) * w;
                }
                c[2 * (size_t)e] = cx / (3.0 * a);
                c[2 * (size_t)e + 1] = cy / (3.0 * a);
            }
        }
        if ((rc = dev_replace(p, &p->d_target_xy, xy.data(), xy.size()))) return rc;
        if ((rc = dev_replace(p, &p->d_target_c, c.data(), c.size()))) return rc;
        p->have_target = true;
    }
    p->v.target_xy = p->d_target_xy;
    p->v.target_c = p->d_target_c;
    if (p->have_reset) {  // distance-to-target of the shaping restarts from the current state
        const size_t N = (size_t)p->v.N;
        std::vector<float> hx(N), hy(N);
        T2D_HIP(p, hipMemcpy(hx.data(), p->v.x, 4 * N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hy.data(), p->v.y, 4 * N, hipMemcpyDeviceToHost));
        if ((rc = init_iou_state(p, nullptr, hx.data(), hy.data()))) return rc;
    }
    return T2D_OK;
}

int t2d_set_status_config(t2d_pool* p, const t2d_status_config* cfg) {
    if (!p) return T2D_ERR_INVALID;
    if (!cfg) return fail(p, T2D_ERR_INVALID, "cfg is null");
    if (cfg->ego_index < 0 || cfg->ego_index >= p->v.A)
        return fail(p, T2D_ERR_INVALID, "ego_index out of range");
    p->status_cfg = *cfg;
    // time-penalty term of ParkingEnv._get_reward (envs/parking.py:156-158) for every possible step count: the
    // epilogue then reads one double instead of evaluating tanh on one lane at the very end of the wave
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    std::vector<double> tp;
    if (cfg->max_step > 0 && cfg->max_step <= (1 << 24)) {
        tp.resize((size_t)cfg->max_step + 1);
        for (int c = 0; c <= cfg->max_step; ++c)
            tp[c] = -tanh((double)c / (double)cfg->max_step) * (double)cfg->time_penalty_scale;
    }
    int rc = dev_replace(p, &p->d_time_penalty, tp.data(), tp.size());
    if (rc != T2D_OK) return rc;
    p->v.time_penalty = p->d_time_penalty;
    return T2D_OK;
}

int t2d_reset(t2d_pool* p, const uint8_t* env_mask, const float* x, const float* y,
              const float* heading, const float* speed, const float* vx, const float* vy,
              const uint8_t* type_id, const uint8_t* active) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_params) return fail(p, T2D_ERR_STATE, "t2d_set_param_table must precede t2d_reset");
    if (!x || !y || !heading || !speed || !type_id || !active)
        return fail(p, T2D_ERR_INVALID, "x, y, heading, speed, type_id, active are required");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    if (!env_mask) p->chain_failed = false;   // (every env gets a new state: whatever a failed t2d_step_n left behind is gone)
    const int E = p->v.n_env, A = p->v.A, N = p->v.N;
    std::vector<float> hx(N), hy(N), hh(N), hs(N), hvx(N), hvy(N);
    std::vector<uint32_t> hids(N), hflags(N);
    std::vector<uint32_t> henv(E);
    std::vector<int32_t> hcnt(E), hframe(E);
    std::vector<uint8_t> hstat(4 * (size_t)E);
    std::vector<float> hrew(E);
    const bool partial = env_mask != nullptr;
    if (partial) {  // read-modify-write of the unselected envs
        T2D_HIP(p, hipMemcpy(hx.data(), p->v.x, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hy.data(), p->v.y, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hh.data(), p->v.heading, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hs.data(), p->v.speed, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hvx.data(), p->v.vx, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hvy.data(), p->v.vy, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hids.data(), p->v.ids, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hflags.data(), p->v.flags, 4 * (size_t)N, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(henv.data(), p->v.env_flags, 4 * (size_t)E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hcnt.data(), p->v.cnt_step, 4 * (size_t)E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hframe.data(), p->v.frame_ms, 4 * (size_t)E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hstat.data(), p->v.status, 4 * (size_t)E, hipMemcpyDeviceToHost));
        T2D_HIP(p, hipMemcpy(hrew.data(), p->v.reward, 4 * (size_t)E, hipMemcpyDeviceToHost));
    }
    uint32_t types_used = partial ? p->types_used : 0u;   // (T2D_MAX_TYPES = 32 rows: one bit each)
    for (int e = 0; e < E; ++e) {
        if (partial && !env_mask[e]) continue;
        for (int a = 0; a < A; ++a) {
            const size_t i = (size_t)e * A + a;
            const int ty = type_id[i];
            if (active[i] && ty >= p->v.n_types)
                return fail(p, T2D_ERR_INVALID, "type_id " + std::to_string(ty) + " not in the parameter table");
            hx[i] = x[i]; hy[i] = y[i]; hh[i] = heading[i]; hs[i] = speed[i];
            if (vx && vy) {
                hvx[i] = vx[i]; hvy[i] = vy[i];
            } else {  // State.velocity: (speed*cos(heading), speed*sin(heading))  state.py:161-166
                hvx[i] = (float)((double)speed[i] * cos((double)heading[i]));
                hvy[i] = (float)((double)speed[i] * sin((double)heading[i]));
            }
            const int model = active[i] ? (int)p->host_params[ty][T2D_P_MODEL] : 0;
            if (active[i]) types_used |= 1u << ty;
            hids[i] = ((uint32_t)model << t2d::kIdsModelShift) | ((uint32_t)ty << t2d::kIdsTypeShift) |
                      ((uint32_t)(active[i] ? 1 : 0) << t2d::kIdsActiveShift);
            hflags[i] = 0;
        }
        henv[e] = 0; hcnt[e] = 0; hframe[e] = 0; hrew[e] = 0.0f;
        hstat[4 * (size_t)e] = T2D_SCENARIO_NORMAL; hstat[4 * (size_t)e + 1] = T2D_TRAFFIC_NORMAL;
        hstat[4 * (size_t)e + 2] = 0; hstat[4 * (size_t)e + 3] = 0;
    }
    T2D_HIP(p, hipMemcpy(p->v.x, hx.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.y, hy.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.heading, hh.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.speed, hs.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.vx, hvx.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.vy, hvy.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.ids, hids.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.flags, hflags.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.env_flags, henv.data(), 4 * (size_t)E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.cnt_step, hcnt.data(), 4 * (size_t)E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.frame_ms, hframe.data(), 4 * (size_t)E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.status, hstat.data(), 4 * (size_t)E, hipMemcpyHostToDevice));
    T2D_HIP(p, hipMemcpy(p->v.reward, hrew.data(), 4 * (size_t)E, hipMemcpyHostToDevice));
    {  // SingleTrackDrift wheel speeds start at rest (the reference's callers pass omega = 0 for a new episode)
        std::vector<float> ho(N, 0.0f);
        for (int f : {T2D_F_OMEGA_F, T2D_F_OMEGA_R}) {
            if (partial) {
                T2D_HIP(p, hipMemcpy(ho.data(), p->field_ptr[f], 4 * (size_t)N, hipMemcpyDeviceToHost));
                for (int e = 0; e < E; ++e)
                    if (env_mask[e]) std::fill(ho.begin() + (size_t)e * A, ho.begin() + (size_t)(e + 1) * A, 0.0f);
            }
            T2D_HIP(p, hipMemcpy(p->field_ptr[f], ho.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
        }
    }
    {
        int rc2 = init_iou_state(p, env_mask, hx.data(), hy.data());
        if (rc2 != T2D_OK) return rc2;
    }
    p->types_used = types_used;
    p->have_reset = true;
    return T2D_OK;
}

// IDM lanes read the pool's own action fields; only while caller-owned actions are bound do the kernels need to tell
// the two apart
static void refresh_idm_view(t2d_pool* p) {
    const bool bound = p->v.act0 != (const float*)p->field_ptr[T2D_F_ACT0];
    const bool sel = p->idm_on && bound;
    p->v.idm_ctrl = sel ? p->d_idm_ctrl : nullptr;
    p->v.own_act0 = sel ? (const float*)p->field_ptr[T2D_F_ACT0] : nullptr;
    p->v.own_act1 = sel ? (const float*)p->field_ptr[T2D_F_ACT1] : nullptr;
}

int t2d_bind_actions_strided(t2d_pool* p, const float* act0_dev, const float* act1_dev, int32_t stride) {
    if (!p) return T2D_ERR_INVALID;
    if ((act0_dev == nullptr) != (act1_dev == nullptr))
        return fail(p, T2D_ERR_INVALID, "bind both action arrays or neither");
    if (act0_dev && stride < 1) return fail(p, T2D_ERR_INVALID, "action stride must be >= 1 element");
    p->v.act0 = act0_dev ? act0_dev : (const float*)p->field_ptr[T2D_F_ACT0];
    p->v.act1 = act1_dev ? act1_dev : (const float*)p->field_ptr[T2D_F_ACT1];
    p->v.act_stride = act0_dev ? stride : 1;
    p->act_in_frame = false;
    p->act_extent = 0;   // how far the new memory reaches is not known until the caller says (t2d_set_action_extent)
    refresh_idm_view(p);
    return T2D_OK;
}

int t2d_set_action_extent(t2d_pool* p, int64_t n_elements) {
    if (!p) return T2D_ERR_INVALID;
    if (n_elements < 0) return fail(p, T2D_ERR_INVALID, "the action extent is a number of elements (0 = not declared)");
    if (n_elements && p->v.act0 == (const float*)p->field_ptr[T2D_F_ACT0])
        return fail(p, T2D_ERR_STATE, "t2d_set_action_extent describes memory bound with t2d_bind_actions[_strided]");
    if (n_elements && (int64_t)(p->v.N - 1) * p->v.act_stride >= n_elements)
        return fail(p, T2D_ERR_INVALID, "the declared extent does not hold one action per participant at the bound stride");
    p->act_extent = n_elements;
    return T2D_OK;
}

int t2d_bind_actions(t2d_pool* p, const float* act0_dev, const float* act1_dev) {
    return t2d_bind_actions_strided(p, act0_dev, act1_dev, 1);
}

static int drift_impl(t2d_pool* p, int interval_ms, hipStream_t s) {
    int rc;
    touch(p, s);
    if ((rc = record_event(p, 5, s, true))) return rc;
    T2D_HIP(p, t2d::launch_drift(p->v, interval_ms, s));
    return record_event(p, 5, s, false);
}

static int idm_impl(t2d_pool* p, hipStream_t s, const int32_t* forced_leader = nullptr) {
    int rc;
    touch(p, s);
    if ((rc = record_event(p, 4, s, true))) return rc;
    T2D_HIP(p, t2d::launch_idm(p->v, p->idm, forced_leader, (float*)p->field_ptr[T2D_F_ACT0],
                               (float*)p->field_ptr[T2D_F_ACT1], s));
    return record_event(p, 4, s, false);
}

// what a stepping launch needs of its interval beside the integer: interval_ms / 1000 (PointMass's dt) in the view, and the
// sub-step counts per type in the device table (one 32-thread launch on the step's stream whenever the interval changes)
// Coefficients of the resummed kinematic step (PoolView::kin_coef) for n sub-steps.  With u = k - m the centred sub-step
// index, w = u / M, the step's Euler sum  sum_k v_k (cos, sin)(theta_k),  theta_k = Theta + a w + b w^2,  v_k = V + ah M w,
// needs the means over k of cos / sin / w cos / w sin of (a w + b w^2); expanding in a (to a^(2 D + 1)) and b (to b^3) and
// using that the odd moments of w vanish leaves eight polynomials in a^2 whose coefficients are moments mu_p = mean(w^p)
// over factorials (SingleTrackKinematics._step, physics/single_track_kinematics.py:126-176, is the sum being restated).
static void kinematics_resum_table(t2d::PoolView& v, int n) {
    v.kin_n = 0;
    if (n < 1) return;
    constexpr int D = t2d::kKinDegree;
    const long double m = ((long double)n - 1) / 2, M = (long double)n / 2;
    long double mu[2 * D + 9];
    for (int q = 0; q <= 2 * D + 8; ++q) mu[q] = 0;
    for (int k = 0; k < n; ++k) {
        const long double w = ((long double)k - m) / M;
        long double pw = 1;
        for (int q = 0; q <= 2 * D + 8; ++q) {
            mu[q] += pw;
            pw *= w;
        }
    }
    for (int q = 0; q <= 2 * D + 8; ++q) mu[q] /= (long double)n;
    long double fact[2 * D + 3];
    fact[0] = 1;
    for (int q = 1; q <= 2 * D + 2; ++q) fact[q] = fact[q - 1] * q;
    for (int i = 0; i <= D; ++i) {
        const long double sg = (i & 1) ? -1.0L : 1.0L;
        const long double qe = sg / fact[2 * i], ro = sg / fact[2 * i + 1];
        v.kin_coef[i][t2d::KIN_Q0] = (double)(qe * mu[2 * i]);                  // E cos:   Q0 + b^2 Q4
        v.kin_coef[i][t2d::KIN_Q4] = (double)(-0.5L * qe * mu[2 * i + 4]);
        v.kin_coef[i][t2d::KIN_Q2] = (double)(qe * mu[2 * i + 2]);              // E sin:   b (Q2 + b^2 Q6)
        v.kin_coef[i][t2d::KIN_Q6] = (double)(-qe * mu[2 * i + 6] / 6);
        v.kin_coef[i][t2d::KIN_R4] = (double)(-ro * mu[2 * i + 4]);             // E w cos: a b (R4 + b^2 R8)
        v.kin_coef[i][t2d::KIN_R8] = (double)(ro * mu[2 * i + 8] / 6);
        v.kin_coef[i][t2d::KIN_R2] = (double)(ro * mu[2 * i + 2]);              // E w sin: a (R2 + b^2 R6)
        v.kin_coef[i][t2d::KIN_R6] = (double)(-0.5L * ro * mu[2 * i + 6]);
    }
    const double g[8] = {(double)m, (double)M, (double)(m - 0.5L), (double)(m * (m - 1) / 2), (double)(M * M / 2),
                         (double)(m + 1), (double)((m + 1) * (m + 1) / 2), (double)n};
    memcpy(v.kin_geo, g, sizeof g);
    v.kin_n = n;
}

static int prepare_interval(t2d_pool* p, int interval_ms, hipStream_t s) {
    if (interval_ms > T2D_MAX_INTERVAL_MS) return fail(p, T2D_ERR_INVALID, "interval_ms must be <= 32767");
    p->v.interval_s = (double)interval_ms / 1000;
    if (p->derived_interval != interval_ms) {
        // the resummed kinematic step is built for ONE sub-step count: that of the first SingleTrackKinematics row (every
        // type shares delta_t = 5 ms unless a caller says otherwise); lanes of a type with another count take the loop
        int kn = 0;
        for (int t = 0; t < p->v.n_types && !kn; ++t)
            if ((int)p->host_params[t][T2D_P_MODEL] == T2D_MODEL_KINEMATICS && p->host_params[t][T2D_P_DELTA_T_MS] >= 1.0)
                kn = interval_ms / (int)p->host_params[t][T2D_P_DELTA_T_MS];
        // ... and only pools that put at least two waves on every SIMD: a lone wave is a latency chain, and the table's scalar
        // loads (cache misses on first touch) lengthen it by more than the twenty loop trips they replace (same-box A/B,
        // 512 x 32 envs one launch per step: 12.8 -> 13.4 us with the series)
        if (p->device_cus == 0) (void)hipDeviceGetAttribute(&p->device_cus, hipDeviceAttributeMultiprocessorCount, p->device);
        const bool fills = (long long)p->v.N >= 2LL * 64 * 4 * (p->device_cus > 0 ? p->device_cus : 256);
        kinematics_resum_table(p->v, p->kin_resum && (fills || p->kin_resum_forced) ? kn : 0);
        touch(p, s);
        T2D_HIP(p, t2d::launch_derive(p->d_params, p->v.n_types, interval_ms, s));
        p->derived_interval = interval_ms;
    }
    return T2D_OK;
}

int t2d_integrate(t2d_pool* p, int32_t interval_ms, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_params || !p->have_reset)
        return fail(p, T2D_ERR_STATE, "t2d_set_param_table and t2d_reset must precede t2d_integrate");
    if (interval_ms <= 0) return fail(p, T2D_ERR_INVALID, "interval_ms must be positive");
    hipStream_t s = (hipStream_t)hip_stream;
    int rc;
    touch(p, s);
    if ((rc = prepare_interval(p, interval_ms, s))) return rc;
    if (p->idm_on && (rc = idm_impl(p, s))) return rc;
    if (p->has_drift && (rc = drift_impl(p, interval_ms, s))) return rc;
    if ((rc = record_event(p, 0, s, true))) return rc;
    // (four participants per lane pay where the model is cheap enough for the kernel to be memory-bound: measured at 4 M
    // participants 72.4 -> 63.5 us for point masses, 74.6 -> 73.7 for the kinematic bicycle, 167.7 -> 170.3 for the dynamics
    // model -- pools whose table holds a dynamics row keep one participant per lane)
    bool wide = true;
    for (int t = 0; t < p->v.n_types; ++t)   // (of the types some active participant HAS: t2d_reset keeps the set)
        wide = wide && !((p->types_used >> t & 1u) && (int)p->host_params[t][T2D_P_MODEL] == T2D_MODEL_DYNAMICS);
    int only_model = -1;   // the one model every active participant has, if there is one: an instantiation that carries it alone
    for (int t = 0; t < p->v.n_types; ++t)
        if (p->types_used >> t & 1u) {
            const int m = (int)p->host_params[t][T2D_P_MODEL];
            only_model = only_model == -1 || only_model == m ? m : -2;
        }
    T2D_HIP(p, t2d::launch_integrate(p->v, interval_ms, p->integrator_variant, wide, only_model, s));
    return record_event(p, 0, s, false);
}

// a pool with installed IDM controllers whose step launch runs them itself (envs of 2..64 participants: an env's positions
// then sit in one wave's LDS slots; no IoU events: those pools keep the general instantiations): every workgroup ahead of
// its integrator (t2d_step, the chained form of t2d_step_n), or the integrator waves of the PIPE form where the pool is
// small enough for it.  t2d_set_step_chaining(pool, 0, *) keeps idm_kernel a launch of its own (tests hold the two
// against each other).
static bool idm_in_step(t2d_pool* p) {
    // (not pools with SingleTrackDrift participants: drift_kernel runs between the controllers and the step launch and reads
    // the accelerations the controllers wrote -- t2d_step's order is IDM, drift, step -- so there idm_kernel stays a launch)
    return p->idm_on && p->chain_steps && p->fused_step && !p->grid_tier && !p->has_drift && p->v.A >= 2 && p->v.A <= 64 &&
           !(p->status_cfg.check_no_action || p->status_cfg.check_arrival);
}
static void fill_idm(t2d::PoolView& v, t2d_pool* p) {
    v.idm_rows = p->idm.rows;
    v.idm_ctrl_all = p->idm.ctrl_id;
    v.idm_leader = p->idm.leader;
    v.idm_n_ctrl = p->idm.n_ctrl;
    v.idm_act0_own = (float*)p->field_ptr[T2D_F_ACT0];
    v.idm_act1_own = (float*)p->field_ptr[T2D_F_ACT1];
}
static bool idm_in_pipe(t2d_pool* p) {
    if (!idm_in_step(p) || !p->chain_loop || !p->chain_pipe) return false;
    if (p->device_cus == 0) (void)hipDeviceGetAttribute(&p->device_cus, hipDeviceAttributeMultiprocessorCount, p->device);
    const int wgs = (p->v.n_env + p->v.geo_layout.epb - 1) / p->v.geo_layout.epb;
    return p->device_cus > 0 && wgs <= p->device_cus;
}

// small pools of 33..64-agent envs: the fused step gives every env a workgroup of its own (collide_kernel<..., SPLIT>)
static bool use_split(t2d_pool* p) {
    if (!p->split_steps) return false;
    if (p->device_cus == 0) (void)hipDeviceGetAttribute(&p->device_cus, hipDeviceAttributeMultiprocessorCount, p->device);
    int log2A = 1;
    while ((1 << log2A) < p->v.A) ++log2A;
    return t2d::split_eligible(p->v, p->status_cfg, log2A, p->device_cus);
}

static int collide_impl(t2d_pool* p, bool with_status, int interval_ms, hipStream_t s, int fuse_variant = -1) {
    int rc;
    touch(p, s);
    const int kid = fuse_variant >= 0 ? 2 : 1;
    if ((rc = record_event(p, kid, s, true))) return rc;
    // single-ego pools (ParkingEnv: one box-shaped participant per env, no lanes) step with one WAVE per env
    // (t2d_ego.hip) instead of one lane per participant; same arithmetic, same results (t2d_set_ego_kernel(pool, 0)
    // keeps such a pool on the general kernel: the two are held against each other in tests/test_gpu_ego.py)
    p->scene_committed_in_step = false;
    if (p->grid_tier) {   // the map's verdicts for the poses as they are now, ahead of the event kernel that ORs them in
        if (fuse_variant >= 0) return fail(p, T2D_ERR_STATE, "internal: a grid-tier pool took the fused step");
        T2D_HIP(p, t2d::launch_map_events(p->v, p->mapgrid, p->d_grid_seg, p->d_map_flags, s));
    }
    if (fuse_variant >= 0 && p->ego_kernel && p->v.A == 1 && p->all_boxes && !p->has_drift && !p->hgeo[1].present) {
        t2d::PoolView v = p->v;
        // staged scene regeneration: the step's own epilogue moves finished envs into their next lot (no launch behind it)
        if (with_status && p->scene_regen && p->scene.ring > 0 && p->d_scene_view) {
            v.regen = p->d_scene_view;
            p->scene_committed_in_step = p->scene_commit_used = true;
        }
        T2D_HIP(p, t2d::launch_ego_step(v, p->status_cfg, interval_ms, fuse_variant, s));
    } else {
        t2d::PoolView v = p->v;
        const bool idm_fused = fuse_variant >= 0 && idm_in_step(p);
        // (one workgroup per env fills the GPU with a quarter of the envs: for a pool that has the GPU to itself, not for env
        // groups whose launches are meant to overlap -- round 4: a group of 1024 envs took every workgroup slot and the groups
        // ran one after the other, 44 us per step of 4 groups)
        v.split_step = fuse_variant >= 0 && !idm_fused && !p->v.overlapped && use_split(p);
        if (idm_fused) fill_idm(v, p);
        T2D_HIP(p, t2d::launch_collide(v, p->status_cfg, with_status, interval_ms, fuse_variant, s));
    }
    return record_event(p, kid, s, false);
}

int t2d_collide(t2d_pool* p, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_params || !p->have_reset)
        return fail(p, T2D_ERR_STATE, "t2d_set_param_table and t2d_reset must precede t2d_collide");
    return collide_impl(p, false, 0, (hipStream_t)hip_stream);
}

// generated parking scenes with regeneration on: envs whose episode ended in the launch just enqueued move on to their next
// scene -- in that launch's own epilogue (ego step kernel) or in one cheap launch on the step's stream: staged scenes are
// copied in; every kSceneRefillPeriod steps the staging ring is topped up on the pool's own stream, off the critical path
// (one scene is a ~46 us single-lane chain).  The step stream waits for refill j before it launches refill j + 1, so the
// commits of steps [8 j, 8 j + 8) rely on refill j - 1 alone: it staged the 16 episodes past what it read at step >= 8 j - 8,
// and an env ends at most one episode in two steps -- 8 episodes by step 8 j + 8: half the ring is margin.  A slot a refill
// rewrites held an episode the env has left.  (Period 4 until round 4: the event wait + the side stream's three launches
// cost the step stream ~1 us per step on average, 36.2 -> 35.0 us per ParkingEnv vector step with period 8.)
constexpr int kSceneRing = 16;
constexpr int kSceneRefillPeriod = 8;
static int regenerate_done_scenes(t2d_pool* p, hipStream_t s) {
    if (!p->scene_regen) return T2D_OK;
    int rc;
    const bool refill = p->scene.ring > 0 && p->step_count % kSceneRefillPeriod == 0;
    if (refill && p->scene_refill_pending) T2D_HIP(p, hipStreamWaitEvent(s, p->ev_scene_refill, 0));
    if (!p->scene_committed_in_step) {   // (the ego step kernel has done it in its epilogue otherwise)
        if ((rc = record_event(p, 6, s, true))) return rc;
        if (p->scene.ring > 0) {   // staged scenes: sixteen lanes per env copy the prepared scene in
            T2D_HIP(p, t2d::launch_scene_commit(p->v, p->scene, p->v.n_env, s));
            p->scene_commit_used = true;
        } else {
            T2D_HIP(p, t2d::launch_parking_scenes(p->v, p->scene, p->v.n_env, 2, s));
        }
        if ((rc = record_event(p, 6, s, false))) return rc;
    }
    if (refill) {
        T2D_HIP(p, hipEventRecord(p->ev_scene_commit, s));
        T2D_HIP(p, hipStreamWaitEvent(p->scene_stream, p->ev_scene_commit, 0));
        T2D_HIP(p, t2d::launch_scene_refill(p->scene, p->v.n_env, p->scene_stream));
        T2D_HIP(p, hipEventRecord(p->ev_scene_refill, p->scene_stream));
        p->scene_refill_pending = true;
    }
    return T2D_OK;
}

// this step's slot of the record ring; a gather still reading that slot (enqueued RING steps ago or less) is waited for
// on the step's stream first -- an event wait, nothing blocks the host
static int claim_record_slot(t2d_pool* p, hipStream_t s) {
    const int k = (int)(p->step_count % T2D_RECORD_RING);
    if (hipEvent_t e = p->slot_event[k]) {   // one wait covers every slot that gather reads (one event per gather)
        T2D_HIP(p, hipStreamWaitEvent(s, e, 0));
        for (hipEvent_t& q : p->slot_event)
            if (q == e) q = nullptr;
    }
    p->v.record = (uint2*)p->field_ptr[T2D_F_RECORD] + (size_t)k * p->v.n_env;
    return T2D_OK;
}

int t2d_check_status(t2d_pool* p, int32_t interval_ms, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_params || !p->have_reset)
        return fail(p, T2D_ERR_STATE, "t2d_set_param_table and t2d_reset must precede t2d_check_status");
    if (interval_ms <= 0) return fail(p, T2D_ERR_INVALID, "interval_ms must be positive");
    int rc;
    if ((rc = claim_record_slot(p, (hipStream_t)hip_stream))) return rc;
    rc = collide_impl(p, true, interval_ms, (hipStream_t)hip_stream);
    if (rc == T2D_OK) p->step_count++;
    if (rc == T2D_OK) rc = regenerate_done_scenes(p, (hipStream_t)hip_stream);
    return rc;
}

int t2d_step(t2d_pool* p, int32_t interval_ms, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->fused_step || p->grid_tier) {
        int rc = t2d_integrate(p, interval_ms, hip_stream);
        if (rc != T2D_OK) return rc;
        return t2d_check_status(p, interval_ms, hip_stream);
    }
    if (!p->have_params || !p->have_reset)
        return fail(p, T2D_ERR_STATE, "t2d_set_param_table and t2d_reset must precede t2d_step");
    if (interval_ms <= 0) return fail(p, T2D_ERR_INVALID, "interval_ms must be positive");
    int rc;
    if ((rc = prepare_interval(p, interval_ms, (hipStream_t)hip_stream))) return rc;
    if ((rc = claim_record_slot(p, (hipStream_t)hip_stream))) return rc;
    // (installed IDM controllers: a launch of their own ahead of the step -- or, idm_in_step, the front of the step launch)
    if (p->idm_on && !idm_in_step(p) && (rc = idm_impl(p, (hipStream_t)hip_stream))) return rc;
    if (p->has_drift && (rc = drift_impl(p, interval_ms, (hipStream_t)hip_stream))) return rc;
    rc = collide_impl(p, true, interval_ms, (hipStream_t)hip_stream, p->integrator_variant);
    if (rc == T2D_OK) p->step_count++;
    if (rc == T2D_OK) rc = regenerate_done_scenes(p, (hipStream_t)hip_stream);
    return rc;
}

int t2d_step_n(t2d_pool* p, int32_t interval_ms, int32_t n_steps, int64_t act_step_stride, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (n_steps < 1 || act_step_stride < 0) return fail(p, T2D_ERR_INVALID, "n_steps must be >= 1 and act_step_stride >= 0");
    // (the pool's own action fields hold ONE action set: a ring needs bound memory of n_steps * act_step_stride elements)
    if (act_step_stride != 0 && p->v.act0 == (const float*)p->field_ptr[T2D_F_ACT0])
        return fail(p, T2D_ERR_INVALID, "act_step_stride > 0 needs an action ring bound with t2d_bind_actions (the pool's own ACT0 / ACT1 hold one set)");
    // a declared extent (t2d_set_action_extent) is held against the furthest element step n_steps - 1 reads
    if (p->act_extent && (int64_t)(p->v.N - 1) * p->v.act_stride + (int64_t)(n_steps - 1) * act_step_stride >= p->act_extent)
        return fail(p, T2D_ERR_INVALID, "step " + std::to_string(n_steps - 1) + " of the fragment would read past the declared extent of the bound actions (" +
                                         std::to_string((long long)p->act_extent) + " elements)");
    if (p->chain_failed || p->scene_commit_failed) return report_chain_failure(p);   // (once; the call after it goes ahead with plain launches)
    if (!p->have_params || !p->have_reset)
        return fail(p, T2D_ERR_STATE, "t2d_set_param_table and t2d_reset must precede t2d_step_n");
    if (interval_ms <= 0) return fail(p, T2D_ERR_INVALID, "interval_ms must be positive");
    hipStream_t s = (hipStream_t)hip_stream;
    const bool ego = p->ego_kernel && p->v.A == 1 && p->all_boxes && !p->has_drift && !p->hgeo[1].present;
    const bool iou = p->status_cfg.check_no_action || p->status_cfg.check_arrival;   // per-env history read by the epilogue
    // (the single-ego kernel has a LOOP form of its own, which reads the history with sc1 loads: IoU events are fine there)
    const bool chain = p->chain_steps && n_steps >= 2 && p->fused_step && !p->grid_tier && (!p->idm_on || idm_in_step(p)) && !p->has_drift &&
                       !p->scene_regen && (ego ? p->chain_loop : !iou);
    const float *a0 = p->v.act0, *a1 = p->v.act1;
    int rc = T2D_OK;
    if (!chain) {   // kernels outside the fused step (IDM, drift, scene regeneration, the single-ego kernel): step by step
        for (int k = 0; k < n_steps && rc == T2D_OK; ++k) {
            p->v.act0 = a0 + (size_t)k * act_step_stride;
            p->v.act1 = a1 + (size_t)k * act_step_stride;
            rc = t2d_step(p, interval_ms, hip_stream);
        }
        p->v.act0 = a0;
        p->v.act1 = a1;
        return rc;
    }
    touch(p, s);
    if ((rc = prepare_interval(p, interval_ms, s))) return rc;
    for (int done = 0; done < n_steps && rc == T2D_OK;) {
        // one launch covers at most a ring of record slots (and a gather that still reads any of them is waited for first)
        int n = std::min(n_steps - done, (int)T2D_RECORD_RING);
        // the chained form of large pools: chain_depth steps per workgroup (launches of a multiple of it; what is left over
        // at the end of a call goes out one step per workgroup)
        const int depth = p->chain_loop && !ego && !p->idm_on && p->chain_depth > 1 && n >= p->chain_depth && log2_pad(p->v.A) <= 6
                              ? p->chain_depth : 1;
        n -= n % depth;
        const int slot0 = (int)(p->step_count % T2D_RECORD_RING);
        for (int k = 0; k < n; ++k) {
            if ((rc = claim_record_slot(p, s))) return rc;
            p->step_count++;
        }
        t2d::PoolView v = p->v;
        v.wgmap = nullptr;
        v.overlapped = p->chain_priority;
        v.act0 = a0 + (size_t)done * act_step_stride;
        v.act1 = a1 + (size_t)done * act_step_stride;
        v.chain_done = p->d_chain;
        v.chain_err = reinterpret_cast<uint32_t*>(p->d_chain + p->chain_slots);
        v.chain_base = p->chain_count;
        v.split_step = use_split(p);
        v.chain_real_wgs = v.split_step ? p->v.n_env : (p->v.n_env + p->v.geo_layout.epb - 1) / p->v.geo_layout.epb;
        v.chain_act_step = act_step_stride;
        // pools whose step launch is at most two workgroups per CU (all resident at once): the workgroups loop over the
        // steps themselves; larger pools chain one workgroup per (env set, step)
        {
            if (p->device_cus == 0) (void)hipDeviceGetAttribute(&p->device_cus, hipDeviceAttributeMultiprocessorCount, p->device);
            const int loop_wgs = (p->v.n_env + p->v.geo_layout.epb - 1) / p->v.geo_layout.epb;
            const bool loop_ok = p->chain_loop && p->device_cus > 0 && loop_wgs <= T2D_LOOP_MAX_WGS_PER_CU * p->device_cus;
            // ... and at most one workgroup per CU (envs of up to 64 participants): integrator waves a step ahead (PIPE)
            v.pipe_step = loop_ok && p->chain_pipe && loop_wgs <= p->device_cus && p->v.A <= 64;
            // (pools with lane polygons: the lane stage on a wave of its own -- needs three sets of waves in a workgroup)
            // -- where the lane stage is long enough to pay for the poses derived a second time: five lane polygons per env
            // and more (measured: 4 per env, the highway pool, 8.8 us per step without lane waves and 10.0 with them; 6 per
            // env 8.1 / 7.6; 4..16 per env, the mixed pool, 11.0 / 9.2)
            const size_t lane_polys = p->hgeo[1].present && !p->hgeo[1].env_off.empty() ? (size_t)p->hgeo[1].env_off.back() : 0;
            if (v.pipe_step && p->chain_pipe2 && p->v.geo && p->v.geo_layout.has[1] && lane_polys >= 5 * (size_t)p->v.n_env &&
                3 * (p->v.geo_layout.epb << log2_pad(p->v.A)) <= 1024)
                v.pipe_step = 2;
            if (v.pipe_step) {   // (preferred to the one-workgroup-per-env chain: the whole step overlaps, not only its stages)
                v.split_step = 0;
                v.chain_real_wgs = loop_wgs;
            }
            if (p->idm_on) {   // the integrator waves run the controllers (no lane waves then) -- or every workgroup of the chained form
                v.pipe_step = idm_in_pipe(p) ? 1 : 0;
                if (!v.pipe_step) {
                    v.split_step = 0;
                    v.chain_real_wgs = loop_wgs;
                }
                fill_idm(v, p);
            }
            v.loop_steps = (!v.split_step && loop_ok && !(p->idm_on && !v.pipe_step)) ? n : 0;
        }
        v.record_ring = (uint2*)p->field_ptr[T2D_F_RECORD];
        v.record_slot0 = slot0;
        // CHAIN forms (one workgroup per (env set, step), ordered by the counters in d_chain): the counters continue from
        // launch to launch only while the launches have one shape -- a pool that changes form (t2d_set_idm, new geometry,
        // t2d_set_split_step, t2d_set_step_chaining) starts them afresh, on the stream -- and every fragment carries the
        // checkpoint a failed hand-off is rolled back to
        const bool chain_form = !ego && v.loop_steps == 0;
        v.chain_k = 0;
        if (chain_form && !v.split_step && depth > 1) {
            v.chain_k = depth;
            v.loop_steps = depth;
        }
        if (chain_form) {
            const uint32_t sig = ((uint32_t)v.chain_real_wgs << 1) | (v.split_step ? 1u : 0u) | 0x80000000u;
            if (sig != p->chain_sig) {
                T2D_HIP(p, hipMemsetAsync(p->d_chain, 0, sizeof(unsigned long long) * (size_t)p->chain_slots, s));
                p->chain_count = 0;
                p->chain_sig = sig;
                v.chain_base = 0;
            }
            if (!p->d_ckpt) {
                const size_t words = 7 * (size_t)p->v.N + 2 * (size_t)p->v.n_env;
                T2D_HIP(p, hipMalloc((void**)&p->d_ckpt, sizeof(uint32_t) * words));
                T2D_HIP(p, hipMemset(p->d_ckpt, 0, sizeof(uint32_t) * words));
            }
            v.ckpt = p->d_ckpt;
            v.ckpt_tag = (uint32_t)(p->step_count - n);   // (low half of the step count this fragment starts from)
            v.chain_fault = p->chain_fault;
        } else {
            v.ckpt = nullptr;
            v.chain_fault = 0;
        }
        // (the checkpoint is what a failure is rolled back to only while every multi-step launch since the last host
        // synchronisation carried one)
        p->ckpt_armed = chain_form && (p->chain_used ? p->ckpt_armed : true);
        if ((rc = record_event(p, 7, s, true))) return rc;
        if (ego) {   // 16 lanes per env, each group of lanes loops over the steps
            v.loop_steps = n;
            // (at most one workgroup of 16 envs per CU: integrator waves a step ahead of the event waves, t2d_ego.hip)
            v.pipe_step = p->chain_pipe && p->device_cus > 0 && (p->v.n_env + 15) / 16 <= p->device_cus;
            T2D_HIP(p, t2d::launch_ego_step(v, p->status_cfg, interval_ms, p->integrator_variant, s));
        } else {
            T2D_HIP(p, t2d::launch_step_chain(v, p->status_cfg, interval_ms, p->integrator_variant, n, s));
        }
        if ((rc = record_event(p, 7, s, false))) return rc;
        if (chain_form) p->chain_count += (uint32_t)n;   // (only these launches move the counters)
        p->chain_used = true;
        done += n;
    }
    p->v.record = (uint2*)p->field_ptr[T2D_F_RECORD] + (size_t)((p->step_count + T2D_RECORD_RING - 1) % T2D_RECORD_RING) * p->v.n_env;
    return rc;
}

int t2d_step_form(t2d_pool* p, int32_t n_steps) {
    if (!p) return -1;
    const bool ego = p->ego_kernel && p->v.A == 1 && p->all_boxes && !p->has_drift && !p->hgeo[1].present;
    const bool iou = p->status_cfg.check_no_action || p->status_cfg.check_arrival;
    if (!p->fused_step || p->grid_tier || p->has_drift || p->scene_regen) return T2D_FORM_UNFUSED;
    const bool chain = p->chain_steps && n_steps >= 2 && (ego ? p->chain_loop : !iou);
    if (p->idm_on) {
        if (ego || !idm_in_step(p)) return T2D_FORM_UNFUSED;
        return !chain ? T2D_FORM_STEP : idm_in_pipe(p) ? T2D_FORM_LOOP_PIPE : T2D_FORM_CHAIN;
    }
    if (ego) {
        if (p->device_cus == 0) (void)hipDeviceGetAttribute(&p->device_cus, hipDeviceAttributeMultiprocessorCount, p->device);
        if (chain && p->chain_pipe && p->device_cus > 0 && (p->v.n_env + 15) / 16 <= p->device_cus) return T2D_FORM_EGO_LOOP_PIPE;
        return chain ? T2D_FORM_EGO_LOOP : T2D_FORM_EGO;
    }
    const bool split = use_split(p);
    if (!chain) return split ? T2D_FORM_STEP_SPLIT : T2D_FORM_STEP;
    const int wgs = (p->v.n_env + p->v.geo_layout.epb - 1) / p->v.geo_layout.epb;
    if (p->device_cus == 0) (void)hipDeviceGetAttribute(&p->device_cus, hipDeviceAttributeMultiprocessorCount, p->device);
    const bool loop_ok = p->chain_loop && p->device_cus > 0 && wgs <= T2D_LOOP_MAX_WGS_PER_CU * p->device_cus;
    if (loop_ok && p->chain_pipe && wgs <= p->device_cus && p->v.A <= 64) return T2D_FORM_LOOP_PIPE;
    if (split) return T2D_FORM_CHAIN_SPLIT;
    return loop_ok ? T2D_FORM_LOOP : T2D_FORM_CHAIN;
}

int t2d_set_split_step(t2d_pool* p, int32_t on) {
    if (!p) return T2D_ERR_INVALID;
    p->split_steps = on != 0;
    return T2D_OK;
}

int t2d_set_step_chaining(t2d_pool* p, int32_t on, int32_t priority_rule) {
    if (!p) return T2D_ERR_INVALID;
    p->chain_steps = on != 0;
    p->chain_loop = on != 2;   // 2: always the chained form, also for small pools (measurements)
    p->chain_pipe = on == 1 || on == 4;   // 3: small pools loop without the integrator waves (measurements, tests)
    p->chain_pipe2 = on == 1;             // 4: integrator waves, but no lane waves
    p->chain_priority = priority_rule != 0;
    return T2D_OK;
}

int t2d_step_groups(t2d_pool* const* pools, const float* const* act0_dev, const float* const* act1_dev,
                    void* const* hip_streams, int32_t n, int32_t interval_ms) {
    if (!pools || !hip_streams || n <= 0 || (act0_dev == nullptr) != (act1_dev == nullptr)) return T2D_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        int rc;
        if (!pools[i]) return T2D_ERR_INVALID;
        if (act0_dev && (rc = t2d_bind_actions(pools[i], act0_dev[i], act1_dev[i])) != T2D_OK) return rc;
        pools[i]->v.overlapped = n > 1;   // several groups in flight: the kernels' wave priorities favour retiring workgroups
        rc = t2d_step(pools[i], interval_ms, hip_streams[i]);
        pools[i]->v.overlapped = 0;
        if (rc != T2D_OK) return rc;
    }
    return T2D_OK;
}

int t2d_set_fused_step(t2d_pool* p, int32_t on) {
    if (!p) return T2D_ERR_INVALID;
    p->fused_step = on != 0;
    return T2D_OK;
}

int t2d_set_ego_kernel(t2d_pool* p, int32_t on) {
    if (!p) return T2D_ERR_INVALID;
    p->ego_kernel = on != 0;
    return T2D_OK;
}

int t2d_snapshot(t2d_pool* p) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_reset) return fail(p, T2D_ERR_STATE, "t2d_reset must precede t2d_snapshot");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    const size_t nb = 4 * (size_t)p->v.N;
    float* src[6] = {p->v.x, p->v.y, p->v.heading, p->v.speed, p->v.vx, p->v.vy};
    for (int k = 0; k < 6; ++k) {
        if (!p->d_snap[k]) T2D_HIP(p, hipMalloc((void**)&p->d_snap[k], nb));
        T2D_HIP(p, hipMemcpy(p->d_snap[k], src[k], nb, hipMemcpyDeviceToDevice));
    }
    for (int k = 0; k < 2; ++k) {
        if (!p->d_snap_omega[k]) T2D_HIP(p, hipMalloc((void**)&p->d_snap_omega[k], nb));
        T2D_HIP(p, hipMemcpy(p->d_snap_omega[k], k ? p->v.omega_r : p->v.omega_f, nb, hipMemcpyDeviceToDevice));
        p->v.snap_omega[k] = p->has_drift ? p->d_snap_omega[k] : nullptr;
    }
    if (!p->d_snap_ids) T2D_HIP(p, hipMalloc((void**)&p->d_snap_ids, nb));
    T2D_HIP(p, hipMemcpy(p->d_snap_ids, p->v.ids, nb, hipMemcpyDeviceToDevice));
    T2D_HIP(p, hipMemcpy(p->d_snap_min_dist, p->d_min_dist, sizeof(double) * p->v.n_env, hipMemcpyDeviceToDevice));
    p->have_snapshot = true;
    for (int k = 0; k < 6; ++k) p->v.snap[k] = p->d_snap[k];
    p->v.snap_ids = p->d_snap_ids;
    p->v.auto_reset = p->auto_reset ? 1 : 0;
    return T2D_OK;
}

int t2d_set_auto_reset(t2d_pool* p, int32_t on) {
    if (!p) return T2D_ERR_INVALID;
    if (on && !p->have_snapshot) return fail(p, T2D_ERR_STATE, "t2d_snapshot must precede t2d_set_auto_reset");
    p->auto_reset = on != 0;
    p->v.auto_reset = p->auto_reset ? 1 : 0;
    return T2D_OK;
}

int t2d_restore(t2d_pool* p, int32_t mode, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_snapshot) return fail(p, T2D_ERR_STATE, "t2d_snapshot must precede t2d_restore");
    if (mode != 0 && mode != 1) return fail(p, T2D_ERR_INVALID, "mode must be 0 (all) or 1 (done envs)");
    touch(p, (hipStream_t)hip_stream);
    if (mode == 0) p->chain_failed = false;   // (see t2d_reset)
    T2D_HIP(p, t2d::launch_restore(p->v, p->d_snap, p->d_snap_ids, mode, (hipStream_t)hip_stream));
    return T2D_OK;
}

int t2d_parking_scenes(t2d_pool* p, uint64_t seed, int64_t first_env, int64_t env_stride, double type_proportion,
                       double vehicle_length, double vehicle_width, int32_t regenerate) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_params) return fail(p, T2D_ERR_STATE, "t2d_set_param_table must precede t2d_parking_scenes");
    if (p->v.A != 1) return fail(p, T2D_ERR_INVALID, "generated parking scenes need a pool with one participant per env");
    if (p->has_drift) return fail(p, T2D_ERR_INVALID, "SingleTrackDrift agents are not supported in generated parking scenes");
    if (p->hgeo[1].present) return fail(p, T2D_ERR_STATE, "lane geometry cannot be combined with generated parking scenes");
    if (env_stride < 0) return fail(p, T2D_ERR_INVALID, "env_stride must be >= 0");
    if (regenerate != 0 && regenerate != 1 && regenerate != 2)
        return fail(p, T2D_ERR_INVALID, "regenerate must be 0 (off), 1 (staged ahead) or 2 (generated in the step's stream)");
    if (vehicle_length < vehicle_width || !(vehicle_length > 0.0) || !(vehicle_width > 0.0)) {
        vehicle_length = 5.3;  // ParkingLotGenerator.__init__ :45-57
        vehicle_width = 2.5;
    }
    if (!(type_proportion >= 0.0)) type_proportion = 0.0;
    if (type_proportion > 1.0) type_proportion = 1.0;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    const int E = p->v.n_env;
    constexpr int K = T2D_GEN_MAX_QUADS;
    int rc;
    // geometry records in capacity layout: K polygon slots of 4 vertices per env, a dead slot = a box nothing meets
    t2d::GeoLayout gl{};
    const int log2A = log2_pad(1);
    int epb = 256 >> log2A;
    for (;; epb >>= 1) {
        const int mp[2] = {K * epb, 0}, mv[2] = {4 * K * epb, 0};
        fill_layout(gl, epb, mp, mv);
        if (gl.stride <= 8192) break;
        if ((epb << log2A) <= 64) return fail(p, T2D_ERR_GEOMETRY, "scene record exceeds the 32 KiB LDS record");
    }
    gl.has[0] = 1;
    gl.has[1] = 0;
    const int nb = (E + epb - 1) / epb;
    {
        std::vector<uint32_t> rec((size_t)nb * gl.stride, 0u);
        const float dead[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
        for (int b = 0; b < nb; ++b) {
            uint32_t* r = rec.data() + (size_t)b * gl.stride;
            int32_t* pstart = reinterpret_cast<int32_t*>(r) + gl.off_pstart[0];
            int32_t* vstart = reinterpret_cast<int32_t*>(r) + gl.off_vstart[0];
            for (int el = 0; el <= epb; ++el) pstart[el] = K * el;
            for (int q = 0; q <= K * epb; ++q) vstart[q] = 4 * q;
            float* bb = reinterpret_cast<float*>(r) + gl.off_aabb[0];
            for (int q = 0; q < K * epb; ++q) memcpy(bb + 4 * q, dead, sizeof dead);
        }
        if ((rc = dev_replace(p, &p->d_geo, rec.data(), rec.size()))) return rc;
    }
    p->hgeo[0] = t2d_pool::HostGeo{};
    p->v.geo = p->d_geo;
    p->v.geo_layout = gl;
    p->v.wgmap = nullptr;   // the launch shape may have changed
    {
        std::vector<float> zf(4 * (size_t)E, 0.f);
        std::vector<double> zd(8 * (size_t)E, 0.0);
        if ((rc = dev_replace(p, &p->d_boundary, zf.data(), zf.size()))) return rc;
        if ((rc = dev_replace<uint8_t>(p, &p->d_boundary_valid, nullptr, 0))) return rc;
        if ((rc = dev_replace(p, &p->d_target_xy, zd.data(), 8 * (size_t)E))) return rc;
        if ((rc = dev_replace(p, &p->d_target_c, zd.data(), 2 * (size_t)E))) return rc;
        std::vector<float> zl(16 * (size_t)K * E, 0.f);   // 4 edge records of 4 floats per polygon slot
        std::vector<int32_t> zi(E, 0);
        if ((rc = dev_replace(p, &p->d_lidar_xy, zl.data(), zl.size()))) return rc;
        if ((rc = dev_replace(p, &p->d_lidar_cnt, zi.data(), zi.size()))) return rc;
        std::vector<uint8_t> zm(4 * (size_t)K * E, 0xff);   // per edge: its ring's slot if the ring may cull (install_quad_slot)
        if ((rc = dev_replace(p, &p->d_lidar_meta, zm.data(), zm.size()))) return rc;
    }
    p->v.boundary = p->d_boundary;
    p->v.boundary_valid = nullptr;
    p->v.target_xy = p->d_target_xy;
    p->v.target_c = p->d_target_c;
    p->have_target = true;
    const size_t nbytes = 4 * (size_t)p->v.N;
    for (int k = 0; k < 6; ++k)
        if (!p->d_snap[k]) T2D_HIP(p, hipMalloc((void**)&p->d_snap[k], nbytes));
    if (!p->d_snap_ids) T2D_HIP(p, hipMalloc((void**)&p->d_snap_ids, nbytes));
    // the per-scene arrays (layout of t2d_generate_parking's outputs): live [E] + the staging ring [E * ring] used when
    // scenes are regenerated (regenerate == 1; == 2 generates on the step's stream instead), episode counters
    const int ring = regenerate == 1 ? kSceneRing : 0;
    const size_t per[8] = {(size_t)K * 8 * sizeof(float), (size_t)K * sizeof(int32_t), sizeof(int32_t), 3 * sizeof(double),
                           8 * sizeof(float), sizeof(double), 4 * sizeof(float), sizeof(uint32_t)};
    const size_t counts[2] = {(size_t)E, (size_t)E * ring};
    size_t off[21], total = 0;
    for (int set = 0; set < 2; ++set)
        for (int k = 0; k < 8; ++k) {
            off[8 * set + k] = total;
            total += (per[k] * counts[set] + 255) & ~(size_t)255;
        }
    off[16] = total; total += ((size_t)E * sizeof(int32_t) + 255) & ~(size_t)255;          // episode
    off[17] = total; total += ((size_t)E * ring * sizeof(int32_t) + 255) & ~(size_t)255;   // staged_ep
    off[18] = total; total += 256;                                                         // commit_err
    off[19] = total; total += 256;                                                         // refill_count
    off[20] = total; total += ((size_t)E * ring * sizeof(uint2) + 255) & ~(size_t)255;     // refill_list
    if (p->scene_stream) T2D_HIP(p, hipStreamSynchronize(p->scene_stream));
    if (p->d_scene_arrays) {
        T2D_HIP(p, hipFree(p->d_scene_arrays));
        p->d_scene_arrays = nullptr;
    }
    T2D_HIP(p, hipMalloc(&p->d_scene_arrays, total));
    T2D_HIP(p, hipMemset(p->d_scene_arrays, 0, total));
    char* base = (char*)p->d_scene_arrays;
    t2d::SceneView& sv = p->scene;
    sv = t2d::SceneView{};
    sv.seed = seed; sv.first_env = first_env; sv.env_stride = env_stride;
    sv.type_proportion = type_proportion; sv.len = vehicle_length; sv.wid = vehicle_width;
    t2d::SceneArrays* sets[2] = {&sv.live, &sv.staged};
    for (int set = 0; set < 2; ++set) {
        char* b = base;
        *sets[set] = t2d::SceneArrays{(float*)(b + off[8 * set + 0]), (int32_t*)(b + off[8 * set + 1]), (int32_t*)(b + off[8 * set + 2]),
                                      (double*)(b + off[8 * set + 3]), (float*)(b + off[8 * set + 4]), (double*)(b + off[8 * set + 5]),
                                      (float*)(b + off[8 * set + 6]), (uint32_t*)(b + off[8 * set + 7])};
    }
    sv.episode = (int32_t*)(base + off[16]);
    sv.staged_ep = (int32_t*)(base + off[17]);
    sv.commit_err = (uint32_t*)(base + off[18]);
    sv.refill_count = (uint32_t*)(base + off[19]);
    sv.refill_list = (uint2*)(base + off[20]);
    p->scene_commit_used = p->scene_commit_failed = false;
    sv.ring = ring;
    if (ring > 0) {
        T2D_HIP(p, hipMemset(sv.staged_ep, 0xff, (size_t)E * ring * sizeof(int32_t)));   // -1 = empty slot
        if (!p->scene_stream) T2D_HIP(p, hipStreamCreateWithFlags(&p->scene_stream, hipStreamNonBlocking));
        if (!p->ev_scene_commit) T2D_HIP(p, hipEventCreateWithFlags(&p->ev_scene_commit, hipEventDisableTiming));
        if (!p->ev_scene_refill) T2D_HIP(p, hipEventCreateWithFlags(&p->ev_scene_refill, hipEventDisableTiming));
    }
    p->scene_refill_pending = false;
    sv.geo = p->d_geo; sv.gl = gl;
    sv.lidar_xy = p->d_lidar_xy; sv.lidar_cnt = p->d_lidar_cnt; sv.lidar_meta = p->d_lidar_meta;
    sv.boundary = p->d_boundary; sv.target_xy = p->d_target_xy; sv.target_c = p->d_target_c;
    for (int k = 0; k < 6; ++k) sv.snap[k] = p->d_snap[k];
    sv.snap_ids = p->d_snap_ids;
    sv.snap_min_dist = p->d_snap_min_dist;
    sv.ids_word = ((uint32_t)(int)p->host_params[0][T2D_P_MODEL] << t2d::kIdsModelShift) | (0u << t2d::kIdsTypeShift) |
                  (1u << t2d::kIdsActiveShift);
    p->scene_mode = true;
    p->scene_regen = regenerate != 0;
    if (!p->d_scene_view) T2D_HIP(p, hipMalloc((void**)&p->d_scene_view, sizeof(t2d::SceneView)));
    T2D_HIP(p, hipMemcpy(p->d_scene_view, &sv, sizeof(sv), hipMemcpyHostToDevice));   // (what the ego step kernel's epilogue reads)
    if ((rc = rebuild_lidar_geo(p))) return rc;
    {  // t2d_reset's remaining columns: wheel speeds start at zero
        for (int f : {T2D_F_OMEGA_F, T2D_F_OMEGA_R}) T2D_HIP(p, hipMemset(p->field_ptr[f], 0, nbytes));
    }
    touch(p, nullptr);   // the two set-up launches below run on the null stream
    T2D_HIP(p, t2d::launch_parking_scenes(p->v, sv, E, 1, nullptr));
    if (ring > 0) T2D_HIP(p, t2d::launch_scene_refill(sv, E, nullptr));   // episodes 1 .. ring of every env
    T2D_HIP(p, quiesce(p));
    p->have_reset = true;
    p->have_snapshot = true;
    for (int k = 0; k < 6; ++k) p->v.snap[k] = p->d_snap[k];
    p->v.snap_ids = p->d_snap_ids;
    p->v.snap_omega[0] = p->v.snap_omega[1] = nullptr;
    p->v.auto_reset = p->auto_reset ? 1 : 0;
    return T2D_OK;
}

int t2d_get_parking_scenes(t2d_pool* p, float* quads, int32_t* quad_id, int32_t* n_quads, double* start, float* target,
                           double* target_heading, float* boundary, uint32_t* info, int32_t* episode) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->scene_mode) return fail(p, T2D_ERR_STATE, "t2d_parking_scenes has not been called on this pool");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    const size_t E = (size_t)p->v.n_env;
    constexpr size_t K = T2D_GEN_MAX_QUADS;
    const t2d::SceneView& sv = p->scene;
    if (quads) T2D_HIP(p, hipMemcpy(quads, sv.live.quads, E * K * 8 * sizeof(float), hipMemcpyDeviceToHost));
    if (quad_id) T2D_HIP(p, hipMemcpy(quad_id, sv.live.quad_id, E * K * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_quads) T2D_HIP(p, hipMemcpy(n_quads, sv.live.n_quads, E * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (start) T2D_HIP(p, hipMemcpy(start, sv.live.start, E * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (target) T2D_HIP(p, hipMemcpy(target, sv.live.target, E * 8 * sizeof(float), hipMemcpyDeviceToHost));
    if (target_heading) T2D_HIP(p, hipMemcpy(target_heading, sv.live.target_heading, E * sizeof(double), hipMemcpyDeviceToHost));
    if (boundary) T2D_HIP(p, hipMemcpy(boundary, sv.live.boundary, E * 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (info) T2D_HIP(p, hipMemcpy(info, sv.live.info, E * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (episode) T2D_HIP(p, hipMemcpy(episode, sv.episode, E * sizeof(int32_t), hipMemcpyDeviceToHost));
    return T2D_OK;
}

#ifdef T2D_TIMING
// profiling builds only (not part of the ABI): read and clear the phase cycle accumulators
int t2d_debug_read(t2d_pool* p, unsigned long long* out, size_t n_words) {
    if (!p || !p->v.dbg) return T2D_ERR_INVALID;
    T2D_HIP(p, quiesce(p));
    T2D_HIP(p, hipMemcpy(out, p->v.dbg, n_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return T2D_OK;
}
#endif

// ---- multi-GPU: the all-gather of the per-env result records ------------------------------------------
int t2d_comm_unique_id(uint8_t* id) {
    if (!id) return T2D_ERR_INVALID;
    if (!rccl().ok) return fail(nullptr, T2D_ERR_HIP, rccl().err);
    ncclUniqueId u;
    const ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, T2D_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
    static_assert(sizeof(u) == T2D_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, sizeof(u));
    return T2D_OK;
}

static int ensure_gather_objects(t2d_pool* p) {
    if (p->gather_stream) return T2D_OK;
    // Lowest stream priority: the collective's workgroups take the slots the step kernel leaves free at
    // its start and tail instead of displacing step workgroups out of the one-wave-round launch.
    int prio_least = 0, prio_greatest = 0;
    T2D_HIP(p, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    T2D_HIP(p, hipStreamCreateWithPriority(&p->gather_stream, hipStreamNonBlocking, prio_least));
    T2D_HIP(p, hipEventCreateWithFlags(&p->ev_frag_ready, hipEventDisableTiming));
    for (hipEvent_t& e : p->ev_gather) T2D_HIP(p, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return T2D_OK;
}

int t2d_comm_init(t2d_pool* p, const uint8_t* id, int32_t rank, int32_t world) {
    if (!p) return T2D_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(p, T2D_ERR_INVALID, "t2d_comm_init: need 0 <= rank < world");
    if (world > 1 && !id) return fail(p, T2D_ERR_INVALID, "t2d_comm_init: the communicator id of t2d_comm_unique_id is required");
    T2D_HIP(p, hipSetDevice(p->device));
    int rc;
    if ((rc = ensure_gather_objects(p))) return rc;
    if (p->comm) {
        (void)rccl().CommDestroy((ncclComm_t)p->comm);
        p->comm = nullptr;
    }
    p->comm_rank = rank;
    p->comm_world = world;
    if (!id) return T2D_OK;   // a world of one without RCCL: t2d_gather degenerates to a copy on the gather stream
    if (!rccl().ok) return fail(p, T2D_ERR_HIP, rccl().err);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    const ncclResult_t r = rccl().CommInitRank(&c, world, u, rank);
    if (r != ncclSuccess) return fail(p, T2D_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    p->comm = c;
    return T2D_OK;
}

#ifdef T2D_DEBUG_HOOKS   // include/t2d_debug.h: libt2d_hip_debug.so only
int t2d_debug_delay_gather(t2d_pool* p, int32_t microseconds) {
    if (!p) return T2D_ERR_INVALID;
    if (microseconds < 0 || microseconds > 200000) return fail(p, T2D_ERR_INVALID, "delay must be 0 .. 200000 us");
    T2D_HIP(p, hipSetDevice(p->device));
    int rc;
    if ((rc = ensure_gather_objects(p))) return rc;
    T2D_HIP(p, t2d::launch_spin(100ll * microseconds, p->gather_stream));
    return T2D_OK;
}
#endif

int t2d_comm_info(t2d_pool* p, int32_t* native_rccl, int32_t* world, int32_t* rank) {
    if (!p) return T2D_ERR_INVALID;
    int w = p->comm_world, r = p->comm_rank;
    if (p->comm) {   // what the communicator itself says, not what the caller passed to t2d_comm_init
        if (!rccl().CommCount || !rccl().CommUserRank) return fail(p, T2D_ERR_HIP, "librccl lacks ncclCommCount / ncclCommUserRank");
        ncclResult_t e = rccl().CommCount((ncclComm_t)p->comm, &w);
        if (e == ncclSuccess) e = rccl().CommUserRank((ncclComm_t)p->comm, &r);
        if (e != ncclSuccess) return fail(p, T2D_ERR_HIP, std::string("ncclCommCount: ") + rccl().GetErrorString(e));
    }
    if (native_rccl) *native_rccl = p->comm != nullptr;
    if (world) *world = w;
    if (rank) *rank = r;
    return T2D_OK;
}

int64_t t2d_step_count(const t2d_pool* p) { return p ? (int64_t)p->step_count : -1; }

int t2d_gather(t2d_pool* p, void* nccl_comm, int32_t n_steps, void* out_dev, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!out_dev) return fail(p, T2D_ERR_INVALID, "t2d_gather: out_dev is null");
    if (n_steps < 1 || T2D_RECORD_RING % n_steps || p->step_count < n_steps || p->step_count % n_steps)
        return fail(p, T2D_ERR_INVALID, "t2d_gather: n_steps must divide the record ring (" + std::to_string(T2D_RECORD_RING) +
                                            ") and the step count (" + std::to_string(p->step_count) + ") must be a positive multiple of it");
    T2D_HIP(p, hipSetDevice(p->device));
    int rc;
    if ((rc = ensure_gather_objects(p))) return rc;
    ncclComm_t comm = nccl_comm ? (ncclComm_t)nccl_comm : (ncclComm_t)p->comm;
    if (!comm && p->comm_world > 1) return fail(p, T2D_ERR_STATE, "t2d_gather: no communicator (t2d_comm_init)");
    if (comm && !rccl().ok) return fail(p, T2D_ERR_HIP, rccl().err);
    hipStream_t s = (hipStream_t)hip_stream;
    const int first = (int)((p->step_count - n_steps) % T2D_RECORD_RING);   // contiguous: n_steps divides the ring
    const size_t count = (size_t)n_steps * p->v.n_env * 2;                  // u32 words of this rank's fragment
    const uint32_t* src = (const uint32_t*)p->field_ptr[T2D_F_RECORD] + (size_t)first * p->v.n_env * 2;
    // the gather stream picks up after the fragment's last step; the step stream does not wait for the collective
    T2D_HIP(p, hipEventRecord(p->ev_frag_ready, s));
    T2D_HIP(p, hipStreamWaitEvent(p->gather_stream, p->ev_frag_ready, 0));
    if (comm) {
        const ncclResult_t r = rccl().AllGather(src, out_dev, count, ncclUint32, comm, p->gather_stream);
        if (r != ncclSuccess) return fail(p, T2D_ERR_HIP, std::string("ncclAllGather: ") + rccl().GetErrorString(r));
    } else {
        T2D_HIP(p, hipMemcpyAsync(out_dev, src, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, p->gather_stream));
    }
    hipEvent_t done = p->ev_gather[first];
    T2D_HIP(p, hipEventRecord(done, p->gather_stream));
    for (int k = first; k < first + n_steps; ++k) p->slot_event[k] = done;   // whoever overwrites these slots waits for it
    p->last_gather = done;
    return T2D_OK;
}

int t2d_gather_wait(t2d_pool* p, void* hip_stream, int32_t block_host) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->last_gather) return T2D_OK;
    T2D_HIP(p, hipSetDevice(p->device));
    if (block_host) T2D_HIP(p, hipEventSynchronize(p->last_gather));
    else T2D_HIP(p, hipStreamWaitEvent((hipStream_t)hip_stream, p->last_gather, 0));
    return T2D_OK;
}

// test hook: the next CHAIN launches of the pool break one hand-off on purpose (include/t2d.h)
#ifdef T2D_DEBUG_HOOKS   // include/t2d_debug.h: libt2d_hip_debug.so only
const char* t2d_debug_last_step_kernel(void) { return t2d::last_collide_form(); }
#endif
#ifdef T2D_DEBUG_HOOKS   // include/t2d_debug.h: libt2d_hip_debug.so only
int t2d_debug_chain_fault(t2d_pool* p, int32_t kind) {
    if (!p) return T2D_ERR_INVALID;
    if (kind < 0 || kind > 3)
        return fail(p, T2D_ERR_INVALID, "fault kind: 0 none, 1 foreign XCC id at step 1, 2 a hand-off that never comes, 3 foreign XCC id at step 0");
    p->chain_fault = (uint32_t)kind;
    return T2D_OK;
}
#endif

#ifdef T2D_DEBUG_HOOKS   // include/t2d_debug.h: libt2d_hip_debug.so only
// placement of the step launch (include/t2d_debug.h): a permutation of its workgroups + a wave rotation per workgroup
int t2d_debug_set_step_placement(t2d_pool* p, const uint32_t* map_host, int32_t n_workgroups) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    if (!map_host || n_workgroups <= 0) {   // back to the identity
        p->v.wgmap = nullptr;
        return T2D_OK;
    }
    const int epb = p->v.geo_layout.epb > 0 ? p->v.geo_layout.epb : 1;
    const int n_blocks = (p->v.n_env + epb - 1) / epb;
    if (n_workgroups != n_blocks) return fail(p, T2D_ERR_INVALID, "placement map must have one entry per workgroup of the step launch");
    if (n_blocks > 65536) return fail(p, T2D_ERR_INVALID, "placement map: at most 65536 workgroups");   // 16-bit workgroup ids
    std::vector<uint8_t> seen((size_t)n_blocks, 0);
    for (int b = 0; b < n_blocks; ++b) {   // a permutation of the workgroups, rotations 0..3
        const uint32_t g = map_host[b] & 0xffffu, r = map_host[b] >> 16;
        if (g >= (uint32_t)n_blocks || r > 3u || seen[g]) return fail(p, T2D_ERR_INVALID, "placement map is not a permutation with rotations 0..3");
        seen[g] = 1;
    }
    if (!p->d_wgmap) T2D_HIP(p, hipMalloc((void**)&p->d_wgmap, sizeof(uint32_t) * 65536));
    T2D_HIP(p, hipMemcpy(p->d_wgmap, map_host, sizeof(uint32_t) * (size_t)n_blocks, hipMemcpyHostToDevice));
    p->v.wgmap = p->d_wgmap;
    return T2D_OK;
}
#endif

// capacity planning: resident workgroups per CU of the fused step kernel for this pool's geometry, and its LDS bytes per
// workgroup -- the regression guard of tests/test_gpu_api.py
int t2d_step_occupancy(t2d_pool* p, int32_t* blocks_per_cu, int64_t* lds_bytes, int64_t* geometry_bytes_per_launch) {
    if (!p || !blocks_per_cu || !lds_bytes) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    int b = 0;
    size_t l = 0;
    T2D_HIP(p, t2d::step_occupancy(p->v, &b, &l));
    *blocks_per_cu = b;
    *lds_bytes = (int64_t)l;
    if (geometry_bytes_per_launch) {   // the packed records every workgroup of a step launch stages into LDS
        const int epb = p->v.geo_layout.epb > 0 ? p->v.geo_layout.epb : 1;
        const int64_t n_blocks = (p->v.n_env + epb - 1) / epb;
        *geometry_bytes_per_launch = p->v.geo ? n_blocks * (int64_t)p->v.geo_layout.stride * 4 : 0;
    }
    return T2D_OK;
}

// a HIP stream of the library's own making (hipStreamNonBlocking, given priority: 0 = default, negative = higher): env groups
// on streams that do not come out of the caller's framework pool
int t2d_stream_create(int32_t device_id, int32_t priority, void** out_stream) {
    if (!out_stream) return T2D_ERR_INVALID;
    hipStream_t s = nullptr;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) {
        (void)hipGetLastError();
        return T2D_ERR_HIP;
    }
    *out_stream = s;
    return T2D_OK;
}
int t2d_stream_destroy(void* stream) {
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? T2D_OK : T2D_ERR_HIP;
}

int t2d_get_field(t2d_pool* p, int32_t f, void** dev_ptr, size_t* nbytes) {
    if (!p) return T2D_ERR_INVALID;
    if (f < 0 || f >= T2D_F_COUNT || !dev_ptr) return fail(p, T2D_ERR_INVALID, "bad field id");
    *dev_ptr = p->field_ptr[f];
    if (nbytes) *nbytes = p->field_bytes[f];
    return T2D_OK;
}

int t2d_download(t2d_pool* p, int32_t f, void* host_dst, size_t nbytes) {
    if (!p) return T2D_ERR_INVALID;
    if (f < 0 || f >= T2D_F_COUNT || !host_dst || nbytes != p->field_bytes[f])
        return fail(p, T2D_ERR_INVALID, "bad field id / size (expected " +
                                            std::to_string(f >= 0 && f < T2D_F_COUNT ? p->field_bytes[f] : 0) + " bytes)");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    if (p->chain_failed || p->scene_commit_failed) return report_chain_failure(p);
    T2D_HIP(p, hipMemcpy(host_dst, p->field_ptr[f], nbytes, hipMemcpyDeviceToHost));
    return T2D_OK;
}

int t2d_upload(t2d_pool* p, int32_t f, const void* host_src, size_t nbytes) {
    if (!p) return T2D_ERR_INVALID;
    if (f < 0 || f >= T2D_F_COUNT || !host_src || nbytes != p->field_bytes[f])
        return fail(p, T2D_ERR_INVALID, "bad field id / size");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    T2D_HIP(p, hipMemcpy(p->field_ptr[f], host_src, nbytes, hipMemcpyHostToDevice));
    // an ids column written by the caller: which types are in use is no longer known -- every row of the table counts (the
    // integrator's single-model / four-per-lane instantiations are chosen from this set: t2d_integrate)
    if (f == T2D_F_IDS) p->types_used = ~0u;
    // actions uploaded into the pool's own fields are the actions from now on: a binding to caller-owned device memory
    // (t2d_bind_actions) ends here, or the kernels would keep reading the caller's stale tensors
    if (f == T2D_F_ACT0 || f == T2D_F_ACT1) {
        p->v.act0 = (const float*)p->field_ptr[T2D_F_ACT0];
        p->v.act1 = (const float*)p->field_ptr[T2D_F_ACT1];
        p->v.act_stride = 1;
        p->act_extent = 0;
        p->act_in_frame = false;
        refresh_idm_view(p);
    }
    return T2D_OK;
}

int t2d_sync(t2d_pool* p) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    if (p->chain_failed || p->scene_commit_failed) return report_chain_failure(p);
    return T2D_OK;
}

// ---- the Gym-API host path (include/t2d.h: t2d_frame_config / t2d_step_host / t2d_frame_fetch) -------------------------------
static void frame_release(t2d_pool* p) {
    // the staging buffers go: a pool whose actions are the ones t2d_step_host staged there falls back to its own action
    // fields (what t2d_step / t2d_step_n / t2d_step_host(NULL) read from now on), never to freed memory
    if (p->act_in_frame) {
        p->v.act0 = (const float*)p->field_ptr[T2D_F_ACT0];
        p->v.act1 = (const float*)p->field_ptr[T2D_F_ACT1];
        p->v.act_stride = 1;
        p->act_extent = 0;
        p->act_in_frame = false;
        refresh_idm_view(p);
    }
    if (p->d_frame) (void)hipFree(p->d_frame);
    if (p->d_actions) (void)hipFree(p->d_actions);
    for (char*& h : p->h_frame) {
        if (h) (void)hipHostFree(h);
        h = nullptr;
    }
    if (p->h_actions) (void)hipHostFree(p->h_actions);
    p->d_frame = nullptr;
    p->d_actions = nullptr;
    p->h_actions = nullptr;
    p->frame_sections = 0;
    p->n_host_frames = 0;
    p->frame_layout = t2d_frame_layout{};
}

int t2d_frame_config(t2d_pool* p, uint32_t sections, int32_t n_host_frames, t2d_frame_layout* layout) {
    if (!p) return T2D_ERR_INVALID;
    if (n_host_frames < 1 || n_host_frames > T2D_MAX_HOST_FRAMES) return fail(p, T2D_ERR_INVALID, "need 1..16 host frames");
    if (sections & ~(T2D_FRAME_LIDAR | T2D_FRAME_TARGET | T2D_FRAME_ZEROCOPY)) return fail(p, T2D_ERR_INVALID, "unknown frame section bits");
    if ((sections & T2D_FRAME_LIDAR) && !p->lidar_on)
        return fail(p, T2D_ERR_STATE, "t2d_lidar_config must precede a frame with a lidar section");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    const size_t E = (size_t)p->v.n_env;
    t2d_frame_layout L{};
    size_t total = 256;   // header
    auto section = [&](size_t bytes) {
        const size_t off = total;
        total += (bytes + 255) & ~(size_t)255;
        return (int64_t)off;
    };
    L.off_rel = section(E * 3 * sizeof(double));
    L.off_obs = section(E * 6 * sizeof(float));
    L.off_reward = section(E * 4);
    L.off_status = section(E * 4);
    L.off_iou = section(E * 4);
    L.off_frame_ms = section(E * 4);
    L.off_cnt_step = section(E * 4);
    L.off_episode = section(E * 4);
    L.off_target = L.off_target_heading = L.off_lidar = -1;
    if (sections & T2D_FRAME_TARGET) {
        L.off_target_heading = section(E * sizeof(double));
        L.off_target = section(E * 8 * sizeof(float));
    }
    L.n_env = p->v.n_env;
    L.n_beams = 0;
    if (sections & T2D_FRAME_LIDAR) {
        L.n_beams = p->lidar.n_beams;
        L.off_lidar = section(E * (size_t)L.n_beams * sizeof(float));
    }
    L.bytes = (int64_t)total;
    // the same configuration again (every VecParkingEnv.reset asks): keep the frames -- a caller may still hold views of them
    if ((p->frame_sections & 0x7fffffffu) == sections && (p->frame_sections & 0x80000000u) && p->n_host_frames == n_host_frames &&
        memcmp(&p->frame_layout, &L, sizeof L) == 0) {
        if (layout) *layout = L;
        return T2D_OK;
    }
    frame_release(p);
    const unsigned host_flags = hipHostMallocMapped | hipHostMallocCoherent;
    for (int k = 0; k < n_host_frames; ++k) {
        T2D_HIP(p, hipHostMalloc((void**)&p->h_frame[k], total, host_flags));
        memset(p->h_frame[k], 0, total);
    }
    p->n_host_frames = n_host_frames;
    const size_t act_bytes = (size_t)p->v.N * 2 * sizeof(float);
    T2D_HIP(p, hipHostMalloc((void**)&p->h_actions, act_bytes, host_flags));
    memset(p->h_actions, 0, act_bytes);
    if (!(sections & T2D_FRAME_ZEROCOPY)) {
        T2D_HIP(p, hipMalloc((void**)&p->d_frame, total));
        T2D_HIP(p, hipMemset(p->d_frame, 0, total));
        T2D_HIP(p, hipMalloc((void**)&p->d_actions, act_bytes));
        T2D_HIP(p, hipMemset(p->d_actions, 0, act_bytes));
    }
    p->frame_sections = sections | 0x80000000u;   // (configured, whatever the section bits)
    p->frame_layout = L;
    p->frame_turn = 0;
    if (layout) *layout = L;
    return T2D_OK;
}

int t2d_set_target_headings(t2d_pool* p, const double* heading_host) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    return dev_replace(p, &p->d_target_heading, heading_host, heading_host ? (size_t)p->v.n_env : 0);
}

// scan (optional) + pack + fetch of the current state on `s`; returns with the host frame filled
static int frame_finish(t2d_pool* p, hipStream_t s, int frame_index, const void** frame_host) {
    const t2d_frame_layout& L = p->frame_layout;
    const bool zero_copy = (p->frame_sections & T2D_FRAME_ZEROCOPY) != 0;
    if (frame_index < 0) {
        frame_index = p->frame_turn;
        p->frame_turn = (p->frame_turn + 1) % p->n_host_frames;
    }
    char* host = p->h_frame[frame_index];
    char* dev_out = p->d_frame;
    if (zero_copy) T2D_HIP(p, hipHostGetDevicePointer((void**)&dev_out, host, 0));
    int rc;
    if (L.off_lidar >= 0) {
        if (!p->lidar_on || p->lidar.n_beams != L.n_beams)
            return fail(p, T2D_ERR_STATE, "the lidar configuration changed: call t2d_frame_config again");
        if ((rc = t2d_lidar_scan(p, reinterpret_cast<float*>(dev_out + L.off_lidar), s))) return rc;
    }
    t2d::FrameView fv{};
    fv.out = dev_out;
    fv.lay = L;
    fv.target_heading = p->scene_mode ? p->scene.live.target_heading : p->d_target_heading;
    fv.target_quads = p->scene_mode ? p->scene.live.target : nullptr;
    fv.episode = p->scene_mode ? p->scene.episode : nullptr;
    fv.commit_err = p->scene_mode && p->scene_regen ? p->scene.commit_err : nullptr;
    fv.step_count = (uint32_t)p->step_count;
    fv.ego_index = p->status_cfg.ego_index;
    touch(p, s);
    T2D_HIP(p, t2d::launch_frame_pack(p->v, fv, s));
    if (!zero_copy) T2D_HIP(p, hipMemcpyAsync(host, p->d_frame, (size_t)L.bytes, hipMemcpyDeviceToHost, s));
    T2D_HIP(p, hipStreamSynchronize(s));
    // `s` is drained: drop it from the streams a later quiesce has to wait for
    for (int k = 0; k < p->n_live_streams; ++k)
        if (p->live_streams[k] == s) {
            p->live_streams[k] = p->live_streams[--p->n_live_streams];
            break;
        }
    if (frame_host) *frame_host = host;
    if (fv.commit_err && reinterpret_cast<const uint32_t*>(host)[1]) {   // (what quiesce would have found with a blocking copy)
        (void)hipMemset(p->scene.commit_err, 0, sizeof(uint32_t));
        p->scene_commit_failed = true;
    }
    if (p->n_live_streams == 0 && !p->live_overflow && !p->chain_used) p->scene_commit_used = false;
    if (p->scene_commit_failed) return report_chain_failure(p);
    return T2D_OK;
}

int t2d_step_host(t2d_pool* p, const float* actions_host, const float* action_box, int32_t interval_ms, void* hip_stream,
                  int32_t frame_index, const void** frame_host) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->frame_sections) return fail(p, T2D_ERR_STATE, "t2d_frame_config must precede t2d_step_host");
    if (frame_index >= p->n_host_frames) return fail(p, T2D_ERR_INVALID, "frame_index out of range");
    T2D_HIP(p, hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (actions_host) {
        const size_t act_bytes = (size_t)p->v.N * 2 * sizeof(float);
        const bool in_place = actions_host == p->h_actions;   // the caller filled the pool's own pinned buffer (t2d_host_action_buffer)
        if (action_box) {   // Box.contains for every row, in the pass that stages the actions
            const float lo0 = action_box[0], hi0 = action_box[1], lo1 = action_box[2], hi1 = action_box[3];
            float* dst = p->h_actions;
            int ok = 1;
            // the verdict first, the copy only behind it: rejected rows (NaN included) never reach the staging buffer, which
            // in zero-copy mode IS what the previous call's binding reads -- a t2d_step after an InvalidAction steps with the
            // last accepted actions (a caller that filled the pool's own buffer in place has overwritten them itself)
            for (size_t i = 0, n = (size_t)p->v.N; i < n; ++i) {
                const float a = actions_host[2 * i], b = actions_host[2 * i + 1];
                ok &= (int)(a >= lo0) & (int)(a <= hi0) & (int)(b >= lo1) & (int)(b <= hi1);
            }
            if (ok && !in_place) memcpy(dst, actions_host, act_bytes);
            if (!ok) {
                size_t bad = 0;
                for (size_t n = (size_t)p->v.N; bad < n; ++bad) {
                    const float a = actions_host[2 * bad], b = actions_host[2 * bad + 1];
                    if (!(a >= lo0 && a <= hi0 && b >= lo1 && b <= hi1)) break;
                }
                return fail(p, T2D_ERR_ACTION, "action row " + std::to_string(bad) + " is not in the action space");
            }
        } else if (!in_place) {
            memcpy(p->h_actions, actions_host, act_bytes);
        }
        const float* dev_act = p->d_actions;
        if (p->frame_sections & T2D_FRAME_ZEROCOPY) {
            T2D_HIP(p, hipHostGetDevicePointer((void**)&dev_act, p->h_actions, 0));
        } else {
            touch(p, s);
            T2D_HIP(p, hipMemcpyAsync(p->d_actions, p->h_actions, act_bytes, hipMemcpyHostToDevice, s));
        }
        // (steering, accel) per participant: act0 (accel) = element 1, act1 (steering) = element 0, stride 2
        p->v.act0 = dev_act + 1;
        p->v.act1 = dev_act;
        p->v.act_stride = 2;
        p->act_extent = 0;
        p->act_in_frame = true;
        refresh_idm_view(p);
    }
    int rc = t2d_step(p, interval_ms, hip_stream);
    if (rc != T2D_OK) return rc;
    return frame_finish(p, s, frame_index, frame_host);
}

int t2d_host_action_buffer(t2d_pool* p, float** actions_host) {
    if (!p || !actions_host) return T2D_ERR_INVALID;
    if (!p->frame_sections) return fail(p, T2D_ERR_STATE, "t2d_frame_config must precede t2d_host_action_buffer");
    *actions_host = p->h_actions;
    return T2D_OK;
}

int t2d_frame_fetch(t2d_pool* p, void* hip_stream, int32_t frame_index, const void** frame_host) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->frame_sections) return fail(p, T2D_ERR_STATE, "t2d_frame_config must precede t2d_frame_fetch");
    if (frame_index >= p->n_host_frames) return fail(p, T2D_ERR_INVALID, "frame_index out of range");
    if (!p->have_params || !p->have_reset) return fail(p, T2D_ERR_STATE, "t2d_reset must precede t2d_frame_fetch");
    T2D_HIP(p, hipSetDevice(p->device));
    return frame_finish(p, (hipStream_t)hip_stream, frame_index, frame_host);
}

int t2d_lidar_config(t2d_pool* p, int32_t n_beams, float max_range, int32_t include_participants,
                     const double* beam_sin, const double* beam_cos) {
    if (!p) return T2D_ERR_INVALID;
    if (n_beams < 1 || n_beams > 4096 || !(max_range > 0.0f))
        return fail(p, T2D_ERR_INVALID, "need 1 <= n_beams <= 4096 and max_range > 0");
    if ((beam_sin == nullptr) != (beam_cos == nullptr)) return fail(p, T2D_ERR_INVALID, "pass both beam tables or neither");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    std::vector<double> bs(n_beams), bc(n_beams);
    for (int k = 0; k < n_beams; ++k) {
        if (beam_sin) {
            bs[k] = beam_sin[k];
            bc[k] = beam_cos[k];
        } else {  // linspace(0, 2 pi, n, endpoint=False) = k * (2 pi / n)
            const double th = (double)k * ((2.0 * 3.141592653589793) / (double)n_beams);
            bs[k] = sin(th);
            bc[k] = cos(th);
        }
    }
    int rc;
    {   // what the determinant solve needs of a beam, once per beam instead of once per candidate (same expressions as
        // lidar.py:161-162, 201-213: a, b of the beam's line, its end point (lx, ly) = (cos, sin) * R widened by 1e-8)
        const double R = (double)max_range, tz = 1e-8;
        std::vector<double> pre(6 * (size_t)n_beams);
        for (int k = 0; k < n_beams; ++k) {
            const double lx = bc[k] * R, ly = bs[k] * R;
            const double mx = tz > lx ? tz : lx, nx = -tz < lx ? -tz : lx;
            const double my = tz > ly ? tz : ly, ny = -tz < ly ? -tz : ly;
            double* r = &pre[6 * (size_t)k];
            r[0] = bs[k]; r[1] = -bc[k]; r[2] = mx + tz; r[3] = nx - tz; r[4] = my + tz; r[5] = ny - tz;
        }
        if ((rc = dev_replace(p, &p->d_beam_sin, pre.data(), pre.size()))) return rc;
    }
    // the scan buffer is kept when the beam count is unchanged (every VecParkingEnv.reset reconfigures the lidar): a
    // zero-copy view a caller took with t2d_get_field stays valid until the size really changes
    const size_t lidar_bytes = (size_t)p->v.n_env * n_beams * sizeof(float);
    if (!p->field_ptr[T2D_F_LIDAR] || p->field_bytes[T2D_F_LIDAR] != lidar_bytes) {
        if (p->field_ptr[T2D_F_LIDAR]) {
            T2D_HIP(p, hipFree(p->field_ptr[T2D_F_LIDAR]));
            p->field_ptr[T2D_F_LIDAR] = nullptr;
        }
        p->field_bytes[T2D_F_LIDAR] = lidar_bytes;
        T2D_HIP(p, hipMalloc(&p->field_ptr[T2D_F_LIDAR], lidar_bytes));
    }
    T2D_HIP(p, hipMemset(p->field_ptr[T2D_F_LIDAR], 0, lidar_bytes));
    p->lidar.beam_pre = p->d_beam_sin;
    p->lidar.max_range = (double)max_range;
    p->lidar.n_beams = n_beams;
    p->lidar.include_participants = include_participants != 0;
    p->lidar_on = true;
    rc = rebuild_lidar_geo(p);
    if (rc != T2D_OK) return rc;
    return T2D_OK;
}

int t2d_lidar_scan(t2d_pool* p, float* out_dev, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->lidar_on) return fail(p, T2D_ERR_STATE, "t2d_lidar_config must precede t2d_lidar_scan");
    if (!p->have_params || !p->have_reset) return fail(p, T2D_ERR_STATE, "t2d_reset must precede t2d_lidar_scan");
    p->lidar.ego_index = p->status_cfg.ego_index;
    hipStream_t s = (hipStream_t)hip_stream;
    int rc;
    touch(p, s);
    if ((rc = record_event(p, 3, s, true))) return rc;
    T2D_HIP(p, t2d::launch_lidar(p->v, p->lidar, out_dev ? out_dev : (float*)p->field_ptr[T2D_F_LIDAR], s));
    return record_event(p, 3, s, false);
}

int t2d_set_idm(t2d_pool* p, const double* ctrl_rows, int32_t n_ctrl, int32_t row_stride, const uint8_t* ctrl_id) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    if (n_ctrl == 0) {
        p->idm_on = false;
        refresh_idm_view(p);
        return T2D_OK;
    }
    if (!ctrl_rows || !ctrl_id || n_ctrl < 0 || n_ctrl >= T2D_IDM_NONE || row_stride < T2D_IDM_COLS)
        return fail(p, T2D_ERR_INVALID, "t2d_set_idm: need 1..254 parameter sets of >= 8 columns and a controller id per participant");
    std::vector<double> rows((size_t)n_ctrl * T2D_IDM_COLS);
    for (int c = 0; c < n_ctrl; ++c) {
        const double* r = ctrl_rows + (size_t)c * row_stride;
        for (int k = 0; k < T2D_IDM_COLS; ++k) rows[(size_t)c * T2D_IDM_COLS + k] = r[k];
        // idm_controller.py divides by sqrt(max_acceleration * comfortable_deceleration) (:121)
        if (!(r[T2D_IDM_MAX_ACCEL] * r[T2D_IDM_COMF_DECEL] > 0.0))
            return fail(p, T2D_ERR_INVALID, "t2d_set_idm: max_acceleration * comfortable_deceleration must be positive (row " +
                                                std::to_string(c) + ")");
        if (!(r[T2D_IDM_LANE_HALF_WIDTH] >= 0.0) || !(r[T2D_IDM_HORIZON] > 0.0))
            return fail(p, T2D_ERR_INVALID, "t2d_set_idm: lane_half_width >= 0 and horizon > 0 required (row " +
                                                std::to_string(c) + ")");
    }
    for (int i = 0; i < p->v.N; ++i)
        if (ctrl_id[i] != T2D_IDM_NONE && ctrl_id[i] >= n_ctrl)
            return fail(p, T2D_ERR_INVALID, "t2d_set_idm: ctrl_id[" + std::to_string(i) + "] = " +
                                                std::to_string((int)ctrl_id[i]) + " out of range");
    int rc;
    if ((rc = dev_replace(p, &p->d_idm_rows, rows.data(), rows.size()))) return rc;
    if ((rc = dev_replace(p, &p->d_idm_ctrl, ctrl_id, (size_t)p->v.N))) return rc;
    p->idm.rows = p->d_idm_rows;
    p->idm.ctrl_id = p->d_idm_ctrl;
    p->idm.leader = (int32_t*)p->field_ptr[T2D_F_LEADER];
    p->idm.n_ctrl = n_ctrl;
    T2D_HIP(p, hipMemset(p->field_ptr[T2D_F_LEADER], 0xff, p->field_bytes[T2D_F_LEADER]));
    p->idm_on = true;
    refresh_idm_view(p);
    return T2D_OK;
}

int t2d_idm_actions(t2d_pool* p, const int32_t* forced_leader_dev, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->idm_on) return fail(p, T2D_ERR_STATE, "t2d_set_idm must precede t2d_idm_actions");
    if (!p->have_reset) return fail(p, T2D_ERR_STATE, "t2d_reset must precede t2d_idm_actions");
    return idm_impl(p, (hipStream_t)hip_stream, forced_leader_dev);
}

int t2d_verify_state(t2d_pool* p, const float* x_dev, const float* y_dev, const float* heading_dev,
                     const float* speed_dev, int32_t interval_ms, uint8_t* valid_dev, void* hip_stream) {
    if (!p) return T2D_ERR_INVALID;
    if (!p->have_params || !p->have_reset)
        return fail(p, T2D_ERR_STATE, "t2d_set_param_table and t2d_reset must precede t2d_verify_state");
    if (!x_dev || !y_dev || !heading_dev || !speed_dev || !valid_dev)
        return fail(p, T2D_ERR_INVALID, "t2d_verify_state: null device array");
    if (interval_ms < 0) return fail(p, T2D_ERR_INVALID, "interval_ms must be >= 0");
    touch(p, (hipStream_t)hip_stream);
    T2D_HIP(p, t2d::launch_verify(p->v, x_dev, y_dev, heading_dev, speed_dev, interval_ms, valid_dev,
                                  (hipStream_t)hip_stream));
    return T2D_OK;
}

int t2d_set_outputs(t2d_pool* p, uint32_t mask) {
    if (!p) return T2D_ERR_INVALID;
    if (mask & ~T2D_OUT_ALL) return fail(p, T2D_ERR_INVALID, "unknown output bit");
    p->v.out_mask = (int32_t)mask;
    return T2D_OK;
}

int t2d_set_integrator_variant(t2d_pool* p, int32_t variant) {
    if (!p) return T2D_ERR_INVALID;
    if (variant < 0 || variant > 3) return fail(p, T2D_ERR_INVALID, "variant must be 0 (exact), 1 (fast), 2 (fast, kinematic steps iterated) or 3 (fast, resummed whatever the pool size)");
    p->integrator_variant = variant ? 1 : 0;
    p->kin_resum = variant != 2;
    p->kin_resum_forced = variant == 3;
    p->derived_interval = -1;   // (the resummation table travels with the interval's derived values)
    return T2D_OK;
}

int t2d_profile_enable(t2d_pool* p, int32_t on) {
    if (!p) return T2D_ERR_INVALID;
    T2D_HIP(p, hipSetDevice(p->device));
    if (on && !p->prof_events) {
        p->prof_events = new hipEvent_t[2 * t2d_pool::kMaxProfSteps];
        for (int i = 0; i < 2 * t2d_pool::kMaxProfSteps; ++i) T2D_HIP(p, hipEventCreate(&p->prof_events[i]));
    }
    p->profiling = on != 0;
    p->prof_count = 0;
    return T2D_OK;
}

int t2d_profile_read(t2d_pool* p, int32_t kernel_id, double* total_ms, int64_t* launches) {
    if (!p) return T2D_ERR_INVALID;
    if (!total_ms || !launches) return fail(p, T2D_ERR_INVALID, "null output");
    T2D_HIP(p, hipSetDevice(p->device));
    T2D_HIP(p, quiesce(p));
    double tot = 0.0;
    int64_t n = 0;
    for (int i = 0; i + 1 < p->prof_count; i += 2) {
        if (p->prof_kernel[i] != kernel_id) continue;
        float ms = 0.f;
        T2D_HIP(p, hipEventElapsedTime(&ms, p->prof_events[i], p->prof_events[i + 1]));
        tot += ms;
        ++n;
    }
    *total_ms = tot;
    *launches = n;
    return T2D_OK;
}

}  // extern "C"
