// t2d_loop.hip -- the closed loop around t2d_step, as the reference's callers run it (envs/parking.py:219-256 inside the
// tutorial's `action = agent.choose_action(obs); obs, reward, ... = env.step(action)` loop), kept on the device:
//     policy kernel (reads the state step k - 1 left behind, writes an [N, 2] action tensor)
//  -> t2d_step (reads that tensor in place: t2d_bind_actions_strided)
//  -> policy kernel ... , nothing synchronises with the host in between.
// Environments never interact (traffic/scenario_manager.py:52-61), so the envs are cut into G groups, each its own pool on
// its own stream: while one group's policy runs -- and while its step launch starts up or drains -- the other groups'
// step kernels fill the GPU.  That is how a closed-loop caller gets the overlap t2d_step_n gives an open-loop one.
//
// What is here: (1) a STAND-IN policy -- per-participant state feedback, a few flops, so that what is measured is the
// step path and not a network -- and (2) a runner that enqueues `n` iterations of (policy, step) per group in one host call,
// three ways: from the calling thread, from one host thread per group, or as one captured hipGraph per group.  Both are
// measurement / test helpers of the C ABI (`t2d_debug_*`): a real caller enqueues its own policy between t2d_step calls
// (tactics2d_amd/pipeline.py EnvGroups), the launch pattern is the same.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "t2d_pool.h"
#include "../../include/t2d_debug.h"
#include <thread>

// a spin-wait hint that exists on every host (the x86 pause instruction where there is one)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}


namespace t2d {
namespace {

// the action a participant takes from its own state: speed control towards v_target, steering that swings with the
// position (so that lanes are left, cars meet and episodes end, as with the bench's random actions).  Output layout: the
// reference's action_space order (steering, accel) -- envs/parking.py:130-139 -- one pair per participant.
__global__ __launch_bounds__(256) void feedback_policy_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ heading, const float* __restrict__ speed,
                                                              float2* __restrict__ out, int n, float v_target, float k_speed,
                                                              float k_steer) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = speed[i], h = heading[i];
    float accel = k_speed * (v_target - v);
    accel = fminf(fmaxf(accel, -3.0f), 2.0f);
    const float steer = k_steer * __sinf(0.05f * x[i] + 0.08f * y[i] + h);
    out[i] = make_float2(steer, accel);
}

}  // namespace
}  // namespace t2d

struct t2d_closed_loop {
    std::vector<t2d_pool*> pools;
    std::vector<hipStream_t> streams;
    std::vector<float*> act;
    int interval_ms = 100, launcher = 0, graph_steps = 0;
    float v_target = 12.0f, k_speed = 0.5f, k_steer = 0.04f;
    // launcher 1: one worker thread per group, parked on `go` between runs
    std::vector<std::thread> workers;
    // (a worker spins on `go` for ~2 ms after its last run -- back-to-back runs, warm-up then timed region, find it hot --
    // and sleeps on the condition variable otherwise)
    std::atomic<int> go{0}, done{0}, quit{0};
    std::atomic<int> steps_req{0};
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> rc;
    // launcher 2: one instantiated graph per group holding graph_steps iterations of (policy, step)
    std::vector<hipGraphExec_t> graphs;
    std::string err;
};

namespace {

hipError_t launch_policy(t2d_closed_loop* L, int g) {
    t2d_pool* p = L->pools[g];
    const int n = p->v.N;
    hipLaunchKernelGGL(t2d::feedback_policy_kernel, dim3((n + 255) / 256), dim3(256), 0, L->streams[g], p->v.x, p->v.y, p->v.heading,
                       p->v.speed, reinterpret_cast<float2*>(L->act[g]), n, L->v_target, L->k_speed, L->k_steer);
    return hipGetLastError();
}

int iterate(t2d_closed_loop* L, int g, int n_steps) {
    t2d_pool* p = L->pools[g];
    for (int k = 0; k < n_steps; ++k) {
        if (launch_policy(L, g) != hipSuccess) return T2D_ERR_HIP;
        p->v.overlapped = L->pools.size() > 1;   // (several groups in flight: the wave-priority rule of overlapping launches)
        const int rc = t2d_step(p, L->interval_ms, L->streams[g]);
        p->v.overlapped = 0;
        if (rc != T2D_OK) return rc;
    }
    return T2D_OK;
}

void worker_main(t2d_closed_loop* L, int g) {
    (void)hipSetDevice(L->pools[g]->device);
    int seen = 0;
    for (;;) {
        int cur;
        auto t0 = std::chrono::steady_clock::now();
        int polls = 0;
        while ((cur = L->go.load(std::memory_order_acquire)) == seen) {
            if (L->quit.load(std::memory_order_acquire)) return;
            cpu_relax();
            if ((++polls & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                std::unique_lock<std::mutex> lk(L->mu);
                L->cv.wait(lk, [&] { return L->go.load(std::memory_order_acquire) != seen || L->quit.load(std::memory_order_acquire); });
                t0 = std::chrono::steady_clock::now();
            }
        }
        seen = cur;
        L->rc[g] = iterate(L, g, L->steps_req.load(std::memory_order_relaxed));
        L->done.fetch_add(1, std::memory_order_release);
    }
}

}  // namespace

extern "C" {

int t2d_debug_feedback_policy(t2d_pool* p, float* act_out_dev, float v_target, float k_speed, float k_steer, void* hip_stream) {
    if (!p || !act_out_dev) return T2D_ERR_INVALID;
    const int n = p->v.N;
    hipLaunchKernelGGL(t2d::feedback_policy_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, p->v.x, p->v.y,
                       p->v.heading, p->v.speed, reinterpret_cast<float2*>(act_out_dev), n, v_target, k_speed, k_steer);
    if (hipGetLastError() != hipSuccess) {
        p->err = "feedback_policy_kernel launch failed";
        return T2D_ERR_HIP;
    }
    return T2D_OK;
}

int t2d_debug_closed_loop_create(t2d_pool* const* pools, void* const* hip_streams, float* const* act_out_dev, int32_t n_groups,
                                 int32_t interval_ms, int32_t launcher, int32_t graph_steps, t2d_closed_loop** out) {
    if (!pools || !hip_streams || !act_out_dev || !out || n_groups < 1 || interval_ms <= 0 || launcher < 0 || launcher > 2)
        return T2D_ERR_INVALID;
    if (launcher == 2 && (graph_steps < 1 || graph_steps > 4096)) return T2D_ERR_INVALID;
    auto L = std::make_unique<t2d_closed_loop>();
    L->interval_ms = interval_ms;
    L->launcher = launcher;
    L->graph_steps = graph_steps;
    for (int g = 0; g < n_groups; ++g) {
        if (!pools[g] || !act_out_dev[g]) return T2D_ERR_INVALID;
        L->pools.push_back(pools[g]);
        L->streams.push_back((hipStream_t)hip_streams[g]);
        L->act.push_back(act_out_dev[g]);
        // the policy's [N, 2] (steering, accel) tensor, read in place: accel = out + 1, steering = out, stride 2
        const int rc = t2d_bind_actions_strided(pools[g], act_out_dev[g] + 1, act_out_dev[g], 2);
        if (rc != T2D_OK) return rc;
    }
    L->rc.assign(n_groups, T2D_OK);
    if (launcher == 1) {
        for (int g = 0; g < n_groups; ++g) L->workers.emplace_back(worker_main, L.get(), g);
    }
    if (launcher == 2) {
        // one graph per group: graph_steps iterations of (policy, step) captured from the group's own stream.  A captured step
        // launch carries its record-ring slot in its arguments, so a replay writes the slots of the capture again: with
        // graph_steps a multiple of T2D_RECORD_RING (or callers that read T2D_F_STATUS / T2D_F_REWARD, not the ring) that is
        // what plain launches would do; the pool's step count is advanced per replay below.
        L->graphs.assign(n_groups, nullptr);
        for (int g = 0; g < n_groups; ++g) {
            t2d_pool* p = L->pools[g];
            (void)hipSetDevice(p->device);
            const long long count0 = p->step_count;
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(L->streams[g], hipStreamCaptureModeThreadLocal);
            int rc = T2D_OK;
            if (e == hipSuccess) rc = iterate(L.get(), g, graph_steps);
            if (e == hipSuccess) e = hipStreamEndCapture(L->streams[g], &graph);
            p->step_count = count0;   // (nothing ran yet)
            if (e == hipSuccess && rc == T2D_OK) e = hipGraphInstantiate(&L->graphs[g], graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (e != hipSuccess || rc != T2D_OK) {
                p->err = std::string("closed loop: graph capture failed: ") + (e != hipSuccess ? hipGetErrorString(e) : p->err.c_str());
                for (hipGraphExec_t ge : L->graphs)
                    if (ge) (void)hipGraphExecDestroy(ge);
                return e != hipSuccess ? T2D_ERR_HIP : rc;
            }
        }
    }
    *out = L.release();
    return T2D_OK;
}

int t2d_debug_closed_loop_run(t2d_closed_loop* L, int32_t n_steps) {
    if (!L || n_steps < 1) return T2D_ERR_INVALID;
    const int G = (int)L->pools.size();
    if (L->launcher == 0) {   // the calling thread: step by step round the groups, like a caller of t2d_step_groups would
        for (int k = 0; k < n_steps; ++k)
            for (int g = 0; g < G; ++g) {
                const int rc = iterate(L, g, 1);
                if (rc != T2D_OK) return rc;
            }
        return T2D_OK;
    }
    if (L->launcher == 1) {   // one host thread per group enqueues that group's whole run
        L->steps_req.store(n_steps, std::memory_order_relaxed);
        L->done.store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(L->mu);
            L->go.fetch_add(1, std::memory_order_release);
        }
        L->cv.notify_all();
        // (bounded: a worker whose t2d_step blocks for good -- a wedged device -- must not hang the caller as well)
        const auto t_wait = std::chrono::steady_clock::now();
        for (long polls = 0; L->done.load(std::memory_order_acquire) < G; ++polls) {
            cpu_relax();
            if ((polls & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(60)) {
                L->pools[0]->err = "closed loop: a group's host thread did not finish enqueuing within 60 s";
                return T2D_ERR_HIP;
            }
        }
        for (int g = 0; g < G; ++g)
            if (L->rc[g] != T2D_OK) return L->rc[g];
        return T2D_OK;
    }
    // graphs: whole replays, then the remainder from this thread
    int left = n_steps;
    while (left >= L->graph_steps) {
        for (int g = 0; g < G; ++g) {
            t2d_pool* p = L->pools[g];
            if (hipGraphLaunch(L->graphs[g], L->streams[g]) != hipSuccess) {
                p->err = "closed loop: hipGraphLaunch failed";
                return T2D_ERR_HIP;
            }
            p->step_count += L->graph_steps;
            t2d::pool_touch(p, L->streams[g]);
        }
        left -= L->graph_steps;
    }
    for (int k = 0; k < left; ++k)
        for (int g = 0; g < G; ++g) {
            const int rc = iterate(L, g, 1);
            if (rc != T2D_OK) return rc;
        }
    return T2D_OK;
}

int t2d_debug_closed_loop_destroy(t2d_closed_loop* L) {
    if (!L) return T2D_OK;
    {
        std::lock_guard<std::mutex> lk(L->mu);
        L->quit.store(1, std::memory_order_release);
    }
    L->cv.notify_all();
    for (std::thread& t : L->workers) t.join();
    for (hipGraphExec_t ge : L->graphs)
        if (ge) (void)hipGraphExecDestroy(ge);
    delete L;
    return T2D_OK;
}

}  // extern "C"
