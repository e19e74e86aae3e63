// stream_overlap.hip -- do kernels on G HIP streams of one process run side by side on this stack, and what does a
// dependent launch cost per stream when several streams are active?  (Why the closed loop of env groups is 50 us per
// step at 4 groups: profiles/r04_*; DESIGN.md "Env groups".)
//   hipcc --offload-arch=gfx950 -O2 scripts/stream_overlap.hip -o scripts/_build/stream_overlap
//   GPU_MAX_HW_QUEUES=8 scripts/_build/stream_overlap
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// every wave spins for `ticks` of the 100 MHz clock; `lds` bytes of dynamic LDS and a register-heavy body keep the
// occupancy at what the step kernel has (4 workgroups of 256 threads per CU)
__global__ __launch_bounds__(256, 4) void spin(long long ticks, float* sink) {
    extern __shared__ float s[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = threadIdx.x;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) { acc = acc * 1.0001f + 1.0f; __builtin_amdgcn_s_sleep(8); }
    if (acc == 12345.678f) sink[0] = acc + s[threadIdx.x];
}
__global__ void tiny(float* sink, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sink[i] = sink[i] * 0.5f + 1.0f;
}

int main() {
    float* sink;
    CHECK(hipMalloc(&sink, 4 << 20));
    CHECK(hipMemset(sink, 0, 4 << 20));
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    printf("{\"GPU_MAX_HW_QUEUES\": \"%s\", \"rows\": [\n", q ? q : "default");
    bool first = true;
    for (int G : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(G);
        for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int pattern = 0; pattern < 4; ++pattern) {
            // 0: K spin kernels of 10 us (1 wave) per stream; 1: (tiny, spin 10 us) pairs; 2: spin kernels the shape of a group's step
            // launch (1024 / G workgroups of 256 threads, 38 KB LDS, 10 us); 3: (tiny, step-shaped spin) pairs
            const int K = 100;
            const long long ticks = 1000;   // 10 us
            auto body = [&](int k) {
                for (int g = 0; g < G; ++g) {
                    if (pattern & 1) hipLaunchKernelGGL(tiny, dim3(256 / G), dim3(256), 0, st[g], sink + g * 65536, 65536 / G);
                    if (pattern < 2) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[g], ticks, sink);
                    else hipLaunchKernelGGL(spin, dim3(1024 / G), dim3(256), 38 * 1024, st[g], ticks, sink);
                }
            };
            for (int k = 0; k < 20; ++k) body(k);
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; ++k) body(k);
            auto t1 = std::chrono::steady_clock::now();
            CHECK(hipDeviceSynchronize());
            auto t2 = std::chrono::steady_clock::now();
            const double us = std::chrono::duration<double, std::micro>(t2 - t0).count() / K;
            const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / K;
            printf("%s  {\"streams\": %d, \"pattern\": %d, \"us_per_round\": %.2f, \"host_enqueue_us_per_round\": %.2f}", first ? "" : ",\n", G, pattern, us, enq);
            first = false;
        }
        for (auto& s : st) CHECK(hipStreamDestroy(s));
    }
    printf("\n],\n \"read\": \"a round = one 10-us kernel per stream (patterns 1, 3: preceded by a tiny kernel on the same stream); streams that run side by side: us_per_round ~ 10-12 whatever the stream count; serialised: ~ 10 x streams\"}\n");
    return 0;
}
