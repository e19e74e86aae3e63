// stream_overlap2.hip -- do HIP streams of DIFFERENT PRIORITIES run side by side where streams of one priority do not?
//   scripts/_build/stream_overlap2 MODE S     (one configuration per process: run each under `timeout`)
// MODE 0: S streams of default priority; 1: priorities alternating highest / lowest; 2: cycling highest / default / lowest.
// One 1-wave 10-us spin kernel per stream and round, 200 rounds: prints us per round (10.9 = all side by side).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(64) void spin(long long ticks, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) sink[0] = 1.0f;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, S = argc > 2 ? atoi(argv[2]) : 4;
    float* sink;
    CHECK(hipMalloc(&sink, 4096));
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least priority (largest number), hi = greatest
    std::vector<hipStream_t> st(S);
    for (int i = 0; i < S; ++i) {
        int pr = 0;
        if (mode == 1) pr = (i & 1) ? lo : hi;
        if (mode == 2) pr = (i % 3 == 0) ? hi : (i % 3 == 1 ? 0 : lo);
        CHECK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, pr));
    }
    const int K = 200;
    auto round = [&]() { for (int i = 0; i < S; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 1000LL, sink); };
    for (int k = 0; k < 20; ++k) round();
    CHECK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; ++k) round();
    CHECK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    printf("{\"GPU_MAX_HW_QUEUES\": \"%s\", \"priority_range\": [%d, %d], \"mode\": %d, \"streams\": %d, \"us_per_round\": %.2f, \"kernels_side_by_side\": %.2f}\n",
           q ? q : "default", lo, hi, mode, S, us, S * 10.9 / us);
    fflush(stdout);
    return 0;
}
