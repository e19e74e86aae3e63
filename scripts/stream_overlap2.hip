// stream_overlap2.hip -- how many kernels of different HIP streams run side by side here?  S streams x creation mode
// (default / per-stream priorities / CU masks), one 1-wave 10-us spin kernel per stream and round.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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
int main() {
    float* sink;
    CHECK(hipMalloc(&sink, 4096));
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("{\"GPU_MAX_HW_QUEUES\": \"%s\", \"priority_range\": [%d, %d], \"rows\": [\n", q ? q : "default", lo, hi);
    bool first = true;
    for (int mode = 0; mode < 4; ++mode)   // 0 default flags, 1 priorities alternating hi / lo, 2 CU masks (disjoint eighths), 3 blocking streams (hipStreamDefault)
        for (int S : {1, 2, 3, 4, 6, 8}) {
            std::vector<hipStream_t> st(S);
            for (int i = 0; i < S; ++i) {
                if (mode == 0) CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
                else if (mode == 1) CHECK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, (i & 1) ? lo : hi));
                else if (mode == 2) {
                    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    mask[i % 8] = 0xffffffffu;   // 32 of the 256 CUs
                    CHECK(hipExtStreamCreateWithCUMask(&st[i], 8, mask));
                } else CHECK(hipStreamCreateWithFlags(&st[i], hipStreamDefault));
            }
            const int K = 200;
            auto round = [&]() { for (int i = 0; i < S; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 1000LL, sink); };
            for (int k = 0; k < 20; ++k) round();
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; ++k) round();
            CHECK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
            printf("%s  {\"mode\": %d, \"streams\": %d, \"us_per_round\": %.2f, \"kernels_side_by_side\": %.2f}", first ? "" : ",\n", mode, S, us, S * 10.9 / us);
            first = false;
            for (auto& s : st) CHECK(hipStreamDestroy(s));
        }
    printf("\n]}\n");
    return 0;
}
