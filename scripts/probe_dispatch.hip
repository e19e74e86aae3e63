// probe_dispatch.hip -- how the hardware hands out the workgroups of a 2-D grid (the assumption behind t2d_step_n's chained
// launch): is workgroup (x, y) placed on XCD (y * gridDim.x + x) mod 8, and do the workgroups of one XCD start in the order
// of their linear ids?   hipcc --offload-arch=gfx950 -O2 scripts/probe_dispatch.hip -o gpurun_out/probe_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned long long* out, int spin_ticks) {
    extern __shared__ unsigned char lds[];   // 40 KB: four workgroups per CU, like the metric step launch
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    if (threadIdx.x == 0) lds[0] = 1;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        const size_t lin = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        out[3 * lin] = t0;
        out[3 * lin + 1] = __builtin_amdgcn_s_memrealtime();
        out[3 * lin + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(63508);   // XCC_ID
    }
}

int main() {
    const int nx = 1024, ny = 6, spin = 2000;   // 20 us per workgroup
    unsigned long long* d;
    std::vector<unsigned long long> h(3 * (size_t)nx * ny);
    hipMalloc(&d, h.size() * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nx, ny), dim3(256), 40 * 1024, 0, d, spin);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    int xcc_match = 0, inversions = 0, n = nx * ny;
    unsigned long long tmin = ~0ull;
    for (int i = 0; i < n; ++i) tmin = std::min(tmin, h[3 * i]);
    long long last_start[8];
    for (int k = 0; k < 8; ++k) last_start[k] = -1;
    long long worst = 0;
    for (int i = 0; i < n; ++i) {
        const int xcc = (int)(h[3 * i + 2] & 15);
        xcc_match += xcc == i % 8;
        const long long st = (long long)(h[3 * i] - tmin);
        if (st + 0 < last_start[xcc]) { ++inversions; worst = std::max(worst, last_start[xcc] - st); }
        last_start[xcc] = std::max(last_start[xcc], st);
    }
    printf("workgroups %d: XCC == linear id mod 8 for %d; start-order inversions within an XCD: %d (worst %lld ticks of 10 ns)\n",
           n, xcc_match, inversions, worst);
    for (int y = 0; y < ny; ++y) {
        long long a = 1ll << 60, b = 0, e = 0;
        for (int x = 0; x < nx; ++x) {
            const size_t i = (size_t)y * nx + x;
            a = std::min(a, (long long)(h[3 * i] - tmin)); b = std::max(b, (long long)(h[3 * i] - tmin));
            e = std::max(e, (long long)(h[3 * i + 1] - tmin));
        }
        printf("  row y=%d: first start %lld, last start %lld, last end %lld (x 10 ns)\n", y, a, b, e);
    }
    return 0;
}
