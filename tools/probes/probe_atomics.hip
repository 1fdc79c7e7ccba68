// Cost model of fire-and-forget global atomics at the END of a short kernel (development probe, gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_atomics.hip -o build/probe_atomics && build/probe_atomics
// Each of B one-wave workgroups issues ONE wave-level atomicMax (lane 0) to slot (blockIdx % C) * stride.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

__global__ void k_atomic(uint32_t* acc, int C, int stride, int per_lane) {
    const uint32_t v = blockIdx.x * 64 + threadIdx.x;
    if (per_lane) atomicMax(acc + (size_t)((blockIdx.x * 64 + threadIdx.x) % C) * stride, v);
    else if (threadIdx.x == 0) atomicMax(acc + (size_t)(blockIdx.x % C) * stride, v);
}
__global__ void k_store(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x;
}
__global__ void k_empty() {}

template <typename F>
float time_us(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ts;
    for (int i = 0; i < 25; ++i) {
        hipEventRecord(a, 0);
        launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (i >= 5) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    uint32_t* acc;
    hipMalloc(&acc, 4096 * 64 * 4);
    hipMemset(acc, 0, 4096 * 64 * 4);
    printf("empty kernel: %.1f us\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(128), dim3(64), 0, 0); }));
    for (int B : {128, 512, 2048, 8192}) {
        printf("B=%5d plain stores: %.1f us\n", B,
               time_us([&] { hipLaunchKernelGGL(k_store, dim3(B), dim3(64), 0, 0, acc); }));
        for (int C : {1, 64, 256, 2048})
            for (int stride : {1, 32})
                for (int per_lane : {0, 1})
                    printf("B=%5d C=%4d stride=%2d %s: %.1f us\n", B, C, stride, per_lane ? "per-lane" : "per-wave",
                           time_us([&] { hipLaunchKernelGGL(k_atomic, dim3(B), dim3(64), 0, 0, acc, C, stride, per_lane); }));
    }
    return 0;
}
