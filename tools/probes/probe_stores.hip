// Write-only and sparse-read streaming ceilings on MI355X (development probe, gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_stores.hip -o build/probe_stores && build/probe_stores
// Question behind it: the elided apply kernels (qs_elementwise.h) degenerate into store-only kernels on pruned rows;
// the bf16 backward then wrote at 3.4-3.9 TB/s while the fp32 forward wrote at 6-6.7 TB/s.  What decides that:
// bytes per wave in flight, contiguity per workgroup, workgroup count, or the non-temporal hint?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each lane issues S 16-byte stores.  LAYOUT 0: wave-contiguous (lane l writes [l*16 + s*1024) of its wave's S KiB),
// LAYOUT 1: workgroup-contiguous (store s covers [s*BS*16, (s+1)*BS*16) of the workgroup's span),
// LAYOUT 2: grid-strided (store s lands a whole grid away, like ew_kernel's UNROLL)
template <int S, int LAYOUT, bool NT, int BS>
__global__ __launch_bounds__(BS) void k_store(u32x4* __restrict__ out, size_t n16) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    const size_t wg = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        size_t i;
        if (LAYOUT == 0) i = (wg * (BS / 64) + wave) * (size_t)(64 * S) + s * 64 + lane;
        else if (LAYOUT == 1) i = wg * (size_t)(BS * S) + s * BS + threadIdx.x;
        else i = ((size_t)s * gridDim.x + wg) * BS + threadIdx.x;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(z, out + i);
            else out[i] = z;
        }
    }
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ts;
    for (int i = 0; i < 5; ++i) launch();
    for (int i = 0; i < 21; ++i) {
        hipEventRecord(a, 0);
        launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <int S, int LAYOUT, bool NT, int BS>
void run(u32x4* buf, size_t bytes, const char* what) {
    const size_t n16 = bytes / 16;
    const size_t per_wg = (size_t)BS * S;
    const int grid = (int)((n16 + per_wg - 1) / per_wg);
    const float ms = time_ms([&] { hipLaunchKernelGGL((k_store<S, LAYOUT, NT, BS>), dim3(grid), dim3(BS), 0, 0, buf, n16); });
    printf("%-22s S=%d BS=%4d nt=%d  %7.1f MB  grid %7d  %.4f ms  %7.1f GB/s\n", what, S, BS, (int)NT, bytes / 1e6, grid, ms,
           bytes / ms / 1e6);
}

int main() {
    const size_t big = (size_t)205520896 * 4;   // the fp32 output of the headline tensor
    u32x4* buf;
    hipMalloc(&buf, big);
    hipMemset(buf, 0, big);
    for (size_t bytes : {big / 2, big}) {
        run<1, 0, true, 256>(buf, bytes, "wave-contig");
        run<1, 0, false, 256>(buf, bytes, "wave-contig");
        run<2, 0, true, 256>(buf, bytes, "wave-contig");
        run<2, 0, false, 256>(buf, bytes, "wave-contig");
        run<4, 0, true, 256>(buf, bytes, "wave-contig");
        run<8, 0, true, 256>(buf, bytes, "wave-contig");
        run<2, 1, true, 256>(buf, bytes, "wg-contig");
        run<4, 1, true, 256>(buf, bytes, "wg-contig");
        run<8, 1, true, 256>(buf, bytes, "wg-contig");
        run<2, 2, true, 256>(buf, bytes, "grid-strided");
        run<4, 2, true, 256>(buf, bytes, "grid-strided");
        run<1, 0, true, 512>(buf, bytes, "wave-contig");
        run<1, 0, true, 1024>(buf, bytes, "wave-contig");
        run<2, 1, true, 1024>(buf, bytes, "wg-contig");
        run<4, 1, true, 1024>(buf, bytes, "wg-contig");
        run<1, 0, true, 128>(buf, bytes, "wave-contig");
        run<2, 0, true, 128>(buf, bytes, "wave-contig");
        run<2, 1, true, 128>(buf, bytes, "wg-contig");
        run<2, 2, true, 128>(buf, bytes, "grid-strided");
        run<4, 0, true, 128>(buf, bytes, "wave-contig");
        run<2, 0, true, 64>(buf, bytes, "wave-contig");
        run<1, 0, true, 64>(buf, bytes, "wave-contig");
        run<4, 0, true, 64>(buf, bytes, "wave-contig");
        printf("\n");
    }
    // hipMemsetAsync as the runtime's own fill
    for (size_t bytes : {big / 2, big}) {
        const float ms = time_ms([&] { hipMemsetAsync(buf, 0, bytes, 0); });
        printf("hipMemsetAsync %7.1f MB  %.4f ms  %7.1f GB/s\n", bytes / 1e6, ms, bytes / ms / 1e6);
    }
    hipFree(buf);
    return 0;
}
