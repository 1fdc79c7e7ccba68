// Multi-tensor weight path: abs-max, running-scale update and quantization of MANY small fp32 tensors (the conv /
// linear weights of a converted network) in three launches instead of three per tensor.  Per tensor these are 3-5 us
// kernels over a few thousand to a few million elements; a ResNet-50 has 54 of them, i.e. ~160 launches per step.
// The tensor list travels by value in the kernel arguments (<= kMultiMax tensors per launch, the host chunks longer
// lists); `block0` is the exclusive prefix sum of the workgroups each tensor gets, a workgroup finds its tensor by
// binary search.  Arithmetic is that of reduce_all_kernel / scale_update_kernel / ew_kernel<ScalerFwdOp|DecimalFwdOp>.
#pragma once
#include "qs_elementwise.h"
#include "qs_reduce.h"

namespace qs {

constexpr int kMultiMax = 48;

struct MultiTensors {           // 48 * (8 + 8 + 8 + 8 + 4 + 4 + 4) + 8 = 2120 bytes of kernel arguments
    const float* x[kMultiMax];
    float* y[kMultiMax];        // quantized output (multi_quant_kernel only)
    float* scale[kMultiMax];    // one-element running scale (QuantizeLayer.weight)
    int64_t numel[kMultiMax];
    int32_t block0[kMultiMax + 1];
    int32_t lo[kMultiMax], hi[kMultiMax];   // code range of the opt-in saturation (multi_quant_kernel; lo > hi: none)
    int32_t n;
};

struct MultiUpdate {            // 48 * (8 + 8 + 8 + 8 + 8 + 8 + 4 + 4) + 8 = 2696 bytes
    uint32_t* amax[kMultiMax];  // one-element abs-max accumulators (max-accumulated, zero between steps)
    float* scale[kMultiMax];
    float* backup[kMultiMax];   // nullable: receives the scale this update replaces (for a caller that may have to undo it)
    float* decimal[kMultiMax];  // nullable: receives rint(log2(1/scale)) for the decimal quantizer
    int64_t* t_dev[kMultiMax];  // nullable: device-resident running-mean counter, read INSTEAD of t and incremented
    int32_t* bump[kMultiMax];   // nullable: the layer's step counter, incremented
    float t[kMultiMax];
    float denom[kMultiMax];     // 2^(bits-1)
    int32_t n;
};

__device__ __forceinline__ int multi_find(const int32_t* block0, int n, int b) {
    int lo = 0, hi = n - 1;     // largest i with block0[i] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block0[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// abs-max of every tensor; a tensor's workgroups stride over its 8-element groups, one atomic per workgroup
static __global__ __launch_bounds__(kBlock) void multi_absmax_kernel(MultiTensors a, MultiUpdate u) {
    const int i = multi_find(a.block0, a.n, blockIdx.x);
    const int nb = a.block0[i + 1] - a.block0[i], b = blockIdx.x - a.block0[i];
    const float* x = a.x[i];
    const int64_t numel = a.numel[i], ngroups = numel / 8;
    RedAcc<QS_F32, false> acc;
    for (int64_t g = (int64_t)b * kBlock + threadIdx.x; g < ngroups; g += (int64_t)nb * kBlock) {
        float v[8];
        unpack8<QS_F32>(load8_raw<QS_F32, false>(x, g), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.add(v[j]);
    }
    if (b == 0 && ngroups * 8 + threadIdx.x < numel) acc.add(x[ngroups * 8 + threadIdx.x]);
    __shared__ uint32_t smx[kBlock / 64];
    acc.wave_reduce();
    if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = acc.mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = smx[0];
        for (int w = 1; w < kBlock / 64; ++w) m = smx[w] > m ? smx[w] : m;
        atomicMax(u.amax[i], m);
    }
}

// one thread per tensor: the running mean of scale_update_kernel, the decimal of decimal_from_scale_kernel, counters
static __global__ void multi_scale_update_kernel(MultiUpdate u) {
    const int i = threadIdx.x;
    if (i >= u.n) return;
    float t = u.t[i];
    if (u.t_dev[i]) t = (float)*u.t_dev[i];
    const float nw = __uint_as_float(*u.amax[i]) / u.denom[i];                 // fp32 weights: no dtype rounding
    const float old = *u.scale[i];
    if (u.backup[i]) *u.backup[i] = old;
    const float s = (t == 0.0f) ? nw : (t * old + nw) / (t + 1.0f);           // quantize.py:344-347
    *u.scale[i] = s;
    *u.amax[i] = 0u;
    if (u.decimal[i]) {
        float r = 1.0f / s;
        if (r == __builtin_inff() || r == -__builtin_inff()) r = 1.0f;          // nan_to_num(posinf=1, neginf=1)
        if (r != r) r = 0.0f;
        *u.decimal[i] = rintf(log2f(r));
    }
    if (u.t_dev[i]) atomicAdd((unsigned long long*)u.t_dev[i], 1ull);
    if (u.bump[i]) atomicAdd(u.bump[i], 1);
}

// y = Q(x) for every tensor: exact grid, 8 elements per lane; DECIMAL selects the truncating power-of-two quantizer
// (its per-tensor parameter is then the decimal, not the scale)
template <bool DECIMAL>
__global__ __launch_bounds__(kBlock) void multi_quant_kernel(MultiTensors a) {
    const int i = multi_find(a.block0, a.n, blockIdx.x);
    const int64_t g = (int64_t)(blockIdx.x - a.block0[i]) * kBlock + threadIdx.x;
    const float* x = a.x[i];
    float* y = a.y[i];
    const int64_t numel = a.numel[i], ngroups = numel / 8;
    int32_t code;
    if constexpr (DECIMAL) {
        const DecimalFwdOp<QS_F32> op{a.scale[i], 0.0f, nullptr, a.lo[i] <= a.hi[i], a.lo[i], a.hi[i], 0};
        const auto p = op.channel(0);
        if (g < ngroups) {
            float v[8];
            unpack8<QS_F32>(load8_raw<QS_F32, false>(x, g), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], p, code);
            store8<QS_F32, false>(y, g, v);
        }
        if (g == 0)
            for (int64_t e = ngroups * 8; e < numel; ++e) y[e] = op.apply(x[e], p, code);
    } else {
        const ScalerFwdOp<QS_F32> op{a.scale[i], 0.0f, nullptr, a.lo[i] <= a.hi[i], a.lo[i], a.hi[i], 0};
        const auto p = op.channel(0);
        if (g < ngroups) {
            float v[8];
            unpack8<QS_F32>(load8_raw<QS_F32, false>(x, g), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], p, code);
            store8<QS_F32, false>(y, g, v);
        }
        if (g == 0)
            for (int64_t e = ngroups * 8; e < numel; ++e) y[e] = op.apply(x[e], p, code);
    }
}

// gx = clamp(g, lo_mul * s, hi_mul * s) for every tensor (SteBwdOp's arithmetic, reference quantize.py:66-77, 120-131): the
// STE backward of MANY weight quantizers in one launch -- gradients of a group of layers handed over together
struct MultiSte {               // 48 * (8 + 8 + 8 + 8 + 4 + 4 + 4) + 8 = 2120 bytes
    const float* g[kMultiMax];
    float* gx[kMultiMax];
    const float* step[kMultiMax];   // one-element scale, or decimal with DECIMAL
    int64_t numel[kMultiMax];
    int32_t block0[kMultiMax + 1];
    float lo_mul[kMultiMax];
    float hi_mul[kMultiMax];
    int32_t n;
};

template <bool DECIMAL>
__global__ __launch_bounds__(kBlock) void multi_ste_kernel(MultiSte a) {
    const int i = multi_find(a.block0, a.n, blockIdx.x);
    const int64_t g0 = (int64_t)(blockIdx.x - a.block0[i]) * kBlock + threadIdx.x;
    const float* g = a.g[i];
    float* gx = a.gx[i];
    const int64_t numel = a.numel[i], ngroups = numel / 8;
    const SteBwdOp op{a.step[i], 0.0f, DECIMAL ? 1 : 0, a.lo_mul[i], a.hi_mul[i], 0, nullptr};
    const auto p = op.channel(0);
    int32_t code;
    if (g0 < ngroups) {
        float v[8];
        unpack8<QS_F32>(load8_raw<QS_F32, false>(g, g0), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], p, code);
        store8<QS_F32, false>(gx, g0, v);
    }
    if (g0 == 0)
        for (int64_t e = ngroups * 8; e < numel; ++e) gx[e] = op.apply(g[e], p, code);
}

}  // namespace qs
