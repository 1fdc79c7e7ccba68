// probe: does a hand-written GELU (erf form) compiled by THIS hipcc reproduce ATen's GPU kernels bit for bit?
// (ATen: aten/src/ATen/native/cuda/ActivationGeluKernel.cu, compiled into torch with ITS ROCm's device library and contraction flags)
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" {
__global__ void gelu_fwd_k(const float* x, float* y, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float kAlpha = (float)0.70710678118654752440;
    float v = x[i];
    y[i] = v * 0.5f * (1.0f + erff(v * kAlpha));
}
// variant 0: no contraction; 1: fma(x, pdf, cdf); 2: also -0.5*x*x as written (same) but cdf via fma(0.5, erf, 0.5)
__global__ void gelu_bwd_k(const float* dy, const float* x, float* dx, long n, int variant) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float kBeta = (float)(1.12837916709551257390 * 0.70710678118654752440 * 0.5);
    const float kAlpha = (float)0.70710678118654752440;
    float v = x[i], g = dy[i];
    float e = erff(v * kAlpha);
    float cdf = variant == 2 ? fmaf(0.5f, e, 0.5f) : 0.5f * (1.0f + e);
    float pdf = expf(-0.5f * v * v) * kBeta;
    float s = variant >= 1 ? fmaf(v, pdf, cdf) : (cdf + v * pdf);
    dx[i] = g * s;
}
void gelu_fwd(const float* x, float* y, long n, void* stream) {
    hipLaunchKernelGGL(gelu_fwd_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
}
void gelu_bwd(const float* dy, const float* x, float* dx, long n, int variant, void* stream) {
    hipLaunchKernelGGL(gelu_bwd_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n, variant);
}
}
