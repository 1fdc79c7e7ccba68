// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), statistics of channels_last activations (qs_mean_dim_cl, qs_mean_cl_w) and the
// fused last two stages (qs_mean_last2): kernels in qs_reduce.h, the NCHW stages in api_mean.hip.
// Host side: argument checks, geometry, launch configuration.  No allocation, no synchronisation.
#include "qs_host.h"

// (the kernels are instantiated per input dtype in api_mean_cl_f32 / _bf16 / _f16.hip, qs_mean_cl_host.h)

int qs_mean_dim_cl_f32(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, float* amax_part, qs_stream_t stream);
int qs_mean_dim_cl_bf16(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, float* amax_part, qs_stream_t stream);
int qs_mean_dim_cl_f16(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, float* amax_part, qs_stream_t stream);
int qs_mean_cl_w_f32(const void* x, void* out, int64_t N, int64_t H, int64_t W, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, qs_stream_t stream);
int qs_mean_cl_w_bf16(const void* x, void* out, int64_t N, int64_t H, int64_t W, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, qs_stream_t stream);
int qs_mean_cl_w_f16(const void* x, void* out, int64_t N, int64_t H, int64_t W, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, qs_stream_t stream);
int qs_mean_last2_f32(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt, const float* amax_part, float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream);
int qs_mean_last2_bf16(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt, const float* amax_part, float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream);
int qs_mean_last2_f16(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt, const float* amax_part, float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream);

extern "C" {

int qs_mean_dim_cl(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, float* amax_part, qs_stream_t stream) {
    switch (xdt) {
        case QS_BF16: return qs_mean_dim_cl_bf16(x, out, n, hw, C, xdt, odt, flags, l0_flag, amax_part, stream);
        case QS_F16: return qs_mean_dim_cl_f16(x, out, n, hw, C, xdt, odt, flags, l0_flag, amax_part, stream);
        default: return qs_mean_dim_cl_f32(x, out, n, hw, C, xdt, odt, flags, l0_flag, amax_part, stream);      // (an unknown dtype is refused by the unit's own checks, in their usual order)
    }
}

int qs_mean_cl_w(const void* x, void* out, int64_t N, int64_t H, int64_t W, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, qs_stream_t stream) {
    switch (xdt) {
        case QS_BF16: return qs_mean_cl_w_bf16(x, out, N, H, W, C, xdt, odt, flags, l0_flag, stream);
        case QS_F16: return qs_mean_cl_w_f16(x, out, N, H, W, C, xdt, odt, flags, l0_flag, stream);
        default: return qs_mean_cl_w_f32(x, out, N, H, W, C, xdt, odt, flags, l0_flag, stream);      // (an unknown dtype is refused by the unit's own checks, in their usual order)
    }
}

int qs_mean_last2(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt, const float* amax_part, float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream) {
    switch (xdt) {
        case QS_BF16: return qs_mean_last2_bf16(x, out, pre, H, W, xdt, odt, amax_part, absmax_out, absmax_stride, record, stream);
        case QS_F16: return qs_mean_last2_f16(x, out, pre, H, W, xdt, odt, amax_part, absmax_out, absmax_stride, record, stream);
        default: return qs_mean_last2_f32(x, out, pre, H, W, xdt, odt, amax_part, absmax_out, absmax_stride, record, stream);      // (an unknown dtype is refused by the unit's own checks, in their usual order)
    }
}

}  // extern "C"
