// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), statistics: the staged means of squeeze_tensor_to_shape in ATen's
// summation order (qs_reduce.h).
// Host side: argument checks, geometry, launch configuration.  No allocation, no synchronisation: every entry
// point only enqueues work on the caller's stream.
// (this unit: the entry points of the NCHW stages and the strided first stage; their kernels are instantiated per input dtype in
//  api_mean_f32 / _bf16 / _f16.hip (qs_mean_host.h); api_mean_cl.hip holds the channels_last stages and the fused last two)
#include "qs_host.h"
#include "qs_reduce.h"

#define QS_MEAN_DTYPE_DECL(SUFFIX)                                                                                                     \
    int qs_mean_dim_##SUFFIX(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int xdt, int odt, int flags,                \
                             const int32_t* l0_flag, float* absmax_out, int64_t absmax_stride, int64_t chan_div, int64_t C,              \
                             int64_t mr_cols, int percol_part, qs_stream_t stream);                                                      \
    int qs_mean_strided_##SUFFIX(const void* x, void* out, int64_t total, const StridedPlan* p, int odt, int flags,                      \
                                 const int32_t* l0_flag, const ActSpec* act, qs_stream_t stream);
QS_MEAN_DTYPE_DECL(f32)
QS_MEAN_DTYPE_DECL(bf16)
QS_MEAN_DTYPE_DECL(f16)

// mr_cols < 0: ATen's rule for a contiguous [pre, n, post] tensor; >= 0 (post > 1): columns [0, mr_cols) of every slice in cascade
// order, the others in row-sum order
static int mean_dim_impl(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int xdt, int odt, int flags,
                         const int32_t* l0_flag, float* absmax_out, int64_t absmax_stride, int64_t chan_div, int64_t C,
                         int64_t mr_cols, qs_stream_t stream, int percol_part = 0) {
    switch (xdt) {
        case QS_F32: return qs_mean_dim_f32(x, out, pre, n, post, xdt, odt, flags, l0_flag, absmax_out, absmax_stride, chan_div, C, mr_cols, percol_part, stream);
        case QS_BF16: return qs_mean_dim_bf16(x, out, pre, n, post, xdt, odt, flags, l0_flag, absmax_out, absmax_stride, chan_div, C, mr_cols, percol_part, stream);
        case QS_F16: return qs_mean_dim_f16(x, out, pre, n, post, xdt, odt, flags, l0_flag, absmax_out, absmax_stride, chan_div, C, mr_cols, percol_part, stream);
    }
    return (x && out && pre >= 1 && n >= 1 && post >= 1) ? QS_ERR_DTYPE : QS_ERR_ARG;
}


extern "C" {

int qs_mean_dim(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int xdt, int odt, int flags,
                const int32_t* l0_flag, float* absmax_out, int64_t absmax_stride, int64_t chan_div, int64_t C,
                qs_stream_t stream) {
    return mean_dim_impl(x, out, pre, n, post, xdt, odt, flags, l0_flag, absmax_out, absmax_stride, chan_div, C, -1, stream);
}

int qs_mean_dim_split(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int64_t mr_cols, int xdt, int odt, int flags,
                      const int32_t* l0_flag, qs_stream_t stream) {
    if (post < 2 || mr_cols < 0 || mr_cols > post) return QS_ERR_ARG;
    return mean_dim_impl(x, out, pre, n, post, xdt, odt, flags, l0_flag, nullptr, 1, 1, 1, mr_cols, stream);
}

int qs_token_stats(const void* x, void* stage, void* stage_mean, float* amax_part, float* chan_absmax, int64_t absmax_stride, int64_t N,
                   int64_t T, int64_t C, int xdt, int flags, qs_stream_t stream) {
    if (!x || !stage || !stage_mean || N < 1 || T < 1 || C < 1) return QS_ERR_ARG;
    if (chan_absmax && absmax_stride < 1) return QS_ERR_ARG;
    const int64_t post = T * C;
    {       // (a folded identity is no activation: mean_act_resolve)
        ActSpec probe;
        int low = flags;
        if (mean_act_resolve(&low, &probe) != QS_OK) return QS_ERR_ARG;
        if (probe.kind == QS_ACT_NONE) flags = low;
    }
    const int f = flags & 0xff;
    // the per-column form serves |x| and |max(x, 0)| of nn.ReLU on rows that are whole 32-column blocks (no generic tail); everything
    // else -- other folded activations, odd widths, unaligned views -- takes the atomics rider of qs_mean_dim (same results)
    const bool relu_only = !(flags & QS_MEAN_RELU) || (flags >> 8) <= 1;
    const bool part_ok = chan_absmax && amax_part && post % 32 == 0 && post >= 64 && (f == QS_MEAN_ABS || (f == (QS_MEAN_ABS | QS_MEAN_RELU) && relu_only)) &&
                         aligned16(x) && aligned16(stage) && aligned16(amax_part);
    int st;
    if (part_ok)
        st = mean_dim_impl(x, stage, 1, N, post, xdt, xdt, flags, nullptr, amax_part, 1, 1, C, -1, stream, 1);
    else
        st = mean_dim_impl(x, stage, 1, N, post, xdt, xdt, flags, nullptr, chan_absmax, chan_absmax ? absmax_stride : 1, 1, C, -1, stream);
    if (st) return st;
    if (T > 1) {
        st = mean_dim_impl(stage, stage_mean, 1, T, C, xdt, xdt, 0, nullptr, nullptr, 1, 1, C, -1, stream, 2);
        if (st) return st;
    } else {
        st = hip_status(hipMemcpyAsync(stage_mean, stage, (size_t)C * (xdt == QS_F32 ? 4 : 2), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        if (st) return st;
    }
    if (part_ok) {
        const dim3 grid((unsigned)((C + 63) / 64), (unsigned)((T + 31) / 32));
        hipLaunchKernelGGL(token_amax_fold_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t*)amax_part, T, C,
                           (uint32_t*)chan_absmax, absmax_stride);
        return launch_status();
    }
    return QS_OK;
}

int qs_mean_strided(const void* x, void* out, int64_t n, int64_t stride, int nkept, const int64_t* kept_size,
                    const int64_t* kept_in_stride, const int64_t* kept_out_stride, int order, int split_dim, int64_t split, int xdt,
                    int odt, int flags, const int32_t* l0_flag, qs_stream_t stream) {
    if (!x || !out || n < 1 || stride < 0 || nkept < 0 || nkept > kStridedMaxKept || order < 0 || order > 2) return QS_ERR_ARG;
    if (nkept > 0 && (!kept_size || !kept_in_stride || !kept_out_stride)) return QS_ERR_ARG;
    if (order == 2 && (split_dim < 0 || split_dim >= nkept || split < 0)) return QS_ERR_ARG;
    if (order == 0 && (stride != 1 || n < 8)) return QS_ERR_ARG;      // ATen's vectorised inner loop needs both
    if (!dt_ok(xdt) || !(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    ActSpec act;
    if (mean_act_resolve(&flags, &act) != QS_OK) return QS_ERR_ARG;
    flags &= 0xff;
    StridedPlan p{};
    p.n = n; p.stride = stride; p.nkept = nkept; p.order = order; p.split_dim = order == 2 ? split_dim : -1; p.split = split;
    int64_t total = 1;
    for (int k = 0; k < nkept; ++k) {
        if (kept_size[k] < 1 || kept_in_stride[k] < 0 || kept_out_stride[k] < 0) return QS_ERR_ARG;
        p.size[k] = kept_size[k]; p.in_stride[k] = kept_in_stride[k]; p.out_stride[k] = kept_out_stride[k];
        if (total > (int64_t)0x7fffffff * kBlock / kept_size[k]) return QS_ERR_ARG;
        total *= kept_size[k];
    }
    switch (xdt) {
        case QS_F32: return qs_mean_strided_f32(x, out, total, &p, odt, flags, l0_flag, &act, stream);
        case QS_BF16: return qs_mean_strided_bf16(x, out, total, &p, odt, flags, l0_flag, &act, stream);
        case QS_F16: return qs_mean_strided_f16(x, out, total, &p, odt, flags, l0_flag, &act, stream);
    }
    return QS_ERR_DTYPE;
}

}  // extern "C"
