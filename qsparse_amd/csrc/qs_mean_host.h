// Host side of the NCHW staged means (qs_mean_dim / qs_mean_dim_split / qs_mean_strided): dispatch and launch configuration for ONE
// input dtype, included by api_mean_f32.hip / api_mean_bf16.hip / api_mean_f16.hip; api_mean.hip holds the C ABI entry points.
#pragma once
#include "qs_host.h"
#include "qs_reduce.h"

// compile-time operand preparation of the one-lane-per-output kernels (mean_prep_t): 1 |x|, 2 |max(x, 0)|, 3 x, 0 run-time flags
static inline int prep_code(int flags, const int32_t* l0_flag, const ActSpec& act) {
    if (l0_flag || (flags & QS_MEAN_L0)) return 0;
    if (flags == QS_MEAN_ABS) return 1;
    if (flags == (QS_MEAN_ABS | QS_MEAN_RELU) && act.kind == QS_ACT_RELU) return 2;
    return flags == 0 ? 3 : 0;
}

// mr_cols < 0: ATen's rule for a contiguous [pre, n, post] tensor; >= 0 (post > 1): columns [0, mr_cols) of every slice in cascade
// order, the others in row-sum order
// ONE input dtype per instantiation: the three dtypes compile in three translation units (api_mean_f32 / _bf16 / _f16.hip), side by
// side -- as one unit this dispatch was the critical path of a clean build (209 s of ~3.5 min)
template <int XD>
static int mean_dim_impl(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int xdt, int odt, int flags,
                         const int32_t* l0_flag, float* absmax_out, int64_t absmax_stride, int64_t chan_div, int64_t C,
                         int64_t mr_cols, int percol_part, qs_stream_t stream) {
    if (!x || !out || pre < 1 || n < 1 || post < 1) return QS_ERR_ARG;
    if (!dt_ok(xdt) || !dt_ok(odt)) return QS_ERR_DTYPE;
    if (!(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    if (absmax_out && (chan_div < 1 || C < 1 || absmax_stride < 1)) return QS_ERR_ARG;
    // the folded activation: nn.ReLU, or the descriptor whose handle rides in the flags' upper bits (QS_MEAN_ACT)
    ActSpec act;
    if (mean_act_resolve(&flags, &act) != QS_OK) return QS_ERR_ARG;
    const bool general_act = act.kind > QS_ACT_RELU;      // (takes the generic modes: every flag tested per element)
    flags &= 0xff;
    const int64_t as = absmax_out ? absmax_stride : 1;
    hipStream_t s = (hipStream_t)stream;
    int64_t vcols = 0;
    const bool ragged_absmax = absmax_out && (chan_div % 8 != 0);
    // percol_part (internal: qs_token_stats): `absmax_out` is a [pre][post] array that receives the abs-max of every COLUMN (the
    // PERCOL form of the vector kernel, no atomics); the caller guarantees post % 32 == 0 (no generic tail) and the |x| / |max(x, 0)|
    // flags
    const bool percol = absmax_out && (percol_part & 1) != 0;
    // bit 1 (internal: the second stage of qs_token_stats, a [T][C] matrix of a few thousand columns): one lane per output for
    // 2-byte inputs as well -- 197 x 3072 bf16: 7 us against 10 (one-wave kernel) / 15 (rows split eight ways); 1024 x 4096: 21 against 42
    const bool prefer_generic = (percol_part & 2) != 0;
    if (percol && (post % 32 != 0 || l0_flag || !(flags == QS_MEAN_ABS || (flags == (QS_MEAN_ABS | QS_MEAN_RELU) && !general_act)) || mr_cols >= 0))
        return QS_ERR_ARG;
    if (percol && !(post >= 64 && aligned16(x) && aligned16(out) && aligned16(absmax_out))) return QS_ERR_ARG;
    if (post >= 64 && post % 8 == 0 && aligned16(x) && aligned16(out) && (!ragged_absmax || chan_div >= 8 || percol))
        vcols = mr_cols >= 0 ? (mr_cols / 8) * 8 : (post / 32) * 32;
    // float32 without the abs-max rider: one lane per output (mean_generic_kernel, 4-byte loads, four times the waves) is as fast as
    // the 8-columns-per-lane kernels on the largest tensors and faster below (256 x 150528: 25 against 35 us, 1024 x 65536: 56 against
    // 100; only very short columns, n < 32, lose) -- QS_MEAN_F32_GENERIC=0 restores the vector kernels
    // (only with a compile-time operand preparation: |x|, |max(x, 0)| or x -- a folded nn.ReLU6 / nn.LeakyReLU has its own modes in
    //  the vector kernels and none here)
    if (xdt == QS_F32 && !absmax_out && n >= 32 && prep_code(flags, l0_flag, act) != 0 && env_int("QS_MEAN_F32_GENERIC", 1)) vcols = 0;
    if (prefer_generic && !absmax_out && n >= 32 && prep_code(flags, l0_flag, act) != 0 && env_int("QS_TOKEN_STAGE2_GENERIC", 1)) vcols = 0;
    uint32_t* am = (uint32_t*)absmax_out;
    // rows are split over R waves per workgroup when there are too few column groups to fill the chip
    const int lp = std::max(4, (n <= 1 ? 0 : 64 - __builtin_clzll((unsigned long long)(n - 1))) / 4);
    const int64_t nchunks = n >> lp;
    int R = 1;
    if (vcols > 0) {
        const int64_t waves = (pre * (vcols / 8) + 63) / 64;
        const int want = env_int("QS_MEAN_SPLIT", 0);
        if (want > 0) R = want;
        else if (waves < 128 && nchunks >= 8 && nchunks <= kMaxSplitChunks) R = 8;   // with narrow waves, see below
        // (the next two with the abs-max rider only: without it the one-wave kernel with 32 rows in flight is the faster one from 128
        //  waves on -- 256 x 768 x 14 x 14 bf16: 20.0 us against 30.8 split four ways, 256 x 512 x 14 x 14: 18.9 against 27.0;
        //  with the rider it is the other way round, 31.4 against 21.9)
        else if (am && waves < 256 && nchunks >= 2 && nchunks <= kMaxSplitChunks) R = 4;   // measured: tools/bench_stats.py
        else if (am && waves < 512 && xdt != QS_F32 && nchunks >= 8 && nchunks <= kMaxSplitChunks) R = 4;   // long columns of 2-byte values
        while (R > 1 && R > nchunks) R >>= 1;
        if (nchunks > kMaxSplitChunks || nchunks < 2) R = 1;
        if (percol) R = 1;          // (the one-wave kernel: the split kernel has no per-column form)
        else if (ragged_absmax && R == 1) R = (nchunks >= 2 && nchunks <= kMaxSplitChunks) ? 2 : 0;   // only the split kernel tracks two channels
        if (R == 0) vcols = 0;
    }
    if (xdt != XD) return QS_ERR_DTYPE;
    {
        const IC<XD> X{};
        auto run = [&](auto O) {
            constexpr int OD = decltype(O)::value;
            if (vcols > 0) {
                const int64_t total = pre * (vcols / 8);
                // tiny tensors (fewer than 128 waves of column groups: 14x14 / 7x7 maps of a few hundred channels) are
                // spread over more workgroups by narrow waves, as in qs_mean_dim_cl (QS_MEAN_NARROW=0: off)
                const bool narrow = R > 1 && (total + 63) / 64 < 128 && env_int("QS_MEAN_NARROW", 1) != 0;
                const int lanes = narrow ? 32 : mean_lanes(total);
                const int blocks = (int)((total + lanes - 1) / lanes);
                const uint32_t Cc = (uint32_t)(C > 0 ? C : 1);
                const size_t lds = (size_t)nchunks * 8 * 64 * sizeof(float);
                if (R == 1) {
                    // (a folded nn.Hardtanh / nn.ReLU6 / nn.LeakyReLU with |.|: its own compile-time mode, 7 / 8 -- through the run-time
                    //  mode 0 the statistics of such a site took 377 us where nn.ReLU's take 80, 256 x 256 x 56 x 56 bf16)
                    const int act_mode = act.kind == QS_ACT_HARDTANH ? 7 : (act.kind == QS_ACT_LEAKY ? 8 : 0);
                    const int mode = l0_flag ? 0 : (flags == QS_MEAN_ABS ? 1 : (flags == (QS_MEAN_ABS | QS_MEAN_RELU) ? (general_act ? act_mode : 2) :
                                                    (flags == 0 && !am ? 3 : 0)));
                    // few waves per CU: keep more rows in flight per wave instead (latency-, not bandwidth-bound)
                    // (2-byte inputs only: 32 fp32 rows of 8 columns do not fit the register file)
                    int depth = env_int("QS_MEAN_DEPTH", 0);
                    if (depth == 0) depth = (blocks < 4 * 256 && n >= 32) ? 32 : QS_MEAN_ROWS_IN_FLIGHT;
                    if (XD == QS_F32) depth = QS_MEAN_ROWS_IN_FLIGHT;
                    auto launch = [&](auto D, auto M) {
                        hipLaunchKernelGGL((mean_outer_vec_kernel<XD, OD, decltype(D)::value, decltype(M)::value>), dim3(blocks),
                                           dim3(64), 0, s, x, out, pre, n, post, vcols, flags, l0_flag, am, as, chan_div, Cc, lanes, act);
                    };
                    auto by_mode = [&](auto D) {
                        if (percol) {
                            if (mode == 1)
                                hipLaunchKernelGGL((mean_outer_vec_kernel<XD, OD, decltype(D)::value, 1, true>), dim3(blocks), dim3(64), 0, s, x,
                                                   out, pre, n, post, vcols, flags, l0_flag, am, as, chan_div, Cc, lanes, act);
                            else
                                hipLaunchKernelGGL((mean_outer_vec_kernel<XD, OD, decltype(D)::value, 2, true>), dim3(blocks), dim3(64), 0, s, x,
                                                   out, pre, n, post, vcols, flags, l0_flag, am, as, chan_div, Cc, lanes, act);
                            return;
                        }
                        if (mode == 3) launch(D, IC<3>{});
                        else if (mode == 1) launch(D, IC<1>{});
                        else if (mode == 2) launch(D, IC<2>{});
                        else if (mode == 7) launch(D, IC<7>{});
                        else if (mode == 8) launch(D, IC<8>{});
                        else launch(D, IC<0>{});
                    };
                    if constexpr (XD != QS_F32) {
                        if (depth >= 32) by_mode(IC<32>{});
                        else by_mode(IC<QS_MEAN_ROWS_IN_FLIGHT>{});
                    } else {
                        by_mode(IC<QS_MEAN_ROWS_IN_FLIGHT>{});
                    }
                }
                else {
                    const int64_t cd = chan_div > 0 ? chan_div : 1;
                    const bool rag = am && cd % 8 != 0;         // a lane's 8 columns may straddle two channels
                    const int smode = (l0_flag || !am) ? 0 : (flags == QS_MEAN_ABS ? 1 :
                                      (flags == (QS_MEAN_ABS | QS_MEAN_RELU) ? (act.kind == QS_ACT_HARDTANH ? 7 : (act.kind == QS_ACT_LEAKY ? 8 :
                                                                                (general_act ? 0 : 2))) : 0));
                    auto launch = [&](auto RR, auto M, auto RG) {
                        constexpr int kR = decltype(RR)::value;
                        hipLaunchKernelGGL((mean_outer_split_kernel<XD, OD, kR, decltype(M)::value, decltype(RG)::value>), dim3(blocks),
                                           dim3(64 * kR), lds, s, x, out, pre, n, post, vcols, flags, l0_flag, am, as, cd, Cc, lanes, act);
                    };
                    auto by_mode = [&](auto RR) {
                        if (smode == 1 && rag) launch(RR, IC<1>{}, std::true_type{});
                        else if (smode == 2 && rag) launch(RR, IC<2>{}, std::true_type{});
                        else if (smode == 7 && rag) launch(RR, IC<7>{}, std::true_type{});
                        else if (smode == 8 && rag) launch(RR, IC<8>{}, std::true_type{});
                        else if (smode == 1) launch(RR, IC<1>{}, std::false_type{});
                        else if (smode == 2) launch(RR, IC<2>{}, std::false_type{});
                        else if (smode == 7) launch(RR, IC<7>{}, std::false_type{});
                        else if (smode == 8) launch(RR, IC<8>{}, std::false_type{});
                        else launch(RR, IC<0>{}, std::false_type{});
                    };
                    if (R == 2) by_mode(IC<2>{});
                    else if (R == 4) by_mode(IC<4>{});
                    else by_mode(IC<8>{});
                }
            }
            if (post == 1 && mr_cols < 0 && n >= 64 && !am && (pre + 1) / 2 <= 0x7fffffff && env_int("QS_MEAN_INNER_WAVE", 1)) {
                // long rows reduced along their own direction: half a wave per row (QS_MEAN_INNER_WAVE=0: one lane per row)
                hipLaunchKernelGGL((mean_inner_wave_kernel<XD, OD>), dim3((unsigned)((pre + 1) / 2)), dim3(64), 0, s, x, out, pre, n,
                                   flags, l0_flag, act);
            } else if (vcols < post) {
                const int64_t total = pre * (post - vcols);
                const dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
                auto launch = [&](auto A, auto P) {
                    hipLaunchKernelGGL((mean_generic_kernel<XD, OD, decltype(A)::value, decltype(P)::value>), grid, dim3(kBlock), 0, s, x, out,
                                       pre, n, post, vcols, flags, l0_flag, am, as, chan_div > 0 ? chan_div : 1, (uint32_t)(C > 0 ? C : 1), act,
                                       mr_cols);
                };
                const int prep = prep_code(flags, l0_flag, act);
                auto by_prep = [&](auto A) {
                    if (prep == 1) launch(A, IC<1>{});
                    else if (prep == 2) launch(A, IC<2>{});
                    else if (prep == 3) launch(A, IC<3>{});
                    else launch(A, IC<0>{});
                };
                if (am) by_prep(std::true_type{});
                else by_prep(std::false_type{});
            }
            return launch_status();
        };
        return (odt == QS_F32) ? run(IC<QS_F32>{}) : run(X);
    }
}

template <int XD>
static int mean_strided_launch(const void* x, void* out, int64_t total, const StridedPlan& p, int odt, int flags, const int32_t* l0_flag,
                               const ActSpec& act, qs_stream_t stream) {
    const dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
    const int prep = prep_code(flags, l0_flag, act);
    auto launch = [&](auto O) {
        auto go = [&](auto P) {
            hipLaunchKernelGGL((mean_strided_kernel<XD, decltype(O)::value, decltype(P)::value>), grid, dim3(kBlock), 0,
                               (hipStream_t)stream, x, out, total, p, flags, l0_flag, act);
        };
        if (prep == 1) go(IC<1>{});
        else if (prep == 2) go(IC<2>{});
        else if (prep == 3) go(IC<3>{});
        else go(IC<0>{});
    };
    if (odt == QS_F32) launch(IC<QS_F32>{});
    else launch(IC<XD>{});
    return launch_status();
}

// the per-dtype entry points of this library's own units (not part of the C ABI: hidden visibility)
#define QS_MEAN_DTYPE_UNIT(SUFFIX, DT)                                                                                                  \
    __attribute__((visibility("hidden"))) int qs_mean_dim_##SUFFIX(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int xdt, \
                                                                   int odt, int flags, const int32_t* l0_flag, float* absmax_out,         \
                                                                   int64_t absmax_stride, int64_t chan_div, int64_t C, int64_t mr_cols,   \
                                                                   int percol_part, qs_stream_t stream) {                                 \
        return mean_dim_impl<DT>(x, out, pre, n, post, xdt, odt, flags, l0_flag, absmax_out, absmax_stride, chan_div, C, mr_cols,         \
                                 percol_part, stream);                                                                                    \
    }                                                                                                                                     \
    __attribute__((visibility("hidden"))) int qs_mean_strided_##SUFFIX(const void* x, void* out, int64_t total, const StridedPlan* p,     \
                                                                       int odt, int flags, const int32_t* l0_flag, const ActSpec* act,    \
                                                                       qs_stream_t stream) {                                              \
        return mean_strided_launch<DT>(x, out, total, *p, odt, flags, l0_flag, *act, stream);                                             \
    }

