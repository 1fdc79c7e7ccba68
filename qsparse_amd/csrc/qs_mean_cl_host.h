// Host side of the channels_last staged means and the fused last two stages (qs_mean_dim_cl / qs_mean_cl_w / qs_mean_last2):
// dispatch and launch configuration for ONE input dtype, included by api_mean_cl_f32.hip / _bf16.hip / _f16.hip; api_mean_cl.hip holds
// the C ABI entry points.  (As one translation unit this dispatch took 126 s of a clean build.)
#pragma once
#include "qs_host.h"
#include "qs_reduce.h"

// compile-time operand preparation of the one-lane-per-output kernels (mean_prep_t): 1 |x|, 2 |max(x, 0)|, 3 x, 0 run-time flags
static inline int prep_code_cl(int flags, const int32_t* l0_flag, const ActSpec& act) {
    if (l0_flag || (flags & QS_MEAN_L0)) return 0;
    if (flags == QS_MEAN_ABS) return 1;
    if (flags == (QS_MEAN_ABS | QS_MEAN_RELU) && act.kind == QS_ACT_RELU) return 2;
    return flags == 0 ? 3 : 0;
}



template <int XD>
static int qs_mean_dim_cl_impl(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags,
                   const int32_t* l0_flag, float* amax_part, qs_stream_t stream) {
    if (!x || !out || n < 1 || hw < 1 || C < 1) return QS_ERR_ARG;
    if (!dt_ok(xdt) || !(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    ActSpec act;
    if (mean_act_resolve(&flags, &act) != QS_OK) return QS_ERR_ARG;
    flags &= 0xff;
    int mode = (flags & QS_MEAN_L0) ? 0 : (flags == QS_MEAN_ABS ? 1 : (flags == (QS_MEAN_ABS | QS_MEAN_RELU) ? 2 :
                                           (flags == 0 ? 3 : 0)));
    // |act(x)| of a folded activation other than nn.ReLU: its kind as a compile-time mode (4 stays for kinds without one)
    if (mode == 2 && act.kind > QS_ACT_RELU) mode = act.kind == QS_ACT_HARDTANH ? 5 : (act.kind == QS_ACT_LEAKY ? 6 : 4);
    if (C % 8 != 0 || mode == 0 || (mode == 3 && amax_part) || !aligned16(x)) {
        // any channel count, the L0 variant, unaligned views: one lane per element of a sample, same summation order
        const int64_t total = hw * C;
        if ((total + kBlock - 1) / kBlock > 0x7fffffff) return QS_ERR_ARG;
        return [&]() {
            const IC<XD> X{};
            (void)X;
            if (xdt != XD) return (int)QS_ERR_DTYPE;
            const dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
            const int prep = prep_code_cl(flags, l0_flag, act);
            auto launch = [&](auto O, auto A) {
                auto go = [&](auto P) {
                    hipLaunchKernelGGL((mean_cl_generic_kernel<XD, decltype(O)::value, decltype(A)::value, decltype(P)::value>), grid,
                                       dim3(kBlock), 0, (hipStream_t)stream, x, out, n, hw, C, flags, l0_flag, (uint32_t*)amax_part, act);
                };
                if (prep == 1) go(IC<1>{});
                else if (prep == 2) go(IC<2>{});
                else if (prep == 3) go(IC<3>{});
                else go(IC<0>{});
            };
            if (odt == QS_F32) { if (amax_part) launch(IC<QS_F32>{}, std::true_type{}); else launch(IC<QS_F32>{}, std::false_type{}); }
            else { if (amax_part) launch(X, std::true_type{}); else launch(X, std::false_type{}); }
            return launch_status();
        }();
    }
    const int64_t main_groups = (hw / 4) * 4 * C / 8, tail_groups = hw * C / 8 - main_groups;   // ATen's split of H*W
    auto chunks_of = [](int64_t items) {               // full level-0 chunks of a cascade over `items` items
        const int lp = std::max(4, (items <= 1 ? 0 : 64 - __builtin_clzll((unsigned long long)(items - 1))) / 4);
        return items >> lp;
    };
    const int64_t nchunks = chunks_of(n), tail_slots = 4 * chunks_of(n / 4);
    // Few waves: ONE launch of the workgroup kernel -- rows shared by the R waves of a workgroup, main and tail positions
    // together, narrow waves when even that leaves CUs idle.  Many waves: the one-wave-per-512-columns kernel for the
    // main positions; the tail positions still take the workgroup kernel.
    const int64_t waves64 = (main_groups + 63) / 64;
    const bool big = waves64 >= 512;
    // measured on the activation shapes of a ResNet-50 step at batch 256 (tools/bench_stats.py --cl --b256; columns of
    // 64-lane waves the main positions would fill -> best waves per workgroup : lanes per wave):
    //   >= 784 -> 1 (the one-wave kernel);  392 and 196 -> 4 : 64 (R = 8 / 16 and narrower waves are slower: 22.5 vs 24-38 us on
    //   256x128x28x28 bf16);  98 and 48 -> 8 : 32 (256x256x14x14 bf16 22.3 -> 15.7 us, 256x512x7x7 17.4 -> 15.5 us)
    const int64_t g = big ? tail_groups : std::max(main_groups, tail_groups);
    const int64_t gwaves = (g + 63) / 64;
    int lanes = env_int("QS_CL_LANES", 0);
    if (lanes != 16 && lanes != 32 && lanes != 64) lanes = gwaves < 128 ? 32 : 64;
    const int64_t slots = std::max<int64_t>(big ? 0 : nchunks, tail_groups > 0 ? tail_slots : 0);
    int R = env_int("QS_MEAN_SPLIT", 0);
    if (R == 0) R = gwaves >= 512 ? 1 : (gwaves < 128 ? 8 : 4);
    R = R >= 16 ? 16 : (R >= 8 ? 8 : (R >= 4 ? 4 : (R >= 2 ? 2 : 1)));
    if (xdt == QS_F32 && R > 8) R = 8;                 // 16 fp32 rows in flight need > 128 VGPRs: 512-thread workgroups at most
    while (R > 1 && (R > slots || (size_t)(slots + R) * 8 * lanes * sizeof(float) > 63 * 1024)) R >>= 1;
    const size_t lds = (size_t)(slots + R) * 8 * lanes * sizeof(float);
    const bool wg_ok = slots >= 1 && lds <= 63 * 1024;
    // without the workgroup kernel (very long batches: the slot sums do not fit the LDS) both parts fall back to their
    // one-wave kernels
    const bool wg_main = wg_ok && !big && main_groups > 0;
    const bool wg_tail = wg_ok && tail_groups > 0;
    const int lanes1 = mean_lanes(main_groups > 0 ? main_groups : 1);
    const int blocks1 = (int)((main_groups + lanes1 - 1) / lanes1);
    const int xcd_wg = env_int("QS_CL_XCD_WG", 1);     // XCD-contiguous order of the main workgroups (0: linear)
    int wg_main_blocks = wg_main ? (int)((main_groups + lanes - 1) / lanes) : 0;
    if (xcd_wg) wg_main_blocks = (wg_main_blocks + 7) / 8 * 8;
    const int wg_tail_blocks = wg_tail ? (int)((tail_groups + lanes - 1) / lanes) : 0;
    const int tail_blocks1 = (int)((tail_groups + 63) / 64);
    hipStream_t s = (hipStream_t)stream;
    return [&]() {
            const IC<XD> X{};
            (void)X;
            if (xdt != XD) return (int)QS_ERR_DTYPE;
        auto run = [&](auto O) {
            constexpr int OD = decltype(O)::value;
            uint32_t* am = (uint32_t*)amax_part;
            auto launch = [&](auto M) {
                constexpr int kM = decltype(M)::value;
                if (!wg_main && main_groups > 0) {
                    // XCD-contiguous wave order (see the kernel); QS_CL_XCD=0 restores the linear order
                    const int xcd_per = env_int("QS_CL_XCD", 1) ? (blocks1 + 7) / 8 : 0;
                    hipLaunchKernelGGL((mean_cl_kernel<XD, OD, kM>), dim3(xcd_per ? 8 * xcd_per : blocks1), dim3(64), 0, s, x, out, n,
                                       hw, C, am, lanes1, main_groups, act, xcd_per);
                }
                if (wg_main || wg_tail) {
                    auto wg = [&](auto RR) {
                        constexpr int kR = decltype(RR)::value;
                        hipLaunchKernelGGL((mean_cl_wg_kernel<XD, OD, kR, kM>), dim3(wg_main_blocks + wg_tail_blocks), dim3(64 * kR),
                                           lds, s, x, out, n, hw, C, am, lanes, main_groups, wg_main_blocks, tail_groups,
                                           (int)slots, act, xcd_wg);
                    };
                    if constexpr (XD != QS_F32) {
                        if (R == 16) wg(IC<16>{});
                    }
                    if (R == 8) wg(IC<8>{});
                    else if (R == 4) wg(IC<4>{});
                    else if (R == 2) wg(IC<2>{});
                    else if (R == 1) wg(IC<1>{});
                }
                if (!wg_tail && tail_groups > 0)
                    hipLaunchKernelGGL((mean_cl_tail_kernel<XD, OD, kM>), dim3(tail_blocks1), dim3(64), 0, s, x, out, n, hw, C,
                                       am, main_groups, tail_groups, act);
            };
            if (mode == 1) launch(IC<1>{});
            else if (mode == 2) launch(IC<2>{});
            else if (mode == 4) launch(IC<4>{});
            else if (mode == 5) launch(IC<5>{});
            else if (mode == 6) launch(IC<6>{});
            else launch(IC<3>{});
            return launch_status();
        };
        return (odt == QS_F32) ? run(IC<QS_F32>{}) : run(X);
    }();
}

template <int XD>
static int qs_mean_cl_w_impl(const void* x, void* out, int64_t N, int64_t H, int64_t W, int64_t C, int xdt, int odt, int flags,
                 const int32_t* l0_flag, qs_stream_t stream) {
    if (!x || !out || N < 1 || H < 1 || W < 1 || C < 1) return QS_ERR_ARG;
    if (!dt_ok(xdt) || !(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    ActSpec act;
    if (mean_act_resolve(&flags, &act) != QS_OK) return QS_ERR_ARG;
    flags &= 0xff;
    const int64_t total = N * H * C;
    if ((total + kBlock - 1) / kBlock > 0x7fffffff) return QS_ERR_ARG;
    return [&]() {
            const IC<XD> X{};
            (void)X;
            if (xdt != XD) return (int)QS_ERR_DTYPE;
        const dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
        if (odt == QS_F32)
            hipLaunchKernelGGL((mean_cl_w_kernel<XD, QS_F32>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, out, N * H, H, W, C, flags,
                               l0_flag, act);
        else
            hipLaunchKernelGGL((mean_cl_w_kernel<XD, XD>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, out, N * H, H, W, C, flags,
                               l0_flag, act);
        return launch_status();
    }();
}

template <int XD>
static int qs_mean_last2_impl(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt, const float* amax_part,
                  float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream) {
    if (!x || !out || pre < 1 || H < 1 || W < 1) return QS_ERR_ARG;
    if ((amax_part || (record && absmax_out)) && (!absmax_out || absmax_stride < 1)) return QS_ERR_ARG;
    if (!dt_ok(xdt) || !(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    const size_t lds = (size_t)(H * W + W + 8) * sizeof(float);
    if (lds > kLast2MaxLds || pre > 0x7fffffff) return QS_ERR_ARG;
    return [&]() {
            const IC<XD> X{};
            (void)X;
            if (xdt != XD) return (int)QS_ERR_DTYPE;
        if (odt == QS_F32)
            hipLaunchKernelGGL((mean_last2_kernel<XD, QS_F32>), dim3((int)pre), dim3(kBlock), lds, (hipStream_t)stream, x,
                               out, (int)H, (int)W, (const uint32_t*)amax_part, (uint32_t*)absmax_out, absmax_stride, record);
        else
            hipLaunchKernelGGL((mean_last2_kernel<XD, XD>), dim3((int)pre), dim3(kBlock), lds, (hipStream_t)stream, x, out,
                               (int)H, (int)W, (const uint32_t*)amax_part, (uint32_t*)absmax_out, absmax_stride, record);
        return launch_status();
    }();
}


#define QS_MEAN_CL_DTYPE_UNIT(SUFFIX, DT) \
    __attribute__((visibility("hidden"))) int qs_mean_dim_cl_##SUFFIX(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, float* amax_part, qs_stream_t stream) { return qs_mean_dim_cl_impl<DT>(x, out, n, hw, C, xdt, odt, flags, l0_flag, amax_part, stream); } \
    __attribute__((visibility("hidden"))) int qs_mean_cl_w_##SUFFIX(const void* x, void* out, int64_t N, int64_t H, int64_t W, int64_t C, int xdt, int odt, int flags, const int32_t* l0_flag, qs_stream_t stream) { return qs_mean_cl_w_impl<DT>(x, out, N, H, W, C, xdt, odt, flags, l0_flag, stream); } \
    __attribute__((visibility("hidden"))) int qs_mean_last2_##SUFFIX(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt, const float* amax_part, float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream) { return qs_mean_last2_impl<DT>(x, out, pre, H, W, xdt, odt, amax_part, absmax_out, absmax_stride, record, stream); }
