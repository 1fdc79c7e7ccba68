// Launch plan of the element-wise kernels (qs_elementwise.h): geometry + channel mode, kernel choice.
#pragma once
#include "qs_host.h"
#include "qs_elementwise.h"

namespace {

// geometry + channel mode of an element-wise launch over [outer, C, inner]
struct EwPlan {
    EwGeom geo;
    int cm;
};

// `last_ok`: the caller's parameter is tensor-wise and its channel mask (if any) is 8-byte aligned, so that a tensor
// whose channel dim is the innermost one (channels_last activations: outer = N*H*W, inner = 1) may take CM_LAST
int plan_ew(int64_t outer, int64_t C, int64_t inner, bool per_channel, EwPlan* plan, bool last_ok = false) {
    if (outer < 0 || C < 1 || inner < 1) return QS_ERR_ARG;
    const int64_t numel = outer * C * inner;
    if (numel / 8 >= ((int64_t)1 << 32) || C >= ((int64_t)1 << 32) || inner >= ((int64_t)1 << 32)) return QS_ERR_ARG;
    plan->geo.numel = numel;
    plan->geo.ngroups = numel / 8;
    plan->geo.C = (uint32_t)C;
    plan->geo.inner = (uint32_t)inner;
    plan->geo.groups_per_row = (uint32_t)(inner / 8);
    plan->geo.reverse = ew_reverse() ? 1u : 0u;
    if (!per_channel) plan->cm = CM_SCALAR;
    else if (inner % 8 == 0) plan->cm = CM_ROW;
    else if (last_ok && inner == 1 && C % 8 == 0) plan->cm = CM_LAST;
    else plan->cm = CM_ELEM;
    return QS_OK;
}

inline int ew_widen() {
    static int v = env_int("QS_EW_WIDEN", 2);
    return v;
}

// `elide`: skip the loads of lanes whose elements are all pruned (qs_elementwise.h, "Mask-aware traffic elision");
// only meaningful for ops that carry a channel mask, in the per-channel modes
template <typename Op, int XDT, int YDT, bool ELIDE>
int launch_ew_impl(const Op& op, const EwPlan& plan, bool param_per_channel, const void* x, void* y, int32_t* codes,
                   hipStream_t s) {
    constexpr bool NT = QS_EW_NT != 0;
    constexpr int U = ELIDE ? QS_EW_UNROLL_ELIDE : QS_EW_UNROLL;
    if constexpr (YDT == QS_F32) {   // QS_EW_WIDEN: 0 off, 1 two-byte inputs only, 2 (default) fp32 inputs as well
        // (its lanes take 4 elements at a time, so rows of 4k elements -- 14x14 maps -- keep one channel per lane as well)
        const int cm_w = (plan.cm == CM_ELEM && plan.geo.inner % 4 == 0) ? CM_ROW : plan.cm;
        if (ew_widen() >= (XDT == QS_F32 ? 2 : 1) && !codes && cm_w != CM_ELEM) {
            const int64_t waves = (plan.geo.ngroups * 8 + 511) / 512;
            const int gridw = (int)std::max<int64_t>(1, (waves + kWidenBlock / 64 - 1) / (kWidenBlock / 64));   // < 8 elements: tail only
            if (cm_w == CM_SCALAR) {
                if constexpr (!ELIDE)
                    hipLaunchKernelGGL((ew_widen_kernel<Op, XDT, CM_SCALAR, false, NT>), dim3(gridw), dim3(kWidenBlock), 0, s, op,
                                       plan.geo, x, (float*)y);
            } else if (cm_w == CM_LAST)
                hipLaunchKernelGGL((ew_widen_kernel<Op, XDT, CM_LAST, false, NT, ELIDE>), dim3(gridw), dim3(kWidenBlock), 0, s, op,
                                   plan.geo, x, (float*)y);
            else if (param_per_channel)
                hipLaunchKernelGGL((ew_widen_kernel<Op, XDT, CM_ROW, true, NT, ELIDE>), dim3(gridw), dim3(kWidenBlock), 0, s, op,
                                   plan.geo, x, (float*)y);
            else
                hipLaunchKernelGGL((ew_widen_kernel<Op, XDT, CM_ROW, false, NT, ELIDE>), dim3(gridw), dim3(kWidenBlock), 0, s, op,
                                   plan.geo, x, (float*)y);
            return launch_status();
        }
    }
    const int grid = grid_for(plan.geo.ngroups, U);
    switch (plan.cm) {
        case CM_SCALAR:
            if constexpr (!ELIDE) {
                constexpr int US = QS_EW_UNROLL_SCALAR;
                hipLaunchKernelGGL((ew_kernel<Op, XDT, YDT, CM_SCALAR, false, NT, US>), dim3(grid_for(plan.geo.ngroups, US)),
                                   dim3(kBlock), 0, s, op, plan.geo, x, y, codes);
            }
            break;
        case CM_ROW: {
            constexpr int UR = ELIDE ? QS_EW_UNROLL_ELIDE : QS_EW_UNROLL_ROW;
            const int grid_r = grid_for(plan.geo.ngroups, UR);
            if (param_per_channel)
                hipLaunchKernelGGL((ew_kernel<Op, XDT, YDT, CM_ROW, true, NT, UR, ELIDE>), dim3(grid_r), dim3(kBlock), 0, s, op,
                                   plan.geo, x, y, codes);
            else
                hipLaunchKernelGGL((ew_kernel<Op, XDT, YDT, CM_ROW, false, NT, UR, ELIDE>), dim3(grid_r), dim3(kBlock), 0, s, op,
                                   plan.geo, x, y, codes);
            break;
        }
        case CM_LAST:
            hipLaunchKernelGGL((ew_kernel<Op, XDT, YDT, CM_LAST, false, NT, U, ELIDE>), dim3(grid), dim3(kBlock), 0, s, op,
                               plan.geo, x, y, codes);
            break;
        default:
            if (param_per_channel)
                hipLaunchKernelGGL((ew_kernel<Op, XDT, YDT, CM_ELEM, true, NT, 1, ELIDE>), dim3(grid_for(plan.geo.ngroups, 1)),
                                   dim3(kBlock), 0, s, op, plan.geo, x, y, codes);
            else
                hipLaunchKernelGGL((ew_kernel<Op, XDT, YDT, CM_ELEM, false, NT, 1, ELIDE>), dim3(grid_for(plan.geo.ngroups, 1)),
                                   dim3(kBlock), 0, s, op, plan.geo, x, y, codes);
            break;
    }
    return launch_status();
}

template <typename Op, int XDT, int YDT>
int launch_ew(const Op& op, const EwPlan& plan, bool param_per_channel, const void* x, void* y, int32_t* codes,
              hipStream_t s, bool elide = false) {
    if (plan.geo.numel == 0) return QS_OK;
    if constexpr (Op::kHasMask && !OpGate<Op>::value) {     // (a gate-recording op loads every element: no elision)
        if (elide && plan.cm != CM_SCALAR && op.mask_ptr() != nullptr)
            return launch_ew_impl<Op, XDT, YDT, true>(op, plan, param_per_channel, x, y, codes, s);
    }
    return launch_ew_impl<Op, XDT, YDT, false>(op, plan, param_per_channel, x, y, codes, s);
}

int check_param(const float* p, int64_t nparam, int64_t C) {
    if (p == nullptr) return nparam == 1 ? QS_OK : QS_ERR_ARG;
    if (nparam != 1 && nparam != C) return QS_ERR_ARG;
    return QS_OK;
}

}  // namespace

namespace {
// (shared by the forward units api_quant_fwd.hip / api_quant_fwd2.hip)
// the image of a quantizer's float32 output is written by the gate-recording widening kernels only (ew_widen_kernel<GateOp<..>>):
// float32 output, no codes, a gate bitmap, and a geometry those kernels serve (launch_ew_impl's own conditions)
// (the same kernels write relu(x) back: xback_out)
inline bool widen_route_ok(int64_t outer, int64_t C, int64_t inner, bool ppc, const uint8_t* chan_mask, const int32_t* codes,
                    const uint8_t* gate_out, int xdt, int ydt) {
    if (!gate_out || codes || ydt != QS_F32) return false;
    EwPlan plan;
    if (plan_ew(outer, C, inner, ppc || chan_mask != nullptr, &plan, !ppc && aligned8(chan_mask)) != QS_OK) return false;
    const int cm_w = (plan.cm == CM_ELEM && plan.geo.inner % 4 == 0) ? CM_ROW : plan.cm;
    return ew_widen() >= (xdt == QS_F32 ? 2 : 1) && cm_w != CM_ELEM;
}
inline bool image_route_ok(int64_t outer, int64_t C, int64_t inner, bool ppc, const uint8_t* chan_mask, const int32_t* codes,
                    const uint8_t* gate_out, int xdt, int ydt, int imgdt, const void* image_out) {
    if ((imgdt != QS_BF16 && imgdt != QS_F16) || !aligned16(image_out)) return false;
    return widen_route_ok(outer, C, inner, ppc, chan_mask, codes, gate_out, xdt, ydt);
}
}  // namespace
