// libqsparse_hip.so -- the STE backward with the CALLER'S activation's backward on the way out (qs_ste_relu_bwd_args::act_x,
// ABI v26; ste_relu_bwd_kernel<..., DACT>).  A translation unit of its own: its kernel instantiations compile next to the others.
#include "qs_host_ew.h"

// called by qs_quant_ste_relu_bwd_v (api_quant_bwd.hip) when the descriptor names an act_x; arguments already copied
int qs_ste_act_bwd_impl(const qs_ste_relu_bwd_args& a) {
    const void *g = a.g, *g2 = a.g2, *x = a.act_x;
    void* gx = a.gx;
    const int gdt = a.gdt, xdt = a.xdt, g2dt = a.g2dt;
    if ((!g && !g2) || !x || !gx || a.act_x_kind != QS_DACT_GELU) return QS_ERR_ARG;
    if (!dt_ok(gdt) || !dt_ok(xdt) || !(gdt == QS_F32 || gdt == xdt)) return QS_ERR_DTYPE;
    if (g2 && (gdt != QS_F32 || (g2dt != QS_BF16 && g2dt != QS_F16))) return QS_ERR_DTYPE;
    if ((g && !aligned16(g)) || !aligned16(x) || !aligned16(gx) || (g2 && !aligned16(g2))) return QS_ERR_ALIGN;
    // the riders of the all-fp32 kernel form (BwdRiders, qs_elementwise.h) keep their conditions
    if (a.g3 && (!g2 || gdt != QS_F32 || xdt != QS_F32)) return QS_ERR_ARG;
    if (a.gx_image && (gdt != QS_F32 || xdt != QS_F32 || (a.gx_image_dt != QS_BF16 && a.gx_image_dt != QS_F16))) return QS_ERR_DTYPE;
    if ((a.g3 && !aligned16(a.g3)) || (a.gx_image && !aligned16(a.gx_image))) return QS_ERR_ALIGN;
    const BwdRiders rd{a.g3, a.gx_image, a.gx_image_dt};
    int st = check_param(a.step, a.nstep, a.C);
    if (st) return st;
    const bool ppc = a.nstep > 1;
    const uint8_t* chan_mask = a.chan_mask;
    EwPlan plan;
    st = plan_ew(a.outer, a.C, a.inner, ppc || chan_mask != nullptr, &plan, !ppc && aligned8(chan_mask));
    if (st) return st;
    if (plan.geo.numel == 0) return QS_OK;
    hipStream_t s = (hipStream_t)a.stream;
    SteBwdOp op{a.step, a.step_host, a.step_is_decimal, a.lo_mul, a.hi_mul, 0, chan_mask};
    const int grid = grid_for(plan.geo.ngroups, 1);
    constexpr bool NT = QS_EW_NT != 0;
    const ActSpec act{QS_ACT_NONE, 0.f, 0.f};
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        auto go = [&](auto G, auto G2) {
            constexpr int GD = decltype(G)::value, G2D = decltype(G2)::value;
            int cm = plan.cm;
            if (GD == QS_F32 && XD == QS_F32 && cm == CM_ELEM && plan.geo.inner % 4 == 0) cm = CM_ROW;   // 4 elements per lane
            switch (cm) {
                case CM_SCALAR:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_SCALAR, NT, false, false, G2D, QS_DACT_GELU>), dim3(grid), dim3(kBlock),
                                       0, s, op, plan.geo, (int)ppc, g, x, gx, act, g2, rd);
                    break;
                case CM_ROW:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ROW, NT, false, false, G2D, QS_DACT_GELU>), dim3(grid), dim3(kBlock),
                                       0, s, op, plan.geo, (int)ppc, g, x, gx, act, g2, rd);
                    break;
                case CM_LAST:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_LAST, NT, false, false, G2D, QS_DACT_GELU>), dim3(grid), dim3(kBlock),
                                       0, s, op, plan.geo, (int)ppc, g, x, gx, act, g2, rd);
                    break;
                default:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ELEM, NT, false, false, G2D, QS_DACT_GELU>), dim3(grid), dim3(kBlock),
                                       0, s, op, plan.geo, (int)ppc, g, x, gx, act, g2, rd);
                    break;
            }
            return launch_status();
        };
        if (g2) return g2dt == QS_BF16 ? go(IC<QS_F32>{}, IC<QS_BF16>{}) : go(IC<QS_F32>{}, IC<QS_F16>{});
        return gdt == QS_F32 ? go(IC<QS_F32>{}, IC<-1>{}) : go(X, IC<-1>{});
    });
}
