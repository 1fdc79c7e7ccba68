// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), STE backward, folded-ReLU backward and mask apply (qs_elementwise.h).
// Host side: argument checks, geometry, launch configuration.  No allocation, no synchronisation: every entry
// point only enqueues work on the caller's stream.
#include "qs_host_ew.h"

extern "C" {

int qs_quant_ste_bwd(const void* g, void* gx, const float* step, int64_t nstep, float step_host, int step_is_decimal,
                     float lo_mul, float hi_mul, int passthrough, const uint8_t* chan_mask, int64_t outer, int64_t C,
                     int64_t inner, int gdt, int gxdt, int elide_masked, qs_stream_t stream) {
    if (!g || !gx) return QS_ERR_ARG;
    if (!dt_ok(gdt) || !dt_ok(gxdt)) return QS_ERR_DTYPE;
    if (!(gdt == QS_F32 || gdt == gxdt)) return QS_ERR_DTYPE;
    if (!aligned16(g) || !aligned16(gx)) return QS_ERR_ALIGN;
    int st = check_param(step, nstep, C);
    if (st) return st;
    const bool ppc = nstep > 1;
    EwPlan plan;
    st = plan_ew(outer, C, inner, ppc || chan_mask != nullptr, &plan, !ppc && aligned8(chan_mask));
    if (st) return st;
    hipStream_t s = (hipStream_t)stream;
    SteBwdOp op{step, step_host, step_is_decimal, lo_mul, hi_mul, passthrough, chan_mask};
    return with_dtype(gxdt, [&](auto GX) {
        constexpr int GXD = decltype(GX)::value;
        if (gdt == QS_F32) return launch_ew<SteBwdOp, QS_F32, GXD>(op, plan, ppc, g, gx, nullptr, s, elide_masked != 0);
        return launch_ew<SteBwdOp, GXD, GXD>(op, plan, ppc, g, gx, nullptr, s, elide_masked != 0);
    });
}

static int ste_relu_bwd_impl(const qs_ste_relu_bwd_args& a) {
    const void *g = a.g, *x = a.x, *g2 = a.g2;
    const uint8_t* gate = a.gate;
    void* gx = a.gx;
    const int gdt = a.gdt, xdt = a.xdt, g2dt = a.g2dt;
    if ((!g && !g2) || (!x && !gate) || !gx) return QS_ERR_ARG;
    ActSpec act;
    if (qs_act_resolve(a.act > 0 ? a.act : 1, &act) != QS_OK) return QS_ERR_ARG;
    if (!dt_ok(gdt) || !dt_ok(xdt) || !(gdt == QS_F32 || gdt == xdt)) return QS_ERR_DTYPE;
    if (g2 && (!gate || gdt != QS_F32 || (g2dt != QS_BF16 && g2dt != QS_F16))) return QS_ERR_DTYPE;
    if ((g && !aligned16(g)) || (!gate && !aligned16(x)) || !aligned16(gx) || (g2 && !aligned16(g2))) return QS_ERR_ALIGN;
    // the riders of the all-fp32 kernel form (BwdRiders, qs_elementwise.h)
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
    const void* second = gate ? (const void*)gate : x;       // the gate bitmap replaces the ReLU's input (GATE kernels)
    if (g2) {
        // two gradient streams (fp32 + a 2-byte one, or the 2-byte one alone): gate bitmap kernels, never eliding
        auto dual = [&](auto X, auto G2) {
            constexpr int XD = decltype(X)::value, G2D = decltype(G2)::value;
            int cm = plan.cm;
            if (XD == QS_F32 && cm == CM_ELEM && plan.geo.inner % 4 == 0) cm = CM_ROW;
            switch (cm) {
                case CM_SCALAR:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<QS_F32, XD, CM_SCALAR, NT, false, true, G2D>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx, act, g2, rd);
                    break;
                case CM_ROW:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<QS_F32, XD, CM_ROW, NT, false, true, G2D>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx, act, g2, rd);
                    break;
                case CM_LAST:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<QS_F32, XD, CM_LAST, NT, false, true, G2D>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx, act, g2, rd);
                    break;
                default:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<QS_F32, XD, CM_ELEM, NT, false, true, G2D>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx, act, g2, rd);
                    break;
            }
            return launch_status();
        };
        return with_dtype(xdt, [&](auto X) { return g2dt == QS_BF16 ? dual(X, IC<QS_BF16>{}) : dual(X, IC<QS_F16>{}); });
    }
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        auto go = [&](auto G, auto GT) {
            constexpr int GD = decltype(G)::value;
            constexpr bool GATE = decltype(GT)::value;
            int cm = plan.cm;
            if (GD == QS_F32 && XD == QS_F32 && cm == CM_ELEM && plan.geo.inner % 4 == 0) cm = CM_ROW;   // 4 elements per lane
            const bool el = a.elide_masked != 0 && chan_mask != nullptr;
            switch (cm) {
                case CM_SCALAR:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_SCALAR, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx, act, nullptr, rd);
                    break;
                case CM_ROW:
                    if (el)
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ROW, NT, true, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx, act, nullptr, rd);
                    else
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ROW, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx, act, nullptr, rd);
                    break;
                case CM_LAST:
                    if (el)
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_LAST, NT, true, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx, act, nullptr, rd);
                    else
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_LAST, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx, act, nullptr, rd);
                    break;
                default:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ELEM, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx, act, nullptr, rd);
                    break;
            }
            return launch_status();
        };
        if (gate) return gdt == QS_F32 ? go(IC<QS_F32>{}, std::true_type{}) : go(X, std::true_type{});
        return gdt == QS_F32 ? go(IC<QS_F32>{}, std::false_type{}) : go(X, std::false_type{});
    });
}

int qs_quant_ste_relu_bwd_v(const qs_ste_relu_bwd_args* args) {
    qs_ste_relu_bwd_args a;
    if (!take_args(args, &a)) return QS_ERR_ARG;
    if (a.act_x || a.act_x_kind) return qs_ste_act_bwd_impl(a);      // (v26) the caller's activation: api_quant_bwd_act.hip
    return ste_relu_bwd_impl(a);
}

// the positional form (ABI <= v24 callers): the descriptor's first 21 fields
int qs_quant_ste_relu_bwd(const void* g, const void* x, const uint8_t* gate, void* gx, const float* step, int64_t nstep,
                          float step_host, int step_is_decimal, float lo_mul, float hi_mul, const uint8_t* chan_mask,
                          int64_t outer, int64_t C, int64_t inner, int gdt, int xdt, int elide_masked, int act_handle, const void* g2,
                          int g2dt, qs_stream_t stream) {
    qs_ste_relu_bwd_args a{};
    a.struct_size = sizeof(a);
    a.gdt = gdt, a.xdt = xdt, a.g2dt = g2dt;
    a.g = g, a.x = x, a.gate = gate, a.gx = gx;
    a.step = step, a.nstep = nstep, a.step_host = step_host, a.step_is_decimal = step_is_decimal;
    a.lo_mul = lo_mul, a.hi_mul = hi_mul, a.chan_mask = chan_mask;
    a.outer = outer, a.C = C, a.inner = inner;
    a.elide_masked = elide_masked, a.act = act_handle, a.g2 = g2, a.stream = stream;
    return ste_relu_bwd_impl(a);
}

// ------------------------------------------------------------------------------------------------
int qs_mask_apply(const void* x, const uint8_t* mask, void* y, int ndim, const int64_t* sizes, const int64_t* mask_strides,
                  int dt, int pre_relu, int elide_masked, uint8_t* gate_out, qs_stream_t stream) {
    if (!x || !mask || !y || !sizes || !mask_strides || ndim < 1 || (gate_out && !pre_relu)) return QS_ERR_ARG;
    ActSpec act;
    if (qs_act_resolve(pre_relu, &act) != QS_OK) return QS_ERR_ARG;
    if (!dt_ok(dt)) return QS_ERR_DTYPE;
    if (!aligned16(x) || !aligned16(y)) return QS_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    // collapse: drop extent-1 dims, merge neighbours that are both broadcast or contiguous in the mask
    int64_t cs[64], cm[64];
    int nd = 0;
    int64_t numel = 1;
    if (ndim > 64) return QS_ERR_RANK;
    for (int d = 0; d < ndim; ++d) {
        if (sizes[d] < 0) return QS_ERR_ARG;
        numel *= sizes[d];
        if (sizes[d] == 1) continue;
        const int64_t ms = mask_strides[d];
        if (nd > 0 && ((cm[nd - 1] == 0 && ms == 0) || (ms != 0 && cm[nd - 1] == ms * sizes[d]))) {
            cs[nd - 1] *= sizes[d];
            cm[nd - 1] = ms;
        } else {
            cs[nd] = sizes[d];
            cm[nd] = ms;
            ++nd;
        }
    }
    if (numel == 0) return QS_OK;
    if (nd == 0) {  // single element
        cs[0] = 1;
        cm[0] = 0;
        nd = 1;
    }
    // pattern A: [outer bcast][C dense, unit stride][inner bcast]
    int64_t outer = 1, C = 1, inner = 1;
    bool pattern_a = false, full = false;
    if (nd == 1 && cm[0] == 1) {
        if (pre_relu) { pattern_a = true; C = cs[0]; }   // every element has its own mask entry: a channel mask with inner = 1
        else full = true;
    }
    else if (nd == 1 && cm[0] == 0) { pattern_a = true; outer = cs[0]; }
    else if (nd == 2 && cm[0] == 0 && cm[1] == 1) { pattern_a = true; outer = cs[0]; C = cs[1]; }
    else if (nd == 2 && cm[0] == 1 && cm[1] == 0) { pattern_a = true; C = cs[0]; inner = cs[1]; }
    else if (nd == 3 && cm[0] == 0 && cm[1] == 1 && cm[2] == 0) { pattern_a = true; outer = cs[0]; C = cs[1]; inner = cs[2]; }

    if (pattern_a) {
        EwPlan plan;
        int st = plan_ew(outer, C, inner, true, &plan, aligned8(mask));
        if (st) return st;
        ChanMaskOp op{mask, act, dt};
        return with_dtype(dt, [&](auto D) {
            constexpr int DD = decltype(D)::value;
            if (gate_out) {      // the folded ReLU's gate bitmap for the backward (GateOp, qs_elementwise.h)
                GateOp<ChanMaskOp> gop{op, gate_out, elide_masked != 0, nullptr, QS_BF16, nullptr};
                return launch_ew<GateOp<ChanMaskOp>, DD, DD>(gop, plan, true, x, y, nullptr, s);
            }
            return launch_ew<ChanMaskOp, DD, DD>(op, plan, true, x, y, nullptr, s, elide_masked != 0);
        });
    }
    if (pre_relu) return QS_ERR_ARG;   // the ReLU fold exists for channel-type masks only
    if (full) {
        const int grid = grid_for(numel / 8, 1);
        return with_dtype(dt, [&](auto D) {
            constexpr int DD = decltype(D)::value;
            hipLaunchKernelGGL((mask_full_kernel<DD, (QS_EW_NT != 0)>), dim3(grid), dim3(kBlock), 0, s, x, mask, y, numel);
            return launch_status();
        });
    }
    if (nd > QS_MAX_DIMS) return QS_ERR_RANK;
    BcastGeom geo;
    geo.ndim = nd;
    for (int d = 0; d < nd; ++d) {
        geo.sizes[d] = cs[d];
        geo.mstrides[d] = cm[d];
    }
    if (cs[nd - 1] % 8 == 0 && (cm[nd - 1] == 0 || cm[nd - 1] == 1) && numel / 8 / kBlock < 0x7fffffff) {
        const int64_t ngroups = numel / 8;    // 16-byte accesses: a lane's 8 elements share every index but the innermost
        return with_dtype(dt, [&](auto D) {
            constexpr int DD = decltype(D)::value;
            hipLaunchKernelGGL((mask_bcast_vec_kernel<DD, (QS_EW_NT != 0)>), dim3((int)((ngroups + kBlock - 1) / kBlock)),
                               dim3(kBlock), 0, s, x, mask, y, ngroups, geo);
            return launch_status();
        });
    }
    int64_t blocks = (numel + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;   // grid-stride
    return with_dtype(dt, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        hipLaunchKernelGGL((mask_bcast_kernel<DD>), dim3((int)blocks), dim3(kBlock), 0, s, x, mask, y, numel, geo);
        return launch_status();
    });
}

}  // extern "C"
