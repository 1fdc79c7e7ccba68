// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), quantizer forward kernels (qs_elementwise.h): the Decimal and Line quantizers
// (api_quant_fwd.hip: Scaler).  Host side: argument checks, geometry, launch configuration.  No allocation, no synchronisation.
#include "qs_host_ew.h"

extern "C" {

int qs_quant_decimal_fwd(const void* x, void* y, int32_t* codes, const float* decimal, int64_t ndecimal,
                         float decimal_host, const uint8_t* chan_mask, int64_t outer, int64_t C, int64_t inner, int xdt,
                         int ydt, int qdt, int saturate, int32_t code_lo, int32_t code_hi, int pre_relu,
                         int elide_masked, uint8_t* gate_out, void* image_out, int imgdt, void* xback_out, qs_stream_t stream) {
    if (!x || !y || (gate_out && !pre_relu)) return QS_ERR_ARG;
    if (image_out && !image_route_ok(outer, C, inner, ndecimal > 1, chan_mask, codes, gate_out, xdt, ydt, imgdt, image_out)) return QS_ERR_ARG;
    ActSpec act;
    if (qs_act_resolve(pre_relu, &act) != QS_OK) return QS_ERR_ARG;
    if (xback_out && (act.kind == QS_ACT_NONE || !aligned16(xback_out) ||
                      !widen_route_ok(outer, C, inner, ndecimal > 1, chan_mask, codes, gate_out, xdt, ydt)))
        return QS_ERR_ARG;
    if (!dt_ok(xdt) || !dt_ok(ydt) || !dt_ok(qdt)) return QS_ERR_DTYPE;
    if (!(ydt == QS_F32 || ydt == xdt) || !(qdt == QS_F32 || qdt == xdt)) return QS_ERR_DTYPE;
    if (!aligned16(x) || !aligned16(y) || (codes && !aligned16(codes))) return QS_ERR_ALIGN;
    int st = check_param(decimal, ndecimal, C);
    if (st) return st;
    const bool ppc = ndecimal > 1;
    EwPlan plan;
    st = plan_ew(outer, C, inner, ppc || chan_mask != nullptr, &plan, !ppc && aligned8(chan_mask));
    if (st) return st;
    hipStream_t s = (hipStream_t)stream;
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        auto go = [&](auto Y, auto Q) {
            constexpr int YD = decltype(Y)::value, QD = decltype(Q)::value;
            DecimalFwdOp<QD> op{decimal, decimal_host, chan_mask, saturate, code_lo, code_hi, act, xdt};
            if (gate_out) {
                // (the gate-recording kernels exist for a float32 quotient only -- what every layer passes; a 2-byte quotient is the
                //  functional API's Python-float-scale corner, quantize.py:109 on a half tensor, which never folds an activation)
                if constexpr (QD == QS_F32) {
                    GateOp<DecimalFwdOp<QD>> gop{op, gate_out, elide_masked != 0 && chan_mask != nullptr, image_out, imgdt, xback_out};
                    return launch_ew<GateOp<DecimalFwdOp<QD>>, XD, YD>(gop, plan, ppc, x, y, codes, s);
                } else {
                    return (int)QS_ERR_DTYPE;
                }
            }
            return launch_ew<DecimalFwdOp<QD>, XD, YD>(op, plan, ppc, x, y, codes, s, elide_masked != 0);
        };
        if (ydt == QS_F32) return (qdt == QS_F32) ? go(IC<QS_F32>{}, IC<QS_F32>{}) : go(IC<QS_F32>{}, X);
        return (qdt == QS_F32) ? go(X, IC<QS_F32>{}) : go(X, X);
    });
}

int qs_quant_line_fwd(const void* x, void* y, int32_t* codes, const float* lines, int64_t nlines, int bits, int float_zero_point,
                      int64_t outer, int64_t C, int64_t inner, int xdt, int ydt, qs_stream_t stream) {
    if (!x || !y || !lines) return QS_ERR_ARG;
    if (!dt_ok(xdt) || ydt != QS_F32) return QS_ERR_DTYPE;
    if (!aligned16(x) || !aligned16(y) || (codes && !aligned16(codes))) return QS_ERR_ALIGN;
    if (bits < 1 || bits > 24) return QS_ERR_ARG;
    int st = check_param(lines, nlines, C);
    if (st) return st;
    const bool ppc = nlines > 1;
    EwPlan plan;
    st = plan_ew(outer, C, inner, ppc, &plan);
    if (st) return st;
    hipStream_t s = (hipStream_t)stream;
    const float nlevels = (float)(1 << bits);
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        if (float_zero_point) {
            LineFwdOp<true> op{lines, nlevels, 1.0f / nlevels};
            return launch_ew<LineFwdOp<true>, XD, QS_F32>(op, plan, ppc, x, y, codes, s);
        }
        LineFwdOp<false> op{lines, nlevels, 1.0f / nlevels};
        return launch_ew<LineFwdOp<false>, XD, QS_F32>(op, plan, ppc, x, y, codes, s);
    });
}

}  // extern "C"
