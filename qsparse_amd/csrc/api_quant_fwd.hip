// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), quantizer forward kernels (qs_elementwise.h): the Scaler quantizer
// (api_quant_fwd2.hip: Decimal and Line -- two units compile side by side).
// Host side: argument checks, geometry, launch configuration.  No allocation, no synchronisation: every entry
// point only enqueues work on the caller's stream.
#include "qs_host_ew.h"

extern "C" {

int qs_quant_image_ok(int64_t outer, int64_t C, int64_t inner, int per_channel_param, int has_mask, int mask_aligned8, int xdt) {
    static const uint8_t dummy_gate = 0;
    alignas(16) static const char img[16] = {0};
    // (only the alignment class of the mask pointer matters to the plan: an aligned or a deliberately odd stand-in)
    alignas(8) static const uint8_t mask8[16] = {0};
    const uint8_t* cm = has_mask ? (mask_aligned8 ? mask8 : mask8 + 1) : nullptr;
    return image_route_ok(outer, C, inner, per_channel_param != 0, cm, nullptr, &dummy_gate, xdt, QS_F32, QS_BF16, img) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
int qs_quant_scaler_fwd(const void* x, void* y, int32_t* codes, const float* scale, int64_t nscale, float scale_host,
                        const uint8_t* chan_mask, int64_t outer, int64_t C, int64_t inner, int xdt, int ydt, int qdt,
                        int saturate, int32_t code_lo, int32_t code_hi, int pre_relu, int elide_masked, uint8_t* gate_out,
                        void* image_out, int imgdt, void* xback_out, qs_stream_t stream) {
    if (!x || !y || (gate_out && !pre_relu)) return QS_ERR_ARG;
    if (image_out && !image_route_ok(outer, C, inner, nscale > 1, chan_mask, codes, gate_out, xdt, ydt, imgdt, image_out)) return QS_ERR_ARG;
    ActSpec act;
    if (qs_act_resolve(pre_relu, &act) != QS_OK) return QS_ERR_ARG;
    if (xback_out && (act.kind == QS_ACT_NONE || !aligned16(xback_out) ||
                      !widen_route_ok(outer, C, inner, nscale > 1, chan_mask, codes, gate_out, xdt, ydt)))
        return QS_ERR_ARG;
    if (!dt_ok(xdt) || !dt_ok(ydt) || !dt_ok(qdt)) return QS_ERR_DTYPE;
    if (!(ydt == QS_F32 || ydt == xdt) || !(qdt == QS_F32 || qdt == xdt)) return QS_ERR_DTYPE;
    if (!aligned16(x) || !aligned16(y) || (codes && !aligned16(codes))) return QS_ERR_ALIGN;
    int st = check_param(scale, nscale, C);
    if (st) return st;
    const bool ppc = nscale > 1;
    EwPlan plan;
    st = plan_ew(outer, C, inner, ppc || chan_mask != nullptr, &plan, !ppc && aligned8(chan_mask));
    if (st) return st;
    hipStream_t s = (hipStream_t)stream;
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        auto go = [&](auto Y, auto Q) {
            constexpr int YD = decltype(Y)::value, QD = decltype(Q)::value;
            ScalerFwdOp<QD> op{scale, scale_host, chan_mask, saturate, code_lo, code_hi, act, xdt};
            if (gate_out) {
                // (the gate-recording kernels exist for a float32 quotient only -- what every layer passes; a 2-byte quotient is the
                //  functional API's Python-float-scale corner, quantize.py:109 on a half tensor, which never folds an activation)
                if constexpr (QD == QS_F32) {
                    GateOp<ScalerFwdOp<QD>> gop{op, gate_out, elide_masked != 0 && chan_mask != nullptr, image_out, imgdt, xback_out};
                    return launch_ew<GateOp<ScalerFwdOp<QD>>, XD, YD>(gop, plan, ppc, x, y, codes, s);
                } else {
                    return (int)QS_ERR_DTYPE;
                }
            }
            return launch_ew<ScalerFwdOp<QD>, XD, YD>(op, plan, ppc, x, y, codes, s, elide_masked != 0);
        };
        if (ydt == QS_F32) return (qdt == QS_F32) ? go(IC<QS_F32>{}, IC<QS_F32>{}) : go(IC<QS_F32>{}, X);
        return (qdt == QS_F32) ? go(X, IC<QS_F32>{}) : go(X, X);
    });
}

}  // extern "C"
