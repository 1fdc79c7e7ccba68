// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), the DESCRIPTOR entry points (qs_*_v, "ABI compatibility" in the header): the
// operands of a call in a size-prefixed, append-only struct.  Host code only.  These four forward to the positional entry points
// (whose prototypes are frozen as of v25); an entry point that grows an operand moves its body behind the descriptor instead, as
// qs_quant_ste_relu_bwd_v / qs_site_bwd_v (api_quant_bwd.hip, api_core.hip) did in v25.
#include "qs_host.h"

extern "C" {

int qs_abi_floor(void) { return 25; }

int qs_quant_fwd_v(const qs_quant_fwd_args* args) {
    qs_quant_fwd_args a;
    if (!take_args(args, &a)) return QS_ERR_ARG;
    if (a.kind == QS_QUANT_SCALER)
        return qs_quant_scaler_fwd(a.x, a.y, a.codes, a.param, a.nparam, a.param_host, a.chan_mask, a.outer, a.C, a.inner, a.xdt, a.ydt,
                                   a.qdt, a.saturate, a.code_lo, a.code_hi, a.pre_relu, a.elide_masked, a.gate_out, a.image_out, a.imgdt,
                                   a.xback_out, a.stream);
    if (a.kind == QS_QUANT_DECIMAL)
        return qs_quant_decimal_fwd(a.x, a.y, a.codes, a.param, a.nparam, a.param_host, a.chan_mask, a.outer, a.C, a.inner, a.xdt, a.ydt,
                                    a.qdt, a.saturate, a.code_lo, a.code_hi, a.pre_relu, a.elide_masked, a.gate_out, a.image_out, a.imgdt,
                                    a.xback_out, a.stream);
    return QS_ERR_ARG;
}

int qs_pq_select_v(const qs_pq_select_args* args) {
    qs_pq_select_args a;
    if (!take_args(args, &a)) return QS_ERR_ARG;
    return qs_pq_select(a.magnitude, a.stage_mean, a.sdt, a.C, a.update_magnitude, a.t_mag, a.refresh_mask, a.k, a.mask, a.chan_absmax,
                        a.chan_absmax_stride, a.update_scale, a.t_q, a.bits, a.scale, a.bump_i32_a, a.bump_i32_b, a.bump_i64_a,
                        a.bump_i64_b, a.t_mag_dev, a.t_q_dev, a.stat_dt, a.gathered, a.world, a.elide_mask_out, a.stream);
}

int qs_site_fwd_v(const qs_site_plan* plan, const qs_site_fwd_args* args) {
    qs_site_fwd_args a;
    if (!take_args(args, &a)) return QS_ERR_ARG;
    return qs_site_fwd(plan, a.x, a.y, a.gate_out, a.flags, a.t_mag, a.k, a.t_q, a.image_out, a.imgdt, a.gathered, a.world, a.xback_out,
                       a.decimal, a.stream);
}

int qs_quantize_step_v(const qs_quantize_step_args* args) {
    qs_quantize_step_args a;
    if (!take_args(args, &a)) return QS_ERR_ARG;
    return qs_quantize_step(a.x, a.y, a.gate_out, a.amax_lines, a.lines, a.scale, a.numel, a.xdt, a.ydt, a.bits, a.t, a.t_dev,
                            a.n_updates, a.pre_relu, a.update, a.saturate, a.code_lo, a.code_hi, a.xback_out, a.image_out, a.imgdt,
                            a.stream);
}

}  // extern "C"
