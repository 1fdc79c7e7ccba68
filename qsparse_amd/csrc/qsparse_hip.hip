// libqsparse_hip.so -- C ABI (include/qsparse_hip.h) over the gfx950 kernels in qs_elementwise.h and
// qs_reduce.h.  Host side: argument checks, geometry, launch configuration.  No allocation, no
// synchronisation: every entry point only enqueues work on the caller's stream.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "qs_elementwise.h"
#include "qs_reduce.h"
#include "qs_multi.h"

#ifndef QS_EW_UNROLL
#define QS_EW_UNROLL 1
#endif
#ifndef QS_EW_UNROLL_ELIDE
#define QS_EW_UNROLL_ELIDE 2   // groups per lane of the 8-per-lane kernels when they elide (their pruned waves only store)
#endif
#ifndef QS_EW_UNROLL_SCALAR
#define QS_EW_UNROLL_SCALAR 1  // groups per lane of the tensor-wise (CM_SCALAR) 8-per-lane kernels
#endif
#ifndef QS_EW_UNROLL_ROW
#define QS_EW_UNROLL_ROW 2     // groups per lane of the dense per-row (CM_ROW) 8-per-lane kernels (three-phase path)
#endif
#ifndef QS_EW_NT
#define QS_EW_NT 1
#endif
#ifndef QS_MEAN_ROWS_IN_FLIGHT
#define QS_MEAN_ROWS_IN_FLIGHT 16   // 16 KiB per wave outstanding; 8 was 8 % slower once the loads bypass the Infinity Cache, 32 no faster
#endif

using namespace qs;

namespace {

inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
// streaming kernels: one workgroup per kBlock*UNROLL groups (measured best on MI355X: exact grids beat a
// capped grid-stride loop by 8-10 %); QS_MAX_BLOCKS caps the grid for experiments
inline int max_blocks() {
    static int v = env_int("QS_MAX_BLOCKS", 1 << 30);
    return v;
}
inline int reduce_blocks() {
    static int v = env_int("QS_REDUCE_BLOCKS", 256);
    return v;
}
inline int reduce_blocks_lines() {
    static int v = env_int("QS_REDUCE_BLOCKS_LINES", 256);   // measured: 512 / 1024 / 2048 workgroups are 9-22 % SLOWER on 411 MB
    return v;
}
// Streaming kernels walk their tensors from the END: the producer (or the statistics pass that has just read
// the same tensor front to back) leaves the tail of the tensor in the 256 MiB Infinity Cache, and an LRU
// cache serves a reverse walk from it where a forward walk would evict it before use.  QS_EW_REVERSE=0 disables.
inline int ew_reverse() {
    static int v = env_int("QS_EW_REVERSE", 1);
    return v;
}
// active lanes per wave of the column-parallel mean kernel: the busiest CU carries ceil(waves / 256) * lanes
// column groups; pick the widest wave that minimises it (QS_MEAN_LANES overrides)
inline int mean_lanes(int64_t total) {
    static int forced = env_int("QS_MEAN_LANES", 0);
    if (forced >= 1 && forced <= 64) return forced;
    const int64_t kCUs = 256;
    int best = 64;
    int64_t best_cost = -1;
    for (int l = 64; l >= 40; --l) {
        const int64_t waves = (total + l - 1) / l;
        const int64_t cost = ((waves + kCUs - 1) / kCUs) * l;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = l;
        }
    }
    return best;
}

constexpr size_t kLast2MaxLds = 63 * 1024;    // qs_mean_last2's [H*W + W + 8] float tile (a workgroup may hold 64 KiB)

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
inline int dt_ok(int dt) { return dt == QS_F32 || dt == QS_BF16 || dt == QS_F16; }
inline int hip_status(hipError_t e) { return (int)e; }
inline int launch_status() { return hip_status(hipGetLastError()); }

template <int V>
using IC = std::integral_constant<int, V>;

template <typename F>
int with_dtype(int dt, F&& f) {
    switch (dt) {
        case QS_F32: return f(IC<QS_F32>{});
        case QS_BF16: return f(IC<QS_BF16>{});
        case QS_F16: return f(IC<QS_F16>{});
    }
    return QS_ERR_DTYPE;
}

int grid_for(int64_t ngroups, int unroll) {
    int64_t b = (ngroups + (int64_t)kBlock * unroll - 1) / ((int64_t)kBlock * unroll);
    if (b < 1) b = 1;
    if (b > max_blocks()) b = max_blocks();
    return (int)b;
}

// geometry + channel mode of an element-wise launch over [outer, C, inner]
struct EwPlan {
    EwGeom geo;
    int cm;
};
inline bool aligned8(const void* p) { return (((uintptr_t)p) & 7u) == 0; }

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

extern "C" {

int qs_version(void) { return QS_ABI_VERSION; }

const char* qs_status_string(int status) {
    switch (status) {
        case QS_OK: return "ok";
        case QS_ERR_DTYPE: return "qsparse_hip: unsupported dtype combination";
        case QS_ERR_ARG: return "qsparse_hip: inconsistent arguments";
        case QS_ERR_ALIGN: return "qsparse_hip: data pointer is not 16-byte aligned";
        case QS_ERR_WORKSPACE: return "qsparse_hip: workspace too small";
        case QS_ERR_RANK: return "qsparse_hip: broadcast pattern has too many dimensions";
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "qsparse_hip: unknown status";
}

size_t qs_workspace_bytes(int op, int64_t n) {
    (void)n;
    switch (op) {
        case QS_WS_KTH_VALUE: return sizeof(SelectState);
        case QS_WS_REDUCE: return n >= 8 && n <= kFewColsMaxCols ? (size_t)2 * kFewColsMaxBlocks * n * sizeof(uint32_t) : 0;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
int qs_quant_scaler_fwd(const void* x, void* y, int32_t* codes, const float* scale, int64_t nscale, float scale_host,
                        const uint8_t* chan_mask, int64_t outer, int64_t C, int64_t inner, int xdt, int ydt, int qdt,
                        int saturate, int32_t code_lo, int32_t code_hi, int pre_relu, int elide_masked, uint8_t* gate_out,
                        qs_stream_t stream) {
    if (!x || !y || (gate_out && !pre_relu)) return QS_ERR_ARG;
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
            ScalerFwdOp<QD> op{scale, scale_host, chan_mask, saturate, code_lo, code_hi, pre_relu};
            if (gate_out) {
                GateOp<ScalerFwdOp<QD>> gop{op, gate_out, elide_masked != 0 && chan_mask != nullptr};
                return launch_ew<GateOp<ScalerFwdOp<QD>>, XD, YD>(gop, plan, ppc, x, y, codes, s);
            }
            return launch_ew<ScalerFwdOp<QD>, XD, YD>(op, plan, ppc, x, y, codes, s, elide_masked != 0);
        };
        if (ydt == QS_F32) return (qdt == QS_F32) ? go(IC<QS_F32>{}, IC<QS_F32>{}) : go(IC<QS_F32>{}, X);
        return (qdt == QS_F32) ? go(X, IC<QS_F32>{}) : go(X, X);
    });
}

int qs_quant_decimal_fwd(const void* x, void* y, int32_t* codes, const float* decimal, int64_t ndecimal,
                         float decimal_host, const uint8_t* chan_mask, int64_t outer, int64_t C, int64_t inner, int xdt,
                         int ydt, int qdt, int saturate, int32_t code_lo, int32_t code_hi, int pre_relu,
                         int elide_masked, uint8_t* gate_out, qs_stream_t stream) {
    if (!x || !y || (gate_out && !pre_relu)) return QS_ERR_ARG;
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
            DecimalFwdOp<QD> op{decimal, decimal_host, chan_mask, saturate, code_lo, code_hi, pre_relu};
            if (gate_out) {
                GateOp<DecimalFwdOp<QD>> gop{op, gate_out, elide_masked != 0 && chan_mask != nullptr};
                return launch_ew<GateOp<DecimalFwdOp<QD>>, XD, YD>(gop, plan, ppc, x, y, codes, s);
            }
            return launch_ew<DecimalFwdOp<QD>, XD, YD>(op, plan, ppc, x, y, codes, s, elide_masked != 0);
        };
        if (ydt == QS_F32) return (qdt == QS_F32) ? go(IC<QS_F32>{}, IC<QS_F32>{}) : go(IC<QS_F32>{}, X);
        return (qdt == QS_F32) ? go(X, IC<QS_F32>{}) : go(X, X);
    });
}

int qs_quant_line_fwd(const void* x, void* y, const float* lines, int64_t nlines, int bits, int float_zero_point,
                      int64_t outer, int64_t C, int64_t inner, int xdt, int ydt, qs_stream_t stream) {
    if (!x || !y || !lines) return QS_ERR_ARG;
    if (!dt_ok(xdt) || ydt != QS_F32) return QS_ERR_DTYPE;
    if (!aligned16(x) || !aligned16(y)) return QS_ERR_ALIGN;
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
            return launch_ew<LineFwdOp<true>, XD, QS_F32>(op, plan, ppc, x, y, nullptr, s);
        }
        LineFwdOp<false> op{lines, nlevels, 1.0f / nlevels};
        return launch_ew<LineFwdOp<false>, XD, QS_F32>(op, plan, ppc, x, y, nullptr, s);
    });
}

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

int qs_quant_ste_relu_bwd(const void* g, const void* x, const uint8_t* gate, void* gx, const float* step, int64_t nstep,
                          float step_host, int step_is_decimal, float lo_mul, float hi_mul, const uint8_t* chan_mask,
                          int64_t outer, int64_t C, int64_t inner, int gdt, int xdt, int elide_masked, qs_stream_t stream) {
    if (!g || (!x && !gate) || !gx) return QS_ERR_ARG;
    if (!dt_ok(gdt) || !dt_ok(xdt) || !(gdt == QS_F32 || gdt == xdt)) return QS_ERR_DTYPE;
    if (!aligned16(g) || (!gate && !aligned16(x)) || !aligned16(gx)) return QS_ERR_ALIGN;
    int st = check_param(step, nstep, C);
    if (st) return st;
    const bool ppc = nstep > 1;
    EwPlan plan;
    st = plan_ew(outer, C, inner, ppc || chan_mask != nullptr, &plan, !ppc && aligned8(chan_mask));
    if (st) return st;
    if (plan.geo.numel == 0) return QS_OK;
    hipStream_t s = (hipStream_t)stream;
    SteBwdOp op{step, step_host, step_is_decimal, lo_mul, hi_mul, 0, chan_mask};
    const int grid = grid_for(plan.geo.ngroups, 1);
    constexpr bool NT = QS_EW_NT != 0;
    const void* second = gate ? (const void*)gate : x;       // the gate bitmap replaces the ReLU's input (GATE kernels)
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        auto go = [&](auto G, auto GT) {
            constexpr int GD = decltype(G)::value;
            constexpr bool GATE = decltype(GT)::value;
            int cm = plan.cm;
            if (GD == QS_F32 && XD == QS_F32 && cm == CM_ELEM && plan.geo.inner % 4 == 0) cm = CM_ROW;   // 4 elements per lane
            const bool el = elide_masked != 0 && chan_mask != nullptr;
            switch (cm) {
                case CM_SCALAR:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_SCALAR, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx);
                    break;
                case CM_ROW:
                    if (el)
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ROW, NT, true, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx);
                    else
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ROW, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx);
                    break;
                case CM_LAST:
                    if (el)
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_LAST, NT, true, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx);
                    else
                        hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_LAST, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                           op, plan.geo, (int)ppc, g, second, gx);
                    break;
                default:
                    hipLaunchKernelGGL((ste_relu_bwd_kernel<GD, XD, CM_ELEM, NT, false, GATE>), dim3(grid), dim3(kBlock), 0, s,
                                       op, plan.geo, (int)ppc, g, second, gx);
                    break;
            }
            return launch_status();
        };
        if (gate) return gdt == QS_F32 ? go(IC<QS_F32>{}, std::true_type{}) : go(X, std::true_type{});
        return gdt == QS_F32 ? go(IC<QS_F32>{}, std::false_type{}) : go(X, std::false_type{});
    });
}

// ------------------------------------------------------------------------------------------------
static int reduce_impl(const void* x, float* out_a, float* out_b, bool minmax, int per_channel, int64_t outer, int64_t C,
                       int64_t inner, int xdt, hipStream_t s, bool accumulate = false, int relu = 0, void* ws = nullptr,
                       size_t ws_bytes = 0, int lines = 1) {
    if (!x || !out_a || (minmax && !out_b)) return QS_ERR_ARG;
    if (!dt_ok(xdt)) return QS_ERR_DTYPE;
    if (outer < 0 || C < 1 || inner < 1) return QS_ERR_ARG;
    const int64_t numel = outer * C * inner;
    const int64_t nout = per_channel ? C : 1;
    uint32_t* omax = (uint32_t*)(minmax ? out_b : out_a);
    uint32_t* omin = minmax ? (uint32_t*)out_a : nullptr;
    const int ib = (int)((nout + 255) / 256);
    const bool vec_ptr = aligned16(x);
    // the few-columns route (channels_last activations, 2-d inputs) ends in a finish kernel that can write the final
    // floats itself: a non-accumulating call then needs neither the key initialisation nor the key -> float launch
    // (two launches instead of four for a per-channel min/max)
    int64_t few_nblk = 0;
    if (per_channel && numel > 0 && vec_ptr && ws) {
        const int64_t cols = C * inner;
        const bool rows_route = !minmax && inner % 8 == 0 && outer >= 16 && cols / 8 >= 64 * 1024;          // column walk
        const bool long_rows = inner >= 64 && C < 65536 && !(inner < 512 && cols % 8 == 0);                     // reduce_rows
        if (!rows_route && !long_rows && cols % 8 == 0 && cols <= kFewColsMaxCols && cols / 8 <= kBlock) {
            const int64_t rows_per_iter = kBlock / (cols / 8);
            static const int fewcols_cap = env_int("QS_FEWCOLS_BLOCKS", kFewColsMaxBlocks);
            int64_t nblk = outer / (rows_per_iter * 32);     // >= 4 rounds of 8 loads per workgroup
            nblk = std::min<int64_t>(std::max<int64_t>(nblk, 1), std::min(fewcols_cap, kFewColsMaxBlocks));
            if (ws_bytes >= (size_t)(2 * nblk * cols) * sizeof(uint32_t) && outer >= 32 * rows_per_iter) few_nblk = nblk;
        }
    }
    const bool finalize = few_nblk > 0 && !accumulate;
    if (!accumulate && !finalize) hipLaunchKernelGGL(keys_init_kernel, dim3(ib), dim3(256), 0, s, omax, omin, nout);
    if (numel > 0) {
        int st = with_dtype(xdt, [&](auto X) {
            constexpr int XD = decltype(X)::value;
            auto run = [&](auto MM) {
                constexpr bool M = decltype(MM)::value != 0;
                if (!per_channel) {
                    if (!vec_ptr) return (int)QS_ERR_ALIGN;
                    // 512-thread workgroups, at most 256 of them: every one ends with an atomic on the same word, which
                    // serialise at ~12 ns each (256x512 vs 512x256 threads: 256x64x56x56 bf16 23.6 -> 21.2 us,
                    // 64x64x56x56 12.4 -> 9.8 us; tools/bench_reduce.py)
                    // `lines` > 1 accumulator lines take the serialised same-address atomics off the kernel's tail
                    // (64x64x56x56 bf16: 10.0 -> 8.6 us); more workgroups than one per CU do not pay (tools/bench_reduce.py)
                    int grid = (grid_for(numel / 8, 4) + 1) / 2;
                    const int cap = lines > 1 ? reduce_blocks_lines() : reduce_blocks();
                    if (grid > cap) grid = cap;
                    hipLaunchKernelGGL((reduce_all_kernel<XD, M, 512>), dim3(grid), dim3(512), 0, s, x, numel, omax, omin, relu,
                                       lines);
                } else if (vec_ptr && inner % 8 == 0 && outer >= 16 && (C * inner) / 8 >= 64 * 1024) {
                    // big tensors ([N, C, H*W] with >= 1024 waves of column groups): the column walk of the statistics
                    // kernel -- a lane keeps 8 adjacent columns and loops over N in registers, one atomic per wave and
                    // channel at the end -- streams at the statistics kernel's rate, where workgroups that hop from row
                    // to row (reduce_rows_kernel) reach 5.4 TB/s (256x256x56x56 bf16: 76 us)
                    const int64_t post = C * inner, total = post / 8;
                    const int lanes = mean_lanes(total);
                    const int blocks = (int)((total + lanes - 1) / lanes);
                    if (M)       // min and max: `out` carries the min keys (MODE 6)
                        hipLaunchKernelGGL((mean_outer_vec_kernel<XD, XD, QS_MEAN_ROWS_IN_FLIGHT, 6>), dim3(blocks), dim3(64), 0, s, x,
                                           (void*)omin, (int64_t)1, outer, post, post, 0, (const int32_t*)nullptr, omax, (int64_t)1,
                                           inner, (uint32_t)C, lanes);
                    else if (relu)
                        hipLaunchKernelGGL((mean_outer_vec_kernel<XD, XD, QS_MEAN_ROWS_IN_FLIGHT, 5>), dim3(blocks), dim3(64), 0, s, x,
                                           (void*)nullptr, (int64_t)1, outer, post, post, 0, (const int32_t*)nullptr, omax, (int64_t)1,
                                           inner, (uint32_t)C, lanes);
                    else
                        hipLaunchKernelGGL((mean_outer_vec_kernel<XD, XD, QS_MEAN_ROWS_IN_FLIGHT, 4>), dim3(blocks), dim3(64), 0, s, x,
                                           (void*)nullptr, (int64_t)1, outer, post, post, 0, (const int32_t*)nullptr, omax, (int64_t)1,
                                           inner, (uint32_t)C, lanes);
                } else if (inner >= 64 && C < 65536 && !(inner < 512 && vec_ptr && (C * inner) % 8 == 0)) {
                    // (rows of 64..511 elements -- 14x14 maps -- go to the column kernel below when it can use vector
                    //  loads: a wave there reads 1 KiB of consecutive columns per row instead of one short ragged row)
                    int64_t slices = (2048 + C - 1) / C;              // ~2048 workgroups, one atomic each
                    if (slices > (outer + 3) / 4) slices = (outer + 3) / 4;
                    if (slices < 1) slices = 1;
                    const int64_t opb = (outer + slices - 1) / slices;
                    const int vec_ok = vec_ptr ? (inner % 8 == 0 ? 1 : 2) : 0;
                    hipLaunchKernelGGL((reduce_rows_kernel<XD, M>), dim3((int)C, (int)((outer + opb - 1) / opb)), dim3(kBlock),
                                       0, s, x, outer, (uint32_t)C, inner, vec_ok, opb, omax, omin, relu);
                } else {
                    const int64_t cols = C * inner;
                    const bool vec = vec_ptr && (cols % 8 == 0);
                    if (few_nblk > 0) {
                        // few columns, many rows: two stages through the caller's workspace, no atomics
                        const int64_t nblk = few_nblk;
                        uint32_t* pmax = (uint32_t*)ws;
                        uint32_t* pmin = pmax + nblk * cols;
                        hipLaunchKernelGGL((reduce_fewcols_kernel<XD, M>), dim3((int)nblk), dim3(kBlock), 0, s, x, outer,
                                           cols, pmax, pmin, relu);
                        hipLaunchKernelGGL((reduce_fewcols_finish_kernel<M>), dim3((int)((C + 15) / 16)), dim3(kBlock), 0, s,
                                           pmax, pmin, (int)nblk, cols, inner, omax, omin, (int)finalize);
                        return launch_status();
                    }
                    const int64_t per_block = vec ? (int64_t)kBlock * 8 : kBlock;
                    const int gx = (int)((cols + per_block - 1) / per_block);
                    int64_t gy = 1;
                    if (gx < 1024) gy = (1024 + gx - 1) / gx;     // enough workgroups to fill the chip
                    if (gy > (outer + 7) / 8) gy = (outer + 7) / 8;
                    if (gy < 1) gy = 1;
                    const int64_t opb = (outer + gy - 1) / gy;
                    if (vec)
                        hipLaunchKernelGGL((reduce_cols_vec_kernel<XD, M>), dim3(gx, (int)gy), dim3(kBlock), 0, s, x, outer,
                                           cols, inner, opb, omax, omin, relu);
                    else
                        hipLaunchKernelGGL((reduce_cols_kernel<XD, M>), dim3(gx, (int)gy), dim3(kBlock), 0, s, x, outer, cols,
                                           inner, opb, omax, omin, relu);
                }
                return launch_status();
            };
            return minmax ? run(IC<1>{}) : run(IC<0>{});
        });
        if (st) return st;
    }
    if (minmax && !finalize && !accumulate) hipLaunchKernelGGL(keys_to_float_kernel, dim3(ib), dim3(256), 0, s, omax, omin, nout);
    return launch_status();
}

int qs_absmax(const void* x, float* out, int per_channel, int64_t outer, int64_t C, int64_t inner, int xdt, int accumulate,
              int pre_relu, int out_lines, void* ws, size_t ws_bytes, qs_stream_t stream) {
    if (out_lines < 1 || out_lines > 64 || (out_lines > 1 && (per_channel || !accumulate))) return QS_ERR_ARG;
    return reduce_impl(x, out, nullptr, false, per_channel, outer, C, inner, xdt, (hipStream_t)stream, accumulate != 0,
                       pre_relu != 0, ws, ws_bytes, out_lines);
}

int qs_minmax(const void* x, float* out_min, float* out_max, int per_channel, int64_t outer, int64_t C, int64_t inner,
              int xdt, int accumulate, void* ws, size_t ws_bytes, qs_stream_t stream) {
    return reduce_impl(x, out_min, out_max, true, per_channel, outer, C, inner, xdt, (hipStream_t)stream, accumulate != 0, 0,
                       ws, ws_bytes);
}

int qs_scale_update(float* absmax, int absmax_lines, float* weight, int64_t n, int64_t t, int64_t* t_dev, int advance_t_dev,
                    int bits, int clear_absmax, int32_t* bump_i32, int stat_dt, qs_stream_t stream) {
    if (!absmax || !weight || n < 0 || t < 0 || bits < 1 || bits > 31) return QS_ERR_ARG;
    if (absmax_lines < 1 || absmax_lines > 64 || (absmax_lines > 1 && n != 1)) return QS_ERR_ARG;
    if (!dt_ok(stat_dt)) return QS_ERR_DTYPE;
    if (n == 0) return QS_OK;
    const int advance = (advance_t_dev && t_dev) ? 1 : 0;
    const int blocks = advance ? 1 : (int)((n + 255) / 256);
    hipLaunchKernelGGL(scale_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, absmax, weight, n, (float)t,
                       (float)(t + 1), (float)((int64_t)1 << (bits - 1)), t_dev, advance, clear_absmax, bump_i32, stat_dt,
                       absmax_lines);
    return launch_status();
}

int qs_lines_update(float* mn, float* mx, float* lines, int64_t n, int64_t t_after, int64_t* t_dev,
                    int advance_t_dev, int from_keys, qs_stream_t stream) {
    if (!mn || !mx || !lines || n < 0 || t_after < 1) return QS_ERR_ARG;
    if (n == 0) return QS_OK;
    const int advance = (advance_t_dev && t_dev) ? 1 : 0;
    const int blocks = advance ? 1 : (int)((n + 255) / 256);
    hipLaunchKernelGGL(lines_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mn, mx, lines, n,
                       (float)(t_after - 1), (float)t_after, t_dev, advance, from_keys != 0);
    return launch_status();
}

int qs_decimal_from_scale(const float* scale, float* decimal, int64_t n, qs_stream_t stream) {
    if (!scale || !decimal || n < 0) return QS_ERR_ARG;
    if (n == 0) return QS_OK;
    hipLaunchKernelGGL(decimal_from_scale_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scale,
                       decimal, n);
    return launch_status();
}

int qs_mean_dim(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int xdt, int odt, int flags,
                const int32_t* l0_flag, float* absmax_out, int64_t absmax_stride, int64_t chan_div, int64_t C,
                qs_stream_t stream) {
    if (!x || !out || pre < 1 || n < 1 || post < 1) return QS_ERR_ARG;
    if (!dt_ok(xdt) || !dt_ok(odt)) return QS_ERR_DTYPE;
    if (!(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    if (absmax_out && (chan_div < 1 || C < 1 || absmax_stride < 1)) return QS_ERR_ARG;
    const int64_t as = absmax_out ? absmax_stride : 1;
    hipStream_t s = (hipStream_t)stream;
    int64_t vcols = 0;
    const bool ragged_absmax = absmax_out && (chan_div % 8 != 0);
    if (post >= 64 && post % 8 == 0 && aligned16(x) && aligned16(out) && (!ragged_absmax || chan_div >= 8))
        vcols = (post / 32) * 32;
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
        else if (waves < 256 && nchunks >= 2 && nchunks <= kMaxSplitChunks) R = 4;   // measured: tools/bench_stats.py
        else if (waves < 512 && xdt != QS_F32 && nchunks >= 8 && nchunks <= kMaxSplitChunks) R = 4;   // long columns of 2-byte values
        while (R > 1 && R > nchunks) R >>= 1;
        if (nchunks > kMaxSplitChunks || nchunks < 2) R = 1;
        if (ragged_absmax && R == 1) R = (nchunks >= 2 && nchunks <= kMaxSplitChunks) ? 2 : 0;   // only the split kernel tracks two channels
        if (R == 0) vcols = 0;
    }
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
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
                    const int mode = l0_flag ? 0 : (flags == QS_MEAN_ABS ? 1 : (flags == (QS_MEAN_ABS | QS_MEAN_RELU) ? 2 :
                                                    (flags == 0 && !am ? 3 : 0)));
                    // few waves per CU: keep more rows in flight per wave instead (latency-, not bandwidth-bound)
                    // (2-byte inputs only: 32 fp32 rows of 8 columns do not fit the register file)
                    int depth = env_int("QS_MEAN_DEPTH", 0);
                    if (depth == 0) depth = (blocks < 4 * 256 && n >= 32) ? 32 : QS_MEAN_ROWS_IN_FLIGHT;
                    if (XD == QS_F32) depth = QS_MEAN_ROWS_IN_FLIGHT;
                    auto launch = [&](auto D, auto M) {
                        hipLaunchKernelGGL((mean_outer_vec_kernel<XD, OD, decltype(D)::value, decltype(M)::value>), dim3(blocks),
                                           dim3(64), 0, s, x, out, pre, n, post, vcols, flags, l0_flag, am, as, chan_div, Cc, lanes);
                    };
                    auto by_mode = [&](auto D) {
                        if (mode == 3) launch(D, IC<3>{});
                        else if (mode == 1) launch(D, IC<1>{});
                        else if (mode == 2) launch(D, IC<2>{});
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
                                      (flags == (QS_MEAN_ABS | QS_MEAN_RELU) ? 2 : 0));
                    auto launch = [&](auto RR, auto M, auto RG) {
                        constexpr int kR = decltype(RR)::value;
                        hipLaunchKernelGGL((mean_outer_split_kernel<XD, OD, kR, decltype(M)::value, decltype(RG)::value>), dim3(blocks),
                                           dim3(64 * kR), lds, s, x, out, pre, n, post, vcols, flags, l0_flag, am, as, cd, Cc, lanes);
                    };
                    auto by_mode = [&](auto RR) {
                        if (smode == 1 && rag) launch(RR, IC<1>{}, std::true_type{});
                        else if (smode == 2 && rag) launch(RR, IC<2>{}, std::true_type{});
                        else if (smode == 1) launch(RR, IC<1>{}, std::false_type{});
                        else if (smode == 2) launch(RR, IC<2>{}, std::false_type{});
                        else launch(RR, IC<0>{}, std::false_type{});
                    };
                    if (R == 2) by_mode(IC<2>{});
                    else if (R == 4) by_mode(IC<4>{});
                    else by_mode(IC<8>{});
                }
            }
            if (vcols < post) {
                const int64_t total = pre * (post - vcols);
                hipLaunchKernelGGL((mean_generic_kernel<XD, OD>), dim3((int)((total + kBlock - 1) / kBlock)), dim3(kBlock),
                                   0, s, x, out, pre, n, post, vcols, flags, l0_flag, am, as, chan_div > 0 ? chan_div : 1,
                                   (uint32_t)(C > 0 ? C : 1));
            }
            return launch_status();
        };
        return (odt == QS_F32) ? run(IC<QS_F32>{}) : run(X);
    });
}

int qs_mean_dim_cl(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags,
                   const int32_t* l0_flag, float* amax_part, qs_stream_t stream) {
    if (!x || !out || n < 1 || hw < 1 || C < 1) return QS_ERR_ARG;
    if (!dt_ok(xdt) || !(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    const int mode = (flags & QS_MEAN_L0) ? 0 : (flags == QS_MEAN_ABS ? 1 : (flags == (QS_MEAN_ABS | QS_MEAN_RELU) ? 2 :
                                                 (flags == 0 ? 3 : 0)));
    if (C % 8 != 0 || mode == 0 || (mode == 3 && amax_part) || !aligned16(x)) {
        // any channel count, the L0 variant, unaligned views: one lane per element of a sample, same summation order
        const int64_t total = hw * C;
        if ((total + kBlock - 1) / kBlock > 0x7fffffff) return QS_ERR_ARG;
        return with_dtype(xdt, [&](auto X) {
            constexpr int XD = decltype(X)::value;
            const dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
            if (odt == QS_F32)
                hipLaunchKernelGGL((mean_cl_generic_kernel<XD, QS_F32>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, out, n, hw,
                                   C, flags, l0_flag, (uint32_t*)amax_part);
            else
                hipLaunchKernelGGL((mean_cl_generic_kernel<XD, XD>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, out, n, hw, C,
                                   flags, l0_flag, (uint32_t*)amax_part);
            return launch_status();
        });
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
    const int wg_main_blocks = wg_main ? (int)((main_groups + lanes - 1) / lanes) : 0;
    const int wg_tail_blocks = wg_tail ? (int)((tail_groups + lanes - 1) / lanes) : 0;
    const int tail_blocks1 = (int)((tail_groups + 63) / 64);
    hipStream_t s = (hipStream_t)stream;
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        auto run = [&](auto O) {
            constexpr int OD = decltype(O)::value;
            uint32_t* am = (uint32_t*)amax_part;
            auto launch = [&](auto M) {
                constexpr int kM = decltype(M)::value;
                if (!wg_main && main_groups > 0)
                    hipLaunchKernelGGL((mean_cl_kernel<XD, OD, kM>), dim3(blocks1), dim3(64), 0, s, x, out, n, hw, C, am, lanes1,
                                       main_groups);
                if (wg_main || wg_tail) {
                    auto wg = [&](auto RR) {
                        constexpr int kR = decltype(RR)::value;
                        hipLaunchKernelGGL((mean_cl_wg_kernel<XD, OD, kR, kM>), dim3(wg_main_blocks + wg_tail_blocks), dim3(64 * kR),
                                           lds, s, x, out, n, hw, C, am, lanes, main_groups, wg_main_blocks, tail_groups,
                                           (int)slots);
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
                                       am, main_groups, tail_groups);
            };
            if (mode == 1) launch(IC<1>{});
            else if (mode == 2) launch(IC<2>{});
            else launch(IC<3>{});
            return launch_status();
        };
        return (odt == QS_F32) ? run(IC<QS_F32>{}) : run(X);
    });
}

int qs_mean_last2(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt, const float* amax_part,
                  float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream) {
    if (!x || !out || pre < 1 || H < 1 || W < 1) return QS_ERR_ARG;
    if ((amax_part || (record && absmax_out)) && (!absmax_out || absmax_stride < 1)) return QS_ERR_ARG;
    if (!dt_ok(xdt) || !(odt == xdt || odt == QS_F32)) return QS_ERR_DTYPE;
    const size_t lds = (size_t)(H * W + W + 8) * sizeof(float);
    if (lds > kLast2MaxLds || pre > 0x7fffffff) return QS_ERR_ARG;
    return with_dtype(xdt, [&](auto X) {
        constexpr int XD = decltype(X)::value;
        if (odt == QS_F32)
            hipLaunchKernelGGL((mean_last2_kernel<XD, QS_F32>), dim3((int)pre), dim3(kBlock), lds, (hipStream_t)stream, x,
                               out, (int)H, (int)W, (const uint32_t*)amax_part, (uint32_t*)absmax_out, absmax_stride, record);
        else
            hipLaunchKernelGGL((mean_last2_kernel<XD, XD>), dim3((int)pre), dim3(kBlock), lds, (hipStream_t)stream, x, out,
                               (int)H, (int)W, (const uint32_t*)amax_part, (uint32_t*)absmax_out, absmax_stride, record);
        return launch_status();
    });
}

int qs_l0_flag(const void* x, int64_t numel, int xdt, int32_t* flag, float* scratch2, qs_stream_t stream) {
    if (!flag || !scratch2) return QS_ERR_ARG;
    int st = reduce_impl(x, scratch2, scratch2 + 1, true, 0, 1, 1, numel > 0 ? numel : 1, xdt, (hipStream_t)stream);
    if (st) return st;
    hipLaunchKernelGGL(l0_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scratch2, flag);
    return launch_status();
}

int qs_running_mean(float* state, const void* newv, int newdt, int64_t n, int64_t t, const int64_t* t_dev,
                    qs_stream_t stream) {
    if (!state || !newv || n < 0 || t < 0) return QS_ERR_ARG;
    if (n == 0) return QS_OK;
    return with_dtype(newdt, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        hipLaunchKernelGGL((running_mean_kernel<DD>), dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           state, newv, n, (float)t, (float)(t + 1), t_dev);
        return launch_status();
    });
}

// ------------------------------------------------------------------------------------------------
int qs_kth_value(const float* imp, int64_t n, int64_t k, float* thr, void* ws, size_t ws_bytes, qs_stream_t stream) {
    if (!imp || !thr || n < 1 || k < 0 || k >= n || n >= ((int64_t)1 << 32)) return QS_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n <= 32768) {
        hipLaunchKernelGGL(kth_small_kernel, dim3(1), dim3(kSelectThreads), 0, s, imp, n, (uint32_t)k, thr);
        return launch_status();
    }
    if (!ws || ws_bytes < sizeof(SelectState)) return QS_ERR_WORKSPACE;
    SelectState* st = (SelectState*)ws;
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, s, st, (uint32_t)k);
    int64_t blocks = (n + kBlock * 8 - 1) / (kBlock * 8);
    if (blocks > 1024) blocks = 1024;
    for (int pass = 3; pass >= 0; --pass) {
        hipLaunchKernelGGL(select_hist_kernel, dim3((int)blocks), dim3(kBlock), 0, s, imp, n, pass, st);
        hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(256), 0, s, st, pass, thr);
    }
    return launch_status();
}

int qs_mask_ge(const float* imp, const float* thr, uint8_t* mask, int64_t n, qs_stream_t stream) {
    if (!imp || !thr || !mask || n < 0) return QS_ERR_ARG;
    if (n == 0) return QS_OK;
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > max_blocks()) blocks = max_blocks();
    hipLaunchKernelGGL(mask_ge_kernel, dim3((int)blocks), dim3(kBlock), 0, (hipStream_t)stream, imp, thr, mask, n);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
int qs_mask_apply(const void* x, const uint8_t* mask, void* y, int ndim, const int64_t* sizes, const int64_t* mask_strides,
                  int dt, int pre_relu, int elide_masked, uint8_t* gate_out, qs_stream_t stream) {
    if (!x || !mask || !y || !sizes || !mask_strides || ndim < 1 || (gate_out && !pre_relu)) return QS_ERR_ARG;
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
        ChanMaskOp op{mask, pre_relu != 0};
        return with_dtype(dt, [&](auto D) {
            constexpr int DD = decltype(D)::value;
            if (gate_out) {      // the folded ReLU's gate bitmap for the backward (GateOp, qs_elementwise.h)
                GateOp<ChanMaskOp> gop{op, gate_out, elide_masked != 0};
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

// ------------------------------------------------------------------------------------------------
static int pq_args(PqArgs* a, float* magnitude, int64_t C, int update_magnitude, int64_t t_mag, int refresh_mask,
                   int64_t k, uint8_t* mask, float* chan_absmax, int64_t amax_stride, int update_scale, int64_t t_q, int bits, float* scale,
                   int32_t* bump_i32_a, int32_t* bump_i32_b, int64_t* bump_i64_a, int64_t* bump_i64_b,
                   const int64_t* t_mag_dev, const int64_t* t_q_dev, const float* gathered, int world) {
    if (!magnitude || !mask || C < 1 || C > 65536) return QS_ERR_ARG;
    if (update_magnitude && t_mag < 0) return QS_ERR_ARG;
    if (refresh_mask && (k < 0 || k >= C)) return QS_ERR_ARG;
    if (gathered && world < 1) return QS_ERR_ARG;
    if (update_scale && ((!chan_absmax && !gathered) || amax_stride < 1 || !scale || bits < 1 || bits > 31 || t_q < 0))
        return QS_ERR_ARG;
    a->gathered = gathered;
    a->world = gathered ? world : 1;
    a->magnitude = magnitude;
    a->C = C;
    a->update_magnitude = update_magnitude;
    a->t_mag = (float)t_mag;
    a->t_mag1 = (float)(t_mag + 1);
    a->refresh_mask = refresh_mask;
    a->k = (uint32_t)k;
    a->mask = mask;
    a->chan_absmax = (uint32_t*)chan_absmax;
    a->amax_stride = amax_stride;
    a->update_scale = update_scale;
    a->t_q = (float)t_q;
    a->t_q1 = (float)(t_q + 1);
    a->denom = (float)((int64_t)1 << (bits > 0 ? bits - 1 : 0));
    a->scale = scale;
    a->bump_a = bump_i32_a;
    a->bump_b = bump_i32_b;
    a->bump_c = bump_i64_a;
    a->bump_d = bump_i64_b;
    a->t_mag_dev = t_mag_dev;
    a->t_q_dev = t_q_dev;
    static const int rank_small = env_int("QS_RANK_SMALL", kRankSmall);
    a->rank_small = rank_small;
    return QS_OK;
}

int qs_pq_select(float* magnitude, const void* stage_mean, int sdt, int64_t C, int update_magnitude, int64_t t_mag,
                 int refresh_mask, int64_t k, uint8_t* mask, float* chan_absmax, int64_t chan_absmax_stride, int update_scale,
                 int64_t t_q, int bits,
                 float* scale, int32_t* bump_i32_a, int32_t* bump_i32_b, int64_t* bump_i64_a, int64_t* bump_i64_b,
                 const int64_t* t_mag_dev, const int64_t* t_q_dev, int stat_dt, const float* gathered, int world,
                 qs_stream_t stream) {
    PqArgs a;
    int st = pq_args(&a, magnitude, C, update_magnitude, t_mag, refresh_mask, k, mask, chan_absmax, chan_absmax_stride, update_scale, t_q, bits,
                     scale, bump_i32_a, bump_i32_b, bump_i64_a, bump_i64_b, t_mag_dev, t_q_dev, gathered, world);
    if (st == QS_OK && update_scale && !dt_ok(stat_dt)) st = QS_ERR_DTYPE;
    a.stat_dt = stat_dt;
    if (st) return st;
    if (update_magnitude && !stage_mean && !gathered) return QS_ERR_ARG;
    if (gathered) sdt = QS_F32;      // the records are float32; `stage_mean` is not read
    if (!dt_ok(sdt)) return QS_ERR_DTYPE;
    return with_dtype(sdt, [&](auto S) {
        constexpr int SD = decltype(S)::value;
        if (C <= 256)
            hipLaunchKernelGGL((pq_select_kernel<SD, 256>), dim3(1), dim3(256), 0, (hipStream_t)stream, a, stage_mean);
        else
            hipLaunchKernelGGL((pq_select_kernel<SD, kSelectThreads>), dim3(1), dim3(kSelectThreads), 0,
                               (hipStream_t)stream, a, stage_mean);
        return launch_status();
    });
}

int qs_stats_pack(const void* stage, int sdt, const float* absmax, int64_t absmax_stride, int64_t C, float* record,
                  qs_stream_t stream) {
    if (!record || C < 1) return QS_ERR_ARG;
    if (stage && !dt_ok(sdt)) return QS_ERR_DTYPE;
    return with_dtype(stage ? sdt : QS_F32, [&](auto S) {
        constexpr int SD = decltype(S)::value;
        hipLaunchKernelGGL((stats_pack_kernel<SD>), dim3((int)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stage,
                           (const uint32_t*)absmax, absmax_stride > 0 ? absmax_stride : 1, C, record);
        return launch_status();
    });
}

int qs_stats_combine(const float* gathered, int world, int64_t C, float* stage_out, float* absmax_out,
                     int64_t absmax_stride, qs_stream_t stream) {
    if (!gathered || world < 1 || C < 1) return QS_ERR_ARG;
    hipLaunchKernelGGL(stats_combine_kernel, dim3((int)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gathered, world,
                       C, stage_out, (uint32_t*)absmax_out, absmax_stride > 0 ? absmax_stride : 1);
    return launch_status();
}

// ---- multi-tensor weight path (qs_multi.h) ----------------------------------------------------------------------
int qs_multi_absmax(int n, const float* const* x, const int64_t* numel, float* const* amax, qs_stream_t stream) {
    if (n < 0 || (n > 0 && (!x || !numel || !amax))) return QS_ERR_ARG;
    for (int base = 0; base < n; base += kMultiMax) {
        MultiTensors a{};
        MultiUpdate u{};
        a.n = u.n = std::min(kMultiMax, n - base);
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const int k = base + i;
            if (!x[k] || !amax[k] || numel[k] < 0) return QS_ERR_ARG;
            if (!aligned16(x[k])) return QS_ERR_ALIGN;
            a.x[i] = x[k];
            a.numel[i] = numel[k];
            u.amax[i] = (uint32_t*)amax[k];
            a.block0[i] = blocks;
            const int64_t want = (numel[k] / 8 + (int64_t)kBlock * 4 - 1) / ((int64_t)kBlock * 4);   // ~4 groups per lane
            blocks += (int)std::min<int64_t>(std::max<int64_t>(want, 1), 64);
        }
        a.block0[a.n] = blocks;
        hipLaunchKernelGGL(multi_absmax_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, a, u);
    }
    return launch_status();
}

int qs_multi_scale_update(int n, float* const* amax, float* const* scale, float* const* decimal, const int64_t* t,
                          int64_t* const* t_dev, const int* bits, int32_t* const* bump, qs_stream_t stream) {
    if (n < 0 || (n > 0 && (!amax || !scale || !t || !bits))) return QS_ERR_ARG;
    for (int base = 0; base < n; base += kMultiMax) {
        MultiUpdate u{};
        u.n = std::min(kMultiMax, n - base);
        for (int i = 0; i < u.n; ++i) {
            const int k = base + i;
            if (!amax[k] || !scale[k] || t[k] < 0 || bits[k] < 1 || bits[k] > 31) return QS_ERR_ARG;
            u.amax[i] = (uint32_t*)amax[k];
            u.scale[i] = scale[k];
            u.decimal[i] = decimal ? decimal[k] : nullptr;
            u.t_dev[i] = t_dev ? t_dev[k] : nullptr;
            u.bump[i] = bump ? bump[k] : nullptr;
            u.t[i] = (float)t[k];
            u.denom[i] = (float)((int64_t)1 << (bits[k] - 1));
        }
        hipLaunchKernelGGL(multi_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, u);
    }
    return launch_status();
}

int qs_multi_quant_fwd(int n, const float* const* x, float* const* y, float* const* param, const int64_t* numel,
                       int decimal, qs_stream_t stream) {
    if (n < 0 || (n > 0 && (!x || !y || !param || !numel))) return QS_ERR_ARG;
    for (int base = 0; base < n; base += kMultiMax) {
        MultiTensors a{};
        a.n = std::min(kMultiMax, n - base);
        int64_t blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const int k = base + i;
            if (!x[k] || !y[k] || !param[k] || numel[k] < 0) return QS_ERR_ARG;
            if (!aligned16(x[k]) || !aligned16(y[k])) return QS_ERR_ALIGN;
            a.x[i] = x[k];
            a.y[i] = y[k];
            a.scale[i] = param[k];
            a.numel[i] = numel[k];
            a.block0[i] = (int32_t)blocks;
            blocks += std::max<int64_t>((numel[k] / 8 + kBlock - 1) / kBlock, 1);
            if (blocks > 0x7fffffff) return QS_ERR_ARG;
        }
        a.block0[a.n] = (int32_t)blocks;
        if (decimal)
            hipLaunchKernelGGL((multi_quant_kernel<true>), dim3((int)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL((multi_quant_kernel<false>), dim3((int)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
    }
    return launch_status();
}

}  // extern "C"
