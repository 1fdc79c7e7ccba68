// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), version / status / workspace queries, k-th value + masks, the C-sized select launch, the statistics
// exchange records and the multi-tensor weight path (qs_reduce.h, qs_multi.h).
// Host side: argument checks, geometry, launch configuration.  No allocation, no synchronisation: every entry
// point only enqueues work on the caller's stream.
#include "qs_host.h"
#include "qs_reduce.h"
#include "qs_multi.h"

#include <mutex>

// ---- interned activation descriptors (qs_activation) ----------------------------------------------------------------
namespace {
std::mutex g_act_mutex;
ActSpec g_act_table[256] = {{QS_ACT_NONE, 0.f, 0.f}, {QS_ACT_RELU, 0.f, 0.f}};
int g_act_count = 2;
}  // namespace

int qs_act_resolve(int pre_relu, ActSpec* out) {
    if (pre_relu < 0) return QS_ERR_ARG;
    if (pre_relu < 2) {          // the two built-in descriptors need no lock
        *out = g_act_table[pre_relu];
        return QS_OK;
    }
    std::lock_guard<std::mutex> lock(g_act_mutex);
    if (pre_relu >= g_act_count) return QS_ERR_ARG;
    *out = g_act_table[pre_relu];
    return QS_OK;
}

extern "C" {

int qs_activation(int kind, float a, float b) {
    if (kind == QS_ACT_NONE) return 0;
    if (kind == QS_ACT_RELU) return 1;
    if (kind != QS_ACT_HARDTANH && kind != QS_ACT_LEAKY) return QS_ERR_ARG;
    if (kind == QS_ACT_HARDTANH && !(a <= b)) return QS_ERR_ARG;
    if (kind == QS_ACT_LEAKY) b = 0.f;
    std::lock_guard<std::mutex> lock(g_act_mutex);
    for (int i = 2; i < g_act_count; ++i)
        if (g_act_table[i].kind == kind && g_act_table[i].a == a && g_act_table[i].b == b) return i;
    if (g_act_count >= 256) return QS_ERR_ARG;
    g_act_table[g_act_count] = ActSpec{kind, a, b};
    return g_act_count++;
}

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
                 uint8_t* elide_mask_out, qs_stream_t stream) {
    PqArgs a;
    int st = pq_args(&a, magnitude, C, update_magnitude, t_mag, refresh_mask, k, mask, chan_absmax, chan_absmax_stride, update_scale, t_q, bits,
                     scale, bump_i32_a, bump_i32_b, bump_i64_a, bump_i64_b, t_mag_dev, t_q_dev, gathered, world);
    if (st == QS_OK && update_scale && !dt_ok(stat_dt)) st = QS_ERR_DTYPE;
    a.stat_dt = stat_dt;
    a.elide_mask = update_scale ? elide_mask_out : nullptr;
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

// ---- one activation site per call: the fine-grained entry points in sequence (no arithmetic of its own) --------------
static int site_plan_ok(const qs_site_plan* p) {
    if (!p || p->N < 1 || p->C < 2 || p->H < 1 || p->W < 1) return 0;
    return p->layout == 0 || p->layout == 1 || (p->layout == 2 && p->H == 1 && p->W == 1) || (p->layout == 3 && p->H >= 2 && p->W == 1);
}

// layout 3, token-major [N][T][C] with the mask on the last dim (T = plan->H): squeeze_tensor_to_shape's two stages (util.py:92-99),
// mean over N of the [N][T*C] matrix, then mean over T of the [T][C] stage, each rounded to xdt; channel of column j is j % C
// (qs_token_stats: the abs-max per column into plan->amax_part -- no atomics -- and folded per channel by a third launch)
static int site_token_means(const qs_site_plan* p, const void* x, int mflags, float* absmax, int64_t absmax_stride, qs_stream_t stream) {
    return qs_token_stats(x, p->stage, p->stage_mean, absmax ? p->amax_part : nullptr, absmax, absmax_stride, p->N, p->H, p->C, p->xdt, mflags,
                          stream);
}

// the statistics launches of a live site step; `record` (nullable): the rank's exchange record, written by the last of them
static int site_statistics(const qs_site_plan* p, const void* x, int pre_relu, float* record, qs_stream_t stream) {
    if (!p->chan_absmax || !p->stage_mean || p->absmax_stride < 1 || (p->layout != 2 && !p->stage)) return QS_ERR_ARG;
    const int64_t hw = p->H * p->W;
    const int mflags = QS_MEAN_ABS | (pre_relu ? QS_MEAN_ACT(pre_relu) : 0);
    if (p->layout == 2) {      // [N][C]: one stage, the per-channel abs-max rides in it; the record is a launch of its own
        int st = qs_mean_dim(x, p->stage_mean, 1, p->N, p->C, p->xdt, p->xdt, mflags, nullptr, p->chan_absmax, p->absmax_stride, 1,
                             p->C, stream);
        if (st || !record) return st;
        return qs_stats_pack(p->stage_mean, p->xdt, p->chan_absmax, p->absmax_stride, p->C, record, stream);
    }
    if (p->layout == 3) {
        int st = site_token_means(p, x, mflags, p->chan_absmax, p->absmax_stride, stream);
        if (st || !record) return st;
        return qs_stats_pack(p->stage_mean, p->xdt, p->chan_absmax, p->absmax_stride, p->C, record, stream);
    }
    if (p->layout == 0) {
        int st = qs_mean_dim(x, p->stage, 1, p->N, p->C * hw, p->xdt, p->xdt, mflags, nullptr, p->chan_absmax, p->absmax_stride,
                             hw, p->C, stream);
        if (st) return st;
        // (the accumulator is complete when this launch starts: with `record` it is read for the record's second half)
        return qs_mean_last2(p->stage, p->stage_mean, p->C, p->H, p->W, p->xdt, p->xdt, nullptr, record ? p->chan_absmax : nullptr,
                             record ? p->absmax_stride : 1, record, stream);
    }
    if (!p->amax_part) return QS_ERR_ARG;
    int st = qs_mean_dim_cl(x, p->stage, p->N, hw, p->C, p->xdt, p->xdt, mflags, nullptr, p->amax_part, stream);
    if (st) return st;
    return qs_mean_last2(p->stage, p->stage_mean, p->C, p->H, p->W, p->xdt, p->xdt, p->amax_part, p->chan_absmax, p->absmax_stride,
                         record, stream);
}

// a PruneLayer alone (QS_SITE_NO_QUANT): geometry of plan as qs_mask_apply wants it
static int site_mask_apply(const qs_site_plan* p, const void* x, void* y, int dt, int pre_relu, int elide, uint8_t* gate_out,
                           qs_stream_t stream) {
    const int64_t hw = p->H * p->W;
    int64_t sizes[3], strides[3];
    int nd;
    if (p->layout == 0) {
        sizes[0] = p->N, sizes[1] = p->C, sizes[2] = hw;
        strides[0] = 0, strides[1] = 1, strides[2] = 0;
        nd = 3;
    } else {
        sizes[0] = p->N * hw, sizes[1] = p->C;
        strides[0] = 0, strides[1] = 1;
        nd = 2;
    }
    return qs_mask_apply(x, p->mask, y, nd, sizes, strides, dt, pre_relu, elide, gate_out, stream);
}

static int site_prune_only_fwd(const qs_site_plan* p, const void* x, void* y, uint8_t* gate_out, int flags, int64_t t_mag, int64_t k,
                               int pre_relu, qs_stream_t stream) {
    if (!p->mask || !p->magnitude || (flags & (QS_SITE_STATS_DONE | QS_SITE_SCALE_ONLY | QS_SITE_NO_MASK))) return QS_ERR_ARG;
    const int64_t hw = p->H * p->W;
    const int update = (flags & QS_SITE_LIVE) ? 1 : 0, refresh = (flags & QS_SITE_REFRESH) ? 1 : 0;
    if (update) {           // the staged mean of |act?(x)| alone: no abs-max rides along
        if (!p->stage_mean || (p->layout != 2 && !p->stage)) return QS_ERR_ARG;
        const int mflags = QS_MEAN_ABS | (pre_relu ? QS_MEAN_ACT(pre_relu) : 0);
        int st;
        if (p->layout == 2) {
            st = qs_mean_dim(x, p->stage_mean, 1, p->N, p->C, p->xdt, p->xdt, mflags, nullptr, nullptr, 1, 1, p->C, stream);
        } else if (p->layout == 3) {
            st = site_token_means(p, x, mflags, nullptr, 1, stream);
        } else {
            st = p->layout == 0 ? qs_mean_dim(x, p->stage, 1, p->N, p->C * hw, p->xdt, p->xdt, mflags, nullptr, nullptr, 1, hw, p->C, stream)
                                : qs_mean_dim_cl(x, p->stage, p->N, hw, p->C, p->xdt, p->xdt, mflags, nullptr, nullptr, stream);
            if (st) return st;
            st = qs_mean_last2(p->stage, p->stage_mean, p->C, p->H, p->W, p->xdt, p->xdt, nullptr, nullptr, 1, nullptr, stream);
        }
        if (st) return st;
    }
    if (update || refresh) {
        int st = qs_pq_select(p->magnitude, update ? p->stage_mean : nullptr, p->xdt, p->C, update, t_mag, refresh, k, p->mask, nullptr, 1,
                              0, 0, 8, nullptr, p->prune_n_updates, nullptr, p->callback_t, nullptr,
                              (update && p->callback_t_from_device) ? p->callback_t : nullptr, nullptr, p->xdt, nullptr, 1, nullptr, stream);
        if (st) return st;
    }
    return site_mask_apply(p, x, y, p->xdt, pre_relu, (flags & QS_SITE_ELIDE) ? 1 : 0, gate_out, stream);
}

int qs_site_stats(const qs_site_plan* p, const void* x, int flags, float* record, qs_stream_t stream) {
    if (!site_plan_ok(p) || !x || !record) return QS_ERR_ARG;
    return site_statistics(p, x, (flags & QS_SITE_PRE_RELU) ? (p->act > 0 ? p->act : 1) : 0, record, stream);
}

int qs_site_fwd(const qs_site_plan* p, const void* x, void* y, uint8_t* gate_out, int flags, int64_t t_mag, int64_t k,
                int64_t t_q, void* image_out, int imgdt, const float* gathered, int world, void* xback_out, float* decimal,
                qs_stream_t stream) {
    if (!site_plan_ok(p) || !x || !y) return QS_ERR_ARG;
    const int pre_relu = (flags & QS_SITE_PRE_RELU) ? (p->act > 0 ? p->act : 1) : 0;     // the folded activation's handle
    if (flags & QS_SITE_NO_QUANT) {
        if (image_out || xback_out || decimal || gathered) return QS_ERR_ARG;
        return site_prune_only_fwd(p, x, y, gate_out, flags, t_mag, k, pre_relu, stream);
    }
    if (!p->mask || !p->scale) return QS_ERR_ARG;
    const int64_t hw = p->H * p->W;
    // the apply launch's operands, fixed by the plan and the flags alone: validated BEFORE the statistics and select launches
    // advance magnitude, mask, scale and counters -- a routing the apply kernels do not serve must not leave the state one step
    // ahead of a call that failed
    const uint8_t* cm = (flags & QS_SITE_NO_MASK) ? nullptr : p->mask;
    const int64_t outer = p->layout == 0 ? p->N : p->N * hw, inner = p->layout == 0 ? hw : 1;
    const int64_t o = cm ? outer : 1, c = cm ? p->C : 1, in = cm ? inner : outer * p->C * inner;
    const int elide = (cm && (flags & QS_SITE_ELIDE)) ? 1 : 0;
    // a step with statistics elides through the select's elision mask: pruned channels that hold a NaN / Inf are loaded, so the
    // result is the loading path's for every input; an eliding step without statistics (the caller's choice) has only the mask
    if (elide && (flags & QS_SITE_LIVE) && p->elide_mask) cm = p->elide_mask;
    if (image_out || xback_out) {
        if (!gate_out || p->ydt != QS_F32 || (xback_out && (!pre_relu || (((uintptr_t)xback_out) & 15u))) ||
            (image_out && ((imgdt != QS_BF16 && imgdt != QS_F16) || (((uintptr_t)image_out) & 15u))) ||
            !qs_quant_image_ok(o, c, in, 0, cm != nullptr, (((uintptr_t)cm) & 7u) == 0, p->xdt))
            return QS_ERR_ARG;
    }
    if (flags & QS_SITE_LIVE) {
        if (flags & QS_SITE_NO_MASK) return QS_ERR_ARG;
        if (flags & QS_SITE_SCALE_ONLY) {      // frozen mask: per-channel abs-max, then the select's scale half alone
            if (!p->absmax_dense || gathered || (flags & (QS_SITE_REFRESH | QS_SITE_STATS_DONE))) return QS_ERR_ARG;
            const int64_t so = p->layout == 0 ? p->N : p->N * hw, si = p->layout == 0 ? hw : 1;
            int st = qs_absmax(x, p->absmax_dense, 1, so, p->C, si, p->xdt, 1, pre_relu, 1, p->reduce_ws,
                               (size_t)p->reduce_ws_bytes, stream);
            if (st) return st;
            st = qs_pq_select(p->magnitude, nullptr, p->xdt, p->C, 0, t_mag, 0, 0, p->mask, p->absmax_dense, 1, 1, t_q, p->bits,
                              p->scale, p->prune_n_updates, p->quant_n_updates, p->callback_t, p->quantizer_t_dev, nullptr,
                              p->quantizer_t_dev, p->xdt, nullptr, 1, p->elide_mask, stream);
            if (st) return st;
        } else {
        if (!p->magnitude || !p->chan_absmax || !p->stage_mean || p->absmax_stride < 1) return QS_ERR_ARG;
        if ((flags & QS_SITE_STATS_DONE) ? (!gathered || world < 1) : (gathered != nullptr)) return QS_ERR_ARG;
        int st;
        if (!(flags & QS_SITE_STATS_DONE)) {
            st = site_statistics(p, x, pre_relu, nullptr, stream);
            if (st) return st;
        }
        st = qs_pq_select(p->magnitude, p->stage_mean, p->xdt, p->C, 1, t_mag, (flags & QS_SITE_REFRESH) ? 1 : 0, k, p->mask,
                          p->chan_absmax, p->absmax_stride, 1, t_q, p->bits, p->scale, p->prune_n_updates, p->quant_n_updates,
                          p->callback_t, p->quantizer_t_dev, p->callback_t_from_device ? p->callback_t : nullptr,
                          p->quantizer_t_dev, p->xdt, gathered, gathered ? world : 1, p->elide_mask, stream);
        if (st) return st;
        }
    }
    if (decimal) {             // DecimalQuantizer: the power-of-two step of THIS call's scale (quantize.py:316)
        int st = qs_decimal_from_scale(p->scale, decimal, 1, stream);
        if (st) return st;
        return qs_quant_decimal_fwd(x, y, nullptr, decimal, 1, 0.0f, cm, o, c, in, p->xdt, p->ydt, QS_F32, p->saturate, p->code_lo,
                                    p->code_hi, pre_relu, elide, gate_out, image_out, imgdt, xback_out, stream);
    }
    return qs_quant_scaler_fwd(x, y, nullptr, p->scale, 1, 0.0f, cm, o, c, in, p->xdt, p->ydt, QS_F32, p->saturate, p->code_lo,
                               p->code_hi, pre_relu, elide, gate_out, image_out, imgdt, xback_out, stream);
}

static int site_bwd_impl(const qs_site_plan* p, const qs_site_bwd_args& a) {
    const void *g = a.g, *g2 = a.g2;
    const uint8_t* gate = a.gate;
    void* gx = a.gx;
    const int flags = a.flags, gdt = a.gdt;
    const float* decimal = a.decimal;
    qs_stream_t stream = a.stream;
    const bool dact = a.act_x != nullptr || a.act_x_kind != 0;      // (v26) the caller's activation: its input replaces the gate
    if (!p || (!g && !g2) || !gx || p->N < 1 || p->C < 1 || p->H < 1 || p->W < 1 || (g2 && !gate && !dact)) return QS_ERR_ARG;
    if (dact && (!a.act_x || (flags & QS_SITE_NO_QUANT))) return QS_ERR_ARG;
    if ((a.g3 || a.gx_image) && ((!gate && !dact) || (flags & QS_SITE_NO_QUANT))) return QS_ERR_ARG;      // riders of the gated quantizer backward
    if (flags & QS_SITE_NO_QUANT) {          // a PruneLayer alone: the backward of act?(x) * mask
        if (!g || !p->mask || g2 || decimal) return QS_ERR_ARG;
        const int64_t hw0 = p->H * p->W;
        const int64_t po = p->layout == 0 ? p->N : p->N * hw0, pi = p->layout == 0 ? hw0 : 1;
        const int el = (flags & QS_SITE_ELIDE) ? 1 : 0;
        if (gate)
            return qs_quant_ste_relu_bwd(g, nullptr, gate, gx, nullptr, 1, 1.0f, 0, -__builtin_inff(), __builtin_inff(), p->mask, po, p->C,
                                         pi, gdt, p->xdt, el, p->act > 0 ? p->act : 1, nullptr, 0, stream);
        return site_mask_apply(p, g, gx, gdt, 0, el, nullptr, stream);
    }
    const float* step = decimal ? decimal : p->scale;
    const int is_decimal = decimal ? 1 : 0;
    const int64_t hw = p->H * p->W;
    const uint8_t* cm = (flags & QS_SITE_NO_MASK) ? nullptr : p->mask;
    const int64_t outer = p->layout == 0 ? p->N : p->N * hw, inner = p->layout == 0 ? hw : 1;
    const int64_t o = cm ? outer : 1, c = cm ? p->C : 1, in = cm ? inner : outer * p->C * inner;
    const int elide = (cm && (flags & QS_SITE_ELIDE)) ? 1 : 0;
    if (gate || dact) {
        qs_ste_relu_bwd_args r{};
        r.struct_size = sizeof(r);
        r.gdt = gdt, r.xdt = p->xdt, r.g2dt = a.g2dt;
        r.g = g, r.gate = dact ? nullptr : gate, r.gx = gx;
        r.act_x = a.act_x, r.act_x_kind = a.act_x_kind;
        r.step = step, r.nstep = 1, r.step_host = 0.0f, r.step_is_decimal = is_decimal;
        r.lo_mul = a.lo_mul, r.hi_mul = a.hi_mul, r.chan_mask = cm;
        r.outer = o, r.C = c, r.inner = in;
        r.elide_masked = g2 ? 0 : elide, r.act = p->act > 0 ? p->act : 1, r.g2 = g2, r.stream = stream;
        r.g3 = a.g3, r.gx_image = a.gx_image, r.gx_image_dt = a.gx_image_dt;
        return qs_quant_ste_relu_bwd_v(&r);
    }
    return qs_quant_ste_bwd(g, gx, step, 1, 0.0f, is_decimal, a.lo_mul, a.hi_mul, 0, cm, o, c, in, gdt, p->xdt, elide, stream);
}

int qs_site_bwd_v(const qs_site_plan* p, const qs_site_bwd_args* args) {
    qs_site_bwd_args a;
    if (!take_args(args, &a)) return QS_ERR_ARG;
    return site_bwd_impl(p, a);
}

// the positional form (ABI <= v24 callers)
int qs_site_bwd(const qs_site_plan* p, const void* g, const uint8_t* gate, void* gx, int gdt, int flags, float lo_mul,
                float hi_mul, const void* g2, int g2dt, const float* decimal, qs_stream_t stream) {
    qs_site_bwd_args a{};
    a.struct_size = sizeof(a);
    a.flags = flags, a.gdt = gdt, a.g2dt = g2dt;
    a.g = g, a.gate = gate, a.gx = gx, a.lo_mul = lo_mul, a.hi_mul = hi_mul, a.g2 = g2, a.decimal = decimal, a.stream = stream;
    return site_bwd_impl(p, a);
}

int qs_quantize_step(const void* x, void* y, uint8_t* gate_out, float* amax_lines, int lines, float* scale, int64_t numel,
                     int xdt, int ydt, int bits, int64_t t, int64_t* t_dev, int32_t* n_updates, int pre_relu, int update,
                     int saturate, int32_t code_lo, int32_t code_hi, void* xback_out, void* image_out, int imgdt,
                     qs_stream_t stream) {
    if (!x || (!y && update != QS_QSTEP_ABSMAX) || !scale || numel < 0 || update < 0 || update > QS_QSTEP_FINISH) return QS_ERR_ARG;
    // (validated before the statistics launches advance the scale and the counters, as in qs_site_fwd)
    if (image_out && (!gate_out || !pre_relu || ydt != QS_F32 || (imgdt != QS_BF16 && imgdt != QS_F16) || (((uintptr_t)image_out) & 15u) ||
                      !qs_quant_image_ok(1, 1, numel, 0, 0, 1, xdt)))
        return QS_ERR_ARG;
    if (numel == 0) return QS_OK;
    if (update != QS_QSTEP_APPLY) {
        if (!amax_lines) return QS_ERR_ARG;
        if (update != QS_QSTEP_FINISH) {
            int st = qs_absmax(x, amax_lines, 0, 1, 1, numel, xdt, 1, pre_relu, lines, nullptr, 0, stream);
            if (st || update == QS_QSTEP_ABSMAX) return st;
        }
        int st = qs_scale_update(amax_lines, lines, scale, 1, t, t_dev, 1, bits, 1, n_updates, xdt, stream);
        if (st) return st;
    }
    return qs_quant_scaler_fwd(x, y, nullptr, scale, 1, 0.0f, nullptr, 1, 1, numel, xdt, ydt, QS_F32, saturate, code_lo, code_hi,
                               pre_relu, 0, gate_out, image_out, imgdt, xback_out, stream);
}

// ---- multi-tensor weight path (qs_multi.h) ----------------------------------------------------------------------
int qs_multi_plan(qs_multi_row* rows, int n, int* absmax_blocks, int* quant_blocks, int* channels, int* hist_blocks_out) {
    if (n < 0 || (n > 0 && !rows) || !absmax_blocks || !quant_blocks || !channels || !hist_blocks_out) return QS_ERR_ARG;
    int64_t ab = 0, qb = 0, ch = 0, hb = 0;
    for (int i = 0; i < n; ++i) {
        qs_multi_row& r = rows[i];
        if (!r.x || (!r.scale && r.kind == 0) || r.numel < 0 || r.C < 1 || r.outer < 1 || r.inner < 1) return QS_ERR_ARG;
        if (r.kind != 0 && ((r.kind != 1 && r.kind != 2) || r.train)) return QS_ERR_ARG;
        if (r.outer * r.C * r.inner != r.numel && r.numel != 0) return QS_ERR_ARG;
        if (r.C > 1 && (r.inner >= ((int64_t)1 << 31) || r.outer >= ((int64_t)1 << 31))) return QS_ERR_ARG;
        if (r.train && (!r.amax || !r.t_dev || !(r.denom > 0.f))) return QS_ERR_ARG;
        if (r.is_decimal && !r.decimal) return QS_ERR_ARG;
        if (r.mask && r.mask_C != 0 && (r.mask_C < 1 || r.mask_inner < 1)) return QS_ERR_ARG;
        if (r.magnitude && (!r.mag_backup || !r.prune_t)) return QS_ERR_ARG;
        if (r.refresh && (!r.mask || r.mask_C != 0 || !r.select_state || !r.mask_backup || r.numel < 1 ||
                          r.numel >= ((int64_t)1 << 32) || (int64_t)r.select_k >= r.numel))
            return QS_ERR_ARG;
        if ((((uintptr_t)r.x) & 3u) != 0) return QS_ERR_ALIGN;
        r.absmax_block0 = (int32_t)ab;
        r.row_splits = 1;
        r.absmax_blocks = 0;
        if (r.train && r.numel > 0) {
            if (r.C == 1) {
                const int64_t want = (r.numel / 8 + (int64_t)kBlock * 4 - 1) / ((int64_t)kBlock * 4);   // ~4 groups per lane
                r.absmax_blocks = (int32_t)std::min<int64_t>(std::max<int64_t>(want, 1), 64);
            } else {
                const int64_t cols = (int64_t)r.C * r.inner;
                const int cpb = r.outer > 1 ? kMultiTallCols : kMultiFlatCols;
                const int64_t col_blocks = (cols + cpb - 1) / cpb;
                // tall and narrow matrices (a 1x1 convolution's [Cout, Cin]): interleaved row sets on more workgroups
                int splits = 1;
                while (splits < 16 && r.outer / (splits * 2 * (kBlock / 64)) >= 64 && col_blocks * splits * 2 <= 512) splits *= 2;
                r.row_splits = splits;
                r.absmax_blocks = (int32_t)std::min<int64_t>(col_blocks * splits, 0x3fffffff);
            }
        }
        ab += r.absmax_blocks;
        r.quant_block0 = (int32_t)qb;
        qb += std::max<int64_t>((r.numel + 8 * kBlock - 1) / (8 * kBlock), 1);      // 8 elements per lane
        r.chan0 = (int32_t)ch;
        ch += r.C;
        r.hist_block0 = (int32_t)hb;
        r.hist_blocks = 0;
        if (r.refresh) {                  // ~32 elements per lane, at most 256 workgroups per tensor
            const int64_t want = (r.numel + (int64_t)kBlock * 32 - 1) / ((int64_t)kBlock * 32);
            r.hist_blocks = (int32_t)std::min<int64_t>(std::max<int64_t>(want, 1), 256);
        }
        hb += r.hist_blocks;
        if (ab > 0x7fffffff || qb > 0x7fffffff || ch > 0x7fffffff || hb > 0x7fffffff) return QS_ERR_ARG;
    }
    *absmax_blocks = (int)ab;
    *quant_blocks = (int)qb;
    *channels = (int)ch;
    *hist_blocks_out = (int)hb;
    return QS_OK;
}

int qs_multi_absmax(const qs_multi_row* rows_dev, int n, int absmax_blocks, qs_stream_t stream) {
    if (n < 0 || absmax_blocks < 0 || (n > 0 && !rows_dev)) return QS_ERR_ARG;
    if (n == 0 || absmax_blocks == 0) return QS_OK;
    hipLaunchKernelGGL(multi_absmax_kernel, dim3(absmax_blocks), dim3(kBlock), 0, (hipStream_t)stream, rows_dev, n);
    return launch_status();
}

int qs_multi_scale_update(const qs_multi_row* rows_dev, int n, int channels, qs_stream_t stream) {
    if (n < 0 || channels < 0 || (n > 0 && !rows_dev)) return QS_ERR_ARG;
    if (n == 0 || channels == 0) return QS_OK;
    hipLaunchKernelGGL(multi_scale_update_kernel, dim3((channels + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       rows_dev, n, channels);
    return launch_status();
}

int qs_multi_quant_fwd(const qs_multi_row* rows_dev, int n, int quant_blocks, float* ybase, int advance, qs_stream_t stream) {
    if (n < 0 || quant_blocks < 0 || (n > 0 && (!rows_dev || !ybase))) return QS_ERR_ARG;
    if (n == 0 || quant_blocks == 0) return QS_OK;
    if (!aligned16(ybase)) return QS_ERR_ALIGN;
    hipLaunchKernelGGL(multi_quant_kernel, dim3(quant_blocks), dim3(kBlock), 0, (hipStream_t)stream, rows_dev, n, ybase, advance);
    return launch_status();
}

int qs_multi_magnitude(const qs_multi_row* rows_dev, int n, int quant_blocks, qs_stream_t stream) {
    if (n < 0 || quant_blocks < 0 || (n > 0 && !rows_dev)) return QS_ERR_ARG;
    if (n == 0 || quant_blocks == 0) return QS_OK;
    hipLaunchKernelGGL(multi_magnitude_kernel, dim3(quant_blocks), dim3(kBlock), 0, (hipStream_t)stream, rows_dev, n);
    return launch_status();
}

int qs_multi_stage_plan(qs_multi_stage* stages, int n, int* blocks) {
    if (n < 0 || (n > 0 && !stages) || !blocks) return QS_ERR_ARG;
    int64_t b = 0;
    for (int i = 0; i < n; ++i) {
        qs_multi_stage& st = stages[i];
        if (!st.x || !st.out || st.pre < 1 || st.n < 1 || st.post < 1 || (st.layout != 0 && st.layout != 1)) return QS_ERR_ARG;
        st.block0 = (int32_t)b;
        b += (st.pre * st.post + kBlock - 1) / kBlock;
        if (b > 0x7fffffff) return QS_ERR_ARG;
    }
    *blocks = (int)b;
    return QS_OK;
}

int qs_multi_stage_mean(const qs_multi_stage* stages_dev, int n, int blocks, qs_stream_t stream) {
    if (n < 0 || blocks < 0 || (n > 0 && !stages_dev)) return QS_ERR_ARG;
    if (n == 0 || blocks == 0) return QS_OK;
    hipLaunchKernelGGL(multi_stage_mean_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, stages_dev, n);
    return launch_status();
}

int qs_multi_mask_refresh(const qs_multi_row* rows_dev, int n, int hist_blocks, int quant_blocks, qs_stream_t stream) {
    if (n < 0 || hist_blocks < 0 || quant_blocks < 0 || (n > 0 && !rows_dev)) return QS_ERR_ARG;
    if (n == 0 || hist_blocks == 0) return QS_OK;
    hipStream_t s = (hipStream_t)stream;
    for (int pass = 3; pass >= 0; --pass) {
        hipLaunchKernelGGL(multi_select_hist_kernel, dim3(hist_blocks), dim3(kBlock), 0, s, rows_dev, n, pass);
        hipLaunchKernelGGL(multi_select_scan_kernel, dim3(n), dim3(256), 0, s, rows_dev, pass);
    }
    hipLaunchKernelGGL(multi_mask_ge_kernel, dim3(quant_blocks), dim3(kBlock), 0, s, rows_dev, n);
    return launch_status();
}

int qs_multi_ste_bwd(int n, const float* const* g, float* const* gx, float* const* step, const int64_t* numel,
                     const int32_t* C, const int64_t* inner, const float* lo_mul, const float* hi_mul, int step_is_decimal,
                     const uint8_t* const* mask, const int32_t* mask_C, const int64_t* mask_inner, qs_stream_t stream) {
    if (n < 0 || (n > 0 && (!g || !gx || !step || !numel || !lo_mul || !hi_mul)) || ((C == nullptr) != (inner == nullptr)))
        return QS_ERR_ARG;
    for (int base = 0; base < n; base += kMultiSteMax) {
        MultiSte a{};
        a.n = std::min(kMultiSteMax, n - base);
        int64_t blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const int k = base + i;
            if (!g[k] || !gx[k] || !step[k] || numel[k] < 0) return QS_ERR_ARG;
            if (!aligned16(g[k]) || !aligned16(gx[k])) return QS_ERR_ALIGN;
            a.g[i] = g[k];
            a.gx[i] = gx[k];
            a.step[i] = step[k];
            a.numel[i] = numel[k];
            a.C[i] = C ? C[k] : 1;
            if (a.C[i] < 1 || (a.C[i] > 1 && (inner[k] < 1 || inner[k] >= ((int64_t)1 << 31)))) return QS_ERR_ARG;
            a.inner[i] = a.C[i] > 1 ? (int32_t)inner[k] : 1;
            a.lo_mul[i] = lo_mul[k];
            a.hi_mul[i] = hi_mul[k];
            a.mask[i] = mask ? mask[k] : nullptr;
            a.mask_C[i] = (a.mask[i] && mask_C) ? mask_C[k] : 0;
            a.mask_inner[i] = (a.mask[i] && mask_inner) ? mask_inner[k] : 1;
            if (a.mask[i] && a.mask_C[i] != 0 && (a.mask_C[i] < 1 || a.mask_inner[i] < 1)) return QS_ERR_ARG;
            a.block0[i] = (int32_t)blocks;
            blocks += std::max<int64_t>((numel[k] + 8 * kBlock - 1) / (8 * kBlock), 1);
            if (blocks > 0x7fffffff) return QS_ERR_ARG;
        }
        a.block0[a.n] = (int32_t)blocks;
        if (step_is_decimal)
            hipLaunchKernelGGL((multi_ste_kernel<true>), dim3((int)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL((multi_ste_kernel<false>), dim3((int)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
    }
    return launch_status();
}

}  // extern "C"
