/*
 * qs_oracle.c -- plain-C restatement of the arithmetic on qsparse's quantize/prune path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: built by __graft_entry__.build() into oracle/libqs_oracle.so and
 * loaded only by tests/ (tests/test_oracle_c.py), where it is cross-checked against oracle/qs_oracle.py,
 * which in turn is pinned bit-for-bit to golden vectors recorded from the real reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared qs_oracle.c -o libqs_oracle.so -lm
 * (every operator of the reference chain is one separately rounded binary32 operation).
 *
 * Tensors are contiguous [outer, C, inner] fp32 arrays (bf16/fp16 inputs are widened by the caller, which
 * is exact); a per-channel parameter has C entries, nparam == 1 means tensor-wise.
 * Reference lines (relative to /root/reference) are cited per function.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int64_t chan_of(int64_t e, int64_t C, int64_t inner) { return (e / inner) % C; }

/* ScalerQuantization.forward, qsparse/quantize.py:100-117: q = int(round(x / s)); y = float(q) * s.
 * The clamp at :110-116 acts on a temporary, so nothing saturates. */
void qo_scaler_fwd(const float* x, const float* scale, int64_t nscale, int64_t outer, int64_t C, int64_t inner,
                   float* y, int32_t* codes) {
    const int64_t n = outer * C * inner;
    for (int64_t e = 0; e < n; ++e) {
        const float s = scale[nscale > 1 ? chan_of(e, C, inner) : 0];
        const float q = x[e] / s;             /* true division            (:109) */
        const int32_t qi = (int32_t)rintf(q); /* round half to even, .int()      */
        if (codes) codes[e] = qi;
        y[e] = (float)qi * s;                 /* q.float() * scaler       (:117) */
    }
}

/* DecimalQuantization.forward, qsparse/quantize.py:44-63: q = int(x * 2^d) (truncation); y = float(q) * 2^-d */
void qo_decimal_fwd(const float* x, const float* decimal, int64_t ndec, int64_t outer, int64_t C, int64_t inner,
                    float* y, int32_t* codes) {
    const int64_t n = outer * C * inner;
    for (int64_t e = 0; e < n; ++e) {
        const float d = decimal[ndec > 1 ? chan_of(e, C, inner) : 0];
        const float toi = ldexpf(1.0f, (int)d), tof = ldexpf(1.0f, -(int)d);
        const int32_t qi = (int32_t)(x[e] * toi);
        if (codes) codes[e] = qi;
        y[e] = (float)qi * tof;
    }
}

/* LineQuantization.forward, qsparse/quantize.py:148-181 */
void qo_line_fwd(const float* x, const float* lines, int64_t nlines, int bits, int float_zero_point, int64_t outer,
                 int64_t C, int64_t inner, float* y) {
    const int64_t n = outer * C * inner;
    const float N = (float)(1 << bits);
    for (int64_t e = 0; e < n; ++e) {
        const int64_t c = nlines > 1 ? chan_of(e, C, inner) : 0;
        const float start = lines[2 * c], end = lines[2 * c + 1];
        float xc = x[e] < start ? start : x[e];          /* clamp(x, start, end)   (:158) */
        xc = xc > end ? end : xc;
        float step = (end - start) / N;                  /* (:159) */
        if (step == 0.0f) step = 0.0001f;                /* (:160) */
        if (float_zero_point) {                          /* (:175-181) */
            float t = xc - start;
            t = t / step;
            t = rintf(t);
            t = t < 0.0f ? 0.0f : (t > N - 1.0f ? N - 1.0f : t);
            t = t * step;
            y[e] = t + start;
        } else {                                         /* (:161-166) */
            float qa = rintf(xc / step);
            const float qs = rintf(start / step);
            qa = qa - qs;
            qa = qa < 0.0f ? 0.0f : (qa > N - 1.0f ? N - 1.0f : qa);
            y[e] = (qa + qs) * step;
        }
    }
}

/* Scaler/DecimalQuantization.backward, qsparse/quantize.py:66-77,120-131: gradient VALUES clamped into
 * [lo_mul*s, hi_mul*s] (the masked assignment at :76/:130 compares the tensor with itself: NaNs -> 0); optional channel mask = the
 * PruneLayer backward g * mask. */
void qo_ste_bwd(const float* g, const float* step, int64_t nstep, float lo_mul, float hi_mul, const uint8_t* mask,
                int64_t outer, int64_t C, int64_t inner, float* gx) {
    const int64_t n = outer * C * inner;
    for (int64_t e = 0; e < n; ++e) {
        const int64_t c = chan_of(e, C, inner);
        const float s = step[nstep > 1 ? c : 0];
        const float lo = lo_mul * s, hi = hi_mul * s;
        float v = g[e] < lo ? lo : g[e];
        v = v > hi ? hi : v;
        /* a NaN bound (a NaN scale): ATen's clamp with tensor bounds returns NaN for every g, and a NaN g stays NaN -- then the
         * masked assignment at :76/:130, which compares the clamped tensor with ITSELF, turns exactly those NaNs into +0.0 */
        if (lo != lo || hi != hi || v != v) v = 0.0f;
        if (mask) v = v * (mask[c] ? 1.0f : 0.0f);
        gx[e] = v;
    }
}

/* x * mask with a per-channel mask, qsparse/sparse.py:66,116,122,263 */
void qo_mask_apply(const float* x, const uint8_t* mask, int64_t outer, int64_t C, int64_t inner, float* y) {
    const int64_t n = outer * C * inner;
    for (int64_t e = 0; e < n; ++e) y[e] = x[e] * (mask[chan_of(e, C, inner)] ? 1.0f : 0.0f);
}

/* max|x| per channel or over the tensor, qsparse/quantize.py:329-340 */
void qo_absmax(const float* x, int per_channel, int64_t outer, int64_t C, int64_t inner, float* out) {
    const int64_t n = outer * C * inner, nout = per_channel ? C : 1;
    for (int64_t i = 0; i < nout; ++i) out[i] = 0.0f;
    for (int64_t e = 0; e < n; ++e) {
        const float a = fabsf(x[e]);
        float* o = out + (per_channel ? chan_of(e, C, inner) : 0);
        if (a > *o) *o = a;
    }
}

/* ---- staged mean: ATen's CPU summation order (aten/src/ATen/native/cpu/SumKernel.cpp, fp32, 8-lane build) --- */
static int ceil_log2(int64_t x) {
    int l = 0;
    while (((int64_t)1 << l) < x) ++l;
    return x <= 1 ? 0 : l;
}
/* "multi-row" order over elements base[i*stride], i < n */
static float sum_multi_row(const float* base, int64_t stride, int64_t n) {
    int lp = ceil_log2(n) / 4;
    if (lp < 4) lp = 4;
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = 0;
    while (i + step <= n) {
        for (int64_t j = 0; j < step; ++j, ++i) acc[0] += base[i * stride];
        for (int j = 1; j < 4; ++j) {
            acc[j] += acc[j - 1];
            acc[j - 1] = 0.f;
            if ((i & (lmask << (j * lp))) != 0) break;
        }
    }
    for (; i < n; ++i) acc[0] += base[i * stride];
    for (int j = 1; j < 4; ++j) acc[0] += acc[j];
    return acc[0];
}
/* "row-sum" order: 4 interleaved multi-row partials */
static float sum_row_sum(const float* base, int64_t stride, int64_t n) {
    const int64_t n4 = n / 4;
    float p[4];
    for (int k = 0; k < 4; ++k) p[k] = sum_multi_row(base + k * stride, 4 * stride, n4);
    for (int64_t i = n4 * 4; i < n; ++i) p[0] += base[i * stride];
    for (int k = 1; k < 4; ++k) p[0] += p[k];
    return p[0];
}
/* One squeeze stage (qsparse/util.py:92-99): x [pre, n, post] -> out [pre, post] = sum / n in fp32.
 * The caller rounds `out` to the tensor dtype (bf16 inputs) before the next stage. */
void qo_mean_dim(const float* x, int64_t pre, int64_t n, int64_t post, float* out) {
    for (int64_t p = 0; p < pre; ++p) {
        const float* xs = x + p * n * post;
        if (post == 1) {
            float s;
            if (n >= 8) {
                const int64_t nv = n / 8;
                float fin = 0.f, lanes[8];
                for (int k = 0; k < 8; ++k) lanes[k] = sum_row_sum(xs + k, 8, nv);
                for (int64_t i = nv * 8; i < n; ++i) fin += xs[i];
                for (int k = 0; k < 8; ++k) fin += lanes[k];
                s = fin;
            } else {
                s = sum_row_sum(xs, 1, n);
            }
            out[p] = s / (float)n;
        } else {
            const int64_t mr = post >= 8 ? (post / 32) * 32 : (post / 4) * 4;
            for (int64_t c = 0; c < post; ++c) {
                const float s = c < mr ? sum_multi_row(xs + c, post, n) : sum_row_sum(xs + c, post, n);
                out[p * post + c] = s / (float)n;
            }
        }
    }
}

/* The same stage for a channels_last (NHWC in memory) activation x[n][hw][C] (what `Tensor.mean(0, keepdim=True)` does
 * on such a tensor; qsparse/util.py:92-99 reaches it whenever the network runs in torch.channels_last): the result is
 * NCHW-contiguous, out[c*hw + pos], and ATen (one intra-op thread) sums position `pos` of every channel in multi-row
 * order when pos < 4*floor(hw/4) and in row-sum order otherwise -- for ANY channel count. */
void qo_mean_dim_cl(const float* x, int64_t n, int64_t hw, int64_t C, float* out) {
    const int64_t sample = hw * C, main_pos = (hw / 4) * 4;
    for (int64_t pos = 0; pos < hw; ++pos)
        for (int64_t c = 0; c < C; ++c) {
            const float* base = x + pos * C + c;
            const float s = pos < main_pos ? sum_multi_row(base, sample, n) : sum_row_sum(base, sample, n);
            out[c * hw + pos] = s / (float)n;
        }
}

/* calculate_mask_given_importance, qsparse/util.py:113-117: thr = sort(imp)[k]; mask = imp >= thr */
static int cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    if (isnan(x)) return isnan(y) ? 0 : 1;   /* NaNs sort last, like torch.sort */
    if (isnan(y)) return -1;
    return (x > y) - (x < y);
}
float qo_mask_from_importance(const float* imp, int64_t n, int64_t k, uint8_t* mask) {
    float* v = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(v, imp, sizeof(float) * (size_t)n);
    qsort(v, (size_t)n, sizeof(float), cmp_float);
    const float thr = v[k];
    free(v);
    for (int64_t i = 0; i < n; ++i) mask[i] = imp[i] >= thr ? 1 : 0;
    return thr;
}

/* running means: quantize.py:344-348 and sparse.py:89 */
void qo_running_mean(float* state, const float* nv, int64_t n, int64_t t) {
    for (int64_t i = 0; i < n; ++i) state[i] = ((float)t * state[i] + nv[i]) / (float)(t + 1);
}
