/*
 * qsparse_hip.h -- C ABI of libqsparse_hip.so: the MI355X (gfx950) implementation of the tensor math
 * behind mlzxy/qsparse's QuantizeLayer / PruneLayer forward + backward.
 *
 * This is the drop-in boundary.  The reference has no native layer (its "kernels" are chains of eager
 * ATen operators inside qsparse/quantize.py, qsparse/sparse.py and qsparse/util.py); every entry point
 * below names the reference lines whose arithmetic it replaces.  A binding needs nothing but a dlopen:
 * plain pointers, sizes and enums, no torch types.
 *
 * Conventions
 *   - Every tensor is CONTIGUOUS and described as a 3-d view [outer, C, inner]: `C` is the extent of the
 *     reference's `channel_index` dimension, `outer`/`inner` the products of the extents before/after it.
 *     Tensor-wise operation: pass nparam == 1 (C and inner may then be any factorisation of numel).
 *   - Per-channel parameters are DEVICE float arrays of length nparam (1 or C).  A NULL parameter
 *     pointer selects the by-value host scalar that follows it in the argument list.
 *   - dtypes: QS_F32 / QS_BF16 / QS_F16.  Arithmetic is IEEE binary32, one rounding per operator of the
 *     reference chain (built with -ffp-contract=off; division is correctly rounded).
 *   - All work is enqueued on `stream` (a hipStream_t); no call allocates, frees or synchronises, so
 *     every call is hipGraph-capturable.  Scratch memory is caller-provided (qs_workspace_bytes).
 *   - Return value: 0 on success, a positive hipError_t from the runtime, or a negative QS_ERR_* for a
 *     rejected argument (nothing is enqueued in that case).  qs_status_string() explains either kind.
 *   - Data pointers of x / y / g must be 16-byte aligned (QS_ERR_ALIGN otherwise).
 */
#ifndef QSPARSE_HIP_H
#define QSPARSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QS_ABI_VERSION 26
/* ABI compatibility (from v25 on).
 *   - Every positional prototype in this header is FROZEN as of v25: a later version never changes the argument list of an
 *     existing symbol.  qs_abi_floor() returns the oldest version whose prototypes this library still honours (25); a binding
 *     written against version V works with any library with qs_abi_floor() <= V <= qs_version().
 *   - New operands arrive through the DESCRIPTOR entry points (qs_*_v): the operands of a call in a struct whose first field,
 *     `struct_size`, is sizeof(the struct) AS THE CALLER COMPILED IT.  Fields are only ever appended.  The library copies
 *     min(struct_size, its own sizeof) bytes into a zeroed struct: a caller built against an older header leaves the newer fields
 *     0 / NULL (every appended field is optional with that default), a caller built against a newer header is understood up to
 *     what this library knows.  The fields are the positional arguments of the entry point of the same name, one for one, followed
 *     by the appended ones, each marked with the version that introduced it.
 *   - The positional entry points remain: thin wrappers that fill the descriptor.  qs_site_plan / qs_multi_row / qs_multi_stage
 *     are caller-built tables whose layout is likewise append-only from v25 on. */

enum qs_dtype { QS_F32 = 0, QS_BF16 = 1, QS_F16 = 2 };

enum qs_error {
    QS_OK = 0,
    QS_ERR_DTYPE = -1,     /* unsupported dtype combination              */
    QS_ERR_ARG = -2,       /* inconsistent sizes / NULL where not allowed */
    QS_ERR_ALIGN = -3,     /* data pointer not 16-byte aligned           */
    QS_ERR_WORKSPACE = -4, /* workspace too small                        */
    QS_ERR_RANK = -5       /* broadcast pattern needs more than QS_MAX_DIMS collapsed dims */
};

enum qs_workspace_op { QS_WS_KTH_VALUE = 1, QS_WS_REDUCE = 2 /* n = C * inner of a per-channel qs_absmax / qs_minmax */ };

#define QS_MAX_DIMS 6

typedef void* qs_stream_t; /* hipStream_t */

/* Folded activations.  convert() wraps whatever activation modules the user names (qsparse/convert.py:214-218); the entry
 * points below that take `pre_relu` -- and qs_mean_dim / qs_mean_dim_cl through their flags -- absorb the activation in front
 * of an operator: its output is never written, statistics and forward read its input, the forward records the ONE bit per
 * element its backward needs (gate_out) and qs_quant_ste_relu_bwd applies it.  `pre_relu` is 0 (none), 1 (nn.ReLU) or a handle
 * from qs_activation() for the others:
 *   QS_ACT_HARDTANH  clamp(x, a, b): nn.Hardtanh(a, b), nn.ReLU6 (a = 0, b = 6);  backward 0 where x <= a or x >= b
 *   QS_ACT_LEAKY     x > 0 ? x : x*a (product rounded to x's dtype): nn.LeakyReLU(a);  backward g*a where x <= 0
 * Handles are interned descriptors: small positive integers, valid for the life of the process, the same (kind, a, b) always
 * yields the same handle; thread-safe.  Returns a negative QS_ERR_* for an unknown kind or a full table (256 entries). */
enum qs_act_kind { QS_ACT_NONE = 0, QS_ACT_RELU = 1, QS_ACT_HARDTANH = 2, QS_ACT_LEAKY = 3 };
int qs_activation(int kind, float a, float b);

int qs_version(void);
int qs_abi_floor(void);      /* (v25) see "ABI compatibility" above */
const char* qs_status_string(int status);
size_t qs_workspace_bytes(int op, int64_t n);

/* ---- quantizers: forward ------------------------------------------------------------------------ */

/* ScalerQuantization.forward, qsparse/quantize.py:100-117.
 *   q = int32(rint(round_to(qdt, f32(x) / s_c)));  y = f32(q) * s_c
 * qdt is the dtype the reference's quotient tensor has (f32 when the scaler is a >=1-d tensor, the input
 * dtype when it is a Python float / 0-d tensor).  `codes` (nullable) receives q.  `chan_mask` (nullable,
 * nparam-independent, length C, one byte per channel) fuses a preceding channel PruneLayer
 * (x * mask, qsparse/sparse.py:116): masked channels are quantised as x*0.  The reference never
 * saturates (its clamp at :110-116 acts on a temporary, `q.float().clamp_(...)`, and is lost); saturate != 0 is that
 * clamp with the assignment it lacks -- q = clamp(q, code_lo, code_hi) on the int32 codes, i.e. after the float -> int
 * conversion, so a NaN input (INT_MIN) lands on code_lo -- as an explicit opt-in: code_lo / code_hi = -2^(bits-1)+notch /
 * 2^(bits-1)-1+notch, or 0 / 2^bits-1 with use_uint (:111-116).  pre_relu != 0 quantises max(x, 0): the nn.ReLU that convert() finds in front of the pair
 * (qsparse/convert.py:214-218) folded into the same pass.
 * elide_masked != 0 (with chan_mask): the x of a pruned channel is not loaded at all -- it only ever meets `* 0`
 * (sparse.py:116) -- and the quantizer is applied to +0.0 instead: bit-identical to the loading path for every finite
 * x; a NaN / Inf on a pruned channel yields f32(0)*s instead of the reference's f32(INT_MIN)*s.  chan_mask bytes are then
 * read as an ELISION MASK: 1 kept, 0 pruned and skipped, 2 pruned but LOADED (a zero factor like 0, so NaN / Inf come out as in
 * the reference) -- qs_pq_select's elide_mask_out marks 2 the pruned channels whose abs-max of this step is not finite, which
 * makes the eliding call bit-identical to the loading one for every x.
 * gate_out (nullable; needs pre_relu != 0; ceil(numel / 8) bytes): the folded ReLU's gate for the backward, one BIT per
 * element in memory order -- bit (e & 7) of gate_out[e >> 3] = !(x[e] <= 0), ATen's threshold_backward -- so that
 * qs_quant_ste_relu_bwd reads one bit instead of x per element and x need not be kept.  Every element is loaded then;
 * with elide_masked != 0 the x of a pruned channel still counts as +0.0, i.e. the result is the eliding call's. */
int qs_quant_scaler_fwd(const void* x, void* y, int32_t* codes,
                        const float* scale, int64_t nscale, float scale_host,
                        const uint8_t* chan_mask,
                        int64_t outer, int64_t C, int64_t inner,
                        int xdt, int ydt, int qdt,
                        int saturate, int32_t code_lo, int32_t code_hi, int pre_relu, int elide_masked,
                        uint8_t* gate_out, void* image_out, int imgdt, void* xback_out, qs_stream_t stream);

/* DecimalQuantization.forward, qsparse/quantize.py:44-63:  q = int32(trunc(x * 2^d)); y = f32(q) * 2^-d */
int qs_quant_decimal_fwd(const void* x, void* y, int32_t* codes,
                         const float* decimal, int64_t ndecimal, float decimal_host,
                         const uint8_t* chan_mask,
                         int64_t outer, int64_t C, int64_t inner,
                         int xdt, int ydt, int qdt,
                         int saturate, int32_t code_lo, int32_t code_hi, int pre_relu, int elide_masked,
                         uint8_t* gate_out, void* image_out, int imgdt, void* xback_out, qs_stream_t stream);

/* (ABI v25) qs_quant_scaler_fwd / qs_quant_decimal_fwd as ONE descriptor entry point; `kind` selects the quantizer, `param` /
 * `nparam` / `param_host` are scale / nscale / scale_host or decimal / ndecimal / decimal_host. */
enum qs_quant_kind { QS_QUANT_SCALER = 0, QS_QUANT_DECIMAL = 1 };
typedef struct qs_quant_fwd_args {
    uint32_t struct_size;        /* sizeof(qs_quant_fwd_args) as the caller compiled it */
    int32_t kind;                /* enum qs_quant_kind */
    const void* x;
    void* y;
    int32_t* codes;
    const float* param;
    int64_t nparam;
    float param_host;
    int32_t xdt, ydt, qdt;
    const uint8_t* chan_mask;
    int64_t outer, C, inner;
    int32_t saturate, code_lo, code_hi, pre_relu, elide_masked, imgdt;
    uint8_t* gate_out;
    void* image_out;
    void* xback_out;
    qs_stream_t stream;
} qs_quant_fwd_args;
int qs_quant_fwd_v(const qs_quant_fwd_args* args);

/* image_out (nullable; qs_quant_scaler_fwd / qs_quant_decimal_fwd, with gate_out, ydt == QS_F32 and codes == NULL): the same
 * pass also writes RNE(y) in imgdt (QS_BF16 / QS_F16) -- the low-precision image autocast would make of y in front of a
 * convolution -- at +2 instead of a 6 B/elem cast pass.  Served by the gate-recording widening kernels; qs_quant_image_ok tells
 * whether a geometry is one of theirs (1) or the call would be rejected with QS_ERR_ARG (0). */
int qs_quant_image_ok(int64_t outer, int64_t C, int64_t inner, int per_channel_param, int has_mask, int mask_aligned8, int xdt);

/* xback_out (nullable; same kernels and conditions as image_out -- gate_out, ydt == QS_F32, codes == NULL, a geometry for which
 * qs_quant_image_ok answers 1): the same pass also stores act(x) in xdt at xback_out[e] -- nn.ReLU: ATen's clamp_min(x, 0), NaN
 * and -0.0 pass; any other folded activation likewise.  With xback_out == x this IS the forward of an nn.ReLU(inplace=True) /
 * nn.ReLU6(inplace=True) standing in front of the operator (what
 * torchvision-style networks carry where the reference's convert() puts its operators, qsparse/convert.py:199-229): other
 * holders of x see relu(x) as they must, at the price of one more store (+2 / +4 B/elem) instead of ATen's read + write pass
 * (4 / 8 B/elem).  Every lane reads the elements it rewrites before it rewrites them; no other lane touches them. */

/* LineQuantization.forward, qsparse/quantize.py:148-181.  lines = device float [nlines, 2] = (start, end).
 * float_zero_point != 0: ((clamp(rint((xc-start)/step),0,N-1))*step)+start   (:168-181)
 * float_zero_point == 0: (clamp(rint(xc/step)-rint(start/step),0,N-1)+rint(start/step))*step  (:161-166)
 * `codes` (nullable, int32, x's shape): the level index in [0, N-1] -- the clamp's result, before `* step + start` /
 * `+ rint(start/step)` -- for the integer export (the arithmetic tests/test_quantize.py:73-101 pins for the decimal
 * quantizer); INT32_MIN where x is NaN. */
int qs_quant_line_fwd(const void* x, void* y, int32_t* codes, const float* lines, int64_t nlines, int bits,
                      int float_zero_point, int64_t outer, int64_t C, int64_t inner,
                      int xdt, int ydt, qs_stream_t stream);

/* ---- quantizers: straight-through backward ------------------------------------------------------ */

/* ScalerQuantization.backward / DecimalQuantization.backward, qsparse/quantize.py:66-77,120-131:
 *   gx = cast(gxdt, min(max(g, lo_mul*step_c), hi_mul*step_c))     (passthrough != 0: gx = cast(g))
 * and +0.0 where that clamp is NaN -- a NaN g, or any g once step_c is NaN: the reference's `v[v != grad_output] = 0` (:76,
 * :130) compares the clamped tensor with itself, i.e. zeroes exactly the NaNs (fixture F17).
 * step_c = scale (step_is_decimal == 0) or 2^-d (step_is_decimal != 0);  lo_mul = -2^(bits-1)+notch,
 * hi_mul = 2^(bits-1)-1+notch.  chan_mask (nullable) fuses the PruneLayer backward g*mask
 * (autograd MulBackward0 of qsparse/sparse.py:116).  g and gx may alias.
 * elide_masked != 0 (with chan_mask): the g of a pruned channel is not loaded and gx = +0.0 there; the reference's
 * g*0 carries the sign of clamp(g), so this is numerically equal, not bit-identical (opt-in). */
int qs_quant_ste_bwd(const void* g, void* gx,
                     const float* step, int64_t nstep, float step_host, int step_is_decimal,
                     float lo_mul, float hi_mul, int passthrough,
                     const uint8_t* chan_mask,
                     int64_t outer, int64_t C, int64_t inner,
                     int gdt, int gxdt, int elide_masked, qs_stream_t stream);

/* The same backward with the gate of a folded activation -- `act`: 1 (nn.ReLU, threshold_backward: 0 where x <= 0) or a
 * qs_activation() handle (hardtanh_backward: 0 where x <= a or x >= b; leaky_relu_backward: v * slope where x <= 0, v the
 * clamped, masked gradient rounded to xdt):
 *   gx = cast(xdt, x <= 0 ? 0 : min(max(g, lo_mul*step_c), hi_mul*step_c) * mask_c)                    (nn.ReLU)
 * x is the ReLU's INPUT (dtype xdt); gx has x's dtype.  gate (nullable): the bitmap a forward call recorded through
 * gate_out over the same [outer, C, inner] geometry; when given, x is not read (it may be NULL) and xdt only names the
 * dtype of gx: 4 + 1/8 + sizeof(xdt) bytes per element instead of 4 + 2 * sizeof(xdt).
 * g2 (nullable, dtype g2dt = QS_BF16 / QS_F16, with gate and gdt == QS_F32 and elide_masked == 0): a second gradient of
 * the same geometry that is ADDED to g in float32 before the clamp -- g may then be NULL (g2 alone).  It is autograd's
 * accumulation of the gradients a float32 output receives under autocast (float32 from float32 consumers, bf16 from the
 * convolution that consumed its bf16 image) done inside the kernel: no cast pass, no add pass, 2 instead of 4 bytes read. */
int qs_quant_ste_relu_bwd(const void* g, const void* x, const uint8_t* gate, void* gx,
                          const float* step, int64_t nstep, float step_host, int step_is_decimal,
                          float lo_mul, float hi_mul, const uint8_t* chan_mask,
                          int64_t outer, int64_t C, int64_t inner, int gdt, int xdt, int elide_masked, int act,
                          const void* g2, int g2dt, qs_stream_t stream);

/* (ABI v25) The same entry point with its operands in a size-prefixed descriptor (see "Descriptor entry points" at the top of
 * this header): fields are only ever appended, a caller compiled against an older header sets a smaller struct_size and the
 * fields it does not know read as 0 / NULL.  The fields up to g2dt are the positional arguments above, one for one.
 *   g3      (v25; nullable; dtype g2dt; needs g2, gdt == xdt == QS_F32) a THIRD gradient of the same geometry, added between g
 *           and g2: gx = gate * clamp((g + f32(g3)) + f32(g2)) * mask, each term optional from the left.  Autograd accumulates
 *           the shares of a tensor's consumers in REVERSE order of the consumers' creation: with float32 consumers created
 *           last, the second 2-byte consumer (g3) before them and the first 2-byte consumer (g2) first of all, this is the
 *           reference's grouping bit for bit -- the site in front of a ResNet block whose down-sampling convolution reads the same
 *           activation as the block's first convolution.
 *   gx_image / gx_image_dt  (v25; nullable; QS_BF16 / QS_F16; needs gdt == xdt == QS_F32, 16-byte aligned) RNE(gx), written by the
 *           same pass next to gx (+2 B/elem): the gradient of the 2-byte operand of a type-promoting add in front of the site
 *           (`bn(conv(h)) + identity` under autocast: bf16 + float32 -> float32), which ATen's AddBackward0 produces with a
 *           cast pass of gx of its own (6 B/elem). */
/* (v26) The caller's activation.  convert(..., activation_layers=[nn.GELU]) puts the operators behind an activation the kernels do
 * not fold: its forward value is the caller's (ATen's) pass, the site reads that.  Its BACKWARD needs only the site's input gradient
 * and the activation's input, so the site's backward evaluates it on the way out:
 *   act_x / act_x_kind   (nullable; QS_DACT_GELU: nn.GELU(approximate='none')) the activation's INPUT, of dtype xdt, gx's shape and
 *           layout, 16-byte aligned.  gx = gelu_backward(RNE_xdt(gx_site), act_x) with gx_site the site's own result (clamp(g [+ g3]
 *           [+ g2]) * mask), in ATen's GPU arithmetic bit for bit (float32 opmath, dy * fma(x, pdf, cdf)).  `x` and `gate` are not
 *           read (the site has no folded activation of its own then, or the identity's); `elide_masked` is ignored; g2 is allowed
 *           without a gate. */
enum qs_dact_kind { QS_DACT_NONE = 0, QS_DACT_GELU = 1 };
typedef struct qs_ste_relu_bwd_args {
    uint32_t struct_size;        /* sizeof(qs_ste_relu_bwd_args) as the caller compiled it */
    int32_t gdt, xdt, g2dt;
    const void* g;
    const void* x;
    const uint8_t* gate;
    void* gx;
    const float* step;
    int64_t nstep;
    float step_host;
    int32_t step_is_decimal;
    float lo_mul, hi_mul;
    const uint8_t* chan_mask;
    int64_t outer, C, inner;
    int32_t elide_masked, act;
    const void* g2;
    qs_stream_t stream;
    /* ---- v25 ---- */
    const void* g3;
    void* gx_image;
    int32_t gx_image_dt;
    int32_t reserved0;
    /* ---- v26 ---- */
    const void* act_x;           /* see "the caller's activation" above; NULL: none */
    int32_t act_x_kind;          /* QS_DACT_GELU */
    int32_t reserved1;
} qs_ste_relu_bwd_args;
int qs_quant_ste_relu_bwd_v(const qs_ste_relu_bwd_args* args);

/* ---- statistics ---------------------------------------------------------------------------------- */

/* max |x| over the tensor (per_channel == 0 -> out[1]) or per channel (out[C]);
 * DecimalQuantizer.optimize, qsparse/quantize.py:329-340.  Order-independent, hence bit-exact.
 * accumulate != 0: out is max-accumulated instead of overwritten (the caller keeps it zeroed between steps, e.g.
 * through qs_scale_update's clear_absmax), which saves the initialisation launch.
 * pre_relu != 0: the statistic of max(x, 0) -- a preceding nn.ReLU folded into the quantizer's kernels.
 * out_lines (1 unless per_channel == 0 and accumulate != 0): the tensor-wise maximum is accumulated into `out_lines`
 * partial accumulators, QS_AMAX_LINE_STRIDE floats (one 128-byte line) apart, out[l * QS_AMAX_LINE_STRIDE]; their
 * maximum is the result (qs_scale_update folds them).  Same-address atomics serialise on MI355X (~12 ns each), which
 * caps a one-word reduction at ~256 workgroups; 16 lines lift that cap.
 * ws (nullable): qs_workspace_bytes(QS_WS_REDUCE, C*inner) bytes of scratch; with it a per-channel reduction over few
 * columns and many rows (channels_last activations: inner == 1, C <= 512) runs as two atomics-free stages; 0 bytes means the shape never takes that route. */
int qs_absmax(const void* x, float* out, int per_channel,
              int64_t outer, int64_t C, int64_t inner, int xdt, int accumulate, int pre_relu, int out_lines,
              void* ws, size_t ws_bytes, qs_stream_t stream);

/* min and max of x over the tensor or per channel; AdaptiveQuantizer.optimize, quantize.py:410-418
 * (min over the batch of per-sample minima == global per-channel minimum).
 * accumulate == 0: out_min / out_max receive floats (three launches: key initialisation, reduction, key -> float).
 * accumulate != 0: out_min / out_max are persistent buffers of order-preserving uint32 KEYS, neutral on entry (min keys
 * 0xffffffff, max keys 0), min- / max-accumulated by ONE launch and left as keys: qs_lines_update(from_keys = 1) turns them
 * into floats and makes them neutral again -- two launches per AdaptiveQuantizer step instead of four (ABI v9). */
int qs_minmax(const void* x, float* out_min, float* out_max, int per_channel,
              int64_t outer, int64_t C, int64_t inner, int xdt, int accumulate, void* ws, size_t ws_bytes,
              qs_stream_t stream);

/* Step counters.  The reference keeps them on the host (Python ints / `.item()` reads) and they enter the
 * arithmetic of every running mean.  A by-value kernel argument is frozen when a launch is captured into a
 * hipGraph, so every entry point that takes a counter `t` also takes `t_dev` (nullable): a device-resident
 * int64 holding the same counter, read by the kernel INSTEAD of `t` when non-NULL.  Counters are advanced
 * either by the caller (any stream-ordered increment), by qs_pq_select's bump_* arguments, or -- qs_scale_update,
 * qs_lines_update with advance_t_dev != 0 -- by the kernel itself after every thread has read them (the update then
 * runs as a single workgroup). */

/* weight[i] <- t == 0 ? new : (t*weight[i] + new)/(t+1),  new = absmax[i] / 2^(bits-1) rounded to stat_dt, the
 * dtype of the tensor the abs-max was taken from: the reference divides in that dtype, which matters for fp16,
 * where small maxima underflow into subnormals (quantize.py:340,344-348).  clear_absmax != 0 zeroes absmax[i]
 * after use; bump_i32 (nullable) is a one-element device counter incremented once (QuantizeLayer._n_updates,
 * quantize.py:515).  absmax_lines > 1 (n == 1 only): the tensor-wise abs-max arrives as that many partial accumulators
 * QS_AMAX_LINE_STRIDE floats apart (qs_absmax with out_lines); their maximum is used and all of them are cleared. */
int qs_scale_update(float* absmax, int absmax_lines, float* weight, int64_t n, int64_t t, int64_t* t_dev,
                    int advance_t_dev, int bits, int clear_absmax, int32_t* bump_i32, int stat_dt, qs_stream_t stream);

/* lines[i] <- (lines[i]*(t-1) + (mn[i], mx[i])) / t   with t already incremented (quantize.py:427-430);
 * t_dev holds the counter BEFORE the increment (t = *t_dev + 1).  from_keys != 0: mn / mx hold the keys of
 * qs_minmax(accumulate = 1); they are converted here and reset to the neutral keys. */
int qs_lines_update(float* mn, float* mx, float* lines, int64_t n, int64_t t_after,
                    int64_t* t_dev, int advance_t_dev, int from_keys, qs_stream_t stream);

/* d[i] = rint(log2(nan_to_num(1/scale[i], posinf=1, neginf=1)))  (quantize.py:316) */
int qs_decimal_from_scale(const float* scale, float* decimal, int64_t n, qs_stream_t stream);

/* SUMMATION-ORDER PIN.  The staged-mean entry points below (qs_mean_dim, qs_mean_dim_split, qs_mean_dim_cl, qs_mean_cl_w,
 * qs_mean_strided, qs_mean_last2, qs_multi_stage_mean) restate the summation order of ATen's CPU SumKernel.cpp with ONE intra-op
 * thread as torch 2.10 computes it -- the version the reference ran under when the golden fixtures were recorded
 * (the .npz files under tests/golden: meta["torch"], meta["intra_op_threads"]).  The host package warns once at import under another torch
 * (qsparse_amd.util.PINNED_TORCH / check_torch_pin); tests/test_aten_contract.py (CPU) and tests/test_aten_contract_gpu.py (the
 * same probes in the `-m gpu` set) detect a moved order. */

/* One stage of squeeze_tensor_to_shape (qsparse/util.py:92-99): mean over the middle dim of a contiguous
 * [pre, n, post] tensor -> [pre, 1, post], fp32 accumulation in ATen's CPU summation order (cascade /
 * 4-way interleave), one fp32 division by n, one rounding to odt.
 * flags: QS_MEAN_ABS takes |x| first (sparse.py:87);  QS_MEAN_L0 maps x -> (x != 0) when *l0_flag != 0
 * (sparse.py:85-86; l0_flag is a device int written by qs_l0_flag).
 * absmax_out (nullable, device float[C * absmax_stride]) is additionally max-ACCUMULATED with per-channel
 * max|x| at absmax_out[channel * absmax_stride], where channel = (column / chan_div) % C  -- the statistics of
 * a following tensor-wise QuantizeLayer fused into the same read of x.  The caller provides it zeroed
 * (qs_pq_select re-zeroes it after use).  The accumulation is one device atomic per wave and channel, and
 * MI355X serialises atomics that fall into the same 128-byte line (~10 ns each, exposed at the end of a short
 * kernel): absmax_stride = QS_AMAX_LINE_STRIDE gives every channel its own line (measured, batch-64 ResNet
 * activations: 33.9 -> 10.3 us on 256x14x14 maps, 87.9 -> 18.4 us on 128 x 64x32x32); 1 is a dense float[C]. */
#define QS_AMAX_LINE_STRIDE 32
#define QS_MEAN_ABS 1
#define QS_MEAN_L0 2
#define QS_MEAN_RELU 4 /* statistics of act(x): a folded preceding activation (abs-max included) -- nn.ReLU, or the one whose
                          qs_activation() handle rides in bits 8.. of the flags: QS_MEAN_ACT(handle) */
#define QS_MEAN_ACT(handle) (QS_MEAN_RELU | ((handle) << 8))
int qs_mean_dim(const void* x, void* out, int64_t pre, int64_t n, int64_t post,
                int xdt, int odt, int flags, const int32_t* l0_flag,
                float* absmax_out, int64_t absmax_stride, int64_t chan_div, int64_t C, qs_stream_t stream);

/* qs_mean_dim for a [pre, n, post] tensor that is the MEMORY of a permuted one (post >= 2): ATen decides between cascade and
 * row-sum order per output from the coordinates of the dim its TensorIterator puts innermost, which for such a tensor need not be
 * the flattened post (qsparse_amd/util.py `aten_reduce_plan`).  Where that set is a prefix of the columns -- the usual case, e.g. a
 * [B, T, C] activation seen as [B, C, T] and reduced over B: t < 4 * floor(T / 4) -- the caller names it: columns [0, mr_cols) of
 * every slice in cascade order, the others in row-sum order, through the same kernels as qs_mean_dim (16-byte loads, 8 columns per
 * lane).  No abs-max rider.  (ABI v23) */
int qs_mean_dim_split(const void* x, void* out, int64_t pre, int64_t n, int64_t post, int64_t mr_cols, int xdt, int odt, int flags,
                      const int32_t* l0_flag, qs_stream_t stream);

/* (ABI v25) The statistics of a token-major activation x[N][T][C] whose mask runs along the LAST dim (`prune(dimensions={2})` on a
 * transformer block's hidden activation; qsparse/sparse.py:231-239, util.py:92-99): the two stages of squeeze_tensor_to_shape --
 * stage[T][C] = mean over N (rounded to xdt), stage_mean[C] = mean over T of that (rounded) -- in ATen's order for the contiguous
 * tensor, i.e. exactly two qs_mean_dim calls, plus the per-channel abs-max of the mean's operand (|x|, or |act(x)| with
 * QS_MEAN_RELU / QS_MEAN_ACT) max-accumulated into chan_absmax[c * absmax_stride] (nullable; zero on entry as for qs_mean_dim).
 * Every column of a row is a channel of its own here, so the abs-max cannot ride per wave as in the NCHW stage: the first stage
 * stores one key per COLUMN into amax_part (nullable workspace, float[T * C], 16-byte aligned; no atomics) and a small third
 * launch folds it per channel.  Without amax_part -- or for widths / flags the per-column kernels do not serve -- the abs-max rides
 * in qs_mean_dim's atomic form (same values).  flags: QS_MEAN_ABS [| QS_MEAN_RELU | QS_MEAN_ACT(handle)]. */
int qs_token_stats(const void* x, void* stage, void* stage_mean, float* amax_part, float* chan_absmax, int64_t absmax_stride, int64_t N,
                   int64_t T, int64_t C, int xdt, int flags, qs_stream_t stream);

/* The first stage for a channels_last (NHWC in memory) activation x[n][hw][C]: mean over n ->
 * out[C][hw], NCHW-contiguous like the result of Tensor.mean(0, keepdim=True) on a channels_last tensor, in the
 * summation order ATen uses for that layout (per channel, positions hw < 4*floor(hw/4) in cascade order, the rest
 * 4-way interleaved; the later stages are the NCHW ones: qs_mean_last2 / qs_mean_dim).  flags as for qs_mean_dim
 * (QS_MEAN_L0 reads l0_flag, nullable otherwise).  C % 8 == 0 with flags 0, QS_MEAN_ABS or QS_MEAN_ABS|QS_MEAN_RELU and
 * a 16-byte aligned x take the vector kernels (16-byte loads, 8 channels per lane); every other channel count and flag
 * combination a scalar kernel in the same order.  amax_part (nullable, device float[C*hw]) receives the maximum over
 * n of |x| (or max(x, 0) with QS_MEAN_RELU) per output element; hand it to qs_mean_last2, which reduces it to the
 * per-channel abs-max without atomics. */
int qs_mean_dim_cl(const void* x, void* out, int64_t n, int64_t hw, int64_t C, int xdt, int odt, int flags,
                   const int32_t* l0_flag, float* amax_part, qs_stream_t stream);

/* The ONE stage of squeeze_tensor_to_shape (qsparse/util.py:92-99) for a channels_last activation x[N][H][W][C] whose first --
 * and then only -- reduced dim is W (a mask that keeps N, C and H; `prune(dimensions={0, 1, 2})`, qsparse/sparse.py:228-249):
 * mean over W -> out[N][C][H], NCHW-contiguous like Tensor.mean(3, keepdim=True) of such a tensor, in ATen's summation order
 * for it (scalar inner sum: four interleaved cascade sums over w, then ((p0 + p1) + p2) + p3; NOT the vectorised order of the
 * contiguous NCHW row).  flags / l0_flag as for qs_mean_dim.  (ABI v20) */
int qs_mean_cl_w(const void* x, void* out, int64_t N, int64_t H, int64_t W, int64_t C, int xdt, int odt, int flags,
                 const int32_t* l0_flag, qs_stream_t stream);

/* The FIRST stage of squeeze_tensor_to_shape (qsparse/util.py:92-99) for a dense tensor laid out in any dim order other than the
 * ones above (a transposed weight, a permuted activation, NDHWC with the batch kept ...): the mean over one dim of `n` elements
 * `stride` elements apart, read where the tensor lies (x needs element alignment only), written to the contiguous result
 * Tensor.mean(dim, keepdim=True) returns.  The caller describes the kept dims (nkept <= 6, after merging what merges) by size,
 * input stride and output stride, the one the lane index should run fastest over first, and names the summation order ATen's
 * TensorIterator + SumKernel.cpp arrive at for this layout (host logic, qsparse_amd/util.py `aten_reduce_plan`):
 *   order 0  vectorised inner sum (needs stride == 1, n >= 8): 8 interleaved row-sums, the n % 8 tail, then the 8 lanes in turn
 *   order 1  row-sum for every output (four interleaved cascade sums, ((p0 + p1) + p2) + p3)
 *   order 2  cascade sum for outputs whose coordinate in kept dim `split_dim` is < `split`, row-sum for the rest
 * flags / l0_flag as for qs_mean_dim (no abs-max rider).  (ABI v22) */
int qs_mean_strided(const void* x, void* out, int64_t n, int64_t stride, int nkept, const int64_t* kept_size,
                    const int64_t* kept_in_stride, const int64_t* kept_out_stride, int order, int split_dim, int64_t split,
                    int xdt, int odt, int flags, const int32_t* l0_flag, qs_stream_t stream);

/* The last two stages of squeeze_tensor_to_shape fused for a contiguous [pre, H, W] tensor whose trailing
 * two dims are both reduced: mean over H (rounded to xdt), then mean over W (rounded to odt) -> out[pre].
 * Same summation order and rounding points as two qs_mean_dim calls.  (H*W + W + 8)*4 bytes of LDS <= 63 KiB
 * (112 x 112 maps fit),
 * otherwise QS_ERR_ARG (use two qs_mean_dim calls).
 * amax_part (nullable, [pre, H, W] from qs_mean_dim_cl): absmax_out[p * absmax_stride] is max-accumulated with the
 * maximum of slice p.
 * record (nullable, device float[2 * pre]): this rank's exchange record in qs_stats_pack's layout, written on the way
 * out -- record[p] = float(out[p]), record[pre + p] = absmax_out[p * absmax_stride] (0 when absmax_out is NULL; with
 * amax_part NULL absmax_out is only read) -- which saves the qs_stats_pack launch of a data-parallel step. */
int qs_mean_last2(const void* x, void* out, int64_t pre, int64_t H, int64_t W, int xdt, int odt,
                  const float* amax_part, float* absmax_out, int64_t absmax_stride, float* record, qs_stream_t stream);

/* *flag = (min(x) == 0), qsparse/sparse.py:85 */
int qs_l0_flag(const void* x, int64_t numel, int xdt, int32_t* flag, float* scratch2, qs_stream_t stream);

/* state[i] <- (t*state[i] + f32(new[i])) / (t+1)   (MagnitudePruningCallback.update_magnitude,
 * qsparse/sparse.py:88-89) */
int qs_running_mean(float* state, const void* newv, int newdt, int64_t n, int64_t t, const int64_t* t_dev,
                    qs_stream_t stream);

/* ---- mask construction --------------------------------------------------------------------------- */

/* thr = sort_ascending(imp)[k]  (calculate_mask_given_importance, qsparse/util.py:113-116; the caller
 * computes k = max(int(s*n-1),0)+1 on the host exactly as the reference does).  NaNs order last.
 * ws: qs_workspace_bytes(QS_WS_KTH_VALUE, n) bytes. */
int qs_kth_value(const float* imp, int64_t n, int64_t k, float* thr, void* ws, size_t ws_bytes,
                 qs_stream_t stream);

/* mask[i] = imp[i] >= *thr   (util.py:117) */
int qs_mask_ge(const float* imp, const float* thr, uint8_t* mask, int64_t n, qs_stream_t stream);

/* ---- mask apply ------------------------------------------------------------------------------------ */

/* y = x * mask with a broadcast bool mask (qsparse/sparse.py:66,116,122,263) and, with g in place of x,
 * its backward g * mask.  The tensor is described by `ndim` collapsed extents `sizes`; `mask_strides[d]`
 * is the mask's element stride along d (0 where the mask has extent 1).
 * pre_relu != 0: y = max(x, 0) * mask, a preceding nn.ReLU folded into the prune site; only for masks that
 * vary along one run of dims (channel masks; QS_ERR_ARG otherwise).  Its backward is
 * qs_quant_ste_relu_bwd with step_host = 1 and lo_mul / hi_mul = -inf / +inf.
 * elide_masked != 0 (channel-type masks): pruned channels are not loaded and y = +0.0 there; the reference's x*0
 * carries the sign of x, so this is numerically equal, not bit-identical (opt-in).
 * gate_out (nullable; needs pre_relu != 0): the ReLU's gate bitmap for that backward, as in qs_quant_scaler_fwd. */
int qs_mask_apply(const void* x, const uint8_t* mask, void* y, int ndim, const int64_t* sizes,
                  const int64_t* mask_strides, int dt, int pre_relu, int elide_masked, uint8_t* gate_out,
                  qs_stream_t stream);

/* ---- fused channel-prune -> tensor-wise-quantize statistics (the headline pair) -------------------- */

/* One launch over C-sized state, replacing (per training step of the pair
 * Sequential(PruneLayer(dims={1}), QuantizeLayer(channelwise=-1)) built by convert(), convert.py:214-218):
 *   magnitude  <- (t_mag*magnitude + f32(stage_mean))/(t_mag+1)       if update_magnitude   (sparse.py:89)
 *   mask       <- magnitude >= sort(magnitude)[k]                      if refresh_mask       (util.py:113-117)
 *   absmax_all <- max over channels with mask != 0 of chan_absmax      (== max|x*mask|, quantize.py:329-340; a pruned
 *                 channel whose chan_absmax is Inf / NaN contributes a NaN: its x * 0 is NaN in the product the reference
 *                 takes x.abs().max() of, so the scale turns NaN there too)
 *   scale      <- t_q == 0 ? new : (t_q*scale + new)/(t_q+1), new = absmax_all/2^(bits-1)  if update_scale
 * stage_mean is the last squeeze stage's output ([C] in dtype sdt).  Single workgroup; C <= 65536.
 * chan_absmax[c * chan_absmax_stride] (stride as for qs_mean_dim's absmax_out, >= 1) is zeroed after use when
 * update_scale != 0.  bump_i32_a / bump_i32_b / bump_i64_a / bump_i64_b
 * (each nullable) are one-element device counters incremented by one at the end: the layers' `_n_updates`,
 * the pruning callback's `t` and the quantizer's device-side `t` (sparse.py:117,272; quantize.py:348,515), so
 * that a step needs no separate counter kernels.  t_mag_dev / t_q_dev: see "Step counters" above.  stat_dt: dtype
 * of the activation (the new scale's quotient is rounded to it, as in qs_scale_update).
 * gathered (nullable, device float[world][2*C]): the all-gathered qs_stats_pack records of a data-parallel run.  When
 * given, channel c's importance is (sum over ranks, in rank order, of gathered[r][c]) / world and its abs-max the
 * maximum over ranks of gathered[r][C + c] -- what qs_stats_combine computes -- read INSTEAD of stage_mean and
 * chan_absmax (which, when non-NULL, is still zeroed).
 * elide_mask_out (nullable, with update_scale): [C] bytes for a forward that skips the loads of pruned channels -- 1 for a kept
 * channel, 0 for a pruned channel whose abs-max this step is finite (x * 0 is a zero whatever x is: its loads may be skipped), 2
 * for a pruned channel that holds a NaN / Inf (x * 0 is NaN there and rounds to INT_MIN, quantize.py:109 on CPU: it must be
 * loaded).  Passed as the forward's chan_mask together with elide_masked, elision is bit-identical to the loading path for EVERY
 * input. */
int qs_pq_select(float* magnitude, const void* stage_mean, int sdt, int64_t C,
                 int update_magnitude, int64_t t_mag,
                 int refresh_mask, int64_t k, uint8_t* mask,
                 float* chan_absmax, int64_t chan_absmax_stride, int update_scale, int64_t t_q, int bits, float* scale,
                 int32_t* bump_i32_a, int32_t* bump_i32_b, int64_t* bump_i64_a, int64_t* bump_i64_b,
                 const int64_t* t_mag_dev, const int64_t* t_q_dev, int stat_dt, const float* gathered, int world,
                 uint8_t* elide_mask_out, qs_stream_t stream);

/* (ABI v25) qs_pq_select with its operands in a descriptor */
typedef struct qs_pq_select_args {
    uint32_t struct_size;        /* sizeof(qs_pq_select_args) as the caller compiled it */
    int32_t sdt, stat_dt, bits, world;
    int32_t update_magnitude, refresh_mask, update_scale;
    float* magnitude;
    const void* stage_mean;
    int64_t C, t_mag, k, t_q;
    uint8_t* mask;
    float* chan_absmax;
    int64_t chan_absmax_stride;
    float* scale;
    int32_t* bump_i32_a;
    int32_t* bump_i32_b;
    int64_t* bump_i64_a;
    int64_t* bump_i64_b;
    const int64_t* t_mag_dev;
    const int64_t* t_q_dev;
    const float* gathered;
    uint8_t* elide_mask_out;
    qs_stream_t stream;
} qs_pq_select_args;
int qs_pq_select_v(const qs_pq_select_args* args);

/* ---- data-parallel statistics exchange (no counterpart in the reference, whose masks and scales drift per rank) -- */

/* record[0..C) = f32(stage[i]) (0 when stage == NULL), record[C..2C) = absmax[i * absmax_stride] (0 when NULL):
 * the per-rank record of the fused pair's statistics, to be all-gathered by the caller (RCCL / any transport). */
int qs_stats_pack(const void* stage, int sdt, const float* absmax, int64_t absmax_stride, int64_t C, float* record,
                  qs_stream_t stream);

/* gathered = `world` records of 2C floats in rank order.  stage_out[i] = (sum over ranks, in rank order) / world;
 * absmax_out[i * absmax_stride] = max over ranks.  Either output may be NULL. */
int qs_stats_combine(const float* gathered, int world, int64_t C, float* stage_out, float* absmax_out,
                     int64_t absmax_stride, qs_stream_t stream);

/* ---- one activation site per call ------------------------------------------------------------------- */

/* A convert-built activation site -- Sequential(Sequential(act, PruneLayer{dims={1}}), QuantizeLayer{tensor-wise
 * ScalerQuantizer}), reference convert.py:214-218 -- runs, per training step, the launches
 *     statistics (qs_mean_dim | qs_mean_dim_cl)  ->  qs_mean_last2  ->  qs_pq_select  ->  qs_quant_scaler_fwd
 * and one launch backward.  Issued one by one from the host language each of them costs an FFI transition plus argument
 * marshalling (20-30 us of Python each); these two entry points enqueue the whole sequence from ONE call.  They add no
 * arithmetic: they call the entry points above with the arguments the plan describes, in that order, on `stream`.
 *
 * The plan names what does not change from step to step: the geometry of x as the kernels address it (NCHW-contiguous:
 * layout 0, x = [N][C][H*W]; dense channels_last: layout 1, x = [N][H*W][C]; a 2-d activation [N][C] -- nn.Linear's output:
 * layout 2 with H = W = 1, whose statistics are ONE qs_mean_dim launch, no qs_mean_last2), dtypes, the layer state (running
 * magnitude, mask, running scale, step counters -- all device pointers, the objects PruneLayer / QuantizeLayer /
 * MagnitudePruningCallback hold, reference sparse.py:58-122, quantize.py:327-349, 473-518) and the caller's workspaces
 * (stage: C*H*W elements of xdt; amax_part: C*H*W floats, channels_last only; chan_absmax: C * absmax_stride floats,
 * zero on entry and re-zeroed by the select).  The caller builds it once per site and input signature and rebuilds it
 * when a pointer or a shape changes. */
typedef struct qs_site_plan {
    int64_t N, C, H, W;
    int32_t layout;              /* 0: NCHW-contiguous, 1: channels_last (NHWC in memory), 2: [N][C] (H = W = 1) */
    int32_t xdt, ydt;            /* input dtype; output dtype (QS_F32: the reference's promotion, or xdt) */
    int32_t bits;                /* of the quantizer */
    float* magnitude;            /* [C]  MagnitudePruningCallback.magnitude */
    uint8_t* mask;               /* [C]  PruneLayer.mask */
    float* scale;                /* [1]  QuantizeLayer.weight */
    float* chan_absmax;          /* [C * absmax_stride] scratch accumulator */
    int64_t absmax_stride;
    void* stage;                 /* [C*H*W] xdt scratch: first-stage means (unused, nullable, for layout 2) */
    float* amax_part;            /* [C*H*W] scratch (layout 1; layout 3, nullable there: qs_token_stats' per-column keys), NULL for layout 0 */
    void* stage_mean;            /* [C] xdt scratch: the importance of this step */
    int32_t* prune_n_updates;    /* nullable: PruneLayer._n_updates, incremented by the select */
    int32_t* quant_n_updates;    /* nullable: QuantizeLayer._n_updates, incremented by the select */
    int64_t* callback_t;         /* nullable: MagnitudePruningCallback.t, incremented by the select */
    int64_t* quantizer_t_dev;    /* nullable: device copy of the quantizer callback's t (graph-safe mode): read instead of
                                    t_q and incremented */
    int32_t callback_t_from_device; /* != 0: the running-magnitude counter is read from *callback_t instead of t_mag */
    int32_t saturate;            /* != 0: codes are clamped to [code_lo, code_hi] (qs_quant_scaler_fwd's opt-in saturation) */
    int32_t code_lo, code_hi;
    int32_t act;                 /* the activation QS_SITE_PRE_RELU folds: 0 / 1 nn.ReLU, else a qs_activation() handle */
    uint8_t* elide_mask;         /* nullable: [C] scratch, qs_pq_select's elide_mask_out: the chan_mask of an eliding forward on the
                                    steps that have statistics (QS_SITE_LIVE), exact for every x; without it, and on steps without
                                    statistics, QS_SITE_ELIDE elides through the mask itself (exact for finite x only) */
    float* absmax_dense;         /* nullable: [C] scratch accumulator of QS_SITE_SCALE_ONLY steps (zero on entry, re-zeroed by the select) */
    void* reduce_ws;             /* nullable: qs_absmax's `ws` for this geometry (QS_SITE_SCALE_ONLY steps) */
    int64_t reduce_ws_bytes;
} qs_site_plan;

/* flags of qs_site_fwd */
#define QS_SITE_LIVE 1        /* training step with live statistics: magnitude and scale are updated (else: apply only) */
#define QS_SITE_REFRESH 2     /* rebuild the mask from the running magnitude, threshold rank k */
#define QS_SITE_PRE_RELU 4    /* x is the input of a folded activation (plan->act; nn.ReLU by default) */
#define QS_SITE_ELIDE 8       /* elide_masked of qs_quant_scaler_fwd (through plan->elide_mask on QS_SITE_LIVE steps) */
#define QS_SITE_NO_MASK 16    /* apply without the channel mask (pruning not started) -- only without QS_SITE_LIVE */
#define QS_SITE_STATS_DONE 32 /* with QS_SITE_LIVE: the statistics launches were already enqueued by qs_site_stats (a data-parallel
                                 step: the caller exchanged the record in between); qs_site_fwd starts at the select */
#define QS_SITE_NO_QUANT 128  /* the site is a PruneLayer ALONE (reference sparse.py:215-273 behind an activation, convert.py:214-218;
                                 plan->scale and the quantizer fields are unused): with QS_SITE_LIVE the statistics are the staged
                                 mean alone (no abs-max), with QS_SITE_LIVE or QS_SITE_REFRESH the select updates magnitude / mask /
                                 the prune counters, and the apply is y = act?(x) * mask in x's dtype (qs_mask_apply); qs_site_bwd
                                 with the flag: gx = gate * g * mask (qs_quant_ste_relu_bwd without a clamp) or g * mask */
#define QS_SITE_SCALE_ONLY 64 /* with QS_SITE_LIVE: the mask is frozen (MagnitudePruningCallback.t passed stop_mask_refresh,
                                 sparse.py:107-116 -- the steady state of devise_layerwise_pruning_schedule recipes, :343-359):
                                 magnitude and mask stay, the statistics are qs_absmax per channel into plan->absmax_dense
                                 and the select updates the scale (max over the kept channels) and the counters */

/* y = Q(relu?(x) * mask); with QS_SITE_LIVE preceded by statistics + select exactly as the four calls above.
 * gate_out, image_out / imgdt, xback_out: nullable, see qs_quant_scaler_fwd.  t_mag / t_q: the running-mean counters of this step (reference
 * sparse.py:88, quantize.py:344), k: threshold rank (util.py:115-116).  gathered / world: nullable / 1; the all-gathered
 * records of qs_site_stats (see there and qs_pq_select).
 * decimal: NULL for a ScalerQuantizer; else the site's quantizer is a DecimalQuantizer (reference quantize.py:275-367, same
 * running scale, power-of-two step): device float[1] of THIS call, which receives qs_decimal_from_scale(plan->scale) after the
 * select (one more launch) and is what qs_quant_decimal_fwd applies -- and what qs_site_bwd of this forward must be given. */
int qs_site_fwd(const qs_site_plan* plan, const void* x, void* y, uint8_t* gate_out, int flags, int64_t t_mag, int64_t k,
                int64_t t_q, void* image_out, int imgdt, const float* gathered, int world, void* xback_out, float* decimal,
                qs_stream_t stream);

/* (ABI v25) qs_site_fwd with its per-call operands in a descriptor */
typedef struct qs_site_fwd_args {
    uint32_t struct_size;        /* sizeof(qs_site_fwd_args) as the caller compiled it */
    int32_t flags, imgdt, world;
    const void* x;
    void* y;
    uint8_t* gate_out;
    int64_t t_mag, k, t_q;
    void* image_out;
    const float* gathered;
    void* xback_out;
    float* decimal;
    qs_stream_t stream;
} qs_site_fwd_args;
int qs_site_fwd_v(const qs_site_plan* plan, const qs_site_fwd_args* args);

/* The statistics half of a live qs_site_fwd on its own -- qs_mean_dim | qs_mean_dim_cl, then qs_mean_last2, which also writes
 * this rank's exchange record (record: device float[2*C] = importance | per-channel abs-max, qs_stats_pack's layout) -- for a
 * data-parallel step: the caller all-gathers the records of all ranks (RCCL / any transport) and passes them to
 * qs_site_fwd(flags | QS_SITE_STATS_DONE, gathered, world), whose select combines them in rank order (qs_pq_select's
 * `gathered`).  Two calls and one collective per site instead of five calls; the reference has no counterpart (its masks and
 * scales drift per rank, sparse.py:58-122 / quantize.py:327-349 run on the rank's shard only).  flags: QS_SITE_PRE_RELU. */
int qs_site_stats(const qs_site_plan* plan, const void* x, int flags, float* record, qs_stream_t stream);

/* gx = gate * clamp(g) * mask in xdt (qs_quant_ste_relu_bwd with the bitmap when `gate` is given, qs_quant_ste_bwd
 * otherwise); g has dtype gdt, the geometry of the plan.  lo_mul / hi_mul as there.  g2 / g2dt: the second gradient of
 * qs_quant_ste_relu_bwd (needs `gate`; g may then be NULL).  decimal: NULL (the clamp follows plan->scale) or the float[1] its
 * forward call filled (DecimalQuantizer: the clamp follows 2^-decimal). */
int qs_site_bwd(const qs_site_plan* plan, const void* g, const uint8_t* gate, void* gx, int gdt, int flags, float lo_mul,
                float hi_mul, const void* g2, int g2dt, const float* decimal, qs_stream_t stream);

/* (ABI v25) qs_site_bwd with its operands in a size-prefixed descriptor; g3 / gx_image / gx_image_dt as in
 * qs_ste_relu_bwd_args (they need `gate` and a float32 site: plan->xdt == gdt == QS_F32). */
typedef struct qs_site_bwd_args {
    uint32_t struct_size;        /* sizeof(qs_site_bwd_args) as the caller compiled it */
    int32_t flags;
    int32_t gdt, g2dt;
    const void* g;
    const uint8_t* gate;
    void* gx;
    float lo_mul, hi_mul;
    const void* g2;
    const float* decimal;
    qs_stream_t stream;
    /* ---- v25 ---- */
    const void* g3;
    void* gx_image;
    int32_t gx_image_dt;
    int32_t reserved0;
    /* ---- v26 ---- */
    const void* act_x;           /* the caller's activation, as in qs_ste_relu_bwd_args (needs a quantizing site; `gate` is not read) */
    int32_t act_x_kind;
    int32_t reserved1;
} qs_site_bwd_args;
int qs_site_bwd_v(const qs_site_plan* plan, const qs_site_bwd_args* args);

/* A lone tensor-wise ScalerQuantizer step -- QuantizeLayer.forward in training (reference quantize.py:473-518 with
 * optimize :327-349 and ScalerQuantization.forward :100-117) -- from ONE call: with `update` != 0
 *     qs_absmax(x, amax_lines, tensor-wise, accumulate, pre_relu, lines)  ->  qs_scale_update(amax_lines, lines, scale, 1, t,
 *     t_dev, advance, bits, clear, n_updates, xdt)  ->  qs_quant_scaler_fwd(x, y, scale, pre_relu, gate_out)
 * and with `update` == 0 the last of them alone (evaluation).  amax_lines: [lines][QS_AMAX_LINE_STRIDE] floats, zero on entry,
 * re-zeroed by the update; n_updates (nullable) is incremented; t_dev (nullable) is read instead of t and incremented.
 * image_out / imgdt (ABI v21): the autocast image of qs_quant_scaler_fwd -- RNE(y) in bf16 / fp16 from the same pass, for the
 * convolution behind a quantize-only activation site (needs pre_relu, gate_out and qs_quant_image_ok(1, 1, numel, ...)). */
#define QS_QSTEP_APPLY 0      /* `update` of qs_quantize_step: quantize only (evaluation) */
#define QS_QSTEP_ALL 1        /* abs-max, running scale, quantize */
#define QS_QSTEP_ABSMAX 2     /* the abs-max launch alone (y may be NULL): a data-parallel step all-reduces (MAX) the accumulator
                                 lines between this call and the next */
#define QS_QSTEP_FINISH 3     /* running scale from the (reduced) accumulator lines, then quantize */
int qs_quantize_step(const void* x, void* y, uint8_t* gate_out, float* amax_lines, int lines, float* scale, int64_t numel,
                     int xdt, int ydt, int bits, int64_t t, int64_t* t_dev, int32_t* n_updates, int pre_relu, int update,
                     int saturate, int32_t code_lo, int32_t code_hi, void* xback_out, void* image_out, int imgdt, qs_stream_t stream);

/* (ABI v25) qs_quantize_step with its operands in a descriptor */
typedef struct qs_quantize_step_args {
    uint32_t struct_size;        /* sizeof(qs_quantize_step_args) as the caller compiled it */
    int32_t lines, xdt, ydt, bits, pre_relu, update, saturate, code_lo, code_hi, imgdt;
    const void* x;
    void* y;
    uint8_t* gate_out;
    float* amax_lines;
    float* scale;
    int64_t numel, t;
    int64_t* t_dev;
    int32_t* n_updates;
    void* xback_out;
    void* image_out;
    qs_stream_t stream;
} qs_quantize_step_args;
int qs_quantize_step_v(const qs_quantize_step_args* args);

/* ---- multi-tensor weight path ---------------------------------------------------------------------- */

/* The weight-side operators of a converted network (quantize(conv) / quantize(linear), reference quantize.py:559-571
 * through imitation.py:61-68) are the same three small kernels per layer and step: abs-max, running scale, quantization.
 * These entry points run them for a LIST of float32 tensors in one launch each.  The list is a DEVICE-resident table of
 * qs_multi_row that the caller builds once per set of layers (host copy -> qs_multi_plan fills the derived fields -> upload)
 * and reuses every step; what changes from step to step -- the output buffer, which rows train -- is either an argument
 * (`ybase`) or a field the caller rewrites.  Tensor-wise AND per-channel Scaler / Decimal quantizers: a tensor is the
 * CONTIGUOUS [outer, C, inner] view around the reference's channel dim (C == 1, outer == 1, inner == numel for tensor-wise;
 * the reference's default for weights is channelwise=1, quantize.py:524); parameters have C entries.
 *   qs_multi_absmax        rows with train != 0: amax[c] = max(amax[c], max |x| over channel c)   (quantize.py:329-340; keep
 *                          the accumulators zero between steps -- the update below re-zeroes them)
 *   qs_multi_scale_update  rows with train != 0, every channel c: t = *t_dev + t_offset;
 *                          scale[c] <- t == 0 ? new : (t*scale[c] + new)/(t+1), new = amax[c] / 2^(bits-1) (denom)   (:344-347);
 *                          amax[c] <- 0; decimal[c] <- rint(log2(nan_to_num(1/scale[c]))) where decimal != NULL (:316);
 *                          backup[c] (nullable) <- the scale this update replaces, so that a caller that evaluated a layer
 *                          ahead of time can restore one the forward pass then never reached (imitation.py:61-68 evaluates
 *                          the operator only when the layer's weight is read)
 *   qs_multi_quant_fwd     every row: y[e] = Q(x[e]) with the scale (is_decimal == 0: qs_quant_scaler_fwd's arithmetic) or the
 *                          decimal (qs_quant_decimal_fwd's) of e's channel, y = ybase + y_off; codes clamped to
 *                          [code_lo, code_hi] where code_lo <= code_hi (the opt-in saturation).  advance != 0: the counters of
 *                          the rows that trained -- *t_dev (quantizer callback's t, quantize.py:348) and *bump (the layer's
 *                          `_n_updates`, :515) -- are incremented here, i.e. after every thread of the preceding
 *                          qs_multi_scale_update has read them (stream order).  A callback shared by a layer's weight and
 *                          bias quantizers (:548,559-571) is two rows with the same t_dev and t_offset 0 / 1.
 * A PRUNED weight -- quantize(prune(conv)): the quantizer's input is weight * mask (sparse.py:263 through imitation.py:61-68) --
 * is a row with `mask` set: every x[e] above reads x[e] * mask[.] (the product itself, so that a non-finite weight under a
 * pruned position behaves as in the reference).  mask_C == 0: one mask byte per element; else the mask varies along one run of
 * dims: byte (e / mask_inner) % mask_C.  Steps on which the prune operator only applies its mask take part (before `start`
 * without a mask; after the schedule with a frozen or a still-averaging full-shape magnitude); its counters -- prune_n_updates
 * (PruneLayer._n_updates, sparse.py:272) and prune_t (MagnitudePruningCallback.t, :117) -- are incremented next to the
 * quantizer's where non-NULL.
 *   qs_multi_magnitude     rows with magnitude != NULL (full-shape masks): t = *prune_t;
 *                          mag_backup[e] <- magnitude[e]; magnitude[e] <- (t*magnitude[e] + |x[e]|)/(t+1)   (sparse.py:82-89);
 *                          launched BEFORE qs_multi_quant_fwd (which advances prune_t)
 *   qs_multi_mask_refresh  rows with refresh != 0 (full-shape masks): the mask rebuild of MagnitudePruningCallback
 *                          (sparse.py:58-66 -> util.py:103-117): thr = the select_k-th smallest importance (importance =
 *                          `importance` when non-NULL -- the running magnitude, after qs_multi_magnitude -- else |x|), by the
 *                          4-pass radix select of qs_kth_value run for all rows at once (select_state: 258 uint32 of scratch
 *                          per row, zero before the first use); mask_backup[e] <- mask[e]; mask[e] <- importance[e] >= thr.
 *                          Nine launches for the whole table; BEFORE qs_multi_absmax (the quantizer sees the new mask). */
typedef struct qs_multi_row {
    const float* x;              /* the tensor, 4-byte aligned (16-byte aligned tensors take the vector paths) */
    float* scale;                /* [C] running scale (QuantizeLayer.weight) */
    float* amax;                 /* [C] abs-max accumulator, zero between steps */
    float* decimal;              /* [C] or NULL: decimals of a DecimalQuantizer (written by the update, read by the quantizer) */
    float* backup;               /* [C] or NULL */
    int64_t* t_dev;              /* device copy of the quantizer callback's running-mean count (required when train != 0) */
    int32_t* bump;               /* nullable: the layer's `_n_updates` */
    int64_t numel, y_off;        /* elements; offset of the output in `ybase`, elements */
    int64_t outer, inner;        /* [outer, C, inner] */
    int32_t C;
    int32_t train;               /* != 0: this row updates its statistics this step */
    int32_t is_decimal;
    int32_t t_offset;            /* added to *t_dev (1 for the bias quantizer that shares its weight quantizer's callback) */
    int32_t code_lo, code_hi;    /* code_lo > code_hi: no saturation */
    float denom;                 /* 2^(bits-1) */
    const uint8_t* mask;         /* nullable: the prune operator's mask (bool bytes) */
    int64_t mask_inner;          /* see above (ignored with mask_C == 0) */
    int32_t mask_C;              /* 0: one mask byte per element */
    int32_t kind;                /* 0: a weight / bias row.  1: a MASK-LEVEL row -- the importance of a pruned weight whose mask
                                    varies along a subset of dims (x: [numel] float, the staged mean of |weight| that
                                    qs_multi_stage_mean produced this step; numel = the mask's): it takes part in
                                    qs_multi_magnitude and qs_multi_mask_refresh only (scale may be NULL, nothing is quantized).
                                    2: a weight that is NOT quantized on this read -- prune(conv) without a quantizer, or a
                                    quantizer still in its identity phase (quantize.py:496-517: it only counts): y = x * mask
                                    (x where mask == NULL), `bump` (nullable: the idle quantizer's `_n_updates`) incremented,
                                    scale may be NULL */
    int32_t* prune_n_updates;    /* nullable */
    int64_t* prune_t;            /* nullable (required with magnitude) */
    float* magnitude;            /* nullable: [numel] running magnitude, updated by qs_multi_magnitude */
    float* mag_backup;           /* [numel] with magnitude: what the update replaced */
    int32_t refresh;             /* != 0: qs_multi_mask_refresh rebuilds this row's (full-shape, writable) mask */
    uint32_t select_k;           /* rank of the threshold element, ascending (util.py:115-116) */
    const float* importance;     /* nullable: [numel] importance to rank by (NULL: |x|) */
    uint32_t* select_state;      /* [258] scratch of the radix select */
    uint8_t* mask_backup;        /* [numel]: what the rebuild replaced */
    /* derived by qs_multi_plan: */
    int32_t row_splits, absmax_block0, absmax_blocks, quant_block0, chan0, hist_block0, hist_blocks, reserved1;
} qs_multi_row;

/* fills the derived fields of a HOST table in place and returns the launch totals; QS_ERR_ARG for an inconsistent row */
int qs_multi_plan(qs_multi_row* rows_host, int n, int* absmax_blocks, int* quant_blocks, int* channels, int* hist_blocks_out);
int qs_multi_absmax(const qs_multi_row* rows_dev, int n, int absmax_blocks, qs_stream_t stream);
int qs_multi_scale_update(const qs_multi_row* rows_dev, int n, int channels, qs_stream_t stream);
int qs_multi_quant_fwd(const qs_multi_row* rows_dev, int n, int quant_blocks, float* ybase, int advance, qs_stream_t stream);
int qs_multi_magnitude(const qs_multi_row* rows_dev, int n, int quant_blocks, qs_stream_t stream);
/* The importance of pruned weights whose mask varies along a SUBSET of dims (prune()'s default dimensions={1}: one mask entry
 * per input channel) is squeeze_tensor_to_shape(|weight|, mask.shape) (reference util.py:79-99 via sparse.py:82-89): one mean per
 * reduced dim, ascending, each rounded to float32.  qs_multi_stage_mean runs ONE such stage for a list of tensors: entry i
 * averages the middle dim of the contiguous [pre, n, post] tensor x into out [pre, 1, post] (layout 0; |x| first with take_abs),
 * or -- layout 1, the first stage of a channels_last weight -- the leading dim of x = [n][hw = pre][C = post] in memory into the
 * NCHW-contiguous out [C][hw], in ATen's CPU summation order for that case (the orders of qs_mean_dim / qs_mean_dim_cl, one
 * lane per output element).  The caller launches it once per stage level; qs_multi_stage_plan fills block0 of a HOST table. */
typedef struct qs_multi_stage {
    const float* x;
    float* out;
    int64_t pre, n, post;
    int32_t take_abs, layout;
    int32_t block0, reserved;     /* derived */
} qs_multi_stage;
int qs_multi_stage_plan(qs_multi_stage* stages_host, int n, int* blocks);
int qs_multi_stage_mean(const qs_multi_stage* stages_dev, int n, int blocks, qs_stream_t stream);

/* hist_blocks: the total qs_multi_plan returned through *hist_blocks_out (0: no row refreshes) */
int qs_multi_mask_refresh(const qs_multi_row* rows_dev, int n, int hist_blocks, int quant_blocks, qs_stream_t stream);
/*   qs_multi_ste_bwd:       gx[i][e] = clamp(g[i][e], lo_mul[i] * s, hi_mul[i] * s) with s = step[i][c] (or 2^-step[i][c] with
 *                           step_is_decimal), c the channel of e in the contiguous [*, C[i], inner[i]] view the gradient has
 *                           (C == NULL: tensor-wise, one step per tensor): qs_quant_ste_bwd's arithmetic (quantize.py:66-77,
 *                           120-131) for the gradients of a GROUP of weight quantizers that are handed over together.  The
 *                           gradients are fresh tensors every step: HOST arrays of length n of device pointers / values.
 *                           mask / mask_C / mask_inner (each array nullable, entries nullable / 0): the pruned weight's
 *                           backward, gx = clamp(g) * mask (the product: a signed zero under a pruned position, sparse.py:263). */
int qs_multi_ste_bwd(int n, const float* const* g, float* const* gx, float* const* step, const int64_t* numel,
                     const int32_t* C, const int64_t* inner, const float* lo_mul, const float* hi_mul, int step_is_decimal,
                     const uint8_t* const* mask, const int32_t* mask_C, const int64_t* mask_inner, qs_stream_t stream);

/* ---- data-parallel statistics exchange through peer-mapped mailboxes (prototype, ABI v24) -----------------------------------
 * The collective form of the exchange (qs_stats_pack / the record written by the last statistics launch -> all-gather ->
 * qs_pq_select / qs_site_fwd with `gathered`) costs a host-bound network ~40 us of host time per site and step.  The mailbox form
 * replaces the collective by two launches.  Every rank allocates one mailbox per site (qs_mailbox_bytes / qs_mailbox_alloc:
 * fine-grained device memory, zeroed), exports it (qs_mailbox_export: a 64-byte hipIpcMemHandle_t the caller ships to the peers by
 * whatever means it has) and maps its peers' (qs_mailbox_open).  Per step, after the statistics launches have written this rank's
 * n-float record: qs_mailbox_publish stores the record into slot `rank` of EVERY mailbox in `boxes` (host array of `world` device
 * pointers, this rank's own included; one-sided stores over xGMI), fences system-wide and raises flag `rank` there to `step`;
 * qs_mailbox_wait spins -- at most max_spins polls per rank, never a hang -- until every flag of the local mailbox shows `step`,
 * and returns in *records the [world][n] float records of this step (a device pointer into the mailbox: the `gathered` argument
 * of qs_pq_select / qs_site_fwd, combined in rank order there as after the all-gather).  A rank whose flag did not arrive in time
 * is FATAL FOR THE STEP (ABI v25): *status = 1 + that rank and its record of this step is overwritten with NaNs before anything
 * reads it, so the statistics -- magnitude, scale, output -- of a rank that missed a peer turn NaN at once instead of drifting on a
 * stale record; the caller polls *status (one word; asynchronously is enough) and stops.
 * qs_mailbox_alloc returns fine-grained memory or an error (ABI v25: no coarse-grained fall-back -- a peer's stores would not be
 * guaranteed visible to the local acquire loads); the caller then keeps the collective exchange.
 * qs_records_max (ABI v25): out[i] = max over ranks of records[r][i] as uint32 keys -- the all-reduce (MAX) of a quantize-only
 * site's abs-max accumulator lines (qs_quantize_step, QS_QSTEP_ABSMAX / _FINISH) taken from the mailbox.
 * `step` counts from 1 and alternates between the mailbox's two halves; no rank can be more than one step ahead of the slowest
 * (its next publish is ordered behind its own select), so a half is never written while it is read.  qs_mailbox_alloc / _free /
 * _open / _close are set-up calls: they allocate, map and synchronise like the hip calls they wrap. */
size_t qs_mailbox_bytes(int world, int64_t n);
int qs_mailbox_alloc(size_t bytes, void** ptr);
int qs_mailbox_free(void* ptr);
int qs_mailbox_export(void* ptr, void* handle64);
int qs_mailbox_open(const void* handle64, void** ptr);
int qs_mailbox_close(void* ptr);
int qs_mailbox_publish(const float* rec, int64_t n, void* const* boxes, int world, int rank, uint32_t step, qs_stream_t stream);
int qs_mailbox_wait(void* box, int world, int64_t n, uint32_t step, int32_t* status, uint32_t max_spins, const float** records,
                    qs_stream_t stream);
int qs_records_max(const float* records, int world, int64_t n, float* out, qs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QSPARSE_HIP_H */
