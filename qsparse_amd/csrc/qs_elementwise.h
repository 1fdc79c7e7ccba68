// Element-wise kernels: quantizer forward (scaler / decimal / line), STE backward, mask apply.
// All of them are HBM-bound streams: 16-byte non-temporal loads and stores per lane, one 256-thread workgroup
// per 2048 elements (exact grids measured faster than capped grid-stride loops on MI355X).
#pragma once
#include "qs_common.h"

namespace qs {

// CM_LAST: the channel dim is the innermost one (channels_last activations viewed as [N*H*W, C]), C % 8 == 0, the
// parameter is tensor-wise and only the channel mask varies: the 8 (4) elements of a lane are 8 (4) consecutive
// channels, whose mask bytes come with one aligned load.
enum ChanMode { CM_SCALAR = 0, CM_ROW = 1, CM_ELEM = 2, CM_LAST = 3 };

struct EwGeom {
    int64_t numel;
    int64_t ngroups;          // numel / 8
    uint32_t C;               // channel extent
    uint32_t inner;           // elements after the channel dim
    uint32_t groups_per_row;  // inner / 8 (CM_ROW only)
    uint32_t reverse;         // walk the tensor from its end (cache-reuse hint, see launch_ew)
};

// ------------------------------------------------------------------------------------------------
// Ops.  Each op exposes
//   struct P                      per-channel parameters held in registers
//   P channel(uint32_t c) const   fetch them (c == 0 in scalar mode)
//   float apply(float v, const P&, int32_t& code) const
// ------------------------------------------------------------------------------------------------

// float -> int32 as ATen's CPU kernels do it (cvttps2dq): NaN and everything outside [-2^31, 2^31) give INT_MIN, the
// x86 "integer indefinite".  Only reachable with a zero / denormal scale (an all-zero tensor), where the reference's
// output is -0.0 everywhere; v_cvt_i32_f32 alone would saturate and give +0.0 for NaN and +inf.
#ifndef QS_X86_CVT
#define QS_X86_CVT 1
#endif
__device__ __forceinline__ int32_t f32_to_i32_x86(float q) {
#if QS_X86_CVT
    return (fabsf(q) < 2147483648.0f) ? (int32_t)q : (int32_t)0x80000000;
#else
    return (int32_t)q;
#endif
}

// rint(RN(v / s)) given r = RN(1/s), without dividing in the common case (see ScalerFwdOp::quotient_rint for the
// error argument); r = NaN forces the division.
__device__ __forceinline__ float rint_of_quotient(float v, float s, float r) {
    const float t = v * r;
    const float n = rintf(t);
    const float off = fabsf(fabsf(t - n) - 0.5f);      // distance of t from the nearest k + 0.5
    if (__builtin_expect(off > fabsf(t) * 4.76837158203125e-07f, 1)) return n;   // 2^-21
    return rintf(v / s);
}
// the same as an int32 code.  The fast path only sees |t| < 2^22 (beyond that `off` is 0.5 or 0 and the test fails),
// so the x86 conversion rule for NaN / out-of-range values costs nothing there.
__device__ __forceinline__ int32_t rint_of_quotient_i32(float v, float s, float r) {
    const float t = v * r;
    const float n = rintf(t);
    const float off = fabsf(fabsf(t - n) - 0.5f);
    if (__builtin_expect(off > fabsf(t) * 4.76837158203125e-07f, 1)) return (int32_t)n;
    return f32_to_i32_x86(rintf(v / s));
}
__device__ __forceinline__ float guarded_reciprocal(float s) {
    float r = 1.0f / s;
    if (fabsf(r) < 1.17549435e-38f) r = __builtin_nanf("");   // subnormal reciprocal: always divide
    return r;
}

// The channel mask byte a quantizer forward sees is the PruneLayer's mask (0 / 1) -- or, where the forward may skip the loads of
// pruned channels, the ELISION MASK the select wrote next to it: 1 kept, 0 pruned and finite (its loads may be skipped: the
// product x * 0 is a zero whatever x is), 2 pruned but this step's statistics saw a NaN / Inf in the channel (x * 0 is NaN there,
// reference quirk B15: it must be loaded).  `keep` carries all three: 1.0, +0.0 (skippable) and -0.0 (a zero factor like any other
// -- the product's zero may take the other sign, the code it rounds to is 0 either way -- but not skippable).
// Decoded arithmetically -- (-f) * (f - 2) for f = float(m) is +0.0, 1.0, -0.0 at m = 0, 1, 2 -- so that the per-element decode
// of the channels-last kernels is a byte-to-float conversion (v_cvt_f32_ubyteN, straight from the packed word), an add and a
// multiply instead of two compare / select pairs (ResNet-50, batch 256: the apply forward family 2.88 -> 2.80 ms per step).  Mask
// bytes are 0 / 1 (torch.bool) or 0 / 1 / 2 (the elision mask); nothing else is defined.
__device__ __forceinline__ float keep_from_byte(uint32_t m) {
    const float f = (float)m;
    return (-f) * (f - 2.0f);
}
template <typename P>
__device__ __forceinline__ bool needs_load(const P& p) { return __float_as_uint(p.keep) != 0u; }

// ScalerQuantization.forward  (reference qsparse/quantize.py:100-117)
template <int QDT>
struct ScalerFwdOp {
    static constexpr bool kHasMask = true;
    const float* scale;     // device, nullable
    float scale_host;
    const uint8_t* cmask;   // device, nullable: fused channel prune
    int saturate;
    int32_t lo, hi;
    ActSpec act;            // folded preceding activation (nn.ReLU: quantise max(x, 0)); act_dt: the dtype of its input
    int act_dt;
    struct P {
        float s;
        float r;      // RN(1/s)
        float keep;
    };
    __device__ __forceinline__ P channel(uint32_t c) const {
        P p;
        p.s = scale ? scale[c] : scale_host;
        p.r = guarded_reciprocal(p.s);
        p.keep = 1.0f;
        return p;
    }
    __device__ __forceinline__ P channel_masked(uint32_t c_scale, uint32_t c_mask) const {
        P p = channel(c_scale);
        if (cmask) p.keep = keep_from_byte(cmask[c_mask]);
        return p;
    }
    __host__ __device__ __forceinline__ const uint8_t* mask_ptr() const { return cmask; }
    __device__ __forceinline__ static P keep_of(P p, uint32_t m) {
        p.keep = keep_from_byte(m);
        return p;
    }
    // rint(RN(v / s)) without dividing in the common case.  t = RN(v * RN(1/s)) differs from RN(v/s) by at
    // most 1.5 * 2^-23 relative, so both round to the same integer unless t lies within |t| * 2^-21 of a
    // half-way point k + 0.5; only then (probability ~2^-18..2^-14 per element for 4..8-bit codes) is the
    // correctly rounded division evaluated.  NaN / inf / huge t fall through to the division as well.
    __device__ __forceinline__ float quotient_rint(float v, const P& p) const {
        if constexpr (QDT != QS_F32) {
            return rintf(round_through<QDT>(v / p.s));   // quotient rounded to the input dtype first
        } else {
            return rint_of_quotient(v, p.s, p.r);
        }
    }
    __device__ __forceinline__ float apply(float v, const P& p, int32_t& code) const {
        if (act.kind) v = act_apply(v, act, act_dt);
        v = v * p.keep;                          // x * mask (exact; keeps the sign of zero)
        int32_t qi;                                   // round(x / s).int(): half-to-even (:109)
        if constexpr (QDT != QS_F32) qi = f32_to_i32_x86(quotient_rint(v, p));
        else qi = rint_of_quotient_i32(v, p.s, p.r);
        if (saturate) qi = qi < lo ? lo : (qi > hi ? hi : qi);
        code = qi;
        return (float)qi * p.s;                  // q.float() * scaler (:117)
    }
};

// DecimalQuantization.forward  (reference qsparse/quantize.py:44-63)
template <int QDT>
struct DecimalFwdOp {
    static constexpr bool kHasMask = true;
    const float* decimal;
    float decimal_host;
    const uint8_t* cmask;
    int saturate;
    int32_t lo, hi;
    ActSpec act;
    int act_dt;
    struct P {
        float toi, tof, keep;
    };
    __device__ __forceinline__ static float pow2(float d) {
        // 2.0 ** d: exact for integral d (what DecimalQuantizer produces)
        return (d == rintf(d) && fabsf(d) < 150.0f) ? ldexpf(1.0f, (int)d) : exp2f(d);
    }
    __device__ __forceinline__ P channel(uint32_t c) const {
        P p;
        float d = decimal ? decimal[c] : decimal_host;
        p.toi = pow2(d);
        p.tof = pow2(-d);
        p.keep = 1.0f;
        return p;
    }
    __device__ __forceinline__ P channel_masked(uint32_t c_par, uint32_t c_mask) const {
        P p = channel(c_par);
        if (cmask) p.keep = keep_from_byte(cmask[c_mask]);
        return p;
    }
    __host__ __device__ __forceinline__ const uint8_t* mask_ptr() const { return cmask; }
    __device__ __forceinline__ static P keep_of(P p, uint32_t m) {
        p.keep = keep_from_byte(m);
        return p;
    }
    __device__ __forceinline__ float apply(float v, const P& p, int32_t& code) const {
        if (act.kind) v = act_apply(v, act, act_dt);
        v = v * p.keep;
        float q = round_through<QDT>(v * p.toi);
        int32_t qi = f32_to_i32_x86(q);          // .int(): truncation toward zero (:55)
        if (saturate) qi = qi < lo ? lo : (qi > hi ? hi : qi);
        code = qi;
        return (float)qi * p.tof;
    }
};

// LineQuantization.forward  (reference qsparse/quantize.py:148-181)
template <bool FLOAT_ZP>
struct LineFwdOp {
    static constexpr bool kHasMask = false;
    const float* lines;  // device [nlines, 2]
    float nlevels;       // 2^bits
    float inv_levels;    // 2^-bits (exact)
    struct P {
        float start, end, step, qstart, r;
    };
    __device__ __forceinline__ P channel(uint32_t c) const {
        P p;
        p.start = lines[2 * c];
        p.end = lines[2 * c + 1];
        float st = (p.end - p.start) * inv_levels;       // :159; nlevels = 2^bits, so the product IS the correctly rounded quotient
        p.step = (st == 0.0f) ? 0.0001f : st;            // :160
        p.r = guarded_reciprocal(p.step);                // every quotient below goes through rint_of_quotient
        p.qstart = FLOAT_ZP ? 0.0f : rint_of_quotient(p.start, p.step, p.r);
        return p;
    }
    __device__ __forceinline__ P channel_masked(uint32_t c, uint32_t) const { return channel(c); }
    __host__ __device__ __forceinline__ const uint8_t* mask_ptr() const { return nullptr; }
    __device__ __forceinline__ static P keep_of(P p, uint32_t) { return p; }
    __device__ __forceinline__ float apply(float v, const P& p, int32_t& code) const {
        if (v != v) {                                    // torch.clamp and everything after it propagate NaN
            code = (int32_t)0x80000000;
            return v;
        }
        float xc = fminf(fmaxf(v, p.start), p.end);      // torch.clamp(x, start, end) (:158)
        const float top = nlevels - 1.0f;
        if constexpr (FLOAT_ZP) {                        // :175-181
            float t = xc - p.start;
            t = rint_of_quotient(t, p.step, p.r);        // ((x - start) / step).round()
            t = fminf(fmaxf(t, 0.0f), top);
            code = (int32_t)t;
            t = t * p.step;
            return t + p.start;
        } else {                                         // :161-166
            float qa = rint_of_quotient(xc, p.step, p.r);  // (x / step).round()
            qa = fminf(fmaxf(qa - p.qstart, 0.0f), top);
            code = (int32_t)qa;
            return (qa + p.qstart) * p.step;
        }
    }
};

// Scaler/DecimalQuantization.backward (reference qsparse/quantize.py:66-77, 120-131), optionally fused
// with the PruneLayer backward g * mask.
struct SteBwdOp {
    static constexpr bool kHasMask = true;
    const float* step;
    float step_host;
    int step_is_decimal;
    float lo_mul, hi_mul;
    int passthrough;
    const uint8_t* cmask;
    struct P {
        float lo, hi, keep;
    };
    // The reference clamps in place and then runs `v[v != grad_output] = 0` with v BEING grad_output (quantize.py:72-76,
    // 126-130): the comparison is true for NaNs only, so a NaN gradient becomes +0.0 -- and so does every element once a NaN
    // scale (a NaN / Inf input reached it) has made the bounds NaN, since ATen's tensor-bound clamp returns NaN then (fixture
    // F17).  NaN bounds are replaced by [+0.0, +0.0] here, per channel: clamp(g, +0, +0) is +0.0 for every g that is not NaN.
    // Mask bytes are 0 / 1 (torch.bool): the factor is a byte-to-float conversion.
    __device__ __forceinline__ static float factor(const P&, uint32_t m) { return (float)m; }
    __device__ __forceinline__ P channel(uint32_t c) const {
        P p;
        float s = step ? step[c] : step_host;
        if (step_is_decimal) s = DecimalFwdOp<QS_F32>::pow2(-s);
        p.lo = passthrough ? -__builtin_inff() : lo_mul * s;
        p.hi = passthrough ? __builtin_inff() : hi_mul * s;
        if (p.lo != p.lo || p.hi != p.hi) p.lo = p.hi = 0.0f;
        p.keep = 1.0f;
        return p;
    }
    __device__ __forceinline__ P channel_masked(uint32_t c_par, uint32_t c_mask) const {
        P p = channel(c_par);
        if (cmask) p.keep = factor(p, cmask[c_mask]);
        return p;
    }
    __host__ __device__ __forceinline__ const uint8_t* mask_ptr() const { return cmask; }
    __device__ __forceinline__ static P keep_of(P p, uint32_t m) {
        p.keep = factor(p, m);
        return p;
    }
    __device__ __forceinline__ float apply(float g, const P& p, int32_t& code) const {
        float v = g;
        if (!passthrough) {
            v = fminf(fmaxf(g, p.lo), p.hi);             // clamp_(lo, hi) == min(max(g, lo), hi)
            if (g != g) v = 0.0f;                        // clamp keeps a NaN; `v[v != v] = 0` then zeroes it
        }
        code = 0;
        return v * p.keep;                               // g * mask keeps the sign of zero
    }
};

// x * mask with a per-channel mask (reference qsparse/sparse.py:66,116,122,263)
struct ChanMaskOp {
    static constexpr bool kHasMask = true;
    const uint8_t* cmask;
    ActSpec act;   // act(x) * mask: a preceding activation folded into the prune site (nn.ReLU: max(x, 0) * mask)
    int act_dt;
    struct P {
        float keep;
    };
    __device__ __forceinline__ P channel(uint32_t c) const {
        P p;
        p.keep = cmask[c] ? 1.0f : 0.0f;
        return p;
    }
    __device__ __forceinline__ P channel_masked(uint32_t, uint32_t c_mask) const { return channel(c_mask); }
    __host__ __device__ __forceinline__ const uint8_t* mask_ptr() const { return cmask; }
    __device__ __forceinline__ static P keep_of(P p, uint32_t m) {
        p.keep = m ? 1.0f : 0.0f;
        return p;
    }
    __device__ __forceinline__ float apply(float v, const P& p, int32_t& code) const {
        code = 0;
        if (act.kind) v = act_apply(v, act, act_dt);   // (ATen's CPU relu, clamp_min = max_ps(0, x): -0.0 and NaN pass through)
        return v * p.keep;
    }
};

// ------------------------------------------------------------------------------------------------
// Gate bitmap of a folded ReLU.  A forward op wrapped in GateOp also records, for every element in MEMORY order, whether
// the ReLU lets the gradient through: bit (e & 7) of gate[e >> 3] = !(x[e] <= 0) (ATen's threshold_backward: NaN passes).
// The fused backward then reads g and one BIT per element instead of g and x -- 2 or 4 bytes per element less, the
// largest avoidable stream of a ReLU -> prune -> quantize site -- and x need not be kept for the backward at all.
// Every kernel path hands a lane 8 consecutive elements (one byte) or 4 (a nibble; lanes 2k and 2k+1 share a byte and
// combine it with one DPP move).  Never combined with ELIDE: a lane that skips its load cannot know its gate bits
// (`zero_pruned` keeps the eliding kernels' arithmetic instead).
// ------------------------------------------------------------------------------------------------
template <typename Base>
struct GateOp : Base {
    uint8_t* gate;      // ceil(numel / 8) bytes
    int zero_pruned;    // the caller asked for elision: a pruned channel's x counts as +0.0, as in the eliding kernels, so that
                        // a site gives the same bits with and without the bitmap (NaN / Inf on a pruned channel, quirk B15)
    void* image;        // nullable: the low-precision IMAGE of the float32 output -- RNE(y) in image_dt, the very cast autocast
    int image_dt;       // applies to y in front of a convolution -- written by the same pass (+2 B/elem instead of a 6 B/elem pass)
    void* xback;        // nullable: relu(x) written back, in x's dtype, to wherever the caller says -- x's own storage for an
                        // nn.ReLU(inplace=True) whose result other holders of x must see: the ReLU's forward pass costs one more
                        // store in this kernel instead of a read + write pass of its own (widening kernels only)
    __device__ __forceinline__ float apply(float v, const typename Base::P& p, int32_t& code) const {
        if (zero_pruned && !needs_load(p)) v = 0.0f;
        return Base::apply(v, p, code);
    }
};
// N (4 or 8) consecutive results starting at element e (a multiple of N) into the image: one 8- / 16-byte store
template <int N>
__device__ __forceinline__ void image_store(void* img, int dt, int64_t e, const float* r) {
    uint32_t w[N / 2];
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        const uint32_t lo = (dt == QS_BF16) ? f32_to_bf16_bits(r[2 * j]) : f32_to_f16_bits(r[2 * j]);
        const uint32_t hi = (dt == QS_BF16) ? f32_to_bf16_bits(r[2 * j + 1]) : f32_to_f16_bits(r[2 * j + 1]);
        w[j] = lo | (hi << 16);
    }
    if constexpr (N == 8) *(u32x4*)((uint16_t*)img + e) = u32x4{w[0], w[1], w[2], w[3]};
    else *(u32x2*)((uint16_t*)img + e) = u32x2{w[0], w[1]};
}
// act(v[0..N)) (nn.ReLU: ATen's clamp_min, NaN and -0.0 pass) back in dtype DT at element e (a multiple of N): one store of N
template <int N, int DT>
__device__ __forceinline__ void xback_store(void* xb, int64_t e, const float* v, const ActSpec& act) {
    if constexpr (DT == QS_F32) {
        static_assert(N == 4, "fp32 inputs take the 4-elements-per-lane paths");
        *(u32x4*)((float*)xb + e) = u32x4{__float_as_uint(act_apply(v[0], act, DT)), __float_as_uint(act_apply(v[1], act, DT)),
                                          __float_as_uint(act_apply(v[2], act, DT)), __float_as_uint(act_apply(v[3], act, DT))};
    } else {
        uint32_t w[N / 2];
#pragma unroll
        for (int j = 0; j < N / 2; ++j) {   // (the values are DT values already: the conversion is exact)
            const float a = act_apply(v[2 * j], act, DT), b = act_apply(v[2 * j + 1], act, DT);
            const uint32_t lo = (DT == QS_BF16) ? f32_to_bf16_bits(a) : f32_to_f16_bits(a);
            const uint32_t hi = (DT == QS_BF16) ? f32_to_bf16_bits(b) : f32_to_f16_bits(b);
            w[j] = lo | (hi << 16);
        }
        if constexpr (N == 8) *(u32x4*)((uint16_t*)xb + e) = u32x4{w[0], w[1], w[2], w[3]};
        else *(u32x2*)((uint16_t*)xb + e) = u32x2{w[0], w[1]};
    }
}
template <typename Op>
struct OpGate {
    static constexpr bool value = false;
    __device__ __forceinline__ static uint8_t* ptr(const Op&) { return nullptr; }
    __device__ __forceinline__ static void* image(const Op&) { return nullptr; }
    __device__ __forceinline__ static int image_dt(const Op&) { return QS_BF16; }
    __device__ __forceinline__ static void* xback(const Op&) { return nullptr; }
    __device__ __forceinline__ static ActSpec act(const Op&) { return ActSpec{0, 0.f, 0.f}; }
};
template <typename Base>
struct OpGate<GateOp<Base>> {
    static constexpr bool value = true;
    __device__ __forceinline__ static uint8_t* ptr(const GateOp<Base>& op) { return op.gate; }
    __device__ __forceinline__ static void* image(const GateOp<Base>& op) { return op.image; }
    __device__ __forceinline__ static int image_dt(const GateOp<Base>& op) { return op.image_dt; }
    __device__ __forceinline__ static void* xback(const GateOp<Base>& op) { return op.xback; }
    __device__ __forceinline__ static ActSpec act(const GateOp<Base>& op) { return op.act; }
};
template <int N>
__device__ __forceinline__ uint32_t gate_bits(const float* v, const ActSpec& act) {
    uint32_t b = 0u;
    if (act.kind <= QS_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < N; ++j) b |= (v[j] <= 0.0f ? 0u : 1u) << j;
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) b |= (act_open(v[j], act) ? 1u : 0u) << j;
    }
    return b;
}
// the nibble of the neighbouring lane (lane ^ 1): quad_perm [1, 0, 3, 2]
__device__ __forceinline__ uint32_t gate_pair_swap(uint32_t nib) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nib, 0xB1, 0xF, 0xF, false);
}
// ragged tail: the (numel % 8) threads that serve it are lanes 0.. of one wave; lane 0 stores their ballot
__device__ __forceinline__ void gate_store_tail(uint8_t* gate, int64_t ngroups, bool open) {
    const uint64_t b = __ballot(open);
    if (threadIdx.x == 0) gate[ngroups] = (uint8_t)(b & 0xffu);
}

// ------------------------------------------------------------------------------------------------
// Mask-aware traffic elision (ELIDE).  With a channel mask the input of a pruned channel only ever meets
// `* 0`: the mask byte is read FIRST and the load of x (or g) is skipped for lanes whose elements are all pruned; the
// op is applied to +0.0 instead.  At 75 % channel sparsity the fused forward reads a quarter of x.  Exactness:
//   * quantizer forward: bit-identical to the loading path for every finite x (x*0 = +-0 -> code 0 -> f32(0)*s),
//     which is why it is the default; a NaN / Inf on a PRUNED channel gives f32(0)*s here and INT_MIN*s in the
//     reference (quirk B15) -- the loading path stays available (elide_masked = 0).
//   * mask apply and the backward kernels: the reference's g*0 / x*0 keeps the sign of its operand, the elided result
//     is always +0.0 -- numerically equal, not bit-identical, hence opt-in.
// Lanes decide individually (a divergent skip still saves the HBM lines no active lane touches); rows of at least
// 8 elements are needed for a lane to be all-pruned in the [outer, C, inner] layout, 8 consecutive pruned channels
// in the channels_last layout (CM_LAST), where the gain is accordingly small.
// ------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ Raw8<DT> zero_raw8() {
    Raw8<DT> r;
    r.a = u32x4{0u, 0u, 0u, 0u};
    r.b = u32x4{0u, 0u, 0u, 0u};
    return r;
}

// first channel of element `e` (a multiple of 4) in the channel-innermost layout (CM_LAST: C % 8 == 0): 32-bit
// arithmetic on the 8-element group index (< 2^32 by plan_ew) instead of a 64-bit modulo per lane
__device__ __forceinline__ uint32_t last_dim_channel(int64_t e, uint32_t C) {
    return (((uint32_t)((uint64_t)e >> 3) % (C >> 3)) << 3) + ((uint32_t)e & 7u);
}

// whether an eliding kernel may skip this lane (its parameter block says the channel is pruned); false for dense kernels
// and for ops without a mask (their P has no `keep`)
template <bool ELIDE, typename P>
__device__ __forceinline__ bool lane_pruned(const P& p) {
    if constexpr (ELIDE) return !needs_load(p);
    else return false;
}

// One mask byte through the SCALAR cache (s_load_dword of the aligned word that holds it): `c` must be wave-uniform.
// The vector memory pipeline of a streaming kernel is full of non-temporal stores, and a per-lane mask load queued
// behind them delays every wave by a memory round trip before it can even decide what to load (measured: elided
// forward no faster than the dense one); the scalar path is separate and short.
typedef const __attribute__((address_space(4))) uint32_t* qs_const_u32_ptr;
__device__ __forceinline__ uint32_t sload_mask_byte(const uint8_t* m, uint32_t c_uniform) {
    const uintptr_t a = (uintptr_t)m + c_uniform;
    const uint32_t w = *(qs_const_u32_ptr)(a & ~(uintptr_t)3);
    return (w >> (8u * (uint32_t)(a & 3u))) & 0xffu;
}

// The at most two rows a wave's 64 consecutive 8-element groups touch when a row has at least 64 groups, with their
// mask bytes: wave-uniform values (SGPRs).  `gw` = first group of the wave (uniform).
struct WaveRows {
    uint32_t c0, c1, k0, k1, split;   // channels, mask bytes, number of the wave's groups that lie in the first row
    __device__ __forceinline__ void seek(uint32_t gw, uint32_t groups_per_row, uint32_t C, const uint8_t* mask) {
        const uint32_t row0 = gw / groups_per_row;
        split = groups_per_row - (gw - row0 * groups_per_row);
        c0 = row0 % C;
        c1 = c0 + 1u == C ? 0u : c0 + 1u;
        k0 = k1 = 1u;
        if (mask) {
            k0 = sload_mask_byte(mask, c0);
            k1 = sload_mask_byte(mask, c1);
        }
    }
    __device__ __forceinline__ bool all_pruned() const { return k0 == 0u && (split >= 64u || k1 == 0u); }
};

// ------------------------------------------------------------------------------------------------
// The streaming kernel.  PARAM_PER_CHANNEL tells whether the op's parameter array is indexed by the
// channel (nparam == C) or is a single value; the channel mask, when present, is always per channel.
// ------------------------------------------------------------------------------------------------
template <typename Op, int XDT, int YDT, int CM, bool PARAM_PER_CHANNEL, bool NT, int UNROLL, bool ELIDE = false>
__global__ __launch_bounds__(kBlock) void ew_kernel(Op op, EwGeom geo, const void* __restrict__ x,
                                                     void* __restrict__ y, int32_t* __restrict__ codes) {
    static_assert(!ELIDE || (Op::kHasMask && CM != CM_SCALAR), "elision needs a channel mask");
    constexpr bool GATE = OpGate<Op>::value;
    static_assert(!(GATE && ELIDE), "a lane that skips its load cannot record the ReLU gate");
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t blk = geo.reverse ? (int64_t)(gridDim.x - 1 - blockIdx.x) : (int64_t)blockIdx.x;
    int64_t g0 = blk * kBlock + threadIdx.x;

    typename Op::P p_scalar = op.channel(0);  // used as is in CM_SCALAR

    for (; g0 < geo.ngroups; g0 += stride * UNROLL) {
        Raw8<XDT> raw[UNROLL];
        if constexpr (CM == CM_ROW) {
            // Three phases over the lane's UNROLL groups, so that their latencies overlap instead of adding up (a wave
            // of an eliding kernel mostly waits: mask look-up -> load or nothing -> store): (A) all parameter / mask
            // look-ups -- wave-uniform through the scalar cache for rows of >= 512 elements, which also replaces a
            // 64-bit division and a byte load per LANE by one 32-bit division per WAVE --, (B) all loads (ELIDE: only
            // the needed ones), (C) compute + store.  With ELIDE a pruned lane applies the op ONCE to +0.0 and stores the
            // result eight times.  The dense kernels take the same path: with two groups per lane the all-kept backward
            // measured 0.2000 ms against 0.2042 ms for the one-group, per-lane look-up version it replaces.
            typename Op::P pp[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t g = g0 + u * stride;
                pp[u] = p_scalar;
                if (g < geo.ngroups) {
                    if (geo.groups_per_row >= 64u) {    // rows of >= 512 elements: wave-uniform look-up through the scalar cache
                        const uint32_t lane = threadIdx.x & 63u;
                        WaveRows wr;
                        wr.seek(__builtin_amdgcn_readfirstlane((uint32_t)g - lane), geo.groups_per_row, geo.C, op.mask_ptr());
                        const bool first = lane < wr.split;
                        pp[u] = Op::keep_of(op.channel(PARAM_PER_CHANNEL ? (first ? wr.c0 : wr.c1) : 0u), first ? wr.k0 : wr.k1);
                    } else {
                        const uint32_t c = (uint32_t)((uint64_t)g / geo.groups_per_row) % geo.C;
                        pp[u] = op.channel_masked(PARAM_PER_CHANNEL ? c : 0u, c);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t g = g0 + u * stride;
                raw[u] = zero_raw8<XDT>();
                if (g < geo.ngroups && !lane_pruned<ELIDE>(pp[u])) raw[u] = load8_raw<XDT, NT>(x, g);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t g = g0 + u * stride;
                if (g >= geo.ngroups) break;
                float v[8];
                int32_t q[8];
                if (lane_pruned<ELIDE>(pp[u])) {
                    const float z = op.apply(0.0f, pp[u], q[0]);
                    store8_splat<YDT, NT>(y, g, z);                 // one conversion, eight copies
                    if (codes) {
#pragma unroll
                        for (int j = 1; j < 8; ++j) q[j] = q[0];
                        store8_i32(codes, g, q);
                    }
                    continue;
                }
                unpack8<XDT>(raw[u], v);
                if constexpr (GATE) OpGate<Op>::ptr(op)[g] = (uint8_t)gate_bits<8>(v, OpGate<Op>::act(op));
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], pp[u], q[j]);
                store8<YDT, NT>(y, g, v);
                if (codes) store8_i32(codes, g, q);
            }
            continue;
        }
        if constexpr (!ELIDE) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t g = g0 + u * stride;
                if (g < geo.ngroups) raw[u] = load8_raw<XDT, NT>(x, g);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t g = g0 + u * stride;
            if (g >= geo.ngroups) break;
            float v[8];
            int32_t q[8];
            if constexpr (!ELIDE) unpack8<XDT>(raw[u], v);
            if constexpr (GATE) OpGate<Op>::ptr(op)[g] = (uint8_t)gate_bits<8>(v, OpGate<Op>::act(op));
            if constexpr (CM == CM_SCALAR) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], p_scalar, q[j]);
            } else if constexpr (CM == CM_LAST) {
                const uint32_t c0 = last_dim_channel(g * 8, geo.C);
                const uint8_t* mp = op.mask_ptr();
                u32x2 mm = {0x01010101u, 0x01010101u};
                if (mp) mm = *(const u32x2*)(mp + c0);
                if constexpr (ELIDE) {      // 8 consecutive channels, all pruned
                    raw[u] = zero_raw8<XDT>();
                    if ((mm[0] | mm[1]) != 0u) raw[u] = load8_raw<XDT, NT>(x, g);
                    unpack8<XDT>(raw[u], v);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = op.apply(v[j], Op::keep_of(p_scalar, (mm[j >> 2] >> (8 * (j & 3))) & 0xffu), q[j]);
            } else {
                ChanIter it;
                it.C = geo.C;
                it.inner = geo.inner;
                it.seek((uint64_t)g * 8);
                if (geo.inner >= 8) {
                    // rows of at least 8 elements: a lane's 8 elements lie in at most two rows -- two parameter / mask
                    // look-ups instead of eight (ragged maps such as 7x7 and 14x14)
                    const uint32_t left = geo.inner - it.r;            // elements of the first row
                    const uint32_t c1 = it.c + 1 == geo.C ? 0u : it.c + 1;
                    const typename Op::P p0 = op.channel_masked(PARAM_PER_CHANNEL ? it.c : 0u, it.c);
                    const typename Op::P p1 = op.channel_masked(PARAM_PER_CHANNEL ? c1 : 0u, c1);      // unconditional: no divergent branch
                    if constexpr (ELIDE) {
                        raw[u] = zero_raw8<XDT>();
                        if (needs_load(p0) || (left < 8u && needs_load(p1))) raw[u] = load8_raw<XDT, NT>(x, g);
                        unpack8<XDT>(raw[u], v);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], (uint32_t)j < left ? p0 : p1, q[j]);
                } else {
                    if constexpr (ELIDE) unpack8<XDT>(load8_raw<XDT, NT>(x, g), v);   // rows shorter than a lane's 8 elements
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        typename Op::P p = op.channel_masked(PARAM_PER_CHANNEL ? it.c : 0u, it.c);
                        v[j] = op.apply(v[j], p, q[j]);
                        it.next();
                    }
                }
            }
            store8<YDT, NT>(y, g, v);
            if (codes) store8_i32(codes, g, q);
        }
    }

    // ragged tail (numel % 8 elements), scalar accesses
    const int64_t tail0 = geo.ngroups * 8;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < geo.numel) {
        const int64_t e = tail0 + threadIdx.x;
        ChanIter it;
        it.C = geo.C;
        it.inner = geo.inner;
        it.seek((uint64_t)e);
        typename Op::P p = (CM == CM_SCALAR) ? p_scalar : op.channel_masked(PARAM_PER_CHANNEL ? it.c : 0u, it.c);
        int32_t qi;
        const float xe = load1<XDT>(x, e);
        if constexpr (GATE) gate_store_tail(OpGate<Op>::ptr(op), geo.ngroups, act_open(xe, OpGate<Op>::act(op)));
        float r = op.apply(xe, p, qi);
        store1<YDT>(y, e, r);
        if (codes) codes[e] = qi;
    }
}

// ------------------------------------------------------------------------------------------------
// fp32-output variant (2-byte or fp32 input -> fp32 output): every store instruction of a wave covers ONE contiguous
// 1 KiB span (lane l writes 16 B at l*16), instead of 16 B at a 32-B stride.
// A wave owns 512 consecutive elements: lane l stores elements [4l, 4l+4) and [256+4l, 256+4l+4).  2-byte inputs are
// LOADED 8 consecutive elements per lane (one 16-byte load) and transposed through LDS (below); fp32 inputs, partial
// waves and ragged rows load what they store (8- / 16-byte loads per half).
// Measured on the headline tensor: 0.2165 -> 0.2025 ms (5.7 -> 6.1 TB/s).  The mirror image for the narrowing
// backward (contiguous 1 KiB loads, 8-byte stores) changed nothing (0.2045 vs 0.2042 ms) and was dropped:
// it is the stores whose per-instruction footprint matters.
// ------------------------------------------------------------------------------------------------
#ifndef QS_WIDEN_BLOCK
#define QS_WIDEN_BLOCK 256
#endif
#ifndef QS_WIDEN_NT_STORE
#define QS_WIDEN_NT_STORE 1
#endif
#ifndef QS_WIDEN_PRUNED_NT_STORE
#define QS_WIDEN_PRUNED_NT_STORE QS_WIDEN_NT_STORE   // the store-only waves of an eliding forward (all rows of the wave pruned)
#endif
constexpr int kWidenBlock = QS_WIDEN_BLOCK;   // threads per workgroup of the widening kernel (a wave owns 512 elements)

template <typename Op, int XDT, int CM, bool PARAM_PER_CHANNEL, bool NT, bool ELIDE = false>
__global__ __launch_bounds__(kWidenBlock) void ew_widen_kernel(Op op, EwGeom geo, const void* __restrict__ x,
                                                           float* __restrict__ y) {
    static_assert(!ELIDE || (Op::kHasMask && (CM == CM_ROW || CM == CM_LAST)), "elision needs a channel mask");
    constexpr bool GATE = OpGate<Op>::value;
    static_assert(!(GATE && ELIDE), "a lane that skips its load cannot record the ReLU gate");
    const int64_t blk = geo.reverse ? (int64_t)(gridDim.x - 1 - blockIdx.x) : (int64_t)blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e_wave = (blk * (kWidenBlock / 64) + wave) * 512;          // first element of this wave
    typename Op::P p_scalar = op.channel(0);
    bool done = false;
    if constexpr (XDT != QS_F32) {
        // 2-byte inputs, whole wave inside the tensor: ONE 16-byte load per lane (1 KiB per wave in flight instead of
        // two dependent 512-byte rounds) and the 8 results transposed through LDS, so that each of the lane's two
        // 16-byte stores still covers one contiguous 1 KiB span per instruction.  The kernel is bound by the bytes its
        // resident waves keep in flight (capping it at 4 / 2 waves per SIMD costs 1.55x / 2.3x): headline forward
        // 0.2008 -> 0.1898 ms (6.1 -> 6.5 TB/s).
        __shared__ __attribute__((aligned(16))) float stage[kWidenBlock * 8];
        if (e_wave + 512 <= (int64_t)geo.ngroups * 8 && (CM != CM_ROW || geo.inner % 8 == 0)) {
            float* ws = stage + wave * 512;
            const int64_t e = e_wave + lane * 8;
            float v[8];
            if constexpr (!ELIDE) unpack8<XDT>(load8_raw<XDT, NT>(x, e / 8), v);
            if constexpr (GATE) {
                OpGate<Op>::ptr(op)[e >> 3] = (uint8_t)gate_bits<8>(v, OpGate<Op>::act(op));
                if (void* xb = OpGate<Op>::xback(op)) xback_store<8, XDT>(xb, e, v, OpGate<Op>::act(op));
            }
            int32_t q;
            u32x4 a, b;
            if constexpr (CM == CM_LAST) {
                const uint8_t* mp = op.mask_ptr();
                u32x2 mm = {0x01010101u, 0x01010101u};
                if (mp) mm = *(const u32x2*)(mp + last_dim_channel(e, geo.C));   // 8 consecutive channels
                if constexpr (ELIDE) {
                    Raw8<XDT> r = zero_raw8<XDT>();
                    if ((mm[0] | mm[1]) != 0u) r = load8_raw<XDT, NT>(x, e / 8);
                    unpack8<XDT>(r, v);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = __float_as_uint(op.apply(v[j], Op::keep_of(p_scalar, (mm[0] >> (8 * j)) & 0xffu), q));
                    b[j] = __float_as_uint(op.apply(v[4 + j], Op::keep_of(p_scalar, (mm[1] >> (8 * j)) & 0xffu), q));
                }
            } else {
                typename Op::P p = p_scalar;
                if constexpr (CM == CM_ROW && ELIDE) {      // the mask byte first; the x of a pruned row is never loaded
                    if (geo.groups_per_row >= 64u) {        // rows of >= 512 elements: wave-uniform look-up through the scalar cache
                        WaveRows wr;
                        wr.seek(__builtin_amdgcn_readfirstlane((uint32_t)(e_wave >> 3)), geo.groups_per_row, geo.C, op.mask_ptr());
                        if (wr.all_pruned()) {
                            // nothing of this wave is kept: no load, no LDS round trip -- store Q(+0.0) of each element's row.
                            // Lane l stores elements [4l, 4l+4) and [256+4l, 256+4l+4); rows change at a multiple of 8 elements.
                            const float z0 = op.apply(0.0f, Op::keep_of(op.channel(PARAM_PER_CHANNEL ? wr.c0 : 0u), 0u), q);
                            const float z1 = op.apply(0.0f, Op::keep_of(op.channel(PARAM_PER_CHANNEL ? wr.c1 : 0u), 0u), q);
                            const uint32_t split_e = wr.split >= 64u ? 512u : wr.split * 8u;
                            const uint32_t za = __float_as_uint((uint32_t)lane * 4u < split_e ? z0 : z1);
                            const uint32_t zb = __float_as_uint(256u + (uint32_t)lane * 4u < split_e ? z0 : z1);
                            st16<(NT && QS_WIDEN_PRUNED_NT_STORE != 0)>((u32x4*)(y + e_wave + lane * 4), u32x4{za, za, za, za});
                            st16<(NT && QS_WIDEN_PRUNED_NT_STORE != 0)>((u32x4*)(y + e_wave + 256 + lane * 4), u32x4{zb, zb, zb, zb});
                            return;     // (inner % 8 == 0 here, so numel % 8 == 0: there is no ragged tail for this wave to serve)
                        }
                        const bool first = (uint32_t)lane < wr.split;
                        p = Op::keep_of(op.channel(PARAM_PER_CHANNEL ? (first ? wr.c0 : wr.c1) : 0u), first ? wr.k0 : wr.k1);
                    } else {
                        const uint32_t c = (uint32_t)((uint64_t)e / geo.inner) % geo.C;
                        p = op.channel_masked(PARAM_PER_CHANNEL ? c : 0u, c);
                    }
                    Raw8<XDT> r = zero_raw8<XDT>();
                    if (needs_load(p)) r = load8_raw<XDT, NT>(x, e / 8);
                    unpack8<XDT>(r, v);
                } else if constexpr (CM == CM_ROW) {
                    if (geo.groups_per_row >= 64u) {        // the dense kernel takes the wave-uniform look-up as well
                        WaveRows wr;
                        wr.seek(__builtin_amdgcn_readfirstlane((uint32_t)(e_wave >> 3)), geo.groups_per_row, geo.C, op.mask_ptr());
                        const bool first = (uint32_t)lane < wr.split;
                        p = Op::keep_of(op.channel(PARAM_PER_CHANNEL ? (first ? wr.c0 : wr.c1) : 0u), first ? wr.k0 : wr.k1);
                    } else {
                        const uint32_t c = (uint32_t)((uint64_t)e / geo.inner) % geo.C;   // inner % 8 == 0: one row per lane
                        p = op.channel_masked(PARAM_PER_CHANNEL ? c : 0u, c);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = __float_as_uint(op.apply(v[j], p, q));
                    b[j] = __float_as_uint(op.apply(v[4 + j], p, q));
                }
            }
            if constexpr (GATE) {
                if (void* img = OpGate<Op>::image(op)) {   // the lane's 8 results are 8 consecutive elements: one 16-byte store
                    float r[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        r[j] = __uint_as_float(a[j]);
                        r[4 + j] = __uint_as_float(b[j]);
                    }
                    image_store<8>(img, OpGate<Op>::image_dt(op), e, r);
                }
            }
            *(u32x4*)(ws + lane * 8) = a;              // wave-private LDS region: no workgroup barrier
            *(u32x4*)(ws + lane * 8 + 4) = b;
            __builtin_amdgcn_wave_barrier();
            const u32x4 o0 = *(const u32x4*)(ws + lane * 4);
            const u32x4 o1 = *(const u32x4*)(ws + 256 + lane * 4);
            st16<(NT && QS_WIDEN_NT_STORE != 0)>((u32x4*)(y + e_wave + lane * 4), o0);
            st16<(NT && QS_WIDEN_NT_STORE != 0)>((u32x4*)(y + e_wave + 256 + lane * 4), o1);
            done = true;
        }
    }
#pragma unroll
    for (int half = 0; half < 2 && !done; ++half) {
        const int64_t e = e_wave + half * 256 + lane * 4;
        if (e + 4 <= geo.ngroups * 8) {
            float v[4];
            u32x2 raw = {0u, 0u};
            bool need = true;                // ELIDE: whether any of the lane's 4 elements is kept
            typename Op::P p = p_scalar;
            uint32_t mm = 0x01010101u;
            if constexpr (ELIDE) {
                if constexpr (CM == CM_ROW) {
                    const uint32_t c = (uint32_t)((uint64_t)e / geo.inner) % geo.C;
                    p = op.channel_masked(PARAM_PER_CHANNEL ? c : 0u, c);
                    need = needs_load(p);
                } else {
                    const uint8_t* mp = op.mask_ptr();
                    if (mp) mm = *(const uint32_t*)(mp + last_dim_channel(e, geo.C));
                    need = mm != 0u;
                }
            }
            if constexpr (XDT == QS_F32) {   // fp32 -> fp32: 16-byte loads and stores, both one contiguous 1 KiB span
                u32x4 r4 = {0u, 0u, 0u, 0u};
                if (need) r4 = ld16<NT>((const u32x4*)((const float*)x + e));
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(r4[j]);
            } else if (need) {
                raw = NT ? __builtin_nontemporal_load((const u32x2*)((const uint16_t*)x + e))
                         : *(const u32x2*)((const uint16_t*)x + e);
            }
            if constexpr (XDT == QS_F32) {
            } else if constexpr (XDT == QS_BF16) {
                v[0] = __uint_as_float(raw[0] << 16);
                v[1] = __uint_as_float(raw[0] & 0xffff0000u);
                v[2] = __uint_as_float(raw[1] << 16);
                v[3] = __uint_as_float(raw[1] & 0xffff0000u);
            } else {
                v[0] = f16_bits_to_f32(raw[0] & 0xffffu);
                v[1] = f16_bits_to_f32(raw[0] >> 16);
                v[2] = f16_bits_to_f32(raw[1] & 0xffffu);
                v[3] = f16_bits_to_f32(raw[1] >> 16);
            }
            if constexpr (!ELIDE) {
                if constexpr (CM == CM_ROW) {
                    const uint32_t c = (uint32_t)((uint64_t)e / geo.inner) % geo.C;   // inner % 4 == 0: 4 elements share a row
                    p = op.channel_masked(PARAM_PER_CHANNEL ? c : 0u, c);
                }
                if constexpr (CM == CM_LAST) {
                    const uint8_t* mp = op.mask_ptr();
                    if (mp) mm = *(const uint32_t*)(mp + last_dim_channel(e, geo.C));   // 4 consecutive channels
                }
            }
            if constexpr (GATE) {       // 4 elements = a nibble; the even lane stores the byte it shares with its neighbour
                const uint32_t nib = gate_bits<4>(v, OpGate<Op>::act(op));
                const uint32_t other = gate_pair_swap(nib);      // both lanes of a pair are inside or outside the tensor together
                if ((lane & 1) == 0) OpGate<Op>::ptr(op)[e >> 3] = (uint8_t)(nib | (other << 4));
                if (void* xb = OpGate<Op>::xback(op)) xback_store<4, XDT>(xb, e, v, OpGate<Op>::act(op));
            }
            int32_t q;
            u32x4 out;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (CM == CM_LAST) out[j] = __float_as_uint(op.apply(v[j], Op::keep_of(p_scalar, (mm >> (8 * j)) & 0xffu), q));
                else out[j] = __float_as_uint(op.apply(v[j], p, q));
            }
            st16<(NT && QS_WIDEN_NT_STORE != 0)>((u32x4*)(y + e), out);
            if constexpr (GATE) {
                if (void* img = OpGate<Op>::image(op)) {
                    const float r[4] = {__uint_as_float(out[0]), __uint_as_float(out[1]), __uint_as_float(out[2]), __uint_as_float(out[3])};
                    image_store<4>(img, OpGate<Op>::image_dt(op), e, r);
                }
            }
        }
    }
    // ragged tail (numel % 8 elements)
    const int64_t tail0 = geo.ngroups * 8;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < geo.numel) {
        const int64_t e = tail0 + threadIdx.x;
        ChanIter it;
        it.C = geo.C;
        it.inner = geo.inner;
        it.seek((uint64_t)e);
        typename Op::P p = (CM == CM_SCALAR) ? p_scalar : op.channel_masked(PARAM_PER_CHANNEL ? it.c : 0u, it.c);
        int32_t qi;
        const float xe = load1<XDT>(x, e);
        if constexpr (GATE) {
            gate_store_tail(OpGate<Op>::ptr(op), geo.ngroups, act_open(xe, OpGate<Op>::act(op)));
            if (void* xb = OpGate<Op>::xback(op)) store1<XDT>(xb, e, act_apply(xe, OpGate<Op>::act(op), XDT));
        }
        y[e] = op.apply(xe, p, qi);
        if constexpr (GATE) {
            if (void* img = OpGate<Op>::image(op)) {
                if (OpGate<Op>::image_dt(op) == QS_BF16) store1<QS_BF16>(img, e, y[e]);
                else store1<QS_F16>(img, e, y[e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// STE backward with the folded ReLU's gate: gx = (x <= 0) ? 0 : clamp(g) * mask.  Two streamed inputs (the
// gradient and the ReLU's input), one output in x's dtype.  Same geometry and channel modes as ew_kernel.
// GATE: `x` is not the ReLU's input but the gate bitmap the forward recorded (GateOp above; one bit per element in
// memory order): the second stream shrinks from 2 / 4 bytes to one bit per element; XDT is then only the output dtype.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void gate_to_floats(uint32_t bits, float* vx) {
#pragma unroll
    for (int j = 0; j < N; ++j) vx[j] = ((bits >> j) & 1u) ? 1.0f : 0.0f;
}

// G2DT >= 0: a SECOND gradient stream `g2` of that dtype is added to `g` in float32 before the clamp, and `g` itself may be
// NULL (then g2 alone is the gradient).  This is autograd's accumulation of the two gradients a site's float32 output
// receives under autocast -- float32 from its float32 consumers, bf16 / fp16 from the convolution that consumed its
// low-precision image (see fused.py, "autocast image") -- evaluated in the kernel instead of by a cast pass plus an add pass.
//
// Riders of the ALL-fp32 form (GDT == XDT == fp32: the site behind a residual add whose result type promotion made float32), both
// nullable and wave-uniform:
//   g3      a THIRD gradient stream of g2's dtype, added BEFORE g2: (g + float(g3)) + float(g2) -- the SECOND autocast consumer of
//           the site's output (a down-sampling convolution next to the block's first one) handed its 2-byte gradient over as
//           well; autograd accumulates the consumers' shares in reverse order of their creation, so the first consumer's (g2)
//           is the last term (fused.py, "second image")
//   gx_img  RNE(gx) in img_dt (bf16 / fp16), written next to gx by the same pass (+2 B/elem): the gradient of the 2-byte operand
//           of the promoting add in front of the site, which ATen's AddBackward would produce with a 6 B/elem cast pass of gx
struct BwdRiders {
    const void* g3;
    void* gx_img;
    int img_dt;
};
// DACT (QS_DACT_GELU): `x` is the input of an activation the CALLER applied in front of the site (nn.GELU by ATen); the site's own
// gradient -- clamp(g) * mask, rounded to x's dtype as autograd hands it on -- is multiplied by that activation's derivative at x
// (gelu_grad_factor): gelu_backward(gx_site, x) without a pass of its own.  No gate, no folded activation of the site's own.
template <int GDT, int XDT, int CM, bool NT, bool ELIDE = false, bool GATE = false, int G2DT = -1, int DACT = 0>
__global__ __launch_bounds__(kBlock) void ste_relu_bwd_kernel(SteBwdOp op, EwGeom geo, int param_per_channel,
                                                              const void* __restrict__ g, const void* __restrict__ x,
                                                              void* __restrict__ gx, ActSpec act, const void* __restrict__ g2 = nullptr,
                                                              BwdRiders rd = BwdRiders{nullptr, nullptr, 0}) {
    static_assert(G2DT < 0 || (!ELIDE && GDT == QS_F32 && G2DT != QS_F32), "the second gradient is a 2-byte stream next to an fp32 one");
    static_assert(DACT == 0 || (!GATE && !ELIDE), "the caller's activation: its input is the second stream, every lane loads it");
    // the activation's backward at one element: `xv` is the activation's input -- or, with GATE, the recorded bit as 1.0 / 0.0;
    // `applied` the clamped, masked gradient.  Open gate: it passes; closed: 0 (rectifiers) or applied * slope (leaky)
    auto gated = [&](float xv, float applied) -> float {
        if constexpr (DACT == QS_DACT_GELU) return round_to_dtype(applied, XDT) * gelu_grad_factor(xv);
        const bool open = GATE ? (xv > 0.0f) : act_open(xv, act);
        return open ? applied : act_closed(applied, act, XDT);
    };
    const int64_t blk = geo.reverse ? (int64_t)(gridDim.x - 1 - blockIdx.x) : (int64_t)blockIdx.x;
    const int64_t grp = blk * kBlock + threadIdx.x;
    if constexpr (GDT == QS_F32 && XDT == QS_F32) {
        // all three streams fp32 (the site behind a residual add): a lane's 8 elements would be 32 bytes, i.e. two
        // 16-byte accesses per stream at a 32-byte stride across the wave.  As in ew_widen_kernel a wave owns 512
        // consecutive elements and lane l takes [4l, 4l+4) and [256+4l, 256+4l+4): every load and store instruction
        // covers one contiguous 1 KiB span (64x256x56x56: 113 -> 100 us, 5.4 -> 6.1 TB/s; tools/bench_relu_bwd.py).
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int64_t e_wave = (blk * (kBlock / 64) + wave) * 512;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int64_t e = e_wave + half * 256 + lane * 4;
            if (e + 4 <= geo.ngroups * 8) {
                u32x4 rg4 = {0u, 0u, 0u, 0u}, rx4 = {0u, 0u, 0u, 0u};
                bool need = true;            // ELIDE: pruned lanes load neither stream (x = +0 closes the gate: gx = +0)
                if constexpr (ELIDE && CM == CM_ROW) {
                    if (geo.groups_per_row >= 64u && geo.inner % 8u == 0u) {     // wave-uniform look-up, see WaveRows
                        WaveRows wr;
                        wr.seek(__builtin_amdgcn_readfirstlane((uint32_t)(e_wave >> 3)), geo.groups_per_row, geo.C, op.cmask);
                        need = ((uint32_t)(half * 256 + lane * 4) < wr.split * 8u ? wr.k0 : wr.k1) != 0u;
                    } else {
                        const uint32_t c = (uint32_t)((uint64_t)e / geo.inner) % geo.C;
                        need = op.cmask[c] != 0;
                    }
                } else if constexpr (ELIDE && CM == CM_LAST) {
                    need = *(const uint32_t*)(op.cmask + last_dim_channel(e, geo.C)) != 0u;
                }
                if (need) {
                    if (G2DT < 0 || g) rg4 = ld16<NT>((const u32x4*)((const float*)g + e));
                    if constexpr (G2DT >= 0) {      // 4 two-byte values of the second gradient: g + float(g2), or float(g2) alone
                        const bool has3 = rd.g3 != nullptr;
                        if (has3) {                  // ... with the third stream in between: (g + float(g3)) + float(g2)
                            const u32x2 r3 = *(const u32x2*)((const uint16_t*)rd.g3 + e);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t h = (j & 1) ? (r3[j >> 1] >> 16) : (r3[j >> 1] & 0xffffu);
                                const float v3 = (G2DT == QS_BF16) ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
                                rg4[j] = __float_as_uint(g ? __uint_as_float(rg4[j]) + v3 : v3);
                            }
                        }
                        const u32x2 r2 = *(const u32x2*)((const uint16_t*)g2 + e);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t h = (j & 1) ? (r2[j >> 1] >> 16) : (r2[j >> 1] & 0xffffu);
                            const float v2 = (G2DT == QS_BF16) ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
                            rg4[j] = __float_as_uint((g || has3) ? __uint_as_float(rg4[j]) + v2 : v2);
                        }
                    }
                    if constexpr (GATE) {
                        const uint32_t bits = (uint32_t)((const uint8_t*)x)[e >> 3] >> ((uint32_t)e & 4u);
#pragma unroll
                        for (int j = 0; j < 4; ++j) rx4[j] = ((bits >> j) & 1u) ? 0x3f800000u : 0u;
                    } else {
                        rx4 = ld16<NT>((const u32x4*)((const float*)x + e));
                    }
                }
                int32_t dummy;
                u32x4 out;
                if constexpr (CM == CM_LAST) {
                    const SteBwdOp::P p0 = op.channel(0);
                    uint32_t mm = 0x01010101u;
                    if (op.cmask) mm = *(const uint32_t*)(op.cmask + last_dim_channel(e, geo.C));   // 4 consecutive channels
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        out[j] = __float_as_uint(gated(__uint_as_float(rx4[j]), op.apply(__uint_as_float(rg4[j]), SteBwdOp::keep_of(p0, (mm >> (8 * j)) & 0xffu), dummy)));
                } else if constexpr (CM == CM_ELEM) {
                    ChanIter it;
                    it.C = geo.C;
                    it.inner = geo.inner;
                    it.seek((uint64_t)e);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const SteBwdOp::P p = op.channel_masked(param_per_channel ? it.c : 0u, it.c);
                        out[j] = __float_as_uint(gated(__uint_as_float(rx4[j]), op.apply(__uint_as_float(rg4[j]), p, dummy)));
                        it.next();
                    }
                } else {
                    SteBwdOp::P p = op.channel(0);
                    if constexpr (CM == CM_ROW) {
                        if (geo.groups_per_row >= 64u && geo.inner % 8u == 0u) {     // wave-uniform look-up (WaveRows), dense too
                            WaveRows wr;
                            wr.seek(__builtin_amdgcn_readfirstlane((uint32_t)(e_wave >> 3)), geo.groups_per_row, geo.C, op.cmask);
                            const bool first = (uint32_t)(half * 256 + lane * 4) < wr.split * 8u;
                            p = SteBwdOp::keep_of(op.channel(param_per_channel ? (first ? wr.c0 : wr.c1) : 0u), first ? wr.k0 : wr.k1);
                        } else {
                            const uint32_t c = (uint32_t)((uint64_t)e / geo.inner) % geo.C;   // inner % 4 == 0: 4 elements share a row
                            p = op.channel_masked(param_per_channel ? c : 0u, c);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        out[j] = __float_as_uint(gated(__uint_as_float(rx4[j]), op.apply(__uint_as_float(rg4[j]), p, dummy)));
                }
                st16<NT>((u32x4*)((float*)gx + e), out);
                if (rd.gx_img) {
                    const float r[4] = {__uint_as_float(out[0]), __uint_as_float(out[1]), __uint_as_float(out[2]), __uint_as_float(out[3])};
                    image_store<4>(rd.gx_img, rd.img_dt, e, r);
                }
            }
        }
    } else
    if (grp < geo.ngroups) {
        Raw8<GDT> rg = zero_raw8<GDT>();
        Raw8<XDT> rx = zero_raw8<XDT>();
        bool need = true;
        if constexpr (ELIDE && CM == CM_ROW) {
            if (geo.groups_per_row >= 64u) {     // wave-uniform look-up, see WaveRows
                const uint32_t lane = threadIdx.x & 63u;
                WaveRows wr;
                wr.seek(__builtin_amdgcn_readfirstlane((uint32_t)grp - lane), geo.groups_per_row, geo.C, op.cmask);
                need = (lane < wr.split ? wr.k0 : wr.k1) != 0u;
            } else {
                const uint32_t c = (uint32_t)((uint64_t)grp / geo.groups_per_row) % geo.C;
                need = op.cmask[c] != 0;
            }
        } else if constexpr (ELIDE && CM == CM_LAST) {
            const u32x2 m8 = *(const u32x2*)(op.cmask + last_dim_channel(grp * 8, geo.C));
            need = (m8[0] | m8[1]) != 0u;
        }
        uint32_t gbits = 0u;
        if (need) {
            if (G2DT < 0 || g) rg = load8_raw<GDT, NT>(g, grp);
            if constexpr (GATE) gbits = ((const uint8_t*)x)[grp];
            else rx = load8_raw<XDT, NT>(x, grp);
        }
        float vg[8], vx[8];
        unpack8<GDT>(rg, vg);
        if constexpr (G2DT >= 0) {
            float v2[8];
            unpack8<G2DT>(load8_raw<G2DT, NT>(g2, grp), v2);
#pragma unroll
            for (int j = 0; j < 8; ++j) vg[j] = g ? vg[j] + v2[j] : v2[j];
        }
        if constexpr (GATE) gate_to_floats<8>(gbits, vx);
        else unpack8<XDT>(rx, vx);
        int32_t dummy;
        if constexpr (CM == CM_SCALAR) {
            const SteBwdOp::P p = op.channel(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) vg[j] = gated(vx[j], op.apply(vg[j], p, dummy));
        } else if constexpr (CM == CM_ROW) {
            SteBwdOp::P p;
            if (geo.groups_per_row >= 64u) {         // wave-uniform look-up (WaveRows), dense too
                const uint32_t lane = threadIdx.x & 63u;
                WaveRows wr;
                wr.seek(__builtin_amdgcn_readfirstlane((uint32_t)grp - lane), geo.groups_per_row, geo.C, op.cmask);
                const bool first = lane < wr.split;
                p = SteBwdOp::keep_of(op.channel(param_per_channel ? (first ? wr.c0 : wr.c1) : 0u), first ? wr.k0 : wr.k1);
            } else {
                const uint32_t c = (uint32_t)((uint64_t)grp / geo.groups_per_row) % geo.C;
                p = op.channel_masked(param_per_channel ? c : 0u, c);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) vg[j] = gated(vx[j], op.apply(vg[j], p, dummy));
        } else if constexpr (CM == CM_LAST) {
            const SteBwdOp::P p0 = op.channel(0);
            const uint32_t c0 = last_dim_channel(grp * 8, geo.C);
            u32x2 mm = {0x01010101u, 0x01010101u};
            if (op.cmask) mm = *(const u32x2*)(op.cmask + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                vg[j] = gated(vx[j], op.apply(vg[j], SteBwdOp::keep_of(p0, (mm[j >> 2] >> (8 * (j & 3))) & 0xffu), dummy));
        } else {
            ChanIter it;
            it.C = geo.C;
            it.inner = geo.inner;
            it.seek((uint64_t)grp * 8);
            if (geo.inner >= 8) {       // at most two rows per lane: two look-ups instead of eight (see ew_kernel)
                const uint32_t left = geo.inner - it.r;
                const uint32_t c1 = it.c + 1 == geo.C ? 0u : it.c + 1;
                const SteBwdOp::P p0 = op.channel_masked(param_per_channel ? it.c : 0u, it.c);
                const SteBwdOp::P p1 = op.channel_masked(param_per_channel ? c1 : 0u, c1);
#pragma unroll
                for (int j = 0; j < 8; ++j) vg[j] = gated(vx[j], op.apply(vg[j], (uint32_t)j < left ? p0 : p1, dummy));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const SteBwdOp::P p = op.channel_masked(param_per_channel ? it.c : 0u, it.c);
                    vg[j] = gated(vx[j], op.apply(vg[j], p, dummy));
                    it.next();
                }
            }
        }
        store8<XDT, NT>(gx, grp, vg);
    }
    const int64_t e = geo.ngroups * 8 + threadIdx.x;
    if (blockIdx.x == 0 && e < geo.numel) {
        ChanIter it;
        it.C = geo.C;
        it.inner = geo.inner;
        it.seek((uint64_t)e);
        const SteBwdOp::P p = (CM == CM_SCALAR) ? op.channel(0) : op.channel_masked(param_per_channel ? it.c : 0u, it.c);
        int32_t dummy;
        float xe;
        if constexpr (GATE) xe = ((((const uint8_t*)x)[e >> 3] >> ((uint32_t)e & 7u)) & 1u) ? 1.0f : 0.0f;
        else xe = load1<XDT>(x, e);
        float ge = (G2DT < 0 || g) ? load1<GDT>(g, e) : 0.0f;
        if constexpr (G2DT >= 0) {
            const bool has3 = rd.g3 != nullptr;
            if (has3) {
                const float v3 = load1<G2DT>(rd.g3, e);
                ge = g ? ge + v3 : v3;
            }
            const float v2 = load1<G2DT>(g2, e);
            ge = (g || has3) ? ge + v2 : v2;
        }
        const float r = gated(xe, op.apply(ge, p, dummy));
        store1<XDT>(gx, e, r);
        if (rd.gx_img) {
            if (rd.img_dt == QS_BF16) store1<QS_BF16>(rd.gx_img, e, r);
            else store1<QS_F16>(rd.gx_img, e, r);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Mask apply with a full-shape (element-wise) mask, and the general broadcast case.
// ------------------------------------------------------------------------------------------------
template <int DT, bool NT>
__global__ __launch_bounds__(kBlock) void mask_full_kernel(const void* __restrict__ x, const uint8_t* __restrict__ m,
                                                            void* __restrict__ y, int64_t numel) {
    const int64_t ngroups = numel / 8;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const bool m_aligned = (((uintptr_t)m) & 7) == 0;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += stride) {
        Raw8<DT> raw = load8_raw<DT, NT>(x, g);
        uint8_t mb[8];
        if (m_aligned) {
            u32x2 mm = *((const u32x2*)m + g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mb[j] = (mm[0] >> (8 * j)) & 0xff;
                mb[4 + j] = (mm[1] >> (8 * j)) & 0xff;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) mb[j] = m[g * 8 + j];
        }
        float v[8];
        unpack8<DT>(raw, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * (mb[j] ? 1.0f : 0.0f);
        store8<DT, NT>(y, g, v);
    }
    const int64_t e = ngroups * 8 + threadIdx.x;
    if (blockIdx.x == 0 && e < numel) store1<DT>(y, e, load1<DT>(x, e) * (m[e] ? 1.0f : 0.0f));
}

struct BcastGeom {
    int ndim;
    int64_t sizes[QS_MAX_DIMS];
    int64_t mstrides[QS_MAX_DIMS];
};

// General broadcast pattern, 8 elements per lane: needs the innermost collapsed extent to be a multiple of 8, so that
// a lane's 16-byte group stays inside one innermost row; its mask bytes are then either 8 consecutive bytes (mask
// dense along that dim) or one byte (mask broadcast along it).  The scalar kernel below ran at 13 % of the HBM peak.
template <int DT, bool NT>
__global__ __launch_bounds__(kBlock) void mask_bcast_vec_kernel(const void* __restrict__ x, const uint8_t* __restrict__ m,
                                                                 void* __restrict__ y, int64_t ngroups, BcastGeom geo) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= ngroups) return;
    const Raw8<DT> raw = load8_raw<DT, NT>(x, g);
    int64_t rem = g * 8, moff = 0;
    const int last = geo.ndim - 1;
    {
        const int64_t q = rem / geo.sizes[last];
        moff = (rem - q * geo.sizes[last]) * geo.mstrides[last];
        rem = q;
    }
    for (int d = last - 1; d >= 0; --d) {
        const int64_t q = rem / geo.sizes[d];
        moff += (rem - q * geo.sizes[d]) * geo.mstrides[d];
        rem = q;
    }
    float v[8];
    unpack8<DT>(raw, v);
    if (geo.mstrides[last] == 0) {
        const float keep = m[moff] ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * keep;
    } else {      // stride 1 (a dense innermost run of the mask)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * (m[moff + j] ? 1.0f : 0.0f);
    }
    store8<DT, NT>(y, g, v);
}

template <int DT>
__global__ __launch_bounds__(kBlock) void mask_bcast_kernel(const void* __restrict__ x, const uint8_t* __restrict__ m,
                                                             void* __restrict__ y, int64_t numel, BcastGeom geo) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < numel; e += stride) {
        int64_t rem = e, moff = 0;
        for (int d = geo.ndim - 1; d >= 0; --d) {
            const int64_t q = rem / geo.sizes[d];
            moff += (rem - q * geo.sizes[d]) * geo.mstrides[d];
            rem = q;
        }
        store1<DT>(y, e, load1<DT>(x, e) * (m[moff] ? 1.0f : 0.0f));
    }
}

}  // namespace qs
