// Device-side helpers shared by the kernels of libqsparse_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qsparse_hip.h"

namespace qs {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;        // 4 waves of 64 lanes

// ---- scalar dtype conversions (round-to-nearest-even, as ATen's CPU casts) ---------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;  // c10::BFloat16 maps every NaN to 0x7FC0
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
    // The empty asm pins `f` as a materialised binary32 value.  Without it hipcc folds a preceding multiply
    // into v_fma_mixlo_f16 (a*b+0 rounded once to f16): that drops the sign of a zero product ((-0)+(+0) = +0)
    // and rounds once where the reference rounds twice (fp32 op, then the cast).
    asm volatile("" : "+v"(f));
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);
}

template <int DT>
__device__ __forceinline__ float round_through(float v) {  // value after one rounding to DT
    if constexpr (DT == QS_BF16) return bf16_bits_to_f32(f32_to_bf16_bits(v));
    if constexpr (DT == QS_F16) return f16_bits_to_f32(f32_to_f16_bits(v));
    return v;
}

// ATen's CPU relu is clamp_min(x, 0) = max_ps(0, x): NaN and -0.0 pass through.  fmaxf would turn NaN into 0 and hide
// a diverged activation that the module-by-module path reports.
__device__ __forceinline__ float relu_aten(float v) { return (v < 0.0f) ? 0.0f : v; }

// ---- folded activations ----------------------------------------------------------------------------
// convert() puts its operators behind whatever activation modules the user names (reference convert.py:214-218): nn.ReLU in
// the BASELINE networks, nn.ReLU6 / nn.Hardtanh / nn.LeakyReLU in others.  A site's kernels absorb the activation -- its output
// is never materialised -- when it is one of these: forward value, and the ONE bit per element its backward needs.
//   QS_ACT_RELU      max(x, 0) as ATen's clamp_min;            backward  x <= 0 ? 0 : g            (threshold_backward)
//   QS_ACT_HARDTANH  clamp(x, a, b) (nn.ReLU6: a = 0, b = 6);  backward  (x > a && x < b) ? g : 0   (hardtanh_backward)
//   QS_ACT_LEAKY     x > 0 ? x : x * a, the product rounded to x's dtype;  backward  x > 0 ? g : g * a (leaky_relu_backward)
struct ActSpec {
    int kind;       // 0: none, else QS_ACT_*
    float a, b;
};
__device__ __forceinline__ float round_to_dtype(float v, int dt) {   // value after one rounding to dtype `dt` (run-time)
    if (dt == QS_F16) return round_through<QS_F16>(v);
    if (dt == QS_BF16) return round_through<QS_BF16>(v);
    return v;
}
// the activation's forward value; `dt`: dtype of the activation's input / output (ATen computes in float and casts back)
__device__ __forceinline__ float act_apply(float v, const ActSpec& s, int dt) {
    if (s.kind == QS_ACT_RELU) return relu_aten(v);
    if (s.kind == QS_ACT_HARDTANH) {       // vec::clamp = minimum(b, maximum(a, x)): NaN passes, -0.0 survives a = +0.0
        // ATen's clamp converts its scalar bounds to the tensor's dtype first (clamp_scalar_kernel: min.to<scalar_t>()): a bound
        // that bf16 / fp16 cannot represent (0.1, 0.7) saturates at its ROUNDED value, which is what the statistics, the mask
        // product and the quotient then see.  (hardtanh_backward compares with the unrounded float bounds: act_open below.)
        const float a = round_to_dtype(s.a, dt), b = round_to_dtype(s.b, dt);
        const float t = (v < a) ? a : v;
        return (t > b) ? b : t;
    }
    if (s.kind == QS_ACT_LEAKY) return (v > 0.0f) ? v : round_to_dtype(v * s.a, dt);
    return v;
}
// the same with the kind known at compile time: straight-line code (the bounds' rounding is loop-invariant), for the statistics
// kernels whose batched loads a run-time switch on the kind would serialise
template <int KIND>
__device__ __forceinline__ float act_apply_k(float v, const ActSpec& s, int dt) {
    if constexpr (KIND == QS_ACT_RELU) return relu_aten(v);
    if constexpr (KIND == QS_ACT_HARDTANH) {
        const float a = round_to_dtype(s.a, dt), b = round_to_dtype(s.b, dt);
        const float t = (v < a) ? a : v;
        return (t > b) ? b : t;
    }
    if constexpr (KIND == QS_ACT_LEAKY) return (v > 0.0f) ? v : round_to_dtype(v * s.a, dt);
    return v;
}
// whether the activation's backward lets the gradient through unchanged at input v (the gate bit a forward records)
__device__ __forceinline__ bool act_open(float v, const ActSpec& s) {
    // (ATen's CPU hardtanh_backward treats a NaN input differently in its vector body -- (x > a) & (x < b): closed -- and in its
    // scalar tail -- (x <= a || x >= b) ? 0 : g: open; the body's rule is the one followed here)
    if (s.kind == QS_ACT_HARDTANH) return (v > s.a) && (v < s.b);
    if (s.kind == QS_ACT_LEAKY) return v > 0.0f;
    return !(v <= 0.0f);
}
// the gradient where the gate is closed: 0 for the rectifiers, g * slope (g in the gradient's dtype, the product rounded to
// it) for the leaky one
__device__ __forceinline__ float act_closed(float applied, const ActSpec& s, int dt) {
    return (s.kind == QS_ACT_LEAKY) ? round_to_dtype(applied, dt) * s.a : 0.0f;
}

// ---- an activation applied by the CALLER in front of a site, whose backward the site's backward kernel evaluates ----------
// convert(..., activation_layers=[nn.GELU]) puts the operators behind an activation the kernels do not fold (its forward value is
// ATen's own pass); its BACKWARD, gelu_backward(gh, x) with gh the site's input gradient in x's dtype, needs nothing but that
// gradient and x -- the site's backward kernel has the former in registers.  The factor below is ATen's GPU kernel
// (ActivationGeluKernel.cu, GeluBackwardCUDAKernelImpl, erf form: dy * (cdf + x * pdf), the sum contracted to an fma by its build;
// opmath float for every dtype), reproduced bit for bit: tools/probes/probe_gelu_bits.py compares every bf16 and fp16 input
// pattern and 2^24 float32 values (0 differing with the fma, 1.76 M of 16.8 M float32 without), tests/test_act_grad_gpu.py repeats it.
__device__ __forceinline__ float gelu_grad_factor(float x) {
    const float kBeta = (float)(1.12837916709551257390 * 0.70710678118654752440 * 0.5);      // M_2_SQRTPI * M_SQRT1_2 * 0.5
    const float kAlpha = (float)0.70710678118654752440;                                     // M_SQRT1_2
    const float cdf = 0.5f * (1.0f + erff(x * kAlpha));
    const float pdf = expf(-0.5f * x * x) * kBeta;
    return fmaf(x, pdf, cdf);
}

// ---- scalar element access ---------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float load1(const void* p, int64_t i) {
    if constexpr (DT == QS_F32) return ((const float*)p)[i];
    if constexpr (DT == QS_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
    return f16_bits_to_f32(((const uint16_t*)p)[i]);
}
template <int DT>
__device__ __forceinline__ void store1(void* p, int64_t i, float v) {
    if constexpr (DT == QS_F32) ((float*)p)[i] = v;
    else if constexpr (DT == QS_BF16) ((uint16_t*)p)[i] = (uint16_t)f32_to_bf16_bits(v);
    else ((uint16_t*)p)[i] = (uint16_t)f32_to_f16_bits(v);
}

// ---- 8-element vector access: 16 B/lane for 2-byte dtypes, 2 x 16 B/lane for fp32 ----------------
// `g` indexes groups of 8 elements.  NT selects non-temporal (streaming) accesses.
template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    return *p;
}
// Streaming stores are non-temporal.  Measured alternatives on MI355X (tools/probe_order.py): plain stores
// are equal within noise; sc1 / sc0 sc1 (write-through) stores make the bf16->fp32 apply kernel 2.4x slower
// and do not relieve the following kernel of the Infinity-Cache write-back.
template <bool NT>
__device__ __forceinline__ void st16(u32x4* p, u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int DT>
struct Raw8 {  // the raw registers of one 8-element group
    u32x4 a;
    u32x4 b;  // only used for fp32
};

template <int DT, bool NT>
__device__ __forceinline__ Raw8<DT> load8_raw(const void* base, int64_t g) {
    Raw8<DT> r;
    if constexpr (DT == QS_F32) {
        const u32x4* p = (const u32x4*)base + 2 * g;
        r.a = ld16<NT>(p);
        r.b = ld16<NT>(p + 1);
    } else {
        r.a = ld16<NT>((const u32x4*)base + g);
    }
    return r;
}

template <int DT>
__device__ __forceinline__ void unpack8(const Raw8<DT>& r, float (&v)[8]) {
    if constexpr (DT == QS_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = __uint_as_float(r.a[j]);
            v[4 + j] = __uint_as_float(r.b[j]);
        }
    } else if constexpr (DT == QS_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = __uint_as_float(r.a[j] << 16);
            v[2 * j + 1] = __uint_as_float(r.a[j] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = f16_bits_to_f32(r.a[j] & 0xffffu);
            v[2 * j + 1] = f16_bits_to_f32(r.a[j] >> 16);
        }
    }
}

template <int DT, bool NT>
__device__ __forceinline__ void store8(void* base, int64_t g, const float (&v)[8]) {
    if constexpr (DT == QS_F32) {
        u32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = __float_as_uint(v[j]);
            b[j] = __float_as_uint(v[4 + j]);
        }
        u32x4* p = (u32x4*)base + 2 * g;
        st16<NT>(p, a);
        st16<NT>(p + 1, b);
    } else {
        u32x4 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t lo, hi;
            if constexpr (DT == QS_BF16) {
                lo = f32_to_bf16_bits(v[2 * j]);
                hi = f32_to_bf16_bits(v[2 * j + 1]);
            } else {
                lo = f32_to_f16_bits(v[2 * j]);
                hi = f32_to_f16_bits(v[2 * j + 1]);
            }
            a[j] = lo | (hi << 16);
        }
        st16<NT>((u32x4*)base + g, a);
    }
}

// eight copies of one value (the eliding kernels' pruned lanes): ONE conversion instead of eight
template <int DT, bool NT>
__device__ __forceinline__ void store8_splat(void* base, int64_t g, float z) {
    if constexpr (DT == QS_F32) {
        const uint32_t w = __float_as_uint(z);
        const u32x4 a = {w, w, w, w};
        u32x4* p = (u32x4*)base + 2 * g;
        st16<NT>(p, a);
        st16<NT>(p + 1, a);
    } else {
        const uint32_t h = (DT == QS_BF16) ? f32_to_bf16_bits(z) : f32_to_f16_bits(z);
        const uint32_t w = h | (h << 16);
        st16<NT>((u32x4*)base + g, u32x4{w, w, w, w});
    }
}

__device__ __forceinline__ void store8_i32(int32_t* base, int64_t g, const int32_t (&q)[8]) {
    u32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = (uint32_t)q[j];
        b[j] = (uint32_t)q[4 + j];
    }
    u32x4* p = (u32x4*)base + 2 * g;
    *p = a;
    *(p + 1) = b;
}

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------------------
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// order-preserving map float -> uint32 (NaN of either sign maps to the top, like torch.sort)
__device__ __forceinline__ uint32_t f32_to_key(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_f32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

// channel of a flat element index for an [outer, C, inner] view
struct ChanIter {
    uint32_t c, r, C, inner;
    __device__ __forceinline__ void seek(uint64_t e) {
        uint64_t row = e / inner;
        r = (uint32_t)(e - row * inner);
        c = (uint32_t)(row % C);
    }
    __device__ __forceinline__ void next() {
        if (++r == inner) {
            r = 0;
            if (++c == C) c = 0;
        }
    }
};

}  // namespace qs
