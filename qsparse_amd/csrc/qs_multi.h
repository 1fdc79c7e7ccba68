// Multi-tensor weight path: abs-max, running-scale update and quantization of MANY small fp32 tensors (the conv /
// linear weights -- and biases -- of a converted network) in three launches instead of three per tensor.  Per tensor these
// are 3-5 us kernels over a few thousand to a few million elements; a ResNet-50 has 54 of them, i.e. ~160 launches per step.
//
// The tensor list is a DEVICE-resident table of `qs_multi_row` descriptors (include/qsparse_hip.h) that the caller builds
// once per set of layers and reuses every step -- any number of tensors per launch, nothing re-marshalled per step.  A
// workgroup finds its tensor by binary search over the table's prefix sums of workgroups.  Tensor-wise AND per-channel
// quantizers (the reference's default for weights is channelwise=1, quantize.py:524): a tensor is the contiguous
// [outer, C, inner] view around its channel dim, C == 1 for tensor-wise.  Arithmetic is that of reduce_all_kernel /
// reduce_rows_kernel, scale_update_kernel, decimal_from_scale_kernel and ew_kernel<ScalerFwdOp | DecimalFwdOp | SteBwdOp>.
#pragma once
#include "qs_elementwise.h"
#include "qs_reduce.h"

namespace qs {

__device__ __forceinline__ int multi_find(const qs_multi_row* __restrict__ rows, int n, int b, int which) {
    int lo = 0, hi = n - 1;     // largest i with block0[i] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const int b0 = which ? rows[mid].quant_block0 : rows[mid].absmax_block0;
        if (b0 <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// the prune operator's mask at element e of a row (1.0f / 0.0f: the quantizer's input is the PRODUCT x * mask, sparse.py:263)
__device__ __forceinline__ float multi_mask_at(const uint8_t* __restrict__ mask, int32_t mask_C, int64_t mask_inner, int64_t e) {
    const int64_t i = mask_C == 0 ? e : (e / mask_inner) % mask_C;
    return mask[i] ? 1.0f : 0.0f;
}

constexpr int kMultiTallCols = 64;      // per-channel abs-max, outer > 1: adjacent columns per workgroup (one coalesced 256-byte row segment)
constexpr int kMultiFlatCols = 2048;    // outer == 1 (the channel dim is the first one): elements per workgroup

// abs-max of every tensor that updates its statistics this step (row.train != 0), max-accumulated into row.amax[c]
// (zero between steps: multi_scale_update_kernel re-zeroes).
//   tensor-wise (C == 1):  the tensor's workgroups stride over its 8-element groups, one atomic per workgroup.
//   per channel:           the tensor is the matrix [outer, cols = C * inner].  outer > 1: a workgroup owns 64 adjacent columns
//                          (and one of `row_splits` interleaved row sets), its four waves take every fourth row, lanes walk
//                          their column (256 contiguous bytes per wave and row); outer == 1: 2048 consecutive elements.  The
//                          columns' maxima are folded per channel in LDS and flushed with ONE atomic per channel touched --
//                          same-line atomics serialise on this chip, so there must be few of them.
static __global__ __launch_bounds__(kBlock) void multi_absmax_kernel(const qs_multi_row* __restrict__ rows, int n) {
    const int i = multi_find(rows, n, blockIdx.x, 0);
    const qs_multi_row r = rows[i];
    const int b = blockIdx.x - r.absmax_block0;
    const float* x = r.x;
    uint32_t* amax = (uint32_t*)r.amax;
    if (r.C == 1) {
        const int nb = r.absmax_blocks;
        const int64_t ngroups = r.numel / 8;
        RedAcc<QS_F32, false> acc;
        if ((((uintptr_t)x) & 15) == 0) {
            for (int64_t g = (int64_t)b * kBlock + threadIdx.x; g < ngroups; g += (int64_t)nb * kBlock) {
                float v[8];
                unpack8<QS_F32>(load8_raw<QS_F32, false>(x, g), v);
                if (r.mask) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= multi_mask_at(r.mask, r.mask_C, r.mask_inner, g * 8 + j);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc.add(v[j]);
            }
            if (b == 0 && ngroups * 8 + threadIdx.x < r.numel) {
                const int64_t e = ngroups * 8 + threadIdx.x;
                acc.add(r.mask ? x[e] * multi_mask_at(r.mask, r.mask_C, r.mask_inner, e) : x[e]);
            }
        } else {
            for (int64_t e = (int64_t)b * kBlock + threadIdx.x; e < r.numel; e += (int64_t)nb * kBlock)
                acc.add(r.mask ? x[e] * multi_mask_at(r.mask, r.mask_C, r.mask_inner, e) : x[e]);
        }
        __shared__ uint32_t smx[kBlock / 64];
        acc.wave_reduce();
        if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = acc.mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = smx[0];
            for (int w = 1; w < kBlock / 64; ++w) m = smx[w] > m ? smx[w] : m;
            atomicMax(amax, m);
        }
        return;
    }
    __shared__ uint32_t chan[kMultiFlatCols + 2];
    const int64_t cols = (int64_t)r.C * r.inner;
    const int cpb = r.outer > 1 ? kMultiTallCols : kMultiFlatCols;
    const int splits = r.row_splits;
    const int cb = b / splits, rs = b - cb * splits;
    const int64_t col0 = (int64_t)cb * cpb;
    const int64_t c_lo = col0 / r.inner;
    for (int t = threadIdx.x; t < cpb + 2; t += kBlock) chan[t] = 0u;
    __syncthreads();
    if (r.outer > 1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int64_t j = col0 + lane;
        if (j < cols) {
            uint32_t m = 0u;
            const float* col = x + j;
            const int64_t stride = (int64_t)splits * (kBlock / 64);              // rows between two visits of this wave
            int64_t row = (int64_t)rs * (kBlock / 64) + wave;
            if (r.mask) {       // a pruned weight: |x * mask|
                for (; row < r.outer; row += stride) {
                    const float v = col[row * cols] * multi_mask_at(r.mask, r.mask_C, r.mask_inner, row * cols + j);
                    const uint32_t k = __float_as_uint(v) & 0x7fffffffu;
                    m = k > m ? k : m;
                }
            }
            for (; row + 3 * stride < r.outer; row += 4 * stride) {             // four rows in flight
                const uint32_t k0 = __float_as_uint(col[row * cols]) & 0x7fffffffu;
                const uint32_t k1 = __float_as_uint(col[(row + stride) * cols]) & 0x7fffffffu;
                const uint32_t k2 = __float_as_uint(col[(row + 2 * stride) * cols]) & 0x7fffffffu;
                const uint32_t k3 = __float_as_uint(col[(row + 3 * stride) * cols]) & 0x7fffffffu;
                const uint32_t a = k0 > k1 ? k0 : k1, c = k2 > k3 ? k2 : k3;
                const uint32_t k = a > c ? a : c;
                m = k > m ? k : m;
            }
            for (; row < r.outer; row += stride) {
                const uint32_t k = __float_as_uint(col[row * cols]) & 0x7fffffffu;
                m = k > m ? k : m;
            }
            atomicMax(&chan[(int)(j / r.inner - c_lo)], m);
        }
    } else {
        for (int t = threadIdx.x; t < cpb; t += kBlock) {
            const int64_t j = col0 + t;
            if (j < cols) {
                const float v = r.mask ? x[j] * multi_mask_at(r.mask, r.mask_C, r.mask_inner, j) : x[j];
                atomicMax(&chan[(int)(j / r.inner - c_lo)], __float_as_uint(v) & 0x7fffffffu);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < cpb + 2; t += kBlock) {
        const int64_t c = c_lo + t;
        if (c < r.C && chan[t] != 0u) atomicMax(amax + c, chan[t]);
    }
}

// one thread per (tensor, channel) of the table: the running mean of scale_update_kernel, the decimal of
// decimal_from_scale_kernel, the backup a caller needs to undo the update.  t comes from row.t_dev (device-resident) plus
// row.t_offset; the counters themselves are advanced by the launch that FOLLOWS in stream order (multi_quant_kernel with
// advance != 0): every thread of this one has read them by then, whichever workgroup it ran in.
static __global__ __launch_bounds__(kBlock) void multi_scale_update_kernel(const qs_multi_row* __restrict__ rows, int n,
                                                                          int total_channels) {
    const int item = blockIdx.x * kBlock + threadIdx.x;
    if (item >= total_channels) return;
    int lo = 0, hi = n - 1;     // the tensor this channel belongs to: largest i with chan0[i] <= item
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rows[mid].chan0 <= item) lo = mid;
        else hi = mid - 1;
    }
    const qs_multi_row& r = rows[lo];
    if (!r.train) return;
    const int c = item - r.chan0;
    uint32_t* amax = (uint32_t*)r.amax;
    float* scale = r.scale;
    const float t = (float)(*r.t_dev + r.t_offset);
    const float nw = __uint_as_float(amax[c]) / r.denom;                   // fp32 weights: no dtype rounding (quantize.py:340)
    const float old = scale[c];
    if (r.backup) r.backup[c] = old;
    const float s = (t == 0.0f) ? nw : (t * old + nw) / (t + 1.0f);       // quantize.py:344-347
    scale[c] = s;
    amax[c] = 0u;
    if (r.decimal) {
        float q = 1.0f / s;
        if (q == __builtin_inff() || q == -__builtin_inff()) q = 1.0f;      // nan_to_num(posinf=1, neginf=1) (:316)
        if (q != q) q = 0.0f;
        r.decimal[c] = rintf(log2f(q));
    }
}

// channel of element e of row r
__device__ __forceinline__ uint32_t multi_channel(const qs_multi_row& r, int64_t e) {
    return (uint32_t)((e / r.inner) % r.C);
}

// y = Q(x) for every tensor of the table: exact grid, 8 elements per lane; the per-tensor parameter is the scale (scaler
// quantizer) or the decimal (row.is_decimal: the truncating power-of-two quantizer), one per channel
static __global__ __launch_bounds__(kBlock) void multi_quant_kernel(const qs_multi_row* __restrict__ rows, int n, float* ybase,
                                                                   int advance) {
    if (advance && blockIdx.x == 0) {
        // the step counters of the rows that updated their statistics (see multi_scale_update_kernel).  A callback shared by
        // a layer's weight and bias quantizers (reference quantize.py:548,559-571) appears in two rows with the same counter
        // and t_offset 0 / 1: the weight's update saw t, the bias's t + 1, the counter moves by two.
        for (int k = threadIdx.x; k < n; k += kBlock) {
            // (the prune operator's counters of a pruned weight: every training read counts, sparse.py:117,272)
            // (a mask-level row names the same count only to READ it in multi_magnitude_kernel)
            if (rows[k].prune_n_updates && rows[k].kind != 1) atomicAdd(rows[k].prune_n_updates, 1);
            if (rows[k].prune_t && rows[k].kind != 1) atomicAdd((unsigned long long*)rows[k].prune_t, 1ull);
            // (kind 2: the quantizer underneath is in its identity phase -- or absent: it only counts the read, quantize.py:515)
            if (rows[k].kind == 2 && rows[k].bump) atomicAdd(rows[k].bump, 1);
            if (!rows[k].train) continue;
            atomicAdd((unsigned long long*)rows[k].t_dev, 1ull);
            if (rows[k].bump) atomicAdd(rows[k].bump, 1);
        }
    }
    const int i = multi_find(rows, n, blockIdx.x, 1);
    const qs_multi_row r = rows[i];
    if (r.kind == 1) return;                // a mask-level row: nothing to quantize
    const int64_t g = (int64_t)(blockIdx.x - r.quant_block0) * kBlock + threadIdx.x;
    const float* x = r.x;
    float* y = ybase + r.y_off;
    if (r.kind == 2) {                      // no quantization (a prune-only weight, or a quantizer in its identity phase): y = x * mask
        for (int64_t e = g * 8; e < r.numel && e < g * 8 + 8; ++e)
            y[e] = r.mask ? x[e] * multi_mask_at(r.mask, r.mask_C, r.mask_inner, e) : x[e];
        return;
    }
    const int64_t numel = r.numel, ngroups = numel / 8;
    const int sat = r.code_lo <= r.code_hi;
    const float* param = r.is_decimal ? r.decimal : r.scale;
    const bool vec = ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    int32_t code;
    // the quantizer's input at element e: x[e], or x[e] * mask for a pruned weight
    auto in1 = [&](int64_t e) -> float { return r.mask ? x[e] * multi_mask_at(r.mask, r.mask_C, r.mask_inner, e) : x[e]; };
    auto in8 = [&](int64_t grp, float (&v)[8]) {
        unpack8<QS_F32>(load8_raw<QS_F32, false>(x, grp), v);
        if (r.mask) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= multi_mask_at(r.mask, r.mask_C, r.mask_inner, grp * 8 + j);
        }
    };
    auto run = [&](auto op) {
        if (r.C == 1) {
            const auto p = op.channel(0);
            if (vec) {
                if (g < ngroups) {
                    float v[8];
                    in8(g, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], p, code);
                    store8<QS_F32, false>(y, g, v);
                }
                if (g == 0)
                    for (int64_t e = ngroups * 8; e < numel; ++e) y[e] = op.apply(in1(e), p, code);
            } else {
                for (int64_t e = g * 8; e < numel && e < g * 8 + 8; ++e) y[e] = op.apply(in1(e), p, code);
            }
            return;
        }
        // per channel: the parameters are looked up again whenever the channel changes along the lane's 8 elements
        const int64_t e0 = g * 8;
        if (e0 >= numel) return;
        uint32_t c = multi_channel(r, e0), rem = (uint32_t)(r.inner - e0 % r.inner);
        auto p = op.channel(c);
        if (vec && e0 + 8 <= numel) {
            float v[8];
            in8(g, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = op.apply(v[j], p, code);
                if (--rem == 0) {
                    rem = (uint32_t)r.inner;
                    c = (c + 1 == (uint32_t)r.C) ? 0u : c + 1;
                    p = op.channel(c);
                }
            }
            store8<QS_F32, false>(y, g, v);
        } else {
            for (int64_t e = e0; e < numel && e < e0 + 8; ++e) {
                y[e] = op.apply(in1(e), p, code);
                if (--rem == 0) {
                    rem = (uint32_t)r.inner;
                    c = (c + 1 == (uint32_t)r.C) ? 0u : c + 1;
                    p = op.channel(c);
                }
            }
        }
    };
    if (r.is_decimal) run(DecimalFwdOp<QS_F32>{param, 0.0f, nullptr, sat, r.code_lo, r.code_hi, ActSpec{0, 0.f, 0.f}, QS_F32});
    else run(ScalerFwdOp<QS_F32>{param, 0.0f, nullptr, sat, r.code_lo, r.code_hi, ActSpec{0, 0.f, 0.f}, QS_F32});
}

// the running magnitude of the pruned weights with a full-shape mask (MagnitudePruningCallback.update_magnitude, reference
// sparse.py:82-89: magnitude <- (t * magnitude + |x|) / (t + 1), every operation rounded to fp32), rows with magnitude != NULL;
// the replaced values go to mag_backup (a caller that evaluated the layer ahead of time restores them if the forward never
// reads the weight).  Same workgroup partition as multi_quant_kernel; t is read from prune_t, which that kernel advances.
static __global__ __launch_bounds__(kBlock) void multi_magnitude_kernel(const qs_multi_row* __restrict__ rows, int n) {
    const int i = multi_find(rows, n, blockIdx.x, 1);
    const qs_multi_row r = rows[i];
    if (!r.magnitude) return;
    const float t = (float)*r.prune_t, tp1 = (float)(*r.prune_t + 1);
    const int64_t e0 = ((int64_t)(blockIdx.x - r.quant_block0) * kBlock + threadIdx.x) * 8;
    for (int64_t e = e0; e < r.numel && e < e0 + 8; ++e) {
        const float old = r.magnitude[e];
        r.mag_backup[e] = old;
        r.magnitude[e] = (t * old + fabsf(r.x[e])) / tp1;
    }
}

// ---- one stage of the staged mean (importance of pruned weights whose mask varies along a subset of dims) for a list of tensors --
// one lane per output element, ATen's CPU summation order: mean_generic_kernel (layout 0) / mean_cl_generic_kernel (layout 1)
static __global__ __launch_bounds__(kBlock) void multi_stage_mean_kernel(const qs_multi_stage* __restrict__ stages, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (stages[mid].block0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const qs_multi_stage st = stages[lo];
    const int64_t t = (int64_t)(blockIdx.x - st.block0) * kBlock + threadIdx.x;
    if (t >= st.pre * st.post) return;
    const float* x = st.x;
    const float fn = (float)st.n;
    if (st.layout == 1) {
        // x: [n][hw][C] in memory, t = pos * C + c; out [C][hw]
        const int64_t hw = st.pre, C = st.post;
        const int64_t pos = t / C, c = t - pos * C;
        const int64_t sample = hw * C;
        auto get = [&](int64_t i) {
            const float v = x[i * sample + t];
            return st.take_abs ? fabsf(v) : v;
        };
        const float s = (pos < (hw / 4) * 4) ? sum_multi_row(st.n, get) : sum_row_sum(st.n, get);
        st.out[c * hw + pos] = s / fn;
        return;
    }
    const int64_t p = t / st.post, col = t - p * st.post;
    const int64_t base = p * st.n * st.post + col;
    auto get = [&](int64_t i) {
        const float v = x[base + i * st.post];
        return st.take_abs ? fabsf(v) : v;
    };
    float s;
    if (st.post == 1) {
        if (st.n >= 8) {   // vectorized inner sum: 8 lanes, row-sum order over n/8 vectors
            const int64_t nv = st.n / 8;
            float lanes[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) lanes[k] = sum_row_sum(nv, [&](int64_t i) { return get(8 * i + k); });
            float fin = 0.f;
            for (int64_t i = nv * 8; i < st.n; ++i) fin += get(i);
#pragma unroll
            for (int k = 0; k < 8; ++k) fin += lanes[k];
            s = fin;
        } else {
            s = sum_row_sum(st.n, get);
        }
    } else {
        const int64_t mr_cols = (st.post >= 8) ? (st.post / 32) * 32 : (st.post / 4) * 4;
        s = (col < mr_cols) ? sum_multi_row(st.n, get) : sum_row_sum(st.n, get);
    }
    st.out[p * st.post + col] = s / fn;
}

// ---- mask rebuild of the pruned weights with a full-shape mask (rows with refresh != 0) -----------------------------------------
// the radix select of qs_kth_value (select_hist_kernel / select_scan_kernel: 4 passes over order-preserving keys, most significant
// byte first) for every such row at once, then mask <- importance >= threshold.  A row's SelectState needs no initialisation
// launch: the first pass ignores what the previous step left (prefix and k are taken as 0 and row.select_k), the scan re-zeroes
// the histogram after every pass.
__device__ __forceinline__ int multi_find_hist(const qs_multi_row* __restrict__ rows, int n, int b) {
    int lo = 0, hi = n - 1;     // largest i with hist_block0[i] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rows[mid].hist_block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float multi_importance(const qs_multi_row& r, int64_t e) {
    return r.importance ? r.importance[e] : fabsf(r.x[e]);
}

static __global__ __launch_bounds__(kBlock) void multi_select_hist_kernel(const qs_multi_row* __restrict__ rows, int n, int pass) {
    const int i = multi_find_hist(rows, n, blockIdx.x);
    const qs_multi_row r = rows[i];
    if (!r.refresh || r.hist_blocks == 0) return;
    SelectState* st = (SelectState*)r.select_state;
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = (pass == 3) ? 0u : st->prefix;
    const int shift = 8 * pass;
    const uint32_t himask = (pass == 3) ? 0u : (0xffffffffu << (shift + 8));
    const int64_t stride = (int64_t)r.hist_blocks * kBlock;
    for (int64_t e = (int64_t)(blockIdx.x - r.hist_block0) * kBlock + threadIdx.x; e < r.numel; e += stride) {
        const uint32_t k = f32_to_key(multi_importance(r, e));
        if ((k & himask) == (prefix & himask)) atomicAdd(&h[(k >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

// one workgroup of 256 threads per row: pick the bin holding rank k, descend (select_scan_kernel)
static __global__ __launch_bounds__(256) void multi_select_scan_kernel(const qs_multi_row* __restrict__ rows, int pass) {
    const qs_multi_row& r = rows[blockIdx.x];
    if (!r.refresh) return;
    SelectState* st = (SelectState*)r.select_state;
    if (threadIdx.x == 0) {
        uint32_t k = (pass == 3) ? r.select_k : st->k, cum = 0;
        int b = 0;
        for (; b < 256; ++b) {
            const uint32_t c = st->hist[b];
            if (cum + c > k) break;
            cum += c;
        }
        if (b == 256) b = 255;
        st->k = k - cum;
        st->prefix = ((pass == 3) ? 0u : st->prefix) | (((uint32_t)b) << (8 * pass));
    }
    __syncthreads();
    st->hist[threadIdx.x] = 0;
}

// mask <- importance >= threshold (mask_ge_kernel), the replaced bytes to mask_backup; the workgroup partition of multi_quant_kernel
static __global__ __launch_bounds__(kBlock) void multi_mask_ge_kernel(const qs_multi_row* __restrict__ rows, int n) {
    const int i = multi_find(rows, n, blockIdx.x, 1);
    const qs_multi_row r = rows[i];
    if (!r.refresh) return;
    const float t = key_to_f32(((const SelectState*)r.select_state)->prefix);
    uint8_t* mask = (uint8_t*)r.mask;
    const int64_t e0 = ((int64_t)(blockIdx.x - r.quant_block0) * kBlock + threadIdx.x) * 8;
    for (int64_t e = e0; e < r.numel && e < e0 + 8; ++e) {
        r.mask_backup[e] = mask[e];
        mask[e] = multi_importance(r, e) >= t ? 1 : 0;
    }
}

// gx = clamp(g, lo_mul * s_c, hi_mul * s_c) for every tensor (SteBwdOp's arithmetic, reference quantize.py:66-77, 120-131): the
// STE backward of a GROUP of weight quantizers in one launch -- gradients of a few layers handed over together.  The
// gradients are fresh tensors every step, so this list travels by value in the kernel arguments.
constexpr int kMultiSteMax = 32;
struct MultiSte {
    const float* g[kMultiSteMax];
    float* gx[kMultiSteMax];
    const float* step[kMultiSteMax];   // scale(s), or decimal(s) with is_decimal: one per channel
    int64_t numel[kMultiSteMax];
    int32_t block0[kMultiSteMax + 1];
    int32_t C[kMultiSteMax], inner[kMultiSteMax];
    float lo_mul[kMultiSteMax];
    float hi_mul[kMultiSteMax];
    const uint8_t* mask[kMultiSteMax];   // nullable: gx = clamp(g) * mask (the pruned weight's backward)
    int64_t mask_inner[kMultiSteMax];
    int32_t mask_C[kMultiSteMax];
    int32_t n;
};

template <bool DECIMAL>
__global__ __launch_bounds__(kBlock) void multi_ste_kernel(MultiSte a) {
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.block0[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const int i = lo;
    const int64_t g0 = (int64_t)(blockIdx.x - a.block0[i]) * kBlock + threadIdx.x;
    const float* g = a.g[i];
    float* gx = a.gx[i];
    const int64_t numel = a.numel[i], ngroups = numel / 8;
    const SteBwdOp op{a.step[i], 0.0f, DECIMAL ? 1 : 0, a.lo_mul[i], a.hi_mul[i], 0, nullptr};
    int32_t code;
    const int C = a.C[i], inner = a.inner[i];
    const uint8_t* mask = a.mask[i];
    const int32_t mC = a.mask_C[i];
    const int64_t mI = a.mask_inner[i];
    if (C == 1) {
        const auto p = op.channel(0);
        if (g0 < ngroups) {
            float v[8];
            unpack8<QS_F32>(load8_raw<QS_F32, false>(g, g0), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = op.apply(v[j], p, code);
            if (mask) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= multi_mask_at(mask, mC, mI, g0 * 8 + j);
            }
            store8<QS_F32, false>(gx, g0, v);
        }
        if (g0 == 0)
            for (int64_t e = ngroups * 8; e < numel; ++e) {
                const float v = op.apply(g[e], p, code);
                gx[e] = mask ? v * multi_mask_at(mask, mC, mI, e) : v;
            }
        return;
    }
    const int64_t e0 = g0 * 8;
    if (e0 >= numel) return;
    uint32_t c = (uint32_t)((e0 / inner) % C), rem = (uint32_t)(inner - e0 % inner);
    auto p = op.channel(c);
    if (e0 + 8 <= numel) {
        float v[8];
        unpack8<QS_F32>(load8_raw<QS_F32, false>(g, g0), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = op.apply(v[j], p, code);
            if (--rem == 0) {
                rem = (uint32_t)inner;
                c = (c + 1 == (uint32_t)C) ? 0u : c + 1;
                p = op.channel(c);
            }
        }
        if (mask) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= multi_mask_at(mask, mC, mI, e0 + j);
        }
        store8<QS_F32, false>(gx, g0, v);
    } else {
        for (int64_t e = e0; e < numel; ++e) {
            const float ve = op.apply(g[e], p, code);
            gx[e] = mask ? ve * multi_mask_at(mask, mC, mI, e) : ve;
            if (--rem == 0) {
                rem = (uint32_t)inner;
                c = (c + 1 == (uint32_t)C) ? 0u : c + 1;
                p = op.channel(c);
            }
        }
    }
}

}  // namespace qs
