// Reductions: abs-max / min-max statistics, staged means in ATen's CPU summation order, k-th order
// statistic (radix select) and the C-sized "select" step of the fused prune->quantize pair.
//
// SUMMATION-ORDER PIN: the staged-mean kernels restate SumKernel.cpp's order (cascade / 4-way row-sum / vectorised inner sum, one
// intra-op thread) as torch 2.10 computes it -- the version recorded in the golden fixtures' meta (tests/golden/*.npz).  A torch
// upgrade that changes that order moves the target of every `mean_*` kernel in this file: tests/test_aten_contract.py (CPU) and
// tests/test_aten_contract_gpu.py (`-m gpu`) detect it, qsparse_amd.util.check_torch_pin warns at import.
#pragma once
#include "qs_common.h"

namespace qs {

// =================================================================================================
// abs-max / min-max.  Results are accumulated with integer atomics on order-preserving keys, so they
// are independent of the order of arrival (bit-exact on any device).
//   kind 0: key = bits of |x|  (non-negative floats order like their bit patterns; NaN sorts on top)
//   kind 1: min and max of x through f32_to_key
// =================================================================================================
template <int DT, bool MINMAX>
struct RedAcc {
    uint32_t mx = 0u, mn = 0xffffffffu;
    // min/max run on the float pipeline with gfx950's IEEE-754-2019 maximum / minimum (v_maximum3_f32 / v_minimum3_f32:
    // -0 < +0, and a NaN operand PROPAGATES, as torch.min / torch.max do): one instruction per two elements and statistic.
    // (Round 2 used v_max_f32 / v_min_f32, which drop NaN operands, plus a NaN flag on the side -- 3 VALU ops per element,
    // and a minimum that ignored NaNs where the reference's rows.min() returns NaN; before that, key conversion + integer
    // min/max at ~9 ops made these kernels VALU-bound at 4.3 TB/s.)
    float fmx = -__builtin_inff(), fmn = __builtin_inff();
    ActSpec act = {0, 0.f, 0.f};   // abs-max of act(x): the statistics of a folded activation (nn.ReLU: max(x, 0))
    __device__ __forceinline__ void add(float v) {
        if (act.kind) v = act_apply(v, act, DT);
        if constexpr (MINMAX) {
            fmx = __builtin_elementwise_maximum(fmx, v);
            fmn = __builtin_elementwise_minimum(fmn, v);
        } else {
            uint32_t k = __float_as_uint(v) & 0x7fffffffu;
            mx = k > mx ? k : mx;
        }
    }
    __device__ __forceinline__ void fold() {      // float state -> order-preserving keys; a NaN wins both: key 0xffffffff on
        if constexpr (MINMAX) {                   // top of the maxima, key 0 below every minimum (both decode to NaN)
            const uint32_t kx = f32_to_key(fmx), kn = (fmn != fmn) ? 0u : f32_to_key(fmn);
            mx = kx > mx ? kx : mx;
            mn = kn < mn ? kn : mn;
            fmx = -__builtin_inff();
            fmn = __builtin_inff();
        }
    }
    __device__ __forceinline__ void wave_reduce() {
        fold();
        mx = wave_max_u32(mx);
        if constexpr (MINMAX) mn = wave_min_u32(mn);
    }
    __device__ __forceinline__ void flush(uint32_t* out_max, uint32_t* out_min, uint32_t idx) {
        fold();
        atomicMax(out_max + idx, mx);
        if constexpr (MINMAX) atomicMin(out_min + idx, mn);
    }
};

// whole tensor -> out[0].  Every workgroup ends with one atomic on the same word (~12 ns each, serialised): BS = 1024
// halves their number at the same number of waves per CU.
template <int DT, bool MINMAX, int BS>
__global__ __launch_bounds__(BS) void reduce_all_kernel(const void* __restrict__ x, int64_t numel,
                                                             uint32_t* out_max, uint32_t* out_min, ActSpec relu, int lines) {
    RedAcc<DT, MINMAX> acc;
    acc.act = relu;
    const int64_t ngroups = numel / 8;
    const int64_t stride = (int64_t)gridDim.x * BS;
    int64_t g = (int64_t)blockIdx.x * BS + threadIdx.x;
    for (; g + 3 * stride < ngroups; g += 4 * stride) {   // four 16-byte loads in flight per lane
        Raw8<DT> r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = load8_raw<DT, false>(x, g + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[8];
            unpack8<DT>(r[u], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc.add(v[j]);
        }
    }
    for (; g < ngroups; g += stride) {
        Raw8<DT> r0 = load8_raw<DT, false>(x, g);
        float v[8];
        unpack8<DT>(r0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.add(v[j]);
    }
    const int64_t e = ngroups * 8 + threadIdx.x;
    if (blockIdx.x == 0 && e < numel) acc.add(load1<DT>(x, e));

    __shared__ uint32_t smx[BS / 64], smn[BS / 64];
    acc.wave_reduce();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        smx[w] = acc.mx;
        smn[w] = acc.mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < BS / 64; ++i) {
            acc.mx = smx[i] > acc.mx ? smx[i] : acc.mx;
            acc.mn = smn[i] < acc.mn ? smn[i] : acc.mn;
        }
        // `lines` accumulators, one 128-byte line each (QS_AMAX_LINE_STRIDE floats apart): same-address atomics serialise
        // at ~12 ns each, so with one word the last of 256 workgroups waits ~3 us; the consumer (scale_update_kernel) folds
        acc.flush(out_max, out_min, (uint32_t)(blockIdx.x % (uint32_t)lines) * (uint32_t)QS_AMAX_LINE_STRIDE);
    }
}

// per channel, rows of `inner` contiguous elements.  Workgroup (c, s) owns channel c and the s-th slice of the
// outer range; its waves walk rows o*C + c and keep their running extrema in registers, so each workgroup
// ends with ONE atomic: same-address device-scope atomics cost ~0.3 us each once hundreds of workgroups
// contend (65 k per-row atomics on 256 words took 80 us), a few per word are free.
template <int DT, bool MINMAX>
__global__ __launch_bounds__(kBlock) void reduce_rows_kernel(const void* __restrict__ x, int64_t outer, uint32_t C,
                                                              int64_t inner, int vec_ok, int64_t outer_per_block,
                                                              uint32_t* out_max, uint32_t* out_min, ActSpec relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x;
    const int64_t o0 = (int64_t)blockIdx.y * outer_per_block;
    const int64_t o1 = o0 + outer_per_block < outer ? o0 + outer_per_block : outer;
    RedAcc<DT, MINMAX> acc;
    acc.act = relu;
    // short rows (<= 512 elements, e.g. 14x14 and 7x7 maps): a row is one vector load per lane, so a wave keeps FOUR rows
    // in flight instead of one (64x1024x14x14 bf16: 26 -> 23 us, tools/bench_reduce.py)
    const bool short_rows = vec_ok != 0 && inner <= 512;
    for (int64_t o = o0 + wave; short_rows && o < o1; o += 4 * (kBlock / 64)) {
        Raw8<DT> r[4];
        float head[4], tail[4];
        bool has_vec[4], has_head[4], has_tail[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t oo = o + u * (kBlock / 64);
            has_vec[u] = has_head[u] = has_tail[u] = false;
            if (oo < o1) {
                const int64_t base = (oo * C + c) * inner, e1 = base + inner;
                const int64_t ga = (base + 7) / 8, gb = e1 / 8;       // the 8-element groups that lie inside the row
                if (ga < gb) {
                    if (ga + lane < gb) {
                        r[u] = load8_raw<DT, false>(x, ga + lane);
                        has_vec[u] = true;
                    }
                    if (base + lane < ga * 8) {
                        head[u] = load1<DT>(x, base + lane);
                        has_head[u] = true;
                    }
                    if (gb * 8 + lane < e1) {
                        tail[u] = load1<DT>(x, gb * 8 + lane);
                        has_tail[u] = true;
                    }
                } else {
                    for (int64_t i = lane; i < inner; i += 64) acc.add(load1<DT>(x, base + i));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (has_vec[u]) {
                float v[8];
                unpack8<DT>(r[u], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc.add(v[j]);
            }
            if (has_head[u]) acc.add(head[u]);
            if (has_tail[u]) acc.add(tail[u]);
        }
    }
    // long aligned rows: TWO rows of the wave per iteration, up to eight 16-byte loads in flight per lane and row -- with
    // one row (6.1 loads per lane on 56x56 maps) a wave idled between its rows (256x256x56x56 bf16: 5.4 TB/s)
    const bool row_pairs = !short_rows && vec_ok == 1;
    for (int64_t o = o0 + wave; row_pairs && o < o1; o += 2 * (kBlock / 64)) {
        const int64_t ng = inner / 8;
        const int64_t ga = ((o * C + c) * inner) / 8;
        const bool two = o + kBlock / 64 < o1;
        const int64_t gb = (((o + kBlock / 64) * C + c) * inner) / 8;
        for (int64_t g = lane; g < ng; g += 8 * 64) {
            Raw8<DT> ra[8], rb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (g + u * 64 < ng) ra[u] = load8_raw<DT, false>(x, ga + g + u * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (two && g + u * 64 < ng) rb[u] = load8_raw<DT, false>(x, gb + g + u * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (g + u * 64 < ng) {
                    float v[8];
                    unpack8<DT>(ra[u], v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc.add(v[j]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (two && g + u * 64 < ng) {
                    float v[8];
                    unpack8<DT>(rb[u], v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc.add(v[j]);
                }
            }
        }
    }
    for (int64_t o = o0 + wave; !short_rows && !row_pairs && o < o1; o += kBlock / 64) {
        const int64_t base = (o * C + c) * inner;
        if (vec_ok == 2) {
            // aligned tensor, ragged rows (14x14, 7x7 maps): the 8-element groups that lie inside the row with vector
            // loads, the few elements in front of the first and behind the last one with scalar loads
            const int64_t e1 = base + inner;
            const int64_t ga = (base + 7) / 8, gb = e1 / 8;
            if (ga < gb) {
                for (int64_t g = ga + lane; g < gb; g += 4 * 64) {
                    Raw8<DT> r[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (g + u * 64 < gb) r[u] = load8_raw<DT, false>(x, g + u * 64);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (g + u * 64 < gb) {
                            float v[8];
                            unpack8<DT>(r[u], v);
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc.add(v[j]);
                        }
                    }
                }
                if (base + lane < ga * 8) acc.add(load1<DT>(x, base + lane));          // head: < 8 elements
                if (gb * 8 + lane < e1) acc.add(load1<DT>(x, gb * 8 + lane));          // tail: < 8 elements
            } else {
                for (int64_t i = lane; i < inner; i += 64) acc.add(load1<DT>(x, base + i));
            }
        } else {
            for (int64_t i = lane; i < inner; i += 64) acc.add(load1<DT>(x, base + i));
        }
    }
    __shared__ uint32_t smx[kBlock / 64], smn[kBlock / 64];
    acc.wave_reduce();
    if (lane == 0) {
        smx[wave] = acc.mx;
        smn[wave] = acc.mn;
    }
    __syncthreads();
    if (threadIdx.x == 0 && o1 > o0) {
        for (int i = 1; i < kBlock / 64; ++i) {
            acc.mx = smx[i] > acc.mx ? smx[i] : acc.mx;
            acc.mn = smn[i] < acc.mn ? smn[i] : acc.mn;
        }
        acc.flush(out_max, out_min, c);
    }
}

// per channel with a short inner extent (< 64): one thread per (c, inner) column, loop over outer
template <int DT, bool MINMAX>
__global__ __launch_bounds__(kBlock) void reduce_cols_kernel(const void* __restrict__ x, int64_t outer, int64_t cols,
                                                              int64_t inner, int64_t outer_per_block,
                                                              uint32_t* out_max, uint32_t* out_min, ActSpec relu) {
    const int64_t col = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (col >= cols) return;
    const int64_t o0 = (int64_t)blockIdx.y * outer_per_block;
    const int64_t o1 = o0 + outer_per_block < outer ? o0 + outer_per_block : outer;
    RedAcc<DT, MINMAX> acc;
    acc.act = relu;
    for (int64_t o = o0; o < o1; ++o) acc.add(load1<DT>(x, o * cols + col));
    if (o1 > o0) acc.flush(out_max, out_min, (uint32_t)(col / inner));
}

// same, 8 adjacent columns per thread with 16-byte loads (cols % 8 == 0, aligned base), 8 rows in flight
template <int DT, bool MINMAX>
__global__ __launch_bounds__(kBlock) void reduce_cols_vec_kernel(const void* __restrict__ x, int64_t outer, int64_t cols,
                                                                  int64_t inner, int64_t outer_per_block,
                                                                  uint32_t* out_max, uint32_t* out_min, ActSpec relu) {
    const int64_t gcols = cols / 8;
    const int64_t gc = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t o0 = (int64_t)blockIdx.y * outer_per_block;
    const int64_t o1 = o0 + outer_per_block < outer ? o0 + outer_per_block : outer;
    RedAcc<DT, MINMAX> acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j].act = relu;
    for (int64_t o = o0; o < o1 && gc < gcols; o += 8) {
        Raw8<DT> r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (o + u < o1) r[u] = load8_raw<DT, false>(x, (o + u) * gcols + gc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (o + u < o1) {
                float v[8];
                unpack8<DT>(r[u], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j].add(v[j]);
            }
        }
    }
    // combine inside the workgroup first (LDS atomics, indexed by channel relative to the block's first one), then
    // one global atomic per (workgroup, channel): global same-address atomics are what limits this kernel
    constexpr int kLocal = kBlock * 8 + 2;
    __shared__ uint32_t lmx[kLocal], lmn[kLocal];
    const int64_t col_first = (int64_t)blockIdx.x * kBlock * 8;
    const uint32_t c_block = (uint32_t)(col_first / inner);
    int64_t col_last = col_first + (int64_t)kBlock * 8 - 1;
    if (col_last >= cols) col_last = cols - 1;
    const int nlocal = (int)(col_last / inner - c_block) + 1;
    for (int i = threadIdx.x; i < nlocal; i += kBlock) {
        lmx[i] = 0u;
        lmn[i] = 0xffffffffu;
    }
    __syncthreads();
    if (gc < gcols && o1 > o0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int li = (int)((gc * 8 + j) / inner - c_block);
            acc[j].fold();
            atomicMax(&lmx[li], acc[j].mx);
            if constexpr (MINMAX) atomicMin(&lmn[li], acc[j].mn);
        }
    }
    __syncthreads();
    if (o1 > o0) {
        for (int i = threadIdx.x; i < nlocal; i += kBlock) {
            atomicMax(out_max + c_block + i, lmx[i]);
            if constexpr (MINMAX) atomicMin(out_min + c_block + i, lmn[i]);
        }
    }
}

// ---- few columns, many rows (channels_last activations viewed as [N*H*W, C]; 2-d [batch, features]) -----------------
// With cols <= 512 the kernel above leaves most lanes of a workgroup without a column.  Here a workgroup reads
// kBlock / (cols/8) whole rows per load instruction (consecutive lanes = consecutive 16-byte groups of consecutive rows,
// i.e. one contiguous span), strides over its share of the rows with eight loads in flight, folds its lanes per column
// through LDS and writes ONE partial row; a second, tiny launch folds the partial rows into the result.  No global
// atomics: 512 workgroups x C channels of them cost more than the whole read (64x64x56x56 bf16: 76 us -> 12 us).
constexpr int kFewColsMaxBlocks = 512;   // 2 workgroups per CU (measured: 256 -> 98 us, 512 -> 73 us, 1024 -> 76 us + a longer finish on 411 MB)
constexpr int kFewColsMaxCols = 512;
#ifndef QS_FEWCOLS_INFLIGHT
#define QS_FEWCOLS_INFLIGHT 8      // 16-byte loads in flight per lane (16: no faster -- channels_last min/max of the headline tensor 101 vs 105 us)
#endif

template <int DT, bool MINMAX>
__global__ __launch_bounds__(kBlock) void reduce_fewcols_kernel(const void* __restrict__ x, int64_t outer, int64_t cols,
                                                                 uint32_t* __restrict__ part_max,
                                                                 uint32_t* __restrict__ part_min, ActSpec relu) {
    __shared__ uint32_t lmx[kFewColsMaxCols], lmn[kFewColsMaxCols];
    const int gcols = (int)(cols / 8);
    const int rows_per_iter = kBlock / gcols;
    const int row_l = threadIdx.x / gcols, gc = threadIdx.x - row_l * gcols;
    // grid-stride over blocks of 8 * rows_per_iter rows: at any moment the whole grid reads ONE moving window of
    // gridDim.x * 32 KiB (as reduce_all_kernel does) instead of gridDim.x far-apart private streams, which cost DRAM page
    // locality (5.5 TB/s where the tensor-wise kernel streams at 6.3)
    const int64_t o0 = (int64_t)blockIdx.x * rows_per_iter * QS_FEWCOLS_INFLIGHT, o1 = outer;
    const int64_t ostride = (int64_t)gridDim.x * rows_per_iter * QS_FEWCOLS_INFLIGHT;
    for (int i = threadIdx.x; i < cols; i += kBlock) {
        lmx[i] = 0u;
        lmn[i] = 0xffffffffu;
    }
    RedAcc<DT, MINMAX> acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j].act = relu;
    if (row_l < rows_per_iter) {
        for (int64_t o = o0 + row_l; o < o1; o += ostride) {
            constexpr int F = QS_FEWCOLS_INFLIGHT;
            Raw8<DT> r[F];
#pragma unroll
            for (int u = 0; u < F; ++u)
                if (o + (int64_t)u * rows_per_iter < o1) r[u] = load8_raw<DT, false>(x, (o + (int64_t)u * rows_per_iter) * gcols + gc);
#pragma unroll
            for (int u = 0; u < F; ++u) {
                if (o + (int64_t)u * rows_per_iter < o1) {
                    float v[8];
                    unpack8<DT>(r[u], v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j].add(v[j]);
                }
            }
        }
    }
    __syncthreads();
    if (row_l < rows_per_iter) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j].fold();
            atomicMax(&lmx[gc * 8 + j], acc[j].mx);
            if constexpr (MINMAX) atomicMin(&lmn[gc * 8 + j], acc[j].mn);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cols; i += kBlock) {
        part_max[(int64_t)blockIdx.x * cols + i] = lmx[i];
        if constexpr (MINMAX) part_min[(int64_t)blockIdx.x * cols + i] = lmn[i];
    }
}

// partial rows -> result (max-accumulated into out_max / min-accumulated into out_min, like the atomics of the other
// kernels).  A workgroup owns 16 channels; its 16 lane groups share the partial rows, four loads in flight each.
template <bool MINMAX>
__global__ __launch_bounds__(kBlock) void reduce_fewcols_finish_kernel(const uint32_t* __restrict__ part_max,
                                                                        const uint32_t* __restrict__ part_min, int nblk,
                                                                        int64_t cols, int64_t inner, uint32_t* out_max,
                                                                        uint32_t* out_min, int finalize) {
    __shared__ uint32_t smx[kBlock], smn[kBlock];
    const int ch_l = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t ch = (int64_t)blockIdx.x * 16 + ch_l, nch = cols / inner;
    uint32_t mx = 0u, mn = 0xffffffffu;
    if (ch < nch) {
        for (int64_t j = 0; j < inner; ++j) {
            const int64_t col = ch * inner + j;
            for (int b = rg; b < nblk; b += 64) {
                uint32_t a[4], c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = b + 16 * u < nblk;
                    a[u] = ok ? part_max[(int64_t)(b + 16 * u) * cols + col] : 0u;
                    c[u] = (MINMAX && ok) ? part_min[(int64_t)(b + 16 * u) * cols + col] : 0xffffffffu;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    mx = a[u] > mx ? a[u] : mx;
                    mn = c[u] < mn ? c[u] : mn;
                }
            }
        }
    }
    smx[threadIdx.x] = mx;
    smn[threadIdx.x] = mn;
    __syncthreads();
    if (rg == 0 && ch < nch) {
        for (int g = 1; g < kBlock / 16; ++g) {
            mx = smx[g * 16 + ch_l] > mx ? smx[g * 16 + ch_l] : mx;
            mn = smn[g * 16 + ch_l] < mn ? smn[g * 16 + ch_l] : mn;
        }
        if (finalize) {      // a non-accumulating call: the final floats, no initialisation / conversion launches around us
            ((float*)out_max)[ch] = MINMAX ? key_to_f32(mx) : __uint_as_float(mx);
            if constexpr (MINMAX) ((float*)out_min)[ch] = key_to_f32(mn);
        } else {
            out_max[ch] = mx > out_max[ch] ? mx : out_max[ch];
            if constexpr (MINMAX) out_min[ch] = mn < out_min[ch] ? mn : out_min[ch];
        }
    }
}

static __global__ void keys_init_kernel(uint32_t* mx, uint32_t* mn, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mx[i] = 0u;
        if (mn) mn[i] = 0xffffffffu;
    }
}
static __global__ void keys_to_float_kernel(uint32_t* mx, uint32_t* mn, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ((float*)mx)[i] = key_to_f32(mx[i]);
        ((float*)mn)[i] = key_to_f32(mn[i]);
    }
}

// =================================================================================================
// C-sized state updates
// =================================================================================================
// `t_dev` (nullable): device-resident step counter read instead of the by-value `t`, so that a captured hipGraph
// replays with the live counter (by-value kernel arguments are frozen at capture time)
// `clear` != 0 zeroes absmax[i] after use (accumulate-mode abs-max buffers); `bump` (nullable) is a one-element
// int32 step counter incremented once (the layer's `_n_updates`, quantize.py:515)
// The reference divides in the dtype of the statistic (`x.abs().max() / 2**(bits-1)` on an fp16 / bf16 tensor): the
// quotient is rounded to that dtype before it enters the float32 running mean.  Exact for bf16 (same exponent range as
// fp32) but not for fp16, where small maxima underflow into subnormals (max|x| = 1e-3, 8 bits: 7.8082e-6, not 7.8157e-6).
static __global__ void scale_update_kernel(float* absmax, float* weight, int64_t n, float t, float tp1, float denom,
                                    int64_t* t_dev, int advance, int clear, int32_t* bump, int stat_dt, int lines) {
    if (lines > 1) {
        // n == 1 (one workgroup): the tensor-wise abs-max arrives in `lines` (<= 64) partial accumulators on lines of their
        // own (reduce_all_kernel).  Lane l fetches line l and the counter / the old scale are fetched alongside: ONE memory
        // round trip.  (A loop over the lines in one thread was sixteen dependent round trips -- 6.8 us per launch in a
        // ResNet-18 step, rocprofv3, for a kernel that computes one number.)
        const int l = threadIdx.x;
        uint32_t k = (l < lines) ? __float_as_uint(absmax[(int64_t)l * QS_AMAX_LINE_STRIDE]) : 0u;
        const float w_old = weight[0];
        if (t_dev) {
            t = (float)*t_dev;
            tp1 = (float)(*t_dev + 1);
        }
        if (clear && l < lines) absmax[(int64_t)l * QS_AMAX_LINE_STRIDE] = 0.0f;
        if (l < 64) k = wave_max_u32(k);      // keys of |x|: non-negative floats (and NaN on top) order like their bits
        if (l == 0) {
            const float nw = round_to_dtype(__uint_as_float(k) / denom, stat_dt);   // max / 2**(bits-1) in x's dtype (quantize.py:340)
            weight[0] = (t == 0.0f) ? nw : (t * w_old + nw) / tp1;                  // (:344-347)
        }
    } else {
        if (t_dev) {
            t = (float)*t_dev;
            tp1 = (float)(*t_dev + 1);
        }
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const float nw = round_to_dtype(absmax[i] / denom, stat_dt);   // max / 2**(bits-1) in x's dtype (quantize.py:340)
            weight[i] = (t == 0.0f) ? nw : (t * weight[i] + nw) / tp1;     // (:344-347)
            if (clear) absmax[i] = 0.0f;
        }
    }
    if (bump && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(bump, 1);
    if (advance && t_dev) {      // single-workgroup launch: every thread has read the counter before it moves
        __syncthreads();
        if (threadIdx.x == 0) *t_dev += 1;
    }
}
// `from_keys`: mn / mx hold the order-preserving keys qs_minmax(accumulate) leaves behind; they are turned into floats here
// and reset to the neutral keys (max 0, min 0xffffffff) for the next statistics pass -- which saves the key-initialisation
// and the key -> float launches of a min/max call
static __global__ void lines_update_kernel(float* mn, float* mx, float* lines, int64_t n, float tm1, float t,
                                    int64_t* t_dev, int advance, int from_keys) {
    if (t_dev) {   // counter BEFORE this step's increment
        tm1 = (float)*t_dev;
        t = (float)(*t_dev + 1);
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float lo = mn[i], hi = mx[i];
        if (from_keys) {
            lo = key_to_f32(__float_as_uint(lo));
            hi = key_to_f32(__float_as_uint(hi));
            ((uint32_t*)mn)[i] = 0xffffffffu;
            ((uint32_t*)mx)[i] = 0u;
        }
        lines[2 * i] = (lines[2 * i] * tm1 + lo) / t;          // (quantize.py:430)
        lines[2 * i + 1] = (lines[2 * i + 1] * tm1 + hi) / t;
    }
    if (advance && t_dev) {
        __syncthreads();
        if (threadIdx.x == 0) *t_dev += 1;
    }
}
static __global__ void decimal_from_scale_kernel(const float* scale, float* d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float r = 1.0f / scale[i];
        if (r == __builtin_inff() || r == -__builtin_inff()) r = 1.0f;   // nan_to_num(posinf=1, neginf=1)
        if (r != r) r = 0.0f;                                             // nan -> 0
        d[i] = rintf(log2f(r));
    }
}
template <int DT>
__global__ void running_mean_kernel(float* state, const void* nv, int64_t n, float t, float tp1, const int64_t* t_dev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t_dev) {
        t = (float)*t_dev;
        tp1 = (float)(*t_dev + 1);
    }
    if (i < n) state[i] = (t * state[i] + load1<DT>(nv, i)) / tp1;       // sparse.py:89
}
static __global__ void l0_flag_kernel(const float* mn, int32_t* flag) { *flag = (mn[0] == 0.0f) ? 1 : 0; }

// =================================================================================================
// Staged mean in ATen's CPU order.  at::mean on CPU (ReduceOps.cpp, mean_out) is
//     sum(x.to(float32)) -> div_(n) -> to(dtype)
// and the fp32 sum kernel (cpu/SumKernel.cpp, AVX2 build also on AVX-512 hosts: Vectorized<float> has 8
// lanes) adds in one of two per-column orders:
//   "multi-row":  sequential over rows in chunks of 2^p (p = max(4, ceil_log2(n)/4)), chunk sums
//                 cascaded through 4 levels, levels added at the end;
//   "row-sum":    rows split 4-way (i mod 4), each part summed in multi-row order over n/4 items, the
//                 n%4 leftover rows added to part 0, then parts 0+1+2+3.
// Outer reduction ([n, post], post contiguous): columns below 32*floor(post/32) use multi-row when
// post >= 8, columns below 4*floor(post/4) when post < 8; all other columns use row-sum.
// Inner reduction (post == 1, n >= 8): 8 lane accumulators over n/8 vectors in row-sum order, then the
// n%8 tail summed sequentially from zero, then the 8 lanes added one by one.  n < 8: row-sum over elements.
// =================================================================================================
__device__ __forceinline__ int ceil_log2_i64(int64_t x) { return x <= 1 ? 0 : 64 - __builtin_clzll((uint64_t)(x - 1)); }

struct Cascade {  // 4-level cascade accumulator for ONE column
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    __device__ __forceinline__ void add(float v) { a0 += v; }
    // called after `i` rows have been consumed, i a multiple of the level step
    __device__ __forceinline__ void carry(int64_t i, int lp, int64_t lmask) {
        a1 += a0;
        a0 = 0.f;
        if ((i & (lmask << lp)) != 0) return;
        a2 += a1;
        a1 = 0.f;
        if ((i & (lmask << (2 * lp))) != 0) return;
        a3 += a2;
        a2 = 0.f;
    }
    __device__ __forceinline__ float total() const { return ((a0 + a1) + a2) + a3; }
};

template <int DT>
__device__ __forceinline__ float mean_prep(float v, int flags, int l0, const ActSpec& act) {
    if (flags & QS_MEAN_RELU) v = act_apply(v, act, DT);   // folded preceding activation
    if (l0) return (v != 0.0f) ? 1.0f : 0.0f;       // (x != 0).float()  (sparse.py:86)
    return (flags & QS_MEAN_ABS) ? fabsf(v) : v;    // x.abs()           (sparse.py:87)
}

// the same with the common flag combinations fixed at compile time (PREP 1: |x|, 2: |max(x, 0)|, 3: x; 0: whatever the flags say):
// the one-lane-per-output kernels request sixteen loads ahead of their ordered adds, and run-time tests between them serialise
// those loads (256 x 150528 fp32 with a folded ReLU: 81 us with the run-time flags, 25 with PREP 2)
template <int DT, int PREP>
__device__ __forceinline__ float mean_prep_t(float v, int flags, int l0, const ActSpec& act) {
    if constexpr (PREP == 1) return fabsf(v);
    else if constexpr (PREP == 2) return fabsf(relu_aten(v));
    else if constexpr (PREP == 3) return v;
    else return mean_prep<DT>(v, flags, l0, act);
}

// multi-row order of element sequence get(0..n-1)
template <typename F>
__device__ __forceinline__ float sum_multi_row(int64_t n, F get) {
    const int lp = max(4, ceil_log2_i64(n) / 4);
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    Cascade c;
    int64_t i = 0;
    // operands are fetched in batches (16 / 8 / 1) ahead of the strictly ordered adds, so that their
    // latencies (global or LDS) overlap; the order of the additions is untouched
    while (i + step <= n) {
        for (int64_t j = 0; j < step; j += 16, i += 16) {   // step is a power of two >= 16
            float b[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) b[u] = get(i + u);
#pragma unroll
            for (int u = 0; u < 16; ++u) c.add(b[u]);
        }
        c.carry(i, lp, lmask);
    }
    for (; i + 8 <= n; i += 8) {
        float b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = get(i + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) c.add(b[u]);
    }
    for (; i < n; ++i) c.add(get(i));
    return c.total();
}
// row-sum order
template <typename F>
__device__ __forceinline__ float sum_row_sum(int64_t n, F get) {
    const int64_t n4 = n / 4;
    float p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = sum_multi_row(n4, [&](int64_t i) { return get(4 * i + k); });
    for (int64_t i = n4 * 4; i < n; ++i) p[0] += get(i);
    return ((p[0] + p[1]) + p[2]) + p[3];
}

// copy tile p (hw elements of dtype DT) into LDS as fp32: 16-byte global loads when the tile is 8-element aligned
template <int DT>
__device__ __forceinline__ void load_tile_f32(const void* src, int64_t p, int hw, float* tile) {
    if ((hw & 7) == 0 && (((uintptr_t)src) & 15) == 0) {
        const int64_t g0 = p * (hw / 8);
        for (int g = threadIdx.x; g < hw / 8; g += blockDim.x) {
            float v[8];
            unpack8<DT>(load8_raw<DT, false>(src, g0 + g), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[g * 8 + j] = v[j];
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += blockDim.x) tile[i] = load1<DT>(src, p * hw + i);
    }
}

// inner-sum order over `W` values held in LDS, evaluated by lanes 0..7 of one wave: lane k owns the k-th
// vector lane (row-sum order over W/8 items), lane 0 then adds the W%8 tail and the 8 lane sums in order.
__device__ __forceinline__ float inner_sum_lds(const float* v, int W, float* lane_out /* 8 floats, LDS */) {
    if (W >= 8) {
        const int nv = W / 8;
        if (threadIdx.x < 8) {
            const int k = threadIdx.x;
            lane_out[k] = sum_row_sum(nv, [&](int64_t i) { return v[8 * i + k]; });
        }
        __syncthreads();
        float fin = 0.f;
        if (threadIdx.x == 0) {
            for (int i = nv * 8; i < W; ++i) fin += v[i];
#pragma unroll
            for (int k = 0; k < 8; ++k) fin += lane_out[k];
        }
        return fin;
    }
    __syncthreads();
    return threadIdx.x == 0 ? sum_row_sum(W, [&](int64_t i) { return v[i]; }) : 0.f;
}

// The big statistics pass reads x with NON-TEMPORAL loads: it streams the tensor once in a row-strided pattern
// and, when it allocates in the Infinity Cache, it is the kernel that has to push out the previous kernel's dirty
// lines -- which it does badly (0.138 ms in-step vs 0.084 ms alone).  Without allocation the next streaming
// kernel pays instead and does it better: step 0.554 -> 0.538 ms (apply forward 0.189 -> 0.215 ms, statistics
// 0.138 -> 0.098 ms).  QS_MEAN_NT_LOADS=false restores cached loads.
#ifndef QS_MEAN_NT_LOADS
#define QS_MEAN_NT_LOADS true
#endif
// The statistics kernels of SMALLER tensors and of channels_last activations read with ordinary (allocating) loads: what
// they leave in the 256 MiB Infinity Cache is what the apply-forward kernel reads next (its reverse walk starts where
// this pass ended).  Measured inside a ResNet-50 step (batch 256, channels_last; rocprofv3, tools/profile_model.sh):
// apply forward 2.877 -> 2.545 ms per step (0.77 -> 0.87 of the roofline), statistics 2.13 -> 2.10 ms, backward
// unchanged -- the 822 MB fp32 sites gain as well (their last 256 MB are re-read from the cache: 114.8 -> 105.7 us).
// The big NCHW column walk above keeps its non-temporal loads (its row-strided pattern leaves nothing useful behind).
#ifndef QS_MEAN_SMALL_NT_LOADS
#define QS_MEAN_SMALL_NT_LOADS false
#endif
// ---- hot stage: outer reduction, 8 adjacent columns per lane, multi-row order ---------------------
// x: [pre, n, post] contiguous; handles columns [0, vcols) of every `pre` slice (vcols % 8 == 0).
// Optionally accumulates per-channel max|x| (channel = (col / chan_div) % C) for a fused abs-max.
// Launch geometry: ONE wave per workgroup, of which only the first `lanes` (<= 64) lanes own a column group.
// A CU sustains ~10 B/clk from HBM whatever runs on it, so the kernel is only as fast as its busiest CU:
// the host picks `lanes` such that the number of waves is (close to) a multiple of the 256 CUs -- e.g.
// 100352 column groups -> 56 lanes x 1792 waves = exactly 7 waves per CU instead of 64 x 1568 (6.1, i.e.
// 7 on some CUs and 6 on others).
// PERCOL (MODE 1 / 2 only): every column has a channel of its own (a token-major [B][T][C] activation reduced over B: channel =
// column % C, a lane's 8 columns are 8 channels).  The abs-max is kept per COLUMN and stored -- no atomics -- into `absmax`, which
// is then a [pre][post] array of keys ("amax_part", as the channels_last kernels leave one); token_amax_fold_kernel folds it per
// channel.  (Max-accumulating per column with atomics, T of them per channel, cost 11 us on 256 x 197 x 3072 and 100 us on
// 64 x 1024 x 4096 -- more than the 97 us the whole column walk takes there.)
template <int DT, int ODT, int ROWS_IN_FLIGHT, int MODE, bool PERCOL = false>
__global__ __launch_bounds__(64) void mean_outer_vec_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                             int64_t pre, int64_t n, int64_t post, int64_t vcols,
                                                             int flags, const int32_t* __restrict__ l0_flag,
                                                             uint32_t* __restrict__ absmax, int64_t astride,
                                                             int64_t chan_div, uint32_t C, int lanes,
                                                             ActSpec act = ActSpec{QS_ACT_RELU, 0.f, 0.f}) {
    const int64_t gcols = vcols / 8;                 // column groups per slice
    const int64_t total = pre * gcols;
    const int64_t t = (int64_t)blockIdx.x * lanes + threadIdx.x;
    const bool active = (int)threadIdx.x < lanes && t < total;
    const int l0 = (flags & QS_MEAN_L0) && l0_flag && *l0_flag;
    const int lp = max(4, ceil_log2_i64(n) / 4);
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    const float fn = (float)n;

    const int64_t tt = active ? t : 0;
    const int64_t p = tt / gcols, gc = tt - p * gcols;
    const int64_t row_groups = post / 8;             // 16-byte groups per row (post % 8 == 0 guaranteed)
    const int64_t g_base = p * n * row_groups + gc;
    uint32_t amax = 0u;
    uint32_t amaxc[PERCOL ? 8 : 1] = {};
    static_assert(!PERCOL || MODE == 1 || MODE == 2, "per-column abs-max: the |x| / max(x, 0) modes");
    RedAcc<DT, true> mm;                             // MODE 6: per-channel min and max (qs_minmax's column walk)

    if (active) {
        Cascade acc[8];
        // MODE 1 (|x|) and 2 (max(x,0)) are the hot configurations: the mean operand and the abs-max key are the
        // same non-negative value, 3-5 VALU ops per element (the wave count per CU is low, so VALU time is not
        // hidden); MODE 3 is the plain mean without abs-max; MODE 0 handles every other flag combination
        auto consume = [&](const Raw8<DT>& r) {
            float v[8];
            unpack8<DT>(r, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (MODE == 3) {          // plain mean (squeeze_tensor_to_shape on its own): no abs, no abs-max
                    acc[j].add(v[j]);
                } else if constexpr (MODE == 0) {
                    if (absmax) {
                        const float av = (flags & QS_MEAN_RELU) ? act_apply(v[j], act, DT) : v[j];
                        const uint32_t k = __float_as_uint(av) & 0x7fffffffu;
                        amax = k > amax ? k : amax;
                    }
                    acc[j].add(mean_prep<DT>(v[j], flags, l0, act));
                } else if constexpr (MODE == 6) {   // min / max ONLY, no sum, no output: the column walk of a per-channel min/max
                    mm.add(v[j]);
                } else if constexpr (MODE == 4 || MODE == 5) {   // abs-max ONLY (4: |x|, 5: max(x, 0)): no sum, no output -- the column walk
                    const float w = (MODE == 5) ? relu_aten(v[j]) : v[j];     // as the per-channel abs-max of a big tensor
                    const uint32_t k = __float_as_uint(w) & 0x7fffffffu;
                    amax = k > amax ? k : amax;
                } else {                            // MODE 1, 2 and -- a folded nn.Hardtanh / nn.ReLU6 (7), nn.LeakyReLU (8) -- |act(x)|
                    const float w = (MODE == 2) ? relu_aten(v[j]) : (MODE == 7) ? act_apply_k<QS_ACT_HARDTANH>(v[j], act, DT)
                                  : (MODE == 8) ? act_apply_k<QS_ACT_LEAKY>(v[j], act, DT) : v[j];
                    const uint32_t k = __float_as_uint(w) & 0x7fffffffu;
                    if constexpr (PERCOL) amaxc[j] = k > amaxc[j] ? k : amaxc[j];
                    else amax = k > amax ? k : amax;
                    acc[j].add(__uint_as_float(k));
                }
            }
        };

        int64_t i = 0;
        if constexpr (ROWS_IN_FLIGHT > 16) {
            // deep variant for launches with few waves per CU (the host picks it): several 16-row chunks are
            // fetched ahead, the adds and carries keep their order.  step == 16 for every n < 2^20.
            if (step == 16) {
                while (i + ROWS_IN_FLIGHT <= n) {
                    Raw8<DT> r[ROWS_IN_FLIGHT];
#pragma unroll
                    for (int u = 0; u < ROWS_IN_FLIGHT; ++u)
                        r[u] = load8_raw<DT, QS_MEAN_NT_LOADS>(x, g_base + (i + u) * row_groups);
                    __builtin_amdgcn_sched_barrier(0);      // all loads are issued before the first add (the scheduler would interleave)
#pragma unroll
                    for (int c = 0; c < ROWS_IN_FLIGHT / 16; ++c) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) consume(r[c * 16 + u]);
                        i += 16;
#pragma unroll
                        for (int j = 0; j < 8; ++j) if constexpr (MODE < 4 || MODE >= 7) acc[j].carry(i, lp, lmask);
                    }
                }
            }
        }
        constexpr int kBatch = ROWS_IN_FLIGHT > 16 ? 16 : ROWS_IN_FLIGHT;
        while (i + step <= n) {
            // step is a power of two >= 16, so it is a multiple of kBatch (8 or 16)
            for (int64_t j = 0; j < step; j += kBatch) {
                Raw8<DT> r[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u)
                    r[u] = load8_raw<DT, QS_MEAN_NT_LOADS>(x, g_base + (i + j + u) * row_groups);
                // (no sched_barrier here: hipcc's own interleaving of these loads and adds measured equal for 2-byte
                //  inputs and 15-20 % faster for fp32 than 16 loads up front -- unlike in mean_cl_kernel below)
#pragma unroll
                for (int u = 0; u < kBatch; ++u) consume(r[u]);
            }
            i += step;
#pragma unroll
            for (int j = 0; j < 8; ++j) if constexpr (MODE < 4 || MODE >= 7) acc[j].carry(i, lp, lmask);
        }
        for (; i < n; ++i) consume(load8_raw<DT, false>(x, g_base + i * row_groups));

        if constexpr (MODE < 4 || MODE >= 7) {
            float m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = acc[j].total() / fn;   // .div_(n) in fp32, then one rounding to ODT
            store8<ODT, false>(out, p * row_groups + gc, m);
        }
    }

    if constexpr (MODE == 6) {
        // keys as RedAcc keeps them (NaN on top of the maxima); `out` carries the MIN keys of this mode, `absmax` the MAX
        // keys; the host guarantees chan_div % 8 == 0 (a lane inside one channel).  Idle lanes hold the neutral keys.
        uint32_t* kmin = (uint32_t*)out;
        mm.fold();
        const uint32_t c = (uint32_t)(((gc * 8) / chan_div) % C);
        const uint32_t c0 = (uint32_t)__shfl((int)c, 0, 64);
        if (__all(!active || c == c0)) {
            const uint32_t hi = wave_max_u32(mm.mx), lo = wave_min_u32(mm.mn);
            if (threadIdx.x == 0) {
                atomicMax(absmax + (size_t)c0 * astride, hi);
                atomicMin(kmin + (size_t)c0 * astride, lo);
            }
        } else if (active) {
            atomicMax(absmax + (size_t)c * astride, mm.mx);
            atomicMin(kmin + (size_t)c * astride, mm.mn);
        }
        return;
    }
    if constexpr (PERCOL) {
        if (absmax && active) {
            u32x4* dst = (u32x4*)(absmax + p * post + gc * 8);
            dst[0] = u32x4{amaxc[0], amaxc[1], amaxc[2], amaxc[3]};
            dst[1] = u32x4{amaxc[4], amaxc[5], amaxc[6], amaxc[7]};
        }
        return;
    }
    if (absmax) {   // whole wave takes part: idle lanes contribute 0
        if (chan_div % 8 == 0) {   // all 8 columns of a lane share a channel
            const uint32_t c = (uint32_t)(((gc * 8) / chan_div) % C);
            const uint32_t c0 = (uint32_t)__shfl((int)c, 0, 64);
            if (__all(!active || c == c0)) {
                amax = wave_max_u32(amax);
                if (threadIdx.x == 0) atomicMax(absmax + (size_t)c0 * astride, amax);
            } else if (active) {
                atomicMax(absmax + (size_t)c * astride, amax);
            }
        } else if (active) {       // ragged rows (e.g. 7x7 maps): handled per column by the caller's generic kernel
            atomicMax(absmax + (size_t)(uint32_t)(((gc * 8) / chan_div) % C) * astride, amax);
        }
    }
}

// chan_absmax[c * astride] <- max(chan_absmax[c * astride], max over t of part[t][c]) for a [T][C] array of per-column keys (the PERCOL
// form above).  grid (ceil(C / 64), ceil(T / 32)), 256 threads: wave w of a workgroup takes 8 tokens of its 32-token slice for
// 64 channels (coalesced 256-byte rows, eight loads in flight), the four waves meet in LDS, one atomic per channel and slice
// (slices of 128 tokens, four dependent batches per wave: 11.4 us on 197 x 3072; of 32: one batch).
static __global__ __launch_bounds__(256) void token_amax_fold_kernel(const uint32_t* __restrict__ part, int64_t T, int64_t C,
                                                                      uint32_t* __restrict__ chan_absmax, int64_t astride) {
    __shared__ uint32_t sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t t0 = (int64_t)blockIdx.y * 32 + w * 8;
    uint32_t m = 0u;
    if (c < C) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t t = t0 + u;
            v[u] = t < T ? part[t * C + c] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) m = v[u] > m ? v[u] : m;
    }
    sh[w][lane] = m;
    __syncthreads();
    if (w == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 4; ++k) m = sh[k][lane] > m ? sh[k][lane] : m;
        atomicMax(chan_absmax + (size_t)c * astride, m);
    }
}

// ---- the same stage for channels_last activations ---------------------------------------------------------
// x: [n][hw][C] in memory (NHWC), C % 8 == 0; out: [C][hw], i.e. NCHW-contiguous -- which is what ATen's reduction of
// a channels_last input returns, so the later stages are the ordinary NCHW ones.  ATen sums such an input over N with
// its scalar outer loop (SumKernel.cpp, scalar_outer_sum over the coalesced H*W dim): per channel, positions
// hw < 4*floor(HW/4) in multi-row order, the remaining HW % 4 positions in row-sum order.  A lane owns 8 consecutive
// channels of one position (one 16-byte load per row).  amax_part (nullable) receives, per output element, the
// maximum over n of the mean's operand (|x| or max(x, 0)) as a uint32 key; the following qs_mean_last2 launch, which
// runs one workgroup per channel anyway, reduces it to the per-channel abs-max -- no atomics at all.
// MODE 1: |x|, 2: max(x, 0), 3: x as it is (no abs-max).
// MODE 4: |act(x)| for a folded activation other than nn.ReLU (its descriptor and the input dtype travel as arguments)
template <int MODE>
__device__ __forceinline__ float mean_cl_prep(float v, uint32_t& am, const ActSpec& act, int dt) {
    if constexpr (MODE == 3) return v;
    // (5 / 6: the kind fixed at compile time -- nn.Hardtanh / nn.ReLU6, nn.LeakyReLU; 4 is left for kinds to come)
    const float w = (MODE == 4) ? act_apply(v, act, dt) : (MODE == 5) ? act_apply_k<QS_ACT_HARDTANH>(v, act, dt)
                  : (MODE == 6) ? act_apply_k<QS_ACT_LEAKY>(v, act, dt) : ((MODE == 2) ? relu_aten(v) : v);
    const uint32_t k = __float_as_uint(w) & 0x7fffffffu;
    am = k > am ? k : am;
    return __uint_as_float(k);
}

// positions hw < 4*floor(HW/4): multi-row order (the host launches it over those positions only)
template <int DT, int ODT, int MODE>
__global__ __launch_bounds__(64) void mean_cl_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t n,
                                                      int64_t hw, int64_t C, uint32_t* __restrict__ amax_part, int lanes,
                                                      int64_t ngroups, ActSpec act, int xcd_per) {
    const int64_t groups = hw * C / 8;                 // 16-byte groups per row of x
    // xcd_per > 0: the launch has 8 * xcd_per workgroups and workgroup b (dispatched to XCD b % 8) takes the wave
    // (b % 8) * xcd_per + b / 8 -- every XCD owns one contiguous eighth of the columns, so the single means and abs-max keys
    // that neighbouring waves store into the same output lines (H*W apart per lane, adjacent across waves) meet in ONE L2 and
    // leave it as full lines instead of as partial writes from eight L2s (a speed choice only: placement is not a contract)
    const int64_t b = xcd_per > 0 ? (int64_t)(blockIdx.x & 7u) * xcd_per + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    const int64_t t = b * lanes + threadIdx.x;
    if ((int)threadIdx.x >= lanes || t >= ngroups) return;
    const int64_t col0 = t * 8;
    const int64_t pos = col0 / C, c0 = col0 - pos * C;
    const int lp = max(4, ceil_log2_i64(n) / 4);
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    uint32_t amax[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) amax[j] = 0u;
    Cascade acc[8];
    auto consume = [&](const Raw8<DT>& r) {
        float v[8];
        unpack8<DT>(r, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j].add(mean_cl_prep<MODE>(v[j], amax[j], act, DT));
    };
    int64_t i = 0;
    while (i + step <= n) {
        for (int64_t j = 0; j < step; j += 16) {
            Raw8<DT> r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = load8_raw<DT, QS_MEAN_SMALL_NT_LOADS>(x, t + (i + j + u) * groups);
            __builtin_amdgcn_sched_barrier(0);          // all 16 rows in flight before the first add
#pragma unroll
            for (int u = 0; u < 16; ++u) consume(r[u]);
        }
        i += step;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j].carry(i, lp, lmask);
    }
    for (; i < n; ++i) consume(load8_raw<DT, false>(x, t + i * groups));
    const float fn = (float)n;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t o = (c0 + j) * hw + pos;
        store1<ODT>(out, o, acc[j].total() / fn);     // .div_(n) in fp32, then one rounding to ODT
        if (MODE != 3 && amax_part) amax_part[o] = amax[j];
    }
}

// mean_cl_kernel AND mean_cl_tail_kernel for launches with few waves (small maps, small batches), in ONE launch: the R
// waves of a workgroup own the same column groups and share the rows.  ATen's order per column is a cascade over level-0
// sums of 2^p consecutive items ("chunks": what its level-0 accumulator holds when it is dumped into level 1).  For a
// main position (hw < 4*floor(HW/4), multi-row order) the items are the n rows; for one of the HW % 4 tail positions
// (row-sum order) there are four interleaved sums k = 0..3 over the rows 4i + k, each a cascade of its own over n/4
// items.  Every (chunk, k) pair is an independent sum from zero -- a "slot" -- so the waves take slots round robin,
// park the slot sums in LDS, and wave 0 then feeds them IN ORDER through levels 1..3, adds the leftover items and
// finishes (tail: ((p0 + p1) + p2) + p3 with the n % 4 last rows added to p0 first).  The waves' abs-max keys are
// combined through LDS as well.  Workgroups [0, main_blocks) serve the main positions, the others the tail positions:
// the tail used to be a launch of its own whose single wave per 512 columns walked all n rows alone (256 x 512 x 7 x 7
// bf16: 32 of the stage's 47 us).  `lanes` (<= 64) lanes of each wave own a column group: narrow waves spread a small
// tensor over more workgroups.  LDS: [slots][8][lanes] floats + [R][8][lanes] keys.
template <int DT, int ODT, int R, int MODE>
__global__ __launch_bounds__(64 * R) void mean_cl_wg_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t n,
                                                             int64_t hw, int64_t C, uint32_t* __restrict__ amax_part,
                                                             int lanes, int64_t main_groups, int main_blocks,
                                                             int64_t tail_groups, int slots, ActSpec act, int xcd) {
    extern __shared__ __attribute__((aligned(16))) float cl_slot_sums[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t groups = hw * C / 8;                 // 16-byte groups per row of x
    const bool tail = (int)blockIdx.x >= main_blocks;
    // xcd != 0 (main_blocks is then a multiple of 8): XCD-contiguous order of the main workgroups, as in mean_cl_kernel
    const int mb = (!tail && xcd) ? (int)(blockIdx.x & 7u) * (main_blocks >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int64_t local = (int64_t)(tail ? (int)blockIdx.x - main_blocks : mb) * lanes + lane;
    const bool active = lane < lanes && local < (tail ? tail_groups : main_groups);
    const int64_t t = (tail ? main_groups : 0) + local;   // the group of every row this lane owns
    const int64_t items = tail ? n / 4 : n;            // items per cascade
    const int lp = max(4, ceil_log2_i64(items) / 4);
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    const int nch = (int)(items / step);               // full chunks per cascade
    const int nslots = tail ? 4 * nch : nch;
    uint32_t* amax_lds = (uint32_t*)(cl_slot_sums + (size_t)slots * 8 * lanes);
    uint32_t amax[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) amax[j] = 0u;
    if (active) {
        for (int v = wave; v < nslots; v += R) {
            // rows of slot v: main  v*step + e;  tail (chunk v >> 2 of sum v & 3)  4*((v >> 2)*step + e) + (v & 3)
            const int64_t r0 = tail ? 4 * (int64_t)(v >> 2) * step + (v & 3) : (int64_t)v * step;
            const int64_t rs = tail ? 4 : 1;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            for (int64_t e = 0; e < step; e += 16) {
                Raw8<DT> r[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) r[u] = load8_raw<DT, QS_MEAN_SMALL_NT_LOADS>(x, t + (r0 + (e + u) * rs) * groups);
                __builtin_amdgcn_sched_barrier(0);      // all 16 rows in flight before the first add
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    float w[8];
                    unpack8<DT>(r[u], w);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += mean_cl_prep<MODE>(w[k], amax[k], act, DT);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) cl_slot_sums[((size_t)v * 8 + k) * lanes + lane] = acc[k];
        }
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) amax_lds[((size_t)wave * 8 + k) * lanes + lane] = amax[k];
        }
    }
    __syncthreads();
    if (wave != 0 || !active) return;
    auto row = [&](int64_t i, auto add) {               // one leftover row, its 8 values handed to add(k, value)
        float w[8];
        unpack8<DT>(load8_raw<DT, false>(x, t + i * groups), w);
#pragma unroll
        for (int k = 0; k < 8; ++k) add(k, mean_cl_prep<MODE>(w[k], amax[k], act, DT));
    };
    float res[8];
    if (!tail) {
        Cascade c[8];
        for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c[k].a0 = cl_slot_sums[((size_t)ch * 8 + k) * lanes + lane];
                c[k].carry((int64_t)(ch + 1) * step, lp, lmask);
            }
        }
        for (int64_t i = (int64_t)nch * step; i < n; ++i) row(i, [&](int k, float val) { c[k].add(val); });
#pragma unroll
        for (int k = 0; k < 8; ++k) res[k] = c[k].total();
    } else {
        const int64_t n4 = items;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {                   // the four interleaved sums, one after the other (registers)
            Cascade c[8];
            for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    c[k].a0 = cl_slot_sums[((size_t)(ch * 4 + q) * 8 + k) * lanes + lane];
                    c[k].carry((int64_t)(ch + 1) * step, lp, lmask);
                }
            }
            for (int64_t e = (int64_t)nch * step; e < n4; ++e) row(4 * e + q, [&](int k, float val) { c[k].add(val); });
            if (q == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) res[k] = c[k].total();
                for (int64_t i = n4 * 4; i < n; ++i) row(i, [&](int k, float val) { res[k] += val; });   // n % 4 rows -> p0
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) res[k] += c[k].total();                                      // ((p0 + p1) + p2) + p3
            }
        }
    }
    for (int w = 1; w < R; ++w) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t o = amax_lds[((size_t)w * 8 + k) * lanes + lane];
            amax[k] = o > amax[k] ? o : amax[k];
        }
    }
    const int64_t col0 = t * 8;
    const int64_t pos = col0 / C, c0 = col0 - pos * C;
    const float fn = (float)n;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t o = (c0 + k) * hw + pos;
        store1<ODT>(out, o, res[k] / fn);               // .div_(n) in fp32, then one rounding to ODT
        if (MODE != 3 && amax_part) amax_part[o] = amax[k];
    }
}

// the remaining HW % 4 positions: row-sum order, i.e. per column four interleaved multi-row sums over n/4 rows each
// (row i feeds sum i % 4), the n % 4 last rows added to the first of them, then ((p0 + p1) + p2) + p3 -- sum_row_sum()
// with the 8 columns of a lane fetched by one 16-byte load per row.  A lane owns group `first_group + t` of every row.
template <int DT, int ODT, int MODE>
__global__ __launch_bounds__(64) void mean_cl_tail_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t n,
                                                           int64_t hw, int64_t C, uint32_t* __restrict__ amax_part,
                                                           int64_t first_group, int64_t ngroups, ActSpec act) {
    const int64_t groups = hw * C / 8;
    const int64_t tt = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tt >= ngroups) return;
    const int64_t t = first_group + tt;
    const int64_t col0 = t * 8;
    const int64_t pos = col0 / C, c0 = col0 - pos * C;
    const int64_t n4 = n / 4;
    const int lp = max(4, ceil_log2_i64(n4) / 4);
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    uint32_t amax[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) amax[j] = 0u;
    Cascade acc[4][8];
    auto consume = [&](const Raw8<DT>& r, Cascade (&a)[8]) {
        float v[8];
        unpack8<DT>(r, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j].add(mean_cl_prep<MODE>(v[j], amax[j], act, DT));
    };
    int64_t e = 0;                                      // elements consumed per interleaved sum
    while (e + step <= n4) {
        for (int64_t j = 0; j < step; j += 4) {         // 16 rows = 4 elements of each of the four sums
            Raw8<DT> r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = load8_raw<DT, false>(x, t + (4 * (e + j) + u) * groups);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 16; ++u) consume(r[u], acc[u & 3]);
        }
        e += step;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j].carry(e, lp, lmask);
    }
    for (; e < n4; ++e) {
        Raw8<DT> r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = load8_raw<DT, false>(x, t + (4 * e + k) * groups);
#pragma unroll
        for (int k = 0; k < 4; ++k) consume(r[k], acc[k]);
    }
    float p[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) p[k][j] = acc[k][j].total();
    for (int64_t i = n4 * 4; i < n; ++i) {
        float v[8];
        unpack8<DT>(load8_raw<DT, false>(x, t + i * groups), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[0][j] += mean_cl_prep<MODE>(v[j], amax[j], act, DT);
    }
    const float fn = (float)n;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t o = (c0 + j) * hw + pos;
        store1<ODT>(out, o, (((p[0][j] + p[1][j]) + p[2][j]) + p[3][j]) / fn);
        if (MODE != 3 && amax_part) amax_part[o] = amax[j];
    }
}

// ---- channels_last first stage for ANY channel count and every flag combination ----------------------------------
// One lane per (position, channel) of a sample, in memory order (so a wave's loads are contiguous), scalar accesses.
// Same summation rule as mean_cl_kernel / mean_cl_tail_kernel: positions below 4*floor(HW/4) in multi-row order, the
// rest in row-sum order (checked against ATen's CPU result for C = 3 ... 100, tests/test_gpu_parity.py).  Serves
// channels_last activations whose C is not a multiple of 8 and the L0 variant (sparse.py:85-86), which used to be
// copied to NCHW first -- and were then summed in NCHW order, i.e. not in the order the reference's CPU path uses.
template <int DT, int ODT, bool AMAX, int PREP>
__global__ __launch_bounds__(kBlock) void mean_cl_generic_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                                  int64_t n, int64_t hw, int64_t C, int flags,
                                                                  const int32_t* __restrict__ l0_flag,
                                                                  uint32_t* __restrict__ amax_part, ActSpec act) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= hw * C) return;
    const int l0 = (flags & QS_MEAN_L0) && l0_flag && *l0_flag;
    const int64_t pos = t / C, c = t - pos * C;
    const int64_t sample = hw * C;
    uint32_t amax = 0u;
    auto get = [&](int64_t i) {
        const float v = load1<DT>(x, i * sample + t);
        if constexpr (AMAX) {      // (compile time: see mean_generic_kernel)
            const float av = (PREP == 2) ? relu_aten(v) : (PREP == 1 || PREP == 3) ? v : ((flags & QS_MEAN_RELU) ? act_apply(v, act, DT) : v);
            const uint32_t k = __float_as_uint(av) & 0x7fffffffu;
            amax = k > amax ? k : amax;
        }
        return mean_prep_t<DT, PREP>(v, flags, l0, act);
    };
    const float s = (pos < (hw / 4) * 4) ? sum_multi_row(n, get) : sum_row_sum(n, get);
    const int64_t o = c * hw + pos;                       // the result is NCHW-contiguous, as ATen's is
    store1<ODT>(out, o, s / (float)n);
    if constexpr (AMAX) amax_part[o] = amax;
}

// ---- channels_last activation whose FIRST reduced dim is W (a mask that keeps N, C and H: `prune(dimensions={0, 1, 2})`) ------
// x: [N][H][W][C] in memory.  ATen's TensorIterator puts the reduced dim (W, C elements apart) innermost and H next; with
// in_stride[0] < in_stride[1] its sum kernel takes scalar_inner_sum (SumKernel.cpp), i.e. row_sum per output: four interleaved
// cascade sums over w (w feeds sum w % 4), the W % 4 last elements added to the first, then ((p0 + p1) + p2) + p3 -- NOT the
// vectorised inner sum it uses for the contiguous NCHW row (checked against Tensor.mean(3) on the CPU, tests/test_aten_contract.py).
// One lane per output (n, h, c), in memory order (a wave's loads are contiguous); the result is NCHW-contiguous [N][C][H] like
// ATen's.  Until ABI v20 such inputs were copied to NCHW and summed in THAT order (a last float32 bit).
template <int DT, int ODT>
__global__ __launch_bounds__(kBlock) void mean_cl_w_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t NH,
                                                            int64_t H, int64_t W, int64_t C, int flags,
                                                            const int32_t* __restrict__ l0_flag, ActSpec act) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // t = p * C + c,  p = n * H + h
    if (t >= NH * C) return;
    const int l0 = (flags & QS_MEAN_L0) && l0_flag && *l0_flag;
    const int64_t p = t / C, c = t - p * C;
    auto get = [&](int64_t i) { return mean_prep<DT>(load1<DT>(x, (p * W + i) * C + c), flags, l0, act); };
    const float s = sum_row_sum(W, get);
    const int64_t n = p / H, h = p - n * H;
    store1<ODT>(out, (n * C + c) * H + h, s / (float)W);
}

// ---- first stage of a dense tensor laid out in ANY dim order (transposed weights, permuted activations, NDHWC with the batch
// kept ...) ---------------------------------------------------------------------------------------------------------------------
// ATen reduces such a tensor where it lies; which of SumKernel.cpp's four loops it takes, and which of the kept coordinates fall in
// the cascade ("multi-row") part of an outer loop, follows from TensorIterator's dim reordering and coalescing -- host logic
// (qsparse_amd/util.py: `aten_reduce_plan`, pinned against Tensor.mean on the CPU in tests/test_aten_contract.py).  This kernel only
// executes the plan: one lane per output element; the lane index runs fastest over the kept dim with the smallest input stride, so a
// wave's loads are as contiguous as the layout allows.  order 0: the vectorised inner sum (8 interleaved row-sums + tail); 1: row-sum
// for every output; 2: cascade for outputs whose coordinate in kept dim `split_dim` is below `split`, row-sum for the others.
constexpr int kStridedMaxKept = 6;
struct StridedPlan {
    int64_t n, stride;                                   // the reduced dim
    int64_t size[kStridedMaxKept], in_stride[kStridedMaxKept], out_stride[kStridedMaxKept];
    int64_t split;
    int nkept, order, split_dim;
};

template <int DT, int ODT, int PREP>
__global__ __launch_bounds__(kBlock) void mean_strided_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t total,
                                                               StridedPlan p, int flags, const int32_t* __restrict__ l0_flag,
                                                               ActSpec act) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total) return;
    const int l0 = (flags & QS_MEAN_L0) && l0_flag && *l0_flag;
    int64_t rest = t, in0 = 0, o = 0, split_coord = 0;
    for (int k = 0; k < p.nkept; ++k) {
        const int64_t q = rest / p.size[k], c = rest - q * p.size[k];
        in0 += c * p.in_stride[k];
        o += c * p.out_stride[k];
        if (k == p.split_dim) split_coord = c;
        rest = q;
    }
    const int64_t n = p.n, st = p.stride;
    auto get = [&](int64_t i) { return mean_prep_t<DT, PREP>(load1<DT>(x, in0 + i * st), flags, l0, act); };
    float s;
    if (p.order == 0) {
        const int64_t nv = n / 8;
        float fin = 0.f;
        for (int64_t i = nv * 8; i < n; ++i) fin += get(i);         // the tail first, then the 8 vector lanes in turn
        for (int k = 0; k < 8; ++k) fin += sum_row_sum(nv, [&](int64_t i) { return get(8 * i + k); });
        s = fin;
    } else if (p.order == 2 && split_coord < p.split) {
        s = sum_multi_row(n, get);
    } else {
        s = sum_row_sum(n, get);
    }
    store1<ODT>(out, o, s / (float)n);
}

// ---- the same stage for tensors with FEW columns: rows split over R waves of one workgroup -------------------
// With C*H*W small (late ResNet stages, small batches of small maps) one wave per 512 columns leaves most CUs
// with one or two waves and the kernel becomes latency-bound.  Here the R waves of a workgroup own the SAME
// column groups and share the rows chunk-wise: wave w sums chunks w, w+R, ... (a chunk = 2^p consecutive rows,
// summed sequentially from zero -- exactly what ATen's level-0 accumulator holds when it is dumped into level
// 1), parks the chunk sums in LDS, and wave 0 then feeds them IN ORDER through the level-1..3 cascade, adds the
// n % 2^p tail rows and finishes.  Bit-identical to the single-wave kernel; needs n / 2^p <= kMaxSplitChunks.
constexpr int kMaxSplitChunks = 32;

// MODE 1 (|x|) / 2 (max(x, 0)) as in mean_outer_vec_kernel: mean operand == abs-max key; RAGGED: a lane's 8 columns may
// straddle two channels (chan_div % 8 != 0: 14x14, 7x7 maps), the key goes to the first or the second of them -- until
// the second half of round 2 such maps took the generic MODE 0 (8 VALU ops per element); MODE 0 handles every other
// flag combination.
template <int DT, int ODT, int R, int MODE, bool RAGGED = false>
__global__ __launch_bounds__(64 * R) void mean_outer_split_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                                   int64_t pre, int64_t n, int64_t post, int64_t vcols,
                                                                   int flags, const int32_t* __restrict__ l0_flag,
                                                                   uint32_t* __restrict__ absmax, int64_t astride,
                                                                   int64_t chan_div, uint32_t C, int lanes, ActSpec act) {
    extern __shared__ __attribute__((aligned(16))) float chunk_sums[];   // [nchunks][8][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gcols = vcols / 8, total = pre * gcols;
    const int64_t t = (int64_t)blockIdx.x * lanes + lane;
    const bool active = lane < lanes && t < total;
    const int l0 = (flags & QS_MEAN_L0) && l0_flag && *l0_flag;
    const int lp = max(4, ceil_log2_i64(n) / 4);
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    const int nchunks = (int)(n / step);
    const int64_t tt = active ? t : 0;
    const int64_t p = tt / gcols, gc = tt - p * gcols;
    const int64_t row_groups = post / 8;
    const int64_t g_base = p * n * row_groups + gc;
    uint32_t amax0 = 0u, amax1 = 0u;
    // channels of the first / last of this lane's 8 columns (they differ when chan_div is not a multiple of 8)
    const uint32_t c_first = absmax ? (uint32_t)(((gc * 8) / chan_div) % C) : 0u;
    const int64_t first_col_next = absmax ? ((gc * 8) / chan_div + 1) * chan_div - gc * 8 : 8;   // #cols in c_first

    auto track = [&](const float (&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float av = (flags & QS_MEAN_RELU) ? act_apply(v[j], act, DT) : v[j];
            const uint32_t k = __float_as_uint(av) & 0x7fffffffu;
            if (j < first_col_next) amax0 = k > amax0 ? k : amax0;
            else amax1 = k > amax1 ? k : amax1;
        }
    };

    auto consume = [&](const Raw8<DT>& r, auto add) {
        float v[8];
        unpack8<DT>(r, v);
        if constexpr (MODE == 1 || MODE == 2 || MODE == 7 || MODE == 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = (MODE == 2) ? relu_aten(v[k]) : (MODE == 7) ? act_apply_k<QS_ACT_HARDTANH>(v[k], act, DT)
                              : (MODE == 8) ? act_apply_k<QS_ACT_LEAKY>(v[k], act, DT) : v[k];
                const uint32_t key = __float_as_uint(w) & 0x7fffffffu;
                if (!RAGGED || k < first_col_next) amax0 = key > amax0 ? key : amax0;
                else amax1 = key > amax1 ? key : amax1;
                add(k, __uint_as_float(key));
            }
        } else {
            if (absmax) track(v);
#pragma unroll
            for (int k = 0; k < 8; ++k) add(k, mean_prep<DT>(v[k], flags, l0, act));
        }
    };

    if (active) {
        for (int ch = wave; ch < nchunks; ch += R) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            const int64_t r0 = (int64_t)ch * step;
            for (int64_t j = 0; j < step; j += 16) {
                Raw8<DT> r[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) r[u] = load8_raw<DT, QS_MEAN_SMALL_NT_LOADS>(x, g_base + (r0 + j + u) * row_groups);
#pragma unroll
                for (int u = 0; u < 16; ++u) consume(r[u], [&](int k, float val) { acc[k] += val; });
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) chunk_sums[(ch * 8 + j) * 64 + lane] = acc[j];
        }
    }
    __syncthreads();
    if (wave == 0 && active) {
        Cascade c[8];
        for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c[j].a0 = chunk_sums[(ch * 8 + j) * 64 + lane];   // level 0 as ATen holds it after this chunk
                c[j].carry((int64_t)(ch + 1) * step, lp, lmask);
            }
        }
        for (int64_t i = (int64_t)nchunks * step; i < n; ++i) {       // n % step tail rows, sequential into level 0
            consume(load8_raw<DT, false>(x, g_base + i * row_groups), [&](int k, float val) { c[k].add(val); });
        }
        float m[8];
        const float fn = (float)n;
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = c[j].total() / fn;
        store8<ODT, false>(out, p * row_groups + gc, m);
    }
    if (absmax) {   // one atomic per wave when the whole wave sits inside one channel, else per lane
        const uint32_t c0 = (uint32_t)__shfl((int)c_first, 0, 64);
        if (__all(!active || (c_first == c0 && first_col_next >= 8))) {
            const uint32_t m = wave_max_u32(active ? amax0 : 0u);
            if (lane == 0) atomicMax(absmax + (size_t)c0 * astride, m);
        } else if (active) {
            atomicMax(absmax + (size_t)c_first * astride, amax0);
            if (first_col_next < 8) atomicMax(absmax + (size_t)(c_first + 1 == C ? 0u : c_first + 1) * astride, amax1);
        }
    }
}

// ---- inner reduction of LONG rows (post == 1, n >= 64): half a wave per row ------------------------------------------------
// ATen's vectorised inner sum is 32 interleaved cascade sums: element e feeds the chain of vector lane e % 8 and row-sum part
// (e / 8) % 4 -- chain e % 32 -- at position e / 32, for the first 32 * floor(n / 32) elements.  Lane c of a half-wave runs chain c
// (its loads and its neighbours' are one contiguous 128-byte line of fp32); then, per vector lane k, the n / 8 % 4 leftover vectors
// go to part 0, the parts are added ((p0 + p1) + p2) + p3, and the row's first lane adds the n % 8 tail from zero and the 8 vector
// lanes in turn.  Same bits as mean_generic_kernel's one-lane-per-row loop, which read a 4096-long row at 48 GB/s.
template <int DT, int ODT>
__global__ __launch_bounds__(64) void mean_inner_wave_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t rows,
                                                              int64_t n, int flags, const int32_t* __restrict__ l0_flag,
                                                              ActSpec act) {
    const int lane = threadIdx.x & 31, half = threadIdx.x >> 5;
    const int64_t row = (int64_t)blockIdx.x * 2 + half;
    const bool live = row < rows;
    const int l0 = (flags & QS_MEAN_L0) && l0_flag && *l0_flag;
    const int64_t base = (live ? row : 0) * n;          // (an odd row count: the idle half re-reads row 0, uniform control flow)
    auto get = [&](int64_t e) { return mean_prep<DT>(load1<DT>(x, base + e), flags, l0, act); };
    const int64_t nv = n / 8, n4 = nv / 4;
    const float chain = sum_multi_row(n4, [&](int64_t m) { return get(32 * m + lane); });
    const int k0 = half * 32 + (lane & 7);
    float p0 = __shfl(chain, k0, 64);
    const float p1 = __shfl(chain, k0 + 8, 64), p2 = __shfl(chain, k0 + 16, 64), p3 = __shfl(chain, k0 + 24, 64);
    if (lane < 8)
        for (int64_t i = n4 * 4; i < nv; ++i) p0 += get(8 * i + lane);
    const float vec_lane = ((p0 + p1) + p2) + p3;
    float fin = 0.f;
    if (lane == 0)
        for (int64_t i = nv * 8; i < n; ++i) fin += get(i);
#pragma unroll
    for (int k = 0; k < 8; ++k) fin += __shfl(vec_lane, half * 32 + k, 64);
    if (lane == 0 && live) store1<ODT>(out, row, fin / (float)n);
}

// ---- generic stage: one thread per output element, either order -------------------------------------
// AMAX (compile time): with the abs-max rider.  A RUNTIME test of the rider's pointer inside `get` puts control flow between the
// sixteen loads sum_multi_row requests ahead of its ordered adds and serialises them: 74 us instead of 25 on 256 x 150528 fp32
template <int DT, int ODT, bool AMAX, int PREP>
__global__ __launch_bounds__(kBlock) void mean_generic_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                               int64_t pre, int64_t n, int64_t post, int64_t col0,
                                                               int flags, const int32_t* __restrict__ l0_flag,
                                                               uint32_t* __restrict__ absmax, int64_t astride,
                                                               int64_t chan_div, uint32_t C, ActSpec act, int64_t mr_override) {
    const int64_t ncols = post - col0;
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= pre * ncols) return;
    const int l0 = (flags & QS_MEAN_L0) && l0_flag && *l0_flag;
    const int64_t p = t / ncols, col = col0 + (t - p * ncols);
    const int64_t base = p * n * post + col;
    uint32_t amax = 0u;
    auto get = [&](int64_t i) {
        const float v = load1<DT>(x, base + i * post);
        if constexpr (AMAX) {
            const float av = (PREP == 2) ? relu_aten(v) : (PREP == 1 || PREP == 3) ? v : ((flags & QS_MEAN_RELU) ? act_apply(v, act, DT) : v);
            const uint32_t k = __float_as_uint(av) & 0x7fffffffu;
            amax = k > amax ? k : amax;
        }
        return mean_prep_t<DT, PREP>(v, flags, l0, act);
    };
    float s;
    if (post == 1) {
        if (n >= 8) {  // vectorized inner sum: 8 lanes, row-sum order over n/8 vectors
            const int64_t nv = n / 8;
            float lanes[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) lanes[k] = sum_row_sum(nv, [&](int64_t i) { return get(8 * i + k); });
            float fin = 0.f;
            for (int64_t i = nv * 8; i < n; ++i) fin += get(i);
#pragma unroll
            for (int k = 0; k < 8; ++k) fin += lanes[k];
            s = fin;
        } else {
            s = sum_row_sum(n, get);
        }
    } else {
        // (mr_override >= 0: the caller names the cascade prefix -- qs_mean_dim_split, a permuted tensor's memory view)
        const int64_t mr_cols = mr_override >= 0 ? mr_override : ((post >= 8) ? (post / 32) * 32 : (post / 4) * 4);
        s = (col < mr_cols) ? sum_multi_row(n, get) : sum_row_sum(n, get);
    }
    store1<ODT>(out, p * post + col, s / (float)n);
    if constexpr (AMAX) atomicMax(absmax + (size_t)(uint32_t)((col / chan_div) % C) * astride, amax);
}

// =================================================================================================
// k-th order statistic by 4-pass radix select on order-preserving keys.
// =================================================================================================
struct SelectState {
    uint32_t prefix;   // key bits fixed so far
    uint32_t k;        // rank still to find inside the surviving set
    uint32_t hist[256];
};

// multi-block pass: histogram of byte `pass` (3 = most significant) among keys matching the prefix
static __global__ __launch_bounds__(kBlock) void select_hist_kernel(const float* __restrict__ imp, int64_t n, int pass,
                                                              SelectState* st) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = st->prefix;
    const int shift = 8 * pass;
    const uint32_t himask = (pass == 3) ? 0u : (0xffffffffu << (shift + 8));
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t k = f32_to_key(imp[i]);
        if ((k & himask) == (prefix & himask)) atomicAdd(&h[(k >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}
// single-thread-block scan: pick the bin holding rank k, descend
static __global__ void select_scan_kernel(SelectState* st, int pass, float* thr) {
    if (threadIdx.x == 0) {
        uint32_t k = st->k, cum = 0;
        int b = 0;
        for (; b < 256; ++b) {
            const uint32_t c = st->hist[b];
            if (cum + c > k) break;
            cum += c;
        }
        if (b == 256) b = 255;
        st->k = k - cum;
        st->prefix |= ((uint32_t)b) << (8 * pass);
        if (pass == 0 && thr) *thr = key_to_f32(st->prefix);
    }
    __syncthreads();
    st->hist[threadIdx.x] = 0;   // blockDim.x == 256
}
static __global__ void select_init_kernel(SelectState* st, uint32_t k) {
    st->hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        st->prefix = 0;
        st->k = k;
    }
}

// in-block selection of the key of rank k (0-based, ascending) among v[0..n):
//   n <= kRankMax: every thread ranks one element against all others held in LDS (ties broken by index, so
//                  ranks are a permutation) -- ~n broadcast LDS reads, no atomics;
//   larger n:      4-pass radix select, bins scanned in parallel (one bin per thread, wave prefix sums).
constexpr int kSelectThreads = 1024;
constexpr int kRankMax = 2048;
// pq_select holds up to this many channels in registers + LDS (round 6: 2048 -> 8192, the hidden widths of transformer blocks --
// a token-major site with C = 3072 took the global-memory passes: 22.8 us)
constexpr int kSelectLdsMax = 8192;

struct SelectShared {
    alignas(16) uint32_t keys[kSelectLdsMax];
    uint32_t hist[256];
    uint32_t wsum[4];
    uint32_t state[2];
};

// rank of (key `mine` at index i) among keys[0..npad): #smaller + #equal-with-lower-index.  npad is n rounded up
// to a multiple of 4 with the padding slots holding 0xffffffff; 16-byte LDS reads, 4 keys per read.
__device__ __forceinline__ uint32_t rank_of(const uint32_t* keys, int npad, uint32_t mine, int i) {
    uint32_t rank = 0;
#pragma unroll 4
    for (int j = 0; j < npad; j += 4) {
        const u32x4 kk = *(const u32x4*)(keys + j);
#pragma unroll
        for (int u = 0; u < 4; ++u) rank += (kk[u] < mine || (kk[u] == mine && j + u < i)) ? 1u : 0u;
    }
    return rank;
}

// 4-pass radix select (8-bit digits, most significant first) over keys produced by `key_at(i)`, i in [0, n).
// Bins are scanned in parallel: one bin per thread, wave prefix sums, three barriers per pass.  sh.hist must be
// zero on entry for the first pass; every pass leaves it zero again.
template <typename KeyAt>
__device__ __forceinline__ uint32_t radix_select(KeyAt key_at, int64_t n, uint32_t k, SelectShared& sh) {
    const int tid = threadIdx.x, nthreads = blockDim.x;   // nthreads >= 256
    if (tid < 256) sh.hist[tid] = 0;
    __syncthreads();
    uint32_t prefix = 0;
    for (int pass = 3; pass >= 0; --pass) {
        const int shift = 8 * pass;
        const uint32_t himask = (pass == 3) ? 0u : (0xffffffffu << (shift + 8));
        for (int64_t i = tid; i < n; i += nthreads) {
            const uint32_t key = key_at(i);
            if ((key & himask) == (prefix & himask)) atomicAdd(&sh.hist[(key >> shift) & 0xff], 1u);
        }
        __syncthreads();
        uint32_t c = 0, incl = 0;
        if (tid < 256) {   // inclusive prefix sum over the 256 bins: 4 waves x 64 lanes
            c = sh.hist[tid];
            sh.hist[tid] = 0;
            incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
                if ((tid & 63) >= off) incl += o;
            }
            if ((tid & 63) == 63) sh.wsum[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            for (int w = 0; w < (tid >> 6); ++w) incl += sh.wsum[w];
            const uint32_t excl = incl - c;
            if (c > 0 && excl <= k && k < incl) {
                sh.state[0] = prefix | (((uint32_t)tid) << shift);
                sh.state[1] = k - excl;
            }
        }
        __syncthreads();
        prefix = sh.state[0];   // rewritten only after two more barriers
        k = sh.state[1];
    }
    __syncthreads();
    return prefix;
}

// keys already in sh.keys[0..n), n <= kRankMax: rank counting up to kRankSmall keys (one LDS sweep per key,
// O(n^2) compares on one CU), radix select above.  Measured per call of the whole select step, rank vs radix:
// C=64 6.4 vs 7.7 us, 256 10.6 vs 7.4, 512 16.9 vs 8.2, 1024 41.6 vs 9.0, 2048 150 vs 9.6.
constexpr int kRankSmall = 128;
__device__ __forceinline__ uint32_t lds_select_key(SelectShared& sh, int n, uint32_t k, int rank_small) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    if (n <= rank_small) {
        const int npad = (n + 3) & ~3;
        for (int i = tid; i < n; i += nthreads) {
            const uint32_t mine = sh.keys[i];
            if (rank_of(sh.keys, npad, mine, i) == k) sh.state[0] = mine;
        }
        __syncthreads();
        const uint32_t r = sh.state[0];
        __syncthreads();
        return r;
    }
    return radix_select([&](int64_t i) { return sh.keys[i]; }, n, k, sh);
}

__device__ __forceinline__ uint32_t block_select_key(const float* v, int64_t n, uint32_t k, SelectShared& sh,
                                                     int rank_small = kRankSmall) {
    const int tid = threadIdx.x, nthreads = blockDim.x;   // nthreads >= 256
    if (n <= kRankMax) {
        const int npad = ((int)n + 3) & ~3;
        for (int i = tid; i < npad; i += nthreads) sh.keys[i] = i < (int)n ? f32_to_key(v[i]) : 0xffffffffu;
        __syncthreads();
        return lds_select_key(sh, (int)n, k, rank_small);
    }
    return radix_select([&](int64_t i) { return f32_to_key(v[i]); }, n, k, sh);
}

static __global__ __launch_bounds__(kSelectThreads) void kth_small_kernel(const float* __restrict__ imp, int64_t n, uint32_t k,
                                                                    float* thr) {
    __shared__ SelectShared sh;
    const uint32_t key = block_select_key(imp, n, k, sh);
    if (threadIdx.x == 0) *thr = key_to_f32(key);
}

static __global__ __launch_bounds__(kBlock) void mask_ge_kernel(const float* __restrict__ imp, const float* __restrict__ thr,
                                                          uint8_t* __restrict__ mask, int64_t n) {
    const float t = *thr;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) mask[i] = imp[i] >= t ? 1 : 0;
}

// =================================================================================================
// fused C-sized step of the channel-prune -> tensor-wise-quantize pair (see qsparse_hip.h)
// =================================================================================================
struct PqArgs {
    float* magnitude;
    int64_t C;
    int update_magnitude;
    float t_mag, t_mag1;
    int refresh_mask;
    uint32_t k;
    uint8_t* mask;
    uint32_t* chan_absmax;
    int64_t amax_stride;        // elements between consecutive channels of chan_absmax
    uint8_t* elide_mask;        // nullable, with update_scale: [C] bytes for the forward's load elision -- 1 kept, 0 pruned and every
                                // value of the channel finite this step, 2 pruned with a NaN / Inf (see keep_from_byte)
    int update_scale;
    float t_q, t_q1, denom;
    float* scale;
    int32_t* bump_a;
    int32_t* bump_b;
    int64_t* bump_c;
    int64_t* bump_d;
    const int64_t* t_mag_dev;   // nullable device counters overriding t_mag / t_q (graph replay)
    const int64_t* t_q_dev;
    int rank_small;             // rank counting up to this many channels, radix select above (kRankSmall)
    int stat_dt;                // dtype of the activation the abs-max was taken from (the quotient is rounded to it)
    // multi-rank run: the all-gathered [world][2C] records of qs_stats_pack (importance | abs-max) are combined here,
    // in rank order, instead of reading `stage` / `chan_absmax` (which is then only re-zeroed)
    const float* gathered;
    int world;
};

// channel i's importance / abs-max key: the local statistics, or the rank-ordered combination of every rank's record
// (same arithmetic as stats_combine_kernel: fp32 sum in rank order divided by the world size; maximum of the keys)
template <int SDT>
__device__ __forceinline__ float pq_stage_value(const PqArgs& a, const void* stage, int64_t i) {
    if (!a.gathered) return load1<SDT>(stage, i);
    float sum = 0.f;
    for (int r = 0; r < a.world; ++r) sum += a.gathered[(int64_t)r * 2 * a.C + i];
    return sum / (float)a.world;
}
__device__ __forceinline__ uint32_t pq_amax_key(const PqArgs& a, int64_t i) {
    if (!a.gathered) return a.chan_absmax[i * a.amax_stride];
    uint32_t mx = 0u;
    for (int r = 0; r < a.world; ++r) {
        const uint32_t k = __float_as_uint(a.gathered[(int64_t)r * 2 * a.C + a.C + i]);
        mx = k > mx ? k : mx;
    }
    return mx;
}

// what a channel contributes to max|x * mask| (quantize.py:340 on the product of sparse.py:263), given its abs-max key of |x|:
// a kept channel its own maximum; a pruned one 0 -- unless it holds a NaN / Inf, whose product with the mask's 0 is a NaN that
// the reference's x.abs().max() propagates into the scale (an Inf gives the default NaN, a NaN keeps its payload)
__device__ __forceinline__ uint32_t pq_scale_key(uint32_t am, bool kept) {
    if (kept) return am;
    return am > 0x7f800000u ? am : (am == 0x7f800000u ? 0x7fc00000u : 0u);
}

__device__ __forceinline__ PqArgs pq_live_counters(PqArgs a) {
    if (a.t_mag_dev) {
        a.t_mag = (float)*a.t_mag_dev;
        a.t_mag1 = (float)(*a.t_mag_dev + 1);
    }
    if (a.t_q_dev) {
        a.t_q = (float)*a.t_q_dev;
        a.t_q1 = (float)(*a.t_q_dev + 1);
    }
    return a;
}

// the C-sized step, run by ONE workgroup of >= 256 threads; `stage` holds the last squeeze stage ([C], dtype SDT)
template <int SDT>
__device__ __forceinline__ void pq_select_body(const PqArgs& a, const void* stage, SelectShared& sh, uint32_t* sh_max) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    if (a.update_magnitude) {
        for (int64_t i = tid; i < a.C; i += nthreads)
            a.magnitude[i] = (a.t_mag * a.magnitude[i] + pq_stage_value<SDT>(a, stage, i)) / a.t_mag1;   // sparse.py:89
        __threadfence_block();
        __syncthreads();
    }
    if (a.refresh_mask) {
        const uint32_t key = block_select_key(a.magnitude, a.C, a.k, sh, a.rank_small);
        const float thr = key_to_f32(key);
        for (int64_t i = tid; i < a.C; i += nthreads) a.mask[i] = a.magnitude[i] >= thr ? 1 : 0;   // util.py:117
        __threadfence_block();
        __syncthreads();
    }
    if (a.update_scale) {
        uint32_t m = 0u;
        for (int64_t i = tid; i < a.C; i += nthreads) {
            const uint32_t am = pq_amax_key(a, i);
            const bool kept = a.mask[i] != 0;
            const uint32_t sk = pq_scale_key(am, kept);
            m = sk > m ? sk : m;
            if (a.elide_mask) a.elide_mask[i] = kept ? 1 : (am >= 0x7f800000u ? 2 : 0);
            if (a.chan_absmax) a.chan_absmax[i * a.amax_stride] = 0u;   // leave the accumulator clean for the next statistics pass
        }
        m = wave_max_u32(m);
        if ((tid & 63) == 0) sh_max[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < nthreads / 64; ++i) m = sh_max[i] > m ? sh_max[i] : m;
            const float nw = round_to_dtype(__uint_as_float(m) / a.denom, a.stat_dt);           // quantize.py:340
            a.scale[0] = (a.t_q == 0.0f) ? nw : (a.t_q * a.scale[0] + nw) / a.t_q1;             // :344-347
        }
    }
    if (tid == 0) {   // step counters of the two layers / the callback (state_dict tensors)
        // fire-and-forget atomics: a plain `*p += 1` is a dependent load -> store round trip per counter
        if (a.bump_a) atomicAdd(a.bump_a, 1);
        if (a.bump_b) atomicAdd(a.bump_b, 1);
        if (a.bump_c) atomicAdd((unsigned long long*)a.bump_c, 1ull);
        if (a.bump_d) atomicAdd((unsigned long long*)a.bump_d, 1ull);
    }
}

// C <= kRankMax: the whole select step with every per-channel value held in registers / LDS (one global
// round trip in, one out).  Same arithmetic and order as pq_select_body.
template <int SDT, int ITEMS>
__device__ __forceinline__ void pq_select_small(const PqArgs& a, const void* stage, SelectShared& sh, uint32_t* sh_max) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const float scale_old = a.update_scale ? a.scale[0] : 0.f;   // fetched with the first round trip, not as one of its own at the end
    float mag[ITEMS];
    uint32_t amax[ITEMS];
    uint8_t keep[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int i = tid + it * nthreads;
        mag[it] = 0.f;
        amax[it] = 0u;
        keep[it] = 0;
        if (i < a.C) {
            mag[it] = a.magnitude[i];
            if (a.update_magnitude) mag[it] = (a.t_mag * mag[it] + pq_stage_value<SDT>(a, stage, i)) / a.t_mag1;   // sparse.py:89
            if (a.update_scale) amax[it] = pq_amax_key(a, i);
            keep[it] = a.mask[i];
            sh.keys[i] = f32_to_key(mag[it]);
        } else if (i < ITEMS * nthreads) {
            sh.keys[i] = 0xffffffffu;   // padding for the 4-wide rank loop
        }
    }
    __syncthreads();
    if (a.refresh_mask) {
        const float thr = key_to_f32(lds_select_key(sh, (int)a.C, a.k, a.rank_small));
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) keep[it] = mag[it] >= thr ? 1 : 0;      // util.py:117
    }
    uint32_t m = 0u;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int i = tid + it * nthreads;
        if (i < a.C) {
            if (a.update_magnitude) a.magnitude[i] = mag[it];
            if (a.refresh_mask) a.mask[i] = keep[it];
            if (a.update_scale) {
                const uint32_t sk = pq_scale_key(amax[it], keep[it] != 0);
                m = sk > m ? sk : m;
                if (a.elide_mask) a.elide_mask[i] = keep[it] ? 1 : (amax[it] >= 0x7f800000u ? 2 : 0);
                if (a.chan_absmax) a.chan_absmax[i * a.amax_stride] = 0u;
            }
        }
    }
    if (a.update_scale) {
        m = wave_max_u32(m);
        if ((tid & 63) == 0) sh_max[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < nthreads / 64; ++i) m = sh_max[i] > m ? sh_max[i] : m;
            const float nw = round_to_dtype(__uint_as_float(m) / a.denom, a.stat_dt);
            a.scale[0] = (a.t_q == 0.0f) ? nw : (a.t_q * scale_old + nw) / a.t_q1;
        }
    }
    if (tid == 0) {
        // fire-and-forget atomics: a plain `*p += 1` is a dependent load -> store round trip per counter
        if (a.bump_a) atomicAdd(a.bump_a, 1);
        if (a.bump_b) atomicAdd(a.bump_b, 1);
        if (a.bump_c) atomicAdd((unsigned long long*)a.bump_c, 1ull);
        if (a.bump_d) atomicAdd((unsigned long long*)a.bump_d, 1ull);
    }
}

// THREADS = 256 for C <= 256 (4 waves synchronise faster than 16), 1024 otherwise
template <int SDT, int THREADS>
__global__ __launch_bounds__(THREADS) void pq_select_kernel(PqArgs a0, const void* __restrict__ stage) {
    const PqArgs a = pq_live_counters(a0);   // every thread reads the counters before thread 0 bumps them (barriers in between)
    __shared__ SelectShared sh;
    __shared__ uint32_t sh_max[THREADS / 64];
    static_assert(THREADS == 256 || (kRankMax % THREADS == 0 && kSelectLdsMax == 8 * THREADS), "items per thread must cover the LDS-resident sizes");
    if constexpr (THREADS == 256) {
        pq_select_small<SDT, 1>(a, stage, sh, sh_max);                       // host guarantees C <= 256
    } else {
        if (a.C <= kRankMax) pq_select_small<SDT, kRankMax / THREADS>(a, stage, sh, sh_max);
        else if (a.C <= 4 * THREADS) pq_select_small<SDT, 4>(a, stage, sh, sh_max);
        else if (a.C <= kSelectLdsMax) pq_select_small<SDT, 8>(a, stage, sh, sh_max);
        else pq_select_body<SDT>(a, stage, sh, sh_max);
    }
}

// =================================================================================================
// Last two stages of the staged mean fused: x [pre, H, W] -> mean over H (rounded to DT) -> mean over W
// (rounded to ODT) -> out [pre].  One workgroup per `pre` slice, tile held in LDS as fp32; both stages
// add in ATen's order (outer rule over H, inner rule over W; see the block comment above).
// =================================================================================================
template <int DT, int ODT>
__global__ __launch_bounds__(kBlock) void mean_last2_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                             int H, int W, const uint32_t* __restrict__ amax_part,
                                                             uint32_t* __restrict__ chan_absmax, int64_t astride,
                                                             float* __restrict__ record) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // H*W + W floats
    float* colmean = tile + (size_t)H * W;
    const int64_t p = blockIdx.x;
    const int hw = H * W;
    if (amax_part) {   // per-element maxima left by mean_cl_kernel -> per-channel abs-max (this workgroup owns channel p)
        __shared__ uint32_t wmax[kBlock / 64];
        uint32_t m = 0u;
        if ((hw & 3) == 0 && (((uintptr_t)amax_part) & 15) == 0) {
            // 16-byte loads, four in flight per thread: 56 x 56 maps took thirteen dependent 4-byte round trips here
            const uint4* part4 = reinterpret_cast<const uint4*>(amax_part + p * hw);
            const int n4 = hw >> 2;
            for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * kBlock) {
                uint4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = (i0 + u * kBlock < n4) ? part4[i0 + u * kBlock] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t a = q[u].x > q[u].y ? q[u].x : q[u].y, b = q[u].z > q[u].w ? q[u].z : q[u].w;
                    const uint32_t k = a > b ? a : b;
                    m = k > m ? k : m;
                }
            }
        } else {
            for (int i = threadIdx.x; i < hw; i += kBlock) {
                const uint32_t k = amax_part[p * hw + i];
                m = k > m ? k : m;
            }
        }
        m = wave_max_u32(m);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kBlock / 64; ++w) m = wmax[w] > m ? wmax[w] : m;
            const uint32_t old = chan_absmax[p * astride];
            chan_absmax[p * astride] = m > old ? m : old;   // max-accumulate, single writer per channel
        }
    }
    load_tile_f32<DT>(x, p, hw, tile);
    __syncthreads();
    const int mr_cols = (W >= 8) ? (W / 32) * 32 : (W / 4) * 4;
    for (int col = threadIdx.x; col < W; col += kBlock) {
        auto get = [&](int64_t i) { return tile[i * W + col]; };
        const float s = (col < mr_cols) ? sum_multi_row(H, get) : sum_row_sum(H, get);
        colmean[col] = round_through<DT>(s / (float)H);
    }
    __syncthreads();
    const float s = inner_sum_lds(colmean, W, colmean + W);
    if (threadIdx.x == 0) {
        const float mean = s / (float)W;
        store1<ODT>(out, p, mean);
        if (record) {   // this rank's exchange record (qs_stats_pack's layout): importance | abs-max, as float32
            record[p] = round_through<ODT>(mean);
            record[gridDim.x + p] = chan_absmax ? __uint_as_float(chan_absmax[p * astride]) : 0.f;
        }
    }
}

// =================================================================================================
// data-parallel statistics exchange: pack (importance, abs-max) into one fp32 record per rank, all-gather,
// combine in rank order (mean of the importances, max of the abs-max) -- one collective per step
// =================================================================================================
template <int SDT>
__global__ void stats_pack_kernel(const void* stage, const uint32_t* absmax, int64_t astride, int64_t C, float* rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {
        rec[i] = stage ? load1<SDT>(stage, i) : 0.f;
        rec[C + i] = absmax ? __uint_as_float(absmax[i * astride]) : 0.f;
    }
}
static __global__ void stats_combine_kernel(const float* gathered, int world, int64_t C, float* stage_out, uint32_t* absmax_out,
                                     int64_t astride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {
        float sum = 0.f;
        uint32_t mx = 0u;
        for (int r = 0; r < world; ++r) {        // fixed rank order: every rank computes the same bits
            sum += gathered[(int64_t)r * 2 * C + i];
            const uint32_t a = __float_as_uint(gathered[(int64_t)r * 2 * C + C + i]);
            mx = a > mx ? a : mx;
        }
        if (stage_out) stage_out[i] = sum / (float)world;
        if (absmax_out) absmax_out[i * astride] = mx;
    }
}

}  // namespace qs
