// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), statistics: abs-max / min-max, running scale and lines, L0 flag, running mean (qs_reduce.h).
// Host side: argument checks, geometry, launch configuration.  No allocation, no synchronisation: every entry
// point only enqueues work on the caller's stream.
#include "qs_host.h"
#include "qs_reduce.h"

extern "C" {

// ------------------------------------------------------------------------------------------------
static int reduce_impl(const void* x, float* out_a, float* out_b, bool minmax, int per_channel, int64_t outer, int64_t C,
                       int64_t inner, int xdt, hipStream_t s, bool accumulate = false, ActSpec relu = ActSpec{0, 0.f, 0.f}, void* ws = nullptr,
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
        const bool rows_route = !minmax && inner % 8 == 0 && outer >= 16 && cols / 8 >= 64 * 1024 && relu.kind <= QS_ACT_RELU;   // column walk
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
                } else if (vec_ptr && inner % 8 == 0 && outer >= 16 && (C * inner) / 8 >= 64 * 1024 && relu.kind <= QS_ACT_RELU) {
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
                    else if (relu.kind)
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
    ActSpec act;
    if (qs_act_resolve(pre_relu, &act) != QS_OK) return QS_ERR_ARG;
    if (act.kind == QS_ACT_LEAKY && act.a == 1.0f) act = ActSpec{0, 0.f, 0.f};      // a folded identity: |x * 1| is |x| (mean_act_resolve)
    return reduce_impl(x, out, nullptr, false, per_channel, outer, C, inner, xdt, (hipStream_t)stream, accumulate != 0, act, ws,
                       ws_bytes, out_lines);
}

int qs_minmax(const void* x, float* out_min, float* out_max, int per_channel, int64_t outer, int64_t C, int64_t inner,
              int xdt, int accumulate, void* ws, size_t ws_bytes, qs_stream_t stream) {
    return reduce_impl(x, out_min, out_max, true, per_channel, outer, C, inner, xdt, (hipStream_t)stream, accumulate != 0,
                       ActSpec{0, 0.f, 0.f}, ws, ws_bytes);
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

}  // extern "C"
