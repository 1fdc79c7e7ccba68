// Host-side helpers shared by the translation units of libqsparse_hip.so (api_*.hip): argument checks, dtype dispatch,
// launch geometry knobs.  Everything here has internal linkage; each TU compiles in parallel (see __graft_entry__.build_hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "qs_common.h"

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

// folded activations: `pre_relu` arguments of the ABI are 0, 1 (nn.ReLU) or a qs_activation() handle; defined in api_core.hip
// (one table for every translation unit).  Returns QS_OK and the descriptor, or QS_ERR_ARG for an unknown handle.
int qs_act_resolve(int pre_relu, ActSpec* out);
// the STE backward with the caller's activation's backward (qs_ste_relu_bwd_args::act_x, v26); defined in api_quant_bwd_act.hip
int qs_ste_act_bwd_impl(const qs_ste_relu_bwd_args& a);

// statistics flags (QS_MEAN_*) -> the folded activation's descriptor and the flags' low byte.  The IDENTITY -- nn.LeakyReLU(1.0),
// `x > 0 ? x : x * 1.0f`, what a site without a foldable activation folds to obtain its autocast image -- is dropped here: |x * 1| is
// |x| in every bit, so such a site's statistics take the kernels of a site with no activation (the vector kernels' compile-time
// modes and qs_token_stats' per-column form; through the run-time leaky mode a 128 x 196 x 3072 bf16 site took 197 us instead of 33)
static inline int mean_act_resolve(int* flags, ActSpec* act) {
    const int f = *flags;
    if (qs_act_resolve((f & QS_MEAN_RELU) ? std::max(f >> 8, 1) : 0, act) != QS_OK) return QS_ERR_ARG;
    *flags = f & 0xff;
    if (act->kind == QS_ACT_LEAKY && act->a == 1.0f) {
        act->kind = QS_ACT_NONE;
        act->a = act->b = 0.f;
        *flags &= ~QS_MEAN_RELU;
    }
    return QS_OK;
}

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
inline int grid_for(int64_t ngroups, int unroll) {
    int64_t b = (ngroups + (int64_t)kBlock * unroll - 1) / ((int64_t)kBlock * unroll);
    if (b < 1) b = 1;
    if (b > max_blocks()) b = max_blocks();
    return (int)b;
}

inline bool aligned8(const void* p) { return (((uintptr_t)p) & 7u) == 0; }

// Descriptor entry points (qs_*_v, include/qsparse_hip.h): the caller's struct -- possibly shorter (compiled against an older
// header) or longer (a newer one) than this library's -- into a zero-initialised local one.  Fields are only ever appended, so
// what the caller did not know about reads as 0 / NULL and what this library does not know about is ignored.
template <class T>
inline bool take_args(const T* in, T* out) {
    if (!in || in->struct_size < sizeof(uint32_t)) return false;
    *out = T{};
    memcpy(out, in, in->struct_size < sizeof(T) ? in->struct_size : sizeof(T));
    return true;
}

}  // namespace
