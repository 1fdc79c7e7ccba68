// libqsparse_hip.so -- C ABI (include/qsparse_hip.h), the MAILBOX form of the data-parallel statistics exchange (a prototype, off
// by default: set_qsparse_options(sync_statistics="mailbox")).  The collective form is one all-gather of a 2C-float record per
// site and step (distributed.py): 2 KB, latency-bound, and ~40 us of host time per call that a host-bound network pays sixteen
// times a step.  Here every rank owns a mailbox in its own HBM that its peers have mapped through hipIpc*: a rank's statistics
// kernels are followed by ONE launch that stores its record into every peer's mailbox (one-sided stores over xGMI) and then raises
// a flag there with system-scope release; ONE more launch waits, with a bounded spin, until the flags of all ranks show this step,
// and the select kernel then combines the records in rank order from local memory exactly as it combines the all-gather's result.
// No host collective, nothing for the host to wait for.
//
// Mailbox layout (uint32 words): flags [2 parities][world] in the first kMailboxHeader words, then records [2 parities][world][n]
// floats.  Steps alternate between the two halves; a peer cannot be more than one step ahead of the slowest rank (its next publish
// is ordered behind its own select, which needs everybody's flag), so the half being read is never the half being written.
#include <cstring>

#include "qs_host.h"

namespace qs {

constexpr int kMailboxHeader = 64;      // uint32 words reserved for the flags: 2 * world <= 64
constexpr int kMailboxMaxWorld = 32;

struct MailboxPeers {
    uint32_t* box[kMailboxMaxWorld];
};

static __global__ __launch_bounds__(256) void mailbox_publish_kernel(const float* __restrict__ rec, int64_t n, MailboxPeers peers,
                                                                    int world, int rank, uint32_t step) {
    uint32_t* box = peers.box[blockIdx.x];
    const int parity = (int)(step & 1u);
    float* dst = reinterpret_cast<float*>(box + kMailboxHeader) + ((size_t)parity * world + rank) * n;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = rec[i];
    __threadfence_system();             // every thread's stores are visible system-wide before the flag goes up
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(box + parity * world + rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one thread per rank; status (device int32, 0 on entry) receives 1 + the first rank whose flag did not arrive within max_spins.
// A timeout is FATAL FOR THE STEP: the missing rank's record of this step is overwritten with NaNs before the select reads it, so
// the masks and scales of a rank that missed a peer cannot silently drift on a stale record (the NaN reaches the running magnitude
// and the scale, i.e. the output, of this very step); the host side raises as soon as it sees the status word
// (distributed.py `_StatusWatch`, polled without a sync at every exchange).
static __global__ void mailbox_wait_kernel(uint32_t* box, int world, int64_t n, uint32_t step, int32_t* status, uint32_t max_spins) {
    const int r = threadIdx.x;
    if (r >= world) return;
    const int parity = (int)(step & 1u);
    uint32_t seen, spins = 0;
    do {
        seen = __hip_atomic_load(box + parity * world + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(seen - step) >= 0) break;
        __builtin_amdgcn_s_sleep(8);
    } while (++spins < max_spins);
    if ((int32_t)(seen - step) < 0) {
        uint32_t* rec = box + kMailboxHeader + ((size_t)parity * world + r) * n;
        for (int64_t i = 0; i < n; ++i) rec[i] = 0x7fc00000u;
        atomicCAS((int*)status, 0, 1 + r);
    }
}

// out[i] <- max over ranks of records[r][i], compared as uint32 keys (non-negative floats keep their order, a NaN stays the
// maximum): the all-reduce (MAX) of a quantize-only site's abs-max accumulator lines, from the mailbox
static __global__ void records_max_kernel(const uint32_t* __restrict__ rec, int world, int64_t n, uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t mx = 0u;
        for (int r = 0; r < world; ++r) {
            const uint32_t k = rec[(int64_t)r * n + i];
            mx = k > mx ? k : mx;
        }
        out[i] = mx;
    }
}

}  // namespace qs

using namespace qs;

extern "C" {

size_t qs_mailbox_bytes(int world, int64_t n) {
    if (world < 1 || world > kMailboxMaxWorld || n < 1) return 0;
    return ((size_t)kMailboxHeader + (size_t)2 * world * n) * sizeof(uint32_t);
}

int qs_mailbox_alloc(size_t bytes, void** ptr) {
    if (!ptr || bytes == 0) return QS_ERR_ARG;
    // fine-grained, or not at all: a peer's stores and flags must not be hidden from this device's acquire loads by its own L2,
    // which coarse-grained memory does not guarantee -- a caller that gets an error here keeps the collective exchange
    hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return hip_status(e);
    }
    return hip_status(hipMemset(*ptr, 0, bytes));
}

int qs_mailbox_free(void* ptr) { return ptr ? hip_status(hipFree(ptr)) : QS_OK; }

int qs_mailbox_export(void* ptr, void* handle64) {
    if (!ptr || !handle64) return QS_ERR_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the binding ships the handle as 64 bytes");
    return hip_status(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr));
}

int qs_mailbox_open(const void* handle64, void** ptr) {
    if (!ptr || !handle64) return QS_ERR_ARG;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof(h));
    return hip_status(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}

int qs_mailbox_close(void* ptr) { return ptr ? hip_status(hipIpcCloseMemHandle(ptr)) : QS_OK; }

int qs_mailbox_publish(const float* rec, int64_t n, void* const* boxes, int world, int rank, uint32_t step, qs_stream_t stream) {
    if (!rec || !boxes || n < 1 || world < 1 || world > kMailboxMaxWorld || rank < 0 || rank >= world || step == 0) return QS_ERR_ARG;
    MailboxPeers peers{};
    for (int r = 0; r < world; ++r) {
        if (!boxes[r]) return QS_ERR_ARG;
        peers.box[r] = reinterpret_cast<uint32_t*>(boxes[r]);
    }
    hipLaunchKernelGGL(mailbox_publish_kernel, dim3(world), dim3(256), 0, (hipStream_t)stream, rec, n, peers, world, rank, step);
    return launch_status();
}

int qs_mailbox_wait(void* box, int world, int64_t n, uint32_t step, int32_t* status, uint32_t max_spins, const float** records,
                    qs_stream_t stream) {
    if (!box || !status || n < 1 || world < 1 || world > kMailboxMaxWorld || step == 0 || max_spins == 0) return QS_ERR_ARG;
    hipLaunchKernelGGL(mailbox_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<uint32_t*>(box), world, n, step,
                       status, max_spins);
    if (records) *records = reinterpret_cast<const float*>(reinterpret_cast<uint32_t*>(box) + kMailboxHeader) + (size_t)(step & 1u) * world * n;
    return launch_status();
}

int qs_records_max(const float* records, int world, int64_t n, float* out, qs_stream_t stream) {
    if (!records || !out || world < 1 || n < 1) return QS_ERR_ARG;
    hipLaunchKernelGGL(records_max_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint32_t*>(records), world, n, reinterpret_cast<uint32_t*>(out));
    return launch_status();
}

}  // extern "C"
