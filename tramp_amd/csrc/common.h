// common.h - context, error handling and wave/block reduction helpers (gfx950).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tramp_hip.h"

// ---- optional per-kernel timing (bench.py roofline): HIP events around tagged launches
enum TrampProfTag {
    PROF_EW = 0,        // elementwise factor update (both passes)
    PROF_MATVEC_NT,     // row-dot matrix pass (GEMV or MFMA)
    PROF_MATVEC_NN,     // column-accumulate matrix pass (+ fold)
    PROF_LIN_SCALARS,   // K9 + spectral coefficients
    PROF_VAR_POST,      // K11
    PROF_OTHER,
    PROF_FFT,           // rocFFT transforms (convolution channel)
    PROF_NTAGS
};

struct TrampProfRec {
    hipEvent_t start, stop;
    int tag;
    double bytes, flops;
};

// host side of prior_hint.h: what tramp_prior_hint announced and whether a producer honoured it
struct TrampPriorHint {
    int kind;                 // 0 = none
    double p0, p1, p2;
    const void* buf;          // the message buffer b the prior will read
    const double* a_msg;      // its precision array
    const void* y;            // likelihood kinds: the observation array the consumer will be called with
    size_t n, batch;
    double* partials;         // owned by the context (hint_ws)
    int nblk;                 // set by the producer
    bool ready;               // a producer has left the partials of the CURRENT contents of buf
};

// Granules of the one-launch form (ew_coop), in two banks of [capacity] block sums followed by the instance totals:
// R <= 16 copies per instance (one per 32 workgroups, so no line has more than 32 pollers), each copy on a line of its
// own 4352 bytes from the next.  batch * R <= capacity / 32 + batch lines.
constexpr size_t kCoopLineStride = 544;            // in 8-byte granules
__host__ __device__ inline size_t coop_total_lines(size_t capacity) { return capacity / 32 + capacity + 2; }
inline size_t coop_slot_bytes(size_t capacity) { return 2 * (capacity + coop_total_lines(capacity) * kCoopLineStride) * 8; }

struct tramp_ctx {
    int device;
    hipStream_t stream;
    double* ws;          // reduction workspace (device)
    size_t ws_bytes;
    double* hscratch;    // pinned host scratch for SE scalar round trips
    size_t hscratch_bytes;
    int num_cu;
    char err[512];
    // profiling
    bool prof_on;
    bool capturing;      // inside tramp_graph_begin/end
    TrampProfRec* prof;
    int prof_cap, prof_n;
    // producer-fused prior moments (prior_hint.h)
    TrampPriorHint hint;
    double* hint_ws;
    size_t hint_ws_bytes;
    unsigned long long hint_armed, hint_used;   // producers that left moments / prior updates that ran single-pass
    unsigned long long resident_runs;           // factor updates that took the resident single-pass pipeline (ew_resident.h)
    volatile unsigned* lean_stat;               // pinned host {flagged instances, batch} of the last lean resident update (pass 1 of its redo: ew_moments_tile, RedoArgs);
                                                // [2]: a resident kernel gave up a hand-off (ew_resident.h: res_expired)
    unsigned long long* coop_slots;             // two banks of granules of the one-launch form (ew_kernels.hip: ew_coop); the bank the next launch uses is all sentinel
    unsigned* coop_epoch;                       // device word: launches of the one-launch form so far (its parity picks the bank)
    size_t capacity_coop;                       // 2 x CUs: the workgroups one such launch may use
    unsigned long long coop_runs;               // updates that took the one-launch form
    size_t fz_peak_bytes;                       // work space of the last tramp_linear_factorize (factorize.hip)
    int resident_errors_seen;                   // the host has seen lean_stat[2] and taken the resident pipeline out of service
    int lean_skip;                              // lean updates to skip after one that flagged more than a quarter of its instances
    void* sliced_reg;                           // int8-digit copies of resident matrices (gemm_sliced.hip), by device pointer
    unsigned long long sliced_runs;             // dense passes that ran on a digit copy
    int graphs_alive;    // captured graphs hold raw workspace pointers: no growth while > 0
    // launch shape of the last MFMA matrix pass (tramp_matvec_last_launch): split-K factor, 64-instance chunks,
    // whether sibling chunks were interleaved in a one-dimensional grid
    int last_nsplit, last_nchunk, last_flat;
    // buffers released while a stream capture is open (the host language's garbage collector may drop a device array at
    // any moment): hipStreamSynchronize / hipFree would invalidate the capture, so they wait for tramp_graph_end
    void** deferred_free;
    int deferred_n, deferred_cap;
};

// RAII scope: records start/stop events on the context's stream when profiling is on.
struct ProfScope {
    tramp_ctx* ctx;
    TrampProfRec* rec;
    ProfScope(tramp_ctx* c, int tag, double bytes, double flops) : ctx(c), rec(nullptr) {
        if (!c->prof_on || c->prof_n >= c->prof_cap) return;
        rec = &c->prof[c->prof_n++];
        rec->tag = tag;
        rec->bytes = bytes;
        rec->flops = flops;
        (void)hipEventRecord(rec->start, c->stream);
    }
    ~ProfScope() {
        if (rec) (void)hipEventRecord(rec->stop, ctx->stream);
    }
};

struct tramp_event {
    hipEvent_t ev;
};

static inline int tramp_fail(tramp_ctx* ctx, int code, const char* what, const char* detail) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, detail ? detail : "");
    return code;
}

#define TRAMP_HIP_CHECK(ctx, call)                                                   \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) return tramp_fail(ctx, TRAMP_EHIP, #call, hipGetErrorString(e_)); \
    } while (0)

#define TRAMP_REQUIRE(ctx, cond)                                                     \
    do {                                                                             \
        if (!(cond)) return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", #cond); \
    } while (0)

#define TRAMP_LAUNCH_CHECK(ctx)                                                      \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) return tramp_fail(ctx, TRAMP_EHIP, "kernel launch", hipGetErrorString(e_)); \
    } while (0)

// Grow the context's reduction workspace to at least `bytes` (synchronises when it grows).
int tramp_ws_reserve(tramp_ctx* ctx, size_t bytes);

constexpr int kBlock = 256;           // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kMaxBlocksPerInst = 1024;

// ---- wave64 / block reductions (deterministic: fixed tree) ------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over a 256-thread block; result valid in every thread. `smem` >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* smem) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();                 // protect smem reuse between calls
    if (lane == 0) smem[w] = v;
    __syncthreads();
    double t = smem[0];
#pragma unroll
    for (int i = 1; i < kBlock / kWave; ++i) t += smem[i];
    return t;
}

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v,
                                                            unsigned long long* smem) {
    v = wave_sum_u64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    unsigned long long t = smem[0];
#pragma unroll
    for (int i = 1; i < kBlock / kWave; ++i) t += smem[i];
    return t;
}

// 16-byte vector views of the element type.
template <typename T> struct Vec;
template <> struct Vec<double> {
    using type = double2;
    static constexpr int N = 2;
};
template <> struct Vec<float> {
    using type = float4;
    static constexpr int N = 4;
};

template <typename T> __device__ __forceinline__ T vget(const typename Vec<T>::type& v, int i);
template <> __device__ __forceinline__ double vget<double>(const double2& v, int i) { return i ? v.y : v.x; }
template <> __device__ __forceinline__ float vget<float>(const float4& v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
template <typename T> __device__ __forceinline__ void vset(typename Vec<T>::type& v, int i, T x);
template <> __device__ __forceinline__ void vset<double>(double2& v, int i, double x) { if (i) v.y = x; else v.x = x; }
template <> __device__ __forceinline__ void vset<float>(float4& v, int i, float x) {
    if (i == 0) v.x = x; else if (i == 1) v.y = x; else if (i == 2) v.z = x; else v.w = x;
}

static inline size_t dtype_size(int dtype) { return dtype == TRAMP_F64 ? 8 : 4; }

// blocks per instance for an n-element streaming pass (>= 4 vectors of work per thread)
static inline int blocks_for(size_t n, int vecN) {
    size_t per_block = (size_t)kBlock * vecN * 4;
    size_t nb = (n + per_block - 1) / per_block;
    if (nb < 1) nb = 1;
    if (nb > kMaxBlocksPerInst) nb = kMaxBlocksPerInst;
    return (int)nb;
}
