// capi.hip - context, device memory and timing entry points of the C ABI.
#include "common.h"
#include "linear.h"
#include "prior_hint.h"
#include <stdlib.h>

static void flush_deferred_frees(tramp_ctx* ctx);
void gemm_sliced_mark_dead(tramp_ctx* ctx, const void* mat_d);
void gemm_sliced_reap(tramp_ctx* ctx);

extern "C" const char* tramp_version(void) { return "tramp_hip 0.1 (gfx950)"; }

extern "C" int tramp_device_count(int* n) {
    if (!n) return TRAMP_EINVAL;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return TRAMP_OK;
}

extern "C" int tramp_ctx_create(int device, tramp_ctx** out) {
    if (!out) return TRAMP_EINVAL;
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
        (void)hipGetLastError();
        return TRAMP_ENODEV;
    }
    if (device < 0 || device >= c) return TRAMP_EINVAL;
    tramp_ctx* ctx = new (std::nothrow) tramp_ctx();
    if (!ctx) return TRAMP_ENOMEM;
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return TRAMP_EHIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
    ctx->ws_bytes = 4u << 20;
    if (hipMalloc((void**)&ctx->ws, ctx->ws_bytes) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return TRAMP_ENOMEM;
    }
    ctx->hscratch_bytes = 1u << 20;
    if (hipHostMalloc((void**)&ctx->hscratch, ctx->hscratch_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(ctx->ws);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return TRAMP_ENOMEM;
    }
    // granules and arrival counters of the one-launch elementwise form: sentinel / zero between calls
    ctx->capacity_coop = (size_t)2 * (size_t)(ctx->num_cu > 0 ? ctx->num_cu : 0);
    if (ctx->capacity_coop) {
        if (hipMalloc((void**)&ctx->coop_slots, coop_slot_bytes(ctx->capacity_coop)) == hipSuccess &&
            hipMalloc((void**)&ctx->coop_epoch, 64) == hipSuccess &&
            hipMemset(ctx->coop_slots, 0xFF, coop_slot_bytes(ctx->capacity_coop)) == hipSuccess &&
            hipMemset(ctx->coop_epoch, 0, 64) == hipSuccess) {
        } else {
            (void)hipGetLastError();
            if (ctx->coop_slots) (void)hipFree(ctx->coop_slots);
            if (ctx->coop_epoch) (void)hipFree(ctx->coop_epoch);
            ctx->coop_slots = nullptr;
            ctx->coop_epoch = nullptr;
            ctx->capacity_coop = 0;
        }
    }
    unsigned* stat = nullptr;
    if (hipHostMalloc((void**)&stat, 64, hipHostMallocDefault) == hipSuccess) {       // (without it the lean updates keep no statistics)
        stat[0] = stat[1] = stat[2] = 0;
        ctx->lean_stat = stat;
    } else {
        (void)hipGetLastError();
    }
    *out = ctx;
    return TRAMP_OK;
}

extern "C" int tramp_ctx_destroy(tramp_ctx* ctx) {
    if (!ctx) return TRAMP_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->lean_stat) (void)hipHostFree((void*)ctx->lean_stat);
    if (ctx->coop_slots) (void)hipFree(ctx->coop_slots);
    if (ctx->coop_epoch) (void)hipFree(ctx->coop_epoch);
    gemm_sliced_ctx_destroy(ctx);
    flush_deferred_frees(ctx);
    free(ctx->deferred_free);
    if (ctx->hint_ws) (void)hipFree(ctx->hint_ws);
    if (ctx->hscratch) (void)hipHostFree(ctx->hscratch);
    if (ctx->prof) {
        for (int i = 0; i < ctx->prof_cap; ++i) {
            (void)hipEventDestroy(ctx->prof[i].start);
            (void)hipEventDestroy(ctx->prof[i].stop);
        }
        delete[] ctx->prof;
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TRAMP_OK;
}

struct tramp_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" int tramp_graph_begin(tramp_ctx* ctx) {
    if (!ctx) return TRAMP_EINVAL;
    if (ctx->prof_on) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_graph_begin", "disable profiling before capture");
    ctx->capturing = true;
    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        ctx->capturing = false;
        return tramp_fail(ctx, TRAMP_EHIP, "hipStreamBeginCapture", hipGetErrorString(e));
    }
    return TRAMP_OK;
}

extern "C" int tramp_graph_end(tramp_ctx* ctx, tramp_graph** out) {
    if (!ctx || !out) return TRAMP_EINVAL;
    *out = nullptr;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    ctx->capturing = false;
    flush_deferred_frees(ctx);
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        return tramp_fail(ctx, TRAMP_EHIP, "hipStreamEndCapture", hipGetErrorString(e));
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return tramp_fail(ctx, TRAMP_EHIP, "hipGraphInstantiate", hipGetErrorString(e));
    }
    tramp_graph* tg = new (std::nothrow) tramp_graph();
    if (!tg) {
        (void)hipGraphExecDestroy(ex);
        (void)hipGraphDestroy(g);
        return TRAMP_ENOMEM;
    }
    tg->graph = g;
    tg->exec = ex;
    *out = tg;
    ctx->graphs_alive += 1;      // the graph holds raw pointers into ctx->ws / hint_ws: no growth until it is gone
    return TRAMP_OK;
}

extern "C" int tramp_graph_launch(tramp_ctx* ctx, tramp_graph* graph) {
    if (!ctx || !graph) return TRAMP_EINVAL;
    TRAMP_HIP_CHECK(ctx, hipGraphLaunch(graph->exec, ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_graph_destroy(tramp_ctx* ctx, tramp_graph* graph) {
    if (!ctx || !graph) return TRAMP_EINVAL;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipGraphExecDestroy(graph->exec);
    (void)hipGraphDestroy(graph->graph);
    delete graph;
    if (ctx->graphs_alive > 0) ctx->graphs_alive -= 1;
    gemm_sliced_reap(ctx);       // digit copies detached while this graph could still replay them (stream synchronised above)
    return TRAMP_OK;
}

extern "C" int tramp_profile_enable(tramp_ctx* ctx, int on) {
    if (!ctx) return TRAMP_EINVAL;
    if (on && !ctx->prof) {
        ctx->prof_cap = 8192;
        ctx->prof = new (std::nothrow) TrampProfRec[ctx->prof_cap];
        if (!ctx->prof) return TRAMP_ENOMEM;
        for (int i = 0; i < ctx->prof_cap; ++i) {
            TRAMP_HIP_CHECK(ctx, hipEventCreate(&ctx->prof[i].start));
            TRAMP_HIP_CHECK(ctx, hipEventCreate(&ctx->prof[i].stop));
        }
    }
    ctx->prof_on = on != 0;
    ctx->prof_n = 0;
    return TRAMP_OK;
}

extern "C" int tramp_profile_report(tramp_ctx* ctx, int* count, double* ms, double* bytes, double* flops) {
    if (!ctx || !count || !ms || !bytes || !flops) return TRAMP_EINVAL;
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < PROF_NTAGS; ++t) {
        count[t] = 0;
        ms[t] = bytes[t] = flops[t] = 0.0;
    }
    for (int i = 0; i < ctx->prof_n; ++i) {
        const TrampProfRec& r = ctx->prof[i];
        float e = 0.f;
        TRAMP_HIP_CHECK(ctx, hipEventElapsedTime(&e, r.start, r.stop));
        count[r.tag] += 1;
        ms[r.tag] += e;
        bytes[r.tag] += r.bytes;
        flops[r.tag] += r.flops;
    }
    ctx->prof_n = 0;
    return TRAMP_OK;
}

extern "C" const char* tramp_last_error(const tramp_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int tramp_ws_reserve(tramp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return TRAMP_OK;
    if (ctx->capturing)
        return tramp_fail(ctx, TRAMP_EINVAL, "workspace", "would have to grow during graph capture; run the "
                          "sequence eagerly once first");
    if (ctx->graphs_alive > 0)
        return tramp_fail(ctx, TRAMP_ENOTSUP, "workspace", "a captured graph of this context still points into the "
                          "workspace, which this call would have to re-allocate; destroy the graph first or use "
                          "another context for the larger call");
    (void)hipSetDevice(ctx->device);
    size_t nb = ctx->ws_bytes;
    while (nb < bytes) nb *= 2;
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    if (hipMalloc((void**)&ctx->ws, nb) != hipSuccess) {
        (void)hipGetLastError();
        return tramp_fail(ctx, TRAMP_ENOMEM, "workspace", "hipMalloc failed");
    }
    ctx->ws_bytes = nb;
    return TRAMP_OK;
}

/* One process may hold contexts on several GPUs.  HIP's "current device" is per host thread: the
 * entry points that allocate set it themselves, everything else (launches on ctx->stream) needs the
 * caller to make the context current when it switches between contexts on one thread.  The Python
 * mirror does this in Context._call. */
extern "C" int tramp_ctx_make_current(tramp_ctx* ctx) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return TRAMP_OK;
}

bool prior_hint_arm(tramp_ctx* ctx, const void* out_buf, size_t len, size_t batch, int nblk, PriorHintDev* dev) {
    dev->kind = PRIOR_HINT_NONE;
    TrampPriorHint& h = ctx->hint;
    if (!h.kind || h.buf != out_buf || h.n != len || h.batch != batch || nblk <= 0) return false;
    const size_t need = batch * (size_t)nblk * sizeof(double);
    if (need > ctx->hint_ws_bytes) {
        if (ctx->capturing || ctx->graphs_alive) return false;      // never move a buffer a graph points at
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->hint_ws) (void)hipFree(ctx->hint_ws);
        ctx->hint_ws = nullptr;
        ctx->hint_ws_bytes = 0;
        if (hipMalloc((void**)&ctx->hint_ws, need) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        ctx->hint_ws_bytes = need;
    }
    h.partials = ctx->hint_ws;
    h.nblk = nblk;
    h.ready = true;
    ctx->hint_armed += 1;
    *dev = PriorHintDev{h.kind, h.p0, h.p1, h.p2, h.a_msg, h.partials, nblk, h.y, h.n};
    return true;
}

extern "C" int tramp_prior_hint_stats(tramp_ctx* ctx, unsigned long long* armed, unsigned long long* used) {
    if (!ctx || !armed || !used) return TRAMP_EINVAL;
    *armed = ctx->hint_armed;
    *used = ctx->hint_used;
    return TRAMP_OK;
}

bool resident_covers(const tramp_ctx* ctx, size_t n, size_t batch, size_t elem_size);
extern "C" int tramp_resident_covers(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int* yes) {
    if (!ctx || !yes || (dtype != TRAMP_F32 && dtype != TRAMP_F64)) return TRAMP_EINVAL;
    *yes = resident_covers(ctx, n, batch, dtype == TRAMP_F32 ? 4 : 8) ? 1 : 0;
    return TRAMP_OK;
}

extern "C" int tramp_resident_stats(tramp_ctx* ctx, unsigned long long* runs) {
    if (!ctx || !runs) return TRAMP_EINVAL;
    *runs = ctx->resident_runs;
    return TRAMP_OK;
}

extern "C" int tramp_coop_stats(tramp_ctx* ctx, unsigned long long* runs) {
    if (!ctx || !runs) return TRAMP_EINVAL;
    *runs = ctx->coop_runs;
    return TRAMP_OK;
}

extern "C" int tramp_resident_errors(tramp_ctx* ctx, unsigned long long* gave_up) {
    if (!ctx || !gave_up) return TRAMP_EINVAL;
    *gave_up = (ctx->lean_stat && ctx->lean_stat[2]) ? 1ull : 0ull;
    return TRAMP_OK;
}

extern "C" int tramp_sync(tramp_ctx* ctx) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_ctx_stream(tramp_ctx* ctx, void** stream) {
    if (!ctx || !stream) return TRAMP_EINVAL;
    *stream = (void*)ctx->stream;
    return TRAMP_OK;
}

extern "C" int tramp_device_name(tramp_ctx* ctx, char* out, size_t cap) {
    if (!ctx || !out || !cap) return TRAMP_EINVAL;
    hipDeviceProp_t prop;
    TRAMP_HIP_CHECK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    snprintf(out, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return TRAMP_OK;
}

extern "C" int tramp_mem_info(tramp_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return TRAMP_EINVAL;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    TRAMP_HIP_CHECK(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return TRAMP_OK;
}

extern "C" int tramp_buf_alloc(tramp_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return TRAMP_EINVAL;
    *dptr = nullptr;
    // A buffer born inside tramp_graph_begin / _end would be baked into the captured kernel nodes and could be freed
    // (deferred, below) before the first replay: no allocation while a capture is open - the caller allocates first.
    if (ctx->capturing) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_buf_alloc", "allocation inside a graph capture");
    if (bytes == 0) bytes = 16;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return tramp_fail(ctx, e == hipErrorOutOfMemory ? TRAMP_ENOMEM : TRAMP_EHIP, "hipMalloc",
                          hipGetErrorString(e));
    }
    return TRAMP_OK;
}

static void flush_deferred_frees(tramp_ctx* ctx) {
    if (!ctx->deferred_n) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < ctx->deferred_n; ++i) (void)hipFree(ctx->deferred_free[i]);      // (their digit copies were marked dead at deferral)
    ctx->deferred_n = 0;
    gemm_sliced_reap(ctx);
}

extern "C" int tramp_buf_free(tramp_ctx* ctx, void* dptr) {
    if (!ctx) return TRAMP_EINVAL;
    if (!dptr) return TRAMP_OK;
    if (ctx->capturing) {
        // inside tramp_graph_begin / _end: a synchronisation or a free here would invalidate the capture
        if (ctx->deferred_n == ctx->deferred_cap) {
            const int cap = ctx->deferred_cap ? 2 * ctx->deferred_cap : 64;
            void** p = (void**)realloc(ctx->deferred_free, (size_t)cap * sizeof(void*));
            if (!p) return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_buf_free", "deferred list");
            ctx->deferred_free = p;
            ctx->deferred_cap = cap;
        }
        ctx->deferred_free[ctx->deferred_n++] = dptr;
        // a digit copy keyed by this pointer is out of use from now on: the address may be handed out again before the flush
        if (ctx->sliced_reg) gemm_sliced_mark_dead(ctx, dptr);
        return TRAMP_OK;
    }
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // a digit copy keyed by this pointer (gemm_sliced.hip) must not outlive the matrix: the next allocation may get
    // the same address
    if (ctx->sliced_reg) (void)tramp_matrix_detach_digits(ctx, dptr);
    TRAMP_HIP_CHECK(ctx, hipFree(dptr));
    return TRAMP_OK;
}

extern "C" int tramp_h2d(tramp_ctx* ctx, void* dst_d, const void* src_h, size_t bytes) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, (dst_d && src_h) || bytes == 0);
    if (!bytes) return TRAMP_OK;
    // pageable source: the call returns once the staging copy is done
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_d2h(tramp_ctx* ctx, void* dst_h, const void* src_d, size_t bytes) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, (dst_h && src_d) || bytes == 0);
    if (!bytes) return TRAMP_OK;
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TRAMP_OK;
}

// device-to-device copy as a KERNEL on the context's stream (16-byte pieces, tail by bytes): the schedulers copy
// messages between launches (variable -> factor relays on a tree, the per-iteration snapshot of restore_messages) and a
// hipMemcpyAsync in the middle of a kernel sequence goes through the copy engine and its cross-queue signals
__global__ __launch_bounds__(256) void d2d_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16,
                                                   unsigned char* __restrict__ dtail, const unsigned char* __restrict__ stail,
                                                   unsigned ntail) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < ntail) dtail[threadIdx.x] = stail[threadIdx.x];
}

extern "C" int tramp_d2d(tramp_ctx* ctx, void* dst_d, const void* src_d, size_t bytes) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, (dst_d && src_d) || bytes == 0);
    if (!bytes) return TRAMP_OK;
    if (((uintptr_t)dst_d | (uintptr_t)src_d) % 16 == 0) {
        const size_t n16 = bytes / 16;
        const unsigned ntail = (unsigned)(bytes % 16);
        size_t g = (n16 + 255) / 256;
        if (g > 65536) g = 65536;
        if (g < 1) g = 1;
        hipLaunchKernelGGL(d2d_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, (uint4*)dst_d, (const uint4*)src_d, n16,
                           (unsigned char*)dst_d + n16 * 16, (const unsigned char*)src_d + n16 * 16, ntail);
        TRAMP_LAUNCH_CHECK(ctx);
        return TRAMP_OK;
    }
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(dst_d, src_d, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_memset(tramp_ctx* ctx, void* dst_d, int value, size_t bytes) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, dst_d || bytes == 0);
    if (!bytes) return TRAMP_OK;
    TRAMP_HIP_CHECK(ctx, hipMemsetAsync(dst_d, value, bytes, ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_event_create(tramp_ctx* ctx, tramp_event** ev) {
    if (!ctx || !ev) return TRAMP_EINVAL;
    tramp_event* e = new (std::nothrow) tramp_event();
    if (!e) return TRAMP_ENOMEM;
    hipError_t rc = hipEventCreate(&e->ev);
    if (rc != hipSuccess) {
        delete e;
        return tramp_fail(ctx, TRAMP_EHIP, "hipEventCreate", hipGetErrorString(rc));
    }
    *ev = e;
    return TRAMP_OK;
}

extern "C" int tramp_event_destroy(tramp_ctx* ctx, tramp_event* ev) {
    if (!ctx || !ev) return TRAMP_EINVAL;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return TRAMP_OK;
}

extern "C" int tramp_event_record(tramp_ctx* ctx, tramp_event* ev) {
    if (!ctx || !ev) return TRAMP_EINVAL;
    TRAMP_HIP_CHECK(ctx, hipEventRecord(ev->ev, ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_event_elapsed_ms(tramp_ctx* ctx, tramp_event* start, tramp_event* stop, float* ms) {
    if (!ctx || !start || !stop || !ms) return TRAMP_EINVAL;
    TRAMP_HIP_CHECK(ctx, hipEventSynchronize(stop->ev));
    TRAMP_HIP_CHECK(ctx, hipEventElapsedTime(ms, start->ev, stop->ev));
    return TRAMP_OK;
}
