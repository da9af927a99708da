// comm.hip - the multi-GPU leg of the C ABI: RCCL over xGMI, one process per GPU.
//
// The path shards along the ensemble axis only: independent problem instances
// (different x, y; same W) are dealt to the ranks, which never exchange anything per iteration.
// What the reference does as a sequential Python loop over seeds
// (tramp/experiments/multiple_experiments.py:30-49) needs exactly three collectives here:
//   * tramp_bcast_matrix  - ONE broadcast of the shared sensing matrix (or of its thin factors, so
//     that the eigensolve runs once instead of once per GPU) at setup;
//   * tramp_gather_scalars - the per-instance results (n_iter, v, mse) at the end;
//   * tramp_comm_max / tramp_comm_barrier - bracketing a timed region.
// librccl is loaded on first use (dlopen): a single-GPU process never maps its 570 MB.
// The unique id travels between the processes by whatever rendezvous the host side has
// (tramp_amd/ensemble.py uses a TCP exchange on MASTER_ADDR).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    char err[256] = {0};
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) {
        snprintf(g_rccl.err, sizeof(g_rccl.err), "dlopen(librccl.so.1): %s", dlerror());
        return;
    }
#define RCCL_SYM(field, name)                                                         \
    do {                                                                              \
        *(void**)(&g_rccl.field) = dlsym(g_rccl.lib, name);                           \
        if (!g_rccl.field) {                                                          \
            snprintf(g_rccl.err, sizeof(g_rccl.err), "librccl lacks %s", name);      \
            return;                                                                   \
        }                                                                             \
    } while (0)
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(Broadcast, "ncclBroadcast");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(AllReduce, "ncclAllReduce");
#undef RCCL_SYM
}

const Rccl* rccl(tramp_ctx* ctx) {
    std::call_once(g_rccl_once, rccl_load);
    if (g_rccl.err[0]) {
        if (ctx) tramp_fail(ctx, TRAMP_ENOTSUP, "RCCL", g_rccl.err);
        return nullptr;
    }
    return &g_rccl;
}

}  // namespace

struct tramp_comm {
    ncclComm_t comm;
    int rank, world;
    double* scratch;      // one double on the device (max / barrier)
};

#define NCCL_CHECK(ctx, R, call)                                                                   \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess) return tramp_fail(ctx, TRAMP_EHIP, #call, (R)->GetErrorString(r_)); \
    } while (0)

// Can this process reach RCCL at all (library maps, every symbol resolves)?  No collective, no device call:
// the ranks exchange this flag over their rendezvous BEFORE anyone enters ncclCommInitRank, so that one
// rank without the library cannot leave the others waiting in the bootstrap.
extern "C" int tramp_comm_available(tramp_ctx* ctx) {
    if (!ctx) return TRAMP_EINVAL;
    return rccl(ctx) ? TRAMP_OK : TRAMP_ENOTSUP;
}

extern "C" int tramp_comm_unique_id(tramp_ctx* ctx, unsigned char* id_out) {
    if (!ctx || !id_out) return TRAMP_EINVAL;
    const Rccl* R = rccl(ctx);
    if (!R) return TRAMP_ENOTSUP;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    NCCL_CHECK(ctx, R, R->GetUniqueId(&id));
    static_assert(sizeof(id) == TRAMP_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return TRAMP_OK;
}

extern "C" int tramp_comm_init(tramp_ctx* ctx, int rank, int world, const unsigned char* id, tramp_comm** out) {
    if (!ctx || !out) return TRAMP_EINVAL;
    *out = nullptr;
    TRAMP_REQUIRE(ctx, id && world >= 1 && rank >= 0 && rank < world);
    const Rccl* R = rccl(ctx);
    if (!R) return TRAMP_ENOTSUP;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    tramp_comm* c = new (std::nothrow) tramp_comm();
    if (!c) return TRAMP_ENOMEM;
    c->rank = rank;
    c->world = world;
    c->scratch = nullptr;
    ncclResult_t r = R->CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return tramp_fail(ctx, TRAMP_EHIP, "ncclCommInitRank", R->GetErrorString(r));
    }
    if (hipMalloc((void**)&c->scratch, 2 * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        (void)R->CommDestroy(c->comm);
        delete c;
        return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_comm_init", "hipMalloc failed");
    }
    *out = c;
    return TRAMP_OK;
}

extern "C" int tramp_comm_destroy(tramp_ctx* ctx, tramp_comm* comm) {
    if (!ctx || !comm) return TRAMP_EINVAL;
    const Rccl* R = rccl(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (R) (void)R->CommDestroy(comm->comm);
    if (comm->scratch) (void)hipFree(comm->scratch);
    delete comm;
    return TRAMP_OK;
}

extern "C" int tramp_comm_rank(const tramp_comm* comm, int* rank, int* world) {
    if (!comm || !rank || !world) return TRAMP_EINVAL;
    *rank = comm->rank;
    *world = comm->world;
    return TRAMP_OK;
}

extern "C" int tramp_bcast_matrix(tramp_ctx* ctx, tramp_comm* comm, void* buf_d, size_t bytes, int root) {
    if (!ctx || !comm) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, (buf_d || bytes == 0) && root >= 0 && root < comm->world);
    const Rccl* R = rccl(ctx);
    if (!R) return TRAMP_ENOTSUP;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // one collective per <= 1 GiB piece: large enough for the ring to run at the link rate,
    // small enough that no 32-bit count inside the library is ever at risk
    const size_t piece = (size_t)1 << 30;
    char* p = (char*)buf_d;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t nb = bytes - off < piece ? bytes - off : piece;
        NCCL_CHECK(ctx, R, R->Broadcast(p + off, p + off, nb, ncclChar, root, comm->comm, ctx->stream));
    }
    return TRAMP_OK;
}

extern "C" int tramp_gather_scalars(tramp_ctx* ctx, tramp_comm* comm, const double* send_d, size_t count,
                                    double* recv_d) {
    if (!ctx || !comm) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, send_d && recv_d && count > 0);
    const Rccl* R = rccl(ctx);
    if (!R) return TRAMP_ENOTSUP;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    NCCL_CHECK(ctx, R, R->AllGather(send_d, recv_d, count, ncclDouble, comm->comm, ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_comm_max(tramp_ctx* ctx, tramp_comm* comm, double* value) {
    if (!ctx || !comm || !value) return TRAMP_EINVAL;
    const Rccl* R = rccl(ctx);
    if (!R) return TRAMP_ENOTSUP;
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(comm->scratch, value, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    NCCL_CHECK(ctx, R, R->AllReduce(comm->scratch, comm->scratch + 1, 1, ncclDouble, ncclMax, comm->comm, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(value, comm->scratch + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_comm_barrier(tramp_ctx* ctx, tramp_comm* comm) {
    double one = 1.0;
    return tramp_comm_max(ctx, comm, &one);      // a collective every rank must enter + a stream sync
}
