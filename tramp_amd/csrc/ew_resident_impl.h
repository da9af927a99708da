// ew_resident_impl.h - definition of run_resident; included only by the ew_resident_*.hip translation units.
#pragma once
#include <stdlib.h>

#include "ew_resident.h"
#include "ew_resident_host.h"

static_assert(kResCombiners == 8 && kResBlocksPerCU == 2, "resident_covers (ew_kernels.hip) counts 2 x CUs - 8 streaming workgroups");

template <typename T, typename Op>
int run_resident(tramp_ctx* ctx, const Op& op, size_t n, size_t batch, const void* in0, const void* in1,
                 const double* a_in_d, const MsgOut& out, const void* bits, size_t bits_stride) {
    constexpr int VN = Vec<T>::N;
    constexpr int XF = ResShape<Op>::FLAGS;
    constexpr int U = ResShape<Op>::U, NT = ResShape<Op>::NT, DL = kResDL, DA = kResDA, BPC = ResShape<Op>::BPC;
    constexpr size_t TV = (size_t)NT * U;
    if (out.mask || out.r || !resident_covers_shape(ctx, n, batch, sizeof(T), TV, BPC)) return kResNotEligible;
    // A resident update of this context missed a hand-off (ew_resident.h: res_expired) - or the process runs under a CU
    // mask, which the occupancy query below does not see: the tile kernels from now on.
    if (ctx->lean_stat && ctx->lean_stat[2]) {
        if (!ctx->resident_errors_seen) ctx->resident_errors_seen = 1;
        return kResNotEligible;
    }
    static const bool cu_masked = (getenv("HSA_CU_MASK") && *getenv("HSA_CU_MASK")) || (getenv("ROC_GLOBAL_CU_MASK") && *getenv("ROC_GLOBAL_CU_MASK"));
    if (cu_masked) return kResNotEligible;
    constexpr bool LEAN = OpLean<Op>::value;
    if (LEAN) {
        // a flagged instance is redone FROM ITS INPUTS (by the caller, on the tile kernels) after the pipeline has written its
        // outputs: no in-place updates
        const char *o0 = reinterpret_cast<const char*>(out.b_new), *o1 = o0 + batch * n * sizeof(T);
        for (const void* in : {in0, in1}) {
            const char* i0 = reinterpret_cast<const char*>(in);
            if (i0 && i0 < o1 && o0 < i0 + batch * n * sizeof(T)) return kResNotEligible;
        }
    }
    const size_t nv = n / VN;
    // all workgroups must be resident at once: two per CU by the launch bounds, checked once per instantiation
    static int occ_plain = -1, occ_damp = -1;
    if (occ_plain < 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_plain, ew_resident<T, Op, U, DL, DA, kResSchedBarrier | XF, BPC, 0, NT>, NT, 0) != hipSuccess) occ_plain = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_damp, ew_resident<T, Op, U, DL, DA, kResDamp | kResSchedBarrier | XF, BPC, 0, NT>, NT, 0) != hipSuccess) occ_damp = 0;
    }
    const bool damp = out.damping != 0.0;
    if ((damp ? occ_damp : occ_plain) < BPC) return kResNotEligible;
    const unsigned P = (unsigned)(BPC * ctx->num_cu), PS = P - kResCombiners;
    const size_t tiles = (nv + TV - 1) / TV;
    ResArgs A{};
    A.n = n;
    A.nv = nv;
    A.batch = (unsigned)batch;
    A.tiles = (unsigned)tiles;
    A.total = (unsigned long long)batch * tiles;
    A.adv_q = PS / A.tiles;
    A.adv_r = PS % A.tiles;
    A.a_in = a_in_d;
    const size_t ops_bytes = ((batch * sizeof(Op) + 255) / 256) * 256;
    const size_t slot_bytes = ((A.total * 8 + 255) / 256) * 256;
    const size_t scal_bytes = ((batch * 8 + 255) / 256) * 256;
    const size_t dummy_bytes = TV * 16;
    const bool count = OpCounts<Op>::value;
    int rc = tramp_ws_reserve(ctx, ops_bytes + slot_bytes * (count ? 2 : 1) + scal_bytes + dummy_bytes);
    if (rc) return rc;
    char* ws = reinterpret_cast<char*>(ctx->ws);
    Op* ops = reinterpret_cast<Op*>(ws);
    ws += ops_bytes;
    A.slots = reinterpret_cast<unsigned long long*>(ws);
    ws += slot_bytes;
    A.cslots = nullptr;
    if (count) {
        A.cslots = reinterpret_cast<unsigned long long*>(ws);
        ws += slot_bytes;
    }
    A.scal = reinterpret_cast<unsigned long long*>(ws);
    ws += scal_bytes;
    A.dummy = ws;
    A.bits = reinterpret_cast<const uint8_t*>(bits);
    A.bits_stride = bits_stride;
    if (OpBits<Op>::value && !bits) return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "packed observation");
    A.out = out;
    if (!damp) { A.out.a_old = nullptr; A.out.b_old = nullptr; }
    static const long long timeout_ms = getenv("TRAMP_EW_RESIDENT_TIMEOUT_MS") ? atoll(getenv("TRAMP_EW_RESIDENT_TIMEOUT_MS")) : 2000;
    static const long long dbg_skip = getenv("TRAMP_EW_RESIDENT_DBG_SKIP") ? atoll(getenv("TRAMP_EW_RESIDENT_DBG_SKIP")) : 0;
    A.err_host = ctx->lean_stat ? ctx->lean_stat + 2 : nullptr;
    A.deadline_ticks = (timeout_ms > 0 ? timeout_ms : 2000) * 100000LL;            // s_memrealtime: 100 MHz
    A.dbg_skip_plus1 = dbg_skip > 0 ? (unsigned long long)dbg_skip : 0ull;
    ProfScope prof(ctx, PROF_EW, (double)(Op::NIN + 1) * sizeof(T) * (double)n * (double)batch, 0.0);
    hipLaunchKernelGGL((ew_res_prepare<Op>), dim3((unsigned)((A.total + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, op, A, ops);
    const T* p0 = reinterpret_cast<const T*>(in0);
    const T* p1 = reinterpret_cast<const T*>(in1);
    if (damp)
        hipLaunchKernelGGL((ew_resident<T, Op, U, DL, DA, kResDamp | kResSchedBarrier | XF, BPC, 0, NT>), dim3(P), dim3(NT), 0, ctx->stream, (const Op*)ops, A, p0, p1);
    else
        hipLaunchKernelGGL((ew_resident<T, Op, U, DL, DA, kResSchedBarrier | XF, BPC, 0, NT>), dim3(P), dim3(NT), 0, ctx->stream, (const Op*)ops, A, p0, p1);
    TRAMP_LAUNCH_CHECK(ctx);
    ctx->resident_runs += 1;
    return TRAMP_OK;
}

#define RES_INSTANTIATE(T, ...)                                                                                     \
    template int run_resident<T, __VA_ARGS__>(tramp_ctx*, const __VA_ARGS__&, size_t, size_t, const void*, const void*, \
                                              const double*, const MsgOut&, const void*, size_t);
