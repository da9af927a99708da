// ew_resident_impl.h - definition of run_resident; included only by the ew_resident_*.hip translation units.
#pragma once
#include "ew_resident.h"
#include "ew_resident_host.h"

static inline size_t res_env(const char* name, size_t dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    const long v = atol(e);
    return v >= 0 ? (size_t)v : dflt;
}

template <typename T, typename Op>
int run_resident(tramp_ctx* ctx, const Op& op, size_t n, size_t batch, const void* in0, const void* in1,
                 const double* a_in_d, const MsgOut& out) {
    constexpr int VN = Vec<T>::N;
    constexpr int U = kResU, DL = kResDL, DA = kResDA, BPC = kResBlocksPerCU;
    constexpr size_t TV = (size_t)kBlock * U;
    static const size_t on = res_env("TRAMP_EW_RESIDENT", 1);
    if (!on || out.mask || out.r || (n % VN) != 0) return kResNotEligible;
    const size_t nv = n / VN;
    if (nv < TV || nv > 0x7fffffffull || batch > 0x3fffffffull) return kResNotEligible;
    // all workgroups must be resident at once: two per CU by the launch bounds, checked once per instantiation
    static int occ_plain = -1, occ_damp = -1;
    if (occ_plain < 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_plain, ew_resident<T, Op, U, DL, DA, 0, BPC>, kBlock, 0) != hipSuccess) occ_plain = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_damp, ew_resident<T, Op, U, DL, DA, kResDamp, BPC>, kBlock, 0) != hipSuccess) occ_damp = 0;
    }
    const bool damp = out.damping != 0.0;
    if ((damp ? occ_damp : occ_plain) < BPC || ctx->num_cu < 16) return kResNotEligible;
    const unsigned P = (unsigned)(BPC * ctx->num_cu), PS = P - kResCombiners;
    const size_t tiles = (nv + TV - 1) / TV;
    // an instance must fit the window of rounds the pipeline keeps in flight, and there must be rounds to overlap
    if (tiles > (size_t)PS * (DA - 2) + 1) return kResNotEligible;
    static const size_t min_rounds = res_env("TRAMP_EW_RESIDENT_MIN_ROUNDS", 4);
    if (batch * tiles < min_rounds * PS) return kResNotEligible;
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
    A.out = out;
    if (!damp) { A.out.a_old = nullptr; A.out.b_old = nullptr; }
    ProfScope prof(ctx, PROF_EW, (double)(Op::NIN + 1) * sizeof(T) * (double)n * (double)batch, 0.0);
    hipLaunchKernelGGL((ew_res_prepare<Op>), dim3((unsigned)((A.total + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, op, A, ops);
    const T* p0 = reinterpret_cast<const T*>(in0);
    const T* p1 = reinterpret_cast<const T*>(in1);
    if (damp)
        hipLaunchKernelGGL((ew_resident<T, Op, U, DL, DA, kResDamp, BPC>), dim3(P), dim3(kBlock), 0, ctx->stream, (const Op*)ops, A, p0, p1);
    else
        hipLaunchKernelGGL((ew_resident<T, Op, U, DL, DA, 0, BPC>), dim3(P), dim3(kBlock), 0, ctx->stream, (const Op*)ops, A, p0, p1);
    TRAMP_LAUNCH_CHECK(ctx);
    ctx->resident_runs += 1;
    return TRAMP_OK;
}

#define RES_INSTANTIATE(T, ...)                                                                                     \
    template int run_resident<T, __VA_ARGS__>(tramp_ctx*, const __VA_ARGS__&, size_t, size_t, const void*, const void*, \
                                              const double*, const MsgOut&);
