// gemm_sliced.hip - host side of the int8-digit dense passes (gemm_sliced.h): digit copies of a resident matrix, attached
// by its device pointer; matvec_dispatch (linear.hip) takes them whenever the matrix of a pass has one.  Opt-in
// (tramp_matrix_attach_digits): the f64 / f32 matrix-core passes of gemm_mfma.hip stay the default engine.
// Replaces the same reference work as gemm_mfma.hip: the V / V^T (U / U^T) products of
// tramp/channels/linear/linear_channel.py:85-118 for an ensemble of instances.
#include "gemm_sliced.h"

#include <stdlib.h>

#include "linear.h"

struct SlicedMat {
    const void* key;      // device pointer of the source matrix [R, C]
    int dtype;
    size_t R, C;          // of the SOURCE matrix
    bool nt;              // the row-dot form (rows = R, contraction = C) or the column form (rows = C, contraction = R)
    int S;
    int8_t* frags;        // [ntile][nkb][S] KiB
    double* scale;        // [ntile 16]
    size_t rows, kdim, ntile, nkb, bytes;
    bool dead;            // the source matrix was released (or the copy detached) while a capture was open or a captured graph
                          // alive: never used again, memory reaped by gemm_sliced_reap once neither is the case
};

struct SlicedReg {
    SlicedMat* v;
    int n, cap;
};

static SlicedReg* reg_of(tramp_ctx* ctx, bool create) {
    if (!ctx->sliced_reg && create) ctx->sliced_reg = calloc(1, sizeof(SlicedReg));
    return (SlicedReg*)ctx->sliced_reg;
}

static const SlicedMat* find(tramp_ctx* ctx, const void* key, int dtype, size_t R, size_t C, bool nt) {
    SlicedReg* rg = reg_of(ctx, false);
    if (!rg) return nullptr;
    for (int i = 0; i < rg->n; ++i) {
        const SlicedMat& m = rg->v[i];
        if (!m.dead && m.key == key && m.dtype == dtype && m.R == R && m.C == C && m.nt == nt) return &m;
    }
    return nullptr;
}

// 2^31 / (S 2^14) contraction indices per accumulator, in whole k blocks
static size_t max_kb_per_split(int S) { return ((size_t)131072 / (size_t)S) / kSlK; }

template <typename T, int S>
static int slice_matrix(tramp_ctx* ctx, SlicedMat& m, const T* mat) {
    const size_t rows_pad = m.ntile * 16;
    unsigned long long* amax = nullptr;
    TRAMP_HIP_CHECK(ctx, hipMalloc((void**)&amax, rows_pad * 8));
    TRAMP_HIP_CHECK(ctx, hipMemsetAsync(amax, 0, rows_pad * 8, ctx->stream));
    const size_t ld = m.C;
    const size_t nkg = m.nkb * 4;
    if (m.nt) {
        hipLaunchKernelGGL((sl_absmax_kernel<T, false>), dim3((unsigned)m.rows, 1), dim3(kBlock), 0, ctx->stream, m.rows, m.kdim, ld, mat, amax);
        const unsigned gy = (unsigned)(m.ntile < 32768 ? m.ntile : 32768), gz = (unsigned)((m.ntile + gy - 1) / gy);
        hipLaunchKernelGGL((sl_slice_kernel<T, S, false, false>), dim3((unsigned)((nkg + 15) / 16), gy, gz), dim3(kBlock), 0, ctx->stream,
                           m.rows, rows_pad, m.kdim, ld, mat, amax, m.nkb, m.frags, m.scale);
    } else {
        hipLaunchKernelGGL((sl_absmax_kernel<T, true>), dim3((unsigned)((m.rows + kBlock - 1) / kBlock), 64), dim3(kBlock), 0, ctx->stream, m.rows, m.kdim, ld, mat, amax);
        const unsigned gy = (unsigned)(nkg < 32768 ? nkg : 32768), gz = (unsigned)((nkg + gy - 1) / gy);
        hipLaunchKernelGGL((sl_slice_kernel<T, S, true, false>), dim3((unsigned)((rows_pad + kBlock - 1) / kBlock), gy, gz), dim3(kBlock), 0, ctx->stream,
                           m.rows, rows_pad, m.kdim, ld, mat, amax, m.nkb, m.frags, m.scale);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(amax);
    if (e != hipSuccess) return tramp_fail(ctx, TRAMP_EHIP, "tramp_matrix_attach_digits", hipGetErrorString(e));
    return TRAMP_OK;
}

template <typename T>
static int slice_matrix_s(tramp_ctx* ctx, SlicedMat& m, const T* mat) {
    switch (m.S) {
        case 4: return slice_matrix<T, 4>(ctx, m, mat);
        case 6: return slice_matrix<T, 6>(ctx, m, mat);
        case 7: return slice_matrix<T, 7>(ctx, m, mat);
        default: return slice_matrix<T, 8>(ctx, m, mat);
    }
}

static void drop(SlicedMat& m) {
    if (m.frags) (void)hipFree(m.frags);
    if (m.scale) (void)hipFree(m.scale);
    m.frags = nullptr;
    m.scale = nullptr;
}

// The copies of mat_d (nullptr: all) are taken out of use at once - no synchronisation, no free: callable while a stream
// capture is open (tramp_buf_free's deferred branch) or a captured graph that recorded a digit pass is alive.  An
// address re-used by a later allocation therefore never meets the old matrix's digits.
void gemm_sliced_mark_dead(tramp_ctx* ctx, const void* mat_d) {
    SlicedReg* rg = reg_of(ctx, false);
    if (!rg) return;
    for (int i = 0; i < rg->n; ++i)
        if (mat_d == nullptr || rg->v[i].key == mat_d) rg->v[i].dead = true;
}

// Free the dead copies; the caller has synchronised the stream.  Nothing happens while a capture is open or a captured
// graph is alive (its kernel nodes hold the fragment pointers).
void gemm_sliced_reap(tramp_ctx* ctx) {
    SlicedReg* rg = reg_of(ctx, false);
    if (!rg || ctx->capturing || ctx->graphs_alive > 0) return;
    int w = 0;
    for (int i = 0; i < rg->n; ++i) {
        if (rg->v[i].dead) drop(rg->v[i]);
        else rg->v[w++] = rg->v[i];
    }
    rg->n = w;
}

extern "C" int tramp_matrix_detach_digits(tramp_ctx* ctx, const void* mat_d) {
    if (!ctx) return TRAMP_EINVAL;
    SlicedReg* rg = reg_of(ctx, false);
    if (!rg) return TRAMP_OK;
    gemm_sliced_mark_dead(ctx, mat_d);
    if (ctx->capturing || ctx->graphs_alive > 0) return TRAMP_OK;       // reaped later (tramp_graph_destroy, tramp_buf_free, context end)
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    gemm_sliced_reap(ctx);
    return TRAMP_OK;
}

void gemm_sliced_ctx_destroy(tramp_ctx* ctx) {
    SlicedReg* rg = reg_of(ctx, false);
    if (!rg) return;
    for (int i = 0; i < rg->n; ++i) drop(rg->v[i]);
    free(rg->v);
    free(rg);
    ctx->sliced_reg = nullptr;
}

// digits: 0 = by dtype (f64: 7, f32: 4); which: bit 0 the row-dot form (tramp_matvec_nt), bit 1 the column form (_nn)
extern "C" int tramp_matrix_attach_digits(tramp_ctx* ctx, int dtype, size_t R, size_t C, const void* mat_d, int digits, int which) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, mat_d && R && C && (dtype == TRAMP_F64 || dtype == TRAMP_F32) && which >= 1 && which <= 3);
    TRAMP_REQUIRE(ctx, !ctx->capturing);
    if (digits == 0) digits = dtype == TRAMP_F64 ? 7 : 4;
    if (digits != 4 && digits != 6 && digits != 7 && digits != 8)
        return tramp_fail(ctx, TRAMP_EINVAL, "tramp_matrix_attach_digits", "digits must be 0 (by dtype), 4, 6, 7 or 8");
    (void)hipSetDevice(ctx->device);
    int rc = tramp_matrix_detach_digits(ctx, mat_d);
    if (rc) return rc;
    SlicedReg* rg = reg_of(ctx, true);
    if (!rg) return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_matrix_attach_digits", "registry");
    for (int form = 0; form < 2; ++form) {
        if (!(which & (1 << form))) continue;
        SlicedMat m{};
        m.key = mat_d; m.dtype = dtype; m.R = R; m.C = C; m.nt = (form == 0); m.S = digits;
        m.rows = m.nt ? R : C;
        m.kdim = m.nt ? C : R;
        m.ntile = (m.rows + 15) / 16;
        m.nkb = (m.kdim + kSlK - 1) / kSlK;
        if (m.nkb * 4 > 32768ull * 65535ull || m.ntile * 16 > 32768ull * 65535ull)
            return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_matrix_attach_digits", "matrix too large for the slicing launch");
        m.bytes = m.ntile * m.nkb * (size_t)m.S * kSlFrag;
        if (hipMalloc((void**)&m.frags, m.bytes) != hipSuccess || hipMalloc((void**)&m.scale, m.ntile * 16 * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            drop(m);
            (void)tramp_matrix_detach_digits(ctx, mat_d);
            return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_matrix_attach_digits", "hipMalloc of the digit copy failed");
        }
        rc = dtype == TRAMP_F64 ? slice_matrix_s<double>(ctx, m, (const double*)mat_d) : slice_matrix_s<float>(ctx, m, (const float*)mat_d);
        if (rc) {
            drop(m);
            (void)tramp_matrix_detach_digits(ctx, mat_d);          // not half a copy: the form already registered goes too
            return rc;
        }
        if (rg->n == rg->cap) {
            const int cap = rg->cap ? 2 * rg->cap : 8;
            SlicedMat* p = (SlicedMat*)realloc(rg->v, (size_t)cap * sizeof(SlicedMat));
            if (!p) {
                drop(m);
                return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_matrix_attach_digits", "registry");
            }
            rg->v = p;
            rg->cap = cap;
        }
        rg->v[rg->n++] = m;
    }
    return TRAMP_OK;
}

// digits of the copy attached to mat_d (0 = none), which forms it has (bit 0 row-dot, bit 1 column) and its device bytes
extern "C" int tramp_matrix_digits_info(tramp_ctx* ctx, const void* mat_d, int* digits, int* which, size_t* bytes) {
    if (!ctx) return TRAMP_EINVAL;
    int d = 0, w = 0;
    size_t b = 0;
    SlicedReg* rg = reg_of(ctx, false);
    if (rg)
        for (int i = 0; i < rg->n; ++i)
            if (!rg->v[i].dead && rg->v[i].key == mat_d) {
                d = rg->v[i].S;
                w |= rg->v[i].nt ? 1 : 2;
                b += rg->v[i].bytes + rg->v[i].ntile * 16 * sizeof(double);
            }
    if (digits) *digits = d;
    if (which) *which = w;
    if (bytes) *bytes = b;
    return TRAMP_OK;
}

template <typename T, int S>
static int run_sliced(tramp_ctx* ctx, const SlicedMat& m, size_t batch, const T* x, const LinEpi& epi) {
    const size_t len = m.rows;
    const size_t nchunk = (batch + 63) / 64, nbt = nchunk * 4;
    // (the fold that ends the pass launches dim3(len / 256, batch): batch itself is bounded by the grid's y limit)
    if (nchunk > 65535 || batch > 65535) return tramp_fail(ctx, TRAMP_ENOTSUP, "digit pass", "more than 65535 instances");
    // splits: a multiple of 8 (= XCDs, as gemm_mfma.hip) that keeps every int32 accumulator exact
    const size_t kmax = max_kb_per_split(S);
    size_t nsplit = 8;
    while ((m.nkb + nsplit - 1) / nsplit > kmax) nsplit += 8;
    if (nsplit > m.nkb) nsplit = m.nkb;
    const size_t kps = (m.nkb + nsplit - 1) / nsplit;
    if (kps > kmax) return tramp_fail(ctx, TRAMP_ENOTSUP, "digit pass", "contraction too long for exact int32 accumulation");
    const size_t part_bytes = ((nsplit * batch * len * sizeof(double) + 255) / 256) * 256;
    const size_t xfrag_bytes = m.nkb * nbt * (size_t)S * kSlFrag;
    const size_t amax_bytes = ((nbt * 16 * 8 + 255) / 256) * 256;
    int rc = tramp_ws_reserve(ctx, part_bytes + xfrag_bytes + 2 * amax_bytes);
    if (rc) return rc;
    char* ws = (char*)ctx->ws;
    double* partials = (double*)ws;
    int8_t* xfr = (int8_t*)(ws + part_bytes);
    unsigned long long* amax = (unsigned long long*)(ws + part_bytes + xfrag_bytes);
    double* xscale = (double*)(ws + part_bytes + xfrag_bytes + amax_bytes);
    TRAMP_HIP_CHECK(ctx, hipMemsetAsync(amax, 0, amax_bytes, ctx->stream));
    hipLaunchKernelGGL((sl_absmax_kernel<T, false>), dim3((unsigned)batch, 32), dim3(kBlock), 0, ctx->stream, batch, m.kdim, m.kdim, x, amax);
    const size_t nkg = m.nkb * 4;
    const unsigned gy = (unsigned)(nkg < 32768 ? nkg : 32768), gz = (unsigned)((nkg + gy - 1) / gy);
    hipLaunchKernelGGL((sl_slice_kernel<T, S, false, true>), dim3((unsigned)((nbt * 16 + kBlock - 1) / kBlock), gy, gz), dim3(kBlock), 0, ctx->stream,
                       batch, nbt * 16, m.kdim, m.kdim, x, amax, nbt, xfr, xscale);
    // waves per workgroup: 8 (one workgroup per CU) halve the x fragments a CU pulls through L2 per matrix byte -
    // 1.3-2.7 % on the power-bound pass with 7 digits and fewer (tools/probes/probe_gemm_sliced.hip), nothing with 8
    constexpr int NW = S == 8 ? 4 : 8;
    const size_t nrb = (m.ntile + NW - 1) / NW;
    if (nrb * nsplit > 0x7fffffffull) return tramp_fail(ctx, TRAMP_ENOTSUP, "digit pass", "grid");
    const size_t ntile = nrb * nsplit;
    const bool flat = nchunk > 1 && ntile % 8 == 0 && ntile * nchunk < (1ull << 31);        // sibling chunks interleaved
    const dim3 grid = flat ? dim3((unsigned)(ntile * nchunk), 1) : dim3((unsigned)ntile, (unsigned)nchunk);
    // instance tiles per workgroup: a single chunk of fewer than 49 instances skips the MFMAs of its empty tiles
    const int bt = nchunk > 1 ? 4 : (int)((batch + 15) / 16);
#define SL_LAUNCH(BTv)                                                                                                         \
    hipLaunchKernelGGL((gemm_sliced_kernel<S, BTv, 0, NW>), grid, dim3(NW * 64), 0, ctx->stream, len, m.nkb, batch, nbt,         \
                       (const int8_t*)m.frags, (const double*)m.scale, (const int8_t*)xfr, (const double*)xscale, kps,         \
                       (unsigned)nsplit, partials)
    switch (bt) {
        case 1: SL_LAUNCH(1); break;
        case 2: SL_LAUNCH(2); break;
        case 3: SL_LAUNCH(3); break;
        default: SL_LAUNCH(4); break;
    }
#undef SL_LAUNCH
    ctx->last_nsplit = (int)nsplit;
    ctx->last_nchunk = (int)nchunk;
    ctx->last_flat = flat ? 1 : 0;
    ctx->sliced_runs++;
    return mfma_fold(ctx, m.dtype, len, batch, (int)nsplit, partials, epi);
}

// Called by matvec_dispatch: runs the pass on the digit copy of mat_d if one is attached in the wanted form.
int gemm_sliced_try(tramp_ctx* ctx, int dtype, bool nt, size_t R, size_t C, size_t batch, const void* mat_d, const void* x_d,
                    const LinEpi& epi, bool* taken) {
    *taken = false;
    const SlicedMat* m = find(ctx, mat_d, dtype, R, C, nt);
    if (!m) return TRAMP_OK;
    *taken = true;
#define RUN(Tv, Sv) return run_sliced<Tv, Sv>(ctx, *m, batch, (const Tv*)x_d, epi)
    if (dtype == TRAMP_F64) {
        switch (m->S) {
            case 4: RUN(double, 4);
            case 6: RUN(double, 6);
            case 7: RUN(double, 7);
            default: RUN(double, 8);
        }
    }
    switch (m->S) {
        case 4: RUN(float, 4);
        case 6: RUN(float, 6);
        case 7: RUN(float, 7);
        default: RUN(float, 8);
    }
#undef RUN
}

extern "C" int tramp_sliced_stats(tramp_ctx* ctx, unsigned long long* runs) {
    if (!ctx) return TRAMP_EINVAL;
    if (runs) *runs = ctx->sliced_runs;
    return TRAMP_OK;
}
