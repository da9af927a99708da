// fft_pow2.hip - batched complex FFT of length h = 2^q in TWO passes over HBM (hand-written,
// for the convolution channel: tramp/channels/linear/conv_channel.py:81-109 needs three
// transforms per EP iteration and rocFFT spends ~5 passes on a 10^6-point real transform).
//
// Decimation in frequency, radix 4 (one closing radix-2 stage when the block length asks for
// it), in place, NO reordering: position p of the result holds frequency kfreq(p), the
// mixed-radix digit reversal of p.  Everything in the spectral domain of the convolution
// channel is elementwise (up to the k <-> h-k pairing of a real signal, which a table
// resolves), so the permuted order is simply kept, and the inverse - decimation in time, the
// same stages backwards - consumes it and returns natural order.
//   pass 1 (strided): the first S1 radix-4 stages couple positions that are multiples of the
//          block length LB apart.  A workgroup takes C consecutive offsets c and all G = 4^S1
//          rows g (position g LB + c): a [G][C] tile in LDS, S1 butterfly stages, twiddles
//          w_L^(m j) from a two-level table of w_h (exponent e: hi[e >> 10] lo[e & 1023]).
//   pass 2 (blocks):  the remaining stages stay inside contiguous blocks of LB points: one
//          block per workgroup in LDS.
// Each pass reads and writes every point once: 2 x (read + write) per transform.
// The inverse runs pass 2 then pass 1 with conjugated twiddles and the transposed butterfly.
#include <algorithm>

#include "common.h"
#include "fft_pow2.h"

namespace {

template <typename T> struct C2;
template <> struct C2<double> { using type = double2; };
template <> struct C2<float> { using type = float2; };

template <typename C> __device__ __forceinline__ C cmul(C a, C b) {
    C r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
template <typename C> __device__ __forceinline__ C cconj(C a) {
    a.y = -a.y;
    return a;
}
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { a.x += b.x; a.y += b.y; return a; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { a.x -= b.x; a.y -= b.y; return a; }
// multiplication by -i (forward) / +i (inverse)
template <typename C> __device__ __forceinline__ C mul_mi(C a) { C r; r.x = a.y; r.y = -a.x; return r; }
template <typename C> __device__ __forceinline__ C mul_pi(C a) { C r; r.x = -a.y; r.y = a.x; return r; }

// w_h^e from the two-level table (e < h)
template <typename C>
__device__ __forceinline__ C twiddle(const C* __restrict__ hi, const C* __restrict__ lo, size_t e) {
    return cmul(hi[e >> 10], lo[e & 1023]);
}

// One radix-4 butterfly.  INV = false: decimation in frequency (forward, outputs multiplied by
// w, w^2, w^3); INV = true: decimation in time with conjugated twiddles (inputs multiplied first).
template <typename C, bool INV>
__device__ __forceinline__ void bfly4(C& a0, C& a1, C& a2, C& a3, C w1) {
    const C w2 = cmul(w1, w1), w3 = cmul(w2, w1);
    if (!INV) {
        const C t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi(csub(a1, a3));
        a0 = cadd(t0, t2);
        a1 = cmul(cadd(t1, t3), w1);
        a2 = cmul(csub(t0, t2), w2);
        a3 = cmul(csub(t1, t3), w3);
    } else {
        const C u1 = cmul(a1, cconj(w1)), u2 = cmul(a2, cconj(w2)), u3 = cmul(a3, cconj(w3));
        const C t0 = cadd(a0, u2), t1 = csub(a0, u2), t2 = cadd(u1, u3), t3 = mul_pi(csub(u1, u3));
        a0 = cadd(t0, t2);
        a1 = cadd(t1, t3);
        a2 = csub(t0, t2);
        a3 = csub(t1, t3);
    }
}

// ---------------------------------------------------------------- pass 1: strided stages
// tile [G][C] (+1 complex of padding per row), G = 4^S1 rows LB apart, C offsets = one 128-byte
// line per row (8 f64 / 16 f32 complex): 35 KiB at G = 256, four workgroups per CU.
template <typename T> constexpr int c1_of() { return 128 / (2 * (int)sizeof(T)); }

template <typename T, bool INV>
__global__ __launch_bounds__(256) void fft_strided_kernel(int q, int lb_log, const typename C2<T>::type* __restrict__ in,
                                                          typename C2<T>::type* __restrict__ out,
                                                          const typename C2<T>::type* __restrict__ tw_hi,
                                                          const typename C2<T>::type* __restrict__ tw_lo) {
    using C = typename C2<T>::type;
    extern __shared__ unsigned char smem_raw[];
    C* tile = reinterpret_cast<C*>(smem_raw);
    const size_t h = (size_t)1 << q, LB = (size_t)1 << lb_log;
    const int G = (int)(h >> lb_log);                 // 4^S1
    const int S1 = (q - lb_log) / 2;
    constexpr int kC1 = c1_of<T>();
    constexpr int P = kC1 + 1;                        // row pitch
    const size_t tiles_per_tf = LB / kC1;
    const size_t tf = blockIdx.x / tiles_per_tf;      // transform (instance)
    const size_t c0 = (blockIdx.x % tiles_per_tf) * kC1;
    const C* src = in + tf * h;
    C* dst = out + tf * h;
    const int c = threadIdx.x % kC1, r0 = threadIdx.x / kC1;          // 16 row slots per sweep
    for (int g = r0; g < G; g += 256 / kC1) tile[g * P + c] = src[(size_t)g * LB + c0 + c];
    __syncthreads();
    const int nb = G / 4;                             // butterflies per column per stage
    for (int st = 0; st < S1; ++st) {
        const int s = INV ? S1 - 1 - st : st;         // DIT runs the stages backwards
        const int sg = G >> (2 * (s + 1));            // stride between the 4 legs, in rows: 4^(S1-s-1)
        const size_t mulh = (size_t)1 << (2 * s);     // w_L^m = w_h^(m 4^s), L = h / 4^s
        for (int b = r0; b < nb; b += 256 / kC1) {
            // butterfly b: block index bh = b / sg, offset bl = b % sg  ->  first leg row = bh * 4 sg + bl
            const int bl = b % sg, bh = b / sg;
            const int g0 = bh * 4 * sg + bl;
            const size_t m = (size_t)bl * LB + c0 + c;                 // position inside the quarter
            const C w1 = twiddle<C>(tw_hi, tw_lo, (m * mulh) & (h - 1));
            C a0 = tile[g0 * P + c], a1 = tile[(g0 + sg) * P + c], a2 = tile[(g0 + 2 * sg) * P + c],
              a3 = tile[(g0 + 3 * sg) * P + c];
            bfly4<C, INV>(a0, a1, a2, a3, w1);
            tile[g0 * P + c] = a0;
            tile[(g0 + sg) * P + c] = a1;
            tile[(g0 + 2 * sg) * P + c] = a2;
            tile[(g0 + 3 * sg) * P + c] = a3;
        }
        __syncthreads();
    }
    for (int g = r0; g < G; g += 256 / kC1) dst[(size_t)g * LB + c0 + c] = tile[g * P + c];
}

// ---------------------------------------------------------------- pass 2: inside blocks of LB
template <typename T, bool INV>
__global__ __launch_bounds__(256) void fft_block_kernel(int q, int lb_log, typename C2<T>::type* __restrict__ data,
                                                        const typename C2<T>::type* __restrict__ tw_hi,
                                                        const typename C2<T>::type* __restrict__ tw_lo) {
    using C = typename C2<T>::type;
    extern __shared__ unsigned char smem_raw[];
    C* tile = reinterpret_cast<C*>(smem_raw);
    const size_t h = (size_t)1 << q;
    const int LB = 1 << lb_log;
    C* blk = data + (size_t)blockIdx.x * LB;          // blocks of all transforms are contiguous
    for (int i = threadIdx.x; i < LB; i += 256) tile[i] = blk[i];
    __syncthreads();
    const int n4 = lb_log / 2;                        // radix-4 stages
    const bool has2 = lb_log & 1;                     // closing radix-2 stage (L = 2)
    const int nst = n4 + (has2 ? 1 : 0);
    for (int st = 0; st < nst; ++st) {
        const int s = INV ? nst - 1 - st : st;
        if (s < n4) {
            const int L = LB >> (2 * s), qd = L / 4;  // sub-length and quarter
            const size_t mulh = h / (size_t)L;        // w_L^m = w_h^(m h / L)
            for (int b = threadIdx.x; b < LB / 4; b += 256) {
                const int m = b % qd, base = (b / qd) * L + m;
                const C w1 = twiddle<C>(tw_hi, tw_lo, (size_t)m * mulh);
                C a0 = tile[base], a1 = tile[base + qd], a2 = tile[base + 2 * qd], a3 = tile[base + 3 * qd];
                bfly4<C, INV>(a0, a1, a2, a3, w1);
                tile[base] = a0;
                tile[base + qd] = a1;
                tile[base + 2 * qd] = a2;
                tile[base + 3 * qd] = a3;
            }
        } else {
            for (int b = threadIdx.x; b < LB / 2; b += 256) {
                const C a0 = tile[2 * b], a1 = tile[2 * b + 1];
                tile[2 * b] = cadd(a0, a1);
                tile[2 * b + 1] = csub(a0, a1);
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < LB; i += 256) blk[i] = tile[i];
}

template <typename T>
__global__ void fft_tables_kernel(int q, typename C2<T>::type* hi, typename C2<T>::type* lo) {
    // w_h^e = exp(-2 pi i e / h) evaluated in double with sincospi
    const size_t h = (size_t)1 << q;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nhi = h > 1024 ? h >> 10 : 1;
    if (i < nhi) {
        double s, c;
        sincospi(-2.0 * (double)(i << 10) / (double)h, &s, &c);
        hi[i].x = (T)c;
        hi[i].y = (T)s;
    }
    if (i < 1024) {
        double s, c;
        sincospi(-2.0 * (double)i / (double)h, &s, &c);
        lo[i].x = (T)c;
        lo[i].y = (T)s;
    }
}

}  // namespace

// frequency held by position p after the forward transform
size_t fft2_kfreq(int q, int lb_log, size_t p) {
    size_t k = 0, weight = 1;
    int bits = q;
    const int n4 = (q - lb_log) / 2 + lb_log / 2;
    for (int s = 0; s < n4; ++s) {                    // most significant base-4 digit of p first
        bits -= 2;
        k += ((p >> bits) & 3) * weight;
        weight *= 4;
    }
    if (bits == 1) k += (p & 1) * weight;
    return k;
}

int fft2_lb_log(int q) {
    // block length 2^lb with h / LB = 4^S1, 4 <= 4^S1 <= 256 (the [G][16] tile of pass 1 must fit
    // the LDS); 2^11 preferred (32 KiB f64 blocks: four workgroups per CU in pass 2)
    if (q < 12 || q > 21) return -1;
    const int order[4] = {11, 10, 12, 13};
    for (int i = 0; i < 4; ++i) {
        const int d = q - order[i];
        if (d >= 2 && d <= 8 && d % 2 == 0) return order[i];
    }
    return -1;
}

template <typename T>
static int plan_create_t(tramp_ctx* ctx, Fft2Plan* pl) {
    using C = typename C2<T>::type;
    const size_t h = (size_t)1 << pl->q;
    const size_t nhi = h >> 10;
    TRAMP_HIP_CHECK(ctx, hipMalloc(&pl->tw_hi, nhi * sizeof(C)));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&pl->tw_lo, 1024 * sizeof(C)));
    const unsigned nb = (unsigned)((std::max<size_t>(nhi, 1024) + 255) / 256);
    hipLaunchKernelGGL((fft_tables_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, pl->q, (C*)pl->tw_hi, (C*)pl->tw_lo);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

int fft2_plan_create(tramp_ctx* ctx, int dtype, int q, Fft2Plan** out) {
    const int lb = fft2_lb_log(q);
    if (lb < 0 || q > 30) return tramp_fail(ctx, TRAMP_ENOTSUP, "fft2_plan_create", "length");
    Fft2Plan* pl = new Fft2Plan();
    pl->dtype = dtype;
    pl->q = q;
    pl->lb_log = lb;
    pl->tw_hi = pl->tw_lo = nullptr;
    int rc = dtype == TRAMP_F64 ? plan_create_t<double>(ctx, pl) : plan_create_t<float>(ctx, pl);
    if (rc) {
        fft2_plan_destroy(pl);
        return rc;
    }
    *out = pl;
    return TRAMP_OK;
}

void fft2_plan_destroy(Fft2Plan* pl) {
    if (!pl) return;
    if (pl->tw_hi) (void)hipFree(pl->tw_hi);
    if (pl->tw_lo) (void)hipFree(pl->tw_lo);
    delete pl;
}

template <typename T>
static int fft2_run_t(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, bool inverse, const void* in, void* out) {
    using C = typename C2<T>::type;
    const int q = pl->q, lb = pl->lb_log;
    const size_t h = (size_t)1 << q, LB = (size_t)1 << lb;
    const int G = (int)(h >> lb);
    constexpr int kC1 = c1_of<T>();
    const size_t lds1 = (size_t)G * (kC1 + 1) * sizeof(C), lds2 = LB * sizeof(C);
    const unsigned g1 = (unsigned)(batch * (LB / kC1)), g2 = (unsigned)(batch * (h / LB));
    static bool attr_done[4] = {false, false, false, false};
    const int slot = (sizeof(T) == 8 ? 0 : 2) + (inverse ? 1 : 0);
    if (!attr_done[slot]) {
        const void* f1 = inverse ? reinterpret_cast<const void*>(fft_strided_kernel<T, true>)
                                 : reinterpret_cast<const void*>(fft_strided_kernel<T, false>);
        const void* f2 = inverse ? reinterpret_cast<const void*>(fft_block_kernel<T, true>)
                                 : reinterpret_cast<const void*>(fft_block_kernel<T, false>);
        (void)hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(f2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done[slot] = true;
    }
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024)
        return tramp_fail(ctx, TRAMP_ENOTSUP, "fft2", "transform too long for the LDS tile");
    if (!inverse) {
        hipLaunchKernelGGL((fft_strided_kernel<T, false>), dim3(g1), dim3(256), lds1, ctx->stream, q, lb,
                           (const C*)in, (C*)out, (const C*)pl->tw_hi, (const C*)pl->tw_lo);
        hipLaunchKernelGGL((fft_block_kernel<T, false>), dim3(g2), dim3(256), lds2, ctx->stream, q, lb, (C*)out,
                           (const C*)pl->tw_hi, (const C*)pl->tw_lo);
    } else {
        // `in` (the permuted spectrum) is overwritten by the block pass
        hipLaunchKernelGGL((fft_block_kernel<T, true>), dim3(g2), dim3(256), lds2, ctx->stream, q, lb,
                           (C*)const_cast<void*>(in), (const C*)pl->tw_hi, (const C*)pl->tw_lo);
        hipLaunchKernelGGL((fft_strided_kernel<T, true>), dim3(g1), dim3(256), lds1, ctx->stream, q, lb,
                           (const C*)in, (C*)out, (const C*)pl->tw_hi, (const C*)pl->tw_lo);
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

int fft2_forward(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, const void* in, void* out) {
    return pl->dtype == TRAMP_F64 ? fft2_run_t<double>(ctx, pl, batch, false, in, out)
                                  : fft2_run_t<float>(ctx, pl, batch, false, in, out);
}

int fft2_inverse(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, void* spectrum, void* out) {
    return pl->dtype == TRAMP_F64 ? fft2_run_t<double>(ctx, pl, batch, true, spectrum, out)
                                  : fft2_run_t<float>(ctx, pl, batch, true, spectrum, out);
}

// ---------------------------------------------------------------- exported for tests
extern "C" int tramp_fft_pow2(tramp_ctx* ctx, int dtype, int log2h, size_t batch, int inverse, void* in_d,
                              void* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, in_d && out_d && (dtype == TRAMP_F64 || dtype == TRAMP_F32));
    Fft2Plan* pl = nullptr;
    int rc = fft2_plan_create(ctx, dtype, log2h, &pl);
    if (rc) return rc;
    rc = inverse ? fft2_inverse(ctx, pl, batch, in_d, out_d) : fft2_forward(ctx, pl, batch, in_d, out_d);
    (void)hipStreamSynchronize(ctx->stream);
    fft2_plan_destroy(pl);
    return rc;
}

extern "C" int tramp_fft_pow2_order(tramp_ctx* ctx, int log2h, size_t count, unsigned long long* kfreq_h) {
    if (!ctx) return TRAMP_EINVAL;
    const int lb = fft2_lb_log(log2h);
    if (lb < 0) return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_fft_pow2_order", "length");
    TRAMP_REQUIRE(ctx, kfreq_h);
    for (size_t p = 0; p < count; ++p) kfreq_h[p] = fft2_kfreq(log2h, lb, p);
    return TRAMP_OK;
}

// Average time of `reps` forward and of `reps` inverse transforms (HIP events on the context's stream).
extern "C" int tramp_fft_pow2_bench(tramp_ctx* ctx, int dtype, int log2h, size_t batch, int reps, void* in_d,
                                    void* out_d, double* ms_fwd, double* ms_inv) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, in_d && out_d && ms_fwd && ms_inv && reps > 0);
    Fft2Plan* pl = nullptr;
    int rc = fft2_plan_create(ctx, dtype, log2h, &pl);
    if (rc) return rc;
    hipEvent_t e0, e1, e2;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    rc = fft2_forward(ctx, pl, batch, in_d, out_d);                      // warm-up
    (void)hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < reps && !rc; ++i) rc = fft2_forward(ctx, pl, batch, in_d, out_d);
    (void)hipEventRecord(e1, ctx->stream);
    for (int i = 0; i < reps && !rc; ++i) rc = fft2_inverse(ctx, pl, batch, out_d, in_d);
    (void)hipEventRecord(e2, ctx->stream);
    (void)hipEventSynchronize(e2);
    float a = 0, b = 0;
    (void)hipEventElapsedTime(&a, e0, e1);
    (void)hipEventElapsedTime(&b, e1, e2);
    *ms_fwd = a / reps;
    *ms_inv = b / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    fft2_plan_destroy(pl);
    return rc;
}
