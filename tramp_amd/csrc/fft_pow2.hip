// fft_pow2.hip - batched complex FFT of length h = 2^a 5^b in TWO passes over HBM (hand-written,
// for the convolution channel: tramp/channels/linear/conv_channel.py:81-109 needs three
// transforms per EP iteration and rocFFT spends ~5 passes on a 10^6-point real transform).
// (The file keeps its first name; powers of two were the first lengths it handled.)
//
// Decimation in frequency, mixed radix 5 / 4 / 2, in place, NO reordering: position p of the
// result holds frequency kfreq(p), the mixed-radix digit reversal of p.  Everything in the
// spectral domain of the convolution channel is elementwise (up to the k <-> h-k pairing of a
// real signal, which a table resolves), so the permuted order is simply kept, and the inverse
// - decimation in time, the same stages backwards - consumes it and returns natural order.
//   pass 1 (strided): the first stages couple positions that are multiples of the block length
//          LB apart.  A workgroup takes C consecutive offsets c (one 128-byte line) and all
//          G = h / LB rows g (position g LB + c): a [G][C] tile in LDS, twiddles w_L^(m j) from
//          a two-level table of w_h (exponent e: hi[e >> 10] lo[e & 1023]).
//   pass 2 (blocks):  the remaining stages stay inside contiguous blocks of LB points: one
//          block per workgroup in LDS.
// Each pass reads and writes every point once: 2 x (read + write) per transform.
// The inverse runs pass 2 then pass 1 with conjugated twiddles and the transposed butterflies.
#include <algorithm>
#include <vector>

#include "common.h"
#include "fft_pow2.h"
#include "ew_resident_host.h"

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
template <typename C, typename S> __device__ __forceinline__ C cscale(C a, S s) { a.x *= s; a.y *= s; return a; }
// multiplication by -i / +i
template <typename C> __device__ __forceinline__ C mul_mi(C a) { C r; r.x = a.y; r.y = -a.x; return r; }
template <typename C> __device__ __forceinline__ C mul_pi(C a) { C r; r.x = -a.y; r.y = a.x; return r; }

// w_h^e from the two-level table (e < h)
template <typename C>
__device__ __forceinline__ C twiddle(const C* __restrict__ hi, const C* __restrict__ lo, size_t e) {
    return cmul(hi[e >> 10], lo[e & 1023]);
}

// Butterflies.  INV = false: decimation in frequency (forward DFT of the legs, outputs multiplied by
// w^j); INV = true: the exact transpose for decimation in time (inputs multiplied by conj(w)^j, then
// the conjugate DFT).  Unnormalised.
template <typename C, bool INV>
__device__ __forceinline__ void bfly2(C& a0, C& a1, C w1) {
    if (!INV) {
        const C s = cadd(a0, a1), d = csub(a0, a1);
        a0 = s;
        a1 = cmul(d, w1);
    } else {
        const C u1 = cmul(a1, cconj(w1));
        const C s = cadd(a0, u1), d = csub(a0, u1);
        a0 = s;
        a1 = d;
    }
}

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

template <typename C, bool INV>
__device__ __forceinline__ void bfly5(C& a0, C& a1, C& a2, C& a3, C& a4, C w1) {
    using S = decltype(a0.x);
    const S c1 = S(0.30901699437494742410), c2 = S(-0.80901699437494742410);   // cos(2 pi/5), cos(4 pi/5)
    const S s1 = S(0.95105651629515357212), s2 = S(0.58778525229247312917);    // sin(2 pi/5), sin(4 pi/5)
    const C w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
    if (INV) {
        a1 = cmul(a1, cconj(w1));
        a2 = cmul(a2, cconj(w2));
        a3 = cmul(a3, cconj(w3));
        a4 = cmul(a4, cconj(w4));
    }
    const C t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
    const C y0 = cadd(a0, cadd(t1, t2));
    const C m1 = cadd(a0, cadd(cscale(t1, c1), cscale(t2, c2)));
    const C m2 = cadd(a0, cadd(cscale(t1, c2), cscale(t2, c1)));
    const C n1 = cadd(cscale(t3, s1), cscale(t4, s2));
    const C n2 = csub(cscale(t3, s2), cscale(t4, s1));
    // forward: y1 = m1 - i n1, y4 = m1 + i n1, y2 = m2 - i n2, y3 = m2 + i n2; inverse: signs swapped
    const C in1 = INV ? mul_pi(n1) : mul_mi(n1), in2 = INV ? mul_pi(n2) : mul_mi(n2);
    C y1 = cadd(m1, in1), y4 = csub(m1, in1), y2 = cadd(m2, in2), y3 = csub(m2, in2);
    if (!INV) {
        y1 = cmul(y1, w1);
        y2 = cmul(y2, w2);
        y3 = cmul(y3, w3);
        y4 = cmul(y4, w4);
    }
    a0 = y0; a1 = y1; a2 = y2; a3 = y3; a4 = y4;
}

// one butterfly of radix r on tile[i0 + j * stride], j < r
template <typename C, bool INV>
__device__ __forceinline__ void bfly_at(C* tile, int i0, int stride, int r, C w1) {
    if (r == 4) {
        C a0 = tile[i0], a1 = tile[i0 + stride], a2 = tile[i0 + 2 * stride], a3 = tile[i0 + 3 * stride];
        bfly4<C, INV>(a0, a1, a2, a3, w1);
        tile[i0] = a0; tile[i0 + stride] = a1; tile[i0 + 2 * stride] = a2; tile[i0 + 3 * stride] = a3;
    } else if (r == 5) {
        C a0 = tile[i0], a1 = tile[i0 + stride], a2 = tile[i0 + 2 * stride], a3 = tile[i0 + 3 * stride],
          a4 = tile[i0 + 4 * stride];
        bfly5<C, INV>(a0, a1, a2, a3, a4, w1);
        tile[i0] = a0; tile[i0 + stride] = a1; tile[i0 + 2 * stride] = a2; tile[i0 + 3 * stride] = a3;
        tile[i0 + 4 * stride] = a4;
    } else {
        C a0 = tile[i0], a1 = tile[i0 + stride];
        bfly2<C, INV>(a0, a1, w1);
        tile[i0] = a0; tile[i0 + stride] = a1;
    }
}

#ifndef FFT_LINE
#define FFT_LINE 128
#endif
// tile [G][C] (+1 complex of padding per row), G rows LB apart, C offsets = one 128-byte line per
// row (8 f64 / 16 f32 complex)
template <typename T> constexpr int c1_of() { return FFT_LINE / (2 * (int)sizeof(T)); }

// ---------------------------------------------------------------- pass 1: strided stages
template <typename T, bool INV, bool EPI = false>
__global__ __launch_bounds__(1024) void fft_strided_kernel(Fft2Shape sh, const typename C2<T>::type* __restrict__ in,
                                                          typename C2<T>::type* __restrict__ out,
                                                          const typename C2<T>::type* __restrict__ tw_hi,
                                                          const typename C2<T>::type* __restrict__ tw_lo,
                                                          Fft2Epi epi = Fft2Epi{}) {
    using C = typename C2<T>::type;
    extern __shared__ unsigned char smem_raw[];
    C* tile = reinterpret_cast<C*>(smem_raw);
    const size_t h = sh.h, LB = sh.LB;
    const int G = sh.G;
    constexpr int kC1 = c1_of<T>();
    constexpr int P = kC1 + 1;                        // row pitch
    const size_t tiles_per_tf = LB / kC1;
    const size_t tf = blockIdx.x / tiles_per_tf;      // transform (instance)
    const size_t c0 = (blockIdx.x % tiles_per_tf) * kC1;
    const C* src = in + tf * h;
    C* dst = out + tf * h;
    const int c = threadIdx.x % kC1, r0 = threadIdx.x / kC1;
    const int rstep = blockDim.x / kC1;               // rows (or butterflies) per sweep of the workgroup
    if constexpr (sizeof(T) == 4) {
        // f32: two complex (16 bytes) per lane on the global side, as in f64
        const int c2 = threadIdx.x % (kC1 / 2), q0 = threadIdx.x / (kC1 / 2);
        for (int g = q0; g < G; g += 2 * rstep) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)g * LB + c0 + 2 * c2);
            tile[g * P + 2 * c2] = C{v.x, v.y};
            tile[g * P + 2 * c2 + 1] = C{v.z, v.w};
        }
    } else {
        for (int g = r0; g < G; g += rstep) tile[g * P + c] = src[(size_t)g * LB + c0 + c];
    }
    // this tile's twiddles (one per stage, leg offset and column) go to LDS while the tile loads are in
    // flight: the stage loops then never wait on global memory
    C* twl = tile + (size_t)G * P;
    for (int s = 0; s < sh.n1; ++s) {
        const size_t mulh = sh.mul1[s];               // w_L^m = w_h^(m * prod of the earlier radices)
        for (int bl = r0; bl < sh.sub1[s]; bl += rstep) {
            const size_t m = (size_t)bl * LB + c0 + c;                 // position inside the sub-block; m mulh < h
            twl[(sh.off1[s] + bl) * kC1 + c] = twiddle<C>(tw_hi, tw_lo, m * mulh);
        }
    }
    __syncthreads();
    for (int st = 0; st < sh.n1; ++st) {
        const int s = INV ? sh.n1 - 1 - st : st;      // DIT runs the stages backwards
        const int r = sh.r1[s];
        const int sg = sh.sub1[s];                    // stride between the legs, in rows
        const int nb = G / r;                         // butterflies per column
        const C* tws = twl + sh.off1[s] * kC1;
        for (int b = r0; b < nb; b += rstep) {
            const int bl = b % sg, bh = b / sg;
            const int g0 = bh * r * sg + bl;
            bfly_at<C, INV>(tile, g0 * P + c, sg * P, r, tws[bl * kC1 + c]);
        }
        __syncthreads();
    }
    if constexpr (EPI) {
        // message update on the way out: the complex value is the pair (r[2 i], r[2 i + 1]) of the mean
        const size_t inst = epi.rows_per_inst > 1 ? tf / epi.rows_per_inst : tf;
        const T asum = (T)(epi.a_in[inst] + epi.a_raw[inst]);
        const T d = (T)epi.damping, omd = (T)(1.0 - epi.damping);
        const C* bin = reinterpret_cast<const C*>(epi.b_in) + tf * h;
        const C* bold = epi.damping != 0.0 ? reinterpret_cast<const C*>(epi.b_old) + tf * h : nullptr;
        C* bo = reinterpret_cast<C*>(epi.out) + tf * h;
        C* ro = epi.r_out ? reinterpret_cast<C*>(epi.r_out) + tf * h : nullptr;
        double vi = 0.0;
        PriorHintEval<T> he;
        if (epi.hint.kind) he.setup(epi.hint, tf);
        for (int g = r0; g < G; g += rstep) {
            const size_t i = (size_t)g * LB + c0 + c;
            const C r = tile[g * P + c], bi = bin[i];
            C bn = {r.x * asum - bi.x, r.y * asum - bi.y};
            if (bold) {
                const C bp = bold[i];
                bn.x = d * bp.x + omd * bn.x;
                bn.y = d * bp.y + omd * bn.y;
            }
            bo[i] = bn;
            if (ro) ro[i] = r;
            if (epi.hint.kind) vi += he.v(bn.x, 2 * i) + he.v(bn.y, 2 * i + 1);
        }
        if (epi.hint.kind) {
            // sum over the workgroup (up to 16 waves; the tile's LDS is free again after the barrier)
            __syncthreads();
            double* red = reinterpret_cast<double*>(smem_raw);
            vi = wave_sum(vi);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vi;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t += red[w];
                epi.hint.partials[tf * (size_t)epi.hint.nblk + blockIdx.x % tiles_per_tf] = t;
            }
        }
        return;
    }
    if constexpr (sizeof(T) == 4) {
        const int c2 = threadIdx.x % (kC1 / 2), q0 = threadIdx.x / (kC1 / 2);
        for (int g = q0; g < G; g += 2 * rstep) {
            const C a = tile[g * P + 2 * c2], b = tile[g * P + 2 * c2 + 1];
            *reinterpret_cast<float4*>(dst + (size_t)g * LB + c0 + 2 * c2) = float4{a.x, a.y, b.x, b.y};
        }
    } else {
        for (int g = r0; g < G; g += rstep) dst[(size_t)g * LB + c0 + c] = tile[g * P + c];
    }
}

// ---------------------------------------------------------------- pass 2: inside blocks of LB
// A workgroup takes tile_pts consecutive points (a whole number of blocks of LB: the stage structure repeats with
// period L <= LB, so several short blocks share one tile); in and out may coincide.
template <typename T, bool INV, bool EPI = false>
__global__ __launch_bounds__(1024) void fft_block_kernel(Fft2Shape sh, const typename C2<T>::type* __restrict__ in,
                                                        typename C2<T>::type* __restrict__ out,
                                                        const typename C2<T>::type* __restrict__ tw_blk, int tile_pts,
                                                        size_t total_pts, Fft2Epi epi = Fft2Epi{},
                                                        const int* __restrict__ pos_of_freq = nullptr, int nat_mode = 0) {
    using C = typename C2<T>::type;
    extern __shared__ unsigned char smem_raw[];
    C* tile = reinterpret_cast<C*>(smem_raw);
    const int nthr = blockDim.x;                      // 64 ... 512 threads per block (TRAMP_FFT_BLOCK_THREADS)
    const size_t first = (size_t)blockIdx.x * tile_pts;
    const int npts = (int)((total_pts - first < (size_t)tile_pts) ? total_pts - first : (size_t)tile_pts);
    const C* blk = in + first;                        // blocks of all transforms are contiguous
    C* dst = out + first;
    const int LBi = (int)sh.LB;
    if (nat_mode == 2) {
        // natural-order spectrum in: element k of a row goes to the position the inverse stages expect (the permutation
        // happens in LDS, the global reads stay contiguous)
        for (int i = threadIdx.x; i < npts; i += nthr) tile[(i / LBi) * LBi + pos_of_freq[i % LBi]] = blk[i];
    } else if constexpr (sizeof(T) == 4) {
        for (int i = threadIdx.x; i < npts / 2; i += nthr)
            reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(blk)[i];
    } else {
        for (int i = threadIdx.x; i < npts; i += nthr) tile[i] = blk[i];
    }
    // the stage twiddles are the same for every block: a small table, copied to LDS beside the tile
    C* twl = tile + tile_pts;
    for (int i = threadIdx.x; i < sh.ntw2; i += nthr) twl[i] = tw_blk[i];
    __syncthreads();
    for (int st = 0; st < sh.n2; ++st) {
        const int s = INV ? sh.n2 - 1 - st : st;
        const int r = sh.r2[s];
        const int sub = sh.sub2[s];                   // L / r, L the current sub-length
        const int L = sub * r;
        const C* tws = twl + sh.off2[s];
        for (int b = threadIdx.x; b < npts / r; b += nthr) {
            const int m = b % sub, base = (b / sub) * L + m;
            bfly_at<C, INV>(tile, base, sub, r, tws[m]);
        }
        __syncthreads();
    }
    if constexpr (EPI) {
        // message update on the way out (block-only inverse of the N-D plans): the complex value is the pair
        // (r[2 i], r[2 i + 1]) of the mean; rows of LB points, rows_per_inst of them per problem instance
        const T d = (T)epi.damping, omd = (T)(1.0 - epi.damping);
        const C* bin = reinterpret_cast<const C*>(epi.b_in) + first;
        const C* bold = epi.damping != 0.0 ? reinterpret_cast<const C*>(epi.b_old) + first : nullptr;
        C* bo = reinterpret_cast<C*>(epi.out) + first;
        C* ro = epi.r_out ? reinterpret_cast<C*>(epi.r_out) + first : nullptr;
        const size_t pts_per_inst = (epi.rows_per_inst ? epi.rows_per_inst : 1) * sh.LB;
        // producer-fused moments (prior_hint.h): armed only when a tile lies inside ONE instance (fft2_run_t checks)
        double vi = 0.0;
        PriorHintEval<T> he;
        const size_t inst0 = first / pts_per_inst;
        if (epi.hint.kind) he.setup(epi.hint, inst0);
        for (int i = threadIdx.x; i < npts; i += nthr) {
            const size_t inst = (first + i) / pts_per_inst;
            const T asum = (T)(epi.a_in[inst] + epi.a_raw[inst]);
            const C r = tile[i], bi = bin[i];
            C bn = {r.x * asum - bi.x, r.y * asum - bi.y};
            if (bold) {
                const C bp = bold[i];
                bn.x = d * bp.x + omd * bn.x;
                bn.y = d * bp.y + omd * bn.y;
            }
            bo[i] = bn;
            if (ro) ro[i] = r;
            if (epi.hint.kind) {
                const size_t idx = 2 * (first + i - inst0 * pts_per_inst);
                vi += he.v(bn.x, idx) + he.v(bn.y, idx + 1);
            }
        }
        if (epi.hint.kind) {
            __syncthreads();
            double* red = reinterpret_cast<double*>(smem_raw);
            vi = wave_sum(vi);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vi;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t += red[w];
                epi.hint.partials[inst0 * (size_t)epi.hint.nblk + (first - inst0 * pts_per_inst) / tile_pts] = t;
            }
        }
        return;
    }
    if (nat_mode == 1) {
        for (int i = threadIdx.x; i < npts; i += nthr) dst[i] = tile[(i / LBi) * LBi + pos_of_freq[i % LBi]];
    } else if constexpr (sizeof(T) == 4) {
        for (int i = threadIdx.x; i < npts / 2; i += nthr)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(tile)[i];
    } else {
        for (int i = threadIdx.x; i < npts; i += nthr) dst[i] = tile[i];
    }
}

// ---------------------------------------------------------------- a whole small image in one workgroup (2-D)
// [n0][hl] complex points of one packed image fit one LDS tile: the row stages (as the block pass, rows are contiguous) and
// the column stages (as the axis pass: lanes run along a row, legs sg rows apart) in ONE launch - small images are bound by
// launches, not by memory.  Forward: rows, then columns; inverse: columns, then rows, then the optional message update.
template <typename T, bool INV, bool EPI = false>
__global__ __launch_bounds__(1024) void fft_image_kernel(Fft2Shape sh, FftAxisPass ps,
                                                        const typename C2<T>::type* __restrict__ in,
                                                        typename C2<T>::type* __restrict__ out,
                                                        const typename C2<T>::type* __restrict__ tw_blk,
                                                        const typename C2<T>::type* __restrict__ tw_hi,
                                                        const typename C2<T>::type* __restrict__ tw_lo,
                                                        Fft2Epi epi = Fft2Epi{}) {
    using C = typename C2<T>::type;
    extern __shared__ unsigned char smem_raw[];
    C* tile = reinterpret_cast<C*>(smem_raw);
    const int nthr = blockDim.x;
    const int hl = (int)sh.LB, G = ps.G, npts = G * hl;
    const size_t first = (size_t)blockIdx.x * npts;
    const C* src = in + first;
    C* dst = out + first;
    if constexpr (sizeof(T) == 4) {
        for (int i = threadIdx.x; i < npts / 2; i += nthr)
            reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < npts; i += nthr) tile[i] = src[i];
    }
    C* twr = tile + npts;                             // row stages: the block pass's table
    C* twc = twr + sh.ntw2;                           // column stages: w_n0^(bl mul[s])
    for (int i = threadIdx.x; i < sh.ntw2; i += nthr) twr[i] = tw_blk[i];
    for (int s = 0; s < ps.nst; ++s)
        for (int bl = threadIdx.x; bl < ps.sub[s]; bl += nthr)
            twc[ps.off[s] + bl] = twiddle<C>(tw_hi, tw_lo, (size_t)bl * ps.mul[s]);
    __syncthreads();
    auto rows = [&]() {
        for (int st = 0; st < sh.n2; ++st) {
            const int s = INV ? sh.n2 - 1 - st : st;
            const int r = sh.r2[s], sub = sh.sub2[s], L = sub * r;
            const C* tws = twr + sh.off2[s];
            for (int b = threadIdx.x; b < npts / r; b += nthr) {
                const int m = b % sub;
                bfly_at<C, INV>(tile, (b / sub) * L + m, sub, r, tws[m]);
            }
            __syncthreads();
        }
    };
    auto cols = [&]() {
        for (int st = 0; st < ps.nst; ++st) {
            const int s = INV ? ps.nst - 1 - st : st;
            const int r = ps.r[s], sg = ps.sub[s], nb = G / r;
            const C* tws = twc + ps.off[s];
            for (int idx = threadIdx.x; idx < nb * hl; idx += nthr) {
                const int c = idx % hl, b = idx / hl;
                const int bl = b % sg, bh = b / sg;
                bfly_at<C, INV>(tile, (bh * r * sg + bl) * hl + c, sg * hl, r, tws[bl]);
            }
            __syncthreads();
        }
    };
    if (!INV) {
        rows();
        cols();
    } else {
        cols();
        rows();
    }
    if constexpr (EPI) {
        const T d = (T)epi.damping, omd = (T)(1.0 - epi.damping);
        const C* bin = reinterpret_cast<const C*>(epi.b_in) + first;
        const C* bold = epi.damping != 0.0 ? reinterpret_cast<const C*>(epi.b_old) + first : nullptr;
        C* bo = reinterpret_cast<C*>(epi.out) + first;
        C* ro = epi.r_out ? reinterpret_cast<C*>(epi.r_out) + first : nullptr;
        const size_t inst = blockIdx.x / (epi.rows_per_inst ? epi.rows_per_inst : 1);   // here: images per instance
        const T asum = (T)(epi.a_in[inst] + epi.a_raw[inst]);
        for (int i = threadIdx.x; i < npts; i += nthr) {
            const C r = tile[i], bi = bin[i];
            C bn = {r.x * asum - bi.x, r.y * asum - bi.y};
            if (bold) {
                const C bp = bold[i];
                bn.x = d * bp.x + omd * bn.x;
                bn.y = d * bp.y + omd * bn.y;
            }
            bo[i] = bn;
            if (ro) ro[i] = r;
        }
        return;
    }
    if constexpr (sizeof(T) == 4) {
        for (int i = threadIdx.x; i < npts / 2; i += nthr)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(tile)[i];
    } else {
        for (int i = threadIdx.x; i < npts; i += nthr) dst[i] = tile[i];
    }
}

// ---------------------------------------------------------------- strided pass along an outer axis (N-D)
// Tile [G][C]: G rows of the axis (Q * W elements apart), C consecutive elements of the inner axes (one 128-byte
// line per row).  The twiddles depend on the row only, not on the column.
template <typename T, bool INV>
__global__ __launch_bounds__(1024) void fft_axis_kernel(FftAxisPass ps, const typename C2<T>::type* __restrict__ in,
                                                       typename C2<T>::type* __restrict__ out,
                                                       const typename C2<T>::type* __restrict__ tw_hi,
                                                       const typename C2<T>::type* __restrict__ tw_lo,
                                                       const int* __restrict__ rowmap = nullptr, int map_mode = 0) {
    using C = typename C2<T>::type;
    extern __shared__ unsigned char smem_raw[];
    C* tile = reinterpret_cast<C*>(smem_raw);
    constexpr int kC1 = c1_of<T>();
    constexpr int P = kC1 + 1;
    const int G = ps.G;
    const size_t ncg = ps.W / kC1;
    size_t t = blockIdx.x;
    const size_t cg = t % ncg;
    t /= ncg;
    const int q = (int)(t % ps.Q);
    t /= ps.Q;
    const size_t a = t % ps.nA, o = t / ps.nA;
    // tile row g is row a G Q + g Q + q of the axis; natural-order mode (rowmap: position -> frequency) reads (2) or
    // writes (1) that row at the place of its frequency instead - whole 128-byte lines either way
    const size_t row0 = a * (size_t)G * ps.Q + q;
    const size_t col0 = o * ps.n * ps.W + cg * kC1;
    auto row_in = [&](int g) -> size_t {
        const size_t r = row0 + (size_t)g * ps.Q;
        return col0 + (size_t)(map_mode == 2 ? (size_t)rowmap[r] : r) * ps.W;
    };
    auto row_out = [&](int g) -> size_t {
        const size_t r = row0 + (size_t)g * ps.Q;
        return col0 + (size_t)(map_mode == 1 ? (size_t)rowmap[r] : r) * ps.W;
    };
    const int c = threadIdx.x % kC1, r0 = threadIdx.x / kC1;
    const int rstep = blockDim.x / kC1;
    if constexpr (sizeof(T) == 4) {
        const int c2 = threadIdx.x % (kC1 / 2), q0 = threadIdx.x / (kC1 / 2);
        for (int g = q0; g < G; g += 2 * rstep) {
            const float4 v = *reinterpret_cast<const float4*>(in + row_in(g) + 2 * c2);
            tile[g * P + 2 * c2] = C{v.x, v.y};
            tile[g * P + 2 * c2 + 1] = C{v.z, v.w};
        }
    } else {
        for (int g = r0; g < G; g += rstep) tile[g * P + c] = in[row_in(g) + c];
    }
    C* twl = tile + (size_t)G * P;
    for (int s = 0; s < ps.nst; ++s)
        for (int bl = threadIdx.x; bl < ps.sub[s]; bl += blockDim.x)
            twl[ps.off[s] + bl] = twiddle<C>(tw_hi, tw_lo, ((size_t)bl * ps.Q + q) * ps.mul[s]);
    __syncthreads();
    for (int st = 0; st < ps.nst; ++st) {
        const int s = INV ? ps.nst - 1 - st : st;
        const int r = ps.r[s];
        const int sg = ps.sub[s];
        const int nb = G / r;
        const C* tws = twl + ps.off[s];
        for (int b = r0; b < nb; b += rstep) {
            const int bl = b % sg, bh = b / sg;
            bfly_at<C, INV>(tile, (bh * r * sg + bl) * P + c, sg * P, r, tws[bl]);
        }
        __syncthreads();
    }
    if constexpr (sizeof(T) == 4) {
        const int c2 = threadIdx.x % (kC1 / 2), q0 = threadIdx.x / (kC1 / 2);
        for (int g = q0; g < G; g += 2 * rstep) {
            const C x = tile[g * P + 2 * c2], y = tile[g * P + 2 * c2 + 1];
            *reinterpret_cast<float4*>(out + row_out(g) + 2 * c2) = float4{x.x, x.y, y.x, y.y};
        }
    } else {
        for (int g = r0; g < G; g += rstep) out[row_out(g) + c] = tile[g * P + c];
    }
}

template <typename T>
__global__ void fft_blk_table_kernel(Fft2Shape sh, typename C2<T>::type* tw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sh.ntw2) return;
    int s = 0;
    while (s + 1 < sh.n2 && i >= sh.off2[s + 1]) ++s;
    const size_t e = (size_t)(i - sh.off2[s]) * sh.mul2[s];
    double sn, cs;
    sincospi(-2.0 * (double)e / (double)sh.h, &sn, &cs);
    tw[i].x = (T)cs;
    tw[i].y = (T)sn;
}

template <typename T>
__global__ void fft_tables_kernel(size_t h, size_t nhi, typename C2<T>::type* hi, typename C2<T>::type* lo) {
    // w_h^e = exp(-2 pi i e / h) evaluated in double with sincospi
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nhi) {
        double s, c;
        sincospi(-2.0 * (double)((i << 10) % h) / (double)h, &s, &c);
        hi[i].x = (T)c;
        hi[i].y = (T)s;
    }
    if (i < 1024) {
        double s, c;
        sincospi(-2.0 * (double)(i % h) / (double)h, &s, &c);
        lo[i].x = (T)c;
        lo[i].y = (T)s;
    }
}

// radices of a 2^a 5^b number, fives first, then fours, then a closing two
int radices_of(size_t v, int* r) {
    int n = 0;
    while (v % 5 == 0) { r[n++] = 5; v /= 5; }
    while (v % 4 == 0) { r[n++] = 4; v /= 4; }
    if (v % 2 == 0) { r[n++] = 2; v /= 2; }
    return v == 1 ? n : -1;
}

}  // namespace

// Split h = G * LB: LB (second pass, one block per workgroup) as close to 2048 as the factors
// allow within [512, 4096], G (rows of the first pass's tile) within [2, 320].
bool fft2_shape(size_t h, Fft2Shape* sh) {
    size_t v = h;
    int a = 0, b = 0;
    while (v && v % 2 == 0) { v /= 2; ++a; }
    while (v && v % 5 == 0) { v /= 5; ++b; }
    if (v != 1 || h < 1024) return false;
    size_t best = 0;
    double best_cost = 1e300;
    for (int a1 = 0; a1 <= a; ++a1) {
        size_t lb = (size_t)1 << a1;
        for (int b1 = 0; b1 <= b; ++b1, lb *= 5) {
            if (lb < 512 || lb > 4096 || lb % 16) continue;
            if (h % lb) continue;
            const size_t g = h / lb;
            if (g < 2 || g > 320) continue;
            const double cost = fabs(log((double)lb / 2048.0));
            if (cost < best_cost) { best_cost = cost; best = lb; }
        }
    }
    if (!best) return false;
    sh->h = h;
    sh->LB = best;
    sh->G = (int)(h / best);
    sh->n1 = radices_of((size_t)sh->G, sh->r1);
    sh->n2 = radices_of(best, sh->r2);
    if (sh->n1 < 0 || sh->n2 < 0 || sh->n1 > 8 || sh->n2 > 8) return false;
    size_t prod = 1;                                  // product of the radices of the earlier stages
    int rows = sh->G;
    for (int s = 0; s < sh->n1; ++s) {
        rows /= sh->r1[s];
        sh->sub1[s] = rows;
        sh->mul1[s] = prod;
        prod *= sh->r1[s];
    }
    size_t L = best;
    for (int s = 0; s < sh->n2; ++s) {
        sh->sub2[s] = (int)(L / sh->r2[s]);
        sh->mul2[s] = h / L;
        L /= sh->r2[s];
    }
    sh->ntw1 = sh->ntw2 = 0;
    for (int s = 0; s < sh->n1; ++s) { sh->off1[s] = sh->ntw1; sh->ntw1 += sh->sub1[s]; }
    for (int s = 0; s < sh->n2; ++s) { sh->off2[s] = sh->ntw2; sh->ntw2 += sh->sub2[s]; }
    return true;
}

// One block pass only: h <= 4096 points stay in a single LDS tile (G = 1, no strided stages).
bool fft2_block_shape(size_t h, Fft2Shape* sh) {
    if (h < 2 || h > 4096 || (h & 1)) return false;
    memset(sh, 0, sizeof(*sh));
    sh->h = h;
    sh->LB = h;
    sh->G = 1;
    sh->n1 = 0;
    sh->n2 = radices_of(h, sh->r2);
    if (sh->n2 < 1 || sh->n2 > 8) return false;
    size_t L = h;
    for (int s = 0; s < sh->n2; ++s) {
        sh->sub2[s] = (int)(L / sh->r2[s]);
        sh->mul2[s] = h / L;
        L /= sh->r2[s];
    }
    sh->ntw1 = sh->ntw2 = 0;
    for (int s = 0; s < sh->n2; ++s) { sh->off2[s] = sh->ntw2; sh->ntw2 += sh->sub2[s]; }
    return true;
}

// frequency held by position p after the forward transform: mixed-radix digit reversal
size_t fft2_kfreq(const Fft2Shape& sh, size_t p) {
    size_t k = 0, weight = 1, len = sh.h;
    for (int s = 0; s < sh.n1 + sh.n2; ++s) {
        const int r = s < sh.n1 ? sh.r1[s] : sh.r2[s - sh.n1];
        len /= r;
        k += (p / len) * weight;                      // most significant remaining digit of p
        p %= len;
        weight *= r;
    }
    return k;
}

template <typename T>
static int plan_create_t(tramp_ctx* ctx, Fft2Plan* pl) {
    using C = typename C2<T>::type;
    const size_t h = pl->shape.h;
    const size_t nhi = (h + 1023) >> 10;
    TRAMP_HIP_CHECK(ctx, hipMalloc(&pl->tw_hi, nhi * sizeof(C)));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&pl->tw_lo, 1024 * sizeof(C)));
    const unsigned nb = (unsigned)((std::max<size_t>(nhi, 1024) + 255) / 256);
    hipLaunchKernelGGL((fft_tables_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, h, nhi, (C*)pl->tw_hi,
                       (C*)pl->tw_lo);
    TRAMP_HIP_CHECK(ctx, hipMalloc(&pl->tw_blk, (size_t)pl->shape.ntw2 * sizeof(C)));
    hipLaunchKernelGGL((fft_blk_table_kernel<T>), dim3((pl->shape.ntw2 + 255) / 256), dim3(256), 0, ctx->stream,
                       pl->shape, (C*)pl->tw_blk);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

int fft2_plan_create(tramp_ctx* ctx, int dtype, size_t h, Fft2Plan** out, bool block_only) {
    Fft2Plan* pl = new Fft2Plan();
    pl->dtype = dtype;
    pl->tw_hi = pl->tw_lo = pl->tw_blk = nullptr;
    if (!(block_only ? fft2_block_shape(h, &pl->shape) : fft2_shape(h, &pl->shape))) {
        delete pl;
        return tramp_fail(ctx, TRAMP_ENOTSUP, "fft2_plan_create", "length is not 2^a 5^b within the two-pass range");
    }
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
    if (pl->tw_blk) (void)hipFree(pl->tw_blk);
    delete pl;
}

template <typename T>
static int fft2_run_t(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, bool inverse, const void* in, void* out,
                      const Fft2Epi* epi = nullptr) {
    using C = typename C2<T>::type;
    const Fft2Shape& sh = pl->shape;
    constexpr int kC1 = c1_of<T>();
    const size_t lds1 = ((size_t)sh.G * (kC1 + 1) + (size_t)sh.ntw1 * kC1) * sizeof(C);
    // block pass: one block of LB per workgroup behind the strided pass; alone (G == 1) as many short blocks as
    // make a tile of ~2048 points
    const size_t total_pts = batch * sh.h;
    int tile_pts = (int)sh.LB;
    if (sh.G == 1 && sh.LB < 2048) tile_pts = (int)(sh.LB * (2048 / sh.LB));
    const size_t lds2 = ((size_t)tile_pts + (size_t)sh.ntw2) * sizeof(C);
    const unsigned g1 = (unsigned)(batch * (sh.LB / kC1)), g2 = (unsigned)((total_pts + tile_pts - 1) / tile_pts);
    // threads per workgroup, measured at 64 x 5 10^5: f64 0.56 ms with 512 / 512 against 0.63 with 256 / 256
    // (more butterflies in flight per LDS tile); f32 is best at 256 / 256
    static const int env_blk = getenv("TRAMP_FFT_BLOCK_THREADS") ? atoi(getenv("TRAMP_FFT_BLOCK_THREADS")) : 0;
    static const int env_str = getenv("TRAMP_FFT_STRIDED_THREADS") ? atoi(getenv("TRAMP_FFT_STRIDED_THREADS")) : 0;
    const int blk_threads = env_blk ? env_blk : (sizeof(T) == 8 ? 512 : 256);
    const int str_threads = env_str ? env_str : (sizeof(T) == 8 ? 512 : 256);
    static bool attr_done[4] = {false, false, false, false};
    const int slot = (sizeof(T) == 8 ? 0 : 2) + (inverse ? 1 : 0);
    if (!attr_done[slot]) {
        const void* f1 = inverse ? reinterpret_cast<const void*>(fft_strided_kernel<T, true>)
                                 : reinterpret_cast<const void*>(fft_strided_kernel<T, false>);
        const void* f2 = inverse ? reinterpret_cast<const void*>(fft_block_kernel<T, true>)
                                 : reinterpret_cast<const void*>(fft_block_kernel<T, false>);
        // dynamic LDS up to the CU's 160 KiB minus what the kernel declares statically (the epilogue kernel
        // carries the erfcx table of a sign-likelihood consumer): asking for more fails with "invalid argument"
        auto allow = [](const void* f) {
            hipFuncAttributes fa{};
            size_t stat = 0;
            if (hipFuncGetAttributes(&fa, f) == hipSuccess) stat = fa.sharedSizeBytes;
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - stat)) != hipSuccess)
                (void)hipGetLastError();
        };
        if (inverse) allow(reinterpret_cast<const void*>(fft_strided_kernel<T, true, true>));
        if (inverse) allow(reinterpret_cast<const void*>(fft_block_kernel<T, true, true>));
        allow(f1);
        allow(f2);
        attr_done[slot] = true;
    }
    // (the epilogue kernel also carries the 2 KiB erfcx table of a sign-likelihood consumer, prior_hint.h)
    if (lds1 + 4096 > 160 * 1024 || lds2 > 160 * 1024)
        return tramp_fail(ctx, TRAMP_ENOTSUP, "fft2", "transform too long for the LDS tile");
    if (sh.G == 1) {
        // the whole transform is one block pass, out of place
        if (epi && epi->out && !inverse) return tramp_fail(ctx, TRAMP_EINVAL, "fft2", "epilogue on a forward transform");
        if (inverse && epi && epi->out) {
            Fft2Epi e2 = *epi;
            e2.hint.kind = PRIOR_HINT_NONE;
            // producer-fused moments for the consumer of this message, as the two-pass epilogue below: one partial per tile,
            // tiles inside one instance (an instance = rows_per_inst rows of LB points)
            const size_t ppi = (e2.rows_per_inst ? e2.rows_per_inst : 1) * sh.LB;
            const int hk = ctx->hint.kind;
            if (ppi % (size_t)tile_pts == 0 && !resident_covers(ctx, 2 * ppi, total_pts / ppi, sizeof(T)) &&
                (hk == PRIOR_HINT_GB || hk == PRIOR_HINT_L1 || (hk == PRIOR_HINT_SGN_LIK && sizeof(T) == 8)))
                prior_hint_arm(ctx, e2.out, 2 * ppi, total_pts / ppi, (int)(ppi / (size_t)tile_pts), &e2.hint);
            hipLaunchKernelGGL((fft_block_kernel<T, true, true>), dim3(g2), dim3(blk_threads), lds2, ctx->stream, sh,
                               (const C*)in, (C*)out, (const C*)pl->tw_blk, tile_pts, total_pts, e2);
        } else if (inverse)
            hipLaunchKernelGGL((fft_block_kernel<T, true>), dim3(g2), dim3(blk_threads), lds2, ctx->stream, sh, (const C*)in,
                               (C*)out, (const C*)pl->tw_blk, tile_pts, total_pts);
        else
            hipLaunchKernelGGL((fft_block_kernel<T, false>), dim3(g2), dim3(blk_threads), lds2, ctx->stream, sh, (const C*)in,
                               (C*)out, (const C*)pl->tw_blk, tile_pts, total_pts);
    } else if (!inverse) {
        hipLaunchKernelGGL((fft_strided_kernel<T, false>), dim3(g1), dim3(str_threads), lds1, ctx->stream, sh, (const C*)in,
                           (C*)out, (const C*)pl->tw_hi, (const C*)pl->tw_lo, Fft2Epi{});
        hipLaunchKernelGGL((fft_block_kernel<T, false>), dim3(g2), dim3(blk_threads), lds2, ctx->stream, sh, (const C*)out,
                           (C*)out, (const C*)pl->tw_blk, tile_pts, total_pts);
    } else {
        // `in` (the permuted spectrum) is overwritten by the block pass
        hipLaunchKernelGGL((fft_block_kernel<T, true>), dim3(g2), dim3(blk_threads), lds2, ctx->stream, sh, (const C*)in,
                           (C*)const_cast<void*>(in), (const C*)pl->tw_blk, tile_pts, total_pts);
        if (epi && epi->out) {
            Fft2Epi e2 = *epi;
            // message length 2 h reals per instance, one partial per tile.  This pass is bound by its own
            // arithmetic (butterflies, twiddles), so moments evaluated here cost their full VALU time: worth it for
            // the priors and the f64 sign likelihood (64 x 1e6: 467-574 us fused against 571-622 on its own), not for
            // the abs likelihood (432-509 against 438-466) or the f32 likelihoods (363 against 242):
            // gpurun_out/r03/fused1.log, fused2.log.  The dense channel's fold kernels arm for every kind.
            e2.hint.kind = PRIOR_HINT_NONE;
            const int hk = ctx->hint.kind;
            if (e2.rows_per_inst <= 1 && !resident_covers(ctx, 2 * sh.h, batch, sizeof(T)) &&
                (hk == PRIOR_HINT_GB || hk == PRIOR_HINT_L1 || (hk == PRIOR_HINT_SGN_LIK && sizeof(T) == 8)))
                prior_hint_arm(ctx, e2.out, 2 * sh.h, batch, (int)(sh.LB / kC1), &e2.hint);
            hipLaunchKernelGGL((fft_strided_kernel<T, true, true>), dim3(g1), dim3(str_threads), lds1, ctx->stream, sh,
                               (const C*)in, (C*)out, (const C*)pl->tw_hi, (const C*)pl->tw_lo, e2);
        }
        else
            hipLaunchKernelGGL((fft_strided_kernel<T, true>), dim3(g1), dim3(str_threads), lds1, ctx->stream, sh, (const C*)in,
                               (C*)out, (const C*)pl->tw_hi, (const C*)pl->tw_lo, Fft2Epi{});
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

int fft2_forward(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, const void* in, void* out) {
    return pl->dtype == TRAMP_F64 ? fft2_run_t<double>(ctx, pl, batch, false, in, out)
                                  : fft2_run_t<float>(ctx, pl, batch, false, in, out);
}

int fft2_inverse(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, void* spectrum, void* out, const Fft2Epi* epi) {
    return pl->dtype == TRAMP_F64 ? fft2_run_t<double>(ctx, pl, batch, true, spectrum, out, epi)
                                  : fft2_run_t<float>(ctx, pl, batch, true, spectrum, out, epi);
}

// ---------------------------------------------------------------- N-D plans
namespace {

constexpr size_t kAxisGmax = 512;      // tile rows of a strided axis pass (f64: 512 x 9 x 16 B = 72 KiB of LDS)

bool smooth25(size_t v) {
    if (!v) return false;
    while (v % 2 == 0) v /= 2;
    while (v % 5 == 0) v /= 5;
    return v == 1;
}

// the one or two strided passes of an axis of length n over rows of W complex elements
bool axis_passes(size_t n, size_t W, int axis, FftAxisPass* out, int* np, int* rad, int* nrad) {
    *np = 0;
    *nrad = 0;
    if (n == 1) return true;
    if (!smooth25(n)) return false;
    size_t ga = n, gb = 1;
    if (n > kAxisGmax) {
        double best = 1e300;
        ga = 0;
        for (size_t d = 2; d <= kAxisGmax; ++d) {
            if (n % d || !smooth25(d) || n / d > kAxisGmax) continue;
            const double cost = fabs(log((double)(n / d) / (double)d));
            if (cost < best) { best = cost; gb = d; ga = n / d; }
        }
        if (!ga) return false;
    }
    auto finish = [&](FftAxisPass& ps) {
        ps.W = W;
        ps.n = n;
        ps.axis = axis;
        ps.ntw = 0;
        for (int s = 0; s < ps.nst; ++s) {
            ps.off[s] = ps.ntw;
            ps.ntw += ps.sub[s];
            rad[(*nrad)++] = ps.r[s];
        }
    };
    FftAxisPass& p1 = out[(*np)++];
    memset(&p1, 0, sizeof(p1));
    p1.G = (int)ga;
    p1.Q = (int)gb;
    p1.nA = 1;
    p1.nst = radices_of(ga, p1.r);
    if (p1.nst < 1 || p1.nst > 8) return false;
    size_t prod = 1;
    int rows = (int)ga;
    for (int s = 0; s < p1.nst; ++s) {
        rows /= p1.r[s];
        p1.sub[s] = rows;
        p1.mul[s] = prod;
        prod *= p1.r[s];
    }
    finish(p1);
    if (gb > 1) {
        FftAxisPass& p2 = out[(*np)++];
        memset(&p2, 0, sizeof(p2));
        p2.G = (int)gb;
        p2.Q = 1;
        p2.nA = (int)ga;
        p2.nst = radices_of(gb, p2.r);
        if (p2.nst < 1 || p2.nst > 8) return false;
        size_t L = gb;
        for (int s = 0; s < p2.nst; ++s) {
            p2.sub[s] = (int)(L / p2.r[s]);
            p2.mul[s] = n / L;
            L /= p2.r[s];
        }
        finish(p2);
    }
    return true;
}

template <typename T>
int axis_tables_t(tramp_ctx* ctx, size_t n, void** hi, void** lo) {
    using C = typename C2<T>::type;
    const size_t nhi = (n + 1023) >> 10;
    TRAMP_HIP_CHECK(ctx, hipMalloc(hi, nhi * sizeof(C)));
    TRAMP_HIP_CHECK(ctx, hipMalloc(lo, 1024 * sizeof(C)));
    hipLaunchKernelGGL((fft_tables_kernel<T>), dim3(4), dim3(256), 0, ctx->stream, n, nhi, (C*)*hi, (C*)*lo);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

template <typename T>
int axis_run_t(tramp_ctx* ctx, const FftNdPlan* pl, const FftAxisPass& ps, size_t batch, bool inverse, void* data,
               void* data_out = nullptr, const int* rowmap = nullptr, int map_mode = 0) {
    if (!data_out) data_out = data;
    using C = typename C2<T>::type;
    constexpr int kC1 = c1_of<T>();
    size_t outer = batch;
    for (int a = 0; a < ps.axis; ++a) outer *= pl->dims[a];
    const size_t tiles = outer * ps.nA * ps.Q * (ps.W / kC1);
    if (tiles > 0x7fffffffull) return tramp_fail(ctx, TRAMP_ENOTSUP, "fftnd", "too many tiles for one launch");
    const size_t lds = ((size_t)ps.G * (kC1 + 1) + (size_t)ps.ntw) * sizeof(C);
    // threads: a quarter of the tile's points (one radix-4 butterfly each), within [64, 512]
    static const int env_thr = getenv("TRAMP_FFT_AXIS_THREADS") ? atoi(getenv("TRAMP_FFT_AXIS_THREADS")) : 0;
    // points per thread: 4 in f64, 8 in f32 (measured: f32 1024^2 x 64 gradient pair 4.36 -> 4.17 ms, 128^3 0.90 -> 0.85;
    // f64 is best at 4: tools/sweep_fftnd.sh)
    static const int env_div = getenv("TRAMP_FFT_AXIS_DIV") ? atoi(getenv("TRAMP_FFT_AXIS_DIV")) : 0;
    const int div = env_div > 0 ? env_div : (sizeof(T) == 8 ? 4 : 8);
    int thr = (ps.G * kC1 / div + 63) / 64 * 64;
    thr = thr < 64 ? 64 : (thr > 512 ? 512 : thr);
    if (env_thr) thr = env_thr;
    static bool attr_done[4] = {false, false, false, false};
    const int slot = (sizeof(T) == 8 ? 0 : 2) + (inverse ? 1 : 0);
    const void* f = inverse ? reinterpret_cast<const void*>(fft_axis_kernel<T, true>)
                            : reinterpret_cast<const void*>(fft_axis_kernel<T, false>);
    if (!attr_done[slot]) {
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_done[slot] = true;
    }
    const int ax = ps.axis;
    if (inverse)
        hipLaunchKernelGGL((fft_axis_kernel<T, true>), dim3((unsigned)tiles), dim3(thr), lds, ctx->stream, ps, (const C*)data,
                           (C*)data_out, (const C*)pl->tw_hi[ax], (const C*)pl->tw_lo[ax], rowmap, map_mode);
    else
        hipLaunchKernelGGL((fft_axis_kernel<T, false>), dim3((unsigned)tiles), dim3(thr), lds, ctx->stream, ps, (const C*)data,
                           (C*)data_out, (const C*)pl->tw_hi[ax], (const C*)pl->tw_lo[ax], rowmap, map_mode);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// one launch per transform for a 2-D image that fits one tile: block-only rows, one axis pass, n0 * hl points in LDS
bool image_fits(const FftNdPlan* pl) {
    static const bool off = getenv("TRAMP_FFT_IMAGE") && atoi(getenv("TRAMP_FFT_IMAGE")) == 0;
    if (off || pl->ndim != 2 || pl->npass != 1 || pl->row->shape.G != 1 || pl->pass[0].Q != 1) return false;
    const size_t pts = pl->h, csz = pl->dtype == TRAMP_F64 ? 16 : 8;
    const size_t lds = (pts + (size_t)pl->row->shape.ntw2 + (size_t)pl->pass[0].ntw) * csz;
    return pts <= 4096 && lds <= 96 * 1024;
}

template <typename T>
int image_run_t(tramp_ctx* ctx, const FftNdPlan* pl, size_t images, bool inverse, const void* in, void* out,
                const Fft2Epi* epi) {
    using C = typename C2<T>::type;
    const Fft2Shape& sh = pl->row->shape;
    const FftAxisPass& ps = pl->pass[0];
    const size_t lds = (pl->h + (size_t)sh.ntw2 + (size_t)ps.ntw) * sizeof(C);
    int thr = (int)((pl->h / 4 + 63) / 64 * 64);
    thr = thr < 64 ? 64 : (thr > 512 ? 512 : thr);
    static bool attr_done[2] = {false, false};
    const int slot = sizeof(T) == 8 ? 0 : 1;
    if (!attr_done[slot]) {
        for (const void* f : {reinterpret_cast<const void*>(fft_image_kernel<T, false>),
                              reinterpret_cast<const void*>(fft_image_kernel<T, true>),
                              reinterpret_cast<const void*>(fft_image_kernel<T, true, true>)})
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
                (void)hipGetLastError();
        attr_done[slot] = true;
    }
    const dim3 grid((unsigned)images), blk(thr);
    const C *hi = (const C*)pl->tw_hi[0], *lo = (const C*)pl->tw_lo[0], *tb = (const C*)pl->row->tw_blk;
    if (!inverse)
        hipLaunchKernelGGL((fft_image_kernel<T, false>), grid, blk, lds, ctx->stream, sh, ps, (const C*)in, (C*)out, tb, hi, lo,
                           Fft2Epi{});
    else if (epi && epi->out)
        hipLaunchKernelGGL((fft_image_kernel<T, true, true>), grid, blk, lds, ctx->stream, sh, ps, (const C*)in, (C*)out, tb, hi,
                           lo, *epi);
    else
        hipLaunchKernelGGL((fft_image_kernel<T, true>), grid, blk, lds, ctx->stream, sh, ps, (const C*)in, (C*)out, tb, hi, lo,
                           Fft2Epi{});
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// the last axis alone: block pass for hl <= 4096, the two-pass engine above that
bool row_shape(size_t hl, Fft2Shape* sh, bool* block_only) {
    if (fft2_block_shape(hl, sh)) {
        *block_only = true;
        return true;
    }
    *block_only = false;
    return fft2_shape(hl, sh);
}

}  // namespace

bool fftnd_supported(int dtype, int ndim, const size_t* rd, bool packed_real) {
    if (ndim < 1 || ndim > 3 || (dtype != TRAMP_F64 && dtype != TRAMP_F32)) return false;
    const size_t nl = rd[ndim - 1];
    if (packed_real && (nl < 4 || (nl & 1))) return false;
    const size_t hl = packed_real ? nl / 2 : nl;
    const size_t c1 = dtype == TRAMP_F64 ? c1_of<double>() : c1_of<float>();
    Fft2Shape sh;
    bool blk;
    if (!row_shape(hl, &sh, &blk)) return false;
    if (ndim > 1 && hl % c1) return false;                  // the strided passes move whole 128-byte lines
    size_t W = hl;
    FftAxisPass tmp[2];
    int np, rad[16], nrad;
    for (int a = ndim - 2; a >= 0; --a) {
        if (!axis_passes(rd[a], W, a, tmp, &np, rad, &nrad)) return false;
        W *= rd[a];
    }
    return true;
}

int fftnd_plan_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* rd, FftNdPlan** out, bool packed_real) {
    if (!fftnd_supported(dtype, ndim, rd, packed_real))
        return tramp_fail(ctx, TRAMP_ENOTSUP, "fftnd_plan_create", "shape outside the own FFT engine");
    FftNdPlan* pl = new FftNdPlan();
    memset(pl, 0, sizeof(*pl));
    pl->dtype = dtype;
    pl->ndim = ndim;
    pl->h = 1;
    for (int a = 0; a < ndim; ++a) {
        pl->dims[a] = (a == ndim - 1 && packed_real) ? rd[a] / 2 : rd[a];
        pl->h *= pl->dims[a];
    }
    const size_t hl = pl->dims[ndim - 1];
    Fft2Shape sh;
    bool blk = false;
    (void)row_shape(hl, &sh, &blk);
    int rc = fft2_plan_create(ctx, dtype, hl, &pl->row, blk);
    // outer axes, innermost first in the forward order
    size_t W = hl;
    for (int a = ndim - 2; a >= 0 && !rc; --a) {
        int np = 0;
        if (!axis_passes(pl->dims[a], W, a, pl->pass + pl->npass, &np, pl->rad[a], &pl->nrad[a])) rc = TRAMP_ENOTSUP;
        pl->npass += np;
        if (!rc && pl->dims[a] > 1)
            rc = dtype == TRAMP_F64 ? axis_tables_t<double>(ctx, pl->dims[a], &pl->tw_hi[a], &pl->tw_lo[a])
                                    : axis_tables_t<float>(ctx, pl->dims[a], &pl->tw_hi[a], &pl->tw_lo[a]);
        W *= pl->dims[a];
    }
    if (rc) {
        fftnd_plan_destroy(pl);
        return rc;
    }
    *out = pl;
    return TRAMP_OK;
}

void fftnd_plan_destroy(FftNdPlan* pl) {
    if (!pl) return;
    fft2_plan_destroy(pl->row);
    for (int a = 0; a < 2; ++a) {
        if (pl->tw_hi[a]) (void)hipFree(pl->tw_hi[a]);
        if (pl->tw_lo[a]) (void)hipFree(pl->tw_lo[a]);
        if (pl->kf_dev[a]) (void)hipFree(pl->kf_dev[a]);
    }
    if (pl->pos_last_dev) (void)hipFree(pl->pos_last_dev);
    delete pl;
}

size_t fftnd_kfreq(const FftNdPlan* pl, int axis, size_t p) {
    if (axis == pl->ndim - 1) return fft2_kfreq(pl->row->shape, p);
    size_t k = 0, weight = 1, len = pl->dims[axis];
    for (int s = 0; s < pl->nrad[axis]; ++s) {
        const int r = pl->rad[axis][s];
        len /= r;
        k += (p / len) * weight;
        p %= len;
        weight *= r;
    }
    return k;
}

int fftnd_forward(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, const void* in, void* out) {
    if (image_fits(pl))
        return pl->dtype == TRAMP_F64 ? image_run_t<double>(ctx, pl, batch, false, in, out, nullptr)
                                      : image_run_t<float>(ctx, pl, batch, false, in, out, nullptr);
    const size_t rows = batch * (pl->h / pl->dims[pl->ndim - 1]);
    int rc = fft2_forward(ctx, pl->row, rows, in, out);
    for (int i = 0; i < pl->npass && !rc; ++i)
        rc = pl->dtype == TRAMP_F64 ? axis_run_t<double>(ctx, pl, pl->pass[i], batch, false, out)
                                    : axis_run_t<float>(ctx, pl, pl->pass[i], batch, false, out);
    return rc;
}

int fftnd_inverse(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, void* spectrum, void* out, const Fft2Epi* epi) {
    if (image_fits(pl)) {
        Fft2Epi e2 = epi ? *epi : Fft2Epi{};
        e2.hint.kind = PRIOR_HINT_NONE;
        // rows_per_inst counts IMAGES per instance for this kernel (the d components of a gradient)
        return pl->dtype == TRAMP_F64 ? image_run_t<double>(ctx, pl, batch, true, spectrum, out, epi ? &e2 : nullptr)
                                      : image_run_t<float>(ctx, pl, batch, true, spectrum, out, epi ? &e2 : nullptr);
    }
    int rc = TRAMP_OK;
    for (int i = pl->npass - 1; i >= 0 && !rc; --i)
        rc = pl->dtype == TRAMP_F64 ? axis_run_t<double>(ctx, pl, pl->pass[i], batch, true, spectrum)
                                    : axis_run_t<float>(ctx, pl, pl->pass[i], batch, true, spectrum);
    if (rc) return rc;
    const size_t rpi = pl->h / pl->dims[pl->ndim - 1];
    if (epi && epi->out) {
        Fft2Epi e2 = *epi;
        e2.rows_per_inst = rpi * (epi->rows_per_inst ? epi->rows_per_inst : 1);
        return fft2_inverse(ctx, pl->row, batch * rpi, spectrum, out, &e2);
    }
    return fft2_inverse(ctx, pl->row, batch * rpi, spectrum, out);
}

// ---- natural-order mode
bool fftnd_natural_ok(const FftNdPlan* pl) { return pl->row->shape.G == 1; }

int fftnd_enable_natural(tramp_ctx* ctx, FftNdPlan* pl) {
    if (pl->pos_last_dev) return TRAMP_OK;
    if (!fftnd_natural_ok(pl)) return tramp_fail(ctx, TRAMP_ENOTSUP, "fftnd natural order", "last axis longer than one block");
    const int nd = pl->ndim;
    const size_t hl = pl->dims[nd - 1];
    std::vector<int> pos(hl);
    for (size_t p = 0; p < hl; ++p) pos[fftnd_kfreq(pl, nd - 1, p)] = (int)p;
    TRAMP_HIP_CHECK(ctx, hipMalloc((void**)&pl->pos_last_dev, hl * sizeof(int)));
    TRAMP_HIP_CHECK(ctx, hipMemcpy(pl->pos_last_dev, pos.data(), hl * sizeof(int), hipMemcpyHostToDevice));
    for (int a = 0; a < nd - 1; ++a) {
        const size_t n = pl->dims[a];
        std::vector<int> kf(n);
        for (size_t p = 0; p < n; ++p) kf[p] = (int)fftnd_kfreq(pl, a, p);
        TRAMP_HIP_CHECK(ctx, hipMalloc((void**)&pl->kf_dev[a], n * sizeof(int)));
        TRAMP_HIP_CHECK(ctx, hipMemcpy(pl->kf_dev[a], kf.data(), n * sizeof(int), hipMemcpyHostToDevice));
    }
    return TRAMP_OK;
}

namespace {

template <typename T>
int row_run_nat_t(tramp_ctx* ctx, const FftNdPlan* pl, size_t rows, bool inverse, const void* in, void* out) {
    using C = typename C2<T>::type;
    const Fft2Shape& sh = pl->row->shape;
    const size_t total_pts = rows * sh.h;
    int tile_pts = (int)sh.LB;
    if (sh.LB < 2048) tile_pts = (int)(sh.LB * (2048 / sh.LB));
    const size_t lds = ((size_t)tile_pts + (size_t)sh.ntw2) * sizeof(C);
    const unsigned g = (unsigned)((total_pts + tile_pts - 1) / tile_pts);
    const int thr = sizeof(T) == 8 ? 512 : 256;
    static bool attr_done[2] = {false, false};
    if (!attr_done[sizeof(T) == 8 ? 0 : 1]) {
        for (const void* f : {reinterpret_cast<const void*>(fft_block_kernel<T, false>),
                              reinterpret_cast<const void*>(fft_block_kernel<T, true>)})
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                (void)hipGetLastError();
        attr_done[sizeof(T) == 8 ? 0 : 1] = true;
    }
    if (inverse)
        hipLaunchKernelGGL((fft_block_kernel<T, true>), dim3(g), dim3(thr), lds, ctx->stream, sh, (const C*)in, (C*)out,
                           (const C*)pl->row->tw_blk, tile_pts, total_pts, Fft2Epi{}, pl->pos_last_dev, 2);
    else
        hipLaunchKernelGGL((fft_block_kernel<T, false>), dim3(g), dim3(thr), lds, ctx->stream, sh, (const C*)in, (C*)out,
                           (const C*)pl->row->tw_blk, tile_pts, total_pts, Fft2Epi{}, pl->pos_last_dev, 1);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

template <typename T>
int fftnd_nat_t(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, bool inverse, void* a, void* b, void** result) {
    const size_t rows = batch * (pl->h / pl->dims[pl->ndim - 1]);
    void *cur = a, *other = b;
    int rc = TRAMP_OK;
    if (!inverse) {
        rc = row_run_nat_t<T>(ctx, pl, rows, false, cur, other);
        std::swap(cur, other);
        for (int i = 0; i < pl->npass && !rc; ++i) {
            const FftAxisPass& ps = pl->pass[i];
            const bool last_of_axis = ps.Q == 1;                 // single pass, or the second of two
            const bool hop = last_of_axis && ps.nA > 1;          // its rows land in other tiles' ranges: out of place
            rc = axis_run_t<T>(ctx, pl, ps, batch, false, cur, hop ? other : cur, last_of_axis ? pl->kf_dev[ps.axis] : nullptr,
                               last_of_axis ? 1 : 0);
            if (hop) std::swap(cur, other);
        }
        *result = cur;
        return rc;
    }
    for (int i = pl->npass - 1; i >= 0 && !rc; --i) {
        const FftAxisPass& ps = pl->pass[i];
        const bool last_of_axis = ps.Q == 1;                     // runs FIRST in the inverse: reads rows by frequency
        const bool hop = last_of_axis && ps.nA > 1;
        rc = axis_run_t<T>(ctx, pl, ps, batch, true, cur, hop ? other : cur, last_of_axis ? pl->kf_dev[ps.axis] : nullptr,
                           last_of_axis ? 2 : 0);
        if (hop) std::swap(cur, other);
    }
    if (rc) return rc;
    rc = row_run_nat_t<T>(ctx, pl, rows, true, cur, other);
    *result = other;
    return rc;
}

}  // namespace

int fftnd_forward_nat(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, void* a, void* b, void** result) {
    if (!pl->pos_last_dev) return tramp_fail(ctx, TRAMP_EINVAL, "fftnd_forward_nat", "natural order not enabled");
    return pl->dtype == TRAMP_F64 ? fftnd_nat_t<double>(ctx, pl, batch, false, a, b, result)
                                  : fftnd_nat_t<float>(ctx, pl, batch, false, a, b, result);
}

int fftnd_inverse_nat(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, void* a, void* b, void** result) {
    if (!pl->pos_last_dev) return tramp_fail(ctx, TRAMP_EINVAL, "fftnd_inverse_nat", "natural order not enabled");
    return pl->dtype == TRAMP_F64 ? fftnd_nat_t<double>(ctx, pl, batch, true, a, b, result)
                                  : fftnd_nat_t<float>(ctx, pl, batch, true, a, b, result);
}

// `signals` consecutive signals share one problem instance (the d components of a gradient): the epilogue's scalar
// tables are indexed by instance
int fftnd_inverse_signals(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, size_t signals, void* spectrum,
                          const Fft2Epi* epi) {
    Fft2Epi e2 = *epi;
    e2.rows_per_inst = signals;
    return fftnd_inverse(ctx, pl, batch * signals, spectrum, nullptr, &e2);
}

// ---------------------------------------------------------------- exported for tests / tools
extern "C" int tramp_fft_len(tramp_ctx* ctx, int dtype, size_t h, size_t batch, int inverse, void* in_d,
                             void* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, in_d && out_d && (dtype == TRAMP_F64 || dtype == TRAMP_F32));
    Fft2Plan* pl = nullptr;
    int rc = fft2_plan_create(ctx, dtype, h, &pl);
    if (rc) return rc;
    rc = inverse ? fft2_inverse(ctx, pl, batch, in_d, out_d) : fft2_forward(ctx, pl, batch, in_d, out_d);
    (void)hipStreamSynchronize(ctx->stream);
    fft2_plan_destroy(pl);
    return rc;
}

extern "C" int tramp_fft_order(tramp_ctx* ctx, size_t h, size_t count, unsigned long long* kfreq_h) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, kfreq_h);
    Fft2Shape sh;
    if (!fft2_shape(h, &sh)) return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_fft_order", "length");
    for (size_t p = 0; p < count; ++p) kfreq_h[p] = fft2_kfreq(sh, p);
    return TRAMP_OK;
}

// Average time of `reps` forward and of `reps` inverse transforms (HIP events on the context's stream).
extern "C" int tramp_fft_bench(tramp_ctx* ctx, int dtype, size_t h, size_t batch, int reps, void* in_d, void* out_d,
                               double* ms_fwd, double* ms_inv) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, in_d && out_d && ms_fwd && ms_inv && reps > 0);
    Fft2Plan* pl = nullptr;
    int rc = fft2_plan_create(ctx, dtype, h, &pl);
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

// N-D complex transform of the packed signal (tests / tools): real dims [ndim], data [batch] x packed complex
extern "C" int tramp_fftn(tramp_ctx* ctx, int dtype, int ndim, const size_t* real_dims, size_t batch, int inverse,
                          void* in_d, void* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, in_d && out_d && real_dims);
    FftNdPlan* pl = nullptr;
    int rc = fftnd_plan_create(ctx, dtype, ndim, real_dims, &pl);
    if (rc) return rc;
    rc = inverse ? fftnd_inverse(ctx, pl, batch, in_d, out_d) : fftnd_forward(ctx, pl, batch, in_d, out_d);
    (void)hipStreamSynchronize(ctx->stream);
    fftnd_plan_destroy(pl);
    return rc;
}

// frequency held by each position of one axis after tramp_fftn's forward transform
extern "C" int tramp_fftn_order(tramp_ctx* ctx, int dtype, int ndim, const size_t* real_dims, int axis,
                                unsigned long long* kfreq_h) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, kfreq_h && real_dims && axis >= 0 && axis < ndim);
    FftNdPlan* pl = nullptr;
    int rc = fftnd_plan_create(ctx, dtype, ndim, real_dims, &pl);
    if (rc) return rc;
    for (size_t p = 0; p < pl->dims[axis]; ++p) kfreq_h[p] = fftnd_kfreq(pl, axis, p);
    (void)hipStreamSynchronize(ctx->stream);
    fftnd_plan_destroy(pl);
    return TRAMP_OK;
}

// 1 when the own engine takes this shape (no device call)
extern "C" int tramp_fftn_supported(int dtype, int ndim, const size_t* real_dims) {
    return real_dims && fftnd_supported(dtype, ndim, real_dims) ? 1 : 0;
}
