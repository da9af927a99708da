// gemm_sliced.h - the dense passes of the linear channel on the int8 matrix cores (an opt-in engine).
//
// gfx950 has half the f64 matrix rate of its predecessor (78.6 TFLOP/s) and twice the int8 rate (~5 POP/s): at 64
// instances the f64 passes of gemm_mfma.hip are bound by v_mfma_f64_16x16x4 (and the package power), not by HBM.
// Here both operands are split ONCE into S signed radix-256 digits per element against a per-row (per-instance)
// power-of-two scale: with N = rint(a 2^(6 + 8 (S - 1) - e_row)) (|N| <= 2^(6 + 8 (S - 1)), an int64),
//     a ~ 2^e_row sum_i A_i 2^-(6 + 8 i),   A_i = the balanced base-256 digits of N in [-128, 127], |A_0| <= 65,
// (least significant first: A = (int8) (N & 255), N = (N - A) >> 8).  Every digit product A_i X_j is accumulated EXACTLY
// in int32 by v_mfma_i32_16x16x64_i8, products of equal weight i + j = d share an accumulator, products with d >= S are
// dropped, and the S accumulators are summed in f64 from the smallest weight up.  S = 7 carries 54 bits of each
// operand.  With u = 2^(2 - 8 S) the result differs from the exact dot product by at most
//     u (4 (S - 1) k max|row| max|x| + max|row| sum|x| + max|x| sum|row|)
// (dropped products, the two roundings to N) - no rounding accumulates along the contraction - against k 2^-53 sum|a x|
// for an f64 FMA chain; measured at k = 4100 ... 20000 on Gaussian data: S = 7 as the f64 matrix-core pass (within a
// factor 1 - 4 either way), S = 8 twenty times below it.  S (S + 1) / 2 = 28 int8 MFMAs of 16 cycles
// replace 16 f64 MFMAs of 64 cycles per 16 x 16 x 64 block (0.44 of the matrix-pipe time) at 7 instead of 8 bytes per
// matrix element.  f32 operands (24 bits) need S = 4: 10 MFMAs of 16 cycles against 16 of 32.
// An accumulator holds at most S products of <= 2^14 per contraction index: the host keeps the contraction length of a
// split below 2^31 / (S 2^14) (18 724 for S = 7) by the number of splits.
//
// Layouts (both are streams of 1 KiB fragments in MFMA operand order: lane l = j + 16 q owns bytes [16 l, 16 l + 16) =
// row (instance) j of the 16-row tile, contraction indices 64 kb + 16 q + (0..15)):
//     matrix:  fragment (row tile rt, k block kb, digit i)          at ((rt nkb + kb) S + i) KiB
//     x:       fragment (k block kb, instance tile bt, digit j)     at ((kb nbt + bt) S + j) KiB
// so a wave streams 8 KiB of CONTIGUOUS matrix per k block and the x fragments go global -> LDS by global_load_lds
// (lane-linear image) with no register hop.  The column pass (out = mat^T y) runs the same kernel on a sliced copy of
// the transpose.
#pragma once
#include "common.h"

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kSlFrag = 1024;            // bytes per fragment
constexpr int kSlK = 64;                 // contraction depth of one fragment
constexpr int kSlMaxS = 8;             // 6 + 8 (S - 1) bits must fit an int64

// power-of-two scale of a row from the bit pattern of its largest magnitude: 2^(ilogb(max) + 1) (>= every |a|; exact
// reciprocal), 1 for a zero row, NaN for a row that holds a non-finite value (the whole output row / instance turns NaN)
__device__ __forceinline__ double sl_scale_of(unsigned long long amax_bits) {
    const double m = __longlong_as_double((long long)amax_bits);
    if (!(m < __builtin_inf())) return __longlong_as_double(0x7ff8000000000000LL);
    if (m == 0.0) return 1.0;
    const int e = (int)((amax_bits >> 52) & 0x7ff);
    // The slicing multiplies by 2^(6 + 8 (S - 1)) / scale <= 2^62 / scale, which must stay finite: no scale below 2^-960
    // (a row whose largest entry is under 1e-289 keeps fewer significant digits - it is zero for every purpose).  A
    // maximum of 2^1023 or more would need the scale 2^1024 = inf: such a row is treated as non-finite (NaN output).
    if (e >= 2046) return __longlong_as_double(0x7ff8000000000000LL);
    const int es = e + 1 < 1023 - 960 ? 1023 - 960 : e + 1;
    return __longlong_as_double((long long)((unsigned long long)es << 52));
}

// ---------------------------------------------------------------------------------------------- largest magnitudes
// amax[r] (bit patterns; atomicMax on them orders non-negative doubles, NaN > inf > finite) over the logical row r.
// TR = false: logical (r, c) = mat[r ld + c], one workgroup per (row, column chunk);
// TR = true:  logical (r, c) = mat[c ld + r], thread per logical row, blockIdx.y = chunk of c.
template <typename T, bool TR>
__global__ __launch_bounds__(kBlock) void sl_absmax_kernel(size_t R, size_t C, size_t ld, const T* __restrict__ mat,
                                                           unsigned long long* __restrict__ amax) {
    const size_t cchunk = (C + gridDim.y - 1) / gridDim.y;
    const size_t c0 = blockIdx.y * cchunk, c1 = (c0 + cchunk < C) ? c0 + cchunk : C;
    if (TR) {
        const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
        if (r >= R) return;
        unsigned long long m = 0;
        for (size_t c = c0; c < c1; ++c) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(fabs((double)mat[c * ld + r]));
            m = b > m ? b : m;
        }
        atomicMax(amax + r, m);
    } else {
        const size_t r = blockIdx.x;
        unsigned long long m = 0;
        constexpr int E = 16 / sizeof(T);               // elements per 16-byte piece
        const T* row = mat + r * ld;
        size_t c = c0;
        if (c0 < c1 && ((size_t)row & 15) == 0 && (c0 % E) == 0) {   // whole pieces, 16 bytes per lane and load (c0 >= c1: an empty chunk)
            typedef T VT __attribute__((ext_vector_type(E)));
            const size_t npiece = (c1 - c0) / E;
            for (size_t i = threadIdx.x; i < npiece; i += kBlock) {
                const VT v = *reinterpret_cast<const VT*>(row + c0 + i * E);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const unsigned long long b = (unsigned long long)__double_as_longlong(fabs((double)v[e]));
                    m = b > m ? b : m;
                }
            }
            c = c0 + npiece * E;
        }
        for (c += threadIdx.x; c < c1; c += kBlock) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(fabs((double)row[c]));
            m = b > m ? b : m;
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_down(m, o, kWave);
            m = other > m ? other : m;
        }
        if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(amax + r, m);
    }
}

// ---------------------------------------------------------------------------------------------- digits
// One thread per (logical row r, group of 16 contraction indices kg): S x 16 bytes into S fragments of tile (r / 16, kg / 4).
// `tiles_outer`: fragment index = ((tile_outer n_inner + tile_inner) S + digit) with
//     matrix (X = false): outer = row tile, inner = k block  (n_inner = nkb)
//     x      (X = true):  outer = k block,  inner = instance tile (n_inner = nbt)
// Rows >= R and contraction indices >= C are zero digits (padding never contributes).
template <typename T, int S, bool TR, bool X>
__global__ __launch_bounds__(kBlock) void sl_slice_kernel(size_t R, size_t Rpad, size_t C, size_t ld, const T* __restrict__ mat,
                                                          const unsigned long long* __restrict__ amax, size_t n_inner,
                                                          int8_t* __restrict__ out, double* __restrict__ scale) {
    // thread -> (r, kg): r fastest, so that the 16 rows of a tile line up in the 16-byte columns of a fragment (256
    // contiguous bytes per store instruction and k group; with kg fastest the digit stores were 16-byte pieces 256
    // bytes apart: 0.8 TB/s).  The contiguous-row matrix form (neither TR nor X) takes a block of 16 rows x 16 groups and
    // reads its 16 x 256 elements through LDS, row by row in 2 KiB wave-contiguous pieces.
    constexpr bool staged = !TR && !X;
    size_t r, kg;
    if (staged) {
        r = (blockIdx.y + (size_t)blockIdx.z * gridDim.y) * 16 + (threadIdx.x & 15);
        kg = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    } else {
        r = (size_t)blockIdx.x * kBlock + threadIdx.x;
        kg = blockIdx.y + (size_t)blockIdx.z * gridDim.y;
    }
    const size_t nkg = (C + kSlK - 1) / kSlK * 4;          // Rpad: rows of whole tiles to write (rows >= R are zeros)
    if (!staged && (r >= Rpad || kg >= nkg)) return;    // (staged: every thread of the block reaches the barrier below)
    double sc = 1.0;
    if (r < R) sc = sl_scale_of(amax[r]);
    if (kg == 0 && r < Rpad) scale[r] = sc;
    __shared__ T stage[staged ? 16 : 1][staged ? kBlock + 1 : 1];
    if (staged) {
        const size_t r0 = (blockIdx.y + (size_t)blockIdx.z * gridDim.y) * 16, c = (size_t)blockIdx.x * kBlock + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 16; ++i) stage[i][threadIdx.x] = (r0 + i < R && c < C) ? mat[(r0 + i) * ld + c] : T(0);
        __syncthreads();
    }
    // exact: sc is a power of two >= 2^-960 (sl_scale_of: the quotient below stays finite), the product a power-of-two multiple of a with |.| <= 2^(6 + 8 (S - 1)) <= 2^62
    const double inv = (sc == sc) ? __longlong_as_double((long long)(1023 + 6 + 8 * (S - 1)) << 52) / sc : 0.0;
    unsigned dig[S][4];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int w = 0; w < 4; ++w) dig[i][w] = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const size_t c = kg * 16 + e;
        long long N = 0;
        if (staged) N = (long long)__builtin_rint((double)stage[threadIdx.x & 15][(threadIdx.x >> 4) * 16 + e] * inv);
        else if (r < R && c < C) N = (long long)__builtin_rint((double)(TR ? mat[c * ld + r] : mat[r * ld + c]) * inv);
#pragma unroll
        for (int i = S - 1; i > 0; --i) {
            const long long d = (long long)(signed char)(N & 0xff);
            dig[i][e >> 2] |= ((unsigned)d & 0xffu) << (8 * (e & 3));
            N = (N - d) >> 8;
        }
        dig[0][e >> 2] |= ((unsigned)N & 0xffu) << (8 * (e & 3));
    }
    if (staged && (r >= Rpad || kg >= nkg)) return;
    const size_t tile_r = r / 16, kb = kg / 4;
    const size_t frag0 = (X ? kb * n_inner + tile_r : tile_r * n_inner + kb) * S;
    const size_t within = ((kg % 4) * 16 + (r % 16)) * 16;
#pragma unroll
    for (int i = 0; i < S; ++i)
        *reinterpret_cast<v4i*>(out + (frag0 + i) * kSlFrag + within) = v4i{(int)dig[i][0], (int)dig[i][1], (int)dig[i][2], (int)dig[i][3]};
}

// ---------------------------------------------------------------------------------------------- the pass
// Workgroup: NW waves (4, or 8 = one workgroup per CU), wave w owns row tile NW rb + w (16 matrix rows) against BT x 16
// instances; blockIdx.x = rb nsplit +
// split (split == XCD under the round-robin dispatch, as in gemm_mfma.hip), blockIdx.y = chunk of 64 instances.
// Per k block and wave: S matrix fragments global -> registers (two alternating sets), BT S x fragments global -> LDS
// (two buffers, each wave moves its quarter), S (S + 1) / 2 x BT MFMAs.  One barrier per k block.
// DBG (probes only): 1 = the matrix pointer does not advance (every k block re-reads the first: the pass without HBM)
template <int S, int BT, int DBG = 0, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_sliced_kernel(size_t R, size_t nkb, size_t B, size_t nbt,
                                                                 const int8_t* __restrict__ mat_s, const double* __restrict__ row_scale,
                                                                 const int8_t* __restrict__ x_s, const double* __restrict__ x_scale,
                                                                 size_t kb_per_split, unsigned nsplit, double* __restrict__ partials) {
    constexpr int NF = BT * S;                         // x fragments per k block
    __shared__ __attribute__((aligned(16))) int8_t xs[2][NF][kSlFrag];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // (tile = row block x split, 64-instance chunk).  Ensembles of more than 64 instances, one-dimensional launch: the
    // sibling chunks of a tile are 8 workgroups apart - the same XCD under the round-robin dispatch and adjacent in
    // time, so the second chunk finds the tile's digits in that XCD's L2 (as wg_item in gemm_mfma.hip)
    size_t tile_id = blockIdx.x, chunk_id = blockIdx.y;
    if (gridDim.y == 1 && B > 64) {
        const size_t nchunk = (B + 63) / 64, g = blockIdx.x / 8, xcd = blockIdx.x % 8;
        chunk_id = g % nchunk;
        tile_id = (g / nchunk) * 8 + xcd;
    }
    const size_t rb = tile_id / nsplit, split = tile_id % nsplit;
    const size_t bchunk = chunk_id * 64;
    const size_t ntile = (R + 15) / 16;
    size_t rt = rb * NW + wave;
    const bool live = rt < ntile;
    if (!live) rt = ntile - 1;                         // a wave past the last tile computes on valid memory and stores nothing
    const size_t kb0 = split * kb_per_split;
    size_t kb1 = kb0 + kb_per_split;
    if (kb1 > nkb) kb1 = nkb;
    const int8_t* mp = mat_s + ((rt * nkb + kb0) * S) * kSlFrag + lane * 16;
    const int8_t* xp = x_s + ((kb0 * nbt + chunk_id * 4) * S) * kSlFrag + lane * 16;
    const size_t x_step = nbt * S * kSlFrag;

    v4i acc[S][BT];
#pragma unroll
    for (int d = 0; d < S; ++d)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) acc[d][bt] = v4i{0, 0, 0, 0};
    v4i ma[S], mb[S];

    auto issue = [&](v4i (&dst)[S], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < S; ++i) dst[i] = *reinterpret_cast<const v4i*>(mp + i * kSlFrag);
#pragma unroll
        for (int f = wave; f < NF; f += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xp + f * kSlFrag),
                                             (__attribute__((address_space(3))) void*)&xs[buf][f][0], 16, 0, 0);
        if (DBG != 1) mp += S * kSlFrag;
        xp += x_step;
    };
    // x fragments are read from LDS two groups ahead of the MFMAs that use them (a group = one fragment against the
    // S - j matrix digits it meets): read just in time, every group began with a wait for the LDS round trip and the
    // matrix pipe was idle half of the time
    constexpr int NG = S * BT, AHEAD = 2;
    auto compute = [&](const v4i (&m)[S], int buf) __attribute__((always_inline)) {
        v4i xa[AHEAD + 1];
        // digit order 0, S-1, 1, S-2, ...: long groups (j small: S - j MFMAs) alternate with short ones, so two groups
        // ahead is always >= S + 1 MFMAs of cover
        auto digit = [](int g) { const int o = g / BT; return (o & 1) ? S - 1 - o / 2 : o / 2; };
        auto read = [&](int g) __attribute__((always_inline)) {
            const int j = digit(g), bt = g % BT;
            return *reinterpret_cast<const v4i*>(&xs[buf][bt * S + j][lane * 16]);
        };
#pragma unroll
        for (int g = 0; g < AHEAD; ++g) xa[g] = read(g);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + AHEAD < NG) xa[(g + AHEAD) % (AHEAD + 1)] = read(g + AHEAD);
            const int j = digit(g), bt = g % BT;
#pragma unroll
            for (int i = 0; i + j < S; ++i)
                acc[i + j][bt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[g % (AHEAD + 1)], m[i], acc[i + j][bt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // or the scheduler sinks every read to just before its first use
        }
    };
    size_t kb = kb0;
    if (kb < kb1) {
        issue(ma, 0);
        __syncthreads();
    }
    // every trip issues the same loads (past the end of the split the pointers are held: the last block is read again)
    for (; kb + 2 <= kb1; kb += 2) {
        issue(mb, 1);
        compute(ma, 0);
        __syncthreads();
        if (kb + 2 >= kb1) { if (DBG != 1) mp -= S * kSlFrag; xp -= x_step; }
        issue(ma, 0);
        compute(mb, 1);
        __syncthreads();
    }
    if (kb < kb1) compute(ma, 0);
    // D[row = 4 (l >> 4) + reg][col = l & 15]: row -> instance of the tile, col -> matrix row
    const int jr = lane & 15, q = lane >> 4;
    const size_t r = rt * 16 + jr;
    if (live && r < R) {
        const double rs = row_scale[r];
        double* pout = partials + (split * B) * R;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const size_t b = bchunk + bt * 16 + 4 * q + g;
                if (b < B) {
                    double s = 0.0;
#pragma unroll
                    for (int d = S - 1; d >= 0; --d)
                        s += (double)acc[d][bt][g] * __longlong_as_double((long long)(1023 - 12 - 8 * d) << 52);
                    pout[b * R + r] = s * rs * x_scale[b];
                }
            }
    }
}
