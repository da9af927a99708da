// gemm_mfma.hip - matrix-core kernels of the dense linear channel for ensembles (batch > 8).
//
// Both primitives stream a large row-major matrix mat [R, C] exactly once from HBM and
// contract it against a small batch of vectors (<= 64 per pass) on v_mfma_f64_16x16x4_f64
// (f32: v_mfma_f32_16x16x4_f32, same fragment shape; one template, 16-byte pieces of E = 2 / 4 elements):
//     NT: out[b, r] = sum_c mat[r, c] x[b, c]        NN: out[b, c] = sum_r mat[r, c] x[b, r]
// At batch 64 / f64 the contraction is balanced between HBM (R*C*8 bytes) and the matrix
// pipe (2*R*C*64 flops), so the design keeps both busy:
//   * mat goes global -> registers directly in MFMA operand layout (no reuse across
//     workgroups exists, LDS would only add a hop).  The MFMA k index is free to permute
//     as long as A and B agree, which lets every lane fetch 16-byte pieces:
//       NT  lane (j = l&15, q = l>>4) loads mat[r0+j][k0 + 8p + 2q + e]  (p,e in {0,1})
//           -> 4 MFMAs per 16-column step; a row contributes 64 contiguous bytes per load
//       NN  lane (j, q) loads mat[k0 + 4q + s][c0 + 2j + {0,1}], s = 0..3
//           -> even / odd output columns are two MFMA n-tiles; 256 contiguous bytes per row
//   * x (shared by all waves of a workgroup, re-read by every row block) is staged through
//     LDS in fragment order (lane l owns bytes [16 l, 16 l + 16) of a 1 KiB fragment):
//     linear ds_write_b128 / ds_read_b128, no bank conflicts; double-buffered, one barrier
//     per 16-deep k step (32 MFMAs = 2048 matrix-pipe cycles per wave between barriers).
//   * split-K: blockIdx = row_block * S + split with S = 8, so split index == XCD under the
//     round-robin dispatch (speed only): each XCD's L2 holds only its eighth of x.
//     Partial tiles go to a workspace and are folded in fixed order by a second kernel that
//     also applies the fused message-update epilogue -> deterministic results.
// One wave: RT x 16 mat rows (NT) or 32 columns (NN) times up to 64 instances:
// 8 accumulator tiles = 64 VGPRs.
#include "linear.h"
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// Per element type: 16-byte piece type, elements per piece E, accumulator tile, MFMA.
template <typename T> struct Mx;
template <> struct Mx<double> {
    using piece = v2d;
    using acc_t = d4;
    static constexpr int E = 2;
    // C/D fragment: col = lane&15; f64: row = (lane>>4) + 4*reg
    static __device__ __forceinline__ int drow(int q, int g) { return q + 4 * g; }
    static __device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
};
template <> struct Mx<float> {
    using piece = f4;
    using acc_t = f4;
    static constexpr int E = 4;
    // f32 16x16 tiles: row = 4*(lane>>4) + reg
    static __device__ __forceinline__ int drow(int q, int g) { return 4 * q + g; }
    static __device__ __forceinline__ f4 mfma(float a, float b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

constexpr int kSplit = 8;     // == number of XCDs
constexpr int kPieces = 2;    // 16-byte pieces per lane per row (NT) / per x fragment, per step
template <typename T> constexpr int kstep_of() { return 4 * kPieces * Mx<T>::E; }   // f64: 16, f32: 32

// Work item of a workgroup: (tile = row / column block x split, 64-instance chunk).  Ensembles of more than 64
// instances have several chunks per tile, each streaming the same piece of the matrix.  In the one-dimensional
// launch sibling chunks are 8 workgroups apart: the same XCD under the round-robin dispatch and adjacent in
// time, so the second chunk finds the tile in that XCD's L2 and HBM sees the matrix once per pass, not once
// per chunk.  (gridDim.y > 1: the plain two-dimensional launch, tile = blockIdx.x, chunk = blockIdx.y.)
__device__ __forceinline__ void wg_item(size_t B, size_t& tile, size_t& chunk) {
    if (gridDim.y > 1 || B <= 64) {
        tile = blockIdx.x;
        chunk = blockIdx.y;
        return;
    }
    const size_t nchunk = (B + 63) / 64;
    const size_t g = blockIdx.x / kSplit, xcd = blockIdx.x % kSplit;
    chunk = g % nchunk;
    tile = (g / nchunk) * kSplit + xcd;
}

// ------------------------------------------------------------------------------ NT
// Workgroup: 4 waves x (RT*16) rows; grid.x = row_blocks * kSplit; grid.y = batch chunks of 64.
// lane (j = l&15, q = l>>4), piece p, element e  <->  column k0 + p*4E + q*E + e.
// TAIL: the contraction length is not a multiple of the k step (guards only compiled in then).
template <typename T, int BT, int RT, bool TAIL>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 3)))
void gemm_nt_kernel(size_t R, size_t C, size_t B, const T* __restrict__ mat, const T* __restrict__ x,
                    size_t steps_per_split, unsigned nsplit, double* __restrict__ partials) {
    using V = typename Mx<T>::piece;
    using A = typename Mx<T>::acc_t;
    constexpr int E = Mx<T>::E;
    constexpr int KS = kstep_of<T>();
    __shared__ V xs[2][BT][kPieces][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    size_t tile_id, chunk_id;
    wg_item(B, tile_id, chunk_id);
    const size_t rb = tile_id / nsplit, split = tile_id % nsplit;
    const size_t bchunk = chunk_id * 64;
    const size_t nb = (B - bchunk < 64) ? B - bchunk : 64;          // instances in this chunk
    const size_t row0 = rb * (64 * RT) + (size_t)wave * (16 * RT);
    const size_t nsteps_total = (C + KS - 1) / KS;      // a ragged k tail is zero-filled
    const size_t sbeg = split * steps_per_split;
    size_t send = sbeg + steps_per_split;
    if (send > nsteps_total) send = nsteps_total;
    const V vzero = {};
    // lane's mat rows (clamped: out-of-range rows are computed on valid memory and never stored)
    const T* mp[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        size_t r = row0 + rt * 16 + j;
        if (r >= R) r = R - 1;
        mp[rt] = mat + r * C + q * (kPieces * E);
    }
    // x loader: wave w stages batch tile w
    size_t xb = bchunk + (size_t)wave * 16 + j;
    if (xb >= B) xb = B - 1;
    const T* xp = x + xb * C + q * (kPieces * E);
    const bool loads_x = wave < BT;

    A acc[RT][BT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) acc[rt][bt] = A{0, 0, 0, 0};

    // Two register sets for the matrix pieces, used alternately (the k loop is unrolled by two): the loads of step
    // s + 1 are issued before the barrier of step s and first READ after the barrier of step s + 1, a whole MFMA
    // block (64 MFMAs) later.  With one set copied into the other mid-block (the earlier form) the compiler had to
    // wait for those loads a few MFMAs after issuing them: every wave sat in s_waitcnt for an HBM round trip per
    // k step (profiles/r02/pmc_wave_waits_matrix_passes_B64.txt: 17-24 % of the wave cycles).
    V ma[RT][kPieces], mb[RT][kPieces], xg[kPieces];
    // only the last k step of the whole contraction can be ragged: every other step loads unguarded
    // (the guard is a wave-uniform branch, not a per-lane select on every load)
    auto load_step = [&](auto guarded, size_t k, V (&dst)[RT][kPieces]) {
#pragma unroll
        for (int p = 0; p < kPieces; ++p) {
            const bool in = !decltype(guarded)::value || k + p * E + q * (kPieces * E) < C;   // C % E == 0: pieces are all-in or all-out
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                dst[rt][p] = in ? *reinterpret_cast<const V*>(mp[rt] + k + p * E) : vzero;
            xg[p] = in ? *reinterpret_cast<const V*>(xp + k + p * E) : vzero;      // every wave (clamped row): no branch
        }
    };
    // Every call issues the same number of loads on every path - past the end of the split the last step is
    // simply loaded again - and nothing is under a wave-dependent branch: the compiler's s_waitcnt vmcnt(N) is
    // a static count that has to be right for EVERY path into the MFMA block, so a path with fewer loads in
    // flight (last iteration, a wave that stages no x) forces a small N, which in the steady state waits for
    // loads that were issued a moment ago.
    auto load_any = [&](size_t s, V (&dst)[RT][kPieces]) {
        const size_t sc = s < send ? s : send - 1;
        if (TAIL && sc + 1 == nsteps_total) load_step(std::true_type{}, sc * KS, dst);
        else load_step(std::false_type{}, sc * KS, dst);
    };
    auto stage_x = [&](int st) {
        if (BT == 4 || loads_x) {
#pragma unroll
            for (int p = 0; p < kPieces; ++p) xs[st][wave][p][lane] = xg[p];
        }
    };
    auto mfma_block = [&](int st, const V (&m)[RT][kPieces]) {
#pragma unroll
        for (int p = 0; p < kPieces; ++p) {
            if constexpr (sizeof(T) == 8) {
                // f64: the matrix operand is held over the BT MFMAs that share it and the x operand changes every MFMA (the
                // other way round - x held over RT MFMAs - the same launch pair takes 0.6 % longer under the power limit,
                // same bits: tools/ab_matvec_pair.py, three alternations on one box)
                V a[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) a[bt] = xs[st][bt][p][lane];
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int bt = 0; bt < BT; ++bt) acc[rt][bt] = Mx<T>::mfma(a[bt][e], m[rt][p][e], acc[rt][bt]);
            } else {
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    const V a = xs[st][bt][p][lane];
#pragma unroll
                    for (int e = 0; e < E; ++e)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][bt] = Mx<T>::mfma(a[e], m[rt][p][e], acc[rt][bt]);
                }
            }
        }
    };
    size_t s = sbeg;
    if (s < send) load_any(s, ma);
    for (; s + 1 < send; s += 2) {                 // pairs of k steps
        stage_x(0);
        load_any(s + 1, mb);
        __syncthreads();
        mfma_block(0, ma);
        stage_x(1);
        load_any(s + 2, ma);
        __syncthreads();
        mfma_block(1, mb);
    }
    if (s < send) {                                // odd step count: the last one
        stage_x(0);
        __syncthreads();
        mfma_block(0, ma);
    }
    // D[row = drow(l>>4, reg)][col = l&15]: row -> instance, col -> mat row
    double* pout = partials + (split * B) * R;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const size_t r = row0 + rt * 16 + j;
        if (r < R) {
#pragma unroll
            for (int bt = 0; bt < BT; ++bt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const size_t b = bt * 16 + Mx<T>::drow(q, g);
                    if (b < nb) pout[(bchunk + b) * R + r] = (double)acc[rt][bt][g];
                }
        }
    }
}

// ------------------------------------------------------------------------------ NN
// Workgroup: 4 waves x (16 E) columns; rows (the contraction) split kSplit ways.
// lane (j, q), piece p, element e <-> row k0 + q*P*E + p*E + e; a 16-byte piece of that row
// holds E adjacent output columns col0 + E j + (0..E-1) = E MFMA n-tiles.
template <typename T, int BT, bool TAIL>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 3)))
void gemm_nn_kernel(size_t R, size_t C, size_t B, const T* __restrict__ mat, const T* __restrict__ x,
                    size_t steps_per_split, unsigned nsplit, double* __restrict__ partials) {
    using V = typename Mx<T>::piece;
    using A = typename Mx<T>::acc_t;
    constexpr int E = Mx<T>::E;
    constexpr int KS = kstep_of<T>();
    constexpr int S = kPieces * E;            // rows per lane per step
    __shared__ V xs[2][BT][kPieces][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    size_t tile_id, chunk_id;
    wg_item(B, tile_id, chunk_id);
    const size_t cb = tile_id / nsplit, split = tile_id % nsplit;
    const size_t bchunk = chunk_id * 64;
    const size_t nb = (B - bchunk < 64) ? B - bchunk : 64;
    const size_t col0 = cb * (64 * E) + (size_t)wave * (16 * E);
    const size_t nsteps_total = (R + KS - 1) / KS;      // a ragged k tail is zero-filled
    const size_t sbeg = split * steps_per_split;
    size_t send = sbeg + steps_per_split;
    if (send > nsteps_total) send = nsteps_total;
    const V vzero = {};
    size_t cl = col0 + E * j;                   // lane's column group (clamped, C % E == 0)
    if (cl + E > C) cl = C - E;
    const T* mp = mat + cl + (size_t)(S * q) * C;
    size_t xb = bchunk + (size_t)wave * 16 + j;
    if (xb >= B) xb = B - 1;
    const T* xp = x + xb * R + S * q;
    const bool loads_x = wave < BT;

    A acc[E][BT];
#pragma unroll
    for (int nt = 0; nt < E; ++nt)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) acc[nt][bt] = A{0, 0, 0, 0};

    V ma[S], mb[S], xg[kPieces];        // two alternating register sets, as in the NT kernel
    // only the contraction's last step is loaded under guards
    auto load_step = [&](auto guarded, size_t k, V (&dst)[S]) {
        constexpr bool G = decltype(guarded)::value;
#pragma unroll
        for (int s4 = 0; s4 < S; ++s4)
            dst[s4] = (!G || k + S * q + s4 < R) ? *reinterpret_cast<const V*>(mp + (k + s4) * C) : vzero;
#pragma unroll
        for (int p = 0; p < kPieces; ++p)
            xg[p] = (!G || k + S * q + p * E < R) ? *reinterpret_cast<const V*>(xp + k + p * E) : vzero;
    };
    auto load_any = [&](size_t s, V (&dst)[S]) {           // same loads on every path, as in the NT kernel
        const size_t sc = s < send ? s : send - 1;
        if (TAIL && sc + 1 == nsteps_total) load_step(std::true_type{}, sc * KS, dst);
        else load_step(std::false_type{}, sc * KS, dst);
    };
    auto stage_x = [&](int st) {
        if (BT == 4 || loads_x) {
#pragma unroll
            for (int p = 0; p < kPieces; ++p) xs[st][wave][p][lane] = xg[p];
        }
    };
    auto mfma_block = [&](int st, const V (&m)[S]) {
#pragma unroll
        for (int p = 0; p < kPieces; ++p) {
            if constexpr (sizeof(T) == 8) {             // as in the row-dot kernel
                V a[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) a[bt] = xs[st][bt][p][lane];
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int nt = 0; nt < E; ++nt)
#pragma unroll
                        for (int bt = 0; bt < BT; ++bt) acc[nt][bt] = Mx<T>::mfma(a[bt][e], m[p * E + e][nt], acc[nt][bt]);
            } else {
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    const V a = xs[st][bt][p][lane];
#pragma unroll
                    for (int e = 0; e < E; ++e)
#pragma unroll
                        for (int nt = 0; nt < E; ++nt)
                            acc[nt][bt] = Mx<T>::mfma(a[e], m[p * E + e][nt], acc[nt][bt]);
                }
            }
        }
    };
    size_t s = sbeg;
    if (s < send) load_any(s, ma);
    for (; s + 1 < send; s += 2) {                 // pairs of k steps
        stage_x(0);
        load_any(s + 1, mb);
        __syncthreads();
        mfma_block(0, ma);
        stage_x(1);
        load_any(s + 2, ma);
        __syncthreads();
        mfma_block(1, mb);
    }
    if (s < send) {                                // odd step count: the last one
        stage_x(0);
        __syncthreads();
        mfma_block(0, ma);
    }
    double* pout = partials + (split * B) * C;
    const size_t c = col0 + E * j;
    if (c + E <= C) {
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const size_t b = bt * 16 + Mx<T>::drow(q, g);
                if (b < nb) {
#pragma unroll
                    for (int nt = 0; nt < E; nt += 2)
                        *reinterpret_cast<double2*>(pout + (bchunk + b) * C + c + nt) =
                            double2{(double)acc[nt][bt][g], (double)acc[nt + 1][bt][g]};
                }
            }
    }
}

// ------------------------------------------------------------------------------ fold + epilogue
template <typename T>
__device__ __forceinline__ T mfma_epilogue(const LinEpi& e, size_t b, size_t idx, size_t len, double acc) {
    if (e.mode == 0) {
        reinterpret_cast<T*>(e.out)[b * len + idx] = T(acc);
        return T(acc);
    }
    const T bin = reinterpret_cast<const T*>(e.b_in)[b * len + idx];
    const double a_in = e.a_in[b];
    double r = acc;
    if (e.add_b_over_a) r += (double)bin / a_in;
    const double an = e.a_new_raw[b];
    T bn = T(r) * T(a_in + an) - bin;          // b_new = r (a + a_new) - b  (tramp/base.py:253-254)
    if (e.damping != 0.0) {
        const T bo = reinterpret_cast<const T*>(e.b_old)[b * len + idx];
        bn = T(e.damping) * bo + T(1.0 - e.damping) * bn;
    }
    reinterpret_cast<T*>(e.out)[b * len + idx] = bn;
    if (e.r_out) reinterpret_cast<T*>(e.r_out)[b * len + idx] = T(r);
    return bn;
}

// Sums the split-K partials in fixed order and applies the message update.  When the consumer of the
// message is a separable prior (prior_hint.h) the block also leaves sum_i v_i of that prior's
// posterior, so that the prior update needs no pass of its own over the message.
template <typename T>
__global__ __launch_bounds__(kBlock) void mfma_fold_kernel(size_t len, size_t B, int nsplit,
                                                           const double* __restrict__ partials, LinEpi epi) {
    __shared__ double smem[kBlock / kWave];
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t b = blockIdx.y;
    double vi = 0.0;
    PriorHintEval<T> he;
    if (epi.hint.kind) he.setup(epi.hint, b);
    if (idx < len) {
        double s = 0.0;
        for (int k = 0; k < nsplit; ++k) s += partials[((size_t)k * B + b) * len + idx];
        const T bn = mfma_epilogue<T>(epi, b, idx, len, s);
        if (epi.hint.kind) vi = he.v(bn, idx);
    }
    if (epi.hint.kind) {
        vi = block_sum(vi, smem);
        if (threadIdx.x == 0) epi.hint.partials[b * (size_t)epi.hint.nblk + blockIdx.x] = vi;
    }
}

// ------------------------------------------------------------------------------ host side
bool gemm_mfma_supported(int dtype, bool nt, size_t R, size_t C, size_t batch) {
    if (dtype != TRAMP_F64 && dtype != TRAMP_F32) return false;
    if (batch == 0 || batch > 65535 * 64ull) return false;
    // 16-byte pieces must not straddle a row end: the contraction length (and, for NN, the row
    // length) has to be a multiple of the piece width; ragged k tails are zero-filled in-kernel
    const size_t e = dtype == TRAMP_F64 ? 2 : 4;
    if (nt) return (C % e) == 0 && C >= e && R >= 1;
    return (R % e) == 0 && R >= e && (C % e) == 0 && C >= e;
}

// Split-K factor: a multiple of 8 (= XCDs: blockIdx = block * nsplit + split, so a split stays on one
// XCD under the round-robin dispatch).  The workgroups of one pass are all alike, and 3 of them fit a CU
// (150-160 VGPRs), so a launch runs in ceil(#workgroups / 768) rounds: with 8 splits the 50000 x 100000
// row-dot pass has 1568 workgroups = 2.04 rounds and pays for 3 (measured 46.3 TFLOP/s against 52.0 with
// 32 splits = 8.2 -> 9 rounds; profiles/r02/sweep_split_100000.txt).  Take the smallest multiple of 8 whose
// last round is at least 90 % full, else the best one up to 32 (more splits only add partial traffic).
static int pick_split(tramp_ctx* ctx, bool nt, size_t nblocks, size_t nchunk, size_t nsteps, int forced) {
    int s = kSplit;
    if (forced > 0) {
        s = forced;
    } else {
        const double slots = 3.0 * (ctx->num_cu > 0 ? ctx->num_cu : 256);
        double best = -1.0;
        for (int cand = 8; cand <= 32; cand += 8) {
            if ((size_t)cand * 16 > nsteps && cand > 8) break;          // keep >= 16 k-steps per split
            const double rounds = (double)nblocks * cand * (double)nchunk / slots;
            const double eff = rounds / ceil(rounds);
            if (eff > best + 1e-9) {
                best = eff;
                s = cand;
            }
            if (eff >= 0.9) break;
        }
    }
    if ((size_t)s > nsteps) s = (int)(nsteps ? nsteps : 1);
    return s;
}

// launch shape for wg_item: one-dimensional with sibling chunks interleaved when it can be (tiles in groups
// of 8), else (tile, chunk); TRAMP_MFMA_CHUNK_2D=1 forces the latter
static dim3 chunk_grid(size_t ntile, unsigned nchunk) {
    static const int flat = getenv("TRAMP_MFMA_CHUNK_2D") ? !atoi(getenv("TRAMP_MFMA_CHUNK_2D")) : 1;
    if (flat && nchunk > 1 && ntile % kSplit == 0 && ntile * nchunk < (1ull << 31)) return dim3((unsigned)(ntile * nchunk), 1);
    return dim3((unsigned)ntile, nchunk);
}

template <typename T>
static int gemm_mfma_t(tramp_ctx* ctx, bool nt, size_t R, size_t C, size_t batch, const T* mat, const T* x,
                       const LinEpi& epi) {
    constexpr int E = Mx<T>::E;
    const size_t len = nt ? R : C;                  // output length per instance
    const size_t kdim = nt ? C : R;
    const size_t nsteps = (kdim + kstep_of<T>() - 1) / kstep_of<T>();
    static const int split_env = getenv("TRAMP_MFMA_SPLIT") ? atoi(getenv("TRAMP_MFMA_SPLIT")) : 0;
    static const int pad_env = getenv("TRAMP_MFMA_LDS_PAD") ? atoi(getenv("TRAMP_MFMA_LDS_PAD")) : 0;
    const unsigned nchunk0 = (unsigned)((batch + 63) / 64);
    const size_t rows_per_wg = 64 * 2;                 // RT = 2 sixteen-row tiles per wave
    const size_t nblocks = nt ? (R + rows_per_wg - 1) / rows_per_wg : (C + 64 * E - 1) / (64 * E);
    const unsigned nsplit = (unsigned)pick_split(ctx, nt, nblocks, nchunk0, nsteps, split_env);
    const size_t sps = (nsteps + nsplit - 1) / nsplit;
    int rc = tramp_ws_reserve(ctx, (size_t)nsplit * batch * len * sizeof(double));
    if (rc) return rc;
    const unsigned nchunk = (unsigned)((batch + 63) / 64);
    const size_t btail = batch - (size_t)(nchunk - 1) * 64;
    // BT is a compile-time tile count; a ragged last chunk simply masks its stores
    const int bt = (nchunk > 1) ? 4 : (int)((btail + 15) / 16);
    dim3 block(kBlock);
    const bool tail = (kdim % kstep_of<T>()) != 0;
    ctx->last_nsplit = (int)nsplit;
    ctx->last_nchunk = (int)nchunk;
    ctx->last_flat = 0;
#define LAUNCH_NT(BTv, TAILv) hipLaunchKernelGGL((gemm_nt_kernel<T, BTv, RT, TAILv>), grid, block, pad_env, ctx->stream, R, C, batch, mat, x, sps, nsplit, ctx->ws)
#define LAUNCH_NN(BTv, TAILv) hipLaunchKernelGGL((gemm_nn_kernel<T, BTv, TAILv>), grid, block, pad_env, ctx->stream, R, C, batch, mat, x, sps, nsplit, ctx->ws)
#define BY_BT(M_, TAILv)                       \
    switch (bt) {                              \
        case 1: M_(1, TAILv); break;           \
        case 2: M_(2, TAILv); break;           \
        case 3: M_(3, TAILv); break;           \
        default: M_(4, TAILv); break;          \
    }
    if (nt) {
        // 16-row tiles per wave.  Round 3, with the alternating register sets: RT = 2 (160 VGPRs, 3 waves per SIMD)
        // 56.1 TFLOP/s against 53.4 for RT = 4 (256 VGPRs, 2 waves) on the 50000 x 100000 x 64 f64 pass; before
        // (loads waited for inside the MFMA block) it was the other way round, 51 against 54.
        constexpr int RT = 2;
        const size_t nrb = (R + 64 * RT - 1) / (64 * RT);
        dim3 grid = chunk_grid(nrb * nsplit, nchunk);
        ctx->last_flat = nchunk > 1 && grid.y == 1;
        if (tail) { BY_BT(LAUNCH_NT, true) } else { BY_BT(LAUNCH_NT, false) }
    } else {
        const size_t ncb = (C + 64 * E - 1) / (64 * E);
        dim3 grid = chunk_grid(ncb * nsplit, nchunk);
        ctx->last_flat = nchunk > 1 && grid.y == 1;
        if (tail) { BY_BT(LAUNCH_NN, true) } else { BY_BT(LAUNCH_NN, false) }
    }
#undef BY_BT
#undef LAUNCH_NT
#undef LAUNCH_NN
    return mfma_fold(ctx, sizeof(T) == 8 ? TRAMP_F64 : TRAMP_F32, len, batch, (int)nsplit, ctx->ws, epi);
}

int mfma_fold(tramp_ctx* ctx, int dtype, size_t len, size_t batch, int nsplit, const double* partials, const LinEpi& epi) {
    dim3 g2((unsigned)((len + kBlock - 1) / kBlock), (unsigned)batch), block(kBlock);
    LinEpi e2 = epi;
    lin_arm_hint(ctx, e2, len, batch, (int)g2.x);
    if (dtype == TRAMP_F64) hipLaunchKernelGGL((mfma_fold_kernel<double>), g2, block, 0, ctx->stream, len, batch, nsplit, partials, e2);
    else hipLaunchKernelGGL((mfma_fold_kernel<float>), g2, block, 0, ctx->stream, len, batch, nsplit, partials, e2);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

int gemm_mfma(tramp_ctx* ctx, int dtype, bool nt, size_t R, size_t C, size_t batch, const void* mat_d,
              const void* x_d, const LinEpi& epi) {
    if (!gemm_mfma_supported(dtype, nt, R, C, batch)) return tramp_fail(ctx, TRAMP_ENOTSUP, "gemm_mfma", "shape");
    if (dtype == TRAMP_F64)
        return gemm_mfma_t<double>(ctx, nt, R, C, batch, (const double*)mat_d, (const double*)x_d, epi);
    return gemm_mfma_t<float>(ctx, nt, R, C, batch, (const float*)mat_d, (const float*)x_d, epi);
}
