// linear.hip - dense linear channel x = W z (K8, K9) for gfx950.
//
// Reference: tramp/channels/linear/linear_channel.py:58-117.  The reference keeps
// the FULL SVD (U [M,M], zero-padded S [M,N], V [N,N]) and W, 3.75 N^2 elements at
// alpha = 0.5, and forms rz = V diag(1/(az+ax s^2)) (V^T bz + S^T U^T bx), rx = W rz.
// Here only the thin factors are resident, k = min(M,N):
//     Uk [M,k], Vt [k,N] (rows = right singular vectors), sing[k]
// and with t = Vt bz, u = Uk^T bx, D = az + ax sing^2 the same posterior means are
//     rx = Uk [ sing (t + sing u) / D ]                                   (FWD)
//     rz = bz/az + Vt^T [ -ax sing^2 t / (D az) + sing u / D ]   (k <  N)   (BWD)
//     rz =         Vt^T [ (t + sing u) / D ]                     (k == N)
// (W V = U S, so rx never needs W or rz).  0.75 N^2 elements at alpha = 0.5 and
// 2MN + 4M^2 (FWD) / 4MN + 2M^2 (BWD) flops per instance instead of 8MN + ...
//
// Two streaming primitives over a row-major matrix mat [R, C]:
//     NT: out[b, r] = sum_c mat[r, c] x[b, c]    (row dot products)
//     NN: out[b, c] = sum_r mat[r, c] x[b, r]    (column accumulation)
// Small ensembles (batch <= 8) are HBM-bound: VALU kernels stream mat exactly once
// with 16-byte loads (this file).  Larger ensembles use the MFMA kernels of
// gemm_mfma.hip.  Both accept the fused message-update epilogue.
#include "common.h"
#include "linear.h"
#include "tramp_math.h"

// ---------------------------------------------------------------- K9 variances
// LinearChannel.compute_n_eff / compute_forward_variance / compute_backward_variance
// (linear_channel.py:58-67, 91-105) + compute_a_new (tramp/base.py:245-248).
__global__ __launch_bounds__(kBlock) void lin_variance_kernel(const double* __restrict__ s2, int k, int rank,
                                                              double Nx, double Nz, int dir,
                                                              const double* __restrict__ az_d,
                                                              const double* __restrict__ ax_d, double damping,
                                                              const double* __restrict__ a_old,
                                                              double* __restrict__ v_out,
                                                              double* __restrict__ a_new_out,
                                                              double* __restrict__ a_new_raw,
                                                              double* __restrict__ n_eff_out) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    const double az = az_d[b], ax = ax_d[b];
    const double azc = (dir == TRAMP_BWD) ? fmax(1e-11, az) : az;     // :92-94
    double n_eff;
    double ssum = 0.0;
    if (ax == 0.0) {
        n_eff = 0.0;                                                  // :60-62
        for (int i = threadIdx.x; i < rank; i += kBlock) ssum += s2[i];
        ssum = block_sum(ssum, smem);
    } else if (azc / ax == 0.0) {
        n_eff = (double)rank / Nz;                                    // :63-65
    } else {
        const double q = azc / ax;
        double s = 0.0;
        // eight loads in flight per thread (one block per instance walks the whole spectrum: with one load at a time the
        // loop ran at the L2 latency, 68 us at k = 50 000); the additions keep their order
        constexpr int UN = 8;
        int i = threadIdx.x;
        for (; i + (UN - 1) * kBlock < rank; i += UN * kBlock) {
            double v[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) v[j] = s2[i + j * kBlock];
#pragma unroll
            for (int j = 0; j < UN; ++j) s += v[j] / (q + v[j]);
        }
        for (; i < rank; i += kBlock) s += s2[i] / (q + s2[i]);
        s = block_sum(s, smem);
        n_eff = s / Nz;                                               // :66-67
    }
    if (threadIdx.x != 0) return;
    double v, a_in;
    if (dir == TRAMP_FWD) {
        if (ax == 0.0) {
            double s_mean = ssum / (double)rank;                      // :100-102
            v = s_mean * (double)rank / (Nx * az);
        } else {
            v = n_eff / ((Nx / Nz) * ax);                             // :103-105
        }
        a_in = ax;
    } else {
        v = (1.0 - n_eff) / azc;                                      // :95-97
        a_in = az;
    }
    const double an = tmath::a_new_of(v, a_in);
    if (a_new_raw) a_new_raw[b] = an;
    if (a_new_out) a_new_out[b] = (damping != 0.0) ? damping * a_old[b] + (1.0 - damping) * an : an;
    if (v_out) v_out[b] = v;
    if (n_eff_out) n_eff_out[b] = n_eff;
}

// ---------------------------------------------------------------- fused epilogue
template <typename T>
__device__ __forceinline__ T lin_epilogue(const LinEpi& e, size_t b, size_t idx, size_t len, double acc) {
    if (e.mode == 0) {
        reinterpret_cast<T*>(e.out)[b * len + idx] = T(acc);
        return T(acc);
    }
    const T bin = reinterpret_cast<const T*>(e.b_in)[b * len + idx];
    const double a_in = e.a_in[b];
    double r = acc;
    if (e.add_b_over_a) r += (double)bin / a_in;
    const double an = e.a_new_raw[b];
    // b_new = r (a + a_new) - b  (tramp/base.py:253-254)
    T bn = T(r) * T(a_in + an) - bin;
    if (e.damping != 0.0) {
        const T bo = reinterpret_cast<const T*>(e.b_old)[b * len + idx];
        bn = T(e.damping) * bo + T(1.0 - e.damping) * bn;
    }
    reinterpret_cast<T*>(e.out)[b * len + idx] = bn;
    if (e.r_out) reinterpret_cast<T*>(e.r_out)[b * len + idx] = T(r);
    return bn;
}

bool lin_arm_hint(tramp_ctx* ctx, LinEpi& epi, size_t len, size_t batch, int nblk) {
    epi.hint.kind = PRIOR_HINT_NONE;
    if (epi.mode != 1) return false;
    return prior_hint_arm(ctx, epi.out, len, batch, nblk, &epi.hint);
}

// ---------------------------------------------------------------- NT: row dot products
// One wave owns ROWS rows; lanes stride over the columns with 16-byte loads, the
// batch of x vectors comes from L1/L2 (x is tiny next to mat).  f32 accumulates in
// f32, f64 in f64; 64-wide shuffle reduction at the end.
template <typename T, int NB, int ROWS, bool VEC>
__global__ __launch_bounds__(kBlock) void gemv_nt_kernel(size_t R, size_t C, int nb, const T* __restrict__ mat,
                                                         const T* __restrict__ x, size_t b0, LinEpi epi) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const size_t r0 = wave * ROWS;
    if (!epi.hint.kind && r0 >= R) return;          // with fused moments every wave reaches the block reduction
    T acc[ROWS][NB];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[i][b] = T(0);
    const T* rowp[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) rowp[i] = mat + (r0 + i < R ? r0 + i : R - 1) * C;
    if (VEC) {
        const size_t nv = C / VN;
        for (size_t j = lane; j < nv; j += 64) {
            V m[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) m[i] = reinterpret_cast<const V*>(rowp[i])[j];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b < nb) {
                    V xv = reinterpret_cast<const V*>(x + (b0 + b) * C)[j];
#pragma unroll
                    for (int i = 0; i < ROWS; ++i)
#pragma unroll
                        for (int q = 0; q < VN; ++q) acc[i][b] += vget<T>(m[i], q) * vget<T>(xv, q);
                }
            }
        }
    } else {
        for (size_t j = lane; j < C; j += 64) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b < nb) {
                    T xv = x[(b0 + b) * C + j];
#pragma unroll
                    for (int i = 0; i < ROWS; ++i) acc[i][b] += rowp[i][j] * xv;
                }
            }
        }
    }
    if (!epi.hint.kind) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                double s = wave_sum((double)acc[i][b]);
                if (lane == 0 && b < nb && r0 + i < R) lin_epilogue<T>(epi, b0 + b, r0 + i, R, s);
            }
        return;
    }
    // producer-fused moments (prior_hint.h): lane 0 of each wave evaluates v_i of the consumer on the ROWS values it
    // stores, the block leaves one partial per instance
    __shared__ double hsm[kBlock / kWave];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b >= nb) break;                                   // uniform
        PriorHintEval<T> he;
        he.setup(epi.hint, b0 + b);                           // all threads (may stage the erfcx table: a barrier)
        double vi = 0.0;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            double s = wave_sum((double)acc[i][b]);
            if (lane == 0 && r0 + i < R) {
                const T bn = lin_epilogue<T>(epi, b0 + b, r0 + i, R, s);
                vi += he.v(bn, r0 + i);
            }
        }
        __syncthreads();
        if (lane == 0) hsm[threadIdx.x >> 6] = vi;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kBlock / kWave; ++w) t += hsm[w];
            epi.hint.partials[(b0 + b) * (size_t)epi.hint.nblk + blockIdx.x] = t;
        }
    }
}

// ---------------------------------------------------------------- NN: column accumulation
// A thread owns VN adjacent columns; blockIdx.y splits the rows into chunks whose
// partial sums go to the workspace and are folded (fixed order) by nn_fold_kernel,
// which also applies the epilogue.  x[b, r] is wave-uniform (scalar loads).
template <typename T, int NB, bool VEC>
__global__ __launch_bounds__(kBlock) void gemv_nn_kernel(size_t R, size_t C, int nb, size_t rows_per_chunk,
                                                         const T* __restrict__ mat, const T* __restrict__ x,
                                                         size_t b0, double* __restrict__ partials) {
    using V = typename Vec<T>::type;
    constexpr int VN = VEC ? Vec<T>::N : 1;
    const size_t cv = (size_t)blockIdx.x * kBlock + threadIdx.x;      // vector-column index
    const size_t c0 = cv * VN;
    if (c0 >= C) return;
    const size_t rbeg = (size_t)blockIdx.y * rows_per_chunk;
    size_t rend = rbeg + rows_per_chunk;
    if (rend > R) rend = R;
    T acc[NB][VN];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < VN; ++q) acc[b][q] = T(0);
    constexpr int UNR = 4;
    size_t r = rbeg;
    for (; r + UNR <= rend; r += UNR) {
        T m[UNR][VN];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (VEC) {
                V mv = reinterpret_cast<const V*>(mat + (r + u) * C)[cv];
#pragma unroll
                for (int q = 0; q < VN; ++q) m[u][q] = vget<T>(mv, q);
            } else {
                m[u][0] = mat[(r + u) * C + c0];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b < nb) {
                    const T xv = x[(b0 + b) * R + r + u];
#pragma unroll
                    for (int q = 0; q < VN; ++q) acc[b][q] += m[u][q] * xv;
                }
            }
    }
    for (; r < rend; ++r) {
#pragma unroll
        for (int q = 0; q < VN; ++q) {
            const T mv = mat[r * C + c0 + q];
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (b < nb) acc[b][q] += mv * x[(b0 + b) * R + r];
        }
    }
    // partials[chunk][b][c]
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b < nb)
#pragma unroll
            for (int q = 0; q < VN; ++q)
                partials[((size_t)blockIdx.y * nb + b) * C + c0 + q] = (double)acc[b][q];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void nn_fold_kernel(size_t C, int nb, int nchunk,
                                                         const double* __restrict__ partials, size_t b0,
                                                         LinEpi epi) {
    __shared__ double smem[kBlock / kWave];
    const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const int b = blockIdx.y;
    double vi = 0.0;
    PriorHintEval<T> he;
    if (epi.hint.kind) he.setup(epi.hint, b0 + b);
    if (c < C) {
        double s = 0.0;
        for (int k = 0; k < nchunk; ++k) s += partials[((size_t)k * nb + b) * C + c];
        const T bn = lin_epilogue<T>(epi, b0 + b, c, C, s);
        if (epi.hint.kind) vi = he.v(bn, c);
    }
    if (epi.hint.kind) {                       // uniform across the block
        vi = block_sum(vi, smem);
        if (threadIdx.x == 0) epi.hint.partials[(b0 + b) * (size_t)epi.hint.nblk + blockIdx.x] = vi;
    }
}

// ---------------------------------------------------------------- launchers (VALU path)
template <typename T, int NB>
static void launch_nt(tramp_ctx* ctx, size_t R, size_t C, int nb, const T* mat, const T* x, size_t b0,
                      const LinEpi& epi) {
    constexpr int ROWS = (NB >= 8) ? 2 : 4;
    const size_t waves = (R + ROWS - 1) / ROWS;
    dim3 grid((unsigned)((waves + 3) / 4)), block(kBlock);
    if (C % Vec<T>::N == 0)
        hipLaunchKernelGGL((gemv_nt_kernel<T, NB, ROWS, true>), grid, block, 0, ctx->stream, R, C, nb, mat, x, b0, epi);
    else
        hipLaunchKernelGGL((gemv_nt_kernel<T, NB, ROWS, false>), grid, block, 0, ctx->stream, R, C, nb, mat, x, b0, epi);
}

template <typename T, int NB>
static int launch_nn(tramp_ctx* ctx, size_t R, size_t C, int nb, const T* mat, const T* x, size_t b0,
                     const LinEpi& epi) {
    const bool vec = (C % Vec<T>::N == 0);
    const int VN = vec ? Vec<T>::N : 1;
    const size_t colthreads = (C + VN - 1) / VN;
    const size_t colwaves = (colthreads + 63) / 64;
    size_t nchunk = (4096 + colwaves - 1) / colwaves;
    if (nchunk > R / 16) nchunk = R / 16;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 256) nchunk = 256;
    const size_t rpc = (R + nchunk - 1) / nchunk;
    nchunk = (R + rpc - 1) / rpc;
    int rc = tramp_ws_reserve(ctx, nchunk * (size_t)nb * C * sizeof(double));
    if (rc) return rc;
    dim3 grid((unsigned)((colthreads + kBlock - 1) / kBlock), (unsigned)nchunk), block(kBlock);
    if (vec)
        hipLaunchKernelGGL((gemv_nn_kernel<T, NB, true>), grid, block, 0, ctx->stream, R, C, nb, rpc, mat, x, b0, ctx->ws);
    else
        hipLaunchKernelGGL((gemv_nn_kernel<T, NB, false>), grid, block, 0, ctx->stream, R, C, nb, rpc, mat, x, b0, ctx->ws);
    dim3 g2((unsigned)((C + kBlock - 1) / kBlock), (unsigned)nb);
    hipLaunchKernelGGL((nn_fold_kernel<T>), g2, block, 0, ctx->stream, C, nb, (int)nchunk, ctx->ws, b0, epi);
    return TRAMP_OK;
}

template <typename T>
int matvec_valu(tramp_ctx* ctx, bool nt, size_t R, size_t C, size_t batch, const void* mat_d, const void* x_d,
                const LinEpi& epi_in) {
    const T* mat = reinterpret_cast<const T*>(mat_d);
    const T* x = reinterpret_cast<const T*>(x_d);
    LinEpi epi = epi_in;
    epi.hint.kind = PRIOR_HINT_NONE;
    if (!nt) lin_arm_hint(ctx, epi, C, batch, (int)((C + kBlock - 1) / kBlock));   // nn_fold_kernel's grid
    else if (batch <= 4) lin_arm_hint(ctx, epi, R, batch, (int)(((R + 3) / 4 + 3) / 4));   // gemv_nt_kernel's grid (ROWS = 4)
    for (size_t b0 = 0; b0 < batch; b0 += TRAMP_GEMV_MAX_BATCH) {
        const int nb = (int)((batch - b0 < TRAMP_GEMV_MAX_BATCH) ? batch - b0 : TRAMP_GEMV_MAX_BATCH);
        int rc = TRAMP_OK;
        if (nt) {
            if (nb == 1) launch_nt<T, 1>(ctx, R, C, nb, mat, x, b0, epi);
            else if (nb == 2) launch_nt<T, 2>(ctx, R, C, nb, mat, x, b0, epi);
            else if (nb <= 4) launch_nt<T, 4>(ctx, R, C, nb, mat, x, b0, epi);
            else launch_nt<T, 8>(ctx, R, C, nb, mat, x, b0, epi);
        } else {
            if (nb == 1) rc = launch_nn<T, 1>(ctx, R, C, nb, mat, x, b0, epi);
            else if (nb == 2) rc = launch_nn<T, 2>(ctx, R, C, nb, mat, x, b0, epi);
            else if (nb <= 4) rc = launch_nn<T, 4>(ctx, R, C, nb, mat, x, b0, epi);
            else rc = launch_nn<T, 8>(ctx, R, C, nb, mat, x, b0, epi);
        }
        if (rc) return rc;
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

template int matvec_valu<float>(tramp_ctx*, bool, size_t, size_t, size_t, const void*, const void*, const LinEpi&);
template int matvec_valu<double>(tramp_ctx*, bool, size_t, size_t, size_t, const void*, const void*, const LinEpi&);

static int matvec_dispatch(tramp_ctx* ctx, int dtype, int algo, bool nt, size_t R, size_t C, size_t batch,
                           const void* mat_d, const void* x_d, const LinEpi& epi) {
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, mat_d && x_d && epi.out);
    if (!R || !C || !batch) return TRAMP_OK;
    // measured on MI355X: the VALU GEMV wins up to 2 instances (6.1 vs 5.7 TB/s), MFMA from 3 on
    static const int algo_env = getenv("TRAMP_MATVEC_ALGO") ? atoi(getenv("TRAMP_MATVEC_ALGO")) : 0;   // debugging
    if (algo == 0 && algo_env) algo = algo_env;
    ProfScope prof(ctx, nt ? PROF_MATVEC_NT : PROF_MATVEC_NN, (double)R * (double)C * dtype_size(dtype),
                   2.0 * (double)R * (double)C * (double)batch);
    if (algo == 3 || algo == 0) {      // an attached int8-digit copy of this matrix (gemm_sliced.hip) takes the pass, at any batch:
                                       // 7 bytes per element instead of 8 even where the pass only streams
        bool taken = false;
        const int rc = gemm_sliced_try(ctx, dtype, nt, R, C, batch, mat_d, x_d, epi, &taken);
        if (taken) return rc;
        if (algo == 3) return tramp_fail(ctx, TRAMP_ENOTSUP, "digit pass", "no digit copy of this matrix is attached in this form");
    }
    bool mfma = (algo == 2) || (algo == 0 && batch > 2);
    if (mfma && !gemm_mfma_supported(dtype, nt, R, C, batch)) {
        if (algo == 2) return tramp_fail(ctx, TRAMP_ENOTSUP, "mfma path", "shape/dtype not supported");
        mfma = false;
    }
    if (mfma) return gemm_mfma(ctx, dtype, nt, R, C, batch, mat_d, x_d, epi);
    if (dtype == TRAMP_F64) return matvec_valu<double>(ctx, nt, R, C, batch, mat_d, x_d, epi);
    return matvec_valu<float>(ctx, nt, R, C, batch, mat_d, x_d, epi);
}

// out = mat x over the ROWS of mat [R, C] (a row-dot pass).  (Round 3 measured the same product as a column pass over a
// resident transposed copy: 57.9 against 55.8 TFLOP/s for the pass alone, 23.29 against 23.24 ms for the EP iteration -
// both passes sit on the package power limit - at 40 GB of extra HBM; DESIGN 9.2.  The copies left the library.)
static int rowdot_dispatch(tramp_ctx* ctx, int dtype, size_t R, size_t C, size_t batch, const void* mat_d,
                           const void* x_d, const LinEpi& epi) {
    return matvec_dispatch(ctx, dtype, 0, true, R, C, batch, mat_d, x_d, epi);
}

// dst [C, R] = src [R, C]^T through 32 x 32 LDS tiles (tramp_transpose: a utility of the set-up path)
template <typename T>
__global__ __launch_bounds__(kBlock) void transpose_kernel(size_t R, size_t C, const T* __restrict__ src, T* __restrict__ dst) {
    __shared__ T tile[32][33];
    const size_t c0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) dst[(c0 + i) * R + r0 + tx] = tile[tx][i];
}

extern "C" int tramp_transpose(tramp_ctx* ctx, int dtype, size_t R, size_t C, const void* src_d, void* dst_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, src_d && dst_d && src_d != dst_d && (dtype == TRAMP_F32 || dtype == TRAMP_F64));
    if (!R || !C) return TRAMP_OK;
    const size_t gx = (C + 31) / 32, gy = (R + 31) / 32;
    if (gy > 65535 || gx > 0x7fffffffull) return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_transpose", "more than 2 million rows");
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((transpose_kernel<double>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, R, C,
                           (const double*)src_d, (double*)dst_d);
    else
        hipLaunchKernelGGL((transpose_kernel<float>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, R, C,
                           (const float*)src_d, (float*)dst_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_matvec_nt(tramp_ctx* ctx, int dtype, int algo, size_t R, size_t C, size_t batch,
                               const void* mat_d, const void* x_d, void* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    LinEpi epi{};
    epi.mode = 0;
    epi.out = out_d;
    return matvec_dispatch(ctx, dtype, algo, true, R, C, batch, mat_d, x_d, epi);
}

extern "C" int tramp_matvec_nn(tramp_ctx* ctx, int dtype, int algo, size_t R, size_t C, size_t batch,
                               const void* mat_d, const void* x_d, void* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    LinEpi epi{};
    epi.mode = 0;
    epi.out = out_d;
    return matvec_dispatch(ctx, dtype, algo, false, R, C, batch, mat_d, x_d, epi);
}

extern "C" int tramp_matvec_last_launch(tramp_ctx* ctx, int* nsplit, int* nchunk, int* flat_chunks) {
    if (!ctx) return TRAMP_EINVAL;
    if (nsplit) *nsplit = ctx->last_nsplit;
    if (nchunk) *nchunk = ctx->last_nchunk;
    if (flat_chunks) *flat_chunks = ctx->last_flat;
    return TRAMP_OK;
}

// ---------------------------------------------------------------- spectral coefficients
// w = f(t, u) on [batch, k]; tiny next to the matrix passes.
template <typename T>
__global__ __launch_bounds__(kBlock) void lin_mid_kernel(int k, int mode, const double* __restrict__ sing,
                                                         const double* __restrict__ az_d,
                                                         const double* __restrict__ ax_d, const T* __restrict__ t,
                                                         const T* __restrict__ u, T* __restrict__ w) {
    const size_t b = blockIdx.y;
    const double az = az_d[b], ax = ax_d[b];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < k; i += gridDim.x * kBlock) {
        const double s = sing[i];
        const double D = az + ax * (s * s);
        const double ti = (double)t[b * k + i], ui = (double)u[b * k + i];
        double o;
        if (mode == 0) o = s * (ti + s * ui) / D;                               // FWD
        else if (mode == 1) o = (-(ax * (s * s)) * ti / az + s * ui) / D;       // BWD, k < N
        else o = (ti + s * ui) / D;                                             // BWD, k == N
        w[b * k + i] = T(o);
    }
}

// ---------------------------------------------------------------- handle + message
extern "C" int tramp_linear_create(tramp_ctx* ctx, int dtype, size_t M, size_t N, size_t rank, const void* Uk_d,
                                   const void* Vt_d, const double* sing_d, tramp_linear** out) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, out && Uk_d && Vt_d && sing_d && M > 0 && N > 0);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    const size_t k = M < N ? M : N;
    TRAMP_REQUIRE(ctx, rank <= k);
    tramp_linear* lin = new (std::nothrow) tramp_linear();
    if (!lin) return TRAMP_ENOMEM;
    memset(lin, 0, sizeof(*lin));
    lin->dtype = dtype;
    lin->M = M;
    lin->N = N;
    lin->k = k;
    lin->rank = rank;
    lin->Uk = Uk_d;
    lin->Vt = Vt_d;
    lin->sing = sing_d;
    if (hipMalloc((void**)&lin->s2, k * sizeof(double)) != hipSuccess) {
        delete lin;
        return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_linear_create", "hipMalloc");
    }
    // spectrum = sing^2  (diag(S^T S), linear_channel.py:41)
    int rc = lin_square(ctx, sing_d, lin->s2, k);
    if (rc) {
        (void)hipFree(lin->s2);
        delete lin;
        return rc;
    }
    *out = lin;
    return TRAMP_OK;
}

extern "C" int tramp_linear_destroy(tramp_ctx* ctx, tramp_linear* lin) {
    if (!ctx || !lin) return TRAMP_EINVAL;
    (void)hipStreamSynchronize(ctx->stream);
    if (lin->s2) (void)(void)hipFree(lin->s2);
    if (lin->work) (void)hipFree(lin->work);
    delete lin;
    return TRAMP_OK;
}

__global__ void square_kernel(const double* in, double* out, size_t n) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = in[i] * in[i];
}

int lin_square(tramp_ctx* ctx, const double* in, double* out, size_t n) {
    hipLaunchKernelGGL(square_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, in, out, n);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

static int lin_reserve(tramp_ctx* ctx, tramp_linear* lin, size_t batch) {
    if (batch <= lin->work_batch) return TRAMP_OK;
    if (ctx->capturing)
        return tramp_fail(ctx, TRAMP_EINVAL, "linear workspace", "cannot allocate during graph capture");
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (lin->work) TRAMP_HIP_CHECK(ctx, hipFree(lin->work));
    lin->work = nullptr;
    lin->work_batch = 0;
    const size_t es = dtype_size(lin->dtype);
    const size_t vec = ((batch * lin->k * es + 255) / 256) * 256;
    const size_t sc = ((batch * sizeof(double) + 255) / 256) * 256;
    if (hipMalloc(&lin->work, 5 * vec + sc) != hipSuccess) {
        (void)hipGetLastError();
        return tramp_fail(ctx, TRAMP_ENOMEM, "linear workspace", "hipMalloc");
    }
    char* p = (char*)lin->work;
    lin->t = p;
    lin->u = p + vec;
    lin->w = p + 2 * vec;
    lin->tc = p + 3 * vec;
    lin->uc = p + 4 * vec;
    lin->tc_valid = lin->uc_valid = false;
    lin->a_raw = (double*)(p + 5 * vec);
    lin->work_batch = batch;
    return TRAMP_OK;
}

extern "C" int tramp_linear_variance(tramp_ctx* ctx, const tramp_linear* lin, int dir, size_t batch,
                                     const double* az_d, const double* ax_d, double* v_d, double* a_new_d,
                                     double* n_eff_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, lin && az_d && ax_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    if (!batch) return TRAMP_OK;
    hipLaunchKernelGGL(lin_variance_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, lin->s2, (int)lin->k,
                       (int)lin->rank, (double)lin->M, (double)lin->N, dir, az_d, ax_d, 0.0, (const double*)nullptr,
                       v_d, a_new_d, (double*)nullptr, n_eff_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

int lin_variance_launch(tramp_ctx* ctx, const double* s2, int k, int rank, double Nx, double Nz, int dir, size_t batch,
                        const double* az_d, const double* ax_d, double damping, const double* a_old_d, double* v_d,
                        double* a_new_d, double* a_raw_d, double* n_eff_d) {
    hipLaunchKernelGGL(lin_variance_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, s2, k, rank, Nx, Nz, dir,
                       az_d, ax_d, damping, a_old_d, v_d, a_new_d, a_raw_d, n_eff_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- rotated output basis
// When the factor behind the channel's output is rotation invariant (an isotropic Gaussian likelihood),
// the whole problem can be written for x' = Uk^T x instead of x (Uk square: M = k): the likelihood keeps
// its form with y' = Uk^T y, every message on that side is the rotated one, norms and variances are
// unchanged - and the channel becomes z -> S Vt z, so the two passes over Uk (u = Uk^T bx, rx = Uk w) drop
// out of the iteration: u = bx', rx' = w.  The caller rotates y once and rotates results back when read.
//
// The mirror image on the INPUT side (bit 1): when the factor behind the channel's input is an isotropic
// zero-mean Gaussian prior, the input variable z is carried as z' = Vt z (k numbers per instance).  The part
// of z orthogonal to the rows of Vt sees only that prior, whose message has b = 0, so it stays exactly zero
// in every message and posterior mean; t = Vt bz is bz' itself and the back-projection Vt^T w is not formed
// (rz' = bz' / az + w).  Scalars (variances, n_eff) keep using N.  The caller expands z = Vt^T z' when read.
extern "C" int tramp_linear_set_rotated(tramp_ctx* ctx, tramp_linear* lin, int on) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, lin && on >= 0 && on <= 3);
    if ((on & 1) && lin->k != lin->M)
        return tramp_fail(ctx, TRAMP_EINVAL, "tramp_linear_set_rotated", "needs M <= N (square left factor)");
    lin->rot = on;
    lin->proj_batch = 0;                      // cached projections belong to the other basis
    lin->tc_valid = lin->uc_valid = false;
    return TRAMP_OK;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void lin_epi_apply_kernel(size_t len, const T* __restrict__ w, LinEpi epi) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    double vi = 0.0;
    PriorHintEval<T> he;
    if (epi.hint.kind) he.setup(epi.hint, b);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (size_t)gridDim.x * kBlock) {
        const T bn = lin_epilogue<T>(epi, b, i, len, (double)w[b * len + i]);
        if (epi.hint.kind) vi += he.v(bn, i);
    }
    if (epi.hint.kind) {
        vi = block_sum(vi, smem);
        if (threadIdx.x == 0) epi.hint.partials[b * (size_t)epi.hint.nblk + blockIdx.x] = vi;
    }
}

static int lin_copy_u(tramp_ctx* ctx, tramp_linear* lin, size_t batch, const void* bx_d) {
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(lin->u, bx_d, batch * lin->k * dtype_size(lin->dtype), hipMemcpyDeviceToDevice,
                                        ctx->stream));
    return TRAMP_OK;
}

// Brings the handle's projections t = Vt bz and u = Uk^T bx up to date (shared by the message
// update and the log-partition); `reuse` as in tramp_linear_message.
int lin_project(tramp_ctx* ctx, tramp_linear* lin, size_t batch, const void* bz_d, const void* bx_d, int reuse) {
    const void* work_before = lin->work;
    int rc = lin_reserve(ctx, lin, batch);
    if (rc) return rc;
    if (lin->work != work_before || lin->proj_batch != batch) reuse = 0;
    lin->proj_batch = batch;
    LinEpi plain{};
    plain.mode = 0;
    plain.out = lin->t;
    if (lin->rot & 2) {
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(lin->t, bz_d, batch * lin->k * dtype_size(lin->dtype),
                                            hipMemcpyDeviceToDevice, ctx->stream));
    } else if (!(reuse & TRAMP_REUSE_T)) {
        rc = rowdot_dispatch(ctx, lin->dtype, lin->k, lin->N, batch, lin->Vt, bz_d, plain);
        if (rc) return rc;
    }
    plain.out = lin->u;
    if (lin->rot & 1) {
        rc = lin_copy_u(ctx, lin, batch, bx_d);
        if (rc) return rc;
    } else if (!(reuse & TRAMP_REUSE_U)) {
        rc = matvec_dispatch(ctx, lin->dtype, 0, false, lin->M, lin->k, batch, lin->Uk, bx_d, plain);
        if (rc) return rc;
    }
    return TRAMP_OK;
}

// p <- d p + (1 - d) pc over [count] elements
template <typename T>
__global__ __launch_bounds__(kBlock) void axpby_kernel(size_t count, T d, T omd, T* __restrict__ p,
                                                       const T* __restrict__ pc) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        p[i] = d * p[i] + omd * pc[i];
}

extern "C" int tramp_linear_project_affine(tramp_ctx* ctx, tramp_linear* lin, int which, size_t batch,
                                           const void* c_d, double damping, int c_changed) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, lin && c_d);
    TRAMP_REQUIRE(ctx, which == TRAMP_REUSE_T || which == TRAMP_REUSE_U);
    if (!batch) return TRAMP_OK;
    if (!lin->work || lin->proj_batch != batch || batch > lin->work_batch)
        return tramp_fail(ctx, TRAMP_EINVAL, "tramp_linear_project_affine",
                          "the handle holds no projection of the message being updated");
    const bool is_t = which == TRAMP_REUSE_T;
    if (lin->rot & (is_t ? 2 : 1)) return TRAMP_OK;  // the projection IS the (already damped) message in a rotated basis
    bool& valid = is_t ? lin->tc_valid : lin->uc_valid;
    void* pc = is_t ? lin->tc : lin->uc;
    void* p = is_t ? lin->t : lin->u;
    if (c_changed || !valid) {
        LinEpi plain{};
        plain.mode = 0;
        plain.out = pc;
        int rc = is_t ? rowdot_dispatch(ctx, lin->dtype, lin->k, lin->N, batch, lin->Vt, c_d, plain)
                      : matvec_dispatch(ctx, lin->dtype, 0, false, lin->M, lin->k, batch, lin->Uk, c_d, plain);
        if (rc) return rc;
        valid = true;
    }
    const size_t count = batch * lin->k;
    const unsigned g = (unsigned)((count + kBlock * 4 - 1) / (kBlock * 4) > 4096 ? 4096
                                                                                : (count + kBlock * 4 - 1) / (kBlock * 4));
    ProfScope ps(ctx, PROF_LIN_SCALARS, 3.0 * count * dtype_size(lin->dtype), 0);
    if (lin->dtype == TRAMP_F64)
        hipLaunchKernelGGL((axpby_kernel<double>), dim3(g), dim3(kBlock), 0, ctx->stream, count, damping,
                           1.0 - damping, (double*)p, (const double*)pc);
    else
        hipLaunchKernelGGL((axpby_kernel<float>), dim3(g), dim3(kBlock), 0, ctx->stream, count, (float)damping,
                           (float)(1.0 - damping), (float*)p, (const float*)pc);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_linear_message(tramp_ctx* ctx, tramp_linear* lin, int dir, size_t batch, const double* az_d,
                                    const void* bz_d, const double* ax_d, const void* bx_d, double damping,
                                    const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d,
                                    void* r_d, double* v_d, int reuse) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, lin && az_d && bz_d && ax_d && bx_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!batch) return TRAMP_OK;
    const void* work_before = lin->work;
    int rc = lin_reserve(ctx, lin, batch);
    if (rc) return rc;
    if (lin->work != work_before || lin->proj_batch != batch) reuse = 0;   // projections are gone
    lin->proj_batch = batch;
    const int dtype = lin->dtype;
    const size_t M = lin->M, N = lin->N, k = lin->k;
    // 1. scalars: v, a_new (K9)
    hipLaunchKernelGGL(lin_variance_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, lin->s2, (int)k,
                       (int)lin->rank, (double)M, (double)N, dir, az_d, ax_d, damping, a_old_d, v_d, a_new_d,
                       lin->a_raw, (double*)nullptr);
    // 2. t = Vt bz  [batch, k]
    LinEpi plain{};
    plain.mode = 0;
    plain.out = lin->t;
    const void* t_ptr = lin->t;
    if (lin->rot & 2) {
        t_ptr = bz_d;                                     // t = Vt bz is bz' itself
    } else if (!(reuse & TRAMP_REUSE_T)) {
        rc = rowdot_dispatch(ctx, dtype, k, N, batch, lin->Vt, bz_d, plain);
        if (rc) return rc;
    }
    // 3. u = Uk^T bx  [batch, k]
    plain.out = lin->u;
    const void* u_ptr = lin->u;
    if (lin->rot & 1) {
        u_ptr = bx_d;                                     // u = Uk^T bx is bx' itself in the rotated basis
    } else if (!(reuse & TRAMP_REUSE_U)) {
        rc = matvec_dispatch(ctx, dtype, 0, false, M, k, batch, lin->Uk, bx_d, plain);
        if (rc) return rc;
    }
    // 4. spectral coefficients
    const int mode = (dir == TRAMP_FWD) ? 0 : (k < N ? 1 : 2);
    dim3 gm((unsigned)((k + kBlock - 1) / kBlock > 1024 ? 1024 : (k + kBlock - 1) / kBlock), (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((lin_mid_kernel<double>), gm, dim3(kBlock), 0, ctx->stream, (int)k, mode, lin->sing, az_d,
                           ax_d, (const double*)t_ptr, (const double*)u_ptr, (double*)lin->w);
    else
        hipLaunchKernelGGL((lin_mid_kernel<float>), gm, dim3(kBlock), 0, ctx->stream, (int)k, mode, lin->sing, az_d,
                           ax_d, (const float*)t_ptr, (const float*)u_ptr, (float*)lin->w);
    // 5. back-projection with the fused message update
    LinEpi epi{};
    epi.mode = 1;
    epi.a_new_raw = lin->a_raw;
    epi.damping = damping;
    epi.b_old = (damping != 0.0) ? b_old_d : nullptr;
    epi.out = b_new_d;
    epi.r_out = r_d;
    if (dir == TRAMP_FWD) {
        epi.a_in = ax_d;
        epi.b_in = bx_d;
        epi.add_b_over_a = 0;
        if (lin->rot & 1) {                                                                 // rx' = w
            ProfScope ps(ctx, PROF_EW, (double)batch * k * dtype_size(dtype) * (damping != 0.0 ? 4 : 3), 0);
            dim3 ge((unsigned)((k + kBlock * 4 - 1) / (kBlock * 4) > 1024 ? 1024 : (k + kBlock * 4 - 1) / (kBlock * 4)),
                    (unsigned)batch);
            if (dtype == TRAMP_F64)
                hipLaunchKernelGGL((lin_epi_apply_kernel<double>), ge, dim3(kBlock), 0, ctx->stream, k,
                                   (const double*)lin->w, epi);
            else
                hipLaunchKernelGGL((lin_epi_apply_kernel<float>), ge, dim3(kBlock), 0, ctx->stream, k,
                                   (const float*)lin->w, epi);
            rc = TRAMP_OK;
        } else
            rc = rowdot_dispatch(ctx, dtype, M, k, batch, lin->Uk, lin->w, epi);   // rx = Uk w
    } else {
        epi.a_in = az_d;
        epi.b_in = bz_d;
        epi.add_b_over_a = (k < N) ? 1 : 0;
        if (lin->rot & 2) {                                                                 // rz' = bz' / az + w
            ProfScope ps(ctx, PROF_EW, (double)batch * k * dtype_size(dtype) * (damping != 0.0 ? 4 : 3), 0);
            dim3 ge((unsigned)((k + kBlock * 4 - 1) / (kBlock * 4) > 1024 ? 1024 : (k + kBlock * 4 - 1) / (kBlock * 4)),
                    (unsigned)batch);
            lin_arm_hint(ctx, epi, k, batch, (int)ge.x);
            if (dtype == TRAMP_F64)
                hipLaunchKernelGGL((lin_epi_apply_kernel<double>), ge, dim3(kBlock), 0, ctx->stream, k,
                                   (const double*)lin->w, epi);
            else
                hipLaunchKernelGGL((lin_epi_apply_kernel<float>), ge, dim3(kBlock), 0, ctx->stream, k,
                                   (const float*)lin->w, epi);
            rc = TRAMP_OK;
        } else
            rc = matvec_dispatch(ctx, dtype, 0, false, k, N, batch, lin->Vt, lin->w, epi);  // rz = Vt^T w
    }
    if (rc) return rc;
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}
