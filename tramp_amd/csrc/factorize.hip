// factorize.hip - device-side thin SVD of W for LinearChannel setup.
//
// Replaces LinearChannel.__init__'s two host SVDs (np.linalg.matrix_rank + full
// np.linalg.svd, tramp/channels/linear/linear_channel.py:8-15,37-41) by a Gram-matrix
// eigendecomposition on the smaller side - library calls are fine here, this is
// one-time setup, not the per-iteration path:
//     M <= N:  G = W W^T  -> eigh -> U, sigma^2;  Vt = Sigma^-1 U^T W
//     M >  N:  G = W^T W  -> eigh -> V, sigma^2;  Uk = W V Sigma^-1
// Only Uk, Vt and sigma leave; W itself is not needed by the message kernels.
//
// eigh.  rocSOLVER 7.2's syevd indexes the k x k matrix with 32-bit ints and faults for
// k > 46340 (BASELINE configs[3]: k = 50000).  Above that cap (TRAMP_EIG_CAP, default 46336)
// the symmetric eigenproblem is halved by a spectral divide and conquer built from
// rocBLAS level-3 calls and small Cholesky factorisations only:
//     sigma = trace(G)/k;  S = sign(G - sigma I) by the dynamically weighted Halley iteration
//     in its Cholesky form  X <- (b/c) X + (a - b/c) X (I + c X^2)^-1   (syrk, potrf, 2 trsm);
//     the weights (a, b, c) are the optimal ones for the lower bound l = sqrt(1 - |X^2 - I|_F)
//     that the residual itself certifies, floored at l = 1e-3 so that c <= 1.6e4 and the
//     Cholesky solve stays accurate (tools/proto_eig_dc.py: backward error 1e-15);
//     P = (I + S)/2 is the spectral projector above sigma, r = trace(P);
//     V1 = CholeskyQR2(P[:, :r]), V2 = CholeskyQR2((I - P)[:, r:])  (their Gram matrices are
//     sub-blocks of P itself because P^2 = P);
//     V1^T G V1 and V2^T G V2 (r and k - r wide) go to the library eigensolver - or recurse -
//     and the eigenvectors are rotated back with one gemm each.
// Every library call sees a sub-block whose (columns x leading dimension) stays below 2^31,
// so nothing depends on 64-bit entry points.
//
// Conditioning.  The Gram form resolves singular values only down to ~1e-8 sigma_max.  The rows
// of U^T W that belong to eigenvalues below 1e-10 lambda_max are therefore factorised a second
// time from their own (small) Gram matrix, whose entries are formed without cancellation; the rank
// then uses NumPy's tolerance sigma_max * max(M,N) * eps (np.linalg.matrix_rank, linear_channel.py:37).
// All arithmetic in f64; f32 inputs/outputs are converted at the boundary.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

constexpr double kEps = 2.220446049250313e-16;

// ------------------------------------------------------------------ small kernels
// NOTE: gridDim.x * blockDim.x wraps at 2^32 in HIP, and k*N = 5e9 at BASELINE configs[3]: every
// elementwise kernel here is a grid-stride loop over a fixed grid (kRedBlocks blocks).
#define GRID_STRIDE(idx, total) \
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * kBlock)

// Uk[i, j] = Q[i + (n-1-j) n]   (eigenvector columns, descending eigenvalue order), M <= N case
__global__ void eigvec_to_uk(const double* __restrict__ Q, size_t n, double* __restrict__ Uk) {
    GRID_STRIDE(idx, n * n) {
        size_t i = idx / n, j = idx % n;
        Uk[idx] = Q[i + (n - 1 - j) * n];
    }
}

// Vt[j, c] = Q[c + (n-1-j) n]   (row reversal), M > N case
__global__ void eigvec_to_vt(const double* __restrict__ Q, size_t n, double* __restrict__ Vt) {
    GRID_STRIDE(idx, n * n) {
        size_t j = idx / n, c = idx % n;
        Vt[idx] = Q[c + (n - 1 - j) * n];
    }
}

__global__ void eig_to_sing(const double* __restrict__ D, size_t n, double* __restrict__ sing, double scale) {
    GRID_STRIDE(j, n) sing[j] = scale * sqrt(fmax(D[n - 1 - j], 0.0));
}

// rows of A [rows, cols] (row-major) scaled by 1/sing[row] (0 at or below tol)
__global__ void scale_rows_inv(double* __restrict__ A, size_t rows, size_t cols, const double* __restrict__ sing,
                               double tol) {
    GRID_STRIDE(idx, rows * cols) {
        double s = sing[idx / cols];
        A[idx] *= (s > tol) ? 1.0 / s : 0.0;
    }
}

__global__ void scale_cols_inv(double* __restrict__ A, size_t rows, size_t cols, const double* __restrict__ sing,
                               double tol) {
    GRID_STRIDE(idx, rows * cols) {
        double s = sing[idx % cols];
        A[idx] *= (s > tol) ? 1.0 / s : 0.0;
    }
}

// ---- f32 inputs without f64 copies of W, U, V (round 5): panels of W are widened to f64 on the fly --------------------
// A = W^T in column-major [N, M] (lda = N), W row-major f32.  rows [n0, n0 + nc) of A, all M columns -> P [nc, M] (ld nc)
__global__ void widen_rows_panel(const float* __restrict__ W, size_t M, size_t N, size_t n0, size_t nc, double* __restrict__ P) {
    GRID_STRIDE(idx, nc * M) {
        const size_t r = idx % nc, m = idx / nc;
        P[idx] = (double)W[m * N + n0 + r];
    }
}
// columns [m0, m0 + mc) of A (= rows of W, contiguous) -> P [N, mc] (ld N)
__global__ void widen_cols_panel(const float* __restrict__ W, size_t N, size_t m0, size_t mc, double* __restrict__ P) {
    GRID_STRIDE(idx, mc * N) P[idx] = (double)W[m0 * N + idx];
}
// Uk32[i, j] = Q[i + (n-1-j) n]  /  Vt32[j, c] = Q[c + (n-1-j) n]: the unit-side factor straight from the eigenvectors
__global__ void eigvec_to_uk_f32(const double* __restrict__ Q, size_t n, float* __restrict__ Uk) {
    GRID_STRIDE(idx, n * n) {
        const size_t i = idx / n, j = idx % n;
        Uk[idx] = (float)Q[i + (n - 1 - j) * n];
    }
}
__global__ void eigvec_to_vt_f32(const double* __restrict__ Q, size_t n, float* __restrict__ Vt) {
    GRID_STRIDE(idx, n * n) {
        const size_t j = idx / n, c = idx % n;
        Vt[idx] = (float)Q[c + (n - 1 - j) * n];
    }
}
// C [nc, k] (ld nc; column i = A q_i for the ASCENDING eigenvector i, rows n0 ..) -> Vt32[j, n0 + r] = C[r, k-1-j] / sing[j]
__global__ void store_vt_panel(const double* __restrict__ C, size_t nc, size_t k, size_t N, size_t n0, const double* __restrict__ sing,
                               double tol, float* __restrict__ Vt) {
    GRID_STRIDE(idx, nc * k) {
        const size_t r = idx % nc, j = idx / nc;
        const double s = sing[j];
        Vt[j * N + n0 + r] = (float)(C[r + (k - 1 - j) * nc] * ((s > tol) ? 1.0 / s : 0.0));
    }
}
// C [k, mc] (ld k; row i = q_i^T A for the ascending eigenvector i, columns m0 ..) -> Uk32[m0 + r, j] = C[k-1-j, r] / sing[j]
__global__ void store_uk_panel(const double* __restrict__ C, size_t k, size_t mc, size_t m0, const double* __restrict__ sing,
                               double tol, float* __restrict__ Uk) {
    GRID_STRIDE(idx, k * mc) {
        const size_t j = idx % k, r = idx / k;
        const double s = sing[j];
        Uk[(m0 + r) * k + j] = (float)(C[(k - 1 - j) + r * k] * ((s > tol) ? 1.0 / s : 0.0));
    }
}

// cross[b + j c] /= sing[b]^2   (j0 x c, column-major)
__global__ void scale_cross(double* __restrict__ cross, size_t j0, size_t c, const double* __restrict__ sing) {
    GRID_STRIDE(idx, j0 * c) {
        const double s = sing[idx % j0];
        cross[idx] = (s > 0.0) ? cross[idx] / (s * s) : 0.0;
    }
}

// X <- s X + d I   (column-major n x n, ld = n)
__global__ void scale_add_identity(double* __restrict__ X, size_t n, double s, double d) {
    const size_t total = n * n;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        double v = s * X[idx];
        if (idx / n == idx % n) v += d;
        X[idx] = v;
    }
}

// X <- p X + q Y
__global__ void axpby_inplace(double* __restrict__ X, const double* __restrict__ Y, size_t total, double p, double q) {
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock)
        X[idx] = p * X[idx] + q * Y[idx];
}

__global__ void get_diag(const double* __restrict__ X, size_t n, double* __restrict__ d) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < n) d[j] = X[j + j * n];
}

// per-block partial sums of X[idx]^2 (fixed grid -> fixed summation order)
__global__ void sumsq_partials(const double* __restrict__ X, size_t total, double* __restrict__ part) {
    __shared__ double sm[kBlock / kWave];
    double s = 0.0;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock)
        s += X[idx] * X[idx];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// |Z - I|_F^2 from the upper triangle of the symmetric Z (column-major; lower triangle not read)
__global__ void resid_upper_partials(const double* __restrict__ Z, size_t n, double* __restrict__ part) {
    __shared__ double sm[kBlock / kWave];
    const size_t total = n * n;
    double s = 0.0;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        const size_t i = idx % n, j = idx / n;
        if (i > j) continue;
        const double z = Z[idx];
        s += (i == j) ? (z - 1.0) * (z - 1.0) : 2.0 * z * z;
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// mode 0: lower <- upper^T (fill the symmetric matrix from its upper triangle)
// mode 1: X <- (X + X^T)/2
// one 32x32 tile pair (ti <= tj) per block, staged through LDS so both sides are coalesced
__global__ void symmetrize_tiles(double* __restrict__ X, size_t n, int mode) {
    const size_t ti = blockIdx.x, tj = blockIdx.y;
    if (ti > tj) return;
    __shared__ double A[32][33], B[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 256 threads: 32 x 8
    // A = block (rows of ti, cols of tj); B = block (rows of tj, cols of ti)
    for (int r = ty; r < 32; r += 8) {
        const size_t ia = ti * 32 + tx, ja = tj * 32 + r;
        A[r][tx] = (ia < n && ja < n) ? X[ia + ja * n] : 0.0;        // A[col][row]
        const size_t ib = tj * 32 + tx, jb = ti * 32 + r;
        B[r][tx] = (ib < n && jb < n) ? X[ib + jb * n] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        // element (row tx, col r) of A and its mirror (row r, col tx) of B
        const size_t ia = ti * 32 + tx, ja = tj * 32 + r;
        const size_t ib = tj * 32 + tx, jb = ti * 32 + r;
        if (ti == tj) {
            if (ia < n && ja < n) {
                double v;
                if (mode == 0) v = (ia <= ja) ? A[r][tx] : A[tx][r];
                else v = 0.5 * (A[r][tx] + A[tx][r]);
                X[ia + ja * n] = v;
            }
        } else {
            if (mode == 0) {
                if (ib < n && jb < n) X[ib + jb * n] = A[tx][r];      // B(row tx, col r) = A(row r, col tx)
            } else {
                if (ia < n && ja < n) X[ia + ja * n] = 0.5 * (A[r][tx] + B[tx][r]);
                if (ib < n && jb < n) X[ib + jb * n] = 0.5 * (B[r][tx] + A[tx][r]);
            }
        }
    }
}

// columns [0, r) <- (X + I)/2, columns [r, n) <- (I - X)/2 : the two spectral projectors side by side
// y = X v for a symmetric X in FULL column-major storage: y_j = <column j, v>; one wave per column, 16-byte loads
__global__ __launch_bounds__(kBlock) void symv_full(const double* __restrict__ X, size_t n, const double* __restrict__ v,
                                                    double* __restrict__ y) {
    const size_t j = (size_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (j >= n) return;
    const int lane = threadIdx.x & 63;
    const double* col = X + j * n;
    double acc = 0.0;
    if ((n & 1) == 0) {
        const size_t nv = n / 2;
        for (size_t i = lane; i < nv; i += 64) {
            const double2 a = reinterpret_cast<const double2*>(col)[i];
            const double2 b = reinterpret_cast<const double2*>(v)[i];
            acc += a.x * b.x + a.y * b.y;
        }
    } else {
        for (size_t i = lane; i < n; i += 64) acc += col[i] * v[i];
    }
    acc = wave_sum(acc);
    if (lane == 0) y[j] = acc;
}
__global__ void vec_scale_copy(const double* __restrict__ src, size_t n, double s, double* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) dst[i] = s * src[i];
}
__global__ void vec_init(double* __restrict__ v, size_t n) {      // a fixed, unstructured start vector
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z ^= z >> 31;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 29;
        v[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}

__global__ void sign_to_projectors(double* __restrict__ X, size_t n, size_t r) {
    const size_t total = n * n;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        const size_t i = idx % n, j = idx / n;
        const double id = (i == j) ? 0.5 : 0.0;
        X[idx] = (j < r) ? id + 0.5 * X[idx] : id - 0.5 * X[idx];
    }
}

inline unsigned nblk(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
constexpr unsigned kRedBlocks = 4096;

// ------------------------------------------------------------------ blocked level-3 layer
struct Fz {
    tramp_ctx* ctx;
    rocblas_handle h;
    rocblas_int* info_d;
    double* part_d;              // kRedBlocks partial sums
    size_t eig_cap;              // largest n handed to rocSOLVER's syevd
    size_t sub_cap;              // ... inside a split (depth >= 1)
    int rc;
    std::vector<void*> owned;
    int fail(int code, const char* what, const char* detail) {
        rc = tramp_fail(ctx, code, what, detail);
        return rc;
    }
    std::vector<size_t> owned_bytes;
    size_t live = 0, peak = 0;           // bytes of work space this factorisation holds / held at most (tramp_factorize_peak_bytes)
    double* alloc(size_t count) {
        void* p = nullptr;
        const size_t nb = (count ? count : 1) * sizeof(double);
        hipError_t e = hipMalloc(&p, nb);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail(TRAMP_ENOMEM, "factorize: hipMalloc", hipGetErrorString(e));
            return nullptr;
        }
        owned.push_back(p);
        owned_bytes.push_back(nb);
        live += nb;
        if (live > peak) peak = live;
        return (double*)p;
    }
    void release(void* p) {
        if (!p) return;
        (void)hipStreamSynchronize(ctx->stream);
        for (size_t i = 0; i < owned.size(); ++i)
            if (owned[i] == p) {
                live -= owned_bytes[i];
                owned.erase(owned.begin() + i);
                owned_bytes.erase(owned_bytes.begin() + i);
                break;
            }
        (void)hipFree(p);
    }
    void release_all() {
        (void)hipStreamSynchronize(ctx->stream);
        for (void* p : owned) (void)hipFree(p);
        owned.clear();
        owned_bytes.clear();
        live = 0;
    }
    // TRAMP_EIG_VERBOSE=1: wall time of each phase of the setup path on stderr (synchronises the stream)
    double t_last = 0.0;
    void tick(const char* what, size_t n = 0) {
        static const bool on = getenv("TRAMP_EIG_VERBOSE") != nullptr;
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        const double now = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
        if (t_last > 0.0) fprintf(stderr, "[tramp factorize] %-28s n=%-6zu %8.3f s\n", what, n, now - t_last);
        t_last = now;
    }
};

#define FZ_RB(fz, call)                                                                             \
    do {                                                                                            \
        rocblas_status s_ = (call);                                                                 \
        if (s_ != rocblas_status_success)                                                           \
            return (fz).fail(s_ == rocblas_status_memory_error ? TRAMP_ENOMEM : TRAMP_EHIP, #call,  \
                             rocblas_status_to_string(s_));                                         \
    } while (0)
#define FZ_HIP(fz, call)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return (fz).fail(e_ == hipErrorOutOfMemory ? TRAMP_ENOMEM : TRAMP_EHIP, #call,          \
                             hipGetErrorString(e_));                                                \
    } while (0)
#define FZ_OK(expr)                    \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != TRAMP_OK) return rc_; \
    } while (0)

// tile edge such that (tile x leading dimension) stays below 2^31
inline size_t tile_for(size_t ld) {
    size_t t = ((size_t)1 << 31) / (ld ? ld : 1);
    if (t > 16384) t = 16384;
    if (t >= 512) t &= ~(size_t)255;
    return t ? t : 1;
}
inline size_t min3(size_t a, size_t b, size_t c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }

// C (m x n) <- alpha op(A) op(B) + beta C, column-major, any size
int bgemm(Fz& fz, rocblas_operation tA, rocblas_operation tB, size_t m, size_t n, size_t kk, double alpha,
          const double* A, size_t lda, const double* B, size_t ldb, double beta, double* C, size_t ldc) {
    if (!m || !n || !kk) return TRAMP_OK;
    const size_t T = min3(tile_for(lda), tile_for(ldb), tile_for(ldc));
    const double one = 1.0;
    for (size_t j0 = 0; j0 < n; j0 += T) {
        const size_t nj = n - j0 < T ? n - j0 : T;
        for (size_t i0 = 0; i0 < m; i0 += T) {
            const size_t mi = m - i0 < T ? m - i0 : T;
            for (size_t p0 = 0; p0 < kk; p0 += T) {
                const size_t kp = kk - p0 < T ? kk - p0 : T;
                const double* Ap = tA == rocblas_operation_none ? A + i0 + p0 * lda : A + p0 + i0 * lda;
                const double* Bp = tB == rocblas_operation_none ? B + p0 + j0 * ldb : B + j0 + p0 * ldb;
                FZ_RB(fz, rocblas_dgemm(fz.h, tA, tB, (rocblas_int)mi, (rocblas_int)nj, (rocblas_int)kp, &alpha, Ap,
                                        (rocblas_int)lda, Bp, (rocblas_int)ldb, p0 == 0 ? &beta : &one,
                                        C + i0 + j0 * ldc, (rocblas_int)ldc));
            }
        }
    }
    return TRAMP_OK;
}

// upper triangle of C (n x n) <- alpha op(A) op(A)^T + beta C;  op(A) is n x kk
// (trans none: A is n x kk;  trans transpose: A is kk x n and the product is A^T A)
int bsyrk_upper(Fz& fz, rocblas_operation trans, size_t n, size_t kk, double alpha, const double* A, size_t lda,
                double beta, double* C, size_t ldc) {
    if (!n) return TRAMP_OK;
    const size_t T = tile_for(lda) < tile_for(ldc) ? tile_for(lda) : tile_for(ldc);
    const double one = 1.0;
    const bool nt = trans == rocblas_operation_none;
    for (size_t j0 = 0; j0 < n; j0 += T) {
        const size_t nj = n - j0 < T ? n - j0 : T;
        for (size_t i0 = 0; i0 <= j0; i0 += T) {
            const size_t mi = n - i0 < T ? n - i0 : T;
            for (size_t p0 = 0; p0 < kk; p0 += T) {
                const size_t kp = kk - p0 < T ? kk - p0 : T;
                const double* Ai = nt ? A + i0 + p0 * lda : A + p0 + i0 * lda;
                const double* Aj = nt ? A + j0 + p0 * lda : A + p0 + j0 * lda;
                const double* bp = p0 == 0 ? &beta : &one;
                if (i0 == j0) {
                    FZ_RB(fz, rocblas_dsyrk(fz.h, rocblas_fill_upper, trans, (rocblas_int)nj, (rocblas_int)kp, &alpha,
                                            Aj, (rocblas_int)lda, bp, C + j0 + j0 * ldc, (rocblas_int)ldc));
                } else {
                    FZ_RB(fz, rocblas_dgemm(fz.h, nt ? rocblas_operation_none : rocblas_operation_transpose,
                                            nt ? rocblas_operation_transpose : rocblas_operation_none,
                                            (rocblas_int)mi, (rocblas_int)nj, (rocblas_int)kp, &alpha, Ai,
                                            (rocblas_int)lda, Aj, (rocblas_int)lda, bp, C + i0 + j0 * ldc,
                                            (rocblas_int)ldc));
                }
            }
        }
    }
    return TRAMP_OK;
}

int check_info(Fz& fz, const char* what) {
    rocblas_int info_h = 0;
    FZ_HIP(fz, hipMemcpyAsync(&info_h, fz.info_d, sizeof(info_h), hipMemcpyDeviceToHost, fz.ctx->stream));
    FZ_HIP(fz, hipStreamSynchronize(fz.ctx->stream));
    if (info_h != 0) return fz.fail(TRAMP_EHIP, what, "factorisation failed (matrix not positive definite / no convergence)");
    return TRAMP_OK;
}

constexpr size_t kCholBlock = 4096;

// Z = U^T U (upper), blocked right-looking; only the upper triangle is read and written
int bpotrf_upper(Fz& fz, double* Z, size_t n, size_t ld) {
    const double one = 1.0;
    const size_t T = tile_for(ld);
    for (size_t j0 = 0; j0 < n; j0 += kCholBlock) {
        const size_t jb = n - j0 < kCholBlock ? n - j0 : kCholBlock;
        double* Ujj = Z + j0 + j0 * ld;
        FZ_RB(fz, rocsolver_dpotrf(fz.h, rocblas_fill_upper, (rocblas_int)jb, Ujj, (rocblas_int)ld, fz.info_d));
        FZ_OK(check_info(fz, "rocsolver_dpotrf"));
        const size_t rest = n - j0 - jb;
        if (!rest) break;
        double* panel = Z + j0 + (j0 + jb) * ld;                   // jb x rest
        for (size_t c0 = 0; c0 < rest; c0 += T) {
            const size_t nc = rest - c0 < T ? rest - c0 : T;
            FZ_RB(fz, rocblas_dtrsm(fz.h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_transpose,
                                    rocblas_diagonal_non_unit, (rocblas_int)jb, (rocblas_int)nc, &one, Ujj,
                                    (rocblas_int)ld, panel + c0 * ld, (rocblas_int)ld));
        }
        FZ_OK(bsyrk_upper(fz, rocblas_operation_transpose, rest, jb, -1.0, panel, ld, 1.0,
                          Z + (j0 + jb) + (j0 + jb) * ld, ld));
    }
    return TRAMP_OK;
}

// X (n x nrhs) <- (U^T U)^-1 X
int bpotrs_upper(Fz& fz, const double* U, size_t n, size_t ldu, double* X, size_t nrhs, size_t ldx) {
    const double one = 1.0;
    const size_t T = tile_for(ldx);
    // U^T T = X
    for (size_t i0 = 0; i0 < n; i0 += kCholBlock) {
        const size_t ib = n - i0 < kCholBlock ? n - i0 : kCholBlock;
        for (size_t c0 = 0; c0 < nrhs; c0 += T) {
            const size_t nc = nrhs - c0 < T ? nrhs - c0 : T;
            FZ_RB(fz, rocblas_dtrsm(fz.h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_transpose,
                                    rocblas_diagonal_non_unit, (rocblas_int)ib, (rocblas_int)nc, &one,
                                    U + i0 + i0 * ldu, (rocblas_int)ldu, X + i0 + c0 * ldx, (rocblas_int)ldx));
        }
        const size_t rest = n - i0 - ib;
        if (rest)
            FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, rest, nrhs, ib, -1.0,
                        U + i0 + (i0 + ib) * ldu, ldu, X + i0, ldx, 1.0, X + i0 + ib, ldx));
    }
    // U Y = T
    const size_t nblocks = (n + kCholBlock - 1) / kCholBlock;
    for (size_t b = nblocks; b-- > 0;) {
        const size_t i0 = b * kCholBlock;
        const size_t ib = n - i0 < kCholBlock ? n - i0 : kCholBlock;
        for (size_t c0 = 0; c0 < nrhs; c0 += T) {
            const size_t nc = nrhs - c0 < T ? nrhs - c0 : T;
            FZ_RB(fz, rocblas_dtrsm(fz.h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_none,
                                    rocblas_diagonal_non_unit, (rocblas_int)ib, (rocblas_int)nc, &one,
                                    U + i0 + i0 * ldu, (rocblas_int)ldu, X + i0 + c0 * ldx, (rocblas_int)ldx));
        }
        if (i0)
            FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, i0, nrhs, ib, -1.0, U + i0 * ldu, ldu,
                        X + i0, ldx, 1.0, X, ldx));
    }
    return TRAMP_OK;
}

// X (m x n) <- X U^-1, U upper n x n
int btrsm_right_upper(Fz& fz, const double* U, size_t n, size_t ldu, double* X, size_t m, size_t ldx) {
    const double one = 1.0;
    for (size_t j0 = 0; j0 < n; j0 += kCholBlock) {
        const size_t jb = n - j0 < kCholBlock ? n - j0 : kCholBlock;
        if (j0)
            FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, m, jb, j0, -1.0, X, ldx, U + j0 * ldu, ldu,
                        1.0, X + j0 * ldx, ldx));
        FZ_RB(fz, rocblas_dtrsm(fz.h, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none,
                                rocblas_diagonal_non_unit, (rocblas_int)m, (rocblas_int)jb, &one, U + j0 + j0 * ldu,
                                (rocblas_int)ldu, X + j0 * ldx, (rocblas_int)ldx));
    }
    return TRAMP_OK;
}

int reduce_partials(Fz& fz, double* out) {
    static thread_local double host[kRedBlocks];
    FZ_HIP(fz, hipMemcpyAsync(host, fz.part_d, sizeof(host), hipMemcpyDeviceToHost, fz.ctx->stream));
    FZ_HIP(fz, hipStreamSynchronize(fz.ctx->stream));
    double s = 0.0;
    for (unsigned i = 0; i < kRedBlocks; ++i) s += host[i];
    *out = s;
    return TRAMP_OK;
}

// dynamically weighted Halley weights for the lower bound l (Nakatsukasa, Bai, Gygi 2010)
void halley_weights(double l, double* a, double* b, double* c) {
    const double l2 = l * l;
    const double dd = cbrt(4.0 * (1.0 / l2 - 1.0) / l2);
    const double sqd = sqrt(1.0 + dd);
    *a = sqd + sqrt(2.0 - dd + 2.0 * (2.0 - l2) / (l2 * sqd));
    *b = (*a - 1.0) * (*a - 1.0) / 4.0;
    *c = *a + *b - 1.0;
}

// |X|_2 of a symmetric X (full storage) by power iteration: an estimate from BELOW (Rayleigh quotients), `iters`
// passes over the matrix (n^2 doubles each - milliseconds next to the n^3 steps it scales)
int norm2_power(Fz& fz, const double* X, size_t n, int iters, double* out) {
    hipStream_t st = fz.ctx->stream;
    double* v = fz.alloc(n);
    double* y = fz.alloc(n);
    if (!v || !y) return fz.rc;
    hipLaunchKernelGGL(vec_init, dim3(nblk(n)), dim3(kBlock), 0, st, v, n);
    double est = 0.0;
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(symv_full, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, st, X, n, v, y);
        hipLaunchKernelGGL(sumsq_partials, dim3(kRedBlocks), dim3(kBlock), 0, st, y, n, fz.part_d);
        double yy = 0.0;
        FZ_OK(reduce_partials(fz, &yy));
        hipLaunchKernelGGL(sumsq_partials, dim3(kRedBlocks), dim3(kBlock), 0, st, v, n, fz.part_d);
        double vv = 0.0;
        FZ_OK(reduce_partials(fz, &vv));
        if (!(yy > 0.0) || !(vv > 0.0) || !isfinite(yy)) break;
        est = sqrt(yy / vv);
        hipLaunchKernelGGL(vec_scale_copy, dim3(nblk(n)), dim3(kBlock), 0, st, y, n, 1.0 / sqrt(yy), v);
    }
    fz.release(v);
    fz.release(y);
    *out = est;
    return TRAMP_OK;
}

// |X (X v) - v| / |v| on a probe vector: 0 for a matrix sign (X^2 = I); two passes over the matrix.  The certificate
// of a sign iteration that stopped on the PREDICTED contraction of its last step (no n^3 residual of the result).
int involution_defect(Fz& fz, const double* X, size_t n, double* out) {
    hipStream_t st = fz.ctx->stream;
    double* v = fz.alloc(n);
    double* y = fz.alloc(n);
    double* z = fz.alloc(n);
    if (!v || !y || !z) return fz.rc;
    hipLaunchKernelGGL(vec_init, dim3(nblk(n)), dim3(kBlock), 0, st, v, n);
    hipLaunchKernelGGL(symv_full, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, st, X, n, v, y);
    hipLaunchKernelGGL(symv_full, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, st, X, n, y, z);
    hipLaunchKernelGGL(sumsq_partials, dim3(kRedBlocks), dim3(kBlock), 0, st, v, n, fz.part_d);
    double vv = 0.0, dd = 0.0;
    FZ_OK(reduce_partials(fz, &vv));
    hipLaunchKernelGGL(axpby_inplace, dim3(kRedBlocks), dim3(kBlock), 0, st, z, v, n, -1.0, 1.0);      // z <- z - v
    hipLaunchKernelGGL(sumsq_partials, dim3(kRedBlocks), dim3(kBlock), 0, st, z, n, fz.part_d);
    FZ_OK(reduce_partials(fz, &dd));
    fz.release(v);
    fz.release(y);
    fz.release(z);
    *out = (vv > 0.0 && isfinite(dd)) ? sqrt(dd / vv) : INFINITY;
    return TRAMP_OK;
}

// orthonormal basis of the columns Y (m x r, ld) by two Cholesky-QR passes.  gram0: an r x r matrix
// holding Y^T Y in its upper triangle if the caller has it for free (consumed), else nullptr.
int cholqr2(Fz& fz, double* Y, size_t m, size_t r, size_t ld, double* gram /* r x r scratch, ld = r */, bool have_gram) {
    if (!r) return TRAMP_OK;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass || !have_gram) FZ_OK(bsyrk_upper(fz, rocblas_operation_transpose, r, m, 1.0, Y, ld, 0.0, gram, r));
        FZ_OK(bpotrf_upper(fz, gram, r, r));
        FZ_OK(btrsm_right_upper(fz, gram, r, r, Y, m, ld));
    }
    return TRAMP_OK;
}

int eigh_rec(Fz& fz, double* A, size_t n, double* w_d, int depth);

#include "tridiag_eig.h"

// TRAMP_EIG_METHOD: unset = "auto" (round 6): orders from 6144 on take the own tridiagonalisation + stedc + back-transformation
// of tridiag_eig.h, smaller ones rocSOLVER's syevd (0.15 s against 0.24 s at n = 4096; 0.67 against 0.61 s at 8192);
// "tridiag" = the tridiagonal path from order 64 on; "split" = rocSOLVER syevd / the spectral divide and conquer above its cap
// (rounds 2-5)
constexpr size_t kTridiagAutoMin = 6144;
int eig_method_from_env() {
    const char* s = getenv("TRAMP_EIG_METHOD");
    if (s && !strcmp(s, "split")) return 0;
    if (s && !strcmp(s, "tridiag")) return 1;
    return 2;
}

// work space of the tridiagonal path beyond A itself: the eigenvector matrix of T, and for the merge the two halves, V,
// the product and the ordered result (dense basis: one more n x n)
bool tridiag_fits(size_t n) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    const double nn = 8.0 * (double)n * (double)n;
    const double need = (n <= kStedcCap ? 1.3 : 3.6) * nn + (double)(64u << 20);
    return (double)free_b > need;
}

// one spectral split of the symmetric A (n x n, full storage, ld = n); eigenvectors overwrite A,
// ascending eigenvalues go to w_d[n]
int eigh_split(Fz& fz, double* A, size_t n, double* w_d, int depth) {
    hipStream_t st = fz.ctx->stream;
    const size_t nn = n * n;
    double* X = fz.alloc(nn);
    double* Y = fz.alloc(nn);
    double* Z = fz.alloc(nn);
    double* dg = fz.alloc(n);
    if (!X || !Y || !Z || !dg) return fz.rc;
    std::vector<double> hd(n);
    // sigma = mean eigenvalue
    hipLaunchKernelGGL(get_diag, dim3(nblk(n)), dim3(kBlock), 0, st, A, n, dg);
    FZ_HIP(fz, hipMemcpyAsync(hd.data(), dg, n * sizeof(double), hipMemcpyDeviceToHost, st));
    FZ_HIP(fz, hipStreamSynchronize(st));
    double sigma = 0.0;
    for (size_t i = 0; i < n; ++i) sigma += hd[i];
    sigma /= (double)n;
    FZ_HIP(fz, hipMemcpyAsync(X, A, nn * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(scale_add_identity, dim3(kRedBlocks), dim3(kBlock), 0, st, X, n, 1.0, -sigma);
    hipLaunchKernelGGL(sumsq_partials, dim3(kRedBlocks), dim3(kBlock), 0, st, X, nn, fz.part_d);
    double fro2 = 0.0;
    FZ_OK(reduce_partials(fz, &fro2));
    if (!isfinite(fro2)) return fz.fail(TRAMP_EINVAL, "factorize", "matrix is not finite");
    if (!(fro2 > 0.0)) {
        // A = sigma I exactly: any orthonormal basis is an eigenbasis
        hipLaunchKernelGGL(scale_add_identity, dim3(kRedBlocks), dim3(kBlock), 0, st, A, n, 0.0, 1.0);
        for (size_t i = 0; i < n; ++i) hd[i] = sigma;
        FZ_HIP(fz, hipMemcpyAsync(w_d, hd.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
        FZ_HIP(fz, hipStreamSynchronize(st));
        fz.release(X);
        fz.release(Y);
        fz.release(Z);
        fz.release(dg);
        return TRAMP_OK;
    }
    // X0 = (A - sigma I) / s with s just above |A - sigma I|_2: the power iteration estimates the norm from below,
    // 6 % of headroom keeps the spectrum of X0 inside [-1, 1] (a slightly larger eigenvalue would still converge).
    // Scaled by the Frobenius norm instead (the earlier form) the smallest |eigenvalue| of X0 - the l0 the
    // iteration count depends on - is sqrt(n)-ish smaller: one or two more n^3 steps at n = 50 000.
    double nrm2 = 0.0;
    FZ_OK(norm2_power(fz, X, n, 24, &nrm2));
    double scale0 = 1.06 * nrm2;
    if (!(scale0 > 0.0) || !(scale0 <= sqrt(fro2))) scale0 = sqrt(fro2);
    hipLaunchKernelGGL(scale_add_identity, dim3(kRedBlocks), dim3(kBlock), 0, st, X, n, 1.0 / scale0, 0.0);

    // weights: follow the predicted lower bound l_pred of the smallest |eigenvalue| of X (floored at
    // l_floor: c <= 1.6e4 keeps the Cholesky solve accurate); when the prediction says "converged"
    // but the residual e = |X^2 - I|_F does not - an eigenvalue closer to sigma than l0 assumed -
    // the bound the residual certifies, sqrt(1 - e), takes over (tools/proto_eig_dc.py)
    const double l_floor = 1e-3;
    double l_pred = 1e-8 * (sqrt(fro2) / scale0);          // the same prior as before, in the new units
    if (l_pred > 1e-5) l_pred = 1e-5;
    const double tol = 1e-15 * (double)n + 1e-13;        // residual that needs no further step at all (exact cases)
    bool converged = false;
    fz.tick("split: scale", n);
    for (int it = 0; it < 60; ++it) {
        // Z0 = X X^T (upper), e = |Z0 - I|_F
        FZ_OK(bsyrk_upper(fz, rocblas_operation_none, n, n, 1.0, X, n, 0.0, Z, n));
        hipLaunchKernelGGL(resid_upper_partials, dim3(kRedBlocks), dim3(kBlock), 0, st, Z, n, fz.part_d);
        double e2 = 0.0;
        FZ_OK(reduce_partials(fz, &e2));
        const double e = sqrt(e2);
        if (getenv("TRAMP_EIG_VERBOSE")) fprintf(stderr, "[tramp eigh] n=%zu depth=%d it=%d resid=%.3e l_pred=%.3e\n", n, depth, it, e, l_pred);
        if (!isfinite(e)) return fz.fail(TRAMP_EHIP, "factorize", "sign iteration produced non-finite values");
        if (e < tol) {
            converged = true;
            break;
        }
        // lower bound of the smallest |eigenvalue|: the predicted one, or - once the residual is below 1 - the
        // one it certifies (every eigenvalue x has |x^2 - 1| <= e), whichever is larger: the weights of the
        // last steps then fit the interval the spectrum really occupies and the final residual lands near the
        // rounding floor (1e-13 instead of 1e-10 at n = 50 000) without a further n^3 step
        double l = l_pred < 1.0 - 1e-8 ? l_pred : 0.0;
        if (e < 1.0) {
            const double lc = sqrt(1.0 - e);
            if (lc > l) l = lc;
        }
        if (l < l_floor) l = l_floor;
        if (l > 1.0 - 1e-12) l = 1.0 - 1e-12;
        double a, b, c;
        halley_weights(l, &a, &b, &c);
        l_pred = l_pred * (a + b * l_pred * l_pred) / (1.0 + c * l_pred * l_pred);
        if (l_pred > 1.0) l_pred = 1.0;
        hipLaunchKernelGGL(scale_add_identity, dim3(kRedBlocks), dim3(kBlock), 0, st, Z, n, c, 1.0);
        FZ_OK(bpotrf_upper(fz, Z, n, n));
        FZ_HIP(fz, hipMemcpyAsync(Y, X, nn * sizeof(double), hipMemcpyDeviceToDevice, st));
        FZ_OK(bpotrs_upper(fz, Z, n, n, Y, n, n));
        hipLaunchKernelGGL(axpby_inplace, dim3(kRedBlocks), dim3(kBlock), 0, st, X, Y, nn, b / c, a - b / c);
        const unsigned nt = (unsigned)((n + 31) / 32);
        hipLaunchKernelGGL(symmetrize_tiles, dim3(nt, nt), dim3(kBlock), 0, st, X, n, 1);
        fz.tick("split: sign iteration", n);
        // The step is cubic near the fixed point: taken from a residual below 1e-4 it leaves ~1e-13, the rounding
        // floor - no further n^3 product to confirm it.  (Stopping one step earlier, at the 1e-10 that the step from
        // 9e-2 leaves, costs three digits in the factors: 2.8e-10 against 3.6e-13 in |Vt Vt^T t - t| at k = 50 000.)
        if (e < 1e-4) {
            // ... but it IS checked, at O(n^2): X must square to the identity on a probe vector (a norm estimate that
            // was too low, or weights clamped at l_floor, would leave a defect far above rounding: iterate on)
            double defect = INFINITY;
            FZ_OK(involution_defect(fz, X, n, &defect));
            if (defect <= 1e-10 * sqrt((double)n)) {
                converged = true;
                break;
            }
        }
    }
    if (!converged) return fz.fail(TRAMP_EHIP, "factorize", "matrix sign iteration did not converge");

    // rank of the upper projector
    hipLaunchKernelGGL(get_diag, dim3(nblk(n)), dim3(kBlock), 0, st, X, n, dg);
    FZ_HIP(fz, hipMemcpyAsync(hd.data(), dg, n * sizeof(double), hipMemcpyDeviceToHost, st));
    FZ_HIP(fz, hipStreamSynchronize(st));
    double tr = 0.0;
    for (size_t i = 0; i < n; ++i) tr += hd[i];
    const long long rr = llround(0.5 * ((double)n + tr));
    if (rr <= 0 || rr >= (long long)n)
        return fz.fail(TRAMP_ENOTSUP, "factorize", "spectral split left one side empty");
    const size_t r = (size_t)rr, q = n - r;
    hipLaunchKernelGGL(sign_to_projectors, dim3(kRedBlocks), dim3(kBlock), 0, st, X, n, r);
    // X = [ P[:, :r] | (I-P)[:, r:] ]  ->  [ V1 | V2 ];  the first Gram matrices are blocks of X itself
    double* V1 = X;
    double* V2 = X + r * n;
    FZ_HIP(fz, hipMemcpy2DAsync(Z, r * sizeof(double), X, n * sizeof(double), r * sizeof(double), r,
                                hipMemcpyDeviceToDevice, st));
    FZ_OK(cholqr2(fz, V1, n, r, n, Z, true));
    FZ_HIP(fz, hipMemcpy2DAsync(Z, q * sizeof(double), X + r + r * n, n * sizeof(double), q * sizeof(double), q,
                                hipMemcpyDeviceToDevice, st));
    FZ_OK(cholqr2(fz, V2, n, q, n, Z, true));
    fz.tick("split: projector bases", n);
    // B1 = V1^T A V1 (r x r), B2 = V2^T A V2 (q x q), both in Z
    double* B1 = Z;
    double* B2 = Z + r * r;
    FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, n, r, n, 1.0, A, n, V1, n, 0.0, Y, n));
    FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, r, r, n, 1.0, V1, n, Y, n, 0.0, B1, r));
    FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, n, q, n, 1.0, A, n, V2, n, 0.0, Y, n));
    FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, q, q, n, 1.0, V2, n, Y, n, 0.0, B2, q));
    fz.release(Y);
    fz.tick("split: project A on bases", n);
    // below sigma first (ascending order)
    FZ_OK(eigh_rec(fz, B2, q, w_d, depth + 1));
    FZ_OK(eigh_rec(fz, B1, r, w_d + q, depth + 1));
    FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, n, q, q, 1.0, V2, n, B2, q, 0.0, A, n));
    FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, n, r, r, 1.0, V1, n, B1, r, 0.0, A + q * n, n));
    fz.tick("split: rotate back", n);
    fz.release(X);
    fz.release(Z);
    fz.release(dg);
    return TRAMP_OK;
}

// symmetric eigendecomposition: A (n x n column-major, upper triangle valid) -> eigenvectors in A,
// ascending eigenvalues in w_d
int eigh_rec(Fz& fz, double* A, size_t n, double* w_d, int depth) {
    const int method = eig_method_from_env();             // read per call, as TRAMP_EIG_CAP is
    if (method != 0 && n >= (method == 1 ? 64 : kTridiagAutoMin) && n <= 2 * kStedcCap && tridiag_fits(n)) {
        const unsigned nts = (unsigned)((n + 31) / 32);
        hipLaunchKernelGGL(symmetrize_tiles, dim3(nts, nts), dim3(kBlock), 0, fz.ctx->stream, A, n, 0);
        return eigh_tridiag(fz, A, n, w_d);
    }
    const size_t cap = (depth == 0 || fz.sub_cap > fz.eig_cap) ? fz.eig_cap : fz.sub_cap;
    if (n <= cap) {
        double* E = fz.alloc(n);
        if (!E) return fz.rc;
        FZ_RB(fz, rocsolver_dsyevd(fz.h, rocblas_evect_original, rocblas_fill_upper, (rocblas_int)n, A,
                                   (rocblas_int)n, w_d, E, fz.info_d));
        FZ_OK(check_info(fz, "rocsolver_dsyevd"));
        fz.release(E);
        fz.tick("rocsolver syevd", n);
        return TRAMP_OK;
    }
    if (depth > 8) return fz.fail(TRAMP_ENOTSUP, "factorize", "spectral divide and conquer recursed too deep");
    const unsigned nt = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(symmetrize_tiles, dim3(nt, nt), dim3(kBlock), 0, fz.ctx->stream, A, n, 0);
    return eigh_split(fz, A, n, w_d, depth);
}

size_t eig_cap_from_env() {
    // rocSOLVER's limit is 46336 (32-bit indexing); its syevd is bound by the memory-bound tridiagonalisation
    // (n = 28802: 23.8 s, n = 21198: 8.5 s on MI355X) while a spectral split runs at the rocBLAS level-3 rate,
    // so the halves of a forced split are split again down to this size (gpurun_out/r03/factorize1.log)
    size_t cap = 46336;
    if (const char* s = getenv("TRAMP_EIG_CAP")) {
        long v = atol(s);
        if (v >= 2 && v <= 46336) cap = (size_t)v;
    }
    return cap;
}

size_t eig_subcap_from_env() {
    size_t cap = 16384;
    if (const char* s = getenv("TRAMP_EIG_SUBCAP")) {
        long v = atol(s);
        if (v >= 2 && v <= 46336) cap = (size_t)v;
    }
    return cap;
}

int factorize_f64(Fz& fz, size_t M, size_t N, const double* W64, double* U64, double* V64, double* sing_d,
                  size_t* rank_out) {
    tramp_ctx* ctx = fz.ctx;
    hipStream_t st = ctx->stream;
    const size_t k = M < N ? M : N, big = M > N ? M : N;
    double* G = fz.alloc(k * k);
    double* D = fz.alloc(k);
    if (!G || !D) return fz.rc;
    fz.tick("start", k);
    // row-major W [M,N] is the column-major matrix A = W^T [N,M], lda = N
    if (M <= N)
        FZ_OK(bsyrk_upper(fz, rocblas_operation_transpose, M, N, 1.0, W64, N, 0.0, G, M));   // A^T A = W W^T
    else
        FZ_OK(bsyrk_upper(fz, rocblas_operation_none, N, M, 1.0, W64, N, 0.0, G, N));        // A A^T = W^T W
    fz.tick("gram matrix", k);
    FZ_OK(eigh_rec(fz, G, k, D, 0));
    hipLaunchKernelGGL(eig_to_sing, dim3(kRedBlocks), dim3(kBlock), 0, st, D, k, sing_d, 1.0);
    // the thin factor on the long side, rows/columns still carrying their singular value
    if (M <= N) {
        hipLaunchKernelGGL(eigvec_to_uk, dim3(kRedBlocks), dim3(kBlock), 0, st, G, k, U64);
        // Vt_cm [N,k] = A [N,M] * Y^T, Y_cm [k,M] = Uk memory (ldb = k)
        FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_transpose, N, k, M, 1.0, W64, N, U64, k, 0.0, V64, N));
    } else {
        hipLaunchKernelGGL(eigvec_to_vt, dim3(kRedBlocks), dim3(kBlock), 0, st, G, k, V64);
        // Uk_cm^T: C_cm [k,M] = X^T [k,N] * A [N,M], X_cm [N,k] = Vt memory (lda = N)
        FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, k, M, N, 1.0, V64, N, W64, N, 0.0, U64, k));
    }
    fz.tick("long-side factor (gemm)", k);
    std::vector<double> sh(k);
    FZ_HIP(fz, hipMemcpyAsync(sh.data(), sing_d, k * sizeof(double), hipMemcpyDeviceToHost, st));
    FZ_HIP(fz, hipStreamSynchronize(st));
    const double smax = sh[0];
    // further levels for the singular values a Gram matrix cannot resolve (below 1e-5 of its largest):
    // the trailing singular directions are re-factorised from the Gram matrix of their own rows, whose
    // entries are formed without cancellation; each level gains five more decades (three levels reach
    // 1e-15 sigma_max, LAPACK's absolute accuracy)
    size_t lo = 0;
    for (int level = 0; level < 3 && smax > 0.0; ++level) {
        size_t c = 0;
        while (c < k - lo && sh[k - 1 - c] <= 1e-5 * sh[lo]) ++c;
        if (c == 0 || c == k - lo) break;
        const size_t j0 = k - c;
        double* G2 = fz.alloc(c * c);
        double* D2 = fz.alloc(c);
        double* Qd = fz.alloc(c * c);
        double* Vo = fz.alloc(N * c);
        double* Uo = fz.alloc(c * M);
        if (!G2 || !D2 || !Qd || !Vo || !Uo) return fz.rc;
        // column-major views: Vt memory = [N, k] (column j = right direction j), Uk memory = [k, M]
        // (row j = left direction j).  The block's own Gram matrix comes from the LONG side.
        double* Vc = V64 + j0 * N;                                               // N x c, ld N
        double* Ur = U64 + j0;                                                   // c x M, ld k
        // The eigenvectors of a Gram matrix are accurate to eps * lambda_max / gap: the trailing block leans
        // into the directions above it by that angle, which for a (near-)null direction is a first-order
        // error eps * sigma_max^2 / sigma_neighbour.  One block Gram-Schmidt step of the block's carrying
        // rows against the j0 directions above (whose dot products are formed without cancellation)
        // removes it; the unit vectors on the other side follow with the same coefficients.
        if (j0 > 0) {
            double* cross = fz.alloc(j0 * c);
            if (!cross) return fz.rc;
            if (M <= N) {
                FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, j0, c, N, 1.0, V64, N, Vc, N, 0.0,
                            cross, j0));
                hipLaunchKernelGGL(scale_cross, dim3(kRedBlocks), dim3(kBlock), 0, st, cross, j0, c, sing_d);
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, N, c, j0, -1.0, V64, N, cross, j0, 1.0,
                            Vc, N));
                FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, c, M, j0, -1.0, cross, j0, U64, k,
                            1.0, Ur, k));
            } else {
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_transpose, j0, c, M, 1.0, U64, k, Ur, k, 0.0,
                            cross, j0));
                hipLaunchKernelGGL(scale_cross, dim3(kRedBlocks), dim3(kBlock), 0, st, cross, j0, c, sing_d);
                FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, c, M, j0, -1.0, cross, j0, U64, k,
                            1.0, Ur, k));
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, N, c, j0, -1.0, V64, N, cross, j0, 1.0,
                            Vc, N));
            }
            // ... and the directions above re-orthogonalised against the corrected block on the UNIT side (the
            // first step moved the block's unit vectors by up to the tilt; without this the two groups would
            // be left non-orthogonal by that angle): D = S_b^T S_c, S_b -= S_c D^T, carrying rows alike.
            if (M <= N) {
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_transpose, j0, c, M, 1.0, U64, k, Ur, k, 0.0,
                            cross, j0));
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, j0, M, c, -1.0, cross, j0, Ur, k, 1.0,
                            U64, k));
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_transpose, N, j0, c, -1.0, Vc, N, cross, j0,
                            1.0, V64, N));
            } else {
                FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, j0, c, N, 1.0, V64, N, Vc, N, 0.0,
                            cross, j0));
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_transpose, N, j0, c, -1.0, Vc, N, cross, j0,
                            1.0, V64, N));
                FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, j0, M, c, -1.0, cross, j0, Ur, k, 1.0,
                            U64, k));
            }
            fz.release(cross);
        }
        // the block's Gram matrix in units of its own largest singular value: the library eigensolver is
        // handed an O(1) matrix whatever the absolute scale (1e-20 ... 1e-44 here)
        const double s0 = sh[j0] > 0.0 ? sh[j0] : 1.0;
        const double gscale = (1.0 / s0) * (1.0 / s0);
        if (M <= N)
            FZ_OK(bsyrk_upper(fz, rocblas_operation_transpose, c, N, gscale, Vc, N, 0.0, G2, c));
        else
            FZ_OK(bsyrk_upper(fz, rocblas_operation_none, c, M, gscale, Ur, k, 0.0, G2, c));
        FZ_OK(eigh_rec(fz, G2, c, D2, 0));
        hipLaunchKernelGGL(eig_to_sing, dim3(kRedBlocks), dim3(kBlock), 0, st, D2, c, sing_d + j0, s0);
        // Qd (row-major) = eigenvectors in descending order = Qrev; as column-major it is Qrev^T
        hipLaunchKernelGGL(eigvec_to_uk, dim3(kRedBlocks), dim3(kBlock), 0, st, G2, c, Qd);
        // right directions: Vc <- Vc Qrev;  left directions: Ur <- Qrev^T Ur
        FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_transpose, N, c, c, 1.0, Vc, N, Qd, c, 0.0, Vo, N));
        FZ_HIP(fz, hipMemcpyAsync(Vc, Vo, N * c * sizeof(double), hipMemcpyDeviceToDevice, st));
        FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, c, M, c, 1.0, Qd, c, Ur, k, 0.0, Uo, c));
        FZ_HIP(fz, hipMemcpy2DAsync(Ur, k * sizeof(double), Uo, c * sizeof(double), c * sizeof(double), M,
                                    hipMemcpyDeviceToDevice, st));
        fz.release(G2);
        fz.release(D2);
        fz.release(Qd);
        fz.release(Vo);
        fz.release(Uo);
        FZ_HIP(fz, hipMemcpyAsync(sh.data(), sing_d, k * sizeof(double), hipMemcpyDeviceToHost, st));
        FZ_HIP(fz, hipStreamSynchronize(st));
        lo = j0;
        if (!(sh[lo] > 0.0)) break;
    }
    // rank with NumPy's tolerance (np.linalg.matrix_rank), floored at what two Gram levels resolve
    double tol = smax * (double)big * kEps;
    const double floor_tol = smax * 64.0 * kEps * sqrt((double)k);
    if (tol < floor_tol) tol = floor_tol;
    size_t r = 0;
    for (size_t j = 0; j < k; ++j) r += sh[j] > tol;
    *rank_out = r;
    if (M <= N)
        hipLaunchKernelGGL(scale_rows_inv, dim3(kRedBlocks), dim3(kBlock), 0, st, V64, k, N, sing_d, tol);
    else
        hipLaunchKernelGGL(scale_cols_inv, dim3(kRedBlocks), dim3(kBlock), 0, st, U64, M, k, sing_d, tol);
    FZ_HIP(fz, hipGetLastError());
    fz.release(G);
    fz.release(D);
    return TRAMP_OK;
}

// The f32 entry without f64 copies of W, Uk, Vt (which capped BASELINE configs[2] / configs[4] at N = 49 152 / 65 536: W,
// U, V in f64 beside the f32 originals and four k x k work matrices is 2.4 x 288 GB at N = 131 072).  The Gram matrix is
// accumulated in f64 from f64-widened PANELS of W (a few GB at a time), the unit-side factor is written in f32 straight
// from the eigenvectors, the long-side factor panel by panel (widened W panel x eigenvectors, scaled by 1 / sigma,
// stored as f32).  Arithmetic identical to the f64 path on the widened matrix up to the order of the panel sums.
// *done = false (nothing written) when the spectrum has singular values below 1e-5 sigma_max: those need the
// re-factorisation levels of factorize_f64 on full f64 factors, which the caller then runs.
int factorize_f32_panels(Fz& fz, size_t M, size_t N, const float* W32, float* Uk32, float* Vt32, double* sing_d, size_t* rank_out,
                         bool* done) {
    tramp_ctx* ctx = fz.ctx;
    hipStream_t st = ctx->stream;
    *done = false;
    const size_t k = M < N ? M : N, big = M > N ? M : N;
    double* G = fz.alloc(k * k);
    double* D = fz.alloc(k);
    if (!G || !D) return fz.rc;
    // panel: pc indices of the LONG side at a time, pc x k doubles <= 2 GiB (and at least 256 indices)
    size_t pc = ((size_t)2 << 30) / (8 * k);
    if (pc < 256) pc = 256;
    if (pc > big) pc = big;
    double* P = fz.alloc(pc * k);
    if (!P) return fz.rc;
    fz.tick("start", k);
    for (size_t p0 = 0; p0 < big; p0 += pc) {
        const size_t c = (big - p0 < pc) ? big - p0 : pc;
        if (M <= N) {
            hipLaunchKernelGGL(widen_rows_panel, dim3(kRedBlocks), dim3(kBlock), 0, st, W32, M, N, p0, c, P);
            FZ_OK(bsyrk_upper(fz, rocblas_operation_transpose, M, c, 1.0, P, c, p0 ? 1.0 : 0.0, G, M));      // += P^T P
        } else {
            hipLaunchKernelGGL(widen_cols_panel, dim3(kRedBlocks), dim3(kBlock), 0, st, W32, N, p0, c, P);
            FZ_OK(bsyrk_upper(fz, rocblas_operation_none, N, c, 1.0, P, N, p0 ? 1.0 : 0.0, G, N));            // += P P^T
        }
    }
    fz.release(P);                                   // not held across the eigensolve (its work matrices need the room)
    fz.tick("gram matrix (f32 panels)", k);
    FZ_OK(eigh_rec(fz, G, k, D, 0));
    hipLaunchKernelGGL(eig_to_sing, dim3(kRedBlocks), dim3(kBlock), 0, st, D, k, sing_d, 1.0);
    std::vector<double> sh(k);
    FZ_HIP(fz, hipMemcpyAsync(sh.data(), sing_d, k * sizeof(double), hipMemcpyDeviceToHost, st));
    FZ_HIP(fz, hipStreamSynchronize(st));
    const double smax = sh[0];
    if (!(smax > 0.0) || sh[k - 1] <= 1e-5 * smax) {         // the Gram matrix does not resolve the trailing directions
        fz.release(G);
        fz.release(D);
        return TRAMP_OK;
    }
    double tol = smax * (double)big * kEps;
    const double floor_tol = smax * 64.0 * kEps * sqrt((double)k);
    if (tol < floor_tol) tol = floor_tol;
    size_t r = 0;
    for (size_t j = 0; j < k; ++j) r += sh[j] > tol;
    *rank_out = r;
    if (M <= N) hipLaunchKernelGGL(eigvec_to_uk_f32, dim3(kRedBlocks), dim3(kBlock), 0, st, G, k, Uk32);
    else hipLaunchKernelGGL(eigvec_to_vt_f32, dim3(kRedBlocks), dim3(kBlock), 0, st, G, k, Vt32);
    // the long side: two panels (widened W, product) of pc2 x k doubles, 1 GiB each
    size_t pc2 = ((size_t)1 << 30) / (8 * k);
    if (pc2 < 256) pc2 = 256;
    if (pc2 > big) pc2 = big;
    P = fz.alloc(pc2 * k);
    double* C = fz.alloc(pc2 * k);
    if (!P || !C) return fz.rc;
    for (size_t p0 = 0; p0 < big; p0 += pc2) {
        const size_t c = (big - p0 < pc2) ? big - p0 : pc2;
        if (M <= N) {
            hipLaunchKernelGGL(widen_rows_panel, dim3(kRedBlocks), dim3(kBlock), 0, st, W32, M, N, p0, c, P);
            FZ_OK(bgemm(fz, rocblas_operation_none, rocblas_operation_none, c, k, M, 1.0, P, c, G, k, 0.0, C, c));     // [c, M] Q
            hipLaunchKernelGGL(store_vt_panel, dim3(kRedBlocks), dim3(kBlock), 0, st, C, c, k, N, p0, sing_d, tol, Vt32);
        } else {
            hipLaunchKernelGGL(widen_cols_panel, dim3(kRedBlocks), dim3(kBlock), 0, st, W32, N, p0, c, P);
            FZ_OK(bgemm(fz, rocblas_operation_transpose, rocblas_operation_none, k, c, N, 1.0, G, k, P, N, 0.0, C, k));  // Q^T [N, c]
            hipLaunchKernelGGL(store_uk_panel, dim3(kRedBlocks), dim3(kBlock), 0, st, C, k, c, p0, sing_d, tol, Uk32);
        }
    }
    FZ_HIP(fz, hipGetLastError());
    fz.tick("long-side factor (f32 panels)", k);
    fz.release(P);
    fz.release(C);
    fz.release(G);
    fz.release(D);
    *done = true;
    return TRAMP_OK;
}

}  // namespace

extern "C" int tramp_linear_factorize(tramp_ctx* ctx, int dtype, size_t M, size_t N, const void* W_d, void* Uk_d,
                                      void* Vt_d, double* sing_d, size_t* rank_out) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, W_d && Uk_d && Vt_d && sing_d && rank_out && M > 0 && N > 0);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, M < (1u << 30) && N < (1u << 30));
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t k = M < N ? M : N;
    Fz fz{};
    fz.ctx = ctx;
    fz.rc = TRAMP_OK;
    fz.eig_cap = eig_cap_from_env();
    fz.sub_cap = eig_subcap_from_env();
    rocblas_status rs = rocblas_create_handle(&fz.h);
    if (rs != rocblas_status_success) return tramp_fail(ctx, TRAMP_EHIP, "rocblas_create_handle", rocblas_status_to_string(rs));
    int rc = TRAMP_OK;
    double *W64 = nullptr, *U64 = nullptr, *V64 = nullptr;
    auto body = [&]() -> int {
        FZ_RB(fz, rocblas_set_stream(fz.h, ctx->stream));
        FZ_RB(fz, rocblas_set_pointer_mode(fz.h, rocblas_pointer_mode_host));
        fz.info_d = (rocblas_int*)fz.alloc(1);
        fz.part_d = fz.alloc(kRedBlocks);
        if (!fz.info_d || !fz.part_d) return fz.rc;
        if (dtype == TRAMP_F32) {
            bool done = false;
            FZ_OK(factorize_f32_panels(fz, M, N, (const float*)W_d, (float*)Uk_d, (float*)Vt_d, sing_d, rank_out, &done));
            if (done) {
                FZ_HIP(fz, hipStreamSynchronize(ctx->stream));
                return TRAMP_OK;
            }
            // singular values below 1e-5 sigma_max: the re-factorisation levels work on full f64 factors
            W64 = fz.alloc(M * N);
            U64 = fz.alloc(M * k);
            V64 = fz.alloc(k * N);
            if (!W64 || !U64 || !V64) return fz.rc;
            FZ_OK(tramp_convert(ctx, TRAMP_F64, W64, TRAMP_F32, W_d, M * N));
        } else {
            W64 = (double*)W_d;
            U64 = (double*)Uk_d;
            V64 = (double*)Vt_d;
        }
        FZ_OK(factorize_f64(fz, M, N, W64, U64, V64, sing_d, rank_out));
        if (dtype == TRAMP_F32) {
            FZ_OK(tramp_convert(ctx, TRAMP_F32, Uk_d, TRAMP_F64, U64, M * k));
            FZ_OK(tramp_convert(ctx, TRAMP_F32, Vt_d, TRAMP_F64, V64, k * N));
        }
        FZ_HIP(fz, hipStreamSynchronize(ctx->stream));
        return TRAMP_OK;
    };
    rc = body();
    ctx->fz_peak_bytes = fz.peak;
    if (getenv("TRAMP_EIG_VERBOSE")) fprintf(stderr, "[tramp factorize] peak work space %.2f GB\n", (double)fz.peak / 1e9);
    fz.release_all();
    (void)rocblas_destroy_handle(fz.h);
    return rc;
}

/* Work space the last tramp_linear_factorize of this context held at most (bytes; the caller's W, Uk, Vt not included). */
extern "C" int tramp_factorize_peak_bytes(tramp_ctx* ctx, size_t* bytes) {
    if (!ctx || !bytes) return TRAMP_EINVAL;
    *bytes = ctx->fz_peak_bytes;
    return TRAMP_OK;
}

/* Symmetric eigendecomposition on the device (the eigensolver of tramp_linear_factorize on its own):
 * A_d [n, n] symmetric (either storage order), overwritten by the eigenvectors as COLUMNS of the
 * column-major matrix (A_d[i + j n] = component i of eigenvector j); w_d[n] ascending eigenvalues. */
extern "C" int tramp_eigh(tramp_ctx* ctx, size_t n, double* A_d, double* w_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, A_d && w_d && n > 0 && n < (1u << 30));
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Fz fz{};
    fz.ctx = ctx;
    fz.rc = TRAMP_OK;
    fz.eig_cap = eig_cap_from_env();
    fz.sub_cap = eig_subcap_from_env();
    rocblas_status rs = rocblas_create_handle(&fz.h);
    if (rs != rocblas_status_success) return tramp_fail(ctx, TRAMP_EHIP, "rocblas_create_handle", rocblas_status_to_string(rs));
    auto body = [&]() -> int {
        FZ_RB(fz, rocblas_set_stream(fz.h, ctx->stream));
        FZ_RB(fz, rocblas_set_pointer_mode(fz.h, rocblas_pointer_mode_host));
        fz.info_d = (rocblas_int*)fz.alloc(1);
        fz.part_d = fz.alloc(kRedBlocks);
        if (!fz.info_d || !fz.part_d) return fz.rc;
        FZ_OK(eigh_rec(fz, A_d, n, w_d, 0));
        FZ_HIP(fz, hipStreamSynchronize(ctx->stream));
        return TRAMP_OK;
    };
    int rc = body();
    fz.release_all();
    (void)rocblas_destroy_handle(fz.h);
    return rc;
}
