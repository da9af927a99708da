// factorize.hip - device-side thin SVD of W for LinearChannel setup.
//
// Replaces LinearChannel.__init__'s two host SVDs (np.linalg.matrix_rank + full
// np.linalg.svd, tramp/channels/linear/linear_channel.py:8-15,37-41) by a Gram-matrix
// eigendecomposition on the smaller side - library calls are fine here, this is
// one-time setup, not the per-iteration path:
//     M <= N:  G = W W^T (rocBLAS syrk)  -> eigh (rocSOLVER syevd) -> U, sigma^2
//              Vt = Sigma^-1 U^T W (rocBLAS gemm)
//     M >  N:  G = W^T W -> eigh -> V, sigma^2;  Uk = W V Sigma^-1
// Only Uk, Vt and sigma leave; W itself is not needed by the message kernels.
// The Gram form squares the condition number: singular values below
// sqrt(eps)*sigma_max are treated as zero (rank).  Ill-conditioned matrices should use
// the host LAPACK path (tramp_amd.channels.thin_svd_host).
// All arithmetic in f64; f32 inputs/outputs are converted at the boundary.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "common.h"

#define RB_CHECK(ctx, call)                                                              \
    do {                                                                                 \
        rocblas_status s_ = (call);                                                      \
        if (s_ != rocblas_status_success)                                                \
            return tramp_fail(ctx, TRAMP_EHIP, #call, rocblas_status_to_string(s_));     \
    } while (0)

// Uk[i, j] = Q[i + (n-1-j) n]   (eigenvector columns, descending eigenvalue order), M <= N case
__global__ void eigvec_to_uk(const double* __restrict__ Q, size_t n, double* __restrict__ Uk) {
    size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= n * n) return;
    size_t i = idx / n, j = idx % n;
    Uk[idx] = Q[i + (n - 1 - j) * n];
}

// Vt[j, c] = Q[c + (n-1-j) n]   (row reversal), M > N case
__global__ void eigvec_to_vt(const double* __restrict__ Q, size_t n, double* __restrict__ Vt) {
    size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= n * n) return;
    size_t j = idx / n, c = idx % n;
    Vt[idx] = Q[c + (n - 1 - j) * n];
}

__global__ void eig_to_sing(const double* __restrict__ D, size_t n, double* __restrict__ sing) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < n) sing[j] = sqrt(fmax(D[n - 1 - j], 0.0));
}

// rows of A [rows, cols] scaled by 1/sing[row] (0 below tol)
__global__ void scale_rows_inv(double* __restrict__ A, size_t rows, size_t cols, const double* __restrict__ sing,
                               double tol) {
    size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= rows * cols) return;
    double s = sing[idx / cols];
    A[idx] *= (s > tol) ? 1.0 / s : 0.0;
}

__global__ void scale_cols_inv(double* __restrict__ A, size_t rows, size_t cols, const double* __restrict__ sing,
                               double tol) {
    size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= rows * cols) return;
    double s = sing[idx % cols];
    A[idx] *= (s > tol) ? 1.0 / s : 0.0;
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

extern "C" int tramp_linear_factorize(tramp_ctx* ctx, int dtype, size_t M, size_t N, const void* W_d, void* Uk_d,
                                      void* Vt_d, double* sing_d, size_t* rank_out) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, W_d && Uk_d && Vt_d && sing_d && rank_out && M > 0 && N > 0);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, M < (1u << 30) && N < (1u << 30));
    const size_t k = M < N ? M : N;
    // rocSOLVER 7.2 has no 64-bit syevd: its kernels index the k x k matrix with 32-bit ints
    // (observed: GPU memory fault at k = 50000).  Refuse instead of faulting.
    if (k * k >= (1ull << 31))
        return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_linear_factorize",
                          "min(M,N) > 46340 exceeds rocSOLVER's 32-bit indexing; factorise on the host "
                          "or pass precomputed factors");
    TRAMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rocblas_handle h = nullptr;
    RB_CHECK(ctx, rocblas_create_handle(&h));
    int rc = TRAMP_OK;
    double *W64 = nullptr, *G = nullptr, *D = nullptr, *E = nullptr, *U64 = nullptr, *V64 = nullptr;
    rocblas_int* info = nullptr;
    const double one = 1.0, zero = 0.0;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        if (dtype == TRAMP_F32) {
            if (W64) (void)hipFree(W64);
            if (U64) (void)hipFree(U64);
            if (V64) (void)hipFree(V64);
        }
        if (G) (void)hipFree(G);
        if (D) (void)hipFree(D);
        if (E) (void)hipFree(E);
        if (info) (void)hipFree(info);
        (void)rocblas_destroy_handle(h);
    };
#define F_CHECK(expr)                      \
    do {                                   \
        rc = (expr);                       \
        if (rc != TRAMP_OK) {              \
            cleanup();                     \
            return rc;                     \
        }                                  \
    } while (0)
#define F_HIP(call)                                                                   \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            cleanup();                                                                \
            return tramp_fail(ctx, e_ == hipErrorOutOfMemory ? TRAMP_ENOMEM : TRAMP_EHIP, #call, hipGetErrorString(e_)); \
        }                                                                             \
    } while (0)
#define F_RB(call)                                                                    \
    do {                                                                              \
        rocblas_status s_ = (call);                                                   \
        if (s_ != rocblas_status_success) {                                           \
            cleanup();                                                                \
            return tramp_fail(ctx, s_ == rocblas_status_memory_error ? TRAMP_ENOMEM : TRAMP_EHIP, #call, rocblas_status_to_string(s_)); \
        }                                                                             \
    } while (0)

    F_RB(rocblas_set_stream(h, ctx->stream));
    F_RB(rocblas_set_pointer_mode(h, rocblas_pointer_mode_host));
    if (dtype == TRAMP_F32) {
        F_HIP(hipMalloc((void**)&W64, M * N * sizeof(double)));
        F_HIP(hipMalloc((void**)&U64, M * k * sizeof(double)));
        F_HIP(hipMalloc((void**)&V64, k * N * sizeof(double)));
        F_CHECK(tramp_convert(ctx, TRAMP_F64, W64, TRAMP_F32, W_d, M * N));
    } else {
        W64 = (double*)W_d;
        U64 = (double*)Uk_d;
        V64 = (double*)Vt_d;
    }
    F_HIP(hipMalloc((void**)&G, k * k * sizeof(double)));
    F_HIP(hipMalloc((void**)&D, k * sizeof(double)));
    F_HIP(hipMalloc((void**)&E, k * sizeof(double)));
    F_HIP(hipMalloc((void**)&info, sizeof(rocblas_int)));
    // row-major W [M,N] is the column-major matrix A = W^T [N,M], lda = N
    if (M <= N) {
        // G = A^T A = W W^T  (upper triangle)
        F_RB(rocblas_dsyrk(h, rocblas_fill_upper, rocblas_operation_transpose, (rocblas_int)M, (rocblas_int)N, &one,
                           W64, (rocblas_int)N, &zero, G, (rocblas_int)M));
    } else {
        // G = A A^T = W^T W
        F_RB(rocblas_dsyrk(h, rocblas_fill_upper, rocblas_operation_none, (rocblas_int)N, (rocblas_int)M, &one, W64,
                           (rocblas_int)N, &zero, G, (rocblas_int)N));
    }
    F_RB(rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, (rocblas_int)k, G, (rocblas_int)k, D, E,
                          info));
    rocblas_int info_h = 0;
    F_HIP(hipMemcpyAsync(&info_h, info, sizeof(info_h), hipMemcpyDeviceToHost, ctx->stream));
    F_HIP(hipStreamSynchronize(ctx->stream));
    if (info_h != 0) {
        cleanup();
        return tramp_fail(ctx, TRAMP_EHIP, "rocsolver_dsyevd", "eigensolver did not converge");
    }
    hipLaunchKernelGGL(eig_to_sing, dim3(nblk(k)), dim3(kBlock), 0, ctx->stream, D, k, sing_d);
    // rank tolerance of the Gram route: sqrt(eps * max(M,N)) * sigma_max
    double smax = 0.0;
    F_HIP(hipMemcpyAsync(&smax, sing_d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    F_HIP(hipStreamSynchronize(ctx->stream));
    const double tol = smax * sqrt(2.220446049250313e-16 * (double)(M > N ? M : N));
    if (M <= N) {
        hipLaunchKernelGGL(eigvec_to_uk, dim3(nblk(k * k)), dim3(kBlock), 0, ctx->stream, G, k, U64);
        // Vt_cm [N,k] = A [N,M] * Y^T, Y_cm [k,M] = Uk memory (ldb = k)
        F_RB(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, (rocblas_int)N, (rocblas_int)k,
                           (rocblas_int)M, &one, W64, (rocblas_int)N, U64, (rocblas_int)k, &zero, V64,
                           (rocblas_int)N));
        hipLaunchKernelGGL(scale_rows_inv, dim3(nblk(k * N)), dim3(kBlock), 0, ctx->stream, V64, k, N, sing_d, tol);
    } else {
        hipLaunchKernelGGL(eigvec_to_vt, dim3(nblk(k * k)), dim3(kBlock), 0, ctx->stream, G, k, V64);
        // Uk_cm^T: C_cm [k,M] = X^T [k,N] * A [N,M], X_cm [N,k] = Vt memory (lda = N)
        F_RB(rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)k, (rocblas_int)M,
                           (rocblas_int)N, &one, V64, (rocblas_int)N, W64, (rocblas_int)N, &zero, U64,
                           (rocblas_int)k));
        hipLaunchKernelGGL(scale_cols_inv, dim3(nblk(M * k)), dim3(kBlock), 0, ctx->stream, U64, M, k, sing_d, tol);
    }
    // rank = #{sing > tol}
    {
        double* sh = (double*)malloc(k * sizeof(double));
        if (!sh) {
            cleanup();
            return TRAMP_ENOMEM;
        }
        hipError_t e = hipMemcpyAsync(sh, sing_d, k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        size_t r = 0;
        for (size_t j = 0; j < k; ++j) r += sh[j] > tol;
        free(sh);
        if (e != hipSuccess) {
            cleanup();
            return tramp_fail(ctx, TRAMP_EHIP, "rank readback", hipGetErrorString(e));
        }
        *rank_out = r;
    }
    if (dtype == TRAMP_F32) {
        F_CHECK(tramp_convert(ctx, TRAMP_F32, Uk_d, TRAMP_F64, U64, M * k));
        F_CHECK(tramp_convert(ctx, TRAMP_F32, Vt_d, TRAMP_F64, V64, k * N));
    }
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            cleanup();
            return tramp_fail(ctx, TRAMP_EHIP, "factorize kernels", hipGetErrorString(e));
        }
    }
    cleanup();
    return TRAMP_OK;
}
