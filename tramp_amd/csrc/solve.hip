// solve.hip - LinearChannel(precompute_svd=False): the branch of the reference that keeps C = W^T W and solves
//     (az I + ax C) rz = bz + W^T bx                         (tramp/channels/linear/linear_channel.py:43-44, 79-82)
// anew on every call instead of rotating into a precomputed singular basis.  It is the slow branch by construction
// (one N x N factorisation per instance and call) and exists so that a model built with precompute_svd=False runs
// unchanged; the per-iteration path of a large run is the thin-SVD handle of linear.hip.
//
// Device formulation, f64 throughout (inputs and outputs in the channel's dtype):
//     setup   C = W^T W (rocBLAS syrk), spectrum = eigvalsh(C) ascending (tramp_eigh);
//     call    g = bz + bx W                     (tramp_matvec_nn on W)
//             A_b = az_b I + ax_b C, Cholesky A_b = U^T U, rz_b = A_b^-1 g_b   (rocSOLVER potrf / potrs, strided
//             batches of as many instances as fit a 2 GiB scratch) - np.linalg.solve uses LU, the matrix is
//             symmetric positive definite whenever az > 0 or C is non-singular, and a failed factorisation is
//             reported like numpy's LinAlgError("Singular matrix");
//             rx = rz W^T                       (tramp_matvec_nt) for the forward direction;
//             variances from the n_eff formula over `singular` (kernel K9, linear.hip) and the message
//             a_new = clip(1 / v - a), b_new = r (a + a_new) - b (tramp/base.py:245-255), damping fused.
// One quirk of the reference is reproduced on purpose: with precompute_svd=False `singular = spectrum[:rank]` is cut
// from the ASCENDING eigenvalues of eigvalsh (linear_channel.py:44-46), i.e. it keeps the rank SMALLEST eigenvalues
// (zeros first when Nz > rank), and n_eff - hence every variance - follows that array.  tramp_lsolve_set_rank takes
// the rank the caller computed the way the reference does (np.linalg.matrix_rank).
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <stdlib.h>

#include <vector>

#include "common.h"
#include "linear.h"
#include "tramp_math.h"

struct tramp_lsolve {
    int dtype;
    size_t M, N, rank;
    const void* W;        // borrowed [M, N] row-major, dtype
    double* C;            // [N, N] W^T W (upper triangle referenced)
    double* spectrum;     // [N] ascending
    double* sing2;        // [rank] = spectrum[:rank] as computed (tramp_lsolve_set_rank copies it): the reference's `singular`
    rocblas_handle h;
    // per-call scratch, grown on demand
    double* A;            // [group][N][N]
    size_t group;
    double* g;            // [batch][N] right-hand sides / solutions
    void* tmp;            // [batch][max(M, N)] dtype
    double* scal;         // a_raw [batch]: the clipped, undamped a_new of the call
    rocblas_int* info;    // [group]
    size_t batch_cap;
};

namespace {

#define LS_RB(ctx, call)                                                                              \
    do {                                                                                              \
        rocblas_status s_ = (call);                                                                   \
        if (s_ != rocblas_status_success)                                                             \
            return tramp_fail(ctx, s_ == rocblas_status_memory_error ? TRAMP_ENOMEM : TRAMP_EHIP, #call, \
                              rocblas_status_to_string(s_));                                          \
    } while (0)

template <typename T>
__global__ __launch_bounds__(kBlock) void ls_to_f64_kernel(size_t count, const T* __restrict__ in, double* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        out[i] = (double)in[i];
}

// g = bz + t   (t = bx W in the channel's dtype)
template <typename T>
__global__ __launch_bounds__(kBlock) void ls_rhs_kernel(size_t count, const T* __restrict__ bz, const T* __restrict__ t,
                                                        double* __restrict__ g) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        g[i] = (double)bz[i] + (double)t[i];
}

// A[j] = az[b0 + j] I + ax[b0 + j] C, j < group (blockIdx.y)
__global__ __launch_bounds__(kBlock) void ls_form_kernel(size_t N, const double* __restrict__ C, const double* __restrict__ az,
                                                         const double* __restrict__ ax, size_t b0, double* __restrict__ A) {
    const size_t j = blockIdx.y, total = N * N;
    const double a = az[b0 + j], x = ax[b0 + j];
    double* Aj = A + j * total;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        double v = x * C[i];
        if (i / N == i % N) v += a;
        Aj[i] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void ls_from_f64_kernel(size_t count, const double* __restrict__ in, T* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        out[i] = (T)in[i];
}

// b_new = r (a_in + a_new) - b_in with the clipped, undamped a_new (tramp/base.py:250-255), damped against b_old;
// optional copy of r
template <typename T, typename R>
__global__ __launch_bounds__(kBlock) void ls_message_kernel(size_t len, const R* __restrict__ r,
                                                            const double* __restrict__ a_in,
                                                            const double* __restrict__ a_raw, const T* __restrict__ b_in,
                                                            double damping, const T* __restrict__ b_old,
                                                            T* __restrict__ b_new, T* __restrict__ r_out) {
    const size_t b = blockIdx.y;
    const double inv = a_in[b] + a_raw[b];
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (size_t)gridDim.x * kBlock) {
        const double ri = (double)r[b * len + i];
        double bn = ri * inv - (double)b_in[b * len + i];
        if (damping != 0.0) bn = damping * (double)b_old[b * len + i] + (1.0 - damping) * bn;
        b_new[b * len + i] = (T)bn;
        if (r_out) r_out[b * len + i] = (T)ri;
    }
}

// logZ_b = (g_b . rz_b) / 2 + sum_i log(2 pi / (az_b + ax_b spectrum_i)) / 2   (linear_channel.py:127-132)
__global__ __launch_bounds__(kBlock) void ls_logz_kernel(size_t N, const double* __restrict__ g, const double* __restrict__ rz,
                                                         const double* __restrict__ spectrum, const double* __restrict__ az,
                                                         const double* __restrict__ ax, double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    const double a = az[b], x = ax[b];
    double s = 0.0;
    for (size_t i = threadIdx.x; i < N; i += kBlock)
        s += g[b * N + i] * rz[b * N + i] + log(6.283185307179586476925 / (a + x * spectrum[i]));
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[b] = 0.5 * s;
}

__global__ __launch_bounds__(kBlock) void ls_mirror_kernel(size_t n, double* __restrict__ Cm) {
    const size_t total = n * n;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        const size_t i = idx % n, j = idx / n;                  // column-major (i, j)
        if (i > j) Cm[idx] = Cm[j + i * n];
    }
}

unsigned ls_grid(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    return (unsigned)(g > 4096 ? 4096 : (g ? g : 1));
}

int ls_reserve(tramp_ctx* ctx, tramp_lsolve* ls, size_t batch) {
    if (batch <= ls->batch_cap) return TRAMP_OK;
    if (ctx->capturing) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_lsolve", "workspace growth inside graph capture");
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (void* p : {(void*)ls->A, (void*)ls->g, ls->tmp, (void*)ls->scal, (void*)ls->info})
        if (p) (void)hipFree(p);
    ls->A = ls->g = ls->scal = nullptr;
    ls->tmp = nullptr;
    ls->info = nullptr;
    ls->batch_cap = 0;
    const size_t nn = ls->N * ls->N * sizeof(double);
    // instances factorised per strided-batched call: what fits TRAMP_LSOLVE_SCRATCH_MB (default 2048) of N x N systems
    static const size_t scratch_mb = getenv("TRAMP_LSOLVE_SCRATCH_MB") ? (size_t)atol(getenv("TRAMP_LSOLVE_SCRATCH_MB")) : 2048;
    size_t group = (scratch_mb << 20) / nn;
    if (group < 1) group = 1;
    if (group > batch) group = batch;
    const size_t mx = ls->M > ls->N ? ls->M : ls->N;
    hipError_t e = hipMalloc((void**)&ls->A, group * nn);
    if (e == hipSuccess) e = hipMalloc((void**)&ls->g, 2 * batch * ls->N * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ls->tmp, batch * mx * dtype_size(ls->dtype));
    if (e == hipSuccess) e = hipMalloc((void**)&ls->scal, 2 * batch * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ls->info, group * sizeof(rocblas_int));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_lsolve workspace", hipGetErrorString(e));
    }
    ls->group = group;
    ls->batch_cap = batch;
    return TRAMP_OK;
}

// rz (f64, in ls->g) for a batch; the right-hand sides are kept in the second half of ls->g
template <typename T>
int ls_solve(tramp_ctx* ctx, tramp_lsolve* ls, size_t batch, const double* az_d, const void* bz_d, const double* ax_d,
             const void* bx_d) {
    const size_t N = ls->N, M = ls->M;
    if (ctx->capturing)
        return tramp_fail(ctx, TRAMP_EINVAL, "tramp_lsolve", "the factorising branch cannot be captured in a graph");
    int rc = ls_reserve(ctx, ls, batch);
    if (rc) return rc;
    // t = bx W  [batch, N]
    rc = tramp_matvec_nn(ctx, ls->dtype, 0, M, N, batch, ls->W, bx_d, ls->tmp);
    if (rc) return rc;
    double* rhs = ls->g + batch * N;
    hipLaunchKernelGGL((ls_rhs_kernel<T>), dim3(ls_grid(batch * N)), dim3(kBlock), 0, ctx->stream, batch * N, (const T*)bz_d,
                       (const T*)ls->tmp, rhs);
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(ls->g, rhs, batch * N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    LS_RB(ctx, rocblas_set_stream(ls->h, ctx->stream));
    std::vector<rocblas_int> info_h(ls->group);
    for (size_t b0 = 0; b0 < batch; b0 += ls->group) {
        const size_t cnt = (batch - b0 < ls->group) ? batch - b0 : ls->group;
        hipLaunchKernelGGL(ls_form_kernel, dim3(ls_grid(N * N), (unsigned)cnt), dim3(kBlock), 0, ctx->stream, N, ls->C, az_d,
                           ax_d, b0, ls->A);
        TRAMP_LAUNCH_CHECK(ctx);
        // C is symmetric: its row-major image is its column-major image, and only the upper triangle is read
        LS_RB(ctx, rocsolver_dpotrf_strided_batched(ls->h, rocblas_fill_upper, (rocblas_int)N, ls->A, (rocblas_int)N,
                                                    (rocblas_stride)(N * N), ls->info, (rocblas_int)cnt));
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(info_h.data(), ls->info, cnt * sizeof(rocblas_int), hipMemcpyDeviceToHost,
                                            ctx->stream));
        TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t j = 0; j < cnt; ++j)
            if (info_h[j] != 0) {
                char msg[128];
                snprintf(msg, sizeof(msg), "az I + ax W^T W of instance %zu is not positive definite (leading minor %d)",
                         b0 + j, (int)info_h[j]);
                return tramp_fail(ctx, TRAMP_EINVAL, "singular matrix", msg);
            }
        LS_RB(ctx, rocsolver_dpotrs_strided_batched(ls->h, rocblas_fill_upper, (rocblas_int)N, 1, ls->A, (rocblas_int)N,
                                                    (rocblas_stride)(N * N), ls->g + b0 * N, (rocblas_int)N,
                                                    (rocblas_stride)N, (rocblas_int)cnt));
    }
    return TRAMP_OK;
}

template <typename T>
int ls_message_t(tramp_ctx* ctx, tramp_lsolve* ls, int dir, size_t batch, const double* az_d, const void* bz_d,
                 const double* ax_d, const void* bx_d, double damping, const double* a_old_d, const void* b_old_d,
                 double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    const size_t N = ls->N, M = ls->M;
    int rc = ls_solve<T>(ctx, ls, batch, az_d, bz_d, ax_d, bx_d);
    if (rc) return rc;
    double* a_raw = ls->scal;
    rc = lin_variance_launch(ctx, ls->sing2, (int)ls->rank, (int)ls->rank, (double)M, (double)N, dir, batch, az_d, ax_d,
                             damping, a_old_d, v_d, a_new_d, a_raw, nullptr);
    if (rc) return rc;
    if (dir == TRAMP_BWD) {
        dim3 g(ls_grid(N), (unsigned)batch);
        hipLaunchKernelGGL((ls_message_kernel<T, double>), g, dim3(kBlock), 0, ctx->stream, N, (const double*)ls->g, az_d,
                           a_raw, (const T*)bz_d, damping, (const T*)b_old_d, (T*)b_new_d, (T*)r_d);
    } else {
        // rx = rz W^T in the channel's dtype; the solutions are converted into the first N columns' worth of tmp2
        T* rz_t = (T*)(ls->g + batch * N);                    // the right-hand sides are no longer needed
        hipLaunchKernelGGL((ls_from_f64_kernel<T>), dim3(ls_grid(batch * N)), dim3(kBlock), 0, ctx->stream, batch * N,
                           (const double*)ls->g, rz_t);
        rc = tramp_matvec_nt(ctx, ls->dtype, 0, M, N, batch, ls->W, rz_t, ls->tmp);
        if (rc) return rc;
        dim3 g(ls_grid(M), (unsigned)batch);
        hipLaunchKernelGGL((ls_message_kernel<T, T>), g, dim3(kBlock), 0, ctx->stream, M, (const T*)ls->tmp, ax_d, a_raw,
                           (const T*)bx_d, damping, (const T*)b_old_d, (T*)b_new_d, (T*)r_d);
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

}  // namespace

extern "C" int tramp_lsolve_destroy(tramp_ctx* ctx, tramp_lsolve* ls) {
    if (!ls) return TRAMP_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (void* p : {(void*)ls->C, (void*)ls->spectrum, (void*)ls->sing2, (void*)ls->A, (void*)ls->g, ls->tmp,
                    (void*)ls->scal, (void*)ls->info})
        if (p) (void)hipFree(p);
    if (ls->h) (void)rocblas_destroy_handle(ls->h);
    delete ls;
    return TRAMP_OK;
}

extern "C" int tramp_lsolve_create(tramp_ctx* ctx, int dtype, size_t M, size_t N, const void* W_d, tramp_lsolve** out) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, out && W_d && M > 0 && N > 0);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F64 || dtype == TRAMP_F32);
    TRAMP_REQUIRE(ctx, N <= 46336 && M <= 0x7fffffff);                    // 32-bit library indexing of the N x N solves
    tramp_lsolve* ls = new tramp_lsolve();
    memset(ls, 0, sizeof(*ls));
    ls->dtype = dtype;
    ls->M = M;
    ls->N = N;
    ls->W = W_d;
    int rc = TRAMP_OK;
    double* W64 = nullptr;
    double* Cc = nullptr;
    auto fail = [&](int code) {
        if (W64) (void)hipFree(W64);
        if (Cc) (void)hipFree(Cc);
        (void)tramp_lsolve_destroy(ctx, ls);
        return code;
    };
    if (rocblas_create_handle(&ls->h) != rocblas_status_success)
        return fail(tramp_fail(ctx, TRAMP_EHIP, "tramp_lsolve_create", "rocblas_create_handle"));
    (void)rocblas_set_stream(ls->h, ctx->stream);
    hipError_t e = hipMalloc((void**)&ls->C, N * N * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ls->spectrum, N * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&Cc, N * N * sizeof(double));
    const double* Wd = (const double*)W_d;
    if (e == hipSuccess && dtype == TRAMP_F32) {
        e = hipMalloc((void**)&W64, M * N * sizeof(double));
        if (e == hipSuccess) {
            hipLaunchKernelGGL((ls_to_f64_kernel<float>), dim3(ls_grid(M * N)), dim3(kBlock), 0, ctx->stream, M * N,
                               (const float*)W_d, W64);
            Wd = W64;
        }
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(tramp_fail(ctx, TRAMP_ENOMEM, "tramp_lsolve_create", hipGetErrorString(e)));
    }
    // row-major W [M, N] is the column-major N x M matrix W^T: C = (W^T)(W^T)^T
    (void)hipMemsetAsync(ls->C, 0, N * N * sizeof(double), ctx->stream);
    const double one = 1.0, zero = 0.0;
    rocblas_status st = rocblas_dsyrk(ls->h, rocblas_fill_upper, rocblas_operation_none, (rocblas_int)N, (rocblas_int)M, &one,
                                      Wd, (rocblas_int)N, &zero, ls->C, (rocblas_int)N);
    if (st != rocblas_status_success)
        return fail(tramp_fail(ctx, TRAMP_EHIP, "rocblas_dsyrk", rocblas_status_to_string(st)));
    // "upper" of the column-major view = entries (i <= j) at C[i + j N]; mirror so that both triangles hold C
    // (the form kernel copies whole matrices, tramp_eigh reads the upper one)
    hipLaunchKernelGGL(ls_mirror_kernel, dim3(ls_grid(N * N)), dim3(kBlock), 0, ctx->stream, N, ls->C);
    e = hipMemcpyAsync(Cc, ls->C, N * N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return fail(tramp_fail(ctx, TRAMP_EHIP, "tramp_lsolve_create", hipGetErrorString(e)));
    rc = tramp_eigh(ctx, N, Cc, ls->spectrum);                            // ascending, like np.linalg.eigvalsh
    if (rc) return fail(rc);
    (void)hipStreamSynchronize(ctx->stream);
    if (W64) (void)hipFree(W64);
    (void)hipFree(Cc);
    *out = ls;
    return TRAMP_OK;
}

// spectrum_h[N]: eigenvalues of W^T W in ascending order (host)
extern "C" int tramp_lsolve_spectrum(tramp_ctx* ctx, const tramp_lsolve* ls, double* spectrum_h) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ls && spectrum_h);
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(spectrum_h, ls->spectrum, ls->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TRAMP_OK;
}

// singular = spectrum[:rank] (ascending order: see the header of this file)
extern "C" int tramp_lsolve_set_rank(tramp_ctx* ctx, tramp_lsolve* ls, size_t rank) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ls && rank <= ls->N && rank <= ls->M);
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ls->sing2) (void)hipFree(ls->sing2);
    ls->sing2 = nullptr;
    ls->rank = rank;
    TRAMP_HIP_CHECK(ctx, hipMalloc((void**)&ls->sing2, (rank ? rank : 1) * sizeof(double)));
    if (rank)
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(ls->sing2, ls->spectrum, rank * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_lsolve_message(tramp_ctx* ctx, tramp_lsolve* ls, int dir, size_t batch, const double* az_d,
                                    const void* bz_d, const double* ax_d, const void* bx_d, double damping,
                                    const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d,
                                    double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ls && ls->sing2 && az_d && bz_d && ax_d && bx_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!batch) return TRAMP_OK;
    return ls->dtype == TRAMP_F64
               ? ls_message_t<double>(ctx, ls, dir, batch, az_d, bz_d, ax_d, bx_d, damping, a_old_d, b_old_d, a_new_d, b_new_d,
                                      r_d, v_d)
               : ls_message_t<float>(ctx, ls, dir, batch, az_d, bz_d, ax_d, bx_d, damping, a_old_d, b_old_d, a_new_d, b_new_d,
                                     r_d, v_d);
}

extern "C" int tramp_lsolve_log_partition(tramp_ctx* ctx, tramp_lsolve* ls, size_t batch, const double* az_d,
                                          const void* bz_d, const double* ax_d, const void* bx_d, double* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ls && az_d && bz_d && ax_d && bx_d && out_d);
    if (!batch) return TRAMP_OK;
    int rc = ls->dtype == TRAMP_F64 ? ls_solve<double>(ctx, ls, batch, az_d, bz_d, ax_d, bx_d)
                                    : ls_solve<float>(ctx, ls, batch, az_d, bz_d, ax_d, bx_d);
    if (rc) return rc;
    hipLaunchKernelGGL(ls_logz_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, ls->N,
                       (const double*)(ls->g + batch * ls->N), (const double*)ls->g, ls->spectrum, az_d, ax_d, out_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}
