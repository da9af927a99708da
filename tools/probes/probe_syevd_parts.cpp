// probe_syevd_parts.cpp - where does rocSOLVER's symmetric eigensolver spend its time?  sytrd (tridiagonalisation), stedc
// (divide and conquer on the tridiagonal matrix, eigenvectors of T), ormtr (back-transformation) and the whole syevd, at a
// few sizes, on a random Gram-like matrix.  Decides whether an own tridiagonalisation could shorten the set-up (round 6).
// Build: hipcc -O2 tools/probes/probe_syevd_parts.cpp -o tools/probes/build/probe_syevd_parts -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define RB(x) do { rocblas_status s_ = (x); if (s_ != rocblas_status_success) { printf("rocblas/rocsolver status %d at line %d\n", (int)s_, __LINE__); exit(1); } } while (0)

__global__ void fill(double* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i % n, c = i / n;
        unsigned long long h = (r < c ? r * n + c : c * n + r) * 0x9E3779B97F4A7C15ull + 12345;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        a[i] = ((h & 0xFFFFFF) / 16777216.0 - 0.5) + (r == c ? 0.5 * (double)(r % 97) / 97.0 : 0.0);
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    rocblas_handle h;
    RB(rocblas_create_handle(&h));
    for (int ai = 1; ai < argc; ++ai) {
        const int n = atoi(argv[ai]);
        double *A, *B, *D, *E, *tau, *C;
        rocblas_int* info;
        CK(hipMalloc(&A, (size_t)n * n * 8)); CK(hipMalloc(&B, (size_t)n * n * 8)); CK(hipMalloc(&C, (size_t)n * n * 8));
        CK(hipMalloc(&D, n * 8)); CK(hipMalloc(&E, n * 8)); CK(hipMalloc(&tau, n * 8)); CK(hipMalloc(&info, 4));
        fill<<<4096, 256>>>(A, n);
        CK(hipMemcpy(B, A, (size_t)n * n * 8, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
        double t0 = now();
        RB(rocsolver_dsytrd(h, rocblas_fill_lower, n, A, n, D, E, tau));
        CK(hipDeviceSynchronize());
        double t1 = now();
        RB(rocsolver_dstedc(h, rocblas_evect_tridiagonal, n, D, E, C, n, info));
        CK(hipDeviceSynchronize());
        double t2 = now();
        RB(rocsolver_dormtr(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, n, A, n, tau, C, n));
        CK(hipDeviceSynchronize());
        double t3 = now();
        RB(rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, B, n, D, E, info));
        CK(hipDeviceSynchronize());
        double t4 = now();
        const double n3 = (double)n * n * n;
        printf("n = %6d: sytrd %8.3f s (%.2e n^3; matrix read once per column = %.0f GB -> %.0f GB/s), stedc %8.3f s, ormtr %8.3f s (%.1f TFLOP/s), "
               "sum %8.3f s; syevd %8.3f s (%.2e n^3)\n", n, t1 - t0, (t1 - t0) / n3, n3 / 3 * 8 / 1e9, n3 / 3 * 8 / 1e9 / (t1 - t0), t2 - t1, t3 - t2,
               2 * n3 / (t3 - t2) / 1e12, t3 - t0, t4 - t3, (t4 - t3) / n3);
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(C); hipFree(D); hipFree(E); hipFree(tau); hipFree(info);
    }
    return 0;
}
