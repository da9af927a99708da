// Does v_mfma_f64_4x4x4_4b keep its 16-cycle issue when every instruction reads different A / B
// registers (as a GEMM does), and with LDS reads interleaved?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, const double* in) {
    __shared__ v2d lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 256) lds[i] = v2d{in[i & 63], in[(i + 1) & 63]};
    __syncthreads();
    double a[4], b[4], c[16];
    for (int i = 0; i < 4; ++i) { a[i] = in[l & 63] + i; b[i] = in[(l + 7) & 63] * (i + 1); }
    for (int i = 0; i < 16; ++i) c[i] = 0;
    int off = l & 15;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[0], c[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
        } else {
            v2d x0 = lds[off], x1 = lds[off + 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 8; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], (i & 1) ? x0[0] : x1[1], c[i], 0, 0, 0);
            off = (off + 32) & 1023 & ~15 | (l & 15);
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * 256 + l] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
    double *out, *in; hipMalloc(&out, 1 << 24); hipMalloc(&in, 64 * 8);
    double h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 1e-3;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int wpc : {4, 8}) {
        int blocks = 256 * wpc / 4;
        double n = (double)blocks * 4 * iters * 16;
        float ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, in); });
        printf("waves/CU=%d same a,b      : %.1f TFLOP/s\n", wpc, n * 512 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, in); });
        printf("waves/CU=%d 4 a x 4 b     : %.1f TFLOP/s\n", wpc, n * 512 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, in); });
        printf("waves/CU=%d + LDS operands: %.1f TFLOP/s\n", wpc, n * 512 / ms / 1e9);
    }
    return 0;
}
