// Issue rate of the two f64 MFMA shapes on the box, with a cycle counter:
//   v_mfma_f64_16x16x4_f64   (2048 flop)      v_mfma_f64_4x4x4_4b_f64 (4 blocks, 512 flop)
// hipcc --offload-arch=gfx950 -O3 probe_mfma64.hip -o build/probe_mfma64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma16(double* out, long long* cyc, int iters, double a, double b) {
    d4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    d4 s = c[0];
    for (int i = 1; i < NACC; ++i) s += c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma4(double* out, long long* cyc, int iters, double a, double b) {
    double c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[k], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 16 * 8 * sizeof(double));
    hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int wpc : {4, 8, 12, 16}) {
        int blocks = 256 * wpc / 4;
        long long h;
#define RUN16(N) { float ms = timeit([&] { hipLaunchKernelGGL(mfma16<N>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0, 2.0); }); \
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); double n = (double)blocks * 4 * iters * N; \
        printf("16x16x4 nacc=%d waves/CU=%2d: %6.1f TFLOP/s  %.1f clk/MFMA/SIMD (wave-clock %.1f per MFMA of this wave)\n", N, wpc, n * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / n, (double)h / (iters * N)); }
#define RUN4(N) { float ms = timeit([&] { hipLaunchKernelGGL(mfma4<N>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0, 2.0); }); \
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); double n = (double)blocks * 4 * iters * N; \
        printf("4x4x4_4b nacc=%d waves/CU=%2d: %6.1f TFLOP/s  (wave-clock %.1f per MFMA of this wave)\n", N, wpc, n * 512 / ms / 1e9, (double)h / (iters * N)); }
        RUN16(1) RUN16(4) RUN16(8)
        RUN4(1) RUN4(4) RUN4(16)
    }
    return 0;
}
