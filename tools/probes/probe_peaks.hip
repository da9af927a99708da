// Measures on the box: (1) v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 issue rate,
// (2) HBM streaming read bandwidth (double2 loads), (3) copy bandwidth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma64(double* out, int iters, double a, double b) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void mfma32(float* out, int iters, float a, float b) {
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void fma64(double* out, int iters, double a, double b) {
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    for (int i = 0; i < iters; ++i) {
        c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
        c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
        a += 1e-9;
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
// co-issue test: waves 0,1 of each block run the MFMA loop, waves 2,3 the v_fma_f64 loop
// (mode 0: all MFMA, 1: all FMA, 2: half/half)
__global__ __launch_bounds__(256) void mixed64(double* out, int iters, double a, double b, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 2);
    double res = 0;
    if (do_mfma) {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
        for (int i = 0; i < iters * 16; ++i) {       // 16*8*64*2 = 16384 flop per iteration per wave = 8 MFMAs worth
            c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
            c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
        }
        res = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    }
    out[blockIdx.x * 256 + threadIdx.x] = res;
}
__global__ __launch_bounds__(256) void stream_read(const double2* __restrict__ in, size_t n, double* out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double2 v = in[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345) out[0] = acc;
}
__global__ __launch_bounds__(256) void stream_copy(const double2* __restrict__ in, double2* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = in[i];
}
// unrolled copy: each thread moves U independent 16-byte pieces per trip
template <int U>
__global__ __launch_bounds__(256) void stream_copy_u(const double2* __restrict__ in, double2* __restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) o[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) o[i] = in[i];
}
// block-contiguous variant: a block owns a contiguous 16 KiB tile per trip
template <int U>
__global__ __launch_bounds__(256) void stream_copy_tile(const double2* __restrict__ in, double2* __restrict__ o, size_t n) {
    const size_t tile = 256 * U;
    for (size_t t = blockIdx.x; t * tile < n; t += gridDim.x) {
        const size_t base = t * tile + threadIdx.x;
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = in[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * 256 < n) o[base + u * 256] = v[u];
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    double* out; hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    for (int wpc : {4, 8, 16}) {       // waves per CU
        int blocks = 256 * wpc / 4;
        float ms = timeit([&] { hipLaunchKernelGGL(mfma64, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0); }, 3);
        double fl = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4 waves/CU=%d: %.1f TFLOP/s\n", wpc, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(mfma32, dim3(blocks), dim3(256), 0, 0, (float*)out, iters, 1.0f, 2.0f); }, 3);
        printf("mfma_f32_16x16x4 waves/CU=%d: %.1f TFLOP/s\n", wpc, fl / ms / 1e9);
    }
    {
        int blocks = 256 * 8;
        float ms = timeit([&] { hipLaunchKernelGGL(fma64, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0); }, 3);
        printf("v_fma_f64: %.1f TFLOP/s\n", (double)blocks * 256 * iters * 8 * 2.0 / ms / 1e9);
    }
    for (int mode : {0, 1, 2}) {
        int blocks = 256 * 4;     // 16 waves per CU
        int it = 5000;
        float ms = timeit([&] { hipLaunchKernelGGL(mixed64, dim3(blocks), dim3(256), 0, 0, out, it, 1.0, 2.0, mode); }, 3);
        double wm = (mode == 0 ? 4 : mode == 2 ? 2 : 0), wf = 4 - wm;
        double fl = (double)blocks * it * (wm * 4 * 2048.0 + wf * 16 * 8 * 64 * 2.0);
        printf("mixed f64 mode=%d (0 mfma,1 fma,2 half/half): %.3f ms  %.1f TFLOP/s total\n", mode, ms, fl / ms / 1e9);
    }
    size_t bytes = 8ull << 30;
    double2 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    size_t n = bytes / 16;
    for (int blocks : {2048, 4096, 8192, 16384}) {
        float ms = timeit([&] { hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, a, n, out); }, 5);
        printf("stream read  %5d blocks: %.0f GB/s\n", blocks, bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(stream_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        printf("stream copy  %5d blocks: %.0f GB/s (read+write)\n", blocks, 2.0 * bytes / ms / 1e6);
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float ms = timeit([&] { hipLaunchKernelGGL(stream_copy_u<4>, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        printf("copy unroll4 strided %5d blocks: %.0f GB/s\n", blocks, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(stream_copy_u<8>, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        printf("copy unroll8 strided %5d blocks: %.0f GB/s\n", blocks, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(stream_copy_tile<4>, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        printf("copy tile4          %5d blocks: %.0f GB/s\n", blocks, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(stream_copy_tile<8>, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        printf("copy tile8          %5d blocks: %.0f GB/s\n", blocks, 2.0 * bytes / ms / 1e6);
    }
    {
        float ms = timeit([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("hipMemcpy D2D: %.0f GB/s (read+write)\n", 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
