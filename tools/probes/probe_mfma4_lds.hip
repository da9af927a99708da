// Consumer-wave model: 16 MFMAs (4x4x4_4b f64) per stage, B operands from LDS prefetched ONE
// stage ahead (2 ds_read_b128 per stage, broadcast pattern of the GEMM kernels), A operands in
// registers.  MODE 0: A fixed; MODE 1: A also re-read from LDS every 16 stages (8 b128 reads).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, const double* in) {
    __shared__ v2d lds[4096];
    const int l = threadIdx.x, lane = l & 63;
    for (int i = l; i < 4096; i += 256) lds[i] = v2d{in[i & 63], in[(i + 1) & 63]};
    __syncthreads();
    v2d a[4][2];
    for (int i = 0; i < 4; ++i) for (int p = 0; p < 2; ++p) a[i][p] = v2d{in[lane] + i, in[(lane + 3) & 63] - p};
    double c[4][16];
    for (int i = 0; i < 4; ++i) for (int t = 0; t < 16; ++t) c[i][t] = 0;
    const int rd = 4 * (lane >> 4) + (lane & 3);
    int base = 0;
    v2d xa = lds[rd], xc = lds[16 + rd];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sg = 0; sg < 16; ++sg) {          // 2 pieces x 8 tile pairs = one k step
            const int p = sg >> 3, t = 2 * (sg & 7);
            const int nb = (base + (sg + 1) * 32) & 4095 & ~31;
            v2d na = lds[nb + rd], nc = lds[nb + 16 + rd];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    c[rt][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[rt][p][e], xa[e], c[rt][t], 0, 0, 0);
                    c[rt][t + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[rt][p][e], xc[e], c[rt][t + 1], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            xa = na; xc = nc;
        }
        base = (base + 512) & 4095;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) a[i][p] = lds[(base + 64 * (2 * i + p) + lane) & 4095];
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) for (int t = 0; t < 16; ++t) s += c[i][t];
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
    const int iters = 2000;
    for (int wpc : {4, 8}) {
        int blocks = 256 * wpc / 4;
        double n = (double)blocks * 4 * iters * 256;
        float ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, in); });
        printf("waves/CU=%d B from LDS (prefetch 1 stage), A fixed : %.1f TFLOP/s\n", wpc, n * 512 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, in); });
        printf("waves/CU=%d B and A from LDS                       : %.1f TFLOP/s\n", wpc, n * 512 / ms / 1e9);
    }
    return 0;
}
