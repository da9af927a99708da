// probe_gemm_sliced.hip - the int8-digit dense pass (tramp_amd/csrc/gemm_sliced.h) alone: correctness against a long-double
// host product on a small shape, then the speed of the headline shape (50000 x 100000 x 64) for S = 6, 7, 8 digits.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/probes/probe_gemm_sliced.hip -o tools/probes/build/probe_gemm_sliced
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "../../tramp_amd/csrc/gemm_sliced.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static double rnd(unsigned long long& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double u1 = ((s >> 11) + 0.5) / 9007199254740992.0;
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double u2 = ((s >> 11) + 0.5) / 9007199254740992.0;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

__global__ void fill_bytes(int8_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        p[i] = (int8_t)((int)(h % 129) - 64);
    }
}
__global__ void fold8(size_t len, size_t B, int nsplit, const double* partials, double* out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (idx >= len) return;
    double s = 0;
    for (int k = 0; k < nsplit; ++k) s += partials[((size_t)k * B + b) * len + idx];
    out[b * len + idx] = s;
}

template <int S, int DBG = 0, int NW = 4>
static void launch(size_t R, size_t nkb, size_t B, size_t nbt, const int8_t* ms, const double* rs, const int8_t* xs, const double* xsc,
                   unsigned nsplit, double* partials) {
    const size_t ntile = (R + 15) / 16, nrb = (ntile + NW - 1) / NW;
    const size_t kps = (nkb + nsplit - 1) / nsplit;
    hipLaunchKernelGGL((gemm_sliced_kernel<S, 4, DBG, NW>), dim3((unsigned)(nrb * nsplit), (unsigned)((B + 63) / 64)), dim3(NW * 64), 0, 0,
                       R, nkb, B, nbt, ms, rs, xs, xsc, kps, nsplit, partials);
}

template <int S> static void check(size_t R, size_t C, size_t B) {
    std::vector<double> hm(R * C), hx(B * C);
    unsigned long long seed = 12345;
    for (auto& v : hm) v = rnd(seed) / sqrt((double)C);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = rnd(seed) * ((i % 7 == 0) ? 5.0 : 0.3);
    double *dm, *dx, *rs, *xsc, *partials, *out;
    unsigned long long *am, *ax;
    int8_t *ms, *xs;
    const size_t nkb = (C + 63) / 64, ntile = (R + 15) / 16, nbt = (B + 63) / 64 * 4;
    CK(hipMalloc(&dm, R * C * 8)); CK(hipMalloc(&dx, B * C * 8));
    CK(hipMemcpy(dm, hm.data(), R * C * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), B * C * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&am, ntile * 16 * 8)); CK(hipMalloc(&ax, nbt * 16 * 8)); CK(hipMalloc(&rs, ntile * 16 * 8)); CK(hipMalloc(&xsc, nbt * 16 * 8));
    CK(hipMemset(am, 0, ntile * 16 * 8)); CK(hipMemset(ax, 0, nbt * 16 * 8));
    CK(hipMalloc(&ms, ntile * nkb * S * kSlFrag)); CK(hipMalloc(&xs, nkb * nbt * S * kSlFrag));
    const unsigned nsplit = 8;
    CK(hipMalloc(&partials, nsplit * B * R * 8)); CK(hipMalloc(&out, B * R * 8));
    hipLaunchKernelGGL((sl_absmax_kernel<double, false>), dim3((unsigned)R, 1), dim3(kBlock), 0, 0, R, C, C, dm, am);
    hipLaunchKernelGGL((sl_absmax_kernel<double, false>), dim3((unsigned)B, 1), dim3(kBlock), 0, 0, B, C, C, dx, ax);
    hipLaunchKernelGGL((sl_slice_kernel<double, S, false, false>), dim3((unsigned)((nkb * 4 + 15) / 16), (unsigned)ntile), dim3(kBlock), 0, 0,
                       R, ntile * 16, C, C, dm, am, nkb, ms, rs);
    hipLaunchKernelGGL((sl_slice_kernel<double, S, false, true>), dim3((unsigned)((nbt * 16 + kBlock - 1) / kBlock), (unsigned)(nkb * 4)), dim3(kBlock), 0, 0,
                       B, nbt * 16, C, C, dx, ax, nbt, xs, xsc);
    launch<S>(R, nkb, B, nbt, ms, rs, xs, xsc, nsplit, partials);
    hipLaunchKernelGGL(fold8, dim3((unsigned)((R + 255) / 256), (unsigned)B), dim3(256), 0, 0, R, B, (int)nsplit, partials, out);
    CK(hipDeviceSynchronize());
    std::vector<double> ho(B * R);
    CK(hipMemcpy(ho.data(), out, B * R * 8, hipMemcpyDeviceToHost));
    double worst = 0, worst64 = 0, scale = 0;
    for (size_t b = 0; b < B; ++b)
        for (size_t r = 0; r < R; ++r) {
            long double s = 0;
            double s64 = 0;
            for (size_t c = 0; c < C; ++c) { s += (long double)hm[r * C + c] * hx[b * C + c]; s64 = fma(hm[r * C + c], hx[b * C + c], s64); }
            worst = std::max(worst, (double)fabsl(s - ho[b * R + r]));
            worst64 = std::max(worst64, (double)fabsl(s - s64));
            scale = std::max(scale, (double)fabsl(s));
        }
    printf("S=%d  R=%zu C=%zu B=%zu: max |sliced - exact| = %.3e, max |f64 fma chain - exact| = %.3e, max |exact| = %.3e\n", S, R, C, B, worst, worst64, scale);
    hipFree(dm); hipFree(dx); hipFree(am); hipFree(ax); hipFree(rs); hipFree(xsc); hipFree(ms); hipFree(xs); hipFree(partials); hipFree(out);
}

template <int S, int DBG = 0, int NW = 4> static void speed(size_t R, size_t C, size_t B, unsigned nsplit, bool zero = false) {
    const size_t nkb = (C + 63) / 64, ntile = (R + 15) / 16, nbt = (B + 63) / 64 * 4;
    int8_t *ms, *xs;
    double *rs, *xsc, *partials;
    CK(hipMalloc(&ms, ntile * nkb * S * kSlFrag)); CK(hipMalloc(&xs, nkb * nbt * S * kSlFrag));
    CK(hipMalloc(&rs, ntile * 16 * 8)); CK(hipMalloc(&xsc, nbt * 16 * 8)); CK(hipMalloc(&partials, (size_t)nsplit * B * R * 8));
    hipLaunchKernelGGL(fill_bytes, dim3(8192), dim3(256), 0, 0, ms, ntile * nkb * S * kSlFrag);
    hipLaunchKernelGGL(fill_bytes, dim3(8192), dim3(256), 0, 0, xs, nkb * nbt * S * kSlFrag);
    if (zero) { CK(hipMemset(ms, 0, ntile * nkb * S * kSlFrag)); CK(hipMemset(xs, 0, nkb * nbt * S * kSlFrag)); }
    CK(hipMemset(rs, 0, ntile * 16 * 8)); CK(hipMemset(xsc, 0, nbt * 16 * 8));
    for (int i = 0; i < 2; ++i) launch<S, DBG, NW>(R, nkb, B, nbt, ms, rs, xs, xsc, nsplit, partials);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch<S, DBG, NW>(R, nkb, B, nbt, ms, rs, xs, xsc, nsplit, partials);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms_; hipEventElapsedTime(&ms_, e0, e1); ms_ /= reps;
    const double bytes = (double)ntile * nkb * S * kSlFrag, flops = 2.0 * R * C * B;
    printf("S=%d NW=%d%s%s  %zu x %zu x %zu split %u: %8.3f ms  matrix stream %6.0f GB/s (%.3f of 8 TB/s)  f64-equivalent %6.1f TFLOP/s  int8 %7.1f TOP/s\n", S, NW, DBG ? " [matrix not advanced]" : "", zero ? " [zero data]" : "", R, C, B, nsplit,
           ms_, bytes / ms_ / 1e6, bytes / ms_ / 1e6 / 8000.0, flops / ms_ / 1e9, flops * (S * (S + 1) / 2) / ms_ / 1e9);
    fflush(stdout);
    hipFree(ms); hipFree(xs); hipFree(rs); hipFree(xsc); hipFree(partials);
}

int main(int argc, char** argv) {
    const size_t R = argc > 1 ? atol(argv[1]) : 50000, C = argc > 2 ? atol(argv[2]) : 100000, B = argc > 3 ? atol(argv[3]) : 64;
    check<8>(200, 1000, 64);
    check<8>(333, 4100, 50);
    check<7>(333, 4100, 50);
    check<6>(333, 4100, 50);
    check<4>(333, 4100, 50);
    check<7>(300, 18000, 20);
    speed<7>(R, C, B, 8);
    speed<7, 0, 8>(R, C, B, 8);
    speed<7>(R, C, B, 8);
    speed<7, 0, 8>(R, C, B, 8);
    speed<7, 0, 8>(R, C, B, 16);
    speed<8>(R, C, B, 8);
    speed<8, 0, 8>(R, C, B, 8);
    speed<7>(R, C, B, 16);
    speed<7>(R, C, B, 8, true);
    speed<7, 1>(R, C, B, 8);
    speed<6>(R, C, B, 8);
    speed<4>(R, C, B, 8);
    speed<7>(R, C, 512, 8);
    return 0;
}
