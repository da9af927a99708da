// Stand-alone bandwidth probe of the tridiagonalisation's symv (csrc/tridiag_eig.h: td_symv_lower) against variants and a plain
// read of the same bytes.   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_symv.hip -o build_ab/probe_symv && build_ab/probe_symv 32768
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kTile = 256;

// ---------------------------------------------------------------- variant A: the shipped kernel (8-byte loads, lane = row)
template <bool FULL>
__device__ __forceinline__ void groupA(const double* __restrict__ At, size_t lda, size_t m, size_t r0, size_t cg, bool diag, int lane,
                                       const double* __restrict__ vcol16, const double (&vr)[4], double (&racc)[4], double (&cacc)[16]) {
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const size_t c = cg + cc;
        const double vc = vcol16[cc];
        const double* col = At + c * lda + r0 + lane;
        double a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (FULL) a[s] = col[s * 64];
            else {
                const size_t r = r0 + (size_t)s * 64 + lane;
                a[s] = ((c < m) && (r < m) && (!diag || r >= c)) ? col[s * 64] : 0.0;
            }
        }
        double cs = 0.0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            racc[s] = fma(a[s], vc, racc[s]);
            const bool strict = FULL || !diag || (r0 + (size_t)s * 64 + lane > c);
            cs = fma(strict ? a[s] : 0.0, vr[s], cs);
        }
        cacc[cc] = cs;
    }
}

template <int MAPPING>
__global__ __launch_bounds__(256, 2) void symvA(const double* __restrict__ At, size_t lda, size_t m, const double* __restrict__ v,
                                                double* __restrict__ row_part, double* __restrict__ col_part, unsigned nbt) {
    unsigned Js, I;
    if (MAPPING == 0) {
        Js = blockIdx.x; I = blockIdx.y;
        if (Js > I) return;
    } else {
        // triangular enumeration: block t -> (I, Js), Js <= I
        const unsigned t = blockIdx.x;
        I = (unsigned)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((size_t)(I + 1) * (I + 2) / 2 <= t) ++I;
        while ((size_t)I * (I + 1) / 2 > t) --I;
        Js = t - (unsigned)((size_t)I * (I + 1) / 2);
    }
    __shared__ double vrow[kTile], vcol[kTile], rsum[4][kTile];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t r0 = (size_t)I * kTile, c0 = (size_t)Js * kTile;
    vrow[tid] = (r0 + tid < m) ? v[r0 + tid] : 0.0;
    vcol[tid] = (c0 + tid < m) ? v[c0 + tid] : 0.0;
    __syncthreads();
    double racc[4] = {0.0, 0.0, 0.0, 0.0};
    const bool diag = (I == Js);
    const bool full = !diag && r0 + kTile <= m;
    const double vr[4] = {vrow[lane], vrow[64 + lane], vrow[128 + lane], vrow[192 + lane]};
    for (int g = 0; g < 4; ++g) {
        const size_t cg = c0 + (size_t)w * 64 + (size_t)g * 16;
        double cacc[16];
        if (full) groupA<true>(At, lda, m, r0, cg, diag, lane, vcol + w * 64 + g * 16, vr, racc, cacc);
        else groupA<false>(At, lda, m, r0, cg, diag, lane, vcol + w * 64 + g * 16, vr, racc, cacc);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const bool up = (lane & off) != 0;
#pragma unroll
            for (int i = 0; i < off; ++i) {
                const double keep = up ? cacc[i + off] : cacc[i];
                const double send = up ? cacc[i] : cacc[i + off];
                cacc[i] = keep + __shfl_xor(send, off);
            }
        }
        double t = cacc[0];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        const size_t c = cg + (lane & 15);
        if (lane < 16 && c < m) col_part[(size_t)I * m + c] = t;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) rsum[w][s * 64 + lane] = racc[s];
    __syncthreads();
    if (r0 + tid < m) row_part[(size_t)Js * m + r0 + tid] = (rsum[0][tid] + rsum[1][tid]) + (rsum[2][tid] + rsum[3][tid]);
}

// ---------------------------------------------------------------- variant B: 16-byte loads, lane = two consecutive rows
typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) d2u { double x, y; };

template <bool FULL>
__device__ __forceinline__ void groupB(const double* __restrict__ At, size_t lda, size_t m, size_t r0, size_t cg, bool diag, int lane,
                                       const double* __restrict__ vcol16, const double (&vr)[4], double (&racc)[4], double (&cacc)[16]) {
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const size_t c = cg + cc;
        const double vc = vcol16[cc];
        const double* col = At + c * lda + r0 + 2 * lane;
        double a[4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (FULL) {
                const d2u t = *reinterpret_cast<const d2u*>(col + s * 128);
                a[2 * s] = t.x;
                a[2 * s + 1] = t.y;
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const size_t r = r0 + (size_t)s * 128 + 2 * lane + h;
                    a[2 * s + h] = ((c < m) && (r < m) && (!diag || r >= c)) ? col[s * 128 + h] : 0.0;
                }
            }
        }
        double cs = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            racc[q] = fma(a[q], vc, racc[q]);
            const size_t r = r0 + (size_t)(q >> 1) * 128 + 2 * lane + (q & 1);
            const bool strict = FULL || !diag || (r > c);
            cs = fma(strict ? a[q] : 0.0, vr[q], cs);
        }
        cacc[cc] = cs;
    }
}

__global__ __launch_bounds__(256, 2) void symvB(const double* __restrict__ At, size_t lda, size_t m, const double* __restrict__ v,
                                                double* __restrict__ row_part, double* __restrict__ col_part, unsigned nbt) {
    const unsigned Js = blockIdx.x, I = blockIdx.y;
    if (Js > I) return;
    __shared__ double vrow[kTile], vcol[kTile], rsum[4][kTile];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t r0 = (size_t)I * kTile, c0 = (size_t)Js * kTile;
    vrow[tid] = (r0 + tid < m) ? v[r0 + tid] : 0.0;
    vcol[tid] = (c0 + tid < m) ? v[c0 + tid] : 0.0;
    __syncthreads();
    double racc[4] = {0.0, 0.0, 0.0, 0.0};
    const bool diag = (I == Js);
    const bool full = !diag && r0 + kTile <= m;
    // rows of this lane: 2 lane, 2 lane + 1, 128 + 2 lane, 128 + 2 lane + 1
    const double vr[4] = {vrow[2 * lane], vrow[2 * lane + 1], vrow[128 + 2 * lane], vrow[128 + 2 * lane + 1]};
    for (int g = 0; g < 4; ++g) {
        const size_t cg = c0 + (size_t)w * 64 + (size_t)g * 16;
        double cacc[16];
        if (full) groupB<true>(At, lda, m, r0, cg, diag, lane, vcol + w * 64 + g * 16, vr, racc, cacc);
        else groupB<false>(At, lda, m, r0, cg, diag, lane, vcol + w * 64 + g * 16, vr, racc, cacc);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const bool up = (lane & off) != 0;
#pragma unroll
            for (int i = 0; i < off; ++i) {
                const double keep = up ? cacc[i + off] : cacc[i];
                const double send = up ? cacc[i] : cacc[i + off];
                cacc[i] = keep + __shfl_xor(send, off);
            }
        }
        double t = cacc[0];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        const size_t c = cg + (lane & 15);
        if (lane < 16 && c < m) col_part[(size_t)I * m + c] = t;
    }
    rsum[w][2 * lane] = racc[0];
    rsum[w][2 * lane + 1] = racc[1];
    rsum[w][128 + 2 * lane] = racc[2];
    rsum[w][128 + 2 * lane + 1] = racc[3];
    __syncthreads();
    if (r0 + tid < m) row_part[(size_t)Js * m + r0 + tid] = (rsum[0][tid] + rsum[1][tid]) + (rsum[2][tid] + rsum[3][tid]);
}


// ---------------------------------------------------------------- variant C: tiles of 64 S rows x 256 columns (S slabs per lane, groups of G columns)
template <bool FULL, int S, int G>
__device__ __forceinline__ void groupC(const double* __restrict__ At, size_t lda, size_t m, size_t r0, size_t cg, int lane,
                                       const double* __restrict__ vcolg, const double (&vr)[S], double (&racc)[S], double (&cacc)[G]) {
#pragma unroll
    for (int cc = 0; cc < G; ++cc) {
        const size_t c = cg + cc;
        const double vc = vcolg[cc];
        const double* col = At + c * lda + r0 + lane;
        double a[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (FULL) a[s] = col[s * 64];
            else {
                const size_t r = r0 + (size_t)s * 64 + lane;
                a[s] = ((c < m) && (r < m) && (r >= c)) ? col[s * 64] : 0.0;
            }
        }
        double cs = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            racc[s] = fma(a[s], vc, racc[s]);
            const bool strict = FULL || (r0 + (size_t)s * 64 + lane > c);
            cs = fma(strict ? a[s] : 0.0, vr[s], cs);
        }
        cacc[cc] = cs;
    }
}

template <int S, int G, int OCC>
__global__ __launch_bounds__(256, OCC) void symvC(const double* __restrict__ At, size_t lda, size_t m, const double* __restrict__ v,
                                                double* __restrict__ row_part, double* __restrict__ col_part) {
    // tile (I, Js): rows [R I, +R), R = 64 S, columns [256 Js, +256), Js <= Q I + Q - 1, Q = R / 256;  t = Q I (I + 1) / 2 + Js
    constexpr int R = 64 * S, Q = R / 256;
    const unsigned t = blockIdx.x;
    unsigned I = (unsigned)((sqrt(8.0 * (double)t / Q + 1.0) - 1.0) * 0.5);
    while ((size_t)Q * (I + 1) * (I + 2) / 2 <= t) ++I;
    while ((size_t)Q * I * (I + 1) / 2 > t) --I;
    const unsigned Js = t - (unsigned)((size_t)Q * I * (I + 1) / 2);
    const size_t r0 = (size_t)I * R, c0 = (size_t)Js * 256;
    if (c0 >= m) return;
    __shared__ double vrow[R], vcol[256], rsum[4][R];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int h = 0; h < Q; ++h) vrow[h * 256 + tid] = (r0 + h * 256 + tid < m) ? v[r0 + h * 256 + tid] : 0.0;
    vcol[tid] = (c0 + tid < m) ? v[c0 + tid] : 0.0;
    __syncthreads();
    double racc[S], vr[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        racc[s] = 0.0;
        vr[s] = vrow[s * 64 + lane];
    }
    const bool full = (c0 + 256 <= r0) && (r0 + R <= m);
    for (int g = 0; g < 64 / G; ++g) {
        const size_t cg = c0 + (size_t)w * 64 + (size_t)g * G;
        double cacc[G];
        if (full) groupC<true, S, G>(At, lda, m, r0, cg, lane, vcol + w * 64 + g * G, vr, racc, cacc);
        else groupC<false, S, G>(At, lda, m, r0, cg, lane, vcol + w * 64 + g * G, vr, racc, cacc);
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const bool up = (lane & off) != 0;
#pragma unroll
            for (int i = 0; i < off; ++i) {
                const double keep = up ? cacc[i + off] : cacc[i];
                const double send = up ? cacc[i] : cacc[i + off];
                cacc[i] = keep + __shfl_xor(send, off);
            }
        }
        double tt = cacc[0];                                             // column (lane & (G - 1)) of the group, summed over G lanes
#pragma unroll
        for (int off = G; off < 64; off <<= 1) tt += __shfl_xor(tt, off);
        const size_t c = cg + (lane & (G - 1));
        if (lane < G && c < m) col_part[(size_t)I * m + c] = tt;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) rsum[w][s * 64 + lane] = racc[s];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < Q; ++h) {
        const int rr = h * 256 + tid;
        if (r0 + rr < m) row_part[(size_t)Js * m + r0 + rr] = (rsum[0][rr] + rsum[1][rr]) + (rsum[2][rr] + rsum[3][rr]);
    }
}

// ---------------------------------------------------------------- plain read of the lower-triangle tiles (ceiling)
__global__ __launch_bounds__(256) void read_tiles(const double* __restrict__ At, size_t lda, size_t m, double* __restrict__ out) {
    const unsigned Js = blockIdx.x, I = blockIdx.y;
    if (Js > I) return;
    const size_t r0 = (size_t)I * kTile, c0 = (size_t)Js * kTile;
    if (r0 + kTile > m) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double acc = 0.0;
    for (int cc = 0; cc < 64; ++cc) {
        const double* col = At + (c0 + w * 64 + cc) * lda + r0 + lane;
        acc += (col[0] + col[64]) + (col[128] + col[192]);
    }
    if (acc == 123.456) out[0] = acc;
}
__global__ __launch_bounds__(256) void read_flat(const double* __restrict__ A, size_t count, double* __restrict__ out) {
    const d2* p = reinterpret_cast<const d2*>(A);
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count / 2; i += (size_t)gridDim.x * 256) {
        const d2 t = p[i];
        acc += t.x + t.y;
    }
    if (acc == 123.456) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 16384;
    const size_t off = argc > 2 ? atol(argv[2]) : 1;              // sub-block offset (alignment as in the reduction)
    const size_t m = n - off;
    double *A, *v, *rp, *cp, *out;
    const unsigned nbt = (unsigned)((m + kTile - 1) / kTile);
    CK(hipMalloc(&A, n * n * 8));
    CK(hipMalloc(&v, n * 8));
    CK(hipMalloc(&rp, (size_t)nbt * n * 8));
    CK(hipMalloc(&cp, (size_t)nbt * n * 8));
    CK(hipMalloc(&out, 64));
    std::vector<double> h(n * n), hv(n);
    srand(1);
    for (size_t j = 0; j < n; ++j)
        for (size_t i = j; i < n; ++i) {
            const double x = (double)rand() / RAND_MAX - 0.5;
            h[i + j * n] = x;
            h[j + i * n] = x;
        }
    for (size_t i = 0; i < n; ++i) hv[i] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), n * 8, hipMemcpyHostToDevice));
    const double* At = A + off + off * n;
    const double* vv = v + off;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = 8.0 * (double)m * (double)m / 2.0;
    int mode = 0;
    auto check = [&](const char* name) {
        if (n > 8192) return;
        std::vector<double> hr((size_t)nbt * m), hc((size_t)nbt * m);
        CK(hipMemcpy(hr.data(), rp, (size_t)nbt * m * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), cp, (size_t)nbt * m * 8, hipMemcpyDeviceToHost));
        double err = 0.0;
        for (size_t q = 0; q < m; ++q) {
            const unsigned I = (unsigned)(q / kTile);
            double y = 0.0;
            for (unsigned Js = 0; Js <= I; ++Js) y += hr[(size_t)Js * m + q];
            if (mode == 0) for (unsigned Ii = I; Ii < nbt; ++Ii) y += hc[(size_t)Ii * m + q];
            else for (unsigned Ii = (unsigned)(q / (256 * mode)); Ii < (nbt + mode - 1) / mode; ++Ii) y += hc[(size_t)Ii * m + q];
            double ref = 0.0;
            for (size_t c = 0; c < m; ++c) ref += h[(off + q) + (off + c) * n] * hv[off + c];
            err = fmax(err, fabs(y - ref));
        }
        printf("   %s: max |y - ref| = %.2e\n", name, err);
    };
    auto time = [&](const char* name, auto launch, bool chk) {
        CK(hipMemset(rp, 0, (size_t)nbt * n * 8));
        CK(hipMemset(cp, 0, (size_t)nbt * n * 8));
        launch();
        CK(hipDeviceSynchronize());
        if (chk) check(name);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 10; ++it) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms / 10);
        }
        printf("%-28s m = %zu: %8.1f us  %6.2f TB/s\n", name, m, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    time("read_flat (same bytes)", [&] { hipLaunchKernelGGL(read_flat, dim3(2048), dim3(256), 0, 0, A, (size_t)(bytes / 8), out); }, false);
    time("read_tiles", [&] { hipLaunchKernelGGL(read_tiles, dim3(nbt, nbt), dim3(256), 0, 0, At, n, m, out); }, false);
    time("symv A (shipped)", [&] { hipLaunchKernelGGL((symvA<0>), dim3(nbt, nbt), dim3(256), 0, 0, At, n, m, vv, rp, cp, nbt); }, true);
    time("symv A, triangular grid", [&] { hipLaunchKernelGGL((symvA<1>), dim3(nbt * (nbt + 1) / 2), dim3(256), 0, 0, At, n, m, vv, rp, cp, nbt); }, true);
    time("symv B (16-byte loads)", [&] { hipLaunchKernelGGL(symvB, dim3(nbt, nbt), dim3(256), 0, 0, At, n, m, vv, rp, cp, nbt); }, true);
    mode = 2;
    { const unsigned nbr = (unsigned)((m + 511) / 512);
      const dim3 g(nbr * (nbr + 1));
      time("C 512x256 G=8 occ2", [&] { hipLaunchKernelGGL((symvC<8, 8, 2>), g, dim3(256), 0, 0, At, n, m, vv, rp, cp); }, true);
      time("C 512x256 G=4 occ2", [&] { hipLaunchKernelGGL((symvC<8, 4, 2>), g, dim3(256), 0, 0, At, n, m, vv, rp, cp); }, true);
      time("C 512x256 G=4 occ3", [&] { hipLaunchKernelGGL((symvC<8, 4, 3>), g, dim3(256), 0, 0, At, n, m, vv, rp, cp); }, true);
      time("C 512x256 G=2 occ4", [&] { hipLaunchKernelGGL((symvC<8, 2, 4>), g, dim3(256), 0, 0, At, n, m, vv, rp, cp); }, true); }
    mode = 4;
    { const unsigned nbr = (unsigned)((m + 1023) / 1024);
      const dim3 g(2 * nbr * (nbr + 1));
      time("C 1024x256 G=2 occ2", [&] { hipLaunchKernelGGL((symvC<16, 2, 2>), g, dim3(256), 0, 0, At, n, m, vv, rp, cp); }, true); }
    mode = 1;
    { const unsigned nbr = (unsigned)((m + 255) / 256);
      const dim3 g(nbr * (nbr + 1) / 2);
      time("C 256x256 G=8 occ3", [&] { hipLaunchKernelGGL((symvC<4, 8, 3>), g, dim3(256), 0, 0, At, n, m, vv, rp, cp); }, true);
      time("C 256x256 G=4 occ4", [&] { hipLaunchKernelGGL((symvC<4, 4, 4>), g, dim3(256), 0, 0, At, n, m, vv, rp, cp); }, true); }
    return 0;
}
