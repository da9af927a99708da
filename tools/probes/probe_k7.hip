// probe_k7.hip - what bounds the moments pass of the two-region channels (K7, Relu backward) at 64 x 1e6?
// One streaming kernel shaped like ew_moments_tile (tile of 256 x U 16-byte pieces, r parked in a scratch, block sum of v),
// with the evaluation swapped by MODE:
//   0  pwl_posterior_fast<T, 2, false, true>   the library's form (moment form; erfcx from ONE table value per evaluation + the
//      derivative recurrence erfcx' = 2 x erfcx - 2/sqrt(pi): 1 LDS read and ~30 VALU)
//   2  erfcx replaced by a few FMAs (no table at all; wrong numbers): the cost of everything that is not erfcx
//   3  the moment form with erfcx_tay called without the wave votes and the fall-back path (inputs stay on the grid)
//   4  no evaluation at all (r = bz + bx, v = bz): the stream
// (The first run of this probe, profiles/r05/probe_k7_moments_pass_what_bounds_it.txt, had round 3's interval table - eight LDS
//  reads per erfcx - as mode 0, the same with one table row per wave as mode 1, and the one-value table as mode 3: 427 / 345 /
//  358 us against 299 us for the bare stream.)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/probes/probe_k7.hip -o tools/probes/build/probe_k7
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../tramp_amd/csrc/ew_ops.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <typename T> __global__ void fill_kernel(T* p, size_t count, double scale, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + seed) * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const double u1 = ((h & 0xFFFFFFFFull) + 0.5) / 4294967296.0, u2 = ((h >> 32) + 0.5) / 4294967296.0;
        p[i] = T(scale * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    }
}

template <int MODE> __device__ __forceinline__ double erfcx_mode(double x) {
    if (MODE == 2) return ::fma(x, x, 1.0);
    return tmath::erfcx_tay(x);
}

// the moment form of tramp_math.h for Relu backward, with the erfcx of MODE (fast path only: the probe's inputs stay on the grid)
template <int MODE> __device__ __forceinline__ void relu_bwd_eval(double bz, double bx, const tmath::PwlConst<double>& c, double& r, double& v) {
    if (MODE == 0) {
        tmath::PwlOut<double> o = tmath::pwl_posterior_fast<double, 2, false, true>(bz, bx, c);
        r = o.r; v = o.v;
        return;
    }
    if (MODE == 4) { r = bz + bx; v = bz; return; }
    const double b0 = bz + bx, b1 = bz;
    const double x0 = b0 * c.kx[0], x1 = b1 * c.kx[1];
    const double e0 = erfcx_mode<MODE>(x0), e1 = erfcx_mode<MODE>(x1);
    const double r00 = b0 * c.reg[0].v0, r01 = b1 * c.reg[1].v0;
    const double W0 = e0 * c.reg[0].s0, W1 = e1 * c.reg[1].s0;
    const double S = W0 + W1, iS = tmath::rcp_pos(S);
    const double cc = (W0 >= W1) ? r00 : r01;
    const double d0 = r00 - cc, d1 = r01 - cc;
    const double n1 = ::fma(W0, d0, ::fma(W1, d1, c.gv[0] + c.gv[1]));
    const double q0 = ::fma(d0, d0, c.reg[0].v0), q1 = ::fma(d1, d1, c.reg[1].v0);
    const double t0 = ::fma(-2.0, cc, r00), t1 = ::fma(-2.0, cc, r01);
    const double n2 = ::fma(W0, q0, ::fma(W1, q1, ::fma(c.gv[0], t0, c.gv[1] * t1)));
    const double m1 = n1 * iS;
    r = cc + m1;
    v = ::fma(-m1, m1, n2 * iS);
}

template <int MODE, int U>
__global__ __launch_bounds__(256) void k7_moments(size_t n, int tpb, const double* __restrict__ bz, const double* __restrict__ bx,
                                                  double* __restrict__ rbuf, double* __restrict__ partials, tmath::PwlConst<double> c) {
    __shared__ double sm[4];
#if defined(__HIP_DEVICE_COMPILE__)
    if (MODE == 0 || MODE == 3) tmath::erfcx_lds_init<true>();
#endif
    const size_t b = blockIdx.y, nv = n / 2;
    const double2* p0 = reinterpret_cast<const double2*>(bz + b * n);
    const double2* p1 = reinterpret_cast<const double2*>(bx + b * n);
    double2* pr = reinterpret_cast<double2*>(rbuf + b * n);
    double acc = 0.0;
    for (int tile = 0; tile < tpb; ++tile) {
        const size_t base = ((size_t)blockIdx.x * tpb + tile) * 256 * U + threadIdx.x;
        if (base - threadIdx.x >= nv) break;
        double2 v0[U], v1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < nv) { v0[u] = p0[i]; v1[u] = p1[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < nv) {
                double2 ro;
                double vv;
                relu_bwd_eval<MODE>(v0[u].x, v1[u].x, c, ro.x, vv); acc += vv;
                relu_bwd_eval<MODE>(v0[u].y, v1[u].y, c, ro.y, vv); acc += vv;
                pr[i] = ro;
            }
        }
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = acc;
}

// ---- design D: persistent workgroups, r parked in an LDS FIFO of D tiles, the target message read a second time at the
// update (no register ring, no unrolling; the hardware's multithreading hides the latency).  The hand-off is left out
// (constant scalar): the ceiling of the stream + evaluation in this shape.  NT threads, one 16-byte piece per thread and tile.
template <int NT, int D, int PF>
__global__ __launch_bounds__(NT) void k7_fifo(size_t n, size_t batch, const double* __restrict__ bz, const double* __restrict__ bx,
                                              double* __restrict__ bnew, double* __restrict__ partials, tmath::PwlConst<double> c) {
    __shared__ double sm[NT / 64];
    __shared__ double2 fifo[D][NT];
#if defined(__HIP_DEVICE_COMPILE__)
    tmath::erfcx_lds_init<true>();
#endif
    const size_t nv = n / 2, tiles = (nv + NT - 1) / NT, total = tiles * batch;
    const unsigned P = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const size_t Kw = total > wg ? (total - wg + P - 1) / P : 0;
    const double2* p0 = reinterpret_cast<const double2*>(bz);
    const double2* p1 = reinterpret_cast<const double2*>(bx);
    double2* po = reinterpret_cast<double2*>(bnew);
    auto addr = [&](size_t k) {            // piece index of this thread in its k-th tile (tiles instance-major; last tile of an instance overlaps)
        const size_t t = wg + k * P, inst = t / tiles, tl = t % tiles;
        const size_t off = (tl * NT + NT <= nv) ? tl * NT : nv - NT;
        return inst * nv + off + tid;
    };
    double2 a0, a1;
    if (PF && Kw) { const size_t i = addr(0); a0 = p0[i]; a1 = p1[i]; }
    double accs = 0.0;
    for (size_t j = 0; j < Kw + D - 1; ++j) {
        double2 told;
        const bool upd = j >= (size_t)(D - 1);
        size_t iu = 0;
        if (upd) { iu = addr(j - (D - 1)); told = p0[iu]; }
        if (j < Kw) {
            double2 x0, x1;
            if (PF) {
                x0 = a0; x1 = a1;
                if (j + 1 < Kw) { const size_t i = addr(j + 1); a0 = p0[i]; a1 = p1[i]; }
            } else {
                const size_t i = addr(j);
                x0 = p0[i]; x1 = p1[i];
            }
            double2 r;
            double v0, v1;
            relu_bwd_eval<0>(x0.x, x1.x, c, r.x, v0);
            relu_bwd_eval<0>(x0.y, x1.y, c, r.y, v1);
            fifo[j % D][tid] = r;
            double acc = block_sum(v0 + v1, sm);
            if (tid == 0) accs += acc;
        }
        if (upd) {
            const double2 r = fifo[(j - (D - 1)) % D][tid];
            double2 o;
            o.x = r.x * 1.5 - told.x;
            o.y = r.y * 1.5 - told.y;
            __builtin_nontemporal_store(o.x, &po[iu].x);
            __builtin_nontemporal_store(o.y, &po[iu].y);
        }
    }
    if (tid == 0) partials[wg] = accs;
}

template <int NT, int D, int PF> static void run_fifo(int bpc, size_t n, size_t batch, const double* bz, const double* bx, double* out, double* part,
                                                      const tmath::PwlConst<double>& c) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const unsigned P = (unsigned)(bpc * prop.multiProcessorCount);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k7_fifo<NT, D, PF>), dim3(P), dim3(NT), 0, 0, n, batch, bz, bx, out, part, c);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k7_fifo<NT, D, PF>), dim3(P), dim3(NT), 0, 0, n, batch, bz, bx, out, part, c);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k7_fifo<NT, D, PF>)));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k7_fifo<NT, D, PF>, NT, 0));
    printf("D fifo NT=%d D=%d prefetch=%d wg/CU=%d (occupancy %d)  %8.1f us  %5.0f GB/s algorithmic (3T)  regs %d lds %zu\n", NT, D, PF, bpc, occ,
           ms / reps * 1e3, 3.0 * 8 * n * batch / (ms / reps) / 1e6, fa.numRegs, (size_t)fa.sharedSizeBytes);
}

template <int MODE, int U> static void run(const char* name, size_t n, size_t batch, const double* bz, const double* bx, double* r, double* part,
                                           const tmath::PwlConst<double>& c, std::vector<double>* sums) {
    const size_t nv = n / 2, tiles = (nv + 256 * U - 1) / (256 * U);
    int tpb = 8;
    while (tpb > 1 && (tiles / tpb) * batch < 4096) tpb >>= 1;
    const unsigned ntile = (unsigned)((tiles + tpb - 1) / tpb);
    dim3 grid(ntile, (unsigned)batch);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k7_moments<MODE, U>), grid, dim3(256), 0, 0, n, tpb, bz, bx, r, part, c);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k7_moments<MODE, U>), grid, dim3(256), 0, 0, n, tpb, bz, bx, r, part, c);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> h((size_t)batch * ntile);
    CK(hipMemcpy(h.data(), part, h.size() * 8, hipMemcpyDeviceToHost));
    double s0 = 0;
    for (unsigned i = 0; i < ntile; ++i) s0 += h[i];
    if (sums) sums->push_back(s0 / n);
    int regs = 0;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k7_moments<MODE, U>)) == hipSuccess) regs = fa.numRegs;
    printf("%-34s U=%d  %8.1f us   %5.0f GB/s moved (3T)   mean v[0] = %.15g   regs %d  lds %zu\n", name, U, ms / reps * 1e3,
           3.0 * 8 * n * batch / (ms / reps) / 1e6, s0 / n, regs, (size_t)fa.sharedSizeBytes);
}

int main(int argc, char** argv) {
    const size_t batch = argc > 1 ? atol(argv[1]) : 64, n = argc > 2 ? atol(argv[2]) : 1000000;
    double *bz, *bx, *r, *part;
    CK(hipMalloc(&bz, batch * n * 8)); CK(hipMalloc(&bx, batch * n * 8)); CK(hipMalloc(&r, batch * n * 8));
    CK(hipMalloc(&part, batch * 4096 * 8));
    hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, bz, batch * n, 1.5, 1ull);
    hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, bx, batch * n, 1.5, 77777777ull);
    const tmath::PwlConst<double> c = tmath::pwl_const<double, 2>(1.3, 0.7);
    std::vector<double> sums;
    run<0, 2>("0 library", n, batch, bz, bx, r, part, c, &sums);
    run<2, 2>("2 erfcx := 1 + x^2", n, batch, bz, bx, r, part, c, nullptr);
    run<3, 2>("3 moment form, no votes", n, batch, bz, bx, r, part, c, &sums);
    run<4, 2>("4 stream only", n, batch, bz, bx, r, part, c, nullptr);
    run<0, 1>("0 library", n, batch, bz, bx, r, part, c, nullptr);
    run<3, 1>("3 moment form, no votes", n, batch, bz, bx, r, part, c, nullptr);
    run<0, 4>("0 library", n, batch, bz, bx, r, part, c, nullptr);
    run<3, 4>("3 moment form, no votes", n, batch, bz, bx, r, part, c, nullptr);
    run_fifo<512, 7, 0>(2, n, batch, bz, bx, r, part, c);
    run_fifo<512, 7, 1>(2, n, batch, bz, bx, r, part, c);
    run_fifo<256, 7, 1>(4, n, batch, bz, bx, r, part, c);
    run_fifo<256, 4, 1>(6, n, batch, bz, bx, r, part, c);
    run_fifo<1024, 7, 1>(1, n, batch, bz, bx, r, part, c);
    run_fifo<1024, 7, 0>(1, n, batch, bz, bx, r, part, c);
    printf("mean v, no-vote form against library: rel. diff %.3e\n", fabs(sums[1] - sums[0]) / fabs(sums[0]));
    return 0;
}
