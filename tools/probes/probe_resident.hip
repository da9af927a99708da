// probe_resident.hip - which shape of the resident single-pass pipeline (tramp_amd/csrc/ew_resident.h) streams fastest?
// Sweeps (U, DL, DA) for the Gauss-Bernoulli / L1 priors in f64 and f32 at 64 x 1e6, with and without the inter-workgroup
// hand-off (DBG = 1), against a plain two-kernel reference for correctness.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/probes/probe_resident.hip -o tools/probes/build/probe_resident
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../tramp_amd/csrc/ew_resident.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <typename T> __global__ void fill_kernel(T* p, size_t count, double ax) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const double u1 = ((h & 0xFFFFFFFFull) + 0.5) / 4294967296.0, u2 = ((h >> 32) + 0.5) / 4294967296.0;
        const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        const bool on = (h % 20) == 0;                                  // rho = 0.05
        p[i] = T((on ? ax * g * 1.3 : 0.0) + sqrt(ax) * sin(1e3 * u2) * 1.2);
    }
}

// reference: two plain kernels (float atomics for the sum: fine for a tolerance check)
template <typename T, typename Op> __global__ void ref_moments(Op op, size_t n, const T* in0, const T* in1, double* sums) {
    const size_t b = blockIdx.y;
    op.setup(b);
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += (double)op.eval(in0[b * n + i], in1 ? in1[b * n + i] : T(0)).v;
    __shared__ double sm[4];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) atomicAdd(&sums[b], acc);
}
template <typename T, typename Op> __global__ void ref_apply(Op op, size_t n, const T* in0, const T* in1, const double* a_in, const double* sums, T* bnew, double* a_new) {
    const size_t b = blockIdx.y;
    op.setup(b);
    const double an = tmath::a_new_of(sums[b] / (double)n, a_in[b]);
    if (blockIdx.x == 0 && threadIdx.x == 0) a_new[b] = an;
    const T vinv = T(a_in[b] + an);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T x0 = in0[b * n + i], x1 = in1 ? in1[b * n + i] : T(0);
        const T tgt = Op::TARGET == 0 ? x0 : x1;
        bnew[b * n + i] = op.eval(x0, x1).r * vinv - tgt;
    }
}

struct Buffers {
    size_t n, batch;
    void *in0, *in1, *in1_buf, *out, *ref, *bits;
    double *a_in, *a_new, *a_ref, *sums;
    void* ws;
    size_t ws_bytes;
    int num_cu;
};

template <typename T, typename Op, int U, int DL, int DA, int BPC, int DBG, int FL = 0, int NT = 256>
static float run_variant(const Buffers& B, Op op, int reps) {
    constexpr int blocks_per_cu = BPC;
    constexpr int VN = Vec<T>::N;
    ResArgs A{};
    A.n = B.n; A.nv = B.n / VN; A.batch = (unsigned)B.batch;
    A.tiles = (unsigned)((A.nv + (size_t)NT * U - 1) / ((size_t)NT * U));
    A.total = (unsigned long long)A.batch * A.tiles;
    const unsigned P = (unsigned)(blocks_per_cu * B.num_cu);
    const unsigned PS = P - kResCombiners;          // streaming workgroups
    A.adv_q = PS / A.tiles; A.adv_r = PS % A.tiles;
    A.a_in = B.a_in;
    char* ws = (char*)B.ws;
    Op* ops = (Op*)ws; ws += ((B.batch * sizeof(Op) + 255) / 256) * 256;
    A.slots = (unsigned long long*)ws; ws += A.total * 8;
    A.cslots = (unsigned long long*)ws; ws += A.total * 8;
    A.scal = (unsigned long long*)ws;
    A.dummy = ws + ((B.batch * 8 + 255) / 256) * 256;
    A.bits = (const uint8_t*)B.bits; A.bits_stride = ((B.n + 63) / 64) * 8;
    A.out = MsgOut{(FL & 1) ? 0.25 : 0.0, (FL & 1) ? B.a_in : nullptr, (FL & 1) ? B.in1_buf : nullptr, B.a_new, B.out, nullptr, nullptr, nullptr, nullptr};
    if (A.tiles > (unsigned)(PS * (DA - 2) + 1) || false) { printf("   (tiles %u do not fit the window)\n", A.tiles); return -1.f; }
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (ew_resident<T, Op, U, DL, DA, FL, BPC, DBG, NT>), NT, 0));
    if (occ < blocks_per_cu) { printf("   (occupancy %d < %d)\n", occ, blocks_per_cu); return -1.f; }
    auto launch = [&] {
        hipLaunchKernelGGL((ew_res_prepare<Op>), dim3((unsigned)((A.total + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, op, A, ops);
        hipLaunchKernelGGL((ew_resident<T, Op, U, DL, DA, FL, BPC, DBG, NT>), dim3(P), dim3(NT), 0, 0, (const Op*)ops, A, (const T*)B.in0, (const T*)B.in1);
    };
    launch(); launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> times;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); times.push_back(ms);
    }
    CK(hipDeviceSynchronize());
    std::sort(times.begin(), times.end());
    return times[times.size() / 2];
}

template <typename T> static double compare(const Buffers& B, bool check_a) {
    const size_t count = B.n * B.batch;
    std::vector<T> a(count), b(count);
    CK(hipMemcpy(a.data(), B.out, count * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), B.ref, count * sizeof(T), hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < count; ++i) { double d = fabs((double)a[i] - (double)b[i]); if (!(d <= worst)) worst = d; double s = fabs((double)b[i]); if (s > scale) scale = s; }
    double wa = 0;
    if (check_a) {
        std::vector<double> x(B.batch), y(B.batch);
        CK(hipMemcpy(x.data(), B.a_new, B.batch * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), B.a_ref, B.batch * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < B.batch; ++i) { double d = fabs(x[i] - y[i]) / fabs(y[i]); if (!(d <= wa)) wa = d; }
    }
    printf("   max |b_new - ref| / max|ref| = %.3e   max rel a_new = %.3e\n", worst / scale, wa);
    return worst / scale;
}

template <typename T, typename Op> static void reference(const Buffers& B, Op op) {
    CK(hipMemset(B.sums, 0, B.batch * 8));
    hipLaunchKernelGGL((ref_moments<T, Op>), dim3(256, (unsigned)B.batch), dim3(256), 0, 0, op, B.n, (const T*)B.in0, (const T*)B.in1, B.sums);
    hipLaunchKernelGGL((ref_apply<T, Op>), dim3(256, (unsigned)B.batch), dim3(256), 0, 0, op, B.n, (const T*)B.in0, (const T*)B.in1, (const double*)B.a_in, (const double*)B.sums, (T*)B.ref, B.a_ref);
    CK(hipDeviceSynchronize());
}

#define VARIANTN(T, Op, op, U, DL, DA, BPC, DBG, FL, NT, alg)                                                              \
    {                                                                                                        \
        float ms = run_variant<T, Op, U, DL, DA, BPC, DBG, FL, NT>(B, op, 9);                                             \
        if (ms > 0) {                                                                                        \
            printf("%-10s %s NT=%d U=%d DL=%d DA=%d wg/CU=%d flags=%d %s: %8.1f us  %6.0f GB/s algorithmic = %.3f of 8 TB/s\n", #Op, sizeof(T) == 8 ? "f64" : "f32", \
                   NT, U, DL, DA, BPC, FL, DBG ? "no-sync" : "sync   ", ms * 1e3, (alg) / ms / 1e6, (alg) / ms / 1e6 / 8000.0);  \
            fflush(stdout);                                                                                  \
            if (!DBG && !((FL) & 1)) compare<T>(B, true);                                                                   \
        }                                                                                                    \
    }

#define VARIANTF(T, Op, op, U, DL, DA, BPC, DBG, FL, alg) VARIANTN(T, Op, op, U, DL, DA, BPC, DBG, FL, 256, alg)
#define VARIANT(T, Op, op, U, DL, DA, BPC, DBG, alg) VARIANTF(T, Op, op, U, DL, DA, BPC, DBG, 0, alg)

template <typename T> __global__ void pack_kernel(size_t n, size_t stride, const T* y, uint8_t* bits) {
    const size_t b = blockIdx.y, q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= stride) return;
    unsigned byte = 0;
    for (int j = 0; j < 8; ++j) if (q * 8 + j < n && y[b * n + q * 8 + j] > T(0)) byte |= 1u << j;
    bits[b * stride + q] = (uint8_t)byte;
}
template <typename T> __global__ void sign_kernel(T* p, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] > T(0) ? T(1) : T(-1);
}

template <typename T> static void sweep(Buffers& B) {
    const double alg2 = 2.0 * sizeof(T) * B.n * B.batch;
    hipLaunchKernelGGL((fill_kernel<T>), dim3(4096), dim3(256), 0, 0, (T*)B.in0, B.n * B.batch, 1.3);
    hipLaunchKernelGGL((fill_kernel<T>), dim3(4096), dim3(256), 0, 0, (T*)B.in1_buf, B.n * B.batch, 0.7);
    hipLaunchKernelGGL((sign_kernel<T>), dim3(4096), dim3(256), 0, 0, (T*)B.in1_buf, B.n * B.batch);
    const size_t stride = ((B.n + 63) / 64) * 8;
    hipLaunchKernelGGL((pack_kernel<T>), dim3((unsigned)((stride + 255) / 256), (unsigned)B.batch), dim3(256), 0, 0, B.n, stride, (const T*)B.in1_buf, (uint8_t*)B.bits);
    CK(hipDeviceSynchronize());
    B.in1 = nullptr;
    {
        // round 6: the Gauss-Bernoulli prior with r parked in LDS (the library's shape), in registers, and recomputed at
        // the update (kResRecompute: no LDS, b ring only) on two, three and four workgroups per CU
        double a0 = 1.0, b0 = 0.0, eta = 0.5 * log(2 * 3.141592653589793) - log(0.05 / 0.95);
        GbOp<T> op{B.a_in, a0, b0, eta};
        reference<T, GbOp<T>>(B, op);
        using Op = GbOp<T>;
        //                 U DL DA wg/CU nosync flags NT
        VARIANTN(T, Op, op, 4, 1, 3, 2, 0, 16, 256, alg2)
        VARIANTN(T, Op, op, 4, 1, 3, 2, 0, 16 | 32, 256, alg2)
        VARIANTN(T, Op, op, 4, 1, 3, 2, 0, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 4, 1, 3, 3, 0, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 4, 2, 3, 2, 0, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 4, 2, 3, 3, 0, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 2, 1, 3, 4, 0, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 2, 2, 3, 4, 0, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 2, 3, 3, 4, 0, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 4, 1, 3, 3, 0, 128, 256, alg2)
        VARIANTN(T, Op, op, 4, 1, 3, 3, 1, 16 | 128, 256, alg2)
        VARIANTN(T, Op, op, 2, 2, 3, 4, 1, 16 | 128, 256, alg2)
    }
}

int main(int argc, char** argv) {
    Buffers B{};
    B.batch = argc > 1 ? atoi(argv[1]) : 64;
    B.n = argc > 2 ? atol(argv[2]) : 1000000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    B.num_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs; %zu instances x %zu\n", prop.name, B.num_cu, B.batch, B.n);
    const size_t bytes = B.n * B.batch * 8;
    CK(hipMalloc(&B.in0, bytes)); CK(hipMalloc(&B.out, bytes)); CK(hipMalloc(&B.ref, bytes));
    B.in1 = nullptr; CK(hipMalloc(&B.in1_buf, bytes)); CK(hipMalloc(&B.bits, bytes / 32 + 4096));
    CK(hipMalloc(&B.a_in, B.batch * 8)); CK(hipMalloc(&B.a_new, B.batch * 8)); CK(hipMalloc(&B.a_ref, B.batch * 8)); CK(hipMalloc(&B.sums, B.batch * 8));
    B.ws_bytes = 64 << 20; CK(hipMalloc(&B.ws, B.ws_bytes));
    std::vector<double> a(B.batch);
    for (size_t i = 0; i < B.batch; ++i) a[i] = 1.3 + 0.01 * i;
    CK(hipMemcpy(B.a_in, a.data(), B.batch * 8, hipMemcpyHostToDevice));
    sweep<double>(B);
    sweep<float>(B);
    return 0;
}
