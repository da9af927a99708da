// probe_flat_resident.hip - the single-pass update WITHOUT a persistent grid: one short-lived workgroup per tile, launched in
// stream order; it loads its tile, takes the moments, publishes its block sum as a granule, WAITS (registers held) for its
// instance's scalar and writes the update.  The hardware's multithreading (4-5 workgroups per CU) does what the software
// pipeline of ew_resident.h does by hand; the workgroup that owns the LAST tile of an instance folds the slots.
// Progress needs the tiles of the oldest unfinished instance to be resident or finished whenever anybody waits: true when
// workgroups start in blockIdx order and the chip holds more workgroups than an instance has tiles - the first is what the
// hardware does, not what HIP promises (every spin is bounded: a violation traps, it does not hang or corrupt).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/probes/probe_flat_resident.hip -o tools/probes/build/probe_flat_resident
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "../../tramp_amd/csrc/ew_resident.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <typename T> __global__ void fill_kernel(T* p, size_t count, double ax) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const double u1 = ((h & 0xFFFFFFFFull) + 0.5) / 4294967296.0, u2 = ((h >> 32) + 0.5) / 4294967296.0;
        const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        const bool on = (h % 20) == 0;
        p[i] = T((on ? ax * g * 1.3 : 0.0) + sqrt(ax) * sin(1e3 * u2) * 1.2);
    }
}

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

// MODE 0: the update; 1: no hand-off (constant scalar): the stream alone
// RLDS: the posterior means wait in LDS (fewer registers per waiting workgroup: more of them per CU);
// COMB: who folds an instance's slots: 0 the owner of its last tile, 1 the owner of its first tile, 2 dedicated blocks
template <typename T, typename Op, int U, int MODE, int WPE, bool RLDS = false, int COMB = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void ew_flat(const Op* __restrict__ ops, ResArgs A, const T* __restrict__ in0, const T* __restrict__ in1) {
    using V = typename Vec<T>::type;
    using NV = typename NativeVec<T>::type;
    constexpr int VN = Vec<T>::N;
    constexpr unsigned TV = kBlock * U;
    __shared__ double smem[kBlock / kWave];
    __shared__ typename Vec<T>::type r_lds[RLDS ? U : 1][kBlock];
    const unsigned tid = threadIdx.x;
    if (COMB == 2 && blockIdx.x < kResCombiners) {
        if (MODE == 0 && tid < kWave) res_combiner<false>(A, blockIdx.x);
        return;
    }
    Op::stage();
    const unsigned bid = blockIdx.x - (COMB == 2 ? kResCombiners : 0);
    const unsigned inst = bid / A.tiles, tile = bid % A.tiles;
    const unsigned nv = (unsigned)A.nv;
    const V* p0 = reinterpret_cast<const V*>(in0) + (size_t)inst * nv;
    const V* p1 = reinterpret_cast<const V*>(in1) + (size_t)inst * nv;
    V* pb = reinterpret_cast<V*>(A.out.b_new) + (size_t)inst * nv;
    const unsigned first = tile * TV;
    const unsigned off = min(first, nv - TV) + tid;
    V x0[U], x1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        x0[u] = p0[off + u * kBlock];
        if (Op::NIN > 1) x1[u] = p1[off + u * kBlock];
    }
    const Op op = ops[inst];
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool counts = off + u * kBlock >= first;
        V vr;
#pragma unroll
        for (int c = 0; c < VN; ++c) {
            const T a0 = vget<T>(x0[u], c);
            const T a1 = (Op::NIN > 1) ? vget<T>(x1[u], c) : a0;
            EwOut<T> o = op.eval(a0, a1);
            double vi = (double)o.v;
            if (not_finite(o.r)) vi = __longlong_as_double(0x7ff8000000000000LL);
            acc += counts ? vi : 0.0;
            vset<T>(vr, c, o.r);
        }
        if (Op::TARGET != 0) x0[u] = x1[u];
        if (RLDS) r_lds[u][tid] = vr;
        else x1[u] = vr;
    }
    unsigned long long sbits = (unsigned long long)__double_as_longlong(1.5);
    if (MODE == 0) {
        acc = block_sum(acc, smem);
        if (tid == 0) {
            unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
            if (ab == kResSentinel) ab = 0x7ff8000000000000ull;
            res_store(A.slots + (size_t)inst * A.tiles + tile, ab);
        }
        if (COMB != 2 && tile == (COMB == 1 ? 0u : A.tiles - 1) && tid < kWave) {
            // the owner of the instance's last tile folds the slots (fixed order) and publishes the scalar
            const unsigned long long* sl = A.slots + (size_t)inst * A.tiles;
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            double s;
            for (;;) {
                bool ok = true;
                s = 0.0;
                for (unsigned q0 = 0; q0 < A.tiles; q0 += 64u * kResQ) {
                    unsigned long long bits[kResQ];
#pragma unroll
                    for (int q = 0; q < kResQ; ++q) {
                        const unsigned i = q0 + tid + 64u * q;
                        bits[q] = (i < A.tiles) ? res_load(sl + i) : 0ull;
                    }
#pragma unroll
                    for (int q = 0; q < kResQ; ++q) {
                        ok = ok && bits[q] != kResSentinel;
                        s += __longlong_as_double((long long)bits[q]);
                    }
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(2);
                res_spin_check(t0);
            }
            s = wave_sum(s);
            if (tid == 0) {
                const double v = s / (double)A.n;
                const double a_in = A.a_in[inst];
                const double a_new = tmath::a_new_of(v, a_in);
                A.out.a_new[inst] = a_new;
                unsigned long long sb = (unsigned long long)__double_as_longlong(a_in + a_new);
                if (sb == kResSentinel) sb = 0x7ff8000000000000ull;
                res_store(A.scal + inst, sb);
            }
        }
        sbits = res_load(A.scal + inst);
        if (sbits == kResSentinel) {
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            do {
                __builtin_amdgcn_s_sleep(4);
                res_spin_check(t0);
                sbits = res_load(A.scal + inst);
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
            } while (sbits == kResSentinel);
        }
    }
    const T vinv = T(__longlong_as_double((long long)sbits));
#pragma unroll
    for (int u = 0; u < U; ++u) {
        V vo;
        const V vr = RLDS ? r_lds[u][tid] : x1[u];
#pragma unroll
        for (int c = 0; c < VN; ++c) vset<T>(vo, c, vget<T>(vr, c) * vinv - vget<T>(x0[u], c));
        V* dst = (off + u * kBlock >= first) ? &pb[off + u * kBlock] : reinterpret_cast<V*>(A.dummy) + tid + u * kBlock;
        __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&vo), reinterpret_cast<NV*>(dst));
    }
}

struct Buffers {
    size_t n, batch;
    void *in0, *in1, *in1_buf, *out, *ref;
    double *a_in, *a_new, *a_ref, *sums;
    void* ws;
    int num_cu;
};

template <typename T, typename Op, int U, int MODE, int WPE, bool RLDS = false, int COMB = 0>
static float run_variant(const Buffers& B, Op op, int reps) {
    constexpr int VN = Vec<T>::N;
    ResArgs A{};
    A.n = B.n; A.nv = B.n / VN; A.batch = (unsigned)B.batch;
    A.tiles = (unsigned)((A.nv + (size_t)kBlock * U - 1) / ((size_t)kBlock * U));
    A.total = (unsigned long long)A.batch * A.tiles;
    A.a_in = B.a_in;
    char* ws = (char*)B.ws;
    Op* ops = (Op*)ws; ws += ((B.batch * sizeof(Op) + 255) / 256) * 256;
    A.slots = (unsigned long long*)ws; ws += A.total * 8;
    A.cslots = nullptr;
    A.scal = (unsigned long long*)ws; ws += ((B.batch * 8 + 255) / 256) * 256;
    A.dummy = ws;
    A.out = MsgOut{0.0, nullptr, nullptr, B.a_new, B.out, nullptr, nullptr, nullptr, nullptr};
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (ew_flat<T, Op, U, MODE, WPE, RLDS, COMB>), kBlock, 0));
    printf("   [occupancy %d workgroups per CU, %u tiles per instance, window %d workgroups]\n", occ, A.tiles, occ * B.num_cu);
    if ((unsigned)(occ * B.num_cu) < A.tiles + 64) { printf("   (window too small)\n"); return -1.f; }
    auto launch = [&] {
        hipLaunchKernelGGL((ew_res_prepare<Op>), dim3((unsigned)((A.total + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, op, A, ops);
        hipLaunchKernelGGL((ew_flat<T, Op, U, MODE, WPE, RLDS, COMB>), dim3((unsigned)A.total + (COMB == 2 ? kResCombiners : 0)), dim3(kBlock), 0, 0, (const Op*)ops, A, (const T*)B.in0, (const T*)B.in1);
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

template <typename T> static void compare(const Buffers& B) {
    const size_t count = B.n * B.batch;
    std::vector<T> a(count), b(count);
    CK(hipMemcpy(a.data(), B.out, count * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), B.ref, count * sizeof(T), hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < count; ++i) { double d = fabs((double)a[i] - (double)b[i]); if (!(d <= worst)) worst = d; double s = fabs((double)b[i]); if (s > scale) scale = s; }
    std::vector<double> x(B.batch), y(B.batch);
    CK(hipMemcpy(x.data(), B.a_new, B.batch * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y.data(), B.a_ref, B.batch * 8, hipMemcpyDeviceToHost));
    double wa = 0;
    for (size_t i = 0; i < B.batch; ++i) { double d = fabs(x[i] - y[i]) / fabs(y[i]); if (!(d <= wa)) wa = d; }
    printf("   max |b_new - ref| / max|ref| = %.3e   max rel a_new = %.3e\n", worst / scale, wa);
}

template <typename T, typename Op> static void reference(const Buffers& B, Op op) {
    CK(hipMemset(B.sums, 0, B.batch * 8));
    hipLaunchKernelGGL((ref_moments<T, Op>), dim3(256, (unsigned)B.batch), dim3(256), 0, 0, op, B.n, (const T*)B.in0, (const T*)B.in1, B.sums);
    hipLaunchKernelGGL((ref_apply<T, Op>), dim3(256, (unsigned)B.batch), dim3(256), 0, 0, op, B.n, (const T*)B.in0, (const T*)B.in1, (const double*)B.a_in, (const double*)B.sums, (T*)B.ref, B.a_ref);
    CK(hipDeviceSynchronize());
}

#define VARIANTX(T, Op, op, U, MODE, WPE, RLDS, COMB, alg)                                                                       \
    {                                                                                                               \
        float ms = run_variant<T, Op, U, MODE, WPE, RLDS, COMB>(B, op, 9);                                                      \
        if (ms > 0) {                                                                                               \
            printf("%-22s %s U=%d waves/SIMD=%d rlds=%d comb=%d %s: %8.1f us  %6.0f GB/s algorithmic = %.3f of 8 TB/s\n", #Op, sizeof(T) == 8 ? "f64" : "f32", U, WPE, (int)RLDS, COMB, \
                   MODE ? "no-sync" : "sync   ", ms * 1e3, (alg) / ms / 1e6, (alg) / ms / 1e6 / 8000.0);            \
            fflush(stdout);                                                                                         \
            if (!MODE) compare<T>(B);                                                                               \
        }                                                                                                           \
    }

#define VARIANT(T, Op, op, U, MODE, WPE, alg) VARIANTX(T, Op, op, U, MODE, WPE, false, 0, alg)

template <typename T> static void sweep(Buffers& B) {
    const double alg2 = 2.0 * sizeof(T) * B.n * B.batch, alg3 = 1.5 * alg2;
    hipLaunchKernelGGL((fill_kernel<T>), dim3(4096), dim3(256), 0, 0, (T*)B.in0, B.n * B.batch, 1.3);
    hipLaunchKernelGGL((fill_kernel<T>), dim3(4096), dim3(256), 0, 0, (T*)B.in1_buf, B.n * B.batch, 0.7);
    CK(hipDeviceSynchronize());
    B.in1 = nullptr;
    {
        using Op = GbOp<T>;
        Op op{};
        op.ax = B.a_in; op.a0 = 1.0; op.b0 = 0.0; op.eta = 0.5 * log(2 * M_PI) - log(0.05 / 0.95);
        reference<T, Op>(B, op);
        VARIANT(T, Op, op, 4, 0, 6, alg2)
        VARIANTX(T, Op, op, 4, 0, 6, false, 1, alg2)
        VARIANTX(T, Op, op, 4, 0, 6, false, 2, alg2)
        VARIANTX(T, Op, op, 4, 0, 8, true, 2, alg2)
        VARIANTX(T, Op, op, 4, 0, 7, true, 2, alg2)
        VARIANTX(T, Op, op, 8, 0, 4, false, 2, alg2)
        VARIANTX(T, Op, op, 8, 0, 5, true, 2, alg2)
        VARIANTX(T, Op, op, 8, 0, 4, true, 2, alg2)
        VARIANTX(T, Op, op, 2, 0, 8, true, 2, alg2)
    }
    {
        using Op = L1Op<T>;
        Op op{};
        op.ax = B.a_in; op.gamma = 1.0;
        reference<T, Op>(B, op);
        VARIANTX(T, Op, op, 4, 0, 8, true, 2, alg2)
    }
    B.in1 = B.in1_buf;
    {
        using Op = AbsLikOp<T>;
        Op op{};
        op.az = B.a_in;
        reference<T, Op>(B, op);
        VARIANTX(T, Op, op, 4, 0, 6, true, 2, alg3)
        VARIANTX(T, Op, op, 4, 0, 5, false, 2, alg3)
    }
    {
        using Op = PwlOp<T, 2, false, true>;
        Op op{};
        op.az = B.a_in; op.ax = B.a_in;
        reference<T, Op>(B, op);
        VARIANTX(T, Op, op, 4, 0, 3, true, 2, alg3)
        VARIANTX(T, Op, op, 2, 0, 4, true, 2, alg3)
    }
    {
        using Op = PwlOp<T, 0, true, true>;
        Op op{};
        op.az = B.a_in; op.ax = B.a_in;
        reference<T, Op>(B, op);
        VARIANT(T, Op, op, 4, 0, 3, alg3)
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
    CK(hipMalloc(&B.in0, bytes)); CK(hipMalloc(&B.in1_buf, bytes)); CK(hipMalloc(&B.out, bytes)); CK(hipMalloc(&B.ref, bytes));
    CK(hipMalloc(&B.a_in, B.batch * 8)); CK(hipMalloc(&B.a_new, B.batch * 8)); CK(hipMalloc(&B.a_ref, B.batch * 8)); CK(hipMalloc(&B.sums, B.batch * 8));
    CK(hipMalloc(&B.ws, 64 << 20));
    std::vector<double> a(B.batch);
    for (size_t i = 0; i < B.batch; ++i) a[i] = 1.3 + 0.01 * i;
    CK(hipMemcpy(B.a_in, a.data(), B.batch * 8, hipMemcpyHostToDevice));
    sweep<double>(B);
    sweep<float>(B);
    return 0;
}
