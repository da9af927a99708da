// ew_kernels.hip - componentwise factor updates (K1-K7, K10-K12) for gfx950.
//
// Every isotropic factor update is "posterior moments (r_i, v_i) per component,
// v = mean_i v_i, then the message update with the scalar a_new(v)"
// (tramp/base.py:250-255).  b_new depends on the global mean, so each update is
// two streaming passes over the instance:
//   pass 1  read inputs, evaluate (r_i, v_i) ONCE, park r_i in a chunk scratch, reduce
//           sum(v_i) per block                            -> partials[chunk, nblk]
//   pass 2  (one generic kernel for all factors) folds the partials in fixed order,
//           forms a_new and writes b_new = r (a + a_new) - b (+ damping, optional r)
// Two realisations of pass 2, chosen per factor (measured on MI355X, 64 x 1e6 f64):
//   light denoisers (GaussBernoulli, L1): pass 2 RECOMPUTES r_i from the inputs - 3T bytes
//     moved for 2T algorithmic; the L1 update then runs at the box's read+write stream
//     limit (~5 TB/s actual), GaussBernoulli ~25 % above it (two exp per component);
//   heavy factors (erfcx / log / tanh: Sgn/Abs likelihoods, Sgn/Abs/Relu channels): pass 1
//     parks r_i in a scratch and pass 2 (one generic kernel) reads it back - the
//     transcendental math runs once; these are f64-VALU-bound, not HBM-bound.
// Chunking instances so that the re-reads hit the 256 MiB Infinity Cache was measured and
// did NOT pay (measured in round 2; the knob is gone).
// 16-byte vector loads/stores, 64-wide shuffles, one LDS hop per block; no atomics:
// results are run-to-run reproducible.
//
// Built with -ffp-contract=off so that f64 branch masks see the same operands as
// the NumPy reference (no FMA contraction).
#include "common.h"
#include "prior_hint.h"
#include "tramp_math.h"

#include "ew_ops.h"
#include "ew_resident_host.h"


// Reads the inputs once, evaluates (r_i, v_i), stores r_i to the chunk scratch and reduces
// sum(v_i) per block.  `inst0` is the first instance of the chunk (blockIdx.y is relative).
template <typename T, typename Op, bool VEC>
__global__ __launch_bounds__(kBlock) void ew_pass1(Op op, size_t n, size_t inst0, const T* __restrict__ in0,
                                                   const T* __restrict__ in1, T* __restrict__ rbuf,
                                                   double* __restrict__ partials,
                                                   unsigned long long* __restrict__ pcount,
                                                   uint8_t* __restrict__ mask_out) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    __shared__ double smem[kBlock / kWave];
    __shared__ unsigned long long smem_u[kBlock / kWave];
    const size_t b = inst0 + blockIdx.y;
    op.setup(b);
    const T* p0 = in0 + b * n;
    const T* p1 = (Op::NIN > 1) ? in1 + b * n : nullptr;
    T* pr = rbuf ? rbuf + (size_t)blockIdx.y * n : nullptr;
    uint8_t* mo = mask_out ? mask_out + b * n : nullptr;
    double acc = 0.0;
    unsigned long long cnt = 0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    if (VEC) {
        const size_t nv = n / VN;
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += stride) {
            V v0 = reinterpret_cast<const V*>(p0)[i];
            V v1 = v0;
            if (Op::NIN > 1) v1 = reinterpret_cast<const V*>(p1)[i];
            V vr;
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                EwOut<T> o = op.eval(vget<T>(v0, j), vget<T>(v1, j));
                acc += (double)o.v;
                if (not_finite(o.r)) acc += __longlong_as_double(0x7ff8000000000000LL);
                if (OpCounts<Op>::value) cnt += o.mask & 1u;
                vset<T>(vr, j, o.r);
                if (mo) mo[i * VN + j] = (uint8_t)o.mask;
            }
            if (pr) reinterpret_cast<V*>(pr)[i] = vr;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            T x0 = p0[i];
            T x1 = (Op::NIN > 1) ? p1[i] : x0;
            EwOut<T> o = op.eval(x0, x1);
            acc += (double)o.v;
            if (not_finite(o.r)) acc += __longlong_as_double(0x7ff8000000000000LL);
            if (OpCounts<Op>::value) cnt += o.mask & 1u;
            if (pr) pr[i] = o.r;
            if (mo) mo[i] = (uint8_t)o.mask;
        }
    }
    acc = block_sum(acc, smem);
    if (pcount) cnt = block_sum_u64(cnt, smem_u);
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
        if (pcount) pcount[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = cnt;
    }
}

// ---------------------------------------------------------------- pass 2: message update

// Generic for every factor: folds the partials (same order in every block), forms
// a_new = clip(1/v - a) and writes b_new = r (a + a_new) - b (tramp/base.py:250-255) with the
// optional damping; r comes from the chunk scratch (cache resident), b from the input message.
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void ew_pass2(size_t n, size_t inst0, const double* __restrict__ a_in_d,
                                                   const T* __restrict__ b_in, const T* __restrict__ rbuf,
                                                   const double* __restrict__ partials,
                                                   const unsigned long long* __restrict__ pcount, int nblk1,
                                                   MsgOut out) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    __shared__ double smem[kBlock / kWave];
    __shared__ unsigned long long smem_u[kBlock / kWave];
    const size_t bl = blockIdx.y, b = inst0 + bl;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk1; i += kBlock) s += partials[bl * nblk1 + i];
    s = block_sum(s, smem);
    const double v = s / (double)n;
    const double a_in = a_in_d[b];
    const double a_new = tmath::a_new_of(v, a_in);
    const double d = out.damping;
    if (blockIdx.x == 0) {
        if (out.count) {
            unsigned long long c = 0;
            for (int i = threadIdx.x; i < nblk1; i += kBlock) c += pcount[bl * nblk1 + i];
            c = block_sum_u64(c, smem_u);
            if (threadIdx.x == 0) out.count[b] = c;
        }
        if (threadIdx.x == 0) {
            double a_out = a_new;
            if (d != 0.0) a_out = d * out.a_old[b] + (1.0 - d) * a_new;
            out.a_new[b] = a_out;
            if (out.v) out.v[b] = v;
        }
    }
    const T vinv = T(a_in + a_new);
    const T dT = T(d), omd = T(1.0 - d);
    const T* pb = b_in + b * n;
    const T* pr = rbuf + bl * n;
    T* bo = reinterpret_cast<T*>(out.b_new) + b * n;
    const T* bold = out.b_old ? reinterpret_cast<const T*>(out.b_old) + b * n : nullptr;
    T* ro = out.r ? reinterpret_cast<T*>(out.r) + b * n : nullptr;
    const size_t stride = (size_t)gridDim.x * kBlock;
    if (VEC) {
        const size_t nv = n / VN;
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += stride) {
            const V vb = reinterpret_cast<const V*>(pb)[i];
            const V vr = reinterpret_cast<const V*>(pr)[i];
            V vo, vold;
            if (bold) vold = reinterpret_cast<const V*>(bold)[i];
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                T bn = vget<T>(vr, j) * vinv - vget<T>(vb, j);
                if (bold) bn = dT * vget<T>(vold, j) + omd * bn;
                vset<T>(vo, j, bn);
            }
            reinterpret_cast<V*>(bo)[i] = vo;
            if (ro) reinterpret_cast<V*>(ro)[i] = vr;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            const T ri = pr[i];
            T bn = ri * vinv - pb[i];
            if (bold) bn = dT * bold[i] + omd * bn;
            bo[i] = bn;
            if (ro) ro[i] = ri;
        }
    }
}

// Variant of pass 2 that recomputes r_i from the inputs instead of reading the scratch
// (cheaper for the light denoisers, whose arithmetic hides under the stream).
template <typename T, typename Op, bool VEC>
__global__ __launch_bounds__(kBlock) void ew_pass2_recompute(Op op, size_t n, size_t inst0,
                                                             const T* __restrict__ in0, const T* __restrict__ in1,
                                                             const double* __restrict__ partials,
                                                             const unsigned long long* __restrict__ pcount,
                                                             int nblk1, MsgOut out) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    __shared__ double smem[kBlock / kWave];
    __shared__ unsigned long long smem_u[kBlock / kWave];
    const size_t bl = blockIdx.y, b = inst0 + bl;
    op.setup(b);
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk1; i += kBlock) s += partials[bl * nblk1 + i];
    s = block_sum(s, smem);
    const double v = s / (double)n;
    const double a_in = op.a_msg;
    const double a_new = tmath::a_new_of(v, a_in);
    const double d = out.damping;
    if (blockIdx.x == 0) {
        if (out.count) {
            unsigned long long c = 0;
            for (int i = threadIdx.x; i < nblk1; i += kBlock) c += pcount[bl * nblk1 + i];
            c = block_sum_u64(c, smem_u);
            if (threadIdx.x == 0) out.count[b] = c;
        }
        if (threadIdx.x == 0) {
            double a_out = a_new;
            if (d != 0.0) a_out = d * out.a_old[b] + (1.0 - d) * a_new;
            out.a_new[b] = a_out;
            if (out.v) out.v[b] = v;
        }
    }
    const T vinv = T(a_in + a_new);
    const T dT = T(d), omd = T(1.0 - d);
    const T* p0 = in0 + b * n;
    const T* p1 = (Op::NIN > 1) ? in1 + b * n : nullptr;
    T* bo = reinterpret_cast<T*>(out.b_new) + b * n;
    const T* bold = out.b_old ? reinterpret_cast<const T*>(out.b_old) + b * n : nullptr;
    T* ro = out.r ? reinterpret_cast<T*>(out.r) + b * n : nullptr;
    const size_t stride = (size_t)gridDim.x * kBlock;
    if (VEC) {
        const size_t nv = n / VN;
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += stride) {
            V v0 = reinterpret_cast<const V*>(p0)[i];
            V v1 = v0;
            if (Op::NIN > 1) v1 = reinterpret_cast<const V*>(p1)[i];
            V vo, vr, vold;
            if (bold) vold = reinterpret_cast<const V*>(bold)[i];
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                T x0 = vget<T>(v0, j), x1 = vget<T>(v1, j);
                EwOut<T> o = op.eval(x0, x1);
                T bn = o.r * vinv - ((Op::TARGET == 0) ? x0 : x1);
                if (bold) bn = dT * vget<T>(vold, j) + omd * bn;
                vset<T>(vo, j, bn);
                vset<T>(vr, j, o.r);
            }
            reinterpret_cast<V*>(bo)[i] = vo;
            if (ro) reinterpret_cast<V*>(ro)[i] = vr;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            T x0 = p0[i];
            T x1 = (Op::NIN > 1) ? p1[i] : x0;
            EwOut<T> o = op.eval(x0, x1);
            T bn = o.r * vinv - ((Op::TARGET == 0) ? x0 : x1);
            if (bold) bn = dT * bold[i] + omd * bn;
            bo[i] = bn;
            if (ro) ro[i] = o.r;
        }
    }
}

// ---------------------------------------------------------------- single-launch form (small n)
// An instance that fits one workgroup's registers (n <= 256 threads x 8 vectors: 4096 f64 /
// 8192 f32 components) needs no second pass: load the tile, (r_i, v_i), block sum, a_new,
// b_new from the registers, store - one launch, each byte moved once.  This is the latency
// regime (BASELINE configs[0], N = 1000): one launch instead of two per factor update.
// The same idea across SEVERAL workgroups (a team of P resident blocks per instance, partial
// sums and arrive/leave counters as agent-scope atomics, tile kept in registers across the
// team barrier) was built and measured on MI355X and does NOT pay: 64 x 10^6 f64 took
// 1.3-1.8 ms against 0.30-0.38 ms for the two-pass kernels below.  A barrier round costs
// ~15-20 us of pure latency (store ack -> atomic -> poll -> partial loads), the chip holds
// only ~50 MB of tiles per round, and three waves per SIMD (147 VGPRs with the tile resident)
// cannot overlap enough rounds; an agent-scope RELEASE on the arrival additionally writes
// back the whole L2 (buffer_wbl2).  Two streaming passes with thousands of independent
// workgroups win by 4x, so that is what large instances use.
constexpr int kSingleVpt = 8;

template <typename T, typename Op>
__global__ __launch_bounds__(kBlock) void ew_single(Op op, size_t n, const T* __restrict__ in0,
                                                    const T* __restrict__ in1, MsgOut out) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    constexpr int VPT = kSingleVpt;
    __shared__ double smem[kBlock / kWave];
    __shared__ unsigned long long smem_u[kBlock / kWave];
    const size_t b = blockIdx.x;
    const size_t nv = n / VN;
    const double d = out.damping;
    const T dT = T(d), omd = T(1.0 - d);
    op.setup(b);
    const V* p0 = reinterpret_cast<const V*>(in0 + b * n);
    const V* p1 = (Op::NIN > 1) ? reinterpret_cast<const V*>(in1 + b * n) : nullptr;
    V tgt[VPT], oth[VPT], rr[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const size_t i = threadIdx.x + (size_t)k * kBlock;
        if (i < nv) {
            const V a0 = p0[i];
            V a1 = a0;
            if (Op::NIN > 1) a1 = p1[i];
            tgt[k] = (Op::TARGET == 0) ? a0 : a1;
            oth[k] = (Op::TARGET == 0) ? a1 : a0;
        }
    }
    double acc = 0.0;
    unsigned long long cnt = 0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const size_t i = threadIdx.x + (size_t)k * kBlock;
        if (i < nv) {
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                const T xt = vget<T>(tgt[k], j), xo = vget<T>(oth[k], j);
                EwOut<T> o = op.eval((Op::TARGET == 0) ? xt : xo, (Op::TARGET == 0) ? xo : xt);
                acc += (double)o.v;
                if (not_finite(o.r)) acc += __longlong_as_double(0x7ff8000000000000LL);
                if (OpCounts<Op>::value) cnt += o.mask & 1u;
                vset<T>(rr[k], j, o.r);
            }
        }
    }
    acc = block_sum(acc, smem);
    if (out.count) cnt = block_sum_u64(cnt, smem_u);
    const double v = acc / (double)n;
    const double a_in = op.a_msg;
    const double a_new = tmath::a_new_of(v, a_in);
    if (threadIdx.x == 0) {
        double a_out = a_new;
        if (d != 0.0) a_out = d * out.a_old[b] + (1.0 - d) * a_new;
        out.a_new[b] = a_out;
        if (out.v) out.v[b] = v;
        if (out.count) out.count[b] = cnt;
    }
    const T vinv = T(a_in + a_new);
    V* bo = reinterpret_cast<V*>(reinterpret_cast<T*>(out.b_new) + b * n);
    const V* bold = (d != 0.0) ? reinterpret_cast<const V*>(reinterpret_cast<const T*>(out.b_old) + b * n) : nullptr;
    V* ro = out.r ? reinterpret_cast<V*>(reinterpret_cast<T*>(out.r) + b * n) : nullptr;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const size_t i = threadIdx.x + (size_t)k * kBlock;
        if (i < nv) {
            V vo, vold;
            if (bold) vold = bold[i];
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                T bn = vget<T>(rr[k], j) * vinv - vget<T>(tgt[k], j);
                if (bold) bn = dT * vget<T>(vold, j) + omd * bn;
                vset<T>(vo, j, bn);
            }
            bo[i] = vo;
            if (ro) ro[i] = rr[k];
        }
    }
}

// ---------------------------------------------------------------- streaming tile form (large n)
// Launch shape measured with tools/probes/probe_copy.hip: a block that owns ONE contiguous tile of
// 256 x U 16-byte pieces, no loop, all loads issued before the first use, grid = tiles x instances
// (workgroups are dispatched in address order, so HBM sees a sequential stream).  The grid-stride
// kernels above (<= 1024 blocks per instance, each block folding the partials before it streams)
// stay for ragged n.  Three kernels:
//   ew_moments_tile  pass 1: (r_i, v_i) per component, sum v_i per tile        -> partials[inst, tile]
//   ew_finalize      one block per instance: fixed-order fold, a_new, scalars -> scal[inst] = a_in + a_new
//   ew_apply_tile    pass 2: b_new = r (a_in + a_new) - b (+ damping), r recomputed (light ops) or
//                    read back from the scratch (heavy ops)
// With producer-fused moments (prior_hint.h) only the last two run: the prior reads b once.
constexpr int kTileU = 2;      // probe_copy: 5.85 TB/s (U = 2) vs 5.77 (U = 4) vs 4.25 (U = 8); 6.2 with streaming stores

// 16-byte store that does not allocate in the caches (outputs larger than the Infinity Cache)
template <typename T, bool NTS>
__device__ __forceinline__ void store_vec(typename Vec<T>::type* dst, const typename Vec<T>::type& v) {
    if (NTS) {
        using N = typename NativeVec<T>::type;
        __builtin_nontemporal_store(*reinterpret_cast<const N*>(&v), reinterpret_cast<N*>(dst));
    } else {
        *dst = v;
    }
}

// The redo of a lean resident update (run_tiles_u: redo_nan): only the instances whose a_new came back NaN are updated,
// in TWO launches - pass 1 decides per instance from a_flag (nothing writes a_new before pass 2), leaves flags[] for pass 2
// and the count of flagged instances for the host's heuristic; pass 2 folds the tile sums itself (the arithmetic of
// ew_finalize, by every workgroup of the instance; tile 0 writes a_new / v).  All null: a plain run.
struct RedoArgs {
    const double* a_flag = nullptr;      // pass 1: instance b is redone iff a_flag[b] is NaN
    unsigned* flags = nullptr;           // pass 1 (tile 0) writes, pass 2 reads
    volatile unsigned* stat = nullptr;   // pass 1, workgroup (0, 0) of the first chunk: {flagged instances, batch} (pinned host)
    size_t batch = 0;
    const double* partials = nullptr;    // pass 2: this chunk's tile sums of pass 1
    int nblk = 0;
    const double* a_in = nullptr;
};

// A block owns `tpb` CONSECUTIVE tiles (a contiguous chunk of the instance): the per-instance setup of the
// functor (a log and a division for the Gauss-Bernoulli prior) and the block reduction are paid once per
// chunk, not once per 1024 components, while neighbouring blocks still stream neighbouring addresses.
template <typename T, typename Op, int U>
__global__ __launch_bounds__(kBlock) void ew_moments_tile(Op op, size_t n, size_t inst0, int tpb, const T* __restrict__ in0,
                                                          const T* __restrict__ in1, T* __restrict__ rbuf,
                                                          double* __restrict__ partials,
                                                          unsigned long long* __restrict__ pcount,
                                                          uint8_t* __restrict__ mask_out, RedoArgs rd = RedoArgs{}) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    __shared__ double smem[kBlock / kWave];
    __shared__ unsigned long long smem_u[kBlock / kWave];
    const size_t b = inst0 + blockIdx.y;
    if (rd.a_flag) {                            // redo of the NaN instances only
        if (rd.stat && inst0 == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
            unsigned long long c = 0;
            for (size_t i = threadIdx.x; i < rd.batch; i += kBlock) {
                const double a = rd.a_flag[i];
                c += (a == a) ? 0u : 1u;
            }
            c = block_sum_u64(c, smem_u);
            if (threadIdx.x == 0) {
                rd.stat[0] = (unsigned)c;
                rd.stat[1] = (unsigned)rd.batch;
            }
        }
        const double a = rd.a_flag[b];
        const bool flagged = !(a == a);
        if (blockIdx.x == 0 && threadIdx.x == 0) rd.flags[b] = flagged ? 1u : 0u;
        if (!flagged) return;
    }
    op.setup(b);
    const size_t nv = n / VN;
    const V* p0 = reinterpret_cast<const V*>(in0 + b * n);
    const V* p1 = (Op::NIN > 1) ? reinterpret_cast<const V*>(in1 + b * n) : nullptr;
    V* pr = rbuf ? reinterpret_cast<V*>(rbuf + (size_t)blockIdx.y * n) : nullptr;
    uint8_t* mo = mask_out ? mask_out + b * n : nullptr;
    double acc = 0.0;
    unsigned long long cnt = 0;
    // (requesting the inputs of tile t+1 before tile t is evaluated was measured: no gain, +30 VGPRs)
    for (int tile = 0; tile < tpb; ++tile) {
        const size_t base = ((size_t)blockIdx.x * tpb + tile) * kBlock * U + threadIdx.x;
        if (base - threadIdx.x >= nv) break;
        V v0[U], v1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                v0[u] = p0[i];
                if (Op::NIN > 1) v1[u] = p1[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                V vr;
#pragma unroll
                for (int j = 0; j < VN; ++j) {
                    const T x0 = vget<T>(v0[u], j);
                    const T x1 = (Op::NIN > 1) ? vget<T>(v1[u], j) : x0;
                    EwOut<T> o = op.eval(x0, x1);
                    acc += (double)o.v;
                    if (not_finite(o.r)) acc += __longlong_as_double(0x7ff8000000000000LL);
                    if (OpCounts<Op>::value) cnt += o.mask & 1u;
                    vset<T>(vr, j, o.r);
                    if (mo) mo[i * VN + j] = (uint8_t)o.mask;
                }
                if (pr) pr[i] = vr;
            }
        }
    }
    acc = block_sum(acc, smem);
    if (pcount) cnt = block_sum_u64(cnt, smem_u);
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
        if (pcount) pcount[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = cnt;
    }
}

// One block per instance of the chunk.  partials[(pinst0 + blockIdx.x) * nblk + i]: the chunk-relative
// layout of ew_moments_tile has pinst0 = 0, a producer's partials are indexed by the global instance.
__global__ __launch_bounds__(kBlock) void ew_finalize(size_t n, size_t inst0, size_t pinst0,
                                                      const double* __restrict__ a_in_d,
                                                      const double* __restrict__ partials,
                                                      const unsigned long long* __restrict__ pcount, int nblk,
                                                      MsgOut out, double* __restrict__ scal) {
    __shared__ double smem[kBlock / kWave];
    __shared__ unsigned long long smem_u[kBlock / kWave];
    const size_t bl = blockIdx.x, b = inst0 + bl;
    const double* pp = partials + (pinst0 + bl) * (size_t)nblk;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) s += pp[i];
    s = block_sum(s, smem);
    unsigned long long c = 0;
    if (out.count) {
        for (int i = threadIdx.x; i < nblk; i += kBlock) c += pcount[(pinst0 + bl) * (size_t)nblk + i];
        c = block_sum_u64(c, smem_u);
    }
    if (threadIdx.x == 0) {
        const double v = s / (double)n;
        const double a_in = a_in_d[b];
        const double a_new = tmath::a_new_of(v, a_in);
        const double d = out.damping;
        out.a_new[b] = (d != 0.0) ? d * out.a_old[b] + (1.0 - d) * a_new : a_new;
        if (out.v) out.v[b] = v;
        if (out.count) out.count[b] = c;
        scal[bl] = a_in + a_new;
    }
}

// RECOMPUTE = true: r from op.eval on the inputs; false: r from the scratch written by pass 1.
template <typename T, typename Op, int U, bool RECOMPUTE, bool NTS>
__global__ __launch_bounds__(kBlock) void ew_apply_tile(Op op, size_t n, size_t inst0, int tpb, const T* __restrict__ in0,
                                                        const T* __restrict__ in1, const T* __restrict__ rbuf,
                                                        const double* __restrict__ scal, MsgOut out,
                                                        RedoArgs rd = RedoArgs{}) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    const size_t bl = blockIdx.y, b = inst0 + bl;
    if (rd.flags && !rd.flags[b]) return;
    const size_t nv = n / VN;
    const V* p0 = reinterpret_cast<const V*>(in0 + b * n);
    const V* p1 = (Op::NIN > 1) ? reinterpret_cast<const V*>(in1 + b * n) : nullptr;
    const V* pr = RECOMPUTE ? nullptr : reinterpret_cast<const V*>(rbuf + bl * n);
    const V* bold = out.b_old ? reinterpret_cast<const V*>(reinterpret_cast<const T*>(out.b_old) + b * n) : nullptr;
    V* bo = reinterpret_cast<V*>(reinterpret_cast<T*>(out.b_new) + b * n);
    V* ro = out.r ? reinterpret_cast<V*>(reinterpret_cast<T*>(out.r) + b * n) : nullptr;
    if (RECOMPUTE) op.setup(b);
    T vinv;
    if (rd.partials) {
        // ew_finalize's arithmetic, by every workgroup of the instance (same fold, same result); tile 0 writes the scalars
        __shared__ double smem[kBlock / kWave];
        const double* pp = rd.partials + bl * (size_t)rd.nblk;
        double s = 0.0;
        for (int i = threadIdx.x; i < rd.nblk; i += kBlock) s += pp[i];
        s = block_sum(s, smem);
        const double v = s / (double)n;
        const double a_in = rd.a_in[b];
        const double a_new = tmath::a_new_of(v, a_in);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const double d = out.damping;
            out.a_new[b] = (d != 0.0) ? d * out.a_old[b] + (1.0 - d) * a_new : a_new;
            if (out.v) out.v[b] = v;
        }
        vinv = T(a_in + a_new);
    } else {
        vinv = T(scal[bl]);
    }
    const T dT = T(out.damping), omd = T(1.0 - out.damping);
    for (int tile = 0; tile < tpb; ++tile) {
        const size_t base = ((size_t)blockIdx.x * tpb + tile) * kBlock * U + threadIdx.x;
        if (base - threadIdx.x >= nv) break;
        V v0[U], v1[U], vold[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                if (RECOMPUTE) {
                    v0[u] = p0[i];
                    if (Op::NIN > 1) v1[u] = p1[i];
                } else {
                    v0[u] = (Op::TARGET == 0) ? p0[i] : p1[i];        // the target message b
                    v1[u] = pr[i];                                    // r
                }
                if (bold) vold[u] = bold[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                V vo, vr;
#pragma unroll
                for (int j = 0; j < VN; ++j) {
                    T r, tgt;
                    if (RECOMPUTE) {
                        const T x0 = vget<T>(v0[u], j);
                        const T x1 = (Op::NIN > 1) ? vget<T>(v1[u], j) : x0;
                        r = op.eval(x0, x1).r;
                        tgt = (Op::TARGET == 0) ? x0 : x1;
                    } else {
                        r = vget<T>(v1[u], j);
                        tgt = vget<T>(v0[u], j);
                    }
                    T bn = r * vinv - tgt;
                    if (bold) bn = dT * vget<T>(vold[u], j) + omd * bn;
                    vset<T>(vo, j, bn);
                    vset<T>(vr, j, r);
                }
                store_vec<T, NTS>(&bo[i], vo);
                if (ro) store_vec<T, NTS>(&ro[i], vr);
            }
        }
    }
}

// Instances are processed in chunks small enough that pass 2 finds r and b in the
// 256 MiB Infinity Cache: HBM then sees each input once and b_new once (2T / 3T per
// component, the algorithmic minimum) although the update is two launches.
static size_t env_size(const char* name, size_t dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    long v = atol(e);
    return v >= 0 ? (size_t)v : dflt;
}

// Streaming-tile realisation of one update.  `hint` != nullptr: a producer left sum v_i partials for
// this very message (prior_hint.h): no pass 1, the message is read once.  redo_nan: only the instances whose out.a_new
// is NaN are updated (the others' workgroups leave at once).
template <typename T, typename Op, int U>
static int run_tiles_u(tramp_ctx* ctx, Op op, size_t n, size_t batch, const void* in0, const void* in1,
                       const double* a_in_d, const MsgOut& out, bool store_r, const TrampPriorHint* hint,
                       bool redo_nan = false) {
    constexpr int VN = Vec<T>::N;
    const size_t nv = n / VN;
    const size_t tiles = (nv + (size_t)kBlock * U - 1) / ((size_t)kBlock * U);
    // consecutive tiles per block: enough blocks to fill the chip several times over, setup amortised
    int tpb = 8;
    while (tpb > 1 && (tiles / tpb) * batch < 4096) tpb >>= 1;
    // redo of a few flagged instances: most workgroups only read their instance's flag and leave - long runs per block keep
    // the dispatch of those short; two launches (RedoArgs), ~6 us on a clean 64 x 1e6 update.  16 tiles per block (round 6;
    // 64 before): a flagged instance of 1e6 components is then 31 workgroups, not 8 - with 2-3 of 64 instances flagged the
    // redo costs 0.07 ms instead of 0.19 ms, and nothing when none is (tools/probes/probe_exp_prior.py)
    static const size_t redo_tpb = env_size("TRAMP_EW_REDO_TPB", 16);
    if (redo_nan) tpb = (int)redo_tpb;
    const unsigned ntile = (unsigned)((tiles + tpb - 1) / tpb);          // blocks (= partials) per instance
    if (hint) {
        store_r = false;
        ctx->hint_used += 1;
    }
    const size_t r_bytes = store_r ? ((batch * n * sizeof(T) + 255) / 256) * 256 : 0;
    const size_t scal_bytes = ((batch * sizeof(double) + 255) / 256) * 256;
    const size_t part_bytes = hint ? 0 : batch * (size_t)ntile * (sizeof(double) + sizeof(unsigned long long));
    const size_t head = redo_nan ? ((batch * sizeof(unsigned) + 255) / 256) * 256 : 0;
    int rc = tramp_ws_reserve(ctx, head + r_bytes + scal_bytes + part_bytes);
    if (rc) return rc;
    char* ws = reinterpret_cast<char*>(ctx->ws) + head;
    RedoArgs rd1, rd2;                  // pass 1 / pass 2 of a redo (RedoArgs); empty otherwise
    if (redo_nan) {
        TRAMP_REQUIRE(ctx, !hint && !out.count && !out.mask);
        rd1.a_flag = out.a_new;
        rd1.flags = reinterpret_cast<unsigned*>(ctx->ws);
        rd1.stat = ctx->lean_stat;
        rd1.batch = batch;
        rd2.flags = rd1.flags;
        rd2.nblk = (int)ntile;
        rd2.a_in = a_in_d;
    }
    T* rbuf = store_r ? reinterpret_cast<T*>(ws) : nullptr;
    double* scal = reinterpret_cast<double*>(ws + r_bytes);
    double* partials = reinterpret_cast<double*>(ws + r_bytes + scal_bytes);
    unsigned long long* pcount = (out.count && !hint) ? reinterpret_cast<unsigned long long*>(partials + batch * (size_t)ntile) : nullptr;
    MsgOut o = out;
    if (o.damping == 0.0) { o.a_old = nullptr; o.b_old = nullptr; }
    const T* p0 = reinterpret_cast<const T*>(in0);
    const T* p1 = reinterpret_cast<const T*>(in1);
    ProfScope prof(ctx, PROF_EW, (double)(Op::NIN + 1) * sizeof(T) * (double)n * (double)batch, 0.0);
    for (size_t inst0 = 0; inst0 < batch; inst0 += 65535) {
        const unsigned nb = (unsigned)((batch - inst0 < 65535) ? batch - inst0 : 65535);
        dim3 grid(ntile, nb), block(kBlock);
        if (hint) {
            hipLaunchKernelGGL(ew_finalize, dim3(nb), block, 0, ctx->stream, n, inst0, inst0, a_in_d, hint->partials,
                               (const unsigned long long*)nullptr, hint->nblk, o, scal + inst0);
        } else {
            hipLaunchKernelGGL((ew_moments_tile<T, Op, U>), grid, block, 0, ctx->stream, op, n, inst0, tpb, p0, p1,
                               rbuf ? rbuf + inst0 * n : nullptr, partials + inst0 * ntile,
                               pcount ? pcount + inst0 * ntile : nullptr, out.mask, rd1);
            if (redo_nan)
                rd2.partials = partials + inst0 * ntile;        // pass 2 finalises inline
            else
                hipLaunchKernelGGL(ew_finalize, dim3(nb), block, 0, ctx->stream, n, inst0, (size_t)0, a_in_d,
                                   partials + inst0 * ntile, pcount ? pcount + inst0 * ntile : nullptr, (int)ntile, o,
                                   scal + inst0);
        }
        // outputs past the Infinity Cache are streamed around the caches (+6 % on the box's copy rate)
        const bool nts = (double)n * (double)batch * sizeof(T) > 192.0 * 1024 * 1024;
        // the apply pass of the light ops has next to no per-block setup to amortise: shorter runs per block
        // stream faster (fused Gauss-Bernoulli / L1 update at 64 x 1e6: 173 / 180 us with 2 tiles per block
        // against 192 / 190 with 8; gpurun_out/r03/fused1.log)
        const int tpa = (Op::HEAVY || redo_nan) ? tpb : (tpb > 2 ? 2 : tpb);
        dim3 grid_a((unsigned)((tiles + tpa - 1) / tpa), nb);
        if (store_r) {
            if (nts)
                hipLaunchKernelGGL((ew_apply_tile<T, Op, U, false, true>), grid_a, block, 0, ctx->stream, op, n, inst0, tpa,
                                   p0, p1, rbuf + inst0 * n, scal + inst0, o, rd2);
            else
                hipLaunchKernelGGL((ew_apply_tile<T, Op, U, false, false>), grid_a, block, 0, ctx->stream, op, n, inst0,
                                   tpa, p0, p1, rbuf + inst0 * n, scal + inst0, o, rd2);
        } else {
            if (nts)
                hipLaunchKernelGGL((ew_apply_tile<T, Op, U, true, true>), grid_a, block, 0, ctx->stream, op, n, inst0, tpa,
                                   p0, p1, (const T*)nullptr, scal + inst0, o, rd2);
            else
                hipLaunchKernelGGL((ew_apply_tile<T, Op, U, true, false>), grid_a, block, 0, ctx->stream, op, n, inst0, tpa,
                                   p0, p1, (const T*)nullptr, scal + inst0, o, rd2);
        }
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// pieces per thread and tile: 1 and 4 measured for the heavy ops (K4 / K5 / K7): within 4 % of 2; one piece per thread
// for the light ops measured too (the flat one-piece copy of tools/probes/probe_copy.hip streams 6 % faster than the
// U = 2 tile): not better in the update
template <typename T, typename Op>
static int run_tiles(tramp_ctx* ctx, Op op, size_t n, size_t batch, const void* in0, const void* in1,
                     const double* a_in_d, const MsgOut& out, bool store_r, const TrampPriorHint* hint, bool redo_nan = false) {
    // two-input heavy factors (K7): one piece per thread and tile - 326 us against 396 (U = 2) and 357 (U = 4) for the
    // moments pass at 64 x 1e6 f64 (tools/probes/probe_k7.hip, round 5: the evaluation's temporaries times U is what the
    // compiler's schedule keeps live)
    if constexpr (Op::HEAVY && Op::NIN > 1 && sizeof(T) == 8)
        return run_tiles_u<T, Op, 1>(ctx, op, n, batch, in0, in1, a_in_d, out, store_r, hint, redo_nan);
    else
        return run_tiles_u<T, Op, kTileU>(ctx, op, n, batch, in0, in1, a_in_d, out, store_r, hint, redo_nan);
}

bool resident_covers_shape(const tramp_ctx* ctx, size_t n, size_t batch, size_t elem_size, size_t tv, int bpc) {
    static const size_t on = env_size("TRAMP_EW_RESIDENT", 1);
    static const size_t min_rounds = env_size("TRAMP_EW_RESIDENT_MIN_ROUNDS", 4);
    const size_t vn = 16 / elem_size;
    if (!on || !ctx || ctx->num_cu < 16 || batch == 0 || (n % vn) != 0) return false;
    const size_t nv = n / vn;
    // smaller instances: one workgroup each (ew_single); the tile offsets are 32-bit
    if (n <= (size_t)kBlock * kSingleVpt * vn || nv < tv || nv > 0x7fffffffull || batch > 0x3fffffffull) return false;
    const size_t ps = (size_t)bpc * ctx->num_cu - 8, tiles = (nv + tv - 1) / tv;
    // an instance must fit the window of rounds the pipeline keeps in flight, and there must be rounds to overlap
    return tiles <= ps * (kResDA - 2) + 1 && batch * tiles >= min_rounds * ps;
}
bool resident_covers(const tramp_ctx* ctx, size_t n, size_t batch, size_t elem_size) {
    return resident_covers_shape(ctx, n, batch, elem_size, (size_t)kBlock * kResU, 2);        // kResBlocksPerCU (ew_resident.h; static_assert in ew_resident_impl.h)
}

// ---------------------------------------------------------------- one launch for a FEW large instances (round 5)
// BASELINE configs[1] read literally is ONE instance of 10^6 components: 16 MB that the tile form handles in three
// launches (14 us, launch-bound).  Here G workgroups share an instance, all of them resident at once (batch * G <= 2 x
// CUs, checked by an occupancy query): each loads its slice into registers, evaluates, and publishes its block sum of v
// in an 8-byte self-validating granule (as ew_resident.h).  Workgroup 0 of the instance polls the G granules, folds them
// in index order (a run-to-run reproducible mean) and publishes the total in up to 16 copies; the others poll one copy
// (<= 32 pollers per line) and write the update from their registers: the message is read once and written once in
// ONE launch.
//
// No atomics and no reset launch: the granules live in TWO banks and a device-side epoch word picks the bank.  Every
// launch reads the epoch first, works in bank (epoch & 1), and puts the whole OTHER bank back to the sentinel (the
// launch before used it and is complete; the next one will use it).  Workgroup (0, 0) adds one to the epoch as its last
// act, after it has seen the total of every instance - by then every workgroup of the launch has read the epoch.  The
// state lives on the device, so the launch replays unchanged inside a hipGraph.  (Measured on the way here: an atomic
// arrival counter costs 0.2 us per same-address atomic - 7 us at 32 per address; one lane resetting 489 granules in turn
// 10 us; every workgroup polling all G granules 26 us.)
//
// A hand-off that has not arrived by the deadline is given up like the resident pipeline's: NaN result, host-visible
// error word, tile kernels afterwards.
constexpr unsigned long long kCoopSentinel = 0xFFFFFFFFFFFFFFFFull;
constexpr unsigned long long kCoopNaN = 0x7ff8000000000000ull;

template <typename T, typename Op, int K>
__global__ __launch_bounds__(kBlock, 2) void ew_coop(Op op, size_t n, unsigned G, size_t chunk, const T* __restrict__ in0,
                                                     const T* __restrict__ in1, MsgOut out, unsigned long long* __restrict__ slots,
                                                     size_t capacity, unsigned* __restrict__ epoch, long long deadline_ticks,
                                                     volatile unsigned* err_host, unsigned dbg_skip_plus1) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    __shared__ double smem[kBlock / kWave];
    __shared__ double s_tot;
    const long long t_start = (long long)__builtin_amdgcn_s_memrealtime();
    const long long lim = deadline_ticks > 0 ? deadline_ticks : 200000000LL;
    const size_t b = blockIdx.y;
    const unsigned g = blockIdx.x, tid = threadIdx.x;
    const unsigned e = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const size_t lines = coop_total_lines(capacity), bank = capacity + lines * kCoopLineStride;
    unsigned long long* mine_bank = slots + (e & 1u) * bank;
    op.setup(b);
    const size_t nv = n / VN;
    const size_t p0 = (size_t)g * chunk, p1 = (p0 + chunk < nv) ? p0 + chunk : nv;
    const V* q0 = reinterpret_cast<const V*>(in0 + b * n);
    const V* q1 = (Op::NIN > 1) ? reinterpret_cast<const V*>(in1 + b * n) : nullptr;
    const double d = out.damping;
    const V* bold = (d != 0.0) ? reinterpret_cast<const V*>(reinterpret_cast<const T*>(out.b_old) + b * n) : nullptr;
    V x0[K], x1[K], vold[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t i = p0 + tid + (size_t)k * kBlock;
        if (i < p1) {
            x0[k] = q0[i];
            if (Op::NIN > 1) x1[k] = q1[i];
            if (bold) vold[k] = bold[i];
        }
    }
    {   // the other bank back to the sentinel: [capacity] block sums + one granule per total line (after the message's
        // loads are out: the stores wait for the epoch word, the loads need not)
        unsigned long long* other = slots + ((e + 1u) & 1u) * bank;
        const size_t npos = capacity + lines, nthr = (size_t)gridDim.x * gridDim.y * kBlock;
        for (size_t pos = ((size_t)b * gridDim.x + g) * kBlock + tid; pos < npos; pos += nthr) {
            const size_t at = pos < capacity ? pos : capacity + (pos - capacity) * kCoopLineStride;
            __hip_atomic_store(other + at, kCoopSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    V rr[K];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t i = p0 + tid + (size_t)k * kBlock;
        if (i < p1) {
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                const T a0 = vget<T>(x0[k], j);
                const T a1 = (Op::NIN > 1) ? vget<T>(x1[k], j) : a0;
                EwOut<T> o = op.eval(a0, a1);
                acc += (double)o.v;
                if (not_finite(o.r)) acc += __longlong_as_double((long long)kCoopNaN);
                vset<T>(rr[k], j, o.r);
            }
            if (Op::TARGET != 0) x0[k] = x1[k];                     // keep the target message
        }
    }
    acc = block_sum(acc, smem);
    unsigned long long* sl = mine_bank + b * (size_t)G;                 // this instance's block sums
    const unsigned R = (G + 31u) / 32u < 16u ? (G + 31u) / 32u : 16u;    // copies of the total, <= 32 pollers each
    unsigned long long* tot = mine_bank + capacity + (b * R) * kCoopLineStride;
    if (tid == 0) {
        unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
        if (ab == kCoopSentinel) ab = kCoopNaN;
        // (debug: workgroup dbg_skip_plus1 - 1 of instance 0 never publishes - the give-up path, tests/test_gpu_coop.py)
        if (!(b == 0 && g + 1 == dbg_skip_plus1)) __hip_atomic_store(sl + g, ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < kWave) {
        double part = 0.0;
        bool gave_up = false;
        if (g == 0) {
            for (;;) {
                bool ok = true;
                part = 0.0;
                for (unsigned i = tid; i < G; i += kWave) {            // lane-strided, then a fixed-order wave sum
                    const unsigned long long bits = __hip_atomic_load(sl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && bits != kCoopSentinel;
                    part += __longlong_as_double((long long)bits);
                }
                if (__all(ok)) break;
                if ((long long)__builtin_amdgcn_s_memrealtime() - t_start > lim) {
                    gave_up = true;
                    break;
                }
            }
            part = __shfl(wave_sum(part), 0);                          // lane 0 holds the sum
            if (gave_up) part = __longlong_as_double((long long)kCoopNaN);
            unsigned long long tb = (unsigned long long)__double_as_longlong(part);
            if (tb == kCoopSentinel) tb = kCoopNaN;
            if (tid < R) __hip_atomic_store(tot + tid * kCoopLineStride, tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long* mine = tot + (g % R) * kCoopLineStride;
            for (;;) {
                const unsigned long long bits = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bits != kCoopSentinel) {
                    part = __longlong_as_double((long long)bits);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if ((long long)__builtin_amdgcn_s_memrealtime() - t_start > lim + 1000000LL) {     // + 10 ms: workgroup 0 reports first
                    gave_up = true;
                    part = __longlong_as_double((long long)kCoopNaN);
                    break;
                }
            }
        }
        if (gave_up && tid == 0 && err_host) *err_host = 1u;
        if (tid == 0) s_tot = part;
    }
    __syncthreads();
    const double v = s_tot / (double)n;
    const double a_in = op.a_msg;
    const double a_new = tmath::a_new_of(v, a_in);
    if (g == 0 && tid == 0) {
        out.a_new[b] = (d != 0.0) ? d * out.a_old[b] + (1.0 - d) * a_new : a_new;
        if (out.v) out.v[b] = v;
    }
    const T vinv = T(a_in + a_new), dT = T(d), omd = T(1.0 - d);
    V* bo = reinterpret_cast<V*>(reinterpret_cast<T*>(out.b_new) + b * n);
    V* ro = out.r ? reinterpret_cast<V*>(reinterpret_cast<T*>(out.r) + b * n) : nullptr;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t i = p0 + tid + (size_t)k * kBlock;
        if (i < p1) {
            V vo;
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                T bn = vget<T>(rr[k], j) * vinv - vget<T>(x0[k], j);
                if (bold) bn = dT * vget<T>(vold[k], j) + omd * bn;
                vset<T>(vo, j, bn);
            }
            bo[i] = vo;
            if (ro) ro[i] = rr[k];
        }
    }
    // the epoch moves on once the total of EVERY instance is out: every workgroup of the launch has then read it
    if (g == 0 && b == 0 && tid < kWave) {
        const unsigned long long* t0 = mine_bank + capacity;
        for (;;) {
            bool ok = true;
            for (unsigned i = tid; i < gridDim.y; i += kWave)
                ok = ok && __hip_atomic_load(t0 + ((size_t)i * R) * kCoopLineStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kCoopSentinel;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((long long)__builtin_amdgcn_s_memrealtime() - t_start > lim + 2000000LL) {
                if (tid == 0 && err_host) *err_host = 1u;
                break;
            }
        }
        if (tid == 0) __hip_atomic_store(epoch, e + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, typename Op, int K>
static int launch_coop(tramp_ctx* ctx, const Op& op, size_t n, size_t batch, unsigned G, size_t chunk, const void* in0, const void* in1,
                       const MsgOut& out) {
    static int occ = -1;
    if (occ < 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ew_coop<T, Op, K>, kBlock, 0) != hipSuccess) occ = 0;
    if (occ < 2) return kResNotEligible;
    static const long long timeout_ms = getenv("TRAMP_EW_RESIDENT_TIMEOUT_MS") ? atoll(getenv("TRAMP_EW_RESIDENT_TIMEOUT_MS")) : 2000;
    static const unsigned dbg_skip = getenv("TRAMP_EW_COOP_DBG_SKIP") ? (unsigned)atoi(getenv("TRAMP_EW_COOP_DBG_SKIP")) : 0u;
    MsgOut o = out;
    if (o.damping == 0.0) { o.a_old = nullptr; o.b_old = nullptr; }
    ProfScope prof(ctx, PROF_EW, (double)(Op::NIN + 1) * sizeof(T) * (double)n * (double)batch, 0.0);
    hipLaunchKernelGGL((ew_coop<T, Op, K>), dim3(G, (unsigned)batch), dim3(kBlock), 0, ctx->stream, op, n, G, chunk,
                       reinterpret_cast<const T*>(in0), reinterpret_cast<const T*>(in1), o, ctx->coop_slots, ctx->capacity_coop, ctx->coop_epoch,
                       (timeout_ms > 0 ? timeout_ms : 2000) * 100000LL, ctx->lean_stat ? ctx->lean_stat + 2 : nullptr, dbg_skip);
    TRAMP_LAUNCH_CHECK(ctx);
    ctx->coop_runs += 1;
    return TRAMP_OK;
}

// kResNotEligible: the shape (or the context's state) is not for the one-launch form
template <typename T, typename Op>
static int run_coop(tramp_ctx* ctx, const Op& op, size_t n, size_t batch, const void* in0, const void* in1, const MsgOut& out) {
    static const size_t on = env_size("TRAMP_EW_COOP", 1);
    static const bool cu_masked = (getenv("HSA_CU_MASK") && *getenv("HSA_CU_MASK")) || (getenv("ROC_GLOBAL_CU_MASK") && *getenv("ROC_GLOBAL_CU_MASK"));
    constexpr int VN = Vec<T>::N;
    if (!on || cu_masked || !ctx->coop_slots || ctx->capacity_coop == 0 || (ctx->lean_stat && ctx->lean_stat[2])) return kResNotEligible;
    if (out.mask || out.count || (n % VN) != 0 || batch == 0 || batch > ctx->capacity_coop) return kResNotEligible;
    const size_t nv = n / VN;
    const size_t gmax = (nv + kBlock - 1) / kBlock;              // at least one piece per thread
    // workgroups per instance, all of them resident at once.  The light updates are bound by the exchange, which is faster
    // among fewer workgroups: for one or two instances one per CU, where eight pieces per thread still cover the instance
    // (1 x 1e6 f64 Gauss-Bernoulli 10.2 -> 9.4 us, f32 9.0 -> 7.7 us; from four instances up it loses); the heavy ones want
    // the arithmetic spread over two per CU
    size_t G = 0, chunk = 0, per_thread = 0;
    for (int pass = (Op::HEAVY || batch > 2) ? 1 : 0; pass < 2; ++pass) {
        const size_t cap = pass == 0 ? ctx->capacity_coop / 2 : ctx->capacity_coop;
        G = cap / batch;
        if (G > gmax) G = gmax;
        if (G < 2) continue;
        chunk = (nv + G - 1) / G;
        G = (nv + chunk - 1) / chunk;
        per_thread = (chunk + kBlock - 1) / kBlock;
        if (per_thread <= 8) break;
    }
    if (G < 2 || per_thread == 0 || per_thread > 8) return kResNotEligible;
    if (per_thread <= 1) return launch_coop<T, Op, 1>(ctx, op, n, batch, (unsigned)G, chunk, in0, in1, out);
    if (per_thread <= 2) return launch_coop<T, Op, 2>(ctx, op, n, batch, (unsigned)G, chunk, in0, in1, out);
    if (per_thread <= 4) return launch_coop<T, Op, 4>(ctx, op, n, batch, (unsigned)G, chunk, in0, in1, out);
    return launch_coop<T, Op, 8>(ctx, op, n, batch, (unsigned)G, chunk, in0, in1, out);
}

template <typename T, typename Op>
static int run_two_pass(tramp_ctx* ctx, Op op, size_t n, size_t batch, const void* in0,
                        const void* in1, const double* a_in_d, const MsgOut& out,
                        const TrampPriorHint* hint = nullptr) {
    TRAMP_REQUIRE(ctx, in0 && out.a_new && out.b_new && a_in_d);
    TRAMP_REQUIRE(ctx, Op::NIN == 1 || in1);
    if (n == 0 || batch == 0) return TRAMP_OK;
    TRAMP_REQUIRE(ctx, out.damping == 0.0 || (out.a_old && out.b_old));
    constexpr int VN = Vec<T>::N;
    // large batches of large instances: the resident single-pass pipeline (each byte crosses HBM once).  A producer
    // that already left the moments (hint) makes the plain apply pass the cheaper single pass.
    if constexpr (ResidentOp<Op>::value) {
        if (!hint) {
            const int rc = run_resident<T, Op>(ctx, op, n, batch, in0, in1, a_in_d, out);
            if (rc != kResNotEligible) return rc;
        }
    }
    const bool vec = (n % VN) == 0;
    // a few instances too large for one workgroup, small enough for one wave of resident workgroups: ONE launch
    if (vec && !hint && n > (size_t)kBlock * kSingleVpt * VN) {
        const int rc = run_coop<T, Op>(ctx, op, n, batch, in0, in1, out);
        if (rc != kResNotEligible) return rc;
    }
    const int nblk = blocks_for(n, VN);
    // heavy factors (erfcx / log / tanh per component) park r in a scratch; light ones recompute
    const bool store_r = Op::HEAVY;
    // heavy factors hold 200+ VGPRs with the tile resident (1 wave/SIMD, 16 serial evaluations per
    // thread): single launch only when each thread has <= 2 vectors or instances alone fill the chip
    const size_t tile = (size_t)kBlock * kSingleVpt * VN;
    const bool single_ok = n <= tile && (!Op::HEAVY || n <= tile / 4 || batch >= 1024);
    if (vec && !out.mask && single_ok) {
        MsgOut o = out;
        if (o.damping == 0.0) { o.a_old = nullptr; o.b_old = nullptr; }
        ProfScope prof(ctx, PROF_EW, (double)(Op::NIN + 1) * sizeof(T) * (double)n * (double)batch, 0.0);
        hipLaunchKernelGGL((ew_single<T, Op>), dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, op, n,
                           reinterpret_cast<const T*>(in0), reinterpret_cast<const T*>(in1), o);
        TRAMP_LAUNCH_CHECK(ctx);
        return TRAMP_OK;
    }
    if (vec) return run_tiles<T, Op>(ctx, op, n, batch, in0, in1, a_in_d, out, store_r, hint);
    // ragged n (not a whole number of 16-byte pieces): the grid-stride kernels, <= 65535 instances per launch.
    // (Chunks of instances sized for the Infinity Cache between the passes were measured: 0.96-1.24 ms against 0.96.)
    const size_t cb = batch > 65535 ? 65535 : batch;
    const size_t r_bytes = store_r ? ((cb * n * sizeof(T) + 255) / 256) * 256 : 0;
    const size_t need = r_bytes + cb * (size_t)nblk * (sizeof(double) + sizeof(unsigned long long));
    int rc = tramp_ws_reserve(ctx, need);
    if (rc) return rc;
    T* rbuf = store_r ? reinterpret_cast<T*>(ctx->ws) : nullptr;
    double* partials = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->ws) + r_bytes);
    unsigned long long* pcount =
        out.count ? reinterpret_cast<unsigned long long*>(partials + cb * (size_t)nblk) : nullptr;
    MsgOut o = out;
    if (o.damping == 0.0) { o.a_old = nullptr; o.b_old = nullptr; }
    const T* p0 = reinterpret_cast<const T*>(in0);
    const T* p1 = reinterpret_cast<const T*>(in1);
    const T* ptgt = (Op::TARGET == 0) ? p0 : p1;
    // algorithmic traffic: read every input once, write b_new once
    ProfScope prof(ctx, PROF_EW, (double)(Op::NIN + 1) * sizeof(T) * (double)n * (double)batch, 0.0);
    for (size_t inst0 = 0; inst0 < batch; inst0 += cb) {
        const size_t nb = (batch - inst0 < cb) ? batch - inst0 : cb;
        dim3 grid(nblk, (unsigned)nb), block(kBlock);
        if (vec)
            hipLaunchKernelGGL((ew_pass1<T, Op, true>), grid, block, 0, ctx->stream, op, n, inst0, p0, p1, rbuf,
                               partials, pcount, out.mask);
        else
            hipLaunchKernelGGL((ew_pass1<T, Op, false>), grid, block, 0, ctx->stream, op, n, inst0, p0, p1, rbuf,
                               partials, pcount, out.mask);
        if (store_r) {
            if (vec)
                hipLaunchKernelGGL((ew_pass2<T, true>), grid, block, 0, ctx->stream, n, inst0, a_in_d, ptgt, rbuf,
                                   partials, pcount, nblk, o);
            else
                hipLaunchKernelGGL((ew_pass2<T, false>), grid, block, 0, ctx->stream, n, inst0, a_in_d, ptgt, rbuf,
                                   partials, pcount, nblk, o);
        } else {
            if (vec)
                hipLaunchKernelGGL((ew_pass2_recompute<T, Op, true>), grid, block, 0, ctx->stream, op, n, inst0, p0,
                                   p1, partials, pcount, nblk, o);
            else
                hipLaunchKernelGGL((ew_pass2_recompute<T, Op, false>), grid, block, 0, ctx->stream, op, n, inst0, p0,
                                   p1, partials, pcount, nblk, o);
        }
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

#define DISPATCH_DTYPE(ctx, dtype, CALL_F32, CALL_F64)                    \
    do {                                                                  \
        if ((dtype) == TRAMP_F32) return CALL_F32;                        \
        if ((dtype) == TRAMP_F64) return CALL_F64;                        \
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype"); \
    } while (0)

// ---------------------------------------------------------------- C ABI: K1, K2, K4, K5, K7
// ---- producer-fused moments: host side (prior_hint.h)
static void gb_natural(double rho, double mean, double var, double* a0, double* b0, double* eta) {
    // natural parameters of the prior (gauss_bernoulli_prior.py:32-36)
    *a0 = 1.0 / var;
    *b0 = mean / var;
    *eta = 0.5 * (*b0 * *b0 / *a0 + log(2 * tmath::kPi / *a0)) - log(rho / (1 - rho));
}

extern "C" int tramp_prior_hint(tramp_ctx* ctx, int kind, double p0, double p1, double p2, size_t n, size_t batch,
                                const void* b_d, const double* a_d) {
    if (!ctx) return TRAMP_EINVAL;
    ctx->hint.kind = PRIOR_HINT_NONE;
    ctx->hint.ready = false;
    if (kind == PRIOR_HINT_NONE) return TRAMP_OK;
    TRAMP_REQUIRE(ctx, (kind == PRIOR_HINT_GB || kind == PRIOR_HINT_L1) && b_d && a_d && n > 0 && batch > 0);
    static const size_t fuse_env = getenv("TRAMP_PRIOR_FUSE") ? (size_t)atol(getenv("TRAMP_PRIOR_FUSE")) : 1;
    if (!fuse_env) return TRAMP_OK;
    TrampPriorHint& h = ctx->hint;
    if (kind == PRIOR_HINT_GB) {
        TRAMP_REQUIRE(ctx, p0 > 0.0 && p0 < 1.0 && p2 > 0.0);
        gb_natural(p0, p1, p2, &h.p0, &h.p1, &h.p2);
    } else {
        TRAMP_REQUIRE(ctx, p0 >= 0.0);
        h.p0 = p0;
        h.p1 = h.p2 = 0.0;
    }
    h.kind = kind;
    h.buf = b_d;
    h.a_msg = a_d;
    h.y = nullptr;
    h.n = n;
    h.batch = batch;
    h.partials = nullptr;
    h.nblk = 0;
    return TRAMP_OK;
}

// The likelihood twin: "the message (a_d, b_d) I am about to write feeds tramp_sgn_lik_bwd / tramp_abs_lik_bwd with
// the observation y_d".  kind: PRIOR_HINT_SGN_LIK (3) or PRIOR_HINT_ABS_LIK (4); 0 clears.
extern "C" int tramp_lik_hint(tramp_ctx* ctx, int kind, size_t n, size_t batch, const void* b_d, const double* a_d,
                              const void* y_d) {
    if (!ctx) return TRAMP_EINVAL;
    ctx->hint.kind = PRIOR_HINT_NONE;
    ctx->hint.ready = false;
    if (kind == PRIOR_HINT_NONE) return TRAMP_OK;
    TRAMP_REQUIRE(ctx, (kind == PRIOR_HINT_SGN_LIK || kind == PRIOR_HINT_ABS_LIK) && b_d && a_d && y_d && n > 0 && batch > 0);
    static const size_t fuse_env = getenv("TRAMP_PRIOR_FUSE") ? (size_t)atol(getenv("TRAMP_PRIOR_FUSE")) : 1;
    if (!fuse_env) return TRAMP_OK;
    TrampPriorHint& h = ctx->hint;
    h.kind = kind;
    h.p0 = h.p1 = h.p2 = 0.0;
    h.buf = b_d;
    h.a_msg = a_d;
    h.y = y_d;
    h.n = n;
    h.batch = batch;
    h.partials = nullptr;
    h.nblk = 0;
    return TRAMP_OK;
}

// The hint is consumed by the first prior call after it, whether it matches or not.
static const TrampPriorHint* take_hint(tramp_ctx* ctx, TrampPriorHint* copy, int kind, double p0, double p1,
                                       double p2, size_t n, size_t batch, const void* b_d, const double* a_d,
                                       const MsgOut& out, const void* y_d = nullptr) {
    *copy = ctx->hint;
    ctx->hint.kind = PRIOR_HINT_NONE;
    ctx->hint.ready = false;
    const bool ok = copy->ready && copy->kind == kind && copy->p0 == p0 && copy->p1 == p1 && copy->p2 == p2 &&
                    copy->n == n && copy->batch == batch && copy->buf == b_d && copy->a_msg == a_d &&
                    copy->y == y_d && copy->partials && copy->nblk > 0 && !out.count && !out.mask;
    return ok ? copy : nullptr;
}

extern "C" int tramp_gb_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                  const double* ax_d, const void* bx_d, double rho, double mean,
                                  double var, double damping, const double* a_old_d,
                                  const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d,
                                  double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax_d && rho > 0.0 && rho < 1.0 && var > 0.0);
    double a0, b0, eta;
    gb_natural(rho, mean, var, &a0, &b0, &eta);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    TrampPriorHint hc;
    const TrampPriorHint* hint = take_hint(ctx, &hc, PRIOR_HINT_GB, a0, b0, eta, n, batch, bx_d, ax_d, out);
    GbOp<float> of{ax_d, a0, b0, eta};
    GbOp<double> od{ax_d, a0, b0, eta};
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bx_d, nullptr, ax_d, out, hint)),
                   (run_two_pass<double>(ctx, od, n, batch, bx_d, nullptr, ax_d, out, hint)));
}

extern "C" int tramp_l1_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                  const double* ax_d, const void* bx_d, double gamma,
                                  double damping, const double* a_old_d, const void* b_old_d,
                                  double* a_new_d, void* b_new_d, void* r_d, double* v_d,
                                  unsigned long long* count_d, uint8_t* mask_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax_d && gamma >= 0.0);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, count_d, mask_d};
    TrampPriorHint hc;
    const TrampPriorHint* hint = take_hint(ctx, &hc, PRIOR_HINT_L1, gamma, 0.0, 0.0, n, batch, bx_d, ax_d, out);
    L1Op<float> of{ax_d, gamma};
    L1Op<double> od{ax_d, gamma};
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bx_d, nullptr, ax_d, out, hint)),
                   (run_two_pass<double>(ctx, od, n, batch, bx_d, nullptr, ax_d, out, hint)));
}

extern "C" int tramp_binary_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                      const double* ax_d, const void* bx_d, double b0,
                                      double damping, const double* a_old_d, const void* b_old_d,
                                      double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax_d);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    BinaryOp<float> of{ax_d, b0};
    BinaryOp<double> od{ax_d, b0};
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bx_d, nullptr, ax_d, out)),
                   (run_two_pass<double>(ctx, od, n, batch, bx_d, nullptr, ax_d, out)));
}

// natural parameters of a Gaussian mixture (gaussian_mixture_prior.py:36-39)
template <typename Op> static bool gm_fill(Op& op, int K, const double* probs, const double* means, const double* vars) {
    op.K = K;
    double ps = 0.0;
    for (int k = 0; k < K; ++k) {
        if (!(probs[k] > 0.0) || !(vars[k] > 0.0)) return false;
        op.a0[k] = 1.0 / vars[k];
        op.b0[k] = means[k] / vars[k];
        op.logp[k] = log(probs[k]);
        ps += probs[k];
    }
    return ps > 0.0;
}

extern "C" int tramp_halfline_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const double* ax_d,
                                        const void* bx_d, double a0, double b0, double damping, const double* a_old_d,
                                        const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax_d && a0 >= 0.0);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    HalfLinePriorOp<float> of{ax_d, a0, b0};
    HalfLinePriorOp<double> od{ax_d, a0, b0};
    // large batches of large instances: the lean form on the resident pipeline (double arithmetic for f32 messages too),
    // then the tile kernels for the instances it could not finish (an erfcx argument beyond the table - an exponential
    // prior at ax << 1 / mean^2 sits there - or a non-finite message); the count of the previous lean call sends wide
    // messages straight to the tile kernels, as for the two-region channels (pwl_dir)
    if ((dtype == TRAMP_F64 || dtype == TRAMP_F32) && n && batch && bx_d && a_new_d && b_new_d && !r_d &&
        (damping == 0.0 || (a_old_d && b_old_d))) {
        static const size_t adapt = env_size("TRAMP_EW_LEAN_ADAPT", 1);
        if (ctx->lean_stat && ctx->lean_stat[1]) {
            if (adapt && 4u * ctx->lean_stat[0] > ctx->lean_stat[1]) ctx->lean_skip = 15;
            ctx->lean_stat[1] = 0;
        }
        if (ctx->lean_skip > 0) {
            ctx->lean_skip -= 1;
        } else if (dtype == TRAMP_F64) {
            const int rc = run_resident<double>(ctx, HalfLinePriorOp<double, true>{ax_d, a0, b0}, n, batch, bx_d, nullptr, ax_d, out);
            if (rc == TRAMP_OK) return run_tiles<double>(ctx, od, n, batch, bx_d, nullptr, ax_d, out, false, nullptr, true);
            if (rc != kResNotEligible) return rc;
        } else {
            const int rc = run_resident<float>(ctx, HalfLinePriorOp<float, true>{ax_d, a0, b0}, n, batch, bx_d, nullptr, ax_d, out);
            if (rc == TRAMP_OK) return run_tiles<float>(ctx, of, n, batch, bx_d, nullptr, ax_d, out, false, nullptr, true);
            if (rc != kResNotEligible) return rc;
        }
    }
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bx_d, nullptr, ax_d, out)),
                   (run_two_pass<double>(ctx, od, n, batch, bx_d, nullptr, ax_d, out)));
}

template <typename T, int KC>
static int gm_run(tramp_ctx* ctx, int K, const double* probs, const double* means, const double* vars, size_t n, size_t batch,
                  const double* ax_d, const void* bx_d, const MsgOut& out) {
    GmOp<T, KC> op{};
    op.ax = ax_d;
    TRAMP_REQUIRE(ctx, gm_fill(op, K, probs, means, vars));
    return run_two_pass<T>(ctx, op, n, batch, bx_d, nullptr, ax_d, out);
}

template <typename T>
static int gm_by_k(tramp_ctx* ctx, int K, const double* probs, const double* means, const double* vars, size_t n, size_t batch,
                   const double* ax_d, const void* bx_d, const MsgOut& out) {
    switch (K) {
        case 2: return gm_run<T, 2>(ctx, K, probs, means, vars, n, batch, ax_d, bx_d, out);
        case 3: return gm_run<T, 3>(ctx, K, probs, means, vars, n, batch, ax_d, bx_d, out);
        case 4: return gm_run<T, 4>(ctx, K, probs, means, vars, n, batch, ax_d, bx_d, out);
        default: return gm_run<T, 0>(ctx, K, probs, means, vars, n, batch, ax_d, bx_d, out);
    }
}

static int gm_dispatch(tramp_ctx* ctx, int dtype, int K, const double* probs, const double* means, const double* vars, size_t n,
                       size_t batch, const double* ax_d, const void* bx_d, const MsgOut& out) {
    DISPATCH_DTYPE(ctx, dtype, (gm_by_k<float>(ctx, K, probs, means, vars, n, batch, ax_d, bx_d, out)),
                   (gm_by_k<double>(ctx, K, probs, means, vars, n, batch, ax_d, bx_d, out)));
}

extern "C" int tramp_gm_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const double* ax_d,
                                  const void* bx_d, int K, const double* probs_h, const double* means_h,
                                  const double* vars_h, double damping, const double* a_old_d, const void* b_old_d,
                                  double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax_d && probs_h && means_h && vars_h && K >= 1 && K <= tmath::kGmMax);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    return gm_dispatch(ctx, dtype, K, probs_h, means_h, vars_h, n, batch, ax_d, bx_d, out);
}

extern "C" int tramp_sgn_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                 const double* az_d, const void* bz_d, const void* y_d,
                                 double damping, const double* a_old_d, const void* b_old_d,
                                 double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    TrampPriorHint hc;
    const TrampPriorHint* hint = take_hint(ctx, &hc, PRIOR_HINT_SGN_LIK, 0.0, 0.0, 0.0, n, batch, bz_d, az_d, out, y_d);
    SgnLikOp<float> of{az_d};
    SgnLikOp<double> od{az_d};
    // large batches of large instances with no producer-fused moments: the lean form on the resident pipeline, then the
    // tile kernels for the instances it could not finish (a_new = NaN: a disagreement beyond the erfcx table, a non-finite
    // message; ~6 us when there is none)
    if (!hint && n && batch && bz_d && y_d && a_new_d && b_new_d && !r_d && (damping == 0.0 || (a_old_d && b_old_d))) {
        int rc = kResNotEligible;
        if (dtype == TRAMP_F32) rc = run_resident<float>(ctx, SgnLikOp<float, true>{az_d}, n, batch, bz_d, y_d, az_d, out);
        else if (dtype == TRAMP_F64) rc = run_resident<double>(ctx, SgnLikOp<double, true>{az_d}, n, batch, bz_d, y_d, az_d, out);
        if (rc == TRAMP_OK) {
            if (dtype == TRAMP_F32) return run_tiles<float>(ctx, of, n, batch, bz_d, y_d, az_d, out, false, nullptr, true);
            return run_tiles<double>(ctx, od, n, batch, bz_d, y_d, az_d, out, false, nullptr, true);
        }
        if (rc != kResNotEligible) return rc;
    }
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bz_d, y_d, az_d, out, hint)),
                   (run_two_pass<double>(ctx, od, n, batch, bz_d, y_d, az_d, out, hint)));
}

// ---- sign observations packed one bit per component (the resident form of K4 then moves 2T + n/8 bytes)
// One thread per 8 components: a byte of the row (bit j of byte q = component 8 q + j is +1); rows of
// stride = 8 ceil(n / 64) bytes, the padding bits zero.  bad counts the entries that are neither -1 nor +1.
template <typename T>
__global__ __launch_bounds__(kBlock) void sgn_pack_kernel(size_t n, size_t stride, const T* __restrict__ y,
                                                          uint8_t* __restrict__ bits,
                                                          unsigned long long* __restrict__ bad) {
    const size_t b = blockIdx.y;
    const size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned nb = 0;
    if (q < stride) {
        unsigned byte = 0;
        for (int j = 0; j < 8; ++j) {
            const size_t i = q * 8 + j;
            if (i < n) {
                const T v = y[b * n + i];
                if (v == T(1)) byte |= 1u << j;
                else if (!(v == T(-1))) nb += 1;
            }
        }
        bits[b * stride + q] = (uint8_t)byte;
    }
    if (__any(nb != 0) && nb) atomicAdd(bad, (unsigned long long)nb);
}

extern "C" int tramp_sgn_pack(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const void* y_d, void* bits_d,
                              unsigned long long* bad_h) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, y_d && bits_d && bad_h && batch <= 65535);
    *bad_h = 0;
    if (n == 0 || batch == 0) return TRAMP_OK;
    const size_t stride = ((n + 63) / 64) * 8;
    int rc = tramp_ws_reserve(ctx, 256);
    if (rc) return rc;
    unsigned long long* bad_d = reinterpret_cast<unsigned long long*>(ctx->ws);
    TRAMP_HIP_CHECK(ctx, hipMemsetAsync(bad_d, 0, 8, ctx->stream));
    dim3 grid((unsigned)((stride + kBlock - 1) / kBlock), (unsigned)batch);
    if (dtype == TRAMP_F32)
        hipLaunchKernelGGL(sgn_pack_kernel<float>, grid, dim3(kBlock), 0, ctx->stream, n, stride,
                           reinterpret_cast<const float*>(y_d), reinterpret_cast<uint8_t*>(bits_d), bad_d);
    else if (dtype == TRAMP_F64)
        hipLaunchKernelGGL(sgn_pack_kernel<double>, grid, dim3(kBlock), 0, ctx->stream, n, stride,
                           reinterpret_cast<const double*>(y_d), reinterpret_cast<uint8_t*>(bits_d), bad_d);
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(bad_h, bad_d, 8, hipMemcpyDeviceToHost, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_sgn_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const double* az_d,
                                 const void* bz_d, const void* y_d, double damping, const double* a_old_d,
                                 const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d, double* v_d);

extern "C" int tramp_sgn_lik_bwd_bits(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                      const double* az_d, const void* bz_d, const void* y_d, const void* bits_d,
                                      double damping, const double* a_old_d, const void* b_old_d,
                                      double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && y_d);
    // a producer that left the moments (prior_hint.h), shapes the resident pipeline does not cover, a wanted
    // posterior: the update on the unpacked observation
    if (bits_d && !r_d && bz_d && a_new_d && b_new_d && n && batch && (damping == 0.0 || (a_old_d && b_old_d)) &&
        !(ctx->hint.ready && ctx->hint.kind == PRIOR_HINT_SGN_LIK) &&
        resident_covers(ctx, n, batch, dtype == TRAMP_F32 ? 4 : 8)) {
        MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, nullptr, v_d, nullptr, nullptr};
        const size_t stride = ((n + 63) / 64) * 8;
        int rc = kResNotEligible;
        if (dtype == TRAMP_F32) rc = run_resident<float>(ctx, SgnLikBitsOp<float, true>{az_d}, n, batch, bz_d, nullptr, az_d, out, bits_d, stride);
        else if (dtype == TRAMP_F64) rc = run_resident<double>(ctx, SgnLikBitsOp<double, true>{az_d}, n, batch, bz_d, nullptr, az_d, out, bits_d, stride);
        if (rc != kResNotEligible) {
            if (rc == TRAMP_OK) {
                ctx->hint.kind = PRIOR_HINT_NONE;       // consumed, as take_hint does
                ctx->hint.ready = false;
                // the lean form's unfinished instances (a_new = NaN) again, by the tile kernels on the unpacked observation
                if (dtype == TRAMP_F32) return run_tiles<float>(ctx, SgnLikOp<float>{az_d}, n, batch, bz_d, y_d, az_d, out, false, nullptr, true);
                return run_tiles<double>(ctx, SgnLikOp<double>{az_d}, n, batch, bz_d, y_d, az_d, out, false, nullptr, true);
            }
            return rc;
        }
    }
    return tramp_sgn_lik_bwd(ctx, dtype, n, batch, az_d, bz_d, y_d, damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d);
}

extern "C" int tramp_abs_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                 const double* az_d, const void* bz_d, const void* y_d,
                                 double damping, const double* a_old_d, const void* b_old_d,
                                 double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    TrampPriorHint hc;
    const TrampPriorHint* hint = take_hint(ctx, &hc, PRIOR_HINT_ABS_LIK, 0.0, 0.0, 0.0, n, batch, bz_d, az_d, out, y_d);
    AbsLikOp<float> of{az_d};
    AbsLikOp<double> od{az_d};
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bz_d, y_d, az_d, out, hint)),
                   (run_two_pass<double>(ctx, od, n, batch, bz_d, y_d, az_d, out, hint)));
}

extern "C" int tramp_relu_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double slope_neg,
                                  const double* az_d, const void* bz_d, const void* y_d, double damping,
                                  const double* a_old_d, const void* b_old_d, double* a_new_d,
                                  void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && slope_neg >= 0.0);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    ReluLikOp<float> of{az_d, slope_neg};
    ReluLikOp<double> od{az_d, slope_neg};
    // Relu proper at large shapes: the lean form on the resident pipeline, then the tile kernels for the instances it could
    // not finish (as tramp_sgn_lik_bwd)
    if (slope_neg == 0.0 && n && batch && bz_d && y_d && a_new_d && b_new_d && !r_d && (damping == 0.0 || (a_old_d && b_old_d))) {
        int rc = kResNotEligible;
        if (dtype == TRAMP_F32) rc = run_resident<float>(ctx, ReluLikOp<float, true>{az_d, 0.0}, n, batch, bz_d, y_d, az_d, out);
        else if (dtype == TRAMP_F64) rc = run_resident<double>(ctx, ReluLikOp<double, true>{az_d, 0.0}, n, batch, bz_d, y_d, az_d, out);
        if (rc == TRAMP_OK) {
            if (dtype == TRAMP_F32) return run_tiles<float>(ctx, of, n, batch, bz_d, y_d, az_d, out, true, nullptr, true);
            return run_tiles<double>(ctx, od, n, batch, bz_d, y_d, az_d, out, true, nullptr, true);
        }
        if (rc != kResNotEligible) return rc;
    }
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bz_d, y_d, az_d, out)),
                   (run_two_pass<double>(ctx, od, n, batch, bz_d, y_d, az_d, out)));
}

template <typename T, int KIND>
static int pwl_dir(tramp_ctx* ctx, int dir, size_t n, size_t batch, const double* az_d,
                   const void* bz_d, const double* ax_d, const void* bx_d, const MsgOut& out,
                   const tmath::PwlSpec& spec = tmath::PwlSpec{}) {
    // Sgn / Abs / Relu at large shapes: the resident single-pass pipeline (ew_resident_pwl.hip: 512 threads x 2 pieces, the
    // 13 KiB one-value erfcx table beside the parked r).
    static const size_t res_k7 = env_size("TRAMP_EW_RESIDENT_K7", 1);
    if constexpr (KIND < 3) {
        if (res_k7 && n && batch && bz_d && bx_d && az_d && ax_d && out.a_new && out.b_new &&
            (out.damping == 0.0 || (out.a_old && out.b_old))) {
            // The lean form covers erfcx arguments in [-7.5, 8) (|b| / sqrt(a) up to ~10: moderate signal-to-noise messages); an instance with a
            // component outside comes back with a_new = NaN and is redone by the tile kernels (run_tiles: redo_nan).  Wide
            // messages flag most instances and would pay both: the count of the PREVIOUS lean call (pinned host word, read
            // without synchronising - a heuristic, either path is exact) sends the next 15 calls straight to the tile kernels.
            static const size_t adapt = env_size("TRAMP_EW_LEAN_ADAPT", 1);
            if (ctx->lean_stat && ctx->lean_stat[1]) {
                if (adapt && 4u * ctx->lean_stat[0] > ctx->lean_stat[1]) ctx->lean_skip = 15;
                ctx->lean_stat[1] = 0;
            }
            if (ctx->lean_skip > 0) {
                ctx->lean_skip -= 1;
            } else {
                int rc;
                if (dir == TRAMP_FWD) rc = run_resident<T>(ctx, PwlOp<T, KIND, true, true, true>{az_d, ax_d, spec}, n, batch, bz_d, bx_d, ax_d, out);
                else rc = run_resident<T>(ctx, PwlOp<T, KIND, false, true, true>{az_d, ax_d, spec}, n, batch, bz_d, bx_d, az_d, out);
                if (rc == TRAMP_OK) {
                    // (resident_covers implies whole 16-byte pieces per instance: the tile kernels apply)
                    if (dir == TRAMP_FWD) return run_tiles<T>(ctx, PwlOp<T, KIND, true>{az_d, ax_d, spec}, n, batch, bz_d, bx_d, ax_d, out, true, nullptr, true);
                    return run_tiles<T>(ctx, PwlOp<T, KIND, false>{az_d, ax_d, spec}, n, batch, bz_d, bx_d, az_d, out, true, nullptr, true);
                }
                if (rc != kResNotEligible) return rc;
            }
        }
    }
    if (dir == TRAMP_FWD) {
        PwlOp<T, KIND, true> op{az_d, ax_d, spec};
        return run_two_pass<T>(ctx, op, n, batch, bz_d, bx_d, ax_d, out);
    }
    PwlOp<T, KIND, false> op{az_d, ax_d, spec};
    return run_two_pass<T>(ctx, op, n, batch, bz_d, bx_d, az_d, out);
}

template <typename T>
static int pwl_kind(tramp_ctx* ctx, int kind, int dir, size_t n, size_t batch, const double* az_d,
                    const void* bz_d, const double* ax_d, const void* bx_d, const MsgOut& out) {
    switch (kind) {
        case TRAMP_SGN: return pwl_dir<T, 0>(ctx, dir, n, batch, az_d, bz_d, ax_d, bx_d, out);
        case TRAMP_ABS: return pwl_dir<T, 1>(ctx, dir, n, batch, az_d, bz_d, ax_d, bx_d, out);
        case TRAMP_RELU: return pwl_dir<T, 2>(ctx, dir, n, batch, az_d, bz_d, ax_d, bx_d, out);
    }
    return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "kind");
}

extern "C" int tramp_pwl_channel(tramp_ctx* ctx, int dtype, int kind, int dir, size_t n,
                                 size_t batch, const double* az_d, const void* bz_d,
                                 const double* ax_d, const void* bx_d, double damping,
                                 const double* a_old_d, const void* b_old_d, double* a_new_d,
                                 void* b_new_d, void* r_d, double* v_d, uint8_t* mask_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && ax_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, mask_d};
    DISPATCH_DTYPE(ctx, dtype,
                   (pwl_kind<float>(ctx, kind, dir, n, batch, az_d, bz_d, ax_d, bx_d, out)),
                   (pwl_kind<double>(ctx, kind, dir, n, batch, az_d, bz_d, ax_d, bx_d, out)));
}

extern "C" int tramp_pwl2_channel(tramp_ctx* ctx, int dtype, int dir, size_t n, size_t batch,
                                  double slope_pos, double slope_neg, double x0_pos, double x0_neg,
                                  double zcut, const double* az_d, const void* bz_d, const double* ax_d,
                                  const void* bx_d, double damping, const double* a_old_d,
                                  const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d,
                                  double* v_d, uint8_t* mask_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && ax_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, mask_d};
    tmath::PwlSpec spec{{slope_pos, slope_neg}, {x0_pos, x0_neg}, zcut};
    DISPATCH_DTYPE(ctx, dtype,
                   (pwl_dir<float, 3>(ctx, dir, n, batch, az_d, bz_d, ax_d, bx_d, out, spec)),
                   (pwl_dir<double, 3>(ctx, dir, n, batch, az_d, bz_d, ax_d, bx_d, out, spec)));
}

template <typename T>
static int pwl3_dir(tramp_ctx* ctx, int dir, size_t n, size_t batch, const double* az_d, const void* bz_d,
                    const double* ax_d, const void* bx_d, const MsgOut& out, const tmath::Pwl3Spec& spec) {
    if (dir == TRAMP_FWD) {
        Pwl3Op<T, true> op{az_d, ax_d, spec};
        return run_two_pass<T>(ctx, op, n, batch, bz_d, bx_d, ax_d, out);
    }
    Pwl3Op<T, false> op{az_d, ax_d, spec};
    return run_two_pass<T>(ctx, op, n, batch, bz_d, bx_d, az_d, out);
}

extern "C" int tramp_pwl3_channel(tramp_ctx* ctx, int dtype, int dir, size_t n, size_t batch, double z_lo,
                                  double z_hi, const double* slopes3, const double* x03, const double* az_d,
                                  const void* bz_d, const double* ax_d, const void* bx_d, double damping,
                                  const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d,
                                  void* r_d, double* v_d, uint8_t* mask_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && ax_d && slopes3 && x03 && z_lo < z_hi);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, mask_d};
    tmath::Pwl3Spec spec{z_lo, z_hi, {slopes3[0], slopes3[1], slopes3[2]}, {x03[0], x03[1], x03[2]}};
    DISPATCH_DTYPE(ctx, dtype, (pwl3_dir<float>(ctx, dir, n, batch, az_d, bz_d, ax_d, bx_d, out, spec)),
                   (pwl3_dir<double>(ctx, dir, n, batch, az_d, bz_d, ax_d, bx_d, out, spec)));
}

extern "C" int tramp_pwl3_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double z_lo, double z_hi,
                                  const double* slopes3, const double* x03, const double* az_d, const void* bz_d,
                                  const void* y_d, double damping, const double* a_old_d, const void* b_old_d,
                                  double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    // z_lo == z_hi with a SLOPED middle region: that region is empty (AsymmetricAbsLikelihood's two half lines); a
    // flat one needs a proper interval for its truncated normal.  Descending branches enter with log|slope|.
    TRAMP_REQUIRE(ctx, az_d && slopes3 && x03 && (z_lo < z_hi || (z_lo == z_hi && slopes3[1] != 0.0)));
    MsgOut out{damping, a_old_d, b_old_d, a_new_d, b_new_d, r_d, v_d, nullptr, nullptr};
    tmath::Pwl3Spec spec{z_lo, z_hi, {slopes3[0], slopes3[1], slopes3[2]}, {x03[0], x03[1], x03[2]}};
    Pwl3LikOp<float> of{az_d, spec};
    Pwl3LikOp<double> od{az_d, spec};
    DISPATCH_DTYPE(ctx, dtype, (run_two_pass<float>(ctx, of, n, batch, bz_d, y_d, az_d, out)),
                   (run_two_pass<double>(ctx, od, n, batch, bz_d, y_d, az_d, out)));
}

// ---------------------------------------------------------------- K3 / K6 constant messages
template <typename T>
__global__ __launch_bounds__(kBlock) void scale_fill_kernel(size_t count, const T* __restrict__ y,
                                                            T scale, T value, T* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        out[i] = y ? y[i] / scale : value;     // y / var as the reference (gaussian_likelihood.py:17)
}

__global__ void fill_f64_kernel(size_t count, double value, double* out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        out[i] = value;
}

static inline int grid_for(size_t count) {
    size_t g = (count + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int tramp_fill_f64(tramp_ctx* ctx, double* dst_d, size_t count, double value) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, dst_d || count == 0);
    if (!count) return TRAMP_OK;
    hipLaunchKernelGGL(fill_f64_kernel, dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream, count, value, dst_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_fill(tramp_ctx* ctx, int dtype, void* dst_d, size_t count, double value) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, dst_d || count == 0);
    if (!count) return TRAMP_OK;
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((scale_fill_kernel<double>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream,
                           count, (const double*)nullptr, 1.0, value, (double*)dst_d);
    else if (dtype == TRAMP_F32)
        hipLaunchKernelGGL((scale_fill_kernel<float>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream,
                           count, (const float*)nullptr, 1.0f, (float)value, (float*)dst_d);
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_gauss_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double mean,
                                     double var, double* a_new_d, void* b_new_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, var > 0.0 && a_new_d && b_new_d);
    int rc = tramp_fill_f64(ctx, a_new_d, batch, 1.0 / var);
    if (rc) return rc;
    return tramp_fill(ctx, dtype, b_new_d, n * batch, mean / var);
}

extern "C" int tramp_gauss_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                   const void* y_d, double var, double* a_new_d, void* b_new_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, var > 0.0 && a_new_d && b_new_d && y_d);
    int rc = tramp_fill_f64(ctx, a_new_d, batch, 1.0 / var);
    if (rc) return rc;
    const size_t count = n * batch;
    if (!count) return TRAMP_OK;
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((scale_fill_kernel<double>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream,
                           count, (const double*)y_d, var, 0.0, (double*)b_new_d);
    else if (dtype == TRAMP_F32)
        hipLaunchKernelGGL((scale_fill_kernel<float>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream,
                           count, (const float*)y_d, (float)var, 0.0f, (float*)b_new_d);
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- diagonal beliefs (K1d..K7d)
// isotropic=False: the precision a is an array like b, the posterior variance stays per component and
// compute_ab_new clips element-wise (tramp/base.py:250-255) - one independent component per thread, no reduction.
// The mode every factor-level test of the reference runs in (tests/test_priors.py:17-51, test_likelihoods.py:22-83).
struct DiagParams { double p[32]; };

template <typename T>
__global__ __launch_bounds__(kBlock) void diag_kernel(int which, size_t count, const T* __restrict__ a,
                                                      const T* __restrict__ b, const T* __restrict__ in2,
                                                      DiagParams P,
                                                      T* __restrict__ a_new, T* __restrict__ b_new,
                                                      T* __restrict__ r_out, T* __restrict__ v_out) {
    const double p0 = P.p[0], p1 = P.p[1], p2 = P.p[2];
    double a0 = 0, b0 = 0, eta = 0;
    if (which == TRAMP_DIAG_GB) {
        a0 = 1.0 / p2;
        b0 = p1 / p2;
        eta = 0.5 * (b0 * b0 / a0 + log(2 * tmath::kPi / a0)) - log(p0 / (1 - p0));
    } else if (which == TRAMP_DIAG_GAUSS_PRIOR) {          // p0 = mean, p1 = var (gaussian_prior.py:13-17)
        a0 = 1.0 / p1;
        b0 = p0 / p1;
    } else if (which == TRAMP_DIAG_GAUSS_LIK) {            // p0 = var (gaussian_likelihood.py:13-17)
        a0 = 1.0 / p0;
    }
    tmath::Pwl3Spec spec{p0, p1, {P.p[2], P.p[3], P.p[4]}, {P.p[5], P.p[6], P.p[7]}};
    // TRAMP_DIAG_GM: p = {K, probs[K], means[K], vars[K]} (gaussian_mixture_prior.py:36-39)
    const int gK = which == TRAMP_DIAG_GM ? (int)p0 : 0;
    double ga0[tmath::kGmMax], gb0[tmath::kGmMax], glogp[tmath::kGmMax];
#pragma unroll
    for (int k = 0; k < tmath::kGmMax; ++k) {
        ga0[k] = gb0[k] = glogp[k] = 0.0;
        if (k < gK) {
            ga0[k] = 1.0 / P.p[1 + 2 * gK + k];
            gb0[k] = P.p[1 + gK + k] / P.p[1 + 2 * gK + k];
            glogp[k] = log(P.p[1 + k]);
        }
    }
    tmath::erfcx_setup<T>();                               // the half-line moments read erfcx from the LDS table
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock) {
        T ai = a[i], bi = b[i];
        RV<T> o;
        switch (which) {
            case TRAMP_DIAG_GB: {
                T at = ai + T(a0);
                o = tmath::gb_posterior<T>(bi, at, tmath::log_(T(2 * tmath::kPi) / at), T(b0), T(eta));
                break;
            }
            case TRAMP_DIAG_L1:
                o = tmath::l1_posterior<T>(bi, T(1) / ai, T(p0));
                break;
            case TRAMP_DIAG_SGN_LIK:
                o = tmath::sgn_lik_posterior<T>(bi, in2[i], ai, T(1) / ai);
                break;
            case TRAMP_DIAG_ABS_LIK:
                o = tmath::abs_lik_posterior<T>(bi, in2[i]);
                break;
            case TRAMP_DIAG_GAUSS_PRIOR: {                 // gaussian_prior.py:63-68
                const T at = ai + T(a0);
                o.r = (bi + T(b0)) / at;
                o.v = T(1) / at;
                break;
            }
            case TRAMP_DIAG_GAUSS_LIK: {                   // gaussian_likelihood.py:43-49
                const T at = ai + T(a0);
                o.r = (bi + T(a0) * in2[i]) / at;
                o.v = T(1) / at;
                break;
            }
            case TRAMP_DIAG_BINARY: {                      // p0 = b0 (binary_prior.py:27,52-58; beliefs/binary.py:8-13)
                const T t = tmath::tanh_(bi + T(p0));
                o.r = t;
                o.v = T(1) - t * t;
                break;
            }
            case TRAMP_DIAG_RELU_LIK:                      // p0 = slope of the negative side (0: Relu)
                o = tmath::relu_lik_posterior<T>(bi, in2[i], T(p0), tmath::region_const<T>((double)ai));
                break;
            case TRAMP_DIAG_HALFLINE: {                    // p0 = a0, p1 = b0 (exponential_prior.py:54-60, positive_prior.py:56-63)
                const T at = ai + T(p0);
                o = tmath::sgn_lik_posterior<T>(bi + T(p1), T(1), at, T(1) / at);
                break;
            }
            case TRAMP_DIAG_GM: {                          // gaussian_mixture_prior.py:88-97 with a per-component precision
                tmath::GmConst<T> g;
                g.K = gK;
#pragma unroll
                for (int k = 0; k < tmath::kGmMax; ++k) {
                    if (k < gK) tmath::gm_const_k<T>(g, k, (double)ai, glogp[k], ga0[k], gb0[k]);
                }
                o = tmath::gm_posterior<T>(bi, g);
                break;
            }
            default: {                                     // TRAMP_DIAG_PWL3_LIK
                tmath::PwlOut<T> w = tmath::pwl3_lik_posterior<T>(bi, in2[i], ai, spec);
                o.r = w.r;
                o.v = w.v;
            }
        }
        // element-wise compute_ab_new (tramp/base.py:250-255)
        double an = tmath::a_new_of((double)o.v, (double)ai);
        T vinv = ai + T(an);
        if (a_new) a_new[i] = T(an);
        if (b_new) b_new[i] = o.r * vinv - bi;
        if (r_out) r_out[i] = o.r;
        if (v_out) v_out[i] = o.v;
    }
}

extern "C" int tramp_diag_factor_p(tramp_ctx* ctx, int dtype, int which, size_t count, const void* a_d,
                                   const void* b_d, const void* in2_d, const double* params, int nparams,
                                   void* a_new_d, void* b_new_d, void* r_d, void* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, which >= 0 && which <= TRAMP_DIAG_GM && a_d && b_d && nparams >= 0 && nparams <= 32);
    TRAMP_REQUIRE(ctx, nparams == 0 || params);
    const bool needs_y = which == TRAMP_DIAG_SGN_LIK || which == TRAMP_DIAG_ABS_LIK || which == TRAMP_DIAG_GAUSS_LIK ||
                         which == TRAMP_DIAG_RELU_LIK || which == TRAMP_DIAG_PWL3_LIK;
    TRAMP_REQUIRE(ctx, !needs_y || in2_d);
    static const int need[] = {3, 1, 0, 0, 2, 1, 1, 1, 8, 2, 4};
    TRAMP_REQUIRE(ctx, nparams >= need[which]);
    DiagParams P{};
    for (int i = 0; i < nparams; ++i) P.p[i] = params[i];
    if (which == TRAMP_DIAG_HALFLINE) TRAMP_REQUIRE(ctx, P.p[0] >= 0.0);
    if (which == TRAMP_DIAG_GM) {
        const int K = (int)P.p[0];
        TRAMP_REQUIRE(ctx, K >= 1 && K <= tmath::kGmMax && nparams >= 1 + 3 * K);
        for (int k = 0; k < K; ++k) TRAMP_REQUIRE(ctx, P.p[1 + k] > 0.0 && P.p[1 + 2 * K + k] > 0.0);
    }
    if (which == TRAMP_DIAG_GB) TRAMP_REQUIRE(ctx, P.p[0] > 0.0 && P.p[0] < 1.0 && P.p[2] > 0.0);
    if (which == TRAMP_DIAG_GAUSS_PRIOR) TRAMP_REQUIRE(ctx, P.p[1] > 0.0);
    if (which == TRAMP_DIAG_GAUSS_LIK) TRAMP_REQUIRE(ctx, P.p[0] > 0.0);
    if (which == TRAMP_DIAG_RELU_LIK) TRAMP_REQUIRE(ctx, P.p[0] >= 0.0);
    if (which == TRAMP_DIAG_PWL3_LIK)
        TRAMP_REQUIRE(ctx, P.p[0] < P.p[1] || (P.p[0] == P.p[1] && P.p[3] != 0.0));     // as tramp_pwl3_lik_bwd
    if (!count) return TRAMP_OK;
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((diag_kernel<double>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream, which,
                           count, (const double*)a_d, (const double*)b_d, (const double*)in2_d, P,
                           (double*)a_new_d, (double*)b_new_d, (double*)r_d, (double*)v_d);
    else if (dtype == TRAMP_F32)
        hipLaunchKernelGGL((diag_kernel<float>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream, which,
                           count, (const float*)a_d, (const float*)b_d, (const float*)in2_d, P,
                           (float*)a_new_d, (float*)b_new_d, (float*)r_d, (float*)v_d);
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_diag_factor(tramp_ctx* ctx, int dtype, int which, size_t count, const void* a_d,
                                 const void* b_d, const void* in2_d, double p0, double p1, double p2,
                                 void* a_new_d, void* b_new_d, void* r_d, void* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, which >= 0 && which <= TRAMP_DIAG_ABS_LIK);
    const double params[3] = {p0, p1, p2};
    return tramp_diag_factor_p(ctx, dtype, which, count, a_d, b_d, in2_d, params, 3, a_new_d, b_new_d, r_d, v_d);
}

// ---------------------------------------------------------------- K11 variable posterior + norms
// One streaming pass (no global dependency): r, and per-block partial sums folded
// by a second tiny kernel.
template <typename T>
__global__ __launch_bounds__(kBlock) void var_post_kernel(size_t n, const double* __restrict__ a1,
                                                          const T* __restrict__ b1,
                                                          const double* __restrict__ a2,
                                                          const T* __restrict__ b2,
                                                          const T* __restrict__ r_old, T* __restrict__ r,
                                                          double* __restrict__ v,
                                                          double* __restrict__ partials) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    // Variable.posterior_ab: python sum() starts from 0 (tramp/base.py:152-155)
    const double a_hat = 0.0 + a1[b] + a2[b];
    const T ah = T(a_hat);
    const T* p1 = b1 + b * n;
    const T* p2 = b2 + b * n;
    const T* po = r_old ? r_old + b * n : nullptr;
    T* pr = r + b * n;
    double sd = 0.0, sn = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        T ri = (p1[i] + p2[i]) / ah;
        T old = po ? po[i] : T(0);
        double dd = (double)ri - (double)old;
        sd += dd * dd;
        sn += (double)ri * (double)ri;
        pr[i] = ri;
    }
    sd = block_sum(sd, smem);
    sn = block_sum(sn, smem);
    if (threadIdx.x == 0) {
        partials[(b * gridDim.x + blockIdx.x) * 2 + 0] = sd;
        partials[(b * gridDim.x + blockIdx.x) * 2 + 1] = sn;
        if (blockIdx.x == 0 && v) v[b] = 1.0 / a_hat;
    }
}

// Streaming-tile form of the same pass (n a multiple of the 16-byte vector): a block owns `tpb` consecutive
// tiles of 256 x U vectors (launch shape of tools/probes/probe_copy.hip), 16-byte loads and stores.
// NTS: arrays past the Infinity Cache - loads and stores that do not allocate in the caches (the three input streams and
// the output are each read / written once: 385 -> 3xx us at 64 x 1e6 f64 with the stopping rule's r_old)
template <typename T, int U, bool NTS>
__global__ __launch_bounds__(kBlock) void var_post_tile(size_t n, int tpb, const double* __restrict__ a1,
                                                        const T* __restrict__ b1, const double* __restrict__ a2,
                                                        const T* __restrict__ b2, const T* __restrict__ r_old,
                                                        T* __restrict__ r, double* __restrict__ v,
                                                        double* __restrict__ partials) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const double a_hat = 0.0 + a1[b] + a2[b];
    const T ah = T(a_hat);
    const size_t nv = n / VN;
    const V* p1 = reinterpret_cast<const V*>(b1 + b * n);
    const V* p2 = reinterpret_cast<const V*>(b2 + b * n);
    const V* po = r_old ? reinterpret_cast<const V*>(r_old + b * n) : nullptr;
    V* pr = reinterpret_cast<V*>(r + b * n);
    double sd = 0.0, sn = 0.0;
    for (int tile = 0; tile < tpb; ++tile) {
        const size_t base = ((size_t)blockIdx.x * tpb + tile) * kBlock * U + threadIdx.x;
        if (base - threadIdx.x >= nv) break;
        V x1[U], x2[U], xo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                if (NTS) {
                    using NVt = typename NativeVec<T>::type;
                    *reinterpret_cast<NVt*>(&x1[u]) = __builtin_nontemporal_load(reinterpret_cast<const NVt*>(p1 + i));
                    *reinterpret_cast<NVt*>(&x2[u]) = __builtin_nontemporal_load(reinterpret_cast<const NVt*>(p2 + i));
                    if (po) *reinterpret_cast<NVt*>(&xo[u]) = __builtin_nontemporal_load(reinterpret_cast<const NVt*>(po + i));
                } else {
                    x1[u] = p1[i];
                    x2[u] = p2[i];
                    if (po) xo[u] = po[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                V out;
#pragma unroll
                for (int j = 0; j < VN; ++j) {
                    const T ri = (vget<T>(x1[u], j) + vget<T>(x2[u], j)) / ah;
                    const T old = po ? vget<T>(xo[u], j) : T(0);
                    const double dd = (double)ri - (double)old;
                    sd += dd * dd;
                    sn += (double)ri * (double)ri;
                    vset<T>(out, j, ri);
                }
                store_vec<T, NTS>(&pr[i], out);
            }
        }
    }
    sd = block_sum(sd, smem);
    sn = block_sum(sn, smem);
    if (threadIdx.x == 0) {
        partials[(b * gridDim.x + blockIdx.x) * 2 + 0] = sd;
        partials[(b * gridDim.x + blockIdx.x) * 2 + 1] = sn;
        if (blockIdx.x == 0 && v) v[b] = 1.0 / a_hat;
    }
}

// tiles per block of a streaming pass: enough blocks to fill the chip several times over
static inline int tiles_per_block(size_t tiles, size_t batch) {
    int tpb = 8;
    while (tpb > 1 && (tiles / tpb) * batch < 4096) tpb >>= 1;
    return tpb;
}

// out_q[b] = sum_k partials[(b*nblk + k)*nq + q]; one block per instance.
__global__ __launch_bounds__(kBlock) void fold_partials_kernel(const double* __restrict__ partials, int nblk,
                                                               int nq, double* out0, double* out1) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    for (int q = 0; q < nq; ++q) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nblk; i += kBlock) s += partials[(b * nblk + i) * nq + q];
        s = block_sum(s, smem);
        double* o = q == 0 ? out0 : out1;
        if (threadIdx.x == 0 && o) o[b] = s;
    }
}

extern "C" int tramp_variable_posterior(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                                        const double* a1_d, const void* b1_d, const double* a2_d,
                                        const void* b2_d, const void* r_old_d, void* r_d, double* v_d,
                                        double* sq_diff_d, double* sq_norm_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, a1_d && b1_d && a2_d && b2_d && r_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    if (!n || !batch) return TRAMP_OK;
    TRAMP_REQUIRE(ctx, batch <= 65535);
    const size_t vn = dtype == TRAMP_F64 ? 2 : 4;
    const bool tiled = (n % vn) == 0;
    constexpr int U = kTileU;
    const size_t tiles = (n / vn + (size_t)kBlock * U - 1) / ((size_t)kBlock * U);
    const int tpb = tiles_per_block(tiles, batch);
    const int nblk = tiled ? (int)((tiles + tpb - 1) / tpb) : blocks_for(n, 1);
    int rc = tramp_ws_reserve(ctx, batch * (size_t)nblk * 2 * sizeof(double));
    if (rc) return rc;
    dim3 grid(nblk, (unsigned)batch);
    ProfScope prof(ctx, PROF_VAR_POST, 3.0 * dtype_size(dtype) * (double)n * (double)batch, 0.0);
    const bool nts = (double)n * (double)batch * (double)dtype_size(dtype) > 192.0 * 1024 * 1024;
    if (tiled && dtype == TRAMP_F64 && nts)
        hipLaunchKernelGGL((var_post_tile<double, U, true>), grid, dim3(kBlock), 0, ctx->stream, n, tpb, a1_d,
                           (const double*)b1_d, a2_d, (const double*)b2_d, (const double*)r_old_d, (double*)r_d, v_d,
                           ctx->ws);
    else if (tiled && dtype == TRAMP_F64)
        hipLaunchKernelGGL((var_post_tile<double, U, false>), grid, dim3(kBlock), 0, ctx->stream, n, tpb, a1_d,
                           (const double*)b1_d, a2_d, (const double*)b2_d, (const double*)r_old_d, (double*)r_d, v_d,
                           ctx->ws);
    else if (tiled && nts)
        hipLaunchKernelGGL((var_post_tile<float, U, true>), grid, dim3(kBlock), 0, ctx->stream, n, tpb, a1_d,
                           (const float*)b1_d, a2_d, (const float*)b2_d, (const float*)r_old_d, (float*)r_d, v_d,
                           ctx->ws);
    else if (tiled)
        hipLaunchKernelGGL((var_post_tile<float, U, false>), grid, dim3(kBlock), 0, ctx->stream, n, tpb, a1_d,
                           (const float*)b1_d, a2_d, (const float*)b2_d, (const float*)r_old_d, (float*)r_d, v_d,
                           ctx->ws);
    else if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((var_post_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, a1_d,
                           (const double*)b1_d, a2_d, (const double*)b2_d, (const double*)r_old_d,
                           (double*)r_d, v_d, ctx->ws);
    else
        hipLaunchKernelGGL((var_post_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, a1_d,
                           (const float*)b1_d, a2_d, (const float*)b2_d, (const float*)r_old_d,
                           (float*)r_d, v_d, ctx->ws);
    if (sq_diff_d || sq_norm_d)
        hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, ctx->ws,
                           nblk, 2, sq_diff_d, sq_norm_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- L21 (group-sparsity) MAP prior
// MAP_L21NormPrior (tramp/priors/map_L21_norm_prior.py:7-20,69-74), the prior of the total-variation model
// (models/total_variation_model.py:28-40): groups run along axis 0 of a [d, n] array (d components per site,
// stride n).  With norm_j = ||b[:, j]||_2:
//     r[c, j] = max(0, 1 - gamma / norm_j) b[c, j] / a,
//     v[c, j] = (norm_j > gamma) (1 + (b[c, j]^2 / norm_j^2 - 1) gamma / norm_j) / a,   v = mean over all d n.
// The expressions are kept in the reference's order (an all-zero group gives 0 * nan = nan there too).
// Two passes like the other priors: pass 1 sums v and counts the supported sites, pass 2 recomputes r and
// writes the message.  One thread per site, the d loads of a site are each coalesced across the wave.
template <typename T>
__global__ __launch_bounds__(kBlock) void l21_pass1_kernel(int d, size_t n, double gamma, const T* __restrict__ bx,
                                                           double* __restrict__ partials,
                                                           unsigned char* __restrict__ mask) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const T* base = bx + b * (size_t)d * n;
    double sv = 0.0, cnt = 0.0;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (size_t)gridDim.x * kBlock) {
        double ss = 0.0;
        for (int c = 0; c < d; ++c) {
            const double x = (double)base[(size_t)c * n + j];
            ss += x * x;
        }
        const double nrm = sqrt(ss);
        const double on = nrm > gamma ? 1.0 : 0.0;
        for (int c = 0; c < d; ++c) {
            const double x = (double)base[(size_t)c * n + j];
            sv += on * (1.0 + (x * x / (nrm * nrm) - 1.0) * gamma / nrm);
        }
        cnt += on;
        if (mask) mask[b * n + j] = nrm > gamma ? 1 : 0;
    }
    sv = block_sum(sv, smem);
    __syncthreads();
    cnt = block_sum(cnt, smem);
    if (threadIdx.x == 0) {
        partials[(b * gridDim.x + blockIdx.x) * 2 + 0] = sv;
        partials[(b * gridDim.x + blockIdx.x) * 2 + 1] = cnt;
    }
}

__global__ __launch_bounds__(kBlock) void l21_finish_kernel(const double* __restrict__ partials, int nblk, double dn,
                                                            const double* __restrict__ ax, double damping,
                                                            const double* a_old, double* a_new, double* a_raw,
                                                            double* v_out, unsigned long long* count) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) {
        s += partials[(b * nblk + i) * 2 + 0];
        c += partials[(b * nblk + i) * 2 + 1];
    }
    s = block_sum(s, smem);
    __syncthreads();
    c = block_sum(c, smem);
    if (threadIdx.x != 0) return;
    const double a = ax[b];
    const double v = (1.0 / a) * (s / dn);
    const double an = tmath::a_new_of(v, a);
    a_raw[b] = an;
    if (a_new) a_new[b] = (damping != 0.0) ? damping * a_old[b] + (1.0 - damping) * an : an;
    if (v_out) v_out[b] = v;
    if (count) count[b] = (unsigned long long)c;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void l21_pass2_kernel(int d, size_t n, double gamma, const double* __restrict__ ax,
                                                           const double* __restrict__ a_raw, const T* bx,
                                                           double damping, const T* b_old, T* b_new, T* r_out) {
    const size_t b = blockIdx.y;
    const double a = ax[b];
    const double asum = a + a_raw[b];
    const size_t off = b * (size_t)d * n;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (size_t)gridDim.x * kBlock) {
        double ss = 0.0;
        for (int c = 0; c < d; ++c) {
            const double x = (double)bx[off + (size_t)c * n + j];
            ss += x * x;
        }
        const double nrm = sqrt(ss);
        const double shrink = fmax(0.0, 1.0 - gamma / nrm);
        for (int c = 0; c < d; ++c) {
            const size_t i = off + (size_t)c * n + j;
            const double x = (double)bx[i];
            const double r = (1.0 / a) * (shrink * x);
            double bn = r * asum - x;
            if (damping != 0.0) bn = damping * (double)b_old[i] + (1.0 - damping) * bn;
            if (b_new) b_new[i] = (T)bn;
            if (r_out) r_out[i] = (T)r;
        }
    }
}

extern "C" int tramp_l21_prior_fwd(tramp_ctx* ctx, int dtype, int d, size_t n, size_t batch, double gamma,
                                   const double* ax_d, const void* bx_d, double damping, const double* a_old_d,
                                   const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d, double* v_d,
                                   unsigned long long* count_d, unsigned char* mask_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax_d && bx_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, d >= 1);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!n || !batch) return TRAMP_OK;
    TRAMP_REQUIRE(ctx, batch <= 65535);
    const int nblk = blocks_for(n, 1);
    int rc = tramp_ws_reserve(ctx, batch * ((size_t)nblk * 2 + 1) * sizeof(double));
    if (rc) return rc;
    double* partials = ctx->ws;
    double* a_raw = ctx->ws + batch * (size_t)nblk * 2;
    dim3 grid(nblk, (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((l21_pass1_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, d, n, gamma,
                           (const double*)bx_d, partials, mask_d);
    else
        hipLaunchKernelGGL((l21_pass1_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, d, n, gamma,
                           (const float*)bx_d, partials, mask_d);
    hipLaunchKernelGGL(l21_finish_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, partials, nblk,
                       (double)d * (double)n, ax_d, damping, a_old_d, a_new_d, a_raw, v_d, count_d);
    if (b_new_d || r_d) {
        if (dtype == TRAMP_F64)
            hipLaunchKernelGGL((l21_pass2_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, d, n, gamma, ax_d, a_raw,
                               (const double*)bx_d, damping, (const double*)b_old_d, (double*)b_new_d, (double*)r_d);
        else
            hipLaunchKernelGGL((l21_pass2_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, d, n, gamma, ax_d, a_raw,
                               (const float*)bx_d, damping, (const float*)b_old_d, (float*)b_new_d, (float*)r_d);
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- K12 damping, distances
template <typename T>
__global__ __launch_bounds__(kBlock) void damp_kernel(size_t n, double d, const double* __restrict__ a_old,
                                                      const T* __restrict__ b_old, double* a, T* bb,
                                                      const double* a_src = nullptr, const T* b_src = nullptr) {
    const size_t b = blockIdx.y;
    const T dT = T(d), omd = T(1.0 - d);
    const T* src = b_src ? b_src : bb;                 // three-operand form (tramp_damp_to): the second term comes from src
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        bb[b * n + i] = dT * b_old[b * n + i] + omd * src[b * n + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) a[b] = d * a_old[b] + (1.0 - d) * (a_src ? a_src[b] : a[b]);
}

// streaming-tile form (n a multiple of the 16-byte vector)
template <typename T, int U>
__global__ __launch_bounds__(kBlock) void damp_tile(size_t n, int tpb, double d, const double* __restrict__ a_old,
                                                    const T* __restrict__ b_old, double* a, T* bb,
                                                    const double* a_src = nullptr, const T* b_src = nullptr) {
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    const size_t b = blockIdx.y;
    const T dT = T(d), omd = T(1.0 - d);
    const size_t nv = n / VN;
    const V* po = reinterpret_cast<const V*>(b_old + b * n);
    V* pb = reinterpret_cast<V*>(bb + b * n);
    const V* ps = reinterpret_cast<const V*>((b_src ? b_src : bb) + b * n);
    for (int tile = 0; tile < tpb; ++tile) {
        const size_t base = ((size_t)blockIdx.x * tpb + tile) * kBlock * U + threadIdx.x;
        if (base - threadIdx.x >= nv) break;
        V xo[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                xo[u] = po[i];
                xb[u] = ps[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kBlock;
            if (i < nv) {
                V out;
#pragma unroll
                for (int j = 0; j < VN; ++j) vset<T>(out, j, dT * vget<T>(xo[u], j) + omd * vget<T>(xb[u], j));
                pb[i] = out;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a[b] = d * a_old[b] + (1.0 - d) * (a_src ? a_src[b] : a[b]);
}

static int damp_launch(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double damping, const double* a_old_d,
                       const void* b_old_d, const double* a_src_d, const void* b_src_d, double* a_d, void* b_d);

extern "C" int tramp_damp(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double damping,
                          const double* a_old_d, const void* b_old_d, double* a_d, void* b_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, a_old_d && b_old_d && a_d && b_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    if (!n || !batch || damping == 0.0) return TRAMP_OK;
    return damp_launch(ctx, dtype, n, batch, damping, a_old_d, b_old_d, nullptr, nullptr, a_d, b_d);
}

// three operands: (a, b) = d (a_old, b_old) + (1 - d) (a_src, b_src); the output may alias either input
extern "C" int tramp_damp_to(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double damping, const double* a_old_d,
                             const void* b_old_d, const double* a_src_d, const void* b_src_d, double* a_d, void* b_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, a_old_d && b_old_d && a_src_d && b_src_d && a_d && b_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    if (!n || !batch) return TRAMP_OK;
    return damp_launch(ctx, dtype, n, batch, damping, a_old_d, b_old_d, a_src_d, b_src_d, a_d, b_d);
}

static int damp_launch(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double damping, const double* a_old_d,
                       const void* b_old_d, const double* a_src_d, const void* b_src_d, double* a_d, void* b_d) {
    const size_t vn = dtype == TRAMP_F64 ? 2 : 4;
    if ((n % vn) == 0) {
        constexpr int U = kTileU;
        const size_t tiles = (n / vn + (size_t)kBlock * U - 1) / ((size_t)kBlock * U);
        const int tpb = tiles_per_block(tiles, batch);
        dim3 g((unsigned)((tiles + tpb - 1) / tpb), (unsigned)batch);
        if (dtype == TRAMP_F64)
            hipLaunchKernelGGL((damp_tile<double, U>), g, dim3(kBlock), 0, ctx->stream, n, tpb, damping, a_old_d,
                               (const double*)b_old_d, a_d, (double*)b_d, a_src_d, (const double*)b_src_d);
        else
            hipLaunchKernelGGL((damp_tile<float, U>), g, dim3(kBlock), 0, ctx->stream, n, tpb, damping, a_old_d,
                               (const float*)b_old_d, a_d, (float*)b_d, a_src_d, (const float*)b_src_d);
        TRAMP_LAUNCH_CHECK(ctx);
        return TRAMP_OK;
    }
    dim3 grid(blocks_for(n, 1), (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((damp_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, damping, a_old_d,
                           (const double*)b_old_d, a_d, (double*)b_d, a_src_d, (const double*)b_src_d);
    else
        hipLaunchKernelGGL((damp_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, damping, a_old_d,
                           (const float*)b_old_d, a_d, (float*)b_d, a_src_d, (const float*)b_src_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- Gaussian noise channel x = z + N(0, var)
// GaussianChannel.compute_forward_message / compute_backward_message
// (tramp/channels/noise/gaussian_channel.py:23-33): the incoming message shrinks by k = a_ch / (a_ch + a_in),
// (a_new, b_new) = (k a_in, k b_in), a_ch = 1 / var.  Constant damping fused.
template <typename T>
__global__ __launch_bounds__(kBlock) void gauss_channel_kernel(size_t n, double a_ch, const double* __restrict__ a_in,
                                                               const T* __restrict__ b_in, double damping,
                                                               const double* a_old, const T* b_old,
                                                               double* a_new, T* b_new) {
    const size_t b = blockIdx.y;
    const double k = a_ch / (a_ch + a_in[b]);
    const T kT = T(k), dT = T(damping), omd = T(1.0 - damping);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        T v = kT * b_in[b * n + i];
        if (damping != 0.0) v = dT * b_old[b * n + i] + omd * v;
        b_new[b * n + i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double an = k * a_in[b];
        if (damping != 0.0) an = damping * a_old[b] + (1.0 - damping) * an;
        a_new[b] = an;
    }
}

extern "C" int tramp_gauss_channel(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double var,
                                   const double* a_in_d, const void* b_in_d, double damping,
                                   const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, a_in_d && b_in_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, var > 0.0);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!n || !batch) return TRAMP_OK;
    dim3 grid(blocks_for(n, 1), (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((gauss_channel_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, 1.0 / var, a_in_d,
                           (const double*)b_in_d, damping, a_old_d, (const double*)b_old_d, a_new_d, (double*)b_new_d);
    else
        hipLaunchKernelGGL((gauss_channel_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, 1.0 / var, a_in_d,
                           (const float*)b_in_d, damping, a_old_d, (const float*)b_old_d, a_new_d, (float*)b_new_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- bias channel x = z + bias
// BiasChannel.compute_forward_message / compute_backward_message (tramp/channels/linear/bias_channel.py:21-29):
// forward (a_new, b_new) = (az, bz + az bias), backward (ax, bx - ax bias); bias is one vector shared by the
// ensemble.  sign = +1 forward, -1 backward.  Constant damping fused.
template <typename T>
__global__ __launch_bounds__(kBlock) void bias_channel_kernel(size_t n, double sign, const T* __restrict__ bias,
                                                              const double* __restrict__ a_in, const T* __restrict__ b_in,
                                                              double damping, const double* a_old, const T* b_old,
                                                              double* a_new, T* b_new) {
    const size_t b = blockIdx.y;
    const T sa = T(sign * a_in[b]), dT = T(damping), omd = T(1.0 - damping);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        T v = b_in[b * n + i] + sa * bias[i];
        if (damping != 0.0) v = dT * b_old[b * n + i] + omd * v;
        b_new[b * n + i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        a_new[b] = (damping != 0.0) ? damping * a_old[b] + (1.0 - damping) * a_in[b] : a_in[b];
}

extern "C" int tramp_bias_channel(tramp_ctx* ctx, int dtype, int dir, size_t n, size_t batch, const void* bias_d,
                                  const double* a_in_d, const void* b_in_d, double damping,
                                  const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, bias_d && a_in_d && b_in_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!n || !batch) return TRAMP_OK;
    const double sign = dir == TRAMP_FWD ? 1.0 : -1.0;
    dim3 grid(blocks_for(n, 1), (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((bias_channel_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, sign,
                           (const double*)bias_d, a_in_d, (const double*)b_in_d, damping, a_old_d,
                           (const double*)b_old_d, a_new_d, (double*)b_new_d);
    else
        hipLaunchKernelGGL((bias_channel_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, sign,
                           (const float*)bias_d, a_in_d, (const float*)b_in_d, damping, a_old_d,
                           (const float*)b_old_d, a_new_d, (float*)b_new_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- variable -> factor message on a tree
// Variable.forward_message / backward_message (tramp/base.py:186-211): the message a variable of degree
// k sends to one neighbour is the sum of ALL its incoming messages minus that neighbour's own,
// (a_hat - a_l, b_hat - b_l), in that order of operations.
constexpr int kMaxMsgs = 8;
struct MsgPtrs {
    const double* a[kMaxMsgs];
    const void* b[kMaxMsgs];
};

template <typename T>
__global__ __launch_bounds__(kBlock) void message_sum_kernel(size_t n, int k, int exclude, MsgPtrs in,
                                                             double* __restrict__ a_out, T* __restrict__ b_out) {
    const size_t b = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        T s = T(0);
        for (int j = 0; j < k; ++j) s += static_cast<const T*>(in.b[j])[b * n + i];
        if (exclude >= 0) s -= static_cast<const T*>(in.b[exclude])[b * n + i];
        b_out[b * n + i] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += in.a[j][b];
        if (exclude >= 0) s -= in.a[exclude][b];
        a_out[b] = s;
    }
}

// ---------------------------------------------------------------- sum channel x = z_1 + ... + z_K
// SumChannel.compute_forward_message / compute_backward_message (tramp/channels/linear/sum_channel.py:26-43):
//     v_bar = sum_k 1 / a_k,  r_bar = sum_k b_k / a_k;
//     to x:   (1 / v_bar, r_bar / v_bar);
//     to z_k: v_k = 1 / ax + v_bar - 1 / a_k,  r_k = bx / ax - r_bar + b_k / a_k,  (1 / v_k, r_k / v_k).
// target = -1: the forward message; target = k: the backward message to z_k.  Constant damping fused.
template <typename T>
__global__ __launch_bounds__(kBlock) void sum_channel_kernel(size_t n, int k, int target, MsgPtrs in,
                                                             const double* __restrict__ ax, const T* __restrict__ bx,
                                                             double damping, const double* a_old, const T* b_old,
                                                             double* a_new, T* b_new) {
    const size_t b = blockIdx.y;
    double v_bar = 0.0;
    for (int j = 0; j < k; ++j) v_bar += 1.0 / in.a[j][b];
    double v_t = v_bar;
    if (target >= 0) v_t = 1.0 / ax[b] + v_bar - 1.0 / in.a[target][b];
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        double r_bar = 0.0;
        for (int j = 0; j < k; ++j) r_bar += (double)static_cast<const T*>(in.b[j])[b * n + i] / in.a[j][b];
        double r = r_bar;
        if (target >= 0)
            r = (double)bx[b * n + i] / ax[b] - r_bar +
                (double)static_cast<const T*>(in.b[target])[b * n + i] / in.a[target][b];
        double out = r / v_t;
        if (damping != 0.0) out = damping * (double)b_old[b * n + i] + (1.0 - damping) * out;
        b_new[b * n + i] = (T)out;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double an = 1.0 / v_t;
        if (damping != 0.0) an = damping * a_old[b] + (1.0 - damping) * an;
        a_new[b] = an;
    }
}

extern "C" int tramp_sum_channel(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int k,
                                 const double* const* az_d, const void* const* bz_d, const double* ax_d,
                                 const void* bx_d, int target, double damping, const double* a_old_d,
                                 const void* b_old_d, double* a_new_d, void* b_new_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && bz_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, k >= 1 && k <= kMaxMsgs && target >= -1 && target < k);
    TRAMP_REQUIRE(ctx, target < 0 || (ax_d && bx_d));
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!n || !batch) return TRAMP_OK;
    MsgPtrs in{};
    for (int j = 0; j < k; ++j) {
        TRAMP_REQUIRE(ctx, az_d[j] && bz_d[j]);
        in.a[j] = az_d[j];
        in.b[j] = bz_d[j];
    }
    dim3 grid(blocks_for(n, 1), (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((sum_channel_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, k, target, in, ax_d,
                           (const double*)bx_d, damping, a_old_d, (const double*)b_old_d, a_new_d, (double*)b_new_d);
    else
        hipLaunchKernelGGL((sum_channel_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, k, target, in, ax_d,
                           (const float*)bx_d, damping, a_old_d, (const float*)b_old_d, a_new_d, (float*)b_new_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- duplicate channel x_1 = ... = x_K = z
// DuplicateChannel (tramp/channels/shape/duplicate_channel.py:24-31 with base_channel.py:102-105 and base.py:250-255):
//     a = az + sum_k ax_k,  b = bz + sum_k bx_k  (in that order),  r = b / a,  v = 1 / a;
//     the message to the neighbour whose own message is (a_t, b_t):  a_new = clip(1 / v - a_t),  b_new = r (a_t + a_new) - b_t.
// target = -1: the backward message to z; target = k: the forward message to x_k.  Constant damping fused.
template <typename T>
__global__ __launch_bounds__(kBlock) void duplicate_channel_kernel(size_t n, int k, int target, const double* __restrict__ az,
                                                                   const T* __restrict__ bz, MsgPtrs xs, double damping,
                                                                   const double* a_old, const T* b_old, double* a_new, T* b_new) {
    const size_t b = blockIdx.y;
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += xs.a[j][b];
    const double a = az[b] + s;
    const double v = 1.0 / a;
    const double a_t = target < 0 ? az[b] : xs.a[target][b];
    const double an = tmath::a_new_of(v, a_t);
    const double asum = a_t + an;
    const T* b_t = target < 0 ? bz : static_cast<const T*>(xs.b[target]);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        double sb = 0.0;
        for (int j = 0; j < k; ++j) sb += (double)static_cast<const T*>(xs.b[j])[b * n + i];
        const double r = ((double)bz[b * n + i] + sb) / a;
        double out = r * asum - (double)b_t[b * n + i];
        if (damping != 0.0) out = damping * (double)b_old[b * n + i] + (1.0 - damping) * out;
        b_new[b * n + i] = (T)out;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a_new[b] = (damping != 0.0) ? damping * a_old[b] + (1.0 - damping) * an : an;
}

extern "C" int tramp_duplicate_channel(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int k, const double* az_d,
                                       const void* bz_d, const double* const* ax_d, const void* const* bx_d, int target,
                                       double damping, const double* a_old_d, const void* b_old_d, double* a_new_d,
                                       void* b_new_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && bz_d && ax_d && bx_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, k >= 1 && k <= kMaxMsgs && target >= -1 && target < k);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!n || !batch) return TRAMP_OK;
    TRAMP_REQUIRE(ctx, batch <= 65535);
    MsgPtrs xs{};
    for (int j = 0; j < k; ++j) {
        TRAMP_REQUIRE(ctx, ax_d[j] && bx_d[j]);
        xs.a[j] = ax_d[j];
        xs.b[j] = bx_d[j];
    }
    dim3 grid(blocks_for(n, 1), (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((duplicate_channel_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, k, target, az_d,
                           (const double*)bz_d, xs, damping, a_old_d, (const double*)b_old_d, a_new_d, (double*)b_new_d);
    else
        hipLaunchKernelGGL((duplicate_channel_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, k, target, az_d,
                           (const float*)bz_d, xs, damping, a_old_d, (const float*)b_old_d, a_new_d, (float*)b_new_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_message_sum(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int k,
                                 const double* const* a_in_d, const void* const* b_in_d, int exclude,
                                 double* a_out_d, void* b_out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, a_in_d && b_in_d && a_out_d && b_out_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, k >= 1 && k <= kMaxMsgs && exclude < k);
    if (!n || !batch) return TRAMP_OK;
    MsgPtrs in{};
    for (int j = 0; j < k; ++j) {
        TRAMP_REQUIRE(ctx, a_in_d[j] && b_in_d[j]);
        in.a[j] = a_in_d[j];
        in.b[j] = b_in_d[j];
    }
    dim3 grid(blocks_for(n, 1), (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((message_sum_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, k, exclude, in,
                           a_out_d, (double*)b_out_d);
    else
        hipLaunchKernelGGL((message_sum_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, k, exclude, in,
                           a_out_d, (float*)b_out_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void sq_dist_kernel(size_t n, const T* __restrict__ x,
                                                         const T* __restrict__ y, double* __restrict__ partials) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        double d = (double)x[b * n + i] - (y ? (double)y[b * n + i] : 0.0);
        s += d * d;
    }
    s = block_sum(s, smem);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = s;
}

extern "C" int tramp_sq_dist(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const void* x_d,
                             const void* y_d, double* sum_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, x_d && sum_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    if (!batch) return TRAMP_OK;
    if (!n) return tramp_fill_f64(ctx, sum_d, batch, 0.0);
    const int nblk = blocks_for(n, 1);
    int rc = tramp_ws_reserve(ctx, batch * (size_t)nblk * sizeof(double));
    if (rc) return rc;
    dim3 grid(nblk, (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((sq_dist_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, (const double*)x_d,
                           (const double*)y_d, ctx->ws);
    else
        hipLaunchKernelGGL((sq_dist_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, (const float*)x_d,
                           (const float*)y_d, ctx->ws);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, ctx->ws, nblk, 1,
                       sum_d, (double*)nullptr);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---------------------------------------------------------------- convert, randn
template <typename D, typename S>
__global__ __launch_bounds__(kBlock) void convert_kernel(size_t count, const S* __restrict__ src, D* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        dst[i] = (D)src[i];
}

extern "C" int tramp_convert(tramp_ctx* ctx, int dst_dtype, void* dst_d, int src_dtype, const void* src_d,
                             size_t count) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, (dst_d && src_d) || count == 0);
    if (!count) return TRAMP_OK;
    dim3 g(grid_for(count)), b(kBlock);
    if (dst_dtype == TRAMP_F64 && src_dtype == TRAMP_F32)
        hipLaunchKernelGGL((convert_kernel<double, float>), g, b, 0, ctx->stream, count, (const float*)src_d, (double*)dst_d);
    else if (dst_dtype == TRAMP_F32 && src_dtype == TRAMP_F64)
        hipLaunchKernelGGL((convert_kernel<float, double>), g, b, 0, ctx->stream, count, (const double*)src_d, (float*)dst_d);
    else if (dst_dtype == src_dtype && (dst_dtype == TRAMP_F32 || dst_dtype == TRAMP_F64))
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(dst_d, src_d, count * dtype_size(dst_dtype), hipMemcpyDeviceToDevice, ctx->stream));
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// Philox-4x32-10 (Salmon et al. 2011), counter = element pair index, key = seed.
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void randn_kernel(size_t count, unsigned long long seed, double scale,
                                                       T* __restrict__ out) {
    // each counter yields 128 bits -> two 53-bit uniforms -> one Box-Muller pair
    const size_t npair = (count + 1) / 2;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npair; p += (size_t)gridDim.x * kBlock) {
        uint32_t c[4] = {(uint32_t)p, (uint32_t)(p >> 32), 0u, 0u};
        philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        uint64_t u0 = ((uint64_t)c[0] << 32 | c[1]) >> 11;
        uint64_t u1 = ((uint64_t)c[2] << 32 | c[3]) >> 11;
        double x0 = ((double)u0 + 0.5) * (1.0 / 9007199254740992.0);   // (0,1)
        double x1 = ((double)u1 + 0.5) * (1.0 / 9007199254740992.0);
        double rad = sqrt(-2.0 * log(x0)) * scale;
        double s, cth;
        sincospi(2.0 * x1, &s, &cth);
        out[2 * p] = (T)(rad * cth);
        if (2 * p + 1 < count) out[2 * p + 1] = (T)(rad * s);
    }
}

extern "C" int tramp_randn(tramp_ctx* ctx, int dtype, void* out_d, size_t count, unsigned long long seed,
                           double scale) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, out_d || count == 0);
    if (!count) return TRAMP_OK;
    size_t g = ((count + 1) / 2 + kBlock - 1) / kBlock;
    if (g > 16384) g = 16384;
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((randn_kernel<double>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, count, seed, scale, (double*)out_d);
    else if (dtype == TRAMP_F32)
        hipLaunchKernelGGL((randn_kernel<float>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, count, seed, scale, (float*)out_d);
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}
