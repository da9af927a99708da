// ew_resident.h - resident single-pass form of the componentwise factor updates (gfx950).
//
// b_new = r (a + a_new) - b needs a_new = clip(1/mean_i v_i - a) (tramp/base.py:245-255): a global mean stands between
// reading a message and writing its update.  The tile kernels of ew_kernels.hip therefore see every input twice (3T or
// 5T moved for 2T / 3T algorithmic).  Here the chip's register files hold the window of the stream that is waiting for
// its mean, so each byte crosses HBM once:
//
//   * a fixed grid of 2 x CUs workgroups, all resident, P of them streaming; tiles of 256 x U 16-byte pieces are numbered instance-major
//     and round k gives workgroup w the tile k P + w, so one round covers about one instance at 64 x 1e6;
//   * every workgroup runs a software pipeline over its rounds: iteration j ISSUES the loads of round j, evaluates the
//     posterior moments (r_i, v_i) of round j - DL (its loads were issued DL iterations ago), and WRITES the update of
//     round j - DL - DA, whose instance mean has had DA iterations to form.  b and r of the DL + DA + 1 rounds in flight
//     live in registers (unrolled ring, static indices);
//   * the mean: a tile's block sum goes to its own 8-byte slot (one relaxed agent-scope store: a self-validating granule,
//     sentinel = not yet written).  The first kResCombiners workgroups of the grid do not stream: combiner c takes the
//     instances i = c, c + kResCombiners, ... in order, sweeps instance i's slots with L1-bypassing loads until none is the
//     sentinel, adds them in FIXED order (run-to-run reproducible: no float atomics), forms a_new / v / the support count
//     and publishes a_in + a_new as one more granule, which every streaming workgroup reads DA iterations after its tile's
//     moments.  No fence, no read-modify-write atomic, no grid barrier; a streaming workgroup waits only in the (rare)
//     re-poll of a granule that has not arrived.  (A streaming workgroup that combines on the side holds back its own
//     tiles while it waits for the others' slots, then everybody waits for its tiles: measured +90...150 us at 64 x 1e6.)
//
// An instance must fit the window: tiles per instance <= P (DA - 2) + 1.  Larger instances and ragged n take the tile
// kernels.  Deadlock freedom: round k's update waits only for moments of rounds <= k + S (S = rounds an instance can
// straddle beyond its first), which every workgroup evaluates before it waits for anything of round > k + S - DA; with
// DA >= S + 2 and all P workgroups resident the dependence is acyclic.  Every wait is bounded by the kernel's deadline
// (res_expired: NaN results and a host-visible error word, never a trap).
#pragma once
#include <type_traits>

#include "ew_ops.h"

constexpr unsigned long long kResSentinel = 0xFFFFFFFFFFFFFFFFull;     // a NaN no arithmetic produces
constexpr int kResBlocksPerCU = 2;      // __launch_bounds__(256, 2): <= 256 VGPRs, two workgroups per CU
constexpr int kResQ = 8;                // the combiner sweeps 8 slots per lane at a time (512 tiles per batch of loads)

struct ResArgs {
    size_t n, nv;                     // components / 16-byte pieces per instance
    unsigned batch, tiles;            // instances; tiles per instance
    unsigned long long total;         // batch * tiles
    unsigned adv_q, adv_r;            // P / tiles, P % tiles: how (instance, tile) advance from one round to the next
    const double* a_in;               // precision of the message being updated
    unsigned long long* slots;        // [batch][tiles] bits of the tile sums of v_i
    unsigned long long* cslots;       // [batch][tiles] support counts (L1 prior) or nullptr
    unsigned long long* scal;         // [batch] bits of a_in + a_new
    const uint8_t* bits;              // OpBits factors: the second input, one bit per component, rows of bits_stride bytes
    size_t bits_stride;
    void* dummy;                      // 256 U 16-byte pieces: where the overlap of an instance's last tile is "stored"
    volatile unsigned* err_host;      // pinned host word: set to 1 by a workgroup whose wait ran past the deadline
    long long deadline_ticks;         // 100 MHz ticks from kernel start after which nobody waits any more (0: 2 s)
    unsigned long long dbg_skip_plus1;  // debug: the tile (instance * tiles + tile, plus 1) whose sum is never published
    MsgOut out;
};

__device__ __forceinline__ unsigned long long res_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // global_load_dwordx2 sc1
}
__device__ __forceinline__ void res_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               // global_store_dwordx2 sc1
}
// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima).  As a builtin
// the compiler's own wait-count bookkeeping sees it: registers loaded before it are known to have arrived.
constexpr int kWaitVm0 = 0x0F70;
// probes only (tools/probes/probe_*resident.hip): the library's kernels never trap - see res_expired
__device__ __forceinline__ void res_spin_check(long long t0) {
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > 200000000LL) __builtin_trap();   // 100 MHz: 2 s
}
// A hand-off that has not arrived by the kernel's deadline (2 s from its start unless ResArgs says otherwise) is given
// up: the waiting workgroup sets the host-visible error word and goes on with NaN - the combiner publishes NaN for the
// instance (a_new, v and the scalar the streaming workgroups wait for), a streaming workgroup writes a NaN update - so
// the kernel drains and ENDS, the affected instances are loudly NaN, and the host (run_resident) takes the tile kernels
// from then on (tramp_resident_errors counts).  What can make a hand-off late: fewer workgroups co-resident than the
// occupancy query promised (a CU mask, another process's kernels on the device).
// The combiners stop publishing real scalars AT the deadline (everything later is NaN); the streaming workgroups give up
// only `grace` ticks after it, so a streaming workgroup never writes a NaN update for an instance whose a_new came out finite.
__device__ __forceinline__ bool res_expired(long long t_start, const ResArgs& A, long long grace = 0) {
    const long long lim = (A.deadline_ticks > 0 ? A.deadline_ticks : 200000000LL) + grace;
    return (long long)__builtin_amdgcn_s_memrealtime() - t_start > lim;
}

template <typename T, bool NT> __device__ __forceinline__ typename Vec<T>::type res_ld(const typename Vec<T>::type* p) {
    if (NT) {
        using NV = typename NativeVec<T>::type;
        const NV v = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
        return *reinterpret_cast<const typename Vec<T>::type*>(&v);
    }
    return *p;
}

template <int S, int E, typename F> __device__ __forceinline__ void res_static_for(F&& f) {
    if constexpr (S < E) {
        f(std::integral_constant<int, S>{});
        res_static_for<S + 1, E>(f);
    }
}

// Sentinels into the slots and the published scalars; the per-instance constants of the factor, once per instance
// (the resident kernel re-reads them through the scalar cache when its round changes instance).
template <typename Op>
__global__ __launch_bounds__(kBlock) void ew_res_prepare(Op op, ResArgs A, Op* __restrict__ ops) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < A.total) {
        A.slots[i] = kResSentinel;
        if (A.cslots) A.cslots[i] = kResSentinel;
    }
    if (i < A.batch) {
        A.scal[i] = kResSentinel;
        Op o = op;
        o.prepare(i);
        ops[i] = o;
    }
}

// Combiner workgroup c (wave 0; the other waves leave): instances c, c + kResCombiners, ... in order.  COUNT: the L1
// prior's support counts travel in a second slot array.
constexpr unsigned kResCombiners = 8;

template <bool COUNT> __device__ __forceinline__ void res_combiner(const ResArgs& A, unsigned c, long long t_start) {
    const unsigned lane = threadIdx.x;
    bool dead = false;                                     // the deadline has passed: NaN for every instance still to come
    for (unsigned inst = c; inst < A.batch; inst += kResCombiners) {
        const unsigned long long* sl = A.slots + (size_t)inst * A.tiles;
        const unsigned long long* cl = COUNT ? A.cslots + (size_t)inst * A.tiles : nullptr;
        double acc;
        unsigned long long cnt;
        for (;;) {
            bool ok = true;
            acc = 0.0;
            cnt = 0;
            for (unsigned q0 = 0; q0 < A.tiles; q0 += 64u * kResQ) {
                unsigned long long bits[kResQ];
#pragma unroll
                for (int q = 0; q < kResQ; ++q) {
                    const unsigned i = q0 + lane + 64u * q;
                    bits[q] = (i < A.tiles) ? res_load(sl + i) : 0ull;
                }
#pragma unroll
                for (int q = 0; q < kResQ; ++q) {
                    ok = ok && bits[q] != kResSentinel;
                    acc += __longlong_as_double((long long)bits[q]);
                }
                if (COUNT) {
#pragma unroll
                    for (int q = 0; q < kResQ; ++q) {
                        const unsigned i = q0 + lane + 64u * q;
                        bits[q] = (i < A.tiles) ? res_load(cl + i) : 0ull;
                    }
#pragma unroll
                    for (int q = 0; q < kResQ; ++q) {
                        ok = ok && bits[q] != kResSentinel;
                        cnt += bits[q];
                    }
                }
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (res_expired(t_start, A)) break;
        }
        if (dead || res_expired(t_start, A)) {             // (wave-uniform)
            dead = true;
            acc = __longlong_as_double(0x7ff8000000000000LL);
            if (lane == 0 && A.err_host) *A.err_host = 1u;
        }
        acc = wave_sum(acc);
        if (COUNT) cnt = wave_sum_u64(cnt);
        if (lane == 0) {
            const double v = acc / (double)A.n;
            const double a_in = A.a_in[inst];
            const double a_new = tmath::a_new_of(v, a_in);
            const double d = A.out.damping;
            A.out.a_new[inst] = (d != 0.0) ? d * A.out.a_old[inst] + (1.0 - d) * a_new : a_new;
            if (A.out.v) A.out.v[inst] = v;
            if (COUNT && A.out.count) A.out.count[inst] = cnt;
            unsigned long long sb = (unsigned long long)__double_as_longlong(a_in + a_new);
            if (sb == kResSentinel) sb = 0x7ff8000000000000ull;
            res_store(A.scal + inst, sb);
        }
    }
}

// block reductions for NT = 256 or 512 threads (fixed order: run-to-run reproducible)
template <int NT> __device__ __forceinline__ double res_block_sum(double v, double* smem) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    double t = smem[0];
#pragma unroll
    for (int i = 1; i < NT / kWave; ++i) t += smem[i];
    return t;
}
template <int NT> __device__ __forceinline__ unsigned long long res_block_sum_u64(unsigned long long v, unsigned long long* smem) {
    v = wave_sum_u64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    unsigned long long t = smem[0];
#pragma unroll
    for (int i = 1; i < NT / kWave; ++i) t += smem[i];
    return t;
}

// Flags: which rarely wanted outputs this instantiation writes (compile time: no branch in the stream).
constexpr int kResDamp = 1, kResStoreR = 2, kResMask = 4, kResNtLoad = 8, kResSchedBarrier = 16, kResRegR = 32;
// kResReloadB: the target message b is NOT kept in the register ring while its round waits for the mean; it is loaded a
// second time one iteration before the update (DL + DA - 1 rounds, ~100 MB of traffic, after its first read: served by the
// 256 MB Infinity Cache, so HBM still sees it once).  The ring then holds two rounds of inputs instead of five of b: the
// heavy two-input factors (K7) evaluate without spilling.
constexpr int kResReloadB = 64;
// kResRecompute (one-input factors; probes only - no library instantiation uses it): the posterior mean r is NOT parked
// while its round waits for the mean - the update stage evaluates the factor a second time from b, which the ring holds
// anyway.  No LDS, half the registers of the register-parked form, so three or four workgroups per CU fit.  Measured in
// round 6 for the Gauss-Bernoulli prior at 64 x 1e6 (tools/probes/probe_resident.hip, profiles/r06/probe_resident_recompute.txt):
// 223 us parked (LDS or registers alike) against 276 / 238 / 269 us recomputed on 2 / 3 / 4 workgroups per CU, and deeper
// load prefetch (DL = 2, 3) is slower still - the pipeline is not short of loads in flight, the second evaluation costs
// what it costs.  Kept as a flag so the measurement can be repeated.
constexpr int kResRecompute = 128;

// DBG (probes only): 1 = no slots, no combiner, no polling (the pipeline as a plain stream with a constant scalar).
// Every tile is FULL: the last tile of an instance starts at nv - 256 U, overlapping its predecessor; the overlap is
// evaluated twice, counted once and stored once (the second copy goes to a dummy tile: a select on the address, so the
// stream has no per-lane control flow at all, and every piece is written by exactly one tile - outputs may alias inputs).
// NT threads per workgroup (256 or 512), BPC workgroups per CU: NT x BPC = 512 threads per CU = two waves per SIMD.
template <typename T, typename Op, int U, int DL, int DA, int FLAGS, int BPC = kResBlocksPerCU, int DBG = 0, int NT = kBlock>
__global__ __launch_bounds__(NT, BPC * NT / 256) void ew_resident(const Op* __restrict__ ops, ResArgs A,
                                                           const T* __restrict__ in0, const T* __restrict__ in1) {
    using V = typename Vec<T>::type;
    using NV = typename NativeVec<T>::type;
    constexpr int VN = Vec<T>::N;
    constexpr int R = DL + DA + 1;
    constexpr unsigned TV = NT * U;                       // 16-byte pieces per tile
    constexpr bool COUNT = OpCounts<Op>::value, BITS = OpBits<Op>::value;
    constexpr bool DAMP = (FLAGS & kResDamp) != 0, STORE_R = (FLAGS & kResStoreR) != 0, MASK = (FLAGS & kResMask) != 0;
    constexpr bool RELOAD = (FLAGS & kResReloadB) != 0;
    __shared__ double smem[NT / kWave];
    __shared__ unsigned long long smem_u[NT / kWave];
    // r of the rounds that wait for their mean: in LDS (each thread parks and fetches its own pieces: no barrier), so the
    // register ring holds only the messages b.  (With r in registers too every factor but the two priors spilled.)
    constexpr bool RECOMP = (FLAGS & kResRecompute) != 0;
    static_assert(!RECOMP || (Op::NIN == 1 && !BITS && !OpLean<Op>::value), "kResRecompute: one-input factors with a full form");
    __shared__ V r_lds[(FLAGS & (kResRegR | kResRecompute)) ? 1 : DA + 1][U][NT];
    const unsigned tid = threadIdx.x;
    const long long t_start = (long long)__builtin_amdgcn_s_memrealtime();
    if (blockIdx.x < kResCombiners) {
        if (DBG != 1 && tid < kWave) res_combiner<COUNT>(A, blockIdx.x, t_start);
        return;
    }
    Op::stage();
    const unsigned P = gridDim.x - kResCombiners, wg = blockIdx.x - kResCombiners;      // streaming workgroups
    const unsigned nv = (unsigned)A.nv;
    const V* p0 = reinterpret_cast<const V*>(in0);
    const V* p1 = reinterpret_cast<const V*>(in1);
    const V* pold = reinterpret_cast<const V*>(A.out.b_old);
    V* pb = reinterpret_cast<V*>(A.out.b_new);
    V* pr = reinterpret_cast<V*>(A.out.r);
    const T dT = T(A.out.damping), omd = T(1.0 - A.out.damping);

    // ring of rounds in flight: x0 / x1 hold the inputs until the moments are taken, then x0 the target message b
    V x0[R][U], x1[R][U], xold[R][U], xt[R][U];
    const V* ptgt = (Op::TARGET == 0) ? p0 : p1;
    unsigned xb[R][U];                         // BITS: the byte that holds this piece's bits of the packed second input
    unsigned lm = 0;                           // LDS slot of the round whose moments are taken: (j - DL) mod (DA + 1)
    unsigned r_inst[R], r_tile[R];
    bool r_live[R];
#pragma unroll
    for (int s = 0; s < R; ++s) r_live[s] = false;
    // (instance, tile) of the round whose loads are issued next
    unsigned l_inst = wg / A.tiles, l_tile = wg % A.tiles;
    // rounds in which THIS workgroup has a tile: iterations DL + DA <= j < Kw run all three stages (the steady state,
    // compiled without a single conditional memory operation: the compiler's s_waitcnt counting takes the path with the
    // fewest younger operations, so one skippable load or store in the window costs the whole prefetch distance)
    const unsigned long long Kw = (A.total > wg) ? (A.total - wg + P - 1) / P : 0;

    // one pipeline iteration; `s` = ring slot of the round whose loads are issued, STEADY = every stage is live
    auto body = [&](auto sc, auto steady_c, const unsigned long long j) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        constexpr bool STEADY = decltype(steady_c)::value;
        constexpr int sM = (s - DL + R) % R;             // round j - DL: moments
        constexpr int sA = (s + 1) % R;                  // round j - DL - DA: update
        constexpr int sN = (s + 2) % R;                  // the round updated by the NEXT iteration (b_old prefetch)
        // ---- ask for the published scalar of the round that is updated below (BEFORE this round's loads: the
        //      counter retires in order, and the update must not wait for the loads issued here)
        const bool do_a = STEADY || (j >= (unsigned long long)(DL + DA) && r_live[sA]);
        unsigned long long sbits = (unsigned long long)__double_as_longlong(1.5);
        if (do_a && DBG != 1) sbits = res_load(A.scal + r_inst[sA]);
        // ---- issue the loads of round j
        if (!STEADY) r_live[s] = false;
        if (STEADY || j < Kw) {
            r_live[s] = true;
            r_inst[s] = l_inst;
            r_tile[s] = l_tile;
            const size_t base = (size_t)l_inst * nv;
            const unsigned off = min(l_tile * TV, nv - TV) + tid;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x0[s][u] = res_ld<T, (FLAGS & kResNtLoad) != 0>(&p0[base + off + u * NT]);
                if (Op::NIN > 1) x1[s][u] = res_ld<T, (FLAGS & kResNtLoad) != 0>(&p1[base + off + u * NT]);
                if (BITS) xb[s][u] = A.bits[(size_t)l_inst * A.bits_stride + (((size_t)(off + u * NT) * VN) >> 3)];
            }
            l_inst += A.adv_q;
            l_tile += A.adv_r;
            if (l_tile >= A.tiles) {
                l_tile -= A.tiles;
                l_inst += 1;
            }
        }
        if (DAMP && (STEADY || (r_live[sN] && j + 1 >= (unsigned long long)(DL + DA)))) {
            const size_t base = (size_t)r_inst[sN] * nv;
            const unsigned off = min(r_tile[sN] * TV, nv - TV) + tid;
#pragma unroll
            for (int u = 0; u < U; ++u) xold[sN][u] = pold[base + off + u * NT];
        }
        if (RELOAD && (STEADY || (r_live[sN] && j + 1 >= (unsigned long long)(DL + DA)))) {
            const size_t base = (size_t)r_inst[sN] * nv;
            const unsigned off = min(r_tile[sN] * TV, nv - TV) + tid;
#pragma unroll
            for (int u = 0; u < U; ++u) xt[sN][u] = ptgt[base + off + u * NT];
        }
        // ---- moments of round j - DL
        if (STEADY || (j >= (unsigned long long)DL && r_live[sM])) {
            const Op op = ops[r_inst[sM]];
            const size_t base = (size_t)r_inst[sM] * nv;
            const unsigned first = r_tile[sM] * TV;                     // pieces below belong to the previous tile
            const unsigned off = min(first, nv - TV) + tid;
            double acc = 0.0;
            unsigned long long cnt = 0;
            T bad = T(0);                                               // lean ops: sum of EwOut::dev (0 = every lane inside the form's domain)
            T poison = T(0);                                            // sum of 0 * r: NaN once any posterior mean is not finite
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned i = off + u * NT;
                const bool counts = i >= first;
                V vr;
                double pacc = 0.0;                                      // this piece's sum of v: the `counts` select once per piece
                if constexpr (OpPair<Op>::value && (VN % 2) == 0 && !MASK) {
#pragma unroll
                    for (int c = 0; c < VN; c += 2) {
                        const T a0 = vget<T>(x0[sM][u], c), a1 = vget<T>(x0[sM][u], c + 1);
                        T y0 = (Op::NIN > 1) ? vget<T>(x1[sM][u], c) : a0, y1 = (Op::NIN > 1) ? vget<T>(x1[sM][u], c + 1) : a1;
                        if (BITS) {
                            y0 = ((xb[sM][u] >> (((i * VN) & 7u) + c)) & 1u) ? T(1) : T(-1);
                            y1 = ((xb[sM][u] >> (((i * VN) & 7u) + c + 1)) & 1u) ? T(1) : T(-1);
                        }
                        EwOut<T> o0, o1;
                        op.eval2(a0, a1, y0, y1, o0, o1);
                        bad += o0.dev + o1.dev;
                        pacc += (double)o0.v;
                        pacc += (double)o1.v;
                        poison = tmath::fma_(o0.r, T(0), poison);
                        poison = tmath::fma_(o1.r, T(0), poison);
                        vset<T>(vr, c, o0.r);
                        vset<T>(vr, c + 1, o1.r);
                    }
                } else {
#pragma unroll
                for (int c = 0; c < VN; ++c) {
                    const T a0 = vget<T>(x0[sM][u], c);
                    T a1 = (Op::NIN > 1) ? vget<T>(x1[sM][u], c) : a0;
                    if (BITS) a1 = ((xb[sM][u] >> (((i * VN) & 7u) + c)) & 1u) ? T(1) : T(-1);
                    EwOut<T> o = op.eval(a0, a1);
                    if (OpLean<Op>::value) {
                        bad += o.dev;
                        if (FLAGS & kResSchedBarrier) __builtin_amdgcn_sched_barrier(0);       // one component at a time
                    }
                    pacc += (double)o.v;
                    poison = tmath::fma_(o.r, T(0), poison);
                    if (OpCounts<Op>::value) cnt += counts ? (o.mask & 1u) : 0u;
                    vset<T>(vr, c, o.r);
                    if (MASK) {
                        uint8_t* mo = counts ? A.out.mask + (size_t)r_inst[sM] * A.n + (size_t)i * VN + c
                                             : reinterpret_cast<uint8_t*>(A.dummy) + tid;
                        *mo = (uint8_t)o.mask;          // (lean ops have no mask output: run_resident refuses out.mask)
                    }
                }
                }
                acc += counts ? pacc : 0.0;
                if (STORE_R) {
                    V* dst = counts ? &pr[base + i] : reinterpret_cast<V*>(A.dummy) + tid + u * NT;
                    __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&vr), reinterpret_cast<NV*>(dst));
                }
                // keep (b, r): b in the ring, r in LDS
                if (Op::TARGET != 0 && !RELOAD) x0[sM][u] = x1[sM][u];
                if (FLAGS & kResRegR) x1[sM][u] = vr;
                else if (!RECOMP) r_lds[lm][u][tid] = vr;
                // one 16-byte piece at a time: interleaving all U evaluations (the scheduler's choice when left alone) needs
                // U times the temporaries, and the ring leaves no room for them
                if (FLAGS & kResSchedBarrier) __builtin_amdgcn_sched_barrier(0);
            }
            // a lean op that left its form's domain anywhere in the tile poisons the tile's sum: the instance's a_new comes
            // out NaN, which is what the caller's redo on the tile kernels looks for (ew_kernels.hip: pwl_dir, RedoArgs) (a select, not a branch: no control flow in the pipelined body)
            if (OpLean<Op>::value) acc += (bad == T(0)) ? 0.0 : __longlong_as_double(0x7ff8000000000000LL);
            acc += (double)poison;                                      // (pieces that do not count are repeats of ones that do)
            acc = res_block_sum<NT>(acc, smem);
            if (COUNT) cnt = res_block_sum_u64<NT>(cnt, smem_u);
            if (tid == 0 && DBG != 1) {
                unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
                if (ab == kResSentinel) ab = 0x7ff8000000000000ull;
                const size_t si = (size_t)r_inst[sM] * A.tiles + r_tile[sM];
                // (debug: one tile's sum goes to the dummy area instead - a select on the address, no branch)
                unsigned long long* sdst = (si + 1 == A.dbg_skip_plus1) ? reinterpret_cast<unsigned long long*>(A.dummy) : A.slots + si;
                res_store(sdst, ab);
                if (COUNT) res_store(A.cslots + si, cnt);
            }
        }
        // ---- update of round j - DL - DA
        if (do_a) {
            if (sbits == kResSentinel) {
                do {
                    __builtin_amdgcn_s_sleep(4);
                    sbits = res_load(A.scal + r_inst[sA]);
                    __builtin_amdgcn_s_waitcnt(kWaitVm0);     // on EVERY way out of this loop nothing is pending
                } while (sbits == kResSentinel && !res_expired(t_start, A, 1000000LL));     // 10 ms after the combiners
                // past the deadline: the sentinel's bits are a NaN - this round's update is written as NaN
                if (sbits == kResSentinel && tid == 0 && A.err_host) *A.err_host = 1u;
            }
            const T vinv = T(__longlong_as_double((long long)sbits));
            const unsigned la = (lm + 1 > DA) ? 0 : lm + 1;             // (j - DL - DA) mod (DA + 1)
            const size_t base = (size_t)r_inst[sA] * nv;
            const unsigned first = r_tile[sA] * TV;
            const unsigned off = min(first, nv - TV) + tid;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                V vo;
                V vr;
                if constexpr (RECOMP) {
                    const Op opA = ops[r_inst[sA]];
#pragma unroll
                    for (int c = 0; c < VN; ++c) vset<T>(vr, c, opA.eval(vget<T>(x0[sA][u], c), T(0)).r);
                } else {
                    vr = (FLAGS & kResRegR) ? x1[sA][u] : r_lds[la][u][tid];
                }
#pragma unroll
                for (int c = 0; c < VN; ++c) {
                    T bn = vget<T>(vr, c) * vinv - vget<T>(RELOAD ? xt[sA][u] : x0[sA][u], c);
                    if (DAMP) bn = dT * vget<T>(xold[sA][u], c) + omd * bn;
                    vset<T>(vo, c, bn);
                }
                V* dst = (off + u * NT >= first) ? &pb[base + off + u * NT] : reinterpret_cast<V*>(A.dummy) + tid + u * NT;
                __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&vo), reinterpret_cast<NV*>(dst));
            }
        }
        if (STEADY || j >= (unsigned long long)DL) lm = (lm + 1 > DA) ? 0 : lm + 1;
    };
    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;

    // prologue: iterations 0 .. R - 2 fill the ring
    res_static_for<0, R - 1>([&](auto sc) __attribute__((always_inline)) { body(sc, No{}, (unsigned long long)decltype(sc)::value); });
    // steady state from iteration R - 1 = DL + DA, R iterations per trip (slots R - 1, 0, 1, ...)
    unsigned long long j0 = R - 1;
    for (; j0 + R <= Kw; j0 += R)
        res_static_for<0, R>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            body(std::integral_constant<int, (s + R - 1) % R>{}, Yes{}, j0 + s);
        });
    // the rest of the rounds, then the drain: fewer than 3 R iterations
    const unsigned long long Jw = Kw + DL + DA;
    for (; j0 < Jw; j0 += R)
        res_static_for<0, R>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            if (j0 + s < Jw) body(std::integral_constant<int, (s + R - 1) % R>{}, No{}, j0 + s);
        });
}

