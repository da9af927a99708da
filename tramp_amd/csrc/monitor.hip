// monitor.hip - the per-iteration bookkeeping of EP on the device.
//
// What the reference does on the host after every iteration - the NaN guard of
// MessagePassing.check_message (tramp/algos/message_passing.py:187-209) and the stopping rule of
// EarlyStoppingEP (tramp/algos/callbacks.py:250-286: stop when max_var |r - r_old| / |r| < tol; give up
// and go back to the previous iterate when it exceeds max_increase after wait_increase iterations) -
// runs here as two tiny kernels behind the sweeps, per problem instance of the ensemble:
//   tramp_ep_monitor  reads the scalar tables A [2L, B] (message precisions) and S [3L, B]
//                     (v, sum (r - r_old)^2, sum r^2 per variable), advances the iteration counter,
//                     and moves each still-running instance to CONVERGED / DIVERGED with the iteration
//                     at which it happened; keeps the v of the iterate to report (the current one for a
//                     converged instance, the previous one for a diverged instance: the reference
//                     restores the previous message dag there);
//   tramp_ep_freeze   copies the posterior mean of the instances that stopped in THIS iteration
//                     (current r, or the previous r for a diverged instance) into the frozen buffer.
// Both are ordinary launches on the context's stream, so they are captured with the iteration in a
// hipGraph; the host only polls the 4-word header every k iterations (iterate(check_every=k)).
#include "common.h"

// state_d layout (int32): [0] iterations completed, [1] instances still running, [2] NaN code
// (0: none; else 1 + edge * B + instance of the first NaN precision), [3] instances diverged,
// then status[B] (0 running, 1 converged, 2 diverged), then n_iter[B] (iteration at which it stopped).
__global__ __launch_bounds__(kBlock) void ep_monitor_kernel(int L, int E, size_t B, const double* __restrict__ A,
                                                            const double* __restrict__ S, double tol,
                                                            double max_increase, int wait_increase,
                                                            int* __restrict__ state, double* __restrict__ v_prev,
                                                            double* __restrict__ v_frozen, double* __restrict__ a_frozen,
                                                            const double* __restrict__ A_prev) {
    __shared__ int s_running, s_nan, s_div;
    int* status = state + 4;
    int* n_iter = state + 4 + B;
    const int i = state[0];                    // 0-based index of the iteration that just finished
    if (threadIdx.x == 0) {
        s_running = 0;
        s_nan = 0;
        s_div = 0;
    }
    __syncthreads();
    for (size_t b = threadIdx.x; b < B; b += kBlock) {
        int st = status[b];
        // NaN guard on every precision of this instance (a non-finite mean poisons the reduced variance,
        // so the a's suffice: DESIGN.md section 1)
        int nan_code = 0;
        for (int e = 0; e < E && !nan_code; ++e) {
            const double a = A[(size_t)e * B + b];
            if (a != a) nan_code = 1 + (int)((size_t)e * B + b);
        }
        if (nan_code) atomicCAS(&s_nan, 0, nan_code);
        if (st == 0) {
            if (i >= 1) {                      // old_rs is None on the first iteration (callbacks.py:261-262)
                double worst = 0.0;
                bool any_nan = false;
                for (int l = 0; l < L; ++l) {
                    const double rel = sqrt(S[(size_t)(3 * l + 1) * B + b]) / sqrt(S[(size_t)(3 * l + 2) * B + b]);
                    if (rel != rel) any_nan = true;          // np.max propagates NaN
                    else if (rel > worst) worst = rel;
                }
                if (!any_nan && worst < tol) st = 1;
                else if (i > wait_increase && (any_nan ? false : worst > max_increase)) st = 2;
                if (st != 0) {
                    status[b] = st;
                    n_iter[b] = i + 1;
                    for (int l = 0; l < L; ++l)
                        v_frozen[(size_t)l * B + b] = (st == 1) ? S[(size_t)(3 * l) * B + b] : v_prev[(size_t)l * B + b];
                    // the message precisions of the stopping iteration; with a snapshot of the previous iteration's
                    // messages (A_prev: restore_messages) a diverged instance goes back to those, as the reference
                    // restores its whole previous message dag (callbacks.py:277-283) - without one it keeps the
                    // diverged precisions and only means and variances are rolled back
                    const double* Asrc = (st == 2 && A_prev) ? A_prev : A;
                    for (int e = 0; e < E; ++e) a_frozen[(size_t)e * B + b] = Asrc[(size_t)e * B + b];
                }
            }
            if (st == 0) atomicAdd(&s_running, 1);
        }
        if (st == 2) atomicAdd(&s_div, 1);
        // the v of this iterate becomes "previous" for the next one
        for (int l = 0; l < L; ++l) v_prev[(size_t)l * B + b] = S[(size_t)(3 * l) * B + b];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        state[0] = i + 1;
        state[1] = s_running;
        if (state[2] == 0) state[2] = s_nan;
        state[3] = s_div;
    }
}

extern "C" int tramp_ep_monitor(tramp_ctx* ctx, int n_vars, int n_edges, size_t batch, const double* A_d,
                                const double* S_d, double tol, double max_increase, int wait_increase,
                                int* state_d, double* v_prev_d, double* v_frozen_d, double* a_frozen_d,
                                const double* A_prev_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, n_vars > 0 && n_edges > 0 && batch > 0 && A_d && S_d && state_d && v_prev_d && v_frozen_d &&
                           a_frozen_d);
    hipLaunchKernelGGL(ep_monitor_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, n_vars, n_edges, batch, A_d, S_d,
                       tol, max_increase, wait_increase, state_d, v_prev_d, v_frozen_d, a_frozen_d, A_prev_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void ep_freeze_kernel(size_t n, size_t B, const int* __restrict__ state,
                                                           const T* __restrict__ r_cur, const T* __restrict__ r_prev,
                                                           T* __restrict__ frozen) {
    const size_t b = blockIdx.y;
    const int* status = state + 4;
    const int* n_iter = state + 4 + B;
    if (n_iter[b] != state[0]) return;          // did not stop in the iteration the monitor just closed
    const T* src = (status[b] == 2 ? r_prev : r_cur) + b * n;
    T* dst = frozen + b * n;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) dst[i] = src[i];
}

extern "C" int tramp_ep_freeze(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const int* state_d,
                               const void* r_cur_d, const void* r_prev_d, void* frozen_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, n > 0 && batch > 0 && batch <= 65535 && state_d && r_cur_d && r_prev_d && frozen_d);
    size_t g = (n + (size_t)kBlock * 8 - 1) / ((size_t)kBlock * 8);
    if (g > 256) g = 256;
    dim3 grid((unsigned)g, (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((ep_freeze_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, batch, state_d,
                           (const double*)r_cur_d, (const double*)r_prev_d, (double*)frozen_d);
    else if (dtype == TRAMP_F32)
        hipLaunchKernelGGL((ep_freeze_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, batch, state_d,
                           (const float*)r_cur_d, (const float*)r_prev_d, (float*)frozen_d);
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// ---- restore: every instance that has stopped gets the messages of ITS stopping iteration back (what a run of
// that instance alone would have left behind: the current ones if it converged, the previous iteration's if it
// gave up after an increase - MessagePassing.reset_message_dag(old_message_dag), callbacks.py:277-283)
template <typename T>
__global__ __launch_bounds__(kBlock) void ep_restore_kernel(size_t n, size_t B, const int* __restrict__ state,
                                                            const T* __restrict__ frozen, T* __restrict__ live, int all) {
    const size_t b = blockIdx.y;
    if (!all && state[4 + b] == 0) return;
    const T* src = frozen + b * n;
    T* dst = live + b * n;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) dst[i] = src[i];
}

extern "C" int tramp_ep_restore(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const int* state_d,
                                const void* frozen_d, void* live_d, int all) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, n > 0 && batch > 0 && batch <= 65535 && state_d && frozen_d && live_d);
    size_t g = (n + (size_t)kBlock * 8 - 1) / ((size_t)kBlock * 8);
    if (g > 256) g = 256;
    dim3 grid((unsigned)g, (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((ep_restore_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, batch, state_d,
                           (const double*)frozen_d, (double*)live_d, all);
    else if (dtype == TRAMP_F32)
        hipLaunchKernelGGL((ep_restore_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, batch, state_d,
                           (const float*)frozen_d, (float*)live_d, all);
    else
        return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "dtype");
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// the precision table A [E, B] (instance fastest): columns of the stopped instances from a_frozen
__global__ __launch_bounds__(kBlock) void ep_restore_a_kernel(int E, size_t B, const int* __restrict__ state,
                                                              const double* __restrict__ a_frozen, double* __restrict__ A,
                                                              int all) {
    for (size_t b = threadIdx.x; b < B; b += kBlock) {
        if (!all && state[4 + b] == 0) continue;
        for (int e = 0; e < E; ++e) A[(size_t)e * B + b] = a_frozen[(size_t)e * B + b];
    }
}

extern "C" int tramp_ep_restore_a(tramp_ctx* ctx, int n_edges, size_t batch, const int* state_d, const double* a_frozen_d,
                                  double* A_d, int all) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, n_edges > 0 && batch > 0 && state_d && a_frozen_d && A_d);
    hipLaunchKernelGGL(ep_restore_a_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, n_edges, batch, state_d, a_frozen_d, A_d,
                       all);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}
