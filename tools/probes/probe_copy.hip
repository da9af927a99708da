// probe_copy.hip - which launch shape streams HBM fastest on this MI355X (16-byte lanes)?
// The guide quotes 6.29 TB/s for a float4 copy; round 1's grid-stride probes read 4.6-5.0 TB/s.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/probe_copy.hip -o tools/probes/build/probe_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef double __attribute__((ext_vector_type(2))) V;   // native vector: nontemporal builtins accept it

// A: one 16-byte piece per thread, grid = n/256 blocks (workgroups dispatched in address order)
__global__ __launch_bounds__(256) void copy_flat(const V* __restrict__ in, V* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = in[i];
}
// B: U pieces per thread, block owns a contiguous tile of 256*U pieces, grid = n/(256 U)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_tile(const V* __restrict__ in, V* __restrict__ o, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    V v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(&in[base + u * 256]) : in[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n) {
            if (NT) __builtin_nontemporal_store(v[u], &o[base + u * 256]);
            else o[base + u * 256] = v[u];
        }
}
// C: grid-stride with a fixed grid (what round 1's kernels do)
__global__ __launch_bounds__(256) void copy_stride(const V* __restrict__ in, V* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = in[i];
}
// D: read only / write only
template <int U>
__global__ __launch_bounds__(256) void read_tile(const V* __restrict__ in, size_t n, double* out) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    double acc = 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n) { V v = in[base + u * 256]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void write_tile(V* __restrict__ o, size_t n, double val) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    V v = {val, val};
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n) o[base + u * 256] = v;
}
// E: copy with the Gauss-Bernoulli arithmetic per component (one exp, one reciprocal): does it hide?
template <int U>
__global__ __launch_bounds__(256) void gb_tile(const V* __restrict__ in, V* __restrict__ o, size_t n, double c1, double c0,
                                               double inv_a, double vinv) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    V v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n) v[u] = in[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n) {
            V w;
            { double b = v[u].x; double xi = b * b * c1 + c0; double s = 1.0 / (1.0 + exp(-xi)); w.x = s * b * inv_a * vinv - b; }
            { double b = v[u].y; double xi = b * b * c1 + c0; double s = 1.0 / (1.0 + exp(-xi)); w.y = s * b * inv_a * vinv - b; }
            o[base + u * 256] = w;
        }
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    double* out; hipMalloc(&out, 1 << 20);
    for (size_t mb : {512, 4096}) {
        size_t bytes = mb << 20;
        V *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
        hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
        size_t n = bytes / 16;
        printf("---- arrays of %zu MiB (copy moves 2x that)\n", mb);
        float ms;
#define REP(name, moved, ...) ms = timeit([&] { __VA_ARGS__; }, 10); printf("%-34s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, (moved) / ms / 1e6);
        REP("copy flat (1 piece/thread)", 2.0 * bytes, hipLaunchKernelGGL(copy_flat, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n))
        REP("copy tile U=2", 2.0 * bytes, hipLaunchKernelGGL((copy_tile<2, false>), dim3((n + 511) / 512), dim3(256), 0, 0, a, b, n))
        REP("copy tile U=4", 2.0 * bytes, hipLaunchKernelGGL((copy_tile<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n))
        REP("copy tile U=8", 2.0 * bytes, hipLaunchKernelGGL((copy_tile<8, false>), dim3((n + 2047) / 2048), dim3(256), 0, 0, a, b, n))
        REP("copy tile U=2 nontemporal", 2.0 * bytes, hipLaunchKernelGGL((copy_tile<2, true>), dim3((n + 511) / 512), dim3(256), 0, 0, a, b, n))
        REP("copy tile U=4 nontemporal", 2.0 * bytes, hipLaunchKernelGGL((copy_tile<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n))
        for (int blocks : {2048, 8192, 32768})  {
            char nm[64]; snprintf(nm, sizeof nm, "copy grid-stride %d blocks", blocks);
            REP(nm, 2.0 * bytes, hipLaunchKernelGGL(copy_stride, dim3(blocks), dim3(256), 0, 0, a, b, n))
        }
        REP("read tile U=4", 1.0 * bytes, hipLaunchKernelGGL((read_tile<4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, n, out))
        REP("read tile U=8", 1.0 * bytes, hipLaunchKernelGGL((read_tile<8>), dim3((n + 2047) / 2048), dim3(256), 0, 0, a, n, out))
        REP("write tile U=4", 1.0 * bytes, hipLaunchKernelGGL((write_tile<4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, b, n, 1.5))
        REP("gb-arith copy tile U=2", 2.0 * bytes, hipLaunchKernelGGL((gb_tile<2>), dim3((n + 511) / 512), dim3(256), 0, 0, a, b, n, 0.2, -3.0, 0.4, 3.1))
        REP("gb-arith copy tile U=4", 2.0 * bytes, hipLaunchKernelGGL((gb_tile<4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n, 0.2, -3.0, 0.4, 3.1))
        REP("hipMemcpy D2D", 2.0 * bytes, (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0))
        hipFree(a); hipFree(b);
    }
    return 0;
}
