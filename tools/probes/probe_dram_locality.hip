// probe_dram_locality.hip - what a DRAM page visit costs: the same bytes read in the order of the row-dot f64 pass
// (a wave: 16 rows 800 KB apart, one 128-byte line of each per step, walking along the rows) and as a contiguous stream
// (a wave: 2 KiB per step).  Loops for ~8 s per pattern so that rocm-smi can be sampled beside it (tools/power_dram.sh).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/probe_dram_locality.hip -o tools/probes/build/probe_dram_locality
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

typedef double v2d __attribute__((ext_vector_type(2)));

// R x C doubles, row-major.  MODE 0: workgroup = 128 rows x a k range (4 waves x 2 x 16 rows; lane (j, q) reads 32
// adjacent bytes of row j per step): the row-dot kernel's loads without its arithmetic.  MODE 1: the same bytes as one
// contiguous run per wave.
template <int MODE>
__global__ __launch_bounds__(256) void reader(const double* __restrict__ mat, size_t R, size_t C, size_t steps, double* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const size_t rb = blockIdx.x / 8, split = blockIdx.x % 8;
    v2d acc = {0, 0};
    if (MODE == 0) {
        const size_t row0 = rb * 128 + wave * 32;
        const double* p0 = mat + (row0 + j) * C + split * steps * 16 + q * 4;
        const double* p1 = p0 + 16 * C;
        for (size_t s = 0; s < steps; ++s) {
            acc += *reinterpret_cast<const v2d*>(p0 + s * 16) + *reinterpret_cast<const v2d*>(p0 + s * 16 + 2);
            acc += *reinterpret_cast<const v2d*>(p1 + s * 16) + *reinterpret_cast<const v2d*>(p1 + s * 16 + 2);
        }
    } else {
        // the same 4 KiB per wave and step, contiguous: workgroup (rb, split) owns a contiguous chunk of 128 x steps x 16 doubles
        const double* base = mat + ((rb * 8 + split) * 4 + wave) * (steps * 512) + lane * 2;
        for (size_t s = 0; s < steps; ++s) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += *reinterpret_cast<const v2d*>(base + s * 512 + u * 128);
        }
    }
    if (acc[0] + acc[1] == 1.2345e300) sink[0] = acc[0];
}

int main(int argc, char** argv) {
    const size_t R = 49920, C = 100000;          // 390 row blocks of 128
    const double secs = argc > 1 ? atof(argv[1]) : 8.0;
    double *mat, *sink;
    hipMalloc(&mat, R * C * 8); hipMalloc(&sink, 8);
    hipMemset(mat, 0, R * C * 8);
    const size_t steps = C / 16 / 8;             // per split
    for (int mode = 0; mode < 2; ++mode) {
        auto launch = [&] {
            if (mode == 0) hipLaunchKernelGGL(reader<0>, dim3((unsigned)(R / 128 * 8)), dim3(256), 0, 0, mat, R, C, steps, sink);
            else hipLaunchKernelGGL(reader<1>, dim3((unsigned)(R / 128 * 8)), dim3(256), 0, 0, mat, R, C, steps, sink);
        };
        launch(); hipDeviceSynchronize();
        printf("[pattern %s] started\n", mode == 0 ? "row-dot (one 128-byte line per row and step)" : "contiguous (4 KiB per wave and step)"); fflush(stdout);
        auto t0 = std::chrono::steady_clock::now();
        size_t n = 0;
        double el = 0;
        do {
            for (int i = 0; i < 20; ++i) launch();
            hipDeviceSynchronize();
            n += 20;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } while (el < secs);
        const double bytes = (double)(R / 128 * 8) * 4 * steps * 4096.0;
        printf("[pattern %d] %.3f ms per sweep of %.1f GB: %.0f GB/s\n", mode, el / n * 1e3, bytes / 1e9, bytes * n / el / 1e9); fflush(stdout);
        // idle gap between the patterns
        auto t1 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() < 3.0) {}
    }
    return 0;
}
