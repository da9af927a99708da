// Lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of D[4x4] += A[4x4] B[4x4]) found empirically:
// for every pair (p, q) run the MFMA with A one-hot at lane p and B one-hot at lane q and
// record which output lanes are non-zero.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned long long* tab) {
    const int l = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            double a = (l == p) ? 1.0 : 0.0, b = (l == q) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (l == 0) tab[p * 64 + q] = m;
        }
}
int main() {
    unsigned long long* d; static unsigned long long h[64 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // D lane o receives sum over pairs (p, q): list them
    for (int o = 0; o < 64; ++o) {
        printf("D lane %2d <-", o);
        for (int p = 0; p < 64; ++p)
            for (int q = 0; q < 64; ++q)
                if (h[p * 64 + q] >> o & 1) printf(" (a%d,b%d)", p, q);
        printf("\n");
    }
    return 0;
}
