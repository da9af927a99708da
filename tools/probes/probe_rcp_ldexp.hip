// Accuracy of v_rcp_f64 (+ Newton steps) and range behaviour of a single v_ldexp_f64 on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_rcp_ldexp.hip -o /tmp/probe_rcp && /tmp/probe_rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k_rcp(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r2[i] = r;
}
__global__ void k_ldexp(const double* p, const int* e, double* one, double* two, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    one[i] = ldexp(p[i], e[i]);
    int h = e[i] / 2;
    two[i] = ldexp(ldexp(p[i], h), e[i] - h);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i) x[i] = (i & 1) ? 1.0 + rnd() : std::exp(40.0 * (rnd() - 0.5));
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k_rcp<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl((a[i] - t) / t));
        e1 = fmax(e1, (double)fabsl((b[i] - t) / t));
        e2 = fmax(e2, (double)fabsl((c[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err: raw %.3e   1 Newton %.3e (%.2f ulp)   2 Newton %.3e (%.2f ulp)\n", e0, e1, e1 / 1.11e-16, e2,
           e2 / 1.11e-16);
    // ldexp
    std::vector<double> p(4096), o1(4096), o2(4096);
    std::vector<int> e(4096);
    for (int i = 0; i < 4096; ++i) { p[i] = 0.70 + 0.72 * rnd(); e[i] = -1100 + (int)(2140 * rnd()); }
    int* de; hipMalloc(&de, 4096 * 4);
    hipMemcpy(dx, p.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipMemcpy(de, e.data(), 4096 * 4, hipMemcpyHostToDevice);
    k_ldexp<<<16, 256>>>(dx, de, d0, d1, 4096);
    hipMemcpy(o1.data(), d0, 4096 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o2.data(), d1, 4096 * 8, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int i = 0; i < 4096; ++i) {
        double ref = std::ldexp(p[i], e[i]);
        if (o1[i] != ref) { if (bad1 < 5) printf("  single ldexp differs: p=%.17g e=%d got %.17g ref %.17g\n", p[i], e[i], o1[i], ref); ++bad1; }
        if (o2[i] != ref) ++bad2;
    }
    printf("ldexp over e in [-1100, 1040): single-step mismatches %d, two-step mismatches %d (of 4096, vs host ldexp)\n", bad1, bad2);
    return 0;
}
