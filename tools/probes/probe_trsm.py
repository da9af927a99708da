"""Rate of rocBLAS dtrsm against dgemm at the shapes the blocked Cholesky solve of the spectral split uses
(csrc/factorize.hip: bpotrs_upper, kCholBlock x nrhs panels): python tools/probes/probe_trsm.py"""
import ctypes as C
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T

ctx = T.default_context()
rb = C.CDLL("librocblas.so")
hip = C.CDLL("libamdhip64.so")
h = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(h)) == 0
one = C.c_double(1.0)
zero = C.c_double(0.0)
NONE, TRANS = 111, 112
LEFT, UPPER, NONUNIT = 141, 121, 131


def sync():
    hip.hipDeviceSynchronize()


def timeit(fn, reps=3):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


for ib, nrhs in ((4096, 16384), (2048, 16384), (1024, 16384), (4096, 50000 // 4)):
    ld = 50000
    U = ctx.array((np.triu(np.random.RandomState(0).standard_normal((ib, ib))) + 40 * np.eye(ib)).T.copy())   # column-major upper
    Ubig = ctx.empty((ib, ld))        # ld x ib column-major storage: leading dimension ld
    X = ctx.randn((nrhs, ld), np.float64, seed=1)
    Y = ctx.randn((nrhs, ld), np.float64, seed=2)
    t_trsm = timeit(lambda: rb.rocblas_dtrsm(h, LEFT, UPPER, TRANS, NONUNIT, ib, nrhs, C.byref(one), C.c_void_p(U.ptr), ib,
                                             C.c_void_p(X.ptr), ld))
    t_gemm = timeit(lambda: rb.rocblas_dgemm(h, TRANS, NONE, ib, nrhs, ib, C.byref(one), C.c_void_p(U.ptr), ib,
                                             C.c_void_p(X.ptr), ld, C.byref(zero), C.c_void_p(Y.ptr), ld))
    fl = ib * ib * nrhs
    print(f"ib={ib} nrhs={nrhs}: dtrsm {t_trsm*1e3:.2f} ms = {fl/t_trsm/1e12:.1f} TFLOP/s (ib^2 nrhs flop); "
          f"dgemm of the same triangle as a dense block {t_gemm*1e3:.2f} ms = {2*fl/t_gemm/1e12:.1f} TFLOP/s", flush=True)
