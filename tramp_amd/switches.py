"""Every environment variable the library reads, with its default and what it is for - the ONE table.

None of them is needed to run anything: the defaults are what `bench.py` and the test suite run (bench.py records the
TRAMP_* variables it finds set as ``env_overrides``; tests/test_cabi_and_host.py checks that every getenv / environ.get in
the sources is listed here and nothing else is).  Kinds: `select` chooses between implementations that are all parity
tested, `tune` moves a launch-shape parameter whose default was measured, `debug` is for diagnosis only.
"""

SWITCHES = {
    # name: (default, kind, meaning)
    "TRAMP_HIP_LIB": ("tramp_amd/libtramp_hip.so", "select", "path of the C-ABI library to load (an experimental build)"),
    # componentwise updates (csrc/ew_kernels.hip, ew_resident.h)
    "TRAMP_EW_RESIDENT": ("1", "select", "0: never take the resident single-pass pipeline (the tile kernels' two passes instead); ROCm's HSA_CU_MASK / ROC_GLOBAL_CU_MASK, if set, have the same effect (the co-residency check cannot see a CU mask)"),
    "TRAMP_EW_RESIDENT_K7": ("1", "select", "0: the Sgn / Abs / Relu channels stay on the tile kernels' two passes"),
    "TRAMP_EW_COOP": ("1", "select", "0: a few large instances take the tile kernels' three launches instead of the one-launch cooperative form"),
    "TRAMP_EW_LEAN_ADAPT": ("1", "select", "0: a lean resident K7 update that had to redo more than a quarter of its instances does not send the next 15 calls to the tile kernels"),
    "TRAMP_EW_REDO_TPB": ("16", "tune", "tiles per workgroup in the redo of a lean resident update's flagged instances (64 before round 6: 0.19 ms with 2-3 of 64 instances flagged, 0.07 ms at 16)"),
    "TRAMP_EW_RESIDENT_TIMEOUT_MS": ("2000", "tune", "deadline of a resident kernel: hand-offs still missing then are given up (NaN update, tile kernels afterwards)"),
    "TRAMP_EW_COOP_DBG_SKIP": (None, "debug", "workgroup index + 1 of instance 0 whose block sum a one-launch update never publishes (forces its give-up path in tests)"),
    "TRAMP_EW_RESIDENT_DBG_SKIP": (None, "debug", "tile index + 1 whose sum a resident kernel never publishes (forces the give-up path in tests)"),
    "TRAMP_EW_RESIDENT_MIN_ROUNDS": ("4", "tune", "rounds of 2 x CUs - 8 tiles below which the pipeline is not worth its fill and drain"),
    "TRAMP_PRIOR_FUSE": ("1", "select", "0: producers never leave prior / likelihood moments (csrc/prior_hint.h)"),
    # dense linear channel (csrc/linear.hip, gemm_mfma.hip)
    "TRAMP_MATVEC_ALGO": ("0", "debug", "force a matrix-pass engine: 1 GEMV (VALU), 2 f64 / f32 MFMA, 3 the int8-digit copy; 0 = the digit copy if attached, else by batch size"),
    "TRAMP_MATVEC_DIGITS": ("auto", "select", "int8 copies of a LinearChannel's factors (csrc/gemm_sliced.h) in radix-256 digits: auto = the chain scheduler attaches 7 (f64) / 4 (f32) for ensembles of >= 3 instances when memory allows; 0 never; 7, 8, 6, 4 at construction"),
    "TRAMP_MFMA_SPLIT": ("0", "tune", "split-K factor of the MFMA passes; 0 = chosen from the grid size"),
    "TRAMP_MFMA_LDS_PAD": ("0", "tune", "extra dynamic LDS per workgroup of the MFMA passes (occupancy experiments)"),
    "TRAMP_MFMA_CHUNK_2D": ("0", "tune", "1: 64-instance chunks on the grid's y axis instead of interleaved in x"),
    "TRAMP_ROTATE_INPUT": ("1", "select", "0: keep the linear channel's input variable in the canonical basis (algos.py)"),
    "TRAMP_ROTATE_OUTPUT": ("1", "select", "0: keep the linear channel's output variable in the canonical basis (algos.py)"),
    "TRAMP_LSOLVE_SCRATCH_MB": ("2048", "tune", "MB of N x N systems SolveLinearChannel factorises per strided-batched call"),
    # factorisation (csrc/factorize.hip)
    "TRAMP_EIG_CAP": ("46336", "tune", "largest symmetric eigenproblem handed to rocSOLVER syevd whole; above it the spectral split"),
    "TRAMP_EIG_SUBCAP": ("16384", "tune", "halves of a forced split are split again down to this size"),
    "TRAMP_EIG_METHOD": ("auto", "select", "auto: symmetric eigenproblems of order 6144 .. 92672 go through the own tridiagonalisation (csrc/tridiag_eig.h), rocSOLVER stedc on the tridiagonal matrix (one Cuppen merge of two halves above 46336) and the blocked back-transformation, memory permitting, the others through `split`; tridiag: the same from order 64 on; split: rocSOLVER syevd below TRAMP_EIG_CAP, spectral divide and conquer above (rounds 2-5)"),
    "TRAMP_TD_BT_PANEL": ("2048", "tune", "reflectors per panel of the tridiagonal path's back-transformation (32 .. 8192)"),
    "TRAMP_TD_MERGE_DENSE": (None, "debug", "set: the merge of the tridiagonal path forms the dense basis blockdiag(Q1, Q2) even when no deflation rotation pairs columns of the two halves (the fallback, reachable in tests)"),
    "TRAMP_TD_MERGE_ABOVE": (None, "debug", "order above which the tridiagonal path merges two stedc halves although one stedc call would do (tests reach the merge at small orders)"),
    "TRAMP_TD_PROFILE": (None, "debug", "set: the tridiagonalisation synchronises after every launch and prints the time of each kernel family"),
    "TRAMP_EIG_VERBOSE": (None, "debug", "set: phase timings of the factorisation on stderr"),
    # structured channels (csrc/conv.hip, fft_pow2.hip)
    "TRAMP_CONV_FFT": ("auto", "select", "own | rocfft: FFT engine of the convolution / gradient channels"),
    "TRAMP_CONV_FFT_MIN_POINTS": ("0", "tune", "signals shorter than this go to rocFFT under TRAMP_CONV_FFT=auto (32768 was the measured break-even of round 3; 0 = the own engine wherever it has a plan)"),
    "TRAMP_CONV_IPB": ("0", "tune", "images per workgroup of the one-launch small-image kernel; 0 = chosen from the shape"),
    "TRAMP_DFT_FFT": ("own", "select", "own | rocfft: FFT engine of DFTChannel (own: natural-order mode of the own engine where it applies, else rocFFT)"),
    "TRAMP_FFT_BLOCK_THREADS": ("0", "tune", "threads of the own engine's block pass; 0 = by length"),
    "TRAMP_FFT_STRIDED_THREADS": ("0", "tune", "threads of the own engine's strided pass; 0 = by length"),
    "TRAMP_FFT_AXIS_THREADS": ("0", "tune", "threads of the own engine's long-axis pass (N-D transforms)"),
    "TRAMP_FFT_AXIS_DIV": ("0", "tune", "columns per workgroup divisor of the long-axis pass"),
    "TRAMP_FFT_IMAGE": ("1", "select", "0: never take the one-launch small-image transform"),
    # state evolution (csrc/se_kernels.hip)
    "TRAMP_SE_2D": ("1", "select", "0: brute-force grid instead of the adaptive Gauss-Legendre panels for the 2-D averages"),
    "TRAMP_SE_PANELS2D": ("768", "tune", "panels per axis of the 2-D state-evolution quadrature (8 ... 4096)"),
    # tree scheduler (tree.py)
    "TRAMP_TREE_ALIAS": ("1", "select", "0: no aliasing of the tree scheduler's message buffers"),
    "TRAMP_TREE_MONITOR": ("device", "select", "host: the tree scheduler's stopping rule on the host"),
}


def overrides(environ=None):
    "the TRAMP_* variables of `environ` (default os.environ) that the library reads: {} means 'all defaults'"
    import os
    env = os.environ if environ is None else environ
    return {k: env[k] for k in sorted(SWITCHES) if k in env}


def markdown_table():
    rows = ["| variable | default | kind | meaning |", "|---|---|---|---|"]
    for k, (d, kind, doc) in SWITCHES.items():
        rows.append(f"| `{k}` | {'unset' if d is None else '`' + str(d) + '`'} | {kind} | {doc} |")
    return "\n".join(rows)
