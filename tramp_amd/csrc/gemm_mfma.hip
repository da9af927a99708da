// gemm_mfma.hip - MFMA (matrix-core) kernels of the dense linear channel for
// ensembles of instances (batch > 8).  Placeholder until the tiled kernels land:
// reports "unsupported" so that the dispatcher streams with the VALU kernels.
#include "linear.h"

bool gemm_mfma_supported(int, bool, size_t, size_t, size_t) { return false; }

int gemm_mfma(tramp_ctx* ctx, int, bool, size_t, size_t, size_t, const void*, const void*, const LinEpi&) {
    return tramp_fail(ctx, TRAMP_ENOTSUP, "gemm_mfma", "not built");
}
