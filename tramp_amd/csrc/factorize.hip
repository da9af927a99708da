// factorize.hip - device-side thin SVD for LinearChannel setup.  Filled in below.
#include "common.h"

extern "C" int tramp_linear_factorize(tramp_ctx* ctx, int, size_t, size_t, const void*, void*, void*, double*,
                                      size_t*) {
    return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_linear_factorize", "not built");
}
