// se_kernels.hip - state-evolution quadratures (K13, K14).  Filled in below.
#include "common.h"
#include "tramp_math.h"

extern "C" int tramp_se_gb_prior(tramp_ctx* ctx, size_t, const double*, const double*, const double*,
                                 const double*, double*) {
    return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_se_gb_prior", "not built");
}
extern "C" int tramp_se_sgn_lik(tramp_ctx* ctx, size_t, const double*, const double*, double*) {
    return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_se_sgn_lik", "not built");
}
extern "C" int tramp_se_abs_lik(tramp_ctx* ctx, size_t, const double*, const double*, double*) {
    return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_se_abs_lik", "not built");
}
extern "C" int tramp_se_pwl_channel(tramp_ctx* ctx, int, int, size_t, const double*, const double*,
                                    const double*, double*) {
    return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_se_pwl_channel", "not built");
}
