// ew_resident_pwl.hip - resident single-pass updates of the two-region channels K7 (Sgn / Abs / Relu, both directions):
// the moment form of tramp_math.h (pwl_moment_form) on the one-value-per-point erfcx table (13 KiB beside the 64 KiB of parked
// posterior means per workgroup), 512 threads x 2 pieces per tile (ResShape).
#include "ew_resident_impl.h"
#define RES_PWL(T, KIND)                                  \
    RES_INSTANTIATE(T, PwlOp<T, KIND, true, true, true>)       \
    RES_INSTANTIATE(T, PwlOp<T, KIND, false, true, true>)
RES_PWL(double, 0)
RES_PWL(double, 1)
RES_PWL(double, 2)
RES_PWL(float, 0)
RES_PWL(float, 1)
RES_PWL(float, 2)
