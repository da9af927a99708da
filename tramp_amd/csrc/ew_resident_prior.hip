// ew_resident_prior.hip - resident single-pass updates of the vector denoisers (K1 Gauss-Bernoulli, K2 L1).
#include "ew_resident_impl.h"
RES_INSTANTIATE(double, GbOp<double>)
RES_INSTANTIATE(float, GbOp<float>)
RES_INSTANTIATE(double, L1Op<double>)
RES_INSTANTIATE(float, L1Op<float>)
RES_INSTANTIATE(double, BinaryOp<double>)
RES_INSTANTIATE(float, BinaryOp<float>)
RES_INSTANTIATE(double, GmOp<double, 2>)
RES_INSTANTIATE(float, GmOp<float, 2>)
RES_INSTANTIATE(double, GmOp<double, 3>)
RES_INSTANTIATE(float, GmOp<float, 3>)
RES_INSTANTIATE(double, HalfLinePriorOp<double, true>)
RES_INSTANTIATE(float, HalfLinePriorOp<float, true>)
