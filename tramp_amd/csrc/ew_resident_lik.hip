// ew_resident_lik.hip - resident single-pass updates of the output denoisers (K4 sign, K5 abs likelihood, the (leaky) relu likelihood).
#include "ew_resident_impl.h"
RES_INSTANTIATE(double, SgnLikOp<double, true>)
RES_INSTANTIATE(float, SgnLikOp<float, true>)
RES_INSTANTIATE(double, AbsLikOp<double>)
RES_INSTANTIATE(float, AbsLikOp<float>)
RES_INSTANTIATE(double, SgnLikBitsOp<double, true>)
RES_INSTANTIATE(float, SgnLikBitsOp<float, true>)
RES_INSTANTIATE(double, ReluLikOp<double>)
RES_INSTANTIATE(float, ReluLikOp<float>)
RES_INSTANTIATE(double, ReluLikOp<double, true>)
RES_INSTANTIATE(float, ReluLikOp<float, true>)
