// fft_pow2.h - internal interface of the two-pass power-of-two complex FFT (fft_pow2.hip).
#pragma once
#include "common.h"

struct Fft2Plan {
    int dtype;
    int q;        // log2 of the transform length h
    int lb_log;   // log2 of the block length handled by the second pass
    void* tw_hi;  // w_h^(1024 i)
    void* tw_lo;  // w_h^i, i < 1024
};

// log2 of the block length for h = 2^q, or -1 when the length is not supported (q in 12..21)
int fft2_lb_log(int q);
// frequency held by position p of a forward-transformed array
size_t fft2_kfreq(int q, int lb_log, size_t p);
int fft2_plan_create(tramp_ctx* ctx, int dtype, int q, Fft2Plan** out);
void fft2_plan_destroy(Fft2Plan* pl);
// natural order -> permuted spectrum (out of place; in and out may coincide)
int fft2_forward(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, const void* in, void* out);
// permuted spectrum (overwritten) -> natural order, unnormalised (x h)
int fft2_inverse(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, void* spectrum, void* out);
