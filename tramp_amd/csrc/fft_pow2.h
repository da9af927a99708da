// fft_pow2.h - internal interface of the two-pass mixed-radix (2^a 5^b) complex FFT (fft_pow2.hip).
#pragma once
#include "common.h"
#include "prior_hint.h"

// h = G * LB; stage lists of the two passes (radices 5 / 4 / 2), with for every stage the stride
// between a butterfly's legs (sub) and the factor that turns an offset into a w_h exponent (mul).
struct Fft2Shape {
    size_t h, LB;
    int G;
    int n1, r1[8], sub1[8];
    size_t mul1[8];
    int n2, r2[8], sub2[8];
    size_t mul2[8];
    // twiddles staged in LDS: per stage one entry per leg offset (pass 2: sub2[s] entries, the same for
    // every block, kept as a table in global memory; pass 1: sub1[s] rows x one line of columns per tile)
    int off1[8], ntw1;    // in rows (multiply by the columns per tile)
    int off2[8], ntw2;
};

struct Fft2Plan {
    int dtype;
    Fft2Shape shape;
    void* tw_hi;  // w_h^(1024 i)
    void* tw_lo;  // w_h^i, i < 1024
    void* tw_blk; // pass 2 stage twiddles w_h^(m mul2[s]) at off2[s] + m
};

// Optional message-update epilogue of the inverse transform (the convolution channel's
//   b_new = r (a_in + a_raw) - b_in, constant damping against b_old, optional copy of r):
// applied to the real pairs as the last pass stores them, instead of a separate streaming kernel.
struct Fft2Epi {
    const double* a_in;   // [batch]
    const double* a_raw;  // [batch]
    const void* b_in;     // [batch, 2 h] reals
    double damping;
    const void* b_old;    // may alias out
    void* out;            // [batch, 2 h]; null: no epilogue (plain store)
    void* r_out;          // optional
    PriorHintDev hint;    // kind != 0: the consumer is a separable prior - leave sum v_i per tile (prior_hint.h)
};

// false when h is not 2^a 5^b or does not split into a [G <= 320] x [512 <= LB <= 4096] pair
bool fft2_shape(size_t h, Fft2Shape* sh);
// frequency held by position p of a forward-transformed array
size_t fft2_kfreq(const Fft2Shape& sh, size_t p);
int fft2_plan_create(tramp_ctx* ctx, int dtype, size_t h, Fft2Plan** out);
void fft2_plan_destroy(Fft2Plan* pl);
// natural order -> permuted spectrum (out of place; in and out may coincide)
int fft2_forward(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, const void* in, void* out);
// permuted spectrum (overwritten) -> natural order, unnormalised (x h)
int fft2_inverse(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, void* spectrum, void* out,
                 const Fft2Epi* epi = nullptr);
