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
    size_t rows_per_inst; // transforms per problem instance (N-D: the rows of one signal); 0 counts as 1
};

// false when h is not 2^a 5^b or does not split into a [G <= 320] x [512 <= LB <= 4096] pair
bool fft2_shape(size_t h, Fft2Shape* sh);
// frequency held by position p of a forward-transformed array
size_t fft2_kfreq(const Fft2Shape& sh, size_t p);
int fft2_plan_create(tramp_ctx* ctx, int dtype, size_t h, Fft2Plan** out, bool block_only = false);
void fft2_plan_destroy(Fft2Plan* pl);
// natural order -> permuted spectrum (out of place; in and out may coincide)
int fft2_forward(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, const void* in, void* out);
// permuted spectrum (overwritten) -> natural order, unnormalised (x h)
int fft2_inverse(tramp_ctx* ctx, const Fft2Plan* pl, size_t batch, void* spectrum, void* out,
                 const Fft2Epi* epi = nullptr);

// ---- N-dimensional transforms (2-D / 3-D convolution and gradient channels): the packed real signal is a complex
// array [d0][d1][hl] (hl = n_last / 2); the last axis is transformed by the block pass (one pass, several rows per
// workgroup) or the two-pass engine above, every outer axis by one or two STRIDED passes of the same tile kernel
// (rows of the tile are lines of the inner axes).  Each axis keeps its own permuted order: position p of axis a holds
// frequency fftnd_kfreq(plan, a, p).
struct FftAxisPass {
    int G, Q, nA;       // tile rows; tile row g is array row a * G * Q + g * Q + q of the axis (q < Q, a < nA)
    size_t W, n;        // complex elements per array row (the inner axes together); axis length
    int nst, r[8], sub[8], off[8], ntw;
    size_t mul[8];      // twiddle of stage s, leg offset bl: w_n^((bl * Q + q) * mul[s])
    int axis;
};

struct FftNdPlan {
    int dtype, ndim;
    size_t dims[3];     // complex dims: outer axes then hl
    size_t h;           // their product
    Fft2Plan* row;      // last axis (shape.G == 1: block pass only)
    int npass;
    FftAxisPass pass[4];
    void* tw_hi[2];     // two-level twiddle tables of the outer axes
    void* tw_lo[2];
    int nrad[2], rad[2][16];
    // natural-order mode (fftnd_enable_natural): position -> frequency of the outer axes, frequency -> position of the last
    int* kf_dev[2];
    int* pos_last_dev;
};

// real dims [ndim] (last axis even); false: some axis is not 2^a 5^b / too long / the lines do not tile.
// packed_real = false: the dims are those of a COMPLEX array (DFT channel, complex convolutions), no halving
bool fftnd_supported(int dtype, int ndim, const size_t* real_dims, bool packed_real = true);
int fftnd_plan_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* real_dims, FftNdPlan** out,
                      bool packed_real = true);
void fftnd_plan_destroy(FftNdPlan* pl);
size_t fftnd_kfreq(const FftNdPlan* pl, int axis, size_t p);
int fftnd_forward(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, const void* in, void* out);
// spectrum is overwritten; with an epilogue the last pass stores the message update instead of the mean
int fftnd_inverse(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, void* spectrum, void* out,
                  const Fft2Epi* epi = nullptr);
int fftnd_inverse_signals(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, size_t signals, void* spectrum,
                          const Fft2Epi* epi);
// Natural-order transforms (the DFT channel, whose spectrum IS the message): the un-permutation costs no pass - the last
// pass of every outer axis writes (forward) or reads (inverse) its rows at the place of their frequency, the block pass of
// the last axis permutes inside LDS.  Needs the block-only last axis (n_last <= 4096 complex).  `a` (input, clobbered) and
// `b` are two buffers of batch * h complex; *result is the one that holds the answer.
bool fftnd_natural_ok(const FftNdPlan* pl);
int fftnd_enable_natural(tramp_ctx* ctx, FftNdPlan* pl);
int fftnd_forward_nat(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, void* a, void* b, void** result);
int fftnd_inverse_nat(tramp_ctx* ctx, const FftNdPlan* pl, size_t batch, void* a, void* b, void** result);
// single block pass for h <= 4096 (G = 1)
bool fft2_block_shape(size_t h, Fft2Shape* sh);
