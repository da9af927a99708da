// linear.h - internal types shared by linear.hip and gemm_mfma.hip.
#pragma once
#include "common.h"
#include "prior_hint.h"

struct tramp_linear {
    int dtype;
    size_t M, N, k, rank;
    const void* Uk;      // [M, k]
    const void* Vt;      // [k, N]
    const double* sing;  // [k]
    double* s2;          // [k] spectrum = sing^2 (owned)
    void* work;          // owned: t, u, w [batch, k] + a_raw[batch]
    size_t work_batch;
    size_t proj_batch;   // batch of the projections currently held in t / u
    void* t;
    void* u;
    void* w;
    void* tc;            // projections of the constant part of a damped constant message
    void* uc;            //   (tramp_linear_project_affine)
    bool tc_valid, uc_valid;
    double* a_raw;
    int rot;             // bit 0: output side in the left singular basis (Uk acts as the identity, k == M);
                         // bit 1: input side in the right singular basis (Vt acts as the identity, k-vectors)
};

// Epilogue applied to every output element acc = (mat x)[b, idx]:
//   mode 0: out = acc
//   mode 1: r = acc (+ b_in/a_in); out = r (a_in + a_new_raw) - b_in, optionally damped
struct LinEpi {
    int mode;
    int add_b_over_a;
    const double* a_in;       // [batch]
    const double* a_new_raw;  // [batch]
    const void* b_in;         // [batch, len]
    double damping;
    const void* b_old;        // [batch, len] or null
    void* out;                // [batch, len]
    void* r_out;              // [batch, len] or null
    PriorHintDev hint;        // kind != 0: also leave sum v_i partials of the consuming prior (prior_hint.h)
};

// If the context holds a hint for the buffer this epilogue writes (same length and batch), arm the
// epilogue with it.  `nblk` = blocks per instance of the kernel that will apply the epilogue.
bool lin_arm_hint(tramp_ctx* ctx, LinEpi& epi, size_t len, size_t batch, int nblk);

int lin_square(tramp_ctx* ctx, const double* in, double* out, size_t n);
// kernel K9 on an explicit list of squared singular values (solve.hip: the precompute_svd=False channel)
int lin_variance_launch(tramp_ctx* ctx, const double* s2, int k, int rank, double Nx, double Nz, int dir, size_t batch,
                        const double* az_d, const double* ax_d, double damping, const double* a_old_d, double* v_d,
                        double* a_new_d, double* a_raw_d, double* n_eff_d);

// MFMA GEMM path (gemm_mfma.hip)
bool gemm_mfma_supported(int dtype, bool nt, size_t R, size_t C, size_t batch);
int gemm_mfma(tramp_ctx* ctx, int dtype, bool nt, size_t R, size_t C, size_t batch, const void* mat_d,
              const void* x_d, const LinEpi& epi);
// split-K fold + message-update epilogue of gemm_mfma.hip on partials [nsplit][batch][len] (shared with gemm_sliced.hip)
int mfma_fold(tramp_ctx* ctx, int dtype, size_t len, size_t batch, int nsplit, const double* partials, const LinEpi& epi);

// int8-digit engine (gemm_sliced.hip): runs the pass if a digit copy of mat_d is attached in the wanted form
int gemm_sliced_try(tramp_ctx* ctx, int dtype, bool nt, size_t R, size_t C, size_t batch, const void* mat_d, const void* x_d,
                    const LinEpi& epi, bool* taken);
void gemm_sliced_ctx_destroy(tramp_ctx* ctx);
