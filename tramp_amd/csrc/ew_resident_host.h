// ew_resident_host.h - host side of the resident single-pass updates (ew_resident.h): eligibility, workspace, launch.
// The kernels are instantiated in ew_resident_{prior,lik,pwl}.hip (one translation unit per factor family: each
// instantiation unrolls its pipeline ring three times over, seconds of compile time apiece); ew_kernels.hip sees only
// the declaration of run_resident and the trait that says which factors have the resident form.
#pragma once
#include "ew_ops.h"

constexpr int kResNotEligible = 1000;      // run_resident: "take the tile kernels" (not an error)

// The shape the pipeline is built for (tools/probes/probe_resident.hip on MI355X, 64 x 1e6: U = 4 pieces per thread and
// tile, loads one iteration ahead, update three iterations behind, two 256-thread workgroups per CU: Gauss-Bernoulli
// 207 us f64 / 94 us f32; deeper rings spill, U = 8 with one workgroup per CU is 5-10 % slower, U = 2 15 % slower).
constexpr int kResU = 4, kResDL = 1, kResDA = 3;

// Threads per workgroup, 16-byte pieces per thread and tile, workgroups per CU, extra kernel flags.  Measured for the
// two-region channels K7 in round 5 and dropped: 512 threads x 2 pieces (four waves per SIMD, <= 128 registers: 500-1300
// spill instructions, 0.8 ms), the same with the target message read a second time instead of held in the ring
// (kResReloadB: worse), 256 threads x 8 pieces on one workgroup per CU (512 registers: still spills).  Their LEAN form
// (no control flow in the pipelined body) fits the standard shape.
// FLAGS: extra kernel flags (ew_resident.h); 64 = kResReloadB, the target message is read a second time instead of being
// held in registers.
template <typename Op> struct ResShape { static constexpr int U = kResU, NT = kBlock, FLAGS = 0, BPC = 2; };
template <typename T, int KIND, bool FWD, bool UNI, bool LEAN> struct ResShape<PwlOp<T, KIND, FWD, UNI, LEAN>> {
    static constexpr int U = sizeof(T) == 4 ? 2 : kResU, NT = kBlock, FLAGS = sizeof(T) == 4 ? 8 : 0, BPC = sizeof(T) == 4 ? 3 : 2;
};

// (the f32 lean likelihoods on three workgroups per CU of half-size tiles, the f32 K7 shape: 166-168 us against 164-170 us
// at 64 x 1e6 - no difference, the standard shape stays)
// FLAGS = 8 (kResNtLoad): non-temporal loads of the messages.  Measured per op at 64 x 1e6 (steady clocks): the updates
// with TWO input streams gain (abs likelihood 270 -> 256 us f64, 143 -> 138 us f32; Relu likelihood 291 -> 280 / 154 -> 147 us;
// K7 f32 1-3 %), the one-input priors do not (f64: no change; f32: 85 -> 101 us - their 256 MB input is partly resident
// in the Infinity Cache from the call before, which a non-temporal load gives up), K7 f64 and the bit-packed sign
// likelihood do not either.
template <typename T> struct ResShape<AbsLikOp<T>> { static constexpr int U = kResU, NT = kBlock, FLAGS = 8, BPC = 2; };
template <typename T, bool L> struct ResShape<ReluLikOp<T, L>> { static constexpr int U = kResU, NT = kBlock, FLAGS = 8, BPC = 2; };
template <> struct ResShape<SgnLikOp<double, true>> { static constexpr int U = kResU, NT = kBlock, FLAGS = 8, BPC = 2; };

template <typename Op> struct ResidentOp { static constexpr bool value = false; };
template <typename T> struct ResidentOp<GbOp<T>> { static constexpr bool value = true; };
template <typename T> struct ResidentOp<L1Op<T>> { static constexpr bool value = true; };
template <typename T> struct ResidentOp<AbsLikOp<T>> { static constexpr bool value = true; };
template <typename T> struct ResidentOp<BinaryOp<T>> { static constexpr bool value = true; };
template <typename T> struct ResidentOp<GmOp<T, 2>> { static constexpr bool value = true; };     // straight-line forms only
template <typename T> struct ResidentOp<GmOp<T, 3>> { static constexpr bool value = true; };
template <typename T, bool L> struct ResidentOp<ReluLikOp<T, L>> { static constexpr bool value = true; };
// (the lean forms - the two-region channels K7, the sign likelihood with its observation unpacked or packed - are
//  launched by their entry points in ew_kernels.hip, which also run the redo of their unfinished instances)

// The shape test (instances of at least one tile that fit the pipeline's window, enough tiles for four rounds, 16-byte
// rows).  Shared with the producers of prior_hint.h: an FFT epilogue whose consumer takes the resident form leaves no
// moments (they cost it +45...65 us at 64 x 1e6, more than the resident update costs over the plain apply pass).
bool resident_covers(const tramp_ctx* ctx, size_t n, size_t batch, size_t elem_size);
// the same test for a pipeline of `bpc` workgroups per CU and tiles of `tile_pieces` 16-byte pieces
bool resident_covers_shape(const tramp_ctx* ctx, size_t n, size_t batch, size_t elem_size, size_t tile_pieces, int bpc);

// TRAMP_OK: the update ran (launched) in the resident form; kResNotEligible: shape or outputs it does not cover;
// anything else: an error, ctx->err is set.
template <typename T, typename Op>
int run_resident(tramp_ctx* ctx, const Op& op, size_t n, size_t batch, const void* in0, const void* in1,
                 const double* a_in_d, const MsgOut& out, const void* bits = nullptr, size_t bits_stride = 0);
