/*
 * deepfluids_hip_debug.h -- tuning / instrumentation entry points of the TUNING build of the library
 * (`make -C deep_fluids_amd/csrc tuning` -> libdeepfluids_hip_tuning.so, compiled with -DDF_TUNING).
 *
 * NOT part of the drop-in boundary: the release library (libdeepfluids_hip.so) does not export these symbols and has
 * no mutable global state.  They exist for the probes under tools/ (cycle profiles and experiment variants of
 * wino3d_kernel, stencil launch knobs).  Everything here is process-global and thread-unsafe by design.
 */
#ifndef DEEPFLUIDS_HIP_DEBUG_H
#define DEEPFLUIDS_HIP_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* wino3d_kernel experiment variant (bits 2..15: the DBG template argument / 4) and slices-per-XCD override (bits 16+) */
void df_debug_set_wino(int v);
/* per-phase cycle counters accumulated by the DBG = 16 variant; out[32]; reset != 0 clears them */
int df_debug_wino_prof(unsigned long long* out, int reset);
/* jacobian3d_fwd launch knobs: non-temporal stores on/off; XCD run length of the block remap */
void df_debug_set_stencil_nt(int v);
void df_debug_set_stencil_group(int v);
void df_debug_set_stencil_lds(int v);      /* 0: register-only adjoints (jacobian3d_bwd_vec_kernel) even where the LDS-staged kernel applies */
/* fused-tail kernel choice: 0 default dispatch, 1 16-byte quad kernels, 2 record-per-lane (12-byte load) kernels */
void df_debug_set_tail(int v);
void df_debug_set_wgrad(int v);            /* wgrad_wxyz_kernel experiments (results wrong by construction): bit 1 = every operand load reads
                                             the cached zero row (no memory latency), 2 = no (z, y) operand combinations, 4 = no x transform; values
                                             1, 2, 3, 6, 7 are instantiated (W = 64 rows) */

/* conv_bf16x3_kernel (the direct split-operand conv) timing-only variants, results wrong by construction: 1 staging loads replaced by zeros,
 * 2 no weight loads, 4 no LDS operand reads, 8 no staging at all (+ sums 3, 6, 10, 14); 0 = production */
void df_debug_set_conv_bf16(int v);
/* The "bf16x3 in the Winograd domain" experiment (wino3d_kernel PREC = 1): arguments as df_wino_pack_weights /
 * df_wino_conv_fwd.  The kernel variant follows the RAW value of df_debug_set_wino: 0 production, 32 xi_x-major MFMA order; timing-only
 * variants (results wrong by construction) 1 no transform, 2 no LDS operand reads, 4 no staging, 8 no weight loads (+ the sums 3, 7, 11, 12, 15),
 * 16 staged chunk stored one k-step later, 64 staging loads ahead of the k-step's MFMAs. */
int df_debug_wino_pack_weights_bf16x3(const float* w, float* wp, int64_t cin, int64_t cout, int mode, void* stream);
int df_debug_wino_conv_fwd_bf16x3(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src, float* y,
                                  int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, void* stream);

/* Round-4 probe (DESIGN.md 4.1b "x-blocked staging"): df_wino_conv_fwd(BIAS | LRELU) with the input staged by LDS-DMA from an x-blocked copy
 * `xg` of x (df_debug_wino_xblk_elems floats; written by the call unless variant & 256).  variant & 15: 1 all waves issue the DMA pieces
 * behind k-step 0's MFMAs, 3 only waves 4-7 (one per SIMD), 5 / 7 the same in front of the MFMAs; 19: variant 3 without any staging (timing). */
int64_t df_debug_wino_xblk_elems(int64_t B, int64_t D, int64_t H, int64_t W, int64_t C);
int df_debug_wino_conv_fwd_xblk(const float* x, float* xg, const float* wp, const float* bias, float* y, int64_t B, int64_t D, int64_t H, int64_t W,
                                int64_t Cin, int64_t Cout, float leak, int variant, void* stream);

#ifdef __cplusplus
}
#endif
#endif
