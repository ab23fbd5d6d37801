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

#ifdef __cplusplus
}
#endif
#endif
