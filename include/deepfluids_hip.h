/*
 * deepfluids_hip.h -- C-ABI of libdeepfluids_hip.so (gfx950 / MI355X only).
 *
 * The drop-in boundary of the Deep Fluids velocity-field train step.  The reference
 * (byungsook/deep-fluids, TF 1.15) has no FFI of its own: its boundary is the Python call
 * surface of ops.py / model.py.  deep_fluids_amd/ops.py + model.py reproduce that surface and
 * call THIS library through ctypes; each entry point below cites the reference lines whose
 * arithmetic it replaces (paths relative to the upstream repo root).
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer to fp32, channels-last, C-contiguous data unless noted;
 *     2-D tensors are [B,Y,X,C], 3-D tensors [B,Z,Y,X,C] ("x: bzyxd", ops.py:228);
 *   - the caller owns and frees every buffer including workspaces (sizes via *_workspace_bytes);
 *     the library keeps no tensor state and allocates nothing;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*), nothing
 *     synchronises internally, so calls compose with PyTorch's current stream and hipGraph capture;
 *   - return 0 on success, <0 = DF_E* argument error, >0 = hipError_t; never throws, never
 *     aborts; a message for the calling thread's last failure is kept in df_last_error();
 *   - re-entrant; no global state besides read-only kernel handles: algorithm choices are call arguments
 *     (`algo`, DF_CONV_VALU_ONLY), never process-wide switches.  (Tuning builds, `make tuning` with -DDF_TUNING, add
 *     the instrumented kernels and knobs declared in deepfluids_hip_debug.h to a SEPARATE library; the release
 *     library exports exactly what this header declares.)
 */
#ifndef DEEPFLUIDS_HIP_H
#define DEEPFLUIDS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DF_VERSION 207 /* 0.2.7: + df_wino2d43_conv_bits, df_wino2d43_signbits_bytes, df_wino2d43_conv_addup_bits, df_lrelu_words2d_bwd_pool2x; 0.2.6: + df_wino43_*, df_wino2d43_*, df_conv_s2_dgrad_form; 0.2.5: + df_adam_tf1_step_dev, df_gd_step(_dev), df_store_scalars (hipGraph replay of the train step); 0.2.4: + df_conv_s2_dgrad; 0.2.3: + df_conv_s2_wgrad; 0.2.2: + df_conv_wgrad_form, df_upconv_wgrad_form; 0.2.1: + df_lrelu_bwd_pool2x, df_wino_conv_fwd_addup_bits, df_lrelu_bits_bwd_pool2x (0.2.0: df_conv_wgrad_algo, df_upconv_wgrad_algo, DF_CONV_VALU_ONLY, df_velocity_loss2d/3d) */

enum {
  DF_OK = 0,
  DF_EINVAL = -1,    /* null pointer / non-positive extent */
  DF_ESHAPE = -2,    /* extent unsupported by this kernel (e.g. forward difference needs n >= 2) */
  DF_EALIGN = -3,    /* pointer not 16-byte aligned */
  DF_EWORKSPACE = -4 /* workspace too small */
};

/* flags for df_conv_fwd */
enum {
  DF_CONV_LRELU = 1,      /* y = max(v, leak*v)            ops.py:9-10 fused into the conv epilogue   */
  DF_CONV_RESIDUAL = 2,   /* y += residual (after act)      model.py:35,40,77,82                       */
  DF_CONV_MASK = 4,       /* y *= (mask_src > 0 ? 1 : leak) lrelu backward fused into the dgrad epilogue */
  DF_CONV_BIAS = 8,       /* v += bias[cout]                slim.conv* biases                          */
  DF_CONV_ADDUP = 16,     /* second output y2 = y + nearest_up2x(xc)  (df_wino_conv_fwd_addup only)     */
  DF_CONV_VALU_ONLY = 32  /* thin layers (Cin or Cout <= 4): take the general-shape vector-ALU kernel even where the
                             matrix-core form exists (the parity tests compare the two)                    */
};

typedef void* df_stream_t; /* hipStream_t */

int df_version(void);
const char* df_last_error(void);

/* ---- forward-difference stencils (HBM-bound) ---------------------------------------------- */

/* curl(x) ops.py:264-274.  psi [B,Y,X,1] -> u [B,Y,X,2] = (D_y psi, -D_x psi). */
int df_curl2d_fwd(const float* psi, float* u, int64_t B, int64_t Y, int64_t X, df_stream_t stream);
/* adjoint of df_curl2d_fwd: gu [B,Y,X,2] -> gpsi [B,Y,X,1]. */
int df_curl2d_bwd(const float* gu, float* gpsi, int64_t B, int64_t Y, int64_t X, df_stream_t stream);

/* jacobian(x) ops.py:205-225.  x [B,Y,X,2] -> j [B,Y,X,4] = (dudx,dudy,dvdx,dvdy), w [B,Y,X,1] = dvdx-dudy.
 * j or w may be NULL (that output is not produced). */
int df_jacobian2d_fwd(const float* x, float* j, float* w, int64_t B, int64_t Y, int64_t X, df_stream_t stream);
/* adjoint: gj [..,4] and/or gw [..,1] (either may be NULL, not both) -> gx [..,2]. */
int df_jacobian2d_bwd(const float* gj, const float* gw, float* gx, int64_t B, int64_t Y, int64_t X,
                      df_stream_t stream);

/* jacobian3(x) ops.py:227-262.  x [B,Z,Y,X,3] -> j [..,9] = (dudx,dudy,dudz,dvdx,dvdy,dvdz,dwdx,dwdy,dwdz),
 * c [..,3] = (dwdy-dvdz, dudz-dwdx, dvdx-dudy).  j or c may be NULL.
 * `curl3(x)` of the north star == df_jacobian3d_fwd(x, NULL, c, ...)  (trainer3.py:18). */
int df_jacobian3d_fwd(const float* x, float* j, float* c, int64_t B, int64_t Z, int64_t Y, int64_t X,
                      df_stream_t stream);
/* adjoint: gj [..,9] and/or gc [..,3] (either may be NULL, not both) -> gx [..,3]. */
int df_jacobian3d_bwd(const float* gj, const float* gc, float* gx, int64_t B, int64_t Z, int64_t Y, int64_t X,
                      df_stream_t stream);

/* divergence ops.py:276-284: x [B,Y,X,2] -> [B,Y-1,X-1,1];  divergence3 ops.py:286-290: x [B,Z,Y,X,3] -> [B,Z-1,Y-1,X-1,1]. */
int df_divergence2d(const float* x, float* d, int64_t B, int64_t Y, int64_t X, df_stream_t stream);
int df_divergence3d(const float* x, float* d, int64_t B, int64_t Z, int64_t Y, int64_t X, df_stream_t stream);

/* ---- fused tail of the velocity-field train step (trainer.py:140-146,170-172; trainer3.py:18-24,49-51) ----------------------
 * u = curl(psi) | jacobian3(psi)[1];  l1 = mean|u - x|;  jl1 = mean|J(u) - J(x)|  with J = jacobian(.)[0] | jacobian3(.)[0] and
 * the ground-truth Jacobian of trainer.py:29-32 recomputed on the fly: one kernel over psi and x (36 B/voxel in 3-D instead of the
 * 240 B/voxel of the five reference ops; the 9-channel Jacobians are never materialised).
 *   psi [B,Z,Y,X,3] | [B,Y,X,1], x [B,Z,Y,X,3] | [B,Y,X,2];  u (nullable) receives the velocity field, bit-identical to
 *   df_jacobian3d_fwd(psi, NULL, u) | df_curl2d_fwd;  l1 / jl1: device scalars.
 * Backward: gpsi = curl^T( g_l1/N1 sign(u - x) + J^T( g_jl1/NJ sign(J(u) - J(x)) ) ) from the saved u and x;  g_l1 / g_jl1 are
 * device scalars (NULL = 1).  One workspace serves both directions (df_velocity_loss*_workspace_bytes, 16-byte aligned). */
int64_t df_velocity_loss3d_workspace_bytes(int64_t B, int64_t Z, int64_t Y, int64_t X);
int df_velocity_loss3d_fwd(const float* psi, const float* x, float* u, float* l1, float* jl1, int64_t B, int64_t Z, int64_t Y,
                           int64_t X, void* workspace, int64_t workspace_bytes, df_stream_t stream);
int df_velocity_loss3d_bwd(const float* u, const float* x, const float* g_l1, const float* g_jl1, float* gpsi, int64_t B, int64_t Z,
                           int64_t Y, int64_t X, void* workspace, int64_t workspace_bytes, df_stream_t stream);
int64_t df_velocity_loss2d_workspace_bytes(int64_t B, int64_t Y, int64_t X);
int df_velocity_loss2d_fwd(const float* psi, const float* x, float* u, float* l1, float* jl1, int64_t B, int64_t Y, int64_t X,
                           void* workspace, int64_t workspace_bytes, df_stream_t stream);
int df_velocity_loss2d_bwd(const float* u, const float* x, const float* g_l1, const float* g_jl1, float* gpsi, int64_t B, int64_t Y,
                           int64_t X, void* workspace, int64_t workspace_bytes, df_stream_t stream);


/* ---- losses (trainer.py:170-172, trainer3.py:49-51) -------------------------------------- */

/* workspace for df_l1_mean_fwd (bytes). */
int64_t df_l1_mean_workspace_bytes(int64_t n);
/* out[0] = mean(|a-b|) over n elements; deterministic two-stage reduction (fp64 partials). */
int df_l1_mean_fwd(const float* a, const float* b, int64_t n, float* out, void* workspace, int64_t workspace_bytes,
                   df_stream_t stream);
/* ga[i] = sign(a[i]-b[i]) * scale * (gout ? gout[0] : 1) / n   (tf Abs grad: sign, 0 at 0). */
int df_l1_mean_bwd(const float* a, const float* b, const float* gout, float scale, float* ga, int64_t n,
                   df_stream_t stream);

/* ---- element-wise / small layers ---------------------------------------------------------- */

/* lrelu(x, leak) ops.py:9-10 and its backward (slope 1 where y > 0 else leak; y = saved OUTPUT). */
int df_lrelu_fwd(const float* x, float* y, float leak, int64_t n, df_stream_t stream);
int df_lrelu_bwd(const float* gy, const float* y, float* gx, float leak, int64_t n, df_stream_t stream);

/* y = a + b (residual add model.py:35,40,77,82), n elements. */
int df_add(const float* a, const float* b, float* y, int64_t n, df_stream_t stream);

/* nearest-neighbour 2x up-sampling of every spatial axis, channels-last (src = dst>>1):
 * upscale ops.py:75-77 (D=1), upscale3 ops.py:79-91.  x [B,D,H,W,C] -> y [B,2D|1,2H,2W,C]; C % 4 == 0. */
int df_upsample2x_fwd(const float* x, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                      df_stream_t stream);
/* The backward tail of an up-sampling generator block (`x = lrelu(conv4(.)) + upscale(xc)`, model.py:36-40 / 78-82) in ONE pass
 * over the incoming gradient: gx = gy * (y > 0 ? 1 : leak)  (== df_lrelu_bwd; y = conv4's saved output, fine resolution
 * [B,2D|1,2H,2W,C]) and gpool = 2x2(x2) sum-pool of gy  (== df_upsample2x_bwd, the skip gradient w.r.t. xc, [B,D,H,W,C]).
 * B, D, H, W are the COARSE extents; C % 4 == 0.  Bit-identical to the two separate calls. */
int df_lrelu_bwd_pool2x(const float* gy, const float* y, float* gx, float* gpool, float leak, int64_t B, int64_t D, int64_t H,
                        int64_t W, int64_t C, int is_3d, df_stream_t stream);
/* adjoint (2x2(x2) sum-pool): gy [B,2D|1,2H,2W,C] -> gx [B,D,H,W,C]. */
int df_upsample2x_bwd(const float* gy, float* gx, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                      df_stream_t stream);

/* ---- wrapper generality (conv_general.hip): the shapes of the reference's call surface that its trainers never use ------------------
 * ops.py:12-16 `conv2d / conv3d(x, o_dim, k=4, s=2, act)` = slim.conv2d / conv3d, padding 'SAME', ANY cubic kernel 1 <= k <= 7 and stride
 * 1 <= s <= 4, any extents (odd ones included): out = ceil(n / s), pad_total = max((out - 1) s + k - n, 0), pad_before = pad_total / 2.
 * 2-D: D = 1, kz = 1; 3-D: kz = k.  Weights in the TF layout [kz, k, k, Cin, Cout] (no packing).  Plain vector-ALU kernels, fixed summation
 * order: correct on every shape, not fast -- k = 3, s in {1, 2} on even extents take the matrix-core kernels above. */
int df_conv_general_out_dims(int64_t D, int64_t H, int64_t W, int kz, int k, int s, int64_t* Do, int64_t* Ho, int64_t* Wo);
/* y [B,Do,Ho,Wo,Cout] = conv(x [B,D,H,W,Cin], w) (+ bias) (+ lrelu);  flags: DF_CONV_BIAS | DF_CONV_LRELU. */
int df_conv_general_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin,
                        int64_t Cout, int kz, int k, int s, int flags, float leak, df_stream_t stream);
/* gx [B,D,H,W,Cin] from gy [B,Do,Ho,Wo,Cout] (B, D, H, W = the INPUT extents of the forward conv). */
int df_conv_general_dgrad(const float* gy, const float* w, float* gx, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz,
                          int k, int s, df_stream_t stream);
/* gw [kz,k,k,Cin,Cout] and gb [Cout] (gb may be NULL).  Expected cost: one workgroup per (tap, input channel) walks ALL output voxels in four
 * sub-ranges (gb: 16 sub-ranges per 64 channels) -- a correctness path, SECONDS per call on 10^7-voxel grids; training at such sizes with a
 * kernel / stride the matrix-core kernels do not take needs a dedicated kernel first. */
int df_conv_general_wgrad(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin,
                          int64_t Cout, int kz, int k, int s, df_stream_t stream);
/* ops.py:66-73 `resize_nearest_neighbor(x, new_size)` = tf.image.resize_nearest_neighbor, align_corners=False, ANY target size:
 * y[.., o, ..] = x[.., min(floor(o * in / out), in - 1), ..] per spatial axis (D = Do = 1 for 2-D); bwd = its adjoint in gather form. */
int df_resize_nn_fwd(const float* x, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int64_t Do, int64_t Ho, int64_t Wo,
                     df_stream_t stream);
int df_resize_nn_bwd(const float* gy, float* gx, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int64_t Do, int64_t Ho, int64_t Wo,
                     df_stream_t stream);

/* linear ops.py:23-24 (slim.fully_connected): y[B,N] = x[B,K] . w[K,N] + bias[N]  (bias may be NULL). */
int64_t df_linear_workspace_bytes(int64_t B, int64_t K, int64_t N); /* > 0 only for the large-K/small-N split-K path */
int df_linear_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t K, int64_t N,
                  void* workspace, int64_t workspace_bytes, df_stream_t stream);
/* backward: gw[K,N] = x^T gy, gb[N] = sum_b gy, gx[B,K] = gy w^T (gx / gw / gb may be NULL). */
int df_linear_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int64_t B,
                  int64_t K, int64_t N, df_stream_t stream);

/* tf.concat([a, b], axis=-1) of two channels-last tensors with `rows` voxels (encoder skips, model.py:138,174) and the
 * reverse split of the gradient (16-byte vectorised when both channel counts are multiples of 4). */
int df_concat2_fwd(const float* a, const float* b, float* y, int64_t rows, int64_t Ca, int64_t Cb, df_stream_t stream);
int df_concat2_bwd(const float* gy, float* ga, float* gb, int64_t rows, int64_t Ca, int64_t Cb, df_stream_t stream);

/* zero insertion used by the stride-2 conv backward: out[B,2D|1,2H,2W,C][2o+1] = g[B,D,H,W,C][o], zeros elsewhere. */
int df_dilate2_odd(const float* g, float* out, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                   df_stream_t stream);

/* tf.sigmoid (AE latent code with use_sparse, model.py:196,210) and its backward from the saved output. */
/* Bernoulli-KL sparsity of the auto-encoder code (trainer3.py:272-277): out = sum_{j<n} KL(Bern(rho) || Bern(mean_b z[b][j]));
 * z is [B, ncol] (sigmoid outputs), the first n columns take part.  bwd: gz[B, ncol] = gout * scale * dloss/dz (0 for j >= n). */
int df_kl_bernoulli_fwd(const float* z, int64_t B, int64_t ncol, int64_t n, float rho, float* out, df_stream_t stream);
int df_kl_bernoulli_bwd(const float* z, const float* gout, float scale, float* gz, int64_t B, int64_t ncol, int64_t n, float rho,
                        df_stream_t stream);
int df_sigmoid_fwd(const float* x, float* y, int64_t n, df_stream_t stream);
int df_sigmoid_bwd(const float* gy, const float* y, float* gx, int64_t n, df_stream_t stream);

/* mean((a-b)^2) (loss_p, trainer.py:395 / trainer3.py:268-270); workspace as df_l1_mean_fwd. */
int df_mse_mean_fwd(const float* a, const float* b, int64_t n, float* out, void* workspace, int64_t workspace_bytes,
                    df_stream_t stream);
int df_mse_mean_bwd(const float* a, const float* b, const float* gout, float scale, float* ga, int64_t n,
                    df_stream_t stream);

/* gb[c] = sum over rows of g[rows, C]  (conv bias gradient), deterministic. */
int64_t df_colsum_workspace_bytes(int64_t rows, int64_t C);
int df_colsum(const float* g, float* gb, int64_t rows, int64_t C, void* workspace, int64_t workspace_bytes,
              df_stream_t stream);

/* tf.train.AdamOptimizer update (trainer.py:160-162,184; "epsilon-hat" form), applied to one
 * flat parameter slab:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr_t m / (sqrt(v) + eps),
 * with lr_t = lr sqrt(1-b2^t)/(1-b1^t) computed by the caller on the host. */
int df_adam_tf1_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                     float eps, float grad_scale, df_stream_t stream);

/* The same update with the two per-step scalars read from DEVICE memory -- scalars[0] = lr_t, scalars[1] = grad_scale -- so that a
 * captured hipGraph of the whole train step (one `sess.run(g_optim)`, trainer.py:265-269) replays with the values the host stored
 * before the launch; arithmetic identical to df_adam_tf1_step (bitwise-equal parameters). */
int df_adam_tf1_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* scalars, float beta1, float beta2,
                         float eps, df_stream_t stream);

/* tf.train.GradientDescentOptimizer (the `gd` option, trainer.py:163-165):  p -= (lr * grad_scale) * g  over one flat slab;
 * _dev: scalars[0] = lr, scalars[1] = grad_scale from device memory (graph replay, as above). */
int df_gd_step(float* p, const float* g, int64_t n, float lr, float grad_scale, df_stream_t stream);
int df_gd_step_dev(float* p, const float* g, int64_t n, const float* scalars, df_stream_t stream);

/* dst[0..n-1] = v0..v(n-1), n <= 4, by a one-thread kernel whose arguments travel by value: the host-side update of the device
 * scalars above, ordered on `stream` before the graph launch, with no host buffer that a later step could overwrite early. */
int df_store_scalars(float* dst, int64_t n, float v0, float v1, float v2, float v3, df_stream_t stream);

/* ---- convolutions on MFMA (exact fp32: v_mfma_f32_32x32x2_f32) ----------------------------- */

/* slim.conv2d / conv3d, k=3, stride 1, padding SAME, channels-last (ops.py:12-16; model.py:26,42,68,84).
 * Weights are given in TF layout [kz,ky,kx,Cin,Cout] (2-D: kz = 1) and are re-packed once per update into
 * the MFMA operand order by df_conv_pack_weights:
 *   mode 0: forward operand;  mode 1: dgrad operand (taps mirrored, Cin/Cout swapped). */
int64_t df_conv_packed_elems(int64_t taps, int64_t cin, int64_t cout, int mode);
int df_conv_pack_weights(const float* w, float* wp, int64_t taps, int64_t cin, int64_t cout, int mode,
                         df_stream_t stream);
/* y[B,D,H,W,Cout] = epilogue( conv_same(x[B,D,H,W,Cin], w) ).  D = 1 and kz = 1 for 2-D.
 * `wp` packed with mode 0 (forward) or mode 1 (then x is dL/dy and y is dL/dx: the dgrad).
 * flags: DF_CONV_*; bias / residual / mask_src are used only when their flag is set. */
int df_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int flags,
                float leak, df_stream_t stream);
/* k=3, stride 2, TF 'SAME' on even input extents (pad 0 before / 1 after): the encoder's down-sampling convs
 * (model.py:141-143, 177-179).  x [B,2Do|1,2Ho,2Wo,Cin] -> y [B,Do,Ho,Wo,Cout]; `wp` packed with mode 0;
 * flags: DF_CONV_BIAS | DF_CONV_LRELU only. */
int df_conv_s2_fwd(const float* x, const float* wp, const float* bias, float* y, int64_t B, int64_t Do, int64_t Ho,
                   int64_t Wo, int64_t Cin, int64_t Cout, int kz, int flags, float leak, df_stream_t stream);
/* Weight (and bias) gradient of that stride-2 conv: gw[tz][ty][tx][ci][co] = sum_{b,o} x[b][2o + t][ci] gy[b][o][co] (TF autodiff of
 * slim.conv3d(stride=2); model.py:141-143, 177-179) computed natively on the OUTPUT grid -- no zero-inserted gradient tensor.
 * x [B,2Do|1,2Ho,2Wo,Cin], gy [B,Do,Ho,Wo,Cout]; gw TF layout [kz,3,3,Cin,Cout]; gb may be NULL.  Instantiated for Wo in {8,16,32,64},
 * even Cin, Cout >= 32 (df_conv_s2_wgrad_workspace_bytes returns 0 otherwise: use df_dilate2_odd + df_conv_wgrad). */
int64_t df_conv_s2_wgrad_workspace_bytes(int64_t B, int64_t Do, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, int kz);
int df_conv_s2_wgrad(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t Do, int64_t Ho, int64_t Wo, int64_t Cin,
                     int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream);
/* Input gradient of that stride-2 conv (TF autodiff of slim.conv2d/conv3d(stride=2): model.py:141-143, 177-179; the discriminator's
 * model.py:94-99): gx [B,2Do|1,2Ho,2Wo,Cin] from gy [B,Do,Ho,Wo,Cout].  `wp` = df_upconv_pack_weights(w, Cin, Cout, kz, mode 2).  Per axis
 * dx[2m] = g[m-1] w[2] + g[m] w[0], dx[2m+1] = g[m] w[1]: 8 (4) parity classes with 8,4,4,2,4,2,2,1 (4,2,2,1) LIVE taps; each class runs a
 * kernel specialised on its tap counts (27 tap-products per coarse voxel and channel pair; the generic 2x2x2-tap parity-class kernel --
 * df_upconv_fwd on the same operand, used here for channel counts without a specialisation -- multiplies 64, a stride-1 dgrad on the
 * zero-inserted gradient 216).  Every output voxel is written exactly once (no accumulation). */
int df_conv_s2_dgrad(const float* gy, const float* wp, float* gx, int64_t B, int64_t Do, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout,
                     int kz, df_stream_t stream);
/* Which kernel df_conv_s2_dgrad takes for this gradient pointer (alignment) and channel counts (Cin / Cout of the FORWARD conv):
 * 1 = each parity class on its live taps only, 0 = the generic 2x2(x2)-tap parity-class kernel (zero-padded taps).  Host-only. */
int df_conv_s2_dgrad_form(const float* gy, int64_t Cin, int64_t Cout);
/* ---- up-sampling-aware first conv of a generator block --------------------------------------------------------------
 * model.py:36-37 / 78-79 feed `upscale(x, 2)` into the next block's first conv.  conv(nearest_up2x(xc), w) is computed
 * WITHOUT materialising the up-sampled tensor as 8 (3-D) / 4 (2-D) parity-class convs with 2x2x2 / 2x2 pre-summed taps
 * on the coarse grid (27/8 = 3.4x fewer FLOPs; identical up to fp32 summation order).
 *   xc [B,Dc|1,Hc,Wc,Cin] (coarse)  ->  y [B,2Dc|1,2Hc,2Wc,Cout] (fine).  Weights: TF layout [kz,3,3,Cin,Cout].
 * Pack modes: 0 forward operand; 1 operand of df_upconv_dgrad; 2 DGRAD OPERAND OF THE STRIDE-2 CONV (df_conv_s2_fwd): the adjoint of
 *   y[o] = sum_t x[2o+t] w[t] is dx[2m] = g[m-1] w[2] + g[m] w[0], dx[2m+1] = g[m] w[1] per axis, i.e. the same parity-class / offset
 *   structure; run it as  df_upconv_fwd(g [B,Do,Ho,Wo,Cout], wp(mode 2), NULL, dx [B,2Do,2Ho,2Wo,Cin], B, Do, Ho, Wo, Cin := Cout,
 *   Cout := Cin, kz, 0, 0, stream)  -- 64 tap-products per coarse voxel instead of the 216 of a stride-1 dgrad on the
 *   zero-inserted gradient (model.py:141-143, 177-179: the encoder's / discriminator's down-sampling layers). */
int64_t df_upconv_packed_elems(int64_t cin, int64_t cout, int kz, int mode);
int df_upconv_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int kz, int mode, df_stream_t stream);
int df_upconv_fwd(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Dc, int64_t Hc,
                  int64_t Wc, int64_t Cin, int64_t Cout, int kz, int flags, float leak, df_stream_t stream);
/* acc[B,Dc,Hc,Wc,Cin] += d(loss)/d(xc) given g = d(loss)/d(y) on the fine grid; `wp` packed with mode 1. */
int df_upconv_dgrad(const float* g, const float* wp, float* acc, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                    int64_t Cin, int64_t Cout, int kz, df_stream_t stream);
/* gw[kz,3,3,Cin,Cout] (the ORIGINAL 3-wide filter) and gb from the coarse input and the fine gradient. */
int64_t df_upconv_wgrad_workspace_bytes(int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, int kz);
int df_upconv_wgrad(const float* xc, const float* gy, float* gw, float* gb, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                    int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream);
/* algo: 0 best available | 1 generic direct kernel on the parity classes | 2, 3 three-product parity-class kernel |
 * 4 the 27-point Winograd-(x,y,z) form wherever instantiated (df_upconv_wgrad == algo 0). */
int df_upconv_wgrad_algo(const float* xc, const float* gy, float* gw, float* gb, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                         int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, int algo,
                         df_stream_t stream);
/* y[fine] = a[fine] + nearest_up2x(bc[coarse])  (block-end residual `x += x0` with x0 = upscale(.), model.py:35-40). */
int df_add_up2x(const float* a, const float* bc, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                df_stream_t stream);

/* ---- Winograd F(2x2x2, 3x3x3) form of the 3-D stride-1 convolution (conv_wino.hip) ------------------------------------
 * Same result as df_conv_fwd with kz = 3 up to fp32 rounding order (all arithmetic fp32, 3.4x fewer matrix-core FLOPs);
 * same epilogue flags.  Needs Cin % 32 == 0, Cout % 32 == 0 and D*H*W*max(Cin,Cout) <= 2^29 (one batch volume below 2 GiB).  Weights: TF layout [3,3,3,Cin,Cout], transformed and
 * packed once per update by df_wino_pack_weights (mode 0: forward operand, mode 1: dgrad operand). */
int64_t df_wino_packed_elems(int64_t cin, int64_t cout, int mode);
int df_wino_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream);
int df_wino_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                     float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak,
                     df_stream_t stream);

/* Up-sampling-aware variant: y[B,2Dc,2Hc,2Wc,Cout] = lrelu(conv_same(nearest_up2x(xc), w) + bias) without materialising the up-sampled
 * tensor and with only 27 of the 64 Winograd products (the others are identically zero for a duplicated input).  `wp` is the mode-0
 * pack of df_wino_pack_weights; flags must be DF_CONV_BIAS | DF_CONV_LRELU. */
int df_wino_upconv_fwd(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Dc, int64_t Hc,
                       int64_t Wc, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream);

/* Its adjoint w.r.t. xc: acc[B,Dc,Hc,Wc,Cin] += sum-pool2x(conv_same^T(g, w)) for g = d(loss)/d(y) on the fine grid
 * [B,2Dc,2Hc,2Wc,Cout]; the pooled inverse transform needs the same 27 points.  `wp` is the mode-1 pack of
 * df_wino_pack_weights(w, wp, Cin, Cout, 1).  Drop-in for df_upconv_dgrad (kz = 3, channels multiples of 32). */
int df_wino_upconv_dgrad(const float* g, const float* wp, float* acc, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin,
                         int64_t Cout, df_stream_t stream);

/* Last conv of an up-sampling generator block with the block-end skip add fused in (model.py:35,40,77,82 `x += x0`, x0 = upscale(xc)):
 *   y = lrelu(conv_same(x, w) + bias)  (kept for the backward pass),   y2 = y + nearest_up2x(xc),  xc [B,D/2,H/2,W/2,Cout].
 * Same packed weights and shape limits as df_wino_conv_fwd; D, H, W even. */
int df_wino_conv_fwd_addup(const float* x, const float* wp, const float* bias, const float* xc, float* y, float* y2, int64_t B,
                           int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, float leak, df_stream_t stream);

/* ---- round 6: F(2,3) x F(2,3) x F(4,3) Winograd family (conv_wino43.hip) -------------------------------------------------------------
 * The same convolution as df_wino_conv_fwd* (slim.conv3d k=3 s=1 SAME, model.py:66-70; dgrad = mode-1 operand) with a 2 x 2 x 4 output
 * tile: 6 matrix multiply-adds per output voxel and (cin, cout) pair instead of 8, about one bit of fp32 accuracy less.
 * ONE entry point covers every fused epilogue of the F(2,3)^3 family; the sign words it writes / reads have the SAME layout
 * (df_wino_signbits_bytes, df_lrelu_bits_bwd_pool2x), so the two families can be mixed layer by layer:
 *   flags        DF_CONV_BIAS | LRELU | RESIDUAL (fine tensor `residual`) | MASK | ADDUP (`residual` = the COARSE tensor, y2 = y + up2x)
 *   mask_src     fp32 activation whose sign is the lrelu mask of DF_CONV_MASK, or
 *   mask_bits    the same mask as sign words (exactly one of the two with DF_CONV_MASK)
 *   sign_bits    non-null: also emit the sign words of y;  y == NULL (only with ADDUP + sign_bits): y itself is not written. */
int64_t df_wino43_packed_elems(int64_t cin, int64_t cout, int mode);
int df_wino43_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream);
int df_wino43_conv(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src, const void* mask_bits,
                   float* y, float* y2, void* sign_bits, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags,
                   float leak, df_stream_t stream);

/* The 2-D twin (conv_wino2d43.hip): slim.conv2d k=3 s=1 SAME (ops.py:12-13; model.py:24-28) as Winograd F(2,3) x F(4,3) -- 3 matrix multiply-adds
 * per output pixel and (cin, cout) pair instead of the 4 of df_wino2d_conv_fwd; same arguments, flags (BIAS | LRELU | RESIDUAL | MASK with an
 * fp32 mask_src) and error convention as df_wino2d_conv_fwd; mode 1 of the pack = the dgrad operand. */
int64_t df_wino2d43_packed_elems(int64_t cin, int64_t cout, int mode);
int df_wino2d43_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream);
int df_wino2d43_conv(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src, float* y, int64_t B,
                     int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream);
/* ... with sign words (the 2-D twin of df_wino_conv_fwd_bits below; reference: the lrelu of slim.conv2d's activation_fn, ops.py:12-13, whose slope TF's
 * autodiff multiplies into the gradient -- model.py:24-28): flags == BIAS | LRELU + sign_bits: the forward conv also writes (activation > 0) as ONE
 * 32-bit word per (tile block of 16 x 32 pixels, 32-cout slice, thread) = its 32 outputs; flags == MASK + mask_bits: the dgrad of the SAME geometry
 * (B, H, W, channel count of the masked tensor) reads those words instead of the fp32 activation (1/32 of the bytes).  df_wino2d43_signbits_bytes
 * = the buffer size (16-byte aligned buffers).  Results bit-identical to the fp32-mask path. */
int64_t df_wino2d43_signbits_bytes(int64_t B, int64_t H, int64_t W, int64_t C);
int df_wino2d43_conv_bits(const float* x, const float* wp, const float* bias, const void* mask_bits, float* y, void* sign_bits, int64_t B, int64_t H,
                          int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream);
/* The block tail of a 2-D up-sampling generator block on sign words (the twins of df_wino_conv_fwd_addup_bits / df_lrelu_bits_bwd_pool2x; reference:
 * model.py:36-40 -- x = conv(...); x += x0 with x0 = upscale(previous block)).  df_wino2d43_conv_addup_bits: y2 = lrelu(conv(x) + bias) + nearest_up2x(xc),
 * xc = the COARSE tensor [B, H/2, W/2, Cout]; the conv's own activation is NOT written, only its sign words (all the backward pass needs of it).
 * df_lrelu_words2d_bwd_pool2x: gx = gy * lrelu'(activation) from those words and gpool[B, Hc, Wc, C] = the 2 x 2 sum-pool of gy (the skip path's
 * gradient), one pass over gy [B, 2 Hc, 2 Wc, C].  Bit-identical to df_wino2d43_conv + df_add_up2x / df_lrelu_bwd_pool2x. */
int df_wino2d43_conv_addup_bits(const float* x, const float* wp, const float* bias, const float* xc, float* y2, void* sign_bits, int64_t B, int64_t H,
                                int64_t W, int64_t Cin, int64_t Cout, float leak, df_stream_t stream);
int df_lrelu_words2d_bwd_pool2x(const float* gy, const void* mask_bits, float* gx, float* gpool, float leak, int64_t B, int64_t Hc, int64_t Wc, int64_t C,
                                df_stream_t stream);

/* Sign-bit masks.  A masked dgrad (DF_CONV_MASK) multiplies its output by the lrelu slope of the layer below, i.e. it needs ONE BIT
 * per element of that layer's activation; read from the fp32 activation that is 3.2 GB per top-level launch at cfg3.  The forward
 * conv that produced the activation can emit the bits instead (one byte per lane: the signs of its 8 outputs, 1/32 of the bytes) and
 * the dgrad of the SAME geometry (B, D, H, W and channel count) reads those:
 *   df_wino_conv_fwd_bits(flags = DF_CONV_BIAS | DF_CONV_LRELU, bias, mask_bits = NULL, sign_bits = out)   forward, writes y and bits(y > 0)
 *   df_wino_upconv_fwd_bits(...)                                                                            the same for the 27-point form
 *   df_wino_conv_fwd_bits(flags = DF_CONV_MASK, bias = NULL, mask_bits = in, sign_bits = NULL)             masked dgrad (mode-1 weights)
 * The byte layout is private to the two kernels (tile block, cout slice, wave, cout block, lane); df_wino_signbits_bytes sizes it. */
int64_t df_wino_signbits_bytes(int64_t B, int64_t D, int64_t H, int64_t W, int64_t C);
int df_wino_conv_fwd_bits(const float* x, const float* wp, const float* bias, const void* mask_bits, float* y, void* sign_bits, int64_t B,
                          int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream);
int df_wino_upconv_fwd_bits(const float* xc, const float* wp, const float* bias, float* y, void* sign_bits, int64_t B, int64_t Dc,
                            int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, float leak, df_stream_t stream);
/* The block tail on sign bits: df_wino_conv_fwd_addup without its first output -- y2 = lrelu(conv_same(x, w) + bias) + nearest_up2x(xc)
 * and the sign bits of the lrelu output (sized by df_wino_signbits_bytes(B, D, H, W, Cout)); the pre-add activation itself (3.2 GB per
 * top-level layer at cfg3) is never written.  df_lrelu_bits_bwd_pool2x is the matching backward tail (== df_lrelu_bwd_pool2x with the
 * mask read from those bits: gx = gy * (bit ? 1 : leak), gpool = 2x2x2 sum-pool of gy); B, Dc, Hc, Wc the COARSE extents
 * (gy / gx [B,2Dc,2Hc,2Wc,C], gpool [B,Dc,Hc,Wc,C]), C % 32 == 0. */
int df_wino_conv_fwd_addup_bits(const float* x, const float* wp, const float* bias, const float* xc, float* y2, void* sign_bits,
                                int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, float leak, df_stream_t stream);
int df_lrelu_bits_bwd_pool2x(const float* gy, const void* mask_bits, float* gx, float* gpool, float leak, int64_t B, int64_t Dc, int64_t Hc,
                             int64_t Wc, int64_t C, df_stream_t stream);

/* 2-D twin: Winograd F(2x2, 3x3) form of the stride-1 3x3 convolution (conv_wino2d.hip): 2.25x fewer matrix-core FLOPs than df_conv_fwd
 * with kz = 1, fp32 throughout, same epilogue flags.  Needs Cin % 32 == 0, Cout % 32 == 0, H*W*max(Cin,Cout) <= 2^29. */

int64_t df_wino2d_packed_elems(int64_t cin, int64_t cout, int mode);
int df_wino2d_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream);
int df_wino2d_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                       float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak,
                       df_stream_t stream);
/* 2-D up-sampling-aware forms (9 of the 16 Winograd products): forward y[B,2Hc,2Wc,Cout] = lrelu(conv_same(nearest_up2x(xc), w) + bias)
 * (mode-0 pack of df_wino2d_pack_weights; flags = DF_CONV_BIAS | DF_CONV_LRELU) and its adjoint acc[B,Hc,Wc,Cin] += sum-pool2x(conv_same^T(g, w))
 * (mode-1 pack).  Drop-ins for df_upconv_fwd / df_upconv_dgrad with kz = 1, channels multiples of 32. */
int df_wino2d_upconv_fwd(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Hc, int64_t Wc, int64_t Cin,
                         int64_t Cout, int flags, float leak, df_stream_t stream);
int df_wino2d_upconv_dgrad(const float* g, const float* wp, float* acc, int64_t B, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout,
                           df_stream_t stream);


/* ---- opt-in "bf16x3" precision mode (conv_bf16.hip) ---------------------------------------------------------------------
 * fp32 operands are split a = hi + lo into two bf16 words and a*b ~= hi*hi + hi*lo + lo*hi runs on the bf16 matrix pipe
 * (3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate): 16 significand bits per operand, ~5x the fp32 MFMA rate.  Same arguments
 * as the fp32 twins; weights must be packed by the matching *_bf16x3 pack call.  Needs Cin % 4 == 0, channels >= 16. */
int64_t df_conv_packed_elems_bf16x3(int64_t taps, int64_t cin, int64_t cout, int mode);   /* in 4-byte units */
int df_conv_pack_weights_bf16x3(const float* w, float* wp, int64_t taps, int64_t cin, int64_t cout, int mode,
                                df_stream_t stream);
int df_conv_fwd_bf16x3(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                       float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int flags,
                       float leak, df_stream_t stream);
int64_t df_upconv_packed_elems_bf16x3(int64_t cin, int64_t cout, int kz, int mode);
int df_upconv_pack_weights_bf16x3(const float* w, float* wp, int64_t cin, int64_t cout, int kz, int mode,
                                  df_stream_t stream);
int df_upconv_fwd_bf16x3(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Dc, int64_t Hc,
                         int64_t Wc, int64_t Cin, int64_t Cout, int kz, int flags, float leak, df_stream_t stream);
int df_upconv_dgrad_bf16x3(const float* g, const float* wp, float* acc, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                           int64_t Cin, int64_t Cout, int kz, df_stream_t stream);
/* weight gradients in the same mode (workspaces: the fp32 *_workspace_bytes); rows with W not in {16,32,64,96,112} or thin
 * channels fall back to the exact fp32 kernels. */
int df_conv_wgrad_bf16x3(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                         int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream);
int df_upconv_wgrad_bf16x3(const float* xc, const float* gy, float* gw, float* gb, int64_t B, int64_t Dc, int64_t Hc,
                           int64_t Wc, int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes,
                           df_stream_t stream);

/* gw[kz,3,3,Cin,Cout] = sum_voxels x[voxel+tap][cin] * gy[voxel][cout]   (split over voxel ranges,
 * deterministic second-pass reduction through the workspace).  If gb != NULL it also receives the bias gradient
 * gb[cout] = sum_voxels gy[voxel][cout] (accumulated on the fly from the operand registers). */
int64_t df_conv_wgrad_workspace_bytes(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz);
int df_conv_wgrad(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                  int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream);
/* The same with the algorithm chosen by the caller instead of by size (df_conv_wgrad == algo 0):
 *   algo & 7: 0 best available | 1 direct kernels only | 2 at most Winograd in x | 3 Winograd F(2x2,3x3) in (x,y) wherever
 *             instantiated | 4 Winograd F(2x2x2,3x3x3) in (x,y,z) wherever instantiated (shapes without that form fall
 *             back towards 0 -- every choice returns the same gradient up to fp32 summation order);
 *   algo >> 3: number of voxel ranges of the partial sums (0 = default; <= 256).
 * Workspace: df_conv_wgrad_workspace_bytes covers every `algo & 7` with the default range count; a range override that needs
 * more is rejected with DF_EWORKSPACE. */
int df_conv_wgrad_algo(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                       int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, int algo,
                       df_stream_t stream);
/* Which kernel family df_conv_wgrad_algo runs for these arguments (16-byte aligned operands assumed) -- the size-based choices made
 * visible so that a silent fall-back to a slower form shows up in the caller's log (bench.py `dispatch`):
 *   0 direct MFMA | 1 Winograd in x | 2 Winograd in (x,y) | 3 Winograd in (x,y,z) | 10 thin layer (Cin or Cout <= 4) on the matrix
 *   cores | 11 thin layer on the vector ALU; negative = invalid arguments.  df_upconv_wgrad_form: 3 = 27-point Winograd-(x,y,z)
 *   form on the coarse input, 0 = parity-class kernels.  (No reference counterpart: TF picks cuDNN algorithms internally.) */
int df_conv_wgrad_form(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int algo);
int df_upconv_wgrad_form(int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, int kz, int algo);

#ifdef __cplusplus
}
#endif
#endif /* DEEPFLUIDS_HIP_H */
