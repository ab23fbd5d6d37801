// HBM-bound small kernels of the train step: L1 losses (trainer.py:170-171), lrelu (ops.py:9-10),
// residual add (model.py:35), nearest 2x up-sampling (ops.py:66-91), fully connected (ops.py:23-24),
// bias-gradient column sums, TF1 Adam (trainer.py:160-162).  All fp32, float4-vectorised where the
// layout allows, grid-stride over at most 2048 workgroups (8 per CU).
#include "df_common.hpp"

namespace {

using df::ceil_div;
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;
constexpr int kEwUnroll = 4;          // float4 per thread of the lrelu / add kernels (one tile per workgroup, no grid-stride loop)

inline unsigned grid_for(int64_t work_items) {
  int64_t g = ceil_div(work_items, kThreads);
  return static_cast<unsigned>(g < 1 ? 1 : (g > kMaxBlocks ? kMaxBlocks : g));
}

inline unsigned ew_grid(int64_t n) {   // tiles of kEwUnroll * 256 float4; >= 1 so that the < 4-element tail is written
  const int64_t g = ceil_div(n >> 2, static_cast<int64_t>(kEwUnroll) * kThreads);
  return static_cast<unsigned>(g < 1 ? 1 : g);
}

// ---- block reduction: wave shuffle -> LDS -> first wave ------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double part[kThreads / 64];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) part[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (wid == 0) {
    r = lane < kThreads / 64 ? part[lane] : 0.0;
    r = wave_sum(r);
  }
  __syncthreads();
  return r;   // valid in thread 0
}

// ---- L1 mean --------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              int64_t n, double* __restrict__ partial) {
  const int64_t n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float acc = 0.f;   // per-thread fp32 partial over <= a few thousand terms, promoted to fp64 below
  double dacc = 0.0;
  int cnt = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const float4 p = a4[i], q = b4[i];
    acc += (fabsf(p.x - q.x) + fabsf(p.y - q.y)) + (fabsf(p.z - q.z) + fabsf(p.w - q.w));
    if (++cnt == 64) { dacc += acc; acc = 0.f; cnt = 0; }
  }
  dacc += acc;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dacc += fabsf(a[(n4 << 2) + threadIdx.x] - b[(n4 << 2) + threadIdx.x]);
  const double s = block_sum(dacc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(kThreads) void l1_final_kernel(const double* __restrict__ partial, int nparts, double inv_n,
                                                            float* __restrict__ out) {
  double v = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kThreads) v += partial[i];
  const double s = block_sum(v);
  if (threadIdx.x == 0) out[0] = static_cast<float>(s * inv_n);
}

__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(kThreads) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ gout, float scale,
                                                          float* __restrict__ ga, int64_t n) {
  const float s = scale * (gout ? gout[0] : 1.f);
  const int64_t n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* g4 = reinterpret_cast<float4*>(ga);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const float4 p = a4[i], q = b4[i];
    float4 r;
    r.x = sgn(p.x - q.x) * s; r.y = sgn(p.y - q.y) * s; r.z = sgn(p.z - q.z) * s; r.w = sgn(p.w - q.w) * s;
    g4[i] = r;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    ga[i] = sgn(a[i] - b[i]) * s;
  }
}

// ---- lrelu / add ----------------------------------------------------------------------------------
template <int OP>   // 0: lrelu fwd (a=x)  1: lrelu bwd (a=gy, b=y)  2: add
__global__ __launch_bounds__(kThreads) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ y, float leak, int64_t n) {
  auto f = [&](float p, float q) -> float {
    if (OP == 0) return fmaxf(p, leak * p);
    if (OP == 1) return q > 0.f ? p : leak * p;
    return p + q;
  };
  const int64_t n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* y4 = reinterpret_cast<float4*>(y);
  // one workgroup = one contiguous tile of kEwUnroll * 256 float4 (16 KiB per operand); all loads first, clamped index
  // instead of a branch.  Measured at 805 M floats: 4.9 TB/s as a grid-stride loop, the tile form below as torch's add (6.0)
  constexpr int U = kEwUnroll;
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * (U * kThreads) + threadIdx.x;
  if (i0 - threadIdx.x < n4) {          // workgroup-uniform (false only for n < 4)
    float4 p[U], q[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i = i0 + k * kThreads;
      const int64_t ic = i < n4 ? i : n4 - 1;
      p[k] = a4[ic];
      q[k] = OP != 0 ? b4[ic] : p[k];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i = i0 + k * kThreads;
      float4 r;
      r.x = f(p[k].x, q[k].x); r.y = f(p[k].y, q[k].y); r.z = f(p[k].z, q[k].z); r.w = f(p[k].w, q[k].w);
      if (i < n4) y4[i] = r;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    y[i] = f(a[i], OP != 0 ? b[i] : 0.f);
  }
}

// ---- channel concat / split (encoder skip connections, model.py:138,174) ---------------------------------------------
// y[r][0:Ca] = a[r], y[r][Ca:Ca+Cb] = b[r]   (DIR 0)   |   the reverse scatter of the gradient (DIR 1)
template <int DIR>
__global__ __launch_bounds__(kThreads) void concat2_kernel(float* __restrict__ a, float* __restrict__ b,
                                                           float* __restrict__ y, int64_t rows, int Ca4, int Cb4) {
  const int C4 = Ca4 + Cb4;
  const int64_t n4 = rows * C4;
  float4* a4 = reinterpret_cast<float4*>(a);
  float4* b4 = reinterpret_cast<float4*>(b);
  float4* y4 = reinterpret_cast<float4*>(y);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t r = i / C4;
    const int c = static_cast<int>(i - r * C4);
    float4* src = c < Ca4 ? a4 + r * Ca4 + c : b4 + r * Cb4 + (c - Ca4);
    if (DIR == 0) y4[i] = *src; else *src = y4[i];
  }
}

// scalar variant for channel counts that are not multiples of 4 (discriminator input: 2+1 / 3+3 channels)
template <int DIR>
__global__ __launch_bounds__(kThreads) void concat2_scalar_kernel(float* __restrict__ a, float* __restrict__ b,
                                                                  float* __restrict__ y, int64_t rows, int Ca, int Cb) {
  const int C = Ca + Cb;
  const int64_t n = rows * C;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t r = i / C;
    const int c = static_cast<int>(i - r * C);
    float* src = c < Ca ? a + r * Ca + c : b + r * Cb + (c - Ca);
    if (DIR == 0) y[i] = *src; else *src = y[i];
  }
}

// ---- zero insertion for the stride-2 conv backward: out[2o+1] = g[o] on every spatial axis, zeros elsewhere ------------
// (dX = SAME-conv(out, mirrored weights) and gW = wgrad(X, out) then reproduce the stride-2 adjoints exactly)
template <bool IS3D>
__global__ __launch_bounds__(kThreads) void dilate2_kernel(const float4* __restrict__ g, float4* __restrict__ out,
                                                           int64_t nout4, int D, int H, int W, int C4) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nout4;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C4);
    int64_t r = i / C4;
    const int w = static_cast<int>(r % (2 * W)); r /= 2 * W;
    const int h = static_cast<int>(r % (2 * H)); r /= 2 * H;
    const int D2 = IS3D ? 2 * D : 1;
    const int d = static_cast<int>(r % D2);
    const int64_t b = r / D2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((w & 1) && (h & 1) && (!IS3D || (d & 1)))
      v = g[(((b * D + (IS3D ? d >> 1 : 0)) * H + (h >> 1)) * W + (w >> 1)) * C4 + c];
    out[i] = v;
  }
}

// ---- Bernoulli-KL sparsity of the AE latent code (trainer3.py:272-277 / trainer.py:389-394) ---------------------------------------
// loss = sum_j KL(Bern(rho) || Bern(rhat_j)),  rhat_j = mean_b z[b][j],  j < n  (the code's first n of ncol columns)
//      = sum_j rho log(rho / rhat_j) + (1 - rho) log((1 - rho) / (1 - rhat_j))            (tf.distributions.kl_divergence)
// One workgroup: the code is [B, 16] in the reference's runs.
__global__ __launch_bounds__(kThreads) void kl_bernoulli_kernel(const float* __restrict__ z, const float* __restrict__ gout,
                                                                float* __restrict__ out, float* __restrict__ gz, int B, int ncol,
                                                                int n, float rho, float scale, int bwd) {
  __shared__ double sacc[kThreads];
  double acc = 0.0;
  for (int j = threadIdx.x; j < ncol; j += kThreads) {
    double m = 0.0;
    if (j < n) {
      for (int b = 0; b < B; ++b) m += static_cast<double>(z[static_cast<int64_t>(b) * ncol + j]);
      m /= B;
      acc += rho * log(rho / m) + (1.0 - rho) * log((1.0 - rho) / (1.0 - m));
    }
    if (bwd) {
      const float g = j < n ? static_cast<float>((-rho / m + (1.0 - rho) / (1.0 - m)) / B) * gout[0] * scale : 0.f;
      for (int b = 0; b < B; ++b) gz[static_cast<int64_t>(b) * ncol + j] = g;
    }
  }
  if (!bwd) {
    sacc[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int i = 0; i < kThreads; ++i) t += sacc[i];      // fixed order
      out[0] = static_cast<float>(t);
    }
  }
}

// ---- sigmoid (AE latent code when use_sparse, model.py:210) -------------------------------------------------------------
template <int DIR>
__global__ __launch_bounds__(kThreads) void sigmoid_kernel(const float* __restrict__ a, const float* __restrict__ yv,
                                                           float* __restrict__ out, int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    if (DIR == 0) out[i] = 1.f / (1.f + expf(-a[i]));
    else { const float s = yv[i]; out[i] = a[i] * s * (1.f - s); }
  }
}

// ---- mean squared difference (loss_p, trainer3.py:268-270) --------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               int64_t n, double* __restrict__ partial) {
  double acc = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const float d = a[i] - b[i];
    acc += static_cast<double>(d * d);
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(kThreads) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ gout, float scale,
                                                           float* __restrict__ ga, int64_t n) {
  const float s = scale * (gout ? gout[0] : 1.f);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kThreads)
    ga[i] = 2.f * (a[i] - b[i]) * s;
}

// ---- fully connected with LARGE K and small N (encoder head: 196,608 -> 16, model.py:150,186) --------------------------
// forward: split-K.  Workgroup g owns rows [g*KC, (g+1)*KC) of W; thread t strides over them and keeps B x N partial
// sums for ONE batch row at a time; block reduction -> partial[g][b][n]; a second kernel adds the partials in order.
constexpr int kFcChunk = 2048;
template <int NMAX>
__global__ __launch_bounds__(kThreads) void linear_splitk_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 float* __restrict__ partial, int B, int64_t K, int N) {
  __shared__ float red[kThreads / 64][NMAX];
  const int64_t k0 = static_cast<int64_t>(blockIdx.x) * kFcChunk;
  const int64_t k1 = k0 + kFcChunk < K ? k0 + kFcChunk : K;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int b = 0; b < B; ++b) {
    float acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += kThreads) {
      const float xv = x[b * K + k];
      const float* wr = w + k * N;
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) acc[n] = fmaf(xv, wr[n], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      float v = acc[n];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wid][n] = v;
    }
    __syncthreads();
    if (threadIdx.x < N) {
      float v = 0.f;
      for (int q = 0; q < kThreads / 64; ++q) v += red[q][threadIdx.x];
      partial[(static_cast<int64_t>(blockIdx.x) * B + b) * N + threadIdx.x] = v;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(kThreads) void linear_splitk_final_kernel(const float* __restrict__ partial,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ y, int nparts, int BN, int N) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= BN) return;
  double acc = 0.0;
  for (int p = 0; p < nparts; ++p) acc += partial[static_cast<int64_t>(p) * BN + i];
  y[i] = static_cast<float>(acc) + (bias ? bias[i % N] : 0.f);
}
// backward for small N: thread = k.  gw[k][:] = sum_b x[b][k] gy[b][:],  gx[b][k] = sum_n gy[b][n] w[k][n]
template <int NMAX>
__global__ __launch_bounds__(kThreads) void linear_bwd_smalln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ gy, float* __restrict__ gx,
                                                                     float* __restrict__ gw, int B, int64_t K, int N) {
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; k < K;
       k += static_cast<int64_t>(gridDim.x) * kThreads) {
    float wr[NMAX], acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) { wr[n] = n < N ? w[k * N + n] : 0.f; acc[n] = 0.f; }
    for (int b = 0; b < B; ++b) {
      const float xv = x[b * K + k];
      float dot = 0.f;
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) { const float g = gy[b * N + n]; acc[n] = fmaf(xv, g, acc[n]); dot = fmaf(g, wr[n], dot); }
      if (gx) gx[b * K + k] = dot;
    }
    if (gw) {
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) gw[k * N + n] = acc[n];
    }
  }
}

// ---- nearest 2x up-sampling -----------------------------------------------------------------------
// one thread per float4 of the SOURCE; it writes the 4 (2-D) or 8 (3-D) destination copies.
template <bool IS3D>
__global__ __launch_bounds__(kThreads) void upsample_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                                int64_t nsrc4, int D, int H, int W, int C4) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nsrc4;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C4);
    int64_t r = i / C4;
    const int w = static_cast<int>(r % W); r /= W;
    const int h = static_cast<int>(r % H); r /= H;
    const int d = static_cast<int>(r % D);
    const int64_t b = r / D;
    const float4 val = x[i];
    const int64_t W2 = 2 * W, H2 = 2 * H;
    const int64_t D2 = IS3D ? 2 * D : 1;
    const int nd = IS3D ? 2 : 1;
    for (int dz = 0; dz < nd; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int64_t zz = IS3D ? 2 * d + dz : 0;
          const int64_t o = (((b * D2 + zz) * H2 + (2 * h + dy)) * W2 + (2 * w + dx)) * C4 + c;
          y[o] = val;
        }
  }
}

template <bool IS3D>
__global__ __launch_bounds__(kThreads) void upsample_bwd_kernel(const float4* __restrict__ gy, float4* __restrict__ gx,
                                                                int64_t nsrc4, int D, int H, int W, int C4) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nsrc4;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C4);
    int64_t r = i / C4;
    const int w = static_cast<int>(r % W); r /= W;
    const int h = static_cast<int>(r % H); r /= H;
    const int d = static_cast<int>(r % D);
    const int64_t b = r / D;
    const int64_t W2 = 2 * W, H2 = 2 * H;
    const int64_t D2 = IS3D ? 2 * D : 1;
    const int nd = IS3D ? 2 : 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // summation order (dz, dy, dx) ascending == the oracle's reshape(...).sum(axes) pairwise-free order
    for (int dz = 0; dz < nd; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int64_t zz = IS3D ? 2 * d + dz : 0;
          const float4 g = gy[(((b * D2 + zz) * H2 + (2 * h + dy)) * W2 + (2 * w + dx)) * C4 + c];
          acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        }
    gx[i] = acc;
  }
}

// The tail of an up-sampling generator block in backward (model.py:36-40 / 78-82): from the incoming gradient dy of
// `lrelu(conv4(.)) + upscale(xc)` BOTH consumers in one pass over dy --
//   gx   = dy * (y > 0 ? 1 : leak)        (the gradient entering conv4: df_lrelu_bwd)
//   gpool = 2x2(x2) sum-pool of dy         (the skip path's gradient w.r.t. xc: df_upsample2x_bwd)
// One thread = one coarse voxel x 4 channels: its 4 | 8 fine float4 of dy and y are read once; summation order (dz, dy, dx)
// ascending as upsample_bwd_kernel (bit-identical results).
template <bool IS3D>
__global__ __launch_bounds__(kThreads) void lrelu_bwd_pool_kernel(const float4* __restrict__ gy, const float4* __restrict__ y,
                                                                  float4* __restrict__ gx, float4* __restrict__ gpool, float leak,
                                                                  int64_t nsrc4, int D, int H, int W, int C4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= nsrc4) return;
  const int c = static_cast<int>(i % C4);
  int64_t r = i / C4;
  const int w = static_cast<int>(r % W); r /= W;
  const int h = static_cast<int>(r % H); r /= H;
  const int d = static_cast<int>(r % D);
  const int64_t b = r / D;
  const int64_t W2 = 2 * W, H2 = 2 * H;
  const int64_t D2 = IS3D ? 2 * D : 1;
  constexpr int ND = IS3D ? 2 : 1;
  float4 g[ND * 4], a[ND * 4];
  int64_t idx[ND * 4];
#pragma unroll
  for (int k = 0; k < ND * 4; ++k) {
    const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
    const int64_t zz = IS3D ? 2 * d + dz : 0;
    idx[k] = (((b * D2 + zz) * H2 + (2 * h + dy)) * W2 + (2 * w + dx)) * C4 + c;
  }
#pragma unroll
  for (int k = 0; k < ND * 4; ++k) { g[k] = gy[idx[k]]; a[k] = y[idx[k]]; }      // all loads first
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < ND * 4; ++k) {
    acc.x += g[k].x; acc.y += g[k].y; acc.z += g[k].z; acc.w += g[k].w;
    float4 o;
    o.x = a[k].x > 0.f ? g[k].x : leak * g[k].x; o.y = a[k].y > 0.f ? g[k].y : leak * g[k].y;
    o.z = a[k].z > 0.f ? g[k].z : leak * g[k].z; o.w = a[k].w > 0.f ? g[k].w : leak * g[k].w;
    gx[idx[k]] = o;
  }
  gpool[i] = acc;
}

// y = a + nearest_up2x(bc): the block-end residual add when the block input is kept only at the coarse resolution
template <bool IS3D>
__global__ __launch_bounds__(kThreads) void add_up_kernel(const float4* __restrict__ a, const float4* __restrict__ bc,
                                                          float4* __restrict__ y, int64_t n4, int D, int H, int W, int C4) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C4);
    int64_t r = i / C4;
    const int w = static_cast<int>(r % (2 * W)); r /= 2 * W;
    const int h = static_cast<int>(r % (2 * H)); r /= 2 * H;
    const int D2 = IS3D ? 2 * D : 1;
    const int d = static_cast<int>(r % D2);
    const int64_t b = r / D2;
    const float4 p = a[i];
    const float4 q = bc[(((b * D + (IS3D ? d >> 1 : 0)) * H + (h >> 1)) * W + (w >> 1)) * C4 + c];
    y[i] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
  }
}

// ---- fully connected with tiny K (K = c_num = 3 in the generator; 16 in the AE decoder) -----------
__global__ __launch_bounds__(kThreads) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int B, int K, int64_t N) {
  // grid.y = batch row; thread -> output column (coalesced over N)
  const int b = blockIdx.y;
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; n < N;
       n += static_cast<int64_t>(gridDim.x) * kThreads) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(x[static_cast<int64_t>(b) * K + k], w[k * N + n], acc);
    y[b * N + n] = bias ? acc + bias[n] : acc;
  }
}

// gw[k][n] = sum_b x[b][k] gy[b][n];  gb[n] = sum_b gy[b][n]   (thread per column n, loop over b: coalesced)
__global__ __launch_bounds__(kThreads) void linear_bwd_w_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ gy, float* __restrict__ gw,
                                                                float* __restrict__ gb, int B, int K, int64_t N) {
  // rows k = blockIdx.y, blockIdx.y + gridDim.y, ... (grid.y is capped at 65535: any K);  k == K -> bias row
  for (int k = blockIdx.y; k <= K; k += gridDim.y)
    for (int64_t n = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; n < N;
         n += static_cast<int64_t>(gridDim.x) * kThreads) {
      float acc = 0.f;
      if (k < K) {
        for (int b = 0; b < B; ++b) acc = fmaf(x[static_cast<int64_t>(b) * K + k], gy[b * N + n], acc);
        if (gw) gw[k * N + n] = acc;
      } else {
        for (int b = 0; b < B; ++b) acc += gy[b * N + n];
        if (gb) gb[n] = acc;
      }
    }
}

// gx[b][k] = sum_n gy[b][n] w[k][n]   (one workgroup per (b,k); only used when the FC input needs a gradient)
__global__ __launch_bounds__(kThreads) void linear_bwd_x_kernel(const float* __restrict__ gy,
                                                                const float* __restrict__ w, float* __restrict__ gx,
                                                                int K, int64_t N) {
  const int b = blockIdx.y, k = blockIdx.x;
  double acc = 0.0;
  for (int64_t n = threadIdx.x; n < N; n += kThreads) acc += static_cast<double>(gy[b * N + n]) * w[k * N + n];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) gx[static_cast<int64_t>(b) * K + k] = static_cast<float>(s);
}

// ---- column sums (bias gradient of a channels-last conv) -------------------------------------------
// stage 1: workgroup `g` sums rows [g*R, (g+1)*R) of g[rows, C] into partial[g][C] (thread per column, coalesced);
// stage 2: sums the partials in fp64.  Deterministic.
constexpr int kColsumRows = 512;
__global__ __launch_bounds__(kThreads) void colsum_partial_kernel(const float* __restrict__ g, float* __restrict__ partial,
                                                                  int64_t rows, int C) {
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kColsumRows;
  const int64_t r1 = r0 + kColsumRows < rows ? r0 + kColsumRows : rows;
  for (int c = threadIdx.x; c < C; c += kThreads) {
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc += g[r * C + c];
    partial[static_cast<int64_t>(blockIdx.x) * C + c] = acc;
  }
}
__global__ __launch_bounds__(kThreads) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ gb,
                                                                int64_t nparts, int C) {
  // one workgroup per column; threads stride over the partials
  const int c = blockIdx.x;
  double acc = 0.0;
  for (int64_t p = threadIdx.x; p < nparts; p += kThreads) acc += partial[p * C + c];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) gb[c] = static_cast<float>(s);
}

// ---- TF1 Adam ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        float lr_t, float b1, float b2, float eps, float gscale) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// The same update with lr_t / grad_scale read from device memory (scal[0], scal[1]): a captured hipGraph of the train step replays with
// new values without touching the kernel node's arguments.  Identical arithmetic to adam_kernel (bitwise-equal parameters).
__global__ __launch_bounds__(kThreads) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                            const float* __restrict__ scal, float b1, float b2, float eps) {
  const float lr_t = scal[0], gscale = scal[1];
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// tf.train.GradientDescentOptimizer (trainer.py:163-165): p -= (lr * grad_scale) * g; DEV: the two scalars come from device memory.
template <bool DEV>
__global__ __launch_bounds__(kThreads) void gd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr,
                                                      float gscale, const float* __restrict__ scal) {
  if (DEV) { lr = scal[0]; gscale = scal[1]; }
  const float a = lr * gscale;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kThreads)
    p[i] = p[i] - a * g[i];
}

__global__ void store_scalars_kernel(float* __restrict__ dst, float v0, float v1, float v2, float v3, int n) {
  if (threadIdx.x == 0) {
    if (n > 0) dst[0] = v0;
    if (n > 1) dst[1] = v1;
    if (n > 2) dst[2] = v2;
    if (n > 3) dst[3] = v3;
  }
}

int check_n(const void* a, int64_t n, const char* fn) {
  DF_REQUIRE(a != nullptr, DF_EINVAL, "%s: null pointer", fn);
  DF_REQUIRE(n > 0, DF_EINVAL, "%s: n must be positive", fn);
  return DF_OK;
}

}  // namespace

extern "C" {

int64_t df_l1_mean_workspace_bytes(int64_t n) { (void)n; return static_cast<int64_t>(kMaxBlocks) * sizeof(double); }

int df_l1_mean_fwd(const float* a, const float* b, int64_t n, float* out, void* workspace, int64_t workspace_bytes,
                   df_stream_t stream) {
  if (int e = check_n(a, n, "df_l1_mean_fwd")) return e;
  DF_REQUIRE(b && out && workspace, DF_EINVAL, "df_l1_mean_fwd: null pointer");
  DF_REQUIRE(df::aligned16(a) && df::aligned16(b), DF_EALIGN, "df_l1_mean_fwd: inputs must be 16-byte aligned");
  DF_REQUIRE(workspace_bytes >= df_l1_mean_workspace_bytes(n), DF_EWORKSPACE, "df_l1_mean_fwd: workspace too small");
  const unsigned grid = grid_for(ceil_div(n, 4 * 8));
  hipStream_t s = df::as_stream(stream);
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(grid), dim3(kThreads), 0, s, a, b, n, part);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)grid, 1.0 / static_cast<double>(n), out);
  return df::launched("df_l1_mean_fwd");
}

int df_l1_mean_bwd(const float* a, const float* b, const float* gout, float scale, float* ga, int64_t n,
                   df_stream_t stream) {
  if (int e = check_n(a, n, "df_l1_mean_bwd")) return e;
  DF_REQUIRE(b && ga, DF_EINVAL, "df_l1_mean_bwd: null pointer");
  DF_REQUIRE(df::aligned16(a) && df::aligned16(b) && df::aligned16(ga), DF_EALIGN, "df_l1_mean_bwd: 16-byte alignment");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(ceil_div(n, 4))), dim3(kThreads), 0, df::as_stream(stream), a, b, gout,
                     scale / static_cast<float>(n), ga, n);
  return df::launched("df_l1_mean_bwd");
}

int df_lrelu_fwd(const float* x, float* y, float leak, int64_t n, df_stream_t stream) {
  if (int e = check_n(x, n, "df_lrelu_fwd")) return e;
  DF_REQUIRE(y, DF_EINVAL, "df_lrelu_fwd: null output");
  DF_REQUIRE(df::aligned16(x) && df::aligned16(y), DF_EALIGN, "df_lrelu_fwd: 16-byte alignment");
  hipLaunchKernelGGL((ew_kernel<0>), dim3(ew_grid(n)), dim3(kThreads), 0, df::as_stream(stream), x, x, y,
                     leak, n);
  return df::launched("df_lrelu_fwd");
}

int df_lrelu_bwd(const float* gy, const float* y, float* gx, float leak, int64_t n, df_stream_t stream) {
  if (int e = check_n(gy, n, "df_lrelu_bwd")) return e;
  DF_REQUIRE(y && gx, DF_EINVAL, "df_lrelu_bwd: null pointer");
  DF_REQUIRE(df::aligned16(gy) && df::aligned16(y) && df::aligned16(gx), DF_EALIGN, "df_lrelu_bwd: 16-byte alignment");
  hipLaunchKernelGGL((ew_kernel<1>), dim3(ew_grid(n)), dim3(kThreads), 0, df::as_stream(stream), gy, y,
                     gx, leak, n);
  return df::launched("df_lrelu_bwd");
}

int df_add(const float* a, const float* b, float* y, int64_t n, df_stream_t stream) {
  if (int e = check_n(a, n, "df_add")) return e;
  DF_REQUIRE(b && y, DF_EINVAL, "df_add: null pointer");
  DF_REQUIRE(df::aligned16(a) && df::aligned16(b) && df::aligned16(y), DF_EALIGN, "df_add: 16-byte alignment");
  hipLaunchKernelGGL((ew_kernel<2>), dim3(ew_grid(n)), dim3(kThreads), 0, df::as_stream(stream), a, b, y,
                     0.f, n);
  return df::launched("df_add");
}

int df_upsample2x_fwd(const float* x, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                      df_stream_t stream) {
  DF_REQUIRE(x && y, DF_EINVAL, "df_upsample2x_fwd: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DF_EINVAL, "df_upsample2x_fwd: non-positive extent");
  DF_REQUIRE(C % 4 == 0, DF_ESHAPE, "df_upsample2x_fwd: C must be a multiple of 4 (got %lld)", (long long)C);
  DF_REQUIRE(is_3d || D == 1, DF_ESHAPE, "df_upsample2x_fwd: D must be 1 for 2-D");
  DF_REQUIRE(df::aligned16(x) && df::aligned16(y), DF_EALIGN, "df_upsample2x_fwd: 16-byte alignment");
  const int64_t n4 = B * D * H * W * (C / 4);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* y4 = reinterpret_cast<float4*>(y);
  if (is_3d)
    hipLaunchKernelGGL((upsample_fwd_kernel<true>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), x4, y4,
                       n4, (int)D, (int)H, (int)W, (int)(C / 4));
  else
    hipLaunchKernelGGL((upsample_fwd_kernel<false>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), x4, y4,
                       n4, (int)D, (int)H, (int)W, (int)(C / 4));
  return df::launched("df_upsample2x_fwd");
}

int df_upsample2x_bwd(const float* gy, float* gx, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                      df_stream_t stream) {
  DF_REQUIRE(gy && gx, DF_EINVAL, "df_upsample2x_bwd: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DF_EINVAL, "df_upsample2x_bwd: non-positive extent");
  DF_REQUIRE(C % 4 == 0, DF_ESHAPE, "df_upsample2x_bwd: C must be a multiple of 4 (got %lld)", (long long)C);
  DF_REQUIRE(is_3d || D == 1, DF_ESHAPE, "df_upsample2x_bwd: D must be 1 for 2-D");
  DF_REQUIRE(df::aligned16(gy) && df::aligned16(gx), DF_EALIGN, "df_upsample2x_bwd: 16-byte alignment");
  const int64_t n4 = B * D * H * W * (C / 4);
  const float4* g4 = reinterpret_cast<const float4*>(gy);
  float4* x4 = reinterpret_cast<float4*>(gx);
  if (is_3d)
    hipLaunchKernelGGL((upsample_bwd_kernel<true>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), g4, x4,
                       n4, (int)D, (int)H, (int)W, (int)(C / 4));
  else
    hipLaunchKernelGGL((upsample_bwd_kernel<false>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), g4, x4,
                       n4, (int)D, (int)H, (int)W, (int)(C / 4));
  return df::launched("df_upsample2x_bwd");
}

int df_lrelu_bwd_pool2x(const float* gy, const float* y, float* gx, float* gpool, float leak, int64_t B, int64_t D, int64_t H,
                        int64_t W, int64_t C, int is_3d, df_stream_t stream) {
  DF_REQUIRE(gy && y && gx && gpool, DF_EINVAL, "df_lrelu_bwd_pool2x: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DF_EINVAL, "df_lrelu_bwd_pool2x: non-positive extent");
  DF_REQUIRE(C % 4 == 0, DF_ESHAPE, "df_lrelu_bwd_pool2x: C must be a multiple of 4 (got %lld)", (long long)C);
  DF_REQUIRE(is_3d || D == 1, DF_ESHAPE, "df_lrelu_bwd_pool2x: D must be 1 for 2-D");
  DF_REQUIRE(df::aligned16(gy) && df::aligned16(y) && df::aligned16(gx) && df::aligned16(gpool), DF_EALIGN,
             "df_lrelu_bwd_pool2x: 16-byte alignment");
  const int64_t n4 = B * D * H * W * (C / 4);
  DF_REQUIRE(ceil_div(n4, kThreads) < (1LL << 31), DF_ESHAPE, "df_lrelu_bwd_pool2x: tensor too large");
  const dim3 grid((unsigned)ceil_div(n4, kThreads));
  const float4* g4 = reinterpret_cast<const float4*>(gy);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  float4* x4 = reinterpret_cast<float4*>(gx);
  float4* p4 = reinterpret_cast<float4*>(gpool);
  if (is_3d)
    hipLaunchKernelGGL((lrelu_bwd_pool_kernel<true>), grid, dim3(kThreads), 0, df::as_stream(stream), g4, y4, x4, p4, leak, n4, (int)D,
                       (int)H, (int)W, (int)(C / 4));
  else
    hipLaunchKernelGGL((lrelu_bwd_pool_kernel<false>), grid, dim3(kThreads), 0, df::as_stream(stream), g4, y4, x4, p4, leak, n4, (int)D,
                       (int)H, (int)W, (int)(C / 4));
  return df::launched("df_lrelu_bwd_pool2x");
}

int64_t df_linear_workspace_bytes(int64_t B, int64_t K, int64_t N) {
  if (N <= 32 && K >= 1024) return ceil_div(K, kFcChunk) * B * N * static_cast<int64_t>(sizeof(float));
  return 0;
}

int df_linear_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t K, int64_t N,
                  void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  DF_REQUIRE(x && w && y, DF_EINVAL, "df_linear_fwd: null pointer");
  DF_REQUIRE(B > 0 && K > 0 && N > 0 && B < 65536, DF_EINVAL, "df_linear_fwd: bad extent");
  if (N <= 32 && K >= 1024) {
    DF_REQUIRE(workspace && workspace_bytes >= df_linear_workspace_bytes(B, K, N), DF_EWORKSPACE,
               "df_linear_fwd: workspace too small for the split-K path");
    const int nparts = (int)ceil_div(K, kFcChunk);
    float* part = static_cast<float*>(workspace);
    hipStream_t s = df::as_stream(stream);
    if (N <= 16) hipLaunchKernelGGL((linear_splitk_kernel<16>), dim3((unsigned)nparts), dim3(kThreads), 0, s, x, w, part, (int)B, K, (int)N);
    else hipLaunchKernelGGL((linear_splitk_kernel<32>), dim3((unsigned)nparts), dim3(kThreads), 0, s, x, w, part, (int)B, K, (int)N);
    hipLaunchKernelGGL(linear_splitk_final_kernel, dim3((unsigned)ceil_div(B * N, kThreads)), dim3(kThreads), 0, s, part, bias, y,
                       nparts, (int)(B * N), (int)N);
    return df::launched("df_linear_fwd(split-K)");
  }
  dim3 grid(grid_for(N), (unsigned)B);
  hipLaunchKernelGGL(linear_fwd_kernel, grid, dim3(kThreads), 0, df::as_stream(stream), x, w, bias, y, (int)B, (int)K, N);
  return df::launched("df_linear_fwd");
}

int df_linear_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int64_t B,
                  int64_t K, int64_t N, df_stream_t stream) {
  DF_REQUIRE(x && w && gy, DF_EINVAL, "df_linear_bwd: null pointer");
  DF_REQUIRE(B > 0 && K > 0 && N > 0 && B < 65536 && K < (1LL << 31) - 1, DF_EINVAL, "df_linear_bwd: bad extent");
  hipStream_t s = df::as_stream(stream);
  if (N <= 32 && K >= 1024) {
    if (gx || gw) {
      if (N <= 16) hipLaunchKernelGGL((linear_bwd_smalln_kernel<16>), dim3(grid_for(K)), dim3(kThreads), 0, s, x, w, gy, gx, gw, (int)B, K, (int)N);
      else hipLaunchKernelGGL((linear_bwd_smalln_kernel<32>), dim3(grid_for(K)), dim3(kThreads), 0, s, x, w, gy, gx, gw, (int)B, K, (int)N);
    }
    if (gb) {   // bias row only: the column-parallel kernel with k == K
      dim3 grid(grid_for(N), 1);
      hipLaunchKernelGGL(linear_bwd_w_kernel, grid, dim3(kThreads), 0, s, x, gy, (float*)nullptr, gb, (int)B, 0, N);
    }
    return df::launched("df_linear_bwd(small-N)");
  }
  if (gw || gb) {
    dim3 grid(grid_for(N), (unsigned)(K + 1 < 65535 ? K + 1 : 65535));
    hipLaunchKernelGGL(linear_bwd_w_kernel, grid, dim3(kThreads), 0, s, x, gy, gw, gb, (int)B, (int)K, N);
  }
  if (gx) {
    dim3 grid((unsigned)K, (unsigned)B);
    hipLaunchKernelGGL(linear_bwd_x_kernel, grid, dim3(kThreads), 0, s, gy, w, gx, (int)K, N);
  }
  return df::launched("df_linear_bwd");
}

int64_t df_colsum_workspace_bytes(int64_t rows, int64_t C) {
  return ceil_div(rows, kColsumRows) * C * static_cast<int64_t>(sizeof(float));
}

int df_colsum(const float* g, float* gb, int64_t rows, int64_t C, void* workspace, int64_t workspace_bytes,
              df_stream_t stream) {
  DF_REQUIRE(g && gb && workspace, DF_EINVAL, "df_colsum: null pointer");
  DF_REQUIRE(rows > 0 && C > 0 && C < (1 << 20), DF_EINVAL, "df_colsum: bad extent");
  DF_REQUIRE(workspace_bytes >= df_colsum_workspace_bytes(rows, C), DF_EWORKSPACE, "df_colsum: workspace too small");
  const int64_t nparts = ceil_div(rows, kColsumRows);
  hipStream_t s = df::as_stream(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nparts), dim3(kThreads), 0, s, g, part, rows, (int)C);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)C), dim3(kThreads), 0, s, part, gb, nparts, (int)C);
  return df::launched("df_colsum");
}

int df_adam_tf1_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                     float eps, float grad_scale, df_stream_t stream) {
  if (int e = check_n(p, n, "df_adam_tf1_step")) return e;
  DF_REQUIRE(g && m && v, DF_EINVAL, "df_adam_tf1_step: null pointer");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(kThreads), 0, df::as_stream(stream), p, g, m, v, n, lr_t, beta1,
                     beta2, eps, grad_scale);
  return df::launched("df_adam_tf1_step");
}

int df_adam_tf1_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* scalars, float beta1, float beta2,
                         float eps, df_stream_t stream) {
  if (int e = check_n(p, n, "df_adam_tf1_step_dev")) return e;
  DF_REQUIRE(g && m && v && scalars, DF_EINVAL, "df_adam_tf1_step_dev: null pointer");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(kThreads), 0, df::as_stream(stream), p, g, m, v, n, scalars, beta1,
                     beta2, eps);
  return df::launched("df_adam_tf1_step_dev");
}

int df_gd_step(float* p, const float* g, int64_t n, float lr, float grad_scale, df_stream_t stream) {
  if (int e = check_n(p, n, "df_gd_step")) return e;
  DF_REQUIRE(g, DF_EINVAL, "df_gd_step: null pointer");
  hipLaunchKernelGGL((gd_kernel<false>), dim3(grid_for(n)), dim3(kThreads), 0, df::as_stream(stream), p, g, n, lr, grad_scale,
                     (const float*)nullptr);
  return df::launched("df_gd_step");
}

int df_gd_step_dev(float* p, const float* g, int64_t n, const float* scalars, df_stream_t stream) {
  if (int e = check_n(p, n, "df_gd_step_dev")) return e;
  DF_REQUIRE(g && scalars, DF_EINVAL, "df_gd_step_dev: null pointer");
  hipLaunchKernelGGL((gd_kernel<true>), dim3(grid_for(n)), dim3(kThreads), 0, df::as_stream(stream), p, g, n, 0.f, 0.f, scalars);
  return df::launched("df_gd_step_dev");
}

int df_store_scalars(float* dst, int64_t n, float v0, float v1, float v2, float v3, df_stream_t stream) {
  DF_REQUIRE(dst, DF_EINVAL, "df_store_scalars: null pointer");
  DF_REQUIRE(n >= 1 && n <= 4, DF_EINVAL, "df_store_scalars: n must be 1..4");
  hipLaunchKernelGGL(store_scalars_kernel, dim3(1), dim3(64), 0, df::as_stream(stream), dst, v0, v1, v2, v3, (int)n);
  return df::launched("df_store_scalars");
}

int df_concat2_fwd(const float* a, const float* b, float* y, int64_t rows, int64_t Ca, int64_t Cb, df_stream_t stream) {
  DF_REQUIRE(a && b && y, DF_EINVAL, "df_concat2_fwd: null pointer");
  DF_REQUIRE(rows > 0 && Ca > 0 && Cb > 0, DF_EINVAL, "df_concat2_fwd: non-positive extent");
  if (Ca % 4 != 0 || Cb % 4 != 0 || !df::aligned16(a) || !df::aligned16(b) || !df::aligned16(y)) {
    hipLaunchKernelGGL((concat2_scalar_kernel<0>), dim3(grid_for(rows * (Ca + Cb))), dim3(kThreads), 0, df::as_stream(stream),
                       const_cast<float*>(a), const_cast<float*>(b), y, rows, (int)Ca, (int)Cb);
    return df::launched("df_concat2_fwd");
  }
  hipLaunchKernelGGL((concat2_kernel<0>), dim3(grid_for(rows * (Ca + Cb) / 4)), dim3(kThreads), 0, df::as_stream(stream),
                     const_cast<float*>(a), const_cast<float*>(b), y, rows, (int)(Ca / 4), (int)(Cb / 4));
  return df::launched("df_concat2_fwd");
}

int df_concat2_bwd(const float* gy, float* ga, float* gb, int64_t rows, int64_t Ca, int64_t Cb, df_stream_t stream) {
  DF_REQUIRE(gy && ga && gb, DF_EINVAL, "df_concat2_bwd: null pointer");
  DF_REQUIRE(rows > 0 && Ca > 0 && Cb > 0, DF_EINVAL, "df_concat2_bwd: non-positive extent");
  if (Ca % 4 != 0 || Cb % 4 != 0 || !df::aligned16(ga) || !df::aligned16(gb) || !df::aligned16(gy)) {
    hipLaunchKernelGGL((concat2_scalar_kernel<1>), dim3(grid_for(rows * (Ca + Cb))), dim3(kThreads), 0, df::as_stream(stream),
                       ga, gb, const_cast<float*>(gy), rows, (int)Ca, (int)Cb);
    return df::launched("df_concat2_bwd");
  }
  hipLaunchKernelGGL((concat2_kernel<1>), dim3(grid_for(rows * (Ca + Cb) / 4)), dim3(kThreads), 0, df::as_stream(stream),
                     ga, gb, const_cast<float*>(gy), rows, (int)(Ca / 4), (int)(Cb / 4));
  return df::launched("df_concat2_bwd");
}

int df_dilate2_odd(const float* g, float* out, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                   df_stream_t stream) {
  DF_REQUIRE(g && out, DF_EINVAL, "df_dilate2_odd: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DF_EINVAL, "df_dilate2_odd: non-positive extent");
  DF_REQUIRE(C % 4 == 0, DF_ESHAPE, "df_dilate2_odd: C must be a multiple of 4");
  DF_REQUIRE(is_3d || D == 1, DF_ESHAPE, "df_dilate2_odd: D must be 1 for 2-D");
  DF_REQUIRE(df::aligned16(g) && df::aligned16(out), DF_EALIGN, "df_dilate2_odd: 16-byte alignment");
  const int64_t n4 = B * (is_3d ? 2 * D : 1) * 2 * H * 2 * W * (C / 4);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* o4 = reinterpret_cast<float4*>(out);
  if (is_3d) hipLaunchKernelGGL((dilate2_kernel<true>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), g4, o4,
                                n4, (int)D, (int)H, (int)W, (int)(C / 4));
  else hipLaunchKernelGGL((dilate2_kernel<false>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), g4, o4, n4,
                          (int)D, (int)H, (int)W, (int)(C / 4));
  return df::launched("df_dilate2_odd");
}

int df_kl_bernoulli_fwd(const float* z, int64_t B, int64_t ncol, int64_t n, float rho, float* out, df_stream_t stream) {
  DF_REQUIRE(z && out, DF_EINVAL, "df_kl_bernoulli_fwd: null pointer");
  DF_REQUIRE(B > 0 && ncol > 0 && n >= 0 && n <= ncol && B < (1 << 24) && ncol < (1 << 24), DF_ESHAPE, "df_kl_bernoulli_fwd: bad shape");
  DF_REQUIRE(rho > 0.f && rho < 1.f, DF_EINVAL, "df_kl_bernoulli_fwd: rho must be in (0, 1)");
  hipLaunchKernelGGL(kl_bernoulli_kernel, dim3(1), dim3(kThreads), 0, df::as_stream(stream), z, nullptr, out, nullptr, (int)B, (int)ncol,
                     (int)n, rho, 1.f, 0);
  return df::launched("df_kl_bernoulli_fwd");
}

int df_kl_bernoulli_bwd(const float* z, const float* gout, float scale, float* gz, int64_t B, int64_t ncol, int64_t n, float rho,
                        df_stream_t stream) {
  DF_REQUIRE(z && gout && gz, DF_EINVAL, "df_kl_bernoulli_bwd: null pointer");
  DF_REQUIRE(B > 0 && ncol > 0 && n >= 0 && n <= ncol && B < (1 << 24) && ncol < (1 << 24), DF_ESHAPE, "df_kl_bernoulli_bwd: bad shape");
  DF_REQUIRE(rho > 0.f && rho < 1.f, DF_EINVAL, "df_kl_bernoulli_bwd: rho must be in (0, 1)");
  hipLaunchKernelGGL(kl_bernoulli_kernel, dim3(1), dim3(kThreads), 0, df::as_stream(stream), z, gout, nullptr, gz, (int)B, (int)ncol,
                     (int)n, rho, scale, 1);
  return df::launched("df_kl_bernoulli_bwd");
}

int df_sigmoid_fwd(const float* x, float* y, int64_t n, df_stream_t stream) {
  if (int e = check_n(x, n, "df_sigmoid_fwd")) return e;
  DF_REQUIRE(y, DF_EINVAL, "df_sigmoid_fwd: null output");
  hipLaunchKernelGGL((sigmoid_kernel<0>), dim3(grid_for(n)), dim3(kThreads), 0, df::as_stream(stream), x, x, y, n);
  return df::launched("df_sigmoid_fwd");
}

int df_sigmoid_bwd(const float* gy, const float* y, float* gx, int64_t n, df_stream_t stream) {
  if (int e = check_n(gy, n, "df_sigmoid_bwd")) return e;
  DF_REQUIRE(y && gx, DF_EINVAL, "df_sigmoid_bwd: null pointer");
  hipLaunchKernelGGL((sigmoid_kernel<1>), dim3(grid_for(n)), dim3(kThreads), 0, df::as_stream(stream), gy, y, gx, n);
  return df::launched("df_sigmoid_bwd");
}

int df_mse_mean_fwd(const float* a, const float* b, int64_t n, float* out, void* workspace, int64_t workspace_bytes,
                    df_stream_t stream) {
  if (int e = check_n(a, n, "df_mse_mean_fwd")) return e;
  DF_REQUIRE(b && out && workspace, DF_EINVAL, "df_mse_mean_fwd: null pointer");
  DF_REQUIRE(workspace_bytes >= df_l1_mean_workspace_bytes(n), DF_EWORKSPACE, "df_mse_mean_fwd: workspace too small");
  const unsigned grid = grid_for(n);
  hipStream_t s = df::as_stream(stream);
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(grid), dim3(kThreads), 0, s, a, b, n, part);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)grid, 1.0 / static_cast<double>(n), out);
  return df::launched("df_mse_mean_fwd");
}

int df_mse_mean_bwd(const float* a, const float* b, const float* gout, float scale, float* ga, int64_t n,
                    df_stream_t stream) {
  if (int e = check_n(a, n, "df_mse_mean_bwd")) return e;
  DF_REQUIRE(b && ga, DF_EINVAL, "df_mse_mean_bwd: null pointer");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, df::as_stream(stream), a, b, gout,
                     scale / static_cast<float>(n), ga, n);
  return df::launched("df_mse_mean_bwd");
}

int df_add_up2x(const float* a, const float* bc, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int is_3d,
                df_stream_t stream) {
  DF_REQUIRE(a && bc && y, DF_EINVAL, "df_add_up2x: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DF_EINVAL, "df_add_up2x: non-positive extent");
  DF_REQUIRE(C % 4 == 0, DF_ESHAPE, "df_add_up2x: C must be a multiple of 4");
  DF_REQUIRE(is_3d || D == 1, DF_ESHAPE, "df_add_up2x: D must be 1 for 2-D");
  DF_REQUIRE(df::aligned16(a) && df::aligned16(bc) && df::aligned16(y), DF_EALIGN, "df_add_up2x: 16-byte alignment");
  const int64_t n4 = B * (is_3d ? 2 * D : 1) * 2 * H * 2 * W * (C / 4);
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(bc);
  float4* y4 = reinterpret_cast<float4*>(y);
  if (is_3d) hipLaunchKernelGGL((add_up_kernel<true>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), a4, b4, y4,
                                n4, (int)D, (int)H, (int)W, (int)(C / 4));
  else hipLaunchKernelGGL((add_up_kernel<false>), dim3(grid_for(n4)), dim3(kThreads), 0, df::as_stream(stream), a4, b4, y4, n4,
                          (int)D, (int)H, (int)W, (int)(C / 4));
  return df::launched("df_add_up2x");
}

}  // extern "C"
