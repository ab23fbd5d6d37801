// General-shape SAME convolution and nearest-neighbour resize: the WRAPPER GENERALITY of the reference's call surface
// (ops.py:12-16 `conv2d / conv3d(x, o_dim, k=4, s=2, ...)` = slim.conv2d / conv3d with ANY cubic kernel and stride;
//  ops.py:66-73 `resize_nearest_neighbor(x, new_size)` = tf.image.resize_nearest_neighbor with ANY target size).
// The reference's trainers only ever call k = 3, s in {1, 2} on even extents and exact 2x resizes -- those shapes run on the matrix-core /
// vectorised kernels of conv*.hip / elementwise.hip.  Everything else lands here: plain vector-ALU kernels, one thread per output element,
// fixed summation order (deterministic), written for correctness on any shape, not for speed.
//
// TF 'SAME' (SURVEY A.3): out = ceil(n / s), pad_total = max((out - 1) s + k - n, 0), pad_before = pad_total / 2 (the extra cell goes AFTER).
// TF1 nearest resize, align_corners = False: src = min(floor(dst * in / out), in - 1).
#include "df_common.hpp"

namespace {

using df::ceil_div;
constexpr int kT = 256;

struct GenGeo {
  int B, D, H, W, Cin, Cout;      // input extents (D = 1 for 2-D)
  int Do, Ho, Wo;                 // output extents
  int kz, k, s;                   // kernel extent in z (1 for 2-D) and in y/x; stride (z stride 1 when kz == 1 and D == 1)
  int pz, py, px;                 // pad_before per axis
  int sz;                         // stride in z
};

__host__ __device__ inline int same_out(int n, int s) { return (n + s - 1) / s; }
__host__ __device__ inline int same_pad_before(int n, int k, int s) {
  const int o = (n + s - 1) / s;
  int pt = (o - 1) * s + k - n;
  if (pt < 0) pt = 0;
  return pt / 2;
}

// y[b, oz, oy, ox, co] = bias[co] + sum_{tz, ty, tx, ci} x[b, oz sz - pz + tz, oy s - py + ty, ox s - px + tx, ci] w[tz][ty][tx][ci][co]
__global__ __launch_bounds__(kT) void conv_gen_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, GenGeo g, int64_t total, int lrelu, float leak) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  if (i >= total) return;
  const int co = static_cast<int>(i % g.Cout);
  int64_t r = i / g.Cout;
  const int ox = static_cast<int>(r % g.Wo); r /= g.Wo;
  const int oy = static_cast<int>(r % g.Ho); r /= g.Ho;
  const int oz = static_cast<int>(r % g.Do);
  const int64_t b = r / g.Do;
  float acc = bias ? bias[co] : 0.f;
  for (int tz = 0; tz < g.kz; ++tz) {
    const int iz = oz * g.sz - g.pz + tz;
    if (iz < 0 || iz >= g.D) continue;
    for (int ty = 0; ty < g.k; ++ty) {
      const int iy = oy * g.s - g.py + ty;
      if (iy < 0 || iy >= g.H) continue;
      for (int tx = 0; tx < g.k; ++tx) {
        const int ix = ox * g.s - g.px + tx;
        if (ix < 0 || ix >= g.W) continue;
        const float* xp = x + (((b * g.D + iz) * g.H + iy) * g.W + ix) * g.Cin;
        const float* wp = w + (static_cast<int64_t>((tz * g.k + ty) * g.k + tx) * g.Cin) * g.Cout + co;
        for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(xp[ci], wp[static_cast<int64_t>(ci) * g.Cout], acc);
      }
    }
  }
  if (lrelu) acc = fmaxf(acc, leak * acc);
  y[i] = acc;
}

// gx[b, iz, iy, ix, ci] = sum over (tap, co) with an output cell o such that o stride - pad + t = i:  gy[b, o, co] w[t][ci][co]
__global__ __launch_bounds__(kT) void conv_gen_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, GenGeo g,
                                                            int64_t total) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  if (i >= total) return;
  const int ci = static_cast<int>(i % g.Cin);
  int64_t r = i / g.Cin;
  const int ix = static_cast<int>(r % g.W); r /= g.W;
  const int iy = static_cast<int>(r % g.H); r /= g.H;
  const int iz = static_cast<int>(r % g.D);
  const int64_t b = r / g.D;
  float acc = 0.f;
  for (int tz = 0; tz < g.kz; ++tz) {
    const int nz = iz + g.pz - tz;
    if (nz < 0 || nz % g.sz) continue;
    const int oz = nz / g.sz;
    if (oz >= g.Do) continue;
    for (int ty = 0; ty < g.k; ++ty) {
      const int ny = iy + g.py - ty;
      if (ny < 0 || ny % g.s) continue;
      const int oy = ny / g.s;
      if (oy >= g.Ho) continue;
      for (int tx = 0; tx < g.k; ++tx) {
        const int nx = ix + g.px - tx;
        if (nx < 0 || nx % g.s) continue;
        const int ox = nx / g.s;
        if (ox >= g.Wo) continue;
        const float* gp = gy + (((b * g.Do + oz) * g.Ho + oy) * g.Wo + ox) * g.Cout;
        const float* wp = w + (static_cast<int64_t>((tz * g.k + ty) * g.k + tx) * g.Cin + ci) * g.Cout;
        for (int co = 0; co < g.Cout; ++co) acc = fmaf(gp[co], wp[co], acc);
      }
    }
  }
  gx[i] = acc;
}

// gw[t][ci][co] = sum_{b, o} x[b, o stride - pad + t, ci] gy[b, o, co];  workgroup = one (tap, ci), 64 output channels at a time x 4 voxel
// sub-ranges (contiguous quarters of the (b, o) sequence, each summed in order, the four partials combined in a fixed order)
__global__ __launch_bounds__(kT) void conv_gen_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gw, GenGeo g) {
  __shared__ float sP[4][64];
  const int tap = blockIdx.x / g.Cin, ci = blockIdx.x % g.Cin;
  const int tx = tap % g.k, ty = (tap / g.k) % g.k, tz = tap / (g.k * g.k);
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t nvox = static_cast<int64_t>(g.B) * g.Do * g.Ho * g.Wo;
  const int64_t per = (nvox + 3) / 4, v0 = sub * per, v1 = v0 + per < nvox ? v0 + per : nvox;
  for (int c0 = 0; c0 < g.Cout; c0 += 64) {
    const int co = c0 + lane;
    float acc = 0.f;
    if (co < g.Cout) {
      for (int64_t v = v0; v < v1; ++v) {
        int64_t r = v;
        const int ox = static_cast<int>(r % g.Wo); r /= g.Wo;
        const int oy = static_cast<int>(r % g.Ho); r /= g.Ho;
        const int oz = static_cast<int>(r % g.Do);
        const int64_t b = r / g.Do;
        const int iz = oz * g.sz - g.pz + tz, iy = oy * g.s - g.py + ty, ix = ox * g.s - g.px + tx;
        if (iz < 0 || iz >= g.D || iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
        acc = fmaf(x[(((b * g.D + iz) * g.H + iy) * g.W + ix) * g.Cin + ci], gy[v * g.Cout + co], acc);
      }
    }
    sP[sub][lane] = acc;
    __syncthreads();
    if (sub == 0 && co < g.Cout) gw[(static_cast<int64_t>(tap) * g.Cin + ci) * g.Cout + co] = ((sP[0][lane] + sP[1][lane]) + sP[2][lane]) + sP[3][lane];
    __syncthreads();
  }
}

// gb[co] = sum over all (b, o) of gy: 64 channels x 16 voxel sub-ranges per workgroup, each sub-range summed in order, the 16 partial sums
// combined in a fixed order (deterministic).  [r5] was one thread per channel over ALL voxels; like conv_gen_wgrad_kernel this stays a
// correctness path (a few thousand threads in flight) for call sites the reference never uses -- seconds per call on 10^7-voxel grids.
__global__ __launch_bounds__(1024) void conv_gen_bgrad_kernel(const float* __restrict__ gy, float* __restrict__ gb, int64_t nvox, int Cout) {
  __shared__ float sB[16][64];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int co = blockIdx.x * 64 + lane;
  const int64_t per = (nvox + 15) / 16, v0 = sub * per, v1 = v0 + per < nvox ? v0 + per : nvox;
  float acc = 0.f;
  if (co < Cout)
    for (int64_t v = v0; v < v1; ++v) acc += gy[v * Cout + co];
  sB[sub][lane] = acc;
  __syncthreads();
  if (sub == 0 && co < Cout) {
    float t = sB[0][lane];
#pragma unroll
    for (int i = 1; i < 16; ++i) t += sB[i][lane];
    gb[co] = t;
  }
}

struct RsGeo {
  int B, D, H, W, C, Do, Ho, Wo;
};
__device__ __forceinline__ int nn_src(int dst, int in, int out) {
  const int s = static_cast<int>((static_cast<int64_t>(dst) * in) / out);
  return s < in - 1 ? s : in - 1;
}
__global__ __launch_bounds__(kT) void resize_nn_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, RsGeo g, int64_t total) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  if (i >= total) return;
  const int c = static_cast<int>(i % g.C);
  int64_t r = i / g.C;
  const int ox = static_cast<int>(r % g.Wo); r /= g.Wo;
  const int oy = static_cast<int>(r % g.Ho); r /= g.Ho;
  const int oz = static_cast<int>(r % g.Do);
  const int64_t b = r / g.Do;
  y[i] = x[(((b * g.D + nn_src(oz, g.D, g.Do)) * g.H + nn_src(oy, g.H, g.Ho)) * g.W + nn_src(ox, g.W, g.Wo)) * g.C + c];
}
// gather form of the adjoint: input cell i collects the output cells that read it, dst in [ceil(i out / in), ceil((i + 1) out / in)), in order
// (the last input cell also takes every dst the clamp sends to it -- none for floor(dst in / out) <= in - 1, which always holds)
__device__ __forceinline__ int nn_lo(int i, int in, int out) { return static_cast<int>((static_cast<int64_t>(i) * out + in - 1) / in); }
__global__ __launch_bounds__(kT) void resize_nn_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, RsGeo g, int64_t total) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  if (i >= total) return;
  const int c = static_cast<int>(i % g.C);
  int64_t r = i / g.C;
  const int ix = static_cast<int>(r % g.W); r /= g.W;
  const int iy = static_cast<int>(r % g.H); r /= g.H;
  const int iz = static_cast<int>(r % g.D);
  const int64_t b = r / g.D;
  const int z0 = nn_lo(iz, g.D, g.Do), z1 = nn_lo(iz + 1, g.D, g.Do), y0 = nn_lo(iy, g.H, g.Ho), y1 = nn_lo(iy + 1, g.H, g.Ho);
  const int x0 = nn_lo(ix, g.W, g.Wo), x1 = nn_lo(ix + 1, g.W, g.Wo);
  float acc = 0.f;
  for (int oz = z0; oz < z1 && oz < g.Do; ++oz)
    for (int oy = y0; oy < y1 && oy < g.Ho; ++oy)
      for (int ox = x0; ox < x1 && ox < g.Wo; ++ox) acc += gy[(((b * g.Do + oz) * g.Ho + oy) * g.Wo + ox) * g.C + c];
  gx[i] = acc;
}

int make_geo(GenGeo& g, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int k, int s, const char* what) {
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "%s: non-positive extent", what);
  DF_REQUIRE(k >= 1 && k <= 7 && s >= 1 && s <= 4 && (kz == k || (kz == 1 && D == 1)), DF_ESHAPE,
             "%s: cubic kernels 1 <= k <= 7 (kz = k, or kz = 1 with D = 1 for 2-D), strides 1 <= s <= 4", what);
  g.B = (int)B; g.D = (int)D; g.H = (int)H; g.W = (int)W; g.Cin = (int)Cin; g.Cout = (int)Cout;
  const bool zaxis = D > 1 || kz > 1;      // 2-D = (D == 1, kz == 1); a 3-D 1x1x1 kernel (kz = k = 1, D > 1) still strides along z
  g.kz = kz; g.k = k; g.s = s; g.sz = zaxis ? s : 1;
  g.Do = zaxis ? same_out(g.D, s) : 1; g.Ho = same_out(g.H, s); g.Wo = same_out(g.W, s);
  g.pz = zaxis ? same_pad_before(g.D, kz, s) : 0; g.py = same_pad_before(g.H, k, s); g.px = same_pad_before(g.W, k, s);
  DF_REQUIRE(B * D * H * W * Cin < (1LL << 40) && B * g.Do * g.Ho * g.Wo * Cout < (1LL << 40), DF_ESHAPE, "%s: tensor too large", what);
  return DF_OK;
}

}  // namespace

extern "C" {

int df_conv_general_out_dims(int64_t D, int64_t H, int64_t W, int kz, int k, int s, int64_t* Do, int64_t* Ho, int64_t* Wo) {
  DF_REQUIRE(Do && Ho && Wo && D > 0 && H > 0 && W > 0 && s >= 1, DF_EINVAL, "df_conv_general_out_dims: bad argument");
  (void)k;
  *Do = (D > 1 || kz > 1) ? same_out((int)D, s) : 1; *Ho = same_out((int)H, s); *Wo = same_out((int)W, s);
  return DF_OK;
}

int df_conv_general_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin,
                        int64_t Cout, int kz, int k, int s, int flags, float leak, df_stream_t stream) {
  DF_REQUIRE(x && w && y, DF_EINVAL, "df_conv_general_fwd: null pointer");
  DF_REQUIRE(!(flags & ~(DF_CONV_BIAS | DF_CONV_LRELU)) && (!(flags & DF_CONV_BIAS) || bias), DF_EINVAL,
             "df_conv_general_fwd: flags DF_CONV_BIAS | DF_CONV_LRELU only; DF_CONV_BIAS needs bias");
  GenGeo g;
  if (int e = make_geo(g, B, D, H, W, Cin, Cout, kz, k, s, "df_conv_general_fwd")) return e;
  const int64_t total = B * g.Do * g.Ho * g.Wo * Cout;
  hipLaunchKernelGGL(conv_gen_fwd_kernel, dim3((unsigned)ceil_div(total, kT)), dim3(kT), 0, df::as_stream(stream), x, w,
                     (flags & DF_CONV_BIAS) ? bias : nullptr, y, g, total, (flags & DF_CONV_LRELU) ? 1 : 0, leak);
  return df::launched("df_conv_general_fwd");
}

int df_conv_general_dgrad(const float* gy, const float* w, float* gx, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz,
                          int k, int s, df_stream_t stream) {
  DF_REQUIRE(gy && w && gx, DF_EINVAL, "df_conv_general_dgrad: null pointer");
  GenGeo g;
  if (int e = make_geo(g, B, D, H, W, Cin, Cout, kz, k, s, "df_conv_general_dgrad")) return e;
  const int64_t total = B * D * H * W * Cin;
  hipLaunchKernelGGL(conv_gen_dgrad_kernel, dim3((unsigned)ceil_div(total, kT)), dim3(kT), 0, df::as_stream(stream), gy, w, gx, g, total);
  return df::launched("df_conv_general_dgrad");
}

int df_conv_general_wgrad(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin,
                          int64_t Cout, int kz, int k, int s, df_stream_t stream) {
  DF_REQUIRE(x && gy && gw, DF_EINVAL, "df_conv_general_wgrad: null pointer");
  GenGeo g;
  if (int e = make_geo(g, B, D, H, W, Cin, Cout, kz, k, s, "df_conv_general_wgrad")) return e;
  const int64_t nblk = static_cast<int64_t>(kz) * k * k * Cin;
  DF_REQUIRE(nblk < (1LL << 31), DF_ESHAPE, "df_conv_general_wgrad: too many (tap, cin) pairs");
  hipLaunchKernelGGL(conv_gen_wgrad_kernel, dim3((unsigned)nblk), dim3(kT), 0, df::as_stream(stream), x, gy, gw, g);
  if (gb)
    hipLaunchKernelGGL(conv_gen_bgrad_kernel, dim3((unsigned)ceil_div(Cout, 64)), dim3(1024), 0, df::as_stream(stream), gy,
                       gb, B * g.Do * g.Ho * g.Wo, (int)Cout);
  return df::launched("df_conv_general_wgrad");
}

int df_resize_nn_fwd(const float* x, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int64_t Do, int64_t Ho, int64_t Wo,
                     df_stream_t stream) {
  DF_REQUIRE(x && y, DF_EINVAL, "df_resize_nn_fwd: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Do > 0 && Ho > 0 && Wo > 0, DF_EINVAL, "df_resize_nn_fwd: non-positive extent");
  const int64_t total = B * Do * Ho * Wo * C;
  DF_REQUIRE(total < (1LL << 40) && B * D * H * W * C < (1LL << 40), DF_ESHAPE, "df_resize_nn_fwd: tensor too large");
  RsGeo g{(int)B, (int)D, (int)H, (int)W, (int)C, (int)Do, (int)Ho, (int)Wo};
  hipLaunchKernelGGL(resize_nn_fwd_kernel, dim3((unsigned)ceil_div(total, kT)), dim3(kT), 0, df::as_stream(stream), x, y, g, total);
  return df::launched("df_resize_nn_fwd");
}

int df_resize_nn_bwd(const float* gy, float* gx, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int64_t Do, int64_t Ho, int64_t Wo,
                     df_stream_t stream) {
  DF_REQUIRE(gy && gx, DF_EINVAL, "df_resize_nn_bwd: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Do > 0 && Ho > 0 && Wo > 0, DF_EINVAL, "df_resize_nn_bwd: non-positive extent");
  const int64_t total = B * D * H * W * C;
  DF_REQUIRE(total < (1LL << 40) && B * Do * Ho * Wo * C < (1LL << 40), DF_ESHAPE, "df_resize_nn_bwd: tensor too large");
  RsGeo g{(int)B, (int)D, (int)H, (int)W, (int)C, (int)Do, (int)Ho, (int)Wo};
  hipLaunchKernelGGL(resize_nn_bwd_kernel, dim3((unsigned)ceil_div(total, kT)), dim3(kT), 0, df::as_stream(stream), gy, gx, g, total);
  return df::launched("df_resize_nn_bwd");
}

}  // extern "C"
