// Fused tail of the velocity-field train step (SURVEY.md 8(b) `velocity_loss2d/3d`; reference graph trainer.py:140-146,170-172 /
// trainer3.py:18-24,49-51 with the ground-truth Jacobian of trainer.py:29-32):
//
//     u      = curl(psi)            | jacobian3(psi)[1]                        (ops.py:264-274 | 255-260)
//     l1     = mean |u - x|                                                      (trainer.py:170)
//     j_l1   = mean |J(u) - J(x)|    J = jacobian | jacobian3 [0]               (trainer.py:171, ops.py:205-262)
//
// The reference graph runs this as 5 ops over 240 B/voxel (3-D): jacobian3 on the ground truth (60), jacobian3(psi)[1] (24),
// jacobian3(u) (60), two reduce_mean(abs(.)) (24 + 72).  Here: ONE forward kernel reads psi and x (each HBM byte once; the
// neighbour records are L1/L2 hits) and writes u and two fp64 partial sums -- the two 9-channel Jacobians are never
// materialised: 36 B/voxel.  Backward: one kernel rebuilds the sign pattern from (u, x) and applies the Jacobian's adjoint in
// gather form, du = w1/N1 sign(u - x) + J^T( w2/N9 sign(J(u) - J(x)) )  (36 B/voxel), then the curl adjoint of stencil.hip maps
// du to dpsi (24 B/voxel).  Every difference uses the reference's rule D f[n-1] = D f[n-2] (ops.py:214-217) and the same
// arithmetic as stencil.hip (u is bit-identical to df_jacobian3d_fwd / df_curl2d_fwd).  Sums: fp32 per thread (<= 12 terms),
// fp64 per workgroup and across workgroups in a fixed order (deterministic).
#include "df_common.hpp"

namespace {

using df::ceil_div;
constexpr int kThreads = 256;

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// two block sums at once; results valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double part[2][kThreads / 64];
  a = wave_sum(a); b = wave_sum(b);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { part[0][wid] = a; part[1][wid] = b; }
  __syncthreads();
  if (wid == 0) {
    a = lane < kThreads / 64 ? part[0][lane] : 0.0;
    b = lane < kThreads / 64 ? part[1][lane] : 0.0;
    a = wave_sum(a); b = wave_sum(b);
  }
  __syncthreads();
}
__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

// adjoint of D along an axis of extent n at position k, given a loader g(i) of the incoming gradient (as stencil.hip)
template <typename G>
__device__ __forceinline__ float adj_at(const G& g, int k, int n) {
  auto gp = [&](int i) { return i == n - 2 ? g(i) + g(i + 1) : g(i); };
  if (k == 0) return -gp(0);
  if (k == n - 1) return gp(n - 2);
  return gp(k - 1) - gp(k);
}

// ================================================= 3-D ===========================================================================
struct Geo3 {
  int64_t nvox;
  int Z, Y, X;
};

// u(w) = (D_y psi_w - D_z psi_v, D_z psi_u - D_x psi_w, D_x psi_v - D_y psi_u) at voxel w = (zz, yy, xx)   (ops.py:255-260)
__device__ __forceinline__ void curl3_at(const float* __restrict__ psi, int64_t w, int xx, int yy, int zz, const Geo3& g,
                                         float (&u)[3]) {
  const int64_t sy = g.X, sz = static_cast<int64_t>(g.X) * g.Y;
  const bool lx = xx == g.X - 1, ly = yy == g.Y - 1, lz = zz == g.Z - 1;
  const float* p = psi + w * 3;
  const float* px = psi + (lx ? w - 1 : w + 1) * 3;
  const float* py = psi + (ly ? w - sy : w + sy) * 3;
  const float* pz = psi + (lz ? w - sz : w + sz) * 3;
  const float o0 = p[0], o1 = p[1], o2 = p[2];
  const float dx1 = lx ? o1 - px[1] : px[1] - o1, dx2 = lx ? o2 - px[2] : px[2] - o2;
  const float dy0 = ly ? o0 - py[0] : py[0] - o0, dy2 = ly ? o2 - py[2] : py[2] - o2;
  const float dz0 = lz ? o0 - pz[0] : pz[0] - o0, dz1 = lz ? o1 - pz[1] : pz[1] - o1;
  u[0] = dy2 - dz1;
  u[1] = dz0 - dx2;
  u[2] = dx1 - dy0;
}

__device__ __forceinline__ void coords3(int64_t v, const Geo3& g, int& xx, int& yy, int& zz) {
  const int64_t row = v / g.X;
  xx = static_cast<int>(v - row * g.X);
  const int64_t slab = row / g.Y;
  yy = static_cast<int>(row - slab * g.Y);
  zz = static_cast<int>(slab % g.Z);
}

constexpr int kVpt3 = 4;      // voxels per thread (strided by the block size: coalesced records)

__global__ __launch_bounds__(kThreads) void velocity_loss3d_fwd_kernel(const float* __restrict__ psi, const float* __restrict__ x,
                                                                       float* __restrict__ u, double* __restrict__ partial, Geo3 g) {
  const int64_t sy = g.X, sz = static_cast<int64_t>(g.X) * g.Y;
  double s1 = 0.0, s9 = 0.0;
#pragma unroll
  for (int it = 0; it < kVpt3; ++it) {
    const int64_t v = (static_cast<int64_t>(blockIdx.x) * kVpt3 + it) * kThreads + threadIdx.x;
    if (v < g.nvox) {
      int xx, yy, zz;
      coords3(v, g, xx, yy, zz);
      float uc[3];
      curl3_at(psi, v, xx, yy, zz, g, uc);
      const float xc[3] = {x[v * 3], x[v * 3 + 1], x[v * 3 + 2]};
      float a1 = (fabsf(uc[0] - xc[0]) + fabsf(uc[1] - xc[1])) + fabsf(uc[2] - xc[2]);
      float a9 = 0.f;
#pragma unroll
      for (int axis = 0; axis < 3; ++axis) {
        const bool last = axis == 0 ? xx == g.X - 1 : axis == 1 ? yy == g.Y - 1 : zz == g.Z - 1;
        const int64_t st = axis == 0 ? 1 : axis == 1 ? sy : sz;
        const int d = last ? -1 : 1;
        const int64_t nb = v + d * st;
        float un[3];
        curl3_at(psi, nb, xx + (axis == 0 ? d : 0), yy + (axis == 1 ? d : 0), zz + (axis == 2 ? d : 0), g, un);
        const float* xn = x + nb * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float ju = last ? uc[c] - un[c] : un[c] - uc[c];
          const float jx = last ? xc[c] - xn[c] : xn[c] - xc[c];
          a9 += fabsf(ju - jx);
        }
      }
      s1 += a1; s9 += a9;
      if (u) { u[v * 3] = uc[0]; u[v * 3 + 1] = uc[1]; u[v * 3 + 2] = uc[2]; }
    }
  }
  block_sum2(s1, s9);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s1; partial[2 * blockIdx.x + 1] = s9; }
}

__global__ __launch_bounds__(kThreads) void velocity_loss_final_kernel(const double* __restrict__ partial, int nparts, double inv_n1,
                                                                       double inv_nj, float* __restrict__ l1, float* __restrict__ jl1) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kThreads) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  block_sum2(a, b);
  if (threadIdx.x == 0) { l1[0] = static_cast<float>(a * inv_n1); jl1[0] = static_cast<float>(b * inv_nj); }
}

// du[v][c] = s1 sign(u - x) + sum_axis adj_axis( s9 sign(D_axis u_c - D_axis x_c) )
__global__ __launch_bounds__(kThreads) void velocity_loss3d_bwd_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                       const float* __restrict__ g_l1, const float* __restrict__ g_jl1,
                                                                       float inv_n1, float inv_nj, float* __restrict__ du, Geo3 g) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= g.nvox) return;
  const float s1 = inv_n1 * (g_l1 ? g_l1[0] : 1.f), s9 = inv_nj * (g_jl1 ? g_jl1[0] : 1.f);
  int xx, yy, zz;
  coords3(v, g, xx, yy, zz);
  const int64_t sy = g.X, sz = static_cast<int64_t>(g.X) * g.Y;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // gradient w.r.t. D_axis(u_c) at position i of the line through v along `axis` (base = the line's first voxel)
    auto line = [&](int64_t base, int64_t st, int n) {
      return [=](int i) -> float {
        const int64_t w = base + static_cast<int64_t>(i) * st;
        const int64_t nb = i == n - 1 ? w - st : w + st;
        const float fu = u[w * 3 + c], fn = u[nb * 3 + c], fx = x[w * 3 + c], fxn = x[nb * 3 + c];
        const float ju = i == n - 1 ? fu - fn : fn - fu;
        const float jx = i == n - 1 ? fx - fxn : fxn - fx;
        return sgn(ju - jx) * s9;
      };
    };
    float acc = adj_at(line(v - xx, 1, g.X), xx, g.X);
    acc += adj_at(line(v - static_cast<int64_t>(yy) * sy, sy, g.Y), yy, g.Y);
    acc += adj_at(line(v - static_cast<int64_t>(zz) * sz, sz, g.Z), zz, g.Z);
    du[v * 3 + c] = sgn(u[v * 3 + c] - x[v * 3 + c]) * s1 + acc;
  }
}

// ================================================= 2-D ===========================================================================
struct Geo2 {
  int64_t npix;
  int Y, X;
};

// u(w) = (D_y psi, -D_x psi)   (ops.py:267-271; the negated difference is replicated like the difference)
__device__ __forceinline__ void curl2_at(const float* __restrict__ psi, int64_t w, int xx, int yy, const Geo2& g, float (&u)[2]) {
  const bool lx = xx == g.X - 1, ly = yy == g.Y - 1;
  const float own = psi[w];
  const float ny = psi[ly ? w - g.X : w + g.X], nx = psi[lx ? w - 1 : w + 1];
  u[0] = ly ? own - ny : ny - own;
  u[1] = lx ? nx - own : own - nx;
}

__global__ __launch_bounds__(kThreads) void velocity_loss2d_fwd_kernel(const float* __restrict__ psi, const float* __restrict__ x,
                                                                       float* __restrict__ u, double* __restrict__ partial, Geo2 g) {
  double s1 = 0.0, s4 = 0.0;
#pragma unroll
  for (int it = 0; it < kVpt3; ++it) {
    const int64_t v = (static_cast<int64_t>(blockIdx.x) * kVpt3 + it) * kThreads + threadIdx.x;
    if (v < g.npix) {
      const int64_t row = v / g.X;
      const int xx = static_cast<int>(v - row * g.X), yy = static_cast<int>(row % g.Y);
      float uc[2];
      curl2_at(psi, v, xx, yy, g, uc);
      const float xc[2] = {x[v * 2], x[v * 2 + 1]};
      const float a1 = fabsf(uc[0] - xc[0]) + fabsf(uc[1] - xc[1]);
      float a4 = 0.f;
#pragma unroll
      for (int axis = 0; axis < 2; ++axis) {      // axis 0: x (dudx, dvdx), axis 1: y (dudy, dvdy)   (ops.py:209-220)
        const bool last = axis == 0 ? xx == g.X - 1 : yy == g.Y - 1;
        const int d = last ? -1 : 1;
        const int64_t nb = v + d * (axis == 0 ? 1 : static_cast<int64_t>(g.X));
        float un[2];
        curl2_at(psi, nb, xx + (axis == 0 ? d : 0), yy + (axis == 1 ? d : 0), g, un);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float ju = last ? uc[c] - un[c] : un[c] - uc[c];
          const float jx = last ? xc[c] - x[nb * 2 + c] : x[nb * 2 + c] - xc[c];
          a4 += fabsf(ju - jx);
        }
      }
      s1 += a1; s4 += a4;
      if (u) { u[v * 2] = uc[0]; u[v * 2 + 1] = uc[1]; }
    }
  }
  block_sum2(s1, s4);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s1; partial[2 * blockIdx.x + 1] = s4; }
}

__global__ __launch_bounds__(kThreads) void velocity_loss2d_bwd_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                       const float* __restrict__ g_l1, const float* __restrict__ g_jl1,
                                                                       float inv_n1, float inv_nj, float* __restrict__ du, Geo2 g) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= g.npix) return;
  const float s1 = inv_n1 * (g_l1 ? g_l1[0] : 1.f), s4 = inv_nj * (g_jl1 ? g_jl1[0] : 1.f);
  const int64_t row = v / g.X;
  const int xx = static_cast<int>(v - row * g.X), yy = static_cast<int>(row % g.Y);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    auto line = [&](int64_t base, int64_t st, int n) {
      return [=](int i) -> float {
        const int64_t w = base + static_cast<int64_t>(i) * st;
        const int64_t nb = i == n - 1 ? w - st : w + st;
        const float fu = u[w * 2 + c], fn = u[nb * 2 + c], fx = x[w * 2 + c], fxn = x[nb * 2 + c];
        const float ju = i == n - 1 ? fu - fn : fn - fu;
        const float jx = i == n - 1 ? fx - fxn : fxn - fx;
        return sgn(ju - jx) * s4;
      };
    };
    float acc = adj_at(line(v - xx, 1, g.X), xx, g.X);
    acc += adj_at(line(v - static_cast<int64_t>(yy) * g.X, g.X, g.Y), yy, g.Y);
    du[v * 2 + c] = sgn(u[v * 2 + c] - x[v * 2 + c]) * s1 + acc;
  }
}

int check(const void* a, const void* b, int64_t B, int64_t Z, int64_t Y, int64_t X, const char* fn) {
  DF_REQUIRE(a != nullptr && b != nullptr, DF_EINVAL, "%s: null input", fn);
  DF_REQUIRE(B > 0 && Z > 0 && Y > 0 && X > 0, DF_EINVAL, "%s: non-positive extent", fn);
  DF_REQUIRE(Z >= 2 && Y >= 2 && X >= 2, DF_ESHAPE, "%s: forward difference needs every extent >= 2", fn);
  DF_REQUIRE(Z < (1 << 30) && Y < (1 << 30) && X < (1 << 30) && B * Z * Y * X < (1LL << 40), DF_ESHAPE, "%s: extent too large", fn);
  return DF_OK;
}

inline int64_t nblocks_fwd(int64_t n) { return ceil_div(n, static_cast<int64_t>(kThreads) * kVpt3); }

}  // namespace

extern "C" {

int df_jacobian3d_bwd(const float* gj, const float* gc, float* gx, int64_t B, int64_t Z, int64_t Y, int64_t X, df_stream_t stream);
int df_curl2d_bwd(const float* gu, float* gpsi, int64_t B, int64_t Y, int64_t X, df_stream_t stream);

int64_t df_velocity_loss3d_workspace_bytes(int64_t B, int64_t Z, int64_t Y, int64_t X) {
  if (B <= 0 || Z <= 0 || Y <= 0 || X <= 0) return 0;
  const int64_t n = B * Z * Y * X;
  const int64_t fwd = nblocks_fwd(n) * 2 * static_cast<int64_t>(sizeof(double));
  const int64_t bwd = n * 3 * static_cast<int64_t>(sizeof(float));
  return fwd > bwd ? fwd : bwd;
}

int df_velocity_loss3d_fwd(const float* psi, const float* x, float* u, float* l1, float* jl1, int64_t B, int64_t Z, int64_t Y,
                           int64_t X, void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  if (int e = check(psi, x, B, Z, Y, X, "df_velocity_loss3d_fwd")) return e;
  DF_REQUIRE(l1 && jl1 && workspace, DF_EINVAL, "df_velocity_loss3d_fwd: null output / workspace");
  const Geo3 g{B * Z * Y * X, (int)Z, (int)Y, (int)X};
  const int64_t nb = nblocks_fwd(g.nvox);
  DF_REQUIRE(workspace_bytes >= nb * 2 * static_cast<int64_t>(sizeof(double)), DF_EWORKSPACE, "df_velocity_loss3d_fwd: workspace too small");
  DF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, DF_EALIGN, "df_velocity_loss3d_fwd: workspace must be 8-byte aligned");
  hipStream_t s = df::as_stream(stream);
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(velocity_loss3d_fwd_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, psi, x, u, part, g);
  hipLaunchKernelGGL(velocity_loss_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)nb, 1.0 / (3.0 * static_cast<double>(g.nvox)),
                     1.0 / (9.0 * static_cast<double>(g.nvox)), l1, jl1);
  return df::launched("df_velocity_loss3d_fwd");
}

int df_velocity_loss3d_bwd(const float* u, const float* x, const float* g_l1, const float* g_jl1, float* gpsi, int64_t B, int64_t Z,
                           int64_t Y, int64_t X, void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  if (int e = check(u, x, B, Z, Y, X, "df_velocity_loss3d_bwd")) return e;
  DF_REQUIRE(gpsi && workspace, DF_EINVAL, "df_velocity_loss3d_bwd: null output / workspace");
  const Geo3 g{B * Z * Y * X, (int)Z, (int)Y, (int)X};
  DF_REQUIRE(workspace_bytes >= g.nvox * 3 * static_cast<int64_t>(sizeof(float)), DF_EWORKSPACE, "df_velocity_loss3d_bwd: workspace too small");
  DF_REQUIRE(df::aligned16(workspace), DF_EALIGN, "df_velocity_loss3d_bwd: workspace must be 16-byte aligned");
  float* du = static_cast<float*>(workspace);
  hipLaunchKernelGGL(velocity_loss3d_bwd_kernel, dim3((unsigned)ceil_div(g.nvox, kThreads)), dim3(kThreads), 0, df::as_stream(stream), u, x,
                     g_l1, g_jl1, 1.f / static_cast<float>(3 * g.nvox), 1.f / static_cast<float>(9 * g.nvox), du, g);
  if (int e = df::launched("df_velocity_loss3d_bwd")) return e;
  return df_jacobian3d_bwd(nullptr, du, gpsi, B, Z, Y, X, stream);      // curl adjoint: du -> dpsi
}

int64_t df_velocity_loss2d_workspace_bytes(int64_t B, int64_t Y, int64_t X) {
  if (B <= 0 || Y <= 0 || X <= 0) return 0;
  const int64_t n = B * Y * X;
  const int64_t fwd = nblocks_fwd(n) * 2 * static_cast<int64_t>(sizeof(double));
  const int64_t bwd = n * 2 * static_cast<int64_t>(sizeof(float));
  return fwd > bwd ? fwd : bwd;
}

int df_velocity_loss2d_fwd(const float* psi, const float* x, float* u, float* l1, float* jl1, int64_t B, int64_t Y, int64_t X,
                           void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  if (int e = check(psi, x, B, 2, Y, X, "df_velocity_loss2d_fwd")) return e;
  DF_REQUIRE(l1 && jl1 && workspace, DF_EINVAL, "df_velocity_loss2d_fwd: null output / workspace");
  const Geo2 g{B * Y * X, (int)Y, (int)X};
  const int64_t nb = nblocks_fwd(g.npix);
  DF_REQUIRE(workspace_bytes >= nb * 2 * static_cast<int64_t>(sizeof(double)), DF_EWORKSPACE, "df_velocity_loss2d_fwd: workspace too small");
  DF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, DF_EALIGN, "df_velocity_loss2d_fwd: workspace must be 8-byte aligned");
  hipStream_t s = df::as_stream(stream);
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(velocity_loss2d_fwd_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, psi, x, u, part, g);
  hipLaunchKernelGGL(velocity_loss_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)nb, 1.0 / (2.0 * static_cast<double>(g.npix)),
                     1.0 / (4.0 * static_cast<double>(g.npix)), l1, jl1);
  return df::launched("df_velocity_loss2d_fwd");
}

int df_velocity_loss2d_bwd(const float* u, const float* x, const float* g_l1, const float* g_jl1, float* gpsi, int64_t B, int64_t Y,
                           int64_t X, void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  if (int e = check(u, x, B, 2, Y, X, "df_velocity_loss2d_bwd")) return e;
  DF_REQUIRE(gpsi && workspace, DF_EINVAL, "df_velocity_loss2d_bwd: null output / workspace");
  const Geo2 g{B * Y * X, (int)Y, (int)X};
  DF_REQUIRE(workspace_bytes >= g.npix * 2 * static_cast<int64_t>(sizeof(float)), DF_EWORKSPACE, "df_velocity_loss2d_bwd: workspace too small");
  float* du = static_cast<float*>(workspace);
  hipLaunchKernelGGL(velocity_loss2d_bwd_kernel, dim3((unsigned)ceil_div(g.npix, kThreads)), dim3(kThreads), 0, df::as_stream(stream), u, x,
                     g_l1, g_jl1, 1.f / static_cast<float>(2 * g.npix), 1.f / static_cast<float>(4 * g.npix), du, g);
  if (int e = df::launched("df_velocity_loss2d_bwd")) return e;
  return df_curl2d_bwd(du, gpsi, B, Y, X, stream);
}

}  // extern "C"
