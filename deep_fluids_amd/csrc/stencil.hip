// Forward-difference stencils of the Deep Fluids velocity field (reference ops.py:205-290),
// written for gfx950: HBM-bound kernels, 64-wide wavefronts, fully coalesced 16-byte stores.
//
//   D_a f[i] = f[i+1] - f[i]  (i <= n-2),   D_a f[n-1] = D_a f[n-2]      (the DIFFERENCE is replicated)
//
// Data layout: channels-last fp32, x fastest: neighbour strides in voxels are 1 (x), X (y), X*Y (z).
//
// jacobian3d_fwd (the >= 70 %-of-HBM-roofline target, 60 B/voxel algorithmic: 12 in + 36 j + 12 c):
//   one workgroup = 1024 consecutive voxels (4 per thread, lane-consecutive so the three-float
//   voxel records of a wave form one contiguous 768-byte span per load);  the own record and the
//   three forward (or, on a boundary, backward) neighbour records come straight from L1/L2 -- the
//   y/z neighbours were or will be some other workgroup's "own" records, so HBM sees each input byte
//   once;  the 9+3 results per voxel are transposed through LDS (stride-9 / stride-3 dword writes are
//   bank-conflict-free, the read-back is ds_read_b128) so that every global store instruction of a
//   wave writes 1 KiB of contiguous output.
#include "df_common.hpp"
#include "stencil_common.hpp"

namespace {

using df::ceil_div;
using namespace dfst;

// forward difference of one 3-float record along one axis with the replicate-the-difference rule
__device__ __forceinline__ void diff3(const float* __restrict__ x, int64_t v, int64_t stride, bool last,
                                      const float (&own)[3], float (&d)[3]) {
  const int64_t nb = last ? v - stride : v + stride;
  const float* p = x + nb * 3;
  const float n0 = p[0], n1 = p[1], n2 = p[2];
  d[0] = last ? own[0] - n0 : n0 - own[0];
  d[1] = last ? own[1] - n1 : n1 - own[1];
  d[2] = last ? own[2] - n2 : n2 - own[2];
}

template <bool WJ, bool WC>
__global__ __launch_bounds__(kThreads) void jacobian3d_fwd_kernel(const float* __restrict__ x, float* __restrict__ j,
                                                                  float* __restrict__ c, Dims3 dm) {
  __shared__ __attribute__((aligned(16))) float smem[(WJ ? kVoxPerBlock * 9 : 0) + (WC ? kVoxPerBlock * 3 : 0)];
  float* sj = smem;
  float* sc = smem + (WJ ? kVoxPerBlock * 9 : 0);
  const int tid = threadIdx.x;
  const int64_t v0 = static_cast<int64_t>(blockIdx.x) * kVoxPerBlock;
  const int64_t sy = dm.X, sz = static_cast<int64_t>(dm.X) * dm.Y;

#pragma unroll
  for (int i = 0; i < kVoxPerThread; ++i) {
    const int lv = i * kThreads + tid;
    const int64_t v = v0 + lv;
    if (v < dm.nvox) {
      const int64_t row = v / dm.X;
      const int xx = static_cast<int>(v - row * dm.X);
      const int64_t slab = row / dm.Y;
      const int yy = static_cast<int>(row - slab * dm.Y);
      const int zz = static_cast<int>(slab % dm.Z);
      const float* p = x + v * 3;
      const float own[3] = {p[0], p[1], p[2]};
      float dx[3], dy[3], dz[3];
      diff3(x, v, 1, xx == dm.X - 1, own, dx);
      diff3(x, v, sy, yy == dm.Y - 1, own, dy);
      diff3(x, v, sz, zz == dm.Z - 1, own, dz);
      if (WJ) {
        float* o = sj + lv * 9;      // (dudx,dudy,dudz, dvdx,dvdy,dvdz, dwdx,dwdy,dwdz)
        o[0] = dx[0]; o[1] = dy[0]; o[2] = dz[0];
        o[3] = dx[1]; o[4] = dy[1]; o[5] = dz[1];
        o[6] = dx[2]; o[7] = dy[2]; o[8] = dz[2];
      }
      if (WC) {
        float* o = sc + lv * 3;      // (dwdy-dvdz, dudz-dwdx, dvdx-dudy)
        o[0] = dy[2] - dz[1];
        o[1] = dz[0] - dx[2];
        o[2] = dx[1] - dy[0];
      }
    }
  }
  __syncthreads();
  const int64_t left = dm.nvox - v0;
  const int64_t nv = left < kVoxPerBlock ? left : kVoxPerBlock;
  if (WJ) flush_lds<false>(sj, j + v0 * 9, nv * 9, tid);
  if (WC) flush_lds<false>(sc, c + v0 * 3, nv * 3, tid);
}

// ---- fast path (X % 4 == 0): 4 consecutive voxels of one row per thread, 16-byte loads ---------------------------
// A thread owns voxels 4t..4t+3 = 12 consecutive floats = three float4 (a fourth gives the x+1 record of its last
// voxel); the y and z neighbours are the same three float4 one row / one slice further (16-byte aligned because
// 3*X*4 bytes is a multiple of 16).  10 global_load_dwordx4 per 4 voxels instead of 48 dword loads; all but the
// own records are L1/L2 hits.  Results leave through the same LDS transpose -> 1 KiB-contiguous stores.
template <bool WJ, bool WC, bool NT>
__global__ __launch_bounds__(kThreads) void jacobian3d_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ j,
                                                                      float* __restrict__ c, Dims3 dm) {
  __shared__ __attribute__((aligned(16))) float smem[(WJ ? kVoxPerBlock * 9 : 0) + (WC ? kVoxPerBlock * 3 : 0)];
  float* sj = smem;
  float* sc = smem + (WJ ? kVoxPerBlock * 9 : 0);
  const int tid = threadIdx.x;
  const int64_t v0 = xcd_block(blockIdx.x, gridDim.x, dm.group) * kVoxPerBlock;
  const int64_t vq = v0 + 4 * static_cast<int64_t>(tid);      // first voxel of this thread's quad
  const int64_t sy = dm.X, sz = static_cast<int64_t>(dm.X) * dm.Y;
  if (vq < dm.nvox) {
    const int64_t row = vq / dm.X;
    const int xx = static_cast<int>(vq - row * dm.X);          // multiple of 4; the quad never straddles a row
    const int64_t slab = row / dm.Y;
    const int yy = static_cast<int>(row - slab * dm.Y);
    const int zz = static_cast<int>(slab % dm.Z);
    const bool ly = yy == dm.Y - 1, lz = zz == dm.Z - 1;
    const f32x4* p = reinterpret_cast<const f32x4*>(x + vq * 3);
    const f32x4* py = reinterpret_cast<const f32x4*>(x + (ly ? vq - sy : vq + sy) * 3);
    const f32x4* pz = reinterpret_cast<const f32x4*>(x + (lz ? vq - sz : vq + sz) * 3);
    asm volatile("" :: "v"(py), "v"(pz));      // addresses first, then all the loads back to back
    float o[16], ny[12], nz[12], jo[36], co[12];
    const bool tail = vq + 4 >= dm.nvox;                       // very last quad: nothing to read behind it
    // the x+1 record of the quad's last voxel: address select (a branch between the loads makes hipcc wait for the earlier ones) and
    // a 12-byte load (hipcc reuses a dead fourth register at once, behind a vmcnt(0) that serialises every later load)
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    const f32x4 a0 = p[0], a1 = p[1], a2 = p[2];
    const f32x3 a3 = *reinterpret_cast<const f32x3*>(p + (tail ? 2 : 3));
    const f32x4 b0 = py[0], b1 = py[1], b2 = py[2];
    const f32x4 c0 = pz[0], c1 = pz[1], c2 = pz[2];
    // all ten loads in flight before the first use: one empty asm that consumes every loaded register (hipcc otherwise sinks the
    // y / z loads below the first uses of the own quad -- two dependent memory round trips)
    asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(c0), "v"(c1), "v"(c2));
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a0[e]; o[4 + e] = a1[e]; o[8 + e] = a2[e]; o[12 + e] = e < 3 ? a3[e < 3 ? e : 0] : 0.f;
                                  ny[e] = b0[e]; ny[4 + e] = b1[e]; ny[8 + e] = b2[e];
                                  nz[e] = c0[e]; nz[4 + e] = c1[e]; nz[8 + e] = c2[e]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool lx = xx + i == dm.X - 1;                      // only possible for i == 3
      float dx[3], dy[3], dz[3];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const float own = o[i * 3 + cc];
        const float nx = (i == 3 && lx) ? o[(i - 1) * 3 + cc] : o[(i + 1) * 3 + cc];
        dx[cc] = (i == 3 && lx) ? own - nx : nx - own;
        dy[cc] = ly ? own - ny[i * 3 + cc] : ny[i * 3 + cc] - own;
        dz[cc] = lz ? own - nz[i * 3 + cc] : nz[i * 3 + cc] - own;
      }
      if (WJ) {
        float* q = jo + i * 9;
        q[0] = dx[0]; q[1] = dy[0]; q[2] = dz[0];
        q[3] = dx[1]; q[4] = dy[1]; q[5] = dz[1];
        q[6] = dx[2]; q[7] = dy[2]; q[8] = dz[2];
      }
      if (WC) {
        float* q = co + i * 3;
        q[0] = dy[2] - dz[1];
        q[1] = dz[0] - dx[2];
        q[2] = dx[1] - dy[0];
      }
    }
    // 16-byte LDS writes: lane stride 144 B (j) / 48 B (c) -> the 8 lanes of a ds_write_b128 group cover all 32 banks
    if (WJ) {
      f32x4* q = reinterpret_cast<f32x4*>(sj + tid * 36);
#pragma unroll
      for (int k = 0; k < 9; ++k) q[k] = f32x4{jo[4 * k], jo[4 * k + 1], jo[4 * k + 2], jo[4 * k + 3]};
    }
    if (WC) {
      f32x4* q = reinterpret_cast<f32x4*>(sc + tid * 12);
#pragma unroll
      for (int k = 0; k < 3; ++k) q[k] = f32x4{co[4 * k], co[4 * k + 1], co[4 * k + 2], co[4 * k + 3]};
    }
  }
  __syncthreads();
  const int64_t left = dm.nvox - v0;
  const int64_t nv = left < kVoxPerBlock ? left : kVoxPerBlock;
  if (WJ) flush_lds<NT>(sj, j + v0 * 9, nv * 9, tid);
  if (WC) flush_lds<NT>(sc, c + v0 * 3, nv * 3, tid);
}

// ---- adjoint helpers ---------------------------------------------------------------------------
// out[k] of the adjoint of D along an axis of extent n, given a loader g(i) of the incoming gradient:
//   gp[i] = g[i] (i < n-2), gp[n-2] = g[n-2] + g[n-1];  out[0] = -gp[0]; out[k] = gp[k-1]-gp[k]; out[n-1] = gp[n-2]
template <typename G>
__device__ __forceinline__ float adj_at(const G& g, int k, int n) {
  auto gp = [&](int i) { return i == n - 2 ? g(i) + g(i + 1) : g(i); };
  if (k == 0) return -gp(0);
  if (k == n - 1) return gp(n - 2);
  return gp(k - 1) - gp(k);
}

template <bool HJ, bool HC>
__global__ __launch_bounds__(kThreads) void jacobian3d_bwd_kernel(const float* __restrict__ gj,
                                                                  const float* __restrict__ gc,
                                                                  float* __restrict__ gx, Dims3 dm) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= dm.nvox) return;
  const int64_t row = v / dm.X;
  const int xx = static_cast<int>(v - row * dm.X);
  const int64_t slab = row / dm.Y;
  const int yy = static_cast<int>(row - slab * dm.Y);
  const int zz = static_cast<int>(slab % dm.Z);
  const int64_t sy = dm.X, sz = static_cast<int64_t>(dm.X) * dm.Y;
  // G[comp][axis] at voxel w: the gradient w.r.t. D_axis(comp) after folding the curl terms
  //   ux=j0  uy=j1-c2  uz=j2+c1 | vx=j3+c2  vy=j4  vz=j5-c0 | wx=j6-c1  wy=j7+c0  wz=j8
  auto G = [&](int64_t w, int comp, int axis) -> float {
    float r = 0.f;
    if (HJ) r = gj[w * 9 + comp * 3 + axis];
    if (HC) {
      const float* q = gc + w * 3;
      const int e = comp * 3 + axis;
      if (e == 1) r -= q[2];
      if (e == 2) r += q[1];
      if (e == 3) r += q[2];
      if (e == 5) r -= q[0];
      if (e == 6) r -= q[1];
      if (e == 7) r += q[0];
    }
    return r;
  };
  float out[3];
#pragma unroll
  for (int comp = 0; comp < 3; ++comp) {
    const int64_t bx = v - xx, by = v - static_cast<int64_t>(yy) * sy, bz = v - static_cast<int64_t>(zz) * sz;
    float acc = adj_at([&](int i) { return G(bx + i, comp, 0); }, xx, dm.X);
    acc += adj_at([&](int i) { return G(by + static_cast<int64_t>(i) * sy, comp, 1); }, yy, dm.Y);
    acc += adj_at([&](int i) { return G(bz + static_cast<int64_t>(i) * sz, comp, 2); }, zz, dm.Z);
    out[comp] = acc;
  }
  float* o = gx + v * 3;
  o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
}

// ---- fast adjoint path (X % 4 == 0): 4 consecutive voxels of one row per thread, 16-byte loads, LDS-transposed stores --------
// Gather form of the adjoint: voxel k of an axis of extent n needs the incoming gradient at k-1, k and -- only on the
// last two positions, where the replicated difference folds back -- k+1 / k-1.  A thread loads the 9 (+3) float4 of its own quad,
// of the quad one row up (y-1) and one slice up (z-1) (L1/L2 hits: they are other threads' own quads), the single record at x-1,
// and on the two boundary rows / slices the few extra records by dword loads.  HBM sees every gradient byte once
// (48 B/voxel with gj, 24 B/voxel with gc only); the 12 results per thread leave through LDS as 1 KiB-contiguous stores.
template <bool HJ, bool HC, bool NT>
__global__ __launch_bounds__(kThreads) void jacobian3d_bwd_vec_kernel(const float* __restrict__ gj, const float* __restrict__ gc,
                                                                      float* __restrict__ gx, Dims3 dm) {
  __shared__ __attribute__((aligned(16))) float so[kVoxPerBlock * 3];
  const int tid = threadIdx.x;
  const int64_t v0 = xcd_block(blockIdx.x, gridDim.x, dm.group) * kVoxPerBlock;
  const int64_t vq = v0 + 4 * static_cast<int64_t>(tid);
  const int64_t sy = dm.X, sz = static_cast<int64_t>(dm.X) * dm.Y;
  if (vq < dm.nvox) {
    const int64_t row = vq / dm.X;
    const int xx = static_cast<int>(vq - row * dm.X);
    const int64_t slab = row / dm.Y;
    const int yy = static_cast<int>(row - slab * dm.Y);
    const int zz = static_cast<int>(slab % dm.Z);
    // G[e] (e = comp * 3 + axis) of the 4 voxels of the quad starting at voxel w: the gradient w.r.t. D_axis(comp) after folding the
    // curl terms   ux=j0  uy=j1-c2  uz=j2+c1 | vx=j3+c2  vy=j4  vz=j5-c0 | wx=j6-c1  wy=j7+c0  wz=j8
    auto load_quad = [&](int64_t w, float (&g)[4][9]) {
      float jv[36], cv[12];
      if (HJ) {
        const f32x4* p = reinterpret_cast<const f32x4*>(gj + w * 9);
#pragma unroll
        for (int k = 0; k < 9; ++k) { const f32x4 t = p[k]; jv[4 * k] = t[0]; jv[4 * k + 1] = t[1]; jv[4 * k + 2] = t[2]; jv[4 * k + 3] = t[3]; }
      }
      if (HC) {
        const f32x4* p = reinterpret_cast<const f32x4*>(gc + w * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) { const f32x4 t = p[k]; cv[4 * k] = t[0]; cv[4 * k + 1] = t[1]; cv[4 * k + 2] = t[2]; cv[4 * k + 3] = t[3]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 9; ++e) g[i][e] = HJ ? jv[i * 9 + e] : 0.f;
        if (HC) {
          const float c0 = cv[i * 3], c1 = cv[i * 3 + 1], c2 = cv[i * 3 + 2];
          g[i][1] -= c2; g[i][2] += c1; g[i][3] += c2; g[i][5] -= c0; g[i][6] -= c1; g[i][7] += c0;
        }
      }
    };
    // one folded component of one voxel by scalar loads (boundary rows / slices and the x-1 record only)
    auto G1 = [&](int64_t w, int e) -> float {
      float r = HJ ? gj[w * 9 + e] : 0.f;
      if (HC) {
        const float* q = gc + w * 3;
        if (e == 1) r -= q[2];
        if (e == 2) r += q[1];
        if (e == 3) r += q[2];
        if (e == 5) r -= q[0];
        if (e == 6) r -= q[1];
        if (e == 7) r += q[0];
      }
      return r;
    };
    // every load of the common path is issued up front, unconditionally (a branch between two groups of loads makes hipcc wait
    // for the first before it issues the second: three dependent memory round trips instead of one); unused values (k == 0)
    // come from a valid address and are masked below
    float own[4][9], upy[4][9], upz[4][9], gm1[3];
    float out[4][3];
    load_quad(vq, own);
    load_quad(yy > 0 ? vq - sy : vq, upy);
    load_quad(zz > 0 ? vq - sz : vq, upz);
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) gm1[comp] = G1(xx > 0 ? vq - 1 : vq, comp * 3);
    __builtin_amdgcn_sched_barrier(0);
    // ---- x axis: inside the row; the quad never straddles a row and X % 4 == 0, so k = n-2, n-1 sit in the last quad ----
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const int e = comp * 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = xx + i;
        const float gk = own[i][e];
        const float gkm = i > 0 ? own[i - 1][e] : gm1[comp];
        float r;
        if (k == dm.X - 1) r = gkm + gk;                                   // gp(n-2) = g(n-2) + g(n-1)
        else if (k == dm.X - 2) r = (k > 0 ? gkm : 0.f) - (gk + own[i < 3 ? i + 1 : 3][e]);   // gp(k-1) - (g(n-2) + g(n-1))
        else r = (k > 0 ? gkm : 0.f) - gk;
        out[i][comp] = r;
      }
    }
    // ---- y and z axes: the same rule across rows / slices (stride sy / sz) ----
#pragma unroll
    for (int axis = 1; axis < 3; ++axis) {
      const int k = axis == 1 ? yy : zz, n = axis == 1 ? dm.Y : dm.Z;
      const int64_t st = axis == 1 ? sy : sz;
      float far[4][3];                                // g at k + 1, needed next to the far face only (k == n - 2): one batch of loads
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) far[i][comp] = 0.f;
      if (k == n - 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int comp = 0; comp < 3; ++comp) far[i][comp] = G1(vq + i + st, comp * 3 + axis);
      }
#pragma unroll
      for (int comp = 0; comp < 3; ++comp) {
        const int e = comp * 3 + axis;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gk = own[i][e];
          const float gkm = k > 0 ? (axis == 1 ? upy[i][e] : upz[i][e]) : 0.f;
          out[i][comp] += k == n - 1 ? gkm + gk : gkm - (gk + far[i][comp]);
        }
      }
    }
    f32x4* q = reinterpret_cast<f32x4*>(so + tid * 12);
    q[0] = f32x4{out[0][0], out[0][1], out[0][2], out[1][0]};
    q[1] = f32x4{out[1][1], out[1][2], out[2][0], out[2][1]};
    q[2] = f32x4{out[2][2], out[3][0], out[3][1], out[3][2]};
  }
  __syncthreads();
  const int64_t left = dm.nvox - v0;
  const int64_t nv = left < kVoxPerBlock ? left : kVoxPerBlock;
  flush_lds<NT>(so, gx + v0 * 3, nv * 3, tid);
}

// ---- LDS-staged adjoint (X % 4 == 0, X <= 128, ONE incoming gradient) ---------------------------------------------------------
// The register-only kernel above is bound by the vector-memory front end, not by HBM: a lane's quad is 144 (gj) / 48 (gc) bytes, so
// every 16-byte-per-lane load instruction of a wave touches 72 / 24 cache lines for 1 KiB of data, and a thread issues 27 of them.
// Here the workgroup copies the three record spans it needs -- its own 1024 records plus the X records before them (the y-1 row
// of its first row, the x-1 record of its first voxel), and the 1024 records one slice up -- into LDS with lane-consecutive 16-byte
// loads (8 lines per instruction), and the threads pick their quads out of LDS (ds_read_b128 at a 144- / 48-byte lane stride is
// bank-conflict-free).  HBM still sees each byte once (the z-1 / y-1 spans are L2 hits).
constexpr int kLdsMaxX = 128;

template <int R>
__device__ __forceinline__ float fold_rec(const float* rec, int e) {   // G[e] of one record (9 = gj, 3 = gc folded through the curl)
  if (R == 9) return rec[e];
  switch (e) {
    case 1: return -rec[2];
    case 2: return rec[1];
    case 3: return rec[2];
    case 5: return -rec[0];
    case 6: return -rec[1];
    case 7: return rec[0];
    default: return 0.f;
  }
}

template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void jacobian3d_bwd_lds_kernel(const float* __restrict__ g, float* __restrict__ gx, Dims3 dm) {
  constexpr int QB = kVoxPerBlock * R / 4;       // float4 per 1024-record span
  constexpr int PER = QB / kThreads;             // 9 | 3 per thread
  __shared__ __attribute__((aligned(16))) float sA[(kVoxPerBlock + kLdsMaxX) * R];   // records [v0 - X, v0 + 1024)
  // records [v0 - X*Y, v0 - X*Y + 1024), later the output.  With gj only the three d/dz components of each record are kept
  // (12 instead of 36 KB: 53.8 KB in all, THREE workgroups per CU)
  constexpr bool ZF = R == 9;
  __shared__ __attribute__((aligned(16))) float sZ[kVoxPerBlock * 3];
  const int tid = threadIdx.x;
  const int64_t v0 = xcd_block(blockIdx.x, gridDim.x, dm.group) * kVoxPerBlock;
  const int64_t sy = dm.X, sz = static_cast<int64_t>(dm.X) * dm.Y;
  const int P = dm.X;
  {
    const int64_t z_first = (v0 - sz) * R / 4;                        // float4 index of the z-1 span in g (negative: before the tensor)
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4 ta[PER], tz[PER], tp[2];
    const int npre = P * R / 4;                                       // <= 288
    // every load is unconditional with a CLAMPED index (a branch around a load makes hipcc drain vmcnt(0) before the next one):
    // float4 that do not exist (before the tensor, past its end) read a neighbouring valid one and are never used
    const int64_t g_last = dm.nvox * R / 4 - 1;
    const int64_t own_first = v0 * R / 4;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int64_t q = own_first + k * kThreads + tid;
      ta[k] = g4[q < g_last ? q : g_last];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int64_t q = z_first + k * kThreads + tid;
      tz[k] = g4[q > 0 ? q : 0];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t q = own_first - npre + k * kThreads + tid;        // second round: only the first npre - 256 lanes are kept
      tp[k] = g4[q > 0 ? q : 0];
      asm volatile("" : "+v"(tp[k]));                                 // keep the load up here (not sunk into the guarded LDS write)
    }
    f32x4* a4 = reinterpret_cast<f32x4*>(sA);
    f32x4* z4 = reinterpret_cast<f32x4*>(sZ);
#pragma unroll
    for (int k = 0; k < PER; ++k) a4[npre + k * kThreads + tid] = ta[k];
    if (ZF) {
#pragma unroll
      for (int k = 0; k < PER; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned idx = static_cast<unsigned>(k * kThreads + tid) * 4u + j;     // float index inside the span
          const unsigned rec = idx / 9u, c = idx - rec * 9u;
          if (c % 3u == 2u) sZ[rec * 3u + c / 3u] = tz[k][j];
        }
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) z4[k * kThreads + tid] = tz[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = k * kThreads + tid;
      if (q < npre) a4[q] = tp[k];
    }
  }
  __syncthreads();
  const int64_t vq = v0 + 4 * static_cast<int64_t>(tid);
  float out[4][3];
  const bool live = vq < dm.nvox;
  if (live) {
    const int64_t row = vq / dm.X;
    const int xx = static_cast<int>(vq - row * dm.X);
    const int64_t slab = row / dm.Y;
    const int yy = static_cast<int>(row - slab * dm.Y);
    const int zz = static_cast<int>(slab % dm.Z);
    const float* la = sA + (P + 4 * tid) * R;         // own quad
    const float* lym = la - P * R;                    // the quad one row up
    const float* lzm = sZ + 4 * tid * 3;              // the quad one slice up (ZF: its d/dz components only)
    auto load_quad = [&](const float* src, float (&q)[4][9]) {
      float raw[4 * R];
      const f32x4* p = reinterpret_cast<const f32x4*>(src);
#pragma unroll
      for (int k = 0; k < R; ++k) { const f32x4 t = p[k]; raw[4 * k] = t[0]; raw[4 * k + 1] = t[1]; raw[4 * k + 2] = t[2]; raw[4 * k + 3] = t[3]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 9; ++e) q[i][e] = fold_rec<R>(raw + i * R, e);
    };
    float own[4][9], up[4][9];
    load_quad(la, own);
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const int e = comp * 3;
      const float gm1 = fold_rec<R>(la - R, e);         // unused when xx == 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = xx + i;
        const float gk = own[i][e];
        const float gkm = i > 0 ? own[i - 1][e] : gm1;
        float r;
        if (k == dm.X - 1) r = gkm + gk;
        else if (k == dm.X - 2) r = (k > 0 ? gkm : 0.f) - (gk + own[i < 3 ? i + 1 : 3][e]);
        else r = (k > 0 ? gkm : 0.f) - gk;
        out[i][comp] = r;
      }
    }
#pragma unroll
    for (int axis = 1; axis < 3; ++axis) {
      const int k = axis == 1 ? yy : zz, n = axis == 1 ? dm.Y : dm.Z;
      const int64_t st = axis == 1 ? sy : sz;
      // unused (k == 0) values are whatever the span held and are masked below
      if (ZF && axis == 2) {
        const f32x4* p = reinterpret_cast<const f32x4*>(lzm);
        const f32x4 t0 = p[0], t1 = p[1], t2 = p[2];
        const float z[12] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int comp = 0; comp < 3; ++comp) up[i][comp * 3 + 2] = z[i * 3 + comp];
      } else {
        load_quad(axis == 1 ? lym : lzm, up);
      }
      float far[4][3];                                // g at k + 1, needed next to the far face only (k == n - 2): one batch of loads
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) far[i][comp] = 0.f;
      if (k == n - 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int comp = 0; comp < 3; ++comp) far[i][comp] = fold_rec<R>(g + (vq + i + st) * R, comp * 3 + axis);
      }
#pragma unroll
      for (int comp = 0; comp < 3; ++comp) {
        const int e = comp * 3 + axis;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gk = own[i][e];
          const float gkm = k > 0 ? up[i][e] : 0.f;
          out[i][comp] += k == n - 1 ? gkm + gk : gkm - (gk + far[i][comp]);
        }
      }
    }
  }
  __syncthreads();                                     // every quad has left sZ: it becomes the output staging buffer
  if (live) {
    f32x4* q = reinterpret_cast<f32x4*>(sZ + tid * 12);
    q[0] = f32x4{out[0][0], out[0][1], out[0][2], out[1][0]};
    q[1] = f32x4{out[1][1], out[1][2], out[2][0], out[2][1]};
    q[2] = f32x4{out[2][2], out[3][0], out[3][1], out[3][2]};
  }
  __syncthreads();
  const int64_t left = dm.nvox - v0;
  const int64_t nv = left < kVoxPerBlock ? left : kVoxPerBlock;
  flush_lds<NT>(sZ, gx + v0 * 3, nv * 3, tid);
}

__global__ __launch_bounds__(kThreads) void divergence3d_kernel(const float* __restrict__ x, float* __restrict__ d,
                                                                int64_t nout, int Z, int Y, int X) {
  const int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (o >= nout) return;
  const int X1 = X - 1, Y1 = Y - 1, Z1 = Z - 1;
  const int xx = static_cast<int>(o % X1);
  const int64_t r = o / X1;
  const int yy = static_cast<int>(r % Y1);
  const int64_t s = r / Y1;
  const int zz = static_cast<int>(s % Z1);
  const int64_t b = s / Z1;
  const int64_t v = ((b * Z + zz) * Y + yy) * X + xx;
  const int64_t sy = X, sz = static_cast<int64_t>(X) * Y;
  const float dudx = x[(v + 1) * 3 + 0] - x[v * 3 + 0];
  const float dvdy = x[(v + sy) * 3 + 1] - x[v * 3 + 1];
  const float dwdz = x[(v + sz) * 3 + 2] - x[v * 3 + 2];
  d[o] = dudx + dvdy + dwdz;
}

// ---- 2-D ---------------------------------------------------------------------------------------
struct Dims2 {
  int64_t npix;
  int Y, X;
};

__global__ __launch_bounds__(kThreads) void curl2d_fwd_kernel(const float* __restrict__ psi, float2* __restrict__ u,
                                                              Dims2 dm) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= dm.npix) return;
  const int64_t row = v / dm.X;
  const int xx = static_cast<int>(v - row * dm.X);
  const int yy = static_cast<int>(row % dm.Y);
  const float own = psi[v];
  const bool ly = yy == dm.Y - 1, lx = xx == dm.X - 1;
  const float ny = psi[ly ? v - dm.X : v + dm.X];
  const float nx = psi[lx ? v - 1 : v + 1];
  float2 r;
  r.x = ly ? own - ny : ny - own;          //  D_y psi
  r.y = lx ? nx - own : own - nx;          // -D_x psi   (ops.py:268: x[:,:,:-1] - x[:,:,1:])
  u[v] = r;
}

__global__ __launch_bounds__(kThreads) void curl2d_bwd_kernel(const float* __restrict__ gu, float* __restrict__ gpsi,
                                                              Dims2 dm) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= dm.npix) return;
  const int64_t row = v / dm.X;
  const int xx = static_cast<int>(v - row * dm.X);
  const int yy = static_cast<int>(row % dm.Y);
  const int64_t bx = v - xx, by = v - static_cast<int64_t>(yy) * dm.X;
  const float a = adj_at([&](int i) { return gu[(by + static_cast<int64_t>(i) * dm.X) * 2 + 0]; }, yy, dm.Y);
  const float b = adj_at([&](int i) { return gu[(bx + i) * 2 + 1]; }, xx, dm.X);
  gpsi[v] = a - b;
}

template <bool WJ, bool WW>
__global__ __launch_bounds__(kThreads) void jacobian2d_fwd_kernel(const float2* __restrict__ x, float4* __restrict__ j,
                                                                  float* __restrict__ w, Dims2 dm) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= dm.npix) return;
  const int64_t row = v / dm.X;
  const int xx = static_cast<int>(v - row * dm.X);
  const int yy = static_cast<int>(row % dm.Y);
  const float2 own = x[v];
  const bool ly = yy == dm.Y - 1, lx = xx == dm.X - 1;
  const float2 ny = x[ly ? v - dm.X : v + dm.X];
  const float2 nx = x[lx ? v - 1 : v + 1];
  float4 r;
  r.x = lx ? own.x - nx.x : nx.x - own.x;   // dudx
  r.y = ly ? own.x - ny.x : ny.x - own.x;   // dudy
  r.z = lx ? own.y - nx.y : nx.y - own.y;   // dvdx
  r.w = ly ? own.y - ny.y : ny.y - own.y;   // dvdy
  if (WJ) j[v] = r;
  if (WW) w[v] = r.z - r.y;
}

template <bool HJ, bool HW>
__global__ __launch_bounds__(kThreads) void jacobian2d_bwd_kernel(const float* __restrict__ gj,
                                                                  const float* __restrict__ gw,
                                                                  float2* __restrict__ gx, Dims2 dm) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= dm.npix) return;
  const int64_t row = v / dm.X;
  const int xx = static_cast<int>(v - row * dm.X);
  const int yy = static_cast<int>(row % dm.Y);
  const int64_t bx = v - xx, by = v - static_cast<int64_t>(yy) * dm.X;
  // G: ux=j0  uy=j1-w | vx=j2+w  vy=j3
  auto G = [&](int64_t p, int e) -> float {
    float r = 0.f;
    if (HJ) r = gj[p * 4 + e];
    if (HW) {
      if (e == 1) r -= gw[p];
      if (e == 2) r += gw[p];
    }
    return r;
  };
  float2 o;
  o.x = adj_at([&](int i) { return G(bx + i, 0); }, xx, dm.X) +
        adj_at([&](int i) { return G(by + static_cast<int64_t>(i) * dm.X, 1); }, yy, dm.Y);
  o.y = adj_at([&](int i) { return G(bx + i, 2); }, xx, dm.X) +
        adj_at([&](int i) { return G(by + static_cast<int64_t>(i) * dm.X, 3); }, yy, dm.Y);
  gx[v] = o;
}

__global__ __launch_bounds__(kThreads) void divergence2d_kernel(const float* __restrict__ x, float* __restrict__ d,
                                                                int64_t nout, int Y, int X) {
  const int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (o >= nout) return;
  const int X1 = X - 1, Y1 = Y - 1;
  const int xx = static_cast<int>(o % X1);
  const int64_t r = o / X1;
  const int yy = static_cast<int>(r % Y1);
  const int64_t b = r / Y1;
  const int64_t v = (b * Y + yy) * X + xx;
  d[o] = (x[(v + 1) * 2 + 0] - x[v * 2 + 0]) + (x[(v + X) * 2 + 1] - x[v * 2 + 1]);
}

#ifdef DF_TUNING      // knobs of the tuning library only (include/deepfluids_hip_debug.h); constants in the release library
int g_stencil_group = 48;
int g_stencil_nt = 1;
int g_stencil_lds = 1;
#else
constexpr int g_stencil_group = dfst::kXcdGroup;
constexpr int g_stencil_nt = 1;      // non-temporal output stores
constexpr int g_stencil_lds = 1;     // LDS-staged adjoints where they apply
#endif

int check3(const void* in, int64_t B, int64_t Z, int64_t Y, int64_t X, const char* fn) {
  DF_REQUIRE(in != nullptr, DF_EINVAL, "%s: null input", fn);
  DF_REQUIRE(B > 0 && Z > 0 && Y > 0 && X > 0, DF_EINVAL, "%s: non-positive extent", fn);
  DF_REQUIRE(Z >= 2 && Y >= 2 && X >= 2, DF_ESHAPE, "%s: forward difference needs every extent >= 2 (got %lld,%lld,%lld)",
             fn, (long long)Z, (long long)Y, (long long)X);
  DF_REQUIRE(Z < (1 << 30) && Y < (1 << 30) && X < (1 << 30), DF_ESHAPE, "%s: extent too large", fn);
  return DF_OK;
}
int check2(const void* in, int64_t B, int64_t Y, int64_t X, const char* fn) {
  DF_REQUIRE(in != nullptr, DF_EINVAL, "%s: null input", fn);
  DF_REQUIRE(B > 0 && Y > 0 && X > 0, DF_EINVAL, "%s: non-positive extent", fn);
  DF_REQUIRE(Y >= 2 && X >= 2, DF_ESHAPE, "%s: forward difference needs every extent >= 2 (got %lld,%lld)", fn,
             (long long)Y, (long long)X);
  DF_REQUIRE(Y < (1 << 30) && X < (1 << 30), DF_ESHAPE, "%s: extent too large", fn);
  return DF_OK;
}

}  // namespace

extern "C" {

#ifdef DF_TUNING
void df_debug_set_stencil_nt(int v) { g_stencil_nt = v; }
void df_debug_set_stencil_group(int v) { g_stencil_group = v; }
void df_debug_set_stencil_lds(int v) { g_stencil_lds = v; }
#endif

int df_jacobian3d_fwd(const float* x, float* j, float* c, int64_t B, int64_t Z, int64_t Y, int64_t X,
                      df_stream_t stream) {
  if (int e = check3(x, B, Z, Y, X, "df_jacobian3d_fwd")) return e;
  DF_REQUIRE(j || c, DF_EINVAL, "df_jacobian3d_fwd: both outputs null");
  DF_REQUIRE(df::aligned16(j) && df::aligned16(c), DF_EALIGN, "df_jacobian3d_fwd: outputs must be 16-byte aligned");
  Dims3 dm{B * Z * Y * X, (int)Z, (int)Y, (int)X, 0};
  dim3 grid((unsigned)ceil_div(dm.nvox, kVoxPerBlock)), block(kThreads);
  if (g_stencil_group > 0 && grid.x % (8 * g_stencil_group) == 0) dm.group = g_stencil_group;
  hipStream_t s = df::as_stream(stream);
  if (X % 4 == 0 && df::aligned16(x)) {
    const int nt = g_stencil_nt;
#define DF_J3(WJ, WC)                                                                                              \
  if (nt) hipLaunchKernelGGL((jacobian3d_fwd_vec_kernel<WJ, WC, true>), grid, block, 0, s, x, j, c, dm);           \
  else hipLaunchKernelGGL((jacobian3d_fwd_vec_kernel<WJ, WC, false>), grid, block, 0, s, x, j, c, dm)
    if (j && c) { DF_J3(true, true); } else if (j) { DF_J3(true, false); } else { DF_J3(false, true); }
#undef DF_J3
    return df::launched("df_jacobian3d_fwd");
  }
  if (j && c) hipLaunchKernelGGL((jacobian3d_fwd_kernel<true, true>), grid, block, 0, s, x, j, c, dm);
  else if (j) hipLaunchKernelGGL((jacobian3d_fwd_kernel<true, false>), grid, block, 0, s, x, j, c, dm);
  else hipLaunchKernelGGL((jacobian3d_fwd_kernel<false, true>), grid, block, 0, s, x, j, c, dm);
  return df::launched("df_jacobian3d_fwd");
}

int df_jacobian3d_bwd(const float* gj, const float* gc, float* gx, int64_t B, int64_t Z, int64_t Y, int64_t X,
                      df_stream_t stream) {
  if (int e = check3(gx, B, Z, Y, X, "df_jacobian3d_bwd")) return e;
  DF_REQUIRE(gj || gc, DF_EINVAL, "df_jacobian3d_bwd: both incoming gradients null");
  Dims3 dm{B * Z * Y * X, (int)Z, (int)Y, (int)X, 0};
  hipStream_t s = df::as_stream(stream);
  if (X % 4 == 0 && df::aligned16(gj) && df::aligned16(gc) && df::aligned16(gx)) {
    dim3 gridv((unsigned)ceil_div(dm.nvox, kVoxPerBlock)), blockv(kThreads);
    if (g_stencil_group > 0 && gridv.x % (8 * g_stencil_group) == 0) dm.group = g_stencil_group;
    const bool lds = X <= kLdsMaxX && g_stencil_lds;
    if (gj && gc) hipLaunchKernelGGL((jacobian3d_bwd_vec_kernel<true, true, true>), gridv, blockv, 0, s, gj, gc, gx, dm);
    else if (gj && lds) hipLaunchKernelGGL((jacobian3d_bwd_lds_kernel<9, true>), gridv, blockv, 0, s, gj, gx, dm);
    else if (gc && lds) hipLaunchKernelGGL((jacobian3d_bwd_lds_kernel<3, true>), gridv, blockv, 0, s, gc, gx, dm);
    else if (gj) hipLaunchKernelGGL((jacobian3d_bwd_vec_kernel<true, false, true>), gridv, blockv, 0, s, gj, gc, gx, dm);
    else hipLaunchKernelGGL((jacobian3d_bwd_vec_kernel<false, true, true>), gridv, blockv, 0, s, gj, gc, gx, dm);
    return df::launched("df_jacobian3d_bwd");
  }
  dim3 grid((unsigned)ceil_div(dm.nvox, kThreads)), block(kThreads);
  if (gj && gc) hipLaunchKernelGGL((jacobian3d_bwd_kernel<true, true>), grid, block, 0, s, gj, gc, gx, dm);
  else if (gj) hipLaunchKernelGGL((jacobian3d_bwd_kernel<true, false>), grid, block, 0, s, gj, gc, gx, dm);
  else hipLaunchKernelGGL((jacobian3d_bwd_kernel<false, true>), grid, block, 0, s, gj, gc, gx, dm);
  return df::launched("df_jacobian3d_bwd");
}

int df_divergence3d(const float* x, float* d, int64_t B, int64_t Z, int64_t Y, int64_t X, df_stream_t stream) {
  if (int e = check3(x, B, Z, Y, X, "df_divergence3d")) return e;
  DF_REQUIRE(d != nullptr, DF_EINVAL, "df_divergence3d: null output");
  const int64_t nout = B * (Z - 1) * (Y - 1) * (X - 1);
  hipLaunchKernelGGL(divergence3d_kernel, dim3((unsigned)ceil_div(nout, kThreads)), dim3(kThreads), 0,
                     df::as_stream(stream), x, d, nout, (int)Z, (int)Y, (int)X);
  return df::launched("df_divergence3d");
}

int df_curl2d_fwd(const float* psi, float* u, int64_t B, int64_t Y, int64_t X, df_stream_t stream) {
  if (int e = check2(psi, B, Y, X, "df_curl2d_fwd")) return e;
  DF_REQUIRE(u != nullptr, DF_EINVAL, "df_curl2d_fwd: null output");
  Dims2 dm{B * Y * X, (int)Y, (int)X};
  hipLaunchKernelGGL(curl2d_fwd_kernel, dim3((unsigned)ceil_div(dm.npix, kThreads)), dim3(kThreads), 0,
                     df::as_stream(stream), psi, reinterpret_cast<float2*>(u), dm);
  return df::launched("df_curl2d_fwd");
}

int df_curl2d_bwd(const float* gu, float* gpsi, int64_t B, int64_t Y, int64_t X, df_stream_t stream) {
  if (int e = check2(gu, B, Y, X, "df_curl2d_bwd")) return e;
  DF_REQUIRE(gpsi != nullptr, DF_EINVAL, "df_curl2d_bwd: null output");
  Dims2 dm{B * Y * X, (int)Y, (int)X};
  hipLaunchKernelGGL(curl2d_bwd_kernel, dim3((unsigned)ceil_div(dm.npix, kThreads)), dim3(kThreads), 0,
                     df::as_stream(stream), gu, gpsi, dm);
  return df::launched("df_curl2d_bwd");
}

int df_jacobian2d_fwd(const float* x, float* j, float* w, int64_t B, int64_t Y, int64_t X, df_stream_t stream) {
  if (int e = check2(x, B, Y, X, "df_jacobian2d_fwd")) return e;
  DF_REQUIRE(j || w, DF_EINVAL, "df_jacobian2d_fwd: both outputs null");
  DF_REQUIRE(df::aligned16(j), DF_EALIGN, "df_jacobian2d_fwd: j must be 16-byte aligned");
  Dims2 dm{B * Y * X, (int)Y, (int)X};
  dim3 grid((unsigned)ceil_div(dm.npix, kThreads)), block(kThreads);
  hipStream_t s = df::as_stream(stream);
  const float2* x2 = reinterpret_cast<const float2*>(x);
  float4* j4 = reinterpret_cast<float4*>(j);
  if (j && w) hipLaunchKernelGGL((jacobian2d_fwd_kernel<true, true>), grid, block, 0, s, x2, j4, w, dm);
  else if (j) hipLaunchKernelGGL((jacobian2d_fwd_kernel<true, false>), grid, block, 0, s, x2, j4, w, dm);
  else hipLaunchKernelGGL((jacobian2d_fwd_kernel<false, true>), grid, block, 0, s, x2, j4, w, dm);
  return df::launched("df_jacobian2d_fwd");
}

int df_jacobian2d_bwd(const float* gj, const float* gw, float* gx, int64_t B, int64_t Y, int64_t X,
                      df_stream_t stream) {
  if (int e = check2(gx, B, Y, X, "df_jacobian2d_bwd")) return e;
  DF_REQUIRE(gj || gw, DF_EINVAL, "df_jacobian2d_bwd: both incoming gradients null");
  Dims2 dm{B * Y * X, (int)Y, (int)X};
  dim3 grid((unsigned)ceil_div(dm.npix, kThreads)), block(kThreads);
  hipStream_t s = df::as_stream(stream);
  float2* g2 = reinterpret_cast<float2*>(gx);
  if (gj && gw) hipLaunchKernelGGL((jacobian2d_bwd_kernel<true, true>), grid, block, 0, s, gj, gw, g2, dm);
  else if (gj) hipLaunchKernelGGL((jacobian2d_bwd_kernel<true, false>), grid, block, 0, s, gj, gw, g2, dm);
  else hipLaunchKernelGGL((jacobian2d_bwd_kernel<false, true>), grid, block, 0, s, gj, gw, g2, dm);
  return df::launched("df_jacobian2d_bwd");
}

int df_divergence2d(const float* x, float* d, int64_t B, int64_t Y, int64_t X, df_stream_t stream) {
  if (int e = check2(x, B, Y, X, "df_divergence2d")) return e;
  DF_REQUIRE(d != nullptr, DF_EINVAL, "df_divergence2d: null output");
  const int64_t nout = B * (Y - 1) * (X - 1);
  hipLaunchKernelGGL(divergence2d_kernel, dim3((unsigned)ceil_div(nout, kThreads)), dim3(kThreads), 0,
                     df::as_stream(stream), x, d, nout, (int)Y, (int)X);
  return df::launched("df_divergence2d");
}

}  // extern "C"
