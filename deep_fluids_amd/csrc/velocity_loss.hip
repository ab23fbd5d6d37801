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
#include "stencil_common.hpp"

namespace {

using df::ceil_div;
using dfst::Dims3;
using dfst::f32x4;
using dfst::kVoxPerBlock;
constexpr int kThreads = dfst::kThreads;

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
// Barrier that orders LDS traffic only.  __syncthreads() is a full fence: its s_waitcnt vmcnt(0) would also wait for every global load
// in flight (the next tile's boxes, requested on purpose a whole tile ahead) and for the acknowledgement of every store of u.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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

// ================================ 3-D fast path (X % 4 == 0): 4 voxels of one row per thread ========================================
// Forward = two launches: u = curl3(psi) by stencil.hip's 16-byte-load kernel (24 B/voxel), then this reduction over (u, x):
// 24 B/voxel read, nothing written -- 48 B/voxel in total (the one-kernel form below is 36 B/voxel but needs psi at ten offsets per
// voxel: L1-bound at 2 TB/s; u comes back from the Infinity Cache / L2 it was just written through).  Loads as in
// jacobian3d_fwd_vec_kernel: own quad + the next record, the quads one row / one slice further (backward on the last row / slice).
__global__ __launch_bounds__(kThreads) void velocity_jl1_3d_vec_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                       double* __restrict__ partial, Dims3 dm) {
  const int tid = threadIdx.x;
  const int64_t v0 = dfst::xcd_block(blockIdx.x, gridDim.x, dm.group) * kVoxPerBlock;
  const int64_t vq = v0 + 4 * static_cast<int64_t>(tid);
  const int64_t sy = dm.X, sz = static_cast<int64_t>(dm.X) * dm.Y;
  double s1 = 0.0, s9 = 0.0;
  if (vq < dm.nvox) {
    const int64_t row = vq / dm.X;
    const int xx = static_cast<int>(vq - row * dm.X);
    const int64_t slab = row / dm.Y;
    const int yy = static_cast<int>(row - slab * dm.Y);
    const int zz = static_cast<int>(slab % dm.Z);
    const bool ly = yy == dm.Y - 1, lz = zz == dm.Z - 1;
    const bool tail = vq + 4 >= dm.nvox;
    auto load = [&](const float* base, int64_t w, float (&o)[16], bool four) {
      const f32x4* p = reinterpret_cast<const f32x4*>(base + w * 3);
      // the x+1 record of the last voxel: address select (no branch between the loads), and a 12-byte load -- a dead fourth
      // register would be reused at once by hipcc, behind a vmcnt(0) that serialises everything after it
      typedef float f32x3 __attribute__((ext_vector_type(3)));
      const f32x4 a0 = p[0], a1 = p[1], a2 = p[2];
      const f32x3 a3 = *reinterpret_cast<const f32x3*>(p + (four ? 3 : 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = a0[e]; o[4 + e] = a1[e]; o[8 + e] = a2[e]; }
      o[12] = a3[0]; o[13] = a3[1]; o[14] = a3[2]; o[15] = 0.f;
    };
    float uo[16], xo[16], uy[16], xy[16], uz[16], xz[16];
    load(u, vq, uo, !tail); load(x, vq, xo, !tail);
    load(u, ly ? vq - sy : vq + sy, uy, false); load(x, ly ? vq - sy : vq + sy, xy, false);
    load(u, lz ? vq - sz : vq + sz, uz, false); load(x, lz ? vq - sz : vq + sz, xz, false);
    __builtin_amdgcn_sched_barrier(0);      // all 22 loads in flight before the first use (hipcc otherwise sinks them into 4 round trips)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool lx = xx + i == dm.X - 1;                      // only possible for i == 3
      const float a1 = (fabsf(uo[i * 3] - xo[i * 3]) + fabsf(uo[i * 3 + 1] - xo[i * 3 + 1])) + fabsf(uo[i * 3 + 2] - xo[i * 3 + 2]);
      float a9 = 0.f;
#pragma unroll
      for (int axis = 0; axis < 3; ++axis) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float fu = uo[i * 3 + c], fx = xo[i * 3 + c];
          float nu, nx;
          bool last;
          if (axis == 0) { last = i == 3 && lx; nu = last ? uo[(i - 1) * 3 + c] : uo[(i + 1) * 3 + c]; nx = last ? xo[(i - 1) * 3 + c] : xo[(i + 1) * 3 + c]; }
          else if (axis == 1) { last = ly; nu = uy[i * 3 + c]; nx = xy[i * 3 + c]; }
          else { last = lz; nu = uz[i * 3 + c]; nx = xz[i * 3 + c]; }
          const float ju = last ? fu - nu : nu - fu;
          const float jx = last ? fx - nx : nx - fx;
          a9 += fabsf(ju - jx);
        }
      }
      s1 += a1; s9 += a9;
    }
  }
  block_sum2(s1, s9);
  if (tid == 0) { partial[2 * blockIdx.x] = s1; partial[2 * blockIdx.x + 1] = s9; }
}

// ================================ 3-D, ONE kernel: persistent LDS-tiled forward (psi, x) -> (u, l1, j_l1) [r3] =========================
// The two-launch fast path above moves 48 B/voxel (u is written, then read back) and its strided 16-byte-per-lane loads are bound by the
// cache-line traffic of the texture path; the one-kernel form further up needs psi at ten offsets per voxel from L1.  Here a persistent
// workgroup walks tiles of TZ x TY rows x X voxels (X = 4 XQ: whole image rows, so every global access is a run of whole rows):
//   1. the (TZ+2) x (TY+2) rows of psi and the (TZ+1) x (TY+1) rows of x a tile needs (forward halos: SURVEY A.1) arrive in REGISTERS as
//      lane-consecutive 16-byte loads -- requested one tile ahead, i.e. in flight during the whole compute phase of the previous tile --
//      and are written to LDS in the same order (contiguous ds_write_b128);
//   2. u = curl3(psi) on the (TZ+1) x (TY+1) rows (tile + the forward neighbours the Jacobian of u needs) into LDS, same arithmetic
//      as stencil.hip (bit-identical u);
//   3. per quad of 4 voxels: |u - x| and |J(u) - J(x)| from LDS, fp32 per thread and tile, fp64 across tiles; u leaves as 16-byte stores.
// HBM sees psi, x and u once each (36 B/voxel: the halo rows are re-reads of rows another workgroup of the same XCD fetched).
// Needs Z % TZ == 0, Y % TY == 0 (the last row / plane of the tensor is then never the first of a tile: its backward difference
// stays inside the tile) and X in {64, 112, 128}.
// [r3] cfg3 shape, B = 16: 50 us (+ 7 us for the final fixed-order reduce launch) against 29 + 40 us for the two launches it replaces.
// Measured alternatives: one workgroup per CU 65 us, 2 x 4-row tiles (three workgroups per CU) 59 us, non-temporal u stores +-0;
// the boxes fed by LDS-DMA into two input buffers (buffer_load ... lds: no staging registers, no ds_write; one 512-thread workgroup per
// CU, loads and stores on different waves so that no wave waits for a store acknowledgement) 58 us -- a tile's compute phases last
// ~1.1 us, less than the ~2.4 us a CU needs to take in the 51 KB of the next tile's boxes (halo rows included: 2.1x the tile's own
// bytes), so a SECOND workgroup's tile in flight matters more than freeing the registers.  What bounds all of them is the fabric
// traffic of the forward halos (psi: 40 rows per 16); marching along z with a ring of planes would fetch every plane once.
constexpr int kTileZ = 2, kTileY = 8;      // default tile: 2 planes x 8 rows x X voxels

struct TileGeo {
  int B, Z, Y;
  int ntz, nty, ntiles;
  unsigned bytes;      // of psi (== of x, of u)
};

template <int XQ, int kTZ, int kTY, bool NT>
__global__ __launch_bounds__(kThreads) void velocity_loss3d_tile_kernel(const float* __restrict__ psi, const float* __restrict__ x,
                                                                        float* __restrict__ u, double* __restrict__ partial, TileGeo g) {
  constexpr int kPsiRows = (kTZ + 2) * (kTY + 2), kURows = (kTZ + 1) * (kTY + 1);
  constexpr int X = 4 * XQ, RF4 = 3 * XQ;                      // voxels / float4 per image row
  // LDS (float4): [psi box: 40 rows, padded to whole 256-thread passes][x box: 27 rows][u box: 27 rows].  The padding makes the psi / x
  // boundary a pass boundary, so a pass loads from ONE tensor (one wave-uniform buffer descriptor).
  constexpr int KPSI = (kPsiRows * RF4 + kThreads - 1) / kThreads, KX = (kURows * RF4 + kThreads - 1) / kThreads, NLD = KPSI + KX;
  extern __shared__ f32x4 smem[];
  f32x4* sPsi = smem;
  f32x4* sX = smem + KPSI * kThreads;
  f32x4* sU = sX + KX * kThreads;
  const int tid = threadIdx.x;
  const int64_t rowf4 = RF4;
  (void)X;

  auto tile_origin = [&](int t, int& b, int& z0, int& y0) {
    const int yt = t % g.nty;
    const int t2 = t / g.nty;
    const int zt = t2 % g.ntz;
    b = t2 / g.ntz; z0 = zt * kTZ; y0 = yt * kTY;
  };
  // Per-thread byte offsets of its pieces relative to the tile's first row -- the same for every tile.  A tile adds its origin; rows
  // past the last row / plane of a batch element alias rows of the next one (never read back: the backward rule), and pieces past the
  // END of the tensor are answered with zeros by the buffer load's range check -- no per-tile clamping arithmetic at all.
  const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(psi), 0, g.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, g.bytes, 0x00020000);
  unsigned koff[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const bool isx = k >= KPSI;
    int i = (isx ? k - KPSI : k) * kThreads + tid;
    const int nrow = isx ? kURows : kPsiRows, ncol = isx ? kTY + 1 : kTY + 2;
    if (i > nrow * RF4 - 1) i = nrow * RF4 - 1;                // (tail threads of a box's last pass re-load its last piece)
    const int row = i / RF4, c4 = i - row * RF4;
    const int dz = row / ncol, dy = row - dz * ncol;
    koff[k] = static_cast<unsigned>((dz * g.Y + dy) * RF4 + c4) * 16u;
  }
  f32x4 pre[NLD];
  auto issue = [&](int t) {
    int b, z0, y0;
    tile_origin(t, b, z0, y0);
    const unsigned base = static_cast<unsigned>(((b * g.Z + z0) * g.Y + y0) * RF4) * 16u;      // wave-uniform
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      // (the range check of a raw buffer covers the VGPR offset only, so the tile origin goes there and not into the scalar offset)
      pre[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k >= KPSI ? xsrd : psrd, koff[k] + base, 0, 0));
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) smem[k * kThreads + tid] = pre[k];      // (a box's padded tail slots receive copies of its last piece)
  };
  // a quad's 12 floats + the record after it (x+1 neighbour of its last voxel; not read at the end of a row)
  auto quad = [&](const f32x4* rowp, int q, float (&o)[16]) {
    const f32x4 a0 = rowp[3 * q], a1 = rowp[3 * q + 1], a2 = rowp[3 * q + 2];
    const f32x4 a3 = rowp[q + 1 < XQ ? 3 * q + 3 : 3 * q + 2];
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a0[e]; o[4 + e] = a1[e]; o[8 + e] = a2[e]; }
    o[12] = a3[0]; o[13] = a3[1]; o[14] = a3[2]; o[15] = 0.f;
  };

  double s1 = 0.0, s9 = 0.0;
  int t = dfst::xcd_block(blockIdx.x, gridDim.x, 0) < g.ntiles ? static_cast<int>(dfst::xcd_block(blockIdx.x, gridDim.x, 0)) : -1;
  // persistent walk: workgroup w (XCD-contiguous numbering) takes tiles w, w + G, ...: at any time the G workgroups cover G consecutive
  // tiles, neighbours in y on the same XCD
  const int G = gridDim.x;
  if (t >= 0) issue(t);
  while (t >= 0) {
    int b, z0, y0;
    tile_origin(t, b, z0, y0);
    commit();
    lds_barrier();
    const int tn = t + G < g.ntiles ? t + G : -1;
    if (tn >= 0) issue(tn);                                    // in flight during both compute phases
    // ---- u = curl3(psi) on the (TZ+1) x (TY+1) rows --------------------------------------------------------------------------------
    for (int it = tid; it < kURows * XQ; it += kThreads) {
      const int r = it / XQ, q = it - r * XQ;
      const int dz = r / (kTY + 1), dy = r - dz * (kTY + 1);
      const int z = z0 + dz, y = y0 + dy;
      if (z < g.Z && y < g.Y) {
        const bool ly = y == g.Y - 1, lz = z == g.Z - 1;
        float po[16], py[16], pz[16];
        quad(sPsi + (dz * (kTY + 2) + dy) * RF4, q, po);
        quad(sPsi + (dz * (kTY + 2) + (ly ? dy - 1 : dy + 1)) * RF4, q, py);
        quad(sPsi + ((lz ? dz - 1 : dz + 1) * (kTY + 2) + dy) * RF4, q, pz);
        float uo[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool lx = q == XQ - 1 && i == 3;
          const float o0 = po[i * 3], o1 = po[i * 3 + 1], o2 = po[i * 3 + 2];
          const float x1 = lx ? po[(i - 1) * 3 + 1] : po[(i + 1) * 3 + 1], x2 = lx ? po[(i - 1) * 3 + 2] : po[(i + 1) * 3 + 2];
          const float dx1 = lx ? o1 - x1 : x1 - o1, dx2 = lx ? o2 - x2 : x2 - o2;
          const float dy0 = ly ? o0 - py[i * 3] : py[i * 3] - o0, dy2 = ly ? o2 - py[i * 3 + 2] : py[i * 3 + 2] - o2;
          const float dz0 = lz ? o0 - pz[i * 3] : pz[i * 3] - o0, dz1 = lz ? o1 - pz[i * 3 + 1] : pz[i * 3 + 1] - o1;
          uo[i * 3] = dy2 - dz1; uo[i * 3 + 1] = dz0 - dx2; uo[i * 3 + 2] = dx1 - dy0;
        }
        f32x4* d = sU + r * RF4 + 3 * q;
        d[0] = f32x4{uo[0], uo[1], uo[2], uo[3]}; d[1] = f32x4{uo[4], uo[5], uo[6], uo[7]}; d[2] = f32x4{uo[8], uo[9], uo[10], uo[11]};
      }
    }
    lds_barrier();
    // ---- losses of the TZ x TY rows + the store of u ---------------------------------------------------------------------------------
    // |J(u) - J(x)| per axis: the backward difference of the last row / plane is the NEGATED forward expression of both terms, and
    // |-(a - b)| == |a - b| bit for bit, so no operand select is needed for y and z; along x the last voxel's difference is the
    // previous voxel's (ops.py:214-217 replicates the DIFFERENCE): its three terms are voxel 2's, counted twice.
    float a1s = 0.f, a9s = 0.f;
    for (int it = tid; it < kTZ * kTY * XQ; it += kThreads) {
      const int r = it / XQ, q = it - r * XQ;
      const int dz = r / kTY, dy = r - dz * kTY;
      const int z = z0 + dz, y = y0 + dy;
      const bool ly = y == g.Y - 1, lz = z == g.Z - 1;
      const int ro = dz * (kTY + 1) + dy, ry = dz * (kTY + 1) + (ly ? dy - 1 : dy + 1), rz = (lz ? dz - 1 : dz + 1) * (kTY + 1) + dy;
      float uo[16], xo[16], uy[16], xy[16], uz[16], xz[16];
      quad(sU + ro * RF4, q, uo); quad(sX + ro * RF4, q, xo);
      quad(sU + ry * RF4, q, uy); quad(sX + ry * RF4, q, xy);
      quad(sU + rz * RF4, q, uz); quad(sX + rz * RF4, q, xz);
      const bool lxq = q == XQ - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a1 = (fabsf(uo[i * 3] - xo[i * 3]) + fabsf(uo[i * 3 + 1] - xo[i * 3 + 1])) + fabsf(uo[i * 3 + 2] - xo[i * 3 + 2]);
        float a9 = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float fu = uo[i * 3 + c], fx = xo[i * 3 + c];
          // x: the difference towards the next record (voxel 3 of a row's last quad: voxel 2's difference again)
          const int ia = i == 3 ? 2 : i;
          const float tx3 = fabsf((uo[(ia + 1) * 3 + c] - uo[ia * 3 + c]) - (xo[(ia + 1) * 3 + c] - xo[ia * 3 + c]));
          const float txn = fabsf((uo[(i + 1) * 3 + c] - fu) - (xo[(i + 1) * 3 + c] - fx));
          a9 += (i == 3 && lxq) ? tx3 : txn;
          a9 += fabsf((uy[i * 3 + c] - fu) - (xy[i * 3 + c] - fx));
          a9 += fabsf((uz[i * 3 + c] - fu) - (xz[i * 3 + c] - fx));
        }
        a1s += a1; a9s += a9;
      }
      f32x4* dst = reinterpret_cast<f32x4*>(u) + ((static_cast<int64_t>(b) * g.Z + z) * g.Y + y) * rowf4 + 3 * q;
      if (NT) {
        __builtin_nontemporal_store(f32x4{uo[0], uo[1], uo[2], uo[3]}, dst); __builtin_nontemporal_store(f32x4{uo[4], uo[5], uo[6], uo[7]}, dst + 1);
        __builtin_nontemporal_store(f32x4{uo[8], uo[9], uo[10], uo[11]}, dst + 2);
      } else {
        dst[0] = f32x4{uo[0], uo[1], uo[2], uo[3]}; dst[1] = f32x4{uo[4], uo[5], uo[6], uo[7]}; dst[2] = f32x4{uo[8], uo[9], uo[10], uo[11]};
      }
    }
    s1 += a1s; s9 += a9s;
    lds_barrier();                                              // everyone is done with the boxes before the next tile overwrites them
    t = tn;
  }
  block_sum2(s1, s9);
  if (tid == 0) { partial[2 * blockIdx.x] = s1; partial[2 * blockIdx.x + 1] = s9; }
}

#ifdef DF_TUNING
// ================================ 3-D, ONE kernel, marching along z [r3; tuning library only] ==========================================
// MEASURED (cfg3, 16x64x96x64): 58.4 us per forward against 57.5 us for the tile kernel above (both incl. the 1-block final reduction), so
// the tiles stay the production path and this kernel is kept under DF_TUNING (df_debug_set_tail(8), tools/tail_probe.py) as the record of
// the experiment: fetching every plane once does not pay because the halo re-reads of the tiles are L2 hits, not HBM traffic.
// The tiles above load (TZ+2)(TY+2) rows of psi per TZ x TY rows of output: 2.5x at 2 x 8.  Here a workgroup owns a COLUMN of TY rows x X
// voxels of one batch element and marches through all Z planes: every plane of psi and x is fetched once (+ the forward halo rows of the
// column: (TY+2)/TY, (TY+1)/TY), by LDS-DMA (`buffer_load_dwordx4 ... lds`: the planes are verbatim runs of image rows) into rings of 8
// planes, SEVEN z-steps ahead of their use -- the memory latency is covered by ring depth, not by occupancy.  6 waves:
//   waves 0-1  issue the DMA of plane s+7 (3 + 3 instructions each), wait for plane s+3 with `s_waitcnt vmcnt(18)` (loads only: these waves
//              never store), and compute u(s+2) = curl3(psi) from psi(s+2), psi(s+3) into a ring of 4 u planes;
//   waves 2-5  take |u - x| and |J(u) - J(x)| of plane s from u(s), u(s+1), x(s), x(s+1) (two threads per quad: x, y terms + l1 | z terms)
//              and store u(s) -- they never wait for a load.
// ONE LDS-only barrier per z-step.  Same arithmetic as the tile kernel (u bit-identical to df_jacobian3d_fwd).  X = 64, Y % TY == 0, Z >= 4.
constexpr int kMarchT = 384, kMarchP = 7;

struct MarchGeo {
  int B, Z, Y, ncol;       // ncol = Y / TY columns per batch element
  unsigned bytes;
};

template <int XQ, int TY>
__global__ __launch_bounds__(kMarchT) void velocity_loss3d_march_kernel(const float* __restrict__ psi, const float* __restrict__ x,
                                                                         float* __restrict__ u, double* __restrict__ partial, MarchGeo g) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int RF4 = 3 * XQ;                       // float4 per image row
  constexpr int PR = TY + 2;                        // rows per ring plane (psi needs TY + 2, x and u TY + 1)
  constexpr int PL = PR * RF4;                      // float4 per ring plane; == 6 DMA instructions of 64 pieces for XQ = 16, TY = 6
  constexpr int NDMA = (PL + 127) / 128;            // DMA instructions per loader wave, plane and tensor
  static_assert(PL % 64 == 0, "plane must be whole 64-piece DMA instructions");
  extern __shared__ f32x4 smem[];                   // [psi ring 8][x ring 8][u ring 4] planes of PL float4
  f32x4* sPsi = smem;
  f32x4* sX = smem + 8 * PL;
  f32x4* sU = smem + 16 * PL;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < 2;
  const int col = blockIdx.x % g.ncol, b = blockIdx.x / g.ncol;
  const int y0 = col * TY;
  const int64_t rowf4 = RF4;

  const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(psi), 0, g.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, g.bytes, 0x00020000);
  // loader lane: piece (wave & 1) * 64 + lane of every 128-piece group of a plane; byte offset within the plane's run of rows
  unsigned koff[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) koff[k] = static_cast<unsigned>(k * 128 + (wave & 1) * 64 + lane) * 16u;
  const unsigned colbase = static_cast<unsigned>((b * g.Z * g.Y + y0) * RF4) * 16u;      // row (b, z = 0, y0)
  const unsigned planeb = static_cast<unsigned>(g.Y * RF4) * 16u;
  auto dma_plane = [&](int z) {      // psi and x rows y0 .. y0 + TY + 1 of plane z -> ring slot z & 7 (rows past the column's needs are never read)
    const unsigned base = colbase + static_cast<unsigned>(z) * planeb;
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      if (k * 128 + (wave & 1) * 64 < PL) {
        lds_ptr dp = (lds_ptr)(sPsi + (z & 7) * PL + k * 128 + (wave & 1) * 64);
        const unsigned vo = koff[k] + base;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(psrd, dp, 16, vo, 0u, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      if (k * 128 + (wave & 1) * 64 < PL) {
        lds_ptr dx = (lds_ptr)(sX + (z & 7) * PL + k * 128 + (wave & 1) * 64);
        const unsigned vo = koff[k] + base;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, dx, 16, vo, 0u, 0, 0);
      }
    }
  };
  auto quad = [&](const f32x4* rowp, int q, float (&o)[16]) {
    const f32x4 a0 = rowp[3 * q], a1 = rowp[3 * q + 1], a2 = rowp[3 * q + 2];
    const f32x4 a3 = rowp[q + 1 < XQ ? 3 * q + 3 : 3 * q + 2];
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a0[e]; o[4 + e] = a1[e]; o[8 + e] = a2[e]; }
    o[12] = a3[0]; o[13] = a3[1]; o[14] = a3[2]; o[15] = 0.f;
  };

  // prologue: planes 0 .. P-3 (step s issues plane s + P; the first step is s = -2)
  if (loader) {
    for (int z = 0; z < kMarchP - 2 && z < g.Z; ++z) dma_plane(z);
  }
  double s1 = 0.0, s9 = 0.0;
  for (int s = -2; s < g.Z; ++s) {
    if (loader) {
      // plane s + 3 (the newest one this step reads) has landed: at most the 3 planes issued after it may still be in flight
      if (s + kMarchP - 1 < g.Z) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * 2 * ((PL / 64 + 1) / 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    if (loader) {
      if (s + kMarchP < g.Z) dma_plane(s + kMarchP);
      // ---- u(s+2) = curl3(psi) on rows 0 .. TY of the column ---------------------------------------------------------------------
      const int zu = s + 2;
      const int it = tid;                                      // 128 loader threads >= (TY + 1) * XQ items
      if (zu < g.Z && it < (TY + 1) * XQ) {
        const int r = it / XQ, q = it - r * XQ;
        const int y = y0 + r;
        if (y < g.Y) {
          const bool ly = y == g.Y - 1, lz = zu == g.Z - 1;
          float po[16], py[16], pz[16];
          quad(sPsi + (zu & 7) * PL + r * RF4, q, po);
          quad(sPsi + (zu & 7) * PL + (ly ? r - 1 : r + 1) * RF4, q, py);
          quad(sPsi + ((lz ? zu - 1 : zu + 1) & 7) * PL + r * RF4, q, pz);
          float uo[12];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool lx = q == XQ - 1 && i == 3;
            const float o0 = po[i * 3], o1 = po[i * 3 + 1], o2 = po[i * 3 + 2];
            const float x1 = lx ? po[(i - 1) * 3 + 1] : po[(i + 1) * 3 + 1], x2 = lx ? po[(i - 1) * 3 + 2] : po[(i + 1) * 3 + 2];
            const float dx1 = lx ? o1 - x1 : x1 - o1, dx2 = lx ? o2 - x2 : x2 - o2;
            const float dy0 = ly ? o0 - py[i * 3] : py[i * 3] - o0, dy2 = ly ? o2 - py[i * 3 + 2] : py[i * 3 + 2] - o2;
            const float dz0 = lz ? o0 - pz[i * 3] : pz[i * 3] - o0, dz1 = lz ? o1 - pz[i * 3 + 1] : pz[i * 3 + 1] - o1;
            uo[i * 3] = dy2 - dz1; uo[i * 3 + 1] = dz0 - dx2; uo[i * 3 + 2] = dx1 - dy0;
          }
          f32x4* d = sU + (zu & 3) * PL + r * RF4 + 3 * q;
          d[0] = f32x4{uo[0], uo[1], uo[2], uo[3]}; d[1] = f32x4{uo[4], uo[5], uo[6], uo[7]}; d[2] = f32x4{uo[8], uo[9], uo[10], uo[11]};
        }
      }
    } else if (s >= 0) {
      // ---- losses of plane s: every quad of the TY rows by two threads ----------------------------------------------------------------
      const int it = tid - 128;                                // 0 .. 255 >= 2 * TY * XQ
      if (it < 2 * TY * XQ) {
        const int role = it / (TY * XQ), iq = it - role * (TY * XQ);
        const int r = iq / XQ, q = iq - r * XQ;
        const int y = y0 + r;
        const bool ly = y == g.Y - 1, lz = s == g.Z - 1;
        const int zn = role == 0 ? s : (lz ? s - 1 : s + 1), rn = role == 0 ? (ly ? r - 1 : r + 1) : r;
        float uo[16], xo[16], un[16], xn[16];
        quad(sU + (s & 3) * PL + r * RF4, q, uo); quad(sX + (s & 7) * PL + r * RF4, q, xo);
        quad(sU + (zn & 3) * PL + rn * RF4, q, un); quad(sX + (zn & 7) * PL + rn * RF4, q, xn);
        const bool lxq = q == XQ - 1;
        float a1 = 0.f, a9 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float fu = uo[i * 3 + c], fx = xo[i * 3 + c];
            a9 += fabsf((un[i * 3 + c] - fu) - (xn[i * 3 + c] - fx));
            if (role == 0) {
              const int ia = i == 3 ? 2 : i;
              const float tx3 = fabsf((uo[(ia + 1) * 3 + c] - uo[ia * 3 + c]) - (xo[(ia + 1) * 3 + c] - xo[ia * 3 + c]));
              const float txn = fabsf((uo[(i + 1) * 3 + c] - fu) - (xo[(i + 1) * 3 + c] - fx));
              a9 += (i == 3 && lxq) ? tx3 : txn;
              a1 += fabsf(fu - fx);
            }
          }
        }
        s1 += a1; s9 += a9;
        if (role == 1) {
          f32x4* dst = reinterpret_cast<f32x4*>(u) + ((static_cast<int64_t>(b) * g.Z + s) * g.Y + y) * rowf4 + 3 * q;
          dst[0] = f32x4{uo[0], uo[1], uo[2], uo[3]}; dst[1] = f32x4{uo[4], uo[5], uo[6], uo[7]}; dst[2] = f32x4{uo[8], uo[9], uo[10], uo[11]};
        }
      }
    }
  }
  // block sums over the 6 waves (the loaders contribute zeros), fixed order
  __syncthreads();
  double* red = reinterpret_cast<double*>(smem);
  s1 = wave_sum(s1); s9 = wave_sum(s9);
  if (lane == 0) { red[wave] = s1; red[8 + wave] = s9; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < kMarchT / 64; ++w) { a += red[w]; c += red[8 + w]; }
    partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = c;
  }
}
#endif  // DF_TUNING

// du of 4 consecutive voxels per thread: the adjoint of D along an axis at position k needs f = (u, x) at k-1, k, k+1 only
// (on the last two positions the replicated difference folds back onto the same three values), i.e. the quads one
// row / slice before and after and the two records left / right of the quad.  Same operation order as velocity_loss3d_bwd_kernel.
template <bool NT>
__global__ __launch_bounds__(kThreads) void velocity_du3d_vec_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                     const float* __restrict__ g_l1, const float* __restrict__ g_jl1,
                                                                     float inv_n1, float inv_nj, float* __restrict__ du, Dims3 dm) {
  __shared__ __attribute__((aligned(16))) float so[kVoxPerBlock * 3];
  const int tid = threadIdx.x;
  const int64_t v0 = dfst::xcd_block(blockIdx.x, gridDim.x, dm.group) * kVoxPerBlock;
  const int64_t vq = v0 + 4 * static_cast<int64_t>(tid);
  const int64_t sy = dm.X, sz = static_cast<int64_t>(dm.X) * dm.Y;
  if (vq < dm.nvox) {
    const float s1 = inv_n1 * (g_l1 ? g_l1[0] : 1.f), s9 = inv_nj * (g_jl1 ? g_jl1[0] : 1.f);
    const int64_t row = vq / dm.X;
    const int xx = static_cast<int>(vq - row * dm.X);
    const int64_t slab = row / dm.Y;
    const int yy = static_cast<int>(row - slab * dm.Y);
    const int zz = static_cast<int>(slab % dm.Z);
    auto load = [&](const float* base, int64_t w, float (&o)[12]) {
      const f32x4* p = reinterpret_cast<const f32x4*>(base + w * 3);
      const f32x4 a0 = p[0], a1 = p[1], a2 = p[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = a0[e]; o[4 + e] = a1[e]; o[8 + e] = a2[e]; }
    };
    // adjoint at position k of an axis of extent n from the three values (fm, f0, fp) of u and x around it; exactly
    // adj_at(line(...)): g(i) = sgn(D u(i) - D x(i)) * s9, gp(n-2) = g(n-2) + g(n-1) with D(n-1) == D(n-2)
    auto adj = [&](float um, float u0, float up, float xm, float x0, float xp, int k, int n) -> float {
      const float gm = sgn((u0 - um) - (x0 - xm)) * s9;      // g(k-1)   (k >= 1)
      const float gk = sgn((up - u0) - (xp - x0)) * s9;      // g(k)     (k <= n-2)
      if (k == n - 1) return gm + gm;                        // gp(n-2)
      const float gpk = k == n - 2 ? gk + gk : gk;
      return k == 0 ? -gpk : gm - gpk;
    };
    // every load is issued up front and unconditionally: at the ends of an axis the missing neighbour is replaced by a valid
    // address (the quad itself) and adj() ignores that side there.  A branch around a group of loads makes hipcc wait for all
    // earlier ones first -- four dependent memory round trips instead of one.
    float uo[12], xo[12], umy[12], xmy[12], upy[12], xpy[12], umz[12], xmz[12], upz[12], xpz[12], out[12];
    float uL[3], xL[3], uR[3], xR[3];
    load(u, vq, uo); load(x, vq, xo);
    {
      const int64_t my = yy > 0 ? vq - sy : vq, py = yy < dm.Y - 1 ? vq + sy : vq;
      const int64_t mz = zz > 0 ? vq - sz : vq, pz = zz < dm.Z - 1 ? vq + sz : vq;
      load(u, my, umy); load(x, my, xmy); load(u, py, upy); load(x, py, xpy);
      load(u, mz, umz); load(x, mz, xmz); load(u, pz, upz); load(x, pz, xpz);
      const int64_t l = xx > 0 ? vq - 1 : vq, r = xx + 4 < dm.X ? vq + 4 : vq + 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) { uL[c] = u[l * 3 + c]; xL[c] = x[l * 3 + c]; uR[c] = u[r * 3 + c]; xR[c] = x[r * 3 + c]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int e = i * 3 + c;
        out[e] = adj(i > 0 ? uo[e - 3] : uL[c], uo[e], i < 3 ? uo[e + 3] : uR[c], i > 0 ? xo[e - 3] : xL[c], xo[e], i < 3 ? xo[e + 3] : xR[c],
                     xx + i, dm.X);
      }
#pragma unroll
    for (int e = 0; e < 12; ++e) out[e] += adj(umy[e], uo[e], upy[e], xmy[e], xo[e], xpy[e], yy, dm.Y);
#pragma unroll
    for (int e = 0; e < 12; ++e) out[e] += adj(umz[e], uo[e], upz[e], xmz[e], xo[e], xpz[e], zz, dm.Z);
#pragma unroll
    for (int e = 0; e < 12; ++e) out[e] = sgn(uo[e] - xo[e]) * s1 + out[e];
    f32x4* q = reinterpret_cast<f32x4*>(so + tid * 12);
    q[0] = f32x4{out[0], out[1], out[2], out[3]};
    q[1] = f32x4{out[4], out[5], out[6], out[7]};
    q[2] = f32x4{out[8], out[9], out[10], out[11]};
  }
  __syncthreads();
  const int64_t left = dm.nvox - v0;
  const int64_t nv = left < kVoxPerBlock ? left : kVoxPerBlock;
  dfst::flush_lds<NT>(so, du + v0 * 3, nv * 3, tid);
}

// ================================ 3-D, one voxel per lane with 12-byte record loads =================================================
// A wave's 64 records form ONE contiguous 768-byte span per load (6 cache lines; the 16-byte-per-lane quads above touch 24 lines per
// load instruction, which bounds them at ~3 TB/s in the texture-address unit).  hipcc merges the struct copies into
// global_load_dwordx3 (own + x-neighbour: dwordx4 + dwordx2).  Any extents >= 2.
struct F3 {
  float a, b, c;
};
__device__ __forceinline__ F3 ld3(const float* __restrict__ p) { return *reinterpret_cast<const F3*>(p); }

__global__ __launch_bounds__(kThreads) void velocity_jl1_3d_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                   double* __restrict__ partial, Geo3 g) {
  const int64_t sy = g.X, sz = static_cast<int64_t>(g.X) * g.Y;
  double s1 = 0.0, s9 = 0.0;
#pragma unroll
  for (int it = 0; it < kVpt3; ++it) {
    const int64_t v = (static_cast<int64_t>(blockIdx.x) * kVpt3 + it) * kThreads + threadIdx.x;
    if (v < g.nvox) {
      int xx, yy, zz;
      coords3(v, g, xx, yy, zz);
      const bool lx = xx == g.X - 1, ly = yy == g.Y - 1, lz = zz == g.Z - 1;
      const int64_t nx = lx ? v - 1 : v + 1, ny = ly ? v - sy : v + sy, nz = lz ? v - sz : v + sz;
      const F3 uc = ld3(u + v * 3), ux = ld3(u + nx * 3), uy = ld3(u + ny * 3), uz = ld3(u + nz * 3);
      const F3 xc = ld3(x + v * 3), xx_ = ld3(x + nx * 3), xy = ld3(x + ny * 3), xz = ld3(x + nz * 3);
      const float a1 = (fabsf(uc.a - xc.a) + fabsf(uc.b - xc.b)) + fabsf(uc.c - xc.c);
      auto d = [](bool last, float f, float n) { return last ? f - n : n - f; };
      float a9 = 0.f;
      a9 += fabsf(d(lx, uc.a, ux.a) - d(lx, xc.a, xx_.a)); a9 += fabsf(d(lx, uc.b, ux.b) - d(lx, xc.b, xx_.b)); a9 += fabsf(d(lx, uc.c, ux.c) - d(lx, xc.c, xx_.c));
      a9 += fabsf(d(ly, uc.a, uy.a) - d(ly, xc.a, xy.a)); a9 += fabsf(d(ly, uc.b, uy.b) - d(ly, xc.b, xy.b)); a9 += fabsf(d(ly, uc.c, uy.c) - d(ly, xc.c, xy.c));
      a9 += fabsf(d(lz, uc.a, uz.a) - d(lz, xc.a, xz.a)); a9 += fabsf(d(lz, uc.b, uz.b) - d(lz, xc.b, xz.b)); a9 += fabsf(d(lz, uc.c, uz.c) - d(lz, xc.c, xz.c));
      s1 += a1; s9 += a9;
    }
  }
  block_sum2(s1, s9);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s1; partial[2 * blockIdx.x + 1] = s9; }
}

__global__ __launch_bounds__(kThreads) void velocity_du3d_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                 const float* __restrict__ g_l1, const float* __restrict__ g_jl1,
                                                                 float inv_n1, float inv_nj, float* __restrict__ du, Geo3 g) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= g.nvox) return;
  const float s1 = inv_n1 * (g_l1 ? g_l1[0] : 1.f), s9 = inv_nj * (g_jl1 ? g_jl1[0] : 1.f);
  int xx, yy, zz;
  coords3(v, g, xx, yy, zz);
  const int64_t sy = g.X, sz = static_cast<int64_t>(g.X) * g.Y;
  // the three values around position k of an axis (clamped at the ends: adj() ignores the clamped side there)
  auto adj = [&](float um, float u0, float up, float xm, float x0, float xp, int k, int n) -> float {
    const float gm = sgn((u0 - um) - (x0 - xm)) * s9;      // g(k-1)   (k >= 1)
    const float gk = sgn((up - u0) - (xp - x0)) * s9;      // g(k)     (k <= n-2)
    if (k == n - 1) return gm + gm;                        // gp(n-2) = g(n-2) + g(n-1), D(n-1) == D(n-2)
    const float gpk = k == n - 2 ? gk + gk : gk;
    return k == 0 ? -gpk : gm - gpk;
  };
  const F3 u0 = ld3(u + v * 3), x0 = ld3(x + v * 3);
  F3 acc;
  {
    const int64_t m = xx > 0 ? v - 1 : v, p = xx < g.X - 1 ? v + 1 : v;
    const F3 um = ld3(u + m * 3), up = ld3(u + p * 3), xm = ld3(x + m * 3), xp = ld3(x + p * 3);
    acc.a = adj(um.a, u0.a, up.a, xm.a, x0.a, xp.a, xx, g.X);
    acc.b = adj(um.b, u0.b, up.b, xm.b, x0.b, xp.b, xx, g.X);
    acc.c = adj(um.c, u0.c, up.c, xm.c, x0.c, xp.c, xx, g.X);
  }
  {
    const int64_t m = yy > 0 ? v - sy : v, p = yy < g.Y - 1 ? v + sy : v;
    const F3 um = ld3(u + m * 3), up = ld3(u + p * 3), xm = ld3(x + m * 3), xp = ld3(x + p * 3);
    acc.a += adj(um.a, u0.a, up.a, xm.a, x0.a, xp.a, yy, g.Y);
    acc.b += adj(um.b, u0.b, up.b, xm.b, x0.b, xp.b, yy, g.Y);
    acc.c += adj(um.c, u0.c, up.c, xm.c, x0.c, xp.c, yy, g.Y);
  }
  {
    const int64_t m = zz > 0 ? v - sz : v, p = zz < g.Z - 1 ? v + sz : v;
    const F3 um = ld3(u + m * 3), up = ld3(u + p * 3), xm = ld3(x + m * 3), xp = ld3(x + p * 3);
    acc.a += adj(um.a, u0.a, up.a, xm.a, x0.a, xp.a, zz, g.Z);
    acc.b += adj(um.b, u0.b, up.b, xm.b, x0.b, xp.b, zz, g.Z);
    acc.c += adj(um.c, u0.c, up.c, xm.c, x0.c, xp.c, zz, g.Z);
  }
  F3 o;
  o.a = sgn(u0.a - x0.a) * s1 + acc.a;
  o.b = sgn(u0.b - x0.b) * s1 + acc.b;
  o.c = sgn(u0.c - x0.c) * s1 + acc.c;
  *reinterpret_cast<F3*>(du + v * 3) = o;
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

#ifdef DF_TUNING      // tuning library only: 0 = default dispatch, 1 = force the 16-byte quad kernels, 2 = force the record-per-lane kernels
int g_tail_variant = 0;
#else
constexpr int g_tail_variant = 0;
#endif

}  // namespace

extern "C" {

#ifdef DF_TUNING
void df_debug_set_tail(int v) { g_tail_variant = v; }
#endif

int df_jacobian3d_fwd(const float* x, float* j, float* c, int64_t B, int64_t Z, int64_t Y, int64_t X, df_stream_t stream);
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
  if (u && g_tail_variant == 2) {
    if (int e = df_jacobian3d_fwd(psi, nullptr, u, B, Z, Y, X, stream)) return e;
    hipLaunchKernelGGL(velocity_jl1_3d_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, u, x, part, g);
    hipLaunchKernelGGL(velocity_loss_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)nb, 1.0 / (3.0 * static_cast<double>(g.nvox)),
                       1.0 / (9.0 * static_cast<double>(g.nvox)), l1, jl1);
    return df::launched("df_velocity_loss3d_fwd");
  }
#ifdef DF_TUNING      // z-marching columns (one workgroup per column of 6 rows: 256 columns at cfg3)
  if (u && g_tail_variant == 8 && X == 64 && Y % 6 == 0 && Z >= 4 && df::aligned16(psi) && df::aligned16(x) &&
      df::aligned16(u) && B * (Y / 6) >= 192 && B * (Y / 6) <= nb && g.nvox * 12 < (1LL << 32)) {
    MarchGeo mg{(int)B, (int)Z, (int)Y, (int)(Y / 6), (unsigned)(g.nvox * 12)};
    const int64_t mgrid = B * (Y / 6);
    const size_t mlds = static_cast<size_t>(20) * (6 + 2) * 3 * 16 * sizeof(f32x4);      // 8 + 8 + 4 ring planes of 8 rows
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&velocity_loss3d_march_kernel<16, 6>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds))
      return df::fail((int)e, "df_velocity_loss3d_fwd: dynamic LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((velocity_loss3d_march_kernel<16, 6>), dim3((unsigned)mgrid), dim3(kMarchT), mlds, s, psi, x, u, part, mg);
    hipLaunchKernelGGL(velocity_loss_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)mgrid, 1.0 / (3.0 * static_cast<double>(g.nvox)),
                       1.0 / (9.0 * static_cast<double>(g.nvox)), l1, jl1);
    return df::launched("df_velocity_loss3d_fwd");
  }
#endif
  // tile shape: 2 x 8 rows (two workgroups per CU at X = 64); (tuning library: df_debug_set_tail 3 = one workgroup per CU, 4 = 2 x 4 rows,
  // 5 = non-temporal u stores)
  const int tzv = kTileZ, tyv = g_tail_variant == 4 ? 4 : kTileY;
  const bool tile_mode = g_tail_variant == 0 || (g_tail_variant >= 3 && g_tail_variant != 8);
  const int64_t ntl = (Z / tzv) * (Y / tyv) * B;
  const int64_t rf4 = 3 * (X / 4), prow = (tzv + 2) * (tyv + 2), urow = (tzv + 1) * (tyv + 1);
  const size_t tile_lds = static_cast<size_t>((ceil_div(prow * rf4, kThreads) + ceil_div(urow * rf4, kThreads)) * kThreads + urow * rf4) * sizeof(f32x4);
  const int64_t lds_cap = df::lds_optin_bytes();          // 160 KiB per CU on an unpartitioned MI355X
  int64_t tile_grid = (g_tail_variant == 3 || tile_lds == 0 ? 1 : static_cast<int64_t>(lds_cap / static_cast<int64_t>(tile_lds))) * df::kCUs;      // workgroups resident at once
  if (tile_grid > 8 * df::kCUs) tile_grid = 8 * df::kCUs;
  if (tile_grid > ntl) tile_grid = ntl;
  if (u && tile_mode && (X == 64 || X == 112 || X == 128) && Z % tzv == 0 && Y % tyv == 0 && df::aligned16(psi) && df::aligned16(x) &&
      df::aligned16(u) && ntl < (1LL << 30) && tile_grid <= nb && tile_grid > 0 && g.nvox * 12 < (1LL << 32) &&
      static_cast<int64_t>(tile_lds) <= lds_cap) {
    // one persistent LDS-tiled kernel: psi, x -> u, l1, j_l1 (36 B/voxel of HBM traffic, nothing read back); grid = the workgroups that
    // are resident at once
    TileGeo tg{(int)B, (int)Z, (int)Y, (int)(Z / tzv), (int)(Y / tyv), (int)ntl, (unsigned)(g.nvox * 12)};
    bool tile_ok = true;      // a refused LDS opt-in (partitioned device) falls through to the two-launch path below instead of failing
#define DF_TILE(XQV, TYV, NTV)                                                                                                             \
  do {                                                                                                                                     \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&velocity_loss3d_tile_kernel<XQV, kTileZ, TYV, NTV>),                            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds) != hipSuccess) {                                    \
      (void)hipGetLastError();                                                                                                             \
      tile_ok = false;                                                                                                                     \
    } else {                                                                                                                               \
      hipLaunchKernelGGL((velocity_loss3d_tile_kernel<XQV, kTileZ, TYV, NTV>), dim3((unsigned)tile_grid), dim3(kThreads), tile_lds, s,     \
                         psi, x, u, part, tg);                                                                                             \
    }                                                                                                                                      \
  } while (0)
#ifdef DF_TUNING
    if (g_tail_variant == 4 && X == 64) DF_TILE(16, 4, false); else if (g_tail_variant == 5 && X == 64) DF_TILE(16, kTileY, true); else
#endif
    if (X == 64) DF_TILE(16, kTileY, false); else if (X == 112) DF_TILE(28, kTileY, false); else DF_TILE(32, kTileY, false);
#undef DF_TILE
    if (tile_ok) {
      hipLaunchKernelGGL(velocity_loss_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)tile_grid, 1.0 / (3.0 * static_cast<double>(g.nvox)),
                         1.0 / (9.0 * static_cast<double>(g.nvox)), l1, jl1);
      return df::launched("df_velocity_loss3d_fwd");
    }
  }
  if (u && X % 4 == 0 && df::aligned16(psi) && df::aligned16(x) && df::aligned16(u)) {
    // fast path: u = curl3(psi) with the 16-byte-load stencil kernel, then one reduction pass over (u, x)
    if (int e = df_jacobian3d_fwd(psi, nullptr, u, B, Z, Y, X, stream)) return e;
    Dims3 dm{g.nvox, (int)Z, (int)Y, (int)X, 0};
    const int64_t nbv = ceil_div(g.nvox, kVoxPerBlock);      // <= nb: the workspace covers it
    if (nbv % (8 * dfst::kXcdGroup) == 0) dm.group = dfst::kXcdGroup;
    hipLaunchKernelGGL(velocity_jl1_3d_vec_kernel, dim3((unsigned)nbv), dim3(kThreads), 0, s, u, x, part, dm);
    hipLaunchKernelGGL(velocity_loss_final_kernel, dim3(1), dim3(kThreads), 0, s, part, (int)nbv, 1.0 / (3.0 * static_cast<double>(g.nvox)),
                       1.0 / (9.0 * static_cast<double>(g.nvox)), l1, jl1);
    return df::launched("df_velocity_loss3d_fwd");
  }
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
  if (g_tail_variant == 2) {
    hipLaunchKernelGGL(velocity_du3d_kernel, dim3((unsigned)ceil_div(g.nvox, kThreads)), dim3(kThreads), 0, df::as_stream(stream), u, x, g_l1,
                       g_jl1, 1.f / static_cast<float>(3 * g.nvox), 1.f / static_cast<float>(9 * g.nvox), du, g);
  } else if (X % 4 == 0 && df::aligned16(u) && df::aligned16(x)) {
    Dims3 dm{g.nvox, (int)Z, (int)Y, (int)X, 0};
    const int64_t nbv = ceil_div(g.nvox, kVoxPerBlock);
    if (nbv % (8 * dfst::kXcdGroup) == 0) dm.group = dfst::kXcdGroup;
    hipLaunchKernelGGL((velocity_du3d_vec_kernel<false>), dim3((unsigned)nbv), dim3(kThreads), 0, df::as_stream(stream), u, x, g_l1, g_jl1,
                       1.f / static_cast<float>(3 * g.nvox), 1.f / static_cast<float>(9 * g.nvox), du, dm);
  } else
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
