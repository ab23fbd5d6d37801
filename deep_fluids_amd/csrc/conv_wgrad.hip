// Weight gradient of the 3x3 / 3x3x3 SAME convolution on the gfx950 matrix cores (fp32-exact MFMA).
// (reference: TF autodiff of slim.conv2d/conv3d, trainer.py:184 `minimize(g_loss, var_list=G_var)`.)
//
//   gW[tap][ci][co] = sum_voxels X[voxel + off(tap)][ci] * G[voxel][co]        (X zero outside the image)
//
// GEMM view: M = ci, N = co, K = voxels (millions) -> v_mfma_f32_32x32x2_f32 with k = 2 voxels per
// instruction.  Both operands are "K-major rows" in channels-last memory (32 lanes x 8 bytes = one
// contiguous 256-byte row segment per half-wave), so they go STRAIGHT from L1/L2 to the MFMA operand
// registers: no LDS, no barriers, waves run free.
//   * a wave owns a 64(ci) x 64(co) quadrant for the three dx taps of one (dz,dy): 12 accumulator
//     tiles = 192 AGPRs; lane (half, r) holds ci = ci0+2r+{0,1} and co = co0+2r+{0,1} (float2 loads);
//   * the two half-waves walk two different image rows along x; X positions x-1, x, x+1 come from an
//     8-deep register ring (one new 8-byte load per step), G from a second ring; loads run 5-6 steps
//     (~4000 cycles) ahead of their use, and the stream is continuous across row pairs;
//   * per step and wave: 2 global_load_dwordx2 for 12 MFMAs (768 cycles);
//   * workgroup = the four quadrants of 128x128; grid = (voxel ranges x (dz,dy)) x ci-blocks x co-blocks;
//     each workgroup writes its partial [3][128][128] to the workspace, a second kernel reduces the
//     ranges in a fixed order (deterministic), and also finishes the bias gradient sum_voxels G.
#include "df_common.hpp"

namespace {

using df::ceil_div;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kMaxRanges = 256;
constexpr int kZeroBytes = 512;      // zeroed slot at the end of the workspace (padding reads; small-N kernel)
// zeroed ROW at the end of the workspace of the MFMA kernel: out-of-image rows / padded channel lanes read it with the
// same per-position offsets as a real row (2x for the strided gradient rows of the up-sampling-aware variant)
inline int64_t zero_row_bytes(int64_t W, int64_t Cin, int64_t Cout) {
  const int64_t Wp = ceil_div(W, 8) * 8 + 8;
  return Wp * (Cin > 2 * Cout ? Cin : 2 * Cout) * 4 + 64;
}
constexpr int kSmallStreams = 2048;  // wave streams of the small-N kernel

#ifdef DF_TUNING
int g_wgrad_dbg = 0;      // df_debug_set_wgrad (include/deepfluids_hip_debug.h)
#endif

struct WgradArgs {
  const float* x;
  const float* g;
  float* partial;      // [nranges][taps][Cinp][Coutp]
  float* bpartial;     // [nranges][Coutp]
  const float* zeros;  // 64 zero bytes (in the workspace, cleared by a memset node ahead of the launch)
  int B, D, H, W, Cin, Cout;
  int Cinp, Coutp;     // padded to 128
  int Wp;              // W rounded up to 8
  int nrows, npairs, nranges, pairs_per_range;
  int ndzdy;           // 9 (3-D) or 3 (2-D); up mode: 32 | 8 combos = parity class x (dz,dy) pair
  int want_bias;
  int nqi, nqj, nsub;  // live 64x64 quadrants per 128x128 block (1|2 each); the 4/(nqi*nqj) spare waves split the voxel range
  int up;              // 1: x is the COARSE input of an up-sampling-aware conv, g the FINE gradient (see df_upconv_wgrad)
  int gD, gH, gW;      // physical extents of g (== D,H,W unless up)
};

// WP8 > 0: the padded row length Wp = 8*WP8 is a compile-time constant and the whole row is unrolled, so the
// compiler's conservative vmcnt merge at a loop header happens once per row pair instead of every 8 steps.
template <bool XVEC, bool GVEC, int WP8>
__global__ __launch_bounds__(kThreads, 1) void wgrad_kernel(const WgradArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = a.nqi * a.nqj;
  const int qi = (wave % nq) / a.nqj, qj = (wave % nq) % a.nqj, sub = wave / nq;
  const int Wc = WP8 > 0 ? WP8 * 8 : a.W;      // row length (compile-time when WP8 > 0: dispatch guarantees W == 8*WP8)
  const int half = lane >> 5, r = lane & 31;

  // workgroup -> (range, dzdy): XCD x gets a contiguous run of ranges, and the 9 (dz,dy) workgroups of one
  // range are dispatched back-to-back on the same XCD so they share its L2 (speed only).
  const int nwg = a.nranges * a.ndzdy;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int range = wg / a.ndzdy, dzdy = wg % a.ndzdy;
  int dz, dy, pz = 0, py = 0, px = 0, dd = 0, cls = 0;
  if (a.up) {            // combo = class * nd2 + (deltaz, deltay);  X row offset = delta + p - 1 per axis
    const int nd2 = a.ndzdy == 32 ? 4 : 2;
    cls = dzdy / nd2; dd = dzdy % nd2;
    pz = a.ndzdy == 32 ? (cls >> 2) & 1 : 0; py = (cls >> 1) & 1; px = cls & 1;
    dz = a.ndzdy == 32 ? (dd >> 1) + pz - 1 : 0;
    dy = (dd & 1) + py - 1;
  } else {
    dz = a.ndzdy == 9 ? dzdy / 3 - 1 : 0;
    dy = (a.ndzdy == 9 ? dzdy % 3 : dzdy) - 1;
  }
  const int gs = a.up ? 2 : 1;
  const int ci0 = blockIdx.y * 128 + qi * 64, co0 = blockIdx.z * 128 + qj * 64;
  if (ci0 >= a.Cin || co0 >= a.Cout) return;   // wave-uniform: quadrant entirely in the padding

  const int ppe = (a.pairs_per_range + a.nsub - 1) / a.nsub;       // pairs per (range, sub-range)
  const int p0 = range * a.pairs_per_range + sub * ppe;
  int p1 = p0 + ppe;
  if (p1 > (range + 1) * a.pairs_per_range) p1 = (range + 1) * a.pairs_per_range;
  if (p1 > a.npairs) p1 = a.npairs;
  const int erange = range * a.nsub + sub;

  const int cia = ci0 + 2 * r, coa = co0 + 2 * r;   // this lane's first ci / co
  const bool ci_ok0 = cia < a.Cin, ci_ok1 = cia + 1 < a.Cin;
  const bool co_ok0 = coa < a.Cout, co_ok1 = coa + 1 < a.Cout;

  // ---- row cursors.  EVERY load in the main loop is unconditional and its result is used as-is: lanes /
  // positions / rows that must contribute zero (SAME padding, channel padding, range tail) read a zeroed
  // 16-byte slot of the workspace instead (address select BEFORE the load).  A load inside a branch, or a
  // select on the loaded value at load time, would make hipcc wait vmcnt(0) right there and serialise the ring.
  const float* zb = a.zeros;
  struct Row { const float* xb; const float* gb; bool xv; bool gv; };
  auto row_setup = [&](int pair) -> Row {
    Row rw;
    const int row = 2 * pair + half;
    const bool ok = pair < p1 && row < a.nrows;
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
    const int zs = z + dz, ys = y + dy;
    rw.gv = ok;
    rw.xv = ok && zs >= 0 && zs < a.D && ys >= 0 && ys < a.H;
    const int64_t gvox = ((static_cast<int64_t>(b) * a.gD + (z * gs + pz)) * a.gH + (y * gs + py)) * a.gW + px;
    const int64_t xvox = ((static_cast<int64_t>(b) * a.D + zs) * a.H + ys) * a.W;
    rw.gb = a.g + gvox * a.Cout + coa;
    rw.xb = a.x + xvox * a.Cin + cia;
    if (WP8 > 0) {       // exact-row variants: an invalid row / padded channel lane points at a zeroed row-sized region,
      if (!(rw.gv && co_ok0)) rw.gb = zb;      // so the per-position loads need no select at all (W == Wp here)
      if (!(rw.xv && ci_ok0)) rw.xb = zb;
    }
    return rw;
  };
  // the empty asm makes the selected address opaque, so the compiler cannot turn `*(c ? p : z)` back into
  // `c ? *p : *z` (two loads behind a divergent branch)
  // (the select is done on an opaque OFFSET from the kernel-argument pointer so the access stays a global_load)
  auto pick = [&](bool c, const float* p) -> const float* {
    int64_t off = c ? p - a.x : zb - a.x;
    asm("" : "+v"(off));
    return a.x + off;
  };
  auto load_x = [&](const Row& rw, int pos) -> f32x2 {
    if (WP8 > 0) return *reinterpret_cast<const f32x2*>(rw.xb + static_cast<int64_t>(pos) * a.Cin);
    const bool k = rw.xv && pos < Wc;
    const float* p = rw.xb + static_cast<int64_t>(pos) * a.Cin;
    f32x2 v;
    if (XVEC) v = *reinterpret_cast<const f32x2*>(pick(k && ci_ok0, p));
    else { v[0] = *pick(k && ci_ok0, p); v[1] = *pick(k && ci_ok1, p + 1); }
    return v;
  };
  auto load_g = [&](const Row& rw, int pos) -> f32x2 {
    if (WP8 > 0) return *reinterpret_cast<const f32x2*>(rw.gb + static_cast<int64_t>(pos) * gs * a.Cout);
    const bool k = rw.gv && pos < Wc;
    const float* p = rw.gb + static_cast<int64_t>(pos) * gs * a.Cout;
    f32x2 v;
    if (GVEC) v = *reinterpret_cast<const f32x2*>(pick(k && co_ok0, p));
    else { v[0] = *pick(k && co_ok0, p); v[1] = *pick(k && co_ok1, p + 1); }
    return v;
  };

  f32x16 acc[3][2][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][s][t][e] = 0.f;
  f32x2 bsum = {0.f, 0.f};
  const bool do_bias = a.want_bias && (a.up ? dd == 0 : dzdy == a.ndzdy / 2) && blockIdx.y == 0 && qi == 0;

  // rings indexed by (position in row) % 8 (Wp % 8 == 0, so the index is continuous across rows);
  // X holds positions x-1 .. x+6, G holds x .. x+5 relative to the compute cursor x
  f32x2 xr[8], gr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { xr[i] = f32x2{0.f, 0.f}; gr[i] = f32x2{0.f, 0.f}; }
  Row cur = row_setup(p0);
#pragma unroll
  for (int i = 0; i < 6; ++i) xr[i] = load_x(cur, i);
#pragma unroll
  for (int i = 0; i < 5; ++i) gr[i] = load_g(cur, i);

  auto step = [&](int u, int x) {
    __builtin_amdgcn_sched_barrier(0);           // keep this step's two prefetch loads ahead of its MFMAs
    f32x2 am = xr[(u + 7) & 7], a0 = xr[u], ap = xr[(u + 1) & 7];
    const f32x2 b = gr[u];
    if (x == 0) am = f32x2{0.f, 0.f};            // left zero padding (the ring slot holds the previous row's tail)
    if (x == Wc - 1) ap = f32x2{0.f, 0.f};      // right zero padding (the ring slot may hold the next row's head)
    bsum[0] += b[0]; bsum[1] += b[1];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[s], b[t], acc[0][s][t], 0, 0, 0);
        acc[1][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b[t], acc[1][s][t], 0, 0, 0);
        acc[2][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s], b[t], acc[2][s][t], 0, 0, 0);
      }
  };

  const int Wp = WP8 > 0 ? WP8 * 8 : a.Wp;
  for (int pair = p0; pair < p1; ++pair) {
    const Row nxt = row_setup(pair + 1);
    if (WP8 > 0) {
#pragma unroll
      for (int x0 = 0; x0 < (WP8 - 1) * 8; x0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xr[(u + 6) & 7] = load_x(cur, x0 + u + 6);
          gr[(u + 5) & 7] = load_g(cur, x0 + u + 5);
          step(u, x0 + u);
        }
      }
    } else {
      for (int x0 = 0; x0 < Wp - 8; x0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xr[(u + 6) & 7] = load_x(cur, x0 + u + 6);
          gr[(u + 5) & 7] = load_g(cur, x0 + u + 5);
          step(u, x0 + u);
        }
      }
    }
    {   // last 8 positions of the row: the prefetch cursor crosses into the next row pair
      const int x0 = Wp - 8;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xr[(u + 6) & 7] = u < 2 ? load_x(cur, x0 + u + 6) : load_x(nxt, u - 2);
        gr[(u + 5) & 7] = u < 3 ? load_g(cur, x0 + u + 5) : load_g(nxt, u - 3);
        step(u, x0 + u);
      }
    }
    cur = nxt;
  }

  // ---- write the partial: D layout col = r (co = co0+2r+t), row i = (e&3)+8(e>>2)+4*half (ci = ci0+2i+s) ------
  const int taps = a.ndzdy * 3;
  float* P = a.partial + static_cast<int64_t>(erange) * taps * a.Cinp * a.Coutp;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int tap = dzdy * 3 + d;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ci = ci0 + 2 * i + s, co = co0 + 2 * r + t;
          P[(static_cast<int64_t>(tap) * a.Cinp + ci) * a.Coutp + co] = acc[d][s][t][e];
        }
  }
  if (do_bias) {
    bsum[0] += __shfl_xor(bsum[0], 32, 64);
    bsum[1] += __shfl_xor(bsum[1], 32, 64);
    if (half == 0) {
      const int ncls = a.up ? (a.ndzdy == 32 ? 8 : 4) : 1;
      float* pb = a.bpartial + (static_cast<int64_t>(erange) * ncls + cls) * a.Coutp + co0 + 2 * r;
      pb[0] = bsum[0]; pb[1] = bsum[1];
    }
  }
}

// ---- weight gradient of the STRIDE-2 conv (TF 'SAME' on even extents: pad 0 before / 1 after; model.py:141-143, 177-179) ----------
//   gW[tz][ty][tx][ci][co] = sum_{b, o} x[b][2 o_z + tz][2 o_y + ty][2 o_x + tx][ci] * g[b][o][co]        (o over the OUTPUT grid)
// Same decomposition as wgrad_kernel: a workgroup owns one (tz, ty) and the three tx taps for a range of OUTPUT rows, a wave a 64x64
// (ci, co) quadrant, both operands straight from L1/L2 into MFMA operand registers.  The gradient ring walks the coarse row, the
// input ring the FINE row two positions per step (16-deep, 12 positions ahead): output position o multiplies X[2o], X[2o+1], X[2o+2]
// with G[o] -- 12 MFMAs for 3 loads.  The only padding is the position past the end of a fine row / plane (2 o + 2 = 2 W).
// The stride-1 kernels on the zero-inserted gradient (round 2) did 8x (3-D) the products of this form -- 2.4x after their Winograd
// saving -- and read a gradient tensor that is 7/8 zeros.  a.D/H/W = OUTPUT extents; partial layout and reduce as wgrad_kernel.
template <int WP8>
__global__ __launch_bounds__(kThreads, 1) void wgrad_s2_kernel(const WgradArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = a.nqi * a.nqj;
  const int qi = (wave % nq) / a.nqj, qj = (wave % nq) % a.nqj, sub = wave / nq;
  constexpr int Wc = WP8 * 8;                 // OUTPUT row length; the input row has 2 * Wc positions
  const int half = lane >> 5, r = lane & 31;
  const int nwg = a.nranges * a.ndzdy;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int range = wg / a.ndzdy, dzdy = wg % a.ndzdy;
  const bool is3d = a.ndzdy == 9;
  const int tz = is3d ? dzdy / 3 : 0, ty = is3d ? dzdy % 3 : dzdy;
  const int Dx = is3d ? 2 * a.D : 1, Hx = 2 * a.H;
  const int ci0 = blockIdx.y * 128 + qi * 64, co0 = blockIdx.z * 128 + qj * 64;
  if (ci0 >= a.Cin || co0 >= a.Cout) return;

  const int ppe = (a.pairs_per_range + a.nsub - 1) / a.nsub;
  const int p0 = range * a.pairs_per_range + sub * ppe;
  int p1 = p0 + ppe;
  if (p1 > (range + 1) * a.pairs_per_range) p1 = (range + 1) * a.pairs_per_range;
  if (p1 > a.npairs) p1 = a.npairs;
  const int erange = range * a.nsub + sub;
  const int cia = ci0 + 2 * r, coa = co0 + 2 * r;
  const bool ci_ok0 = cia < a.Cin, co_ok0 = coa < a.Cout;
  const float* zb = a.zeros;
  struct Row { const float* xb; const float* gb; };
  auto row_setup = [&](int pair) -> Row {
    Row rw;
    const int row = 2 * pair + half;
    const bool ok = pair < p1 && row < a.nrows;
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
    const int zs = is3d ? 2 * z + tz : 0, ys = 2 * y + ty;
    const bool xv = ok && zs < Dx && ys < Hx;
    rw.gb = (ok && co_ok0) ? a.g + (static_cast<int64_t>(row) * Wc) * a.Cout + coa : zb;
    rw.xb = (xv && ci_ok0) ? a.x + (((static_cast<int64_t>(b) * Dx + zs) * Hx + ys) * (2 * Wc)) * a.Cin + cia : zb;
    return rw;
  };
  auto load_x = [&](const Row& rw, int pos) -> f32x2 { return *reinterpret_cast<const f32x2*>(rw.xb + static_cast<int64_t>(pos) * a.Cin); };
  auto load_g = [&](const Row& rw, int pos) -> f32x2 { return *reinterpret_cast<const f32x2*>(rw.gb + static_cast<int64_t>(pos) * a.Cout); };

  f32x16 acc[3][2][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][s][t][e] = 0.f;
  f32x2 bsum = {0.f, 0.f};
  const bool do_bias = a.want_bias && dzdy == a.ndzdy / 2 && blockIdx.y == 0 && qi == 0;

  // X ring by (fine position) % 16, G ring by (coarse position) % 8: both continuous across rows (2 Wc % 16 == 0, Wc % 8 == 0)
  f32x2 xr[16], gr[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) xr[i] = f32x2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) gr[i] = f32x2{0.f, 0.f};
  Row cur = row_setup(p0);
#pragma unroll
  for (int i = 0; i < 12; ++i) xr[i] = load_x(cur, i);
#pragma unroll
  for (int i = 0; i < 5; ++i) gr[i] = load_g(cur, i);

  auto step = [&](int u, int o) {
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 a0 = xr[(2 * u) & 15], a1 = xr[(2 * u + 1) & 15];
    f32x2 a2 = xr[(2 * u + 2) & 15];
    const f32x2 b = gr[u];
    if (o == Wc - 1) a2 = f32x2{0.f, 0.f};      // fine position 2 Wc: the one padded position (the ring slot holds the next row's head)
    bsum[0] += b[0]; bsum[1] += b[1];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b[t], acc[0][s][t], 0, 0, 0);
        acc[1][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b[t], acc[1][s][t], 0, 0, 0);
        acc[2][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], b[t], acc[2][s][t], 0, 0, 0);
      }
  };

  for (int pair = p0; pair < p1; ++pair) {
    const Row nxt = row_setup(pair + 1);
#pragma unroll
    for (int x0 = 0; x0 < (WP8 - 1) * 8; x0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xr[(2 * u + 12) & 15] = load_x(cur, 2 * (x0 + u) + 12);
        xr[(2 * u + 13) & 15] = load_x(cur, 2 * (x0 + u) + 13);
        gr[(u + 5) & 7] = load_g(cur, x0 + u + 5);
        step(u, x0 + u);
      }
    }
    {   // last 8 output positions of the row: the prefetch cursors cross into the next row pair
      constexpr int x0 = Wc - 8;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xr[(2 * u + 12) & 15] = u < 2 ? load_x(cur, 2 * (x0 + u) + 12) : load_x(nxt, 2 * u - 4);
        xr[(2 * u + 13) & 15] = u < 2 ? load_x(cur, 2 * (x0 + u) + 13) : load_x(nxt, 2 * u - 3);
        gr[(u + 5) & 7] = u < 3 ? load_g(cur, x0 + u + 5) : load_g(nxt, u - 3);
        step(u, x0 + u);
      }
    }
    cur = nxt;
  }

  const int taps = a.ndzdy * 3;
  float* P = a.partial + static_cast<int64_t>(erange) * taps * a.Cinp * a.Coutp;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int tap = dzdy * 3 + d;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ci = ci0 + 2 * i + s, co = co0 + 2 * r + t;
          P[(static_cast<int64_t>(tap) * a.Cinp + ci) * a.Coutp + co] = acc[d][s][t][e];
        }
  }
  if (do_bias) {
    bsum[0] += __shfl_xor(bsum[0], 32, 64);
    bsum[1] += __shfl_xor(bsum[1], 32, 64);
    if (half == 0) {
      float* pb = a.bpartial + static_cast<int64_t>(erange) * a.Coutp + co0 + 2 * r;
      pb[0] = bsum[0]; pb[1] = bsum[1];
    }
  }
}

__device__ __forceinline__ f32x2 wpk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 wpk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}


// ---- up-sampling-aware weight gradient, both x-parity classes in one workgroup ----------------------------------------------------
// In up mode (df_upconv_wgrad) a parity class has TWO taps per axis: for px = 0 the coarse offsets (-1, 0), for px = 1 (0, +1); the
// generic kernel computes three and discards one.  Here a workgroup owns the two classes that differ only in px: one coarse X ring,
// two fine-gradient rings (fine positions 2x and 2x+1), and exactly the four useful products per position
//   X[x-1] g0 -> (px=0, slot 0)   X[x] g0 -> (px=0, slot 1)   X[x] g1 -> (px=1, slot 1)   X[x+1] g1 -> (px=1, slot 2)
// = 16 MFMAs for 3 loads (instead of 2 x 12 MFMAs for 2 x 2 loads, a third of them wasted).  Partial layout and reduce unchanged.
// CS > 0: Cin == Cout == CS at compile time (position offsets become load immediates; with run-time strides the fully unrolled
// W = 32 body needs more scalar registers than exist and hipcc's SGPR spilling produced wrong sums -- those shapes keep the
// generic kernel)
template <int WP8, int CS>
__global__ __launch_bounds__(kThreads, 1) void wgrad_up2_kernel(const WgradArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = a.nqi * a.nqj;
  const int qi = (wave % nq) / a.nqj, qj = (wave % nq) % a.nqj, sub = wave / nq;
  constexpr int Wc = WP8 * 8;
  const int half = lane >> 5, r = lane & 31;
  const bool is3d = a.ndzdy == 32;
  const int nhc = a.ndzdy / 2;                   // half-combos: (pz,py) class pair x (dz,dy) delta
  const int nd2 = is3d ? 4 : 2;

  const int nwg = a.nranges * nhc;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int range = wg / nhc, hc = wg % nhc;
  const int clsh = hc / nd2, dd = hc % nd2;
  const int pz = is3d ? (clsh >> 1) & 1 : 0, py = clsh & 1;
  const int dz = is3d ? (dd >> 1) + pz - 1 : 0;
  const int dy = (dd & 1) + py - 1;
  const int cls0 = clsh * 2, cls1 = clsh * 2 + 1;
  const int ci0 = blockIdx.y * 128 + qi * 64, co0 = blockIdx.z * 128 + qj * 64;
  if (ci0 >= a.Cin || co0 >= a.Cout) return;

  const int ppe = (a.pairs_per_range + a.nsub - 1) / a.nsub;
  const int p0 = range * a.pairs_per_range + sub * ppe;
  int p1 = p0 + ppe;
  if (p1 > (range + 1) * a.pairs_per_range) p1 = (range + 1) * a.pairs_per_range;
  if (p1 > a.npairs) p1 = a.npairs;
  const int erange = range * a.nsub + sub;

  const int cia = ci0 + 2 * r, coa = co0 + 2 * r;
  const bool ci_ok0 = cia < a.Cin, co_ok0 = coa < a.Cout;
  const float* zb = a.zeros;
  struct Row { const float* xb; const float* gb; };
  auto row_setup = [&](int pair) -> Row {
    Row rw;
    const int row = 2 * pair + half;
    const bool ok = pair < p1 && row < a.nrows;
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
    const int zs = z + dz, ys = y + dy;
    const bool xv = ok && zs >= 0 && zs < a.D && ys >= 0 && ys < a.H;
    const int64_t gvox = ((static_cast<int64_t>(b) * a.gD + (z * 2 + pz)) * a.gH + (y * 2 + py)) * a.gW;
    const int64_t xvox = ((static_cast<int64_t>(b) * a.D + zs) * a.H + ys) * a.W;
    rw.gb = (ok && co_ok0) ? a.g + gvox * a.Cout + coa : zb;
    rw.xb = (xv && ci_ok0) ? a.x + xvox * a.Cin + cia : zb;
    return rw;
  };
  const int xs = CS ? CS : a.Cin, gs_ = CS ? CS : a.Cout;
  auto load_x = [&](const Row& rw, int pos) -> f32x2 { return *reinterpret_cast<const f32x2*>(rw.xb + static_cast<int64_t>(pos) * xs); };
  auto load_g = [&](const Row& rw, int pos, int px) -> f32x2 {
    return *reinterpret_cast<const f32x2*>(rw.gb + static_cast<int64_t>(2 * pos + px) * gs_);
  };

  // Three products instead of four per position (the reduce kernel only ever needs their sums k0 = X[x-1] g0 + X[x] g1,
  // k1 = X[x] (g0 + g1), k2 = X[x] g0 + X[x+1] g1):  m1 = X[x] (g0 + g1),  m2 = (X[x-1] - X[x]) g0,  m3 = (X[x+1] - X[x]) g1,
  // k0 = m1 + m2, k1 = m1, k2 = m1 + m3.
  f32x16 acc[3][2][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][s][t][e] = 0.f;
  f32x2 bsum0 = {0.f, 0.f}, bsum1 = {0.f, 0.f};
  const bool do_bias = a.want_bias && dd == 0 && blockIdx.y == 0 && qi == 0;

  f32x2 xr[8], g0r[8], g1r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { xr[i] = g0r[i] = g1r[i] = f32x2{0.f, 0.f}; }
  Row cur = row_setup(p0);
#pragma unroll
  for (int i = 0; i < 6; ++i) xr[i] = load_x(cur, i);
#pragma unroll
  for (int i = 0; i < 5; ++i) { g0r[i] = load_g(cur, i, 0); g1r[i] = load_g(cur, i, 1); }

  auto step = [&](int u, int x) {
    __builtin_amdgcn_sched_barrier(0);
    f32x2 am = xr[(u + 7) & 7], a0 = xr[u], ap = xr[(u + 1) & 7];
    const f32x2 g0 = g0r[u], g1 = g1r[u];
    if (x == 0) am = f32x2{0.f, 0.f};
    if (x == Wc - 1) ap = f32x2{0.f, 0.f};
    bsum0 = wpk_add(bsum0, g0); bsum1 = wpk_add(bsum1, g1);
    const f32x2 gs = wpk_add(g0, g1), dm = wpk_sub(am, a0), dp = wpk_sub(ap, a0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], gs[t], acc[0][s][t], 0, 0, 0);
        acc[1][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dm[s], g0[t], acc[1][s][t], 0, 0, 0);
        acc[2][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[s], g1[t], acc[2][s][t], 0, 0, 0);
      }
  };

  for (int pair = p0; pair < p1; ++pair) {
    const Row nxt = row_setup(pair + 1);
#pragma unroll
    for (int x0 = 0; x0 < Wc - 8; x0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xr[(u + 6) & 7] = load_x(cur, x0 + u + 6);
        g0r[(u + 5) & 7] = load_g(cur, x0 + u + 5, 0);
        g1r[(u + 5) & 7] = load_g(cur, x0 + u + 5, 1);
        step(u, x0 + u);
      }
    }
    {
      constexpr int x0 = Wc - 8;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xr[(u + 6) & 7] = u < 2 ? load_x(cur, x0 + u + 6) : load_x(nxt, u - 2);
        g0r[(u + 5) & 7] = u < 3 ? load_g(cur, x0 + u + 5, 0) : load_g(nxt, u - 3, 0);
        g1r[(u + 5) & 7] = u < 3 ? load_g(cur, x0 + u + 5, 1) : load_g(nxt, u - 3, 1);
        step(u, x0 + u);
      }
    }
    cur = nxt;
  }

  // partial slot = (class * nd2 + dd) * 3 + (coarse offset + 1): (cls0: 0, 1), (cls1: 1, 2).  The reduce kernel forms
  // k0 = s(cls0,0) + s(cls1,1), k1 = s(cls0,1) + s(cls1,1), k2 = s(cls0,1) + s(cls1,2); so write s(cls0,0) = m2, s(cls0,1) = 0,
  // s(cls1,1) = m1, s(cls1,2) = m1 + m3.
  float* P = a.partial + static_cast<int64_t>(erange) * a.ndzdy * 3 * a.Cinp * a.Coutp;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int slot = ((d < 2 ? cls0 : cls1) * nd2 + dd) * 3 + (d < 2 ? d : d - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ci = ci0 + 2 * i + s, co = co0 + 2 * r + t;
          const float v = d == 0 ? acc[1][s][t][e] : d == 1 ? 0.f : d == 2 ? acc[0][s][t][e] : acc[0][s][t][e] + acc[2][s][t][e];
          P[(static_cast<int64_t>(slot) * a.Cinp + ci) * a.Coutp + co] = v;
        }
  }
  if (do_bias) {
    bsum0[0] += __shfl_xor(bsum0[0], 32, 64); bsum0[1] += __shfl_xor(bsum0[1], 32, 64);
    bsum1[0] += __shfl_xor(bsum1[0], 32, 64); bsum1[1] += __shfl_xor(bsum1[1], 32, 64);
    if (half == 0) {
      const int ncls = is3d ? 8 : 4;
      float* pb0 = a.bpartial + (static_cast<int64_t>(erange) * ncls + cls0) * a.Coutp + co0 + 2 * r;
      float* pb1 = a.bpartial + (static_cast<int64_t>(erange) * ncls + cls1) * a.Coutp + co0 + 2 * r;
      pb0[0] = bsum0[0]; pb0[1] = bsum0[1];
      pb1[0] = bsum1[0]; pb1[1] = bsum1[1];
    }
  }
}

// ---- Winograd-in-x weight gradient ---------------------------------------------------------------------------------------------
// F(2,3) along the image rows:  for an x-tile of two output positions (2t, 2t+1) and its four inputs d = X[2t-1 .. 2t+2],
//   dU_xi += (B^T d)_xi * (A dy)_xi ,  xi = 0..3,    B^T d = (d0-d2, d1+d2, d2-d1, d1-d3),  A dy = (g0, g0+g1, g0-g1, -g1)
//   gW[dx] = G^T dU :  gW0 = U0 + (U1+U2)/2,  gW1 = (U1-U2)/2,  gW2 = (U1+U2)/2 + U3
// i.e. FOUR transform-domain products per two positions instead of 3 taps x 2 positions: 16 instead of 24 MFMAs per tile
// (1.5x fewer matrix FLOPs), for 6 packed-fp32 transform ops per tile and lane.  Same decomposition, operand streaming and
// register rings as wgrad_kernel (exact-row variants only: W == 8*WP8); the sign of U3 and the G^T combination are applied by
// wgrad_wx_reduce_kernel.  All arithmetic fp32; differs from the direct sum by rounding order only.
// CS > 0: Cin == Cout == CS at compile time (the operand position offsets become load immediates)
template <int WP8, int CS>
__global__ __launch_bounds__(kThreads, 1) void wgrad_wx_kernel(const WgradArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = a.nqi * a.nqj;
  const int qi = (wave % nq) / a.nqj, qj = (wave % nq) % a.nqj, sub = wave / nq;
  constexpr int Wc = WP8 * 8;
  const int half = lane >> 5, r = lane & 31;

  const int nwg = a.nranges * a.ndzdy;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int range = wg / a.ndzdy, dzdy = wg % a.ndzdy;
  const int dz = a.ndzdy == 9 ? dzdy / 3 - 1 : 0;
  const int dy = (a.ndzdy == 9 ? dzdy % 3 : dzdy) - 1;
  const int ci0 = blockIdx.y * 128 + qi * 64, co0 = blockIdx.z * 128 + qj * 64;
  if (ci0 >= a.Cin || co0 >= a.Cout) return;   // wave-uniform: quadrant entirely in the padding

  const int ppe = (a.pairs_per_range + a.nsub - 1) / a.nsub;
  const int p0 = range * a.pairs_per_range + sub * ppe;
  int p1 = p0 + ppe;
  if (p1 > (range + 1) * a.pairs_per_range) p1 = (range + 1) * a.pairs_per_range;
  if (p1 > a.npairs) p1 = a.npairs;
  const int erange = range * a.nsub + sub;

  const int cia = ci0 + 2 * r, coa = co0 + 2 * r;
  const bool ci_ok0 = cia < a.Cin, co_ok0 = coa < a.Cout;
  const float* zb = a.zeros;
  struct Row { const float* xb; const float* gb; };
  auto row_setup = [&](int pair) -> Row {      // invalid rows / padded channel lanes point at the zeroed row-sized region
    Row rw;
    const int row = 2 * pair + half;
    const bool ok = pair < p1 && row < a.nrows;
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
    const int zs = z + dz, ys = y + dy;
    const bool xv = ok && zs >= 0 && zs < a.D && ys >= 0 && ys < a.H;
    const int64_t gvox = ((static_cast<int64_t>(b) * a.D + z) * a.H + y) * a.W;
    const int64_t xvox = ((static_cast<int64_t>(b) * a.D + zs) * a.H + ys) * a.W;
    rw.gb = (ok && co_ok0) ? a.g + gvox * a.Cout + coa : zb;
    rw.xb = (xv && ci_ok0) ? a.x + xvox * a.Cin + cia : zb;
    return rw;
  };
  const int xs = CS ? CS : a.Cin, gs = CS ? CS : a.Cout;
  auto load_x = [&](const Row& rw, int pos) -> f32x2 { return *reinterpret_cast<const f32x2*>(rw.xb + static_cast<int64_t>(pos) * xs); };
  auto load_g = [&](const Row& rw, int pos) -> f32x2 { return *reinterpret_cast<const f32x2*>(rw.gb + static_cast<int64_t>(pos) * gs); };

  f32x16 acc[4][2][2];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][s][t][e] = 0.f;
  f32x2 bsum = {0.f, 0.f};
  const bool do_bias = a.want_bias && dzdy == a.ndzdy / 2 && blockIdx.y == 0 && qi == 0;

  // rings indexed by (position in row) % 8: X holds positions x-1 .. x+5 (+ the two being fetched), G x .. x+5
  f32x2 xr[8], gr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { xr[i] = f32x2{0.f, 0.f}; gr[i] = f32x2{0.f, 0.f}; }
  Row cur = row_setup(p0);
#pragma unroll
  for (int i = 0; i < 6; ++i) { xr[i] = load_x(cur, i); gr[i] = load_g(cur, i); }

  // one x-tile: positions x, x+1 (ring slots u, u+1; u even).  `nx`/`ng` = the four loads issued behind the transform.
  auto tile = [&](int u, int x, const Row& lr, int lpos) {
    __builtin_amdgcn_sched_barrier(0);
    f32x2 dm = xr[(u + 7) & 7], d0 = xr[u], d1 = xr[(u + 1) & 7], d2 = xr[(u + 2) & 7];
    const f32x2 g0 = gr[u], g1 = gr[(u + 1) & 7];
    if (x == 0) dm = f32x2{0.f, 0.f};            // left zero padding (the slot holds the previous row's tail)
    if (x == Wc - 2) d2 = f32x2{0.f, 0.f};       // right zero padding (the slot holds the next row's head)
    const f32x2 v0 = wpk_sub(dm, d1), v1 = wpk_add(d0, d1), v2 = wpk_sub(d1, d0), v3 = wpk_sub(d0, d2);
    const f32x2 m1 = wpk_add(g0, g1), m2 = wpk_sub(g0, g1);
    bsum = wpk_add(bsum, m1);
    __builtin_amdgcn_sched_barrier(0);
    xr[(u + 6) & 7] = load_x(lr, lpos);          // positions x+6, x+7 (slot u+7 held d[-1], consumed above)
    xr[(u + 7) & 7] = load_x(lr, lpos + 1);
    gr[(u + 6) & 7] = load_g(lr, lpos);
    gr[(u + 7) & 7] = load_g(lr, lpos + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[s], g0[t], acc[0][s][t], 0, 0, 0);
        acc[1][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s], m1[t], acc[1][s][t], 0, 0, 0);
        acc[2][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[s], m2[t], acc[2][s][t], 0, 0, 0);
        acc[3][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3[s], g1[t], acc[3][s][t], 0, 0, 0);
      }
  };

  for (int pair = p0; pair < p1; ++pair) {
    const Row nxt = row_setup(pair + 1);
#pragma unroll
    for (int x0 = 0; x0 < Wc - 8; x0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; u += 2) tile(u, x0 + u, cur, x0 + u + 6);
    }
    {   // last 8 positions of the row: the prefetch cursor crosses into the next row pair
      constexpr int x0 = Wc - 8;
      tile(0, x0, cur, x0 + 6);
#pragma unroll
      for (int u = 2; u < 8; u += 2) tile(u, x0 + u, nxt, u - 2);
    }
    cur = nxt;
  }

  // ---- partial: 4 transform-domain slots per (dz,dy); D layout as in wgrad_kernel ---------------------------------------------
  float* P = a.partial + static_cast<int64_t>(erange) * a.ndzdy * 4 * a.Cinp * a.Coutp;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int slot = dzdy * 4 + d;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ci = ci0 + 2 * i + s, co = co0 + 2 * r + t;
          P[(static_cast<int64_t>(slot) * a.Cinp + ci) * a.Coutp + co] = acc[d][s][t][e];
        }
  }
  if (do_bias) {
    bsum[0] += __shfl_xor(bsum[0], 32, 64);
    bsum[1] += __shfl_xor(bsum[1], 32, 64);
    if (half == 0) {
      float* pb = a.bpartial + static_cast<int64_t>(erange) * a.Coutp + co0 + 2 * r;
      pb[0] = bsum[0]; pb[1] = bsum[1];
    }
  }
}

// gb[32 cb .. 32 cb + 31] = sum of the n partial bias rows, by ONE workgroup: thread = (column, range group), every 8th row per thread, the 8
// group sums combined in a fixed order through `sh` (8 x 32 floats of LDS) -- deterministic.  [r6] replaces "workgroup 0 walks all rows serially"
// (up to 256 dependent load rounds that alone set the duration of every reduce kernel).  Call with a block-uniform condition.
__device__ __forceinline__ void bias_reduce_cols(const float* __restrict__ bpartial, float* __restrict__ gb, int n, int Cout, int Coutp,
                                                 float* sh, int cb) {
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = cb * 32 + el;
  float acc = 0.f;
  if (c < Cout)
    for (int rg = grp; rg < n; rg += 8) acc += bpartial[static_cast<int64_t>(rg) * Coutp + c];
  __syncthreads();
  sh[grp * 32 + el] = acc;
  __syncthreads();
  if (grp == 0 && c < Cout) {
    float t = sh[el];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += sh[g * 32 + el];
    gb[c] = t;
  }
}

// gw[dzdy][dx][ci][co] from the summed (fixed order) transform-domain partials U0..U3 (U3 was accumulated with +g1)
__global__ __launch_bounds__(kThreads) void wgrad_wx_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial,
                                                                   float* __restrict__ gw, float* __restrict__ gb, int nranges,
                                                                   int ndzdy, int Cin, int Cout, int Cinp, int Coutp) {
  // [r6] workgroup = 32 consecutive (dzdy, ci, co) elements x 8 range groups (thread sums every 8th range; the 8 group sums are combined in a
  // fixed order through LDS: deterministic).  Rounds 1-5 had ONE thread walk all ranges of an element -- a serial chain of up to 256 dependent
  // load rounds, and workgroup 0 alone another one for the bias: 33-127 us per call at the reference's default batch sizes, 20-30 % of the
  // 2-D steps' kernel time (profiles/r06_probes.md section 5).  Workgroups past the weight elements reduce the bias the same way.
  __shared__ float sU[8][4][32];
  const int64_t total = static_cast<int64_t>(ndzdy) * Cin * Cout;
  const int64_t nwg = (total + 31) / 32;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  if (static_cast<int64_t>(blockIdx.x) >= nwg) {      // bias workgroups
    const int co = static_cast<int>(blockIdx.x - nwg) * 32 + el;
    float acc = 0.f;
    if (gb && co < Cout)
      for (int rg = grp; rg < nranges; rg += 8) acc += bpartial[static_cast<int64_t>(rg) * Coutp + co];
    sU[grp][0][el] = acc;
    __syncthreads();
    if (grp == 0 && gb && co < Cout) {
      float t = sU[0][0][el];
#pragma unroll
      for (int g = 1; g < 8; ++g) t += sU[g][0][el];
      gb[co] = t;
    }
    return;
  }
  const int64_t slot = static_cast<int64_t>(Cinp) * Coutp;
  const int64_t pstride = static_cast<int64_t>(ndzdy) * 4 * slot;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 32 + el;
  const bool ok = i < total;
  const int co = ok ? static_cast<int>(i % Cout) : 0;
  const int64_t t2 = ok ? i / Cout : 0;
  const int ci = static_cast<int>(t2 % Cin);
  const int dzdy = static_cast<int>(t2 / Cin);
  const float* p = partial + (static_cast<int64_t>(dzdy) * 4 * Cinp + ci) * Coutp + co;
  float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
  if (ok)
    for (int rg = grp; rg < nranges; rg += 8) {
      const float* q = p + rg * pstride;
      u0 += q[0]; u1 += q[slot]; u2 += q[2 * slot]; u3 += q[3 * slot];
    }
  sU[grp][0][el] = u0; sU[grp][1][el] = u1; sU[grp][2][el] = u2; sU[grp][3][el] = u3;
  __syncthreads();
  if (grp == 0 && ok) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = sU[0][k][el];
#pragma unroll
      for (int g = 1; g < 8; ++g) t += sU[g][k][el];
      v[k] = t;
    }
    const float h = 0.5f * (v[1] + v[2]);
    float* o = gw + (static_cast<int64_t>(dzdy) * 3 * Cin + ci) * Cout + co;
    o[0] = v[0] + h;
    o[static_cast<int64_t>(Cin) * Cout] = 0.5f * (v[1] - v[2]);
    o[2 * static_cast<int64_t>(Cin) * Cout] = h - v[3];
  }
}

// ---- Winograd-in-(x,y) weight gradient -------------------------------------------------------------------------------------------
// F(2x2,3x3) over the image plane, direct in z: tile = 2x2 output positions, 4x4 inputs.  A workgroup owns ONE (dz, xi_y) and the
// four xi_x of wgrad_wx_kernel; the y part of both transforms is a two-row combination with workgroup-uniform coefficients,
//   X_c = X[row a] + sx * X[row b],   G_c = G[row a] + sg * G[row b]
//   xi_y = 0: X rows (2t-1, 2t+1, -1), G (2t, -, 0) | 1: (2t, 2t+1, +1), (2t, 2t+1, +1) | 2: (2t+1, 2t, -1), (2t, 2t+1, -1)
//          3: (2t, 2t+2, -1), (2t+1, -, 0)   [true (A dy)_3 = -g1: the sign is applied by the reduce kernel, as for xi_x = 3]
// after which a tile row is processed exactly like an image row of the x-only kernel: 3*16 = 48 transform-domain products per 2x2
// positions instead of 27*4 = 108 (2.25x fewer matrix FLOPs), 11 packed-fp32 ops and 8 8-byte loads per x-tile and lane.
// The two half-waves walk two different tile rows.  H, W even; exact-row variants only.
struct WxyArgs {
  WgradArgs w;
  int Ht, ntrows;      // tile rows per (b,z) = H/2; B*D*Ht
};

// GB: the workgroup's xi_y needs the second gradient row (xi_y = 1, 2); xi_y = 0, 3 use one row and skip those loads.
// The two kinds are launched separately (6 | 2 workgroup types each), `a.ndzdy` = types per launch.
template <int WP8, int CS, bool GB>
__global__ __launch_bounds__(kThreads, 1) void wgrad_wxy_kernel(const WxyArgs aa) {
  const WgradArgs& a = aa.w;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = a.nqi * a.nqj;
  const int qi = (wave % nq) / a.nqj, qj = (wave % nq) % a.nqj, sub = wave / nq;
  constexpr int Wc = WP8 * 8;
  const int half = lane >> 5, r = lane & 31;

  const int nwg = a.nranges * a.ndzdy;          // ndzdy = (3 | 1) dz x 4 xi_y
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int range = wg / a.ndzdy, dzsel = wg % a.ndzdy;
  const int dzi = dzsel >> 1;                          // 0..2 (3-D) | 0 (2-D)
  const int dz = a.ndzdy == 6 ? dzi - 1 : 0;
  const int xiy = GB ? 1 + (dzsel & 1) : 3 * (dzsel & 1);
  const int dzxy = dzi * 4 + xiy;                      // partial slot group
  const int ci0 = blockIdx.y * 128 + qi * 64, co0 = blockIdx.z * 128 + qj * 64;
  if (ci0 >= a.Cin || co0 >= a.Cout) return;

  const int ppe = (a.pairs_per_range + a.nsub - 1) / a.nsub;
  const int p0 = range * a.pairs_per_range + sub * ppe;
  int p1 = p0 + ppe;
  if (p1 > (range + 1) * a.pairs_per_range) p1 = (range + 1) * a.pairs_per_range;
  if (p1 > a.npairs) p1 = a.npairs;
  const int erange = range * a.nsub + sub;

  const int cia = ci0 + 2 * r, coa = co0 + 2 * r;
  const bool ci_ok0 = cia < a.Cin, co_ok0 = coa < a.Cout;
  const float* zb = a.zeros;
  // row offsets (relative to 2t) and coefficients of this workgroup's xi_y
  const int xoa = xiy == 0 ? -1 : xiy == 2 ? 1 : 0, xob = xiy == 0 ? 1 : xiy == 1 ? 1 : xiy == 2 ? 0 : 2;
  const float sxf = xiy == 1 ? 1.f : -1.f;
  const int goa = xiy == 3 ? 1 : 0;
  const float sgf = xiy == 1 ? 1.f : xiy == 2 ? -1.f : 0.f;
  const f32x2 sx2 = {sxf, sxf}, sg2 = {sgf, sgf};
  struct Row { const float* xa; const float* xb; const float* ga; const float* gb; };
  auto row_setup = [&](int pair) -> Row {
    Row rw;
    const int trow = 2 * pair + half;
    const bool ok = pair < p1 && trow < aa.ntrows;
    const int yt = trow % aa.Ht;
    const int t = trow / aa.Ht;
    const int z = t % a.D;
    const int b = t / a.D;
    const int zs = z + dz, y0 = 2 * yt;
    const bool zv = ok && zs >= 0 && zs < a.D;
    const int ya = y0 + xoa, yb = y0 + xob;
    const int64_t gbase = (static_cast<int64_t>(b) * a.D + z) * a.H;
    const int64_t xbase = (static_cast<int64_t>(b) * a.D + zs) * a.H;
    rw.xa = (zv && ya >= 0 && ya < a.H && ci_ok0) ? a.x + (xbase + ya) * a.W * a.Cin + cia : zb;
    rw.xb = (zv && yb >= 0 && yb < a.H && ci_ok0) ? a.x + (xbase + yb) * a.W * a.Cin + cia : zb;
    rw.ga = (ok && co_ok0) ? a.g + (gbase + y0 + goa) * a.W * a.Cout + coa : zb;
    rw.gb = (GB && ok && co_ok0) ? a.g + (gbase + y0 + 1) * a.W * a.Cout + coa : zb;
    return rw;
  };
  const int xs = CS ? CS : a.Cin, gs = CS ? CS : a.Cout;
  auto ld = [&](const float* base, int pos, int stride) -> f32x2 {
    return *reinterpret_cast<const f32x2*>(base + static_cast<int64_t>(pos) * stride);
  };
  auto pkfma = [&](f32x2 x, f32x2 y, f32x2 z) -> f32x2 {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
  };

  f32x16 acc[4][2][2];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][s][t][e] = 0.f;
  f32x2 bsum = {0.f, 0.f};
  const bool do_bias = a.want_bias && dz == 0 && xiy == 1 && blockIdx.y == 0 && qi == 0;     // G_c = g(2t) + g(2t+1)

  // raw rings by (position in row) % 8 and the ring of y-combined X values
  f32x2 xra[8], xrb[8], gra[8], grb[8], xc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { xra[i] = xrb[i] = gra[i] = grb[i] = xc[i] = f32x2{0.f, 0.f}; }
  Row cur = row_setup(p0);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    xra[i] = ld(cur.xa, i, xs); xrb[i] = ld(cur.xb, i, xs);
    gra[i] = ld(cur.ga, i, gs);
    if (GB) grb[i] = ld(cur.gb, i, gs);
  }
  xc[0] = pkfma(xrb[0], sx2, xra[0]);

  auto tile = [&](int u, int x, const Row& lr, int lpos) {
    __builtin_amdgcn_sched_barrier(0);
    // y-combination of the two positions this tile is the first to need, and of its two gradient positions
    xc[(u + 1) & 7] = pkfma(xrb[(u + 1) & 7], sx2, xra[(u + 1) & 7]);
    xc[(u + 2) & 7] = pkfma(xrb[(u + 2) & 7], sx2, xra[(u + 2) & 7]);
    const f32x2 g0 = GB ? pkfma(grb[u], sg2, gra[u]) : gra[u], g1 = GB ? pkfma(grb[(u + 1) & 7], sg2, gra[(u + 1) & 7]) : gra[(u + 1) & 7];
    f32x2 dm = xc[(u + 7) & 7], d0 = xc[u], d1 = xc[(u + 1) & 7], d2 = xc[(u + 2) & 7];
    if (x == 0) dm = f32x2{0.f, 0.f};
    if (x == Wc - 2) d2 = f32x2{0.f, 0.f};
    const f32x2 v0 = wpk_sub(dm, d1), v1 = wpk_add(d0, d1), v2 = wpk_sub(d1, d0), v3 = wpk_sub(d0, d2);
    const f32x2 m1 = wpk_add(g0, g1), m2 = wpk_sub(g0, g1);
    bsum = wpk_add(bsum, m1);
    __builtin_amdgcn_sched_barrier(0);
    xra[(u + 6) & 7] = ld(lr.xa, lpos, xs); xra[(u + 7) & 7] = ld(lr.xa, lpos + 1, xs);
    xrb[(u + 6) & 7] = ld(lr.xb, lpos, xs); xrb[(u + 7) & 7] = ld(lr.xb, lpos + 1, xs);
    gra[(u + 6) & 7] = ld(lr.ga, lpos, gs); gra[(u + 7) & 7] = ld(lr.ga, lpos + 1, gs);
    if (GB) { grb[(u + 6) & 7] = ld(lr.gb, lpos, gs); grb[(u + 7) & 7] = ld(lr.gb, lpos + 1, gs); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[s], g0[t], acc[0][s][t], 0, 0, 0);
        acc[1][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s], m1[t], acc[1][s][t], 0, 0, 0);
        acc[2][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[s], m2[t], acc[2][s][t], 0, 0, 0);
        acc[3][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3[s], g1[t], acc[3][s][t], 0, 0, 0);
      }
  };

  for (int pair = p0; pair < p1; ++pair) {
    const Row nxt = row_setup(pair + 1);
#pragma unroll
    for (int x0 = 0; x0 < Wc - 8; x0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; u += 2) tile(u, x0 + u, cur, x0 + u + 6);
    }
    {
      constexpr int x0 = Wc - 8;
      tile(0, x0, cur, x0 + 6);
#pragma unroll
      for (int u = 2; u < 8; u += 2) tile(u, x0 + u, nxt, u - 2);
    }
    cur = nxt;
  }

  // ---- partial: slot = (dz, xi_y) * 4 + xi_x ------------------------------------------------------------------------------------
  float* P = a.partial + static_cast<int64_t>(erange) * (a.ndzdy * 2) * 4 * a.Cinp * a.Coutp;      // 2 launches x ndzdy groups
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int slot = dzxy * 4 + d;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ci = ci0 + 2 * i + s, co = co0 + 2 * r + t;
          P[(static_cast<int64_t>(slot) * a.Cinp + ci) * a.Coutp + co] = acc[d][s][t][e];
        }
  }
  if (do_bias) {
    bsum[0] += __shfl_xor(bsum[0], 32, 64);
    bsum[1] += __shfl_xor(bsum[1], 32, 64);
    if (half == 0) {
      float* pb = a.bpartial + static_cast<int64_t>(erange) * a.Coutp + co0 + 2 * r;
      pb[0] = bsum[0]; pb[1] = bsum[1];
    }
  }
}

// gw[dz][dy][dx][ci][co] = G^T_y G^T_x of the summed (fixed order) partials U[xi_y][xi_x]; U[3][.] and U[.][3] carry a flipped sign.
// Workgroup = 32 consecutive (dz,ci,co) elements x 8 range groups: each thread sums every 8th range, the 8 group sums are combined
// in a fixed order through LDS (deterministic), then one thread per element applies the two G^T.
__global__ __launch_bounds__(kThreads) void wgrad_wxy_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial,
                                                                    float* __restrict__ gw, float* __restrict__ gb, int nranges, int ndz,
                                                                    int Cin, int Cout, int Cinp, int Coutp) {
  __shared__ float sU[8][16][32];
  const int64_t total = static_cast<int64_t>(ndz) * Cin * Cout;
  const int64_t slot = static_cast<int64_t>(Cinp) * Coutp;
  const int64_t pstride = static_cast<int64_t>(ndz) * 16 * slot;
  const int64_t tapstride = static_cast<int64_t>(Cin) * Cout;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 32 + el;
  const bool ok = i < total;
  const int co = ok ? static_cast<int>(i % Cout) : 0;
  const int64_t t2 = ok ? i / Cout : 0;
  const int ci = static_cast<int>(t2 % Cin);
  const int dzi = static_cast<int>(t2 / Cin);
  const float* p = partial + (static_cast<int64_t>(dzi) * 16 * Cinp + ci) * Coutp + co;
  float u[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) u[k] = 0.f;
  if (ok) {
    for (int rg = grp; rg < nranges; rg += 8) {
      const float* q = p + rg * pstride;
#pragma unroll
      for (int k = 0; k < 16; ++k) u[k] += q[k * slot];
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) sU[grp][k][el] = u[k];
  __syncthreads();
  if (grp == 0 && ok) {
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float acc = sU[0][k][el];
#pragma unroll
      for (int g = 1; g < 8; ++g) acc += sU[g][k][el];
      v[k >> 2][k & 3] = acc;
    }
    float w[3][4];      // G^T along y
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float h = 0.5f * (v[1][k] + v[2][k]);
      w[0][k] = v[0][k] + h;
      w[1][k] = 0.5f * (v[1][k] - v[2][k]);
      w[2][k] = h - v[3][k];
    }
    float* o = gw + (static_cast<int64_t>(dzi) * 9 * Cin + ci) * Cout + co;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float h = 0.5f * (w[dy][1] + w[dy][2]);
      o[(dy * 3 + 0) * tapstride] = w[dy][0] + h;
      o[(dy * 3 + 1) * tapstride] = 0.5f * (w[dy][1] - w[dy][2]);
      o[(dy * 3 + 2) * tapstride] = h - w[dy][3];
    }
  }
  if (gb && static_cast<int>(blockIdx.x) * 32 < Cout) bias_reduce_cols(bpartial, gb, nranges, Cout, Coutp, &sU[0][0][0], blockIdx.x);
}

// ---- Winograd-in-(x,y,z) weight gradient ------------------------------------------------------------------------------------------
// F(2x2x2, 3x3x3): the z axis gets the same two-term combination as y in wgrad_wxy_kernel, so a workgroup owns ONE (xi_z, xi_y) and
// the four xi_x, and both operands are combinations of 2 planes x 2 rows with workgroup-uniform coefficients
//   X_c = (X[za][ya] + sy X[za][yb]) + sz (X[zb][ya] + sy X[zb][yb]),   G_c likewise over (1 | 2 planes) x (1 | 2 rows).
// 64 transform-domain products per 2x2x2 positions instead of 96 in the (x,y) form (216 direct): 1.5x fewer matrix FLOPs for twice the
// operand loads per x-tile (16 8-byte loads, 17 packed-fp32 ops, 16 MFMAs).  Tile rows are (b, z pair, y pair); D, H, W even.
// GZ / GY: xi_z / xi_y in {1, 2} (the gradient combination needs the second plane / row); {0, 3} use one: four compile-time
// specialisations of the body, dispatched per workgroup by the one-launch kernels below (wgrad_wxyz_fused_kernel / _up_fused_kernel).
struct WxyzArgs {
  WgradArgs w;
  int Ht, Dt, ntrows;      // tile rows per plane pair = H/2; plane pairs per batch = D/2; B*Dt*Ht
};

// UP: x is the COARSE tensor of an up-sampling-aware conv (fine position p reads xc[p >> 1] on every axis; a.D/H/W are the fine extents):
// the transform points with index 2 vanish for the duplicated input, so only xi_z, xi_y in {0, 1, 3} workgroup types exist and the
// xi_x = 2 products are skipped -- 27 of the 64 products (wgrad_up2_kernel's parity-class form needs 48 per coarse voxel).
// (66,560 B of static LDS: above the 64 KiB every CDNA part before gfx950 gives a workgroup -- this library is gfx950-only, Makefile ARCH)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "conv_wgrad.hip: the 64-channel (x,y,z) weight-gradient kernels need gfx950's 160 KB LDS (build with --offload-arch=gfx950)"
#endif
constexpr int kWxyzLds = 4 * 64 * 64 + 4 * 64;      // floats: the 64 -> 64 kernels' in-workgroup sum of the four waves' partials (one xi_x slot at a time)
// XCD-aware bijective block remap shared by the kernels below: workgroup b runs on XCD b % 8; every XCD gets a contiguous run
__device__ __forceinline__ int wxyz_wg(int nwg) {
  const int bid = blockIdx.x, q = nwg >> 3, rem = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

// the work of ONE workgroup: voxel range `range`, transform point (xi_z, xi_y) number `sel` of the (GZ, GY) class
template <int WP8, int CS, bool GZ, bool GY, bool UP, int DBG>
__device__ __forceinline__ void wgrad_wxyz_body(const WxyzArgs& aa, int range, int sel, float* lds) {      // lds: kWxyzLds floats when CS == 64
  const WgradArgs& a = aa.w;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = a.nqi * a.nqj;
  const int qi = (wave % nq) / a.nqj, qj = (wave % nq) % a.nqj, sub = wave / nq;
  constexpr int Wc = WP8 * 8;
  const int half = lane >> 5, r = lane & 31;

  constexpr int NTY = (UP && GY) ? 1 : 2;
  const int xiz = GZ ? 1 + sel / NTY : 3 * (sel / NTY);
  const int xiy = GY ? 1 + sel % NTY : 3 * (sel % NTY);
  const int zy = xiz * 4 + xiy;                        // partial slot group
  const int ci0 = blockIdx.y * 128 + qi * 64, co0 = blockIdx.z * 128 + qj * 64;
  if (ci0 >= a.Cin || co0 >= a.Cout) return;

  const int ppe = (a.pairs_per_range + a.nsub - 1) / a.nsub;
  // [r5] 64 -> 64 layers: the nsub = 4 waves that split a range take INTERLEAVED tile-row pairs (pair = first + sub + nsub i), not contiguous
  // sub-ranges: at any time they stream neighbouring rows and share the y-halo rows of x through the L1 -- 5.84 -> 5.68 ms at 128^3 x 4
  // (tools/r05_wgrad64_probe.py; tuning variant 16 = the contiguous split of rounds 3-4).  Same partial slots, fixed summation order.
  constexpr bool IL = (CS == 64) != ((DBG & 16) != 0);
  const int pstep = IL ? a.nsub : 1;
  const int p0 = range * a.pairs_per_range + (IL ? sub : sub * ppe);
  int p1 = IL ? (range + 1) * a.pairs_per_range : p0 + ppe;
  if (p1 > (range + 1) * a.pairs_per_range) p1 = (range + 1) * a.pairs_per_range;
  if (p1 > a.npairs) p1 = a.npairs;
  const int erange = range * a.nsub + sub;

  const int cia = ci0 + 2 * r, coa = co0 + 2 * r;
  const bool ci_ok0 = cia < a.Cin, co_ok0 = coa < a.Cout;
  const float* zb = a.zeros;
  // offsets (relative to 2t) and coefficients of this workgroup's xi_y / xi_z (same table for both axes)
  const int yoa = xiy == 0 ? -1 : xiy == 2 ? 1 : 0, yob = xiy == 0 ? 1 : xiy == 1 ? 1 : xiy == 2 ? 0 : 2;
  const int zoa = xiz == 0 ? -1 : xiz == 2 ? 1 : 0, zob = xiz == 0 ? 1 : xiz == 1 ? 1 : xiz == 2 ? 0 : 2;
  const float syf = xiy == 1 ? 1.f : -1.f, szf = xiz == 1 ? 1.f : -1.f;
  const int gya = xiy == 3 ? 1 : 0, gza = xiz == 3 ? 1 : 0;
  const float sgyf = xiy == 1 ? 1.f : -1.f, sgzf = xiz == 1 ? 1.f : -1.f;      // (used only for xi in {1, 2})
  const f32x2 sy2 = {syf, syf}, sz2 = {szf, szf}, sgy2 = {sgyf, sgyf}, sgz2 = {sgzf, sgzf};
  struct Row { const float* x[2][2]; const float* g[2][2]; };      // [plane a|b][row a|b]
  auto row_setup = [&](int pair) -> Row {
    Row rw;
    const int trow = (DBG & 8) ? 2 * (pair & 3) + half : 2 * pair + half;      // (tuning library, DBG 8: every range reads the same 8 tile rows -- an L2-resident working set)
    const bool ok = pair < p1 && trow < aa.ntrows;
    const int yt = trow % aa.Ht;
    const int t = trow / aa.Ht;
    const int zt = t % aa.Dt;
    const int b = t / aa.Dt;
    const int z0 = 2 * zt, y0 = 2 * yt;
    const int64_t bbase = static_cast<int64_t>(b) * a.D;
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int zs = z0 + (pz ? zob : zoa);
      const bool zv = ok && zs >= 0 && zs < a.D && ci_ok0;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int ys = y0 + (py ? yob : yoa);
        if (UP) rw.x[pz][py] = (zv && ys >= 0 && ys < a.H)
                                   ? a.x + ((static_cast<int64_t>(b) * (a.D >> 1) + (zs >> 1)) * (a.H >> 1) + (ys >> 1)) * (a.W >> 1) * a.Cin + cia : zb;
        else rw.x[pz][py] = (zv && ys >= 0 && ys < a.H) ? a.x + ((bbase + zs) * a.H + ys) * a.W * a.Cin + cia : zb;
      }
      const int gz = z0 + (pz ? 1 : gza);
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int gy = y0 + (py ? 1 : gya);
        const bool need = (pz == 0 || GZ) && (py == 0 || GY);
        rw.g[pz][py] = (need && ok && co_ok0) ? a.g + ((bbase + gz) * a.H + gy) * a.W * a.Cout + coa : zb;
      }
    }
    if (DBG & 1) {        // (tuning library) latency experiment: every operand load reads the cached zero row
#pragma unroll
      for (int q = 0; q < 4; ++q) { rw.x[q >> 1][q & 1] = zb; rw.g[q >> 1][q & 1] = zb; }
    }
    return rw;
  };
  const int xs = CS ? CS : a.Cin, gs = CS ? CS : a.Cout;
  auto ld = [&](const float* base, int pos, int stride) -> f32x2 {
    return *reinterpret_cast<const f32x2*>(base + static_cast<int64_t>(pos) * stride);
  };
  auto pkfma = [&](f32x2 x, f32x2 y, f32x2 z) -> f32x2 {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
  };

  f32x16 acc[4][2][2];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][s][t][e] = 0.f;
  f32x2 bsum = {0.f, 0.f};
  const bool do_bias = a.want_bias && GZ && GY && sel == 0 && blockIdx.y == 0 && qi == 0;     // G_c = sum of the 2x2 (z, y) rows

  // raw rings by (position in row) % 8 and the ring of combined X values
  f32x2 xr[2][2][8], gr[2][2][8], xc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { xr[q >> 1][q & 1][i] = f32x2{0.f, 0.f}; gr[q >> 1][q & 1][i] = f32x2{0.f, 0.f}; }
    xc[i] = f32x2{0.f, 0.f};
  }
  auto load_pos = [&](const Row& lr, int slot, int pos) {
#pragma unroll
    for (int q = 0; q < 4; ++q) xr[q >> 1][q & 1][slot] = ld(lr.x[q >> 1][q & 1], UP ? pos >> 1 : pos, xs);
    gr[0][0][slot] = ld(lr.g[0][0], pos, gs);
    if (GY) gr[0][1][slot] = ld(lr.g[0][1], pos, gs);
    if (GZ) gr[1][0][slot] = ld(lr.g[1][0], pos, gs);
    if (GZ && GY) gr[1][1][slot] = ld(lr.g[1][1], pos, gs);
  };
  auto xcomb = [&](int slot) -> f32x2 {
    if (DBG & 2) return xr[0][0][slot];      // (tuning library) no (z, y) combination
    const f32x2 ta = pkfma(xr[0][1][slot], sy2, xr[0][0][slot]), tb = pkfma(xr[1][1][slot], sy2, xr[1][0][slot]);
    return pkfma(tb, sz2, ta);
  };
  auto gcomb = [&](int slot) -> f32x2 {
    if (DBG & 2) return gr[0][0][slot];
    if (GZ && GY) {
      const f32x2 ta = pkfma(gr[0][1][slot], sgy2, gr[0][0][slot]), tb = pkfma(gr[1][1][slot], sgy2, gr[1][0][slot]);
      return pkfma(tb, sgz2, ta);
    }
    if (GY) return pkfma(gr[0][1][slot], sgy2, gr[0][0][slot]);
    if (GZ) return pkfma(gr[1][0][slot], sgz2, gr[0][0][slot]);
    return gr[0][0][slot];
  };
  Row cur = row_setup(p0);
#pragma unroll
  for (int i = 0; i < 6; ++i) load_pos(cur, i, i);
  xc[0] = xcomb(0);

  auto tile = [&](int u, int x, const Row& lr, int lpos) {
    __builtin_amdgcn_sched_barrier(0);
    xc[(u + 1) & 7] = xcomb((u + 1) & 7);
    xc[(u + 2) & 7] = xcomb((u + 2) & 7);
    const f32x2 g0 = gcomb(u), g1 = gcomb((u + 1) & 7);
    f32x2 dm = xc[(u + 7) & 7], d0 = xc[u], d1 = xc[(u + 1) & 7], d2 = xc[(u + 2) & 7];
    if (x == 0) dm = f32x2{0.f, 0.f};
    if (x == Wc - 2) d2 = f32x2{0.f, 0.f};
    const f32x2 v0 = (DBG & 4) ? dm : wpk_sub(dm, d1), v1 = (DBG & 4) ? d0 : wpk_add(d0, d1), v2 = (DBG & 4) ? d1 : wpk_sub(d1, d0),
                v3 = (DBG & 4) ? d2 : wpk_sub(d0, d2);      // (DBG & 4, tuning library: no x transform)
    const f32x2 m1 = (DBG & 4) ? g0 : wpk_add(g0, g1), m2 = (DBG & 4) ? g1 : wpk_sub(g0, g1);
    if (!(DBG & 4)) bsum = wpk_add(bsum, m1);
    __builtin_amdgcn_sched_barrier(0);
    load_pos(lr, (u + 6) & 7, lpos);
    load_pos(lr, (u + 7) & 7, lpos + 1);
    if (UP) {      // lpos is even: both fine positions read the same coarse voxel -- keep one load
#pragma unroll
      for (int q = 0; q < 4; ++q) xr[q >> 1][q & 1][(u + 7) & 7] = xr[q >> 1][q & 1][(u + 6) & 7];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[s], g0[t], acc[0][s][t], 0, 0, 0);
        acc[1][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s], m1[t], acc[1][s][t], 0, 0, 0);
        if (!UP) acc[2][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[s], m2[t], acc[2][s][t], 0, 0, 0);
        acc[3][s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3[s], g1[t], acc[3][s][t], 0, 0, 0);
      }
  };

  for (int pair = p0; pair < p1; pair += pstep) {
    const Row nxt = row_setup(pair + pstep);
#pragma unroll
    for (int x0 = 0; x0 < Wc - 8; x0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; u += 2) tile(u, x0 + u, cur, x0 + u + 6);
    }
    {
      constexpr int x0 = Wc - 8;
      tile(0, x0, cur, x0 + 6);
#pragma unroll
      for (int u = 2; u < 8; u += 2) tile(u, x0 + u, nxt, u - 2);
    }
    cur = nxt;
  }

  // ---- partial: slot = (xi_z, xi_y) * 4 + xi_x -----------------------------------------------------------------------------------
  if constexpr (CS == 64) {
    // [r5] 64 -> 64: the four waves of the workgroup hold four partial sums of the SAME 64 x 64 tile (they split the range).  They are added
    // here, in wave order, through LDS -- one partial per range instead of four: a quarter of the partial writes and of the fixed-order
    // reduce's reads (cfg5: 25 reduce launches of 146 us per step).  One xi_x slot (4 x 16 KB) at a time.
    float (*sR)[64 * 64] = reinterpret_cast<float (*)[64 * 64]>(lds);
    float (*sBs)[64] = reinterpret_cast<float (*)[64]>(lds + 4 * 64 * 64);
    float* P1 = a.partial + static_cast<int64_t>(range) * 64 * a.Cinp * a.Coutp;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (UP && d == 2) continue;
      const int slot = zy * 4 + d;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
            sR[sub][(2 * i + s) * 64 + 2 * r + t] = acc[d][s][t][e];
          }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = q * kThreads + tid;
        const float v = ((sR[0][j] + sR[1][j]) + sR[2][j]) + sR[3][j];
        P1[(static_cast<int64_t>(slot) * a.Cinp + ci0 + (j >> 6)) * a.Coutp + co0 + (j & 63)] = v;
      }
      __syncthreads();
    }
    if (a.want_bias && GZ && GY && sel == 0 && blockIdx.y == 0) {      // (workgroup-uniform: all four waves sit on quadrant (0, 0))
      bsum[0] += __shfl_xor(bsum[0], 32, 64);
      bsum[1] += __shfl_xor(bsum[1], 32, 64);
      if (half == 0) { sBs[sub][2 * r] = bsum[0]; sBs[sub][2 * r + 1] = bsum[1]; }
      __syncthreads();
      if (tid < 64) a.bpartial[static_cast<int64_t>(range) * a.Coutp + co0 + tid] = ((sBs[0][tid] + sBs[1][tid]) + sBs[2][tid]) + sBs[3][tid];
    }
    return;
  }
  float* P = a.partial + static_cast<int64_t>(erange) * 64 * a.Cinp * a.Coutp;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    if (UP && d == 2) continue;
    const int slot = zy * 4 + d;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ci = ci0 + 2 * i + s, co = co0 + 2 * r + t;
          P[(static_cast<int64_t>(slot) * a.Cinp + ci) * a.Coutp + co] = acc[d][s][t][e];
        }
  }
  if (do_bias) {
    bsum[0] += __shfl_xor(bsum[0], 32, 64);
    bsum[1] += __shfl_xor(bsum[1], 32, 64);
    if (half == 0) {
      float* pb = a.bpartial + static_cast<int64_t>(erange) * a.Coutp + co0 + 2 * r;
      pb[0] = bsum[0]; pb[1] = bsum[1];
    }
  }
}

template <int WP8, int CS, bool GZ, bool GY, bool UP = false, int DBG = 0>
__global__ __launch_bounds__(kThreads, 1) void wgrad_wxyz_kernel(const WxyzArgs aa) {
  constexpr int NT = ((UP && GZ) ? 1 : 2) * ((UP && GY) ? 1 : 2);      // workgroup types of this launch
  __shared__ float sLds[CS == 64 ? kWxyzLds : 1];
  const int wg = wxyz_wg(aa.w.nranges * NT);
  wgrad_wxyz_body<WP8, CS, GZ, GY, UP, DBG>(aa, wg / NT, wg % NT, sLds);
}

// All 16 (xi_z, xi_y) types of a voxel range in ONE launch, adjacent in the grid: they run at the same time on the same XCD, so the
// rows of x and g a range streams are fetched from the fabric once and the other 15 readers hit that XCD's L2 (four launches by
// (GZ, GY) class re-read both tensors from HBM four times, and an HBM miss outlasts the 6-position register prefetch).
template <int WP8, int CS, int DBG = 0>
__global__ __launch_bounds__(kThreads, 1) void wgrad_wxyz_fused_kernel(const WxyzArgs aa) {
  const int wg = wxyz_wg(aa.w.nranges * 16);
  const int range = wg >> 4, type = wg & 15;
  const int xiz = type >> 2, xiy = type & 3;
  const bool gz = xiz == 1 || xiz == 2, gy = xiy == 1 || xiy == 2;
  const int sel = (gz ? xiz - 1 : xiz / 3) * 2 + (gy ? xiy - 1 : xiy / 3);
  __shared__ float sLds[CS == 64 ? kWxyzLds : 1];
  if (gz && gy) wgrad_wxyz_body<WP8, CS, true, true, false, DBG>(aa, range, sel, sLds);
  else if (gz) wgrad_wxyz_body<WP8, CS, true, false, false, DBG>(aa, range, sel, sLds);
  else if (gy) wgrad_wxyz_body<WP8, CS, false, true, false, DBG>(aa, range, sel, sLds);
  else wgrad_wxyz_body<WP8, CS, false, false, false, DBG>(aa, range, sel, sLds);
}

// The up-sampling-aware form has 9 live (xi_z, xi_y) types (xi in {0, 1, 3} per axis): one launch as well (four launches by class
// had 4 | 2 | 2 | 1 types -- grids of nranges x {4, 2, 2, 1} workgroups, three of them at most one round of the chip).
template <int WP8, int CS>
__global__ __launch_bounds__(kThreads, 1) void wgrad_wxyz_up_fused_kernel(const WxyzArgs aa) {
  const int wg = wxyz_wg(aa.w.nranges * 9);
  const int range = wg / 9, t9 = wg % 9;
  const int zi = t9 / 3, yi = t9 % 3;                      // 0 -> xi = 0, 1 -> xi = 1, 2 -> xi = 3
  const bool gz = zi == 1, gy = yi == 1;
  const int selz = zi == 2 ? 1 : 0, sely = yi == 2 ? 1 : 0;
  __shared__ float sLds[CS == 64 ? kWxyzLds : 1];
  if (gz && gy) wgrad_wxyz_body<WP8, CS, true, true, true, 0>(aa, range, 0, sLds);
  else if (gz) wgrad_wxyz_body<WP8, CS, true, false, true, 0>(aa, range, sely, sLds);
  else if (gy) wgrad_wxyz_body<WP8, CS, false, true, true, 0>(aa, range, selz, sLds);
  else wgrad_wxyz_body<WP8, CS, false, false, true, 0>(aa, range, selz * 2 + sely, sLds);
}

// gw[dz][dy][dx][ci][co] = G^T_z G^T_y G^T_x of the summed (fixed order) partials U[xi_z][xi_y][xi_x]; index 3 of every axis carries a
// flipped sign.  Workgroup = 32 consecutive (ci, co) elements x 8 range groups; every group applies the (linear) transform to its own
// sums, the 8 x 27 results are combined in a fixed order through LDS.
__global__ __launch_bounds__(kThreads) void wgrad_wxyz_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial,
                                                                     float* __restrict__ gw, float* __restrict__ gb, int nranges, int Cin,
                                                                     int Cout, int Cinp, int Coutp, int up) {
  __shared__ float sV[8][27][32];
  const int64_t total = static_cast<int64_t>(Cin) * Cout;
  const int64_t slot = static_cast<int64_t>(Cinp) * Coutp;
  const int64_t pstride = 64 * slot;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 32 + el;
  const bool ok = i < total;
  const int co = ok ? static_cast<int>(i % Cout) : 0;
  const int ci = ok ? static_cast<int>(i / Cout) : 0;
  const float* p = partial + static_cast<int64_t>(ci) * Coutp + co;
  float u[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) u[k] = 0.f;
  if (ok) {
    for (int rg = grp; rg < nranges; rg += 8) {
      const float* q = p + rg * pstride;
#pragma unroll
      for (int k = 0; k < 64; ++k)
        if (!(up && ((k >> 4) == 2 || ((k >> 2) & 3) == 2 || (k & 3) == 2))) u[k] += q[k * slot];      // (never written in up mode)
    }
  }
  // x, then y, then z:  (u0 + h, (u1 - u2)/2, h - u3) with h = (u1 + u2)/2
  float a1[16][3];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float h = 0.5f * (u[k * 4 + 1] + u[k * 4 + 2]);
    a1[k][0] = u[k * 4] + h; a1[k][1] = 0.5f * (u[k * 4 + 1] - u[k * 4 + 2]); a1[k][2] = h - u[k * 4 + 3];
  }
  float a2[4][3][3];      // [xi_z][dy][dx]
#pragma unroll
  for (int z = 0; z < 4; ++z)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const float h = 0.5f * (a1[z * 4 + 1][dx] + a1[z * 4 + 2][dx]);
      a2[z][0][dx] = a1[z * 4][dx] + h; a2[z][1][dx] = 0.5f * (a1[z * 4 + 1][dx] - a1[z * 4 + 2][dx]); a2[z][2][dx] = h - a1[z * 4 + 3][dx];
    }
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const float h = 0.5f * (a2[1][dy][dx] + a2[2][dy][dx]);
      sV[grp][0 * 9 + dy * 3 + dx][el] = a2[0][dy][dx] + h;
      sV[grp][1 * 9 + dy * 3 + dx][el] = 0.5f * (a2[1][dy][dx] - a2[2][dy][dx]);
      sV[grp][2 * 9 + dy * 3 + dx][el] = h - a2[3][dy][dx];
    }
  __syncthreads();
  for (int k = grp; k < 27; k += 8) {
    if (!ok) break;
    float t = sV[0][k][el];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += sV[g][k][el];
    gw[(static_cast<int64_t>(k) * Cin + ci) * Cout + co] = t;
  }
  if (gb && static_cast<int>(blockIdx.x) * 32 < Cout) bias_reduce_cols(bpartial, gb, nranges, Cout, Coutp, &sV[0][0][0], blockIdx.x);
}

// the Winograd-in-x variant exists for the fully unrolled row lengths below (even channel counts: float2 operand loads)
inline bool wx_ok(int64_t W, int64_t Cin, int64_t Cout) {
  return (W == 16 || W == 32 || W == 64 || W == 56 || W == 112 || W == 128 || W == 96 || W == 48) && Cin % 2 == 0 && Cout % 2 == 0 && Cin >= 32 && Cout >= 32;
}
inline bool wxy_ok(int64_t H, int64_t W, int64_t Cin, int64_t Cout) { return wx_ok(W, Cin, Cout) && H % 2 == 0 && H >= 4; }
// (x,y,z): instantiated for the 128 -> 128 layers at W = 64 | 32 | 16 (cfg3), 112 | 56 (cfg4), 128, and for the 64 -> 64 layers of the
// auto-encoder (cfg5, F = 64) at W = 128 | 64 | 32 | 16
inline bool wxyz_ok(int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz) {
  if (!(kz == 3 && wxy_ok(H, W, Cin, Cout) && D % 2 == 0 && D >= 4)) return false;
  if (Cin == 128 && Cout == 128) return W == 64 || W == 32 || W == 16 || W == 112 || W == 56 || W == 128;
  if (Cin == 64 && Cout == 64) return W == 128 || W == 64 || W == 32 || W == 16;
  return false;
}
// `req` = the caller's algorithm request (the `algo` argument of df_conv_wgrad_algo, low 3 bits): 0: best available,
// 1: always the direct kernel, 2: at most Winograd-in-x, 3: (x,y) wherever it exists, 4: (x,y,z) wherever it exists.
// returns 0 direct | 1 Winograd in x | 2 Winograd in (x,y) | 3 Winograd in (x,y,z)
inline int wgrad_algo(int req, int64_t rows, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz) {
  if (req == 1) return 0;
  // (x,y,z) wherever it is instantiated, whatever the size: since its 16 workgroup types run as ONE launch (16 partial ranges fill the
  // chip) it wins at every level and batch -- [r3] batch 2 | 4 | 16, ms: 16x24x16 x 0.155 | 0.220 | 0.645, xy 0.143 | 0.189 | 0.492,
  // xyz 0.077 | 0.102 | 0.276;  32x48x32 x 0.64 | 1.20 | 3.70, xy 0.49 | 0.87 | 2.54, xyz 0.27 | 0.49 | 1.83 (the round-1 threshold of 4096
  // image rows sent the per-GPU batches of the strong-scaling runs, 16 / 8 GPUs = 2, to the x form: 1.4 ms of a 30 ms step)
  if ((req == 0 || req == 4) && wxyz_ok(D, H, W, Cin, Cout, kz)) return 3;
  // (x,y) pays a larger partial buffer than x: worth it from ~4096 image rows where (x,y,z) does not exist (2-D; Cin | Cout != 128)
  if (req != 2 && wxy_ok(H, W, Cin, Cout) && (rows >= 4096 || req >= 3)) return 2;
  return wx_ok(W, Cin, Cout) ? 1 : 0;
}

// ---- bf16x3 weight gradient (opt-in precision mode, see conv_bf16.hip) ---------------------------------------------------
// Same decomposition as wgrad_kernel (workgroup = (voxel range, (dz,dy) group), wave = 64x64 (ci,co) quadrant x 3 dx taps,
// operands straight from L1/L2), on v_mfma_f32_32x32x16_bf16: K = 16 consecutive voxels of ONE image row per instruction
// (half-wave 0: x0..x0+7, half-wave 1: x0+8..x0+15).  A lane's 8 k-values of a channel are 8 row positions, i.e. 8 separate
// coalesced loads -- so the x-1 / x / x+1 windows are just different registers of ONE set of 10 position loads (no
// alignment problem), converted to (hi, lo) bf16 in registers: 3 MFMAs (lo*hi, hi*lo, hi*hi) per 32x32x16 block.
// Loads run 3 steps ahead in a 4-deep register ring, continuous across rows.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = static_cast<__bf16>(x[e]);
    hi[e] = h;
    lo[e] = static_cast<__bf16>(x[e] - static_cast<float>(h));
  }
}

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }

template <int W16>
__global__ __launch_bounds__(kThreads, 1) void wgrad_bf16x3_kernel(const WgradArgs a) {
  constexpr int U = 4 / cgcd(W16, 4);          // rows per unrolled iteration: U*W16 steps, a multiple of the ring depth 4
  constexpr int NS = U * W16;
  constexpr int E = (3 + W16 - 1) / W16;       // extra rows the 3-step-ahead loads can reach into
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = a.nqi * a.nqj;
  const int qi = (wave % nq) / a.nqj, qj = (wave % nq) % a.nqj, sub = wave / nq;
  const int half = lane >> 5, r = lane & 31;

  const int nwg = a.nranges * a.ndzdy;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int range = wg / a.ndzdy, dzdy = wg % a.ndzdy;
  int dz, dy, pz = 0, py = 0, px = 0, dd = 0, cls = 0;
  if (a.up) {
    const int nd2 = a.ndzdy == 32 ? 4 : 2;
    cls = dzdy / nd2; dd = dzdy % nd2;
    pz = a.ndzdy == 32 ? (cls >> 2) & 1 : 0; py = (cls >> 1) & 1; px = cls & 1;
    dz = a.ndzdy == 32 ? (dd >> 1) + pz - 1 : 0;
    dy = (dd & 1) + py - 1;
  } else {
    dz = a.ndzdy == 9 ? dzdy / 3 - 1 : 0;
    dy = (a.ndzdy == 9 ? dzdy % 3 : dzdy) - 1;
  }
  const int gs = a.up ? 2 : 1;
  const int ci0 = blockIdx.y * 128 + qi * 64, co0 = blockIdx.z * 128 + qj * 64;
  if (ci0 >= a.Cin || co0 >= a.Cout) return;

  const int ppe = (a.pairs_per_range + a.nsub - 1) / a.nsub;
  const int row0 = 2 * (range * a.pairs_per_range + sub * ppe);
  int row1 = row0 + 2 * ppe;
  if (row1 > 2 * (range + 1) * a.pairs_per_range) row1 = 2 * (range + 1) * a.pairs_per_range;
  if (row1 > a.nrows) row1 = a.nrows;
  const int erange = range * a.nsub + sub;
  const int cia = ci0 + 2 * r, coa = co0 + 2 * r;
  const bool ci_ok = cia < a.Cin, co_ok = coa < a.Cout;
  const float* zb = a.zeros;

  struct Row { const float* xb; const float* gb; bool xz; };
  auto row_setup = [&](int row) -> Row {
    Row rw;
    const bool ok = row < row1;
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
    const int zs = z + dz, ys = y + dy;
    const bool xv = ok && zs >= 0 && zs < a.D && ys >= 0 && ys < a.H && ci_ok;
    const bool gv = ok && co_ok;
    const int64_t gvox = ((static_cast<int64_t>(b) * a.gD + (z * gs + pz)) * a.gH + (y * gs + py)) * a.gW + px;
    const int64_t xvox = ((static_cast<int64_t>(b) * a.D + zs) * a.H + ys) * a.W;
    rw.xb = xv ? a.x + (xvox + 8 * half) * a.Cin + cia : zb;            // the +8*half: half-wave 1 holds k = 8..15
    rw.gb = gv ? a.g + (gvox + 8 * half * gs) * a.Cout + coa : zb;
    rw.xz = !xv;
    return rw;
  };
  struct Raw { f32x2 x[10]; f32x2 g[8]; };
  auto issue = [&](const Row& rw, int k16, Raw& dst) {
    const int x0 = 16 * k16;
#pragma unroll
    for (int j = 0; j < 10; ++j) {            // positions u = x0 - 1 + j (+8 in half-wave 1)
      const float* p = rw.xb + static_cast<int64_t>(x0 - 1 + j) * a.Cin;
      if (k16 == 0 && j == 0) p = (half == 0 || rw.xz) ? zb : p;               // x = -1: left zero padding
      if (k16 == W16 - 1 && j == 9) p = (half == 1 || rw.xz) ? zb : p;         // x = W: right zero padding
      dst.x[j] = *reinterpret_cast<const f32x2*>(p);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      dst.g[j] = *reinterpret_cast<const f32x2*>(rw.gb + static_cast<int64_t>(x0 + j) * gs * a.Cout);
  };

  f32x16 acc[3][2][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][s][t][e] = 0.f;
  f32x2 bsum = {0.f, 0.f};
  const bool do_bias = a.want_bias && (a.up ? dd == 0 : dzdy == a.ndzdy / 2) && blockIdx.y == 0 && qi == 0;

  auto compute = [&](const Raw& rw) {
    bf16x8 bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rw.g[e][t];
      split8(v, bh[t], bl[t]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsum[0] += rw.g[e][0]; bsum[1] += rw.g[e][1]; }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rw.x[d + e][s];
        bf16x8 ah, al;
        split8(v, ah, al);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[d][s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[d][s][t], 0, 0, 0);
          acc[d][s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[d][s][t], 0, 0, 0);
          acc[d][s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[d][s][t], 0, 0, 0);
        }
      }
  };

  Row rows[U + E];
#pragma unroll
  for (int i = 0; i < U + E; ++i) rows[i] = row_setup(row0 + i);
  Raw ring[4];
#pragma unroll
  for (int sidx = 0; sidx < 3; ++sidx) issue(rows[sidx / W16], sidx % W16, ring[sidx]);

  for (int rbase = row0; rbase < row1; rbase += U) {
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      issue(rows[(sidx + 3) / W16], (sidx + 3) % W16, ring[(sidx + 3) & 3]);
      __builtin_amdgcn_sched_barrier(0);
      compute(ring[sidx & 3]);
    }
#pragma unroll
    for (int i = 0; i < E; ++i) rows[i] = rows[i + U];
#pragma unroll
    for (int i = E; i < U + E; ++i) rows[i] = row_setup(rbase + U + i);
  }

  const int taps = a.ndzdy * 3;
  float* P = a.partial + static_cast<int64_t>(erange) * taps * a.Cinp * a.Coutp;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int tap = dzdy * 3 + d;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ci = ci0 + 2 * i + s, co = co0 + 2 * r + t;
          P[(static_cast<int64_t>(tap) * a.Cinp + ci) * a.Coutp + co] = acc[d][s][t][e];
        }
  }
  if (do_bias) {
    bsum[0] += __shfl_xor(bsum[0], 32, 64);
    bsum[1] += __shfl_xor(bsum[1], 32, 64);
    if (half == 0) {
      const int ncls = a.up ? (a.ndzdy == 32 ? 8 : 4) : 1;
      float* pb = a.bpartial + (static_cast<int64_t>(erange) * ncls + cls) * a.Coutp + co0 + 2 * r;
      pb[0] = bsum[0]; pb[1] = bsum[1];
    }
  }
}

inline bool wgrad_bf16x3_ok(int64_t W, int64_t Cin, int64_t Cout) {
  // W = 112 (cfg4) is NOT listed: its 28-step unrolled variant spilt 720 B of scratch per lane and ran 148 ms per launch on
  // MI355X; those rows take the fp32 Winograd weight gradients instead (more accurate, ~40x faster)
  return (W == 16 || W == 32 || W == 64 || W == 96) && Cin % 2 == 0 && Cout % 2 == 0 && Cin >= 16 && Cout >= 16;
}
template <typename A>
inline void launch_wgrad_bf16x3(int64_t W, dim3 grid, hipStream_t s, const A& a) {
  switch (W / 16) {
    case 1: hipLaunchKernelGGL((wgrad_bf16x3_kernel<1>), grid, dim3(kThreads), 0, s, a); break;
    case 2: hipLaunchKernelGGL((wgrad_bf16x3_kernel<2>), grid, dim3(kThreads), 0, s, a); break;
    case 4: hipLaunchKernelGGL((wgrad_bf16x3_kernel<4>), grid, dim3(kThreads), 0, s, a); break;
    default: hipLaunchKernelGGL((wgrad_bf16x3_kernel<6>), grid, dim3(kThreads), 0, s, a); break;
  }
}

// gw[tap][ci][co] = sum_range partial[range][tap][ci][co]  (fixed order);  gb likewise
__global__ __launch_bounds__(kThreads) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                const float* __restrict__ bpartial,
                                                                float* __restrict__ gw, float* __restrict__ gb,
                                                                int nranges, int taps, int Cin, int Cout, int Cinp,
                                                                int Coutp) {
  // [r6] 32 consecutive elements x 8 range groups per workgroup, group sums combined in a fixed order through LDS (see wgrad_wx_reduce_kernel);
  // the workgroups past the weight elements reduce the bias
  __shared__ float sP[8][32];
  const int64_t total = static_cast<int64_t>(taps) * Cin * Cout;
  const int64_t nwg = (total + 31) / 32;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float acc = 0.f;
  bool ok;
  float* dst;
  if (static_cast<int64_t>(blockIdx.x) >= nwg) {
    const int co = static_cast<int>(blockIdx.x - nwg) * 32 + el;
    ok = gb && co < Cout;
    dst = ok ? gb + co : nullptr;
    if (ok)
      for (int rg = grp; rg < nranges; rg += 8) acc += bpartial[static_cast<int64_t>(rg) * Coutp + co];
  } else {
    const int64_t pstride = static_cast<int64_t>(taps) * Cinp * Coutp;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 32 + el;
    ok = i < total;
    dst = ok ? gw + i : nullptr;
    if (ok) {
      const int co = static_cast<int>(i % Cout);
      const int64_t t2 = i / Cout;
      const int ci = static_cast<int>(t2 % Cin);
      const int tap = static_cast<int>(t2 / Cin);
      const float* p = partial + (static_cast<int64_t>(tap) * Cinp + ci) * Coutp + co;
      for (int rg = grp; rg < nranges; rg += 8) acc += p[rg * pstride];
    }
  }
  sP[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && ok) {
    float t = sP[0][el];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += sP[g][el];
    *dst = t;
  }
}

// ---- small-N weight gradient (Cout <= 4: the generator's last layer) on the vector ALU ---------------------------
// thread = input channel (64 per wave), all NTAP x CO accumulators in registers.  A wave walks whole image rows
// along x; rows are laid end to end with >= 1 zero position between them (Wp = round_up(W + 1, 8)), so the
// x-1 / x+1 neighbours at a row's ends are zeros WITHOUT any masking.  Per voxel step: 9 | 3 coalesced 256-byte
// loads (one per (dz,dy), 5-6 steps ahead in an 8-deep ring), the wave-uniform gradient record via scalar loads,
// 27*CO | 9*CO FMAs.  Out-of-image source rows and the padding read a zeroed workspace slot (address select on an
// opaque scalar offset: no branches, no post-load selects -> counted vmcnt waits only).
struct SmallWgradArgs {
  const float* x;
  const float* g;
  float* partial;      // [nstreams][taps][Cin][CO]
  float* bpartial;     // [nstreams][CO]
  const float* zeros;
  int B, D, H, W, Cin, Cout;
  int Wp, nrows, nstreams, rows_per, ncib;
};

template <int KZ, int CO>
__global__ __launch_bounds__(kThreads, 1) void wgrad_small_n_kernel(const SmallWgradArgs a) {
  constexpr int NDZDY = KZ * 3, NTAP = NDZDY * 3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;
  const int cib = gw % a.ncib, stream = gw / a.ncib;
  const int ci = cib * 64 + lane;
  const int r0 = stream * a.rows_per;
  int r1 = r0 + a.rows_per;
  if (r1 > a.nrows) r1 = a.nrows;
  const int64_t zoff_x = a.zeros - a.x, zoff_g = a.zeros - a.g;

  struct Row { int64_t xo[NDZDY]; int64_t go; };
  auto row_setup = [&](int row) -> Row {      // everything here is wave-uniform
    Row rw;
    const bool ok = row < r1;
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
    rw.go = ok ? (static_cast<int64_t>(row) * a.W) * a.Cout : -1;
#pragma unroll
    for (int k = 0; k < NDZDY; ++k) {
      const int zs = z + (KZ == 3 ? k / 3 - 1 : 0), ys = y + k % 3 - 1;
      const bool v = ok && zs >= 0 && zs < a.D && ys >= 0 && ys < a.H;
      rw.xo[k] = v ? (((static_cast<int64_t>(b) * a.D + zs) * a.H + ys) * a.W) * a.Cin + cib * 64 : -1;
    }
    return rw;
  };
  auto load_x = [&](const Row& rw, int k, int pos) -> float {
    int64_t off = (rw.xo[k] >= 0 && pos < a.W) ? rw.xo[k] + static_cast<int64_t>(pos) * a.Cin : zoff_x;
    asm("" : "+s"(off));
    return a.x[off + lane];
  };
  struct GRec { float v[CO]; };
  auto load_g = [&](const Row& rw, int pos) -> GRec {
    int64_t off = (rw.go >= 0 && pos < a.W) ? rw.go + static_cast<int64_t>(pos) * a.Cout : zoff_g;
    asm("" : "+s"(off));
    GRec r;
#pragma unroll
    for (int c = 0; c < CO; ++c) r.v[c] = a.g[off + c];
    return r;
  };

  float acc[NTAP][CO];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[t][c] = 0.f;

  float xr[NDZDY][8];
  GRec gr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int k = 0; k < NDZDY; ++k) xr[k][i] = 0.f;
#pragma unroll
    for (int c = 0; c < CO; ++c) gr[i].v[c] = 0.f;
  }
  Row cur = row_setup(r0);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k < NDZDY; ++k) xr[k][i] = load_x(cur, k, i);
#pragma unroll
  for (int i = 0; i < 5; ++i) gr[i] = load_g(cur, i);

  float bsum[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) bsum[c] = 0.f;
  auto step = [&](int u) {
    __builtin_amdgcn_sched_barrier(0);
    const GRec g = gr[u];
#pragma unroll
    for (int c = 0; c < CO; ++c) bsum[c] += g.v[c];
#pragma unroll
    for (int k = 0; k < NDZDY; ++k)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float xv = xr[k][(u + 7 + d) & 7];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[k * 3 + d][c] = fmaf(xv, g.v[c], acc[k * 3 + d][c]);
      }
  };

  for (int row = r0; row < r1; ++row) {
    const Row nxt = row_setup(row + 1);
    for (int x0 = 0; x0 < a.Wp - 8; x0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int k = 0; k < NDZDY; ++k) xr[k][(u + 6) & 7] = load_x(cur, k, x0 + u + 6);
        gr[(u + 5) & 7] = load_g(cur, x0 + u + 5);
        step(u);
      }
    }
    {
      const int x0 = a.Wp - 8;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int k = 0; k < NDZDY; ++k) xr[k][(u + 6) & 7] = u < 2 ? load_x(cur, k, x0 + u + 6) : load_x(nxt, k, u - 2);
        gr[(u + 5) & 7] = u < 3 ? load_g(cur, x0 + u + 5) : load_g(nxt, u - 3);
        step(u);
      }
    }
    cur = nxt;
  }

  float* P = a.partial + (static_cast<int64_t>(stream) * NTAP * a.Cin + ci) * CO;
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int c = 0; c < CO; ++c) P[static_cast<int64_t>(t) * a.Cin * CO + c] = acc[t][c];
  if (cib == 0 && lane == 0) {
#pragma unroll
    for (int c = 0; c < CO; ++c) a.bpartial[stream * CO + c] = bsum[c];
  }
}

// gw[tap][ci][co] = sum_stream partial[stream][tap][ci][co]  (fixed order: workgroup = 32 consecutive elements x 8 interleaved stream
// groups, the 8 group sums combined in order through LDS)
__global__ __launch_bounds__(kThreads) void wgrad_small_reduce_kernel(const float* __restrict__ partial,
                                                                      const float* __restrict__ bpartial,
                                                                      float* __restrict__ gw, float* __restrict__ gb,
                                                                      int nstreams, int64_t per_stream, int CO,
                                                                      int Cout) {
  __shared__ float sP[8][32];
  __shared__ float sB[8 * 32];
  // (bias: Cout <= 4 columns over up to 2048 stream rows -- grouped, not one serial walk by workgroup 0: that alone was 126 us per call)
  if (gb && blockIdx.x == 0) bias_reduce_cols(bpartial, gb, nstreams, Cout, CO, sB, 0);
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 32 + el;
  float acc = 0.f;
  if (i < per_stream)
    for (int sidx = grp; sidx < nstreams; sidx += 8) acc += partial[sidx * per_stream + i];
  sP[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && i < per_stream) {
    float t = sP[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += sP[k][el];
    const int co = static_cast<int>(i % CO);
    if (co < Cout) gw[(i / CO) * Cout + co] = t;
  }
}

// ---- small-N weight gradient on the matrix cores (3-D, Cin = 128, Cout <= 4: the generator's last layer) -----------------------
// The 27 taps are folded into the M side of ONE GEMM:  P[(tap, co), ci] = sum_v G'[v, (tap, co)] * x[v, ci]  with
// G'[v, (tap, co)] = g[v - (tap - 1), co]  (zero outside the image) -- M = 27*Cout <= 96|128 rows, N = 128 input channels, K = voxels.
// x is then a plain LINEAR stream (no halo, read exactly once: one 1 KB buffer load per wave feeds 12|16 v_mfma_f32_32x32x2), and
// the shifted copies of the 3|4-channel gradient are gathered from a 9-row LDS tile private to the wave (x-halo of zeros, zero rows
// for SAME padding, double-buffered per image row: no workgroup barriers at all).  B operand columns are permuted (lane l's float4
// = channels 4*(l%32)..+3 feed the four N blocks), which makes the partial-sum stores float4-contiguous.
// One wave per SIMD (192|256 accumulator registers); latency is covered by an 8-step ring of x loads.
struct ThinWgradArgs {
  const float* x;      // the WIDE tensor (NJ*32 channels), streamed linearly
  const float* g;      // the THIN tensor (CT <= 4 channels), gathered from the per-wave LDS tile
  float* partial;      // [nstreams][27*CT][NJ*32]
  float* bpartial;     // [nstreams][CT]  (SWAP: [nstreams][NJ*32])
  int B, D, H, W;
  int nrows, nstreams, rows_per, RS;      // RS: LDS row stride in floats = W*CT + 8
  unsigned x_bytes_lo, x_bytes_hi;        // size of the wide tensor in bytes
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t thin_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// CT: channels of the thin tensor; NJ: the wide tensor has NJ*32 channels (lane l holds channels NJ*(l%32)..+NJ-1 of voxel l/32);
// NP: 64-lane passes per thin row (W*CT/4 float4 pieces).  SWAP = false: wide = x (Cin), thin = g (Cout <= 4), the gradient of a
// F -> 1..4 layer; SWAP = true: wide = g (Cout), thin = x (Cin <= 4), the gradient of a 1..4 -> F layer (gather offsets mirrored,
// bias gradient = column sums of the wide stream).
template <int CT, int NJ, bool SWAP, int NP>
__global__ __launch_bounds__(kThreads, 1) void wgrad_thin_mfma_kernel(const ThinWgradArgs a) {
  constexpr int NM = 27 * CT, MB = (NM + 31) / 32, MBC = (13 * CT) / 32;      // MBC: the M block that holds the centre tap
  constexpr int NCH = NJ * 32;

  typedef float vt __attribute__((ext_vector_type(NJ)));
  extern __shared__ __attribute__((aligned(16))) float smem_thin[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wave;
  const int RS = a.RS, W = a.W, WC = W * CT;
  float* sG = smem_thin + wave * 19 * RS;          // [2 buffers][9 (tz, ty) rows][RS] + one zero row
  const int r0 = stream * a.rows_per;
  int r1 = r0 + a.rows_per;
  if (r1 > a.nrows) r1 = a.nrows;
  for (int i = lane; i < 19 * RS; i += 64) sG[i] = 0.f;
  constexpr int NBIAS = SWAP ? NCH : CT;
  if (r0 >= r1) {      // (only when nrows is not a multiple of the stream count) -- still publish zero partials
    for (int i = lane; i < NM * NCH; i += 64) a.partial[static_cast<int64_t>(stream) * NM * NCH + i] = 0.f;
    for (int i = lane; i < NBIAS; i += 64) a.bpartial[stream * NBIAS + i] = 0.f;
    return;
  }

  // ---- thin rows of image row `row` -> registers -> LDS buffer ----------------------------------------------------------------------
  const int nl4 = WC / 4;      // float4 pieces per thin row (<= 64 * NP)
  f32x4 gq[9][NP];
  auto load_g = [&](int row) {
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int zs = SWAP ? z + (k / 3 - 1) : z - (k / 3 - 1), ys = SWAP ? y + (k % 3 - 1) : y - (k % 3 - 1);
      const bool ok = zs >= 0 && zs < a.D && ys >= 0 && ys < a.H;      // wave-uniform
      const int zc = ok ? zs : z, yc = ok ? ys : y;
      const float* src = a.g + ((static_cast<int64_t>(b) * a.D + zc) * a.H + yc) * WC;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (lane + pp * 64 < nl4) v = *reinterpret_cast<const f32x4*>(src + (lane + pp * 64) * 4);
        gq[k][pp] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto store_g = [&](int buf) {
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
      if (lane + pp * 64 < nl4) {
#pragma unroll
        for (int k = 0; k < 9; ++k) *reinterpret_cast<f32x4*>(sG + (buf * 9 + k) * RS + 4 + (lane + pp * 64) * 4) = gq[k][pp];
      }
  };

  // ---- A operand gather: lane (m = l % 32 of block mb, voxel kk = l / 32 of the pair) --------------------------------------------
  const int kk = lane >> 5;
  int aoff[MB], adb[MB];      // byte offset inside buffer 0; byte distance to buffer 1 (0 for the zero row)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 32 + (lane & 31);
    if (m < NM) {
      const int tap = m / CT, co = m % CT;
      const int t9 = tap / 3, tx = tap % 3;
      aoff[mb] = (t9 * RS + 4 + (SWAP ? kk + (tx - 1) : kk - (tx - 1)) * CT + co) * 4;
      adb[mb] = 9 * RS * 4;
    } else {
      aoff[mb] = (18 * RS + 4) * 4;
      adb[mb] = 0;
    }
  }
  const char* sGb = reinterpret_cast<const char*>(sG);
  auto gather = [&](int buf, int pair, float (&out)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      out[mb] = *reinterpret_cast<const float*>(sGb + aoff[mb] + buf * adb[mb] + pair * (2 * CT * 4));
  };

  // ---- wide stream -------------------------------------------------------------------------------------------------------------------
  const int64_t xoff0 = static_cast<int64_t>(r0) * W * NCH;      // floats
  const uint64_t xbytes = (static_cast<uint64_t>(a.x_bytes_hi) << 32) | a.x_bytes_lo;
  const uint64_t remain = xbytes - static_cast<uint64_t>(xoff0) * 4u;
  const __amdgpu_buffer_rsrc_t xsrd = thin_srd(a.x + xoff0, remain > 0xffffffffull ? 0xffffffffu : static_cast<unsigned>(remain));
  const unsigned lanev = static_cast<unsigned>(lane) * (NJ * 4u);
  auto load_x = [&](unsigned step) -> vt {       // step = voxel pair index within the stream; past the tensor's end reads zeros
    if constexpr (NJ == 4) return __builtin_bit_cast(vt, __builtin_amdgcn_raw_buffer_load_b128(xsrd, lanev, step * (2u * NCH * 4u), 0));
    else return __builtin_bit_cast(vt, __builtin_amdgcn_raw_buffer_load_b64(xsrd, lanev, step * (2u * NCH * 4u), 0));
  };

  f32x16 acc[MB][NJ];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][j][r] = 0.f;
  float bs = 0.f;
  vt bsw;
#pragma unroll
  for (int j = 0; j < NJ; ++j) bsw[j] = 0.f;

  vt xr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xr[i] = load_x(i);
  load_g(r0);
  store_g(0);
  float an[MB];
  gather(0, 0, an);

  const int ngrp = W >> 4;      // groups of 8 voxel pairs per image row
  unsigned step = 0;
  int buf = 0;
  for (int row = r0; row < r1; ++row) {
    const bool more = row + 1 < r1;
    if (more) load_g(row + 1);
    for (int grp = 0; grp < ngrp; ++grp) {
      const bool lastg = grp + 1 == ngrp;
      if (lastg && more) store_g(buf ^ 1);      // the row's last pair prefetches the first gather of the next row
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float ac[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) ac[mb] = an[mb];
        if (i < 7) gather(buf, grp * 8 + i + 1, an);
        else if (!lastg) gather(buf, grp * 8 + 8, an);
        else gather(buf ^ 1, 0, an);
        const vt xv = xr[i];
        xr[i] = load_x(step + 8);
        ++step;
        if (SWAP) bsw += xv; else bs += ac[MBC];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mb], xv[j], acc[mb][j], 0, 0, 0);
      }
    }
    buf ^= 1;
  }

  // ---- partial sums: row m of block mb lives in register r of lanes with l / 32 == ((m % 8) / 4) --------------------------------------
  float* P = a.partial + static_cast<int64_t>(stream) * NM * NCH + (lane & 31) * NJ;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb * 32 + (r >> 2) * 8 + kk * 4 + (r & 3);
      if (m < NM) {
        vt o;
#pragma unroll
        for (int j = 0; j < NJ; ++j) o[j] = acc[mb][j][r];
        *reinterpret_cast<vt*>(P + m * NCH) = o;
      }
    }
  if (SWAP) {      // bias gradient = column sums of the wide stream (two voxel parities per lane pair)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float t = bsw[j] + __shfl_xor(bsw[j], 32);
      if (lane < 32) a.bpartial[stream * NCH + lane * NJ + j] = t;
    }
  } else {         // lanes (13*CT + c) % 32 of both halves hold the two voxel-parity sums of thin[., c]
    const float other = __shfl_xor(bs, 32);
    const int l0 = (13 * CT) & 31;
    if (lane >= l0 && lane < l0 + CT) a.bpartial[stream * CT + (lane - l0)] = bs + other;
  }
}

// gw[tap][ci][co] = sum_stream partial[stream][tap*CT + c][ch]  (fixed order: 8 interleaved stream groups, combined in order);
// SWAP = 0: (ci, co) = (ch, c);  SWAP = 1: (ci, co) = (c, ch)
__global__ __launch_bounds__(kThreads) void wgrad_thin_reduce_kernel(const float* __restrict__ partial,
                                                                     const float* __restrict__ bpartial,
                                                                     float* __restrict__ gw, float* __restrict__ gb,
                                                                     int nstreams, int NM, int CT, int NCH, int swap) {
  __shared__ float sP[8][32];
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int nq = NCH >> 5;
  const int m = blockIdx.x / nq, ch = (blockIdx.x % nq) * 32 + el;
  float acc = 0.f;
  for (int sidx = grp; sidx < nstreams; sidx += 8) acc += partial[(static_cast<int64_t>(sidx) * NM + m) * NCH + ch];
  sP[grp][el] = acc;
  __syncthreads();
  if (grp == 0) {
    float t = sP[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += sP[k][el];
    if (swap) gw[static_cast<int64_t>(m) * NCH + ch] = t;
    else gw[(static_cast<int64_t>(m / CT) * NCH + ch) * CT + (m % CT)] = t;
  }
  const int nbias = swap ? NCH : CT;
  if (gb && static_cast<int>(blockIdx.x) * 32 < nbias) bias_reduce_cols(bpartial, gb, nstreams, nbias, nbias, &sP[0][0], blockIdx.x);
}

// wide channels 64 | 128, thin <= 4, rows of 16-voxel groups, at most two 64-lane passes per thin row, LDS 4 waves x 19 rows x
// (W*CT + 8) floats <= 150 KB (above 64 KB the launch opts in)
inline bool thin_mfma_ok(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cwide, int64_t Cthin, int kz) {
  return kz == 3 && (Cwide == 128 || Cwide == 64) && Cthin >= 1 && Cthin <= 4 && W % 16 == 0 && (W * Cthin) % 16 == 0 && W * Cthin <= 480 &&
         B * D * H >= 4 && B * D * H * W * Cwide * 4 < (1LL << 44);
}
// instantiated combinations (others fall back to the generic kernels)
inline bool thin_mfma_inst(int64_t W, int64_t Cwide, int64_t Cthin, bool swap) {
  const bool two = W * Cthin > 256;
  if (!swap && Cwide == 128 && !two) return true;                 // CT 1..4
  return Cthin == 3;                                              // every (NJ, SWAP, NP) for three thin channels
}
// streams (waves) of the matrix-core thin weight gradient and its workspace floats
static int thin_wgrad_streams(int64_t B, int64_t D, int64_t H) {
  int64_t ns = 4 * df::kCUs;
  const int64_t nrows = B * D * H;
  if (ns > nrows) ns = nrows / 4 * 4;
  return (int)ns;
}
static int64_t thin_wgrad_ws_floats(int64_t B, int64_t D, int64_t H, int64_t Cw, int64_t Ct, bool swap) {
  return static_cast<int64_t>(thin_wgrad_streams(B, D, H)) * (27 * Ct * Cw + (swap ? Cw : Ct));
}

inline bool small_n_ok(int64_t Cin, int64_t Cout) { return Cout <= 4 && Cin % 64 == 0; }

struct SmallPlan { int nrows, ncib, nstreams, rows_per, taps, CO; int64_t partial_elems; };
SmallPlan make_small_plan(int64_t B, int64_t D, int64_t H, int64_t Cin, int64_t Cout, int kz) {
  SmallPlan p;
  p.nrows = (int)(B * D * H);
  p.ncib = (int)(Cin / 64);
  int ns = kSmallStreams / p.ncib;
  if (ns > p.nrows) ns = p.nrows;
  ns = (ns * p.ncib + 3) / 4 * 4 / p.ncib;              // whole workgroups of 4 waves
  if (ns < 1) ns = 1;
  while ((ns * p.ncib) % 4 != 0) ++ns;
  p.nstreams = ns;
  p.rows_per = (p.nrows + ns - 1) / ns;
  p.taps = kz == 3 ? 27 : 9;
  p.CO = (int)Cout;
  p.partial_elems = static_cast<int64_t>(ns) * p.taps * Cin * p.CO + static_cast<int64_t>(ns) * p.CO;
  return p;
}

// up mode: gw[kz][ky][kx] = sum over the (class, delta) pairs whose coarse offset the original tap k feeds (per axis:
// k=0 <- (p0,d0),(p1,d0); k=1 <- (p0,d1),(p1,d0); k=2 <- (p0,d1),(p1,d1)); partial slot = combo*3 + (deltax + px).
__global__ __launch_bounds__(kThreads) void wgrad_up_reduce_kernel(const float* __restrict__ partial,
                                                                   const float* __restrict__ bpartial,
                                                                   float* __restrict__ gw, float* __restrict__ gb,
                                                                   int nranges, int kz, int Cin, int Cout, int Cinp,
                                                                   int Coutp) {
  // workgroup = 32 consecutive output elements x 8 range groups (fixed-order combine through LDS: deterministic)
  __shared__ float sP[8][32];
  const int taps = kz * 9, ncombo = kz == 3 ? 32 : 8, nd2 = kz == 3 ? 4 : 2, ncls = kz == 3 ? 8 : 4;
  const int64_t total = static_cast<int64_t>(taps) * Cin * Cout;
  const int64_t slot = static_cast<int64_t>(Cinp) * Coutp;
  const int64_t pstride = static_cast<int64_t>(ncombo) * 3 * slot;
  const int P[3][2] = {{0, 1}, {0, 1}, {0, 1}}, Dl[3][2] = {{0, 0}, {1, 0}, {1, 1}};   // k -> (p, delta) pairs
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 32 + el;
  const bool ok = i < total;
  float acc = 0.f;
  if (ok) {
    const int co = static_cast<int>(i % Cout);
    const int64_t t2 = i / Cout;
    const int ci = static_cast<int>(t2 % Cin);
    const int tap = static_cast<int>(t2 / Cin);
    const int kx = tap % 3, ky = (tap / 3) % 3, kzz = tap / 9;
    for (int rg = grp; rg < nranges; rg += 8) {
      const float* pr = partial + rg * pstride + static_cast<int64_t>(ci) * Coutp + co;
      for (int az = 0; az < (kz == 3 ? 2 : 1); ++az)
        for (int ay = 0; ay < 2; ++ay)
          for (int ax = 0; ax < 2; ++ax) {
            const int pz = kz == 3 ? P[kzz][az] : 0, dzl = kz == 3 ? Dl[kzz][az] : 0;
            const int py = P[ky][ay], dyl = Dl[ky][ay], px = P[kx][ax], dxl = Dl[kx][ax];
            const int cls = (pz << 2) | (py << 1) | px;
            const int combo = cls * nd2 + (kz == 3 ? (dzl << 1) | dyl : dyl);
            acc += pr[(static_cast<int64_t>(combo) * 3 + (dxl + px)) * slot];
          }
    }
  }
  sP[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && ok) {
    float t = sP[0][el];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += sP[g][el];
    gw[i] = t;
  }
  if (gb && static_cast<int>(blockIdx.x) * 32 < Cout) bias_reduce_cols(bpartial, gb, nranges * ncls, Cout, Coutp, &sP[0][0], blockIdx.x);
}

struct Plan {
  int nrows, npairs, nranges, ppr, Cinp, Coutp, taps, ndzdy, nqi, nqj, nsub;
  int64_t partial_elems, bpartial_elems;
};

Plan make_plan(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int algo = 0, int ranges = 0) {
  Plan p;
  p.nrows = (int)(algo == 3 ? B * (D / 2) * (H / 2) : algo == 2 ? B * D * (H / 2) : B * D * H);       // image rows, or tile rows
  p.npairs = (p.nrows + 1) / 2;
  p.ndzdy = algo == 3 ? 16 : algo == 2 ? (kz == 3 ? 12 : 4) : (kz == 3 ? 9 : 3);  // workgroup types: (dz,dy), (dz,xi_y) or (xi_z,xi_y)
  p.taps = kz == 3 ? 27 : 9;
  // The kernel runs ONE workgroup per CU (192-256 accumulator registers per lane): a grid that is not a multiple
  // of 256 leaves most of the chip idle in its last round.  256 equal voxel ranges x (9 | 3 | 12 | 4) groups is
  // a whole number of rounds; smaller problems get one range per row pair.
  (void)W;
  // (the (x,y) Winograd kernel has 12 | 4 workgroup types and 16 partial slots per dz: 128 ranges = 6 | 2 whole rounds and half
  //  the partial traffic of 256)
  // (small problems: the fixed-order reduce of the partial sums costs as much as the products -- 64 ranges measured best below
  //  4096 row pairs: 0.45 -> 0.22 ms at 16 x 8x12x8, 0.84 -> 0.67 ms at 16 x 16x24x16)
  // ... but never fewer ranges than it takes to put one workgroup on every CU: a LAUNCH has ndzdy (direct, x), ndzdy / 2 ((x,y): two
  // launches) or 16 ((x,y,z): one launch) workgroup types -- in 2-D that is only 3 | 2 types, and 64 ranges left half of the chip
  // idle (2-D 128x96, batch 64: Winograd-(x,y) 1.59 -> 0.92 ms with 128 ranges)
  const int per_launch = algo == 3 ? 16 : algo == 2 ? p.ndzdy / 2 : p.ndzdy;
  const int fill = (256 + per_launch - 1) / per_launch;
  // ((x,y,z), one launch of 16 types: 16 ranges = one whole round below 4096 row pairs -- 0.45 -> 0.30 ms at 16 x 16x24x16)
  // ((x,y,z) above 4096 tile-row pairs [r3], top level B = 16: 32 ranges 14.96 ms, 64 14.47, 128 14.66, 256 14.96 -- and half the partial
  //  buffer / reduce of 128)
  int heur = p.npairs < 4096 ? (algo == 3 ? 16 : 64) : (algo == 3 ? 64 : algo == 2 ? 128 : kMaxRanges);
  if (algo >= 2 && heur < fill) heur = fill <= kMaxRanges ? fill : kMaxRanges;      // (the direct / x kernels measured slower with more ranges)
  p.Cinp = (int)(ceil_div(Cin, 64) * 64);
  p.Coutp = (int)(ceil_div(Cout, 64) * 64);
  p.nqi = Cin <= 64 ? 1 : 2; p.nqj = Cout <= 64 ? 1 : 2;
  p.nsub = 4 / (p.nqi * p.nqj);
  // 64 -> 64 layers ((x,y,z) form, cfg5): the four waves of a workgroup split its voxel range, i.e. write nsub = 4 partials per range -- 128
  // ranges were 512 partials = 537 MB for the fixed-order reduce (0.38 ms per layer, 9.5 ms per AE step); 32 x 16 types still fill the chip twice
  if (algo == 3 && p.nsub > 1 && heur > 128 / p.nsub) heur = 128 / p.nsub < fill ? fill : 128 / p.nsub;
  const int maxr = (ranges > 0 && ranges <= kMaxRanges) ? ranges : heur;
  int nr = p.npairs >= maxr ? maxr : p.npairs;
  p.ppr = (p.npairs + nr - 1) / nr;
  p.nranges = (p.npairs + p.ppr - 1) / p.ppr;
  const int slots = algo == 0 ? p.taps : p.ndzdy * 4;             // the Winograd kernels write 4 xi_x slots per group
  p.partial_elems = static_cast<int64_t>(p.nranges) * p.nsub * slots * p.Cinp * p.Coutp;
  p.bpartial_elems = static_cast<int64_t>(p.nranges) * p.nsub * p.Coutp;
  return p;
}

// ---- row lengths without an (x,y,z) Winograd instantiation (W = 28, 14: levels of cfg4's 112-wide grid) ----------------------------------
// gW of a SAME conv does not change when both operands get zero columns appended (the added gradient columns are zero, and the added input
// columns are what SAME padding supplies anyway): copy x and gy into rows of the next instantiated length and run that kernel -- two
// copies of a low-resolution level against a 2-3x faster weight gradient.  Returns the padded row length, 0 = does not apply.
inline int64_t wxyz_padded_w(int req, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz) {
  if (!(req == 0 || req == 4) || kz != 3 || W < 12) return 0;
  if (wxyz_ok(D, H, W, Cin, Cout, kz)) return 0;
  static const int64_t kLens[] = {16, 32, 56, 64, 112, 128};
  for (int64_t L : kLens)
    if (L > W && 8 * (L - W) <= L + 7 && wxyz_ok(D, H, L, Cin, Cout, kz)) return L;      // at most 1/8 of the row is padding
  return 0;
}
inline int64_t up256(int64_t n) { return (n + 255) / 256 * 256; }
inline int64_t wxyz_pad_bytes(int64_t B, int64_t D, int64_t H, int64_t Wp, int64_t Cin, int64_t Cout) {
  return up256(B * D * H * Wp * Cin * 4) + up256(B * D * H * Wp * Cout * 4);
}

// dst[row][0..Wp) = src[row][0..W) followed by zeros; one float4 of channels per thread
__global__ __launch_bounds__(kThreads) void pad_rows_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t nrows, int W, int Wp, int C4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int64_t per = static_cast<int64_t>(Wp) * C4;
  if (i >= nrows * per) return;
  const int64_t row = i / per;
  const int r = static_cast<int>(i - row * per);
  const int xx = r / C4;
  dst[i] = xx < W ? src[(row * W + xx) * C4 + (r - xx * C4)] : f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace

extern "C" {

#ifdef DF_TUNING
void df_debug_set_wgrad(int v) { g_wgrad_dbg = v; }
#endif

int64_t df_conv_wgrad_workspace_bytes(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (small_n_ok(Cin, Cout)) {
    const SmallPlan sp = make_small_plan(B, D, H, Cin, Cout, kz);
    int64_t fl = sp.partial_elems;
    if (thin_mfma_ok(B, D, H, W, Cin, Cout, kz)) {
      const int64_t t = thin_wgrad_ws_floats(B, D, H, Cin, Cout, false);
      if (t > fl) fl = t;
    }
    return fl * static_cast<int64_t>(sizeof(float)) + kZeroBytes;
  }
  if (const int64_t Wp = wxyz_padded_w(0, D, H, W, Cin, Cout, kz))      // zero-padded rows + the workspace of the padded shape
    return df_conv_wgrad_workspace_bytes(B, D, H, Wp, Cin, Cout, kz) + 256 + wxyz_pad_bytes(B, D, H, Wp, Cin, Cout);
  int64_t best = 0;
  if (Cin <= 4 && Cout >= 64 && thin_mfma_ok(B, D, H, W, Cout, Cin, kz))
    best = thin_wgrad_ws_floats(B, D, H, Cout, Cin, true) * static_cast<int64_t>(sizeof(float));
  for (int algo = 0; algo <= 3; ++algo) {       // the launch may fall back (operand alignment), so size for the largest
    if ((algo == 1 && !wx_ok(W, Cin, Cout)) || (algo == 2 && !wxy_ok(H, W, Cin, Cout)) || (algo == 3 && !wxyz_ok(D, H, W, Cin, Cout, kz))) continue;
    const Plan p = make_plan(B, D, H, W, Cin, Cout, kz, algo);
    const int64_t n = (p.partial_elems + p.bpartial_elems) * static_cast<int64_t>(sizeof(float));
    if (n > best) best = n;
  }
  return best + zero_row_bytes(W, Cin, Cout);
}

// "not applicable, take the generic path": a value no hipError_t (> 0) and no DF_E* code (-1..-4) can take, so a real launch
// failure of the thin kernel is never mistaken for the fall-back signal
constexpr int kThinNotApplicable = -1000;
static int launch_thin_wgrad(const float* wide, const float* thin, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                             int64_t Cw, int64_t Ct, bool swap, void* workspace, df_stream_t stream) {
  ThinWgradArgs ta;
  const int ns = thin_wgrad_streams(B, D, H);
  const int nrows = (int)(B * D * H);
  ta.x = wide; ta.g = thin; ta.partial = static_cast<float*>(workspace);
  ta.bpartial = ta.partial + static_cast<int64_t>(ns) * 27 * Ct * Cw;
  ta.B = (int)B; ta.D = (int)D; ta.H = (int)H; ta.W = (int)W;
  ta.nrows = nrows; ta.nstreams = ns; ta.rows_per = (nrows + ns - 1) / ns;
  ta.RS = (int)(W * Ct + 8);
  const uint64_t xb = static_cast<uint64_t>(B * D * H * W) * static_cast<uint64_t>(Cw) * 4u;
  ta.x_bytes_lo = static_cast<unsigned>(xb & 0xffffffffu); ta.x_bytes_hi = static_cast<unsigned>(xb >> 32);
  if (!(static_cast<int64_t>(ta.rows_per) * W * Cw * 4 < (1LL << 32) && ns >= 4)) return kThinNotApplicable;
  if (static_cast<int64_t>(4) * 19 * ta.RS * 4 > df::lds_optin_bytes()) return kThinNotApplicable;      // LDS opt-in not available: generic path
  hipStream_t s = df::as_stream(stream);
  const size_t lds = static_cast<size_t>(4) * 19 * ta.RS * sizeof(float);
  const dim3 grid((unsigned)(ns / 4));
  const bool two = W * Ct > 256;
#define DF_WT(CT, NJ, SW, NP)                                                                                                         \
  do {                                                                                                                                \
    if (lds > 65536) {                                                                                                                \
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_thin_mfma_kernel<CT, NJ, SW, NP>),                  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))                                   \
        return df::fail((int)e, "df_conv_wgrad: LDS opt-in: %s", hipGetErrorString(e));                                               \
    }                                                                                                                                 \
    hipLaunchKernelGGL((wgrad_thin_mfma_kernel<CT, NJ, SW, NP>), grid, dim3(kThreads), lds, s, ta);                                   \
  } while (0)
  if (!swap && Cw == 128 && !two) {
    if (Ct == 1) DF_WT(1, 4, false, 1); else if (Ct == 2) DF_WT(2, 4, false, 1); else if (Ct == 3) DF_WT(3, 4, false, 1); else DF_WT(4, 4, false, 1);
  } else if (!swap && Cw == 128) DF_WT(3, 4, false, 2);
  else if (!swap && !two) DF_WT(3, 2, false, 1);
  else if (!swap) DF_WT(3, 2, false, 2);
  else if (Cw == 128 && !two) DF_WT(3, 4, true, 1);
  else if (Cw == 128) DF_WT(3, 4, true, 2);
  else if (!two) DF_WT(3, 2, true, 1);
  else DF_WT(3, 2, true, 2);
#undef DF_WT
  hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((unsigned)(27 * Ct * (Cw / 32))), dim3(kThreads), 0, s, ta.partial, ta.bpartial, gw, gb,
                     ns, (int)(27 * Ct), (int)Ct, (int)Cw, swap ? 1 : 0);
  return df::launched("df_conv_wgrad(thin mfma)");
}

static int conv_wgrad_impl(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                           int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream,
                           int prec, int algo_arg) {
  DF_REQUIRE(algo_arg >= 0 && (algo_arg & 7) <= 4, DF_EINVAL, "df_conv_wgrad: algo must be 0..4 (+ 8 * partial-range override)");
  const int req = algo_arg & 7, req_ranges = algo_arg >> 3;
  DF_REQUIRE(x && gy && gw && workspace, DF_EINVAL, "df_conv_wgrad: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "df_conv_wgrad: non-positive extent");
  DF_REQUIRE(kz == 1 || kz == 3, DF_ESHAPE, "df_conv_wgrad: kz must be 1 (2-D) or 3 (3-D)");
  DF_REQUIRE(kz == 3 || D == 1, DF_ESHAPE, "df_conv_wgrad: D must be 1 when kz == 1");
  DF_REQUIRE(B * D * H < (1LL << 30) && W < (1 << 24), DF_ESHAPE, "df_conv_wgrad: tensor too large");
  DF_REQUIRE(df::aligned16(workspace), DF_EALIGN, "df_conv_wgrad: workspace must be 16-byte aligned");
  DF_REQUIRE(workspace_bytes >= df_conv_wgrad_workspace_bytes(B, D, H, W, Cin, Cout, kz), DF_EWORKSPACE,
             "df_conv_wgrad: workspace too small");
  {
    // matrix-core form of the thin layers: F -> 1..4 (wide = x, thin = gy) or 1..4 -> F (SWAP: wide = gy, thin = x)
    const bool swap = Cin <= 4 && Cout >= 64;
    const int64_t Cw = swap ? Cout : Cin, Ct = swap ? Cin : Cout;
    if ((swap || small_n_ok(Cin, Cout)) && req != 1 && prec == 0 && thin_mfma_ok(B, D, H, W, Cw, Ct, kz) &&
        thin_mfma_inst(W, Cw, Ct, swap) && df::aligned16(x) && df::aligned16(gy)) {
      const int rc = launch_thin_wgrad(swap ? gy : x, swap ? x : gy, gw, gb, B, D, H, W, Cw, Ct, swap, workspace, stream);
      if (rc != kThinNotApplicable) return rc;      // shape outside the kernel's 32-bit stream offsets -> generic path below
    }
  }
  if (small_n_ok(Cin, Cout)) {
    const SmallPlan sp = make_small_plan(B, D, H, Cin, Cout, kz);
    SmallWgradArgs sa;
    sa.x = x; sa.g = gy; sa.partial = static_cast<float*>(workspace);
    float* zeros = sa.partial + sp.partial_elems;
    sa.zeros = zeros;
    sa.bpartial = zeros - static_cast<int64_t>(sp.nstreams) * sp.CO;
    sa.B = (int)B; sa.D = (int)D; sa.H = (int)H; sa.W = (int)W; sa.Cin = (int)Cin; sa.Cout = (int)Cout;
    sa.Wp = (int)(ceil_div(W + 1, 8) * 8);
    sa.nrows = sp.nrows; sa.nstreams = sp.nstreams; sa.rows_per = sp.rows_per; sa.ncib = sp.ncib;
    hipStream_t s = df::as_stream(stream);
    if (hipError_t e = df::zero_async(zeros, kZeroBytes, s)) return df::fail((int)e, "df_conv_wgrad: memset: %s", hipGetErrorString(e));
    dim3 grid((unsigned)(sp.nstreams * sp.ncib / 4));
#define DF_WS(KZ, CO) hipLaunchKernelGGL((wgrad_small_n_kernel<KZ, CO>), grid, dim3(kThreads), 0, s, sa)
    if (kz == 3) { if (Cout == 1) DF_WS(3, 1); else if (Cout == 2) DF_WS(3, 2); else if (Cout == 3) DF_WS(3, 3); else DF_WS(3, 4); }
    else { if (Cout == 1) DF_WS(1, 1); else if (Cout == 2) DF_WS(1, 2); else if (Cout == 3) DF_WS(1, 3); else DF_WS(1, 4); }
#undef DF_WS
    const int64_t per = static_cast<int64_t>(sp.taps) * Cin * sp.CO;
    int64_t rg = ceil_div(per, 32);
    hipLaunchKernelGGL(wgrad_small_reduce_kernel, dim3((unsigned)rg), dim3(kThreads), 0, s, sa.partial, sa.bpartial, gw,
                       gb, sp.nstreams, per, sp.CO, (int)Cout);
    return df::launched("df_conv_wgrad(small-N)");
  }
  // (both precision modes: where the (x,y,z) form exists the bf16x3 mode keeps it -- see use_bf16x3 below -- so rows that reach an
  //  instantiated length by zero padding take it too instead of the direct kernel; df_conv_wgrad_form reports 3 for them in both modes)
  if (const int64_t Wp = df::aligned16(x) && df::aligned16(gy) ? wxyz_padded_w(req, D, H, W, Cin, Cout, kz) : 0) {
    const int64_t inner = up256(df_conv_wgrad_workspace_bytes(B, D, H, Wp, Cin, Cout, kz));
    float* xp = reinterpret_cast<float*>(static_cast<char*>(workspace) + inner);
    float* gp = reinterpret_cast<float*>(reinterpret_cast<char*>(xp) + up256(B * D * H * Wp * Cin * 4));
    hipStream_t s = df::as_stream(stream);
    const int64_t rows = B * D * H;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ceil_div(rows * Wp * (Cin / 4), kThreads)), dim3(kThreads), 0, s,
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(xp), rows, (int)W, (int)Wp, (int)(Cin / 4));
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ceil_div(rows * Wp * (Cout / 4), kThreads)), dim3(kThreads), 0, s,
                       reinterpret_cast<const f32x4*>(gy), reinterpret_cast<f32x4*>(gp), rows, (int)W, (int)Wp, (int)(Cout / 4));
    if (int e = df::launched("df_conv_wgrad(pad rows)")) return e;
    return conv_wgrad_impl(xp, gp, gw, gb, B, D, H, Wp, Cin, Cout, kz, workspace, inner, stream, prec, algo_arg);
  }
  const bool xvec = (Cin % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) & 7u) == 0);
  const bool gvec = (Cout % 2 == 0) && ((reinterpret_cast<uintptr_t>(gy) & 7u) == 0);
  // bf16x3 mode: rows without a bf16x3 variant (W = 56 | 28 | ... of cfg4) take the fp32 Winograd forms, not the 2.6x slower direct kernel
  // ... and where the fp32 (x,y,z) Winograd form exists it is FASTER than the split-operand kernel (cfg3 top level 14.2 vs 15.2 ms, one level
  // down 1.76 vs 2.03) and exact: the opt-in precision mode keeps it
  const bool use_bf16x3 = prec == 1 && xvec && gvec && wgrad_bf16x3_ok(W, Cin, Cout) &&
                          !((req == 0 || req == 4) && wxyz_ok(D, H, W, Cin, Cout, kz) && df::aligned16(x) && df::aligned16(gy));
  const int algo = (!use_bf16x3 && xvec && gvec) ? wgrad_algo(req, B * D * H, D, H, W, Cin, Cout, kz) : 0;
  const Plan p = make_plan(B, D, H, W, Cin, Cout, kz, algo, req_ranges);
  // a caller-chosen number of partial ranges may need more room than df_conv_wgrad_workspace_bytes (sized for the defaults) promises
  DF_REQUIRE(workspace_bytes >= (p.partial_elems + p.bpartial_elems) * static_cast<int64_t>(sizeof(float)) + zero_row_bytes(W, Cin, Cout),
             DF_EWORKSPACE, "df_conv_wgrad: workspace too small for %d partial ranges", p.nranges);
  WgradArgs a;
  a.x = x; a.g = gy;
  a.partial = static_cast<float*>(workspace);
  a.bpartial = a.partial + p.partial_elems;
  float* zeros = a.bpartial + p.bpartial_elems;
  a.zeros = zeros;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.Cinp = p.Cinp; a.Coutp = p.Coutp;
  a.Wp = (int)(ceil_div(W, 8) * 8);
  a.nrows = p.nrows; a.npairs = p.npairs; a.nranges = p.nranges; a.pairs_per_range = p.ppr;
  a.ndzdy = p.ndzdy; a.want_bias = gb != nullptr;
  a.nqi = p.nqi; a.nqj = p.nqj; a.nsub = p.nsub;
  a.up = 0; a.gD = a.D; a.gH = a.H; a.gW = a.W;
  hipStream_t s = df::as_stream(stream);
  if (hipError_t e = df::zero_async(zeros, zero_row_bytes(W, Cin, Cout), s)) return df::fail((int)e, "df_conv_wgrad: memset: %s", hipGetErrorString(e));
  dim3 grid((unsigned)(p.nranges * p.ndzdy), (unsigned)ceil_div(Cin, 128), (unsigned)ceil_div(Cout, 128));
  if (use_bf16x3) {
    launch_wgrad_bf16x3(W, grid, s, a);
  } else if (algo == 3) {
    WxyzArgs aa;
    aa.w = a; aa.Ht = (int)(H / 2); aa.Dt = (int)(D / 2); aa.ntrows = p.nrows;
#ifdef DF_TUNING
    const dim3 gridq((unsigned)(p.nranges * 4), grid.y, grid.z);      // 4 workgroup types per launch
#endif
    const dim3 gridf((unsigned)(p.nranges * 16), grid.y, grid.z);       // all 16 (xi_z, xi_y) types of a range, adjacent
#define DF_WXYZ(WP) hipLaunchKernelGGL((wgrad_wxyz_fused_kernel<WP, 128>), gridf, dim3(kThreads), 0, s, aa)
#ifdef DF_TUNING
    // experiments of tools/wgrad_diag.py (W = 64 rows): four launches by (GZ, GY) class (the round-1 form) with DBGV variants
#define DF_WXYZ_DBG(DBGV)                                                                                              \
  do {                                                                                                                 \
    hipLaunchKernelGGL((wgrad_wxyz_kernel<8, 128, true, true, false, DBGV>), gridq, dim3(kThreads), 0, s, aa);         \
    hipLaunchKernelGGL((wgrad_wxyz_kernel<8, 128, true, false, false, DBGV>), gridq, dim3(kThreads), 0, s, aa);        \
    hipLaunchKernelGGL((wgrad_wxyz_kernel<8, 128, false, true, false, DBGV>), gridq, dim3(kThreads), 0, s, aa);        \
    hipLaunchKernelGGL((wgrad_wxyz_kernel<8, 128, false, false, false, DBGV>), gridq, dim3(kThreads), 0, s, aa);       \
  } while (0)
    if (W == 64 && g_wgrad_dbg == 1) DF_WXYZ_DBG(1); else if (W == 64 && g_wgrad_dbg == 2) DF_WXYZ_DBG(2);
    else if (W == 64 && g_wgrad_dbg == 3) DF_WXYZ_DBG(3); else if (W == 64 && g_wgrad_dbg == 6) DF_WXYZ_DBG(6);
    else if (W == 64 && g_wgrad_dbg == 7) DF_WXYZ_DBG(7); else if (W == 64 && g_wgrad_dbg == 8) DF_WXYZ_DBG(0);
    else if (W == 64 && g_wgrad_dbg == 9) hipLaunchKernelGGL((wgrad_wxyz_fused_kernel<8, 128, 1>), gridf, dim3(kThreads), 0, s, aa);
    else if (W == 64 && g_wgrad_dbg == 10) hipLaunchKernelGGL((wgrad_wxyz_fused_kernel<8, 128, 8>), gridf, dim3(kThreads), 0, s, aa);
    else
#undef DF_WXYZ_DBG
#endif
#define DF_WXYZ64(WP) hipLaunchKernelGGL((wgrad_wxyz_fused_kernel<WP, 64>), gridf, dim3(kThreads), 0, s, aa)
#ifdef DF_TUNING
    if (Cin == 64 && W == 128 && g_wgrad_dbg == 16) hipLaunchKernelGGL((wgrad_wxyz_fused_kernel<16, 64, 16>), gridf, dim3(kThreads), 0, s, aa);
    else if (Cin == 64 && W == 64 && g_wgrad_dbg == 16) hipLaunchKernelGGL((wgrad_wxyz_fused_kernel<8, 64, 16>), gridf, dim3(kThreads), 0, s, aa);
    else
#endif
    if (Cin == 64) { if (W == 128) DF_WXYZ64(16); else if (W == 64) DF_WXYZ64(8); else if (W == 32) DF_WXYZ64(4); else DF_WXYZ64(2); }
    else
    if (W == 64) DF_WXYZ(8); else if (W == 32) DF_WXYZ(4); else if (W == 16) DF_WXYZ(2); else if (W == 112) DF_WXYZ(14); else if (W == 128) DF_WXYZ(16); else DF_WXYZ(7);
#undef DF_WXYZ
#undef DF_WXYZ64
    const int64_t rgx = ceil_div(Cin * Cout, 32);
    hipLaunchKernelGGL(wgrad_wxyz_reduce_kernel, dim3((unsigned)rgx), dim3(kThreads), 0, s, a.partial, a.bpartial, gw, gb,
                       Cin == 64 ? p.nranges : p.nranges * p.nsub, (int)Cin, (int)Cout, p.Cinp, p.Coutp, 0);      // (64 -> 64: summed in the workgroup)
    return df::launched("df_conv_wgrad(winograd-xyz)");
  } else if (algo == 2) {
    WxyArgs aa;
    aa.w = a; aa.Ht = (int)(H / 2); aa.ntrows = p.nrows;
    const bool c128 = Cin == 128 && Cout == 128;
    aa.w.ndzdy = p.ndzdy / 2;                         // workgroup types per launch: (dz) x 2 xi_y
    const dim3 gridh((unsigned)(p.nranges * aa.w.ndzdy), grid.y, grid.z);
#define DF_WXY(WP, CSV)                                                                                  \
  do {                                                                                                   \
    hipLaunchKernelGGL((wgrad_wxy_kernel<WP, CSV, true>), gridh, dim3(kThreads), 0, s, aa);              \
    hipLaunchKernelGGL((wgrad_wxy_kernel<WP, CSV, false>), gridh, dim3(kThreads), 0, s, aa);             \
  } while (0)
    if (W == 64 && c128) DF_WXY(8, 128);
    else if (W == 64) DF_WXY(8, 0);
    else if (W == 128) DF_WXY(16, 0);      // AE3 128^3 (cfg5)
    else if (W == 96 && c128) DF_WXY(12, 128);      // 2-D 128x96 (cfg1/cfg2) and its 64x48 level
    else if (W == 96) DF_WXY(12, 0);
    else if (W == 48 && c128) DF_WXY(6, 128);
    else if (W == 48) DF_WXY(6, 0);
    else if (W == 112) DF_WXY(14, 0);      // cfg4 row lengths
    else if (W == 56) DF_WXY(7, 0);
    else if (W == 32 && c128) DF_WXY(4, 128);
    else if (W == 32) DF_WXY(4, 0);
    else if (c128) DF_WXY(2, 128);
    else DF_WXY(2, 0);
#undef DF_WXY
    const int ndz = kz == 3 ? 3 : 1;
    const int64_t tot = static_cast<int64_t>(ndz) * Cin * Cout;
    const int64_t rgx = ceil_div(tot, 32);
    hipLaunchKernelGGL(wgrad_wxy_reduce_kernel, dim3((unsigned)rgx), dim3(kThreads), 0, s, a.partial, a.bpartial, gw, gb,
                       p.nranges * p.nsub, ndz, (int)Cin, (int)Cout, p.Cinp, p.Coutp);
    return df::launched("df_conv_wgrad(winograd-xy)");
  } else if (algo == 1) {
    const bool c128 = Cin == 128 && Cout == 128;
    if (W == 64 && c128) hipLaunchKernelGGL((wgrad_wx_kernel<8, 128>), grid, dim3(kThreads), 0, s, a);
    else if (W == 64) hipLaunchKernelGGL((wgrad_wx_kernel<8, 0>), grid, dim3(kThreads), 0, s, a);
    else if (W == 128) hipLaunchKernelGGL((wgrad_wx_kernel<16, 0>), grid, dim3(kThreads), 0, s, a);
    else if (W == 96) hipLaunchKernelGGL((wgrad_wx_kernel<12, 0>), grid, dim3(kThreads), 0, s, a);
    else if (W == 48) hipLaunchKernelGGL((wgrad_wx_kernel<6, 0>), grid, dim3(kThreads), 0, s, a);
    else if (W == 112) hipLaunchKernelGGL((wgrad_wx_kernel<14, 0>), grid, dim3(kThreads), 0, s, a);
    else if (W == 56) hipLaunchKernelGGL((wgrad_wx_kernel<7, 0>), grid, dim3(kThreads), 0, s, a);
    else if (W == 32 && c128) hipLaunchKernelGGL((wgrad_wx_kernel<4, 128>), grid, dim3(kThreads), 0, s, a);
    else if (W == 32) hipLaunchKernelGGL((wgrad_wx_kernel<4, 0>), grid, dim3(kThreads), 0, s, a);
    else if (c128) hipLaunchKernelGGL((wgrad_wx_kernel<2, 128>), grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((wgrad_wx_kernel<2, 0>), grid, dim3(kThreads), 0, s, a);
    const int64_t tot = static_cast<int64_t>(p.ndzdy) * Cin * Cout;
    const int64_t rgx = ceil_div(tot, 32) + (gb ? ceil_div(Cout, 32) : 0);      // 32 elements per workgroup + the bias workgroups
    hipLaunchKernelGGL(wgrad_wx_reduce_kernel, dim3((unsigned)rgx), dim3(kThreads), 0, s, a.partial, a.bpartial, gw, gb,
                       p.nranges * p.nsub, p.ndzdy, (int)Cin, (int)Cout, p.Cinp, p.Coutp);
    return df::launched("df_conv_wgrad(winograd-x)");
  } else {
  const int wp8 = a.Wp / 8;
  const bool exact = (W % 8) == 0;
  if (xvec && gvec && exact && wp8 == 8) hipLaunchKernelGGL((wgrad_kernel<true, true, 8>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 16) hipLaunchKernelGGL((wgrad_kernel<true, true, 16>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 4) hipLaunchKernelGGL((wgrad_kernel<true, true, 4>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 14) hipLaunchKernelGGL((wgrad_kernel<true, true, 14>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 12) hipLaunchKernelGGL((wgrad_kernel<true, true, 12>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 7) hipLaunchKernelGGL((wgrad_kernel<true, true, 7>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 2) hipLaunchKernelGGL((wgrad_kernel<true, true, 2>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 3) hipLaunchKernelGGL((wgrad_kernel<true, true, 3>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec && exact && wp8 == 6) hipLaunchKernelGGL((wgrad_kernel<true, true, 6>), grid, dim3(kThreads), 0, s, a);
  else if (xvec && gvec) hipLaunchKernelGGL((wgrad_kernel<true, true, 0>), grid, dim3(kThreads), 0, s, a);
  else if (xvec) hipLaunchKernelGGL((wgrad_kernel<true, false, 0>), grid, dim3(kThreads), 0, s, a);
  else if (gvec) hipLaunchKernelGGL((wgrad_kernel<false, true, 0>), grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((wgrad_kernel<false, false, 0>), grid, dim3(kThreads), 0, s, a);
  }
  const int64_t total = static_cast<int64_t>(p.taps) * Cin * Cout;
  const int64_t rg = ceil_div(total, 32) + (gb ? ceil_div(Cout, 32) : 0);      // 32 elements per workgroup + the bias workgroups
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rg), dim3(kThreads), 0, s, a.partial, a.bpartial, gw, gb,
                     p.nranges * p.nsub, p.taps, (int)Cin, (int)Cout, p.Cinp, p.Coutp);
  return df::launched("df_conv_wgrad");
}

int df_conv_wgrad(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                  int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  return conv_wgrad_impl(x, gy, gw, gb, B, D, H, W, Cin, Cout, kz, workspace, workspace_bytes, stream, 0, 0);
}
int df_conv_wgrad_algo(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                       int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, int algo, df_stream_t stream) {
  return conv_wgrad_impl(x, gy, gw, gb, B, D, H, W, Cin, Cout, kz, workspace, workspace_bytes, stream, 0, algo);
}
int df_conv_wgrad_bf16x3(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t D, int64_t H, int64_t W,
                         int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  return conv_wgrad_impl(x, gy, gw, gb, B, D, H, W, Cin, Cout, kz, workspace, workspace_bytes, stream, 1, 0);
}

// ---- stride-2 weight gradient (see wgrad_s2_kernel) ---------------------------------------------------------------------------------
static bool s2_wgrad_ok(int64_t Wo, int64_t Cin, int64_t Cout) {
  return (Wo == 8 || Wo == 16 || Wo == 32 || Wo == 64) && Cin % 2 == 0 && Cout % 2 == 0 && Cin >= 32 && Cout >= 32;
}
int64_t df_conv_s2_wgrad_workspace_bytes(int64_t B, int64_t Do, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, int kz) {
  if (B <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || !s2_wgrad_ok(Wo, Cin, Cout)) return 0;
  const Plan p = make_plan(B, Do, Ho, Wo, Cin, Cout, kz);
  return (p.partial_elems + p.bpartial_elems) * static_cast<int64_t>(sizeof(float)) + zero_row_bytes(2 * Wo, Cin, Cout);
}
int df_conv_s2_wgrad(const float* x, const float* gy, float* gw, float* gb, int64_t B, int64_t Do, int64_t Ho, int64_t Wo, int64_t Cin,
                     int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  DF_REQUIRE(x && gy && gw && workspace, DF_EINVAL, "df_conv_s2_wgrad: null pointer");
  DF_REQUIRE(B > 0 && Do > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "df_conv_s2_wgrad: non-positive extent");
  DF_REQUIRE(kz == 1 || kz == 3, DF_ESHAPE, "df_conv_s2_wgrad: kz must be 1 (2-D) or 3 (3-D)");
  DF_REQUIRE(kz == 3 || Do == 1, DF_ESHAPE, "df_conv_s2_wgrad: Do must be 1 when kz == 1");
  DF_REQUIRE(s2_wgrad_ok(Wo, Cin, Cout), DF_ESHAPE,
             "df_conv_s2_wgrad: output rows of 8 | 16 | 32 | 64 voxels and even channel counts >= 32 only (other shapes: df_dilate2_odd + df_conv_wgrad)");
  DF_REQUIRE(B * Do * Ho < (1LL << 30) && 8 * B * Do * Ho * Wo * Cin < (1LL << 40), DF_ESHAPE, "df_conv_s2_wgrad: tensor too large");
  DF_REQUIRE(df::aligned16(workspace) && (reinterpret_cast<uintptr_t>(x) & 7u) == 0 && (reinterpret_cast<uintptr_t>(gy) & 7u) == 0, DF_EALIGN,
             "df_conv_s2_wgrad: workspace must be 16-byte, x and gy 8-byte aligned");
  DF_REQUIRE(workspace_bytes >= df_conv_s2_wgrad_workspace_bytes(B, Do, Ho, Wo, Cin, Cout, kz), DF_EWORKSPACE, "df_conv_s2_wgrad: workspace too small");
  const Plan p = make_plan(B, Do, Ho, Wo, Cin, Cout, kz);
  WgradArgs a;
  a.x = x; a.g = gy;
  a.partial = static_cast<float*>(workspace);
  a.bpartial = a.partial + p.partial_elems;
  float* zeros = a.bpartial + p.bpartial_elems;
  a.zeros = zeros;
  a.B = (int)B; a.D = (int)Do; a.H = (int)Ho; a.W = (int)Wo; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.Cinp = p.Cinp; a.Coutp = p.Coutp;
  a.Wp = (int)Wo;
  a.nrows = p.nrows; a.npairs = p.npairs; a.nranges = p.nranges; a.pairs_per_range = p.ppr;
  a.ndzdy = p.ndzdy; a.want_bias = gb != nullptr;
  a.nqi = p.nqi; a.nqj = p.nqj; a.nsub = p.nsub;
  a.up = 0; a.gD = a.D; a.gH = a.H; a.gW = a.W;
  hipStream_t s = df::as_stream(stream);
  if (hipError_t e = df::zero_async(zeros, zero_row_bytes(2 * Wo, Cin, Cout), s)) return df::fail((int)e, "df_conv_s2_wgrad: memset: %s", hipGetErrorString(e));
  dim3 grid((unsigned)(p.nranges * p.ndzdy), (unsigned)ceil_div(Cin, 128), (unsigned)ceil_div(Cout, 128));
  if (Wo == 64) hipLaunchKernelGGL((wgrad_s2_kernel<8>), grid, dim3(kThreads), 0, s, a);
  else if (Wo == 32) hipLaunchKernelGGL((wgrad_s2_kernel<4>), grid, dim3(kThreads), 0, s, a);
  else if (Wo == 16) hipLaunchKernelGGL((wgrad_s2_kernel<2>), grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((wgrad_s2_kernel<1>), grid, dim3(kThreads), 0, s, a);
  const int64_t total = static_cast<int64_t>(p.taps) * Cin * Cout;
  const int64_t rg = ceil_div(total, 32) + (gb ? ceil_div(Cout, 32) : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rg), dim3(kThreads), 0, s, a.partial, a.bpartial, gw, gb,
                     p.nranges * p.nsub, p.taps, (int)Cin, (int)Cout, p.Cinp, p.Coutp);
  return df::launched("df_conv_s2_wgrad");
}

// Which kernel family a df_conv_wgrad_algo call with these arguments runs (16-byte aligned operands assumed): the silent size-based
// choices made visible to the caller.  0 direct MFMA | 1 Winograd in x | 2 Winograd in (x,y) | 3 Winograd in (x,y,z) |
// 10 thin layer on the matrix cores | 11 thin layer on the vector ALU (general-shape fallback); < 0: invalid arguments.
int df_conv_wgrad_form(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int algo) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (kz != 1 && kz != 3) || algo < 0 || (algo & 7) > 4) return DF_EINVAL;
  const int req = algo & 7;
  const bool swap = Cin <= 4 && Cout >= 64;
  const int64_t Cw = swap ? Cout : Cin, Ct = swap ? Cin : Cout;
  if ((swap || small_n_ok(Cin, Cout)) && req != 1 && thin_mfma_ok(B, D, H, W, Cw, Ct, kz) && thin_mfma_inst(W, Cw, Ct, swap)) return 10;
  if (small_n_ok(Cin, Cout)) return 11;
  if (Cin % 2 || Cout % 2) return 0;
  if (wxyz_padded_w(req, D, H, W, Cin, Cout, kz)) return 3;      // (x,y,z) on zero-padded rows
  return wgrad_algo(req, B * D * H, D, H, W, Cin, Cout, kz);
}

static Plan make_up_plan(int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, int kz) {
  Plan p = make_plan(B, Dc, Hc, Wc, Cin, Cout, kz);
  p.ndzdy = kz == 3 ? 32 : 8;
  // 32 | 8 (class, dz/dy) combos per range: 64 ranges x 32 combos = 2048 workgroups = 8 whole rounds of 256 CUs, and the
  // partial buffer (96 | 24 slots per range) stays 4x smaller than with 256 ranges
  const int nr = p.npairs >= 64 ? 64 : p.npairs;
  p.ppr = (p.npairs + nr - 1) / nr;
  p.nranges = (p.npairs + p.ppr - 1) / p.ppr;
  p.taps = p.ndzdy * 3;                                  // partial slots per range
  p.partial_elems = static_cast<int64_t>(p.nranges) * p.nsub * p.taps * p.Cinp * p.Coutp;
  p.bpartial_elems = static_cast<int64_t>(p.nranges) * p.nsub * (kz == 3 ? 8 : 4) * p.Coutp;
  return p;
}

// the 27-point Winograd-(x,y,z) form of the up-sampling-aware weight gradient (fine extents 2Dc x 2Hc x 2Wc)
static bool up_wxyz_ok(int req, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, int kz) {
  // at every size ([r3] batch 2 | 4 | 16, fine 16x24x16: parity-class 0.216 | 0.315 | 0.443 ms, 27-point 0.048 | 0.058 | 0.132; fine 32x48x32:
  // 0.442 | 0.597 | 1.484 vs 0.129 | 0.222 | 0.787): the parity-class form pays a fixed-cost reduce of its 32-combo partials
  (void)B;
  return wxyz_ok(2 * Dc, 2 * Hc, 2 * Wc, Cin, Cout, kz) && req != 1 && req != 2 && req != 3;
}

// partial ranges of the 27-point (x,y,z) form: 9 workgroup types per range in one launch -- 28 ranges (252 workgroups: one round of
// the 256 CUs) below 4096 tile-row pairs, 113 (1017: four rounds) above
// (64 -> 64: the spare waves split each range four ways -- 28 ranges are already 112 partials)
static int up_wxyz_ranges(int64_t B, int64_t Dc, int64_t Hc, int64_t Cin = 128, int64_t Cout = 128) {
  return ((B * Dc * Hc + 1) / 2 < 4096 || (Cin <= 64 && Cout <= 64)) ? 28 : 113;
}

// The same for df_upconv_wgrad_algo: 3 = the 27-point Winograd-(x,y,z) form on the coarse input, 0 = the three-product parity-class
// kernel (algo 0 | 2) or the generic direct kernel on class-strided views (algo 1).
int df_upconv_wgrad_form(int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, int kz, int algo) {
  if (B <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0 || Cin <= 0 || Cout <= 0 || (kz != 1 && kz != 3) || algo < 0 || algo > 4) return DF_EINVAL;
  return up_wxyz_ok(algo, B, Dc, Hc, Wc, Cin, Cout, kz) ? 3 : 0;
}

int64_t df_upconv_wgrad_workspace_bytes(int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, int kz) {
  if (B <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (wxyz_ok(2 * Dc, 2 * Hc, 2 * Wc, Cin, Cout, kz)) {      // sized for either form (the choice depends on a debug switch)
    const Plan q = make_plan(B, 2 * Dc, 2 * Hc, 2 * Wc, Cin, Cout, kz, 3, up_wxyz_ranges(B, Dc, Hc, Cin, Cout));
    const Plan p = make_up_plan(B, Dc, Hc, Wc, Cin, Cout, kz);
    const int64_t nq = (q.partial_elems + q.bpartial_elems) * static_cast<int64_t>(sizeof(float)) + zero_row_bytes(2 * Wc, Cin, Cout);
    const int64_t np = (p.partial_elems + p.bpartial_elems) * static_cast<int64_t>(sizeof(float)) + zero_row_bytes(Wc, Cin, Cout);
    return nq > np ? nq : np;
  }
  const Plan p = make_up_plan(B, Dc, Hc, Wc, Cin, Cout, kz);
  return (p.partial_elems + p.bpartial_elems) * static_cast<int64_t>(sizeof(float)) + zero_row_bytes(Wc, Cin, Cout);
}

static int upconv_wgrad_impl(const float* xc, const float* gy, float* gw, float* gb, int64_t B, int64_t Dc, int64_t Hc,
                             int64_t Wc, int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes,
                             df_stream_t stream, int prec, int req) {
  DF_REQUIRE(req >= 0 && req <= 4, DF_EINVAL, "df_upconv_wgrad: algo must be 0..4");
  DF_REQUIRE(xc && gy && gw && workspace, DF_EINVAL, "df_upconv_wgrad: null pointer");
  DF_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "df_upconv_wgrad: non-positive extent");
  DF_REQUIRE(kz == 1 || kz == 3, DF_ESHAPE, "df_upconv_wgrad: kz must be 1 (2-D) or 3 (3-D)");
  DF_REQUIRE(kz == 3 || Dc == 1, DF_ESHAPE, "df_upconv_wgrad: Dc must be 1 when kz == 1");
  DF_REQUIRE(Cin % 2 == 0 && Cout % 2 == 0, DF_ESHAPE, "df_upconv_wgrad: even channel counts only");
  DF_REQUIRE(df::aligned16(workspace), DF_EALIGN, "df_upconv_wgrad: workspace must be 16-byte aligned");
  DF_REQUIRE(workspace_bytes >= df_upconv_wgrad_workspace_bytes(B, Dc, Hc, Wc, Cin, Cout, kz), DF_EWORKSPACE,
             "df_upconv_wgrad: workspace too small");
  // (the 27-point fp32 form beats the bf16x3 parity-class kernel too -- 5.8 vs 7.0 ms at the cfg3 top level -- so both precision modes take it)
  if (up_wxyz_ok(req, B, Dc, Hc, Wc, Cin, Cout, kz) && df::aligned16(xc) && df::aligned16(gy)) {
    const int64_t D = 2 * Dc, H = 2 * Hc, W = 2 * Wc;
    const Plan p = make_plan(B, D, H, W, Cin, Cout, kz, 3, up_wxyz_ranges(B, Dc, Hc, Cin, Cout));
    WxyzArgs aa;
    WgradArgs& a = aa.w;
    a.x = xc; a.g = gy;
    a.partial = static_cast<float*>(workspace);
    a.bpartial = a.partial + p.partial_elems;
    float* zeros = a.bpartial + p.bpartial_elems;
    a.zeros = zeros;
    a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
    a.Cinp = p.Cinp; a.Coutp = p.Coutp;
    a.Wp = (int)W;
    a.nrows = p.nrows; a.npairs = p.npairs; a.nranges = p.nranges; a.pairs_per_range = p.ppr;
    a.ndzdy = p.ndzdy; a.want_bias = gb != nullptr;
    a.nqi = p.nqi; a.nqj = p.nqj; a.nsub = p.nsub;
    a.up = 0; a.gD = a.D; a.gH = a.H; a.gW = a.W;
    aa.Ht = (int)Hc; aa.Dt = (int)Dc; aa.ntrows = p.nrows;
    hipStream_t s = df::as_stream(stream);
    if (hipError_t e = df::zero_async(zeros, zero_row_bytes(W, Cin, Cout), s)) return df::fail((int)e, "df_upconv_wgrad: memset: %s", hipGetErrorString(e));
    const unsigned gy_ = (unsigned)ceil_div(Cin, 128), gz_ = (unsigned)ceil_div(Cout, 128);
    const dim3 gridu((unsigned)(p.nranges * 9), gy_, gz_);          // all 9 live (xi_z, xi_y) types of a range, adjacent
#define DF_UWXYZ(WP) hipLaunchKernelGGL((wgrad_wxyz_up_fused_kernel<WP, 128>), gridu, dim3(kThreads), 0, s, aa)
#define DF_UWXYZ64(WP) hipLaunchKernelGGL((wgrad_wxyz_up_fused_kernel<WP, 64>), gridu, dim3(kThreads), 0, s, aa)
    if (Cin == 64) { if (W == 128) DF_UWXYZ64(16); else if (W == 64) DF_UWXYZ64(8); else if (W == 32) DF_UWXYZ64(4); else DF_UWXYZ64(2); }
    else
    if (W == 64) DF_UWXYZ(8); else if (W == 32) DF_UWXYZ(4); else if (W == 16) DF_UWXYZ(2); else if (W == 112) DF_UWXYZ(14); else if (W == 128) DF_UWXYZ(16); else DF_UWXYZ(7);
#undef DF_UWXYZ
#undef DF_UWXYZ64
    const int64_t rgx = ceil_div(Cin * Cout, 32);
    hipLaunchKernelGGL(wgrad_wxyz_reduce_kernel, dim3((unsigned)rgx), dim3(kThreads), 0, s, a.partial, a.bpartial, gw, gb,
                       Cin == 64 ? p.nranges : p.nranges * p.nsub, (int)Cin, (int)Cout, p.Cinp, p.Coutp, 1);
    return df::launched("df_upconv_wgrad(winograd-xyz, 27-point)");
  }
  const Plan p = make_up_plan(B, Dc, Hc, Wc, Cin, Cout, kz);
  WgradArgs a;
  a.x = xc; a.g = gy;
  a.partial = static_cast<float*>(workspace);
  a.bpartial = a.partial + p.partial_elems;
  float* zeros = a.bpartial + p.bpartial_elems;
  a.zeros = zeros;
  a.B = (int)B; a.D = (int)Dc; a.H = (int)Hc; a.W = (int)Wc; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.Cinp = p.Cinp; a.Coutp = p.Coutp;
  a.Wp = (int)(ceil_div(Wc, 8) * 8);
  a.nrows = p.nrows; a.npairs = p.npairs; a.nranges = p.nranges; a.pairs_per_range = p.ppr;
  a.ndzdy = p.ndzdy; a.want_bias = gb != nullptr;
  a.nqi = p.nqi; a.nqj = p.nqj; a.nsub = p.nsub;
  a.up = 1; a.gD = kz == 3 ? 2 * a.D : 1; a.gH = 2 * a.H; a.gW = 2 * a.W;
  hipStream_t s = df::as_stream(stream);
  if (hipError_t e = df::zero_async(zeros, zero_row_bytes(Wc, Cin, Cout), s)) return df::fail((int)e, "df_upconv_wgrad: memset: %s", hipGetErrorString(e));
  dim3 grid((unsigned)(p.nranges * p.ndzdy), (unsigned)ceil_div(Cin, 128), (unsigned)ceil_div(Cout, 128));
  const int wp8 = a.Wp / 8;
  const bool exact = (Wc % 8) == 0;
  const bool aligned8 = ((reinterpret_cast<uintptr_t>(xc) | reinterpret_cast<uintptr_t>(gy)) & 7u) == 0;
  const dim3 grid2((unsigned)(p.nranges * (p.ndzdy / 2)), grid.y, grid.z);     // one workgroup per x-parity class PAIR
  if (prec == 1 && wgrad_bf16x3_ok(Wc, Cin, Cout)) launch_wgrad_bf16x3(Wc, grid, s, a);
  else if (aligned8 && exact && wp8 == 4 && Cin == 128 && Cout == 128 && req != 1) hipLaunchKernelGGL((wgrad_up2_kernel<4, 128>), grid2, dim3(kThreads), 0, s, a);
  else if (aligned8 && exact && wp8 == 6 && Cin == 128 && Cout == 128 && req != 1) hipLaunchKernelGGL((wgrad_up2_kernel<6, 128>), grid2, dim3(kThreads), 0, s, a);      // 2-D 128x96: coarse W = 48 | 24
  else if (aligned8 && exact && wp8 == 3 && Cin == 128 && Cout == 128 && req != 1) hipLaunchKernelGGL((wgrad_up2_kernel<3, 128>), grid2, dim3(kThreads), 0, s, a);
  else if (aligned8 && exact && wp8 == 2 && req != 1) hipLaunchKernelGGL((wgrad_up2_kernel<2, 0>), grid2, dim3(kThreads), 0, s, a);
  else if (aligned8 && exact && wp8 == 1 && req != 1) hipLaunchKernelGGL((wgrad_up2_kernel<1, 0>), grid2, dim3(kThreads), 0, s, a);
  else if (exact && wp8 == 8) hipLaunchKernelGGL((wgrad_kernel<true, true, 8>), grid, dim3(kThreads), 0, s, a);
  else if (exact && wp8 == 4) hipLaunchKernelGGL((wgrad_kernel<true, true, 4>), grid, dim3(kThreads), 0, s, a);
  else if (exact && wp8 == 2) hipLaunchKernelGGL((wgrad_kernel<true, true, 2>), grid, dim3(kThreads), 0, s, a);
  else if (exact && wp8 == 7) hipLaunchKernelGGL((wgrad_kernel<true, true, 7>), grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((wgrad_kernel<true, true, 0>), grid, dim3(kThreads), 0, s, a);
  const int64_t total = static_cast<int64_t>(kz * 9) * Cin * Cout;
  const int64_t rg = ceil_div(total, 32);
  hipLaunchKernelGGL(wgrad_up_reduce_kernel, dim3((unsigned)rg), dim3(kThreads), 0, s, a.partial, a.bpartial, gw, gb,
                     p.nranges * p.nsub, kz, (int)Cin, (int)Cout, p.Cinp, p.Coutp);
  return df::launched("df_upconv_wgrad");
}

int df_upconv_wgrad(const float* xc, const float* gy, float* gw, float* gb, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                    int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, df_stream_t stream) {
  return upconv_wgrad_impl(xc, gy, gw, gb, B, Dc, Hc, Wc, Cin, Cout, kz, workspace, workspace_bytes, stream, 0, 0);
}
int df_upconv_wgrad_algo(const float* xc, const float* gy, float* gw, float* gb, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                         int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes, int algo, df_stream_t stream) {
  return upconv_wgrad_impl(xc, gy, gw, gb, B, Dc, Hc, Wc, Cin, Cout, kz, workspace, workspace_bytes, stream, 0, algo);
}
int df_upconv_wgrad_bf16x3(const float* xc, const float* gy, float* gw, float* gb, int64_t B, int64_t Dc, int64_t Hc,
                           int64_t Wc, int64_t Cin, int64_t Cout, int kz, void* workspace, int64_t workspace_bytes,
                           df_stream_t stream) {
  return upconv_wgrad_impl(xc, gy, gw, gb, B, Dc, Hc, Wc, Cin, Cout, kz, workspace, workspace_bytes, stream, 1, 0);
}

}  // extern "C"
