// VALU kernels for the thin layers at the end of the generator (model.py:42,84: conv 128 -> 1|3, no activation).
// A 32-wide MFMA column block would waste 29/32 of the matrix core on 3 output channels, so these run on the
// vector ALU (64-wide wavefronts, the tiny operand held in SGPRs / broadcast from LDS):
//
//   conv_small_n : Cout <= 4.  thread = output voxel; the halo'd input tile is staged per 16-channel chunk in LDS
//                  exactly as in conv.hip; the weights of a (tap, channel-quad) are 16 consecutive floats of the
//                  packed filter bank at a wave-uniform address -> scalar loads, FMAs take them as SGPR operands.
//   conv_small_k : Cin  <= 4 (the dgrad of the last layer, 3 -> 128).  thread = output channel with its 27x4
//                  filter taps in registers; the 4-float input records of the halo tile sit in LDS and are read
//                  at wave-uniform addresses (broadcast).  Full fused epilogue (bias, lrelu, residual, mask).
//   wgrad_small_n: gW[tap][ci][co<=4].  thread = input channel with all 27 x CO accumulators in registers; a wave
//                  walks image rows along x with an 8-deep register ring per (dz,dy) (9 coalesced 256-byte loads per
//                  voxel step, 5-6 steps ahead), the gradient record G[voxel][co] is wave-uniform (scalar loads).
#include "conv_args.hpp"

namespace dfconv {
namespace {

using df::ceil_div;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
template <int KZ, int TZ, int TY, int TX, int CO, bool VEC>
__global__ __launch_bounds__(kThreads) void conv_small_n_kernel(const ConvArgs a) {
  static_assert(TZ * TY * TX == kThreads, "one thread per output voxel");
  constexpr int PZ = KZ / 2;
  constexpr int HZ = TZ + KZ - 1, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int NPIECE = HV * (CK / 4);
  constexpr int NLOAD = (NPIECE + kThreads - 1) / kThreads;
  constexpr int NTAP = KZ * 9;
  constexpr int S4 = LDS_STRIDE / 4;
  __shared__ __attribute__((aligned(16))) float sA[HV * LDS_STRIDE];
  // the chunk's weights [tap][channel quad][n < 4][4]: 6.9 KB.  (They used to be SGPR operands fetched by 108 scattered scalar
  //  loads per chunk -- a 55 KB footprint that thrashes the 16 KB scalar cache; an LDS broadcast read has no such cliff.)
  __shared__ __attribute__((aligned(16))) f32x4 sWt[NTAP * 4 * 4];
  const f32x4* sA4 = reinterpret_cast<const f32x4*>(sA);

  const int tid = threadIdx.x;
  const int tile = xcd_tile(blockIdx.x, a.ntiles);
  const int ix = tile % a.nx;
  int t2 = tile / a.nx;
  const int iy = t2 % a.ny; t2 /= a.ny;
  const int iz = t2 % a.nz;
  const int b = t2 / a.nz;
  const int tz0 = iz * TZ, ty0 = iy * TY, tx0 = ix * TX;
  const int lx = tid % TX, ly = (tid / TX) % TY, lz = tid / (TX * TY);
  const int aidx = ((lz * HY + ly) * HX + lx) * S4;
  const int K8 = a.Kpad >> 3;

  // two partial sums per output channel (even / odd input channel of a pair): the inner product runs as packed fp32 FMAs
  f32x2 acc2[CO];
#pragma unroll
  for (int n = 0; n < CO; ++n) acc2[n] = f32x2{0.f, 0.f};

  const int nchunk = a.Kpad / CK;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    float4 stg[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int p = it * kThreads + tid;
      const int hv = p >> 2, q = p & 3;
      const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
      const int gz = tz0 + hz - PZ, gy = ty0 + hy - 1, gx = tx0 + hx - 1;
      const int ch = chunk * CK + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool inb = p < NPIECE && gz >= 0 && gz < a.Di && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi;
      if (inb) {
        const int64_t vox = ((static_cast<int64_t>(b) * a.D + gz) * a.H + gy) * a.W + gx;
        const float* src = a.x + vox * a.Cin + ch;
        if (VEC) {
          if (ch < a.Cin) v = *reinterpret_cast<const float4*>(src);
        } else {
          if (ch + 0 < a.Cin) v.x = src[0];
          if (ch + 1 < a.Cin) v.y = src[1];
          if (ch + 2 < a.Cin) v.z = src[2];
          if (ch + 3 < a.Cin) v.w = src[3];
        }
      }
      stg[it] = v;
    }
    f32x4 wst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = i * kThreads + tid;            // record (tap, q, n): r = (tap*4 + q)*4 + n
      if (r < NTAP * 16) {
        const int n = r & 3, q = (r >> 2) & 3, tap = r >> 4;
        const int k8 = chunk * 2 + (q >> 1), half = q & 1;
        wst[i] = a.wp[((static_cast<int64_t>(tap) * K8 + k8) * 2 + half) * a.Npad + n];
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int p = it * kThreads + tid;
      if (p < NPIECE) *reinterpret_cast<float4*>(&sA[(p >> 2) * LDS_STRIDE + (p & 3) * 4]) = stg[it];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i * kThreads + tid < NTAP * 16) sWt[i * kThreads + tid] = wst[i];
    __syncthreads();

#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int toff = ((dz * HY + dy) * HX + dx) * S4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 xv = sA4[aidx + toff + q];
        const f32x2 x01 = {xv[0], xv[1]}, x23 = {xv[2], xv[3]};
#pragma unroll
        for (int n = 0; n < CO; ++n) {
          const f32x4 wn = sWt[(tap * 4 + q) * 4 + n];                 // same address in every lane: LDS broadcast
          const f32x2 w01 = {wn[0], wn[1]}, w23 = {wn[2], wn[3]};
          acc2[n] = __builtin_elementwise_fma(x01, w01, acc2[n]);
          acc2[n] = __builtin_elementwise_fma(x23, w23, acc2[n]);
        }
      }
    }
  }
  float acc[CO];
#pragma unroll
  for (int n = 0; n < CO; ++n) acc[n] = acc2[n][0] + acc2[n][1];

  const int gz = tz0 + lz, gy = ty0 + ly, gx = tx0 + lx;
  if (gz < a.D && gy < a.H && gx < a.W) {
    const int64_t o = (((static_cast<int64_t>(b) * a.D + gz) * a.H + gy) * a.W + gx) * a.Cout;
#pragma unroll
    for (int n = 0; n < CO; ++n) {
      if (n < a.Cout) {
        float v = acc[n];
        if (a.flags & DF_CONV_BIAS) v += a.bias[n];
        if (a.flags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
        if (a.flags & DF_CONV_RESIDUAL) v += a.residual[o + n];
        if (a.flags & DF_CONV_MASK) v = a.mask_src[o + n] > 0.f ? v : a.leak * v;
        a.y[o + n] = v;
      }
    }
  }
}

template <int KZ, int TZ, int TY, int TX>
int launch_small_n_t(ConvArgs a, hipStream_t s) {
  a.nz = (int)ceil_div(a.D, TZ); a.ny = (int)ceil_div(a.H, TY); a.nx = (int)ceil_div(a.W, TX);
  const int64_t nt = static_cast<int64_t>(a.B) * a.nz * a.ny * a.nx;
  DF_REQUIRE(nt < (1LL << 31), DF_ESHAPE, "df_conv_fwd: too many tiles");
  a.ntiles = (int)nt;
  dim3 grid((unsigned)nt);
  const bool vec = (a.Cin % 4 == 0) && df::aligned16(a.x);
#define DF_SN(CO)                                                                                                  \
  if (vec) hipLaunchKernelGGL((conv_small_n_kernel<KZ, TZ, TY, TX, CO, true>), grid, dim3(kThreads), 0, s, a);     \
  else hipLaunchKernelGGL((conv_small_n_kernel<KZ, TZ, TY, TX, CO, false>), grid, dim3(kThreads), 0, s, a)
  switch (a.Cout) {
    case 1: DF_SN(1); break;
    case 2: DF_SN(2); break;
    case 3: DF_SN(3); break;
    default: DF_SN(4); break;
  }
#undef DF_SN
  return df::launched("df_conv_fwd(small-N)");
}

// ---------------------------------------------------------------------------------------------------------------
template <int KZ, int TZ, int TY, int TX>
__global__ __launch_bounds__(kThreads) void conv_small_k_kernel(const ConvArgs a, int nwn) {
  constexpr int NV = TZ * TY * TX;    // 128 voxels per workgroup
  constexpr int PZ = KZ / 2;
  constexpr int HZ = TZ + KZ - 1, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int NTAP = KZ * 9;
  __shared__ f32x4 sG[HV];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_tile(blockIdx.x, a.ntiles);
  const int ix = tile % a.nx;
  int t2 = tile / a.nx;
  const int iy = t2 % a.ny; t2 /= a.ny;
  const int iz = t2 % a.nz;
  const int b = t2 / a.nz;
  const int tz0 = iz * TZ, ty0 = iy * TY, tx0 = ix * TX;

  for (int hv = tid; hv < HV; hv += kThreads) {
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
    const int gz = tz0 + hz - PZ, gy = ty0 + hy - 1, gx = tx0 + hx - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (gz >= 0 && gz < a.Di && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi) {
      const float* src = a.x + (((static_cast<int64_t>(b) * a.D + gz) * a.H + gy) * a.W + gx) * a.Cin;
      v[0] = src[0];
      if (a.Cin > 1) v[1] = src[1];
      if (a.Cin > 2) v[2] = src[2];
      if (a.Cin > 3) v[3] = src[3];
    }
    sG[hv] = v;
  }

  // waves: nwn of them side by side over the output channels (64 each), the rest are independent voxel streams
  const int wn = wave % nwn, ws = wave / nwn, nstream = 4 / nwn;
  const int n = blockIdx.y * (nwn * 64) + wn * 64 + lane;
  const bool nok = n < a.Cout;
  f32x4 w[NTAP];                      // this output channel's taps: packed record [tap][k8=0][half=0][n][k]
#pragma unroll
  for (int tap = 0; tap < NTAP; ++tap)
    w[tap] = a.wp[(static_cast<int64_t>(tap) * (a.Kpad >> 3) * 2) * a.Npad + (nok ? n : 0)];
  const float bv = (a.flags & DF_CONV_BIAS) && nok ? a.bias[n] : 0.f;
  __syncthreads();

  for (int m = ws; m < NV; m += nstream) {
    const int lx = m % TX, ly = (m / TX) % TY, lz = m / (TX * TY);
    const int gz = tz0 + lz, gy = ty0 + ly, gx = tx0 + lx;
    if (gz >= a.D || gy >= a.H || gx >= a.W) continue;          // wave-uniform
    const int h0 = (lz * HY + ly) * HX + lx;
    f32x2 acc2 = {0.f, 0.f};            // packed fp32 FMAs: (k = 0, 2) and (k = 1, 3) partial sums
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const f32x4 g = sG[h0 + (dz * HY + dy) * HX + dx];          // wave-uniform address: LDS broadcast
      acc2 = __builtin_elementwise_fma(f32x2{g[0], g[1]}, f32x2{w[tap][0], w[tap][1]}, acc2);
      acc2 = __builtin_elementwise_fma(f32x2{g[2], g[3]}, f32x2{w[tap][2], w[tap][3]}, acc2);
    }
    const float acc = acc2[0] + acc2[1];
    if (nok) {
      const int64_t o = (((static_cast<int64_t>(b) * a.D + gz) * a.H + gy) * a.W + gx) * a.Cout + n;
      float v = acc + bv;
      if (a.flags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
      if (a.flags & DF_CONV_RESIDUAL) v += a.residual[o];
      if (a.flags & DF_CONV_MASK) v = a.mask_src[o] > 0.f ? v : a.leak * v;
      a.y[o] = v;
    }
  }
}

// ---- thin-K conv on the matrix cores (3-D, Cin <= 4 -> 128 channels: the dgrad of the generator's last layer) ------------------------
// im2col GEMM with the taps on the K side:  y[v, n] = sum_{(tap, c)} A[v, (tap, c)] * W[(tap, c), n],  A[v, (tap, c)] = x[v + tap - 1, c],
// K = 27*Cin (81 | 108) in steps of 2 for v_mfma_f32_32x32x2.  The whole filter bank lives in registers (K/2 x 2 values per lane), a
// wave walks image rows in 32-voxel chunks, and the shifted 3|4-channel input records are gathered from a 9-row LDS tile private to the
// wave (zero x-halo, zero rows for SAME padding, double-buffered per image row: no workgroup barriers).  Two waves share a row stream,
// 64 output channels each; N-block j of lane l holds channel 2*(l%32)+j of the wave's half, so every output / residual / mask access is
// a float2 and a wave instruction covers two voxels' 256-byte half rows.  [r3] CO = 64 (the auto-encoder's 3 -> 64 dgrad: one wave per row
// stream), rows of any multiple of 8 voxels >= 32 (the last chunk of a row then starts at W - 32: it recomputes up to 24 voxels of the one
// before and stores the same values again) and rows of up to 448 floats (NP = 2 load passes; LDS opt-in above 64 KB) -- cfg4's W = 112.
struct ThinKArgs {
  const float* x;
  const f32x4* wp;
  const float* bias;
  const float* residual;
  const float* mask_src;
  float* y;
  int D, H, W;
  int tapstride;       // float4 records between taps of the packed filter bank
  int nrows, rows_per, RS;
  int flags;
  float leak;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CI, int CO, int NP>
__global__ __launch_bounds__(kThreads, 1) void conv_thin_k_mfma_kernel(const ThinKArgs a) {
  constexpr int NK = 27 * CI, KS = (NK + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float smem_thin_k[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gwave = blockIdx.x * 4 + wave;
  // CO = 128: two waves per row stream, 64 output channels each (the filter bank fits the registers); CO = 64: one
  const int stream = CO == 128 ? gwave >> 1 : gwave, half = CO == 128 ? gwave & 1 : 0;
  const int RS = a.RS, W = a.W, WC = W * CI;
  float* sG = smem_thin_k + wave * 19 * RS;          // [2 buffers][9 (dz, dy) rows][RS] + one zero row
  const int r0 = stream * a.rows_per;
  int r1 = r0 + a.rows_per;
  if (r1 > a.nrows) r1 = a.nrows;
  for (int i = lane; i < 19 * RS; i += 64) sG[i] = 0.f;
  if (r0 >= r1) return;
  const int kk = lane >> 5, n4 = half * 64 + (lane & 31) * 2;      // this lane's channel pair

  // ---- filter bank -> registers: wr[s][j] = W[k = 2s + kk][channel n4 + j] -----------------------------------------------------------
  float wr[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + kk;
    const bool ok = k < NK;
    const int tap = ok ? k / CI : 0, c = ok ? k % CI : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 rec = a.wp[static_cast<int64_t>(tap) * a.tapstride + n4 + j];
      const float v = c == 0 ? rec[0] : c == 1 ? rec[1] : c == 2 ? rec[2] : rec[3];
      wr[s][j] = ok ? v : 0.f;
    }
  }
  f32x2 bias2 = {0.f, 0.f};
  if (a.flags & DF_CONV_BIAS) bias2 = *reinterpret_cast<const f32x2*>(a.bias + n4);

  // ---- input rows of image row `row` -> registers -> LDS buffer ----------------------------------------------------------------------
  const int nl4 = WC / 4;
  f32x4 gq[NP][9];
  auto load_g = [&](int row) {
    const int y = row % a.H;
    const int t = row / a.H;
    const int z = t % a.D;
    const int b = t / a.D;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int zs = z + (k / 3 - 1), ys = y + (k % 3 - 1);
      const bool ok = zs >= 0 && zs < a.D && ys >= 0 && ys < a.H;      // wave-uniform
      const int zc = ok ? zs : z, yc = ok ? ys : y;
      const float* src = a.x + ((static_cast<int64_t>(b) * a.D + zc) * a.H + yc) * WC;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (lane + 64 * p < nl4) v = *reinterpret_cast<const f32x4*>(src + (lane + 64 * p) * 4);
        gq[p][k] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto store_g = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (lane + 64 * p < nl4) {
#pragma unroll
        for (int k = 0; k < 9; ++k) *reinterpret_cast<f32x4*>(sG + (buf * 9 + k) * RS + 4 + (lane + 64 * p) * 4) = gq[p][k];
      }
  };

  // ---- A operand gather: lane (voxel m = l % 32 of the chunk, k = 2s + l / 32) ---------------------------------------------------------
  const char* sGb = reinterpret_cast<const char*>(sG);
  const int lanebase = (lane & 31) * CI * 4;
  auto gather = [&](int buf, int x0, float (&av)[KS]) {      // x0: first voxel of the chunk
    const int cb = lanebase + x0 * (CI * 4);
    const int bb = buf * (9 * RS * 4);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int o0 = ((k0 / CI / 3) * RS + 4 + ((k0 / CI) % 3 - 1) * CI + k0 % CI) * 4 + bb;
      const int o1 = k1 < NK ? ((k1 / CI / 3) * RS + 4 + ((k1 / CI) % 3 - 1) * CI + k1 % CI) * 4 + bb : (18 * RS + 4) * 4;
      av[s] = *reinterpret_cast<const float*>(sGb + cb + (kk ? o1 : o0));
    }
  };

  load_g(r0);
  store_g(0);
  const int nchunk = (W + 31) >> 5;
  int buf = 0;
  for (int row = r0; row < r1; ++row) {
    const bool more = row + 1 < r1;
    if (more) load_g(row + 1);
    const int64_t obase = static_cast<int64_t>(row) * W * CO + n4;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      float av[KS];
      const int x0 = chunk * 32 + 32 <= W ? chunk * 32 : W - 32;
      gather(buf, x0, av);
      const int64_t oc = obase + static_cast<int64_t>(x0) * CO;
      f32x2 mk[16];
      if (a.flags & DF_CONV_MASK) {      // requested now, used after the chunk's 2*KS MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[r] = *reinterpret_cast<const f32x2*>(a.mask_src + oc + ((r >> 2) * 8 + kk * 4 + (r & 3)) * CO);
      }
      f32x16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wr[s][j], acc[j], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t o = oc + ((r >> 2) * 8 + kk * 4 + (r & 3)) * CO;
        f32x2 v = f32x2{acc[0][r], acc[1][r]} + bias2;
        if (a.flags & DF_CONV_LRELU) {
#pragma unroll
          for (int j = 0; j < 2; ++j) v[j] = fmaxf(v[j], a.leak * v[j]);
        }
        if (a.flags & DF_CONV_RESIDUAL) v += *reinterpret_cast<const f32x2*>(a.residual + o);
        if (a.flags & DF_CONV_MASK) {
#pragma unroll
          for (int j = 0; j < 2; ++j) v[j] = mk[r][j] > 0.f ? v[j] : a.leak * v[j];
        }
        *reinterpret_cast<f32x2*>(a.y + o) = v;
      }
      if (chunk == 0 && more) store_g(buf ^ 1);
    }
    buf ^= 1;
  }
}

// (LDS: 4 waves x 19 rows x (W*Cin + 8) floats: opt-in above the 64 KB a launch gets by default)
inline bool thin_k_mfma_ok(const ConvArgs& a, int kz) {
  return kz == 3 && a.Cin <= 4 && (a.Cout == 128 || a.Cout == 64) && a.W % 8 == 0 && a.W >= 32 && a.W * a.Cin <= 448 && (a.W * a.Cin) % 16 == 0 &&
         static_cast<int64_t>(a.B) * a.D * a.H >= 4 && a.nclass == 1 && df::aligned16(a.x) && df::aligned16(a.y) &&
         (!(a.flags & DF_CONV_BIAS) || df::aligned16(a.bias)) && (!(a.flags & DF_CONV_RESIDUAL) || df::aligned16(a.residual)) &&
         (!(a.flags & DF_CONV_MASK) || df::aligned16(a.mask_src)) &&
         static_cast<int64_t>(4) * 19 * (a.W * a.Cin + 8) * 4 <= df::lds_optin_bytes();      // else: the vector-ALU kernel
}

int launch_thin_k_mfma(const ConvArgs& a, hipStream_t s) {
  ThinKArgs t;
  t.x = a.x; t.wp = a.wp; t.bias = a.bias; t.residual = a.residual; t.mask_src = a.mask_src; t.y = a.y;
  t.D = a.D; t.H = a.H; t.W = a.W;
  t.tapstride = (a.Kpad >> 3) * 2 * a.Npad;
  t.nrows = a.B * a.D * a.H;
  t.RS = a.W * a.Cin + 8;
  const size_t lds = static_cast<size_t>(4) * 19 * t.RS * sizeof(float);
  const int wps = a.Cout == 128 ? 2 : 1;                       // waves per row stream
  int ns = 4 / wps * df::kCUs;                                 // row streams: one workgroup (4 waves) per CU
  if (ns > t.nrows) ns = t.nrows / 2 * 2;
  t.rows_per = (t.nrows + ns - 1) / ns;
  t.flags = a.flags; t.leak = a.leak;
  dim3 grid((unsigned)ceil_div(ceil_div(t.nrows, t.rows_per) * wps, 4));
  const bool two = a.W * a.Cin > 256;                          // rows of more than 64 float4: two load passes
#define DF_TK2(CI, COV, NPV)                                                                                                                \
  do {                                                                                                                                      \
    if (lds > 64 * 1024)                                                                                                                    \
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_thin_k_mfma_kernel<CI, COV, NPV>),                         \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))                                         \
        return df::fail((int)e, "df_conv_fwd(thin-K mfma): dynamic LDS opt-in: %s", hipGetErrorString(e));                                  \
    hipLaunchKernelGGL((conv_thin_k_mfma_kernel<CI, COV, NPV>), grid, dim3(kThreads), lds, s, t);                                           \
  } while (0)
#define DF_TK(CI)                                                                                                                           \
  do {                                                                                                                                      \
    if (a.Cout == 128) { if (two) DF_TK2(CI, 128, 2); else DF_TK2(CI, 128, 1); }                                                            \
    else { if (two) DF_TK2(CI, 64, 2); else DF_TK2(CI, 64, 1); }                                                                            \
  } while (0)
  switch (a.Cin) {
    case 1: DF_TK(1); break;
    case 2: DF_TK(2); break;
    case 3: DF_TK(3); break;
    default: DF_TK(4); break;
  }
#undef DF_TK
#undef DF_TK2
  return df::launched("df_conv_fwd(thin-K mfma)");
}

// ---- thin-N conv on the matrix cores (3-D, 128 -> Cout <= 3: the generator's last layer, forward) ---------------------------------
// The in-plane taps go to the N side and the z taps to the K side of one GEMM per input row:
//   T[(y', x'), (dy, dx, co)] = sum_{dz, c} x[z + dz - 1, y', x', c] * W[dz, dy, dx][c][co]        (N = 9*Cout <= 27 of 32, K = 3*128)
//   y[z, y, x, co] = b[co] + sum_{dy, dx} T[(y + dy - 1, x + dx - 1), (dy, dx, co)]
// A wave owns a range of rows of ONE (b, z) plane and walks it row by row in 32-voxel chunks: 48 float4 loads per lane (each 128-byte
// line of x is consumed by four back-to-back loads) feed 192 v_mfma_f32_32x32x2 whose filter operands all stay in registers; the
// 9-way (dy, dx) shift-add of the product tile goes through a three-row LDS ring private to the wave (ds_add_f32 in program order:
// deterministic), and a finished output row leaves as 3*W contiguous floats.  x is read by the three waves of neighbouring planes
// at about the same time (L2 / Infinity Cache), 16 loads in flight per lane.  [r3] CIN = 64 (the auto-encoder's 64 -> 3 layer: K = 3*64, 8
// loads in flight), rows of any multiple of 8 voxels >= 32 (the last chunk of a row then starts at W - 32 and the lanes of the voxels the
// chunk before already covered load through an out-of-range offset: zeros, their products add nothing) and output rows of up to 512
// floats (two store passes) -- cfg4's W = 112.
struct ThinNArgs {
  const float* x;
  const f32x4* wp;
  const float* bias;
  float* y;
  int B, D, H, W;
  int K8, Npad;            // packed filter bank geometry: record index ((tap*K8 + c/8)*2 + (c/4)%2)*Npad + n, element c%4
  int nsplit, rows_per;    // y ranges per plane
  int flags;
  float leak;
};

template <int CO, int CIN>
__global__ __launch_bounds__(kThreads, 1) void conv_thin_n_mfma_kernel(const ThinNArgs a) {
  constexpr int NC = 9 * CO;
  constexpr int QN = CIN / 8;        // 8-channel items per z tap (lane half kk takes 4 of the 8)
  constexpr int NIT = 3 * QN;
  extern __shared__ __attribute__((aligned(16))) float smem_thin_n[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroup g runs on XCD g % 8: give every XCD a contiguous run of planes, so the three readers of a plane share one L2
  const int stream = xcd_tile(blockIdx.x, gridDim.x) * 4 + wave;
  const int W = a.W, H = a.H, RS = W * CO + 8;
  float* ring = smem_thin_n + wave * 3 * RS;       // [3 output rows][4 + W*CO + 4]: position of (x, co) = 4 + x*CO + co
  const int plane = stream / a.nsplit, part = stream % a.nsplit;
  if (plane >= a.B * a.D) return;
  const int y0 = part * a.rows_per;
  int y1 = y0 + a.rows_per;
  if (y1 > H) y1 = H;
  if (y0 >= y1) return;
  const int z = plane % a.D, b = plane / a.D;
  for (int i = lane; i < 3 * RS; i += 64) ring[i] = 0.f;
  const int n = lane & 31, kk = lane >> 5;

  // ---- filter bank -> registers: item (dz, q) holds W[dz, dy(n), dx(n)][c = 8q + 4kk + t][co(n)], t = 0..3 ------------------------
  f32x4 wr[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int dz = it / QN, q = it % QN;
    const int tap = dz * 9 + (n < NC ? n / CO : 0);
    const f32x4 rec = a.wp[((static_cast<int64_t>(tap) * a.K8 + q) * 2 + kk) * a.Npad + (n < NC ? n % CO : 0)];
    wr[it] = n < NC ? rec : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // output-side constants of this lane's column
  const int dy = n / (3 * CO), dx = (n / CO) % 3, co = n % CO;
  float bias3[CO];
#pragma unroll
  for (int j = 0; j < CO; ++j) bias3[j] = (a.flags & DF_CONV_BIAS) ? a.bias[j] : 0.f;

  // ---- x loads: batch volume as a buffer; lane (voxel m = n, half kk) reads channels 8q + 4kk .. +3 ----------------------------------
  const int64_t vol = static_cast<int64_t>(a.D) * H * W * CIN;
  const __amdgpu_buffer_rsrc_t xsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + static_cast<int64_t>(b) * vol), 0, static_cast<unsigned>(vol * 4), 0x00020000);
  const unsigned lanev = static_cast<unsigned>(n) * (CIN * 4u) + static_cast<unsigned>(kk) * 16u;
  const int nchunk = (W + 31) >> 5;
  const int xlast = W - 32;                                                 // first voxel of a row's last chunk (== (nchunk - 1) * 32 when W % 32 == 0)
  const unsigned lastbit = xlast + n >= (nchunk - 1) * 32 ? 0u : 0x80000000u;      // ... whose already-covered voxels read zeros
  unsigned vplane[3];      // per z tap: lane offset, or an out-of-range offset (reads zeros) when the plane is outside the volume
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) vplane[dz] = (z + dz - 1 >= 0 && z + dz - 1 < a.D) ? lanev : 0x80000000u;
  auto soff_of = [&](int dz, int row, int chunk) -> unsigned {      // wave-uniform byte offset of (plane, row, chunk); clamped plane
    int zz = z + dz - 1;
    zz = zz < 0 ? 0 : (zz >= a.D ? a.D - 1 : zz);
    return static_cast<unsigned>(((zz * H + row) * W + (chunk + 1 == nchunk ? xlast : chunk * 32)) * (CIN * 4));
  };
  auto load_x = [&](unsigned voff, unsigned soff, int q) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrd, voff + static_cast<unsigned>(q) * 32u, soff, 0));
  };

  const int ra = y0 > 0 ? y0 - 1 : 0, rb = y1 < H ? y1 + 1 : H;      // input rows [ra, rb)
  f32x4 xr[QN];
  {
    const unsigned s0 = soff_of(0, ra, 0);
    const unsigned v0 = vplane[0] | (nchunk == 1 ? lastbit : 0u);
#pragma unroll
    for (int i = 0; i < QN; ++i) xr[i] = load_x(v0, s0, i);
  }

  typedef float f32x16 __attribute__((ext_vector_type(16)));
  for (int row = ra; row < rb; ++row) {
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      // where the loads issued during this chunk go: planes 1, 2 of this chunk, then plane 0 of the next chunk / row
      const bool lastc = chunk + 1 == nchunk;
      const int nrow = lastc ? row + 1 : row, nchk = lastc ? 0 : chunk + 1;
      const bool more = nrow < rb;
      const unsigned s1 = soff_of(1, row, chunk), s2 = soff_of(2, row, chunk);
      const unsigned sn = soff_of(0, more ? nrow : row, more ? nchk : chunk);
      const unsigned vn = more ? (vplane[0] | (nchk == nchunk - 1 ? lastbit : 0u)) : 0x80000000u;
      const unsigned cbit = lastc ? lastbit : 0u;
      const unsigned v1 = vplane[1] | cbit, v2 = vplane[2] | cbit;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const f32x4 xv = xr[it % QN];
        const int nit = it + QN;      // the item this slot is refilled for
        if (nit < 2 * QN) xr[it % QN] = load_x(v1, s1, nit % QN);
        else if (nit < 3 * QN) xr[it % QN] = load_x(v2, s2, nit % QN);
        else xr[it % QN] = load_x(vn, sn, nit % QN);
        __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks every load to just before its use: vmcnt(0) per item)
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[0], wr[it][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[1], wr[it][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[2], wr[it][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[3], wr[it][3], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- shift-add: column (dy, dx, co) of input row `row` belongs to output row row - (dy - 1), position x' - (dx - 1) --------
      const int orow = row + 1 - dy;
      if (n < NC && orow >= y0 && orow < y1) {
        float* dst = ring + (orow % 3) * RS + 4 + ((lastc ? xlast : chunk * 32) + kk * 4 + 1 - dx) * CO + co;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __hip_atomic_fetch_add(dst + ((r >> 2) * 8 + (r & 3)) * CO, acc0[r] + acc1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    // ---- output rows completed by this input row: row - 1, and row itself when it is the last one of the plane --------------------
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int fr = k == 0 ? row - 1 : row;
      const bool go = k == 0 ? (fr >= y0 && fr < y1) : (row == H - 1 && fr >= y0 && fr < y1);
      if (!go) continue;
      float* src = ring + (fr % 3) * RS + 4;
      for (int i0 = lane * 4; i0 < W * CO; i0 += 256) {      // one pass up to 64 x 4 floats per row, two above
        f32x4 v = *reinterpret_cast<const f32x4*>(src + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (i0 + j) % CO;
          v[j] += CO == 1 ? bias3[0] : c == 0 ? bias3[0] : c == 1 ? bias3[1 % CO] : bias3[2 % CO];
        }
        if (a.flags & DF_CONV_LRELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], a.leak * v[j]);
        }
        *reinterpret_cast<f32x4*>(a.y + ((static_cast<int64_t>(plane) * H + fr) * W) * CO + i0) = v;
        *reinterpret_cast<f32x4*>(src + i0) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
}

inline bool thin_n_mfma_ok(const ConvArgs& a, int kz) {
  return kz == 3 && (a.Cin == 128 || a.Cin == 64) && a.Cout <= 3 && a.W % 8 == 0 && a.W >= 32 && a.W * a.Cout <= 512 && (a.W * a.Cout) % 4 == 0 &&
         a.nclass == 1 && !(a.flags & (DF_CONV_RESIDUAL | DF_CONV_MASK)) && df::aligned16(a.x) && df::aligned16(a.y) &&
         static_cast<int64_t>(a.D) * a.H * a.W * a.Cin * 4 < (1LL << 31);
}

int launch_thin_n_mfma(const ConvArgs& a, hipStream_t s) {
  ThinNArgs t;
  t.x = a.x; t.wp = a.wp; t.bias = a.bias; t.y = a.y;
  t.B = a.B; t.D = a.D; t.H = a.H; t.W = a.W;
  t.K8 = a.Kpad >> 3; t.Npad = a.Npad;
  const int64_t planes = static_cast<int64_t>(a.B) * a.D;
  // y ranges per plane: one workgroup (4 waves = 4 ranges) per CU and round, so the launch takes rounds x (rows of a range + its two halo rows);
  // [r5] pick the split that minimises that product instead of the smallest one that fills the chip (cfg4, 448 planes: 3 ranges per plane were
  // 336 workgroups = two rounds of 54-row ranges; 2 ranges are one round of 80-row ranges)
  int nsplit = 1;
  {
    int64_t best = -1;
    const int smax = a.H / 4 > 1 ? a.H / 4 : 1;
    for (int sp = 1; sp <= smax && sp <= 64; ++sp) {
      const int64_t rows = ceil_div(a.H, sp);
      const int64_t cost = ceil_div(ceil_div(planes * ceil_div(a.H, rows), 4), df::kCUs) * (rows + 2);
      if (best < 0 || cost < best) { best = cost; nsplit = sp; }
    }
  }
  t.rows_per = (int)ceil_div(a.H, nsplit);
  t.nsplit = (int)ceil_div(a.H, t.rows_per);
  t.flags = a.flags; t.leak = a.leak;
  const size_t lds = static_cast<size_t>(4) * 3 * (a.W * a.Cout + 8) * sizeof(float);
  dim3 grid((unsigned)ceil_div(planes * t.nsplit, 4));
#define DF_TN(CO)                                                                                                  \
  do {                                                                                                             \
    if (a.Cin == 128) hipLaunchKernelGGL((conv_thin_n_mfma_kernel<CO, 128>), grid, dim3(kThreads), lds, s, t);     \
    else hipLaunchKernelGGL((conv_thin_n_mfma_kernel<CO, 64>), grid, dim3(kThreads), lds, s, t);                   \
  } while (0)
  switch (a.Cout) {
    case 1: DF_TN(1); break;
    case 2: DF_TN(2); break;
    default: DF_TN(3); break;
  }
#undef DF_TN
  return df::launched("df_conv_fwd(thin-N mfma)");
}

template <int KZ, int TZ, int TY, int TX>
int launch_small_k_t(ConvArgs a, hipStream_t s) {
  a.nz = (int)ceil_div(a.D, TZ); a.ny = (int)ceil_div(a.H, TY); a.nx = (int)ceil_div(a.W, TX);
  const int64_t nt = static_cast<int64_t>(a.B) * a.nz * a.ny * a.nx;
  DF_REQUIRE(nt < (1LL << 31), DF_ESHAPE, "df_conv_fwd: too many tiles");
  a.ntiles = (int)nt;
  const int nwn = a.Cout > 128 ? 4 : (a.Cout > 64 ? 2 : 1);
  dim3 grid((unsigned)nt, (unsigned)ceil_div(a.Cout, nwn * 64));
  hipLaunchKernelGGL((conv_small_k_kernel<KZ, TZ, TY, TX>), grid, dim3(kThreads), 0, s, a, nwn);
  return df::launched("df_conv_fwd(small-K)");
}

}  // namespace

int launch_small_n(const ConvArgs& a, int kz, hipStream_t s) {
  if (thin_n_mfma_ok(a, kz) && !(a.flags & DF_CONV_VALU_ONLY)) return launch_thin_n_mfma(a, s);
  if (kz == 3) return a.W >= 12 ? launch_small_n_t<3, 4, 4, 16>(a, s) : launch_small_n_t<3, 4, 8, 8>(a, s);
  return a.W >= 12 ? launch_small_n_t<1, 1, 16, 16>(a, s) : launch_small_n_t<1, 1, 32, 8>(a, s);
}

int launch_small_k(const ConvArgs& a, int kz, hipStream_t s) {
  if (thin_k_mfma_ok(a, kz) && !(a.flags & DF_CONV_VALU_ONLY)) return launch_thin_k_mfma(a, s);
  if (kz == 3) return a.W >= 12 ? launch_small_k_t<3, 2, 4, 16>(a, s) : launch_small_k_t<3, 4, 4, 8>(a, s);
  return a.W >= 12 ? launch_small_k_t<1, 1, 8, 16>(a, s) : launch_small_k_t<1, 1, 16, 8>(a, s);
}

}  // namespace dfconv

