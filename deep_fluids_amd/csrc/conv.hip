// 3x3 / 3x3x3 SAME convolution, channels-last fp32, on the gfx950 matrix cores
// (reference: slim.conv2d / slim.conv3d behind ops.py:12-16, called from model.py:26,42,68,84).
//
// Implicit GEMM   D[voxel][cout] = sum_{tap,cin} X[voxel + off(tap)][cin] * W[tap][cin][cout]
// with v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles per instruction per SIMD,
// 157 TFLOP/s chip peak -- the reference computes in fp32, so does this path.
//
// Workgroup = 256 threads (4 waves) = one spatial tile of 128 output voxels (TZ x TY x TX) times an
// N-tile of 32/64/128 output channels.  The K loop runs over chunks of 16 input channels:
//   * A operand: the tile's HALO'd input block [TZ+kz-1][TY+2][TX+2][16ch] is staged once per chunk in
//     LDS (zero-filled outside the image = SAME padding); all 9/27 taps then read it at constant
//     offsets.  Row stride 20 floats: a wave's ds_read_b128 (lane = voxel, 4 consecutive channels)
//     is bank-conflict-free (20/4 odd -> the 16 lanes of a b128 group hit 16 distinct 16-byte slots).
//     Lanes 0-31 read channels c..c+3 and lanes 32-63 channels c+4..c+7 of their voxel, so ONE
//     ds_read_b128 feeds FOUR MFMAs (k = lane>>5 selects the channel quad).
//   * B operand: weights are re-packed once per update (df_conv_pack_weights) to
//     Wp[tap][cin/8][half][cout][4] so that the same lane->k assignment is ONE coalesced 16-byte
//     global load per four MFMAs, served from L2 (the whole filter bank is 1.7 MB);  no LDS, no
//     barrier on the weight stream.
//   => per wave and step: MB ds_read_b128 + NB global_load_dwordx4 for 4*MB*NB MFMAs (16 at 2x2),
//      one barrier pair per 16-channel chunk (every 27*32 = 864 MFMAs per wave in 3-D).
// Epilogue (fused): bias, lrelu, residual add, lrelu-backward mask (dgrad).
// The dgrad is the same kernel on mirrored/transposed packed weights (mode 1).
#include "df_common.hpp"
#include "conv_args.hpp"

namespace {

using df::ceil_div;
using namespace dfconv;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- weight packing ---------------------------------------------------------------------------------
// mode 0: Wp[tap][k8][half][n][s] = w[tap][k8*8+half*4+s][n]            (K = cin,  N = cout)
// mode 1: Wp[tap][k8][half][n][s] = w[T-1-tap][n][k8*8+half*4+s]        (K = cout, N = cin; taps mirrored)
__global__ __launch_bounds__(kThreads) void pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int taps,
                                                        int cin, int cout, int Kpad, int Npad, int mode) {
  const int64_t total = static_cast<int64_t>(taps) * Kpad * Npad;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int s = static_cast<int>(i & 3);
    int64_t r = i >> 2;
    const int n = static_cast<int>(r % Npad); r /= Npad;
    const int half = static_cast<int>(r & 1); r >>= 1;
    const int k8 = static_cast<int>(r % (Kpad / 8));
    const int tap = static_cast<int>(r / (Kpad / 8));
    const int k = k8 * 8 + half * 4 + s;
    const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
    float v = 0.f;
    if (k < K && n < N) {
      v = mode == 0 ? w[(static_cast<int64_t>(tap) * cin + k) * cout + n]
                    : w[(static_cast<int64_t>(taps - 1 - tap) * cin + n) * cout + k];
    }
    wp[i] = v;
  }
}

// ---- weights of the up-sampling-aware conv ---------------------------------------------------------------------------
// conv(nearest_up2x(Yc), W)[2m+p] = sum_{d in {0,1}^n} Yc[m + d + p - 1] * Wc[p][d]   with, per axis,
//   p = 0: d=0 <- {k0}, d=1 <- {k1,k2};   p = 1: d=0 <- {k0,k1}, d=1 <- {k2}      (k = tap of the 3-wide filter)
// i.e. 8 (3-D) / 4 (2-D) parity classes of a 2x2x2 / 2x2-tap conv on the COARSE grid: 27/8 = 3.4x fewer FLOPs than
// convolving the materialised up-sampled tensor.  Packed per class like pack_kernel: [class][tap][k8][half][n][4];
// mode 1 = dgrad operand (taps mirrored, Cin/Cout swapped).
__device__ __forceinline__ bool in_set(int p, int d, int k) {   // does original tap k feed class-p tap d (one axis)?
  return p == 0 ? (d == 0 ? k == 0 : k >= 1) : (d == 0 ? k <= 1 : k == 2);
}
__global__ __launch_bounds__(kThreads) void upconv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int kz,
                                                               int cin, int cout, int Kpad, int Npad, int mode) {
  const int ncls = kz == 3 ? 8 : 4, ntap = kz == 3 ? 8 : 4;
  const int64_t per_class = static_cast<int64_t>(ntap) * Kpad * Npad;
  const int64_t total = per_class * ncls;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i / per_class);
    int64_t r = i - c * per_class;
    const int s = static_cast<int>(r & 3); r >>= 2;
    const int n = static_cast<int>(r % Npad); r /= Npad;
    const int half = static_cast<int>(r & 1); r >>= 1;
    const int k8 = static_cast<int>(r % (Kpad / 8));
    int tap = static_cast<int>(r / (Kpad / 8));
    const int k = k8 * 8 + half * 4 + s;
    const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
    float v = 0.f;
    if (mode == 2) {
      // dgrad operand of the STRIDE-2 conv (TF SAME on even extents: y[o] = sum_t x[2o + t] w[t]):
      //   dx[2m]   = g[m-1] w[2] + g[m] w[0]     -> class p = 0, taps d = 0 (offset -1), d = 1 (offset 0)
      //   dx[2m+1] = g[m] w[1]                    -> class p = 1, tap  d = 0 (offset 0);  d = 1 (offset +1) is zero
      // per axis -- exactly the class / offset structure of the up-sampling-aware forward conv, with K = cout, N = cin.
      if (k < K && n < N) {
        const int dzt = kz == 3 ? (tap >> 2) & 1 : 0, dyt = (tap >> 1) & 1, dxt = tap & 1;
        const int pz = kz == 3 ? (c >> 2) & 1 : 0, py = (c >> 1) & 1, px = c & 1;
        auto src = [](int p, int d) { return p == 0 ? (d == 0 ? 2 : 0) : (d == 0 ? 1 : -1); };
        const int tz = kz == 3 ? src(pz, dzt) : 0, ty = src(py, dyt), tx = src(px, dxt);
        if (tz >= 0 && ty >= 0 && tx >= 0) v = w[(static_cast<int64_t>((tz * 3 + ty) * 3 + tx) * cin + n) * cout + k];
      }
      wp[i] = v;
      continue;
    }
    if (k < K && n < N) {
      if (mode == 1) tap = ntap - 1 - tap;                       // mirrored taps
      const int dzt = kz == 3 ? (tap >> 2) & 1 : 0, dyt = (tap >> 1) & 1, dxt = tap & 1;
      const int pz = kz == 3 ? (c >> 2) & 1 : 0, py = (c >> 1) & 1, px = c & 1;
      const int ci = mode == 0 ? k : n, co = mode == 0 ? n : k;
      for (int z = 0; z < kz; ++z)
        for (int y = 0; y < 3; ++y)
          for (int x = 0; x < 3; ++x)
            if ((kz == 1 || in_set(pz, dzt, z)) && in_set(py, dyt, y) && in_set(px, dxt, x))
              v += w[(static_cast<int64_t>((z * 3 + y) * 3 + x) * cin + ci) * cout + co];
    }
    wp[i] = v;
  }
}

inline int ntile_for(int64_t N) { return N > 64 ? 128 : (N > 32 ? 64 : 32); }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---- main kernel --------------------------------------------------------------------------------------
// S = stride (1 | 2).  Stride 2 follows TF 'SAME' on even extents: pad 0 before / 1 after, i.e.
// out[o] = sum_k in[2o + k] w[k]; the LDS tile is the (2T+1)-wide input footprint of the output tile.
// KT = taps per in-plane axis (3; 2 for the parity-class convs of an up-sampled input), KZ = taps along z.
// CKT = input channels per LDS chunk (16; 32 for the 8-tap parity-class convs, whose 16-channel chunks are only 16 steps long)
// KTX = taps along x when it differs from KT (= taps along y); P8: the packed weights of a class are laid out for 2x2(x2) taps
// (tap id dz*4 + dy*2 + dx) although this instantiation walks only the KZ x KT x KTX taps that are not identically zero -- the
// parity classes of the STRIDE-2 dgrad (pack mode 2), whose odd-parity axes have ONE tap: 27 live taps of the 64 stored.
template <int KZ, int TZ, int TY, int TX, int WM, int WN, int MB, int NB, bool VEC, int S, int KT, int CKT = CK, int KTX = KT, bool P8 = false>
__global__ __launch_bounds__(kThreads) void conv_mfma_kernel(const ConvArgs a_in) {
  static_assert(TZ * TY * TX == 128 && WM * MB * 32 == 128 && WM * WN == 4, "tile shape");
  constexpr int HZ = (TZ - 1) * S + KZ, HY = (TY - 1) * S + KT, HX = (TX - 1) * S + KTX, HV = HZ * HY * HX;
  constexpr int LSTR = CKT + 4;             // LDS row stride (floats): (CKT+4)/4 odd -> conflict-free ds_read_b128
  constexpr int QPV = CKT / 4;              // float4 pieces per staged voxel
  constexpr int NPIECE = HV * QPV;
  constexpr int NLOAD = (NPIECE + kThreads - 1) / kThreads;
  constexpr int LBATCH = NLOAD < 8 ? NLOAD : 8;        // staging loads in flight per thread (bounds the registers)
  constexpr int NTAP = KZ * KT * KTX;
  constexpr int NTILE = WN * NB * 32;
  __shared__ __attribute__((aligned(16))) float sA[HV * LSTR];

  ConvArgs a = a_in;
  if (a.nclass > 1) {          // parity class of the up-sampling-aware conv (wave-uniform)
    const int c = blockIdx.z;
    const int bz = KZ > 1 ? (c >> 2) & 1 : 0, by = (c >> 1) & 1, bx = c & 1;
    a.pz = KZ > 1 ? 1 - bz : 0; a.py = 1 - by; a.px = 1 - bx;
    a.oz = bz; a.oy = by; a.ox = bx;
    a.wp += static_cast<int64_t>(c) * a.wclass;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, r = lane & 31;

  // XCD-aware, bijective tile mapping: workgroup b runs on XCD b % 8 (observed); give each XCD a contiguous
  // run of tiles so that neighbouring tiles (which share halos) share an L2.  Speed only, never correctness.
  int tile;
  {
    const int bid = blockIdx.x, q = a.ntiles >> 3, rem = a.ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int ix = tile % a.nx;
  int t2 = tile / a.nx;
  const int iy = t2 % a.ny; t2 /= a.ny;
  const int iz = t2 % a.nz;
  const int b = t2 / a.nz;
  const int tz0 = iz * TZ, ty0 = iy * TY, tx0 = ix * TX;
  const int n0 = blockIdx.y * NTILE;

  // per-lane LDS index (16-byte units: 5 per staged voxel) of this lane's voxel row in each M block,
  // at tap (0,0,0), channel quad `half`
  const f32x4* sA4 = reinterpret_cast<const f32x4*>(sA);
  constexpr int S4 = LSTR / 4;
  int aidx[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = (wm * MB + mb) * 32 + r;
    const int lx = m % TX, ly = (m / TX) % TY, lz = m / (TX * TY);
    aidx[mb] = ((lz * S * HY + ly * S) * HX + lx * S) * S4 + half;
  }
  // packed weights (float4 units): ((tapk8 * 2 + half) * Npad + col) = wave-uniform base + 32-bit per-lane offset, so every
  // B load is `global_load_dwordx4 v, v_off, s[base]` and no per-step 64-bit address lives in VGPRs
  const int64_t bstep = 2LL * a.Npad;            // float4s per (tap,k8) record
  const int K8 = a.Kpad >> 3;
  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) boff[nb] = half * a.Npad + (wn * NB + nb) * 32 + r;
  const f32x4* wbase = a.wp + n0;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;

  const int nchunk = a.Kpad / CKT;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    // ---- stage the halo'd input block of this 16-channel chunk ------------------------------------
    auto stage_load = [&](int it) -> float4 {
      const int p = it * kThreads + tid;
      const int hv = p / QPV, q = p % QPV;
      const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
      const int gz = tz0 * S + hz - a.pz, gy = ty0 * S + hy - a.py, gx = tx0 * S + hx - a.px;
      const int ch = chunk * CKT + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool inb = p < NPIECE && gz >= 0 && gz < a.Di && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi;
      if (inb) {
        const int64_t vox = ((static_cast<int64_t>(b) * a.xD + (gz * a.is + a.iz)) * a.xH + (gy * a.is + a.iy)) * a.xW +
                            (gx * a.is + a.ix);
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
      return v;
    };
    auto stage_store = [&](int it, const float4& v) {
      const int p = it * kThreads + tid;
      if (p < NPIECE) *reinterpret_cast<float4*>(&sA[(p / QPV) * LSTR + (p % QPV) * 4]) = v;
    };
    if (NLOAD <= LBATCH) {
      float4 stg[LBATCH];
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) stg[it] = stage_load(it);
      __syncthreads();   // every wave has finished reading the previous chunk
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) stage_store(it, stg[it]);
    } else {
      __syncthreads();
      for (int it0 = 0; it0 < NLOAD; it0 += LBATCH) {
        float4 stg[LBATCH];
#pragma unroll
        for (int i = 0; i < LBATCH; ++i) stg[i] = stage_load(it0 + i);
#pragma unroll
        for (int i = 0; i < LBATCH; ++i) stage_store(it0 + i, stg[i]);
      }
    }
    __syncthreads();

    // ---- 9 / 27 taps x 2 channel-octets, software-pipelined by one step ---------------------------
    constexpr int C8 = CKT / 8;             // channel octets per chunk
    const f32x4* bchunk = wbase + static_cast<int64_t>(chunk * C8) * bstep;      // wave-uniform
    const int64_t tapstride = static_cast<int64_t>(K8) * bstep;

    // A fragments: one step ahead (LDS);  B fragments: two steps ahead (L2), ring of three
    f32x4 af[2][MB], bf[3][NB];
    auto lds_a = [&](int step, f32x4 (&dst)[MB]) {
      const int tap = step / C8, c8 = step % C8;
      const int dz = tap / (KT * KTX), dy = (tap / KTX) % KT, dx = tap % KTX;
      const int toff = ((dz * HY + dy) * HX + dx) * S4 + c8 * 2;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) dst[mb] = sA4[aidx[mb] + toff];
    };
    auto glb_b = [&](int step, f32x4 (&dst)[NB]) {
      const int tap = step / C8, c8 = step % C8;
      const int wtap = P8 ? (tap / (KT * KTX)) * 4 + ((tap / KTX) % KT) * 2 + tap % KTX : tap;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dst[nb] = (bchunk + (wtap * tapstride + c8 * bstep))[boff[nb]];
    };
    glb_b(0, bf[0]);
    glb_b(1, bf[1]);
    lds_a(0, af[0]);

#pragma unroll
    for (int step = 0; step < NTAP * C8; ++step) {
      const int ca = step & 1, cb = step % 3;
      if (step + 2 < NTAP * C8) glb_b(step + 2, bf[(step + 2) % 3]);
      if (step + 1 < NTAP * C8) lds_a(step + 1, af[ca ^ 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetches at the head of the step
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ca][mb][s], bf[cb][nb][s], acc[mb][nb], 0, 0, 0);
    }
  }

  // ---- epilogue: D layout col = lane&31 (cout), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (voxel) ----------
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = n0 + (wn * NB + nb) * 32 + r;
    const bool colok = col < a.Cout;
    const float bv = (a.flags & DF_CONV_BIAS) && colok ? a.bias[col] : 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = (wm * MB + mb) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const int lx = m % TX, ly = (m / TX) % TY, lz = m / (TX * TY);
        const int gz = tz0 + lz, gy = ty0 + ly, gx = tx0 + lx;
        if (colok && gz < a.D && gy < a.H && gx < a.W) {
          const int64_t o = (((static_cast<int64_t>(b) * a.yD + (gz * a.os + a.oz)) * a.yH + (gy * a.os + a.oy)) * a.yW +
                             (gx * a.os + a.ox)) * a.Cout + col;
          float v = acc[mb][nb][e] + bv;
          if (a.flags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
          if (a.flags & DF_CONV_RESIDUAL) v += a.residual[o];
          if (a.flags & DF_CONV_MASK) v = a.mask_src[o] > 0.f ? v : a.leak * v;
          a.y[o] = v;
        }
      }
    }
  }
}

template <int KZ, int TZ, int TY, int TX, int WM, int WN, int MB, int NB, int S, int KT = 3>
int launch(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  a.nz = (int)ceil_div(a.D, TZ); a.ny = (int)ceil_div(a.H, TY); a.nx = (int)ceil_div(a.W, TX);
  const int64_t nt = static_cast<int64_t>(a.B) * a.nz * a.ny * a.nx;
  DF_REQUIRE(nt < (1LL << 31), DF_ESHAPE, "df_conv_fwd: too many tiles");
  a.ntiles = (int)nt;
  dim3 grid((unsigned)nt, (unsigned)(a.Npad / (WN * NB * 32)), (unsigned)(a.nclass > 1 ? a.nclass : 1));
  const bool vec = (a.Cin % 4 == 0) && df::aligned16(a.x);
  if (KT == 2 && KZ == 2 && vec && a.Kpad % 32 == 0)     // 8-tap parity-class convs: 32-channel chunks (half the barriers per MFMA)
    hipLaunchKernelGGL((conv_mfma_kernel<KZ, TZ, TY, TX, WM, WN, MB, NB, true, S, KT, (KT == 2 && KZ == 2 ? 32 : CK)>), grid, dim3(kThreads), 0, s, a);
  else if (vec) hipLaunchKernelGGL((conv_mfma_kernel<KZ, TZ, TY, TX, WM, WN, MB, NB, true, S, KT>), grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((conv_mfma_kernel<KZ, TZ, TY, TX, WM, WN, MB, NB, false, S, KT>), grid, dim3(kThreads), 0, s, a);
  return df::launched("df_conv_fwd");
}

template <int KZ, int TZ, int TY, int TX, int S, int KT = 3>
int launch_n(const ConvArgs& a, hipStream_t s) {
  int nt = ntile_for(a.Cout);
  // Small launches (the low-resolution levels at the reference's default batch sizes, config.py:40): a workgroup's four waves run
  // 2 x 2 blocks of 32 x 32 over the whole K = taps x Cin serially -- 61 us of MFMA issue at K = 1152 whatever the grid -- so with fewer
  // workgroups than CUs the N tile is narrowed (same packed operand: Npad is a multiple of every N tile; same summation order per
  // output, bitwise equal) until the grid fills the chip or the tile is 32 wide: 4x the workgroups, a quarter of the chain each.
  const int64_t tiles = static_cast<int64_t>(a.B) * ceil_div(a.D, TZ) * ceil_div(a.H, TY) * ceil_div(a.W, TX) * (a.nclass > 1 ? a.nclass : 1);
  while (nt > 32 && tiles * (a.Npad / nt) * 2 <= df::kCUs) nt >>= 1;
  if (nt == 128) return launch<KZ, TZ, TY, TX, 2, 2, 2, 2, S, KT>(a, s);
  if (nt == 64) return launch<KZ, TZ, TY, TX, 2, 2, 2, 1, S, KT>(a, s);
  return launch<KZ, TZ, TY, TX, 4, 1, 1, 1, S, KT>(a, s);
}

// ---- the lowest-resolution 2-D levels (8x6 ... 16x16 pixels per image): small blocks, short chains ---------------------------------------------
// conv_mfma_kernel gives a wave a 32 x 32 block and the WHOLE K = 9 Cin as one dependent MFMA chain: 576 x 64 cycles = 15 us at Cin = 128 before
// any staging, on 8 - 32 workgroups -- 28-30 us per launch, 16 such launches on the critical path of a 2-D step at the reference's default batch
// (config.py:40; profiles/r06_probes.md section 1).  Here a wave owns 16 pixels x 16 couts on v_mfma_f32_16x16x4_f32: a quarter of the chain
// (288 x 32 cycles), four times the waves, no LDS and no barrier -- both operands are 16-byte loads straight from L1 / L2 (a level's activations and
// the filter bank are a few hundred KB): lane (row | col r, k-quad kq) loads x[pixel r + tap][16 g + 4 kq .. + 3] and the SAME packed operand as
// conv_mfma_kernel, Wp[tap][2 g + (kq >> 1)][kq & 1][n0 + r][0..3], and feeds element j of both to MFMA j of the group (the K order inside a
// 16-channel group is a fixed permutation).  SAME padding through the buffer range check.  Chosen by the image size alone (not the batch): results do
// not depend on the batch size.  Epilogues as conv_mfma_kernel (bias, lrelu, residual, lrelu mask).
__device__ __forceinline__ f32x4 tiny_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__global__ __launch_bounds__(kThreads) void conv_tiny2d_kernel(const ConvArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int n0 = (blockIdx.y * 4 + wave) * 16;
  if (n0 >= a.Cout) return;                                   // wave-uniform
  const int npix = a.B * a.H * a.W;
  const int p = blockIdx.x * 16 + r;                          // this lane's A row
  const bool pok = p < npix;
  const int px = p % a.W, py = (p / a.W) % a.H, pb = p / (a.W * a.H);
  const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, static_cast<unsigned>(npix) * a.Cin * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(a.wp), 0, 9u * a.Kpad * a.Npad * 4u, 0x00020000);
  const int ng = a.Cin >> 4;                                  // 16-channel groups
  const unsigned wlane = static_cast<unsigned>(((kq >> 1) * 2 + (kq & 1)) * a.Npad + n0 + r) * 16u;      // (k8 = 2 g + (kq >> 1), half = kq & 1, n): 16 bytes each
  const unsigned wgroup = static_cast<unsigned>(a.Npad) * 64u;                                           // two k8 = four (k8, half) rows per group
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
    const bool ok = pok && static_cast<unsigned>(yy) < static_cast<unsigned>(a.H) && static_cast<unsigned>(xx) < static_cast<unsigned>(a.W);
    const unsigned xo = ok ? static_cast<unsigned>(((pb * a.H + yy) * a.W + xx) * a.Cin + 4 * kq) * 4u : 0x80000000u;
    const unsigned wt = static_cast<unsigned>(tap) * static_cast<unsigned>(a.Kpad / 8) * 2u * a.Npad * 16u;
    for (int g = 0; g < ng; g += 2) {      // (Cin % 32 == 0) two groups per trip: four loads in flight in front of eight MFMAs
      const f32x4 av0 = tiny_load16(xs, xo, static_cast<unsigned>(g) * 64u), av1 = tiny_load16(xs, xo, static_cast<unsigned>(g + 1) * 64u);
      const f32x4 bv0 = tiny_load16(ws, wlane, wt + static_cast<unsigned>(g) * wgroup), bv1 = tiny_load16(ws, wlane, wt + static_cast<unsigned>(g + 1) * wgroup);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bv0[j], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bv1[j], acc, 0, 0, 0);
    }
  }
  // D layout: acc[e] = (pixel 4 kq + e of the tile, cout n0 + r)
  const int col = n0 + r;
  const float bias = (a.flags & DF_CONV_BIAS) ? a.bias[col] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = blockIdx.x * 16 + 4 * kq + e;
    if (q < npix) {
      const int64_t o = static_cast<int64_t>(q) * a.Cout + col;
      float v = acc[e] + bias;
      if (a.flags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
      if (a.flags & DF_CONV_RESIDUAL) v += a.residual[o];
      if (a.flags & DF_CONV_MASK) v = a.mask_src[o] > 0.f ? v : a.leak * v;
      a.y[o] = v;
    }
  }
}
// the images this kernel takes: stride-1 2-D convs of at most 256 pixels per image (the levels below the Winograd kernels' 16 x 24), channel counts the
// 16-wide blocks tile exactly
inline bool tiny2d_ok(const ConvArgs& a, int kz, int stride) {
  return kz == 1 && stride == 1 && a.nclass <= 1 && static_cast<int64_t>(a.H) * a.W <= 256 && a.Cin % 32 == 0 && a.Cout % 16 == 0 && a.Cout >= 32 &&
         df::aligned16(a.x) && static_cast<int64_t>(a.B) * a.H * a.W * (a.Cin > a.Cout ? a.Cin : a.Cout) < (1LL << 29);
}
int launch_tiny2d(const ConvArgs& a, hipStream_t s) {
  const int64_t npix = static_cast<int64_t>(a.B) * a.H * a.W;
  dim3 grid((unsigned)ceil_div(npix, 16), (unsigned)ceil_div(a.Cout, 64), 1u);
  hipLaunchKernelGGL(conv_tiny2d_kernel, grid, dim3(kThreads), 0, s, a);
  return df::launched("df_conv_fwd");
}

// One parity class of the stride-2 dgrad with its LIVE taps only (nz x ny x nx in {1, 2}^3; 128-wide N tile, 64 | 32-channel chunks).
template <int KZ, int KTY, int KTX, int TZ, int TY, int TX>
int launch_s2d_class(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  a.nz = (int)ceil_div(a.D, TZ); a.ny = (int)ceil_div(a.H, TY); a.nx = (int)ceil_div(a.W, TX);
  const int64_t nt = static_cast<int64_t>(a.B) * a.nz * a.ny * a.nx;
  DF_REQUIRE(nt < (1LL << 31), DF_ESHAPE, "df_conv_s2_dgrad: too many tiles");
  a.ntiles = (int)nt;
  dim3 grid((unsigned)nt, (unsigned)(a.Npad / 128), 1u);
  constexpr int CKT = KZ * KTY * KTX >= 4 ? 32 : 64;      // few taps: longer chunks (a chunk is taps x CKT / 8 MFMA steps between two barriers)
  hipLaunchKernelGGL((conv_mfma_kernel<KZ, TZ, TY, TX, 2, 2, 2, 2, true, 1, KTY, CKT, KTX, true>), grid, dim3(kThreads), 0, s, a);
  return df::launched("df_conv_s2_dgrad");
}
template <int TZ, int TY, int TX>
int launch_s2d_3d(const ConvArgs& a, int nz, int ny, int nx, hipStream_t s) {
  switch (nz * 4 + ny * 2 + nx) {      // taps per axis in {1, 2}
    case 7: return launch_s2d_class<1, 1, 1, TZ, TY, TX>(a, s);
    case 8: return launch_s2d_class<1, 1, 2, TZ, TY, TX>(a, s);
    case 9: return launch_s2d_class<1, 2, 1, TZ, TY, TX>(a, s);
    case 10: return launch_s2d_class<1, 2, 2, TZ, TY, TX>(a, s);
    case 11: return launch_s2d_class<2, 1, 1, TZ, TY, TX>(a, s);
    case 12: return launch_s2d_class<2, 1, 2, TZ, TY, TX>(a, s);
    case 13: return launch_s2d_class<2, 2, 1, TZ, TY, TX>(a, s);
    default: return launch_s2d_class<2, 2, 2, TZ, TY, TX>(a, s);
  }
}
template <int TY, int TX>
int launch_s2d_2d(const ConvArgs& a, int ny, int nx, hipStream_t s) {
  switch (ny * 2 + nx) {
    case 3: return launch_s2d_class<1, 1, 1, 1, TY, TX>(a, s);
    case 4: return launch_s2d_class<1, 1, 2, 1, TY, TX>(a, s);
    case 5: return launch_s2d_class<1, 2, 1, 1, TY, TX>(a, s);
    default: return launch_s2d_class<1, 2, 2, 1, TY, TX>(a, s);
  }
}

}  // namespace

#ifdef DF_TUNING
namespace dfconv { void set_bf16_dbg(int v); }
#endif

extern "C" {

#ifdef DF_TUNING
void df_debug_set_conv_bf16(int v) { dfconv::set_bf16_dbg(v); }      // timing-only variants of conv_bf16x3_kernel (conv_bf16.hip)
#endif

int64_t df_conv_packed_elems(int64_t taps, int64_t cin, int64_t cout, int mode) {
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  return taps * round_up(K, CK) * round_up(N, ntile_for(N));
}

int df_conv_pack_weights(const float* w, float* wp, int64_t taps, int64_t cin, int64_t cout, int mode,
                         df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_conv_pack_weights: null pointer");
  DF_REQUIRE((taps == 9 || taps == 27) && cin > 0 && cout > 0 && (mode == 0 || mode == 1), DF_EINVAL,
             "df_conv_pack_weights: taps must be 9 or 27, mode 0|1");
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int Kpad = (int)round_up(K, CK), Npad = (int)round_up(N, ntile_for(N));
  const int64_t total = taps * Kpad * Npad;
  int64_t g = ceil_div(total, kThreads);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)g), dim3(kThreads), 0, df::as_stream(stream), w, wp, (int)taps, (int)cin,
                     (int)cout, Kpad, Npad, mode);
  return df::launched("df_conv_pack_weights");
}

static int conv_common(const char* fn, const float* x, const float* wp, const float* bias, const float* residual,
                       const float* mask_src, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin,
                       int64_t Cout, int kz, int stride, int flags, float leak, df_stream_t stream, int prec = 0) {
  DF_REQUIRE(x && wp && y, DF_EINVAL, "%s: null pointer", fn);
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "%s: non-positive extent", fn);
  DF_REQUIRE(kz == 1 || kz == 3, DF_ESHAPE, "%s: kz must be 1 (2-D) or 3 (3-D)", fn);
  DF_REQUIRE(kz == 3 || D == 1, DF_ESHAPE, "%s: D must be 1 when kz == 1", fn);
  DF_REQUIRE(!(flags & DF_CONV_BIAS) || bias, DF_EINVAL, "%s: DF_CONV_BIAS without bias", fn);
  DF_REQUIRE(!(flags & DF_CONV_RESIDUAL) || residual, DF_EINVAL, "%s: DF_CONV_RESIDUAL without residual", fn);
  DF_REQUIRE(!(flags & DF_CONV_MASK) || mask_src, DF_EINVAL, "%s: DF_CONV_MASK without mask_src", fn);
  DF_REQUIRE(df::aligned16(wp), DF_EALIGN, "%s: packed weights must be 16-byte aligned", fn);
  DF_REQUIRE(B * D * H * W * stride * stride * stride * (Cin > Cout ? Cin : Cout) < (1LL << 40), DF_ESHAPE,
             "%s: tensor too large", fn);
  ConvArgs a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.bias = bias; a.residual = residual; a.mask_src = mask_src;
  a.y = y;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.Di = (int)(kz == 3 ? D * stride : D); a.Hi = (int)(H * stride); a.Wi = (int)(W * stride);
  a.pz = stride == 1 ? kz / 2 : 0; a.py = a.px = stride == 1 ? 1 : 0;
  a.is = 1; a.iz = a.iy = a.ix = 0; a.xD = a.Di; a.xH = a.Hi; a.xW = a.Wi;
  a.os = 1; a.oz = a.oy = a.ox = 0; a.yD = a.D; a.yH = a.H; a.yW = a.W;
  a.nclass = 1; a.wclass = 0;
  a.Kpad = (int)round_up(Cin, CK); a.Npad = (int)round_up(Cout, ntile_for(Cout));
  a.flags = flags; a.leak = leak;
  a.nz = a.ny = a.nx = a.ntiles = 0;
  hipStream_t s = df::as_stream(stream);
  if (prec == 1) {
    DF_REQUIRE(stride == 1 && bf16x3_supported(a), DF_ESHAPE, "%s: bf16x3 needs stride 1, Cin %% 4 == 0, channels >= 16", fn);
    a.Kpad = (int)bf16x3_kpad(Cin);
    return launch_bf16x3(a, kz, 3, s);
  }
  if (stride == 2) {
    if (kz == 3) {
      if (W >= 12) return launch_n<3, 2, 4, 16, 2>(a, s);
      return launch_n<3, 4, 4, 8, 2>(a, s);
    }
    if (W >= 12) return launch_n<1, 1, 8, 16, 2>(a, s);
    return launch_n<1, 1, 16, 8, 2>(a, s);
  }
  if (Cout <= 4) return launch_small_n(a, kz, s);                 // thin output: vector-ALU kernel
  if (Cin <= 4 && Cout >= 32) return launch_small_k(a, kz, s);    // thin input (dgrad of the last layer; 3 -> F)
  if (tiny2d_ok(a, kz, stride)) return launch_tiny2d(a, s);      // the lowest 2-D levels: 16 x 16 blocks, short chains
  if (kz == 3) {
    if (W >= 12) return launch_n<3, 2, 4, 16, 1>(a, s);
    return launch_n<3, 4, 4, 8, 1>(a, s);
  }
  if (W >= 12) return launch_n<1, 1, 8, 16, 1>(a, s);
  return launch_n<1, 1, 16, 8, 1>(a, s);
}

int df_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int flags,
                float leak, df_stream_t stream) {
  return conv_common("df_conv_fwd", x, wp, bias, residual, mask_src, y, B, D, H, W, Cin, Cout, kz, 1, flags, leak,
                     stream);
}

int df_conv_s2_fwd(const float* x, const float* wp, const float* bias, float* y, int64_t B, int64_t Do, int64_t Ho,
                   int64_t Wo, int64_t Cin, int64_t Cout, int kz, int flags, float leak, df_stream_t stream) {
  DF_REQUIRE(!(flags & (DF_CONV_RESIDUAL | DF_CONV_MASK)), DF_EINVAL, "df_conv_s2_fwd: only BIAS / LRELU epilogues");
  return conv_common("df_conv_s2_fwd", x, wp, bias, nullptr, nullptr, y, B, Do, Ho, Wo, Cin, Cout, kz, 2, flags, leak,
                     stream);
}

int64_t df_upconv_packed_elems(int64_t cin, int64_t cout, int kz, int mode) {
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int64_t nct = kz == 3 ? 64 : 16;    // classes * taps
  return nct * round_up(K, CK) * round_up(N, ntile_for(N));
}

int df_upconv_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int kz, int mode, df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_upconv_pack_weights: null pointer");
  DF_REQUIRE((kz == 1 || kz == 3) && cin > 0 && cout > 0 && mode >= 0 && mode <= 2, DF_EINVAL,
             "df_upconv_pack_weights: kz must be 1|3, mode 0|1|2");
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int Kpad = (int)round_up(K, CK), Npad = (int)round_up(N, ntile_for(N));
  const int64_t total = df_upconv_packed_elems(cin, cout, kz, mode);
  int64_t g = ceil_div(total, kThreads);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(upconv_pack_kernel, dim3((unsigned)g), dim3(kThreads), 0, df::as_stream(stream), w, wp, kz, (int)cin,
                     (int)cout, Kpad, Npad, mode);
  return df::launched("df_upconv_pack_weights");
}

static int upconv_launch(const ConvArgs& a, int kz, hipStream_t s) {
  if (kz == 3) {
    if (a.W >= 12) return launch_n<2, 2, 4, 16, 1, 2>(a, s);
    return launch_n<2, 4, 4, 8, 1, 2>(a, s);
  }
  if (a.W >= 12) return launch_n<1, 1, 8, 16, 1, 2>(a, s);
  return launch_n<1, 1, 16, 8, 1, 2>(a, s);
}

static int upconv_fwd_impl(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Dc, int64_t Hc,
                           int64_t Wc, int64_t Cin, int64_t Cout, int kz, int flags, float leak, df_stream_t stream, int prec) {
  DF_REQUIRE(xc && wp && y, DF_EINVAL, "df_upconv_fwd: null pointer");
  DF_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "df_upconv_fwd: non-positive extent");
  DF_REQUIRE(kz == 1 || kz == 3, DF_ESHAPE, "df_upconv_fwd: kz must be 1 (2-D) or 3 (3-D)");
  DF_REQUIRE(kz == 3 || Dc == 1, DF_ESHAPE, "df_upconv_fwd: Dc must be 1 when kz == 1");
  DF_REQUIRE(Cin > 4 && Cout > 4, DF_ESHAPE, "df_upconv_fwd: MFMA path only (channels > 4)");
  DF_REQUIRE(!(flags & (DF_CONV_RESIDUAL | DF_CONV_MASK)), DF_EINVAL, "df_upconv_fwd: only BIAS / LRELU epilogues");
  DF_REQUIRE(!(flags & DF_CONV_BIAS) || bias, DF_EINVAL, "df_upconv_fwd: DF_CONV_BIAS without bias");
  DF_REQUIRE(df::aligned16(wp), DF_EALIGN, "df_upconv_fwd: packed weights must be 16-byte aligned");
  ConvArgs a;
  a.x = xc; a.wp = reinterpret_cast<const f32x4*>(wp); a.bias = bias; a.residual = nullptr; a.mask_src = nullptr; a.y = y;
  a.B = (int)B; a.D = (int)Dc; a.H = (int)Hc; a.W = (int)Wc; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.Di = a.D; a.Hi = a.H; a.Wi = a.W;
  a.Kpad = (int)round_up(Cin, CK); a.Npad = (int)round_up(Cout, ntile_for(Cout));
  a.flags = flags; a.leak = leak;
  a.nz = a.ny = a.nx = a.ntiles = 0;
  a.pz = a.py = a.px = 0;                        // set per class in the kernel
  a.is = 1; a.iz = a.iy = a.ix = 0; a.xD = a.D; a.xH = a.H; a.xW = a.W;
  a.os = 2; a.oz = a.oy = a.ox = 0;              // scatter to the fine grid; offsets per class in the kernel
  a.yD = kz == 3 ? 2 * a.D : 1; a.yH = 2 * a.H; a.yW = 2 * a.W;
  if (kz == 1) a.os = 2;                         // (z is untouched in 2-D: D == 1, oz == 0 -> 0*2+0)
  a.nclass = kz == 3 ? 8 : 4;
  if (prec == 1) {
    DF_REQUIRE(bf16x3_supported(a), DF_ESHAPE, "df_upconv_fwd_bf16x3: Cin %% 4 == 0, channels >= 16");
    a.Kpad = (int)bf16x3_kpad(Cin);
  }
  a.wclass = static_cast<int64_t>(kz == 3 ? 8 : 4) * a.Kpad * a.Npad / 4;
  if (prec == 1) return launch_bf16x3(a, kz, 2, df::as_stream(stream));
  return upconv_launch(a, kz, df::as_stream(stream));
}

int df_upconv_fwd(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Dc, int64_t Hc,
                  int64_t Wc, int64_t Cin, int64_t Cout, int kz, int flags, float leak, df_stream_t stream) {
  return upconv_fwd_impl(xc, wp, bias, y, B, Dc, Hc, Wc, Cin, Cout, kz, flags, leak, stream, 0);
}

// Input gradient of the stride-2 conv (df_conv_s2_fwd; reference: the encoder's / discriminator's down-sampling layers, model.py:141-143,
// 177-179, 94-99): gx[B, 2Do, 2Ho, 2Wo, Cin] from gy[B, Do, Ho, Wo, Cout] and the mode-2 packed weights of df_upconv_pack_weights.
// Per axis dx[2m] = g[m-1] w[2] + g[m] w[0] (two taps), dx[2m+1] = g[m] w[1] (ONE tap): the 8 (4) parity classes have 8, 4, 4, 2, 4, 2, 2, 1
// live taps -- 27 of the 64 the generic parity-class kernel (df_upconv_fwd on the same operand, the fallback below) multiplies.
// 1: the live-tap parity-class kernels (27 | 9 tap-products); 0: the generic 2x2(x2)-tap parity-class kernel on the zero-padded taps
static bool s2_dgrad_special(const float* gy, int64_t Cin, int64_t Cout) {
  return ntile_for(Cin) == 128 && round_up(Cout, CK) % 64 == 0 && Cout % 4 == 0 && df::aligned16(gy);
}

int df_conv_s2_dgrad_form(const float* gy, int64_t Cin, int64_t Cout) { return s2_dgrad_special(gy, Cin, Cout) ? 1 : 0; }

int df_conv_s2_dgrad(const float* gy, const float* wp, float* gx, int64_t B, int64_t Do, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout,
                     int kz, df_stream_t stream) {
  DF_REQUIRE(gy && wp && gx, DF_EINVAL, "df_conv_s2_dgrad: null pointer");
  DF_REQUIRE(B > 0 && Do > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "df_conv_s2_dgrad: non-positive extent");
  DF_REQUIRE(kz == 1 || kz == 3, DF_ESHAPE, "df_conv_s2_dgrad: kz must be 1 (2-D) or 3 (3-D)");
  DF_REQUIRE(kz == 3 || Do == 1, DF_ESHAPE, "df_conv_s2_dgrad: Do must be 1 when kz == 1");
  DF_REQUIRE(Cin > 4 && Cout > 4, DF_ESHAPE, "df_conv_s2_dgrad: MFMA path only (channels > 4)");
  DF_REQUIRE(df::aligned16(wp), DF_EALIGN, "df_conv_s2_dgrad: packed weights must be 16-byte aligned");
  const int64_t Kpad = round_up(Cout, CK), Npad = round_up(Cin, ntile_for(Cin));      // K = the forward conv's Cout, N = its Cin
  const bool special = s2_dgrad_special(gy, Cin, Cout);
  if (!special)      // other channel counts: the generic 2x2(x2)-tap parity-class kernel on the zero-padded taps
    return upconv_fwd_impl(gy, wp, nullptr, gx, B, Do, Ho, Wo, Cout, Cin, kz, 0, 0.f, stream, 0);
  hipStream_t s = df::as_stream(stream);
  const int ncls = kz == 3 ? 8 : 4;
  for (int c = 0; c < ncls; ++c) {
    const int bz = kz == 3 ? (c >> 2) & 1 : 0, by = (c >> 1) & 1, bx = c & 1;      // output parity per axis: 0 -> two taps, 1 -> one
    ConvArgs a;
    a.x = gy; a.bias = nullptr; a.residual = nullptr; a.mask_src = nullptr; a.y = gx;
    a.B = (int)B; a.D = (int)Do; a.H = (int)Ho; a.W = (int)Wo; a.Cin = (int)Cout; a.Cout = (int)Cin;
    a.Di = a.D; a.Hi = a.H; a.Wi = a.W;
    a.Kpad = (int)Kpad; a.Npad = (int)Npad;
    a.wp = reinterpret_cast<const f32x4*>(wp) + static_cast<int64_t>(c) * (kz == 3 ? 8 : 4) * Kpad * Npad / 4;
    a.flags = 0; a.leak = 0.f;
    a.nz = a.ny = a.nx = a.ntiles = 0;
    a.pz = kz == 3 ? 1 - bz : 0; a.py = 1 - by; a.px = 1 - bx;      // as conv_mfma_kernel sets them per class of a one-launch parity conv
    a.is = 1; a.iz = a.iy = a.ix = 0; a.xD = a.D; a.xH = a.H; a.xW = a.W;
    a.os = 2; a.oz = bz; a.oy = by; a.ox = bx;
    a.yD = kz == 3 ? 2 * a.D : 1; a.yH = 2 * a.H; a.yW = 2 * a.W;
    a.nclass = 1; a.wclass = 0;
    int e;
    if (kz == 3) e = a.W >= 12 ? launch_s2d_3d<2, 4, 16>(a, 2 - bz, 2 - by, 2 - bx, s) : launch_s2d_3d<4, 4, 8>(a, 2 - bz, 2 - by, 2 - bx, s);
    else e = a.W >= 12 ? launch_s2d_2d<8, 16>(a, 2 - by, 2 - bx, s) : launch_s2d_2d<16, 8>(a, 2 - by, 2 - bx, s);
    if (e) return e;
  }
  return DF_OK;
}
int df_upconv_fwd_bf16x3(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Dc, int64_t Hc,
                         int64_t Wc, int64_t Cin, int64_t Cout, int kz, int flags, float leak, df_stream_t stream) {
  return upconv_fwd_impl(xc, wp, bias, y, B, Dc, Hc, Wc, Cin, Cout, kz, flags, leak, stream, 1);
}

static int upconv_dgrad_impl(const float* g, const float* wp, float* acc, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                             int64_t Cin, int64_t Cout, int kz, df_stream_t stream, int prec) {
  DF_REQUIRE(g && wp && acc, DF_EINVAL, "df_upconv_dgrad: null pointer");
  DF_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && Cin > 0 && Cout > 0, DF_EINVAL, "df_upconv_dgrad: non-positive extent");
  DF_REQUIRE(kz == 1 || kz == 3, DF_ESHAPE, "df_upconv_dgrad: kz must be 1 (2-D) or 3 (3-D)");
  DF_REQUIRE(kz == 3 || Dc == 1, DF_ESHAPE, "df_upconv_dgrad: Dc must be 1 when kz == 1");
  DF_REQUIRE(Cin > 4 && Cout > 4, DF_ESHAPE, "df_upconv_dgrad: MFMA path only (channels > 4)");
  DF_REQUIRE(df::aligned16(wp), DF_EALIGN, "df_upconv_dgrad: packed weights must be 16-byte aligned");
  // acc[n] += sum_p sum_t G_p[n + t - p] * Wd_p[t],  G_p[m] = g[2m + p]: one launch per parity class, accumulated in
  // place through the residual epilogue (stream order makes the read-modify-write safe)
  hipStream_t s = df::as_stream(stream);
  const int ncls = kz == 3 ? 8 : 4;
  for (int c = 0; c < ncls; ++c) {
    const int bz = kz == 3 ? (c >> 2) & 1 : 0, by = (c >> 1) & 1, bx = c & 1;
    ConvArgs a;
    a.x = g; a.bias = nullptr; a.residual = acc; a.mask_src = nullptr; a.y = acc;
    a.B = (int)B; a.D = (int)Dc; a.H = (int)Hc; a.W = (int)Wc; a.Cin = (int)Cout; a.Cout = (int)Cin;   // K = fwd Cout, N = fwd Cin
    a.Di = a.D; a.Hi = a.H; a.Wi = a.W;
    a.Kpad = (int)(prec == 1 ? bf16x3_kpad(Cout) : round_up(Cout, CK)); a.Npad = (int)round_up(Cin, ntile_for(Cin));
    a.wp = reinterpret_cast<const f32x4*>(wp) + static_cast<int64_t>(c) * (kz == 3 ? 8 : 4) * a.Kpad * a.Npad / 4;
    a.flags = DF_CONV_RESIDUAL; a.leak = 0.f;
    a.nz = a.ny = a.nx = a.ntiles = 0;
    a.pz = bz; a.py = by; a.px = bx;                               // offsets {0,+1} for p = 0, {-1,0} for p = 1
    a.is = 2; a.iz = bz; a.iy = by; a.ix = bx;                      // gather the class-p sub-grid of the fine gradient
    a.xD = kz == 3 ? 2 * a.D : 1; a.xH = 2 * a.H; a.xW = 2 * a.W;
    if (kz == 1) { a.iz = 0; }
    a.os = 1; a.oz = a.oy = a.ox = 0; a.yD = a.D; a.yH = a.H; a.yW = a.W;
    a.nclass = 1; a.wclass = 0;
    if (prec == 1) {
      DF_REQUIRE(bf16x3_supported(a), DF_ESHAPE, "df_upconv_dgrad_bf16x3: channels %% 4 == 0, >= 16");
      if (int e = launch_bf16x3(a, kz, 2, s)) return e;
    } else if (int e = upconv_launch(a, kz, s)) {
      return e;
    }
  }
  return DF_OK;
}

int df_upconv_dgrad(const float* g, const float* wp, float* acc, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                    int64_t Cin, int64_t Cout, int kz, df_stream_t stream) {
  return upconv_dgrad_impl(g, wp, acc, B, Dc, Hc, Wc, Cin, Cout, kz, stream, 0);
}
int df_upconv_dgrad_bf16x3(const float* g, const float* wp, float* acc, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc,
                           int64_t Cin, int64_t Cout, int kz, df_stream_t stream) {
  return upconv_dgrad_impl(g, wp, acc, B, Dc, Hc, Wc, Cin, Cout, kz, stream, 1);
}

/* ---- bf16x3 split-precision entry points (conv_bf16.hip): same arguments as their fp32 twins ------------------------------ */
int64_t df_conv_packed_elems_bf16x3(int64_t taps, int64_t cin, int64_t cout, int mode) {
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  return taps * bf16x3_kpad(K) * round_up(N, ntile_for(N));       // 4-byte units (a hi and a lo bf16 per element)
}
int df_conv_pack_weights_bf16x3(const float* w, float* wp, int64_t taps, int64_t cin, int64_t cout, int mode,
                                df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_conv_pack_weights_bf16x3: null pointer");
  DF_REQUIRE((taps == 9 || taps == 27) && cin > 0 && cout > 0 && (mode == 0 || mode == 1), DF_EINVAL,
             "df_conv_pack_weights_bf16x3: taps must be 9 or 27, mode 0|1");
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  return pack_bf16x3(w, wp, (int)taps, (int)cin, (int)cout, (int)bf16x3_kpad(K), (int)round_up(N, ntile_for(N)), mode,
                     df::as_stream(stream));
}
int df_conv_fwd_bf16x3(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                       float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kz, int flags,
                       float leak, df_stream_t stream) {
  return conv_common("df_conv_fwd_bf16x3", x, wp, bias, residual, mask_src, y, B, D, H, W, Cin, Cout, kz, 1, flags, leak,
                     stream, 1);
}
int64_t df_upconv_packed_elems_bf16x3(int64_t cin, int64_t cout, int kz, int mode) {
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  return (kz == 3 ? 64 : 16) * bf16x3_kpad(K) * round_up(N, ntile_for(N));
}
int df_upconv_pack_weights_bf16x3(const float* w, float* wp, int64_t cin, int64_t cout, int kz, int mode,
                                  df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_upconv_pack_weights_bf16x3: null pointer");
  DF_REQUIRE((kz == 1 || kz == 3) && cin > 0 && cout > 0 && (mode == 0 || mode == 1), DF_EINVAL,
             "df_upconv_pack_weights_bf16x3: kz must be 1|3, mode 0|1");
  const int64_t K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  return upconv_pack_bf16x3(w, wp, kz, (int)cin, (int)cout, (int)bf16x3_kpad(K), (int)round_up(N, ntile_for(N)), mode,
                            df::as_stream(stream));
}

}  // extern "C"
