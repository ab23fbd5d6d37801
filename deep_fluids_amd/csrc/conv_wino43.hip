// 3x3x3 SAME stride-1 convolution as Winograd F(2,3) x F(2,3) x F(4,3) (z, y, x) on the fp32 matrix cores -- the round-6 kernel family
// for the 128 -> 128-class layers of GeneratorBE3 (reference: slim.conv3d behind ops.py:15-16, called from model.py:66-70).
//
//   2 x 2 x 4 output tile from a 4 x 4 x 6 input tile: 96 transform points per 16 outputs = 6 MFMA MACs per output voxel and (cin, cout)
//   pair, where F(2,3)^3 (conv_wino.hip) needs 8 and the direct form 27.  All arithmetic fp32; the F(4,3) transform constants (4, 5, 2, 8;
//   1/4, 1/6, 1/12, 1/24 in the fp64-transformed weights) cost about one bit: relative L1 error of one layer 9e-7 against 4e-7 of
//   F(2,3)^3 and 9e-7 of the direct fp32 sum (profiles/r06_probes.md, section 3).
//
// Around the 96 GEMMs the decomposition is conv_wino.hip's, kept on purpose so that the two families are interchangeable:
//   * persistent 8-wave workgroup per CU, XCD-pinned (cout slice, tile block) items; tile block = 4 x 8 x 8 output voxels x 32 couts;
//   * double-buffered LDS staging of a 16-channel chunk of the 6 x 10 x 10 halo block, one LDS-only barrier per chunk, SAME padding by the
//     buffer range check, the next block's first chunk staged during the last chunk;
//   * after the combine a lane holds the SAME 2 x 2 x 2 output cube of one cout as in conv_wino.hip (wave = (z pair, x pair), lane =
//     (y pair, cout)), so the fused epilogues -- bias, lrelu, residual, lrelu mask, skip add-up, sign words out / mask from sign words --
//     and the sign-word layout (ops.sign_bits_to_mask, df_lrelu_bits_bwd_pool2x) are shared bit for bit.
// What differs:
//   * a tile block is 16 tiles (tz 2 x ty 4 x tx 2) = ONE MFMA row block; wave = (xi_z, xi_y pair) owns 2 x 6 points for all 16 tiles and
//     both 16-cout blocks: 24 MFMA 16x16x4 per k-step (32 before), 96 accumulators (128);
//   * the z part of B^T moves to the STAGING threads: thread = (halo row, column, channel quad) loads its z column (6 float4; a plane outside
//     the tensor through a zero-length descriptor) and writes the 8 planes (tile z-row, xi_z) -- the A path then reads ONE plane: 3 rows x
//     6 columns = 9 ds_read_b64 per k-step (conv_wino.hip: 16), no z stage, 18 fewer registers, one staging offset per thread instead of
//     ten; y stage 6 packed ops, x stage = B^T of F(4,3) in 6 packed ops per row (op_sel picks the halves);
//   * B operand: U packed [cs][xi_z][xi_y pair][cin/4][cout block][cin%4][cout%16][12 points]: three 16-byte loads per lane and block;
//   * inverse: x (A^T of F(4,3), 6 -> 4) in registers, the two xi_y halves through LDS (one extra exchange), then the xi_z combine; full blocks store
//     through a 4 x 4 lane-quad transpose (DPP): 2 float4 stores per lane and cout block instead of 8 scalar ones, the sign-word layout untouched.
// Measured (profiles/r06_probes.md section 3): 128 -> 128 at 64x96x64, B = 16: 13.3 ms per launch against 16.4 ms (F(2,3)^3), bit-compatible
// epilogues; 0 spills at 204-217 VGPRs, 156,032 B of LDS.
#include <type_traits>
#include "df_common.hpp"
#include "conv_args.hpp"

namespace {

using df::ceil_div;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;                   // 8 waves
constexpr int CKW = 16;                   // input channels per LDS chunk (4 k-steps)
// LDS holds the z-TRANSFORMED halo block of a chunk: per channel 8 planes (tile z-row tz, xi_z) of 10 rows x 10 columns -- the staging
// threads own a z column of the 6 x 10 x 10 halo block and apply the z part of B^T once per chunk, so the A path reads ONE plane
// (9 ds_read_b64 per k-step, not 18) and has no z stage.  Pitches: row 12, plane 120, channel 8 * 120 + 2 = 962 dwords: a wave's
// ds_read_b64 (lane = (tz, ty, txh, cin)) touch 32 distinct bank pairs per half (2 * 120 = 16 mod 32 separates tz, 962 / 2 odd separates cin).
constexpr int PY = 12, PP = 120, NPL = 8;
constexpr int CP = NPL * PP + 2;          // dwords per channel (962)
constexpr int NCOL = 100 * 4;             // staging threads: (hy, hx) x channel quad, 6 loads (z = 0..5) each
constexpr int BUFF = CKW * CP;            // floats per LDS buffer (61,568 B)
constexpr int kSignBits = 64, kMaskBits = 128, kNoPrimary = 256;      // internal epilogue flags, as conv_wino.hip
constexpr int kBitBytesPerBlock = 1024;
constexpr int kPts = 96;                  // 4 x 4 x 6 transform points

struct W43Args {
  const float* x;
  const float* wp;
  const float* bias;
  const float* residual;
  const float* mask_src;
  float* y;
  float* y2;
  unsigned char* bits_out;
  const unsigned char* bits_in;
  int B, D, H, W, Cin, Cout;
  int nbz, nby, nbx, ntb, ncs;
  int flags;
  float leak;
  int spx;
};

// ---- weight transform + packing ------------------------------------------------------------------------------------------------
// mode 0: g[tap][k][n] = w[tap][k][n];  mode 1: g[tap][k][n] = w[26 - tap][n][k]  (dgrad operand)
// Up[cs][xz][yh][k4][nb][kq][j][(xy & 1) * 6 + xx] = sum_taps G2[xz][tz] G2[xy][ty] G4[xx][tx] g[tap][4 k4 + kq][32 cs + 16 nb + j],  xy = 2 yh + (xy & 1)
__global__ __launch_bounds__(64) void wino43_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout, int mode) {
  const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int64_t nfil = static_cast<int64_t>(K) * N;
  for (int64_t f = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; f < nfil; f += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(f % N), k = static_cast<int>(f / N);
    double g[27];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
      g[tap] = static_cast<double>(mode == 0 ? w[(static_cast<int64_t>(tap) * cin + k) * cout + n]
                                             : w[(static_cast<int64_t>(26 - tap) * cin + n) * cout + k]);
    double gx[9][6];      // x: [tz][ty][3] -> [tz][ty][6]  (G of F(4,3))
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const double a = g[r * 3], b = g[r * 3 + 1], c = g[r * 3 + 2];
      gx[r][0] = a / 4.0;
      gx[r][1] = -(a + b + c) / 6.0;
      gx[r][2] = -(a - b + c) / 6.0;
      gx[r][3] = a / 24.0 + b / 12.0 + c / 6.0;
      gx[r][4] = a / 24.0 - b / 12.0 + c / 6.0;
      gx[r][5] = c;
    }
    const int cs = n >> 5, nb = (n >> 4) & 1, j = n & 15, k4 = k >> 2, kq = k & 3;
#pragma unroll
    for (int xz = 0; xz < 4; ++xz) {
      double gz[3][6];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int xx = 0; xx < 6; ++xx) {
          const double a = gx[ty][xx], b = gx[3 + ty][xx], c = gx[6 + ty][xx];
          gz[ty][xx] = xz == 0 ? a : xz == 3 ? c : xz == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c);
        }
#pragma unroll
      for (int yh = 0; yh < 2; ++yh) {      // a lane record = the 12 points of an xi_y pair: three 16-byte stores (24 scattered 4-byte ones before)
        float o[12];
#pragma unroll
        for (int xyl = 0; xyl < 2; ++xyl) {
          const int xy = 2 * yh + xyl;
#pragma unroll
          for (int xx = 0; xx < 6; ++xx) {
            const double a = gz[0][xx], b = gz[1][xx], c = gz[2][xx];
            o[xyl * 6 + xx] = static_cast<float>(xy == 0 ? a : xy == 3 ? c : xy == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c));
          }
        }
        const int64_t idx = (((((((static_cast<int64_t>(cs) * 4 + xz) * 2 + yh) * (K / 4) + k4) * 2 + nb) * 4 + kq) * 16 + j) * 12);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4*>(wp + idx + 4 * q) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
      }
    }
  }
}

struct BlockInfo {
  const float* xb;
  int hoff;
  int b, z0, y0, x0;
  int id;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// FL >= 0: the epilogue flags are the compile-time constant FL; FL < 0: run-time a.flags (public DF_CONV_* bits only)
// DBG (probe build -DDF_W43_PROBE only, results wrong by construction): 1 no input transform, 2 no LDS operand reads, 4 no staging,
// 8 no weight reloads, 64 staging loads kept but nothing written to LDS, 128 staged zeros (z transform + LDS writes, no loads)
// -- tools/r06_wino43_probe.py's time breakdown
template <int FL, int DBG = 0>
__global__ __launch_bounds__(kT, 1) void wino43_kernel(const W43Args a) {
#ifndef DF_W43_PROBE
  static_assert(DBG == 0, "wino43_kernel: probe variants exist in the -DDF_W43_PROBE build only");
#endif
  __shared__ __attribute__((aligned(16))) float sIn[2 * BUFF];
  __shared__ __attribute__((aligned(16))) float sXc[8192];      // second exchange area (32 KB): the (y, x)-complete partials per xi_z
  __shared__ float sBias[32];

  const int eflags = FL >= 0 ? FL : a.flags;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 15, kq = lane >> 4;
  // main-loop role: (xi_z, xi_y pair);  combine / store role: (z pair th, x pair xz) -- the latter exactly conv_wino.hip's
  const int mz = wave & 3, yh = wave >> 2;
  const int xz = wave & 3, th = wave >> 2;

  int cs, tb, tstride;
  {
    const int g = blockIdx.x, G = gridDim.x;
    if ((8 % a.ncs) == 0 && (G & 7) == 0) {
      const int spx = a.spx, xpg = a.ncs / spx;
      const int xcd = g & 7, slot = g >> 3, wx = G >> 3;
      const int ngroups = 8 / xpg, tw = wx / spx;
      cs = (xcd % xpg) * spx + slot % spx;
      tb = (xcd / xpg) * tw + slot / spx;
      tstride = ngroups * tw;
      if (slot / spx >= tw) return;
    } else {
      const int nw = G / a.ncs;
      cs = g % a.ncs;
      tb = g / a.ncs;
      tstride = nw;
      if (tb >= nw) return;
    }
  }
  if (tb >= a.ntb) return;
  const int n0 = cs * 32;
  if (tid < 32) sBias[tid] = (eflags & DF_CONV_BIAS) ? a.bias[n0 + tid] : 0.f;
  const int tb0 = tb;
  const int niter = (a.ntb - tb0 + tstride - 1) / tstride;

  auto decode = [&](int t) -> BlockInfo {
    BlockInfo bi;
    bi.id = t;
    const int bx = t % a.nbx;
    int t2 = t / a.nbx;
    const int by = t2 % a.nby; t2 /= a.nby;
    const int bz = t2 % a.nbz;
    bi.b = t2 / a.nbz;
    bi.z0 = bz * 4; bi.y0 = by * 8; bi.x0 = bx * 8;
    bi.xb = a.x + static_cast<int64_t>(bi.b) * a.D * a.H * a.W * a.Cin;
    bi.hoff = (((bi.z0 - 1) * a.H + (bi.y0 - 1)) * a.W + (bi.x0 - 1)) * a.Cin;
    return bi;
  };

  // ---- staging plan: thread t < 400 = (halo row hy, column hx, channel quad q4) owns the z column: 6 float4 loads, 8 planes x 4 channels written --
  const int scol = tid < NCOL ? tid : NCOL - 1;      // (threads 400..511 stage nothing; they mirror the last column's addresses)
  const int sq4 = scol & 3, shy = (scol >> 2) / 10, shx = (scol >> 2) % 10;
  const int ldst = ((sq4 * 4) * CP + shy * PY + shx) * 4;      // bytes: plane 0, channel 0 of the quad, buffer 0
  const bool stager = wave < 7;                                  // wave-uniform (wave 6: 16 active lanes)
  const unsigned vol_bytes = static_cast<unsigned>(a.D * a.H * a.W) * a.Cin * 4u;
  unsigned so;       // byte offset of (z = 0, gy, gx, quad) inside the batch volume, or out of range: the SAME padding in y / x
  auto set_offs = [&](const BlockInfo& bi) {
    const int gy = bi.y0 - 1 + shy, gx = bi.x0 - 1 + shx;
    const bool ok = tid < NCOL && static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) && static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
    so = ok ? static_cast<unsigned>((gy * a.W + gx) * a.Cin + sq4 * 4) * 4u : 0x80000000u;
  };
  char* sInB = reinterpret_cast<char*>(sIn);
  // plane z of the column: scalar offset of the plane (+ the chunk); a plane outside the tensor reads through a zero-length descriptor
  auto stage_load = [&](const BlockInfo& bi, unsigned chunkbytes, f32x4 (&v)[6]) {
    const int plane_bytes = a.H * a.W * a.Cin * 4;
#pragma unroll
    for (int z = 0; z < 6; ++z) {
      const int gz = bi.z0 - 1 + z;
      const bool zok = static_cast<unsigned>(gz) < static_cast<unsigned>(a.D);
      const __amdgpu_buffer_rsrc_t srd = make_srd(bi.xb, zok ? vol_bytes : 0u);
      v[z] = buf_load16(srd, so, static_cast<unsigned>(zok ? gz : 0) * static_cast<unsigned>(plane_bytes) + chunkbytes);
    }
  };
  auto stage_store = [&](int bufbytes, const f32x4 (&v)[6]) {      // z part of B^T:  (d0 - d2, d1 + d2, d2 - d1, d1 - d3) per tile z-row
    if (tid < NCOL) {
      char* d = sInB + (ldst + bufbytes);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 p0 = v[2 * t] - v[2 * t + 2], p1 = v[2 * t + 1] + v[2 * t + 2], p2 = v[2 * t + 2] - v[2 * t + 1], p3 = v[2 * t + 1] - v[2 * t + 3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float* dc = reinterpret_cast<float*>(d + c * CP * 4);
          dc[(t * 4 + 0) * PP] = p0[c]; dc[(t * 4 + 1) * PP] = p1[c]; dc[(t * 4 + 2) * PP] = p2[c]; dc[(t * 4 + 3) * PP] = p3[c];
        }
      }
    }
  };

  // ---- A operand: lane = (tile tl = (tz, ty, txh), cin kq); rows yh .. yh + 2 of the tile's 4 rows, all 6 columns, plane (tz, xi_z) ------
  const int tz = tl >> 3, ty = (tl >> 1) & 3, txh = tl & 1;
  const int offAb = (kq * CP + (tz * 4 + mz) * PP + (2 * ty + yh) * PY + 4 * txh) * 4;      // bytes (multiple of 8)
  f32x2 ra[9];             // [row][x pair]
  float Av[12];            // A operands of a k-step: [xi_y local][xi_x]
  auto raw_read = [&](int idxbytes) {
    int ia = idxbytes + offAb;
    asm volatile("" : "+v"(ia));
    __builtin_assume((ia & 7) == 0);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) ra[r * 3 + j] = *reinterpret_cast<const f32x2*>(sInB + ia + (r * PY + 2 * j) * 4);
  };
  // B^T of F(4,3) on a row (d0 d1 | d2 d3 | d4 d5) in SIX packed ops (op_sel picks the halves):
  //   (o0, o5) = 4 (d0, d1) - 5 (d2, d3) + (d4, d5);   (a, c) = d4 + (-4, -1) d2;   (b, e) = d3 + (-4, -1) d1;
  //   (o1, o2) = a +- b;   (o3, o4) = c +- 2 e
  // (the four constant pairs sit in SGPR pairs: VOP3P takes scalar sources, and the kernel has no VGPRs to spare)
  auto xstage = [&](const f32x2 (&u)[3], float* o) {
    const f32x2 kM5 = {-5.f, -5.f}, kP4 = {4.f, 4.f}, kM41 = {-4.f, -1.f}, kP2M2 = {2.f, -2.f};
    f32x2 p, o05, ac, be, o12, o34;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "s"(kM5), "v"(u[1]), "v"(u[2]));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o05) : "s"(kP4), "v"(u[0]), "v"(p));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ac) : "s"(kM41), "v"(u[1]), "v"(u[2]));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(be) : "s"(kM41), "v"(u[0]), "v"(u[1]));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(o12) : "v"(ac), "v"(be));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(o34) : "s"(kP2M2), "v"(be), "v"(ac));
    o[0] = o05[0]; o[5] = o05[1]; o[1] = o12[0]; o[2] = o12[1]; o[3] = o34[0]; o[4] = o34[1];
  };
  auto transform = [&]() {
    f32x2 Ua[3], Ub[3];
    if (yh == 0) {      // rows (0, 1, 2): xi_y 0 = r0 - r2, xi_y 1 = r1 + r2          (wave-uniform branch)
#pragma unroll
      for (int j = 0; j < 3; ++j) { Ua[j] = pk_sub(ra[j], ra[6 + j]); Ub[j] = pk_add(ra[3 + j], ra[6 + j]); }
    } else {            // rows (1, 2, 3): xi_y 2 = r2 - r1, xi_y 3 = r1 - r3
#pragma unroll
      for (int j = 0; j < 3; ++j) { Ua[j] = pk_sub(ra[3 + j], ra[j]); Ub[j] = pk_sub(ra[j], ra[6 + j]); }
    }
    xstage(Ua, Av);
    xstage(Ub, Av + 6);
  };

  // ---- B operand: three float4 per lane and cout block = its 12 points --------------------------------------------------------------------
  const int nk4 = a.Cin >> 2;
#ifndef W43_NBQ
#define W43_NBQ 1      // 2: two weight register sets (even / odd k-steps), every reload two k-steps ahead
#endif
  constexpr int NBQ = W43_NBQ;
  f32x4 bq[NBQ][2][3];
  const unsigned laneb = static_cast<unsigned>(lane) * 48u;
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, static_cast<unsigned>(a.Cin) * a.Cout * (kPts * 4u));
  const unsigned wbase_b = static_cast<unsigned>(((cs * 4 + mz) * 2 + yh) * nk4) * 6144u;
  auto issue_b = [&](int nb, int k4) {
    const int kl = k4 < nk4 ? k4 : 0;
    const unsigned sb = wbase_b + static_cast<unsigned>(kl) * 6144u + nb * 3072u;
#pragma unroll
    for (int q = 0; q < 3; ++q) bq[k4 & (NBQ - 1)][nb][q] = buf_load16(wsrd, laneb + q * 16u, sb);
  };

  f32x4 acc[2][12];
  const int nchunk = a.Cin / CKW;

  // ---- prologue ------------------------------------------------------------------------------------------------------------------------------
  BlockInfo cur = decode(tb0);
  {
    set_offs(cur);
    f32x4 stg[6];
    if (stager) { stage_load(cur, 0u, stg); stage_store(0, stg); }
  }
  __syncthreads();

  int pb = 0;
  for (int itb = 0; itb < niter; ++itb) {
    const BlockInfo nxt = decode(tb0 + (itb + 1 < niter ? itb + 1 : itb) * tstride);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[nb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    raw_read(pb * BUFF * 4);
    if (DBG & 1) {
#pragma unroll
      for (int i = 0; i < 12; ++i) Av[i] = ra[i % 9][0] + static_cast<float>(i);
    }
    if (itb == 0) {
      issue_b(0, 0); issue_b(1, 0);
      if (NBQ == 2) { issue_b(0, 1); issue_b(1, 1); }
    }      // (later blocks: the last k-step of the previous block reloaded k-step 0's weights)

    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int bo = ((chunk + pb) & 1) * BUFF * 4, bn = BUFF * 4 - bo;
      const bool lastc = chunk + 1 == nchunk;
      if (lastc) set_offs(nxt);
      const unsigned schunk = static_cast<unsigned>(lastc ? 0 : chunk + 1) * (CKW * 4u);
      f32x4 stg[6];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (!(DBG & 1)) transform();
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 2 && !(DBG & 4) && stager) {
          if (DBG & 64) {      // (probe: the loads stay alive, nothing is written)
#pragma unroll
            for (int z = 0; z < 6; ++z) asm volatile("" :: "v"(stg[z]));
          } else {
            stage_store(bn, stg);
          }
        }
        if (ks == 3) lds_barrier();
        if (!(DBG & 2)) raw_read(ks < 3 ? bo + (ks + 1) * 16 * CP : bn);
        __builtin_amdgcn_sched_barrier(0);
        const int k4n = chunk * 4 + ks + NBQ;
        const unsigned sbn = wbase_b + static_cast<unsigned>(k4n < nk4 ? k4n : k4n - nk4) * 6144u;      // wraps to the next block's first k-step(s)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
          for (int i = 0; i < 12; ++i) {
            acc[nb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(Av[i], bq[ks & (NBQ - 1)][nb][i >> 2][i & 3], acc[nb][i], 0, 0, 0);
            if ((i & 3) == 3) {      // this quad's four products are issued: reload it with the next k-step's weights
              __builtin_amdgcn_sched_barrier(0);
              if (!(DBG & 8)) bq[ks & (NBQ - 1)][nb][i >> 2] = buf_load16(wsrd, laneb + (i >> 2) * 16u, sbn + nb * 3072u);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (ks == 0 && !(DBG & 4) && stager) {      // the next chunk's z columns, behind a weight batch
          if (DBG & 128) {      // (probe: staged zeros -- the z transform and the LDS writes without the loads)
#pragma unroll
            for (int z = 0; z < 6; ++z) stg[z] = f32x4{0.f, 0.f, 0.f, 0.f};
          } else {
            stage_load(lastc ? nxt : cur, schunk, stg);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------------------------
    // 1. x inverse (A^T of F(4,3)) in registers; the xi_y 2, 3 waves hand their y-partials to the xi_y 0, 1 wave of the same xi_z through the
    //    idle input buffer; 2. that wave completes (y, x) and writes the float4 (oy0 ox0, oy0 ox1, oy1 ox0, oy1 ox1) of every (z-row, x pair,
    //    y tile, cout) where conv_wino.hip's combine expects it; 3. the xi_z combine and everything after it is conv_wino.hip's epilogue.
    const int lb = ((nchunk - 1 + pb) & 1) * BUFF;
    f32x4* sY = reinterpret_cast<f32x4*>(sIn + lb);      // [xi_z][e][x pair p][lane]
    f32x4* sO = reinterpret_cast<f32x4*>(sXc);           // [xi_z][z-row][x pair 0..3][y tile * 16 + cout]
    const int oz0 = cur.z0 + 2 * th, oy0 = cur.y0 + 2 * kq, ox0 = cur.x0 + 2 * xz;
    const int64_t sW = a.Cout, sH = static_cast<int64_t>(a.W) * a.Cout, sD = sH * a.H;
    const int64_t obase = (((static_cast<int64_t>(cur.b) * a.D + oz0) * a.H + oy0) * a.W + ox0) * a.Cout + n0 + tl;
    const bool full = cur.z0 + 4 <= a.D && cur.y0 + 8 <= a.H && cur.x0 + 8 <= a.W;
    constexpr bool SB = FL >= 0 && (FL & kSignBits) != 0, MB = FL >= 0 && (FL & kMaskBits) != 0, NOY = FL >= 0 && (FL & kNoPrimary) != 0;
    const int64_t wbase = (static_cast<int64_t>(cur.id) * a.ncs + cs) * kBitBytesPerBlock + wave * 128 + lane;
    // DF_CONV_ADDUP: the 2x2x2 outputs of this lane share ONE coarse voxel of the skip tensor per cout block; both are requested here, in front
    // of the inverse transform and the exchanges (behind the combine their latency sat in front of the stores: 14.1 vs 13.4 ms per launch)
    float rupv[2] = {0.f, 0.f};
    if ((eflags & DF_CONV_ADDUP) && oz0 < a.D && oy0 < a.H && ox0 < a.W) {
      const float* rp_ = a.residual + (((static_cast<int64_t>(cur.b) * (a.D >> 1) + (oz0 >> 1)) * (a.H >> 1) + (oy0 >> 1)) * (a.W >> 1) + (ox0 >> 1)) * a.Cout + n0 + tl;
      rupv[0] = rp_[0]; rupv[1] = rp_[16];
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float bv = sBias[nb * 16 + tl];
      unsigned mbyte = 0u, sbyte = 0u;
      if (MB) mbyte = a.bits_in[wbase + nb * 64];
      // x inverse of both xi_y of this wave, per accumulator element e (tile 4 kq + e):  X[xyl][ox]
      float X[4][2][4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          const float m0 = acc[nb][l * 6 + 0][e], m1 = acc[nb][l * 6 + 1][e], m2 = acc[nb][l * 6 + 2][e], m3 = acc[nb][l * 6 + 3][e],
                      m4 = acc[nb][l * 6 + 4][e], m5 = acc[nb][l * 6 + 5][e];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          X[e][l][0] = (m0 + s12) + s34;
          X[e][l][1] = __builtin_fmaf(2.f, d34, d12);
          X[e][l][2] = __builtin_fmaf(4.f, s34, s12);
          X[e][l][3] = __builtin_fmaf(8.f, d34, d12) + m5;
        }
      if (yh == 1) {      // xi_y 2, 3:  oy0 += X2,  oy1 += -X2 - X3
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            sY[((mz * 4 + e) * 2 + p) * 64 + lane] = f32x4{X[e][0][2 * p], X[e][0][2 * p + 1], -X[e][0][2 * p] - X[e][1][2 * p],
                                                         -X[e][0][2 * p + 1] - X[e][1][2 * p + 1]};
      }
      float rres[8];
      if (full && (eflags & DF_CONV_RESIDUAL)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) rres[s] = a.residual[obase + nb * 16 + (s >> 2) * sD + ((s >> 1) & 1) * sH + (s & 1) * sW];
      }
      lds_barrier();
      if (yh == 0) {      // xi_y 0, 1:  oy0 = X0 + X1 (+ partner),  oy1 = X1 (+ partner);  tile 4 kq + e = (tz, ty, txh)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const f32x4 q = sY[((mz * 4 + e) * 2 + p) * 64 + lane];
            const f32x4 v = {X[e][0][2 * p] + X[e][1][2 * p] + q[0], X[e][0][2 * p + 1] + X[e][1][2 * p + 1] + q[1], X[e][1][2 * p] + q[2],
                             X[e][1][2 * p + 1] + q[3]};
            const int etz = kq >> 1, ety = (kq & 1) * 2 + (e >> 1), exq = 2 * (e & 1) + p;
            sO[((mz * 2 + etz) * 4 + exq) * 64 + ety * 16 + tl] = v;
          }
      }
      lds_barrier();
      const f32x4 m0 = sO[((0 * 2 + th) * 4 + xz) * 64 + lane], m1 = sO[((1 * 2 + th) * 4 + xz) * 64 + lane];
      const f32x4 m2 = sO[((2 * 2 + th) * 4 + xz) * 64 + lane], m3 = sO[((3 * 2 + th) * 4 + xz) * 64 + lane];
      const f32x4 lo = m0 + m1 + m2, hi = m1 - m2 - m3;
      const float rup = rupv[nb];
      if (full) {
        // Full blocks: everything per-cout (bias, lrelu, sign bits, mask from sign words) happens in the accumulator layout -- one lane = one cout, its
        // 2 x 2 x 2 outputs -- then a 4 x 4 transpose inside every lane quad (conv_wino2d43.hip::quad_t) hands lane i of the quad the outputs (dy, dx) = i
        // of both z for the quad's four couts: the stores (and the fp32 residual / mask operands where a variant has them) are 16-byte accesses, 2 per
        // lane and cout block instead of 8.  Same values, same operations, same order per element: bit-identical.
        f32x4 vz[2];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          float v = (s < 4 ? lo[s & 3] : hi[s & 3]) + bv;
          if (eflags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
          if (SB) sbyte |= v > 0.f ? (1u << s) : 0u;
          vz[s >> 2][s & 3] = v;
        }
        const int qi = tl & 3;
        const bool odd1 = (qi & 1) != 0, odd2 = (qi & 2) != 0;
        auto quad_t = [&](const f32x4& v) -> f32x4 {
          auto dpp1 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)); };
          auto dpp2 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)); };
          const float p0 = dpp1(v[0]), p1 = dpp1(v[1]), p2 = dpp1(v[2]), p3 = dpp1(v[3]);
          const f32x4 a1 = {odd1 ? p1 : v[0], odd1 ? v[1] : p0, odd1 ? p3 : v[2], odd1 ? v[3] : p2};
          const float r0 = dpp2(a1[0]), r1 = dpp2(a1[1]), r2 = dpp2(a1[2]), r3 = dpp2(a1[3]);
          return f32x4{odd2 ? r2 : a1[0], odd2 ? r3 : a1[1], odd2 ? a1[2] : r0, odd2 ? a1[3] : r1};
        };
        // this lane after the transpose: voxel (oz0 + z, oy0 + (qi >> 1), ox0 + (qi & 1)), couts n0 + nb * 16 + 4 (tl >> 2) .. + 3
        const int64_t o4 = obase - tl + 4 * (tl >> 2) + nb * 16 + (qi >> 1) * sH + (qi & 1) * sW;
        const float rup4s = rup;
#pragma unroll
        for (int z = 0; z < 2; ++z) {
          if (eflags & DF_CONV_RESIDUAL) {      // (per-cout layout: the operand of the lane's own 4 outputs of this z)
#pragma unroll
            for (int c = 0; c < 4; ++c) vz[z][c] += rres[z * 4 + c];
          }
          if (eflags & DF_CONV_MASK) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int sidx = z * 4 + c;
              const bool pos = MB ? ((mbyte >> sidx) & 1u) != 0u
                                  : a.mask_src[obase + nb * 16 + (sidx >> 2) * sD + ((sidx >> 1) & 1) * sH + (sidx & 1) * sW] > 0.f;
              vz[z][c] = pos ? vz[z][c] : a.leak * vz[z][c];
            }
          }
          const int64_t o = o4 + z * sD;
          if (!NOY) *reinterpret_cast<f32x4*>(a.y + o) = quad_t(vz[z]);
          if (eflags & DF_CONV_ADDUP) {
            const f32x4 w = {vz[z][0] + rup4s, vz[z][1] + rup4s, vz[z][2] + rup4s, vz[z][3] + rup4s};
            *reinterpret_cast<f32x4*>(a.y2 + o) = quad_t(w);
          }
        }
      } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float v = (s < 4 ? lo[s & 3] : hi[s & 3]) + bv;
        if (eflags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
        const int64_t o = obase + nb * 16 + (s >> 2) * sD + ((s >> 1) & 1) * sH + (s & 1) * sW;
        if (SB) sbyte |= v > 0.f ? (1u << s) : 0u;
        const bool mpos = !MB || ((mbyte >> s) & 1u) != 0u;
        if (oz0 + (s >> 2) < a.D && oy0 + ((s >> 1) & 1) < a.H && ox0 + (s & 1) < a.W) {
          if (eflags & DF_CONV_RESIDUAL) v += a.residual[o];
          if (eflags & DF_CONV_MASK) v = (MB ? mpos : a.mask_src[o] > 0.f) ? v : a.leak * v;
          if (!NOY) a.y[o] = v;
          if (eflags & DF_CONV_ADDUP) a.y2[o] = v + rup;
        }
      }
      }
      if (SB) a.bits_out[wbase + nb * 64] = static_cast<unsigned char>(sbyte);
    }
    pb = (pb + nchunk) & 1;
    cur = nxt;
  }
}

int64_t w43_grid(W43Args& a, int64_t ntb) {
  int64_t grid = df::kCUs;
  a.spx = 1;
  if (8 % a.ncs == 0) {
#ifndef W43_SPX
#define W43_SPX 2      // cout slices per XCD (probe builds: 1 | 2 | 4)
#endif
    a.spx = a.ncs % W43_SPX == 0 ? W43_SPX : (a.ncs % 2 == 0 ? 2 : 1);
    const int xpg = a.ncs / a.spx, ngroups = 8 / xpg;
    const int64_t need = ceil_div(ntb, ngroups) * a.spx * 8;
    if (need < grid) grid = need;
    if ((grid >> 3) % a.spx) grid = ((grid >> 3) / a.spx + 1) * a.spx * 8;
    if (grid > df::kCUs) grid = df::kCUs;
  } else {
    grid = (grid / a.ncs) * a.ncs;
    if (ntb * a.ncs < grid) grid = ntb * a.ncs;
  }
  return grid;
}

}  // namespace

extern "C" {

int64_t df_wino43_packed_elems(int64_t cin, int64_t cout, int mode) {
  (void)mode;
  return kPts * cin * cout;
}

int df_wino43_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_wino43_pack_weights: null pointer");
  DF_REQUIRE(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0 && (mode == 0 || mode == 1), DF_ESHAPE,
             "df_wino43_pack_weights: cin, cout must be multiples of 32; mode 0|1");
  int64_t g = ceil_div(cin * cout, 64);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(wino43_pack_kernel, dim3((unsigned)g), dim3(64), 0, df::as_stream(stream), w, wp, (int)cin, (int)cout, mode);
  return df::launched("df_wino43_pack_weights");
}

int df_wino43_conv(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src, const void* mask_bits,
                   float* y, float* y2, void* sign_bits, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags,
                   float leak, df_stream_t stream) {
  DF_REQUIRE(x && wp && (y || (y2 && sign_bits)), DF_EINVAL, "df_wino43_conv: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DF_EINVAL, "df_wino43_conv: non-positive extent");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "df_wino43_conv: Cin, Cout must be multiples of 32");
  DF_REQUIRE(D * H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 22), DF_ESHAPE,
             "df_wino43_conv: one batch volume must stay below 2 GiB, Cin * Cout below 4 Mi");
  DF_REQUIRE(!(flags & ~(DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_RESIDUAL | DF_CONV_MASK | DF_CONV_ADDUP)), DF_EINVAL, "df_wino43_conv: unknown flag");
  DF_REQUIRE(!(flags & DF_CONV_BIAS) || bias, DF_EINVAL, "df_wino43_conv: DF_CONV_BIAS without bias");
  DF_REQUIRE(!(flags & (DF_CONV_RESIDUAL | DF_CONV_ADDUP)) || residual, DF_EINVAL, "df_wino43_conv: RESIDUAL / ADDUP without the tensor to add");
  DF_REQUIRE(!((flags & DF_CONV_RESIDUAL) && (flags & DF_CONV_ADDUP)), DF_EINVAL, "df_wino43_conv: RESIDUAL and ADDUP exclude each other");
  DF_REQUIRE(!(flags & DF_CONV_ADDUP) || (y2 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0), DF_EINVAL,
             "df_wino43_conv: ADDUP needs y2 and even extents (the output of a 2x up-sampling block)");
  DF_REQUIRE(!(flags & DF_CONV_MASK) || ((mask_src != nullptr) != (mask_bits != nullptr)), DF_EINVAL,
             "df_wino43_conv: DF_CONV_MASK needs exactly one of mask_src / mask_bits");
  DF_REQUIRE(y || ((flags & DF_CONV_ADDUP) && sign_bits), DF_EINVAL, "df_wino43_conv: y may be null only with ADDUP + sign_bits");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x) && df::aligned16(mask_bits) && df::aligned16(sign_bits) && df::aligned16(y) && df::aligned16(y2),
             DF_EALIGN, "df_wino43_conv: x, y, y2, packed weights and bit words must be 16-byte aligned");
  W43Args a;
  a.x = x; a.wp = wp; a.bias = bias; a.residual = residual; a.mask_src = mask_src; a.y = y; a.y2 = y2;
  a.bits_out = static_cast<unsigned char*>(sign_bits); a.bits_in = static_cast<const unsigned char*>(mask_bits);
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino43_conv: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = flags; a.leak = leak;
  const int64_t grid = w43_grid(a, ntb);
  hipStream_t s = df::as_stream(stream);
  const int f = flags | (sign_bits ? kSignBits : 0) | (mask_bits ? kMaskBits : 0) | (!y ? kNoPrimary : 0);
#define DF_W43(F) hipLaunchKernelGGL((wino43_kernel<F>), dim3((unsigned)grid), dim3(kT), 0, s, a)
  switch (f) {
    case DF_CONV_BIAS | DF_CONV_LRELU: DF_W43(DF_CONV_BIAS | DF_CONV_LRELU); break;
    case DF_CONV_BIAS | DF_CONV_LRELU | kSignBits: DF_W43(DF_CONV_BIAS | DF_CONV_LRELU | kSignBits); break;
    case DF_CONV_MASK | kMaskBits: DF_W43(DF_CONV_MASK | kMaskBits); break;
    case DF_CONV_MASK: DF_W43(DF_CONV_MASK); break;
    case DF_CONV_RESIDUAL: DF_W43(DF_CONV_RESIDUAL); break;
    case 0: DF_W43(0); break;
    case DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP: DF_W43(DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP); break;
    case DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP | kSignBits | kNoPrimary: DF_W43(DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP | kSignBits | kNoPrimary); break;
    default:
      DF_REQUIRE(!sign_bits && !mask_bits && y, DF_EINVAL, "df_wino43_conv: this flag combination has no sign-word variant");
      DF_W43(-1);
  }
#undef DF_W43
  return df::launched("df_wino43_conv");
}

#ifdef DF_W43_PROBE
// probe build: df_wino43_conv(BIAS | LRELU) with a DBG variant (results wrong by construction unless dbg == 0)
int df_wino43_probe(const float* x, const float* wp, const float* bias, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin,
                    int64_t Cout, float leak, int dbg, df_stream_t stream) {
  W43Args a;
  a.x = x; a.wp = wp; a.bias = bias; a.residual = nullptr; a.mask_src = nullptr; a.y = y; a.y2 = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32); a.ntb = (int)ntb; a.flags = 9; a.leak = leak;
  const int64_t grid = w43_grid(a, ntb);
  hipStream_t s = df::as_stream(stream);
#define DF_P(V) case V: hipLaunchKernelGGL((wino43_kernel<9, V>), dim3((unsigned)grid), dim3(kT), 0, s, a); break
  switch (dbg) { DF_P(0); DF_P(1); DF_P(2); DF_P(3); DF_P(4); DF_P(8); DF_P(7); DF_P(15); DF_P(12); DF_P(64); DF_P(128); default: return DF_EINVAL; }
#undef DF_P
  return df::launched("df_wino43_probe");
}
#endif

}  // extern "C"
