// 3x3 SAME stride-1 convolution (2-D) as Winograd F(2,3) x F(4,3) (y, x) on the fp32 matrix cores -- the 2-D twin of conv_wino43.hip
// (reference: slim.conv2d behind ops.py:12-13, called from model.py:24-28 -- the 128 -> 128 layers of GeneratorBE).
//
//   2 x 4 output tile from a 4 x 6 input tile: 24 transform points per 8 outputs = 3 MFMA MACs per output pixel and (cin, cout) pair, where
//   F(2,3)^2 (conv_wino2d.hip) needs 4 and the direct form 9.  All arithmetic fp32; accuracy as the 3-D family (about one bit below F(2,3)^2).
//
// Decomposition: persistent 8-wave workgroup per CU, XCD-pinned (cout slice, tile block) items as conv_wino2d.hip; tile block = 16 x 32 output
// pixels = 8 x 8 tiles = four MFMA row blocks, x 32 couts.  Wave = (tile-row pair rp, xi_y pair yh): 16 tiles x 12 points x 2 cout blocks = 24 MFMA
// 16x16x4 per k-step (conv_wino2d.hip: 32), 96 accumulators; the two waves of a SIMD are the two xi_y halves of the same tiles.
//   * input: the 18 x 34 halo block of a 16-channel chunk in LDS, channel-major [c][y*48 + x] (row pitch 48: the two tile rows of a lane group
//     sit 16 bank pairs apart, channel pitch 866: 433 odd separates cin -- a wave's ds_read_b64 are conflict-free), double buffered, one LDS-only
//     barrier per chunk, SAME padding by the buffer range check;
//   * A operand: lane = (tile, cin % 4) reads 3 rows x 6 columns (9 ds_read_b64), y stage 6 packed ops, x stage = B^T of F(4,3), 6 packed ops per row;
//   * B operand: U packed [cs][xi_y pair][cin/4][cout block][cin%4][cout%16][12 points]: three 16-byte loads per lane and block straight from
//     L2 / L1 (the four waves of an xi_y half read the same words);
//   * inverse: x (A^T of F(4,3)) in registers; the xi_y halves meet through the idle LDS buffer in ONE exchange -- the yh = 0 wave finishes the
//     x pairs 0 of its tiles, the yh = 1 wave the pairs 1, so all eight waves store;
//   * epilogue: a 4 x 4 transpose inside every lane quad (DPP quad_perm) turns the accumulator layout (one cout per lane, its 2 x 2 pixel patch) into
//     (one pixel per lane, four couts): every store / residual / mask access is 16 bytes per lane.  Bias, lrelu, residual, lrelu mask from an fp32
//     activation or from SIGN WORDS (one 32-bit word per thread and tile block = its 32 outputs; written by the forward conv, read by the dgrad of the
//     same geometry), and the block tail of an up-sampling block (add-up of the coarse skip tensor, only the sign words of the activation kept) with its
//     backward twin lrelu_words2d_bwd_pool_kernel.  Forward convs AND dgrads of the 2-D generator run here (ops.WINO2D_FAMILY "f24").
#include "df_common.hpp"
#include "conv_args.hpp"

namespace {

using df::ceil_div;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;
constexpr int CKW = 16;
constexpr int HY = 18, HX = 34, HV = HY * HX;      // halo block (16 + 2) x (32 + 2)
constexpr int PY = 48;                             // LDS row pitch (dwords)
constexpr int CP = HY * PY + 2;                    // dwords per channel plane (866)
constexpr int NLOAD = 5;                           // ceil(612 * 4 float4 pieces / 512 threads)
constexpr int BUFF = CKW * CP;
constexpr int kPts = 24;
constexpr int kSignBits = 64, kMaskBits = 128, kNoPrimary = 256;      // internal epilogue flags (compile-time variants only), as conv_wino43.hip

struct W2Args {
  const float* x;
  const float* wp;
  const float* bias;
  const float* residual;
  const float* mask_src;
  float* y;
  float* y2;                   // DF_CONV_ADDUP: y2 = y + nearest_up2x(residual), residual = the COARSE tensor [B, H/2, W/2, Cout]
  unsigned* bits_out;          // sign words of the output (kSignBits): one 32-bit word per (tile block, cout slice, thread)
  const unsigned* bits_in;     // ... of the activation whose lrelu slope masks this dgrad (kMaskBits)
  int B, H, W, Cin, Cout;
  int nby, nbx, ntb, ncs, spx;
  int flags;
  float leak;
};

// ---- weight transform + packing: Up[cs][yh][k4][nb][kq][j][(xy & 1) * 6 + xx] = sum_taps G2[xy][ty] G4[xx][tx] g[tap][4 k4 + kq][32 cs + 16 nb + j]
// mode 0: g[tap][k][n] = w[tap][k][n];  mode 1: g[tap][k][n] = w[8 - tap][n][k]  (dgrad operand)
__global__ __launch_bounds__(64) void wino2d43_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout, int mode) {
  const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int64_t nfil = static_cast<int64_t>(K) * N;
  for (int64_t f = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; f < nfil; f += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(f % N), k = static_cast<int>(f / N);
    double g[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      g[tap] = static_cast<double>(mode == 0 ? w[(static_cast<int64_t>(tap) * cin + k) * cout + n] : w[(static_cast<int64_t>(8 - tap) * cin + n) * cout + k]);
    double gx[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
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
    for (int yh = 0; yh < 2; ++yh) {      // a lane record = the 12 points of an xi_y pair: written as three 16-byte stores (24 scattered 4-byte ones before)
      float o[12];
#pragma unroll
      for (int xyl = 0; xyl < 2; ++xyl) {
        const int xy = 2 * yh + xyl;
#pragma unroll
        for (int xx = 0; xx < 6; ++xx) {
          const double a = gx[0][xx], b = gx[1][xx], c = gx[2][xx];
          o[xyl * 6 + xx] = static_cast<float>(xy == 0 ? a : xy == 3 ? c : xy == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c));
        }
      }
      const int64_t idx = ((((((static_cast<int64_t>(cs) * 2 + yh) * (K / 4) + k4) * 2 + nb) * 4 + kq) * 16 + j) * 12);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4*>(wp + idx + 4 * q) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
    }
  }
}

struct Blk { const float* xb; int hoff, b, y0, x0, id; };

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


template <int FL>
__global__ __launch_bounds__(kT, 1) void wino2d43_kernel(const W2Args a) {
  __shared__ __attribute__((aligned(16))) float sIn[2 * BUFF];
  __shared__ __attribute__((aligned(16))) float sXb[8192];      // exchange area of cout block 1 (block 0 goes through the idle input buffer)
  __shared__ __attribute__((aligned(16))) float sBias[32];

  const int eflags = FL >= 0 ? FL : a.flags;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 15, kq = lane >> 4;
  const int rp = wave & 3, yh = wave >> 2;      // tile-row pair, xi_y pair

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
  auto decode = [&](int t) -> Blk {
    Blk bi;
    bi.id = t;
    const int bx = t % a.nbx;
    const int t2 = t / a.nbx;
    const int by = t2 % a.nby;
    bi.b = t2 / a.nby;
    bi.y0 = by * 16; bi.x0 = bx * 32;
    bi.xb = a.x + static_cast<int64_t>(bi.b) * a.H * a.W * a.Cin;
    bi.hoff = ((bi.y0 - 1) * a.W + (bi.x0 - 1)) * a.Cin;
    return bi;
  };

  // ---- staging plan: 5 float4 pieces of the 612-pixel x 16-channel halo block per thread ------------------------------------------------------
  int ldst[NLOAD];
  unsigned so[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    int p = it * kT + tid;
    if (p > HV * 4 - 1) p = HV * 4 - 1;
    const int hv = p >> 2, q4 = p & 3;
    ldst[it] = ((q4 * 4) * CP + (hv / HX) * PY + hv % HX) * 4;
  }
  const unsigned img_bytes = static_cast<unsigned>(a.H * a.W) * a.Cin * 4u;
  auto set_offs = [&](const Blk& bi) {
    int tido = tid;      // (opaque: keeps LLVM from hoisting -- and spilling -- the pieces' decomposition, see conv_wino43.hip)
    asm volatile("" : "+v"(tido));
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      int p = it * kT + tido;
      if (p > HV * 4 - 1) p = HV * 4 - 1;
      const int hv = p >> 2, q4 = p & 3;
      const int hx = hv % HX, hy = hv / HX;
      const int gy = bi.y0 - 1 + hy, gx = bi.x0 - 1 + hx;
      const bool ok = static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) && static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
      so[it] = ok ? static_cast<unsigned>(bi.hoff + (hy * a.W + hx) * a.Cin + q4 * 4) * 4u : 0x80000000u;
    }
  };
  char* sInB = reinterpret_cast<char*>(sIn);
  auto stage_store = [&](int it, int bufbytes, const f32x4& v) {
    float* d = reinterpret_cast<float*>(sInB + (ldst[it] + bufbytes));
    d[0] = v[0]; d[CP] = v[1]; d[2 * CP] = v[2]; d[3 * CP] = v[3];
  };

  // ---- A operand: lane = (tile (tr, tx), cin kq): rows yh .. yh + 2 of the tile's 4 halo rows, 6 columns ------------------------------------
  const int tr = tl >> 3, tx = tl & 7;
  const int offAb = (kq * CP + (2 * (2 * rp + tr) + yh) * PY + 4 * tx) * 4;
  f32x2 ra[9];
  float Av[12];
  auto raw_read = [&](int idxbytes) {
    int ia = idxbytes + offAb;
    asm volatile("" : "+v"(ia));
    __builtin_assume((ia & 7) == 0);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) ra[r * 3 + j] = *reinterpret_cast<const f32x2*>(sInB + ia + (r * PY + 2 * j) * 4);
  };
  auto xstage = [&](const f32x2 (&u)[3], float* o) {      // B^T of F(4,3) in six packed ops, as conv_wino43.hip
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
    if (yh == 0) {      // rows (0, 1, 2): xi_y 0 = r0 - r2, xi_y 1 = r1 + r2
#pragma unroll
      for (int j = 0; j < 3; ++j) { Ua[j] = pk_sub(ra[j], ra[6 + j]); Ub[j] = pk_add(ra[3 + j], ra[6 + j]); }
    } else {            // rows (1, 2, 3): xi_y 2 = r2 - r1, xi_y 3 = r1 - r3
#pragma unroll
      for (int j = 0; j < 3; ++j) { Ua[j] = pk_sub(ra[3 + j], ra[j]); Ub[j] = pk_sub(ra[j], ra[6 + j]); }
    }
    xstage(Ua, Av);
    xstage(Ub, Av + 6);
  };

  // ---- B operand ------------------------------------------------------------------------------------------------------------------------------
  const int nk4 = a.Cin >> 2;
  f32x4 bq[2][3];
  const unsigned laneb = static_cast<unsigned>(lane) * 48u;
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, static_cast<unsigned>(a.Cin) * a.Cout * (kPts * 4u));
  const unsigned wbase_b = static_cast<unsigned>((cs * 2 + yh) * nk4) * 6144u;
  auto issue_b = [&](int nb, int k4) {
    const unsigned sb = wbase_b + static_cast<unsigned>(k4 < nk4 ? k4 : 0) * 6144u + nb * 3072u;
#pragma unroll
    for (int q = 0; q < 3; ++q) bq[nb][q] = buf_load16(wsrd, laneb + q * 16u, sb);
  };

  f32x4 acc[2][12];
  const int nchunk = a.Cin / CKW;

  Blk cur = decode(tb0);
  {
    set_offs(cur);
    const __amdgpu_buffer_rsrc_t srd0 = make_srd(cur.xb, img_bytes);
    f32x4 stg[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) stg[it] = buf_load16(srd0, so[it], 0u);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) stage_store(it, 0, stg[it]);
  }
  __syncthreads();

  int pb = 0;
  for (int itb = 0; itb < niter; ++itb) {
    const Blk nxt = decode(tb0 + (itb + 1 < niter ? itb + 1 : itb) * tstride);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[nb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    raw_read(pb * BUFF * 4);
    if (itb == 0) { issue_b(0, 0); issue_b(1, 0); }

    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int bo = ((chunk + pb) & 1) * BUFF * 4, bn = BUFF * 4 - bo;
      const bool lastc = chunk + 1 == nchunk;
      if (lastc) set_offs(nxt);
      const __amdgpu_buffer_rsrc_t ssrd = make_srd(lastc ? nxt.xb : cur.xb, img_bytes);
      const unsigned schunk = static_cast<unsigned>(lastc ? 0 : chunk + 1) * (CKW * 4u);
      f32x4 stg[NLOAD];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        transform();
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 2) {
#pragma unroll
          for (int it = 0; it < NLOAD; ++it) stage_store(it, bn, stg[it]);
        }
        if (ks == 3) lds_barrier();
        raw_read(ks < 3 ? bo + (ks + 1) * 16 * CP : bn);
        __builtin_amdgcn_sched_barrier(0);
        const int k4n = chunk * 4 + ks + 1;
        const unsigned sbn = wbase_b + static_cast<unsigned>(k4n < nk4 ? k4n : 0) * 6144u;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
          for (int i = 0; i < 12; ++i) {
            acc[nb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(Av[i], bq[nb][i >> 2][i & 3], acc[nb][i], 0, 0, 0);
            if ((i & 3) == 3) {
              __builtin_amdgcn_sched_barrier(0);
              bq[nb][i >> 2] = buf_load16(wsrd, laneb + (i >> 2) * 16u, sbn + nb * 3072u);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (ks == 0) {
#pragma unroll
          for (int it = 0; it < NLOAD; ++it) stg[it] = buf_load16(ssrd, so[it], schunk);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: x inverse in registers; the two xi_y halves of a tile-row pair meet through the idle LDS buffer.  The yh = 0 wave finishes the
    // x pairs p = 0 of every tile (output columns 4 tx, 4 tx + 1), the yh = 1 wave the pairs p = 1: each sends the other half of its partials and
    // receives the partner's -- one exchange for both cout blocks, all eight waves store.
    const int lb = ((nchunk - 1 + pb) & 1) * BUFF;
    f32x4* sYn[2] = {reinterpret_cast<f32x4*>(sIn + lb), reinterpret_cast<f32x4*>(sXb)};      // per cout block: [yh of the sender][rp][e][lane], 32 KB
    const int etr = kq >> 1;                             // tile of accumulator element e: row etr, column (kq & 1) * 4 + e
    const int oy = cur.y0 + 2 * (2 * rp + etr);
    const bool full = cur.y0 + 16 <= a.H && cur.x0 + 32 <= a.W;      // workgroup-uniform
    f32x4 mine[2][4];      // [nb][e]: this wave's contribution to the x pair it finishes
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float X[2][4];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          const float m0 = acc[nb][l * 6 + 0][e], m1 = acc[nb][l * 6 + 1][e], m2 = acc[nb][l * 6 + 2][e], m3 = acc[nb][l * 6 + 3][e],
                      m4 = acc[nb][l * 6 + 4][e], m5 = acc[nb][l * 6 + 5][e];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          X[l][0] = (m0 + s12) + s34;
          X[l][1] = __builtin_fmaf(2.f, d34, d12);
          X[l][2] = __builtin_fmaf(4.f, s34, s12);
          X[l][3] = __builtin_fmaf(8.f, d34, d12) + m5;
        }
        f32x4 part[2];      // [x pair p]: (oy0 ox 2p, oy0 ox 2p+1, oy1 ox 2p, oy1 ox 2p+1)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (yh == 0) part[p] = f32x4{X[0][2 * p] + X[1][2 * p], X[0][2 * p + 1] + X[1][2 * p + 1], X[1][2 * p], X[1][2 * p + 1]};           // xi_y 0, 1
          else part[p] = f32x4{X[0][2 * p], X[0][2 * p + 1], -X[0][2 * p] - X[1][2 * p], -X[0][2 * p + 1] - X[1][2 * p + 1]};              // xi_y 2, 3
        }
        mine[nb][e] = yh == 0 ? part[0] : part[1];
        sYn[nb][((yh * 4 + rp) * 4 + e) * 64 + lane] = yh == 0 ? part[1] : part[0];
      }
    }
    const int etx0 = (kq & 1) * 4;
    // The combined 2 x 2 output patch of a tile's x pair sits in ONE lane per cout (lane = (cout tl, tile quad kq), element e): a 4 x 4 transpose
    // inside every lane quad (couts 4 m .. 4 m + 3) x (the patch's 4 pixels) hands lane i of the quad pixel i for the four couts -- every store,
    // residual load and lrelu-mask load of the epilogue is then ONE 16-byte access per lane (four lanes = the 64 contiguous bytes of a pixel's 16
    // couts) instead of four 4-byte ones: 8 stores per lane and tile block instead of 32, and the masked dgrad's 32 mask loads become 8.
    // Same values through the same operations in the same order: bit-identical to the per-cout epilogue.
    const int qi = tl & 3, qm = tl >> 2;
    const bool odd1 = (qi & 1) != 0, odd2 = (qi & 2) != 0;
    auto quad_t = [&](const f32x4& v) -> f32x4 {
      // stage 1: swap bit 0 of (lane, register): registers (0, 1) and (2, 3) with lane ^ 1;  stage 2: bit 1: registers (0, 2) and (1, 3) with lane ^ 2
      auto dpp1 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)); };
      auto dpp2 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)); };
      const float p0 = dpp1(v[0]), p1 = dpp1(v[1]), p2 = dpp1(v[2]), p3 = dpp1(v[3]);
      const f32x4 a1 = {odd1 ? p1 : v[0], odd1 ? v[1] : p0, odd1 ? p3 : v[2], odd1 ? v[3] : p2};
      const float r0 = dpp2(a1[0]), r1 = dpp2(a1[1]), r2 = dpp2(a1[2]), r3 = dpp2(a1[3]);
      return f32x4{odd2 ? r2 : a1[0], odd2 ? r3 : a1[1], odd2 ? a1[2] : r0, odd2 ? a1[3] : r1};
    };
    const int py = oy + (qi >> 1);
    // Sign words (the 2-D twin of conv_wino43.hip's): after the transpose a lane holds, per cout block nb and tile column e, pixel qi of the patch
    // for couts 4 qm .. 4 qm + 3 -- 2 x 4 x 4 = 32 outputs per tile block = ONE 32-bit word, bit (nb * 4 + e) * 4 + c.  The forward conv whose
    // output only serves as the lrelu mask of the next layer's dgrad writes (activation > 0) there; that dgrad -- the same kernel on a tensor of the
    // same shape, hence the same (tile block, cout slice, thread) <-> output mapping -- reads the word instead of 32 fp32 activations.
    constexpr bool SB = FL >= 0 && (FL & kSignBits) != 0, MB = FL >= 0 && (FL & kMaskBits) != 0, NOY = FL >= 0 && (FL & kNoPrimary) != 0;
    const int64_t widx = (static_cast<int64_t>(cur.id) * a.ncs + cs) * kT + tid;
    unsigned sword = 0u, mword = 0u;
    if (MB) mword = a.bits_in[widx];
    const int64_t sW_ = a.Cout;
    // element offset of (pixel qi of the patch of tile column etx0, couts 4 qm ..): + 4 e sW_ per tile column, + 16 per cout block
    const int64_t obase = ((static_cast<int64_t>(cur.b) * a.H + py) * a.W + cur.x0 + 4 * etx0 + 2 * yh + (qi & 1)) * a.Cout + n0 + 4 * qm;
    // ... and of the coarse pixel (py >> 1, px >> 1) of the add-up operand [B, H/2, W/2, Cout]: + 2 e sW_ per tile column (4 fine = 2 coarse pixels)
    const int64_t cbase = ((static_cast<int64_t>(cur.b) * (a.H >> 1) + (py >> 1)) * (a.W >> 1) + ((cur.x0 + 4 * etx0 + 2 * yh) >> 1)) * a.Cout + n0 + 4 * qm;
    lds_barrier();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const f32x4 bv4 = *reinterpret_cast<const f32x4*>(&sBias[nb * 16 + 4 * qm]);
      f32x4 rres[4], rmsk[4];      // residual / mask / add-up operands of this cout block: the loads are issued together (full blocks)
      if (full && (eflags & (DF_CONV_RESIDUAL | DF_CONV_MASK | DF_CONV_ADDUP))) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t o = obase + nb * 16 + 4 * e * sW_;
          if (eflags & DF_CONV_RESIDUAL) rres[e] = *reinterpret_cast<const f32x4*>(a.residual + o);
          if ((eflags & DF_CONV_MASK) && !MB) rmsk[e] = *reinterpret_cast<const f32x4*>(a.mask_src + o);
          // ADDUP: the four lanes of a quad (the 2 x 2 patch) share ONE coarse pixel of the skip tensor
          if (eflags & DF_CONV_ADDUP) rres[e] = *reinterpret_cast<const f32x4*>(a.residual + cbase + nb * 16 + 2 * e * sW_);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 q = sYn[nb][(((1 - yh) * 4 + rp) * 4 + e) * 64 + lane];
        // (fixed order: the xi_y 0, 1 half first, then the xi_y 2, 3 half -- whichever wave finishes)
        const f32x4 v4 = quad_t(yh == 0 ? mine[nb][e] + q : q + mine[nb][e]);      // couts 4 qm .. 4 qm + 3 of pixel qi
        const int px = cur.x0 + 4 * (etx0 + e) + 2 * yh + (qi & 1);
        const int64_t o = obase + nb * 16 + 4 * e * sW_;
        f32x4 v = v4 + bv4;
        if (eflags & DF_CONV_LRELU) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], a.leak * v[c]);
        }
        if (SB) {
#pragma unroll
          for (int c = 0; c < 4; ++c) sword |= v[c] > 0.f ? (1u << ((nb * 4 + e) * 4 + c)) : 0u;
        }
        if (full) {
          if (eflags & DF_CONV_RESIDUAL) v += rres[e];
          if (eflags & DF_CONV_MASK) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (MB ? ((mword >> ((nb * 4 + e) * 4 + c)) & 1u) != 0u : rmsk[e][c] > 0.f) ? v[c] : a.leak * v[c];
          }
          if (!NOY) *reinterpret_cast<f32x4*>(a.y + o) = v;
          if (eflags & DF_CONV_ADDUP) *reinterpret_cast<f32x4*>(a.y2 + o) = v + rres[e];
        } else if (py < a.H && px < a.W) {
          if (eflags & DF_CONV_RESIDUAL) v += *reinterpret_cast<const f32x4*>(a.residual + o);
          if (eflags & DF_CONV_MASK) {
            f32x4 mk = {0.f, 0.f, 0.f, 0.f};
            if (!MB) mk = *reinterpret_cast<const f32x4*>(a.mask_src + o);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (MB ? ((mword >> ((nb * 4 + e) * 4 + c)) & 1u) != 0u : mk[c] > 0.f) ? v[c] : a.leak * v[c];
          }
          if (!NOY) *reinterpret_cast<f32x4*>(a.y + o) = v;
          if (eflags & DF_CONV_ADDUP) *reinterpret_cast<f32x4*>(a.y2 + o) = v + *reinterpret_cast<const f32x4*>(a.residual + cbase + nb * 16 + 2 * e * sW_);
        }
      }
    }
    if (SB) a.bits_out[widx] = sword;
    lds_barrier();      // the exchange area is the next block's staging buffer
    pb = (pb + nchunk) & 1;
    cur = nxt;
  }
}

// The backward tail of a 2-D up-sampling block on the sign words of its last conv (the 2-D twin of conv_wino.hip's lrelu_bits_bwd_pool_kernel): one
// workgroup = the 512 words of a (tile block, cout slice), thread = word = the same (pixel qi of a 2 x 2 patch, couts 4 qm ..) x (cout block nb, tile
// column e) outputs as in wino2d43_kernel's epilogue.  gx = gy * lrelu'(activation) from the bits; gpool = the 2 x 2 sum-pool of gy (the skip path's
// gradient): the patch is the lane quad, summed (dy, dx) ascending like lrelu_bwd_pool_kernel / upsample_bwd_kernel (bit-identical).
__global__ __launch_bounds__(kT) void lrelu_words2d_bwd_pool_kernel(const float* __restrict__ gy, const unsigned* __restrict__ bits, float* __restrict__ gx,
                                                                   float* __restrict__ gpool, float leak, int H, int W, int C, int nby, int nbx, int ncs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 15, kq = lane >> 4, rp = wave & 3, yh = wave >> 2;
  const int qi = tl & 3, qm = tl >> 2;
  const int cs = blockIdx.x % ncs;
  int t = blockIdx.x / ncs;
  const int bx = t % nbx; t /= nbx;
  const int by = t % nby;
  const int b = t / nby;
  const unsigned word = bits[static_cast<int64_t>(blockIdx.x) * kT + tid];
  const int py = by * 16 + 2 * (2 * rp + (kq >> 1)) + (qi >> 1);
  const int px0 = bx * 32 + 4 * ((kq & 1) * 4) + 2 * yh + (qi & 1);
  const int c0 = cs * 32 + 4 * qm;
  auto bc = [](float x, int k) -> float {      // broadcast lane k of the quad
    const int v = __builtin_bit_cast(int, x);
    const int r = k == 0 ? __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true) : k == 1 ? __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true)
                : k == 2 ? __builtin_amdgcn_mov_dpp(v, 0xAA, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(v, 0xFF, 0xf, 0xf, true);
    return __builtin_bit_cast(float, r);
  };
  // (e outer, nb inner: the two 64-byte halves of a pixel's 128-byte cout-slice line are requested back to back; all eight loads first)
  f32x4 gq[4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int px = px0 + 4 * e;
      gq[e][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (py < H && px < W) gq[e][nb] = *reinterpret_cast<const f32x4*>(gy + ((static_cast<int64_t>(b) * H + py) * W + px) * C + c0 + nb * 16);
    }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int px = px0 + 4 * e;
      const bool ok = py < H && px < W;      // (H, W even: a 2 x 2 patch is inside or outside as a whole)
      const int64_t o = ((static_cast<int64_t>(b) * H + py) * W + px) * C + c0 + nb * 16;
      const f32x4 g = gq[e][nb];
      f32x4 d, p;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d[c] = ((word >> ((nb * 4 + e) * 4 + c)) & 1u) != 0u ? g[c] : leak * g[c];
        float acc = 0.f;
        acc += bc(g[c], 0); acc += bc(g[c], 1); acc += bc(g[c], 2); acc += bc(g[c], 3);
        p[c] = acc;
      }
      if (ok) {
        *reinterpret_cast<f32x4*>(gx + o) = d;
        if (qi == 0) *reinterpret_cast<f32x4*>(gpool + ((static_cast<int64_t>(b) * (H >> 1) + (py >> 1)) * (W >> 1) + (px >> 1)) * C + c0 + nb * 16) = p;
      }
    }
}

int64_t w2_grid(W2Args& a, int64_t ntb) {
  int64_t grid = df::kCUs;
  a.spx = 1;
  if (8 % a.ncs == 0) {
    a.spx = a.ncs % 2 == 0 ? 2 : 1;
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

int64_t df_wino2d43_packed_elems(int64_t cin, int64_t cout, int mode) {
  (void)mode;
  return kPts * cin * cout;
}

int df_wino2d43_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_wino2d43_pack_weights: null pointer");
  DF_REQUIRE(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0 && (mode == 0 || mode == 1), DF_ESHAPE,
             "df_wino2d43_pack_weights: cin, cout must be multiples of 32; mode 0|1");
  int64_t g = ceil_div(cin * cout, 64);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(wino2d43_pack_kernel, dim3((unsigned)g), dim3(64), 0, df::as_stream(stream), w, wp, (int)cin, (int)cout, mode);
  return df::launched("df_wino2d43_pack_weights");
}

static int w2_conv(const char* fn, const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src, const void* mask_bits,
                   float* y, void* sign_bits, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream) {
  DF_REQUIRE(x && wp && y, DF_EINVAL, "%s: null pointer", fn);
  DF_REQUIRE(B > 0 && H > 0 && W > 0, DF_EINVAL, "%s: non-positive extent", fn);
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "%s: Cin, Cout must be multiples of 32", fn);
  DF_REQUIRE(H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE, "%s: one image must stay below 2 GiB", fn);
  DF_REQUIRE(!(flags & ~(DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_RESIDUAL | DF_CONV_MASK)), DF_EINVAL, "%s: unknown flag", fn);
  DF_REQUIRE(!(flags & DF_CONV_BIAS) || bias, DF_EINVAL, "%s: DF_CONV_BIAS without bias", fn);
  DF_REQUIRE(!(flags & DF_CONV_RESIDUAL) || residual, DF_EINVAL, "%s: DF_CONV_RESIDUAL without residual", fn);
  DF_REQUIRE(!(flags & DF_CONV_MASK) || ((mask_src != nullptr) != (mask_bits != nullptr)), DF_EINVAL, "%s: DF_CONV_MASK needs exactly one of mask_src / mask_bits", fn);
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x) && df::aligned16(y) && (!(flags & DF_CONV_RESIDUAL) || df::aligned16(residual)) &&
                 (!(flags & DF_CONV_MASK) || df::aligned16(mask_src)) && df::aligned16(mask_bits) && df::aligned16(sign_bits),
             DF_EALIGN, "%s: x, y, residual, mask_src, the bit words and the packed weights must be 16-byte aligned", fn);
  W2Args a;
  a.x = x; a.wp = wp; a.bias = bias; a.residual = residual; a.mask_src = mask_src; a.y = y; a.y2 = nullptr;
  a.bits_out = static_cast<unsigned*>(sign_bits); a.bits_in = static_cast<const unsigned*>(mask_bits);
  a.B = (int)B; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nby = (int)ceil_div(H, 16); a.nbx = (int)ceil_div(W, 32);
  const int64_t ntb = B * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "%s: too many workgroups", fn);
  a.ntb = (int)ntb;
  a.flags = flags; a.leak = leak;
  const int64_t grid = w2_grid(a, ntb);
  const dim3 g((unsigned)grid), b(kT);
  hipStream_t s = df::as_stream(stream);
  if (sign_bits || mask_bits) {
    if (flags == (DF_CONV_BIAS | DF_CONV_LRELU) && sign_bits && !mask_bits) hipLaunchKernelGGL((wino2d43_kernel<DF_CONV_BIAS | DF_CONV_LRELU | kSignBits>), g, b, 0, s, a);
    else if (flags == DF_CONV_MASK && mask_bits && !sign_bits) hipLaunchKernelGGL((wino2d43_kernel<DF_CONV_MASK | kMaskBits>), g, b, 0, s, a);
    else return df::fail(DF_EINVAL, "%s: sign words exist for BIAS | LRELU (+ sign_bits) and MASK (+ mask_bits) only", fn);
  }
  else if (flags == (DF_CONV_BIAS | DF_CONV_LRELU)) hipLaunchKernelGGL((wino2d43_kernel<DF_CONV_BIAS | DF_CONV_LRELU>), g, b, 0, s, a);
  else if (flags == DF_CONV_MASK) hipLaunchKernelGGL((wino2d43_kernel<DF_CONV_MASK>), g, b, 0, s, a);
  else if (flags == DF_CONV_RESIDUAL) hipLaunchKernelGGL((wino2d43_kernel<DF_CONV_RESIDUAL>), g, b, 0, s, a);
  else hipLaunchKernelGGL((wino2d43_kernel<-1>), g, b, 0, s, a);
  return df::launched(fn);
}

int df_wino2d43_conv(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src, float* y, int64_t B,
                     int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream) {
  return w2_conv("df_wino2d43_conv", x, wp, bias, residual, mask_src, nullptr, y, nullptr, B, H, W, Cin, Cout, flags, leak, stream);
}

int64_t df_wino2d43_signbits_bytes(int64_t B, int64_t H, int64_t W, int64_t C) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 32) return 0;
  return B * ceil_div(H, 16) * ceil_div(W, 32) * (C / 32) * kT * 4;
}

int df_wino2d43_conv_bits(const float* x, const float* wp, const float* bias, const void* mask_bits, float* y, void* sign_bits, int64_t B, int64_t H,
                          int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream) {
  DF_REQUIRE((sign_bits != nullptr) != (mask_bits != nullptr), DF_EINVAL, "df_wino2d43_conv_bits: exactly one of sign_bits (forward) / mask_bits (dgrad)");
  return w2_conv("df_wino2d43_conv_bits", x, wp, bias, nullptr, nullptr, mask_bits, y, sign_bits, B, H, W, Cin, Cout, flags, leak, stream);
}

int df_wino2d43_conv_addup_bits(const float* x, const float* wp, const float* bias, const float* xc, float* y2, void* sign_bits, int64_t B, int64_t H,
                                int64_t W, int64_t Cin, int64_t Cout, float leak, df_stream_t stream) {
  const char* fn = "df_wino2d43_conv_addup_bits";
  DF_REQUIRE(x && wp && bias && xc && y2 && sign_bits, DF_EINVAL, "%s: null pointer", fn);
  DF_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, DF_EINVAL, "%s: extents must be positive and even (the output of a 2x up-sampling block)", fn);
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "%s: Cin, Cout must be multiples of 32", fn);
  DF_REQUIRE(H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE, "%s: one image must stay below 2 GiB", fn);
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x) && df::aligned16(xc) && df::aligned16(y2) && df::aligned16(sign_bits), DF_EALIGN,
             "%s: x, xc, y2, the bit words and the packed weights must be 16-byte aligned", fn);
  W2Args a;
  a.x = x; a.wp = wp; a.bias = bias; a.residual = xc; a.mask_src = nullptr; a.y = nullptr; a.y2 = y2;
  a.bits_out = static_cast<unsigned*>(sign_bits); a.bits_in = nullptr;
  a.B = (int)B; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nby = (int)ceil_div(H, 16); a.nbx = (int)ceil_div(W, 32);
  const int64_t ntb = B * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "%s: too many workgroups", fn);
  a.ntb = (int)ntb;
  a.flags = DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP; a.leak = leak;
  const int64_t grid = w2_grid(a, ntb);
  hipLaunchKernelGGL((wino2d43_kernel<DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP | kSignBits | kNoPrimary>), dim3((unsigned)grid), dim3(kT), 0,
                     df::as_stream(stream), a);
  return df::launched(fn);
}

int df_lrelu_words2d_bwd_pool2x(const float* gy, const void* mask_bits, float* gx, float* gpool, float leak, int64_t B, int64_t Hc, int64_t Wc, int64_t C,
                                df_stream_t stream) {
  const char* fn = "df_lrelu_words2d_bwd_pool2x";
  DF_REQUIRE(gy && mask_bits && gx && gpool, DF_EINVAL, "%s: null pointer", fn);
  DF_REQUIRE(B > 0 && Hc > 0 && Wc > 0, DF_EINVAL, "%s: non-positive extent", fn);
  DF_REQUIRE(C > 0 && C % 32 == 0, DF_ESHAPE, "%s: C must be a multiple of 32 (the sign words' cout slices)", fn);
  DF_REQUIRE(df::aligned16(gy) && df::aligned16(gx) && df::aligned16(gpool) && df::aligned16(mask_bits), DF_EALIGN, "%s: 16-byte alignment", fn);
  const int64_t H = 2 * Hc, W = 2 * Wc, nby = ceil_div(H, 16), nbx = ceil_div(W, 32), ncs = C / 32;
  DF_REQUIRE(B * nby * nbx * ncs < (1LL << 31) && H * W * C <= (1LL << 29), DF_ESHAPE, "%s: tensor too large", fn);
  hipLaunchKernelGGL(lrelu_words2d_bwd_pool_kernel, dim3((unsigned)(B * nby * nbx * ncs)), dim3(kT), 0, df::as_stream(stream), gy,
                     static_cast<const unsigned*>(mask_bits), gx, gpool, leak, (int)H, (int)W, (int)C, (int)nby, (int)nbx, (int)ncs);
  return df::launched(fn);
}

}  // extern "C"
