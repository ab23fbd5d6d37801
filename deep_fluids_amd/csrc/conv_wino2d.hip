// 3x3 SAME stride-1 convolution (2-D) as Winograd F(2x2, 3x3) on the fp32 matrix cores
// (reference: slim.conv2d behind ops.py:12-13, called from model.py:26,42 -- the 128->128 layers of GeneratorBE).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A      per 2x2 output tile / 4x4 input tile, 16 transform points xi:
// 4 MFMA MACs per output pixel and (cin,cout) pair instead of 9 (2.25x fewer matrix FLOPs); all arithmetic fp32.
//
// Same building blocks as the 3-D kernel (conv_wino.hip), with a simpler decomposition because all 16 transform points fit in
// one wave's accumulators:
//   * persistent workgroup = 8 waves = a block of 8 x 16 tiles (16 x 32 output pixels) x 32 output channels; wave w owns tile
//     row w (16 tiles) and ALL 16 transform points: 32 MFMA 16x16x4 per k-step (4 input channels), 128 accumulators; the inverse
//     transform is done in registers by the wave itself (no exchange through LDS, no barrier in the epilogue).
//   * input: the 18 x 34 halo block of a 16-channel chunk in LDS, channel-major [c][y*34 + x] (a wave's ds_read2_b64 touch 16
//     consecutive even dword pairs: conflict-free); SAME padding = buffer-load range check.
//   * weights: the 8 waves use the SAME transformed weights (same xi, same cout slice), so a chunk's 32 KB go through LDS once per
//     workgroup (4 loads + 4 ds_write_b128 per thread and chunk) and reach the MFMA operand registers by ds_read_b128.  All
//     vector-memory loads of a chunk (5 input pieces + 4 weight pieces per thread) are therefore waited for ONCE per chunk -- a
//     whole chunk of latency tolerance, unlike the 3-D kernel whose per-k-step weight waits also wait for its staging loads.
//   * both LDS areas double buffered: one barrier per chunk.
#include "df_common.hpp"
#include "conv_args.hpp"

namespace {

using df::ceil_div;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 512;
constexpr int kPackT = 256;
constexpr int CKW = 16;                  // input channels per chunk (4 k-steps)
constexpr int HY = 18, HX = 34;          // halo block (16 + 2) x (32 + 2)
constexpr int PY = 34;
constexpr int CP = HY * PY + 2;          // 614 dwords per channel plane (+2: staging writes spread over the banks)
constexpr int HV = HY * HX;              // 612 staged pixels
constexpr int NLOAD = 5;                 // ceil(612 * 4 float4 pieces / 512 threads)
constexpr int BUF = CKW * CP;            // dwords per input buffer
constexpr int WBUF = 4 * 2048;           // dwords per weight buffer: 4 k-steps x [nb 2][xi_y 4][lane 64][xi_x 4]

struct Wino2dArgs {
  const float* x;
  const f32x4* wp;
  const float* bias;
  const float* residual;
  const float* mask_src;
  float* y;
  int B, H, W, Cin, Cout;
  int nby, nbx, ntb, ncs, spx;
  int flags;
  float leak;
};

// ---- weight transform + packing: Up[cs][k4][nb][xy][kq][j][xx] = sum_taps G[xy][ty] G[xx][tx] g[tap][4 k4 + kq][32 cs + 16 nb + j]
// mode 0: g[tap][k][n] = w[tap][k][n];  mode 1: g[tap][k][n] = w[8 - tap][n][k]  (dgrad operand)
// [r6] thread = one (k, n) filter: its 9 taps are read once and all 16 transform points come out of the separable G transform in fp64, written as
// four float4 (the four xi_x of a xi_y) -- 5 us per 128 x 128 layer.  (Rounds 2-5: one thread per OUTPUT element re-reading the taps, 18 us.)
__global__ __launch_bounds__(kPackT) void wino2d_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout,
                                                             int mode, int64_t total) {
  (void)total;
  const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int64_t nfil = static_cast<int64_t>(K) * N;
  for (int64_t f = static_cast<int64_t>(blockIdx.x) * kPackT + threadIdx.x; f < nfil; f += static_cast<int64_t>(gridDim.x) * kPackT) {
    const int n = static_cast<int>(f % N), k = static_cast<int>(f / N);
    double g[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      g[tap] = static_cast<double>(mode == 0 ? w[(static_cast<int64_t>(tap) * cin + k) * cout + n]
                                             : w[(static_cast<int64_t>(8 - tap) * cin + n) * cout + k]);
    double gx[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double a = g[r * 3], b = g[r * 3 + 1], c = g[r * 3 + 2];
      gx[r][0] = a; gx[r][1] = 0.5 * (a + b + c); gx[r][2] = 0.5 * (a - b + c); gx[r][3] = c;
    }
    const int cs = n >> 5, nb = (n >> 4) & 1, j = n & 15, k4 = k >> 2, kq = k & 3;
#pragma unroll
    for (int xy = 0; xy < 4; ++xy) {
      f32x4 o;
#pragma unroll
      for (int xx = 0; xx < 4; ++xx) {
        const double a = gx[0][xx], b = gx[1][xx], c = gx[2][xx];
        o[xx] = static_cast<float>(xy == 0 ? a : xy == 3 ? c : xy == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c));
      }
      const int64_t idx = (((((static_cast<int64_t>(cs) * (K / 4) + k4) * 2 + nb) * 4 + xy) * 4 + kq) * 16 + j) * 4;
      *reinterpret_cast<f32x4*>(wp + idx) = o;
    }
  }
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
__device__ __forceinline__ f32x2 pk_bt01(f32x2 p, f32x2 q) {      // (x0 - x2, x1 + x2)
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}
__device__ __forceinline__ f32x2 pk_bt23(f32x2 p, f32x2 q) {      // (x2 - x1, x1 - x3)
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

struct Blk { const float* xb; int hoff, b, y0, x0; };

// MODE 1 (UP): x is the COARSE image of an up-sampling-aware conv (fine pixel p reads xc[p >> 1]; a.H, a.W are the fine extents): the
// transform points with index 2 vanish for the duplicated input, 9 of the 16 products remain.  MODE 2 (POOL): the adjoint -- the output
// is the 2x2 sum-pool of the convolution, accumulated into the coarse image y; the pooled inverse transform is (1, 2, 0, -1) per axis.
template <int FL, int MODE = 0>
__global__ __launch_bounds__(kT, 1) void wino2d_kernel(const Wino2dArgs a) {
  constexpr bool UP = MODE == 1, POOL = MODE == 2, P9 = MODE != 0;
  __shared__ __attribute__((aligned(16))) float sIn[2 * BUF];
  __shared__ __attribute__((aligned(16))) float sW[2 * WBUF];
  const int eflags = FL >= 0 ? FL : a.flags;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // tile row of the block
  const int tl = lane & 15, kq = lane >> 4;                        // tile in the row; cin % 4

  // persistent worker -> (cout slice, sequence of tile blocks), as in conv_wino.hip
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
  const int tb0 = tb;
  const int niter = (a.ntb - tb0 + tstride - 1) / tstride;
  auto decode = [&](int t) -> Blk {
    Blk bi;
    const int bx = t % a.nbx;
    const int t2 = t / a.nbx;
    const int by = t2 % a.nby;
    bi.b = t2 / a.nby;
    bi.y0 = by * 16; bi.x0 = bx * 32;
    bi.xb = a.x + static_cast<int64_t>(bi.b) * (UP ? (a.H >> 1) * (a.W >> 1) : a.H * a.W) * a.Cin;
    bi.hoff = ((bi.y0 - 1) * a.W + (bi.x0 - 1)) * a.Cin;
    return bi;
  };

  // ---- staging plan: 5 float4 pieces of the 612-pixel x 16-channel halo block per thread; 4 float4 of the chunk's weights ----
  int ldst[NLOAD];
  unsigned so[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    int p = it * kT + tid;
    if (p > HV * 4 - 1) p = HV * 4 - 1;
    const int hv = p >> 2, q4 = p & 3;
    ldst[it] = ((q4 * 4) * CP + (hv / HX) * PY + hv % HX) * 4;
  }
  const unsigned img_bytes = static_cast<unsigned>(UP ? (a.H >> 1) * (a.W >> 1) : a.H * a.W) * a.Cin * 4u;
  auto set_offs = [&](const Blk& bi) {
    // [r6] opaque thread id: LLVM otherwise hoists the five pieces' (hy, hx) decomposition out of the block loop and SPILLS it (39-72 spilled VGPRs
    // in every instantiation; the reloads sat in the main loop behind `s_waitcnt vmcnt(0)`) -- see conv_wino.hip / profiles/r06_probes.md
    int tido = tid;
    asm volatile("" : "+v"(tido));
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      int p = it * kT + tido;
      if (p > HV * 4 - 1) p = HV * 4 - 1;
      const int hv = p >> 2, q4 = p & 3;
      const int hx = hv % HX, hy = hv / HX;
      const int gy = bi.y0 - 1 + hy, gx = bi.x0 - 1 + hx;
      const bool ok = static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) && static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
      if (UP) so[it] = ok ? static_cast<unsigned>(((gy >> 1) * (a.W >> 1) + (gx >> 1)) * a.Cin + q4 * 4) * 4u : 0x80000000u;
      else so[it] = ok ? static_cast<unsigned>(bi.hoff + (hy * a.W + hx) * a.Cin + q4 * 4) * 4u : 0x80000000u;
    }
  };
  char* sInB = reinterpret_cast<char*>(sIn);
  auto stage_store = [&](int it, int bufbytes, const f32x4& v) {
    float* d = reinterpret_cast<float*>(sInB + (ldst[it] + bufbytes));
    d[0] = v[0]; d[CP] = v[1]; d[2 * CP] = v[2]; d[3 * CP] = v[3];
  };
  const int nk4 = a.Cin >> 2, nchunk = a.Cin / CKW;
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, static_cast<unsigned>(a.Cin) * a.Cout * 64u);
  const unsigned wslice = static_cast<unsigned>(cs * nk4) * 8192u;          // bytes: this slice's [k4][2048 floats]
  f32x4* sW4 = reinterpret_cast<f32x4*>(sW);
  // chunk weights global -> LDS without passing through registers (gfx950 16-byte LDS DMA): wave w fills float4s
  // [buf*2048 + i*512 + w*64 + lane], i = 0..3, i.e. the packed layout verbatim
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto load_weights = [&](int wbuf, unsigned soff) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_ptr dst = (lds_ptr)(sW4 + wbuf * (WBUF / 4) + i * kT + wave * 64);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, dst, 16, static_cast<unsigned>(i * kT + tid) * 16u, soff, 0, 0);
    }
  };

  // ---- A operand: lane = (tile tl of row `wave`, cin kq) -------------------------------------------------------------------
  const int offb = (kq * CP + (2 * wave) * PY + 2 * tl) * 4;       // bytes (multiple of 8)
  f32x2 raw[8];            // [y][x pair] of the 4x4 patch
  f32x2 A2[8];             // A2[xi_y*2 + h] = (xi_x = 2h, 2h+1)
  auto raw_read = [&](int idxbytes) {
    int ia = idxbytes + offb;
    asm volatile("" : "+v"(ia));
    __builtin_assume((ia & 7) == 0);
#pragma unroll
    for (int yy = 0; yy < 4; ++yy) {
      raw[yy * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ia + (yy * PY) * 4);
      raw[yy * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ia + (yy * PY + 2) * 4);
    }
  };
  auto transform = [&]() {
    f32x2 U[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                  // y: rows (0,1,2,3) -> xi_y
      U[0 + h] = pk_sub(raw[0 + h], raw[4 + h]);
      U[2 + h] = pk_add(raw[2 + h], raw[4 + h]);
      U[4 + h] = pk_sub(raw[4 + h], raw[2 + h]);
      U[6 + h] = pk_sub(raw[2 + h], raw[6 + h]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                  // x
      A2[k * 2 + 0] = pk_bt01(U[k * 2], U[k * 2 + 1]);
      A2[k * 2 + 1] = pk_bt23(U[k * 2], U[k * 2 + 1]);
    }
  };

  f32x4 acc[2][16];
  f32x4 bq[2][4];
  auto read_b = [&](int wbuf, int ks) {      // this k-step's weights: LDS -> MFMA operand registers
    const f32x4* p = sW4 + wbuf * (WBUF / 4) + ks * 512 + lane;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[nb][q] = p[(nb * 4 + q) * 64];
  };

  // ---- prologue: first block's chunk 0 (input + weights) -> buffers 0 ------------------------------------------------------------
  Blk cur = decode(tb0);
  set_offs(cur);
  {
    const __amdgpu_buffer_rsrc_t srd0 = make_srd(cur.xb, img_bytes);
    f32x4 stg[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) stg[it] = buf_load16(srd0, so[it], 0u);
    load_weights(0, wslice);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) stage_store(it, 0, stg[it]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int pb = 0;
  for (int itb = 0; itb < niter; ++itb) {
    const Blk nxt = decode(tb0 + (itb + 1 < niter ? itb + 1 : itb) * tstride);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    raw_read(pb * BUF * 4);

    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int par = (chunk + pb) & 1;
      const int bo = par * BUF * 4, bn = BUF * 4 - bo;
      const bool lastc = chunk + 1 == nchunk;
      if (lastc) set_offs(nxt);
      const __amdgpu_buffer_rsrc_t ssrd = make_srd(lastc ? nxt.xb : cur.xb, img_bytes);
      const unsigned schunk = lastc ? 0u : static_cast<unsigned>(chunk + 1) * (CKW * 4u);
      const unsigned wchunk = wslice + static_cast<unsigned>(lastc ? 0 : chunk + 1) * 32768u;      // next chunk's 4 k-steps of weights
      f32x4 stg[NLOAD];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        transform();
        __builtin_amdgcn_sched_barrier(0);
        read_b(par, ks);
        if (ks == 0) {          // next chunk's input pieces and weights: consumed three k-steps from now
#pragma unroll
          for (int it = 0; it < NLOAD; ++it) stg[it] = buf_load16(ssrd, so[it], schunk);
          load_weights(par ^ 1, wchunk);      // (that buffer was last read at ks = 3 of the previous chunk, before its barrier)
        }
        if (ks == 3) {
#pragma unroll
          for (int it = 0; it < NLOAD; ++it) stage_store(it, bn, stg[it]);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA weight loads have landed
          __syncthreads();      // next chunk staged by everyone (its buffers were last read one chunk ago)
        }
        raw_read(ks < 3 ? bo + (ks + 1) * 16 * CP : bn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (!P9 || ((i >> 2) != 2 && (i & 3) != 2))
              acc[nb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[i >> 1][i & 1], bq[nb][i >> 2][i & 3], acc[nb][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: inverse transform in registers; accumulator element e of lane (kq, tl) = tile 4 kq + e, cout = tl (+16 nb) ---
    if constexpr (POOL) {
      const int Hc = a.H >> 1, Wc = a.W >> 1;
      const int cy = (cur.y0 >> 1) + wave;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cx = (cur.x0 >> 1) + 4 * kq + e;
          float py[4];
#pragma unroll
          for (int yy = 0; yy < 4; ++yy)
            py[yy] = yy == 2 ? 0.f : acc[nb][yy * 4 + 0][e] + 2.f * acc[nb][yy * 4 + 1][e] - acc[nb][yy * 4 + 3][e];
          if (cy < Hc && cx < Wc) {
            float* yo = a.y + ((static_cast<int64_t>(cur.b) * Hc + cy) * Wc + cx) * a.Cout + n0 + nb * 16 + tl;
            *yo += py[0] + 2.f * py[1] - py[3];
          }
        }
    } else {
      const int oy0 = cur.y0 + 2 * wave;
      const int64_t sW_ = a.Cout, sH_ = static_cast<int64_t>(a.W) * a.Cout;
      const bool full = cur.y0 + 16 <= a.H && cur.x0 + 32 <= a.W;      // workgroup-uniform
      const int64_t obase0 = ((static_cast<int64_t>(cur.b) * a.H + oy0) * a.W + cur.x0 + 8 * kq) * a.Cout + n0 + tl;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int col = n0 + nb * 16 + tl;
        const float bv = (eflags & DF_CONV_BIAS) ? a.bias[col] : 0.f;
        float rres[16], rmask[16];
        if (full && (eflags & (DF_CONV_RESIDUAL | DF_CONV_MASK))) {      // all 16 outputs of this lane: loads issued together
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int64_t o = obase0 + nb * 16 + (2 * e + (s & 1)) * sW_ + (s >> 1) * sH_;
              rres[e * 4 + s] = (eflags & DF_CONV_RESIDUAL) ? a.residual[o] : 0.f;
              rmask[e * 4 + s] = (eflags & DF_CONV_MASK) ? a.mask_src[o] : 1.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f32x2 px[4];
#pragma unroll
          for (int yy = 0; yy < 4; ++yy) {
            if (P9) {      // the xi = 2 points were never multiplied
              px[yy][0] = acc[nb][yy * 4 + 0][e] + acc[nb][yy * 4 + 1][e];
              px[yy][1] = acc[nb][yy * 4 + 1][e] - acc[nb][yy * 4 + 3][e];
            } else {
              px[yy][0] = acc[nb][yy * 4 + 0][e] + acc[nb][yy * 4 + 1][e] + acc[nb][yy * 4 + 2][e];
              px[yy][1] = acc[nb][yy * 4 + 1][e] - acc[nb][yy * 4 + 2][e] - acc[nb][yy * 4 + 3][e];
            }
          }
          const f32x2 o01 = P9 ? px[0] + px[1] : px[0] + px[1] + px[2], o23 = P9 ? px[1] - px[3] : px[1] - px[2] - px[3];
          const float ov[4] = {o01[0], o01[1], o23[0], o23[1]};
          const int ox0 = cur.x0 + 2 * (4 * kq + e);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int64_t o = obase0 + nb * 16 + (2 * e + (s & 1)) * sW_ + (s >> 1) * sH_;
            float v = ov[s] + bv;
            if (eflags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
            if (full) {
              if (eflags & DF_CONV_RESIDUAL) v += rres[e * 4 + s];
              if (eflags & DF_CONV_MASK) v = rmask[e * 4 + s] > 0.f ? v : a.leak * v;
              a.y[o] = v;
            } else if (oy0 + (s >> 1) < a.H && ox0 + (s & 1) < a.W) {
              if (eflags & DF_CONV_RESIDUAL) v += a.residual[o];
              if (eflags & DF_CONV_MASK) v = a.mask_src[o] > 0.f ? v : a.leak * v;
              a.y[o] = v;
            }
          }
        }
      }
    }
    pb = (pb + nchunk) & 1;
    cur = nxt;
  }
}

int g_wino2d_spx = 0;

}  // namespace

extern "C" {

int64_t df_wino2d_packed_elems(int64_t cin, int64_t cout, int mode) {
  (void)mode;
  return 16 * cin * cout;
}

int df_wino2d_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_wino2d_pack_weights: null pointer");
  DF_REQUIRE(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0 && (mode == 0 || mode == 1), DF_ESHAPE,
             "df_wino2d_pack_weights: cin, cout must be multiples of 32; mode 0|1");
  const int64_t total = 16 * cin * cout;
  int64_t g = ceil_div(cin * cout, kPackT);      // one thread per (k, n) filter
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(wino2d_pack_kernel, dim3((unsigned)g), dim3(kPackT), 0, df::as_stream(stream), w, wp, (int)cin, (int)cout, mode,
                     total);
  return df::launched("df_wino2d_pack_weights");
}

static int64_t wino2d_grid(Wino2dArgs& a, int64_t ntb) {
  int64_t grid = df::kCUs;
  a.spx = 1;
  if (8 % a.ncs == 0) {
    a.spx = (g_wino2d_spx > 0 && a.ncs % g_wino2d_spx == 0) ? g_wino2d_spx : (a.ncs % 2 == 0 ? 2 : 1);
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

int df_wino2d_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                       float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak,
                       df_stream_t stream) {
  DF_REQUIRE(x && wp && y, DF_EINVAL, "df_wino2d_conv_fwd: null pointer");
  DF_REQUIRE(B > 0 && H > 0 && W > 0, DF_EINVAL, "df_wino2d_conv_fwd: non-positive extent");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE,
             "df_wino2d_conv_fwd: Cin, Cout must be multiples of 32 (use df_conv_fwd otherwise)");
  DF_REQUIRE(H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino2d_conv_fwd: one image must stay below 2 GiB (use df_conv_fwd)");
  DF_REQUIRE(!(flags & DF_CONV_BIAS) || bias, DF_EINVAL, "df_wino2d_conv_fwd: DF_CONV_BIAS without bias");
  DF_REQUIRE(!(flags & DF_CONV_RESIDUAL) || residual, DF_EINVAL, "df_wino2d_conv_fwd: DF_CONV_RESIDUAL without residual");
  DF_REQUIRE(!(flags & DF_CONV_MASK) || mask_src, DF_EINVAL, "df_wino2d_conv_fwd: DF_CONV_MASK without mask_src");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x), DF_EALIGN, "df_wino2d_conv_fwd: x and packed weights must be 16-byte aligned");
  Wino2dArgs a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp);
  a.bias = bias; a.residual = residual; a.mask_src = mask_src; a.y = y;
  a.B = (int)B; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nby = (int)ceil_div(H, 16); a.nbx = (int)ceil_div(W, 32);
  const int64_t ntb = B * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino2d_conv_fwd: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = flags; a.leak = leak;
  const int64_t grid = wino2d_grid(a, ntb);
  const dim3 g((unsigned)grid), b(kT);
  hipStream_t s = df::as_stream(stream);
  if (flags == (DF_CONV_BIAS | DF_CONV_LRELU)) hipLaunchKernelGGL((wino2d_kernel<DF_CONV_BIAS | DF_CONV_LRELU>), g, b, 0, s, a);
  else if (flags == DF_CONV_MASK) hipLaunchKernelGGL((wino2d_kernel<DF_CONV_MASK>), g, b, 0, s, a);
  else if (flags == DF_CONV_RESIDUAL) hipLaunchKernelGGL((wino2d_kernel<DF_CONV_RESIDUAL>), g, b, 0, s, a);
  else hipLaunchKernelGGL((wino2d_kernel<-1>), g, b, 0, s, a);
  return df::launched("df_wino2d_conv_fwd");
}

static int wino2d_up_common(const char* fn, Wino2dArgs& a, const float* wp, int64_t B, int64_t Hc, int64_t Wc, int64_t K, int64_t N) {
  DF_REQUIRE(B > 0 && Hc > 0 && Wc > 0, DF_EINVAL, "%s: non-positive extent", fn);
  DF_REQUIRE(K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0, DF_ESHAPE, "%s: channel counts must be multiples of 32", fn);
  DF_REQUIRE(4 * Hc * Wc * (K > N ? K : N) <= (1LL << 29) && K * N <= (1LL << 24), DF_ESHAPE, "%s: one image must stay below 2 GiB", fn);
  DF_REQUIRE(df::aligned16(wp), DF_EALIGN, "%s: packed weights must be 16-byte aligned", fn);
  a.wp = reinterpret_cast<const f32x4*>(wp);
  a.B = (int)B; a.H = (int)(2 * Hc); a.W = (int)(2 * Wc); a.Cin = (int)K; a.Cout = (int)N;      // fine extents
  a.nby = (int)ceil_div(a.H, 16); a.nbx = (int)ceil_div(a.W, 32);
  const int64_t ntb = B * a.nby * a.nbx;
  a.ncs = (int)(N / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "%s: too many workgroups", fn);
  a.ntb = (int)ntb;
  return DF_OK;
}

int df_wino2d_upconv_fwd(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Hc, int64_t Wc, int64_t Cin,
                         int64_t Cout, int flags, float leak, df_stream_t stream) {
  DF_REQUIRE(xc && wp && y && bias, DF_EINVAL, "df_wino2d_upconv_fwd: null pointer");
  DF_REQUIRE(flags == (DF_CONV_BIAS | DF_CONV_LRELU), DF_EINVAL, "df_wino2d_upconv_fwd: flags must be DF_CONV_BIAS | DF_CONV_LRELU");
  DF_REQUIRE(df::aligned16(xc), DF_EALIGN, "df_wino2d_upconv_fwd: xc must be 16-byte aligned");
  Wino2dArgs a;
  if (int rc = wino2d_up_common("df_wino2d_upconv_fwd", a, wp, B, Hc, Wc, Cin, Cout)) return rc;
  a.x = xc; a.bias = bias; a.residual = nullptr; a.mask_src = nullptr; a.y = y;
  a.flags = flags; a.leak = leak;
  const int64_t grid = wino2d_grid(a, a.ntb);
  hipLaunchKernelGGL((wino2d_kernel<DF_CONV_BIAS | DF_CONV_LRELU, 1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  return df::launched("df_wino2d_upconv_fwd");
}

int df_wino2d_upconv_dgrad(const float* g, const float* wp, float* acc, int64_t B, int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout,
                           df_stream_t stream) {
  DF_REQUIRE(g && wp && acc, DF_EINVAL, "df_wino2d_upconv_dgrad: null pointer");
  DF_REQUIRE(df::aligned16(g), DF_EALIGN, "df_wino2d_upconv_dgrad: g must be 16-byte aligned");
  Wino2dArgs a;      // the adjoint conv reads g (Cout channels, fine grid) and produces Cin channels on the coarse grid
  if (int rc = wino2d_up_common("df_wino2d_upconv_dgrad", a, wp, B, Hc, Wc, Cout, Cin)) return rc;
  a.x = g; a.bias = nullptr; a.residual = nullptr; a.mask_src = nullptr; a.y = acc;
  a.flags = 0; a.leak = 0.f;
  const int64_t grid = wino2d_grid(a, a.ntb);
  hipLaunchKernelGGL((wino2d_kernel<0, 2>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  return df::launched("df_wino2d_upconv_dgrad");
}

}  // extern "C"
