// Pooled adjoint of the up-sampling-aware first conv of a generator block (d/d(xc) of conv3d(upscale3(xc, 2)), model.py:76-82 / ops.py:79-91):
//   acc[B, D/2, H/2, W/2, N] += 2x2x2 sum-pool of the 3x3x3 SAME convolution of the fine gradient g[B, D, H, W, K] with the dgrad operand,
// as the 27-of-64-point form of Winograd F(2,3)^3 (the pooled inverse transform (1, 2, 0, -1) per axis never reads the points with index 2).
//
// Round 6 re-write of conv_wino.hip's MODE 2 in the style of conv_wino43.hip: the z part of the input transform runs in the STAGING threads --
// thread = (halo row, column, channel quad) loads its z column and writes the (tile z-row, xi_z in {0, 1, 3}) planes -- so a lane's operand read
// is ONE plane (8 ds_read_b64 per k-step instead of 16), its transform 12 packed ops instead of 20, and the staging plan two registers instead
// of ten (no spills).  Same packed operand (df_wino_pack_weights, mode 1), same wave roles (waves 0-3: xi_z in {0, 1} x tile z-row with both cout
// blocks, waves 4-7: the two xi_z = 3 roles split by cout block: 27 MFMAs per SIMD and k-step), same summation order per output as MODE 2:
// bit-identical results (tests/test_gpu_layers.py::test_wino_upconv_dgrad_vs_oracle, test_gpu_wino_families.py).
#include <type_traits>
#include "df_common.hpp"
#include "conv_args.hpp"

namespace {

using df::ceil_div;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;
constexpr int CKW = 16;
constexpr int PY = 12, PP = 120;      // the LDS layout of conv_wino43.hip: row 12, plane (tile z-row, xi_z) 120, channel 962 dwords
#ifndef W27_CP
#define W27_CP (8 * PP + 2)
#endif
constexpr int CP = W27_CP;
constexpr int NCOL = 100 * 4;
constexpr int BUFF = CKW * CP;

struct W27Args {
  const float* x;      // the fine gradient [B, D, H, W, Cin]
  const float* wp;     // df_wino_pack_weights(mode 1)
  float* y;            // the coarse tensor [B, D/2, H/2, W/2, Cout], accumulated into
  int B, D, H, W, Cin, Cout;
  int nbz, nby, nbx, ntb, ncs;
  int spx;
};

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


// (x stage of F(2,3) on a row (d0 d1 | d2 d3):  (d0 - d2, d1 + d2)  and  (d2 - d1, d1 - d3))
__device__ __forceinline__ f32x2 pk_bt01(f32x2 p, f32x2 q) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}
__device__ __forceinline__ f32x2 pk_bt23(f32x2 p, f32x2 q) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}

__global__ __launch_bounds__(kT, 1) void wino27_pool_kernel(const W27Args a) {
  __shared__ __attribute__((aligned(16))) float sIn[2 * BUFF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xz = wave & 3, th = wave >> 2;      // combine role: coarse x offset, tile z-row
  const int tl = lane & 15, kq = lane >> 4;

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
    bi.hoff = 0;
    return bi;
  };

  // ---- staging: thread t < 400 = (halo row, column, channel quad) owns the z column; planes (tile z-row, xi_z in {0, 1, 3}) written ------
  const int scol = tid < NCOL ? tid : NCOL - 1;
  const int sq4 = scol & 3, shy = (scol >> 2) / 10, shx = (scol >> 2) % 10;
  const int ldst = ((sq4 * 4) * CP + shy * PY + shx) * 4;
  const bool stager = wave < 7;
  const unsigned vol_bytes = static_cast<unsigned>(a.D * a.H * a.W) * a.Cin * 4u;
  unsigned so;
  auto set_offs = [&](const BlockInfo& bi) {
    const int gy = bi.y0 - 1 + shy, gx = bi.x0 - 1 + shx;
    const bool ok = tid < NCOL && static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) && static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
    so = ok ? static_cast<unsigned>((gy * a.W + gx) * a.Cin + sq4 * 4) * 4u : 0x80000000u;
  };
  char* sInB = reinterpret_cast<char*>(sIn);
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
  // z part of B^T, the operations of conv_wino.hip's lanes (T = ra + qs * rb as one fma with qs = +-1: exactly the sum / difference):
  //   xi_z 0: d0 - d2;  1: d1 + d2;  3: d1 - d3   per tile z-row (plane 2 is never read by the 27-point form: not written)
  auto stage_store = [&](int bufbytes, const f32x4 (&v)[6]) {
    if (tid < NCOL) {
      char* d = sInB + (ldst + bufbytes);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 p0 = v[2 * t] - v[2 * t + 2], p1 = v[2 * t + 1] + v[2 * t + 2], p3 = v[2 * t + 1] - v[2 * t + 3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float* dc = reinterpret_cast<float*>(d + c * CP * 4);
          dc[(t * 4 + 0) * PP] = p0[c]; dc[(t * 4 + 1) * PP] = p1[c]; dc[(t * 4 + 3) * PP] = p3[c];
        }
      }
    }
  };

  // ---- main-loop roles as MODE 2: waves 0-3 = (xi_z in {0, 1}) x (tile z-row), both cout blocks; waves 4-7 = xi_z 3, z-row, ONE cout block ------
  const bool half = wave >= 4;
  const int mz = wave < 4 ? (wave >> 1) : 3;
  const int mth = wave < 4 ? (wave & 1) : ((wave >> 1) & 1);
  const int hnb = half ? (wave & 1) : 0;
  const int tx = tl & 3, ty = tl >> 2;
  const int offAb = (kq * CP + (mth * 4 + mz) * PP + (2 * ty) * PY + 2 * tx) * 4;
  f32x2 ra[8];             // [row][x pair]
  f32x2 A2[8];             // A2[xi_y * 2 + h] = (xi_x = 2h, 2h + 1)
  auto raw_read = [&](int idxbytes) {
    int ia = idxbytes + offAb;
    asm volatile("" : "+v"(ia));
    __builtin_assume((ia & 7) == 0);
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      ra[y * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ia + (y * PY) * 4);
      ra[y * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ia + (y * PY + 2) * 4);
    }
  };
  auto transform = [&]() {
    f32x2 U[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                           // y (xi_y = 2 is never multiplied)
      U[0 + h] = pk_sub(ra[0 + h], ra[4 + h]);
      U[2 + h] = pk_add(ra[2 + h], ra[4 + h]);
      U[6 + h] = pk_sub(ra[2 + h], ra[6 + h]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                           // x
      if (k == 2) continue;
      A2[k * 2 + 0] = pk_bt01(U[k * 2], U[k * 2 + 1]);
      A2[k * 2 + 1] = pk_bt23(U[k * 2], U[k * 2 + 1]);
    }
  };

  // ---- B operand: the F(2,3)^3 pack [cs][xz][k4][nb][xy][kq][j][xx] -------------------------------------------------------------------------
  const int nk4 = a.Cin >> 2;
  f32x4 bq[2][4];
  const unsigned laneb = static_cast<unsigned>(lane) * 16u;
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, static_cast<unsigned>(a.Cin) * a.Cout * 256u);
  const unsigned wbase_b = static_cast<unsigned>((cs * 4 + mz) * nk4) * 8192u + static_cast<unsigned>(hnb) * 4096u;
  auto issue_b = [&](int nb, int k4) {
    const int kl = k4 < nk4 ? k4 : 0;
    const unsigned sb = wbase_b + static_cast<unsigned>(kl) * 8192u + nb * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (q != 2) bq[nb][q] = buf_load16(wsrd, laneb + q * 1024u, sb);
  };

  f32x4 acc[2][16];
  const int nchunk = a.Cin / CKW;

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
      for (int i = 0; i < 16; ++i) acc[nb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    raw_read(pb * BUFF * 4);
    if (itb == 0) {
      issue_b(0, 0);
      if (!half) issue_b(1, 0);
    }

    {
      // (ONE copy of the loop with a wave-uniform branch around the second cout block: as a generic lambda instantiated for both roles the
      //  block descriptors passed through memory, their buffer resources lost wave-uniformity and every staging load became a waterfall loop)
      for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int bo = ((chunk + pb) & 1) * BUFF * 4, bn = BUFF * 4 - bo;
        const bool lastc = chunk + 1 == nchunk;
        if (lastc) set_offs(nxt);
        const unsigned schunk = static_cast<unsigned>(lastc ? 0 : chunk + 1) * (CKW * 4u);
        f32x4 stg[6];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          transform();
          __builtin_amdgcn_sched_barrier(0);
          if (ks == 2 && stager) stage_store(bn, stg);
          if (ks == 3) lds_barrier();
          raw_read(ks < 3 ? bo + (ks + 1) * 16 * CP : bn);
          __builtin_amdgcn_sched_barrier(0);
          const int k4n = chunk * 4 + ks + 1;
          const unsigned sbn = wbase_b + static_cast<unsigned>(k4n < nk4 ? k4n : 0) * 8192u;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            if (nb == 1 && half) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              if ((i >> 2) != 2 && (i & 3) != 2)
                acc[nb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[i >> 1][i & 1], bq[nb][i >> 2][i & 3], acc[nb][i], 0, 0, 0);
              if ((i & 3) == 3 && (i >> 2) != 2) {
                __builtin_amdgcn_sched_barrier(0);
                bq[nb][i >> 2] = buf_load16(wsrd, laneb + (i >> 2) * 1024u, sbn + nb * 4096u);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          if (ks == 0 && stager) stage_load(lastc ? nxt : cur, schunk, stg);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    // ---- epilogue: pooled inverse transform (1, 2, 0, -1) in y, x per accumulator element; z across the waves through the idle buffer ------
    {
      const int lb = ((nchunk - 1 + pb) & 1) * BUFF;
      float* sP = sIn + lb;      // [xi_z][tz][e][lane]
      const int Dc = a.D >> 1, Hc = a.H >> 1, Wc = a.W >> 1;
      const int cz = (cur.z0 >> 1) + th, cy = (cur.y0 >> 1) + kq, cx = (cur.x0 >> 1) + xz;
      const bool inb = cz < Dc && cy < Hc && cx < Wc;
      float* yo = a.y + (((static_cast<int64_t>(cur.b) * Dc + cz) * Hc + cy) * Wc + cx) * a.Cout + n0 + tl;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float prev = inb ? yo[nb * 16] : 0.f;
        auto emit = [&](const f32x4 (&c)[16]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float py[4];
#pragma unroll
            for (int yy = 0; yy < 4; ++yy)
              py[yy] = yy == 2 ? 0.f : c[yy * 4 + 0][e] + 2.f * c[yy * 4 + 1][e] - c[yy * 4 + 3][e];
            sP[((mz * 2 + mth) * 4 + e) * 64 + lane] = py[0] + 2.f * py[1] - py[3];
          }
        };
        if (!half) emit(acc[nb]);
        else if (nb == hnb) emit(acc[0]);
        lds_barrier();
        const float m0 = sP[((0 * 2 + th) * 4 + xz) * 64 + lane], m1 = sP[((1 * 2 + th) * 4 + xz) * 64 + lane];
        const float m3 = sP[((3 * 2 + th) * 4 + xz) * 64 + lane];
        if (inb) yo[nb * 16] = prev + (m0 + 2.f * m1 - m3);
        lds_barrier();
      }
    }
    pb = (pb + nchunk) & 1;
    cur = nxt;
  }
}

}  // namespace

namespace dfw27 {

// df_wino_upconv_dgrad's launch (conv_wino.hip validates the arguments): the F(2,3)^3 dgrad operand, accumulate into the coarse tensor
int launch_pool(const float* g, const float* wp, float* acc, int64_t B, int64_t D, int64_t H, int64_t W, int64_t K, int64_t N, hipStream_t s) {
  W27Args a;
  a.x = g; a.wp = wp; a.y = acc;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)K; a.Cout = (int)N;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(N / 32);
  a.ntb = (int)ntb;
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
  hipLaunchKernelGGL(wino27_pool_kernel, dim3((unsigned)grid), dim3(kT), 0, s, a);
  return df::launched("df_wino_upconv_dgrad");
}

}  // namespace dfw27
