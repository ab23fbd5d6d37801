// 3x3x3 SAME stride-1 convolution as Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores
// (reference: slim.conv3d behind ops.py:12-16, called from model.py:68,84 -- the 128->128 layers of GeneratorBE3).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        per 2x2x2 output tile / 4x4x4 input tile, 64 transform points xi
// = 64 independent GEMMs  M_xi[tile][cout] = sum_cin V_xi[tile][cin] * U_xi[cin][cout]:   8 MFMA MACs per output voxel
// and (cin,cout) pair instead of 27 -> 3.375x fewer matrix-core FLOPs than the direct kernel (conv.hip); all arithmetic
// stays fp32 (v_mfma_f32_16x16x4_f32 + exact +-1 / 0.5 transform factors), results differ from the direct sum only by
// fp32 rounding order (tests: same tolerance as the direct kernel against the fp64 oracle).
//
// Persistent workgroup = 8 waves (two per SIMD: measured on gfx950, a wave's own VALU / LDS / VMEM instructions do NOT
// overlap its own MFMAs -- a second wave on the SIMD is what keeps the matrix pipe busy) = one block of 2x4x4 tiles
// (4x8x8 output voxels) x 32 output channels at a time.  Wave (xi_z, tz) owns the 16 transform points with that xi_z
// for the 16 tiles of z-row tz: 32 MFMA 16x16x4 per k-step (4 input channels), 128 accumulator registers.
//   * input: the 6x10x10 halo block of a 16-channel chunk is staged in LDS channel-major ([c][z*144 + y*12 + x], the
//     pitches make every wave-wide ds_read2_b64 below bank-conflict free), double buffered, one barrier per chunk;
//     the next tile block's first chunk is staged during the last chunk of the current one.
//   * A operand: lane = (tile, cin%4) reads the 2 z-planes x 4 x 4 inputs of its tile that xi_z needs (8 ds_read2_b64),
//     runs the separable B^T transform in registers (24 packed-fp32 VALU ops) and so produces the 16 A values of one
//     k-step directly in MFMA A layout -- the transformed input never touches memory.
//   * B operand: transformed weights U are packed [cout/32][xi_z][cin/4][cout/16%2][xi_y][cin%4][cout%16][xi_x] so that
//     a lane needs 8 coalesced 16-byte global loads per k-step (L2 / L1 hits: the two tz waves read the same words).
//   * epilogue: inverse transform in y,x in registers, the four xi_z partial planes are combined through the idle LDS
//     buffer, then bias / lrelu / residual / lrelu-mask as in conv.hip.  The dgrad is the same kernel on mode-1 weights.
//
// Compile-time switches of wino3d_kernel.  The release library instantiates DBG = 0, PREC = 0, XS = 0 only (FL = the epilogue, MODE = 0 plain |
// 3 up-sampling-aware forward with the coarse halo block staged (round 5; MODE 1 = its round-2 form staging the fine positions, tuning library
// only) | 2 pooled adjoint); everything else exists in the -DDF_TUNING library for tools/r05_breakdown_probe.py, r05_upc_check.py and
// tools/archive/wino_diag.py, wino_probe.py, wino_xblk_probe.py:
//   DBG bits (diagnosis: results wrong by construction unless noted)   1 no input transform | 2 no raw LDS reads | 4 no staging | 8 no weight loads |
//     16 per-phase cycle counters (correct results) | 64 staging loads kept alive, no LDS writes | 128 staged zeros (LDS writes only) |
//     256 staging loads read an always-cached address | 384 ... confined to a 1 MB L2-resident window | 512 no output stores |
//     262144 / 524288 weight loads of the z-row-1 waves / of all waves through a zero-length descriptor |
//     1048576 / 2097152 (correct results) staging loads one per MFMA row during k-step 1 / 0 instead of one burst |
//     4194304 staging loads fully coalesced (1 KiB contiguous per wave instruction), still L1-missing
//   XS bits (round-4 probe, correct results unless noted): 1 LDS-DMA staging from the x-blocked copy | 2 waves 4-7 issue | 4 in front of k-step 0's
//     MFMAs | 8 unrolled issue | 16 at the start of k-step 0 | 32 two weight register sets | 64 "weights" from LDS (timing only) | 128 three buffers
//   PREC 1: bf16x3 in the Winograd domain (round 3).   The switches of rounds 1-3 whose experiments were negative (cache policies, chunk-pair
//   permutation, split staging, reload orders, full barriers, setprio) were removed in round 4; their numbers are in profiles/LAB_NOTES.md.
#include <type_traits>
#include "df_common.hpp"
#include "conv_args.hpp"

namespace {

using df::ceil_div;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;                   // 8 waves: (xi_z, tile z-row)
constexpr int kPackT = 256;
constexpr int CKW = 16;                   // input channels per LDS chunk (4 k-steps)
constexpr int PY = 12, PZ = 144;          // LDS pitches (dwords) of a channel plane: 6 z x 10 y x 10 x, padded
constexpr int CP = 866;                   // dwords per channel plane (6*144 = 864, +2: staging writes spread over banks)
constexpr int HY = 10, HX = 10, HV = 600; // halo block 6 x 10 x 10
constexpr int NLOAD = 5;                  // ceil(600 * 4 float4 pieces / 512 threads)
// MODE 3 (round 5): the up-sampling-aware forward stages the COARSE halo block itself -- 4 x 6 x 6 coarse voxels instead of the 6 x 10 x 10
// fine positions that repeat them (4.2x fewer staging loads) -- and builds the 27 live transform points from the 3 x 3 coarse patch per
// plane a tile sees.  A coarse row (pz, py) of a channel is stored as FOUR 16-byte windows, window t = (c[t], c[t+1], c[t+2], -) = the three
// columns tile column tx = t needs, so a lane fetches a patch row with ONE aligned ds_read_b128 (the 16 (ty, tx) lanes of a channel read 64
// consecutive dwords: conflict-free).  Staging is by WINDOW: thread = (window t of row (pz, py), channel quad) loads the three coarse voxels
// px = t, t+1, t+2 (16 bytes = its 4 channels each; neighbours' loads hit the same lines) and writes its four channels' windows with four
// ds_write_b128 (16 lanes = 4 quads x 4 windows cover 64 banks: conflict-free): 4 x 6 x 4 windows x 4 quads = 384 threads (waves 0-5).
// (First version: every thread staged one voxel and scattered it to its <= 3 windows with 12 ds_write_b32 -- all lanes of an instruction on
//  banks = k (mod 4), >= 4-way conflicts, 10 % of the kernel; profiles/r05_probes.md.)
constexpr int UPY = 16, UPZ = 6 * UPY, UCP = 4 * UPZ + 4;
constexpr int UWIN = 4 * 6 * 4;           // windows of the coarse halo block: 4 planes x 6 rows x 4 windows
constexpr int NLU = 2;                    // loads per staging thread: the window's outer voxels t and t + 2; the middle one comes from the
                                          // neighbouring window's thread by DPP (the 16 lanes of a DPP row = 4 quads x the 4 windows of a row)
// internal epilogue flags (beyond the public DF_CONV_*), part of the compile-time FL of the specialised instantiations:
//   kSignBits: also emit the sign pattern of the output, one byte per lane and cout block holding the signs of the lane's 8 outputs
//              (what a later masked dgrad of the same geometry needs of it: 1/32 of the activation's bytes);
//   kMaskBits: DF_CONV_MASK reads those bytes (2 byte loads per lane and tile block) instead of 16 fp32 activations per lane.
constexpr int kSignBits = 64, kMaskBits = 128;
constexpr int kNoPrimary = 256;            // with DF_CONV_ADDUP + kSignBits: y (the pre-add activation) is NOT written -- its sign bits are all
                                           // the backward pass needs of it (df_wino_conv_fwd_addup_bits / df_lrelu_bits_bwd_pool2x)
constexpr int kBitBytesPerBlock = 1024;    // 8 waves x 2 cout blocks x 64 lanes, per (tile block, cout slice)
constexpr int kZeroFloats = 1024;         // zeroed tail of the packed weights: SAME padding reads it, one 64-byte step per chunk
// PREC = 1 ("bf16x3" in the Winograd domain, see wino3d_kernel): LDS holds the z-TRANSFORMED input of a chunk, position-major with the 16
// channels of a position contiguous: [plane = (z-row, xi_z)][y 0..9][x 0..9 of pitch 12][16 channels + 4 pad floats]
constexpr int BPOS = 80;                  // bytes per position (16 channels + pad: 5 sixteen-byte slots, odd -> conflict-free ds_read_b128)
constexpr int BPY = 12;                   // positions per row (pitch)
constexpr int BPLANE = (9 * BPY + 10) * BPOS;      // bytes per plane (the last row needs 10 positions only)
constexpr int BBUF = 8 * BPLANE;          // bytes per LDS buffer (75520)
constexpr int BCOL = 400;                 // staging threads: one (y, x, channel quad) z-column of the halo block each

struct WinoArgs {
  const float* x;
  const f32x4* wp;
  const float* zeros;
  const float* bias;
  const float* residual;
  const float* mask_src;
  float* y;
  float* y2;           // DF_CONV_ADDUP: second output  y2 = y + nearest_up2x(residual)  (residual = the COARSE tensor)
  unsigned char* bits_out;        // kSignBits: sign pattern of y, one byte per (tile block, cout slice, wave, cout block, lane): bit s = output s
  const unsigned char* bits_in;   // kMaskBits: the lrelu mask of DF_CONV_MASK as such bytes instead of mask_src
  int B, D, H, W, Cin, Cout;
  int nbz, nby, nbx, ntb, ncs;
  int flags;
  float leak;
  int spx;
  int Wb;              // XS != 0: x-blocks of 4 per row of the x-blocked copy (2 nbx + 1); a.x then points to that copy
};

// ---- weight transform + packing ------------------------------------------------------------------------------------------
// mode 0: g[tap][k][n] = w[tap][k][n]          (K = cin,  N = cout)
// mode 1: g[tap][k][n] = w[26 - tap][n][k]     (K = cout, N = cin; taps mirrored)  -> dgrad operand
// Up[cs][xz][k4][nb][xy][kq][j][xx] = sum_taps G[xz][tz] G[xy][ty] G[xx][tx] g[tap][4 k4 + kq][32 cs + 16 nb + j]
// thread = one (k, n) filter: its 27 taps are read once and all 64 transform points come out of the separable G transform in fp64
// (3 -> 4 points per axis), written as 16 float4 (the four xi_x of a (xi_z, xi_y)); the float4 of consecutive n are contiguous.
__global__ __launch_bounds__(kPackT) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout,
                                                           int mode, int64_t total) {
  const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int64_t nfil = static_cast<int64_t>(K) * N;
  for (int64_t f = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; f < nfil; f += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(f % N), k = static_cast<int>(f / N);
    double g[27];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
      g[tap] = static_cast<double>(mode == 0 ? w[(static_cast<int64_t>(tap) * cin + k) * cout + n]
                                             : w[(static_cast<int64_t>(26 - tap) * cin + n) * cout + k]);
    // x: [tz][ty][3] -> [tz][ty][4]
    double gx[9][4];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const double a = g[r * 3], b = g[r * 3 + 1], c = g[r * 3 + 2];
      gx[r][0] = a; gx[r][1] = 0.5 * (a + b + c); gx[r][2] = 0.5 * (a - b + c); gx[r][3] = c;
    }
    const int cs = n >> 5, nb = (n >> 4) & 1, j = n & 15, k4 = k >> 2, kq = k & 3;
#pragma unroll
    for (int xz = 0; xz < 4; ++xz) {
      double gz[3][4];      // z combined: [ty][xx]
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
          const double a = gx[ty][xx], b = gx[3 + ty][xx], c = gx[6 + ty][xx];
          gz[ty][xx] = xz == 0 ? a : xz == 3 ? c : xz == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c);
        }
#pragma unroll
      for (int xy = 0; xy < 4; ++xy) {
        f32x4 o;
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
          const double a = gz[0][xx], b = gz[1][xx], c = gz[2][xx];
          o[xx] = static_cast<float>(xy == 0 ? a : xy == 3 ? c : xy == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c));
        }
        const int64_t idx = ((((((static_cast<int64_t>(cs) * 4 + xz) * (K / 4) + k4) * 2 + nb) * 4 + xy) * 4 + kq) * 16 + j) * 4;
        *reinterpret_cast<f32x4*>(wp + idx) = o;
      }
    }
  }
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < kZeroFloats; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    wp[total + i] = 0.f;
}

// "bf16x3" operand: the same transform, every U split into two bf16 words (hi = rne(U), lo = rne(U - hi)) and laid out for
// v_mfma_f32_16x16x16_bf16: Ub[cs][xz][k16][xy][xx][nb][lane = (k % 16) / 4 * 16 + n % 16][hi(k % 4 = 0..3) | lo(0..3)] -- one 16-byte load per
// lane gives the B operand (hi and lo) of a (transform point, cout 16-block) for the 16 input channels of a chunk.
__global__ __launch_bounds__(kPackT) void wino_pack_bf16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int cin, int cout,
                                                                  int mode, int64_t total) {
  const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
  const int64_t nfil = static_cast<int64_t>(K) * N;
  for (int64_t f = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; f < nfil; f += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(f % N), k = static_cast<int>(f / N);
    double g[27];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
      g[tap] = static_cast<double>(mode == 0 ? w[(static_cast<int64_t>(tap) * cin + k) * cout + n]
                                             : w[(static_cast<int64_t>(26 - tap) * cin + n) * cout + k]);
    double gx[9][4];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const double a = g[r * 3], b = g[r * 3 + 1], c = g[r * 3 + 2];
      gx[r][0] = a; gx[r][1] = 0.5 * (a + b + c); gx[r][2] = 0.5 * (a - b + c); gx[r][3] = c;
    }
    const int cs = n >> 5, nb = (n >> 4) & 1, j = n & 15, k16 = k >> 4, kq = (k >> 2) & 3, e = k & 3;
#pragma unroll
    for (int xz = 0; xz < 4; ++xz) {
      double gz[3][4];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
          const double a = gx[ty][xx], b = gx[3 + ty][xx], c = gx[6 + ty][xx];
          gz[ty][xx] = xz == 0 ? a : xz == 3 ? c : xz == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c);
        }
#pragma unroll
      for (int xy = 0; xy < 4; ++xy)
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
          const double a = gz[0][xx], b = gz[1][xx], c = gz[2][xx];
          const float u = static_cast<float>(xy == 0 ? a : xy == 3 ? c : xy == 1 ? 0.5 * (a + b + c) : 0.5 * (a - b + c));
          const __bf16 h = static_cast<__bf16>(u);
          const __bf16 l = static_cast<__bf16>(u - static_cast<float>(h));
          const int64_t rec = ((((((static_cast<int64_t>(cs) * 4 + xz) * (K / 16) + k16) * 4 + xy) * 4 + xx) * 2 + nb) * 64 + kq * 16 + j) * 8;
          wp[rec + e] = h;
          wp[rec + 4 + e] = l;
        }
    }
  }
  float* wz = reinterpret_cast<float*>(wp);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < kZeroFloats; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    wz[total + i] = 0.f;
}

// ---- packed-fp32 helpers (VOP3P): one instruction = two lanes of the separable B^T transform --------------------------------
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
// P = (x0,x1), Q = (x2,x3):  (x0 - x2, x1 + x2)
__device__ __forceinline__ f32x2 pk_bt01(f32x2 p, f32x2 q) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}
// (x2 - x1, x1 - x3)
__device__ __forceinline__ f32x2 pk_bt23(f32x2 p, f32x2 q) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}

// MODE 3 x stage on a coarse row (c0, c1 | c2, -):  P = (c0, c1) -> (c0 - c1, c1 + c1);  P, Q = (c2, -) -> (-, c1 - c2).  Packed results on
// purpose: a single-register temporary gets allocated into the never-read xi_x = 2 word of a weight quad whose load is still in flight
// (27-point modes), and its write then waits on vmcnt.
__device__ __forceinline__ f32x2 pk_x01(f32x2 p) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(p));
  return d;
}
__device__ __forceinline__ f32x2 pk_x3(f32x2 p, f32x2 q) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}

__device__ unsigned long long g_wino_prof[32];

struct BlockInfo {
  const float* xb;     // &x[b][0][0][0][0]
  int hoff;            // element offset of the block's halo origin (z0-1, y0-1, x0-1) inside the batch volume (may be < 0)
  int b, z0, y0, x0;
  int id;              // tile block index (for the sign-bit words)
};

// raw buffer descriptor (gfx9 data format word): out-of-range offsets read as zero -- the SAME padding of the staging loads
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Barrier that orders LDS traffic only.  __syncthreads() is a full fence: its s_waitcnt vmcnt(0) also waits for the acknowledgement of
// every global STORE issued before it -- in the epilogue that exposed two HBM write round trips per tile block.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FL >= 0: the epilogue flags are the compile-time constant FL (no per-element flag tests); FL < 0: run-time a.flags
// UP: the input is the COARSE tensor of an up-sampling-aware conv (x = nearest_up2x(xc) is never materialised): the staging reads
// xc[g >> 1] for fine halo coordinate g, and because every 2x2x2 tile then sees each coarse value twice, the transform points with
// index 2 in any axis are identically zero (B^T d = (c-1 - c0, 2 c0, 0, c0 - c1)): only 27 of the 64 points are multiplied
// (no wave takes xi_z = 2 and the xi_y|xi_x = 2 MFMAs are skipped) -- the 27-product form of the parity-class convolution.
// POOL (MODE 2): the adjoint case -- the output is the 2x2x2 sum-pool of the convolution (d/d(xc) of the up-sampling-aware conv: x is
// the fine gradient, y the COARSE tensor [B, D/2, H/2, W/2, Cout], accumulated into).  The pooled inverse transform is
// (A^T row 0 + row 1) = (1, 2, 0, -1) per axis, so the same 27 points are the only ones needed and a tile block writes 32 voxels, not 256.
//
// PREC = 1 -- "bf16x3" IN THE WINOGRAD DOMAIN (opt-in precision mode, ops.CONV_PRECISION = "bf16x3"; the default stays exact fp32).
// Same decomposition, staging loads, weights-from-L2 stream and epilogue; the 64 GEMMs run on the bf16 matrix pipe with both operands split
// into bf16 (hi, lo) words AFTER the fp32 transforms, three v_mfma_f32_16x16x16_bf16 per product block (lo*hi + hi*lo + hi*hi, fp32
// accumulate; tools/ubench/mfma_bf16_overlap.hip: 16.5 cycles each and -- unlike the fp32 MFMA -- plain VALU work hides beside them):
//   * the staging threads apply the z part of B^T once per chunk (a thread owns a z-column of the halo block: 6 loads -> the 8 (z-row, xi_z)
//     planes) and store 16 bytes = 4 channels of a position at a time: LDS = [plane][y][x][16 channels], 75.5 KB per buffer;
//   * a k-step is (chunk of 16 channels, xi_y): lane = (tile, channel quad) reads 2 rows x 4 positions of its plane as ds_read_b128 (its 4
//     channels at once), applies the y and x parts (32 plain v_add/v_sub: v_pk_* cost 15 cycles beside a bf16 MFMA) and splits the 4 points
//     x 4 channels into packed bf16 hi / lo (v_cvt_pk_bf16_f32; 48 ops) = the A operands of 4 points, K = 16;
//   * weights: U pre-split by wino_pack_bf16x3_kernel, one 16-byte load per lane = (hi, lo) of a (point, cout block): 8 loads per k-step.
// The fp32 values that are split are bit-identical to the PREC = 0 kernel's operands; per-product error 2^-17 (16 significand bits).
//
// XS != 0 -- STAGING BY LDS-DMA FROM AN X-BLOCKED COPY OF THE INPUT (round 4).  a.x = G[b][z][y][xb][c][4], element i of granule (xb, c) =
// x[b][z][y][4 xb - 1 + i][c] (zero outside the row: the SAME padding in x is part of the layout).  The 10 halo positions x0-1 .. x0+8 of a
// tile block (x0 = 8 bx) are the first 10 of the 12 floats of granules xb = 2 bx .. 2 bx + 2, and the 16 channels of a chunk of one granule
// column are 256 contiguous bytes: a halo ROW of a chunk is three such pieces = one `buffer_load_dwordx4 ... lds` of 51 lanes (lane =
// (piece g, channel c | pad)), global -> LDS with no staging registers, no ds_write, no transpose, two full 128-byte lines per piece
// instead of a 64-byte piece of every 512-byte voxel record.  LDS = [plane hz][row hy][piece g][17 slots: 16 channels + 1 pad][4 x]:
// piece pitch 68 dwords, row pitch 204 -- the wave-wide ds_read2_b64 of the A path (lane = (tile, channel); the two x pairs of a tile row
// sit in one or two pieces) stay bank-conflict free.  The row's address is scalar (SALU); a row outside the tensor reads through a
// zero-length descriptor (zeros = SAME padding in y, z).  60 DMA instructions per chunk.
// XS & 2: only waves 4-7 (one per SIMD) issue them -- the other wave of every SIMD never has a staging load in its in-order vmcnt queue and
// keeps the matrix pipe busy while its partner waits for HBM.  XS & 4: issued in front of k-step 0's MFMAs instead of behind them.
template <int DBG, int FL = -1, int MODE = 0, int PREC = 0, int XS = 0>
__global__ __launch_bounds__(kT, 1) void wino3d_kernel(const WinoArgs a) {
#ifndef DF_TUNING
  // The release translation unit can only instantiate the production combination: every DBG / XS / PREC experiment path below is dead code
  // there BY CONSTRUCTION (tests/test_cabi.py lists the instantiations the shipped library holds; DESIGN.md section 4 names them).
  static_assert(DBG == 0 && PREC == 0 && XS == 0 && MODE != 1, "wino3d_kernel: experiment variants exist in the -DDF_TUNING library only");
#endif
  constexpr bool UPC = MODE == 3;                    // UP with the coarse halo block staged as such (see UPY / UCP above)
  constexpr bool UP = MODE == 1 || UPC, POOL = MODE == 2, P27 = MODE != 0;
  constexpr bool BX = PREC == 1;
  constexpr bool XB = XS != 0, XASYM = (XS & 2) != 0;
  static_assert(!XB || (!UP && !BX), "x-blocked staging: plain / pooled fp32 modes only");
  static_assert(!UPC || !BX, "coarse staging: fp32 mode only");
  constexpr int PZk = UPC ? UPZ : PZ, CPk = UPC ? UCP : CP;      // pitches INSIDE a buffer; the buffer stride below keeps the plain size (the
                                                                // epilogue's 32 KB exchange area lives in the idle buffer)
  constexpr int NL = UPC ? NLU : NLOAD;              // staging pieces per thread and chunk
#ifndef DF_KEEPQ      // (-DDF_KEEPQ=1: keep the whole weight quad of a 27-point row live up to its reload -- the cure of the "dead xi_x = 2 word" hazard
#define DF_KEEPQ 0    //  that round 5 used before all transform temporaries were register PAIRS; costs 8 VGPRs, not needed any more)
#endif
  constexpr bool KEEPQ = DF_KEEPQ != 0;
  constexpr int XROWB = 51 * 16, XROWS = 60;        // XS: bytes per LDS row (3 pieces x 17 slots x 16 B), halo rows per chunk (6 planes x 10)
  constexpr int BUFk = XB ? XROWS * XROWB / 4 : CKW * CP;
  constexpr int BUFF = BX ? BBUF / 4 : BUFk;      // floats per LDS buffer
  // (experiment XS & 128: THREE staging buffers -- the pieces of chunk c + 2 are issued at the start of chunk c and waited for at the end of
  //  chunk c + 1: ~7 k-steps of slack instead of 2-3)
  constexpr bool TRI = (XS & 128) != 0;
  constexpr int NSB = TRI ? 3 : 2;
  __shared__ __attribute__((aligned(16))) float sIn[NSB * BUFF];
  __shared__ float sBias[32];        // this worker's cout slice of the bias (the slice is fixed for the worker's whole life)
  // lrelu-mask operands of a tile block's outputs, fetched by LDS-DMA loads (no registers); PREC = 1: behind the epilogue's exchange area in
  // the idle input buffer (32 + 32 of its 75.5 KB)
  __shared__ float sMs[BX ? 4 : 16 * kT];

  const int eflags = FL >= 0 ? FL : a.flags;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xz = wave & 3, th = wave >> 2;      // xi_z; tile z-row
  const int tl = lane & 15, kq = lane >> 4;     // tile within the row block (ty = tl >> 2, tx = tl & 3); cin % 4

  // ---- persistent worker -> (cout slice, sequence of tile blocks) ------------------------------------------------------------
  // Workgroup g runs on XCD g % 8.  When the slice count divides 8 an XCD only ever sees ONE slice (its 1 MB of transformed
  // weights stays L2-resident), its 32 workers walk 32 neighbouring tile blocks at a time (shared halos hit the L2) and the
  // other slices of those blocks run at the same time on the neighbouring XCDs (Infinity Cache serves the repeats).
  int cs, tb, tstride;
  {
    const int g = blockIdx.x, G = gridDim.x;
    if ((8 % a.ncs) == 0 && (G & 7) == 0) {
      // spx slices per XCD (spx | ncs): the XCD's weights are spx MB, and the spx workgroups of a tile block that share its L2
      // run at the same time on neighbouring CUs, so that block's input is fetched from the fabric once per XCD, not per slice
      const int spx = a.spx, xpg = a.ncs / spx;            // XCDs per group (a group covers all slices of its tile blocks)
      const int xcd = g & 7, slot = g >> 3, wx = G >> 3;   // wx workers per XCD
      const int ngroups = 8 / xpg, tw = wx / spx;          // tile workers per XCD
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
  if (tid < 32) sBias[tid] = (eflags & DF_CONV_BIAS) ? a.bias[n0 + tid] : 0.f;      // visible after the prologue's barrier
  const int tb0 = tb;
  const int niter = (a.ntb - tb0 + tstride - 1) / tstride;
  auto seq = [&](int k) -> int { return tb0 + k * tstride; };

  auto decode = [&](int t) -> BlockInfo {
    BlockInfo bi;
    bi.id = t;
    const int bx = t % a.nbx;
    int t2 = t / a.nbx;
    const int by = t2 % a.nby; t2 /= a.nby;
    const int bz = t2 % a.nbz;
    bi.b = t2 / a.nbz;
    bi.z0 = bz * 4; bi.y0 = by * 8; bi.x0 = bx * 8;
    bi.xb = a.x + static_cast<int64_t>(bi.b) * (XB ? a.D * a.H * a.Wb * 4 : UP ? (a.D >> 1) * (a.H >> 1) * (a.W >> 1) : a.D * a.H * a.W) * a.Cin;
    bi.hoff = (((bi.z0 - 1) * a.H + (bi.y0 - 1)) * a.W + (bi.x0 - 1)) * a.Cin;
    return bi;
  };

  const __amdgpu_buffer_rsrc_t wsrd_dbg = make_srd(a.wp, 4096u);
  // ---- staging plan (per thread: 5 float4 pieces of the 600-voxel x 16-channel halo block) ---------------------------------
  // so[it] = byte offset of piece `it` inside the batch volume, or an out-of-range offset for halo voxels outside the tensor:
  // the buffer load's range check then returns the zeros of the SAME padding.  Set once per tile block; a staging pass adds
  // only the chunk's scalar offset.
  int ldst[NLOAD];
  unsigned so[NLOAD];
#pragma unroll
  for (int it = 0; it < NL; ++it) {
    int p = it * kT + tid;
    if (UPC) {      // thread tid < 384 = (window w = tid >> 2, channel quad tid & 3); the three loads share one LDS destination (its window)
      const int wq = tid < UWIN * 4 ? tid : UWIN * 4 - 1, q4 = wq & 3, w = wq >> 2;
      const int t = w & 3, py = (w >> 2) % 6, pz = w / 24;
      ldst[it] = ((q4 * 4) * UCP + pz * UPZ + py * UPY + t * 4) * 4;
      continue;
    }
    if (p > HV * 4 - 1) p = HV * 4 - 1;     // the tail threads of the last pass duplicate the last piece (same data, same slot)
    const int hv = p >> 2, q4 = p & 3;
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
    ldst[it] = ((q4 * 4) * CPk + hz * PZk + hy * PY + hx) * 4;        // bytes, buffer 0
  }
  // pass `it` of the staging is taken by every wave, except the coarse block's second pass (64 pieces: wave 0) -- wave-uniform
  auto stage_pass = [&](int) -> bool { return !UPC || wave < UWIN * 4 / 64; };      // MODE 3: waves 0-5 stage (wave-uniform)
  const unsigned vol_bytes = static_cast<unsigned>(XB ? a.D * a.H * a.Wb * 4 : UP ? (a.D >> 1) * (a.H >> 1) * (a.W >> 1) : a.D * a.H * a.W) * a.Cin * 4u;
  auto set_offs = [&](const BlockInfo& bi) {
    // [r6] opaque thread id: LLVM otherwise hoists the five pieces' (hz, hy, hx, roff) out of the block loop -- 20 registers that it then spills
    // (37-56 spilled VGPRs in the plain instantiations, 15 in the pooled adjoint), and the reloads sit in the main loop behind `s_waitcnt vmcnt(0)`
    int tido = tid;
    asm volatile("" : "+v"(tido));
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      int p = it * kT + tido;
      if (UPC) {      // coarse voxel (pz, py, px = t + it) of the 4 x 6 x 6 block whose origin is the coarse voxel under fine (z0 - 1, y0 - 1, x0 - 1)
        const int wq = tid < UWIN * 4 ? tid : UWIN * 4 - 1, q4 = wq & 3, w = wq >> 2;
        const int px = (w & 3) + 2 * it, py = (w >> 2) % 6, pz = w / 24;
        const int Dc = a.D >> 1, Hc = a.H >> 1, Wc = a.W >> 1;
        const int cz = (bi.z0 >> 1) - 1 + pz, cy = (bi.y0 >> 1) - 1 + py, cx = (bi.x0 >> 1) - 1 + px;
        bool ok = static_cast<unsigned>(cz) < static_cast<unsigned>(Dc) && static_cast<unsigned>(cy) < static_cast<unsigned>(Hc) &&
                  static_cast<unsigned>(cx) < static_cast<unsigned>(Wc);      // fine g in range <=> coarse g >> 1 in range (even extents)
        if (DBG & 4) ok = false;
        so[it] = ok ? static_cast<unsigned>(((cz * Hc + cy) * Wc + cx) * a.Cin + q4 * 4) * 4u : 0x80000000u;
        continue;
      }
      if (p > HV * 4 - 1) p = HV * 4 - 1;
      const int hv = p >> 2, q4 = p & 3;
      const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
      const int roff = ((hz * a.H + hy) * a.W + hx) * a.Cin + q4 * 4;
      const int gz = bi.z0 - 1 + hz, gy = bi.y0 - 1 + hy, gx = bi.x0 - 1 + hx;
      bool ok = static_cast<unsigned>(gz) < static_cast<unsigned>(a.D) && static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) &&
                static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
      if (DBG & 4) ok = false;
      if (UP) {
        const int Hc = a.H >> 1, Wc = a.W >> 1;
        so[it] = ok ? static_cast<unsigned>((((gz >> 1) * Hc + (gy >> 1)) * Wc + (gx >> 1)) * a.Cin + q4 * 4) * 4u : 0x80000000u;
      } else {
        so[it] = ok ? static_cast<unsigned>(bi.hoff + roff) * 4u : 0x80000000u;
        if ((DBG & 384) == 384 && ok) so[it] &= 0xFFFC0u;      // (experiment 384: every staging load inside one 1 MB window -- L2 hits, L1 misses)
      }
    }
  };
  auto stage_load = [&](int it, __amdgpu_buffer_rsrc_t srd, unsigned chunkbytes) -> f32x4 {
    if ((DBG & 384) == 256) return buf_load16(wsrd_dbg, static_cast<unsigned>(lane) * 16u, 0u);      // always-cached address (latency experiment)
    if (DBG & 4194304) {      // (experiment, timing only: FULLY COALESCED staging loads -- lane i reads 16 B at base + 16 i, 1 KiB contiguous per wave
                              //  instruction, 8 full lines -- that miss L1: the base walks the tensor with the real offsets' block / chunk part)
      const unsigned basev = (so[it] == 0x80000000u ? 0u : so[it]) & 0x7FFFFC00u;
      return buf_load16(srd, __builtin_amdgcn_readfirstlane(basev) + static_cast<unsigned>(lane) * 16u, chunkbytes & ~1023u);
    }
    return buf_load16(srd, so[it], chunkbytes);
  };
  char* sInB = reinterpret_cast<char*>(sIn);
  auto stage_store = [&](int it, int bufbytes, const f32x4& v) {
    float* d = reinterpret_cast<float*>(sInB + (ldst[it] + bufbytes));
    d[0] = v[0]; d[CPk] = v[1]; d[2 * CPk] = v[2]; d[3 * CPk] = v[3];
  };
  // all pieces of a thread; MODE 3: the three loaded voxels (c0, c1, c2) x 4 channels -> the window (c0, c1, c2, -) of each channel
  auto stage_store_all = [&](int bufbytes, const auto& v) {
    if constexpr (XB) {      // (x-blocked LDS-DMA staging: nothing passes through registers)
    } else if constexpr (UPC) {
      if (!stage_pass(0)) return;
      char* d = sInB + (ldst[0] + bufbytes);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // lane = quad + 4 t inside a 16-lane DPP row: voxel t + 1 = first voxel of lane + 4 (row_shl:4); lanes 12-15 (t = 3) have no lane + 4 and
        // keep `old` = the second voxel (t + 2 = 4) of lane - 4 (row_shr:4)
        const float f0 = v[0][e], f2 = v[1][e];      // (bit_cast of a vector-ELEMENT lvalue reads element 0: copy to scalars first)
        const int c0 = __builtin_bit_cast(int, f0), c2 = __builtin_bit_cast(int, f2);
        const int fromlo = __builtin_amdgcn_update_dpp(0, c2, 0x114, 0xF, 0xF, false);            // row_shr:4
        const int c1 = __builtin_amdgcn_update_dpp(fromlo, c0, 0x104, 0xF, 0xF, false);           // row_shl:4, out-of-row lanes keep fromlo
        *reinterpret_cast<f32x4*>(d + e * UCP * 4) = f32x4{v[0][e], __builtin_bit_cast(float, c1), v[1][e], 0.f};
      }
    } else {
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) stage_store(it, bufbytes, v[it]);
    }
  };

  // ---- XS staging: DMA instruction = halo row r = hz * 10 + hy; lane = (piece g, slot c), 51 of 64 lanes active -------------------------
  const bool xissuer = XB && (!XASYM || wave >= 4);
  const int xg = lane / 17, xc = lane - xg * 17;
  const unsigned xvoff = (lane < 51 && xc < 16) ? static_cast<unsigned>((xg * a.Cin + xc) * 16) : 0x80000000u;      // pad slot: zeros
  auto issue_dma = [&](int bufbytes, const BlockInfo& bi, unsigned chunkbytes) {      // chunkbytes = chunk * 256 (16 channels x 16 B)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    if (!xissuer) return;
    const __amdgpu_buffer_rsrc_t dsrd = make_srd(bi.xb, vol_bytes);
    const int rowb = a.Wb * a.Cin * 16;                                             // bytes per (z, y) row of the copy
    const int base = (((bi.z0 - 1) * a.H + (bi.y0 - 1)) * a.Wb + (bi.x0 >> 2)) * a.Cin * 16 + static_cast<int>(chunkbytes);
    // a rolled loop of scalar row arithmetic (nothing of it lives in registers across the main loop); a row outside the tensor reads
    // through the out-of-range lane offset (zeros = the SAME padding in y, z)
    auto one_row = [&](int r) {
      const int hz = r / 10, hy = r - hz * 10;
      const bool ok = static_cast<unsigned>(bi.z0 - 1 + hz) < static_cast<unsigned>(a.D) &&
                      static_cast<unsigned>(bi.y0 - 1 + hy) < static_cast<unsigned>(a.H);
      // branch-free: a row outside the tensor keeps its (meaningless, never dereferenced) scalar offset and gets the out-of-range bit in
      // the LANE offset -- the range check looks at the lane offset only
      const unsigned soff = static_cast<unsigned>(base + (hz * a.H + hy) * rowb);
      const unsigned vo = xvoff | (ok ? 0u : 0x80000000u);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dsrd, (lds_ptr)(sInB + bufbytes + r * XROWB), 16, vo, soff, 0, 0);
    };
    if (lane < 51) {
      if constexpr (TRI) {        // exactly 8 pieces per wave (the surplus one of waves 4-7 repeats row 59): the chunk-end wait counts on it
        static_assert(!TRI || !XASYM, "three staging buffers: symmetric issue only");
#pragma unroll 1
        for (int j = 0; j < 8; ++j) { const int r0 = wave + 8 * j; one_row(r0 < XROWS ? r0 : XROWS - 1); }
      } else if constexpr ((XS & 8) != 0) {      // unrolled: the compiler counts the pieces in its vmcnt waits (a rolled loop makes it assume none were issued)
        int w0 = XASYM ? wave - 4 : wave;
#pragma unroll
        for (int j = 0; j < (XASYM ? 15 : 8); ++j) {
          asm volatile("" : "+s"(w0));      // opaque: the row arithmetic is redone per piece instead of living in SGPRs across the main loop
          const int r0 = w0 + (XASYM ? 4 : 8) * j;
          one_row(r0 < XROWS ? r0 : XROWS - 1);      // (the surplus piece of waves 4-7 repeats row 59: same bytes, same place)
        }
      } else {
#pragma unroll 1
        for (int r = XASYM ? wave - 4 : wave; r < XROWS; r += XASYM ? 4 : 8) one_row(r);
      }
    }
  };

  // (experiment XS & 256, with three buffers: piece j of the wave is issued behind MFMA row j of k-step 0 instead of all 8 in one burst)
  auto issue_piece = [&](int j, int bufbytes, const BlockInfo& bi, unsigned chunkbytes) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const __amdgpu_buffer_rsrc_t dsrd = make_srd(bi.xb, vol_bytes);
    const int rowb = a.Wb * a.Cin * 16;
    const int base = (((bi.z0 - 1) * a.H + (bi.y0 - 1)) * a.Wb + (bi.x0 >> 2)) * a.Cin * 16 + static_cast<int>(chunkbytes);
    const int r0 = wave + 8 * j, r = r0 < XROWS ? r0 : XROWS - 1;
    const int hz = r / 10, hy = r - hz * 10;
    const bool ok = static_cast<unsigned>(bi.z0 - 1 + hz) < static_cast<unsigned>(a.D) && static_cast<unsigned>(bi.y0 - 1 + hy) < static_cast<unsigned>(a.H);
    const unsigned soff = static_cast<unsigned>(base + (hz * a.H + hy) * rowb);
    const unsigned vo = xvoff | (ok ? 0u : 0x80000000u);      // (lanes >= 51 carry the out-of-range bit in xvoff already ... but they must not WRITE: exec mask)
    if (lane < 51) __builtin_amdgcn_raw_ptr_buffer_load_lds(dsrd, (lds_ptr)(sInB + bufbytes + r * XROWB), 16, vo, soff, 0, 0);
  };

  // ---- PREC = 1 staging plan: thread p < 400 owns the z-column (hy, hx, channel quad) of the halo block -------------------------
  unsigned soz[6];
  const int colp = tid < BCOL ? tid : BCOL - 1;
  const int chy = colp / 40, chx = (colp >> 2) % 10, cq4 = colp & 3;
  const int ldsz = (chy * BPY + chx) * BPOS + cq4 * 16;      // bytes, plane 0 of buffer 0
  auto set_offs_b = [&](const BlockInfo& bi) {
    const int gy = bi.y0 - 1 + chy, gx = bi.x0 - 1 + chx;
    const bool okyx = tid < BCOL && static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) && static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
#pragma unroll
    for (int z = 0; z < 6; ++z) {
      const int gz = bi.z0 - 1 + z;
      const bool ok = okyx && static_cast<unsigned>(gz) < static_cast<unsigned>(a.D);
      if (UP) {
        const int Hc = a.H >> 1, Wc = a.W >> 1;
        soz[z] = ok ? static_cast<unsigned>((((gz >> 1) * Hc + (gy >> 1)) * Wc + (gx >> 1)) * a.Cin + cq4 * 4) * 4u : 0x80000000u;
      } else {
        soz[z] = ok ? static_cast<unsigned>(bi.hoff + ((z * a.H + chy) * a.W + chx) * a.Cin + cq4 * 4) * 4u : 0x80000000u;
      }
    }
  };
  auto vsub4 = [](const f32x4& p, const f32x4& q) -> f32x4 {      // plain v_sub_f32 x 4 (hipcc would form v_pk_add_f32)
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float t; asm("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(p[e]), "v"(q[e])); d[e] = t; }
    return d;
  };
  auto vadd4 = [](const f32x4& p, const f32x4& q) -> f32x4 {
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float t; asm("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(p[e]), "v"(q[e])); d[e] = t; }
    return d;
  };
  auto stage_store_b = [&](int bufbytes, const f32x4 (&pl)[6]) {      // z part of B^T (same operations as the PREC = 0 lanes apply) + 8 stores
    if (tid < BCOL) {
      char* d = sInB + (bufbytes + ldsz);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        *reinterpret_cast<f32x4*>(d + (t * 4 + 0) * BPLANE) = vsub4(pl[2 * t], pl[2 * t + 2]);
        *reinterpret_cast<f32x4*>(d + (t * 4 + 1) * BPLANE) = vadd4(pl[2 * t + 1], pl[2 * t + 2]);
        if (!P27) *reinterpret_cast<f32x4*>(d + (t * 4 + 2) * BPLANE) = vsub4(pl[2 * t + 2], pl[2 * t + 1]);
        *reinterpret_cast<f32x4*>(d + (t * 4 + 3) * BPLANE) = vsub4(pl[2 * t + 1], pl[2 * t + 3]);
      }
    }
  };

  // ---- A operand: this lane's tile, planes (za, zb) of xi_z ---------------------------------------------------------------
  const int tx = tl & 3, ty = tl >> 2;
  // main-loop role (mz = xi_z, mth = tile z-row).  Plain conv: the wave's (xz, th).  27-point modes: only xi_z in {0, 1, 3} is
  // multiplied, 6 (xi_z, z-row) roles of 2 x 9 MFMAs per k-step: waves 0-3 take (xi_z in {0, 1}) x (z-row) with both 16-cout blocks,
  // waves 4-7 split the two xi_z = 3 roles by cout block (hnb) -- every SIMD (wave & 3) then issues 27 MFMAs per k-step.
  const bool half = P27 && wave >= 4;
  const int mz = !P27 ? xz : (wave < 4 ? (wave >> 1) : 3);
  const int mth = !P27 ? th : (wave < 4 ? (wave & 1) : ((wave >> 1) & 1));
  const int hnb = half ? (wave & 1) : 0;
  const int za = mz == 0 ? 0 : mz == 2 ? 2 : 1;
  const int zb = mz == 0 ? 2 : mz == 1 ? 2 : mz == 2 ? 1 : 3;
  const float qs = mz == 1 ? 1.f : -1.f;
  const f32x2 qs2 = {qs, qs};
  const int abase = kq * CPk + (2 * mth) * PZk + (2 * ty) * PY + 2 * tx;
  const int offA = abase + za * PZk, offB = abase + zb * PZk;

  // MODE 3: the tile's fine planes 2 mth + {0..3} repeat the coarse planes mth + {0, 1, 1, 2} (fine halo index h <-> coarse (h + 1) >> 1), its
  // fine rows / columns the coarse rows ty + {0, 1, 1, 2} / columns tx + {0, 1, 1, 2}: the lane reads the 3 x 3 patch of the two coarse planes
  // its xi_z combines and applies the SAME operations to the same values as the fine-grid transform (bit-identical operands):
  //   per axis  (d0 - d2, d1 + d2, -, d1 - d3)  with  d = (c0, c1, c1, c2)   ->   (c0 - c1, c1 + c1, -, c1 - c2)
  const int uoffA = (kq * UCP + (mth + ((za + 1) >> 1)) * UPZ + ty * UPY + tx * 4) * 4;      // bytes: window tx of coarse row ty
  const int uoffB = (kq * UCP + (mth + ((zb + 1) >> 1)) * UPZ + ty * UPY + tx * 4) * 4;
  f32x4 ua[3], ub[3];      // [coarse row]: (c0, c1, c2, -)
  f32x2 ra[8], rb[8];      // raw inputs [y][x pair] of planes za / zb
  f32x2 T[8], U[8];        // after the z / y transform
  f32x2 A2[8];             // A operands of a k-step: A2[xi_y*2 + h] = (xi_x = 2h, 2h+1)
  const int offAb = offA * 4, offBb = offB * 4;      // bytes (multiples of 8)
  // XS: x pair h of this lane's tile = floats 2 tx + 2 h, + 1 of the row = piece (tx + h) >> 1, element 2 ((tx + h) & 1)
  const int xrowA = ((2 * mth + za) * 10 + 2 * ty) * XROWB + kq * 16, xrowB = ((2 * mth + zb) * 10 + 2 * ty) * XROWB + kq * 16;
  const int xp0 = (tx >> 1) * (17 * 16) + (tx & 1) * 8, xp1 = ((tx + 1) >> 1) * (17 * 16) + ((tx + 1) & 1) * 8;
  auto raw_read_up = [&](int idxbytes) {      // idxbytes: LDS byte offset of channel plane 4 ks (+ buffer)
    int ia = idxbytes + uoffA, ib = idxbytes + uoffB;
    asm volatile("" : "+v"(ia), "+v"(ib));
    __builtin_assume((ia & 15) == 0);
    __builtin_assume((ib & 15) == 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ua[r] = *reinterpret_cast<const f32x4*>(sInB + ia + r * UPY * 4);
      ub[r] = *reinterpret_cast<const f32x4*>(sInB + ib + r * UPY * 4);
    }
  };
  auto raw_read_x = [&](int idxbytes) {   // idxbytes: buffer + 64 ks (the k-step's 4 channels are slots 4 ks .. 4 ks + 3)
    int ia0 = idxbytes + xrowA + xp0, ia1 = idxbytes + xrowA + xp1, ib0 = idxbytes + xrowB + xp0, ib1 = idxbytes + xrowB + xp1;
    asm volatile("" : "+v"(ia0), "+v"(ia1), "+v"(ib0), "+v"(ib1));
    __builtin_assume((ia0 & 7) == 0); __builtin_assume((ia1 & 7) == 0); __builtin_assume((ib0 & 7) == 0); __builtin_assume((ib1 & 7) == 0);
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      ra[y * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ia0 + y * XROWB);
      ra[y * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ia1 + y * XROWB);
      rb[y * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ib0 + y * XROWB);
      rb[y * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ib1 + y * XROWB);
    }
  };
  auto raw_read = [&](int idxbytes) {   // idxbytes: LDS byte offset of plane 4 ks (+ buffer), without this lane's offset
    if constexpr (UPC) { raw_read_up(idxbytes); return; }
    int ia = idxbytes + offAb, ib = idxbytes + offBb;
    asm volatile("" : "+v"(ia), "+v"(ib));     // opaque: the 16 row reads become 8 ds_read2_b64 with small immediate offsets
    __builtin_assume((ia & 7) == 0);
    __builtin_assume((ib & 7) == 0);
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      ra[y * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ia + (y * PY) * 4);
      ra[y * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ia + (y * PY + 2) * 4);
      rb[y * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ib + (y * PY) * 4);
      rb[y * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ib + (y * PY + 2) * 4);
    }
  };
  auto transform_up = [&]() {
    f32x2 tl2[3], th2[3];      // z: rows as (c0, c1) and (c2, -)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      tl2[r] = pk_fma(f32x2{ub[r][0], ub[r][1]}, qs2, f32x2{ua[r][0], ua[r][1]});
      th2[r] = pk_fma(f32x2{ub[r][2], ub[r][3]}, qs2, f32x2{ua[r][2], ua[r][3]});
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k == 2) { A2[4] = f32x2{0.f, 0.f}; A2[5] = f32x2{0.f, 0.f}; continue; }      // xi_y = 2 is never multiplied
      const f32x2 ul = k == 0 ? pk_sub(tl2[0], tl2[1]) : k == 1 ? pk_add(tl2[1], tl2[1]) : pk_sub(tl2[1], tl2[2]);      // y
      const f32x2 uh = k == 0 ? pk_sub(th2[0], th2[1]) : k == 1 ? pk_add(th2[1], th2[1]) : pk_sub(th2[1], th2[2]);
      A2[k * 2 + 0] = pk_x01(ul);                                                       // x: xi_x = 0, 1: (c0 - c1, c1 + c1)
      A2[k * 2 + 1] = pk_x3(ul, uh);                                                    // xi_x = (2), 3: (-, c1 - c2)
    }
  };
  auto transform = [&]() {
    if constexpr (UPC) { transform_up(); return; }
#pragma unroll
    for (int j = 0; j < 8; ++j) T[j] = pk_fma(rb[j], qs2, ra[j]);          // z
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                           // y
      U[0 + h] = pk_sub(T[0 + h], T[4 + h]);
      U[2 + h] = pk_add(T[2 + h], T[4 + h]);
      U[4 + h] = pk_sub(T[4 + h], T[2 + h]);
      U[6 + h] = pk_sub(T[2 + h], T[6 + h]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                           // x
      A2[k * 2 + 0] = pk_bt01(U[k * 2], U[k * 2 + 1]);
      A2[k * 2 + 1] = pk_bt23(U[k * 2], U[k * 2 + 1]);
    }
  };

  // ---- B operand ------------------------------------------------------------------------------------------------------------
  const int nk4 = a.Cin >> 2;
  // XS & 32: TWO weight register sets (even / odd k-steps): a row is reloaded right behind its MFMAs with the weights of k-step + 2, i.e.
  // every weight load has two k-steps (~3000 cycles) to come back instead of one -- the staging pieces' HBM misses sit in front of the
  // weight loads in the CU's in-order vector-memory return path (the DMA staging freed the registers for it)
  constexpr int NBQ = (XS & 32) ? 2 : 1;
  f32x4 bq[NBQ][2][4];     // [k-step parity][cout 16-block][xi_y] = (xi_x 0..3)
  const unsigned laneb = static_cast<unsigned>(lane) * 16u;
  // (experiments 262144 / 524288, timing only: the weight loads of the tile z-row 1 waves / of every wave go through a zero-length descriptor
  //  -- same instructions, no cache traffic: what the DUPLICATE weight stream of the two z-row waves costs the CU's vector-memory path)
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, ((DBG & 524288) || ((DBG & 262144) && th == 1)) ? 0u : static_cast<unsigned>(a.Cin) * a.Cout * 256u);
  const unsigned wbase_b = static_cast<unsigned>((cs * 4 + mz) * nk4) * 8192u + static_cast<unsigned>(hnb) * 4096u;
  auto issue_b = [&](int nb, int k4) {
    const int kl = k4 < nk4 ? k4 : 0;        // wraps to the first k-step of the next tile block
    const unsigned sb = wbase_b + static_cast<unsigned>(kl) * 8192u + nb * 4096u;      // wave-uniform
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[0][nb][q] = buf_load16(wsrd, laneb + q * 1024u, sb);
  };
  auto issue_b1 = [&](int nb) {      // NBQ == 2: k-step 1 into the odd set
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[NBQ - 1][nb][q] = buf_load16(wsrd, laneb + q * 1024u, wbase_b + 8192u + nb * 4096u);
  };

  // ---- PREC = 1: A operands of a k-step = (chunk, xi_y = j); B operands ------------------------------------------------------------
  const int aoffB = (mth * 4 + mz) * BPLANE + ((2 * ty) * BPY + 2 * tx) * BPOS + kq * 16;      // this lane's tile, its plane, its channel quad
  f32x4 rwa[4], rwb[4];      // the two rows of the y combination: 4 positions x 4 channels each
  u32x2 Ah[4], Al[4];        // [xi_x]: packed bf16 (hi | lo) of the lane's 4 channels
  auto raw_read_b = [&](int bufbytes, int j) {
    const int rA = j == 0 ? 0 : j == 2 ? 2 : 1, rB = j == 0 ? 2 : j == 1 ? 2 : j == 2 ? 1 : 3;      // U = row rA -+ row rB
    int ia = bufbytes + aoffB;
    asm volatile("" : "+v"(ia));
    __builtin_assume((ia & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rwa[i] = *reinterpret_cast<const f32x4*>(sInB + ia + (rA * BPY + i) * BPOS);
      rwb[i] = *reinterpret_cast<const f32x4*>(sInB + ia + (rB * BPY + i) * BPOS);
    }
  };
  auto split2 = [](float v0, float v1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    float l0, l1;
    asm("v_sub_f32 %0, %1, %2" : "=v"(l0) : "v"(v0), "v"(h0));
    asm("v_sub_f32 %0, %1, %2" : "=v"(l1) : "v"(v1), "v"(h1));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(l0), "v"(l1));
  };
  auto transform_b = [&](int j) {
    f32x4 Uy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Uy[i] = j == 1 ? vadd4(rwa[i], rwb[i]) : vsub4(rwa[i], rwb[i]);
#pragma unroll
    for (int xx = 0; xx < 4; ++xx) {
      if (P27 && xx == 2) continue;
      const f32x4 v = xx == 0 ? vsub4(Uy[0], Uy[2]) : xx == 1 ? vadd4(Uy[1], Uy[2]) : xx == 2 ? vsub4(Uy[2], Uy[1]) : vsub4(Uy[1], Uy[3]);
      unsigned h01, l01, h23, l23;
      split2(v[0], v[1], h01, l01);
      split2(v[2], v[3], h23, l23);
      Ah[xx] = u32x2{h01, h23};
      Al[xx] = u32x2{l01, l23};
    }
  };
  u32x4 bw[2][4];            // [cout 16-block][xi_x] = (hi k0k1, hi k2k3, lo k0k1, lo k2k3)
  const unsigned wbaseB = static_cast<unsigned>((cs * 4 + mz) * (a.Cin >> 4)) * 32768u + static_cast<unsigned>(hnb) * 1024u;
  auto issue_bw = [&](int nb, int k16, int j) {      // record (k16, xi_y = j): 4 xi_x x 2 cout blocks x 1 KB
    const unsigned sb = wbaseB + static_cast<unsigned>(k16 * 4 + j) * 8192u + nb * 1024u;      // wave-uniform
#pragma unroll
    for (int xx = 0; xx < 4; ++xx)
      if (!(P27 && xx == 2)) bw[nb][xx] = __builtin_bit_cast(u32x4, buf_load16(wsrd, laneb, sb + xx * 2048u));
  };

  f32x4 acc[2][16];
  const int nchunk = a.Cin / CKW;
  // (experiments 1048576 / 2097152: the 5 staging loads of a thread are not issued as one burst behind k-step 0's MFMAs -- all 8 waves at the same
  //  moment: 40 instructions of 16 lines each into the CU's vector-memory path -- but ONE per MFMA row, behind that row's weight reload, during
  //  k-step 1 (stores just before the chunk barrier) / during k-step 0 (stores in k-step 2 as in production))
  constexpr bool SPREAD1 = (DBG & 1048576) != 0, SPREAD0 = (DBG & 2097152) != 0, SPREAD = SPREAD1 || SPREAD0;

  // ---- prologue: first block's chunk 0 -> buffer 0 ------------------------------------------------------------------------------
  BlockInfo cur = decode(seq(0));
  if constexpr (BX) {
    set_offs_b(cur);
    const __amdgpu_buffer_rsrc_t srd0 = make_srd(cur.xb, vol_bytes);
    f32x4 stz[6];
#pragma unroll
    for (int z = 0; z < 6; ++z) stz[z] = buf_load16(srd0, soz[z], 0u);
    stage_store_b(0, stz);
  } else if constexpr (XB) {
    issue_dma(0, cur, 0u);
    if constexpr (TRI) { issue_dma(BUFk * 4, cur, 256u); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    set_offs(cur);
    const __amdgpu_buffer_rsrc_t srd0 = make_srd(cur.xb, vol_bytes);
    f32x4 stg[NLOAD];
#pragma unroll
    for (int it = 0; it < NL; ++it) if (stage_pass(it)) stg[it] = stage_load(it, srd0, 0u);
    stage_store_all(0, stg);
  }
  __syncthreads();

  int pb = 0;          // buffer parity of the block's chunk 0
  for (int it = 0; it < niter; ++it) {
    const unsigned long long tp0 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
    const BlockInfo nxt = decode(seq(it + 1 < niter ? it + 1 : it));
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // first raw inputs of this block (and, for the worker's first block, the weights of k-step 0)
    if constexpr (BX) {
      raw_read_b(pb * BBUF, 0);
      issue_bw(0, 0, 0);
      if (!half) issue_bw(1, 0, 0);
    } else {
      if constexpr (XB) raw_read_x(pb * BUFk * 4); else raw_read(pb * BUFk * 4);      // (TRI: pb = chunk counter mod 3)
      // [r3] the weights of k-step 0 are the same for every tile block of the worker and the last k-step of a block has already reloaded
      // them (issue_b / reload_row wrap around): they stay in their registers across the epilogue (16.13 -> 16.05 ms per top-level
      // launch; tuning variant 65536 = reloaded at every block start)
      if (it == 0) {
        issue_b(0, 0);
        if (!half) issue_b(1, 0);
        if (NBQ == 2) { issue_b1(0); if (!half) issue_b1(1); }
      }
    }

    const unsigned long long tp1 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // The 16 lrelu-mask operands of this lane's outputs go global -> LDS by DMA loads (in registers hipcc spills each one to scratch
    // behind its own vmcnt(0): 14 serial HBM round trips).  Issued at the top of the epilogue and read back after the first combine
    // (issuing them inside the last chunk costs more than it hides: +20 % kernel time); every thread reads only what it requested.
    auto mask_dma = [&]() {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      const int oz0 = cur.z0 + 2 * th, oy0 = cur.y0 + 2 * kq, ox0 = cur.x0 + 2 * xz;
      const int sW = a.Cout, sH = a.W * a.Cout, sD = sH * a.H;
      const __amdgpu_buffer_rsrc_t msrd = make_srd(a.mask_src + static_cast<int64_t>(cur.b) * a.D * a.H * a.W * a.Cout,
                                                   static_cast<unsigned>(a.D) * a.H * a.W * a.Cout * 4u);
      const unsigned mv = static_cast<unsigned>(((oz0 * a.H + oy0) * a.W + ox0) * a.Cout + n0 + tl) * 4u;
      float* const sM = BX ? sIn + ((nchunk - 1 + pb) & 1) * BUFF + 8192 : sMs;
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const unsigned so_ = static_cast<unsigned>(n2 * 16 + (s >> 2) * sD + ((s >> 1) & 1) * sH + (s & 1) * sW) * 4u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(msrd, (lds_ptr)(sM + (n2 * 8 + s) * kT + wave * 64), 4, mv, so_, 0, 0);
        }
    };
    auto main_loop = [&](auto half_c, auto lw_c) {
    constexpr bool HALF = decltype(half_c)::value;      // this wave owns one 16-cout block only (acc[0], weights of block hnb)
    // (experiment XS & 64, timing only: this copy of the loop -- taken by waves 4-7 -- reads its "weights" from LDS (whatever is there) instead of
    //  global memory: the vmcnt queue of the waves that stage (4-7 with XS & 2) then holds the staging pieces only -- what a design in which
    //  the weights arrive through an LDS ring would give them; the ring's own fill traffic is not modelled)
    constexpr bool LW = decltype(lw_c)::value;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int bo = TRI ? ((chunk + pb) % 3) * BUFk * 4 : ((chunk + pb) & 1) * BUFk * 4;      // byte offsets of this / the next chunk's buffer
      const int bn = TRI ? ((chunk + 1 + pb) % 3) * BUFk * 4 : BUFk * 4 - bo;
      const int bn2 = ((chunk + 2 + pb) % 3) * BUFk * 4;                                       // TRI: where the pieces of chunk + 2 go
      const bool lastc = chunk + 1 == nchunk;
      if constexpr (!XB) {
        if (lastc) set_offs(nxt);                    // the last chunk stages the next tile block's first chunk
      }
      const __amdgpu_buffer_rsrc_t ssrd = make_srd(lastc ? nxt.xb : cur.xb, vol_bytes);
      const unsigned schunk = XB ? static_cast<unsigned>(lastc ? 0 : chunk + 1) * 256u : static_cast<unsigned>(lastc ? 0 : chunk + 1) * (CKW * 4u);
      f32x4 stg[XB ? 1 : NLOAD];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // -- A operands of this k-step (its raw inputs were requested during the previous one) --
        const unsigned long long q0 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        if (DBG & 16) asm volatile("s_waitcnt lgkmcnt(0)");
        const unsigned long long q1 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        if (XB && TRI && !(XS & 256) && ks == 0 && !(DBG & 4)) {
          const int cc = chunk + 2;
          issue_dma(bn2, cc < nchunk ? cur : nxt, static_cast<unsigned>(cc < nchunk ? cc : cc - nchunk) * 256u);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (XB && !TRI && (XS & 16) && ks == 0 && !(DBG & 4)) { issue_dma(bn, lastc ? nxt : cur, schunk); __builtin_amdgcn_sched_barrier(0); }
        if (!(DBG & 1)) transform();
        __builtin_amdgcn_sched_barrier(0);
        if (!XB && ks == (SPREAD1 ? 3 : 2) && !(DBG & 4) && !(DBG & 64)) {
          stage_store_all(bn, stg);
        }
        // XS: the DMA pieces of the next chunk have landed once at most the 8 (4) weight reloads of k-step 2 are outstanding behind them
        if (XB && ks == 3) {      // (every wave: one that issued no pieces has only those reloads outstanding and does not wait)
          constexpr int NWL = (LW ? 0 : (P27 ? 3 : 4) * (HALF ? 1 : 2) * NBQ) + (TRI ? 8 : 0);      // loads issued behind the pieces that must have landed
          if constexpr (NWL == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if constexpr (NWL == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else if constexpr (NWL == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else if constexpr (NWL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else if constexpr (NWL == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        if (ks == 2 && (DBG & 64)) {       // keep the loads alive without the LDS writes
#pragma unroll
          for (int it = 0; it < NL; ++it) if (stage_pass(it)) asm volatile("" :: "v"(stg[it]));
        }
        // next chunk staged by everyone; everyone is done reading the planes it overwrote.  [r3] An LDS-only barrier: the staged data was
        // already waited for at the LDS writes of ks == 2, and __syncthreads()' vmcnt(0) drained the weight loads in flight (this and the
        // two epilogue barriers: 16.21 -> 16.05 ms per top-level launch, round 3)
        if (ks == 3) lds_barrier();
        if (!(DBG & 2)) {                                                    // raw inputs of the next k-step
          if constexpr (XB) raw_read_x(ks < 3 ? bo + (ks + 1) * 64 : bn); else raw_read(ks < 3 ? bo + (ks + 1) * 16 * CPk : bn);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (XB && !TRI && (XS & 20) == 4 && ks == 0 && !(DBG & 4)) { issue_dma(bn, lastc ? nxt : cur, schunk); __builtin_amdgcn_sched_barrier(0); }
        const unsigned long long q2 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        if (DBG & 16) asm volatile("s_waitcnt vmcnt(4)");
        const unsigned long long q3 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        // [r3] a xi_y row's weight registers are reloaded right after its 4 MFMAs, not after the cout block's 16: every weight load is in
        // flight up to 12 MFMAs longer before the next k-step needs it (16.52 -> 16.27 ms per top-level launch; tuning variant 196608 = the
        // old order)
        auto reload_row = [&](int nb, int q) {
          if ((DBG & 8) || (P27 && q == 2)) return;
          const int k4 = chunk * 4 + ks + NBQ;
          const int kl = k4 < nk4 ? k4 : k4 - nk4;      // wraps to the first k-step(s) of the next tile block
          const unsigned sb = wbase_b + static_cast<unsigned>(kl) * 8192u + nb * 4096u;
          if constexpr (LW) bq[ks & (NBQ - 1)][nb][q] = *reinterpret_cast<const f32x4*>(sInB + bo + (nb * 4 + q) * 1024 + lane * 16);
          else bq[ks & (NBQ - 1)][nb][q] = buf_load16(wsrd, laneb + q * 1024u, sb);
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (!P27 || ((i >> 2) != 2 && (i & 3) != 2))
            acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[i >> 1][i & 1], bq[ks & (NBQ - 1)][0][i >> 2][i & 3], acc[0][i], 0, 0, 0);
          if ((i & 3) == 3) {
            // 27-point modes never read the xi_x = 2 word of a weight quad: without this use of the WHOLE quad the register allocator hands that
            // dead register to an A operand, whose write then has to wait (s_waitcnt vmcnt) for the quad's load issued a moment ago
            if (P27 && KEEPQ) asm volatile("" :: "v"(bq[ks & (NBQ - 1)][0][i >> 2]));
            __builtin_amdgcn_sched_barrier(0);
            reload_row(0, i >> 2);
            if (XB && TRI && (XS & 256) && ks == 0 && !(DBG & 4)) {
              const int cc = chunk + 2;
              issue_piece(i >> 2, bn2, cc < nchunk ? cur : nxt, static_cast<unsigned>(cc < nchunk ? cc : cc - nchunk) * 256u);
            }
            if (!XB && SPREAD && ks == (SPREAD1 ? 1 : 0) && !(DBG & 4)) stg[i >> 2] = stage_load(i >> 2, ssrd, schunk);      // pieces 0..3
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!HALF) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (!P27 || ((i >> 2) != 2 && (i & 3) != 2))
              acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[i >> 1][i & 1], bq[ks & (NBQ - 1)][1][i >> 2][i & 3], acc[1][i], 0, 0, 0);
            if ((i & 3) == 3) {
              if (P27 && KEEPQ) asm volatile("" :: "v"(bq[ks & (NBQ - 1)][1][i >> 2]));
              __builtin_amdgcn_sched_barrier(0);
              reload_row(1, i >> 2);
              if (XB && TRI && (XS & 256) && ks == 0 && !(DBG & 4)) {
                const int cc = chunk + 2;
                issue_piece(4 + (i >> 2), bn2, cc < nchunk ? cur : nxt, static_cast<unsigned>(cc < nchunk ? cc : cc - nchunk) * 256u);
              }
              if (!XB && SPREAD && ks == (SPREAD1 ? 1 : 0) && !(DBG & 4) && (i >> 2) == 1) stg[4] = stage_load(4, ssrd, schunk);      // piece 4
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (XB && !TRI && !(XS & 20) && ks == 0 && !(DBG & 4)) issue_dma(bn, lastc ? nxt : cur, schunk);
        if (!XB && !SPREAD && ks == 0 && !(DBG & 4)) {     // staging loads of the next chunk, right behind a weight batch: vmcnt retires in order, so
                                         // the first wait that covers them is the one for the NEXT weight batch (1.5 k-steps away)
#pragma unroll
          for (int it = 0; it < NL; ++it)
            if (stage_pass(it)) stg[it] = (DBG & 384) == 128 ? f32x4{0.f, 0.f, 0.f, 0.f} : stage_load(it, ssrd, schunk);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 16) {
          const unsigned long long q4 = __builtin_readcyclecounter();
          ph[0] += q1 - q0; ph[1] += q2 - q1; ph[2] += q3 - q2; ph[3] += q4 - q3;
        }
      }
    }
    };
    // PREC = 1: k-step = (chunk, xi_y); the 27-point modes skip xi_y = 2 (and the xi_x = 2 products)
    auto main_loop_b = [&](auto half_c) {
      constexpr bool HALF = decltype(half_c)::value;
      constexpr int NST = P27 ? 3 : 4;
      for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int bo = ((chunk + pb) & 1) * BBUF, bn = BBUF - bo;
        const bool lastc = chunk + 1 == nchunk;
        if (lastc) set_offs_b(nxt);
        const __amdgpu_buffer_rsrc_t ssrd = make_srd(lastc ? nxt.xb : cur.xb, vol_bytes);
        const unsigned schunk = static_cast<unsigned>(lastc ? 0 : chunk + 1) * (CKW * 4u);
        const int cnext = lastc ? 0 : chunk + 1;      // (the weights of the next tile block's first k-step)
        f32x4 stz[6];
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          const int j = P27 && st == 2 ? 3 : st;                                   // xi_y of this k-step
          const int jn = st + 1 < NST ? (P27 && st + 1 == 2 ? 3 : st + 1) : 0;     // ... and of the next one
          const int kn = st + 1 < NST ? chunk : cnext;
          if (!(DBG & 1)) transform_b(j);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int STS = (DBG & 16) ? NST - 1 : NST - 2;      // (experiment 16: store the staged chunk one k-step later)
          if (st == STS && !(DBG & 4)) stage_store_b(bn, stz);
          if (st == NST - 1) __syncthreads();
          if (!(DBG & 2)) raw_read_b(st + 1 < NST ? bo : bn, jn);
          __builtin_amdgcn_sched_barrier(0);
          auto mfma_nb = [&](int nb) {
            if (DBG & 32) {      // (experiment 32: xi_x-major -- a point's three products back to back, its weight registers reloaded at once)
#pragma unroll
              for (int xx = 0; xx < 4; ++xx)
                if (!(P27 && xx == 2)) {
#pragma unroll
                  for (int t = 0; t < 3; ++t) {
                    const u32x2 av = t == 0 ? Al[xx] : Ah[xx];
                    const u32x2 bv = t == 1 ? u32x2{bw[nb][xx][2], bw[nb][xx][3]} : u32x2{bw[nb][xx][0], bw[nb][xx][1]};
                    acc[nb][j * 4 + xx] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, av), __builtin_bit_cast(s16x4, bv),
                                                                                    acc[nb][j * 4 + xx], 0, 0, 0);
                  }
                  __builtin_amdgcn_sched_barrier(0);
                  if (!(DBG & 8)) {
                    const unsigned sb = wbaseB + static_cast<unsigned>(kn * 4 + jn) * 8192u + nb * 1024u;
                    bw[nb][xx] = __builtin_bit_cast(u32x4, buf_load16(wsrd, laneb, sb + xx * 2048u));
                  }
                  __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
              for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int xx = 0; xx < 4; ++xx)
                  if (!(P27 && xx == 2)) {
                    const u32x2 av = t == 0 ? Al[xx] : Ah[xx];
                    const u32x2 bv = t == 1 ? u32x2{bw[nb][xx][2], bw[nb][xx][3]} : u32x2{bw[nb][xx][0], bw[nb][xx][1]};
                    acc[nb][j * 4 + xx] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, av), __builtin_bit_cast(s16x4, bv),
                                                                                    acc[nb][j * 4 + xx], 0, 0, 0);
                  }
              __builtin_amdgcn_sched_barrier(0);
              if (!(DBG & 8)) issue_bw(nb, kn, jn);
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          if (st == 0 && (DBG & 64)) {      // (experiment 64: staging loads ahead of the k-step's MFMAs)
#pragma unroll
            for (int z = 0; z < 6; ++z) stz[z] = buf_load16(ssrd, soz[z], schunk);
            __builtin_amdgcn_sched_barrier(0);
          }
          mfma_nb(0);
          if (!HALF) mfma_nb(1);
          if (st == 0 && !(DBG & 4) && !(DBG & 64)) {      // staging loads of the next chunk, right behind a weight batch (see the PREC = 0 loop)
#pragma unroll
            for (int z = 0; z < 6; ++z) stz[z] = buf_load16(ssrd, soz[z], schunk);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if constexpr (BX) {
      if (half) main_loop_b(std::true_type{}); else main_loop_b(std::false_type{});
    } else {
      if constexpr ((XS & 64) != 0) {
        main_loop(std::false_type{}, std::true_type{});      // (every wave: a single copy of the loop -- two copies in one kernel spill 150-250 B)
      } else {
        if (half) main_loop(std::true_type{}, std::false_type{}); else main_loop(std::false_type{}, std::false_type{});
      }
    }

    const unsigned long long tp2 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
    // ---- epilogue: inverse transform in x, y per accumulator element; z across the waves through the idle LDS buffer ---------
    // (the buffer of the last chunk is free since that chunk's barrier; the other one holds the next block's chunk 0)
    if constexpr (POOL) {
      const int lb = ((nchunk - 1 + pb) & 1) * BUFF;
      float* sP = sIn + lb;      // [xi_z][tz][e][lane]: the (y, x)-pooled value of accumulator element e
      const int Dc = a.D >> 1, Hc = a.H >> 1, Wc = a.W >> 1;
      const int cz = (cur.z0 >> 1) + th, cy = (cur.y0 >> 1) + kq, cx = (cur.x0 >> 1) + xz;      // this wave combines e = xz of its z-row
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
    } else {
      const int lb = TRI ? ((nchunk - 1 + pb) % 3) * BUFF : ((nchunk - 1 + pb) & 1) * BUFF;
      const float* const sM = BX ? sIn + lb + 8192 : sMs;
      f32x4* sO = reinterpret_cast<f32x4*>(sIn + lb);      // [xi_z][tz][e][lane] float4 = (oy0ox0, oy0ox1, oy1ox0, oy1ox1)
      // this wave combines accumulator element e = xi_z of its own z-row:  tile (ty = lane>>4, tx = e), cout = lane & 15
      const int oz0 = cur.z0 + 2 * th, oy0 = cur.y0 + 2 * kq, ox0 = cur.x0 + 2 * xz;
      const int64_t sW = a.Cout, sH = static_cast<int64_t>(a.W) * a.Cout, sD = sH * a.H;
      const int64_t obase = (((static_cast<int64_t>(cur.b) * a.D + oz0) * a.H + oy0) * a.W + ox0) * a.Cout + n0 + tl;
      const bool full = cur.z0 + 4 <= a.D && cur.y0 + 8 <= a.H && cur.x0 + 8 <= a.W;
      const unsigned lane_off = static_cast<unsigned>(((oz0 * a.H + oy0) * a.W + ox0) * a.Cout + n0 + tl) * 4u;      // bytes within the batch volume
      constexpr bool SB = FL >= 0 && (FL & kSignBits) != 0, MB = FL >= 0 && (FL & kMaskBits) != 0, NOY = FL >= 0 && (FL & kNoPrimary) != 0;
      if (full && (eflags & DF_CONV_MASK) && !MB) mask_dma();
      // the fp32-mask DMA path counts on vmcnt(8) with nothing but its own loads outstanding: keep the draining barriers there
      const bool full_bar = (eflags & DF_CONV_MASK) && !MB;
      const int64_t wbase = (static_cast<int64_t>(cur.id) * a.ncs + cs) * kBitBytesPerBlock + wave * 128 + lane;
      float rres[2][8];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float bv = sBias[nb * 16 + tl];
        unsigned mbyte = 0u, sbyte = 0u;      // kMaskBits: this lane's 8 mask bits of the cout block (one byte load, used after the combine)
        if (MB) mbyte = a.bits_in[wbase + nb * 64];
        auto emit = [&](const f32x4 (&c)[16]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x2 px[4];
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
              if (P27) {      // the xi = 2 points were never multiplied
                px[yy][0] = c[yy * 4 + 0][e] + c[yy * 4 + 1][e];
                px[yy][1] = c[yy * 4 + 1][e] - c[yy * 4 + 3][e];
              } else {
                px[yy][0] = c[yy * 4 + 0][e] + c[yy * 4 + 1][e] + c[yy * 4 + 2][e];
                px[yy][1] = c[yy * 4 + 1][e] - c[yy * 4 + 2][e] - c[yy * 4 + 3][e];
              }
            }
            const f32x2 o01 = P27 ? px[0] + px[1] : px[0] + px[1] + px[2], o23 = P27 ? px[1] - px[3] : px[1] - px[2] - px[3];
            sO[((mz * 2 + mth) * 4 + e) * 64 + lane] = f32x4{o01[0], o01[1], o23[0], o23[1]};
          }
        };
        const unsigned long long e0 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        if (!half) emit(acc[nb]);
        else if (nb == hnb) emit(acc[0]);
        if (DBG & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long e1 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        if (nb == 0 && full && (eflags & DF_CONV_RESIDUAL)) {
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int s = 0; s < 8; ++s)
              rres[n2][s] = a.residual[obase + n2 * 16 + (s >> 2) * sD + ((s >> 1) & 1) * sH + (s & 1) * sW];
        }
        if (full_bar) __syncthreads(); else lds_barrier();      // (the combine below reads LDS only; the residual loads are waited for at their use)
        const unsigned long long e2 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        const f32x4 m0 = sO[((0 * 2 + th) * 4 + xz) * 64 + lane], m1 = sO[((1 * 2 + th) * 4 + xz) * 64 + lane];
        const f32x4 m3 = sO[((3 * 2 + th) * 4 + xz) * 64 + lane];
        f32x4 lo = m0 + m1, hi = m1 - m3;
        if (!P27) {
          const f32x4 m2 = sO[((2 * 2 + th) * 4 + xz) * 64 + lane];
          lo += m2; hi -= m2;
        }
        // mask DMA of THIS cout block landed: the 16 loads were issued block 0 first and vmcnt retires in order, so "at most 8
        // outstanding" covers block 0 while block 1's eight are still in flight, and block 1 while block 0's eight stores are
        if (full && (eflags & DF_CONV_MASK) && !MB) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        float rup = 0.f;      // DF_CONV_ADDUP: the 2x2x2 outputs of this lane's tile share ONE coarse voxel of the skip tensor
        if ((eflags & DF_CONV_ADDUP) && oz0 < a.D && oy0 < a.H && ox0 < a.W)
          rup = a.residual[(((static_cast<int64_t>(cur.b) * (a.D >> 1) + (oz0 >> 1)) * (a.H >> 1) + (oy0 >> 1)) * (a.W >> 1) + (ox0 >> 1)) * a.Cout +
                           n0 + nb * 16 + tl];
        // Full blocks of the variants without a per-output operand (bias / lrelu / sign bits only: the 27-point up-sampling-aware forward and the plain
        // forward): the stores go through the 4 x 4 lane-quad transpose of conv_wino43.hip -- 2 float4 stores per lane and cout block instead of 8 scalar
        // ones; bias, lrelu and the sign bits stay in the accumulator layout (the sign-byte layout is unchanged).  Bit-identical.
        const bool wide = full && !(eflags & (DF_CONV_RESIDUAL | DF_CONV_MASK | DF_CONV_ADDUP)) && !NOY && !(DBG & 512);
        if (wide) {
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
          const int64_t o4 = obase - tl + 4 * (tl >> 2) + nb * 16 + (qi >> 1) * sH + (qi & 1) * sW;
          *reinterpret_cast<f32x4*>(a.y + o4) = quad_t(vz[0]);
          *reinterpret_cast<f32x4*>(a.y + o4 + sD) = quad_t(vz[1]);
        } else
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          float v = (s < 4 ? lo[s & 3] : hi[s & 3]) + bv;
          if (eflags & DF_CONV_LRELU) v = fmaxf(v, a.leak * v);
          const int64_t o = obase + nb * 16 + (s >> 2) * sD + ((s >> 1) & 1) * sH + (s & 1) * sW;
          if (SB) sbyte |= v > 0.f ? (1u << s) : 0u;      // (outputs outside the tensor carry don't-care bits)
          const bool mpos = !MB || ((mbyte >> s) & 1u) != 0u;
          if (full) {
            if (eflags & DF_CONV_RESIDUAL) v += rres[nb][s];
            if (eflags & DF_CONV_MASK) v = (MB ? mpos : sM[(nb * 8 + s) * kT + tid] > 0.f) ? v : a.leak * v;
            // wave-uniform base (batch volume + this output's scalar offset) + 32-bit lane offset: no 64-bit address per output
            char* yb = reinterpret_cast<char*>(a.y + static_cast<int64_t>(cur.b) * a.D * a.H * a.W * a.Cout +
                                               ((s >> 2) * sD + ((s >> 1) & 1) * sH + (s & 1) * sW) + nb * 16);
            if (DBG & 512) asm volatile("" :: "v"(v));      // (experiment: no stores)
            else if (!NOY) *reinterpret_cast<float*>(yb + lane_off) = v;
            if (eflags & DF_CONV_ADDUP) {
              char* yb2 = reinterpret_cast<char*>(a.y2 + static_cast<int64_t>(cur.b) * a.D * a.H * a.W * a.Cout +
                                                  ((s >> 2) * sD + ((s >> 1) & 1) * sH + (s & 1) * sW) + nb * 16);
              *reinterpret_cast<float*>(yb2 + lane_off) = v + rup;
            }
          } else if (oz0 + (s >> 2) < a.D && oy0 + ((s >> 1) & 1) < a.H && ox0 + (s & 1) < a.W) {
            if (eflags & DF_CONV_RESIDUAL) v += a.residual[o];
            if (eflags & DF_CONV_MASK) v = (MB ? mpos : a.mask_src[o] > 0.f) ? v : a.leak * v;
            if (!NOY) a.y[o] = v;
            if (eflags & DF_CONV_ADDUP) a.y2[o] = v + rup;
          }
        }
        if (SB) a.bits_out[wbase + nb * 64] = static_cast<unsigned char>(sbyte);
        const unsigned long long e3 = (DBG & 16) ? __builtin_readcyclecounter() : 0ull;
        if (full_bar) __syncthreads(); else lds_barrier();      // (the output stores drain behind the next cout block's exchange / the next tile block)
        if (DBG & 16) {
          const unsigned long long e4 = __builtin_readcyclecounter();
          ph[4 + 0] += e1 - e0; ph[4 + 1] += e2 - e1; ph[4 + 2] += e3 - e2; ph[4 + 3] += e4 - e3;
        }
      }
    }
    if ((DBG & 16) && blockIdx.x == 8 && tid == 0) {
      const unsigned long long tp3 = __builtin_readcyclecounter();
      g_wino_prof[0] += tp1 - tp0; g_wino_prof[1] += tp2 - tp1; g_wino_prof[2] += tp3 - tp2; g_wino_prof[3] += 1;
      for (int i = 0; i < 8; ++i) g_wino_prof[4 + i] += ph[i];
    }
    pb = TRI ? (pb + nchunk) % 3 : (pb + nchunk) & 1;
    cur = nxt;
  }
}

// The backward tail of an up-sampling generator block from the SIGN BITS of its last conv (kSignBits of wino3d_kernel):
//   gx = gy * (bit ? 1 : leak),   gpool = 2x2x2 sum-pool of gy.
// One thread = one coarse voxel x 4 channels.  The 8 fine voxels under a coarse voxel are exactly the 8 outputs of ONE lane of the
// producing kernel's epilogue (oz0 = z0 + 2 th, oy0 = y0 + 2 kq, ox0 = x0 + 2 xi_z; bit s = (dz, dy, dx)), so the 8 signs of a channel
// are one byte and the thread's 4 channels one aligned 32-bit word.  Summation order (dz, dy, dx) ascending as upsample_bwd_kernel.
__global__ __launch_bounds__(256) void lrelu_bits_bwd_pool_kernel(const float4* __restrict__ gy, const unsigned* __restrict__ bits,
                                                                  float4* __restrict__ gx, float4* __restrict__ gpool, float leak,
                                                                  int64_t nsrc4, int Dc, int Hc, int Wc, int C4, int nbz, int nby, int nbx,
                                                                  int ncs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= nsrc4) return;
  const int c4 = static_cast<int>(i % C4);
  int64_t r = i / C4;
  const int xc = static_cast<int>(r % Wc); r /= Wc;
  const int yc = static_cast<int>(r % Hc); r /= Hc;
  const int zc = static_cast<int>(r % Dc);
  const int64_t b = r / Dc;
  const int64_t id = ((b * nbz + (zc >> 1)) * nby + (yc >> 2)) * nbx + (xc >> 2);
  const int wave = (zc & 1) * 4 + (xc & 3), kq = yc & 3;
  const int cs = c4 >> 3, nb = (c4 >> 2) & 1, tl = (c4 & 3) * 4;
  const unsigned word = bits[((id * ncs + cs) * kBitBytesPerBlock + wave * 128 + nb * 64 + kq * 16 + tl) >> 2];
  const int64_t W2 = 2 * Wc, H2 = 2 * Hc, D2 = 2 * Dc;
  float4 g[8];
  int64_t idx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
    idx[k] = (((b * D2 + (2 * zc + dz)) * H2 + (2 * yc + dy)) * W2 + (2 * xc + dx)) * C4 + c4;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = gy[idx[k]];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    acc.x += g[k].x; acc.y += g[k].y; acc.z += g[k].z; acc.w += g[k].w;
    float4 o;
    o.x = ((word >> k) & 1u) ? g[k].x : leak * g[k].x;
    o.y = ((word >> (8 + k)) & 1u) ? g[k].y : leak * g[k].y;
    o.z = ((word >> (16 + k)) & 1u) ? g[k].z : leak * g[k].z;
    o.w = ((word >> (24 + k)) & 1u) ? g[k].w : leak * g[k].w;
    gx[idx[k]] = o;
  }
  gpool[i] = acc;
}

// NDHWC -> the x-blocked layout of the XS staging (see wino3d_kernel): G[b][z][y][xb][c][0..3] = x[b][z][y][4 xb - 1 + i][c], zero outside.
// One thread = one 16-byte granule; the lanes of a wave are consecutive channels: four 256-byte reads, one 1 KiB write.
__global__ __launch_bounds__(256) void to_xblk_kernel(const float* __restrict__ x, f32x4* __restrict__ g, int64_t ngran, int W, int Wb, int C) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= ngran) return;
  const int c = static_cast<int>(i % C);
  const int64_t r = i / C;
  const int xb = static_cast<int>(r % Wb);
  const int64_t row = r / Wb;
  const float* src = x + (row * W) * C + c;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int xs = 4 * xb - 1 + e;
    v[e] = (xs >= 0 && xs < W) ? src[static_cast<int64_t>(xs) * C] : 0.f;
  }
  __builtin_nontemporal_store(v, g + i);
}

#ifdef DF_TUNING      // instrumented kernel variants + knobs of the tuning library only (include/deepfluids_hip_debug.h)
int g_wino_dbg = 0;
int g_wino_spx = 0;     // slices per XCD override
#else
constexpr int g_wino_spx = 0;
#endif
}  // namespace

extern "C" {

#ifdef DF_TUNING
void df_debug_set_wino(int v) { g_wino_dbg = v & 0x3ffffff; g_wino_spx = v >> 26; }
int df_debug_wino_prof(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_prof), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_prof), 32 * sizeof(unsigned long long));
}
#endif

#ifdef DF_TUNING
static int64_t wino_grid(WinoArgs& a, int64_t ntb);
// Round-4 probe: df_wino_conv_fwd (flags = DF_CONV_BIAS | DF_CONV_LRELU) with the input staged by LDS-DMA from an x-blocked copy `xg`
// ([B, D, H, 2 ceil(W/8) + 1, Cin, 4] floats, written here by to_xblk_kernel unless variant & 256).  variant & 15 = XS of wino3d_kernel:
// 1 all waves issue the DMA pieces behind k-step 0's MFMAs | 3 waves 4-7 only | 5 / 7 the same in front of the MFMAs;
// variant & 16: DBG 4 (no staging at all, timing only).
int64_t df_debug_wino_xblk_elems(int64_t B, int64_t D, int64_t H, int64_t W, int64_t C) { return B * D * H * (2 * ceil_div(W, 8) + 1) * C * 4; }
int df_debug_wino_conv_fwd_xblk(const float* x, float* xg, const float* wp, const float* bias, float* y, int64_t B, int64_t D, int64_t H, int64_t W,
                                int64_t Cin, int64_t Cout, float leak, int variant, df_stream_t stream) {
  DF_REQUIRE(x && xg && wp && bias && y, DF_EINVAL, "df_debug_wino_conv_fwd_xblk: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "df_debug_wino_conv_fwd_xblk: shape");
  WinoArgs a;
  a.x = xg; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = nullptr; a.mask_src = nullptr; a.y = y; a.y2 = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  a.Wb = 2 * a.nbx + 1;
  DF_REQUIRE(D * H * a.Wb * 4 * Cin <= (1LL << 29), DF_ESHAPE, "df_debug_wino_conv_fwd_xblk: one batch volume of the copy must stay below 2 GiB");
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  a.ntb = (int)ntb;
  a.flags = DF_CONV_BIAS | DF_CONV_LRELU; a.leak = leak;
  hipStream_t s = df::as_stream(stream);
  if (!(variant & 256)) {
    const int64_t ngran = B * D * H * a.Wb * Cin;
    hipLaunchKernelGGL(to_xblk_kernel, dim3((unsigned)ceil_div(ngran, 256)), dim3(256), 0, s, x, reinterpret_cast<f32x4*>(xg), ngran, (int)W, a.Wb, (int)Cin);
  }
  const int64_t grid = wino_grid(a, ntb);
  constexpr int F = DF_CONV_BIAS | DF_CONV_LRELU;
  switch (variant & 127) {
    case 1: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 1>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 3: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 3>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 5: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 5>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 7: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 7>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 9: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 9>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 11: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 11>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 13: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 13>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 15: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 15>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 17: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 17>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 25: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 25>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 27: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 27>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 32 + 1: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 33>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 32 + 9: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 41>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 32 + 11: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 43>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 32 + 13: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 45>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 32 + 17: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 49>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 32 + 31: hipLaunchKernelGGL((wino3d_kernel<4, F, 0, 0, 33>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 3 + 64: hipLaunchKernelGGL((wino3d_kernel<4, F, 0, 0, 3>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 104: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 256 + 128 + 64 + 1>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;           // three buffers + LDS "weights", pieces spread over k-step 0
    case 106: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 256 + 128 + 1>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;                // three buffers, global weights, pieces spread
    case 98: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 128 + 64 + 1>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;                 // three buffers + LDS "weights"
    case 102: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 128 + 1>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;                     // three buffers, global weights
    case 64 + 32 + 1: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 64 + 1>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;            // all waves stage, LDS "weights"
    case 64 + 32 + 17: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 64 + 17>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;
    case 64 + 32 + 4: hipLaunchKernelGGL((wino3d_kernel<4, F, 0, 0, 64 + 3>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;            // no staging, LDS "weights"
    case 64 + 32 + 3: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 64 + 3>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;            // waves 4-7 stage (behind the MFMAs, rolled) and take LDS "weights"
    case 64 + 32 + 19: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 64 + 19>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;          // ... pieces at the start of k-step 0
    case 64 + 32 + 7: hipLaunchKernelGGL((wino3d_kernel<0, F, 0, 0, 64 + 7>), dim3((unsigned)grid), dim3(kT), 0, s, a); break;            // ... in front of the MFMAs
    default: return df::fail(DF_EINVAL, "df_debug_wino_conv_fwd_xblk: unknown variant");
  }
  return df::launched("df_debug_wino_conv_fwd_xblk");
}
// The "bf16x3 in the Winograd domain" experiment (wino3d_kernel PREC = 1), tuning library only: same arguments as df_wino_pack_weights /
// df_wino_conv_fwd.  df_debug_set_wino: 0 production order | 32 xi_x-major MFMA order | diagnosis variants (results wrong by construction)
// 1 no transform | 2 no LDS operand reads | 4 no staging | 8 no weight loads and sums | 16 staged chunk stored one k-step later |
// 64 staging loads ahead of the k-step's MFMAs.
int df_debug_wino_pack_weights_bf16x3(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_debug_wino_pack_weights_bf16x3: null pointer");
  DF_REQUIRE(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0 && (mode == 0 || mode == 1), DF_ESHAPE,
             "df_debug_wino_pack_weights_bf16x3: cin, cout must be multiples of 32; mode 0|1");
  const int64_t total = 64 * cin * cout;      // in fp32 units: (hi, lo) bf16 = 4 bytes per transformed weight
  int64_t g = ceil_div(cin * cout, 64);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(wino_pack_bf16x3_kernel, dim3((unsigned)g), dim3(64), 0, df::as_stream(stream), w, reinterpret_cast<__bf16*>(wp), (int)cin,
                     (int)cout, mode, total);
  return df::launched("df_debug_wino_pack_weights_bf16x3");
}
int df_debug_wino_conv_fwd_bf16x3(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src, float* y,
                                  int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak,
                                  df_stream_t stream) {
  DF_REQUIRE(x && wp && y, DF_EINVAL, "df_debug_wino_conv_fwd_bf16x3: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE,
             "df_debug_wino_conv_fwd_bf16x3: positive extents, Cin, Cout multiples of 32");
  DF_REQUIRE(D * H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_debug_wino_conv_fwd_bf16x3: one batch volume must stay below 2 GiB");
  DF_REQUIRE(!(flags & DF_CONV_ADDUP) && (!(flags & DF_CONV_BIAS) || bias) && (!(flags & DF_CONV_RESIDUAL) || residual) &&
                 (!(flags & DF_CONV_MASK) || mask_src), DF_EINVAL, "df_debug_wino_conv_fwd_bf16x3: flag without its operand");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x), DF_EALIGN, "df_debug_wino_conv_fwd_bf16x3: x and packed weights must be 16-byte aligned");
  WinoArgs a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = residual; a.mask_src = mask_src; a.y = y; a.y2 = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_debug_wino_conv_fwd_bf16x3: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = flags; a.leak = leak;
  const int64_t grid = wino_grid(a, ntb);
#define DF_WB(V) case V: hipLaunchKernelGGL((wino3d_kernel<V, DF_CONV_BIAS | DF_CONV_LRELU, 0, 1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break
  switch (g_wino_dbg) {
    case 0: hipLaunchKernelGGL((wino3d_kernel<0, -1, 0, 1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    DF_WB(1); DF_WB(2); DF_WB(3); DF_WB(4); DF_WB(8); DF_WB(12); DF_WB(7); DF_WB(15); DF_WB(11); DF_WB(16); DF_WB(32); DF_WB(64);
    default: return df::fail(DF_EINVAL, "df_debug_wino_conv_fwd_bf16x3: unknown debug variant");
  }
#undef DF_WB
  return df::launched("df_debug_wino_conv_fwd_bf16x3");
}
#endif

int64_t df_wino_packed_elems(int64_t cin, int64_t cout, int mode) {
  (void)mode;
  return 64 * cin * cout + kZeroFloats;
}

int df_wino_pack_weights(const float* w, float* wp, int64_t cin, int64_t cout, int mode, df_stream_t stream) {
  DF_REQUIRE(w && wp, DF_EINVAL, "df_wino_pack_weights: null pointer");
  DF_REQUIRE(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0 && (mode == 0 || mode == 1), DF_ESHAPE,
             "df_wino_pack_weights: cin, cout must be multiples of 32; mode 0|1");
  const int64_t total = 64 * cin * cout;
  int64_t g = ceil_div(cin * cout, 64);      // one wave per workgroup: 16 K filters still cover all 256 CUs
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)g), dim3(64), 0, df::as_stream(stream), w, wp, (int)cin, (int)cout, mode,
                     total);
  return df::launched("df_wino_pack_weights");
}

static int64_t wino_grid(WinoArgs& a, int64_t ntb) {
  // persistent workers: one workgroup per CU (a workgroup owns a CU's whole register file and most of its LDS)
  int64_t grid = df::kCUs;
  a.spx = 1;
  if (8 % a.ncs == 0) {
    a.spx = (g_wino_spx > 0 && a.ncs % g_wino_spx == 0) ? g_wino_spx : (a.ncs % 2 == 0 ? 2 : 1);
    const int xpg = a.ncs / a.spx, ngroups = 8 / xpg;
    const int64_t need = ceil_div(ntb, ngroups) * a.spx * 8;      // workers that get at least one tile block
    if (need < grid) grid = need;
    if ((grid >> 3) % a.spx) grid = ((grid >> 3) / a.spx + 1) * a.spx * 8;
    if (grid > df::kCUs) grid = df::kCUs;
  } else {
    grid = (grid / a.ncs) * a.ncs;
    if (ntb * a.ncs < grid) grid = ntb * a.ncs;
  }
  return grid;
}

int df_wino_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, const float* mask_src,
                     float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak,
                     df_stream_t stream) {
  DF_REQUIRE(x && wp && y, DF_EINVAL, "df_wino_conv_fwd: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DF_EINVAL, "df_wino_conv_fwd: non-positive extent");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE,
             "df_wino_conv_fwd: Cin, Cout must be multiples of 32 (use df_conv_fwd otherwise)");
  // (staging goes through 32-bit byte offsets into one batch volume, with 0x80000000 as the out-of-range sentinel)
  DF_REQUIRE(D * H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino_conv_fwd: one batch volume must stay below 2 GiB (use df_conv_fwd)");
  DF_REQUIRE(!(flags & DF_CONV_ADDUP), DF_EINVAL, "df_wino_conv_fwd: DF_CONV_ADDUP needs df_wino_conv_fwd_addup");
  DF_REQUIRE(!(flags & DF_CONV_BIAS) || bias, DF_EINVAL, "df_wino_conv_fwd: DF_CONV_BIAS without bias");
  DF_REQUIRE(!(flags & DF_CONV_RESIDUAL) || residual, DF_EINVAL, "df_wino_conv_fwd: DF_CONV_RESIDUAL without residual");
  DF_REQUIRE(!(flags & DF_CONV_MASK) || mask_src, DF_EINVAL, "df_wino_conv_fwd: DF_CONV_MASK without mask_src");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x) && df::aligned16(y), DF_EALIGN, "df_wino_conv_fwd: x, y and packed weights must be 16-byte aligned");
  WinoArgs a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = residual; a.mask_src = mask_src; a.y = y; a.y2 = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino_conv_fwd: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = flags; a.leak = leak;
  const int64_t grid = wino_grid(a, ntb);
#ifdef DF_TUNING
  const int variant = g_wino_dbg >> 2;
#else
  constexpr int variant = 0;
#endif
  switch (variant) {
    case 0:
      if (flags == (DF_CONV_BIAS | DF_CONV_LRELU)) hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      else if (flags == DF_CONV_MASK) hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_MASK>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      else if (flags == DF_CONV_RESIDUAL) hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_RESIDUAL>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      else if (flags == 0) hipLaunchKernelGGL((wino3d_kernel<0, 0>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      else hipLaunchKernelGGL((wino3d_kernel<0, -1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      break;
#ifdef DF_TUNING
    case 1: hipLaunchKernelGGL((wino3d_kernel<1, -1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 2: hipLaunchKernelGGL((wino3d_kernel<2, -1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 3: hipLaunchKernelGGL((wino3d_kernel<3, -1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 7: hipLaunchKernelGGL((wino3d_kernel<7, -1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 272: hipLaunchKernelGGL((wino3d_kernel<272, -1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 16:      // cycle profile (df_debug_wino_prof) of the SPECIALISED epilogues where they exist
      if (flags == (DF_CONV_BIAS | DF_CONV_LRELU)) hipLaunchKernelGGL((wino3d_kernel<16, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      else if (flags == DF_CONV_MASK) hipLaunchKernelGGL((wino3d_kernel<16, DF_CONV_MASK>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      else hipLaunchKernelGGL((wino3d_kernel<16, -1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
      break;
    // diagnosis variants (results are wrong by construction, timing only): 4 no staging at all | 8 no weight loads | 64 staging loads
    // kept alive but not written to LDS | 128 staging loads replaced by zeros | 256 staging loads read an always-cached address
    case 4: hipLaunchKernelGGL((wino3d_kernel<4, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 8: hipLaunchKernelGGL((wino3d_kernel<8, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 12: hipLaunchKernelGGL((wino3d_kernel<12, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 64: hipLaunchKernelGGL((wino3d_kernel<64, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 128: hipLaunchKernelGGL((wino3d_kernel<128, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 256: hipLaunchKernelGGL((wino3d_kernel<256, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 512: hipLaunchKernelGGL((wino3d_kernel<512, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 9: hipLaunchKernelGGL((wino3d_kernel<9, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 384: hipLaunchKernelGGL((wino3d_kernel<384, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;      // staging loads confined to a 1 MB window (L2-resident, L1 misses)
    case 100: hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;      // production kernel, compile-time flags (baseline of the experiments)
    case 4194304: hipLaunchKernelGGL((wino3d_kernel<4194304, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 1048576: hipLaunchKernelGGL((wino3d_kernel<1048576, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 2097152: hipLaunchKernelGGL((wino3d_kernel<2097152, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 262144: hipLaunchKernelGGL((wino3d_kernel<262144, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case 524288: hipLaunchKernelGGL((wino3d_kernel<524288, DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
    case (262144 | 4): hipLaunchKernelGGL((wino3d_kernel<(262144 | 4), DF_CONV_BIAS | DF_CONV_LRELU>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); break;
#endif
    default: return df::fail(DF_EINVAL, "df_wino_conv_fwd: unknown debug variant");
  }
  return df::launched("df_wino_conv_fwd");
}

int df_wino_upconv_fwd(const float* xc, const float* wp, const float* bias, float* y, int64_t B, int64_t Dc, int64_t Hc,
                       int64_t Wc, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream) {
  DF_REQUIRE(xc && wp && y, DF_EINVAL, "df_wino_upconv_fwd: null pointer");
  DF_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0, DF_EINVAL, "df_wino_upconv_fwd: non-positive extent");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE,
             "df_wino_upconv_fwd: Cin, Cout must be multiples of 32 (use df_upconv_fwd otherwise)");
  DF_REQUIRE(flags == (DF_CONV_BIAS | DF_CONV_LRELU) && bias, DF_EINVAL, "df_wino_upconv_fwd: flags must be DF_CONV_BIAS | DF_CONV_LRELU");
  DF_REQUIRE(8 * Dc * Hc * Wc * Cout <= (1LL << 29) && Dc * Hc * Wc * Cin <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino_upconv_fwd: one batch volume must stay below 2 GiB (use df_upconv_fwd)");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(xc) && df::aligned16(y), DF_EALIGN, "df_wino_upconv_fwd: xc, y and packed weights must be 16-byte aligned");
  WinoArgs a;
  a.x = xc; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = nullptr; a.mask_src = nullptr; a.y = y; a.y2 = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)(2 * Dc); a.H = (int)(2 * Hc); a.W = (int)(2 * Wc); a.Cin = (int)Cin; a.Cout = (int)Cout;      // OUTPUT (fine) extents
  a.nbz = (int)ceil_div(a.D, 4); a.nby = (int)ceil_div(a.H, 8); a.nbx = (int)ceil_div(a.W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino_upconv_fwd: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = flags; a.leak = leak;
  const int64_t grid = wino_grid(a, ntb);
#ifdef DF_TUNING      // round-5 breakdown of the 27-point form (tools/r05_breakdown_probe.py; results wrong by construction, timing only)
#define DF_WU(V) case V: hipLaunchKernelGGL((wino3d_kernel<V, DF_CONV_BIAS | DF_CONV_LRELU, 3>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); return df::launched("df_wino_upconv_fwd")
#define DF_WU1(V) case 1000 + V: hipLaunchKernelGGL((wino3d_kernel<V, DF_CONV_BIAS | DF_CONV_LRELU, 1>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); return df::launched("df_wino_upconv_fwd")
  switch (g_wino_dbg >> 2) {      // V: coarse-block staging (MODE 3, production); 1000 + V: the fine-grid staging of rounds 2-4 (MODE 1)
    DF_WU(1); DF_WU(2); DF_WU(3); DF_WU(4); DF_WU(7); DF_WU(8); DF_WU(64); DF_WU(128); DF_WU(512);
    DF_WU1(0); DF_WU1(1); DF_WU1(2); DF_WU1(3); DF_WU1(4); DF_WU1(7); DF_WU1(8); DF_WU1(64); DF_WU1(128); DF_WU1(512);
    default: break;
  }
#undef DF_WU
#undef DF_WU1
#endif
  hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_BIAS | DF_CONV_LRELU, 3>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  return df::launched("df_wino_upconv_fwd");
}

int64_t df_wino_signbits_bytes(int64_t B, int64_t D, int64_t H, int64_t W, int64_t C) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 32) return 0;
  return B * ceil_div(D, 4) * ceil_div(H, 8) * ceil_div(W, 8) * (C / 32) * kBitBytesPerBlock;
}

int df_wino_conv_fwd_bits(const float* x, const float* wp, const float* bias, const void* mask_bits, float* y, void* sign_bits, int64_t B,
                          int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flags, float leak, df_stream_t stream) {
  DF_REQUIRE(x && wp && y, DF_EINVAL, "df_wino_conv_fwd_bits: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DF_EINVAL, "df_wino_conv_fwd_bits: non-positive extent");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "df_wino_conv_fwd_bits: Cin, Cout must be multiples of 32");
  DF_REQUIRE(D * H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino_conv_fwd_bits: one batch volume must stay below 2 GiB");
  const bool fwd = flags == (DF_CONV_BIAS | DF_CONV_LRELU) && bias && sign_bits && !mask_bits;
  const bool dgr = flags == DF_CONV_MASK && mask_bits && !sign_bits;
  DF_REQUIRE(fwd || dgr, DF_EINVAL, "df_wino_conv_fwd_bits: either (BIAS|LRELU, bias, sign_bits out) or (MASK, mask_bits in)");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x) && df::aligned16(y) && df::aligned16(mask_bits) && df::aligned16(sign_bits), DF_EALIGN,
             "df_wino_conv_fwd_bits: x, y, packed weights and bit words must be 16-byte aligned");
  WinoArgs a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = nullptr; a.mask_src = nullptr; a.y = y; a.y2 = nullptr;
  a.bits_out = static_cast<unsigned char*>(sign_bits); a.bits_in = static_cast<const unsigned char*>(mask_bits);
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino_conv_fwd_bits: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = flags; a.leak = leak;
  const int64_t grid = wino_grid(a, ntb);
  if (fwd) hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_BIAS | DF_CONV_LRELU | kSignBits>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  else hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_MASK | kMaskBits>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  return df::launched("df_wino_conv_fwd_bits");
}

int df_wino_upconv_fwd_bits(const float* xc, const float* wp, const float* bias, float* y, void* sign_bits, int64_t B, int64_t Dc,
                            int64_t Hc, int64_t Wc, int64_t Cin, int64_t Cout, float leak, df_stream_t stream) {
  DF_REQUIRE(xc && wp && y && bias && sign_bits, DF_EINVAL, "df_wino_upconv_fwd_bits: null pointer");
  DF_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0, DF_EINVAL, "df_wino_upconv_fwd_bits: non-positive extent");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "df_wino_upconv_fwd_bits: Cin, Cout must be multiples of 32");
  DF_REQUIRE(8 * Dc * Hc * Wc * Cout <= (1LL << 29) && Dc * Hc * Wc * Cin <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino_upconv_fwd_bits: one batch volume must stay below 2 GiB");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(xc) && df::aligned16(y) && df::aligned16(sign_bits), DF_EALIGN,
             "df_wino_upconv_fwd_bits: xc, y, packed weights and bit words must be 16-byte aligned");
  WinoArgs a;
  a.x = xc; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = nullptr; a.mask_src = nullptr; a.y = y; a.y2 = nullptr;
  a.bits_out = static_cast<unsigned char*>(sign_bits); a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)(2 * Dc); a.H = (int)(2 * Hc); a.W = (int)(2 * Wc); a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(a.D, 4); a.nby = (int)ceil_div(a.H, 8); a.nbx = (int)ceil_div(a.W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino_upconv_fwd_bits: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = DF_CONV_BIAS | DF_CONV_LRELU; a.leak = leak;
  const int64_t grid = wino_grid(a, ntb);
  hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_BIAS | DF_CONV_LRELU | kSignBits, 3>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  return df::launched("df_wino_upconv_fwd_bits");
}

int df_wino_upconv_dgrad(const float* g, const float* wp, float* acc, int64_t B, int64_t Dc, int64_t Hc, int64_t Wc, int64_t Cin,
                         int64_t Cout, df_stream_t stream) {
  DF_REQUIRE(g && wp && acc, DF_EINVAL, "df_wino_upconv_dgrad: null pointer");
  DF_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0, DF_EINVAL, "df_wino_upconv_dgrad: non-positive extent");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE,
             "df_wino_upconv_dgrad: Cin, Cout must be multiples of 32 (use df_upconv_dgrad otherwise)");
  DF_REQUIRE(8 * Dc * Hc * Wc * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino_upconv_dgrad: one batch volume must stay below 2 GiB (use df_upconv_dgrad)");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(g), DF_EALIGN, "df_wino_upconv_dgrad: g and packed weights must be 16-byte aligned");
  WinoArgs a;      // the adjoint conv reads g (Cout channels, fine grid) and produces Cin channels
  a.x = g; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = nullptr; a.residual = nullptr; a.mask_src = nullptr; a.y = acc; a.y2 = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)(2 * Dc); a.H = (int)(2 * Hc); a.W = (int)(2 * Wc); a.Cin = (int)Cout; a.Cout = (int)Cin;
  a.nbz = (int)ceil_div(a.D, 4); a.nby = (int)ceil_div(a.H, 8); a.nbx = (int)ceil_div(a.W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cin / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino_upconv_dgrad: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = 0; a.leak = 0.f;
  const int64_t grid = wino_grid(a, ntb);
#ifdef DF_TUNING
#define DF_WP(V) case V: hipLaunchKernelGGL((wino3d_kernel<V, 0, 2>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a); return df::launched("df_wino_upconv_dgrad")
  switch (g_wino_dbg >> 2) {
    DF_WP(1); DF_WP(2); DF_WP(3); DF_WP(4); DF_WP(7); DF_WP(8); DF_WP(64); DF_WP(128);
    default: break;
  }
#undef DF_WP
#endif
  hipLaunchKernelGGL((wino3d_kernel<0, 0, 2>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  return df::launched("df_wino_upconv_dgrad");
}

int df_wino_conv_fwd_addup(const float* x, const float* wp, const float* bias, const float* xc, float* y, float* y2, int64_t B,
                           int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, float leak, df_stream_t stream) {
  DF_REQUIRE(x && wp && bias && xc && y && y2, DF_EINVAL, "df_wino_conv_fwd_addup: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, DF_EINVAL,
             "df_wino_conv_fwd_addup: extents must be positive and even (the output of a 2x up-sampling block)");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "df_wino_conv_fwd_addup: Cin, Cout must be multiples of 32");
  DF_REQUIRE(D * H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino_conv_fwd_addup: one batch volume must stay below 2 GiB");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x), DF_EALIGN, "df_wino_conv_fwd_addup: x and packed weights must be 16-byte aligned");
  WinoArgs a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = xc; a.mask_src = nullptr; a.y = y; a.y2 = y2; a.bits_out = nullptr; a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino_conv_fwd_addup: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP; a.leak = leak;
  const int64_t grid = wino_grid(a, ntb);
  hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP>), dim3((unsigned)grid), dim3(kT), 0, df::as_stream(stream), a);
  return df::launched("df_wino_conv_fwd_addup");
}

int df_wino_conv_fwd_addup_bits(const float* x, const float* wp, const float* bias, const float* xc, float* y2, void* sign_bits,
                                int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, float leak, df_stream_t stream) {
  DF_REQUIRE(x && wp && bias && xc && y2 && sign_bits, DF_EINVAL, "df_wino_conv_fwd_addup_bits: null pointer");
  DF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, DF_EINVAL,
             "df_wino_conv_fwd_addup_bits: extents must be positive and even (the output of a 2x up-sampling block)");
  DF_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DF_ESHAPE, "df_wino_conv_fwd_addup_bits: Cin, Cout must be multiples of 32");
  DF_REQUIRE(D * H * W * (Cin > Cout ? Cin : Cout) <= (1LL << 29) && Cin * Cout <= (1LL << 24), DF_ESHAPE,
             "df_wino_conv_fwd_addup_bits: one batch volume must stay below 2 GiB");
  DF_REQUIRE(df::aligned16(wp) && df::aligned16(x) && df::aligned16(sign_bits), DF_EALIGN,
             "df_wino_conv_fwd_addup_bits: x, packed weights and bit words must be 16-byte aligned");
  WinoArgs a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.zeros = wp + 64 * Cin * Cout;
  a.bias = bias; a.residual = xc; a.mask_src = nullptr; a.y = nullptr; a.y2 = y2;
  a.bits_out = static_cast<unsigned char*>(sign_bits); a.bits_in = nullptr;
  a.B = (int)B; a.D = (int)D; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout;
  a.nbz = (int)ceil_div(D, 4); a.nby = (int)ceil_div(H, 8); a.nbx = (int)ceil_div(W, 8);
  const int64_t ntb = B * a.nbz * a.nby * a.nbx;
  a.ncs = (int)(Cout / 32);
  DF_REQUIRE(ntb * a.ncs < (1LL << 31), DF_ESHAPE, "df_wino_conv_fwd_addup_bits: too many workgroups");
  a.ntb = (int)ntb;
  a.flags = DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP; a.leak = leak;
  const int64_t grid = wino_grid(a, ntb);
  hipLaunchKernelGGL((wino3d_kernel<0, DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_ADDUP | kSignBits | kNoPrimary>), dim3((unsigned)grid), dim3(kT), 0,
                     df::as_stream(stream), a);
  return df::launched("df_wino_conv_fwd_addup_bits");
}

int df_lrelu_bits_bwd_pool2x(const float* gy, const void* mask_bits, float* gx, float* gpool, float leak, int64_t B, int64_t Dc, int64_t Hc,
                             int64_t Wc, int64_t C, df_stream_t stream) {
  DF_REQUIRE(gy && mask_bits && gx && gpool, DF_EINVAL, "df_lrelu_bits_bwd_pool2x: null pointer");
  DF_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0, DF_EINVAL, "df_lrelu_bits_bwd_pool2x: non-positive extent");
  DF_REQUIRE(C > 0 && C % 32 == 0, DF_ESHAPE, "df_lrelu_bits_bwd_pool2x: C must be a multiple of 32 (the sign bits' cout slices)");
  DF_REQUIRE(df::aligned16(gy) && df::aligned16(gx) && df::aligned16(gpool) && df::aligned16(mask_bits), DF_EALIGN,
             "df_lrelu_bits_bwd_pool2x: 16-byte alignment");
  const int64_t n4 = B * Dc * Hc * Wc * (C / 4);
  DF_REQUIRE(ceil_div(n4, 256) < (1LL << 31), DF_ESHAPE, "df_lrelu_bits_bwd_pool2x: tensor too large");
  hipLaunchKernelGGL(lrelu_bits_bwd_pool_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, df::as_stream(stream),
                     reinterpret_cast<const float4*>(gy), static_cast<const unsigned*>(mask_bits), reinterpret_cast<float4*>(gx),
                     reinterpret_cast<float4*>(gpool), leak, n4, (int)Dc, (int)Hc, (int)Wc, (int)(C / 4), (int)ceil_div(2 * Dc, 4),
                     (int)ceil_div(2 * Hc, 8), (int)ceil_div(2 * Wc, 8), (int)(C / 32));
  return df::launched("df_lrelu_bits_bwd_pool2x");
}

}  // extern "C"
