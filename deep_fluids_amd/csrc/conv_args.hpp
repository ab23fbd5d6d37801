// Shared argument block / constants of the convolution kernels (conv.hip, conv_small.hip).
#pragma once
#include "df_common.hpp"

namespace dfconv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int CK = 16;          // input channels per LDS chunk
constexpr int LDS_STRIDE = 20;  // floats per staged voxel (16 + 4 pad)

struct ConvArgs {
  const float* x;
  const f32x4* wp;
  const float* bias;
  const float* residual;
  const float* mask_src;
  float* y;
  int B, D, H, W, Cin, Cout;   // D,H,W: OUTPUT extents
  int Di, Hi, Wi;              // LOGICAL input extents the taps slide over (== output for stride 1; 2x for stride 2)
  // generalisation used by the up-sampling-aware first conv of a generator block (conv on the NN-upsampled
  // input == 8 parity-class 2x2x2-tap convs on the coarse grid) and its dgrad; identity values for a plain conv:
  int pz, py, px;              // logical input coordinate of tap 0 = out*S - p   (plain k=3: 1; stride 2: 0)
  int is, iz, iy, ix;          // physical input voxel  = logical*is + i?  (gather stride / offset)
  int xD, xH, xW;              // physical extents of the x tensor (addressing)
  int os, oz, oy, ox;          // physical output voxel = logical*os + o?  (scatter stride / offset)
  int yD, yH, yW;              // physical extents of y / residual / mask_src
  int nclass;                  // > 1: blockIdx.z = parity class c; p = 1 - bit(c), o = bit(c), weights += c*wclass
  int64_t wclass;              // packed-weight stride between classes (float4 units)
  int Kpad, Npad;       // padded K (multiple of 16) and N (multiple of the N tile) of the packed weights
  int nz, ny, nx;       // tiles per axis
  int ntiles;
  int flags;
  float leak;
};


// XCD-aware, bijective workgroup -> tile mapping (workgroup b runs on XCD b % 8; speed only)
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
  const int q = ntiles >> 3, rem = ntiles & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

// VALU kernels for the generator's last layer (conv_small.hip); return DF_OK or an error
int launch_small_n(const ConvArgs& a, int kz, hipStream_t s);   // Cout <= 4  (128 -> 3 | 1)
int launch_small_k(const ConvArgs& a, int kz, hipStream_t s);   // Cin  <= 4  (dgrad of the last layer; 3 -> F)

// bf16x3 split-precision MFMA path (conv_bf16.hip)
bool bf16x3_supported(const ConvArgs& a);
int64_t bf16x3_kpad(int64_t K);
int launch_bf16x3(const ConvArgs& a, int kz, int kt, hipStream_t s);
int pack_bf16x3(const float* w, void* wp, int taps, int cin, int cout, int Kpad, int Npad, int mode, hipStream_t s);
int upconv_pack_bf16x3(const float* w, void* wp, int kz, int cin, int cout, int Kpad, int Npad, int mode, hipStream_t s);

}  // namespace dfconv
