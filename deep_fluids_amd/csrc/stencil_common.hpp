// Shared pieces of the HBM-bound stencil kernels (stencil.hip, velocity_loss.hip): block geometry, the XCD-aware block remap and the
// LDS -> global flush that turns per-thread results into 1 KiB-contiguous stores.
#ifndef DF_STENCIL_COMMON_HPP
#define DF_STENCIL_COMMON_HPP
#include "df_common.hpp"

namespace dfst {


constexpr int kThreads = 256;
constexpr int kVoxPerThread = 4;
constexpr int kVoxPerBlock = kThreads * kVoxPerThread;   // 1024

struct Dims3 {
  int64_t nvox;   // B*Z*Y*X
  int Z, Y, X;
  int group;      // XCD remap granularity (blocks); 0 = one contiguous chunk per XCD
};

// XCD-aware, bijective block remap: workgroup b runs on XCD b % 8 (observed); hand each XCD a contiguous run of
// blocks so the y/z neighbour records a block re-reads were fetched into the SAME XCD's L2 by its own neighbours.
// Speed only (measured: halves FETCH_SIZE of jacobian3d_fwd), never correctness.
__device__ __forceinline__ int64_t xcd_block(int bid, int nblk, int group) {
  const int xcd = bid & 7, idx = bid >> 3;
  if (group > 0) {                  // runs of `group` consecutive blocks dealt round-robin to the XCDs (nblk % (8*group) == 0)
    return (static_cast<int64_t>(idx / group) * 8 + xcd) * group + idx % group;
  }
  const int q = nblk >> 3, rem = nblk & 7;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

// copy `nfloats` floats from LDS (16-byte aligned) to global `dst` (16-byte aligned base) with
// 16-byte stores; the ragged tail (only in the last workgroup) falls back to dword stores.
template <bool NT>
__device__ __forceinline__ void flush_lds(const float* __restrict__ s, float* __restrict__ dst, int64_t nfloats,
                                          int tid) {
  const int64_t nq = nfloats >> 2;
  const float4* s4 = reinterpret_cast<const float4*>(s);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int64_t q = tid; q < nq; q += kThreads) {
    if (NT) {
      typedef float v4 __attribute__((ext_vector_type(4)));
      const v4 val = reinterpret_cast<const v4*>(s)[q];
      __builtin_nontemporal_store(val, reinterpret_cast<v4*>(dst) + q);
    } else {
      d4[q] = s4[q];
    }
  }
  const int64_t done = nq << 2;
  if (tid < nfloats - done) dst[done + tid] = s[done + tid];
}


typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kXcdGroup = 48;      // runs of 48 blocks per XCD (sweep in tools/stencil_probe.py: best warm + cold)

}  // namespace dfst
#endif
