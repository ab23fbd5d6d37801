// Shared host-side plumbing of libdeepfluids_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/deepfluids_hip.h"

namespace df {

// thread-local last-error message (df_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(df_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// check the launch that was just enqueued; never synchronises
inline int launched(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
  return DF_OK;
}

// LDS a single workgroup may opt in to on the current device (hipDeviceAttributeSharedMemPerBlockOptin; 160 KiB on an unpartitioned
// MI355X), queried once per process.  Kernels that size themselves for it fall back to their smaller-footprint path when it is less.
int64_t lds_optin_bytes();

// Clear of a small workspace region as a KERNEL node.  Not hipMemsetAsync: inside a captured hipGraph (Trainer(graph=True)) ROCm 7.2 runs
// memset nodes outside the order the capture recorded -- a replay that starts on an idle device executed them late, over memory the
// graph's pool had meanwhile handed to another tensor (profiles/r06_probes.md, section 1).  `bytes` must be a multiple of 4.
static __global__ void zero_words_kernel(uint32_t* __restrict__ p, int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = 0u;
}
inline hipError_t zero_async(void* p, int64_t bytes, hipStream_t s) {
  const int64_t n = bytes / 4;
  if (n <= 0) return hipSuccess;
  const unsigned blocks = static_cast<unsigned>((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, s, static_cast<uint32_t*>(p), n);
  return hipGetLastError();
}

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCUs = 256;          // MI355X

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace df

#define DF_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return df::fail(code, __VA_ARGS__); \
  } while (0)
