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

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCUs = 256;          // MI355X

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace df

#define DF_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return df::fail(code, __VA_ARGS__); \
  } while (0)
