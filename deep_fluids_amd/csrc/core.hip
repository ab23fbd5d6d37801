// Error reporting + version of libdeepfluids_hip.so.
#include <atomic>
#include <cstdio>
#include "df_common.hpp"

namespace df {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// LDS a workgroup may opt in to on the CURRENT device (160 KiB per CU on an unpartitioned MI355X).  Cached per device id (a process may
// drive several GPUs); when the opt-in attribute is not reported the fast paths that need > 64 KiB (thin-K MFMA conv, thin weight gradient,
// one-kernel velocity-loss tiles) are gated off by their callers -- said ONCE on stderr, because that is a large silent slow-down otherwise.
int64_t lds_optin_bytes() {
  constexpr int kMaxDev = 64;
  static std::atomic<int64_t> cache[kMaxDev];      // 0 = not asked yet
  static std::atomic<bool> warned{false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  const bool cached = dev >= 0 && dev < kMaxDev;
  if (cached) {
    const int64_t c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess || v <= 0) {
    (void)hipGetLastError();
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) {
      (void)hipGetLastError();
      v = 64 * 1024;
    }
    if (!warned.exchange(true))
      fprintf(stderr, "libdeepfluids_hip: device %d does not report an LDS opt-in size; assuming %d bytes per workgroup -- the kernels that need "
                      "more (thin-layer MFMA paths, one-kernel velocity-loss tiles) fall back to their slower general forms\n", dev, v);
  }
  if (cached) cache[dev].store(static_cast<int64_t>(v), std::memory_order_relaxed);
  return static_cast<int64_t>(v);
}

}  // namespace df

extern "C" {
int df_version(void) { return DF_VERSION; }
const char* df_last_error(void) { return df::err_buf(); }
}
