// Error reporting + version of libdeepfluids_hip.so.
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

int64_t lds_optin_bytes() {
  static const int64_t bytes = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess || v <= 0) {
      (void)hipGetLastError();
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) {
        (void)hipGetLastError();
        v = 64 * 1024;
      }
    }
    return static_cast<int64_t>(v);
  }();
  return bytes;
}

}  // namespace df

extern "C" {
int df_version(void) { return DF_VERSION; }
const char* df_last_error(void) { return df::err_buf(); }
}
