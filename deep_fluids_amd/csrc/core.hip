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

}  // namespace df

extern "C" {
int df_version(void) { return DF_VERSION; }
const char* df_last_error(void) { return df::err_buf(); }
}
