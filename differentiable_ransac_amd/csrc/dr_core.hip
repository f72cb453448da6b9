// Library-level entry points: version and per-thread error string.
#include <stdarg.h>

#include "dr_common.hpp"

namespace dr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dr

extern "C" {
int dr_version(void) { return DR_ABI_VERSION; }
const char *dr_last_error(void) { return dr::g_err; }
}
