#include <stdarg.h>
#include <string.h>

#include "smos_common.h"

namespace smos {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace smos

extern "C" int smos_abi_version(void) { return SMOS_ABI_VERSION; }
extern "C" const char* smos_last_error(void) { return smos::g_err; }
