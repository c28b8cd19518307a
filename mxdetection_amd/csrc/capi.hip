// capi.hip -- error string and version entry points of libmxdet_hip.so.
#include <stdarg.h>

#include "common.h"

namespace mxdet {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
void clear_error() { g_err[0] = 0; }

}  // namespace mxdet

extern "C" const char* mxdet_last_error(void) { return mxdet::g_err; }
extern "C" const char* mxdet_version(void) { return "mxdet-hip 0.1 gfx950"; }
