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

// plan-time thresholds, index = MXDET_TUNE_* (include/mxdet_debug.h)
static const long long kTuneDefault[MXDET_TUNE_COUNT] = {400, 1536, 1600, 3072, 64, 128, 1, 1536, 32, 2, 2, 2, 0, 2, 1, 1000000, 1, 0, 192};
static long long g_tune[MXDET_TUNE_COUNT] = {400, 1536, 1600, 3072, 64, 128, 1, 1536, 32, 2, 2, 2, 0, 2, 1, 1000000, 1, 0, 192};
long long tuning(int which) { return g_tune[which]; }

}  // namespace mxdet

extern "C" const char* mxdet_last_error(void) { return mxdet::g_err; }
extern "C" const char* mxdet_version(void) { return "mxdet-hip 0.2 gfx950"; }

extern "C" int mxdet_debug_set_tuning(int32_t which, int64_t value) {
  mxdet::clear_error();
  MXDET_REQUIRE(which >= 0 && which < MXDET_TUNE_COUNT, MXDET_EINVAL, "debug_set_tuning: unknown key %d", which);
  mxdet::g_tune[which] = value < 0 ? mxdet::kTuneDefault[which] : (long long)value;
  return MXDET_OK;
}
