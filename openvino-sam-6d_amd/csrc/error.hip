// Error text plumbing + ABI version for libsam6d_hip.so.
#include "common.h"
#include "../../include/sam6d_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void sam6d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sam6d_last_error(void) { return g_err; }
extern "C" int sam6d_abi_version(void) { return SAM6D_ABI_VERSION; }
