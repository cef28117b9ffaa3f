// Error plumbing of the C ABI (see include/simt_hip.h).
#include "common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void simt_set_error(const char* file, int line, const char* msg) {
  const char* base = strrchr(file, '/');
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", base ? base + 1 : file, line, msg);
}

extern "C" const char* simt_last_error(void) { return g_err; }
extern "C" int simt_abi_version(void) { return 1; }
