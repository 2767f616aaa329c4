// Host-side error slot + version for libpmce_hip.so (no global mutable state besides a thread-local string and the thread-local
// overflow / clock-probe sinks of the call in progress).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"

static thread_local char g_err[512] = "";
static thread_local unsigned* g_overflow_sink = nullptr;
static thread_local unsigned long long* g_clock_sink = nullptr;

unsigned* pmce_overflow_sink(void) { return g_overflow_sink; }
void pmce_set_overflow_sink(unsigned* w) { g_overflow_sink = w; }
unsigned long long* pmce_clock_sink(void) { return g_clock_sink; }
void pmce_set_clock_sink(unsigned long long* w) { g_clock_sink = w; }

void pmce_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int pmce_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pmce_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return PMCE_ERR_LAUNCH;
  }
  return PMCE_OK;
}

int pmce_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

extern "C" const char* pmce_last_error_string(void) { return g_err; }
extern "C" int pmce_version(void) { return 100; }  // 0.1.0
// sha256 (first 16 hex digits) over the sources this library was built from (pmce_amd/build.py source_id(), passed as a macro): what the
// committed profiler summaries under profiles/ record, so that bench.py can tell counters of THIS build from stale ones
#ifndef PMCE_BUILD_ID
#define PMCE_BUILD_ID "unknown"
#endif
extern "C" const char* pmce_build_id(void) { return PMCE_BUILD_ID; }
