// Error plumbing shared by every translation unit of libmudiff_hip.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mudiff_hip.h"

static thread_local char g_err[512] = "";

extern "C" void mud_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mud_last_error(void) { return g_err; }
extern "C" int mud_version(void) { return 110; }
// The shipped build carries no experiment flags; scripts/build_variants.py compiles every translation unit of a variant with
// -DMUD_BUILD_FLAGS="\"...\"" so that a library can always be asked what it is.
#ifndef MUD_BUILD_FLAGS
#define MUD_BUILD_FLAGS ""
#endif
extern "C" const char* mud_build_flags(void) { return MUD_BUILD_FLAGS; }
