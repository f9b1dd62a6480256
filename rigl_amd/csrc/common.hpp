// Shared host-side helpers for the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rigl_hip.h"

namespace rigl {

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(rigl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the last launch.  hipGetLastError is cheap and does not synchronise.
#define RIGL_CHECK_LAUNCH(what)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess)                                                   \
      return ::rigl::fail(RIGL_ELAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
  } while (0)

#define RIGL_HIP(call)                                                       \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess)                                                   \
      return ::rigl::fail(RIGL_ELAUNCH, "%s: %s", #call, hipGetErrorString(e__)); \
  } while (0)

// ---- optional per-launch timing (rigl_prof_*) -------------------------------
enum ProfKind { PROF_CONV_FWD = 0, PROF_CONV_DGRAD = 1, PROF_CONV_WGRAD = 2,
                PROF_PRUNE_REGROW = 3, PROF_SGD = 4, PROF_PACK = 5, PROF_CONV_BWD = 6 };
bool prof_enabled();
void prof_begin(int kind, hipStream_t s);
void prof_end(int kind, hipStream_t s);

struct ProfScope {
  int kind; hipStream_t s; bool on;
  ProfScope(int k, hipStream_t st) : kind(k), s(st), on(prof_enabled()) { if (on) prof_begin(kind, s); }
  ~ProfScope() { if (on) prof_end(kind, s); }
};

inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace rigl
