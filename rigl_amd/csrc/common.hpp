// Shared host-side helpers for the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rigl_hip.h"

namespace rigl {

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

// Process-wide development knobs (rigl_tune_set): the run-time twin of the RIGL_* environment variables, for
// A/B runs inside one process.  Unknown keys read as the caller's default.
int tune_get(const char* key, int dflt);
// Per-call-site cache of a knob: valid until the next rigl_tune_set / rigl_tune_unset (a generation counter), so the
// launch paths read their knobs with two atomic loads instead of a mutex and a string scan.  Use through RIGL_TUNE.
struct TuneSite { std::atomic<uint64_t> gen{0}; std::atomic<int> value{0}; };
uint64_t tune_generation();
int tune_cached(TuneSite& site, const char* key, int dflt);
#define RIGL_TUNE(key, dflt) ([]() -> int { static ::rigl::TuneSite site__; return ::rigl::tune_cached(site__, key, dflt); }())

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
                PROF_PRUNE_REGROW = 3, PROF_SGD = 4, PROF_PACK = 5, PROF_CONV_BWD = 6, PROF_DEPTHWISE = 7 };
bool prof_enabled();
void prof_begin(int kind, hipStream_t s);
void prof_end(int kind, hipStream_t s);

// Per-kernel timing without extra queue packets: while profiling is on, K1 kernels are launched with
// hipExtLaunchKernelGGL, whose dispatch packet itself stamps a start / stop event pair (the
// hipEventRecord pairs of ProfScope put two barrier packets around every launch: measured 0.66 ms per
// ResNet-50 step for the 165 K1 launches).  The kernel family is the thread's current ProfFamily.
hipEvent_t prof_get_event();
void prof_add_pair(int kind, hipEvent_t a, hipEvent_t b);
int& prof_current_kind();
void prof_set_tag(const RiglConvDesc* d);      // the conv descriptor the following K1 launches of this thread belong to
struct ProfFamily {
  int prev;
  explicit ProfFamily(int k) : prev(prof_current_kind()) { prof_current_kind() = k; }
  ~ProfFamily() { prof_current_kind() = prev; }
};

struct ProfScope {
  int kind; hipStream_t s; bool on;
  ProfScope(int k, hipStream_t st) : kind(k), s(st), on(prof_enabled()) { if (on) prof_begin(kind, s); }
  ~ProfScope() { if (on) prof_end(kind, s); }
};

template <typename F, typename... Args>
inline void prof_launch(F kernel, dim3 grid, dim3 blk, unsigned lds, hipStream_t st, Args... args) {
  hipEvent_t a = prof_get_event(), b = prof_get_event();
  if (!a || !b) { hipLaunchKernelGGL(kernel, grid, blk, lds, st, args...); return; }
  hipExtLaunchKernelGGL(kernel, grid, blk, lds, st, a, b, 0, args...);
  prof_add_pair(prof_current_kind(), a, b);
}
#define RIGL_K_LAUNCH(kernel, grid, blk, lds, st, ...)                                      \
  do {                                                                                      \
    if (::rigl::prof_enabled()) ::rigl::prof_launch(kernel, grid, blk, lds, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, blk, lds, st, __VA_ARGS__);                       \
  } while (0)

inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace rigl
