// Error channel, version, and the optional HIP-event profiler of the C ABI.
#include <mutex>
#include <vector>

#include "common.hpp"

namespace rigl {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- profiler ---------------------------------------------------------------
// Event pairs are recorded on the launch stream, so the measured interval is
// the device-side duration of exactly the kernels launched in between.
struct EvPair { hipEvent_t a, b; int kind; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<EvPair> g_pending;
static std::vector<hipEvent_t> g_pool;
static thread_local hipEvent_t t_open[RIGL_PROF_KINDS];

bool prof_enabled() { return g_prof_on; }

static hipEvent_t get_event() {
  std::lock_guard<std::mutex> l(g_prof_mu);
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void prof_begin(int kind, hipStream_t s) {
  hipEvent_t e = get_event();
  t_open[kind] = e;
  if (e) (void)hipEventRecord(e, s);
}

void prof_end(int kind, hipStream_t s) {
  hipEvent_t a = t_open[kind];
  if (!a) return;
  hipEvent_t b = get_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  std::lock_guard<std::mutex> l(g_prof_mu);
  g_pending.push_back({a, b, kind});
}

}  // namespace rigl

extern "C" {

int rigl_version(void) { return RIGL_ABI_VERSION; }

const char* rigl_last_error(void) { return rigl::g_err; }

int rigl_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> l(rigl::g_prof_mu);
  rigl::g_prof_on = on != 0;
  return RIGL_OK;
}

int rigl_prof_collect(double* ms_per_kind, int64_t* launches) {
  if (!ms_per_kind || !launches) return rigl::fail(RIGL_EINVAL, "rigl_prof_collect: NULL output");
  for (int i = 0; i < RIGL_PROF_KINDS; ++i) { ms_per_kind[i] = 0.0; launches[i] = 0; }
  std::vector<rigl::EvPair> pend;
  {
    std::lock_guard<std::mutex> l(rigl::g_prof_mu);
    pend.swap(rigl::g_pending);
  }
  for (auto& p : pend) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ms_per_kind[p.kind] += ms;
      launches[p.kind] += 1;
    }
    std::lock_guard<std::mutex> l(rigl::g_prof_mu);
    rigl::g_pool.push_back(p.a);
    rigl::g_pool.push_back(p.b);
  }
  return RIGL_OK;
}

}  // extern "C"
