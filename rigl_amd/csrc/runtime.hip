// Error channel, version, and the optional HIP-event profiler of the C ABI.
#include <ctype.h>
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>
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

// ---- development knobs ------------------------------------------------------
struct Knob { char key[32]; int value; bool set; bool env_set; int env_value; };
static std::mutex g_tune_mu;
static std::vector<Knob> g_knobs;
// Bumped by every rigl_tune_set / rigl_tune_unset: call sites keep (generation, value) in a static TuneSite and only
// take the mutex and scan the table again after a change (the scan ran several times per conv / BN launch).
static std::atomic<uint64_t> g_tune_gen{1};

uint64_t tune_generation() { return g_tune_gen.load(std::memory_order_acquire); }

static Knob* find_or_add_knob(const char* key) {
  for (Knob& k : g_knobs)
    if (strncmp(k.key, key, sizeof(k.key)) == 0) return &k;
  Knob k;
  memset(&k, 0, sizeof(k));
  strncpy(k.key, key, sizeof(k.key) - 1);
  // A knob never set by rigl_tune_set reads the environment variable RIGL_<KEY IN UPPER CASE> once (so that
  // subprocess-per-setting test runners can select kernels the way they always did), else the caller's default.
  char env[48] = "RIGL_";
  size_t n = 5;
  for (const char* c = key; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
  env[n] = 0;
  const char* e = getenv(env);
  k.env_set = e != nullptr && *e != 0;
  k.env_value = k.env_set ? atoi(e) : 0;
  g_knobs.push_back(k);
  return &g_knobs.back();
}

int tune_get(const char* key, int dflt) {
  std::lock_guard<std::mutex> l(g_tune_mu);
  const Knob* k = find_or_add_knob(key);
  return k->set ? k->value : (k->env_set ? k->env_value : dflt);
}

int tune_cached(TuneSite& site, const char* key, int dflt) {
  const uint64_t g = tune_generation();
  if (site.gen.load(std::memory_order_acquire) == g) return site.value.load(std::memory_order_relaxed);
  const int v = tune_get(key, dflt);
  site.value.store(v, std::memory_order_relaxed);
  site.gen.store(g, std::memory_order_release);
  return v;
}

// ---- profiler ---------------------------------------------------------------
// Event pairs are recorded on the launch stream, so the measured interval is
// the device-side duration of exactly the kernels launched in between.
struct EvPair { hipEvent_t a, b; int kind; int tag[6]; };
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

hipEvent_t prof_get_event() { return get_event(); }
static thread_local int t_tag[6] = {0, 0, 0, 0, 0, 0};
void prof_set_tag(const RiglConvDesc* d) {
  if (d) { t_tag[0] = d->h; t_tag[1] = d->w; t_tag[2] = d->cin; t_tag[3] = d->cout; t_tag[4] = d->kh; t_tag[5] = d->stride_h; }
  else for (int i = 0; i < 6; ++i) t_tag[i] = 0;
}
static EvPair make_pair(int kind, hipEvent_t a, hipEvent_t b) {
  EvPair p;
  p.a = a; p.b = b; p.kind = kind;
  const bool conv = kind == PROF_CONV_FWD || kind == PROF_CONV_DGRAD || kind == PROF_CONV_WGRAD || kind == PROF_CONV_BWD ||
                    kind == PROF_DEPTHWISE;
  for (int i = 0; i < 6; ++i) p.tag[i] = conv ? t_tag[i] : 0;
  return p;
}
void prof_add_pair(int kind, hipEvent_t a, hipEvent_t b) {
  std::lock_guard<std::mutex> l(g_prof_mu);
  g_pending.push_back(make_pair(kind, a, b));
}
int& prof_current_kind() {
  static thread_local int k = PROF_CONV_FWD;
  return k;
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
  g_pending.push_back(make_pair(kind, a, b));
}

}  // namespace rigl

extern "C" {

int rigl_version(void) { return RIGL_ABI_VERSION; }

// CRC-32C (Castagnoli, reflected 0x82F63B78), slicing-by-8, host only: the
// checksum of TensorFlow's checkpoint bundles (rigl_amd/tf_checkpoint.py).
uint32_t rigl_crc32c(const void* data, size_t n, uint32_t crc) {
  static uint32_t T[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
      T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFFu];
    ready = true;
  }
  const unsigned char* p = static_cast<const unsigned char*>(data);
  crc = ~crc;
  while (n >= 8) {
    const uint32_t lo = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    const uint32_t a = crc ^ lo;
    crc = T[7][a & 0xFFu] ^ T[6][(a >> 8) & 0xFFu] ^ T[5][(a >> 16) & 0xFFu] ^ T[4][a >> 24] ^
          T[3][p[4]] ^ T[2][p[5]] ^ T[1][p[6]] ^ T[0][p[7]];
    p += 8; n -= 8;
  }
  while (n--) crc = (crc >> 8) ^ T[0][(crc ^ *p++) & 0xFFu];
  return ~crc;
}

const char* rigl_last_error(void) { return rigl::g_err; }

int rigl_tune_set(const char* key, int32_t value) {
  if (!key || !*key || strlen(key) >= sizeof(rigl::Knob::key)) return rigl::fail(RIGL_EINVAL, "rigl_tune_set: bad key");
  if (value == INT32_MIN) return rigl_tune_unset(key);
  std::lock_guard<std::mutex> l(rigl::g_tune_mu);
  rigl::Knob* k = rigl::find_or_add_knob(key);
  k->value = value;
  k->set = true;
  rigl::g_tune_gen.fetch_add(1, std::memory_order_acq_rel);
  return RIGL_OK;
}

// Back to "never set": the knob reads its RIGL_<KEY> environment variable again, else every call site's own default.
int rigl_tune_unset(const char* key) {
  if (!key || !*key || strlen(key) >= sizeof(rigl::Knob::key)) return rigl::fail(RIGL_EINVAL, "rigl_tune_unset: bad key");
  std::lock_guard<std::mutex> l(rigl::g_tune_mu);
  rigl::Knob* k = rigl::find_or_add_knob(key);
  k->set = false;
  rigl::g_tune_gen.fetch_add(1, std::memory_order_acq_rel);
  return RIGL_OK;
}

int32_t rigl_tune_get(const char* key, int32_t dflt) { return key ? rigl::tune_get(key, dflt) : dflt; }

uint64_t rigl_tune_generation(void) { return rigl::tune_generation(); }
const volatile uint64_t* rigl_tune_generation_addr(void) {
  static_assert(sizeof(rigl::g_tune_gen) == sizeof(uint64_t), "the counter is a plain 64-bit word");
  return reinterpret_cast<const volatile uint64_t*>(&rigl::g_tune_gen);
}

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

int rigl_prof_collect_launches(RiglProfLaunch* out, int64_t cap, int64_t* n_launches) {
  if (!n_launches || (cap > 0 && !out)) return rigl::fail(RIGL_EINVAL, "rigl_prof_collect_launches: NULL output");
  std::vector<rigl::EvPair> pend;
  {
    std::lock_guard<std::mutex> l(rigl::g_prof_mu);
    pend.swap(rigl::g_pending);
  }
  int64_t n = 0;
  for (auto& p : pend) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      if (n < cap) {
        out[n].kind = p.kind;
        for (int i = 0; i < 6; ++i) out[n].tag[i] = p.tag[i];
        out[n].ms = ms;
      }
      ++n;
    }
    std::lock_guard<std::mutex> l(rigl::g_prof_mu);
    rigl::g_pool.push_back(p.a);
    rigl::g_pool.push_back(p.b);
  }
  *n_launches = n;
  return RIGL_OK;
}

}  // extern "C"
