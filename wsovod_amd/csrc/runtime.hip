// Error reporting + hipEvent-based per-kernel profiling table of the wsovod_hip C-ABI.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace wsovod {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------
// Profiling table.  Slots are registered lazily by name; ProfScope brackets a launch
// with two events on the launch stream and parks the pair until collect().
// ---------------------------------------------------------------------------------
struct Slot {
  const char* name;
  long long launches;
  double ms, flops, bytes;
};
struct Pending {
  int id;
  hipEvent_t e0, e1;
};
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Slot> g_slots;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;

int prof_slot(const char* name) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < g_slots.size(); ++i)
    if (strcmp(g_slots[i].name, name) == 0) return (int)i;
  g_slots.push_back(Slot{strdup(name), 0, 0.0, 0.0, 0.0});
  return (int)g_slots.size() - 1;
}

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(int kernel_id, hipStream_t s, double flops, double bytes)
    : id(kernel_id), stream(s), on(false) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  on = true;
  e0 = get_event();
  e1 = get_event();
  g_slots[id].launches += 1;
  g_slots[id].flops += flops;
  g_slots[id].bytes += bytes;
  (void)hipEventRecord(e0, stream);
}
ProfScope::~ProfScope() {
  if (!on) return;
  (void)hipEventRecord(e1, stream);
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending.push_back(Pending{id, e0, e1});
}

static void drain_pending() {
  for (auto& p : g_pending) {
    (void)hipEventSynchronize(p.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) g_slots[p.id].ms += ms;
    g_pool.push_back(p.e0);
    g_pool.push_back(p.e1);
  }
  g_pending.clear();
}

}  // namespace wsovod

extern "C" {

const char* wsovod_last_error(void) { return wsovod::g_err; }
int wsovod_abi_version(void) { return 9; }

int wsovod_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(wsovod::g_mu);
  int prev = wsovod::g_on ? 1 : 0;
  wsovod::g_on = on != 0;
  return prev;
}
int wsovod_profile_reset(void) {
  std::lock_guard<std::mutex> lk(wsovod::g_mu);
  wsovod::drain_pending();
  for (auto& s : wsovod::g_slots) {
    s.launches = 0;
    s.ms = s.flops = s.bytes = 0.0;
  }
  return 0;
}
int wsovod_profile_collect(wsovod_prof_entry* out, int cap) {
  std::lock_guard<std::mutex> lk(wsovod::g_mu);
  wsovod::drain_pending();
  int n = 0;
  for (auto& s : wsovod::g_slots) {
    if (n >= cap) break;
    out[n].name = s.name;
    out[n].launches = s.launches;
    out[n].ms = s.ms;
    out[n].flops = s.flops;
    out[n].bytes = s.bytes;
    ++n;
  }
  return n;
}

}  // extern "C"
