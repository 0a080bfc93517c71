// Optional per-launch timing with HIP events on the launch stream (used by bench.py for the roofline line).
#include "common.h"
#include <vector>
#include <mutex>

namespace {
struct Rec { int kid; double flops, bytes; hipEvent_t e0, e1; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
int g_active = 0;
std::mutex g_mu;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
}  // namespace

extern "C" int lcgan_prof_active() { return g_active; }

void lcgan_prof_begin(int kid, double flops, double bytes, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r; r.kid = kid; r.flops = flops; r.bytes = bytes; r.e0 = get_event(); r.e1 = get_event();
  hipEventRecord(r.e0, s);
  g_recs.push_back(r);
}
void lcgan_prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_recs.empty()) hipEventRecord(g_recs.back().e1, s);
}

extern "C" {
// enable (1) / disable (0) per-launch event recording; enabling clears earlier records
int lcgan_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (on) { for (auto& r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); } g_recs.clear(); }
  g_active = on ? 1 : 0;
  return LCGAN_OK;
}
// Synchronises and sums the records per kernel family: out_ms[KID_COUNT], out_flops[KID_COUNT], out_bytes[KID_COUNT],
// out_count[KID_COUNT]. Returns the number of kernel families (KID_COUNT).
int lcgan_prof_collect(double* out_ms, double* out_flops, double* out_bytes, long long* out_count) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < KID_COUNT; ++i) { out_ms[i] = 0; out_flops[i] = 0; out_bytes[i] = 0; out_count[i] = 0; }
  for (auto& r : g_recs) {
    hipEventSynchronize(r.e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    out_ms[r.kid] += ms; out_flops[r.kid] += r.flops; out_bytes[r.kid] += r.bytes; out_count[r.kid] += 1;
    g_pool.push_back(r.e0); g_pool.push_back(r.e1);
  }
  g_recs.clear();
  return KID_COUNT;
}
}
