// Optional per-launch timing with HIP events on the launch stream (used by bench.py for the roofline line).
#include "common.h"
#include <cstdio>
#include <vector>
#include <mutex>

namespace {
struct Rec { int kid; double flops, bytes; hipEvent_t e0, e1; char tag[96]; };
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

int lcgan_prof_begin(int kid, double flops, double bytes, hipStream_t s, const char* tag) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r; r.kid = kid; r.flops = flops; r.bytes = bytes; r.e0 = get_event(); r.e1 = get_event();
  snprintf(r.tag, sizeof(r.tag), "%s", tag ? tag : "");
  hipEventRecord(r.e0, s);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
// scopes nest (a convolution's generic path launches lcgan_scale_reduce, which has a scope of its own): close by index
void lcgan_prof_end(int idx, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx >= 0 && idx < (int)g_recs.size()) hipEventRecord(g_recs[idx].e1, s);
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
  (void)hipGetLastError();
  return KID_COUNT;
}
// Synchronises and writes one CSV row per recorded launch (kid, ms, flops, bytes, tag -- convolutions tag their geometry) to `path`,
// in launch order, then clears the records: the per-launch in-iteration table of bench.py --launch-table.
int lcgan_prof_dump(const char* path) {
  std::lock_guard<std::mutex> lk(g_mu);
  FILE* f = fopen(path, "w");
  if (!f) return LCGAN_EINVAL;
  fprintf(f, "kid,ms,flops,bytes,tag\n");
  for (auto& r : g_recs) {
    hipEventSynchronize(r.e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    fprintf(f, "%d,%.6f,%.0f,%.0f,%s\n", r.kid, ms, r.flops, r.bytes, r.tag);
    g_pool.push_back(r.e0); g_pool.push_back(r.e1);
  }
  fclose(f);
  g_recs.clear();
  (void)hipGetLastError();
  return LCGAN_OK;
}
}
