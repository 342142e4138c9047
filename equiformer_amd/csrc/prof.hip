// Optional per-launch timing of the library's kernels with HIP events recorded on the launch stream.
// bench.py uses it to measure the dominant kernel's average launch duration inside the timed region
// (roofline.achieved); it is off by default and costs nothing then.
#include <mutex>
#include <string>
#include <vector>
#include <map>
#include <cstring>
#include <cstdio>
#include "common.h"
#include "prof.h"

namespace {
struct Rec {
  std::string name;
  hipEvent_t a, b;
  double flops, bytes;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::string g_filter;
bool g_on = false;
}  // namespace

int eqf_prof_begin(const char* name, hipStream_t st, double flops, double bytes) {
  if (!g_on) return -1;
  if (!g_filter.empty() && std::strstr(name, g_filter.c_str()) == nullptr) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.name = name;
  r.flops = flops;
  r.bytes = bytes;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void eqf_prof_end(int idx, hipStream_t st) {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEventRecord(g_recs[idx].b, st);
}

extern "C" {

int eqf_prof_enable(const char* filter) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = filter != nullptr;
  g_filter = filter ? filter : "";
  return 0;
}

// Synchronises the recorded events, writes one line per kernel name: "name count total_ms flops bytes\n",
// clears the records.  Returns the number of bytes written (truncated to buflen-1), or <0 on error.
int eqf_prof_report(char* buf, int buflen) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, double[4]> agg;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& a = agg[r.name];
      a[0] += 1, a[1] += ms, a[2] += r.flops, a[3] += r.bytes;
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_recs.clear();
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof line, "%s %.0f %.6f %.6e %.6e\n", kv.first.c_str(), kv.second[0], kv.second[1], kv.second[2],
             kv.second[3]);
    out += line;
  }
  if (!buf || buflen <= 0) return -1;
  int n = (int)out.size() < buflen - 1 ? (int)out.size() : buflen - 1;
  memcpy(buf, out.data(), n);
  buf[n] = 0;
  return n;
}
}
