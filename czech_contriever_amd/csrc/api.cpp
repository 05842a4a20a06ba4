// api.cpp — error plumbing of the C ABI (host only).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "czc_common.h"

namespace czc {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
  return CZC_ERR_HIP;
}

// ---- per-operation timing ----
bool g_prof_on = false;
int g_defer_partials = 0;
namespace {
struct ProfEntry {
  std::string name;
  double flops = 0.0, bytes = 0.0;   // of ONE call (the last one seen under this name)
  std::vector<hipEvent_t> ev;        // start, stop, start, stop, ...
};
std::vector<ProfEntry> g_prof;
}  // namespace

int prof_begin(hipStream_t st, const char* name, double flops, double bytes) {
  size_t i = 0;
  for (; i < g_prof.size(); ++i)
    if (g_prof[i].name == name) break;
  if (i == g_prof.size()) {
    if (g_prof.size() >= 4096) return -1;
    g_prof.emplace_back();
    g_prof.back().name = name;
  }
  ProfEntry& e = g_prof[i];
  if (e.ev.size() >= 2 * 16384) return -1;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess) return -1;
  if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1; }
  e.flops = flops; e.bytes = bytes;
  e.ev.push_back(a); e.ev.push_back(b);
  (void)hipEventRecord(a, st);
  return (int)i;
}
void prof_end(hipStream_t st, int handle) {
  if (handle < 0 || (size_t)handle >= g_prof.size() || g_prof[handle].ev.empty()) return;
  (void)hipEventRecord(g_prof[handle].ev.back(), st);
}
static void prof_clear() {
  for (auto& e : g_prof)
    for (hipEvent_t ev : e.ev) (void)hipEventDestroy(ev);
  g_prof.clear();
}
}  // namespace czc

extern "C" int czc_prof_enable(int on) {
  czc::prof_clear();
  czc::g_prof_on = on != 0;
  return CZC_OK;
}
// one line per operation name: name<TAB>calls<TAB>total_ms<TAB>flops_per_call<TAB>bytes_per_call ; returns the text length
// (including the terminating 0) that the full report needs, writes at most `len` bytes
extern "C" int64_t czc_prof_read(char* buf, int64_t len) {
  std::string out;
  char line[512];
  for (auto& e : czc::g_prof) {
    double tot = 0.0;
    int calls = 0;
    for (size_t i = 0; i + 1 < e.ev.size(); i += 2) {
      float ms = 0.f;
      if (hipEventSynchronize(e.ev[i + 1]) != hipSuccess) continue;
      if (hipEventElapsedTime(&ms, e.ev[i], e.ev[i + 1]) != hipSuccess) continue;
      tot += ms;
      ++calls;
    }
    snprintf(line, sizeof(line), "%s\t%d\t%.6f\t%.6e\t%.6e\n", e.name.c_str(), calls, tot, e.flops, e.bytes);
    out += line;
  }
  if (buf && len > 0) {
    const size_t n = out.size() + 1 <= (size_t)len ? out.size() : (size_t)len - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int64_t)out.size() + 1;
}

extern "C" const char* czc_last_error(void) { return czc::g_err; }
extern "C" int czc_abi_version(void) { return CZC_ABI_VERSION; }
#ifndef CZC_SOURCE_HASH
#define CZC_SOURCE_HASH "unknown"
#endif
// (the marker lets the loader read the hash from the file's bytes, without mapping a binary it may be about to replace)
static const char k_source_hash[] = "CZC_SOURCE_HASH=" CZC_SOURCE_HASH;
extern "C" const char* czc_source_hash(void) { return k_source_hash + 16; }

extern "C" int czc_event_create(czc_event_t* ev) {
  CZC_REQUIRE(ev, "czc_event_create: null argument");
  hipEvent_t e;
  CZC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *ev = (czc_event_t)e;
  return CZC_OK;
}
extern "C" int czc_event_destroy(czc_event_t ev) {
  if (ev) CZC_HIP(hipEventDestroy((hipEvent_t)ev));
  return CZC_OK;
}
extern "C" int czc_event_record(czc_event_t ev, czc_stream_t stream) {
  CZC_REQUIRE(ev, "czc_event_record: null event");
  CZC_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return CZC_OK;
}
extern "C" int czc_stream_wait_event(czc_stream_t stream, czc_event_t ev) {
  CZC_REQUIRE(ev, "czc_stream_wait_event: null event");
  CZC_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
  return CZC_OK;
}
