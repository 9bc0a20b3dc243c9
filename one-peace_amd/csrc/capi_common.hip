// Error reporting + live kernel timing for the C ABI (include/onepeace_hip.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = {0};

extern "C" void op_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* op_last_error(void) { return g_err; }

extern "C" int op_abi_version(void) { return 8; }

// ---- live per-kernel-family timing with HIP events on the launch stream ---------------------------
// bench.py enables this around its timed region; gemm launches then record an event pair on the
// stream they are enqueued on.  Disabled (the default) it costs one relaxed load per launch.
namespace {
struct ProfRec { hipEvent_t a, b; int family; double work; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
int g_prof_on = 0;
}  // namespace

extern "C" int op_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on;
  return OP_OK;
}

extern "C" int op_prof_active(void) { return g_prof_on; }

// begin: records the start event, returns a slot (or -1 when profiling is off)
extern "C" int op_prof_begin(int family, double work, void* stream) {
  if (!g_prof_on) return -1;
  ProfRec r;
  r.family = family;
  r.work = work;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  hipEventRecord(r.a, (hipStream_t)stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
  return (int)g_prof.size() - 1;
}

extern "C" void op_prof_end(int slot, void* stream) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (slot < (int)g_prof.size()) hipEventRecord(g_prof[slot].b, (hipStream_t)stream);
}

// Synchronises on the recorded events and accumulates, per family (0..n_families-1):
// total milliseconds, launch count, total work units (flops or bytes).  Clears the records.
extern "C" int op_prof_collect(double* ms, int64_t* count, double* work, int n_families) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < n_families; ++i) { ms[i] = 0; count[i] = 0; work[i] = 0; }
  for (auto& r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess &&
        r.family >= 0 && r.family < n_families) {
      ms[r.family] += t;
      count[r.family] += 1;
      work[r.family] += r.work;
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_prof.clear();
  return OP_OK;
}
