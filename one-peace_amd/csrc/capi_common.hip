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

extern "C" int op_abi_version(void) { return 9; }

// ---- live per-kernel-family timing with HIP events on the launch stream ---------------------------
// bench.py enables this around its timed region; gemm launches then record an event pair on the
// stream they are enqueued on.  Disabled (the default) it costs one relaxed load per launch.
namespace {
struct ProfRec { hipEvent_t a, b; int family; double work; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_free;  // events of collected records, reused: a profiled step records ~6 000 of them
int g_prof_on = 0;

bool prof_event(hipEvent_t* e) {  // (under g_prof_mu)
  if (!g_prof_free.empty()) {
    *e = g_prof_free.back();
    g_prof_free.pop_back();
    return true;
  }
  // no system-scope fence at the event: the default flavour writes back / invalidates the caches at every record, which costs the
  // FOLLOWING kernel (measured: a profiled headline step +28 ms with 1 700 default events).  Timing stays enabled.
  return hipEventCreateWithFlags(e, hipEventDisableSystemFence) == hipSuccess;
}
}  // namespace

extern "C" int op_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on;
  return OP_OK;
}

extern "C" int op_prof_active(void) { return g_prof_on; }

// Pre-creates `events` events (two per profiled launch) so that the first profiled step does not pay for their creation.
extern "C" int op_prof_reserve(int events) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  while ((int)g_prof_free.size() < events) {
    hipEvent_t e;
    hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    if (err != hipSuccess) {
      op_set_error("op_prof_reserve: hipEventCreateWithFlags failed: %s", hipGetErrorString(err));
      return (int)err;
    }
    g_prof_free.push_back(e);
  }
  return OP_OK;
}

// begin: records the start event, returns a slot (or -1 when profiling is off)
extern "C" int op_prof_begin(int family, double work, void* stream) {
  if (!g_prof_on) return -1;
  ProfRec r;
  r.family = family;
  r.work = work;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!prof_event(&r.a)) return -1;
  if (!prof_event(&r.b)) {
    g_prof_free.push_back(r.a);
    return -1;
  }
  hipEventRecord(r.a, (hipStream_t)stream);
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
    g_prof_free.push_back(r.a);
    g_prof_free.push_back(r.b);
  }
  g_prof.clear();
  return OP_OK;
}
