// Library-side launch timing: while sopro_prof_enable(1) is in effect the stage sequences (stages.hip) bracket their heavy
// launches with two HIP events on the stream the launch goes to, tagged with a kernel family and the algorithmic flops of the
// launch.  bench.py's roofline leg reads the totals with sopro_prof_collect.  Nothing is recorded while a stream is capturing.
// A sample whose stream was idle when its first event was recorded is host-bound (the span contains the host's time between
// the record and the launch): such samples are counted apart (gpu_bound / ms_bound / flops_bound are the others).
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace {
struct Rec {
  const char* family;
  double flops;
  hipEvent_t e0, e1;
  bool bound;
};
std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<Rec*> g_recs;
}  // namespace

sopro_prof_scope::sopro_prof_scope(const char* family, double flops, hipStream_t st) : rec(nullptr), s(st) {
  if (!family || !g_on.load(std::memory_order_relaxed)) return;  // (family NULL: a launch this call skips - see mimi_decode_core's parts)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return;
  }
  Rec* r = new Rec{family, flops, nullptr, nullptr, false};
  if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) {
    (void)hipGetLastError();
    delete r;
    return;
  }
  r->bound = hipStreamQuery(st) == hipErrorNotReady;  // work still queued in front of the launch: the span is GPU time
  (void)hipGetLastError();                             // (hipErrorNotReady is an answer, not a failure of the launch that follows)
  (void)hipEventRecord(r->e0, st);
  rec = r;
}

sopro_prof_scope::~sopro_prof_scope() {
  if (!rec) return;
  Rec* r = static_cast<Rec*>(rec);
  (void)hipEventRecord(r->e1, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(r);
}

extern "C" int sopro_prof_enable(int on) {
  g_on.store(on ? 1 : 0);
  return 0;
}

extern "C" int sopro_prof_collect(sopro_prof_row* rows, int32_t cap, int32_t* n_rows) {
  SOPRO_CHECK_ARG(n_rows != nullptr && (rows != nullptr || cap == 0), "rows / n_rows is NULL");
  std::vector<Rec*> recs;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    recs.swap(g_recs);
  }
  int n = 0, rc = 0;
  for (Rec* r : recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r->e1) == hipSuccess && hipEventElapsedTime(&ms, r->e0, r->e1) == hipSuccess) {
      int i = 0;
      while (i < n && strncmp(rows[i].family, r->family, sizeof(rows[i].family)) != 0) ++i;
      if (i == n && n < cap) {
        memset(&rows[n], 0, sizeof(rows[n]));
        strncpy(rows[n].family, r->family, sizeof(rows[n].family) - 1);
        ++n;
      }
      if (i < n) {
        sopro_prof_row& d = rows[i];
        d.launches += 1;
        d.flops += r->flops;
        d.ms_all += ms;
        if (r->bound) {
          d.gpu_bound += 1;
          d.ms_bound += ms;
          d.flops_bound += r->flops;
        }
      } else {
        rc = -1;
      }
    } else {
      (void)hipGetLastError();
    }
    (void)hipEventDestroy(r->e0);
    (void)hipEventDestroy(r->e1);
    delete r;
  }
  *n_rows = n;
  if (rc != 0) sopro_set_error("sopro_prof_collect: more than %d families", cap);
  return rc;
}
