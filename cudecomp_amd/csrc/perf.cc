// perf.cc -- opt-in per-operation timing (CUDECOMP_ENABLE_PERFORMANCE_REPORT=1), the counterpart of the
// reference's performance report (src/performance.cc; record sites include/internal/transpose.h:307-321,
// 897-904): every transpose records [start, packed, exchanged, done] events on the caller's stream, the
// report splits the time into local kernels and exchange and derives the all-to-all bandwidth.
#include <cstdio>

#include "errors.h"
#include "internal.h"

namespace cudecomp {

namespace {
constexpr int kRing = 32;
const char* kOpNames[4] = {"TransposeXY", "TransposeYZ", "TransposeZY", "TransposeYX"};
}  // namespace

hipEvent_t* perfBegin(cudecompHandle_t h, cudecompGridDesc_t gd, int op, int64_t pencil_bytes, hipStream_t stream) {
  if (!h->performance_report_enable) return nullptr;
  auto& ring = gd->perf[op];
  if (ring.empty()) {
    ring.resize(kRing);
    for (auto& s : ring)
      for (auto& e : s.ev) CD_CHECK_HIP(hipEventCreate(&e));
  }
  auto& s = ring[gd->perf_calls[op] % kRing];
  gd->perf_calls[op]++;
  gd->perf_bytes[op] = pencil_bytes;
  s.used = true;
  CD_CHECK_HIP(hipEventRecord(s.ev[0], stream));
  return s.ev;
}

TransposeTimings perfCollect(cudecompGridDesc_t gd, int op) {
  TransposeTimings t;
  t.calls = gd->perf_calls[op];
  t.pencil_bytes = gd->perf_bytes[op];
  if (gd->perf[op].empty()) return t;
  (void)hipDeviceSynchronize();
  for (auto& s : gd->perf[op]) {
    if (!s.used) continue;
    float a = 0, b = 0, c = 0;
    if (hipEventElapsedTime(&a, s.ev[0], s.ev[1]) != hipSuccess || hipEventElapsedTime(&b, s.ev[1], s.ev[2]) != hipSuccess ||
        hipEventElapsedTime(&c, s.ev[2], s.ev[3]) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    t.pack_ms += a;
    t.exchange_ms += b;
    t.unpack_ms += c;
    t.samples++;
  }
  if (t.samples) {
    t.pack_ms /= t.samples;
    t.exchange_ms /= t.samples;
    t.unpack_ms /= t.samples;
    t.total_ms = t.pack_ms + t.exchange_ms + t.unpack_ms;
  }
  return t;
}

void perfReset(cudecompGridDesc_t gd) {
  for (int op = 0; op < 4; ++op) {
    for (auto& s : gd->perf[op]) s.used = false;
    gd->perf_calls[op] = 0;
  }
}

void perfReport(cudecompHandle_t h, cudecompGridDesc_t gd) {
  if (!h->performance_report_enable) return;
  bool any = false;
  for (int op = 0; op < 4; ++op) any = any || gd->perf_calls[op] > 0;
  if (!any) return;
  // per-rank numbers reduced to min / max / avg over ranks (collective: destroy is collective)
  if (h->rank == 0) {
    printf("CUDECOMP: ===== Performance Summary =====\n");
    printf("CUDECOMP: grid %d x %d x %d, process grid %d x %d, transpose backend %s\n", gd->config.gdims[0],
           gd->config.gdims[1], gd->config.gdims[2], gd->config.pdims[0], gd->config.pdims[1],
           cudecompTransposeCommBackendToString(gd->config.transpose_comm_backend));
    printf("CUDECOMP: %-12s %8s %12s %12s %12s %12s %14s\n", "operation", "calls", "total [ms]", "pack [ms]",
           "a2a [ms]", "unpack [ms]", "a2a BW [GB/s]");
  }
  for (int op = 0; op < 4; ++op) {
    const TransposeTimings t = perfCollect(gd, op);
    const double total = h->boot->allreduceMax(t.total_ms), pack = h->boot->allreduceMax(t.pack_ms),
                 xch = h->boot->allreduceMax(t.exchange_ms), unp = h->boot->allreduceMax(t.unpack_ms);
    const int64_t calls = h->boot->allreduceMaxI64(t.calls);
    if (h->rank == 0 && calls > 0)
      printf("CUDECOMP: %-12s %8lld %12.4f %12.4f %12.4f %12.4f %14.1f\n", kOpNames[op], (long long)calls, total, pack,
             xch, unp, xch > 0 ? (double)t.pencil_bytes / (xch * 1e6) : 0.0);
  }
  if (h->rank == 0) fflush(stdout);
}

}  // namespace cudecomp
