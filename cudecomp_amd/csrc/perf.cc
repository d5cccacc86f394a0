// perf.cc -- opt-in performance report (CUDECOMP_ENABLE_PERFORMANCE_REPORT=1), the counterpart of the reference's
// src/performance.cc (record sites include/internal/transpose.h:307-321, 897-904, halo.h:76-83, 232-238).
//
// Every transpose / halo update records four events on the caller's stream: [start, first local phase done,
// exchange done, end].  Samples are kept per call configuration -- (op, dtype, halos, padding, in place) for
// transposes, (axis, dim, dtype, halos, periods, padding) for halos -- in a ring of
// CUDECOMP_PERFORMANCE_REPORT_SAMPLES entries after skipping CUDECOMP_PERFORMANCE_REPORT_WARMUP_SAMPLES calls.
// When the grid descriptor is destroyed the ranks' tables are gathered on rank 0, averaged over samples and
// ranks, printed in the reference's table layout and, with CUDECOMP_PERFORMANCE_REPORT_WRITE_DIR, written as CSV
// files under the reference's file names and column headers, so scripts that parse one parse the other.
//
// The gather ships each rank's table as one opaque blob: ranks need not hold the same set of configurations
// (a rank with nothing to exchange along a halo dim records nothing), rank 0 merges by key.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "errors.h"
#include "internal.h"

namespace cudecomp {

namespace {

const char* kTransposeNames[4] = {"TransposeXY", "TransposeYZ", "TransposeZY", "TransposeYX"};
const char* kHaloNames[3] = {"HaloX", "HaloY", "HaloZ"};

// S, D, C, Z in this order (CUDECOMP_FLOAT = -1 ... CUDECOMP_DOUBLE_COMPLEX = -4)
int dtypeRank(int dtype) { return -dtype - 1; }
const char* dtypeLetter(int rank) {
  static const char* names[4] = {"S", "D", "C", "Z"};
  return (rank >= 0 && rank < 4) ? names[rank] : "unknown";
}

hipEvent_t* beginSample(cudecompHandle_t h, cudecompGridDesc::PerfCollection& c, int64_t wire_bytes, hipStream_t stream) {
  const int64_t call = c.calls++;
  c.wire_bytes = wire_bytes;
  if (call < h->performance_report_warmup_samples) return nullptr;
  if (c.ring.empty()) c.ring.resize(h->performance_report_samples);
  auto& s = c.ring[(call - h->performance_report_warmup_samples) % (int64_t)c.ring.size()];
  if (!s.ev[0])
    for (auto& e : s.ev) CD_CHECK_HIP(hipEventCreate(&e));
  s.used = true;
  CD_CHECK_HIP(hipEventRecord(s.ev[0], stream));
  return s.ev;
}

// one configuration's retained samples, in milliseconds
struct Series {
  std::vector<float> total, exchange, first, last;
};

Series readSeries(const cudecompGridDesc::PerfCollection& c) {
  Series out;
  for (const auto& s : c.ring) {
    if (!s.used) continue;
    float a = 0, b = 0, d = 0;
    if (hipEventElapsedTime(&a, s.ev[0], s.ev[1]) != hipSuccess || hipEventElapsedTime(&b, s.ev[1], s.ev[2]) != hipSuccess ||
        hipEventElapsedTime(&d, s.ev[2], s.ev[3]) != hipSuccess) {
      (void)hipGetLastError();  // sample still in flight or never completed: leave it out
      continue;
    }
    out.first.push_back(a);
    out.exchange.push_back(b);
    out.last.push_back(d);
    out.total.push_back(a + b + d);
  }
  return out;
}

// ---- wire format of one table row --------------------------------------------------------------------------
struct RowKey {
  int32_t is_halo, op, dtype_rank, dim, inplace;
  int32_t a[12];  // transposes: in halo, out halo, in pad, out pad; halos: halo, periods, padding
  bool operator<(const RowKey& o) const { return std::memcmp(this, &o, sizeof(RowKey)) < 0; }
  bool operator==(const RowKey& o) const { return std::memcmp(this, &o, sizeof(RowKey)) == 0; }
};

struct Row {
  RowKey key;
  int64_t wire_bytes = 0;
  std::vector<float> total, exchange;  // per sample
};

void put(std::vector<char>& blob, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  blob.insert(blob.end(), c, c + n);
}

std::vector<char> serialise(const std::vector<Row>& rows) {
  std::vector<char> blob;
  const int64_t n = (int64_t)rows.size();
  put(blob, &n, sizeof(n));
  for (const Row& r : rows) {
    put(blob, &r.key, sizeof(RowKey));
    put(blob, &r.wire_bytes, sizeof(int64_t));
    const int64_t ns = (int64_t)r.total.size();
    put(blob, &ns, sizeof(ns));
    put(blob, r.total.data(), ns * sizeof(float));
    put(blob, r.exchange.data(), ns * sizeof(float));
  }
  return blob;
}

std::vector<Row> deserialise(const char* p) {
  std::vector<Row> rows;
  int64_t n;
  std::memcpy(&n, p, sizeof(n));
  p += sizeof(n);
  for (int64_t i = 0; i < n; ++i) {
    Row r;
    std::memcpy(&r.key, p, sizeof(RowKey));
    p += sizeof(RowKey);
    std::memcpy(&r.wire_bytes, p, sizeof(int64_t));
    p += sizeof(int64_t);
    int64_t ns;
    std::memcpy(&ns, p, sizeof(ns));
    p += sizeof(ns);
    r.total.resize(ns);
    r.exchange.resize(ns);
    std::memcpy(r.total.data(), p, ns * sizeof(float));
    p += ns * sizeof(float);
    std::memcpy(r.exchange.data(), p, ns * sizeof(float));
    p += ns * sizeof(float);
    rows.push_back(std::move(r));
  }
  return rows;
}

std::vector<Row> localRows(cudecompGridDesc_t gd) {
  std::vector<Row> rows;
  for (const auto& kv : gd->perf_transpose) {
    const Series s = readSeries(kv.second);
    if (s.total.empty()) continue;
    Row r;
    std::memset(&r.key, 0, sizeof(RowKey));
    r.key.is_halo = 0;
    r.key.op = std::get<0>(kv.first);
    r.key.dtype_rank = dtypeRank(std::get<1>(kv.first));
    r.key.inplace = std::get<3>(kv.first) ? 1 : 0;
    for (int i = 0; i < 12; ++i) r.key.a[i] = std::get<2>(kv.first)[i];
    r.wire_bytes = kv.second.wire_bytes;
    r.total = s.total;
    r.exchange = s.exchange;
    if (r.wire_bytes == 0) std::fill(r.exchange.begin(), r.exchange.end(), 0.0f);  // local-only: no exchange phase
    rows.push_back(std::move(r));
  }
  for (const auto& kv : gd->perf_halo) {
    const Series s = readSeries(kv.second);
    if (s.total.empty()) continue;
    Row r;
    std::memset(&r.key, 0, sizeof(RowKey));
    r.key.is_halo = 1;
    r.key.op = std::get<0>(kv.first);
    r.key.dim = std::get<1>(kv.first);
    r.key.dtype_rank = dtypeRank(std::get<2>(kv.first));
    for (int i = 0; i < 3; ++i) {
      r.key.a[i] = std::get<3>(kv.first)[i];
      r.key.a[3 + i] = std::get<4>(kv.first)[i] ? 1 : 0;
      r.key.a[6 + i] = std::get<5>(kv.first)[i];
    }
    r.wire_bytes = kv.second.wire_bytes;
    r.total = s.total;
    r.exchange = s.exchange;
    if (r.wire_bytes == 0) std::fill(r.exchange.begin(), r.exchange.end(), 0.0f);  // self-periodic wrap: nothing sent
    rows.push_back(std::move(r));
  }
  return rows;
}

// ---- formatting ----------------------------------------------------------------------------------------------
std::string triple(const int32_t* v) {
  char buf[64];
  std::snprintf(buf, sizeof(buf), "[%d,%d,%d]", v[0], v[1], v[2]);
  return buf;
}

struct Merged {
  RowKey key;
  std::vector<int> ranks;                 // contributing ranks
  std::vector<std::vector<float>> total;  // [contributor][sample]
  std::vector<std::vector<float>> exchange;
  std::vector<int64_t> wire_bytes;
  int samples = 0;  // largest per-rank sample count
  double total_avg = 0, exchange_avg = 0, local_avg = 0, bw_avg = 0;
};

// order of the reference's tables: transposes by (op, dtype, halos, padding, in place), halos by (op, dim, dtype, ...)
bool tableOrder(const Merged& x, const Merged& y) {
  const RowKey &a = x.key, &b = y.key;
  if (a.is_halo != b.is_halo) return a.is_halo < b.is_halo;
  if (a.op != b.op) return a.op < b.op;
  if (a.is_halo && a.dim != b.dim) return a.dim < b.dim;
  if (a.dtype_rank != b.dtype_rank) return a.dtype_rank < b.dtype_rank;
  const int c = std::memcmp(a.a, b.a, sizeof(a.a));
  if (c != 0) return c < 0;
  return a.inplace < b.inplace;
}

void finish(Merged& m) {
  // averages: per rank over its samples, then over the contributing ranks (reference computeGlobalAverage)
  double t = 0, x = 0, l = 0, bw = 0;
  for (size_t r = 0; r < m.total.size(); ++r) {
    double rt = 0, rx = 0, rbw = 0;
    const size_t n = m.total[r].size();
    for (size_t s = 0; s < n; ++s) {
      rt += m.total[r][s];
      rx += m.exchange[r][s];
      if (m.exchange[r][s] > 0) rbw += (double)m.wire_bytes[r] * 1e-6 / m.exchange[r][s];
    }
    if (n) {
      t += rt / n;
      x += rx / n;
      l += (rt - rx) / n;
      bw += rbw / n;
    }
    m.samples = std::max(m.samples, (int)n);
  }
  const double nr = (double)std::max<size_t>(1, m.total.size());
  m.total_avg = t / nr;
  m.exchange_avg = x / nr;
  m.local_avg = l / nr;
  m.bw_avg = bw / nr;
}

std::string csvName(cudecompGridDesc_t gd, const std::string& dir, const char* table) {
  std::ostringstream f;
  f << dir << (dir.empty() || dir.back() == '/' ? "" : "/") << "cudecomp-perf-report-" << table << "-tcomm_"
    << (int)gd->config.transpose_comm_backend << "-hcomm_" << (int)gd->config.halo_comm_backend << "-pdims_"
    << gd->config.pdims[0] << "x" << gd->config.pdims[1] << "-gdims_" << gd->config.gdims[0] << "x"
    << gd->config.gdims[1] << "x" << gd->config.gdims[2] << "-memorder_";
  for (int ax = 0; ax < 3; ++ax)
    for (int i = 0; i < 3; ++i) f << gd->config.transpose_mem_order[ax][i];
  f << ".csv";
  return f.str();
}

void csvPreamble(std::ofstream& f, cudecompGridDesc_t gd) {
  f << "# Transpose backend: " << cudecompTransposeCommBackendToString(gd->config.transpose_comm_backend) << "\n";
  f << "# Halo backend: " << cudecompHaloCommBackendToString(gd->config.halo_comm_backend) << "\n";
  f << "# Process grid: [" << gd->config.pdims[0] << ", " << gd->config.pdims[1] << "]\n";
  f << "# Global dimensions: [" << gd->config.gdims[0] << ", " << gd->config.gdims[1] << ", " << gd->config.gdims[2]
    << "]\n";
  f << "# Memory order: ";
  for (int ax = 0; ax < 3; ++ax)
    f << "[" << gd->config.transpose_mem_order[ax][0] << "," << gd->config.transpose_mem_order[ax][1] << ","
      << gd->config.transpose_mem_order[ax][2] << "]" << (ax < 2 ? "; " : "");
  f << "\n#\n";
}

std::string fixed3(double v) {
  char buf[32];
  std::snprintf(buf, sizeof(buf), "%.3f", v);
  return buf;
}

std::string keyColumnsCsv(const RowKey& k) {
  std::ostringstream s;
  if (!k.is_halo)
    s << kTransposeNames[k.op] << "," << dtypeLetter(k.dtype_rank) << ",\"" << triple(k.a) << "\",\"" << triple(k.a + 3)
      << "\",\"" << triple(k.a + 6) << "\",\"" << triple(k.a + 9) << "\"," << (k.inplace ? "T" : "F") << ",F";
  else
    s << kHaloNames[k.op] << "," << dtypeLetter(k.dtype_rank) << "," << k.dim << ",\"" << triple(k.a) << "\",\""
      << triple(k.a + 3) << "\",\"" << triple(k.a + 6) << "\",F";
  return s.str();
}

bool openCsv(std::ofstream& f, const std::string& name) {
  f.open(name);
  if (!f.is_open()) {
    printf("CUDECOMP:WARN: Could not open file %s for writing\n", name.c_str());
    return false;
  }
  return true;
}

void printSummary(cudecompGridDesc_t gd, const std::vector<Merged>& rows, bool halo) {
  if (!halo) {
    printf("CUDECOMP: Transpose Performance Data:\nCUDECOMP:\n");
    printf("CUDECOMP: %-12s %-6s %-15s %-15s %-8s %-8s %-8s %-9s %-9s %-9s %-9s\n", "operation", "dtype", "halo extents",
           "padding", "inplace", "managed", "samples", "total", "A2A", "local", "A2A BW");
    printf("CUDECOMP: %-12s %-6s %-15s %-15s %-8s %-8s %-8s %-9s %-9s %-9s %-9s\n", "", "", "", "", "", "", "", "[ms]",
           "[ms]", "[ms]", "[GB/s]");
    printf("CUDECOMP: %s\n", std::string(120, '-').c_str());
    for (const Merged& m : rows)
      if (!m.key.is_halo)
        printf("CUDECOMP: %-12s %-6s %-7s/%-7s %-7s/%-7s %-8s %-8s %-8d %-9.3f %-9.3f %-9.3f %-9.3f\n",
               kTransposeNames[m.key.op], dtypeLetter(m.key.dtype_rank), triple(m.key.a).c_str(),
               triple(m.key.a + 3).c_str(), triple(m.key.a + 6).c_str(), triple(m.key.a + 9).c_str(),
               m.key.inplace ? "T" : "F", "F", m.samples, m.total_avg, m.exchange_avg, m.local_avg, m.bw_avg);
  } else {
    printf("CUDECOMP:\nCUDECOMP: Halo Performance Data:\nCUDECOMP:\n");
    printf("CUDECOMP: %-12s %-6s %-5s %-12s %-12s %-12s %-8s %-8s %-9s %-9s %-9s %-9s\n", "operation", "dtype", "dim",
           "halo extent", "periods", "padding", "managed", "samples", "total", "SR", "local", "SR BW");
    printf("CUDECOMP: %-12s %-6s %-5s %-12s %-12s %-12s %-8s %-8s %-9s %-9s %-9s %-9s\n", "", "", "", "", "", "", "", "",
           "[ms]", "[ms]", "[ms]", "[GB/s]");
    printf("CUDECOMP: %s\n", std::string(125, '-').c_str());
    for (const Merged& m : rows)
      if (m.key.is_halo)
        printf("CUDECOMP: %-12s %-6s %-5d %-12s %-12s %-12s %-8s %-8d %-9.3f %-9.3f %-9.3f %-9.3f\n", kHaloNames[m.key.op],
               dtypeLetter(m.key.dtype_rank), m.key.dim, triple(m.key.a).c_str(), triple(m.key.a + 3).c_str(),
               triple(m.key.a + 6).c_str(), "F", m.samples, m.total_avg, m.exchange_avg, m.local_avg, m.bw_avg);
  }
  (void)gd;
}

void writeSummaryCsv(cudecompHandle_t h, cudecompGridDesc_t gd, const std::vector<Merged>& rows, bool halo) {
  const std::string name = csvName(gd, h->performance_report_write_dir, halo ? "halo-aggregated" : "transpose-aggregated");
  std::ofstream f;
  if (!openCsv(f, name)) return;
  csvPreamble(f, gd);
  if (!halo)
    f << "operation,dtype,input_halo_extents,output_halo_extents,input_padding,output_padding,inplace,managed,samples,"
         "total_ms,A2A_ms,local_ms,A2A_BW_GBps\n";
  else
    f << "operation,dtype,dim,halo_extent,periods,padding,managed,samples,total_ms,SR_ms,local_ms,SR_BW_GBps\n";
  for (const Merged& m : rows)
    if ((m.key.is_halo != 0) == halo)
      f << keyColumnsCsv(m.key) << "," << m.samples << "," << fixed3(m.total_avg) << "," << fixed3(m.exchange_avg) << ","
        << fixed3(m.local_avg) << "," << fixed3(m.bw_avg) << "\n";
  printf("CUDECOMP:\nCUDECOMP: Wrote %s performance data to %s\n", halo ? "halo" : "transpose", name.c_str());
}

void printSamples(cudecompHandle_t h, cudecompGridDesc_t gd, const std::vector<Merged>& rows, bool halo) {
  const bool all_ranks = h->performance_report_detail >= 2;
  std::ofstream csv;
  bool csv_ok = false;
  std::string name;
  if (!h->performance_report_write_dir.empty()) {
    name = csvName(gd, h->performance_report_write_dir, halo ? "halo-samples" : "transpose-samples");
    csv_ok = openCsv(csv, name);
    if (csv_ok) {
      csvPreamble(csv, gd);
      if (!halo)
        csv << "operation,dtype,input_halo_extents,output_halo_extents,input_padding,output_padding,inplace,managed,"
               "rank,sample,total_ms,A2A_ms,local_ms,A2A_BW_GBps\n";
      else
        csv << "operation,dtype,dim,halo_extent,periods,padding,managed,rank,sample,total_ms,SR_ms,local_ms,SR_BW_GBps\n";
    }
  }
  for (const Merged& m : rows) {
    if ((m.key.is_halo != 0) != halo) continue;
    const RowKey& k = m.key;
    if (!halo)
      printf("CUDECOMP: %s (dtype=%s, halo extents=%s/%s, padding=%s/%s, inplace=%s, managed=%s) samples:\n",
             kTransposeNames[k.op], dtypeLetter(k.dtype_rank), triple(k.a).c_str(), triple(k.a + 3).c_str(),
             triple(k.a + 6).c_str(), triple(k.a + 9).c_str(), k.inplace ? "T" : "F", "F");
    else
      printf("CUDECOMP: %s (dtype=%s, dim=%d, halos=%s, periods=%s, padding=%s, managed=%s) samples:\n", kHaloNames[k.op],
             dtypeLetter(k.dtype_rank), k.dim, triple(k.a).c_str(), triple(k.a + 3).c_str(), triple(k.a + 6).c_str(), "F");
    printf("CUDECOMP: %-6s %-12s %-9s %-9s %-9s %-9s\n", "rank", "sample", "total", halo ? "SR" : "A2A", "local",
           halo ? "SR BW" : "A2A BW");
    printf("CUDECOMP: %-6s %-12s %-9s %-9s %-9s %-9s\n", "", "", "[ms]", "[ms]", "[ms]", "[GB/s]");
    for (size_t r = 0; r < m.ranks.size(); ++r) {
      if (!all_ranks && m.ranks[r] != 0) continue;
      for (size_t s = 0; s < m.total[r].size(); ++s) {
        const double t = m.total[r][s], x = m.exchange[r][s];
        const double bw = x > 0 ? (double)m.wire_bytes[r] * 1e-6 / x : 0.0;
        printf("CUDECOMP: %-6d %-12d %-9.3f %-9.3f %-9.3f %-9.3f\n", m.ranks[r], (int)s, t, x, t - x, bw);
        if (csv_ok)
          csv << keyColumnsCsv(k) << "," << m.ranks[r] << "," << s << "," << fixed3(t) << "," << fixed3(x) << ","
              << fixed3(t - x) << "," << fixed3(bw) << "\n";
      }
    }
    printf("CUDECOMP:\n");
  }
  if (csv_ok) printf("CUDECOMP:\nCUDECOMP: Wrote per-sample %s data to %s\n", halo ? "halo" : "transpose", name.c_str());
}

}  // namespace

hipEvent_t* perfBeginTranspose(cudecompHandle_t h, cudecompGridDesc_t gd, int op, cudecompDataType_t dtype,
                               const std::array<int32_t, 12>& halos_pads, bool inplace, int64_t wire_bytes,
                               hipStream_t stream) {
  if (!h->performance_report_enable) return nullptr;
  auto& c = gd->perf_transpose[cudecompGridDesc::TransposePerfKey{op, (int)dtype, halos_pads, inplace}];
  return beginSample(h, c, wire_bytes, stream);
}

hipEvent_t* perfBeginHalo(cudecompHandle_t h, cudecompGridDesc_t gd, int axis, int dim, cudecompDataType_t dtype,
                          const std::array<int32_t, 3>& halo, const std::array<bool, 3>& periods,
                          const std::array<int32_t, 3>& padding, int64_t wire_bytes, hipStream_t stream) {
  if (!h->performance_report_enable) return nullptr;
  auto& c = gd->perf_halo[cudecompGridDesc::HaloPerfKey{axis, dim, (int)dtype, halo, periods, padding}];
  return beginSample(h, c, wire_bytes, stream);
}

// harness view (cudecompExtGetTransposeTimings): all retained samples of one op, whatever their configuration
TransposeTimings perfCollect(cudecompGridDesc_t gd, int op) {
  TransposeTimings t;
  bool any = false;
  for (const auto& kv : gd->perf_transpose)
    if (std::get<0>(kv.first) == op) {
      any = any || !kv.second.ring.empty();
      t.calls += kv.second.calls;
      t.pencil_bytes = kv.second.wire_bytes;
    }
  if (!any) return t;
  (void)hipDeviceSynchronize();
  for (const auto& kv : gd->perf_transpose) {
    if (std::get<0>(kv.first) != op) continue;
    const Series s = readSeries(kv.second);
    for (size_t i = 0; i < s.total.size(); ++i) {
      t.pack_ms += s.first[i];
      t.exchange_ms += s.exchange[i];
      t.unpack_ms += s.last[i];
      t.samples++;
    }
  }
  if (t.samples) {
    t.pack_ms /= t.samples;
    t.exchange_ms /= t.samples;
    t.unpack_ms /= t.samples;
    t.total_ms = t.pack_ms + t.exchange_ms + t.unpack_ms;
  }
  return t;
}

// the same for halo updates: all retained samples of (pencil axis, dim); the three phases are pack / exchange / unpack
// in the plain sequence (CUDECOMP_DISABLE_HALO_OVERLAP=1) and [0, whole update, 0] in the overlapped one
TransposeTimings perfCollectHalo(cudecompGridDesc_t gd, int axis, int dim) {
  TransposeTimings t;
  bool any = false;
  for (const auto& kv : gd->perf_halo)
    if (std::get<0>(kv.first) == axis && std::get<1>(kv.first) == dim) {
      any = any || !kv.second.ring.empty();
      t.calls += kv.second.calls;
      t.pencil_bytes = kv.second.wire_bytes;
    }
  if (!any) return t;
  (void)hipDeviceSynchronize();
  for (const auto& kv : gd->perf_halo) {
    if (std::get<0>(kv.first) != axis || std::get<1>(kv.first) != dim) continue;
    const Series s = readSeries(kv.second);
    for (size_t i = 0; i < s.total.size(); ++i) {
      t.pack_ms += s.first[i];
      t.exchange_ms += s.exchange[i];
      t.unpack_ms += s.last[i];
      t.samples++;
    }
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
  for (auto& kv : gd->perf_transpose) {
    for (auto& s : kv.second.ring) s.used = false;
    kv.second.calls = 0;
  }
  for (auto& kv : gd->perf_halo) {
    for (auto& s : kv.second.ring) s.used = false;
    kv.second.calls = 0;
  }
}

void perfDestroy(cudecompGridDesc_t gd) {
  auto drop = [](cudecompGridDesc::PerfCollection& c) {
    for (auto& s : c.ring)
      for (hipEvent_t e : s.ev)
        if (e) (void)hipEventDestroy(e);
  };
  for (auto& kv : gd->perf_transpose) drop(kv.second);
  for (auto& kv : gd->perf_halo) drop(kv.second);
  gd->perf_transpose.clear();
  gd->perf_halo.clear();
}

// collective over the handle's communicator (called from cudecompGridDescDestroy, which is collective)
void perfReport(cudecompHandle_t h, cudecompGridDesc_t gd) {
  if (!h->performance_report_enable) return;
  bool touched = false;
  for (const auto& kv : gd->perf_transpose) touched = touched || !kv.second.ring.empty();
  for (const auto& kv : gd->perf_halo) touched = touched || !kv.second.ring.empty();
  if (touched) (void)hipDeviceSynchronize();

  const std::vector<char> mine = serialise(localRows(gd));
  const int64_t my_size = (int64_t)mine.size();
  std::vector<int64_t> sizes(h->nranks);
  h->boot->allgather(&my_size, sizes.data(), sizeof(int64_t));
  const int64_t slot = *std::max_element(sizes.begin(), sizes.end());
  std::vector<char> padded(slot, 0), all((size_t)slot * h->nranks);
  std::memcpy(padded.data(), mine.data(), mine.size());
  h->boot->allgather(padded.data(), all.data(), (size_t)slot);
  if (h->rank != 0) return;

  std::vector<Merged> rows;
  for (int r = 0; r < h->nranks; ++r)
    for (Row& row : deserialise(all.data() + (size_t)r * slot)) {
      auto it = std::find_if(rows.begin(), rows.end(), [&](const Merged& m) { return m.key == row.key; });
      if (it == rows.end()) {
        rows.emplace_back();
        it = rows.end() - 1;
        it->key = row.key;
      }
      it->ranks.push_back(r);
      it->total.push_back(std::move(row.total));
      it->exchange.push_back(std::move(row.exchange));
      it->wire_bytes.push_back(row.wire_bytes);
    }
  for (Merged& m : rows) finish(m);
  std::sort(rows.begin(), rows.end(), tableOrder);
  const bool have_t = std::any_of(rows.begin(), rows.end(), [](const Merged& m) { return !m.key.is_halo; });
  const bool have_h = std::any_of(rows.begin(), rows.end(), [](const Merged& m) { return m.key.is_halo != 0; });

  printf("CUDECOMP:\nCUDECOMP: ===== Performance Summary =====\nCUDECOMP: Grid Configuration:\n");
  printf("CUDECOMP:\tTranspose backend: %s\n", cudecompTransposeCommBackendToString(gd->config.transpose_comm_backend));
  printf("CUDECOMP:\tHalo backend: %s\n", cudecompHaloCommBackendToString(gd->config.halo_comm_backend));
  printf("CUDECOMP:\tProcess grid: [%d, %d]\n", gd->config.pdims[0], gd->config.pdims[1]);
  printf("CUDECOMP:\tGlobal dimensions: [%d, %d, %d]\n", gd->config.gdims[0], gd->config.gdims[1], gd->config.gdims[2]);
  printf("CUDECOMP:\tMemory order: ");
  for (int ax = 0; ax < 3; ++ax)
    printf("[%d,%d,%d]%s", gd->config.transpose_mem_order[ax][0], gd->config.transpose_mem_order[ax][1],
           gd->config.transpose_mem_order[ax][2], ax < 2 ? "; " : "");
  printf("\nCUDECOMP:\n");
  if (!have_t && !have_h) {
    printf("CUDECOMP: No performance data collected\nCUDECOMP: ================================\nCUDECOMP:\n");
    fflush(stdout);
    return;
  }
  const bool csv = !h->performance_report_write_dir.empty();
  if (have_t) {
    printSummary(gd, rows, false);
    if (csv) writeSummaryCsv(h, gd, rows, false);
  }
  if (have_h) {
    printSummary(gd, rows, true);
    if (csv) writeSummaryCsv(h, gd, rows, true);
  }
  if (h->performance_report_detail > 0) {
    printf("CUDECOMP:\nCUDECOMP: Per-Sample Details:\nCUDECOMP:\n");
    if (have_t) printSamples(h, gd, rows, false);
    if (have_h) printSamples(h, gd, rows, true);
  }
  printf("CUDECOMP: ================================\nCUDECOMP:\n");
  fflush(stdout);
}

}  // namespace cudecomp
