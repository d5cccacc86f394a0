// native_test.h -- shared pieces of the native (C++) test programs in this directory.
//
// transpose_test / halo_test take the SAME command lines, test-file mode and output protocol as the reference's
// tests/cc/transpose_test.cc and tests/cc/halo_test.cc ("command: ...", " PASSED" / " FAILED", "Passed all tests."),
// so the reference's tests/test_runner.py and its case matrices can drive this library unchanged
// (`--launcher_cmd "mpirun -np 4"` works: ranks are discovered from the launcher environment).  The programs are
// written against the public C API only and check every result against the closed-form pencil contents
// (value = gx + X*(gy + Y*gz) at interior cells, -1 elsewhere; periodic wrap for halos).
#pragma once
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#ifdef NATIVE_WITH_MPI
// MPI build of the same programs (make -C tests/native mpi): a real MPI program -- MPI_Init, MPI_COMM_WORLD handed to
// cudecompInit, verdicts reduced with MPI_Allreduce -- linked against libcudecomp_mpi.so, whose MPI_* backend enums
// exchange through MPI (csrc/bootstrap_mpi.cc; reference include/internal/comm_routines.h:325-413, 708-762).  At the end
// rank 0 prints how many transposes took the library's MPI path ("MPI-path transposes: N").
#define MPICH_SKIP_MPICXX 1
#define OMPI_SKIP_MPICXX 1
#include <mpi.h>
#endif
#include "cudecomp.h"
#include "cudecomp_ext.h"

#if defined(R32)
using elem_t = float;
static const cudecompDataType_t kDtype = CUDECOMP_FLOAT;
#elif defined(C32)
using elem_t = std::complex<float>;
static const cudecompDataType_t kDtype = CUDECOMP_FLOAT_COMPLEX;
#elif defined(C64)
using elem_t = std::complex<double>;
static const cudecompDataType_t kDtype = CUDECOMP_DOUBLE_COMPLEX;
#else
using elem_t = double;
static const cudecompDataType_t kDtype = CUDECOMP_DOUBLE;
#endif

inline void make(float& e, double v) { e = (float)v; }
inline void make(double& e, double v) { e = v; }
inline void make(std::complex<float>& e, double v) { e = std::complex<float>((float)v, (float)-v); }
inline void make(std::complex<double>& e, double v) { e = std::complex<double>(v, -v); }

struct TestFailure : std::runtime_error {
  using std::runtime_error::runtime_error;
};
#define T_CHECK_HIP(x)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (x);                                                                                 \
    if (e_ != hipSuccess) throw TestFailure(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #x); \
  } while (0)
#define T_CHECK_CD(x)                                                                                    \
  do {                                                                                                   \
    cudecompResult_t r_ = (x);                                                                           \
    if (r_ != CUDECOMP_RESULT_SUCCESS) throw TestFailure("cuDecomp error " + std::to_string((int)r_) + " at " #x); \
  } while (0)

// ---- command lines: "--name v1 [v2 v3 ...]" and single-letter flags, from argv or from one line of a test file ----
struct Options {
  std::map<std::string, std::vector<std::string>> values;
  bool has(const std::string& k) const { return values.count(k) != 0; }
  int geti(const std::string& k, int dflt, size_t idx = 0) const {
    auto it = values.find(k);
    return (it == values.end() || it->second.size() <= idx) ? dflt : std::atoi(it->second[idx].c_str());
  }
  std::array<int, 3> get3(const std::string& k, std::array<int, 3> dflt) const {
    auto it = values.find(k);
    if (it == values.end() || it->second.size() < 3) return dflt;
    return {std::atoi(it->second[0].c_str()), std::atoi(it->second[1].c_str()), std::atoi(it->second[2].c_str())};
  }
};

inline bool isOptionToken(const std::string& t) {
  return t.size() >= 2 && t[0] == '-' && !(t[1] >= '0' && t[1] <= '9');
}

inline Options parseOptions(const std::string& line) {
  Options o;
  std::istringstream in(line);
  std::string tok, cur;
  while (in >> tok) {
    if (isOptionToken(tok)) {
      cur = tok.substr(tok[1] == '-' ? 2 : 1);
      o.values[cur];
    } else if (!cur.empty()) {
      o.values[cur].push_back(tok);
    }
  }
  return o;
}

// ---- launcher environment ---------------------------------------------------------------------------------------
inline int envRankOr(const char* const* names, int dflt) {
  for (int i = 0; names[i]; ++i)
    if (const char* v = std::getenv(names[i])) return std::atoi(v);
  return dflt;
}
inline int envIntOr(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}
#ifdef NATIVE_WITH_MPI
inline int worldRank() {
  int r = 0;
  MPI_Comm_rank(MPI_COMM_WORLD, &r);
  return r;
}
inline int worldSize() {
  int n = 1;
  MPI_Comm_size(MPI_COMM_WORLD, &n);
  return n;
}
inline int localRank() {
  static const char* n[] = {"MPI_LOCALRANKID", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", nullptr};
  return envRankOr(n, worldRank());
}
#else
inline int worldRank() {
  static const char* n[] = {"RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID", nullptr};
  return envRankOr(n, 0);
}
inline int worldSize() {
  static const char* n[] = {"WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE", "SLURM_NTASKS", nullptr};
  return envRankOr(n, 1);
}
inline int localRank() {
  static const char* n[] = {"LOCAL_RANK", "MPI_LOCALRANKID", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", nullptr};
  return envRankOr(n, worldRank());
}
#endif

// where the time of a case goes (CUDECOMP_TEST_PHASE_TIMES=1 prints the sums of rank 0 at the end)
struct PhaseTimes {
  static constexpr int N = 8;
  double sum[N] = {0, 0, 0, 0, 0, 0, 0, 0};
  const char* name[N] = {"create", "alloc", "fill+h2d", "transposes", "sync", "d2h+compare", "free", "destroy"};
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void start() { t = std::chrono::steady_clock::now(); }
  void mark(int i) {
    const auto now = std::chrono::steady_clock::now();
    sum[i] += std::chrono::duration<double>(now - t).count();
    t = now;
  }
};
inline PhaseTimes& phaseTimes() {
  static PhaseTimes p;
  return p;
}

// path counters of the library, summed over the cases of this process (printed at the end by the MPI build)
inline int64_t& mpiPathTransposes() {
  static int64_t n = 0;
  return n;
}
inline int64_t (&queueCensus())[2] {  // max compute queues seen on the device, its hardware queue slots
  static int64_t v[2] = {-1, 0};
  return v;
}
inline void notePaths(cudecompHandle_t handle, cudecompGridDesc_t gdesc) {
  cudecompExtCounters_t c;
  if (cudecompExtGetCounters(handle, gdesc, &c) != CUDECOMP_RESULT_SUCCESS) return;
  mpiPathTransposes() += c.mpi;
}
inline void noteQueues(cudecompHandle_t handle) {  // a fresh census: not per case (it is not free), see nativeMain
  int32_t c = -1, s = 0;
  if (cudecompExtQueueCensus(handle, &c, &s) != CUDECOMP_RESULT_SUCCESS) return;
  if (c > queueCensus()[0]) queueCensus()[0] = c;
  queueCensus()[1] = s;
}

// Verdict of a case over all ranks without MPI: every rank drops a one-byte file into a job directory under /dev/shm,
// rank 0 collects them (the ranks of these tests share a node).  Returns the maximum over ranks on rank 0.
inline int reduceVerdict(int mine, int case_index) {
#ifdef NATIVE_WITH_MPI
  (void)case_index;
  int over_all = mine;
  MPI_Allreduce(&mine, &over_all, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
  return over_all;
#else
  const int rank = worldRank(), n = worldSize();
  if (n == 1) return mine;
  const char* job = std::getenv("CUDECOMP_TEST_JOB");
  if (!job) job = std::getenv("CUDECOMP_BOOTSTRAP_PORT");
  if (!job) job = std::getenv("MASTER_PORT");
  if (!job) job = std::getenv("PMI_ID");
  const std::string dir = std::string("/dev/shm/cudecomp_native_") + (job ? job : "job");
  ::mkdir(dir.c_str(), 0700);
  auto name = [&](int r) { return dir + "/case" + std::to_string(case_index) + "_rank" + std::to_string(r); };
  {
    const std::string tmp = name(rank) + ".tmp";
    std::ofstream(tmp) << mine;
    ::rename(tmp.c_str(), name(rank).c_str());
  }
  if (rank != 0) return mine;
  int worst = mine;
  // (how long rank 0 waits for a rank that has died without a verdict: CUDECOMP_TEST_VERDICT_TIMEOUT seconds, default 300)
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(envIntOr("CUDECOMP_TEST_VERDICT_TIMEOUT", 300));
  for (int r = 0; r < n; ++r) {
    for (;;) {
      std::ifstream f(name(r));
      int v;
      if (f && (f >> v)) {
        worst = std::max(worst, v);
        break;
      }
      if (std::chrono::steady_clock::now() > deadline) return 1;
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    ::unlink(name(r).c_str());
  }
  return worst;
#endif
}

// ---- closed-form pencil contents ------------------------------------------------------------------------------------
// interior cells hold their global linear index; `wrap`: halo cells hold the periodic neighbour's value (or -1
// where the domain ends), padding -1.  Without `wrap` everything outside the interior is -1.
inline void fillPencil(std::vector<elem_t>& out, const cudecompPencilInfo_t& p, const std::array<int, 3>& g, bool wrap,
                       const std::array<bool, 3>& periods) {
  out.resize(p.size);
  int64_t idx = 0;
  for (int i2 = 0; i2 < p.shape[2]; ++i2)
    for (int i1 = 0; i1 < p.shape[1]; ++i1)
      for (int i0 = 0; i0 < p.shape[0]; ++i0, ++idx) {
        const int l[3] = {i0, i1, i2};
        int64_t gc[3];
        bool valid = true;
        for (int k = 0; k < 3; ++k) {
          const int ax = p.order[k], h = p.halo_extents[ax];
          const int interior = p.hi[k] - p.lo[k] + 1;
          int64_t c = p.lo[k] + (l[k] - h);
          if (l[k] >= interior + 2 * h) valid = false;  // padding
          if (l[k] < h || l[k] >= interior + h) {       // halo cell
            if (!wrap) valid = false;
            else if (c < 0 || c >= g[ax]) {
              if (periods[ax]) c = ((c % g[ax]) + g[ax]) % g[ax];
              else valid = false;
            }
          }
          gc[ax] = c;
        }
        make(out[idx], valid ? (double)(gc[0] + g[0] * (gc[1] + (int64_t)g[1] * gc[2])) : -1.0);
      }
}

// compare: the interior only (transposes leave halos unspecified) or the whole pencil (halo updates)
inline int64_t countMismatches(const std::vector<elem_t>& got, const std::vector<elem_t>& ref, const cudecompPencilInfo_t& p,
                               bool interior_only) {
  int64_t bad = 0, idx = 0;
  for (int i2 = 0; i2 < p.shape[2]; ++i2)
    for (int i1 = 0; i1 < p.shape[1]; ++i1)
      for (int i0 = 0; i0 < p.shape[0]; ++i0, ++idx) {
        if (interior_only) {
          const int l[3] = {i0, i1, i2};
          bool inside = true;
          for (int k = 0; k < 3; ++k) {
            const int h = p.halo_extents[p.order[k]];
            if (l[k] < h || l[k] >= (p.hi[k] - p.lo[k] + 1) + h) inside = false;
          }
          if (!inside) continue;
        }
        if (!(got[idx] == ref[idx])) ++bad;
      }
  return bad;
}

// ---- failure diagnostics (the 8-shared-rank hunt) -------------------------------------------------
// When a result differs, say WHAT the wrong cells hold and WHO can see the right ones, so that one failing case tells
// "stores late" from "stores lost" from "stores visible through one XCD only":
//   1. classify the wrong cells of the first read: sentinel (CUDECOMP_TEST_SENTINEL=1 pre-fills out-of-place outputs),
//      the buffer's previous content, -1, anything else; index range;
//   2. read again with hipMemcpy 1 ms later (late vs lost);
//   3. read the pencil with a KERNEL, eight passes, pass x using only the workgroups that run on XCD x (workgroup b runs on
//      XCD b % 8; HW_REG_XCC_ID is recorded to check that), into pinned host memory: a result that only one XCD can see sits
//      in that XCD's L2 or behind that XCD's address translation;
//   4. read a third time with hipMemcpy: the end of the reader kernels wrote every L2 back, so data that was merely
//      parked in an L2 is in memory now, data behind a stale translation is not.
__global__ void diag_read_k(const elem_t* src, elem_t* dst, long long n, int xcd, unsigned int* xcc_seen) {
  if ((int)(blockIdx.x % 8) != xcd) return;
  if (threadIdx.x == 0) {
    const unsigned int id = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xfu;  // HW_REG_XCC_ID[3:0]
    atomicOr(xcc_seen, 1u << id);
  }
  const long long nb = gridDim.x / 8, b = blockIdx.x / 8;
  for (long long i = b * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) dst[i] = src[i];
}

inline bool sentinelRequested() {
  const char* v = std::getenv("CUDECOMP_TEST_SENTINEL");
  return v && std::atoi(v) != 0;
}
inline elem_t sentinelValue() {
  elem_t e;
  std::memset(&e, 0xEE, sizeof(e));
  return e;
}
inline bool sameBits(const elem_t& a, const elem_t& b) { return std::memcmp(&a, &b, sizeof(elem_t)) == 0; }

inline void diagnoseMismatch(const char* what, const elem_t* dev, const std::vector<elem_t>& first,
                             const std::vector<elem_t>& ref, const std::vector<elem_t>* previous,
                             const cudecompPencilInfo_t& p, bool interior_only) {
  const int rank = worldRank();
  const int64_t n = p.size;
  const elem_t sent = sentinelValue();
  elem_t minus1;
  make(minus1, -1.0);
  auto inside = [&](int64_t idx) {
    if (!interior_only) return true;
    int64_t l[3] = {idx % p.shape[0], (idx / p.shape[0]) % p.shape[1], idx / ((int64_t)p.shape[0] * p.shape[1])};
    for (int k = 0; k < 3; ++k) {
      const int h = p.halo_extents[p.order[k]];
      if (l[k] < h || l[k] >= (p.hi[k] - p.lo[k] + 1) + h) return false;
    }
    return true;
  };
  auto countBad = [&](const std::vector<elem_t>& got) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i)
      if (inside(i) && !(got[i] == ref[i])) ++bad;
    return bad;
  };
  int64_t bad = 0, lo = -1, hi = -1, n_sent = 0, n_prev = 0, n_m1 = 0, runs = 0, last_bad = -2;
  for (int64_t i = 0; i < n; ++i) {
    if (!inside(i) || first[i] == ref[i]) continue;
    ++bad;
    if (lo < 0) lo = i;
    hi = i;
    if (i != last_bad + 1) ++runs;
    last_bad = i;
    if (sameBits(first[i], sent)) ++n_sent;
    else if (previous && (int64_t)previous->size() > i && sameBits(first[i], (*previous)[i])) ++n_prev;
    else if (sameBits(first[i], minus1)) ++n_m1;
  }
  const int64_t plane = (int64_t)p.shape[0] * p.shape[1];
  fprintf(stderr,
          "DIAG rank %d %s: %lld wrong cells in [%lld, %lld] (planes %lld..%lld of %d, %lld runs, pencil %d x %d x %d, dev %p): "
          "%lld hold the sentinel, %lld the buffer's previous content, %lld -1, %lld something else\n",
          rank, what, (long long)bad, (long long)lo, (long long)hi, (long long)(lo / plane), (long long)(hi / plane), p.shape[2],
          (long long)runs, p.shape[0], p.shape[1], p.shape[2], (const void*)dev, (long long)n_sent, (long long)n_prev,
          (long long)n_m1, (long long)(bad - n_sent - n_prev - n_m1));
  std::vector<elem_t> again(n);
  std::this_thread::sleep_for(std::chrono::milliseconds(1));
  if (hipMemcpy(again.data(), dev, n * sizeof(elem_t), hipMemcpyDeviceToHost) != hipSuccess) return;
  const int64_t bad1 = countBad(again);
  elem_t* pinned = nullptr;
  unsigned int* seen = nullptr;
  long long per_xcd[8];
  unsigned int seen_mask[8];
  if (hipHostMalloc((void**)&pinned, n * sizeof(elem_t), hipHostMallocMapped) == hipSuccess &&
      hipHostMalloc((void**)&seen, 64, hipHostMallocMapped) == hipSuccess) {
    for (int x = 0; x < 8; ++x) {
      std::memset(pinned, 0x5a, n * sizeof(elem_t));
      *seen = 0;
      diag_read_k<<<8 * 64, 256>>>(dev, pinned, n, x, seen);
      (void)hipDeviceSynchronize();
      int64_t b = 0;
      for (int64_t i = 0; i < n; ++i)
        if (inside(i) && !(pinned[i] == ref[i])) ++b;
      per_xcd[x] = b;
      seen_mask[x] = *seen;
    }
    (void)hipHostFree(pinned);
    (void)hipHostFree(seen);
  } else {
    for (int x = 0; x < 8; ++x) per_xcd[x] = -1, seen_mask[x] = 0;
  }
  int64_t bad2 = -1;
  if (hipMemcpy(again.data(), dev, n * sizeof(elem_t), hipMemcpyDeviceToHost) == hipSuccess) bad2 = countBad(again);
  fprintf(stderr,
          "DIAG rank %d %s: wrong cells: first read %lld, hipMemcpy 1 ms later %lld, kernel readers on XCD 0..7: %lld %lld %lld %lld "
          "%lld %lld %lld %lld (XCC_ID masks %x %x %x %x %x %x %x %x), hipMemcpy after the reader kernels %lld\n",
          rank, what, (long long)bad, (long long)bad1, per_xcd[0], per_xcd[1], per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5],
          per_xcd[6], per_xcd[7], seen_mask[0], seen_mask[1], seen_mask[2], seen_mask[3], seen_mask[4], seen_mask[5], seen_mask[6],
          seen_mask[7], (long long)bad2);
}

// How the test programs upload a pencil.  Default: the synchronous hipMemcpy from pageable memory the reference's programs use.
// CUDECOMP_TEST_UPLOAD=sync adds a hipDeviceSynchronize behind it; =pinned goes through a pinned staging buffer with
// hipMemcpyAsync + hipStreamSynchronize on the null stream.  Arms of the hunt (did a consumer kernel ever
// run before the pageable upload had landed?).
inline void uploadPencil(void* dst, const void* src, size_t bytes) {
  static const int mode = [] {
    const char* v = std::getenv("CUDECOMP_TEST_UPLOAD");
    return !v ? 0 : (!std::strcmp(v, "sync") ? 1 : (!std::strcmp(v, "pinned") ? 2 : 0));
  }();
  if (mode == 2) {
    static void* pinned = nullptr;
    static size_t cap = 0;
    if (cap < bytes) {
      if (pinned) T_CHECK_HIP(hipHostFree(pinned));
      cap = std::max<size_t>(bytes, (size_t)16 << 20);
      T_CHECK_HIP(hipHostMalloc(&pinned, cap, hipHostMallocDefault));
    }
    std::memcpy(pinned, src, bytes);
    T_CHECK_HIP(hipMemcpyAsync(dst, pinned, bytes, hipMemcpyHostToDevice, nullptr));
    T_CHECK_HIP(hipStreamSynchronize(nullptr));
    return;
  }
  T_CHECK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  if (mode == 1) T_CHECK_HIP(hipDeviceSynchronize());
}

// ---- input-integrity gate (round 5) ------------------------------------------------------------------
// Decides the "upload hypothesis" behind the rare wrong results of ranks sharing a GPU: did a consumer kernel ever see a
// pencil whose (synchronous, pageable) upload had not landed?  Right after uploadPencil a checksum KERNEL reads the whole
// pencil on the stream the library call will use and the sum is compared with the host's sum of what was uploaded, BEFORE the
// call: a difference prints "DIAG ... input stale before call".  After the call the same kernel sums the result on that
// stream and the sum is compared with the host copy the verdict is computed from ("download differs from the device's view").
// On by default (one 1-wave-per-CU kernel and an 8-byte copy per check); CUDECOMP_TEST_INPUT_GATE=0 switches it off.
// The sum is position-weighted (a stale contiguous part cannot cancel) and order-independent (mod 2^64).
__global__ void gate_sum_k(const unsigned int* __restrict__ w, long long nwords, unsigned long long* out) {
  unsigned long long acc = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x)
    acc += (unsigned long long)(w[i] ^ (unsigned int)((unsigned long long)i * 2654435761ull)) * (unsigned long long)(2 * i + 1);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// the same sum by the workgroups of ONE XCD only (workgroup b runs on XCD b % 8): which XCD's view of the buffer is stale?
__global__ void gate_sum_xcd_k(const unsigned int* __restrict__ w, long long nwords, unsigned long long* out, int xcd) {
  if ((int)(blockIdx.x % 8) != xcd) return;
  const long long nb = gridDim.x / 8, b = blockIdx.x / 8;
  unsigned long long acc = 0;
  for (long long i = b * blockDim.x + threadIdx.x; i < nwords; i += nb * blockDim.x)
    acc += (unsigned long long)(w[i] ^ (unsigned int)((unsigned long long)i * 2654435761ull)) * (unsigned long long)(2 * i + 1);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

struct InputGate {
  long long checks = 0, stale_inputs = 0, stale_downloads = 0, interior_overwritten = 0;
  unsigned long long* d_sum = nullptr;
  unsigned long long* h_sum = nullptr;
  static InputGate& get() {
    static InputGate g;
    return g;
  }
  static bool enabled() {
    static const bool on = [] { const char* v = std::getenv("CUDECOMP_TEST_INPUT_GATE"); return !v || std::atoi(v) != 0; }();
    return on;
  }
  static unsigned long long hostSum(const void* p, size_t bytes) {
    const unsigned int* w = static_cast<const unsigned int*>(p);
    const long long n = (long long)(bytes / 4);
    unsigned long long acc = 0;
    for (long long i = 0; i < n; ++i)
      acc += (unsigned long long)(w[i] ^ (unsigned int)((unsigned long long)i * 2654435761ull)) * (unsigned long long)(2 * i + 1);
    return acc;
  }
  unsigned long long deviceSum(const void* dev, size_t bytes, hipStream_t stream) {
    if (!d_sum) {
      T_CHECK_HIP(hipMalloc((void**)&d_sum, 8));
      T_CHECK_HIP(hipHostMalloc((void**)&h_sum, 8, hipHostMallocDefault));
    }
    T_CHECK_HIP(hipMemsetAsync(d_sum, 0, 8, stream));
    gate_sum_k<<<256, 256, 0, stream>>>(static_cast<const unsigned int*>(dev), (long long)(bytes / 4), d_sum);
    T_CHECK_HIP(hipGetLastError());
    T_CHECK_HIP(hipMemcpyAsync(h_sum, d_sum, 8, hipMemcpyDeviceToHost, stream));
    T_CHECK_HIP(hipStreamSynchronize(stream));
    return *h_sum;
  }
  // BEFORE the call: does a kernel on `stream` see what the host uploaded?  Returns true when the input was stale.
  bool checkInput(const char* what, const elem_t* dev, const std::vector<elem_t>& uploaded, int64_t nel, hipStream_t stream) {
    if (!enabled()) return false;
    ++checks;
    const size_t bytes = (size_t)nel * sizeof(elem_t);
    const unsigned long long want = hostSum(uploaded.data(), bytes), got = deviceSum(dev, bytes, stream);
    if (got == want) return false;
    ++stale_inputs;
    // how stale, and is it merely late?  a second kernel look after a device-wide sync, then the cells themselves
    T_CHECK_HIP(hipDeviceSynchronize());
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
    const unsigned long long again = deviceSum(dev, bytes, stream);
    // which XCD disagrees?  the same sum computed by the workgroups of one XCD at a time
    char per_xcd[8 * 12 + 1] = "";
    for (int x = 0; x < 8; ++x) {
      T_CHECK_HIP(hipMemsetAsync(d_sum, 0, 8, stream));
      gate_sum_xcd_k<<<8 * 32, 256, 0, stream>>>(static_cast<const unsigned int*>(static_cast<const void*>(dev)), (long long)(bytes / 4), d_sum, x);
      T_CHECK_HIP(hipMemcpyAsync(h_sum, d_sum, 8, hipMemcpyDeviceToHost, stream));
      T_CHECK_HIP(hipStreamSynchronize(stream));
      std::snprintf(per_xcd + std::strlen(per_xcd), 12, " %s", *h_sum == want ? "ok" : "STALE");
    }
    std::vector<elem_t> back((size_t)nel);
    T_CHECK_HIP(hipMemcpy(back.data(), dev, bytes, hipMemcpyDeviceToHost));
    int64_t bad = 0, lo = -1, hi = -1;
    for (int64_t i = 0; i < nel; ++i)
      if (!sameBits(back[i], uploaded[i])) {
        ++bad;
        if (lo < 0) lo = i;
        hi = i;
      }
    fprintf(stderr,
            "DIAG rank %d %s: input stale before call: kernel checksum %016llx, host %016llx; second kernel look after a device sync "
            "%s (%016llx); per XCD 0..7:%s; read back with hipMemcpy: %lld of %lld cells differ from the upload, in [%lld, %lld] (dev %p)\n",
            worldRank(), what, got, want, again == want ? "RIGHT (the upload landed late)" : (again == got ? "the SAME wrong sum" : "another wrong sum"),
            again, per_xcd, (long long)bad, (long long)nel, (long long)lo, (long long)hi, (const void*)dev);
    return true;
  }
  // AFTER the call (device idle): does the host copy the verdict is computed from agree with what a kernel sees?
  void checkDownload(const char* what, const elem_t* dev, const std::vector<elem_t>& host, int64_t nel, hipStream_t stream) {
    if (!enabled()) return;
    const size_t bytes = (size_t)nel * sizeof(elem_t);
    const unsigned long long want = hostSum(host.data(), bytes), got = deviceSum(dev, bytes, stream);
    if (got == want) return;
    ++stale_downloads;
    fprintf(stderr, "DIAG rank %d %s: download differs from the device's view: kernel checksum %016llx, host copy %016llx (dev %p)\n",
            worldRank(), what, got, want, (const void*)dev);
  }
  void report() const {
    if (!enabled()) return;
    if (worldRank() == 0 || stale_inputs || stale_downloads || interior_overwritten)
      fprintf(stderr, "Input gate rank %d: %lld uploads checked, %lld stale before the call, %lld downloads differing, %lld halo updates that overwrote interior cells\n",
              worldRank(), checks, stale_inputs, stale_downloads, interior_overwritten);
  }
};

// Data buffers of the test programs: grown once and kept for the process (default), or -- CUDECOMP_TEST_REUSE_BUFFERS=0 --
// hipMalloc / hipFree per case as the reference's programs do.  Why the default deviates from the reference: with eight
// processes SHARING one GPU, a buffer that was freed and re-allocated per case is sometimes seen STALE by exactly one XCD --
// the input-integrity gate below caught it before any library call: the per-XCD checksums of a freshly uploaded pencil read
// "ok ok ok ok ok ok STALE ok", the same wrong sum after a device synchronisation, while hipMemcpy reads the right data back
// (profiles/r05_stale_xcd_view.md).  The reference's 8-rank transpose_test_cc list (R64) failed in 7 of 7
// runs with per-case allocation and in 0 of 3 with kept buffers, with this round's library and with round 4's alike.
struct TestBuffer {
  static bool reuse() {
    static const bool r = [] { const char* v = std::getenv("CUDECOMP_TEST_REUSE_BUFFERS"); return !v || std::atoi(v) != 0; }();
    return r;
  }
  static elem_t* get(int slot, int64_t nel) {
    if (!reuse()) {
      elem_t* q = nullptr;
      T_CHECK_HIP(hipMalloc((void**)&q, nel * sizeof(elem_t)));
      return q;
    }
    static elem_t* kept[4] = {nullptr, nullptr, nullptr, nullptr};
    static int64_t cap[4] = {0, 0, 0, 0};
    if (cap[slot] < nel) {
      if (kept[slot]) T_CHECK_HIP(hipFree(kept[slot]));
      const int64_t want = std::max<int64_t>(nel, 1 << 22);
      T_CHECK_HIP(hipMalloc((void**)&kept[slot], want * sizeof(elem_t)));
      cap[slot] = want;
    }
    return kept[slot];
  }
  static void put(elem_t* q) {
    if (!reuse() && q) T_CHECK_HIP(hipFree(q));
  }
};

inline std::vector<std::string> readTestFile(const std::string& path) {
  std::vector<std::string> lines;
  std::ifstream f(path);
  if (!f) throw TestFailure("cannot open test file " + path);
  std::string line;
  while (std::getline(f, line))
    if (line.find_first_not_of(" \t\r") != std::string::npos) lines.push_back(line);
  return lines;
}

// main loop shared by both programs: argv or --testfile, the reference's output protocol
template <typename RunCase>
int nativeMain(int argc, char** argv, RunCase run_case) {
#ifdef NATIVE_WITH_MPI
  MPI_Init(&argc, &argv);
#endif
  const int rank = worldRank();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "No HIP devices available.\n");
    return EXIT_FAILURE;
  }
  (void)hipSetDevice(localRank() % ndev);

  std::string testfile;
  for (int i = 1; i + 1 < argc; ++i)
    if (!strcmp(argv[i], "-f") || !strcmp(argv[i], "--testfile")) testfile = argv[i + 1];
  std::vector<std::string> cases;
  const bool from_file = !testfile.empty();
  if (from_file) {
    cases = readTestFile(testfile);
  } else {
    std::string line;
    for (int i = 1; i < argc; ++i) line += std::string(i > 1 ? " " : "") + argv[i];
    cases.push_back(line);
  }

  cudecompHandle_t handle;
  if (cudecompInit(&handle, MPI_COMM_WORLD) != CUDECOMP_RESULT_SUCCESS) return EXIT_FAILURE;
  std::vector<std::string> failed;
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (from_file && rank == 0) printf("Running %d tests...\n", (int)cases.size());
  int any_local_failure = 0;
  const bool stop_at_first_failure = envIntOr("CUDECOMP_TEST_STOP_AT_FIRST_FAILURE", 0) != 0;
  for (size_t i = 0; i < cases.size(); ++i) {
    if (from_file && rank == 0) printf("command: %s %s\n", argv[0], cases[i].c_str());
    int res = 1;
    try {
      res = run_case(handle, parseOptions(cases[i]), from_file);
    } catch (const std::exception& e) {
      fprintf(stderr, "rank %d: %s\n", rank, e.what());
    }
    any_local_failure |= res;
    if (rank == 0 && (res || i == cases.size() / 2)) noteQueues(handle);  // at failures, and once while everybody is busy
    const int mine = res;
    res = reduceVerdict(res, (int)i);
    if (rank == 0) {
      if (from_file) printf(res ? " FAILED\n" : " PASSED\n");
      if (res) failed.push_back(cases[i]);
      if (from_file && (i + 1) % 10 == 0)
        printf("Completed %d/%d tests, running time %f s\n", (int)i + 1, (int)cases.size(), elapsed());
      fflush(stdout);
    }
    // CUDECOMP_TEST_STOP_AT_FIRST_FAILURE=1 (the pytest harness): a rank leaves the list at the first case that failed for it
    // (rank 0: for anybody).  After a failed case the ranks are no longer in step -- a rank that threw skipped collectives the others
    // entered -- so the cases that follow fail for that reason and, with peers gone, only after time-outs.  The reference's
    // protocol (run everything, list the failing cases) stays the default.
    if (stop_at_first_failure && (mine || (rank == 0 && res))) {
      if (rank == 0) printf("Stopping at the first failing case (%d of %d run).\n", (int)i + 1, (int)cases.size());
      break;
    }
  }
  InputGate::get().report();
  (void)cudecompFinalize(handle);
#ifdef NATIVE_WITH_MPI
  {
    long long mine = (long long)mpiPathTransposes(), all = 0;
    MPI_Reduce(&mine, &all, 1, MPI_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
    if (rank == 0) printf("MPI-path transposes: %lld\n", all);
    MPI_Finalize();
  }
#endif
  if (rank == 0 && std::getenv("CUDECOMP_TEST_PHASE_TIMES")) {
    printf("Phase times [s]:");
    for (int i = 0; i < PhaseTimes::N; ++i) printf(" %s %.3f", phaseTimes().name[i], phaseTimes().sum[i]);
    printf("\n");
  }
  if (rank == 0 && queueCensus()[0] >= 0)  // (ranks sharing a GPU: more compute queues than slots = the driver time-slices them)
    printf("Device queues: at most %lld compute queues of all processes on this GPU, %lld hardware queue slots\n",
           (long long)queueCensus()[0], (long long)queueCensus()[1]);
  if (rank == 0) {
    if (from_file) printf("Completed all tests, running time %f s,\n", elapsed());
    if (failed.empty()) {
      printf(from_file ? "Passed all tests.\n" : "PASSED\n");
    } else {
      printf("Failed %d/%d tests. Failing cases:\n", (int)failed.size(), (int)cases.size());
      for (auto& c : failed) printf("%s %s\n", argv[0], c.c_str());
    }
    return failed.empty() ? EXIT_SUCCESS : EXIT_FAILURE;
  }
  return any_local_failure ? EXIT_FAILURE : EXIT_SUCCESS;
}
