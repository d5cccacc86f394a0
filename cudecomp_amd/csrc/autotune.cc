// autotune.cc -- process-grid and backend selection by measurement.
//
// Entry points and candidate rules follow NVIDIA/cuDecomp (reference src/autotune.cc:82-273 for the
// candidate lists and environment filters, :275-769 and :771-1124 for the two sweeps): every candidate
// process grid x backend runs n_warmup_trials + n_trials X->Y->Z->Y->X cycles (or halo sweeps) through the
// PUBLIC transposes / halo updates, timed with device events, reduced over ranks, and the argmin of the
// weighted average wins (first seen wins ties).
#include <algorithm>
#include <array>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>
#include <string_view>

#include "errors.h"
#include "internal.h"
#include "transport.h"

namespace cudecomp {

namespace {

template <typename B>
std::vector<B> filterByEnv(const char* env_name, const std::vector<std::pair<std::string_view, B>>& names,
                           std::vector<B> available) {
  const char* env = std::getenv(env_name);
  if (!env) return available;
  std::string_view value(env);
  const bool exclude = !value.empty() && value.front() == '^';
  if (exclude) value.remove_prefix(1);
  std::vector<B> listed;
  size_t start = 0;
  for (;;) {
    size_t end = value.find(',', start);
    if (end == std::string_view::npos) end = value.size();
    const auto name = value.substr(start, end - start);
    auto m = std::find_if(names.begin(), names.end(), [&](const auto& e) { return e.first == name; });
    if (m == names.end())
      CD_INVALID_USAGE(std::string(env_name) + " contains unknown or empty backend name '" + std::string(name) + "'");
    listed.push_back(m->second);
    if (end == value.size()) break;
    start = end + 1;
  }
  available.erase(std::remove_if(available.begin(), available.end(),
                                 [&](B b) {
                                   const bool in = std::find(listed.begin(), listed.end(), b) != listed.end();
                                   return exclude ? in : !in;
                                 }),
                  available.end());
  return available;
}

std::pair<int32_t, int32_t> parseRange(const char* env_name) {
  std::string_view v(std::getenv(env_name));
  const size_t comma = v.find(',');
  int32_t lo = 0, hi = 0;
  bool ok = comma != std::string_view::npos && comma > 0 && comma + 1 < v.size() &&
            v.find(',', comma + 1) == std::string_view::npos;
  if (ok) {
    auto a = std::from_chars(v.data(), v.data() + comma, lo);
    auto b = std::from_chars(v.data() + comma + 1, v.data() + v.size(), hi);
    ok = a.ec == std::errc() && a.ptr == v.data() + comma && b.ec == std::errc() && b.ptr == v.data() + v.size() &&
         lo >= 0 && hi > 0 && lo <= hi;
  }
  if (!ok)
    CD_INVALID_USAGE(std::string(env_name) + " must be comma-separated nonnegative min and positive max with min <= max");
  return {lo, hi};
}

}  // namespace

std::vector<cudecompTransposeCommBackend_t> transposeBackendCandidates(const cudecompGridDescAutotuneOptions_t* opt) {
  static const std::vector<std::pair<std::string_view, cudecompTransposeCommBackend_t>> names = {
      {"MPI_P2P", CUDECOMP_TRANSPOSE_COMM_MPI_P2P},       {"MPI_P2P_PL", CUDECOMP_TRANSPOSE_COMM_MPI_P2P_PL},
      {"MPI_A2A", CUDECOMP_TRANSPOSE_COMM_MPI_A2A},       {"NCCL", CUDECOMP_TRANSPOSE_COMM_NCCL},
      {"NCCL_PL", CUDECOMP_TRANSPOSE_COMM_NCCL_PL},       {"NVSHMEM", CUDECOMP_TRANSPOSE_COMM_NVSHMEM},
      {"NVSHMEM_PL", CUDECOMP_TRANSPOSE_COMM_NVSHMEM_PL}, {"NVSHMEM_SM", CUDECOMP_TRANSPOSE_COMM_NVSHMEM_SM}};
  std::vector<cudecompTransposeCommBackend_t> c;
  for (auto& n : names) c.push_back(n.second);
  c = filterByEnv("CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS", names, std::move(c));
  c.erase(std::remove_if(c.begin(), c.end(),
                         [&](auto b) {
                           return (opt->disable_mpi_backends && transposeBackendIsMpi(b)) ||
                                  (opt->disable_nccl_backends && transposeBackendIsRccl(b)) ||
                                  (opt->disable_nvshmem_backends && transposeBackendIsPeer(b));
                         }),
          c.end());
  if (c.empty()) CD_INVALID_USAGE("Transpose backend autotuning has no usable candidates after applying filters");
  return c;
}

std::vector<cudecompHaloCommBackend_t> haloBackendCandidates(const cudecompGridDescAutotuneOptions_t* opt) {
  static const std::vector<std::pair<std::string_view, cudecompHaloCommBackend_t>> names = {
      {"MPI", CUDECOMP_HALO_COMM_MPI},
      {"MPI_BLOCKING", CUDECOMP_HALO_COMM_MPI_BLOCKING},
      {"NCCL", CUDECOMP_HALO_COMM_NCCL},
      {"NVSHMEM", CUDECOMP_HALO_COMM_NVSHMEM},
      {"NVSHMEM_BLOCKING", CUDECOMP_HALO_COMM_NVSHMEM_BLOCKING}};
  std::vector<cudecompHaloCommBackend_t> c;
  for (auto& n : names) c.push_back(n.second);
  c = filterByEnv("CUDECOMP_AUTOTUNE_HALO_BACKENDS", names, std::move(c));
  c.erase(std::remove_if(c.begin(), c.end(),
                         [&](auto b) {
                           return (opt->disable_mpi_backends && haloBackendIsMpi(b)) ||
                                  (opt->disable_nccl_backends && haloBackendIsRccl(b)) ||
                                  (opt->disable_nvshmem_backends && haloBackendIsPeer(b));
                         }),
          c.end());
  if (c.empty()) CD_INVALID_USAGE("Halo backend autotuning has no usable candidates after applying filters");
  return c;
}

std::vector<std::array<int32_t, 2>> pdimCandidates(int nranks, bool col_major) {
  // factor pairs, growing the grid dimension mapped to consecutive ranks first (locality first)
  std::vector<int> factors;
  for (int i = 1; i <= nranks; ++i)
    if (nranks % i == 0) factors.push_back(i);
  std::vector<std::array<int32_t, 2>> c;
  for (int f : factors) c.push_back(col_major ? std::array<int32_t, 2>{f, nranks / f} : std::array<int32_t, 2>{nranks / f, f});
  std::pair<int32_t, int32_t> rows{1, INT32_MAX}, cols{1, INT32_MAX};
  if (std::getenv("CUDECOMP_AUTOTUNE_P_ROW_RANGE")) rows = parseRange("CUDECOMP_AUTOTUNE_P_ROW_RANGE");
  if (std::getenv("CUDECOMP_AUTOTUNE_P_COL_RANGE")) cols = parseRange("CUDECOMP_AUTOTUNE_P_COL_RANGE");
  c.erase(std::remove_if(c.begin(), c.end(),
                         [&](const auto& p) {
                           return p[0] < rows.first || p[0] > rows.second || p[1] < cols.first || p[1] > cols.second;
                         }),
          c.end());
  if (c.empty()) CD_INVALID_USAGE("Process-grid autotuning has no usable candidates after applying filters");
  return c;
}

void autotuneTranspose(cudecompHandle_t, cudecompGridDesc_t, const cudecompGridDescAutotuneOptions_t*, bool, bool) {
  CD_NOT_SUPPORTED("transpose autotuning is not implemented yet");
}
void autotuneHalo(cudecompHandle_t, cudecompGridDesc_t, const cudecompGridDescAutotuneOptions_t*, bool, bool) {
  CD_NOT_SUPPORTED("halo autotuning is not implemented yet");
}

}  // namespace cudecomp
