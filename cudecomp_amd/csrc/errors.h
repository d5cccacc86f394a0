// errors.h -- typed exceptions that the C API turns into cudecompResult_t codes.
// Convention of the boundary (reference include/internal/exceptions.h:63-145, src/cudecomp.cc:416-443):
// nothing is thrown across extern "C"; the message "CUDECOMP:ERROR: file:line kind (detail)" goes
// to stderr and the matching result code is returned.
#pragma once
#include <exception>
#include <string>

#include <hip/hip_runtime_api.h>

#include "cudecomp.h"

namespace cudecomp {

class Error : public std::exception {
 public:
  Error(cudecompResult_t code, const char* kind, const char* file, int line, const std::string& detail) : code_(code) {
    msg_ = std::string("CUDECOMP:ERROR: ") + file + ":" + std::to_string(line) + " " + kind;
    if (!detail.empty()) msg_ += " (" + detail + ")";
    msg_ += "\n";
  }
  const char* what() const noexcept override { return msg_.c_str(); }
  cudecompResult_t code() const { return code_; }

 private:
  cudecompResult_t code_;
  std::string msg_;
};

}  // namespace cudecomp

#define CD_THROW(code, kind, detail) throw ::cudecomp::Error(code, kind, __FILE__, __LINE__, detail)
#define CD_INVALID_USAGE(detail) CD_THROW(CUDECOMP_RESULT_INVALID_USAGE, "Invalid usage.", detail)
#define CD_NOT_SUPPORTED(detail) CD_THROW(CUDECOMP_RESULT_NOT_SUPPORTED, "Not supported.", detail)
#define CD_INTERNAL_ERROR(detail) CD_THROW(CUDECOMP_RESULT_INTERNAL_ERROR, "Internal error.", detail)
#define CD_BOOTSTRAP_ERROR(detail) CD_THROW(CUDECOMP_RESULT_MPI_ERROR, "MPI error.", detail)
#define CD_HIP_ERROR(detail) CD_THROW(CUDECOMP_RESULT_CUDA_ERROR, "CUDA error.", detail)
#define CD_PEER_ERROR(detail) CD_THROW(CUDECOMP_RESULT_NVSHMEM_ERROR, "NVSHMEM error.", detail)

#define CD_CHECK_HIP(expr)                                                                         \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess) CD_THROW(CUDECOMP_RESULT_CUDA_ERROR, "CUDA error.", hipGetErrorString(e__)); \
  } while (0)

#define CD_CHECK_RCCL(expr)                                                                        \
  do {                                                                                             \
    ncclResult_t e__ = (expr);                                                                     \
    if (e__ != ncclSuccess) CD_THROW(CUDECOMP_RESULT_NCCL_ERROR, "NCCL error.", ncclGetErrorString(e__)); \
  } while (0)
