// bootstrap_mpi.cc -- MPI flavour (make MPI=1): the control plane over a real MPI communicator and the
// MPI_* transports: ROCm-aware MPI on device pointers when CUDECOMP_MPI_GPU_AWARE=1, otherwise staged
// through pinned host buffers (works with any MPI, e.g. the host-only MPICH of this image).
// Replaces reference include/internal/comm_routines.h:325-413 (MPI_P2P / MPI_A2A), :708-762 (halo MPI).
#ifdef CUDECOMP_WITH_MPI
#include <mpi.h>

#include <cstdlib>
#include <cstring>
#include <limits>

#include "errors.h"
#include "transport.h"

namespace cudecomp {

#define CD_CHECK_MPI(expr)                                                                   \
  do {                                                                                       \
    int e__ = (expr);                                                                        \
    if (e__ != MPI_SUCCESS) {                                                                \
      char msg__[MPI_MAX_ERROR_STRING];                                                      \
      int len__ = 0;                                                                         \
      MPI_Error_string(e__, msg__, &len__);                                                  \
      CD_THROW(CUDECOMP_RESULT_MPI_ERROR, "MPI error.", std::string(msg__, len__));          \
    }                                                                                        \
  } while (0)

namespace {

class MpiBootstrap : public Bootstrap {
 public:
  MpiBootstrap(MPI_Comm comm, bool owned) : comm_(comm), owned_(owned) {
    CD_CHECK_MPI(MPI_Comm_rank(comm_, &rank_));
    CD_CHECK_MPI(MPI_Comm_size(comm_, &size_));
  }
  ~MpiBootstrap() override {
    int finalized = 0;
    MPI_Finalized(&finalized);
    if (owned_ && !finalized && comm_ != MPI_COMM_NULL) MPI_Comm_free(&comm_);
  }
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  void allgather(const void* send, void* recv, size_t bytes) override {
    if (bytes > (size_t)std::numeric_limits<int>::max()) CD_BOOTSTRAP_ERROR("allgather payload too large");
    CD_CHECK_MPI(MPI_Allgather(send, (int)bytes, MPI_BYTE, recv, (int)bytes, MPI_BYTE, comm_));
  }
  std::unique_ptr<Bootstrap> split(int color, int key) override {
    MPI_Comm sub;
    CD_CHECK_MPI(MPI_Comm_split(comm_, color, key, &sub));
    return std::make_unique<MpiBootstrap>(sub, true);
  }
  void* nativeComm() override { return &comm_; }

 private:
  MPI_Comm comm_;
  bool owned_;
  int rank_ = 0, size_ = 1;
};

MPI_Datatype mpiType(int es) { return es == 4 ? MPI_FLOAT : (es == 8 ? MPI_DOUBLE : MPI_C_DOUBLE_COMPLEX); }

int toInt(i64 v, const char* what) {
  if (v > std::numeric_limits<int32_t>::max())
    CD_NOT_SUPPORTED(std::string("MPI count and/or offset argument exeeding int32_t limit in ") + what);
  return (int)v;
}

bool gpuAware() {
  const char* v = std::getenv("CUDECOMP_MPI_GPU_AWARE");
  return v && std::strtol(v, nullptr, 10) == 1;
}

struct HostStage {  // pinned bounce buffers, grown on demand, one pair per process
  char *send = nullptr, *recv = nullptr;
  size_t send_bytes = 0, recv_bytes = 0;
  void grow(size_t s, size_t r) {
    if (s > send_bytes) {
      if (send) (void)hipHostFree(send);
      CD_CHECK_HIP(hipHostMalloc((void**)&send, s, hipHostMallocDefault));
      send_bytes = s;
    }
    if (r > recv_bytes) {
      if (recv) (void)hipHostFree(recv);
      CD_CHECK_HIP(hipHostMalloc((void**)&recv, r, hipHostMallocDefault));
      recv_bytes = r;
    }
  }
};
HostStage g_stage;

}  // namespace

std::unique_ptr<Bootstrap> makeWorldBootstrap(MPI_Comm comm, int instance) {
  int inited = 0;
  MPI_Initialized(&inited);
  if (!inited) {
    // e.g. a torchrun-launched harness using the MPI flavour: fall back to the environment bootstrap
    const LaunchEnv env = detectLaunchEnv();
    if (env.size == 1) return makeLocalBootstrap();
    return makeTcpBootstrap(env, instance);
  }
  if (comm == MPI_COMM_NULL) CD_INVALID_USAGE("null communicator");
  return std::make_unique<MpiBootstrap>(comm, false);
}

MPI_Comm commFromFortran(MPI_Fint f) { return MPI_Comm_f2c(f); }

bool mpiTransportAvailable(cudecompCommInfo& ci) { return ci.boot && ci.boot->nativeComm() != nullptr; }

// all-to-all(v) of the plan's chunks; host blocks (the reference's MPI backends do too)
void mpiAlltoall(cudecompHandle_t, cudecompCommInfo& ci, const TransposePlan& p, const ExchangeBuffers& b, int es,
                 hipStream_t stream) {
  MPI_Comm comm = *static_cast<MPI_Comm*>(ci.boot->nativeComm());
  const int P = ci.nranks, me = ci.rank;
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  std::vector<int> sc(P), so(P), rc(P), ro(P);
  i64 send_hi = 0, recv_hi = 0;
  for (int i = 0; i < P; ++i) {
    sc[i] = toInt(p.send_cnt[i], "transpose backend");
    so[i] = toInt(p.send_off[i], "transpose backend");
    rc[i] = toInt(p.recv_cnt[i], "transpose backend");
    ro[i] = toInt(p.recv_off[i], "transpose backend");
    send_hi = std::max(send_hi, p.send_off[i] + p.send_cnt[i]);
    recv_hi = std::max(recv_hi, p.recv_off[i] + p.recv_cnt[i]);
  }
  // my own chunk never touches MPI
  CD_CHECK_HIP(hipMemcpyAsync(b.recv + p.recv_off[me] * es, b.send + p.send_off[me] * es, (size_t)p.send_cnt[me] * es,
                              hipMemcpyDeviceToDevice, stream));
  sc[me] = 0;
  rc[me] = 0;
  if (gpuAware()) {
    CD_CHECK_MPI(MPI_Alltoallv(b.send, sc.data(), so.data(), mpiType(es), b.recv, rc.data(), ro.data(), mpiType(es), comm));
  } else {
    g_stage.grow((size_t)send_hi * es, (size_t)recv_hi * es);
    CD_CHECK_HIP(hipMemcpy(g_stage.send, b.send, (size_t)send_hi * es, hipMemcpyDeviceToHost));
    CD_CHECK_MPI(MPI_Alltoallv(g_stage.send, sc.data(), so.data(), mpiType(es), g_stage.recv, rc.data(), ro.data(),
                               mpiType(es), comm));
    for (int i = 0; i < P; ++i)
      if (rc[i])
        CD_CHECK_HIP(hipMemcpyAsync(b.recv + p.recv_off[i] * es, g_stage.recv + p.recv_off[i] * es, (size_t)rc[i] * es,
                                    hipMemcpyHostToDevice, stream));
    CD_CHECK_HIP(hipStreamSynchronize(stream));
  }
}

void mpiHaloExchange(cudecompHandle_t h, const HaloExchange& x, hipStream_t stream) {
  MPI_Comm comm = *static_cast<MPI_Comm*>(h->boot->nativeComm());
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  const int n = toInt(x.bytes, "halo backend");
  const bool aware = gpuAware();
  char *sbase = x.send, *rbase = x.recv;
  if (!aware) {
    // stage only the two faces / two slots
    g_stage.grow((size_t)2 * x.bytes, (size_t)2 * x.bytes);
    for (int i = 0; i < 2; ++i)
      if (x.neighbor[i] != -1)
        CD_CHECK_HIP(hipMemcpy(g_stage.send + i * x.bytes, x.send + x.send_off[i], (size_t)x.bytes, hipMemcpyDeviceToHost));
  }
  MPI_Request reqs[4] = {MPI_REQUEST_NULL, MPI_REQUEST_NULL, MPI_REQUEST_NULL, MPI_REQUEST_NULL};
  for (int i = 0; i < 2; ++i) {
    if (x.neighbor[i] == -1) continue;
    // tags tell the two faces apart when both neighbours are the same rank: a face sent towards the low
    // side (tag 0) fills the receiver's HIGH slot
    char* rptr = aware ? rbase + x.recv_off[i] : g_stage.recv + i * x.bytes;
    char* sptr = aware ? sbase + x.send_off[i] : g_stage.send + i * x.bytes;
    CD_CHECK_MPI(MPI_Irecv(rptr, n, MPI_BYTE, x.neighbor[i], 1 - i, comm, &reqs[i]));
    CD_CHECK_MPI(MPI_Isend(sptr, n, MPI_BYTE, x.neighbor[i], i, comm, &reqs[2 + i]));
  }
  CD_CHECK_MPI(MPI_Waitall(4, reqs, MPI_STATUSES_IGNORE));
  if (!aware) {
    for (int i = 0; i < 2; ++i)
      if (x.neighbor[i] != -1)
        CD_CHECK_HIP(hipMemcpyAsync(x.recv + x.recv_off[i], g_stage.recv + i * x.bytes, (size_t)x.bytes,
                                    hipMemcpyHostToDevice, stream));
    CD_CHECK_HIP(hipStreamSynchronize(stream));
  }
}

}  // namespace cudecomp
#endif
