// bootstrap.h -- the control plane: the handful of host-side collectives the library needs to set
// itself up (rank discovery, communicator splits, exchange of RCCL ids / IPC handles, autotune
// reductions).  It plays the role MPI plays in the reference (the 21 MPI entry points listed in
// SURVEY.md section 2.3) but never carries array data: device traffic goes over RCCL / xGMI.
//
// Providers:
//   LocalBootstrap  one process, one rank.
//   TcpBootstrap    N processes started by any launcher that exports a rank / world size
//                   (torchrun: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT; MPICH/hydra: PMI_RANK,
//                   PMI_SIZE; Open MPI: OMPI_COMM_WORLD_*; Slurm: SLURM_PROCID, SLURM_NTASKS).
//                   Rank 0 runs a small hub thread; every collective is an all-gather at the hub.
//   DynMpiBootstrap (default build, bootstrap_dynmpi.cc) the communicator the caller passed -- sub-communicators
//                   included -- through the MPICH-ABI MPI already present and initialised in the process
//                   (entry points resolved at run time; no link-time dependency on MPI).
//   MpiBootstrap    (only in the MPI=1 build, bootstrap_mpi.cc) thin wrapper over a real MPI_Comm.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace cudecomp {

class Bootstrap {
 public:
  virtual ~Bootstrap() = default;
  virtual int rank() const = 0;
  virtual int size() const = 0;
  // every member contributes `bytes` bytes; recv gets size()*bytes, ordered by member rank
  virtual void allgather(const void* send, void* recv, size_t bytes) = 0;
  // members with the same color form a new communicator, ordered by (key, parent rank)
  virtual std::unique_ptr<Bootstrap> split(int color, int key) = 0;
  // MPI flavour only: pointer to the underlying MPI_Comm (nullptr for the other providers)
  virtual void* nativeComm() { return nullptr; }

  void barrier();
  void bcast(void* buf, size_t bytes, int root);
  double allreduceMin(double v);
  double allreduceMax(double v);
  double allreduceSum(double v);
  int64_t allreduceMaxI64(int64_t v);
  bool allreduceOr(bool v);
};

struct LaunchEnv {
  int rank = 0;
  int size = 1;
  std::string addr = "127.0.0.1";
  int port = 29617;
};
LaunchEnv detectLaunchEnv();

std::unique_ptr<Bootstrap> makeLocalBootstrap();
// default build: control plane over the MPICH-ABI MPI the calling program brought with it (bootstrap_dynmpi.cc);
// nullptr if there is none
std::unique_ptr<Bootstrap> makeDynMpiBootstrap(int comm);
// instance: n-th bootstrap created by this process (informational; the communicator id counts per hub connection, bootstrap.cc)
std::unique_ptr<Bootstrap> makeTcpBootstrap(const LaunchEnv& env, int instance);

}  // namespace cudecomp
