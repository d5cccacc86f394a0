// ipc_remap_probe.cpp -- does a one-sided write through a FRESH IPC mapping always land, when N processes time-share
// one GPU?  No library code, no flag protocol: host barriers and full device synchronisation only, so anything that
// goes wrong here is the platform's (IPC re-mapping / translation / cache behaviour), not the exchange protocol's.
//
//   ipc_remap_probe NPROCS ITERS MODE [KiB] [EXTRA_STREAMS]
//     MODE bit 0: re-create the buffers and re-open every peer mapping in EVERY iteration (else: once)
//          bit 1: write with hipMemcpyAsync (else: a copy kernel)
//          bit 2: write-through (sc0 sc1) stores in the copy kernel (else: plain stores)
//          bit 3: keep a spinning 1-wave kernel on EXTRA_STREAMS streams during the writes (queue pressure, as the
//                 library's wait kernels produce it)
//          bit 4: the parent process holds a HIP context of its own (a ninth process on the device)
//          bit 5: never hipFree a buffer before the end (new allocations cannot reuse an address or a handle)
//          bit 6: keep the buffers, only CLOSE and RE-OPEN the peers' mappings in every iteration
//          bit 7: print the IPC handle bytes of rank 0's buffer in every iteration
//          bit 8: importers never CLOSE a mapping (the owners still free and re-create their buffers)
//          bit 9: the owner sleeps 20 ms between hipFree and the next hipMalloc (is it a race with a deferred release?)
//
// Children are forked before HIP starts.  Every iteration: each rank owns `buf` (N slices); rank r writes slice r of
// EVERY rank's buf with f(r, iteration, index); after a device sync + barrier every rank checks its whole buf on the
// device.  A mismatch is reported with the byte range, its alignment and what the cells hold instead (the fill
// pattern of this iteration = "never arrived here"; the previous iteration's data = "stale").
#include <hip/hip_runtime.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static int g_me = -1;
struct Shared;
static Shared* g_sh = nullptr;
#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("[%d] %s:%d %s -> %s\n", g_me, __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      fflush(stdout);                                                                          \
      if (g_sh) g_sh->abort_all.store(1);                                                      \
      _exit(2);                                                                                \
    }                                                                                          \
  } while (0)

constexpr int kMaxProcs = 16;
struct Shared {
  std::atomic<uint64_t> arrive[kMaxProcs];
  hipIpcMemHandle_t handle[kMaxProcs];
  std::atomic<uint64_t> bad_total;
  std::atomic<uint64_t> stop_spin;
  std::atomic<uint64_t> abort_all;
};

using u64 = unsigned long long;
__host__ __device__ inline u64 pattern(u64 writer, u64 iter, u64 idx) {
  u64 x = (writer + 1) * 0x9E3779B97F4A7C15ull + iter * 0xBF58476D1CE4E5B9ull + idx * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x | 1ull;  // never equal to a fill value (those are even)
}
__host__ __device__ inline u64 fillValue(u64 owner, u64 iter) { return ((owner + 1) << 40 | iter << 8) & ~1ull; }

__global__ void fill_k(u64* p, size_t n, u64 v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void gen_k(u64* src, size_t n, u64 writer, u64 iter) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    src[i] = pattern(writer, iter, i);
}
template <bool WT>
__global__ void copy_k(u64* dst, const u64* src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (WT) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else dst[i] = src[i];
  }
}
struct Report {
  u64 bad, first, last, is_fill, is_prev;
};
__global__ void verify_k(const u64* buf, size_t slice, int nprocs, u64 iter, u64 owner, Report* rep) {
  const size_t n = slice * nprocs;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const u64 w = i / slice, j = i % slice;
    const u64 got = buf[i];
    if (got != pattern(w, iter, j)) {
      Report* r = rep + w;
      atomicAdd(&r->bad, 1ull);
      atomicMin(&r->first, (u64)i);
      atomicMax(&r->last, (u64)i);
      if (got == fillValue(owner, iter)) atomicAdd(&r->is_fill, 1ull);
      if (iter > 0 && got == pattern(w, iter - 1, j)) atomicAdd(&r->is_prev, 1ull);
    }
  }
}
__global__ void spin_k(const u64* stop) {
  if (threadIdx.x == 0)
    while (__hip_atomic_load(stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0) __builtin_amdgcn_s_sleep(16);
}

static void barrier(Shared* sh, int n, uint64_t& epoch) {
  ++epoch;
  sh->arrive[g_me].store(epoch, std::memory_order_release);
  for (int p = 0; p < n; ++p)
    while (sh->arrive[p].load(std::memory_order_acquire) < epoch) {
      if (sh->abort_all.load()) _exit(3);  // a sibling failed: do not wait for it forever
      std::this_thread::yield();
    }
}

static int child(Shared* sh, int me, int n, int iters, int mode, size_t kib, int extra_streams) {
  g_me = me;
  g_sh = sh;
  prctl(PR_SET_PDEATHSIG, SIGKILL);  // never outlive the launcher (a timeout kills only the parent)
  const bool leak = mode & 32, reopen_only = mode & 64, show = mode & 128, no_close = mode & 256, nap = mode & 512;
  std::vector<u64*> parked;
  const bool remap = mode & 1, use_memcpy = mode & 2, wt = mode & 4, spin = mode & 8;
  CK(hipSetDevice(0));
  uint64_t epoch = 0;
  size_t slice = kib * 1024 / 8 / n;  // u64 per slice
  std::vector<hipStream_t> streams(n);
  for (auto& s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<hipStream_t> extra(spin ? extra_streams : 0);
  for (auto& s : extra) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  u64* dstop = nullptr;
  if (spin) {
    CK(hipHostRegister(&sh->stop_spin, sizeof(u64), hipHostRegisterMapped));
    CK(hipHostGetDevicePointer((void**)&dstop, &sh->stop_spin, 0));
  }
  u64 *buf = nullptr, *src = nullptr;
  Report* rep = nullptr;
  CK(hipMalloc(&rep, sizeof(Report) * n));
  std::vector<u64*> remote(n, nullptr);
  u64 my_bad = 0;
  auto open_all = [&](u64 iter) {
    // sizes wander a little so that the allocator does not hand back exactly the same block every time
    const size_t bytes = slice * n * 8 + ((iter % 3) << 16);
    if (!(reopen_only && buf)) CK(hipMalloc(&buf, bytes));
    fill_k<<<256, 256>>>(buf, slice * n, fillValue(me, iter));
    CK(hipDeviceSynchronize());
    CK(hipIpcGetMemHandle(&sh->handle[me], buf));
    if (show && me == 0) {
      const unsigned char* hb = reinterpret_cast<const unsigned char*>(&sh->handle[me]);
      printf("[0] iter %llu buf %p handle", (u64)iter, (void*)buf);
      for (int k = 0; k < 64; ++k) printf("%s%02x", k % 8 ? "" : " ", hb[k]);
      printf("\n");
      fflush(stdout);
    }
    barrier(sh, n, epoch);
    for (int p = 0; p < n; ++p) {
      if (p == me) {
        remote[p] = buf;
        continue;
      }
      void* m = nullptr;
      CK(hipIpcOpenMemHandle(&m, sh->handle[p], hipIpcMemLazyEnablePeerAccess));
      remote[p] = (u64*)m;
    }
    barrier(sh, n, epoch);
  };
  auto close_all = [&]() {
    CK(hipDeviceSynchronize());
    barrier(sh, n, epoch);
    for (int p = 0; p < n; ++p)
      if (p != me && !no_close) CK(hipIpcCloseMemHandle(remote[p]));
    barrier(sh, n, epoch);
    if (reopen_only) return;
    if (leak) parked.push_back(buf);
    else CK(hipFree(buf));
    buf = nullptr;
    if (nap) std::this_thread::sleep_for(std::chrono::milliseconds(20));
  };
  CK(hipMalloc(&src, slice * 8));
  if (!remap) open_all(0);
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 0; it < iters; ++it) {
    if (remap) open_all(it);
    else {
      fill_k<<<256, 256>>>(buf, slice * n, fillValue(me, it));
      CK(hipDeviceSynchronize());
      barrier(sh, n, epoch);
    }
    gen_k<<<256, 256>>>(src, slice, me, it);
    CK(hipDeviceSynchronize());
    if (spin) {
      if (me == 0) sh->stop_spin.store(0);
      barrier(sh, n, epoch);
      for (auto& s : extra) spin_k<<<1, 64, 0, s>>>(dstop);
    }
    for (int p = 0; p < n; ++p) {
      u64* dst = remote[p] + (size_t)me * slice;
      if (use_memcpy) CK(hipMemcpyAsync(dst, src, slice * 8, hipMemcpyDefault, streams[p]));
      else if (wt) copy_k<true><<<64, 256, 0, streams[p]>>>(dst, src, slice);
      else copy_k<false><<<64, 256, 0, streams[p]>>>(dst, src, slice);
    }
    for (int p = 0; p < n; ++p) CK(hipStreamSynchronize(streams[p]));
    if (spin) {
      barrier(sh, n, epoch);
      sh->stop_spin.store(1);
    }
    CK(hipDeviceSynchronize());
    barrier(sh, n, epoch);
    // check my whole buffer
    std::vector<Report> zero(n);
    for (auto& r : zero) r = Report{0, ~0ull, 0, 0, 0};
    CK(hipMemcpy(rep, zero.data(), sizeof(Report) * n, hipMemcpyHostToDevice));
    verify_k<<<256, 256>>>(buf, slice, n, it, me, rep);
    CK(hipDeviceSynchronize());
    std::vector<Report> got(n);
    CK(hipMemcpy(got.data(), rep, sizeof(Report) * n, hipMemcpyDeviceToHost));
    for (int w = 0; w < n; ++w) {
      if (!got[w].bad) continue;
      my_bad += got[w].bad;
      static int reports = 0;
      if (++reports > 6) continue;
      const u64 b0 = got[w].first * 8, b1 = (got[w].last + 1) * 8;
      printf("[%d] iter %d: slice written by %d: %llu bad u64 in bytes [%llu, %llu) of my buffer (span %llu KiB, start %% 64K = %llu, "
             "%% 4K = %llu); %llu hold this iteration's fill (never arrived), %llu hold the previous iteration's data (stale)\n",
             me, it, w, got[w].bad, b0, b1, (b1 - b0) / 1024, b0 % 65536, b0 % 4096, got[w].is_fill, got[w].is_prev);
      fflush(stdout);
      // late or lost?  look again after everybody idled for a while
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
      CK(hipMemcpy(rep, zero.data(), sizeof(Report) * n, hipMemcpyHostToDevice));
      verify_k<<<256, 256>>>(buf, slice, n, it, me, rep);
      CK(hipDeviceSynchronize());
      std::vector<Report> again(n);
      CK(hipMemcpy(again.data(), rep, sizeof(Report) * n, hipMemcpyDeviceToHost));
      printf("[%d] iter %d: slice %d re-checked 5 ms later: %llu bad\n", me, it, w, again[w].bad);
      fflush(stdout);
    }
    barrier(sh, n, epoch);
    if (remap) close_all();
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!remap) close_all();
  for (u64* q : parked) CK(hipFree(q));
  sh->bad_total.fetch_add(my_bad);
  barrier(sh, n, epoch);
  if (me == 0) {
    printf("RESULT nprocs %d iters %d mode %d (remap %d memcpy %d write-through %d spin %d) KiB %zu: %llu bad u64 in total, %.1f s\n", n,
           iters, mode, (int)remap, (int)use_memcpy, (int)wt, (int)spin, kib, (u64)sh->bad_total.load(), secs);
    fflush(stdout);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    printf("usage: %s NPROCS ITERS MODE [KiB] [EXTRA_STREAMS]\n", argv[0]);
    return 1;
  }
  const int n = atoi(argv[1]), iters = atoi(argv[2]), mode = atoi(argv[3]);
  const size_t kib = argc > 4 ? (size_t)atoll(argv[4]) : 4096;
  const int extra = argc > 5 ? atoi(argv[5]) : 3;
  if (n < 1 || n > kMaxProcs) return 1;
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  new (sh) Shared();
  std::vector<pid_t> kids;
  for (int r = 0; r < n; ++r) {
    pid_t p = fork();
    if (p == 0) _exit(child(sh, r, n, iters, mode, kib, extra));
    kids.push_back(p);
  }
  if (mode & 16) {  // a process that only HOLDS a context (like a test driver that imported a GPU framework)
    void* p = nullptr;
    (void)hipMalloc(&p, 1 << 20);
  }
  int rc = 0;
  for (pid_t p : kids) {
    int st = 0;
    waitpid(p, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
  }
  return rc;
}
