// sync.hip -- device-side ordering of the one-sided (xGMI peer) exchanges.
//
// The reference orders its NVSHMEM exchanges on the stream with signal / wait operations issued from small kernels
// (include/internal/cudecomp_kernels.cuh:51-122, include/internal/comm_routines.h:122-258).  Here the signals are
// 64-bit epoch counters in a host-pinned shared-memory board that every rank of the node maps into its GPU
// (transport.cc: PeerContext): system-scope atomics on pinned host memory are coherent across GPUs and processes
// by construction, whatever the caching policy of the data buffers.  Three 1-wave kernels:
//
//   (flag value = call number * kFlagScale + step, see below)
//   epoch_begin_k   epoch += 1 (the call's number, kept in DEVICE memory so that a captured graph replays with fresh
//                   epochs) and publish it in my `ready` flag: "everything before this call on my stream is done;
//                   my receive area / output pencil may be written".
//   signal_k        store the call's epoch into up to kMaxFlags flags (release, system scope).
//   wait_k          lane i spins until flag i >= epoch (acquire, system scope), sleeping between polls; gives up after
//                   `timeout_ticks` of the 100 MHz wall clock and reports through `status` (host-visible).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "errors.h"
#include "kernels.h"

namespace cudecomp {

namespace {

using u64 = unsigned long long;

// A flag holds  call number * kFlagScale + step:  step 0 = "the call has begun" (ready flags), steps 1 .. kFlagScale-2 =
// stages of a staged exchange that have completely landed, kFlagScale-1 = "everything of this call has landed".
__global__ void epoch_begin_k(u64* epoch, const FlagList begun) {
  // every lane reads the old value before lane 0 bumps it (one wave: the read precedes the write in program order)
  // (system-scope atomics on an uncached cell: no compute die's cache ever holds the counter, see transport.cc devEpoch)
  const u64 e = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) __hip_atomic_store(epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int i = threadIdx.x;
  if (i < begun.n) __hip_atomic_store(begun.f[i], e * kFlagScale, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void signal_k(const u64* epoch, const FlagList flags, u64 step) {
  const int i = threadIdx.x;
  if (i < flags.n)
    __hip_atomic_store(flags.f[i], __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) * kFlagScale + step, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void wait_k(const u64* epoch, const FlagList flags, u64* status, long long timeout_ticks, u64 step) {
  const int i = threadIdx.x;
  if (i < flags.n) {
    const u64 call = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const u64 e = call * kFlagScale + step;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flags.f[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > timeout_ticks) {
        // leave a trace for the host (checked at the next library call) and let the stream drain
        __hip_atomic_store(status, (call << 8) | (u64)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
}

// position-weighted checksum of a byte range taken as 32-bit words: out[0] += sum(w_i), out[1] += sum(w_i * (i + 1))
// (mod 2^64) -- a moved, missing or stale block changes it.  Debug aid of the one-sided exchanges
// (CUDECOMP_DEBUG_VERIFY_EXCHANGE), not part of any data path.
__global__ void checksum_k(const unsigned int* p, unsigned long long words, u64* out) {
  u64 s1 = 0, s2 = 0;
  for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < words; i += (u64)gridDim.x * blockDim.x) {
    const u64 w = __builtin_nontemporal_load(p + i);
    s1 += w;
    s2 += w * (i + 1);
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off);
    s2 += __shfl_down(s2, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, s1);
    atomicAdd(out + 1, s2);
  }
}

// Page tags of a freshly shared buffer (transport.cc: registerRegion): the owner stamps the first word of every 4-KiB
// page with a value derived from a per-registration seed, every importer reads the stamps back THROUGH ITS NEW MAPPING.
// A mapping that does not lead to the owner's pages (stale IPC mappings of re-created allocations) shows
// as mismatching pages.
__device__ __forceinline__ u64 pageTag(u64 seed, u64 page) {
  u64 x = seed + page * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 32;
  return x;
}
__global__ void tag_pages_k(u64* base, u64 pages, u64 seed) {
  for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < pages; i += (u64)gridDim.x * blockDim.x)
    __hip_atomic_store(base + i * 512, pageTag(seed, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void check_pages_k(const u64* base, u64 pages, u64 seed, u64* bad) {
  u64 mine = 0;
  for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < pages; i += (u64)gridDim.x * blockDim.x)
    if (__hip_atomic_load(base + i * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != pageTag(seed, i)) ++mine;
  if (mine) atomicAdd(bad, mine);
}

}  // namespace

void launchTagPages(void* base, size_t bytes, unsigned long long seed, hipStream_t stream) {
  const unsigned long long pages = bytes / 4096;
  if (pages == 0) return;
  tag_pages_k<<<(unsigned int)std::min<unsigned long long>((pages + 255) / 256, 1024), 256, 0, stream>>>(static_cast<u64*>(base), pages, seed);
  CD_CHECK_HIP(hipGetLastError());
}

void launchCheckPages(const void* base, size_t bytes, unsigned long long seed, unsigned long long* bad, hipStream_t stream) {
  const unsigned long long pages = bytes / 4096;
  if (pages == 0) return;
  check_pages_k<<<(unsigned int)std::min<unsigned long long>((pages + 255) / 256, 1024), 256, 0, stream>>>(static_cast<const u64*>(base), pages, seed, bad);
  CD_CHECK_HIP(hipGetLastError());
}

void launchChecksum(const void* p, size_t bytes, unsigned long long* out2, hipStream_t stream) {
  if (bytes < 4) return;
  const unsigned long long words = bytes / 4;
  const unsigned int blocks = (unsigned int)std::min<unsigned long long>((words + 1023) / 1024, 2048);
  checksum_k<<<blocks, 256, 0, stream>>>(static_cast<const unsigned int*>(p), words, out2);
  CD_CHECK_HIP(hipGetLastError());
}

void launchEpochBegin(unsigned long long* epoch, const FlagList& begun, hipStream_t stream) {
  epoch_begin_k<<<1, 64, 0, stream>>>(epoch, begun);
  CD_CHECK_HIP(hipGetLastError());
}

void launchSignal(const unsigned long long* epoch, const FlagList& flags, hipStream_t stream, int step) {
  if (flags.n == 0) return;
  signal_k<<<1, 64, 0, stream>>>(epoch, flags, (u64)step);
  CD_CHECK_HIP(hipGetLastError());
}

void launchWait(const unsigned long long* epoch, const FlagList& flags, unsigned long long* status, double timeout_s,
                hipStream_t stream, int step) {
  if (flags.n == 0) return;
  wait_k<<<1, 64, 0, stream>>>(epoch, flags, status, (long long)(timeout_s * 1e8), (u64)step);
  CD_CHECK_HIP(hipGetLastError());
}

}  // namespace cudecomp
