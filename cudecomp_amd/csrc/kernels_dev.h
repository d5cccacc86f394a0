// kernels_dev.h -- device-side helpers shared by the data-movement kernels: lane payload types, access policies, the
// workgroup -> (move, block) decode.  (kernels_batch.h: the launch descriptor, and why the kernels live in several code objects.)
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_batch.h"

namespace cudecomp {
namespace kern {

// N-byte lane payloads as native vector types (kept in VGPRs; a struct-of-array payload gets
// "promoted" to LDS by the compiler, which costs occupancy and LDS bandwidth).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int N> struct BytesOf;
template <> struct BytesOf<4> { using type = unsigned int; };
template <> struct BytesOf<8> { using type = u32x2; };
template <> struct BytesOf<16> { using type = u32x4; };
template <int N> using Bytes = typename BytesOf<N>::type;

// The same payloads for GLOBAL memory at element (not vector) alignment.  gfx950 global loads/stores of 8 and
// 16 bytes only need dword-aligned addresses, so a halo-shifted pencil (interior starting one fp64 past a 16-byte
// boundary, say) still moves 16 bytes per lane; a wavefront then touches one extra cache line per KiB.
typedef u32x2 __attribute__((aligned(4))) u32x2_g;
typedef u32x4 __attribute__((aligned(4))) u32x4_g;
template <int N> struct GlobalBytesOf;
template <> struct GlobalBytesOf<4> { using type = unsigned int; };
template <> struct GlobalBytesOf<8> { using type = u32x2_g; };
template <> struct GlobalBytesOf<16> { using type = u32x4_g; };
template <int N> using GlobalBytes = typename GlobalBytesOf<N>::type;

// element v (ES bytes) of a VW-element vector
template <int ES, int VW> struct Lane;
template <int ES> struct Lane<ES, 1> {
  static __device__ __forceinline__ Bytes<ES> get(const Bytes<ES>& x, int) { return x; }
  static __device__ __forceinline__ void set(Bytes<ES>& x, int, const Bytes<ES>& e) { x = e; }
};
template <> struct Lane<4, 4> {
  static __device__ __forceinline__ unsigned int get(const u32x4& x, int v) { return x[v]; }
  static __device__ __forceinline__ void set(u32x4& x, int v, unsigned int e) { x[v] = e; }
};
template <> struct Lane<4, 2> {
  static __device__ __forceinline__ unsigned int get(const u32x2& x, int v) { return x[v]; }
  static __device__ __forceinline__ void set(u32x2& x, int v, unsigned int e) { x[v] = e; }
};
template <> struct Lane<8, 2> {
  static __device__ __forceinline__ u32x2 get(const u32x4& x, int v) { return v == 0 ? x.xy : x.zw; }
  static __device__ __forceinline__ void set(u32x4& x, int v, const u32x2& e) {
    if (v == 0) x.xy = e; else x.zw = e;
  }
};

// Streaming (non-temporal) access for moves far larger than the caches: measured +3..15 % on the 1024^3
// permutations (profiles/r01_tuning.md); small moves keep the default policy so a following kernel can
// still find the data in L2 / Infinity Cache.
template <bool STREAM, int N>
__device__ __forceinline__ Bytes<N> loadVec(const void* p) {
  const GlobalBytes<N>* q = static_cast<const GlobalBytes<N>*>(p);
  if constexpr (STREAM) return __builtin_nontemporal_load(q);
  else return *q;
}
// Store policies: ST_CACHED default, ST_STREAM non-temporal, ST_REMOTE system-scope write-through (sc0 sc1) for
// destinations in ANOTHER GPU's memory (one-sided puts over xGMI).  A plain store to peer memory may linger as a
// dirty line in this XCD's L2 until some later system-scope release; the stream-ordered exchanges signal the
// receiver from the NEXT kernel on the stream, whose release would only write back the L2 of the one XCD it runs
// on.  Write-through stores need no flush: once the wave's stores are acknowledged (s_waitcnt vmcnt(0) at the end of
// the kernel, remoteStoresDone()) they are in the peer's memory.  (A `volatile` store gives the same cache bits but
// makes the compiler wait for every single store, which serialises a lane's 4-8 stores.)
enum StorePolicy { ST_CACHED = 0, ST_STREAM = 1, ST_REMOTE = 2 };
template <int N> __device__ __forceinline__ void storeRemote(void* p, const Bytes<N>& v);
template <> __device__ __forceinline__ void storeRemote<4>(void* p, const Bytes<4>& v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <> __device__ __forceinline__ void storeRemote<8>(void* p, const Bytes<8>& v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <> __device__ __forceinline__ void storeRemote<16>(void* p, const Bytes<16>& v) {
  // gfx940+ hazard: a VALU write to the data VGPRs of a store wider than 64 bits needs 2 wait states after the
  // store.  The compiler inserts them for its own stores but cannot see inside inline assembly (without the s_nop a
  // few cells per GiB arrived holding the next tile's address arithmetic instead of data).
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void remoteStoresDone() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int POLICY, int N>
__device__ __forceinline__ void storeVec(void* p, const Bytes<N>& v) {
  GlobalBytes<N>* q = static_cast<GlobalBytes<N>*>(p);
  if constexpr (POLICY == ST_REMOTE) storeRemote<N>(p, v);
  else if constexpr (POLICY == ST_STREAM) __builtin_nontemporal_store(v, q);
  else *q = v;
}
// STREAM template parameter of the kernels: 0 = default caching, 1 = non-temporal loads, 2 = non-temporal loads
// and stores, 3 = non-temporal loads + system-scope write-through stores (remote destinations), 4 = cached loads +
// non-temporal stores (misaligned sources: neighbouring tiles share the partially used lines through L2)
template <int STREAM> constexpr int storePolicyOf() {
  return STREAM == 3 ? ST_REMOTE : ((STREAM == 2 || STREAM == 4) ? ST_STREAM : ST_CACHED);
}
template <int STREAM> constexpr bool loadsStream() { return STREAM >= 1 && STREAM <= 3; }

__device__ __forceinline__ int findMove(const Batch& b, unsigned int block) {
  int mi = 0;
#pragma unroll
  for (int i = 1; i < kMaxBatch; ++i)
    if (i < b.n && block >= b.first_block[i]) mi = i;
  return mi;
}

// workgroup -> (move, workgroup index inside the move); false for the filler workgroups of an interleaved launch
__device__ __forceinline__ bool locate(const Batch& b, unsigned int block, int& mi, unsigned int& lb) {
  if (b.interleave) {
    mi = (int)(block % (unsigned int)b.n);
    lb = block / (unsigned int)b.n;
    return lb < b.first_block[mi + 1] - b.first_block[mi];
  }
  mi = findMove(b, block);
  lb = block - b.first_block[mi];
  return true;
}

}  // namespace kern
}  // namespace cudecomp
