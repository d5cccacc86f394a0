// kernels_batch.h -- what csrc/kernels.cc (host: classification, batching) and the kernel code objects (kernels_rows.hip,
// kernels_transpose.hip, kernels_window.hip) share: the launch descriptor and one launcher per code object.
//
// Why several translation units: every .hip file becomes ONE code object inside the library's .hip_fatbin, and a single
// code object beyond roughly 0.6-0.7 MB puts the whole process into a regime where every small synchronous operation costs
// 14 ms (bisected in round 5, profiles/r05_code_size.md: the same 0.74 MB of device code split over two code objects is
// harmless, in one code object it is not).  The kernels therefore live in five small code objects -- row copies + generic,
// the LDS-tiled transposes per element size (kernels_transpose.hip compiled three times), the window transposes -- and
// tests/test_abi.py guards the size of each.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace cudecomp {
namespace kern {

constexpr int kMaxBatch = 8;
constexpr int kThreads = 256;
constexpr int kRowsUnroll = 4;
constexpr long long kStreamBytes = 32ll << 20;  // moves at least this large use non-temporal access

struct DevMove {
  const char* src;
  char* dst;
  long long e[3];   // extents   (units depend on the kernel, see the launchers)
  long long ss[3];  // src strides
  long long ds[3];  // dst strides
};

struct Batch {
  int n;
  int interleave;  // 1: workgroup b serves move b % n (moves with REMOTE destinations: keeps every xGMI link busy
                   // for the whole launch instead of draining one peer's chunk after the other)
  int p0[kMaxBatch];                      // kernel-specific small parameter
  int p1[kMaxBatch];                      // second small parameter (transpose: XCD-contiguous tile walk)
  unsigned int first_block[kMaxBatch + 1];
  unsigned int t0[kMaxBatch];             // tiles along dim 0
  unsigned int t1[kMaxBatch];             // tiles along dim 1
  DevMove m[kMaxBatch];
};

}  // namespace kern

// ---- launchers, one per code object (host side; csrc/kernels.cc decides what runs) -------------------------------------------
// stream_access: 0 default caching, 1 non-temporal loads, 2 non-temporal loads + stores, 3 non-temporal loads + remote
// (system-scope write-through) stores, 4 cached loads + non-temporal stores (see storePolicyOf)
// rows: mode 0 plain, 1 shifted (lanes on the destination's 64-byte grid), 2 dense (whole lines across row ends, Move3D::dst_row_pitch)
void launchRowsBatch(int mode, int vector_bytes, int stream_access, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
int rowsDenseBytesPerBlock();  // bytes of a plane's span one workgroup of the dense row copy covers
void launchGenericBatch(int es, bool remote, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
// transposes: `variant` = elements per 16-byte lane group (1 = element-wise lanes), plus 300 for the 64 x 128 tile of 4-byte
// elements (tuning builds: 200 = 128 x 64, 0 = 64 x 64)
void launchTransposeBatch4(int variant, int stream_access, bool swizzle, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
void launchTransposeBatch8(int variant, int stream_access, bool swizzle, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
void launchTransposeBatch16(int variant, int stream_access, bool swizzle, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
void launchWindowBatch(int es, int variant, bool wide, int stream_access, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
// kernels_lines.hip: windows over the destination's linear positions across row ends (unit_bytes: 128; 64 in tuning builds)
void launchLinesBatch(int es, int variant, int stream_access, int unit_bytes, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
int linesUnitBytes(int unit_choice);  // the unit the build really has for a wish
// kernels_rowlines.hip: the same idea for destinations whose adjacent rows are the tile's own rows (128-byte units)
void launchRowLinesBatch(int es, int variant, int stream_access, const kern::Batch& b, unsigned int blocks, hipStream_t stream);
// kernels_rotate.hip: in-place rotation of a cubic n^3 array (direction +1: new[p0,p1,p2] = old[p2,p0,p1]; -1: the inverse)
bool rotateSupported(int es, long long n);
void launchRotate(void* buffer, long long n, int es, int direction, hipStream_t stream, int walk = -1);

}  // namespace cudecomp
