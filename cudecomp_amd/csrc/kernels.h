// kernels.h -- host entry points of the HIP data-movement kernels (kernels.cc: classification and batching; kernels_rows.hip, kernels_transpose.hip, kernels_window.hip: the kernels).
#pragma once
#include <hip/hip_runtime_api.h>

#include "plan.h"

namespace cudecomp {

// How a normalized move is executed on the GPU.
enum MoveClass {
  MOVE_ROWS_VEC = 0,   // rows contiguous on both sides, everything 16-byte aligned: 16 B/lane streaming copy
  MOVE_TRANSPOSE = 1,  // fastest source dim != fastest destination dim: LDS-tiled transpose (vector or scalar lanes)
  MOVE_GENERIC = 2,    // anything else (odd extents, unaligned bases, degenerate dims): element-wise
  MOVE_CLASS_COUNT = 3
};

struct KernelStats {  // filled per launch when requested (tests, bench bookkeeping)
  int launches[MOVE_CLASS_COUNT] = {0, 0, 0};
  i64 elements[MOVE_CLASS_COUNT] = {0, 0, 0};
};

struct KernelTuning {
  int force_class = -1;          // tests / CUDECOMP_FORCE_GENERIC_KERNELS: force MOVE_GENERIC (2) to cross-check the fast paths
  bool no_streaming = false;     // never use non-temporal access (CUDECOMP_DISABLE_STREAMING_ACCESS=1)
  bool force_streaming = false;  // tests: non-temporal access regardless of the move size
  // tuning switches (read from the environment by `make TUNING_VARIANTS=1` builds only, csrc/api.cc):
  int walk_order = -1;           // transposes walk tiles i first (0) / j first (1); -1 = by strides (CUDECOMP_TILE_WALK)
  int interleave_rows = 1;       // batched row copies: workgroups serve the moves round robin (0: one move after the other)
  int dense_rows = -1;           // moves of whole rows onto halo-carrying pencils: 0 = never rewrite the halo / padding cells between
                                 // consecutive rows (no rows_dense_kernel, no transpose_lines_kernel / transpose_rowlines_kernel: the shifted / window kernels
                                 // instead); CUDECOMP_PRESERVE_OUTPUT_HALOS=1 in every build
  int lines_mode = -1;           // permutations onto halo-carrying pencils whose consecutive batch planes are adjacent rows:
                                 // -1 transpose_lines_kernel when it applies, 0 never (CUDECOMP_LINES_MODE, tuning builds)
  int lines_unit = 128;          // its alignment unit in bytes (64: tuning builds only, CUDECOMP_LINES_UNIT)
  int lines_walk = 0;            // 2 = no gap gather (timing only, results WRONG in the gap cells; CUDECOMP_LINES_WALK, tuning builds)
  int lines_group = 16;          // its tile walk: tile rows per group, 0 = all (CUDECOMP_LINES_GROUP)
  int lines_run_kib = -1;        // ... and KiB of every destination slab written before the next tile row of the group; 0 = one
                                 // window (tile rows first inside the group), -1 = by shape (CUDECOMP_LINES_RUN_KIB)
  int rotate_walk = -1;          // in-place rotation: the orbit walk, -1 = default (kernels_rotate.hip; CUDECOMP_ROTATE_WALK)
  int window_mode = -1;          // transposes onto rows off the 64-byte grid: -1 window kernel for moves >= 1 MiB, 0 never, 1 always
  int window_wide = 0;           // window kernel, 8-byte elements: 1 = 128 x 64 tiles with 512 threads (CUDECOMP_WINDOW_WIDE=1)
  int tile_shape = -1;           // 4-byte transposes with 16-byte lanes: 64 x 128 tiles (2, the default), 64 x 64 (0) or 128 x 64 (1);
                                 // CUDECOMP_TILE_SHAPE; measured on the 8-GiB fp32 cycle (profiles/r04_tuning.md): 11.22 / 11.69 / 11.69 ms
};

// Execute `n` independent moves (disjoint destinations) of `es`-byte elements.  bufs[BufId] are the
// device pointers of the input / output / workspace buffers.  Asynchronous on `stream`.
void launchMoves(const Move3D* moves, int n, void* const bufs[3], int es, hipStream_t stream,
                 const KernelTuning* tuning = nullptr, KernelStats* stats = nullptr,
                 void* const* dst_base_override = nullptr);  // per-move destination base (remote buffers)

// How a move WOULD run (no launch, no device needed): class, kernel variant, tile, tile counts, walk parameters, access mode.
// out[10] = {class, variant, tile_i (row copies: 0 plain / 1 shifted / 2 dense kernel), tile_j, tiles_i, tiles_j, batch, p0 (run length), p1 (walk bits: 1 XCD-contiguous, 2 j first,
// 4 runs over batch planes, 8 transpose_lines_kernel, 16 transpose_rowlines_kernel), access mode}.  (Tests of the planning logic: tests/test_kernel_plan.py.)
void describeMove(const Move3D& m, const void* src, void* dst, int es, const KernelTuning* tuning, long long out[10]);

// name (template spelling) of the data-movement kernel launched last by this process, "" before the first launch
const char* lastKernelName();

// ---- sync.hip: device-side signals of the one-sided exchanges ------------------------------------------------
constexpr int kMaxFlags = 64;  // one lane per flag
struct FlagList {
  int n = 0;
  unsigned long long* f[kMaxFlags];
  void add(unsigned long long* p) { f[n++] = p; }
};
// Flags hold  call number * kFlagScale + step  (monotonic): step 0 = the call has begun, 1 .. kFlagScale-2 = that many
// stages of a staged exchange have landed, kFlagDone = everything of the call has landed.
constexpr unsigned long long kFlagScale = 16;
constexpr int kFlagBegun = 0, kFlagDone = (int)kFlagScale - 1;
// epoch (device memory) += 1; every flag of `begun` = epoch * kFlagScale
void launchEpochBegin(unsigned long long* epoch, const FlagList& begun, hipStream_t stream);
// every flag = *epoch * kFlagScale + step
void launchSignal(const unsigned long long* epoch, const FlagList& flags, hipStream_t stream, int step = kFlagDone);
// returns (on the stream) when every flag >= *epoch * kFlagScale + step; after timeout_s seconds writes a code to *status
// and gives up
void launchWait(const unsigned long long* epoch, const FlagList& flags, unsigned long long* status, double timeout_s,
                hipStream_t stream, int step = kFlagDone);

// stamp / verify the first word of every 4-KiB page of a shared buffer (see sync.hip)
void launchTagPages(void* base, size_t bytes, unsigned long long seed, hipStream_t stream);
void launchCheckPages(const void* base, size_t bytes, unsigned long long seed, unsigned long long* bad, hipStream_t stream);
// debug aid: out2[0] += sum of the 32-bit words of [p, p + bytes), out2[1] += position-weighted sum (mod 2^64)
void launchChecksum(const void* p, size_t bytes, unsigned long long* out2, hipStream_t stream);

}  // namespace cudecomp
