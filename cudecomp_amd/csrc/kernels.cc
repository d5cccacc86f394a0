// kernels.cc -- host side of the data-movement kernels: how a Move3D (plan.h) is executed on the GPU.
//
// A move is normalised (unit dims dropped, contiguous dims fused, dims sorted by source stride), classified -- row copy,
// LDS-tiled transposition (plain, "window" for destinations off the 64-byte grid, "lines" / "row lines" when whole rows of a
// halo-carrying pencil are written), generic element-wise -- and batched with its
// siblings (up to kMaxBatch moves, e.g. the per-peer pack copies of one transpose, share one launch; the descriptors travel in
// the kernel argument segment).  The kernels themselves live in kernels_rows.hip, kernels_transpose.hip (one code object per
// element size), kernels_window.hip, kernels_lines.hip, kernels_rowlines.hip and kernels_rotate.hip (the in-place rotation:
// launched by the executor, transpose.cc); kernels_batch.h says why they are separate code objects.
//
// Pure data movement: no MFMA; the bound is HBM (8 TB/s spec, ~6.3 TB/s achievable copy rate).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "errors.h"
#include "kernels.h"
#include "kernels_batch.h"

namespace cudecomp {

using namespace kern;

namespace {

// ---------------------------------------------------------------------------------------------
// classification
// ---------------------------------------------------------------------------------------------
struct Classified {
  MoveClass cls;
  int variant;  // rows: vector bytes; transpose: elements per vector
  DevMove dm;
  int p0, p1;
  int stream;  // 0 default caching, 1 streaming loads, 2 streaming loads + stores, 3 streaming loads + remote stores
  bool swizzle = false;  // transposes: XOR-swizzled LDS tile (else padded rows)
  bool window = false;   // transposes: destination rows off the 64-byte grid -> transpose_window_kernel (rows: rows_shifted_kernel)
  bool dense = false;    // rows, with window: whole lines across the row ends (rows_dense_kernel)
  bool lines = false;    // transposes, with window: windows over the linear positions of adjacent rows (transpose_lines_kernel)
  bool rowlines = false; // transposes, with window: the tile's own rows are the adjacent ones (transpose_rowlines_kernel)
  int unit = 0;          // ... and its alignment unit in bytes
  unsigned int t0, t1;
  unsigned long long blocks;
  i64 elements;
};

constexpr long long kDenseMaxGapBytes = 512;  // widest gap between rows the dense row copy rewrites (halo + padding cells)

int ilog2ceil(long long x) {
  int l = 0;
  while ((1LL << l) < x) ++l;
  return l;
}

Classified classify(const Move3D& in, void* const bufs[3], int es, const KernelTuning* tuning, void* dst_base,
                    bool remote) {
  Move3D m = in;
  normalizeMove(m);
  Classified c{};
  c.elements = m.elements();
  c.stream = ((c.elements * es >= kStreamBytes || (tuning && tuning->force_streaming)) && !(tuning && tuning->no_streaming)) ? 2 : 0;
  if (remote) c.stream = 3;  // destination in a peer's memory: write-through stores, whatever the size
  c.dm.src = static_cast<const char*>(bufs[m.src_buf]) + m.src_off * es;
  c.dm.dst = static_cast<char*>(dst_base ? dst_base : bufs[m.dst_buf]) + m.dst_off * es;
  const bool force_generic = tuning && tuning->force_class == MOVE_GENERIC;

  if (!force_generic && m.ss[0] <= 1 && m.ds[0] <= 1) {
    // rows contiguous on both sides (also the all-extents-1 case).  Widest vector that divides the row length;
    // addresses only need the element's natural alignment (see GlobalBytes).
    int vb = 16;
    while (vb > es && (m.extent[0] * es) % vb != 0) vb >>= 1;
    c.cls = MOVE_ROWS_VEC;
    c.variant = vb;
    c.dm.e[0] = m.extent[0] * es / vb;
    c.dm.e[1] = m.extent[1];
    c.dm.e[2] = m.extent[2];
    for (int i = 1; i < 3; ++i) {
      c.dm.ss[i] = m.ss[i] * es;
      c.dm.ds[i] = m.ds[i] * es;
    }
    // rows that land off the 64-byte grid (and are long enough for it to matter): lanes laid out from the unit boundary
    // below each row's start (rows_shifted_kernel), one unit of slack vectors per row
    const uintptr_t dst_bits = reinterpret_cast<uintptr_t>(c.dm.dst) | (uintptr_t)c.dm.ds[1] | (uintptr_t)c.dm.ds[2];
    const int shift_mode = tuning ? tuning->window_mode : -1;
    if ((dst_bits & 63) != 0 && m.extent[0] * es >= 256 && shift_mode != 0 && (shift_mode == 1 || c.elements * es >= (1ll << 20))) {
      c.window = true;
      c.p1 = (int)(m.extent[0] * es);  // row length in bytes (rows longer than 2 GiB keep the plain kernel)
      if (m.extent[0] * es > 0x7fffffffLL) c.window = false;
    }
    // ... and when the move covers whole interior rows of a halo-carrying pencil (the planner says so and names the pencil's
    // row pitch: dst_row_pitch) and the gap between consecutive rows is a few halo / padding cells: the dense walk of
    // rows_dense_kernel, which writes whole lines across the row ends.  Local destinations only; dim 1 must be the one that
    // steps by the pencil's row pitch (a move one row tall per plane has no such dim: its rows are whole planes apart, with
    // other moves' rows in between).  A row that normalizeMove has fused with the next dim -- no gap -- keeps the shifted kernel.
    i64 planned_row = -1;
    for (int i = 0; i < 3; ++i)
      if (in.ss[i] == 1 && in.ds[i] == 1 && in.extent[i] > 1) planned_row = in.extent[i];
    if (c.window && in.dst_row_pitch > 0 && !remote && planned_row == m.extent[0] && c.dm.e[1] > 1 &&
        (!tuning || tuning->dense_rows != 0)) {
      DevMove d = c.dm;
      if (d.e[2] > 1 && d.ds[2] < d.ds[1]) {
        std::swap(d.e[1], d.e[2]);
        std::swap(d.ss[1], d.ss[2]);
        std::swap(d.ds[1], d.ds[2]);
      }
      const long long row_bytes = m.extent[0] * es, gap = d.ds[1] - row_bytes;
      const long long span = (d.e[1] - 1) * d.ds[1] + row_bytes;
      const bool planes_apart = d.e[2] == 1 || span <= d.ds[2];
      if (d.e[1] > 1 && d.ds[1] == in.dst_row_pitch * es && gap > 0 && gap <= kDenseMaxGapBytes && gap * 8 <= row_bytes && planes_apart) {
        c.dense = true;
        c.variant = 16;
        c.dm = d;
        c.dm.e[0] = row_bytes;
        c.p0 = 0;
        const long long per = rowsDenseBytesPerBlock();
        c.t0 = (unsigned int)((span + 63 + per - 1) / per);  // (+63: the lanes start at the 64-byte boundary below the first row)
        c.t1 = 1;
        c.blocks = (unsigned long long)c.t0 * (unsigned long long)c.dm.e[2];
        return c;
      }
    }
    if (c.window) c.dm.e[0] += 64 / vb;  // one unit of slack vectors per row (see rows_shifted_kernel)
    c.p0 = std::min(8, ilog2ceil(c.dm.e[0]));
    const long long lpr = 1LL << c.p0, rows_per_block = (long long)(kThreads >> c.p0) * kRowsUnroll;
    c.t0 = (unsigned int)((c.dm.e[0] + lpr - 1) / lpr);
    c.t1 = (unsigned int)((c.dm.e[1] + rows_per_block - 1) / rows_per_block);
    c.blocks = (unsigned long long)c.t0 * c.t1 * (unsigned long long)c.dm.e[2];
    return c;
  }

  int t = -1;
  if (!force_generic && m.ss[0] == 1) {
    if (m.ds[1] == 1) t = 1;
    if (m.ds[2] == 1) t = 2;
  }
  if (t > 0 && m.extent[0] >= 4 && m.extent[t] >= 4) {
    const int k = 3 - t;
    c.cls = MOVE_TRANSPOSE;
    c.dm.e[0] = m.extent[0];
    c.dm.e[1] = m.extent[t];
    c.dm.e[2] = m.extent[k];
    c.dm.ss[0] = 1;
    c.dm.ss[1] = m.ss[t];
    c.dm.ss[2] = m.ss[k];
    c.dm.ds[0] = m.ds[0];
    c.dm.ds[1] = 1;
    c.dm.ds[2] = m.ds[k];
    // 16 bytes per lane whenever both tile edges hold whole vectors (no alignment requirement, see GlobalBytes)
    int vw = 16 / es;
    if (c.dm.e[0] % vw != 0 || c.dm.e[1] % vw != 0) vw = 1;
    c.variant = vw;
    c.p1 = 1;  // XCD-contiguous tile walk
    c.swizzle = es != 16;
    // Tile walk order inside an XCD's run: j first makes consecutive tiles extend the same DESTINATION rows
    // (contiguous write stream per row), i first the same source rows.  Measured on 8 GiB permutations
    // (profiles/r01_tuning.md): j first wins or ties for line-aligned moves (8-11 % at 16-byte elements and on
    // the strided-read side at 4-byte elements; 4-byte moves whose destination rows are the far-strided side
    // lose 1-3 % and keep i first), i first wins by 5-10 % for misaligned moves, where L2 merges the
    // partially read lines of neighbouring tiles.
    bool j_first = es != 4 || c.dm.ss[1] > c.dm.ds[0];
    // Rows that do not start on cache-line boundaries (halo-shifted or odd-extent pencils) leave partially covered
    // lines at both ends of every tile row.
    //  * Misaligned SOURCE rows only: the partially used lines are shared with the neighbouring tile; cached loads let
    //    L2 serve the second use (non-temporal loads fetch them twice), the aligned stores keep streaming.
    //  * Misaligned DESTINATION rows: partial 64-byte units written by two tiles are what costs (a cached store lets L2
    //    merge some: fp32 3.0 -> 4.4 TB/s, fp64 3.9 -> 4.8 TB/s on a halo-shifted 8 GiB permutation); the window kernel
    //    writes whole units instead (4.8 -> 5.1-5.3 TB/s), with streaming stores.
    const uintptr_t src_bits = reinterpret_cast<uintptr_t>(c.dm.src) | (uintptr_t)(c.dm.ss[1] * es) | (uintptr_t)(c.dm.ss[2] * es);
    const uintptr_t dst_bits = reinterpret_cast<uintptr_t>(c.dm.dst) | (uintptr_t)(c.dm.ds[0] * es) | (uintptr_t)(c.dm.ds[2] * es);
    const uintptr_t align_req = 128;
    const bool src_mis = src_bits % align_req != 0, dst_mis = dst_bits % 64 != 0;
    const int window_mode = tuning ? tuning->window_mode : -1;  // -1 auto, 0 never, 1 whenever the destination is misaligned
    c.window = dst_mis && window_mode != 0 && (window_mode == 1 || c.elements * es >= (1ll << 20));
    if (c.window) {
      if (c.stream == 2) c.stream = 4;  // cached loads (the overlap rows hit in L2), streaming whole-unit stores
      j_first = true;
    } else if (src_mis || dst_bits % align_req != 0) {
      if (c.stream == 2) {
        if (dst_bits % align_req != 0) c.stream = 0;
        else c.stream = 4;
      }
      j_first = false;
    }
    // One measured outlier: 16-byte elements whose destination batch stride is not a multiple of 4 KiB (rows padded by a
    // cache line) lose a third of their rate with streaming stores (8 GiB permutation: 4.0 ms streaming, 3.4 ms cached;
    // 4- and 8-byte elements with the same padding prefer streaming, profiles/r02_tuning.md).
    if (es == 16 && c.stream == 2 && !c.window && c.dm.e[2] > 1 && ((uintptr_t)(c.dm.ds[2] * es) % 4096) != 0) c.stream = 0;
    const bool walk_forced = tuning && tuning->walk_order >= 0;
    if (walk_forced) j_first = tuning->walk_order == 1;
    // (window kernel, 4-byte elements: 64 x 128 tiles -- a 64-byte unit is 16 elements, the longer window halves the
    // share of overlap rows)
    // (window kernel, 8-byte elements, optional: 128 x 64 tiles with 512 threads -- 1-KiB source segments span nine
    // lines instead of 2 x five; variant 102)
#ifdef CUDECOMP_TUNING_VARIANTS
    const bool wide = c.window && es == 8 && vw == 2 && tuning && tuning->window_wide == 1;
#else
    const bool wide = false;  // (the 128 x 64 / 512-thread window variant exists in tuning builds only)
#endif
    if (wide) c.variant = 102;
    // (4-byte elements, 16-byte lanes, plain kernel: optional 128 x 64 / 64 x 128 tiles -- variants 204 / 304)
    int shape = 0;
    if (!c.window && es == 4 && vw == 4) shape = 2;
#ifdef CUDECOMP_TUNING_VARIANTS
    if (!c.window && es == 4 && vw == 4 && tuning && tuning->tile_shape >= 0) shape = tuning->tile_shape;
#endif
    if (shape == 1) c.variant = 204;
    else if (shape == 2) c.variant = 304;
    // (shape 0 keeps variant 4 = 64 x 64 tiles: only in builds with CUDECOMP_TUNING_VARIANTS)
    // Large line-aligned moves whose SOURCE rows are the far-strided side (the inverse hops of an axis-contiguous cycle): twice
    // as many source rows per tile, 1-KiB destination segments.  Measured on the 8-GiB permutations (profiles/r05_tuning.md):
    // fp64 64 x 128 2.69 -> 2.65 ms, complex128 32 x 64 2.70 -> 2.66 ms; the forward hops lose with these tiles and keep theirs.
    const bool aligned = !c.window && !src_mis && dst_bits % align_req == 0;
    const bool far_src = aligned && c.stream == 2 && c.dm.ss[1] > 8 * c.dm.ds[0];
    bool tall = false;
    if (far_src && es == 8 && vw == 2 && c.swizzle) {
      c.variant = 302;
      tall = true;
    } else if (far_src && es == 16 && !c.swizzle) {
      c.variant = 301;
      tall = true;
    }
    const int ti = (es == 16) ? 32 : ((wide || shape == 1) ? 128 : 64);
    const int tj = (es == 16) ? (tall ? 64 : 32) : (((c.window && es == 4) || shape == 2 || tall) ? 128 : 64);
    c.t0 = (unsigned int)((c.dm.e[0] + ti - 1) / ti);
    c.t1 = (unsigned int)((c.dm.e[1] + (c.window ? 64 / es - 1 : 0) + tj - 1) / tj);
    // Far-strided DESTINATION (the forward hops of an axis-contiguous cycle: destination rows e.g. 8 MiB apart, source rows
    // near): walk j in RUNS -- kRunBytes of every destination row of a tile row, then the next tile row, then the next run.
    // The workgroups in flight on an XCD then write a few long contiguous runs (64 rows x 256 KiB) instead of one short run
    // in very many rows (plain j first) or 512-byte pieces of 1024 rows (i first).  When the rows of consecutive batch planes
    // are adjacent in the destination the planner has fused them into j (normalizeMove), so a run spans planes; if they are
    // not fused (padded planes) the run is over batch planes instead (p1 bit 4).  Measured on the 8-GiB permutations, two
    // boxes (profiles/r05_tuning.md): fp64 2.91-2.93 -> 2.81 ms, complex128 2.99 -> 2.86, fp32 2.89 -> 2.86; runs of
    // 128 KiB ... 2 MiB are within 1 %.
    c.p0 = 0;
    const bool far_dst = aligned && c.stream == 2 && c.dm.ds[0] > 8 * c.dm.ss[1];
    if (far_dst && !walk_forced) {
      constexpr long long kRunBytes = 256 << 10;
      const long long want = std::max<long long>(1, kRunBytes / ((long long)tj * es));  // tiles of one run
      if ((long long)c.t1 >= 2 * want) {
        long long run = want;
        while (run > 1 && c.t1 % run != 0) --run;  // (a divisor of the tile count: the walk stays a plain mixed-radix number)
        if (run >= want / 4 && run >= 4) {
          c.p0 = (int)run;
          j_first = true;
        }
      } else if (c.dm.e[2] > 1 && c.dm.ds[2] < c.dm.ds[0] && (long long)c.t1 * tj * es <= (64 << 10)) {
        long long run = std::max<long long>(1, kRunBytes / std::max<long long>(1, c.dm.ds[2] * es));
        while (run > 1 && c.dm.e[2] % run != 0) --run;
        if (run >= 4) {
          c.p0 = (int)run;
          c.p1 |= 4;
          j_first = true;
        }
      }
    }
    if (j_first) c.p1 |= 2;
    c.blocks = (unsigned long long)c.t0 * c.t1 * (unsigned long long)c.dm.e[2];
    // Destination rows off the 64-byte grid AND the rows of consecutive batch planes adjacent in memory (forward hops of an
    // axis-contiguous cycle onto a halo-carrying pencil): every row begins and ends inside a cache line whose other part
    // belongs to the next plane -- for the window kernel another workgroup, much later; the partly written lines cost the
    // forward hops a sixth of their rate (0.60 against 0.72 of the HBM peak, profiles/r05_tuning.md section 3).  When the
    // planner says the move covers whole interior rows of the pencil (dst_row_pitch: the gap cells are then halo / padding
    // cells nobody else writes during the operation, the contract of rows_dense_kernel) j and k are fused into the slab's
    // linear positions and the windows run ACROSS the row ends (transpose_lines_kernel, kernels_lines.hip).
    if (c.window && !wide && in.dst_row_pitch > 0 && !remote && (!tuning || (tuning->dense_rows != 0 && tuning->lines_mode != 0))) {
      i64 planned_row = -1;
      for (int i = 0; i < 3; ++i)
        if (in.ds[i] == 1 && in.extent[i] > 1) planned_row = in.extent[i];
      const int ub = linesUnitBytes(tuning ? tuning->lines_unit : 128);
      const long long ej = c.dm.e[1], ek = c.dm.e[2], dk = c.dm.ds[2], gap = dk - ej;
      const long long span = (ek - 1) * dk + ej;
      if (ek > 1 && planned_row == ej && dk == in.dst_row_pitch && gap > 0 && gap * es <= kDenseMaxGapBytes && gap * 8 <= ej &&
          c.dm.ds[0] >= span && span < (1ll << 30) && c.dm.e[0] < (1ll << 30) && dk >= tj + ub / es) {
        c.lines = true;
        c.unit = ub;
        // 16-byte lanes need whole vectors along i only: the windows run over linear positions, whatever the row length
        c.variant = (es < 16 && c.dm.e[0] % (16 / es) == 0) ? 16 / es : 1;
        c.t1 = (unsigned int)((span + ub / es - 1 + tj - 1) / tj);  // windows along the linear positions (+ one unit of phase slack)
        // Tile walk (kernels_lines.hip): groups of 16 tile rows; inside a group short runs of 2 KiB per slab (four windows of
        // 8-byte elements) for all its rows, then the next run -- the source is read plane by plane in whole rows, every
        // slab's write stream advances steadily.  Wider moves (several groups: more than 1024 slabs) take runs of 32 KiB.
        // Measured on two boxes, fp64 forward hops onto halo pencils (profiles/r06_tuning.md): 1024^3 halo 1 window kernel
        // 3.52 ms -> 3.14 (2 KiB; 32 KiB 3.19-3.29, 256 KiB 3.45-3.77); config 5's pencil X->Y (2048 slabs) 1.72-1.77 -> 1.55-1.61
        // (32 KiB; 2 KiB 1.66), Y->Z (260-element rows) 1.81 -> 1.66 (2 KiB; 32 KiB 1.79-1.86).
        const long long group = tuning ? tuning->lines_group : 16;
        const bool several_groups = group > 0 && group < (long long)c.t0;
        const long long run_kib = (tuning && tuning->lines_run_kib >= 0) ? tuning->lines_run_kib : (several_groups ? 32 : 2);
        const long long run = run_kib > 0 ? std::max<long long>(1, (run_kib << 10) / ((long long)tj * es)) : 1;
        c.p0 = (long long)c.t1 >= 2 * run ? (int)run : 0;
        c.p1 = 1 | 2 | 8;  // XCD-contiguous, along the destination first, "lines"
        if (group > 0 && group < (long long)c.t0) c.p1 |= (int)(group << 8);
#ifdef CUDECOMP_TUNING_VARIANTS
        if (tuning && tuning->lines_walk == 2) c.p1 |= 32;
#endif
        c.blocks = (unsigned long long)c.t0 * c.t1;
      }
      // ... and the other orientation: the tile's OWN rows i are the adjacent ones (inverse hops of the cycle, unpack-side
      // permutations; batch planes far apart).  The line at the end of row i holds the gap and the head of row i + 1 -- the
      // same tile column of the next row: a row's last window runs on into it (transpose_rowlines_kernel, kernels_rowlines.hip).
      const long long ei = c.dm.e[0], di = c.dm.ds[0], rgap = di - ej;
      if (!c.lines && planned_row == ej && di == in.dst_row_pitch && rgap > 0 && rgap * es <= kDenseMaxGapBytes && rgap * 8 <= ej &&
          ej > 2 * (tj + ub / es) && ei >= 2 && ei < (1ll << 30) && di < (1ll << 30) && (ek == 1 || dk >= (ei - 1) * di + ej) &&
          ub == 128) {
        c.rowlines = true;
        c.unit = ub;
        c.variant = (es < 16 && ei % (16 / es) == 0) ? 16 / es : 1;
        c.t1 = (unsigned int)((di - 1 + ub / es - 1) / tj + 1);  // windows per row: through the one that holds the last gap cell
        c.p0 = 0;
        c.p1 = 1 | 2 | 16;  // XCD-contiguous, windows first, "row lines"
        c.blocks = (unsigned long long)c.t0 * c.t1 * (unsigned long long)ek;
      }
    }
    return c;
  }

  c.cls = MOVE_GENERIC;
  c.variant = es;
  for (int i = 0; i < 3; ++i) {
    c.dm.e[i] = m.extent[i];
    c.dm.ss[i] = m.ss[i];
    c.dm.ds[i] = m.ds[i];
  }
  c.p0 = 0;
  for (int i = 0; i < 3; ++i)
    if (m.ds[i] == 1 && m.extent[i] > 1) c.p0 = i;
  const unsigned long long want = ((unsigned long long)c.elements + kThreads - 1) / kThreads;
  c.blocks = std::min<unsigned long long>(std::max<unsigned long long>(want, 1), 8192);
  return c;
}

char g_last_kernel[96] = "";

void tileOf(int es, int variant, bool window, int* ti, int* tj) {
  if (es == 16) {
    *ti = 32;
    *tj = variant == 301 ? 64 : 32;
  } else if (window) {
    *ti = variant >= 100 ? 128 : 64;
    *tj = es == 4 ? 128 : 64;
  } else {
    *ti = variant == 204 ? 128 : 64;
    *tj = (variant == 304 || variant == 302) ? 128 : 64;
  }
}

void launchBatch(MoveClass cls, int variant, int stream_access, bool swizzle, bool window, bool dense, int lines_unit, bool rowlines,
                 int es, const Batch& b, unsigned int blocks, hipStream_t stream) {
  // what ran last, in the words of the kernel templates (bench.py reports its dominant kernel from here)
  int ti = 0, tj = 0;
  tileOf(es, variant, window, &ti, &tj);
  if (cls == MOVE_ROWS_VEC && dense)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "rows_dense_kernel<%d>", stream_access >= 1 ? 1 : 0);
  else if (cls == MOVE_ROWS_VEC)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%d,%d>", window ? "rows_shifted_kernel" : "rows_kernel",
             variant, stream_access == 3 ? 3 : (stream_access >= 1 ? 1 : 0));
  else if (cls == MOVE_TRANSPOSE && rowlines)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "transpose_rowlines_kernel<%d,%d,%d,%d,%d,%d>", es, variant % 100, ti, tj,
             (stream_access == 2 || stream_access == 4) ? 4 : 0, lines_unit);
  else if (cls == MOVE_TRANSPOSE && lines_unit)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "transpose_lines_kernel<%d,%d,%d,%d,%d,%d>", es, variant % 100, ti, tj,
             (stream_access == 2 || stream_access == 4) ? 4 : 0, lines_unit);
  else if (cls == MOVE_TRANSPOSE && window)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "transpose_window_kernel<%d,%d,%d,%d,%d>", es, variant % 100, ti, tj,
             (stream_access == 2 || stream_access == 4) ? 4 : stream_access);
  else if (cls == MOVE_TRANSPOSE)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "transpose_kernel<%d,%d,%d,%d,%d,%s>", es, variant % 100, ti, tj, stream_access,
             swizzle ? "true" : "false");
  else
    snprintf(g_last_kernel, sizeof(g_last_kernel), "generic_kernel<%d,%s>", es, stream_access == 3 ? "true" : "false");
  switch (cls) {
    case MOVE_ROWS_VEC:
      launchRowsBatch(dense ? 2 : (window ? 1 : 0), variant, stream_access, b, blocks, stream);
      break;
    case MOVE_TRANSPOSE:
      if (rowlines) launchRowLinesBatch(es, variant % 100, stream_access, b, blocks, stream);
      else if (lines_unit) launchLinesBatch(es, variant % 100, stream_access, lines_unit, b, blocks, stream);
      else if (window) launchWindowBatch(es, variant % 100, variant >= 100, stream_access, b, blocks, stream);
      else if (es == 4) launchTransposeBatch4(variant, stream_access, swizzle, b, blocks, stream);
      else if (es == 8) launchTransposeBatch8(variant, stream_access, swizzle, b, blocks, stream);
      else launchTransposeBatch16(variant, stream_access, swizzle, b, blocks, stream);
      break;
    default:
      launchGenericBatch(es, stream_access == 3, b, blocks, stream);
      break;
  }
}

}  // namespace

const char* lastKernelName() { return g_last_kernel; }

void describeMove(const Move3D& m, const void* src, void* dst, int es, const KernelTuning* tuning, long long out[10]) {
  Move3D mm = m;  // (keeps dst_row_pitch)
  mm.src_buf = BUF_IN;
  mm.dst_buf = BUF_OUT;
  mm.src_off = mm.dst_off = 0;
  void* bufs[3] = {const_cast<void*>(src), dst, nullptr};
  const Classified c = classify(mm, bufs, es, tuning, nullptr, false);
  int ti = 0, tj = 0;
  if (c.cls == MOVE_TRANSPOSE) tileOf(es, c.variant, c.window, &ti, &tj);
  else if (c.cls == MOVE_ROWS_VEC) ti = c.dense ? 2 : (c.window ? 1 : 0);  // rows: the kernel (plain / shifted / dense) in the tile_i slot
  const long long v[10] = {(long long)c.cls, c.variant, ti, tj, c.t0, c.t1, c.dm.e[2], c.p0, c.p1, c.stream};
  for (int i = 0; i < 10; ++i) out[i] = v[i];
}

void launchMoves(const Move3D* moves, int n, void* const bufs[3], int es, hipStream_t stream,
                 const KernelTuning* tuning, KernelStats* stats, void* const* dst_base_override) {
  const bool remote = dst_base_override != nullptr;
  if (es != 4 && es != 8 && es != 16) CD_INTERNAL_ERROR("unsupported element size");
  std::vector<Classified> cs;
  cs.reserve(n);
  for (int i = 0; i < n; ++i) {
    if (moves[i].elements() == 0) continue;
    cs.push_back(classify(moves[i], bufs, es, tuning, dst_base_override ? dst_base_override[i] : nullptr, remote));
  }
  // moves of one phase are independent, so they may be regrouped by kernel flavour
  std::vector<bool> done(cs.size(), false);
  for (size_t i = 0; i < cs.size(); ++i) {
    if (done[i]) continue;
    Batch b{};
    unsigned long long blocks = 0;
    for (size_t j = i; j < cs.size() && b.n < kMaxBatch; ++j) {
      if (done[j] || cs[j].cls != cs[i].cls || cs[j].variant != cs[i].variant || cs[j].stream != cs[i].stream ||
          cs[j].swizzle != cs[i].swizzle || cs[j].window != cs[i].window || cs[j].dense != cs[i].dense ||
          cs[j].lines != cs[i].lines || cs[j].unit != cs[i].unit || cs[j].rowlines != cs[i].rowlines)
        continue;
      if (blocks + cs[j].blocks > 0x7fffffffULL) {
        if (b.n == 0) CD_NOT_SUPPORTED("single block move too large for one launch");
        break;
      }
      b.first_block[b.n] = (unsigned int)blocks;
      b.m[b.n] = cs[j].dm;
      b.p0[b.n] = cs[j].p0;
      b.p1[b.n] = cs[j].p1;
      b.t0[b.n] = cs[j].t0;
      b.t1[b.n] = cs[j].t1;
      blocks += cs[j].blocks;
      if (stats) stats->elements[cs[j].cls] += cs[j].elements;
      ++b.n;
      done[j] = true;
    }
    b.first_block[b.n] = (unsigned int)blocks;
    // Sibling row copies of one phase (the P chunks of an unpack, say) each touch one slice of every destination row:
    // run one after the other, a 2-KiB slice of every 8-KiB row keeps part of the memory channels idle.  Served round
    // robin, the workgroups in flight cover whole rows (C3 per-rank unpacks: 0.43-0.47 -> 0.35 ms, r02_tuning.md).
    // Transposes keep their XCD-contiguous tile walk (interleaving them measured slightly slower).
    const bool il_local = cs[i].cls != MOVE_TRANSPOSE && (!tuning || tuning->interleave_rows != 0);
    if ((dst_base_override || il_local) && b.n > 1) {
      unsigned long long widest = 0;
      for (int k = 0; k < b.n; ++k) widest = std::max<unsigned long long>(widest, b.first_block[k + 1] - b.first_block[k]);
      if (widest * b.n <= 0x7fffffffULL) {
        b.interleave = 1;
        blocks = widest * b.n;
      }
    }
    launchBatch(cs[i].cls, cs[i].variant, cs[i].stream, cs[i].swizzle, cs[i].window, cs[i].dense,
                (cs[i].lines || cs[i].rowlines) ? cs[i].unit : 0, cs[i].rowlines, es, b, (unsigned int)blocks, stream);
    if (stats) stats->launches[cs[i].cls] += 1;
  }
}

}  // namespace cudecomp
