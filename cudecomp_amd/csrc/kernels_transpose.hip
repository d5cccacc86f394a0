// kernels_transpose.hip -- instantiations of the LDS-tiled transposition (kernels_tile.h) for ONE element size; compiled three
// times (-DCUDECOMP_TRANSPOSE_ES=4 | 8 | 16), i.e. three code objects (see kernels_dev.h for why there are several).
//
// Tiles per element size (elements, i x j; i runs along the source rows, j along the destination rows):
//    4-byte   64 x 128 with 16-byte lanes (512-byte destination segments: profiles/r04_tuning.md), 64 x 64 element-wise
//    8-byte   64 x 64; 64 x 128 for large moves whose SOURCE rows are the far-strided side (profiles/r05_tuning.md)
//   16-byte   32 x 32 (padded LDS rows; the swizzled layout measures slower for them); 32 x 64 for far-strided sources
// Access modes 0 / 2 / 3 / 4 (storePolicyOf); mode 1 and the other tile shapes exist in `make TUNING_VARIANTS=1` builds only.
#include "kernels_tile.h"

#include "errors.h"

#ifndef CUDECOMP_TRANSPOSE_ES
#error "compile with -DCUDECOMP_TRANSPOSE_ES=4, 8 or 16"
#endif

namespace cudecomp {
using namespace kern;

namespace {

template <int STREAM, bool SWZ>
void launchT(int variant, const Batch& b, unsigned int blocks, hipStream_t stream) {
  const dim3 grid(blocks), block(kThreads);
#if CUDECOMP_TRANSPOSE_ES == 4
#ifdef CUDECOMP_TUNING_VARIANTS  // the 64 x 64 and 128 x 64 tiles of the round-4 A/B
  if (variant == 204) transpose_kernel<4, 4, 128, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
  else if (variant == 4) transpose_kernel<4, 4, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
  else
#endif
  if (variant == 304) transpose_kernel<4, 4, 64, 128, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
  else transpose_kernel<4, 1, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
#elif CUDECOMP_TRANSPOSE_ES == 8
  if (variant == 302) {
    if constexpr (STREAM == 2 && SWZ) transpose_kernel<8, 2, 64, 128, 2, true><<<grid, block, 0, stream>>>(b);
    else CD_INTERNAL_ERROR("64 x 128 tiles are instantiated for streaming moves only");
  } else if (variant == 2) transpose_kernel<8, 2, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
  else transpose_kernel<8, 1, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
#else
  if (variant == 301) {
    if constexpr (STREAM == 2 && !SWZ) transpose_kernel<16, 1, 32, 64, 2, false><<<grid, block, 0, stream>>>(b);
    else CD_INTERNAL_ERROR("32 x 64 tiles are instantiated for streaming moves only");
  } else transpose_kernel<16, 1, 32, 32, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
#endif
  CD_CHECK_HIP(hipGetLastError());
}

// default builds: 4- and 8-byte elements use the XOR-swizzled LDS tile, 16-byte elements the padded one; tuning builds have both
template <int STREAM>
void launchS(int variant, bool swizzle, const Batch& b, unsigned int blocks, hipStream_t stream) {
#ifdef CUDECOMP_TUNING_VARIANTS
  if (swizzle) launchT<STREAM, true>(variant, b, blocks, stream);
  else launchT<STREAM, false>(variant, b, blocks, stream);
#else
  constexpr bool kSwizzled = CUDECOMP_TRANSPOSE_ES != 16;
  if (swizzle != kSwizzled) CD_INTERNAL_ERROR("this LDS tile layout exists in `make TUNING_VARIANTS=1` builds only");
  launchT<STREAM, kSwizzled>(variant, b, blocks, stream);
#endif
}

void launchAny(int variant, int stream_access, bool swizzle, const Batch& b, unsigned int blocks, hipStream_t stream) {
  if (stream_access == 4) launchS<4>(variant, swizzle, b, blocks, stream);
  else if (stream_access == 3) launchS<3>(variant, swizzle, b, blocks, stream);
  else if (stream_access == 2) launchS<2>(variant, swizzle, b, blocks, stream);
#ifdef CUDECOMP_TUNING_VARIANTS
  else if (stream_access == 1) launchS<1>(variant, swizzle, b, blocks, stream);
#endif
  else launchS<0>(variant, swizzle, b, blocks, stream);
}

}  // namespace

#if CUDECOMP_TRANSPOSE_ES == 4
void launchTransposeBatch4(int variant, int stream_access, bool swizzle, const Batch& b, unsigned int blocks, hipStream_t stream) {
  launchAny(variant, stream_access, swizzle, b, blocks, stream);
}
#elif CUDECOMP_TRANSPOSE_ES == 8
void launchTransposeBatch8(int variant, int stream_access, bool swizzle, const Batch& b, unsigned int blocks, hipStream_t stream) {
  launchAny(variant, stream_access, swizzle, b, blocks, stream);
}
#else
void launchTransposeBatch16(int variant, int stream_access, bool swizzle, const Batch& b, unsigned int blocks, hipStream_t stream) {
  launchAny(variant, stream_access, swizzle, b, blocks, stream);
}
#endif

}  // namespace cudecomp
