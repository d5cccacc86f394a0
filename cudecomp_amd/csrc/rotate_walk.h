// rotate_walk.h -- which orbit of tiles a workgroup of rotate_kernel (kernels_rotate.hip) takes.  One definition for the kernel
// (device) and for cudecompExtRotateWalk (host): tests/test_kernel_plan.py checks on the CPU that every walk visits every block
// triple exactly once.
#pragma once

#if defined(__HIPCC__)
#define CD_WALK_HD __host__ __device__
#else
#define CD_WALK_HD
#endif

namespace cudecomp {

// A walk is  cl | a << 4 | b << 8 | c << 12:
//   cl = 15: per XCD.  Workgroup w runs on XCD w % 8 (round-robin dispatch, used for speed only); s = w / 8 numbers the workgroups
//            of an XCD: b0 = s % nb, (b1, b2) from the rest and the XCD.
//   cl = 0..5 (tuning builds, CUDECOMP_ROTATE_WALK): b0 fastest inside cubes of 2^cl blocks per edge, cubes c0 fastest.
//   then the shears  b2 += b * b0 + c * b1,  b1 += a * b0  (mod nb): bijections of the triples.
// Why: the three tiles of an orbit (b0,b1,b2), (b2,b0,b1), (b1,b2,b0) have the p0 blocks b0, b2 and b1 -- address bits 7 and up,
// the bits that choose the L2 channel inside an XCD and the HBM channel behind it.  With b0 = w % nb an XCD would see an eighth of
// the b0 values and ONE value of b1 and of b2 for thousands of consecutive workgroups: 0.56 of the HBM peak at 1024^3 fp64, 0.50
// for complex128; per XCD with shears 0.64 / 0.62 (profiles/r06_tuning.md section 8: 265 walks, the best dozen within 1 %).
// Measured beside it (TCC_REQ per XCC, profiles/r06_rotate_l2_requests_per_xcc.json): with b0 = w % nb the XCDs do not get the same
// amount of work either -- owners are the triples whose b2 is the smallest, so they are denser at large b0, and XCD x only sees
// b0 = x mod 8: XCD 7 serves 8 % more than the mean, XCD 0 9 % less; the per-XCD walk is level within 0.7 %.
constexpr int kRotateWalk = 15 | 1 << 4 | 3 << 8 | 2 << 12;

// the walk as launched for nb blocks per edge: the cube edge never exceeds the array
inline int rotateWalkFor(int walk, long long nb) {
  if (walk < 0) walk = kRotateWalk;
  int cl = walk & 15;
  if (cl != 15) {
    if (cl > 5) cl = 5;
    while (cl > 0 && (1ll << cl) > nb) --cl;
  }
  return (walk & 0xfff0) | cl;
}

// workgroups to launch (some map to no block: padding of the last cube / of the last group of eight)
inline long long rotateWalkGrid(long long nb, int walk) {
  const int cl = walk & 15;
  if (cl == 15) return 8 * nb * ((nb * nb + 7) / 8);
  const long long nc = (nb + (1ll << cl) - 1) >> cl;
  return nc * nc * nc << (3 * cl);
}

// block triple of workgroup wg; false: none
CD_WALK_HD inline bool rotateWalkBlock(unsigned int wg, unsigned int nb, int walk, int* b0_out, int* b1_out, int* b2_out) {
  const int cl = walk & 15;
  unsigned int b0, b1, b2;
  if (cl == 15) {
    const unsigned int x = wg & 7u, s = wg >> 3, m = (s / nb) * 8u + x;
    if (m >= nb * nb) return false;
    b0 = s % nb, b1 = m % nb, b2 = m / nb;
  } else {
    const unsigned int cm = (1u << cl) - 1u;
    const unsigned int within = wg & ((1u << (3 * cl)) - 1u), cube = wg >> (3 * cl);
    const unsigned int nc = (nb + cm) >> cl;
    b0 = ((cube % nc) << cl) + (within & cm);
    b1 = (((cube / nc) % nc) << cl) + ((within >> cl) & cm);
    b2 = ((cube / (nc * nc)) << cl) + (within >> (2 * cl));
    if (b0 >= nb || b1 >= nb || b2 >= nb) return false;
  }
  const unsigned int a = (unsigned int)(walk >> 4) & 15u, b = (unsigned int)(walk >> 8) & 15u, c = (unsigned int)(walk >> 12) & 15u;
  b2 = (b2 + b * b0 + c * b1) % nb;
  b1 = (b1 + a * b0) % nb;
  *b0_out = (int)b0, *b1_out = (int)b1, *b2_out = (int)b2;
  return true;
}

}  // namespace cudecomp
