"""Case matrices shared by the oracle tests (CPU) and the HIP parity tests (GPU).

They restate the reference's own test matrices:
  * tests/ctest/transpose_tests.cc:163-273  (baseline sweep + coverage cases)
  * tests/ctest/halo_tests.cc:103-146
  * tests/test_config.yaml + tests/test_runner.py:80-90 (legacy sweep: uneven 128x124x132 grid,
    all 36 (X,Y,Z=X) memory-order combinations, halos, padding, gdims_dist, rank orders)
"""
import itertools

OPS = ["XToY", "YToZ", "ZToY", "YToX"]
BASE_GDIMS = (9, 10, 11)
DEFAULT_AC = (0, 0, 0)
ALL_AC = (1, 1, 1)
# tests/ctest/transpose_tests.cc:35-40
IN_HALO, OUT_HALO, IN_PAD, OUT_PAD = (1, 2, 1), (2, 1, 1), (1, 1, 2), (2, 1, 1)
ZERO = (0, 0, 0)


def tcase(name, op, gdims=BASE_GDIMS, pdims=(2, 2), kind=0, out_of_place=False, ac=DEFAULT_AC, mem_order=None,
          in_halo=ZERO, out_halo=ZERO, in_pad=ZERO, out_pad=ZERO, rank_order=0, gdims_dist=None):
    return dict(name=name, op=op, gdims=gdims, pdims=pdims, kind=kind, out_of_place=out_of_place, ac=ac,
                mem_order=mem_order, in_halo=in_halo, out_halo=out_halo, in_pad=in_pad, out_pad=out_pad,
                rank_order=rank_order, gdims_dist=gdims_dist)


def with_hp(c):
    c = dict(c)
    c.update(in_halo=IN_HALO, out_halo=OUT_HALO, in_pad=IN_PAD, out_pad=OUT_PAD)
    return c


def ctest_transpose_cases(pdims_list=((1, 1), (2, 2))):
    """Baseline + coverage cases of tests/ctest/transpose_tests.cc."""
    cases = []
    for lname, ac in (("DefaultLayout", DEFAULT_AC), ("AxisContiguous", ALL_AC)):
        for pdims in pdims_list:
            for kind in (0, 2):
                for op in OPS:
                    for oop in (False, True):
                        cases.append(tcase("Baseline" + lname, op, pdims=pdims, kind=kind, out_of_place=oop, ac=ac))
    unpack = ((0, 1, 2), (0, 1, 2), (0, 1, 2))
    t_unpack = ((0, 2, 1), (0, 1, 2), (0, 1, 2))
    split = ((0, 1, 2), (0, 2, 1), (1, 2, 0))
    cases.append(with_hp(tcase("ExplicitMemOrderUnpack", "XToY", out_of_place=True, mem_order=unpack)))
    cases.append(with_hp(tcase("ExplicitMemOrderTransposeUnpack", "XToY", out_of_place=True, mem_order=t_unpack)))
    for op in OPS:
        cases.append(with_hp(tcase("ExplicitMemOrderSplitUnpack", op, out_of_place=True, mem_order=split)))
    cases.append(tcase("ColumnMajorRankOrder", "XToY", rank_order=2))
    cases.append(with_hp(tcase("NonPowerOfTwoCommunicator", "XToY", pdims=(3, 1), out_of_place=True,
                               mem_order=unpack)))
    cases.append(with_hp(tcase("DtypeWorkspacePadding", "XToY", kind=1, out_of_place=True, mem_order=split)))
    cases.append(with_hp(tcase("DtypeWorkspacePadding", "YToZ", kind=3, out_of_place=True, mem_order=split)))
    # pipelined-only coverage (transpose_tests.cc:239-273)
    cases.append(with_hp(tcase("ExplicitMemOrderTransposePackOffset", "XToY", out_of_place=True,
                               mem_order=((1, 0, 2), (1, 2, 0), (0, 1, 2)))))
    cases.append(with_hp(tcase("DirectTransposePackOffset", "XToY", pdims=(1, 1), out_of_place=True,
                               mem_order=((1, 0, 2), (2, 1, 0), (0, 1, 2)))))
    cases.append(with_hp(tcase("DirectTransposeUnpackOffset", "XToY", pdims=(1, 1), out_of_place=True,
                               mem_order=((1, 0, 2), (0, 1, 2), (0, 1, 2)))))
    cases.append(tcase("NativeAlltoAllPath", "YToZ", gdims=(8, 8, 8), pdims=(1, 4), out_of_place=True))
    return cases


def case_id(c):
    mo = "" if c["mem_order"] is None else "_mo" + "".join("".join(map(str, r)) for r in c["mem_order"])
    hp = "_hp" if any(c["in_halo"]) or any(c["in_pad"]) or any(c["out_halo"]) or any(c["out_pad"]) else ""
    return "%s_%s_k%d_P%dx%d_%s%s%s_ac%s_ro%d" % (
        c["name"], c["op"], c["kind"], c["pdims"][0], c["pdims"][1], "oop" if c["out_of_place"] else "inp", mo, hp,
        "".join(map(str, c["ac"])), c["rank_order"])


def mem_order_combos():
    """tests/test_runner.py:80-90: 36 combinations, Z order == X order."""
    perms = list(itertools.permutations((0, 1, 2)))
    return [(x, y, x) for x, y in itertools.product(perms, perms)]


def pdims_for(nranks):
    """tests/test_runner.py: first / middle / last factor of the rank count."""
    f = [i for i in range(1, nranks + 1) if nranks % i == 0]
    if len(f) > 3:
        f = [f[0], f[len(f) // 2], f[-1]]
    return [(p, nranks // p) for p in f]


# halo cases: tests/ctest/halo_tests.cc:103-146
HALO_EXT = (1, 3, 2)  # kBaselineHaloExtents (halo_tests.cc)
HALO_PAD = (1, 2, 1)  # kNonzeroPadding


def hcase(name, axis, gdims=BASE_GDIMS, pdims=(2, 2), kind=0, ac=DEFAULT_AC, mem_order=None, halo=HALO_EXT,
          periods=(1, 1, 1), padding=ZERO, rank_order=0):
    return dict(name=name, axis=axis, gdims=gdims, pdims=pdims, kind=kind, ac=ac, mem_order=mem_order, halo=halo,
                periods=periods, padding=padding, rank_order=rank_order)


def ctest_halo_cases():
    cases = []
    for lname, ac in (("DefaultLayout", DEFAULT_AC), ("AxisContiguous", ALL_AC)):
        for axis in range(3):
            for kind in (0, 2):
                cases.append(hcase("Baseline%sPeriodic" % lname, axis, kind=kind, ac=ac, periods=(1, 1, 1)))
                cases.append(hcase("Baseline%sNonPeriodic" % lname, axis, kind=kind, ac=ac, periods=(0, 0, 0)))
    cases.append(hcase("NonzeroPadding", 0, padding=HALO_PAD))
    cases.append(hcase("ColumnMajorRankOrder", 0, rank_order=2))
    cases.append(hcase("InteriorNonPeriodicNeighbors", 0, pdims=(3, 1), periods=(0, 0, 0)))
    cases.append(hcase("DtypeWorkspacePadding", 0, kind=1, padding=HALO_PAD))
    cases.append(hcase("DtypeWorkspacePadding", 0, kind=3, padding=HALO_PAD))
    return cases


def hcase_id(c):
    return "%s_ax%d_k%d_P%dx%d_per%s_pad%s_ac%s_ro%d" % (
        c["name"], c["axis"], c["kind"], c["pdims"][0], c["pdims"][1], "".join(map(str, c["periods"])),
        "".join(map(str, c["padding"])), "".join(map(str, c["ac"])), c["rank_order"])
