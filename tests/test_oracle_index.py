"""The oracle's index maps against the reference's golden vectors (tests/golden/*.json, extracted
from tests/ctest/api_tests.cc by tests/golden/make_golden.py)."""
import json
import os

import pytest

from oracle import oracle as orc


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("variant", ["row_major", "col_major", "gdims_dist"])
def test_pencil_info_golden(golden_dir, variant):
    gold = load(golden_dir, "pencil_info.json")
    g = orc.Grid(gold["gdims"], gold["pdims"],
                 gdims_dist=gold["gdims_dist_case"] if variant == "gdims_dist" else None,
                 rank_order=orc.COL_MAJOR if variant == "col_major" else 0)
    assert len(gold[variant]) == 12
    for rec in gold[variant]:
        p = g.pencil_info(rec["rank"], rec["axis"], gold["halo_extents"], gold["padding"]).as_dict()
        for key in ("shape", "lo", "hi", "order", "halo_extents", "padding", "size"):
            assert p[key] == rec[key], (variant, rec["axis"], rec["rank"], key)


@pytest.mark.parametrize("variant", ["row_major", "col_major"])
def test_shifted_rank_golden(golden_dir, variant):
    gold = load(golden_dir, "shifted_rank.json")
    g = orc.Grid(gold["gdims"], gold["pdims"], rank_order=orc.COL_MAJOR if variant == "col_major" else 0)
    for rec in gold[variant]:
        got = [g.shifted_rank(r, rec["axis"], rec["dim"], rec["displacement"], rec["periodic"]) for r in range(4)]
        assert got == rec["expected_by_rank"], rec


def test_shifted_rank_axis_aligned_and_zero():
    # tests/ctest/api_tests.cc:1410-1433
    g = orc.Grid((9, 10, 11), (2, 2))
    for r in range(4):
        assert g.shifted_rank(r, 0, 1, 0, False) == r
        assert g.shifted_rank(r, 0, 0, 1, False) == -1
        assert g.shifted_rank(r, 0, 0, 1, True) == r
        assert g.shifted_rank(r, 0, 1, 2, True) == r
        assert g.shifted_rank(r, 0, 1, 2, False) == -1


def test_pencil_info_rejects_bad_extents():
    # tests/ctest/api_tests.cc:1310-1352
    g = orc.Grid((9, 10, 11), (2, 2))
    assert g.pencil_info_rc(0, 0, (-1, 0, 0), None)[0] == orc.INVALID_USAGE
    assert g.pencil_info_rc(0, 0, None, (0, -1, 0))[0] == orc.INVALID_USAGE
    assert g.pencil_info_rc(0, 0, (2**31 - 1, 0, 0), None)[0] == orc.INVALID_USAGE
    assert g.pencil_info_rc(0, 0, None, (2**31 - 1, 0, 0))[0] == orc.INVALID_USAGE
    assert g.pencil_info_rc(0, 3, None, None)[0] == orc.INVALID_USAGE
    big = orc.Grid((2**31 - 1,) * 3, (2, 2))
    assert big.pencil_info_rc(0, 0, None, None)[0] == orc.INVALID_USAGE  # size overflow


def test_empty_pencils_have_zero_size():
    # tests/ctest/api_tests.cc:1292-1308 (more ranks than points along an axis)
    g = orc.Grid((1, 2, 2), (2, 2))
    for axis in range(3):
        for r in range(4):
            p = g.pencil_info(r, axis)
            if 0 in list(p.shape):
                assert p.size == 0


def test_splits_and_alignment():
    assert orc.get_splits(10, 4, 0) == [3, 3, 2, 2]
    assert orc.get_splits(9, 2, 1) == [5, 5]          # gdims 10, gdims_dist 9: surplus on the last populated rank
    assert orc.get_splits(2, 4, 3) == [1, 4, 0, 0]    # fewer points than ranks
    assert orc.align_count(1) == 64 and orc.align_count(64) == 64 and orc.align_count(65) == 128
    assert orc.align_count(0) == 0


def test_workspace_sizes_config_table():
    # SURVEY.md section 8 config table (derived from src/cudecomp.cc:1411-1459)
    g = orc.Grid((1024, 1024, 1024), (2, 4))
    assert g.transpose_workspace_size() == 2 * 134217728
    g5 = orc.Grid((2048, 2048, 1024), (2, 4))
    h = (2, 2, 2)
    assert g5.halo_workspace_size(0, 0, h) == 4 * orc.align_count(2052 * 1028 * 2)


@pytest.mark.parametrize("n,npg", [(2, 2), (4, 4), (8, 8), (3, 3), (6, 6), (6, 3), (6, 2), (12, 4), (5, 1)])
def test_peer_schedule_is_a_matching(n, npg):
    # every step pairs each rank's destination with that destination's source, and all peers are visited once
    for it in range(1, n):
        for r in range(n):
            s, d = orc.peer_ranks(n, npg, r, it)
            assert orc.peer_ranks(n, npg, d, it)[0] == r
            assert orc.peer_ranks(n, npg, s, it)[1] == r
    for r in range(n):
        assert sorted(orc.peer_ranks(n, npg, r, it)[1] for it in range(n)) == list(range(n))
