"""Index maps through the real C ABI (libcudecomp.so) on 4 ranks against the reference's golden vectors,
and on several grids against the oracle.  CPU only: geometry queries need no GPU."""
import json
import os

import pytest

from oracle import oracle as orc
from tests import bodies
from tests.mp import run_ranks


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("variant", ["row_major", "col_major", "gdims_dist"])
def test_pencil_info_golden_4_ranks(golden_dir, variant):
    gold = load(golden_dir, "pencil_info.json")
    args = {"gdims": gold["gdims"], "pdims": gold["pdims"], "halo": gold["halo_extents"],
            "padding": gold["padding"], "rank_order": 2 if variant == "col_major" else 0,
            "gdims_dist": gold["gdims_dist_case"] if variant == "gdims_dist" else None}
    res = {r["rank"]: r for r in run_ranks(4, "tests.bodies", "index_queries", args)}
    for rec in gold[variant]:
        got = res[rec["rank"]]["pencil"][rec["axis"]]
        for key in ("shape", "lo", "hi", "order", "halo_extents", "padding", "size"):
            assert got[key] == rec[key], (variant, rec["axis"], rec["rank"], key)


@pytest.mark.parametrize("variant", ["row_major", "col_major"])
def test_shifted_rank_golden_4_ranks(golden_dir, variant):
    gold = load(golden_dir, "shifted_rank.json")
    args = {"gdims": gold["gdims"], "pdims": gold["pdims"], "rank_order": 2 if variant == "col_major" else 0,
            "shifted_queries": gold[variant]}
    res = {r["rank"]: r for r in run_ranks(4, "tests.bodies", "index_queries", args)}
    for qi, rec in enumerate(gold[variant]):
        assert [res[r]["shifted"][qi] for r in range(4)] == rec["expected_by_rank"], rec


@pytest.mark.parametrize("nranks,pdims", [(1, (1, 1)), (2, (2, 1)), (2, (1, 2)), (6, (2, 3)), (6, (3, 2)), (8, (2, 4))])
def test_index_queries_match_oracle(nranks, pdims):
    for ac, mo, gdd, ro in (((0, 0, 0), None, None, 0), ((1, 1, 1), None, (20, 13, 17), 2),
                            ((0, 0, 0), ((2, 0, 1), (1, 0, 2), (0, 2, 1)), None, 1)):
        args = {"gdims": (23, 14, 19), "pdims": pdims, "halo": (1, 2, 0), "padding": (0, 1, 3), "ac": ac,
                "mem_order": mo, "gdims_dist": gdd, "rank_order": ro}
        g = orc.Grid(args["gdims"], pdims, gdims_dist=gdd, rank_order=ro, axis_contiguous=ac, mem_order=mo)
        if nranks == 1:
            res = [bodies.index_queries(0, 1, args)]
        else:
            res = run_ranks(nranks, "tests.bodies", "index_queries", args)
        for r in res:
            for axis in range(3):
                assert r["pencil"][axis] == g.pencil_info(r["rank"], axis, args["halo"], args["padding"]).as_dict()
                assert r["halo_ws"][axis] == g.halo_workspace_size(r["rank"], axis, args["halo"])
            assert r["transpose_ws"] == g.transpose_workspace_size()
            assert r["config_pdims"] == list(pdims)
            assert r["config_rank_order"] == (2 if ro == 2 else 1)
