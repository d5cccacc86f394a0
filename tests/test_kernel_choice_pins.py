"""The CHOICES of the kernel layer, pinned: for the moves the bench, BASELINE config 3 and config 5 produce (plans from the
stateless planner, addresses as cudecompMalloc / hipMalloc hand them out: 256-byte aligned bases) the classifier of
csrc/kernels.cc must pick exactly the kernel, tile, tile walk and access mode recorded in tests/golden/kernel_choice_pins.json.
Every threshold in classify() cites a measurement; this table makes an edit of one of them show up as a diff of the choice, not
only -- maybe -- of a timing.  After a DELIBERATE change:  python tests/test_kernel_choice_pins.py --regen  and commit the diff.
Entry = [class (0 rows, 1 LDS transpose, 2 generic), variant, tile_i (rows: 0 plain / 1 shifted / 2 dense), tile_j, p0 (run length /
lanes-per-row log2), p1 (walk bits; rows: row bytes of the shifted kernels), access mode]; no GPU needed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import cudecomp_amd as cd  # noqa: E402

PINS = os.path.join(ROOT, "tests", "golden", "kernel_choice_pins.json")
BASE = {0: 1 << 32, 1: 1 << 36, 2: 1 << 40}   # input, output, workspace
KEYS = ("cls", "variant", "tile_i", "tile_j", "run", "walk", "access")


def _orders(ac):
    return [[(ax + i) % 3 if ac[ax] else i for i in range(3)] for ax in range(3)]


def _describe(m, es):
    d = cd.cudecompExtDescribeMove(BASE[m.src_buf] + m.src_off * es, BASE[m.dst_buf] + m.dst_off * es, es, list(m.extent),
                                   list(m.ss), list(m.ds), row_pitch=m.row_pitch)
    return [d[k] for k in KEYS]


def _transposes(gdims, pdims, ac, es, rank=0, halo=None, inplace=False, **kw):
    g = cd.make_grid_spec(gdims, pdims, _orders(ac))
    out = {}
    for op in cd.OPS:
        p = cd.cudecompExtPlanTranspose(g, rank, op, halo, halo, None, None, inplace, **kw)
        pack = sorted({tuple(_describe(m, es)) for m in list(p.pack)[:p.n_pack]})
        unpack = sorted({tuple(_describe(m, es)) for m in list(p.unpack)[:p.n_unpack]})
        out[op] = {"pack": [list(t) for t in pack], "unpack": [list(t) for t in unpack], "rotate": p.rotate}
    return out


def _halos(gdims, pdims, ac, es, halo, rank=0):
    g = cd.make_grid_spec(gdims, pdims, _orders(ac))
    out = {}
    for axis in range(3):
        for dim in range(3):
            p = cd.cudecompExtPlanHalo(g, rank, axis, halo, (1, 1, 1), dim)
            moves = list(p.pre)[:p.n_pre] + list(p.post)[:p.n_post]
            out["%s%d" % ("XYZ"[axis], dim)] = [list(t) for t in sorted({tuple(_describe(m, es)) for m in moves})]
    return out


def current():
    n = 1024
    return {
        "bench_1x1_contiguous_f64": _transposes((n, n, n), (1, 1), (1, 1, 1), 8),
        "bench_1x1_contiguous_f64_halo1": _transposes((n, n, n), (1, 1), (1, 1, 1), 8, halo=(1, 1, 1)),
        "bench_1x1_default_f64_halo1": _transposes((n, n, n), (1, 1), (0, 0, 0), 8, halo=(1, 1, 1)),
        "bench_1x1_contiguous_f64_in_place": _transposes((n, n, n), (1, 1), (1, 1, 1), 8, inplace=True),
        "dtypes_1x1_contiguous_f32_2048x1024x1024": _transposes((2048, n, n), (1, 1), (1, 1, 1), 4),
        "dtypes_1x1_contiguous_c128_1024x1024x512": _transposes((n, n, 512), (1, 1), (1, 1, 1), 16),
        "config1_256cube_f32_2x1_rank0": _transposes((256, 256, 256), (2, 1), (0, 0, 0), 4),
        "config2_512cube_f64_2x1_rank1": _transposes((512, 512, 512), (2, 1), (0, 0, 0), 8, rank=1),
        "config3_2x4_default_f64_rank0": _transposes((n, n, n), (2, 4), (0, 0, 0), 8),
        "config3_1x8_contiguous_f64_rank0": _transposes((n, n, n), (1, 8), (1, 1, 1), 8),
        "config3_2x4_contiguous_f64_staged_pipeline_rank0": _transposes((n, n, n), (2, 4), (1, 1, 1), 8, pipelined=True, symmetric_recv=True),
        "config4_2x4_contiguous_c64_rank5": _transposes((n, n, n), (2, 4), (1, 1, 1), 8, rank=5),
        "config5_pencil_1x1_f64_halo2_transposes": _transposes((2048, 1024, 256), (1, 1), (1, 1, 1), 8, halo=(2, 2, 2)),
        "config5_2x4_f64_halo2_faces_rank0": _halos((2048, 2048, 1024), (2, 4), (0, 0, 0), 8, (2, 2, 2)),
    }


def test_kernel_choices_are_the_pinned_ones():
    with open(PINS) as f:
        pinned = json.load(f)
    now = current()
    assert sorted(now) == sorted(pinned), "scenario list changed: regenerate the pins"
    diffs = []
    for name in sorted(now):
        for op in sorted(now[name]):
            if now[name][op] != pinned[name][op]:
                diffs.append("%s %s: pinned %s, now %s" % (name, op, pinned[name][op], now[name][op]))
    assert not diffs, "kernel choices differ from tests/golden/kernel_choice_pins.json:\n" + "\n".join(diffs)


def test_pins_say_what_the_design_says():
    """A few readable facts of the table, so that a regenerated file that silently lost a fast path fails here."""
    with open(PINS) as f:
        p = json.load(f)
    b = p["bench_1x1_contiguous_f64"]
    assert b["XToY"]["pack"] == [[1, 2, 64, 64, 512, 3, 2]] and b["YToX"]["pack"] == [[1, 302, 64, 128, 0, 3, 2]]   # run walk / tall tiles
    h = p["bench_1x1_contiguous_f64_halo1"]
    assert h["XToY"]["pack"][0][5] & 8 and h["XToY"]["pack"][0][6] == 4      # forward hops onto halo pencils: transpose_lines_kernel
    assert not h["ZToY"]["pack"][0][5] & 8 and h["ZToY"]["pack"][0][6] == 4  # inverse hops: the window kernel
    assert all(v["pack"][0][:3] == [0, 16, 2] for v in p["bench_1x1_default_f64_halo1"].values())   # rows_dense_kernel
    assert [p["bench_1x1_contiguous_f64_in_place"][op]["rotate"] for op in cd.OPS] == [1, 1, -1, -1]
    assert p["config5_2x4_f64_halo2_faces_rank0"]["X2"] == []               # contiguous faces travel without a kernel


def dump(table, f):
    """One line per (scenario, operation): a change of one choice is a one-line diff."""
    f.write("{\n")
    names = sorted(table)
    for i, name in enumerate(names):
        f.write(' %s: {\n' % json.dumps(name))
        ops = sorted(table[name])
        for k, op in enumerate(ops):
            f.write('  %s: %s%s\n' % (json.dumps(op), json.dumps(table[name][op], sort_keys=True), "," if k + 1 < len(ops) else ""))
        f.write(" }%s\n" % ("," if i + 1 < len(names) else ""))
    f.write("}\n")


if __name__ == "__main__" and "--regen" in sys.argv:
    with open(PINS, "w") as f:
        dump(current(), f)
    print("wrote", PINS)
