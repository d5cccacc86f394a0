"""The C-ABI library: loads, exports every symbol the headers declare, struct layouts, constants, argument
checking.  No GPU needed (no compute entry point is called with valid buffers)."""
import ctypes as C
import json
import os
import re

import pytest

import cudecomp_amd as cd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(cudecomp\w+)\s*\(", src))
    inline = set(re.findall(r"static inline \w+\s+(cudecomp\w+)\s*\(", src))
    return names - inline


def test_library_exports_every_declared_symbol():
    L = cd.lib()
    declared = declared_functions("cudecomp.h")
    assert declared == set(cd.API_SYMBOLS)
    for name in sorted(declared | declared_functions("cudecomp_ext.h")):
        assert hasattr(L, name), name
    assert declared_functions("cudecomp_ext.h") == set(cd.EXT_SYMBOLS)


def test_struct_layouts_match_reference_abi():
    # reference src/cudecomp.cc:216,242,268 and the field offsets of include/cudecomp.h:128-238
    assert C.sizeof(cd.GridDescConfig) == 104
    assert C.sizeof(cd.GridDescAutotuneOptions) == 320
    assert C.sizeof(cd.PencilInfo) == 96
    assert cd.GridDescConfig.transpose_mem_order.offset == 60
    assert cd.GridDescConfig.halo_comm_backend.offset == 96
    assert cd.GridDescAutotuneOptions.skip_threshold.offset == 40
    assert cd.GridDescAutotuneOptions.halo_padding.offset == 304
    assert cd.PencilInfo.size.offset == 88


def test_defaults_match_golden(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "api_constants.json")))
    c = cd.cudecompGridDescConfigSetDefaults()
    assert (c.struct_size, c.magic, c.version) == (104, cd.GRID_DESC_CONFIG_MAGIC, 1)
    assert gold["config_defaults"]["transpose_comm_backend"] == "CUDECOMP_TRANSPOSE_COMM_MPI_P2P"
    assert c.transpose_comm_backend == cd.TRANSPOSE_COMM_MPI_P2P
    assert c.halo_comm_backend == cd.HALO_COMM_MPI and c.rank_order == cd.RANK_ORDER_DEFAULT
    assert list(c.pdims) == [0, 0] and list(c.gdims) == [0, 0, 0] and list(c.gdims_dist) == [0, 0, 0]
    assert not any(c.transpose_axis_contiguous)
    assert all(v == -1 for row in c.transpose_mem_order for v in row)
    o = cd.cudecompGridDescAutotuneOptionsSetDefaults()
    d = gold["autotune_option_defaults"]
    assert (o.struct_size, o.magic, o.version) == (320, cd.GRID_DESC_AUTOTUNE_OPTIONS_MAGIC, 1)
    assert o.n_warmup_trials == int(d["n_warmup_trials"]) and o.n_trials == int(d["n_trials"])
    assert o.grid_mode == cd.AUTOTUNE_GRID_TRANSPOSE and o.dtype == cd.DOUBLE
    assert o.allow_uneven_decompositions and not o.disable_mpi_backends and not o.disable_nccl_backends
    assert not o.disable_nvshmem_backends and o.skip_threshold == 0.0
    assert not o.autotune_transpose_backend and not o.autotune_halo_backend and o.halo_axis == 0
    assert list(o.transpose_op_weights) == [1.0] * 4 and not any(o.transpose_use_inplace_buffers)
    for tbl in (o.transpose_input_halo_extents, o.transpose_output_halo_extents, o.transpose_input_padding,
                o.transpose_output_padding):
        assert all(v == 0 for row in tbl for v in row)
    assert list(o.halo_extents) == [0, 0, 0] and not any(o.halo_periods) and list(o.halo_padding) == [0, 0, 0]


def test_dtype_sizes_and_backend_strings(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "api_constants.json")))
    enum = {"CUDECOMP_FLOAT": cd.FLOAT, "CUDECOMP_DOUBLE": cd.DOUBLE, "CUDECOMP_FLOAT_COMPLEX": cd.FLOAT_COMPLEX,
            "CUDECOMP_DOUBLE_COMPLEX": cd.DOUBLE_COMPLEX}
    for name, size in gold["dtype_sizes"].items():
        assert cd.cudecompGetDataTypeSize(enum[name]) == size
    for name, text in gold["transpose_backend_strings"].items():
        assert cd.cudecompTransposeCommBackendToString(getattr(cd, name[len("CUDECOMP_"):])) == text
    for name, text in gold["halo_backend_strings"].items():
        assert cd.cudecompHaloCommBackendToString(getattr(cd, name[len("CUDECOMP_"):])) == text
    assert cd.cudecompTransposeCommBackendToString(999) == "ERROR"
    assert cd.cudecompHaloCommBackendToString(999) == "ERROR"
    L = cd.lib()
    n = C.c_int64()
    assert L.cudecompGetDataTypeSize(cd.FLOAT, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetDataTypeSize(999, C.byref(n)) == cd.RESULT_INVALID_USAGE


def test_setdefaults_and_init_reject_bad_arguments(capfd):
    L = cd.lib()
    assert L.cudecompGridDescConfigSetDefaultsVersioned(None, 104, 1) == cd.RESULT_INVALID_USAGE
    c = cd.GridDescConfig()
    assert L.cudecompGridDescConfigSetDefaultsVersioned(C.byref(c), 103, 1) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGridDescConfigSetDefaultsVersioned(C.byref(c), 104, 2) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGridDescAutotuneOptionsSetDefaultsVersioned(None, 320, 1) == cd.RESULT_INVALID_USAGE
    assert L.cudecompInit(None, cd.MPI_COMM_WORLD) == cd.RESULT_INVALID_USAGE
    assert L.cudecompFinalize(None) == cd.RESULT_INVALID_USAGE
    err = capfd.readouterr().err
    assert "CUDECOMP:ERROR:" in err and "Invalid usage." in err  # reference message convention


@pytest.fixture()
def handle():
    h = cd.cudecompInit()
    yield h
    cd.cudecompFinalize(h)


def test_grid_desc_create_validation(handle):
    # reference tests/ctest/api_tests.cc (GridDescCreate suites): every malformed config is INVALID_USAGE
    L = cd.lib()
    gd = C.c_void_p()

    def create(cfg, opt=None, size=104, ver=1):
        return L.cudecompGridDescCreateVersioned(handle, C.byref(gd), C.byref(cfg), size, ver,
                                                 C.byref(opt) if opt is not None else None,
                                                 320 if opt is not None else 0, 1 if opt is not None else 0)

    good = cd.make_config((8, 8, 8), (1, 1))
    assert L.cudecompGridDescCreateVersioned(handle, None, C.byref(good), 104, 1, None, 0, 0) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGridDescCreateVersioned(handle, C.byref(gd), None, 104, 1, None, 0, 0) == cd.RESULT_INVALID_USAGE
    assert create(good, size=103) == cd.RESULT_INVALID_USAGE
    assert create(good, ver=2) == cd.RESULT_INVALID_USAGE
    raw = cd.GridDescConfig()  # never initialised: wrong magic
    assert create(raw) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (2, 1))  # product != nranks
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (-1, -1))
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (0, 0))  # autotune pdims needs options
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (1, 1), transpose_backend=99)
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (1, 1), halo_backend=99)
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (1, 1), rank_order=7)
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (1, 1), gdims_dist=(9, 8, 8))
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (1, 1))
    bad.transpose_mem_order[0][0] = 0  # partially set
    assert create(bad) == cd.RESULT_INVALID_USAGE
    bad = cd.make_config((8, 8, 8), (1, 1), mem_order=((0, 1, 2), (0, 0, 2), (0, 1, 2)))
    assert create(bad) == cd.RESULT_INVALID_USAGE
    assert create(good) == cd.RESULT_SUCCESS
    # create reports "default" fields back as defaults (in/out config)
    assert list(good.gdims_dist) == [0, 0, 0] and good.transpose_mem_order[0][0] == -1
    assert good.rank_order == cd.RANK_ORDER_ROW_MAJOR  # DEFAULT resolved
    other = cd.cudecompInit()
    assert L.cudecompGridDescDestroy(other, gd) == cd.RESULT_INVALID_USAGE  # belongs to another handle
    assert L.cudecompGetGridDescConfigVersioned(other, gd, C.byref(cd.GridDescConfig()), 104, 1) == cd.RESULT_INVALID_USAGE
    cd.cudecompFinalize(other)
    cd.cudecompGridDescDestroy(handle, gd)
    assert L.cudecompGridDescDestroy(handle, None) == cd.RESULT_INVALID_USAGE


def test_query_argument_checks(handle):
    L = cd.lib()
    gd = cd.cudecompGridDescCreate(handle, cd.make_config((9, 10, 11), (1, 1)))
    p = cd.PencilInfo()
    i3 = (C.c_int32 * 3)
    assert L.cudecompGetPencilInfoVersioned(handle, gd, None, 96, 1, 0, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetPencilInfoVersioned(handle, gd, C.byref(p), 95, 1, 0, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetPencilInfoVersioned(handle, gd, C.byref(p), 96, 2, 0, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetPencilInfoVersioned(handle, gd, C.byref(p), 96, 1, -1, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetPencilInfoVersioned(handle, gd, C.byref(p), 96, 1, 0, i3(-1, 0, 0), None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetPencilInfoVersioned(handle, gd, C.byref(p), 96, 1, 0, None, i3(0, -1, 0)) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetPencilInfoVersioned(handle, gd, C.byref(p), 96, 1, 0, i3(2**31 - 1, 0, 0), None) == cd.RESULT_INVALID_USAGE
    n = C.c_int64()
    assert L.cudecompGetTransposeWorkspaceSize(handle, gd, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetTransposeWorkspaceSize(handle, None, C.byref(n)) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetHaloWorkspaceSize(handle, gd, 0, None, C.byref(n)) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetHaloWorkspaceSize(handle, gd, 3, i3(1, 1, 1), C.byref(n)) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetHaloWorkspaceSize(handle, gd, 0, i3(1, 1, 1), None) == cd.RESULT_INVALID_USAGE
    r = C.c_int32()
    assert L.cudecompGetShiftedRank(handle, gd, 3, 0, 1, False, C.byref(r)) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetShiftedRank(handle, gd, 0, 3, 1, False, C.byref(r)) == cd.RESULT_INVALID_USAGE
    assert L.cudecompGetShiftedRank(handle, gd, 0, 1, 1, False, None) == cd.RESULT_INVALID_USAGE
    # transposes / halos: argument validation happens before any device work
    assert L.cudecompTransposeXToY(handle, gd, None, 1, 1, cd.FLOAT, None, None, None, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompTransposeXToY(handle, gd, 1, None, 1, cd.FLOAT, None, None, None, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompTransposeXToY(handle, gd, 1, 1, None, cd.FLOAT, None, None, None, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompTransposeYToZ(handle, gd, 1, 1, 1, 5, None, None, None, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompUpdateHalosX(handle, gd, 1, 1, cd.FLOAT, None, None, 0, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompUpdateHalosX(handle, gd, None, 1, cd.FLOAT, i3(1, 1, 1), None, 0, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompUpdateHalosX(handle, gd, 1, 1, cd.FLOAT, i3(1, 1, 1), None, 3, None, None) == cd.RESULT_INVALID_USAGE
    assert L.cudecompUpdateHalosX(handle, gd, None, None, cd.FLOAT, i3(0, 0, 0), None, 0, None, None) == cd.RESULT_SUCCESS
    buf = C.c_void_p()
    assert L.cudecompMalloc(handle, gd, None, 16) == cd.RESULT_INVALID_USAGE
    assert L.cudecompMalloc(handle, gd, C.byref(buf), 0) == cd.RESULT_INVALID_USAGE
    big = cd.cudecompGridDescCreate(handle, cd.make_config((2**31 - 1,) * 3, (1, 1)))
    assert L.cudecompGetPencilInfoVersioned(handle, big, C.byref(p), 96, 1, 0, None, None) == cd.RESULT_INVALID_USAGE
    cd.cudecompGridDescDestroy(handle, big)
    cd.cudecompGridDescDestroy(handle, gd)


def test_autotune_candidate_filters(handle, monkeypatch):
    # reference tests/ctest/api_tests.cc:319-443: environment filters are validated at grid-descriptor creation
    L = cd.lib()
    gd = C.c_void_p()
    opt = cd.cudecompGridDescAutotuneOptionsSetDefaults()
    opt.autotune_transpose_backend = True

    def create(cfg, o):
        return L.cudecompGridDescCreateVersioned(handle, C.byref(gd), C.byref(cfg), 104, 1, C.byref(o), 320, 1)

    cfg = cd.make_config((8, 8, 8), (1, 1))
    monkeypatch.setenv("CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS", "NCCL,BOGUS")
    assert create(cfg, opt) == cd.RESULT_INVALID_USAGE
    monkeypatch.setenv("CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS", "NCCL,")
    assert create(cfg, opt) == cd.RESULT_INVALID_USAGE
    monkeypatch.setenv("CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS", "MPI_P2P,MPI_P2P_PL,MPI_A2A")
    opt.disable_mpi_backends = True
    assert create(cfg, opt) == cd.RESULT_INVALID_USAGE  # nothing left after the filters
    opt.disable_mpi_backends = False
    monkeypatch.setenv("CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS",
                       "^MPI_P2P,MPI_P2P_PL,MPI_A2A,NCCL,NCCL_PL,NVSHMEM,NVSHMEM_PL,NVSHMEM_SM")
    assert create(cfg, opt) == cd.RESULT_INVALID_USAGE
    monkeypatch.delenv("CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS")
    opt.autotune_transpose_backend = False
    opt.autotune_halo_backend = True
    monkeypatch.setenv("CUDECOMP_AUTOTUNE_HALO_BACKENDS", "MPI,NOPE")
    assert create(cfg, opt) == cd.RESULT_INVALID_USAGE
    monkeypatch.delenv("CUDECOMP_AUTOTUNE_HALO_BACKENDS")
    opt.autotune_halo_backend = False
    cfg0 = cd.make_config((8, 8, 8), (0, 0))
    for bad in ("3", "2,1", "a,b", "1,2,3", "-1,4"):
        monkeypatch.setenv("CUDECOMP_AUTOTUNE_P_ROW_RANGE", bad)
        assert create(cfg0, opt) == cd.RESULT_INVALID_USAGE, bad
    monkeypatch.setenv("CUDECOMP_AUTOTUNE_P_ROW_RANGE", "2,4")  # no grid of one rank has 2..4 rows
    assert create(cfg0, opt) == cd.RESULT_INVALID_USAGE
    monkeypatch.delenv("CUDECOMP_AUTOTUNE_P_ROW_RANGE")
    opt.grid_mode = 7
    assert create(cfg0, opt) in (cd.RESULT_INVALID_USAGE, cd.RESULT_CUDA_ERROR)


def test_empty_pencils_and_wide_halos_are_rejected_by_the_planner(handle):
    # reference include/internal/transpose.h:257-259, halo.h:57-59 (NOT_SUPPORTED) and halo.h:120-144 (INVALID_USAGE);
    # the planner runs before any device work, so this needs no GPU
    L = cd.lib()
    i3 = (C.c_int32 * 3)
    # one rank can never have empty pencils; a zero-extent grid is the degenerate way to get one
    gd = cd.cudecompGridDescCreate(handle, cd.make_config((0, 4, 4), (1, 1)))
    plan = cd.ExtTransposePlan()
    rc = L.cudecompExtGetTransposePlan(handle, gd, 0, None, None, None, None, False, 0, C.byref(plan))
    assert rc == cd.RESULT_NOT_SUPPORTED
    hp = cd.ExtHaloPlan()
    rc = L.cudecompExtGetHaloPlan(handle, gd, 1, i3(1, 1, 1), None, 2, None, 0, C.byref(hp))  # Y pencils split X
    assert rc == cd.RESULT_NOT_SUPPORTED
    assert L.cudecompTransposeXToY(handle, gd, 8, 8, 8, cd.FLOAT, None, None, None, None, None) == cd.RESULT_NOT_SUPPORTED
    cd.cudecompGridDescDestroy(handle, gd)
    # negative halo extents reach the planner through the transposes too
    gd = cd.cudecompGridDescCreate(handle, cd.make_config((8, 8, 8), (1, 1)))
    assert L.cudecompTransposeXToY(handle, gd, 8, 16, 8, cd.FLOAT, i3(-1, 0, 0), None, None, None, None) == cd.RESULT_INVALID_USAGE
    # single rank, in place, identical layout: nothing to do, returns before touching the device
    assert L.cudecompTransposeXToY(handle, gd, 8, 8, 8, cd.FLOAT, None, None, None, None, None) == cd.RESULT_SUCCESS
    cd.cudecompGridDescDestroy(handle, gd)


def test_multiple_live_handles_single_process():
    """api_tests.cc:575-656 (SupportsMultipleLiveHandlesWithIndependentResources,
    FinalizesMultipleHandlesInCreationOrder) in one process."""
    from tests import bodies
    out = bodies.two_handles(0, 1, {"gdims": (9, 10, 11), "pdims": (1, 1)})
    assert out["cross"] == [cd.RESULT_INVALID_USAGE] * 3 and out["unused_is_null"]
    assert out["rank_orders"] == [cd.RANK_ORDER_ROW_MAJOR, cd.RANK_ORDER_COL_MAJOR]
    assert out["pencil_row_major"] == out["pencil_col_major"]  # one rank: the order does not matter


def test_multiple_live_handles_four_ranks(golden_dir):
    """The same on a 2 x 2 job: each handle's descriptor yields ITS rank order's pencils (the reference's golden
    vectors for 9 x 10 x 11, tests/ctest/api_tests.cc:92-153, row- and column-major), descriptors are rejected by the
    other handle on every rank, finalisation in creation order leaves the second handle usable."""
    from tests.mp import run_ranks
    gold = json.load(open(os.path.join(golden_dir, "pencil_info.json")))
    args = {"gdims": gold["gdims"], "pdims": gold["pdims"], "halo": gold["halo_extents"], "padding": gold["padding"]}
    res = run_ranks(4, "tests.bodies", "two_handles", args)
    by_rank = {r["rank"]: r for r in res}
    for r in res:
        assert r["cross"] == [cd.RESULT_INVALID_USAGE] * 3 and r["unused_is_null"]
        assert r["rank_orders"] == [cd.RANK_ORDER_ROW_MAJOR, cd.RANK_ORDER_COL_MAJOR]
        assert r["after_first_finalize"] == r["pencil_row_major"][0]
    for variant in ("row_major", "col_major"):
        assert len(gold[variant]) == 12
        for g in gold[variant]:
            mine = by_rank[g["rank"]]["pencil_" + variant][g["axis"]]
            assert mine == {k: v for k, v in g.items() if k not in ("axis", "rank")}, (variant, g["axis"], g["rank"])
    # and the two rank orders really differ on the off-diagonal ranks
    assert by_rank[1]["pencil_row_major"] != by_rank[1]["pencil_col_major"]
    assert by_rank[1]["pencil_col_major"] == by_rank[2]["pencil_row_major"]


def test_autotune_prior_follows_the_plans(handle, monkeypatch):
    """The analytic prior that orders the autotuner's candidates (csrc/autotune.cc estimateTransposeCycleMs; reference
    src/autotune.cc:94-106, 675 only orders grids by factor) charges the phases the plan EXECUTES and one chunk per
    link: on a full mesh slab grids beat 2 x 4, the fused put with library buffers (direct to destination: one pass) beats
    the staged transports, the staged pipeline beats the unpipelined exchange, and elided phases are not charged."""
    monkeypatch.setenv("CUDECOMP_MODEL_XGMI_LINK_GBPS", "76.8")
    monkeypatch.setenv("CUDECOMP_MODEL_HBM_GBPS", "6000")
    contiguous = [(0, 1, 2), (1, 2, 0), (2, 0, 1)]
    default = [(0, 1, 2)] * 3

    def est(pdims, backend, orders=contiguous, lib=False, inplace=False):
        spec = cd.make_grid_spec((1024, 1024, 1024), pdims, orders)
        # a one-rank handle: the model treats communicators larger than the node as off-node, so use the NIC = link rate
        return cd.cudecompExtEstimateCycleMs(handle, spec, 8, backend, lib, inplace)

    monkeypatch.setenv("CUDECOMP_MODEL_NIC_GBPS", str(76.8 * 7))  # 7 links' worth: what a full mesh gives one GPU
    b = cd.TRANSPOSE_COMM_NVSHMEM
    assert est((1, 8), b) < est((2, 4), b) and est((8, 1), b) < est((4, 2), b)
    assert est((1, 8), cd.TRANSPOSE_COMM_NVSHMEM_SM, lib=True) < est((1, 8), cd.TRANSPOSE_COMM_NVSHMEM_SM, lib=False)
    assert est((1, 8), cd.TRANSPOSE_COMM_NVSHMEM_SM, lib=True) < est((1, 8), cd.TRANSPOSE_COMM_NVSHMEM)
    assert est((1, 8), cd.TRANSPOSE_COMM_NVSHMEM_PL) < est((1, 8), cd.TRANSPOSE_COMM_NVSHMEM)
    # default layout on 1 x 8: Y->Z needs no unpack and Z->Y no pack (transpose.h:395-402) with a two-sided transport
    assert est((1, 8), cd.TRANSPOSE_COMM_NCCL, orders=default) < est((1, 8), cd.TRANSPOSE_COMM_NCCL, orders=contiguous)
    # 1 x 1: four local permutations, no exchange; in place costs the staging pass
    assert est((1, 1), b) < est((1, 1), b, inplace=True)
    assert est((1, 1), b, orders=default, inplace=True) == 0.0  # identical layouts in place: nothing to do


def _code_objects(lib):
    """Sizes of the gfx code objects inside a library's .hip_fatbin (one clang offload bundle per translation unit)."""
    import struct
    import subprocess
    out = subprocess.run(["readelf", "-S", "-W", lib], capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if " .hip_fatbin " in l]
    assert line, "no .hip_fatbin section found"
    off, size = int(line[0].split()[4], 16), int(line[0].split()[5], 16)
    with open(lib, "rb") as f:
        f.seek(off)
        data = f.read(size)
    magic, sizes, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        p = i + 32
        for _ in range(struct.unpack_from("<Q", data, i + 24)[0]):
            _, esize, tsize = struct.unpack_from("<QQQ", data, p)
            if b"gfx" in data[p + 24:p + 24 + tsize]:
                sizes.append(esize)
            p += 24 + tsize
        pos = i + 24
    return sizes


def test_every_code_object_stays_small():
    """A measured platform limit, not a style rule (profiles/r05_code_size.md): ONE code object of 0.72 MB in the library puts
    the processes that load it into a regime where every small synchronous operation takes 14 ms (descriptor destruction 27
    ms; the multi-rank test sweeps run five times longer); the same amount of device code split over two code objects does
    not, and neither does a 0.60 MB one.  Every .hip file of the library is one code object: the kernels are spread over
    several (csrc/kernels_batch.h) and none may come near the limit."""
    for name in ("lib", "lib_tuning"):
        lib = os.path.join(ROOT, "cudecomp_amd", name, "libcudecomp.so")
        if not os.path.exists(lib):
            continue
        sizes = _code_objects(lib)
        assert len(sizes) >= 5, sizes
        assert max(sizes) < 400_000, "a code object of %s grew to %d bytes: split its translation unit" % (name, max(sizes))


def test_tuning_switches_are_ignored_loudly_by_the_default_build():
    """Tuning switches (tile walks, tile shapes, window variants) and the kernel variants they select exist in `make
    TUNING_VARIANTS=1` builds only (cudecomp_amd/lib_tuning).  The default build must not follow them silently (an A/B made with
    it would be misattributed): cudecompInit says once, on rank 0, that the switch is ignored."""
    import subprocess
    import sys
    code = "import cudecomp_amd as cd; h = cd.cudecompInit(); cd.cudecompFinalize(h)"
    env = dict(os.environ, CUDECOMP_TILE_WALK="0", CUDECOMP_WINDOW_WIDE="1", PYTHONPATH=ROOT)
    env.pop("CUDECOMP_AMD_LIBRARY", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    text = r.stdout + r.stderr
    assert "CUDECOMP:WARN: CUDECOMP_TILE_WALK is a tuning switch" in text, text[-2000:]
    assert "CUDECOMP:WARN: CUDECOMP_WINDOW_WIDE is a tuning switch" in text
    tuning = os.path.join(ROOT, "cudecomp_amd", "lib_tuning", "libcudecomp.so")
    if os.path.exists(tuning):  # the tuning build reads them and stays quiet
        r = subprocess.run([sys.executable, "-c", code], env=dict(env, CUDECOMP_AMD_LIBRARY=tuning), capture_output=True, text=True,
                           timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "tuning switch" not in r.stdout + r.stderr
