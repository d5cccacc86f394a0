"""cudecomp_amd -- ctypes front end of libcudecomp.so, the MI355X-native drop-in for cuDecomp's C API.

This module is plumbing for harnesses (tests, bench.py): it mirrors cudecomp.h one-to-one (same names,
argument meaning and result codes) and adds nothing of its own.  All work happens in the shared library
(cudecomp_amd/csrc: C++ host code + hand-written gfx950 HIP kernels); if the library has not been built
the import fails loudly -- there is no Python / CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CUDECOMP_AMD_LIBRARY: another build of the same library, e.g. cudecomp_amd/lib_asan/libcudecomp.so for sanitizer runs)
LIB_PATH = os.environ.get("CUDECOMP_AMD_LIBRARY") or os.path.join(_HERE, "lib", "libcudecomp.so")

# ---- enums (cudecomp.h) ---------------------------------------------------------------------------
TRANSPOSE_COMM_MPI_P2P, TRANSPOSE_COMM_MPI_P2P_PL, TRANSPOSE_COMM_MPI_A2A = 1, 2, 3
TRANSPOSE_COMM_NCCL, TRANSPOSE_COMM_NCCL_PL = 4, 5
TRANSPOSE_COMM_NVSHMEM, TRANSPOSE_COMM_NVSHMEM_PL, TRANSPOSE_COMM_NVSHMEM_SM = 6, 7, 8
HALO_COMM_MPI, HALO_COMM_MPI_BLOCKING, HALO_COMM_NCCL, HALO_COMM_NVSHMEM, HALO_COMM_NVSHMEM_BLOCKING = 1, 2, 3, 4, 5
FLOAT, DOUBLE, FLOAT_COMPLEX, DOUBLE_COMPLEX = -1, -2, -3, -4
AUTOTUNE_GRID_TRANSPOSE, AUTOTUNE_GRID_HALO = 0, 1
RANK_ORDER_DEFAULT, RANK_ORDER_ROW_MAJOR, RANK_ORDER_COL_MAJOR = 0, 1, 2
(RESULT_SUCCESS, RESULT_INVALID_USAGE, RESULT_NOT_SUPPORTED, RESULT_INTERNAL_ERROR, RESULT_CUDA_ERROR,
 RESULT_CUTENSOR_ERROR, RESULT_MPI_ERROR, RESULT_NCCL_ERROR, RESULT_NVSHMEM_ERROR, RESULT_NVML_ERROR) = range(10)

GRID_DESC_CONFIG_MAGIC = 0x434F4E46
GRID_DESC_AUTOTUNE_OPTIONS_MAGIC = 0x4155544F
PENCIL_INFO_MAGIC = 0x50494E46
MPI_COMM_WORLD = 0x44000000  # cudecomp_mpi_compat.h (MPICH ABI value)
MPI_COMM_NULL = 0x04000000

DTYPE_OF_KIND = {0: FLOAT, 1: DOUBLE, 2: FLOAT_COMPLEX, 3: DOUBLE_COMPLEX}
OPS = ("XToY", "YToZ", "ZToY", "YToX")


class GridDescConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int64), ("magic", C.c_int32), ("version", C.c_int32),
                ("gdims", C.c_int32 * 3), ("gdims_dist", C.c_int32 * 3), ("pdims", C.c_int32 * 2),
                ("rank_order", C.c_int32), ("transpose_comm_backend", C.c_int32),
                ("transpose_axis_contiguous", C.c_bool * 3), ("transpose_mem_order", (C.c_int32 * 3) * 3),
                ("halo_comm_backend", C.c_int32)]


class GridDescAutotuneOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int64), ("magic", C.c_int32), ("version", C.c_int32),
                ("n_warmup_trials", C.c_int32), ("n_trials", C.c_int32), ("grid_mode", C.c_int32),
                ("dtype", C.c_int32), ("allow_uneven_decompositions", C.c_bool), ("disable_mpi_backends", C.c_bool),
                ("disable_nccl_backends", C.c_bool), ("disable_nvshmem_backends", C.c_bool),
                ("skip_threshold", C.c_double), ("autotune_transpose_backend", C.c_bool),
                ("transpose_use_inplace_buffers", C.c_bool * 4), ("transpose_op_weights", C.c_double * 4),
                ("transpose_input_halo_extents", (C.c_int32 * 3) * 4),
                ("transpose_output_halo_extents", (C.c_int32 * 3) * 4),
                ("transpose_input_padding", (C.c_int32 * 3) * 4), ("transpose_output_padding", (C.c_int32 * 3) * 4),
                ("autotune_halo_backend", C.c_bool), ("halo_extents", C.c_int32 * 3), ("halo_periods", C.c_bool * 3),
                ("halo_axis", C.c_int32), ("halo_padding", C.c_int32 * 3)]


class PencilInfo(C.Structure):
    _fields_ = [("struct_size", C.c_int64), ("magic", C.c_int32), ("version", C.c_int32),
                ("shape", C.c_int32 * 3), ("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3), ("order", C.c_int32 * 3),
                ("halo_extents", C.c_int32 * 3), ("padding", C.c_int32 * 3), ("size", C.c_int64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k != "size" else int(self.size))
                for k in ("shape", "lo", "hi", "order", "halo_extents", "padding", "size")}


assert C.sizeof(GridDescConfig) == 104 and C.sizeof(GridDescAutotuneOptions) == 320 and C.sizeof(PencilInfo) == 96

EXT_MAX_MEMBERS = 64


class ExtMove(C.Structure):
    _fields_ = [("src_buf", C.c_int32), ("dst_buf", C.c_int32), ("src_off", C.c_int64), ("dst_off", C.c_int64),
                ("extent", C.c_int64 * 3), ("ss", C.c_int64 * 3), ("ds", C.c_int64 * 3), ("peer", C.c_int32),
                ("row_pitch", C.c_int32)]


class ExtTransposePlan(C.Structure):
    _fields_ = [("noop", C.c_int32), ("exchange", C.c_int32), ("comm_axis", C.c_int32), ("nranks", C.c_int32),
                ("comm_rank", C.c_int32), ("send_buf", C.c_int32), ("recv_buf", C.c_int32), ("n_pack", C.c_int32),
                ("n_unpack", C.c_int32), ("rotate", C.c_int32), ("send_base", C.c_int64), ("recv_base", C.c_int64),
                ("send_cnt", C.c_int64 * EXT_MAX_MEMBERS), ("send_off", C.c_int64 * EXT_MAX_MEMBERS),
                ("recv_cnt", C.c_int64 * EXT_MAX_MEMBERS), ("recv_off", C.c_int64 * EXT_MAX_MEMBERS),
                ("remote_recv_off", C.c_int64 * EXT_MAX_MEMBERS),
                ("member_global_rank", C.c_int32 * EXT_MAX_MEMBERS), ("schedule_dst", C.c_int32 * EXT_MAX_MEMBERS),
                ("pack", ExtMove * EXT_MAX_MEMBERS), ("unpack", ExtMove * EXT_MAX_MEMBERS),
                ("n_direct", C.c_int32), ("reserved2", C.c_int32), ("direct", ExtMove * EXT_MAX_MEMBERS),
                ("stage_axis", C.c_int32), ("reserved3", C.c_int32), ("stage_limit", C.c_int64),
                ("send_n", C.c_int64 * EXT_MAX_MEMBERS), ("recv_n", C.c_int64 * EXT_MAX_MEMBERS)]


class ExtHaloPlan(C.Structure):
    _fields_ = [("kind", C.c_int32), ("comm_axis", C.c_int32), ("neighbor", C.c_int32 * 2), ("xbuf", C.c_int32),
                ("n_pre", C.c_int32), ("n_post", C.c_int32), ("reserved", C.c_int32), ("face_elements", C.c_int64),
                ("send_off", C.c_int64 * 2), ("recv_off", C.c_int64 * 2), ("pre", ExtMove * 2), ("post", ExtMove * 2)]


class ExtRelayMove(C.Structure):
    _fields_ = [("dst_rank", C.c_int32), ("to_relay", C.c_int32), ("src_off", C.c_int64), ("dst_off", C.c_int64),
                ("count", C.c_int64)]


EXT_MAX_RELAY_MOVES = 2 * 64 * 4


class ExtRelayPlan(C.Structure):
    _fields_ = [("applies", C.c_int32), ("nranks", C.c_int32), ("slots_per_source", C.c_int32), ("n_scatter", C.c_int32),
                ("n_forward", C.c_int32), ("reserved", C.c_int32), ("slot_elements", C.c_int64), ("relay_elements", C.c_int64),
                ("scatter", ExtRelayMove * EXT_MAX_RELAY_MOVES), ("forward", ExtRelayMove * EXT_MAX_RELAY_MOVES)]


class ExtCounters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("graphs_captured", "graph_launches", "local", "rccl", "mpi", "peer_barrier",
                                          "peer_fused", "peer_pipelined", "direct_puts", "workspace_pool_hits",
                                          "stale_ipc_mappings", "workspace_pool_bytes", "retired_imports", "compute_queues_on_device",
                                          "hardware_queue_slots", "relayed", "rotations")]


class ExtLinkInfo(C.Structure):
    _fields_ = [("gbps_sdma", C.c_double), ("gbps_cu", C.c_double), ("measured", C.c_int32),
                ("crosses_devices", C.c_int32), ("copy_engine", C.c_int32), ("reserved", C.c_int32)]


class ExtGridSpec(C.Structure):
    _fields_ = [("gdims", C.c_int32 * 3), ("gdims_dist", C.c_int32 * 3), ("pdims", C.c_int32 * 2),
                ("col_major", C.c_int32), ("mem_order", (C.c_int32 * 3) * 3)]


# every symbol include/cudecomp.h and include/cudecomp_ext.h declare (checked by tests/test_abi.py)
API_SYMBOLS = [
    "cudecompInit", "cudecompInit_F", "cudecompFinalize", "cudecompGridDescCreateVersioned",
    "cudecompGridDescDestroy", "cudecompGridDescConfigSetDefaultsVersioned",
    "cudecompGridDescAutotuneOptionsSetDefaultsVersioned", "cudecompGetGridDescConfigVersioned",
    "cudecompGetPencilInfoVersioned", "cudecompGetTransposeWorkspaceSize", "cudecompGetHaloWorkspaceSize",
    "cudecompGetDataTypeSize", "cudecompGetShiftedRank", "cudecompTransposeCommBackendToString",
    "cudecompHaloCommBackendToString", "cudecompMalloc", "cudecompFree", "cudecompTransposeXToY",
    "cudecompTransposeYToZ", "cudecompTransposeZToY", "cudecompTransposeYToX", "cudecompUpdateHalosX",
    "cudecompUpdateHalosY", "cudecompUpdateHalosZ",
]
EXT_SYMBOLS = ["cudecompExtGetTransposePlan", "cudecompExtGetHaloPlan", "cudecompExtMove3D",
               "cudecompExtGetTransposeTimings", "cudecompExtGetHaloTimings", "cudecompExtPeerProbe", "cudecompExtGetCounters",
               "cudecompExtPlanTranspose", "cudecompExtPlanHalo", "cudecompExtPencilInfo", "cudecompExtShiftedRank",
               "cudecompExtWorkspaceSizes", "cudecompExtGetLinkInfo", "cudecompExtLastKernelName",
               "cudecompExtRunLocalPhases", "cudecompExtEstimateCycleMs", "cudecompExtTrimWorkspacePool", "cudecompExtPlanRelay", "cudecompExtQueueCensus",
               "cudecompExtDescribeMove", "cudecompExtRotateWalk"]


class ExtTransposeTimings(C.Structure):
    _fields_ = [("calls", C.c_int64), ("samples", C.c_int64), ("total_ms", C.c_double), ("pack_ms", C.c_double),
                ("exchange_ms", C.c_double), ("unpack_ms", C.c_double), ("pencil_bytes", C.c_int64)]


class CudecompError(RuntimeError):
    def __init__(self, code, where):
        super().__init__("%s failed with cudecompResult_t %d" % (where, code))
        self.code = code


_lib = None


def lib():
    """The loaded libcudecomp.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libcudecomp.so is missing (%s): build it with `make -C cudecomp_amd` or "
                              "__graft_entry__.build(); there is no fallback path" % LIB_PATH)
        # torch ships its own copy of the HIP runtime under a different file name; whichever runtime is loaded
        # first must serve both, so bring torch's in before our DT_NEEDED libamdhip64.so.7 gets resolved
        # (two runtimes in one process corrupt the heap at exit).  A C/C++ solver never hits this.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        pi32 = C.POINTER(C.c_int32)
        L.cudecompInit.argtypes = [C.POINTER(vp), C.c_int]
        L.cudecompInit_F.argtypes = [C.POINTER(vp), C.c_int]
        L.cudecompFinalize.argtypes = [vp]
        L.cudecompGridDescCreateVersioned.argtypes = [vp, C.POINTER(vp), C.POINTER(GridDescConfig), i64, i32,
                                                      C.POINTER(GridDescAutotuneOptions), i64, i32]
        L.cudecompGridDescDestroy.argtypes = [vp, vp]
        L.cudecompGridDescConfigSetDefaultsVersioned.argtypes = [C.POINTER(GridDescConfig), i64, i32]
        L.cudecompGridDescAutotuneOptionsSetDefaultsVersioned.argtypes = [C.POINTER(GridDescAutotuneOptions), i64, i32]
        L.cudecompGetGridDescConfigVersioned.argtypes = [vp, vp, C.POINTER(GridDescConfig), i64, i32]
        L.cudecompGetPencilInfoVersioned.argtypes = [vp, vp, C.POINTER(PencilInfo), i64, i32, i32, pi32, pi32]
        L.cudecompGetTransposeWorkspaceSize.argtypes = [vp, vp, C.POINTER(i64)]
        L.cudecompGetHaloWorkspaceSize.argtypes = [vp, vp, i32, pi32, C.POINTER(i64)]
        L.cudecompGetDataTypeSize.argtypes = [i32, C.POINTER(i64)]
        L.cudecompGetShiftedRank.argtypes = [vp, vp, i32, i32, i32, C.c_bool, pi32]
        L.cudecompTransposeCommBackendToString.argtypes = [i32]
        L.cudecompTransposeCommBackendToString.restype = C.c_char_p
        L.cudecompHaloCommBackendToString.argtypes = [i32]
        L.cudecompHaloCommBackendToString.restype = C.c_char_p
        L.cudecompMalloc.argtypes = [vp, vp, C.POINTER(vp), C.c_size_t]
        L.cudecompFree.argtypes = [vp, vp, vp]
        for name in ("cudecompTransposeXToY", "cudecompTransposeYToZ", "cudecompTransposeZToY",
                     "cudecompTransposeYToX"):
            getattr(L, name).argtypes = [vp, vp, vp, vp, vp, i32, pi32, pi32, pi32, pi32, vp]
        for name in ("cudecompUpdateHalosX", "cudecompUpdateHalosY", "cudecompUpdateHalosZ"):
            getattr(L, name).argtypes = [vp, vp, vp, vp, i32, pi32, C.POINTER(C.c_bool), i32, pi32, vp]
        L.cudecompExtGetTransposePlan.argtypes = [vp, vp, i32, pi32, pi32, pi32, pi32, C.c_bool, i32,
                                                  C.POINTER(ExtTransposePlan)]
        L.cudecompExtGetHaloPlan.argtypes = [vp, vp, i32, pi32, C.POINTER(C.c_bool), i32, pi32, i32,
                                             C.POINTER(ExtHaloPlan)]
        L.cudecompExtGetTransposeTimings.argtypes = [vp, vp, i32, C.POINTER(ExtTransposeTimings)]
        L.cudecompExtGetHaloTimings.argtypes = [vp, vp, i32, i32, C.POINTER(ExtTransposeTimings)]
        L.cudecompExtPeerProbe.argtypes = [vp, vp, C.c_size_t, pi32]
        L.cudecompExtGetCounters.argtypes = [vp, vp, C.POINTER(ExtCounters)]
        L.cudecompExtTrimWorkspacePool.argtypes = [vp]
        L.cudecompExtQueueCensus.argtypes = [vp, pi32, pi32]
        L.cudecompExtPlanTranspose.argtypes = [C.POINTER(ExtGridSpec), i32, i32, pi32, pi32, pi32, pi32, C.c_bool, i32,
                                               i32, i32, C.POINTER(ExtTransposePlan)]
        L.cudecompExtPencilInfo.argtypes = [C.POINTER(ExtGridSpec), i32, i32, pi32, pi32, C.POINTER(PencilInfo)]
        L.cudecompExtShiftedRank.argtypes = [C.POINTER(ExtGridSpec), i32, i32, i32, i32, C.c_bool, pi32]
        L.cudecompExtWorkspaceSizes.argtypes = [C.POINTER(ExtGridSpec), i32, i32, pi32, C.POINTER(C.c_int64),
                                                C.POINTER(C.c_int64)]
        L.cudecompExtPlanRelay.argtypes = [C.POINTER(ExtGridSpec), i32, i32, pi32, pi32, pi32, pi32, C.c_bool,
                                           C.POINTER(ExtRelayPlan)]
        L.cudecompExtPlanHalo.argtypes = [C.POINTER(ExtGridSpec), i32, i32, pi32, C.POINTER(C.c_bool), i32, pi32, i32,
                                          C.POINTER(ExtHaloPlan)]
        L.cudecompExtGetLinkInfo.argtypes = [vp, C.POINTER(ExtLinkInfo)]
        L.cudecompExtEstimateCycleMs.argtypes = [vp, C.POINTER(ExtGridSpec), i32, i32, i32, i32, C.POINTER(C.c_double)]
        L.cudecompExtRunLocalPhases.argtypes = [C.POINTER(ExtGridSpec), i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]
        L.cudecompExtLastKernelName.argtypes = []
        L.cudecompExtLastKernelName.restype = C.c_char_p
        L.cudecompExtMove3D.argtypes = [vp, vp, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), i32, pi32, vp]
        L.cudecompExtDescribeMove.argtypes = [C.c_uint64, C.c_uint64, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), i32,
                                              C.POINTER(i64)]
        L.cudecompExtRotateWalk.argtypes = [i32, i32, i64, i64, pi32, C.POINTER(i64)]
        _lib = L
    return _lib


def _check(code, where):
    if code != RESULT_SUCCESS:
        raise CudecompError(code, where)


def _i3(v):
    return None if v is None else (C.c_int32 * 3)(*[int(x) for x in v])


def _b3(v):
    return None if v is None else (C.c_bool * 3)(*[bool(x) for x in v])


# ---- thin wrappers with the reference's names -------------------------------------------------------
def cudecompInit(comm=MPI_COMM_WORLD):
    h = C.c_void_p()
    _check(lib().cudecompInit(C.byref(h), comm), "cudecompInit")
    return h


def cudecompFinalize(handle):
    _check(lib().cudecompFinalize(handle), "cudecompFinalize")


def cudecompGridDescConfigSetDefaults():
    c = GridDescConfig()
    _check(lib().cudecompGridDescConfigSetDefaultsVersioned(C.byref(c), C.sizeof(c), 1),
           "cudecompGridDescConfigSetDefaults")
    return c


def cudecompGridDescAutotuneOptionsSetDefaults():
    o = GridDescAutotuneOptions()
    _check(lib().cudecompGridDescAutotuneOptionsSetDefaultsVersioned(C.byref(o), C.sizeof(o), 1),
           "cudecompGridDescAutotuneOptionsSetDefaults")
    return o


def cudecompGridDescCreate(handle, config, options=None):
    gd = C.c_void_p()
    _check(lib().cudecompGridDescCreateVersioned(handle, C.byref(gd), C.byref(config), C.sizeof(config), 1,
                                                 C.byref(options) if options is not None else None,
                                                 C.sizeof(options) if options is not None else 0,
                                                 1 if options is not None else 0), "cudecompGridDescCreate")
    return gd


def cudecompGridDescDestroy(handle, gd):
    _check(lib().cudecompGridDescDestroy(handle, gd), "cudecompGridDescDestroy")


def cudecompGetGridDescConfig(handle, gd):
    c = GridDescConfig()
    _check(lib().cudecompGetGridDescConfigVersioned(handle, gd, C.byref(c), C.sizeof(c), 1),
           "cudecompGetGridDescConfig")
    return c


def cudecompGetPencilInfo(handle, gd, axis, halo_extents=None, padding=None):
    p = PencilInfo()
    _check(lib().cudecompGetPencilInfoVersioned(handle, gd, C.byref(p), C.sizeof(p), 1, axis, _i3(halo_extents),
                                                _i3(padding)), "cudecompGetPencilInfo")
    return p


def cudecompGetTransposeWorkspaceSize(handle, gd):
    n = C.c_int64()
    _check(lib().cudecompGetTransposeWorkspaceSize(handle, gd, C.byref(n)), "cudecompGetTransposeWorkspaceSize")
    return n.value


def cudecompGetHaloWorkspaceSize(handle, gd, axis, halo_extents):
    n = C.c_int64()
    _check(lib().cudecompGetHaloWorkspaceSize(handle, gd, axis, _i3(halo_extents), C.byref(n)),
           "cudecompGetHaloWorkspaceSize")
    return n.value


def cudecompGetDataTypeSize(dtype):
    n = C.c_int64()
    _check(lib().cudecompGetDataTypeSize(dtype, C.byref(n)), "cudecompGetDataTypeSize")
    return n.value


def cudecompGetShiftedRank(handle, gd, axis, dim, displacement, periodic):
    r = C.c_int32(-2)
    _check(lib().cudecompGetShiftedRank(handle, gd, axis, dim, displacement, bool(periodic), C.byref(r)),
           "cudecompGetShiftedRank")
    return r.value


def cudecompTransposeCommBackendToString(b):
    return lib().cudecompTransposeCommBackendToString(b).decode()


def cudecompHaloCommBackendToString(b):
    return lib().cudecompHaloCommBackendToString(b).decode()


def cudecompMalloc(handle, gd, nbytes):
    p = C.c_void_p()
    _check(lib().cudecompMalloc(handle, gd, C.byref(p), nbytes), "cudecompMalloc")
    return p.value


def cudecompFree(handle, gd, ptr):
    _check(lib().cudecompFree(handle, gd, ptr), "cudecompFree")


def cudecompTranspose(op, handle, gd, inp, out, work, dtype, in_halo=None, out_halo=None, in_pad=None, out_pad=None,
                      stream=None):
    """op in OPS; inp/out/work are device pointers (ints)."""
    fn = getattr(lib(), "cudecompTranspose" + op)
    _check(fn(handle, gd, inp, out, work, dtype, _i3(in_halo), _i3(out_halo), _i3(in_pad), _i3(out_pad), stream),
           "cudecompTranspose" + op)


def cudecompUpdateHalos(axis, handle, gd, inp, work, dtype, halo_extents, halo_periods, dim, padding=None,
                        stream=None):
    fn = getattr(lib(), "cudecompUpdateHalos" + "XYZ"[axis])
    _check(fn(handle, gd, inp, work, dtype, _i3(halo_extents), _b3(halo_periods), dim, _i3(padding), stream),
           "cudecompUpdateHalos" + "XYZ"[axis])


def cudecompExtGetTransposePlan(handle, gd, op, in_halo=None, out_halo=None, in_pad=None, out_pad=None, inplace=False,
                                backend_override=0):
    p = ExtTransposePlan()
    _check(lib().cudecompExtGetTransposePlan(handle, gd, OPS.index(op), _i3(in_halo), _i3(out_halo), _i3(in_pad),
                                             _i3(out_pad), bool(inplace), backend_override, C.byref(p)),
           "cudecompExtGetTransposePlan")
    return p


def cudecompExtGetHaloPlan(handle, gd, axis, halo_extents, halo_periods, dim, padding=None, backend_override=0):
    p = ExtHaloPlan()
    _check(lib().cudecompExtGetHaloPlan(handle, gd, axis, _i3(halo_extents), _b3(halo_periods), dim, _i3(padding),
                                        backend_override, C.byref(p)), "cudecompExtGetHaloPlan")
    return p


def cudecompExtGetHaloTimings(handle, gd, axis, dim):
    t = ExtTransposeTimings()
    _check(lib().cudecompExtGetHaloTimings(handle, gd, axis, dim, C.byref(t)), "cudecompExtGetHaloTimings")
    return {k: getattr(t, k) for k, _ in ExtTransposeTimings._fields_}


def cudecompExtGetTransposeTimings(handle, gd, op):
    t = ExtTransposeTimings()
    _check(lib().cudecompExtGetTransposeTimings(handle, gd, OPS.index(op), C.byref(t)), "cudecompExtGetTransposeTimings")
    return {k: getattr(t, k) for k, _ in ExtTransposeTimings._fields_}


def make_grid_spec(gdims, pdims, mem_order, gdims_dist=None, col_major=False):
    g = ExtGridSpec()
    for i in range(3):
        g.gdims[i] = gdims[i]
        g.gdims_dist[i] = gdims_dist[i] if gdims_dist else 0
        for j in range(3):
            g.mem_order[i][j] = mem_order[i][j]
    g.pdims[0], g.pdims[1] = pdims
    g.col_major = 1 if col_major else 0
    return g


def cudecompExtPlanTranspose(grid, rank, op, in_halo=None, out_halo=None, in_pad=None, out_pad=None, inplace=False,
                             pipelined=False, symmetric_recv=False, npergroup=0):
    """Stateless planner (no handle, no communicator): the plan of `rank` in the decomposition `grid`."""
    p = ExtTransposePlan()
    _check(lib().cudecompExtPlanTranspose(C.byref(grid), rank, OPS.index(op), _i3(in_halo), _i3(out_halo), _i3(in_pad),
                                          _i3(out_pad), bool(inplace), int(pipelined), int(symmetric_recv), npergroup,
                                          C.byref(p)), "cudecompExtPlanTranspose")
    return p


def cudecompExtPlanRelay(grid, rank, op, in_halo=None, out_halo=None, in_pad=None, out_pad=None, inplace=False):
    """Two-hop relay moves of `rank` for transpose `op` (stateless; see cudecomp_ext.h)."""
    p = ExtRelayPlan()
    _check(lib().cudecompExtPlanRelay(C.byref(grid), rank, OPS.index(op), _i3(in_halo), _i3(out_halo), _i3(in_pad),
                                      _i3(out_pad), bool(inplace), C.byref(p)), "cudecompExtPlanRelay")
    return p


def cudecompExtPlanHalo(grid, rank, axis, halo_extents, halo_periods, dim, padding=None, force_packed=False):
    p = ExtHaloPlan()
    _check(lib().cudecompExtPlanHalo(C.byref(grid), rank, axis, _i3(halo_extents), _b3(halo_periods), dim, _i3(padding),
                                     int(force_packed), C.byref(p)), "cudecompExtPlanHalo")
    return p


def cudecompExtRunLocalPhases(grid, rank, op, phases, input, output, work, es, stream=None, pipelined=False,
                              symmetric_recv=False):
    """Pack (phases & 1) and / or unpack (phases & 2) kernels of `rank`'s transpose, launched as the executor would,
    no exchange (kernel timing at multi-GPU per-rank shapes on one GPU)."""
    _check(lib().cudecompExtRunLocalPhases(C.byref(grid), rank, OPS.index(op), int(pipelined), int(symmetric_recv),
                                           int(phases), input, output, work, es, stream),
           "cudecompExtRunLocalPhases")


def cudecompExtPencilInfo(grid, rank, axis, halo_extents=None, padding=None):
    p = PencilInfo()
    _check(lib().cudecompExtPencilInfo(C.byref(grid), rank, axis, _i3(halo_extents), _i3(padding), C.byref(p)),
           "cudecompExtPencilInfo")
    return p


def cudecompExtShiftedRank(grid, rank, axis, dim, displacement, periodic):
    out = C.c_int32(0)
    _check(lib().cudecompExtShiftedRank(C.byref(grid), rank, axis, dim, displacement, bool(periodic), C.byref(out)),
           "cudecompExtShiftedRank")
    return out.value


def cudecompExtWorkspaceSizes(grid, rank, axis, halo_extents):
    t, h = C.c_int64(0), C.c_int64(0)
    _check(lib().cudecompExtWorkspaceSizes(C.byref(grid), rank, axis, _i3(halo_extents), C.byref(t), C.byref(h)),
           "cudecompExtWorkspaceSizes")
    return t.value, h.value


def cudecompExtQueueCensus(handle):
    """(compute queues of all processes on this process's GPU, its hardware queue slots); (-1, 0) if unreadable."""
    c, s = C.c_int32(-1), C.c_int32(0)
    _check(lib().cudecompExtQueueCensus(handle, C.byref(c), C.byref(s)), "cudecompExtQueueCensus")
    return c.value, s.value


def cudecompExtTrimWorkspacePool(handle):
    """Really release what cudecompFree has parked in the workspace pool (collective)."""
    _check(lib().cudecompExtTrimWorkspacePool(handle), "cudecompExtTrimWorkspacePool")


def cudecompExtGetCounters(handle, gd):
    """dict of executor-path counters of this descriptor (see cudecompExtCounters_t)."""
    c = ExtCounters()
    _check(lib().cudecompExtGetCounters(handle, gd, C.byref(c)), "cudecompExtGetCounters")
    return {name: getattr(c, name) for name, _ in ExtCounters._fields_}


def cudecompExtLastKernelName():
    return lib().cudecompExtLastKernelName().decode()


def cudecompExtEstimateCycleMs(handle, grid_spec, es, backend, library_buffers=False, inplace=False):
    """the autotuner's analytic prior for one (grid, backend) candidate, in ms per X->Y->Z->Y->X cycle"""
    ms = C.c_double(0)
    _check(lib().cudecompExtEstimateCycleMs(handle, C.byref(grid_spec), es, backend, int(library_buffers), int(inplace),
                                            C.byref(ms)), "cudecompExtEstimateCycleMs")
    return ms.value


def cudecompExtGetLinkInfo(handle):
    """dict: one-direction copy rate to the next rank measured at start-up (see cudecompExtLinkInfo_t)."""
    i = ExtLinkInfo()
    _check(lib().cudecompExtGetLinkInfo(handle, C.byref(i)), "cudecompExtGetLinkInfo")
    return {name: getattr(i, name) for name, _ in ExtLinkInfo._fields_ if name != "reserved"}


def cudecompExtPeerProbe(handle, buffer, nbytes):
    bad = C.c_int32(-1)
    _check(lib().cudecompExtPeerProbe(handle, buffer, nbytes, C.byref(bad)), "cudecompExtPeerProbe")
    return bad.value


def cudecompExtMove3D(src, dst, es, extent, ss, ds, force_generic=False, stream=None):
    cls = C.c_int32(-1)
    a = lambda v: (C.c_int64 * 3)(*[int(x) for x in v])
    _check(lib().cudecompExtMove3D(src, dst, es, a(extent), a(ss), a(ds), int(force_generic), C.byref(cls), stream),
           "cudecompExtMove3D")
    return cls.value


def cudecompExtRotateWalk(nb, walk=-1):
    """The in-place rotation kernel's orbit walk for nb blocks per edge (no launch, no GPU): (grid, blocks) with blocks an
    int32 array of shape (grid, 3): the block triple of every workgroup, -1 -1 -1 for the ones that map to none."""
    import numpy as np
    grid = C.c_int64()
    _check(lib().cudecompExtRotateWalk(nb, walk, 0, 0, None, C.byref(grid)), "cudecompExtRotateWalk")
    blocks = np.empty((grid.value, 3), dtype=np.int32)
    _check(lib().cudecompExtRotateWalk(nb, walk, 0, grid.value, blocks.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(grid)),
           "cudecompExtRotateWalk")
    return grid.value, blocks


def cudecompExtDescribeMove(src_address, dst_address, es, extent, ss, ds, flags=0, row_pitch=0):
    """How the kernel layer would run a move (no launch, no GPU): dict of class, variant, tile, tile counts, walk, access mode.
    row_pitch: the planner's word for the move (ExtMove.row_pitch), see cudecomp_ext.h."""
    a = lambda v: (C.c_int64 * 3)(*[int(x) for x in v])
    out = (C.c_int64 * 10)()
    assert 0 <= row_pitch < (1 << 19)
    _check(lib().cudecompExtDescribeMove(int(src_address), int(dst_address), es, a(extent), a(ss), a(ds),
                                         int(flags) | (int(row_pitch) << 12), out),
           "cudecompExtDescribeMove")
    keys = ("cls", "variant", "tile_i", "tile_j", "tiles_i", "tiles_j", "batch", "run", "walk", "access")
    return dict(zip(keys, [int(x) for x in out]))


def make_config(gdims, pdims, gdims_dist=None, rank_order=0, axis_contiguous=(0, 0, 0), mem_order=None,
                transpose_backend=None, halo_backend=None):
    """Convenience: a cudecompGridDescConfig_t filled the way the reference's tests fill theirs."""
    c = cudecompGridDescConfigSetDefaults()
    for i in range(3):
        c.gdims[i] = gdims[i]
        c.gdims_dist[i] = gdims_dist[i] if gdims_dist is not None else 0
        c.transpose_axis_contiguous[i] = bool(axis_contiguous[i])
    c.pdims[0], c.pdims[1] = pdims
    c.rank_order = rank_order
    if mem_order is not None:
        for i in range(3):
            for j in range(3):
                c.transpose_mem_order[i][j] = mem_order[i][j]
    if transpose_backend is not None:
        c.transpose_comm_backend = transpose_backend
    if halo_backend is not None:
        c.halo_comm_backend = halo_backend
    return c
