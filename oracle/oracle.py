"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY - see cudecomp_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcudecomp_oracle.so")

OK, INVALID_USAGE, NOT_SUPPORTED = 0, 1, 2
ROW_MAJOR, COL_MAJOR = 1, 2

# op name -> (axis, dir) as in include/internal/transpose.h:907-953
OPS = {"XToY": (0, 1), "YToZ": (1, 1), "ZToY": (2, -1), "YToX": (1, -1)}
OP_AXES = {"XToY": (0, 1), "YToZ": (1, 2), "ZToY": (2, 1), "YToX": (1, 0)}

# dtype kind -> (numpy dtype, element bytes); kinds follow cudecompDataType_t order
KINDS = {0: (np.float32, 4), 1: (np.float64, 8), 2: (np.complex64, 8), 3: (np.complex128, 16)}


def build():
    src = [os.path.join(_HERE, f) for f in ("cudecomp_oracle.c", "cudecomp_oracle.h")]
    if os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src):
        return _LIB
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


class GridT(C.Structure):
    _fields_ = [("gdims", C.c_int32 * 3), ("gdims_dist", C.c_int32 * 3), ("pdims", C.c_int32 * 2),
                ("rank_order", C.c_int32), ("mem_order", (C.c_int32 * 3) * 3)]


class PInfoT(C.Structure):
    _fields_ = [("shape", C.c_int32 * 3), ("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3), ("order", C.c_int32 * 3),
                ("halo_extents", C.c_int32 * 3), ("padding", C.c_int32 * 3), ("size", C.c_int64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k != "size" else int(self.size)) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_align_count.restype = C.c_int64
        _lib.orc_align_count.argtypes = [C.c_int64, C.c_int]
        _lib.orc_transpose_workspace_size.restype = C.c_int64
        _lib.orc_compare_pencil.restype = C.c_int64
    return _lib


def _i3(v):
    if v is None:
        return None
    return (C.c_int32 * 3)(*[int(x) for x in v])


class Grid:
    """A resolved grid description plus helpers that run the oracle for every rank."""

    def __init__(self, gdims, pdims, gdims_dist=None, rank_order=0, axis_contiguous=(0, 0, 0), mem_order=None):
        self.g = GridT()
        mo = None
        if mem_order is not None:
            flat = [int(x) for row in mem_order for x in row]
            mo = (C.c_int32 * 9)(*flat)
        rc = lib().orc_grid_init(C.byref(self.g), _i3(gdims), _i3(gdims_dist), (C.c_int32 * 2)(*pdims),
                                 int(rank_order), _i3([int(bool(x)) for x in axis_contiguous]), mo)
        if rc != OK:
            raise ValueError("orc_grid_init failed: %d" % rc)
        self.gdims = [int(x) for x in gdims]
        self.pdims = [int(x) for x in pdims]
        self.nranks = self.pdims[0] * self.pdims[1]

    def pencil_info(self, rank, axis, halo=None, padding=None):
        p = PInfoT()
        rc = lib().orc_pencil_info(C.byref(self.g), rank, axis, _i3(halo), _i3(padding), C.byref(p))
        if rc != OK:
            raise ValueError("orc_pencil_info failed: %d" % rc)
        return p

    def pencil_info_rc(self, rank, axis, halo=None, padding=None):
        p = PInfoT()
        return lib().orc_pencil_info(C.byref(self.g), rank, axis, _i3(halo), _i3(padding), C.byref(p)), p

    def shifted_rank(self, rank, axis, dim, displacement, periodic):
        out = C.c_int32(-2)
        rc = lib().orc_shifted_rank(C.byref(self.g), rank, axis, dim, displacement, int(periodic), C.byref(out))
        if rc != OK:
            raise ValueError("orc_shifted_rank failed: %d" % rc)
        return out.value

    def transpose_workspace_size(self):
        return int(lib().orc_transpose_workspace_size(C.byref(self.g)))

    def halo_workspace_size(self, rank, axis, halo):
        out = C.c_int64(0)
        rc = lib().orc_halo_workspace_size(C.byref(self.g), rank, axis, _i3(halo), C.byref(out))
        if rc != OK:
            raise ValueError("orc_halo_workspace_size failed: %d" % rc)
        return out.value

    def global_rank(self, rank, comm_axis, comm_rank):
        return lib().orc_global_rank(C.byref(self.g), rank, comm_axis, comm_rank)

    def pidx(self, rank):
        out = (C.c_int32 * 2)()
        lib().orc_pidx(C.byref(self.g), rank, out)
        return [out[0], out[1]]

    # -- data movement -----------------------------------------------------------------------
    def transpose(self, op, kind, inputs, outputs, works, in_halo=None, out_halo=None, in_pad=None, out_pad=None,
                  pipelined=False):
        """inputs/outputs/works: lists (one numpy array per rank). outputs[r] is inputs[r] => in-place."""
        ax, direction = OPS[op]
        es = KINDS[kind][1]
        n = self.nranks
        pin = (C.c_void_p * n)(*[a.ctypes.data for a in inputs])
        pout = (C.c_void_p * n)(*[a.ctypes.data for a in outputs])
        pw = (C.c_void_p * n)(*[a.ctypes.data for a in works])
        return lib().orc_transpose(C.byref(self.g), ax, direction, es, pin, pout, pw, _i3(in_halo), _i3(out_halo),
                                   _i3(in_pad), _i3(out_pad), int(pipelined))

    def update_halos(self, axis, kind, data, works, halo, periods, dim, padding=None, staged=False):
        es = KINDS[kind][1]
        n = self.nranks
        pd = (C.c_void_p * n)(*[a.ctypes.data for a in data])
        pw = (C.c_void_p * n)(*[a.ctypes.data for a in works])
        per = _i3([int(bool(x)) for x in periods]) if periods is not None else None
        return lib().orc_update_halos(C.byref(self.g), axis, es, pd, pw, _i3(halo), per, dim, _i3(padding),
                                      int(staged))

    # -- the reference's analytic oracle -------------------------------------------------------
    def fill_pencil(self, pinfo, kind, halo_style=False):
        arr = np.empty(pinfo.size, dtype=KINDS[kind][0])
        lib().orc_fill_pencil(C.byref(pinfo), _i3(self.gdims), kind, int(halo_style), C.c_void_p(arr.ctypes.data))
        return arr

    def fill_halo_reference(self, pinfo, kind, periods):
        arr = np.empty(pinfo.size, dtype=KINDS[kind][0])
        lib().orc_fill_halo_reference(C.byref(pinfo), _i3(self.gdims), _i3([int(bool(x)) for x in periods]), kind,
                                      C.c_void_p(arr.ctypes.data))
        return arr


def compare_pencil(pinfo, kind, expected, actual, interior_only):
    es = KINDS[kind][1]
    assert expected.nbytes == actual.nbytes == pinfo.size * es
    return int(lib().orc_compare_pencil(C.byref(pinfo), es, C.c_void_p(expected.ctypes.data),
                                        C.c_void_p(actual.ctypes.data), int(interior_only)))


def align_count(count, nbytes=256):
    return int(lib().orc_align_count(int(count), int(nbytes)))


def get_splits(n, nchunks, pad):
    out = (C.c_int64 * nchunks)()
    lib().orc_get_splits(C.c_int64(n), nchunks, pad, out)
    return list(out)


def peer_ranks(nranks, npergroup, rank, it):
    s, d = C.c_int(0), C.c_int(0)
    lib().orc_peer_ranks(nranks, npergroup, rank, it, C.byref(s), C.byref(d))
    return s.value, d.value


def move3d_reference(src, dst, extent, ss, ds, src_off=0, dst_off=0):
    """numpy restatement of one strided block move (the K1 copy of cudecomp_kernels.cuh:125-180 and the
    cutensorPermute call of transpose.h:80-157 are both instances):
        dst[dst_off + k0*ds0 + k1*ds1 + k2*ds2] = src[src_off + k0*ss0 + k1*ss1 + k2*ss2]
    src/dst are 1-D numpy arrays of the element dtype; strides in elements."""
    es = src.itemsize
    shape = tuple(int(e) for e in extent)[::-1]
    if 0 in shape:
        return
    sv = np.lib.stride_tricks.as_strided(src[src_off:], shape=shape, strides=tuple(int(s) * es for s in ss)[::-1])
    dv = np.lib.stride_tricks.as_strided(dst[dst_off:], shape=shape, strides=tuple(int(s) * es for s in ds)[::-1])
    dv[...] = sv
