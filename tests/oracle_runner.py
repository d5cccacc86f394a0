"""Run transpose / halo cases through the CPU oracle for all ranks of a simulated grid and check
them against the reference's analytic (closed-form) test oracle."""
import numpy as np

from oracle import oracle as orc
from tests import cases as K


def make_grid(c):
    return orc.Grid(c["gdims"], c["pdims"], gdims_dist=c.get("gdims_dist"), rank_order=c["rank_order"],
                    axis_contiguous=c["ac"], mem_order=c["mem_order"])


def run_transpose_case(c, pipelined=False, grid=None):
    """Returns (ok, message).  One direct transpose, as tests/ctest/transpose_tests.cc:380-428."""
    g = grid or make_grid(c)
    kind = c["kind"]
    dt, es = orc.KINDS[kind]
    ax_in, ax_out = orc.OP_AXES[c["op"]]
    n = g.nranks
    pin = [g.pencil_info(r, ax_in, c["in_halo"], c["in_pad"]) for r in range(n)]
    pout = [g.pencil_info(r, ax_out, c["out_halo"], c["out_pad"]) for r in range(n)]
    wsz = g.transpose_workspace_size()
    ins, outs, works = [], [], []
    for r in range(n):
        nel = max(pin[r].size, pout[r].size)
        a = np.zeros(nel, dtype=dt)
        a[:pin[r].size] = g.fill_pencil(pin[r], kind)
        ins.append(a)
        outs.append(np.zeros(nel, dtype=dt) if c["out_of_place"] else a)
        works.append(np.zeros(wsz, dtype=dt))
    rc = g.transpose(c["op"], kind, ins, outs, works, c["in_halo"], c["out_halo"], c["in_pad"], c["out_pad"],
                     pipelined=pipelined)
    if rc != orc.OK:
        return False, "orc_transpose rc=%d" % rc
    for r in range(n):
        exp = g.fill_pencil(pout[r], kind)
        bad = orc.compare_pencil(pout[r], kind, exp, np.ascontiguousarray(outs[r][:pout[r].size]), True)
        if bad:
            return False, "rank %d mismatch at %d: exp %r got %r" % (r, bad - 1, exp[bad - 1], outs[r][bad - 1])
    return True, ""


def run_transpose_cycle(g, kind, halos, pads, out_of_place, pipelined=False):
    """X->Y->Z->Y->X chain checked after every hop (tests/cc/transpose_test.cc:516-559).
    halos/pads: per-axis (x, y, z) triples."""
    dt, es = orc.KINDS[kind]
    n = g.nranks
    pinfo = [[g.pencil_info(r, ax, halos[ax], pads[ax]) for r in range(n)] for ax in range(3)]
    wsz = g.transpose_workspace_size()
    nel = [max(pinfo[ax][r].size for ax in range(3)) for r in range(n)]
    a = [np.zeros(nel[r], dtype=dt) for r in range(n)]
    b = [np.zeros(nel[r], dtype=dt) for r in range(n)] if out_of_place else a
    works = [np.zeros(wsz, dtype=dt) for r in range(n)]
    for r in range(n):
        a[r][:pinfo[0][r].size] = g.fill_pencil(pinfo[0][r], kind)
    cur, nxt = a, b
    for op in K.OPS:
        ai, ao = orc.OP_AXES[op]
        for w in works:
            w[:] = 0
        rc = g.transpose(op, kind, cur, nxt, works, halos[ai], halos[ao], pads[ai], pads[ao], pipelined=pipelined)
        if rc != orc.OK:
            return False, "%s rc=%d" % (op, rc)
        for r in range(n):
            exp = g.fill_pencil(pinfo[ao][r], kind)
            got = np.ascontiguousarray(nxt[r][:pinfo[ao][r].size])
            bad = orc.compare_pencil(pinfo[ao][r], kind, exp, got, True)
            if bad:
                return False, "%s rank %d mismatch at %d" % (op, r, bad - 1)
        if out_of_place:
            cur, nxt = nxt, cur
    return True, ""


def run_halo_case(c, staged=False):
    """tests/ctest/halo_tests.cc:330-380: UpdateHalos for dim 0,1,2 in sequence, whole-buffer compare."""
    g = orc.Grid(c["gdims"], c["pdims"], rank_order=c["rank_order"], axis_contiguous=c["ac"],
                 mem_order=c["mem_order"])
    kind = c["kind"]
    dt, es = orc.KINDS[kind]
    n = g.nranks
    ax = c["axis"]
    pinfo = [g.pencil_info(r, ax, c["halo"], c["padding"]) for r in range(n)]
    data = [g.fill_pencil(pinfo[r], kind, halo_style=True) for r in range(n)]
    works = [np.zeros(max(1, g.halo_workspace_size(r, ax, c["halo"])), dtype=dt) for r in range(n)]
    for dim in range(3):
        rc = g.update_halos(ax, kind, data, works, c["halo"], c["periods"], dim, c["padding"], staged=staged)
        if rc != orc.OK:
            return False, "orc_update_halos rc=%d" % rc
    for r in range(n):
        exp = g.fill_halo_reference(pinfo[r], kind, c["periods"])
        bad = orc.compare_pencil(pinfo[r], kind, exp, data[r], False)
        if bad:
            return False, "rank %d mismatch at %d: exp %r got %r" % (r, bad - 1, exp[bad - 1], data[r][bad - 1])
    return True, ""
