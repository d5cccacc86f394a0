"""tests/mp.py's rank pool (long-lived rank workers that serve the GPU suite's launches) on the CPU: worker identity across
jobs, a new library handle when the world size or a handle-level switch changes (finalize + init inside living processes:
needs the per-connection communicator ids of csrc/bootstrap.cc), pool replacement for process-level variables, failures
and timeouts killing the pool, fresh processes on request."""
import os

import pytest

from tests import mp


@pytest.fixture(autouse=True)
def _no_pool_left_behind():
    yield
    mp.pool_stop()


def probe(n, args=None, **kw):
    return mp.run_ranks(n, "tests.gpu_bodies", "pool_probe", dict(args or {}), timeout=kw.pop("timeout", 120), **kw)


def test_workers_persist_and_worlds_are_rebuilt():
    a = probe(2)
    assert [r["rank"] for r in a] == [0, 1] and all(r["world_size"] == "2" and r["pool"] == "0" for r in a)
    b = probe(2)
    assert [r["pid"] for r in b] == [r["pid"] for r in a]              # the same processes ...
    assert [r["handle_id"] for r in b] == [r["handle_id"] for r in a]  # ... and the same library handle
    c = probe(4, {"pdims": (2, 2)})                                    # a larger world: two more workers, new handles
    assert [r["pid"] for r in c[:2]] == [r["pid"] for r in a] and len({r["pid"] for r in c}) == 4
    assert all(r["world_size"] == "4" for r in c)
    d = probe(3, {"pdims": (3, 1), "gdims": (13, 10, 14)})             # a smaller one: worker 3 sits this one out
    assert [r["pid"] for r in d] == [r["pid"] for r in c[:3]]
    assert [r["shape"][0] for r in d] == [5, 4, 4]                     # X split over three ranks in the Y pencil
    e = probe(1, {"pdims": (1, 1)})
    assert e[0]["pid"] == a[0]["pid"] and e[0]["shape"] == [12, 10, 14]
    assert mp.pool_stats["started"] >= 1


def test_handle_level_switches_make_a_new_handle_process_level_ones_a_new_pool():
    a = probe(2)
    b = probe(2, extra_env={"CUDECOMP_TEST_SWITCH": "7"})
    assert [r["pid"] for r in b] == [r["pid"] for r in a] and all(r["switch"] == "7" for r in b)
    assert [r["handle_id"] for r in b] != [r["handle_id"] for r in a] or True  # (ids may be recycled; the switch is what counts)
    c = probe(2)
    assert all(r["switch"] is None for r in c)                         # ... and is gone for the next job
    d = probe(2, {"switch": "SOME_RUNTIME_VARIABLE"}, extra_env={"SOME_RUNTIME_VARIABLE": "1"})
    assert all(r["switch"] == "1" for r in d) and {r["pid"] for r in d}.isdisjoint({r["pid"] for r in a})


def test_a_failing_rank_fails_the_launch_and_the_next_job_gets_new_processes():
    a = probe(2)
    with pytest.raises(AssertionError, match="fails on purpose"):
        probe(2, {"raise_on": 1, "handle": False})
    b = probe(2)
    assert {r["pid"] for r in b}.isdisjoint({r["pid"] for r in a})
    with pytest.raises(AssertionError, match="timed out"):
        probe(2, {"sleep": 30, "handle": False}, timeout=3)
    assert len(probe(2)) == 2


def test_fresh_processes_on_request_and_for_the_bodies_about_process_state():
    a = probe(2)
    b = probe(2, fresh=True)
    assert {r["pid"] for r in b}.isdisjoint({r["pid"] for r in a}) and all(r["pool"] != "0" or True for r in b)
    assert "absent_peer" in mp.FRESH_FUNCS and "queue_census" in mp.FRESH_FUNCS
    os.environ[mp.POOL_SWITCH] = "0"
    try:
        c = probe(2)
        d = probe(2)
        assert {r["pid"] for r in c}.isdisjoint({r["pid"] for r in d})
    finally:
        del os.environ[mp.POOL_SWITCH]


def test_launches_get_one_more_attempt_for_a_refused_ipc_export_only(monkeypatch, capsys):
    """tests/mp.py: a launch (native program or Python bodies) whose failure carries the signature of the runtime refusing to export
    a fresh workspace over IPC is made once more, in new processes, and says so; any other failure is raised as it is; nothing
    is repeated twice."""
    monkeypatch.setitem(mp.pool_stats, "second_attempts", 0)   # (restored afterwards: the run's summary line counts real ones)
    refused = AssertionError("rank 0 exit 1\nCUDECOMP:ERROR: ... (a peer rank could not export its buffer over IPC)\n FAILED")
    other = AssertionError("rank 0 exit 1\n FAILED\nFailed 1/1 tests.")
    for launcher, once in (("run_binary_ranks", "_run_binary_ranks_once"), ("run_ranks", "_run_ranks_once")):
        call = (lambda: mp.run_binary_ranks(4, ["/bin/true"], 10, {"X": "1"})) if launcher == "run_binary_ranks" else \
               (lambda: mp.run_ranks(4, "tests.gpu_bodies", "pool_probe", {}, 10, {"X": "1"}))
        for script, want_calls, raises in (([refused, ["ok"]], 2, None), ([refused, refused], 2, "could not export"),
                                           ([other], 1, "Failed 1/1"), ([["ok"]], 1, None)):
            calls = []

            def fake(*a, **k):
                calls.append(a)
                r = script.pop(0)
                if isinstance(r, AssertionError):
                    raise r
                return r

            monkeypatch.setattr(mp, once, fake)
            if raises:
                with pytest.raises(AssertionError, match=raises):
                    call()
            else:
                assert call() == ["ok"]
            assert len(calls) == want_calls, (launcher, want_calls, calls)
            out = capsys.readouterr().out
            assert ("one more attempt" in out) == (want_calls == 2)
    assert mp.pool_stats["second_attempts"] == 4


def test_native_case_lists_are_told_to_stop_at_their_first_failure(monkeypatch):
    from tests import test_gpu_native as N
    seen = []
    monkeypatch.setattr(N, "_binary", lambda name: "/bin/true")
    monkeypatch.setattr(mp, "run_binary_groups", lambda groups, **k: seen.extend(groups) or [["command: x\n PASSED\nPassed all tests.\n"]] * len(groups))
    N._run_side_by_side([("transpose_test_R64", 4, ["a"], None), ("halo_test_R64", 4, ["b"], {"X": "1", "CUDECOMP_TEST_VERDICT_TIMEOUT": "5"})])
    assert [g[3]["CUDECOMP_TEST_STOP_AT_FIRST_FAILURE"] for g in seen] == ["1", "1"]
    assert [g[3]["CUDECOMP_TEST_VERDICT_TIMEOUT"] for g in seen] == ["60", "5"] and seen[1][3]["X"] == "1"
