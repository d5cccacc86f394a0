"""The isolation of tests that use the GPU in the process that runs them (tests/conftest.py): such tests run in a forked
child of the pytest process, consecutive ones share a child, module-scoped fixtures are set up and torn down inside it,
and results -- passes, failures, skips -- come back as ordinary reports.  (This module needs no GPU; it is listed among
the isolated modules so that the machinery is exercised by the CPU suite as well.)"""
import os

import pytest

_seen = []


@pytest.fixture(scope="module")
def module_state():
    state = {"pid": os.getpid(), "alive": True}
    yield state
    state["alive"] = False


def _isolated():
    return hasattr(os, "fork") and not os.environ.get("CUDECOMP_TEST_NO_FORK")


def test_runs_in_a_child_of_the_pytest_process(module_state):
    main = int(os.environ["CUDECOMP_PYTEST_MAIN_PID"])
    if _isolated():
        assert os.getpid() != main and os.getppid() == main
    assert module_state["pid"] == os.getpid()
    _seen.append(os.getpid())


def test_consecutive_tests_share_the_child(module_state):
    assert module_state["alive"]
    if os.environ.get("PYTEST_XDIST_WORKER") and not _seen:
        pytest.skip("xdist dealt the previous test of this module to another worker")
    assert _seen == [os.getpid()]


def test_skips_travel_back():
    pytest.skip("reported by the child, shown by the parent")


@pytest.mark.xfail(strict=True, reason="a failure inside the child must arrive as a failure")
def test_failures_travel_back():
    assert False
