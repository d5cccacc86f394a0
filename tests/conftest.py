import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "suite_order(n): position of a GPU test inside its module's slot of the suite order")
    config.addinivalue_line("markers", "extended: long arms that are not part of the default -m gpu run (CUDECOMP_TEST_EXTENDED=1 runs them; "
                                       "their logs go to profiles/)")


def pytest_sessionstart(session):
    os.environ["CUDECOMP_PYTEST_MAIN_PID"] = str(os.getpid())
    _pytest_sessionstart_build(session)


def _pytest_sessionstart_build(session):
    """A fresh checkout has no binaries (they are git-ignored): build the product library before the first test
    needs it.  hipcc cross-compiles gfx950 without a GPU; with everything built this is a no-op `make`.  The product
    itself never builds or falls back on its own -- cudecomp_amd.lib() raises if the library is missing."""
    import subprocess
    if os.environ.get("PYTEST_XDIST_WORKER"):
        return
    lib = os.path.join(ROOT, "cudecomp_amd", "lib", "libcudecomp.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j", str(max(2, os.cpu_count() or 2)), "-C", os.path.join(ROOT, "cudecomp_amd")])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- tests that use the GPU in the process that runs them ------------------------------------------------------------
# Most GPU tests only LAUNCH ranks (tests/mp.py).  The modules below also call the library from the test function
# itself.  If the pytest process did that, it would hold a GPU context for the rest of the session and be a NINTH
# process on the device whenever an 8-rank case runs: nine processes are more than one MI355X has address-space slots
# (VMIDs) for, the kernel driver then time-slices ALL processes (4-5x slower) and -- measured, DESIGN.md section 9 --
# about one case in a few thousand comes back wrong in that regime, whatever the order of the tests.  So the pytest
# process never opens the GPU: consecutive tests of these modules run in a forked child (one child per run of such
# tests, so the HIP start-up is paid once per module, not per test) whose reports are replayed here; the child is gone
# before the next rank-launching test starts.  Round 3 ordered the suite around the problem instead
# (pytest_collection_modifyitems); that hook is gone.  CUDECOMP_TEST_NO_FORK=1 runs everything in this process.
_INPROCESS_GPU_MODULES = ("test_gpu_autotune", "test_gpu_halo", "test_gpu_kernels", "test_gpu_transpose",
                          "test_gpu_dense_rows", "test_inprocess_isolation")
_LAUNCHERS = ("run_ranks(", "run_binary_ranks(")


def _inprocess(item):
    """A test of those modules that calls the library from its own function.  Tests that only LAUNCH ranks (their source
    says so) stay in the pytest process: their ranks come from the rank pool of tests/mp.py, which this process owns."""
    cached = getattr(item, "_cudecomp_inprocess", None)
    if cached is not None:
        return cached
    name = os.path.basename(str(item.fspath))
    res = any(name.startswith(m) for m in _INPROCESS_GPU_MODULES)
    if res and name != "test_inprocess_isolation.py":
        import inspect
        try:
            src = inspect.getsource(item.function)
            res = not any(w in src for w in _LAUNCHERS)
        except (OSError, TypeError, AttributeError):
            pass
    item._cudecomp_inprocess = res
    return res


# ---- order of the GPU suite -------------------------------------------------------------------------------------------
# The evidence that matters first (the driver's run has a time limit, and round 5's was cut at 66 %): the kernels against the
# oracle, the transposes / halos over the reference's case matrices, real librccl, the reference's own runner lists, the
# BASELINE configurations at full size -- every row of SURVEY.md section 8 inside the first minutes -- then the transports'
# extras, the harnesses around the path, Fortran and the MPI flavour last.  Inside a module the tests that run in the forked
# child come first, then the ones whose ranks come from the pool (one child / one pool per run of such tests).
_ORDER = ["test_gpu_kernels", "test_gpu_dense_rows", "test_gpu_transpose", "test_gpu_halo", "test_gpu_self_exchange",
          "test_gpu_runner_cases", "test_gpu_baseline_configs", "test_gpu_rccl_path", "test_gpu_autotune", "test_gpu_async",
          "test_gpu_graphs", "test_gpu_relay", "test_gpu_workspace_pool", "test_gpu_perf_report", "test_gpu_failure_detection",
          "test_gpu_queue_census", "test_gpu_fft3d", "test_gpu_c_example", "test_gpu_native", "test_gpu_native_sweep",
          "test_gpu_multi_device", "test_gpu_mpi_flavour", "test_fortran"]


def pytest_collection_modifyitems(session, config, items):
    if not os.environ.get("CUDECOMP_TEST_EXTENDED"):
        skip = pytest.mark.skip(reason="extended arm: set CUDECOMP_TEST_EXTENDED=1")
        for item in items:
            if item.get_closest_marker("extended") is not None:
                item.add_marker(skip)
    if os.environ.get("CUDECOMP_TEST_KEEP_ORDER"):
        return
    rank = {m: i for i, m in enumerate(_ORDER)}

    def key(pair):
        pos, item = pair
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if item.get_closest_marker("gpu") is None:
            return (0, 0, 0, pos)               # CPU tests: as collected
        sub = item.get_closest_marker("suite_order")
        return (1, rank.get(name, len(_ORDER)), (sub.args[0] if sub else (0 if _inprocess(item) else 1)), pos)

    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


class _Child:
    """A forked copy of the pytest process that runs the items it is told to and sends their reports back."""

    def __init__(self, session):
        import pickle
        self.pickle = pickle
        self.session = session
        c2p_r, c2p_w = os.pipe()
        p2c_r, p2c_w = os.pipe()
        sys.stdout.flush()
        sys.stderr.flush()
        from tests import mp
        mp.pool_stop()  # the child is one more process on the GPU: never beside the (up to eight) pool workers
        self.pid = os.fork()
        if self.pid == 0:
            os.environ[mp.POOL_SWITCH] = "0"  # launches from inside the child use fresh processes (it cannot own a pool)
            os.close(c2p_r)
            os.close(p2c_w)
            self._serve(os.fdopen(p2c_r, "rb"), os.fdopen(c2p_w, "wb"))
            os._exit(0)
        os.close(c2p_w)
        os.close(p2c_r)
        self.rx, self.tx = os.fdopen(c2p_r, "rb"), os.fdopen(p2c_w, "wb")

    def _serve(self, rx, tx):
        from _pytest.runner import runtestprotocol
        items = self.session.items
        try:
            while True:
                try:
                    index, next_index = self.pickle.load(rx)
                except EOFError:
                    break
                item = items[index]
                nextitem = items[next_index] if next_index is not None else None
                reports = runtestprotocol(item, nextitem=nextitem, log=False)
                data = [item.config.hook.pytest_report_to_serializable(config=item.config, report=r) for r in reports]
                self.pickle.dump(data, tx)
                tx.flush()
                if nextitem is None:
                    break
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)

    def run(self, index, next_index):
        """Reports of item `index`, or None if the child died while running it."""
        try:
            self.pickle.dump((index, next_index), self.tx)
            self.tx.flush()
            return self.pickle.load(self.rx)
        except (EOFError, BrokenPipeError, OSError):
            return None

    def close(self):
        for f in (self.tx, self.rx):
            try:
                f.close()
            except OSError:
                pass
        try:
            _, status = os.waitpid(self.pid, 0)
        except ChildProcessError:
            status = 0
        return status


_child = [None]


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_protocol(item, nextitem):
    if not _inprocess(item) or not hasattr(os, "fork") or os.environ.get("CUDECOMP_TEST_NO_FORK"):
        return None
    from _pytest.reports import TestReport
    session = item.session
    index = session.items.index(item)
    keep = nextitem is not None and _inprocess(nextitem)
    if _child[0] is None:
        _child[0] = _Child(session)
    ihook = item.ihook
    ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    data = _child[0].run(index, session.items.index(nextitem) if keep else None)
    if data is None:
        status = _child[0].close()
        _child[0] = None
        reports = [TestReport(nodeid=item.nodeid, location=item.location, keywords={}, outcome="failed", when="call",
                              longrepr="the forked GPU child died while running this test (wait status %d)" % status)]
    else:
        reports = [item.config.hook.pytest_report_from_serializable(config=item.config, data=d) for d in data]
        if not keep:
            _child[0].close()
            _child[0] = None
    for rep in reports:
        ihook.pytest_runtest_logreport(report=rep)
    ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True


# modules whose tests start GPU processes by themselves (subprocess / mpirun, not through tests/mp.py): the rank pool must be
# gone before they do (at most eight processes on the device)
_DIRECT_LAUNCHERS = ("test_gpu_fft3d", "test_gpu_c_example", "test_gpu_mpi_flavour", "test_fortran", "test_gpu_perf_report")


def pytest_runtest_setup(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _DIRECT_LAUNCHERS:
        from tests import mp
        mp.pool_stop()


def pytest_sessionfinish(session, exitstatus):
    if _child[0] is not None:
        _child[0].close()
        _child[0] = None
    from tests import mp
    mp.pool_stop()


def pytest_terminal_summary(terminalreporter):
    from tests import mp
    st = mp.pool_stats
    if st["started"] or st["fresh_launches"]:
        terminalreporter.write_line("rank launches: %d jobs on %d rank pool(s), %d launches of fresh processes"
                                    % (st["jobs"], st["started"], st["fresh_launches"]))
    if st.get("second_attempts"):  # (tests/mp.py IPC_EXPORT_REFUSED: said here because the output of passing tests is not shown)
        terminalreporter.write_line("launches repeated once because the runtime refused an IPC export: %d" % st["second_attempts"])
