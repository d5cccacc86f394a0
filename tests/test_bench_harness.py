"""bench.py's host-side helpers (no GPU): the autotuner log parser behind `config.also_measured`, the statistics block
and the CPU baseline record.  The log format is the reference's (src/autotune.cc:639-668), which the library reproduces
so that the reference's benchmark_runner.py can parse it too."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LOG = """CUDECOMP: Running transpose autotuning...
CUDECOMP:\tgrid: 8 x 1, backend: NCCL \nCUDECOMP:\tTotal time min/max/avg/std [ms]: 5.1/5.4/5.2/0.1
CUDECOMP:\t           min/max/avg/std [ms]: 5.0/5.3/5.15/0.1 (weighted)
CUDECOMP:\tTransposeXY time min/max/avg/std [ms]: 1.0/1.1/1.05/0.0
CUDECOMP:\txGMI-mesh model estimate [ms]: 5.543700
CUDECOMP:\tgrid: 8 x 1, backend: NVSHMEM (pipelined) \nCUDECOMP:\tTotal time min/max/avg/std [ms]: 4.1/4.4/4.2/0.1
CUDECOMP:\t           min/max/avg/std [ms]: 4.0/4.3/4.25/0.1 (weighted)
CUDECOMP:\txGMI-mesh model estimate [ms]: 4.519500
CUDECOMP:\tgrid: 4 x 2, backend: MPI_P2P \nCUDECOMP:\t(skipped) \nCUDECOMP:\tgrid: 2 x 4, backend: NVSHMEM_SM \nCUDECOMP:\t(failed, skipped) \nCUDECOMP: SELECTED: grid: 8 x 1, backend: NVSHMEM (pipelined), Avg. time (weighted) [ms]: 4.250000
"""


def test_parse_sweep_lists_every_candidate():
    got = bench.parse_sweep(LOG)
    assert [(c["pdims"], c["transport"], c["status"]) for c in got] == [
        ([8, 1], "NCCL", "measured"), ([8, 1], "NVSHMEM (pipelined)", "measured"), ([4, 2], "MPI_P2P", "skipped"),
        ([2, 4], "NVSHMEM_SM", "failed")]
    assert got[0]["avg_ms"] == 5.15 and got[0]["model_ms"] == 5.5437
    assert got[1]["avg_ms"] == 4.25 and got[1]["model_ms"] == 4.5195
    assert got[2]["avg_ms"] is None and got[3]["avg_ms"] is None
    assert bench.parse_sweep("") == []


def test_stats_block():
    s = bench.stats_of([1.0, 2.0, 3.0])
    assert s == {"min": 1.0, "max": 3.0, "avg": 2.0, "std": 0.8165, "n": 3}


def test_cpu_baseline_record_has_the_required_fields():
    rec = bench.cpu_baseline(32, 1024, "contiguous")  # a small sample: this is a format check, not a measurement
    for key in ("value", "unit", "cores", "kind", "sample",
                "cpu_model", "mpi_version", "ranks", "grid", "pinning", "host_cores"):  # provenance as keys
        assert key in rec
    assert rec["ranks"] == rec["cores"] == rec["grid"][0] * rec["grid"][1] and rec["cpu_model"]
    assert rec["kind"] == "port" and rec["unit"] == "GB/s" and rec["value"] > 0 and rec["cores"] >= 1
    if "config1_256cube_fp32_2_ranks" in rec:  # host MPI present: BASELINE config 1 at its own shape rides along
        c1 = rec["config1_256cube_fp32_2_ranks"]
        assert c1["2x1"]["round_trip_ok"] and c1["1x2"]["round_trip_ok"]
        assert set(rec["cycle_s"]) == {"avg", "min", "max", "std"}
        assert rec["mpi_version"] and rec["pinning"] == "mpirun -bind-to core" and rec["distinct_cores"] >= 1
        slots = rec["one_rank_per_gpu_slot"]  # one rank per GPU slot, config 3's 2x4 grid
        if slots is not None and "unavailable" not in slots:
            assert slots["ranks"] == 8 and slots["grid"] == [2, 4] and slots["round_trip_ok"] and slots["value"] > 0


def test_gpus_n_as_a_plain_command_launches_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus 4 --steps 2 --warmup 1` with no launcher around it must behave like the driver's
    torch.distributed.run line: N ranks on this node, rendezvous on 127.0.0.1, the ranks' ONE JSON line passed through."""
    import subprocess
    import types
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "2"], 4, 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511" and cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    seen = {}

    def fake_run(c, env=None, stdout=None, timeout=None):
        seen.update(cmd=c, env=env, timeout=timeout)
        return types.SimpleNamespace(returncode=0, stdout=b'noise\n{"n_gpus": 4, "value": 1.0}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    args = bench.parse()
    assert bench.self_launch(args) == 0
    assert capsys.readouterr().out.strip() == '{"n_gpus": 4, "value": 1.0}'
    assert seen["cmd"][-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["timeout"] == args.watchdog + 300
    # ranks that print nothing parseable: a non-zero exit code, no line
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout=b"nothing\n"))
    assert bench.self_launch(args) != 0
