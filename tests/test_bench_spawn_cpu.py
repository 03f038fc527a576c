"""`python bench.py --gpus N` must start N ranks itself (round-5 verdict: --gpus was parsed and never read, so an N-GPU
command would have timed ONE GPU).  No GPU needed: the launcher call is intercepted."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _run_main(monkeypatch, argv, env=None, devices=0):
    calls = []
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MOFA_BENCH_ONE_GPU"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: devices)
    monkeypatch.setattr(subprocess, "call", lambda cmd, **kw: calls.append((cmd, kw)) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main()
    return e.value, calls


def test_gpus_n_spawns_n_ranks(monkeypatch):
    code, calls = _run_main(monkeypatch, ["--gpus", "4", "--steps", "3", "--warmup", "1"], devices=8)
    assert code.code == 0 and len(calls) == 1
    cmd, kw = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]       # the ranks get the caller's own arguments
    assert kw["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and kw["env"]["MASTER_ADDR"] == "127.0.0.1"


def test_fewer_gpus_than_requested_is_an_error_not_a_smaller_run(monkeypatch):
    code, calls = _run_main(monkeypatch, ["--gpus", "8"], devices=1)
    assert not calls and "refusing" in str(code.code)


def test_one_gpu_functional_check_may_oversubscribe(monkeypatch):
    code, calls = _run_main(monkeypatch, ["--gpus", "2", "--backend", "gloo"], env={"MOFA_BENCH_ONE_GPU": "1"}, devices=1)
    assert code.code == 0 and len(calls) == 1 and "--nproc-per-node=2" in calls[0][0]


def test_launcher_world_size_must_match_gpus(monkeypatch):
    code, calls = _run_main(monkeypatch, ["--gpus", "8"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, devices=8)
    assert not calls and "WORLD_SIZE=2" in str(code.code)


def test_gpus_1_does_not_spawn(monkeypatch):
    """--cpu-baseline-full returns before any GPU work; here we only check that --gpus 1 takes no launcher path"""
    monkeypatch.setattr(bench, "cpu_baseline_full", lambda: {"ok": True})
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, **kw: calls.append(cmd) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--cpu-baseline-full"])
    bench.main()
    assert not calls
