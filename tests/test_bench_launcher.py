"""`python bench.py --gpus N` (N > 1) without a launcher must start its own ranks -- as fresh child processes, before
this process has made any GPU call -- with the contract's `python -m torch.distributed.run ...` command line."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_command_is_the_contracts_torchrun_line(bench):
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "c5"], port=29617)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29617"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "c5"]
    # a free port is chosen when none is given
    auto = bench.self_launch_command(2, [])
    port = int(auto[auto.index("--master-port") + 1])
    assert 1024 < port < 65536


def test_gpus_n_without_a_launcher_spawns_children_and_never_touches_the_gpu(bench, monkeypatch):
    import subprocess

    import torch
    calls = []

    def no_gpu(*a, **k):
        raise AssertionError("the launching process must not initialise the GPU")

    for name in ("is_available", "init", "set_device", "device_count", "synchronize"):
        monkeypatch.setattr(torch.cuda, name, no_gpu)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--config", "c2", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                      # the launcher's exit status is ours
    (cmd, env), = calls
    assert "--nproc-per-node=2" in cmd and cmd[-6:] == ["--gpus", "2", "--config", "c2", "--steps", "2"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_launched_rank_does_not_launch_again(bench, monkeypatch):
    """Under the launcher (RANK / WORLD_SIZE set) the self-launch branch is skipped: a world-size mismatch is an error."""
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not spawn")))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)
