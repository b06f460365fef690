"""bench.py --gpus N outside torchrun must start its own N ranks as a child job (never re-exec, never touch
the GPU in the parent) and pass the child's exit code on."""
import os
import sys

import helpers as H


def test_self_launch_builds_a_torchrun_child(monkeypatch):
    sys.path.insert(0, H.ROOT)
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench._parse()
    assert bench._self_launch(args) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--master-addr" in cmd and "127.0.0.1" in cmd
    i = cmd.index(os.path.join(H.ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in getattr(bench, "__dict__", {}), "bench.py must not import torch at module level"


def test_workload_table_covers_the_survey_configs():
    import bench

    wl = bench.build_workload("C4", 2000, 0)
    assert wl["cfg"].is_dual and wl["algo_bytes"] == 174 and wl["outputs"] == ("bc1", "bc2", "keep_start", "keep_end")
    wl = bench.build_workload("C5", 50, 0)
    assert wl["algo_bytes"] == 212 and wl["survey_bytes"] == 10012 and len(wl["seq"]) == 50 * 10000
    wl = bench.build_workload("C2d", 1000, 0)
    assert wl["cfg"].max_error_rate == 0.2 and wl["algo_bytes"] == 162
