"""The N > 1 branches of csrc/bdx_comm.cpp (bdx_comm_init_all, bdx_allreduce_counts_all with four ranks, the
max-all-reduce that agrees on the statistics tables' height) on a 1-GPU box: a child process binds the library to
tests/fake_rccl.cpp (BDX_RCCL_LIB), a one-process stand-in for RCCL, and drives four contexts on device 0.  The real
RCCL keeps its own tests (test_comm_gpu.py, one rank); an 8-GPU node runs the real collective in bench.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _fake_lib():
    so = os.path.join(HERE, "libbdx_fake_rccl.so")
    src = os.path.join(HERE, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-Wno-unused-value", "-o", so, src])
    return so


def test_four_contexts_through_the_collective_branches():
    env = dict(os.environ, BDX_RCCL_LIB=_fake_lib())
    r = subprocess.run([sys.executable, os.path.join(HERE, "fake_comm_driver.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ok"] and len(res["cases"]) == 2
    # the second config's length table is not fixed-height: it must have grown with the longer batch
    grown = res["cases"][1]["table_rows_before_after"]["0:len"]
    assert grown[1] > grown[0], res
