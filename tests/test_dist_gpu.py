"""RCCL path of the train step on one GPU: `TrainStep(distributed=True)` inside a world-size-1 "nccl" process group, in a
fresh child process (the group is initialised there before any other GPU call; nothing is re-exec'ed), must be
bit-identical to the non-distributed step -- eager and as a replayed hipGraph pair with the collective in between.
N > 1 is covered on CPU by tests/test_ddp_gloo.py; no multi-GPU curve has been measured on hardware."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trainstep_distributed_world1_bit_identical():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               C2S_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DIST_OK rank 0 world 1" in r.stdout
