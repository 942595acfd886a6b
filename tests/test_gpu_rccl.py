"""RCCL on the GPU box: the collectives of the sharded path run through the "nccl" backend with device tensors
(world size 1 -- see tests/_rccl_worker.py for why; world size 2 is covered on gloo)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_collectives_of_the_sharded_path_run_on_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    out = subprocess.run([sys.executable, worker], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RCCL_WORLD1_OK" in out.stdout
