"""N > 1 on CPU: world_size-2 (and 4; 8 = the rank count of BASELINE configs[3]) gloo runs of the product's partition / halo-plan
code with torch.distributed as the setup exchange, checked against the oracle and the
MPI reference histories (see tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("case,size", [("hpcg", 2), ("klein", 2), ("hpcg", 4), ("hpcg", 8), ("distribute", 2), ("distribute", 3), ("distribute", 4),
                                       ("irregular", 2), ("irregular", 3)])
def test_partition_and_halo_plan_multi_process(case, size):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(size),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), case]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:]
    assert "DIST_OK %s %d" % (case, size) in text, text[-3000:]
