"""The multi-rank drivers of scmiac_amd.sharded with the real kernels and more than one rank
on the one GPU a test box has: WORLD_SIZE processes share GPU 0, transport gloo (RCCL refuses
two ranks on one device).  What differs from a multi-GPU run is the transport alone."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_drivers_with_real_kernels_ranks_sharing_one_gpu(world):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "tests" / "shared_gpu_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        assert f"rank {r} ok" in p.stdout
