"""Multi-process path on the CPU: world_size 2 (and 3) over gloo, launched the way
the multi-GPU runs are launched (torch.distributed.run, rendezvous on 127.0.0.1)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_over_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / "gloo_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        assert f"rank {r} ok" in p.stdout


def test_shard_bounds_cover_rows_once():
    from scmiac_amd.sharded import shard_bounds

    for n in (0, 1, 7, 1000, 1_000_000):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert max(hi - lo for lo, hi in cuts) <= (n + world - 1) // world
