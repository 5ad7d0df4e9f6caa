"""``bench.py --gpus N`` must produce N ranks whoever starts it.  Run here on the CPU with
``--dry-launch`` (gloo instead of RCCL, the oracle standing in for the HIP library, injected
from this side of the fence: ``bench.py`` itself never names the oracle outside its CPU
baseline leg)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(OMP_NUM_THREADS="2", **(env_extra or {}))
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=300)


def _json_line(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "small", "--n", "301", "--m", "400",
            "--dry-launch"]
    out = _json_line(_run(args, {"SCMIAC_BENCH_DRY_BACKEND": "tests.gloo_worker:OracleBackend"}))
    assert out["n_gpus"] == 2 and out["dry_launch"] is True and out["steps"] == 2
    assert out["value"] is None and out["roofline"] is None      # a rehearsal claims no throughput
    assert "sharded" in out["config"]["sharding"]
    # the gathered table is the unsharded oracle's: both ranks searched their own rows
    sys.path.insert(0, str(ROOT))
    import bench
    from oracle import oracle
    U, V, *_ = bench.make_latents("small", 301, 400)
    idx, _ = oracle.knn_dist2(U, V, 30)
    assert out["config"]["table_checksum"] == int(idx.astype(np.int64).sum())


def test_ring_mode_over_two_ranks_and_with_a_single_rank():
    """``--ring``: reference rows sharded too, blocks passed round the ring; the gathered table is
    still the unsharded oracle's.  With one rank the same code path runs in a process group of one."""
    sys.path.insert(0, str(ROOT))
    import bench
    from oracle import oracle
    U, V, *_ = bench.make_latents("small", 301, 401)
    idx, _ = oracle.knn_dist2(U, V, 30)
    for gpus in ("2", "1"):
        args = ["--gpus", gpus, "--steps", "1", "--warmup", "1", "--workload", "small", "--n", "301", "--m", "401",
                "--ring", "--dry-launch"]
        out = _json_line(_run(args, {"SCMIAC_BENCH_DRY_BACKEND": "tests.gloo_worker:OracleBackend"}))
        assert out["n_gpus"] == int(gpus) and "ring" in out["config"]["sharding"]
        assert out["config"]["table_checksum"] == int(idx.astype(np.int64).sum())


def test_gpus_3_no_compute_backend():
    out = _json_line(_run(["--gpus", "3", "--steps", "1", "--warmup", "0", "--workload", "small", "--n", "100",
                           "--m", "50", "--dry-launch"]))
    assert out["n_gpus"] == 3 and out["config"]["backend"] == "_NoComputeBackend"


def test_world_size_mismatch_fails_loudly():
    # a launcher that started one rank for a two-GPU request
    p = _run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr
    # and the other way round
    p = _run(["--gpus", "1", "--dry-launch"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr


def test_single_rank_without_gpu_fails_loudly_not_silently_on_cpu():
    import torch
    if torch.cuda.is_available():
        return
    p = _run(["--gpus", "1", "--steps", "1", "--workload", "small"])
    assert p.returncode != 0 and "{" not in p.stdout
