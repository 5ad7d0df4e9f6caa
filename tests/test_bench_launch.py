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
    assert out["n_gpus"] == 3 and out["config"]["search_backend"] == "_NoComputeBackend"
    assert out["config"]["backend"] == "gloo" and out["config"]["world"] == 3 and len(out["config"]["devices"]) == 3


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


def test_cpu_baseline_uses_the_host_cores_under_torchruns_omp_default():
    """``torch.distributed.run`` gives each of several workers ``OMP_NUM_THREADS=1`` when the variable
    is unset; the CPU baseline must still run on, and report, the cores of the host."""
    code = (
        "import os, sys, json, numpy as np, torch\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import bench\n"
        "from oracle import oracle\n"
        "U, V, *_ = bench.make_latents('small', 600, 700)\n"
        "idx, d2 = oracle.knn_dist2(U, V, 30)\n"
        "st = type('S', (), {'idx': torch.from_numpy(idx), 'dist2': torch.from_numpy(d2)})()\n"
        "cores = bench.host_cores()\n"
        "cpu, parity = bench.cpu_baseline_and_parity(U, V, 30, st, 200, cores)\n"
        "print(json.dumps({'cores': cpu['cores'], 'affinity': len(os.sched_getaffinity(0)),\n"
        "                  'omp_threads': oracle.host_threads(), 'bad': parity['rows_with_index_mismatch'],\n"
        "                  'bits': parity['dist2_bit_identical']}))\n")
    env = {k: v for k, v in os.environ.items()}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    out = _json_line(p)
    assert out["cores"] == out["affinity"] == out["omp_threads"]
    assert out["bad"] == 0 and out["bits"] is True
    # an operator's explicit share (the GPU boxes export 16 on machines whose mask shows every core) bounds it
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print('{\"c\": %%d}' %% bench.host_cores())" % str(ROOT)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert _json_line(p)["c"] == min(2, len(os.sched_getaffinity(0)))


def test_stored_traffic_is_found_per_shard_size(tmp_path, monkeypatch):
    """A rank's launch of the sharded run (n_local rows of the workload against all references) finds
    the counter profile taken at that shard size; other sizes, kernels or sources find none."""
    sys.path.insert(0, str(ROOT))
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    kern = "fused_knn_kernel<25, 12, 320, 64, true, false>"
    for n_local, nbytes in ((1_000_000, 111), (125_000, 222)):
        (prof / f"x{n_local}_hbm_traffic.json").write_text(json.dumps(
            {"workload": "c3", "n_local": n_local, "kernel": "scmiac::" + kern,
             "source_sha256": bench.kernel_source_hash(), "hbm_bytes_per_launch": nbytes}))
    (prof / "stale_hbm_traffic.json").write_text(json.dumps(
        {"workload": "c3", "n_local": 250_000, "kernel": "scmiac::" + kern, "source_sha256": "0", "hbm_bytes_per_launch": 5}))
    real_hash = bench.kernel_source_hash()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: real_hash)
    assert bench.stored_traffic("c3", 1_000_000, 1_000_000, kern)[0] == 111
    assert bench.stored_traffic("c3", 125_000, 1_000_000, kern)[0] == 222
    assert bench.stored_traffic("c3", 250_000, 1_000_000, kern) == (None, "kernel sources changed since the stored profile stale_hbm_traffic.json was taken")
    assert bench.stored_traffic("c3", 500_000, 1_000_000, kern)[0] is None
    assert bench.stored_traffic("c3", 125_000, 1_000_000, "other")[0] is None
    assert bench.stored_traffic(None, 125_000, 1_000_000, kern)[0] is None
