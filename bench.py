#!/usr/bin/env python3
"""Anchor-search benchmark: query cells per second of the fused distance + top-k
search (one cross-modal k-NN table: N RNA query cells against M ATAC reference
cells), on synthetic Gaussian-mixture latents resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W             (one MI355X)
    python bench.py --gpus N ...            (starts N ranks itself, one per GPU, as a child torchrun)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one whole table: pack both sets, fused fp32-MFMA distance + top-k,
exact refine, direct-evaluation pass for uncertified rows.  With N > 1 ranks the
query rows are sharded (references replicated) and the per-shard tables are
all-gathered over RCCL inside the timed region; total work is fixed, so the
scaling reported is "strong".  ``--ring`` shards the reference rows as well: every
rank holds one block of them, the blocks travel round the ring of ranks and are
merged into the running top-k (the 4M x 2M, d=64 cosine atlas: ``--workload c5 --ring``).

Rank 0 prints one JSON line.  Next to the headline it carries
  roofline     -- the fused kernel against the fp32-MFMA peak: algorithmic flops
                  (2*N*M*d per table) / its average duration, timed with HIP events
                  on the launch stream inside the timed region
  cpu_baseline -- the CPU oracle (exact C/OpenMP restatement of the reference path)
                  timed on this box's host cores on a bounded sample of query rows;
                  the same rows double as the parity check of the GPU result
  full_search  -- all four tables + MNN + SNN score + label filter (one anchor search)
  pruned_exact_search -- the same table in the pruned mode (exact, not the dense contraction), and what
                  mode="auto" of the host layer picks for this workload and takes, probe included
  pcie_inclusive -- one table with inputs starting and the result ending in host memory
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import re
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from scmiac_amd import _hip, anchor_search, sharded  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz)

WORKLOADS = {
    # name: (n, m, d, mixture components, seeds (centres, rna, atac))
    "c3": (1_000_000, 1_000_000, 50, 100, (2000, 2001, 2002)),
    "c2": (100_000, 100_000, 50, 50, (1000, 1001, 1002)),
    "small": (20_000, 20_000, 50, 20, (900, 901, 902)),
    # imbalanced atlas: 200 cell types, 50 of them rare (0.1 % of the cells each), the other 150
    # sharing 95 % with weights ~ 1/rank; 10 % of the ATAC labels replaced by another type -- the
    # cell-type filter at work (the full_search object then reports what the rare types keep)
    "c4": (100_000, 100_000, 50, 200, (3000, 3001, 3002)),
    # one GPU's share of the 4M x 2M, d=64 cosine atlas on 8 GPUs: its 500k query rows against all 2M
    # reference rows (what the ring delivers block by block); rows L2-normalised, so Euclidean = cosine order
    "c5shard": (500_000, 2_000_000, 64, 200, (5000, 5001, 5002)),
    # the whole atlas (for --ring on 8 GPUs; one GPU needs about 7 s per table)
    "c5": (4_000_000, 2_000_000, 64, 200, (5000, 5001, 5002)),
    # the product's own shape: 30 canonical components, rows L2-normalised (anchors.py:877-878)
    "cca30": (1_000_000, 1_000_000, 30, 100, (3000, 3001, 3002)),
    # a wide embedding (tuning aid for the d > 64 kernel variants)
    "wide": (200_000, 500_000, 100, 100, (7000, 7001, 7002)),
    "wide128": (200_000, 500_000, 128, 100, (7100, 7101, 7102)),
    "wide200": (200_000, 500_000, 200, 100, (7200, 7201, 7202)),
    "wide256": (200_000, 500_000, 256, 100, (7300, 7301, 7302)),
}
UNIT_NORM = {"c5shard", "c5", "cca30"}
RARE_TYPES = 50  # of workload c4: labels 0..49


def make_latents(name, n=None, m=None):
    """Gaussian-mixture latents: centres ~ N(0, 9 I); RNA row = centre + N(0, I);
    ATAC row = centre + fixed modality offset (sd 0.25) + N(0, I).  float32."""
    n0, m0, d, C, (s_c, s_r, s_a) = WORKLOADS[name]
    n, m = n or n0, m or m0
    rc = np.random.default_rng(s_c)
    centres = rc.normal(scale=3.0, size=(C, d)).astype(np.float32)
    offset = rc.normal(scale=0.25, size=d).astype(np.float32)
    rr, ra = np.random.default_rng(s_r), np.random.default_rng(s_a)
    if name == "c4":
        common = 1.0 / np.arange(1, C - RARE_TYPES + 1)
        prior = np.r_[np.full(RARE_TYPES, 0.001), common / common.sum() * (1.0 - 0.001 * RARE_TYPES)]
        l1, l2 = rr.choice(C, n, p=prior), ra.choice(C, m, p=prior)
    else:
        l1, l2 = rr.integers(0, C, n), ra.integers(0, C, m)
    U = centres[l1] + rr.standard_normal((n, d), dtype=np.float32)
    V = centres[l2] + offset + ra.standard_normal((m, d), dtype=np.float32)
    if name == "c4":  # the labels the filter sees on the ATAC side are predictions: 10 % of them wrong
        flip = np.random.default_rng(3003).random(m) < 0.10
        l2 = l2.copy()
        l2[flip] = (l2[flip] + np.random.default_rng(3004).integers(1, C, int(flip.sum()))) % C
    if name in UNIT_NORM:
        U /= np.linalg.norm(U, axis=1, keepdims=True)
        V /= np.linalg.norm(V, axis=1, keepdims=True)
    return U, V, l1.astype(np.int32), l2.astype(np.int32), d


class _stdout_to_stderr:
    """Whatever native libraries print to stdout inside this block (RCCL announces its version
    there when a communicator is created) goes to stderr instead: stdout carries rank 0's one
    JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # the C library's buffered stdout as well, while it still points at stderr
        os.dup2(self._saved, 1)
        os.close(self._saved)


def spawn_ranks(n_ranks):
    """Start this script once per GPU under torch.distributed.run and return the launcher's
    exit status.  A child process, never an exec, and nothing here has touched the GPU yet."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           "--",  # what follows is the script and its own options, whatever they look like to the launcher
           str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd).returncode


class _NoComputeBackend:
    """``--dry-launch`` without an injected backend: the launch plumbing (ranks, shards, gather,
    timing, JSON) runs, the search itself is replaced by a table of reference row 0."""

    @staticmethod
    def knn_search(q, r, k, **_):
        st = _hip.KnnState(q.shape[0], k, q.device)
        st.idx.zero_()
        st.dist2.zero_()
        return st


def host_cores():
    """Host cores the CPU baseline runs on: the process's affinity mask, bounded by the cgroup's
    CPU quota where one is set.  An ``OMP_NUM_THREADS`` above 1 is the operator's stated share of a
    machine whose mask shows every core, and bounds the count too; a value of 1 is what
    ``torch.distributed.run`` gives each of several workers by default and is NOT taken for the
    size of the host."""
    cores = len(os.sched_getaffinity(0))
    try:  # cgroup v2: "<quota> <period>" or "max <period>"
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    try:
        stated = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        stated = 0
    if stated > 1:
        cores = min(cores, stated)
    return max(cores, 1)


def device_identity(dev, dry):
    """What tells this rank's GPU from every other one of the node: uuid and PCI address."""
    if dry:
        return {"device": f"cpu (rank {os.environ.get('RANK', '0')})", "id": None}
    p = torch.cuda.get_device_properties(dev)
    pci = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    uuid = str(getattr(p, "uuid", "")) or None
    return {"device": f"cuda:{dev.index} {p.name} pci {pci}", "id": f"{socket.gethostname()}/{uuid or pci}"}


def _dry_backend():
    spec = os.environ.get("SCMIAC_BENCH_DRY_BACKEND")  # "module:attribute", injected by the CPU tests
    if not spec:
        return _NoComputeBackend()
    mod, attr = spec.split(":")
    return getattr(importlib.import_module(mod), attr)()


def stored_traffic(workload, n_local, m_local, kernel_name):
    """HBM bytes per launch of the fused kernel from a stored counter profile -- only when a
    profile under ``profiles/`` was taken on this workload, this many query and reference rows per
    launch, this launch's kernel instantiation and these kernel sources; ``(None, reason)``
    otherwise.  It is never measured in the bench run itself."""
    files = sorted((ROOT / "profiles").glob("*hbm_traffic.json"))
    if not files:
        return None, "no stored counter profile"
    want_hash, reason = kernel_source_hash(), "no stored profile of this workload and shard size"
    for tfile in files:
        try:
            rec = json.loads(tfile.read_text())
        except Exception:
            continue
        if rec.get("workload") != workload or rec.get("n_local") != n_local:
            continue
        if rec.get("m_local") not in (None, m_local):
            continue
        if rec.get("kernel", "").replace("scmiac::", "") != kernel_name:
            reason = f"stored profile {tfile.name} is of {rec.get('kernel')}, this run launched {kernel_name}"
            continue
        if rec.get("source_sha256") != want_hash:
            reason = f"kernel sources changed since the stored profile {tfile.name} was taken"
            continue
        return rec.get("hbm_bytes_per_launch"), (f"stored rocprofv3 --pmc profile {tfile.relative_to(ROOT)} "
                                                 f"({rec.get('kernel')}), not measured in this run")
    return None, reason


def kernel_source_hash():
    """SHA-256 of the fused kernels' sources as the compiler sees them: comments removed and runs of
    white space collapsed, so that a reworded comment does not orphan the stored counter profiles."""
    h = hashlib.sha256()
    for name in ("knn_kernels.hip", "fused_knn.h", "fused_knn_share.h", "fused_variants.h", "fused_common.h", "kernels.h"):
        text = (ROOT / "scmiac_amd" / "csrc" / name).read_text()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
        text = re.sub(r"//[^\n]*", " ", text)                    # line comments (no string literal of these files holds "//")
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=None, help="override the number of query cells")
    ap.add_argument("--m", type=int, default=None, help="override the number of reference cells")
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU baseline sample (0 = auto: about 15 s of CPU work)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline (0 = every host core this process may use, see host_cores)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity sample")
    ap.add_argument("--no-full", action="store_true", help="skip the full four-table anchor search timing")
    ap.add_argument("--ring", action="store_true",
                    help="shard the reference rows too and pass the blocks round the ring of ranks "
                         "(sharded.knn_ring) instead of replicating them")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the N-rank launch without GPUs: gloo on the CPU, the search replaced by a "
                         "stand-in; the JSON line is marked dry_launch and carries no throughput claim")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    # `python bench.py --gpus N` with no launcher around it: become the launcher.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}: one rank per GPU is the contract")
    # SCMIAC_BENCH_FORCE_DIST=1 runs the multi-rank code path (process group, gather) with a
    # single rank: a rehearsal of the N > 1 launch on a one-GPU box
    distributed = world > 1 or args.ring or os.environ.get("SCMIAC_BENCH_FORCE_DIST") == "1"
    if distributed and world == 1:  # a process group of one: the same code path, no launcher needed
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sock:  # any free port: nobody else joins this group
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dry = args.dry_launch
    if dry:
        dev = torch.device("cpu")
        backend = _dry_backend()
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: GPU {local_rank} does not exist ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        backend = _hip
    if distributed:
        import torch.distributed as dist
        with _stdout_to_stderr():
            if dry:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()  # the communicator exists from here on
        # evidence that the ranks sit on distinct GPUs (RCCL itself refuses two ranks on one device)
        ident = device_identity(dev, dry)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        ids = [i["id"] for i in idents]
        if not dry and len(set(ids)) != world:
            raise SystemExit(f"rank {rank}: the {world} ranks do not sit on {world} distinct GPUs: {ids}")
        launch_info = {"backend": dist.get_backend(), "world": dist.get_world_size(),
                       "devices": [i["device"] for i in idents]}
    else:
        launch_info = {"backend": None, "world": 1, "devices": [device_identity(dev, dry)["device"]]}

    U, V, l1, l2, d = make_latents(args.workload, args.n, args.m)
    n, m, k = U.shape[0], V.shape[0], args.k
    # query-row shard of this rank (contiguous blocks; references replicated)
    lo, hi = sharded.shard_bounds(n, world, rank)
    q_dev = torch.from_numpy(U[lo:hi]).to(dev)
    # references: all of them on every rank, or (--ring) this rank's block of them
    r_lo, r_hi = sharded.shard_bounds(m, world, rank) if args.ring else (0, m)
    r_dev = torch.from_numpy(V[r_lo:r_hi]).to(dev)

    # With several ranks every step ends with an all-gather of the shards' tables (every rank ends
    # up with the whole table).  It is issued as soon as the search is queued and waited for one
    # step later: the table travels over xGMI while the next one is searched.  All of it has
    # arrived before the closing fence.
    in_flight = []  # [(state, pending gather)] of the step before

    def land():
        while in_flight:
            st_prev, pending = in_flight.pop()
            st_prev.gathered = pending.wait()

    def step():
        if args.ring:
            st = sharded.knn_ring(q_dev, r_dev, k, backend=backend)
        else:
            st = sharded.knn_replicated(q_dev, r_dev, k, backend=backend)
        if distributed:
            pending = sharded.gather_rows_async(st.idx, n)
            land()
            in_flight.append((st, pending))
        return st

    def fence():
        land()
        if distributed:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    if not dry:
        _hip.set_timing(True)  # HIP events around every fused-kernel launch of the timed region
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    fence()
    elapsed = time.perf_counter() - t0
    fused_total_ms, fused_launches = (0.0, 0) if dry else _hip.fused_ms()
    if not dry:
        _hip.set_timing(False)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n / (elapsed / args.steps)

    n_local = hi - lo
    # one launch of the fused kernel: this rank's queries against all references, or (--ring)
    # against one of the `world` blocks
    flops = 2.0 * n_local * m * d / (world if args.ring else 1)
    if dry:
        roofline = None
    else:
        fused_avg_ms = fused_total_ms / max(fused_launches, 1)
        achieved = flops / (fused_avg_ms * 1e-3) / 1e12
        kernel_name = _hip.knn_kernel_name(n_local, r_hi - r_lo, d, k)
        # (a stored profile describes a workload's own row counts: an overridden --n / --m has none,
        # a rank's shard of the workload's rows has one when it was profiled as `--n <shard>`)
        traffic, traffic_source = stored_traffic(args.workload if m == WORKLOADS[args.workload][1] else None,
                                                 n_local, r_hi - r_lo, kernel_name)
        roofline = {"bound": "mfma", "kernel": kernel_name, "achieved": round(achieved, 2),
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "kernel_ms": round(fused_avg_ms, 3), "flops_per_launch": flops}
        share = _hip.knn_share_info(n_local, r_hi - r_lo, d, k, device=dev, read_stats=True)
        if share["lists"] > 0:  # single-round plan: blocks out of work joined unfinished streams (last step's counters)
            roofline["stream_sharing"] = {"streams": share["streams"], "helper_lists_per_stream": share["lists"],
                                          "stages_per_run": share["run"], "streams_joined": share["joined"],
                                          "stages_scored_by_helpers": share["helper_stages"]}

    out = {
        "metric": "anchor-search query cells/sec (N=M=1M, d=50, k=30)" if args.workload == "c3" and not (args.n or args.m)
        else f"anchor-search query cells/sec (N={n}, M={m}, d={d}, k={k})",
        "value": round(value, 1), "unit": "query cells/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: one cross-modal k-NN table, N={n} query x M={m} reference "
                               f"Gaussian-mixture latents, d={d}, k={k}",
                   "sharding": ("query and reference rows sharded, reference blocks passed round the ring, all-gather of the tables" if args.ring
                                else "query rows sharded, references replicated, all-gather of the tables" if distributed else "single GPU"),
                   "uncertified_rows_last_step": int(st.n_uncertified.item()),
                   "backend": launch_info["backend"], "world": launch_info["world"], "devices": launch_info["devices"]},
        "roofline": roofline,
    }
    if dry:
        out.update(dry_launch=True, value=None, ms_per_step=None, data="none (launch rehearsal on the CPU)")
        out["config"]["search_backend"] = type(backend).__name__
        out["config"]["table_checksum"] = int(st.gathered.to(torch.int64).sum()) if distributed else int(st.idx.to(torch.int64).sum())

    if distributed:  # the gathered table is the shards' tables in rank order
        assert torch.equal(st.gathered[lo:hi], st.idx)

    if rank == 0 and not distributed and not args.no_full and not dry:
        out["pcie_inclusive"] = pcie_inclusive(U, V, k, dev)

    if rank == 0 and not args.no_cpu and not dry:
        out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(U[lo:hi], V, k, st, args.cpu_rows,
                                                                     args.cpu_threads or host_cores())

    if rank == 0 and not args.no_full and not distributed and not dry:
        out["pruned_exact_search"] = pruned_search(q_dev, r_dev, k, st)
        out["full_search"] = full_search(U, V, l1, l2, k, dev, rare=RARE_TYPES if args.workload == "c4" else 0,
                                         cpu=out.get("cpu_baseline"))

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def pcie_inclusive(U, V, k, dev):
    """One table with the inputs starting in (pinned) host memory and the index table
    ending there: H2D of Q and R + search + D2H.  Reported next to the headline, never as it."""
    u, v = torch.from_numpy(U).pin_memory(), torch.from_numpy(V).pin_memory()
    out = torch.empty((U.shape[0], k), dtype=torch.int32).pin_memory()
    best = float("inf")
    for _ in range(2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        st = _hip.knn_search(u.to(dev, non_blocking=True), v.to(dev, non_blocking=True), k)
        out.copy_(st.idx, non_blocking=True)
        torch.cuda.synchronize(dev)
        best = min(best, time.perf_counter() - t0)
    return {"ms_per_table": round(best * 1e3, 3), "cells_per_s": round(U.shape[0] / best, 1),
            "h2d_bytes": int(u.numel() + v.numel()) * 4, "d2h_bytes": int(out.numel()) * 4}


def cpu_baseline_and_parity(Q, R, k, st, rows_arg, cores):
    """Time the CPU oracle's k-NN search on a sample of query rows (all references), on ``cores``
    threads, and compare the GPU result on those rows.  (The rest of the anchor search -- mutual
    nearest neighbours, shared-neighbour counts, min_max, label filter -- is timed on the CPU by
    :func:`full_search`, which adds it to this object.)"""
    from oracle import oracle  # test infrastructure: checker and reported baseline only

    oracle.build()
    oracle.set_threads(cores)
    rng = np.random.default_rng(12345)
    n = Q.shape[0]
    if rows_arg > 0:
        nrows = min(rows_arg, n)
    else:
        probe = rng.choice(n, size=min(n, 16 * cores), replace=False)
        oracle.knn_dist2(Q[probe[:cores]], R, k, nthreads=cores)  # thread pool and caches warm
        t0 = time.perf_counter()
        oracle.knn_dist2(Q[probe], R, k, nthreads=cores)
        rate = len(probe) / max(time.perf_counter() - t0, 1e-6)
        nrows = int(min(n, max(64, rate * 15.0)))
    rows = np.sort(rng.choice(n, size=nrows, replace=False))
    t0 = time.perf_counter()
    idx_o, d2_o = oracle.knn_dist2(Q[rows], R, k, nthreads=cores)
    dt = time.perf_counter() - t0
    rows_t = torch.from_numpy(rows).to(st.idx.device)
    idx_g = st.idx[rows_t].cpu().numpy()
    d2_g = st.dist2[rows_t].cpu().numpy()
    bad_rows = int((idx_g != idx_o).any(axis=1).sum())
    cpu = {"value": round(nrows / dt, 2), "unit": "query cells/s", "cores": cores, "kind": "port",
           "sample": f"{nrows} query rows (seeded sample of this rank's shard) against all {R.shape[0]} reference rows, "
                     f"{dt:.1f} s, exact C/OpenMP oracle of the reference path on {cores} threads "
                     f"(OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}, "
                     f"{len(os.sched_getaffinity(0))} cores in the affinity mask)"}
    parity = {"rows_checked": nrows, "rows_with_index_mismatch": bad_rows,
              "dist2_bit_identical": bool(np.array_equal(d2_g, d2_o))}
    return cpu, parity


def pruned_search(q_dev, r_dev, k, dense):
    """The same table with the pruned search (``mode="pruned"``: farthest-point pivots, and every
    block streams only the cells within reach of its queries).  Exact -- compared here with the
    dense search's table bit for bit -- but not the dense contraction the headline and the roofline
    are about: reported next to them, never as them.  None where this shape does not prune."""
    n, m, d = q_dev.shape[0], r_dev.shape[0], q_dev.shape[1]
    if not _hip.knn_grid_plan(n, m, d, k, mode="pruned")["prunes"]:
        return None
    _hip.knn_search(q_dev, r_dev, k, mode="pruned")
    torch.cuda.synchronize()
    _hip.set_timing(True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        st = _hip.knn_search(q_dev, r_dev, k, mode="pruned")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fused_ms, launches = _hip.fused_ms()
    _hip.set_timing(False)
    kept, dense_stages = _hip.knn_prune_stats(n, m, d, k)
    frac = kept / max(dense_stages, 1)
    kernel_ms = fused_ms / max(launches, 1)
    # the host layer's own choice between the two modes: a probe search first, then the search
    choice = _hip.choose_mode(q_dev, r_dev, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    auto = _hip.knn_search(q_dev, r_dev, k, mode="auto")
    torch.cuda.synchronize()
    auto_dt = time.perf_counter() - t0
    auto_mode = {"picks": choice, "ms_per_table_with_probe": round(auto_dt * 1e3, 3),
                 "identical_to_dense_search": bool(torch.equal(auto.idx, dense.idx) and torch.equal(auto.dist2, dense.dist2))}
    # the contraction the kernel still performed: kept share of 2 N M d, over the kernel's time
    streamed_tflops = 2.0 * n * m * d * frac / (kernel_ms * 1e-3) / 1e12
    return {"ms_per_table": round(dt * 1e3, 3), "cells_per_s": round(n / dt, 1),
            "kernel": _hip.knn_kernel_name(n, m, d, k, mode="pruned"), "kernel_ms": round(kernel_ms, 3),
            "cells": _hip.knn_grid_plan(n, m, d, k, mode="pruned")["cells"],
            "kept_stage_fraction": round(frac, 4),
            "mfma_frac_on_streamed_stages": round(streamed_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
            "identical_to_dense_search": bool(torch.equal(st.idx, dense.idx) and torch.equal(st.dist2, dense.dist2)),
            "auto_mode": auto_mode,
            "note": "scmiac_knn_search(mode = SCMIAC_KNN_PRUNED): exact pruning by pivot cells; skips most of the dense "
                    "contraction on clustered data, so it has no roofline of its own (kernel_ms includes the blocks' "
                    "planning; ms_per_table the choice of pivots and the partition)"}


def full_search(U, V, l1, l2, k, dev, rare=0, cpu=None):
    """One whole anchor search on the device: four tables, MNN, SNN score, label filter.
    ``rare`` > 0: labels below it are the rare cell types; report what they keep.
    ``cpu``: the ``cpu_baseline`` object; the oracle's mutual-nearest-neighbour search, shared-neighbour
    counts, ``min_max`` and label filter are then run on the same four tables on the host, compared
    with the device's anchors and added to it together with an estimate of the whole search."""
    u, v = torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev)
    c1, c2 = anchor_search.label_codes(l1, l2)
    c1, c2 = torch.from_numpy(c1).to(dev), torch.from_numpy(c2).to(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    tables = anchor_search.find_neighbor_device(u, v, k)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    scores, n_pairs = anchor_search.mnn_and_score(tables, 5, min(30, k))
    kept = anchor_search.filter_anchors(scores, n_pairs, c1, c2)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    res = {"tables_s": round(t1 - t0, 4), "mnn_score_filter_s": round(t2 - t1, 4),
           "cells_per_s": round((U.shape[0] + V.shape[0]) / (t2 - t0), 1), "raw_anchors": scores.n,
           "anchors_after_filter": int(len(kept.x1)), "uncertified_rows": list(tables.uncertified)}
    if rare:
        per_type = np.bincount(l1[kept.x1][l1[kept.x1] < rare], minlength=rare)
        cells = np.bincount(l1[l1 < rare], minlength=rare)
        res["rare_types"] = {"types": int(rare), "cells_per_type_median": float(np.median(cells)),
                             "anchors_kept_total": int(per_type.sum()), "anchors_kept_per_type_min": int(per_type.min()),
                             "anchors_kept_per_type_median": float(np.median(per_type)),
                             "types_with_no_anchor": int((per_type == 0).sum())}
    if cpu is not None:
        res["anchors_identical_to_oracle_on_these_tables"] = cpu_anchor_tail(tables, scores, kept, l1, l2, k, cpu,
                                                                            U.shape[0], V.shape[0])
    return res


def cpu_anchor_tail(tables, scores, kept, l1, l2, k, cpu, n1, n2):
    """The rest of the reference path after the k-NN tables (anchors.py:569-581, 631-679 and
    scmiac.py:89-103), on the host with the oracle, on the device's four tables: timed into
    ``cpu["anchor_tail"]`` and compared with the device's anchors (returns whether all is equal)."""
    from oracle import oracle  # checker and reported baseline only

    cores = cpu["cores"]
    oracle.set_threads(cores)
    G11, G12, G21, G22 = tables.numpy()
    t0 = time.perf_counter()
    pairs = oracle.find_mnn(G12, G21, 5)
    t1 = time.perf_counter()
    raw = oracle.snn_counts(pairs, G11, G12, G21, G22, min(30, k))
    t2 = time.perf_counter()
    import pandas as pd
    df = pd.DataFrame(pairs.reshape(-1, 2), columns=["x1", "x2"])
    with np.errstate(all="ignore"):
        df["score"] = oracle.min_max(raw.astype(np.int64)) if len(raw) else np.empty(0)
    want = oracle.label_filter(df, l1, l2)
    t3 = time.perf_counter()
    same = bool(len(pairs.reshape(-1, 2)) == scores.n
                and np.array_equal(pairs.reshape(-1, 2), scores.pairs[:scores.n].cpu().numpy())
                and np.array_equal(raw, scores.raw[:scores.n].cpu().numpy())
                and np.array_equal(want["x1"].values, kept.x1) and np.array_equal(want["x2"].values, kept.x2)
                and np.array_equal(want["score"].values, kept.score))
    tail_s = t3 - t0
    tables_s = 2.0 * (n1 + n2) / cpu["value"]  # four tables: every cell is a query row twice, at the sampled rate
    cpu["anchor_tail"] = {"find_mnn_s": round(t1 - t0, 3), "snn_counts_s": round(t2 - t1, 3),
                          "min_max_label_filter_s": round(t3 - t2, 3), "total_s": round(tail_s, 3),
                          "raw_anchors": int(len(raw)), "anchors_after_filter": int(len(want)),
                          "cores": cores,
                          "note": "oracle find_mnn (serial C loop) + shared-neighbour counts (C/OpenMP) + numpy min_max + "
                                  "pandas label filter on the four full tables of full_search"}
    cpu["whole_search"] = {"estimate_s": round(tables_s + tail_s, 1), "cells_per_s": round((n1 + n2) / (tables_s + tail_s), 2),
                           "note": f"four exact tables at the sampled rate ({2 * (n1 + n2)} query rows / {cpu['value']} rows/s, "
                                   f"extrapolated) + the measured anchor tail; cells = {n1} + {n2}"}
    return same


if __name__ == "__main__":
    main()
