"""Randomised parity sweep: k-NN tables and closer-than-match counts on the GPU against
the CPU oracle over random shapes (every feature-dimension bucket of the kernels, ragged
row counts, duplicates, unit rows).  Exits non-zero on the first difference.

    python tests/checks/stress_parity.py [--cases 60] [--seed 0] [--max-n 3000] [--max-m 40000]
    python tests/checks/stress_parity.py --locality [--prune | --share] [--cases 60]     shapes large enough for the locality partition
                                  (n >= 2048, m >= 16384), random grid plans (pretend small GPUs, forced
                                  chunk counts), random partition knobs, and searches merged over reference blocks
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle  # noqa: E402  (checker)
from scmiac_amd import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-n", type=int, default=3000)
    ap.add_argument("--max-m", type=int, default=40000)
    ap.add_argument("--locality", action="store_true")
    ap.add_argument("--prune", action="store_true", help="with --locality: the pruned search (mode='pruned')")
    ap.add_argument("--share", action="store_true",
                    help="with --locality: stream sharing forced on with random (helper lists, stages per run, stages left)")
    a = ap.parse_args()
    if a.locality:
        a.max_n, a.max_m = max(a.max_n, 9000), max(a.max_m, 70000)
    rng = np.random.default_rng(a.seed)
    oracle.build()
    t0 = time.time()
    for case in range(a.cases):
        d = int(rng.choice([1, 2, 3, 7, 16, 17, 30, 31, 33, 40, 41, 50, 51, 64, 65, 80, 81, 100, 101, 128, 129, 160,
                            161, 192, 200, 224, 225, 256]))
        n = int(rng.integers(1, a.max_n))
        m = int(rng.integers(1, a.max_m))
        knobs, blocks = {}, 1
        if a.locality:
            d = int(rng.choice([2, 7, 16, 30, 31, 33, 40, 50, 51, 64, 64, 65, 80, 100, 128, 161, 200, 224, 256]))
            n, m = int(rng.integers(2048, a.max_n)), int(rng.integers(16384, a.max_m))
            if rng.random() < 0.5:
                knobs.update(slots=int(rng.choice([4, 8, 24])), tail_min_rows=int(rng.choice([512, 1024, 4096])))
            if rng.random() < 0.5:
                knobs.update(cell_rows=int(rng.choice([40, 200, 1024, 5000])), home_rows=int(rng.choice([640, 8192, 30000])),
                             seed_rows=int(rng.choice([320, 3072, 12000])))
            blocks = int(rng.choice([1, 1, 2, 3]))
        k = int(min(m, rng.integers(1, 65)))
        nc = int(rng.integers(1, 12))
        centres = rng.normal(size=(nc, d)) * rng.uniform(0.1, 4)
        r = (centres[rng.integers(0, nc, m)] + rng.normal(size=(m, d)) * rng.uniform(0.01, 2)).astype(np.float32)
        q = (centres[rng.integers(0, nc, n)] + rng.normal(size=(n, d)) * rng.uniform(0.01, 2)).astype(np.float32)
        if rng.random() < 0.3:
            r /= np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-6)
            q /= np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-6)
        if rng.random() < 0.4 and m > 20:
            r[rng.integers(0, m, 10)] = r[rng.integers(0, m)]      # duplicated references
            q[: min(n, 5)] = r[: min(n, 5)]                        # queries sitting on references
        if rng.random() < 0.2:
            r *= 1e3
            q *= 1e3
        qd, rd = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
        idx_o, d2_o = oracle.knn_dist2(q, r, k)
        _hip.set_default_mode("pruned" if a.prune else "dense")
        _hip.set_tuning(**knobs)
        share = {}
        if a.share:
            share = dict(lists=int(rng.integers(1, 5)), run=int(rng.choice([2, 3, 4, 8])), min_left=int(rng.choice([1, 2, 8])))
            _hip.set_sharing(**share)
        joined = None
        if blocks == 1:
            splits = int(rng.choice([0, 0, 3])) if a.locality else 0
            st = _hip.knn_search(qd, rd, k, splits=splits)
            if share:
                joined = _hip.knn_share_info(n, m, d, k, splits=splits, read_stats=True).get("joined")
        else:  # reference blocks merged one after the other, direct evaluation for what stays uncertified
            cuts = np.linspace(0, m, blocks + 1).astype(int)
            st = None
            for lo, hi in zip(cuts, cuts[1:]):
                st = _hip.knn_search(qd, rd[lo:hi], k, col_offset=int(lo), state=st, exact_fallback=False)
            n_unc = int(st.n_uncertified.item())
            rows = st.uncertified_rows[:max(n_unc, 1)].clone()
            n_rows = torch.tensor([n_unc], dtype=torch.int32, device="cuda")
            for lo, hi in zip(cuts, cuts[1:]):
                _hip.knn_exact_rows(qd, rd[lo:hi], k, rows, n_rows, st, col_offset=int(lo))
        ok = np.array_equal(st.idx.cpu().numpy(), idx_o) and np.array_equal(st.dist2.cpu().numpy(), d2_o)
        match = rng.integers(0, m, n).astype(np.int32)
        cnt_o = oracle.closer_count(q, r, match)
        cnt = _hip.closer_count(qd, rd, match=torch.from_numpy(match).cuda()).cpu().numpy()
        ok2 = np.array_equal(cnt, cnt_o.astype(np.int64))
        _hip.set_tuning()
        _hip.set_sharing()
        if share:
            knobs = dict(knobs, share=share, joined=joined)
        print(f"case {case:3d} n={n:5d} m={m:6d} d={d:3d} k={k:2d}  knn {'ok' if ok else 'DIFFERS'}  "
              f"count {'ok' if ok2 else 'DIFFERS'}  uncertified={int(st.n_uncertified.item())} blocks={blocks} {knobs}", flush=True)
        if not (ok and ok2):
            sys.exit(1)
    print(f"{a.cases} cases identical in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
