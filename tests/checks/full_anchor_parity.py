"""The whole anchor search of a bench.py workload (four tables, mutual nearest neighbours,
shared-neighbour scores, label filter) on the GPU against the CPU oracle: every table entry,
every anchor, every score.

    python tests/checks/full_anchor_parity.py [--workload c3 --n 250000 --m 250000] [--out file.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from oracle import oracle  # noqa: E402  (checker)
from scmiac_amd import anchor_search  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--n", type=int, default=250000)
    ap.add_argument("--m", type=int, default=250000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    U, V, l1, l2, d = bench.make_latents(a.workload, a.n, a.m)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    tables = anchor_search.find_neighbor_device(torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev), a.k)
    scores, n_pairs = anchor_search.mnn_and_score(tables, k_anchor=5, k_score=min(30, a.k))
    c1, c2 = anchor_search.label_codes(l1, l2)
    kept = anchor_search.filter_anchors(scores, n_pairs, torch.from_numpy(c1).to(dev), torch.from_numpy(c2).to(dev))
    torch.cuda.synchronize()
    gpu_s = time.time() - t0
    print(f"GPU: {scores.n} raw anchors, {len(kept.x1)} kept, {gpu_s:.2f} s", flush=True)

    oracle.build()
    t0 = time.time()
    want_tables = []
    names = ("G11", "G12", "G21", "G22")
    odf, G, raw = oracle.pairwise_anchors(U, V, k_anchor=5, k_score=min(30, a.k))
    print(f"oracle: {len(odf)} raw anchors, {time.time() - t0:.0f} s", flush=True)
    want = oracle.label_filter(odf, l1, l2)
    diff = {}
    for name, got, ref in zip(names, tables.numpy(), G):
        diff[name] = int((got != ref).any(axis=1).sum())
    pairs_g = scores.pairs[:scores.n].cpu().numpy()
    same_pairs = pairs_g.shape[0] == len(odf) and bool(np.array_equal(pairs_g[:, 0], odf["x1"].values)
                                                       and np.array_equal(pairs_g[:, 1], odf["x2"].values))
    same_scores = same_pairs and bool(np.array_equal(scores.score, odf["score"].values))
    same_kept = bool(len(kept.x1) == len(want) and np.array_equal(kept.x1, want["x1"].values)
                     and np.array_equal(kept.x2, want["x2"].values) and np.array_equal(kept.score, want["score"].values))
    out = {"workload": a.workload, "n": int(U.shape[0]), "m": int(V.shape[0]), "d": int(d), "k": a.k,
           "rows_with_differences_per_table": diff, "raw_anchors": int(scores.n), "raw_anchor_pairs_identical": same_pairs,
           "scores_identical": same_scores, "anchors_after_filter": int(len(kept.x1)), "filtered_table_identical": same_kept,
           "gpu_seconds": round(gpu_s, 3), "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": oracle.host_threads()}
    print(json.dumps(out))
    if a.out:
        Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
    ok = not any(diff.values()) and same_pairs and same_scores and same_kept
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
