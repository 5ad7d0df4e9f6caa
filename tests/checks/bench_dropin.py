#!/usr/bin/env python3
"""Time the drop-in numpy API (find_neighbor -> find_mnn -> score_anchor) at the size of
the 10x PBMC multiome set (10 412 x 10 412 cells, d=30, unit-norm rows), host arrays in,
host arrays out, next to the CPU oracle on the same inputs.

    python tests/checks/bench_dropin.py [--n 10412] [--d 30]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle  # noqa: E402  (checker / CPU baseline only)
from scmiac_amd.preprocessing import anchors as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10412)
    ap.add_argument("--d", type=int, default=30)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--genes", type=int, default=3000, help="genes of the count matrices for the whole-stage run")
    a = ap.parse_args()
    rng = np.random.default_rng(1001)
    centres = rng.normal(scale=3.0, size=(12, a.d))
    U = centres[rng.integers(0, 12, a.n)] + rng.normal(size=(a.n, a.d))
    V = centres[rng.integers(0, 12, a.n)] + 0.25 * rng.normal(size=a.d) + rng.normal(size=(a.n, a.d))
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    k = 30

    def run():
        G11, G12, G21, G22 = A.find_neighbor(U, V, k)
        pairs = A.find_mnn(G12, G21, 5)
        return (G11, G12, G21, G22), pairs, A.score_anchor(pairs, G11, G12, G21, G22, k_score=30, k_local=None)

    run()  # library load, workspace
    times = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tables, pairs, df = run()
        times.append(time.perf_counter() - t0)
    oracle.build()
    t0 = time.perf_counter()
    ot = oracle.find_neighbor(U, V, k)
    t1 = time.perf_counter()
    op = oracle.find_mnn(ot[1], ot[2], 5)
    odf = oracle.score_anchor(op, *ot, k_score=30, k_local=None)
    t2 = time.perf_counter()
    same = all(np.array_equal(x, y) for x, y in zip(tables, ot)) and np.array_equal(pairs, op) \
        and np.array_equal(df["score"].values, odf["score"].values)
    print(f"{a.n} x {a.n} cells, d={a.d}: drop-in API on the GPU {1e3 * np.median(times):.1f} ms "
          f"(min {1e3 * min(times):.1f}); CPU oracle {t1 - t0:.2f} s tables + {t2 - t1:.2f} s MNN/score on "
          f"{oracle.host_threads()} cores; {len(pairs)} anchors; identical: {same}")


    whole_stage(a)


class _Counts:
    """The AnnData members SeuratIntegration touches."""

    def __init__(self, X):
        import pandas as pd
        self.X, self.shape, self.obsm = X, X.shape, {}
        self.var = type("V", (), {"index": pd.Index([f"g{i}" for i in range(X.shape[1])])})()


def whole_stage(a):
    """The whole anchor stage from count matrices: SeuratIntegration.find_anchor(dim_red="cca")
    as find_anchors calls it (scmiac.py:62-68), CCA and search on the GPU, next to the oracle's
    CCA (dense product + scikit-learn) and search on the CPU."""
    import contextlib
    import io

    rng = np.random.default_rng(7)
    prof = rng.gamma(2.0, 1.0, size=(12, a.genes)).astype(np.float32)
    X1 = np.log1p(rng.poisson(prof[rng.integers(0, 12, a.n)] * 0.5)).astype(np.float64)
    X2 = np.log1p(rng.poisson(prof[rng.integers(0, 12, a.n)] * 0.3)).astype(np.float64)
    kw = dict(k_local=None, key_anchor="X", dim_red="cca", max_cc_cells=50000, k_anchor=5, k_score=30,
              scale1=True, scale2=True, n_components=30, k_filter=None, alignments=[[[0], [1]]])

    def run():
        integ = A.SeuratIntegration()
        with contextlib.redirect_stdout(io.StringIO()):
            integ.find_anchor([_Counts(X1), _Counts(X2)], **kw)
        return integ

    run()
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        integ = run()
        times.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        U, V, _ = oracle.cca(X1, X2, n_components=30, max_cc_cell=50000)
    t1 = time.perf_counter()
    odf, _, _ = oracle.pairwise_anchors(U, V, k_anchor=5, k_score=30, normalize_rows=True)
    t2 = time.perf_counter()
    got = integ.anchor[(0, 1)]
    same = np.array_equal(got["x1"].values, odf["x1"].values) and np.array_equal(got["x2"].values, odf["x2"].values) \
        and np.array_equal(got["score"].values, odf["score"].values)
    print(f"whole anchor stage from {a.n} x {a.genes} count matrices: GPU {1e3 * np.median(times):.0f} ms; "
          f"CPU {t1 - t0:.2f} s CCA + {t2 - t1:.2f} s search/score on {oracle.host_threads()} cores; "
          f"{len(got)} anchors; identical: {same}")


if __name__ == "__main__":
    main()
