"""The reference's own CPU-runnable case at its size -- the 10x PBMC multiome of
``docs/tutorial.ipynb`` cell 6: 10 412 RNA + 10 412 ATAC cells, 3 000 selected genes, 30 canonical
components, k = 30, k_anchor = 5 -- on seeded synthetic counts (the h5ad files are not available
here): count matrices -> CCA -> unit rows -> four tables -> anchors -> scores -> label filter ->
``anchors.csv`` on the GPU against the CPU oracle (scmiac/modeling/scmiac.py:60-111).  Also
``find_neighbor`` beyond 64 neighbours (``--k-anchor`` is a user's choice, scmiac/cli.py:187,
and ``_k = max(k_anchor, k_score)``, anchors.py:885-886) and the benchmark's RCCL code path."""
import contextlib
import io
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from oracle import oracle
from scmiac_amd.modeling import scmiac as M
from scmiac_amd.preprocessing import anchors as A

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
N_CELLS, N_GENES, N_TYPES = 10412, 3000, 20


class _Counts:
    """The AnnData members ``SeuratIntegration`` touches."""

    def __init__(self, X):
        self.X, self.shape, self.obsm = X, X.shape, {}
        self.var = type("V", (), {"index": pd.Index([f"g{i}" for i in range(X.shape[1])])})()


def _pbmc_like_counts():
    rng = np.random.default_rng(10412)
    profiles = rng.gamma(2.0, 1.0, size=(N_TYPES, N_GENES)).astype(np.float32)
    l1, l2 = rng.integers(0, N_TYPES, N_CELLS), rng.integers(0, N_TYPES, N_CELLS)
    X1 = np.log1p(rng.poisson(profiles[l1] * 0.5)).astype(np.float64)
    X2 = np.log1p(rng.poisson(profiles[l2] * 0.3)).astype(np.float64)
    names = np.array([f"type {t:02d}" for t in range(N_TYPES)], dtype=object)
    lab2 = names[l2].copy()
    wrong = rng.random(N_CELLS) < 0.10                      # the ATAC labels are predictions: 10 % wrong
    lab2[wrong] = names[(l2[wrong] + rng.integers(1, N_TYPES, int(wrong.sum()))) % N_TYPES]
    return X1, X2, names[l1], lab2


def test_config1_counts_to_anchors_csv():
    X1, X2, lab1, lab2 = _pbmc_like_counts()
    integ = A.SeuratIntegration()
    with contextlib.redirect_stdout(io.StringIO()) as log:
        integ.find_anchor([_Counts(X1), _Counts(X2)], k_local=None, key_local="X_pca", k_anchor=5, key_anchor="X",
                          dim_red="cca", max_cc_cells=50000, k_score=30, scale1=True, scale2=True, n_components=30,
                          k_filter=None, n_features=200, alignments=[[[0], [1]]])
    assert "Find Anchors using k=30" in log.getvalue()
    got_df, n_pass = M.label_filter(integ, lab1, lab2, ct_filter=True)
    got_csv = got_df.to_csv(index=False)
    assert integ.U.shape == (N_CELLS, 30) and integ.V.shape == (N_CELLS, 30)

    # (1) the search, scoring, filter and CSV on the embeddings the device CCA produced: bit for bit
    odf, tables, raw_anchors = oracle.pairwise_anchors(integ.U, integ.V, k_anchor=5, k_score=30)
    for got, want in zip(integ.mutual_knn[(0, 1)], tables):
        np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(integ.raw_anchor[(0, 1)], raw_anchors)
    np.testing.assert_array_equal(integ.anchor[(0, 1)]["score"].values, odf["score"].values)
    want_df = oracle.label_filter(odf, lab1, lab2)
    assert got_csv == want_df.to_csv(index=False)
    assert n_pass == int((odf["score"] > 0.2).sum()) and len(got_df) > 5000
    np.testing.assert_array_equal(got_df.index.values, want_df.index.values)

    # (2) the CCA itself against the oracle's (dense product + scikit-learn's randomized SVD): float64
    # rounding apart.  The float32 cast of two embeddings that differ by 1e-13 can differ in an element
    # or two of 600 000, so the anchors of the two routes are compared as sets, not bit for bit.
    U, V, _ = oracle.cca(X1, X2, n_components=30, max_cc_cell=50000)
    Un, Vn = oracle.l2_normalize_rows(U), oracle.l2_normalize_rows(V)
    np.testing.assert_allclose(integ.U, Un, rtol=0, atol=1e-10)
    np.testing.assert_allclose(integ.V, Vn, rtol=0, atol=1e-10)
    odf2, _, _ = oracle.pairwise_anchors(U, V, k_anchor=5, k_score=30, normalize_rows=True)
    want2 = oracle.label_filter(odf2, lab1, lab2)
    a = set(zip(got_df["x1"].values.tolist(), got_df["x2"].values.tolist()))
    b = set(zip(want2["x1"].values.tolist(), want2["x2"].values.tolist()))
    assert len(a ^ b) <= 0.001 * len(b)


def test_config1_drop_in_numpy_api_on_unit_rows():
    """``find_neighbor -> find_mnn -> score_anchor`` with host arrays in and out, 10 412 x 10 412
    cells, d = 30, unit rows (what ``normalize`` hands to the search, anchors.py:877-878)."""
    rng = np.random.default_rng(1001)
    centres = rng.normal(scale=3.0, size=(12, 30))
    U = centres[rng.integers(0, 12, N_CELLS)] + rng.normal(size=(N_CELLS, 30))
    V = centres[rng.integers(0, 12, N_CELLS)] + 0.25 * rng.normal(size=30) + rng.normal(size=(N_CELLS, 30))
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    G = A.find_neighbor(U, V, 30)
    pairs = A.find_mnn(G[1], G[2], 5)
    df = A.score_anchor(pairs, *G, k_score=30, k_local=None)
    want_G = oracle.find_neighbor(U, V, 30)
    for got, want in zip(G, want_G):
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, want)
    want_pairs = oracle.find_mnn(want_G[1], want_G[2], 5)
    np.testing.assert_array_equal(pairs, want_pairs)
    want_df = oracle.score_anchor(want_pairs, *want_G, k_score=30, k_local=None)
    np.testing.assert_array_equal(df["score"].values, want_df["score"].values)
    assert len(pairs) > 10000


@pytest.mark.parametrize("k_anchor", [70, 64])
def test_find_anchor_with_more_than_64_neighbours(k_anchor):
    """``k_anchor = 70`` makes ``find_neighbor``'s k 70 (71 for the self tables)."""
    rng = np.random.default_rng(70)
    centres = rng.normal(scale=3.0, size=(5, 24))
    U = (centres[rng.integers(0, 5, 1500)] + rng.normal(size=(1500, 24))).astype(np.float32)
    V = (centres[rng.integers(0, 5, 1300)] + 0.2 + rng.normal(size=(1300, 24))).astype(np.float32)
    V[100:140] = V[7]  # duplicated rows: (distance, row number) ties inside the wide lists
    for got, want in zip(A.find_neighbor(U, V, k_anchor), oracle.find_neighbor(U, V, k_anchor)):
        assert got.shape[1] == k_anchor
        np.testing.assert_array_equal(got, want)
    a1, a2 = _Counts(np.zeros((1500, 2))), _Counts(np.zeros((1300, 2)))
    a1.obsm["lat"], a2.obsm["lat"] = U, V
    integ = A.SeuratIntegration()
    with contextlib.redirect_stdout(io.StringIO()) as log:
        integ.find_anchor([a1, a2], k_local=None, key_anchor="lat", dim_red="precomputed", k_anchor=k_anchor,
                          k_score=30, alignments=[[[0], [1]]])
    assert f"Find Anchors using k={k_anchor}" in log.getvalue()
    odf, tables, raw_anchors = oracle.pairwise_anchors(U, V, k_anchor=k_anchor, k_score=30)
    for got, want in zip(integ.mutual_knn[(0, 1)], tables):
        np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(integ.raw_anchor[(0, 1)], raw_anchors)
    got = integ.anchor[(0, 1)]
    np.testing.assert_array_equal(got["x1"].values, odf["x1"].values)
    np.testing.assert_array_equal(got["x2"].values, odf["x2"].values)
    np.testing.assert_array_equal(got["score"].values, odf["score"].values)


@pytest.mark.parametrize("extra", [[], ["--ring"]])
def test_bench_rccl_code_path_with_one_rank(extra):
    """``bench.py`` with its multi-rank code path forced on one GPU: RCCL process group of one,
    device evidence in the line, gather (and ring) inside the timed region, parity sample clean."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SCMIAC_BENCH_FORCE_DIST"] = "1"
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--workload", "small", "--steps", "1",
                        "--warmup", "0", "--cpu-rows", "400"] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert cfg["backend"] == "nccl" and cfg["world"] == 1 and len(cfg["devices"]) == 1 and "pci" in cfg["devices"][0]
    assert ("ring" in cfg["sharding"]) == bool(extra)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["parity"]["rows_with_index_mismatch"] == 0 and out["parity"]["dist2_bit_identical"] is True
    assert out["cpu_baseline"]["cores"] >= 1
