"""TEST INFRASTRUCTURE -- regenerates tests/golden/*.npz from the reference.

Runs only where the reference checkout is mounted (``/root/reference``); the
committed fixtures are what the tests read.  The reference package cannot be
imported as shipped (``anndata``, ``scanpy`` and ``pynndescent`` are not
installed), so its two hot-path files are loaded BY PATH with empty stand-in
modules for the absent imports, and ``pynndescent.NNDescent`` is served by the
oracle's exact search (``oracle.ExactNNDescent``).  Everything else that runs --
``find_neighbor``, ``find_mnn``, ``min_max``, ``score_anchor``,
``SeuratIntegration.find_anchor`` (``cca`` and ``precomputed`` branches) and
``scmiac.find_anchors``'s label join and filters -- is the reference's own code
on real numpy / pandas / scikit-learn.  Only input and output arrays are saved.

    python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import contextlib
import importlib.util
import io
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parents[1]
sys.path[0] = str(ROOT)  # the script's own directory would shadow the package
from oracle import oracle  # noqa: E402


# ----------------------------------------------------------------- reference loader
def load_reference(ref_root: Path):
    """Return (anchors_module, scmiac_module) loaded from the reference tree."""
    for name in ("anndata", "scanpy"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    nnd = types.ModuleType("pynndescent")
    nnd.NNDescent = oracle.ExactNNDescent
    sys.modules["pynndescent"] = nnd

    pkg_root = ref_root / "scmiac"
    # Bare package shells so the relative imports inside modeling/scmiac.py
    # resolve without executing the package __init__ files (which pull in
    # scanpy-dependent modules that are off the path).
    for name, sub in (("scmiac", ""), ("scmiac.preprocessing", "preprocessing"),
                      ("scmiac.modeling", "modeling")):
        mod = types.ModuleType(name)
        mod.__path__ = [str(pkg_root / sub) if sub else str(pkg_root)]
        sys.modules[name] = mod

    def _load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, str(pkg_root / relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    anchors = _load("scmiac.preprocessing.anchors", "preprocessing/anchors.py")
    _load("scmiac.modeling.dataset", "modeling/dataset.py")
    _load("scmiac.modeling.model", "modeling/model.py")
    _load("scmiac.modeling.loss", "modeling/loss.py")
    scm = _load("scmiac.modeling.scmiac", "modeling/scmiac.py")
    return anchors, scm


# ------------------------------------------------------------------- fake AnnData
class _Var:
    def __init__(self, names):
        self.index = pd.Index(names)


class FakeAnnData:
    """The handful of AnnData members the anchor path touches."""

    def __init__(self, X, obs=None, obsm=None, var_names=None):
        self.X = np.asarray(X)
        self.shape = self.X.shape
        self.obs = obs if obs is not None else pd.DataFrame(index=np.arange(self.X.shape[0]).astype(str))
        self.obsm = dict(obsm or {})
        names = var_names if var_names is not None else [f"g{i}" for i in range(self.X.shape[1])]
        self.var = _Var(names)
        self.var_names = self.var.index


# -------------------------------------------------------------------- data makers
def mixture(rng, n, d, centres, offset=None, unit=False, dtype=np.float64):
    lab = rng.integers(0, centres.shape[0], size=n)
    X = centres[lab] + rng.normal(size=(n, d))
    if offset is not None:
        X = X + offset
    if unit:
        X = X / np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(dtype), lab


def quiet(fn, *a, **kw):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **kw)
    return out, buf.getvalue()


def cca_fixture(ref, out):
    """F5: the reference's ``cca`` (anchors.py:171-242) on count-like matrices: all cells
    in the decomposition (both orientations), and more cells than ``max_cc_cell`` on both
    sides (sub-sample + chunked projection, with the feature selection of ``k_filter``)."""
    rng = np.random.default_rng(77)
    f5 = {}
    cases = {"a": (500, 420, 90, 16, 50000, 50000, None),    # n1 > n2
             "b": (380, 520, 90, 16, 50000, 50000, None),    # n1 < n2: transposed inside the SVD
             "c": (900, 800, 110, 24, 500, 300, 200),        # both sides sub-sampled
             "d": (300, 700, 80, 12, 400, 256, None)}        # one side sub-sampled
    for tag, (n1, n2, g, nc, cap, chunk, kf) in cases.items():
        prof = rng.gamma(2.0, 1.0, size=(6, g))
        l1 = rng.integers(0, 6, n1)
        l2 = rng.integers(0, 6, n2)
        X1 = np.log1p(rng.poisson(prof[l1] * 2.0)).astype(np.float64)
        X2 = np.log1p(rng.poisson(prof[l2] * 1.5)).astype(np.float64)
        X1[:, 3] = 0.0  # a gene with no counts: constant column through the scaler
        (U, V, hdf), _ = quiet(ref.cca, X1, X2, scale1=True, scale2=(tag != "d"), n_components=nc,
                               max_cc_cell=cap, chunk_size=chunk, random_state=0, k_filter=kf)
        f5[f"{tag}_X1"], f5[f"{tag}_X2"] = X1, X2
        f5[f"{tag}_args"] = np.array([nc, cap, chunk, kf if kf else 0, int(tag != "d")], dtype=np.int64)
        f5[f"{tag}_U"], f5[f"{tag}_V"] = np.asarray(U, np.float64), np.asarray(V, np.float64)
        if hdf is not None:
            f5[f"{tag}_hdf"] = np.asarray(hdf, dtype=np.int64)
    np.savez_compressed(out / "f5_cca.npz", **f5)


def ntxent_fixture(out):
    """F6: the reference's ``NTXentLoss2`` (modeling/loss.py:85-133, loaded by
    ``load_reference``) on the CPU in float32: loss and its gradients."""
    import torch
    loss_mod = sys.modules["scmiac.modeling.loss"]
    rng = np.random.default_rng(606)
    f6 = {}
    for tag, (B, d, T) in {"a": (50, 30, 0.5), "b": (256, 30, 0.5), "c": (300, 50, 0.1), "d": (7, 3, 1.0)}.items():
        base = rng.normal(size=(B, d))
        zi = (base + 0.5 * rng.normal(size=(B, d))).astype(np.float32) * 3.0
        zj = (base + 0.5 * rng.normal(size=(B, d))).astype(np.float32)
        ti = torch.tensor(zi, requires_grad=True)
        tj = torch.tensor(zj, requires_grad=True)
        loss = loss_mod.NTXentLoss2(temperature=T)(ti, tj)
        (loss * 300.0).backward()  # lambda_contra of the training loop (scmiac.py:181, 264)
        f6[f"{tag}_zi"], f6[f"{tag}_zj"] = zi, zj
        f6[f"{tag}_T"] = np.float64(T)
        f6[f"{tag}_loss"] = loss.detach().numpy()
        f6[f"{tag}_dzi"], f6[f"{tag}_dzj"] = ti.grad.numpy(), tj.grad.numpy()
    np.savez_compressed(out / "f6_ntxent.npz", **f6)


def evaluation_fixture(ref_root, out):
    """F7: the reference's distance-based metrics (evaluation/benchmark.py, loaded by
    path; it needs only numpy / scipy / scikit-learn / torch) on small embeddings."""
    spec = importlib.util.spec_from_file_location("scmiac_eval_benchmark",
                                                  str(ref_root / "scmiac" / "evaluation" / "benchmark.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    rng = np.random.default_rng(707)
    f7 = {}
    n, d = 600, 30
    centres = rng.normal(size=(8, d)) * 2.0
    lab = rng.integers(0, 8, n)
    x = (centres[lab] + rng.normal(size=(n, d))).astype(np.float32)
    y = (x + 0.9 * rng.normal(size=(n, d))).astype(np.float32)
    f7["nc_10"] = np.float64(bm.neighbor_conservation(x, y, n_neighbors=10))
    f7["nc_15_ex1"] = np.float64(bm.neighbor_conservation(x, y, n_neighbors=15, exclude_top=1))
    for k in (1, 5):
        res = bm.knn_matching(x, y, k=k)
        f7[f"match_k{k}"] = np.array([res["rna_to_atac_accuracy"], res["atac_to_rna_accuracy"],
                                      res["overall_accuracy"]], dtype=np.float64)
    f7.update(x_clean=x.copy(), y_clean=y.copy())
    x[7] = x[3]          # duplicated cells: equal distances, decided by the strict "<"
    y[11] = y[5]
    y[20] = x[20]        # a perfect match: distance zero
    fx, fy = bm.foscttm(x, y)
    f7.update(x=x, y=y, fos_x=fx, fos_y=fy)
    xu = x / np.linalg.norm(x, axis=1, keepdims=True)
    yu = y / np.linalg.norm(y, axis=1, keepdims=True)
    fxu, fyu = bm.foscttm(xu, yu)
    f7.update(xu=xu, yu=yu, fosu_x=fxu, fosu_y=fyu)
    # partially matched: x has 500 of the cells, y 450, 380 in common, shuffled
    names = np.array([f"cell{i:04d}" for i in range(n)])
    px = rng.permutation(n)[:500]
    py = np.concatenate([px[:380], np.setdiff1d(np.arange(n), px)[:70]])
    rng.shuffle(py)
    (pfx, pfy), _ = quiet(bm.partially_matched_foscttm, x[px].astype(np.float64), y[py].astype(np.float64),
                          names[px], names[py])
    f7.update(part_px=px, part_py=py, part_fx=pfx, part_fy=pfy)
    np.savez_compressed(out / "f7_evaluation.npz", **f7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden"))
    ap.add_argument("--only", default="", help="'f5' / 'f6': regenerate only the CCA / the loss fixture")
    args = ap.parse_args()
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    ref, scm = load_reference(Path(args.ref))
    if args.only == "f5":
        cca_fixture(ref, out)
        return
    if args.only == "f6":
        ntxent_fixture(out)
        return
    if args.only == "f7":
        evaluation_fixture(Path(args.ref), out)
        return

    # ---- F1: find_mnn on random tables (duplicates inside rows allowed) --------
    rng = np.random.default_rng(11)
    n1, n2, k = 300, 260, 30
    G12 = rng.integers(0, n2, size=(n1, k), dtype=np.int32)
    G21 = rng.integers(0, n1, size=(n2, k), dtype=np.int32)
    f1 = {"G12": G12, "G21": G21}
    for ka in (1, 5, 30):
        f1[f"pairs_ka{ka}"] = np.asarray(ref.find_mnn(G12, G21, ka), dtype=np.int64)
    # a table with no mutual pair at all: result has shape (0,)
    G12z = np.zeros((4, 3), dtype=np.int32)
    G21z = np.full((4, 3), 7, dtype=np.int32)  # 7 is never a query row number
    f1["G12_empty"], f1["G21_empty"] = G12z, G21z
    f1["pairs_empty_shape"] = np.asarray(ref.find_mnn(G12z, G21z, 3).shape, dtype=np.int64)
    np.savez_compressed(out / "f1_find_mnn.npz", **f1)

    # ---- F2: score_anchor + min_max on random tables --------------------------
    rng = np.random.default_rng(12)
    n1, n2, k = 200, 180, 30
    # narrow value range so rows share neighbours and hold repeated values
    G11 = rng.integers(0, 60, size=(n1, k), dtype=np.int32)
    G21 = rng.integers(0, 60, size=(n2, k), dtype=np.int32)
    G12 = rng.integers(0, 70, size=(n1, k), dtype=np.int32)
    G22 = rng.integers(0, 70, size=(n2, k), dtype=np.int32)
    pairs = np.stack([rng.integers(0, n1, 500), rng.integers(0, n2, 500)], axis=1).astype(np.int64)
    f2 = {"G11": G11, "G12": G12, "G21": G21, "G22": G22, "pairs": pairs}
    for ks in (30, 10):
        df = ref.score_anchor(pairs, G11, G12, G21, G22, k_score=ks, k_local=None)
        f2[f"score_ks{ks}"] = df["score"].values.astype(np.float64)
        f2[f"raw_ks{ks}"] = np.asarray(
            [len(set(G11[x, :ks]).intersection(G21[y, :ks])) + len(set(G12[x, :ks]).intersection(G22[y, :ks]))
             for x, y in pairs], dtype=np.int32)
    f2["minmax_in"] = rng.integers(0, 61, size=997).astype(np.int64)
    f2["minmax_out"] = ref.min_max([int(v) for v in f2["minmax_in"]])
    with np.errstate(all="ignore"):  # degenerate p90 == p1 -> nan, kept as the reference yields it
        f2["minmax_const_out"] = ref.min_max([7] * 50)
    # the local-structure branch (k_local): drawn last so the arrays above stay as they were
    Gp1 = rng.integers(0, 60, size=(n1, 20), dtype=np.int32)
    Gp2 = rng.integers(0, 70, size=(n2, 20), dtype=np.int32)
    dfl = ref.score_anchor(pairs, G11, G12, G21, G22, k_score=30, Gp1=Gp1, Gp2=Gp2, k_local=20)
    f2.update(Gp1=Gp1, Gp2=Gp2, local_score=dfl["score"].values.astype(np.float64),
              local_1=dfl["score_local1"].values.astype(np.float64),
              local_2=dfl["score_local2"].values.astype(np.float64))
    np.savez_compressed(out / "f2_score_anchor.npz", **f2)

    # ---- F3: find_neighbor on Gaussian mixtures --------------------------------
    cases = {
        "a": dict(n1=257, n2=193, d=30, C=6, seed=31, unit=True),
        "b": dict(n1=1000, n2=1000, d=50, C=10, seed=32, unit=False),
        "c": dict(n1=2048, n2=1536, d=64, C=12, seed=33, unit=True),
        "dup": dict(n1=400, n2=300, d=30, C=4, seed=34, unit=False, dup=True),
        "tiny": dict(n1=9, n2=7, d=16, C=2, seed=35, unit=False),
    }
    f3 = {}
    for name, c in cases.items():
        rng = np.random.default_rng(c["seed"])
        centres = rng.normal(scale=3.0, size=(c["C"], c["d"]))
        off = rng.normal(scale=0.25, size=c["d"])
        U, _ = mixture(rng, c["n1"], c["d"], centres, unit=c["unit"], dtype=np.float32)
        V, _ = mixture(rng, c["n2"], c["d"], centres, offset=off, unit=c["unit"], dtype=np.float32)
        if c.get("dup"):
            # exact duplicates inside and across the two sets: ties in distance
            U[50:90] = U[10]
            V[20:45] = V[3]
            V[100:110] = U[10]
            U[200:203] = V[3]
        k = min(min(c["n1"], c["n2"]) - 2, 30)
        G11, G12, G21, G22 = ref.find_neighbor(U, V, k=k)
        f3[f"{name}_U"], f3[f"{name}_V"] = U, V
        f3[f"{name}_k"] = np.int64(k)
        for nm, g in zip(("G11", "G12", "G21", "G22"), (G11, G12, G21, G22)):
            f3[f"{name}_{nm}"] = np.asarray(g, dtype=np.int32)
        # float64 reference distances of the returned cross neighbours
        f3[f"{name}_D12"] = np.sqrt(((U[:, None, :].astype(np.float64)
                                      - V[G12].astype(np.float64)) ** 2).sum(-1))
    np.savez_compressed(out / "f3_find_neighbor.npz", **f3)

    # ---- F4: SeuratIntegration.find_anchor + find_anchors label filter ---------
    f4 = {}
    rng = np.random.default_rng(41)
    C, d = 20, 50
    centres = rng.normal(scale=3.0, size=(C, d))
    off = rng.normal(scale=0.25, size=d)
    n1, n2 = 1500, 1300
    prior = np.r_[np.full(C - 1, (1 - 0.001) / (C - 1)), 0.001]
    lab1 = rng.choice(C, size=n1, p=prior)
    lab2 = rng.choice(C, size=n2, p=prior)
    lab1[:3] = C - 1  # make sure the rare label exists on both sides
    lab2[:3] = C - 1
    U = (centres[lab1] + rng.normal(size=(n1, d))).astype(np.float32)
    V = (centres[lab2] + off + rng.normal(size=(n2, d))).astype(np.float32)
    noisy = lab2.copy()
    flip = rng.random(n2) < 0.10
    noisy[flip] = (noisy[flip] + rng.integers(1, C, size=flip.sum())) % C
    names1 = np.array([f"type_{v:02d}" for v in lab1], dtype=object)
    names2 = np.array([f"type_{v:02d}" for v in noisy], dtype=object)

    a1 = FakeAnnData(np.zeros((n1, 3)), obsm={"lat": U})
    a2 = FakeAnnData(np.zeros((n2, 3)), obsm={"lat": V})
    integ = ref.SeuratIntegration()
    quiet(integ.find_anchor, [a1, a2], k_local=None, key_anchor="lat", dim_red="precomputed",
          k_anchor=5, k_score=30, k_filter=None, alignments=[[[0], [1]]])
    adf = integ.anchor[(0, 1)]
    f4.update(pre_U=U, pre_V=V, pre_lab1=names1.astype(str), pre_lab2=names2.astype(str),
              pre_x1=adf["x1"].values.astype(np.int64), pre_x2=adf["x2"].values.astype(np.int64),
              pre_score=adf["score"].values.astype(np.float64),
              pre_raw=np.asarray(integ.raw_anchor[(0, 1)], dtype=np.int64))
    for nm, g in zip(("G11", "G12", "G21", "G22"), integ.mutual_knn[(0, 1)]):
        f4[f"pre_{nm}"] = np.asarray(g, dtype=np.int32)

    # nearest anchors of every query cell with 100 neighbours (anchors.py:1189-1241)
    (anc, G, D, cum), _ = quiet(integ.find_nearest_anchor, [U, V], V, [0], [1], key_correct="lat", npc=30,
                                k_weight=100)
    f4.update(nearest_anchor=np.asarray(anc, dtype=np.int64), nearest_G=np.asarray(G, dtype=np.int32),
              nearest_D=np.asarray(D, dtype=np.float64), nearest_cum=np.asarray(cum, dtype=np.int64))

    # label join + filters exactly as scmiac.find_anchors does them (scmiac.py:90-99):
    # drive the reference function itself with the two heavy upstream steps replaced
    class _Integ(ref.SeuratIntegration):
        def find_anchor(self, adata_list, **kw):
            kw.update(key_anchor="lat", dim_red="precomputed")
            return super().find_anchor([a1, a2], **kw)

    scm.get_cca_adata_list = lambda adatas, **kw: adatas
    scm.SeuratIntegration = _Integ
    a1.obs = pd.DataFrame({"cell_type": names1})
    a2.obs = pd.DataFrame({"pred": names2})
    for ct_filter in (True, False):
        (fdf, log) = quiet(scm.find_anchors, a1, a2, ct_filter=ct_filter)
        tag = "filt" if ct_filter else "nofilt"
        f4[f"{tag}_x1"] = fdf["x1"].values.astype(np.int64)
        f4[f"{tag}_x2"] = fdf["x2"].values.astype(np.int64)
        f4[f"{tag}_score"] = fdf["score"].values.astype(np.float64)
        f4[f"{tag}_x1_ct"] = fdf["x1_ct"].values.astype(str)
        f4[f"{tag}_x2_ct"] = fdf["x2_ct"].values.astype(str)
        f4[f"{tag}_is_same"] = fdf["is_same"].values.astype(bool)
        f4[f"{tag}_index"] = fdf.index.values.astype(np.int64)
        f4[f"{tag}_csv"] = np.array(fdf.to_csv(index=False))
        f4[f"{tag}_log"] = np.array("\n".join(l for l in log.splitlines() if l.startswith("Number of")))

    # "cca" branch end to end on a small count-like matrix pair
    rng = np.random.default_rng(42)
    g, n1c, n2c, Cc = 120, 700, 600, 5
    prof = rng.gamma(2.0, 1.0, size=(Cc, g))
    l1 = rng.integers(0, Cc, n1c)
    l2 = rng.integers(0, Cc, n2c)
    X1 = np.log1p(rng.poisson(prof[l1] * 2.0)).astype(np.float64)
    X2 = np.log1p(rng.poisson(prof[l2] * 1.5)).astype(np.float64)
    c1, c2 = FakeAnnData(X1), FakeAnnData(X2)
    integ = ref.SeuratIntegration()
    quiet(integ.find_anchor, [c1, c2], k_local=None, key_anchor="X", dim_red="cca", max_cc_cells=50000,
          k_anchor=5, k_score=30, scale1=True, scale2=True, n_components=20, k_filter=None,
          alignments=[[[0], [1]]])
    cdf = integ.anchor[(0, 1)]
    f4.update(cca_X1=X1, cca_X2=X2, cca_U=np.asarray(integ.U, dtype=np.float64),
              cca_V=np.asarray(integ.V, dtype=np.float64),
              cca_x1=cdf["x1"].values.astype(np.int64), cca_x2=cdf["x2"].values.astype(np.int64),
              cca_score=cdf["score"].values.astype(np.float64))
    np.savez_compressed(out / "f4_find_anchor.npz", **f4)
    cca_fixture(ref, out)
    ntxent_fixture(out)
    evaluation_fixture(Path(args.ref), out)

    for f in sorted(out.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
