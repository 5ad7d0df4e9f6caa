"""The anchor table's consumer on the GPU: the fused symmetric NT-Xent loss against the
reference's ``NTXentLoss2`` (fixture F6, made by the reference's own module on the CPU
in float32) and the float64 oracle, and the device-resident loaders against the
reference's dataset semantics."""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import oracle
from scmiac_amd.modeling.dataset import DeviceAnchorLoader, DeviceMultiOmicLoader, anchor_frame, make_loaders
from scmiac_amd.modeling.loss import NTXentLoss2, ntxent2

pytestmark = pytest.mark.gpu

# float32 kernels against a float32 reference: both carry rounding of the order of
# 1e-7 relative per operation over sums of a few hundred terms
LOSS_RTOL = 2e-6
GRAD_RTOL, GRAD_ATOL_REL = 1e-4, 2e-6  # atol = GRAD_ATOL_REL * max|grad|


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_ntxent2_matches_reference_fixture(golden, tag):
    f = golden["f6_ntxent"]
    zi = torch.tensor(f[f"{tag}_zi"], device="cuda", requires_grad=True)
    zj = torch.tensor(f[f"{tag}_zj"], device="cuda", requires_grad=True)
    loss = NTXentLoss2(temperature=float(f[f"{tag}_T"]))(zi, zj)
    (loss * 300.0).backward()
    np.testing.assert_allclose(loss.item(), float(f[f"{tag}_loss"]), rtol=LOSS_RTOL)
    for got, want in ((zi.grad, f[f"{tag}_dzi"]), (zj.grad, f[f"{tag}_dzj"])):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=GRAD_RTOL,
                                   atol=GRAD_ATOL_REL * float(np.abs(want).max()))


@pytest.mark.parametrize("B,d,T", [(1, 5, 0.5), (2, 1, 0.5), (63, 32, 0.2), (65, 33, 0.5), (129, 64, 0.07),
                                   (1000, 30, 0.5), (4096, 16, 0.5),
                                   # the MFMA kernels' range: partial chunks, partial tiles, both widths
                                   (512, 30, 0.5), (513, 33, 0.2), (777, 50, 0.1), (2049, 64, 0.07)])
def test_ntxent2_shapes_against_oracle(B, d, T):
    rng = np.random.default_rng(B * 131 + d)
    base = rng.normal(size=(B, d))
    zi = (base + 0.3 * rng.normal(size=(B, d))).astype(np.float32)
    zj = (base + 0.3 * rng.normal(size=(B, d))).astype(np.float32) * 0.01
    want_loss, want_dzi, want_dzj = oracle.ntxent2(zi, zj, T)
    ti = torch.tensor(zi, device="cuda", requires_grad=True)
    tj = torch.tensor(zj, device="cuda", requires_grad=True)
    loss = ntxent2(ti, tj, T)
    loss.backward()
    np.testing.assert_allclose(loss.item(), want_loss, rtol=1e-5, atol=1e-6)
    # the gradient is a sum of (softmax - one-hot) terms: float32 rounding of the
    # similarities (~1e-7 / T) survives the cancellation as an absolute error, so the
    # bound is relative to the largest gradient entry and widens with 1 / T
    for got, want in ((ti.grad, want_dzi), (tj.grad, want_dzj)):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4,
                                   atol=2e-5 / T * float(np.abs(want).max()) + 1e-12)


def test_ntxent2_in_a_training_step_and_errors():
    torch.manual_seed(0)
    enc_a = torch.nn.Linear(20, 30).cuda()
    enc_b = torch.nn.Linear(24, 30).cuda()
    xa, xb = torch.randn(256, 20, device="cuda"), torch.randn(256, 24, device="cuda")
    loss = NTXentLoss2(0.5)(enc_a(xa), enc_b(xb)) * 300
    loss.backward()
    # the same step with the reference's formulation in eager PyTorch
    ref_a, ref_b = torch.nn.Linear(20, 30).cuda(), torch.nn.Linear(24, 30).cuda()
    ref_a.load_state_dict(enc_a.state_dict())
    ref_b.load_state_dict(enc_b.state_dict())
    zi = torch.nn.functional.normalize(ref_a(xa), dim=1)
    zj = torch.nn.functional.normalize(ref_b(xb), dim=1)
    S = zi @ zj.t() / 0.5
    tgt = torch.arange(256, device="cuda")
    eager = 0.5 * (torch.nn.functional.cross_entropy(S, tgt) + torch.nn.functional.cross_entropy(S.t(), tgt)) * 300
    eager.backward()
    np.testing.assert_allclose(loss.item(), eager.item(), rtol=1e-5)
    np.testing.assert_allclose(enc_a.weight.grad.cpu().numpy(), ref_a.weight.grad.cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(enc_b.bias.grad.cpu().numpy(), ref_b.bias.grad.cpu().numpy(), rtol=1e-3, atol=1e-4)
    with pytest.raises(ValueError):
        ntxent2(torch.zeros(4, 65, device="cuda"), torch.zeros(4, 65, device="cuda"))
    with pytest.raises(TypeError):
        ntxent2(torch.zeros(4, 8, device="cuda", dtype=torch.float64), torch.zeros(4, 8, device="cuda"))
    with pytest.raises(ValueError):
        ntxent2(torch.zeros(4, 8, device="cuda"), torch.zeros(5, 8, device="cuda"))


def _toy(n1=700, n2=500, d1=12, d2=9, n_anchor=900, seed=0):
    rng = np.random.default_rng(seed)
    rna = rng.normal(size=(n1, d1)) * rng.uniform(0.5, 3, d1) + rng.normal(size=d1)
    atac = rng.normal(size=(n2, d2)) * rng.uniform(0.5, 3, d2)
    atac[:, 2] = 1.5  # constant column: scale 1 in StandardScaler
    ct = rng.choice(["B", "T", "NK", "rare"], size=n_anchor, p=[0.55, 0.3, 0.14, 0.01])
    ct[:2] = "rare"
    df = pd.DataFrame({"x1": rng.integers(0, n1, n_anchor), "x2": rng.integers(0, n2, n_anchor),
                       "score": rng.uniform(0.2, 1, n_anchor), "x1_ct": ct})
    return rna, atac, df


def test_loaders_follow_reference_dataset_semantics():
    from sklearn.preprocessing import StandardScaler
    rna, atac, df = _toy()
    want_rna = StandardScaler().fit_transform(rna).astype(np.float32)   # dataset.py:42-45, 62-63
    want_atac = StandardScaler().fit_transform(atac).astype(np.float32)
    all_cells = DeviceMultiOmicLoader(rna, atac, batch_size=64, seed=5)
    np.testing.assert_allclose(all_cells.dataset.rna_data, want_rna, rtol=0, atol=1e-6)
    np.testing.assert_allclose(all_cells.dataset.atac_data, want_atac, rtol=0, atol=1e-6)
    assert len(all_cells) == 700 // 64
    seen = []
    for batch in all_cells:
        assert batch["rna"].shape == (64, 12) and batch["atac"].shape == (64, 9) and batch["rna"].is_cuda
        seen.append(batch)
    assert len(seen) == len(all_cells)
    # unshuffled: sample i pairs RNA row i % n_rna with ATAC row i % n_atac (dataset.py:55-56)
    plain = DeviceMultiOmicLoader(rna, atac, batch_size=100, shuffle=False, drop_last=False)
    batches = list(plain)
    assert len(batches) == 7
    got_atac = torch.cat([b["atac"] for b in batches]).cpu().numpy()
    np.testing.assert_array_equal(got_atac, all_cells.dataset.atac_data[np.arange(700) % 500])

    anchors = DeviceAnchorLoader(rna, atac, df, batch_size=128, balanced=False, seed=1)
    order = anchors.draw()
    assert sorted(order.cpu().tolist()) == list(range(900))          # every anchor once
    assert len(anchors) == 900 // 128
    anchors.generator.manual_seed(1)
    order = anchors.draw().cpu().numpy()
    anchors.generator.manual_seed(1)
    for b, batch in enumerate(anchors):
        idx = order[b * 128:(b + 1) * 128]
        np.testing.assert_array_equal(batch["rna_anchor"].cpu().numpy(),
                                      all_cells.dataset.rna_data[df["x1"].values[idx]])   # dataset.py:86-90
        np.testing.assert_array_equal(batch["atac_anchor"].cpu().numpy(),
                                      all_cells.dataset.atac_data[df["x2"].values[idx]])
        assert list(anchors.ct_categories[batch["ct"].cpu().numpy()]) == list(df["x1_ct"].values[idx])
    bad = df.copy()
    bad.loc[3, "x2"] = 500
    with pytest.raises(IndexError):
        DeviceAnchorLoader(rna, atac, bad)


def test_balanced_anchor_sampling_and_make_loaders():
    rna, atac, df = _toy(n_anchor=20000, seed=3)

    class _Ad:
        def __init__(self, key, arr):
            self.obsm = {key: arr}

    all_cells, anchors = make_loaders(_Ad("X_pca", rna), _Ad("lsi49", atac), df, batch_size=256, seed=11)
    assert anchors.rna.data_ptr() == all_cells.rna.data_ptr()          # one resident copy
    draws = torch.cat([anchors.draw() for _ in range(10)]).cpu().numpy()
    assert draws.shape == (200000,)
    share = pd.Series(df["x1_ct"].values[draws]).value_counts(normalize=True)
    # weights 1 / count(type) (scmiac.py:153-156): every type is drawn equally often
    assert set(share.index) == {"B", "T", "NK", "rare"}
    assert np.abs(share.values - 0.25).max() < 0.01
    n = 0
    for full, anc in zip(all_cells, anchors):                         # scmiac.py:239
        assert full["rna"].shape[0] == anc["rna_anchor"].shape[0] == 256
        n += 1
    assert n == min(len(all_cells), len(anchors)) == 2
    # same seed, same stream
    again = make_loaders(_Ad("X_pca", rna), _Ad("lsi49", atac), df, batch_size=256, seed=11)[1]
    anchors.generator.manual_seed(12)
    np.testing.assert_array_equal(anchors.draw().cpu().numpy(), again.draw().cpu().numpy())


def test_anchor_loader_from_the_device_side_filter_result():
    """Anchors straight from the search (index arrays + labels), without a CSV in between."""
    from scmiac_amd import anchor_search

    rng = np.random.default_rng(2)
    centres = rng.normal(scale=3.0, size=(5, 20))
    l1, l2 = rng.integers(0, 5, 600), rng.integers(0, 5, 500)
    U = (centres[l1] + rng.normal(size=(600, 20))).astype(np.float32)
    V = (centres[l2] + rng.normal(size=(500, 20))).astype(np.float32)
    tables = anchor_search.find_neighbor_device(torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda(), 30)
    scores, n_pairs = anchor_search.mnn_and_score(tables, k_anchor=5, k_score=30)
    c1, c2 = anchor_search.label_codes(l1, l2)
    kept = anchor_search.filter_anchors(scores, n_pairs, c1, c2)
    frame = anchor_frame(kept.x1, kept.x2, labels=l1)
    loader = DeviceAnchorLoader(U, V, frame, batch_size=64, seed=0)
    batch = next(iter(loader))
    assert batch["rna_anchor"].shape == (64, 20) and batch["atac_anchor"].shape == (64, 20)
    # anchors passed the label filter: both cells of a drawn pair carry the same label
    codes = batch["ct"].cpu().numpy()
    assert set(loader.ct_categories[codes]) <= {str(v) for v in range(5)}
    assert len(loader) == len(kept.x1) // 64
