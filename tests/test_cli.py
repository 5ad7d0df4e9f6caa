"""The command line keeps the reference's `scmiac train` surface (scmiac/cli.py:168-228):
flag names, defaults, the missing-argument message and the help-on-no-arguments exit."""
import pytest

from scmiac_amd import cli

REFERENCE_DEFAULTS = {  # scmiac/cli.py:182-208
    "rna_h5ad": None, "atac_h5ad": None, "output_dir": None, "all_nfeatures": 3000, "single_nfeatures": 2000,
    "k_anchor": 5, "n_components": 30, "mode": None, "rna_latent_key": "X_pca", "atac_latent_key": "lsi49",
    "rna_celltype_key": "cell_type", "atac_celltype_key": "pred", "batch_size": 256, "hidden_dims": [128, 64],
    "latent_dim": 30, "balanced_sampler": True, "device": "cuda:0", "num_epochs": 2000, "lambda_rna_kl": 1.0,
    "lambda_atac_kl": 1.0, "alpha_rna_rec": 20.0, "alpha_atac_rec": 20.0, "lambda_contra": 300.0,
    "temperature": 0.5, "learning_rate": 1e-3, "print_step": 10,
}


def test_train_flags_and_defaults():
    args = cli.build_parser().parse_args(["train"])
    got = {k: v for k, v in vars(args).items() if k not in ("command", "func")}
    # two options of this package's own, both defaulting to the reference's behaviour
    assert got.pop("search_mode") == "dense" and got.pop("device_consumer") is False
    assert got == REFERENCE_DEFAULTS
    args = cli.build_parser().parse_args(
        ["train", "--mode", "v", "--hidden-dims", "64, 32", "--no-balanced-sampler", "--k-anchor", "7"])
    assert (args.mode, args.hidden_dims, args.balanced_sampler, args.k_anchor) == ("v", [64, 32], False, 7)
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["train", "--mode", "d"])
    for mode in ("dense", "pruned", "auto"):
        assert cli.build_parser().parse_args(["train", "--search-mode", mode]).search_mode == mode
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["train", "--search-mode", "fast"])


def test_missing_required_argument_message():
    with pytest.raises(SystemExit) as e:
        cli.main(["train", "--atac-h5ad", "a.h5ad", "--output-dir", "out"])
    assert str(e.value) == "Missing required argument: --rna-h5ad. Provide it via CLI or YAML config."


def test_no_arguments_prints_help_and_returns_1(capsys):
    assert cli.main([]) == 1
    assert "train" in capsys.readouterr().out


def test_anchor_csv_layout(tmp_path):
    import pandas as pd

    df = pd.DataFrame({"x1": [3, 1], "x2": [0, 2], "score": [0.25, 1.0], "x1_ct": ["B", "T"],
                       "x2_ct": ["B", "T"], "is_same": [True, True]}, index=[0, 4])
    path = cli.write_anchors(df, tmp_path / "nested" / "out")
    assert path.name == "anchors.csv"
    assert path.read_text() == "x1,x2,score,x1_ct,x2_ct,is_same\n3,0,0.25,B,B,True\n1,2,1.0,T,T,True\n"


def test_device_consumer_hands_device_loaders_and_fused_loss_to_the_reference(tmp_path, monkeypatch):
    """``--device-consumer``: the reference's ``train_model`` receives this package's loaders, and the
    loss class it looks up in its own module is this package's; without the flag nothing is touched."""
    import sys
    import types

    import pandas as pd
    import torch

    from scmiac_amd.modeling import dataset as device_dataset
    from scmiac_amd.modeling.loss import NTXentLoss2

    seen = {}
    ref_mod = types.ModuleType("fake_scmiac_modeling")
    ref_mod.NTXentLoss2 = "the reference's eager loss"

    def preprocess(rna, atac, anchors, **kw):
        return torch.nn.Linear(2, 2), torch.nn.Linear(2, 2), "reference loader A", "reference loader B"

    def train_model(rna_vae, atac_vae, all_loader, anchor_loader, **kw):
        seen["loaders"] = (all_loader, anchor_loader)
        seen["loss"] = ref_mod.NTXentLoss2
        return rna_vae, atac_vae

    def model_inference(rna_vae, atac_vae, loader, device="cpu"):
        return [[0.0]], [[1.0]]

    train_model.__module__ = ref_mod.__name__
    monkeypatch.setitem(sys.modules, ref_mod.__name__, ref_mod)
    reference = types.SimpleNamespace(preprocess=preprocess, train_model=train_model, model_inference=model_inference)
    monkeypatch.setattr(device_dataset, "make_loaders", lambda *a, **kw: ("device loader A", "device loader B"))
    adata = types.SimpleNamespace(obs_names=["c0"])
    for flag, want_loaders, want_loss in (([], ("reference loader A", "reference loader B"), "the reference's eager loss"),
                                          (["--device-consumer"], ("device loader A", "device loader B"), NTXentLoss2)):
        args = cli.build_parser().parse_args(["train", "--output-dir", str(tmp_path), "--device", "cpu"] + flag)
        assert cli._continue_with_reference(reference, args, adata, adata, pd.DataFrame()) == 0
        assert seen["loaders"] == want_loaders and seen["loss"] is want_loss or seen["loss"] == want_loss
    assert (tmp_path / "rna_embeddings.csv").exists() and (tmp_path / "atac_vae.pth").exists()


def test_a_failing_rank_leaves_the_group_without_waiting_for_the_others(monkeypatch, tmp_path):
    """An error in the anchor stage on one rank of a torchrun job: the drop-in API is switched back,
    no barrier is entered (the other ranks may never arrive), and the error propagates."""
    import types

    from scmiac_amd import sharded
    from scmiac_amd.modeling import scmiac as M

    calls = []
    monkeypatch.setattr(cli, "_read_h5ad", lambda path: types.SimpleNamespace())
    monkeypatch.setattr(cli, "_join_ranks", lambda: (1, 4))
    monkeypatch.setattr(sharded, "use_process_group", lambda group=None, enable=True: calls.append(("group", enable)))
    import torch.distributed as dist
    monkeypatch.setattr(dist, "barrier", lambda *a, **kw: calls.append("barrier"))
    monkeypatch.setattr(dist, "destroy_process_group", lambda *a, **kw: calls.append("destroy"))

    def boom(*a, **kw):
        raise KeyError("cell_type")

    monkeypatch.setattr(M, "find_anchors", boom)
    args = cli.build_parser().parse_args(["train", "--rna-h5ad", "r", "--atac-h5ad", "a", "--output-dir", str(tmp_path)])
    with pytest.raises(KeyError):
        cli.run_train(args)
    assert calls == [("group", False)]
