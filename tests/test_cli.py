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
    assert got == REFERENCE_DEFAULTS
    args = cli.build_parser().parse_args(
        ["train", "--mode", "v", "--hidden-dims", "64, 32", "--no-balanced-sampler", "--k-anchor", "7"])
    assert (args.mode, args.hidden_dims, args.balanced_sampler, args.k_anchor) == ("v", [64, 32], False, 7)
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["train", "--mode", "d"])


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
