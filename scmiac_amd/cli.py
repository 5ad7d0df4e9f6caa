"""``scmiac train`` with the reference's flags (``scmiac/cli.py:168-210``).

This command runs the stage this package implements -- anchor generation -- and
writes ``<output-dir>/anchors.csv`` exactly where and how the reference does
(``cli.py:57, 78-79``).  The VAE stages that follow in the reference
(``cli.py:81-163``) are not part of this package: when the reference package
``scmiac`` is importable they are handed the anchors and run unchanged, otherwise
the command stops after the anchors with a notice.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path
from typing import Sequence


MISSING = "Missing required argument: {flag}. Provide it via CLI or YAML config."  # reference wording (cli.py:32)


def _demand(args, *dests):
    """Stop with the reference's message for the first of ``dests`` the user left out."""
    for dest in dests:
        if getattr(args, dest) is None:
            raise SystemExit(MISSING.format(flag="--" + dest.replace("_", "-")))


def _layer_widths(text: str) -> list:
    """``"128,64"`` -> ``[128, 64]``; blanks around and between the commas are ignored."""
    fields = [f for f in (piece.strip() for piece in text.split(",")) if f]
    if not fields or not all(f.lstrip("+-").isdigit() for f in fields):
        raise argparse.ArgumentTypeError(f"expected integers separated by commas, got {text!r}")
    return [int(f) for f in fields]


def _read_h5ad(path):
    try:
        import anndata
    except ImportError as exc:  # pragma: no cover - depends on the environment
        raise SystemExit(f"Reading {path} needs the 'anndata' package: {exc}")
    return anndata.read_h5ad(path)


def _join_ranks():
    """Started by ``torchrun --nproc-per-node N``: one rank per GPU, RCCL process group, and the
    anchor search's k-NN tables shared out over the ranks.  Returns ``(rank, world)``; ``(0, 1)``
    for a plain single-process start."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist

    from . import sharded

    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    sharded.use_process_group()
    return dist.get_rank(), world


def _leave_ranks(failed=False):
    """Back to a plain process.  After an error on this rank there is no barrier: the other ranks
    may be waiting in a collective this rank will never join, and the launcher ends them when this
    process exits with its error."""
    import torch.distributed as dist

    from . import sharded

    sharded.use_process_group(enable=False)
    if not failed:
        dist.barrier()
        dist.destroy_process_group()


def write_anchors(anchor_df, output_dir) -> Path:
    """``anchors.csv`` as the reference writes it: header x1,x2,score,x1_ct,x2_ct,is_same, no index."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    anchor_path = output_dir / "anchors.csv"
    anchor_df.to_csv(anchor_path, index=False)
    return anchor_path


def run_train(args: argparse.Namespace) -> int:
    _demand(args, "rna_h5ad", "atac_h5ad", "output_dir")

    from .modeling.scmiac import find_anchors

    adata_rna = _read_h5ad(args.rna_h5ad)
    adata_atac = _read_h5ad(args.atac_h5ad)
    if args.search_mode != "dense":
        from . import _hip
        _hip.set_default_mode(args.search_mode)

    # Under torchrun: one rank per GPU.  Every rank reads both files and runs the whole stage (gene
    # selection, CCA, scoring and filter replicated); only the k-NN tables' query rows are shared
    # out over the ranks.
    rank, world = _join_ranks()
    print("Generating MNN anchors...")
    try:
        anchor_df = find_anchors(
            adata_rna, adata_atac,
            all_nfeatures=args.all_nfeatures, single_nfeatures=args.single_nfeatures,
            k_anchor=args.k_anchor, n_components=args.n_components,
            ct_filter=True,  # the command always filters by cell type, like the reference's
            mode=args.mode,
            rna_celltype_key=args.rna_celltype_key, atac_celltype_key=args.atac_celltype_key,
        )
    except BaseException:
        if world > 1:
            _leave_ranks(failed=True)
        raise
    if world > 1:
        _leave_ranks()
        if rank != 0:  # every rank holds the same table: one of them writes it and goes on
            return 0
    anchor_path = write_anchors(anchor_df, args.output_dir)
    print(f"MNN anchors saved to {anchor_path} ({len(anchor_df)} pairs)")

    try:
        import scmiac as reference  # the VAE stages live in the reference package
    except ImportError:
        print("Anchor generation finished. The VAE training stages are not part of this package; "
              "install scMIAC to continue from anchors.csv.")
        return 0
    return _continue_with_reference(reference, args, adata_rna, adata_atac, anchor_df)


def _continue_with_reference(reference, args, adata_rna, adata_atac, anchor_df) -> int:  # pragma: no cover
    import pandas as pd
    import torch

    output_dir = Path(args.output_dir)
    rna_vae, atac_vae, all_cells_loader, anchor_cells_loader = reference.preprocess(
        adata_rna, adata_atac, anchor_df,
        rna_latent_key=args.rna_latent_key, atac_latent_key=args.atac_latent_key,
        batch_size=args.batch_size, hidden_dims=args.hidden_dims, latent_dim=args.latent_dim,
        balanced_sampler=args.balanced_sampler, device=args.device)
    if args.device_consumer:
        # the reference's training loop, fed by this package's device-resident loaders and fused
        # anchor loss: train_model iterates whatever loaders it is handed (scmiac.py:239) and looks
        # its loss class up in its own module (scmiac.py:219), so neither needs an edit of scMIAC
        import importlib

        from .modeling import dataset as device_dataset
        from .modeling.loss import NTXentLoss2

        all_cells_loader, anchor_cells_loader = device_dataset.make_loaders(
            adata_rna, adata_atac, anchor_df, rna_latent_key=args.rna_latent_key,
            atac_latent_key=args.atac_latent_key, batch_size=args.batch_size,
            balanced_sampler=args.balanced_sampler, device=args.device)
        importlib.import_module(reference.train_model.__module__).NTXentLoss2 = NTXentLoss2
    rna_vae, atac_vae = reference.train_model(
        rna_vae, atac_vae, all_cells_loader, anchor_cells_loader, device=args.device,
        num_epoches=args.num_epochs, lambda_rna_kl=args.lambda_rna_kl, lambda_atac_kl=args.lambda_atac_kl,
        alpha_rna_rec=args.alpha_rna_rec, alpha_atac_rec=args.alpha_atac_rec,
        lambda_contra=args.lambda_contra, temperature=args.temperature,
        anchor_loss_type="contrastive", lr=args.learning_rate, print_step=args.print_step, save_model=False)
    torch.save(rna_vae.state_dict(), output_dir / "rna_vae.pth")
    torch.save(atac_vae.state_dict(), output_dir / "atac_vae.pth")
    rna_emb, atac_emb = reference.model_inference(rna_vae, atac_vae, all_cells_loader, device=args.device)
    pd.DataFrame(rna_emb, index=adata_rna.obs_names).to_csv(output_dir / "rna_embeddings.csv")
    pd.DataFrame(atac_emb, index=adata_atac.obs_names).to_csv(output_dir / "atac_embeddings.csv")
    return 0


# `scmiac train` options: (flag, default, type or None, what it is).  Names and defaults are the
# reference's schema (cli.py:182-208); scripts written for it must keep working.
_TRAIN_OPTIONS = (
    ("--rna-h5ad", None, None, "h5ad file of the RNA cells"),
    ("--atac-h5ad", None, None, "h5ad file of the ATAC cells"),
    ("--output-dir", None, None, "where anchors.csv (and, with scMIAC installed, models and embeddings) go"),
    ("--all-nfeatures", 3000, int, "genes kept for the CCA over both modalities"),
    ("--single-nfeatures", 2000, int, "variable genes ranked per modality"),
    ("--k-anchor", 5, int, "mutual-neighbour depth: a pair is an anchor if each cell is among the other's k nearest"),
    ("--n-components", 30, int, "canonical components searched in"),
    ("--rna-latent-key", "X_pca", None, "obsm key of the RNA input of the VAE"),
    ("--atac-latent-key", "lsi49", None, "obsm key of the ATAC input of the VAE"),
    ("--rna-celltype-key", "cell_type", None, "obs column with the RNA cell types"),
    ("--atac-celltype-key", "pred", None, "obs column with the (predicted) ATAC cell types"),
    ("--batch-size", 256, int, "cells per batch"),
    ("--hidden-dims", [128, 64], _layer_widths, "encoder layer widths, e.g. 128,64"),
    ("--latent-dim", 30, int, "width of the VAE latent space"),
    ("--device", "cuda:0", None, "torch device of the training stages"),
    ("--num-epochs", 2000, int, "epochs"),
    ("--lambda-rna-kl", 1.0, float, "KL weight, RNA"),
    ("--lambda-atac-kl", 1.0, float, "KL weight, ATAC"),
    ("--alpha-rna-rec", 20.0, float, "reconstruction weight, RNA"),
    ("--alpha-atac-rec", 20.0, float, "reconstruction weight, ATAC"),
    ("--lambda-contra", 300.0, float, "weight of the contrastive anchor loss"),
    ("--temperature", 0.5, float, "softmax temperature of the anchor loss"),
    ("--learning-rate", 1e-3, float, "Adam step size"),
    ("--print-step", 10, int, "epochs between log lines"),
)


def build_parser() -> argparse.ArgumentParser:
    top = argparse.ArgumentParser(
        prog="scmiac", description="scMIAC anchor search on MI355X: drop-in for the anchor stage of `scmiac train`")
    commands = top.add_subparsers(dest="command", help="Available commands")
    train = commands.add_parser(
        "train", help="find cell-type-filtered MNN anchors and write anchors.csv (training follows if scMIAC is installed)")
    for flag, default, kind, text in _TRAIN_OPTIONS:
        extra = {} if kind is None else {"type": kind}
        train.add_argument(flag, default=default, help=f"{text} (default: {default})", **extra)
    train.add_argument("--mode", choices=["v"], default=None,
                       help="v: the two files hold the same cells in the same order; pair them by position, no search")
    train.add_argument("--no-balanced-sampler", dest="balanced_sampler", action="store_false", default=True,
                       help="draw anchor batches uniformly instead of weighting by 1 / cell-type frequency")
    # two options the reference does not have (its scripts never pass them; the defaults keep its behaviour)
    train.add_argument("--search-mode", choices=["dense", "pruned", "auto"], default="dense",
                       help="how the exact k-NN search goes through the cells: every query against every cell, "
                            "only the pivot cells that can hold a neighbour (same tables, much faster on clustered data), "
                            "or whichever of the two a short probe search favours")
    train.add_argument("--device-consumer", action="store_true", default=False,
                       help="with scMIAC installed: feed its training loop from device-resident loaders and the fused "
                            "anchor loss of this package instead of its DataLoaders and eager loss")
    train.set_defaults(func=run_train)
    return top


def main(argv: Sequence[str] | None = None) -> int:
    """Entry point of the ``scmiac`` command: 1 (after printing the help) when no command was
    given, otherwise the command's own status."""
    words = list(sys.argv[1:] if argv is None else argv)
    parser = build_parser()
    args = parser.parse_args(words) if words else None
    handler = getattr(args, "func", None)
    if handler is None:
        parser.print_help()
        return 1
    return handler(args)


if __name__ == "__main__":
    raise SystemExit(main())
