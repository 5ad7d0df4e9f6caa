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


def require_argument(value, name: str) -> None:
    if value is None:
        raise SystemExit(f"Missing required argument: {name}. Provide it via CLI or YAML config.")


def parse_hidden_dims(value: str) -> list:
    parts = [item.strip() for item in value.split(",") if item.strip()]
    if not parts:
        raise argparse.ArgumentTypeError("hidden_dims must contain at least one integer")
    try:
        return [int(item) for item in parts]
    except ValueError as exc:
        raise argparse.ArgumentTypeError("hidden_dims must be a comma-separated list of integers") from exc


def _read_h5ad(path):
    try:
        import anndata
    except ImportError as exc:  # pragma: no cover - depends on the environment
        raise SystemExit(f"Reading {path} needs the 'anndata' package: {exc}")
    return anndata.read_h5ad(path)


def write_anchors(anchor_df, output_dir) -> Path:
    """``anchors.csv`` as the reference writes it: header x1,x2,score,x1_ct,x2_ct,is_same, no index."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    anchor_path = output_dir / "anchors.csv"
    anchor_df.to_csv(anchor_path, index=False)
    return anchor_path


def run_train(args: argparse.Namespace) -> int:
    require_argument(args.rna_h5ad, "--rna-h5ad")
    require_argument(args.atac_h5ad, "--atac-h5ad")
    require_argument(args.output_dir, "--output-dir")

    from .modeling.scmiac import find_anchors

    adata_rna = _read_h5ad(args.rna_h5ad)
    adata_atac = _read_h5ad(args.atac_h5ad)

    print("Generating MNN anchors...")
    anchor_df = find_anchors(
        adata_rna, adata_atac,
        all_nfeatures=args.all_nfeatures, single_nfeatures=args.single_nfeatures,
        k_anchor=args.k_anchor, n_components=args.n_components,
        ct_filter=True,  # the command always filters by cell type, like the reference's
        mode=args.mode,
        rna_celltype_key=args.rna_celltype_key, atac_celltype_key=args.atac_celltype_key,
    )
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


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(
        prog="scmiac",
        description="scMIAC anchor search on MI355X (drop-in for the anchor stage of `scmiac train`)")
    subparsers = parser.add_subparsers(dest="command", help="Available commands")
    p = subparsers.add_parser("train", help="Generate cell-type-filtered MNN anchors (then train, if scMIAC is installed)")
    p.add_argument("--rna-h5ad", default=None, help="Path to RNA AnnData file")
    p.add_argument("--atac-h5ad", default=None, help="Path to ATAC AnnData file")
    p.add_argument("--output-dir", default=None, help="Output directory (anchors.csv is written here)")
    p.add_argument("--all-nfeatures", type=int, default=3000, help="Number of features for integration when generating anchors")
    p.add_argument("--single-nfeatures", type=int, default=2000, help="Number of features per modality when generating anchors")
    p.add_argument("--k-anchor", type=int, default=5, help="Number of neighbors when selecting anchors")
    p.add_argument("--n-components", type=int, default=30, help="Dimensionality for CCA during anchor finding")
    p.add_argument("--mode", choices=["v"], default=None, help="'v': positional anchors for paired data")
    p.add_argument("--rna-latent-key", default="X_pca", help="Key for RNA latent representation in AnnData")
    p.add_argument("--atac-latent-key", default="lsi49", help="Key for ATAC latent representation in AnnData")
    p.add_argument("--rna-celltype-key", default="cell_type", help="Key for RNA cell type annotation in AnnData.obs")
    p.add_argument("--atac-celltype-key", default="pred", help="Key for ATAC cell type annotation in AnnData.obs")
    p.add_argument("--batch-size", type=int, default=256, help="Batch size for DataLoaders")
    p.add_argument("--hidden-dims", type=parse_hidden_dims, default=[128, 64], help="Comma-separated hidden dimensions for VAE")
    p.add_argument("--latent-dim", type=int, default=30, help="Latent dimension size for VAE")
    p.add_argument("--no-balanced-sampler", dest="balanced_sampler", action="store_false", help="Disable balanced anchor sampling")
    p.add_argument("--device", default="cuda:0", help="Torch device to use")
    p.add_argument("--num-epochs", type=int, default=2000, help="Training epochs")
    p.add_argument("--lambda-rna-kl", type=float, default=1.0, help="Weight for RNA KL divergence")
    p.add_argument("--lambda-atac-kl", type=float, default=1.0, help="Weight for ATAC KL divergence")
    p.add_argument("--alpha-rna-rec", type=float, default=20.0, help="Weight for RNA reconstruction loss")
    p.add_argument("--alpha-atac-rec", type=float, default=20.0, help="Weight for ATAC reconstruction loss")
    p.add_argument("--lambda-contra", type=float, default=300.0, help="Weight for anchor alignment loss")
    p.add_argument("--temperature", type=float, default=0.5, help="Temperature for contrastive loss")
    p.add_argument("--learning-rate", type=float, default=1e-3, help="Learning rate")
    p.add_argument("--print-step", type=int, default=10, help="Logging interval in epochs")
    p.set_defaults(func=run_train, balanced_sampler=True)
    return parser


def main(argv: Sequence[str] | None = None) -> int:
    parser = build_parser()
    if argv is None:
        argv = sys.argv[1:]
    if not argv:
        parser.print_help()
        return 1
    args = parser.parse_args(argv)
    if not hasattr(args, "func"):
        parser.print_help()
        return 1
    return args.func(args)


if __name__ == "__main__":
    raise SystemExit(main())
