"""Times the anchor table's consumer: the fused symmetric NT-Xent loss against the
eager formulation of the reference (scmiac/modeling/loss.py:108-132) on the same GPU,
and the device-resident anchor loader against per-sample ``DataFrame.iloc`` access as
in the reference's ``AnchorCellsDataset.__getitem__`` (dataset.py:85-96; one process,
no DataLoader workers).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scmiac_amd.modeling.dataset import DeviceAnchorLoader  # noqa: E402
from scmiac_amd.modeling.loss import ntxent2  # noqa: E402


def eager_loss(z_i, z_j, T):
    z_i = torch.nn.functional.normalize(z_i, dim=1)
    z_j = torch.nn.functional.normalize(z_j, dim=1)
    out = 0.0
    for a, b in ((z_i, z_j), (z_j, z_i)):
        S = torch.mm(a, b.t()) / T
        pos = torch.diag(S)
        m, _ = torch.max(S, dim=1, keepdim=True)
        lse = m.squeeze() + torch.log(torch.exp(S - m).sum(dim=1))
        out = out + (-pos + lse).mean()
    return 0.5 * out


def time_step(fn, zi, zj, reps=200):
    for _ in range(20):
        zi.grad = zj.grad = None
        fn(zi, zj, 0.5).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        zi.grad = zj.grad = None
        fn(zi, zj, 0.5).backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    out = {}
    for B, d in ((256, 30), (1024, 30), (4096, 50)):
        zi = torch.randn(B, d, device="cuda", requires_grad=True)
        zj = torch.randn(B, d, device="cuda", requires_grad=True)
        out[f"loss_fwd_bwd_us_B{B}_d{d}"] = {"fused": round(time_step(ntxent2, zi, zj), 1),
                                             "eager": round(time_step(eager_loss, zi, zj), 1)}
    rng = np.random.default_rng(0)
    n_cells, n_anchor, d = 100000, 200000, 50
    rna, atac = rng.normal(size=(n_cells, d)), rng.normal(size=(n_cells, d))
    df = pd.DataFrame({"x1": rng.integers(0, n_cells, n_anchor), "x2": rng.integers(0, n_cells, n_anchor),
                       "x1_ct": rng.choice([f"t{i}" for i in range(20)], n_anchor)})
    loader = DeviceAnchorLoader(rna, atac, df, batch_size=256, balanced=True, seed=0)
    for _ in loader:
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch in loader:
        n += batch["rna_anchor"].shape[0]
    torch.cuda.synchronize()
    out["device_loader_samples_per_s"] = round(n / (time.perf_counter() - t0))
    # the reference's per-sample access pattern, 20 batches of 256
    rna32, atac32 = rna.astype(np.float32), atac.astype(np.float32)
    frame = df.reset_index(drop=True)
    t0 = time.perf_counter()
    for idx in rng.integers(0, n_anchor, 20 * 256):
        r = frame.iloc[idx]["x1"]
        a = frame.iloc[idx]["x2"]
        _ = (torch.tensor(rna32[r]), torch.tensor(atac32[a]), frame.iloc[idx]["x1_ct"])
    out["iloc_dataset_samples_per_s_one_process"] = round(20 * 256 / (time.perf_counter() - t0))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
