"""Worker of tests/test_gpu_shared_ranks.py: one of WORLD_SIZE processes that all use GPU 0.

RCCL refuses two ranks on one device, so the transport here is gloo on host copies of the
tensors (see ``stage_collectives_through_host``); everything else is the multi-GPU product path: the real HIP
kernels behind scmiac_amd._hip, query rows sharded, reference blocks passed round the ring
and merged with ``col_offset`` / ``merge``, the sharded anchor tail, the sharded counts.
Results are compared with the unsharded CPU oracle (test infrastructure)."""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import oracle  # noqa: E402
from scmiac_amd import _hip, sharded  # noqa: E402


def mixture(seed, n, d, unit=False):
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(77).normal(scale=3.0, size=(12, d))
    x = (centres[rng.integers(0, 12, n)] + rng.normal(size=(n, d))).astype(np.float32)
    if unit:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


class _Landed:
    """Stand-in for the work handle of a collective that has already completed."""

    def __init__(self, then=None):
        self._then = then

    def wait(self):
        if self._then is not None:
            self._then()
            self._then = None
        return True


def stage_collectives_through_host():
    """The collectives scmiac_amd.sharded uses, on host copies of device tensors: gloo moves
    host memory, the result is copied back on the current stream."""
    real_gather, real_reduce, real_batch = dist.all_gather_into_tensor, dist.all_reduce, dist.batch_isend_irecv

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        host = torch.empty(out.shape, dtype=out.dtype)
        real_gather(host, inp.cpu(), group=group)
        out.copy_(host)
        return _Landed() if async_op else None

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        host = t.cpu()
        real_reduce(host, op=op, group=group)
        t.copy_(host)
        return _Landed() if async_op else None

    def batch_isend_irecv(ops):
        staged, landings = [], []
        for o in ops:
            if o.op is dist.isend:
                staged.append(dist.P2POp(dist.isend, o.tensor.cpu(), o.peer, o.group))
            else:
                host = torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                staged.append(dist.P2POp(dist.irecv, host, o.peer, o.group))
                landings.append((o.tensor, host))
        reqs = real_batch(staged)

        def land():
            for r in reqs:
                r.wait()
            for dst, host in landings:
                dst.copy_(host)

        return [_Landed(land)]

    dist.all_gather_into_tensor, dist.all_reduce, dist.batch_isend_irecv = all_gather_into_tensor, all_reduce, batch_isend_irecv


def main():
    dist.init_process_group("gloo")
    stage_collectives_through_host()
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if rank == 0:
        oracle.build()  # one rank compiles the checker if it is missing; the others wait
    dist.barrier()

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def note(what):
        print(f"rank {rank}: {what}", file=sys.stderr, flush=True)

    # --- ring: query rows and reference rows sharded, uneven blocks, unit rows, d=64; large enough
    # for the locality partition and merge-mode threshold seeds to be at work in every step
    d, k = 64, 30
    Q, R = mixture(11, 6000, d, unit=True), mixture(12, 70_000, d, unit=True)
    R[40:60] = R[65_000]  # duplicates across blocks: (D, index) ties decide
    cuts = np.linspace(0, len(R), world + 1).astype(int)
    cuts[1] += 1234  # uneven blocks
    lo, hi = sharded.shard_bounds(len(Q), world, rank)
    idx_o, d2_o = oracle.knn_dist2(Q, R, k)
    note('ring')
    st = sharded.knn_ring(up(Q[lo:hi]), up(R[cuts[rank]:cuts[rank + 1]]), k)
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o[lo:hi])
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o[lo:hi])
    full = sharded.gather_rows_async(st.idx, len(Q)).wait()
    np.testing.assert_array_equal(full.cpu().numpy(), idx_o)

    # --- four tables, query rows sharded, references replicated (gathers in flight while searching)
    d, k = 50, 30
    U, V = mixture(13, 20_001, d), mixture(14, 17_003, d)
    note('four tables')
    t = sharded.find_neighbor_sharded(up(U), up(V), k)
    want = oracle.find_neighbor(U, V, k)
    for got, w in zip(t.numpy(), want):
        np.testing.assert_array_equal(got, w)

    # --- the whole search sharded by the rows of the first set
    note('sharded anchors')
    pairs, raw = sharded.anchors_sharded(up(U), up(V), k, k_anchor=5, k_score=30)
    G11, G12, G21, G22 = want
    want_pairs = np.asarray(oracle.find_mnn(G12, G21, 5), dtype=np.int64).reshape(-1, 2)
    np.testing.assert_array_equal(pairs.cpu().numpy(), want_pairs)
    np.testing.assert_array_equal(raw.cpu().numpy(),
                                  np.asarray(oracle.snn_counts(want_pairs, G11, G12, G21, G22, 30), dtype=np.int32))

    # --- the drop-in API with the searches shared out over the ranks (opt-in): numpy in, numpy /
    # pandas out on every rank, equal to the single-process oracle
    from scmiac_amd.preprocessing import anchors as drop_in

    sharded.use_process_group()
    note('drop-in API over the ranks')
    for got, w in zip(drop_in.find_neighbor(U, V, k), want):
        np.testing.assert_array_equal(got, w)

    class _Ad:  # the handful of AnnData members the precomputed branch touches
        def __init__(self, x):
            self.obsm, self.shape = {"lat": x}, (x.shape[0], 1)

    integ = drop_in.SeuratIntegration()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        integ.find_anchor([_Ad(U.astype(np.float64)), _Ad(V.astype(np.float64))], key_anchor="lat", dim_red="precomputed",
                          k_anchor=5, k_score=30)
    odf, _, _ = oracle.pairwise_anchors(U, V, k_anchor=5, k_score=30)
    got_df = integ.anchor[(0, 1)]
    np.testing.assert_array_equal(got_df["x1"].values, odf["x1"].values)
    np.testing.assert_array_equal(got_df["x2"].values, odf["x2"].values)
    np.testing.assert_array_equal(got_df["score"].values, odf["score"].values)
    sharded.use_process_group(enable=False)

    # --- closer-than-match counts, query rows sharded
    X = mixture(15, 5003, 30)
    Y = (X + np.random.default_rng(16).normal(size=X.shape)).astype(np.float32)
    note('counts')
    fx, fy = sharded.foscttm_sharded(up(X), up(Y))
    want_x, want_y = oracle.foscttm(X, Y)
    np.testing.assert_array_equal(fx, want_x)
    np.testing.assert_array_equal(fy, want_y)

    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok ({len(want_pairs)} anchors)", flush=True)


if __name__ == "__main__":
    main()
