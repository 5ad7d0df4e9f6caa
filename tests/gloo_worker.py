"""Worker of tests/test_sharded_gloo.py: one of WORLD_SIZE processes on the CPU
(backend gloo).  The host-side sharding / ring / gather logic of
scmiac_amd.sharded is run with a CPU stand-in for the HIP library (the oracle,
test infrastructure) and compared with the unsharded oracle result."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import oracle  # noqa: E402
from scmiac_amd import _hip, sharded  # noqa: E402


class OracleBackend:
    """Same members as scmiac_amd._hip, on CPU tensors.  Rows in ``flaky`` behave like
    rows whose certificate fails: the search leaves blocks of odd-numbered owners out
    of their result and reports them uncertified."""

    def __init__(self, flaky=(), block_starts=()):
        self.flaky = np.asarray(sorted(flaky), dtype=np.int64)
        self.block_starts = list(block_starts)
        self.exact_calls = 0

    @staticmethod
    def _merge(state, rows, idx, d2):
        k = state.idx.shape[1]
        old_i, old_d = state.idx.numpy()[rows], state.dist2.numpy()[rows]
        all_i = np.concatenate([old_i, idx], axis=1)
        all_d = np.concatenate([old_d, d2], axis=1)
        for out_r, (ii, dd) in zip(rows, zip(all_i, all_d)):
            keep = ii >= 0
            pairs = sorted(set(zip(dd[keep].tolist(), ii[keep].tolist())))[:k]  # (D, index) order, no duplicates
            state.idx[out_r] = -1
            state.dist2[out_r] = float("inf")
            for c, (dv, iv) in enumerate(pairs):
                state.idx[out_r, c] = iv
                state.dist2[out_r, c] = dv

    @staticmethod
    def _block_knn(q, r, k, col_offset):
        n = q.shape[0]
        if n == 0 or r.shape[0] == 0:
            return np.full((n, k), -1, np.int32), np.full((n, k), np.inf, np.float32)
        idx, d2 = oracle.knn_dist2(np.ascontiguousarray(q.numpy()), np.ascontiguousarray(r.numpy()), k)
        return np.where(idx >= 0, idx + col_offset, -1).astype(np.int32), d2

    def knn_search(self, q, r, k, *, col_offset=0, state=None, splits=0, exact_fallback=True):
        n = q.shape[0]
        if state is None:
            state = _hip.KnnState(n, k, torch.device("cpu"))
            state.idx.fill_(-1)
            state.dist2.fill_(float("inf"))
        idx, d2 = self._block_knn(q, r, k, col_offset)
        rows = np.arange(n)
        flaky = self.flaky[self.flaky < n]
        if not exact_fallback and len(flaky) and self.block_starts.index(col_offset) % 2 == 1:
            rows = np.setdiff1d(rows, flaky)
        self._merge(state, rows, idx[rows], d2[rows])
        unc = flaky if not exact_fallback else flaky[:0]
        state.uncertified_rows[:len(unc)] = torch.from_numpy(unc.astype(np.int32))
        state.n_uncertified[0] = len(unc)
        return state

    def knn_exact_rows(self, q, r, k, rows, n_rows, state, *, col_offset=0):
        self.exact_calls += 1
        rows = rows[:int(n_rows.item())].numpy().astype(np.int64)
        if len(rows):
            idx, d2 = self._block_knn(q[rows], r, k, col_offset)
            self._merge(state, rows, idx, d2)
        return state


    def find_mnn(self, g12, g21, ka):
        pairs = np.asarray(oracle.find_mnn(g12.numpy(), g21.numpy(), ka), dtype=np.int64).reshape(-1, 2)
        out = torch.zeros((max(g12.shape[0] * ka, 1), 2), dtype=torch.int64)
        out[:len(pairs)] = torch.from_numpy(pairs)
        return out, torch.tensor([len(pairs)], dtype=torch.int64), torch.zeros(1, dtype=torch.int32)

    def snn_score(self, pairs, n_pairs, g11, g12, g21, g22, ks, *, n_pairs_max=None, want_hist=False):
        n = int(n_pairs.item())
        raw = torch.zeros((max(n_pairs_max or n, 1),), dtype=torch.int32)
        if n:
            cnt = oracle.snn_counts(pairs[:n].numpy(), g11.numpy(), g12.numpy(), g21.numpy(), g22.numpy(), ks)
            raw[:n] = torch.from_numpy(np.asarray(cnt, dtype=np.int32))
        return raw, None

    def closer_count(self, q, r, match=None):
        if q.shape[0] == 0:
            return torch.zeros(0, dtype=torch.int64)
        c = oracle.closer_count(np.ascontiguousarray(q.numpy()), np.ascontiguousarray(r.numpy()),
                                None if match is None else match.numpy())
        return torch.from_numpy(c.astype(np.int64))


def mixture(seed, n, d):
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(77).normal(scale=3.0, size=(6, d))
    return (centres[rng.integers(0, 6, n)] + rng.normal(size=(n, d))).astype(np.float32)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == 0:
        oracle.build()  # one rank compiles the checker if it is missing; the others wait
    dist.barrier()

    # --- shard bounds and the gather, ragged and empty shards
    for n in (7, 1, 2 * world, 0):
        lo, hi = sharded.shard_bounds(n, world, rank)
        covered = torch.zeros(world, 2, dtype=torch.int64)
        dist.all_gather_into_tensor(covered, torch.tensor([[lo, hi]]))
        assert covered[0, 0] == 0 and covered[-1, 1] == n
        assert all(covered[i, 1] == covered[i + 1, 0] for i in range(world - 1))
        local = torch.arange(lo, hi, dtype=torch.int32).reshape(-1, 1).repeat(1, 3)
        full = sharded.gather_rows(local, n)
        assert full.shape == (n, 3) and torch.equal(full[:, 0], torch.arange(n, dtype=torch.int32))

    # --- ring: query rows AND reference rows sharded, uneven blocks
    d, k = 20, 9
    Q, R = mixture(1, 230, d), mixture(2, 470, d)
    R[40:60] = R[400]  # duplicates across blocks: (D, index) ties decide
    r_cuts = [0, 300, 470] if world == 2 else np.linspace(0, 470, world + 1).astype(int).tolist()
    lo, hi = sharded.shard_bounds(len(Q), world, rank)
    idx_o, d2_o = oracle.knn_dist2(Q, R, k)
    for flaky in ((), (0, 3, 17)):
        be = OracleBackend(flaky=flaky if rank == 0 else (), block_starts=r_cuts[:-1])
        st = sharded.knn_ring(torch.from_numpy(Q[lo:hi]), torch.from_numpy(R[r_cuts[rank]:r_cuts[rank + 1]]), k,
                              backend=be)
        np.testing.assert_array_equal(st.idx.numpy(), idx_o[lo:hi])
        np.testing.assert_array_equal(st.dist2.numpy(), d2_o[lo:hi])
        # the second trip turns the ring on every rank iff some rank has uncertified rows, but only
        # the ranks that have such rows evaluate anything
        assert be.exact_calls == (world if (flaky and rank == 0) else 0), (rank, be.exact_calls)
        full = sharded.gather_rows(st.idx, len(Q))
        np.testing.assert_array_equal(full.numpy(), idx_o)

    # --- four tables, query rows sharded, references replicated
    U, V = mixture(3, 301, d), mixture(4, 257, d)
    t = sharded.find_neighbor_sharded(torch.from_numpy(U), torch.from_numpy(V), 12, backend=OracleBackend())
    for got, want in zip(t.numpy(), oracle.find_neighbor(U, V, 12)):
        np.testing.assert_array_equal(got, want)

    # --- the same with more neighbours than one 64-neighbour search returns (k_anchor = 70 makes
    # find_neighbor's k 70): every table is then the group-wise merge, shard by shard
    t = sharded.find_neighbor_sharded(torch.from_numpy(U), torch.from_numpy(V), 70, backend=OracleBackend())
    for got, want in zip(t.numpy(), oracle.find_neighbor(U, V, 70)):
        np.testing.assert_array_equal(got, want)

    # --- the whole search sharded by the rows of the first set: only G21 / G22 are gathered, the
    # anchors of all ranks in rank order are the unsharded (i, j) order
    cases = [(301, 257, 12), (7, 40, 3)] + ([(world - 1, 33, 1)] if world >= 3 else [])   # ragged, tiny, an empty shard
    for nu, nv, kq in cases:
        Us, Vs = U[:nu], V[:nv]
        pairs, raw = sharded.anchors_sharded(torch.from_numpy(Us), torch.from_numpy(Vs), kq, k_anchor=min(5, kq),
                                             k_score=30, backend=OracleBackend())
        G11, G12, G21, G22 = oracle.find_neighbor(Us, Vs, kq)
        want = np.asarray(oracle.find_mnn(G12, G21, min(5, kq)), dtype=np.int64).reshape(-1, 2)
        np.testing.assert_array_equal(pairs.numpy(), want)
        if len(want):
            np.testing.assert_array_equal(raw.numpy(), np.asarray(oracle.snn_counts(want, G11, G12, G21, G22, min(30, kq)),
                                                                 dtype=np.int32))

    # --- closer-than-match counts, query rows sharded (ragged last shard)
    X = mixture(5, 203, d)
    Y = (X + np.random.default_rng(6).normal(size=X.shape)).astype(np.float32)
    fx, fy = sharded.foscttm_sharded(torch.from_numpy(X), torch.from_numpy(Y), backend=OracleBackend())
    want_x, want_y = oracle.foscttm(X, Y)
    np.testing.assert_array_equal(fx, want_x)
    np.testing.assert_array_equal(fy, want_y)

    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
