"""N > 1 host path on CPU: world_size-2 (and 3) gloo runs of movie-recommender-system_amd/sharded.py.

The sharding logic (user block partition, padded all-gather of the owned segments, all-reduce of
the MAE partial sums) is backend-agnostic; here the per-rank engine is a test double backed by the
CPU oracle, so the multi-process plumbing is exercised without a GPU.  The GPU engine implements the
same four calls (fit_device / shard_tensors / shard_commit / mae_device)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "movie-recommender-system_amd"


class OracleShardEngine:
    """Per-rank double: owns a block of dense users, computes only their K2/K3 outputs."""

    def __init__(self, oracle, rank, world, k):
        self.O, self.rank, self.world, self.k = oracle, rank, world, k

    def fit_device(self, users, items, ratings):
        import torch

        O = self.O
        self.u, self.i, self.r = users.numpy(), items.numpy(), ratings.numpy()
        self.model = O.Model(self.u, self.i, self.r)
        order = self.model.user_iteration_order()  # dense user index == HashSet iteration rank (N3)
        dense_of = {int(x): d for d, x in enumerate(order)}
        du = np.array([dense_of[int(x)] for x in self.u])
        ikey = np.array([O.trie_key(O.improve(int(x))) for x in self.i], dtype=np.int64)
        self.perm = np.lexsort((ikey, du))  # user-major, items in trie order inside a user
        self.du_sorted = du[self.perm]
        U = len(order)
        sharded = importlib.import_module(PKG + ".sharded")
        self.lo, self.hi = sharded.user_block(U, self.rank, self.world)
        ptr = np.searchsorted(self.du_sorted, np.arange(U + 1))
        self.nlo, self.nhi = int(ptr[self.lo]), int(ptr[self.hi])
        dev = self.model.normalized_deviations()[self.perm]
        pre = self.model.preprocessed()[self.perm]
        avg = np.array([self.model.users_avg(int(x)) for x in order])
        norm = np.array([self.model.user_weight(int(x)) for x in order])
        self.full = {"user_avg": avg, "user_norm": norm, "dev": dev, "pre": pre}

        def owned(a, lo, hi):
            out = np.zeros_like(a)
            out[lo:hi] = a[lo:hi]
            return torch.from_numpy(out)

        self.t = {"user_range": (self.lo, self.hi), "nnz_range": (self.nlo, self.nhi),
                  "user_avg": owned(avg, self.lo, self.hi), "user_norm": owned(norm, self.lo, self.hi),
                  "dev": owned(dev, self.nlo, self.nhi), "pre": owned(pre, self.nlo, self.nhi)}
        self.order = order
        self.committed = False

    def shard_tensors(self):
        return self.t

    def shard_commit(self):
        for k in ("user_avg", "user_norm"):  # the exchange must have filled every user's mean and norm
            np.testing.assert_array_equal(self.t[k].numpy(), self.full[k])
        # knncf_shard_commit recomputes the other users' deviations / preprocessed ratings from them (prep_complete_rows)
        own = np.zeros(len(self.full["dev"]), dtype=bool)
        own[self.nlo:self.nhi] = True
        for k in ("dev", "pre"):
            assert not self.t[k].numpy()[~own].any()   # nothing of them travelled
            self.t[k].numpy()[~own] = self.full[k][~own]
        self.committed = True

    def mae_device(self, predictor, users, items, ratings):
        assert self.committed
        tu, ti, tr = users.numpy(), items.numpy(), ratings.numpy()
        mine = set(int(x) for x in self.order[self.lo:self.hi])
        known = set(int(x) for x in self.order)
        own = np.array([(int(x) in mine) or (self.rank == 0 and int(x) not in known) for x in tu])
        pipe = self.model.pipeline(self.O.SIM_COSINE, self.k)
        # same closure history as the single-process run for the users this rank owns
        _, preds = pipe.mae(tu[own], ti[own], tr[own], True)
        return float(np.abs(tr[own] - preds).sum()), int(own.sum())


class PoisonedEngine(OracleShardEngine):
    """fails on one rank only, the way a scale() == 0 user owned by that rank makes knncf_fit_device fail there"""

    class Boom(RuntimeError):
        status = -2  # KNNCF_E_NONFINITE

    def __init__(self, oracle, rank, world, k, bad_rank, where="fit"):
        super().__init__(oracle, rank, world, k)
        self.bad_rank, self.where = bad_rank, where

    def fit_device(self, users, items, ratings):
        if self.rank == self.bad_rank and self.where == "fit":
            raise PoisonedEngine.Boom("non-finite normalized deviation")
        super().fit_device(users, items, ratings)

    def mae_device(self, predictor, users, items, ratings):
        if self.rank == self.bad_rank and self.where == "mae":  # e.g. KNNCF_E_NOMEM in the lazy neighbour build
            raise PoisonedEngine.Boom("out of memory in the neighbour build")
        return super().mae_device(predictor, users, items, ratings)


def _poisoned_worker(rank, world, port, out, where="fit"):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import knncf_oracle as O

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    synth = importlib.import_module(PKG + ".synth")
    sharded = importlib.import_module(PKG + ".sharded")
    d = synth.syn_scaled(60, 60, 1500, seed=3, half_stars=True)
    tr = tuple(torch.from_numpy(a) for a in (d.train.users, d.train.items, d.train.ratings))
    model = sharded.ShardedKnn(PoisonedEngine(O, rank, world, 5, bad_rank=1, where=where), dist, rank, world)
    try:
        model.fit(*tr)
        if where == "mae":
            te = tuple(torch.from_numpy(a) for a in (d.test.users, d.test.items, d.test.ratings))
            model.mae(5, *te)
        out.put((rank, "no error", 0))
    except PoisonedEngine.Boom as e:
        out.put((rank, "own", e.status))
    except sharded.ShardFitError as e:
        out.put((rank, "other", e.status))
    dist.barrier()  # nobody is stuck in the all-gather
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ["fit", "mae"])
def test_failure_on_one_rank_raises_on_every_rank(where):
    """a rank whose fit — or whose lazy neighbour build inside mae — fails: every rank raises, none blocks in a collective"""
    import torch.multiprocessing as mp

    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_poisoned_worker, args=(r, world, port, out, where)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict((r, (kind, st)) for r, kind, st in (out.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[1] == ("own", -2)
    assert results[0] == ("other", -2) and results[2] == ("other", -2)


def _worker(rank, world, port, k, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import knncf_oracle as O

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    synth = importlib.import_module(PKG + ".synth")
    sharded = importlib.import_module(PKG + ".sharded")
    d = synth.syn_scaled(97, 60, 2400, seed=7 + world, half_stars=True, shuffle=True)
    tr = tuple(torch.from_numpy(a) for a in (d.train.users, d.train.items, d.train.ratings))
    te_u = np.concatenate([d.test.users, [424242]]).astype(np.int32)  # + a user absent from train
    te_i = np.concatenate([d.test.items, d.test.items[:1]]).astype(np.int32)
    te_r = np.concatenate([d.test.ratings, [3.5]])
    te = tuple(torch.from_numpy(a) for a in (te_u, te_i, te_r))
    model = sharded.ShardedKnn(OracleShardEngine(O, rank, world, k), dist if world > 1 else None, rank, world)
    model.fit(*tr)
    mae, count = model.mae(5, *te)
    want = O.Model(d.train.users, d.train.items, d.train.ratings).pipeline(O.SIM_COSINE, k).mae(te_u, te_i, te_r)
    out.put((rank, mae, count, want, len(te_u)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_host_path_matches_single_process(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 7, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mae, count, want, n in results:
        assert count == n                      # every test row is owned by exactly one rank
        assert mae == pytest.approx(want, abs=1e-12)  # only the final all-reduce order differs
    assert len({round(m, 15) for _, m, _, _, _ in results}) == 1  # every rank holds the same MAE


def test_user_block_partition(pkg):
    sharded = importlib.import_module(pkg.__name__ + ".sharded")
    for U in (1, 7, 943, 162_541):
        for world in (1, 2, 3, 8):
            blocks = [sharded.user_block(U, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == U
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(h - l for l, h in blocks) <= -(-U // world)
