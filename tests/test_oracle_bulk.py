"""The threaded bulk form of the oracle (orc_knn_table: row-wise accumulation, one user per thread) against the
per-pair closures it restates (orc_pipeline: cos_value / getNeighbors / weightedSumDeviation literally).  The bulk
form is what makes full-coverage parity affordable at the ml-25m shape (tests/test_gpu_parity.py) and the all-cores
cpu_baseline of bench.py; it is only trusted because it is pinned here, bit for bit, to the literal form."""
import numpy as np
import pytest

from tests.test_oracle_semantics import _cols, _no_zero_scale, _random_case


def _check(oracle, tr, te, k, users=None, threads=3):
    m = oracle.Model(*tr)
    t = m.knn_table(k, users=users, threads=threads)
    p = m.pipeline(oracle.SIM_COSINE, k)
    assert t.width == min(k, m.num_users - 1)
    for r, u in enumerate(t.row_user):
        ids, sims = p.neighbors(int(u))
        assert t.ids[r].tolist() == ids.tolist(), f"user {u}"
        assert t.sims[r].tolist() == sims.tolist(), f"user {u}"
    if users is None:
        want, preds = p.mae(*te, True)
        got, gp = t.mae(*te)
        np.testing.assert_array_equal(gp, preds)
        assert got == want  # the same left fold
    return t


def test_bulk_equals_per_pair_on_the_ml100k_shape(oracle, syn100k):
    d = syn100k
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    for k in (10, 300, 2000):
        _check(oracle, tr, te, k)
    # a subset of users, in a caller-chosen order
    sub = np.unique(d.train.users)[::50][::-1]
    t = _check(oracle, tr, te, 25, users=sub)
    assert t.row_user.tolist() == sub.tolist()


@pytest.mark.parametrize("seed", range(6))
def test_bulk_equals_per_pair_on_random_cases(oracle, seed):
    rng = np.random.default_rng(500 + seed)
    rows = _random_case(rng, n_users=12 + 2 * seed, n_items=15, n_ratings=120 + 10 * seed, half=(seed % 2 == 1), tiny_rows=0)
    cut = len(rows) * 4 // 5
    train, test = rows[:cut], rows[cut:]
    if not _no_zero_scale(train):
        pytest.skip("scale() == 0 corner")
    tr = _cols(train)
    if min(np.bincount(np.unique(tr[0], return_inverse=True)[1])) <= 4:
        with pytest.raises(oracle.OracleError):
            oracle.Model(*tr).knn_table(3)
        return
    test += [(999_999, train[0][1], 3.0), (train[0][0], 888_888, 4.0)]
    te = _cols(test)
    for k in (1, 5, 100):
        _check(oracle, tr, te, k, threads=2)


def test_bulk_refuses_memo_dependent_inputs(oracle):
    """a user with <= 4 ratings makes values depend on the closures' evaluation history (N6)"""
    users = [1] * 6 + [2] * 6 + [3] * 2
    items = list(range(6)) + list(range(6)) + [0, 1]
    ratings = [1.0, 2, 3, 4, 5, 3, 2, 2, 4, 4, 5, 1, 3, 4]
    m = oracle.Model(users, items, ratings)
    with pytest.raises(oracle.OracleError):
        m.knn_table(2)
