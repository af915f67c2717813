"""The C oracle against (i) the hand-derived micro-fixture of SURVEY.md §4 (spec Eqs. 2-10),
(ii) the literal Python model of the Scala closures (tests/scala_model.py), (iii) structural
invariants of the reference's math."""
from fractions import Fraction as F

import numpy as np
import pytest

from tests import scala_model as sm

TRAIN = [(1, 1, 5.0), (1, 2, 3.0), (1, 3, 1.0), (2, 1, 4.0), (2, 2, 2.0), (3, 2, 5.0), (3, 3, 4.0)]
TEST = [(2, 3, 2.0), (3, 1, 5.0)]


def _cols(rows):
    u, i, r = zip(*rows)
    return list(u), list(i), list(r)


def test_micro_fixture_hand_derived(oracle):
    m = oracle.Model(*_cols(TRAIN))
    assert m.average() == pytest.approx(24 / 7, rel=1e-15)
    assert [m.users_avg(u) for u in (1, 2, 3)] == [3.0, 3.0, 4.5]
    np.testing.assert_allclose(m.normalized_deviations(), [1, 0, -1, 0.5, -0.5, 1, -1 / 7], rtol=1e-15)
    # adjusted cosine: s12 = 0.5, s13 = 0.1, s23 = -0.7
    for (a, b), want in {(1, 2): 0.5, (1, 3): 0.1, (2, 3): -0.7}.items():
        assert m.fresh_similarity(oracle.SIM_COSINE, a, b) == pytest.approx(want, rel=1e-12)
        assert m.fresh_similarity(oracle.SIM_COSINE, b, a) == pytest.approx(want, rel=1e-12)
    tu, ti, tr = _cols(TEST)
    # k = 1: N(1)={2}, N(2)={1}, N(3)={1}; predictions 1.0 and 5.0; MAE 0.5
    p = m.pipeline(oracle.SIM_COSINE, 1)
    assert [p.neighbors(u)[0].tolist() for u in (1, 2, 3)] == [[2], [1], [1]]
    mae, preds = p.mae(tu, ti, tr, True)
    np.testing.assert_allclose(preds, [1.0, 5.0], rtol=1e-12)
    assert mae == pytest.approx(0.5, rel=1e-12)
    # k = 2: predictions 7/3 and 3.40625; MAE 0.963541666...
    mae, preds = m.pipeline(oracle.SIM_COSINE, 2).mae(tu, ti, tr, True)
    np.testing.assert_allclose(preds, [7 / 3, 3.40625], rtol=1e-12)
    assert mae == pytest.approx(float((F(7, 3) - 2 + 5 - F(109, 32)) / 2), rel=1e-12)
    # baseline: item deviations (0.75, 1/6, -4/7); predictions 13/7 and 4.875
    np.testing.assert_allclose([m.items_avg_dev(i) for i in (1, 2, 3)], [0.75, 1 / 6, -4 / 7], rtol=1e-12)
    mae, preds = m.mae(oracle.KIND_BASELINE, tu, ti, tr, True)
    np.testing.assert_allclose(preds, [13 / 7, 4.875], rtol=1e-12)
    assert mae == pytest.approx(0.13392857142857142, rel=1e-12)


def _random_case(rng, n_users, n_items, n_ratings, half=False, tiny_rows=0):
    pairs = set()
    rows = []
    while len(rows) < n_ratings:
        u, i = int(rng.integers(1, n_users + 1)), int(rng.integers(1, n_items + 1))
        if (u, i) in pairs:
            continue
        pairs.add((u, i))
        r = float(rng.integers(1, 11)) / 2 if half else float(rng.integers(1, 6))
        rows.append((u * 7 + 3, i * 13 + 1, r))  # sparse raw ids
    for t in range(tiny_rows):  # users with 1..4 ratings (Set1..Set4 order matters, N2)
        u = 10_000 + t
        for i in rng.choice(n_items, size=int(rng.integers(1, 5)), replace=False):
            rows.append((u, int(i + 1) * 13 + 1, float(rng.integers(1, 6))))
    rng.shuffle(rows)
    return rows


def _no_zero_scale(rows):
    m = sm.users_avg(rows)
    return all(sm.scale(r, m[u]) != 0 for (u, _, r) in rows)


@pytest.mark.parametrize("seed", range(6))
def test_oracle_equals_literal_model_bitwise(oracle, seed):
    rng = np.random.default_rng(seed)
    rows = _random_case(rng, n_users=12 + seed, n_items=14, n_ratings=70 + 5 * seed,
                        half=(seed % 2 == 1), tiny_rows=seed % 3)
    cut = len(rows) * 4 // 5
    train, test = rows[:cut], rows[cut:]
    if not _no_zero_scale(train):
        pytest.skip("scale() == 0 corner")
    test += [(999_999, train[0][1], 3.0), (train[0][0], 888_888, 4.0)]  # unseen user / unseen item
    m = oracle.Model(*_cols(train))
    tu, ti, tr = _cols(test)

    assert m.average() == sm.average(train)
    ua, ia = sm.users_avg(train), sm.items_avg(train)
    assert all(m.users_avg(u) == v for u, v in ua.items())
    assert all(m.items_avg(i) == v for i, v in ia.items())
    nd = sm.compute_normalize_deviation(train)
    assert m.normalized_deviations().tolist() == [nd.d[(u, i)] for (u, i, _) in train]
    pre = sm.preprocessed_rating(train)
    assert m.preprocessed().tolist() == [pre[(u, i)] for (u, i, _) in train]
    dev = sm.items_avg_dev(train)
    assert all(m.items_avg_dev(i) == v for i, v in dev.items())

    base = sm.compute_prediction(train)
    assert m.mae(oracle.KIND_BASELINE, tu, ti, tr) == sm.mae(base, test)
    assert m.mae(oracle.KIND_GLOBAL, tu, ti, tr) == sm.mae(lambda u, i: sm.average(train), test)
    g = sm.average(train)
    assert m.mae(oracle.KIND_USER, tu, ti, tr) == sm.mae(lambda u, i: ua.get(u, g), test)
    assert m.mae(oracle.KIND_ITEM, tu, ti, tr) == sm.mae(lambda u, i: ia.get(i, g), test)

    users = sorted(set(u for (u, _, _) in train))
    for k in (1, 3, len(users) + 5):
        cos = sm.adjusted_cosine_similarity_function(train)
        nn = sm.get_neighbors(train, k, cos)
        p = m.pipeline(oracle.SIM_COSINE, k)
        # same evaluation history on both sides: MAE over the test set first (predict/kNN.scala:42-45)
        want = sm.mae(sm.predictor(train, sm.weighted_sum_deviation(train, sm.get_similarity(train, k, cos))), test)
        got, preds = p.mae(tu, ti, tr, True)
        assert got == want
        # neighbour lists: ids and similarities, bit for bit, in order (fresh closures, same query order)
        p2 = m.pipeline(oracle.SIM_COSINE, k)
        for u in users[::2] + users[1::2]:
            ids, sims = p2.neighbors(u)
            ref = nn(u)
            assert ids.tolist() == [x for x, _ in ref]
            assert sims.tolist() == [s for _, s in ref]

    # Personalized (no k): similarity one / adjusted cosine / jaccard (predict/Personalized.scala:61-72)
    for kind, f in ((oracle.SIM_ONE, sm.similarity_one()),
                    (oracle.SIM_COSINE, sm.adjusted_cosine_similarity_function(train)),
                    (oracle.SIM_JACCARD, sm.jaccard_coefficient(train))):
        want = sm.mae(sm.predictor(train, sm.weighted_sum_deviation(train, f)), test)
        assert m.pipeline(kind, -1).mae(tu, ti, tr) == want


@pytest.mark.parametrize("seed", range(4))
def test_recommendations_equal_literal_model(oracle, seed):
    """recommendations :651-674: unrated items by (prediction desc, id asc), first n — kNN and closed-form predictors,
    a known user, a user without ratings in train, n larger than the number of unrated items, ties"""
    rng = np.random.default_rng(100 + seed)
    rows = _random_case(rng, n_users=10 + seed, n_items=16, n_ratings=60 + 6 * seed, half=(seed % 2 == 0), tiny_rows=seed % 2)
    if not _no_zero_scale(rows):
        pytest.skip("scale() == 0 corner")
    m = oracle.Model(*_cols(rows))
    users = sorted(set(u for (u, _, _) in rows))
    k = 3
    cos = sm.adjusted_cosine_similarity_function(rows)
    knn = sm.recommendations(rows, sm.predictor(rows, sm.weighted_sum_deviation(rows, sm.get_similarity(rows, k, cos))))
    p = m.pipeline(oracle.SIM_COSINE, k)
    for u in users[:4] + [424242]:
        for n in (1, 3, 100):
            want = knn(u, n)
            ids, preds = p.recommend(u, n)
            assert ids.tolist() == [x for x, _ in want]
            assert preds.tolist() == [v for _, v in want]
    base = sm.recommendations(rows, sm.compute_prediction(rows))
    g = sm.average(rows)
    glob = sm.recommendations(rows, lambda u, i: g)  # all predictions equal: pure id order
    for u in users[:3] + [424242]:
        for n in (2, 50):
            ids, preds = m.recommend(oracle.KIND_BASELINE, u, n)
            assert (ids.tolist(), preds.tolist()) == ([x for x, _ in base(u, n)], [v for _, v in base(u, n)])
            ids, preds = m.recommend(oracle.KIND_GLOBAL, u, n)
            assert (ids.tolist(), preds.tolist()) == ([x for x, _ in glob(u, n)], [v for _, v in glob(u, n)])
            assert ids.tolist() == sorted(ids.tolist())


def test_invariants_on_ml100k_shape(oracle, syn100k):
    tr, te = syn100k.train, syn100k.test
    m = oracle.Model(tr.users, tr.items, tr.ratings)
    U = m.num_users
    assert U == 943
    # kNN with k >= U-1 is the full adjusted-cosine Personalized predictor (knn-100k.json k=943 row
    # equals personalized-100k.json P.2 MAE in the reference)
    sub = slice(0, 3000)
    full = m.pipeline(oracle.SIM_COSINE, -1).mae(te.users[sub], te.items[sub], te.ratings[sub])
    kmax = m.pipeline(oracle.SIM_COSINE, 943).mae(te.users[sub], te.items[sub], te.ratings[sub])
    assert full == pytest.approx(kmax, abs=1e-12)
    # similarityOne personalized == baseline up to rounding (P.1 == B.2 in the reference)
    ones = m.pipeline(oracle.SIM_ONE, -1).mae(te.users, te.items, te.ratings)
    base = m.mae(oracle.KIND_BASELINE, te.users, te.items, te.ratings)
    assert ones == pytest.approx(base, abs=1e-12)
    p = m.pipeline(oracle.SIM_COSINE, 10)
    order = {int(u): n for n, u in enumerate(m.user_iteration_order())}
    for u in (1, 2, 57, 400, 943):
        ids, sims = p.neighbors(u)
        assert len(ids) == 10 and u not in ids.tolist()
        assert p.knn_similarity(u, u) == 0.0                      # N1: self is never a neighbour
        assert all(abs(s) <= 1 + 1e-12 for s in sims)
        for a in range(9):                                         # N3: sorted desc, ties in Set order
            assert sims[a] > sims[a + 1] or (sims[a] == sims[a + 1] and order[int(ids[a])] < order[int(ids[a + 1])])
        v = int(ids[0])
        assert m.fresh_similarity(oracle.SIM_COSINE, u, v) == pytest.approx(
            m.fresh_similarity(oracle.SIM_COSINE, v, u), abs=1e-15)
    mae, preds = m.pipeline(oracle.SIM_COSINE, 300).mae(te.users[sub], te.items[sub], te.ratings[sub], True)
    assert np.all(preds >= 1.0 - 1e-12) and np.all(preds <= 5.0 + 1e-12)
    assert 0.5 < mae < 1.2


def test_fit_rejects_bad_input(oracle):
    with pytest.raises(oracle.OracleError) as e:  # duplicate (user, item)
        oracle.Model([1, 1, 2], [5, 5, 5], [3.0, 4.0, 2.0])
    assert e.value.status == -3
    with pytest.raises(oracle.OracleError) as e:  # mean 1.0 with a 0.5 rating: scale() == 0 (N5)
        oracle.Model([1, 1, 1], [1, 2, 3], [0.5, 1.0, 1.5])
    assert e.value.status == -2
