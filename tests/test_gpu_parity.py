"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same inputs.

Bars: bit-exact neighbour ids (and, because every per-pair quantity is computed in the reference's
order, bit-exact fp64 similarities and predictions); |dMAE| <= 1e-9 here (north_star asks 1e-6;
only the final fixed-shape reduction differs from the reference's left fold)."""
import importlib

import numpy as np
import pytest

from tests.test_oracle_semantics import TEST, TRAIN, _cols, _no_zero_scale, _random_case

pytestmark = pytest.mark.gpu
MAE_TOL = 1e-9


@pytest.fixture(scope="module")
def kn(pkg):
    mod = importlib.import_module(pkg.__name__ + ".knncf")
    mod.load_library()
    return mod


def _engine(kn, train, k=300, sim=0, flags=0, head_items=0):
    e = kn.Engine(k=k, similarity=sim, flags=flags, head_items=head_items)
    e.fit(*train)
    return e


def test_micro_fixture(kn, oracle):
    tr, te = _cols(TRAIN), _cols(TEST)
    m = oracle.Model(*tr)
    e = _engine(kn, tr, k=1)
    assert e.num_users == 3 and e.num_items == 3
    assert e.global_avg() == m.average()
    assert [e.user_avg(u) for u in (1, 2, 3)] == [3.0, 3.0, 4.5]
    assert e.user_avg(77) == m.average()
    assert [e.item_avg_dev(i) for i in (1, 2, 3)] == [m.items_avg_dev(i) for i in (1, 2, 3)]
    for a, b in ((1, 2), (2, 1), (1, 3), (2, 3), (1, 1)):
        assert e.similarity(a, b) == m.fresh_similarity(0, a, b)
    assert [e.neighbors(u)[0].tolist() for u in (1, 2, 3)] == [[2], [1], [1]]
    assert e.mae(kn.PRED_KNN, *te) == pytest.approx(0.5, abs=1e-12)
    np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), [1.0, 5.0])
    e.set_k(2)
    p = m.pipeline(0, 2)
    want, preds = p.mae(*te, True)
    np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
    assert e.mae(kn.PRED_KNN, *te) == pytest.approx(want, abs=1e-15)
    assert e.mae(kn.PRED_BASELINE, *te) == pytest.approx(0.13392857142857142, abs=1e-15)


@pytest.mark.parametrize("seed", range(8))
def test_random_small_cases_bitwise(kn, oracle, seed):
    rng = np.random.default_rng(100 + seed)
    rows = _random_case(rng, n_users=10 + 3 * seed, n_items=17, n_ratings=80 + 11 * seed,
                        half=(seed % 2 == 1), tiny_rows=seed % 4)
    cut = len(rows) * 4 // 5
    train, test = rows[:cut], rows[cut:]
    if not _no_zero_scale(train):
        pytest.skip("scale() == 0 corner")
    test += [(999_999, train[0][1], 3.0), (train[0][0], 888_888, 4.0)]
    tr, te = _cols(train), _cols(test)
    m = oracle.Model(*tr)
    users = sorted(set(tr[0]))
    items = sorted(set(tr[1]))
    for k in (1, 4, len(users) + 3):
        e = _engine(kn, tr, k=k, flags=kn.FLAG_VERIFY_BOUND)
        assert e.global_avg() == m.average()
        assert [e.user_avg(u) for u in users] == [m.users_avg(u) for u in users]
        assert [e.item_avg(i) for i in items] == [m.items_avg(i) for i in items]
        assert [e.item_avg_dev(i) for i in items] == [m.items_avg_dev(i) for i in items]
        for kind, okind in ((kn.PRED_GLOBAL_AVG, 0), (kn.PRED_USER_AVG, 1), (kn.PRED_ITEM_AVG, 2),
                            (kn.PRED_BASELINE, 3), (kn.PRED_BASELINE_RDD, 4)):
            want, preds = m.mae(okind, *te, True)
            np.testing.assert_array_equal(e.predict_batch(kind, te[0], te[1]), preds)
            assert e.mae(kind, *te) == pytest.approx(want, abs=1e-13)
        # the timed expression of predict/kNN.scala:42-45: same closure history on both sides
        p = m.pipeline(oracle.SIM_COSINE, k)
        want, preds = p.mae(*te, True)
        got = e.mae(kn.PRED_KNN, *te)
        np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
        assert got == pytest.approx(want, abs=1e-13)
        for u in users:
            ids, sims = e.neighbors(u)
            oids, osims = p.neighbors(u)
            assert ids.tolist() == oids.tolist()
            assert sims.tolist() == osims.tolist()
            assert e.knn_similarity(u, u) == 0.0
        assert e.timings()["max_bound_violation"] <= 0.0
        e.close()
    one = _engine(kn, tr, sim=kn.SIM_ONE)
    want, preds = m.pipeline(oracle.SIM_ONE, -1).mae(*te, True)
    np.testing.assert_array_equal(one.predict_batch(kn.PRED_PERSONALIZED, te[0], te[1]), preds)
    jac = _engine(kn, tr, sim=kn.SIM_JACCARD)
    for a, b in zip(users[:6], users[3:9]):
        assert jac.similarity(a, b) == m.fresh_similarity(oracle.SIM_JACCARD, a, b)
    # Personalized without a neighbourhood cut: Jaccard always; the adjusted cosine when no user has <= 4 ratings
    # (otherwise the reference's summation order depends on its memo history pair by pair: refused, not approximated).
    # Training pairs are predicted too: the user is then one of the item's raters and weighs in with sim(u, u).
    pu = np.concatenate([te[0], tr[0][:12]])
    pi = np.concatenate([te[1], tr[1][:12]])
    want = np.array([m.pipeline(oracle.SIM_JACCARD, -1).predict(int(a), int(b)) for a, b in zip(pu, pi)])
    np.testing.assert_array_equal(jac.predict_batch(kn.PRED_PERSONALIZED, pu, pi), want)
    cosp = _engine(kn, tr, sim=kn.SIM_COSINE)
    if min(np.bincount(np.unique(tr[0], return_inverse=True)[1])) > 4:
        pc = m.pipeline(oracle.SIM_COSINE, -1)
        want = np.array([pc.predict(int(a), int(b)) for a, b in zip(pu, pi)])
        np.testing.assert_array_equal(cosp.predict_batch(kn.PRED_PERSONALIZED, pu, pi), want)
    else:
        with pytest.raises(kn.KnncfError):
            cosp.predict_batch(kn.PRED_PERSONALIZED, pu, pi)


@pytest.mark.parametrize("shuffle", [False, True])
def test_ml100k_shape_all_neighbours_and_predictions(kn, oracle, synth, shuffle):
    d = synth.syn_100k(shuffle=shuffle)
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    m = oracle.Model(*tr)
    users = np.unique(d.train.users)
    for k in (10, 300, 943):
        e = _engine(kn, tr, k=k, flags=kn.FLAG_VERIFY_BOUND)
        p = m.pipeline(oracle.SIM_COSINE, k)
        want, preds = p.mae(*te, True)
        got = e.mae(kn.PRED_KNN, *te)
        gp = e.predict_batch(kn.PRED_KNN, te[0], te[1])
        assert abs(got - want) <= MAE_TOL
        np.testing.assert_array_equal(gp, preds)
        for u in users[:: 7 if k != 10 else 1]:
            ids, sims = e.neighbors(int(u))
            oids, osims = p.neighbors(int(u))
            assert ids.tolist() == oids.tolist(), f"user {u} k {k}"
            assert sims.tolist() == osims.tolist()
        t = e.timings()
        assert t["max_bound_violation"] <= 0.0
        assert t["gemm_launches"] >= 1
        e.close()


def test_personalized_cosine_and_jaccard_ml100k_shape(kn, oracle, synth):
    """predict/Personalized.scala P.2 / P.3: predictor(train, weightedSumDeviation(train, sim)) with the adjusted cosine
    and the Jaccard coefficient themselves (no k), test rows plus some training pairs (self term), bit for bit"""
    d = synth.syn_100k()
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    m = oracle.Model(*tr)
    pu = np.concatenate([te[0], tr[0][:200]])
    pi = np.concatenate([te[1], tr[1][:200]])
    pr = np.concatenate([te[2], tr[2][:200]])
    for sim_o, sim_k in ((oracle.SIM_COSINE, kn.SIM_COSINE), (oracle.SIM_JACCARD, kn.SIM_JACCARD)):
        e = _engine(kn, tr, sim=sim_k)
        want, preds = m.pipeline(sim_o, -1).mae(pu, pi, pr, True)
        np.testing.assert_array_equal(e.predict_batch(kn.PRED_PERSONALIZED, pu, pi), preds)
        assert abs(e.mae(kn.PRED_PERSONALIZED, pu, pi, pr) - want) <= MAE_TOL
        e.close()


@pytest.mark.parametrize("head", [0, 64, 0xFFFFFFFF])
def test_knn_with_the_jaccard_coefficient(kn, oracle, syn100k, head):
    """getSimilarity(train, k, jaccardCoefficient(train)) — the k-nearest-neighbour closures over the reference's other
    similarity (shared/predictions.scala:440-464, :596-649) — through the SAME pipeline as the adjusted cosine: the 0/1
    operand panel's MFMA GEMM and the tail's LDS atomics count the common items exactly, select.hip turns the counts into
    count / (|I(u)| + |I(v)| - count) in fp32, the re-rank recomputes the shortlist exactly (fp64 quotient of exact
    integers).  Jaccard values tie massively; ties are broken by Set order like everywhere else (N3).  Every split of the
    items into dense head and sparse tail must give the same neighbours."""
    d = syn100k
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    m = oracle.Model(*tr)
    users = np.unique(d.train.users)
    for k in (10, 300):
        e = _engine(kn, tr, k=k, sim=kn.SIM_JACCARD, flags=kn.FLAG_VERIFY_BOUND, head_items=head)
        p = m.pipeline(oracle.SIM_JACCARD, k)
        want, preds = p.mae(*te, True)
        got = e.mae(kn.PRED_KNN, *te)
        np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
        assert abs(got - want) <= MAE_TOL
        ids, sims, counts = e.neighbors_batch(users[::5])
        for row, u in enumerate(users[::5]):
            oids, osims = p.neighbors(int(u))
            assert ids[row, :counts[row]].tolist() == oids.tolist(), f"user {u} k {k}"
            assert sims[row, :counts[row]].tolist() == osims.tolist()
        assert e.knn_similarity(int(users[0]), int(users[0])) == 0.0
        t = e.timings()
        assert t["max_bound_violation"] <= 0.0 and t["gemm_launches"] >= 1
        e.close()


@pytest.mark.parametrize("symmetric", [True, False])
def test_jaccard_knn_beyond_the_personalized_table(kn, oracle, synth, symmetric, monkeypatch):
    """U = 20 000 (the U x U table of PERSONALIZED stops at 2048 users): Jaccard neighbourhoods through the counting GEMM —
    the symmetric launch and the row-block launches — two column tiles per row, sampled users and their predictions bit
    for bit against the oracle's per-pair closures"""
    if not symmetric:
        monkeypatch.setenv("KNNCF_DEBUG_NO_SYMMETRIC_GEMM", "1")
    d = synth.syn_scaled(20_000, 3_000, 1_500_000, seed=41, half_stars=True)
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    k = 50
    e = _engine(kn, tr, k=k, sim=kn.SIM_JACCARD, flags=kn.FLAG_VERIFY_BOUND)
    preds = e.predict_batch(kn.PRED_KNN, te[0], te[1])
    t = e.timings()
    assert t["max_bound_violation"] <= 0.0 and t["fallback_rows"] == 0
    assert t["gemm_launches"] == 1 and t["select_launches"] == 1
    p = oracle.Model(*tr).pipeline(oracle.SIM_JACCARD, k)
    users = np.unique(d.train.users)
    sample = users[:: len(users) // 24]
    ids, sims, counts = e.neighbors_batch(sample)
    for row, u in enumerate(sample):
        oids, osims = p.neighbors(int(u))
        assert ids[row, :counts[row]].tolist() == oids.tolist(), f"user {u}"
        assert sims[row, :counts[row]].tolist() == osims.tolist()
    mask = np.isin(te[0], sample)
    _, opreds = p.mae(te[0][mask], te[1][mask], te[2][mask], True)
    np.testing.assert_array_equal(preds[mask], opreds)
    e.close()


@pytest.mark.parametrize("seed", range(4))
def test_jaccard_knn_random_small_cases(kn, oracle, seed):
    """tiny inputs, users with <= 4 ratings included (the Jaccard coefficient has no summation order to depend on),
    unknown users and items among the test rows"""
    rng = np.random.default_rng(900 + seed)
    rows = _random_case(rng, n_users=14 + 4 * seed, n_items=19, n_ratings=90 + 15 * seed, half=(seed % 2 == 1), tiny_rows=seed % 3)
    cut = len(rows) * 4 // 5
    train, test = rows[:cut], rows[cut:]
    if not _no_zero_scale(train):
        pytest.skip("scale() == 0 corner")
    test += [(999_999, train[0][1], 3.0), (train[0][0], 888_888, 4.0)]
    tr, te = _cols(train), _cols(test)
    m = oracle.Model(*tr)
    users = sorted(set(tr[0]))
    for k in (1, 3, len(users) + 2):
        e = _engine(kn, tr, k=k, sim=kn.SIM_JACCARD, flags=kn.FLAG_VERIFY_BOUND)
        p = m.pipeline(oracle.SIM_JACCARD, k)
        want, preds = p.mae(*te, True)
        np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
        assert e.mae(kn.PRED_KNN, *te) == pytest.approx(want, abs=1e-13)
        for u in users:
            ids, sims = e.neighbors(u)
            oids, osims = p.neighbors(u)
            assert ids.tolist() == oids.tolist(), f"user {u} k {k}"
            assert sims.tolist() == osims.tolist()
        assert e.timings()["max_bound_violation"] <= 0.0
        e.close()


def test_recommendations_equal_oracle(kn, oracle, synth):
    """recommendations :651-674: ids and predictions bit for bit, ml-100k shape, the Recommender's
    n = 3 and a longer list; kNN k = 300 (recommend/Recommender.scala:85-88), the baseline, and the all-ties case"""
    d = synth.syn_100k()
    tr = (d.train.users, d.train.items, d.train.ratings)
    m = oracle.Model(*tr)
    e = _engine(kn, tr, k=300)
    p = m.pipeline(oracle.SIM_COSINE, 300)
    users = np.unique(d.train.users)
    n_items = len(np.unique(d.train.items))
    for u in [int(x) for x in users[::97]] + [987654]:
        for n in (3, 25):
            ids, preds = e.recommend(kn.PRED_KNN, u, n)
            oi, op = p.recommend(u, n)
            assert ids.tolist() == oi.tolist(), f"user {u} n {n}"
            assert preds.tolist() == op.tolist()
        ids, preds = e.recommend(kn.PRED_BASELINE, u, 10)
        oi, op = m.recommend(oracle.KIND_BASELINE, u, 10)
        assert (ids.tolist(), preds.tolist()) == (oi.tolist(), op.tolist())
        ids, preds = e.recommend(kn.PRED_GLOBAL_AVG, u, 7)  # every prediction equal: the 7 smallest unrated ids
        oi, op = m.recommend(oracle.KIND_GLOBAL, u, 7)
        assert (ids.tolist(), preds.tolist()) == (oi.tolist(), op.tolist())
        assert ids.tolist() == sorted(ids.tolist())
    # n beyond the number of unrated items: everything unrated, still ordered
    u = int(users[0])
    rated = int((d.train.users == u).sum())
    ids, preds = e.recommend(kn.PRED_KNN, u, 10 * n_items)
    assert len(ids) == n_items - rated
    oi, op = p.recommend(u, 10 * n_items)
    assert (ids.tolist(), preds.tolist()) == (oi.tolist(), op.tolist())
    assert not set(ids.tolist()) & set(d.train.items[d.train.users == u].tolist())
    e.close()


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("head", [64, 320, 0xFFFFFFFF])
def test_hybrid_head_tail_split_is_exact(kn, oracle, syn100k, head, bf16):
    """Dense MFMA head (fp16 or bf16 operands) + sparse fp32 LDS-atomic tail: any split and either
    filter precision must give the same exact neighbours, inside the rigorous error band."""
    d = syn100k
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    m = oracle.Model(*tr)
    p = m.pipeline(oracle.SIM_COSINE, 50)
    want, preds = p.mae(*te, True)
    e = _engine(kn, tr, k=50, flags=kn.FLAG_VERIFY_BOUND | (kn.FLAG_BF16_FILTER if bf16 else 0), head_items=head)
    got = e.mae(kn.PRED_KNN, *te)
    np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
    assert abs(got - want) <= MAE_TOL
    t = e.timings()
    assert t["max_bound_violation"] <= 0.0
    assert t["head_items"] == min(head, e.num_items)
    assert (t["tail_pair_updates"] > 0) == (head < e.num_items)
    for u in np.unique(d.train.users)[::11]:
        ids, sims = e.neighbors(int(u))
        oids, osims = p.neighbors(int(u))
        assert ids.tolist() == oids.tolist() and sims.tolist() == osims.tolist()


@pytest.mark.parametrize("bf16", [False, True])
def test_fp32_panel_gives_the_same_neighbours(kn, oracle, syn100k, bf16):
    """KNNCF_FLAG_F32_PANEL: the GEMM's fp32 epilogue (four LDS passes per tile) and the select kernel's fp32 loads —
    a narrower error band, the same exact neighbours and predictions."""
    d = syn100k
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    p = oracle.Model(*tr).pipeline(oracle.SIM_COSINE, 50)
    want, preds = p.mae(*te, True)
    e = _engine(kn, tr, k=50, flags=kn.FLAG_VERIFY_BOUND | kn.FLAG_F32_PANEL | (kn.FLAG_BF16_FILTER if bf16 else 0), head_items=320)
    np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
    assert abs(e.mae(kn.PRED_KNN, *te) - want) <= MAE_TOL
    assert e.timings()["max_bound_violation"] <= 0.0
    for u in np.unique(d.train.users)[::37]:
        ids, sims = e.neighbors(int(u))
        oids, osims = p.neighbors(int(u))
        assert ids.tolist() == oids.tolist() and sims.tolist() == osims.tolist()
    e.close()


@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("flags", [0, 2])
def test_several_row_blocks_and_the_overlap_flag(kn, oracle, syn100k, flags, symmetric, monkeypatch):
    """A small workspace cuts the users into 256-row blocks (4 at ml-100k shape); KNNCF_FLAG_OVERLAP (= 2) then runs
    the GEMM of block b + 1 on a second stream into a second panel slot while block b is selected and re-ranked.
    Whole-matrix builds take the symmetric GEMM (one launch, tiles on/above the diagonal mirrored) in front of the same
    row blocks; KNNCF_DEBUG_NO_SYMMETRIC_GEMM forces the row-block GEMMs that sharded and partial builds always use."""
    if not symmetric:
        monkeypatch.setenv("KNNCF_DEBUG_NO_SYMMETRIC_GEMM", "1")
    d = syn100k
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    p = oracle.Model(*tr).pipeline(oracle.SIM_COSINE, 50)
    want, preds = p.mae(*te, True)
    e = kn.Engine(k=50, flags=flags | kn.FLAG_VERIFY_BOUND, head_items=128, workspace_bytes=1 << 20)
    e.fit(*tr)
    np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
    assert abs(e.mae(kn.PRED_KNN, *te) - want) <= MAE_TOL
    t = e.timings()
    assert t["select_launches"] >= 4 and t["max_bound_violation"] <= 0.0
    assert t["gemm_launches"] == (1 if symmetric else t["select_launches"])
    for u in np.unique(d.train.users)[::29]:
        ids, sims = e.neighbors(int(u))
        oids, osims = p.neighbors(int(u))
        assert ids.tolist() == oids.tolist() and sims.tolist() == osims.tolist()
    e.close()


@pytest.mark.parametrize("global_sorts", [False, True])
def test_two_shards_on_one_gpu_equal_single_engine(kn, pkg, oracle, synth, monkeypatch, global_sorts):
    """The C-ABI shard protocol (view -> exchange -> commit -> partial MAE) with two handles in one
    process; the exchange that RCCL's all-gather performs between GPUs is done here by device copies.
    Once with every shard ordering its users' positions by per-user LDS sorts, once by the global (slice) radix sorts that
    a file with a very long row takes."""
    import torch

    if global_sorts:
        monkeypatch.setenv("KNNCF_DEBUG_GLOBAL_HASH_ORDER", "1")
    else:
        monkeypatch.delenv("KNNCF_DEBUG_GLOBAL_HASH_ORDER", raising=False)

    sharded = importlib.import_module(pkg.__name__ + ".sharded")
    d = synth.syn_scaled(700, 400, 42_000, seed=11, half_stars=True, shuffle=True)
    dev = torch.device("cuda", 0)
    tr = tuple(torch.from_numpy(a).to(dev) for a in (d.train.users, d.train.items, d.train.ratings))
    te = tuple(torch.from_numpy(a).to(dev) for a in (d.test.users, d.test.items, d.test.ratings))
    k = 40
    single = kn.Engine(k=k)
    single.fit_device(*tr)
    s1, c1 = single.mae_device(kn.PRED_KNN, *te)
    engines = [kn.Engine(k=k, shard_rank=r, shard_count=2) for r in range(2)]
    views = []
    for e in engines:
        e.fit_device(*tr)
        views.append(sharded.DeviceEngineAdapter(e, dev).shard_tensors())
    assert views[0]["user_range"][1] == views[1]["user_range"][0]
    for me, other in ((0, 1), (1, 0)):  # the exchange: the other shard's per-user (mean, norm); nothing per rating travels
        ulo, uhi = views[other]["user_range"]
        for key in ("user_avg", "user_norm"):
            views[me][key][ulo:uhi] = views[other][key][ulo:uhi]
    torch.cuda.synchronize()
    total, count = 0.0, 0
    preds = torch.zeros(len(d.test.users), dtype=torch.float64, device=dev)
    for e in engines:
        with pytest.raises(kn.KnncfError):  # not committed yet
            e.mae_device(kn.PRED_KNN, *te)
        e.shard_commit()
        s, c = e.mae_device(kn.PRED_KNN, *te, pred_out=preds)
        total += s
        count += c
    assert count == c1 == len(d.test.users)
    assert total / count == pytest.approx(s1 / c1, abs=1e-13)
    p = oracle.Model(d.train.users, d.train.items, d.train.ratings).pipeline(oracle.SIM_COSINE, k)
    want, opreds = p.mae(d.test.users, d.test.items, d.test.ratings, True)
    np.testing.assert_array_equal(preds.cpu().numpy(), opreds)
    assert total / count == pytest.approx(want, abs=MAE_TOL)


def test_wide_shape_takes_the_large_u_paths(kn, oracle, synth):
    """U = 300 000 users: the similarity row spans 19 column tiles (more than the per-entry counts held in
    registers: select.hip reads the tile table per tile) and the item bitmaps no longer fit in LDS (prediction falls
    back to k_predict_knn_rows: bitmaps in global memory, rows sorted by user).  Sampled users bit for bit."""
    import torch

    d = synth.syn_scaled(300_000, 4_000, 6_000_000, seed=77, half_stars=False)
    dev = torch.device("cuda", 0)
    tr = tuple(torch.from_numpy(a).to(dev) for a in (d.train.users, d.train.items, d.train.ratings))
    te = tuple(torch.from_numpy(a).to(dev) for a in (d.test.users, d.test.items, d.test.ratings))
    k = 50
    e = kn.Engine(k=k, flags=kn.FLAG_VERIFY_BOUND)
    e.fit_device(*tr)
    preds = torch.zeros(len(d.test.users), dtype=torch.float64, device=dev)
    s, c = e.mae_device(kn.PRED_KNN, *te, pred_out=preds)
    preds = preds.cpu().numpy()
    assert c == len(d.test.users) and np.isfinite(preds).all()
    t = e.timings()
    assert t["max_bound_violation"] <= 0.0
    m = oracle.Model(d.train.users, d.train.items, d.train.ratings)
    p = m.pipeline(oracle.SIM_COSINE, k)
    users = np.unique(d.train.users)
    for u in users[:: len(users) // 12][:12]:
        ids, sims = e.neighbors(int(u))
        oids, osims = p.neighbors(int(u))
        assert ids.tolist() == oids.tolist(), f"user {u}"
        assert sims.tolist() == osims.tolist()
        rows = np.nonzero(d.test.users == u)[0]
        for r_ in rows[:8]:
            assert preds[r_] == p.predict(int(u), int(d.test.items[r_]))
    e.close()


@pytest.mark.parametrize("path", ["symmetric", "one_row_block", "row_blocks"])
def test_overshooting_anticipated_thresholds_are_caught(kn, synth, monkeypatch, path):
    """select.hip emits against ANTICIPATED thresholds (rank k f + 7 sigma ... of the columns seen) and verifies them at the
    end of the row; with the margin cut to 2.5 sigma (test hook) the guess overshoots in some rows, the final check must
    notice, and those rows are rebuilt exactly: every neighbour list then still equals the default build's, which the
    full-size and wide-shape tests pin to the oracle.  With no margin at all the guess fails in a large part of the rows: the
    host then sends the marked rows through select + re-rank once more with the plain thresholds instead of through the
    per-row exact path (no fallback row is left) — on the whole-matrix path, on the one-block row-block path that sharded
    handles take, and per block when a capped workspace cuts the users into several row blocks (syn-1M's path)."""
    import torch

    if path != "symmetric":
        monkeypatch.setenv("KNNCF_DEBUG_NO_SYMMETRIC_GEMM", "1")
    else:
        monkeypatch.delenv("KNNCF_DEBUG_NO_SYMMETRIC_GEMM", raising=False)
    workspace = (6 << 30) if path == "row_blocks" else 0      # ~16 k rows per block: six blocks
    d = synth.syn_scaled(90_000, 3_000, 3_000_000, seed=31, half_stars=True)   # six column tiles
    dev = torch.device("cuda", 0)
    tr = tuple(torch.from_numpy(a).to(dev) for a in (d.train.users, d.train.items, d.train.ratings))
    users = np.unique(d.train.users)
    out = {}
    for sigma in (None, "2.5", "0", "-1"):
        if sigma is None:
            monkeypatch.delenv("KNNCF_DEBUG_ANTICIPATE_SIGMA", raising=False)
        else:
            monkeypatch.setenv("KNNCF_DEBUG_ANTICIPATE_SIGMA", sigma)
        e = kn.Engine(k=100, flags=kn.FLAG_VERIFY_BOUND, workspace_bytes=workspace)
        e.fit_device(*tr)
        ids, sims, counts = e.neighbors_batch(users)
        t = e.timings()
        assert t["max_bound_violation"] <= 0.0
        out[sigma] = (ids, sims, counts, t["fallback_rows"], t["select_launches"], t["gemm_launches"])
        e.close()
    blocks = 1 if path == "symmetric" else out[None][5]  # row blocks of the build
    assert blocks == 1 if path != "row_blocks" else blocks >= 3
    assert out[None][3] == 0 and out["-1"][3] == 0        # 7 sigma / no anticipation: no row needs the fallback
    assert 0 < out["2.5"][3] < len(users) // 10           # the hook really produced overshoots, and not everywhere
    # too many overshoots for the per-row path: one plain second pass per block, no fallback row left
    assert out["0"][3] == 0 and out["0"][4] == out[None][4] + blocks
    for sigma in ("2.5", "0", "-1"):
        assert np.array_equal(out[sigma][0], out[None][0]) and np.array_equal(out[sigma][2], out[None][2])
        assert np.array_equal(out[sigma][1].view(np.int64), out[None][1].view(np.int64))


def test_ids_outside_the_direct_tables(kn, oracle, synth, monkeypatch):
    """raw ids that are negative or >= 2^24 (not MovieLens, but legal Ints for `load`) take the general id path: every row's
    key sorted + unique, hash + binary search per row; and the same path forced on ordinary ids"""
    d = synth.syn_scaled(300, 120, 9_000, seed=5, half_stars=True)
    big = lambda a, off: (a.astype(np.int64) * 7919 + off).astype(np.int32)
    tr = (big(d.train.users, -40_000), big(d.train.items, 1 << 25), d.train.ratings)
    te = (big(d.test.users, -40_000), big(d.test.items, 1 << 25), d.test.ratings)
    for force, (a, b) in ((False, (tr, te)), (True, ((d.train.users, d.train.items, d.train.ratings), (d.test.users, d.test.items, d.test.ratings)))):
        if force:
            monkeypatch.setenv("KNNCF_DEBUG_NO_ID_TABLES", "1")
        e = _engine(kn, a, k=20)
        p = oracle.Model(*a).pipeline(oracle.SIM_COSINE, 20)
        want, preds = p.mae(*b, True)
        np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, b[0], b[1]), preds)
        assert abs(e.mae(kn.PRED_KNN, *b) - want) <= MAE_TOL
        for u in np.unique(a[0])[::37]:
            ids, sims = e.neighbors(int(u))
            oids, osims = p.neighbors(int(u))
            assert ids.tolist() == oids.tolist() and sims.tolist() == osims.tolist()
        e.close()


def test_user_rows_of_every_size_class(kn, oracle, monkeypatch):
    """the canonical (user, item) order and the (user, HashMap order) / (user, file row) positions come from per-user LDS
    sorts in three size classes (<= 512, <= 2048, <= 8192 ratings) and from the global radix sorts when some user has more:
    files with users of 3 000, 1 000 and ~60 ratings, with and without one of 12 000 (10 800 of them in the training part);
    the norms (hence every similarity, bit for bit) depend on those orders.  Also with the global sorts forced,
    and with non-dyadic ratings (then usersAvg really folds in file order)."""
    for big, variants in ((7_000, ((True, False), (True, True), (False, False), (False, True))), (12_000, ((True, False), (False, False)))):
        rng = np.random.default_rng(11)
        n_items = 20_000
        users, items = [], []
        for u, cnt in [(1, big), (2, 3_000), (3, 1_000)] + [(10 + j, int(rng.integers(30, 90))) for j in range(40)]:
            its = rng.choice(n_items, size=cnt, replace=False) + 1
            users += [u] * cnt
            items += its.tolist()
        users, items = np.asarray(users, np.int32), np.asarray(items, np.int32)
        order = rng.permutation(len(users))
        users, items = users[order], items[order]
        cut = len(users) * 9 // 10
        assert (np.sum(users[:cut] == 1) > 8192) == (big > 10_000)  # (the heaviest row is beyond the last LDS class or inside it)
        made = {}
        for dyadic, forced in variants:
            if dyadic not in made:
                ratings = rng.integers(1, 11, size=len(users)) / 2.0
                if not dyadic:
                    ratings = ratings + rng.integers(0, 7, size=len(users)) * 0.1  # (sums now depend on the order)
                    ratings = np.minimum(ratings, 5.0)
                tr = (users[:cut], items[:cut], ratings[:cut])
                te = (users[cut:], items[cut:], ratings[cut:])
                p = oracle.Model(*tr).pipeline(oracle.SIM_COSINE, 10)
                made[dyadic] = (tr, te) + p.mae(*te, True)
            tr, te, want, preds = made[dyadic]
            if forced:
                monkeypatch.setenv("KNNCF_DEBUG_GLOBAL_HASH_ORDER", "1")
            else:
                monkeypatch.delenv("KNNCF_DEBUG_GLOBAL_HASH_ORDER", raising=False)
            e = _engine(kn, tr, k=10)
            np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
            assert abs(e.mae(kn.PRED_KNN, *te) - want) <= MAE_TOL
            m2 = oracle.Model(*tr)
            p2 = m2.pipeline(oracle.SIM_COSINE, 10)
            for u in (1, 2, 3, 10, 25, 49):
                ids, sims = e.neighbors(u)
                oids, osims = p2.neighbors(u)
                assert ids.tolist() == oids.tolist() and sims.tolist() == osims.tolist()
                assert e.user_avg(u) == m2.users_avg(u)
            e.close()


def test_prediction_without_item_bitmaps(kn, oracle, synth, monkeypatch):
    """shapes whose rater bitmaps would not fit in HBM predict through binary searches (k_predict_knn): forced here"""
    monkeypatch.setenv("KNNCF_DEBUG_NO_ITEM_BITMAPS", "1")
    d = synth.syn_100k()
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    e = _engine(kn, tr, k=40)
    want, preds = oracle.Model(*tr).pipeline(oracle.SIM_COSINE, 40).mae(*te, True)
    np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
    assert abs(e.mae(kn.PRED_KNN, *te) - want) <= MAE_TOL
    e.close()


@pytest.fixture(scope="module")
def full25m(kn, oracle, synth):
    """BASELINE's metric configuration (syn-25m: 162 541 x 59 047, 20 M / 5 M ratings, k = 300) run ONCE through the
    single-GPU engine with KNNCF_FLAG_VERIFY_BOUND, and the oracle's bulk form (all cores) of the same closures."""
    import torch

    d = synth.syn_25m()
    dev = torch.device("cuda", 0)
    tr = tuple(torch.from_numpy(a).to(dev) for a in (d.train.users, d.train.items, d.train.ratings))
    te = tuple(torch.from_numpy(a).to(dev) for a in (d.test.users, d.test.items, d.test.ratings))
    e = kn.Engine(k=300, flags=kn.FLAG_VERIFY_BOUND)
    e.fit_device(*tr)
    preds = torch.zeros(len(d.test.users), dtype=torch.float64, device=dev)
    s, c = e.mae_device(kn.PRED_KNN, *te, pred_out=preds)
    m = oracle.Model(d.train.users, d.train.items, d.train.ratings)
    out = {"d": d, "tr": tr, "te": te, "engine": e, "preds": preds.cpu().numpy(), "sum": s, "count": c, "model": m,
           "timings": e.timings()}
    yield out
    e.close()


def test_full_size_ml25m_shape_every_row_against_the_oracle(kn, oracle, full25m):
    """predict/kNN.scala:42-45 at the headline shape, COMPLETE coverage: all 162 541 neighbour lists (ids and fp64
    similarities, bit for bit), all 5 000 019 predictions (bit for bit) and the MAE (1e-9; north_star 1e-6) against the
    oracle's bulk form (tests/test_oracle_bulk.py pins that form to the literal per-pair closures).  A true neighbour
    dropped by the 16-bit filter would show as a differing list, so this also validates the error band of
    select.hip at the one shape that is benchmarked."""
    f = full25m
    d, e, m, preds = f["d"], f["engine"], f["model"], f["preds"]
    n_test = len(d.test.users)
    assert f["count"] == n_test and np.all(np.isfinite(preds))
    t = f["timings"]
    assert t["fallback_rows"] == 0 and t["head_items"] > 0
    assert t["max_bound_violation"] <= 0.0          # |approx - exact| <= eps on every re-ranked pair
    table = m.knn_table(300)                        # every user, all host cores
    assert table.rows == e.num_users == 162_541 and table.width == 300
    ids, sims, counts = e.neighbors_batch(table.row_user)
    assert (counts == 300).all()
    bad = np.flatnonzero((ids != table.ids).any(axis=1))
    assert len(bad) == 0, f"{len(bad)} neighbour lists differ, first user {table.row_user[bad[0]]}"
    assert np.array_equal(sims.view(np.int64), table.sims.view(np.int64))   # fp64 similarities, bitwise
    del ids, sims
    want, opreds = table.mae(d.test.users, d.test.items, d.test.ratings)
    diff = np.flatnonzero(preds.view(np.int64) != opreds.view(np.int64))
    assert len(diff) == 0, f"{len(diff)} of {n_test} predictions differ, first row {diff[0]}"
    mae = f["sum"] / f["count"]
    assert abs(mae - want) <= MAE_TOL, (mae, want)
    # idempotence: a second pass over the same test set (neighbourhoods already built) gives the same sums
    s2, c2 = e.mae_device(kn.PRED_KNN, *f["te"])
    assert (s2, c2) == (f["sum"], f["count"])
    # the closed forms at this size
    assert e.global_avg() == m.average()
    for u in table.row_user[:: 40_000]:
        assert e.user_avg(int(u)) == m.users_avg(int(u))
    for i in np.unique(d.test.items)[:5]:
        assert e.item_avg_dev(int(i)) == m.items_avg_dev(int(i))
    want_b = m.mae(oracle.KIND_BASELINE, d.test.users, d.test.items, d.test.ratings)
    sb, cb = e.mae_device(kn.PRED_BASELINE, *f["te"])
    assert sb / cb == pytest.approx(want_b, abs=MAE_TOL)


@pytest.mark.parametrize("k", [30, 1000])
def test_full_size_ml25m_shape_other_k_bulk_lists(kn, oracle, full25m, k):
    """The headline shape at a small and a large k: select.hip anticipates its emission thresholds from the rank k f +
    7 sqrt(k f (1 - f)) + 3 (f = fraction of the row seen), which behaves differently at k = 30 (the anticipated rank is
    close to k) and k = 1000 (far below it, 1024-entry shortlist tiles in the re-rank).  Every user's neighbour list (ids and
    fp64 similarities) against the oracle's bulk form, the error band verified, no fallback row."""
    f = full25m
    m = f["model"]
    e = kn.Engine(k=k, flags=kn.FLAG_VERIFY_BOUND)
    try:
        e.fit_device(*f["tr"])
        table = m.knn_table(k)
        ids, sims, counts = e.neighbors_batch(table.row_user)
        t = e.timings()
        assert t["fallback_rows"] == 0 and t["max_bound_violation"] <= 0.0
        assert (counts == k).all()
        bad = np.flatnonzero((ids != table.ids).any(axis=1))
        assert len(bad) == 0, f"{len(bad)} neighbour lists differ at k = {k}, first user {table.row_user[bad[0]]}"
        assert np.array_equal(sims.view(np.int64), table.sims.view(np.int64))
    finally:
        e.close()


def test_ml25m_shape_eight_shards_on_one_gpu(kn, pkg, full25m):
    """BASELINE config 4 (kNN k = 300 on ml-25m shape, users sharded x8) rehearsed on ONE GPU: eight handles with
    shard_rank 0..7 go through the C-ABI shard protocol (fit -> view -> exchange of the per-user (mean, norm) -> commit
    -> partial MAE); the exchange that RCCL's all-gather performs between GPUs is done by device copies.  Every
    prediction must equal the single engine's bit for bit (which the test above pins to the oracle) and the partial sums
    must add up to its MAE.  (The per-shard timings behind DESIGN.md's projected scaling come from
    scripts/shard_rehearsal.py — a measurement script, not this test.)"""
    import torch

    sharded = importlib.import_module(pkg.__name__ + ".sharded")
    f = full25m
    d, tr, te = f["d"], f["tr"], f["te"]
    dev = tr[0].device
    world = 8
    engines = [kn.Engine(k=300, shard_rank=r, shard_count=world) for r in range(world)]
    # two passes over the whole protocol: the first one allocates every buffer (hipMalloc stalls of up to seconds were seen
    # in it), the second is the steady state the timings are taken from — like bench.py's warm-up steps
    for rehearsal in range(2):
        views = []
        for e in engines:
            e.reset_timings()
            e.fit_device(*tr)
            views.append(sharded.DeviceEngineAdapter(e, dev).shard_tensors())
        for r in range(1, world):
            assert views[r - 1]["user_range"][1] == views[r]["user_range"][0]
        assert views[0]["user_range"][0] == 0 and views[-1]["user_range"][1] == engines[0].num_users
        for me in range(world):
            for other in range(world):
                if other == me:
                    continue
                ulo, uhi = views[other]["user_range"]
                for key in ("user_avg", "user_norm"):
                    views[me][key][ulo:uhi] = views[other][key][ulo:uhi]
        torch.cuda.synchronize()
        total, count = 0.0, 0
        preds = torch.full((len(d.test.users),), float("nan"), dtype=torch.float64, device=dev)
        for r, e in enumerate(engines):
            e.shard_commit()
            s, c = e.mae_device(kn.PRED_KNN, *te, pred_out=preds)
            total += s
            count += c
            assert e.timings()["fallback_rows"] == 0
    for e in engines:
        e.close()
    assert count == f["count"] == len(d.test.users)
    got = preds.cpu().numpy()
    assert np.array_equal(got.view(np.int64), f["preds"].view(np.int64))    # every prediction, bit for bit
    assert total / count == pytest.approx(f["sum"] / f["count"], abs=1e-12)  # only the order of the final sum differs


@pytest.mark.parametrize("seed", range(4))
def test_sharded_fit_with_tiny_rows_equals_single_handle(kn, pkg, oracle, seed):
    """A <= 4-rating user makes a pair's summation order depend on which closure evaluated it first (Set1..Set4
    iterate in insertion order, the cosine memo is symmetric).  One handle models that history with a build sequence number
    per user; shards derive the SAME numbers without any exchange — the test rows are replicated, so every shard knows the
    first test row of every user — and rerank.hip applies the owner rule unchanged.  Two and three shards on one GPU
    against the single handle and the oracle, bit for bit, on inputs with rows of 1..4 ratings."""
    import torch

    sharded = importlib.import_module(pkg.__name__ + ".sharded")
    rng = np.random.default_rng(4100 + seed)
    rows = _random_case(rng, n_users=14 + 5 * seed, n_items=19, n_ratings=110 + 30 * seed, half=(seed % 2 == 1), tiny_rows=3 + seed)
    cut = len(rows) * 4 // 5
    train, test = rows[:cut], rows[cut:]
    if not _no_zero_scale(train):
        pytest.skip("scale() == 0 corner")
    trc, tec = _cols(train), _cols(test)
    assert min(np.bincount(np.unique(trc[0], return_inverse=True)[1])) <= 4
    dev = torch.device("cuda", 0)
    tr = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (np.asarray(trc[0], np.int32), np.asarray(trc[1], np.int32), np.asarray(trc[2], np.float64)))
    te = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (np.asarray(tec[0], np.int32), np.asarray(tec[1], np.int32), np.asarray(tec[2], np.float64)))
    for k in (2, 5):
        want, opreds = oracle.Model(*trc).pipeline(oracle.SIM_COSINE, k).mae(*tec, True)
        for world in (2, 3):
            engines = [kn.Engine(k=k, shard_rank=r, shard_count=world) for r in range(world)]
            views = []
            for e in engines:
                e.fit_device(*tr)
                views.append(sharded.DeviceEngineAdapter(e, dev).shard_tensors())
            for me in range(world):
                for other in range(world):
                    if other != me:
                        lo, hi = views[other]["user_range"]
                        for key in ("user_avg", "user_norm"):
                            views[me][key][lo:hi] = views[other][key][lo:hi]
            torch.cuda.synchronize()
            preds = torch.full((len(tec[0]),), float("nan"), dtype=torch.float64, device=dev)
            total, count = 0.0, 0
            for e in engines:
                e.shard_commit()
                s, c = e.mae_device(kn.PRED_KNN, *te, pred_out=preds)
                total += s
                count += c
                e.close()
            assert count == len(tec[0])
            np.testing.assert_array_equal(preds.cpu().numpy(), opreds)
            assert total / count == pytest.approx(want, abs=1e-13)


def test_k_beyond_1024(kn, oracle, synth):
    """k = 1500 (the re-rank's 4096-entry shortlist tile, the k <= 2048 prediction kernel): neighbours and predictions against
    the oracle on a 2 400-user shape; k = 2049 is refused loudly"""
    d = synth.syn_scaled(2400, 500, 160_000, seed=41, half_stars=True, shuffle=True)
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users[:4000], d.test.items[:4000], d.test.ratings[:4000])
    k = 1500
    p = oracle.Model(*tr).pipeline(oracle.SIM_COSINE, k)
    want, preds = p.mae(*te, True)
    e = kn.Engine(k=k, flags=kn.FLAG_VERIFY_BOUND)
    e.fit(*tr)
    np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
    assert abs(e.mae(kn.PRED_KNN, *te) - want) <= MAE_TOL
    assert e.timings()["max_bound_violation"] <= 0.0
    for u in np.unique(d.test.users[:4000])[::97]:
        ids, sims = e.neighbors(int(u))
        oids, osims = p.neighbors(int(u))
        assert len(ids) == k and ids.tolist() == oids.tolist() and sims.tolist() == osims.tolist()
    e.close()
    e = kn.Engine(k=2049)
    e.fit(*tr)
    with pytest.raises(kn.KnncfError) as ex:
        e.mae(kn.PRED_KNN, *te)
    assert ex.value.status == kn.E_UNSUPPORTED
    e.close()


@pytest.mark.parametrize("case", ["cosine", "jaccard", "k1500", "empty_slices"])
def test_heavy_rows_reranked_as_slices(kn, oracle, synth, syn100k, monkeypatch, case):
    """the heaviest rows of a row block are re-ranked as P slices of their shortlists + a merge (rerank.hip; on by itself for
    blocks of 4096 .. 65 536 rows: the eight-shard test of the ml-25m shape runs it that way): forced on here for the first
    200 rows — or for ALL rows, where most slices of the short shortlists are empty — and compared on every third user with
    the oracle, lists and predictions bit for bit"""
    monkeypatch.setenv("KNNCF_DEBUG_SLICE_ROWS", "100000" if case == "empty_slices" else "200")
    if case == "k1500":
        d = synth.syn_scaled(2400, 500, 160_000, seed=43, half_stars=True, shuffle=True)
        tr = (d.train.users, d.train.items, d.train.ratings)
        te = (d.test.users[:4000], d.test.items[:4000], d.test.ratings[:4000])
        ks, sim_o, sim_k = (1500,), oracle.SIM_COSINE, kn.SIM_COSINE
    else:
        d = syn100k
        tr = (d.train.users, d.train.items, d.train.ratings)
        te = (d.test.users, d.test.items, d.test.ratings)
        ks = (3, 300, 943) if case != "jaccard" else (50,)
        sim_o, sim_k = (oracle.SIM_JACCARD, kn.SIM_JACCARD) if case == "jaccard" else (oracle.SIM_COSINE, kn.SIM_COSINE)
    m = oracle.Model(*tr)
    users = np.unique(tr[0])
    for k in ks:
        e = _engine(kn, tr, k=k, sim=sim_k, flags=kn.FLAG_VERIFY_BOUND)
        p = m.pipeline(sim_o, k)
        want, preds = p.mae(*te, True)
        np.testing.assert_array_equal(e.predict_batch(kn.PRED_KNN, te[0], te[1]), preds)
        assert abs(e.mae(kn.PRED_KNN, *te) - want) <= MAE_TOL
        ids, sims, counts = e.neighbors_batch(users[::3])
        for row, u in enumerate(users[::3]):
            oids, osims = p.neighbors(int(u))
            assert counts[row] == len(oids), f"user {u} k {k}"
            assert ids[row, : counts[row]].tolist() == oids.tolist(), f"user {u} k {k}"
            assert sims[row, : counts[row]].tolist() == osims.tolist()
        t = e.timings()
        assert t["max_bound_violation"] <= 0.0 and t["fallback_rows"] == 0
        e.close()


def test_group_of_one_device_equals_plain_handle(kn, oracle, syn100k):
    """knncf_group_* (one process, RCCL inside the library) with the one GPU this box has: ncclCommInitAll over [0], the
    padded ncclAllGather of the (mean, norm) segments, the collective status and the ncclAllReduce of (sum |err|, rows) all
    execute — with no peer, so the transport itself is not exercised — and every number must equal the plain handle's
    (and the oracle's), bit for bit."""
    d = syn100k
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    k = 40
    want, opreds = oracle.Model(*tr).pipeline(oracle.SIM_COSINE, k).mae(*te, True)
    plain = kn.Engine(k=k)
    plain.fit(*tr)
    g = kn.Group([0], k=k)
    try:
        for _ in range(2):  # a re-fit of the same group reuses communicators and staging buffers
            g.fit(*tr)
            assert g.mae(kn.PRED_KNN, *te) == plain.mae(kn.PRED_KNN, *te)
        assert abs(g.mae(kn.PRED_KNN, *te) - want) <= MAE_TOL
        np.testing.assert_array_equal(g.predict_batch(kn.PRED_KNN, te[0], te[1]), opreds)
        assert g.mae(kn.PRED_BASELINE, *te) == plain.mae(kn.PRED_BASELINE, *te)
        sh = g.shard(0)
        for u in (1, 2, 500, 943):
            a, b = sh.neighbors(u), plain.neighbors(u)
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
        with pytest.raises(kn.KnncfError) as ex:  # errors of a shard surface through the group with the rank named
            g.fit([1, 1, 2], [5, 5, 5], [3.0, 4.0, 2.0])
        assert ex.value.status == kn.E_DUPLICATE and "rank 0" in str(ex.value)
    finally:
        g.close()
        plain.close()
    with pytest.raises(kn.KnncfError):
        kn.Group([0, 0], k=3)  # one shard per GPU


def test_two_handles_two_threads(kn, oracle, syn100k):
    """per-device kernel state is shared by every handle of the process (common.h: PerDeviceState): two handles driven
    from two threads at once must both come out right"""
    import threading

    d = syn100k
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    want = {k: oracle.Model(*tr).pipeline(oracle.SIM_COSINE, k).mae(*te) for k in (10, 60)}
    got, errs = {}, []

    def run(k):
        try:
            e = kn.Engine(k=k)
            e.fit(*tr)
            got[k] = e.mae(kn.PRED_KNN, *te)
            e.close()
        except Exception as ex:  # surfaced below
            errs.append(ex)

    threads = [threading.Thread(target=run, args=(k,)) for k in (10, 60)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for k in (10, 60):
        assert abs(got[k] - want[k]) <= MAE_TOL


def test_fit_errors(kn):
    e = kn.Engine(k=3)
    with pytest.raises(kn.KnncfError) as ex:
        e.fit([1, 1, 2], [5, 5, 5], [3.0, 4.0, 2.0])
    assert ex.value.status == kn.E_DUPLICATE
    with pytest.raises(kn.KnncfError) as ex:
        e.fit([1, 1, 1], [1, 2, 3], [0.5, 1.0, 1.5])
    assert ex.value.status == kn.E_NONFINITE
    with pytest.raises(kn.KnncfError) as ex:
        kn.Engine(k=3).mae(kn.PRED_KNN, [1], [1], [1.0])
    assert ex.value.status == kn.E_STATE
