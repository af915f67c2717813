"""Randomised GPU-vs-oracle parity sweep (run on an MI355X): many small shapes, random engine flags / head widths / k /
similarity (adjusted cosine, Jaccard) / GEMM form (symmetric, row blocks), then a few shapes that cross the 16 384-column
tile of select.hip (sampled users).
usage: python scripts/fuzz_parity.py [first_seed] [count] [scale] [big_cases]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_oracle_semantics import _cols, _no_zero_scale, _random_case  # noqa: E402

kn = importlib.import_module("movie-recommender-system_amd.knncf")
oracle = importlib.import_module("oracle.knncf_oracle")
kn.load_library()

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # multiplies the shape ranges
big_cases = int(sys.argv[4]) if len(sys.argv) > 4 else 0
synth = importlib.import_module("movie-recommender-system_amd.synth")
bad = 0
done = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n_users = int(rng.integers(3, 400 * scale))
    n_items = int(rng.integers(6, 300 * scale))
    n_ratings = int(rng.integers(max(n_users, 10), min(n_users * n_items // 2 + 11, 12000 * scale * scale)))
    rows = _random_case(rng, n_users=n_users, n_items=n_items, n_ratings=n_ratings, half=bool(rng.integers(0, 2)),
                        tiny_rows=int(rng.integers(0, 4)))
    cut = len(rows) * 4 // 5
    train, test = rows[:cut], rows[cut:]
    if len(train) < 4 or not _no_zero_scale(train):
        continue
    tr = tuple(np.asarray(c) for c in _cols(train))
    te = tuple(np.asarray(c) for c in _cols(test))
    if len(te[0]) == 0:
        continue
    k = int(rng.choice([1, 2, 5, 17, 64, 300, 1000]))
    flags = int(rng.choice([0, 1, 4, 8, 5, 9, 12, 13, 2, 3, 7, 10]))
    head = int(rng.choice([0, 1, 7, 64, 0xFFFFFFFF]))
    sim = int(rng.choice([oracle.SIM_COSINE, oracle.SIM_COSINE, oracle.SIM_JACCARD]))  # (same codes on both sides)
    if rng.integers(0, 2):
        os.environ["KNNCF_DEBUG_NO_SYMMETRIC_GEMM"] = "1"
    else:
        os.environ.pop("KNNCF_DEBUG_NO_SYMMETRIC_GEMM", None)
    m = oracle.Model(*tr)
    p = m.pipeline(sim, k)
    try:
        e = kn.Engine(k=k, similarity=sim, flags=flags, head_items=head).fit(*tr)
        want, preds = p.mae(*te, True)
        got = e.predict_batch(kn.PRED_KNN, te[0], te[1])
        ok = np.array_equal(got, preds) and abs(e.mae(kn.PRED_KNN, *te) - want) <= 1e-9
        users = sorted(set(tr[0].tolist()))
        p2 = m.pipeline(sim, k)
        e.reset_neighbors()
        for u in users[:: max(1, len(users) // 25)]:
            ids, sims = e.neighbors(int(u))
            oi, os_ = p2.neighbors(int(u))
            ok = ok and ids.tolist() == oi.tolist() and sims.tolist() == os_.tolist()
        if flags & 1:
            ok = ok and e.timings()["max_bound_violation"] <= 0.0
        u0 = int(users[int(rng.integers(0, len(users)))])
        ri, rp = e.recommend(kn.PRED_KNN, u0, 5)
        oi, op = p2.recommend(u0, 5)  # the same closures, hence the same memo history, as the engine's
        ok = ok and ri.tolist() == oi.tolist() and rp.tolist() == op.tolist()
        e.close()
    except Exception as ex:  # noqa: BLE001
        ok = False
        print("seed", seed, "exception", repr(ex))
    done += 1
    if done % 20 == 0:
        print(f"... {done} cases, {bad} mismatches so far", flush=True)  # (a silent GPU command is taken to be hung)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, dict(n_users=n_users, n_items=n_items, n_ratings=n_ratings, k=k, flags=flags, head=head, sim=sim,
                                         sym="KNNCF_DEBUG_NO_SYMMETRIC_GEMM" not in os.environ))
print(f"fuzz: {done} cases, {bad} mismatches", flush=True)

# shapes that span several column tiles / row blocks: sampled users and their predictions
for j in range(big_cases):
    rng = np.random.default_rng(77_000 + first + j)
    n_users = int(rng.integers(16_400, 60_000))
    n_items = int(rng.integers(500, 4_000))
    n_ratings = int(n_users * rng.integers(25, 60))
    sim = int(rng.choice([oracle.SIM_COSINE, oracle.SIM_JACCARD]))
    k = int(rng.choice([10, 100, 300]))
    flags = int(rng.choice([1, 5, 9]))
    head = int(rng.choice([0, 0, 128, 320]))
    if rng.integers(0, 2):
        os.environ["KNNCF_DEBUG_NO_SYMMETRIC_GEMM"] = "1"
    else:
        os.environ.pop("KNNCF_DEBUG_NO_SYMMETRIC_GEMM", None)
    ws = int(rng.choice([0, 0, 1 << 28]))  # sometimes many small row blocks
    d = synth.syn_scaled(n_users, n_items, n_ratings, seed=int(rng.integers(1, 1 << 30)), half_stars=bool(rng.integers(0, 2)),
                         shuffle=bool(rng.integers(0, 2)))
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    ok = True
    try:
        e = kn.Engine(k=k, similarity=sim, flags=flags, head_items=head, workspace_bytes=ws).fit(*tr)
        preds = e.predict_batch(kn.PRED_KNN, te[0], te[1])
        p = oracle.Model(*tr).pipeline(sim, k)
        users = np.unique(tr[0])
        sample = users[:: max(1, len(users) // 16)]
        ids, sims, counts = e.neighbors_batch(sample)
        for row, u in enumerate(sample):
            oi, os_ = p.neighbors(int(u))
            ok = ok and ids[row, :counts[row]].tolist() == oi.tolist() and sims[row, :counts[row]].tolist() == os_.tolist()
        mask = np.isin(te[0], sample)
        _, opreds = p.mae(te[0][mask], te[1][mask], te[2][mask], True)
        ok = ok and np.array_equal(preds[mask], opreds)
        t = e.timings()
        ok = ok and t["max_bound_violation"] <= 0.0
        e.close()
    except Exception as ex:  # noqa: BLE001
        ok = False
        print("big case", j, "exception", repr(ex))
    print("big case", j, dict(n_users=n_users, n_items=n_items, n_ratings=n_ratings, k=k, flags=flags, head=head, sim=sim, ws=ws,
                              sym="KNNCF_DEBUG_NO_SYMMETRIC_GEMM" not in os.environ), "ok" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
sys.exit(1 if bad else 0)
