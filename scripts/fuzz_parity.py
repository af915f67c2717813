"""Randomised GPU-vs-oracle parity sweep (run on an MI355X): many small shapes, random engine flags / head widths / k.
usage: python scripts/fuzz_parity.py [first_seed] [count] [scale]"""
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
    m = oracle.Model(*tr)
    p = m.pipeline(oracle.SIM_COSINE, k)
    try:
        e = kn.Engine(k=k, flags=flags, head_items=head).fit(*tr)
        want, preds = p.mae(*te, True)
        got = e.predict_batch(kn.PRED_KNN, te[0], te[1])
        ok = np.array_equal(got, preds) and abs(e.mae(kn.PRED_KNN, *te) - want) <= 1e-9
        users = sorted(set(tr[0].tolist()))
        p2 = m.pipeline(oracle.SIM_COSINE, k)
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
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, dict(n_users=n_users, n_items=n_items, n_ratings=n_ratings, k=k, flags=flags, head=head))
print(f"fuzz: {done} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
