"""Seeded synthetic MovieLens-shaped rating sets.

MovieLens itself is not available here or on the GPU box, so every test and
bench workload is generated: ml-100k-shaped (943 x 1682, integer ratings),
ml-25m-shaped (162 541 users, 59 047 items with sparse raw ids <= 209 171,
half-star ratings, 25 000 095 ratings) and the 1M x 100k roofline-stress shape.
The generators are deterministic functions of their arguments (numpy
Generator(PCG64(seed))), so the CPU oracle and the GPU engine see identical
inputs on any machine.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class RatingSet:
    users: np.ndarray    # int32 raw ids, file order
    items: np.ndarray    # int32 raw ids
    ratings: np.ndarray  # float64

    def __len__(self):
        return len(self.users)


@dataclass
class Split:
    train: RatingSet
    test: RatingSet
    name: str


def _user_counts(rng, n_users, n_ratings, min_per_user, max_per_user, sigma):
    """Log-normal activity, >= min_per_user each, summing exactly to n_ratings."""
    extra = n_ratings - n_users * min_per_user
    if extra < 0:
        raise ValueError("n_ratings too small for min_per_user")
    w = rng.lognormal(mean=0.0, sigma=sigma, size=n_users)
    cap = max_per_user - min_per_user
    counts = np.zeros(n_users, dtype=np.int64)
    remaining = extra
    for _ in range(64):  # water-filling under the per-user cap
        free = counts < cap
        if remaining <= 0 or not free.any():
            break
        share = np.floor(remaining * w * free / (w * free).sum()).astype(np.int64)
        share = np.minimum(share, cap - counts)
        if share.sum() * 16 < remaining or share.sum() == 0:
            # floors no longer make progress: one each to the heaviest free users
            idx = np.flatnonzero(free)
            idx = idx[np.argsort(-w[idx], kind="stable")][: int(remaining)]
            share = np.zeros_like(counts)
            share[idx] = 1
        counts += share
        remaining = extra - counts.sum()
    if remaining != 0:
        raise ValueError("could not distribute ratings under max_per_user")
    return counts + min_per_user


def make_ratings(n_users, n_items, n_ratings, *, seed, half_stars, max_item_id=None,
                 min_per_user=20, activity_sigma=1.0, zipf_s=1.0, zipf_q=25.0, item_seed=None, first_user=1):
    """(user, item, rating) triples sorted by (user, item) — the order of MovieLens' own files.

    item_seed (optional): the item side (popularity permutation, raw ids, biases) comes from its own generator, so that
    several calls with different `seed` / `first_user` produce disjoint blocks of users over the SAME item catalogue
    (syn_1m builds its 10^6 users as independent blocks, in parallel)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    irng = rng if item_seed is None else np.random.Generator(np.random.PCG64(item_seed))
    max_per_user = max(min_per_user, int(n_items * 0.55))
    counts = _user_counts(rng, n_users, n_ratings, min_per_user, max_per_user, activity_sigma)

    # item popularity: shifted Zipf over a random permutation of the item slots
    w = 1.0 / (np.arange(n_items, dtype=np.float64) + zipf_q) ** zipf_s
    cdf = np.cumsum(w / w.sum())
    cdf[-1] = 1.0
    slot_of_rank = irng.permutation(n_items).astype(np.int64)

    # Draw items WITH replacement and de-duplicate.  d draws give
    # E(d) = sum_i 1 - (1 - p_i)^d distinct items; invert E on a grid so that one round
    # lands every user just above its target, then trim the surplus and top up the deficits.
    p_item = w / w.sum()
    grid = np.unique(np.round(np.geomspace(1, 60.0 * max_per_user, 400)).astype(np.int64))
    log1mp = np.log1p(-p_item)
    e_grid = np.array([n_items - np.exp(d * log1mp).sum() for d in grid])

    # inverse-CDF lookup table (up to 2^24 quantiles): one gather per draw instead of a binary search
    tbl = int(min(1 << 24, max(1 << 16, 256 * n_items)))
    q = (np.arange(tbl, dtype=np.float64) + 0.5) / tbl
    slot_table = slot_of_rank[np.searchsorted(cdf, q, side="left")].astype(np.int64)
    del q

    def draws_for(target):
        return np.ceil(np.interp(target, e_grid, grid.astype(np.float64))).astype(np.int64)

    def sample(user_ids, n_draws):
        u = np.repeat(user_ids.astype(np.int64), n_draws)
        return u * n_items + slot_table[rng.integers(0, len(slot_table), size=len(u))]

    all_users = np.arange(n_users, dtype=np.int64)
    have = np.unique(sample(all_users, draws_for(np.minimum(counts * 1.02 + 2, max_per_user))))
    for _round in range(60):
        got = np.bincount(have // n_items, minlength=n_users)
        short = np.flatnonzero(got < counts)
        if len(short) == 0:
            break
        deficit = counts[short] - got[short]
        extra_keys = sample(short, draws_for(np.minimum(got[short] + deficit * 1.1 + 2, max_per_user))
                            - draws_for(got[short]) + 2)
        have = np.union1d(have, extra_keys)
    else:
        raise RuntimeError("generator did not converge")
    # trim: per user keep `counts` entries chosen at random (one uint64 sort)
    got = np.bincount(have // n_items, minlength=n_users)
    if (got > counts).any():
        mu = have // n_items
        prio = rng.integers(0, 1 << 31, size=len(have), dtype=np.int64)
        order = np.argsort((mu << 31) | prio, kind="stable")
        start = np.concatenate([[0], np.cumsum(got)[:-1]])
        pos_in_user = np.arange(len(have)) - np.repeat(start, got)
        have = np.sort(have[order][pos_in_user < np.repeat(counts, got)])

    users = (have // n_items).astype(np.int64)
    slots = (have % n_items).astype(np.int64)
    if max_item_id is None:
        raw_items = np.arange(1, n_items + 1, dtype=np.int64)
    else:
        raw_items = np.sort(irng.choice(max_item_id, size=n_items, replace=False).astype(np.int64) + 1)
    # popularity rank -> bias: popular items rate slightly higher, like MovieLens
    rank_of_slot = np.empty(n_items, dtype=np.int64)
    rank_of_slot[slot_of_rank] = np.arange(n_items)
    item_bias = irng.normal(0.0, 0.45, n_items) + 0.35 * (1.0 - rank_of_slot / n_items) - 0.15
    user_bias = rng.normal(0.0, 0.45, n_users)
    raw = 3.35 + user_bias[users] + item_bias[slots] + rng.normal(0.0, 0.95, len(users))
    if half_stars:
        ratings = np.clip(np.round(raw * 2.0) / 2.0, 0.5, 5.0)
        # SURVEY N5 guard: scale() is 0 only for a user whose mean is exactly 1 with a rating < 1
        s = np.bincount(users, weights=ratings, minlength=n_users)
        bad = np.flatnonzero(s == counts.astype(np.float64))
        if len(bad):
            m = np.isin(users, bad) & (ratings < 1.0)
            ratings[m] = 1.0
    else:
        ratings = np.clip(np.round(raw), 1.0, 5.0)
    return RatingSet((users + first_user).astype(np.int32), raw_items[slots].astype(np.int32),
                     ratings.astype(np.float64))


def split_80_20(rs, *, seed, shuffle=False, name="syn"):
    """Random 80/20 split; every user keeps at least one training rating."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    n = len(rs)
    is_test = np.zeros(n, dtype=bool)
    is_test[rng.permutation(n)[: n // 5]] = True
    # users whose ratings all went to test get their first rating back
    order = np.argsort(rs.users, kind="stable")
    su = rs.users[order]
    first = np.concatenate([[True], su[1:] != su[:-1]])
    n_train_u = np.add.reduceat((~is_test[order]).astype(np.int64), np.flatnonzero(first))
    empties = np.flatnonzero(n_train_u == 0)
    if len(empties):
        is_test[order[np.flatnonzero(first)[empties]]] = False
    tr = np.flatnonzero(~is_test)
    te = np.flatnonzero(is_test)
    if shuffle:
        tr = rng.permutation(tr)
        te = rng.permutation(te)
    pick = lambda ix: RatingSet(rs.users[ix].copy(), rs.items[ix].copy(), rs.ratings[ix].copy())
    return Split(pick(tr), pick(te), name)


def syn_100k(seed=2, shuffle=False):
    """ml-100k u2.base/u2.test shape: 943 x 1682, 100 000 integer ratings, 80/20."""
    rs = make_ratings(943, 1682, 100_000, seed=seed, half_stars=False, min_per_user=20,
                      activity_sigma=1.0, zipf_s=0.9, zipf_q=12.0)
    return split_80_20(rs, seed=seed, shuffle=shuffle, name="syn-100k")


def syn_25m(seed=25, shuffle=False):
    """ml-25m r2.train/r2.test shape: 162 541 users, 59 047 items (raw ids <= 209 171),
    25 000 095 half-star ratings, 80/20."""
    rs = make_ratings(162_541, 59_047, 25_000_095, seed=seed, half_stars=True, max_item_id=209_171,
                      min_per_user=20, activity_sigma=1.15, zipf_s=1.05, zipf_q=18.0)
    return split_80_20(rs, seed=seed, shuffle=shuffle, name="syn-25m")


def syn_blocks(n_users, n_items, n_ratings, *, seed, n_blocks, half_stars, name, workers=0, **kw):
    """A big set as n_blocks independent blocks of users over one item catalogue (item_seed = seed), generated and split
    80/20 block by block on a thread pool (numpy releases the GIL in its sorts and gathers) and concatenated in user
    order.  Deterministic in (arguments, n_blocks); independent of the number of workers."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    if n_users % n_blocks or n_ratings % n_blocks:
        raise ValueError("n_users and n_ratings must be multiples of n_blocks")
    bu, bn = n_users // n_blocks, n_ratings // n_blocks

    def one(b):
        rs = make_ratings(bu, n_items, bn, seed=seed + 7919 * (b + 1), half_stars=half_stars, item_seed=seed,
                          first_user=1 + b * bu, **kw)
        sp = split_80_20(rs, seed=seed + 7919 * (b + 1))
        return sp.train, sp.test

    if workers <= 0:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
    workers = max(1, min(workers, n_blocks, 32))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        parts = list(ex.map(one, range(n_blocks)))
    cat = lambda j: RatingSet(np.concatenate([getattr(p[j], "users") for p in parts]),
                              np.concatenate([getattr(p[j], "items") for p in parts]),
                              np.concatenate([getattr(p[j], "ratings") for p in parts]))
    return Split(cat(0), cat(1), name)


def syn_1m(seed=1000, n_blocks=64, workers=0):
    """Roofline-stress shape (BASELINE config 5): 1 M users x 100 k items, 250 M integer ratings, 80/20 — built as 64
    blocks of 15 625 users (3 906 250 ratings each) over one item catalogue."""
    return syn_blocks(1_000_000, 100_000, 250_000_000, seed=seed, n_blocks=n_blocks, half_stars=False, name="syn-1M",
                      workers=workers, min_per_user=20, activity_sigma=1.0, zipf_s=1.0, zipf_q=25.0)


def syn_scaled(n_users, n_items, n_ratings, seed, half_stars=True, shuffle=False, max_item_id=None):
    """A smaller ml-25m-like set for parity tests (same generator, free shape)."""
    rs = make_ratings(n_users, n_items, n_ratings, seed=seed, half_stars=half_stars,
                      max_item_id=max_item_id, min_per_user=min(20, max(1, n_ratings // n_users)),
                      activity_sigma=1.0, zipf_s=1.0, zipf_q=10.0)
    return split_80_20(rs, seed=seed, shuffle=shuffle, name=f"syn-{n_users}x{n_items}")
