"""A deliberately naive, literal Python model of the reference's closures.

It re-states shared/predictions.scala (file:line cited per function) with the
*same data structures* the Scala code uses — immutable Set / Map with Scala
2.11 iteration order, per-pair memo maps, lazy views — so that the optimised C
oracle (oracle/knncf_oracle.c: CSR rows, no per-pair maps, owner-tracking memo)
can be cross-checked on small inputs.  Quadratic and slow by design.  Only the
hash functions are shared with the oracle (they have their own known-answer
tests).
"""
import math

M32 = 0xFFFFFFFF


def improve(h):  # HashSet.improve / HashMap.improve, Scala 2.11.12
    h &= M32
    h = (h + (~(h << 9) & M32)) & M32
    h ^= h >> 14
    h = (h + (h << 4)) & M32
    h ^= h >> 10
    return h & M32


def trie_key(h):
    d = [(h >> (5 * i)) & 31 for i in range(7)]
    return (d[0] << 27) | (d[1] << 22) | (d[2] << 17) | (d[3] << 12) | (d[4] << 7) | (d[5] << 2) | d[6]


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & M32


def _mix_last(h, k):
    k = (k * 0xCC9E2D51) & M32
    k = _rotl(k, 15)
    k = (k * 0x1B873593) & M32
    return h ^ k


def _mix(h, k):
    h = _mix_last(h, k)
    h = _rotl(h, 13)
    return (h * 5 + 0xE6546B64) & M32


def tuple2_hash(a, b):  # MurmurHash3.productHash((a, b), 0xcafebabe)
    h = 0xCAFEBABE
    h = _mix(h, a & M32)
    h = _mix(h, b & M32)
    h ^= 2
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def _hash(x):
    return tuple2_hash(*x) if isinstance(x, tuple) else (x & M32)


def scala_order(keys):
    """Iteration order of an immutable Set/Map built by inserting `keys` in order."""
    seen, uniq = set(), []
    for k in keys:
        if k not in seen:
            seen.add(k)
            uniq.append(k)
    if len(uniq) <= 4:  # Set1..Set4 / Map1..Map4
        return uniq
    pos = {k: i for i, k in enumerate(uniq)}
    return sorted(uniq, key=lambda k: (trie_key(improve(_hash(k))), pos[k]))


class SMap:
    """immutable.Map built from pairs (last value wins, first insertion fixes the slot)."""

    def __init__(self, pairs):
        self.d = {}
        keys = []
        for k, v in pairs:
            keys.append(k)
            self.d[k] = v
        self.keys = scala_order(keys)

    def items(self):
        return [(k, self.d[k]) for k in self.keys]

    def get(self, k, default):
        return self.d.get(k, default)

    def group_by(self, f):
        groups, order = {}, []
        for k, v in self.items():
            g = f((k, v))
            if g not in groups:
                groups[g] = []
                order.append(g)
            groups[g].append((k, v))
        return {g: SMap(groups[g]) for g in order}


def scale(x, y):  # :57-61
    if x > y:
        return 5 - y
    elif x < y:
        return y - 1
    return 1


def mean(s):  # :18
    if len(s) > 0:
        acc = s[0]
        for x in s[1:]:
            acc = acc + x
        return acc / len(s)
    return 0.0


def ssum(xs):  # TraversableOnce.sum = foldLeft(0.0)(_ + _)
    acc = 0.0
    for x in xs:
        acc = acc + x
    return acc


def average(ratings):  # :94
    return mean([r[2] for r in ratings])


def group_seq(ratings, idx):  # Seq.groupBy keeps the order inside each group
    g = {}
    for r in ratings:
        g.setdefault(r[idx], []).append(r)
    return g


def users_avg(ratings):  # :113 (lazy mapValues view: recomputed per lookup, same value)
    return {u: average(rs) for u, rs in group_seq(ratings, 0).items()}


def items_avg(ratings):  # :134
    return {i: average(rs) for i, rs in group_seq(ratings, 1).items()}


def compute_normalize_deviation(ratings):  # :155-169
    ua = users_avg(ratings)
    g = average(ratings)
    return SMap([((u, i), (r - ua.get(u, g)) / scale(r, ua.get(u, g))) for (u, i, r) in ratings])


def items_avg_dev(ratings):  # :176-186
    acc = {}
    for (k, v) in compute_normalize_deviation(ratings).items():
        cur = acc.get(k[1], (0.0, 0))
        acc[k[1]] = (v + cur[0], 1 + cur[1])
    return {i: s / c for i, (s, c) in acc.items()}


def compute_prediction(ratings):  # :205-237
    ua = users_avg(ratings)
    dev = items_avg_dev(ratings)
    g = average(ratings)

    def pred(u, i):
        a = ua.get(u, -1.0)
        if a < 0.0:
            return g
        d = dev.get(i, 0.0)
        return a + d * scale(a + d, a)

    return pred


def preprocessed_rating(ratings):  # :470-481
    nd = compute_normalize_deviation(ratings)
    weights = {u: math.sqrt(ssum([v * v for (_, v) in grp.items()]))
               for u, grp in nd.group_by(lambda kv: kv[0][0]).items()}
    out = {}
    for (k, v) in nd.items():
        w = weights.get(k[0], 0.0)
        out[k] = v / w if w != 0 else 0.0
    return out


def similarity_one():  # :400
    return lambda u, v: 1.0


def adjusted_cosine_similarity_function(ratings):  # :407-433
    pre = preprocessed_rating(ratings)
    by_user = group_seq(ratings, 0)
    memo = {}

    def sim(u, v):
        s = memo.get((u, v), -1.0)
        if s < 0.0:
            u_items = scala_order([r[1] for r in by_user.get(u, [])])
            v_items = set(r[1] for r in by_user.get(v, []))
            both = [i for i in u_items if i in v_items]  # intersect = filter, keeps uItems' order
            s = ssum([pre.get((u, i), 0.0) * pre.get((v, i), 0.0) for i in both])
            memo[(u, v)] = s
            memo[(v, u)] = s
        return s

    return sim


def jaccard_coefficient(ratings):  # :440-464
    by_user = group_seq(ratings, 0)

    def coeff(u, v):
        ur, vr = by_user.get(u, []), by_user.get(v, [])
        both = len(set(r[1] for r in ur) & set(r[1] for r in vr))
        den = len(ur) + len(vr) - both
        return float(both) / den if den != 0 else float("nan")

    return coeff


def get_neighbors(ratings, k, sim):  # :596-617
    all_users = scala_order([r[0] for r in ratings])
    memo = {}

    def nn(u):
        got = memo.get(u, [])
        if not got:
            others = [x for x in all_users if x != u]
            scored = [(x, sim(u, x)) for x in others]
            got = sorted(scored, key=lambda t: -t[1])[:k]  # Python's sort is stable, like TimSort
            memo[u] = got
        return got

    return nn


def get_similarity(ratings, k, sim):  # :626-649
    nn = get_neighbors(ratings, k, sim)
    return lambda u1, u2: ssum([s if x == u2 else 0.0 for (x, s) in nn(u1)])


def weighted_sum_deviation(ratings, sim):  # :489-549
    rated_i = group_seq(ratings, 1)
    g = average(ratings)
    ua = users_avg(ratings)

    def wsd(u, i):
        num, den = 0.0, 0.0
        for (xu, _, xr) in rated_i.get(i, []):
            a = ua.get(xu, g)
            d = (xr - a) / scale(xr, a)
            s = sim(u, xu)
            num = num + d * s
            den = den + abs(s)
        return num / den if den > 0 else 0.0

    return wsd


def predictor(ratings, wsd):  # :557-586
    g = average(ratings)
    ua = users_avg(ratings)

    def pred(u, i):
        a = ua.get(u, -1.0)
        if a < 0.0:
            return g
        w = wsd(u, i)
        return a + w * scale(a + w, a)

    return pred


def mae(predict, data):  # :69-86
    acc, n = 0.0, 0
    for (u, i, r) in data:
        acc, n = abs(r - predict(u, i)) + acc, n + 1
    return acc / n


def recommendations(ratings, predict):  # :651-674
    import functools

    def order(x, y):  # sortWith(order): x before y
        return x[0] < y[0] if x[1] == y[1] else x[1] > y[1]

    cache = {}

    def reco(user, n):
        if (user, n) not in cache or not cache[(user, n)]:
            not_rated = scala_order(set(i for (_, i, _) in ratings) - set(i for (u, i, _) in ratings if u == user))
            cand = [(x, predict(user, x)) for x in not_rated]  # notRated.toSeq.map: HashSet iteration order
            cand.sort(key=functools.cmp_to_key(lambda x, y: -1 if order(x, y) else (1 if order(y, x) else 0)))  # stable
            cache[(user, n)] = cand[:n]
        return cache[(user, n)]

    return reco
