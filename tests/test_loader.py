"""the multithreaded loader (csrc/loader.cpp, `load` shared/predictions.scala:35-49) and the neighbour
checkpoint (knncf_neighbors_save / _load).  The loader runs on the CPU; the checkpoint needs the GPU."""
import importlib
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kn(pkg):
    mod = importlib.import_module(pkg.__name__ + ".knncf")
    mod.load_library()
    return mod


def _reference_load(text, sep):
    """literal model of load :35-49 on the text of a file"""
    rows = []
    for line in text.split("\n"):
        if line.endswith("\r"):
            line = line[:-1]
        cols = line.split(sep)
        while len(cols) > 1 and cols[-1] == "":  # String.split drops trailing empty strings
            cols.pop()
        cols = [c.strip(" \t\r\n\x0b\x0c\x00") for c in cols]
        try:
            first = cols[0]
            if not first or first.lstrip("+-") == "" or not first.lstrip("+-").isdigit() or not -2**31 <= int(first) <= 2**31 - 1:
                continue
            u = int(first)
        except ValueError:
            continue
        rows.append((u, int(cols[1]), float(cols[2])))  # raises like the reference on a malformed kept line
    return rows


def test_loader_quirks_match_reference_model(kn, tmp_path):
    text = ("userId,movieId,rating,timestamp\n1,10,4.5,111\n 2 , 11 ,3.0,112\nfoo,1,1\n\n3,12,5\n+4,13,0.5,9\n"
            "-5,14,2\r\n99999999999,1,1\n6, 15 ,1e0\n")
    p = tmp_path / "r.csv"
    p.write_text(text)
    u, i, r = kn.load_file(str(p), ",")
    want = _reference_load(text, ",")
    assert list(zip(u.tolist(), i.tolist(), r.tolist())) == want
    assert want[0] == (1, 10, 4.5) and want[-1] == (6, 15, 1.0) and (-5, 14, 2.0) in want
    for bad, what in (("1,2\n", "malformed rating row"), ("1,x,3\n", "malformed rating row"), ("1,2,abc\n", "malformed rating value"),
                      ("1,2,\n", "malformed rating")):
        q = tmp_path / "bad.csv"
        q.write_text("h,h,h\n" + bad)
        with pytest.raises(kn.KnncfError) as e:
            kn.load_file(str(q), ",")
        assert what in str(e.value) and ":2:" in str(e.value)
    with pytest.raises(kn.KnncfError):
        kn.load_file(str(tmp_path / "missing.csv"), ",")


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_loader_is_order_preserving_across_threads(kn, synth, tmp_path, threads):
    """a file big enough to be cut into byte ranges: same rows, same (file) order, whatever the thread count"""
    d = synth.syn_100k()
    p = tmp_path / "u.data"
    with open(p, "w") as f:
        f.write("user\titem\trating\tts\n")
        for rep in range(12):  # ~1.2 M lines, ~20 MB
            for u, i, r in zip(d.train.users, d.train.items, d.train.ratings):
                f.write(f"{u + rep * 1000}\t{i}\t{r:g}\t881250949\n")
    u, i, r = kn.load_file(str(p), "\t", threads)
    n = len(d.train.users)
    assert len(u) == 12 * n
    for rep in (0, 5, 11):
        np.testing.assert_array_equal(u[rep * n:(rep + 1) * n], d.train.users + rep * 1000)
        np.testing.assert_array_equal(i[rep * n:(rep + 1) * n], d.train.items)
        np.testing.assert_array_equal(r[rep * n:(rep + 1) * n], d.train.ratings)


def test_ratings_cache_round_trip_and_invalidation(kn, synth, tmp_path):
    """SURVEY 8f.2, the on-disk cache: the second load of an unchanged file comes from the binary cache and is the same
    triples in the same (file) order; a changed file, another separator, a truncated or a corrupted cache are not used."""
    d = synth.syn_scaled(300, 200, 12_000, seed=5, shuffle=True)
    src = tmp_path / "u.data"
    with open(src, "w") as fh:
        fh.write("userId\titemId\trating\n")
        for u, i, r in zip(d.train.users, d.train.items, d.train.ratings):
            fh.write(f"{u}\t{i}\t{r}\n")
    cache = tmp_path / "u.data.knncf"
    plain = kn.load_file(str(src), "\t")
    info = {}
    first = kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is False and cache.exists()
    second = kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is True
    for a, b, c in zip(plain, first, second):
        assert np.array_equal(a, b) and np.array_equal(a, c) and a.dtype == c.dtype
    # another separator: the cache is stamped with the one it was split on
    other = kn.load_file(str(src), ",", cache=str(cache), info=info)  # (no ',' in the file: no line's column 0 is an Int)
    assert info["from_cache"] is False and len(other[0]) == 0
    back = kn.load_file(str(src), "\t", cache=str(cache), info=info)  # the cache now belongs to the ',' parse: not used
    assert info["from_cache"] is False and np.array_equal(back[0], plain[0])
    kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is True
    # a corrupted and a truncated cache are ignored and rewritten
    blob = bytearray(cache.read_bytes())
    blob[-9] ^= 0x40
    cache.write_bytes(bytes(blob))
    again = kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is False and np.array_equal(again[2], plain[2])
    cache.write_bytes(cache.read_bytes()[:-100])
    kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is False
    kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is True
    # the file changes (one more row): stale cache, re-parsed, re-stamped
    with open(src, "a") as fh:
        fh.write("7\t9\t4.5\n")
    grown = kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is False and len(grown[0]) == len(plain[0]) + 1 and grown[2][-1] == 4.5
    cached = kn.load_file(str(src), "\t", cache=str(cache), info=info)
    assert info["from_cache"] is True and np.array_equal(cached[0], grown[0])
    # no cache path: the plain loader
    assert np.array_equal(kn.load_file(str(src), "\t", cache=None)[1], grown[1])


@pytest.mark.gpu
def test_neighbour_checkpoint_round_trip(kn, oracle, synth, tmp_path):
    """save after a full build, load into a fresh handle fitted on the same rows: identical lists and predictions with
    no neighbour build; a handle fitted on other rows (or another k) refuses the file"""
    d = synth.syn_100k()
    tr = (d.train.users, d.train.items, d.train.ratings)
    te = (d.test.users, d.test.items, d.test.ratings)
    k = 30
    a = kn.Engine(k=k).fit(*tr)
    want = a.predict_batch(kn.PRED_KNN, te[0], te[1])
    path = str(tmp_path / "nbr.bin")
    a.neighbors_save(path)
    assert os.path.getsize(path) > 0
    b = kn.Engine(k=k).fit(*tr)
    b.reset_timings()
    b.neighbors_load(path)
    got = b.predict_batch(kn.PRED_KNN, te[0], te[1])
    np.testing.assert_array_equal(got, want)
    assert b.timings()["gemm_launches"] == 0  # nothing was rebuilt
    users = np.unique(d.train.users)
    for u in users[::50]:
        ia, sa = a.neighbors(int(u))
        ib, sb = b.neighbors(int(u))
        assert ia.tolist() == ib.tolist() and sa.tolist() == sb.tolist()
    c = kn.Engine(k=k).fit(tr[0][:-7], tr[1][:-7], tr[2][:-7])
    with pytest.raises(kn.KnncfError):
        c.neighbors_load(path)
    e = kn.Engine(k=k + 1).fit(*tr)
    with pytest.raises(kn.KnncfError):
        e.neighbors_load(path)
    for eng in (a, b, c, e):
        eng.close()
