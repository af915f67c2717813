"""knncf_load_personal (csrc/loader.cpp) against the one input file the reference itself holds: data/personal.csv, committed
unchanged as tests/golden/personal.csv (data, not source).  recommend/Recommender.scala:40-54 turns it into the ratings
of user 944 (`Rating(944, cols(0).toInt, cols(2).toDouble)`, header and unrated rows filtered by `rating != 0`) and
the id -> title map.  Expected values below were derived from the file with Python's csv module, independently of the
loader.  Runs on the CPU: the loader needs no GPU."""
import csv
import importlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "personal.csv")


@pytest.fixture(scope="module")
def kn(pkg):
    mod = importlib.import_module(pkg.__name__ + ".knncf")
    mod.load_library()
    return mod


def test_personal_csv_of_the_reference(kn):
    names, (users, items, ratings) = kn.load_personal(GOLDEN)
    # every line is a row of the name list: the header as (0, "header") + 1682 movies (:51-54)
    assert len(names) == 1683 and names[0] == (0, "header")
    assert [i for i, _ in names[1:]] == list(range(1, 1683))
    assert dict(names)[1] == "Toy Story (1995)" and dict(names)[3] == "Four Rooms (1995)"
    # 76 rated movies, all by user 944, rating sum 236 (21 x 3, 18 x 5, 13 x 1, 13 x 2, 11 x 4)
    assert len(users) == len(items) == len(ratings) == 76
    assert set(users.tolist()) == {944}
    assert ratings.sum() == 236.0
    assert sorted(np.unique(ratings, return_counts=True)[1].tolist()) == [11, 13, 13, 18, 21]
    assert items[:10].tolist() == [1, 2, 22, 28, 50, 56, 64, 69, 71, 82]
    # the same through an independent parser
    with open(GOLDEN, newline="") as f:
        rows = list(csv.reader(f))
    want = [(int(r[0]), float(r[2])) for r in rows[1:] if len(r) > 2 and r[2].strip() and float(r[2]) != 0]
    assert list(zip(items.tolist(), ratings.tolist())) == want
    # titles are cols(1) of a plain split(","): the one quoted title of the file keeps its quotes (:53)
    quoted = {177: '"The Good the Bad and the Ugly"'}
    assert [n for _, n in names[1:]] == [quoted.get(int(r[0]), r[1].strip()) for r in rows[1:]]


def test_personal_quirks(kn, tmp_path):
    p = tmp_path / "p.csv"
    # header; rated; unrated (trailing empty dropped by String.split); explicit 0 is filtered; spaces are trimmed;
    # another user id is honoured; CRLF
    p.write_text("id,title,rating\n1,A,5\n2,B,\n3,C,0\n 4 , D , 2.5 \n5,E\r\n6,F,1\r\n")
    names, (u, i, r) = kn.load_personal(str(p), user=7)
    assert names == [(0, "header"), (1, "A"), (2, "B"), (3, "C"), (4, "D"), (5, "E"), (6, "F")]
    assert (u.tolist(), i.tolist(), r.tolist()) == ([7, 7, 7], [1, 4, 6], [5.0, 2.5, 1.0])
    # the reference throws on these: loud failures with the line number
    for bad, what in (("id,title,rating\nx,A,5\n", ":2:"), ("1,A,five\n", ":1:"), ("1\n", ":1:"), ("1,A, \n", ":1:")):
        p.write_text(bad)
        with pytest.raises(kn.KnncfError) as ex:
            kn.load_personal(str(p))
        assert ex.value.status == kn.E_INVALID and what in str(ex.value)
    with pytest.raises(kn.KnncfError):
        kn.load_personal(str(tmp_path / "missing.csv"))
