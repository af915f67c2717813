"""Pins for the Scala 2.11 collection-order policies the oracle depends on."""
import numpy as np

from tests import scala_model as sm


def test_hashset_known_answers(oracle):
    # the classic REPL outputs: (1 to 10).toSet / Set(1,2,3,4,5) / (1 to 20).toSet
    assert oracle.int_set_order(list(range(1, 11))) == [5, 10, 1, 6, 9, 2, 7, 3, 8, 4]
    assert oracle.int_set_order([1, 2, 3, 4, 5]) == [5, 1, 2, 3, 4]
    assert oracle.int_set_order(list(range(1, 21))) == [5, 10, 14, 20, 1, 6, 9, 13, 2, 17, 12, 7, 3,
                                                        18, 16, 11, 8, 19, 4, 15]


def test_small_sets_keep_insertion_order(oracle):
    assert oracle.int_set_order([9, 3, 7, 1]) == [9, 3, 7, 1]
    assert oracle.int_set_order([42]) == [42]
    assert oracle.int_set_order([]) == []


def test_insertion_order_is_irrelevant_for_tries(oracle):
    rng = np.random.default_rng(0)
    ids = rng.choice(10_000_000, size=500, replace=False).astype(np.int32)
    a = oracle.int_set_order(ids)
    b = oracle.int_set_order(ids[::-1].copy())
    assert a == b


def test_improve_and_trie_key_are_bijections(oracle):
    xs = np.arange(0, 1 << 16, dtype=np.int64) * 65521 % (1 << 32)
    keys = {oracle.trie_key(oracle.improve(int(x))) for x in xs}
    assert len(keys) == len(set(xs.tolist()))
    # negative ids (Int.MinValue ..) behave as their 32-bit pattern
    assert oracle.improve(-1 & 0xFFFFFFFF) == sm.improve(-1)


def test_c_and_python_hashes_agree(oracle):
    rng = np.random.default_rng(1)
    for a, b in rng.integers(-2**31, 2**31 - 1, size=(200, 2)):
        a, b = int(a), int(b)
        assert oracle.tuple2_hash(a, b) == sm.tuple2_hash(a, b)
        assert oracle.improve(a & 0xFFFFFFFF) == sm.improve(a)
        assert oracle.trie_key(a & 0xFFFFFFFF) == sm.trie_key(a & 0xFFFFFFFF)


def test_murmur3_reference_vector():
    # MurmurHash3 x86_32 building blocks against the public algorithm's test vector:
    # hashing the 4-byte little-endian block 0x00000000 with seed 0 gives 0x2362F9DE.
    h = sm._mix(0, 0)
    h ^= 4
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    assert h == 0x2362F9DE


def test_scale(oracle):
    assert oracle.scale(4.0, 3.0) == 2.0   # x > y: 5 - y
    assert oracle.scale(2.0, 3.0) == 2.0   # x < y: y - 1
    assert oracle.scale(3.0, 3.0) == 1.0
    assert oracle.scale(0.5, 1.0) == 0.0   # the half-star division-by-zero corner (N5)
