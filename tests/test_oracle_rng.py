"""Oracle RNG twin (oracle/mt19937_legacy.c) vs numpy's legacy RandomState and vs the
golden index vectors the reference's her.py produced (tests/golden/rng_kat.npz)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.her_replay import draw_her_indices, future_probability
from oracle.mt_c import MT


@pytest.mark.parametrize("seed", [0, 1, 125, 2**32 - 1])
def test_seed_state_matches_numpy(seed):
    rs = np.random.RandomState(seed)
    key, pos = rs.get_state()[1:3]
    m = MT(seed)
    k2, p2 = m.get_state()
    assert pos == p2 == 624 and np.array_equal(key, k2)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 100, 128, 129, 5000, 2**20 + 1])
def test_randint_stream(n):
    rs = np.random.RandomState(7)
    m = MT(7)
    for size in (1, 5, 700, 1300):
        assert np.array_equal(rs.randint(0, n, size), m.randint(0, n, size))
    assert np.array_equal(rs.get_state()[1], m.get_state()[0]) and rs.get_state()[2] == m.get_state()[1]


def test_uniform_stream_and_interleave():
    rs = np.random.RandomState(99)
    m = MT(99)
    assert np.array_equal(rs.uniform(size=1000), m.random_sample(1000))
    assert np.array_equal(rs.randint(0, 100, 333), m.randint(0, 100, 333))
    assert np.array_equal(rs.uniform(size=3), m.random_sample(3))
    assert rs.get_state()[2] == m.get_state()[1]


def test_words_consumed_matches_survey_probe():
    # seed 125, B=256: N=7 draw uses 292 words, then T=100 uses 324
    m = MT(125)
    m.randint(0, 7, 256)
    assert m.consumed() == 292
    m.randint(0, 100, 256)
    assert m.consumed() == 292 + 324


def test_set_state_mid_block():
    rs = np.random.RandomState(5)
    rs.uniform(size=100)
    st = rs.get_state()
    m = MT()
    m.set_state(st[1], st[2])
    assert np.array_equal(rs.randint(0, 5000, 2000), m.randint(0, 5000, 2000))


def test_rng_kat_golden():
    g = load_golden("rng_kat.npz")
    for tag in g["cases"]:
        tag = str(tag)
        seed, n, B, k = (int(x[1:]) for x in tag.split("_"))
        fp = future_probability("future", k)
        # numpy-backed oracle
        rs = np.random.RandomState(seed)
        e, t, her, fut = draw_her_indices(rs, n, 100, B, fp)
        # C twin
        m = MT(seed)
        ce, ct, cher, cfut, _, _ = m.her_draw(n, 100, B, fp)
        for a, c in ((e, ce), (t, ct), (her, cher), (fut, cfut)):
            assert np.array_equal(a, c), tag
        assert np.array_equal(e, g[tag + "_e"]) and np.array_equal(t, g[tag + "_t"]), tag
        assert np.array_equal(her, g[tag + "_her"]), tag
        assert np.array_equal(fut[her], g[tag + "_future_t"][her]), tag
        assert np.all((fut >= t + 1) & (fut <= 100))
        key, pos = m.get_state()
        assert np.array_equal(key, g[tag + "_key"]) and pos == int(g[tag + "_pos"]), tag


def test_philox4x32_10_known_answers():
    """The fast draw's generator (opt-in rng_mode of SURVEY 8b) against Random123's published known-answer vectors
    (kat_vectors: philox4x32 10), and the draw built on it: in range, reproducible, call-dependent."""
    from oracle.her_replay import draw_her_indices_fast, philox4x32_10
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], *key)
        assert tuple(int(x[0]) for x in got) == want
    e, t, her, fut = draw_her_indices_fast(5000, 100, 4096, 0.8, seed=125, call=7)
    assert e.min() >= 0 and e.max() < 5000 and t.min() >= 0 and t.max() < 100
    assert np.all(fut > t) and np.all(fut <= 100) and 0.75 < her.mean() < 0.85
    again = draw_her_indices_fast(5000, 100, 4096, 0.8, seed=125, call=7)
    other = draw_her_indices_fast(5000, 100, 4096, 0.8, seed=125, call=8)
    assert all(np.array_equal(a, b) for a, b in zip((e, t, her, fut), again)) and not np.array_equal(e, other[0])
