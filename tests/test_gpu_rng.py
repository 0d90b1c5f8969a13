"""Device MT19937 + numpy-legacy draws (csrc/mt19937_device.h) vs numpy RandomState and the
golden index vectors the reference's her.py produced."""
import numpy as np
import pytest

from conftest import load_golden
from gpu_common import DeviceEpisodeBuffer, fresh_rng, state_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 125, 2**32 - 1])
def test_seed_state(seed):
    rs = np.random.RandomState(seed)
    assert state_equal(fresh_rng(seed), *rs.get_state()[1:3])


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 100, 128, 129, 5000, 2**20 + 1, 2**31 - 1])
def test_randint_stream(n):
    rs, dev = np.random.RandomState(7), fresh_rng(7)
    for size in (1, 5, 255, 256, 257, 700, 1300, 5000):
        assert np.array_equal(rs.randint(0, n, size), dev.randint(0, n, size)), (n, size)
        assert state_equal(dev, *rs.get_state()[1:3]), (n, size)


def test_uniform_and_interleaving():
    rs, dev = np.random.RandomState(99), fresh_rng(99)
    for size in (1, 3, 255, 256, 257, 312, 1000, 4096):
        assert np.array_equal(rs.uniform(size=size), dev.uniform(size)), size
        assert np.array_equal(rs.randint(0, 100, 333), dev.randint(0, 100, 333))
        assert state_equal(dev, *rs.get_state()[1:3])


def test_block_boundary_positions():
    # consume exactly to a block boundary (pos == 624), and start from every kind of position
    rs, dev = np.random.RandomState(3), fresh_rng(3)
    assert np.array_equal(rs.randint(0, 2**16, 624), dev.randint(0, 2**16, 624))   # power of two: no rejection
    assert rs.get_state()[2] == 624 and state_equal(dev, *rs.get_state()[1:3])
    assert np.array_equal(rs.randint(0, 2**16, 1248), dev.randint(0, 2**16, 1248))
    assert state_equal(dev, *rs.get_state()[1:3])
    for pos_words in (1, 622, 623, 625):
        rs.randint(0, 2**16, pos_words); dev.randint(0, 2**16, pos_words)
        assert np.array_equal(rs.uniform(size=700), dev.uniform(700))
        assert state_equal(dev, *rs.get_state()[1:3])


def test_set_get_state_roundtrip_with_numpy():
    rs = np.random.RandomState(2024)
    rs.uniform(size=1234)
    dev = fresh_rng()
    dev.set_state(rs.get_state())
    assert np.array_equal(rs.randint(0, 5000, 3000), dev.randint(0, 5000, 3000))
    rs2 = np.random.RandomState()
    rs2.set_state(dev.get_state())
    assert np.array_equal(rs2.uniform(size=10), rs.uniform(size=10))


def test_state_hand_off_keeps_numpys_cached_gaussian():
    """learn() hands numpy's global stream to the device for the learner phase and takes it back.  After an ODD number of
    randn draws numpy holds a cached second normal (has_gauss = 1); randint / random_sample never touch it, so it must
    survive the round trip -- otherwise an env with an odd number of exploration normals per cycle leaves the reference's
    stream (ddpg_agent.py:177-183 share np.random with her.py:24-31)."""
    rs = np.random.RandomState(99)
    twin = np.random.RandomState(99)
    rs.randn(3); twin.randn(3)                          # odd count: one normal is cached
    assert rs.get_state()[3] == 1
    dev = fresh_rng()
    dev.set_state(rs.get_state())
    got = dev.randint(0, 5000, 256)                     # the learner phase draws indices on the device ...
    assert np.array_equal(got, twin.randint(0, 5000, 256))
    st = dev.get_state()
    assert st[3] == 1 and st[4] == rs.get_state()[4]
    rs.set_state(st)                                    # ... and numpy continues where the reference would be
    assert np.array_equal(rs.randn(4), twin.randn(4))
    dev.seed(5)
    assert dev.get_state()[3] == 0


def test_randint_errors_like_numpy():
    dev = fresh_rng(0)
    with pytest.raises(ValueError):
        dev.randint(0, 0, 4)
    with pytest.raises(ValueError):
        dev.seed(2**32)


def test_rng_kat_golden_through_sampler():
    """F1: (e, t, her, future_t) and the final stream state for 38 (seed, N, B, k) cases."""
    g = load_golden("rng_kat.npz")
    bufs = {}
    for tag in g["cases"]:
        tag = str(tag)
        seed, n, B, k = (int(x[1:]) for x in tag.split("_"))
        if n not in bufs:
            b = DeviceEpisodeBuffer(n, 100, 1, 1, 1)
            z = np.zeros
            b.store(fresh_rng(0), [z((n, 101, 1)), z((n, 101, 1)), z((n, 100, 1)), z((n, 100, 1))])
            bufs[n] = b
        dev = fresh_rng(seed)
        _, idx = bufs[n].sample(dev, B, 1 - 1.0 / (1 + k), 0.0025, with_indices=True)
        her = g[tag + "_her"]
        assert np.array_equal(idx["e"], g[tag + "_e"]), tag
        assert np.array_equal(idx["t"], g[tag + "_t"]), tag
        assert np.array_equal(idx["her"], her), tag
        assert np.array_equal(idx["future_t"][her], g[tag + "_future_t"][her]), tag
        assert np.all((idx["future_t"] >= idx["t"] + 1) & (idx["future_t"] <= 100)), tag
        assert state_equal(dev, g[tag + "_key"], g[tag + "_pos"]), tag
