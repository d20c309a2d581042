"""The engine's device-wide primitives (longqc_amd/csrc/kernels_isort.hpp): the stable radix sort that stands where the
reference sorts a bucket's minimizers (index.c:150-201; stable on (hash, y) pairs emitted in ascending y == the reference's
"occurrences of a hash in ascending y") and the exclusive scan.  Checked against numpy's stable sort / cumsum on the test
emulator (no GPU) and, marked gpu, on the device through the C ABI."""
import numpy as np
import pytest

from longqc_amd import api


def _engine(lib):
    p = api.Params()
    lib.lqcov_params_default(p)
    return api.Engine(p, 0, lib=lib)


def _keys(rng, n, bits, kind):
    if kind == "uniform":
        return rng.integers(0, 1 << bits, size=n, dtype=np.uint64)
    if kind == "skewed":                                     # minimizer-like: the minimum of a few hashes, and heavy repeats
        k = rng.integers(0, 1 << bits, size=(n, 5), dtype=np.uint64).min(axis=1)
        if n:
            k[rng.random(n) < 0.2] = k[0]
        return k
    if kind == "one":
        return np.full(n, (1 << bits) - 1, dtype=np.uint64)
    raise ValueError(kind)


def _check_sort(eng, rng, n, bits, key_bytes, kind, pairs=True):
    keys = _keys(rng, n, bits, kind)
    vals = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(7) if pairs else None     # ascending, like y
    k, v = eng.debug_sort_pairs(keys, vals, bits, key_bytes)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order])
    if pairs:
        assert np.array_equal(v, vals[order])


SIZES = [0, 1, 63, 64, 65, 3839, 3840, 3841, 5119, 5120, 5121, 11000, 40000]


def _run_all(eng, sizes, big):
    rng = np.random.default_rng(11)
    for n in sizes:
        for bits, kb in ((24, 4), (30, 4), (32, 4), (38, 8), (7, 4), (64, 8)):
            _check_sort(eng, rng, n, bits, kb, "uniform")
        _check_sort(eng, rng, n, 24, 4, "skewed")
        _check_sort(eng, rng, n, 24, 4, "one")
        _check_sort(eng, rng, n, 32, 4, "skewed", pairs=False)
        c = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        want = np.zeros(n, dtype=np.uint64); want[1:] = np.cumsum(c.astype(np.uint64))[:-1]   # (all in uint64)
        assert np.array_equal(eng.debug_scan(c), want)
    if big:
        _check_sort(eng, rng, big, 24, 4, "skewed")
        c = rng.integers(0, 1 << 20, size=big, dtype=np.uint64).astype(np.uint32)
        want = np.zeros(big, dtype=np.uint64); want[1:] = np.cumsum(c.astype(np.uint64))[:-1]
        assert np.array_equal(eng.debug_scan(c), want)


@pytest.mark.parametrize("ranges", ["1", "8", "3"])
def test_emulated_radix_sort_and_scan(emu_lib, monkeypatch, ranges):
    """every size class (empty, below / at / above a wave, a tile, many tiles), every key width the index uses (2k = 24, 30,
    32 bits in 4-byte keys, 38 in 8-byte keys), tiles dealt in one range and in one range per XCD (with blocks that take
    tiles of other ranges: the emulator's blocks run one after the other)"""
    monkeypatch.setenv("LQCOV_IS_RANGES", ranges); monkeypatch.setenv("LQCOV_IS_RANGES_MIN_TILES", "2")
    _run_all(_engine(emu_lib), SIZES, 0)


@pytest.mark.parametrize("order", ["reverse", "random:5"])
def test_emulated_radix_sort_whatever_the_thread_order(emu_lib, monkeypatch, order):
    monkeypatch.setenv("LQ_EMU_ORDER", order); monkeypatch.setenv("LQCOV_IS_RANGES_MIN_TILES", "2")
    _run_all(_engine(emu_lib), [65, 3841, 11000], 0)


@pytest.mark.gpu
@pytest.mark.parametrize("ranges", ["1", "8"])
def test_gpu_radix_sort_and_scan(gpu_lib, monkeypatch, ranges):
    """the same on the device, plus 30 M pairs: thousands of tiles in flight, look-back across XCDs"""
    monkeypatch.setenv("LQCOV_IS_RANGES", ranges)
    _run_all(_engine(gpu_lib), SIZES + [1000003], 30_000_000)
