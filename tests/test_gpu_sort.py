"""GPU tests of the device sorts behind the suffix array (pytest -m gpu): the splitter-partition sort (csrc/ssort.hip) and the
LSD radix sort against numpy, on key distributions that stress the partition (heavy keys -> equality leaves, all keys equal,
few distinct keys, sorted / reversed input, keys that differ only in the high or only in the low bits), with every number of
partition levels; then the suffix-array pipeline with the number of levels forced."""
import os

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _cases(n, rng):
    yield "uniform", rng.integers(0, 2**64, size=n, dtype=np.uint64)
    yield "low_bits_only", rng.integers(0, 1 << 20, size=n, dtype=np.uint64)
    yield "high_bits_only", rng.integers(0, 1 << 20, size=n, dtype=np.uint64) << np.uint64(44)
    yield "all_equal", np.full(n, 0x0123456789ABCDEF, dtype=np.uint64)
    yield "two_values", rng.integers(0, 2, size=n, dtype=np.uint64) * np.uint64(0xFFFFFFFFFFFFFFFF)
    few = rng.integers(0, 2**64, size=37, dtype=np.uint64)
    yield "37_values", few[rng.integers(0, 37, size=n)]
    # Zipf-like: half of the pairs share 5 heavy keys, the rest is uniform (the shape of text keys)
    z = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    heavy = rng.integers(0, 2**64, size=5, dtype=np.uint64)
    m = rng.random(n) < 0.5
    z[m] = heavy[rng.integers(0, 5, size=int(m.sum()))]
    yield "heavy_keys", z
    yield "sorted", np.sort(rng.integers(0, 2**64, size=n, dtype=np.uint64))
    yield "reversed", np.sort(rng.integers(0, 2**64, size=n, dtype=np.uint64))[::-1].copy()
    yield "max_keys", np.where(rng.random(n) < 0.3, np.uint64(0xFFFFFFFFFFFFFFFF), rng.integers(0, 2**64, size=n, dtype=np.uint64))


def _check(ctx, keys, algo):
    n = len(keys)
    vals = np.arange(n, dtype=np.uint32)
    k, v = ctx.sort_pairs_u64(keys, vals, algo)
    assert np.array_equal(k, np.sort(keys))
    assert np.array_equal(keys[v], k)                           # every value still travels with its key
    assert np.array_equal(np.sort(v), vals)                     # and the values are a permutation


@pytest.mark.parametrize("levels", [0, 1, 2, 3])
def test_splitter_sort_vs_numpy(levels):
    rng = np.random.default_rng(1234 + levels)
    with T.Context(0, options={"ssort_levels": levels} if levels else None) as ctx:
        sizes = (1, 2, 777, 5000, 70001, (1 << 20) + 123) if levels in (0, 1) else (4096, 70001, (1 << 20) + 123, 3_000_001)
        for n in sizes:
            if levels == 1 and n > 300000:
                continue                                     # one level: at most 256 range leaves of <= 8192 pairs
            for name, keys in _cases(n, rng):
                _check(ctx, keys, 1)


def test_lsd_sort_vs_numpy(gpu_ctx):
    rng = np.random.default_rng(99)
    for n in (1, 3000, (1 << 20) + 5):
        for name, keys in _cases(n, rng):
            _check(gpu_ctx, keys, 0)


@pytest.mark.parametrize("levels", [2, 3])
def test_suffix_array_pipeline_with_forced_levels(levels):
    """the suffix array's initial sort (keys computed from the text) and the sorts of the doubling rounds through the
    splitter sort with 2 / 3 partition levels: stream byte-identical to the oracle's"""
    with T.Context(0, options={"ssort_levels": levels}) as ctx:
        for gen, n, thr in (("english", 1 << 22, 2), ("dna", (1 << 21) + 17, 5)):
            data = (T.gen_english(n, 42) if gen == "english" else T.gen_dna(n, 7)).tobytes()
            text = O.escape(data)
            want, _ = O.lcpcomp_huff_compress(text, thr, 1)
            got, st = ctx.lcpcomp_compress(text, thr, 1)
            assert got == want, (gen, levels)
        # a text that is one long run plus noise: almost all keys of the initial sort are equal
        data = b"a" * 1500000 + bytes(np.random.default_rng(5).integers(97, 101, size=600000, dtype=np.uint8))
        text = O.escape(data)
        sa, isa = ctx.suffix_array(text)
        assert np.array_equal(sa, O.suffix_array(text))
