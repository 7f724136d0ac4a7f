"""GPU parity of the eager runs of small levels (pytest -m gpu; csrc/factorize_eager.hip): texts with hundreds or thousands of LCP levels
-- long repeats, copied blocks, periodic stretches -- whose levels above the window cut mostly hold a handful of entries.  One
workgroup takes a run of such levels inside one launch: natural candidates + truncated heads per level, the cut run's head and the
right head are the only entries that change lists (model: tests/models/position_space.py::factorize_heads).  The lazy level loop takes
over where a level is too large; both formulations restart from cur[] and the residence marks.  Streams must equal the oracle's
(compressors/lcpcomp/compress/ArraysComp.hpp:36-117 + the rest of the path)."""
import os
import random

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O
from tests import corpus
from tests.test_gpu_levels import stale_levels_text, _ctx_env

pytestmark = pytest.mark.gpu


def _copied_blocks(n, sigma, block, seed, mutate=0.0):
    """random blocks, every third one a copy of an earlier window (textgen.c's DNA recipe with a free block size), optionally with point
    mutations: overlapping copies of copies give long PLCP ramps that cut each other"""
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype=np.uint8)
    ln = 0
    while ln < n:
        m = min(block, n - ln)
        if ln >= block and rng.integers(0, 3) == 0:
            src = int(rng.integers(0, ln - m + 1))
            out[ln:ln + m] = out[src:src + m]
            if mutate:
                k = rng.random(m) < mutate
                out[ln:ln + m][k] = rng.integers(65, 65 + sigma, int(k.sum()), dtype=np.uint8)
        else:
            out[ln:ln + m] = rng.integers(65, 65 + sigma, m, dtype=np.uint8)
        ln += m
    return out.tobytes()


def _texts():
    rng = random.Random(5)
    return [
        ("dna_3M", T.gen_dna(3_000_000, 7).tobytes()),                       # 4096-base copies: ~4 100 levels
        ("copies_s4_b1000", _copied_blocks(600_000, 4, 1000, 1)),
        ("copies_s20_b700_mut", _copied_blocks(500_000, 20, 700, 2, mutate=0.002)),
        ("copies_s3_b5000", _copied_blocks(400_000, 3, 5000, 3)),
        ("stale_levels_900", stale_levels_text(900, 900, False)),
        ("stale_levels_3000s", stale_levels_text(3000, 3000, True)),
        ("planted4_300", corpus.planted(300_000, 4, rng, replen=600)),
        ("runs_and_text", b"ab" * 50_000 + T.gen_english(200_000, 3).tobytes() + b"abc" * 40_000 + T.gen_dna(150_000, 2).tobytes()),
        ("long_run", b"x" + b"a" * 300_000 + T.gen_english(100_000, 8).tobytes()),
        ("fib26", corpus.fib_word(26)[:250_000]),
    ]


TEXTS = _texts()


@pytest.mark.parametrize("name,data", TEXTS, ids=[t[0] for t in TEXTS])
def test_eager_runs_match_oracle(gpu_ctx, name, data):
    text = O.escape(data)
    for thr in (2, 5):
        want, _ = O.lcpcomp_huff_compress(text, thr, 1)
        got, st = gpu_ctx.lcpcomp_compress(text, threshold=thr, flatten=1)
        assert got == want, "%s t=%d: stream differs (%d vs %d bytes; eager phases %d, levels %d)" % (
            name, thr, len(got), len(want), st["eager_phases"], st["eager_levels"])


def test_eager_runs_are_taken_and_equal_the_lazy_loop():
    """the copied-block texts must actually go through eager phases (thousands of levels, hundreds of them with work), and the same context type with TDC_GPU_EAGER=0
    -- every level through the lazy loop -- must give the same stream"""
    data = T.gen_dna(3_000_000, 7).tobytes()
    text = O.escape(data)
    on = _ctx_env({"TDC_GPU_EAGER": "1"})
    off = _ctx_env({"TDC_GPU_EAGER": "0"})
    try:
        a, sa = on.lcpcomp_compress(text, threshold=2, flatten=1)
        b, sb = off.lcpcomp_compress(text, threshold=2, flatten=1)
        assert a == b
        assert sa["eager_phases"] >= 1 and sa["eager_levels"] >= 100, sa      # (levels with work: the runs of levels without a head are skipped 1 024 at a time)
        assert sb["eager_phases"] == 0
        assert sa["factors"] == sb["factors"]
    finally:
        on.close()
        off.close()


def test_eager_with_the_window_pass_forced_to_fail():
    """the window pass fails -> the lists of the low levels are rebuilt from cur[] (after an eager phase there is nothing else to
    rebuild them from) and the lazy loop finishes the text"""
    data = _copied_blocks(500_000, 4, 1500, 9)
    text = O.escape(data)
    want, _ = O.lcpcomp_huff_compress(text, 2, 1)
    ctx = _ctx_env({"TDC_GPU_WINDOW_FORCE_FAIL": "1"})
    try:
        got, st = ctx.lcpcomp_compress(text, threshold=2, flatten=1)
        assert got == want and st["window_pass"] == 2
    finally:
        ctx.close()
