"""GPU parity of the one-workgroup levels of the factorizer with MANY survivors (pytest -m gpu): csrc/factorize.hip runs a level whose
list holds up to 2 048 alive entries in one 256-thread workgroup and up to 4 096 in a 512-thread instance of the same kernel; more go
through the multi-launch path.  The text below is built so that fifty consecutive levels (101 .. 150) each hold K stale entries: K
blocks U_i V_i W_i where V_i W_i (300 bytes) and U_i V_i (150 bytes) both occur once more in a dictionary part, each followed by a
smaller byte -- PLCP is 300 at the start of V_i and 150 - j at byte j of U_i; the factors of level 300 truncate the latter to 50 - j,
which leaves them as stale entries of the lists 150 - j and sends all K of a level to ONE target list (the duplicate-target path of
the push ordering).  Streams must equal the oracle's (compressors/lcpcomp/compress/ArraysComp.hpp:36-117) for every instance choice."""
import os

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def stale_levels_text(K, seed):
    rng = np.random.default_rng(seed)
    letters = lambda k: rng.integers(ord("b"), ord("z") + 1, k, dtype=np.uint8).tobytes()
    main, dic = [], []
    for _ in range(K):
        U, V, W = letters(50), letters(100), letters(200)
        main.append(letters(int(rng.integers(20, 60))) + U + V + W + b"z" + letters(8))
        dic.append(V + W + b"a" + letters(int(rng.integers(5, 30))))       # 'a' < every byte that follows the main copy
        dic.append(U + V + b"a" + letters(int(rng.integers(5, 30))))
    for L in range(60, 170):                                                  # a live entry in every level, so that the levels are not
        R = letters(L)                                                        # collected in one go (no-live runs are batched)
        main.append(letters(int(rng.integers(10, 30))) + R + b"z" + letters(6))
        dic.append(R + b"a" + letters(int(rng.integers(5, 30))))
    order = rng.permutation(len(dic))
    return b"".join(main) + b"".join(dic[int(i)] for i in order)


def _ctx_env(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return T.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("K", [900, 3000, 4600])
def test_levels_with_thousands_of_stale_entries(K):
    """900: the 256-thread instance holds every level; 3 000: it gives up and the 512-thread one takes over; 4 600: that one gives up
    as well and the multi-launch path runs the level."""
    text = O.escape(stale_levels_text(K, K))
    want, _ = O.lcpcomp_huff_compress(text, 3, 1)
    seen = {}
    for mode in ("1", "0", "2"):
        ctx = _ctx_env({"TDC_GPU_SMALL_BIG": mode})
        try:
            got, st = ctx.lcpcomp_compress(text, threshold=3, flatten=1)
        finally:
            ctx.close()
        seen[mode] = (st["small_levels"], st["levels"])
        assert got == want, "K=%d TDC_GPU_SMALL_BIG=%s: stream differs (%d vs %d bytes)" % (K, mode, len(got), len(want))
    if K == 3000:       # the 512-thread instance keeps the fifty crowded levels off the multi-launch path
        assert seen["1"][0] >= seen["0"][0] + 40, seen


def test_every_small_level_on_the_large_instance():
    ctx = _ctx_env({"TDC_GPU_SMALL_BIG": "2"})
    try:
        for name, data in (("english_600k", T.gen_english(600_000, 5).tobytes()), ("dna_500k", T.gen_dna(500_000, 9).tobytes()),
                           ("runs", b"ab" * 5000 + b"c" + b"abc" * 7000 + bytes(range(1, 200)) * 40)):
            text = O.escape(data)
            for thr in (2, 5):
                want, _ = O.lcpcomp_huff_compress(text, thr, 1)
                got, _ = ctx.lcpcomp_compress(text, threshold=thr, flatten=1)
                assert got == want, (name, thr)
    finally:
        ctx.close()
