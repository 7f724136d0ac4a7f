"""GPU parity of the one-workgroup levels of the factorizer with MANY survivors (pytest -m gpu): csrc/factorize.hip runs a level whose
list holds up to 2 048 alive entries in one 256-thread workgroup and up to 4 096 in a 512-thread instance of the same kernel; more go
through the multi-launch path (a 1 024-thread instance with a slimmer LDS layout holds 8 192).  The text below is built so that fifty
consecutive levels each hold K stale entries: K blocks U_i V_i W_i where V_i W_i and U_i + the start of V_i W_i both occur once more in
a dictionary part, each followed by a smaller byte -- PLCP is |V W| at the start of V_i and S - j at byte j of U_i; the factors of the
top level truncate the latter to |U_i| - j, which leaves them as stale entries of the lists S - j.  With |U_i| fixed all K pushes of a
level go to ONE target list (the duplicate-target orderings), with |U_i| spread over 180 values the counting sort applies.  Streams
must equal the oracle's (compressors/lcpcomp/compress/ArraysComp.hpp:36-117) for every instance choice."""
import os

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def stale_levels_text(K, seed, spread):
    """spread = False: every block truncates its U part to the same value (all K pushes of a level share ONE target: the duplicate-
    target orderings); True: U has 60 .. 239 bytes, the K pushes of a level go to 180 different targets (the counting sort)."""
    rng = np.random.default_rng(seed)
    letters = lambda k: rng.integers(ord("b"), ord("z") + 1, k, dtype=np.uint8).tobytes()
    S, B = (250, 400) if spread else (150, 300)                              # lengths of the two repeats of a block
    main, dic = [], []
    for _ in range(K):
        u = int(rng.integers(60, 240)) if spread else 50
        U, VW = letters(u), letters(B)
        main.append(letters(int(rng.integers(20, 60))) + U + VW + b"z" + letters(8))
        dic.append(VW + b"a" + letters(int(rng.integers(5, 30))))            # 'a' < every byte that follows the main copy
        dic.append(U + VW[:S - u] + b"a" + letters(int(rng.integers(5, 30))))
    for L in range(60, S + 20):                                               # a live entry in every level, so that the levels are not
        R = letters(L)                                                        # collected in one go (no-live runs are batched)
        main.append(letters(int(rng.integers(10, 30))) + R + b"z" + letters(6))
        dic.append(R + b"a" + letters(int(rng.integers(5, 30))))
    order = rng.permutation(len(dic))
    return b"".join(main) + b"".join(dic[int(i)] for i in order)


def _ctx_env(env):
    """a context with these options (tdc_gpu_ctx_set_option: the library does not read the environment)"""
    return T.Context(0, options=env)


@pytest.mark.parametrize("K,spread", [(900, False), (3000, False), (3000, True), (4600, True), (9000, True)])
def test_levels_with_thousands_of_stale_entries(K, spread):
    """900: the 256-thread instance holds every level; 3 000: it gives up and the 512-thread one takes over; 4 600: that one gives up
    as well and the 1 024-thread instance (slim LDS layout) runs the level; 9 000: the multi-launch path."""
    text = O.escape(stale_levels_text(K, K, spread))
    want, _ = O.lcpcomp_huff_compress(text, 3, 1)
    seen = {}
    for mode in ("1", "0", "2", "3"):
        ctx = _ctx_env({"TDC_GPU_SMALL_BIG": mode, "TDC_GPU_EAGER": "0"})       # (the instances of the lazy one-workgroup kernel are what is tested here)
        try:
            got, st = ctx.lcpcomp_compress(text, threshold=3, flatten=1)
        finally:
            ctx.close()
        seen[mode] = (st["small_levels"], st["levels"])
        assert got == want, "K=%d TDC_GPU_SMALL_BIG=%s: stream differs (%d vs %d bytes)" % (K, mode, len(got), len(want))
    if K in (3000, 4600):       # the larger instances keep the fifty crowded levels off the multi-launch path
        assert seen["1"][0] >= seen["0"][0] + 40, (K, spread, seen)


@pytest.mark.parametrize("mode", ["2", "3"])
def test_every_small_level_on_the_large_instance(mode):
    ctx = _ctx_env({"TDC_GPU_SMALL_BIG": mode, "TDC_GPU_EAGER": "0"})
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


def test_one_long_run_keeps_its_level_tables_outside_the_arena(gpu_ctx):
    """A text that is one run has as many LCP levels as positions: the two tables with a word (two) per level do not fit the arena that
    tdc_gpu_arena_bytes() budgets per position (64 MB of one letter: out of memory in round 6's robustness run) -- beyond 2^22 levels
    they live in an allocation of their own (factorize.hip LevelTables).  Streams against the oracle's, and back."""
    for name, data in (("a^N", b"a" * 4_600_000), ("(ab)^N/2", b"ab" * 2_300_000), ("x a^N", b"x" + b"a" * 4_600_000)):
        text = O.escape(data)
        want, _ = O.lcpcomp_huff_compress(text, 5, 1)
        got, st = gpu_ctx.lcpcomp_compress(text, threshold=5, flatten=1)
        assert st["maxlcp"] + 2 > (1 << 22), name
        assert got == want, name
        back, _ = gpu_ctx.lcpcomp_decompress(got)
        assert back == text, name
