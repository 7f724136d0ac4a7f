"""GPU parity of the hybrid factorizer (pytest -m gpu): levels above TDC_GPU_WINDOW_LCUT through the global level loop,
the levels below window by window in one launch (factorize_tiles.hip) -- including the fallback when a window cannot be
completed.  Texts are just long enough for the window path (>= 64 Ki positions) so the oracle stays fast."""
import os
import random

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O
from tests import corpus

pytestmark = pytest.mark.gpu


def _texts():
    rng = random.Random(77)
    out = [
        ("english_96k", T.gen_english(96 * 1024, 42).tobytes()),
        ("english_300k", T.gen_english(300_000, 9).tobytes()),
        ("dna_128k", T.gen_dna(128 * 1024, 7).tobytes()),
        ("rand_s2_80k", bytes(rng.randrange(1, 3) for _ in range(80_000))),
        ("rand_s4_100k", bytes(rng.randrange(1, 5) for _ in range(100_000))),
        ("rand_s26_70k", bytes(rng.randrange(97, 123) for _ in range(70_000))),
        ("planted4_90k", corpus.planted(90_000, 4, rng, replen=300)),
        ("planted26_120k", corpus.planted(120_000, 26, rng, replen=40)),
        ("runrich_70k", corpus.run_rich(70_000, rng)),
        ("fib24", corpus.fib_word(24)[:100_000]),
        ("thue17", corpus.thue_morse(17)),
        ("periodic_7", (b"abcdefg" * 12_000)),
        ("words_66k", b" ".join(rng.choice([b"alpha", b"beta", b"gamma", b"delta", b"pi", b"rho"]) for _ in range(20_000))),
    ]
    return out


TEXTS = _texts()


@pytest.fixture(scope="module")
def ctx_for():
    """Contexts per window_lcut value (option of the context)."""
    made = {}

    def get(lcut):
        if lcut not in made:
            made[lcut] = T.Context(0, options={"window_lcut": lcut})
        return made[lcut]
    yield get
    for c in made.values():
        c.close()


@pytest.mark.parametrize("name,data", TEXTS, ids=[t[0] for t in TEXTS])
def test_window_levels_match_oracle(ctx_for, name, data):
    text = O.escape(data)
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    lcp = O.lcp_array(sa, plcp)
    seen = set()
    for thr in (1, 2, 3, 5):
        ref = O.sort_factors(O.arrays_comp(sa, isa, lcp, maxlcp, thr))
        fl, nf, md = O.flatten(ref)
        for lcut in (48, 3, 12, 63, 0):
            ctx = ctx_for(lcut)
            pos, src, ln, st = ctx.factorize(text, thr, flatten=1)
            seen.add(st["window_pass"])
            assert st["factors"] == len(ref), "%s t=%d lcut=%d: %d factors, want %d (window_pass %d)" % (
                name, thr, lcut, st["factors"], len(ref), st["window_pass"])
            assert np.array_equal(pos, ref["pos"]) and np.array_equal(ln, ref["len"]), "%s t=%d lcut=%d" % (name, thr, lcut)
            assert np.array_equal(src, fl["src"]), "%s t=%d lcut=%d: flattened sources" % (name, thr, lcut)
            assert (st["num_flattened"], st["max_depth_lb"]) == (nf, md)
            if lcut == 0:
                assert st["window_pass"] == 0
    assert seen & {1, 2}, "the window pass never ran"


def test_smallest_halo_and_retry_with_the_largest():
    """The borders of a window's known range only move where an unknown factor can exist, so the first attempt runs with a small
    halo (TDC_GPU_WINDOW_HALO; clamped to 2 * lcut + 64) and a failed border retries with the largest one.  Both outcomes must
    give the oracle's factors: texts whose repeats are longer than the halo force the retry."""
    rng = np.random.default_rng(99)
    unit = bytes(rng.integers(97, 101, 150, dtype=np.uint8))
    texts = [("english", T.gen_english(1 << 19, 3).tobytes()),
             ("periodic", (unit * 2000)[:200000] + bytes(rng.integers(97, 123, 70000, dtype=np.uint8))),
             ("mutated", b"".join(bytes([c if rng.random() > 0.002 else 120 for c in unit]) for _ in range(1500)))]
    with T.Context(0, options={"window_halo": 0, "window_lcut": 48}) as ctx:
        for name, data in texts:
            text = O.escape(data)
            for thr in (2, 5):
                want, _ = O.lcpcomp_huff_compress(text, thr, 1)
                got, st = ctx.lcpcomp_compress(text, threshold=thr, flatten=1)
                assert got == want, "%s t=%d (window_pass %d)" % (name, thr, st["window_pass"])


def test_window_pass_is_used_and_stream_bit_exact(ctx_for):
    """End to end on the bench corpus generator: the default configuration takes the window path (no fallback) and the
    stream equals the oracle's."""
    data = T.gen_english(1 << 21, 42).tobytes()
    text = O.escape(data)
    want, _ = O.lcpcomp_huff_compress(text, 2, 1)
    got, st = ctx_for(48).lcpcomp_compress(text, threshold=2, flatten=1)
    assert st["window_pass"] == 1
    assert got == want
    got0, st0 = ctx_for(0).lcpcomp_compress(text, threshold=2, flatten=1)
    assert st0["window_pass"] == 0 and got0 == want


def test_long_repeats_skip_erased_levels(ctx_for):
    """Texts with long repeats: tens of thousands of consecutive levels hold one erased candidate each (the PLCP ramp inside
    a repeat); the level loop probes how far such a run goes instead of paying a launch per level.  Bit-exact, and the
    number of processed levels stays small."""
    rng = random.Random(4)
    blk = bytes(rng.randrange(97, 123) for _ in range(120_000))
    cases = {
        "twice": blk + b"#" + blk + T.gen_english(100_000, 5).tobytes(),
        "nested": blk + b"#" + blk[1000:90_000] + b"$" + blk[500:60_000] + b"%" + blk[30_000:110_000] + T.gen_dna(80_000, 3).tobytes(),
        "overlap": (blk[:50_000] * 3) + b"&" + blk[10_000:70_000],
    }
    for name, data in cases.items():
        text = O.escape(data)
        for thr in (2, 5):
            want, wst = O.lcpcomp_huff_compress(text, thr, 1)
            got, st = ctx_for(48).lcpcomp_compress(text, threshold=thr, flatten=1)
            assert got == want, "%s t=%d" % (name, thr)
            assert st["factors"] == wst["factors"] and st["maxlcp"] == wst["maxlcp"] and st["maxlcp"] >= 40_000
            assert st["levels"] < 5_000, "%s t=%d: %d levels were processed one by one" % (name, thr, st["levels"])


def test_structured_random_texts(ctx_for):
    """Randomised structure (planted repeats of many lengths, runs, periodic stretches, alphabet sizes 2..200), just above
    the window path's minimum length: full stream against the oracle for random thresholds / flatten settings."""
    rng = random.Random(2026)
    for case in range(40):
        sigma = rng.choice([2, 3, 4, 8, 26, 200])
        target = rng.randrange(66_000, 140_000)
        parts, total = [], 0
        pool = [bytes(rng.randrange(1, 1 + sigma) for _ in range(rng.choice([5, 40, 300, 3000, 20_000]))) for _ in range(6)]
        while total < target:
            r = rng.random()
            if r < 0.35:
                src = rng.choice(pool)
                a = rng.randrange(len(src)); b = rng.randrange(a, len(src)) + 1
                piece = src[a:b]
            elif r < 0.45:
                piece = bytes([rng.randrange(1, 1 + sigma)]) * rng.randrange(1, 400)
            elif r < 0.55:
                unit = bytes(rng.randrange(1, 1 + sigma) for _ in range(rng.randrange(2, 9)))
                piece = unit * rng.randrange(2, 200)
            else:
                piece = bytes(rng.randrange(1, 1 + sigma) for _ in range(rng.randrange(1, 2000)))
            parts.append(piece); total += len(piece)
        text = O.escape(b"".join(parts))
        thr = rng.choice([1, 2, 3, 5, 8, 20])
        fl = rng.choice([0, 1])
        want, wst = O.lcpcomp_huff_compress(text, thr, fl)
        got, st = ctx_for(48).lcpcomp_compress(text, threshold=thr, flatten=fl)
        assert got == want, "case %d (sigma %d, n %d, t %d, flatten %d, maxlcp %d, window_pass %d)" % (
            case, sigma, len(text), thr, fl, wst["maxlcp"], st["window_pass"])


@pytest.mark.gpu
def test_extreme_repeat_structure(gpu_ctx):
    """Runs, periodic texts and a Fibonacci word (a ramp of PLCP values over hundreds of thousands of levels, almost all of them
    holding only erased or stale entries): both strategies equal the oracle, and the level loop skips / batches instead of
    visiting every level."""
    import time
    N = 300_000
    a, b = b"a", b"ab"
    while len(b) < N:
        a, b = b, b + a
    cases = {"a^N": b"a" * N, "(ab)^N/2": b"ab" * (N // 2), "(abc)^k x (abc)^k": b"abc" * (N // 6) + b"x" + b"abc" * (N // 6),
             "fibonacci": b[:N], "x a^N": b"x" + b"a" * N, "a^N x a^N y": b"a" * N + b"x" + b"a" * N + b"y"}
    for name, data in cases.items():
        text = O.escape(data)
        t0 = time.time()
        got, _ = gpu_ctx.lzss_lcp_compress(text, 3)                     # the LCE passes share the PLCP kernels
        assert got == O.lzss_lcp_huff_compress(text, 3)[0] and time.time() - t0 < 1.5, name
        for comp, fn in ((T.COMP_ARRAYS, O.lcpcomp_huff_compress), (T.COMP_MAXLCP, O.lcpcomp_maxlcp_huff_compress)):
            for thr in (2, 5):
                t0 = time.time()
                got, st = gpu_ctx.lcpcomp_compress(text, thr, 1, T.CODER_HUFF, comp)
                dt = time.time() - t0
                assert got == fn(text, thr, 1)[0], (name, comp, thr)
                assert dt < 1.5, "%s comp=%d: %.2f s (a level-by-level walk over the ramp)" % (name, comp, dt)


def test_large_lists_give_the_same_factors():
    """TDC_GPU_WINDOW_LARGE=1: the window pass starts with the large per-level lists (one workgroup per CU).  Texts with one crowded
    level overflow the small lists and end up there by themselves; here every text is forced through them."""
    with T.Context(0, options={"window_large": 1}) as ctx:
        for name, data in TEXTS[:7] + [("dna_1M", T.gen_dna(1 << 20, 3).tobytes())]:
            text = O.escape(data)
            for thr in (2, 5):
                want, _ = O.lcpcomp_huff_compress(text, thr, 1)
                got, st = ctx.lcpcomp_compress(text, threshold=thr, flatten=1)
                assert st["window_pass"] in (1, 2), (name, thr)
                assert got == want, "%s t=%d (window_pass %d)" % (name, thr, st["window_pass"])


def test_discarded_window_pass_falls_back_to_the_level_loop():
    """TDC_GPU_WINDOW_FORCE_FAIL=1: every window pass is discarded as if a border had failed, the global level loop evaluates the low
    levels from the residence bytes.  On a text that takes the fused ISA / PLCP scatter without a Phi array (1 MiB and more, no deep
    repeats) the sources of those levels' factors must then come from SA[ISA[p] - 1] saved by cand_rebuild_class_kernel."""
    with T.Context(0, options={"window_force_fail": 1}) as ctx:
        for name, data, thr in (("english_3M", T.gen_english(3_000_000, 12).tobytes(), 2), ("english_1.5M_t5", T.gen_english(1_500_000, 13).tobytes(), 5)):
            text = O.escape(data)
            for fl in (1, 0):
                want, _ = O.lcpcomp_huff_compress(text, thr, fl)
                got, st = ctx.lcpcomp_compress(text, threshold=thr, flatten=fl)
                assert st["window_pass"] == 2 and st["sa_mode"] == 1, (name, st["window_pass"], st["sa_mode"])
                assert got == want, (name, fl)
