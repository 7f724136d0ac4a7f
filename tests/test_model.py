"""CPU tests: the position-space / round-based formulation used by the HIP kernels (tests/models/position_space.py)
reproduces the reference's sequential ArraysComp and flatten exactly (factor lists incl. emission order)."""
import pytest

from oracle import oracle as O
from tests import corpus
from tests.models.position_space import factorize_position_space, flatten_rounds


def _check(data, thresholds):
    text = O.escape(data)
    n = len(text)
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    lcp = O.lcp_array(sa, plcp)
    for thr in thresholds:
        ref = O.arrays_comp(sa, isa, lcp, maxlcp, thr)
        ref_l = [(int(a), int(b), int(c)) for a, b, c in ref]
        got, _, _ = factorize_position_space(n, isa, phi, plcp, maxlcp, thr)
        assert got == ref_l
        srt = O.sort_factors(ref)
        fl, nf, md = O.flatten(srt)
        mine, nf2, md2, _ = flatten_rounds([(int(a), int(b), int(c)) for a, b, c in srt])
        assert mine == [(int(a), int(b), int(c)) for a, b, c in fl]
        assert (nf, md) == (nf2, md2)


@pytest.mark.parametrize("name,data", [c for c in corpus.small_corpus() if len(c[1]) <= 3000], ids=lambda x: x if isinstance(x, str) else "")
def test_model_on_corpus(name, data):
    _check(data, (1, 2, 5))


def test_model_on_random_inputs():
    for _, data in corpus.random_small(400, seed=7):
        _check(data, (1, 2, 3, 5))


def test_explicit_priority_model_is_order_independent():
    """The second formulation (unordered lists + explicit priorities, as implemented in factorize.hip) reproduces the
    reference's factor list incl. emission order even when every list is shuffled."""
    import random
    from tests.models.position_space import factorize_explicit_priority
    rng = random.Random(5)
    cases = corpus.random_small(300, seed=11) + [c for c in corpus.small_corpus() if len(c[1]) <= 1500]
    for _, data in cases:
        text = O.escape(data)
        n = len(text)
        sa = O.suffix_array(text)
        isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
        lcp = O.lcp_array(sa, plcp)
        for thr in (1, 2, 5):
            ref = [(int(a), int(b), int(c)) for a, b, c in O.arrays_comp(sa, isa, lcp, maxlcp, thr)]
            assert factorize_explicit_priority(n, isa, phi, plcp, maxlcp, thr, rng) == ref


def _tile_check(data, thr, lcut, interior, halo):
    """Hybrid factorizer (global levels above lcut, window-local levels below; factorize_tiles.hip): every window that
    reports itself valid must reproduce the reference's factors inside its interior exactly."""
    from tests.models.position_space import factorize_hybrid_tiles
    text = O.escape(data)
    n = len(text)
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    lcp = O.lcp_array(sa, plcp)
    ref = {(int(a), int(c)) for a, b, c in O.arrays_comp(sa, isa, lcp, maxlcp, thr)}
    got, tiles, invalid = factorize_hybrid_tiles(n, isa, phi, plcp, maxlcp, thr, lcut, interior, halo)
    inv = [(x[1], x[2]) for x in got if x[0] == "invalid"]
    g = {x for x in got if x[0] != "invalid"}
    covered = lambda p: any(a <= p < b for a, b in inv)
    assert all(f in g for f in ref if not (f[1] <= lcut and covered(f[0])))
    assert all(f in ref for f in g)
    return tiles, invalid


def test_tile_local_levels_model():
    tiles = invalid = 0
    cases = corpus.random_small(120, seed=11) + [c for c in corpus.small_corpus() if len(c[1]) <= 2000]
    for _, data in cases:
        for thr in (1, 2, 5):
            for lcut, interior, halo in ((4, 32, 16), (6, 64, 40), (3, 16, 8), (100, 64, 32)):
                t, i = _tile_check(data, thr, lcut, interior, halo)
                tiles += t
                invalid += i
    assert 0 < invalid < tiles // 2          # both outcomes are exercised


def test_max_lcp_position_space_model():
    """Device formulation of MaxLCPStrategy (stack order as explicit priorities, eager decreases as one push per factor and
    level) == the oracle's linked-list restatement, including the emission order."""
    import numpy as np
    from tests.models.position_space import max_lcp_position_space
    rng = np.random.default_rng(1)
    cases = [("ex", b"abcdebcdeabcd abcdebcdeabcd banana bandana")] + [c for c in corpus.small_corpus() if len(c[1]) <= 1500]
    for i in range(20):
        sig = int(rng.integers(2, 5))
        cases.append(("r%d" % i, bytes(rng.integers(97, 97 + sig, int(rng.integers(20, 300)), dtype=np.uint8))))
    for name, data in cases:
        t = O.escape(data)
        sa = O.suffix_array(t)
        isa, phi, plcp, maxlcp = O.isa_phi_plcp(t, sa)
        lcp = O.lcp_array(sa, plcp)
        for thr in (1, 2, 5):
            want = [(int(f["pos"]), int(f["src"]), int(f["len"])) for f in O.max_lcp(sa, isa, lcp, maxlcp, thr)]
            assert max_lcp_position_space(len(t), isa, phi, plcp, maxlcp, thr) == want, (name, thr)


def test_eager_factor_set_model():
    """Round 5: ArraysComp as a function of the factor SET (the factors are sorted by position before anything reads them).  The order
    among pushed-down entries is irrelevant -- truncated entries of one level never meet, and they only lose against natural entries
    to their left -- so a cut entry may move to list cur[x] at once.  factorize_eager visits the truncated entries of every level in
    SHUFFLED order and must still produce the oracle's factor set."""
    import random
    from tests.models.position_space import factorize_eager
    rng = random.Random(3)
    cases = corpus.random_small(500, seed=21) + [c for c in corpus.small_corpus() if len(c[1]) <= 3000]
    total_visits = 0
    for name, data in cases:
        text = O.escape(data)
        n = len(text)
        sa = O.suffix_array(text)
        isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
        for thr in (1, 2, 3, 5):
            lcp = O.lcp_array(sa, plcp)
            ref = {(int(a), int(c)) for a, b, c in O.arrays_comp(sa, isa, lcp, maxlcp, thr)}
            got, visits = factorize_eager(n, isa, plcp, maxlcp, thr, rng)
            assert got == ref, (name, thr)
            total_visits += visits
    assert total_visits > 0


def test_eager_tile_model():
    """The window pass of round 5 (factorize_tiles.hip, window_eager_kernel): one state byte per position, eager push-down, priorities
    only between natural entries, known-range borders as before -- every window that reports itself valid reproduces the oracle's
    factors inside its interior."""
    from tests.models.position_space import factorize_hybrid_tiles_eager
    tiles = invalid = 0
    cases = corpus.random_small(100, seed=11) + [c for c in corpus.small_corpus() if len(c[1]) <= 2000]
    for name, data in cases:
        text = O.escape(data)
        n = len(text)
        sa = O.suffix_array(text)
        isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
        for thr in (1, 2, 5):
            lcp = O.lcp_array(sa, plcp)
            ref = {(int(a), int(c)) for a, b, c in O.arrays_comp(sa, isa, lcp, maxlcp, thr)}
            for lcut, interior, halo in ((4, 32, 16), (6, 64, 40), (3, 16, 8), (100, 64, 32)):
                got, t, i = factorize_hybrid_tiles_eager(n, isa, plcp, maxlcp, thr, lcut, interior, halo)
                inv = [(x[1], x[2]) for x in got if x[0] == "invalid"]
                g = {x for x in got if x[0] != "invalid"}
                covered = lambda p: any(a <= p < b for a, b in inv)
                assert all(f in g for f in ref if not (f[1] <= lcut and covered(f[0]))), (name, thr, lcut)
                assert all(f in ref for f in g), (name, thr, lcut)
                tiles += t
                invalid += i
    assert 0 < invalid < tiles // 2
