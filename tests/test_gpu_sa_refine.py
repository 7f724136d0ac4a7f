"""GPU parity of the suffix-array refinement pass (pytest -m gpu): suffix_array.hip sa_refine_kernel orders the small groups of the
initial order by their next k symbols straight from the text.  It only runs for texts of at least 2^16 bytes, so the small corpus
never reaches it: these texts do -- many small groups, groups on tile borders, groups of every length around the limit of 256, long
periodic stretches, tiny and full byte alphabets -- and SA / ISA / Phi / PLCP must equal the oracle's (divsufsort semantics:
ds/SADivSufSort.hpp:27-51, ds/ISAFromSA.hpp:30-43, ds/PhiFromSA.hpp:35-45, ds/PLCPFromPhi.hpp:27-53) with the pass on and off."""
import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _texts():
    rng = np.random.default_rng(2024)
    out = []
    out.append(("english_200k", T.gen_english(200_000, 5).tobytes()))
    out.append(("dna_150k", T.gen_dna(150_000, 3).tobytes()))
    # words from a tiny vocabulary: thousands of groups of every size, most of them tying on the second key as well
    voc = [bytes(rng.integers(97, 101, int(rng.integers(1, 6)), dtype=np.uint8)) for _ in range(40)]
    out.append(("tiny_vocabulary", b" ".join(voc[int(i)] for i in rng.integers(0, 40, 40_000))))
    # blocks repeated r times for r = 1 .. 600: groups of exactly r members (around the limit of 256 and beyond a tile)
    blocks = []
    for r in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 600]:
        blk = bytes(rng.integers(65, 91, 24, dtype=np.uint8))
        sep = [bytes(rng.integers(97, 123, 3, dtype=np.uint8)) for _ in range(r)]
        blocks.append(b"".join(blk + s for s in sep))
    rep = b"".join(blocks)
    out.append(("group_sizes", rep + bytes(rng.integers(48, 58, 70_000 - min(len(rep), 60_000), dtype=np.uint8))))
    out.append(("periodic", (b"abcab" * 30_000)[:100_000] + bytes(rng.integers(97, 99, 30_000, dtype=np.uint8))))
    out.append(("all_bytes", bytes(rng.integers(1, 255, 90_000, dtype=np.uint8))))
    out.append(("two_letters", bytes(rng.integers(97, 99, 120_000, dtype=np.uint8))))
    out.append(("just_above_limit", bytes(rng.integers(97, 100, 65_535, dtype=np.uint8))))        # 65 536 bytes with the sentinel
    return out


TEXTS = _texts()


def test_round_sort_choice_on_large_groups(gpu_ctx, ctx_plain, ctx_global_rounds):
    """4 MiB of a tiny vocabulary: millions of unresolved suffixes in large groups after the initial sort -- the rounds take the
    global splitter sort (by the library's own choice in the first two contexts, forced in the third); all three must give the
    oracle's suffix array."""
    rng = np.random.default_rng(77)
    voc = [bytes(rng.integers(97, 100, int(rng.integers(2, 7)), dtype=np.uint8)) for _ in range(12)]
    data = b" ".join(voc[int(i)] for i in rng.integers(0, 12, 900_000))[:4_000_000]
    text = O.escape(data)
    sa = O.suffix_array(text)
    for ctx in (gpu_ctx, ctx_plain, ctx_global_rounds):
        g_sa, g_isa = ctx.suffix_array(text)
        assert np.array_equal(g_sa, sa)
        assert np.array_equal(g_isa[sa], np.arange(len(sa), dtype=np.uint32))


def _ctx_with(var, value):
    return T.Context(0, options={var: value})


@pytest.fixture(scope="module")
def ctx_plain():
    c = _ctx_with("TDC_GPU_SA_REFINE", "0")
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_global_rounds():
    """Doubling rounds sorted by one splitter sort of the whole active list (the choice the library makes by itself where the
    unresolved groups are large; forced here so that small texts take it too -- it needs 2^20 active suffixes to apply)."""
    c = _ctx_with("TDC_GPU_SA_LOCAL", "2")
    yield c
    c.close()


@pytest.mark.parametrize("name,data", TEXTS, ids=[t[0] for t in TEXTS])
def test_text_index_with_and_without_refinement(gpu_ctx, ctx_plain, ctx_global_rounds, name, data):
    text = O.escape(data)
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    for label, ctx in (("refined", gpu_ctx), ("plain", ctx_plain), ("global rounds", ctx_global_rounds)):
        g = ctx.textds(text)
        assert np.array_equal(g["sa"], sa), "%s %s: SA" % (name, label)
        assert np.array_equal(g["isa"], isa), "%s %s: ISA" % (name, label)
        assert np.array_equal(g["phi"], phi), "%s %s: Phi" % (name, label)
        assert np.array_equal(g["plcp"], plcp), "%s %s: PLCP" % (name, label)
        assert g["maxlcp"] == maxlcp


def _copied_text(n, sigma, blk, seed, copies_of_copies=True):
    """random text in which every fourth block is a copy of an earlier window -- sources overlap, copies are copied again: groups of
    two, three and more suffixes with common extensions of up to `blk` symbols (the structure of the SURVEY 8d DNA generator)"""
    rng = np.random.default_rng(seed)
    out = np.empty(n + blk, dtype=np.uint8)
    pos = 0
    while pos < n:
        if pos >= blk and rng.integers(0, 4) == 0:
            src = int(rng.integers(0, pos - blk + 1)) if copies_of_copies else int(rng.integers(0, max(1, pos // 4)))
            out[pos:pos + blk] = out[src:src + blk]
        else:
            out[pos:pos + blk] = rng.integers(0, sigma, blk, dtype=np.uint8) + 65
        pos += blk
    return out[:n].tobytes()


def test_star_step_orders_groups_of_any_size(gpu_ctx):
    """Round 6: in the doubling fall-back of the wide path every group of unresolved suffixes is ordered against its smallest member
    from the text (lce along chains of consecutive positions) before the doubling rounds.  Texts of 2^25 bytes and more with copied
    blocks take it (sa_star_chains > 0): suffix array and stream equal the oracle's / the pair step's; a periodic text (one head per
    member) must NOT take it."""
    plain = T.Context(0, options={"sa_stars": 0})
    try:
        n = (1 << 25) + 4097
        for name, data in (("dna_copies", _copied_text(n, 4, 4096, 1)), ("sigma2_short_copies", _copied_text(n, 2, 700, 2)),
                           ("sigma20_long_copies", _copied_text(n, 20, 20000, 3, False)), ("dna_gen", T.gen_dna(n, 11).tobytes()),
                           # extensions beyond the length field of the round's key (16 bits at this size): the clamped class must not
                           # share its value with members whose extension really is that long (tools/star_stress.py found it)
                           ("copies_of_70000", _copied_text(n, 4, 70000, 7))):
            text = O.escape(data)
            a, st = gpu_ctx.lcpcomp_compress(text, threshold=3, flatten=1)
            b, sb = plain.lcpcomp_compress(text, threshold=3, flatten=1)
            assert st["sa_mode"] == 0 and st["sa_star_chains"] > 0, (name, st)
            assert sb["sa_star_chains"] == 0
            assert a == b, name
            sa, isa = gpu_ctx.suffix_array(text)
            sa2, _ = plain.suffix_array(text)
            assert np.array_equal(sa, sa2), name
            assert np.array_equal(isa[sa], np.arange(len(sa), dtype=np.uint32))
        # the oracle itself on the smallest interesting size
        text = O.escape(_copied_text(n, 4, 4096, 5))
        sa, _ = gpu_ctx.suffix_array(text)
        assert np.array_equal(sa, O.suffix_array(text))
        # periodic: every suffix of a run has the same representative -- one chain per member, the guard leaves it to the other paths
        text = O.escape((b"abcab" * 8_000_000)[:n])
        got, st = gpu_ctx.lcpcomp_compress(text, threshold=3, flatten=1)
        assert st["sa_star_chains"] == 0
        assert got == plain.lcpcomp_compress(text, threshold=3, flatten=1)[0]
    finally:
        plain.close()


def _phrases_text(seed):
    """random letters with phrases of 40 letters that occur r times each, every occurrence followed by other letters: after the wide
    sort's 25 symbols the suffixes inside a phrase form groups of exactly r members -- r from 2 to 3 000, around every limit of
    wsort_sorted_runs (16 members by themselves in an LDS tile, 32 / 64 / 256 / 1 024 by the run kernels, more through the record sort)"""
    rng = np.random.default_rng(seed)
    letters = lambda k: rng.integers(97, 123, k, dtype=np.uint8).tobytes()
    parts = []
    for r in (2, 3, 5, 9, 15, 16, 17, 18, 31, 32, 33, 64, 65, 200, 256, 257, 1000, 1024, 1025, 1500, 3000, 2, 7, 1100):
        ph = letters(40)
        for _ in range(r):
            parts.append(ph + letters(int(rng.integers(8, 30))))
    order = rng.permutation(len(parts))
    body = b"".join(parts[int(i)] for i in order)
    return letters(1_300_000) + body + letters(1_000_000)


def test_text_rounds_order_their_groups_in_place():
    """Round 6: the text rounds of the wide path order every group of still-equal suffixes in place (wsort_sorted_runs) instead of sorting
    their list as a whole.  Groups of every size around its limits; with the list of long groups too small (sa_seg_bigcap = 1) the round
    must fall back to the record sort on a list whose short groups have been ordered already.  Suffix array = the oracle's, streams equal."""
    text = O.escape(_phrases_text(77))
    want_sa = O.suffix_array(text)
    streams = []
    for opts in ({}, {"sa_seg_rounds": 0}, {"sa_seg_rounds": 1}, {"sa_seg_rounds": 2}, {"sa_seg_bigcap": 1}, {"sa_seg_bigcap": 0}, {"sa_seg_rounds": 2, "sa_seg_bigcap": 1}):
        with T.Context(0, options=opts) as ctx:
            got, st = ctx.lcpcomp_compress(text, threshold=2, flatten=1)
            assert st["sa_mode"] == 1 and st["sa_text_rounds"] >= 1, (opts, st)
            streams.append(got)
            sa, isa = ctx.suffix_array(text)
            assert np.array_equal(sa, want_sa), opts
    assert all(s == streams[0] for s in streams)
    assert streams[0] == O.lcpcomp_huff_compress(text, 2, 1)[0]
