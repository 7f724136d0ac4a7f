"""GPU parity of the wide-key suffix sort and what hangs on it (pytest -m gpu): wsort.hip (bit-packed one- or two-word keys through
sampled splitters, the leaf kernel that orders runs of equal first words by the second word and emits head flags + LCPs), the rank-free
text rounds of suffix_array.hip and the fused ISA / Phi / PLCP scatter of fused.hip.  The path only takes texts of 2^20 bytes by
default; TDC_GPU_WSORT_MIN lowers that so that the adversarial texts of test_gpu_sa_refine.py -- groups of every size around the
counting limit of 256, periodic stretches, tiny and full byte alphabets -- reach it too.  SA / ISA / Phi / PLCP must equal the
oracle's (ds/SADivSufSort.hpp:27-51, ds/ISAFromSA.hpp:30-43, ds/PhiFromSA.hpp:35-45, ds/PLCPFromPhi.hpp:27-53) in every variant:
one / two key words, 1 / 2 / 3 partition levels, the chunk iterations forced on every run, text rounds on, cut short, and off,
runs ordered inside the sort kernel or by the counting kernel, counting limits 16 and 64."""
import os

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O
from tests.test_gpu_sa_refine import TEXTS

pytestmark = pytest.mark.gpu


def _ctx_env(env):
    """a context with these options (tdc_gpu_ctx_set_option: the library does not read the environment)"""
    return T.Context(0, options=env)


VARIANTS = {
    "auto":        {"TDC_GPU_WSORT_MIN": "4096"},
    "kw1":         {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_WSORT_KW": "1"},
    "kw2_L2":      {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_WSORT_KW": "2", "TDC_GPU_SSORT_LEVELS": "2"},
    "L3_chunks":   {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_SSORT_LEVELS": "3", "TDC_GPU_WSORT_SMALLRUN": "1"},
    "no_rounds":   {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_WSORT_ROUNDS": "0"},
    "one_round":   {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_WSORT_ROUNDS": "1", "TDC_GPU_WSORT_KW": "1"},
    "two_wide":    {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_WSORT_TWO": "2"},     # two levels of 1024 buckets (texts of 4 MB and more)
    "no_fuse":     {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_WSORT_FUSE": "0"},    # runs of a unit ordered by the separate counting kernel
    "cmax_64":     {"TDC_GPU_WSORT_MIN": "4096", "TDC_GPU_WSORT_CMAX": "64"},   # counting up to 64 members (default 16)
}


@pytest.fixture(scope="module")
def ctxs():
    cs = {k: _ctx_env(v) for k, v in VARIANTS.items()}
    yield cs
    for c in cs.values():
        c.close()


def _more_texts():
    rng = np.random.default_rng(99)
    out = list(TEXTS)
    out.append(("english_5M", T.gen_english(5_000_000, 11).tobytes()))
    out.append(("dna_2M", T.gen_dna(2_000_000, 5).tobytes()))
    # long repeats: the text rounds cannot finish (LCP of thousands) -- the doubling fall-back must take over from the depth reached
    blk = bytes(rng.integers(97, 123, 5000, dtype=np.uint8))
    out.append(("long_repeats", blk + b"x" + blk[:4000] + b"y" + blk + bytes(rng.integers(97, 123, 60_000, dtype=np.uint8))))
    # heavy keys: a handful of phrases make up most of the text (equality leaves, pure units, long runs inside mixed units)
    phr = [bytes(rng.integers(97, 105, int(rng.integers(20, 40)), dtype=np.uint8)) for _ in range(6)]
    out.append(("heavy_phrases", b"".join(phr[int(i)] + bytes(rng.integers(97, 105, int(rng.integers(0, 4)), dtype=np.uint8))
                                         for i in rng.integers(0, 6, 12_000))))
    return out


ALL_TEXTS = _more_texts()


def _check(ctx, label, text, ref):
    sa, isa, phi, plcp, maxlcp = ref
    g = ctx.textds(text)
    for k, want in (("sa", sa), ("isa", isa), ("phi", phi), ("plcp", plcp)):
        got = g[k]
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s: %s differs at %d of %d slots, first %d: got %s want %s" % (
            label, k, bad.size, len(want), bad[0], got[bad[0]:bad[0] + 6], want[bad[0]:bad[0] + 6])
    assert g["maxlcp"] == maxlcp, label
    s2, i2 = ctx.suffix_array(text)                            # SA + ISA only: the wide sort followed by the classic rank path
    assert np.array_equal(s2, sa) and np.array_equal(i2, isa), label + ": suffix_array()"


@pytest.mark.parametrize("name,data", ALL_TEXTS, ids=[t[0] for t in ALL_TEXTS])
def test_text_index_through_the_wide_path(ctxs, name, data):
    text = O.escape(data)
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    for label, ctx in ctxs.items():
        _check(ctx, "%s [%s]" % (name, label), text, (sa, isa, phi, plcp, maxlcp))


def test_fast_mode_is_taken_and_streams_match(ctxs):
    data = T.gen_english(2_500_000, 3).tobytes()
    text = O.escape(data)
    want, _ = O.lcpcomp_huff_compress(text, 2, 1)
    got, st = ctxs["auto"].lcpcomp_compress(text, threshold=2, flatten=1)
    assert st["sa_key_words"] == 2 and st["sa_mode"] == 1 and st["sa_text_rounds"] >= 1, st
    assert got == want
    got2, st2 = ctxs["no_rounds"].lcpcomp_compress(text, threshold=2, flatten=1)
    assert st2["sa_mode"] == 0 and got2 == want


def test_default_context_takes_the_wide_path_for_large_texts(gpu_ctx):
    data = T.gen_english(1_200_000, 8).tobytes()
    text = O.escape(data)
    want, _ = O.lcpcomp_huff_compress(text, 2, 1)
    got, st = gpu_ctx.lcpcomp_compress(text, threshold=2, flatten=1)
    assert st["sa_key_words"] == 2, st
    assert got == want


def test_level_one_behind_the_upload(gpu_ctx):
    """Host-buffer calls of 2^26 bytes and more run the first partition level of the suffix sort chunk by chunk behind the upload
    (code map and splitters from chunk 0).  Same stream as with the overlap switched off; a text whose later chunks bring byte values
    chunk 0 does not have must fall back (and still give the same stream)."""
    plain = _ctx_env({"TDC_GPU_WSORT_OVERLAP": "0"})
    try:
        n = (1 << 26) + 12345
        for name, data in (("english", T.gen_english(n, 21)), ("dna", T.gen_dna(n, 4))):
            text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
            a, sa_ = gpu_ctx.lcpcomp_compress(text, threshold=3, flatten=1)
            b, sb_ = plain.lcpcomp_compress(text, threshold=3, flatten=1)
            assert sa_["sa_overlapped"] == 1 and sb_["sa_overlapped"] == 0, (name, sa_["sa_overlapped"], sb_["sa_overlapped"])
            assert a == b, name
        # new byte values in the second half: the provisional code map is wrong, the overlapped work is discarded
        data = T.gen_english(n, 22)
        data[n // 2:] = np.where(data[n // 2:] == ord("e"), ord("E"), data[n // 2:])
        text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
        a, sa_ = gpu_ctx.lcpcomp_compress(text, threshold=3, flatten=1)
        b, sb_ = plain.lcpcomp_compress(text, threshold=3, flatten=1)
        assert sa_["sa_overlapped"] == 0
        assert a == b
    finally:
        plain.close()
