"""Full-size GPU tests of the BASELINE.json configurations (pytest -m gpu).

  * the metric's configuration -- 2*10^9 B English-like text through the end-to-end entry point with pinned host buffers:
    size-independent properties (the oracle's decoder reproduces the input from the GPU stream; header fields = reported
    statistics) plus a bit-exact comparison of a 32 MiB prefix text against the oracle's compressor;
  * configs[1], configs[2] (lcpcomp + ArithmeticCoder, 10^9 B DNA) and configs[3] (lz78 + Elias-gamma, 10^9 B) at full size against the
    committed hashes of the ORACLE's streams (tests/golden/oracle_fullsize.json), plus the same coders byte for byte against the oracle's
    compressors at sizes it takes seconds for.
"""
import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O
from tests.util import sha256, load_json

pytestmark = pytest.mark.gpu

# size + SHA-256 of the ORACLE's streams at full size (tests/make_fullsize_golden.py; minutes and tens of GB per entry, so they are
# committed instead of recomputed): the device stream is compared with them byte for byte, by hash
FULL = load_json("oracle_fullsize.json")


def _assert_golden(name, stream, st=None):
    g = FULL[name]
    assert len(stream) == g["size"], (name, len(stream), g["size"])
    assert sha256(stream) == g["sha256"], name
    if st is not None:
        for k in ("factors", "num_flattened", "maxlcp"):
            if k in g:
                assert st[k] == g[k], (name, k, st[k], g[k])


class _Bits:
    """MSB-first bit reader (io/BitIStream.hpp) for the stream header only."""

    def __init__(self, data):
        self.d, self.p = data, 0

    def bit(self):
        b = (int(self.d[self.p >> 3]) >> (7 - (self.p & 7))) & 1
        self.p += 1
        return b

    def int(self, bits):
        v = 0
        for _ in range(bits):
            v = (v << 1) | self.bit()
        return v

    def compressed_int(self, b=7):            # io/BitIStream.hpp read_compressed_int: b-bit groups, each preceded by a "more" bit
        v, shift = 0, 0
        while True:
            more = self.bit()
            v |= self.int(b) << shift
            shift += b
            if not more:
                return v


def _header(stream):
    """(n, flen_min, flen_max, fdist_max) of a lcpcomp(coder=huff) stream (LZSSCoding.hpp:27-41 behind the Huffman table)."""
    r = _Bits(stream)
    if r.bit():                               # huffmantable_encode (HuffmanCoder.hpp:264-273)
        longest = r.compressed_int()
        for _ in range(longest):
            r.compressed_int()
        sigma = r.compressed_int()
        for _ in range(sigma):
            r.int(8)
    n = r.int(32)
    w = O.bits_for(n)
    return n, r.int(w), r.int(w), r.int(w)


def test_metric_config_2e9_end_to_end(gpu_ctx):
    N = 2_000_000_000
    n = N + 1
    h_text = T.PinnedBuffer(n)
    h_out = T.PinnedBuffer(N)
    try:
        T.gen_english(N, 42, out=h_text.a)
        h_text.a[N] = 0
        out_len, st = gpu_ctx.lcpcomp_compress_into(h_text, n, h_out, 2, 1)
        assert st["n"] == n and st["out_len"] == out_len and 0 < out_len < N
        assert st["ms_h2d"] > 0 and st["ms_d2h"] > 0 and st["ms_total"] >= st["ms_h2d"] + st["ms_d2h"]
        stream = h_out.a[:out_len]
        assert sha256(h_text.a[:N]) == FULL["english_2e9"]["text_sha256"]
        _assert_golden("english_2e9", stream, st)          # byte for byte (by hash) the oracle's stream of the same 2*10^9 B text
        hn, fmin, fmax, dmax = _header(stream)
        assert (hn, fmin, fmax, dmax) == (n, st["flen_min"], st["flen_max"], st["fdist_max"])
        assert fmin >= 2 and fmax <= st["maxlcp"]
        back = O.lcpcomp_huff_decompress(stream)
        assert len(back) == n
        assert sha256(back) == sha256(h_text.a)
        del back
        # ... and the device decoder (token stream parsed on the device in six segments of 2^30 bit positions) into a pinned buffer
        h_back = T.PinnedBuffer(n)
        try:
            nb, dst = gpu_ctx.lcpcomp_decompress_into(h_out.a[:out_len], h_back)
            assert nb == n and dst["device_parse"] == 1 and dst["factors"] == st["factors"]
            assert sha256(h_back.a) == sha256(h_text.a)
        finally:
            h_back.free()
        # a buffer that is too small is refused with the required size, nothing is written past it
        tiny = np.concatenate([h_text.a[:1 << 22], np.zeros(1, dtype=np.uint8)])
        small = np.full(4096, 0xA5, dtype=np.uint8)
        with pytest.raises(T.TdcGpuError) as e:
            gpu_ctx.lcpcomp_compress_into(tiny, len(tiny), small[:1024], 2, 1)
        assert bool((small[1024:] == 0xA5).all())
        assert e.value.status == -5
        # bit-exact against the oracle's compressor on a prefix text (the generator's shorter outputs are prefixes)
        m = 1 << 25
        sample = np.concatenate([h_text.a[:m], np.zeros(1, dtype=np.uint8)])
        want, _ = O.lcpcomp_huff_compress(sample, 2, 1)
        got_len, _ = gpu_ctx.lcpcomp_compress_into(sample, m + 1, h_out, 2, 1)
        assert got_len == len(want) and h_out.a[:got_len].tobytes() == want
    finally:
        h_text.free()
        h_out.free()


def test_configs1_english_256MiB_golden(gpu_ctx):
    """BASELINE configs[1] through the end-to-end entry point against the committed oracle hash"""
    N = 1 << 28
    text = np.concatenate([T.gen_english(N, 42), np.zeros(1, dtype=np.uint8)])
    assert sha256(text[:N]) == FULL["english_256MiB"]["text_sha256"]
    got, st = gpu_ctx.lcpcomp_compress(text, 2, 1)
    _assert_golden("english_256MiB", got, st)


def test_configs2_dna_1e9_arith_golden(gpu_ctx):
    """BASELINE configs[2] at its full size (10^9 B DNA, LCPCompressor + ArithmeticCoder, threshold 5) against the committed oracle hash"""
    N = 10**9
    text = np.concatenate([T.gen_dna(N, 7), np.zeros(1, dtype=np.uint8)])
    assert sha256(text[:N]) == FULL["dna_1e9_arith"]["text_sha256"]
    got, st = gpu_ctx.lcpcomp_compress(text, 5, 1, T.CODER_ARITH)
    _assert_golden("dna_1e9_arith", got, st)


def test_config2_arithmetic_dna_64MiB_vs_oracle(gpu_ctx):
    """the same coder / generator byte for byte against the oracle's compressor at a size the oracle takes seconds for (the full
    size is covered by the committed hash above)"""
    N = 1 << 26
    text = np.concatenate([T.gen_dna(N, 7), np.zeros(1, dtype=np.uint8)])
    got, st = gpu_ctx.lcpcomp_compress(text, 5, 1, T.CODER_ARITH)
    want, _ = O.lcpcomp_arith_compress(text, 5, 1)
    assert len(got) == len(want) and sha256(got) == sha256(want)
    assert st["maxlcp"] >= 4096


def test_config3_lz78_gamma_32MiB_vs_oracle(gpu_ctx):
    N = 1 << 25
    data = T.gen_english(N, 42)
    got, st = gpu_ctx.lz78_compress(data)
    want = O.lz78_gamma_compress(data)
    assert len(got) == len(want) and sha256(got) == sha256(want)
    assert st["factors"] > 0


def test_configs3_lz78_1e9_golden(gpu_ctx):
    """BASELINE configs[3] at its full size (10^9 B English, LZ78Compressor + EliasGammaCoder) against the committed oracle hash (the parse
    runs on the host -- SURVEY 8 a17 --, the gamma pack on the device: about a minute)"""
    N = 10**9
    data = T.gen_english(N, 42)
    assert sha256(data) == FULL["lz78_1e9"]["text_sha256"]
    got, st = gpu_ctx.lz78_compress(data)
    g = FULL["lz78_1e9"]
    assert len(got) == g["size"], (len(got), g["size"])
    assert sha256(got) == g["sha256"]
