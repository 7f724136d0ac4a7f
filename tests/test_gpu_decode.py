"""GPU tests of the device-side parse of the lcpcomp(coder=huff) token stream (pytest -m gpu; SURVEY 8f #2, first half:
decode_text_internal, compressors/LCPCompressor.hpp:23-76, with HuffmanCoder::Decoder, coders/HuffmanCoder.hpp:377-397 / :572-612).

A context created with TDC_GPU_DEC_PARSE=2 parses EVERY stream on the device (the default takes streams of 1 MiB and more), so the
small corpus, random inputs, the reference-held decode vectors and damaged streams all go through the next() / chain-marking /
count / scan / emit kernels; the oracle's streams must decode to the oracle's texts, damaged streams must be refused, never crash."""
import os

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O
from tests import corpus
from tests.util import load_json, decode_sequence_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["lean", "general"])
def dev_ctx(request):
    """both markings of the device parse: the lean one (per-tile exits of the few possible entries; streams of short tokens -- the
    default for them) and the general one (next() / exits of every bit position; TDC_GPU_DEC_LEAN=0 gives it every stream)"""
    ctx = T.Context(0, options={"dec_parse": 2, "dec_lean": 1 if request.param == "lean" else 0})
    yield ctx
    ctx.close()


def _decodable(stream, text):
    try:
        return O.lcpcomp_huff_decompress(stream) == text
    except RuntimeError:
        return False


def test_small_corpus_streams_parse_on_the_device(dev_ctx):
    cases = corpus.small_corpus() + [("english_300k", T.gen_english(300_000, 9).tobytes()), ("dna_200k", T.gen_dna(200_000, 7).tobytes())]
    on_device = 0
    for name, data in cases:
        text = O.escape(data)
        for thr, fl in ((1, 1), (2, 0), (2, 1), (5, 1)):
            stream, _ = O.lcpcomp_huff_compress(text, thr, fl)
            if not _decodable(stream, text):          # (256 equal-length codes: the reference cannot decode it either)
                try:
                    dev_ctx.lcpcomp_decompress(stream)
                except T.TdcGpuError:
                    pass
                continue
            back, st = dev_ctx.lcpcomp_decompress(stream)
            assert back == text, "%s t=%d flatten=%d" % (name, thr, fl)
            on_device += st["device_parse"]
        s2, _ = O.lzss_lcp_huff_compress(text, 3)     # same format (LZSSLCPCompressor.hpp:125-130)
        assert dev_ctx.lcpcomp_decompress(s2)[0] == text, name
    assert on_device > len(cases)                     # (streams whose longest literal run exceeds 512 keep the host parse)


def test_reference_decode_sequences_on_the_device(dev_ctx):
    for k in load_json("reference_kats.json")["decode_sequences"]:
        text, f = decode_sequence_case(k)
        stream, _ = O.encode_huff(text, f)
        back, st = dev_ctx.lcpcomp_decompress(stream)
        assert back == text and st["device_parse"] == 1 and st["factors"] == len(f), k["source"]


def test_random_texts_roundtrip_through_the_device_parse(dev_ctx):
    rng = np.random.default_rng(99)
    for trial in range(40):
        sigma = int(rng.integers(2, 60))
        n = int(rng.integers(50, 40_000))
        base = rng.integers(1, sigma + 1, size=n, dtype=np.uint8) + 64
        # planted repeats so that factors of many lengths occur
        for _ in range(int(rng.integers(0, 30))):
            a, l = int(rng.integers(0, n)), int(rng.integers(2, 300))
            b = int(rng.integers(0, n))
            l = min(l, n - a, n - b)
            base[b:b + l] = base[a:a + l].copy()
        text = O.escape(base.tobytes())
        thr = int(rng.integers(1, 7))
        stream, _ = O.lcpcomp_huff_compress(text, thr, int(rng.integers(0, 2)))
        back, st = dev_ctx.lcpcomp_decompress(stream)
        assert back == text, trial


def test_large_stream_takes_the_device_parse_by_default(gpu_ctx):
    """32 MiB English through the product's default decoder path: device parse, references on the device"""
    data = T.gen_english(1 << 25, 42)
    text = np.concatenate([data, np.zeros(1, dtype=np.uint8)]).tobytes()
    got, cst = gpu_ctx.lcpcomp_compress(text, 2, 1)
    back, st = gpu_ctx.lcpcomp_decompress(got)
    assert back == text and st["device_parse"] == 1 and st["factors"] == cst["factors"]
    got5, cst5 = gpu_ctx.lcpcomp_compress(text, 5, 0)
    back, st = gpu_ctx.lcpcomp_decompress(got5)
    assert back == text and st["device_parse"] == 1 and st["factors"] == cst5["factors"]
    dna = np.concatenate([T.gen_dna(1 << 24, 7), np.zeros(1, dtype=np.uint8)]).tobytes()
    gd, cd = gpu_ctx.lcpcomp_compress(dna, 5, 1)
    back, st = gpu_ctx.lcpcomp_decompress(gd)
    assert back == dna and st["factors"] == cd["factors"]


@pytest.mark.parametrize("lean", ["1", "0"])
def test_streams_longer_than_one_segment(lean):
    """the chain marking runs in segments of bit positions (2^30 by default: streams above 128 MiB); with 20 000-bit segments a
    3 MB text takes hundreds of them -- the exit of one segment is the entry of the next"""
    with T.Context(0, options={"dec_parse": 2, "dec_lean": int(lean), "dec_seg": 20000}) as ctx:
        for name, data, thr in (("english", T.gen_english(3_000_000, 4).tobytes(), 2), ("dna", T.gen_dna(1_000_000, 7).tobytes(), 5),
                                ("small", b"abcabcabc hello hello abcabc", 2)):
            text = O.escape(data)
            stream, _ = O.lcpcomp_huff_compress(text, thr, 1)
            back, st = ctx.lcpcomp_decompress(stream)
            assert back == text and st["device_parse"] == 1, name


def test_damaged_streams_are_refused_by_the_device_parse(dev_ctx):
    text = O.escape(T.gen_english(20_000, 3).tobytes())
    good = O.lcpcomp_huff_compress(text, 2, 1)[0]
    assert dev_ctx.lcpcomp_decompress(good)[0] == text
    rng = np.random.default_rng(5)
    refused = 0
    for trial in range(150):
        bad = bytearray(good)
        where = int(rng.integers(0, len(bad)))        # anywhere: header, table, tokens, terminator
        bad[where] ^= 1 << int(rng.integers(0, 8))
        try:
            back, _ = dev_ctx.lcpcomp_decompress(bytes(bad))
            assert len(back) > 0                      # (a flipped literal bit may still be a well-formed stream)
        except T.TdcGpuError as e:
            assert e.status in (-2, -5)
            refused += 1
    assert refused > 0
    for bad in (b"", b"\x00", good[:1000], good[:len(good) // 2] + b"\x05"):
        with pytest.raises(T.TdcGpuError):
            dev_ctx.lcpcomp_decompress(bad)


def test_decompress_into_caller_buffer(gpu_ctx):
    """tdc_gpu_lcpcomp_decompress_into: the text lands in the caller's (pinned) buffer; a buffer that is too small is refused"""
    data = T.gen_english(3_000_000, 21)
    text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
    stream, _ = gpu_ctx.lcpcomp_compress(text, 2, 1)
    out = T.PinnedBuffer(len(text) + 100)
    try:
        out.a[:] = 0xA5
        n, st = gpu_ctx.lcpcomp_decompress_into(stream, out)
        assert n == len(text) and out.a[:n].tobytes() == text.tobytes() and st["device_parse"] == 1
        assert bool((out.a[n:] == 0xA5).all())
        small = np.zeros(1000, dtype=np.uint8)
        with pytest.raises(T.TdcGpuError) as e:
            gpu_ctx.lcpcomp_decompress_into(stream, small)
        assert e.value.status == -5
        tiny_text = O.escape(b"abcabcabc hello hello")
        s2, _ = O.lcpcomp_huff_compress(tiny_text, 2, 1)              # (host parse path into a caller buffer)
        n2, st2 = gpu_ctx.lcpcomp_decompress_into(s2, out)
        assert out.a[:n2].tobytes() == tiny_text and st2["device_parse"] == 0
    finally:
        out.free()


def test_token_counts_that_wrap_32_bits_are_refused(dev_ctx):
    """ADVICE r4 (decode.hip): three factor tokens of length n = 2^31 - 2 plus four literals add up to n modulo 2^32.  With 32-bit
    counts the total looked right and the emit pass wrote literals 4 GiB behind the text; the count pass now sums in 64 bits, a
    token that claims more than the text is an error by itself and the emit pass clamps its room."""
    n = (1 << 31) - 2
    W = 31
    ops = [(0, 0, 1),                               # no Huffman table: literals are 8 raw bits
           (1, n, 32), (1, 0, W), (1, n, W), (1, 4, W)]          # n, flen_min, flen_max, fdist_max -> lbits = 31, dbits = 3
    for _ in range(3):
        ops += [(0, 0, 1), (1, 0, W), (1, n, 31)]   # factor token: no literals, source 0, length n
    ops += [(0, 1, 1), (1, 4, 3)] + [(1, 65, 8)] * 4             # last token: four literals
    stream = O.bitstream_script(ops)
    with pytest.raises(T.TdcGpuError) as e:
        dev_ctx.lcpcomp_decompress(stream)
    assert e.value.status in (-2, -5)
    # many tokens that each fit but wrap together
    n2 = 1 << 30
    ops = [(0, 0, 1), (1, n2, 32), (1, 0, 31), (1, n2, 31), (1, 4, 31)]
    for _ in range(5):
        ops += [(0, 0, 1), (1, 0, 31), (1, n2 if _ < 4 else n2, 31)]    # 5 * 2^30 = 2^32 + 2^30
    stream = O.bitstream_script(ops)
    with pytest.raises(T.TdcGpuError):
        dev_ctx.lcpcomp_decompress(stream)
