"""CPU tests of the host-side Coder surface (tudocomp_amd/host/tdc_coders.hpp: tdc_amd::Encoder / Decoder, HuffmanCoder,
ASCIICoder, EliasGammaCoder, lzss::encode_text / decode_text_internal) through tudocomp_amd/bin/coder_tool: the streams the
Encoder classes write for the oracle's factor lists are the oracle's streams byte for byte (= the device's, tests/test_gpu_parity),
and the Decoder classes read them back; plus the block container through the `tdc` command line."""
import os
import subprocess

import numpy as np
import pytest

import tudocomp_amd as T
from tudocomp_amd import blocks
from oracle import oracle as O
from tests import corpus
from tests.util import factors_struct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tudocomp_amd", "bin", "coder_tool")
TDC = os.path.join(ROOT, "tudocomp_amd", "bin", "tdc")
SMALL = [c for c in corpus.small_corpus() if len(c[1]) <= 8000]


def _factors(text, threshold):
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    lcp = O.lcp_array(sa, plcp)
    f = O.flatten(O.sort_factors(O.arrays_comp(sa, isa, lcp, maxlcp, threshold)))
    return f[0] if isinstance(f, tuple) else f


@pytest.mark.parametrize("coder", ["huff", "ascii"])
def test_encoder_decoder_classes_match_oracle(tmp_path, coder):
    for name, data in SMALL:
        text = O.escape(data)
        for thr in (2, 5):
            f = _factors(text, thr)
            tri = np.stack([f["pos"], f["src"], f["len"]], axis=1).astype(np.uint32)
            (tmp_path / "t").write_bytes(text)
            (tmp_path / "f").write_bytes(tri.tobytes())
            subprocess.check_call([TOOL, "encode", coder, str(tmp_path / "t"), str(tmp_path / "f"), str(tmp_path / "o")])
            got = (tmp_path / "o").read_bytes()
            want = (O.encode_huff(text, f) if coder == "huff" else O.encode_ascii(text, f))[0]
            assert got == want, (name, thr, coder)
            subprocess.check_call([TOOL, "decode", coder, str(tmp_path / "o"), str(tmp_path / "d")])
            assert (tmp_path / "d").read_bytes() == text, (name, thr, coder)


def test_gamma_coder_classes_match_oracle(tmp_path):
    for name, data in SMALL:
        if not data or data[-1] >= 0x80:
            continue
        ids, chars = O.lz78_factors(data)
        if len(ids) == 0:
            continue
        pairs = np.stack([np.asarray(ids, dtype=np.uint32), np.frombuffer(chars, dtype=np.uint8).astype(np.uint32)], axis=1)
        (tmp_path / "p").write_bytes(pairs.tobytes())
        subprocess.check_call([TOOL, "gamma", str(tmp_path / "p"), str(tmp_path / "o")])
        assert (tmp_path / "o").read_bytes() == O.lz78_gamma_compress(data), name
        subprocess.check_call([TOOL, "ungamma", str(tmp_path / "o"), str(tmp_path / "q")])
        assert (tmp_path / "q").read_bytes() == pairs.tobytes(), name


def test_corrupt_huffman_table_is_rejected(tmp_path):
    """a table that violates Kraft must be refused instead of indexing behind `order[]`"""
    text = O.escape(b"abracadabra abracadabra, simsalabim")
    good = O.lcpcomp_huff_compress(text, 2, 1)[0]
    rng = np.random.default_rng(7)
    refused = 0
    for trial in range(64):                        # (seeded: none of these headers asks for gigabytes -- a tiny stream may legitimately decode to 2 GB)
        bad = bytearray(good)
        for _ in range(3):
            bad[int(rng.integers(0, 6))] ^= 1 << int(rng.integers(0, 8))      # flips inside the table (the text length follows it)
        (tmp_path / "b").write_bytes(bytes(bad))
        rc = subprocess.call([TOOL, "decode", "huff", str(tmp_path / "b"), str(tmp_path / "x")], stderr=subprocess.DEVNULL)
        assert rc in (0, 1)                        # decoded something or refused -- never crashed
        refused += rc
    assert refused > 0


def test_block_container_through_the_command_line(tmp_path):
    """a container of oracle streams (what the GPU block mode writes, byte for byte) is detected and decoded block by block"""
    data = T.gen_english(50000, 42).tobytes() + b"\x00\xff tail with escapes \x00" + T.gen_dna(7000, 7).tobytes()
    bs = 20011
    parts = [data[o:o + bs] for o in range(0, len(data), bs)]
    blob = blocks.pack_container([len(p) for p in parts], [O.lcpcomp_huff_compress(O.escape(p), 2, 1)[0] for p in parts])
    (tmp_path / "c.tdc").write_bytes(b"lcpcomp(coder=huff,threshold=2)%" + blob)
    subprocess.check_call([TDC, "-d", "-f", "-o", str(tmp_path / "back"), str(tmp_path / "c.tdc")])
    assert (tmp_path / "back").read_bytes() == data
    assert blocks.decompress_container(blob, lambda s: O.unescape(O.lcpcomp_huff_decompress(s))) == data
    bad = bytearray(blob)
    bad[20] ^= 0xFF                                # directory entry
    (tmp_path / "bad.tdc").write_bytes(b"lcpcomp(coder=huff,threshold=2)%" + bytes(bad))
    assert subprocess.call([TDC, "-d", "-f", "-o", str(tmp_path / "x"), str(tmp_path / "bad.tdc")], stderr=subprocess.DEVNULL) == 1


def test_lz78_host_parse_matches_oracle():
    """The host LZ78 parse (csrc/lz78_host.cpp: nodes placed at the hash of the string they spell, the slots of a phrase's next depths
    requested ahead) against the oracle's trie walk (compressors/LZ78Compressor.hpp:97-131; the reference's own vector is in
    test_oracle.py): identical (id, char) pairs incl. the leftover phrase, on edge cases, small alphabets (long phrases, window
    top-ups), table growth and texts."""
    import random
    rng = random.Random(1)

    def check(d, name):
        gi, gc = T.lz78_factors(d)
        wi, wc = O.lz78_factors(bytes(d))
        assert len(gi) == len(wi), (name, len(gi), len(wi))
        assert (gi == wi).all() and bytes(gc) == bytes(wc), name

    for i, d in enumerate([b"", b"a", b"aa", b"ab", b"abab" * 10, b"a" * 1000, bytes(range(256)) * 3, b"abcabcabcabd" * 50, b"\xff\x00" * 70]):
        check(d, "case%d" % i)
    for t in range(200):
        n = rng.randrange(1, 5000)
        sig = rng.choice([1, 2, 3, 4, 26, 256])
        check(bytes(rng.randrange(sig) for _ in range(n)), "rand%d" % t)
    check(T.gen_english(1 << 20, 7).tobytes(), "english")
    check(T.gen_dna(1 << 20, 7).tobytes(), "dna")
    check(b"a" * 300000, "run")                                    # phrases of up to 774 bytes: the window is topped up hundreds of times
    check(bytes(rng.randrange(256) for _ in range(200000)), "random bytes")   # z ~ n / 2.3: the table grows
