"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU oracle and the
committed golden vectors.  Bit-exact everywhere (integer / byte / index work)."""
import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O
from tests import corpus
from tests.util import load_json, sha256, factors_struct

pytestmark = pytest.mark.gpu

ANCH = load_json("survey_anchors.json")
SMALL = corpus.small_corpus()
IDS = [c[0] for c in SMALL]


def _eq(name, got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s != %s" % (name, got.shape, want.shape)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "%s: %d mismatches, first at %d: got %s want %s" % (
        name, bad.size, bad[0], got[bad[0]:bad[0] + 8], want[bad[0]:bad[0] + 8])


def _oracle_stages(text):
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    lcp = O.lcp_array(sa, plcp)
    return sa, isa, phi, plcp, lcp, maxlcp


@pytest.mark.parametrize("name,data", SMALL, ids=IDS)
def test_textds_arrays(gpu_ctx, name, data):
    text = O.escape(data)
    sa, isa, phi, plcp, lcp, maxlcp = _oracle_stages(text)
    g = gpu_ctx.textds(text)
    _eq("sa", g["sa"], sa)
    _eq("isa", g["isa"], isa)
    _eq("phi", g["phi"], phi)
    _eq("plcp", g["plcp"], plcp)
    _eq("lcp", g["lcp"], lcp)
    assert g["maxlcp"] == maxlcp


@pytest.mark.parametrize("name,data", SMALL, ids=IDS)
def test_factorize_and_flatten(gpu_ctx, name, data):
    text = O.escape(data)
    sa, isa, phi, plcp, lcp, maxlcp = _oracle_stages(text)
    for thr in (1, 2, 3, 5):
        ref = O.sort_factors(O.arrays_comp(sa, isa, lcp, maxlcp, thr))
        pos, src, ln, st = gpu_ctx.factorize(text, thr, flatten=0)
        _eq("pos t=%d" % thr, pos, ref["pos"])
        _eq("len t=%d" % thr, ln, ref["len"])
        _eq("src t=%d" % thr, src, ref["src"])
        assert st["factors"] == len(ref) and st["maxlcp"] == maxlcp
        fl, nf, md = O.flatten(ref)
        pos2, src2, ln2, st2 = gpu_ctx.factorize(text, thr, flatten=1)
        _eq("flat src t=%d" % thr, src2, fl["src"])
        assert (st2["num_flattened"], st2["max_depth_lb"]) == (nf, md)
        # stand-alone stage entry points on the oracle's factor list
        src3, nf3, md3 = gpu_ctx.flatten(len(text), ref["pos"], ref["src"], ref["len"])
        _eq("flatten() t=%d" % thr, src3, fl["src"])
        assert (nf3, md3) == (nf, md)
        want, _ = O.encode_huff(text, fl)
        got = gpu_ctx.encode_huff(text, fl["pos"], fl["src"], fl["len"])
        assert got == want, "encode_huff t=%d: %d vs %d bytes" % (thr, len(got), len(want))


@pytest.mark.parametrize("name,data", SMALL, ids=IDS)
def test_compress_bitexact_small(gpu_ctx, name, data):
    text = O.escape(data)
    for thr in (1, 2, 5):
        for fl in (0, 1):
            want, wst = O.lcpcomp_huff_compress(text, thr, fl)
            got, st = gpu_ctx.lcpcomp_compress(text, thr, fl)
            assert got == want, "t=%d flatten=%d: %d vs %d bytes" % (thr, fl, len(got), len(want))
            for k in ("factors", "maxlcp", "num_flattened", "max_depth_lb", "flen_max", "fdist_max"):
                assert st[k] == wst[k], k


def test_compress_bitexact_random(gpu_ctx):
    for name, data in corpus.random_small(300, seed=99):
        text = O.escape(data)
        for thr in (1, 2, 5):
            want, _ = O.lcpcomp_huff_compress(text, thr, 1)
            got, _ = gpu_ctx.lcpcomp_compress(text, thr, 1)
            assert got == want, (name, thr, data)


def _ctx_env(env):
    """a context with these options (tdc_gpu_ctx_set_option: the library does not read the environment)"""
    return T.Context(0, options=env)


@pytest.mark.parametrize("mode", ["2", "2_norec", "2_slowread", "0"])
def test_first_half_of_the_encoder_inside_the_flatten_stage(gpu_ctx, mode):
    """TDC_GPU_ENC_EARLY: texts of 1 MiB and more run gaps / histogram / code table / bits per tile on the copy stream next to the
    first flatten round, and the pack takes lengths and flattened sources from the flatten stage's records (the default context does,
    see the medium and full-size tests); 2 does so for every text, 0 for none; TDC_GPU_ENC_REC=0 keeps the early half but packs from
    flen[] / fsrc[]; TDC_GPU_FASTREAD=0 takes the read-backs of the steps through hipMemcpy instead of the mapped host area.  Same streams every way, with and without factors, flatten on and off, also for the coders that never take the
    early half."""
    ctx = _ctx_env({"TDC_GPU_ENC_EARLY": mode[0], "TDC_GPU_ENC_REC": "0" if mode.endswith("norec") else "1",
                    "TDC_GPU_FASTREAD": "0" if mode.endswith("slowread") else "1"})
    try:
        cases = list(SMALL) + list(corpus.random_small(60, seed=5)) + [("english_3M", T.gen_english(3_000_000, 8).tobytes()),
                                                                       ("dna_2M", T.gen_dna(2_000_000, 3).tobytes())]
        for name, data in cases:
            text = O.escape(data)
            for thr in ((2, 5) if len(text) < 100000 else (2,)):
                for fl in (1, 0):
                    got, st = ctx.lcpcomp_compress(text, thr, fl)
                    if len(text) < 100000:
                        want, wst = O.lcpcomp_huff_compress(text, thr, fl)
                        assert st["num_flattened"] == wst["num_flattened"], (name, thr, fl)
                    else:
                        want, _ = gpu_ctx.lcpcomp_compress(text, thr, fl)          # (the default context is checked against the oracle elsewhere)
                        assert O.lcpcomp_huff_decompress(got) == text
                    assert got == want, (name, thr, fl, len(got), len(want))
        text = O.escape(T.gen_english(200_000, 4).tobytes())
        for coder in (T.CODER_ARITH, T.CODER_ASCII):
            a, _ = ctx.lcpcomp_compress(text, 2, 1, coder=coder)
            b, _ = gpu_ctx.lcpcomp_compress(text, 2, 1, coder=coder)
            assert a == b
    finally:
        ctx.close()


def test_survey_example(gpu_ctx):
    e = ANCH["example"]
    comp = T.LCPCompressor(gpu_ctx, coder="huff", threshold=e["threshold"])
    assert comp.compress(e["text"].encode()).hex() == e["output_hex"]
    pos, src, ln, _ = gpu_ctx.factorize(O.escape(e["text"].encode()), e["threshold"], flatten=1)
    assert [list(map(int, t)) for t in zip(pos, src, ln)] == e["factors"]


@pytest.mark.parametrize("a", ANCH["lcpcomp_huff"], ids=lambda a: "%s_t%d" % (a["text"], a["threshold"]))
def test_reference_anchors(gpu_ctx, a):
    t = ANCH["texts"][a["text"]]
    data = (T.gen_english if t["gen"] == "english" else T.gen_dna)(t["n"], t["seed"]).tobytes()
    out = T.LCPCompressor(gpu_ctx, threshold=a["threshold"]).compress(data)
    assert len(out) == a["size"] and sha256(out) == a["sha256"]


@pytest.mark.parametrize("gen,n,thr", [("english", 1 << 22, 2), ("dna", 1 << 22, 5), ("english", 1 << 24, 2), ("english", 1 << 26, 2)])
def test_compress_bitexact_medium(gpu_ctx, gen, n, thr):
    data = (T.gen_english(n, 42) if gen == "english" else T.gen_dna(n, 7)).tobytes()
    text = O.escape(data)
    want, wst = O.lcpcomp_huff_compress(text, thr, 1)
    got, st = gpu_ctx.lcpcomp_compress(text, thr, 1)
    assert len(got) == len(want) and sha256(got) == sha256(want)
    for k in ("factors", "maxlcp", "num_flattened", "max_depth_lb"):
        assert st[k] == wst[k], k
    if gen == "english" and n == 1 << 24:
        a = ANCH["lcpcomp_huff_16MiB"]
        assert (len(got), st["factors"], st["maxlcp"], st["num_flattened"], st["max_depth_lb"]) == (
            a["size"], a["factors"], a["maxlcp"], a["num_flattened"], a["max_depth_lb"])


def test_full_size_roundtrip_256MiB(gpu_ctx):
    """BASELINE.json config 2 at full size: size-independent property -- the oracle's decoder must reproduce the
    input from the GPU stream, and the stream's header fields must be consistent with the reported statistics."""
    n = 1 << 28
    data = T.gen_english(n, 42)
    text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])      # generator emits no 0x00 / 0xFF
    got, st = gpu_ctx.lcpcomp_compress(text, 2, 1)
    assert st["n"] == n + 1 and st["out_len"] == len(got)
    back = O.lcpcomp_huff_decompress(got)
    assert len(back) == n + 1
    assert sha256(back) == sha256(text.tobytes())


def test_error_codes(gpu_ctx):
    with pytest.raises(T.TdcGpuError) as e:
        gpu_ctx.lcpcomp_compress(b"abc", 2, 1)
    assert e.value.status == -3                      # no sentinel: reference throws std::logic_error (TextDS.hpp:132-138)
    with pytest.raises(T.TdcGpuError) as e:
        gpu_ctx.lcpcomp_compress(b"a\x00b\x00", 2, 1)
    assert e.value.status == -2                      # unescaped 0 inside the text
    with pytest.raises(T.TdcGpuError) as e:
        gpu_ctx.lcpcomp_compress(b"abc\x00", 0, 1)
    assert e.value.status == -2
    with pytest.raises(RuntimeError, match="No implementation found"):
        T.LCPCompressor(gpu_ctx, coder="bit")
    with pytest.raises(RuntimeError, match="No implementation found"):
        T.LCPCompressor(gpu_ctx, comp="bogus")
    # the context is still usable afterwards
    want, _ = O.lcpcomp_huff_compress(b"abcabc\x00", 2, 1)
    assert gpu_ctx.lcpcomp_compress(b"abcabc\x00", 2, 1)[0] == want


# ---- BASELINE.json configs[3]: lz78(coder=gamma) ----------------------------------------------------------------
@pytest.mark.parametrize("a", ANCH["lz78_gamma"], ids=lambda a: a["text"])
def test_lz78_gamma_reference_anchors(gpu_ctx, a):
    t = ANCH["texts"][a["text"]]
    data = T.gen_english(t["n"], t["seed"]).tobytes()
    out = T.LZ78Compressor(gpu_ctx, coder="gamma").compress(data)
    assert len(out) == a["size"] and sha256(out) == a["sha256"]


def test_lz78_gamma_bitexact(gpu_ctx):
    cases = [(n, d) for n, d in SMALL if not d or d[-1] < 0x80] + corpus.random_small(100, seed=3)
    for name, data in cases:
        if data and data[-1] >= 0x80:
            continue                                  # left-over phrase as signed char: undefined in the reference
        got, st = gpu_ctx.lz78_compress(data)
        assert got == O.lz78_gamma_compress(data), name
        ids, _ = O.lz78_factors(data)
        assert st["factors"] == len(ids)
    big = T.gen_dna(1 << 22, 7).tobytes()
    assert gpu_ctx.lz78_compress(big)[0] == O.lz78_gamma_compress(big)
    with pytest.raises(RuntimeError, match="No implementation found"):
        T.LZ78Compressor(gpu_ctx, coder="bit")


# ---- lzss_lcp(coder=huff) -----------------------------------------------------------------------
@pytest.mark.parametrize("name,data", SMALL, ids=IDS)
def test_lzss_lcp_small(gpu_ctx, name, data):
    text = O.escape(data)
    sa, isa, phi, plcp, lcp, maxlcp = _oracle_stages(text)
    for thr in (1, 2, 3, 5):
        ref = O.lzss_lcp_factorize(sa, isa, lcp, thr)
        pos, src, ln = gpu_ctx.lzss_lcp_factorize(text, thr)
        _eq("pos t=%d" % thr, pos, ref["pos"])
        _eq("len t=%d" % thr, ln, ref["len"])
        _eq("src t=%d" % thr, src, ref["src"])
        want, _ = O.lzss_lcp_huff_compress(text, thr)
        got, st = gpu_ctx.lzss_lcp_compress(text, thr)
        assert got == want and st["factors"] == len(ref)


def test_lzss_lcp_random_and_medium(gpu_ctx):
    for name, data in corpus.random_small(200, seed=17):
        text = O.escape(data)
        for thr in (1, 3):
            assert gpu_ctx.lzss_lcp_compress(text, thr)[0] == O.lzss_lcp_huff_compress(text, thr)[0], (name, thr, data)
    for gen, n in (("english", 1 << 22), ("dna", 1 << 21), ("english", 3 << 20)):
        data = (T.gen_english(n, 42) if gen == "english" else T.gen_dna(n, 7)).tobytes()
        text = O.escape(data)
        want, wst = O.lzss_lcp_huff_compress(text, 3)
        got, st = gpu_ctx.lzss_lcp_compress(text, 3)
        assert sha256(got) == sha256(want) and st["factors"] == wst["factors"]
        assert O.unescape(O.lcpcomp_huff_decompress(got)) == data
    with pytest.raises(RuntimeError, match="No implementation found"):
        T.LZSSLCPCompressor(gpu_ctx, coder="sle")


# ---- BASELINE.json configs[2]: LCPCompressor + ArithmeticCoder (compress side only) ---------------------
@pytest.mark.parametrize("a", ANCH["lcpcomp_arith"], ids=lambda a: "%s_t%d" % (a["text"], a["threshold"]))
def test_lcpcomp_arithmetic_reference_anchor(gpu_ctx, a):
    t = ANCH["texts"][a["text"]]
    data = T.gen_dna(t["n"], t["seed"]).tobytes()
    out = T.LCPCompressor(gpu_ctx, coder="arithmetic", threshold=a["threshold"]).compress(data)
    assert len(out) == a["size"] and sha256(out) == a["sha256"]


def _arith_or_unsupported(gpu_ctx, text, thr, fl):
    try:
        want, _ = O.lcpcomp_arith_compress(text, thr, fl)
    except RuntimeError:                                  # the reference divides by zero on this input
        with pytest.raises(T.TdcGpuError) as e:
            gpu_ctx.lcpcomp_compress(text, thr, fl, T.CODER_ARITH)
        assert e.value.status == -6
        return
    got, _ = gpu_ctx.lcpcomp_compress(text, thr, fl, T.CODER_ARITH)
    assert got == want, "arithmetic t=%d: %d vs %d bytes" % (thr, len(got), len(want))


@pytest.mark.parametrize("name,data", SMALL, ids=IDS)
def test_lcpcomp_arithmetic_small(gpu_ctx, name, data):
    text = O.escape(data)
    for thr in (1, 2, 5):
        _arith_or_unsupported(gpu_ctx, text, thr, 1)


def test_lcpcomp_arithmetic_random_and_medium(gpu_ctx):
    for name, data in corpus.random_small(150, seed=23):
        _arith_or_unsupported(gpu_ctx, O.escape(data), 2, 1)
    for gen, n, thr in (("english", 1 << 22, 2), ("dna", 1 << 22, 5), ("english", 1 << 24, 5)):
        data = (T.gen_english(n, 42) if gen == "english" else T.gen_dna(n, 7)).tobytes()
        text = O.escape(data)
        want, _ = O.lcpcomp_arith_compress(text, thr, 1)
        got, _ = gpu_ctx.lcpcomp_compress(text, thr, 1, T.CODER_ARITH)
        assert len(got) == len(want) and sha256(got) == sha256(want)
    # a skewed literal distribution (long segments): exercises the sequential fall-back
    skew = bytes([97] * 200000 + [98] * 3) * 3 + bytes(range(33, 120))
    _arith_or_unsupported(gpu_ctx, O.escape(skew), 50000, 1)


# ---- escaping + sentinel on the device --------------------------------------------------------------
def test_device_escape_matches_host(gpu_ctx):
    k = load_json("reference_kats.json")["escaping"]
    raw = bytes.fromhex(k["raw_hex"])
    assert gpu_ctx.lcpcomp_compress_raw(raw, 2, 1)[0] == O.lcpcomp_huff_compress(bytes.fromhex(k["escaped_hex"]), 2, 1)[0]
    rng = np.random.default_rng(3)
    cases = [d for _, d in SMALL] + [rng.integers(0, 256, size=int(s), dtype=np.uint8).tobytes() for s in (1, 77, 4097, 300000)]
    cases.append(bytes([0, 255]) * 5000)
    for data in cases:
        want, _ = O.lcpcomp_huff_compress(O.escape(data), 3, 1)
        got, st = gpu_ctx.lcpcomp_compress_raw(data, 3, 1)
        assert got == want and st["n"] == len(O.escape(data))


def test_lcpcomp_ascii_coder(gpu_ctx):
    """lcpcomp(coder=ascii): device stream == oracle == the reference's recorded example; round trip through
    the oracle's ASCII decoder."""
    e = ANCH["example"]
    text = O.escape(e["text"].encode())
    got, _ = gpu_ctx.lcpcomp_compress(text, e["threshold"], 1, T.CODER_ASCII)
    assert got == e["ascii_output"].encode("latin-1") + bytes(1)
    cases = [c for c in SMALL] + [("english_300k", T.gen_english(300_000, 3).tobytes()), ("dna_100k", T.gen_dna(100_000, 7).tobytes())]
    for name, data in cases:
        text = O.escape(data)
        for thr in (1, 2, 5):
            want, _ = O.lcpcomp_ascii_compress(text, thr, 1)
            got, st = gpu_ctx.lcpcomp_compress(text, thr, 1, T.CODER_ASCII)
            assert got == want, "%s t=%d: %d vs %d bytes" % (name, thr, len(got), len(want))
    # stand-alone encoder entry point on a given factor list
    text = O.escape(T.gen_english(50_000, 11).tobytes())
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    f = O.flatten(O.sort_factors(O.arrays_comp(sa, isa, O.lcp_array(sa, plcp), maxlcp, 2)))[0]
    want, _ = O.encode_ascii(text, f)
    assert gpu_ctx.encode_ascii(text, f["pos"], f["src"], f["len"]) == want
    assert O.lcpcomp_ascii_decompress(want) == text


def test_device_decompress(gpu_ctx):
    """LCPCompressor::decompress with the references resolved on the device: oracle streams and GPU streams
    decode to the escaped text; lzss_lcp streams share the format; malformed input is rejected, not crashed on."""
    cases = [c for c in SMALL] + [("english_1M", T.gen_english(1 << 20, 42).tobytes()), ("dna_300k", T.gen_dna(300_000, 7).tobytes())]
    for name, data in cases:
        text = O.escape(data)
        for thr, fl in ((1, 1), (2, 0), (5, 1)):
            stream, _ = O.lcpcomp_huff_compress(text, thr, fl)
            try:
                decodable = O.lcpcomp_huff_decompress(stream) == text
            except RuntimeError:
                decodable = False
            if not decodable:
                # 256 literal codes of one length overflow the reference's u8 counters (HuffmanCoder.hpp:173-187): the
                # reference cannot decode such a stream either; the device path must fail cleanly or agree with it
                try:
                    gpu_ctx.lcpcomp_decompress(stream)
                except T.TdcGpuError:
                    pass
                continue
            back, st = gpu_ctx.lcpcomp_decompress(stream)
            assert back == text, "%s t=%d flatten=%d" % (name, thr, fl)
        s2, _ = O.lzss_lcp_huff_compress(text, 3)
        assert gpu_ctx.lcpcomp_decompress(s2)[0] == text, name
    data = T.gen_english(1 << 24, 42)
    text = np.concatenate([data, np.zeros(1, dtype=np.uint8)]).tobytes()
    got, cst = gpu_ctx.lcpcomp_compress(text, 2, 0)                     # unflattened: deep source chains
    back, st = gpu_ctx.lcpcomp_decompress(got)
    assert back == text and st["factors"] == cst["factors"] and st["rounds"] >= 2
    for bad in (b"", b"\x00", got[:1000], got[:len(got) // 2] + b"\x05"):
        with pytest.raises(T.TdcGpuError):
            gpu_ctx.lcpcomp_decompress(bad)


def test_lcpcomp_plcppeaks_strategy(gpu_ctx):
    """lcpcomp(comp=plcppeaks): the peak scan as the orbit of position 0, bit-exact with the oracle's
    restatement of PLCPPeaksStrategy (which the reference's tests do not pin: checked by its properties -- valid copies,
    round trip through the decoder)."""
    cases = [c for c in SMALL] + [("english_400k", T.gen_english(400_000, 21).tobytes()), ("dna_200k", T.gen_dna(200_000, 7).tobytes())]
    for name, data in cases:
        text = O.escape(data)
        for thr in (1, 2, 5):
            want, wst = O.lcpcomp_peaks_huff_compress(text, thr, 1)
            got, st = gpu_ctx.lcpcomp_compress(text, thr, 1, T.CODER_HUFF, T.COMP_PLCPPEAKS)
            assert got == want, "%s t=%d: %d vs %d bytes" % (name, thr, len(got), len(want))
            assert st["factors"] == wst["factors"]
            try:
                ok = O.lcpcomp_huff_decompress(want) == text
            except RuntimeError:
                ok = None                       # 256 equal-length codes: undecodable for the reference as well
            assert ok in (True, None), name


def _sle_cases():
    rng = np.random.default_rng(11)
    cases = [c for c in SMALL]
    cases += [("english_300k", T.gen_english(300_000, 3).tobytes()), ("dna_100k", T.gen_dna(100_000, 7).tobytes()),
              ("random_100k", rng.integers(0, 256, 100_000, dtype=np.uint8).tobytes()),           # sigma_bits >= 7, literal runs of many tiles
              ("abc_periodic", b"abc" * 20_000 + b"xyz" + b"bca" * 10_000)]
    for sig in (2, 5, 9, 17, 33, 65, 130):                                                        # every rank class of encode_sym
        cases.append(("sigma%d" % sig, bytes(1 + int(x) % sig for x in rng.integers(0, 1 << 30, 20_000))))
    return cases


def test_lcpcomp_sle_coder(gpu_ctx):
    """lcpcomp(coder=sle(kmer)) (coders/SLECoder.hpp): device stream == oracle for kmer 1..7 (counter table for
    kmer <= 3, sorted windows above), round trip through the oracle's SLE decoder."""
    for name, data in _sle_cases():
        text = O.escape(data)
        for thr in (2, 5):
            for k in (1, 2, 3, 4, 5, 7):
                want, _ = O.lcpcomp_sle_compress(text, thr, 1, k)
                got, st = gpu_ctx.lcpcomp_compress(text, thr, 1, T.CODER_SLE | (k << 8))
                assert got == want, "%s t=%d k=%d: %d vs %d bytes" % (name, thr, k, len(got), len(want))
        assert O.lcpcomp_sle_decompress(got, k) == text
    text = O.escape(T.gen_english(5000, 1).tobytes())
    got, _ = gpu_ctx.lcpcomp_compress(text, 5, 1, T.CODER_SLE)                                    # kmer omitted = 3
    assert got == O.lcpcomp_sle_compress(text, 5, 1, 3)[0]
    with pytest.raises(T.TdcGpuError):
        gpu_ctx.lcpcomp_compress(text, 5, 1, T.CODER_SLE | (8 << 8))                              # max_kmer = 7 (SLECoder.hpp:12)
    c = T.LCPCompressor(gpu_ctx, coder="sle", threshold=5)
    data = T.gen_dna(50_000, 3).tobytes() + bytes([0, 255])
    assert O.unescape(O.lcpcomp_sle_decompress(c.compress(data), 3)) == data


def test_sle_literal_runs_without_factors(gpu_ctx):
    """SLE k-mer buffer state carried across tiles: one literal run over the whole text (no factors), texts in which all
    three alignments of a k-mer are ranked, and factor lists that cut the runs at every residue."""
    rng = np.random.default_rng(3)
    texts = [b"abc" * 30_000 + b"\0", b"ab" * 40_000 + b"\0", O.escape(T.gen_english(100_000, 9).tobytes()),
             bytes(rng.integers(1, 4, 70_000, dtype=np.uint8)) + b"\0"]
    none = np.zeros(0, dtype=np.uint32)
    for text in texts:
        for k in (1, 2, 3, 4, 6):
            want, _ = O.encode_sle(text, factors_struct(none, none, none), k)
            assert gpu_ctx.encode_sle(text, none, none, none, k) == want
            # factors of length 2 every 7..23 positions (sources are irrelevant to the coder)
            pos = np.cumsum(rng.integers(7, 24, len(text) // 16)).astype(np.uint32)
            pos = pos[pos + 2 < len(text) - 1]
            src = np.zeros_like(pos); ln = np.full_like(pos, 2)
            want, _ = O.encode_sle(text, factors_struct(pos, src, ln), k)
            assert gpu_ctx.encode_sle(text, pos, src, ln, k) == want


def test_lcpcomp_max_lcp_strategy(gpu_ctx):
    """lcpcomp(comp=max_lcp) (MaxLCPStrategy.hpp:36-100 over MaxLCPSuffixList.hpp): the per-level stacks and the
    eager key decreases as explicit priorities; bit-exact with the oracle's linked-list restatement (different tie order
    than ArraysComp, so the streams differ from comp=arrays)."""
    rng = np.random.default_rng(17)
    cases = [c for c in SMALL] + [("english_400k", T.gen_english(400_000, 21).tobytes()), ("dna_200k", T.gen_dna(200_000, 7).tobytes()),
                                  ("abc_periodic", b"abc" * 5000 + b"x" + b"cab" * 3000),
                                  ("run_a", b"a" * 3000 + b"b" + b"a" * 2000),
                                  ("sigma2", bytes(rng.integers(97, 99, 60_000, dtype=np.uint8)))]
    differs = 0
    for name, data in cases:
        text = O.escape(data)
        for thr in (1, 2, 5):
            want, wst = O.lcpcomp_maxlcp_huff_compress(text, thr, 1)
            got, st = gpu_ctx.lcpcomp_compress(text, thr, 1, T.CODER_HUFF, T.COMP_MAXLCP)
            assert got == want, "%s t=%d: %d vs %d bytes" % (name, thr, len(got), len(want))
            assert st["factors"] == wst["factors"]
            differs += got != O.lcpcomp_huff_compress(text, thr, 1)[0]
    assert differs > 0
    c = T.LCPCompressor(gpu_ctx, coder="huff", threshold=3, comp="max_lcp")
    data = T.gen_english(80_000, 2).tobytes() + bytes([0, 255, 0])
    assert c.decompress(c.compress(data)) == data


def test_device_decompress_reference_sequences(gpu_ctx):
    """the three literal / factor sequences of test/lzss_test.cpp:141-189 (back references, chained forward references, several
    forward references into one factor; golden/reference_kats.json) through the device encoder and the device's reference resolver"""
    from tests.util import decode_sequence_case
    for k in load_json("reference_kats.json")["decode_sequences"]:
        text, f = decode_sequence_case(k)
        stream = gpu_ctx.encode_huff(text, f["pos"], f["src"], f["len"])
        assert stream == O.encode_huff(text, f)[0], k["source"]
        back, st = gpu_ctx.lcpcomp_decompress(stream)
        assert back == text and st["factors"] == len(f), k["source"]
        a = gpu_ctx.encode_ascii(text, f["pos"], f["src"], f["len"])
        assert gpu_ctx.lcpcomp_decompress(a, T.CODER_ASCII)[0] == text, k["source"]


def test_device_decompress_rejects_corrupt_tables(gpu_ctx):
    """fuzzed Huffman tables / headers: the host parse in front of the device resolver refuses or decodes, never reads
    behind its tables"""
    text = O.escape(T.gen_english(5000, 3).tobytes())
    good = O.lcpcomp_huff_compress(text, 2, 1)[0]
    rng = np.random.default_rng(17)
    refused = 0
    for trial in range(100):
        bad = bytearray(good)
        for _ in range(3):
            bad[int(rng.integers(0, 16))] ^= 1 << int(rng.integers(0, 8))
        try:
            gpu_ctx.lcpcomp_decompress(bytes(bad))
        except T.TdcGpuError as e:
            assert e.status in (-2, -5)
            refused += 1
    assert refused > 0
    assert gpu_ctx.lcpcomp_decompress(good)[0] == text


def test_committed_stage_fixture(gpu_ctx):
    """the HIP path against COMMITTED per-stage data (tests/golden/oracle_stages.json, tests/make_golden.py), no live oracle involved"""
    fx = load_json("oracle_stages.json")
    for name, g in fx.items():
        text = bytes.fromhex(g["text_hex"])
        thr = g["threshold"]
        arrs = gpu_ctx.textds(text)
        _eq(name + " sa", arrs["sa"], g["sa"]); _eq(name + " isa", arrs["isa"], g["isa"])
        _eq(name + " phi", arrs["phi"][:-1], g["phi"]); _eq(name + " plcp", arrs["plcp"][:-1], g["plcp"])
        assert arrs["maxlcp"] == g["maxlcp"]
        pos, src, length, st = gpu_ctx.factorize(text, thr, 0)
        assert [list(map(int, t)) for t in zip(pos, src, length)] == g["factors_sorted"], name
        pos, src, length, st = gpu_ctx.factorize(text, thr, 1)
        assert [list(map(int, t)) for t in zip(pos, src, length)] == g["factors_flattened"], name
        assert (st["num_flattened"], st["max_depth_lb"]) == (g["num_flattened"], g["max_depth_lb"])
        assert gpu_ctx.lcpcomp_compress(text, thr, 1)[0].hex() == g["stream_hex"], name
        assert T.escape(bytes.fromhex(g["data_hex"])) == text


def test_lcpcomp_heap_strategy(gpu_ctx):
    """comp=heap (lcpcomp::MaxHeapStrategy, MaxHeapStrategy.hpp:36-101 over ds/ArrayMaxHeap.hpp): the device replays the reference's
    heap loop; streams byte-identical to the oracle's restatement (which the reference does not pin: no vector for this strategy),
    valid round trips, and a different factorization than comp=arrays where ties are broken differently"""
    differs = 0
    cases = SMALL + corpus.random_small(120, seed=77) + [("english64k", T.gen_english(1 << 16, 42).tobytes()), ("dna32k", T.gen_dna(1 << 15, 7).tobytes())]
    for name, data in cases:
        text = O.escape(data)
        for thr in (2, 5):
            want, _ = O.lcpcomp_heap_huff_compress(text, thr, 1)
            got, st = gpu_ctx.lcpcomp_compress(text, thr, 1, T.CODER_HUFF, T.COMP_HEAP)
            assert got == want, (name, thr)
            assert O.lcpcomp_huff_decompress(got) == text
            differs += got != O.lcpcomp_huff_compress(text, thr, 1)[0]
    assert differs > 0
    assert T.LCPCompressor(gpu_ctx, comp="heap", threshold=2).compress(b"abcabcabc abcabc") == O.lcpcomp_heap_huff_compress(O.escape(b"abcabcabc abcabc"), 2, 1)[0]


def test_fuzz_all_variants(gpu_ctx):
    """Structured random texts (tiny alphabets, planted repeats, runs, bytes that need escaping) through every coder and
    strategy of the lcpcomp entry point: device stream == oracle stream."""
    rng = np.random.default_rng(2024)

    def make(i):
        kind = i % 5
        n = int(rng.integers(1, 2500))
        if kind == 0:
            return bytes(rng.integers(97, 97 + int(rng.integers(1, 5)), n, dtype=np.uint8))
        if kind == 1:
            base = bytes(rng.integers(0, 256, max(1, n // 8), dtype=np.uint8))
            out = bytearray()
            while len(out) < n:
                s = int(rng.integers(0, len(base)))
                out += base[s:s + int(rng.integers(1, 60))]
            return bytes(out[:n])
        if kind == 2:
            return bytes([int(rng.integers(65, 70))]) * int(rng.integers(1, 400)) + bytes(rng.integers(65, 70, n // 4 + 1, dtype=np.uint8)) * 3
        if kind == 3:
            w = [bytes(rng.integers(97, 123, int(rng.integers(1, 7)), dtype=np.uint8)) for _ in range(12)]
            return b" ".join(w[int(x)] for x in rng.integers(0, 12, n // 4 + 1))[:n]
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))

    variants = [("huff/arrays", T.CODER_HUFF, T.COMP_ARRAYS, O.lcpcomp_huff_compress),
                ("huff/max_lcp", T.CODER_HUFF, T.COMP_MAXLCP, O.lcpcomp_maxlcp_huff_compress),
                ("huff/plcppeaks", T.CODER_HUFF, T.COMP_PLCPPEAKS, O.lcpcomp_peaks_huff_compress),
                ("ascii/arrays", T.CODER_ASCII, T.COMP_ARRAYS, O.lcpcomp_ascii_compress)]
    for k in (1, 2, 3, 4, 6):
        variants.append(("sle%d/arrays" % k, T.CODER_SLE | (k << 8), T.COMP_ARRAYS,
                         lambda t, thr, fl, k=k: O.lcpcomp_sle_compress(t, thr, fl, k)))
    for i in range(120):
        text = O.escape(make(i))
        thr = int(rng.integers(1, 7))
        fl = int(rng.integers(0, 2))
        for name, coder, comp, oracle_fn in variants:
            want, _ = oracle_fn(text, thr, fl)
            got, _ = gpu_ctx.lcpcomp_compress(text, thr, fl, coder, comp)
            assert got == want, "case %d %s t=%d flatten=%d n=%d" % (i, name, thr, fl, len(text))


def test_device_decompress_sle_and_ascii(gpu_ctx):
    """tdc_gpu_lcpcomp_decompress_coder: SLE and ASCII streams (host parse, references resolved on the device) -- oracle
    streams decode to the text, device streams round-trip, damaged streams are rejected."""
    cases = [("english", T.gen_english(200_000, 5).tobytes()), ("dna", T.gen_dna(120_000, 7).tobytes()),
             ("bytes", bytes(np.random.default_rng(8).integers(0, 256, 30_000, dtype=np.uint8)))]
    for name, data in cases:
        text = O.escape(data)
        for k in (1, 3, 5):
            s, _ = O.lcpcomp_sle_compress(text, 3, 1, k)
            back, st = gpu_ctx.lcpcomp_decompress(s, T.CODER_SLE | (k << 8))
            assert back == text, (name, k)
            g, _ = gpu_ctx.lcpcomp_compress(text, 3, 1, T.CODER_SLE | (k << 8), T.COMP_MAXLCP)
            assert gpu_ctx.lcpcomp_decompress(g, T.CODER_SLE | (k << 8))[0] == text
        s, _ = O.lcpcomp_ascii_compress(text, 4, 1)
        back, st = gpu_ctx.lcpcomp_decompress(s, T.CODER_ASCII)
        assert back == text, name
    c = T.LCPCompressor(gpu_ctx, coder="sle", threshold=4, kmer=2)
    data = T.gen_english(50_000, 1).tobytes() + bytes([0, 255])
    assert c.decompress(c.compress(data)) == data
    s, _ = O.lcpcomp_sle_compress(O.escape(b"abcabcabcabc hello hello hello"), 2, 1, 3)
    for bad in (s[:3], b"\xff" * 40, s[:len(s) // 2]):
        with pytest.raises(T.TdcGpuError):
            gpu_ctx.lcpcomp_decompress(bad, T.CODER_SLE)
    with pytest.raises(T.TdcGpuError):
        gpu_ctx.lcpcomp_decompress(b"12:x", T.CODER_ASCII)


def test_poisoned_environment_does_not_change_the_library(monkeypatch):
    """The shipped library reads no TDC_GPU_* option from the environment: with every algorithm switch set to its
    non-default value in the environment -- and TDC_GPU_DEBUG_KNOBS not set -- a new context takes exactly the default paths (wide
    suffix sort with level 1 behind the upload semantics, window pass, eager levels, device parse) and produces the oracle's stream.
    With TDC_GPU_DEBUG_KNOBS=1 the same variables are applied (development aid), through the same function as explicit options."""
    poison = {"TDC_GPU_WSORT": "0", "TDC_GPU_SSORT": "0", "TDC_GPU_EAGER": "0", "TDC_GPU_WINDOW_LCUT": "0", "TDC_GPU_ENC_EARLY": "0",
              "TDC_GPU_PHI_LAZY": "0", "TDC_GPU_FLEN_BYTES": "0", "TDC_GPU_DEC_PARSE": "0", "TDC_GPU_FASTREAD": "0", "TDC_GPU_SA_REFINE": "0",
              "TDC_GPU_WINDOW_FORCE_FAIL": "1", "TDC_GPU_WSORT_KW": "1", "TDC_GPU_BUCKET_SCATTER": "0", "TDC_GPU_RADIX_LDS": "0"}
    monkeypatch.delenv("TDC_GPU_DEBUG_KNOBS", raising=False)
    for k, v in poison.items():
        monkeypatch.setenv(k, v)
    data = T.gen_english(3_000_000, 42).tobytes()
    text = O.escape(data)
    want, _ = O.lcpcomp_huff_compress(text, 2, 1)
    with T.Context(0) as plain:
        got, st = plain.lcpcomp_compress(text, threshold=2, flatten=1)
        assert got == want
        assert st["window_pass"] == 1 and st["sa_key_words"] == 2, st          # the default paths, whatever the environment says
        back, dst = plain.lcpcomp_decompress(got)
        assert back == text and dst["device_parse"] == 1
    monkeypatch.setenv("TDC_GPU_DEBUG_KNOBS", "1")
    with T.Context(0) as dev:
        got2, st2 = dev.lcpcomp_compress(text, threshold=2, flatten=1)
        assert got2 == want
        assert st2["window_pass"] == 0 and st2["sa_key_words"] != 2, st2        # now the variables count


def test_options_api():
    """tdc_gpu_ctx_set_option: every name of the table is accepted (with and without the TDC_GPU_ prefix, any case), unknown names are
    refused, and an option set after creation takes effect on the next call."""
    names = T.option_names()
    assert len(names) >= 40 and len(set(names)) == len(names)
    with T.Context(0) as ctx:
        with pytest.raises(T.TdcGpuError):
            ctx.set_option("no_such_option", 1)
        text = O.escape(T.gen_english(1 << 21, 5).tobytes())
        want, _ = O.lcpcomp_huff_compress(text, 2, 1)
        a, sa_ = ctx.lcpcomp_compress(text, threshold=2, flatten=1)
        ctx.set_option("TDC_GPU_WINDOW_LCUT", 0)
        b, sb_ = ctx.lcpcomp_compress(text, threshold=2, flatten=1)
        ctx.set_option("Window_Lcut", 48)
        c, sc_ = ctx.lcpcomp_compress(text, threshold=2, flatten=1)
        assert a == want and b == want and c == want
        assert (sa_["window_pass"], sb_["window_pass"], sc_["window_pass"]) == (1, 0, 1)


# every algorithm switch of the option table with its non-default values (diagnostic options -- *_log, eager_dump, small_prof, arena_log --
# only print; dec_* are exercised by test_gpu_decode.py, ssort_levels / wsort_* in depth by test_gpu_sort.py / test_gpu_wsort.py)
OPTION_VALUES = {
    "fastread": [0], "sa_local": [0, 2], "radix_waves": [8], "window_lcut": [0, 20, 48], "window_halo": [0, 384], "window_force_fail": [1],
    "window_large": [1], "window_src": [0], "plcp_samples": [0], "small_pipeline": [0], "small_big": [0], "phi_lazy": [0], "fs_pair": [0],
    "enc_early": [0, 2], "enc_rec": [0], "owner_rem": [0, 1, 3], "level_purge": [0], "eager": [0], "flen_bytes": [0], "flatten_steps": [0, 4], "flatten_growth": [2],
    "sa_refine": [0], "sa_pairs": [0], "sa_stars": [0], "sa_seg_rounds": [0, 1], "sa_seg_bigcap": [0], "sa_fused_init": [0], "sa_init_syms": [5], "radix_lds": [0, 2], "xcd_remap": [0, 2],
    "bucket_scatter": [0], "ssort": [0], "ssort_levels": [2], "msd_partition": [0], "wsort": [0], "wsort_min": [4096], "wsort_syms": [19],
    "wsort_kw": [1], "wsort_rounds": [0], "wsort_smallrun": [1], "wsort_overlap": [0], "wsort_predig": [0], "wsort_predig_skip": [0, 6], "wsort_prehist": [0], "wsort_run_streams": [0], "wsort_fuse": [0], "wsort_order": [0],
    "wsort_two": [2], "wsort_leaf": [1024], "wsort_pack": [2048, 4096], "wsort_cmax": [8, 16, 64], "upload_chunks": [4, 24], "upload_tail_n": [0, 8], "upload_tail_pct": [85],
    "dec_seg": [1 << 20], "dec_lean": [0], "dec_parse": [0, 2], "dec_done": [0],
}


def test_every_option_value_is_bit_exact(gpu_ctx):
    """One named test for every switch of the library: each option of tdc_gpu_ctx_set_option's table, at each of its
    non-default values, compresses an English-like text (wide suffix sort, window pass) and a DNA text with copied blocks (doubling
    fall-back, eager levels, one-workgroup levels) to the oracle's stream and decompresses it again; the options that only matter for
    host-buffer calls of 2^26 bytes and more are compared with the default context on such a text."""
    names = set(T.option_names())
    diag = {n for n in names if n.endswith("_log")} | {"eager_dump", "small_prof"}
    assert names - diag == set(OPTION_VALUES), sorted((names - diag) ^ set(OPTION_VALUES))
    rng = np.random.default_rng(3)
    blk = bytes(rng.integers(0, 4, 40_000, dtype=np.uint8).astype(np.uint8) + 65)
    texts = [O.escape(T.gen_english(3_000_000, 17).tobytes()),
             O.escape(T.gen_dna(1_200_000, 5).tobytes() + blk + b"#" + blk[100:30_000] + T.gen_dna(300_000, 6).tobytes() + blk[5_000:])]
    wants = [O.lcpcomp_huff_compress(t, 2, 1)[0] for t in texts]
    big = np.concatenate([T.gen_english((1 << 26) + 4321, 3), np.zeros(1, dtype=np.uint8)])
    big_want, _ = gpu_ctx.lcpcomp_compress(big, threshold=2, flatten=1)
    for name, values in sorted(OPTION_VALUES.items()):
        for v in values:
            with T.Context(0, options={name: v}) as ctx:
                for t, w in zip(texts, wants):
                    got, st = ctx.lcpcomp_compress(t, threshold=2, flatten=1)
                    assert got == w, (name, v)
                    back, _ = ctx.lcpcomp_decompress(got)
                    assert back == t, (name, v)
                if name in ("upload_chunks", "upload_tail_n", "upload_tail_pct", "wsort_overlap", "wsort_predig", "wsort_predig_skip", "wsort_prehist", "xcd_remap", "wsort_two"):
                    got, _ = ctx.lcpcomp_compress(big, threshold=2, flatten=1)
                    assert got == big_want, (name, v)
