"""The mini `tdc` command line (tudocomp_amd/host): header contract, decompression (host code), error paths on CPU;
compression + round trip on the GPU (BASELINE.json configs[0]: 64 KiB ASCII, lcpcomp(coder=huff,threshold=2))."""
import os
import subprocess

import pytest

import tudocomp_amd as T
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TDC = os.path.join(ROOT, "tudocomp_amd", "bin", "tdc")
ALGO = "lcpcomp(coder=huff,threshold=2)"


def _run(*args):
    return subprocess.run([TDC] + list(args), capture_output=True, text=True)


@pytest.fixture(scope="module", autouse=True)
def _built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tudocomp_amd", "host")])


def test_list():
    r = _run("-l")
    assert r.returncode == 0 and "lcpcomp(coder=huff" in r.stdout


def test_decompress_reads_header(tmp_path):
    data = T.gen_english(5000, 3).tobytes() + b"\x00\xff tail"
    payload, _ = O.lcpcomp_huff_compress(O.escape(data), 2, 1)
    f = tmp_path / "x.tdc"
    f.write_bytes(ALGO.encode() + b"%" + payload)              # tudocomp_driver.cpp:261-266
    out = tmp_path / "x.out"
    r = _run("-d", "-o", str(out), str(f))
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data
    # --raw needs the algorithm on the command line
    f2 = tmp_path / "y.raw"
    f2.write_bytes(payload)
    out2 = tmp_path / "y.out"
    assert _run("-d", "--raw", "-a", ALGO, "-o", str(out2), str(f2)).returncode == 0
    assert out2.read_bytes() == data
    # existing output without -f is refused
    assert _run("-d", "-o", str(out), str(f)).returncode == 1


def test_ascii_coder_decompress(tmp_path):
    """lcpcomp(coder=ascii): host decoder of the facade (ASCIICoder::Decoder, coders/ASCIICoder.hpp:53-84)."""
    data = T.gen_english(3000, 4).tobytes() + bytes([0, 255]) + b" 12:34:"
    payload, _ = O.lcpcomp_ascii_compress(O.escape(data), 3, 1)
    f = tmp_path / "a.tdc"
    f.write_bytes(b"lcpcomp(coder=ascii,threshold=3)%" + payload)
    out = tmp_path / "a.out"
    r = _run("-d", "-o", str(out), str(f))
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data


def test_host_decoders_under_asan(tmp_path):
    """The host decoders (token-stream parse, chain resolution, Huffman / SLE / ASCII / gamma coders of host/tdc_coders.hpp) parse
    untrusted streams: the AddressSanitizer + UBSan build of the command line decodes valid streams of every coder and a series of
    damaged ones (truncated, bit-flipped) without a sanitizer report -- a damaged stream may be refused or decode to something else."""
    host = os.path.join(ROOT, "tudocomp_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host, "asan"])
    tdc_asan = os.path.join(ROOT, "tudocomp_amd", "bin", "tdc_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="halt_on_error=1:exitcode=87")
    data = T.gen_english(4000, 9).tobytes() + b"\x00\xff xyz " * 5
    text = O.escape(data)
    streams = [
        (b"lcpcomp(coder=huff,threshold=2)%", O.lcpcomp_huff_compress(text, 2, 1)[0]),
        (b"lcpcomp(coder=ascii,threshold=3)%", O.lcpcomp_ascii_compress(text, 3, 1)[0]),
        (b"lcpcomp(coder=sle,threshold=3)%", O.lcpcomp_sle_compress(text, 3, 1, 3)[0]),
        (b"lz78(coder=gamma)%", O.lz78_gamma_compress(data)),
    ]
    import random
    rnd = random.Random(5)
    for i, (hdr, payload) in enumerate(streams):
        f = tmp_path / ("s%d.tdc" % i)
        f.write_bytes(hdr + payload)
        out = tmp_path / ("s%d.out" % i)
        r = subprocess.run([tdc_asan, "-d", "-f", "-o", str(out), str(f)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert out.read_bytes() == data
        for trial in range(12):                                   # damaged variants: any exit status but the sanitizers'
            bad = bytearray(payload)
            if trial % 3 == 0:
                bad = bad[:rnd.randrange(0, len(bad))]
            else:
                for _ in range(1 + trial % 4):
                    bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
            g = tmp_path / "bad.tdc"
            g.write_bytes(hdr + bytes(bad))
            r = subprocess.run([tdc_asan, "-d", "-f", "-o", str(tmp_path / "bad.out"), str(g)], capture_output=True, text=True, env=env)
            assert r.returncode not in (86, 87) and "Sanitizer" not in r.stderr, (hdr, trial, r.stderr[-2000:])


@pytest.mark.parametrize("algo,k", [("lcpcomp(coder=sle,threshold=3)", 3), ("lcpcomp(coder=sle(kmer=2),threshold=3)", 2),
                                    ("lcpcomp(coder=sle(1),threshold=3)", 1), ("lcpcomp(coder=sle(kmer=5),threshold=3)", 5)])
def test_sle_coder_decompress(tmp_path, algo, k):
    """lcpcomp(coder=sle(kmer)): host decoder of the facade (SLECoder::Decoder, coders/SLECoder.hpp:301-453)."""
    data = T.gen_english(20000, 4).tobytes() + bytes([0, 255, 7]) + b"the the the "
    payload, _ = O.lcpcomp_sle_compress(O.escape(data), 3, 1, k)
    f = tmp_path / "s.tdc"
    f.write_bytes(algo.encode() + b"%" + payload)
    out = tmp_path / "s.out"
    r = _run("-d", "-o", str(out), str(f))
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data


def test_lzss_lcp_decompress(tmp_path):
    data = T.gen_english(4000, 5).tobytes() + b"\x00\xff"
    payload, _ = O.lzss_lcp_huff_compress(O.escape(data), 3)
    f = tmp_path / "l.tdc"
    f.write_bytes(b"lzss_lcp(coder=huff)%" + payload)
    out = tmp_path / "l.out"
    assert _run("-d", "-o", str(out), str(f)).returncode == 0
    assert out.read_bytes() == data


def test_lz78_decompress(tmp_path):
    data = T.gen_english(3000, 9).tobytes() + bytes(range(256))
    payload = O.lz78_gamma_compress(data[:-128])             # keep the left-over phrase ASCII
    f = tmp_path / "z.tdc"
    f.write_bytes(b"lz78(coder=gamma)%" + payload)
    out = tmp_path / "z.out"
    assert _run("-d", "-o", str(out), str(f)).returncode == 0
    assert out.read_bytes() == data[:-128]


def test_errors(tmp_path):
    f = tmp_path / "in.txt"
    f.write_bytes(b"no header here")
    r = _run("-d", "-o", str(tmp_path / "o"), str(f))
    assert r.returncode == 1 and "algorithm header" in r.stderr
    r = _run("-a", "lz78(coder=bit)", "-o", str(tmp_path / "o2"), str(f))
    assert r.returncode == 1 and "No implementation found" in r.stderr
    r = _run("-a", "lzss_lcp(coder=sle)", "-o", str(tmp_path / "o4"), str(f))
    assert r.returncode == 1 and "No implementation found" in r.stderr
    r = _run("-a", "lzw(coder=huff)", "-o", str(tmp_path / "o5"), str(f))
    assert r.returncode == 1 and "No implementation found" in r.stderr
    r = _run("-a", "lcpcomp(coder=bit)", "-o", str(tmp_path / "o3"), str(f))
    assert r.returncode == 1 and "No implementation found" in r.stderr
    r = _run("-a", "lcpcomp(coder=sle(kmer=9))", "-o", str(tmp_path / "o7"), str(f))
    assert r.returncode == 1 and "kmer" in r.stderr
    g = tmp_path / "arith.tdc"
    g.write_bytes(b"lcpcomp(coder=arithmetic)%" + b"\x00" * 16)
    r = _run("-d", "-o", str(tmp_path / "o6"), str(g))
    assert r.returncode == 1 and "cannot be decoded" in r.stderr


@pytest.mark.gpu
def test_config0_compress_roundtrip_64KiB(tmp_path):
    data = T.gen_english(65536, 42).tobytes()
    f = tmp_path / "english64k.txt"
    f.write_bytes(data)
    r = _run("-a", ALGO, "--stats", str(f))
    assert r.returncode == 0, r.stderr
    comp = (tmp_path / "english64k.txt.tdc").read_bytes()            # default output name: FILE.tdc
    want, _ = O.lcpcomp_huff_compress(O.escape(data), 2, 1)
    assert comp == ALGO.encode() + b"%" + want                       # header + bit-exact payload (38 687 B)
    assert len(want) == 38687
    assert '"factors"' in r.stdout
    # --stats: the reference's schema (tudocomp_driver.cpp:361-391, tudocomp_stat/PhaseData.hpp:79-111)
    import json
    js = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(js["meta"]) >= {"title", "startTime", "config", "input", "inputSize", "output", "outputSize", "rate"}
    assert js["meta"]["config"] == ALGO and js["meta"]["inputSize"] == 65536 and js["meta"]["outputSize"] == len(comp)
    root = js["data"]
    assert set(root) == {"title", "timeStart", "timeEnd", "memOff", "memPeak", "memFinal", "stats", "sub"}
    titles = [p["title"] for p in root["sub"]]
    assert titles == ["Construct Text DS", "Factorize", "Flatten Factors", "Encode Factors"]
    assert [p["title"] for p in root["sub"][0]["sub"]] == ["Construct SA", "Construct Phi Array", "Construct PLCP Array"]
    logged = {kv["key"]: kv["value"] for p in root["sub"] for kv in p["stats"]}
    _, wst = O.lcpcomp_huff_compress(O.escape(data), 2, 1)
    assert logged == {"maxlcp": str(wst["maxlcp"]), "entries": logged["entries"], "threshold": "2", "factors": str(wst["factors"]),
                      "num_flattened": str(wst["num_flattened"]), "max_depth_lb": str(wst["max_depth_lb"])}
    out = tmp_path / "back.txt"
    assert _run("-d", "-o", str(out), str(tmp_path / "english64k.txt.tdc")).returncode == 0
    assert out.read_bytes() == data


@pytest.mark.gpu
def test_decompress_sle_on_device(tmp_path):
    """tdc -d with coder=sle and dec=gpu (tdc_gpu_lcpcomp_decompress_coder)."""
    data = T.gen_english(100_000, 3).tobytes() + bytes([255, 0])
    payload, _ = O.lcpcomp_sle_compress(O.escape(data), 5, 1, 3)
    f = tmp_path / "s.tdc"
    f.write_bytes(b"lcpcomp(coder=sle,threshold=5,dec=gpu)%" + payload)
    out = tmp_path / "s.out"
    r = _run("-d", "-o", str(out), str(f))
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data


@pytest.mark.gpu
def test_decompress_on_device(tmp_path):
    """tdc -d with dec=gpu: host parse, references resolved on the device (tdc_gpu_lcpcomp_decompress)."""
    data = T.gen_english(300_000, 8).tobytes() + bytes([0, 255, 0])
    payload, _ = O.lcpcomp_huff_compress(O.escape(data), 2, 0)
    f = tmp_path / "g.tdc"
    f.write_bytes(b"lcpcomp(coder=huff,threshold=2,dec=gpu)%" + payload)
    out = tmp_path / "g.out"
    r = _run("-d", "-o", str(out), str(f))
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data


@pytest.mark.gpu
def test_config3_lz78_gamma_cli(tmp_path):
    data = T.gen_english(65536, 42).tobytes()
    f = tmp_path / "e.txt"
    f.write_bytes(data)
    assert _run("-a", "lz78(coder=gamma)", str(f)).returncode == 0
    comp = (tmp_path / "e.txt.tdc").read_bytes()
    assert comp == b"lz78(coder=gamma)%" + O.lz78_gamma_compress(data)
    out = tmp_path / "e.back"
    assert _run("-d", "-o", str(out), str(tmp_path / "e.txt.tdc")).returncode == 0
    assert out.read_bytes() == data


@pytest.mark.gpu
def test_lzss_lcp_cli(tmp_path):
    data = T.gen_english(65536, 42).tobytes()
    f = tmp_path / "e.txt"
    f.write_bytes(data)
    assert _run("-a", "lzss_lcp(coder=huff,threshold=3)", str(f)).returncode == 0
    comp = (tmp_path / "e.txt.tdc").read_bytes()
    assert comp == b"lzss_lcp(coder=huff,threshold=3)%" + O.lzss_lcp_huff_compress(O.escape(data), 3)[0]
    out = tmp_path / "e.back"
    assert _run("-d", "-o", str(out), str(tmp_path / "e.txt.tdc")).returncode == 0
    assert out.read_bytes() == data
