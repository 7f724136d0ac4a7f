"""CPU tests: the C-ABI library loads, exports every symbol include/tdc_gpu.h declares, its host-side helpers
agree with the oracle, and it fails loudly (no CPU fallback) when no GPU is usable."""
import os
import re

import numpy as np
import pytest

import tudocomp_amd as T
from oracle import oracle as O
from tests.util import load_json, sha256

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "tdc_gpu.h")).read()
    declared = set(re.findall(r"\b(tdc_(?:gpu_)?[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tdc_gpu_status", "tdc_gpu_ctx", "tdc_gpu_stats"}
    assert declared == set(T.SYMBOLS)
    L = T._native.load()
    for s in declared:
        assert hasattr(L, s), s


def test_strerror():
    L = T._native.load()
    assert L.tdc_gpu_strerror(0) == b"success"
    assert b"sentinel" in L.tdc_gpu_strerror(-3)


def test_escape_matches_oracle_and_kat():
    k = load_json("reference_kats.json")["escaping"]
    assert T.escape(bytes.fromhex(k["raw_hex"])).hex() == k["escaped_hex"]
    rng = np.random.default_rng(1)
    for _ in range(50):
        b = rng.integers(0, 256, size=int(rng.integers(0, 300)), dtype=np.uint8).tobytes()
        if rng.random() < 0.5:
            b = bytes(x if x not in (1, 2) else (0 if x == 1 else 255) for x in b)
        assert T.escape(b) == O.escape(b)
        assert T.unescape(T.escape(b)) == b


def test_generators_match_survey_hashes():
    a = load_json("survey_anchors.json")["texts"]
    for t in a.values():
        data = (T.gen_english if t["gen"] == "english" else T.gen_dna)(t["n"], t["seed"])
        assert sha256(data.tobytes()) == t["sha256"]


def test_host_huffman_table_matches_oracle_restatement():
    """The product calls std::make_heap/pop_heap/push_heap/std::sort like the reference; the oracle restates
    libstdc++'s algorithms.  They must agree, also for sigma > 16 where std::sort is not stable."""
    rng = np.random.default_rng(5)
    for it in range(400):
        sigma = int(rng.integers(2, 257))
        C = np.zeros(256, dtype=np.uint32)
        syms = rng.choice(256, size=sigma, replace=False)
        if it % 3 == 0:
            C[syms] = rng.integers(1, 6, size=sigma)            # many ties
        elif it % 3 == 1:
            C[syms] = rng.integers(1, 1 << 20, size=sigma)
        else:
            C[syms] = (rng.zipf(1.3, size=sigma) % 100000) + 1
        t = T.huffman_table(C)
        o = O.huffman_table(C)
        assert t["sigma"] == o.sigma == sigma and t["longest"] == o.longest
        assert list(t["order"][:sigma]) == list(o.order[:sigma])
        assert list(t["len_of"]) == list(o.len_of)
        assert list(t["code_of"]) == list(o.code_of)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(T.TdcGpuError):
        T.Context(0)


def test_huffman_selfcheck_passes_and_fixture_is_current():
    """the start-up drift check of tdc_gpu_ctx_create (no GPU needed): this build's libstdc++ reproduces the fixture tables, and
    the committed fixture equals what the generator script derives from the oracle today"""
    import subprocess, sys, os
    L = T._native.load()
    assert L.tdc_huffman_selfcheck() == 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "tudocomp_amd", "csrc", "huffman_selfcheck.inc")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:        # (never rewrites the tracked file: a new mtime would make `make` relink the library)
        fresh = os.path.join(tmp, "huffman_selfcheck.inc")
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_huffman_selfcheck.py"), fresh], stdout=subprocess.DEVNULL)
        assert open(fresh).read() == open(inc).read()
