#!/usr/bin/env python3
"""Writes tests/golden/oracle_stages.json: per-stage data of the pinned oracle for a handful of small inputs -- SA, ISA, Phi, PLCP,
maxlcp, ArraysComp factors (emission order, sorted, flattened), flatten statistics, the lcpcomp(coder=huff) stream -- so that the
GPU parity tests (tests/test_gpu_parity.py::test_committed_stage_fixture) can also run against committed data, and a change of the
oracle itself is caught on the CPU (tests/test_oracle.py::test_oracle_reproduces_committed_stage_fixture).
Usage: python tests/make_golden.py        (run from the repository root; the oracle must be built)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O
from tests import corpus

INPUTS = [("survey_example", b"abcdebcdeabcd abcdebcdeabcd banana bandana", 2),
          ("fib12", corpus.fib_word(12), 2),
          ("thue10", corpus.thue_morse(10), 3),
          ("zeros_ff", b"\x00\x00\xff\xfeabc\x00\x00abc\xff", 2),
          ("english_2k", T.gen_english(2048, 42).tobytes(), 2),
          ("dna_3k", T.gen_dna(3000, 7).tobytes(), 5)]


def stages(data, thr):
    text = O.escape(data)
    sa = O.suffix_array(text)
    isa, phi, plcp, maxlcp = O.isa_phi_plcp(text, sa)
    lcp = O.lcp_array(sa, plcp)
    raw = O.arrays_comp(sa, isa, lcp, maxlcp, thr)
    srt = O.sort_factors(raw)
    flat, nf, md = O.flatten(srt)
    stream, st = O.lcpcomp_huff_compress(text, thr, 1)
    tri = lambda f: [[int(a), int(b), int(c)] for a, b, c in zip(f["pos"], f["src"], f["len"])]
    return {"threshold": thr, "data_hex": data.hex(), "text_hex": bytes(text).hex(), "sa": sa.tolist(), "isa": isa.tolist(),
            "phi": phi[:-1].tolist(), "plcp": plcp[:-1].tolist(), "maxlcp": int(maxlcp),
            "factors_emitted": tri(raw), "factors_sorted": tri(srt), "factors_flattened": tri(flat),
            "num_flattened": int(nf), "max_depth_lb": int(md), "stream_hex": stream.hex()}


if __name__ == "__main__":
    out = {name: stages(data, thr) for name, data, thr in INPUTS}
    path = os.path.join(ROOT, "tests", "golden", "oracle_stages.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("written", path, os.path.getsize(path), "bytes")
