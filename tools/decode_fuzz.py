"""Robustness run for the device decoder (not a test: minutes of GPU time).  Streams of several texts are damaged at random (bit flips,
truncations, spliced tails) and decoded with both markings of the device parse (lean / general, TDC_GPU_DEC_PARSE=2) and small segments;
every call must either return a text or raise TdcGpuError(-2 / -5) -- never crash, hang or fault.  The undamaged streams must decode to
their texts.  Usage: python3 tools/decode_fuzz.py [trials per text and mode]"""
import os; os.environ.setdefault("TDC_GPU_DEBUG_KNOBS", "1")   # (development tool: the TDC_GPU_* variables below are applied -- include/tdc_gpu.h, options)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(2025)
texts = [("english_60k", T.gen_english(60_000, 3).tobytes(), 2), ("dna_50k", T.gen_dna(50_000, 7).tobytes(), 5),
         ("english_20k_t5", T.gen_english(20_000, 11).tobytes(), 5), ("runs", (b"ab" * 5000 + b"xyz" * 3000 + bytes(range(1, 200)) * 20), 2),
         ("random_64", bytes(rng.integers(65, 129, size=30_000, dtype=np.uint8)), 3)]
streams = [(name, O.escape(d), thr) for name, d, thr in texts]
streams = [(name, text, O.lcpcomp_huff_compress(text, thr, 1)[0]) for name, text, thr in streams]
total = refused = same = other = both = only_gpu = only_oracle = 0
t0 = time.time()
for lean, seg in (("1", None), ("0", None), ("1", "4096"), ("0", "4096")):
    os.environ["TDC_GPU_DEC_PARSE"] = "2"
    os.environ["TDC_GPU_DEC_LEAN"] = lean
    if seg: os.environ["TDC_GPU_DEC_SEG"] = seg
    elif "TDC_GPU_DEC_SEG" in os.environ: del os.environ["TDC_GPU_DEC_SEG"]
    with T.Context(0) as ctx:
        for name, text, good in streams:
            back, st = ctx.lcpcomp_decompress(good)
            assert back == text and st["device_parse"] == 1, (name, lean, seg)
            for trial in range(trials):
                bad = bytearray(good)
                kind = int(rng.integers(0, 10))
                if kind < 7:
                    for _ in range(int(rng.integers(1, 4))):
                        w = int(rng.integers(0, len(bad))); bad[w] ^= 1 << int(rng.integers(0, 8))
                elif kind == 7:
                    bad = bad[:int(rng.integers(1, len(bad)))]
                elif kind == 8:
                    cut = int(rng.integers(1, len(bad))); bad = bad[:cut] + bytes(rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8))
                else:
                    w = int(rng.integers(0, len(bad))); bad[w:w + 8] = bytes(rng.integers(0, 256, size=len(bad[w:w + 8]), dtype=np.uint8))
                total += 1
                try:
                    want = O.lcpcomp_huff_decompress(bytes(bad))          # the oracle's decoder on the same damaged stream
                except RuntimeError:
                    want = None
                try:
                    back, _ = ctx.lcpcomp_decompress(bytes(bad))
                    if back == text: same += 1
                    else: other += 1
                    if want is None: only_gpu += 1
                    else:
                        assert back == want, "device and oracle decode a damaged stream to different texts (%s trial %d)" % (name, trial)
                        both += 1
                except T.TdcGpuError as e:
                    assert e.status in (-2, -5), e.status
                    refused += 1
                    if want is not None: only_oracle += 1
            print("lean=%s seg=%s %-16s done: %d calls, %d refused, %d decoded to the text, %d to another text (%.0f s)"
                  % (lean, seg, name, total, refused, same, other, time.time() - t0), flush=True)
print("decode fuzz: %d damaged streams, %d refused, %d still the text, %d another well-formed text; no crash" % (total, refused, same, other))
print("against the oracle's decoder: %d decoded by both to the same text, %d accepted only by the device, %d only by the oracle" % (both, only_gpu, only_oracle))
