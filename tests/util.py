import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def pack_fields(fields):
    """(value, bits) pairs -> bytes, MSB first (test::assert_eq_binary of the reference, test/test/util.hpp)."""
    bits = []
    for v, b in fields:
        bits.extend((v >> i) & 1 for i in range(b - 1, -1, -1))
    while len(bits) % 8:
        bits.append(0)
    return bytes(sum(bit << (7 - j) for j, bit in enumerate(bits[i:i + 8])) for i in range(0, len(bits), 8))


def sha256(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def naive_suffix_array(text):
    return np.array(sorted(range(len(text)), key=lambda i: text[i:]), dtype=np.uint32)


def factors_struct(pos, src, length):
    from oracle import oracle as O
    f = np.empty(len(pos), dtype=O.FACTOR_DTYPE)
    f["pos"], f["src"], f["len"] = pos, src, length
    return f


def decode_sequence_case(k):
    """a literal / factor(src, len) sequence of the reference's decode-buffer tests (golden/reference_kats.json "decode_sequences")
    -> (0-terminated text the sequence must decode to, its factors as a FACTOR_DTYPE array in position order).  The literals of the
    sequence are checked against the expected text on the way."""
    want = k["expected"].encode()
    pos, src, length, p = [], [], [], 0
    for tok in k["tokens"]:
        if tok[0] == "lit":
            assert want[p:p + 1] == tok[1].encode(), (k["source"], p)
            p += 1
        else:
            pos.append(p); src.append(tok[1]); length.append(tok[2])
            p += tok[2]
    assert p == len(want)
    return want + b"\0", factors_struct(pos, src, length)
