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
