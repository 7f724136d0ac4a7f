"""Small input corpus shared by the CPU and GPU parity tests.

Strings in the spirit of the reference's roundtrip_batch / string generators (test/test/util.hpp:98-207):
empty and tiny inputs, periodic and run-rich words, Fibonacci and Thue-Morse words, random texts over small and
large alphabets, inputs containing 0x00 / 0xFF (escaping changes n) and planted long repeats.
"""
import random


def fib_word(k):
    a, b = b"b", b"a"
    for _ in range(k):
        a, b = b, b + a
    return b


def thue_morse(k):
    s = b"a"
    for _ in range(k):
        s = s + bytes(ord("a") + ord("b") - c for c in s)
    return s


def run_rich(n, rng):
    out = b""
    while len(out) < n:
        out += bytes([rng.randrange(97, 100)]) * rng.randrange(1, 40)
    return out[:n]


def planted(n, sigma, rng, replen=64):
    base = bytes(rng.randrange(65, 65 + sigma) for _ in range(replen))
    out = b""
    while len(out) < n:
        if rng.random() < 0.3:
            k = rng.randrange(1, replen + 1)
            o = rng.randrange(0, replen - k + 1)
            out += base[o:o + k]
        else:
            out += bytes(rng.randrange(65, 65 + sigma) for _ in range(rng.randrange(1, 20)))
    return out[:n]


def small_corpus():
    rng = random.Random(20260101)
    c = [
        ("empty", b""),
        ("a", b"a"),
        ("ab", b"ab"),
        ("aa", b"aa"),
        ("survey_example", b"abcdebcdeabcd abcdebcdeabcd banana bandana"),
        ("abcabc", b"abcabcabcabcabcabcabcabc"),
        ("a^100", b"a" * 100),
        ("a^1000", b"a" * 1000),
        ("ab^300", b"ab" * 300),
        ("banana", b"bananabanana"),
        ("sentence", b"This is a test. This is only a test. Testing, testing, one two three."),
        ("utf8", "größe straße ünïcödé größe straße".encode("utf-8")),
        ("zeros", b"\x00\x00\x00\x00abc\x00\x00"),
        ("ff", b"\xff\xfe\xff\xff\x00\x01\xff\xfe\x00" * 7),
        ("all_bytes", bytes(range(256)) * 3),
        ("fib12", fib_word(12)),
        ("fib17", fib_word(17)),
        ("thue10", thue_morse(10)),
        ("thue13", thue_morse(13)),
        ("runrich", run_rich(3000, rng)),
        ("planted2", planted(5000, 2, rng)),
        ("planted4", planted(20000, 4, rng, replen=300)),
        ("planted26", planted(30000, 26, rng, replen=1000)),
    ]
    for sigma in (2, 3, 5, 17, 40, 200):
        for n in (50, 700, 6000):
            c.append(("rand_s%d_n%d" % (sigma, n), bytes(rng.randrange(1, 1 + sigma) for _ in range(n))))
    return c


def random_small(count, seed):
    rng = random.Random(seed)
    out = []
    for i in range(count):
        kind = rng.randrange(5)
        n = rng.randrange(1, 600)
        sigma = rng.randrange(1, 6)
        if kind == 0:
            s = bytes(rng.randrange(97, 97 + sigma) for _ in range(n))
        elif kind == 1:
            s = planted(n, sigma, rng, replen=rng.randrange(2, 50))
        elif kind == 2:
            s = run_rich(n, rng)
        elif kind == 3:
            w = bytes(rng.randrange(97, 97 + sigma) for _ in range(rng.randrange(1, 8)))
            s = (w * (n // len(w) + 1))[:n]
        else:
            s = bytes(rng.choice([0, 255, 97, 98]) for _ in range(n))
        out.append(("r%d" % i, s))
    return out
