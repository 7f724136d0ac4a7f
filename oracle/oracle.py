"""ctypes binding of the CPU oracle (oracle/tdc_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (tudocomp_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("tdc_oracle.c", "tdc_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


class Factor(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_uint32), ("src", ctypes.c_uint32), ("len", ctypes.c_uint32)]


class HuffTable(ctypes.Structure):
    _fields_ = [("sigma", ctypes.c_uint32), ("longest", ctypes.c_uint32),
                ("numl", ctypes.c_uint8 * 256), ("order", ctypes.c_uint8 * 256),
                ("len_of", ctypes.c_uint8 * 256), ("code_of", ctypes.c_uint64 * 256)]


class Stats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in
                ("n", "factors", "maxlcp", "num_flattened", "max_depth_lb", "flen_min", "flen_max", "fdist_max")] + \
               [(k, ctypes.c_double) for k in
                ("t_sa", "t_phi", "t_plcp", "t_isa", "t_factorize", "t_sort", "t_flatten", "t_encode", "t_total")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


FACTOR_DTYPE = np.dtype([("pos", "<u4"), ("src", "<u4"), ("len", "<u4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        L = _lib
        u8p, u32p, u64p = (ctypes.POINTER(t) for t in (ctypes.c_uint8, ctypes.c_uint32, ctypes.c_uint64))
        sz = ctypes.c_size_t
        L.orc_bits_for.restype = ctypes.c_uint
        L.orc_bits_for.argtypes = [ctypes.c_uint64]
        for name in ("orc_escape", "orc_unescape"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p]
        L.orc_suffix_array.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p]
        L.orc_isa.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p]
        L.orc_phi.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p]
        L.orc_plcp.restype = ctypes.c_uint32
        L.orc_plcp.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_lcp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, sz, ctypes.c_void_p]
        L.orc_arrays_comp.restype = sz
        L.orc_arrays_comp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, sz, ctypes.c_uint32,
                                      ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.orc_sort_factors.argtypes = [ctypes.c_void_p, sz]
        L.orc_flatten.argtypes = [ctypes.c_void_p, sz, u64p, u64p]
        L.orc_literal_histogram.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, sz, ctypes.c_void_p]
        L.orc_literal_positions.restype = sz
        L.orc_literal_positions.argtypes = [sz, ctypes.c_void_p, sz, ctypes.c_void_p]
        L.orc_huffman_table.argtypes = [ctypes.c_void_p, ctypes.POINTER(HuffTable)]
        L.orc_lcpcomp_huff_compress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint32, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz),
                                                ctypes.POINTER(Stats)]
        L.orc_max_lcp.restype = sz
        L.orc_max_lcp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, sz, ctypes.c_uint32,
                                  ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.orc_lcpcomp_maxlcp_huff_compress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint32, ctypes.c_int,
                                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_lcpcomp_peaks_huff_compress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint32, ctypes.c_int,
                                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_lcpcomp_ascii_compress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint32, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_encode_ascii.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_lcpcomp_ascii_decompress.argtypes = [ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]
        L.orc_lcpcomp_sle_compress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint,
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_encode_sle.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, sz, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_lcpcomp_sle_decompress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]
        L.orc_lcpcomp_arith_compress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint32, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz),
                                                 ctypes.POINTER(Stats)]
        L.orc_encode_arith.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_encode_huff.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_lcpcomp_huff_decompress.argtypes = [ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p),
                                                  ctypes.POINTER(sz)]
        L.orc_lzss_lcp_factorize.restype = sz
        L.orc_lzss_lcp_factorize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, sz, ctypes.c_uint32,
                                             ctypes.POINTER(ctypes.c_void_p)]
        L.orc_lzss_lcp_huff_compress.argtypes = [ctypes.c_void_p, sz, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p),
                                                 ctypes.POINTER(sz), ctypes.POINTER(Stats)]
        L.orc_huff_encode_literals.argtypes = [ctypes.c_void_p, sz, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                               ctypes.POINTER(sz)]
        L.orc_bitstream_script.argtypes = [ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]
        L.orc_bitstream_count_bits.restype = sz
        L.orc_bitstream_count_bits.argtypes = [ctypes.c_void_p, sz]
        L.orc_lz78_gamma_compress.argtypes = [ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]
        L.orc_lz78_factors.restype = sz
        L.orc_lz78_factors.argtypes = [ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_void_p)]
        L.orc_free.argtypes = [ctypes.c_void_p]
    return _lib


def _buf(b):
    """bytes/ndarray -> (keepalive, void*)"""
    a = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _take(ptr, n):
    out = ctypes.string_at(ptr, n) if n else b""
    lib().orc_free(ptr)
    return out


def bits_for(v):
    return lib().orc_bits_for(v)


def escape(data):
    a, p = _buf(data)
    out = np.empty(2 * len(a) + 1, dtype=np.uint8)
    n = lib().orc_escape(p, len(a), out.ctypes.data_as(ctypes.c_void_p))
    return out[:n].tobytes()


def unescape(data):
    a, p = _buf(data)
    out = np.empty(len(a) + 1, dtype=np.uint8)
    n = lib().orc_unescape(p, len(a), out.ctypes.data_as(ctypes.c_void_p))
    return out[:n].tobytes()


def suffix_array(text):
    a, p = _buf(text)
    sa = np.empty(len(a), dtype=np.uint32)
    rc = lib().orc_suffix_array(p, len(a), sa.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError("orc_suffix_array rc=%d" % rc)
    return sa


def isa_phi_plcp(text, sa):
    """returns isa, phi, plcp (plcp[n-1] = 0), maxlcp"""
    a, p = _buf(text)
    n = len(a)
    isa = np.empty(n, dtype=np.uint32)
    phi = np.empty(n, dtype=np.uint32)
    plcp = np.empty(n, dtype=np.uint32)
    L = lib()
    L.orc_isa(sa.ctypes.data_as(ctypes.c_void_p), n, isa.ctypes.data_as(ctypes.c_void_p))
    L.orc_phi(sa.ctypes.data_as(ctypes.c_void_p), n, phi.ctypes.data_as(ctypes.c_void_p))
    m = L.orc_plcp(p, n, phi.ctypes.data_as(ctypes.c_void_p), plcp.ctypes.data_as(ctypes.c_void_p))
    return isa, phi, plcp, int(m)


def lcp_array(sa, plcp):
    lcp = np.empty(len(sa), dtype=np.uint32)
    lib().orc_lcp(sa.ctypes.data_as(ctypes.c_void_p), plcp.ctypes.data_as(ctypes.c_void_p), len(sa),
                  lcp.ctypes.data_as(ctypes.c_void_p))
    return lcp


def arrays_comp(sa, isa, lcp, maxlcp, threshold):
    """ArraysComp factor list in EMISSION order (structured array pos/src/len)."""
    lcp = lcp.copy()
    out = ctypes.c_void_p()
    z = lib().orc_arrays_comp(sa.ctypes.data_as(ctypes.c_void_p), isa.ctypes.data_as(ctypes.c_void_p),
                              lcp.ctypes.data_as(ctypes.c_void_p), len(sa), maxlcp, threshold, ctypes.byref(out))
    raw = _take(out, z * 12) if out.value else b""
    return np.frombuffer(raw, dtype=FACTOR_DTYPE).copy()


def max_lcp(sa, isa, lcp, maxlcp, threshold):
    """MaxLCPStrategy factor list in EMISSION order (compressors/lcpcomp/compress/MaxLCPStrategy.hpp:36-100)."""
    lcp = lcp.copy()
    out = ctypes.c_void_p()
    z = lib().orc_max_lcp(sa.ctypes.data_as(ctypes.c_void_p), isa.ctypes.data_as(ctypes.c_void_p),
                          lcp.ctypes.data_as(ctypes.c_void_p), len(sa), maxlcp, threshold, ctypes.byref(out))
    raw = _take(out, z * 12) if out.value else b""
    return np.frombuffer(raw, dtype=FACTOR_DTYPE).copy()


def max_heap(sa, isa, lcp, threshold):
    """MaxHeapStrategy factor list in EMISSION order (compressors/lcpcomp/compress/MaxHeapStrategy.hpp:36-101)."""
    lcp = lcp.copy()
    out = ctypes.c_void_p()
    L = lib()
    L.orc_max_heap.restype = ctypes.c_size_t
    L.orc_max_heap.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
    z = L.orc_max_heap(sa.ctypes.data_as(ctypes.c_void_p), isa.ctypes.data_as(ctypes.c_void_p), lcp.ctypes.data_as(ctypes.c_void_p),
                       len(sa), threshold, ctypes.byref(out))
    raw = _take(out, z * 12) if out.value else b""
    return np.frombuffer(raw, dtype=FACTOR_DTYPE).copy()


def lcpcomp_heap_huff_compress(text, threshold=5, flatten=1):
    """lcpcomp(coder=huff, comp=heap)"""
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    L = lib()
    L.orc_lcpcomp_heap_huff_compress.argtypes = L.orc_lcpcomp_maxlcp_huff_compress.argtypes
    rc = L.orc_lcpcomp_heap_huff_compress(p, len(a), threshold, flatten, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_lcpcomp_heap_huff_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def sort_factors(f):
    f = f.copy()
    lib().orc_sort_factors(f.ctypes.data_as(ctypes.c_void_p), len(f))
    return f


def flatten(f):
    f = f.copy()
    nf, md = ctypes.c_uint64(), ctypes.c_uint64()
    lib().orc_flatten(f.ctypes.data_as(ctypes.c_void_p), len(f), ctypes.byref(nf), ctypes.byref(md))
    return f, nf.value, md.value


def literal_histogram(text, f):
    a, p = _buf(text)
    C = np.zeros(256, dtype=np.uint32)
    f = np.ascontiguousarray(f)
    lib().orc_literal_histogram(p, len(a), f.ctypes.data_as(ctypes.c_void_p), len(f), C.ctypes.data_as(ctypes.c_void_p))
    return C


def literal_positions(n, f):
    f = np.ascontiguousarray(f)
    pos = np.empty(n, dtype=np.uint32)
    k = lib().orc_literal_positions(n, f.ctypes.data_as(ctypes.c_void_p), len(f), pos.ctypes.data_as(ctypes.c_void_p))
    return pos[:k].copy()


def huffman_table(C):
    C = np.ascontiguousarray(C, dtype=np.uint32)
    t = HuffTable()
    lib().orc_huffman_table(C.ctypes.data_as(ctypes.c_void_p), ctypes.byref(t))
    return t


def lcpcomp_huff_compress(text, threshold=5, flatten=1):
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_lcpcomp_huff_compress(p, len(a), threshold, flatten, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_lcpcomp_huff_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lcpcomp_arith_compress(text, threshold=5, flatten=1):
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_lcpcomp_arith_compress(p, len(a), threshold, flatten, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        if out.value:
            lib().orc_free(out)
        raise RuntimeError("orc_lcpcomp_arith_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lcpcomp_maxlcp_huff_compress(text, threshold=5, flatten=1):
    """lcpcomp(coder=huff, comp=max_lcp)"""
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_lcpcomp_maxlcp_huff_compress(p, len(a), threshold, flatten, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_lcpcomp_maxlcp_huff_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lcpcomp_peaks_huff_compress(text, threshold=5, flatten=1):
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_lcpcomp_peaks_huff_compress(p, len(a), threshold, flatten, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_lcpcomp_peaks_huff_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lcpcomp_ascii_compress(text, threshold=5, flatten=1):
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_lcpcomp_ascii_compress(p, len(a), threshold, flatten, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_lcpcomp_ascii_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def encode_ascii(text, f):
    a, p = _buf(text)
    f = np.ascontiguousarray(f)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_encode_ascii(p, len(a), f.ctypes.data_as(ctypes.c_void_p), len(f), ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_encode_ascii rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lcpcomp_ascii_decompress(stream):
    a, p = _buf(stream)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    rc = lib().orc_lcpcomp_ascii_decompress(p, len(a), ctypes.byref(out), ctypes.byref(n))
    if rc:
        raise RuntimeError("orc_lcpcomp_ascii_decompress rc=%d" % rc)
    return _take(out, n.value)


def lcpcomp_sle_compress(text, threshold=5, flatten=1, kmer=3):
    """lcpcomp(coder=sle(kmer)) -- coders/SLECoder.hpp"""
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_lcpcomp_sle_compress(p, len(a), threshold, flatten, kmer, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_lcpcomp_sle_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def encode_sle(text, f, kmer=3):
    a, p = _buf(text)
    f = np.ascontiguousarray(f)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_encode_sle(p, len(a), f.ctypes.data_as(ctypes.c_void_p), len(f), kmer, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_encode_sle rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lcpcomp_sle_decompress(stream, kmer=3):
    a, p = _buf(stream)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    rc = lib().orc_lcpcomp_sle_decompress(p, len(a), kmer, ctypes.byref(out), ctypes.byref(n))
    if rc:
        raise RuntimeError("orc_lcpcomp_sle_decompress rc=%d" % rc)
    return _take(out, n.value)


def encode_arith(text, f):
    a, p = _buf(text)
    f = np.ascontiguousarray(f)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_encode_arith(p, len(a), f.ctypes.data_as(ctypes.c_void_p), len(f), ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        if out.value:
            lib().orc_free(out)
        raise RuntimeError("orc_encode_arith rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def encode_huff(text, f):
    a, p = _buf(text)
    f = np.ascontiguousarray(f)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_encode_huff(p, len(a), f.ctypes.data_as(ctypes.c_void_p), len(f), ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_encode_huff rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lzss_lcp_factorize(sa, isa, lcp, threshold):
    out = ctypes.c_void_p()
    z = lib().orc_lzss_lcp_factorize(sa.ctypes.data_as(ctypes.c_void_p), isa.ctypes.data_as(ctypes.c_void_p),
                                     lcp.ctypes.data_as(ctypes.c_void_p), len(sa), threshold, ctypes.byref(out))
    raw = _take(out, z * 12) if out.value else b""
    return np.frombuffer(raw, dtype=FACTOR_DTYPE).copy()


def lzss_lcp_huff_compress(text, threshold=3):
    a, p = _buf(text)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
    rc = lib().orc_lzss_lcp_huff_compress(p, len(a), threshold, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError("orc_lzss_lcp_huff_compress rc=%d" % rc)
    return _take(out, n.value), st.as_dict()


def lcpcomp_huff_decompress(data):
    a, p = _buf(data)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    rc = lib().orc_lcpcomp_huff_decompress(p, len(a), ctypes.byref(out), ctypes.byref(n))
    if rc:
        raise RuntimeError("orc_lcpcomp_huff_decompress rc=%d" % rc)
    return _take(out, n.value)


def huff_encode_literals(lits, interleave=False):
    a, p = _buf(lits)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    lib().orc_huff_encode_literals(p, len(a), int(interleave), ctypes.byref(out), ctypes.byref(n))
    return _take(out, n.value)


def bitstream_script(ops):
    """ops: list of (kind, value, bits); kind 0 bit, 1 int, 2 compressed int"""
    arr = np.array([[k, v & 0xFFFFFFFFFFFFFFFF, b] for k, v, b in ops], dtype=np.uint64).reshape(-1)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    lib().orc_bitstream_script(arr.ctypes.data_as(ctypes.c_void_p), len(ops), ctypes.byref(out), ctypes.byref(n))
    return _take(out, n.value)


def bitstream_count_bits(data):
    a, p = _buf(data)
    return lib().orc_bitstream_count_bits(p, len(a))


def lz78_gamma_compress(data):
    a, p = _buf(data)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    lib().orc_lz78_gamma_compress(p, len(a), ctypes.byref(out), ctypes.byref(n))
    return _take(out, n.value)


def lz78_factors(data):
    a, p = _buf(data)
    ids, ch = ctypes.c_void_p(), ctypes.c_void_p()
    z = lib().orc_lz78_factors(p, len(a), ctypes.byref(ids), ctypes.byref(ch))
    i = np.frombuffer(_take(ids, z * 4), dtype=np.uint32).copy()
    c = _take(ch, z)
    return i, c
