"""tudocomp_amd -- MI355X-native lcpcomp hot path behind tudocomp's Compressor surface.

Python face of the C ABI (include/tdc_gpu.h) used by the parity tests and bench.py.  The C++ facade that mirrors
tdc::Compressor / the `tdc` command line lives in tudocomp_amd/host/.  Nothing here computes on the CPU: every
method forwards to the HIP library and raises if it (or a GPU) is unavailable.
"""
import ctypes

import numpy as np

from . import _native
from ._native import Stats, LIB_PATH, SYMBOLS  # noqa: F401

CODER_HUFF = 0
CODER_GAMMA = 1
CODER_ARITH = 2
CODER_ASCII = 3
CODER_SLE = 4            # coder=sle(kmer=k): CODER_SLE | (k << 8), k = 0 means the reference's default 3
COMP_ARRAYS = 0
COMP_PLCPPEAKS = 1
COMP_MAXLCP = 2
COMP_HEAP = 3


class TdcGpuError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = _native.load().tdc_gpu_strerror(status).decode()
        super().__init__("%s (status %d)%s" % (msg, status, (": " + detail) if detail else ""))


def _u8(b):
    a = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray, memoryview)) else np.ascontiguousarray(b, dtype=np.uint8)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def escape(data):
    """Input restrictions 'escape {0} + null-terminate' (io/RestrictedBuffer.hpp:43-74): what Input::as_view() yields."""
    L = _native.load()
    a = _u8(data)
    out = np.empty(2 * len(a) + 1, dtype=np.uint8)
    n = L.tdc_escape(_ptr(a), len(a), _ptr(out))
    return out[:n].tobytes()


def unescape(data):
    L = _native.load()
    a = _u8(data)
    out = np.empty(len(a) + 1, dtype=np.uint8)
    n = L.tdc_unescape(_ptr(a), len(a), _ptr(out))
    return out[:n].tobytes()


def _gen_target(n, out):
    """the array a native generator fills: a fresh one, or the caller's -- which must really hold n contiguous bytes"""
    if out is None:
        return np.empty(n, dtype=np.uint8)
    if not isinstance(out, np.ndarray) or out.dtype != np.uint8 or out.ndim != 1 or not out.flags.c_contiguous or out.size < n:
        raise ValueError("out must be a contiguous one-dimensional uint8 array of at least n bytes")
    return out


def gen_english(n, seed=42, out=None):
    """SURVEY.md 8d English-like generator; `out` (optional): a uint8 array of >= n bytes to fill in place."""
    out = _gen_target(n, out)
    _native.load().tdc_gen_english(_ptr(out), n, seed)
    return out[:n]


def gen_dna(n, seed=7, out=None):
    """SURVEY.md 8d DNA generator; `out` as in gen_english."""
    out = _gen_target(n, out)
    _native.load().tdc_gen_dna(_ptr(out), n, seed)
    return out[:n]


def lz78_factors(data):
    """The LZ78 parse on its own (host; compressors/LZ78Compressor.hpp:97-131): (ids, chars) as numpy arrays."""
    L = _native.load()
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    ids, ch, z = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
    rc = L.tdc_lz78_factors(_ptr(a) if len(a) else None, len(a), ctypes.byref(ids), ctypes.byref(ch), ctypes.byref(z))
    if rc:
        raise TdcGpuError(rc)
    try:
        i = np.ctypeslib.as_array(ctypes.cast(ids, ctypes.POINTER(ctypes.c_uint32)), (max(z.value, 1),))[:z.value].copy()
        c = np.ctypeslib.as_array(ctypes.cast(ch, ctypes.POINTER(ctypes.c_uint8)), (max(z.value, 1),))[:z.value].copy()
    finally:
        L.tdc_gpu_free(ids); L.tdc_gpu_free(ch)
    return i, c


def option_names():
    """Names of the library's options (tdc_gpu_ctx_set_option)."""
    L = _native.load()
    return [L.tdc_gpu_option_name(i).decode() for i in range(L.tdc_gpu_option_count())]


def huffman_table(counts):
    L = _native.load()
    C = np.ascontiguousarray(counts, dtype=np.uint32)
    sigma, longest = ctypes.c_uint32(), ctypes.c_uint32()
    order = np.zeros(256, dtype=np.uint8)
    len_of = np.zeros(256, dtype=np.uint8)
    code_of = np.zeros(256, dtype=np.uint64)
    rc = L.tdc_huffman_table(_ptr(C), ctypes.byref(sigma), ctypes.byref(longest), _ptr(order), _ptr(len_of), _ptr(code_of))
    if rc:
        raise TdcGpuError(rc)
    return {"sigma": sigma.value, "longest": longest.value, "order": order, "len_of": len_of, "code_of": code_of}


def device_count():
    return _native.load().tdc_gpu_device_count()


def blocks_compress(data, block_size, threshold=5, flatten=1, coder=CODER_HUFF, devices=None):
    """Block mode (tdc_gpu_blocks_compress): `data` (unrestricted bytes) is cut into blocks of block_size bytes, the blocks are
    spread over `devices` (default: all visible ones), every block becomes a complete lcpcomp stream.  Returns (container bytes,
    list of per-block stats dicts)."""
    L = _native.load()
    a = _u8(data)
    if devices is None:
        devices = list(range(max(1, device_count())))
    devs = (ctypes.c_int * len(devices))(*devices)
    G = L.tdc_gpu_blocks_count(len(a), block_size)
    st = (Stats * max(G, 1))()
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    rc = L.tdc_gpu_blocks_compress(devs, len(devices), _ptr(a), len(a), block_size, threshold, int(flatten), coder,
                                   ctypes.byref(out), ctypes.byref(n), st)
    if rc:
        raise TdcGpuError(rc, "tdc_gpu_blocks_compress")
    blob = ctypes.string_at(out, n.value)
    L.tdc_gpu_free(out)
    return blob, [st[i].as_dict() for i in range(G)]


class PinnedBuffer:
    """Page-locked host memory (tdc_gpu_host_alloc) as a numpy uint8 array `.a`; the buffers of the end-to-end entry point."""

    def __init__(self, nbytes):
        self._L = _native.load()
        self.nbytes = int(nbytes)
        self.ptr = self._L.tdc_gpu_host_alloc(self.nbytes)
        if not self.ptr:
            raise TdcGpuError(-5, "tdc_gpu_host_alloc(%d)" % self.nbytes)
        self.a = np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(max(self.nbytes, 1),))[:self.nbytes]

    def free(self):
        if getattr(self, "ptr", None):
            self.a = None
            self._L.tdc_gpu_host_free(self.ptr)
            self.ptr = None

    __del__ = free


def host_register(a):
    """page-lock the memory of a numpy array the caller allocated itself (e.g. a memory-mapped shared segment): True on success"""
    if not (isinstance(a, np.ndarray) and a.flags.c_contiguous):
        raise ValueError("host_register: a C-contiguous numpy array is needed")
    return _native.load().tdc_gpu_host_register(ctypes.c_void_p(a.ctypes.data), a.nbytes) == 0


def host_unregister(a):
    return _native.load().tdc_gpu_host_unregister(ctypes.c_void_p(a.ctypes.data)) == 0


class Context:
    """One GPU, one HIP stream, one device arena (tdc_gpu_ctx)."""

    def __init__(self, device=0, options=None):
        """options: {name: value} applied through tdc_gpu_ctx_set_option ("wsort_min" or "TDC_GPU_WSORT_MIN": the same option) --
        the way tests and A/B runs reach the library's switches; the environment is not read (include/tdc_gpu.h)."""
        self._L = _native.load()
        h = ctypes.c_void_p()
        rc = self._L.tdc_gpu_ctx_create(device, ctypes.byref(h))
        if rc:
            raise TdcGpuError(rc, "tdc_gpu_ctx_create(device=%d)" % device)
        self._h = h
        self.device = device
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name, value):
        rc = self._L.tdc_gpu_ctx_set_option(self._h, str(name).encode(), int(value))
        if rc:
            raise TdcGpuError(rc, "unknown option %r" % (name,))

    def close(self):
        if getattr(self, "_h", None):
            self._L.tdc_gpu_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc:
            raise TdcGpuError(rc, self._L.tdc_gpu_last_error(self._h).decode())

    def _take(self, ptr, nbytes):
        out = ctypes.string_at(ptr, nbytes) if nbytes else b""
        self._L.tdc_gpu_free(ptr)
        return out

    def set_profiling(self, enabled=True):
        self._check(self._L.tdc_gpu_ctx_set_profiling(self._h, int(enabled)))

    def reset_profile(self):
        self._L.tdc_gpu_ctx_reset_profile(self._h)

    def kernel_profile(self):
        """{kernel name: {"ms", "launches", "bytes"}} since the last reset (only while profiling is enabled)."""
        out, i = {}, 0
        while True:
            ms, ln, by = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
            name = self._L.tdc_gpu_ctx_kernel_profile(self._h, i, ctypes.byref(ms), ctypes.byref(ln), ctypes.byref(by))
            if name is None:
                return out
            out[name.decode()] = {"ms": ms.value, "launches": ln.value, "bytes": by.value}
            i += 1

    def reserve(self, n):
        self._check(self._L.tdc_gpu_ctx_reserve(self._h, n))

    # ---- hot path --------------------------------------------------------------------------------------
    def lcpcomp_compress(self, text, threshold=5, flatten=1, coder=CODER_HUFF, comp=COMP_ARRAYS):
        """text: escaped + 0-terminated view.  Returns (compressed bytes, stats dict)."""
        a = _u8(text)
        out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
        if comp == COMP_ARRAYS:
            rc = self._L.tdc_gpu_lcpcomp_compress(self._h, _ptr(a), len(a), threshold, int(flatten), coder,
                                                  ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
        else:
            rc = self._L.tdc_gpu_lcpcomp_compress_comp(self._h, _ptr(a), len(a), threshold, int(flatten), coder, comp,
                                                       ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
        self._check(rc)
        return self._take(out, n.value), st.as_dict()

    def lcpcomp_compress_into(self, text, n, out, threshold=5, flatten=1, coder=CODER_HUFF, comp=COMP_ARRAYS):
        """End-to-end entry point: text (n bytes incl. sentinel) and out are host buffers (PinnedBuffer or numpy uint8 arrays);
        the stream is written into out.  Returns (out_len, stats)."""
        ta = text.a if isinstance(text, PinnedBuffer) else _u8(text)
        oa = out.a if isinstance(out, PinnedBuffer) else out
        ol, st = ctypes.c_size_t(), Stats()
        self._check(self._L.tdc_gpu_lcpcomp_compress_into(self._h, _ptr(ta), n, threshold, int(flatten), coder, comp, _ptr(oa), len(oa),
                                                          ctypes.byref(ol), ctypes.byref(st)))
        return ol.value, st.as_dict()

    def lcpcomp_compress_keep(self, text, n, threshold=5, flatten=1, coder=CODER_HUFF, comp=COMP_ARRAYS):
        """lcpcomp_compress_into without the download: the stream stays on the device until the next call on this context
        (stream_fetch copies it out).  Returns (out_len, stats)."""
        ta = text.a if isinstance(text, PinnedBuffer) else _u8(text)
        ol, st = ctypes.c_size_t(), Stats()
        self._check(self._L.tdc_gpu_lcpcomp_compress_keep(self._h, _ptr(ta), n, threshold, int(flatten), coder, comp, ctypes.byref(ol), ctypes.byref(st)))
        return ol.value, st.as_dict()

    def stream_fetch(self, out):
        """copy the stream kept by lcpcomp_compress_keep into the host buffer `out` (a writable, contiguous numpy uint8 array or a
        PinnedBuffer -- e.g. this rank's slice of a container in shared memory).  Returns the stream length."""
        oa = out.a if isinstance(out, PinnedBuffer) else out
        if not (isinstance(oa, np.ndarray) and oa.dtype == np.uint8 and oa.flags.c_contiguous and oa.flags.writeable):
            raise ValueError("stream_fetch: `out` must be a writable, C-contiguous uint8 array")
        ln = ctypes.c_size_t()
        self._check(self._L.tdc_gpu_stream_fetch(self._h, _ptr(oa), oa.size, ctypes.byref(ln)))
        return ln.value

    def stream_fetch_dev(self, d_dst, cap):
        """copy the kept stream to device memory of this context's GPU (`d_dst`: raw device pointer, `cap` bytes).  Returns its length."""
        ln = ctypes.c_size_t()
        self._check(self._L.tdc_gpu_stream_fetch_dev(self._h, ctypes.c_void_p(d_dst), cap, ctypes.byref(ln)))
        return ln.value

    def lcpcomp_compress_raw(self, data, threshold=5, flatten=1, coder=CODER_HUFF):
        """data: unrestricted input; escaping + sentinel happen on the device.  Returns (compressed bytes, stats dict)."""
        a = _u8(data)
        out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
        self._check(self._L.tdc_gpu_lcpcomp_compress_raw(self._h, _ptr(a), len(a), threshold, int(flatten), coder,
                                                         ctypes.byref(out), ctypes.byref(n), ctypes.byref(st)))
        return self._take(out, n.value), st.as_dict()

    def lcpcomp_compress_dev(self, d_text, n, d_out, out_cap, threshold=5, flatten=1, coder=CODER_HUFF):
        """Device-resident variant: d_text / d_out are raw device pointers (ints).  Returns (out_len, stats)."""
        ol, st = ctypes.c_size_t(), Stats()
        self._check(self._L.tdc_gpu_lcpcomp_compress_dev(self._h, ctypes.c_void_p(d_text), n, threshold, int(flatten),
                                                         coder, ctypes.c_void_p(d_out), out_cap, ctypes.byref(ol),
                                                         ctypes.byref(st)))
        return ol.value, st.as_dict()

    def lzss_lcp_compress(self, text, threshold=3, coder=CODER_HUFF):
        """LZSSLCPCompressor<HuffmanCoder>::compress on an escaped + 0-terminated view.  Returns (stream, stats)."""
        a = _u8(text)
        out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
        self._check(self._L.tdc_gpu_lzss_lcp_compress(self._h, _ptr(a), len(a), threshold, coder, ctypes.byref(out),
                                                      ctypes.byref(n), ctypes.byref(st)))
        return self._take(out, n.value), st.as_dict()

    def lzss_lcp_factorize(self, text, threshold=3):
        a = _u8(text)
        p, s, l, z = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._L.tdc_gpu_lzss_lcp_factorize(self._h, _ptr(a), len(a), threshold, ctypes.byref(p), ctypes.byref(s),
                                                       ctypes.byref(l), ctypes.byref(z)))
        out = [np.frombuffer(self._take(x, z.value * 4), dtype=np.uint32).copy() for x in (p, s, l)]
        return out[0], out[1], out[2]

    def lz78_compress(self, data, coder=CODER_GAMMA):
        """LZ78Compressor<EliasGammaCoder>::compress on raw bytes (no escaping).  Returns (stream, stats)."""
        a = _u8(data)
        out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), Stats()
        self._check(self._L.tdc_gpu_lz78_compress(self._h, _ptr(a), len(a), coder, ctypes.byref(out), ctypes.byref(n),
                                                  ctypes.byref(st)))
        return self._take(out, n.value), st.as_dict()

    def bound(self, n, coder=None):
        return self._L.tdc_gpu_lcpcomp_bound(n) if coder is None else self._L.tdc_gpu_lcpcomp_bound_coder(n, coder)

    # ---- stages ----------------------------------------------------------------------------------------
    def sort_pairs_u64(self, keys, vals, algo=1):
        """The device sorts behind the suffix array (algo 0: LSD radix, 1: splitter partition); returns sorted copies."""
        k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self._check(self._L.tdc_gpu_sort_pairs_u64(self._h, _ptr(k), _ptr(v), len(k), int(algo)))
        return k, v

    def suffix_array(self, text):
        a = _u8(text)
        sa = np.empty(len(a), dtype=np.uint32)
        isa = np.empty(len(a), dtype=np.uint32)
        self._check(self._L.tdc_gpu_suffix_array(self._h, _ptr(a), len(a), _ptr(sa), _ptr(isa)))
        return sa, isa

    def textds(self, text):
        a = _u8(text)
        n = len(a)
        arrs = {k: np.empty(n, dtype=np.uint32) for k in ("sa", "isa", "phi", "plcp", "lcp")}
        m = ctypes.c_uint32()
        self._check(self._L.tdc_gpu_textds(self._h, _ptr(a), n, _ptr(arrs["sa"]), _ptr(arrs["isa"]), _ptr(arrs["phi"]),
                                           _ptr(arrs["plcp"]), _ptr(arrs["lcp"]), ctypes.byref(m)))
        arrs["maxlcp"] = m.value
        return arrs

    def lcpcomp_decompress(self, stream, coder=CODER_HUFF):
        """LCPCompressor::decompress on a lcpcomp / lzss_lcp stream written with coder huff, ascii or sle (CODER_SLE | kmer << 8):
        returns the escaped, 0-terminated text and {"factors", "rounds"}."""
        a = _u8(stream)
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        f, r = ctypes.c_uint64(), ctypes.c_uint32()
        self._check(self._L.tdc_gpu_lcpcomp_decompress_coder(self._h, _ptr(a), len(a), coder, ctypes.byref(p), ctypes.byref(n),
                                                             ctypes.byref(f), ctypes.byref(r)))
        return self._take(p, n.value), {"factors": f.value, "rounds": r.value,
                                        "device_parse": int(self._L.tdc_gpu_ctx_last_decode_on_device(self._h))}

    def lcpcomp_decompress_into(self, stream, out, coder=CODER_HUFF):
        """lcpcomp_decompress into a caller-owned buffer (a PinnedBuffer or a writable uint8 array; `stream` may be a PinnedBuffer too):
        returns (text length, {"factors", "rounds", "device_parse"})."""
        a = stream.a if isinstance(stream, PinnedBuffer) else _u8(stream)
        oa = out.a if isinstance(out, PinnedBuffer) else out
        n = ctypes.c_size_t()
        f, r = ctypes.c_uint64(), ctypes.c_uint32()
        self._check(self._L.tdc_gpu_lcpcomp_decompress_into(self._h, _ptr(a), len(a), coder, _ptr(oa), oa.size, ctypes.byref(n),
                                                            ctypes.byref(f), ctypes.byref(r)))
        return n.value, {"factors": f.value, "rounds": r.value, "device_parse": int(self._L.tdc_gpu_ctx_last_decode_on_device(self._h))}

    def blocks_decompress(self, blob, coder=CODER_HUFF):
        """inverse of blocks_compress on this context's device: the concatenated raw bytes"""
        a = _u8(blob)
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._L.tdc_gpu_blocks_decompress(self._h, _ptr(a), len(a), coder, ctypes.byref(p), ctypes.byref(n)))
        return self._take(p, n.value)

    def factorize(self, text, threshold=5, flatten=0):
        """Returns (pos, src, len) sorted by pos and the stats dict."""
        a = _u8(text)
        p, s, l = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        z, st = ctypes.c_size_t(), Stats()
        self._check(self._L.tdc_gpu_lcpcomp_factorize(self._h, _ptr(a), len(a), threshold, int(flatten), ctypes.byref(p),
                                                      ctypes.byref(s), ctypes.byref(l), ctypes.byref(z), ctypes.byref(st)))
        out = [np.frombuffer(self._take(x, z.value * 4), dtype=np.uint32).copy() for x in (p, s, l)]
        return out[0], out[1], out[2], st.as_dict()

    def flatten(self, n, pos, src, length):
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        src = np.ascontiguousarray(src, dtype=np.uint32).copy()
        length = np.ascontiguousarray(length, dtype=np.uint32)
        nf, md = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._L.tdc_gpu_flatten(self._h, n, _ptr(pos), _ptr(src), _ptr(length), len(pos), ctypes.byref(nf),
                                            ctypes.byref(md)))
        return src, nf.value, md.value

    def encode_sle(self, text, pos, src, length, kmer=3):
        """SLECoder::Encoder (coders/SLECoder.hpp:42-298) + lzss::encode_text on a given factor list"""
        a = _u8(text)
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        src = np.ascontiguousarray(src, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._L.tdc_gpu_encode_sle(self._h, _ptr(a), len(a), _ptr(pos), _ptr(src), _ptr(length), len(pos), int(kmer),
                                               ctypes.byref(out), ctypes.byref(n)))
        return self._take(out, n.value)

    def encode_ascii(self, text, pos, src, length):
        return self.encode_huff(text, pos, src, length, _fn="tdc_gpu_encode_ascii")

    def encode_arith(self, text, pos, src, length):
        return self.encode_huff(text, pos, src, length, _fn="tdc_gpu_encode_arith")

    def encode_huff(self, text, pos, src, length, _fn="tdc_gpu_encode_huff"):
        a = _u8(text)
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        src = np.ascontiguousarray(src, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(getattr(self._L, _fn)(self._h, _ptr(a), len(a), _ptr(pos), _ptr(src), _ptr(length), len(pos),
                                                ctypes.byref(out), ctypes.byref(n)))
        return self._take(out, n.value)


class LCPCompressor:
    """Mirror of tdc::LCPCompressor<coder, ArraysComp, ...> (compressors/LCPCompressor.hpp:79-151) as the
    reference's test harness drives it (test/test/util.hpp:442-463): the input is wrapped with the compressor's
    input restrictions (escape {0}, null-terminate) and handed to compress()."""

    def __init__(self, ctx, coder="huff", threshold=5, flatten=1, comp="arrays", kmer=3):
        if coder not in ("huff", "arithmetic", "ascii", "sle") or comp not in ("arrays", "plcppeaks", "max_lcp", "heap"):
            # same wording as Registry.hpp:214
            raise RuntimeError("No implementation found for compressor lcpcomp(coder=%s,comp=%s)" % (coder, comp))
        self.ctx, self.threshold, self.flatten = ctx, int(threshold), int(flatten)
        self.coder = {"huff": CODER_HUFF, "arithmetic": CODER_ARITH, "ascii": CODER_ASCII, "sle": CODER_SLE | (int(kmer) << 8)}[coder]
        self.comp = {"arrays": COMP_ARRAYS, "plcppeaks": COMP_PLCPPEAKS, "max_lcp": COMP_MAXLCP, "heap": COMP_HEAP}[comp]
        self.last_stats = None

    def compress(self, data):
        out, st = self.ctx.lcpcomp_compress(escape(data), self.threshold, self.flatten, self.coder, self.comp)
        self.last_stats = st
        return out

    def decompress(self, stream):
        """LCPCompressor::decompress (coder huff / ascii / sle): references resolved on the device; the harness then removes
        the input restrictions again (unescape, drop the sentinel)."""
        if self.coder == CODER_ARITH:
            raise RuntimeError("lcpcomp(coder=arithmetic) streams cannot be decoded (neither can the reference)")
        text, _ = self.ctx.lcpcomp_decompress(stream, self.coder)
        return unescape(text)


class LZ78Compressor:
    """Mirror of tdc::LZ78Compressor<coder, trie> (compressors/LZ78Compressor.hpp:45-161); no input restrictions."""

    def __init__(self, ctx, coder="gamma", lz78trie="ternary"):
        if coder != "gamma":
            raise RuntimeError("No implementation found for compressor lz78(coder=%s,lz78trie=%s)" % (coder, lz78trie))
        self.ctx = ctx
        self.last_stats = None

    def compress(self, data):
        out, st = self.ctx.lz78_compress(data)
        self.last_stats = st
        return out


class LZSSLCPCompressor:
    """Mirror of tdc::LZSSLCPCompressor<coder> (compressors/LZSSLCPCompressor.hpp:22-132): threshold defaults to 3."""

    def __init__(self, ctx, coder="huff", threshold=3):
        if coder != "huff":
            raise RuntimeError("No implementation found for compressor lzss_lcp(coder=%s)" % coder)
        self.ctx, self.threshold = ctx, int(threshold)
        self.last_stats = None

    def compress(self, data):
        out, st = self.ctx.lzss_lcp_compress(escape(data), self.threshold)
        self.last_stats = st
        return out
