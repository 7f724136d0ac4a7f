"""ctypes binding of libtdc_gpu.so (C ABI: include/tdc_gpu.h).

There is no CPU fallback anywhere in this package: if the HIP library is missing or no GPU is usable, the
calls raise.  The oracle under oracle/ is test infrastructure and is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TDC_GPU_LIB: development aid for A/B runs of kernel variants (tools/build_variant.sh); the product is always lib/libtdc_gpu.so
LIB_PATH = os.environ.get("TDC_GPU_LIB") or os.path.join(_HERE, "lib", "libtdc_gpu.so")

# every symbol include/tdc_gpu.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "tdc_gpu_ctx_create", "tdc_gpu_ctx_destroy", "tdc_gpu_ctx_reserve", "tdc_gpu_strerror", "tdc_gpu_last_error",
    "tdc_gpu_ctx_set_profiling", "tdc_gpu_ctx_reset_profile", "tdc_gpu_ctx_kernel_profile",
    "tdc_gpu_free", "tdc_gpu_lcpcomp_compress", "tdc_gpu_lcpcomp_compress_raw", "tdc_gpu_lcpcomp_compress_dev",
    "tdc_gpu_lcpcomp_bound", "tdc_gpu_lcpcomp_bound_coder", "tdc_gpu_lcpcomp_compress_into", "tdc_gpu_host_alloc", "tdc_gpu_host_free",
    "tdc_gpu_lz78_compress", "tdc_gpu_lzss_lcp_compress", "tdc_gpu_lzss_lcp_factorize",
    "tdc_gpu_sort_pairs_u64", "tdc_gpu_suffix_array", "tdc_gpu_textds", "tdc_gpu_lcpcomp_factorize", "tdc_gpu_flatten", "tdc_gpu_encode_huff",
    "tdc_gpu_lcpcomp_decompress",
    "tdc_gpu_lcpcomp_compress_comp",
    "tdc_gpu_encode_arith",
    "tdc_gpu_encode_ascii",
    "tdc_gpu_encode_sle",
    "tdc_gpu_lcpcomp_decompress_coder", "tdc_gpu_ctx_last_decode_on_device", "tdc_gpu_lcpcomp_decompress_into",
    "tdc_gpu_ctx_set_option", "tdc_gpu_option_count", "tdc_gpu_option_name", "tdc_escape", "tdc_unescape", "tdc_lz78_factors", "tdc_huffman_table", "tdc_huffman_selfcheck", "tdc_gpu_blocks_count", "tdc_gpu_blocks_compress", "tdc_gpu_blocks_decompress", "tdc_gpu_device_count", "tdc_gen_english", "tdc_gen_dna",
    "tdc_gpu_arena_bytes", "tdc_gpu_device_memory",
    "tdc_gpu_lcpcomp_compress_keep", "tdc_gpu_stream_fetch", "tdc_gpu_stream_fetch_dev", "tdc_gpu_host_register", "tdc_gpu_host_unregister",
]


class Stats(ctypes.Structure):
    _fields_ = (
        [(k, ctypes.c_uint64) for k in ("n", "out_len", "factors", "maxlcp", "entries", "num_flattened", "max_depth_lb",
                                        "flen_min", "flen_max", "fdist_max", "pushes")] +
        [(k, ctypes.c_uint32) for k in ("sa_rounds", "sa_init_syms", "levels", "mis_rounds", "flatten_rounds", "sigma")] +
        [(k, ctypes.c_uint64) for k in ("sa_sorted_elems", "arena_bytes")] +
        [(k, ctypes.c_float) for k in ("ms_h2d", "ms_sa", "ms_phi", "ms_plcp", "ms_factorize", "ms_flatten", "ms_encode",
                                       "ms_d2h", "ms_total")] +
        [(k, ctypes.c_uint32) for k in ("small_levels", "purges", "window_pass", "window_lcut",
                                        "sa_key_words", "sa_text_rounds", "sa_mode", "sa_overlapped", "eager_levels", "eager_phases", "sa_star_chains")])

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load():
    """Load libtdc_gpu.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("tudocomp_amd: %s is missing -- build it with `make -C tudocomp_amd/csrc` "
                           "(there is no CPU fallback)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u32, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
    pvp, psz = ctypes.POINTER(vp), ctypes.POINTER(sz)
    L.tdc_gpu_ctx_create.argtypes = [i32, pvp]
    L.tdc_gpu_ctx_destroy.argtypes = [vp]
    L.tdc_gpu_ctx_destroy.restype = None
    L.tdc_gpu_ctx_reserve.argtypes = [vp, sz]
    L.tdc_gpu_ctx_set_profiling.argtypes = [vp, i32]
    L.tdc_gpu_ctx_reset_profile.argtypes = [vp]
    L.tdc_gpu_ctx_reset_profile.restype = None
    L.tdc_gpu_ctx_kernel_profile.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                                             ctypes.POINTER(ctypes.c_uint64)]
    L.tdc_gpu_ctx_kernel_profile.restype = ctypes.c_char_p
    L.tdc_gpu_strerror.argtypes = [i32]
    L.tdc_gpu_strerror.restype = ctypes.c_char_p
    L.tdc_gpu_last_error.argtypes = [vp]
    L.tdc_gpu_last_error.restype = ctypes.c_char_p
    L.tdc_gpu_free.argtypes = [vp]
    L.tdc_gpu_free.restype = None
    L.tdc_gpu_lcpcomp_compress.argtypes = [vp, vp, sz, u32, i32, i32, pvp, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_lcpcomp_compress_comp.argtypes = [vp, vp, sz, u32, i32, i32, i32, pvp, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_lcpcomp_compress_raw.argtypes = [vp, vp, sz, u32, i32, i32, pvp, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_lcpcomp_compress_dev.argtypes = [vp, vp, sz, u32, i32, i32, vp, sz, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_lz78_compress.argtypes = [vp, vp, sz, i32, pvp, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_lzss_lcp_compress.argtypes = [vp, vp, sz, u32, i32, pvp, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_lzss_lcp_factorize.argtypes = [vp, vp, sz, u32, pvp, pvp, pvp, psz]
    L.tdc_gpu_arena_bytes.argtypes = [sz]
    L.tdc_gpu_arena_bytes.restype = sz
    L.tdc_gpu_device_memory.argtypes = [i32, psz, psz]
    L.tdc_gpu_lcpcomp_bound.argtypes = [sz]
    L.tdc_gpu_lcpcomp_bound.restype = sz
    L.tdc_gpu_lcpcomp_bound_coder.argtypes = [sz, i32]
    L.tdc_gpu_lcpcomp_bound_coder.restype = sz
    L.tdc_gpu_lcpcomp_compress_into.argtypes = [vp, vp, sz, u32, i32, i32, i32, vp, sz, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_lcpcomp_compress_keep.argtypes = [vp, vp, sz, u32, i32, i32, i32, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_stream_fetch.argtypes = [vp, vp, sz, psz]
    L.tdc_gpu_stream_fetch_dev.argtypes = [vp, vp, sz, psz]
    L.tdc_gpu_ctx_last_decode_on_device.argtypes = [vp]
    L.tdc_gpu_lcpcomp_decompress_into.argtypes = [vp, vp, sz, i32, vp, sz, psz, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    L.tdc_gpu_host_register.argtypes = [vp, sz]
    L.tdc_gpu_host_unregister.argtypes = [vp]
    L.tdc_gpu_blocks_count.argtypes = [sz, sz]
    L.tdc_gpu_blocks_count.restype = sz
    L.tdc_gpu_blocks_compress.argtypes = [ctypes.POINTER(i32), i32, vp, sz, sz, u32, i32, i32, pvp, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_blocks_decompress.argtypes = [vp, vp, sz, i32, pvp, psz]
    L.tdc_gpu_host_alloc.argtypes = [sz]
    L.tdc_gpu_host_alloc.restype = vp
    L.tdc_gpu_host_free.argtypes = [vp]
    L.tdc_gpu_host_free.restype = None
    L.tdc_gpu_sort_pairs_u64.argtypes = [vp, vp, vp, sz, i32]
    L.tdc_gpu_suffix_array.argtypes = [vp, vp, sz, vp, vp]
    L.tdc_gpu_textds.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, ctypes.POINTER(u32)]
    L.tdc_gpu_lcpcomp_factorize.argtypes = [vp, vp, sz, u32, i32, pvp, pvp, pvp, psz, ctypes.POINTER(Stats)]
    L.tdc_gpu_flatten.argtypes = [vp, sz, vp, vp, vp, sz, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.tdc_gpu_encode_huff.argtypes = [vp, vp, sz, vp, vp, vp, sz, pvp, psz]
    L.tdc_gpu_lcpcomp_decompress.argtypes = [vp, vp, sz, pvp, psz, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    L.tdc_gpu_lcpcomp_decompress_coder.argtypes = [vp, vp, sz, ctypes.c_int, pvp, psz, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    L.tdc_gpu_encode_arith.argtypes = [vp, vp, sz, vp, vp, vp, sz, pvp, psz]
    L.tdc_gpu_encode_ascii.argtypes = [vp, vp, sz, vp, vp, vp, sz, pvp, psz]
    L.tdc_gpu_encode_sle.argtypes = [vp, vp, sz, vp, vp, vp, sz, ctypes.c_uint32, pvp, psz]
    L.tdc_escape.argtypes = [vp, sz, vp]
    L.tdc_escape.restype = sz
    L.tdc_unescape.argtypes = [vp, sz, vp]
    L.tdc_unescape.restype = sz
    L.tdc_lz78_factors.argtypes = [vp, sz, pvp, pvp, psz]
    L.tdc_gpu_ctx_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_long]
    L.tdc_gpu_option_count.argtypes = []
    L.tdc_gpu_option_name.argtypes = [i32]
    L.tdc_gpu_option_name.restype = ctypes.c_char_p
    L.tdc_huffman_table.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32), vp, vp, vp]
    L.tdc_gen_english.argtypes = [vp, sz, ctypes.c_uint64]
    L.tdc_gen_dna.argtypes = [vp, sz, ctypes.c_uint64]
    _lib = L
    return L
