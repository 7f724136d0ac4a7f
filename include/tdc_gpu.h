/*
 * tdc_gpu.h -- C ABI of the MI355X-native lcpcomp hot path (libtdc_gpu.so).
 *
 * This is the drop-in boundary for tudocomp's lcpcomp compressor: a tudocomp maintainer binds these entry
 * points from  tdc::LCPCompressor<HuffmanCoder, lcpcomp::ArraysComp, ...>::compress
 * (/root/reference/include/tudocomp/compressors/LCPCompressor.hpp:100-138); INTEGRATION.md shows the binding.
 * Plain pointers and sizes only; no C++/torch types; no exception crosses the boundary.
 *
 * Conventions
 *   - `text`/`n` is what Input::as_view() hands to compress(): already escaped, terminated by ONE 0 byte that
 *     occurs nowhere else (ds/SADivSufSort.hpp:20-25, ds/TextDS.hpp:132-138).  n < 2^31 (32-bit len_t, def.hpp:103).
 *   - every function returns 0 on success or a negative tdc_gpu_status; tdc_gpu_strerror() explains it.
 *   - host output buffers returned through `uint8_t** out` are malloc'd by the library: free with tdc_gpu_free().
 *   - a context owns one HIP stream and one device arena on one GPU; it is not thread-safe, use one per thread.
 *     Every call switches the calling thread to the context's device and restores the previous current device on return.
 *     The library NEVER falls back to a CPU path: without a usable GPU every compute call fails with TDC_GPU_ERR_HIP.
 */
#ifndef TDC_GPU_H
#define TDC_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TDC_GPU_OK = 0,
    TDC_GPU_ERR_HIP = -1,          /* HIP runtime error (no device, launch failure, ...) */
    TDC_GPU_ERR_ARG = -2,          /* invalid argument (NULL pointer, threshold 0, extra 0 bytes in the text, ...) */
    TDC_GPU_ERR_NO_SENTINEL = -3,  /* text does not end with 0: reference throws std::logic_error (TextDS.hpp:132-138) */
    TDC_GPU_ERR_TOO_LARGE = -4,    /* n >= 2^31 */
    TDC_GPU_ERR_OOM = -5,          /* device or host memory exhausted */
    TDC_GPU_ERR_UNSUPPORTED = -6,  /* coder / strategy not available in this build */
    TDC_GPU_ERR_INTERNAL = -7      /* invariant violated */
} tdc_gpu_status;

/* coder ids (option `coder`, etc/registry_config.py:28-31,138-142) */
enum { TDC_GPU_CODER_HUFF = 0, TDC_GPU_CODER_GAMMA = 1, TDC_GPU_CODER_ARITH = 2, TDC_GPU_CODER_ASCII = 3, TDC_GPU_CODER_SLE = 4 };
/* coder=sle(kmer=K) (coders/SLECoder.hpp:36-40; the reference's default is 3): the option travels in bits 8.. of `coder` */
#define TDC_GPU_CODER_SLE_K(K) (TDC_GPU_CODER_SLE | ((K) << 8))
/* factorization strategy of lcpcomp (option `comp`, LCPCompressor.hpp:87): ArraysComp or PLCPPeaksStrategy */
enum { TDC_GPU_COMP_ARRAYS = 0, TDC_GPU_COMP_PLCPPEAKS = 1, TDC_GPU_COMP_MAXLCP = 2, TDC_GPU_COMP_HEAP = 3 };

typedef struct tdc_gpu_ctx tdc_gpu_ctx;

/* Replaces the StatPhase log of LCPCompressor::compress (same keys: LCPCompressor.hpp:117-118, ArraysComp.hpp:43,60,
 * LZSSFactors.hpp:130-131) plus device timings per phase in milliseconds (hipEvent). */
typedef struct {
    uint64_t n;                 /* text length incl. sentinel                                  */
    uint64_t out_len;           /* compressed bytes                                            */
    uint64_t factors;           /* "factors"                                                   */
    uint64_t maxlcp;            /* "maxlcp"                                                    */
    uint64_t entries;           /* "entries" (initial candidates)                              */
    uint64_t num_flattened;     /* "num_flattened"                                             */
    uint64_t max_depth_lb;      /* "max_depth_lb"                                              */
    uint64_t flen_min, flen_max, fdist_max;
    uint64_t pushes;            /* lazily pushed-down candidates                               */
    uint32_t sa_rounds;         /* prefix-doubling rounds (incl. the initial sort)             */
    uint32_t sa_init_syms;      /* symbols packed into the initial sort key                    */
    uint32_t levels;            /* non-empty LCP levels processed                              */
    uint32_t mis_rounds;        /* selection rounds over all levels                            */
    uint32_t flatten_rounds;
    uint32_t sigma;             /* literal alphabet size                                       */
    uint64_t sa_sorted_elems;   /* total elements that went through the radix sort during SA   */
    uint64_t arena_bytes;       /* device memory high-water mark                               */
    float ms_h2d, ms_sa, ms_phi, ms_plcp, ms_factorize, ms_flatten, ms_encode, ms_d2h, ms_total;
    uint32_t small_levels;      /* levels processed by the one-workgroup kernel                 */
    uint32_t purges;            /* bulk removals of erased candidates                           */
    uint32_t window_pass;       /* low levels window-local in one launch: 0 not used, 1 done, 2 fell back to the level loop */
    uint32_t window_lcut;       /* highest level handed to the (last) window pass                */
    uint32_t sa_key_words;      /* 64-bit words of the initial suffix-sort key (0: classic path, 1 | 2: wide bit-packed keys) */
    uint32_t sa_text_rounds;    /* rank-free refinement rounds keyed from the text (wide path)     */
    uint32_t sa_mode;           /* 1: ISA / Phi / PLCP came from the fused scatter of the final suffix array, 0: classic */
    uint32_t sa_overlapped;     /* 1: the first partition level of the suffix sort ran chunk by chunk behind the upload */
    uint32_t eager_levels;      /* levels processed inside one-launch runs of small levels (factorize_eager.hip)           */
    uint32_t eager_phases;      /* such runs                                                                            */
    uint32_t sa_star_chains;    /* chains of the star step of the suffix array's doubling fall-back (0: the step was not taken) */
} tdc_gpu_stats;

/* ---- context -------------------------------------------------------------------------------------------- */
int  tdc_gpu_ctx_create(int device, tdc_gpu_ctx** ctx);
void tdc_gpu_ctx_destroy(tdc_gpu_ctx* ctx);
/* Options (round 6).  The library's defaults are the product; every switch that tests, A/B measurements and diagnostics need is an
 * option of the context, set through this function and through nothing else: the library does not read TDC_GPU_* environment variables
 * (an embedding process cannot change the algorithm by accident) -- unless TDC_GPU_DEBUG_KNOBS=1 is set, in which case
 * tdc_gpu_ctx_create() applies every TDC_GPU_<OPTION NAME IN UPPER CASE> variable through this same function (development aid,
 * tools/ab.sh).  `name`: an option name (README.md lists them; "wsort_min" and "TDC_GPU_WSORT_MIN" are the same option); out-of-range
 * values are clamped.  TDC_GPU_ERR_ARG for an unknown name.  tdc_gpu_option_count / _name enumerate the table. */
int tdc_gpu_ctx_set_option(tdc_gpu_ctx* ctx, const char* name, long value);
int tdc_gpu_option_count(void);
const char* tdc_gpu_option_name(int i);
/* Pre-size the device arena for texts up to n bytes (optional; otherwise grown on demand). */
int  tdc_gpu_ctx_reserve(tdc_gpu_ctx* ctx, size_t n);
/* Device memory a context holds while it works on a text of n bytes (its arena; 112 bytes per text byte + 192 MiB), and what the
 * device has: callers that place several contexts on one device (block mode, one process per GPU next to RCCL buffers) size
 * their shards with it.  A call whose arena does not fit fails with TDC_GPU_ERR_OOM and a message that names both numbers. */
size_t tdc_gpu_arena_bytes(size_t n);
int  tdc_gpu_device_memory(int device, size_t* free_bytes, size_t* total_bytes);
/* Live kernel timing for bench.py's roofline: when enabled, HIP events are recorded (on the launching stream) around
 * every launch of the instrumented kernels; tdc_gpu_ctx_kernel_profile() returns the sums since the last reset.
 * idx enumerates the instrumented kernel classes from 0; the function returns the class name, or NULL once idx is
 * out of range. `bytes` = algorithmic bytes summed over the launches. */
int  tdc_gpu_ctx_set_profiling(tdc_gpu_ctx* ctx, int enabled);
void tdc_gpu_ctx_reset_profile(tdc_gpu_ctx* ctx);
const char* tdc_gpu_ctx_kernel_profile(const tdc_gpu_ctx* ctx, int idx, double* ms, uint64_t* launches, uint64_t* bytes);
const char* tdc_gpu_strerror(int status);
/* Human-readable detail of the last failure on this context ("" if none). */
const char* tdc_gpu_last_error(const tdc_gpu_ctx* ctx);
void tdc_gpu_free(void* p);

/* ---- the hot path: replaces LCPCompressor::compress (LCPCompressor.hpp:100-138) --------------------------- */
/* Host buffers in, host buffer out (H2D + all kernels + D2H).  threshold/flatten = the dynamic options of the
 * same name (LCPCompressor.hpp:92-93, defaults 5 and 1).  `stats` may be NULL.
 * coder: TDC_GPU_CODER_HUFF (HuffmanCoder) or TDC_GPU_CODER_ARITH (ArithmeticCoder, coders/ArithmeticCoder.hpp:35-177 --
 * compress side only: the reference itself cannot decode lcpcomp + arithmetic; returns
 * TDC_GPU_ERR_UNSUPPORTED for inputs on which the reference divides by zero), TDC_GPU_CODER_ASCII (ASCIICoder) or
 * TDC_GPU_CODER_SLE / TDC_GPU_CODER_SLE_K(k) (SLECoder, the coder of the reference's published lcpcomp runs,
 * etc/compare-suites/default.suite:5; k in 1..7 = the reference's max_kmer, SLECoder.hpp:12). */
int tdc_gpu_lcpcomp_compress(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                             int coder, uint8_t** out, size_t* out_len, tdc_gpu_stats* stats);
/* The same with a selectable factorization strategy: comp = TDC_GPU_COMP_ARRAYS (lcpcomp::ArraysComp, the default of the
 * entry point above) or TDC_GPU_COMP_PLCPPEAKS (lcpcomp::PLCPPeaksStrategy, compressors/lcpcomp/compress/PLCPPeaksStrategy.hpp:36-80:
 * strict local maxima of the PLCP array, one left-to-right scan) or TDC_GPU_COMP_MAXLCP (lcpcomp::MaxLCPStrategy,
 * compressors/lcpcomp/compress/MaxLCPStrategy.hpp:36-100: the same greedy rule as ArraysComp with the tie order of its
 * per-level stacks and eager key decreases) or TDC_GPU_COMP_HEAP (lcpcomp::MaxHeapStrategy, MaxHeapStrategy.hpp:36-101 over
 * ds/ArrayMaxHeap.hpp: the strategy of the reference's published heap run; its tie order is the layout history of a binary
 * heap, so the device replays the reference's loop with ONE thread -- a parity row, about a minute per MiB of text). */
int tdc_gpu_lcpcomp_compress_comp(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                                  int comp, uint8_t** out, size_t* out_len, tdc_gpu_stats* stats);

/* The metric's entry point (pinned host text -> compressed bytes in host memory): the same as
 * tdc_gpu_lcpcomp_compress_comp, but the stream is written into the CALLER's buffer `out` of out_cap bytes (no allocation
 * in the call).  *out_len receives the stream length; if it exceeds out_cap the call fails with TDC_GPU_ERR_OOM and *out_len
 * holds the required size.  `text` and `out` should be pinned host memory (tdc_gpu_host_alloc, or hipHostMalloc /
 * hipHostRegister by the embedding program): the two transfers then run at PCIe rate; pageable memory works but is staged
 * by the runtime.  stats->ms_h2d / ms_d2h / ms_total cover the transfers. */
int tdc_gpu_lcpcomp_compress_into(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                                  int comp, uint8_t* out, size_t out_cap, size_t* out_len, tdc_gpu_stats* stats);
/* One process per GPU, one container per node (SURVEY.md 8e replaces nothing of the reference: its 32-bit len_t has
 * no block mode).  tdc_gpu_lcpcomp_compress_keep is tdc_gpu_lcpcomp_compress_into without the download: the stream STAYS on the
 * device inside the context until the next call on it, *out_len receives its length.  Once the ranks have exchanged their lengths,
 * tdc_gpu_stream_fetch copies the kept stream to `dst` -- this rank's offset in a container that all ranks map, e.g. a POSIX
 * shared-memory segment page-locked with tdc_gpu_host_register -- so every shard travels over its own GPU's host link instead of
 * all of them through rank 0.  tdc_gpu_stream_fetch: *len (nullable) receives the stream length; TDC_GPU_ERR_OOM if cap is smaller. */
int tdc_gpu_lcpcomp_compress_keep(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                                  int comp, size_t* out_len, tdc_gpu_stats* stats);
int tdc_gpu_stream_fetch(tdc_gpu_ctx* ctx, uint8_t* dst, size_t cap, size_t* len);
/* The same with a DEVICE destination on the context's GPU (e.g. the send buffer of an RCCL gather of the per-block streams to one
 * rank, SURVEY.md 8e "Collective"): a device-to-device copy on the context's stream, synchronised before the call returns. */
int tdc_gpu_stream_fetch_dev(tdc_gpu_ctx* ctx, void* d_dst, size_t cap, size_t* len);
/* Page-lock / release host memory the embedding program allocated itself (hipHostRegister): transfers then run at PCIe rate. */
int tdc_gpu_host_register(void* p, size_t bytes);
int tdc_gpu_host_unregister(void* p);
/* Pinned (page-locked) host memory for the buffers above; NULL on failure.  Free with tdc_gpu_host_free. */
void* tdc_gpu_host_alloc(size_t bytes);
void  tdc_gpu_host_free(void* p);

/* Raw input variant: `data`/`n` is the UNRESTRICTED input (any bytes, no sentinel).  The library applies the
 * compressor's input restrictions on the device -- escape {0} + null-terminate, i.e. what Input(inp, restrictions) does
 * in tudocomp_driver.cpp:268-270 (io/RestrictedBuffer.hpp:43-74) -- and then compresses.  The escaped text (n + number of
 * 0x00 / 0xFF bytes + 1) must stay below 2^31 - 1 bytes. */
int tdc_gpu_lcpcomp_compress_raw(tdc_gpu_ctx* ctx, const uint8_t* data, size_t n, uint32_t threshold, int flatten,
                                 int coder, uint8_t** out, size_t* out_len, tdc_gpu_stats* stats);
/* Device-resident variant: d_text and d_out are device pointers on ctx's GPU (d_out 8-byte aligned, capacity out_cap
 * bytes; tdc_gpu_lcpcomp_bound_coder(n, coder) always suffices -- tdc_gpu_lcpcomp_bound(n) is that bound for huff and
 * arithmetic; ascii needs twice as much).  The call runs on the context's own stream: the caller must have finished
 * (synchronised) whatever produced d_text before calling, and the stream is synchronised before the call returns. */
int tdc_gpu_lcpcomp_compress_dev(tdc_gpu_ctx* ctx, const void* d_text, size_t n, uint32_t threshold, int flatten,
                                 int coder, void* d_out, size_t out_cap, size_t* out_len, tdc_gpu_stats* stats);
size_t tdc_gpu_lcpcomp_bound(size_t n);
size_t tdc_gpu_lcpcomp_bound_coder(size_t n, int coder);     /* 0 for an unknown coder */

/* ---- block mode (north_star: inputs above one-GPU size shard into independent blocks; BASELINE.json configs[4]) ----
 * `data`/`n` is the UNRESTRICTED input.  It is cut into ceil(n / block_size) blocks of block_size bytes (the last one shorter;
 * block_size < 2^31 - 2 and small enough that the ESCAPED block stays below 2^31 - 1 bytes); block k is compressed exactly like
 * tdc_gpu_lcpcomp_compress_raw(data + k * block_size, ...) -- own escaping + sentinel, own suffix array, factors and Huffman
 * table -- on one of the `ndev` devices listed in `devices` (one host thread and one context per device, blocks handed out
 * from a shared counter).  *out (malloc'd, tdc_gpu_free) receives the container
 *     "tdcgpu-blocks%" | u32 G | G x { u64 raw_len, u64 comp_len } | payload_0 | ... | payload_{G-1}      (little endian)
 * whose payloads are byte-identical to the single-block streams.  per_block (nullable): G stats records.
 * The reference has no counterpart (32-bit len_t: an input is at most 2^31 - 1 bytes, def.hpp:103). */
size_t tdc_gpu_blocks_count(size_t n, size_t block_size);
int tdc_gpu_blocks_compress(const int* devices, int ndev, const uint8_t* data, size_t n, size_t block_size, uint32_t threshold,
                            int flatten, int coder, uint8_t** out, size_t* out_len, tdc_gpu_stats* per_block);
/* Inverse: every payload through tdc_gpu_lcpcomp_decompress_coder on ctx's device, restrictions removed (unescape, sentinel
 * dropped), blocks concatenated.  A malformed container: TDC_GPU_ERR_ARG. */
int tdc_gpu_blocks_decompress(tdc_gpu_ctx* ctx, const uint8_t* container, size_t len, int coder, uint8_t** out, size_t* out_len);
/* number of visible devices (0 if none / no runtime) */
int tdc_gpu_device_count(void);

/* ---- LZ78 (BASELINE.json configs[3]): replaces LZ78Compressor<EliasGammaCoder, ...>::compress
 * (compressors/LZ78Compressor.hpp:64-140).  No input restrictions (no escaping, no sentinel).  The parse is sequential
 * and runs on the host; the Elias-gamma stream is packed on the GPU.  coder must be TDC_GPU_CODER_GAMMA.
 * stats (may be NULL): n, out_len, factors (= number of phrases) and ms_encode / ms_total are filled. */
int tdc_gpu_lz78_compress(tdc_gpu_ctx* ctx, const uint8_t* in, size_t n, int coder, uint8_t** out, size_t* out_len,
                          tdc_gpu_stats* stats);

/* ---- lzss_lcp (SURVEY.md 8a row a18 / 8f "next" #1): replaces LZSSLCPCompressor<HuffmanCoder>::compress
 * (compressors/LZSSLCPCompressor.hpp:41-123): greedy LZ77 parse via previous / next smaller values of the suffix array.
 * Same text contract as lcpcomp (escaped, 0-terminated); option threshold (default 3, :30). */
int tdc_gpu_lzss_lcp_compress(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int coder,
                              uint8_t** out, size_t* out_len, tdc_gpu_stats* stats);
/* the factor list of LZSSLCPCompressor.hpp:60-115 (sorted by pos), three malloc'd arrays */
int tdc_gpu_lzss_lcp_factorize(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold,
                               uint32_t** pos, uint32_t** src, uint32_t** len, size_t* z);

/* ---- stage-level entry points (host buffers), used by the parity tests ------------------------------------ */
/* the device sorts behind the suffix array (no reference counterpart; for the tests): sorts n (key, value) pairs in place by
 * the 64-bit key; algo 0 = stable 8-bit LSD radix sort, 1 = splitter-partition sort (unstable) */
int tdc_gpu_sort_pairs_u64(tdc_gpu_ctx* ctx, uint64_t* keys, uint32_t* vals, size_t n, int algo);
/* ds/SADivSufSort.hpp:27-51 + ds/ISAFromSA.hpp:30-43 : sa / isa may be NULL */
int tdc_gpu_suffix_array(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t* sa, uint32_t* isa);
/* TextDS::require(SA|ISA|PHI|PLCP|LCP) (ds/TextDS.hpp:247-292); any output may be NULL; plcp[n-1] = 0 */
int tdc_gpu_textds(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t* sa, uint32_t* isa, uint32_t* phi,
                   uint32_t* plcp, uint32_t* lcp, uint32_t* maxlcp);
/* ArraysComp::factorize + FactorBuffer::sort (+ flatten if requested): factors sorted by pos in three malloc'd arrays */
int tdc_gpu_lcpcomp_factorize(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                              uint32_t** pos, uint32_t** src, uint32_t** len, size_t* z, tdc_gpu_stats* stats);
/* FactorBuffer::flatten on a caller-supplied factor list sorted by pos (LZSSFactors.hpp:79-132); src rewritten in place */
int tdc_gpu_flatten(tdc_gpu_ctx* ctx, size_t n, const uint32_t* pos, uint32_t* src, const uint32_t* len, size_t z,
                    uint64_t* num_flattened, uint64_t* max_depth_lb);
/* ---- LCPCompressor::decompress (LCPCompressor.hpp:140-150 -> decode_text_internal :23-76, HuffmanCoder::Decoder
 * coders/HuffmanCoder.hpp:572-612); lzss_lcp(coder=huff) streams have the same format (LZSSLCPCompressor.hpp:125-130).
 * Streams of 1 MiB and more whose longest literal run is at most 512 are parsed ON THE DEVICE (rounds 4-5, DESIGN.md section 5: where
 * the token that starts at a bit position ends is evaluated for every bit position, the real token starts are the orbit of the
 * first one); smaller streams, longer literal runs and the SLE / ASCII coders take the host parse.  The references -- what ScanDec /
 * CompactDec spend their time on (lcpcomp/decompress/ScanDec.hpp:146-247) -- are resolved on the device by pointer jumping.
 * *out (malloc'd, free with tdc_gpu_free) receives the escaped, 0-terminated text exactly as compress() was given it.
 * factors / rounds (nullable): number of factors in the stream / pointer-jumping rounds.  Malformed input: TDC_GPU_ERR_ARG. */
int tdc_gpu_lcpcomp_decompress(tdc_gpu_ctx* ctx, const uint8_t* stream, size_t len, uint8_t** out, size_t* out_len,
                               uint64_t* factors, uint32_t* rounds);
/* The same for a stream written with another coder: TDC_GPU_CODER_HUFF, TDC_GPU_CODER_ASCII (ASCIICoder::Decoder,
 * coders/ASCIICoder.hpp:53-84) or TDC_GPU_CODER_SLE / TDC_GPU_CODER_SLE_K(k) (SLECoder::Decoder, coders/SLECoder.hpp:301-453). */
int tdc_gpu_lcpcomp_decompress_coder(tdc_gpu_ctx* ctx, const uint8_t* stream, size_t len, int coder, uint8_t** out, size_t* out_len,
                                     uint64_t* factors, uint32_t* rounds);
/* The same into the CALLER's buffer `out` of out_cap bytes (pinned host memory -- tdc_gpu_host_alloc -- receives the text at the host
 * link's rate; the malloc'd variants pay for the page faults of a fresh buffer).  TDC_GPU_ERR_OOM if the text does not fit. */
int tdc_gpu_lcpcomp_decompress_into(tdc_gpu_ctx* ctx, const uint8_t* stream, size_t len, int coder, uint8_t* out, size_t out_cap,
                                    size_t* out_len, uint64_t* factors, uint32_t* rounds);
/* 1 if the last tdc_gpu_lcpcomp_decompress / _decompress_coder / _decompress_into call on this context succeeded and parsed the token stream on the device (coder=huff streams
 * of 1 MiB and more whose longest literal run is at most 512; env TDC_GPU_DEC_PARSE = 0 never / 2 every size; TDC_GPU_DEC_LEAN = 0: the
 * general marking also for streams of short tokens -- tests), 0 if on the host. */
int tdc_gpu_ctx_last_decode_on_device(const tdc_gpu_ctx* ctx);

/* HuffmanCoder::Encoder + lzss::encode_text on a caller-supplied factor list sorted by pos (LZSSCoding.hpp:18-92) */
int tdc_gpu_encode_huff(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                        const uint32_t* len, size_t z, uint8_t** out, size_t* out_len);

/* the same with ArithmeticCoder::Encoder as the literal coder (coders/ArithmeticCoder.hpp:35-177) */
int tdc_gpu_encode_arith(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                         const uint32_t* len, size_t z, uint8_t** out, size_t* out_len);
/* coder = ASCIICoder (coders/ASCIICoder.hpp:29-50) */
int tdc_gpu_encode_ascii(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                         const uint32_t* len, size_t z, uint8_t** out, size_t* out_len);
/* coder = SLECoder (coders/SLECoder.hpp:42-298), kmer = its option of that name (0 = default 3; 1..7) */
int tdc_gpu_encode_sle(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                       const uint32_t* len, size_t z, uint32_t kmer, uint8_t** out, size_t* out_len);

/* ---- host-side helpers (no GPU) --------------------------------------------------------------------------- */
/* io/RestrictedBuffer.hpp:43-74 + io/EscapeMap.hpp:39-64 : 0x00 -> FF FE, 0xFF -> FF FF, append 0.
 * out must hold 2*n+1 bytes; returns the escaped length. */
size_t tdc_escape(const uint8_t* in, size_t n, uint8_t* out);
/* io/RestrictedIOStream.hpp:13-89 : inverse (drops the final 0); returns the length. */
size_t tdc_unescape(const uint8_t* in, size_t n, uint8_t* out);
/* coders/HuffmanCoder.hpp:442-474 : canonical code for a literal histogram (for tests of the host table builder) */
int tdc_huffman_table(const uint32_t counts[256], uint32_t* sigma, uint32_t* longest, uint8_t order[256],
                      uint8_t len_of[256], uint64_t code_of[256]);
/* compressors/LZ78Compressor.hpp:97-131 : the LZ78 parse on its own (host; what tdc_gpu_lz78_compress codes on the device).
 * ids[k] = id of the longest dictionary phrase at the start of factor k (0: none; ids count from 1 in insertion order), chars[k] = the byte
 * behind it, a leftover phrase at the end of the text as (parent id, last byte).  *ids / *chars are malloc'd (tdc_gpu_free). */
int tdc_lz78_factors(const uint8_t* in, size_t n, uint32_t** ids, uint8_t** chars, size_t* z);
/* The start-up check of tdc_gpu_ctx_create() on its own (no GPU): rebuilds two built-in fixture tables (sigma 40 and 200, many
 * equal counts) and compares them with what the reference build yields; TDC_GPU_ERR_INTERNAL if this build's C++ library
 * orders ties differently (coders/HuffmanCoder.hpp:88-120 heap functions, :455 unstable std::sort) -- every call with
 * coder=huff on a context then fails with the same code (the other coders, lz78 and decompression are not affected). */
int tdc_huffman_selfcheck(void);
/* synthetic corpora of the benchmark configurations */
int tdc_gen_english(uint8_t* out, size_t n, uint64_t seed);
int tdc_gen_dna(uint8_t* out, size_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
