/*
 * tdc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the tudocomp CPU hot path
 * (lcpcomp + HuffmanCoder, plus the stages it is made of).  It exists to check the
 * HIP path; it is never linked into, imported by or executed from the product
 * (tudocomp_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - bit stream, Huffman table/stream, escaping, TextLiterals, bits_for: pinned by the
 *     reference's own known-answer tests (tests/golden/reference_kats.json).
 *   - lcpcomp end-to-end byte streams: the reference's tests hold no golden for them;
 *     pinned by the outputs of the reference recorded in SURVEY.md section 8c
 *     (tests/golden/survey_anchors.json).  The reference cannot be rebuilt in this
 *     image (needs glog + sdsl-lite, both absent), so there is no oracle/_ref.
 *
 * All citations are relative to /root/reference/include/tudocomp/.
 */
#ifndef TDC_ORACLE_H
#define TDC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint32_t pos, src, len; } orc_factor;   /* compressors/lzss/LZSSFactors.hpp:13-20 */

typedef struct {
    uint32_t sigma;            /* effective alphabet size                                  */
    uint32_t longest;          /* longest code word                                        */
    uint8_t  numl[256];        /* numl[l-1] = #code words of length l (u8 like the reference) */
    uint8_t  order[256];       /* symbols in canonical (sorted-by-length) order             */
    uint8_t  len_of[256];      /* per byte value: code length (0 = absent)                 */
    uint64_t code_of[256];     /* per byte value: code word                                */
} orc_hufftable;

typedef struct {
    uint64_t n;                /* text length incl. sentinel */
    uint64_t factors;
    uint64_t maxlcp;
    uint64_t num_flattened;
    uint64_t max_depth_lb;
    uint64_t flen_min, flen_max, fdist_max;
    double   t_sa, t_phi, t_plcp, t_isa, t_factorize, t_sort, t_flatten, t_encode, t_total;
} orc_stats;

/* util.hpp:194 */
unsigned orc_bits_for(uint64_t v);

/* io/RestrictedBuffer.hpp:43-74 + io/EscapeMap.hpp:39-64 : 0x00->FF FE, 0xFF->FF FF, append 0.
 * returns escaped length (incl. sentinel); out must hold 2*n+1 bytes. */
size_t orc_escape(const uint8_t* in, size_t n, uint8_t* out);
/* io/RestrictedIOStream.hpp:13-89 : inverse, drops the final 0. returns length. */
size_t orc_unescape(const uint8_t* in, size_t n, uint8_t* out);

/* ds/SADivSufSort.hpp:27-51 (semantics: the suffix array is unique). text[n-1] must be the unique 0. */
int orc_suffix_array(const uint8_t* text, size_t n, uint32_t* sa);
/* ds/ISAFromSA.hpp:37-39, ds/PhiFromSA.hpp:37-41, ds/PLCPFromPhi.hpp:38-44 (plcp[n-1] := 0), ds/LCPFromPLCP.hpp:43-47 */
void orc_isa(const uint32_t* sa, size_t n, uint32_t* isa);
void orc_phi(const uint32_t* sa, size_t n, uint32_t* phi);
uint32_t orc_plcp(const uint8_t* text, size_t n, const uint32_t* phi, uint32_t* plcp);
void orc_lcp(const uint32_t* sa, const uint32_t* plcp, size_t n, uint32_t* lcp);

/* compressors/lcpcomp/compress/ArraysComp.hpp:36-117. lcp is modified. Returns #factors, *out malloc'd
 * in EMISSION order. */
size_t orc_arrays_comp(const uint32_t* sa, const uint32_t* isa, uint32_t* lcp, size_t n,
                       uint32_t maxlcp, uint32_t threshold, orc_factor** out);
/* LZSSFactors.hpp:69-76 */
void orc_sort_factors(orc_factor* f, size_t z);
/* LZSSFactors.hpp:79-132 */
void orc_flatten(orc_factor* f, size_t z, uint64_t* num_flattened, uint64_t* max_depth);
/* LZSSLiterals.hpp:10-50 + coders/HuffmanCoder.hpp:37-48 */
void orc_literal_histogram(const uint8_t* text, size_t n, const orc_factor* f, size_t z, uint32_t C[256]);
/* LZSSLiterals positions: writes positions of literals, returns count (lzss_test.cpp:95-139) */
size_t orc_literal_positions(size_t n, const orc_factor* f, size_t z, uint32_t* positions);
/* HuffmanCoder.hpp:442-474 (gen_huffmantable) with the libstdc++ heap/introsort behaviour restated */
void orc_huffman_table(const uint32_t C[256], orc_hufftable* t);

/* Whole pipeline: LCPCompressor.hpp:100-138 with coder = HuffmanCoder.
 * text = escaped + 0-terminated view (what Input::as_view() hands to compress()).
 * *out malloc'd; returns 0 on success. */
int orc_lcpcomp_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                              uint8_t** out, size_t* out_len, orc_stats* stats);
/* lcpcomp(comp=plcppeaks): PLCPPeaksStrategy (compressors/lcpcomp/compress/PLCPPeaksStrategy.hpp:36-80) instead of ArraysComp.
 * The reference's tests hold no vector for it and none was recorded: pinned by its properties only (valid copies, round trip). */
/* lcpcomp::MaxLCPStrategy (compressors/lcpcomp/compress/MaxLCPStrategy.hpp:36-100); lcp is modified; factors in emission order */
/* lcpcomp::MaxHeapStrategy (MaxHeapStrategy.hpp:36-101, ds/ArrayMaxHeap.hpp): factors in emission order; lcp is modified */
size_t orc_max_heap(const uint32_t* sa, const uint32_t* isa, uint32_t* lcp, size_t n, uint32_t threshold, orc_factor** out);
int orc_lcpcomp_heap_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                                   uint8_t** out, size_t* out_len, orc_stats* stats);
size_t orc_max_lcp(const uint32_t* sa, const uint32_t* isa, uint32_t* lcp, size_t n,
                   uint32_t maxlcp, uint32_t threshold, orc_factor** out);
int orc_lcpcomp_maxlcp_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                                     uint8_t** out, size_t* out_len, orc_stats* stats);
size_t orc_plcp_peaks(const uint32_t* sa, const uint32_t* isa, const uint32_t* plcp, size_t n, uint32_t threshold, orc_factor** out);
int orc_lcpcomp_peaks_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                                    uint8_t** out, size_t* out_len, orc_stats* stats);
/* Encode a given, sorted factor list (LZSSCoding.hpp:18-92 + HuffmanCoder::Encoder :526-569 + BitOStream dtor). */
int orc_encode_huff(const uint8_t* text, size_t n, const orc_factor* f, size_t z,
                    uint8_t** out, size_t* out_len, orc_stats* stats);
/* The same with coder = ArithmeticCoder (coders/ArithmeticCoder.hpp:35-177; BASELINE.json configs[2]).  Compress side
 * only: the reference cannot decode this combination.  Returns -8 where the reference would divide by 0. */
int orc_lcpcomp_arith_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                               uint8_t** out, size_t* out_len, orc_stats* stats);
int orc_encode_arith(const uint8_t* text, size_t n, const orc_factor* f, size_t z,
                     uint8_t** out, size_t* out_len, orc_stats* stats);
/* The same with coder = ASCIICoder (coders/ASCIICoder.hpp:29-84): decimal integers + ':', '0'/'1' bits, raw literals.
 * Pinned by the stream the reference printed for the SURVEY 8c example text. */
int orc_lcpcomp_ascii_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                               uint8_t** out, size_t* out_len, orc_stats* stats);
int orc_encode_ascii(const uint8_t* text, size_t n, const orc_factor* f, size_t z,
                     uint8_t** out, size_t* out_len, orc_stats* stats);
int orc_lcpcomp_ascii_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
/* lcpcomp(coder=sle(kmer)) -- SLECoder (coders/SLECoder.hpp); kmer in 1..7 (the reference's default is 3) */
int orc_lcpcomp_sle_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten, unsigned kmer,
                             uint8_t** out, size_t* out_len, orc_stats* stats);
int orc_encode_sle(const uint8_t* text, size_t n, const orc_factor* f, size_t z, unsigned kmer,
                   uint8_t** out, size_t* out_len, orc_stats* st);
int orc_lcpcomp_sle_decompress(const uint8_t* in, size_t in_len, unsigned kmer, uint8_t** out, size_t* out_len);
/* LCPCompressor.hpp:140-150 / decode_text_internal :23-76 with HuffmanCoder::Decoder.
 * Produces the (still escaped, 0-terminated) text. *out malloc'd. */
int orc_lcpcomp_huff_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);

/* coders/HuffmanCoder.hpp encoder fed literal by literal (test/test/util.hpp:577-602 test_binary_out):
 * interleave!=0 writes 0b01010101 (8 bits) before the first literal and after every 0 literal. */
int orc_huff_encode_literals(const uint8_t* lits, size_t n, int interleave, uint8_t** out, size_t* out_len);

/* io/BitOStream.hpp scripted writer for the bit-IO KATs: ops[i] = {kind, value, bits};
 * kind 0 = write_bit(value), 1 = write_int(value,bits), 2 = write_compressed_int(value, bits) */
int orc_bitstream_script(const uint64_t* ops, size_t n_ops, uint8_t** out, size_t* out_len);
/* io/BitIStream.hpp: count the bits readable until eof() (tudocomp_tests.cpp:700-727) */
size_t orc_bitstream_count_bits(const uint8_t* in, size_t n);

/* compressors/LZSSLCPCompressor.hpp:60-115 : greedy left-to-right LZ77 via ISA + naive PSV/NSV scans over SA/LCP.
 * Factors come out sorted by pos.  Returns #factors, *out malloc'd. */
size_t orc_lzss_lcp_factorize(const uint32_t* sa, const uint32_t* isa, const uint32_t* lcp, size_t n, uint32_t threshold,
                              orc_factor** out);
/* LZSSLCPCompressor::compress (:41-123) with coder = HuffmanCoder (default threshold 3, no flatten). */
int orc_lzss_lcp_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, uint8_t** out, size_t* out_len, orc_stats* stats);

/* compressors/LZ78Compressor.hpp:64-140 + coders/EliasGammaCoder.hpp:26-29 (config 4) */
int orc_lz78_gamma_compress(const uint8_t* in, size_t n, uint8_t** out, size_t* out_len);
/* LZ78 factor list (parent id, char) for the cedar_tests KATs; returns #pairs, arrays malloc'd */
size_t orc_lz78_factors(const uint8_t* in, size_t n, uint32_t** ids, uint8_t** chars);

void orc_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
