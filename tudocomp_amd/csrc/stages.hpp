// stages.hpp -- host-callable stage functions of the device pipeline (all arrays are device pointers).
// Each stage replaces one row of SURVEY.md section 8a; citations are relative to
// /root/reference/include/tudocomp/.
#pragma once
#include "common.hpp"
#include <functional>

namespace tdc {

// a1: io/RestrictedBuffer.hpp:43-74 + io/EscapeMap.hpp:39-64 on the device (d_out: n + #escapes + 1 bytes); returns the escaped length
size_t escape_device(Ctx& c, const u8* d_in, size_t n, u8* d_out);
size_t count_escapes_device(Ctx& c, const u8* d_in, size_t n);     // bytes the escaping adds

struct SAStats { u32 rounds = 0; u32 sym_bits = 0; u32 init_syms = 0; u64 sorted_elems = 0;
                 u32 wide_kw = 0; /* key words of the wide initial sort (0: classic path) */ u32 text_rounds = 0; u64 wide_nonheads = 0;
                 u32 overlapped = 0; /* 1: level 1 of the wide sort ran behind the upload */
                 u64 pair_resolved = 0; /* suffixes of two-member groups ordered by the one-pass pair step in front of the doubling rounds */
                 u64 star_chains = 0; /* chains of the star step (round 6: every group ordered against its smallest member in one pass) */ };
// Optional sink of the wide path (suffix_array.hip): lcp8 = n bytes for the LCP (in symbols) of every suffix-array slot with its
// predecessor.  mode (out): 0 = classic result (sa and isa written), 1 = sa final and lcp8 valid but isa NOT written -- the caller
// derives ISA, Phi and PLCP with build_isa_phi_plcp_fused().
struct SAExtra { u8* lcp8 = nullptr; int mode = 0; };

// a2+a3: ds/SADivSufSort.hpp:27-51 and ds/ISAFromSA.hpp:30-43.
// Prefix doubling; text[n-1] must be the unique 0.  sa and isa are caller-provided (n entries each).
void build_suffix_array(Ctx& c, const u8* text, size_t n, u32* sa, u32* isa, SAStats* st, SAExtra* ex = nullptr);
// a3 + a4 + a5 in one pass over a final suffix array whose neighbour LCPs are known (lcp8[i] = lcp(T[sa[i-1]..], T[sa[i]..]), i >= 1):
// isa[sa[i]] = i, phi[sa[i]] = sa[i-1] (phi[sa[0]] = sa[n-1]), plcp[sa[i]] = lcp8[i] (plcp[sa[0]] = 0); d_maxlcp receives the maximum.
// ds/ISAFromSA.hpp:30-43, ds/PhiFromSA.hpp:35-45, ds/PLCPFromPhi.hpp:27-53 (same arrays, computed from the sort instead of the text)
void build_isa_phi_plcp_fused(Ctx& c, const u32* sa, const u8* lcp8, size_t n, u32* isa, u32* phi, u32* plcp, u32* d_maxlcp);
// byte histogram of a text into the context's cache (c.hist_cache / hist_ptr / hist_n): add() per part, finish() once
void text_histogram_add(Ctx& c, const u8* part, size_t len, u32* d_hist);
void text_histogram_finish(Ctx& c, const u8* text, size_t n, const u32* d_hist);

// a4: ds/PhiFromSA.hpp:35-45
void build_phi(Ctx& c, const u32* sa, size_t n, u32* phi);
// a5: ds/PLCPFromPhi.hpp:27-53 ; plcp[n-1] := 0 ; d_maxlcp (device u32) receives max PLCP
void build_plcp(Ctx& c, const u8* text, size_t n, const u32* phi, u32* plcp, u32* d_maxlcp);
// len[i] = lcp(T[i..], T[src[i]..]) for sources with len[i] >= len[i-1] - 1 (same chunked carry as build_plcp);
// src[i] == NONE32 -> 0.  d_max receives the maximum.
void build_lce_with_carry(Ctx& c, const u8* text, size_t n, const u32* src, u32* len, u32* d_max);
// a6 (debug/fixtures only): ds/LCPFromPLCP.hpp:27-56
void build_lcp(Ctx& c, const u32* sa, const u32* plcp, size_t n, u32* lcp);

// Position-space factor representation shared by factorize / flatten / encode:
//   flen[p]  : length of the factor that STARTS at p, 0 otherwise (after mark_literal_runs: run length at
//              the first position of every literal run)
//   owner[p] : identifies the factor covering p (its index in position order; lzss_lcp writes the start position
//              instead), NONE32 if p is a literal.  A factor starts at p iff owner[p] != NONE32 and owner[p-1] != owner[p].
//   fsrc[p]  : source of the factor starting at p (valid where flen[p] > 0 and owner[p] == p)
struct FactorSpace {
    u32* flen = nullptr;
    // optional (round 5, the metric's path): the factor lengths as BYTES until build_owner() has turned them into the list -- flen8[p] = length
    // of the factor that starts at p (255: the length is flen[p]), 0 elsewhere.  Only flen8 is zero-filled then (2 instead of 8 bytes of
    // traffic per position for the fill and for either pass of build_owner); flen[] is valid at the starts of factors of 255 and more
    // positions only, until expand_flen8() fills it for a caller that wants the dense array after all.
    u8* flen8 = nullptr;
    u32* owner = nullptr;
    u32* fsrc = nullptr;
    // optional: the factor starts in position order (n entries of capacity).  build_owner() fills it and sets
    // have_list; flatten and encode then skip their own extraction (factor positions never change after that).
    u32* fpos = nullptr;
    u32* flenl = nullptr;      // optional companion of fpos: the factor lengths in the same order (filled with it)
    size_t nfact = 0;
    bool have_list = false;
    // optional: one class byte per position -- 0 literal, 2 factor start, 3 covered by a factor that started earlier.  build_owner()
    // fills it (have_cls); the encoder's streaming passes then read 1 byte instead of the 4-byte owner word per position.
    u8* cls = nullptr;
    bool have_cls = false;
    // optional (round 6, the metric's path): the top owner_rem_bits bits of a covered position's owner word hold how far its factor still
    // reaches -- q = min(end of the factor - p - 1, 2^bits - 1); the rank sits in the bits below.  A flatten step that lands on p with a
    // copy longer than q + 1 (q not saturated) knows that the copy does not fit WITHOUT fetching the covering factor's record: 43 % of all
    // chain visits on English text.  Requested with want_owner_rem by a caller whose later stages never read owner[] (they read cls[] and
    // the flatten records); build_owner() takes the bits the ranks leave free (at most want_owner_rem, at most 8) and sets owner_rem_bits.  NONE32 stays NONE32.
    u32 want_owner_rem = 0;     // 0: plain ranks; else the most bits to take (8 is all build_owner() ever takes)
    u32 owner_rem_bits = 0;
    // lazy sources (round 4): set by factorize_arrays when there is no Phi array -- fsrc[] then holds the sources of the factors of the
    // global levels only; the source of any factor start p is  src_prio[p] < src_n ? src_sa[src_prio[p] - 1] : fsrc[p]  (src_prio = ISA
    // unless a push at a global level overwrote it, and such a position had its source saved first).  flatten_factors computes it
    // while it builds its records; everyone else calls materialize_sources() first.
    const u32* src_prio = nullptr;
    const u32* src_sa = nullptr;
    size_t src_n = 0;
};
// owner[] for a reader that compares whole words (everyone but the flatten rounds): refuses an array that carries remainder bits
inline const u32* plain_owner(const FactorSpace& fs) {
    if (fs.owner_rem_bits) throw HipError{hipErrorUnknown, "owner[] carries remainder bits (FactorSpace::owner_rem_bits): this stage reads plain ranks", (int)__LINE__};
    return fs.owner;
}
void materialize_sources(Ctx& c, size_t n, FactorSpace& fs);    // fills fsrc[] at every factor start, clears src_prio

struct FactorizeStats { u64 factors = 0; u32 maxlcp = 0; u32 levels = 0; u32 rounds = 0; u64 pushes = 0; u64 entries = 0; u32 small_levels = 0; u32 purges = 0;
                        u32 window_pass = 0; /* 0 not used, 1 low levels done window-local, 2 window pass failed -> global loop */
                        u32 window_lcut = 0; /* highest level of the last window pass */
                        u32 probes = 0; /* skip-ahead probes over runs of erased levels */
                        u32 eager_levels = 0, eager_phases = 0; /* levels inside one-launch runs of small levels (factorize_eager.hip), runs */ };

// a8: compressors/lcpcomp/compress/ArraysComp.hpp:36-117 in position space.
// Inputs: isa, phi, plcp.  isa and plcp are consumed: they become the working priority / LCP arrays.
// Outputs: fs.flen / fs.owner filled, fs.fsrc[p] = phi[p] at factor starts.
void factorize_arrays(Ctx& c, size_t n, const u32* sa, u32* isa, const u32* phi, u32* plcp, u32 maxlcp,
                      u32 threshold, FactorSpace& fs, FactorizeStats* st);

// owner[] from the factor starts (flen[p] != 0 exactly at factor starts): owner[q] = start of the factor covering q, else NONE32
void build_owner(Ctx& c, size_t n, FactorSpace& fs);
// compact lengths (fs.flen8) -> the dense flen[] array; fs.flen8 is cleared
void expand_flen8(Ctx& c, size_t n, FactorSpace& fs);

// lcpcomp(comp=plcppeaks): lcpcomp::PLCPPeaksStrategy (compressors/lcpcomp/compress/PLCPPeaksStrategy.hpp:36-80); fills fs
// (flen, fsrc, owner, factor list) like factorize_arrays
void plcp_peaks_factorize(Ctx& c, size_t n, const u32* phi, const u32* plcp, u32 threshold, FactorSpace& fs, u64* nfactors);
// lcpcomp::MaxLCPStrategy (compressors/lcpcomp/compress/MaxLCPStrategy.hpp:36-100) in position space; isa and plcp are
// consumed like in factorize_arrays
void factorize_max_lcp(Ctx& c, size_t n, u32* isa, const u32* phi, u32* plcp, u32 maxlcp, u32 threshold, FactorSpace& fs,
                       FactorizeStats* st);

// lcpcomp::MaxHeapStrategy (compressors/lcpcomp/compress/MaxHeapStrategy.hpp:36-101, ds/ArrayMaxHeap.hpp): sequential replay on
// the device (a parity row: its tie order is the layout history of a binary heap); sa / isa / plcp are only read
void factorize_max_heap(Ctx& c, size_t n, const u32* sa, const u32* isa, const u32* plcp, u32 maxlcp, u32 threshold, FactorSpace& fs,
                        FactorizeStats* st);

struct FlattenStats { u64 num_flattened = 0; u64 max_depth_lb = 0; u32 rounds = 0; };
// a10: compressors/lzss/LZSSFactors.hpp:79-132 ; rewrites fs.fsrc in place.
// `between` (optional) is called with r = 1, 2, ... once round r has been enqueued and before the host waits for its count, and with 0
// when the rounds are over: the place for work that does not need the flattened sources (api.hip runs the first half of the encoder
// there, step by step on another stream).  Whatever it takes from the arena is gone when flatten_factors returns.
// rec_keep (optional, room for 16 bytes per factor): the records {pos, len, original source, final source} in position order are built
// there and stay valid for the caller; fs.fsrc is NOT rewritten then.
void flatten_factors(Ctx& c, size_t n, FactorSpace fs, FlattenStats* st, const std::function<void(int)>& between = {}, void* rec_keep = nullptr);

// a9: extract the factor list sorted by pos (LZSSFactors.hpp:69-76): pos[], src[], len[] (z entries each,
// arrays caller-provided with capacity cap).  Returns z.
size_t extract_factors(Ctx& c, size_t n, FactorSpace fs, u32* pos, u32* src, u32* len, size_t cap);
// inverse: scatter a sorted factor list into position space (used by the stage-level test entry points)
void scatter_factors(Ctx& c, size_t n, const u32* pos, const u32* src, const u32* len, size_t z, FactorSpace fs);

struct EncodeStats { u64 factors = 0; u64 flen_min = 0, flen_max = 0, fdist_max = 0; u64 out_bits = 0; u32 sigma = 0; };
// a11-a14: LZSSLiterals.hpp:10-50, HuffmanCoder.hpp:37-48/442-474/526-569, LZSSCoding.hpp:18-92, BitOStream.hpp:53-64.
// Writes the complete stream (incl. terminator) to d_out (capacity out_cap bytes); returns its length.
// fs.flen is modified (literal-run lengths are stored at run starts).
size_t encode_huff(Ctx& c, const u8* text, size_t n, FactorSpace fs, u8* d_out, size_t out_cap, EncodeStats* st);
// the same with a selectable coder: 0 = HuffmanCoder, 1 = ArithmeticCoder (a15: coders/ArithmeticCoder.hpp:35-177),
// 2 = ASCIICoder (coders/ASCIICoder.hpp:29-50; every integer and bit of the token stream as text)
// `early`: the first half (everything in front of the pack: gaps, histogram, coder header, bits per tile and their scan -- none of
// it reads the factors' sources) has already run, see encode_early_*
struct EncodeEarly;
size_t encode_stream(Ctx& c, const u8* text, size_t n, FactorSpace fs, int coder, u8* d_out, size_t out_cap, EncodeStats* st, EncodeEarly* early = nullptr);
// the first half on its own, for coder 0 and a factor space with list and class bytes (build_owner): _reserve takes the scratch from
// the arena (call it BEFORE flatten_factors takes its lists), _run enqueues on c.stream and waits for the results
// z_rec > 0: also room for the z_rec records of the flatten stage (encode_early_rec; hand it to flatten_factors as rec_keep): the pack
// then takes lengths and flattened sources from the records and flatten_factors leaves fs.fsrc as it is
EncodeEarly* encode_early_reserve(Ctx& c, size_t n, size_t z_rec = 0);
void* encode_early_rec(EncodeEarly* e);
// _run: the next of its three steps (gaps + histogram | code table, bits per tile, scans | collect), or with finish = true all that are left.
// Only the last one waits for the device if the steps are spread over time (their read-backs travel through the mapped host area).
void encode_early_run(Ctx& c, const u8* text, size_t n, FactorSpace fs, int coder, EncodeEarly* e, bool finish = true);
void encode_early_free(EncodeEarly* e);

// worst-case output size of encode_huff for a text of n bytes
size_t encode_bound(size_t n);
size_t encode_bound_coder(size_t n, int coder);    // coder as in encode_stream (2 = ASCIICoder needs twice as much)

struct LzssStats { u64 factors = 0; };
// a18: compressors/LZSSLCPCompressor.hpp:60-115 (lzss_lcp): greedy LZ77 parse from the previous / next smaller values of
// the suffix array (ANSV) -- fills fs like factorize_arrays (no flatten for this compressor).
void lzss_lcp_factorize(Ctx& c, const u8* text, size_t n, const u32* sa, const u32* isa, u32 threshold, FactorSpace fs,
                        LzssStats* st);

}  // namespace tdc
#include <vector>
namespace tdc {
// LCPCompressor::decompress (LCPCompressor.hpp:140-150; also lzss_lcp streams): host parse of the Huffman token stream,
// references resolved on the device by pointer jumping.  `text` receives the (still escaped, 0-terminated) text.
struct DecodeStats { u64 factors = 0; u32 rounds = 0; u32 device_parse = 0; };
struct StreamFormatError { const char* what; };          // malformed input
// destination of a decoded text: `into` (cap bytes) if set, else `owned` is allocated by the decoder (release with free())
struct DecodeOut { u8* into = nullptr; size_t cap = 0; u8* owned = nullptr; };
// the same for streams written with another coder: 0 = HuffmanCoder, 2 = ASCIICoder, 3 | kmer << 8 = SLECoder
size_t decode_lzss(Ctx& c, const u8* stream, size_t len, int coder, DecodeOut& out, DecodeStats* st);
// a17: compressors/LZ78Compressor.hpp:64-140 -- sequential parse on the host; returns the number of (id, char) pairs
size_t lz78_parse_host(const u8* in, size_t n, std::vector<u32>& ids, std::vector<u8>& chars, bool* leftover_is_high);
// a16: coders/EliasGammaCoder.hpp:26-29 + io/BitOStream.hpp:105-129 on the device; returns the stream length
size_t lz78_gamma_encode(Ctx& c, const u32* d_ids, const u8* d_chars, size_t z, u8* d_out, size_t out_cap);

}  // namespace tdc
