// prim.hpp -- device-wide primitives used by every stage: scans and a stable LSD radix sort.
// All of them are HBM-bound; they are written for 64-lane waves and 256-thread workgroups.
#pragma once
#include "common.hpp"

namespace tdc {

// out[i] = sum_{j<i} in[j]   (in may alias out).  If d_total != nullptr it receives the grand total.
void exclusive_sum_u32(Ctx& c, const u32* in, u32* out, size_t n, u32* d_total);
void exclusive_sum_u64(Ctx& c, const u64* in, u64* out, size_t n, u64* d_total);
// out[i] = max_{j<=i} in[j]
void inclusive_max_u32(Ctx& c, const u32* in, u32* out, size_t n);

// Stable LSD radix sort of (key, value) pairs on bits [begin_bit, end_bit) of the key, 8 bits per pass.
// keys[0]/vals[0] hold the input; the function ping-pongs between [0] and [1] and returns the index of
// the buffer pair that holds the sorted result.  Temporary storage comes from the arena (released on return).
int radix_sort_pairs_u64(Ctx& c, u64* keys[2], u32* vals[2], size_t n, int begin_bit, int end_bit);
int radix_sort_pairs_u32(Ctx& c, u32* keys[2], u32* vals[2], size_t n, int begin_bit, int end_bit);

// The suffix array's initial sort: key(i) = the k recoded bytes text[i .. i+k) as a number in base sigma (zeros behind the
// text), value(i) = i, i < n; sorted on bits [0, end_bit).  Pass 0 computes the keys from the text (no key / index arrays
// are written beforehand); returns the index of the buffer pair that holds the result.
struct TextKeyGen { const u8* text; size_t n; u32 sigma; int k, chunk; u64 top /* sigma^(k-1) */; u8 code[256]; };
int radix_sort_text_keys_u64(Ctx& c, const TextKeyGen& g, u64* keys[2], u32* vals[2], int end_bit);

// Splitter-partition sort (ssort.hip): UNSTABLE sort of the pairs by the whole 64-bit key in 2-3 partition levels on key ranks
// (sampled splitters) plus one in-LDS sort of the leaves -- instead of eight LSD passes.  gen != nullptr: the pairs are
// (key(i), i) of the text as in radix_sort_text_keys_u64 and keys[0] / vals[0] are not read; else keys[0] / vals[0] hold the
// input.  Both buffer pairs are used; returns the index of the pair that holds the result.
struct SplitSortStats { u32 levels = 0, range_leaves = 0, samples = 0, units = 0, large_leaves = 0, wide_units = 0; u64 large_pairs = 0; };
bool splitter_sort_applicable(size_t n);
int splitter_sort_pairs_u64(Ctx& c, u64* keys[2], u32* vals[2], size_t n, const TextKeyGen* gen, SplitSortStats* st);

// Two-level MSD partition (ssort.hip): idx / val (m pairs, only read) -> out_idx / out_val grouped by idx >> (bits - 2 * db), any
// order inside a group; db = digit bits per level (8 or 9); tmp_idx / tmp_val: m entries of scratch each.  bits > 2 * db.
void msd_partition_pairs_u32(Ctx& c, const u32* idx, const u32* val, size_t m, int bits, int db, u32* out_idx, u32* out_val, u32* tmp_idx, u32* tmp_val);

// ---- internals of the splitter sort shared between ssort.hip and wsort.hip ------------------------------------------------------
struct SegTables { u32* blk_start; u32* blk_seg; u32* counts; u32* bs; u32 R, blocks_ub, rows; };
struct UnitTables { u32* unit_rng; u32* cls_list; u32* large; u32 cap, large_cap; u32 hc[6]; /* large leaves, units, units per size class */ int wide_classes = 0; /* size classes of wsort.hip */ u32 whc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; /* units per wide class */ };
// size classes of the leaf kernels of wsort.hip (records): the kernel of a class has just enough rows
constexpr int WIDE_NCLS = 10;
// classes 0 .. 4: 4 .. 8 rows of four waves (1 024 .. 2 048 records), 5 .. 8: 5 .. 8 rows of eight waves (2 560 .. 4 096), 9: 16 rows (8 192)
__host__ __device__ inline u32 wide_class(u32 m) {
    if (m <= 2048u) return m <= 1024u ? 0u : (m - 769u) / 256u;          // 1025..1280 -> 1, ..1536 -> 2, ..1792 -> 3, ..2048 -> 4
    if (m <= 4096u) return 5u + (m <= 2560u ? 0u : (m - 2049u) / 512u);  // ..2560 -> 5, ..3072 -> 6, ..3584 -> 7, ..4096 -> 8
    return 9u;
}
u32* ss_first_segment(Ctx& c, size_t n);                                           // seg_start[2] = { 0, n } on the device
void ss_level_tables(Ctx& c, const u32* seg_start, u32 nseg, size_t n, u32 D, SegTables& T);      // row-block tables + count arrays (arena)
void ss_level_offsets(Ctx& c, const SegTables& T, const u32* seg_start, u32 nseg, u32 D, u32* nstart, size_t n);   // counts -> offsets, next segment starts
void ss_level_offsets_sub(Ctx& c, const SegTables& T, u32 nsub, const u32* blk_super, const u32* out_start, u32 nsuper, u32 D, u32* nstart, size_t n);
void ss_build_units(Ctx& c, const u32* leaf_start, u32 nleaf, UnitTables& U, u32 small = 0);   // leaves -> units of <= 8192 pairs by size class (synchronises);
                                                                                   // leaves of <= `small` pairs (0: 4096) are packed into units of <= 2 * small
void ss_fanouts(Ctx& c, size_t n, int& L, u32 F[3], u32& os, u32 leaf3 = 0, int wide2 = 0);       // levels, fan-outs and oversampling for n pairs (leaf3: leaf size aimed at with three levels, 0 = 2048)

// ---- wide-key suffix sort (wsort.hip) ----------------------------------------------------------------------------------------------
// key(p) = the s recoded bytes text[p .. p+s) as s fields of b bits, left-aligned in KW 64-bit words (pad zero bits below; zeros behind
// the text).  Bit-packed keys make the common prefix of two keys a count of leading zeros, which is where the LCP values come from.
struct WKeyGen { const u8* text; size_t n; int b, s, pad; u32 inv /* ceil(65536 / b) */; u8 code[256]; };
struct WSortStats { u32 levels = 0, range_leaves = 0, samples = 0, units = 0, large_leaves = 0, kw = 0, refined_units = 0, trunc_units = 0, longrun_units = 0, leaf_stages = 0, wave_runs = 0; u64 large_pairs = 0; u64 nonheads = 0; };
// Level 1 of wsort_suffixes done chunk by chunk BEHIND THE UPLOAD of a host text (api.hip compress_host): the code map comes from the
// bytes of chunk 0 (the full histogram confirms it afterwards, else the work is thrown away), the splitters from a sample of chunk 0,
// and every chunk is counted and scattered into its own part of the record buffers as soon as it (and the chunk behind it: a key reads
// up to 64 bytes ahead) has arrived.  Level 2 then reads the buckets of all chunks as pieces of one segment.  Buffers live at the top of
// the arena (Arena::alloc_top) until the suffix array is done.
struct WPre {
    bool begun = false, active = false;
    const u8* text = nullptr; size_t n = 0;
    int KW = 0; WKeyGen g;
    int L = 0; u32 F[3] = { 1, 1, 1 }; u32 os = 0, NLr = 0, NS = 0, S = 0;
    u64* K1[2] = { nullptr, nullptr }; u64* K2[2] = { nullptr, nullptr }; u32* V[2] = { nullptr, nullptr };
    u16* digits = nullptr; u64* sp1 = nullptr; u64* sp2 = nullptr;
    u32 nchunks = 0; size_t chunk_off[33] = {};   // chunk q = text positions [chunk_off[q], chunk_off[q + 1]) (multiples of the partition tile; the last one ends at n)
    u32* nstart_all = nullptr;       // [nchunks][F[0] + 1]: bucket starts of every chunk (absolute slots)
    const u32* dig2_counts[32] = {}; const u32* dig2_blk[32] = {}; u32 dig2_R[32] = {};   // ... and the tile histograms of that count pass, per chunk (rows of dig2_R tiles per block, block starts per bucket)
    bool dig2 = false; u32 dig2_chunks = 0;   // the digits of the merging level are computed chunk by chunk as well, for the first dig2_chunks chunks (its count pass only reads them)
    u32 present[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // byte values of the provisional code map
};
bool wsort_pre_begin(Ctx& c, WPre& P, const u8* text, size_t n, const size_t* chunk_off, u32 nchunks, const u32* hist0);   // false: not applicable
void wsort_pre_chunk(Ctx& c, WPre& P, u32 q);                  // level 1 of chunk q (the chunk and the 64 bytes behind it must have arrived)
void wsort_pre_finish(Ctx& c, WPre& P, const u32* hist_full);  // sets P.active if the full histogram confirms the code map
bool wsort_applicable(const Ctx& c, size_t n);
void wsort_make_keygen(const Ctx& c, const u8* text, size_t n, u32 sigma, const u8* code, int& KW, WKeyGen& g);   // key layout for an alphabet
int wsort_result_index(Ctx& c, size_t n);     // which of the two V buffers wsort_suffixes will return (so that the caller can make it sa[])
// Sorts all suffixes of the text by key(p).  K1 / K2 / V: two buffers each (n entries; K2 unused for KW = 1).  Returns x: V[x] holds the
// positions in sorted order; flags[i] = 1 where slot i starts a group of equal keys; lcp8[i] (valid where flags[i] = 1, i > 0) =
// number of leading symbols the keys of slot i - 1 and slot i have in common.  st->nonheads = number of slots with flags = 0.
int wsort_suffixes(Ctx& c, int KW, const WKeyGen& g, u64* K1[2], u64* K2[2], u32* V[2], size_t n, u8* flags, u8* lcp8, WSortStats* st);
// the same behind a completed WPre (levels 2 .. L + leaves); the sorted positions land in v_final
void wsort_suffixes_pre(Ctx& c, const WPre& P, u32* v_final, u8* flags, u8* lcp8, WSortStats* st);
// Sorts m records (K1[0][j], K2[0][j], V[0][j]) by (k1, k2); k1_bits = significant bits of k1.  Returns the buffer index of the result.
int wsort_records(Ctx& c, u64* K1[2], u64* K2[2], u32* V[2], size_t m, int k1_bits, WSortStats* st);
// Records whose first words are already in order (k1[j] non-decreasing): every run of equal first words is sorted by k2 in place (k2 and v
// move).  Runs of more members than one wave orders (1 024) go through wsort_records in a list of their own.  Returns false if there are more of
// them than its table holds (65 536) -- shorter runs may have been ordered by then, the records are still the same list: the caller sorts
// it as a whole.
bool wsort_sorted_runs(Ctx& c, const u64* k1, u64* k2, u32* v, size_t m, int k1_bits);
// the same for the records of a text round, made on the way: record i = (a_r1[i], the next g.s symbols of suffix a_sa[i] from text position
// a_sa[i] + h, a_sa[i]) lands in (k1, k2, v) -- whatever the result, the three arrays hold the whole list afterwards
bool wsort_sorted_runs_from_text(Ctx& c, const u32* a_sa, const u32* a_r1, size_t m, const WKeyGen& g, u32 h, u64* k1, u64* k2, u32* v, int k1_bits);

// Same contract as radix_sort_pairs_u64 for keys that are pairwise DISTINCT on the sorted bits (stability is then
// irrelevant): inputs of at most 2048 pairs are sorted by one workgroup in LDS (bitonic network), larger ones by the
// radix sort.  Used for the many tiny per-level sorts of the factorizer.
int sort_pairs_u64_distinct(Ctx& c, u64* keys[2], u32* vals[2], size_t n, int begin_bit, int end_bit);

// dst[idx[j]] = val[j], j < m, for pairwise distinct idx[j] < n_dst: a radix partition by the top 16 bits of idx (two
// 8-bit passes; only the first one if tmp_idx2 / tmp_val2 are null), then a scatter whose writes stay inside one small
// window per run.  idx / val are only read; the tmp arrays (m entries each) receive the partitioned pairs.
// permutation: idx[] holds every index of [0, m) exactly once (the caller's promise) -- the final pass then writes whole
// destination windows from an LDS image.
void bucketed_scatter_u32(Ctx& c, const u32* idx, const u32* val, size_t m, u32* dst, size_t n_dst, u32* tmp_idx, u32* tmp_val,
                          u32* tmp_idx2, u32* tmp_val2, bool permutation = false);

// Order-preserving selection: for the k (ascending) with cls[k] == want, outA[j] = srcA[k] (or k itself if srcA ==
// nullptr) and outB[j] = srcB[k] if srcB != nullptr; *d_count (device) receives the number of selected elements.
void select_by_class(Ctx& c, const u8* cls, u8 want, size_t m, const u32* srcA, u32* outA, const u64* srcB, u64* outB,
                     u32* d_count);

// Orbit of element 0 under a strictly increasing successor function: next[i] > i, next[i] == n ends the chain.
// mark[i] = 1 for every element on the chain 0, next[0], next[next[0]], ... ; 0 elsewhere.
// scratch1 / scratch2: n u32 each.  Hierarchical: exit of every element from its 1024-tile (pointer doubling in LDS),
// exit from its 2^20 super-tile (right-to-left sweep), a serial walk over the super-tile entries, then tile entries and
// chain elements in parallel.
void mark_orbit_u32(Ctx& c, const u32* next, size_t n, u8* mark, u32* scratch1, u32* scratch2);

// fill / iota helpers
void fill_u32(Ctx& c, u32* p, size_t n, u32 v);
void fill_u8(Ctx& c, u8* p, size_t n, u8 v);

}  // namespace tdc
