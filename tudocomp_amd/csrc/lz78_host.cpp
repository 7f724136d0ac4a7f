// lz78_host.cpp -- the LZ78 parse of LZ78Compressor (compressors/LZ78Compressor.hpp:64-140), host side (g++, no HIP).
//
// The parse is sequential by nature (phrase k + 1 starts where phrase k ends and is matched against the dictionary of the first k
// phrases: DESIGN.md section 8), so it stays on the host; its output (ids, chars) is coded on the device (lz78.hip).  All trie
// back-ends of the reference yield identical factor ids by contract (test/lz78_trie_tests.cpp:61-100); this one is a hashed
// (parent, byte) -> child dictionary.
//
// Rounds 1-5 placed a node at hash(parent id, byte): a step down the trie could only be looked up once the step above it had
// returned the parent's id -- one DEPENDENT cache miss per text byte below the cached top of the trie (1 GB of text: a 4 GB table,
// ~4 misses of ~100 ns per phrase, 17-23 MB/s).  Round 6 places a node at the hash of the STRING it spells (a rolling hash of the
// phrase prefix, which needs no memory access at all): the slots of the next 16 depths of a phrase are all known -- and requested --
// before the first of them is looked at; the walk then verifies them one after the other against the exact key (parent id, byte),
// already in cache.  A phrase costs about one memory latency instead of one per byte.  Nothing is approximate: the hash only says
// where a node is kept (linear probing behind it), what is compared is (parent, byte) as before.
#include "stages_host.hpp"

#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <new>
#include <utility>

namespace tdc {
namespace {

struct PhraseTable {                 // open addressing; key = (parent << 8 | byte) + 1 (0: empty), value = child id
    struct Slot { uint64_t key; uint32_t val; uint32_t htop; };     // htop: upper half of the string hash (placement after a growth)
    struct Buf {
        Slot* p = nullptr; size_t n = 0;
        ~Buf() { free(p); }
        void alloc(size_t count) {
            free(p); p = nullptr; n = count;
            const size_t bytes = (count * sizeof(Slot) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            p = (Slot*)aligned_alloc((size_t)2 << 20, bytes);       // the table of a 1 GB input is gigabytes large and every miss lands on
            if (!p) throw std::bad_alloc();                         // a random page: huge pages where the kernel grants them
#ifdef MADV_HUGEPAGE
            (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
            memset(p, 0, count * sizeof(Slot));
        }
        void swap(Buf& o) { std::swap(p, o.p); std::swap(n, o.n); }
    } slots;
    uint64_t mask = 0;
    int shift = 64;                                                 // slot of a hash: h >> shift
    size_t used = 0;
    void init(size_t cap_pow2) {
        slots.alloc(cap_pow2); mask = cap_pow2 - 1; used = 0;
        shift = 64; for (size_t c = cap_pow2; c > 1; c >>= 1) --shift;
    }
    size_t home(uint64_t h) const { return shift == 64 ? 0 : (size_t)(h >> shift); }
    void grow() {
        Buf os; os.swap(slots);
        init((mask + 1) * 2);
        for (size_t i = 0; i < os.n; ++i) if (os.p[i].key) {
            size_t at = home((uint64_t)os.p[i].htop << 32);
            while (slots.p[at].key) at = (at + 1) & mask;
            slots.p[at] = os.p[i]; ++used;
        }
    }
};

// hash of a phrase prefix from the hash of the prefix one byte shorter (a function of the string alone)
inline uint64_t roll(uint64_t h, uint8_t c) {
    h = (h ^ ((uint64_t)c + 1)) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

}  // namespace

size_t lz78_parse_host(const uint8_t* in, size_t n, std::vector<uint32_t>& ids, std::vector<uint8_t>& chars, bool* leftover_is_high) {
    ids.clear(); chars.clear();
    if (leftover_is_high) *leftover_is_high = false;
    PhraseTable tab;
    size_t cap = 1024;
    while (cap < n / 4 + 16) cap <<= 1;
    if (cap > ((size_t)1 << 32)) cap = (size_t)1 << 32;            // (htop holds 32 placement bits)
    tab.init(cap);
    ids.reserve(n / 6 + 16); chars.reserve(n / 6 + 16);
    // depths requested ahead of the one being verified: a little more than a phrase is long (measured at 1e8 B of English-like text, 9.3
    // bytes per phrase: W = 10 -> 11 MB/s, 12 -> 27, 16 -> 25: a phrase that outruns its window waits for memory at every further
    // step, a window far beyond the phrase's end costs page walks for slots nobody looks at).  The window follows the running mean.
    constexpr size_t RING = 64;                                    // (a power of two > the largest window)
    size_t W = 12, nphr = 0, last_i = 0;
    uint32_t next_id = 1;                                          // root = 0, ids in insertion order from 1 (LZ78Compressor.hpp:78-84)
    size_t i = 0;
    uint32_t node = 0, parent = 0;
    uint8_t c = 0;
    uint64_t hs[RING];
    while (i < n) {                                                // one phrase per iteration (:97-121)
        if ((tab.used + 1) * 2 >= tab.mask) tab.grow();
        node = 0; parent = 0;
        uint64_t h = 0x243F6A8885A308D3ull;                        // hash of the empty prefix
        size_t pa = i;                                             // the prefixes text[i .. pa) have been hashed and their slots requested
        for (;;) {
            // (a sliding window: every verified depth requests one more, so a long phrase never stops to wait for a new batch)
            const size_t lim = (i + W < n) ? i + W : n;
            while (pa < lim) { h = roll(h, in[pa]); hs[pa % RING] = h; __builtin_prefetch(&tab.slots.p[tab.home(h)], 1, 0); ++pa; }
            if (i >= n) break;                                     // the text ended inside the dictionary: leftover phrase below
            c = in[i];
            const uint64_t hk = hs[i % RING];
            const uint64_t key = (((uint64_t)node << 8) | c) + 1;
            size_t at = tab.home(hk);
            PhraseTable::Slot* s = &tab.slots.p[at];
            while (s->key && s->key != key) { at = (at + 1) & tab.mask; s = &tab.slots.p[at]; }
            ++i;
            if (s->key) { parent = node; node = s->val; continue; }
            s->key = key; s->val = next_id++; s->htop = (uint32_t)(hk >> 32); ++tab.used;           // new phrase = matched node + c
            ids.push_back(node); chars.push_back(c);               // encode(node.id(), Range(factor_count)); encode(c, literal_r)  :101-102
            node = 0; parent = 0;
            if ((++nphr & 0xFFFFu) == 0) {                          // mean phrase length of the last 65 536 phrases + 3, within [8, 48]
                const size_t mean = (i - last_i + 0x8000u) >> 16;
                W = mean + 3 < 8 ? 8 : (mean + 3 > 48 ? 48 : mean + 3);
                last_i = i;
            }
            break;
        }
    }
    if (node != 0) {                                               // :124-131 leftover phrase: (parent.id(), c)
        ids.push_back(parent); chars.push_back(c);
        if (leftover_is_high && c >= 0x80) *leftover_is_high = true;   // the reference passes a signed char here
    }
    return ids.size();
}

}  // namespace tdc

extern "C" int tdc_lz78_factors(const uint8_t* in, size_t n, uint32_t** ids_out, uint8_t** chars_out, size_t* z_out) {
    if ((!in && n) || !ids_out || !chars_out || !z_out) return -2;
    *ids_out = nullptr; *chars_out = nullptr; *z_out = 0;
    try {
        std::vector<uint32_t> ids; std::vector<uint8_t> chars;
        const size_t z = tdc::lz78_parse_host(in, n, ids, chars, nullptr);
        uint32_t* a = (uint32_t*)malloc((z ? z : 1) * sizeof(uint32_t));
        uint8_t* b = (uint8_t*)malloc(z ? z : 1);
        if (!a || !b) { free(a); free(b); return -5; }
        if (z) { memcpy(a, ids.data(), z * sizeof(uint32_t)); memcpy(b, chars.data(), z); }
        *ids_out = a; *chars_out = b; *z_out = z;
    } catch (...) { return -5; }
    return 0;
}
