// huffman_host.cpp -- see huffman_host.hpp
#include "huffman_host.hpp"

#include <algorithm>
#include <numeric>
#include <cstring>

namespace tdc {

void build_huffman_table(const uint32_t C[256], HuffTable* t) {
    *t = HuffTable();
    size_t sigma = 0;
    uint8_t from_eff[256];
    for (int i = 0; i < 256; ++i) if (C[i]) from_eff[sigma++] = (uint8_t)i;          // HuffmanCoder.hpp:51-78
    t->sigma = (uint32_t)sigma;
    if (sigma <= 1) return;                                                           // :529-536 : no table

    // gen_codelengths :88-141 (Managing Gigabytes in-array Huffman)
    std::vector<size_t> Av(2 * sigma);
    size_t* A = Av.data();
    for (size_t i = 0; i < sigma; ++i) { A[sigma + i] = C[from_eff[i]]; A[i] = sigma + i; }
    auto comp = [A](const size_t a, const size_t b) -> bool { return A[a] > A[b]; };
    std::make_heap(&A[0], &A[sigma], comp);
    size_t h = sigma - 1;
    while (h > 0) {
        std::pop_heap(A, A + h + 1, comp);
        const size_t m1 = A[h];
        --h;
        std::pop_heap(A, A + h + 1, comp);
        const size_t m2 = A[h];
        A[h + 1] = A[m1] + A[m2];
        A[h] = h + 1;
        A[m1] = A[m2] = h + 1;
        std::push_heap(A, A + h + 1, comp);
    }
    A[1] = 0;
    for (size_t i = 2; i < 2 * sigma; ++i) A[i] = A[A[i]] + 1;
    uint8_t codelengths[256];
    for (size_t i = 0; i < sigma; ++i) codelengths[i] = (uint8_t)A[sigma + i];

    // gen_huffmantable :450-466 : same container type, same comparator signature as the reference
    std::vector<size_t> order(sigma);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.data(), order.data() + sigma,
              [&](const uint8_t& i, const uint8_t& j) { return codelengths[i] < codelengths[j]; });
    const uint8_t longest = *std::max_element(codelengths, codelengths + sigma);
    uint8_t ordered_len[256];
    for (size_t i = 0; i < sigma; ++i) { ordered_len[i] = codelengths[order[i]]; t->order[i] = from_eff[order[i]]; }
    t->longest = longest;
    for (size_t i = 0; i < sigma; ++i) ++t->numl[ordered_len[i] - 1];               // gen_numl :173-187 (u8 counters)
    uint64_t firstcode[256];                                                          // gen_first_codes :192-198
    firstcode[longest - 1] = 0;
    for (size_t i = longest - 1; i > 0; --i) firstcode[i - 1] = (firstcode[i] + t->numl[i]) / 2;
    for (size_t i = 0; i < sigma; ++i) {                                              // gen_codewords :202-218
        const uint8_t sym = t->order[i];
        t->len_of[sym] = ordered_len[i];
        t->code_of[sym] = firstcode[ordered_len[i] - 1]++;
    }
}

// Start-up self-check: the table depends on the tie behaviour of the C++ library's heap functions
// and of its unstable std::sort.  Two fixtures with many equal counts are rebuilt and compared with the tables the oracle's
// restatement of the reference build's behaviour yields (tools/make_huffman_selfcheck.py); a drifted library is refused.
#include "huffman_selfcheck.inc"
bool huffman_selfcheck() {
    for (const auto& f : HUFF_SELFCHECK) {
        uint32_t C[256];
        for (int i = 0; i < 256; ++i) C[i] = f.counts[i];
        HuffTable t;
        build_huffman_table(C, &t);
        if (t.sigma != f.sigma || t.longest != f.longest) return false;
        if (memcmp(t.order, f.order, f.sigma) != 0 || memcmp(t.len_of, f.len_of, 256) != 0) return false;
    }
    return true;
}

void write_huffman_header(HostBitWriter& w, const HuffTable& t) {
    if (t.sigma <= 1) { w.write_bit(0); return; }                                     // :538-540
    w.write_bit(1);                                                                   // :542
    w.write_compressed_int(t.longest);                                                // huffmantable_encode :264-273
    for (uint32_t i = 0; i < t.longest; ++i) w.write_compressed_int(t.numl[i]);
    w.write_compressed_int(t.sigma);
    for (uint32_t i = 0; i < t.sigma; ++i) w.write_int(t.order[i], 8);
}

}  // namespace tdc
