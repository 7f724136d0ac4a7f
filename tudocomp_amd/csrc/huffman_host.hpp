// huffman_host.hpp -- canonical Huffman table + stream header, built on the host (sigma <= 256, microseconds).
//
// Follows coders/HuffmanCoder.hpp of the reference: gen_codelengths :88-169, the std::sort at :455, gen_numl :173,
// gen_first_codes :192-198, gen_codewords :202-218, huffmantable_encode :264-273 and the Encoder ctor :526-547.
// Code lengths depend on the tie behaviour of libstdc++'s heap algorithms and the symbol order on its (unstable)
// std::sort, so exactly those std:: calls are made here, on the same initial data.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace tdc {

struct HuffTable {
    uint32_t sigma = 0;             // effective alphabet size
    uint32_t longest = 0;
    uint8_t  numl[256] = {0};
    uint8_t  order[256] = {0};      // symbols in canonical order
    uint8_t  len_of[256] = {0};     // code length per byte value (0 = absent)
    uint64_t code_of[256] = {0};    // code word per byte value
};

// MSB-first bit writer (io/BitOStream.hpp:79-163) for the few header bits written on the host.
struct HostBitWriter {
    std::vector<uint8_t> bytes;
    uint64_t nbits = 0;
    void write_bit(bool b) {
        if ((nbits & 7) == 0) bytes.push_back(0);
        if (b) bytes.back() |= (uint8_t)(0x80u >> (nbits & 7));
        ++nbits;
    }
    void write_int(uint64_t v, unsigned bits) {
        for (int i = (int)bits - 1; i >= 0; --i) write_bit(i < 64 ? ((v >> i) & 1) : 0);
    }
    void write_compressed_int(uint64_t v, unsigned b = 7) {
        do {
            const uint64_t cur = v;
            v >>= b;
            write_bit(v > 0);
            write_int(cur, b);
        } while (v > 0);
    }
};

void build_huffman_table(const uint32_t C[256], HuffTable* t);
// true iff this build's C++ library reproduces the reference's tables on the built-in fixtures (checked when a context is created)
bool huffman_selfcheck();
// HuffmanCoder::Encoder ctor: "0" if sigma <= 1, else "1" + table
void write_huffman_header(HostBitWriter& w, const HuffTable& t);

}  // namespace tdc
