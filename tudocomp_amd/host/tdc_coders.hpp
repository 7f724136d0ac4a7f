// tdc_coders.hpp -- host-side mirror of tudocomp's Coder plugin surface for the lcpcomp / lz78 path.
//
//   tdc::Range, MinDistributedRange, TypeRange, FixedRange, LiteralRange, LengthRange, BitRange, literal_r / bit_r / len_r
//                                         include/tudocomp/Range.hpp:7-119          -> tdc_amd::Range ...
//   tdc::BitOStream                       include/tudocomp/io/BitOStream.hpp:17-164 -> tdc_amd::BitOStream
//   tdc::Encoder / tdc::Decoder           include/tudocomp/Coder.hpp:14-151         -> tdc_amd::Encoder / tdc_amd::Decoder
//   HuffmanCoder::{Encoder,Decoder}       coders/HuffmanCoder.hpp:521-613           -> tdc_amd::HuffmanCoder::{Encoder,Decoder}
//   EliasGammaCoder::{Encoder,Decoder}    coders/EliasGammaCoder.hpp:20-43          -> tdc_amd::EliasGammaCoder::...
//   ASCIICoder::{Encoder,Decoder}         coders/ASCIICoder.hpp:26-84               -> tdc_amd::ASCIICoder::...
//   lzss::encode_text / decode_text_internal  compressors/lzss/LZSSCoding.hpp:18-92, LCPCompressor.hpp:23-76
//
// As in the reference, overload resolution on the STATIC type of the range tag selects the literal coder (LiteralRange ->
// Huffman code) or the default binary coding (Range -> v - min in bits_for(max - min) bits, BitRange -> one bit).
// The GPU path never calls these per symbol (the whole token stream is produced on the device); they serve decompress(), the
// parity tests of the token format (tests/test_host_coders.py: host Encoder stream == oracle stream == device stream) and
// inputs too small to be worth a launch.  The Huffman table comes from tdc_huffman_table() of the C ABI, i.e. from the same
// std:: calls as the device path (huffman_host.cpp).
#pragma once

#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tdc_gpu.h"

namespace tdc_amd {

using uliteral_t = uint8_t;
using len_t = uint32_t;                                     // def.hpp:103

// ---- Range.hpp ---------------------------------------------------------------------------------------------------
class Range {
    size_t m_min, m_max;
public:
    constexpr Range(size_t max) : m_min(0), m_max(max) {}
    constexpr Range(size_t min, size_t max) : m_min(min), m_max(max) {}
    size_t min() const { return m_min; }
    size_t max() const { return m_max; }
    size_t delta() const { return m_max - m_min; }
};
class MinDistributedRange : public Range {
public:
    constexpr MinDistributedRange(size_t max) : Range(0, max) {}
    constexpr MinDistributedRange(size_t min, size_t max) : Range(min, max) {}
};
template <typename T> class TypeRange : public Range {
public:
    constexpr TypeRange() : Range(0, std::numeric_limits<T>::max()) {}
};
template <size_t t_min, size_t t_max> class FixedRange : public Range {
public:
    constexpr FixedRange() : Range(t_min, t_max) {}
};
class LiteralRange : public TypeRange<uliteral_t> { public: constexpr LiteralRange() {} };
class LengthRange : public TypeRange<len_t> { public: constexpr LengthRange() {} };
using BitRange = FixedRange<0, 1>;
constexpr auto bit_r = BitRange();
constexpr auto literal_r = LiteralRange();
constexpr auto len_r = LengthRange();

inline unsigned coder_bits_for(uint64_t v) { unsigned b = 0; if (!v) return 1; while (v) { ++b; v >>= 1; } return b; }   // util.hpp:194

// ---- io/BitOStream.hpp: MSB-first bit writer; the destructor (here: finish()) appends the 3-bit terminator (:53-64) ---
class BitOStream {
    std::vector<uint8_t>* m_sink;
    uint8_t m_next = 0;
    int m_cursor = 7;
    bool m_dirty = false, m_finished = false;
    void write_next() { if (m_dirty) { m_sink->push_back(m_next); m_next = 0; m_cursor = 7; m_dirty = false; } }
public:
    explicit BitOStream(std::vector<uint8_t>& sink) : m_sink(&sink) {}
    BitOStream(const BitOStream&) = delete;
    ~BitOStream() { finish(); }
    void finish() {
        if (m_finished) return;
        m_finished = true;
        const uint8_t set = (uint8_t)(7 - m_cursor);
        if (m_cursor >= 2) m_next |= set;
        else { write_next(); m_next = set; }
        m_dirty = true;
        write_next();
    }
    void write_bit(bool set) {
        if (set) m_next |= (uint8_t)(1u << m_cursor);
        m_dirty = true;
        if (--m_cursor < 0) write_next();
    }
    template <typename T> void write_int(T value, size_t bits = sizeof(T) * 8) {
        for (int i = (int)bits - 1; i >= 0; --i) write_bit(i < 64 ? (((uint64_t)value >> i) & 1u) != 0 : false);
    }
    template <typename T> void write_compressed_int(T v, size_t b = 7) {      // :150-163
        uint64_t x = (uint64_t)v;
        do {
            const uint64_t cur = x;
            x >>= b;
            write_bit(x > 0);
            write_int(cur, b);
        } while (x > 0);
    }
    template <typename T> void write_unary(T v) { uint64_t x = (uint64_t)v; while (x--) write_bit(0); write_bit(1); }
    template <typename T> void write_elias_gamma(T v) { write_unary(coder_bits_for((uint64_t)v)); write_int((uint64_t)v, coder_bits_for((uint64_t)v)); }
};

// ---- io/BitIStream.hpp:16-195 ------------------------------------------------------------------------------------
class BitIStream {
    const uint8_t* m_p; size_t m_n, m_idx = 0;
    uint8_t m_current = 0, m_next = 0, m_final_bits = 0, m_cursor = 0;
    bool m_is_final = false;
    void read_next() {
        m_current = m_next; m_cursor = 7;
        if (m_idx < m_n) {
            m_next = m_p[m_idx++];
            if (m_idx == m_n) { m_final_bits = m_next & 7; if (m_final_bits >= 6) { m_is_final = true; m_next = 0; } }
        } else { m_is_final = true; m_final_bits = m_current & 7; m_next = 0; }
    }
public:
    BitIStream(const uint8_t* p, size_t n) : m_p(p), m_n(n) {
        if (n) { m_next = m_p[m_idx++]; read_next(); } else { m_is_final = true; }
    }
    bool eof() const { return m_is_final && m_cursor <= (7 - m_final_bits); }
    size_t size_bytes() const { return m_n; }
    unsigned read_bit() {
        if (eof()) return 0;
        unsigned bit = (m_current >> m_cursor) & 1;
        if (m_cursor) --m_cursor; else read_next();
        return bit;
    }
    uint64_t read_int(unsigned bits) { uint64_t v = 0; while (bits--) v = (v << 1) | read_bit(); return v; }
    template <typename T> T read_int() { return (T)read_int(sizeof(T) * 8); }
    uint64_t read_compressed_int(unsigned b = 7) {
        uint64_t v = 0; unsigned i = 0; bool more;
        do { more = read_bit(); v |= read_int(b) << (b * i++); } while (more);
        return v;
    }
    uint64_t read_unary() { uint64_t v = 0; while (!read_bit()) { if (eof() || ++v > 64) throw std::runtime_error("corrupt unary code"); } return v; }
    uint64_t read_elias_gamma() { const unsigned b = (unsigned)read_unary(); return read_int(b); }
};

// ---- Coder.hpp:14-151 -----------------------------------------------------------------------------------------------
class Encoder {
protected:
    std::shared_ptr<BitOStream> m_out;
public:
    template <typename literals_t> Encoder(std::shared_ptr<BitOStream> out, literals_t&&) : m_out(std::move(out)) {}
    template <typename value_t> void encode(value_t v, const Range& r) { m_out->write_int((uint64_t)v - r.min(), coder_bits_for(r.max() - r.min())); }
    template <typename value_t> void encode(value_t v, const BitRange&) { m_out->write_bit(v != 0); }
    const std::shared_ptr<BitOStream>& stream() { return m_out; }
};
class Decoder {
protected:
    std::shared_ptr<BitIStream> m_in;
public:
    explicit Decoder(std::shared_ptr<BitIStream> in) : m_in(std::move(in)) {}
    bool eof() const { return m_in->eof(); }
    template <typename value_t> value_t decode(const Range& r) { return (value_t)(r.min() + m_in->read_int(coder_bits_for(r.max() - r.min()))); }
    template <typename value_t> value_t decode(const BitRange&) { return (value_t)m_in->read_bit(); }
    const std::shared_ptr<BitIStream>& stream() { return m_in; }
};

// literal iterators (Literal.hpp: has_next() / next() yielding {c, pos})
struct Literal { uliteral_t c; size_t pos; };
struct NoLiterals { bool has_next() const { return false; } Literal next() { return {0, 0}; } };
// lzss::TextLiterals (compressors/lzss/LZSSLiterals.hpp:10-50): the positions no factor covers, in text order
struct Factor { len_t pos, src, len; };
class TextLiterals {
    const uint8_t* m_text; size_t m_n; const std::vector<Factor>* m_f; size_t m_pos = 0, m_next = 0;
    void skip() { while (m_next < m_f->size() && m_pos == (*m_f)[m_next].pos) { m_pos += (*m_f)[m_next].len; ++m_next; } }
public:
    TextLiterals(const uint8_t* text, size_t n, const std::vector<Factor>& f) : m_text(text), m_n(n), m_f(&f) { skip(); }
    bool has_next() const { return m_pos < m_n; }
    Literal next() { const Literal l{m_text[m_pos], m_pos}; ++m_pos; skip(); return l; }
};

// ---- coders/HuffmanCoder.hpp:521-613 ------------------------------------------------------------------------------
struct HuffmanCoder {
    class Encoder : public tdc_amd::Encoder {
        uint32_t m_sigma = 0;
        uint8_t m_len[256] = {0};
        uint64_t m_code[256] = {0};
    public:
        template <typename literals_t> Encoder(std::shared_ptr<BitOStream> out, literals_t&& literals) : tdc_amd::Encoder(out, NoLiterals()) {
            uint32_t C[256] = {0};                                           // huff::count_alphabet_literals :37-48
            while (literals.has_next()) ++C[literals.next().c];
            uint32_t longest = 0; uint8_t order[256];
            if (tdc_huffman_table(C, &m_sigma, &longest, order, m_len, m_code)) throw std::runtime_error("tdc_huffman_table failed");
            if (m_sigma <= 1) { m_out->write_bit(0); return; }               // :538-540
            m_out->write_bit(1);                                             // :542 + huffmantable_encode :264-273
            uint8_t numl[256] = {0};
            for (int s = 0; s < 256; ++s) if (m_len[s]) ++numl[m_len[s] - 1];
            m_out->write_compressed_int(longest);
            for (uint32_t i = 0; i < longest; ++i) m_out->write_compressed_int(numl[i]);
            m_out->write_compressed_int(m_sigma);
            for (uint32_t i = 0; i < m_sigma; ++i) m_out->write_int(order[i], 8);
        }
        using tdc_amd::Encoder::encode;                                      // default encoding as fallback
        template <typename value_t> void encode(value_t v, const LiteralRange&) {      // :562-569
            const uint8_t c = (uint8_t)v;
            if (m_sigma <= 1) m_out->write_int(c, 8);
            else m_out->write_int(m_code[c], m_len[c]);
        }
    };
    class Decoder : public tdc_amd::Decoder {
        bool m_table = false;
        uint8_t m_order[256]; uint64_t m_first[256]; size_t m_prefix[256]; uint8_t m_numl[256]; unsigned m_longest = 0; size_t m_sigma = 0;
    public:
        explicit Decoder(std::shared_ptr<BitIStream> in) : tdc_amd::Decoder(std::move(in)) {   // :581-597
            m_table = m_in->read_bit();
            if (!m_table) return;
            m_longest = (unsigned)(m_in->read_compressed_int() & 0xFF);
            if (!m_longest) throw std::runtime_error("corrupt Huffman table");
            for (unsigned i = 0; i < m_longest; ++i) m_numl[i] = (uint8_t)m_in->read_compressed_int();
            m_sigma = m_in->read_compressed_int();
            if (m_sigma > 256) throw std::runtime_error("corrupt Huffman table");
            for (size_t i = 0; i < m_sigma; ++i) m_order[i] = (uint8_t)m_in->read_int(8);
            m_first[m_longest - 1] = 0;                                      // gen_first_codes :192-198
            for (unsigned i = m_longest - 1; i > 0; --i) m_first[i - 1] = (m_first[i] + m_numl[i]) / 2;
            size_t acc = 0;                                                  // gen_prefix_sum_lengths :350-370
            for (unsigned l = 0; l < m_longest; ++l) { m_prefix[l] = acc; acc += m_numl[l]; }
        }
        using tdc_amd::Decoder::decode;
        template <typename value_t> value_t decode(const LiteralRange&) {     // :606-611, huffman_decode :377-397
            if (!m_table) return (value_t)m_in->read_int(8);
            uint64_t value = 0; unsigned length = 0;
            do { value = (value << 1) + m_in->read_bit(); ++length; } while (length <= m_longest && value < m_first[length - 1]);
            if (length > m_longest) throw std::runtime_error("corrupt Huffman code");
            --length;
            const uint64_t off = value - m_first[length];
            if (off >= m_numl[length] || m_prefix[length] + off >= m_sigma) throw std::runtime_error("corrupt Huffman code");   // a table that violates Kraft
            return (value_t)m_order[m_prefix[length] + off];
        }
    };
};

// ---- coders/EliasGammaCoder.hpp:20-43 -------------------------------------------------------------------------------
struct EliasGammaCoder {
    class Encoder : public tdc_amd::Encoder {
    public:
        using tdc_amd::Encoder::Encoder;
        using tdc_amd::Encoder::encode;
        template <typename value_t> void encode(value_t v, const Range&) { m_out->write_elias_gamma((uint64_t)v); }
    };
    class Decoder : public tdc_amd::Decoder {
    public:
        using tdc_amd::Decoder::Decoder;
        using tdc_amd::Decoder::decode;
        template <typename value_t> value_t decode(const Range&) { return (value_t)m_in->read_elias_gamma(); }
    };
};

// ---- coders/ASCIICoder.hpp:26-84 -------------------------------------------------------------------------------------
struct ASCIICoder {
    class Encoder : public tdc_amd::Encoder {
    public:
        using tdc_amd::Encoder::Encoder;
        template <typename value_t> void encode(value_t v, const Range&) {
            const std::string s = std::to_string((uint64_t)v);
            for (uint8_t c : s) m_out->write_int(c, 8);
            m_out->write_int((uint8_t)':', 8);
        }
        template <typename value_t> void encode(value_t v, const LiteralRange&) { m_out->write_int((uint8_t)v, 8); }
        template <typename value_t> void encode(value_t v, const BitRange&) { m_out->write_int((uint8_t)(v ? '1' : '0'), 8); }
    };
    class Decoder : public tdc_amd::Decoder {
    public:
        using tdc_amd::Decoder::Decoder;
        template <typename value_t> value_t decode(const Range&) {
            uint64_t v = 0; int digits = 0;
            for (uint8_t c = (uint8_t)m_in->read_int(8); c >= '0' && c <= '9'; c = (uint8_t)m_in->read_int(8)) {
                v = v * 10 + (c - '0'); ++digits;
                if (m_in->eof()) break;
            }
            if (!digits) throw std::runtime_error("corrupt stream: integer expected");
            return (value_t)v;
        }
        template <typename value_t> value_t decode(const LiteralRange&) { return (value_t)m_in->read_int(8); }
        template <typename value_t> value_t decode(const BitRange&) { return (value_t)((uint8_t)m_in->read_int(8) != '0'); }
    };
};

// ---- lzss::encode_text (compressors/lzss/LZSSCoding.hpp:18-92): factors sorted by pos -------------------------------
template <typename coder_t>
inline void encode_text(coder_t& coder, const uint8_t* text, size_t n, const std::vector<Factor>& factors) {
    size_t flen_min = std::numeric_limits<len_t>::max(), flen_max = 0, fdist_max = 0;      // FactorBuffer :41-47 (empty: max / 0)
    {
        size_t p = 0;
        for (const Factor& f : factors) {
            if (f.len < flen_min) flen_min = f.len;
            if (f.len > flen_max) flen_max = f.len;
            if (f.pos - p > fdist_max) fdist_max = f.pos - p;
            p = (size_t)f.pos + f.len;
        }
        if (n - p > fdist_max) fdist_max = n - p;
    }
    if (factors.empty()) { flen_min = std::numeric_limits<len_t>::max(); flen_max = 0; }
    const Range text_r(n);
    const MinDistributedRange flen_r(flen_min, flen_max);
    const Range fdist_r(fdist_max);
    coder.encode(n, len_r);
    coder.encode(flen_min, text_r);
    coder.encode(flen_max, text_r);
    coder.encode(fdist_max, text_r);
    size_t p = 0;
    for (const Factor& f : factors) {
        if (f.pos == p) coder.encode(false, bit_r);
        else { coder.encode(true, bit_r); coder.encode(f.pos - p, fdist_r); }
        while (p < f.pos) coder.encode(text[p++], literal_r);
        coder.encode(f.src, text_r);
        coder.encode(f.len, flen_r);
        p += (size_t)f.len;
    }
    if (p < n) { coder.encode(true, bit_r); coder.encode(n - p, fdist_r); }
    while (p < n) coder.encode(text[p++], literal_r);
}

// ---- lcpcomp::decode_text_internal (LCPCompressor.hpp:23-76): token stream -> text + reference forest --------------
// The decoded text is unique, so the references are resolved by following source chains instead of the reference's
// ScanDec buffers (forward and backward references alike: lcpcomp and lzss_lcp).
template <typename decoder_t>
inline void decode_text(decoder_t& decoder, std::vector<uint8_t>& text) {
    const size_t n = decoder.template decode<size_t>(len_r);
    if (n >= 0x7FFFFFFFull) throw std::runtime_error("corrupt stream: text length");       // 32-bit len_t, texts stay below 2^31 - 1
    const Range text_r(n);
    const size_t flen_min = decoder.template decode<size_t>(text_r);
    const size_t flen_max = decoder.template decode<size_t>(text_r);
    const size_t fdist_max = decoder.template decode<size_t>(text_r);
    {   // plausibility (corrupt headers would otherwise ask for gigabytes): a literal costs at least one bit, a factor at least
        // bits_for(n) and covers at most flen_max positions
        const size_t bits = decoder.stream()->size_bytes() * 8;
        if (n > bits + (bits / coder_bits_for(n) + 1) * (flen_max ? flen_max : 1)) throw std::runtime_error("corrupt stream: text length");
    }
    const MinDistributedRange flen_r(flen_min, flen_max >= flen_min ? flen_max : flen_min);
    const Range fdist_r(fdist_max);
    text.assign(n, 0);
    std::vector<uint32_t> ref(n, 0xFFFFFFFFu);
    size_t p = 0;
    while (!decoder.eof()) {
        size_t num = decoder.template decode<bool>(bit_r) ? decoder.template decode<size_t>(fdist_r) : 0;
        if (p + num > n) throw std::runtime_error("corrupt stream: too many literals");
        while (num--) text[p++] = decoder.template decode<uliteral_t>(literal_r);
        if (!decoder.eof()) {
            const size_t src = decoder.template decode<size_t>(text_r), len = decoder.template decode<size_t>(flen_r);
            if (len == 0 || p + len > n || src + len > n) throw std::runtime_error("corrupt stream: factor out of range");
            for (size_t j = 0; j < len; ++j) ref[p + j] = (uint32_t)(src + j);
            p += len;
        }
    }
    if (p != n) throw std::runtime_error("corrupt stream: length mismatch");
    std::vector<uint32_t> stack;
    for (size_t i = 0; i < n; ++i) {
        if (ref[i] == 0xFFFFFFFFu) continue;
        stack.clear();
        uint32_t q = (uint32_t)i;
        while (ref[q] != 0xFFFFFFFFu) {
            if (stack.size() > n) throw std::runtime_error("corrupt stream: reference cycle");
            stack.push_back(q); q = ref[q];
        }
        for (uint32_t r : stack) { text[r] = text[q]; ref[r] = 0xFFFFFFFFu; }
    }
}

}  // namespace tdc_amd
