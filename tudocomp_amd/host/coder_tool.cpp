// coder_tool.cpp -- drives the host Encoder / Decoder classes of tdc_coders.hpp for the parity tests (tests/test_host_coders.py):
//   coder_tool encode huff|ascii TEXT FACTORS OUT     lzss::encode_text with coder_t::Encoder; FACTORS = u32 triples (pos, src, len)
//   coder_tool decode huff|ascii IN OUT               decode_text_internal with coder_t::Decoder (escaped text, sentinel included)
//   coder_tool gamma PAIRS OUT                        LZ78Compressor's coder calls with EliasGammaCoder::Encoder; PAIRS = u32 (id, char)
//   coder_tool ungamma IN OUT                         the pairs back (u32 id, u32 char), EliasGammaCoder::Decoder
#include "tdc_coders.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

using namespace tdc_amd;
using bytes = std::vector<uint8_t>;

static bytes slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::cerr << "cannot read " << path << "\n"; exit(2); }
    return bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const char* path, const bytes& b) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    f.write((const char*)b.data(), (std::streamsize)b.size());
}
static uint32_t u32at(const bytes& b, size_t i) { uint32_t v; memcpy(&v, b.data() + 4 * i, 4); return v; }

template <typename coder_t> static bytes do_encode(const bytes& text, const std::vector<Factor>& f) {
    bytes out;
    {
        auto bits = std::make_shared<BitOStream>(out);
        typename coder_t::Encoder coder(bits, TextLiterals(text.data(), text.size(), f));      // LCPCompressor.hpp:127-131
        encode_text(coder, text.data(), text.size(), f);
        bits->finish();
    }
    return out;
}
template <typename coder_t> static bytes do_decode(const bytes& in) {
    typename coder_t::Decoder decoder(std::make_shared<BitIStream>(in.data(), in.size()));
    bytes text;
    decode_text(decoder, text);
    return text;
}

int main(int argc, char** argv) {
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "encode" && argc == 6) {
            const bytes text = slurp(argv[3]), fb = slurp(argv[4]);
            std::vector<Factor> f(fb.size() / 12);
            for (size_t i = 0; i < f.size(); ++i) f[i] = Factor{u32at(fb, 3 * i), u32at(fb, 3 * i + 1), u32at(fb, 3 * i + 2)};
            spit(argv[5], std::string(argv[2]) == "ascii" ? do_encode<ASCIICoder>(text, f) : do_encode<HuffmanCoder>(text, f));
        } else if (mode == "decode" && argc == 5) {
            const bytes in = slurp(argv[3]);
            spit(argv[4], std::string(argv[2]) == "ascii" ? do_decode<ASCIICoder>(in) : do_decode<HuffmanCoder>(in));
        } else if (mode == "gamma" && argc == 4) {
            const bytes pb = slurp(argv[2]);
            bytes out;
            {
                auto bits = std::make_shared<BitOStream>(out);
                EliasGammaCoder::Encoder coder(bits, NoLiterals());
                size_t factor_count = 0;
                for (size_t i = 0; i + 1 < pb.size() / 4; i += 2) {                // LZ78Compressor.hpp:97-121
                    coder.encode(u32at(pb, i), Range(factor_count));
                    coder.encode((uliteral_t)u32at(pb, i + 1), literal_r);
                    ++factor_count;
                }
                bits->finish();
            }
            spit(argv[3], out);
        } else if (mode == "ungamma" && argc == 4) {
            const bytes in = slurp(argv[2]);
            EliasGammaCoder::Decoder decoder(std::make_shared<BitIStream>(in.data(), in.size()));
            bytes out;
            size_t factor_count = 0;
            while (!decoder.eof()) {
                const uint32_t id = decoder.decode<uint32_t>(Range(factor_count)), c = decoder.decode<uint32_t>(literal_r);
                const uint32_t pair[2] = { id, c };
                out.insert(out.end(), (const uint8_t*)pair, (const uint8_t*)pair + 8);
                ++factor_count;
            }
            spit(argv[3], out);
        } else {
            std::cerr << "usage: coder_tool encode|decode huff|ascii ... | gamma PAIRS OUT | ungamma IN OUT\n";
            return 2;
        }
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }
}
