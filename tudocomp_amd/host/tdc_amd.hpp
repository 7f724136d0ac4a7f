// tdc_amd.hpp -- host-side C++ mirror of the slice of tudocomp's plugin surface that the lcpcomp hot path needs.
//
// Same names, argument meaning and error behaviour as the reference, so that its tests and driver read the same:
//   tdc::Compressor          include/tudocomp/Compressor.hpp:19-43        -> tdc_amd::Compressor
//   tdc::LCPCompressor       include/tudocomp/compressors/LCPCompressor.hpp:79-151 -> tdc_amd::LCPCompressor
//   Input / Output wrapping  include/tudocomp/io/{Input,Output}.hpp, restrictions "escape {0} + null-terminate"
//   Registry                 include/tudocomp/pre_header/Registry.hpp:11-25,204-231 (parse_algorithm_id / select_algorithm)
// compress() forwards to the C ABI (include/tdc_gpu.h) -- there is no CPU compress path.
// decompress() is host code, like in the reference (decode_text_internal + HuffmanCoder::Decoder, LCPCompressor.hpp:23-76,
// coders/HuffmanCoder.hpp:572-612, io/BitIStream.hpp); the decoded text is unique, so references are resolved by
// following source chains instead of the reference's ScanDec buffers.
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tdc_gpu.h"
#include "tdc_coders.hpp"

namespace tdc_amd {

using bytes = std::vector<uint8_t>;

// ---- Input / Output ---------------------------------------------------------------------------------------
struct InputRestrictions {
    bool escape_zero = false;       // escape {0}        (ds/SADivSufSort.hpp:20-25)
    bool null_terminate = false;    // append sentinel
    bool has_restrictions() const { return escape_zero || null_terminate; }
};

class Input {
    bytes m_data;
    InputRestrictions m_restr;
public:
    Input() = default;
    explicit Input(bytes data) : m_data(std::move(data)) {}
    static Input from_memory(const void* p, size_t n) { return Input(bytes((const uint8_t*)p, (const uint8_t*)p + n)); }
    static Input from_file(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("Could not open file for reading: " + path);
        return Input(bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()));
    }
    Input(const Input& other, InputRestrictions r) : m_data(other.m_data), m_restr(r) {}
    Input(const Input& other, size_t from) : m_data(other.m_data.begin() + from, other.m_data.end()), m_restr(other.m_restr) {}
    size_t size() const { return m_data.size(); }
    const bytes& raw() const { return m_data; }
    // Input::as_view(): materialises the restricted view (io/RestrictedBuffer.hpp:108-254)
    bytes as_view() const {
        if (!m_restr.has_restrictions()) return m_data;
        bytes out(2 * m_data.size() + 1);
        out.resize(tdc_escape(m_data.data(), m_data.size(), out.data()));
        return out;
    }
};

class Output {
    bytes* m_vec = nullptr;
    InputRestrictions m_restr;
public:
    Output() = default;
    explicit Output(bytes& v) : m_vec(&v) {}
    Output(const Output& other, InputRestrictions r) : m_vec(other.m_vec), m_restr(r) {}
    // bytes that are already unrestricted (the blocks of a container remove their own restrictions)
    void write_plain(const uint8_t* p, size_t n) { m_vec->insert(m_vec->end(), p, p + n); }
    void write(const uint8_t* p, size_t n) {
        if (m_restr.has_restrictions()) {                       // un-escaping ostream filter (io/RestrictedIOStream.hpp:13-89)
            bytes tmp(n + 1);
            tmp.resize(tdc_unescape(p, n, tmp.data()));
            m_vec->insert(m_vec->end(), tmp.begin(), tmp.end());
        } else m_vec->insert(m_vec->end(), p, p + n);
    }
};

// ---- options ------------------------------------------------------------------------------------------------
struct AlgorithmValue {
    std::string name;
    std::map<std::string, std::string> args;      // key -> value text ("huff", "2", "scan(scans=6)")
    std::string get(const std::string& k, const std::string& dflt) const {
        auto it = args.find(k);
        return it == args.end() ? dflt : it->second;
    }
    long get_int(const std::string& k, long dflt) const {
        auto it = args.find(k);
        return it == args.end() ? dflt : std::stol(it->second);
    }
};

// name(arg, key=value, key=algo(...))  -- the subset of AlgorithmStringParser.hpp:94-300 this path needs
inline AlgorithmValue parse_algorithm_id(const std::string& s, const std::vector<std::string>& positional = {}) {
    AlgorithmValue av;
    size_t i = 0;
    auto skip = [&] { while (i < s.size() && isspace((unsigned char)s[i])) ++i; };
    skip();
    size_t b = i;
    while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_')) ++i;
    av.name = s.substr(b, i - b);
    if (av.name.empty()) throw std::runtime_error("Expected an algorithm name in '" + s + "'");
    skip();
    if (i == s.size()) return av;
    if (s[i] != '(') throw std::runtime_error("Unexpected character in algorithm string '" + s + "'");
    ++i;
    size_t pos_idx = 0;
    while (true) {
        skip();
        if (i < s.size() && s[i] == ')') { ++i; break; }
        size_t st = i; int depth = 0;
        while (i < s.size() && (depth > 0 || (s[i] != ',' && s[i] != ')'))) {
            if (s[i] == '(') ++depth;
            if (s[i] == ')') --depth;
            ++i;
        }
        if (i >= s.size()) throw std::runtime_error("Unbalanced parentheses in algorithm string '" + s + "'");
        std::string item = s.substr(st, i - st);
        while (!item.empty() && isspace((unsigned char)item.back())) item.pop_back();
        size_t eq = std::string::npos; depth = 0;
        for (size_t j = 0; j < item.size(); ++j) {
            if (item[j] == '(') ++depth; else if (item[j] == ')') --depth;
            else if (item[j] == '=' && depth == 0) { eq = j; break; }
        }
        if (eq == std::string::npos) {
            if (pos_idx >= positional.size()) throw std::runtime_error("Too many positional arguments in '" + s + "'");
            av.args[positional[pos_idx++]] = item;
        } else {
            std::string k = item.substr(0, eq), v = item.substr(eq + 1);
            while (!k.empty() && isspace((unsigned char)k.back())) k.pop_back();
            while (!v.empty() && isspace((unsigned char)v.front())) v.erase(v.begin());
            av.args[k] = v;
        }
        if (s[i] == ',') ++i;
    }
    return av;
}

inline unsigned bits_for(uint64_t v) { return coder_bits_for(v); }

// ---- Compressor ---------------------------------------------------------------------------------------------
class Compressor {
public:
    virtual ~Compressor() = default;
    virtual void compress(Input& input, Output& output) = 0;      // Compressor.hpp:36
    virtual void decompress(Input& input, Output& output) = 0;    // Compressor.hpp:42
    virtual InputRestrictions input_restrictions() const { return {}; }
};

struct GpuContext {
    tdc_gpu_ctx* h = nullptr;
    explicit GpuContext(int device = 0) {
        int rc = tdc_gpu_ctx_create(device, &h);
        if (rc) throw std::runtime_error(std::string("tdc_gpu_ctx_create: ") + tdc_gpu_strerror(rc));
    }
    ~GpuContext() { tdc_gpu_ctx_destroy(h); }
    GpuContext(const GpuContext&) = delete;
    GpuContext& operator=(const GpuContext&) = delete;
};

// coder_t::Decoder + lzss token stream (decode_text_internal / lzss::decode_text), shared by lcpcomp and lzss_lcp: the
// Decoder classes of tdc_coders.hpp (HuffmanCoder::Decoder, coders/HuffmanCoder.hpp:572-612; ASCIICoder::Decoder,
// coders/ASCIICoder.hpp:53-84).  References may point forwards (lcpcomp) or backwards (lzss_lcp).
template <typename coder_t>
inline void lzss_decode(Input& input, Output& output) {
    const bytes& in = input.raw();
    typename coder_t::Decoder decoder(std::make_shared<BitIStream>(in.data(), in.size()));
    bytes text;
    decode_text(decoder, text);
    output.write(text.data(), text.size());
}
inline void lzss_huff_decode(Input& input, Output& output) { lzss_decode<HuffmanCoder>(input, output); }
inline void lzss_ascii_decode(Input& input, Output& output) { lzss_decode<ASCIICoder>(input, output); }

// ---- block container (written by tdc_gpu_blocks_compress / tudocomp_amd.blocks): "tdcgpu-blocks%" | u32 G |
//      G x { u64 raw_len, u64 comp_len } | payloads, little endian.  Every payload is a complete stream of its block.
struct BlockContainer {
    struct Block { uint64_t raw_len; const uint8_t* data; size_t len; };
    static constexpr const char* MAGIC = "tdcgpu-blocks%";
    static bool is_container(const uint8_t* p, size_t n) { return n >= 14 && std::memcmp(p, MAGIC, 14) == 0; }
    static std::vector<Block> parse(const uint8_t* p, size_t n) {
        if (!is_container(p, n) || n < 18) throw std::runtime_error("not a tdcgpu-blocks container");
        auto u32at = [&](size_t o) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)p[o + i] << (8 * i); return v; };
        auto u64at = [&](size_t o) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[o + i] << (8 * i); return v; };
        const size_t G = u32at(14);
        if (n < 18 + 16 * G) throw std::runtime_error("corrupt container: directory");
        std::vector<Block> out;
        size_t at = 18 + 16 * G;
        for (size_t k = 0; k < G; ++k) {
            const uint64_t raw = u64at(18 + 16 * k), comp = u64at(18 + 16 * k + 8);
            if (comp > n - at) throw std::runtime_error("corrupt container: payload exceeds the file");
            out.push_back(Block{raw, p + at, (size_t)comp});
            at += comp;
        }
        if (at != n) throw std::runtime_error("corrupt container: trailing bytes");
        return out;
    }
};

// SLECoder::Decoder (coders/SLECoder.hpp:301-453) + the same token stream: ranking header, rank class codes, k-mer
// symbols expand to k literals; the factor length is a MinDistributedRange (:413-431).
inline void lzss_sle_decode(Input& input, Output& output, unsigned k) {
    const bytes& in = input.raw();
    BitIStream bs(in.data(), in.size());
    const size_t sigma = (size_t)bs.read_compressed_int();                   // Decoder ctor :325-340
    if (sigma == 0 || sigma > 4096) throw std::runtime_error("corrupt SLE ranking");
    const unsigned sb = bits_for(sigma - 1);
    std::vector<uint64_t> inv(sigma);
    for (size_t r = 0; r < sigma; ++r) inv[r] = bs.read_compressed_int();
    auto read_rank = [&]() -> uint64_t {                                      // :367-397
        if (sb < 4) return bs.read_int(sb);
        if (sb < 6) return bs.read_bit() ? bs.read_int(sb) : bs.read_int(2);
        if (sb == 6) {
            switch (bs.read_int(2)) {
                case 0: return bs.read_int(3);
                case 1: return 8 + bs.read_int(3);
                case 2: return 16 + bs.read_int(4);
                default: return bs.read_int(sb);
            }
        }
        const uint64_t cls = bs.read_int(3);
        if (cls < 4) return 4 * cls + bs.read_int(2);
        if (cls < 7) return 16 + 8 * (cls - 4) + bs.read_int(3);
        return bs.read_int(sb);
    };
    uint8_t kmer[8]; size_t kread = (size_t)-1;
    const uint64_t n = bs.read_int(32);
    const unsigned W = bits_for(n);
    const uint64_t flen_min = bs.read_int(W), flen_max = bs.read_int(W), fdist_max = bs.read_int(W);
    const unsigned lbits = bits_for(flen_max - flen_min), dbits = bits_for(fdist_max);
    bytes text(n);
    std::vector<uint32_t> ref(n, 0xFFFFFFFFu);
    uint64_t p = 0;
    auto eof = [&] { return kread < k ? false : bs.eof(); };                  // :351-359
    while (!eof()) {
        kread = (size_t)-1;                                                   // decode(BitRange) resets the k-mer :433-437
        uint64_t num = bs.read_bit() ? bs.read_int(dbits) : 0;
        while (num--) {
            uint8_t ch;
            if (kread < k) ch = kmer[kread++];
            else {
                const uint64_t r = read_rank();
                if (r >= sigma) throw std::runtime_error("corrupt stream: rank out of range");
                const uint64_t x = inv[r];
                if ((x >> 56) == 0xFF) { for (unsigned i = 0; i < k; ++i) kmer[k - 1 - i] = (uint8_t)(x >> (8 * i)); kread = 1; ch = kmer[0]; }
                else ch = (uint8_t)x;
            }
            if (p >= n) throw std::runtime_error("corrupt stream: too many literals");
            text[p++] = ch;
        }
        if (!eof()) {
            kread = (size_t)-1;
            const uint64_t src = bs.read_int(W);
            uint64_t v;
            if (lbits <= 5) v = bs.read_int(lbits);
            else switch (bs.read_int(2)) {
                case 0: v = bs.read_int(3); break;
                case 1: v = 8 + bs.read_int(3); break;
                case 2: v = 16 + bs.read_int(4); break;
                default: v = bs.read_int(lbits); break;
            }
            const uint64_t len = flen_min + v;
            if (len == 0 || p + len > n || src + len > n) throw std::runtime_error("corrupt stream: factor out of range");
            for (uint64_t j = 0; j < len; ++j) ref[p + j] = (uint32_t)(src + j);
            p += len;
        }
    }
    if (p != n) throw std::runtime_error("corrupt stream: length mismatch");
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t q = (uint32_t)i; uint64_t guard = 0;
        while (ref[q] != 0xFFFFFFFFu) { q = ref[q]; if (++guard > n) throw std::runtime_error("corrupt stream: reference cycle"); }
        text[i] = text[q];
    }
    output.write(text.data(), text.size());
}

class LCPCompressor : public Compressor {
    AlgorithmValue m_opts;
    std::shared_ptr<GpuContext> m_ctx;       // created lazily by the first compress(): decompress() needs no GPU
    int m_device = 0;
    int m_coder = TDC_GPU_CODER_HUFF;
    int m_comp = TDC_GPU_COMP_ARRAYS;
    unsigned m_kmer = 3;
public:
    tdc_gpu_stats last_stats{};
    void set_device(int d) { m_device = d; }
    long threshold() const { return m_opts.get_int("threshold", 5); }
    // meta: type "compressor", name "lcpcomp", options coder, comp=arrays, dec=scan, threshold=5, flatten=1 (LCPCompressor.hpp:85-95)
    LCPCompressor(AlgorithmValue opts, std::shared_ptr<GpuContext> ctx) : m_opts(std::move(opts)), m_ctx(std::move(ctx)) {
        const std::string coder = m_opts.get("coder", ""), comp = m_opts.get("comp", "arrays");
        // `arithmetic` is not in the reference's lcpcomp registry (etc/registry_config.py:138-142) but the template
        // instantiates; BASELINE.json configs[2] asks for it, compress side only
        m_comp = (comp == "plcppeaks" || comp == "plcppeaks()") ? TDC_GPU_COMP_PLCPPEAKS
               : (comp == "max_lcp" || comp == "max_lcp()") ? TDC_GPU_COMP_MAXLCP
               : (comp == "heap" || comp == "heap()") ? TDC_GPU_COMP_HEAP : TDC_GPU_COMP_ARRAYS;
        const AlgorithmValue cv = coder.empty() ? AlgorithmValue() : parse_algorithm_id(coder, {"kmer"});
        const std::string& cname = cv.name;
        if ((cname != "huff" && cname != "arithmetic" && cname != "ascii" && cname != "sle") ||
            (comp != "arrays" && comp != "arrays()" && comp != "plcppeaks" && comp != "plcppeaks()" && comp != "max_lcp" && comp != "max_lcp()" &&
             comp != "heap" && comp != "heap()"))
            throw std::runtime_error("No implementation found for compressor lcpcomp(coder=" + coder + ",comp=" + comp + ")");   // Registry.hpp:214
        m_coder = (cname == "huff") ? TDC_GPU_CODER_HUFF : (cname == "ascii" ? TDC_GPU_CODER_ASCII : TDC_GPU_CODER_ARITH);
        if (cname == "sle") {                                                 // option kmer = 3 (SLECoder.hpp:38)
            m_kmer = (unsigned)cv.get_int("kmer", 3);
            if (m_kmer < 1 || m_kmer > 7) throw std::runtime_error("sle: kmer must be in 1..7");
            m_coder = TDC_GPU_CODER_SLE_K((int)m_kmer);
        }
    }
    InputRestrictions input_restrictions() const override { return {true, true}; }   // uses_textds (Meta.hpp:277-282)

    void compress(Input& input, Output& output) override {
        if (!m_ctx) m_ctx = std::make_shared<GpuContext>(m_device);
        const bytes view = input.as_view();
        uint8_t* out = nullptr; size_t out_len = 0;
        const int rc = tdc_gpu_lcpcomp_compress_comp(m_ctx->h, view.data(), view.size(), (uint32_t)m_opts.get_int("threshold", 5),
                                                     (int)m_opts.get_int("flatten", 1), m_coder, m_comp, &out, &out_len, &last_stats);
        if (rc == TDC_GPU_ERR_NO_SENTINEL) throw std::logic_error(tdc_gpu_strerror(rc));          // ds/TextDS.hpp:132-138
        if (rc) throw std::runtime_error(std::string(tdc_gpu_strerror(rc)) + ": " + tdc_gpu_last_error(m_ctx->h));
        output.write(out, out_len);
        tdc_gpu_free(out);
    }

    // block mode (tdc --blocks SIZE): the UNRESTRICTED input is cut into blocks of block_size bytes, every block is compressed
    // on one of the visible devices as a stream of its own (tdc_gpu_blocks_compress) and framed in the block container
    void compress_blocks(const bytes& raw, size_t block_size, Output& output, int ndev = 0) {
        int have = tdc_gpu_device_count();
        if (have <= 0) throw std::runtime_error(std::string("tdc_gpu_blocks_compress: ") + tdc_gpu_strerror(TDC_GPU_ERR_HIP));
        if (ndev > 0 && ndev < have) have = ndev;
        std::vector<int> devs((size_t)have);
        for (int i = 0; i < have; ++i) devs[(size_t)i] = i;
        uint8_t* out = nullptr; size_t out_len = 0;
        const int rc = tdc_gpu_blocks_compress(devs.data(), have, raw.data(), raw.size(), block_size, (uint32_t)m_opts.get_int("threshold", 5),
                                               (int)m_opts.get_int("flatten", 1), m_coder, &out, &out_len, nullptr);
        if (rc) throw std::runtime_error(std::string("tdc_gpu_blocks_compress: ") + tdc_gpu_strerror(rc));
        output.write(out, out_len);
        tdc_gpu_free(out);
    }

    void decompress(Input& input, Output& output) override {
        if (BlockContainer::is_container(input.raw().data(), input.raw().size())) {      // every payload is a stream of its own
            const bytes& in = input.raw();
            for (const BlockContainer::Block& b : BlockContainer::parse(in.data(), in.size())) {
                Input part = Input::from_memory(b.data, b.len);
                bytes plain;
                Output po(plain);
                Output restricted(po, input_restrictions());                              // the block's own escaping + sentinel
                decompress(part, restricted);
                if (plain.size() != b.raw_len) throw std::runtime_error("corrupt container: block length mismatch");
                output.write_plain(plain.data(), plain.size());
            }
            return;
        }
        if (m_coder == TDC_GPU_CODER_ARITH)
            throw std::runtime_error("lcpcomp(coder=arithmetic) streams cannot be decoded (neither can the reference: "
                                     "consuming coders corrupt interleaved streams, docs/Documentation.md:1190-1203)");
        const bool on_device = m_opts.get("dec", "scan") == "gpu";
        if (!on_device && m_coder == TDC_GPU_CODER_ASCII) { lzss_ascii_decode(input, output); return; }
        if (!on_device && (m_coder & 0xFF) == TDC_GPU_CODER_SLE) { lzss_sle_decode(input, output, m_kmer); return; }
        // dec=gpu (an addition to the reference's decoder strategies scan / compact / ..., LCPCompressor.hpp:88): the stream is
        // parsed on the host, the references are resolved on the device (tdc_gpu_lcpcomp_decompress)
        if (m_opts.get("dec", "scan") == "gpu") {
            if (!m_ctx) m_ctx = std::make_shared<GpuContext>(m_device);
            const bytes& in = input.raw();
            uint8_t* out = nullptr; size_t out_len = 0;
            const int rc = tdc_gpu_lcpcomp_decompress_coder(m_ctx->h, in.data(), in.size(), m_coder, &out, &out_len, nullptr, nullptr);
            if (rc) throw std::runtime_error(std::string(tdc_gpu_strerror(rc)) + ": " + tdc_gpu_last_error(m_ctx->h));
            output.write(out, out_len);
            tdc_gpu_free(out);
            return;
        }
        lzss_huff_decode(input, output);
    }
};

// tdc::LZSSLCPCompressor<HuffmanCoder>  (compressors/LZSSLCPCompressor.hpp:22-132): threshold defaults to 3.
class LZSSLCPCompressor : public Compressor {
    AlgorithmValue m_opts;
    std::shared_ptr<GpuContext> m_ctx;
    int m_device = 0;
public:
    tdc_gpu_stats last_stats{};
    void set_device(int d) { m_device = d; }
    LZSSLCPCompressor(AlgorithmValue opts, std::shared_ptr<GpuContext> ctx) : m_opts(std::move(opts)), m_ctx(std::move(ctx)) {
        const std::string coder = m_opts.get("coder", "");
        if (coder != "huff") throw std::runtime_error("No implementation found for compressor lzss_lcp(coder=" + coder + ")");
    }
    InputRestrictions input_restrictions() const override { return {true, true}; }
    void compress(Input& input, Output& output) override {
        if (!m_ctx) m_ctx = std::make_shared<GpuContext>(m_device);
        const bytes view = input.as_view();
        uint8_t* out = nullptr; size_t out_len = 0;
        const int rc = tdc_gpu_lzss_lcp_compress(m_ctx->h, view.data(), view.size(), (uint32_t)m_opts.get_int("threshold", 3),
                                                 TDC_GPU_CODER_HUFF, &out, &out_len, &last_stats);
        if (rc == TDC_GPU_ERR_NO_SENTINEL) throw std::logic_error(tdc_gpu_strerror(rc));
        if (rc) throw std::runtime_error(std::string(tdc_gpu_strerror(rc)) + ": " + tdc_gpu_last_error(m_ctx->h));
        output.write(out, out_len);
        tdc_gpu_free(out);
    }
    void decompress(Input& input, Output& output) override { lzss_huff_decode(input, output); }   // :125-130 (DecodeBackBuffer)
};

// tdc::LZ78Compressor<EliasGammaCoder, trie>  (compressors/LZ78Compressor.hpp:45-161): no input restrictions.
class LZ78Compressor : public Compressor {
    AlgorithmValue m_opts;
    std::shared_ptr<GpuContext> m_ctx;
    int m_device = 0;
public:
    tdc_gpu_stats last_stats{};
    void set_device(int d) { m_device = d; }
    LZ78Compressor(AlgorithmValue opts, std::shared_ptr<GpuContext> ctx) : m_opts(std::move(opts)), m_ctx(std::move(ctx)) {
        const std::string coder = m_opts.get("coder", "bit");                 // the reference's default coder is BitCoder
        if (coder != "gamma")
            throw std::runtime_error("No implementation found for compressor lz78(coder=" + coder + ")");   // Registry.hpp:214
    }
    void compress(Input& input, Output& output) override {
        if (!m_ctx) m_ctx = std::make_shared<GpuContext>(m_device);
        const bytes& in = input.raw();
        uint8_t* out = nullptr; size_t out_len = 0;
        const int rc = tdc_gpu_lz78_compress(m_ctx->h, in.data(), in.size(), TDC_GPU_CODER_GAMMA, &out, &out_len, &last_stats);
        if (rc) throw std::runtime_error(std::string(tdc_gpu_strerror(rc)) + ": " + tdc_gpu_last_error(m_ctx->h));
        output.write(out, out_len);
        tdc_gpu_free(out);
    }
    // LZ78Compressor::decompress (:142-160) with EliasGammaCoder::Decoder: pairs until BitIStream eof
    void decompress(Input& input, Output& output) override {
        const bytes& in = input.raw();
        EliasGammaCoder::Decoder decoder(std::make_shared<BitIStream>(in.data(), in.size()));
        std::vector<uint32_t> parent(1, 0);
        std::vector<uint8_t> chr(1, 0);
        bytes text, tmp;
        const Range factor_r(0, std::numeric_limits<uint32_t>::max());       // the ranges are ignored by this coder (:26-29)
        while (!decoder.eof()) {
            const uint64_t id = decoder.decode<uint64_t>(factor_r);
            const uint64_t c = decoder.decode<uint64_t>(factor_r);
            if (id >= parent.size()) throw std::runtime_error("corrupt stream: unknown phrase id");
            tmp.clear();
            for (uint64_t x = id; x != 0; x = parent[x]) tmp.push_back(chr[x]);
            text.insert(text.end(), tmp.rbegin(), tmp.rend());
            text.push_back((uint8_t)c);
            parent.push_back((uint32_t)id);
            chr.push_back((uint8_t)c);
        }
        output.write(text.data(), text.size());
    }
};

// ---- registry (what is actually registered; Registry.hpp:204-231) ---------------------------------------------
struct Selection {
    std::string id_string;
    std::unique_ptr<Compressor> compressor;
    InputRestrictions restrictions;
};

inline std::vector<std::string> registered_algorithms() {
    return { "lcpcomp(coder=huff, comp=arrays, dec=scan(scans=6), threshold=5, flatten=1)   [MI355X, libtdc_gpu.so]",
             "lcpcomp(coder=huff, comp=plcppeaks, threshold=5, flatten=1)                 [MI355X, peak scan as an orbit marking]",
             "lcpcomp(coder=huff, dec=gpu)                                                [decompression: host parse, references resolved on the MI355X]",
             "lcpcomp(coder=ascii, comp=arrays, threshold=5, flatten=1)                   [MI355X; host decoder]",
             "lcpcomp(coder=sle(kmer=3), comp=arrays, threshold=5, flatten=1)             [MI355X; host decoder]",
             "lcpcomp(coder=..., comp=max_lcp | plcppeaks, ...)                           [MI355X]",
             "lcpcomp(coder=..., comp=heap, ...)                                          [MI355X, sequential replay of the reference's heap: a parity row, about a minute per MiB -- inputs of a few hundred KiB at most]",
             "lcpcomp(coder=arithmetic, comp=arrays, threshold=5, flatten=1)              [MI355X, compress only]",
             "lzss_lcp(coder=huff, threshold=3)                                           [MI355X, libtdc_gpu.so]",
             "lz78(coder=gamma)                                                           [host parse + MI355X gamma packer]" };
}

inline Selection select_algorithm(const std::string& id, std::shared_ptr<GpuContext> ctx = nullptr, int device = 0) {
    AlgorithmValue av = parse_algorithm_id(id, {"coder", "comp", "dec", "textds"});
    Selection s;
    s.id_string = id;
    if (av.name == "lz78") {
        auto z = std::make_unique<LZ78Compressor>(parse_algorithm_id(id, {"coder", "lz78trie"}), std::move(ctx));
        z->set_device(device);
        s.compressor = std::move(z);
        return s;
    }
    if (av.name == "lzss_lcp") {
        auto z = std::make_unique<LZSSLCPCompressor>(parse_algorithm_id(id, {"coder", "textds"}), std::move(ctx));
        z->set_device(device);
        s.restrictions = z->input_restrictions();
        s.compressor = std::move(z);
        return s;
    }
    if (av.name != "lcpcomp") throw std::runtime_error("No implementation found for compressor " + id);
    auto c = std::make_unique<LCPCompressor>(av, std::move(ctx));
    c->set_device(device);
    s.restrictions = c->input_restrictions();
    s.compressor = std::move(c);
    return s;
}

}  // namespace tdc_amd
