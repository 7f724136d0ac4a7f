// decode.hip -- LCPCompressor::decompress (compressors/LCPCompressor.hpp:140-150): decode_text_internal (:23-76) with
// HuffmanCoder::Decoder (coders/HuffmanCoder.hpp:572-612), references resolved on the device.
//
// The token stream has no synchronisation points (fixed-width fields interleaved with Huffman codes, run lengths of any
// size), so it is parsed sequentially on the host -- with a table-driven Huffman decoder and a 64-bit bit window -- into
// the literal bytes (already at their text positions) and the factor list.  What the reference spends its decompression
// time on, resolving the (forward and backward) references (lcpcomp/decompress/ScanDec.hpp:146-247: repeated scans;
// CompactDec / MultiMapBuffer: hash maps), is data parallel:
//   ref[p] = source position of text position p (NONE for literals);
//   pointer jumping in place: ref[p] <- ref[ref[p]] while ref[p] is not a literal.  Any value ever stored in ref[p] lies
//   on p's source chain, so unsynchronised rounds are safe and every round at least halves the remaining depth;
//   text[p] = text[ref[p]].
// The decoded text is unique, so the result equals the reference's for every valid stream (lzss_lcp streams have the same
// format: LZSSLCPCompressor.hpp:125-130).
#include "stages.hpp"
#include "prim.hpp"

#include <vector>

namespace tdc {

namespace {

typedef StreamFormatError StreamError;

// MSB-first reader over the reference's bit stream incl. its terminator rule (io/BitIStream.hpp:27-63, :191-193):
// the low 3 bits of the last byte give the number of valid bits of the final data byte (6 and 7 live in an extra byte).
struct FastBits {
    const u8* p;
    size_t nbytes;
    u64 total = 0, pos = 0;
    FastBits(const u8* in, size_t n) : p(in), nbytes(n) {
        if (n == 0) return;
        const unsigned fb = in[n - 1] & 7u;
        if (fb >= 6) { if (n < 2) throw StreamError{"truncated stream"}; total = 8ull * (n - 2) + fb; }
        else total = 8ull * (n - 1) + fb;
    }
    bool eof() const { return pos >= total; }
    // next 57 bits, left-aligned in the result's top bits (zeros beyond the end, like BitIStream::read_bit at eof)
    u64 peek() const {
        const size_t byte = (size_t)(pos >> 3);
        u64 w = 0;
        if (byte + 8 <= nbytes) { u64 t; memcpy(&t, p + byte, 8); w = __builtin_bswap64(t); }
        else for (size_t i = 0; i < 8; ++i) w = (w << 8) | (byte + i < nbytes ? p[byte + i] : 0);
        w <<= (pos & 7);
        if (pos + 57 > total) {                               // mask the bits behind the end of the stream
            const u64 valid = total > pos ? total - pos : 0;
            w = valid == 0 ? 0 : (w & (~0ull << (64 - valid)));
        }
        return w;
    }
    u64 read(unsigned bits) {                                 // bits <= 57
        if (bits == 0) return 0;
        const u64 v = peek() >> (64 - bits);
        pos += bits;
        return v;
    }
    u64 read_compressed_int() {                               // io/BitIStream.hpp:174-188, 7-bit groups
        u64 v = 0; unsigned i = 0; bool more;
        do { more = read(1) != 0; v |= read(7) << (7 * i++); } while (more && i < 10);
        return v;
    }
};

__global__ void ref_scatter_kernel(const u32* __restrict__ pos, const u32* __restrict__ src, const u32* __restrict__ len, size_t z,
                                   int G, u32* __restrict__ ref) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = threadIdx.x % G;
    if (i >= z) return;
    const u32 p = pos[i], s = src[i], l = len[i];
    for (u32 j = sub; j < l; j += G) ref[p + j] = s + j;
}

// one round of in-place pointer jumping; *changed != 0 if some reference moved
__global__ __launch_bounds__(256) void ref_jump_kernel(u32* ref, size_t n, u32* __restrict__ changed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool any = false;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const u32 q = ref[p];
        if (q == NONE32) continue;
        const u32 r = ref[q];
        if (r != NONE32) { ref[p] = r; any = true; }
    }
    if (__any(any) && lane_id() == 0) atomicOr(changed, 1u);
}

__global__ void ref_copy_kernel(const u32* __restrict__ ref, size_t n, u8* text) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 q = ref[p];
    if (q != NONE32) text[p] = text[q];          // q is a literal position: never written by this kernel
}

}  // namespace

// Host parse: literals go straight to their text positions in `text` (n bytes), factors into the three vectors.
// Returns n.  Throws StreamError for malformed input.
static u64 parse_lzss_huff_stream(const u8* in, size_t len, std::vector<u8>& text, std::vector<u32>& fpos, std::vector<u32>& fsrc,
                                  std::vector<u32>& flen) {
    FastBits bs(in, len);
    // HuffmanCoder::Decoder ctor (HuffmanCoder.hpp:581-597) + huffmantable_decode (:278-290)
    const bool have_table = bs.read(1) != 0;
    u8 order[256];
    u64 firstcode[64];
    size_t prefix_sum[64];
    unsigned longest = 0;
    u8 numl[64] = {0};
    size_t sigma = 0;
    constexpr unsigned LUT_BITS = 12;
    std::vector<unsigned short> lut;                          // (symbol << 4) | code length, 0 = longer than LUT_BITS / invalid
    if (have_table) {
        longest = (unsigned)(bs.read_compressed_int() & 0xFF);
        if (longest == 0 || longest > 57) throw StreamError{"corrupt Huffman table"};
        for (unsigned i = 0; i < longest; ++i) numl[i] = (u8)bs.read_compressed_int();
        sigma = (size_t)bs.read_compressed_int();
        if (sigma > 256) throw StreamError{"corrupt Huffman table"};
        for (size_t i = 0; i < sigma; ++i) order[i] = (u8)bs.read(8);
        firstcode[longest - 1] = 0;                                              // gen_first_codes :192-198
        for (unsigned i = longest - 1; i > 0; --i) firstcode[i - 1] = (firstcode[i] + numl[i]) / 2;
        size_t acc = 0;                                                           // gen_prefix_sum_lengths :350-370
        for (unsigned l = 0; l < longest; ++l) { prefix_sum[l] = acc; acc += numl[l]; }
        if (acc > sigma) throw StreamError{"corrupt Huffman table"};
        lut.assign((size_t)1 << LUT_BITS, 0);
        for (unsigned l = 1; l <= longest && l <= LUT_BITS; ++l)
            for (unsigned k = 0; k < numl[l - 1]; ++k) {
                const u64 code = firstcode[l - 1] + k;
                if (code >> l) throw StreamError{"corrupt Huffman table"};
                const unsigned short e = (unsigned short)((order[prefix_sum[l - 1] + k] << 4) | l);
                const size_t base = (size_t)code << (LUT_BITS - l);
                for (size_t x = 0; x < ((size_t)1 << (LUT_BITS - l)); ++x) lut[base + x] = e;
            }
    }
    // decode_text_internal (LCPCompressor.hpp:23-76)
    const u64 n = bs.read(32);
    const unsigned W = bits_for(n);
    const u64 flen_min = bs.read(W), flen_max = bs.read(W), fdist_max = bs.read(W);
    const unsigned lbits = bits_for(flen_max - flen_min), dbits = bits_for(fdist_max);
    if (n == 0 || n >= 0x7FFFFFFFull) throw StreamError{"text length out of range"};     // a text always holds its sentinel
    {   // plausibility (a corrupt header would otherwise ask for gigabytes): a literal costs at least one bit, a factor at least W
        // bits and covers at most flen_max positions
        const u64 bits = (u64)len * 8;
        if (n > bits + (bits / W + 1) * (flen_max ? flen_max : 1)) throw StreamError{"text length out of range"};
    }
    text.assign((size_t)n, 0);
    u64 p = 0;
    while (!bs.eof()) {
        u64 num = bs.read(1) ? bs.read(dbits) : 0;
        if (p + num > n) throw StreamError{"corrupt stream: too many literals"};
        if (!have_table) {
            while (num--) text[(size_t)p++] = (u8)bs.read(8);                     // HuffmanCoder.hpp:606-607
        } else {
            while (num--) {                                                       // huffman_decode :377-397
                const u64 w = bs.peek();
                const unsigned short e = lut[(size_t)(w >> (64 - LUT_BITS))];
                if (e) { text[(size_t)p++] = (u8)(e >> 4); bs.pos += e & 15u; continue; }
                u64 value = 0; unsigned length = 0;
                do { value = (value << 1) | ((w >> (63 - length)) & 1u); ++length; } while (length <= longest && value < firstcode[length - 1]);
                if (length > longest) throw StreamError{"corrupt Huffman code"};
                --length;
                const u64 off = value - firstcode[length];                        // a table that violates Kraft would index behind order[]
                if (off >= numl[length] || prefix_sum[length] + off >= sigma) throw StreamError{"corrupt Huffman code"};
                text[(size_t)p++] = order[prefix_sum[length] + off];
                bs.pos += length + 1;
            }
        }
        if (!bs.eof()) {
            const u64 src = bs.read(W), l = flen_min + bs.read(lbits);
            if (l == 0 || p + l > n || src + l > n) throw StreamError{"corrupt stream: factor out of range"};
            fpos.push_back((u32)p); fsrc.push_back((u32)src); flen.push_back((u32)l);
            p += l;
        }
    }
    if (p != n) throw StreamError{"corrupt stream: length mismatch"};
    return n;
}

// The same token stream written with SLECoder (coders/SLECoder.hpp:301-453: ranking header, rank class codes, k-mer symbols
// expand to k literals, MinDistributedRange for the factor length).
static u64 parse_lzss_sle_stream(const u8* in, size_t len, unsigned k, std::vector<u8>& text, std::vector<u32>& fpos,
                                 std::vector<u32>& fsrc, std::vector<u32>& flen) {
    FastBits bs(in, len);
    const size_t sigma = (size_t)bs.read_compressed_int();                  // Decoder ctor :325-340
    if (sigma == 0 || sigma > 4096) throw StreamError{"corrupt SLE ranking"};
    const unsigned sb = bits_for(sigma - 1);
    std::vector<u64> inv(sigma);
    for (size_t r = 0; r < sigma; ++r) inv[r] = bs.read_compressed_int();
    auto read_rank = [&]() -> u64 {                                           // :367-397
        if (sb < 4) return bs.read(sb);
        if (sb < 6) return bs.read(1) ? bs.read(sb) : bs.read(2);
        if (sb == 6) {
            switch (bs.read(2)) {
                case 0: return bs.read(3);
                case 1: return 8 + bs.read(3);
                case 2: return 16 + bs.read(4);
                default: return bs.read(sb);
            }
        }
        const u64 cls = bs.read(3);
        if (cls < 4) return 4 * cls + bs.read(2);
        if (cls < 7) return 16 + 8 * (cls - 4) + bs.read(3);
        return bs.read(sb);
    };
    const u64 n = bs.read(32);
    const unsigned W = bits_for(n);
    const u64 flen_min = bs.read(W), flen_max = bs.read(W), fdist_max = bs.read(W);
    const unsigned lbits = bits_for(flen_max - flen_min), dbits = bits_for(fdist_max);
    if (n == 0 || n >= 0x7FFFFFFFull) throw StreamError{"text length out of range"};
    text.assign((size_t)n, 0);
    u8 kmer[8]; size_t kread = (size_t)-1;
    u64 p = 0;
    auto eof = [&] { return kread < k ? false : bs.eof(); };                  // :351-359
    while (!eof()) {
        kread = (size_t)-1;
        u64 num = bs.read(1) ? bs.read(dbits) : 0;
        if (p + num > n) throw StreamError{"corrupt stream: too many literals"};
        while (num--) {
            u8 ch;
            if (kread < k) ch = kmer[kread++];
            else {
                const u64 r = read_rank();
                if (r >= sigma) throw StreamError{"corrupt stream: rank out of range"};
                const u64 x = inv[(size_t)r];
                if ((x >> 56) == 0xFF) { for (unsigned i = 0; i < k; ++i) kmer[k - 1 - i] = (u8)(x >> (8 * i)); kread = 1; ch = kmer[0]; }
                else ch = (u8)x;
            }
            text[(size_t)p++] = ch;
        }
        if (!eof()) {
            kread = (size_t)-1;
            const u64 src = bs.read(W);
            u64 v;                                                           // decode(MinDistributedRange) :413-431
            if (lbits <= 5) v = bs.read(lbits);
            else switch (bs.read(2)) {
                case 0: v = bs.read(3); break;
                case 1: v = 8 + bs.read(3); break;
                case 2: v = 16 + bs.read(4); break;
                default: v = bs.read(lbits); break;
            }
            const u64 l = flen_min + v;
            if (l == 0 || p + l > n || src + l > n) throw StreamError{"corrupt stream: factor out of range"};
            fpos.push_back((u32)p); fsrc.push_back((u32)src); flen.push_back((u32)l);
            p += l;
        }
    }
    if (p != n) throw StreamError{"corrupt stream: length mismatch"};
    return n;
}

// ... and with ASCIICoder (coders/ASCIICoder.hpp:53-84): decimal integers up to the first non-digit, a bit is any byte but
// '0', literals are raw bytes; the BitOStream terminator byte ends the stream.
static u64 parse_lzss_ascii_stream(const u8* in, size_t in_len, std::vector<u8>& text, std::vector<u32>& fpos, std::vector<u32>& fsrc,
                                   std::vector<u32>& flen) {
    if (in_len == 0) throw StreamError{"corrupt stream: empty"};
    const size_t len = in_len - 1;
    size_t at = 0;
    auto read_int = [&]() -> u64 {
        u64 v = 0; int digits = 0;
        while (at < len) {
            const u8 ch = in[at++];
            if (ch < '0' || ch > '9') { if (!digits) break; return v; }
            if (digits >= 18) break;
            v = v * 10 + (ch - '0'); ++digits;
        }
        throw StreamError{"corrupt stream: integer expected"};
    };
    const u64 n = read_int();
    read_int(); read_int(); read_int();                                       // flen_min, flen_max, fdist_max: unused by this coder
    if (n == 0 || n >= 0x7FFFFFFFull) throw StreamError{"text length out of range"};
    text.assign((size_t)n, 0);
    u64 p = 0;
    while (at < len) {
        u64 num = (in[at++] != '0') ? read_int() : 0;
        if (p + num > n || at + num > len) throw StreamError{"corrupt stream: too many literals"};
        while (num--) text[(size_t)p++] = in[at++];
        if (at < len) {
            const u64 src = read_int(), l = read_int();
            if (l == 0 || p + l > n || src + l > n) throw StreamError{"corrupt stream: factor out of range"};
            fpos.push_back((u32)p); fsrc.push_back((u32)src); flen.push_back((u32)l);
            p += l;
        }
    }
    if (p != n) throw StreamError{"corrupt stream: length mismatch"};
    return n;
}

size_t decode_lzss_huff(Ctx& c, const u8* stream, size_t len, std::vector<u8>& text, DecodeStats* st) {
    return decode_lzss(c, stream, len, 0, text, st);
}

// coder: 0 = HuffmanCoder, 2 = ASCIICoder, 3 | kmer << 8 = SLECoder (the coder ids of encode_stream)
size_t decode_lzss(Ctx& c, const u8* stream, size_t len, int coder, std::vector<u8>& text, DecodeStats* st) {
    DecodeStats local;
    if (!st) st = &local;
    *st = DecodeStats();
    std::vector<u32> fpos, fsrc, flen;
    u64 n;
    if ((coder & 0xFF) == 3) n = parse_lzss_sle_stream(stream, len, (unsigned)(coder >> 8) ? (unsigned)(coder >> 8) : 3u, text, fpos, fsrc, flen);
    else if (coder == 2) n = parse_lzss_ascii_stream(stream, len, text, fpos, fsrc, flen);
    else n = parse_lzss_huff_stream(stream, len, text, fpos, fsrc, flen);
    const size_t z = fpos.size();
    st->factors = z;
    if (n == 0 || z == 0) return (size_t)n;
    hipStream_t s = c.stream;
    c.ensure_arena((size_t)n * 5 + z * 12 + ((size_t)16 << 20));
    const size_t mark = c.arena.mark();
    u8* d_text = c.arena.get<u8>((size_t)n);
    u32* d_ref = c.arena.get<u32>((size_t)n);
    u32* d_pos = c.arena.get<u32>(z), *d_src = c.arena.get<u32>(z), *d_len = c.arena.get<u32>(z);
    u32* d_changed = c.arena.get<u32>(1);
    HIP_TRY(hipMemcpyAsync(d_text, text.data(), (size_t)n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_pos, fpos.data(), z * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_src, fsrc.data(), z * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_len, flen.data(), z * 4, hipMemcpyHostToDevice, s));
    fill_u32(c, d_ref, (size_t)n, NONE32);
    const int G = (z * 64 > n) ? 8 : 64;
    ref_scatter_kernel<<<cdiv(z * G, 256), 256, 0, s>>>(d_pos, d_src, d_len, z, G, d_ref);
    LAUNCH_CHECK();
    unsigned g = cdiv((size_t)n, 256 * 8); if (g > 16384) g = 16384;
    for (u32 round = 0;; ++round) {
        if (round > 40) throw StreamFormatError{"corrupt stream: reference cycle"};     // depth < 2^31
        HIP_TRY(hipMemsetAsync(d_changed, 0, sizeof(u32), s));
        ref_jump_kernel<<<g, 256, 0, s>>>(d_ref, (size_t)n, d_changed);
        LAUNCH_CHECK();
        st->rounds = round + 1;
        if (c.read(d_changed) == 0) break;
    }
    ref_copy_kernel<<<cdiv((size_t)n, 256), 256, 0, s>>>(d_ref, (size_t)n, d_text);
    LAUNCH_CHECK();
    HIP_TRY(hipMemcpyAsync(text.data(), d_text, (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c.arena.release(mark);
    return (size_t)n;
}

}  // namespace tdc
