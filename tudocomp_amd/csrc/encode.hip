// encode.hip -- literal histogram, per-position bit cost, exclusive scan, MSB-first bit packing.
//
// Replaces, for coder = HuffmanCoder:
//   lzss::TextLiterals + huff::count_alphabet_literals   (compressors/lzss/LZSSLiterals.hpp:10-50, coders/HuffmanCoder.hpp:37-48)
//   lzss::encode_text                                    (compressors/lzss/LZSSCoding.hpp:18-92)
//   tdc::Encoder::encode(v, Range/BitRange)              (Coder.hpp:61-77)
//   HuffmanCoder::Encoder::encode(v, LiteralRange)       (coders/HuffmanCoder.hpp:562-569)
//   io::BitOStream incl. the destructor's terminator     (io/BitOStream.hpp:53-64, 79-102)
//
// Token stream in position space.  Every literal run -- the tail run included -- is preceded by
// "1" + run length in bits_for(fdist_max) bits; a factor that directly follows another factor (or starts the text)
// is preceded by "0"; a factor is src in W bits + (len - flen_min) in bits_for(flen_max - flen_min) bits.
// So the bits a text position contributes depend only on local data:
//   literal p        : [ "1" + runlen  if p starts a run ] + code(T[p])
//   factor start p   : [ "0"  if p == 0 or p-1 is covered ] + src + len
//   covered, not start: nothing
// Two tiled passes: (1) bits per 2048-position tile, exclusive scan of the tile sums (u64);
// (2) recompute the costs, scan inside the workgroup, OR the bits into the zeroed output (big-endian 64-bit words).
#include "stages.hpp"
#include "prim.hpp"
#include "huffman_host.hpp"
#include "arith.hpp"

#include <string.h>

namespace tdc {

struct CodeTable {
    u64 code[256];
    u8 len[256];
};

struct EncParams {
    u32 W, lbits, dbits, flen_min;
    u32 raw_literals;     // sigma <= 1: literals are written as 8 raw bits (HuffmanCoder.hpp:565-566)
    u32 ascii;            // ASCIICoder (coders/ASCIICoder.hpp:29-50): decimal integers + ':', '0'/'1', raw literals, no "- min"
};

// number of characters ASCIICoder writes for an integer: its decimal digits and the ':'
__device__ __forceinline__ u32 ascii_int_chars(u32 v) {
    u32 d = 1;
    if (v >= 1000000000u) d = 10; else if (v >= 100000000u) d = 9; else if (v >= 10000000u) d = 8; else if (v >= 1000000u) d = 7;
    else if (v >= 100000u) d = 6; else if (v >= 10000u) d = 5; else if (v >= 1000u) d = 4; else if (v >= 100u) d = 3; else if (v >= 10u) d = 2;
    return d + 1;
}

struct EncScalars { u32 flen_min, flen_max, fdist_max, pad; };

// literal coder = ArithmeticCoder: what a literal contributes is looked up per literal ordinal (arith.hip)
struct ArithDev {
    const u32* litidx;     // nullptr: Huffman mode
    const u8* amark;
    const u64* fval;
    u32 lc_index;
    u64 pp_lb;
};

constexpr int ENC_PER_THREAD = 8;
constexpr int ENC_TILE = 256 * ENC_PER_THREAD;     // 2048 text positions per workgroup

// gaps between consecutive factors (LZSSCoding.hpp:28-38) + min/max factor length (LZSSFactors.hpp:41-47);
// stores the literal-run length at the first position of every run.
__global__ __launch_bounds__(256) void gaps_kernel(const u32* __restrict__ fpos, const u32* __restrict__ flen_list, size_t z,
                                                    size_t n, u32* __restrict__ flen, EncScalars* __restrict__ sc) {
    __shared__ u32 sg[4], smax[4], smin[4];
    u32 gmax = 0, lmin = 0xFFFFFFFFu, lmax = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < z; i += stride) {
        // flen_list == nullptr: the lengths are read at the factor starts (run lengths are written at run starts, never there)
        const u32 p = fpos[i], l = flen_list ? flen_list[i] : flen[p];
        const u32 prev_end = (i == 0) ? 0u : fpos[i - 1] + (flen_list ? flen_list[i - 1] : flen[fpos[i - 1]]);
        const u32 gap = p - prev_end;
        if (gap) flen[prev_end] = gap;
        gmax = max(gmax, gap);
        lmin = min(lmin, l);
        lmax = max(lmax, l);
        if (i + 1 == z) {
            const u32 end = p + l;
            if ((size_t)end < n) { const u32 tail = (u32)(n - end); flen[end] = tail; gmax = max(gmax, tail); }
        }
    }
    gmax = wave_reduce_max(gmax);
    lmax = wave_reduce_max(lmax);
    lmin = wave_reduce_min(lmin);
    if (lane_id() == 0) { sg[wave_id()] = gmax; smax[wave_id()] = lmax; smin[wave_id()] = lmin; }
    __syncthreads();
    if (threadIdx.x == 0) {      // one atomic triple per workgroup (the grid is capped, so a few thousand in total)
        atomicMax(&sc->fdist_max, max(max(sg[0], sg[1]), max(sg[2], sg[3])));
        atomicMax(&sc->flen_max, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
        atomicMin(&sc->flen_min, min(min(smin[0], smin[1]), min(smin[2], smin[3])));
    }
}

__global__ __launch_bounds__(256) void literal_hist_kernel(const u8* __restrict__ text, const u32* __restrict__ owner, size_t n,
                                                            u32* __restrict__ hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride)
        if (owner[p] == NONE32) atomicAdd(&h[text[p]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// bits contributed by position p (see file header)
template <bool ASCII>
__device__ __forceinline__ u32 position_cost(u32 own, u32 own_prev, bool first, u32 fl, u8 ch, u32 p, const u8* __restrict__ clen,
                                             const u32* __restrict__ fsrc,
                                             const EncParams& P, const ArithDev& A) {
    if (ASCII) {
        if (own == NONE32) return 8u + (fl ? 8u * (1u + ascii_int_chars(fl)) : 0u);
        if (first || own != own_prev) return ((first || own_prev != NONE32) ? 8u : 0u) + 8u * (ascii_int_chars(fsrc[p]) + ascii_int_chars(fl));
        return 0u;
    }
    if (own == NONE32) {
        u32 c;
        if (A.litidx) { const u32 k = A.litidx[p]; c = (A.amark[k] ? 64u : 0u) + (k == A.lc_index ? 128u : 0u); }
        else c = P.raw_literals ? 8u : (u32)clen[ch];
        if (fl) c += 1u + P.dbits;
        return c;
    }
    if (first || own != own_prev) {                 // a factor starts where the covering factor changes
        u32 c = P.W + P.lbits;
        if (first || own_prev != NONE32) c += 1u;
        return c;
    }
    return 0u;
}

template <bool ASCII>
__global__ __launch_bounds__(256) void tile_bits_kernel(const u8* __restrict__ text, const u32* __restrict__ owner,
                                                         const u32* __restrict__ flen, const u32* __restrict__ fsrc, size_t n, CodeTable tab, EncParams P,
                                                         ArithDev A, u64* __restrict__ tile_bits) {
    __shared__ u8 clen[256];
    __shared__ u32 sm[4];
    clen[threadIdx.x] = tab.len[threadIdx.x];
    __syncthreads();
    const size_t p0 = (size_t)blockIdx.x * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    u32 sum = 0;
    if (p0 < n) {
        u32 prev = (p0 == 0) ? 0u : owner[p0 - 1];
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) {
            const size_t p = p0 + j;
            if (p < n) {
                const u32 own = owner[p];
                sum += position_cost<ASCII>(own, prev, p == 0, flen[p], text[p], (u32)p, clen, fsrc, P, A);
                prev = own;
            }
        }
    }
    sum = wave_reduce_sum(sum);
    if (lane_id() == 0) sm[wave_id()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tile_bits[blockIdx.x] = (u64)sm[0] + sm[1] + sm[2] + sm[3];
}

// Append `nbits` (1..64) bits of `val` at absolute bit position `bitpos`; the stream is MSB-first, so the output is
// treated as big-endian 64-bit words.  Different threads own disjoint bit ranges: OR is order-independent.
__device__ __forceinline__ void put_bits(u64* __restrict__ out, u64 bitpos, u64 val, u32 nbits) {
    const u64 w = bitpos >> 6;
    const u32 off = (u32)(bitpos & 63);
    const u32 avail = 64 - off;
    if (nbits <= avail) {
        const u64 x = (nbits == 64) ? val : (val << (avail - nbits));
        atomicOr((unsigned long long*)&out[w], (unsigned long long)__builtin_bswap64(x));
    } else {
        const u32 rem = nbits - avail;
        atomicOr((unsigned long long*)&out[w], (unsigned long long)__builtin_bswap64(val >> rem));
        atomicOr((unsigned long long*)&out[w + 1], (unsigned long long)__builtin_bswap64(val << (64 - rem)));
    }
}

struct BitSink {
    u64* out;
    u64 pos;     // bit position of the first pending bit
    u64 acc;     // pending bits, right-aligned
    u32 cnt;
    __device__ __forceinline__ void flush() {
        if (cnt) { put_bits(out, pos, acc, cnt); pos += cnt; cnt = 0; acc = 0; }
    }
    __device__ __forceinline__ void append(u64 v, u32 nb) {     // v < 2^nb, 0 <= nb <= 64
        if (nb == 0) return;
        if (cnt + nb > 64) flush();
        acc = (nb == 64) ? v : ((acc << nb) | v);
        cnt += nb;
    }
};

// ASCIICoder::Encoder::encode(v, Range) (ASCIICoder.hpp:33-39): decimal digits, most significant first, then ':'
__device__ __forceinline__ void append_ascii_int(BitSink& sink, u32 v) {
    u32 pw = 1000000000u;
    bool started = false;
    for (int i = 0; i < 10; ++i) {
        const u32 d = v / pw;
        if (d || started || pw == 1u) { sink.append('0' + d, 8); started = true; }
        v -= d * pw;
        pw /= 10u;
    }
    sink.append(':', 8);
}

template <bool ASCII>
__global__ __launch_bounds__(256) void pack_kernel(const u8* __restrict__ text, const u32* __restrict__ owner,
                                                    const u32* __restrict__ flen, const u32* __restrict__ fsrc, size_t n,
                                                    CodeTable tab, EncParams P, ArithDev A, const u64* __restrict__ tile_off,
                                                    u64 base_bits, u64* __restrict__ out) {
    __shared__ u8 clen[256];
    __shared__ u64 code[256];
    __shared__ u32 sm[5];
    clen[threadIdx.x] = tab.len[threadIdx.x];
    code[threadIdx.x] = tab.code[threadIdx.x];
    __syncthreads();
    const size_t p0 = (size_t)blockIdx.x * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    u32 own[ENC_PER_THREAD], fl[ENC_PER_THREAD];
    u8 ch[ENC_PER_THREAD];
    u32 prev0 = 0, sum = 0;
    if (p0 < n) {
        prev0 = (p0 == 0) ? 0u : owner[p0 - 1];
        u32 prev = prev0;
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) {
            const size_t p = p0 + j;
            if (p < n) {
                own[j] = owner[p]; fl[j] = flen[p]; ch[j] = text[p];
                sum += position_cost<ASCII>(own[j], prev, p == 0, fl[j], ch[j], (u32)p, clen, fsrc, P, A);
                prev = own[j];
            } else { own[j] = 0; fl[j] = 0; ch[j] = 0; }
        }
    }
    u32 total;
    const u32 excl = block_exclusive_sum<u32, 4>(sum, sm, total);
    if (p0 >= n) return;
    BitSink sink;
    sink.out = out;
    sink.pos = base_bits + tile_off[blockIdx.x] + excl;
    sink.acc = 0;
    sink.cnt = 0;
    u32 prev = prev0;
#pragma unroll
    for (int j = 0; j < ENC_PER_THREAD; ++j) {
        const size_t p = p0 + j;
        if (p < n) {
            const u32 o = own[j];
            if (ASCII) {
                if (o == NONE32) {
                    if (fl[j]) { sink.append('1', 8); append_ascii_int(sink, fl[j]); }
                    sink.append(ch[j], 8);
                } else if (p == 0 || o != prev) {
                    if (p == 0 || prev != NONE32) sink.append('0', 8);
                    append_ascii_int(sink, fsrc[p]);
                    append_ascii_int(sink, fl[j]);                                           // no "- flen_min" (ASCIICoder ignores the range)
                }
            } else if (o == NONE32) {
                if (fl[j]) { sink.append(1, 1); sink.append(fl[j], P.dbits); }           // LZSSCoding.hpp:62-68, :83-86
                if (A.litidx) {                                                           // ArithmeticCoder.hpp:96-104, :151-155
                    const u32 k = A.litidx[p];
                    if (A.amark[k]) sink.append(A.fval[k], 64);
                    if (k == A.lc_index) { sink.append(A.pp_lb, 64); sink.append(~0ull, 64); }
                }
                else if (P.raw_literals) sink.append(ch[j], 8);                           // HuffmanCoder.hpp:565-566
                else sink.append(code[ch[j]], clen[ch[j]]);                               // :568 huffman_encode
            } else if (p == 0 || o != prev) {
                if (p == 0 || prev != NONE32) sink.append(0, 1);                          // LZSSCoding.hpp:57-59
                sink.append(fsrc[p], P.W);                                                // :77
                sink.append(fl[j] - P.flen_min, P.lbits);                                 // :78
            }
            prev = o;
        }
    }
    sink.flush();
}

// io/BitOStream.hpp:53-64 : u = bits used in the last byte; u <= 5: OR u into that byte, else append a byte holding u.
__global__ void terminator_kernel(u8* out, u64 total_bits) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u64 byte = total_bits >> 3;
    const u32 u = (u32)(total_bits & 7);
    if (u <= 5) out[byte] |= (u8)u;
    else out[byte + 1] = (u8)u;
}

// worst case per position: 1 + 32 bits of run header + 64 (+128 once) of arithmetic words; ASCIICoder (coder 2): a one-byte
// factor costs '0' + two integers of up to 10 digits and ':' each
size_t encode_bound(size_t n) { return 12 * n + 4096; }
size_t encode_bound_coder(size_t n, int coder) { return (coder == 2 ? 24 : 12) * n + 4096; }

size_t encode_huff(Ctx& c, const u8* text, size_t n, FactorSpace fs, u8* d_out, size_t out_cap, EncodeStats* st) {
    return encode_stream(c, text, n, fs, 0, d_out, out_cap, st);
}

size_t encode_stream(Ctx& c, const u8* text, size_t n, FactorSpace fs, int coder, u8* d_out, size_t out_cap, EncodeStats* st) {
    EncodeStats local;
    if (!st) st = &local;
    *st = EncodeStats();
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();

    // ---- factor list (position order), gaps, min/max lengths ------------------------------------------------
    u32* fpos = fs.have_list ? fs.fpos : c.arena.get<u32>(n);
    u32* flist = fs.have_list ? nullptr : c.arena.get<u32>(n);
    const size_t z = fs.have_list ? fs.nfact : extract_factors(c, n, fs, fpos, nullptr, flist, n);
    EncScalars* d_sc = (EncScalars*)c.arena.alloc(sizeof(EncScalars));
    EncScalars h_sc = { 0xFFFFFFFFu, 0u, 0u, 0u };          // LZSSFactors.hpp:33-38 : INDEX_MAX / 0
    HIP_TRY(hipMemcpyAsync(d_sc, &h_sc, sizeof(h_sc), hipMemcpyHostToDevice, s));
    u32* d_hist = c.arena.get<u32>(256);
    HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), s));
    if (z) {
        unsigned g = cdiv(z, 256); if (g > 2048) g = 2048;
        Ctx::ProfScope prof(c, K_ENC_GAPS, (u64)z * 12);
        gaps_kernel<<<g, 256, 0, s>>>(fpos, flist, z, n, fs.flen, d_sc);
        LAUNCH_CHECK();
    } else {
        // no factor: one literal run covering the whole text (LZSSCoding.hpp:38, :83-91)
        const u32 run = (u32)n;
        HIP_TRY(hipMemcpyAsync(fs.flen, &run, sizeof(u32), hipMemcpyHostToDevice, s));
        h_sc.fdist_max = run;
        HIP_TRY(hipMemcpyAsync(d_sc, &h_sc, sizeof(h_sc), hipMemcpyHostToDevice, s));
    }
    {
        unsigned g = cdiv(n, 256 * 16); if (g > 2048) g = 2048; if (g == 0) g = 1;
        Ctx::ProfScope prof(c, K_ENC_HIST, (u64)n * 5);
        literal_hist_kernel<<<g, 256, 0, s>>>(text, fs.owner, n, d_hist);
        LAUNCH_CHECK();
    }
    u32 h_hist[256];
    c.read_n(d_hist, h_hist, 256);
    h_sc = c.read(d_sc);

    // ---- host: coder header (HuffmanCoder::Encoder ctor :526-547 / ArithmeticCoder::Encoder ctor :158-164),
    //      then the fields of LZSSCoding.hpp:47-50
    HuffTable ht;
    HostBitWriter hw;
    ArithDev A = { nullptr, nullptr, nullptr, 0, 0 };
    if (coder == 2) {
        memset(&ht, 0, sizeof(ht));                                        // ASCIICoder writes no header
    } else if (coder == 0) {
        build_huffman_table(h_hist, &ht);
        write_huffman_header(hw, ht);
    } else {
        ArithModel am;
        if (!arith_build_model(h_hist, &am, hw))
            throw HipError{hipErrorInvalidValue, "arithmetic coder: all literal bytes >= 1 are absent (the reference divides by zero)", -1};
        ArithPlan plan;
        arith_prepare(c, text, n, fs.owner, am, &plan);
        A.litidx = plan.litidx; A.amark = plan.amark; A.fval = plan.fval; A.lc_index = plan.lc_index; A.pp_lb = plan.pp_lb;
    }
    EncParams P;
    P.W = bits_for(n);
    P.lbits = bits_for((u64)h_sc.flen_max - (u64)h_sc.flen_min);       // only used when z > 0
    P.dbits = bits_for(h_sc.fdist_max);
    P.flen_min = h_sc.flen_min;
    P.raw_literals = (ht.sigma <= 1) ? 1u : 0u;
    P.ascii = (coder == 2) ? 1u : 0u;
    if (coder == 2) {
        const u64 fields[4] = { (u64)n, h_sc.flen_min, h_sc.flen_max, h_sc.fdist_max };
        for (u64 v : fields) {
            char tmp[24]; int k = 0;
            do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
            while (k) hw.write_int((u8)tmp[--k], 8);
            hw.write_int(':', 8);
        }
    } else {
        hw.write_int(n, 32);
        hw.write_int(h_sc.flen_min, P.W);
        hw.write_int(h_sc.flen_max, P.W);
        hw.write_int(h_sc.fdist_max, P.W);
    }
    const u64 base_bits = hw.nbits;
    CodeTable tab;
    memcpy(tab.code, ht.code_of, sizeof(tab.code));
    memcpy(tab.len, ht.len_of, sizeof(tab.len));

    // ---- pass 1: bits per tile, scan -------------------------------------------------------------------------
    const unsigned tiles = cdiv(n, ENC_TILE);
    u64* tile_bits = c.arena.get<u64>(tiles + 1);
    u64* d_total = c.arena.get<u64>(1);
    {
        Ctx::ProfScope prof(c, K_ENC_TILE_BITS, (u64)n * 9);
        if (P.ascii) tile_bits_kernel<true><<<tiles, 256, 0, s>>>(text, fs.owner, fs.flen, fs.fsrc, n, tab, P, A, tile_bits);
        else         tile_bits_kernel<false><<<tiles, 256, 0, s>>>(text, fs.owner, fs.flen, fs.fsrc, n, tab, P, A, tile_bits);
        LAUNCH_CHECK();
    }
    exclusive_sum_u64(c, tile_bits, tile_bits, tiles, d_total);
    const u64 total_bits = base_bits + c.read(d_total);
    const size_t out_len = (size_t)(total_bits >> 3) + ((total_bits & 7) <= 5 ? 1 : 2);
    const size_t padded = align_up(out_len + 8, 8);
    if (padded > out_cap) throw HipError{hipErrorOutOfMemory, "encode: output buffer too small", (int)__LINE__};

    // ---- pass 2: pack ----------------------------------------------------------------------------------------
    HIP_TRY(hipMemsetAsync(d_out, 0, padded, s));
    HIP_TRY(hipMemcpyAsync(d_out, hw.bytes.data(), hw.bytes.size(), hipMemcpyHostToDevice, s));
    {
        Ctx::ProfScope prof(c, K_ENC_PACK, (u64)n * 9 + (u64)z * 4 + out_len);
        if (P.ascii) pack_kernel<true><<<tiles, 256, 0, s>>>(text, fs.owner, fs.flen, fs.fsrc, n, tab, P, A, tile_bits, base_bits, (u64*)d_out);
        else         pack_kernel<false><<<tiles, 256, 0, s>>>(text, fs.owner, fs.flen, fs.fsrc, n, tab, P, A, tile_bits, base_bits, (u64*)d_out);
        LAUNCH_CHECK();
    }
    terminator_kernel<<<1, 64, 0, s>>>(d_out, total_bits);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(s));      // hw.bytes must outlive the async copy

    st->factors = z;
    st->flen_min = h_sc.flen_min; st->flen_max = h_sc.flen_max; st->fdist_max = h_sc.fdist_max;
    st->out_bits = total_bits;
    st->sigma = ht.sigma;
    c.arena.release(mark);
    return out_len;
}

}  // namespace tdc
