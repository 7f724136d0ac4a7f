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
#include <algorithm>
#include <vector>

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


// A thread's ENC_PER_THREAD (= 8) consecutive positions start at a multiple of 8: two 16-byte loads per u32 array and one 8-byte
// load of the text instead of eight scalar loads each (the scalar form issues eight times the memory instructions for the same lines).
__device__ __forceinline__ void load8_u32(const u32* __restrict__ a, size_t p0, size_t n, u32 (&v)[ENC_PER_THREAD]) {
    if (p0 + ENC_PER_THREAD <= n) {
        const uint4 x = *(const uint4*)(a + p0), y = *(const uint4*)(a + p0 + 4);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
    } else {
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) v[j] = (p0 + j < n) ? a[p0 + j] : 0u;
    }
}
__device__ __forceinline__ void load8_u8(const u8* __restrict__ a, size_t p0, size_t n, u8 (&v)[ENC_PER_THREAD]) {
    if (p0 + ENC_PER_THREAD <= n) {
        const u64 x = *(const u64*)(a + p0);
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) v[j] = (u8)(x >> (8 * j));
    } else {
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) v[j] = (p0 + j < n) ? a[p0 + j] : (u8)0;
    }
}

// gaps between consecutive factors (LZSSCoding.hpp:28-38) + min/max factor length (LZSSFactors.hpp:41-47);
// stores the literal-run length at the first position of every run.
// store = false: only the maxima (a pack that takes the run lengths from the flatten records, pack_cls_kernel<true>)
__global__ __launch_bounds__(256) void gaps_kernel(const u32* __restrict__ fpos, const u32* __restrict__ flen_list, size_t z,
                                                    size_t n, u32* __restrict__ flen, EncScalars* __restrict__ sc, bool store) {
    __shared__ u32 sg[4], smax[4], smin[4];
    u32 gmax = 0, lmin = 0xFFFFFFFFu, lmax = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < z; i += stride) {
        // flen_list == nullptr: the lengths are read at the factor starts (run lengths are written at run starts, never there)
        const u32 p = fpos[i], l = flen_list ? flen_list[i] : flen[p];
        const u32 prev_end = (i == 0) ? 0u : fpos[i - 1] + (flen_list ? flen_list[i - 1] : flen[fpos[i - 1]]);
        const u32 gap = p - prev_end;
        if (gap && store) flen[prev_end] = gap;
        gmax = max(gmax, gap);
        lmin = min(lmin, l);
        lmax = max(lmax, l);
        if (i + 1 == z) {
            const u32 end = p + l;
            if ((size_t)end < n) { const u32 tail = (u32)(n - end); if (store) flen[end] = tail; gmax = max(gmax, tail); }
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
// the same from the class bytes of FactorSpace::cls (0 = literal): 2 bytes per position instead of 5, sixteen positions per load
__global__ __launch_bounds__(256) void literal_hist_cls_kernel(const u8* __restrict__ text, const u8* __restrict__ cls, size_t n,
                                                                u32* __restrict__ hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    for (size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; p < n; p += stride) {
        if (p + 16 <= n && ((((size_t)text) | ((size_t)cls)) & 15) == 0) {
            const uint4 t = *(const uint4*)(text + p), k = *(const uint4*)(cls + p);
            const u32 tw[4] = { t.x, t.y, t.z, t.w }, kw[4] = { k.x, k.y, k.z, k.w };
#pragma unroll
            for (int j = 0; j < 16; ++j) if (((kw[j >> 2] >> (8 * (j & 3))) & 0xFFu) == 0u) atomicAdd(&h[(tw[j >> 2] >> (8 * (j & 3))) & 0xFFu], 1u);
        } else {
            for (size_t q = p; q < n && q < p + 16; ++q) if (cls[q] == 0) atomicAdd(&h[text[q]], 1u);
        }
    }
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

// the same from class bytes (FactorSpace::cls; not for the ASCII coder, which needs the values): cl / prevcl = class of p / of p - 1
__device__ __forceinline__ u32 position_cost_cls(u32 cl, u32 prevcl, bool first, u8 ch, u32 p, const u8* __restrict__ clen,
                                                 const EncParams& P, const ArithDev& A) {
    if (cl == 0u) {
        u32 c;
        if (A.litidx) { const u32 k = A.litidx[p]; c = (A.amark[k] ? 64u : 0u) + (k == A.lc_index ? 128u : 0u); }
        else c = P.raw_literals ? 8u : (u32)clen[ch];
        if (first || prevcl != 0u) c += 1u + P.dbits;           // a literal run starts here: flen[p] holds its length
        return c;
    }
    if (cl == 2u) return P.W + P.lbits + ((first || prevcl != 0u) ? 1u : 0u);
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
        u32 own[ENC_PER_THREAD], fl[ENC_PER_THREAD];
        u8 ch[ENC_PER_THREAD];
        load8_u32(owner, p0, n, own); load8_u32(flen, p0, n, fl); load8_u8(text, p0, n, ch);
        u32 prev = (p0 == 0) ? 0u : owner[p0 - 1];
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) {
            const size_t p = p0 + j;
            if (p < n) {
                sum += position_cost<ASCII>(own[j], prev, p == 0, fl[j], ch[j], (u32)p, clen, fsrc, P, A);
                prev = own[j];
            }
        }
    }
    sum = wave_reduce_sum(sum);
    if (lane_id() == 0) sm[wave_id()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tile_bits[blockIdx.x] = (u64)sm[0] + sm[1] + sm[2] + sm[3];
}

// tile_starts (optional): the number of factors that start in the tile (their scan gives pack_cls_kernel<true> the rank of a tile's first factor)
__global__ __launch_bounds__(256) void tile_bits_cls_kernel(const u8* __restrict__ text, const u8* __restrict__ cls, size_t n, CodeTable tab, EncParams P,
                                                             ArithDev A, u64* __restrict__ tile_bits, u32* __restrict__ tile_starts) {
    __shared__ u8 clen[256];
    __shared__ u32 sm[4], sc[4];
    clen[threadIdx.x] = tab.len[threadIdx.x];
    __syncthreads();
    const size_t p0 = (size_t)blockIdx.x * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    u32 sum = 0, starts = 0;
    if (p0 < n) {
        u8 cl[ENC_PER_THREAD], ch[ENC_PER_THREAD];
        load8_u8(cls, p0, n, cl); load8_u8(text, p0, n, ch);
        u32 prev = (p0 == 0) ? 0u : (u32)cls[p0 - 1];
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) {
            const size_t p = p0 + j;
            if (p < n) {
                sum += position_cost_cls(cl[j], prev, p == 0, ch[j], (u32)p, clen, P, A);
                starts += (cl[j] == 2u) ? 1u : 0u;
                prev = cl[j];
            }
        }
    }
    sum = wave_reduce_sum(sum);
    starts = wave_reduce_sum(starts);
    if (lane_id() == 0) { sm[wave_id()] = sum; sc[wave_id()] = starts; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_bits[blockIdx.x] = (u64)sm[0] + sm[1] + sm[2] + sm[3];
        if (tile_starts) tile_starts[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
    }
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
                                                    u64 base_bits, u64* __restrict__ out, u32 tile0) {
    __shared__ u8 clen[256];
    __shared__ u64 code[256];
    __shared__ u32 sm[5];
    clen[threadIdx.x] = tab.len[threadIdx.x];
    code[threadIdx.x] = tab.code[threadIdx.x];
    __syncthreads();
    const u32 tile = blockIdx.x + tile0;
    const size_t p0 = (size_t)tile * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    u32 own[ENC_PER_THREAD], fl[ENC_PER_THREAD];
    u8 ch[ENC_PER_THREAD];
    u32 prev0 = 0, sum = 0;
    if (p0 < n) {
        load8_u32(owner, p0, n, own); load8_u32(flen, p0, n, fl); load8_u8(text, p0, n, ch);
        prev0 = (p0 == 0) ? 0u : owner[p0 - 1];
        u32 prev = prev0;
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) {
            const size_t p = p0 + j;
            if (p < n) {
                sum += position_cost<ASCII>(own[j], prev, p == 0, fl[j], ch[j], (u32)p, clen, fsrc, P, A);
                prev = own[j];
            }
        }
    }
    u32 total;
    const u32 excl = block_exclusive_sum<u32, 4>(sum, sm, total);
    if (p0 >= n) return;
    BitSink sink;
    sink.out = out;
    sink.pos = base_bits + tile_off[tile] + excl;
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

// pack_kernel<false> on the class bytes of FactorSpace::cls instead of owner[] (3 bytes less per position in each of the two passes of a tile).
// REC: length and final source of a factor come from the records of the flatten stage (flatten.hip: {pos, len, original source, final
// source}, indexed by the factor's rank in position order; the rank of a tile's first factor from the scanned start counts of
// tile_bits_cls_kernel, inside the tile by the scan that also gives the bit offsets), and the length of a literal run from the position
// of the factor that follows it: no flen[] / fsrc[] stream at all (8 + 8 bytes per position), and the flatten stage need not write its
// results back into fsrc[].
template <bool REC>
__global__ __launch_bounds__(256) void pack_cls_kernel(const u8* __restrict__ text, const u8* __restrict__ cls,
                                                        const u32* __restrict__ flen, const u32* __restrict__ fsrc, size_t n,
                                                        CodeTable tab, EncParams P, ArithDev A, const u64* __restrict__ tile_off,
                                                        u64 base_bits, u64* __restrict__ out, u32 tile0,
                                                        const uint4* __restrict__ rec, const u32* __restrict__ tile_rank, u32 z) {
    __shared__ u8 clen[256];
    __shared__ u64 code[256];
    __shared__ u64 sm[5];
    clen[threadIdx.x] = tab.len[threadIdx.x];
    code[threadIdx.x] = tab.code[threadIdx.x];
    __syncthreads();
    const u32 tile = blockIdx.x + tile0;
    const size_t p0 = (size_t)tile * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    u32 fl[ENC_PER_THREAD];
    u8 cl[ENC_PER_THREAD], ch[ENC_PER_THREAD];
    u32 prev0 = 0, sum = 0, starts = 0;
    if (p0 < n) {
        load8_u8(cls, p0, n, cl); load8_u8(text, p0, n, ch);
        if (!REC) load8_u32(flen, p0, n, fl);
        prev0 = (p0 == 0) ? 0u : (u32)cls[p0 - 1];
        u32 prev = prev0;
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) {
            const size_t p = p0 + j;
            if (p < n) {
                sum += position_cost_cls(cl[j], prev, p == 0, ch[j], (u32)p, clen, P, A);
                if (REC) starts += (cl[j] == 2u) ? 1u : 0u;
                prev = cl[j];
            }
        }
    }
    u64 total;
    const u64 ex = block_exclusive_sum<u64, 4>((u64)sum | ((u64)starts << 40), sm, total);     // (bits of a tile < 2^40, starts <= 2048)
    const u64 excl = ex & ((1ull << 40) - 1ull);
    if (p0 >= n) return;
    u32 k = REC ? tile_rank[tile] + (u32)(ex >> 40) : 0u;          // rank of the next factor that starts at or behind the thread's first position
    BitSink sink;
    sink.out = out;
    sink.pos = base_bits + tile_off[tile] + excl;
    sink.acc = 0;
    sink.cnt = 0;
    u32 prev = prev0;
#pragma unroll
    for (int j = 0; j < ENC_PER_THREAD; ++j) {
        const size_t p = p0 + j;
        if (p < n) {
            const u32 o = cl[j];
            if (o == 0u) {
                if (p == 0 || prev != 0u) {                                               // LZSSCoding.hpp:62-68, :83-86 (a literal run starts)
                    const u32 run = REC ? ((k < z ? rec[k].x : (u32)n) - (u32)p) : fl[j];
                    sink.append(1, 1); sink.append(run, P.dbits);
                }
                if (A.litidx) {                                                           // ArithmeticCoder.hpp:96-104, :151-155
                    const u32 q = A.litidx[p];
                    if (A.amark[q]) sink.append(A.fval[q], 64);
                    if (q == A.lc_index) { sink.append(A.pp_lb, 64); sink.append(~0ull, 64); }
                }
                else if (P.raw_literals) sink.append(ch[j], 8);                           // HuffmanCoder.hpp:565-566
                else sink.append(code[ch[j]], clen[ch[j]]);                               // :568 huffman_encode
            } else if (o == 2u) {
                if (p == 0 || prev != 0u) sink.append(0, 1);                              // LZSSCoding.hpp:57-59
                if (REC) {
                    const uint4 r = rec[k++];
                    sink.append(r.w, P.W);                                                // :77 (the flattened source)
                    sink.append(r.y - P.flen_min, P.lbits);                               // :78
                } else {
                    sink.append(fsrc[p], P.W);                                            // :77
                    sink.append(fl[j] - P.flen_min, P.lbits);                             // :78
                }
            }
            prev = o;
        }
    }
    sink.flush();
}

// out[q] = excl[(tiles * (q + 1)) / parts] for q < parts - 1 (bit offset at which chunk q of the pack ends), out[parts - 1] unused
__global__ void pick_u64_kernel(const u64* __restrict__ excl, u32 tiles, u32 parts, u64* __restrict__ out) {
    const u32 q = threadIdx.x;
    if (q < parts) out[q] = (q + 1 < parts) ? excl[(u32)((u64)tiles * (q + 1) / parts)] : 0ull;
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
// factor costs '0' + two integers of up to 10 digits and ':' each; SLECoder (coder 3): at most 13 bits per literal, 67 per factor,
// and a ranking header of at most 1024 symbols of 10 bytes
size_t encode_bound(size_t n) { return 12 * n + 4096; }
size_t encode_bound_coder(size_t n, int coder) { return ((coder & 0xFF) == 2 ? 24 : 12) * n + ((coder & 0xFF) == 3 ? 32768 : 4096); }

size_t encode_huff(Ctx& c, const u8* text, size_t n, FactorSpace fs, u8* d_out, size_t out_cap, EncodeStats* st) {
    return encode_stream(c, text, n, fs, 0, d_out, out_cap, st, nullptr);
}

// factor list in position order, gaps + min/max lengths (run lengths stored at run starts), literal histogram
struct EncPrelude { size_t z; u32 hist[256]; EncScalars sc; };
// r_hist (optional): 260 words -- the histogram, then the scalars (one read-back for both)
static void encode_prelude_launch(Ctx& c, const u8* text, size_t n, FactorSpace& fs, EncPrelude& pre, u32* r_hist, bool store_runs, u32** o_hist, EncScalars** o_sc) {
    hipStream_t s = c.stream;
    u32* fpos = fs.have_list ? fs.fpos : c.arena.get<u32>(n);
    u32* flist = fs.have_list ? fs.flenl : c.arena.get<u32>(n);        // (nullptr: the lengths are read at the factor starts)
    const size_t z = fs.have_list ? fs.nfact : extract_factors(c, n, fs, fpos, nullptr, flist, n);
    u32* d_hist = r_hist ? r_hist : c.arena.get<u32>(256 + 4);
    EncScalars* d_sc = (EncScalars*)(d_hist + 256);
    EncScalars h_sc = { 0xFFFFFFFFu, 0u, 0u, 0u };          // LZSSFactors.hpp:33-38 : INDEX_MAX / 0
    static_assert(sizeof(EncScalars) == 16, "EncScalars: four words");
    HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(h_sc), s));       // (two fills instead of a copy from pageable memory)
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_sc, 0xFFFFFFFFu, 1, s));
    HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), s));
    if (z) {
        unsigned g = cdiv(z, 256); if (g > 2048) g = 2048;
        Ctx::ProfScope prof(c, K_ENC_GAPS, (u64)z * 12);
        gaps_kernel<<<g, 256, 0, s>>>(fpos, flist, z, n, fs.flen, d_sc, store_runs || !flist);
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
        if (fs.have_cls) literal_hist_cls_kernel<<<g, 256, 0, s>>>(text, fs.cls, n, d_hist);
        else literal_hist_kernel<<<g, 256, 0, s>>>(text, plain_owner(fs), n, d_hist);
        LAUNCH_CHECK();
    }
    pre.z = z;
    *o_hist = d_hist; *o_sc = d_sc;
}
static void encode_prelude(Ctx& c, const u8* text, size_t n, FactorSpace& fs, EncPrelude& pre) {
    u32* d_hist; EncScalars* d_sc;
    encode_prelude_launch(c, text, n, fs, pre, nullptr, true, &d_hist, &d_sc);
    u32 h[260];
    c.read_n(d_hist, h, 260);
    memcpy(pre.hist, h, sizeof(pre.hist));
    memcpy(&pre.sc, h + 256, sizeof(pre.sc));
}

static size_t encode_sle(Ctx& c, const u8* text, size_t n, FactorSpace fs, u32 k, u8* d_out, size_t out_cap, EncodeStats* st);

// The encoder in two halves.  The first one -- gaps, literal histogram, coder header, bits per tile, their scan -- needs the factors'
// positions and lengths but not their sources, so the caller may run it while the factors are still being flattened (api.hip,
// run_factorize: on the copy stream, next to the first flatten round); the second half packs the bits and needs everything.
#ifndef TDC_PACK_CH
#define TDC_PACK_CH 16     // (8: 0.3 ms more behind the last chunk at 2e9 B, 32: the same as 16)
#endif
constexpr u32 PACK_CH = TDC_PACK_CH;
struct EncodeEarly {
    // scratch; reserved ahead of the flatten stage when the first half runs inside it (arena order: these, then flatten's lists)
    u32* d_hist = nullptr; u64* tile_bits = nullptr; u64* d_tp = nullptr;   // d_hist: 256 counters + the EncScalars; d_tp: total, then PACK_CH chunk ends
    // the three steps of the first half: A gaps + histogram, B code table + bits per tile + scans, C collect; seq_*: read-backs under way
    // (Ctx::publish_async; 0: read synchronously)
    int step = 0;
    u32 seq_a = 0, seq_b = 0;
    bool may_overlap = false;
    u32* tile_rank = nullptr;      // rank of the first factor of every tile (only with rec)
    uint4* rec = nullptr;          // the records of the flatten stage, kept for the pack (flatten_factors fills them; z_rec entries)
    size_t z_rec = 0;
    unsigned tiles = 0;
    bool done = false;
    // results of the first half
    size_t z = 0;
    EncScalars sc;
    HuffTable ht;
    HostBitWriter hw;
    ArithDev A;
    EncParams P;
    CodeTable tab;
    u64 base_bits = 0, total_bits = 0;
    size_t out_len = 0;
    bool overlap = false;
    u64 h_end[PACK_CH];
};

static void encode_reserve(Ctx& c, size_t n, EncodeEarly& E, size_t z_rec = 0) {
    E.tiles = cdiv(n, ENC_TILE);
    if (z_rec) {
        E.z_rec = z_rec;
        E.rec = (uint4*)c.arena.alloc(z_rec * sizeof(uint4));
        E.tile_rank = c.arena.get<u32>(E.tiles + 1);
    }
    E.d_hist = c.arena.get<u32>(256 + 4);
    E.tile_bits = c.arena.get<u64>(E.tiles + 1);
    E.d_tp = c.arena.get<u64>(1 + PACK_CH);
}

// Blocks 1 and 2 of the mapped host area carry the read-backs of steps A and B (free here: they belong to the one-workgroup levels of the
// factorizer), so a step never waits for the step before it unless it has to.
static void encode_step_a(Ctx& c, const u8* text, size_t n, FactorSpace& fs, EncodeEarly& E) {
    EncPrelude pre;
    u32* d_hist; EncScalars* d_sc;
    encode_prelude_launch(c, text, n, fs, pre, E.d_hist, E.rec == nullptr, &d_hist, &d_sc);
    E.z = pre.z;
    E.seq_a = c.publish_async(E.d_hist, 260 * sizeof(u32), 1);
    E.step = 1;
}

static void encode_step_b(Ctx& c, const u8* text, size_t n, FactorSpace& fs, int coder, EncodeEarly& E) {
    hipStream_t s = c.stream;
    u32 h[260];
    if (E.seq_a) c.publish_wait(E.seq_a, h, sizeof(h), 1); else c.read_n(E.d_hist, h, 260);
    const u32* h_hist = h;
    memcpy(&E.sc, h + 256, sizeof(E.sc));
    const EncScalars& h_sc = E.sc;

    // ---- host: coder header (HuffmanCoder::Encoder ctor :526-547 / ArithmeticCoder::Encoder ctor :158-164),
    //      then the fields of LZSSCoding.hpp:47-50
    HuffTable& ht = E.ht;
    HostBitWriter& hw = E.hw;
    E.A = { nullptr, nullptr, nullptr, 0, 0 };
    if (coder == 2) {
        memset(&ht, 0, sizeof(ht));                                        // ASCIICoder writes no header
    } else if (coder == 0) {
        if (!c.huff_ok)
            throw HipError{hipErrorUnknown, "Huffman self-check failed: this C++ library breaks ties differently from the reference build, coder=huff streams would differ", (int)__LINE__};
        build_huffman_table(h_hist, &ht);
        write_huffman_header(hw, ht);
    } else {
        ArithModel am;
        if (!arith_build_model(h_hist, &am, hw))
            throw HipError{hipErrorInvalidValue, "arithmetic coder: all literal bytes >= 1 are absent (the reference divides by zero)", -1};
        ArithPlan plan;
        arith_prepare(c, text, n, plain_owner(fs), am, &plan);
        E.A.litidx = plan.litidx; E.A.amark = plan.amark; E.A.fval = plan.fval; E.A.lc_index = plan.lc_index; E.A.pp_lb = plan.pp_lb;
    }
    EncParams& P = E.P;
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
    E.base_bits = hw.nbits;
    memcpy(E.tab.code, ht.code_of, sizeof(E.tab.code));
    memcpy(E.tab.len, ht.len_of, sizeof(E.tab.len));

    // ---- pass 1: bits per tile, scan -------------------------------------------------------------------------
    const unsigned tiles = E.tiles;
    u64* tile_bits = E.tile_bits;
    {
        Ctx::ProfScope prof(c, K_ENC_TILE_BITS, (u64)n * 9);
        if (P.ascii) tile_bits_kernel<true><<<tiles, 256, 0, s>>>(text, plain_owner(fs), fs.flen, fs.fsrc, n, E.tab, P, E.A, tile_bits);
        else if (fs.have_cls) tile_bits_cls_kernel<<<tiles, 256, 0, s>>>(text, fs.cls, n, E.tab, P, E.A, tile_bits, E.tile_rank);
        else         tile_bits_kernel<false><<<tiles, 256, 0, s>>>(text, plain_owner(fs), fs.flen, fs.fsrc, n, E.tab, P, E.A, tile_bits);
        LAUNCH_CHECK();
    }
    exclusive_sum_u64(c, tile_bits, tile_bits, tiles, E.d_tp);
    if (E.tile_rank) exclusive_sum_u32(c, E.tile_rank, E.tile_rank, tiles, nullptr);
    // With a host destination (end-to-end entry point) the pack runs in PACK_CH chunks of tiles (second half): the bit offsets at
    // which the chunks end travel with the total
    E.may_overlap = c.d2h_host && tiles >= 64 * PACK_CH && c.copy_stream;
    pick_u64_kernel<<<1, 64, 0, s>>>(tile_bits, tiles, PACK_CH, E.d_tp + 1);
    LAUNCH_CHECK();
    E.seq_b = c.publish_async(E.d_tp, (1 + PACK_CH) * sizeof(u64), 2);
    E.step = 2;
}

static void encode_step_c(Ctx& c, EncodeEarly& E) {
    u64 h_tp[1 + PACK_CH] = { 0 };
    if (E.seq_b) c.publish_wait(E.seq_b, h_tp, sizeof(h_tp), 2); else c.read_n(E.d_tp, h_tp, 1 + PACK_CH);
    const bool may_overlap = E.may_overlap;
    E.total_bits = E.base_bits + h_tp[0];
    E.out_len = (size_t)(E.total_bits >> 3) + ((E.total_bits & 7) <= 5 ? 1 : 2);
    E.overlap = may_overlap && E.out_len <= c.d2h_cap;
    for (u32 q = 0; q < PACK_CH; ++q) E.h_end[q] = h_tp[1 + q];
    E.step = 3;
    E.done = true;
}

// the next step(s) of the first half: one per call (finish = false), or everything that is left
static void encode_first_half(Ctx& c, const u8* text, size_t n, FactorSpace& fs, int coder, EncodeEarly& E, bool finish = true) {
    do {
        if (E.step == 0) encode_step_a(c, text, n, fs, E);
        else if (E.step == 1) encode_step_b(c, text, n, fs, coder, E);
        else if (E.step == 2) encode_step_c(c, E);
    } while (finish && E.step < 3);
}

EncodeEarly* encode_early_reserve(Ctx& c, size_t n, size_t z_rec) {
    EncodeEarly* E = new EncodeEarly();
    try { encode_reserve(c, n, *E, z_rec); } catch (...) { delete E; throw; }
    return E;
}
void* encode_early_rec(EncodeEarly* E) { return E->rec; }
void encode_early_run(Ctx& c, const u8* text, size_t n, FactorSpace fs, int coder, EncodeEarly* E, bool finish) { encode_first_half(c, text, n, fs, coder, *E, finish); }
void encode_early_free(EncodeEarly* E) { delete E; }

size_t encode_stream(Ctx& c, const u8* text, size_t n, FactorSpace fs, int coder, u8* d_out, size_t out_cap, EncodeStats* st, EncodeEarly* early) {
    EncodeStats local;
    if (!st) st = &local;
    *st = EncodeStats();
    if ((coder & 0xFF) == 3) return encode_sle(c, text, n, fs, (u32)(coder >> 8), d_out, out_cap, st);
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();

    EncodeEarly mine;
    // An early first half that did not finish must not be replaced silently: with its flatten records kept (rec != nullptr) the
    // flattened sources were never written back to fs.fsrc, a private first half would pack the unflattened ones.
    if (early && !early->done)
        throw HipError{hipErrorUnknown, "encode: the first half started inside the flatten stage did not complete", (int)__LINE__};
    if (!early) {
        early = &mine;
        encode_reserve(c, n, mine);
        encode_first_half(c, text, n, fs, coder, mine);
    }
    EncodeEarly& E = *early;
    const size_t z = E.z;
    const EncScalars h_sc = E.sc;
    const EncParams P = E.P;
    const ArithDev A = E.A;
    const CodeTable& tab = E.tab;
    HostBitWriter& hw = E.hw;
    const u64 base_bits = E.base_bits, total_bits = E.total_bits;
    const unsigned tiles = E.tiles;
    const u64* tile_bits = E.tile_bits;
    const size_t out_len = E.out_len;
    const size_t padded = align_up(out_len + 8, 8);
    if (padded > out_cap) throw HipError{hipErrorOutOfMemory, "encode: output buffer too small", (int)__LINE__};

    // ---- pass 2: pack ----------------------------------------------------------------------------------------
    HIP_TRY(hipMemsetAsync(d_out, 0, padded, s));
    if (c.pinned_hdr && hw.bytes.size() <= Ctx::PINNED_HDR) {        // (through the context's page-locked block: a copy from pageable memory drains the stream)
        memcpy(c.pinned_hdr, hw.bytes.data(), hw.bytes.size());
        HIP_TRY(hipMemcpyAsync(d_out, c.pinned_hdr, hw.bytes.size(), hipMemcpyHostToDevice, s));
    } else HIP_TRY(hipMemcpyAsync(d_out, hw.bytes.data(), hw.bytes.size(), hipMemcpyHostToDevice, s));
    {
        Ctx::ProfScope prof(c, K_ENC_PACK, (u64)n * 9 + (u64)z * 4 + out_len);
        // in PACK_CH chunks of tiles when the stream goes to the host: once a chunk is done the bytes in front of its last
        // (possibly shared) 64-bit word are final and start their way to the host on the copy stream, while the next chunk is
        // being packed.
        constexpr u32 CH = PACK_CH;
        const bool overlap = E.overlap;
        const u64* h_end = E.h_end;
        size_t copied = 0;
        for (u32 q = 0; q < (overlap ? CH : 1u); ++q) {
            const u32 t0 = overlap ? (u32)((u64)tiles * q / CH) : 0u, t1 = overlap ? (u32)((u64)tiles * (q + 1) / CH) : tiles;
            if (P.ascii) pack_kernel<true><<<t1 - t0, 256, 0, s>>>(text, plain_owner(fs), fs.flen, fs.fsrc, n, tab, P, A, tile_bits, base_bits, (u64*)d_out, t0);
            else if (fs.have_cls && E.rec) pack_cls_kernel<true><<<t1 - t0, 256, 0, s>>>(text, fs.cls, fs.flen, fs.fsrc, n, tab, P, A, tile_bits, base_bits, (u64*)d_out, t0, E.rec, E.tile_rank, (u32)E.z_rec);
            else if (fs.have_cls) pack_cls_kernel<false><<<t1 - t0, 256, 0, s>>>(text, fs.cls, fs.flen, fs.fsrc, n, tab, P, A, tile_bits, base_bits, (u64*)d_out, t0, nullptr, nullptr, 0u);
            else         pack_kernel<false><<<t1 - t0, 256, 0, s>>>(text, plain_owner(fs), fs.flen, fs.fsrc, n, tab, P, A, tile_bits, base_bits, (u64*)d_out, t0);
            LAUNCH_CHECK();
            if (overlap && q + 1 < CH) {
                const size_t safe = (size_t)((base_bits + h_end[q]) / 64) * 8;          // bytes in front of the word the next chunk may still touch
                if (safe > copied) {
                    HIP_TRY(hipEventRecord(c.ev_copy[q], s));
                    HIP_TRY(hipStreamWaitEvent(c.copy_stream, c.ev_copy[q], 0));
                    HIP_TRY(hipMemcpyAsync(c.d2h_host + copied, d_out + copied, safe - copied, hipMemcpyDeviceToHost, c.copy_stream));
                    copied = safe;
                    c.d2h_done = copied;                     // (an error further down must still wait for this copy: api.hip SinkGuard)
                }
            }
        }
        c.d2h_done = copied;
    }
    terminator_kernel<<<1, 64, 0, s>>>(d_out, total_bits);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(s));      // hw.bytes must outlive the async copy

    st->factors = z;
    st->flen_min = h_sc.flen_min; st->flen_max = h_sc.flen_max; st->fdist_max = h_sc.fdist_max;
    st->out_bits = total_bits;
    st->sigma = E.ht.sigma;
    c.arena.release(mark);
    return out_len;
}

// ============================================================================================================
// coder = SLECoder (coders/SLECoder.hpp): literal bytes + the eta most frequent k-mers of the literal runs form one
// alphabet, ranked by (count descending, symbol ascending) (util/Counter.hpp:44-70); a rank is written in a fixed class
// code that depends on sigma_bits (:182-245); the factor length (a MinDistributedRange, LZSSCoding.hpp:42) has its own
// class code (:274-296).  Inside a literal run the encoder keeps the last <= k literals in a buffer (:55-66, :249-266):
// a full buffer whose k-mer is ranked is written as ONE symbol and the buffer restarts; otherwise the oldest byte leaves
// as a single symbol; every non-literal write flushes the buffer as single symbols (:172-180).
//
// The buffer fill s in 0..k is the only state (the buffer content is the last s text bytes), and a literal position i
// maps s -> fired(i) ? 0 : min(s+1, k) with fired(i) = ranked(i) && s >= k-1; a non-literal position maps every s to 0.
// These maps compose associatively, so the fill at every position comes from a scan of 4-entry maps: per tile of 2048
// positions, over the tile composites (one workgroup), and again inside the tile.  fired(i) makes i-k+1 the start of
// a k-mer symbol and i-k+2..i its interior (no bits); every other literal is a single symbol.  Symbols are emitted in
// text order (a rolled-out byte is older than everything in the buffer), so the stream is again a per-position cost.
// ============================================================================================================
constexpr u32 SLE_MAX_KMERS = 1024;           // eta <= 2^(8+2) - 129
constexpr u16 SLE_LIT = 0x8000, SLE_RANKED = 0x4000, SLE_FIRED = 0x2000, SLE_RANK_MASK = 0x03FF;

struct SleDev {
    u16 rank_byte[256];
    u32 sb, k, nk;
    u32 f0, f1;              // state maps of a literal position without / with a ranked k-mer ending there (4 bits per state, 8 states)
    const u64* kval;         // ranked k-mers (byte string as an integer, first byte most significant), ascending
    const u16* krank;
};

__device__ __forceinline__ u64 sle_key(const u8* __restrict__ text, size_t p, u32 k) {     // compile_kmer :18-26 without the marker byte
    u64 x = 0;
    for (u32 j = 0; j < k; ++j) x = (x << 8) | text[p - (k - 1) + j];
    return x;
}

// Encoder ctor :96-120: every k-mer window that lies inside one literal run is counted
__global__ __launch_bounds__(256) void sle_kmer_count_kernel(const u8* __restrict__ text, const u32* __restrict__ owner, size_t n, u32 k,
                                                              u32* __restrict__ cnt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x + (k - 1); p < n; p += stride) {
        bool ok = true;
        for (u32 j = 0; j < k; ++j) ok = ok && owner[p - j] == NONE32;
        if (ok) atomicAdd(&cnt[sle_key(text, p, k)], 1u);
    }
}
// k > 3 (no 256^k table): the windows themselves, to be sorted and run-length counted
__global__ __launch_bounds__(256) void sle_kmer_keys_kernel(const u8* __restrict__ text, const u32* __restrict__ owner, size_t n, u32 k,
                                                             u64* __restrict__ keys, u32* __restrict__ d_count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p0 = (size_t)blockIdx.x * blockDim.x + (k - 1); p0 < n; p0 += stride) {
        const size_t p = p0 + threadIdx.x;
        bool ok = p < n;
        for (u32 j = 0; j < k && ok; ++j) ok = owner[p - j] == NONE32;
        const u64 m = __ballot(ok);
        if (m == 0) continue;
        u32 base = 0;
        if (lane_id() == 0) base = atomicAdd(d_count, (u32)__popcll(m));
        base = __shfl(base, 0);
        if (ok) keys[base + __popcll(m & ((1ull << lane_id()) - 1))] = sle_key(text, p, k);
    }
}
__global__ void sle_run_heads_kernel(const u64* __restrict__ keys, size_t m, u8* __restrict__ head) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
// run j = [start[j], start[j+1]): sort key ~count (the k-mers are already ascending, the count sort is stable)
__global__ void sle_run_counts_kernel(const u32* __restrict__ start, u32 runs, u32 total, u64* __restrict__ ckey, u32* __restrict__ cval) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= runs) return;
    const u32 end = (j + 1 < runs) ? start[j + 1] : total;
    ckey[j] = (u64)(0xFFFFFFFFu - (end - start[j]));
    cval[j] = start[j];
}
__global__ void sle_top_kernel(const u64* __restrict__ ckey, const u32* __restrict__ cval, u32 take, const u64* __restrict__ keys,
                               u64* __restrict__ out_kmer, u32* __restrict__ out_cnt) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= take) return;
    out_kmer[j] = keys[cval[j]];
    out_cnt[j] = 0xFFFFFFFFu - (u32)ckey[j];
}
// distinct k-mers as sort keys: ascending key order == Counter::getSorted (count descending, k-mer ascending)
__global__ __launch_bounds__(256) void sle_kmer_compact_kernel(const u32* __restrict__ cnt, size_t tsize, u64* __restrict__ keys,
                                                                u32* __restrict__ d_count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x; i0 < tsize; i0 += stride) {
        const size_t i = i0 + threadIdx.x;
        const u32 v = (i < tsize) ? cnt[i] : 0u;
        const u64 m = __ballot(v != 0);
        if (m == 0) continue;
        u32 base = 0;
        if (lane_id() == 0) base = atomicAdd(d_count, (u32)__popcll(m));
        base = __shfl(base, 0);
        if (v) keys[base + __popcll(m & ((1ull << lane_id()) - 1))] = ((u64)(0xFFFFFFFFu - v) << 24) | (u64)i;
    }
}

// per position: literal? is the k-mer ending here ranked (and which rank)?
__global__ __launch_bounds__(256) void sle_minfo_kernel(const u8* __restrict__ text, const u32* __restrict__ owner, size_t n, SleDev D,
                                                         u16* __restrict__ minfo) {
    __shared__ u64 kv[SLE_MAX_KMERS];
    __shared__ u16 kr[SLE_MAX_KMERS];
    for (u32 i = threadIdx.x; i < D.nk; i += blockDim.x) { kv[i] = D.kval[i]; kr[i] = D.krank[i]; }
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        u16 v = 0;
        if (owner[p] == NONE32) {
            v = SLE_LIT;
            if (D.nk && p + 1 >= D.k) {
                const u64 key = sle_key(text, p, D.k);
                u32 lo = 0, hi = D.nk;                       // first entry >= key
                while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (kv[mid] < key) lo = mid + 1; else hi = mid; }
                if (lo < D.nk && kv[lo] == key) v |= SLE_RANKED | kr[lo];
            }
        }
        minfo[p] = v;
    }
}

// state maps: 4 bits per state, states 0..7 (k <= 7)
constexpr u32 SLE_FN_ID = 0x76543210u;
__device__ __forceinline__ u32 sle_fn_apply(u32 f, u32 s) { return (f >> (4 * s)) & 15u; }
__device__ __forceinline__ u32 sle_fn_compose(u32 first, u32 then) {
    u32 r = 0;
#pragma unroll
    for (u32 s = 0; s < 8; ++s) r |= sle_fn_apply(then, sle_fn_apply(first, s)) << (4 * s);
    return r;
}
__device__ __forceinline__ u32 sle_fn_of(u16 v, const SleDev& D) { return !(v & SLE_LIT) ? 0u : ((v & SLE_RANKED) ? D.f1 : D.f0); }

// composite of the thread's ENC_PER_THREAD positions; exclusive scan over the workgroup -> map from the tile's entry
// state to the state in front of the thread's first position; *tile_total = composite of the whole tile
__device__ __forceinline__ u32 sle_block_scan(u32 mine, u32* smem, u32* tile_total) {
    const int lane = lane_id(), w = wave_id();
    u32 inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d);
        if (lane >= d) inc = sle_fn_compose(t, inc);
    }
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    u32 pre = SLE_FN_ID;
    for (int i = 0; i < w; ++i) pre = sle_fn_compose(pre, smem[i]);
    u32 excl = __shfl_up(inc, 1);
    if (lane == 0) excl = SLE_FN_ID;
    if (tile_total) {
        u32 tot = SLE_FN_ID;
        for (int i = 0; i < 4; ++i) tot = sle_fn_compose(tot, smem[i]);
        *tile_total = tot;
    }
    return sle_fn_compose(pre, excl);
}
__device__ __forceinline__ u32 sle_thread_fn(const u16* __restrict__ minfo, size_t p0, size_t n, const SleDev& D) {
    u32 f = SLE_FN_ID;
#pragma unroll
    for (int j = 0; j < ENC_PER_THREAD; ++j) {
        const size_t p = p0 + j;
        if (p < n) f = sle_fn_compose(f, sle_fn_of(minfo[p], D));
    }
    return f;
}
__global__ __launch_bounds__(256) void sle_tile_fn_kernel(const u16* __restrict__ minfo, size_t n, SleDev D, u32* __restrict__ tile_fn) {
    __shared__ u32 sm[4];
    const size_t p0 = (size_t)blockIdx.x * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    u32 tot;
    sle_block_scan(sle_thread_fn(minfo, p0, n, D), sm, &tot);
    if (threadIdx.x == 0) tile_fn[blockIdx.x] = tot;
}
// one workgroup: state in front of every tile (the text starts with an empty buffer)
__global__ __launch_bounds__(1024) void sle_tile_scan_kernel(const u32* __restrict__ tile_fn, u32 tiles, u32* __restrict__ tile_in) {
    __shared__ u32 f[1024];
    __shared__ u32 sin[1024];
    const u32 chunk = (tiles + 1023u) / 1024u;
    const u32 t0 = threadIdx.x * chunk, t1 = min(tiles, t0 + chunk);
    u32 comp = SLE_FN_ID;
    for (u32 t = t0; t < t1; ++t) comp = sle_fn_compose(comp, tile_fn[t]);
    f[threadIdx.x] = comp;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 state = 0;
        for (u32 i = 0; i < 1024; ++i) { sin[i] = state; state = sle_fn_apply(f[i], state); }
    }
    __syncthreads();
    u32 state = sin[threadIdx.x];
    for (u32 t = t0; t < t1; ++t) { tile_in[t] = state; state = sle_fn_apply(tile_fn[t], state); }
}
// marks the positions where a k-mer symbol is written (its last byte)
__global__ __launch_bounds__(256) void sle_fire_kernel(u16* __restrict__ minfo, size_t n, SleDev D, const u32* __restrict__ tile_in) {
    __shared__ u32 sm[4];
    const size_t p0 = (size_t)blockIdx.x * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    const u32 pre = sle_block_scan(sle_thread_fn(minfo, p0, n, D), sm, nullptr);
    u32 s = sle_fn_apply(pre, tile_in[blockIdx.x]);
#pragma unroll
    for (int j = 0; j < ENC_PER_THREAD; ++j) {
        const size_t p = p0 + j;
        if (p >= n) break;
        const u16 v = minfo[p];
        if (!(v & SLE_LIT)) s = 0;
        else if ((v & SLE_RANKED) && s + 1 >= D.k) { minfo[p] = v | SLE_FIRED; s = 0; }      // :258-265
        else s = min(s + 1, D.k);                                                            // kmer_roll :55-66
    }
}

__device__ __forceinline__ void sle_sym(u32 r, u32 sb, u64& code, u32& len) {                // encode_sym :182-245
    if (sb < 4) { code = r; len = sb; }
    else if (sb < 6) {
        if (r < 4) { code = r; len = 3; } else { code = (1u << sb) | r; len = sb + 1; }
    } else if (sb == 6) {
        if (r < 8) { code = r; len = 5; }
        else if (r < 16) { code = (1u << 3) | (r - 8); len = 5; }
        else if (r < 32) { code = (2u << 4) | (r - 16); len = 6; }
        else { code = (3u << 6) | r; len = 8; }
    } else {
        if (r < 16) { code = ((r >> 2) << 2) | (r & 3); len = 5; }                           // classes 0..3: 2 value bits
        else if (r < 40) { const u32 cls = 4 + ((r - 16) >> 3); code = (cls << 3) | ((r - 16) & 7); len = 6; }
        else { code = (7u << sb) | r; len = 3 + sb; }
    }
}
__device__ __forceinline__ void sle_mdr(u32 v, u32 bits, u64& code, u32& len) {              // encode(v, MinDistributedRange) :274-296
    if (bits <= 5) { code = v; len = bits; }
    else if (v < 8) { code = v; len = 5; }
    else if (v < 16) { code = (1u << 3) | (v - 8); len = 5; }
    else if (v < 32) { code = (2u << 4) | (v - 16); len = 6; }
    else { code = (3ull << bits) | v; len = 2 + bits; }
}
// the literal symbol written AT position p: 0 bits for the interior of a k-mer
__device__ __forceinline__ void sle_literal(const u16* __restrict__ minfo, size_t p, size_t n, u8 ch, const u16* __restrict__ rank_byte,
                                            const SleDev& D, u64& code, u32& len) {
    code = 0; len = 0;
    for (u32 j = 0; j + 1 < D.k; ++j) if (p + j < n && (minfo[p + j] & SLE_FIRED)) return;  // inside the k-mer that ends at p+j
    if (D.k > 1 && p + D.k - 1 < n) {
        const u16 v = minfo[p + D.k - 1];
        if (v & SLE_FIRED) { sle_sym(v & SLE_RANK_MASK, D.sb, code, len); return; }
    }
    sle_sym(rank_byte[ch], D.sb, code, len);
}

template <bool PACK>
__global__ __launch_bounds__(256) void sle_stream_kernel(const u8* __restrict__ text, const u32* __restrict__ owner,
                                                          const u32* __restrict__ flen, const u32* __restrict__ fsrc, const u16* __restrict__ minfo,
                                                          size_t n, SleDev D, EncParams P, u64* __restrict__ tile_bits, u64 base_bits,
                                                          u64* __restrict__ out) {
    __shared__ u16 rank_byte[256];
    __shared__ u32 sm[5];
    rank_byte[threadIdx.x] = D.rank_byte[threadIdx.x];
    __syncthreads();
    const size_t p0 = (size_t)blockIdx.x * ENC_TILE + (size_t)threadIdx.x * ENC_PER_THREAD;
    u64 code[ENC_PER_THREAD];       // [0] run header / "0" bit, then literal symbol or src + len
    u32 hdr[ENC_PER_THREAD], hlen[ENC_PER_THREAD], clen[ENC_PER_THREAD], slen[ENC_PER_THREAD], src[ENC_PER_THREAD];
    u32 sum = 0;
    if (p0 < n) {
        u32 prev = (p0 == 0) ? 0u : owner[p0 - 1];
#pragma unroll
        for (int j = 0; j < ENC_PER_THREAD; ++j) {
            const size_t p = p0 + j;
            hdr[j] = 0; hlen[j] = 0; clen[j] = 0; slen[j] = 0; code[j] = 0; src[j] = 0;
            if (p < n) {
                const u32 own = owner[p], fl = flen[p];
                if (own == NONE32) {
                    if (fl) { hdr[j] = (1u << P.dbits) | fl; hlen[j] = 1 + P.dbits; }             // LZSSCoding.hpp:62-68 (dbits <= 31)
                    sle_literal(minfo, p, n, text[p], rank_byte, D, code[j], clen[j]);
                } else if (p == 0 || own != prev) {
                    if (p == 0 || prev != NONE32) hlen[j] = 1;                                     // "0"
                    src[j] = fsrc[p]; slen[j] = P.W;
                    sle_mdr(fl - P.flen_min, P.lbits, code[j], clen[j]);
                }
                prev = own;
                sum += hlen[j] + slen[j] + clen[j];
            }
        }
    }
    if (!PACK) {
        sum = wave_reduce_sum(sum);
        if (lane_id() == 0) sm[wave_id()] = sum;
        __syncthreads();
        if (threadIdx.x == 0) tile_bits[blockIdx.x] = (u64)sm[0] + sm[1] + sm[2] + sm[3];
        return;
    }
    u32 total;
    const u32 excl = block_exclusive_sum<u32, 4>(sum, sm, total);
    if (p0 >= n) return;
    BitSink sink;
    sink.out = out;
    sink.pos = base_bits + tile_bits[blockIdx.x] + excl;
    sink.acc = 0;
    sink.cnt = 0;
#pragma unroll
    for (int j = 0; j < ENC_PER_THREAD; ++j) {
        if (p0 + j >= n) break;
        sink.append(hdr[j], hlen[j]);
        sink.append(src[j], slen[j]);
        sink.append(code[j], clen[j]);
    }
    sink.flush();
}

static size_t encode_sle(Ctx& c, const u8* text, size_t n, FactorSpace fs, u32 k, u8* d_out, size_t out_cap, EncodeStats* st) {
    if (k == 0) k = 3;                                                       // option "kmer", SLECoder.hpp:38
    if (k > 7) throw HipError{hipErrorInvalidValue, "sle: kmer must be in 1..7 (SLECoder.hpp:12)", -1};
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    EncPrelude pre;
    encode_prelude(c, text, n, fs, pre);
    const size_t z = pre.z;

    // ---- alphabet (Encoder ctor :96-160) ---------------------------------------------------------------------
    struct Ent { u64 sym, cnt; };
    std::vector<Ent> alpha;
    for (u32 ch = 0; ch < 256; ++ch) if (pre.hist[ch]) alpha.push_back({ (u64)ch, (u64)pre.hist[ch] });
    size_t sigma = alpha.size();
    u32 sb = bits_for(sigma - 1);
    if (k > 1) {
        const u32 add = (((size_t)1 << sb) == sigma) ? 1u : 2u;
        const size_t eta = ((size_t)1 << (sb + add)) - sigma;
        const size_t kmark = c.arena.mark();
        u32* d_count = c.arena.get<u32>(1);
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(u32), s));
        if (k <= 3) {
            // a 256^k table of counters; the non-zero entries become sort keys (~count, k-mer)
            const size_t tsize = (size_t)1 << (8 * k);
            u32* cnt = c.arena.get<u32>(tsize);
            HIP_TRY(hipMemsetAsync(cnt, 0, tsize * sizeof(u32), s));
            if (n >= k) {
                unsigned g = cdiv(n, 256 * 8); if (g > 8192) g = 8192; if (g == 0) g = 1;
                sle_kmer_count_kernel<<<g, 256, 0, s>>>(text, plain_owner(fs), n, k, cnt);
                LAUNCH_CHECK();
            }
            const size_t cap = std::min(tsize, n) + 64;
            u64* keys[2] = { c.arena.get<u64>(cap), c.arena.get<u64>(cap) };
            u32* vals[2] = { c.arena.get<u32>(cap), c.arena.get<u32>(cap) };
            {
                unsigned g = cdiv(tsize, 256 * 8); if (g > 8192) g = 8192;
                sle_kmer_compact_kernel<<<g, 256, 0, s>>>(cnt, tsize, keys[0], d_count);
                LAUNCH_CHECK();
            }
            const size_t distinct = c.read(d_count);
            const size_t take = std::min(eta, distinct);
            if (take) {
                const int which = sort_pairs_u64_distinct(c, keys, vals, distinct, 0, 56);
                std::vector<u64> top(take);
                c.read_n(keys[which], top.data(), take);
                for (u64 key : top)                                           // :132-136 (the most frequent eta k-mers join the alphabet)
                    alpha.push_back({ (key & 0xFFFFFFull) | (0xFFull << 56), (u64)(0xFFFFFFFFu - (u32)(key >> 24)) });
            }
        } else if (n >= k) {
            // the windows as 8k-bit keys: sort, run-length count, stable sort of the runs by descending count
            u64* keys[2] = { c.arena.get<u64>(n), c.arena.get<u64>(n) };
            u32* vals[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
            {
                unsigned g = cdiv(n, 256 * 8); if (g > 8192) g = 8192; if (g == 0) g = 1;
                sle_kmer_keys_kernel<<<g, 256, 0, s>>>(text, plain_owner(fs), n, k, keys[0], d_count);
                LAUNCH_CHECK();
            }
            const size_t windows = c.read(d_count);
            if (windows) {
                const int w = radix_sort_pairs_u64(c, keys, vals, windows, 0, 8 * (int)k);
                u8* head = c.arena.get<u8>(windows);
                sle_run_heads_kernel<<<cdiv(windows, 256), 256, 0, s>>>(keys[w], windows, head);
                LAUNCH_CHECK();
                u32* start = vals[w];                                         // the values of the first sort carry no information
                select_by_class(c, head, 1, windows, nullptr, start, nullptr, nullptr, d_count);
                const u32 runs = c.read(d_count);
                u64* ckeys[2] = { keys[w ^ 1], c.arena.get<u64>(runs) };
                u32* cvals[2] = { vals[w ^ 1], c.arena.get<u32>(runs) };
                sle_run_counts_kernel<<<cdiv(runs, 256), 256, 0, s>>>(start, runs, (u32)windows, ckeys[0], cvals[0]);
                LAUNCH_CHECK();
                const int v = radix_sort_pairs_u64(c, ckeys, cvals, runs, 0, 32);
                const u32 take = (u32)std::min<size_t>(eta, runs);
                u64* d_km = c.arena.get<u64>(take);
                u32* d_ct = c.arena.get<u32>(take);
                sle_top_kernel<<<cdiv(take, 256), 256, 0, s>>>(ckeys[v], cvals[v], take, keys[w], d_km, d_ct);
                LAUNCH_CHECK();
                std::vector<u64> hk(take); std::vector<u32> hc(take);
                c.read_n(d_km, hk.data(), take);
                c.read_n(d_ct, hc.data(), take);
                for (u32 j = 0; j < take; ++j) alpha.push_back({ hk[j] | (0xFFull << 56), (u64)hc[j] });
            }
        }
        c.arena.release(kmark);
        sigma = alpha.size();
        sb = bits_for(sigma - 1);
    }
    std::sort(alpha.begin(), alpha.end(), [](const Ent& a, const Ent& b) {     // Counter::getSorted :44-57 (a total order)
        return a.cnt != b.cnt ? a.cnt > b.cnt : a.sym < b.sym;
    });
    HostBitWriter hw;
    hw.write_compressed_int(sigma);                                           // :155-158
    for (const Ent& e : alpha) hw.write_compressed_int(e.sym);

    SleDev D;
    memset(&D, 0, sizeof(D));
    D.sb = sb; D.k = k;
    D.f0 = 0; D.f1 = 0;
    for (u32 st8 = 0; st8 < 8; ++st8) {
        D.f0 |= std::min(st8 + 1, k) << (4 * st8);
        D.f1 |= ((st8 + 1 >= k) ? 0u : st8 + 1) << (4 * st8);
    }
    std::vector<std::pair<u64, u16>> km;
    for (size_t r = 0; r < alpha.size(); ++r) {
        if (alpha[r].sym >> 56) km.push_back({ alpha[r].sym & 0x00FFFFFFFFFFFFFFull, (u16)r });
        else D.rank_byte[alpha[r].sym] = (u16)r;
    }
    std::sort(km.begin(), km.end());
    D.nk = (u32)km.size();
    u64 h_kval[SLE_MAX_KMERS]; u16 h_krank[SLE_MAX_KMERS];
    if (D.nk > SLE_MAX_KMERS) throw HipError{hipErrorInvalidValue, "sle: alphabet extension too large", -1};
    for (u32 i = 0; i < D.nk; ++i) { h_kval[i] = km[i].first; h_krank[i] = km[i].second; }
    u64* d_kval = c.arena.get<u64>(SLE_MAX_KMERS);
    u16* d_krank = c.arena.get<u16>(SLE_MAX_KMERS);
    if (D.nk) {
        HIP_TRY(hipMemcpyAsync(d_kval, h_kval, D.nk * sizeof(u64), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_krank, h_krank, D.nk * sizeof(u16), hipMemcpyHostToDevice, s));
    }
    D.kval = d_kval; D.krank = d_krank;

    // ---- k-mer symbols: ranked windows, buffer-fill scan, fired positions ---------------------------------------
    const unsigned tiles = cdiv(n, ENC_TILE);
    u16* minfo = c.arena.get<u16>(n + 8);
    {
        unsigned g = cdiv(n, 256 * 8); if (g > 8192) g = 8192; if (g == 0) g = 1;
        sle_minfo_kernel<<<g, 256, 0, s>>>(text, plain_owner(fs), n, D, minfo);
        LAUNCH_CHECK();
    }
    if (D.nk) {
        u32* tile_fn = c.arena.get<u32>(tiles);
        u32* tile_in = c.arena.get<u32>(tiles);
        sle_tile_fn_kernel<<<tiles, 256, 0, s>>>(minfo, n, D, tile_fn);
        LAUNCH_CHECK();
        sle_tile_scan_kernel<<<1, 1024, 0, s>>>(tile_fn, tiles, tile_in);
        LAUNCH_CHECK();
        sle_fire_kernel<<<tiles, 256, 0, s>>>(minfo, n, D, tile_in);
        LAUNCH_CHECK();
    }

    // ---- fields of LZSSCoding.hpp:47-50, then the token stream -----------------------------------------------------
    EncParams P;
    memset(&P, 0, sizeof(P));
    P.W = bits_for(n);
    P.lbits = bits_for((u64)pre.sc.flen_max - (u64)pre.sc.flen_min);
    P.dbits = bits_for(pre.sc.fdist_max);
    P.flen_min = pre.sc.flen_min;
    hw.write_int(n, 32);
    hw.write_int(pre.sc.flen_min, P.W);
    hw.write_int(pre.sc.flen_max, P.W);
    hw.write_int(pre.sc.fdist_max, P.W);
    const u64 base_bits = hw.nbits;
    u64* tile_bits = c.arena.get<u64>(tiles + 1);
    u64* d_total = c.arena.get<u64>(1);
    sle_stream_kernel<false><<<tiles, 256, 0, s>>>(text, plain_owner(fs), fs.flen, fs.fsrc, minfo, n, D, P, tile_bits, 0, nullptr);
    LAUNCH_CHECK();
    exclusive_sum_u64(c, tile_bits, tile_bits, tiles, d_total);
    const u64 total_bits = base_bits + c.read(d_total);
    const size_t out_len = (size_t)(total_bits >> 3) + ((total_bits & 7) <= 5 ? 1 : 2);
    const size_t padded = align_up(out_len + 8, 8);
    if (padded > out_cap) throw HipError{hipErrorOutOfMemory, "encode: output buffer too small", (int)__LINE__};
    HIP_TRY(hipMemsetAsync(d_out, 0, padded, s));
    HIP_TRY(hipMemcpyAsync(d_out, hw.bytes.data(), hw.bytes.size(), hipMemcpyHostToDevice, s));
    sle_stream_kernel<true><<<tiles, 256, 0, s>>>(text, plain_owner(fs), fs.flen, fs.fsrc, minfo, n, D, P, tile_bits, base_bits, (u64*)d_out);
    LAUNCH_CHECK();
    terminator_kernel<<<1, 64, 0, s>>>(d_out, total_bits);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(s));      // hw.bytes and the host tables must outlive the async copies

    st->factors = z;
    st->flen_min = pre.sc.flen_min; st->flen_max = pre.sc.flen_max; st->fdist_max = pre.sc.fdist_max;
    st->out_bits = total_bits;
    st->sigma = (u32)sigma;
    c.arena.release(mark);
    return out_len;
}

}  // namespace tdc
