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

#include <chrono>
#include <new>
#include <stdlib.h>
#include <sys/mman.h>
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

// one round of in-place pointer jumping: a position follows its chain for up to HOPS references and stops early at a literal
// position (ref == NONE32: final).  *changed != 0 only if some position used up its hops without seeing the literal -- a round in
// which every chain ended needs no further round to confirm it.  (Concurrent updates of ref[] by other threads are benign: every
// value a position can read lies on its own chain, closer to the literal.)
#ifndef TDC_DEC_HOPS
#define TDC_DEC_HOPS 16
#endif
__global__ __launch_bounds__(256) void ref_jump_kernel(u32* ref, size_t n, u32* __restrict__ changed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool any = false;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const u32 q = ref[p];
        if (q == NONE32) continue;
        u32 r = ref[q];
        if (r == NONE32) continue;                              // q is a literal position: final already
        bool open = true;
#pragma unroll
        for (int k = 1; k < TDC_DEC_HOPS; ++k) {
            const u32 r2 = ref[r];
            if (r2 == NONE32) { open = false; break; }
            r = r2;
        }
        ref[p] = r;
        any = any || open;
    }
    if (__any(any) && lane_id() == 0) atomicOr(changed, 1u);
}
// The same with one "final" bit per position (a wave owns 64 consecutive positions = one word of done[], written without atomics):
// positions that are final cost one broadcast word per wave in the later rounds instead of two loads each.
__global__ __launch_bounds__(256) void ref_jump_done_kernel(u32* ref, size_t n, u64* done, u32* __restrict__ changed) {
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = lane_id();
    bool any = false;
    for (size_t b = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 6; b < n; b += nw << 6) {
        const u64 d = done[b >> 6];
        if (d == ~0ull) continue;
        const size_t p = b + lane;
        bool fin = ((d >> lane) & 1ull) != 0 || p >= n;
        if (!fin) {
            const u32 q = ref[p];
            if (q == NONE32) fin = true;
            else {
                u32 r = ref[q];
                if (r == NONE32) fin = true;
                else {
#pragma unroll
                    for (int k = 1; k < TDC_DEC_HOPS; ++k) {
                        const u32 r2 = ref[r];
                        if (r2 == NONE32) { fin = true; break; }
                        r = r2;
                    }
                    ref[p] = r;
                }
            }
        }
        const u64 nd = __ballot(fin);
        if (lane == 0 && nd != d) done[b >> 6] = nd;
        any = any || !fin;
    }
    if (__any(any) && lane_id() == 0) atomicOr(changed, 1u);
}

__global__ void ref_copy_kernel(const u32* __restrict__ ref, size_t a, size_t n, u8* text) {      // positions [a, n)
    const size_t p = a + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 q = ref[p];
    if (q != NONE32) text[p] = text[q];          // q is a literal position: never written by this kernel
}

constexpr unsigned LUT_BITS = 12;

// ============================================================================================================
// Device parse of the Huffman-coded token stream (first half).
//
// The stream has no synchronisation points, but "where does the token that starts at bit x end" is a function of the bits behind x
// alone:   next(x) = x + 1 [+ dbits + the r literal codes] + W + lbits      (LZSSCoding.hpp:57-80 / decode_text_internal :23-76)
// It is evaluated for EVERY bit position of the stream in parallel (a candidate whose run length exceeds fdist_max is no token: the
// field bounds the work of a candidate), the real token starts are the orbit of the first token start under next() -- marked by the
// hierarchical chain marking the lzss_lcp parse and the arithmetic coder's flush points use (prim.hip mark_orbit_u32) --, and the
// tokens are then decoded side by side: count pass (literals + factor length per token), exclusive scan = text positions, emit
// pass (literal bytes, factor list).  Streams above 2^30 bits take several segments, the exit of one is the entry of the next.
// Streams whose longest literal run exceeds DEC_MAX_RUN (poorly compressible inputs: a candidate would decode thousands of
// codes) keep the sequential host parse.
// ============================================================================================================
struct DevTab {                     // Huffman decode tables as the host parser builds them (HuffHeader)
    u16 lut[1 << LUT_BITS];
    u64 firstcode[64];
    u32 prefix_sum[64];
    u32 longest, sigma, have_table, pad;
    u8 numl[64];
    u8 order[256];
};
static_assert(sizeof(DevTab) % 4 == 0, "copied word by word");
struct ParseParams { u64 total, n, flen_min; u32 W, lbits, dbits, fdist_max; };
constexpr u64 DEC_MAX_RUN = 512;            // longest literal run (fdist_max) the device parse takes
constexpr u32 DEC_TILE = 32768;             // bit positions per workgroup of the next() pass
constexpr size_t DEC_SEG = (size_t)1 << 30; // bit positions per segment of the chain marking

// 64 stream bits from absolute bit position x on (MSB first), zeros behind `total` (BitIStream reads zeros at eof).  `words` are the
// stream's 32-bit words, byte-swapped so that bit 31 of word k is stream bit 32 k; word index kb is words[0].
struct BitWin {
    const u32* words; u64 kb; u64 total;
    __device__ __forceinline__ u64 peek(u64 x) const {
        if (x >= total) return 0ull;
        const u64 k = (x >> 5) - kb;
        const u32 sh = (u32)x & 31u;
        const u64 hi = ((u64)words[k] << 32) | words[k + 1];
        u64 w = sh ? (hi << sh) | (u64)(words[k + 2] >> (32 - sh)) : hi;
        if (x + 64 > total) w &= ~0ull << (64 - (total - x));
        return w;
    }
};
// the same over the stream in global memory (bytes; the buffer is padded with 16 zero bytes)
struct BitWinG {
    const u32* s32; u64 total;
    __device__ __forceinline__ u64 peek(u64 x) const {
        if (x >= total) return 0ull;
        const u64 k = x >> 5;
        const u32 sh = (u32)x & 31u;
        const u32 a = __builtin_bswap32(s32[k]), b = __builtin_bswap32(s32[k + 1]), c = __builtin_bswap32(s32[k + 2]);
        const u64 hi = ((u64)a << 32) | b;
        u64 w = sh ? (hi << sh) | (u64)(c >> (32 - sh)) : hi;
        if (x + 64 > total) w &= ~0ull << (64 - (total - x));
        return w;
    }
};

// huffman_decode (HuffmanCoder.hpp:377-397) on a 64-bit window: code length (0: no code of the table) and symbol
__device__ __forceinline__ u32 dec_code(const DevTab* T, u64 w, u32& sym) {
    if (!T->have_table) { sym = (u32)(w >> 56); return 8u; }                     // :606-607
    const u32 e = T->lut[w >> (64 - LUT_BITS)];
    if (e) { sym = e >> 4; return e & 15u; }
    u64 value = 0; u32 length = 0;
    do { value = (value << 1) | ((w >> (63 - length)) & 1u); ++length; } while (length <= T->longest && value < T->firstcode[length - 1]);
    if (length > T->longest) return 0u;
    --length;
    const u64 off = value - T->firstcode[length];
    if (off >= T->numl[length] || T->prefix_sum[length] + off >= T->sigma) return 0u;
    sym = T->order[T->prefix_sum[length] + off];
    return length + 1;
}

// The token that starts at bit x.  Returns 0: literals + factor, 1: literals, then the stream ends (:83-91), < 0: no token.
template <typename Win, typename Lit>
__device__ __forceinline__ int dec_token(const Win& bw, u64 x, const ParseParams& P, const DevTab* T, Lit&& lit, u64& next, u32& r, u32& src, u32& len) {
    r = 0; src = 0; len = 0; next = x;
    if (x >= P.total) return -1;
    u64 w = bw.peek(x);
    u64 y = x + 1;
    if (w >> 63) {
        r = (u32)((w << 1) >> (64 - P.dbits));
        y += P.dbits;
        if (r > P.fdist_max) return -2;
        for (u32 i = 0; i < r; ++i) {
            u32 sym = 0;
            const u32 l = dec_code(T, bw.peek(y), sym);
            if (!l) return -3;
            lit(i, (u8)sym);
            y += l;
        }
        if (y > P.total) return -4;
    }
    if (y >= P.total) { next = y; return 1; }
    w = bw.peek(y);
    src = (u32)(w >> (64 - P.W));
    len = (u32)P.flen_min + (u32)((w << P.W) >> (64 - P.lbits));
    y += P.W + P.lbits;
    if (y > P.total) return -5;
    next = y;
    return 0;
}

__device__ __forceinline__ void dec_tab_to_lds(const DevTab* g, DevTab* l) {
    const u32* a = (const u32*)g; u32* b = (u32*)l;
    for (u32 i = threadIdx.x; i < sizeof(DevTab) / 4; i += blockDim.x) b[i] = a[i];
}

// next() for the bit positions x_in .. x_in + m - 1 (array index = position - x_in; m = "leaves the segment"; no token: m as well --
// the chain ends there and the count pass reports it)
__global__ __launch_bounds__(256) void dec_next_kernel(const u32* __restrict__ s32, u64 x_in, u32 m, ParseParams P, const DevTab* __restrict__ gT,
                                                        u32 nwords, u32* __restrict__ next) {
    extern __shared__ __attribute__((aligned(16))) u32 dyn[];
    DevTab* T = (DevTab*)dyn;
    u32* sw = dyn + sizeof(DevTab) / 4;
    dec_tab_to_lds(gT, T);
    const u32 i0 = blockIdx.x * DEC_TILE;
    const u64 a0 = x_in + i0;
    const u64 kb = a0 >> 5;
    const u64 wmax = (P.total + 31) / 32 + 3;                 // (the buffer is padded: words up to here exist)
    for (u32 k = threadIdx.x; k < nwords; k += 256) sw[k] = (kb + k < wmax) ? __builtin_bswap32(s32[kb + k]) : 0u;
    __syncthreads();
    const BitWin bw{sw, kb, P.total};
    auto nolit = [](u32, u8) {};
    for (u32 i = threadIdx.x; i < DEC_TILE; i += 256) {
        const u32 idx = i0 + i;
        if (idx >= m) break;
        u64 nx; u32 r, src, len;
        const int st = dec_token(bw, x_in + idx, P, T, nolit, nx, r, src, len);
        u32 v = m;
        if (st >= 0 && nx > x_in + idx) { const u64 d = nx - x_in; v = d < (u64)m ? (u32)d : m; }
        next[idx] = v;
    }
}

struct DecScalars { u64 exit_bit; u64 total_out; u32 exit_status; u32 err; };

// the tokens of a segment: absolute bit position, text positions it produces (literals + factor length)
__global__ __launch_bounds__(256) void dec_count_kernel(const u32* __restrict__ s32, u64 x_in, const u32* __restrict__ idx, u32 cnt, ParseParams P,
                                                         const DevTab* __restrict__ gT, u64* __restrict__ tokx, u32* __restrict__ outc, DecScalars* __restrict__ sc) {
    __shared__ DevTab T;
    dec_tab_to_lds(gT, &T);
    __syncthreads();
    const BitWinG bw{s32, P.total};
    auto nolit = [](u32, u8) {};
    u64 acc = 0;
    for (u32 j = blockIdx.x * 256 + threadIdx.x; j < cnt; j += gridDim.x * 256) {
        const u64 x = x_in + idx[j];
        u64 nx; u32 r, src, len;
        const int st = dec_token(bw, x, P, &T, nolit, nx, r, src, len);
        tokx[j] = x;
        // (64-bit: the u32 counts and their u32 scan must not be able to wrap for a crafted stream -- a token that claims more than the
        //  text is an error by itself, and the exact total is checked on the host before anything is written through base[])
        const u64 produced = (st >= 0) ? (u64)r + (u64)len : 0ull;
        outc[j] = produced > P.n ? (u32)P.n + 1u : (u32)produced;
        acc += produced > P.n ? P.n + 1 : produced;
        if (st < 0 || (st == 0 && len == 0) || produced > P.n) atomicOr(&sc->err, 1u);
        if (j == cnt - 1) { sc->exit_bit = nx; sc->exit_status = (u32)(st < 0 ? 2 : st); }
    }
    acc = wave_reduce_sum(acc);                                  // one atomic per wave of the capped grid
    if (lane_id() == 0 && acc) atomicAdd((unsigned long long*)&sc->total_out, (unsigned long long)acc);
}

// literal bytes to their text positions, the factor list (a token without factor gets length 0)
__global__ __launch_bounds__(256) void dec_emit_kernel(const u32* __restrict__ s32, const u64* __restrict__ tokx, const u32* __restrict__ base, u32 z, ParseParams P,
                                                        const DevTab* __restrict__ gT, u8* __restrict__ text, u32* __restrict__ fpos, u32* __restrict__ fsrc,
                                                        u32* __restrict__ flen, DecScalars* __restrict__ sc) {
    __shared__ DevTab T;
    dec_tab_to_lds(gT, &T);
    __syncthreads();
    const BitWinG bw{s32, P.total};
    for (u32 j = blockIdx.x * 256 + threadIdx.x; j < z; j += gridDim.x * 256) {
        const u64 p = base[j];
        u8* dst = text + p;
        const u64 room = p < P.n ? P.n - p : 0;                   // (base[j] <= n is checked on the host through the 64-bit total; clamped all the same)
        u64 nx; u32 r, src, len;
        const int st = dec_token(bw, tokx[j], P, &T, [&](u32 i, u8 b) { if (i < room) dst[i] = b; }, nx, r, src, len);
        bool bad = st < 0 || p + r > P.n;
        if (st == 0) bad = bad || len == 0 || p + r + len > P.n || (u64)src + len > P.n;
        if (bad) { atomicOr(&sc->err, 2u); len = 0; }
        fpos[j] = (u32)(p + r); fsrc[j] = src; flen[j] = (st == 0) ? len : 0u;
    }
}


// ------------------------------------------------------------------------------------------------------------
// Lean marking for streams whose longest possible token is short (la <= DL_LA_MAX bits: literal runs of a few codes -- every
// well-compressible text).  A token chain can only enter a tile of DL_T bit positions within the first la positions behind the
// tile's start, so next() never leaves the workgroup that computes it: per tile only "where does the chain that enters at offset
// o leave" is written (la 16-bit words instead of 2048 x (next, exit1, exit2, mark) = 13 bytes per bit position), a second level
// composes DL_G tiles, one thread walks the groups, and the tokens are then decoded tile by tile from their entries -- counted,
// scanned per tile, and emitted (literals and factor list) by a second walk.  Same results as the general path below
// (dec_next_kernel + mark_orbit_u32 + token list + count / emit per token), which keeps the streams with longer tokens.
// ------------------------------------------------------------------------------------------------------------
constexpr u32 DL_T = 2048;                  // bit positions per tile
constexpr u32 DL_G = 512;                   // tiles per group (2^20 bit positions)
constexpr u32 DL_CH = 16384;                // bit positions per workgroup of the exit pass
constexpr u32 DL_LA_MAX = 1024;             // longest token (bits) the lean path takes
constexpr u16 DL_NONE = 0xFFFFu;

__global__ __launch_bounds__(256) void dec_lean_exit_kernel(const u32* __restrict__ s32, u64 x_in, u32 m, ParseParams P, const DevTab* __restrict__ gT,
                                                             u32 nwords, u32 LA, u16* __restrict__ exit1) {
    extern __shared__ __attribute__((aligned(16))) u32 dyn[];
    DevTab* T = (DevTab*)dyn;
    u32* sw = dyn + sizeof(DevTab) / 4;
    u16* nxl = (u16*)(sw + nwords);                            // next(x) - x of the workgroup's positions; 0: the chain ends at x
    dec_tab_to_lds(gT, T);
    const u32 i0 = blockIdx.x * DL_CH;
    const u64 a0 = x_in + i0;
    const u64 kb = a0 >> 5;
    const u64 wmax = (P.total + 31) / 32 + 3;
    for (u32 k = threadIdx.x; k < nwords; k += 256) sw[k] = (kb + k < wmax) ? __builtin_bswap32(s32[kb + k]) : 0u;
    __syncthreads();
    const BitWin bw{sw, kb, P.total};
    auto nolit = [](u32, u8) {};
    const u64 seg_end = x_in + m;
    for (u32 i = threadIdx.x; i < DL_CH; i += 256) {
        const u32 idx = i0 + i;
        u32 d = 0;
        if (idx < m) {
            u64 nx; u32 r, src, len;
            const int st = dec_token(bw, x_in + idx, P, T, nolit, nx, r, src, len);
            if (st == 0 && nx < seg_end) d = (u32)(nx - (x_in + idx));          // (st == 0: nx > x; at most la <= DL_LA_MAX)
        }
        nxl[i] = (u16)d;
    }
    __syncthreads();
    constexpr u32 TPW = DL_CH / DL_T;
    for (u32 w = threadIdx.x; w < TPW * LA; w += 256) {
        const u32 tt = w / LA, o = w - tt * LA;
        const u32 tile = blockIdx.x * TPW + tt;
        if ((u64)tile * DL_T >= m) continue;
        const u32 tend = (tt + 1) * DL_T;
        u32 e = tt * DL_T + o, res = DL_NONE;
        for (u32 guard = 0; guard <= DL_T; ++guard) {
            if (e >= tend) { res = e - tend; break; }
            const u32 d = nxl[e];
            if (!d) break;
            e += d;
        }
        exit1[(size_t)tile * LA + o] = (u16)res;
    }
}
__global__ void dec_lean_exit2_kernel(const u16* __restrict__ exit1, u32 ntiles, u32 ngroups, u32 LA, u16* __restrict__ exit2) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ngroups * LA) return;
    const u32 g = w / LA, o = w - g * LA;
    const u32 t1 = min((g + 1) * DL_G, ntiles);
    u32 e = o;
    for (u32 t = g * DL_G; t < t1 && e != DL_NONE; ++t) e = exit1[(size_t)t * LA + e];
    exit2[w] = (u16)e;
}
__global__ void dec_lean_groups_kernel(const u16* __restrict__ exit2, u32 ngroups, u32 LA, u16* __restrict__ group_entry) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 e = 0;                                                  // the segment's first token starts at its first bit position
    for (u32 g = 0; g < ngroups; ++g) {
        group_entry[g] = (u16)e;
        if (e != DL_NONE) e = exit2[(size_t)g * LA + e];
    }
}
__global__ void dec_lean_tiles_kernel(const u16* __restrict__ exit1, const u16* __restrict__ group_entry, u32 ntiles, u32 ngroups, u32 LA,
                                      u16* __restrict__ tile_entry) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const u32 t1 = min((g + 1) * DL_G, ntiles);
    u32 e = group_entry[g];
    for (u32 t = g * DL_G; t < t1; ++t) {
        tile_entry[t] = (u16)e;
        if (e != DL_NONE) e = exit1[(size_t)t * LA + e];
    }
}
// the tokens of every tile, walked from the tile's entry: EMIT = false counts them (tokens, text positions they produce), EMIT = true
// writes the literal bytes and the factor list (tok0 / out0: what the earlier segments hold; tbase_*: exclusive sums over the tiles)
template <bool EMIT>
__global__ __launch_bounds__(256) void dec_lean_walk_kernel(const u32* __restrict__ s32, u64 x_in, u32 m, ParseParams P, const DevTab* __restrict__ gT,
                                                             const u16* __restrict__ tile_entry, u32 ntiles, u32* __restrict__ tcount, u32* __restrict__ tout,
                                                             u64 tok0, u64 out0, u8* __restrict__ text, u32* __restrict__ fpos, u32* __restrict__ fsrc,
                                                             u32* __restrict__ flen, DecScalars* __restrict__ sc) {
    __shared__ DevTab T;
    dec_tab_to_lds(gT, &T);
    __syncthreads();
    const BitWinG bw{s32, P.total};
    const u64 seg_end = x_in + m;
    u64 acc = 0;
    for (u32 t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) {
        const u32 e = tile_entry[t];
        u32 cnt = 0;
        u64 out = 0;
        if (e != DL_NONE) {
            u64 x = x_in + (u64)t * DL_T + e;
            const u64 tile_end = min(x_in + (u64)(t + 1) * DL_T, seg_end);
            u64 j = EMIT ? tok0 + tcount[t] : 0;
            u64 p = EMIT ? out0 + tout[t] : 0;
            for (u32 guard = 0; x < tile_end && guard <= DL_T; ++guard) {
                u64 nx; u32 r, src, len;
                int st;
                if (EMIT) {
                    u8* dst = text + (p < P.n ? p : P.n);
                    const u64 room = p < P.n ? P.n - p : 0;
                    st = dec_token(bw, x, P, &T, [&](u32 i, u8 b) { if (i < room) dst[i] = b; }, nx, r, src, len);
                    bool bad = st < 0 || p + r > P.n;
                    if (st == 0) bad = bad || len == 0 || p + r + len > P.n || (u64)src + len > P.n;
                    if (bad) { atomicOr(&sc->err, 2u); len = 0; }
                    fpos[j] = (u32)(p + r); fsrc[j] = src; flen[j] = (st == 0) ? len : 0u;
                    ++j;
                    p += (u64)r + len;
                } else {
                    auto nolit = [](u32, u8) {};
                    st = dec_token(bw, x, P, &T, nolit, nx, r, src, len);
                    const u64 produced = (st >= 0) ? (u64)r + (u64)len : 0ull;
                    out += produced > P.n ? P.n + 1 : produced;
                    if (st < 0 || (st == 0 && len == 0) || produced > P.n) atomicOr(&sc->err, 1u);
                }
                ++cnt;
                if (st != 0 || nx >= seg_end || nx <= x) {         // the segment's last token
                    if (!EMIT) { sc->exit_bit = nx; sc->exit_status = (u32)(st < 0 ? 2 : st); }
                    break;
                }
                x = nx;
            }
        }
        if (!EMIT) { tcount[t] = cnt; tout[t] = out > P.n ? (u32)P.n + 1u : (u32)out; acc += out; }
    }
    if (!EMIT) {
        acc = wave_reduce_sum(acc);
        if (lane_id() == 0 && acc) atomicAdd((unsigned long long*)&sc->total_out, (unsigned long long)acc);
    }
}


}  // namespace

// Header of a lcpcomp(coder=huff) stream: HuffmanCoder::Decoder ctor (HuffmanCoder.hpp:581-597) + huffmantable_decode (:278-290),
// then the four fields of decode_text_internal (LCPCompressor.hpp:23-76).  Shared by the host parse and the device parse.
struct HuffHeader {
    bool have_table = false;
    u8 order[256];
    u64 firstcode[64];
    size_t prefix_sum[64];
    unsigned longest = 0;
    u8 numl[64] = {0};
    size_t sigma = 0;
    std::vector<unsigned short> lut;                          // (symbol << 4) | code length, 0 = longer than LUT_BITS / invalid
    u64 n = 0, flen_min = 0, flen_max = 0, fdist_max = 0;
    unsigned W = 0, lbits = 0, dbits = 0;
};
static void parse_huff_header(FastBits& bs, size_t len, HuffHeader& H) {
    H.have_table = bs.read(1) != 0;
    if (H.have_table) {
        H.longest = (unsigned)(bs.read_compressed_int() & 0xFF);
        if (H.longest == 0 || H.longest > 57) throw StreamError{"corrupt Huffman table"};
        for (unsigned i = 0; i < H.longest; ++i) H.numl[i] = (u8)bs.read_compressed_int();
        H.sigma = (size_t)bs.read_compressed_int();
        if (H.sigma > 256) throw StreamError{"corrupt Huffman table"};
        for (size_t i = 0; i < H.sigma; ++i) H.order[i] = (u8)bs.read(8);
        H.firstcode[H.longest - 1] = 0;                                              // gen_first_codes :192-198
        for (unsigned i = H.longest - 1; i > 0; --i) H.firstcode[i - 1] = (H.firstcode[i] + H.numl[i]) / 2;
        size_t acc = 0;                                                               // gen_prefix_sum_lengths :350-370
        for (unsigned l = 0; l < H.longest; ++l) { H.prefix_sum[l] = acc; acc += H.numl[l]; }
        if (acc > H.sigma) throw StreamError{"corrupt Huffman table"};
        H.lut.assign((size_t)1 << LUT_BITS, 0);
        for (unsigned l = 1; l <= H.longest && l <= LUT_BITS; ++l)
            for (unsigned k = 0; k < H.numl[l - 1]; ++k) {
                const u64 code = H.firstcode[l - 1] + k;
                if (code >> l) throw StreamError{"corrupt Huffman table"};
                const unsigned short e = (unsigned short)((H.order[H.prefix_sum[l - 1] + k] << 4) | l);
                const size_t base = (size_t)code << (LUT_BITS - l);
                for (size_t x = 0; x < ((size_t)1 << (LUT_BITS - l)); ++x) H.lut[base + x] = e;
            }
    }
    // decode_text_internal (LCPCompressor.hpp:23-76)
    H.n = bs.read(32);
    H.W = bits_for(H.n);
    H.flen_min = bs.read(H.W); H.flen_max = bs.read(H.W); H.fdist_max = bs.read(H.W);
    H.lbits = bits_for(H.flen_max - H.flen_min); H.dbits = bits_for(H.fdist_max);
    if (H.n == 0 || H.n >= 0x7FFFFFFFull) throw StreamError{"text length out of range"};     // a text always holds its sentinel
    {   // plausibility (a corrupt header would otherwise ask for gigabytes): a literal costs at least one bit, a factor at least W
        // bits and covers at most flen_max positions
        const u64 bits = (u64)len * 8;
        if (H.n > bits + (bits / H.W + 1) * (H.flen_max ? H.flen_max : 1)) throw StreamError{"text length out of range"};
    }
}

// Host parse: literals go straight to their text positions in `text` (n bytes), factors into the three vectors.
// Returns n.  Throws StreamError for malformed input.
static u64 parse_lzss_huff_stream(FastBits& bs, const HuffHeader& H, std::vector<u8>& text, std::vector<u32>& fpos, std::vector<u32>& fsrc,
                                  std::vector<u32>& flen) {
    const bool have_table = H.have_table;
    const unsigned longest = H.longest, W = H.W, lbits = H.lbits, dbits = H.dbits;
    const u64 n = H.n, flen_min = H.flen_min;
    const u64* firstcode = H.firstcode; const size_t* prefix_sum = H.prefix_sum; const u8* numl = H.numl; const u8* order = H.order;
    const size_t sigma = H.sigma;
    const std::vector<unsigned short>& lut = H.lut;
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

// page-locked host memory (hipHostMalloc / hipHostRegister)?  Asynchronous copies overlap with kernels only for such buffers; a copy
// from or to pageable memory is staged by the runtime and holds the calling thread.
static bool host_pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// where the decoded text goes: the caller's buffer, or a buffer allocated here (2 MiB aligned, transparent huge pages asked for: a
// 256 MiB text otherwise pays 65 536 page faults in front of the download) that the caller releases with free()
static u8* decode_dest(DecodeOut& o, size_t n) {
    if (o.into) {
        if (n > o.cap) throw HipError{hipErrorOutOfMemory, "decompress: output buffer too small", (int)__LINE__};
        return o.into;
    }
    void* p = nullptr;
    const size_t bytes = n ? n : 1;
    if (bytes >= ((size_t)4 << 20)) {
        if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) p = nullptr;
        else (void)madvise(p, bytes, MADV_HUGEPAGE);
    } else p = malloc(bytes);
    if (!p) throw std::bad_alloc();
    o.owned = (u8*)p;
    return o.owned;
}

// Resolves the reference forest of n text positions on the device (d_text holds the literals at their positions, the factor list is
// on the device as well) and downloads the text.
static void resolve_and_download(Ctx& c, size_t n, u8* d_text, u32* d_ref, const u32* d_pos, const u32* d_src, const u32* d_len, size_t z,
                                 u32* d_changed, DecodeOut& out, DecodeStats* st) {
    hipStream_t s = c.stream;
    const bool dlog = c.dec_log != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!dlog) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "decode:   %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    fill_u32(c, d_ref, n, NONE32);
    const int G = (z * 64 > n) ? 8 : 64;
    ref_scatter_kernel<<<cdiv(z * G, 256), 256, 0, s>>>(d_pos, d_src, d_len, z, G, d_ref);
    LAUNCH_CHECK();
    tick("fill + reference scatter");
    unsigned g = cdiv(n, 256 * 8); if (g > 16384) g = 16384;
    const bool use_done = c.dec_done != 0;      // (A/B switch: option dec_done)
    u64* d_done = nullptr;
    if (use_done) {
        d_done = c.arena.get<u64>(n / 64 + 1);
        HIP_TRY(hipMemsetAsync(d_done, 0, (n / 64 + 1) * sizeof(u64), s));
    }
    for (u32 round = 0;; ++round) {
        if (round > 40) throw StreamFormatError{"corrupt stream: reference cycle"};     // depth < 2^31
        HIP_TRY(hipMemsetAsync(d_changed, 0, sizeof(u32), s));
        if (d_done) ref_jump_done_kernel<<<g, 256, 0, s>>>(d_ref, n, d_done, d_changed);
        else ref_jump_kernel<<<g, 256, 0, s>>>(d_ref, n, d_changed);
        LAUNCH_CHECK();
        st->rounds = round + 1;
        if (c.read(d_changed) == 0) break;
    }
    tick("pointer jumping");
    u8* dst = decode_dest(out, n);
    tick("host buffer");
    constexpr size_t CH = (size_t)64 << 20;
    if (c.copy_stream && n >= 2 * CH && !dlog && host_pinned(dst)) {
        // the copy pass in chunks, every chunk downloaded on the second stream while the next one is copied (a chunk reads literal
        // positions only, wherever they are: the chunks do not depend on one another)
        struct Drain { Ctx& c; ~Drain() { (void)hipStreamSynchronize(c.copy_stream); } } drain{c};   // nothing of this call stays behind on the second stream
        u32 k = 0;
        for (size_t a = 0; a < n; a += CH, ++k) {
            const size_t b = std::min(n, a + CH);
            ref_copy_kernel<<<cdiv(b - a, 256), 256, 0, s>>>(d_ref, a, b, d_text);
            LAUNCH_CHECK();
            hipEvent_t ev = c.ev_copy[k & 31];
            HIP_TRY(hipEventRecord(ev, s));
            HIP_TRY(hipStreamWaitEvent(c.copy_stream, ev, 0));
            HIP_TRY(hipMemcpyAsync(dst + a, d_text + a, b - a, hipMemcpyDeviceToHost, c.copy_stream));
        }
        HIP_TRY(hipStreamSynchronize(c.copy_stream));
        HIP_TRY(hipStreamSynchronize(s));
        return;
    }
    ref_copy_kernel<<<cdiv(n, 256), 256, 0, s>>>(d_ref, 0, n, d_text);
    LAUNCH_CHECK();
    tick("copy pass");
    HIP_TRY(hipMemcpyAsync(dst, d_text, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tick("download");
}

// lcpcomp(coder=huff) with the token stream parsed on the device.  Returns false if this stream keeps the host parse (long literal
// runs); throws StreamFormatError for malformed input.
static bool decode_lzss_huff_device(Ctx& c, const u8* stream, size_t len, const HuffHeader& H, u64 x0, u64 total, DecodeOut& out,
                                    DecodeStats* st) {
    if (H.fdist_max > DEC_MAX_RUN) return false;
    const u64 code_max = H.have_table ? H.longest : 8;
    const u64 la_bits = 1 + H.dbits + H.fdist_max * code_max + H.W + H.lbits;          // the longest token a candidate may read
    const u32 nwords = (u32)((DEC_TILE + la_bits + 31) / 32 + 4);
    const size_t lds = sizeof(DevTab) + (size_t)nwords * 4;
    if (lds > 60 * 1024) return false;
    const size_t n = (size_t)H.n;
    const u64 min_tok = 1 + H.W + H.lbits;
    const size_t zmax = (size_t)((total - x0) / min_tok + 2);
    const u64 seg_bits = c.dec_seg ? (u64)c.dec_seg : (u64)DEC_SEG;                    // (tests shrink the segments)
    const size_t seg = (size_t)std::min<u64>(seg_bits, total - x0 + 1);
    hipStream_t s = c.stream;
    const bool dlog = c.dec_log != 0;                                                    // stage times on stderr (synchronises)
    auto t_last = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!dlog) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "decode: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (out.into && n > out.cap) throw HipError{hipErrorOutOfMemory, "decode: the caller's buffer is too small for the text", (int)__LINE__};   // (known from the header: before any device work)
    // (the device parse needs ~17 bytes of arena per stream bit of a segment on top of 5 n + 28 zmax; the host parse needs 5 n + 12 z: if
    //  the device cannot provide the former, the host parse takes the stream instead of the call failing)
    try { c.ensure_arena(len + 64 + n * 5 + n / 8 + zmax * 28 + seg * 17 + sizeof(DevTab) + ((size_t)16 << 20)); }
    catch (const HipError& e) { if (e.e != hipErrorOutOfMemory) throw; (void)hipGetLastError(); return false; }
    const size_t mark0 = c.arena.mark();
    u8* d_stream = c.arena.get<u8>(len + 64);
    // long streams go up in chunks on the second stream; a segment waits for the chunks its bit positions (and the longest token behind
    // them) lie in, so the parse of segment k runs while segment k + 1 is still on its way
    struct Upload {
        Ctx& c; size_t chunk = 0, nchunks = 0, waited = 0;
        explicit Upload(Ctx& cc) : c(cc) {}
        ~Upload() { if (nchunks) (void)hipStreamSynchronize(c.copy_stream); }    // (the caller's buffer is not read after the call returns)
        void need(size_t bytes, size_t len, hipStream_t s) {
            if (!nchunks) return;
            const size_t upto = std::min(nchunks, (std::min(bytes, len) + chunk - 1) / chunk);
            for (; waited < upto; ++waited) HIP_TRY(hipStreamWaitEvent(s, c.ev_copy[waited], 0));
        }
    } up(c);
    if (c.copy_stream && len >= ((size_t)64 << 20) && host_pinned(stream)) {
        up.chunk = std::max((size_t)32 << 20, (len + 31) / 32);
        up.chunk = (up.chunk + 4095) & ~(size_t)4095;
        up.nchunks = (len + up.chunk - 1) / up.chunk;
        // (decompression has ev_copy[0 .. 31] to itself -- the compress path's slots 0, 8, 9, 16 .. 39 belong to calls that never overlap
        //  this one, every path drains the copy stream before it returns --; the chunk size above keeps the count within them)
        if (up.nchunks > 32) throw HipError{hipErrorUnknown, "decode: more upload chunks than copy events", (int)__LINE__};
        for (size_t k = 0; k < up.nchunks; ++k) {
            const size_t a = k * up.chunk, b = std::min(len, a + up.chunk);
            HIP_TRY(hipMemcpyAsync(d_stream + a, stream + a, b - a, hipMemcpyHostToDevice, c.copy_stream));
            HIP_TRY(hipEventRecord(c.ev_copy[k], c.copy_stream));
        }
        (void)hipStreamQuery(c.copy_stream);                                                 // (submit now)
    } else {
        HIP_TRY(hipMemcpyAsync(d_stream, stream, len, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemsetAsync(d_stream + len, 0, 64, s));
    const u32* s32 = (const u32*)d_stream;                                              // (arena allocations are 256-byte aligned)
    DevTab* d_tab = (DevTab*)c.arena.alloc(sizeof(DevTab));
    {
        std::vector<u8> hb(sizeof(DevTab), 0);
        DevTab* t = (DevTab*)hb.data();
        t->have_table = H.have_table ? 1u : 0u; t->longest = H.longest; t->sigma = (u32)H.sigma;
        if (H.have_table) {
            memcpy(t->lut, H.lut.data(), sizeof(t->lut));
            for (unsigned i = 0; i < 64; ++i) { t->firstcode[i] = i < H.longest ? H.firstcode[i] : 0; t->prefix_sum[i] = i < H.longest ? (u32)H.prefix_sum[i] : 0; t->numl[i] = H.numl[i]; }
            memcpy(t->order, H.order, 256);
        }
        HIP_TRY(hipMemcpyAsync(d_tab, hb.data(), sizeof(DevTab), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                                               // (hb leaves scope)
    }
    const ParseParams P{ total, H.n, H.flen_min, H.W, H.lbits, H.dbits, (u32)H.fdist_max };
    if (la_bits <= DL_LA_MAX && c.dec_lean) {
        // ---- lean marking (short tokens): per segment exits per tile entry -> groups -> tile entries -> count walk -> scans -> emit walk
        const u32 LA = (u32)la_bits;
        u32* d_pos = c.arena.get<u32>(zmax), *d_src = c.arena.get<u32>(zmax), *d_len = c.arena.get<u32>(zmax);
        DecScalars* d_sc = (DecScalars*)c.arena.alloc(sizeof(DecScalars));
        u32* d_cnt = c.arena.get<u32>(2);
        u8* d_text = c.arena.get<u8>(n + 64);
        u32* d_ref = c.arena.get<u32>(n);
        HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(DecScalars), s));
        HIP_TRY(hipMemsetAsync(d_text, 0, n, s));
        const u32 nw_l = (u32)((DL_CH + la_bits + 31) / 32 + 4);
        const size_t lds_l = sizeof(DevTab) + (size_t)nw_l * 4 + (size_t)DL_CH * 2;
        size_t z = 0;
        u64 x_in = x0, out0 = 0;
        u32 last_status = 0;
        while (x_in < total) {
            const u32 m = (u32)std::min<u64>(seg_bits, total - x_in);
            const u32 ntiles = cdiv(m, DL_T), ngroups = cdiv(ntiles, DL_G);
            const size_t mk = c.arena.mark();
            u16* exit1 = c.arena.get<u16>((size_t)ntiles * LA);
            u16* exit2 = c.arena.get<u16>((size_t)ngroups * LA);
            u16* gentry = c.arena.get<u16>(ngroups);
            u16* tentry = c.arena.get<u16>(ntiles);
            u32* tcount = c.arena.get<u32>(ntiles), *tout = c.arena.get<u32>(ntiles);
            up.need((size_t)((x_in + m + la_bits) / 8 + 64), len, s);
            tick("upload + tables");
            dec_lean_exit_kernel<<<cdiv(m, DL_CH), 256, lds_l, s>>>(s32, x_in, m, P, d_tab, nw_l, LA, exit1);
            LAUNCH_CHECK();
            tick("next() + tile exits");
            dec_lean_exit2_kernel<<<cdiv((size_t)ngroups * LA, 256), 256, 0, s>>>(exit1, ntiles, ngroups, LA, exit2);
            LAUNCH_CHECK();
            dec_lean_groups_kernel<<<1, 64, 0, s>>>(exit2, ngroups, LA, gentry);
            LAUNCH_CHECK();
            dec_lean_tiles_kernel<<<cdiv(ngroups, 64), 64, 0, s>>>(exit1, gentry, ntiles, ngroups, LA, tentry);
            LAUNCH_CHECK();
            tick("group + tile entries");
            const unsigned gw = std::min<u32>(cdiv(ntiles, 256), 4096u);
            dec_lean_walk_kernel<false><<<gw, 256, 0, s>>>(s32, x_in, m, P, d_tab, tentry, ntiles, tcount, tout, 0, 0, nullptr, nullptr, nullptr, nullptr, d_sc);
            LAUNCH_CHECK();
            exclusive_sum_u32(c, tcount, tcount, ntiles, d_cnt);
            const u32 cnt = c.read(d_cnt);
            exclusive_sum_u32(c, tout, tout, ntiles, d_cnt);
            const DecScalars h = c.read(d_sc);
            tick("count walk + scans");
            if (cnt == 0 || z + cnt > zmax) throw StreamFormatError{"corrupt stream: token chain"};
            if (h.err || h.exit_status == 2) throw StreamFormatError{"corrupt stream: malformed token"};
            if (h.total_out > H.n) throw StreamFormatError{"corrupt stream: length mismatch"};   // (64-bit total over all segments so far: no 32-bit sum below it can have wrapped)
            dec_lean_walk_kernel<true><<<gw, 256, 0, s>>>(s32, x_in, m, P, d_tab, tentry, ntiles, tcount, tout, (u64)z, out0, d_text, d_pos, d_src, d_len, d_sc);
            LAUNCH_CHECK();
            tick("emit walk");
            c.arena.release(mk);
            z += cnt;
            out0 = h.total_out;
            last_status = h.exit_status;
            if (h.exit_status == 1 || h.exit_bit >= total) break;
            if (h.exit_bit <= x_in) throw StreamFormatError{"corrupt stream: token chain"};
            x_in = h.exit_bit;
        }
        if (out0 != H.n) throw StreamFormatError{"corrupt stream: length mismatch"};
        const DecScalars h = c.read(d_sc);
        if (h.err) throw StreamFormatError{"corrupt stream: factor out of range"};
        st->factors = z ? z - 1 : 0;
        if (z && last_status == 0) st->factors = z;
        resolve_and_download(c, n, d_text, d_ref, d_pos, d_src, d_len, z, d_cnt, out, st);
        tick("references + download");
        c.arena.release(mark0);
        return true;
    }
    u64* tokx = c.arena.get<u64>(zmax);
    u32* outc = c.arena.get<u32>(zmax + 1);
    u32* d_pos = c.arena.get<u32>(zmax), *d_src = c.arena.get<u32>(zmax), *d_len = c.arena.get<u32>(zmax);
    DecScalars* d_sc = (DecScalars*)c.arena.alloc(sizeof(DecScalars));
    u32* d_cnt = c.arena.get<u32>(2);
    HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(DecScalars), s));
    size_t z = 0;
    u64 x_in = x0;
    while (x_in < total) {
        const u32 m = (u32)std::min<u64>(seg_bits, total - x_in);
        const size_t mk = c.arena.mark();
        u32* next = c.arena.get<u32>(m);
        u32* e1 = c.arena.get<u32>(m), *e2 = c.arena.get<u32>(m);
        u8* mark = c.arena.get<u8>(m);
        up.need((size_t)((x_in + m + la_bits) / 8 + 64), len, s);
        tick("upload + tables");
        dec_next_kernel<<<cdiv(m, DEC_TILE), 256, lds, s>>>(s32, x_in, m, P, d_tab, nwords, next);
        LAUNCH_CHECK();
        tick("next() of every bit");
        mark_orbit_u32(c, next, m, mark, e1, e2);
        tick("chain marking");
        u32* idx = e1;                                                                  // (the exit arrays are free again)
        select_by_class(c, mark, 1, m, nullptr, idx, nullptr, nullptr, d_cnt);
        const u32 cnt = c.read(d_cnt);
        tick("token list");
        if (cnt == 0 || z + cnt > zmax) throw StreamFormatError{"corrupt stream: token chain"};
        dec_count_kernel<<<std::min<u32>(cdiv(cnt, 256), 4096u), 256, 0, s>>>(s32, x_in, idx, cnt, P, d_tab, tokx + z, outc + z, d_sc);
        LAUNCH_CHECK();
        const DecScalars h = c.read(d_sc);
        tick("count pass");
        c.arena.release(mk);
        z += cnt;
        if (h.err || h.exit_status == 2) throw StreamFormatError{"corrupt stream: malformed token"};
        if (h.exit_status == 1 || h.exit_bit >= total) break;
        if (h.exit_bit <= x_in) throw StreamFormatError{"corrupt stream: token chain"};
        x_in = h.exit_bit;
    }
    // text positions of the tokens
    u32* base = outc;                                                                   // in place
    exclusive_sum_u32(c, outc, base, z, d_cnt);
    const u32 produced = c.read(d_cnt);
    {   // the exact 64-bit total of the count passes: equal to n means that no partial sum of the 32-bit scan can have wrapped
        const DecScalars ht = c.read(d_sc);
        if (ht.total_out != H.n) throw StreamFormatError{"corrupt stream: length mismatch"};
    }
    if ((u64)produced != H.n) throw StreamFormatError{"corrupt stream: length mismatch"};
    u8* d_text = c.arena.get<u8>(n + 64);
    u32* d_ref = c.arena.get<u32>(n);
    HIP_TRY(hipMemsetAsync(d_text, 0, n, s));
    dec_emit_kernel<<<std::min<u32>(cdiv(z, 256), 4096u), 256, 0, s>>>(s32, tokx, base, (u32)z, P, d_tab, d_text, d_pos, d_src, d_len, d_sc);
    LAUNCH_CHECK();
    const DecScalars h = c.read(d_sc);
    tick("scan + emit pass");
    if (h.err) throw StreamFormatError{"corrupt stream: factor out of range"};
    st->factors = z ? z - 1 : 0;                                                        // (every token but the last carries a factor)
    if (z && h.exit_status == 0) st->factors = z;
    resolve_and_download(c, n, d_text, d_ref, d_pos, d_src, d_len, z, d_cnt, out, st);
    tick("references + download");
    c.arena.release(mark0);
    return true;
}

// coder: 0 = HuffmanCoder, 2 = ASCIICoder, 3 | kmer << 8 = SLECoder (the coder ids of encode_stream)
size_t decode_lzss(Ctx& c, const u8* stream, size_t len, int coder, DecodeOut& out, DecodeStats* st) {
    DecodeStats local;
    if (!st) st = &local;
    *st = DecodeStats();
    std::vector<u8> text;                                    // (host parse only: the literals at their text positions)
    std::vector<u32> fpos, fsrc, flen;
    u64 n;
    if ((coder & 0xFF) == 3) n = parse_lzss_sle_stream(stream, len, (unsigned)(coder >> 8) ? (unsigned)(coder >> 8) : 3u, text, fpos, fsrc, flen);
    else if (coder == 2) n = parse_lzss_ascii_stream(stream, len, text, fpos, fsrc, flen);
    else {
        FastBits bs(stream, len);
        HuffHeader H;
        parse_huff_header(bs, len, H);
        // the token stream itself: on the device (streams of 1 MiB and more; TDC_GPU_DEC_PARSE = 0 never / 2 always: tests), else on the host
        if (bs.pos < bs.total && c.dec_parse && (c.dec_parse >= 2 || len >= ((size_t)1 << 20)) &&
            decode_lzss_huff_device(c, stream, len, H, bs.pos, bs.total, out, st)) {
            st->device_parse = 1;
            return (size_t)H.n;
        }
        n = parse_lzss_huff_stream(bs, H, text, fpos, fsrc, flen);
    }
    const size_t z = fpos.size();
    st->factors = z;
    if (n == 0 || z == 0) {                                   // nothing to resolve
        u8* dst = decode_dest(out, (size_t)n);
        if (n) memcpy(dst, text.data(), (size_t)n);
        return (size_t)n;
    }
    hipStream_t s = c.stream;
    c.ensure_arena((size_t)n * 5 + (size_t)n / 8 + z * 12 + ((size_t)16 << 20));
    const size_t mark = c.arena.mark();
    u8* d_text = c.arena.get<u8>((size_t)n);
    u32* d_ref = c.arena.get<u32>((size_t)n);
    u32* d_pos = c.arena.get<u32>(z), *d_src = c.arena.get<u32>(z), *d_len = c.arena.get<u32>(z);
    u32* d_changed = c.arena.get<u32>(1);
    HIP_TRY(hipMemcpyAsync(d_text, text.data(), (size_t)n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_pos, fpos.data(), z * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_src, fsrc.data(), z * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_len, flen.data(), z * 4, hipMemcpyHostToDevice, s));
    resolve_and_download(c, (size_t)n, d_text, d_ref, d_pos, d_src, d_len, z, d_changed, out, st);
    c.arena.release(mark);
    return (size_t)n;
}

}  // namespace tdc
