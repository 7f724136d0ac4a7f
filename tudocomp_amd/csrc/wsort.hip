// wsort.hip -- wide-key splitter sort of all suffixes of a text (gfx950, wave64): the suffix array's initial order.
//
// Replaces, for large texts, the 64-bit initial sort + refinement pass + first doubling round of suffix_array.hip
// (ds/SADivSufSort.hpp:27-51 semantics: the suffix array is unique, any correct construction is bit-compatible).
//
// Keys are BIT-PACKED: the recoded symbols text[p .. p+s) as s fields of b bits, left-aligned in KW = 1 or 2 64-bit words.  Two keys
// that differ share floor(clz(k ^ k') / b) symbols -- the LCP value of the two suffixes, for free, wherever the sort separates them.
// With two words an English-like text (29 symbols, b = 5) is sorted by 25 symbols in one go: ~5 % of the suffixes of a 2 GB text
// still tie afterwards, instead of two thirds after 13 symbols.
//
// Structure = ssort.hip (sampled splitters, 2-3 partition levels on key RANKS, in-LDS leaf sort), with these differences:
//   * records are (k1, k2, position): 20 bytes; level 1 computes its keys from the text (no key array is read); the scatter pass
//     stages one stream at a time through LDS, so the record width costs no registers;
//   * splitter comparisons are 128-bit, so a heavy 64-bit prefix (a frequent word) is split by its second word across many leaves;
//   * the leaf kernel sorts a unit (<= 8192 records) by k1 with the composite-word LSD passes of ssort.hip, then orders the runs
//     of equal k1 by k2: runs of <= 256 by counting in LDS, longer ones by LSD passes on (run head, 16-bit chunk of k2) -- and
//     writes, instead of the sorted keys, what the suffix array needs: the positions, one head flag per slot (slot starts a group of
//     equal keys) and the LCP (in symbols) of every head with its predecessor.  Unit-border LCPs come from a tiny fix-up kernel
//     that re-reads the two suffixes from the text.
// The sort is not stable (equal keys form one group of the later rounds).
#include "prim.hpp"

#include <vector>
#include <algorithm>
#include <stdlib.h>

namespace tdc {

constexpr int WS_TILE = 4096;        // records per partition tile: 256 threads x 16
constexpr int WS_ITEMS = 16;
constexpr int WS_HALO = 64;          // symbols staged beyond a tile (s <= 64)
#ifndef TDC_WS_PEEL
#define TDC_WS_PEEL 0
#endif
#ifndef TDC_WS_CMAX
#define TDC_WS_CMAX 64
#endif
constexpr u32 WS_CMAX = TDC_WS_CMAX; // runs of equal k1 up to this length are ordered by counting (a member costs its run's length)
constexpr u32 WS_WAVE_MAX = 1024;    // longer runs up to this length: one WAVE sorts the run (ws_run_wave_kernel); beyond: kernel A again

struct WSLevel {
    const u64* k1_in; const u64* k2_in; const u32* v_in;
    u64* k1_out; u64* k2_out; u32* v_out;
    u32* counts;                 // [rows][D]
    const u32* blk_seg; const u32* blk_start; const u32* seg_start;
    const u32* seg_end;          // end of every segment (seg_start + 1 where the segments are contiguous)
    u32 sub;                     // pieces per splitter segment (1; the number of chunks on the level that merges a chunk-wise first level)
    const u64* sp1; const u64* sp2;      // [NS + 1] splitters (high / low word), sp[NS] = ~0
    u16* digits;
    u32 nseg, F, stride, R, D, per_xcd;
    u32 pre_counts = 0;          // ... and their tile histograms have been copied into `counts` already (ws_pre_hist_kernel): the count pass skips their rows
    u32 pre_digits = 0;          // count pass of the merging level: the pieces of the first pre_digits chunks have their digits in `digits` already (computed chunk by chunk behind the upload): read them, search nothing
    size_t gen_off, gen_len;     // GEN level: the tiles cover text positions [gen_off, gen_off + gen_len)
};

__device__ __forceinline__ bool ws_row(const WSLevel& P, u32 row, u32& s, size_t& base, u32& cnt) {
    const u32 blk = row / P.R;
    if (blk >= P.blk_start[P.nseg]) return false;
    s = P.blk_seg[blk];
    const u64 t = (u64)(blk - P.blk_start[s]) * P.R + row % P.R;
    const u32 s0 = P.seg_start[s], s1 = P.seg_end[s];
    const u64 off = t * WS_TILE;
    base = s0; cnt = 0;
    if (off >= (u64)(s1 - s0)) return true;
    base = (size_t)s0 + off;
    const u64 left = (u64)(s1 - s0) - off;
    cnt = left < WS_TILE ? (u32)left : (u32)WS_TILE;
    return true;
}

template <int KW, bool LAST, int FMAX>
__device__ __forceinline__ void ws_load_splitters(const WSLevel& P, u32 s_piece, u64* spl1, u64* spl2) {
    const u32 s = s_piece / P.sub;
    for (u32 i = threadIdx.x; i < (u32)FMAX; i += blockDim.x) {
        u64 a = ~0ull, b = ~0ull;
        size_t idx = 0; bool have = false;
        if (LAST) { if (i < P.F) { idx = (size_t)s * P.F + i; have = true; } }
        else if (i + 1 < P.F) { idx = ((size_t)s * P.F + i + 1) * P.stride - 1; have = true; }
        if (have) { a = P.sp1[idx]; if (KW == 2) b = P.sp2[idx]; }
        spl1[i] = a;
        if (KW == 2) spl2[i] = b;
    }
}

// #splitters < x among spl[0 .. FMAX - 1) (branch-free binary search, lexicographic on the two words), then the digit
template <int KW, bool LAST, int FMAX>
__device__ __forceinline__ u32 ws_digit(const u64* spl1, const u64* spl2, u64 x1, u64 x2) {
    u32 lo = 0;
#pragma unroll
    for (u32 step = FMAX / 2; step >= 1; step >>= 1) {
        const u32 i = lo + step - 1;
        const u64 a = spl1[i];
        bool lt = a < x1;
        if (KW == 2) { const u64 b = spl2[i]; lt = lt || (a == x1 && b < x2); }
        lo += lt ? step : 0u;
    }
    if (LAST) {
        bool eq = spl1[lo] == x1;
        if (KW == 2) eq = eq && spl2[lo] == x2;
        return 2 * lo + (eq ? 1u : 0u);
    }
    return lo;
}

// ---- keys from the text -----------------------------------------------------------------------------------------------------------
// recoded bytes of the tile [t0, t0 + 4096) (+ halo) into LDS; positions >= n read as 0
__device__ __forceinline__ void ws_gen_stage(const WKeyGen& g, size_t t0, const u8* __restrict__ code, u8* __restrict__ sy) {
    const size_t p = t0 + (size_t)threadIdx.x * 16;
    u8 b[16];
    if (p + 16 <= g.n && (((size_t)(g.text + p)) & 15) == 0) {
        const uint4 v = *(const uint4*)(g.text + p);
        const u32 wv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (u8)(wv[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (p + i < g.n) ? g.text[p + i] : (u8)0;
    }
    u32 o[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i >> 2] |= (u32)((p + i < g.n) ? code[b[i]] : (u8)0) << (8 * (i & 3));
    *(uint4*)(sy + (size_t)threadIdx.x * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    if (threadIdx.x < WS_HALO) {
        const size_t q = t0 + WS_TILE + threadIdx.x;
        sy[WS_TILE + threadIdx.x] = (q < g.n) ? code[g.text[q]] : (u8)0;
    }
}
// key of the position whose symbols start at sy[lb]
template <int KW>
__device__ __forceinline__ void ws_key_first(const WKeyGen& g, const u8* sy, int lb, u64& a, u64& b) {
    if (KW == 1) {
        u64 k = 0;
        for (int t = 0; t < g.s; ++t) k = (k << g.b) | sy[lb + t];
        a = k << g.pad; b = 0;
    } else {
        unsigned __int128 k = 0;
        for (int t = 0; t < g.s; ++t) k = (k << g.b) | sy[lb + t];
        k <<= g.pad;
        a = (u64)(k >> 64); b = (u64)k;
    }
}
// key of the next position: the first symbol leaves at the top, `sym` enters behind the last one
template <int KW>
__device__ __forceinline__ void ws_key_roll(const WKeyGen& g, u64& a, u64& b, u32 sym) {
    if (KW == 1) a = (a << g.b) | ((u64)sym << g.pad);
    else {
        unsigned __int128 k = ((unsigned __int128)a << 64) | b;
        k = (k << g.b) | ((unsigned __int128)sym << g.pad);
        a = (u64)(k >> 64); b = (u64)k;
    }
}
// key of text position p straight from global memory (samples, unit borders).  The s <= 64 bytes are fetched as 8-byte words (round
// 5: one byte load per symbol made the sample kernel re-fetch its line for every symbol -- 24 GB of HBM reads per step for 16 M
// samples of 25 symbols, with a hundred thousand of them in flight the lines did not survive in L2 between two loads of a thread).
template <int KW>
__device__ __forceinline__ void ws_key_global(const WKeyGen& g, const u8* __restrict__ code, size_t p, u64& a, u64& b) {
    unsigned __int128 k = 0;
    const int nw = (g.s + 7) >> 3;
    u64 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        w[i] = 0;
        if (i < nw) {
            const size_t q = p + (size_t)i * 8;
            if (q + 8 <= g.n) __builtin_memcpy(&w[i], g.text + q, 8);
            else for (int t = 0; t < 8; ++t) if (q + t < g.n) w[i] |= (u64)g.text[q + t] << (8 * t);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < nw) {
            for (int t = 0; t < 8 && i * 8 + t < g.s; ++t) {
                const size_t q = p + (size_t)(i * 8 + t);
                k = (k << g.b) | ((q < g.n) ? (u32)code[(w[i] >> (8 * t)) & 0xFFu] : 0u);
            }
        }
    }
    if (KW == 1) { a = (u64)k << g.pad; b = 0; }
    else { k <<= g.pad; a = (u64)(k >> 64); b = (u64)k; }
}

// ---- sampling ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t ws_sample_pos(u32 i, size_t n) {
    const u64 h = ((u64)i + 1) * 0x9E3779B97F4A7C15ull;
    return (size_t)(((unsigned __int128)(h ^ (h >> 29)) * n) >> 64);
}
template <int KW, bool GEN>
__global__ __launch_bounds__(256) void ws_sample_kernel(const u64* __restrict__ k1, const u64* __restrict__ k2, WKeyGen g, size_t n, u32 S,
                                                        u64* __restrict__ o1, u64* __restrict__ o2, u32* __restrict__ idx) {
    __shared__ u8 code[256];
    if (GEN) { code[threadIdx.x] = g.code[threadIdx.x]; __syncthreads(); }
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const size_t p = ws_sample_pos(i, n);
    u64 a, b = 0;
    if (GEN) ws_key_global<KW>(g, code, p, a, b);
    else { a = k1[p]; if (KW == 2) b = k2[p]; }
    o1[i] = a;
    if (KW == 2) o2[i] = b;
    idx[i] = i;
}
__global__ void ws_pick_kernel(const u64* __restrict__ s1, const u64* __restrict__ s2, u32 NS, u32 os, u64* __restrict__ sp1, u64* __restrict__ sp2) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NS) { sp1[i] = s1[(size_t)(i + 1) * os - 1]; if (sp2) sp2[i] = s2[(size_t)(i + 1) * os - 1]; }
    else if (i == NS) { sp1[i] = ~0ull; if (sp2) sp2[i] = ~0ull; }
}

// ---- LSD sort of wide records through the 64-bit radix sort (samples, small lists, oversized leaves) ----------------------------------
__global__ void ws_iota_kernel(u32* __restrict__ p, size_t m) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) p[i] = (u32)i;
}
__global__ void ws_gather64_kernel(const u64* __restrict__ src, const u32* __restrict__ perm, size_t m, u64* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[i] = src[perm[i]];
}
__global__ void ws_gather32_kernel(const u32* __restrict__ src, const u32* __restrict__ perm, size_t m, u32* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[i] = src[perm[i]];
}
// (k1[0], k2[0], v[0]) -> (k1[1], k2[1], v[1]) sorted by (k1, k2), stable; k2 / v may be null pairs.  Scratch from the arena.
static void ws_lsd_sort_wide(Ctx& c, u64* k1[2], u64* k2[2], u32* v[2], size_t m, int k1_bits) {
    if (m == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    const unsigned g = cdiv(m, 256);
    u64* A[2] = { c.arena.get<u64>(m), c.arena.get<u64>(m) };
    u32* P[2] = { c.arena.get<u32>(m), c.arena.get<u32>(m) };
    ws_iota_kernel<<<g, 256, 0, s>>>(P[0], m);
    LAUNCH_CHECK();
    int x = 0;
    if (k2) {
        HIP_TRY(hipMemcpyAsync(A[0], k2[0], m * sizeof(u64), hipMemcpyDeviceToDevice, s));
        x = radix_sort_pairs_u64(c, A, P, m, 0, 64);
    }
    // stable by k1 on top of the k2 order
    ws_gather64_kernel<<<g, 256, 0, s>>>(k1[0], P[x], m, A[x ^ 1]);
    LAUNCH_CHECK();
    u64* B[2] = { A[x ^ 1], A[x] };
    u32* Q[2] = { P[x], P[x ^ 1] };
    const int y = radix_sort_pairs_u64(c, B, Q, m, 0, k1_bits < 1 ? 1 : (k1_bits > 64 ? 64 : k1_bits));
    HIP_TRY(hipMemcpyAsync(k1[1], B[y], m * sizeof(u64), hipMemcpyDeviceToDevice, s));
    if (k2) { ws_gather64_kernel<<<g, 256, 0, s>>>(k2[0], Q[y], m, k2[1]); LAUNCH_CHECK(); }
    if (v) { ws_gather32_kernel<<<g, 256, 0, s>>>(v[0], Q[y], m, v[1]); LAUNCH_CHECK(); }
    c.arena.release(mark);
}

// ---- count ------------------------------------------------------------------------------------------------------------------------
template <int KW, bool GEN, bool LAST, int FMAX>
__global__ __launch_bounds__(256) void ws_count_kernel(WSLevel P, WKeyGen g, u32 rows) {
    __shared__ u32 hist[2 * FMAX];
    __shared__ u64 spl1[FMAX];
    __shared__ u64 spl2[KW == 2 ? FMAX : 1];
    __shared__ u8 code[256];
    __shared__ __align__(16) u8 sy[GEN ? WS_TILE + WS_HALO : 16];
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    u32 s, cnt; size_t base;
    if (!ws_row(P, row, s, base, cnt)) return;
    if (!GEN && (s % P.sub) < P.pre_counts) return;          // (histogram and digits of this piece come from the chunk-wise pass behind the upload)
    for (u32 i = threadIdx.x; i < P.D; i += 256) hist[i] = 0;
    if (cnt == 0) {
        for (u32 i = threadIdx.x; i < P.D; i += 256) P.counts[(size_t)row * P.D + i] = 0;
        return;
    }
    ws_load_splitters<KW, LAST, FMAX>(P, s, spl1, spl2);
    if (GEN) code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    if (GEN) { ws_gen_stage(g, P.gen_off + base, code, sy); __syncthreads(); }
    if (GEN) {
        const int lb = wave_id() * (64 * WS_ITEMS) + lane_id() * WS_ITEMS;      // a lane owns 16 consecutive positions
        u64 ka, kb;
        ws_key_first<KW>(g, sy, lb, ka, kb);
        u32 pk[WS_ITEMS / 2];
#pragma unroll
        for (int j = 0; j < WS_ITEMS; ++j) {
            const bool valid = (u32)(lb + j) < cnt;
            const u32 d = valid ? ws_digit<KW, LAST, FMAX>(spl1, spl2, ka, kb) : 0u;
            ws_key_roll<KW>(g, ka, kb, sy[lb + j + g.s]);
            pk[j >> 1] = (j & 1) ? (pk[j >> 1] | (d << 16)) : d;
            const u32 d0 = __builtin_amdgcn_readfirstlane(d);
            if (__all(valid && d == d0)) { if (lane_id() == 0) atomicAdd(&hist[d0], 64u); }
            else if (valid) atomicAdd(&hist[d], 1u);
        }
        uint4* dp = (uint4*)(P.digits + base + lb);                             // 32 bytes, 32-byte aligned; the array is padded to whole tiles
        dp[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dp[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    } else {
        const bool have_digits = (s % P.sub) < P.pre_digits;          // (piece s = bucket * chunks + chunk)
        const u32 lb = wave_id() * (64 * WS_ITEMS) + lane_id();
        const u64* kp1 = P.k1_in + base + lb;
        const u64* kp2 = KW == 2 ? P.k2_in + base + lb : nullptr;
        u16* dgp = P.digits + base + lb;
#ifndef TDC_WC_PF
#define TDC_WC_PF 1      // (measured at 2e9 B, class rs_count_kernel: 0 -> 38.4 ms, 1 -> 35.5, 2 -> 37.8, 4 -> 37.4)
#endif
#if TDC_WC_PF > 0
        // the keys of the next TDC_WC_PF rows are requested before a row is searched (the search is a chain of dependent LDS loads: without
        // this a wave has one row's 16 bytes per lane in flight, and none while it searches)
        constexpr int PF = TDC_WC_PF;
        u64 q1[PF], q2[PF];
        if (!have_digits) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const bool in = lb + (u32)j * 64 < cnt;
                q1[j] = in ? kp1[j * 64] : 0ull;
                q2[j] = (KW == 2 && in) ? kp2[j * 64] : 0ull;
            }
        }
#pragma unroll
        for (int j = 0; j < WS_ITEMS; ++j) {
            const u32 e = lb + (u32)j * 64;
            const bool valid = e < cnt;
            u32 d = 0;
            if (have_digits) { if (valid) d = dgp[j * 64]; }
            else {
                const u64 x1 = q1[j % PF], x2 = q2[j % PF];
                if (j + PF < WS_ITEMS) {
                    const bool in = lb + (u32)(j + PF) * 64 < cnt;
                    q1[j % PF] = in ? kp1[(j + PF) * 64] : 0ull;
                    q2[j % PF] = (KW == 2 && in) ? kp2[(j + PF) * 64] : 0ull;
                }
                d = valid ? ws_digit<KW, LAST, FMAX>(spl1, spl2, x1, x2) : 0u;
                if (valid) dgp[j * 64] = (u16)d;
            }
            const u32 d0 = __builtin_amdgcn_readfirstlane(d);
            if (__all(valid && d == d0)) { if (lane_id() == 0) atomicAdd(&hist[d0], 64u); }
            else if (valid) atomicAdd(&hist[d], 1u);
        }
#else
#pragma unroll 4
        for (int j = 0; j < WS_ITEMS; ++j) {
            const u32 e = lb + (u32)j * 64;
            const bool valid = e < cnt;
            u32 d = 0;
            if (have_digits) { if (valid) d = dgp[j * 64]; }
            else {
                const u64 x1 = valid ? kp1[j * 64] : 0ull;
                const u64 x2 = (KW == 2 && valid) ? kp2[j * 64] : 0ull;
                d = valid ? ws_digit<KW, LAST, FMAX>(spl1, spl2, x1, x2) : 0u;
                if (valid) dgp[j * 64] = (u16)d;
            }
            const u32 d0 = __builtin_amdgcn_readfirstlane(d);
            if (__all(valid && d == d0)) { if (lane_id() == 0) atomicAdd(&hist[d0], 64u); }
            else if (valid) atomicAdd(&hist[d], 1u);
        }
#endif
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < P.D; i += 256) P.counts[(size_t)row * P.D + i] = hist[i];
}

// ---- scatter ----------------------------------------------------------------------------------------------------------------------
// Ranks inside the tile as in ssort.hip (LDS atomics with the two-group peel on the inner levels; the wave-level LDS match on the last
// level, whose equality digits are skewed by construction).  The tile is then written stream by stream -- positions, k1, k2 -- each
// staged in LDS in digit order first, so that consecutive lanes write consecutive records of a digit's run and the record width
// costs no registers: keys are loaded (or, on the text level, generated) right before their stream is staged.
template <int KW, bool GEN, bool LAST, int FMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FMAX > 256 ? 3 : 4, FMAX > 256 ? 3 : 4))) void ws_scatter_kernel(WSLevel P, WKeyGen g, u32 rows) {
    constexpr int DMAX = LAST ? 2 * FMAX : FMAX;
    // ranks from the wave-level LDS match (its tables need 8 bytes per wave and digit).  Round 3 measured it on the inner levels too: LDS
    // atomics with the two-group peel serialise on the 64 counters of a skewed level -- 22.6 -> 19 ms for level 2 at 2e9
    constexpr bool MATCH = FMAX <= 256;
    __shared__ u32 tcnt[MATCH ? 4 : DMAX];
    __shared__ __align__(16) u16 wcnt[MATCH ? 4 : 1][MATCH ? DMAX : 8];
    __shared__ u32 gbase[DMAX];
    __shared__ __align__(16) u64 stage[WS_TILE];
    __shared__ u32 scan_sm[5];
    __shared__ u8 code[GEN ? 256 : 4];
    __shared__ __align__(16) u8 sy[GEN ? WS_TILE + WS_HALO : 16];
    const int lane = lane_id(), w = wave_id();
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    u32 s, cnt; size_t base;
    if (!ws_row(P, row, s, base, cnt) || cnt == 0) return;
    u16* stage_d = (u16*)(stage + 2048);                    // digits of the staged values: second half of the buffer
    u32* stage32 = (u32*)stage;
    unsigned long long* M = (unsigned long long*)stage + 1024 + w * (MATCH ? DMAX : 0);    // MATCH: lane-mask tables of the LDS match (bytes 8 K .. 24 K)
    if constexpr (MATCH) { for (int i = threadIdx.x; i < 4 * DMAX; i += 256) { (&wcnt[0][0])[i] = 0; ((unsigned long long*)stage)[1024 + i] = 0; } }
    else { for (int i = threadIdx.x; i < DMAX; i += 256) tcnt[i] = 0; }
    if (GEN) code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    if (GEN) { ws_gen_stage(g, P.gen_off + base, code, sy); __syncthreads(); }

    u32 ld[WS_ITEMS];                                       // digit << 16 | rank inside the digit's (wave's) run, later position in the sorted tile
    const u32 lbs = GEN ? (u32)(w * (64 * WS_ITEMS) + lane * WS_ITEMS) : (u32)(w * (64 * WS_ITEMS) + lane);
    const u32 estep = GEN ? 1u : 64u;                       // element e of item j: lbs + j * estep
    // Round 5: on the carried levels the positions and the first key word of the thread's records are requested up front, the second key
    // word while the first one is staged -- the staging phases then do not start with a round trip to HBM each (registers: 118 of the 128
    // the occupancy of four workgroups per CU leaves).
#ifndef TDC_WS_PRELOAD
#define TDC_WS_PRELOAD 1
#endif
    constexpr bool PRE = TDC_WS_PRELOAD && !GEN;
    u32 vpre[PRE ? WS_ITEMS : 1];
    u64 k1pre[PRE ? WS_ITEMS : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < WS_ITEMS; ++j) {
            const bool in = lbs + (u32)j * 64 < cnt;
            vpre[j] = in ? P.v_in[base + lbs + (u32)j * 64] : 0u;
            k1pre[j] = in ? P.k1_in[base + lbs + (u32)j * 64] : 0ull;
        }
    }
    {
        u32 dg[WS_ITEMS];
        if (GEN) {
            const uint4* dp = (const uint4*)(P.digits + base + lbs);
            const uint4 q0 = dp[0], q1 = dp[1];
            const u32 pk[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
            for (int j = 0; j < WS_ITEMS; ++j) dg[j] = (pk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        } else {
            const u16* dgp = P.digits + base + lbs;
#pragma unroll
            for (int j = 0; j < WS_ITEMS; ++j) dg[j] = (lbs + (u32)j * 64 < cnt) ? (u32)dgp[j * 64] : 0u;
        }
        if constexpr (MATCH) {
            u16* mycnt = wcnt[w];
            const u64 lanebit = 1ull << lane;
            const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
            for (int j = 0; j < WS_ITEMS; ++j) {
                const bool valid = lbs + (u32)j * estep < cnt;
                const u32 d = valid ? dg[j] : 0u;
#if TDC_WS_PEEL > 0
                const u64 peers = wave_match_peel<TDC_WS_PEEL, 8>(M, d, valid, lanebit);
#else
                const u64 peers = wave_match_lds(M, d, valid, lanebit);
#endif
                const u32 prefix = lds_load(&mycnt[d]);
                const u32 rank = (u32)__popcll(peers & lt_mask);
                ld[j] = (d << 16) | (prefix + rank);
                if (valid && rank == 0) lds_store(&mycnt[d], (u16)(prefix + (u32)__popcll(peers)));
            }
        } else {
#pragma unroll
            for (int j = 0; j < WS_ITEMS; ++j) {
                const bool valid = lbs + (u32)j * estep < cnt;
                const u32 d = valid ? dg[j] : 0u;
                u32 rank = 0;
                u64 rem = __ballot(valid);
                bool done = !valid;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    if (!rem) break;
                    const int lead = __ffsll((long long)rem) - 1;
                    const u32 dl = __shfl(d, lead, 64);
                    const u64 grp = __ballot(!done && d == dl);
                    const u32 cg = (u32)__popcll(grp);
                    if (cg < 8) break;
                    u32 b0 = 0;
                    if (lane == lead) b0 = atomicAdd(&tcnt[dl], cg);
                    b0 = __shfl(b0, lead, 64);
                    if ((grp >> lane) & 1ull) { rank = b0 + (u32)__popcll(grp & ((1ull << lane) - 1)); done = true; }
                    rem &= ~grp;
                }
                if (!done) rank = atomicAdd(&tcnt[d], 1u);
                ld[j] = (d << 16) | rank;
            }
        }
    }
    __syncthreads();
    {   // digit runs inside the sorted tile (exclusive scan over the digits), global base
        const u32 t = threadIdx.x;
        u32 tot[DMAX / 256], sum = 0;
#pragma unroll
        for (int q = 0; q < DMAX / 256; ++q) {
            const u32 d = t * (DMAX / 256) + q;
            if constexpr (MATCH) tot[q] = (u32)wcnt[0][d] + wcnt[1][d] + wcnt[2][d] + wcnt[3][d];
            else tot[q] = tcnt[d];
            sum += tot[q];
        }
        u32 total;
        u32 start = block_exclusive_sum<u32, 4>(sum, scan_sm, total);
#pragma unroll
        for (int q = 0; q < DMAX / 256; ++q) {
            const u32 d = t * (DMAX / 256) + q;
            if constexpr (MATCH) {
                u32 run = start;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const u32 cw = wcnt[i][d]; wcnt[i][d] = (u16)run; run += cw; }
            } else tcnt[d] = start;
            gbase[d] = (d < P.D ? P.counts[(size_t)row * P.D + d] : 0u) - start;
            start += tot[q];
        }
    }
    __syncthreads();                                       // (the match tables in the staging buffer are dead from here on)
    u32 dst[WS_ITEMS];
#pragma unroll
    for (int j = 0; j < WS_ITEMS; ++j) {
        const u32 e = lbs + (u32)j * estep;
        const u32 d = ld[j] >> 16;
        u32 pos = ld[j] & 0xFFFFu;
        if constexpr (MATCH) pos += (u32)wcnt[w][d]; else pos += tcnt[d];
        ld[j] = pos;
        if (e < cnt) { stage32[pos] = GEN ? (u32)(P.gen_off + base + e) : (PRE ? vpre[PRE ? j : 0] : P.v_in[base + e]); stage_d[pos] = (u16)d; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < WS_ITEMS; ++r) {
        const u32 sp = (u32)r * 256 + threadIdx.x;
        dst[r] = 0xFFFFFFFFu;
        if (sp < cnt) {
            dst[r] = gbase[stage_d[sp]] + sp;
            P.v_out[dst[r]] = stage32[sp];
        }
    }
    __syncthreads();
    // key words, one stream at a time
    u64 k2pre[(PRE && KW == 2) ? WS_ITEMS : 1];
#pragma unroll
    for (int word = 0; word < KW; ++word) {
        if (GEN) {
            u64 ka, kb;
            ws_key_first<KW>(g, sy, (int)lbs, ka, kb);
#pragma unroll
            for (int j = 0; j < WS_ITEMS; ++j) {
                if (lbs + (u32)j < cnt) stage[ld[j]] = word == 0 ? ka : kb;
                ws_key_roll<KW>(g, ka, kb, sy[lbs + j + g.s]);
            }
        } else {
            if constexpr (PRE) {
                if (word == 0) {
                    // (the second key word goes on its way before the first one is staged)
                    if (KW == 2) {
#pragma unroll
                        for (int j = 0; j < WS_ITEMS; ++j) k2pre[j] = (lbs + (u32)j * 64 < cnt) ? P.k2_in[base + lbs + (u32)j * 64] : 0ull;
                    }
#pragma unroll
                    for (int j = 0; j < WS_ITEMS; ++j) if (lbs + (u32)j * 64 < cnt) stage[ld[j]] = k1pre[PRE ? j : 0];
                } else {
#pragma unroll
                    for (int j = 0; j < WS_ITEMS; ++j) if (lbs + (u32)j * 64 < cnt) stage[ld[j]] = k2pre[(PRE && KW == 2) ? j : 0];
                }
            } else {
                const u64* kp = (word == 0 ? P.k1_in : P.k2_in) + base + lbs;
                u64 kk[WS_ITEMS];
#pragma unroll
                for (int j = 0; j < WS_ITEMS; ++j) kk[j] = (lbs + (u32)j * 64 < cnt) ? kp[j * 64] : 0ull;
#pragma unroll
                for (int j = 0; j < WS_ITEMS; ++j) if (lbs + (u32)j * 64 < cnt) stage[ld[j]] = kk[j];
            }
        }
        __syncthreads();
        u64* outp = word == 0 ? P.k1_out : P.k2_out;
#pragma unroll
        for (int r = 0; r < WS_ITEMS; ++r) {
            const u32 sp = (u32)r * 256 + threadIdx.x;
            if (dst[r] != 0xFFFFFFFFu) outp[dst[r]] = stage[sp];
        }
        if (word + 1 < KW) __syncthreads();
    }
}

// ---- leaf sort ------------------------------------------------------------------------------------------------------------------------
constexpr int WL_NW = 8;             // waves per leaf workgroup
#ifndef WL_NBMAX
#define WL_NBMAX 51
#endif
#ifndef TDC_WL_PEEL
#define TDC_WL_PEEL 1
#endif
#ifndef TDC_WL_MING
#define TDC_WL_MING 8
#endif
__device__ __forceinline__ u32 wl_rank(u32 d, bool valid, u32* mycnt, unsigned long long* M, u64 lanebit, u64 lt_mask) {
#if TDC_WL_PEEL > 0
    const u64 peers = wave_match_peel<TDC_WL_PEEL, TDC_WL_MING>(M, d, valid, lanebit);
#else
    const u64 peers = wave_match_lds(M, d, valid, lanebit);
#endif
    const u32 prefix = lds_load(&mycnt[d]);
    const u32 rank = (u32)__popcll(peers & lt_mask);
    if (valid && rank == 0) lds_store(&mycnt[d], prefix + (u32)__popcll(peers));
    return prefix + rank;
}
// A "unit" is a slot range [unit_rng[2u], unit_rng[2u + 1]) of at most 8192 records.  Stage 0: the units of ss_build_units (whole
// leaves).  Later stages: long runs of tying records that the counting kernel handed back (see ws_leaf_count_kernel).
struct WLeaf {
    u64* k1; u64* k2; u32* v;
    const u32* unit_rng;
    u8* flags; u8* lcp;                  // head flag of every slot; suffix mode: LCP (symbols) of every head with its predecessor
    u32* d_err;
    u32 inv;                             // ceil(65536 / b)
    u32 cmax;                            // counting limit (WS_CMAX; 1 in the test mode that hands every run back)
};
// lists a leaf kernel appends to
struct WLists {
    u32* rlist;         // kernel A: units whose records tie on k1 (not pure, nothing truncated): runs to be ordered by k2
    u32* tlist;         // kernel A: units sorted by a TRUNCATED word that left ties (bit 31: pure, the word was k2)
    u32* counters;      // [0] = |rlist|, [1] = |tlist|
};
struct WEmit {          // the counting kernel: long runs become units of the next stage
    u32* rng;           // pairs (first slot, end slot | pure << 31), indexed by first slot / div: a run is longer than cmax = div - 1, so
    u32 div;            // no two runs share an entry -- no atomics; empty entries are (0, 0).  ws_emit_compact_kernel lists them by class
};

template <int ROWS, int NW = WL_NW>
struct WLState {
    u32 (*wcnt)[256]; u32 (*wst)[256]; unsigned long long (*wm)[256]; u64* stage;
    u32 m, wbase;
    int lane, w;
};

// Stable LSD sort of the unit's words by the bits [shift0, shift0 + nb); on return k[j] = word at slot wbase + 64 j.
// Two workgroup barriers per pass (ssort.hip needs four): every wave derives the starts of ITS (wave, digit) runs from all waves'
// counters by itself (eight 16-byte loads per lane) instead of waiting for wave 0 to do it for everybody, and the match tables of
// the ranking have an LDS area of their own, so the ranks of the next pass may be taken while other waves still read the staging buffer.
template <int ROWS, int NW>
__device__ __forceinline__ void wl_lsd(const WLState<ROWS, NW>& T, u64 (&k)[ROWS], int shift0, int nb) {
    u32* mycnt = T.wcnt[T.w];
    u32* myst = T.wst[T.w];
    unsigned long long* M = T.wm[T.w];
    const u64 lanebit = 1ull << T.lane;
    const u64 lt_mask = (T.lane == 0) ? 0ull : (~0ull >> (64 - T.lane));
    for (int i = T.lane; i < 256; i += 64) { mycnt[i] = 0; M[i] = 0; }
    u32 loc[ROWS];
    for (int shift = shift0; shift < shift0 + nb; shift += 8) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            loc[j] = wl_rank((u32)(k[j] >> shift) & 255u, T.wbase + (u32)j * 64 < T.m, mycnt, M, lanebit, lt_mask);
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (overlap the LDS operations of a few rows, not of all: registers)
        }
        __syncthreads();
        {   // starts of this wave's runs: lane l owns the digits 4l .. 4l+3
            u32 t0 = 0, t1 = 0, t2 = 0, t3 = 0, p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const uint4 cc = *(const uint4*)&T.wcnt[i][4 * T.lane];
                if (i < T.w) { p0 += cc.x; p1 += cc.y; p2 += cc.z; p3 += cc.w; }
                t0 += cc.x; t1 += cc.y; t2 += cc.z; t3 += cc.w;
            }
            const u32 sum = t0 + t1 + t2 + t3;
            const u32 r0 = wave_inclusive_sum(sum) - sum;
            *(uint4*)&myst[4 * T.lane] = make_uint4(r0 + p0, r0 + t0 + p1, r0 + t0 + t1 + p2, r0 + t0 + t1 + t2 + p3);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
            if (T.wbase + (u32)j * 64 < T.m) T.stage[loc[j] + lds_load(&myst[(u32)(k[j] >> shift) & 255u])] = k[j];
        __syncthreads();
        for (int i = T.lane; i < 256; i += 64) mycnt[i] = 0;    // (every wave has read the counters: they did so before their scatter)
#pragma unroll
        for (int j = 0; j < ROWS; ++j) if (T.wbase + (u32)j * 64 < T.m) k[j] = T.stage[T.wbase + (u32)j * 64];
    }
    __syncthreads();                                            // (callers reuse the staging buffer)
}

template <int ROWS, int NW>
__device__ __forceinline__ void wl_minmax(u64 (*red)[NW], const u64 (&k)[ROWS], u32 m, u32 wbase, int lane, int w, u64& kmin, u64& kmax) {
    kmin = ~0ull; kmax = 0;
#pragma unroll
    for (int j = 0; j < ROWS; ++j)
        if (wbase + (u32)j * 64 < m) { kmin = k[j] < kmin ? k[j] : kmin; kmax = k[j] > kmax ? k[j] : kmax; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 o1 = __shfl_xor(kmin, d, 64), o2 = __shfl_xor(kmax, d, 64);
        kmin = o1 < kmin ? o1 : kmin; kmax = o2 > kmax ? o2 : kmax;
    }
    if (lane == 0) { red[0][w] = kmin; red[1][w] = kmax; }
    __syncthreads();
    kmin = red[0][0]; kmax = red[1][0];
#pragma unroll
    for (int i = 1; i < NW; ++i) { kmin = red[0][i] < kmin ? red[0][i] : kmin; kmax = red[1][i] > kmax ? red[1][i] : kmax; }
    __syncthreads();
}

// the run of slot s in the head bitmap hb (bit m is set: end sentinel; bit 0 is set): 0 = singleton, 1 = run [rs, re) of 2 .. cmax
// slots, 2 = longer run
__device__ __forceinline__ int wl_run(const u64* hb, u32 s, u32 m, u32 cmax, u32& rs, u32& re) {
    const u32 wi = s >> 6, bt = s & 63, wlast = m >> 6;
    const u64 cw = hb[wi];
    u64 x = cw & ((bt == 63) ? ~0ull : ((2ull << bt) - 1ull));
    u32 wl = wi;
    while (!x && wl > 0 && wi - wl < WS_CMAX / 64 + 1) x = hb[--wl];
    u64 y = (bt == 63) ? 0ull : (cw & (~0ull << (bt + 1)));
    u32 wr = wi;
    while (!y && wr < wlast && wr - wi < WS_CMAX / 64 + 1) y = hb[++wr];
    if (!x || !y) return 2;
    rs = wl * 64 + 63 - (u32)__builtin_clzll(x);
    re = wr * 64 + (u32)__builtin_ctzll(y);
    const u32 len = re - rs;
    if (len == 1) return 0;
    return len <= cmax ? 1 : 2;
}
// end of the run that starts at slot s (the next head behind s; the sentinel at m ends the walk)
__device__ __forceinline__ u32 wl_run_end(const u64* hb, u32 s) {
    u32 wi = s >> 6;
    const u32 bt = s & 63;
    u64 y = (bt == 63) ? 0ull : (hb[wi] & (~0ull << (bt + 1)));
    while (!y) y = hb[++wi];
    return wi * 64 + (u32)__builtin_ctzll(y);
}
// ---- leaf kernel A: one workgroup (512 threads) sorts one unit of <= ROWS * 512 records by its first differing word -----------------
// X = k1, or k2 when all k1 of the unit are equal ("pure": a frequent first word fills whole leaves; so does every long run of equal
// k1 that comes back as a unit of a later stage).  Composite-word LSD passes as in ssort.hip: (differing bits of X) << 13 | slot.  When
// X differs in more than 51 bits the composite takes the TOP 51 of them; records that tie on those stay a run.  Written back: the
// positions in sorted order, a head flag per slot (slot starts a run of tying records), suffix mode: the LCP (symbols) of every head
// with its predecessor; X and k2 in the same order where a later kernel (or the caller: PAIRS) reads them.  Units with runs left go
// to `rlist` (ties on the whole of k1: to be ordered by k2) or `tlist` (ties on a truncated word: to be ordered by the word itself
// first); the counting kernel takes them from there.
#ifndef TDC_WL_FILL
#define TDC_WL_FILL 2
#endif
#ifndef TDC_WL_PRE
#define TDC_WL_PRE 1
#endif
#ifndef TDC_WL_W58
#define TDC_WL_W58 6
#endif
#ifndef TDC_WL_W44
#define TDC_WL_W44 6
#endif
#ifndef TDC_WL_W4X
#define TDC_WL_W4X 5
#endif
// (waves per SIMD the register budget is set for.  The kernel waits on LDS most of the time, so its throughput follows the number of
//  resident workgroups: the 8-wave instances with up to 5 rows hold 53 KB of LDS -- three of them fit a CU if they stay within 80
//  registers (two with the 128 of round 4: leaf stage 48.9 -> 45.4 ms); the 4-wave instances with up to 5 rows hold 27 KB)
#define WL_WPE(R, W) ((W) == 8 && (R) <= 5 ? TDC_WL_W58 : ((W) == 4 && (R) <= 5 ? TDC_WL_W44 : ((W) == 4 && (R) <= 8 ? TDC_WL_W4X : 4)))
#ifdef TDC_WL_PROF
__device__ unsigned long long wl_prof[16];
#define WLP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&wl_prof[i], t_ - tp_); tp_ = t_; } } while (0)
#else
#define WLP(i) do { } while (0)
#endif
template <int KW, int ROWS, int NW, bool PAIRS>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WL_WPE(ROWS, NW), WL_WPE(ROWS, NW)))) void ws_leaf_sort_kernel(WLeaf A, const u32* __restrict__ list, u32 count, WLists Q, WEmit E) {
    constexpr u32 CAP = (u32)ROWS * NW * 64;
    constexpr bool FUSE = KW == 2 && CAP <= 2u * NW * 256u;   // the counting step below needs a word per slot in the counter tables
    __shared__ __align__(16) u32 wtab[2][NW][256];
    u32 (*wcnt)[256] = wtab[0];
    u32 (*wst)[256] = wtab[1];
    __shared__ unsigned long long wm[NW][256];
    __shared__ __align__(16) u64 stage[CAP];
    __shared__ u64 red[2][NW];
    __shared__ u32 s_any;
    if (blockIdx.x >= count) return;
    const u32 u = list[blockIdx.x];
    const u32 a = A.unit_rng[2 * u], braw = A.unit_rng[2 * u + 1];
    const bool known_pure = KW == 2 && (braw >> 31) != 0;     // a run handed back by the counting kernel: its records tie on all of k1
    const u32 bnd = braw & 0x7FFFFFFFu;
    const u32 m = bnd - a;
    if (m == 0) return;
    if (m > CAP) { if (threadIdx.x == 0) atomicOr(A.d_err, 2u); return; }
    if (threadIdx.x == 0) { A.flags[a] = 1; s_any = 0; }      // a unit starts at a leaf start / run start
    if (m == 1) return;
#ifdef TDC_WL_PROF
    unsigned long long tp_ = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { atomicAdd(&wl_prof[15], 1ull); atomicAdd(&wl_prof[14], (unsigned long long)m); }
#endif
    const int lane = lane_id(), w = wave_id();
    WLState<ROWS, NW> T;
    T.wcnt = wcnt; T.wst = wst; T.wm = wm; T.stage = stage;
    T.m = m; T.wbase = (u32)w * ROWS * 64 + (u32)lane; T.lane = lane; T.w = w;
    const u32 wbase = T.wbase;
    u64* K1 = A.k1 + a;
    u64* K2 = KW == 2 ? A.k2 + a : nullptr;

    // Round 6: the kernel used to spend more than half of its time waiting for global memory in front of its later phases (positions after the
    // passes: 23 %, the gather of the second words: 8 %, the second words of a pure unit).  Now everything a unit needs is requested in
    // its first instructions -- k1, the positions and the second words, all in slot order -- and travels while the passes run; positions and
    // second words then follow the permutation through LDS (v in the counter tables, k2 in the staging buffer, one barrier for both).
    constexpr bool PRE = TDC_WL_PRE && ROWS <= 8;               // (16 rows: the positions do not fit the counter tables -- the old order of things)
    u64 c[ROWS];
    u32 vp[PRE ? ROWS : 1];
    u64 t2[(PRE && KW == 2) ? ROWS : 1];
    u64 kmin = 0, kmax = 0;
    if (!known_pure) {                                          // (the k1 slots of such a run were never brought into sorted order)
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; c[j] = (L < m) ? K1[L] : 0ull; }
    }
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; vp[j] = (L < m) ? A.v[(size_t)a + L] : 0u; }
        if constexpr (KW == 2) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; t2[j] = (L < m) ? K2[L] : 0ull; }
        }
    }
    if (!known_pure) wl_minmax<ROWS, NW>(red, c, m, wbase, lane, w, kmin, kmax);
    bool pure = false;
    if (kmin == kmax) {
        if (KW == 1) return;                                    // one group: nothing moves
        if constexpr (PRE && KW == 2) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) c[j] = t2[j];
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; c[j] = (L < m) ? K2[L] : 0ull; }
        }
        wl_minmax<ROWS, NW>(red, c, m, wbase, lane, w, kmin, kmax);
        pure = true;
        if (kmin == kmax) return;
    }
    u64* Xp = pure ? K2 : K1;
    const int nbits = 64 - __builtin_clzll(kmin ^ kmax);
    const int nb = nbits > WL_NBMAX ? WL_NBMAX : nbits;         // bits of X in the composite
    const int sh = nbits - nb;                                  // low bits of X left out (ties on the rest are handed on)
    const u64 cmask = (1ull << nb) - 1;
    // The passes sort whole digits: the bits of the last digit that X does not fill take the leading bits of the second word (KW == 2,
    // X = k1) -- the runs that are left then tie on k1 AND on those bits, for nothing (TDC_WL_FILL == 2: one more digit of them)
    int nbf = 0;
    if (TDC_WL_FILL && KW == 2 && !pure && nbits < WL_NBMAX) {
        int tgt = ((nbits + 7) & ~7) + (TDC_WL_FILL == 2 ? 8 : 0);
        if (tgt > WL_NBMAX) tgt = WL_NBMAX;
        nbf = tgt - nbits;
    }
    if (nbf) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            u64 f;
            if constexpr (PRE && KW == 2) f = t2[j] >> (64 - nbf);
            else f = (L < m) ? (K2[L] >> (64 - nbf)) : 0ull;
            c[j] = ((((c[j] & cmask) << nbf) | f) << 13) | (u64)L;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) c[j] = (((c[j] >> sh) & cmask) << 13) | (u64)(wbase + (u32)j * 64);
    }
    WLP(0);
#ifdef TDC_WL_PROF
    if (threadIdx.x == 0) atomicAdd(&wl_prof[13], (unsigned long long)((nb + nbf + 7) / 8));
#endif
    wl_lsd<ROWS, NW>(T, c, 13, nb + nbf);
    WLP(1);
    // heads, LCPs
    const u32 base_bits = pure ? 64u : 0u;
    bool anyrun = false;
    u64* hb = (u64*)&wm[0][0];                                  // head bitmap of the unit (the match tables are dead); bit m: end sentinel
    u8* hf = (u8*)(hb + 128);
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        bool hd = L == 0;
        if (L < m && L > 0) {
            const u64 x = (c[j] >> 13) ^ (stage[L - 1] >> 13);
            A.flags[(size_t)a + L] = x ? 1 : 0;
            if (x) { hd = true; if (!PAIRS) A.lcp[(size_t)a + L] = (u8)(((base_bits + (u32)__builtin_clzll(x) - (u32)sh + (u32)nbf) * A.inv) >> 16); }
            else anyrun = true;
        }
        if (FUSE) { const u64 bm = __ballot(hd || L == m); if (lane == 0) hb[w * ROWS + j] = bm; }
    }
    if (FUSE && threadIdx.x == 0) hb[ROWS * NW] = (m == CAP) ? 1ull : 0ull;
    if (__any(anyrun) && lane == 0) s_any = 1;
    __syncthreads();                                            // (the staged composites have been read)
    const bool runs = s_any != 0;
    const bool trunc_ties = sh > 0 && runs;                     // the counting kernel orders these runs by X itself first
    const bool k2_follows = KW == 2 && !pure && (runs || PAIRS);
    WLP(2);
    // positions (and second words) in sorted order
    u32 vs[ROWS];
    u64 t[KW == 2 ? ROWS : 1];
    if constexpr (PRE) {
        u32* v32 = &wtab[0][0][0];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            if (L < m) { v32[L] = vp[j]; if constexpr (KW == 2) { if (k2_follows) stage[L] = t2[j]; } }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            const u32 idx = (u32)c[j] & 8191u;
            vs[j] = (L < m) ? v32[idx] : 0u;
            if constexpr (KW == 2) t[j] = (L < m && k2_follows) ? stage[idx] : 0ull;
        }
    } else {
        u32* stage32 = (u32*)stage;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) stage32[L] = A.v[(size_t)a + L]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; vs[j] = (L < m) ? stage32[(u32)c[j] & 8191u] : 0u; }
    }
    WLP(3);
    if (FUSE && runs && !pure && !trunc_ties && E.rng) {
        // The runs that are left tie on all of k1: ordered by k2 right here, by counting, as ws_leaf_count_kernel does it for the units
        // of the lists (the sorted k2, the positions and the head bits never leave the workgroup; runs of more than cmax records are
        // handed on).  All per-slot state in LDS, the row loops are real loops.
        u64* Wl = stage;
        u32* P32 = &wtab[0][0][0];
        if constexpr (!PRE) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; t[j] = (L < m) ? K2[c[j] & 8191ull] : 0ull; }
        }
        __syncthreads();                                        // (the staged positions / second words have been read)
        if (PAIRS) {                                            // k1 in sorted order (rebuilt from the composite: sh == 0)
            const u64 high = kmin & ~cmask;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) K1[L] = ((c[j] >> (13 + nbf)) & cmask) | high; }
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) Wl[L] = t[j]; }
        if (threadIdx.x == 0) s_any = 0;                        // from here on: some run of the unit is handed on
        __syncthreads();
        WLP(4);
#pragma unroll 1
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            if (L < m) {
                u32 rs = 0, re = 0;
                u32 tgt = L, h = (u32)((hb[L >> 6] >> (L & 63)) & 1ull);
                const int kind = wl_run(hb, L, m, A.cmax, rs, re);
                if (kind == 1) {
                    u32 less = 0, eqb = 0;
                    const u64 me = Wl[L];
                    for (u32 q = rs; q < re; ++q) {
                        const u64 kq = Wl[q];
                        less += (kq < me) ? 1u : 0u;
                        eqb += (kq == me && q < L) ? 1u : 0u;
                    }
                    tgt = rs + less + eqb;
                    h = (eqb == 0) ? 1u : 0u;
                } else if (kind == 2) {
                    if (h) {                                    // the first slot of a long run hands it on
                        const u32 e = wl_run_end(hb, L);
                        const size_t i = (size_t)((a + L) / E.div);
                        E.rng[2 * i] = a + L;
                        E.rng[2 * i + 1] = (a + e) | 0x80000000u;
                        s_any = 1;
                    }
                    h |= 2u;                                    // (bit 1: the slot lies in a run that goes on -- its second word is written back)
                }
                P32[L] = tgt | (h << 16);
            }
        }
        __syncthreads();
        WLP(5);
        u32 tr[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; tr[j] = (L < m) ? P32[L] : 0u; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            if (L < m) { const u32 d = tr[j] & 0xFFFFu; Wl[d] = t[j]; P32[d] = vs[j]; hf[d] = (u8)(tr[j] >> 16); }
        }
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            if (L < m) {
                const u32 hfl = hf[L];
                const bool h = (hfl & 1u) != 0;
                const bool old = (hb[L >> 6] >> (L & 63)) & 1ull;
                A.v[(size_t)a + L] = P32[L];
                if (PAIRS || (hfl & 2u)) K2[L] = Wl[L];         // (a run that goes on is read from here by the next stage; nobody reads the rest)
                if (h && !old) {
                    const u64 x = Wl[L] ^ Wl[L - 1];
                    A.flags[(size_t)a + L] = 1;
                    if (!PAIRS) A.lcp[(size_t)a + L] = (u8)(((64u + (x ? (u32)__builtin_clzll(x) : 64u)) * A.inv) >> 16);
                }
            }
        }
        WLP(6);
        return;
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) A.v[(size_t)a + L] = vs[j]; }
    if (PAIRS || trunc_ties) {                                  // X in sorted order
        if (sh == 0) {                                          // rebuilt from the composite
            const u64 high = kmin & ~cmask;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) Xp[L] = ((c[j] >> (13 + nbf)) & cmask) | high; }
        } else {                                                // (all reads before the first write)
            u64 tx[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; tx[j] = (L < m) ? Xp[c[j] & 8191ull] : 0ull; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) Xp[L] = tx[j]; }
        }
    }
    if (k2_follows) {                                           // k2 follows
        if constexpr (PRE && KW == 2) {                         // (every slot's old word has been read: they were requested up front)
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) K2[L] = t[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; t[j] = (L < m) ? K2[c[j] & 8191ull] : 0ull; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) K2[L] = t[j]; }
        }
    }
    WLP(7);
    if (threadIdx.x == 0) {
        if (trunc_ties) Q.tlist[atomicAdd(Q.counters + 1, 1u)] = u | (pure ? 0x80000000u : 0u);
        else if (runs && KW == 2 && !pure) Q.rlist[atomicAdd(Q.counters, 1u)] = u;
    }
}

// ---- leaf kernel B: the runs of a unit (head flags as left by kernel A) ordered by the word W, by COUNTING ------------------------
// Every member of a run of 2 .. cmax slots counts the smaller words of its run in LDS (and the equal ones in front of it: stable); new
// heads where the words differ.  `carry` (nullable): a second word array that follows the permutation.  base_bits: key bits in front
// of W (0 for k1, 64 for k2) -- the LCP of a new head is (base_bits + common leading bits of W) / bits per symbol.  A longer run is left
// as it is and, if E.unit_rng is set, becomes a unit of the next stage: kernel A sorts it (its records tie on everything in front of
// W, so it is "pure" there, or differs only in the low bits a truncated composite left out).
// All per-slot state lives in LDS and the row loops are real loops: a few dozen registers.
struct WCount { u64* W; u64* carry; u32 base_bits; };
template <int ROWS, int NW, bool PAIRS>
__global__ __launch_bounds__(NW * 64) void ws_leaf_count_kernel(WLeaf A, const u32* __restrict__ list, u32 count, u32 want_mask, u32 want_value,
                                                                WCount R, WEmit E) {
    constexpr u32 CAP = (u32)ROWS * NW * 64;
    __shared__ u64 Wl[CAP];
    __shared__ u32 P32[CAP];
    __shared__ u8 hf[CAP];
    __shared__ u64 hb[CAP / 64 + 1];
    if (blockIdx.x >= count) return;
    const u32 uraw = list[blockIdx.x];
    if ((uraw & want_mask) != want_value) return;
    const u32 u = uraw & 0x7FFFFFFFu;
    const u32 a = A.unit_rng[2 * u], bnd = A.unit_rng[2 * u + 1] & 0x7FFFFFFFu;
    const u32 m = bnd - a;
    if (m <= 1 || m > CAP) return;
    const int lane = lane_id(), w = wave_id();
    const u32 wbase = (u32)w * ROWS * 64 + (u32)lane;
    u64* W = R.W + a;
#pragma unroll 1
    for (int j = 0; j < ROWS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        Wl[L] = (L < m) ? W[L] : 0ull;
        const bool h = (L < m) && (L == 0 || A.flags[(size_t)a + L] != 0);
        const u64 bm = __ballot(h || L == m);
        if (lane == 0) hb[w * ROWS + j] = bm;
    }
    if (threadIdx.x == 0) hb[ROWS * NW] = (m == CAP) ? 1ull : 0ull;
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < ROWS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        if (L < m) {
            u32 rs = 0, re = 0;
            u32 tgt = L, h = (u32)((hb[L >> 6] >> (L & 63)) & 1ull);
            const int kind = wl_run(hb, L, m, A.cmax, rs, re);
            if (kind == 1) {
                u32 less = 0, eqb = 0;
                const u64 me = Wl[L];
                for (u32 q = rs; q < re; ++q) {
                    const u64 kq = Wl[q];
                    less += (kq < me) ? 1u : 0u;
                    eqb += (kq == me && q < L) ? 1u : 0u;
                }
                tgt = rs + less + eqb;
                h = (eqb == 0) ? 1u : 0u;
            } else if (kind == 2 && h && E.rng) {                // the first slot of a long run hands it on
                const u32 e = wl_run_end(hb, L);
                const size_t i = (size_t)((a + L) / E.div);
                E.rng[2 * i] = a + L;
                E.rng[2 * i + 1] = (a + e) | (R.base_bits ? 0x80000000u : 0u);
            }
            P32[L] = tgt | (h << 16);
        }
    }
    __syncthreads();
    {   // the words, the head bits and the positions to their final slots
        u64 wr[ROWS];
        u32 tr[ROWS], vr[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            wr[j] = (L < m) ? Wl[L] : 0ull;
            tr[j] = (L < m) ? P32[L] : 0u;
            vr[j] = (L < m) ? A.v[(size_t)a + L] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u32 L = wbase + (u32)j * 64;
            if (L < m) { const u32 t = tr[j] & 0xFFFFu; Wl[t] = wr[j]; P32[t] = vr[j]; hf[t] = (u8)(tr[j] >> 16); }
        }
        if (R.carry) {                                          // (reads of the whole unit first, then the scattered writes)
            u64* Cw = R.carry + a;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; wr[j] = (L < m) ? Cw[L] : 0ull; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { const u32 L = wbase + (u32)j * 64; if (L < m) Cw[tr[j] & 0xFFFFu] = wr[j]; }
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < ROWS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        if (L < m) {
            const bool h = hf[L] != 0;
            const bool old = (hb[L >> 6] >> (L & 63)) & 1ull;
            A.v[(size_t)a + L] = P32[L];
            if (PAIRS) W[L] = Wl[L];
            if (h && !old) {
                const u64 x = Wl[L] ^ Wl[L - 1];
                A.flags[(size_t)a + L] = 1;
                if (!PAIRS) A.lcp[(size_t)a + L] = (u8)(((R.base_bits + (x ? (u32)__builtin_clzll(x) : 64u)) * A.inv) >> 16);
            }
        }
    }
}
// the runs handed on by the counting kernel, listed by class.  Runs whose records tie on all of k1 ("pure"): 0 = <= 32 records and
// 1 = <= 64 (lane kernel), 2 = <= 256 and 3 = <= 1024 (wave kernel); everything else: 4 .. 7 = kernel A by size
constexpr u32 EC_TILE = 256 * 32;
constexpr int EC_NCLS = 4 + WIDE_NCLS;
__global__ __launch_bounds__(256) void ws_emit_compact_kernel(const u32* __restrict__ rng, u32 nent, u32* __restrict__ lists, u32 cap, u32* __restrict__ counters) {
    __shared__ u32 cnt[EC_NCLS], base[EC_NCLS], recs[EC_NCLS];
    if (threadIdx.x < EC_NCLS) { cnt[threadIdx.x] = 0; recs[threadIdx.x] = 0; }
    __syncthreads();
    const u32 i0 = blockIdx.x * EC_TILE + threadIdx.x;
    u32 codes[4] = { 0, 0, 0, 0 };                             // 4 bits per entry: class + 1 (0 = empty)
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const u32 i = i0 + (u32)q * 256;
        u32 code = 0, braw_m = 0;
        if (i < nent) {
            const u32 a = rng[2 * (size_t)i], braw = rng[2 * (size_t)i + 1];
            const u32 m = (braw & 0x7FFFFFFFu) - a;
            braw_m = m;
            if (m > 0) {
                if ((braw >> 31) && m <= WS_WAVE_MAX) code = m <= 32 ? 1u : (m <= 64 ? 2u : (m <= 256 ? 3u : 4u));
                else code = 5u + wide_class(m);
            }
        }
        codes[q >> 3] |= code << (4 * (q & 7));
        if (code) { atomicAdd(&cnt[code - 1], 1u); atomicAdd(&recs[code - 1], (braw_m)); }
    }
    __syncthreads();
    if (threadIdx.x < EC_NCLS) {
        const u32 c = cnt[threadIdx.x];
        base[threadIdx.x] = c ? atomicAdd(counters + threadIdx.x, c) : 0u; cnt[threadIdx.x] = 0;
        if (recs[threadIdx.x]) atomicAdd(counters + EC_NCLS + threadIdx.x, recs[threadIdx.x]);      // records per class (the bench table's byte counts)
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const u32 code = (codes[q >> 3] >> (4 * (q & 7))) & 15u;
        if (code) {
            const u32 k = base[code - 1] + atomicAdd(&cnt[code - 1], 1u);
            if (k < cap) lists[(size_t)(code - 1) * cap + k] = i0 + (u32)q * 256;
        }
    }
}
__global__ __launch_bounds__(256) void ws_count_flags_kernel(const u8* __restrict__ flags, size_t n, unsigned long long* __restrict__ d_nonheads) {
    __shared__ u32 part[4];
    u32 cnt = 0;
    const size_t stride = (size_t)gridDim.x * 256 * 16;
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i0 < n; i0 += stride) {
        if (i0 + 16 <= n && (((size_t)flags) & 15) == 0) {
            const uint4 v = *(const uint4*)(flags + i0);
            const u32 wv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int q = 0; q < 4; ++q) cnt += 4u - (u32)__popc(wv[q] & 0x01010101u);
        } else for (size_t i = i0; i < n && i < i0 + 16; ++i) cnt += flags[i] ? 0u : 1u;
    }
    cnt = wave_reduce_sum(cnt);
    if (lane_id() == 0) part[wave_id()] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { const u32 t = part[0] + part[1] + part[2] + part[3]; if (t) atomicAdd(d_nonheads, (unsigned long long)t); }
}

// ---- runs of <= 64 records that tie on k1: one record per LANE ------------------------------------------------------------------------
// A bitonic network over W = 32 or 64 lanes on (word, lane) -- the lane number breaks ties and keeps the padding lanes (all-ones word,
// lane >= m) behind every record.  No LDS, no passes, a dozen registers; W = 32: every wave sorts two runs at once.
template <int W, bool PAIRS>
__global__ __launch_bounds__(256) void ws_run_lane_kernel(WLeaf A, const u32* __restrict__ list, u32 count) {
    const int lane = lane_id();
    const u32 sub = (u32)lane / W, ll = (u32)lane % W;
    const u32 r = (blockIdx.x * 4 + (u32)wave_id()) * (64 / W) + sub;
    u32 a = 0, m = 0;
    if (r < count) { const u32 u = list[r]; a = A.unit_rng[2 * u]; m = (A.unit_rng[2 * u + 1] & 0x7FFFFFFFu) - a; }
    if (m > (u32)W) m = 0;                                     // (not this kernel's class: leave it alone)
    u64* Wd = A.k2 + a;
    u64 key = (ll < m) ? Wd[ll] : ~0ull;
    u32 idl = ll;
    const u32 vold = (ll < m) ? A.v[(size_t)a + ll] : 0u;
#pragma unroll
    for (u32 k2 = 2; k2 <= (u32)W; k2 <<= 1) {
#pragma unroll
        for (u32 j = k2 >> 1; j > 0; j >>= 1) {
            const u64 ok = __shfl_xor(key, (int)j, 64);
            const u32 oi = __shfl_xor(idl, (int)j, 64);
            const bool up = ((ll & k2) == 0);
            const bool lower = ((ll & j) == 0);
            const bool gt = key > ok || (key == ok && idl > oi);           // mine > the partner's
            const bool take = lower ? (gt == up) : (!gt == up);
            if (take) { key = ok; idl = oi; }
        }
    }
    const u64 prev = __shfl_up(key, 1, 64);
    const u32 vnew = __shfl(vold, (int)(sub * W + idl), 64);
    if (ll < m && m > 1) {
        A.v[(size_t)a + ll] = vnew;
        if (PAIRS) Wd[ll] = key;
        if (ll > 0) {
            const u64 x = key ^ prev;
            A.flags[(size_t)a + ll] = x ? 1 : 0;
            if (x && !PAIRS) A.lcp[(size_t)a + ll] = (u8)(((64u + (u32)__builtin_clzll(x)) * A.inv) >> 16);
        }
    }
}

// ---- one WAVE sorts one run of <= CAPR records that tie on k1, by the whole of k2 ---------------------------------------------------------
// The runs of 65 .. 1024 equal first words are the mid-frequency phrases of a text: hundreds of thousands of them, each far too small
// for a 512-thread workgroup (kernel A spends its time in barriers and idle lanes there).  Here a wave owns a run: LSD passes over the
// differing bits of k2 on (word, record number) pairs held in registers, ranks from the wave-level LDS match, no workgroup barrier
// anywhere -- the four waves of a workgroup work on four runs independently.  Written back: positions in k2 order, head flags, LCPs
// (base: the 64 bits of k1), PAIRS: k2 itself.  CAPR = 256 | 1024: the small variant keeps a quarter of the LDS (more waves per CU).
template <int CAPR, bool PAIRS>
__global__ __launch_bounds__(256) void ws_run_wave_kernel(WLeaf A, const u32* __restrict__ list, u32 count) {
    constexpr int WR_ROWS = CAPR / 64;
    __shared__ u64 sk[4][CAPR];
    __shared__ u16 sid[4][CAPR];
    __shared__ __align__(16) u32 cnt[4][256];
    __shared__ u64 mt[4][256];
    const int lane = lane_id(), w = wave_id();
    const u32 r = blockIdx.x * 4 + (u32)w;
    if (r >= count) return;
    const u32 u = list[r];
    const u32 a = A.unit_rng[2 * u], m = (A.unit_rng[2 * u + 1] & 0x7FFFFFFFu) - a;
    if (m <= 1 || m > (u32)CAPR) return;
    u64* W = A.k2 + a;
    u64* K = sk[w]; u16* S = sid[w]; u32* mycnt = cnt[w];
    unsigned long long* M = (unsigned long long*)mt[w];
    const int rows = (int)((m + 63) >> 6);
    u64 k[WR_ROWS];
    u32 id[WR_ROWS];
    u64 orv = 0, andv = ~0ull;
#pragma unroll
    for (int j = 0; j < WR_ROWS; ++j) {
        const u32 L = (u32)j * 64 + (u32)lane;
        k[j] = (j < rows && L < m) ? W[L] : 0ull;
        id[j] = L;
        if (j < rows && L < m) { orv |= k[j]; andv &= k[j]; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { orv |= __shfl_xor(orv, d, 64); andv &= __shfl_xor(andv, d, 64); }
    const u64 diff = orv ^ andv;                               // bits in which the words of the run differ
    for (int i = lane; i < 256; i += 64) { mycnt[i] = 0; M[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    if (diff) {
        const int hi = 64 - __builtin_clzll(diff), lo = __builtin_ctzll(diff);
        const u64 lanebit = 1ull << lane;
        const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        for (int shift = lo; shift < hi; shift += 8) {
            if (((diff >> shift) & 255ull) == 0) continue;     // (a digit no word differs in)
            u32 loc[WR_ROWS];
#pragma unroll
            for (int j = 0; j < WR_ROWS; ++j)
                if (j < rows) loc[j] = wl_rank((u32)(k[j] >> shift) & 255u, (u32)j * 64 + (u32)lane < m, mycnt, M, lanebit, lt_mask);
            __builtin_amdgcn_wave_barrier();
            {   // digit starts: lane l owns the digits 4l .. 4l+3
                const uint4 cc = *(const uint4*)&mycnt[4 * lane];
                const u32 sum = cc.x + cc.y + cc.z + cc.w;
                const u32 r0 = wave_inclusive_sum(sum) - sum;
                *(uint4*)&mycnt[4 * lane] = make_uint4(r0, r0 + cc.x, r0 + cc.x + cc.y, r0 + cc.x + cc.y + cc.z);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < WR_ROWS; ++j) {
                if (j < rows && (u32)j * 64 + (u32)lane < m) {
                    const u32 p = loc[j] + lds_load(&mycnt[(u32)(k[j] >> shift) & 255u]);
                    lds_store(&K[p], k[j]); lds_store(&S[p], (u16)id[j]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < 256; i += 64) mycnt[i] = 0;
#pragma unroll
            for (int j = 0; j < WR_ROWS; ++j) {
                const u32 L = (u32)j * 64 + (u32)lane;
                if (j < rows && L < m) { k[j] = lds_load(&K[L]); id[j] = (u32)lds_load(&S[L]); }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // heads + LCPs from the sorted words (K[] holds them when a pass ran; else all words are equal)
#pragma unroll
    for (int j = 0; j < WR_ROWS; ++j) {
        const u32 L = (u32)j * 64 + (u32)lane;
        if (j < rows && L < m && L > 0) {
            const u64 x = diff ? (k[j] ^ lds_load(&K[L - 1])) : 0ull;
            A.flags[(size_t)a + L] = x ? 1 : 0;
            if (x && !PAIRS) A.lcp[(size_t)a + L] = (u8)(((64u + (u32)__builtin_clzll(x)) * A.inv) >> 16);
        }
    }
    if (!diff) return;
    __builtin_amdgcn_wave_barrier();
    // positions (parked in the staging words), k2
    u32* P = (u32*)K;
    u32 vr[WR_ROWS];
#pragma unroll
    for (int j = 0; j < WR_ROWS; ++j) { const u32 L = (u32)j * 64 + (u32)lane; vr[j] = (j < rows && L < m) ? A.v[(size_t)a + L] : 0u; }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < WR_ROWS; ++j) { const u32 L = (u32)j * 64 + (u32)lane; if (j < rows && L < m) lds_store(&P[L], vr[j]); }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < WR_ROWS; ++j) {
        const u32 L = (u32)j * 64 + (u32)lane;
        if (j < rows && L < m) {
            A.v[(size_t)a + L] = lds_load(&P[id[j]]);
            if (PAIRS) W[L] = k[j];
        }
    }
}

// ---- suffix mode: LCP at the leaf starts ---------------------------------------------------------------------------------------------
// Every non-empty leaf starts a group (keys of different leaves differ); the LCP of its first slot with the slot in front of it is
// counted from the text (two scattered reads per leaf: a few million in all).
template <int KW>
__global__ __launch_bounds__(256) void ws_fix_kernel(const u32* __restrict__ leaf_start, u32 nleaf, const u32* __restrict__ v, WKeyGen g,
                                                     u8* __restrict__ flags, u8* __restrict__ lcp) {
    __shared__ u8 code[256];
    code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    const u32 l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nleaf) return;
    const u32 s = leaf_start[l];
    if (s == leaf_start[l + 1]) return;
    flags[s] = 1;
    if (s == 0) { lcp[0] = 0; return; }
    const size_t pa = v[s - 1], pb = v[s];
    // (eight text bytes per step: distinct bytes have distinct codes, so the first differing byte is the first differing symbol; a
    //  suffix that ends inside the key reads as code 0 from there on, like the keys the sort compared)
    u32 t = 0;
    while (t < (u32)g.s) {
        u64 wa = 0, wb = 0;
        if (pa + t + 8 <= g.n) __builtin_memcpy(&wa, g.text + pa + t, 8);
        else for (int e = 0; e < 8; ++e) if (pa + t + e < g.n) wa |= (u64)g.text[pa + t + e] << (8 * e);
        if (pb + t + 8 <= g.n) __builtin_memcpy(&wb, g.text + pb + t, 8);
        else for (int e = 0; e < 8; ++e) if (pb + t + e < g.n) wb |= (u64)g.text[pb + t + e] << (8 * e);
        bool stop = false;
        for (int e = 0; e < 8 && t < (u32)g.s; ++e, ++t) {
            const u32 ca = (pa + t < g.n) ? code[(wa >> (8 * e)) & 0xFFu] : 0u, cb = (pb + t < g.n) ? code[(wb >> (8 * e)) & 0xFFu] : 0u;
            if (ca != cb) { stop = true; break; }
        }
        if (stop) break;
    }
    lcp[s] = (u8)t;
}
// flags / LCP of a range whose sorted keys are at hand (oversized leaves that went through the LSD fall-back); slot a itself is a
// leaf start (ws_fix_kernel)
template <int KW>
__global__ void ws_range_flags_kernel(const u64* __restrict__ k1, const u64* __restrict__ k2, u32 a, u32 bnd, u32 inv, u8* __restrict__ flags,
                                      u8* __restrict__ lcp) {
    const u32 i = a + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bnd) return;
    const u64 x1 = k1[i] ^ k1[i - 1];
    const u64 x2 = KW == 2 ? (k2[i] ^ k2[i - 1]) : 0ull;
    const bool h = x1 != 0 || x2 != 0;
    flags[i] = h ? 1 : 0;
    if (h && lcp) lcp[i] = (u8)(((x1 ? (u32)__builtin_clzll(x1) : 64u + (u32)__builtin_clzll(x2)) * inv) >> 16);
}

// ---- tables of the level that merges a chunk-wise level 1 -----------------------------------------------------------------------------
// nstart_all[q][b]: first slot of bucket b inside chunk q ([F0] = end of the chunk's part).  Piece b * nch + q = that range; out_start[b] =
// first output slot of bucket b = total size of the buckets in front of it.
__global__ __launch_bounds__(1024) void ws_sub_tables_kernel(const u32* __restrict__ nstart_all, u32 F0, u32 nch, u32* __restrict__ seg_begin,
                                                              u32* __restrict__ seg_end, u32* __restrict__ out_start) {
    __shared__ u32 tot[1024];
    const u32 b = threadIdx.x;
    u32 t = 0;
    if (b < F0) {
        for (u32 q = 0; q < nch; ++q) {
            const u32 lo = nstart_all[(size_t)q * (F0 + 1) + b], hi = nstart_all[(size_t)q * (F0 + 1) + b + 1];
            seg_begin[b * nch + q] = lo; seg_end[b * nch + q] = hi;
            t += hi - lo;
        }
    }
    tot[b] = t;
    __syncthreads();
    if (b == 0) {
        u32 run = 0;
        for (u32 i = 0; i < F0; ++i) { out_start[i] = run; run += tot[i]; }
        out_start[F0] = run;
    }
}
__global__ void ws_sub_nblk_kernel(const u32* __restrict__ seg_begin, const u32* __restrict__ seg_end, u32 nsub, u32 R, u32* __restrict__ nblk) {
    const u32 sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx > nsub) return;
    if (sidx == nsub) { nblk[sidx] = 0; return; }
    const u64 size = seg_end[sidx] - seg_begin[sidx];
    nblk[sidx] = (u32)((size + (u64)WS_TILE * R - 1) / ((u64)WS_TILE * R));
}
__global__ void ws_blkseg_kernel(const u32* __restrict__ blk_start, u32 nseg, u32* __restrict__ blk_seg) {
    const u32 sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nseg) return;
    for (u32 b = blk_start[sidx]; b < blk_start[sidx + 1]; ++b) blk_seg[b] = sidx;
}
__global__ void ws_blk_super_kernel(const u32* __restrict__ blk_start, u32 nsuper, u32 sub, u32* __restrict__ blk_super) {
    const u32 S = blockIdx.x * blockDim.x + threadIdx.x;
    if (S <= nsuper) blk_super[S] = blk_start[(size_t)S * sub];
}
__global__ void ws_set2_kernel(u32* p, u32 a, u32 b) { p[0] = a; p[1] = b; }

// The tile histograms of the pieces whose level-2 digits were computed chunk by chunk behind the upload (wsort_pre_chunk), copied from
// the chunks' count tables into the merging level's: piece s = bucket * nch + chunk; its tile t is row lblk[bucket] * Rq + t of chunk
// q's table (blocks of Rq rows per bucket) and row (blk_start[s]) * R + t of the merged one.  One 64-thread workgroup per merged row.
struct WPreHist { const u32* counts[32]; const u32* blk[32]; u32 R[32]; u32 nch, chunks; };
__global__ __launch_bounds__(64) void ws_pre_hist_kernel(WPreHist H, const u32* __restrict__ blk_start, const u32* __restrict__ blk_seg, u32 nsub, u32 R, u32 D,
                                                         u32 rows, u32* __restrict__ counts) {
    const u32 row = blockIdx.x;
    if (row >= rows) return;
    const u32 blk = row / R;
    if (blk >= blk_start[nsub]) return;
    const u32 s = blk_seg[blk];
    const u32 q = s % H.nch, b = s / H.nch;
    if (q >= H.chunks) return;
    const u32 t = (blk - blk_start[s]) * R + row % R;
    const u32 lb0 = H.blk[q][b], lb1 = H.blk[q][b + 1], Rq = H.R[q];
    const bool in = t < (lb1 - lb0) * Rq;
    const u32* src = H.counts[q] + ((size_t)lb0 * Rq + t) * D;
    for (u32 d = threadIdx.x; d < D; d += 64) counts[(size_t)row * D + d] = in ? src[d] : 0u;
}

// ---- host -------------------------------------------------------------------------------------------------------------------------
void wsort_make_keygen(const Ctx& c, const u8* text, size_t n, u32 sigma, const u8* code, int& KW, WKeyGen& g) {
    const int b = (int)bits_for(sigma > 1 ? sigma - 1 : 1);
    const int per_word = 64 / b;
    KW = c.wsort_kw ? c.wsort_kw : (per_word < 16 ? 2 : 1);
    g.text = text; g.n = n; g.b = b;
    g.s = (64 * KW) / b; if (g.s > 64) g.s = 64;
    if (c.wsort_syms > 0 && g.s > c.wsort_syms) g.s = c.wsort_syms;      // (measurements: what would a narrower record cost in ties?)
    g.pad = 64 * KW - g.s * b;
    g.inv = (65536u + (u32)b - 1) / (u32)b;
    memcpy(g.code, code, 256);
}
bool wsort_applicable(const Ctx& c, size_t n) { return c.wsort && n >= c.wsort_min && n < ((size_t)1 << 32); }
int wsort_result_index(Ctx& c, size_t n) {                    // index of the V buffer that will hold wsort_suffixes' result
    int L; u32 F[3], os;
    ss_fanouts(c, n, L, F, os, (u32)c.wsort_leaf, c.wsort_two);
    if (c.wsort_order == 1 && L == 3) std::swap(F[0], F[2]);   // the widest level first (it runs behind the upload)
    return (L - 1) & 1;
}

namespace {
struct WPlan { int L; u32 F[3]; u32 os, NLr, NS, S; };

template <int KW, bool GEN>
void ws_splitters(Ctx& c, const WPlan& pl, const u64* k1, const u64* k2, const WKeyGen& g, size_t n, u64* sp1, u64* sp2) {
    hipStream_t s = c.stream;
    const size_t m2 = c.arena.mark();
    const u32 S = pl.S;
    u64* a1[2] = { c.arena.get<u64>(S), c.arena.get<u64>(S) };
    u64* a2[2] = { nullptr, nullptr };
    if (KW == 2) { a2[0] = c.arena.get<u64>(S); a2[1] = c.arena.get<u64>(S); }
    u32* iv[2] = { c.arena.get<u32>(S), c.arena.get<u32>(S) };
    ws_sample_kernel<KW, GEN><<<cdiv(S, 256), 256, 0, s>>>(k1, k2, g, n, S, a1[0], a2[0], iv[0]);
    LAUNCH_CHECK();
    if (KW == 1) {
        const int x = radix_sort_pairs_u64(c, a1, iv, S, 0, 64);
        ws_pick_kernel<<<cdiv((size_t)pl.NS + 1, 256), 256, 0, s>>>(a1[x], nullptr, pl.NS, pl.os, sp1, nullptr);
        LAUNCH_CHECK();
    } else {
        ws_lsd_sort_wide(c, a1, a2, nullptr, S, 64);
        ws_pick_kernel<<<cdiv((size_t)pl.NS + 1, 256), 256, 0, s>>>(a1[1], a2[1], pl.NS, pl.os, sp1, sp2);
        LAUNCH_CHECK();
    }
    c.arena.release(m2);
}

// pre != nullptr: level 1 was done chunk by chunk behind the upload (WPre); the levels continue from its buckets and the sorted
// positions land in v_final
template <int KW, bool GEN, bool PAIRS>
int ws_sort_impl(Ctx& c, const WKeyGen* gen, u64* K1[2], u64* K2[2], u32* V[2], size_t n, int k1_bits, u8* flags, u8* lcp8, WSortStats* st,
                 const WPre* pre = nullptr, u32* v_final = nullptr) {
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    WKeyGen g;
    if (gen) g = *gen; else memset(&g, 0, sizeof(g));
    WPlan pl;
    u64* sp1; u64* sp2;
    if (pre) {
        pl.L = pre->L; pl.F[0] = pre->F[0]; pl.F[1] = pre->F[1]; pl.F[2] = pre->F[2]; pl.os = pre->os;
        pl.NLr = pre->NLr; pl.NS = pre->NS; pl.S = pre->S;
        sp1 = pre->sp1; sp2 = pre->sp2;
    } else {
        ss_fanouts(c, n, pl.L, pl.F, pl.os, (u32)c.wsort_leaf, c.wsort_two);
        if (c.wsort_order == 1 && pl.L == 3) std::swap(pl.F[0], pl.F[2]);
        pl.NLr = pl.F[0] * pl.F[1] * pl.F[2]; pl.NS = pl.NLr - 1; pl.S = pl.os * pl.NLr;
        sp1 = c.arena.get<u64>((size_t)pl.NS + 1);
        sp2 = KW == 2 ? c.arena.get<u64>((size_t)pl.NS + 1) : nullptr;
        ws_splitters<KW, GEN>(c, pl, K1[0], KW == 2 ? K2[0] : nullptr, g, n, sp1, sp2);
    }
    const int L = pl.L;
    st->levels = (u32)L; st->range_leaves = pl.NLr; st->samples = pl.S; st->kw = KW;

    unsigned long long* d_nonheads = (unsigned long long*)c.arena.get<u64>(1);
    HIP_TRY(hipMemsetAsync(d_nonheads, 0, sizeof(u64), s));
    if (PAIRS) flags = c.arena.get<u8>(n + 8);                 // (internal in this mode: run heads between the two leaf kernels)
    HIP_TRY(hipMemsetAsync(flags, 0, n, s));

    // ---- partition levels ----
    u16* digits = pre ? pre->digits : c.arena.get<u16>(align_up(n, WS_TILE) + WS_TILE);
    const u32* seg_start = pre ? nullptr : ss_first_segment(c, n);
    u32 nseg = 1;
    int cur = GEN ? -1 : 0;
    u32* Vlast = nullptr;                                      // where the last level put the positions
    if (pre) {
        // ---- the level that merges the chunk-wise level 1: piece (b, q) = bucket b of chunk q, numbered b * nchunks + q, lies wherever
        //      chunk q's scatter put it; the pieces of bucket b share its splitters and write into one output segment ----
        const int l = 1;
        const bool last = (l == L - 1);
        const u32 F0 = pl.F[0], nch = pre->nchunks, nsub = F0 * nch;
        const u32 D = last ? 2 * pl.F[l] : pl.F[l];
        u32 stride = 1;
        for (int q = l + 1; q < L; ++q) stride *= pl.F[q];
        u32* nstart = c.arena.get<u32>((size_t)F0 * D + 1);
        const size_t lm2 = c.arena.mark();
        u32* seg_begin = c.arena.get<u32>(nsub + 1);
        u32* seg_end = c.arena.get<u32>(nsub + 1);
        u32* out_start = c.arena.get<u32>(F0 + 1);
        ws_sub_tables_kernel<<<1, 1024, 0, s>>>(pre->nstart_all, F0, nch, seg_begin, seg_end, out_start);
        LAUNCH_CHECK();
        SegTables Tb;
        {
            const u64 tiles = (n + WS_TILE - 1) / WS_TILE;
            u32 R = 128;
            while (R > 1 && (u64)nsub * R > tiles / 8 + 64) R >>= 1;
            Tb.R = R; Tb.blocks_ub = (u32)((tiles + R - 1) / R) + nsub; Tb.rows = Tb.blocks_ub * R;
            Tb.blk_start = c.arena.get<u32>((size_t)nsub + 1);
            Tb.blk_seg = c.arena.get<u32>(Tb.blocks_ub);
            Tb.counts = c.arena.get<u32>((size_t)Tb.rows * D);
            Tb.bs = c.arena.get<u32>((size_t)Tb.blocks_ub * D);
            ws_sub_nblk_kernel<<<cdiv((size_t)nsub + 1, 256), 256, 0, s>>>(seg_begin, seg_end, nsub, R, Tb.blk_start);
            LAUNCH_CHECK();
            exclusive_sum_u32(c, Tb.blk_start, Tb.blk_start, (size_t)nsub + 1, nullptr);
            ws_blkseg_kernel<<<cdiv(nsub, 256), 256, 0, s>>>(Tb.blk_start, nsub, Tb.blk_seg);
            LAUNCH_CHECK();
        }
        u32* blk_super = c.arena.get<u32>(F0 + 1);
        ws_blk_super_kernel<<<cdiv(F0 + 1, 256), 256, 0, s>>>(Tb.blk_start, F0, nch, blk_super);
        LAUNCH_CHECK();
        WSLevel P;
        P.k1_in = pre->K1[0]; P.k2_in = KW == 2 ? pre->K2[0] : nullptr; P.v_in = pre->V[0];
        Vlast = last ? v_final : pre->V[1];
        P.k1_out = pre->K1[1]; P.k2_out = KW == 2 ? pre->K2[1] : nullptr; P.v_out = Vlast;
        P.digits = digits;
        P.counts = Tb.counts; P.blk_seg = Tb.blk_seg; P.blk_start = Tb.blk_start; P.seg_start = seg_begin; P.seg_end = seg_end; P.sub = nch;
        P.sp1 = sp1; P.sp2 = sp2;
        P.nseg = nsub; P.F = pl.F[l]; P.stride = stride; P.R = Tb.R; P.D = D;
        P.pre_digits = pre->dig2 ? pre->dig2_chunks : 0u;
        P.pre_counts = (pre->dig2 && c.wsort_prehist) ? pre->dig2_chunks : 0u;
        P.gen_off = 0; P.gen_len = n;
        const u32 rows = Tb.rows;
        P.per_xcd = (c.xcd_remap == 1 && rows >= 64) ? cdiv(rows, 8) : 0u;
        const u32 grid = P.per_xcd ? 8 * P.per_xcd : rows;
        if (P.pre_counts) {
            WPreHist H;
            for (u32 q = 0; q < 32; ++q) { H.counts[q] = pre->dig2_counts[q]; H.blk[q] = pre->dig2_blk[q]; H.R[q] = pre->dig2_R[q]; }
            H.nch = nch; H.chunks = P.pre_counts;
            ws_pre_hist_kernel<<<Tb.rows, 64, 0, s>>>(H, Tb.blk_start, Tb.blk_seg, nsub, Tb.R, D, Tb.rows, Tb.counts);
            LAUNCH_CHECK();
        }
        {
            const int pc = c.prof_begin(K_RS_COUNT, (u64)n * 8 * KW);
            // (the splitter search takes log2 FMAX steps whatever the fan-out: the 64-way levels get an instance of their own -- six steps instead of eight)
            if (last) { if (P.F > 256) ws_count_kernel<KW, false, true, 1024><<<grid, 256, 0, s>>>(P, g, rows); else if (P.F > 64) ws_count_kernel<KW, false, true, 256><<<grid, 256, 0, s>>>(P, g, rows); else ws_count_kernel<KW, false, true, 64><<<grid, 256, 0, s>>>(P, g, rows); }
            else { if (P.F > 256) ws_count_kernel<KW, false, false, 1024><<<grid, 256, 0, s>>>(P, g, rows); else if (P.F > 64) ws_count_kernel<KW, false, false, 256><<<grid, 256, 0, s>>>(P, g, rows); else ws_count_kernel<KW, false, false, 64><<<grid, 256, 0, s>>>(P, g, rows); }
            LAUNCH_CHECK();
            c.prof_end(pc);
        }
        ss_level_offsets_sub(c, Tb, nsub, blk_super, out_start, F0, D, nstart, n);
        {
            const int ps = c.prof_begin(K_RS_SCATTER_U64, (u64)n * (2 * (4 + 8 * KW) + 2));
            if (last) { if (P.F > 256) ws_scatter_kernel<KW, false, true, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_scatter_kernel<KW, false, true, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            else { if (P.F > 256) ws_scatter_kernel<KW, false, false, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_scatter_kernel<KW, false, false, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            LAUNCH_CHECK();
            c.prof_end(ps);
        }
        c.arena.release(lm2);
        seg_start = nstart;
        nseg = F0 * D;
        cur = 1;
        K1 = const_cast<u64**>(pre->K1); K2 = const_cast<u64**>(pre->K2); V = const_cast<u32**>(pre->V);
    }
    for (int l = pre ? 2 : 0; l < L; ++l) {
        const bool last = (l == L - 1);
        const bool gl = GEN && l == 0;
        const u32 D = last ? 2 * pl.F[l] : pl.F[l];
        u32 stride = 1;
        for (int q = l + 1; q < L; ++q) stride *= pl.F[q];
        u32* nstart = c.arena.get<u32>((size_t)nseg * D + 1);
        const size_t lm2 = c.arena.mark();
        SegTables Tb;
        ss_level_tables(c, seg_start, nseg, n, D, Tb);
        WSLevel P;
        P.k1_in = cur >= 0 ? K1[cur] : nullptr; P.k2_in = (cur >= 0 && KW == 2) ? K2[cur] : nullptr;
        P.v_in = pre ? Vlast : (cur >= 0 ? V[cur] : nullptr);
        const int nxt = cur < 0 ? 0 : (cur ^ 1);
        P.k1_out = K1[nxt]; P.k2_out = KW == 2 ? K2[nxt] : nullptr;
        P.v_out = (pre && last) ? v_final : V[nxt];
        Vlast = P.v_out;
        P.digits = digits;
        P.counts = Tb.counts; P.blk_seg = Tb.blk_seg; P.blk_start = Tb.blk_start; P.seg_start = seg_start; P.sp1 = sp1; P.sp2 = sp2;
        P.seg_end = seg_start + 1; P.sub = 1;
        P.nseg = nseg; P.F = pl.F[l]; P.stride = stride; P.R = Tb.R; P.D = D;
        P.gen_off = 0; P.gen_len = n;
        const u32 rows = Tb.rows;
        P.per_xcd = (c.xcd_remap == 1 && rows >= 64) ? cdiv(rows, 8) : 0u;
        const u32 grid = P.per_xcd ? 8 * P.per_xcd : rows;
        {
            const int pc = c.prof_begin(K_RS_COUNT, (u64)n * (gl ? 1 : 8 * KW));
            if (gl && last) { if (P.F > 256) ws_count_kernel<KW, GEN, true, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_count_kernel<KW, GEN, true, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            else if (gl) { if (P.F > 256) ws_count_kernel<KW, GEN, false, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_count_kernel<KW, GEN, false, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            else if (last) { if (P.F > 256) ws_count_kernel<KW, false, true, 1024><<<grid, 256, 0, s>>>(P, g, rows); else if (P.F > 64) ws_count_kernel<KW, false, true, 256><<<grid, 256, 0, s>>>(P, g, rows); else ws_count_kernel<KW, false, true, 64><<<grid, 256, 0, s>>>(P, g, rows); }
            else { if (P.F > 256) ws_count_kernel<KW, false, false, 1024><<<grid, 256, 0, s>>>(P, g, rows); else if (P.F > 64) ws_count_kernel<KW, false, false, 256><<<grid, 256, 0, s>>>(P, g, rows); else ws_count_kernel<KW, false, false, 64><<<grid, 256, 0, s>>>(P, g, rows); }
            LAUNCH_CHECK();
            c.prof_end(pc);
        }
        ss_level_offsets(c, Tb, seg_start, nseg, D, nstart, n);
        {
            const int ps = c.prof_begin(K_RS_SCATTER_U64, (u64)n * (gl ? 3 + 4 + 8 * KW : 2 * (4 + 8 * KW) + 2));
            if (gl && last) { if (P.F > 256) ws_scatter_kernel<KW, GEN, true, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_scatter_kernel<KW, GEN, true, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            else if (gl) { if (P.F > 256) ws_scatter_kernel<KW, GEN, false, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_scatter_kernel<KW, GEN, false, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            else if (last) { if (P.F > 256) ws_scatter_kernel<KW, false, true, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_scatter_kernel<KW, false, true, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            else { if (P.F > 256) ws_scatter_kernel<KW, false, false, 1024><<<grid, 256, 0, s>>>(P, g, rows); else ws_scatter_kernel<KW, false, false, 256><<<grid, 256, 0, s>>>(P, g, rows); }
            LAUNCH_CHECK();
            c.prof_end(ps);
        }
        c.arena.release(lm2);
        seg_start = nstart;
        nseg = nseg * D;
        cur = nxt;
    }
    const u32 nleaf = nseg;
    const u32* leaf_start = seg_start;

    // ---- units + leaf sort ----
    UnitTables U;
    U.wide_classes = 1;                                        // the six size classes of prim.hpp wide_class
    ss_build_units(c, leaf_start, nleaf, U, (u32)c.wsort_pack);   // small units: the 12- and 16-row leaf kernels run out of registers
    const u32 nlarge = U.hc[0];
    st->units = U.hc[1]; st->large_leaves = nlarge;
    WLeaf A;
    A.k1 = K1[cur]; A.k2 = KW == 2 ? K2[cur] : nullptr; A.v = Vlast; A.unit_rng = U.unit_rng; A.flags = flags; A.lcp = lcp8;
    A.d_err = c.d_err; A.inv = g.b ? (65536u + (u32)g.b - 1) / (u32)g.b : 65536u;
    A.cmax = c.wsort_small ? 1u : (c.wsort_cmax < 1 ? 1u : (c.wsort_cmax > (int)WS_CMAX ? WS_CMAX : (u32)c.wsort_cmax));
    {
        // Stage 0: the units of the leaves.  Kernel A sorts every unit by its first differing word; the counting kernel orders the runs
        // that are left (<= 256 members) by the next word and hands longer runs back as the units of the next stage.
        const u32* cur_rng = U.unit_rng;
        const u32* cur_cls = U.cls_list;
        size_t cur_cap = U.cap;
        constexpr int NB = WIDE_NCLS;
        u32 cur_cnt[NB];
        for (int q = 0; q < NB; ++q) cur_cnt[q] = U.whc[q];
        u32 wave_cnt[4] = { 0, 0, 0, 0 };                        // runs for the lane / wave kernels (classes 0 .. 3 of the previous stage's hand-over)
        u64 wave_recs = 0;                                       // ... and the records in them
        const u32* wave_list = nullptr;
        const u32 ediv = A.cmax + 1;                            // a run that is handed on has more than cmax members
        const u32 ecap2 = (u32)(n / ediv + 2);
        u32* e_rng[2] = { c.arena.get<u32>(2 * (size_t)ecap2), c.arena.get<u32>(2 * (size_t)ecap2) };
        u32* e_cls[2] = { c.arena.get<u32>(EC_NCLS * (size_t)ecap2), c.arena.get<u32>(EC_NCLS * (size_t)ecap2) };
        u32* lc = c.arena.get<u32>(64);                          // [0 .. 2 NB): per class |rlist|, |tlist|; [2 NB ..): next stage's runs / units per class
        for (int stage = 0; stage < 8; ++stage) {
            if (wave_cnt[0] | wave_cnt[1] | wave_cnt[2] | wave_cnt[3]) {
                Ctx::ProfScope prof(c, K_WS_RUN, wave_recs * 18);          // per record: second word + position in (12 B), position + flag + LCP out (6 B)
                A.unit_rng = cur_rng;
                const u32* l0 = wave_list, *l1 = wave_list + cur_cap, *l2 = wave_list + 2 * cur_cap, *l3 = wave_list + 3 * cur_cap;
                // The four kernels work on disjoint runs and each is a chain of dependent loads per wave with little work behind it (none of
                // them fills the device): with the context's two other streams at hand they run side by side (option wsort_run_streams).
                const bool fork = c.wsort_run_streams && c.copy_stream && c.aux_stream && (wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]) >= 4096;
                hipStream_t sl = fork ? c.copy_stream : s, sw = fork ? c.aux_stream : s;
                if (fork) {
                    HIP_TRY(hipEventRecord(c.ev_copy[10], s));
                    HIP_TRY(hipStreamWaitEvent(sl, c.ev_copy[10], 0));
                    HIP_TRY(hipStreamWaitEvent(sw, c.ev_copy[10], 0));
                }
                if (wave_cnt[3]) { ws_run_wave_kernel<1024, PAIRS><<<cdiv(wave_cnt[3], 4), 256, 0, s>>>(A, l3, wave_cnt[3]); LAUNCH_CHECK(); }
                if (wave_cnt[2]) { ws_run_wave_kernel<256, PAIRS><<<cdiv(wave_cnt[2], 4), 256, 0, sw>>>(A, l2, wave_cnt[2]); LAUNCH_CHECK(); }
                if (wave_cnt[0]) { ws_run_lane_kernel<32, PAIRS><<<cdiv(wave_cnt[0], 8), 256, 0, sl>>>(A, l0, wave_cnt[0]); LAUNCH_CHECK(); }
                if (wave_cnt[1]) { ws_run_lane_kernel<64, PAIRS><<<cdiv(wave_cnt[1], 4), 256, 0, sl>>>(A, l1, wave_cnt[1]); LAUNCH_CHECK(); }
                if (fork) {
                    HIP_TRY(hipEventRecord(c.ev_copy[11], sl));
                    HIP_TRY(hipEventRecord(c.ev_copy[12], sw));
                    HIP_TRY(hipStreamWaitEvent(s, c.ev_copy[11], 0));
                    HIP_TRY(hipStreamWaitEvent(s, c.ev_copy[12], 0));
                }
                st->wave_runs += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
                wave_cnt[0] = wave_cnt[1] = wave_cnt[2] = wave_cnt[3] = 0;
            }
            { u32 any = 0; for (int q = 0; q < NB; ++q) any |= cur_cnt[q]; if (!any) break; }
            if (stage == 7) throw HipError{hipErrorUnknown, "wide splitter sort: leaf stages did not converge", (int)__LINE__};
            const size_t lm3 = c.arena.mark();
            u32* rl = c.arena.get<u32>(NB * cur_cap);
            u32* tl = c.arena.get<u32>(NB * cur_cap);
            HIP_TRY(hipMemsetAsync(lc, 0, 64 * sizeof(u32), s));
            A.unit_rng = cur_rng;
            // (stage 0 sorts every record once: keys + position in, position + flag + LCP out; the later stages re-sort the long runs)
            HIP_TRY(hipMemsetAsync(e_rng[stage & 1], 0, 2 * (size_t)ecap2 * sizeof(u32), s));
            const WEmit E = { e_rng[stage & 1], ediv };
            const WEmit EA = { c.wsort_fuse ? E.rng : nullptr, ediv };   // kernel A orders the short runs of its units itself
            const int pa = c.prof_begin(K_WS_LEAF_SORT, stage == 0 ? (u64)n * (4 + 8 * KW + 6) : 0);
            for (int q = 0; q < NB; ++q) {
                const u32 cnt = cur_cnt[q];
                if (!cnt) continue;
                const WLists Q = { rl + q * cur_cap, tl + q * cur_cap, lc + 2 * q };
                const u32* lst = cur_cls + q * cur_cap;
                // (units of <= 2048 records: four waves -- a smaller workgroup, more units in flight per CU; every class has just the rows
                //  its units need: prim.hpp wide_class -- 4 .. 8 rows of four waves, 5 .. 8 rows of eight, 16 rows of eight)
                switch (q) {
                case 0: ws_leaf_sort_kernel<KW, 4, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 1: ws_leaf_sort_kernel<KW, 5, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 2: ws_leaf_sort_kernel<KW, 6, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 3: ws_leaf_sort_kernel<KW, 7, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 4: ws_leaf_sort_kernel<KW, 8, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 5: ws_leaf_sort_kernel<KW, 5, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 6: ws_leaf_sort_kernel<KW, 6, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 7: ws_leaf_sort_kernel<KW, 7, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                case 8: ws_leaf_sort_kernel<KW, 8, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                default: ws_leaf_sort_kernel<KW, 16, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, Q, EA); break;
                }
                LAUNCH_CHECK();
            }
            c.prof_end(pa);
            u32 hl[2 * NB];
            c.read_n(lc, hl, 2 * NB);
            const int pb = c.prof_begin(K_WS_LEAF_COUNT, stage == 0 ? (u64)n * (8 + 4 + 1 + 4 + 2) : 0);
            const WEmit noE = { nullptr, 1 };
            auto count_pass = [&](int q, const u32* lst, u32 cnt, u32 mask, u32 val, const WCount& R, const WEmit& Em) {
                if (!cnt) return;
                switch (q) {
                case 0: ws_leaf_count_kernel<4, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 1: ws_leaf_count_kernel<5, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 2: ws_leaf_count_kernel<6, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 3: ws_leaf_count_kernel<7, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 4: ws_leaf_count_kernel<8, 4, PAIRS><<<cnt, 4 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 5: ws_leaf_count_kernel<5, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 6: ws_leaf_count_kernel<6, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 7: ws_leaf_count_kernel<7, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                case 8: ws_leaf_count_kernel<8, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                default: ws_leaf_count_kernel<16, 8, PAIRS><<<cnt, 8 * 64, 0, s>>>(A, lst, cnt, mask, val, R, Em); break;
                }
                LAUNCH_CHECK();
            };
            for (int q = 0; q < NB; ++q) {
                const u32 nt = hl[2 * q + 1], nr = hl[2 * q];
                if (nt) {                                        // ties on a truncated word: by the word itself ...
                    st->trunc_units += nt;
                    const WCount t1 = { A.k1, KW == 2 ? A.k2 : nullptr, 0u };
                    count_pass(q, tl + q * cur_cap, nt, 0x80000000u, 0u, t1, E);              // not pure: k1 (k2 follows)
                    if (KW == 2) {
                        const WCount t2 = { A.k2, nullptr, 64u };
                        count_pass(q, tl + q * cur_cap, nt, 0x80000000u, 0x80000000u, t2, E);  // pure: k2
                        count_pass(q, tl + q * cur_cap, nt, 0x80000000u, 0u, t2, noE);         // ... then the not-pure ones by k2 (their long
                    }                                                                           //     runs are on their way already)
                }
                if (KW == 2 && nr) {
                    st->refined_units += nr;
                    const WCount r2 = { A.k2, nullptr, 64u };
                    count_pass(q, rl + q * cur_cap, nr, 0u, 0u, r2, E);
                }
            }
            c.prof_end(pb);
            ws_emit_compact_kernel<<<cdiv(ecap2, EC_TILE), 256, 0, s>>>(e_rng[stage & 1], ecap2, e_cls[stage & 1], ecap2, lc + 2 * NB);
            LAUNCH_CHECK();
            u32 he[2 * EC_NCLS];                                 // runs / units per class, then their records per class
            c.read_n(lc + 2 * NB, he, 2 * EC_NCLS);
            if (c.wsort_log) {
                fprintf(stderr, "[wsort] n=%zu stage %d: units", n, stage);
                for (int q = 0; q < NB; ++q) fprintf(stderr, " %u", cur_cnt[q]);
                fprintf(stderr, " | r/t lists");
                for (int q = 0; q < NB; ++q) fprintf(stderr, " %u/%u", hl[2 * q], hl[2 * q + 1]);
                fprintf(stderr, " | handed on: lane %u %u wave %u %u block", he[0], he[1], he[2], he[3]);
                for (int q = 0; q < NB; ++q) fprintf(stderr, " %u", he[4 + q]);
                fprintf(stderr, "\n");
            }
            c.arena.release(lm3);
            for (int q = 0; q < EC_NCLS; ++q) st->longrun_units += he[q];
            cur_rng = e_rng[stage & 1]; cur_cls = e_cls[stage & 1] + 4 * (size_t)ecap2; cur_cap = ecap2;
            wave_list = e_cls[stage & 1];
            for (int q = 0; q < 4; ++q) wave_cnt[q] = he[q];
            wave_recs = (u64)he[EC_NCLS] + he[EC_NCLS + 1] + he[EC_NCLS + 2] + he[EC_NCLS + 3];
            for (int q = 0; q < NB; ++q) cur_cnt[q] = he[4 + q];
            st->leaf_stages = (u32)stage + 1;
        }
    }
    if (nlarge) {                                              // leaves above the workgroup capacity: LSD sort, one by one
        if (nlarge > U.large_cap) throw HipError{hipErrorUnknown, "wide splitter sort: too many oversized leaves", (int)__LINE__};
        std::vector<u32> ll(nlarge), ls((size_t)nleaf + 1);
        HIP_TRY(hipMemcpyAsync(ll.data(), U.large + 6, nlarge * sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ls.data(), leaf_start, ((size_t)nleaf + 1) * sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (u32 i = 0; i < nlarge; ++i) {
            const size_t a = ls[ll[i]], b = ls[ll[i] + 1];
            u64* kk1[2] = { K1[cur] + a, K1[cur ^ 1] + a };
            u64* kk2[2] = { KW == 2 ? K2[cur] + a : nullptr, KW == 2 ? K2[cur ^ 1] + a : nullptr };
            u32* vv[2] = { Vlast + a, V[cur ^ 1] + a };
            ws_lsd_sort_wide(c, kk1, KW == 2 ? kk2 : nullptr, vv, b - a, 64);
            HIP_TRY(hipMemcpyAsync(kk1[0], kk1[1], (b - a) * sizeof(u64), hipMemcpyDeviceToDevice, s));
            if (KW == 2) HIP_TRY(hipMemcpyAsync(kk2[0], kk2[1], (b - a) * sizeof(u64), hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(vv[0], vv[1], (b - a) * sizeof(u32), hipMemcpyDeviceToDevice, s));
            if (!PAIRS) {
                ws_range_flags_kernel<KW><<<cdiv(b - a, 256), 256, 0, s>>>(kk1[0], kk2[0], 0u, (u32)(b - a), A.inv, flags + a, lcp8 + a);
                LAUNCH_CHECK();
            }
            st->large_pairs += b - a;
        }
    }
    if (!PAIRS) {
        ws_fix_kernel<KW><<<cdiv(nleaf, 256), 256, 0, s>>>(leaf_start, nleaf, Vlast, g, flags, lcp8);
        LAUNCH_CHECK();
        if (c.wsort_count_nonheads) {                           // (the suffix array counts the unresolved slots itself: one pass over the flags less)
            { unsigned gq = cdiv(n, 256 * 16); if (gq > 4096) gq = 4096; ws_count_flags_kernel<<<gq, 256, 0, s>>>(flags, n, d_nonheads); }
            LAUNCH_CHECK();
            u64 nh = 0;
            c.read_n((const u64*)d_nonheads, &nh, 1);
            st->nonheads = nh;
        }
    }
#ifdef TDC_WL_PROF
    if (!PAIRS) {
        unsigned long long hp[16];
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpyFromSymbol(hp, HIP_SYMBOL(wl_prof), sizeof(hp)));
        fprintf(stderr, "[wl_prof] wgs %llu records %llu passes %llu | ticks: load %llu lsd %llu heads %llu pos %llu k2 %llu count %llu fin %llu nofuse %llu\n", hp[15], hp[14], hp[13],
                hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7]);
        memset(hp, 0, sizeof(hp));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(wl_prof), hp, sizeof(hp)));
    }
#endif
    c.arena.release(mark);
    return cur;
}
}  // namespace

int wsort_suffixes(Ctx& c, int KW, const WKeyGen& g, u64* K1[2], u64* K2[2], u32* V[2], size_t n, u8* flags, u8* lcp8, WSortStats* st) {
    WSortStats local;
    if (!st) st = &local;
    *st = WSortStats();
    if (KW == 2) return ws_sort_impl<2, true, false>(c, &g, K1, K2, V, n, 64, flags, lcp8, st);
    return ws_sort_impl<1, true, false>(c, &g, K1, K2, V, n, 64, flags, lcp8, st);
}

void wsort_suffixes_pre(Ctx& c, const WPre& P, u32* v_final, u8* flags, u8* lcp8, WSortStats* st) {
    WSortStats local;
    if (!st) st = &local;
    *st = WSortStats();
    u64* K1[2] = { P.K1[0], P.K1[1] }; u64* K2[2] = { P.K2[0], P.K2[1] }; u32* V[2] = { P.V[0], P.V[1] };
    if (P.KW == 2) (void)ws_sort_impl<2, true, false>(c, &P.g, K1, K2, V, P.n, 64, flags, lcp8, st, &P, v_final);
    else (void)ws_sort_impl<1, true, false>(c, &P.g, K1, K2, V, P.n, 64, flags, lcp8, st, &P, v_final);
}

bool wsort_pre_begin(Ctx& c, WPre& P, const u8* text, size_t n, const size_t* chunk_off, u32 nchunks, const u32* hist0) {
    P = WPre();
    if (!c.wsort_overlap || !wsort_applicable(c, n) || nchunks < 2 || nchunks > 32 || chunk_off[1] < ((size_t)1 << 22)) return false;
    for (u32 q = 0; q < nchunks; ++q) if (chunk_off[q] % WS_TILE != 0 || chunk_off[q] > chunk_off[q + 1]) return false;
    if (chunk_off[0] != 0 || chunk_off[nchunks] != n) return false;
    // provisional code map: the sentinel and the byte values of chunk 0, dense and in byte order
    u8 code[256];
    u32 sigma = 0;
    for (int i = 0; i < 256; ++i) {
        code[i] = (u8)sigma;
        if (i == 0 || hist0[i]) { ++sigma; P.present[i >> 5] |= 1u << (i & 31); }
    }
    P.text = text; P.n = n; P.nchunks = nchunks;
    for (u32 q = 0; q <= nchunks; ++q) P.chunk_off[q] = chunk_off[q];
    wsort_make_keygen(c, text, n, sigma, code, P.KW, P.g);
    ss_fanouts(c, n, P.L, P.F, P.os, (u32)c.wsort_leaf, c.wsort_two);
    if (c.wsort_order == 1 && P.L == 3) std::swap(P.F[0], P.F[2]);
    if (P.L < 2) return false;
    P.NLr = P.F[0] * P.F[1] * P.F[2]; P.NS = P.NLr - 1; P.S = P.os * P.NLr;
    Arena& A = c.arena;
    P.K1[0] = A.get_top<u64>(n); P.K1[1] = A.get_top<u64>(n);
    if (P.KW == 2) { P.K2[0] = A.get_top<u64>(n); P.K2[1] = A.get_top<u64>(n); }
    P.V[0] = A.get_top<u32>(n); P.V[1] = A.get_top<u32>(n);
    P.digits = A.get_top<u16>(align_up(n, WS_TILE) + WS_TILE);
    P.sp1 = A.get_top<u64>((size_t)P.NS + 1);
    P.sp2 = P.KW == 2 ? A.get_top<u64>((size_t)P.NS + 1) : nullptr;
    P.nstart_all = A.get_top<u32>((size_t)nchunks * (P.F[0] + 1));
    P.dig2 = c.wsort_predig != 0 && c.aux_stream != nullptr;
    // (the last chunks keep the search in the merging level: their digits would only be ready after the last copy)
    P.dig2_chunks = P.dig2 ? (nchunks > (u32)c.wsort_predig_skip ? nchunks - (u32)c.wsort_predig_skip : 0u) : 0u;
    // splitters from a sample of chunk 0 (a key reads up to 64 bytes ahead: chunk 1 need not be there)
    WPlan pl;
    pl.L = P.L; pl.F[0] = P.F[0]; pl.F[1] = P.F[1]; pl.F[2] = P.F[2]; pl.os = P.os; pl.NLr = P.NLr; pl.NS = P.NS; pl.S = P.S;
    const size_t n_sample = chunk_off[1] - 64;
    if (P.KW == 2) ws_splitters<2, true>(c, pl, nullptr, nullptr, P.g, n_sample, P.sp1, P.sp2);
    else ws_splitters<1, true>(c, pl, nullptr, nullptr, P.g, n_sample, P.sp1, nullptr);
    P.begun = true;
    return true;
}

void wsort_pre_chunk(Ctx& c, WPre& P, u32 q) {
    if (!P.begun) return;
    hipStream_t s = c.stream;
    const size_t off = P.chunk_off[q];
    if (off >= P.n) {                                          // (more chunks than text: an empty part)
        ws_set2_kernel<<<1, 1, 0, s>>>(P.nstart_all + (size_t)q * (P.F[0] + 1), (u32)P.n, (u32)P.n);
        return;
    }
    const size_t len = P.chunk_off[q + 1] - off;
    const size_t mark = c.arena.mark();
    u32* seg2 = c.arena.get<u32>(2);
    ws_set2_kernel<<<1, 1, 0, s>>>(seg2, (u32)off, (u32)(off + len));
    LAUNCH_CHECK();
    const u32 D = P.F[0];
    SegTables Tb;
    ss_level_tables(c, seg2, 1, len, D, Tb);
    WSLevel Lv;
    Lv.k1_in = nullptr; Lv.k2_in = nullptr; Lv.v_in = nullptr;
    Lv.k1_out = P.K1[0]; Lv.k2_out = P.KW == 2 ? P.K2[0] : nullptr; Lv.v_out = P.V[0];
    Lv.digits = P.digits;
    Lv.counts = Tb.counts; Lv.blk_seg = Tb.blk_seg; Lv.blk_start = Tb.blk_start; Lv.seg_start = seg2; Lv.seg_end = seg2 + 1; Lv.sub = 1;
    Lv.sp1 = P.sp1; Lv.sp2 = P.sp2;
    Lv.nseg = 1; Lv.F = P.F[0]; Lv.stride = P.F[1] * P.F[2]; Lv.R = Tb.R; Lv.D = D;
    Lv.gen_off = 0; Lv.gen_len = P.n;
    const u32 rows = Tb.rows;
    Lv.per_xcd = (c.xcd_remap == 1 && rows >= 64) ? cdiv(rows, 8) : 0u;
    const u32 grid = Lv.per_xcd ? 8 * Lv.per_xcd : rows;
    {
        const int pc = c.prof_begin(K_RS_COUNT, (u64)len);
        if (P.KW == 2) { if (Lv.F > 256) ws_count_kernel<2, true, false, 1024><<<grid, 256, 0, s>>>(Lv, P.g, rows); else ws_count_kernel<2, true, false, 256><<<grid, 256, 0, s>>>(Lv, P.g, rows); }
        else { if (Lv.F > 256) ws_count_kernel<1, true, false, 1024><<<grid, 256, 0, s>>>(Lv, P.g, rows); else ws_count_kernel<1, true, false, 256><<<grid, 256, 0, s>>>(Lv, P.g, rows); }
        LAUNCH_CHECK();
        c.prof_end(pc);
    }
    ss_level_offsets(c, Tb, seg2, 1, D, P.nstart_all + (size_t)q * (P.F[0] + 1), off + len);
    {
        const int ps = c.prof_begin(K_RS_SCATTER_U64, (u64)len * (3 + 4 + 8 * P.KW));
        if (P.KW == 2) { if (Lv.F > 256) ws_scatter_kernel<2, true, false, 1024><<<grid, 256, 0, s>>>(Lv, P.g, rows); else ws_scatter_kernel<2, true, false, 256><<<grid, 256, 0, s>>>(Lv, P.g, rows); }
        else { if (Lv.F > 256) ws_scatter_kernel<1, true, false, 1024><<<grid, 256, 0, s>>>(Lv, P.g, rows); else ws_scatter_kernel<1, true, false, 256><<<grid, 256, 0, s>>>(Lv, P.g, rows); }
        LAUNCH_CHECK();
        c.prof_end(ps);
    }
    c.arena.release(mark);
    if (P.dig2 && q < P.dig2_chunks) {
        // (on the context's low-priority side stream, behind this chunk's level 1: the chain copy -> level 1 -> next copy is not lengthened,
        //  the digits fill the device's idle time behind the upload.  Its tables are allocated where this chunk's level-1 scratch was and
        //  stay: the next chunk's scratch lies above them, the side stream only touches them once level 1 is through -- the event)
        HIP_TRY(hipEventRecord(c.ev_aux[0], c.stream));
        HIP_TRY(hipStreamWaitEvent(c.aux_stream, c.ev_aux[0], 0));
        struct Swap { Ctx& c; hipStream_t saved; ~Swap() { c.stream = saved; } } swap{c, c.stream};
        c.stream = c.aux_stream;
        hipStream_t s = c.stream;
        // ---- round 6 (round 4's variant re-measured now that a chunk's level 1 takes 1.3 instead of 1.7 ms of its 2.2 ms copy): the
        //      level-2 DIGITS of this chunk's records behind the upload as well -- the chunk's F0 buckets are the segments, every bucket is
        //      searched against its own splitters; the merging level then counts from the digits (2 bytes per record instead of the 16
        //      bytes of keys and a splitter search) ----
        const u32* seg_start = P.nstart_all + (size_t)q * (P.F[0] + 1);
        const bool last2 = (P.L == 2);
        const u32 nseg = P.F[0], D2 = last2 ? 2 * P.F[1] : P.F[1];
        SegTables T2;
        ss_level_tables(c, seg_start, nseg, len, D2, T2);
        P.dig2_counts[q] = T2.counts; P.dig2_blk[q] = T2.blk_start; P.dig2_R[q] = T2.R;
        WSLevel L2;
        L2.k1_in = P.K1[0]; L2.k2_in = P.KW == 2 ? P.K2[0] : nullptr; L2.v_in = P.V[0];
        L2.k1_out = nullptr; L2.k2_out = nullptr; L2.v_out = nullptr;
        L2.digits = P.digits;
        L2.counts = T2.counts; L2.blk_seg = T2.blk_seg; L2.blk_start = T2.blk_start; L2.seg_start = seg_start; L2.seg_end = seg_start + 1; L2.sub = 1;
        L2.sp1 = P.sp1; L2.sp2 = P.sp2;
        L2.nseg = nseg; L2.F = P.F[1]; L2.stride = last2 ? 1u : P.F[2]; L2.R = T2.R; L2.D = D2;
        L2.gen_off = 0; L2.gen_len = P.n;
        const u32 rows2 = T2.rows;
        L2.per_xcd = (c.xcd_remap == 1 && rows2 >= 64) ? cdiv(rows2, 8) : 0u;
        const u32 grid2 = L2.per_xcd ? 8 * L2.per_xcd : rows2;
        const int pc = c.prof_begin(K_RS_COUNT, (u64)len * 8 * P.KW);
        auto launch = [&](auto kw, auto lastc) {
            constexpr int KWc = decltype(kw)::value; constexpr bool LASTc = decltype(lastc)::value;
            if (L2.F > 256) ws_count_kernel<KWc, false, LASTc, 1024><<<grid2, 256, 0, s>>>(L2, P.g, rows2);
            else if (L2.F > 64) ws_count_kernel<KWc, false, LASTc, 256><<<grid2, 256, 0, s>>>(L2, P.g, rows2);
            else ws_count_kernel<KWc, false, LASTc, 64><<<grid2, 256, 0, s>>>(L2, P.g, rows2);
        };
        if (P.KW == 2) { if (last2) launch(std::integral_constant<int, 2>{}, std::true_type{}); else launch(std::integral_constant<int, 2>{}, std::false_type{}); }
        else { if (last2) launch(std::integral_constant<int, 1>{}, std::true_type{}); else launch(std::integral_constant<int, 1>{}, std::false_type{}); }
        LAUNCH_CHECK();
        c.prof_end(pc);
        HIP_TRY(hipEventRecord(c.ev_aux[1], c.aux_stream));
    }
}

void wsort_pre_finish(Ctx& c, WPre& P, const u32* hist_full) {
    P.active = false;
    if (!P.begun) return;
    if (P.dig2) HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_aux[1], 0));     // (the level-2 digits of the last chunks)
    for (int i = 1; i < 256; ++i) {
        const bool in_map = (P.present[i >> 5] >> (i & 31)) & 1u;
        if ((hist_full[i] != 0) != in_map) return;            // a byte value chunk 0 did not show (or the reverse): the keys are worthless
    }
    P.active = true;
}

// ---- records whose first word is already in order: every run of equal first words sorted by the second (text rounds) -----------------
// The runs of a text round are the groups of still-equal suffixes: 2.5 members on average on English text, so most of them are ordered
// by their members themselves -- a tile of records in LDS, every record counts the members of its run that go before it (runs of up to
// SG_MAXG members; a run that crosses the tile's end is seen whole through the halo, and belongs to the tile it starts in).  Longer runs
// (up to WS_WAVE_MAX) are entered into the table the run kernels' lists are compiled from (entry start / (SG_MAXG + 1): such a run has
// more members than that, so two never share an entry); beyond that the caller sorts the list as a whole.
#ifndef TDC_SG_MAXG
#define TDC_SG_MAXG 16
#endif
#ifndef TDC_SG_T
#define TDC_SG_T 1024
#endif
constexpr u32 SG_T = TDC_SG_T, SG_H = TDC_SG_MAXG, SG_MAXG = TDC_SG_MAXG, SG_BIG_CAP = 1u << 16;
constexpr u32 SG_EDIV = SG_MAXG + 1;      // a run that is entered into the table has more than SG_MAXG members: entry = start / SG_EDIV, no two runs share one
struct SegCounters { u32 overflow, nbig, big_recs, pad; };
__global__ __launch_bounds__(256) void ws_seg_tile_kernel(const u64* __restrict__ k1, u64* k2, u32* v, u32 m, u32* __restrict__ rng, u32* __restrict__ big,
                                                          u32 big_cap, SegCounters* __restrict__ sc) {
    __shared__ u64 sk[SG_T + SG_H];
    __shared__ u32 sv[SG_T + SG_H];
    __shared__ u8 sh[SG_T + SG_H];           // record t0 + j starts a run (records behind the list count as starts)
    __shared__ u8 sg[SG_T];                  // members of the run that starts at j, if this tile orders it (else 0)
    const u32 t0 = blockIdx.x * SG_T;
    for (u32 j = threadIdx.x; j < SG_T + SG_H; j += 256) {
        const u32 i = t0 + j;
        bool head = true; u64 b = 0; u32 x = 0;
        if (i < m) { const u64 a = k1[i]; head = (i == 0) || (k1[i - 1] != a); b = k2[i]; x = v[i]; }
        sk[j] = b; sv[j] = x; sh[j] = head ? 1 : 0;
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < SG_T; j += 256) {
        u32 size = 0;
        if (t0 + j < m && sh[j]) {
            u32 g = 1;
            while (g <= SG_MAXG && !sh[j + g]) ++g;            // (j + g <= j + SG_MAXG < SG_T + SG_H)
            if (g <= SG_MAXG) { if (g >= 2) size = g; }
            else {                                              // a longer run: its end from the list itself
                // (the first words are in order: the end of the run by galloping + bisection -- a dozen dependent loads, not one per member)
                const u32 i = t0 + j;
                const u64 k = k1[i];
                u32 lo = i + g, hi = m;                          // records i .. lo - 1 belong to the run; the end lies in [lo, hi]
                for (u32 step = 16; lo < hi; step *= 2) {
                    const u32 p = (hi - lo > step) ? lo + step - 1 : hi - 1;
                    if (k1[p] == k) lo = p + 1; else { hi = p; break; }
                }
                while (lo < hi) { const u32 mid = lo + (hi - lo) / 2; if (k1[mid] == k) lo = mid + 1; else hi = mid; }
                const u32 e = lo;
                if (e - i <= WS_WAVE_MAX) { rng[2 * (size_t)(i / SG_EDIV)] = i; rng[2 * (size_t)(i / SG_EDIV) + 1] = e | 0x80000000u; }     // bit 31: the records tie on all of k1
                else {                                          // beyond what a wave orders
                    const u32 slot = atomicAdd(&sc->nbig, 1u);
                    if (slot < big_cap) { big[2 * slot] = i; big[2 * slot + 1] = e; atomicAdd(&sc->big_recs, e - i); }
                    else sc->overflow = 1u;
                }
            }
        }
        sg[j] = (u8)size;
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < SG_T + SG_H; j += 256) {
        if (t0 + j >= m) continue;
        u32 st = j, steps = 0;
        while (!sh[st] && st > 0 && steps < SG_MAXG) { --st; ++steps; }
        if (!sh[st] || st >= SG_T) continue;                  // the run started in front of the tile (or too far back), or starts in the halo
        const u32 g = sg[st];
        if (g < 2 || j - st >= g) continue;
        const u64 kj = sk[j];
        u32 r = 0;
        for (u32 l = st; l < st + g; ++l) { const u64 kl = sk[l]; r += (kl < kj || (kl == kj && l < j)) ? 1u : 0u; }
        k2[(size_t)t0 + st + r] = kj;
        v[(size_t)t0 + st + r] = sv[j];
    }
}
// The same with the records MADE here (text rounds: record i = (slot of the group head r1[i], the next g.s symbols of suffix sa[i] from text
// position sa[i] + h, sa[i])): the text gathers of a round's key pass run inside this kernel, whose LDS work hides behind them, and the
// list is written once instead of written, read and written.  Every record has to be written by exactly one tile: a run of up to SG_MAXG
// members by the tile it starts in (all of it, sorted: it reaches at most SG_H - 1 records into the right halo), every other record --
// members of longer runs -- in place by the tile it lies in.  A halo on the left as well tells a tile whether a run that reaches into
// it from the tile before is one of the short ones (then that tile writes it).
__global__ __launch_bounds__(256) void ws_seg_tile_keys_kernel(const u32* __restrict__ a_sa, const u32* __restrict__ a_r1, u32 m, WKeyGen g, u32 h,
                                                               u64* __restrict__ k1, u64* __restrict__ k2, u32* __restrict__ v,
                                                               u32* __restrict__ rng, u32* __restrict__ big, u32 big_cap, SegCounters* __restrict__ sc) {
    constexpr u32 LN = SG_T + 2 * SG_H;
    __shared__ u64 sk[LN];
    __shared__ u32 sv[LN];
    __shared__ u8 sh[LN];                    // record (t0 - SG_H + j) starts a run (records outside the list count as starts)
    __shared__ u8 sg[LN];                    // members of the run that starts at j if it has at most SG_MAXG (else SG_MAXG + 1; 0: no start)
    __shared__ u8 code[256];
    code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    const long long t0 = (long long)blockIdx.x * SG_T - (long long)SG_H;      // list index of LDS slot 0
    for (u32 j = threadIdx.x; j < LN; j += 256) {
        const long long i = t0 + j;
        bool head = true; u64 b = 0; u32 x = 0;
        if (i >= 0 && i < (long long)m) {
            const u32 r = a_r1[i];
            head = (i == 0) || (a_r1[i - 1] != r);
            x = a_sa[i];
            u64 a2 = 0;
            ws_key_global<1>(g, code, (size_t)x + h, b, a2);
            if (j >= SG_H && j < SG_H + SG_T) k1[i] = (u64)r;
        }
        sk[j] = b; sv[j] = x; sh[j] = head ? 1 : 0;
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < LN; j += 256) {
        u32 size = 0;
        const long long i = t0 + j;
        if (i >= 0 && i < (long long)m && sh[j] && j < SG_H + SG_T) {          // starts in the left halo and in the tile (the right halo's belong to the next tile)
            u32 gsz = 1;
            while (gsz <= SG_MAXG && j + gsz < LN && !sh[j + gsz]) ++gsz;      // (j + SG_MAXG < LN for the starts looked at)
            size = gsz <= SG_MAXG ? gsz : SG_MAXG + 1;
            if (gsz > SG_MAXG && j >= SG_H) {                                   // a longer run that starts in this tile: its end from the list itself
                const u32 r = a_r1[i];
                u32 lo = (u32)i + gsz, hi = m;
                for (u32 step = 16; lo < hi; step *= 2) {
                    const u32 p = (hi - lo > step) ? lo + step - 1 : hi - 1;
                    if (a_r1[p] == r) lo = p + 1; else { hi = p; break; }
                }
                while (lo < hi) { const u32 mid = lo + (hi - lo) / 2; if (a_r1[mid] == r) lo = mid + 1; else hi = mid; }
                const u32 e = lo, i32 = (u32)i;
                if (e - i32 <= WS_WAVE_MAX) { rng[2 * (size_t)(i32 / SG_EDIV)] = i32; rng[2 * (size_t)(i32 / SG_EDIV) + 1] = e | 0x80000000u; }
                else {
                    const u32 slot = atomicAdd(&sc->nbig, 1u);
                    if (slot < big_cap) { big[2 * slot] = i32; big[2 * slot + 1] = e; atomicAdd(&sc->big_recs, e - i32); }
                    else sc->overflow = 1u;
                }
            }
        }
        sg[j] = (u8)size;
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < LN; j += 256) {
        const long long i = t0 + j;
        if (i < 0 || i >= (long long)m) continue;
        const bool own = j >= SG_H && j < SG_H + SG_T;
        u32 st = j, steps = 0;
        while (!sh[st] && st > 0 && steps < SG_MAXG) { --st; ++steps; }
        const u32 gsz = sh[st] ? sg[st] : SG_MAXG + 1;          // (no start within SG_MAXG records in front of it: a long run)
        if (gsz <= SG_MAXG && gsz >= 1) {                       // a short run (one member: a record alone -- cannot happen in a list of groups, written in place)
            if (st < SG_H || st >= SG_H + SG_T) continue;      // ... that starts in another tile: written there
            const u64 kj = sk[j];
            u32 r = 0;
            for (u32 l = st; l < st + gsz; ++l) { const u64 kl = sk[l]; r += (kl < kj || (kl == kj && l < j)) ? 1u : 0u; }
            const size_t o = (size_t)(t0 + st + r);
            k2[o] = kj; v[o] = sv[j];
        } else if (own) { k2[i] = sk[j]; v[i] = sv[j]; }       // a member of a long run: in place
    }
}
// the long runs to / from a list of their own: tab[r] = { first record, end, offset in the list }; one workgroup per run
struct SegBig { u32 a, e, off, pad; };
__global__ __launch_bounds__(256) void ws_seg_gather_kernel(const SegBig* __restrict__ tab, const u64* __restrict__ k1, const u64* __restrict__ k2, const u32* __restrict__ v,
                                                            u64* __restrict__ t1, u64* __restrict__ t2, u32* __restrict__ tv) {
    const SegBig b = tab[blockIdx.x];
    for (u32 j = threadIdx.x; j < b.e - b.a; j += 256) { t1[b.off + j] = k1[b.a + j]; t2[b.off + j] = k2[b.a + j]; tv[b.off + j] = v[b.a + j]; }
}
__global__ __launch_bounds__(256) void ws_seg_putback_kernel(const SegBig* __restrict__ tab, const u64* __restrict__ t2, const u32* __restrict__ tv, u64* __restrict__ k2, u32* __restrict__ v) {
    const SegBig b = tab[blockIdx.x];
    for (u32 j = threadIdx.x; j < b.e - b.a; j += 256) { k2[b.a + j] = t2[b.off + j]; v[b.a + j] = tv[b.off + j]; }
}

static bool ws_sorted_runs_impl(Ctx& c, const u64* k1, u64* k2, u32* v, size_t m, int k1_bits, const u32* a_sa, const u32* a_r1, const WKeyGen* g, u32 h, u64* k1_out);
bool wsort_sorted_runs(Ctx& c, const u64* k1, u64* k2, u32* v, size_t m, int k1_bits) { return ws_sorted_runs_impl(c, k1, k2, v, m, k1_bits, nullptr, nullptr, nullptr, 0, nullptr); }
bool wsort_sorted_runs_from_text(Ctx& c, const u32* a_sa, const u32* a_r1, size_t m, const WKeyGen& g, u32 h, u64* k1, u64* k2, u32* v, int k1_bits) {
    return ws_sorted_runs_impl(c, k1, k2, v, m, k1_bits, a_sa, a_r1, &g, h, k1);
}
static bool ws_sorted_runs_impl(Ctx& c, const u64* k1, u64* k2, u32* v, size_t m, int k1_bits, const u32* a_sa, const u32* a_r1, const WKeyGen* g, u32 h, u64* k1_out) {
    if (m < 2 && !g) return true;
    if (m >= ((size_t)1 << 31) || m == 0) return false;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    const u32 nent = (u32)(m / SG_EDIV + 2);
    u32* rng = c.arena.get<u32>(2 * (size_t)nent);
    u32* lists = c.arena.get<u32>(4 * (size_t)nent);              // (classes 0 .. 3 only: no entry is longer than WS_WAVE_MAX)
    u32* cnt = c.arena.get<u32>(2 * EC_NCLS);                      // per class: runs, records
    SegCounters* d_sc = (SegCounters*)c.arena.alloc(sizeof(SegCounters));
    const u32 big_cap = (u32)std::min<long>(std::max<long>(c.sa_seg_bigcap, 0), (long)SG_BIG_CAP);   // (option sa_seg_bigcap: tests make the table overflow)
    u32* big = c.arena.get<u32>(2 * (size_t)SG_BIG_CAP);
    u8* flags = c.arena.get<u8>(m + 8);                            // (the run kernels mark the heads inside a run: not used here)
    HIP_TRY(hipMemsetAsync(rng, 0, 2 * (size_t)nent * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(cnt, 0, 2 * EC_NCLS * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(SegCounters), s));
    {
        Ctx::ProfScope prof(c, K_WS_RUN, (u64)m * 32 + (u64)nent * 16);
        if (g) ws_seg_tile_keys_kernel<<<cdiv(m, SG_T), 256, 0, s>>>(a_sa, a_r1, (u32)m, *g, h, k1_out, k2, v, rng, big, big_cap, d_sc);
        else ws_seg_tile_kernel<<<cdiv(m, SG_T), 256, 0, s>>>(k1, k2, v, (u32)m, rng, big, big_cap, d_sc);
        LAUNCH_CHECK();
        ws_emit_compact_kernel<<<cdiv(nent, EC_TILE), 256, 0, s>>>(rng, nent, lists, nent, cnt);
        LAUNCH_CHECK();
    }
    u32 hc[2 * EC_NCLS];
    c.read_n(cnt, hc, 2 * EC_NCLS);
    const SegCounters hs = c.read(d_sc);
    if (c.wsort_log) {
        fprintf(stderr, "[sorted runs] m %zu | runs (records) of <= 32 / 64 / 256 / 1024:", m);
        for (int q = 0; q < 4; ++q) fprintf(stderr, " %u (%u)", hc[q], hc[EC_NCLS + q]);
        fprintf(stderr, " | longer: %u (%u)%s\n", hs.nbig, hs.big_recs, hs.overflow ? " -- too many" : "");
    }
    bool ok = !hs.overflow;
    for (int q = 4; q < EC_NCLS; ++q) ok = ok && hc[q] == 0;      // (cannot happen: such runs went to the list of long ones)
    if (ok) {
        WLeaf A;
        A.k1 = nullptr; A.k2 = k2; A.v = v; A.unit_rng = rng; A.flags = flags; A.lcp = nullptr; A.d_err = c.d_err; A.inv = 0; A.cmax = 1;
        const u64 recs = (u64)hc[EC_NCLS] + hc[EC_NCLS + 1] + hc[EC_NCLS + 2] + hc[EC_NCLS + 3];
        {
            Ctx::ProfScope prof(c, K_WS_RUN, recs * 18);
            const u32* l0 = lists, *l1 = lists + nent, *l2 = lists + 2 * (size_t)nent, *l3 = lists + 3 * (size_t)nent;
            if (hc[3]) { ws_run_wave_kernel<1024, true><<<cdiv(hc[3], 4), 256, 0, s>>>(A, l3, hc[3]); LAUNCH_CHECK(); }
            if (hc[2]) { ws_run_wave_kernel<256, true><<<cdiv(hc[2], 4), 256, 0, s>>>(A, l2, hc[2]); LAUNCH_CHECK(); }
            if (hc[0]) { ws_run_lane_kernel<32, true><<<cdiv(hc[0], 8), 256, 0, s>>>(A, l0, hc[0]); LAUNCH_CHECK(); }
            if (hc[1]) { ws_run_lane_kernel<64, true><<<cdiv(hc[1], 4), 256, 0, s>>>(A, l1, hc[1]); LAUNCH_CHECK(); }
        }
        if (hs.nbig) {
            // the runs no wave orders (a phrase that occurs thousands of times: 1 700 runs with 3 % of the records in the first round at 2e9 B
            // of English): copied into a list of their own, sorted as a whole -- first words included, so the runs stay in their order --,
            // copied back
            const u32 nb = hs.nbig;
            const size_t mb = hs.big_recs;
            std::vector<u32> hb(2 * (size_t)nb);
            c.read_n(big, hb.data(), 2 * (size_t)nb);
            std::vector<std::pair<u32, u32>> runs(nb);
            for (u32 r = 0; r < nb; ++r) runs[r] = { hb[2 * r], hb[2 * r + 1] };
            std::sort(runs.begin(), runs.end());
            SegBig* h_tab = (SegBig*)c.pinned_table(sizeof(SegBig) * nb);
            u32 off = 0;
            for (u32 r = 0; r < nb; ++r) { h_tab[r] = { runs[r].first, runs[r].second, off, 0u }; off += runs[r].second - runs[r].first; }
            if (off != mb) throw HipError{hipErrorUnknown, "sorted runs: the long runs do not add up", (int)__LINE__};
            SegBig* d_tab = (SegBig*)c.arena.alloc(sizeof(SegBig) * nb);
            HIP_TRY(hipMemcpyAsync(d_tab, h_tab, sizeof(SegBig) * nb, hipMemcpyHostToDevice, s));
            u64* T1[2] = { c.arena.get<u64>(mb), c.arena.get<u64>(mb) };
            u64* T2[2] = { c.arena.get<u64>(mb), c.arena.get<u64>(mb) };
            u32* TV[2] = { c.arena.get<u32>(mb), c.arena.get<u32>(mb) };
            ws_seg_gather_kernel<<<nb, 256, 0, s>>>(d_tab, k1, k2, v, T1[0], T2[0], TV[0]);
            LAUNCH_CHECK();
            // (a few million records: two stable LSD sorts -- 66 launches, 0.87 ms of kernels in the trace of a step -- against the partition
            //  path's 192 launches and 2.4 ms)
            int y = 1;
            if (mb < ((size_t)1 << 23)) ws_lsd_sort_wide(c, T1, T2, TV, mb, k1_bits);
            else y = wsort_records(c, T1, T2, TV, mb, k1_bits, nullptr);
            ws_seg_putback_kernel<<<nb, 256, 0, s>>>(d_tab, T2[y], TV[y], k2, v);
            LAUNCH_CHECK();
            HIP_TRY(hipStreamSynchronize(s));                      // (the pinned table is the context's: nobody else may fill it before the copy is through)
        }
    }
    c.arena.release(mark);
    return ok;
}

int wsort_records(Ctx& c, u64* K1[2], u64* K2[2], u32* V[2], size_t m, int k1_bits, WSortStats* st) {
    WSortStats local;
    if (!st) st = &local;
    *st = WSortStats();
    if (m < ((size_t)1 << 16)) {                               // small lists: two stable LSD sorts
        ws_lsd_sort_wide(c, K1, K2, V, m, k1_bits);
        return 1;
    }
    return ws_sort_impl<2, false, true>(c, nullptr, K1, K2, V, m, k1_bits, nullptr, nullptr, st);
}

}  // namespace tdc
