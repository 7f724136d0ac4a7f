// suffix_array.hip -- suffix array + inverse suffix array by prefix doubling (gfx950).
//
// Replaces ds/SADivSufSort.hpp:27-51 (divsufsort, util/divsufsort.hpp:46-279) and ds/ISAFromSA.hpp:30-43.
// The suffix array of a text is unique, so the result is bit-identical to the reference's.
//
// Algorithm (Larsson/Sadakane-style doubling, all data-parallel):
//   1. Pack the first k symbols of every suffix into a 64-bit key (symbols re-coded densely to b bits,
//      k = floor(64 / b)), radix sort (key, position).  Suffixes are now sorted by their first h = k symbols.
//   2. rank[i] := start index of i's group in the sorted order.  Groups of size 1 are final.
//   3. While unresolved groups exist: for every unresolved suffix i build key (rank[i], rank[i + h]), sort the
//      unresolved suffixes by it (groups stay inside their own index range), recompute group heads, drop the
//      new singletons, h *= 2.
//   The final rank array is the inverse suffix array.
// The sentinel (unique smallest byte at n-1) guarantees i + h <= n-1 for every unresolved suffix.
#include "stages.hpp"
#include "prim.hpp"

#include <stdlib.h>

namespace tdc {

struct CodeMap { u8 code[256]; };

// (round 6: 16 bytes per thread and step, eight copies of the histogram in rows of 257 words -- lane l counts in copy l % 8, the copies of
//  one byte value lie in eight different banks: the blanks of a text, a sixth of its bytes, no longer queue up on one LDS word.  One byte
//  per thread and one histogram: 0.15 ms per 125 MB chunk = 0.8 TB/s, 2.5 ms of device time per step behind the upload)
__global__ __launch_bounds__(256) void byte_hist_kernel(const u8* __restrict__ text, size_t n, u32* __restrict__ hist) {
    __shared__ u32 h[8 * 257];
    for (int i = threadIdx.x; i < 8 * 257; i += 256) h[i] = 0;
    __syncthreads();
    u32* mine = h + (threadIdx.x & 7) * 257;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    const bool aligned = (((size_t)text) & 15) == 0;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        if (aligned && i + 16 <= n) {
            const uint4 v = *(const uint4*)(text + i);
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int q = 0; q < 16; ++q) atomicAdd(&mine[(w[q >> 2] >> (8 * (q & 3))) & 0xFFu], 1u);
        } else {
            for (size_t j = i; j < n && j < i + 16; ++j) atomicAdd(&mine[text[j]], 1u);
        }
    }
    __syncthreads();
    u32 t = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += h[r * 257 + threadIdx.x];
    if (t) atomicAdd(&hist[threadIdx.x], t);
}

void text_histogram_add(Ctx& c, const u8* part, size_t len, u32* d_hist) {       // d_hist: 256 device counters, accumulated
    if (!len) return;
    unsigned g = cdiv(len, 256 * 16); if (g > 2048) g = 2048; if (g == 0) g = 1;
    byte_hist_kernel<<<g, 256, 0, c.stream>>>(part, len, d_hist);
    LAUNCH_CHECK();
}
void text_histogram_finish(Ctx& c, const u8* text, size_t n, const u32* d_hist) {   // read-back into the context's cache
    c.read_n(d_hist, c.hist_cache, 256);
    c.hist_ptr = text; c.hist_n = n;
}

// 1024 positions per workgroup, symbols staged (already re-coded) in LDS.
// key = the first k symbols of the suffix as a k-digit number in base sigma (dense codes): order-preserving and as many
// symbols as 64 bits can hold (13 instead of 12 for the 29 symbols of the English-like corpus, 27 instead of 21 for DNA)
// (evaluated in chunks of `chunk` symbols that fit 32 bits: one 64-bit multiply per chunk instead of one per symbol)
__global__ __launch_bounds__(256) void sa_init_keys_kernel(const u8* __restrict__ text, size_t n, CodeMap cm, u32 sigma, int k,
                                                            int chunk, u64* __restrict__ keys, u32* __restrict__ vals) {
    __shared__ u8 s[1024 + 64];
    __shared__ u8 code[256];
    code[threadIdx.x] = cm.code[threadIdx.x];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 1024;
    for (int i = threadIdx.x; i < 1024 + 64; i += 256) {
        const size_t p = base + i;
        s[i] = (p < n) ? code[text[p]] : (u8)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int local = r * 256 + threadIdx.x;
        const size_t p = base + local;
        if (p < n) {
            u64 key = 0;
            for (int j0 = 0; j0 < k; j0 += chunk) {
                const int len = (k - j0 < chunk) ? k - j0 : chunk;
                u32 acc = 0, scale = 1;
                for (int j = 0; j < len; ++j) { acc = acc * sigma + s[local + j0 + j]; scale *= sigma; }
                key = (j0 == 0) ? (u64)acc : key * scale + acc;
            }
            keys[p] = key;
            vals[p] = (u32)p;
        }
    }
}

__global__ void sa_build_keys_kernel(const u32* __restrict__ a_sa, const u32* __restrict__ a_r1, size_t m, size_t n, u32 h,
                                     int bn, const u32* __restrict__ rank, u64* __restrict__ keys, u32* __restrict__ vals) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    const u32 s = a_sa[a];
    const size_t t = (size_t)s + h;
    const u32 r2 = (t < n) ? rank[t] : 0u;
    keys[a] = ((u64)a_r1[a] << bn) | r2;
    vals[a] = s;
}

// Doubling round, local part: the active list is grouped by r1 (runs of equal high key half), and sorting by
// (r1, r2) only permutes elements inside their run.  A 2048-element tile is sorted entirely in registers/LDS
// (bitonic network); runs that continue across a tile border are only partially ordered by that and are flagged
// (cls = 1) for the global radix sort.  Most runs are a handful of suffixes, so the global sort shrinks to a
// small fraction of the active list.
__global__ __launch_bounds__(256) void sa_local_sort_kernel(u64* __restrict__ keys, u32* __restrict__ vals, size_t m, int bn,
                                                             u8* __restrict__ cls) {
    __shared__ u64 xk[2048];
    __shared__ u32 xv[2048];
    const size_t base = (size_t)blockIdx.x * 2048;
    const size_t end = (base + 2048 < m) ? base + 2048 : m;
    // runs that cross the tile borders
    const u64 first_r1 = keys[base] >> bn, last_r1 = keys[end - 1] >> bn;
    const bool open_l = base > 0 && (keys[base - 1] >> bn) == first_r1;
    const bool open_r = end < m && (keys[end] >> bn) == last_r1;
    if (first_r1 == last_r1 && (open_l || open_r)) {         // the whole tile lies inside one long run: left to the global sort
        for (size_t i = base + threadIdx.x; i < end; i += 256) cls[i] = 1;
        return;
    }
    u64 k[8];
    u32 v[8];
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const size_t i = base + threadIdx.x * 8 + r;
        k[r] = (i < m) ? keys[i] : ~0ull;
        v[r] = (i < m) ? vals[i] : 0u;
    }
    block_bitonic_sort_2048(k, v, xk, xv);
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const size_t i = base + threadIdx.x * 8 + r;
        if (i < m) {
            keys[i] = k[r];
            vals[i] = v[r];
            const u64 r1 = k[r] >> bn;                      // the sort keeps every element inside its run
            cls[i] = ((open_l && r1 == first_r1) || (open_r && r1 == last_r1)) ? 1 : 0;
        }
    }
}
__global__ void sa_scatter_back_kernel(const u32* __restrict__ opos, const u64* __restrict__ okeys, const u32* __restrict__ ovals,
                                       size_t mo, u64* __restrict__ keys, u32* __restrict__ vals) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= mo) return;
    const u32 p = opos[j];
    keys[p] = okeys[j];
    vals[p] = ovals[j];
}

// ---- refinement of the initial order from the text ------------------------------------------------------------------------------
// After the initial sort the suffixes are ordered by their first k symbols; most groups of equal keys are a handful of suffixes,
// yet the first doubling round would send all their members (two thirds of a 2 GB English text) through a rank scatter, a
// scattered rank gather, a sort and a second rank scatter.  Here every group of 2 .. RF_MAXRUN members that lies inside one tile is
// ordered by the NEXT k symbols of its members right away: the keys come from the text (one scattered read per member -- the same
// read the round's gather would have cost), the order inside the group from counting (lengths are tiny), and the groups' new
// boundaries go out as one flag byte per element, which is all the bookkeeping pass needs.  Longer groups and groups that cross a
// tile border (the frequent words) keep their order and stay one group: the doubling rounds, which start at h = k as before, deal
// with them -- every group still shares at least k symbols -- and with the few short groups that tie on 2k symbols.
#ifndef TDC_RF_MAXRUN
#define TDC_RF_MAXRUN 256
#endif
constexpr u32 RF_TILE = 2048, RF_MAXRUN = TDC_RF_MAXRUN;     // (counting costs a group its length squared)
struct RefineGen { const u8* text; size_t n; u32 sigma; int k, chunk; u8 code[256]; };
__device__ __forceinline__ u64 rf_text_key(const RefineGen& g, const u8* __restrict__ code, size_t q) {
    u64 w[4] = { 0, 0, 0, 0 };
    if (q + 32 <= g.n) __builtin_memcpy(w, g.text + q, 32);   // four unaligned 8-byte words
    else {
#pragma unroll
        for (int t = 0; t < 32; ++t) if (q + t < g.n) w[t >> 3] |= (u64)g.text[q + t] << (8 * (t & 7));
    }
    // the k-digit number in base sigma, evaluated in chunks that fit 32 bits (as the initial keys); static indices throughout
    u64 key = 0;
    u32 acc = 0, scale = 1;
    int cnt = 0;
    bool first = true;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j < g.k) {
            const u32 sym = (q + j < g.n) ? (u32)code[(u8)(w[j >> 3] >> (8 * (j & 7)))] : 0u;   // beyond the text: the padding the initial keys use
            acc = acc * g.sigma + sym; scale *= g.sigma; ++cnt;
            if (cnt == g.chunk || j == g.k - 1) { key = first ? (u64)acc : key * scale + acc; first = false; acc = 0; scale = 1; cnt = 0; }
        }
    }
    return key;
}
__global__ __launch_bounds__(256) void sa_refine_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals, size_t m, RefineGen g,
                                                        u8* __restrict__ flags, u32* __restrict__ vals_out) {
    __shared__ u64 sk[RF_TILE + 2];          // keys of the elements t0 - 1 .. t0 + 2048, later the second keys of the refined elements
    __shared__ u32 sp[RF_TILE], snew[RF_TILE];
    __shared__ u64 hb[RF_TILE / 64 + 1];     // group heads of the tile as a bitmap; bit 2048: the element behind the tile starts a group
    __shared__ u8 sflag[RF_TILE];
    __shared__ u8 code[256];
    __shared__ u32 s_any;
    const size_t t0 = (size_t)blockIdx.x * RF_TILE;
    const int lane = lane_id(), w = wave_id();
    code[threadIdx.x] = g.code[threadIdx.x];
    if (threadIdx.x == 0) s_any = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32 e = (u32)r * 256 + threadIdx.x;
        const size_t i = t0 + e;
        sk[e + 1] = (i < m) ? keys[i] : 0ull;
        const u32 pv = (i < m) ? vals[i] : 0u;
        sp[e] = pv; snew[e] = pv;
    }
    if (threadIdx.x == 0) {
        sk[0] = (t0 >= 1) ? keys[t0 - 1] : 0ull;
        sk[RF_TILE + 1] = (t0 + RF_TILE < m) ? keys[t0 + RF_TILE] : 0ull;
    }
    __syncthreads();
    // heads: one ballot per row of 64 consecutive elements = one word of the bitmap
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32 e = (u32)r * 256 + threadIdx.x;
        const size_t i = t0 + e;
        const bool head = i < m && (i == 0 || sk[e + 1] != sk[e]);
        const u64 bm = __ballot(head || i >= m);               // (positions behind the end count as heads: they close the last group)
        if (lane == 0) hb[r * 4 + w] = bm;
        sflag[e] = head ? 1 : 0;
    }
    if (threadIdx.x == 0) hb[RF_TILE / 64] = (t0 + RF_TILE >= m || sk[RF_TILE + 1] != sk[RF_TILE]) ? 1ull : 0ull;
    __syncthreads();
    // the group of every element: [rs, re) if it lies inside the tile and has 2 .. RF_MAXRUN members
    u32 rs[8], re[8];
    u64 k2[8];
    u32 mine = 0;                                              // bit r: element r of this thread is refined
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32 e = (u32)r * 256 + threadIdx.x;
        rs[r] = re[r] = 0; k2[r] = 0;
        if (t0 + e >= m) continue;
        const u32 wi = e >> 6, b = e & 63;
        const u64 cw = hb[wi];
        // nearest head at or before e / behind e, looking at most RF_MAXRUN elements far
        u64 x = cw & ((b == 63) ? ~0ull : ((2ull << b) - 1ull));
        u32 wl = wi;
        while (!x && wl > 0 && wi - wl < RF_MAXRUN / 64 + 1) x = hb[--wl];
        if (!x) continue;                                      // the group starts further back, or in front of the tile
        const u32 s0 = wl * 64 + 63 - (u32)__builtin_clzll(x);
        u64 y = (b == 63) ? 0ull : (cw & (~0ull << (b + 1)));
        u32 wr = wi;
        while (!y && wr < RF_TILE / 64 && wr - wi < RF_MAXRUN / 64 + 1) y = hb[++wr];
        if (!y) continue;                                      // ... ends further ahead, or behind the tile
        const u32 e1 = wr * 64 + (u32)__builtin_ctzll(y);
        const u32 len = e1 - s0;
        if (len < 2 || len > RF_MAXRUN) continue;
        rs[r] = s0; re[r] = e1; mine |= 1u << r;
    }
    // second keys of the refined elements, all of a thread's scattered reads in flight together
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (mine & (1u << r)) k2[r] = rf_text_key(g, code, (size_t)sp[(u32)r * 256 + threadIdx.x] + (size_t)g.k);
    if (__any(mine != 0) && lane == 0) s_any = 1;
    __syncthreads();                                           // (every thread is done with the first keys)
    if (s_any == 0) {                                          // nothing to refine here: the order as it is, and the flags
#pragma unroll
        for (int r = 0; r < 8; ++r) { const u32 e = (u32)r * 256 + threadIdx.x; if (t0 + e < m) { vals_out[t0 + e] = sp[e]; flags[t0 + e] = sflag[e]; } }
        return;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) if (mine & (1u << r)) sk[(u32)r * 256 + threadIdx.x] = k2[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (!(mine & (1u << r))) continue;
        const u32 e = (u32)r * 256 + threadIdx.x;
        u32 less = 0, eqb = 0;
        for (u32 j = rs[r]; j < re[r]; ++j) {
            const u64 kj = sk[j];
            less += (kj < k2[r]) ? 1u : 0u;
            eqb += (kj == k2[r] && j < e) ? 1u : 0u;
        }
        const u32 t = rs[r] + less + eqb;
        snew[t] = sp[e];
        sflag[t] = (eqb == 0) ? 1 : 0;                          // the first member of its (k, 2k)-symbol group
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32 e = (u32)r * 256 + threadIdx.x;
        if (t0 + e < m) { vals_out[t0 + e] = snew[e]; flags[t0 + e] = sflag[e]; }
    }
}

// ---- group bookkeeping of one round in ONE pass ------------------------------------------------------------------------------
// Input: the (key, value) list sorted inside its groups; equal keys = one (new) group.  Per element: pos = its suffix-array slot
// (the index itself in the first round, else a_pos[]), head = pos of the first element of its group.  Output: sa[pos] = value,
// the new rank (= head) either into newrank_out[] (for the bucketed scatter) or straight into rank[], and the compacted list
// (value, pos, head) of the elements whose group is not a singleton.  Replaces five passes (head flags, max-scan, update, sum-scan,
// compaction: ~70 B per element) by one (~26 B): a tile of 2048 elements, the two running values that cross tile borders -- the
// number of kept elements and the head of the run that is open at the border -- travel through a decoupled look-back chain
// (one 64-bit descriptor per tile: 2 flag bits | 31-bit count | 31-bit head position).
constexpr u32 GR_TILE = 2048, GR_NONE = 0x7FFFFFFFu;
constexpr u64 GR_FLAG_AGG = 1ull << 62, GR_FLAG_INC = 2ull << 62;
__device__ __forceinline__ u64 gr_pack(u64 flag, u32 count, u32 hp) { return flag | ((u64)count << 31) | (u64)hp; }
__device__ __forceinline__ u32 gr_pad(u32 i) { return i + (i >> 3); }      // LDS slot of tile element i: threads read 8 consecutive elements
// WIDE = 1: a text round of the rank-free path (suffix_array.hip build_suffix_array_wide).  keys = slot of the group head, keys2 = the
// next 64 key bits of the suffix (bit-packed symbols from position + h): a new group starts where either differs.  A start inside a
// parent group (keys equal, keys2 different) gets flags_out[pos] = 1 and lcp_out[pos] = h + number of leading symbols the two
// second keys share (wr.inv = ceil(65536 / bits per symbol)); no rank is written.
struct WideRound { const u64* keys2; u8* flags_out; u8* lcp_out; u32 h, inv; };
template <bool FIRST, int WIDE>
__global__ __launch_bounds__(256) void sa_groups_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals, const u32* __restrict__ a_pos,
                                                        size_t m, int bn, u32* __restrict__ sa, u32* __restrict__ rank, u32* __restrict__ newrank_out,
                                                        u32* __restrict__ o_sa, u32* __restrict__ o_pos, u32* __restrict__ o_r1,
                                                        u64* desc, u32* __restrict__ d_total, u32* err, u32 numTiles,
                                                        const u8* __restrict__ hflags, WideRound wr) {
    __shared__ u32 s_hp[4], s_cnt[5];
    __shared__ u32 s_carry_hp, s_carry_cnt;
    __shared__ u64 sk[GR_TILE + GR_TILE / 8 + 8];       // keys of the elements tile0 - 1 .. tile0 + 2048 (slot 0 = the predecessor)
    __shared__ u64 sk2[WIDE ? GR_TILE + GR_TILE / 8 + 8 : 1];
    __shared__ u32 sv[GR_TILE + GR_TILE / 8], sp[GR_TILE + GR_TILE / 8];
    // Tiles are numbered by blockIdx: workgroups are dispatched in that order, so the predecessors of a running tile have been
    // dispatched (a ticket counter would make that formal, but one device-wide atomic per tile on ONE address costs ~25 ns each:
    // 3.6 ms for the 131 072 tiles of a 256 MiB text, more than the whole pass).  The look-back spin is bounded by wall-clock time
    // (30 s: another context's long kernel on the same device only delays it), then the error flag is raised instead of hanging.
    const u32 tile = blockIdx.x;
    const int lane = lane_id(), w = wave_id();
    const size_t t0 = (size_t)tile * GR_TILE;
    // coalesced loads (a row of 256 consecutive elements per step) into LDS; sa[pos] = value on the way
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32 e = (u32)r * 256 + threadIdx.x;
        const size_t i = t0 + e;
        u64 kk = 0; u32 vv = 0, pp = 0;
        if (i < m) {
            kk = hflags ? 0ull : keys[i]; vv = vals[i]; pp = FIRST ? (u32)i : a_pos[i];
            if (!(FIRST && vals == sa)) sa[pp] = vv;      // (after a refinement the refined order already IS the suffix array so far)
        }
        sk[gr_pad(e + 1)] = kk; sv[gr_pad(e)] = vv; sp[gr_pad(e)] = pp;
        if (WIDE) sk2[gr_pad(e + 1)] = (i < m) ? wr.keys2[i] : 0ull;
    }
    if (threadIdx.x == 0 && !hflags) { sk[0] = (t0 >= 1) ? keys[t0 - 1] : 0ull; sk[gr_pad(GR_TILE + 1)] = (t0 + GR_TILE < m) ? keys[t0 + GR_TILE] : 0ull; }
    if (WIDE && threadIdx.x == 0) { sk2[0] = (t0 >= 1) ? wr.keys2[t0 - 1] : 0ull; sk2[gr_pad(GR_TILE + 1)] = (t0 + GR_TILE < m) ? wr.keys2[t0 + GR_TILE] : 0ull; }
    __syncthreads();
    const u32 l0 = threadIdx.x * 8;                     // the thread's 8 consecutive elements
    const size_t i0 = t0 + l0;
    u64 k[10];
    u32 v[8], ps[8];
    u64 k2[WIDE ? 10 : 1];
#pragma unroll
    for (int r = 0; r < 10; ++r) k[r] = sk[gr_pad(l0 + r)];      // k[r] = key of element i0 + r - 1
    if (WIDE) {
#pragma unroll
        for (int r = 0; r < 10; ++r) k2[r] = sk2[gr_pad(l0 + r)];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r] = sv[gr_pad(l0 + r)]; ps[r] = sp[gr_pad(l0 + r)]; }
    // run starts, head position of every element as far as the thread can tell, number of elements to keep
    u32 starts = 0;
    if (hflags) {
        // group heads from the flags of the refinement pass (they also split groups of equal FIRST keys): the thread's eight flag
        // bytes are one aligned 8-byte word
        u64 fw = 0;
        if (i0 + 8 <= m) fw = *(const u64*)(hflags + i0);
        else for (int r = 0; r < 8; ++r) if (i0 + r < m) fw |= (u64)hflags[i0 + r] << (8 * r);
#pragma unroll
        for (int r = 0; r < 8; ++r) if ((fw >> (8 * r)) & 0xFFull) starts |= 1u << r;
        if (i0 + 8 < m) { if (hflags[i0 + 8]) starts |= 1u << 8; }
#pragma unroll
        for (int r = 0; r < 9; ++r) if (i0 + r == m) starts |= 1u << r;     // the end of the list closes the last run
    } else {
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const size_t i = i0 + r;
            bool differ = k[r + 1] != k[r];
            if (WIDE) differ = differ || k2[r + 1] != k2[r];
            if (i < m && (i == 0 || differ)) starts |= 1u << r;
            if (i == m) starts |= 1u << r;                  // the end of the list closes the last run
        }
    }
    if (WIDE) {         // new group heads inside a parent group: flag and LCP (the parent borders have theirs already)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const size_t i = i0 + r;
            if (i < m && i > 0 && ((starts >> r) & 1u) && k[r + 1] == k[r]) {
                const u64 x = k2[r + 1] ^ k2[r];
                wr.flags_out[ps[r]] = 1;
                wr.lcp_out[ps[r]] = (u8)(wr.h + ((((u32)__builtin_clzll(x)) * wr.inv) >> 16));
            }
        }
    }
    u32 hp[8];
    u32 run = GR_NONE, keepmask = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const size_t i = i0 + r;
        if ((starts & (1u << r)) && i < m) run = ps[r];
        hp[r] = run;
        const bool single = ((starts >> r) & 3u) == 3u;  // starts a run and the next element starts one too
        if (i < m && !single) keepmask |= 1u << r;
    }
    // every 64th tile: how many elements stay unresolved and in how many groups (d_total[1], [2]) -- the host picks the sort of the
    // next round by the average group size
    if ((tile & 63u) == 0) {
        const u32 kc = wave_reduce_sum((u32)__popc(keepmask)), gc = wave_reduce_sum((u32)__popc(keepmask & starts & 0xFFu));
        if (lane == 0 && kc) { atomicAdd(d_total + 1, kc); atomicAdd(d_total + 2, gc); }
    }
    // across the threads of the tile: last run start before this thread, exclusive count of kept elements
    u32 inc_hp = run;                                   // inclusive "last start" scan: the right operand wins unless it has none
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl_up(inc_hp, d, 64);
        if (lane >= d && inc_hp == GR_NONE) inc_hp = o;
    }
    const u32 mycnt = (u32)__popc(keepmask);
    const u32 inc_cnt = wave_inclusive_sum(mycnt);
    if (lane == 63) { s_hp[w] = inc_hp; s_cnt[w] = inc_cnt; }
    __syncthreads();                                    // (also: every thread holds its elements in registers, the LDS tile is free)
    u32 pre_hp = __shfl_up(inc_hp, 1, 64);              // last start in the earlier lanes of the wave
    if (lane == 0) pre_hp = GR_NONE;
    u32 pre_cnt = inc_cnt - mycnt;
    u32 tile_hp = GR_NONE, tile_cnt = 0;
    {   // earlier waves: the LAST one that has a start wins; counts add up
        u32 whp = GR_NONE, wc = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < w) { if (s_hp[i] != GR_NONE) whp = s_hp[i]; wc += s_cnt[i]; }
            if (s_hp[i] != GR_NONE) tile_hp = s_hp[i];
            tile_cnt += s_cnt[i];
        }
        if (pre_hp == GR_NONE) pre_hp = whp;
        pre_cnt += wc;
    }
    // look-back: running values at the left border of the tile.  Wave 0 inspects 64 predecessors per step (lane l: tile - 1 - l -
    // 64 step): the walk ends at the nearest tile that has published its inclusive values, and it only has to wait for tiles that
    // have not even published their aggregate.
    if (w == 0) {
        u32 c_cnt = 0, c_hp = GR_NONE;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(desc, gr_pack(GR_FLAG_INC, tile_cnt, tile_hp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(desc + tile, gr_pack(GR_FLAG_AGG, tile_cnt, tile_hp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long base = (long)tile - 1;
            u32 spins = 0;
            const unsigned long long spin_t0 = wall_clock64();
            for (;;) {
                const long idx = base - lane;
                const u64 d64 = (idx >= 0) ? __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : gr_pack(GR_FLAG_INC, 0u, GR_NONE);      // in front of tile 0: neutral, ends the walk
                const u64 f = d64 >> 62;
                const u64 m_inc = __ballot(f == 2), m_zero = __ballot(f == 0);
                const int first_inc = m_inc ? __ffsll((long long)m_inc) - 1 : 64;
                const u64 need = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1);   // the lanes up to the nearest inclusive one
                if (m_zero & need) {
                    if ((++spins & 1023u) == 0 && wall_clock64() - spin_t0 > 3000000000ull) {   // (100 MHz clock) never hang the GPU: report and leave
                        if (lane == 0) atomicOr(err, 4u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                const bool mine = (need >> lane) & 1ull;
                c_cnt += wave_reduce_sum(mine ? (u32)((d64 >> 31) & 0x7FFFFFFFu) : 0u);
                const u32 hpv = mine ? (u32)(d64 & 0x7FFFFFFFu) : GR_NONE;
                const u64 m_hp = __ballot(hpv != GR_NONE);
                if (c_hp == GR_NONE && m_hp) c_hp = __shfl(hpv, __ffsll((long long)m_hp) - 1, 64);   // the nearest tile that has a run start
                if (first_inc < 64) break;
                base -= 64;
            }
            if (lane == 0) __hip_atomic_store(desc + tile, gr_pack(GR_FLAG_INC, c_cnt + tile_cnt, tile_hp != GR_NONE ? tile_hp : c_hp), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_carry_cnt = c_cnt; s_carry_hp = c_hp;
            if (tile == numTiles - 1) *d_total = c_cnt + tile_cnt;
        }
    }
    __syncthreads();
    if (pre_hp == GR_NONE) pre_hp = s_carry_hp;
    // results into LDS (heads by tile slot, the kept elements compacted), then coalesced stores
    u32* c_sa = (u32*)sk; u32* c_pos = c_sa + GR_TILE; u32* c_r1 = sv; u32* s_h = sp;
    u32 o = pre_cnt;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32 h = (hp[r] != GR_NONE) ? hp[r] : pre_hp;
        s_h[gr_pad(l0 + r)] = h;
        if (keepmask & (1u << r)) { c_sa[o] = v[r]; c_pos[o] = ps[r]; c_r1[o] = h; ++o; }
    }
    __syncthreads();
    const size_t obase = s_carry_cnt;
    for (u32 q = threadIdx.x; q < tile_cnt; q += 256) { o_sa[obase + q] = c_sa[q]; o_pos[obase + q] = c_pos[q]; o_r1[obase + q] = c_r1[q]; }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32 e = (u32)r * 256 + threadIdx.x;
        const size_t i = t0 + e;
        if (i < m) {
            const u32 h = s_h[gr_pad(e)];
            if (newrank_out) newrank_out[i] = h;
            else if (!WIDE && rank) {                   // small inputs / rounds: straight into rank[]; later rounds only where the group was split
                const u32 vv = vals[i];
                if (FIRST || h != (u32)(keys[i] >> bn)) rank[vv] = h;
            }
        }
    }
}

// ---- text rounds of the rank-free path: keys of the unresolved suffixes -----------------------------------------------------------
// k1 = slot of the group head, k2 = the s1 symbols behind the h the group already shares (bit-packed, left-aligned), v = position
__global__ __launch_bounds__(256) void sa_round_keys_kernel(const u32* __restrict__ a_sa, const u32* __restrict__ a_r1, size_t m, WKeyGen g, u32 h,
                                                            u64* __restrict__ k1, u64* __restrict__ k2, u32* __restrict__ v) {
    __shared__ u8 code[256];
    code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const u32 p = a_sa[j];
    const size_t q = (size_t)p + h;
    u64 w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const int nw = (g.s + 7) >> 3;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < nw) {
            if (q + 8 * (size_t)t + 8 <= g.n) __builtin_memcpy(&w[t], g.text + q + 8 * t, 8);
            else for (int e = 0; e < 8; ++e) if (q + 8 * (size_t)t + e < g.n) w[t] |= (u64)g.text[q + 8 * t + e] << (8 * e);
        }
    }
    u64 key = 0;
#pragma unroll
    for (int t = 0; t < 64; ++t)
        if (t < g.s) key = (key << g.b) | ((q + t < g.n) ? (u64)code[(u8)(w[t >> 3] >> (8 * (t & 7)))] : 0ull);
    k1[j] = a_r1[j];
    k2[j] = key << g.pad;
    v[j] = p;
}
__global__ void sa_isa_direct_kernel(const u32* __restrict__ sa, size_t n, u32* __restrict__ isa) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) isa[sa[i]] = (u32)i;
}

// ---- fast path: the unresolved suffixes behind the wide sort, from the head flags alone ------------------------------------------------
// A slot is resolved when it is a head and its successor is one too (a group of one).  Everything else -- a few per cent of a
// natural-language text -- is compacted into the active list (position, slot, slot of its group's head) in slot order.  Two streaming
// passes over the flag bytes (tiles of 4096 slots) with two small scans in between; the suffix array itself is not touched.
constexpr u32 FC_TILE = 4096;
__device__ __forceinline__ u32 fc_head_mask(const u8* __restrict__ flags, size_t n, size_t i0) {     // bit q: slot i0 + q is a head (q <= 16; slots >= n count as heads)
    u32 mask = 0;
    if (i0 + 17 <= n && ((i0 & 15) == 0) && ((((size_t)flags) & 15) == 0)) {
        const uint4 v = *(const uint4*)(flags + i0);
        const u32 wv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int q = 0; q < 16; ++q) if ((wv[q >> 2] >> (8 * (q & 3))) & 0xFFu) mask |= 1u << q;
        if (flags[i0 + 16]) mask |= 1u << 16;
    } else {
#pragma unroll
        for (int q = 0; q < 17; ++q) if (i0 + q >= n || flags[i0 + q]) mask |= 1u << q;
    }
    if (i0 == 0) mask |= 1u;
    return mask;
}
__global__ __launch_bounds__(256) void sa_flag_count_kernel(const u8* __restrict__ flags, size_t n, u32* __restrict__ tile_cnt, u32* __restrict__ tile_last) {
    __shared__ u32 sm[5], sm2[5];
    const size_t i0 = (size_t)blockIdx.x * FC_TILE + (size_t)threadIdx.x * 16;
    u32 cnt = 0, last = 0;
    if (i0 < n) {
        const u32 h = fc_head_mask(flags, n, i0);
        const u32 valid = (i0 + 16 <= n) ? 0xFFFFu : ((1u << (n - i0)) - 1u);
        const u32 single = h & (h >> 1) & valid;
        cnt = (u32)__popc(valid & ~single);
        const u32 hv = h & valid;
        if (hv) last = (u32)(i0 + 31 - __builtin_clz(hv)) + 1u;
    }
    u32 total;
    (void)block_exclusive_sum<u32, 4>(cnt, sm, total);
    u32 tmax;
    (void)block_inclusive_max<4>(last, sm2, tmax);
    if (threadIdx.x == 0) { tile_cnt[blockIdx.x] = total; tile_last[blockIdx.x] = tmax; }
}
__global__ __launch_bounds__(256) void sa_flag_compact_kernel(const u8* __restrict__ flags, const u32* __restrict__ v, size_t n, const u32* __restrict__ tile_off,
                                                               const u32* __restrict__ tile_lastscan, u32* __restrict__ o_sa, u32* __restrict__ o_pos,
                                                               u32* __restrict__ o_r1) {
    __shared__ u32 sm[5], sm2[5];
    const size_t i0 = (size_t)blockIdx.x * FC_TILE + (size_t)threadIdx.x * 16;
    u32 cnt = 0, last = 0, h = 0, valid = 0, single = 0;
    if (i0 < n) {
        h = fc_head_mask(flags, n, i0);
        valid = (i0 + 16 <= n) ? 0xFFFFu : ((1u << (n - i0)) - 1u);
        single = h & (h >> 1) & valid;
        cnt = (u32)__popc(valid & ~single);
        const u32 hv = h & valid;
        if (hv) last = (u32)(i0 + 31 - __builtin_clz(hv)) + 1u;
    }
    u32 total;
    const u32 pre = block_exclusive_sum<u32, 4>(cnt, sm, total);
    u32 tmax;
    const u32 inc = block_inclusive_max<4>(last, sm2, tmax);
    // head of the run that is open at the thread's first slot: the last head in front of it (earlier threads, earlier tiles)
    u32 run = __shfl_up(inc, 1, 64);
    if (lane_id() == 0) run = sm2[wave_id()];                       // (block_inclusive_max leaves the exclusive wave prefixes in sm2[0..3])
    // (round 6: a thread owns 16 slots of which one is listed on average -- loading and storing from there meant sixteen loads and
    //  forty-eight stores per wave with a few lanes each.  The listed slots go through LDS instead: the thread enters its slots and
    //  their heads, then consecutive lanes take consecutive entries -- every load and store of the list has all its lanes, the
    //  stores are whole lines.  2.78 -> see DESIGN section 9)
    __shared__ unsigned short l_slot[FC_TILE];
    __shared__ u32 l_head[FC_TILE];
    if (cnt) {
        if (blockIdx.x > 0) run = max(run, tile_lastscan[blockIdx.x - 1]);
        u32 o = pre;
        const u32 want = valid & ~single;                            // the slots this thread lists
        const u32 hv = h & valid;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if ((want >> q) & 1u) {
                const u32 hq = hv & ((2u << q) - 1u);                // heads at or in front of slot q inside the thread's range
                const u32 r = hq ? (u32)(i0 + 31 - __builtin_clz(hq)) + 1u : run;
                l_slot[o] = (unsigned short)(threadIdx.x * 16 + q); l_head[o] = r - 1u; ++o;
            }
        }
    }
    __syncthreads();
    const size_t t0 = (size_t)blockIdx.x * FC_TILE;
    const u32 obase = tile_off[blockIdx.x];
    for (u32 e = threadIdx.x; e < total; e += 256) {
        const size_t slot = t0 + l_slot[e];
        o_sa[obase + e] = v[slot]; o_pos[obase + e] = (u32)slot; o_r1[obase + e] = l_head[e];
    }
}

namespace {
struct SABufs {
    u64* keys[2]; u32* vals[2];
    u32 *head, *keep;
    u32 *A_sa, *A_pos, *A_r1, *B_sa, *B_pos, *B_r1;
    u32* d_total; u64* lkeys; u32* lvals; u64* gdesc;
};
const WideRound NO_WIDE = { nullptr, nullptr, nullptr, 0, 0 };

// first bookkeeping pass over the initial order (keys[x] / vals, or the head flags of a refinement / the wide sort) + the rank
// scatter; returns the number of unresolved suffixes (h_tot = the kernel's counters)
size_t first_groups(Ctx& c, size_t n, int bn, const u64* keys, const u32* vals, const u8* hflags, u32* sa, u32* rank, SABufs& B, bool want_ranks,
                    u32 h_tot[4]) {
    hipStream_t s = c.stream;
    const u32 tiles = cdiv(n, GR_TILE);
    HIP_TRY(hipMemsetAsync(B.gdesc, 0, (size_t)tiles * sizeof(u64), s));
    HIP_TRY(hipMemsetAsync(B.d_total, 0, 4 * sizeof(u32), s));
    const bool bucketed = want_ranks && c.bucket_scatter && n >= ((size_t)1 << 22);
    {   // per element: read key + value (12 B), write sa + head (8 B) + the kept elements (12 B each, about half of them)
        Ctx::ProfScope prof(c, K_SA_RANK_SCATTER, (u64)n * 26);
        sa_groups_kernel<true, 0><<<tiles, 256, 0, s>>>(keys, vals, nullptr, n, bn, sa, want_ranks ? rank : nullptr, bucketed ? B.head : nullptr,
                                                        B.A_sa, B.A_pos, B.A_r1, B.gdesc, B.d_total, c.d_err, tiles, hflags, NO_WIDE);
        LAUNCH_CHECK();
    }
    // rank[vals[j]] = head[j] through a partition by destination window; the second key buffer and the B lists are free scratch
    if (bucketed) bucketed_scatter_u32(c, vals, B.head, n, rank, n, (u32*)B.keys[1], B.vals[1], B.B_sa, B.B_pos, true);   // every position once
    c.read_n(B.d_total, h_tot, 4);
    return h_tot[0];
}

// ---- groups of exactly two suffixes, ordered from the text in one pass (round 5) -------------------------------------------------------
// Texts with long repeats (copied blocks) leave most unresolved suffixes in PAIRS {x, y}: a position of the copy and the same position
// of its source.  Doubling orders such a pair once h exceeds the rest of the repeat -- log(4096 / h) rounds over every position of
// every repeat (10^9 B of DNA: 8 rounds of ~33 ms over 250 M suffixes).  But the longest common extension of a pair is the one of the
// pair one position earlier minus one (as long as both are pairs, the partner of x + 1 is the partner of x plus one): the values
// lce(x, partner) satisfy the bound of the PLCP array, len[x] >= len[x - 1] - 1, and the chunked carry evaluation of build_plcp
// (textds.hip, build_lce_with_carry) computes all of them in one pass over the text.  The first byte behind the common extension
// decides the order.  What remains for the doubling rounds are the groups of three and more.
__global__ __launch_bounds__(256) void sa_pair_mark_kernel(const u32* __restrict__ a_sa, const u32* __restrict__ a_r1, size_t m, u32* __restrict__ src) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a + 1 >= m) return;
    const u32 r = a_r1[a];
    if (a_r1[a + 1] != r || (a > 0 && a_r1[a - 1] == r) || (a + 2 < m && a_r1[a + 2] == r)) return;       // not the first member of a pair
    const u32 x = a_sa[a], y = a_sa[a + 1];
    if (x > y) src[x] = y; else src[y] = x;         // the later position carries the pair: consecutive pairs of one repeat then sit at consecutive positions
}
__global__ __launch_bounds__(256) void sa_pair_resolve_kernel(const u8* __restrict__ text, size_t n, const u32* __restrict__ a_sa, const u32* __restrict__ a_pos,
                                                              const u32* __restrict__ a_r1, size_t m, const u32* __restrict__ len, u32* __restrict__ sa,
                                                              u32* __restrict__ rank, u8* __restrict__ keep) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    const u32 r = a_r1[a];
    const bool eq_prev = a > 0 && a_r1[a - 1] == r, eq_next = a + 1 < m && a_r1[a + 1] == r;
    const bool first = !eq_prev && eq_next && !(a + 2 < m && a_r1[a + 2] == r);
    const bool second = eq_prev && !eq_next && !(a >= 2 && a_r1[a - 2] == r);
    keep[a] = (first || second) ? 0 : 1;
    if (!first) return;
    const u32 x = a_sa[a], y = a_sa[a + 1];
    const u32 hi = x > y ? x : y, lo = x > y ? y : x;
    const u32 L = len[hi];
    // the unique sentinel ends the later suffix first: hi + L < n always; the byte behind the common extension decides
    const u8 ch = ((size_t)hi + L < n) ? text[(size_t)hi + L] : (u8)0, cl = ((size_t)lo + L < n) ? text[(size_t)lo + L] : (u8)0;
    const u32 small = (ch < cl) ? hi : lo, large = (ch < cl) ? lo : hi;
    const u32 p0 = a_pos[a];
    sa[p0] = small; sa[p0 + 1] = large;
    rank[small] = p0; rank[large] = p0 + 1;
}

// ---- star step (round 6): every group ordered against ONE of its members, from the text, in one pass ------------------------------------
// The pair step generalised.  r = the smallest position of a group; for a member p != r let l = lce(p, r) and (cp, cr) the bytes behind
// the common extension.  Members with cp < cr lie below r, the others above; below r a SMALLER l sorts first (p leaves r's path earlier,
// downwards), above r a LARGER l sorts first; equal l: by cp; equal (side, l, cp): the members agree on l + 1 >= h symbols and stay one
// group.  So one sort by (group, side, l', cp) orders every group up to such ties -- whatever its size -- and what the doubling rounds
// have left are the tie groups (copies of copies that continue alike).
// The lce values come cheap: inside a repeat the members at consecutive positions p, p + 1, ... have the representatives r, r + 1, ...
// (a CHAIN), and lce(p + k, r + k) = lce(p, r) - k exactly, with the same two bytes behind it.  One wave per chain head compares from
// the text (512 bytes per step, from the known common prefix h on) and then walks its chain, writing (lce, side, byte) for 64
// positions per step.  10^9 B of DNA with copied 4 KiB blocks: 400 M unresolved suffixes in 1.4 M chains.
constexpr u64 STAR_REP = ~0ull;
// longest common extension of the suffixes i and j from offset l on (l symbols are known to agree), a whole wave at 512 bytes per step
// (textds.hip wave_lcp)
__device__ __forceinline__ u32 wave_lcp_ext(const u8* __restrict__ text, size_t n, size_t i, size_t j, u32 l) {
    const size_t lim = n - (i > j ? i : j);                   // the unique sentinel ends the comparison before either suffix leaves the text
    const int lane = lane_id();
    for (;;) {
        const size_t off = (size_t)l + 8 * (size_t)lane;
        const bool full = off + 8 <= lim;
        u64 a = 0, b = 0;
        if (full) { __builtin_memcpy(&a, text + i + off, 8); __builtin_memcpy(&b, text + j + off, 8); }
        const u64 x = a ^ b;
        const u64 bad = __ballot(!full || x != 0);
        if (bad == 0) { l += 512; continue; }
        const int f = __builtin_ctzll(bad);                    // first lane with a mismatch or a word that sticks out of the text
        const u64 xf = __shfl(x, f);
        const bool ff = __shfl((int)full, f) != 0;
        l += 8 * (u32)f;
        if (ff) return l + ((u32)__builtin_ctzll(xf) >> 3);
        while ((size_t)l < lim && text[i + l] == text[j + l]) ++l;     // the last few bytes in front of the sentinel
        return l;
    }
}
__global__ __launch_bounds__(256) void sa_star_init_kernel(const u32* __restrict__ a_r1, size_t m, u32* __restrict__ gmin) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < m) gmin[a_r1[a]] = NONE32;
}
__global__ __launch_bounds__(256) void sa_star_min_kernel(const u32* __restrict__ a_sa, const u32* __restrict__ a_r1, size_t m, u32* __restrict__ gmin) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    // (the members of a group are neighbours in the list: a lane whose left neighbour belongs to the same group and holds a smaller
    //  position leaves the atomic to it -- most of a large group's atomics go away)
    const u32 r = a_r1[a], p = a_sa[a];
    if (a > 0 && (threadIdx.x & 63) != 0 && a_r1[a - 1] == r && a_sa[a - 1] < p) return;
    atomicMin(&gmin[r], p);
}
__global__ __launch_bounds__(256) void sa_star_mark_kernel(const u32* __restrict__ a_sa, const u32* __restrict__ a_r1, size_t m, const u32* __restrict__ gmin,
                                                           u32* __restrict__ src, u64* __restrict__ pk) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    const u32 p = a_sa[a], r = gmin[a_r1[a]];
    if (p != r) src[p] = r; else pk[p] = STAR_REP;
}
__global__ __launch_bounds__(256) void sa_star_headflag_kernel(const u32* __restrict__ src, size_t n, u8* __restrict__ cls) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 s = src[p];
    cls[p] = (s != NONE32 && (p == 0 || src[p - 1] == NONE32 || src[p - 1] + 1u != s)) ? 1 : 0;
}
// pk[q] = lce | side << 32 (0: below the representative, 2: above) | byte << 40 for every position q of the chain
__global__ __launch_bounds__(256) void sa_star_chain_kernel(const u8* __restrict__ text, size_t n, const u32* __restrict__ heads, u32 nheads,
                                                            const u32* __restrict__ src, u32 h0, u64* __restrict__ pk) {
    const u32 k = (u32)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (k >= nheads) return;
    const int lane = lane_id();
    const u32 p = heads[k], r = src[p];
    // (h0 symbols are known to agree; the suffix with the smaller remainder ends at the unique sentinel, so the comparison stops in the text)
    const u32 L = wave_lcp_ext(text, n, p, r, h0);
    const u8 cp = ((size_t)p + L < n) ? text[(size_t)p + L] : (u8)0, cr = ((size_t)r + L < n) ? text[(size_t)r + L] : (u8)0;
    const u64 code = ((u64)(cp < cr ? 0u : 2u) << 32) | ((u64)cp << 40);
    for (u32 k0 = 0;; k0 += 64) {
        const u32 d = k0 + (u32)lane;
        const size_t q = (size_t)p + d;
        const bool ok = d <= L && q < n && (d == 0 || src[q] == r + d);
        const u64 okm = __ballot(ok);
        const int run = (~okm == 0ull) ? 64 : __builtin_ctzll(~okm);          // the chain goes on for `run` more positions
        if (lane < run) pk[q] = (u64)(L - d) | code;
        if (run < 64) break;
    }
}
__global__ __launch_bounds__(256) void sa_star_keys_kernel(const u32* __restrict__ a_sa, const u32* __restrict__ a_r1, size_t m, int bn,
                                                           const u64* __restrict__ pk, u64* __restrict__ keys, u32* __restrict__ vals) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    const u32 p = a_sa[a];
    const u64 v = pk[p];
    const int W = bn - 10;                                       // bits of the length field (bn >= 26 here: lce up to 2^16 and more)
    const u32 maxl = (1u << W) - 1u;
    u32 second;
    if (v == STAR_REP) second = 1u << (bn - 2);
    else {
        u32 L = (u32)v;
        const u32 side = (u32)(v >> 32) & 3u;
        u32 ch = (u32)(v >> 40) & 0xFFu;
        // (beyond the field: such members tie -- the doubling rounds order them.  The clamped class takes the value maxl - 1 for itself:
        //  a member whose extension really is maxl - 1 long must not share it, its byte would order it against members it ties with)
        if (L >= maxl - 1u) { L = maxl - 1u; ch = 0; }
        const u32 f = side == 0 ? L : maxl - L;
        second = (side << (bn - 2)) | (f << 8) | ch;
    }
    keys[a] = ((u64)a_r1[a] << bn) | second;
    vals[a] = p;
}

// prefix doubling from h on: the active list (A_sa, A_pos, A_r1) holds the m unresolved suffixes, rank[] is up to date
// keys_ready: the first round's keys (keys[0] / vals[0]) have been built by the caller (star step) -- that round orders the groups by
// them and leaves h as it is
void doubling_rounds(Ctx& c, size_t n, int bn, u32* sa, u32* rank, SABufs& B, size_t m, u64 h, u32 h_tot[4], SAStats* st, bool keys_ready = false) {
    hipStream_t s = c.stream;
    u64** keys = B.keys; u32** vals = B.vals;
    // Sort of a round: groups of a handful of suffixes (random texts, DNA) are sorted inside 2048-element tiles and only the
    // groups that cross tile borders go through a global sort; where the unresolved suffixes sit in large groups (the frequent
    // words of a natural-language text: most of them would be border-crossing "open" groups anyway) ONE splitter sort of the whole
    // active list is cheaper (2e9 B English: -14 ms; DNA: +11 ms the other way round).
    auto big_groups = [&]() { return h_tot[2] != 0 && (u64)h_tot[1] >= 64ull * h_tot[2]; };
    int x;
    while (m > 0) {
        if (h >= n) throw HipError{hipErrorUnknown, "suffix_array: doubling did not converge", (int)__LINE__};
        const unsigned gm = cdiv(m, 256);
        const bool ready = keys_ready;
        keys_ready = false;
        if (!ready) {   // per element: read sa + r1 (8 B), gather rank[sa+h] (4 B), write key + value (12 B)
            Ctx::ProfScope prof(c, K_SA_BUILD_KEYS, (u64)m * 24);
            sa_build_keys_kernel<<<gm, 256, 0, s>>>(B.A_sa, B.A_r1, m, n, (u32)h, bn, rank, keys[0], vals[0]);
            LAUNCH_CHECK();
        }
        if ((c.sa_local_sort == 2 || (c.sa_local_sort == 1 && big_groups())) && c.ssort && splitter_sort_applicable(m)) {
            x = splitter_sort_pairs_u64(c, keys, vals, m, nullptr, nullptr);
            st->sorted_elems += m;
        } else if (c.sa_local_sort) {
            // local part: whole runs inside 2048-element tiles; global part: only the runs that cross a tile border
            u8* cls = (u8*)B.keep;                                 // scratch
            {
                Ctx::ProfScope prof(c, K_SA_LOCAL_SORT, (u64)m * 25);
                sa_local_sort_kernel<<<cdiv(m, 2048), 256, 0, s>>>(keys[0], vals[0], m, bn, cls);
                LAUNCH_CHECK();
            }
            u32* opos = B.B_sa;                                    // B_* are free until the compaction of this round
            select_by_class(c, cls, 1, m, nullptr, opos, nullptr, nullptr, B.d_total);
            const size_t mo = c.read(B.d_total);
            if (mo > n / 2) {                                      // a few giant runs: sort everything globally
                x = (c.ssort && splitter_sort_applicable(m)) ? splitter_sort_pairs_u64(c, keys, vals, m, nullptr, nullptr)
                                                              : radix_sort_pairs_u64(c, keys, vals, m, 0, 2 * bn);
                st->sorted_elems += m;
            } else {
            st->sorted_elems += mo;
            if (mo) {
                u64* ok2[2] = { keys[1], B.lkeys };
                u32* ov2[2] = { vals[1], B.lvals };
                select_by_class(c, cls, 1, m, vals[0], ov2[0], keys[0], ok2[0], B.d_total);
                const int y = (c.ssort && splitter_sort_applicable(mo)) ? splitter_sort_pairs_u64(c, ok2, ov2, mo, nullptr, nullptr)
                                                                        : radix_sort_pairs_u64(c, ok2, ov2, mo, 0, 2 * bn);
                sa_scatter_back_kernel<<<cdiv(mo, 256), 256, 0, s>>>(opos, ok2[y], ov2[y], mo, keys[0], vals[0]);
                LAUNCH_CHECK();
            }
            x = 0;
            }
        } else {
            x = radix_sort_pairs_u64(c, keys, vals, m, 0, 2 * bn);
            st->sorted_elems += m;
        }
        {
            // a large round scatters its ranks through the bucketed scatter (scratch: the other sort buffers, head / keep)
            const bool bucketed = c.bucket_scatter && m >= ((size_t)1 << 24);
            u32* nr = (u32*)keys[x ^ 1];
            const u32 tiles = cdiv(m, GR_TILE);
            HIP_TRY(hipMemsetAsync(B.gdesc, 0, (size_t)tiles * sizeof(u64), s));
            HIP_TRY(hipMemsetAsync(B.d_total, 0, 4 * sizeof(u32), s));
            {   // per element: key, value, position (16 B), sa + new rank (8 B), the kept elements (12 B each)
                Ctx::ProfScope prof(c, K_SA_RANK_SCATTER, (u64)m * 30);
                sa_groups_kernel<false, 0><<<tiles, 256, 0, s>>>(keys[x], vals[x], B.A_pos, m, bn, sa, rank, bucketed ? nr : nullptr, B.B_sa, B.B_pos, B.B_r1,
                                                                 B.gdesc, B.d_total, c.d_err, tiles, nullptr, NO_WIDE);
                LAUNCH_CHECK();
            }
            if (bucketed) bucketed_scatter_u32(c, vals[x], nr, m, rank, n, nr + m, vals[x ^ 1], B.head, B.keep);
        }
        c.read_n(B.d_total, h_tot, 4);
        m = h_tot[0];
        u32* t;
        t = B.A_sa; B.A_sa = B.B_sa; B.B_sa = t;
        t = B.A_pos; B.A_pos = B.B_pos; B.B_pos = t;
        t = B.A_r1; B.A_r1 = B.B_r1; B.B_r1 = t;
        if (!ready) h *= 2;
        st->rounds++;
    }
}

// The wide path (large texts): bit-packed 1- or 2-word keys through wsort.hip, then either
//   * text rounds -- the unresolved suffixes (a few per cent) are sorted by (group head, next 64 key bits read from the text) until every
//     group is a singleton; no rank array exists, the LCP of neighbouring suffixes falls out of the keys (lcp8), and ISA / Phi / PLCP are
//     produced afterwards by ONE fused scatter (fused.hip).  Result mode 1: sa final, lcp8 valid, isa NOT written;
//   * or, for texts with deep repeats (many unresolved suffixes, or the rounds do not get anywhere), the rank scatter + prefix doubling
//     of the classic path from the depth reached so far.  Result mode 0.
int build_suffix_array_wide(Ctx& c, const u8* text, size_t n, u32* sa, u32* isa, SAStats* st, SAExtra* ex, const CodeMap& cm, u32 sigma) {
    hipStream_t s = c.stream;
    int KW;
    WKeyGen g;
    wsort_make_keygen(c, text, n, sigma, cm.code, KW, g);
    const int b = g.b;
    const int per_word = 64 / b;
    WKeyGen g1 = g;                                            // the 64-bit keys of the text rounds
    g1.s = per_word > 64 ? 64 : per_word; g1.pad = 64 - g1.s * b;
    st->sym_bits = b; st->init_syms = g.s;
    // level 1 may have been done behind the upload already (api.hip): its record buffers sit at the top of the arena
    WPre* pre = (c.wpre && c.wpre->active && c.wpre->text == text && c.wpre->n == n && c.wpre->KW == KW) ? c.wpre : nullptr;
    struct TopGuard { Ctx& c; WPre* p; ~TopGuard() { if (p) { p->active = false; p->begun = false; c.arena.release_top(); } } } top_guard{c, pre};

    SABufs B;
    u64* K1[2]; u64* K2[2]; u32* V[2];
    u32* vspare = nullptr;                                     // the sorted positions land in sa[] itself: no copy afterwards
    if (pre) {
        K1[0] = pre->K1[0]; K1[1] = pre->K1[1]; V[0] = pre->V[0]; V[1] = pre->V[1];
        if (pre->KW == 2) { K2[0] = pre->K2[0]; K2[1] = pre->K2[1]; }
        else { K2[0] = c.arena.get<u64>(n); K2[1] = c.arena.get<u64>(n); }       // (the rounds sort two-word records whatever the initial width)
    } else {
        K1[0] = c.arena.get<u64>(n); K1[1] = c.arena.get<u64>(n);
        K2[0] = c.arena.get<u64>(n); K2[1] = c.arena.get<u64>(n);
        V[0] = c.arena.get<u32>(n); V[1] = c.arena.get<u32>(n);
        const int ri = wsort_result_index(c, n); vspare = V[ri]; V[ri] = sa;
    }
    B.keys[0] = K1[0]; B.keys[1] = K1[1];
    u8* flags = c.arena.get<u8>(n + 8);
    u8* lcp8 = (ex && ex->lcp8) ? ex->lcp8 : c.arena.get<u8>(n + 8);
    B.head = c.arena.get<u32>(n); B.keep = c.arena.get<u32>(n);
    B.A_sa = c.arena.get<u32>(n); B.A_pos = c.arena.get<u32>(n); B.A_r1 = c.arena.get<u32>(n);
    B.B_sa = c.arena.get<u32>(n); B.B_pos = c.arena.get<u32>(n); B.B_r1 = c.arena.get<u32>(n);
    B.d_total = c.arena.get<u32>(4);
    B.lkeys = c.arena.get<u64>(n / 2 + 2048);
    B.lvals = c.arena.get<u32>(n / 2 + 2048);
    B.gdesc = c.arena.get<u64>(cdiv(n, GR_TILE) + 1);
    const int bn = (int)bits_for(n - 1);

    WSortStats ws;
    int x = 0;
    // (with a sink the unresolved slots are counted below, by the first pass of their compaction: the sort need not count its non-heads)
    const bool count_here = ex != nullptr && c.wsort_rounds > 0;
    {
        struct Restore { Ctx& c; ~Restore() { c.wsort_count_nonheads = true; } } restore{c};
        c.wsort_count_nonheads = !count_here;
        if (pre) wsort_suffixes_pre(c, *pre, sa, flags, lcp8, &ws);
        else x = wsort_suffixes(c, KW, g, K1, K2, V, n, flags, lcp8, &ws);
    }
    st->sorted_elems += n;
    st->overlapped = pre ? 1u : 0u;
    st->wide_kw = (u32)KW; st->wide_nonheads = ws.nonheads;

    // the text rounds pay while the unresolved suffixes are few (<= n / 8; unresolved <= 2 * (slots that are not group heads))
    bool fast = count_here;
    u32 h_tot[4] = { 0, 0, 0, 0 };
    size_t m;
    if (!pre) {
        if (V[x] != sa) {                                      // (cannot happen: wsort_result_index names the buffer wsort returns)
            HIP_TRY(hipMemcpyAsync(sa, V[x], n * sizeof(u32), hipMemcpyDeviceToDevice, s));
        } else V[x] = vspare;                                  // sa[] is the result now; the rounds get the spare buffer to sort in
    }
    B.vals[0] = V[0]; B.vals[1] = V[1];                        // (scratch of the rank scatter / buffers of the doubling rounds)
    if (fast) {
        const u32 tiles = cdiv(n, FC_TILE);
        u32* tile_cnt = c.arena.get<u32>(tiles + 1);
        u32* tile_last = c.arena.get<u32>(tiles + 1);
        const int pf = c.prof_begin(K_SA_RANK_SCATTER, (u64)n);
        sa_flag_count_kernel<<<tiles, 256, 0, s>>>(flags, n, tile_cnt, tile_last);
        LAUNCH_CHECK();
        exclusive_sum_u32(c, tile_cnt, tile_cnt, tiles, B.d_total);
        c.prof_end(pf);
        m = c.read(B.d_total);
        st->wide_nonheads = m;                                 // (slots in groups of two and more: between the non-heads and twice their number)
        if (m > n / 8) fast = false;
        else {
            Ctx::ProfScope prof(c, K_SA_RANK_SCATTER, (u64)n + (u64)m * 16);
            inclusive_max_u32(c, tile_last, tile_last, tiles);
            sa_flag_compact_kernel<<<tiles, 256, 0, s>>>(flags, sa, n, tile_cnt, tile_last, B.A_sa, B.A_pos, B.A_r1);
            LAUNCH_CHECK();
        }
    }
    if (!fast) m = first_groups(c, n, bn, nullptr, sa, flags, sa, isa, B, true, h_tot);
    st->rounds = 1;
    u32 h = (u32)g.s;
    int text_rounds = 0;
    while (fast && m > 0) {
        if (text_rounds >= c.wsort_rounds || h + (u32)g1.s > 250u) { fast = false; break; }
        // (the list is in slot order, its first words -- the slots of the group heads -- are in order already: the round only has to order the
        //  members of every group by their next key bits, in place; option sa_seg_rounds.  2: the records are made by that pass itself --
        //  per element: position, head (8 B), one scattered text read, three record words (20 B))
        int y;
        if (c.sa_seg_rounds >= 2 && g1.s <= 64) {
            y = wsort_sorted_runs_from_text(c, B.A_sa, B.A_r1, m, g1, h, K1[0], K2[0], V[0], bn) ? 0 : wsort_records(c, K1, K2, V, m, bn, nullptr);
        } else {
            {
                Ctx::ProfScope prof(c, K_SA_BUILD_KEYS, (u64)m * 28);
                sa_round_keys_kernel<<<cdiv(m, 256), 256, 0, s>>>(B.A_sa, B.A_r1, m, g1, h, K1[0], K2[0], V[0]);
                LAUNCH_CHECK();
            }
            y = (c.sa_seg_rounds && wsort_sorted_runs(c, K1[0], K2[0], V[0], m, bn)) ? 0 : wsort_records(c, K1, K2, V, m, bn, nullptr);
        }
        st->sorted_elems += m;
        const u32 tiles = cdiv(m, GR_TILE);
        HIP_TRY(hipMemsetAsync(B.gdesc, 0, (size_t)tiles * sizeof(u64), s));
        HIP_TRY(hipMemsetAsync(B.d_total, 0, 4 * sizeof(u32), s));
        {
            Ctx::ProfScope prof(c, K_SA_RANK_SCATTER, (u64)m * 38);
            const WideRound wr = { K2[y], flags, lcp8, h, g.inv };
            sa_groups_kernel<false, 1><<<tiles, 256, 0, s>>>(K1[y], V[y], B.A_pos, m, bn, sa, nullptr, nullptr, B.B_sa, B.B_pos, B.B_r1,
                                                             B.gdesc, B.d_total, c.d_err, tiles, nullptr, wr);
            LAUNCH_CHECK();
        }
        c.read_n(B.d_total, h_tot, 4);
        m = h_tot[0];
        u32* t;
        t = B.A_sa; B.A_sa = B.B_sa; B.B_sa = t;
        t = B.A_pos; B.A_pos = B.B_pos; B.B_pos = t;
        t = B.A_r1; B.A_r1 = B.B_r1; B.B_r1 = t;
        h += (u32)g1.s;
        ++text_rounds;
        st->rounds++;
    }
    st->text_rounds = (u32)text_rounds;
    if (fast) {                                                // every suffix is a group of its own: sa is final
        if (!ex) throw HipError{hipErrorUnknown, "suffix_array: internal (fast path without a sink)", (int)__LINE__};
        return 1;
    }
    if (text_rounds > 0) {
        // the rounds gave up: ranks of the state reached so far (sa + head flags), then doubling from the common depth h
        m = first_groups(c, n, bn, nullptr, sa, flags, sa, isa, B, true, h_tot);
    }
    bool keys_ready = false;
    if (c.sa_stars && m >= n / 64 && m >= ((size_t)1 << 20) && bn >= 26 && h <= 0xFFFFu) {
        // star step: the keys of one ordering round from the text (see sa_star_* above); the round itself is the first one of doubling_rounds
        const size_t pm = c.arena.mark();
        u32* src = (u32*)K2[0];                            // (the second key words of the wide sort are not used any more: 2 x 8 n bytes)
        u32* gmin = src + n;
        u64* pk = K2[1];
        u8* cls = (u8*)B.keep;
        u32* heads = c.arena.get<u32>(m);
        fill_u32(c, src, n, NONE32);
        {
            Ctx::ProfScope prof(c, K_SA_BUILD_KEYS, (u64)m * 28);
            sa_star_init_kernel<<<cdiv(m, 256), 256, 0, s>>>(B.A_r1, m, gmin);
            LAUNCH_CHECK();
            sa_star_min_kernel<<<cdiv(m, 256), 256, 0, s>>>(B.A_sa, B.A_r1, m, gmin);
            LAUNCH_CHECK();
            sa_star_mark_kernel<<<cdiv(m, 256), 256, 0, s>>>(B.A_sa, B.A_r1, m, gmin, src, pk);
            LAUNCH_CHECK();
            sa_star_headflag_kernel<<<cdiv(n, 256), 256, 0, s>>>(src, n, cls);
            LAUNCH_CHECK();
        }
        select_by_class(c, cls, 1, n, nullptr, heads, nullptr, nullptr, B.d_total);
        const size_t nheads = c.read(B.d_total);
        if (nheads > m) throw HipError{hipErrorUnknown, "suffix_array: star step (more chain heads than members)", (int)__LINE__};
        // (a chain head costs a comparison over its whole common extension.  Copied blocks give chains of hundreds of positions; a
        //  periodic text -- a^N: every suffix of the run has the same representative -- gives one head per member, each with an
        //  extension as long as the run: the step is only taken where the chains are long)
        if (nheads <= m / 16) {
        {
            Ctx::ProfScope prof(c, K_PLCP, (u64)m * 12 + (u64)nheads * 64);
            if (nheads) sa_star_chain_kernel<<<cdiv(nheads * 64, 256), 256, 0, s>>>(text, n, heads, (u32)nheads, src, (u32)h, pk);
            LAUNCH_CHECK();
        }
        {
            Ctx::ProfScope prof(c, K_SA_BUILD_KEYS, (u64)m * 28);
            sa_star_keys_kernel<<<cdiv(m, 256), 256, 0, s>>>(B.A_sa, B.A_r1, m, bn, pk, B.keys[0], B.vals[0]);
            LAUNCH_CHECK();
        }
        st->star_chains = (u64)nheads;
        keys_ready = true;
        }
        c.arena.release(pm);
    }
    if (!keys_ready && c.sa_pairs && m >= n / 64 && m >= ((size_t)1 << 20)) {
        // pairs first: one pass over the text instead of log(repeat length / h) rounds over them
        const size_t pm = c.arena.mark();
        u32* src = (u32*)K2[0];                            // (the second key words of the wide sort are not used any more: 8 n bytes)
        u32* len = src + n;
        u32* d_mx = c.arena.get<u32>(1);
        u8* keep = (u8*)B.keep;
        fill_u32(c, src, n, NONE32);
        {
            Ctx::ProfScope prof(c, K_SA_BUILD_KEYS, (u64)m * 12);
            sa_pair_mark_kernel<<<cdiv(m, 256), 256, 0, s>>>(B.A_sa, B.A_r1, m, src);
            LAUNCH_CHECK();
        }
        build_lce_with_carry(c, text, n, src, len, d_mx);
        {
            Ctx::ProfScope prof(c, K_SA_RANK_SCATTER, (u64)m * 30);
            sa_pair_resolve_kernel<<<cdiv(m, 256), 256, 0, s>>>(text, n, B.A_sa, B.A_pos, B.A_r1, m, len, sa, isa, keep);
            LAUNCH_CHECK();
        }
        select_by_class(c, keep, 1, m, B.A_sa, B.B_sa, nullptr, nullptr, B.d_total);
        select_by_class(c, keep, 1, m, B.A_pos, B.B_pos, nullptr, nullptr, B.d_total);
        select_by_class(c, keep, 1, m, B.A_r1, B.B_r1, nullptr, nullptr, B.d_total);
        const size_t m2 = c.read(B.d_total);
        u32* t;
        t = B.A_sa; B.A_sa = B.B_sa; B.B_sa = t;
        t = B.A_pos; B.A_pos = B.B_pos; B.B_pos = t;
        t = B.A_r1; B.A_r1 = B.B_r1; B.B_r1 = t;
        st->pair_resolved = (u64)(m - m2);
        m = m2;
        c.arena.release(pm);
    }
    doubling_rounds(c, n, bn, sa, isa, B, m, (u64)h, h_tot, st, keys_ready);
    return 0;
}
}  // namespace

void build_suffix_array(Ctx& c, const u8* text, size_t n, u32* sa, u32* isa, SAStats* st, SAExtra* ex) {
    SAStats local;
    if (!st) st = &local;
    *st = SAStats();
    if (ex) ex->mode = 0;
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();

    // --- dense symbol codes -----------------------------------------------------------------------
    if (!(c.hist_ptr == text && c.hist_n == n)) {            // (a pipeline call has usually counted the bytes already: sentinel check)
        u32* d_hist = c.arena.get<u32>(256);
        HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), s));
        text_histogram_add(c, text, n, d_hist);
        text_histogram_finish(c, text, n, d_hist);
    }
    const u32* h_hist = c.hist_cache;
    CodeMap cm;
    u32 sigma = 0;
    for (int i = 0; i < 256; ++i) { cm.code[i] = (u8)sigma; if (h_hist[i]) ++sigma; }
    // a byte that does not occur keeps the code of the next present byte; irrelevant (never looked up)

    if (wsort_applicable(c, n)) {
        const int mode = build_suffix_array_wide(c, text, n, sa, isa, st, ex, cm, sigma);
        if (ex) ex->mode = mode;
        c.arena.release(mark);
        return;
    }

    const int b = (int)bits_for(sigma > 1 ? sigma - 1 : 1);
    const u32 base = sigma > 2 ? sigma : 2;
    int k = 0;                                               // largest k with base^k <= 2^64, at most 32 (LDS halo of the key kernel)
    int key_bits = 64;
    {
        unsigned __int128 pw = 1;
        while (k < 32 && pw * base <= ((unsigned __int128)1 << 64)) { pw *= base; ++k; }
        if (c.sa_init_syms >= 1 && c.sa_init_syms < k) { k = c.sa_init_syms; pw = 1; for (int i = 0; i < k; ++i) pw *= base; }   // tuning knob
        key_bits = (pw > ((unsigned __int128)1 << 63)) ? 64 : (int)bits_for((u64)(pw - 1));
    }
    st->sym_bits = b; st->init_syms = k;

    // --- buffers ----------------------------------------------------------------------------------
    SABufs B;
    B.keys[0] = c.arena.get<u64>(n); B.keys[1] = c.arena.get<u64>(n);
    B.vals[0] = c.arena.get<u32>(n); B.vals[1] = c.arena.get<u32>(n);
    B.head = c.arena.get<u32>(n);
    B.keep = c.arena.get<u32>(n);
    B.A_sa = c.arena.get<u32>(n); B.A_pos = c.arena.get<u32>(n); B.A_r1 = c.arena.get<u32>(n);
    B.B_sa = c.arena.get<u32>(n); B.B_pos = c.arena.get<u32>(n); B.B_r1 = c.arena.get<u32>(n);
    B.d_total = c.arena.get<u32>(4);                // [0] unresolved elements after a bookkeeping pass; [1], [2]: sampled elements / groups
    B.lkeys = c.arena.get<u64>(n / 2 + 2048);       // second buffers of the "open run" sort (at most half of the list...)
    B.lvals = c.arena.get<u32>(n / 2 + 2048);
    u64** keys = B.keys; u32** vals = B.vals;
    u32* rank = isa;

    // --- initial sort by the first k symbols ------------------------------------------------------
    int chunk = 1;                                           // largest chunk with base^chunk < 2^32
    { u64 pw = base; while (pw * base < (1ull << 32)) { pw *= base; ++chunk; } }
    int x;
    if (c.sa_fused_init && k <= 32) {                        // pass 0 of the sort computes the keys from the text
        TextKeyGen g;
        g.text = text; g.n = n; g.sigma = base; g.k = k; g.chunk = chunk;
        g.top = 1; for (int i = 1; i < k; ++i) g.top *= base;
        memcpy(g.code, cm.code, 256);
        if (c.ssort && splitter_sort_applicable(n)) x = splitter_sort_pairs_u64(c, keys, vals, n, &g, nullptr);   // 2-3 partition levels + leaf sort
        else x = radix_sort_text_keys_u64(c, g, keys, vals, key_bits);
    } else {
        sa_init_keys_kernel<<<cdiv(n, 1024), 256, 0, s>>>(text, n, cm, base, k, chunk, keys[0], vals[0]);
        LAUNCH_CHECK();
        x = radix_sort_pairs_u64(c, keys, vals, n, 0, key_bits);
    }
    st->sorted_elems += n;
    // groups of a few suffixes are ordered by their next k symbols straight from the text (see sa_refine_kernel)
    const u8* hflags = nullptr;
    if (c.sa_refine && k <= 32 && n >= ((size_t)1 << 16)) {
        RefineGen rg;
        rg.text = text; rg.n = n; rg.sigma = base; rg.k = k; rg.chunk = chunk;
        memcpy(rg.code, cm.code, 256);
        u8* fl = (u8*)B.keep;                                // free until the rounds use it as class bytes
        Ctx::ProfScope prof(c, K_SA_BUILD_KEYS, (u64)n * 17 + (u64)n * 8);     // key + position in, position + flag out; ~0.4 scattered text reads per element
        sa_refine_kernel<<<cdiv(n, RF_TILE), 256, 0, s>>>(keys[x], vals[x], n, rg, fl, sa);       // the refined order goes straight into sa[]
        LAUNCH_CHECK();
        hflags = fl;
    }
    const int bn = (int)bits_for(n - 1);
    B.gdesc = c.arena.get<u64>(cdiv(n, GR_TILE) + 1);       // look-back descriptors of the group kernel
    u32 h_tot[4];
    // (the rank scatter of first_groups uses keys[1] / vals[1] as scratch: the sorted pairs must be in the [0] buffers by then)
    if (x != 0) { u64* tk = keys[0]; keys[0] = keys[1]; keys[1] = tk; u32* tv = vals[0]; vals[0] = vals[1]; vals[1] = tv; x = 0; }
    const size_t m = first_groups(c, n, bn, keys[0], hflags ? sa : vals[0], hflags, sa, rank, B, true, h_tot);
    st->rounds = 1;
    doubling_rounds(c, n, bn, sa, rank, B, m, (u64)k, h_tot, st);
    c.arena.release(mark);
}

}  // namespace tdc
