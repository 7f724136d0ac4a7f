// ssort.hip -- splitter-partition sort of (u64 key, u32 value) pairs (gfx950, wave64).
//
// The LSD radix sort of prim.hip moves every pair once per 8 key bits: eight passes for the 64-bit keys of the suffix
// array's initial sort, whatever the keys look like.  Text keys are heavily skewed (a 13-symbol window of an English-like
// text has far less than 64 bits of entropy), so a most-significant-digit split on key BITS leaves most pairs in huge
// buckets.  This sort splits on key RANKS instead:
//
//   1. sample: S = os * (NS + 1) keys at pseudo-random positions, sorted with the LSD sort; every os-th one becomes a
//      splitter sp[0 .. NS) (NS + 1 = F_1 * ... * F_L, the product of the level fan-outs).
//   2. L partition levels (L <= 3).  lb(x) = #splitters < x.  Level l moves x into the sub-segment given by the l-th digit of
//      lb(x) in the mixed radix (F_1, ..., F_L); the last level also splits off the keys EQUAL to a splitter
//      (digit = 2 * (lb mod F_L) + [x == sp[lb]]), so a heavy key -- a key that occurs more than n / (NS + 1) times ends
//      up in the sample several times -- lands in a leaf of its own that needs no sorting at all.  One level = a count pass
//      and a scatter pass over tiles of 4096 pairs that never straddle a segment; the digit of a key is a binary search
//      among the <= 255 splitters of its segment, held in LDS.
//   3. leaves: the array is now ordered leaf by leaf, range leaves hold ~n / (NS + 1) keys (binomial spread; the
//      oversampling factor os keeps them below the capacity of one workgroup with overwhelming probability).  Consecutive
//      small leaves are packed into units of <= 8192 pairs; one workgroup sorts a unit entirely in LDS (LSD radix on the
//      bits in which the unit's keys differ).  A leaf above 8192 pairs that is not an equality leaf goes through the LSD
//      sort (never observed; kept for correctness).
//
// Traffic per pair: L * (8 + 24) + 24 bytes instead of 8 * (8 + 24); level 1 of the suffix array's sort computes its keys
// from the text (no key array is read).  The sort is NOT stable (nothing downstream needs it: pairs with equal keys form one
// group of the prefix doubling).
#include "prim.hpp"

#include <vector>

namespace tdc {

constexpr int SS_TILE = 4096;        // pairs per partition tile: 256 threads x 16
constexpr int SS_ITEMS = 16;
constexpr int SS_SMALL = 4096;       // leaves up to this size are packed into units
constexpr int SS_UNIT_MAX = 8192;    // capacity of the leaf sort (one workgroup of 512 threads x 16)
constexpr int SS_GEN_HALO = 32;

struct SSLevel {
    const u64* keys_in; const u32* vals_in;
    u64* keys_out; u32* vals_out;
    u32* counts;                 // [rows][D]: per-tile digit counts, then absolute offsets
    const u32* blk_seg;          // [blocks] segment of every row block
    const u32* blk_start;        // [nseg + 1] first row block of every segment
    const u32* seg_start;        // [nseg + 1] first pair of every segment
    const u64* sp;               // [NS + 1] splitters, sp[NS] = ~0
    u16* digits;                 // [n] the level's digit of every pair of the input, written by the count pass for the scatter pass
    u32 nseg, F, stride, R, D, per_xcd;
};

// row (tile slot) -> segment, first pair, number of pairs; false: behind the last real row block
__device__ __forceinline__ bool ss_row(const SSLevel& P, u32 row, u32& s, size_t& base, u32& cnt) {
    const u32 blk = row / P.R;
    if (blk >= P.blk_start[P.nseg]) return false;
    s = P.blk_seg[blk];
    const u64 t = (u64)(blk - P.blk_start[s]) * P.R + row % P.R;
    const u32 s0 = P.seg_start[s], s1 = P.seg_start[s + 1];
    const u64 off = t * SS_TILE;
    base = s0; cnt = 0;
    if (off >= (u64)(s1 - s0)) return true;              // padding row of the segment's last row block
    base = (size_t)s0 + off;
    const u64 left = (u64)(s1 - s0) - off;
    cnt = left < SS_TILE ? (u32)left : (u32)SS_TILE;
    return true;
}

// the (<= 256) splitters that matter inside segment s, padded with ~0
template <bool LAST>
__device__ __forceinline__ void ss_load_splitters(const SSLevel& P, u32 s, u64* spl) {
    for (u32 i = threadIdx.x; i < 256; i += blockDim.x) {
        u64 v = ~0ull;
        if (LAST) { if (i < P.F) v = P.sp[(size_t)s * P.F + i]; }
        else if (i + 1 < P.F) v = P.sp[((size_t)s * P.F + i + 1) * P.stride - 1];
        spl[i] = v;
    }
}

// #splitters < x among spl[0 .. 255) (branch-free binary search), then the digit
template <bool LAST>
__device__ __forceinline__ u32 ss_digit(const u64* spl, u64 x) {
    u32 lo = 0;
#pragma unroll
    for (u32 step = 128; step >= 1; step >>= 1) lo += (spl[lo + step - 1] < x) ? step : 0u;
    if (LAST) return 2 * lo + (spl[lo] == x ? 1u : 0u);
    return lo;
}

// recoded bytes of one tile (+ halo) into LDS (see gen_stage_tile in prim.hip)
__device__ __forceinline__ void ss_gen_stage(const TextKeyGen& g, size_t t0, const u8* __restrict__ code, u8* __restrict__ sy) {
    const size_t p = t0 + (size_t)threadIdx.x * 16;
    u8 b[16];
    if (p + 16 <= g.n && (((size_t)g.text) & 15) == 0) {
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
    if (threadIdx.x < SS_GEN_HALO) {
        const size_t q = t0 + SS_TILE + threadIdx.x;
        sy[SS_TILE + threadIdx.x] = (q < g.n) ? code[g.text[q]] : (u8)0;
    }
}
__device__ __forceinline__ u64 ss_gen_key(const TextKeyGen& g, const u8* sy, int lb) {
    u64 key = 0;
    for (int j0 = 0; j0 < g.k; j0 += g.chunk) {
        const int len = (g.k - j0 < g.chunk) ? g.k - j0 : g.chunk;
        u32 acc = 0, scale = 1;
        for (int t = 0; t < len; ++t) { acc = acc * g.sigma + sy[lb + j0 + t]; scale *= g.sigma; }
        key = (j0 == 0) ? (u64)acc : key * scale + acc;
    }
    return key;
}

// ---- sampling -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t ss_sample_pos(u32 i, size_t n) {
    const u64 h = ((u64)i + 1) * 0x9E3779B97F4A7C15ull;
    return (size_t)(((unsigned __int128)(h ^ (h >> 29)) * n) >> 64);
}
template <bool GEN>
__global__ __launch_bounds__(256) void ss_sample_kernel(const u64* __restrict__ keys, TextKeyGen g, size_t n, u32 S, u64* __restrict__ out,
                                                        u32* __restrict__ dummy) {
    __shared__ u8 code[256];
    if (GEN) { code[threadIdx.x] = g.code[threadIdx.x]; __syncthreads(); }
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const size_t p = ss_sample_pos(i, n);
    u64 key;
    if (GEN) {
        key = 0;
        for (int j0 = 0; j0 < g.k; j0 += g.chunk) {
            const int len = (g.k - j0 < g.chunk) ? g.k - j0 : g.chunk;
            u32 acc = 0, scale = 1;
            for (int t = 0; t < len; ++t) {
                const size_t q = p + j0 + t;
                acc = acc * g.sigma + ((q < n) ? code[g.text[q]] : 0u);
                scale *= g.sigma;
            }
            key = (j0 == 0) ? (u64)acc : key * scale + acc;
        }
    } else key = keys[p];
    out[i] = key;
    dummy[i] = i;
}
__global__ void ss_pick_kernel(const u64* __restrict__ sorted, u32 NS, u32 os, u64* __restrict__ sp) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NS) sp[i] = sorted[(size_t)(i + 1) * os - 1];
    else if (i == NS) sp[i] = ~0ull;
}

// ---- row-block tables ------------------------------------------------------------------------------------------------------------
__global__ void ss_nblk_kernel(const u32* __restrict__ seg_start, u32 nseg, u32 R, u32* __restrict__ nblk) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nseg) return;
    if (s == nseg) { nblk[s] = 0; return; }
    const u64 size = seg_start[s + 1] - seg_start[s];
    nblk[s] = (u32)((size + (u64)SS_TILE * R - 1) / ((u64)SS_TILE * R));
}
__global__ void ss_blkseg_kernel(const u32* __restrict__ blk_start, u32 nseg, u32* __restrict__ blk_seg) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    for (u32 b = blk_start[s]; b < blk_start[s + 1]; ++b) blk_seg[b] = s;
}

// ---- count --------------------------------------------------------------------------------------------------------------------
template <bool GEN, bool LAST>
__global__ __launch_bounds__(256) void ss_count_kernel(SSLevel P, TextKeyGen g, u32 rows) {
    __shared__ u32 hist[512];
    __shared__ u64 spl[256];
    __shared__ u8 code[256];
    __shared__ __align__(16) u8 sy[GEN ? SS_TILE + SS_GEN_HALO : 16];
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    u32 s, cnt; size_t base;
    if (!ss_row(P, row, s, base, cnt)) return;
    for (u32 i = threadIdx.x; i < P.D; i += 256) hist[i] = 0;
    if (cnt == 0) {
        for (u32 i = threadIdx.x; i < P.D; i += 256) P.counts[(size_t)row * P.D + i] = 0;
        return;
    }
    ss_load_splitters<LAST>(P, s, spl);
    if (GEN) code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    if (GEN) { ss_gen_stage(g, base, code, sy); __syncthreads(); }
    if (GEN) {
        const int lb = wave_id() * (64 * SS_ITEMS) + lane_id() * SS_ITEMS;      // a lane owns 16 consecutive positions
        u64 key = ss_gen_key(g, sy, lb);
        u32 pk[SS_ITEMS / 2];
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) {
            const bool valid = (u32)(lb + j) < cnt;
            const u32 d = valid ? ss_digit<LAST>(spl, key) : 0u;
            key = (key - (u64)sy[lb + j] * g.top) * g.sigma + sy[lb + j + g.k];
            pk[j >> 1] = (j & 1) ? (pk[j >> 1] | (d << 16)) : d;
            const u32 d0 = __builtin_amdgcn_readfirstlane(d);
            if (__all(valid && d == d0)) { if (lane_id() == 0) atomicAdd(&hist[d0], 64u); }
            else if (valid) atomicAdd(&hist[d], 1u);
        }
        // the lane's 16 digits: 32 bytes, 32-byte aligned (tiles start at multiples of 4096); the array is padded to whole tiles
        uint4* dp = (uint4*)(P.digits + base + lb);
        dp[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dp[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    } else {
        const u32 lb = wave_id() * (64 * SS_ITEMS) + lane_id();
        const u64* kp = P.keys_in + base + lb;
        u16* dgp = P.digits + base + lb;
        u64 kk[SS_ITEMS];
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) kk[j] = (lb + (u32)j * 64 < cnt) ? kp[j * 64] : 0ull;
#pragma unroll 4
        for (int j = 0; j < SS_ITEMS; ++j) {
            const u32 e = lb + (u32)j * 64;
            const bool valid = e < cnt;
            const u32 d = valid ? ss_digit<LAST>(spl, kk[j]) : 0u;
            if (valid) dgp[j * 64] = (u16)d;
            const u32 d0 = __builtin_amdgcn_readfirstlane(d);
            if (__all(valid && d == d0)) { if (lane_id() == 0) atomicAdd(&hist[d0], 64u); }
            else if (valid) atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < P.D; i += 256) P.counts[(size_t)row * P.D + i] = hist[i];
}

// ---- offsets: counts[row][d] -> first output slot of the (row, d) run ----------------------------------------------------------
// (a) sum of every row block, (b) per segment: prefix over its row blocks + start of every digit (= the next level's segment
// starts), (c) running prefix inside the row block
__global__ __launch_bounds__(256) void ss_blocksum_kernel(const u32* __restrict__ counts, const u32* __restrict__ blk_start, u32 nseg,
                                                          u32 R, u32 D, u32* __restrict__ bs) {
    const u32 b = blockIdx.x;
    if (b >= blk_start[nseg]) return;
    for (u32 d = threadIdx.x; d < D; d += 256) {
        u32 acc = 0;
        for (u32 r = 0; r < R; ++r) acc += counts[((size_t)b * R + r) * D + d];
        bs[(size_t)b * D + d] = acc;
    }
}
// one workgroup of 1024 threads per segment; thread (g, d): digit d (and d + 256 when D = 512 ... handled by a loop), row-block
// group g of G = 1024 / 256 = 4
__global__ __launch_bounds__(1024) void ss_segbase_kernel(u32* __restrict__ bs, const u32* __restrict__ blk_start, const u32* __restrict__ seg_start,
                                                          u32 D, u32* __restrict__ next_start) {
    __shared__ u32 part[4][2048];
    __shared__ u32 dstart[2048];
    __shared__ u32 wsum[16];
    const u32 s = blockIdx.x;
    const u32 b0 = blk_start[s], b1 = blk_start[s + 1];
    const u32 t = threadIdx.x & 255u, g = threadIdx.x >> 8;
    const u32 nb = b1 - b0, per = (nb + 3) / 4;
    const u32 lo = b0 + (g * per < nb ? g * per : nb);
    const u32 hi = (lo + per < b1) ? lo + per : b1;
    for (u32 d = t; d < D; d += 256) {
        u32 acc = 0;
        u32 b = lo;
        for (; b + 8 <= hi; b += 8) {
            u32 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = bs[(size_t)(b + i) * D + d];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
        for (; b < hi; ++b) acc += bs[(size_t)b * D + d];
        part[g][d] = acc;
    }
    __syncthreads();
    // exclusive scan of the D digit totals (D <= 2048): thread i owns the digits 2i and 2i + 1
    const u32 d0 = 2 * threadIdx.x, d1 = d0 + 1;
    const u32 tot0 = d0 < D ? part[0][d0] + part[1][d0] + part[2][d0] + part[3][d0] : 0u;
    const u32 tot1 = d1 < D ? part[0][d1] + part[1][d1] + part[2][d1] + part[3][d1] : 0u;
    const u32 tot = tot0 + tot1;
    u32 inc = wave_inclusive_sum(tot);
    if (lane_id() == 63) wsum[wave_id()] = inc;
    __syncthreads();
    {
        u32 run = 0;
        for (int w = 0; w < wave_id(); ++w) run += wsum[w];
        const u32 st = seg_start[s] + run + inc - tot;
        if (d0 < D) { dstart[d0] = st; next_start[(size_t)s * D + d0] = st; }
        if (d1 < D) { dstart[d1] = st + tot0; next_start[(size_t)s * D + d1] = st + tot0; }
    }
    __syncthreads();
    for (u32 d = t; d < D; d += 256) {
        u32 run = dstart[d];
        for (u32 k = 0; k < g; ++k) run += part[k][d];
        u32 b = lo;
        for (; b + 8 <= hi; b += 8) {
            u32 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = bs[(size_t)(b + i) * D + d];
#pragma unroll
            for (int i = 0; i < 8; ++i) { bs[(size_t)(b + i) * D + d] = run; run += v[i]; }
        }
        for (; b < hi; ++b) { const u32 v = bs[(size_t)b * D + d]; bs[(size_t)b * D + d] = run; run += v; }
    }
}
__global__ __launch_bounds__(256) void ss_apply_kernel(u32* __restrict__ counts, const u32* __restrict__ blk_start, u32 nseg, u32 R, u32 D,
                                                       const u32* __restrict__ bs) {
    const u32 b = blockIdx.x;
    if (b >= blk_start[nseg]) return;
    for (u32 d = threadIdx.x; d < D; d += 256) {
        u32 run = bs[(size_t)b * D + d];
        for (u32 r = 0; r < R; ++r) {
            const size_t i = ((size_t)b * R + r) * D + d;
            const u32 v = counts[i];
            counts[i] = run;
            run += v;
        }
    }
}
__global__ void ss_set_word_kernel(u32* p, u32 v) { *p = v; }

// ---- scatter ------------------------------------------------------------------------------------------------------------------
// rs_scatter_lds_kernel of prim.hip with a splitter digit: ranks inside the tile from a wave-level match (NB ballots per key)
// plus per-wave counters; the tile is written to LDS in digit order first, so that consecutive lanes write consecutive pairs of
// a digit's run.  The splitters (and, for GEN, the tile's recoded bytes) live in the staging buffer until the digits are known.
template <bool GEN, bool LAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ss_scatter_kernel(SSLevel P, TextKeyGen g, u32 rows) {
    constexpr int DMAX = LAST ? 512 : 256;
    __shared__ u32 tcnt[DMAX];            // keys of the tile per digit, then the start of the digit's run inside the sorted tile
    __shared__ u32 gbase[DMAX];
    __shared__ __align__(16) u64 stage[SS_TILE];
    __shared__ u32 scan_sm[5];
    __shared__ u8 code[GEN ? 256 : 4];
    const int lane = lane_id(), w = wave_id();
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    u32 s, cnt; size_t base;
    if (!ss_row(P, row, s, base, cnt) || cnt == 0) return;
    u8* sy = (u8*)stage;                                    // GEN: recoded bytes of the tile, 4 KB + halo
    u16* stage_d = (u16*)(stage + 2048);                    // digits of the staged values: second half of the buffer
    u32* stage32 = (u32*)stage;
    for (int i = threadIdx.x; i < DMAX; i += 256) tcnt[i] = 0;
    if (GEN) code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    if (GEN) { ss_gen_stage(g, base, code, sy); __syncthreads(); }

    u64 k[SS_ITEMS];
    u32 v[SS_ITEMS];
    u32 ld[SS_ITEMS];                                       // digit << 16 | rank inside the tile's digit run, later position in the tile
    const u32 lbs = GEN ? (u32)(w * (64 * SS_ITEMS) + lane * SS_ITEMS) : (u32)(w * (64 * SS_ITEMS) + lane);
    u64 gkey = 0;
    u32 pk[SS_ITEMS / 2];                                   // GEN: the lane's 16 digits as written by the count pass
    if (GEN) {
        gkey = ss_gen_key(g, sy, (int)lbs);
        const uint4* dp = (const uint4*)(P.digits + base + lbs);
        const uint4 q0 = dp[0], q1 = dp[1];
        pk[0] = q0.x; pk[1] = q0.y; pk[2] = q0.z; pk[3] = q0.w; pk[4] = q1.x; pk[5] = q1.y; pk[6] = q1.z; pk[7] = q1.w;
    }
    const u64* kp = GEN ? nullptr : P.keys_in + base + lbs;
    const u32* vp = GEN ? nullptr : P.vals_in + base + lbs;
    const u16* dgp = P.digits + base + lbs;
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {                    // all global loads first: they overlap, the ranking below is a chain of LDS operations
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        const bool valid = e < cnt;
        if (GEN) {
            k[j] = gkey; v[j] = (u32)(base + e);
            gkey = (gkey - (u64)sy[lbs + j] * g.top) * g.sigma + sy[lbs + j + g.k];
            ld[j] = (pk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        } else {
            k[j] = valid ? kp[j * 64] : 0ull;                   // one address per array, the row as an immediate offset
            v[j] = valid ? vp[j * 64] : 0u;
            ld[j] = valid ? (u32)dgp[j * 64] : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        const bool valid = e < cnt;
        const u32 d = valid ? ld[j] : 0u;
        // a partition need not be stable: the rank inside the digit's run is whatever an LDS atomic hands out (a row whose
        // lanes all hold the same digit -- heavy keys, high levels -- takes 64 ranks with one atomic)
        u32 rank = 0;
        {   // lanes that share a digit with many others (heavy keys fill a third of some tiles) would serialise their atomics:
            // up to two big groups of the row are peeled off with one atomic each, the rest takes one atomic per lane
            u64 rem = __ballot(valid);
            bool done = !valid;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (!rem) break;
                const int lead = __ffsll((long long)rem) - 1;
                const u32 dl = __shfl(d, lead, 64);
                const u64 grp = __ballot(!done && d == dl);
                const u32 c = (u32)__popcll(grp);
                if (c < 8) break;
                u32 b0 = 0;
                if (lane == lead) b0 = atomicAdd(&tcnt[dl], c);
                b0 = __shfl(b0, lead, 64);
                if ((grp >> lane) & 1ull) { rank = b0 + (u32)__popcll(grp & ((1ull << lane) - 1)); done = true; }
                rem &= ~grp;
            }
            if (!done) rank = atomicAdd(&tcnt[d], 1u);
        }
        ld[j] = (d << 16) | rank;
    }
    __syncthreads();
    {   // digit runs inside the sorted tile (exclusive scan over the digits), per-wave starts, global base
        const u32 t = threadIdx.x;
        u32 tot[DMAX / 256], sum = 0;
#pragma unroll
        for (int q = 0; q < DMAX / 256; ++q) {               // thread t owns the digits t * (DMAX / 256) + q
            tot[q] = tcnt[t * (DMAX / 256) + q];
            sum += tot[q];
        }
        u32 total;
        u32 start = block_exclusive_sum<u32, 4>(sum, scan_sm, total);
#pragma unroll
        for (int q = 0; q < DMAX / 256; ++q) {
            const u32 d = t * (DMAX / 256) + q;
            tcnt[d] = start;
            gbase[d] = (d < P.D ? P.counts[(size_t)row * P.D + d] : 0u) - start;
            start += tot[q];
        }
    }
    __syncthreads();                                       // splitters / text bytes are dead from here on
    u32 dst[SS_ITEMS];
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        const u32 d = ld[j] >> 16;
        const u32 pos = (ld[j] & 0xFFFFu) + tcnt[d];           // position inside the sorted tile
        ld[j] = pos;
        if (e < cnt) { stage32[pos] = v[j]; stage_d[pos] = (u16)d; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; ++r) {
        const u32 sp = (u32)r * 256 + threadIdx.x;
        dst[r] = 0xFFFFFFFFu;
        if (sp < cnt) {
            dst[r] = gbase[stage_d[sp]] + sp;
            P.vals_out[dst[r]] = stage32[sp];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        if (e < cnt) stage[ld[j]] = k[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; ++r) {
        const u32 sp = (u32)r * 256 + threadIdx.x;
        if (dst[r] != 0xFFFFFFFFu) P.keys_out[dst[r]] = stage[sp];
    }
}

// ---- scatter, stable variant: ranks from the wave-level LDS match (used for the LAST level, whose equality digits are skewed
//      by construction: a heavy key fills a third of some tiles, and same-address LDS atomics serialise) ---------------------------
// rs_scatter_lds_kernel of prim.hip with a splitter digit: ranks inside the tile from a wave-level match (NB ballots per key)
// plus per-wave counters; the tile is written to LDS in digit order first, so that consecutive lanes write consecutive pairs of
// a digit's run.  The splitters (and, for GEN, the tile's recoded bytes) live in the staging buffer until the digits are known.
template <bool GEN, bool LAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ss_scatter_stable_kernel(SSLevel P, TextKeyGen g, u32 rows) {
    constexpr int DMAX = LAST ? 512 : 256;
    __shared__ __align__(16) u16 wcnt[4][DMAX];
    __shared__ u32 gbase[DMAX];
    __shared__ __align__(16) u64 stage[SS_TILE];
    __shared__ u32 scan_sm[5];
    __shared__ u8 code[GEN ? 256 : 4];
    const int lane = lane_id(), w = wave_id();
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    u32 s, cnt; size_t base;
    if (!ss_row(P, row, s, base, cnt) || cnt == 0) return;
    u8* sy = (u8*)stage;                                    // GEN: recoded bytes of the tile, 4 KB + halo
    u16* stage_d = (u16*)(stage + 2048);                    // digits of the staged values: second half of the buffer
    u32* stage32 = (u32*)stage;
    unsigned long long* M = (unsigned long long*)stage + 1024 + w * DMAX;    // lane-mask tables of the LDS match: bytes 8 K .. 24 K
    for (int i = threadIdx.x; i < 4 * DMAX; i += 256) { (&wcnt[0][0])[i] = 0; ((unsigned long long*)stage)[1024 + i] = 0; }
    if (GEN) code[threadIdx.x] = g.code[threadIdx.x];
    __syncthreads();
    if (GEN) { ss_gen_stage(g, base, code, sy); __syncthreads(); }

    u64 k[SS_ITEMS];
    u32 v[SS_ITEMS];
    u32 ld[SS_ITEMS];                                       // digit << 16 | rank inside the (wave, digit) run, later position in the tile
    u16* mycnt = wcnt[w];
    const u64 lanebit = 1ull << lane;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const u32 lbs = GEN ? (u32)(w * (64 * SS_ITEMS) + lane * SS_ITEMS) : (u32)(w * (64 * SS_ITEMS) + lane);
    u64 gkey = 0;
    u32 pk[SS_ITEMS / 2];                                   // GEN: the lane's 16 digits as written by the count pass
    if (GEN) {
        gkey = ss_gen_key(g, sy, (int)lbs);
        const uint4* dp = (const uint4*)(P.digits + base + lbs);
        const uint4 q0 = dp[0], q1 = dp[1];
        pk[0] = q0.x; pk[1] = q0.y; pk[2] = q0.z; pk[3] = q0.w; pk[4] = q1.x; pk[5] = q1.y; pk[6] = q1.z; pk[7] = q1.w;
    }
    const u64* kp = GEN ? nullptr : P.keys_in + base + lbs;
    const u32* vp = GEN ? nullptr : P.vals_in + base + lbs;
    const u16* dgp = P.digits + base + lbs;
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {                    // all global loads first: they overlap, the ranking below is a chain of LDS operations
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        const bool valid = e < cnt;
        if (GEN) {
            k[j] = gkey; v[j] = (u32)(base + e);
            gkey = (gkey - (u64)sy[lbs + j] * g.top) * g.sigma + sy[lbs + j + g.k];
            ld[j] = (pk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        } else {
            k[j] = valid ? kp[j * 64] : 0ull;                   // one address per array, the row as an immediate offset
            v[j] = valid ? vp[j * 64] : 0u;
            ld[j] = valid ? (u32)dgp[j * 64] : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        const bool valid = e < cnt;
        const u32 d = valid ? ld[j] : 0u;
        const u64 peers = wave_match_lds(M, d, valid, lanebit);
        const u32 prefix = lds_load(&mycnt[d]);
        const u32 rank = (u32)__popcll(peers & lt_mask);
        ld[j] = (d << 16) | (prefix + rank);
        if (valid && rank == 0) lds_store(&mycnt[d], (u16)(prefix + (u32)__popcll(peers)));
    }
    __syncthreads();
    {   // digit runs inside the sorted tile (exclusive scan over the digits), per-wave starts, global base
        const u32 t = threadIdx.x;
        u32 tot[DMAX / 256], sum = 0;
#pragma unroll
        for (int q = 0; q < DMAX / 256; ++q) {               // thread t owns the digits t * (DMAX / 256) + q
            const u32 d = t * (DMAX / 256) + q;
            tot[q] = (u32)wcnt[0][d] + wcnt[1][d] + wcnt[2][d] + wcnt[3][d];
            sum += tot[q];
        }
        u32 total;
        u32 start = block_exclusive_sum<u32, 4>(sum, scan_sm, total);
#pragma unroll
        for (int q = 0; q < DMAX / 256; ++q) {
            const u32 d = t * (DMAX / 256) + q;
            u32 run = start;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const u32 c = wcnt[i][d]; wcnt[i][d] = (u16)run; run += c; }
            gbase[d] = (d < P.D ? P.counts[(size_t)row * P.D + d] : 0u) - start;
            start += tot[q];
        }
    }
    __syncthreads();                                       // splitters / text bytes are dead from here on
    u32 dst[SS_ITEMS];
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        const u32 d = ld[j] >> 16;
        const u32 pos = (ld[j] & 0xFFFFu) + wcnt[w][d];        // position inside the sorted tile
        ld[j] = pos;
        if (e < cnt) { stage32[pos] = v[j]; stage_d[pos] = (u16)d; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; ++r) {
        const u32 sp = (u32)r * 256 + threadIdx.x;
        dst[r] = 0xFFFFFFFFu;
        if (sp < cnt) {
            dst[r] = gbase[stage_d[sp]] + sp;
            P.vals_out[dst[r]] = stage32[sp];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 e = GEN ? lbs + (u32)j : lbs + (u32)j * 64;
        if (e < cnt) stage[ld[j]] = k[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; ++r) {
        const u32 sp = (u32)r * 256 + threadIdx.x;
        if (dst[r] != 0xFFFFFFFFu) P.keys_out[dst[r]] = stage[sp];
    }
}

// ---- two-level MSD partition of (u32 index, u32 value) pairs by the top 16 bits of the index ------------------------------------
// The first two passes of the bucketed scatter (prim.hip): dst[idx[j]] = val[j] wants its pairs grouped by destination window, in
// ANY order inside a window.  Two stable LSD passes (low digit, then high digit) do that with a wave-level match per key; an
// MSD split needs no stability at all -- level 1 on the top 8 bits, level 2 on the next 8 bits inside every level-1 bucket (the
// segment machinery of the splitter sort) -- so a key's slot is whatever one LDS atomic hands out.
struct PSLevel {
    const u32* idx_in; const u32* val_in; u32* idx_out; u32* val_out;
    u32* counts; const u32* blk_seg; const u32* blk_start; const u32* seg_start;
    u32 nseg, R, per_xcd; int shift;
};
__device__ __forceinline__ bool ps_row(const PSLevel& P, u32 row, size_t& base, u32& cnt) {
    const u32 blk = row / P.R;
    if (blk >= P.blk_start[P.nseg]) return false;
    const u32 s = P.blk_seg[blk];
    const u64 t = (u64)(blk - P.blk_start[s]) * P.R + row % P.R;
    const u32 s0 = P.seg_start[s], s1 = P.seg_start[s + 1];
    const u64 off = t * SS_TILE;
    base = s0; cnt = 0;
    if (off >= (u64)(s1 - s0)) return true;
    base = (size_t)s0 + off;
    const u64 left = (u64)(s1 - s0) - off;
    cnt = left < SS_TILE ? (u32)left : (u32)SS_TILE;
    return true;
}
template <int DB>       // digit bits: 8, or 9 when the destination windows would otherwise exceed the LDS image of the final pass
__global__ __launch_bounds__(256) void ps_count_kernel(PSLevel P, u32 rows) {
    constexpr u32 D = 1u << DB;
    __shared__ u32 hist[D];
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    size_t base; u32 cnt;
    if (!ps_row(P, row, base, cnt)) return;
    for (u32 i = threadIdx.x; i < D; i += 256) hist[i] = 0;
    if (cnt == 0) { for (u32 i = threadIdx.x; i < D; i += 256) P.counts[(size_t)row * D + i] = 0; return; }
    __syncthreads();
    const u32 lb = wave_id() * (64 * SS_ITEMS) + lane_id();
    const u32* ip = P.idx_in + base + lb;
    u32 kk[SS_ITEMS];
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) kk[j] = (lb + (u32)j * 64 < cnt) ? ip[j * 64] : 0u;
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const bool valid = lb + (u32)j * 64 < cnt;
        const u32 d = (kk[j] >> P.shift) & (D - 1);
        const u32 d0 = __builtin_amdgcn_readfirstlane(d);
        if (__all(valid && d == d0)) { if (lane_id() == 0) atomicAdd(&hist[d0], 64u); }
        else if (valid) atomicAdd(&hist[d], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < D; i += 256) P.counts[(size_t)row * D + i] = hist[i];
}
template <int DB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ps_scatter_kernel(PSLevel P, u32 rows) {
    constexpr u32 D = 1u << DB;
    __shared__ u32 tcnt[D];
    __shared__ u32 gbase[D];
    __shared__ u32 stage_i[SS_TILE];
    __shared__ u32 stage_v[SS_TILE];
    __shared__ u32 scan_sm[5];
    const int lane = lane_id();
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    size_t base; u32 cnt;
    if (!ps_row(P, row, base, cnt) || cnt == 0) return;
    for (u32 i = threadIdx.x; i < D; i += 256) tcnt[i] = 0;
    __syncthreads();
    const u32 lb = wave_id() * (64 * SS_ITEMS) + lane;
    const u32* ip = P.idx_in + base + lb;
    const u32* vp = P.val_in + base + lb;
    u32 k[SS_ITEMS], v[SS_ITEMS], ld[SS_ITEMS];
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const bool valid = lb + (u32)j * 64 < cnt;
        k[j] = valid ? ip[j * 64] : 0u;
        v[j] = valid ? vp[j * 64] : 0u;
    }
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const bool valid = lb + (u32)j * 64 < cnt;
        const u32 d = (k[j] >> P.shift) & (D - 1);
        const u32 d0 = __builtin_amdgcn_readfirstlane(d);
        u32 rank = 0;
        if (__all(valid && d == d0)) {
            if (lane == 0) rank = atomicAdd(&tcnt[d0], 64u);
            rank = __builtin_amdgcn_readfirstlane(rank) + (u32)lane;
        } else if (valid) rank = atomicAdd(&tcnt[d], 1u);
        ld[j] = rank;
    }
    __syncthreads();
    {
        const u32 t = threadIdx.x;
        u32 tot[D / 256], sum = 0;
#pragma unroll
        for (u32 q = 0; q < D / 256; ++q) { tot[q] = tcnt[t * (D / 256) + q]; sum += tot[q]; }
        u32 total;
        u32 start = block_exclusive_sum<u32, 4>(sum, scan_sm, total);
#pragma unroll
        for (u32 q = 0; q < D / 256; ++q) {
            const u32 d = t * (D / 256) + q;
            tcnt[d] = start;
            gbase[d] = P.counts[(size_t)row * D + d] - start;
            start += tot[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        if (lb + (u32)j * 64 < cnt) {
            const u32 pos = ld[j] + tcnt[(k[j] >> P.shift) & (D - 1)];
            stage_i[pos] = k[j]; stage_v[pos] = v[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; ++r) {
        const u32 sp = (u32)r * 256 + threadIdx.x;
        if (sp < cnt) {
            const u32 key = stage_i[sp];
            const u32 dst = gbase[(key >> P.shift) & (D - 1)] + sp;
            P.idx_out[dst] = key;
            P.val_out[dst] = stage_v[sp];
        }
    }
}

// idx / val (m pairs, only read) -> out_idx / out_val grouped by idx >> (bits - 2 * db), db = digit bits (8 or 9); tmp_*: m entries
// of scratch each
void msd_partition_pairs_u32(Ctx& c, const u32* idx, const u32* val, size_t m, int bits, int db, u32* out_idx, u32* out_val, u32* tmp_idx, u32* tmp_val) {
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    const u32 D = 1u << db;
    u32* seg_start = c.arena.get<u32>(2);
    ss_set_word_kernel<<<1, 1, 0, s>>>(seg_start, 0u);
    LAUNCH_CHECK();
    ss_set_word_kernel<<<1, 1, 0, s>>>(seg_start + 1, (u32)m);
    LAUNCH_CHECK();
    u32 nseg = 1;
    const u64 tiles = (m + SS_TILE - 1) / SS_TILE;
    for (int l = 0; l < 2; ++l) {
        u32 R = 128;
        while (R > 1 && (u64)nseg * R > tiles / 8 + 64) R >>= 1;
        const u32 blocks_ub = (u32)((tiles + R - 1) / R) + nseg;
        const u32 rows = blocks_ub * R;
        u32* nstart = c.arena.get<u32>((size_t)nseg * D + 1);
        const size_t lm = c.arena.mark();
        u32* blk_start = c.arena.get<u32>((size_t)nseg + 1);
        u32* blk_seg = c.arena.get<u32>(blocks_ub);
        u32* counts = c.arena.get<u32>((size_t)rows * D);
        u32* bs = c.arena.get<u32>((size_t)blocks_ub * D);
        ss_nblk_kernel<<<cdiv((size_t)nseg + 1, 256), 256, 0, s>>>(seg_start, nseg, R, blk_start);
        LAUNCH_CHECK();
        exclusive_sum_u32(c, blk_start, blk_start, (size_t)nseg + 1, nullptr);
        ss_blkseg_kernel<<<cdiv(nseg, 256), 256, 0, s>>>(blk_start, nseg, blk_seg);
        LAUNCH_CHECK();
        PSLevel P;
        P.idx_in = l == 0 ? idx : tmp_idx; P.val_in = l == 0 ? val : tmp_val;
        P.idx_out = l == 0 ? tmp_idx : out_idx; P.val_out = l == 0 ? tmp_val : out_val;
        P.counts = counts; P.blk_seg = blk_seg; P.blk_start = blk_start; P.seg_start = seg_start;
        P.nseg = nseg; P.R = R; P.shift = bits - db * (l + 1);
        P.per_xcd = (c.xcd_remap == 1 && rows >= 64) ? cdiv(rows, 8) : 0u;
        const u32 grid = P.per_xcd ? 8 * P.per_xcd : rows;
        {
            const int pc = c.prof_begin(K_RS_COUNT, (u64)m * 4);
            if (db == 9) ps_count_kernel<9><<<grid, 256, 0, s>>>(P, rows); else ps_count_kernel<8><<<grid, 256, 0, s>>>(P, rows);
            LAUNCH_CHECK();
            c.prof_end(pc);
        }
        {
            Ctx::ProfScope prof(c, K_SCAN, (u64)rows * D * 12);
            ss_blocksum_kernel<<<blocks_ub, 256, 0, s>>>(counts, blk_start, nseg, R, D, bs);
            LAUNCH_CHECK();
            ss_segbase_kernel<<<nseg, 1024, 0, s>>>(bs, blk_start, seg_start, D, nstart);
            LAUNCH_CHECK();
            ss_set_word_kernel<<<1, 1, 0, s>>>(nstart + (size_t)nseg * D, (u32)m);
            LAUNCH_CHECK();
            ss_apply_kernel<<<blocks_ub, 256, 0, s>>>(counts, blk_start, nseg, R, D, bs);
            LAUNCH_CHECK();
        }
        {
            const int ps = c.prof_begin(K_RS_SCATTER_U32, (u64)m * 16);
            if (db == 9) ps_scatter_kernel<9><<<grid, 256, 0, s>>>(P, rows); else ps_scatter_kernel<8><<<grid, 256, 0, s>>>(P, rows);
            LAUNCH_CHECK();
            c.prof_end(ps);
        }
        c.arena.release(lm);
        seg_start = nstart;
        nseg *= D;
    }
    c.arena.release(mark);
}

// ---- leaves -> units ----------------------------------------------------------------------------------------------------------
// class of leaf l: 0 small (<= SS_SMALL pairs, packed with its neighbours), 1 medium (a unit of its own), 2 skip (a large
// equality leaf: all keys equal), 3 large (LSD fall-back)
__device__ __forceinline__ u32 ss_leaf_class(const u32* __restrict__ leaf_start, u32 l, u32 small = SS_SMALL) {
    const u32 size = leaf_start[l + 1] - leaf_start[l];
    if (size <= small) return 0;
    if (l & 1u) return 2;
    return size <= SS_UNIT_MAX ? 1u : 3u;
}
__device__ __forceinline__ bool ss_unit_first(const u32* __restrict__ leaf_start, u32 l, u32 cls, u32 small = SS_SMALL) {
    if (cls == 1) return true;
    if (cls != 0) return false;
    if (l == 0) return true;
    return ss_leaf_class(leaf_start, l - 1, small) != 0 || (leaf_start[l] / small) != (leaf_start[l - 1] / small);
}
__global__ void ss_unit_flag_kernel(const u32* __restrict__ leaf_start, u32 nleaf, u32* __restrict__ flag, u32* __restrict__ large_list,
                                    u32* __restrict__ large_count, u32 large_cap, u32 small = SS_SMALL) {
    const u32 l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaf) return;
    const u32 cls = ss_leaf_class(leaf_start, l, small);
    flag[l] = ss_unit_first(leaf_start, l, cls, small) ? 1u : 0u;
    if (cls == 3) { const u32 i = atomicAdd(large_count, 1u); if (i < large_cap) large_list[i] = l; }
}
// uidx = exclusive scan of flag; unit_rng[2u] / [2u + 1] = first pair / end of unit u
__global__ void ss_unit_fill_kernel(const u32* __restrict__ leaf_start, u32 nleaf, const u32* __restrict__ uidx, u32* __restrict__ unit_rng, u32 small = SS_SMALL) {
    const u32 l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaf) return;
    const u32 cls = ss_leaf_class(leaf_start, l, small);
    if (cls > 1) return;
    const bool first = ss_unit_first(leaf_start, l, cls, small);
    const u32 u = uidx[l] - (first ? 0u : 1u);               // exclusive scan: a first leaf sees its own index, the others the next one
    if (first) unit_rng[2 * u] = leaf_start[l];
    bool last = (cls == 1) || (l + 1 == nleaf);
    if (!last) {
        const u32 c2 = ss_leaf_class(leaf_start, l + 1, small);
        last = c2 != 0 || (leaf_start[l + 1] / small) != (leaf_start[l] / small);
    }
    if (last) unit_rng[2 * u + 1] = leaf_start[l + 1];
}

// ---- leaf sort: one workgroup sorts one unit of <= 8192 pairs in LDS (stable LSD passes on the bits in which its keys differ) --
// Ranks inside a wave come from a wave-level match (8 ballots per key); the running per-wave digit counters are bumped with
// LDS atomics that RETURN the old value (ds_add_rtn): the 16 rows of a pass are then independent instructions in the wave's
// in-order LDS queue, instead of a read-modify-write chain of 16 round trips.  When the differing key bits and a 13-bit slot
// number fit one 64-bit word (almost always: the keys of a unit share their leading bits), only that word travels through
// LDS in every pass and the values are fetched once at the end.
constexpr int LS_NW = 8;
__device__ __forceinline__ u32 ls_rank(u32 d, bool valid, u32* mycnt, unsigned long long* M, u64 lanebit, u64 lt_mask) {
    const u64 peers = wave_match_lds(M, d, valid, lanebit);
    const u32 prefix = lds_load(&mycnt[d]);
    const u32 rank = (u32)__popcll(peers & lt_mask);
    if (valid && rank == 0) lds_store(&mycnt[d], prefix + (u32)__popcll(peers));
    return prefix + rank;
}
// per-wave digit counts -> start of every (wave, digit) run inside the sorted unit; executed by ONE wave (lane l owns the
// digits 4l .. 4l+3), the others wait at the barrier that follows
__device__ __forceinline__ void ls_digit_starts(u32 (*wcnt)[256], int lane) {
    uint4 c[LS_NW];
    u32 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
    for (int i = 0; i < LS_NW; ++i) {
        c[i] = *(const uint4*)&wcnt[i][4 * lane];
        t0 += c[i].x; t1 += c[i].y; t2 += c[i].z; t3 += c[i].w;
    }
    const u32 sum = t0 + t1 + t2 + t3;
    u32 r0 = wave_inclusive_sum(sum) - sum;
    u32 r1 = r0 + t0, r2 = r1 + t1, r3 = r2 + t2;
#pragma unroll
    for (int i = 0; i < LS_NW; ++i) {
        *(uint4*)&wcnt[i][4 * lane] = make_uint4(r0, r1, r2, r3);
        r0 += c[i].x; r1 += c[i].y; r2 += c[i].z; r3 += c[i].w;
    }
}
// Composite units: ROWS 64-pair rows per wave (the unit holds at most ROWS * 512 pairs; the host launches one variant per
// size class, so the row loops carry no run-time bounds and the LDS operations of neighbouring rows overlap).  Units whose
// differing bits do not fit the composite word are appended to wide_list (wide_list[0] = count) for ss_leaf_wide_kernel.
template <int ROWS>
__global__ __launch_bounds__(LS_NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void ss_leaf_comp_kernel(u64* __restrict__ keys, u32* __restrict__ vals, const u32* __restrict__ unit_rng,
                                                                   const u32* __restrict__ list, u32 count, u32* __restrict__ d_err, u32* __restrict__ wide_list) {
    __shared__ __align__(16) u32 wcnt[LS_NW][256];
    __shared__ u64 stage[ROWS * LS_NW * 64];
    __shared__ u64 red[2][LS_NW];
    if (blockIdx.x >= count) return;
    const u32 u = list[blockIdx.x];
    const u32 a = unit_rng[2 * u], b = unit_rng[2 * u + 1];
    const u32 m = b - a;
    if (m <= 1) return;
    if (m > (u32)ROWS * LS_NW * 64) { if (threadIdx.x == 0) atomicOr(d_err, 2u); return; }
    const int lane = lane_id(), w = wave_id();
    const u32 wbase = (u32)w * ROWS * 64 + (u32)lane;           // logical slot of (w, j, lane) = wbase + 64 j
    u64 k[ROWS];
    u64 kmin = ~0ull, kmax = 0;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        const bool valid = L < m;
        k[j] = valid ? keys[(size_t)a + L] : 0ull;
        if (valid) { kmin = k[j] < kmin ? k[j] : kmin; kmax = k[j] > kmax ? k[j] : kmax; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 o1 = __shfl_xor(kmin, d, 64), o2 = __shfl_xor(kmax, d, 64);
        kmin = o1 < kmin ? o1 : kmin; kmax = o2 > kmax ? o2 : kmax;
    }
    if (lane == 0) { red[0][w] = kmin; red[1][w] = kmax; }
    __syncthreads();
    kmin = red[0][0]; kmax = red[1][0];
#pragma unroll
    for (int i = 1; i < LS_NW; ++i) { kmin = red[0][i] < kmin ? red[0][i] : kmin; kmax = red[1][i] > kmax ? red[1][i] : kmax; }
    const u64 diff = kmin ^ kmax;
    if (diff == 0) return;                                      // all keys equal
    const int nbits = 64 - __clzll((long long)diff);
    if (nbits + 13 > 64) { if (threadIdx.x == 0) wide_list[1 + atomicAdd(wide_list, 1u)] = u; return; }
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u32* mycnt = wcnt[w];
    u32* stage32 = (u32*)stage;
    // lane-mask table of the LDS match: in the staging buffer, idle while ranking (for ROWS = 4 the buffer is exactly the 8 tables)
    unsigned long long* M = (unsigned long long*)stage + w * 256;
    const u64 lanebit = 1ull << lane;
    // composite word: (differing key bits) << 13 | slot
    const u64 lowmask = (nbits == 64) ? ~0ull : ((1ull << nbits) - 1);
    const u64 high = kmin & ~lowmask;                           // the bits all keys share
#pragma unroll
    for (int j = 0; j < ROWS; ++j) k[j] = ((k[j] & lowmask) << 13) | (u64)(wbase + (u32)j * 64);
    for (int i = lane; i < 256; i += 64) mycnt[i] = 0;          // every wave owns (and clears) its own counters
    u32 loc[ROWS];
    for (int shift = 13; shift < 13 + nbits; shift += 8) {
        for (int i = lane; i < 256; i += 64) M[i] = 0;          // the wave's own table; the staging buffer is idle here
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            loc[j] = ls_rank((u32)(k[j] >> shift) & 255u, wbase + (u32)j * 64 < m, mycnt, M, lanebit, lt_mask);
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // overlap the LDS operations of four rows, not of all (registers)
        }
        __syncthreads();
        if (w == 0) ls_digit_starts(wcnt, lane);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
            if (wbase + (u32)j * 64 < m) stage[loc[j] + mycnt[(u32)(k[j] >> shift) & 255u]] = k[j];
        for (int i = lane; i < 256; i += 64) mycnt[i] = 0;      // after this wave's last use; only this wave touches them before the next barrier
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) if (wbase + (u32)j * 64 < m) k[j] = stage[wbase + (u32)j * 64];
        __syncthreads();                                        // the match tables of the next pass overwrite the buffer
    }
    // values: they were not carried through the passes (registers); every thread parks the values of its original slots
    // (still untouched in memory) in LDS, then picks up those of its sorted words
#pragma unroll
    for (int j = 0; j < ROWS; ++j) if (wbase + (u32)j * 64 < m) stage32[wbase + (u32)j * 64] = vals[(size_t)a + wbase + (u32)j * 64];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        if (L < m) {
            keys[(size_t)a + L] = (k[j] >> 13) | high;
            vals[(size_t)a + L] = stage32[(u32)k[j] & 8191u];
        }
    }
}

// The units whose keys differ in more than 51 bits: (key, value) pairs through LDS, run-time row count.
__global__ __launch_bounds__(LS_NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void ss_leaf_wide_kernel(u64* __restrict__ keys, u32* __restrict__ vals, const u32* __restrict__ unit_rng,
                                                                   const u32* __restrict__ wide_list, u32 count, u32* __restrict__ d_err) {
    __shared__ __align__(16) u32 wcnt[LS_NW][256];
    __shared__ u64 stage[SS_UNIT_MAX];
    if (blockIdx.x >= count) return;
    const u32 u = wide_list[1 + blockIdx.x];
    const u32 a = unit_rng[2 * u], b = unit_rng[2 * u + 1];
    const u32 m = b - a;
    if (m <= 1) return;
    if (m > SS_UNIT_MAX) { if (threadIdx.x == 0) atomicOr(d_err, 2u); return; }
    const int lane = lane_id(), w = wave_id();
    const u32 rows = (m + LS_NW * 64 - 1) / (LS_NW * 64);      // 64-pair rows per wave
    const u32 wbase = (u32)w * rows * 64 + (u32)lane;
    u64 k[SS_ITEMS];
    u32 v[SS_ITEMS];
    u32 loc[SS_ITEMS];
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        const bool valid = (u32)j < rows && L < m;
        k[j] = valid ? keys[(size_t)a + L] : 0ull;
        v[j] = valid ? vals[(size_t)a + L] : 0u;
    }
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u32* mycnt = wcnt[w];
    u32* stage32 = (u32*)stage;
    unsigned long long* M = (unsigned long long*)stage + w * 256;
    const u64 lanebit = 1ull << lane;
    for (int shift = 0; shift < 64; shift += 8) {
        for (int i = threadIdx.x; i < LS_NW * 256; i += LS_NW * 64) { (&wcnt[0][0])[i] = 0; stage[i] = 0; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j)
            if ((u32)j < rows) loc[j] = ls_rank((u32)(k[j] >> shift) & 255u, wbase + (u32)j * 64 < m, mycnt, M, lanebit, lt_mask);
        __syncthreads();
        if (w == 0) ls_digit_starts(wcnt, lane);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) {
            if ((u32)j < rows && wbase + (u32)j * 64 < m) {
                loc[j] += wcnt[w][(u32)(k[j] >> shift) & 255u];
                stage[loc[j]] = k[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) if ((u32)j < rows && wbase + (u32)j * 64 < m) k[j] = stage[wbase + (u32)j * 64];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) if ((u32)j < rows && wbase + (u32)j * 64 < m) stage32[loc[j]] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) if ((u32)j < rows && wbase + (u32)j * 64 < m) v[j] = stage32[wbase + (u32)j * 64];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < SS_ITEMS; ++j) {
        const u32 L = wbase + (u32)j * 64;
        if ((u32)j < rows && L < m) { keys[(size_t)a + L] = k[j]; vals[(size_t)a + L] = v[j]; }
    }
}

// units by size class (rows per wave of the composite kernel): class c = units of (2048 c, 2048 (c + 1)] pairs
// wide = 1 (wsort.hip): the six size classes of its leaf kernels (prim.hpp wide_class: a unit of 2 600 records in a 4 096-slot
// workgroup runs a quarter of its rows empty)
constexpr int UC_T = 1024;                                       // threads per workgroup of ss_unit_class_kernel
__global__ __launch_bounds__(UC_T) void ss_unit_class_kernel(const u32* __restrict__ unit_rng, const u32* __restrict__ d_nunits, u32* __restrict__ cls_count,
                                                            u32* __restrict__ cls_list, u32 cap, int wide = 0) {
    // one atomic per WORKGROUP and class (round 6: per wave it was 500 000 atomics on ten addresses at 2e9 B -- 1.8 ms of a kernel that
    // moves 16 MB; the lists stay in nearly ascending unit order)
    __shared__ u32 wcnt[UC_T / 64][16];
    __shared__ u32 wbase[16];
    const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 ncls = wide ? (u32)WIDE_NCLS : 4u;
    u32 c = 15u;                                                 // (no unit)
    if (u < *d_nunits) {
        const u32 m = unit_rng[2 * u + 1] - unit_rng[2 * u];
        if (m > 1 && m <= SS_UNIT_MAX) c = wide ? wide_class(m) : (m - 1) / 2048;
    }
    const int lane = lane_id(), w = wave_id();
    u32 below = 0;                                               // lanes of this wave in front of me with my class
    for (u32 q = 0; q < ncls; ++q) {
        const u64 mask = __ballot(c == q);
        if (lane == 0) wcnt[w][q] = (u32)__popcll(mask);
        if (c == q) below = (u32)__popcll(mask & ((1ull << lane) - 1));
    }
    __syncthreads();
    if (threadIdx.x < ncls) {
        const u32 q = threadIdx.x;
        u32 tot = 0;
        for (int i = 0; i < UC_T / 64; ++i) { const u32 t = wcnt[i][q]; wcnt[i][q] = tot; tot += t; }
        wbase[q] = tot ? atomicAdd(&cls_count[q], tot) : 0u;
    }
    __syncthreads();
    if (c < ncls) cls_list[(size_t)c * cap + wbase[c] + wcnt[w][c] + below] = u;
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
static u32 pow2_ceil(u64 x) { u32 p = 1; while ((u64)p < x) p <<= 1; return p; }

bool splitter_sort_applicable(size_t n) { return n >= ((size_t)1 << 20) && n < ((size_t)1 << 32); }

int splitter_sort_pairs_u64(Ctx& c, u64* keys[2], u32* vals[2], size_t n, const TextKeyGen* gen, SplitSortStats* st) {
    SplitSortStats local;
    if (!st) st = &local;
    *st = SplitSortStats();
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    TextKeyGen g;
    if (gen) g = *gen; else memset(&g, 0, sizeof(g));

    // ---- fan-outs: two levels while 65536 range leaves keep the average leaf <= 4352 pairs (oversampling 64 then keeps a
    // leaf below the 8192 of one workgroup: 5 sigma), three levels with leaves of ~2048 above; c.ssort_levels forces L (tests)
    int L = n <= (size_t)256 * 3072 ? 1 : ((n + 65535) / 65536 <= 4352 ? 2 : 3);
    if (c.ssort_levels >= 1 && c.ssort_levels <= 3) L = c.ssort_levels;
    u32 F[3] = { 1, 1, 1 };
    u32 os;
    {
        const u32 cap = L == 1 ? 256u : (L == 2 ? 65536u : (1u << 24));
        u32 nl = pow2_ceil((n + (L == 3 ? 2047 : 3071)) / (L == 3 ? 2048 : 3072));
        if (nl > cap) nl = cap;
        if (nl < (2u << (L - 1))) nl = 2u << (L - 1);           // every level splits at least in two
        const u32 lastF = nl > 256 ? 256u : (L == 1 ? nl : nl >> (L - 1));
        F[L - 1] = lastF;
        u32 rest = nl / lastF;
        if (L == 2) F[0] = rest;
        if (L == 3) { F[0] = 1; while ((u64)F[0] * F[0] < rest) F[0] <<= 1; F[1] = rest / F[0]; if (F[1] < 2) { F[1] = 2; } }
        const u64 avg = (n + nl - 1) / nl;
        os = avg > 3072 ? 64 : (avg > 2304 ? 32 : 16);
    }
    const u32 NLr = F[0] * F[1] * F[2];          // range leaves
    const u32 NS = NLr - 1;
    const u32 S = os * NLr;
    st->levels = (u32)L; st->range_leaves = NLr; st->samples = S;

    // ---- splitters -----------------------------------------------------------------------------------------------------------
    u64* sp = c.arena.get<u64>((size_t)NS + 1);
    {
        const size_t m2 = c.arena.mark();
        u64* sk[2] = { c.arena.get<u64>(S), c.arena.get<u64>(S) };
        u32* sv[2] = { c.arena.get<u32>(S), c.arena.get<u32>(S) };
        if (gen) ss_sample_kernel<true><<<cdiv(S, 256), 256, 0, s>>>(nullptr, g, n, S, sk[0], sv[0]);
        else ss_sample_kernel<false><<<cdiv(S, 256), 256, 0, s>>>(keys[0], g, n, S, sk[0], sv[0]);
        LAUNCH_CHECK();
        const int x = radix_sort_pairs_u64(c, sk, sv, S, 0, 64);
        ss_pick_kernel<<<cdiv((size_t)NS + 1, 256), 256, 0, s>>>(sk[x], NS, os, sp);
        LAUNCH_CHECK();
        c.arena.release(m2);
    }

    // ---- partition levels ----------------------------------------------------------------------------------------------------
    u16* digits = c.arena.get<u16>(align_up(n, SS_TILE) + SS_TILE);
    u32* seg_start = c.arena.get<u32>(2);
    ss_set_word_kernel<<<1, 1, 0, s>>>(seg_start, 0u);
    LAUNCH_CHECK();
    ss_set_word_kernel<<<1, 1, 0, s>>>(seg_start + 1, (u32)n);
    LAUNCH_CHECK();
    u32 nseg = 1;
    int cur = gen ? -1 : 0;                                   // buffer pair that holds the level's input (-1: the text)
    for (int l = 0; l < L; ++l) {
        const bool last = (l == L - 1);
        const bool gl = gen && l == 0;
        const u32 D = last ? 2 * F[l] : F[l];
        u32 stride = 1;
        for (int q = l + 1; q < L; ++q) stride *= F[q];
        // rows per row block: long blocks while there are few segments, short ones when segments are small
        const u64 tiles = (n + SS_TILE - 1) / SS_TILE;
        u32 R = 128;
        while (R > 1 && (u64)nseg * R > tiles / 8 + 64) R >>= 1;
        const u32 blocks_ub = (u32)((tiles + R - 1) / R) + nseg;
        const u32 rows = blocks_ub * R;
        // the next level's segment starts are allocated below the level's scratch, so they survive its release
        u32* nstart = c.arena.get<u32>((size_t)nseg * D + 1);
        const size_t lm2 = c.arena.mark();
        u32* blk_start = c.arena.get<u32>((size_t)nseg + 1);
        u32* blk_seg = c.arena.get<u32>(blocks_ub);
        u32* counts = c.arena.get<u32>((size_t)rows * D);
        u32* bs = c.arena.get<u32>((size_t)blocks_ub * D);
        ss_nblk_kernel<<<cdiv((size_t)nseg + 1, 256), 256, 0, s>>>(seg_start, nseg, R, blk_start);
        LAUNCH_CHECK();
        exclusive_sum_u32(c, blk_start, blk_start, (size_t)nseg + 1, nullptr);
        ss_blkseg_kernel<<<cdiv(nseg, 256), 256, 0, s>>>(blk_start, nseg, blk_seg);
        LAUNCH_CHECK();
        SSLevel P;
        P.keys_in = cur >= 0 ? keys[cur] : nullptr; P.vals_in = cur >= 0 ? vals[cur] : nullptr;
        const int nxt = cur < 0 ? 0 : (cur ^ 1);
        P.keys_out = keys[nxt]; P.vals_out = vals[nxt];
        P.digits = digits;
        P.counts = counts; P.blk_seg = blk_seg; P.blk_start = blk_start; P.seg_start = seg_start; P.sp = sp;
        P.nseg = nseg; P.F = F[l]; P.stride = stride; P.R = R; P.D = D;
        P.per_xcd = (c.xcd_remap == 1 && rows >= 64) ? cdiv(rows, 8) : 0u;
        const u32 grid = P.per_xcd ? 8 * P.per_xcd : rows;
        {
            const int pc = c.prof_begin(K_RS_COUNT, (u64)n * (gl ? 1 : 8));
            if (gl && last) ss_count_kernel<true, true><<<grid, 256, 0, s>>>(P, g, rows);
            else if (gl) ss_count_kernel<true, false><<<grid, 256, 0, s>>>(P, g, rows);
            else if (last) ss_count_kernel<false, true><<<grid, 256, 0, s>>>(P, g, rows);
            else ss_count_kernel<false, false><<<grid, 256, 0, s>>>(P, g, rows);
            LAUNCH_CHECK();
            c.prof_end(pc);
        }
        {
            Ctx::ProfScope prof(c, K_SCAN, (u64)rows * D * 12);
            ss_blocksum_kernel<<<blocks_ub, 256, 0, s>>>(counts, blk_start, nseg, R, D, bs);
            LAUNCH_CHECK();
            ss_segbase_kernel<<<nseg, 1024, 0, s>>>(bs, blk_start, seg_start, D, nstart);
            LAUNCH_CHECK();
            ss_set_word_kernel<<<1, 1, 0, s>>>(nstart + (size_t)nseg * D, (u32)n);
            LAUNCH_CHECK();
            ss_apply_kernel<<<blocks_ub, 256, 0, s>>>(counts, blk_start, nseg, R, D, bs);
            LAUNCH_CHECK();
        }
        {
            const int ps = c.prof_begin(K_RS_SCATTER_U64, (u64)n * (gl ? 13 : 24));
            if (gl && last) ss_scatter_stable_kernel<true, true><<<grid, 256, 0, s>>>(P, g, rows);
            else if (gl) ss_scatter_kernel<true, false><<<grid, 256, 0, s>>>(P, g, rows);
            else if (last) ss_scatter_stable_kernel<false, true><<<grid, 256, 0, s>>>(P, g, rows);
            else ss_scatter_kernel<false, false><<<grid, 256, 0, s>>>(P, g, rows);
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

    // ---- units + leaf sort ---------------------------------------------------------------------------------------------------
    constexpr u32 LARGE_CAP = 1024;
    u32* flag = c.arena.get<u32>((size_t)nleaf + 1);
    u32* unit_rng = c.arena.get<u32>(2 * ((size_t)nleaf + 1));
    u32* large = c.arena.get<u32>(LARGE_CAP + 6);              // [0] = count of large leaves, [1] = number of units, [2..5] units per size class, then the list
    u32* wide_list = c.arena.get<u32>((size_t)nleaf + 2);       // [0] = number of, [1..] = the units whose differing key bits do not fit the composite word
    u32* cls_list = c.arena.get<u32>(4 * ((size_t)nleaf + 1));  // the units by size class
    HIP_TRY(hipMemsetAsync(large, 0, 6 * sizeof(u32), s));       // [0] large leaves, [1] units, [2..5] units per size class
    HIP_TRY(hipMemsetAsync(wide_list, 0, sizeof(u32), s));
    ss_unit_flag_kernel<<<cdiv(nleaf, 256), 256, 0, s>>>(leaf_start, nleaf, flag, large + 6, large, LARGE_CAP);
    LAUNCH_CHECK();
    exclusive_sum_u32(c, flag, flag, nleaf, large + 1);
    ss_unit_fill_kernel<<<cdiv(nleaf, 256), 256, 0, s>>>(leaf_start, nleaf, flag, unit_rng);
    LAUNCH_CHECK();
    ss_unit_class_kernel<<<cdiv((size_t)nleaf + 1, UC_T), UC_T, 0, s>>>(unit_rng, large + 1, large + 2, cls_list, nleaf + 1);
    LAUNCH_CHECK();
    u32 hc[6];
    c.read_n(large, hc, 6);
    const u32 nlarge = hc[0], nunits = hc[1];
    st->units = nunits; st->large_leaves = nlarge;
    {
        Ctx::ProfScope prof(c, K_SS_LEAF, (u64)n * 24);
        u64* K = keys[cur]; u32* V = vals[cur];
        const u32 cap = nleaf + 1;
        if (hc[2]) { ss_leaf_comp_kernel<4><<<hc[2], LS_NW * 64, 0, s>>>(K, V, unit_rng, cls_list, hc[2], c.d_err, wide_list); LAUNCH_CHECK(); }
        if (hc[3]) { ss_leaf_comp_kernel<8><<<hc[3], LS_NW * 64, 0, s>>>(K, V, unit_rng, cls_list + (size_t)cap, hc[3], c.d_err, wide_list); LAUNCH_CHECK(); }
        if (hc[4]) { ss_leaf_comp_kernel<12><<<hc[4], LS_NW * 64, 0, s>>>(K, V, unit_rng, cls_list + 2 * (size_t)cap, hc[4], c.d_err, wide_list); LAUNCH_CHECK(); }
        if (hc[5]) { ss_leaf_comp_kernel<16><<<hc[5], LS_NW * 64, 0, s>>>(K, V, unit_rng, cls_list + 3 * (size_t)cap, hc[5], c.d_err, wide_list); LAUNCH_CHECK(); }
        const u32 nwide = (hc[2] | hc[3] | hc[4] | hc[5]) ? c.read(wide_list) : 0u;
        if (nwide) {                                            // rare: units whose keys differ in more than 51 bits
            ss_leaf_wide_kernel<<<nwide, LS_NW * 64, 0, s>>>(K, V, unit_rng, wide_list, nwide, c.d_err);
            LAUNCH_CHECK();
            st->wide_units = nwide;
        }
    }
    if (nlarge) {                                              // leaves above the workgroup capacity: LSD sort, one by one
        if (nlarge > LARGE_CAP) throw HipError{hipErrorUnknown, "splitter sort: too many oversized leaves", (int)__LINE__};
        std::vector<u32> ll(nlarge), ls((size_t)nleaf + 1);
        HIP_TRY(hipMemcpyAsync(ll.data(), large + 6, nlarge * sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ls.data(), leaf_start, ((size_t)nleaf + 1) * sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (u32 i = 0; i < nlarge; ++i) {
            const size_t a = ls[ll[i]], b = ls[ll[i] + 1];
            u64* kk[2] = { keys[cur] + a, keys[cur ^ 1] + a };
            u32* vv[2] = { vals[cur] + a, vals[cur ^ 1] + a };
            const int y = radix_sort_pairs_u64(c, kk, vv, b - a, 0, 64);
            if (y == 1) {
                HIP_TRY(hipMemcpyAsync(kk[0], kk[1], (b - a) * sizeof(u64), hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipMemcpyAsync(vv[0], vv[1], (b - a) * sizeof(u32), hipMemcpyDeviceToDevice, s));
            }
            st->large_pairs += b - a;
        }
    }
    c.arena.release(mark);
    return cur;
}

// ---- the level / unit bookkeeping as host functions, shared with wsort.hip (same kernels as above) ------------------------------
void ss_level_tables(Ctx& c, const u32* seg_start, u32 nseg, size_t n, u32 D, SegTables& T) {
    hipStream_t s = c.stream;
    const u64 tiles = (n + SS_TILE - 1) / SS_TILE;
    u32 R = 128;
    while (R > 1 && (u64)nseg * R > tiles / 8 + 64) R >>= 1;
    T.R = R;
    T.blocks_ub = (u32)((tiles + R - 1) / R) + nseg;
    T.rows = T.blocks_ub * R;
    T.blk_start = c.arena.get<u32>((size_t)nseg + 1);
    T.blk_seg = c.arena.get<u32>(T.blocks_ub);
    T.counts = c.arena.get<u32>((size_t)T.rows * D);
    T.bs = c.arena.get<u32>((size_t)T.blocks_ub * D);
    ss_nblk_kernel<<<cdiv((size_t)nseg + 1, 256), 256, 0, s>>>(seg_start, nseg, R, T.blk_start);
    LAUNCH_CHECK();
    exclusive_sum_u32(c, T.blk_start, T.blk_start, (size_t)nseg + 1, nullptr);
    ss_blkseg_kernel<<<cdiv(nseg, 256), 256, 0, s>>>(T.blk_start, nseg, T.blk_seg);
    LAUNCH_CHECK();
}
void ss_level_offsets(Ctx& c, const SegTables& T, const u32* seg_start, u32 nseg, u32 D, u32* nstart, size_t n) {
    hipStream_t s = c.stream;
    Ctx::ProfScope prof(c, K_SCAN, (u64)T.rows * D * 12);
    ss_blocksum_kernel<<<T.blocks_ub, 256, 0, s>>>(T.counts, T.blk_start, nseg, T.R, D, T.bs);
    LAUNCH_CHECK();
    ss_segbase_kernel<<<nseg, 1024, 0, s>>>(T.bs, T.blk_start, seg_start, D, nstart);
    LAUNCH_CHECK();
    ss_set_word_kernel<<<1, 1, 0, s>>>(nstart + (size_t)nseg * D, (u32)n);
    LAUNCH_CHECK();
    ss_apply_kernel<<<T.blocks_ub, 256, 0, s>>>(T.counts, T.blk_start, nseg, T.R, D, T.bs);
    LAUNCH_CHECK();
}
// the same for a level whose `nsub` input segments are pieces of `nsuper` output segments (consecutive groups of nsub / nsuper pieces:
// wsort.hip merges the chunk-wise first level that way): blk_super[S] = first row block of super-segment S, out_start[S] = its first
// output slot; nstart receives the nsuper * D + 1 digit starts
void ss_level_offsets_sub(Ctx& c, const SegTables& T, u32 nsub, const u32* blk_super, const u32* out_start, u32 nsuper, u32 D, u32* nstart, size_t n) {
    hipStream_t s = c.stream;
    Ctx::ProfScope prof(c, K_SCAN, (u64)T.rows * D * 12);
    ss_blocksum_kernel<<<T.blocks_ub, 256, 0, s>>>(T.counts, T.blk_start, nsub, T.R, D, T.bs);
    LAUNCH_CHECK();
    ss_segbase_kernel<<<nsuper, 1024, 0, s>>>(T.bs, blk_super, out_start, D, nstart);
    LAUNCH_CHECK();
    ss_set_word_kernel<<<1, 1, 0, s>>>(nstart + (size_t)nsuper * D, (u32)n);
    LAUNCH_CHECK();
    ss_apply_kernel<<<T.blocks_ub, 256, 0, s>>>(T.counts, T.blk_start, nsub, T.R, D, T.bs);
    LAUNCH_CHECK();
}
u32* ss_first_segment(Ctx& c, size_t n) {
    u32* seg_start = c.arena.get<u32>(2);
    ss_set_word_kernel<<<1, 1, 0, c.stream>>>(seg_start, 0u);
    LAUNCH_CHECK();
    ss_set_word_kernel<<<1, 1, 0, c.stream>>>(seg_start + 1, (u32)n);
    LAUNCH_CHECK();
    return seg_start;
}
void ss_build_units(Ctx& c, const u32* leaf_start, u32 nleaf, UnitTables& U, u32 small) {
    if (small == 0) small = SS_SMALL;
    hipStream_t s = c.stream;
    constexpr u32 LARGE_CAP = 1024;
    U.large_cap = LARGE_CAP;
    u32* flag = c.arena.get<u32>((size_t)nleaf + 1);
    U.unit_rng = c.arena.get<u32>(2 * ((size_t)nleaf + 1));
    U.large = c.arena.get<u32>(LARGE_CAP + 6);
    U.cls_list = c.arena.get<u32>((U.wide_classes ? (size_t)WIDE_NCLS : 4) * ((size_t)nleaf + 1));
    u32* wcnt = U.wide_classes ? c.arena.get<u32>(16) : nullptr;
    if (wcnt) HIP_TRY(hipMemsetAsync(wcnt, 0, 16 * sizeof(u32), s));
    U.cap = nleaf + 1;
    HIP_TRY(hipMemsetAsync(U.large, 0, 6 * sizeof(u32), s));
    ss_unit_flag_kernel<<<cdiv(nleaf, 256), 256, 0, s>>>(leaf_start, nleaf, flag, U.large + 6, U.large, LARGE_CAP, small);
    LAUNCH_CHECK();
    exclusive_sum_u32(c, flag, flag, nleaf, U.large + 1);
    ss_unit_fill_kernel<<<cdiv(nleaf, 256), 256, 0, s>>>(leaf_start, nleaf, flag, U.unit_rng, small);
    LAUNCH_CHECK();
    ss_unit_class_kernel<<<cdiv((size_t)nleaf + 1, UC_T), UC_T, 0, s>>>(U.unit_rng, U.large + 1, wcnt ? wcnt : U.large + 2, U.cls_list, nleaf + 1, U.wide_classes);
    LAUNCH_CHECK();
    c.read_n(U.large, U.hc, 6);
    if (wcnt) c.read_n(wcnt, U.whc, 16);
}
void ss_fanouts(Ctx& c, size_t n, int& L, u32 F[3], u32& os, u32 leaf3, int wide2) {
    // two levels reach 65 536 leaves: beyond 3 584 pairs per leaf the largest (slowest) size class of the leaf kernels takes over and a
    // third level is cheaper (256 MiB of text, 4 096 per leaf: suffix array 25.0 -> 20.8 ms, the call 55.9 -> 51.9 ms; 220 M, 3 357 per
    // leaf: no difference; the limit was 4 352)
    L = n <= (size_t)256 * 3072 ? 1 : ((n + 65535) / 65536 <= 3584 ? 2 : 3);
    if (c.ssort_levels >= 1 && c.ssort_levels <= 3) L = c.ssort_levels;
    F[0] = F[1] = F[2] = 1;
    if (leaf3 == 0) leaf3 = 2048;                              // target leaf size with three levels
    if ((wide2 == 1 && L == 3) || wide2 == 2) {                // two levels of up to 1024 buckets instead of three of up to 256 (wsort.hip;
        u32 nl2 = pow2_ceil((n + leaf3 - 1) / leaf3);          //  measured slower at 2e9: the 1024-way first level no longer hides behind the upload)
        if (nl2 <= (1u << 20) && nl2 >= 2048u) {               // (2: whenever there are enough leaves for it -- tests)
            L = 2;
            F[1] = 1024u;
            F[0] = nl2 / F[1];
            const u64 avg2 = (n + nl2 - 1) / nl2;
            os = avg2 > 3072 ? 64 : (avg2 > 2304 ? 32 : 16);
            return;
        }
    }
    const u32 cap = L == 1 ? 256u : (L == 2 ? 65536u : (1u << 24));
    u32 nl = pow2_ceil((n + (L == 3 ? leaf3 - 1 : 3071)) / (L == 3 ? leaf3 : 3072));
    if (nl > cap) nl = cap;
    if (nl < (2u << (L - 1))) nl = 2u << (L - 1);
    const u32 lastF = nl > 256 ? 256u : (L == 1 ? nl : nl >> (L - 1));
    F[L - 1] = lastF;
    u32 rest = nl / lastF;
    if (L == 2) F[0] = rest;
    if (L == 3) { F[0] = 1; while ((u64)F[0] * F[0] < rest) F[0] <<= 1; F[1] = rest / F[0]; if (F[1] < 2) { F[1] = 2; } }
    const u64 avg = (n + nl - 1) / nl;
    os = avg > 3072 ? 64 : (avg > 2304 ? 32 : 16);
}

}  // namespace tdc
