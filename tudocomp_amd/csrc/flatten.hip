// flatten.hip -- factor list <-> position space, and lzss::FactorBuffer::flatten
// (compressors/lzss/LZSSFactors.hpp:79-132).
//
// flatten redirects the source of factor f through the factor s covering that source while the whole copy fits
// inside s (:106-119).  The reference does it sequentially in position order, so f sees the ALREADY FLATTENED source
// of an earlier factor (s.pos < f.pos) and the ORIGINAL source of a later one.  Device formulation: rounds; a factor
// advances along its chain as long as every earlier factor it touches is final, otherwise it waits for the next
// round.  A final source is published as ONE 32-bit word (NOT_DONE until then), so no flag/data ordering is needed.
#include "stages.hpp"
#include "prim.hpp"

#include <stdlib.h>

namespace tdc {

__global__ void fstart_flag_kernel(const u32* __restrict__ flen, const u32* __restrict__ owner, size_t n, u32* __restrict__ flag) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 o = owner[p];                                     // a factor starts where the covering factor changes
    flag[p] = (flen[p] != 0 && o != NONE32 && (p == 0 || owner[p - 1] != o)) ? 1u : 0u;
}
__global__ void fstart_scatter_kernel(const u32* __restrict__ flen, const u32* __restrict__ owner, const u32* __restrict__ fsrc,
                                      const u32* __restrict__ offs, size_t n, size_t cap, u32* __restrict__ pos,
                                      u32* __restrict__ src, u32* __restrict__ len) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 l = flen[p];
    const u32 ow = owner[p];
    if (l != 0 && ow != NONE32 && (p == 0 || owner[p - 1] != ow)) {
        const u32 o = offs[p];
        if (o < cap) {
            pos[o] = (u32)p;
            if (src) src[o] = fsrc[p];
            if (len) len[o] = l;
        }
    }
}

size_t extract_factors(Ctx& c, size_t n, FactorSpace fs, u32* pos, u32* src, u32* len, size_t cap) {
    if (n == 0) return 0;
    if (fs.owner_rem_bits) throw HipError{hipErrorUnknown, "extract_factors: owner[] carries remainder bits", (int)__LINE__};
    const size_t mark = c.arena.mark();
    u32* offs = c.arena.get<u32>(n);
    u32* d_total = c.arena.get<u32>(1);
    const unsigned gn = cdiv(n, 256);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * 12);
        fstart_flag_kernel<<<gn, 256, 0, c.stream>>>(fs.flen, fs.owner, n, offs);
        LAUNCH_CHECK();
    }
    exclusive_sum_u32(c, offs, offs, n, d_total);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * 12);
        fstart_scatter_kernel<<<gn, 256, 0, c.stream>>>(fs.flen, fs.owner, fs.fsrc, offs, n, cap, pos, src, len);
        LAUNCH_CHECK();
    }
    const size_t z = c.read(d_total);
    c.arena.release(mark);
    return z;
}

// ---- owner[] from the factor starts: owner[q] = index (in position order) of the factor covering q ------------------
// Two streaming passes over flen[] (a factor starts at p iff flen[p] != 0; factors are disjoint):
//   owner_count_kernel : number of starts per tile of 4096 positions            (reads 4 B per position)
//   owner_build_kernel : the tile's starts get consecutive ranks (tile base + rank inside the tile), their positions go to pos[],
//                        and every position of the tile gets the rank of the covering factor or NONE32 -- last start at or before
//                        it by a max-scan, "covered" by comparing with that start's length (reads 4 B, writes 4 B per position);
//                        a factor that extends beyond its tile is recorded and
//   owner_cross_kernel : fills the part of such a factor that lies in the following tiles (they hold no start there).
// (Before: flag array, scan, scatter of the starts and a fill pass -- 72 B of traffic per position.)
constexpr int OW_T = 256, OW_PER = 16, OW_TILE = OW_T * OW_PER;
// FactorSpace::owner_rem_bits for z factors: the bits the ranks 0 .. z - 1 leave free in a 32-bit word that must not read NONE32 (at most cap <= 8)
__host__ __device__ __forceinline__ u32 owner_rem_bits_for(u32 z, u32 cap) {
    u32 b = 0;
    while (b < cap && (((u64)z + 1) >> (31 - b)) == 0) ++b;       // (b + 1 bits are free iff z + 1 < 2^(31 - b))
    return b;
}
__device__ __forceinline__ u32 ow_idx(u32 i) { return i + (i >> 4); }      // LDS slot of tile position i: a thread reads 16 consecutive positions

__global__ __launch_bounds__(OW_T) void owner_count_kernel(const u32* __restrict__ flen, const u8* __restrict__ flen8, size_t n, u32* __restrict__ tilecnt) {
    __shared__ u32 sm[OW_T / 64];
    const size_t base = (size_t)blockIdx.x * OW_TILE;
    u32 cnt = 0;
    if (flen8) {                                                // compact lengths: 16 positions per thread in one 16-byte load
        const size_t p = base + (size_t)threadIdx.x * OW_PER;
        if (p + OW_PER <= n) {
            const uint4 v = *(const uint4*)(flen8 + p);
            const u32 w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // bytes that are not zero
                const u32 w = w4[k];
                const u32 nz = ((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w;
                cnt += (u32)__popc(nz & 0x80808080u);
            }
        } else {
            for (int k = 0; k < OW_PER; ++k) if (p + k < n && flen8[p + k] != 0) ++cnt;
        }
    } else {
#pragma unroll
        for (int r = 0; r < OW_PER / 4; ++r) {
            const size_t p = base + ((size_t)r * OW_T + threadIdx.x) * 4;
            if (p + 4 <= n) {
                const uint4 v = *(const uint4*)(flen + p);
                cnt += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
            } else {
                for (int k = 0; k < 4; ++k) if (p + k < n && flen[p + k] != 0) ++cnt;
            }
        }
    }
    cnt = wave_reduce_sum(cnt);
    if (lane_id() == 0) sm[wave_id()] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < OW_T / 64; ++i) t += sm[i]; tilecnt[blockIdx.x] = t; }
}

__global__ __launch_bounds__(OW_T) void owner_build_kernel(const u32* __restrict__ flen, const u8* __restrict__ flen8, size_t n, const u32* __restrict__ tilebase,
                                                           u32* __restrict__ pos, u32* __restrict__ owner, uint2* __restrict__ cross,
                                                           u8* __restrict__ cls, u32* __restrict__ lenl, const u32* __restrict__ rem_total, u32 rem_cap) {
    __shared__ u32 s[OW_TILE + OW_TILE / 16 + 16];
    __shared__ u32 sm[OW_T / 64 + 1], smx[OW_T / 64];
    const size_t base = (size_t)blockIdx.x * OW_TILE;
    const int lane = lane_id(), w = wave_id();
    if (flen8) {                                                // compact lengths: a thread's 16 positions are one 16-byte load (255: flen[p])
        const u32 i = threadIdx.x * OW_PER;
        const size_t p = base + i;
        u32 w4[4] = { 0, 0, 0, 0 };
        if (p + OW_PER <= n) { const uint4 v = *(const uint4*)(flen8 + p); w4[0] = v.x; w4[1] = v.y; w4[2] = v.z; w4[3] = v.w; }
        else for (int k = 0; k < OW_PER; ++k) if (p + k < n) w4[k >> 2] |= (u32)flen8[p + k] << (8 * (k & 3));
#pragma unroll
        for (int k = 0; k < OW_PER; ++k) {
            u32 l = (w4[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            if (l == 255u) l = flen[p + k];
            s[ow_idx(i + k)] = l;
        }
    } else {
#pragma unroll
    for (int r = 0; r < OW_PER / 4; ++r) {                      // coalesced loads into LDS
        const u32 i = ((u32)r * OW_T + threadIdx.x) * 4;
        const size_t p = base + i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p + 4 <= n) v = *(const uint4*)(flen + p);
        else {
            if (p < n) v.x = flen[p];
            if (p + 1 < n) v.y = flen[p + 1];
            if (p + 2 < n) v.z = flen[p + 2];
        }
        const u32 q = ow_idx(i);                                // i is a multiple of 4: the four slots are consecutive
        s[q] = v.x; s[q + 1] = v.y; s[q + 2] = v.z; s[q + 3] = v.w;
    }
    }
    __syncthreads();
    const u32 l0 = threadIdx.x * OW_PER;
    u32 f[OW_PER];
    u32 cnt = 0, mylast = 0;                                    // mylast: 1 + tile position of the thread's last start (0: none)
#pragma unroll
    for (int k = 0; k < OW_PER; ++k) {
        f[k] = s[ow_idx(l0 + k)];
        if (f[k] != 0) { ++cnt; mylast = l0 + k + 1; }
    }
    u32 total;
    const u32 excl = block_exclusive_sum<u32, OW_T / 64>(cnt, sm, total);
    // last start in the earlier threads of the tile (exclusive max-scan)
    const u32 inc = wave_inclusive_max(mylast);
    if (lane == 63) smx[w] = inc;
    __syncthreads();
    u32 cur_start = __shfl_up(inc, 1, 64);
    if (lane == 0) cur_start = 0;
    for (int i = 0; i < w; ++i) cur_start = max(cur_start, smx[i]);
    u32 rank = tilebase[blockIdx.x] + excl;                     // rank of the thread's next start
    u32 cur_rank = rank - 1;                                    // rank of the start at cur_start (if any)
    u32 cur_len = cur_start ? s[ow_idx(cur_start - 1)] : 0u;
    u32 out[OW_PER];
    // (round 6: a thread's 16 positions hold 1.2 factor starts on average; storing their list entries from the loop below meant 32 store
    //  instructions per wave with a few lanes each.  With at most half a tile of starts -- always, for a threshold of 2 and more -- they
    //  go through the LDS tile once the owner words have left it, and leave as whole lines)
    const bool stage_starts = total <= (u32)OW_TILE / 2;
    const u32 remb = rem_total ? owner_rem_bits_for(*rem_total, rem_cap) : 0u;     // FactorSpace::owner_rem_bits (the scan has written the total)
    const u32 qmax = (1u << remb) - 1u, rsh = (32u - remb) & 31u;         // (remb == 0: qmax == 0, the shifted field is 0)
#pragma unroll
    for (int k = 0; k < OW_PER; ++k) {
        const u32 pl = l0 + k;
        if (f[k] != 0) {
            cur_start = pl + 1; cur_len = f[k]; cur_rank = rank++;
            if (!stage_starts && base + pl < n) { pos[cur_rank] = (u32)(base + pl); if (lenl) lenl[cur_rank] = f[k]; }
        }
        const u32 d = pl - (cur_start - 1);                             // distance from the start of the last factor at or before pl
        out[k] = (cur_start != 0 && d < cur_len) ? (cur_rank | (min(cur_len - d - 1u, qmax) << rsh)) : NONE32;
    }
    if (cls) {                                                  // class bytes of the thread's 16 positions (FactorSpace::cls)
        u32 w4[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < OW_PER; ++k) {
            const u32 cl = (out[k] == NONE32) ? 0u : (f[k] != 0 ? 2u : 3u);
            w4[k >> 2] |= cl << (8 * (k & 3));
        }
        const size_t p = base + l0;
        if (p + OW_PER <= n) *(uint4*)(cls + p) = make_uint4(w4[0], w4[1], w4[2], w4[3]);      // (cls is 16-byte aligned, l0 a multiple of 16)
        else for (int k = 0; k < OW_PER; ++k) if (p + k < n) cls[p + k] = (u8)(w4[k >> 2] >> (8 * (k & 3)));
    }
    if (threadIdx.x == OW_T - 1) {                              // the tile's last factor may extend into the following tiles
        const u32 end = cur_start ? cur_start - 1 + cur_len : 0u;          // (lengths are < 2^31)
        cross[blockIdx.x] = (end > (u32)OW_TILE) ? make_uint2(cur_rank, end - (u32)OW_TILE) : make_uint2(NONE32, 0u);
    }
    __syncthreads();                                            // every thread has read what it needs from the tile
#pragma unroll
    for (int k = 0; k < OW_PER; ++k) s[ow_idx(l0 + k)] = out[k];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < OW_PER / 4; ++r) {                      // coalesced stores
        const u32 i = ((u32)r * OW_T + threadIdx.x) * 4;
        const size_t p = base + i;
        const u32 q = ow_idx(i);
        if (p + 4 <= n) *(uint4*)(owner + p) = make_uint4(s[q], s[q + 1], s[q + 2], s[q + 3]);
        else for (int k = 0; k < 4; ++k) if (p + k < n) owner[p + k] = s[q + k];
    }
    if (stage_starts && total) {
        __syncthreads();                                        // the owner words have left the tile
        u32 o = excl;
#pragma unroll
        for (int k = 0; k < OW_PER; ++k) if (f[k] != 0) { s[o] = l0 + k; s[OW_TILE / 2 + o] = f[k]; ++o; }
        __syncthreads();
        const u32 tb = tilebase[blockIdx.x];
        for (u32 e = threadIdx.x; e < total; e += OW_T) {
            const size_t p = base + s[e];
            if (p < n) { pos[tb + e] = (u32)p; if (lenl) lenl[tb + e] = s[OW_TILE / 2 + e]; }
        }
    }
}

__global__ __launch_bounds__(256) void owner_cross_kernel(const uint2* __restrict__ cross, size_t n, u32* __restrict__ owner, u8* __restrict__ cls,
                                                          const u32* __restrict__ rem_total, u32 rem_cap) {
    const uint2 cr = cross[blockIdx.x];
    if (cr.y == 0) return;
    const u32 remb = rem_total ? owner_rem_bits_for(*rem_total, rem_cap) : 0u;
    const u32 qmax = (1u << remb) - 1u, rsh = (32u - remb) & 31u;
    const size_t start = ((size_t)blockIdx.x + 1) * OW_TILE;
    for (size_t j = threadIdx.x; j < cr.y && start + j < n; j += 256) {
        owner[start + j] = cr.x | (min(cr.y - (u32)j - 1u, qmax) << rsh);
        if (cls) cls[start + j] = (u8)3;
    }
}

void build_owner(Ctx& c, size_t n, FactorSpace& fs) {
    fs.have_list = false;
    fs.have_cls = false;
    fs.owner_rem_bits = 0;
    if (n == 0) return;
    const size_t mark = c.arena.mark();
    const u32 tiles = cdiv(n, OW_TILE);
    u32* tilecnt = c.arena.get<u32>(tiles);
    uint2* cross = (uint2*)c.arena.alloc((size_t)tiles * sizeof(uint2));
    u32* pos = fs.fpos ? fs.fpos : c.arena.get<u32>(n);
    u32* d_total = c.arena.get<u32>(1);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * (fs.flen8 ? 1 : 4));
        owner_count_kernel<<<tiles, OW_T, 0, c.stream>>>(fs.flen, fs.flen8, n, tilecnt);
        LAUNCH_CHECK();
    }
    exclusive_sum_u32(c, tilecnt, tilecnt, tiles, d_total);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * (fs.flen8 ? 6 : 9));
        const u32* rem_total = fs.want_owner_rem ? d_total : nullptr;
        const u32 rem_cap = std::min<u32>(fs.want_owner_rem, 8u);
        owner_build_kernel<<<tiles, OW_T, 0, c.stream>>>(fs.flen, fs.flen8, n, tilecnt, pos, fs.owner, cross, fs.cls, fs.fpos ? fs.flenl : nullptr, rem_total, rem_cap);
        LAUNCH_CHECK();
        owner_cross_kernel<<<tiles, 256, 0, c.stream>>>(cross, n, fs.owner, fs.cls, rem_total, rem_cap);
        fs.have_cls = fs.cls != nullptr;
        LAUNCH_CHECK();
    }
    const size_t z = c.read(d_total);
    if (fs.fpos) { fs.nfact = z; fs.have_list = true; }
    fs.owner_rem_bits = fs.want_owner_rem ? owner_rem_bits_for((u32)z, std::min<u32>(fs.want_owner_rem, 8u)) : 0u;
    c.arena.release(mark);
}

__global__ void expand_flen8_kernel(const u8* __restrict__ flen8, size_t n, u32* __restrict__ flen) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 l = flen8[p];
    if (l != 255u) flen[p] = l;                                 // (255: flen[p] already holds the length)
}
void expand_flen8(Ctx& c, size_t n, FactorSpace& fs) {
    if (!fs.flen8) return;
    if (n) { expand_flen8_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(fs.flen8, n, fs.flen); LAUNCH_CHECK(); }
    fs.flen8 = nullptr;
}

__global__ void fspace_clear_kernel(size_t n, u32* __restrict__ flen, u32* __restrict__ owner) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flen[p] = 0;
    owner[p] = NONE32;
}
__global__ void fspace_scatter_kernel(const u32* __restrict__ pos, const u32* __restrict__ src, const u32* __restrict__ len,
                                      size_t z, size_t n, u32* __restrict__ flen, u32* __restrict__ owner, u32* __restrict__ fsrc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= z) return;
    const u32 p = pos[i], l = len[i];
    if ((size_t)p + l > n) return;        // validated on the host; never write out of bounds
    flen[p] = l;
    fsrc[p] = src[i];
    for (u32 j = 0; j < l; ++j) owner[p + j] = (u32)i;
}

void scatter_factors(Ctx& c, size_t n, const u32* pos, const u32* src, const u32* len, size_t z, FactorSpace fs) {
    if (n == 0) return;
    fspace_clear_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(n, fs.flen, fs.owner);
    LAUNCH_CHECK();
    if (z) {
        fspace_scatter_kernel<<<cdiv(z, 256), 256, 0, c.stream>>>(pos, src, len, z, n, fs.flen, fs.owner, fs.fsrc);
        LAUNCH_CHECK();
    }
}

// ------------------------------------------------------------------------------------------------------------
constexpr u32 NOT_DONE = 0xFFFFFFFFu;

struct FlattenScalars { u32 waiting; u32 num_flattened; u32 max_depth; u32 pad; unsigned long long steps, waits; };

// Everything a chain step needs about a factor lives in ONE 16-byte record, indexed by the factor's rank r in position order:
//   rec[r] = { pos, len, original source, final source (NOT_DONE until known) }
// (one scattered line per step besides owner[src]; separate arrays for (pos, len), the original and the final source cost a line
// each.  Finding the covering factor through a table of the first factor per 64-position block plus a scan of consecutive
// records -- no owner[] line at all -- was built and measured slower: 6.6 vs 5.7 ms, the scan is a chain of dependent loads.)
// A factor that has to wait for the next round travels as a 16-byte work item { rank, len, current source, depth }: the
// rounds read and write their work lists sequentially (appended per wave, in any order -- the result of a factor does not
// depend on the order in which the waiting ones are visited), nothing about a waiting factor is gathered again.
// the source of the factor that starts at p when there is no Phi array (FactorSpace::src_prio): SA[ISA[p] - 1], ds/PhiFromSA.hpp:35-45
__device__ __forceinline__ u32 lazy_source(const u32* __restrict__ prio, const u32* __restrict__ sa, size_t n, const u32* __restrict__ fsrc, u32 p) {
    const u32 r = prio[p];
    return r < (u32)n ? (r ? sa[r - 1] : sa[n - 1]) : fsrc[p];
}
__global__ void flatten_init_kernel(const u32* __restrict__ fpos, size_t z, const u32* __restrict__ flen, const u32* __restrict__ orig,
                                    uint4* __restrict__ rec, const u32* __restrict__ lenl, const u32* __restrict__ src_prio,
                                    const u32* __restrict__ src_sa, size_t src_n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= z) return;
    const u32 p = fpos[i];
    const u32 o = src_prio ? lazy_source(src_prio, src_sa, src_n, orig, p) : orig[p];
    rec[i] = make_uint4(p, lenl ? lenl[i] : flen[p], o, NOT_DONE);     // (lenl: the lengths in list order -- one scattered line less per factor)
}
__global__ void sources_fill_kernel(const u32* __restrict__ fpos, size_t z, const u32* __restrict__ src_prio, const u32* __restrict__ src_sa,
                                    size_t src_n, u32* fsrc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= z) return;
    const u32 p = fpos[i];
    fsrc[p] = lazy_source(src_prio, src_sa, src_n, fsrc, p);
}
__global__ void sources_fill_pos_kernel(const u32* __restrict__ flen, size_t n, const u32* __restrict__ src_prio, const u32* __restrict__ src_sa,
                                        size_t src_n, u32* fsrc) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && flen[p]) fsrc[p] = lazy_source(src_prio, src_sa, src_n, fsrc, (u32)p);
}

// One round over the still-waiting factors.  FIRST: every factor, state taken from its record; else the items of `work`.
// A thread walks FL_K chains side by side: per step first the owner[] words of all of them, then the records -- FL_K independent
// scattered loads in flight per thread instead of one.  The factors of a workgroup that have to wait are appended to the next
// list with ONE atomic per workgroup (per wave it would be millions of atomics on one address: ~25 ns each, serialised).
#ifndef TDC_FL_K
#define TDC_FL_K 4
#endif
constexpr int FL_K = TDC_FL_K;
template <bool FIRST, bool REM>    // REM: owner words carry rem_bits remainder bits above the rank (FactorSpace::owner_rem_bits >= 1)
__global__ __launch_bounds__(256) void flatten_round_kernel(const uint4* __restrict__ work, u32 nwork, size_t n, const u32* __restrict__ owner,
                                                             uint4* rec, uint4* __restrict__ next, FlattenScalars* __restrict__ sc, u32 max_steps, u32 rem_bits) {
    const u32 rsh = 32u - rem_bits, qmax = (1u << rem_bits) - 1u, rmask = REM ? ((1u << rsh) - 1u) : 0xFFFFFFFFu;
    __shared__ u32 sm[5];
    __shared__ u32 s_base;
    const u32 base = blockIdx.x * (256u * FL_K);
    u32 fi[FL_K], len[FL_K], src[FL_K], dep[FL_K];
    u32 act = 0, fin = 0, valid = 0;             // bit r: chain r is still walking / finished / exists
#pragma unroll
    for (int r = 0; r < FL_K; ++r) {
        const u32 j = base + (u32)r * 256u + threadIdx.x;
        fi[r] = 0; len[r] = 0; src[r] = 0; dep[r] = 0;
        if (j < nwork) {
            if (FIRST) { const uint4 me = rec[j]; fi[r] = j; len[r] = me.y; src[r] = me.z; }
            else { const uint4 it = work[j]; fi[r] = it.x; len[r] = it.y; src[r] = it.z; dep[r] = it.w; }
            valid |= 1u << r;
        }
    }
    act = valid;
#ifdef TDC_FL_PROF
    u32 psteps = 0, pwaits = 0;
#endif
    for (u32 step = 0; step < max_steps && __any(act != 0); ++step) {   // a long chain continues in the next round, regrouped with its peers
        u32 rr[FL_K];
#pragma unroll
        for (int r = 0; r < FL_K; ++r) {
            rr[r] = NONE32;
            if (act & (1u << r)) {
                if ((size_t)src[r] >= n || dep[r] >= n) { act &= ~(1u << r); fin |= 1u << r; }   // :106 src < fmap.size()  (dep bound: no endless chains)
                else rr[r] = owner[src[r]];
            }
        }
#ifdef TDC_FL_PROF
        psteps += (u32)__popc(act);
#endif
        uint4 sr[FL_K];
#pragma unroll
        for (int r = 0; r < FL_K; ++r) {
            sr[r] = make_uint4(0, 0, 0, 0);
            if (act & (1u << r)) {
                if (rr[r] == NONE32) { act &= ~(1u << r); fin |= 1u << r; }                       // :106 fmap[src] == 0
                else {
                    if (REM) {                                                                  // the covering factor ends q + 1 positions on (q < qmax)
                        const u32 q = rr[r] >> rsh;
                        rr[r] &= rmask;
                        if (q < qmax && len[r] > q + 1u) { act &= ~(1u << r); fin |= 1u << r; continue; }   // :110 without the record
                    }
                    sr[r] = rec[rr[r]];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < FL_K; ++r) {
            if (!(act & (1u << r))) continue;
            const u32 d = src[r] - sr[r].x;
            if ((u64)d + len[r] > sr[r].y) { act &= ~(1u << r); fin |= 1u << r; continue; }      // :110 copy does not fit inside s
            u32 ssrc;
            if (rr[r] < fi[r]) {                                            // earlier factor: needs its final source
                ssrc = sr[r].w;
                if (ssrc == NOT_DONE) { act &= ~(1u << r);
#ifdef TDC_FL_PROF
                    ++pwaits;
#endif
                    continue; }      // wait for the next round
            } else {
                ssrc = sr[r].z;                                             // later factor: still unflattened at this point
            }
            src[r] = ssrc + d;                                              // :111
            ++dep[r];
        }
    }
#ifdef TDC_FL_PROF
    { const u32 a = wave_reduce_sum(psteps), b = wave_reduce_sum(pwaits);
      if (lane_id() == 0) { atomicAdd(&sc->steps, (unsigned long long)a); atomicAdd(&sc->waits, (unsigned long long)b); } }
#endif
    u32 nflat = 0, mxdep = 0;
#pragma unroll
    for (int r = 0; r < FL_K; ++r) {
        if (fin & (1u << r)) {
            ((u32*)&rec[fi[r]])[3] = src[r];                                // :122-124 (dep == 0: src is still the original source)
            if (dep[r]) { ++nflat; mxdep = max(mxdep, dep[r]); }
        }
    }
    // the waiting factors of the workgroup go to the next list
    const u32 wmask = valid & ~fin;
    u32 total;
    u32 off = block_exclusive_sum<u32, 4>((u32)__popc(wmask), sm, total);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&sc->waiting, total) : 0u;
    __syncthreads();
    off += s_base;
#pragma unroll
    for (int r = 0; r < FL_K; ++r)
        if (wmask & (1u << r)) next[off++] = make_uint4(fi[r], len[r], src[r], dep[r]);
    // statistics: one atomic pair per wave
    nflat = wave_reduce_sum(nflat);
    mxdep = wave_reduce_max(mxdep);
    if (lane_id() == 0 && nflat) { atomicAdd(&sc->num_flattened, nflat); atomicMax(&sc->max_depth, mxdep); }
}

__global__ void flatten_commit_kernel(size_t z, const uint4* __restrict__ rec, u32* __restrict__ fsrc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= z) return;
    const uint4 q = rec[i];
    if (q.w != q.z) fsrc[q.x] = q.w;              // only the flattened ones moved
}

void materialize_sources(Ctx& c, size_t n, FactorSpace& fs) {
    if (!fs.src_prio) return;
    if (n) {
        if (fs.have_list) { if (fs.nfact) sources_fill_kernel<<<cdiv(fs.nfact, 256), 256, 0, c.stream>>>(fs.fpos, fs.nfact, fs.src_prio, fs.src_sa, fs.src_n, fs.fsrc); }
        else sources_fill_pos_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(fs.flen, n, fs.src_prio, fs.src_sa, fs.src_n, fs.fsrc);
        LAUNCH_CHECK();
    }
    fs.src_prio = nullptr;
}

void flatten_factors(Ctx& c, size_t n, FactorSpace fs, FlattenStats* st, const std::function<void(int)>& between, void* rec_keep) {
    if (fs.src_prio && !rec_keep) materialize_sources(c, n, fs);       // (the commit pass only writes the sources that moved)
    FlattenStats local;
    if (!st) st = &local;
    *st = FlattenStats();
    if (n == 0) { if (between) between(0); return; }
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    u32* fpos = fs.have_list ? fs.fpos : c.arena.get<u32>(n);
    const size_t z = fs.have_list ? fs.nfact : extract_factors(c, n, fs, fpos, nullptr, nullptr, n);
    if (z == 0) { if (between) between(0); c.arena.release(mark); return; }
    uint4* rec = rec_keep ? (uint4*)rec_keep : (uint4*)c.arena.alloc(z * sizeof(uint4));
    FlattenScalars* d_sc = (FlattenScalars*)c.arena.alloc(sizeof(FlattenScalars));
    HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(FlattenScalars), s));
    const unsigned gz = cdiv(z, 256);
    // (lazy sources are only left to this kernel by a caller that keeps the records: nothing is written back into fsrc[] then)
    flatten_init_kernel<<<gz, 256, 0, s>>>(fpos, z, fs.flen, fs.fsrc, rec, fs.have_list ? fs.flenl : nullptr, fs.src_prio, fs.src_sa, fs.src_n);
    LAUNCH_CHECK();
    // work lists of the still-waiting factors (the first round visits every factor)
    uint4* work[2] = { (uint4*)c.arena.alloc(z * sizeof(uint4)), nullptr };
    u32 waiting = (u32)z;
    int cur_w = -1;                               // -1: every factor (first round)
    // steps per round: few in the first rounds (most chains are short; the lanes of a wave wait for the longest one),
    // growing afterwards
    u32 max_steps = (u32)c.flatten_steps;   // (option flatten_steps; measured: 1,2,4,.. 8.5 ms; unlimited 11.2 ms)
    if (max_steps == 0) max_steps = 1u << 30;
    // budget growth per round (measured at 256 MiB: x2 8.4 ms, x4 7.4 ms, x8 7.0 ms)
    u32 flat_growth = (u32)c.flatten_growth;
    if (flat_growth < 2) flat_growth = 2;
    u32 stalled = 0;
    while (waiting) {
        const int nxt = cur_w < 0 ? 0 : (cur_w ^ 1);
        if (!work[nxt]) work[nxt] = (uint4*)c.arena.alloc((size_t)waiting * sizeof(uint4));   // (the second list: at most the survivors of round 1)
        HIP_TRY(hipMemsetAsync(&d_sc->waiting, 0, sizeof(u32), s));
        {   // per waiting factor: its item (16) + one chain step (owner word + record: 20) + item / final source out (16)
            Ctx::ProfScope prof(c, K_FLATTEN_ROUND, (u64)waiting * 52);
            const u32 rb = fs.owner_rem_bits;
            const unsigned g = cdiv(waiting, 256 * FL_K);
            if (cur_w < 0) { if (rb) flatten_round_kernel<true, true><<<g, 256, 0, s>>>(nullptr, waiting, n, fs.owner, rec, work[nxt], d_sc, max_steps, rb);
                             else flatten_round_kernel<true, false><<<g, 256, 0, s>>>(nullptr, waiting, n, fs.owner, rec, work[nxt], d_sc, max_steps, 0u); }
            else { if (rb) flatten_round_kernel<false, true><<<g, 256, 0, s>>>(work[cur_w], waiting, n, fs.owner, rec, work[nxt], d_sc, max_steps, rb);
                   else flatten_round_kernel<false, false><<<g, 256, 0, s>>>(work[cur_w], waiting, n, fs.owner, rec, work[nxt], d_sc, max_steps, 0u); }
            LAUNCH_CHECK();
        }
        // (this round is on its way: 150 M factors in the first one at 2e9 B, 55 M in the second, 9 M in the third; the later ones are too
        //  short to hide anything behind)
        if (between) between((int)st->rounds + 1);
        const u32 now = c.read(&d_sc->waiting);
        st->rounds++;
        if (c.level_log) {
            const FlattenScalars hs = c.read(d_sc);
            fprintf(stderr, "flatten round %u: %u waiting -> %u (budget %u steps; cumulative: %llu visits, %llu of them waits)\n", st->rounds, waiting, now, max_steps, hs.steps, hs.waits);
        }
        stalled = (now == waiting) ? stalled + 1 : 0;         // (a round with a small budget may finish nothing; never many in a row)
        if (now > waiting || (now == waiting && max_steps >= (1u << 30)) || stalled > 40)
            throw HipError{hipErrorUnknown, "flatten: rounds made no progress", (int)__LINE__};
        waiting = now;
        cur_w = nxt;
        if (max_steps < (1u << 30)) max_steps = (max_steps > (1u << 30) / flat_growth) ? (1u << 30) : max_steps * flat_growth;
    }
    if (between) between(0);
    if (!rec_keep) {                                   // (a caller that keeps the records reads the final sources there)
        flatten_commit_kernel<<<gz, 256, 0, s>>>(z, rec, fs.fsrc);
        LAUNCH_CHECK();
    }
    FlattenScalars h = c.read(d_sc);
    st->num_flattened = h.num_flattened;
    st->max_depth_lb = h.max_depth;
    c.arena.release(mark);
}

}  // namespace tdc
