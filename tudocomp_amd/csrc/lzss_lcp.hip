// lzss_lcp.hip -- LZSSLCPCompressor's factorization (compressors/LZSSLCPCompressor.hpp:60-115) on the GPU.
//
// Reference, for every text position i (sequentially, skipping over emitted factors): walk the suffix array upwards from
// isa[i] to the first suffix that starts EARLIER in the text (sa[r'] < i), taking the minimum LCP on the way; the same
// downwards; the longer of the two common prefixes wins (ties: the upward side); if it reaches `threshold`, emit the
// factor (i, that suffix, length) and continue at i + length, else at i + 1.
//
// Device formulation:
//   1. ANSV: previous / next SMALLER VALUE of every suffix-array entry, with a three-level minimum hierarchy
//      (groups of 32, tiles of 1024, super-tiles of 2^20 entries): the expected distance to the nearest smaller value
//      is logarithmic, so most queries end inside their own group; the rest skip whole groups / tiles by their minima.
//   2. the two common-prefix lengths by direct comparison with the chunked carry of the PLCP kernel
//      (len[i] >= len[i-1] - 1 holds for either side: shifting the previous position's match by one gives a candidate).
//   3. the greedy parse is the orbit of position 0 under next(i) = i + (len >= threshold ? len : 1).  It is marked
//      hierarchically: exit of every position from its 1024-tile (pointer doubling in LDS), exit from its super-tile
//      (right-to-left sweep over the tiles of a super-tile), a serial walk over the few super-tile entries, then the
//      tile entries and finally the chain positions inside every tile, each level in parallel.
#include "stages.hpp"
#include "prim.hpp"

namespace tdc {

constexpr int ANSV_TILE = 1024, ANSV_SUPER = 1 << 20;     // plus groups of 32 entries inside a tile

// ---- 1. ANSV ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ansv_mins_kernel(const u32* __restrict__ sa, size_t n, u32* __restrict__ gmin, u32* __restrict__ tmin) {
    __shared__ u32 s[ANSV_TILE];
    __shared__ u32 g[32];
    const size_t base = (size_t)blockIdx.x * ANSV_TILE;
    for (int k = threadIdx.x; k < ANSV_TILE; k += 256) s[k] = (base + k < n) ? sa[base + k] : NONE32;
    __syncthreads();
    if (threadIdx.x < 32) {
        u32 m = NONE32;
        for (int k = 0; k < 32; ++k) m = min(m, s[threadIdx.x * 32 + k]);
        g[threadIdx.x] = m;
        if (base + (size_t)threadIdx.x * 32 < n) gmin[base / 32 + threadIdx.x] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = NONE32;
        for (int k = 0; k < 32; ++k) m = min(m, g[k]);
        tmin[blockIdx.x] = m;
    }
}
__global__ void ansv_smin_kernel(const u32* __restrict__ tmin, size_t ntiles, u32* __restrict__ smin) {
    const size_t sidx = blockIdx.x;                        // one workgroup per super-tile
    __shared__ u32 sm[4];
    u32 m = NONE32;
    const size_t t0 = sidx * (ANSV_SUPER / ANSV_TILE);
    for (size_t t = t0 + threadIdx.x; t < t0 + ANSV_SUPER / ANSV_TILE && t < ntiles; t += blockDim.x) m = min(m, tmin[t]);
    m = wave_reduce_min(m);
    if (lane_id() == 0) sm[wave_id()] = m;
    __syncthreads();
    if (threadIdx.x == 0) smin[sidx] = min(min(sm[0], sm[1]), min(sm[2], sm[3]));
}

struct MinTree { const u32* sa; const u32* gmin; const u32* tmin; const u32* smin; size_t n; };

__device__ __forceinline__ u32 last_smaller_in_group(const MinTree& T, size_t gg, u32 x) {
    size_t q = gg * 32 + 31;
    if (q > T.n - 1) q = T.n - 1;
    for (;; --q) { if (T.sa[q] < x) return (u32)q; if (q == gg * 32) break; }
    return NONE32;
}
__device__ __forceinline__ u32 last_smaller_in_tile(const MinTree& T, size_t tt, u32 x) {
    size_t gg = tt * 32 + 31;
    const size_t glast = (T.n - 1) / 32;
    if (gg > glast) gg = glast;
    for (;; --gg) { if (T.gmin[gg] < x) return last_smaller_in_group(T, gg, x); if (gg == tt * 32) break; }
    return NONE32;
}
__device__ u32 prev_smaller(const MinTree& T, size_t r, u32 x) {
    for (size_t q = r; q > (r & ~(size_t)31);) { --q; if (T.sa[q] < x) return (u32)q; }
    for (size_t gg = r >> 5; gg > ((r >> 10) << 5);) { --gg; if (T.gmin[gg] < x) return last_smaller_in_group(T, gg, x); }
    for (size_t tt = r >> 10; tt > ((r >> 20) << 10);) { --tt; if (T.tmin[tt] < x) return last_smaller_in_tile(T, tt, x); }
    for (size_t ss = r >> 20; ss > 0;) {
        --ss;
        if (T.smin[ss] < x) {
            size_t tt = ss * 1024 + 1023;
            for (;; --tt) { if (T.tmin[tt] < x) return last_smaller_in_tile(T, tt, x); if (tt == ss * 1024) break; }
        }
    }
    return NONE32;
}
__device__ __forceinline__ u32 first_smaller_in_group(const MinTree& T, size_t gg, u32 x) {
    for (size_t q = gg * 32; q < gg * 32 + 32 && q < T.n; ++q) if (T.sa[q] < x) return (u32)q;
    return NONE32;
}
__device__ __forceinline__ u32 first_smaller_in_tile(const MinTree& T, size_t tt, u32 x) {
    const size_t ng = (T.n + 31) / 32;
    for (size_t gg = tt * 32; gg < tt * 32 + 32 && gg < ng; ++gg) if (T.gmin[gg] < x) return first_smaller_in_group(T, gg, x);
    return NONE32;
}
__device__ u32 next_smaller(const MinTree& T, size_t r, u32 x) {
    const size_t ng = (T.n + 31) / 32, nt = (T.n + 1023) / 1024, ns = (T.n + ANSV_SUPER - 1) / ANSV_SUPER;
    for (size_t q = r + 1; q < ((r >> 5) + 1) * 32 && q < T.n; ++q) if (T.sa[q] < x) return (u32)q;
    for (size_t gg = (r >> 5) + 1; gg < ((r >> 10) + 1) * 32 && gg < ng; ++gg) if (T.gmin[gg] < x) return first_smaller_in_group(T, gg, x);
    for (size_t tt = (r >> 10) + 1; tt < ((r >> 20) + 1) * 1024 && tt < nt; ++tt) if (T.tmin[tt] < x) return first_smaller_in_tile(T, tt, x);
    for (size_t ss = (r >> 20) + 1; ss < ns; ++ss) {
        if (T.smin[ss] < x) {
            for (size_t tt = ss * 1024; tt < ss * 1024 + 1024 && tt < nt; ++tt) if (T.tmin[tt] < x) return first_smaller_in_tile(T, tt, x);
        }
    }
    return NONE32;
}

// psrc[i] / nsrc[i] = text position of the previous / next suffix-array neighbour of suffix i that starts before i
__global__ void ansv_query_kernel(MinTree T, u32* __restrict__ psrc, u32* __restrict__ nsrc) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= T.n) return;
    const u32 x = T.sa[r];
    const u32 pi = prev_smaller(T, r, x), ni = next_smaller(T, r, x);
    psrc[x] = (pi == NONE32) ? NONE32 : T.sa[pi];
    nsrc[x] = (ni == NONE32) ? NONE32 : T.sa[ni];
}

// ---- 3. candidate per position and the parse chain ---------------------------------------------------------------
// clen / csrc overwrite plen / psrc in place
__global__ void lzss_choose_kernel(size_t n, u32 threshold, u32* __restrict__ plen, const u32* __restrict__ nlen, u32* __restrict__ psrc,
                                   const u32* __restrict__ nsrc, u32* __restrict__ next) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 len = 0, src = NONE32;
    if (i + 1 < n) {                                     // the loop of the reference omits T[n-1] (:62)
        const u32 pl = plen[i], nl = nlen[i];
        const u32 best = max(pl, nl);                    // :99
        if (best >= threshold) { len = best; src = (best == pl) ? psrc[i] : nsrc[i]; }   // :101 ties go to the PSV side
    }
    plen[i] = len;
    psrc[i] = src;
    const u64 nx = (u64)i + (len ? len : 1u);            // :107 / :109
    next[i] = (u32)(nx > n ? n : nx);
}

__global__ void lzss_fspace_init_kernel(size_t n, u32* __restrict__ flen, u32* __restrict__ owner) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) { flen[p] = 0; owner[p] = NONE32; }
}
__global__ void lzss_emit_kernel(size_t n, const u8* __restrict__ mark, const u32* __restrict__ clen, const u32* __restrict__ csrc,
                                 u32* __restrict__ flen, u32* __restrict__ owner, u32* __restrict__ fsrc, u32* __restrict__ count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool f = false;
    if (i < n && mark[i]) {
        const u32 len = clen[i];
        if (len) {
            f = true;
            flen[i] = len;
            fsrc[i] = csrc[i];
            for (u32 j = 0; j < len && i + j < n; ++j) owner[i + j] = (u32)i;
        }
    }
    const u64 b = __ballot(f);
    if (b && lane_id() == __builtin_ctzll(b)) atomicAdd(count, (u32)__popcll(b));
}

// ---- lcpcomp(comp=plcppeaks) ------------------------------------------------------------------------------------------------
// lcpcomp::PLCPPeaksStrategy::factorize (compressors/lcpcomp/compress/PLCPPeaksStrategy.hpp:36-80): a left-to-right scan; a
// position whose PLCP value is a strict local maximum and >= threshold becomes the factor (i, Phi[i], PLCP[i]) and the scan
// jumps behind it.  `last_replacement_pos` of the reference only ever equals i at i = 0, so "peak" is a local property and
// the scan is the orbit of position 0 under next(i) = peak(i) ? i + PLCP[i] : i + 1 -- the same chain marking as the lzss_lcp
// parse.  The reference's PLCP array still holds Phi[n-1] in its last entry (ds/PLCPFromPhi.hpp:27-53) and the scan reads
// it at i = n-2; the device PLCP has 0 there, so that one comparison takes the value from Phi.
__device__ __forceinline__ bool plcp_peak(const u32* __restrict__ plcp, const u32* __restrict__ phi, size_t n, size_t i, u32 threshold) {
    if (i + 1 >= n) return false;
    const u32 v = plcp[i];
    if (v < threshold) return false;
    if (i != 0 && !(v > plcp[i - 1])) return false;
    const u32 right = (i + 1 == n - 1) ? phi[n - 1] : plcp[i + 1];
    return v > right;
}
__global__ void peaks_next_kernel(const u32* __restrict__ plcp, const u32* __restrict__ phi, size_t n, u32 threshold, u32* __restrict__ next) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    next[i] = plcp_peak(plcp, phi, n, i, threshold) ? (u32)(i + plcp[i]) : (u32)(i + 1);
}
__global__ void peaks_emit_kernel(const u32* __restrict__ plcp, const u32* __restrict__ phi, size_t n, u32 threshold, const u8* __restrict__ mark,
                                  u32* __restrict__ flen, u32* __restrict__ fsrc, u32* __restrict__ count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool f = false;
    if (i < n) {
        f = mark[i] && plcp_peak(plcp, phi, n, i, threshold);
        flen[i] = f ? plcp[i] : 0u;
        if (f) fsrc[i] = phi[i];
    }
    const u64 b = __ballot(f);
    if (b && lane_id() == __builtin_ctzll(b)) atomicAdd(count, (u32)__popcll(b));
}

void plcp_peaks_factorize(Ctx& c, size_t n, const u32* phi, const u32* plcp, u32 threshold, FactorSpace& fs, u64* nfactors) {
    *nfactors = 0;
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark0 = c.arena.mark();
    const unsigned gn = cdiv(n, 256);
    u32* next = c.arena.get<u32>(n);
    u32* s1 = c.arena.get<u32>(n), *s2 = c.arena.get<u32>(n);
    u8* mark = c.arena.get<u8>(n);
    u32* d_cnt = c.arena.get<u32>(1);
    peaks_next_kernel<<<gn, 256, 0, s>>>(plcp, phi, n, threshold, next);
    LAUNCH_CHECK();
    mark_orbit_u32(c, next, n, mark, s1, s2);
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(u32), s));
    peaks_emit_kernel<<<gn, 256, 0, s>>>(plcp, phi, n, threshold, mark, fs.flen, fs.fsrc, d_cnt);
    LAUNCH_CHECK();
    *nfactors = c.read(d_cnt);
    c.arena.release(mark0);
    build_owner(c, n, fs);
}

void lzss_lcp_factorize(Ctx& c, const u8* text, size_t n, const u32* sa, const u32* isa, u32 threshold, FactorSpace fs, LzssStats* st) {
    (void)isa;
    LzssStats local;
    if (!st) st = &local;
    *st = LzssStats();
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark0 = c.arena.mark();
    const unsigned gn = cdiv(n, 256);
    const size_t ntiles = (n + ANSV_TILE - 1) / ANSV_TILE, ngroups = (n + 31) / 32, nsuper = (n + ANSV_SUPER - 1) / ANSV_SUPER;

    // 1. ANSV over the suffix array
    u32* gmin = c.arena.get<u32>(ngroups);
    u32* tmin = c.arena.get<u32>(ntiles);
    u32* smin = c.arena.get<u32>(nsuper);
    u32* psrc = c.arena.get<u32>(n), *nsrc = c.arena.get<u32>(n);
    ansv_mins_kernel<<<(unsigned)ntiles, 256, 0, s>>>(sa, n, gmin, tmin);
    LAUNCH_CHECK();
    ansv_smin_kernel<<<(unsigned)nsuper, 256, 0, s>>>(tmin, ntiles, smin);
    LAUNCH_CHECK();
    MinTree T{sa, gmin, tmin, smin, n};
    ansv_query_kernel<<<gn, 256, 0, s>>>(T, psrc, nsrc);
    LAUNCH_CHECK();

    // 2. common-prefix lengths of both sides
    u32* plen = c.arena.get<u32>(n), *nlen = c.arena.get<u32>(n);
    u32* d_tmp = c.arena.get<u32>(4);
    build_lce_with_carry(c, text, n, psrc, plen, d_tmp);
    build_lce_with_carry(c, text, n, nsrc, nlen, d_tmp + 1);

    // 3. candidate per position, then the orbit of position 0
    u32* next = c.arena.get<u32>(n);
    lzss_choose_kernel<<<gn, 256, 0, s>>>(n, threshold, plen, nlen, psrc, nsrc, next);
    LAUNCH_CHECK();
    u32* clen = plen, *csrc = psrc;
    u8* mark = c.arena.get<u8>(n);
    mark_orbit_u32(c, next, n, mark, nsrc, nlen);        // nsrc / nlen are free now: scratch

    // 4. factors into position space
    HIP_TRY(hipMemsetAsync(d_tmp + 2, 0, sizeof(u32), s));
    lzss_fspace_init_kernel<<<gn, 256, 0, s>>>(n, fs.flen, fs.owner);
    LAUNCH_CHECK();
    lzss_emit_kernel<<<gn, 256, 0, s>>>(n, mark, clen, csrc, fs.flen, fs.owner, fs.fsrc, d_tmp + 2);
    LAUNCH_CHECK();
    st->factors = c.read(d_tmp + 2);
    c.arena.release(mark0);
}

}  // namespace tdc
