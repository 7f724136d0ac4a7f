// textds.hip -- Phi, PLCP and (debug) LCP arrays.
//   build_phi  : ds/PhiFromSA.hpp:35-45        phi[sa[i]] = sa[i-1], phi[sa[0]] = sa[n-1]
//   build_plcp : ds/PLCPFromPhi.hpp:27-53      plcp[i] = lcp(T[i..], T[phi[i]..]) for i < n-1 ; plcp[n-1] := 0
//   build_lcp  : ds/LCPFromPLCP.hpp:27-56      lcp[0] = 0, lcp[i] = plcp[sa[i]]
// The PLCP array is a function of the text alone, so the chunked evaluation below (each thread restarts the
// Phi-algorithm's carry l = 0 at the start of its chunk and then uses plcp[i+1] >= plcp[i] - 1) is bit-identical
// to the reference's sequential loop.
#include "stages.hpp"
#include "prim.hpp"

namespace tdc {

__global__ void phi_kernel(const u32* __restrict__ sa, size_t n, u32* __restrict__ phi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 s = sa[i];
    phi[s] = (i == 0) ? sa[n - 1] : sa[i - 1];
}

__global__ void phi_first_kernel(const u32* __restrict__ sa, size_t n, u32* __restrict__ phi) { phi[sa[0]] = sa[n - 1]; }

void build_phi(Ctx& c, const u32* sa, size_t n, u32* phi) {
    if (!n) return;
    if (c.bucket_scatter && n >= ((size_t)1 << 22)) {           // (timed under the radix and window_scatter classes)
        // pairs (sa[i], sa[i-1]), i = 1 .. n-1, partitioned by destination window, then scattered
        const size_t mark = c.arena.mark();
        u32* ti = c.arena.get<u32>(n);
        u32* tv = c.arena.get<u32>(n);
        u32* ti2 = c.arena.get<u32>(n);
        u32* tv2 = c.arena.get<u32>(n);
        bucketed_scatter_u32(c, sa + 1, sa, n - 1, phi, n, ti, tv, ti2, tv2, true);     // sa[1..n-1] = every position but n-1 (= sa[0]) once
        phi_first_kernel<<<1, 1, 0, c.stream>>>(sa, n, phi);
        LAUNCH_CHECK();
        c.arena.release(mark);
        return;
    }
    Ctx::ProfScope prof(c, K_PHI, (u64)n * 8);                  // read SA, scatter Phi
    phi_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(sa, n, phi);
    LAUNCH_CHECK();
}

#ifndef TDC_PLCP_CHUNK
#define TDC_PLCP_CHUNK 16
#endif
constexpr int PLCP_CHUNK = TDC_PLCP_CHUNK;                    // consecutive text positions per thread
constexpr int PLCP_TILE = 256 * PLCP_CHUNK;       // 8192 positions per workgroup
constexpr int PLCP_HALO = 512;                    // text bytes staged beyond the tile for the T[i+l] side

// A thread walks 16 consecutive positions (measured: 32 -> 7.8 ms, 16 -> 5.6 ms, 8 -> 7.3 ms at 2^28: the kernel is bound by the
// latency of its dependent text reads, and LDS per workgroup sets how many threads a CU holds; shorter chunks restart the
// carry more often), so its Phi reads / PLCP writes are strided by 64 B across the lanes of a
// wave; going through LDS (row-padded to 33 words: conflict-free) turns both into fully coalesced 1 KiB transfers,
// and the T[i+l] side of every comparison is served from a staged copy of the tile (+halo).  Only T[Phi[i]+l] stays
// a global (data-dependent) read.
// ALLOW_NONE: src[i] == NONE32 means "no source", result 0 (used by the lzss_lcp PSV/NSV sides, where the same
// lower bound len[i] >= len[i-1] - 1 holds).
// ---- sampled PLCP values, coarse to fine ----------------------------------------------------------------------------------
// PLCP[i] >= PLCP[b] - (i - b) for b < i.  The positions 1024 k are computed exactly, one wave each, in five levels of spacing
// 2^26, 2^22, 2^18, 2^14, 2^10: a sample starts from the bound given by the next coarser sample to its left.  A peak of height H
// is followed by at least H positions of ramp, so over all levels the comparisons that the bounds do not save add up to
// O(n) bytes; the top level (at most 32 samples) compares from scratch, 512 bytes per step and wave.
constexpr int PLCP_SAMPLE = 1024;

__device__ __forceinline__ u32 wave_lcp(const u8* __restrict__ text, size_t n, size_t i, size_t j, u32 l) {
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

template <bool ALLOW_NONE>
__global__ __launch_bounds__(256) void plcp_sample_kernel(const u8* __restrict__ text, size_t n, const u32* __restrict__ phi,
                                                           u32* __restrict__ samples, u32 nsamp, u32 step, u32 parent_step) {
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t k = w * step;
    if (k >= nsamp) return;
    if (parent_step && k % parent_step == 0) return;           // computed by a coarser level
    const size_t i = k * PLCP_SAMPLE;
    u32 l = 0;
    if (i + 1 < n && !(ALLOW_NONE && phi[i] == NONE32)) {     // no source: length 0 (the bound of such a position is never positive)
        if (parent_step) {
            const size_t kb = k - k % parent_step;
            const u64 d = (u64)(k - kb) * PLCP_SAMPLE;
            const u32 sb = samples[kb];
            l = sb > d ? (u32)(sb - d) : 0u;
        }
        l = wave_lcp(text, n, i, phi[i], l);
    }
    if (lane_id() == 0) samples[k] = l;
}

// The bounds are loose behind a jump (len[b] small, len[b+1] huge: the position behind a mismatching byte in front of a long
// repeat, or behind a position without source in the lzss_lcp passes): the one thread that owns the jump would walk the whole
// repeat byte by byte (250 ns per dependent load), and so would every later chunk start of that sample interval.  An interval
// [1024 k, 1024 (k+1)) can only hide a jump of more than 2048 if S[k+1] > S[k] + 1024 (values fall by at most one per
// position); those intervals -- and the last, open one -- are computed here, position by position with the carry, one wave per
// interval and 512 bytes per step; plcp_kernel then only copies them.
template <bool ALLOW_NONE>
__global__ __launch_bounds__(256) void plcp_refine_kernel(const u8* __restrict__ text, size_t n, const u32* __restrict__ src,
                                                           const u32* __restrict__ samples, u32 nsamp, u32* __restrict__ out,
                                                           u8* __restrict__ iflag, u32* __restrict__ d_max) {
    const size_t k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= nsamp) return;
    const u32 s0 = samples[k];
    if (k + 1 < nsamp && samples[k + 1] <= s0 + PLCP_SAMPLE) { if (lane_id() == 0) iflag[k] = 0; return; }
    const size_t b = k * PLCP_SAMPLE;
    u32 l = s0, mx = s0;
    if (lane_id() == 0) { iflag[k] = 1; out[b] = s0; }
    for (int t = 1; t < PLCP_SAMPLE; ++t) {
        const size_t i = b + (size_t)t;
        if (i >= n) break;
        if (l) --l;
        if (i + 1 >= n) l = 0;
        else {
            const u32 j = src[i];
            if (ALLOW_NONE && j == NONE32) l = 0;
            else l = wave_lcp(text, n, i, j, l);
        }
        if (lane_id() == 0) out[i] = l;
        mx = max(mx, l);
    }
    if (lane_id() == 0 && mx) atomicMax(d_max, mx);
}

// `samples` (nullable): exact results of the positions 1024 k (plcp_sample_kernel).  A chunk then starts from the lower bound
// sample - distance instead of 0: without it every chunk start of a text like a^N walks the whole repeat again
// (n * average LCP / 16 byte steps: 7 s for 16 MB of one letter).
template <bool ALLOW_NONE>
__global__ __launch_bounds__(256) void plcp_kernel(const u8* __restrict__ text, size_t n, const u32* __restrict__ phi,
                                                    u32* __restrict__ plcp, u32* __restrict__ d_max, const u32* __restrict__ samples,
                                                    const u8* __restrict__ iflag) {
    __shared__ u32 sphi[256 * (PLCP_CHUNK + 1)];
    __shared__ __attribute__((aligned(8))) u8 stext[PLCP_TILE + PLCP_HALO + 8];    // (+8: the funnel shift reads the next aligned word)
    const size_t base = (size_t)blockIdx.x * PLCP_TILE;
    for (int k = threadIdx.x; k < PLCP_TILE; k += 256) {
        const size_t p = base + k;
        sphi[(k / PLCP_CHUNK) * (PLCP_CHUNK + 1) + (k % PLCP_CHUNK)] = (p < n) ? phi[p] : 0u;
    }
    for (int k = threadIdx.x * 4; k < PLCP_TILE + PLCP_HALO; k += 1024) {       // 4 bytes per lane (base is 4-aligned)
        const size_t p = base + k;
        u32 wv = 0;
        if (p + 4 <= n) wv = *(const u32*)(text + p);
        else for (int b = 0; b < 4; ++b) if (p + b < n) wv |= (u32)text[p + b] << (8 * b);
        *(u32*)(stext + k) = wv;
    }
    __syncthreads();
    u32 mx = 0;
    u32* row = sphi + threadIdx.x * (PLCP_CHUNK + 1);
    const size_t begin = base + (size_t)threadIdx.x * PLCP_CHUNK;
    if (begin < n && samples && iflag[begin / PLCP_SAMPLE]) {
        for (int c = 0; c < PLCP_CHUNK && begin + c < n; ++c) { row[c] = plcp[begin + c]; }     // computed by plcp_refine_kernel (its maximum too)
    } else if (begin < n) {
        u32 l = 0;
        if (samples) {
            const size_t b = begin & ~(size_t)(PLCP_SAMPLE - 1);
            const u32 sb = samples[b / PLCP_SAMPLE], d = (u32)(begin - b);
            l = sb > d ? sb - d : 0u;                         // len[i] >= len[b] - (i - b)
        }
        // Eight bytes per step: the T[i+l] side from the staged text (two aligned LDS words, funnel-shifted), the T[Phi[i]+l] side
        // as one unaligned 8-byte word that stays in registers -- when Phi[i+1] = Phi[i] + 1 (the common case) the comparison of
        // position i+1 resumes at the very byte where that of position i stopped, i.e. inside the word already held (with ~1800
        // threads per CU neither L1 nor L2 would still have the line).
        uintptr_t w_addr = ~(uintptr_t)0xFF;                // address of the byte in bits 7..0 of w_val
        u64 w_val = 0;
        for (int c = 0; c < PLCP_CHUNK; ++c) {
            const size_t i = begin + c;
            if (i >= n) break;
            if (i == n - 1) { row[c] = 0; break; }
            const size_t j = row[c];
            if (ALLOW_NONE && row[c] == NONE32) { row[c] = 0; l = 0; continue; }
            const u32 li = (u32)(i - base);                 // offset of i inside the staged text
            // the sentinel T[n-1] is unique, so the comparison stops before either index leaves the text;
            // the explicit bounds only keep a corrupted Phi from faulting
            for (;;) {
                if (i + l >= n || j + l >= n) break;
                const u32 off = li + l;
                const size_t ri = n - (i + l), rj = n - (j + l);
                if (ri >= 8 && rj >= 8 && off + 8 <= (u32)(PLCP_TILE + PLCP_HALO)) {
                    const u32 ao = off & ~7u, as = (off & 7u) * 8u;
                    const u64 alo = *(const u64*)(stext + ao), ahi = *(const u64*)(stext + ao + 8);
                    const u64 a = as ? ((alo >> as) | (ahi << (64u - as))) : alo;
                    const uintptr_t pa = (uintptr_t)(text + j + l);
                    const uintptr_t sh = pa - w_addr;
                    u64 b;
                    u32 valid = 8;
                    if (sh < 8) { b = w_val >> (8 * sh); valid = 8u - (u32)sh; }       // the tail of the word already held
                    else { __builtin_memcpy(&w_val, (const void*)pa, 8); w_addr = pa; b = w_val; }
                    u64 x = a ^ b;
                    if (valid < 8) x &= (1ull << (8 * valid)) - 1ull;
                    if (x) { l += (u32)__builtin_ctzll(x) >> 3; break; }
                    l += valid;
                } else {                                    // the last bytes of the text, or beyond the staged halo
                    const u8 a1 = (off < (u32)(PLCP_TILE + PLCP_HALO)) ? stext[off] : text[i + l];
                    if (a1 != text[j + l]) break;
                    ++l;
                }
            }
            row[c] = l;
            mx = max(mx, l);
            if (l) --l;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < PLCP_TILE; k += 256) {
        const size_t p = base + k;
        if (p < n) plcp[p] = sphi[(k / PLCP_CHUNK) * (PLCP_CHUNK + 1) + (k % PLCP_CHUNK)];
    }
    mx = wave_reduce_max(mx);
    if (lane_id() == 0 && mx) atomicMax(d_max, mx);
}

// exact values at the positions 1024 k, coarse to fine; the intervals that hide a jump are computed completely (into `out`)
// (nullptr for short texts or when switched off)
template <bool ALLOW_NONE>
static u32* plcp_samples(Ctx& c, const u8* text, size_t n, const u32* src, u32* out, u32* d_max, u8** iflag_out) {
    *iflag_out = nullptr;
    if (!c.plcp_samples || n < ((size_t)1 << 16)) return nullptr;
    const u32 nsamp = (u32)cdiv(n, PLCP_SAMPLE);
    u32* samples = c.arena.get<u32>(nsamp);
    u32 parent = 0;
    for (u32 step = 1u << 16; step >= 1; step >>= 4) {            // spacing 2^26 ... 2^10 text positions
        if (step < nsamp || step == 1) {
            const size_t waves = cdiv(nsamp, step);
            plcp_sample_kernel<ALLOW_NONE><<<cdiv(waves * 64, 256), 256, 0, c.stream>>>(text, n, src, samples, nsamp, step, parent);
            LAUNCH_CHECK();
            parent = step;
        }
        if (step == 1) break;
    }
    u8* iflag = c.arena.get<u8>(nsamp);
    plcp_refine_kernel<ALLOW_NONE><<<cdiv((size_t)nsamp * 64, 256), 256, 0, c.stream>>>(text, n, src, samples, nsamp, out, iflag, d_max);
    LAUNCH_CHECK();
    *iflag_out = iflag;
    return samples;
}

void build_plcp(Ctx& c, const u8* text, size_t n, const u32* phi, u32* plcp, u32* d_maxlcp) {
    HIP_TRY(hipMemsetAsync(d_maxlcp, 0, sizeof(u32), c.stream));
    if (!n) return;
    Ctx::ProfScope prof(c, K_PLCP, (u64)n * 10);                // Phi (4) + two text bytes + PLCP (4), SURVEY 8d
    const size_t mark = c.arena.mark();
    u8* iflag = nullptr;
    u32* samples = plcp_samples<false>(c, text, n, phi, plcp, d_maxlcp, &iflag);
    plcp_kernel<false><<<cdiv(n, PLCP_TILE), 256, 0, c.stream>>>(text, n, phi, plcp, d_maxlcp, samples, iflag);
    LAUNCH_CHECK();
    c.arena.release(mark);
}

void build_lce_with_carry(Ctx& c, const u8* text, size_t n, const u32* src, u32* len, u32* d_max) {
    HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(u32), c.stream));
    if (!n) return;
    Ctx::ProfScope prof(c, K_PLCP, (u64)n * 10);
    const size_t mark = c.arena.mark();
    u8* iflag = nullptr;
    u32* samples = plcp_samples<true>(c, text, n, src, len, d_max, &iflag);
    plcp_kernel<true><<<cdiv(n, PLCP_TILE), 256, 0, c.stream>>>(text, n, src, len, d_max, samples, iflag);
    LAUNCH_CHECK();
    c.arena.release(mark);
}

__global__ void lcp_kernel(const u32* __restrict__ sa, const u32* __restrict__ plcp, size_t n, u32* __restrict__ lcp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lcp[i] = (i == 0) ? 0u : plcp[sa[i]];
}

void build_lcp(Ctx& c, const u32* sa, const u32* plcp, size_t n, u32* lcp) {
    if (!n) return;
    lcp_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(sa, plcp, n, lcp);
    LAUNCH_CHECK();
}

}  // namespace tdc
