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

__global__ __launch_bounds__(256) void byte_hist_kernel(const u8* __restrict__ text, size_t n, u32* __restrict__ hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&h[text[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
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

// head marker: index of the element if it starts a new group, else 0 (a max-scan then yields the group head)
__global__ void sa_heads_kernel(const u64* __restrict__ keys, size_t m, u32* __restrict__ head) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    head[a] = (a == 0 || keys[a] != keys[a - 1]) ? (u32)a : 0u;
}

// First round: every position is "active", pos[a] == a.
template <bool SCATTER_RANK>
__global__ void sa_first_update_kernel(const u32* __restrict__ vals, const u32* __restrict__ head, size_t n,
                                       u32* __restrict__ sa, u32* __restrict__ rank, u32* __restrict__ keep) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 h = head[j];
    const u32 s = vals[j];
    sa[j] = s;
    if (SCATTER_RANK) rank[s] = h;
    const bool single = (h == (u32)j) && (j + 1 == n || head[j + 1] == (u32)(j + 1));
    keep[j] = single ? 0u : 1u;
}
__global__ void sa_first_compact_kernel(const u32* __restrict__ vals, const u32* __restrict__ head, const u32* __restrict__ offs,
                                        size_t n, u32* __restrict__ a_sa, u32* __restrict__ a_pos, u32* __restrict__ a_r1) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 h = head[j];
    const bool single = (h == (u32)j) && (j + 1 == n || head[j + 1] == (u32)(j + 1));
    if (!single) {
        const u32 o = offs[j];
        a_sa[o] = vals[j];
        a_pos[o] = (u32)j;
        a_r1[o] = h;
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

// newrank_out != nullptr: the new ranks are only written out (in list order); the caller scatters them to rank[] through the
// bucketed scatter (writing an unchanged rank again is harmless)
__global__ void sa_update_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals, const u32* __restrict__ head,
                                 const u32* __restrict__ a_pos, size_t m, int bn, u32* __restrict__ sa, u32* __restrict__ rank,
                                 u32* __restrict__ keep, u32* __restrict__ newrank_out) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    const u32 h = head[a];
    const u32 s = vals[a];
    const u32 newrank = a_pos[h];
    sa[a_pos[a]] = s;
    if (newrank_out) newrank_out[a] = newrank;
    else if (newrank != (u32)(keys[a] >> bn)) rank[s] = newrank;     // rank only moves when the group was split
    const bool single = (h == (u32)a) && (a + 1 == m || head[a + 1] == (u32)(a + 1));
    keep[a] = single ? 0u : 1u;
}
__global__ void sa_compact_kernel(const u32* __restrict__ vals, const u32* __restrict__ head, const u32* __restrict__ offs,
                                  const u32* __restrict__ a_pos, size_t m, u32* __restrict__ b_sa, u32* __restrict__ b_pos,
                                  u32* __restrict__ b_r1) {
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    const u32 h = head[a];
    const bool single = (h == (u32)a) && (a + 1 == m || head[a + 1] == (u32)(a + 1));
    if (!single) {
        const u32 o = offs[a];
        b_sa[o] = vals[a];
        b_pos[o] = a_pos[a];
        b_r1[o] = a_pos[h];
    }
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

void build_suffix_array(Ctx& c, const u8* text, size_t n, u32* sa, u32* isa, SAStats* st) {
    SAStats local;
    if (!st) st = &local;
    *st = SAStats();
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();

    // --- dense symbol codes -----------------------------------------------------------------------
    u32* d_hist = c.arena.get<u32>(256);
    HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), s));
    {
        unsigned g = cdiv(n, 256 * 16); if (g > 2048) g = 2048; if (g == 0) g = 1;
        byte_hist_kernel<<<g, 256, 0, s>>>(text, n, d_hist);
        LAUNCH_CHECK();
    }
    u32 h_hist[256];
    c.read_n(d_hist, h_hist, 256);
    CodeMap cm;
    u32 sigma = 0;
    for (int i = 0; i < 256; ++i) { cm.code[i] = (u8)sigma; if (h_hist[i]) ++sigma; }
    // a byte that does not occur keeps the code of the next present byte; irrelevant (never looked up)
    const int b = (int)bits_for(sigma > 1 ? sigma - 1 : 1);
    const u32 base = sigma > 2 ? sigma : 2;
    int k = 0;                                               // largest k with base^k <= 2^64, at most 32 (LDS halo of the key kernel)
    int key_bits = 64;
    {
        unsigned __int128 pw = 1;
        while (k < 32 && pw * base <= ((unsigned __int128)1 << 64)) { pw *= base; ++k; }
        if (const char* e = getenv("TDC_GPU_SA_INIT_SYMS")) { const int v = atoi(e); if (v >= 1 && v < k) { k = v; pw = 1; for (int i = 0; i < k; ++i) pw *= base; } }   // tuning knob
        key_bits = (pw > ((unsigned __int128)1 << 63)) ? 64 : (int)bits_for((u64)(pw - 1));
    }
    st->sym_bits = b; st->init_syms = k;

    // --- buffers ----------------------------------------------------------------------------------
    u64* keys[2] = { c.arena.get<u64>(n), c.arena.get<u64>(n) };
    u32* vals[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    u32* head = c.arena.get<u32>(n);
    u32* keep = c.arena.get<u32>(n);
    u32* A_sa = c.arena.get<u32>(n), *A_pos = c.arena.get<u32>(n), *A_r1 = c.arena.get<u32>(n);
    u32* B_sa = c.arena.get<u32>(n), *B_pos = c.arena.get<u32>(n), *B_r1 = c.arena.get<u32>(n);
    u32* d_total = c.arena.get<u32>(1);
    u64* lkeys = c.arena.get<u64>(n / 2 + 2048);       // second buffers of the "open run" sort (at most half of the list...)
    u32* lvals = c.arena.get<u32>(n / 2 + 2048);
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
    const unsigned gn = cdiv(n, 256);
    sa_heads_kernel<<<gn, 256, 0, s>>>(keys[x], n, head);
    LAUNCH_CHECK();
    inclusive_max_u32(c, head, head, n);
    {   // per element: read value + head (8 B), write sa + keep (8 B), scatter rank (4 B)
        Ctx::ProfScope prof(c, K_SA_RANK_SCATTER, (u64)n * 20);
        if (c.bucket_scatter && n >= ((size_t)1 << 22)) {
            // rank[vals[j]] = head[j] through a partition by destination window; the other key buffer is free scratch
            sa_first_update_kernel<false><<<gn, 256, 0, s>>>(vals[x], head, n, sa, rank, keep);
            LAUNCH_CHECK();
            bucketed_scatter_u32(c, vals[x], head, n, rank, n, (u32*)keys[x ^ 1], vals[x ^ 1], B_sa, B_pos, true);   // B_* are free until the first compaction; vals[x] = every position once
        } else {
            sa_first_update_kernel<true><<<gn, 256, 0, s>>>(vals[x], head, n, sa, rank, keep);
            LAUNCH_CHECK();
        }
    }
    exclusive_sum_u32(c, keep, keep, n, d_total);
    sa_first_compact_kernel<<<gn, 256, 0, s>>>(vals[x], head, keep, n, A_sa, A_pos, A_r1);
    LAUNCH_CHECK();
    size_t m = c.read(d_total);
    st->rounds = 1;

    // --- doubling rounds --------------------------------------------------------------------------
    const int bn = (int)bits_for(n - 1);
    u64 h = (u64)k;
    while (m > 0) {
        if (h >= n) throw HipError{hipErrorUnknown, "suffix_array: doubling did not converge", (int)__LINE__};
        const unsigned gm = cdiv(m, 256);
        {   // per element: read sa + r1 (8 B), gather rank[sa+h] (4 B), write key + value (12 B)
            Ctx::ProfScope prof(c, K_SA_BUILD_KEYS, (u64)m * 24);
            sa_build_keys_kernel<<<gm, 256, 0, s>>>(A_sa, A_r1, m, n, (u32)h, bn, rank, keys[0], vals[0]);
            LAUNCH_CHECK();
        }
        if (c.sa_local_sort) {
            // local part: whole runs inside 2048-element tiles; global part: only the runs that cross a tile border
            u8* cls = (u8*)keep;                                   // scratch (keep is rewritten by sa_update_kernel)
            {
                Ctx::ProfScope prof(c, K_SA_LOCAL_SORT, (u64)m * 25);
                sa_local_sort_kernel<<<cdiv(m, 2048), 256, 0, s>>>(keys[0], vals[0], m, bn, cls);
                LAUNCH_CHECK();
            }
            u32* opos = B_sa;                                      // B_* are free until the compaction of this round
            select_by_class(c, cls, 1, m, nullptr, opos, nullptr, nullptr, d_total);
            const size_t mo = c.read(d_total);
            if (mo > n / 2) {                                      // a few giant runs: sort everything globally
                x = (c.ssort && splitter_sort_applicable(m)) ? splitter_sort_pairs_u64(c, keys, vals, m, nullptr, nullptr)
                                                              : radix_sort_pairs_u64(c, keys, vals, m, 0, 2 * bn);
                st->sorted_elems += m;
            } else {
            st->sorted_elems += mo;
            if (mo) {
                u64* ok2[2] = { keys[1], lkeys };
                u32* ov2[2] = { vals[1], lvals };
                select_by_class(c, cls, 1, m, vals[0], ov2[0], keys[0], ok2[0], d_total);
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
        sa_heads_kernel<<<gm, 256, 0, s>>>(keys[x], m, head);
        LAUNCH_CHECK();
        inclusive_max_u32(c, head, head, m);
        {
            Ctx::ProfScope prof(c, K_SA_RANK_SCATTER, (u64)m * 36);
            // a large round scatters its ranks through the bucketed scatter (scratch: the other sort buffers, the B lists)
            const bool bucketed = c.bucket_scatter && m >= ((size_t)1 << 24);
            u32* nr = (u32*)keys[x ^ 1];
            sa_update_kernel<<<gm, 256, 0, s>>>(keys[x], vals[x], head, A_pos, m, bn, sa, rank, keep, bucketed ? nr : nullptr);
            LAUNCH_CHECK();
            if (bucketed) bucketed_scatter_u32(c, vals[x], nr, m, rank, n, nr + m, vals[x ^ 1], B_sa, B_pos);
        }
        exclusive_sum_u32(c, keep, keep, m, d_total);
        sa_compact_kernel<<<gm, 256, 0, s>>>(vals[x], head, keep, A_pos, m, B_sa, B_pos, B_r1);
        LAUNCH_CHECK();
        m = c.read(d_total);
        u32* t;
        t = A_sa; A_sa = B_sa; B_sa = t;
        t = A_pos; A_pos = B_pos; B_pos = t;
        t = A_r1; A_r1 = B_r1; B_r1 = t;
        h *= 2;
        st->rounds++;
    }
    c.arena.release(mark);
}

}  // namespace tdc
