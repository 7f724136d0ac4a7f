// factorize.hip -- lcpcomp::ArraysComp (compressors/lcpcomp/compress/ArraysComp.hpp:36-117) in position space.
//
// The reference walks LCP levels L = maxlcp .. threshold; inside a level it scans a candidate list in order
// (originals by ascending SA index, then lazily pushed-down entries in encounter order), emits a factor for every
// entry whose LCP value still equals L, zeroes the LCP of the L covered text positions (:99-101) and truncates the
// LCP of the up to L positions in front of it (:103-109).
//
// Device formulation (validated against the oracle by tests/models/position_space.py):
//   * all state is indexed by TEXT POSITION p: cur[p] = lcp[isa[p]] (initially PLCP[p]), source = Phi[p];
//   * resid[p] = level whose list currently holds p's entry; the pushed part of every list is kept in one
//     time-ordered pool (pool index = encounter order), so list L = originals(L) ++ {pool entries with target L};
//   * inside a level only "live" entries (cur == L) can be selected, and an entry is selected iff no EARLIER list
//     entry within text distance < L is selected: the lexicographically-first maximal independent set in list
//     order, computed by rounds in which an entry decides once all its earlier live neighbours have decided;
//   * every non-selected entry then gets its encounter value v = cur reduced by the selected earlier neighbours
//     (left neighbour covers it -> 0, right neighbour at distance d -> min(v, d)) and is appended to list v
//     (or dropped if v < threshold);
//   * kills (cur = 0, owner = factor start) and truncations (atomicMin) of all selected entries are applied last;
//     they commute, so their order does not matter.
#include "stages.hpp"
#include "prim.hpp"

namespace tdc {

struct LevelScalars {
    u32 live;        // != 0 iff some entry has cur == L
    u32 alive;       // != 0 iff some entry has cur >= threshold
    u32 undecided;   // live entries still undecided after the last round
    u32 selected;    // factors emitted in this level
    u32 npush;       // entries pushed down from this level
    u32 pad[3];
};

// ---- candidates ------------------------------------------------------------------------------------------
__global__ void cand_flag_kernel(const u32* __restrict__ sa, const u32* __restrict__ plcp, size_t n, u32 threshold,
                                 u32* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i >= 1 && plcp[sa[i]] >= threshold) ? 1u : 0u;
}
__global__ void cand_scatter_kernel(const u32* __restrict__ sa, const u32* __restrict__ plcp, const u32* __restrict__ offs,
                                    size_t n, u32 threshold, u32* __restrict__ keys, u32* __restrict__ vals) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i == 0) return;
    const u32 p = sa[i];
    const u32 v = plcp[p];
    if (v >= threshold) { const u32 o = offs[i]; keys[o] = v; vals[o] = p; }
}
__global__ void seg_bounds_kernel(const u32* __restrict__ keys, size_t m, u32* __restrict__ segstart, u32* __restrict__ segend) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const u32 k = keys[j];
    if (j == 0 || keys[j - 1] != k) segstart[k] = (u32)j;
    if (j + 1 == m || keys[j + 1] != k) segend[k] = (u32)(j + 1);
}
__global__ void resid_init_kernel(const u32* __restrict__ plcp, size_t n, u32 threshold, u32* __restrict__ resid,
                                  u32* __restrict__ flen, u32* __restrict__ owner) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 v = plcp[p];
    resid[p] = (v >= threshold) ? v : 0u;
    flen[p] = 0;
    owner[p] = NONE32;
}

// ---- per level ---------------------------------------------------------------------------------------------
__global__ void pool_flag_kernel(const u32* __restrict__ pool_t, size_t top, u32 L, u32* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= top) return;
    flag[i] = (pool_t[i] == L) ? 1u : 0u;
}
__global__ void pool_gather_kernel(const u32* __restrict__ pool_p, const u32* __restrict__ pool_t, const u32* __restrict__ offs,
                                   size_t top, u32 L, u32* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= top) return;
    if (pool_t[i] == L) dst[offs[i]] = pool_p[i];
}

__global__ void level_init_kernel(const u32* __restrict__ orig, u32 m0, const u32* __restrict__ pushed, u32 m, u32 L,
                                  u32 threshold, const u32* __restrict__ cur, u32* __restrict__ list, u32* __restrict__ lidx,
                                  u32* __restrict__ vcur, u32* __restrict__ state, LevelScalars* __restrict__ sc) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false, alive = false;
    if (k < m) {
        const u32 p = (k < m0) ? orig[k] : pushed[k - m0];
        const u32 v = cur[p];
        list[k] = p;
        lidx[p] = k;
        vcur[k] = v;
        live = (v == L);
        alive = (v >= threshold);
        state[k] = live ? 0u : 2u;
    }
    // only "any live / any alive" is needed: plain flag stores (all writers store the same value) instead of
    // millions of same-address atomics
    const u64 bl = __ballot(live), ba = __ballot(alive);
    if (lane_id() == 0) {
        if (bl) sc->live = 1u;
        if (ba) sc->alive = 1u;
    }
}

// One MIS round.  G lanes cooperate on one entry (G = 1 for short levels, 64 for long ones).
template <int G>
__global__ __launch_bounds__(256) void mis_round_kernel(const u32* __restrict__ list, u32 m, u32 L, size_t n,
                                                         const u32* __restrict__ resid, const u32* __restrict__ lidx,
                                                         u32* state, LevelScalars* __restrict__ sc) {
    const u32 gid = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    const u32 k = gid;
    bool active = (k < m);
    if (active) active = (state[k] == 0u);
    bool hit = false, blocked = false;
    if (G == 1) { if (!active) return; }
    else { if (!__any(active)) return; }
    if (active) {
        const u32 p = list[k];
        const size_t lo = (p >= L - 1) ? (size_t)p - (L - 1) : 0;
        size_t hi = (size_t)p + (L - 1);
        if (hi > n - 1) hi = n - 1;
        for (size_t q = lo + sub; q <= hi; q += G) {
            if (q == p) continue;
            if (resid[q] != L) continue;
            const u32 kq = lidx[q];
            if (kq >= k) continue;
            const u32 stq = state[kq];
            if (stq == 1u) { hit = true; break; }
            if (stq == 0u) blocked = true;
        }
    }
    if (G > 1) {   // all 64 lanes of the wave work on the same entry
        hit = __any(hit);
        blocked = __any(blocked);
        if (sub != 0) return;
    }
    if (!active) return;
    if (hit) state[k] = 2u;
    else if (!blocked) state[k] = 1u;
    else atomicAdd(&sc->undecided, 1u);
}

// Encounter value of every non-selected entry; pushtgt[k] = new list (0 = dropped / selected).
template <int G>
__global__ __launch_bounds__(256) void resolve_kernel(const u32* __restrict__ list, u32 m, u32 L, u32 threshold, size_t n,
                                                       const u32* __restrict__ resid, const u32* __restrict__ lidx,
                                                       const u32* __restrict__ state, const u32* __restrict__ vcur,
                                                       u32* __restrict__ pushtgt, u32* __restrict__ pushbin,
                                                       LevelScalars* __restrict__ sc) {
    const u32 k = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    if (k >= m) return;                       // G divides the wave: whole groups leave together
    const u32 st = state[k];
    u32 v = (st == 1u) ? 0u : vcur[k];
    if (st != 1u && v >= threshold) {
        const u32 p = list[k];
        const size_t lo = (p >= L - 1) ? (size_t)p - (L - 1) : 0;
        size_t hi = (size_t)p + (L - 1);
        if (hi > n - 1) hi = n - 1;
        for (size_t q = lo + sub; q <= hi; q += G) {
            if (q == p) continue;
            if (resid[q] != L) continue;
            const u32 kq = lidx[q];
            if (kq >= k) continue;
            if (state[kq] != 1u) continue;
            if (q < p) { v = 0; break; }       // covered by a factor starting to the left  (:99-101)
            const u32 d = (u32)(q - p);         // truncated by a factor starting to the right (:103-109)
            if (d < v) v = d;
        }
    }
    if (G > 1) {
        v = wave_reduce_min(v);
        if (sub != 0) return;
    }
    const bool push = (st != 1u) && (v >= threshold);
    pushtgt[k] = push ? v : 0u;
    pushbin[k] = push ? 1u : 0u;
    if (G > 1) { if (st == 1u) atomicAdd(&sc->selected, 1u); }
    else {
        const u64 bs = __ballot(st == 1u);
        if (bs && (u64)lane_id() == (u64)__builtin_ctzll(bs)) atomicAdd(&sc->selected, (u32)__popcll(bs));
    }
}

// pushcnt[v] counts the pool entries per target level.  Targets cluster on a few small levels, so the counts are
// first accumulated in an LDS histogram (targets < 1024) and flushed with one global atomic per non-empty bin.
__global__ __launch_bounds__(256) void push_kernel(const u32* __restrict__ list, u32 m, const u32* __restrict__ pushtgt,
                                                    const u32* __restrict__ poffs, u32 pool_top, u32* __restrict__ pool_p,
                                                    u32* __restrict__ pool_t, u32* __restrict__ resid, u32* __restrict__ pushcnt) {
    __shared__ u32 h[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) h[i] = 0;
    __syncthreads();
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) {
        const u32 v = pushtgt[k];
        if (v) {
            const u32 p = list[k];
            const u32 idx = pool_top + poffs[k];
            pool_p[idx] = p;
            pool_t[idx] = v;
            resid[p] = v;
            if (v < 1024) atomicAdd(&h[v], 1u); else atomicAdd(&pushcnt[v], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) if (h[i]) atomicAdd(&pushcnt[i], h[i]);
}

// Emit the selected entries: factor (p, Phi[p], L); kill the covered positions, truncate the ones in front.
template <int G>
__global__ __launch_bounds__(256) void apply_kernel(const u32* __restrict__ list, u32 m, u32 L, size_t n, const u32* __restrict__ state,
                                                     const u32* __restrict__ phi, u32* __restrict__ cur, u32* __restrict__ flen,
                                                     u32* __restrict__ owner, u32* __restrict__ fsrc) {
    const u32 k = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    if (k >= m) return;
    if (state[k] != 1u) return;
    const u32 p = list[k];
    if (sub == 0) { flen[p] = L; fsrc[p] = phi[p]; }
    for (u32 j = sub; j < L && (size_t)p + j < n; j += G) {   // :99-101
        cur[p + j] = 0;
        owner[p + j] = p;
    }
    const u32 aff = (L < p) ? L : p;                     // :103
    for (u32 j = sub; j < aff; j += G) atomicMin(&cur[p - 1 - j], j + 1);   // :105-109
}

void factorize_arrays(Ctx& c, size_t n, const u32* sa, const u32* isa, const u32* phi, u32* plcp, u32 maxlcp, u32 threshold,
                      FactorSpace fs, FactorizeStats* st) {
    (void)isa;   // priorities are implicit in the candidate order (ascending SA index), the ISA itself is not needed
    FactorizeStats local;
    if (!st) st = &local;
    *st = FactorizeStats();
    st->maxlcp = maxlcp;
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    const unsigned gn = cdiv(n, 256);
    u32* cur = plcp;

    u32* resid = c.arena.get<u32>(n);
    resid_init_kernel<<<gn, 256, 0, s>>>(plcp, n, threshold, resid, fs.flen, fs.owner);
    LAUNCH_CHECK();
    if ((u64)maxlcp + 1 <= threshold || threshold == 0) { c.arena.release(mark); return; }   // ArraysComp.hpp:50

    // ---- "Fill candidates" (:54-66): positions with LCP >= threshold in SA order, stably sorted by LCP value
    u32* tmpA = c.arena.get<u32>(n);      // flags / offsets, later per-level temporaries
    u32* tmpB = c.arena.get<u32>(n);
    u32* ckeys[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    u32* cvals[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    u32* d_total = c.arena.get<u32>(1);
    {
        Ctx::ProfScope prof(c, K_CAND, (u64)n * 12);
        cand_flag_kernel<<<gn, 256, 0, s>>>(sa, plcp, n, threshold, tmpA);
        LAUNCH_CHECK();
    }
    exclusive_sum_u32(c, tmpA, tmpA, n, d_total);
    {
        Ctx::ProfScope prof(c, K_CAND, (u64)n * 20);
        cand_scatter_kernel<<<gn, 256, 0, s>>>(sa, plcp, tmpA, n, threshold, ckeys[0], cvals[0]);
        LAUNCH_CHECK();
    }
    const size_t entries = c.read(d_total);
    st->entries = entries;
    const int x = radix_sort_pairs_u32(c, ckeys, cvals, entries, 0, (int)bits_for(maxlcp));
    const u32* cand = cvals[x];
    const size_t nlev = (size_t)maxlcp + 2;
    u32* d_segstart = c.arena.get<u32>(nlev);
    u32* d_segend = c.arena.get<u32>(nlev);
    u32* pushcnt = c.arena.get<u32>(nlev);
    HIP_TRY(hipMemsetAsync(d_segstart, 0, nlev * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(d_segend, 0, nlev * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(pushcnt, 0, nlev * sizeof(u32), s));
    if (entries) {
        seg_bounds_kernel<<<cdiv(entries, 256), 256, 0, s>>>(ckeys[x], entries, d_segstart, d_segend);
        LAUNCH_CHECK();
    }
    u32* h_segstart = (u32*)malloc(nlev * sizeof(u32));
    u32* h_segend = (u32*)malloc(nlev * sizeof(u32));
    if (!h_segstart || !h_segend) { free(h_segstart); free(h_segend); throw HipError{hipErrorOutOfMemory, "host", (int)__LINE__}; }
    try {
        c.read_n(d_segstart, h_segstart, nlev);
        c.read_n(d_segend, h_segend, nlev);

        // ---- per-level state -------------------------------------------------------------------------
        u32* lidx = c.arena.get<u32>(n);
        u32* pool_p = c.arena.get<u32>(n);
        u32* pool_t = c.arena.get<u32>(n);
        u32* list = ckeys[x ^ 1];           // the sort's scratch buffers are free now
        u32* vcur = cvals[x ^ 1];
        u32* state = ckeys[x];              // keys of the sorted candidates are no longer needed either
        u32* pushed = c.arena.get<u32>(n);
        u32* pushtgt = tmpA;
        u32* pushbin = tmpB;
        LevelScalars* d_sc = (LevelScalars*)c.arena.alloc(sizeof(LevelScalars));
        LevelScalars h_sc;
        size_t pool_top = 0;

        for (u32 L = maxlcp; L >= threshold; --L) {
            const u32 m0 = h_segend[L] - h_segstart[L];
            u32 m1 = 0;
            if (pool_top > 0) m1 = c.read(&pushcnt[L]);
            const u32 m = m0 + m1;
            if (m == 0) continue;
            st->levels++;
            if (m1) {     // pushed part of the list: pool entries with target L, in pool (= encounter) order
                const unsigned gp = cdiv(pool_top, 256);
                Ctx::ProfScope prof(c, K_POOL, (u64)pool_top * 16 + (u64)m1 * 8);
                pool_flag_kernel<<<gp, 256, 0, s>>>(pool_t, pool_top, L, pushbin);
                LAUNCH_CHECK();
                exclusive_sum_u32(c, pushbin, pushbin, pool_top, nullptr);
                pool_gather_kernel<<<gp, 256, 0, s>>>(pool_p, pool_t, pushbin, pool_top, L, pushed);
                LAUNCH_CHECK();
            }
            HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(LevelScalars), s));
            const unsigned gm = cdiv(m, 256);
            {   // per entry: list source (4) + cur gather (4) + list/vcur/state (12) + lidx scatter (4)
                Ctx::ProfScope prof(c, K_LEVEL_INIT, (u64)m * 24);
                level_init_kernel<<<gm, 256, 0, s>>>(cand + h_segstart[L], m0, pushed, m, L, threshold, cur, list, lidx, vcur, state, d_sc);
                LAUNCH_CHECK();
            }
            h_sc = c.read(d_sc);
            if (h_sc.alive == 0) continue;          // every entry already erased (:86)
            const bool wide = (L > 24);
            const unsigned gw = wide ? cdiv((size_t)m * 64, 256) : gm;
            if (h_sc.live) {
                u32 undecided = m;
                while (undecided) {
                    HIP_TRY(hipMemsetAsync(&d_sc->undecided, 0, sizeof(u32), s));
                    {   // per undecided entry: list + state (8) + a window of 2L-1 resid words
                        Ctx::ProfScope prof(c, K_MIS_ROUND, (u64)m * 4 + (u64)undecided * (4 + 4ull * (2 * L - 1)));
                        if (wide) mis_round_kernel<64><<<gw, 256, 0, s>>>(list, m, L, n, resid, lidx, state, d_sc);
                        else      mis_round_kernel<1><<<gw, 256, 0, s>>>(list, m, L, n, resid, lidx, state, d_sc);
                        LAUNCH_CHECK();
                    }
                    const u32 now = c.read(&d_sc->undecided);
                    st->rounds++;
                    if (now >= undecided && now != 0) {
                        // the earliest undecided entry can always decide: no progress means a bug
                        throw HipError{hipErrorUnknown, "factorize: MIS rounds made no progress", (int)__LINE__};
                    }
                    undecided = now;
                }
            }
            {   // per entry: state + vcur + list (12), outputs (8), and for live-or-stale entries the resid window
                Ctx::ProfScope prof(c, K_RESOLVE, (u64)m * (20 + 4ull * (2 * L - 1)));
                if (wide) resolve_kernel<64><<<gw, 256, 0, s>>>(list, m, L, threshold, n, resid, lidx, state, vcur, pushtgt, pushbin, d_sc);
                else      resolve_kernel<1><<<gw, 256, 0, s>>>(list, m, L, threshold, n, resid, lidx, state, vcur, pushtgt, pushbin, d_sc);
                LAUNCH_CHECK();
            }
            exclusive_sum_u32(c, pushbin, pushbin, m, &d_sc->npush);
            h_sc = c.read(d_sc);
            // every push is caused by a truncation of a position in front of a factor, and factors are disjoint,
            // so the pool never needs more than n slots; checked before anything is written
            if (pool_top + h_sc.npush > n) throw HipError{hipErrorUnknown, "factorize: push pool overflow", (int)__LINE__};
            if (h_sc.npush) {
                Ctx::ProfScope prof(c, K_PUSH, (u64)m * 8 + (u64)h_sc.npush * 16);
                push_kernel<<<gm, 256, 0, s>>>(list, m, pushtgt, pushbin, (u32)pool_top, pool_p, pool_t, resid, pushcnt);
                LAUNCH_CHECK();
            }
            if (h_sc.selected) {
                // per entry: state + list (8); per factor: Phi, flen, fsrc (12) + L kills (8 B each) + L truncations (4 B each)
                Ctx::ProfScope prof(c, K_APPLY, (u64)m * 8 + (u64)h_sc.selected * (12 + 12ull * L));
                if (wide) apply_kernel<64><<<gw, 256, 0, s>>>(list, m, L, n, state, phi, cur, fs.flen, fs.owner, fs.fsrc);
                else      apply_kernel<1><<<gw, 256, 0, s>>>(list, m, L, n, state, phi, cur, fs.flen, fs.owner, fs.fsrc);
                LAUNCH_CHECK();
            }
            pool_top += h_sc.npush;
            st->factors += h_sc.selected;
            st->pushes += h_sc.npush;
            if (L == 0) break;
        }
    } catch (...) {
        free(h_segstart); free(h_segend);
        throw;
    }
    free(h_segstart); free(h_segend);
    c.arena.release(mark);
}

}  // namespace tdc
