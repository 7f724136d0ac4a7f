// factorize.hip -- lcpcomp::ArraysComp (compressors/lcpcomp/compress/ArraysComp.hpp:36-117) in position space.
//
// The reference walks LCP levels L = maxlcp .. threshold; inside a level it scans a candidate list in order
// (originals by ascending SA index, then lazily pushed-down entries in encounter order), emits a factor for every
// entry whose LCP value still equals L, zeroes the LCP of the L covered text positions (:99-101) and truncates the
// LCP of the up to L positions in front of it (:103-109).
//
// Device formulation (models: tests/models/position_space.py, checked against the oracle incl. emission order):
//   * all state is indexed by TEXT POSITION p: cur[p] = lcp[isa[p]] (initially PLCP[p]), source = Phi[p];
//   * every entry lives in exactly one list (its original level's segment of the sorted candidates, or one segment
//     of the push pool).  Lists are UNORDERED sets kept in position order (so all per-level kernels walk the
//     position-indexed arrays monotonically); the reference's list order lives in an
//     explicit priority prio[p]: ISA[p] for original candidates (ascending SA index), and base + rank for entries
//     pushed down from level L, where rank is the index after sorting that level's pushes by (target, old priority)
//     and base grows monotonically -- pushed entries follow the originals, later pushes follow earlier ones, pushes of
//     one level keep their order: exactly the reference's append order;
//   * inside a level only "live" entries (cur == L) can be selected, and one is selected iff no live entry of HIGHER
//     priority within text distance < L is selected: the lexicographically-first maximal independent set, computed by
//     rounds in which an entry decides once all its higher-priority live neighbours have decided;
//   * every other entry (stale or rejected) gets its encounter value v = cur reduced by the selected neighbours of
//     higher priority (left neighbour covers it -> 0, right neighbour at distance d -> min(v, d)) and moves to list v
//     (or is dropped if v < threshold);
//   * kills (cur = 0, owner = factor start) and truncations (atomicMin) of all selected entries are applied last;
//     they commute, so their order does not matter;
//   * the per-level state of a position is 2 bits in a bitmap (01 undecided live entry, 10 selected), so the
//     "neighbours within distance < L" scans read one or two 64-bit words instead of 2L-1 array elements, and a
//     decision is ONE atomic (xor 11: undecided -> selected, xor 01: undecided -> rejected).
#include "stages.hpp"
#include "prim.hpp"
#include "factorize_tiles.hpp"
#include "factorize_eager.hpp"

#include <algorithm>
#include <map>
#include <unordered_map>
#include <chrono>
#include <stdlib.h>
#include <vector>

namespace tdc {

enum : u8 { CL_DEAD = 0, CL_LIVE = 1, CL_STALE = 2 };

struct PushSeg { u32 target, start; };
struct GatherSeg { u32 src_off, dst_off; };   // pushed part of a level's list = concatenation of pool segments
constexpr u32 SEG_INLINE = 508;      // segment descriptors that travel with the scalars in one read-back

// Two one-workgroup levels may be in flight: the kernel of level X is queued behind the kernel of the level Y above it before the host
// has seen Y's result ("speculative": its list was put together without Y's pushes).  It checks Y's published result on the device
// -- Y gave up, left its factors to a chip-wide launch, or pushed something into X: then X gives up as well, before it has changed
// anything -- and takes its pool offset and priority base from this block, which every level updates.
struct SmallCtl { u32 pool_top, prio_base; };

struct LevelScalars {
    u32 nlive, nstale;   // entries with cur == L / threshold <= cur < L
    u32 undecided;       // live entries still undecided after the last round
    u32 selected;        // factors emitted in this level
    u32 npush;           // entries pushed down from this level
    u32 nseg;            // distinct push targets of this level
    u32 pad[2];
    PushSeg segs[SEG_INLINE];
};

// ---- per-level state bitmap: 2 bits per text position, 32 positions per 64-bit word -----------------------------
constexpr u64 BM_UNDECIDED_ALL = 0x5555555555555555ull;   // bit 0 of every pair
constexpr u64 BM_SELECTED_ALL  = 0xAAAAAAAAAAAAAAAAull;   // bit 1 of every pair

__device__ __forceinline__ u64 bm_range_mask(size_t word, size_t lo, size_t hi) {   // pairs of positions lo..hi inside `word`
    const size_t w0 = word * 32;
    const size_t first = lo > w0 ? lo - w0 : 0;
    const size_t last = (hi < w0 + 31) ? hi - w0 : 31;
    const u64 upto = (last == 31) ? ~0ull : ((1ull << (2 * (last + 1))) - 1);
    return upto & ~((1ull << (2 * first)) - 1);
}
__device__ __forceinline__ u64 bm_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 bm_state(const u64* __restrict__ bm, u32 p) { return (u32)(bm[p >> 5] >> (2 * (p & 31))) & 3u; }

// Residence byte res8[] (present whenever a window pass or an eager phase may follow): min(PLCP, 255) while the entry is NATURAL (its
// working value is still its PLCP value), a MARK from its first effective cut on -- 0x80 | t for values t <= 63 (the window pass reads
// those), 0 above -- written by every cut (lazy apply kernels, eager kernel) and every push.  A mark never equals the working value,
// so  natural <=> (cur < 255 ? res8 == cur : res8 == 255)  holds for every alive entry, whatever formulation processed the levels
// above: the window pass tells natural from truncated entries by it, and the lists of either formulation can be rebuilt from cur[].
__device__ __forceinline__ u8 res8_mark(u32 t) { return (u8)(t <= 63u ? (0x80u | t) : 0u); }
__device__ __forceinline__ u8 res8_pushed(u32 t) { return res8_mark(t); }
// a factor of length L starts at p: the dense u32 array, or -- FactorSpace::flen8 -- the byte array (255: the length is flen[p])
__device__ __forceinline__ void put_flen(u32* __restrict__ flen, u8* __restrict__ flen8, u32 p, u32 L) {
    if (flen8) { flen8[p] = (u8)(L < 255u ? L : 255u); if (L >= 255u) flen[p] = L; }
    else flen[p] = L;
}

// ---- candidates ("Fill candidates", :54-66) ----------------------------------------------------------------
// cls[p] = 1 for the candidates whose level is above `lo` (the lists of the levels <= lo are only materialised if the
// window pass fails); *d_entries counts all candidates ("entries" of the reference's log)
// lvl_hist (nullable, 64 counters): every 16th candidate of a level below 64 is counted -- an estimate of how many entries a window of
// the window pass will find in one level (its per-level LDS lists come in two sizes)
// src_sa != nullptr (no Phi array: fused scatter without it): the source of a factor at p is SA[ISA[p] - 1] (ds/PhiFromSA.hpp:35-45); the
// candidates of the global levels get it NOW, into fsrc[] -- their priorities (= ISA) may be overwritten by a push before they are selected
__global__ __launch_bounds__(256) void cand_class_kernel(const u32* __restrict__ plcp, size_t n, u32 threshold, u32 lo, u8* __restrict__ cls,
                                                          u32* __restrict__ flen, u8* __restrict__ res8, u32* __restrict__ d_entries,
                                                          u32* __restrict__ lvl_hist, const u32* __restrict__ src_sa, const u32* __restrict__ src_isa,
                                                          u32* __restrict__ fsrc, u8* __restrict__ flen8) {
    __shared__ u32 sm[4];
    __shared__ u32 sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = 0;
    __syncthreads();
    u32 cnt = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // four positions per thread and step: one 16-byte load, 16- / 4-byte stores (the arrays are 16-byte aligned: arena)
    const size_t n4 = n / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const uint4 x = ((const uint4*)plcp)[q];
        const u32 v[4] = { x.x, x.y, x.z, x.w };
        u32 cw = 0, rw = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 is_cand = (v[j] >= threshold) ? 1u : 0u;   // PLCP[n-1] = 0, so the sentinel (SA index 0) is never a candidate
            cw |= ((is_cand && v[j] > lo) ? 1u : 0u) << (8 * j);
            rw |= (is_cand ? (v[j] > 255u ? 255u : v[j]) : 0u) << (8 * j);     // list that holds the entry of p (saturated)
            cnt += is_cand;
            if (lvl_hist && is_cand && v[j] < 64u && j == 0 && (q & 3) == 0) atomicAdd(&sh[v[j]], 1u);     // (every 16th position)
            if (src_sa && is_cand && v[j] > lo) { const u32 r = src_isa[4 * q + j]; fsrc[4 * q + j] = r ? src_sa[r - 1] : src_sa[n - 1]; }
        }
        ((u32*)cls)[q] = cw;
        if (flen8) ((u32*)flen8)[q] = 0; else ((uint4*)flen)[q] = make_uint4(0, 0, 0, 0);
        if (res8) ((u32*)res8)[q] = rw;
    }
    for (size_t p = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const u32 v = plcp[p];
        const u32 is_cand = (v >= threshold) ? 1u : 0u;
        cls[p] = (is_cand && v > lo) ? 1 : 0;
        if (flen8) flen8[p] = 0; else flen[p] = 0;
        if (res8) res8[p] = is_cand ? (u8)(v > 255u ? 255u : v) : (u8)0;
        cnt += is_cand;
        if (lvl_hist && is_cand && v < 64u && (p & 15) == 0) atomicAdd(&sh[v], 1u);
        if (src_sa && is_cand && v > lo) { const u32 r = src_isa[p]; fsrc[p] = r ? src_sa[r - 1] : src_sa[n - 1]; }
    }
    cnt = wave_reduce_sum(cnt);
    if (lane_id() == 0) sm[wave_id()] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { const u32 t = sm[0] + sm[1] + sm[2] + sm[3]; if (t) atomicAdd(d_entries, t); }   // one atomic per workgroup (capped grid)
    if (lvl_hist && threadIdx.x < 64 && sh[threadIdx.x]) atomicAdd(&lvl_hist[threadIdx.x], sh[threadIdx.x]);
}
__global__ void gather_kernel(const u32* __restrict__ idx, size_t m, const u32* __restrict__ src, u32* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[i] = src[idx[i]];
}
__global__ void seg_bounds_kernel(const u32* __restrict__ keys, size_t m, u32* __restrict__ segstart, u32* __restrict__ segend) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const u32 k = keys[j];
    if (j == 0 || keys[j - 1] != k) segstart[k] = (u32)j;
    if (j + 1 == m || keys[j + 1] != k) segend[k] = (u32)(j + 1);
}

// ---- per level ---------------------------------------------------------------------------------------------
// classify the entries of the level: live / stale / dead; live entries are marked undecided in the bitmap
__global__ void classify_kernel(const u32* __restrict__ orig, u32 m0, const u32* __restrict__ pushed, u32 m, u32 L,
                                u32 threshold, const u32* __restrict__ cur, u32* __restrict__ ent, u8* __restrict__ cls,
                                u64* __restrict__ bm) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const u32 p = (k < m0) ? orig[k] : pushed[k - m0];
    const u32 v = cur[p];
    const u8 c = (v == L) ? CL_LIVE : (v >= threshold ? CL_STALE : CL_DEAD);
    ent[k] = p;
    cls[k] = c;
    if (c == CL_LIVE) atomicOr((unsigned long long*)&bm[p >> 5], 1ull << (2 * (p & 31)));
}

// Mid-size levels: the same classification, but the live / stale lists are filled by wave-aggregated atomic appends (lists are
// unordered sets; position order only buys locality, which a list of < 2^20 entries does not need) -- one launch instead
// of seven.
__device__ __forceinline__ u32 wave_append(bool want, u32* counter) {
    const u64 b = __ballot(want);
    if (!b) return 0;
    const int leader = __builtin_ctzll(b);
    u32 base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, (u32)__popcll(b));
    base = __shfl(base, leader, 64);
    return base + (u32)__popcll(b & ((lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()))));
}
__global__ void classify_append_kernel(const u32* __restrict__ orig, u32 m0, const u32* __restrict__ pushed, u32 m, u32 L,
                                       u32 threshold, const u32* __restrict__ cur, u32* __restrict__ live, u32* __restrict__ stale,
                                       u64* __restrict__ bm, LevelScalars* __restrict__ sc) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    u32 p = 0, v = 0;
    if (k < m) { p = (k < m0) ? orig[k] : pushed[k - m0]; v = cur[p]; }
    const bool is_live = k < m && v == L, is_stale = k < m && v != L && v >= threshold;
    const u32 il = wave_append(is_live, &sc->nlive);
    const u32 is = wave_append(is_stale, &sc->nstale);
    if (is_live) { live[il] = p; atomicOr((unsigned long long*)&bm[p >> 5], 1ull << (2 * (p & 31))); }
    if (is_stale) stale[is] = p;
}

// Selection rounds over the live entries.  G lanes cooperate on one entry (1 for short levels, 64 for long ones).
// A blocked entry retries a few times inside the launch: the bitmap words are re-read with agent-scope (L1-bypassing)
// loads, so decisions of other workgroups become visible without a kernel boundary.  Safe without any ordering:
// a state only ever moves undecided -> {selected, rejected}, and a stale "undecided" merely postpones a decision.
constexpr int MIS_TRIES = 6;

template <int G>
__global__ __launch_bounds__(256) void mis_round_kernel(const u32* __restrict__ live, u32 nl, u32 L, size_t n,
                                                         const u32* __restrict__ prio, u64* bm, LevelScalars* __restrict__ sc) {
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    if (i >= nl) return;                                  // whole groups leave together
    const u32 p = live[i];
    if (((bm_load(&bm[p >> 5]) >> (2 * (p & 31))) & 3u) != 1u) return;   // already decided (uniform inside a group)
    const u32 pr = prio[p];
    const size_t lo = (p >= L - 1) ? (size_t)p - (L - 1) : 0;
    size_t hi = (size_t)p + (L - 1);
    if (hi > n - 1) hi = n - 1;
    for (int attempt = 0; attempt < MIS_TRIES; ++attempt) {
        bool hit = false, blocked = false;
        for (size_t w = (lo >> 5) + sub; w <= (hi >> 5) && !hit; w += G) {
            const u64 word = bm_load(&bm[w]) & bm_range_mask(w, lo, hi);
            if (word & BM_SELECTED_ALL) { hit = true; break; }   // a selected neighbour always outranks an undecided entry
            u64 und = word & BM_UNDECIDED_ALL;
            if (w == (p >> 5)) und &= ~(1ull << (2 * (p & 31)));
            while (und) {
                const int b = __builtin_ctzll(und);
                und &= und - 1;
                if (prio[w * 32 + (b >> 1)] < pr) { blocked = true; break; }
            }
        }
        if (G > 1) {
            hit = __any(hit);
            blocked = __any(blocked);
        }
        // one atomic per decision: xor 01 = undecided -> rejected (00), xor 11 = undecided -> selected (10)
        if (hit) { if (sub == 0) atomicXor((unsigned long long*)&bm[p >> 5], 1ull << (2 * (p & 31))); return; }
        if (!blocked) { if (sub == 0) atomicXor((unsigned long long*)&bm[p >> 5], 3ull << (2 * (p & 31))); return; }
        __builtin_amdgcn_s_sleep(8);
    }
    if (sub == 0) atomicAdd(&sc->undecided, 1u);
}

// Encounter value of the non-selected entries (:85-89): key = (target << 32) | old priority, rc = 1 if pushed down.
template <int G>
__global__ __launch_bounds__(256) void resolve_kernel(const u32* __restrict__ list, u32 cnt, bool live_list, u32 L, u32 threshold,
                                                       size_t n, const u32* __restrict__ prio, const u64* __restrict__ bm,
                                                       const u32* __restrict__ cur, u64* __restrict__ rkey, u32* __restrict__ rval,
                                                       u8* __restrict__ rc, u32* __restrict__ append_counter) {
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    if (i >= cnt) { if (append_counter && G == 1) (void)wave_append(false, append_counter); return; }
    const u32 p = list[i];
    const bool skip = live_list && bm_state(bm, p) == 2u;      // selected entries of the live list are not pushed
    u32 v = skip ? 0u : cur[p];
    const u32 pr = prio[p];
    if (!skip) {
        const size_t lo = (p >= L - 1) ? (size_t)p - (L - 1) : 0;
        size_t hi = (size_t)p + (L - 1);
        if (hi > n - 1) hi = n - 1;
        for (size_t w = (lo >> 5) + sub; w <= (hi >> 5) && v; w += G) {
            u64 sel = bm[w] & bm_range_mask(w, lo, hi) & BM_SELECTED_ALL;
            while (sel) {
                const int b = __builtin_ctzll(sel);
                sel &= sel - 1;
                const size_t q = w * 32 + (b >> 1);
                if (prio[q] >= pr) continue;             // selected later in the list: does not affect the encounter value
                if (q < p) { v = 0; break; }             // covered by a factor starting to the left   (:99-101)
                const u32 d = (u32)(q - p);               // truncated by a factor starting to the right (:103-109)
                if (d < v) v = d;
            }
        }
    }
    if (G > 1) {
        v = wave_reduce_min(v);
        if (sub != 0) return;
    }
    const bool push = v >= threshold;
    if (append_counter) {                                   // mid-size levels: push records appended directly (sorted afterwards)
        u32 o;
        if (G == 1) o = wave_append(push, append_counter);
        else o = push ? atomicAdd(append_counter, 1u) : 0u;  // one entry per wave
        if (push) { rkey[o] = ((u64)v << 32) | pr; rval[o] = p; }
        return;
    }
    rc[i] = push ? 1 : 0;
    rkey[i] = ((u64)v << 32) | pr;
    rval[i] = p;
}

// ---- whole level in ONE workgroup -----------------------------------------------------------------------------
// Levels with at most SMALL_M entries (the long tail of every text, and after a purge most levels of a text with long
// repeats) are processed by a single launch: classify, selection rounds, encounter values, sort of the pushes
// (bitonic, LDS), new priorities / pool slots / segments, apply.  One read-back per level instead of four or five.
constexpr u32 SMALL_M = 2048;

constexpr u32 SMALL_APPLY_INLINE_MAX_L = 64;      // longer factors are applied by a separate, chip-wide launch

// pushed part of a small level's list: read straight from the pool segments; their table (up to SMALL_GATHER entries) sits in
// mapped host memory and is copied into LDS first
constexpr u32 SMALL_GATHER = 2048;
constexpr u32 SMALL_RAW = 8192;     // list entries a small level may hold before the erased ones are dropped (SMALL_M must survive)
// A second instance of the kernel with 512 threads holds twice as many survivors: texts with long repeats have thousands of levels in a
// row whose lists hold 5 000 - 10 000 entries, 2 500 - 5 000 of them still alive (10^9 B of DNA: 1 383 levels, 0.36 ms each on the
// multi-launch path against ~0.06 ms here).
constexpr u32 SMALL_M_BIG = 4096;
// ... and a third one with 1 024 threads holds 8 192 (the SLIM layout of the kernel: LDS is what limits it)
constexpr u32 SMALL_M_SLIM = 8192;
constexpr u32 SMALL_RAW_SLIM = 32768;
constexpr u32 SMALL_OUT_WORDS = 8 + 2 * SEG_INLINE;
constexpr u32 SMALL_RANKSORT = 1024; // up to here a counting sort in LDS beats the bitonic network
constexpr u32 SMALL_SELSCAN = 32;    // up to this many selected entries the encounter values scan the selected list, not the neighbours

template <int NT, bool SLIM = false>
__global__ __launch_bounds__(NT) void small_level_kernel(const u32* __restrict__ orig, u32 m0_arg, const u32* __restrict__ pushed, u32 m_raw_arg,
                                                           const u32* __restrict__ pool_all, const GatherSeg* __restrict__ gtab, u32 gn,
                                                           u32 L, u32 threshold, size_t n, u32* cur, u32* prio,
                                                           const u32* phi, u32* __restrict__ flen, u8* __restrict__ res8,
                                                           u32* fsrc, u32 pool_top_arg, u32 prio_base_arg,      // (phi may BE fsrc -- no Phi array: fsrc[p] = phi[p] -- so neither is restrict)
                                                           PushSeg* __restrict__ segs, u32 seg_cap, u32* __restrict__ sel_list,
                                                           u32 inline_budget, LevelScalars* __restrict__ sc,
                                                           u32* zc_dst, u32* zc_flag, u32 zc_seq, unsigned long long* prof, u32* zc_segs,
                                                           SmallCtl* __restrict__ ctl, u32 spec, const LevelScalars* prev_sc, const PushSeg* prev_segs,
                                                           const u32* __restrict__ m0_dev, u8* __restrict__ flen8) {
    // Everything the first step needs from global memory is requested before anything waits: the purged length of the list, the first
    // entries of the original candidates (entries behind the purged length are copies of one erased entry: harmless) and, further
    // down, the gather table -- a one-workgroup kernel has nothing else to hide these round trips behind (20 us of the 40 a level takes).
    const u32 m0_purged = m0_dev ? *m0_dev : NONE32;
    u32 pre_orig[8];
#pragma unroll
    for (u32 r = 0; r < 8; ++r) { const u32 i = r * (u32)NT + threadIdx.x; pre_orig[r] = (i < m0_arg) ? orig[i] : NONE32; }
#define SPROF(k) do { if (prof) { const unsigned long long now_ = wall_clock64(); acc_prof[k] = now_ - t_prof; t_prof = now_; } } while (0)
    constexpr u32 SM = (u32)NT * 8;                          // survivors the workgroup holds (eight per thread)
    constexpr u32 NWV = (u32)NT / 64;
    unsigned long long t_prof = prof ? wall_clock64() : 0;
    unsigned long long acc_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // The entries are sorted by text position (bitonic network in registers), so "neighbours within distance < L" are
    // adjacent slots of LDS arrays: the whole level runs without the global state bitmap.
    // SLIM (1 024 threads, 8 192 survivors): 15 bytes of LDS per survivor instead of 27 -- the sort scratch lies over the priority / value
    // arrays and the target table (all dead while the positions are sorted), a push is recorded IN PLACE (value and state of its
    // entry) and ordered by the counting sort on the target only; a level that needs the general orderings goes to the multi-launch path.
    constexpr u32 HB = SLIM ? 13u : (NT >= 512 ? 12u : 11u);    // target counters: 2^HB
    __shared__ __align__(16) u64 skey_a[SLIM ? 1 : SM];         // sort scratch, later the push records (target << 32 | priority)
    __shared__ u32 sval_a[SLIM ? 1 : SM];
    __shared__ u32 pos_s[SM];
    __shared__ __align__(16) u32 pr_v[2 * SM];
    __shared__ u32 s_hcnt[1u << HB];
    u32* const pr_s = pr_v;
    u32* const v_s = pr_v + SM;
    u64* const skey = SLIM ? (u64*)pr_v : skey_a;
    u32* const sval = SLIM ? s_hcnt : sval_a;
    static_assert(!SLIM || (1u << HB) >= SM, "the table doubles as the 32-bit sort scratch");
    __shared__ u8 st[SM];             // 0 undecided, 1 selected, 2 stale, 3 rejected, 4 dead, 5 (SLIM) pushes: v_s holds the target
    __shared__ u32 s_und, s_npush, s_sel, s_live, s_alive, s_cnt, s_lst;
    __shared__ u32 s_sellist[SMALL_SELSCAN];
    __shared__ u32 s_out[SMALL_OUT_WORDS]; // the LevelScalars of this level: nlive nstale undecided selected npush nseg deferred bailed, segments
    const u32 tid = threadIdx.x;
    __shared__ u32 s_bail;
    GatherSeg* gt = (GatherSeg*)skey;                     // skey is not used before the sorts
    for (u32 i = tid; i < gn; i += NT) gt[i] = gtab[i];  // (mapped host memory: the longest of the round trips)
    // (level_purge_kernel may have shortened the original part of the list: the rest of the segment are copies of one erased entry)
    u32 m0 = m0_arg, m_raw = m_raw_arg;
    if (m0_purged < m0) { m_raw -= m0 - m0_purged; m0 = m0_purged; }
    if (tid == 0) { s_npush = 0; s_sel = 0; s_live = 0; s_alive = 0; s_cnt = 0; s_lst = 0; s_bail = 0; }
    if (tid < 8) s_out[tid] = 0;
    __syncthreads();
    u32 pool_top = pool_top_arg, prio_base = prio_base_arg;
    if (spec) {                                             // (see SmallCtl)
        const u32* ps = (const u32*)prev_sc;
        const u32 pn = ps[5];
        if (ps[6] || ps[7] || pn > seg_cap) { if (tid == 0) s_bail = 1; }
        else for (u32 i = tid; i < pn; i += NT) if (prev_segs[i].target == L) s_bail = 1;
        __syncthreads();
        pool_top = ctl->pool_top; prio_base = ctl->prio_base;
    } else if (tid == 0) { ctl->pool_top = pool_top; ctl->prio_base = prio_base; }
    u32* const pool = const_cast<u32*>(pool_all) + pool_top;
    // the result goes to the scalars block and, without a further launch, into the mapped host block the host spins on
    auto publish = [&]() {
        __syncthreads();
        const u32 words = 8 + 2 * min(s_out[5], SEG_INLINE);
        for (u32 i = tid; i < words; i += NT) {
            ((u32*)sc)[i] = s_out[i];
            if (zc_dst) __hip_atomic_store(&zc_dst[i], s_out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (zc_dst) {
            __threadfence_system();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(zc_flag, zc_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    if (s_bail) { if (tid == 0) s_out[7] = 3; publish(); return; }     // the speculation failed: nothing has been touched
    // 0. drop the erased entries (texts with long repeats carry thousands of them per level).  Eight entries per thread and
    //    step: the position loads are all in flight together, then the eight dependent cur[] loads (one workgroup has no
    //    other way to hide the two round trips)
    // The survivors keep the order of the list (row-wise ballots + a prefix over the rows and waves of a step): the original
    // candidates of a level are in position order, so their survivors come out sorted and only the pushed ones -- usually a
    // handful -- have to be ranked against them (step 1).
    __shared__ u32 s_rowcnt[8][NWV];
    __shared__ u32 s_run, s_nA;
    if (tid == 0) { s_run = 0; s_nA = 0; }
    __syncthreads();
    for (u32 base = 0; base < m_raw; base += NT * 8) {
        u32 pp[8], cc[8];
#pragma unroll
        for (u32 r = 0; r < 8; ++r) {
            const u32 i = base + r * NT + tid;
            pp[r] = NONE32;
            if (i < m0) pp[r] = (base == 0) ? pre_orig[r] : orig[i];
            else if (i < m_raw) {
                if (gn == 0) pp[r] = pushed[i - m0];
                else {
                    const u32 q = i - m0;
                    u32 lo = 0, hi = gn - 1;              // last segment with dst_off <= q
                    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (gt[mid].dst_off <= q) lo = mid; else hi = mid - 1; }
                    pp[r] = pool_all[gt[lo].src_off + (q - gt[lo].dst_off)];
                }
            }
        }
#pragma unroll
        for (u32 r = 0; r < 8; ++r) cc[r] = (pp[r] != NONE32) ? cur[pp[r]] : 0u;
        u32 within[8];
        u32 keepm = 0;
#pragma unroll
        for (u32 r = 0; r < 8; ++r) {
            const bool keep = pp[r] != NONE32 && cc[r] >= threshold;
            const u64 bm = __ballot(keep);
            within[r] = (u32)__popcll(bm & ((1ull << (tid & 63)) - 1ull));
            if ((tid & 63) == 0) s_rowcnt[r][tid >> 6] = (u32)__popcll(bm);
            if (keep) keepm |= 1u << r;
        }
        __syncthreads();
        const u32 run0 = s_run;
        u32 nA_add = 0;
        // slot of (row r, wave w, lane): everything in earlier rows, earlier waves of the row, earlier lanes of the wave
        u32 before_row = 0;
#pragma unroll
        for (u32 r = 0; r < 8; ++r) {
            u32 bw = 0;
#pragma unroll
            for (u32 wq = 0; wq < NWV; ++wq) if (wq < (tid >> 6)) bw += s_rowcnt[r][wq];
            if (keepm & (1u << r)) {
                const u32 slot = run0 + before_row + bw + within[r];
                if (slot < SM) pos_s[slot] = pp[r];
                if (base + r * NT + tid < m0) ++nA_add;
            }
            for (u32 wq = 0; wq < NWV; ++wq) before_row += s_rowcnt[r][wq];
        }
        nA_add = wave_reduce_sum(nA_add);
        if ((tid & 63) == 0 && nA_add) atomicAdd(&s_nA, nA_add);
        __syncthreads();
        if (tid == 0) { s_run = run0 + before_row; s_cnt = run0 + before_row; }
        __syncthreads();
    }
    __syncthreads();
    SPROF(0);
    const u32 m = s_cnt;
    if (m > SM) { if (tid == 0) s_out[7] = 1; publish(); return; }     // too many survivors: the general path takes the level
    if (m == 0) { publish(); return; }                                      // every entry already erased (:86)
    u64 k[8];
    u32 v[8];
    // 1. sort by position (positions are distinct)
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const u32 i = tid * 8 + r;
        k[r] = (i < m) ? (u64)pos_s[i] : ~0ull;
        v[r] = 0;
    }
    __syncthreads();
    const u32 nA = s_nA, nB = m - nA;                       // survivors of the original candidates (in list order) / of the pushed part
    // are the originals' survivors really ascending?  (they are whenever the candidate segment is in position order; checked, not assumed)
    __shared__ u32 s_unsorted;
    if (tid == 0) s_unsorted = 0;
    __syncthreads();
    for (u32 i = tid + 1; i < nA; i += NT) if (pos_s[i - 1] >= pos_s[i]) s_unsorted = 1;
    __syncthreads();
    if (!s_unsorted && (u64)nA * nB + (u64)nB * nB <= (u64)NT * 768ull) {
        // sorted run A + a few unsorted entries B: an entry of A moves up by the number of smaller entries of B, an entry of B goes
        // behind the smaller entries of A (binary search) and of B
        for (u32 i = tid; i < m; i += NT) {
            const u32 p = pos_s[i];
            u32 rk;
            if (i < nA) {
                rk = i;
                for (u32 j = nA; j < m; ++j) rk += (pos_s[j] < p) ? 1u : 0u;
            } else {
                u32 lo = 0, hi = nA;                          // number of entries of A below p
                while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (pos_s[mid] < p) lo = mid + 1; else hi = mid; }
                rk = lo;
                for (u32 j = nA; j < m; ++j) rk += (pos_s[j] < p) ? 1u : 0u;
            }
            sval[rk] = p;
        }
        __syncthreads();
#pragma unroll
        for (u32 r = 0; r < 8; ++r) { const u32 i = tid * 8 + r; k[r] = (i < m) ? (u64)sval[i] : ~0ull; }
        __syncthreads();
    } else if (m <= SMALL_RANKSORT) {
        // few survivors: every entry counts the smaller ones (independent LDS reads: no chain of dependent steps)
        for (u32 i = tid; i < m; i += NT) {
            const u32 p = pos_s[i];
            u32 rk = 0;
#pragma unroll 16
            for (u32 j = 0; j < m; ++j) rk += (pos_s[j] < p) ? 1u : 0u;
            sval[rk] = p;
        }
        __syncthreads();
#pragma unroll
        for (u32 r = 0; r < 8; ++r) { const u32 i = tid * 8 + r; k[r] = (i < m) ? (u64)sval[i] : ~0ull; }
        __syncthreads();
    } else {
        u32 mp2 = 8;
        while (mp2 < m) mp2 <<= 1;
        block_bitonic_sort_2048(k, v, skey, sval, mp2);
    }
    SPROF(1);
    // 2. classify
    u32 nlive_t = 0, nalive_t = 0;
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const u32 i = tid * 8 + r;
        if (i < m) {
            const u32 p = (u32)k[r];
            const u32 val = cur[p];
            pos_s[i] = p; v_s[i] = val; pr_s[i] = prio[p];
            const u8 c = (val == L) ? 0 : (val >= threshold ? 2 : 4);
            st[i] = c;
            nlive_t += (c == 0) ? 1u : 0u;
            nalive_t += (c != 4) ? 1u : 0u;
        }
    }
    // (one atomic per wave: thousands of survivors counting on two LDS words one by one cost 10 us)
    nlive_t = wave_reduce_sum(nlive_t); nalive_t = wave_reduce_sum(nalive_t);
    if ((tid & 63) == 0) { if (nlive_t) atomicAdd(&s_live, nlive_t); if (nalive_t) atomicAdd(&s_alive, nalive_t); }
    __syncthreads();
    if (s_alive == 0) { publish(); return; }
    SPROF(2);
    // 3. selection rounds (Jacobi: decisions are published after a barrier)
    for (u32 round = 0; round <= m && s_live; ++round) {
        if (tid == 0) s_und = 0;
        __syncthreads();
        u8 dec[8];
#pragma unroll
        for (u32 r = 0; r < 8; ++r) {
            const u32 i = tid * 8 + r;
            dec[r] = 255;
            if (i < m && st[i] == 0) {
                const u32 p = pos_s[i], pr = pr_s[i];
                bool hit = false, blocked = false;
                for (u32 j = i; j-- > 0;) {               // left neighbours
                    if (p - pos_s[j] >= L) break;
                    const u8 sj = st[j];
                    if (sj == 1) { hit = true; break; }
                    if (sj == 0 && pr_s[j] < pr) blocked = true;
                }
                for (u32 j = i + 1; j < m && !hit; ++j) { // right neighbours
                    if (pos_s[j] - p >= L) break;
                    const u8 sj = st[j];
                    if (sj == 1) { hit = true; break; }
                    if (sj == 0 && pr_s[j] < pr) blocked = true;
                }
                if (hit) dec[r] = 3; else if (!blocked) dec[r] = 1; else atomicAdd(&s_und, 1u);
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 r = 0; r < 8; ++r) { const u32 i = tid * 8 + r; if (dec[r] != 255) st[i] = dec[r]; }
        const u32 und = s_und;                            // read before the barrier, reset (by thread 0) after it
        __syncthreads();
        if (und == 0) break;
    }
    __syncthreads();
    SPROF(3);
    // 4. encounter values of the stale and the rejected entries -> push records
    if (tid == 0) s_lst = 0;
    __syncthreads();
    for (u32 i = tid; i < m; i += NT) if (st[i] == 1) { const u32 o = atomicAdd(&s_lst, 1u); if (o < SMALL_SELSCAN) s_sellist[o] = i; }
    __syncthreads();
    const u32 nsel_scan = s_lst;
    for (u32 i0 = 0; i0 < m; i0 += NT) {
        const u32 i = i0 + tid;
        const u8 c = (i < m) ? st[i] : (u8)4;
        bool pushes = false;
        if (c == 2 || c == 3) {
        const u32 p = pos_s[i], pr = pr_s[i];
        u32 val = v_s[i];
        if (nsel_scan <= SMALL_SELSCAN) {                 // few factors in this level: look at them only
            for (u32 f = 0; f < nsel_scan && val; ++f) {
                const u32 j = s_sellist[f];
                if (pr_s[j] >= pr) continue;
                const u32 q = pos_s[j];
                if (q < p) { if (p - q < L) val = 0; }     // covered by a factor starting to the left   (:99-101)
                else if (q - p < L && q - p < val) val = q - p;   // truncated by a factor starting to the right (:103-109)
            }
        } else {
            for (u32 j = i; j-- > 0 && val;) {            // a selected left neighbour of higher priority covers p (:99-101)
                if (p - pos_s[j] >= L) break;
                if (st[j] == 1 && pr_s[j] < pr) val = 0;
            }
            for (u32 j = i + 1; j < m && val; ++j) {      // a selected right neighbour truncates (:103-109)
                const u32 d = pos_s[j] - p;
                if (d >= L) break;
                if (st[j] == 1 && pr_s[j] < pr && d < val) val = d;
            }
        }
        if (val >= threshold) {
            if constexpr (SLIM) { v_s[i] = val; st[i] = 5; pushes = true; }
            else {
                const u32 idx = atomicAdd(&s_npush, 1u);
                skey[idx] = ((u64)val << 32) | pr;
                sval[idx] = p;
            }
        }
        }
        if constexpr (SLIM) {                              // (one atomic per wave)
            const u64 bp = __ballot(pushes);
            if ((tid & 63) == 0 && bp) atomicAdd(&s_npush, (u32)__popcll(bp));
        }
    }
    __syncthreads();
    // 5. order the pushes by (target, old priority); afterwards pr_s[i] = target, v_s[i] = position of the i-th push.
    //    Only the order INSIDE a target list matters (new priorities are only ever compared inside one list, and the pool keeps one
    //    segment per target), and on texts with long repeats nearly every push of a level has a target of its own: targets are
    //    counted in a hashed table; a push whose slot it has for itself takes the next free place (any order), the others -- real
    //    duplicates and the rare hash collisions -- are ranked among themselves and follow.
    const u32 npush = s_npush;
    __shared__ unsigned short s_dl[SM];
    __shared__ u32 s_nu, s_nd;
    __shared__ u32 seg_sm[NWV + 1];
    if constexpr (SLIM) {
        // 5' + 6'. counting sort of the pushing entries on their target; pool slots, priorities and segments straight from the table
        if (npush && L > (1u << HB)) { if (tid == 0) { s_out[7] = 2; s_out[1] = m; } publish(); return; }      // code 2 + the survivors: this instance cannot order the pushes (nothing has been written yet)
        u32 nseg = 0;
        if (npush) {
            constexpr u32 TE = (1u << HB) / NT;
            for (u32 i = tid; i < (1u << HB); i += NT) s_hcnt[i] = 0;
            if (tid == 0) s_nd = 0;
            __syncthreads();
            for (u32 i = tid; i < m; i += NT) if (st[i] == 5) atomicAdd(&s_hcnt[v_s[i]], 1u);
            __syncthreads();
            u32 loc[TE], sum = 0, gmax = 0, nz = 0;
#pragma unroll
            for (u32 e = 0; e < TE; ++e) { loc[e] = s_hcnt[tid * TE + e]; sum += loc[e]; gmax = max(gmax, loc[e]); nz += loc[e] ? 1u : 0u; }
            u32 total;
            u32 start = block_exclusive_sum<u32, (int)NWV>(sum, seg_sm, total);
            __syncthreads();
            u32 o = block_exclusive_sum<u32, (int)NWV>(nz, seg_sm, nseg);
            gmax = wave_reduce_max(gmax);
            if ((tid & 63) == 0 && gmax) atomicMax(&s_nd, gmax);
            __syncthreads();
            if (s_nd > 64) {                                    // a crowded target: ranking inside its group would be quadratic
                __syncthreads();
                if (tid == 0) { s_out[7] = 2; s_out[1] = m; }
                publish();
                return;
            }
#pragma unroll
            for (u32 e = 0; e < TE; ++e) {
                const u32 t = tid * TE + e;
                s_hcnt[t] = start;
                if (loc[e]) {                                    // a segment of the pool: target t, first slot `start`
                    if (o < SEG_INLINE) { s_out[8 + 2 * o] = t; s_out[9 + 2 * o] = start; }
                    if (zc_segs && nseg > SEG_INLINE && o < SM) {
                        __hip_atomic_store(&zc_segs[2 * o], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(&zc_segs[2 * o + 1], start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    if (o < seg_cap) segs[o] = PushSeg{t, start};
                    ++o;
                }
                start += loc[e];
            }
            __syncthreads();
            for (u32 i = tid; i < m; i += NT) if (st[i] == 5) s_dl[atomicAdd(&s_hcnt[v_s[i]], 1u)] = (unsigned short)i;
            __syncthreads();                                    // s_hcnt[t] is the END of target t's group now
            for (u32 a = tid; a < npush; a += NT) {
                const u32 i = s_dl[a];
                const u32 t = v_s[i], pr = pr_s[i];
                const u32 lo = t ? s_hcnt[t - 1] : 0u, hi = s_hcnt[t];
                u32 rk = 0;
                if (hi - lo > 1) for (u32 b = lo; b < hi; ++b) rk += (pr_s[s_dl[b]] < pr) ? 1u : 0u;   // (priorities are distinct)
                const u32 f = lo + rk, p = pos_s[i];
                prio[p] = prio_base + f;
                pool[f] = p;
                if (res8) res8[p] = res8_pushed(t);
            }
        }
        if (tid == 0) { s_out[5] = nseg; s_out[4] = npush; s_sel = 0; ctl->pool_top = pool_top + npush; ctl->prio_base = prio_base + npush; }
    } else {
    bool ordered = false;
    if (npush > 64 && L <= (1u << HB)) {
        // Every target (< L) has a counter of its own: counting sort by target, then the members of a target's group -- mostly one, a
        // handful at most on texts with long repeats -- rank themselves by priority inside the group.  (The hashed counters below
        // send two pushes in five to the "duplicates" through collisions alone once a level pushes thousands of entries, and more than
        // 512 of those mean a full bitonic sort: 65 us of the 140 us such a level takes at 10^9 B of DNA.)
        constexpr u32 TE = (1u << HB) / NT;
        __shared__ u32 cs_sm[NWV + 1];
        for (u32 i = tid; i < (1u << HB); i += NT) s_hcnt[i] = 0;
        if (tid == 0) s_nd = 0;
        __syncthreads();
        for (u32 i = tid; i < npush; i += NT) atomicAdd(&s_hcnt[(u32)(skey[i] >> 32)], 1u);
        __syncthreads();
        u32 loc[TE], sum = 0, gmax = 0;
#pragma unroll
        for (u32 e = 0; e < TE; ++e) { loc[e] = s_hcnt[tid * TE + e]; sum += loc[e]; gmax = max(gmax, loc[e]); }
        u32 total;
        u32 start = block_exclusive_sum<u32, (int)NWV>(sum, cs_sm, total);
#pragma unroll
        for (u32 e = 0; e < TE; ++e) { s_hcnt[tid * TE + e] = start; start += loc[e]; }
        gmax = wave_reduce_max(gmax);
        if ((tid & 63) == 0 && gmax) atomicMax(&s_nd, gmax);
        __syncthreads();
        const u32 group_max = s_nd;
        __syncthreads();                                        // (s_nd is reset below)
        if (group_max <= 64) {                                  // (a crowded target: the general orderings below)
            for (u32 i = tid; i < npush; i += NT) s_dl[atomicAdd(&s_hcnt[(u32)(skey[i] >> 32)], 1u)] = (unsigned short)i;
            __syncthreads();                                    // s_hcnt[t] is the END of target t's group now, i.e. the start of the next one
            for (u32 a = tid; a < npush; a += NT) {
                const u32 i = s_dl[a];
                const u64 key = skey[i];
                const u32 t = (u32)(key >> 32);
                const u32 lo = t ? s_hcnt[t - 1] : 0u, hi = s_hcnt[t];
                u32 rk = 0;
                if (hi - lo > 1) for (u32 b = lo; b < hi; ++b) rk += (skey[s_dl[b]] < key) ? 1u : 0u;   // (keys are distinct: the priority is part of them)
                pr_s[lo + rk] = t;
                v_s[lo + rk] = sval[i];
            }
            __syncthreads();
            ordered = true;
        }
    }
    if (ordered) {
    } else if (npush > 64) {
        for (u32 i = tid; i < (1u << HB); i += NT) s_hcnt[i] = 0;
        if (tid == 0) { s_nu = 0; s_nd = 0; }
        __syncthreads();
        for (u32 i = tid; i < npush; i += NT) atomicAdd(&s_hcnt[((u32)(skey[i] >> 32) * 2654435761u) >> (32 - HB)], 1u);
        __syncthreads();
        for (u32 i0 = 0; i0 < npush; i0 += NT) {
            const u32 i = i0 + tid;
            const bool have = i < npush;
            const u64 key = have ? skey[i] : 0ull;
            const bool uniq = have && s_hcnt[((u32)(key >> 32) * 2654435761u) >> (32 - HB)] == 1u;
            const u64 um = __ballot(uniq), dm = __ballot(have && !uniq);
            u32 ub = 0, db = 0;
            if ((tid & 63) == 0) { if (um) ub = atomicAdd(&s_nu, (u32)__popcll(um)); if (dm) db = atomicAdd(&s_nd, (u32)__popcll(dm)); }
            ub = __builtin_amdgcn_readfirstlane(ub); db = __builtin_amdgcn_readfirstlane(db);
            const u64 lt = (1ull << (tid & 63)) - 1ull;
            if (uniq) { const u32 o = ub + (u32)__popcll(um & lt); pr_s[o] = (u32)(key >> 32); v_s[o] = sval[i]; }
            else if (have) s_dl[db + (u32)__popcll(dm & lt)] = (unsigned short)i;
        }
        __syncthreads();
        const u32 nu = s_nu, nd = s_nd;
        if (nd > 512) goto full_sort;                           // many shared targets: ranking them by counting would cost nd^2
        for (u32 a = tid; a < nd; a += NT) {
            const u32 i = s_dl[a];
            const u64 key = skey[i];
            u32 rk = 0;
            for (u32 b = 0; b < nd; ++b) rk += (skey[s_dl[b]] < key) ? 1u : 0u;      // (keys are distinct: the priority is part of them)
            pr_s[nu + rk] = (u32)(key >> 32);
            v_s[nu + rk] = sval[i];
        }
        __syncthreads();
    } else {
full_sort:
    if (npush <= SMALL_RANKSORT) {
        for (u32 i = tid; i < npush; i += NT) {
            const u64 key = skey[i];
            u32 rk = 0;
#pragma unroll 16
            for (u32 j = 0; j < npush; ++j) rk += (skey[j] < key) ? 1u : 0u;
            pr_s[rk] = (u32)(key >> 32);
            v_s[rk] = sval[i];
        }
        __syncthreads();
    } else {
#pragma unroll
        for (u32 r = 0; r < 8; ++r) {
            const u32 i = tid * 8 + r;
            k[r] = (i < npush) ? skey[i] : ~0ull;
            v[r] = (i < npush) ? sval[i] : NONE32;
        }
        u32 qp2 = 8;
        while (qp2 < npush) qp2 <<= 1;
        block_bitonic_sort_2048(k, v, skey, sval, qp2);
#pragma unroll
        for (u32 r = 0; r < 8; ++r) { const u32 i = tid * 8 + r; pr_s[i] = (u32)(k[r] >> 32); v_s[i] = v[r]; }
        __syncthreads();
    }
    }
    // 6. new priorities, pool slots, segments
    for (u32 i = tid; i < npush; i += NT) {
        const u32 p = v_s[i];
        prio[p] = prio_base + i;
        pool[i] = p;
        if (res8) res8[p] = res8_pushed(pr_s[i]);
    }
    {   // segment starts (a new target level), numbered in order of their start: flags + workgroup prefix sum
        u32 heads = 0, cnt = 0;
#pragma unroll
        for (u32 r = 0; r < 8; ++r) {
            const u32 i = tid * 8 + r;
            if (i < npush && (i == 0 || pr_s[i - 1] != pr_s[i])) { heads |= 1u << r; ++cnt; }
        }
        u32 nseg;
        u32 o = block_exclusive_sum<u32, (int)NWV>(cnt, seg_sm, nseg);
#pragma unroll
        for (u32 r = 0; r < 8; ++r) {
            if (!(heads & (1u << r))) continue;
            const u32 i = tid * 8 + r, tgt = pr_s[i];
            if (o < SEG_INLINE) { s_out[8 + 2 * o] = tgt; s_out[9 + 2 * o] = i; }
            if (zc_segs && nseg > SEG_INLINE && o < SM) {    // more than the inline part holds: the whole list goes straight into the mapped host block (no read-back)
                __hip_atomic_store(&zc_segs[2 * o], tgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&zc_segs[2 * o + 1], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (o < seg_cap) segs[o] = PushSeg{tgt, i};
            ++o;
        }
        if (tid == 0) { s_out[5] = nseg; s_out[4] = npush; s_sel = 0; ctl->pool_top = pool_top + npush; ctl->prio_base = prio_base + npush; }
    }
    }
    __syncthreads();
    SPROF(4);
    // 7. the selected entries: applied here (short factors: one wave per factor; long ones: the whole workgroup per factor
    //    while the level's total stays small) or listed for a chip-wide launch
    for (u32 i = tid; i < m; i += NT) if (st[i] == 1) atomicAdd(&s_sel, 1u);
    if (tid == 0) s_lst = 0;
    __syncthreads();
    const u32 nsel = s_sel;
    if (L <= SMALL_APPLY_INLINE_MAX_L || (u64)nsel * L <= inline_budget) {
        // one wave per factor (the factors of the level side by side); the truncation candidates are read eight per lane
        // and step, so one factor costs a round trip or two instead of one per 64 positions
        const u32 lane = tid & 63, wv = tid >> 6;
        for (u32 i = tid; i < m; i += NT) if (st[i] == 1) sval[atomicAdd(&s_lst, 1u)] = pos_s[i];   // sval: free again after step 6
        __syncthreads();
        for (u32 f = wv; f < nsel; f += NWV) {
            const u32 p = sval[f];
            if (lane == 0) { put_flen(flen, flen8, p, L); if (phi != fsrc) fsrc[p] = phi[p]; }      // (no Phi array: phi IS fsrc, already filled for this candidate)
            for (u32 j = lane; j < L && (size_t)p + j < n; j += 64) cur[p + j] = 0;
            const u32 aff = (L < p) ? L : p;
            for (u32 j0 = 0; j0 < aff; j0 += 64 * 8) {
                u32 vv[8];
#pragma unroll
                for (u32 r = 0; r < 8; ++r) { const u32 j = j0 + r * 64 + lane; vv[r] = (j < aff) ? cur[p - 1 - j] : 0u; }
#pragma unroll
                for (u32 r = 0; r < 8; ++r) { const u32 j = j0 + r * 64 + lane; if (j < aff && vv[r] > j + 1) { if (atomicMin(&cur[p - 1 - j], j + 1) > j + 1 && res8) res8[p - 1 - j] = res8_mark(j + 1); } }
            }
        }
    } else {
        for (u32 i = tid; i < m; i += NT) if (st[i] == 1) sel_list[atomicAdd(&s_lst, 1u)] = pos_s[i];
        if (tid == 0) s_out[6] = 1;                       // deferred: the host launches apply_list_kernel
    }
    if (tid == 0) { s_out[3] = nsel; s_out[0] = s_live; s_out[1] = s_alive - s_live; }
    __syncthreads();
    SPROF(5);
    publish();
    SPROF(6);
    if (prof && threadIdx.x == 0) {
        unsigned long long* pf = prof + (m > SMALL_RANKSORT ? 16 : 0);
        for (int q = 0; q < 7; ++q) pf[q] += acc_prof[q];
        pf[8] += 1; pf[9] += m_raw; pf[10] += m; pf[11] += nsel; pf[12] += npush;
    }
#undef SPROF
}

// ---- purge ahead: the lists of the next 64 levels, in place ------------------------------------------------------------------------
// Every factor of a level above leaves one erased candidate behind in each list below it (the ramp of its repeat), so on texts with long
// repeats a list of 10 000 candidates holds a few hundred alive ones -- and the one-workgroup level kernel spends half its time loading
// the rest.  The global purge (below) rewrites all lists still to come and costs more than it saves once the lists fit one workgroup.
// This kernel only touches the lists of the next 64 levels, one workgroup per list: alive entries first, in their old order (the level
// kernel relies on it), the rest of the segment filled with ONE of the list's erased positions -- such an entry stays erased for good
// (cur only decreases), every consumer drops it, and sixty gathers of one address cost one.  Segment bounds do not change.
struct PurgeWin { u32 start[64]; u32 cnt[64]; u32 level[64]; u32 n; };
constexpr u32 PURGE_CAP = 32768;      // (128 KB of LDS: as long as the longest list a one-workgroup level may hold)
__global__ __launch_bounds__(1024) void level_purge_kernel(u32* __restrict__ cand, PurgeWin W, u32 threshold, const u32* __restrict__ cur, u32* __restrict__ lcount) {
    __shared__ u32 buf[PURGE_CAP];
    __shared__ u32 s_w[16], s_dead;
    const u32 lv = blockIdx.x;
    if (lv >= W.n) return;
    const u32 m = W.cnt[lv];
    if (m < 1024 || m > PURGE_CAP) return;
    u32* list = cand + W.start[lv];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_dead = NONE32;
    __syncthreads();
    u32 base = 0;
    for (u32 i0 = 0; i0 < m; i0 += 1024 * 4) {
        u32 p[4];
        bool al[4];
#pragma unroll
        for (u32 r = 0; r < 4; ++r) { const u32 i = i0 + r * 1024 + tid; p[r] = (i < m) ? list[i] : NONE32; }
#pragma unroll
        for (u32 r = 0; r < 4; ++r) al[r] = p[r] != NONE32 && cur[p[r]] >= threshold;
#pragma unroll
        for (u32 r = 0; r < 4; ++r) {
            const u64 bm = __ballot(al[r]);
            if (lane == 0) s_w[wv] = (u32)__popcll(bm);
            __syncthreads();
            u32 off = base, tot = 0;
#pragma unroll
            for (u32 q = 0; q < 16; ++q) { const u32 t = s_w[q]; if (q < wv) off += t; tot += t; }
            if (al[r]) buf[off + (u32)__popcll(bm & ((1ull << lane) - 1ull))] = p[r];
            else if (p[r] != NONE32) s_dead = p[r];
            base += tot;
            __syncthreads();
        }
    }
    const u32 dead = s_dead;
    if (tid == 0) lcount[W.level[lv]] = base;               // (the level kernel stops there)
    if (base == m) return;                                  // nothing erased
    for (u32 i = tid; i < m; i += 1024) list[i] = (i < base) ? buf[i] : dead;
}

// apply for a list of selected positions whose length is only known on the device (one wave per factor)
__global__ __launch_bounds__(256) void apply_list_kernel(const u32* __restrict__ list, const u32* __restrict__ d_count, u32 L, size_t n,
                                                          const u32* phi, u32* __restrict__ cur, u32* __restrict__ flen,
                                                          u32* fsrc, u8* __restrict__ res8, u8* __restrict__ flen8) {      // (phi may be fsrc)
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 lane = threadIdx.x & 63;
    if (i >= *d_count) return;
    const u32 p = list[i];
    if (lane == 0) { put_flen(flen, flen8, p, L); if (phi != fsrc) fsrc[p] = phi[p]; }
    for (u32 j = lane; j < L && (size_t)p + j < n; j += 64) cur[p + j] = 0;
    const u32 aff = (L < p) ? L : p;
    for (u32 j = lane; j < aff; j += 64) { u32* q = &cur[p - 1 - j]; if (*q > j + 1) { if (atomicMin(q, j + 1) > j + 1 && res8) res8[p - 1 - j] = res8_mark(j + 1); } }
}

// ---- purge: drop the candidates that were erased by longer factors from all levels that are still to come ------------
__global__ void purge_class_kernel(const u32* __restrict__ cand, size_t cnt, u32 threshold, const u32* __restrict__ cur, u8* __restrict__ cls) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < cnt) cls[j] = (cur[cand[j]] >= threshold) ? 1 : 0;
}

// Highest level in a range of the sorted candidate array that still holds an alive entry (0: none).  Texts with long repeats
// have one erased candidate per level over tens of thousands of consecutive levels (the ramp PLCP = R, R-1, ... inside a
// repeat of length R, erased by the factor of level R): one probe replaces a launch and a read-back per level.
__global__ __launch_bounds__(256) void alive_max_level_kernel(const u32* __restrict__ levels, const u32* __restrict__ pos, size_t lo, size_t hi,
                                                               const u32* __restrict__ cur, u32 threshold, u32* __restrict__ d_max,
                                                               u32* __restrict__ d_maxcur = nullptr) {
    u32 best = 0, bestcur = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < hi; k += stride) {
        const u32 v = cur[pos[k]];
        if (v >= threshold) { best = max(best, levels[k]); bestcur = max(bestcur, v); }
    }
    best = wave_reduce_max(best);
    if (lane_id() == 0 && best) atomicMax(d_max, best);
    if (d_maxcur) { bestcur = wave_reduce_max(bestcur); if (lane_id() == 0 && bestcur) atomicMax(d_maxcur, bestcur); }
}

// The same question for pushed entries: highest target level among the given pool segments that still hold an alive entry.
struct ProbeSeg { u32 off, cnt, level; };
__global__ __launch_bounds__(256) void alive_max_pool_kernel(const u32* __restrict__ pool, const ProbeSeg* __restrict__ tab, size_t nseg,
                                                              const u32* __restrict__ cur, u32 threshold, u32* __restrict__ d_max,
                                                              u32* __restrict__ d_maxcur) {
    u32 best = 0, bestcur = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < nseg; k += stride) {
        const ProbeSeg sg = tab[k];
        if (!d_maxcur && sg.level <= best) continue;
        for (u32 j = 0; j < sg.cnt; ++j) {
            const u32 v = cur[pool[sg.off + j]];
            if (v >= threshold) { best = max(best, sg.level); bestcur = max(bestcur, v); if (!d_maxcur) break; }
        }
    }
    best = wave_reduce_max(best);
    if (lane_id() == 0 && best) atomicMax(d_max, best);
    if (d_maxcur) { bestcur = wave_reduce_max(bestcur); if (lane_id() == 0 && bestcur) atomicMax(d_maxcur, bestcur); }
}

// A run of levels (floor, top] without a live entry only moves its stale entries down (:85-89): collected in one go.  Record:
// key = (top - level) << 32 | old priority (the order in which the level-by-level loop would meet the entries), value = position.
__global__ __launch_bounds__(256) void stale_collect_orig_kernel(const u32* __restrict__ levels, const u32* __restrict__ pos, size_t lo, size_t hi,
                                                                  u32 floor_lv, u32 top, const u32* __restrict__ cur, const u32* __restrict__ prio,
                                                                  u32 threshold, u64* __restrict__ rkey, u32* __restrict__ rval, u32* __restrict__ d_count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k0 = lo + (size_t)blockIdx.x * blockDim.x; k0 < hi; k0 += stride) {
        const size_t k = k0 + threadIdx.x;
        u32 p = 0, lv = 0;
        bool want = false;
        if (k < hi) { lv = levels[k]; p = pos[k]; want = lv > floor_lv && lv <= top && cur[p] >= threshold; }
        const u32 o = wave_append(want, d_count);
        if (want) { rkey[o] = ((u64)(top - lv) << 32) | prio[p]; rval[o] = p; }
    }
}
__global__ __launch_bounds__(256) void stale_collect_pool_kernel(const u32* __restrict__ pool, const ProbeSeg* __restrict__ tab, size_t nseg,
                                                                  u32 floor_lv, u32 top, const u32* __restrict__ cur, const u32* __restrict__ prio,
                                                                  u32 threshold, u64* __restrict__ rkey, u32* __restrict__ rval, u32* __restrict__ d_count) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nseg) return;
    const ProbeSeg sg = tab[k];
    if (sg.level <= floor_lv || sg.level > top) return;
    for (u32 j = 0; j < sg.cnt; ++j) {
        const u32 p = pool[sg.off + j];
        if (cur[p] >= threshold) { const u32 o = atomicAdd(d_count, 1u); rkey[o] = ((u64)(top - sg.level) << 32) | prio[p]; rval[o] = p; }
    }
}
__global__ void stale_target_keys_kernel(const u32* __restrict__ vals, u32 m, const u32* __restrict__ cur, u64* __restrict__ keys) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) keys[i] = (u64)cur[vals[i]] << 32;
}

// Pushed part of a level's list: concatenation of its pool segments (table: source offset / destination offset).
__global__ void gather_segments_kernel(const u32* __restrict__ pool, const GatherSeg* __restrict__ tab, u32 nseg, u32 total,
                                       u32* __restrict__ dst) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    u32 lo = 0, hi = nseg - 1;                       // last segment with dst_off <= k
    while (lo < hi) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (tab[mid].dst_off <= k) lo = mid; else hi = mid - 1;
    }
    dst[k] = pool[tab[lo].src_off + (k - tab[lo].dst_off)];
}

// The level's pushes, sorted by (target, old priority): new residence, new priority, pool slot, segment starts.
__global__ void push_finalize_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals, u32 npush, u32 prio_base,
                                     u32* __restrict__ prio, u32* __restrict__ pool, u8* __restrict__ res8,
                                     PushSeg* __restrict__ segs, u32 seg_cap, LevelScalars* __restrict__ sc) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npush) return;
    const u32 tgt = (u32)(keys[i] >> 32);
    const u32 p = vals[i];
    prio[p] = prio_base + i;
    pool[i] = p;
    if (res8) res8[p] = res8_pushed(tgt);
    if (i == 0 || (u32)(keys[i - 1] >> 32) != tgt) {
        const u32 j = atomicAdd(&sc->nseg, 1u);
        if (j < SEG_INLINE) sc->segs[j] = PushSeg{tgt, i};
        if (j < seg_cap) segs[j] = PushSeg{tgt, i};
    }
}

// Emit the selected entries: factor (p, Phi[p], L); kill the covered positions, truncate the ones in front.
template <int G>
__global__ __launch_bounds__(256) void apply_kernel(const u32* __restrict__ live, u32 nl, u32 L, size_t n, u64* bm,
                                                     const u32* phi, u32* __restrict__ cur, u32* __restrict__ flen,
                                                     u32* fsrc, LevelScalars* __restrict__ sc, u8* __restrict__ res8, u8* __restrict__ flen8) {      // (phi may be fsrc)
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    bool sel = false;
    u32 p = 0;
    if (i < nl) { p = live[i]; sel = (bm_state(bm, p) == 2u); }
    if (sel) {
        if (sub == 0) { put_flen(flen, flen8, p, L); if (phi != fsrc) fsrc[p] = phi[p]; }
        for (u32 j = sub; j < L && (size_t)p + j < n; j += G) cur[p + j] = 0;   // :99-101
        const u32 aff = (L < p) ? L : p;                           // :103
        for (u32 j = sub; j < aff; j += G) {                         // :105-109
            u32* q = &cur[p - 1 - j];
            if (*q > j + 1) { if (atomicMin(q, j + 1) > j + 1 && res8) res8[p - 1 - j] = res8_mark(j + 1); }   // values only ever decrease: a stale read can only cause a redundant atomic
        }
    }
    // leave the bitmap all-zero for the next level (every reader of this level's state ran in an earlier kernel)
    if (G > 1) __builtin_amdgcn_wave_barrier();
    if (sel && sub == 0) atomicAnd((unsigned long long*)&bm[p >> 5], ~(3ull << (2 * (p & 31))));
    // factor count: one atomic per workgroup
    __shared__ u32 cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const u64 b = __ballot(sel && sub == 0);
    if (lane_id() == 0 && b) atomicAdd(&cnt, (u32)__popcll(b));
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(&sc->selected, cnt);
}

void factorize_arrays(Ctx& c, size_t n, const u32* sa, u32* isa, const u32* phi, u32* plcp, u32 maxlcp, u32 threshold,
                      FactorSpace& fs, FactorizeStats* st) {
    // (the candidate order of the reference -- ascending SA index -- is carried by prio[] = ISA.)  phi == nullptr: there is no Phi array;
    // the source of a factor at p is SA[ISA[p] - 1].  The candidates of the global levels get theirs into fsrc[] up front
    // (cand_class_kernel), so the global kernels read "Phi" from fsrc[] itself (fsrc[p] = fsrc[p] at a selection); the window kernel
    // computes it at its factor starts (their ISA is intact: window-local pushes do not touch prio[]).
    const u32* phi_eff = phi ? phi : fs.fsrc;
    FactorizeStats local;
    if (!st) st = &local;
    *st = FactorizeStats();
    st->maxlcp = maxlcp;
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    const unsigned gn = cdiv(n, 256);
    u32* cur = plcp;
    u32* prio = isa;

    // ---- candidates: positions with PLCP >= threshold, in position order, stably sorted by PLCP value -----------
    u8* cls = c.arena.get<u8>(n);
    // the low levels run window-local (factorize_tiles.hip) when the text is long enough; res8 = residence level per position
    u32 lcut = 0;
    if (c.window_lcut > 0 && n >= window_levels_min_text()) {
        lcut = std::min<u32>((u32)c.window_lcut, window_levels_max_lcut());
        if (lcut > maxlcp) lcut = maxlcp;
        if (lcut < threshold) lcut = 0;
    }
    u8* res8 = lcut ? c.arena.get<u8>(n) : nullptr;
    u32* ckeys[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    u32* cvals[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    u32* d_cnt = c.arena.get<u32>(4);
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 4 * sizeof(u32), s));
    u32* d_lvlhist = c.arena.get<u32>(64);
    HIP_TRY(hipMemsetAsync(d_lvlhist, 0, 64 * sizeof(u32), s));
    {
        Ctx::ProfScope prof(c, K_CAND, (u64)n * (lcut ? 10 : 9) - (fs.flen8 ? (u64)n * 3 : 0));
        cand_class_kernel<<<(gn < 8192u ? gn : 8192u), 256, 0, s>>>(plcp, n, threshold, lcut, cls, fs.flen, res8, d_cnt + 1, lcut ? d_lvlhist : nullptr,
                                                                    phi ? nullptr : sa, isa, fs.fsrc, fs.flen8);
        LAUNCH_CHECK();
    }
    if ((u64)maxlcp + 1 <= threshold || threshold == 0) { c.arena.release(mark); build_owner(c, n, fs); if (!phi) { fs.src_prio = prio; fs.src_sa = sa; fs.src_n = n; } return; }   // ArraysComp.hpp:50
    const size_t nlev = (size_t)maxlcp + 2;
    // first / last candidate index per level (device copies only live from seg_bounds_kernel to the read-back: they share the
    // memory of the push records of the general path; a text that is one long run has as many levels as positions)
    u64* rkey = c.arena.get<u64>(n + 2);
    u32* d_segstart = (u32*)rkey;
    u32* d_segend = d_segstart + nlev;
    std::vector<u32> h_segstart(nlev), h_segend(nlev);
    int x = 0;
    size_t cand_count = 0;
    // candidate lists: the class-1 positions in position order, stably sorted by their level (= the working value: the PLCP value at the start)
    auto build_lists = [&](u32 max_level) {
        select_by_class(c, cls, 1, n, nullptr, cvals[0], nullptr, nullptr, d_cnt);
        cand_count = c.read(d_cnt);
        if (cand_count) {
            Ctx::ProfScope prof(c, K_CAND, (u64)cand_count * 12);
            gather_kernel<<<cdiv(cand_count, 256), 256, 0, s>>>(cvals[0], cand_count, plcp, ckeys[0]);
            LAUNCH_CHECK();
        }
        x = radix_sort_pairs_u32(c, ckeys, cvals, cand_count, 0, (int)bits_for(max_level));
        HIP_TRY(hipMemsetAsync(d_segstart, 0, nlev * sizeof(u32), s));
        HIP_TRY(hipMemsetAsync(d_segend, 0, nlev * sizeof(u32), s));
        if (cand_count) {
            seg_bounds_kernel<<<cdiv(cand_count, 256), 256, 0, s>>>(ckeys[x], cand_count, d_segstart, d_segend);
            LAUNCH_CHECK();
        }
        c.read_n(d_segstart, h_segstart.data(), nlev);
        c.read_n(d_segend, h_segend.data(), nlev);
    };
    build_lists(maxlcp);
    st->entries = c.read(d_cnt + 1);
    const u32* cand = cvals[x];

    // ---- per-level state ------------------------------------------------------------------------------------
    u32* ent = c.arena.get<u32>(n);
    u32* live = c.arena.get<u32>(n);
    u32* stale = c.arena.get<u32>(n);
    const size_t bm_words = n / 32 + 2;
    u64* bm = c.arena.get<u64>(bm_words);       // 2 state bits per position; all-zero between levels
    HIP_TRY(hipMemsetAsync(bm, 0, bm_words * sizeof(u64), s));
    u8* rc = c.arena.get<u8>(n);
    u32* pool = c.arena.get<u32>(n);            // pushed entries, grouped by (source level, target)
    u32* pushed = c.arena.get<u32>(n);
    u32* rval = c.arena.get<u32>(n);
    u64* skeys[2] = { c.arena.get<u64>(n), c.arena.get<u64>(n) };
    u32* svals[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    LevelScalars* d_sc = (LevelScalars*)c.arena.alloc(sizeof(LevelScalars));
    LevelScalars* d_sc2[2] = { d_sc, (LevelScalars*)c.arena.alloc(sizeof(LevelScalars)) };     // two one-workgroup levels in flight
    SmallCtl* d_ctl = (SmallCtl*)c.arena.alloc(sizeof(SmallCtl));
    const u32 seg_cap = 1u << 16;
    PushSeg* d_segs = (PushSeg*)c.arena.alloc(sizeof(PushSeg) * seg_cap);
    PushSeg* d_segs2[2] = { d_segs, (PushSeg*)c.arena.alloc(sizeof(PushSeg) * seg_cap) };
    u32 force_general_level = NONE32;                      // a level the one-workgroup kernel gave up on
    std::vector<PushSeg> h_segs(seg_cap);
    LevelScalars h_sc;

    const u32 gtab_cap = 1u << 16;
    GatherSeg* d_gtab = (GatherSeg*)c.arena.alloc(sizeof(GatherSeg) * gtab_cap);
    // (pinned, so the H2D copy is truly asynchronous; owned by the context: allocating and freeing it per call cost a millisecond of idle device)
    GatherSeg* h_gtab = (GatherSeg*)c.pinned_table(sizeof(GatherSeg) * gtab_cap);
    GatherSeg* d_hgtab = nullptr;               // the same table as seen from the device (small levels read it in place)
    if (hipHostGetDevicePointer((void**)&d_hgtab, h_gtab, 0) != hipSuccess) { d_hgtab = nullptr; (void)hipGetLastError(); }
    struct PoolSeg { u32 off, cnt; };
    struct LevelLists {                                      // per target level: its segments of the pool.  Flat table while the number
        std::vector<std::vector<PoolSeg>> flat;              // of levels is moderate (a push costs one vector append: texts with long
        std::unordered_map<u32, std::vector<PoolSeg>> m;     // repeats make hundreds of them per level), a sparse map for texts that are
        void init(size_t levels) { if (levels <= ((size_t)1 << 20)) flat.resize(levels); }   // one long run (as many levels as positions)
        const std::vector<PoolSeg>& get(u32 lv) const {
            static const std::vector<PoolSeg> none;
            if (!flat.empty()) return lv < flat.size() ? flat[lv] : none;
            if (m.empty()) return none;
            auto it = m.find(lv);
            return it == m.end() ? none : it->second;
        }
        std::vector<PoolSeg>& operator[](u32 lv) { return !flat.empty() ? flat[lv] : m[lv]; }
        void drop(u32 lv) {
            if (!flat.empty()) { if (lv < flat.size()) std::vector<PoolSeg>().swap(flat[lv]); }
            else m.erase(lv);
        }
    } pushed_into;
    pushed_into.init((size_t)maxlcp + 2);
    size_t pool_top = 0;
    u32 prio_base = (u32)n;
    bool window_src_done = false;

    u32 dead_streak = 0, levels_since_purge = 1u << 30;
    bool purge_pays = false;                               // a level too large for the one-workgroup path consisted mostly of erased entries
    u32 dead_levels_run = 0;                               // consecutive levels whose entries were all erased
    u32 last_alive = 0;                                    // survivors of the last one-workgroup level (chooses the instance of the next one)
    int force_inst = -1;                                   // the instance a level is run again on
    u32 slim_penalty = 0;                                  // levels for which the 1 024-thread instance is not tried (it gave up on a level)
    // Two tables have one word (two) per LEVEL.  A text that is one long run has as many levels as positions: 12 more bytes per position
    // than tdc_gpu_arena_bytes() budgets (64 MB of one letter ran out of arena here).  Beyond 2^22 levels they live in an allocation of
    // their own for the duration of the call -- a hipMalloc / hipFree pair only such texts pay for.
    struct LevelTables { u32* p = nullptr; ~LevelTables() { if (p) (void)hipFree(p); } } level_tables;
    const bool levels_outside = nlev > ((size_t)1 << 22);
    if (levels_outside) HIP_TRY(hipMalloc((void**)&level_tables.p, 3 * nlev * sizeof(u32)));
    u32* d_lcount = levels_outside ? level_tables.p : c.arena.get<u32>(nlev);   // per level: entries of the original list that are left after level_purge_kernel (all ones: not purged)
    HIP_TRY(hipMemsetAsync(d_lcount, 0xFF, nlev * sizeof(u32), s));
    u32 purge_next = 0xFFFFFFFFu;                          // level_purge_kernel has been run for the levels >= purge_next
    auto purge_ahead = [&](u32 lv) {                       // called before anything reads the list of level lv
        if (!c.level_purge || lv >= purge_next) return;
        PurgeWin W;
        W.n = 0;
        const u32 floor_lv = std::max<u32>(lcut + 1, threshold);
        u32 v = lv;
        bool any = false;
        for (; W.n < 64 && v >= floor_lv; --v) {
            W.start[W.n] = h_segstart[v];
            W.cnt[W.n] = h_segend[v] - h_segstart[v];
            W.level[W.n] = v;
            if (W.cnt[W.n] >= 1024 && W.cnt[W.n] <= PURGE_CAP) any = true;
            ++W.n;
            if (v == 0) break;
        }
        purge_next = (lv >= 64) ? lv - 63 : 0;
        if (!any) return;
        level_purge_kernel<<<W.n, 1024, 0, s>>>(cvals[x], W, threshold, cur, d_lcount);
        LAUNCH_CHECK();
    };
    u32 nolive_run = 0, stale_trigger = 8;                 // consecutive levels without a live entry; run length that triggers the batch push
    double host_prof[5] = {0, 0, 0, 0, 0};                // small levels, host side: prepare / launch / wait / bookkeeping (us), count
    unsigned long long* d_sprof = nullptr;                 // TDC_GPU_SMALL_PROF=1: phase times of the small-level kernel on stderr
    if (c.small_prof) { d_sprof = (unsigned long long*)c.arena.alloc(32 * sizeof(unsigned long long)); HIP_TRY(hipMemsetAsync(d_sprof, 0, 32 * sizeof(unsigned long long), s)); }
    bool probe_dead = false;                    // the previous level held erased candidates only

    const bool level_log = c.level_log != 0;     // debugging aid: one line per large level on stderr
    // ---- lists from cur[]: both formulations can restart from the working values and the residence marks alone (res8_mark) ----------
    // The lazy formulation: every alive position of the levels (lo, hi] becomes an entry of list cur[q] (natural ones keep their ISA as
    // priority, truncated ones follow them); the push pool is forgotten -- its alive entries are among the rebuilt ones.
    auto rebuild_from_cur = [&](u32 lo, u32 hi) {
        {
            Ctx::ProfScope prof(c, K_CAND, (u64)n * 5);
            lazy_rebuild_class(c, cur, n, lo, hi, cls);
        }
        build_lists(hi);
        cand = cvals[x];
        // (truncated entries take prio_base + their index inside their level: the rebuild consumes what its longest level holds)
        u32 longest = 0;
        for (u32 v = lo + 1; v <= hi && v < (u32)nlev; ++v) { longest = std::max(longest, h_segend[v] - h_segstart[v]); if (v == 0xFFFFFFFFu) break; }
        if ((u64)prio_base + longest > 0xFFFFFFFFull) throw HipError{hipErrorUnknown, "factorize: priority space exhausted", (int)__LINE__};
        lazy_rebuild_prio(c, cand, cand_count, cur, res8, n, prio, prio_base, d_segstart, phi ? nullptr : sa, fs.fsrc);
        prio_base += longest;
        for (u32 v = lo + 1; v <= hi; ++v) { pushed_into.drop(v); if (v == 0xFFFFFFFFu) break; }
        pool_top = 0;
        HIP_TRY(hipMemsetAsync(d_lcount, 0xFF, nlev * sizeof(u32), s));
        purge_next = 0xFFFFFFFFu;
        dead_streak = 0; levels_since_purge = 1u << 30; purge_pays = false; dead_levels_run = 0; nolive_run = 0; probe_dead = false;
        force_general_level = NONE32;
    };
    // The eager formulation (factorize_eager.hip): a run of small levels inside one launch.  Taken when the levels ahead are small and
    // many (texts with long repeats); at most a few phases per call, each framed by two dense passes.
    const u32 eager_floor = std::max<u32>(threshold, lcut ? lcut + 1 : 0);       // lowest level of the global loop
    // (the eager kernel packs an entry as position << 1 | truncated in 32 bits: n <= 2^31)
    const bool eager_possible = c.eager_levels && res8 && nlev <= ((size_t)1 << 22) && maxlcp > eager_floor + 256 && n <= ((size_t)1 << 31);
    u32 eager_phases = 0, small_lazy_streak = 0;
    u32* d_eseg = !eager_possible ? nullptr : (levels_outside ? level_tables.p + nlev : c.arena.get<u32>(2 * nlev));        // tstart | tend
    u32* d_ehead = eager_possible ? c.arena.get<u32>(nlev) : nullptr;
    EagerCtl* d_ectl = eager_possible ? (EagerCtl*)c.arena.alloc(sizeof(EagerCtl)) : nullptr;
    auto run_eager = [&](u32 Lfrom) -> u32 {                                     // returns the next level to be processed
        // run heads of the levels [eager_floor, Lfrom] (natural and truncated alike): dense pass, then sorted by level like the candidates
        {
            Ctx::ProfScope prof(c, K_CAND, (u64)n * 6);
            eager_heads_class(c, cur, n, eager_floor - 1, Lfrom, cls);
        }
        select_by_class(c, cls, 1, n, nullptr, ent, nullptr, nullptr, d_cnt);
        const size_t tcount = c.read(d_cnt);
        u32* tk[2] = { live, stale };
        u32* tv[2] = { ent, rval };
        int y = 0;
        HIP_TRY(hipMemsetAsync(d_eseg, 0, 2 * nlev * sizeof(u32), s));
        if (tcount) {
            gather_kernel<<<cdiv(tcount, 256), 256, 0, s>>>(ent, tcount, cur, live);
            LAUNCH_CHECK();
            y = radix_sort_pairs_u32(c, tk, tv, tcount, 0, (int)bits_for(Lfrom));
            seg_bounds_kernel<<<cdiv(tcount, 256), 256, 0, s>>>(tk[y], tcount, d_eseg, d_eseg + nlev);
            LAUNCH_CHECK();
        }
        HIP_TRY(hipMemsetAsync(d_ehead, 0, nlev * sizeof(u32), s));
        HIP_TRY(hipMemsetAsync(d_ectl, 0, sizeof(EagerCtl), s));
        EagerParams P;
        P.tcand = tv[y]; P.tstart = d_eseg; P.tend = d_eseg + nlev;
        P.head = d_ehead;
        P.blk = pool; P.blk_cap = (u32)std::min<size_t>(n / 16, (size_t)1 << 28);   // (the push pool of the lazy formulation is forgotten anyway)
        P.cur = cur; P.prio = prio; P.phi = phi_eff; P.flen = fs.flen; P.flen8 = fs.flen8; P.fsrc = fs.fsrc; P.res8 = res8;
        P.n = n; P.threshold = threshold; P.L_from = Lfrom; P.L_stop = eager_floor; P.raw_cap = eager_levels_raw_cap();
        P.ctl = d_ectl;
        P.dbg = nullptr;
        const size_t dbg_mark = c.arena.mark();
        if (level_log) { P.dbg = c.arena.get<u32>(4 * nlev); HIP_TRY(hipMemsetAsync(P.dbg, 0, 4 * nlev * sizeof(u32), s)); }
        {
            Ctx::ProfScope prof(c, K_SMALL_LEVEL, (u64)tcount * 13 + (u64)(Lfrom - eager_floor + 1) * 16);   // per run head: position, working value, residence byte, priority; per level: its two table words + head
            eager_levels_launch(c, P);
        }
        EagerCtl h;
        HIP_TRY(hipMemcpyAsync(&h, d_ectl, sizeof(EagerCtl), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (P.dbg) {
            std::vector<u32> dv(4 * nlev);
            HIP_TRY(hipMemcpy(dv.data(), P.dbg, 4 * nlev * sizeof(u32), hipMemcpyDeviceToHost));
            unsigned long long tot = 0;
            for (size_t lv = 0; lv < nlev; ++lv) tot += dv[4 * lv + 2];
            unsigned long long acc = 0; u32 shown = 0;
            const bool dump_all = c.eager_dump != 0;          // (tools/eager_levels_log.py)
            for (size_t lv = nlev; lv-- > 0;) {
                if (!dv[4 * lv + 2]) continue;
                if (dump_all) { fprintf(stderr, "eager level %zu: entries %u factors %u cycles %u\n", lv, dv[4 * lv], dv[4 * lv + 1], dv[4 * lv + 2]); continue; }
                acc += dv[4 * lv + 2];
                if (dv[4 * lv + 2] > 400000u || (shown++ % 256) == 0) fprintf(stderr, "eager level %zu: %u entries of %u candidates, %u factors, %u cycles (running %.1f %% of %llu)\n", lv, dv[4 * lv], dv[4 * lv + 3], dv[4 * lv + 1], dv[4 * lv + 2], 100.0 * (double)acc / (double)(tot ? tot : 1), tot);
            }
            c.arena.release(dbg_mark);
        }
        st->factors += h.factors;
        st->levels += h.levels_done;
        st->small_levels += h.levels_done;
        st->eager_levels += h.levels_done;
        ++eager_phases;
        st->eager_phases = eager_phases;
        if (level_log) fprintf(stderr, "eager phase %u: levels %u .. %u, %u processed, %llu factors, %zu heads at the start, %u blocks, status %u\n",
                               eager_phases, Lfrom, h.level + 1, h.levels_done, h.factors, tcount, h.nblk, h.status);
        return h.level;
    };
    auto t_prev = std::chrono::steady_clock::now();
    for (u32 L = maxlcp; L >= threshold; --L) {
        if (lcut && L == lcut) {
            // ---- all remaining levels window by window inside one launch; the global state stays untouched, so a failed
            //      pass (a window whose known range shrank into its interior) simply continues with the loop below and
            //      tries once more further down, where the borders of the known range move half as far
            u64 nf = 0;
            // per-level lists of the window pass: start with the large ones if a level is expected to hold more entries per window than
            // the small ones take (sampled histogram of the candidates' levels; most of a level's candidates are usually erased by then,
            // hence the factor -- a wrong guess only costs time, the pass retries with the other size)
            bool start_large = false;
            {
                u32 hh[64];
                c.read_n(d_lvlhist, hh, 64);
                u64 mx = 0;
                for (u32 v = threshold; v <= lcut && v < 64u; ++v) mx = std::max<u64>(mx, hh[v]);
                start_large = (double)mx * 16.0 * (double)window_levels_window() / (double)n > 4.0 * (double)window_levels_small_list();
            }
            const bool wsrc = !phi && c.window_src;               // (option window_src: the window kernel fetches the sources of its factors itself)
            const int why = factorize_window_levels(c, n, cur, prio, res8, phi, lcut, threshold, fs, &nf, start_large, wsrc ? sa : nullptr);
            const bool ok = why == 0;
            if (ok && wsrc) window_src_done = true;
            st->window_pass = ok ? 1 : 2;
            st->window_lcut = lcut;
            if (ok) { st->factors += nf; break; }
            // the lists of the levels L .. threshold were never materialised: every alive position becomes an entry of list cur[q]
            rebuild_from_cur(threshold - 1, L);
            eager_phases = 99;                                 // (the floor of the global loop moves: no eager phase any more)
            lcut = (why == 1 && L > 24 && threshold <= 24) ? 24 : 0;      // borders only: they move half as far from level 24 on
        }
        // ---- a run of small levels ahead: the eager formulation takes them inside one launch ---------------------------------------
        if (eager_possible && eager_phases < 4 && L > eager_floor + 256 && small_lazy_streak >= 4 && L != force_general_level) {
            const bool small_ahead = true;                     // (the eager lists hold run heads only: the size of the candidate segments does not matter)
            if (small_ahead) {
                const u32 next = run_eager(L);
                small_lazy_streak = 0;
                if (next >= eager_floor && next >= threshold) rebuild_from_cur(eager_floor - 1, next);     // it gave up on level `next`: the lazy loop goes on there
                else { for (u32 v = eager_floor; v <= L; ++v) pushed_into.drop(v); }
                L = next + 1;                                  // (the loop's --L lands on `next`; below the floor: the window pass or the end)
                continue;
            }
        }
        // ---- purge: after a run of large levels whose entries were (almost) all erased, drop the erased candidates of
        //      every level still to come (they can never come back to life: cur only decreases)
        if ((dead_streak >= 4 && levels_since_purge >= 16) || purge_pays) {
            purge_pays = false;
            size_t cnt = 0;                                // candidates of the levels <= L form a prefix of the sorted array
            for (u32 v = L;; --v) { if (h_segend[v] > h_segstart[v]) { cnt = h_segend[v]; break; } if (v == threshold) break; }
            if (cnt > 65536) {
                purge_class_kernel<<<cdiv(cnt, 256), 256, 0, s>>>(cvals[x], cnt, threshold, cur, cls);
                LAUNCH_CHECK();
                select_by_class(c, cls, 1, cnt, cvals[x], cvals[x ^ 1], nullptr, nullptr, d_cnt);
                select_by_class(c, cls, 1, cnt, ckeys[x], ckeys[x ^ 1], nullptr, nullptr, d_cnt);
                x ^= 1;
                cand = cvals[x];
                cand_count = c.read(d_cnt);
                HIP_TRY(hipMemsetAsync(d_segstart, 0, ((size_t)L + 1) * sizeof(u32), s));
                HIP_TRY(hipMemsetAsync(d_segend, 0, ((size_t)L + 1) * sizeof(u32), s));
                if (cand_count) {
                    seg_bounds_kernel<<<cdiv(cand_count, 256), 256, 0, s>>>(ckeys[x], cand_count, d_segstart, d_segend);
                    LAUNCH_CHECK();
                }
                c.read_n(d_segstart, h_segstart.data(), (size_t)L + 1);
                c.read_n(d_segend, h_segend.data(), (size_t)L + 1);
                st->purges++;
                HIP_TRY(hipMemsetAsync(d_lcount, 0xFF, nlev * sizeof(u32), s));      // new lists: counts unknown, nothing purged ahead
                purge_next = 0xFFFFFFFFu;
            }
            levels_since_purge = 0;
            dead_streak = 0;
        }
        ++levels_since_purge;
        const bool trigger_stale = nolive_run >= stale_trigger;     // a run of levels without a live entry (erased or stale entries only)
        if ((probe_dead && (pushed_into.get(L).empty() || dead_levels_run >= 8)) || trigger_stale) {
            // the last level was completely erased: look how far down that goes.  Range: the levels below L (never past the
            // window cut) holding at most 4 Mi candidates; levels with pushed entries end the range -- unless a long run of
            // erased levels has been seen, then their pool segments are examined as well (texts like Fibonacci words leave
            // erased pushed entries in hundreds of thousands of consecutive levels).  After a long run of levels without a live
            // entry the probe also returns the highest current value of an alive entry: no level above it can select
            // anything, their stale entries are pushed down in one step.
            const u32 floor_level = std::max<u32>(threshold, lcut ? lcut + 1 : 0);
            const bool deep = dead_levels_run >= 8 || trigger_stale;
            std::vector<ProbeSeg> ptab;
            u32 Lb = L;
            size_t lo = (size_t)-1, hi = 0, cnt = 0;
            for (u32 v = L;; --v) {
                const std::vector<PoolSeg>& pv = pushed_into.get(v);
                if (!pv.empty()) {
                    if (v != L && (!deep || ptab.size() + pv.size() > ((size_t)4 << 20))) break;
                    for (const PoolSeg& sg : pv) ptab.push_back(ProbeSeg{sg.off, sg.cnt, v});
                }
                if (h_segend[v] > h_segstart[v]) {
                    cnt += h_segend[v] - h_segstart[v];
                    if (cnt > ((size_t)4 << 20) && v != L) break;
                    lo = std::min(lo, (size_t)h_segstart[v]); hi = std::max(hi, (size_t)h_segend[v]);
                }
                Lb = v;
                if (v == floor_level) break;
            }
            if (Lb < L) {
                u32 alive = 0, livecur = 0;
                const size_t pmark = c.arena.mark();
                ProbeSeg* d_ptab = nullptr;
                if (hi > lo || !ptab.empty()) {
                    HIP_TRY(hipMemsetAsync(d_cnt + 2, 0, 2 * sizeof(u32), s));
                    u32* d_maxcur = trigger_stale ? d_cnt + 3 : nullptr;
                    if (hi > lo) {
                        unsigned g = cdiv(hi - lo, 256 * 4); if (g > 4096) g = 4096; if (g == 0) g = 1;
                        alive_max_level_kernel<<<g, 256, 0, s>>>(ckeys[x], cvals[x], lo, hi, cur, threshold, d_cnt + 2, d_maxcur);
                        LAUNCH_CHECK();
                    }
                    if (!ptab.empty()) {
                        d_ptab = (ProbeSeg*)c.arena.alloc(ptab.size() * sizeof(ProbeSeg));
                        HIP_TRY(hipMemcpyAsync(d_ptab, ptab.data(), ptab.size() * sizeof(ProbeSeg), hipMemcpyHostToDevice, s));
                        unsigned g = cdiv(ptab.size(), 256); if (g > 4096) g = 4096;
                        alive_max_pool_kernel<<<g, 256, 0, s>>>(pool, d_ptab, ptab.size(), cur, threshold, d_cnt + 2, d_maxcur);
                        LAUNCH_CHECK();
                    }
                    u32 two[2];
                    c.read_n(d_cnt + 2, two, 2);           // synchronises: the host table has been copied
                    alive = two[0]; livecur = two[1];
                }
                st->probes++;
                if (alive < Lb) { c.arena.release(pmark); L = Lb; probe_dead = true; continue; }          // [Lb, L] all erased: the loop's --L goes on below Lb
                if (alive < L) { c.arena.release(pmark); L = alive + 1; probe_dead = false; continue; }   // (alive, L] erased: --L lands on `alive`
                if (trigger_stale) {
                    // alive == L.  No alive entry of the range has a value above livecur, so the levels (floor, L] with
                    // floor = max(livecur, Lb - 1) only push: all of them at once, numbered as the loop would number them
                    const u32 floor_lv = std::max(livecur, Lb - 1);
                    const u32 span = L - floor_lv;
                    if (floor_lv < L && span >= 4) {
                        HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(u32), s));
                        if (hi > lo) {
                            unsigned g = cdiv(hi - lo, 256 * 4); if (g > 4096) g = 4096; if (g == 0) g = 1;
                            stale_collect_orig_kernel<<<g, 256, 0, s>>>(ckeys[x], cvals[x], lo, hi, floor_lv, L, cur, prio, threshold, skeys[0], svals[0], d_cnt);
                            LAUNCH_CHECK();
                        }
                        if (!ptab.empty()) {
                            stale_collect_pool_kernel<<<cdiv(ptab.size(), 256), 256, 0, s>>>(pool, d_ptab, ptab.size(), floor_lv, L, cur, prio, threshold,
                                                                                         skeys[0], svals[0], d_cnt);
                            LAUNCH_CHECK();
                        }
                        const u32 R = c.read(d_cnt);
                        if (R) {
                            if (pool_top + R > n || (u64)prio_base + R > 0xFFFFFFFFull)
                                throw HipError{hipErrorUnknown, "factorize: push pool overflow", (int)__LINE__};
                            const int ya = radix_sort_pairs_u64(c, skeys, svals, R, 0, 32 + (int)bits_for(span));     // order of the level-by-level loop
                            stale_target_keys_kernel<<<cdiv(R, 256), 256, 0, s>>>(svals[ya], R, cur, skeys[ya]);
                            LAUNCH_CHECK();
                            u64* kb[2] = { skeys[ya], skeys[ya ^ 1] };
                            u32* vb[2] = { svals[ya], svals[ya ^ 1] };
                            const int yb = radix_sort_pairs_u64(c, kb, vb, R, 32, 32 + (int)bits_for(floor_lv));        // stable: grouped by target
                            PushSeg* d_bsegs = (PushSeg*)c.arena.alloc((size_t)R * sizeof(PushSeg));
                            HIP_TRY(hipMemsetAsync(d_sc, 0, 8 * sizeof(u32), s));
                            push_finalize_kernel<<<cdiv(R, 256), 256, 0, s>>>(kb[yb], vb[yb], R, prio_base, prio, pool + pool_top, res8, d_bsegs, R, d_sc);
                            LAUNCH_CHECK();
                            const u32 nseg = c.read(&d_sc->nseg);
                            std::vector<PushSeg> bsegs(nseg);
                            c.read_n(d_bsegs, bsegs.data(), nseg);
                            std::sort(bsegs.begin(), bsegs.end(), [](const PushSeg& a, const PushSeg& b) { return a.start < b.start; });
                            for (u32 j = 0; j < nseg; ++j) {
                                const u32 end = (j + 1 < nseg) ? bsegs[j + 1].start : R;
                                const u32 tgt = bsegs[j].target;
                                if (tgt > floor_lv || tgt < threshold) throw HipError{hipErrorUnknown, "factorize: bad push target", (int)__LINE__};
                                pushed_into[tgt].push_back(PoolSeg{(u32)pool_top + bsegs[j].start, end - bsegs[j].start});
                            }
                            pool_top += R;
                            prio_base += R;
                            st->pushes += R;
                        }
                        for (const ProbeSeg& sg : ptab) if (sg.level > floor_lv) pushed_into.drop(sg.level);
                        c.arena.release(pmark);
                        st->levels += span;
                        stale_trigger = span >= 64 ? 8u : std::min<u32>(stale_trigger * 2, 4096u);
                        nolive_run = 0;
                        L = floor_lv + 1;                  // the loop's --L lands on floor_lv
                        probe_dead = false;
                        continue;
                    }
                    stale_trigger = std::min<u32>(stale_trigger * 2, 4096u);     // nothing to gain here: ask less often
                    nolive_run = 0;
                }
                c.arena.release(pmark);
            } else if (trigger_stale) { stale_trigger = std::min<u32>(stale_trigger * 2, 4096u); nolive_run = 0; }
            probe_dead = false;
        }
        const u32 m0 = h_segend[L] - h_segstart[L];
        u32 m1 = 0;
        const auto hp0 = std::chrono::steady_clock::now();
        const std::vector<PoolSeg>& segsL = pushed_into.get(L);
        u32 m1_total = 0;
        for (const PoolSeg& sg : segsL) m1_total += sg.cnt;
        bool gathered = false;
        auto gather_all = [&]() {   // the pushed part of the list: one kernel per (at most gtab_cap) pool segments
            size_t done = 0;
            while (done < segsL.size()) {
                const size_t cntseg = std::min(segsL.size() - done, (size_t)gtab_cap);
                u32 tot = 0;
                for (size_t j = 0; j < cntseg; ++j) { h_gtab[j] = GatherSeg{segsL[done + j].off, tot}; tot += segsL[done + j].cnt; }
                if (cntseg == 1) {
                    HIP_TRY(hipMemcpyAsync(pushed + m1, pool + h_gtab[0].src_off, (size_t)tot * sizeof(u32), hipMemcpyDeviceToDevice, s));
                } else {
                    // every level ends with a synchronising read-back, so the pinned table is free again when the next level
                    // fills it; only a second chunk inside the same level has to wait
                    if (done) HIP_TRY(hipStreamSynchronize(s));
                    HIP_TRY(hipMemcpyAsync(d_gtab, h_gtab, cntseg * sizeof(GatherSeg), hipMemcpyHostToDevice, s));
                    gather_segments_kernel<<<cdiv(tot, 256), 256, 0, s>>>(pool, d_gtab, (u32)cntseg, tot, pushed + m1);
                    LAUNCH_CHECK();
                }
                m1 += tot;
                done += cntseg;
            }
            gathered = true;
        };
        const u32 m = m0 + m1_total;
        if (m == 0) { pushed_into.drop(L); continue; }
        st->levels++;
        if (m <= (c.small_big ? SMALL_RAW_SLIM : SMALL_RAW) && L != force_general_level) {
            // ---- whole level in one workgroup: ONE launch (list read from the pool segments, result published into mapped
            //      host memory), falling back to the general path if more than SMALL_M entries are still alive.  While the kernel
            //      of a level runs, the kernel of the level below it is already queued ("speculative", see SmallCtl): the host's
            //      turnaround between two levels -- result, bookkeeping, launch -- no longer leaves the GPU idle.
            struct Flight { u32 L, m, m0, slot, zseq; bool zc, spec; int inst; };
            auto list_size = [&](u32 lv, u32* m0_out, size_t* nsegs) {       // entries of a level as far as the host knows them
                const u32 a0 = h_segend[lv] - h_segstart[lv];
                const std::vector<PoolSeg>& sv = pushed_into.get(lv);
                u64 t = a0;
                for (const PoolSeg& sg : sv) t += sg.cnt;
                *m0_out = a0; *nsegs = sv.size();
                return t;
            };
            u64 inflight_push_max = 0;                              // upper bound of what the levels in flight may still push
            const u32 raw_cap = c.small_big ? SMALL_RAW_SLIM : SMALL_RAW;
            auto launch_small = [&](u32 lv, u32 slot, bool spec, Flight* f) -> bool {
                u32 a0; size_t ns;
                const u64 mm = list_size(lv, &a0, &ns);
                const std::vector<PoolSeg>& sv = pushed_into.get(lv);
                const u32 half = gtab_cap / 2;
                if ((spec || lv != L) && (mm == 0 || mm > raw_cap || ns > SMALL_GATHER || ns > half || !d_hgtab)) return false;
                purge_ahead(lv);
                // the 512-thread instance where the list is long or the levels above had many survivors (a level that overflows the
                // small instance would be run twice)
                // instance: 0 = 256 threads, 1 = 512, 2 = 1 024 (SLIM); small_big 2 / 3: always the second / third (tests)
                int inst = 0;
                if (c.small_big == 2) inst = 1;
                else if (c.small_big == 3) inst = 2;
                else if (c.small_big) {
                    // The 1 024-thread instance is the fastest for every level (fewer dependent rounds in each step: 37 against 49 us
                    // for a level with 300 survivors) but cannot order a crowded target or targets above its table: where it gave
                    // up recently, or the level lies above the table, the instance is chosen by the survivors of the level above.
                    if (lv <= (1u << 13) && slim_penalty == 0) inst = 2;
                    else if (last_alive > SMALL_M * 3 / 4 || mm > SMALL_RAW) inst = 1;
                    if (force_inst >= 0 && !spec) { inst = force_inst; force_inst = -1; }       // (a level that is run again)
                }
                const u32 m_inst = inst == 2 ? SMALL_M_SLIM : (inst ? SMALL_M_BIG : SMALL_M);
                const u32 push_max = (u32)std::min<u64>(mm, m_inst);          // at most one push per surviving entry
                if (pool_top + inflight_push_max + push_max > n || (u64)prio_base + inflight_push_max + push_max > 0xFFFFFFFFull) {
                    if (spec) return false;
                    throw HipError{hipErrorUnknown, "factorize: push pool overflow", (int)__LINE__};
                }
                u32 gn = 0;
                const u32* pushed_src = pushed;
                if (ns <= SMALL_GATHER && ns <= half && d_hgtab) {
                    // (each of the two levels in flight has its own half of the table)
                    GatherSeg* ht = h_gtab + (size_t)slot * half;
                    u32 tot = 0;
                    for (size_t j2 = 0; j2 < ns; ++j2) { ht[j2] = GatherSeg{sv[j2].off, tot}; tot += sv[j2].cnt; }
                    gn = (u32)ns;
                } else gather_all();                                 // (never for a speculative launch)
                u32* zdst = nullptr; u32* zflag = nullptr; u32 zseq = 0;
                const bool zc = c.publish_begin(&zdst, &zflag, &zseq, 1 + slot);
                if (spec && !zc) return false;
                {
                    Ctx::ProfScope prof(c, K_SMALL_LEVEL, mm * 16);
                    u32* zsegs = zc ? c.zc_dev + (size_t)(1 + slot) * Ctx::ZC_WORDS + Ctx::ZC_SEG_OFF : nullptr;
                    if (inst == 2)
                        small_level_kernel<1024, true><<<1, 1024, 0, s>>>(cand + h_segstart[lv], a0, pushed_src, (u32)mm, pool, d_hgtab ? d_hgtab + (size_t)slot * half : nullptr, gn, lv,
                                                                  threshold, n, cur, prio, phi_eff, fs.flen, res8, fs.fsrc, (u32)pool_top, prio_base, d_segs2[slot], seg_cap, live,
                                                                  /*inline_budget=*/1u << 17, d_sc2[slot], zdst, zflag, zseq, d_sprof, zsegs,
                                                                  d_ctl, spec ? 1u : 0u, d_sc2[slot ^ 1], d_segs2[slot ^ 1], d_lcount + lv, fs.flen8);
                    else if (inst == 1)
                        small_level_kernel<512><<<1, 512, 0, s>>>(cand + h_segstart[lv], a0, pushed_src, (u32)mm, pool, d_hgtab ? d_hgtab + (size_t)slot * half : nullptr, gn, lv,
                                                                  threshold, n, cur, prio, phi_eff, fs.flen, res8, fs.fsrc, (u32)pool_top, prio_base, d_segs2[slot], seg_cap, live,
                                                                  /*inline_budget=*/1u << 17, d_sc2[slot], zdst, zflag, zseq, d_sprof, zsegs,
                                                                  d_ctl, spec ? 1u : 0u, d_sc2[slot ^ 1], d_segs2[slot ^ 1], d_lcount + lv, fs.flen8);
                    else
                        small_level_kernel<256><<<1, 256, 0, s>>>(cand + h_segstart[lv], a0, pushed_src, (u32)mm, pool, d_hgtab ? d_hgtab + (size_t)slot * half : nullptr, gn, lv,
                                                                  threshold, n, cur, prio, phi_eff, fs.flen, res8, fs.fsrc, (u32)pool_top, prio_base, d_segs2[slot], seg_cap, live,
                                                                  /*inline_budget=*/1u << 17, d_sc2[slot], zdst, zflag, zseq, d_sprof, zsegs,
                                                                  d_ctl, spec ? 1u : 0u, d_sc2[slot ^ 1], d_segs2[slot ^ 1], d_lcount + lv, fs.flen8);
                    LAUNCH_CHECK();
                }
                inflight_push_max += push_max;
                *f = Flight{lv, (u32)mm, a0, slot, zseq, zc, spec, inst};
                return true;
            };
            auto wait_small = [&](const Flight& f) {
                if (f.zc) c.publish_wait(f.zseq, &h_sc, sizeof(LevelScalars), 1 + f.slot);
                else c.read_n((const u32*)d_sc2[f.slot], (u32*)&h_sc, sizeof(LevelScalars) / sizeof(u32));
            };
            Flight cur_f, next_f;
            const auto hp1 = std::chrono::steady_clock::now();
            launch_small(L, 0, false, &cur_f);
            const auto hp2 = std::chrono::steady_clock::now();
            host_prof[0] += std::chrono::duration<double, std::micro>(hp1 - hp0).count();
            host_prof[1] += std::chrono::duration<double, std::micro>(hp2 - hp1).count();
            bool general_path = false, stop_chain = false;
            for (;;) {
                // queue the level below behind it, unless something the host has to decide first may be due there
                bool has_next = false;
                if (c.small_pipeline && !stop_chain && cur_f.L > threshold && cur_f.L - 1 > lcut && !purge_pays && dead_streak < 4 &&
                    nolive_run + 2 < stale_trigger)
                    has_next = launch_small(cur_f.L - 1, cur_f.slot ^ 1, true, &next_f);
                const auto hw0 = std::chrono::steady_clock::now();
                wait_small(cur_f);
                const auto hw1 = std::chrono::steady_clock::now();
                host_prof[2] += std::chrono::duration<double, std::micro>(hw1 - hw0).count();
                host_prof[4] += 1;
                struct PostTimer { double* acc; std::chrono::steady_clock::time_point t0; ~PostTimer() { *acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } } post_timer{&host_prof[3], hw1};
                const u32 LL = cur_f.L;
                bool redo = false;
                if (h_sc.pad[1] == 3) redo = true;                   // the speculation failed: the level has not been touched
                else if (h_sc.pad[1] && c.small_big == 1) {                // gave up (1: too many survivors, 2: cannot order these pushes): another instance?
                    int nxt = -1;
                    if (h_sc.pad[1] == 2) {                                  // the 1 024-thread instance: a crowded target, or targets above its table
                        slim_penalty = 8;                                    // (not again for a few levels)
                        if (h_sc.nstale <= SMALL_M_BIG) nxt = h_sc.nstale > SMALL_M ? 1 : 0;     // (nstale: the survivors, published with the code)
                    } else if (cur_f.inst == 0) nxt = 1;
                    else if (cur_f.inst == 1 && cur_f.L <= (1u << 13) && slim_penalty == 0) nxt = 2;
                    if (nxt >= 0) { force_inst = nxt; redo = true; } else general_path = true;
                }
                else if (h_sc.pad[1]) general_path = true;           // too many survivors: the multi-launch path takes the level
                else {
                    last_alive = h_sc.nlive + h_sc.nstale;
                    if (slim_penalty) --slim_penalty;
                    if (h_sc.pad[0]) {                             // many long factors: the kills are spread over the whole chip
                        apply_list_kernel<<<cdiv((size_t)h_sc.selected * 64, 256), 256, 0, s>>>(live, &d_sc2[cur_f.slot]->selected, LL, n, phi_eff, cur, fs.flen, fs.fsrc, res8, fs.flen8);
                        LAUNCH_CHECK();
                    }
                    pushed_into.drop(LL);
                    st->small_levels++;
                    if (level_log) fprintf(stderr, "small %u m %u m0 %u live %u stale %u npush %u big %d\n", LL, cur_f.m, cur_f.m0, h_sc.nlive, h_sc.nstale, h_sc.npush, cur_f.inst);
                    if (LL != L) { st->levels++; ++levels_since_purge; }   // (the level the outer loop stands on has been counted)
                    if (h_sc.nlive == 0) ++nolive_run; else nolive_run = 0;
                    if (h_sc.nlive <= 1024) ++small_lazy_streak; else small_lazy_streak = 0;
                    if (h_sc.nlive == 0 && h_sc.nstale == 0) { probe_dead = true; ++dead_levels_run; stop_chain = true; }   // small levels do not count for the purge heuristic
                    else {
                        probe_dead = false;
                        dead_levels_run = 0;
                        st->factors += h_sc.selected;
                        const u32 npush = h_sc.npush, nseg = h_sc.nseg;
                        if (npush) {
                            if (nseg > seg_cap) throw HipError{hipErrorUnknown, "factorize: too many push targets in one level", (int)__LINE__};
                            if (nseg <= SEG_INLINE) std::copy(h_sc.segs, h_sc.segs + nseg, h_segs.begin());
                            else if (cur_f.zc && nseg <= SMALL_M_SLIM)
                                memcpy(h_segs.data(), c.zc_host + (size_t)(1 + cur_f.slot) * Ctx::ZC_WORDS + Ctx::ZC_SEG_OFF, (size_t)nseg * sizeof(PushSeg));   // published next to the scalars
                            else c.read_n(d_segs2[cur_f.slot], h_segs.data(), nseg);
                            for (u32 j2 = 0; j2 < nseg; ++j2) {       // written in order of `start` by one thread
                                const u32 end = (j2 + 1 < nseg) ? h_segs[j2 + 1].start : npush;
                                const u32 tgt = h_segs[j2].target;
                                if (tgt >= LL || tgt < threshold) throw HipError{hipErrorUnknown, "factorize: bad push target", (int)__LINE__};
                                pushed_into[tgt].push_back(PoolSeg{(u32)pool_top + h_segs[j2].start, end - h_segs[j2].start});
                            }
                            pool_top += npush;
                            prio_base += npush;
                            st->pushes += npush;
                        }
                    }
                    if (nolive_run + 2 >= stale_trigger) stop_chain = true;
                }
                inflight_push_max = 0;
                if (general_path) {
                    if (has_next) wait_small(next_f);               // (it gave up as well: the level above it did)
                    force_general_level = LL;
                    if (LL != L) { L = LL + 1; general_path = false; }   // the outer loop comes back to this level and takes the multi-launch path
                    break;
                }
                if (redo) {                                          // run the level again with the list the host knows now
                    if (has_next) wait_small(next_f);               // (gave up as well)
                    u32 a0; size_t ns;
                    const u64 mm = list_size(LL, &a0, &ns);
                    if (mm == 0 || mm > raw_cap || !launch_small(LL, 0, false, &cur_f)) { L = LL + 1; break; }   // (too large now: the outer loop decides)
                    continue;
                }
                if (!has_next) { L = LL; break; }                    // the outer loop goes on below this level
                // the level below is in flight (or already done): it is the current one now
                if (h_sc.pad[0]) {
                    // this level's factors were applied by a separate launch behind the queued kernel, which gave up for that reason
                    wait_small(next_f);
                    u32 a0; size_t ns;
                    const u64 mm = list_size(LL - 1, &a0, &ns);
                    if (mm == 0 || mm > raw_cap || !launch_small(LL - 1, 0, false, &cur_f)) { L = LL; break; }
                    continue;
                }
                inflight_push_max = std::min<u64>(next_f.m, next_f.inst == 2 ? SMALL_M_SLIM : (next_f.inst ? SMALL_M_BIG : SMALL_M));
                cur_f = next_f;
            }
            if (!general_path) continue;
            // (general path for level L: its list is re-read below)
        }
        purge_ahead(L);
        if (slim_penalty) --slim_penalty;                   // (a level on the multi-launch path counts as well)
        if (!gathered) gather_all();
        pushed_into.drop(L);
        HIP_TRY(hipMemsetAsync(d_sc, 0, 8 * sizeof(u32), s));
        const unsigned gm = cdiv(m, 256);
        const bool mid = m <= (1u << 20);                    // few enough entries: unordered lists, fewer launches
        if (mid) {
            Ctx::ProfScope prof(c, K_LEVEL_INIT, (u64)m * 12);
            classify_append_kernel<<<gm, 256, 0, s>>>(cand + h_segstart[L], m0, pushed, m, L, threshold, cur, live, stale, bm, d_sc);
            LAUNCH_CHECK();
        } else {
            {   // per entry: list (4) + cur (4) + ent (4) + class byte (1)
                Ctx::ProfScope prof(c, K_LEVEL_INIT, (u64)m * 13);
                classify_kernel<<<gm, 256, 0, s>>>(cand + h_segstart[L], m0, pushed, m, L, threshold, cur, ent, cls, bm);
                LAUNCH_CHECK();
            }
            select_by_class(c, cls, CL_LIVE, m, ent, live, nullptr, nullptr, &d_sc->nlive);
            select_by_class(c, cls, CL_STALE, m, ent, stale, nullptr, nullptr, &d_sc->nstale);
        }
        c.read_n((const u32*)d_sc, (u32*)&h_sc, 8);
        const u32 nl = h_sc.nlive, ns = h_sc.nstale;
        if (((u64)nl + ns) * 16 < m) ++dead_streak; else dead_streak = 0;     // (almost) all entries already erased
        // Texts with long repeats: every level holds one candidate per PLCP ramp, nearly all of them erased by the ramp's first
        // factor, and there are thousands of such levels.  Dropping the erased candidates of all remaining levels costs one pass
        // over them (~0.1 ns per candidate); it pays when it turns enough of the levels to come into one-workgroup levels
        // (~0.15 ms saved per level).
        if (m > SMALL_RAW && ((u64)nl + ns) * 4 < m && levels_since_purge >= 16) {
            const u32 floor_lv = std::max<u32>(lcut, threshold);
            size_t cnt = 0;
            for (u32 v = L;; --v) { if (h_segend[v] > h_segstart[v]) { cnt = h_segend[v]; break; } if (v == threshold) break; }
            if (L > floor_lv && (u64)(L - floor_lv) * 1500000ull > cnt) purge_pays = true;
        }
        if (nl == 0) ++nolive_run; else nolive_run = 0;
        if (nl <= 1024) ++small_lazy_streak; else small_lazy_streak = 0;
        if (nl == 0 && ns == 0) { probe_dead = true; ++dead_levels_run; continue; }   // every entry already erased (:86)
        dead_levels_run = 0;
        const bool wide = (L > 24);
        const unsigned gl = wide ? cdiv((size_t)nl * 64, 256) : cdiv(nl, 256);
        const unsigned gs = wide ? cdiv((size_t)ns * 64, 256) : cdiv(ns, 256);
        if (nl) {
            u32 undecided = nl;
            while (undecided) {
                HIP_TRY(hipMemsetAsync(&d_sc->undecided, 0, sizeof(u32), s));
                {   // per live entry: list + its bitmap word (12); per undecided entry: prio (4) + the window's bitmap words
                    Ctx::ProfScope prof(c, K_MIS_ROUND, (u64)nl * 12 + (u64)undecided * (4 + 8ull * ((2 * L - 2) / 32 + 2)));
                    if (wide) mis_round_kernel<64><<<gl, 256, 0, s>>>(live, nl, L, n, prio, bm, d_sc);
                    else      mis_round_kernel<1><<<gl, 256, 0, s>>>(live, nl, L, n, prio, bm, d_sc);
                    LAUNCH_CHECK();
                }
                const u32 now = c.read(&d_sc->undecided);
                st->rounds++;
                // the undecided entry of highest priority can always decide: no progress means a bug
                if (now >= undecided) throw HipError{hipErrorUnknown, "factorize: selection rounds made no progress", (int)__LINE__};
                undecided = now;
            }
        }
        {   // per entry: list, cur, prio (12) + the window's bitmap words + outputs (13)
            Ctx::ProfScope prof(c, K_RESOLVE, (u64)(ns + nl) * (25 + 8ull * ((2 * L - 2) / 32 + 2)));
            u32* app = mid ? &d_sc->npush : nullptr;         // mid: push records appended straight into the sort input
            u64* rk = mid ? skeys[0] : rkey;
            u32* rv = mid ? svals[0] : rval;
            if (ns) {
                if (wide) resolve_kernel<64><<<gs, 256, 0, s>>>(stale, ns, false, L, threshold, n, prio, bm, cur, rk, rv, rc, app);
                else      resolve_kernel<1><<<gs, 256, 0, s>>>(stale, ns, false, L, threshold, n, prio, bm, cur, rk, rv, rc, app);
                LAUNCH_CHECK();
            }
            if (nl) {
                const size_t o = mid ? 0 : ns;
                if (wide) resolve_kernel<64><<<gl, 256, 0, s>>>(live, nl, true, L, threshold, n, prio, bm, cur, rk + o, rv + o, rc + o, app);
                else      resolve_kernel<1><<<gl, 256, 0, s>>>(live, nl, true, L, threshold, n, prio, bm, cur, rk + o, rv + o, rc + o, app);
                LAUNCH_CHECK();
            }
        }
        if (!mid) select_by_class(c, rc, 1, (size_t)ns + nl, rval, svals[0], rkey, skeys[0], &d_sc->npush);
        if (nl) {
            // per entry: list + state (5); per factor: Phi, flen, fsrc (12) + L kills (8 B each) + L truncations (4 B each)
            Ctx::ProfScope prof(c, K_APPLY, (u64)nl * 5 + (u64)nl * (12 + 12ull * L));
            if (wide) apply_kernel<64><<<gl, 256, 0, s>>>(live, nl, L, n, bm, phi_eff, cur, fs.flen, fs.fsrc, d_sc, res8, fs.flen8);
            else      apply_kernel<1><<<gl, 256, 0, s>>>(live, nl, L, n, bm, phi_eff, cur, fs.flen, fs.fsrc, d_sc, res8, fs.flen8);
            LAUNCH_CHECK();
        }
        c.read_n((const u32*)d_sc, (u32*)&h_sc, 8);
        st->factors += h_sc.selected;
        const u32 npush = h_sc.npush;
        const u32 rounds_before = st->rounds;
        (void)rounds_before;
        if (npush) {
            // every push is caused by a truncation of a position in front of a factor, and factors are disjoint,
            // so the pool never needs more than n slots; checked before anything is written
            if (pool_top + npush > n || (u64)prio_base + npush > 0xFFFFFFFFull)
                throw HipError{hipErrorUnknown, "factorize: push pool overflow", (int)__LINE__};
            const int y = sort_pairs_u64_distinct(c, skeys, svals, npush, 0, 32 + (int)bits_for(L));   // priorities are distinct
            {
                Ctx::ProfScope prof(c, K_PUSH, (u64)npush * 24);
                push_finalize_kernel<<<cdiv(npush, 256), 256, 0, s>>>(skeys[y], svals[y], npush, prio_base, prio,
                                                                      pool + pool_top, res8, d_segs, seg_cap, d_sc);
                LAUNCH_CHECK();
            }
            c.read_n((const u32*)d_sc, (u32*)&h_sc, sizeof(LevelScalars) / sizeof(u32));   // scalars + inline segments: one sync
            const u32 nseg = h_sc.nseg;
            if (nseg > seg_cap) throw HipError{hipErrorUnknown, "factorize: too many push targets in one level", (int)__LINE__};
            if (nseg <= SEG_INLINE) std::copy(h_sc.segs, h_sc.segs + nseg, h_segs.begin());
            else c.read_n(d_segs, h_segs.data(), nseg);
            std::sort(h_segs.begin(), h_segs.begin() + nseg, [](const PushSeg& a, const PushSeg& b) { return a.start < b.start; });
            for (u32 j = 0; j < nseg; ++j) {
                const u32 end = (j + 1 < nseg) ? h_segs[j + 1].start : npush;
                const u32 tgt = h_segs[j].target;
                if (tgt >= L || tgt < threshold) throw HipError{hipErrorUnknown, "factorize: bad push target", (int)__LINE__};
                pushed_into[tgt].push_back(PoolSeg{(u32)pool_top + h_segs[j].start, end - h_segs[j].start});
            }
            pool_top += npush;
            prio_base += npush;
            st->pushes += npush;
        }
        if (level_log) {
            const auto t_now = std::chrono::steady_clock::now();
            fprintf(stderr, "level %u m0 %u m1 %u live %u stale %u selected %u npush %u ms %.3f\n", L, m0, m1, nl, ns, h_sc.selected, npush,
                    std::chrono::duration<double, std::milli>(t_now - t_prev).count());
            t_prev = t_now;
        }
        if (L == 0) break;
    }
    if (d_sprof) {
        unsigned long long hh[32];
        HIP_TRY(hipMemcpyAsync(hh, d_sprof, sizeof(hh), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (host_prof[4] > 0)
            fprintf(stderr, "small levels, host side, us per level: prepare %.1f launch %.1f wait %.1f bookkeeping %.1f\n", host_prof[0] / host_prof[4],
                    host_prof[1] / host_prof[4], host_prof[2] / host_prof[4], host_prof[3] / host_prof[4]);
        for (int part = 0; part < 2; ++part) {
            const unsigned long long* h = hh + 16 * part;
            const double lv = h[8] ? (double)h[8] : 1.0;
            fprintf(stderr, "small levels with %s survivors: %llu; per level: raw %.0f alive %.1f selected %.2f pushes %.1f; us (100 MHz clock): load %.1f sort %.1f "
                            "classify %.1f select %.1f push %.1f apply %.1f publish %.1f\n", part ? "> 1024" : "<= 1024", h[8], h[9] / lv, h[10] / lv, h[11] / lv,
                    h[12] / lv, h[0] / 100.0 / lv, h[1] / 100.0 / lv, h[2] / 100.0 / lv, h[3] / 100.0 / lv, h[4] / 100.0 / lv, h[5] / 100.0 / lv, h[6] / 100.0 / lv);
        }
    }
    c.arena.release(mark);
    // (window_src_done: every factor start has its source in fsrc[] -- the candidates of the global levels since cand_class_kernel, pushed
    //  entries since their push, the window pass' natural entries since the window kernel)
    build_owner(c, n, fs); if (!phi && !window_src_done) { fs.src_prio = prio; fs.src_sa = sa; fs.src_n = n; }
}

// ============================================================================================================
// lcpcomp::MaxLCPStrategy (compressors/lcpcomp/compress/MaxLCPStrategy.hpp:36-100) over MaxLCPSuffixList
// (compressors/lcpcomp/MaxLCPSuffixList.hpp), in position space.
//
// The reference's list is one stack per LCP level: insert() puts an entry in FRONT of its level (:86-124) and the head of
// the highest level is taken (:62-64).  Keys are decreased AT ONCE, when the truncating factor is selected (:82-94), and
// only into lower levels -- so when level L is reached its stack is final: the entries that were truncated to L, most
// recently truncated first, followed by the original entries by DESCENDING suffix-array index (the ctor inserts them in
// ascending order, :74-78).  A factor truncates at most one entry to L (the one L positions in front of it), so "most
// recently" is the selection time of that factor.  With
//     prio[p] = 2^31 - 1 - t          for an entry whose last truncation came from the t-th selected factor
//     prio[p] = 2^31 + (n - 1 - ISA[p])   for an entry that was never truncated
// level L is the set {p : cur[p] == L} and the selected entries are its lexicographically first maximal independent set
// under prio (conflict: text distance < L) -- the same selection kernel as ArraysComp.  Selected factors are numbered in
// prio order (one small sort per level); kills and truncations commute; the entry that ends a level with
// cur[s] == distance to a factor selected in this level, and was lowered in this level (stamp), is that factor's one
// push.  Model: tests/models/position_space.py::max_lcp_position_space (equal to the oracle incl. emission order).
// ============================================================================================================
__global__ void mlcp_init_prio_kernel(u32* __restrict__ prio, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) prio[p] = 0x80000000u + (u32)(n - 1 - prio[p]);
}
__global__ void mlcp_classify_kernel(const u32* __restrict__ orig, u32 m0, const u32* __restrict__ pushed, u32 m, u32 L,
                                     const u32* __restrict__ cur, u32* __restrict__ live, u64* __restrict__ bm, LevelScalars* __restrict__ sc) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    u32 p = 0;
    bool is_live = false;
    if (k < m) { p = (k < m0) ? orig[k] : pushed[k - m0]; is_live = cur[p] == L; }
    const u32 il = wave_append(is_live, &sc->nlive);
    if (is_live) { live[il] = p; atomicOr((unsigned long long*)&bm[p >> 5], 1ull << (2 * (p & 31))); }
}
// selected entries -> (prio, position) records; the bitmap is left all-zero for the next level
__global__ void mlcp_collect_kernel(const u32* __restrict__ live, u32 nl, const u32* __restrict__ prio, u64* bm,
                                    u64* __restrict__ skey, u32* __restrict__ sval, LevelScalars* __restrict__ sc) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 p = 0;
    bool sel = false;
    if (i < nl) { p = live[i]; sel = bm_state(bm, p) == 2u; }
    const u32 o = wave_append(sel, &sc->selected);
    if (sel) {
        skey[o] = prio[p]; sval[o] = p;
        atomicAnd((unsigned long long*)&bm[p >> 5], ~(3ull << (2 * (p & 31))));
    }
}
// factors of the level (sel[] in selection order): emit, kill the covered positions (:74-79), truncate the ones in front
// (:82-94; the nearest start wins); stamp[s] = L marks the positions that were lowered in this level
template <int G>
__global__ __launch_bounds__(256) void mlcp_apply_kernel(const u32* __restrict__ sel, u32 nsel, u32 L, size_t n, const u32* __restrict__ phi,
                                                          u32* __restrict__ cur, u32* __restrict__ flen, u32* __restrict__ fsrc,
                                                          u32* __restrict__ stamp) {
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    if (i >= nsel) return;
    const u32 p = sel[i];
    if (sub == 0) { flen[p] = L; fsrc[p] = phi[p]; }
    for (u32 j = sub; j < L && (size_t)p + j < n; j += G) cur[p + j] = 0;
    const u32 aff = (L < p) ? L : p;
    for (u32 j = sub; j < aff; j += G) {
        u32* q = &cur[p - 1 - j];
        if (*q > j + 1) { if (atomicMin(q, j + 1) > j + 1) stamp[p - 1 - j] = L; }
    }
}
// decrease_key events (:88-90): record (new level << 32 | position), new priority = selection time of the factor
template <int G>
__global__ __launch_bounds__(256) void mlcp_push_kernel(const u32* __restrict__ sel, u32 nsel, u32 L, u32 threshold, u32 tbase,
                                                         const u32* __restrict__ cur, const u32* __restrict__ stamp, u32* __restrict__ prio,
                                                         u64* __restrict__ bkey, LevelScalars* __restrict__ sc) {
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = (G == 1) ? 0u : (threadIdx.x & (G - 1));
    if (i >= nsel) return;
    const u32 p = sel[i];
    const u32 aff = (L < p) ? L : p;
    for (u32 j = sub; j < aff; j += G) {
        const u32 sp = p - 1 - j, d = j + 1;
        if (d >= threshold && stamp[sp] == L && cur[sp] == d) {
            prio[sp] = 0x7FFFFFFFu - (tbase + i);
            bkey[atomicAdd(&sc->npush, 1u)] = ((u64)d << 32) | sp;
        }
    }
}
__global__ void mlcp_pool_kernel(const u64* __restrict__ keys, u32 npush, u32* __restrict__ pool, PushSeg* __restrict__ segs, u32 seg_cap,
                                 LevelScalars* __restrict__ sc) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npush) return;
    const u32 tgt = (u32)(keys[i] >> 32);
    pool[i] = (u32)keys[i];
    if (i == 0 || (u32)(keys[i - 1] >> 32) != tgt) {
        const u32 j = atomicAdd(&sc->nseg, 1u);
        if (j < SEG_INLINE) sc->segs[j] = PushSeg{tgt, i};
        if (j < seg_cap) segs[j] = PushSeg{tgt, i};
    }
}

// MaxLCPStrategy keeps an entry in the list of its CURRENT value (eager decreases), so older copies in higher lists and in the
// candidate array are dead weight: highest level that holds an entry whose value still equals the level of its list
__global__ __launch_bounds__(256) void live_max_level_kernel(const u32* __restrict__ levels, const u32* __restrict__ pos, size_t lo, size_t hi,
                                                              const u32* __restrict__ cur, u32* __restrict__ d_max) {
    u32 best = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < hi; k += stride) {
        const u32 lv = levels[k];
        if (lv > best && cur[pos[k]] == lv) best = lv;
    }
    best = wave_reduce_max(best);
    if (lane_id() == 0 && best) atomicMax(d_max, best);
}
__global__ __launch_bounds__(256) void live_max_pool_kernel(const u32* __restrict__ pool, const ProbeSeg* __restrict__ tab, size_t nseg,
                                                             const u32* __restrict__ cur, u32* __restrict__ d_max) {
    u32 best = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < nseg; k += stride) {
        const ProbeSeg sg = tab[k];
        if (sg.level <= best) continue;
        for (u32 j = 0; j < sg.cnt; ++j) if (cur[pool[sg.off + j]] == sg.level) { best = sg.level; break; }
    }
    best = wave_reduce_max(best);
    if (lane_id() == 0 && best) atomicMax(d_max, best);
}

// ---- lcpcomp::MaxHeapStrategy (compressors/lcpcomp/compress/MaxHeapStrategy.hpp:36-101 over ds/ArrayMaxHeap.hpp:14-241) ------
// The tie order of this strategy is the LAYOUT of a binary heap after every insert / remove / decrease_key -- remove() moves the last
// element into the hole and only ever sifts it down, perlocate_down() compares an element index with a heap position -- i.e. a
// function of the whole operation history, one operation at a time.  There is no order-independent description a data-parallel
// selection could evaluate, so this is a PARITY row, not a tuned one: ONE thread replays the reference's loop on the device
// arrays (heap and back mapping in global memory; every step is a chain of dependent loads: ~1 minute per MiB of English text).
__device__ __forceinline__ void amh_put(u32* heap, u32* hpos, u32 p, u32 i) { heap[p] = i; hpos[i] = p; }
__device__ void amh_perlocate_down(const u32* key, u32* heap, u32* hpos, u32 size, u32 p, u32 k) {
    const u32 kk = key[k];
    int dir;
    do {
        const u32 lc = 2 * p + 1, rc = 2 * p + 2;
        const u32 kl = (lc < size) ? key[heap[lc]] : 0u;
        const u32 kr = (rc < size) ? key[heap[rc]] : 0u;
        if (kk < kl && kk < kr) dir = (kl > kr) ? 1 : 2;
        else if (kk < kl) dir = 1;
        else if (kk < kr) dir = 2;
        else if (kk == kl && kk == kr) dir = (lc < rc) ? 1 : 2;
        else if (kk == kl && k > lc) dir = 1;                  // element index against heap position, as the reference writes it
        else if (kk == kr && k > rc) dir = 2;
        else dir = 0;
        if (dir == 1) { amh_put(heap, hpos, p, heap[lc]); p = lc; }
        else if (dir == 2) { amh_put(heap, hpos, p, heap[rc]); p = rc; }
    } while (dir != 0);
    amh_put(heap, hpos, p, k);
}
__global__ void max_heap_strategy_kernel(const u32* __restrict__ sa, const u32* __restrict__ isa, u32* lcp, u32 n, u32 threshold,
                                         u32* heap, u32* hpos, u32* __restrict__ flen, u32* __restrict__ fsrc, unsigned long long* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    u32 heap_size = 0;
    for (u32 i = 1; i < n; ++i) if (lcp[i] >= threshold) ++heap_size;                 // :52-55
    const u32 undef = heap_size;
    u32 size = 0;
    for (u32 i = 0; i < n; ++i) hpos[i] = undef;
    for (u32 i = 1; i < n; ++i) {                                                      // :58-61, ArrayMaxHeap::insert :61-78
        if (lcp[i] < threshold) continue;
        u32 p = size++;
        const u32 ki = lcp[i];
        while (p > 0 && ki > lcp[heap[(p - 1) / 2]]) { amh_put(heap, hpos, p, heap[(p - 1) / 2]); p = (p - 1) / 2; }
        amh_put(heap, hpos, p, i);
    }
    unsigned long long z = 0;
    while (size > 0) {                                                                 // :68-97
        const u32 m = heap[0];
        const u32 fpos = sa[m], fs_ = sa[m - 1], fl = lcp[m];
        flen[fpos] = fl; fsrc[fpos] = fs_; ++z;
        for (u32 k = 0; k < fl; ++k) {                                                 // remove overlapped entries :79-81
            const u32 i = isa[fpos + k];
            const u32 p = hpos[i];
            if (p != undef) { const u32 last = heap[--size]; amh_perlocate_down(lcp, heap, hpos, size, p, last); hpos[i] = undef; }
        }
        for (u32 k = 0; k < fl && fpos > k; ++k) {                                     // correct intersecting entries :84-96
            const u32 sp = fpos - k - 1;
            const u32 i = isa[sp];
            if (hpos[i] != undef && (u64)sp + lcp[i] > fpos) {
                const u32 nl = fpos - sp;
                if (nl >= threshold) { lcp[i] = nl; amh_perlocate_down(lcp, heap, hpos, size, hpos[i], i); }
                else { const u32 p = hpos[i]; const u32 last = heap[--size]; amh_perlocate_down(lcp, heap, hpos, size, p, last); hpos[i] = undef; }
            }
        }
    }
    out[0] = z; out[1] = heap_size;
}
__global__ void fill_zero_u32_kernel(u32* p, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 0; }

// sa / isa: the suffix array and its inverse (isa is only read here), plcp in position space; lcp = PLCP[SA[.]] is built into scratch
void factorize_max_heap(Ctx& c, size_t n, const u32* sa, const u32* isa, const u32* plcp, u32 maxlcp, u32 threshold, FactorSpace& fs,
                        FactorizeStats* st) {
    FactorizeStats local;
    if (!st) st = &local;
    *st = FactorizeStats();
    st->maxlcp = maxlcp;
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    fill_zero_u32_kernel<<<cdiv(n, 256), 256, 0, s>>>(fs.flen, n);
    LAUNCH_CHECK();
    if (maxlcp >= threshold && threshold > 0) {
        u32* lcp = c.arena.get<u32>(n);
        u32* heap = c.arena.get<u32>(n);
        u32* hpos = c.arena.get<u32>(n);
        unsigned long long* d_out = (unsigned long long*)c.arena.alloc(2 * sizeof(unsigned long long));
        build_lcp(c, sa, plcp, n, lcp);
        max_heap_strategy_kernel<<<1, 64, 0, s>>>(sa, isa, lcp, (u32)n, threshold, heap, hpos, fs.flen, fs.fsrc, d_out);
        LAUNCH_CHECK();
        u32 w[4];
        c.read_n((const u32*)d_out, w, 4);
        st->factors = (u64)w[0] | ((u64)w[1] << 32);
        st->entries = (u64)w[2] | ((u64)w[3] << 32);
    }
    c.arena.release(mark);
    build_owner(c, n, fs);
}

void factorize_max_lcp(Ctx& c, size_t n, u32* isa, const u32* phi, u32* plcp, u32 maxlcp, u32 threshold, FactorSpace& fs,
                       FactorizeStats* st) {
    FactorizeStats local;
    if (!st) st = &local;
    *st = FactorizeStats();
    st->maxlcp = maxlcp;
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    const unsigned gn = cdiv(n, 256);
    u32* cur = plcp;
    u32* prio = isa;

    // ---- candidates sorted by level (ctor :74-78; the order inside a level lives in prio) --------------------------
    u8* cls = c.arena.get<u8>(n);
    u32* ckeys[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    u32* cvals[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    u32* d_cnt = c.arena.get<u32>(4);
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 4 * sizeof(u32), s));
    cand_class_kernel<<<(gn < 8192u ? gn : 8192u), 256, 0, s>>>(plcp, n, threshold, 0u, cls, fs.flen, nullptr, d_cnt + 1, nullptr, nullptr, nullptr, nullptr, nullptr);
    LAUNCH_CHECK();
    if (maxlcp < threshold || threshold == 0) { c.arena.release(mark); build_owner(c, n, fs); return; }
    mlcp_init_prio_kernel<<<gn, 256, 0, s>>>(prio, n);
    LAUNCH_CHECK();
    select_by_class(c, cls, 1, n, nullptr, cvals[0], nullptr, nullptr, d_cnt);
    const size_t cand_count = c.read(d_cnt);
    st->entries = cand_count;
    if (cand_count == 0) { c.arena.release(mark); build_owner(c, n, fs); return; }
    gather_kernel<<<cdiv(cand_count, 256), 256, 0, s>>>(cvals[0], cand_count, plcp, ckeys[0]);
    LAUNCH_CHECK();
    const int x = radix_sort_pairs_u32(c, ckeys, cvals, cand_count, 0, (int)bits_for(maxlcp));
    const size_t nlev = (size_t)maxlcp + 2;
    u64* skeys0 = c.arena.get<u64>(n + 2);                       // later the selection records; first the per-level candidate ranges
    u32* d_segstart = (u32*)skeys0;
    u32* d_segend = d_segstart + nlev;
    HIP_TRY(hipMemsetAsync(d_segstart, 0, nlev * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(d_segend, 0, nlev * sizeof(u32), s));
    seg_bounds_kernel<<<cdiv(cand_count, 256), 256, 0, s>>>(ckeys[x], cand_count, d_segstart, d_segend);
    LAUNCH_CHECK();
    std::vector<u32> h_segstart(nlev), h_segend(nlev);
    c.read_n(d_segstart, h_segstart.data(), nlev);
    c.read_n(d_segend, h_segend.data(), nlev);
    std::vector<u32> init_levels;                               // levels with original entries, descending
    for (u32 v = maxlcp; v >= threshold; --v) { if (h_segend[v] > h_segstart[v]) init_levels.push_back(v); if (v == 0) break; }
    const u32* cand = cvals[x];

    // ---- per-level state ------------------------------------------------------------------------------------------
    u32* live = c.arena.get<u32>(n);
    u32* pushed = c.arena.get<u32>(n);
    u32* stamp = c.arena.get<u32>(n);
    HIP_TRY(hipMemsetAsync(stamp, 0, n * sizeof(u32), s));
    const size_t bm_words = n / 32 + 2;
    u64* bm = c.arena.get<u64>(bm_words);
    HIP_TRY(hipMemsetAsync(bm, 0, bm_words * sizeof(u64), s));
    u64* skeys[2] = { skeys0, c.arena.get<u64>(n) };              // selected (prio, pos) records, later the level's pushes
    u32* svals[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
    const size_t pool_cap = 4 * n;
    u32* pool = c.arena.get<u32>(pool_cap);
    LevelScalars* d_sc = (LevelScalars*)c.arena.alloc(sizeof(LevelScalars));
    // one factor of length L can send up to L entries to L different levels (eager decreases): room for min(n, maxlcp + 2, 64 Mi) targets
    // (the ramp of one giant repeat -- a Fibonacci word of 40 M symbols -- fills millions of them)
    const u32 seg_cap = (u32)std::max<size_t>((size_t)1 << 16, std::min<size_t>(std::min<size_t>(n, (size_t)maxlcp + 2), (size_t)64 << 20));
    PushSeg* d_segs = (PushSeg*)c.arena.alloc(sizeof(PushSeg) * seg_cap);
    std::vector<PushSeg> h_segs;
    LevelScalars h_sc;
    const u32 gtab_cap = 1u << 16;
    GatherSeg* d_gtab = (GatherSeg*)c.arena.alloc(sizeof(GatherSeg) * gtab_cap);
    std::vector<GatherSeg> h_gtab(gtab_cap);
    struct PoolSeg { u32 off, cnt; };
    std::map<u32, std::vector<PoolSeg>> pushed_into;            // per target level: its segments of the pool
    size_t pool_top = 0, ip = 0;
    u32 tbase = 0, dead_streak = 0;

    for (;;) {
        const u32 l_init = ip < init_levels.size() ? init_levels[ip] : 0u;
        const u32 l_pool = pushed_into.empty() ? 0u : pushed_into.rbegin()->first;
        const u32 L = std::max(l_init, l_pool);
        if (L < threshold || L == 0) break;
        // a run of levels without a live entry: skip to the highest level that still holds one (candidates whose value is
        // still their PLCP value; pool segments -- at most 4 Mi of them, from the top -- with an entry whose value is the
        // segment's level)
        if (dead_streak >= 8) {
            HIP_TRY(hipMemsetAsync(d_cnt + 2, 0, sizeof(u32), s));
            if (l_init && h_segend[l_init] > 0) {
                const size_t hi = h_segend[l_init];
                unsigned g = cdiv(hi, 256 * 4); if (g > 4096) g = 4096; if (g == 0) g = 1;
                live_max_level_kernel<<<g, 256, 0, s>>>(ckeys[x], cvals[x], 0, hi, cur, d_cnt + 2);
                LAUNCH_CHECK();
            }
            std::vector<ProbeSeg> ptab;
            u32 pool_floor = 0;                                   // levels below this one were not examined
            for (auto itp = pushed_into.rbegin(); itp != pushed_into.rend(); ++itp) {
                if (ptab.size() + itp->second.size() > ((size_t)4 << 20) && !ptab.empty()) { pool_floor = itp->first + 1; break; }
                for (const PoolSeg& sg : itp->second) ptab.push_back(ProbeSeg{sg.off, sg.cnt, itp->first});
            }
            const size_t pmark = c.arena.mark();
            if (!ptab.empty()) {
                ProbeSeg* d_ptab = (ProbeSeg*)c.arena.alloc(ptab.size() * sizeof(ProbeSeg));
                HIP_TRY(hipMemcpyAsync(d_ptab, ptab.data(), ptab.size() * sizeof(ProbeSeg), hipMemcpyHostToDevice, s));
                unsigned g = cdiv(ptab.size(), 256); if (g > 4096) g = 4096;
                live_max_pool_kernel<<<g, 256, 0, s>>>(pool, d_ptab, ptab.size(), cur, d_cnt + 2);
                LAUNCH_CHECK();
            }
            const u32 live = std::max(c.read(d_cnt + 2), pool_floor ? pool_floor - 1 : 0u);   // synchronises: the host table has been copied
            c.arena.release(pmark);
            st->probes++;
            dead_streak = 0;
            if (live < L) {
                while (ip < init_levels.size() && init_levels[ip] > live) ++ip;
                while (!pushed_into.empty() && pushed_into.rbegin()->first > live) pushed_into.erase(std::prev(pushed_into.end()));
                continue;
            }
        }
        u32 m0 = 0;
        if (l_init == L) { m0 = h_segend[L] - h_segstart[L]; ++ip; }
        u32 m1 = 0;
        auto it = pushed_into.find(L);
        if (it != pushed_into.end()) {
            const std::vector<PoolSeg>& segsL = it->second;
            size_t done = 0;
            while (done < segsL.size()) {
                const size_t cntseg = std::min(segsL.size() - done, (size_t)gtab_cap);
                u32 tot = 0;
                for (size_t j = 0; j < cntseg; ++j) { h_gtab[j] = GatherSeg{segsL[done + j].off, tot}; tot += segsL[done + j].cnt; }
                if ((size_t)m1 + tot > n) throw HipError{hipErrorUnknown, "max_lcp: level list larger than the text", (int)__LINE__};
                if (cntseg == 1) {
                    HIP_TRY(hipMemcpyAsync(pushed + m1, pool + h_gtab[0].src_off, (size_t)tot * sizeof(u32), hipMemcpyDeviceToDevice, s));
                } else {
                    HIP_TRY(hipMemcpyAsync(d_gtab, h_gtab.data(), cntseg * sizeof(GatherSeg), hipMemcpyHostToDevice, s));
                    gather_segments_kernel<<<cdiv(tot, 256), 256, 0, s>>>(pool, d_gtab, (u32)cntseg, tot, pushed + m1);
                    LAUNCH_CHECK();
                    HIP_TRY(hipStreamSynchronize(s));            // h_gtab is reused by the next chunk
                }
                m1 += tot;
                done += cntseg;
            }
            pushed_into.erase(it);
        }
        const u32 m = m0 + m1;
        st->levels++;
        HIP_TRY(hipMemsetAsync(d_sc, 0, 8 * sizeof(u32), s));
        mlcp_classify_kernel<<<cdiv(m, 256), 256, 0, s>>>(cand + h_segstart[L], m0, pushed, m, L, cur, live, bm, d_sc);
        LAUNCH_CHECK();
        const u32 nl = c.read(&d_sc->nlive);
        if (nl == 0) { ++dead_streak; continue; }
        dead_streak = 0;
        const bool wide = L >= 128;
        const unsigned gl = wide ? cdiv((size_t)nl * 64, 256) : cdiv(nl, 256);
        for (u32 round = 0;; ++round) {
            HIP_TRY(hipMemsetAsync(&d_sc->undecided, 0, sizeof(u32), s));
            if (wide) mis_round_kernel<64><<<gl, 256, 0, s>>>(live, nl, L, n, prio, bm, d_sc);
            else      mis_round_kernel<1><<<gl, 256, 0, s>>>(live, nl, L, n, prio, bm, d_sc);
            LAUNCH_CHECK();
            st->rounds++;
            if (c.read(&d_sc->undecided) == 0) break;
            if (round > nl + 8) throw HipError{hipErrorUnknown, "max_lcp: selection does not converge", (int)__LINE__};
        }
        mlcp_collect_kernel<<<cdiv(nl, 256), 256, 0, s>>>(live, nl, prio, bm, skeys[0], svals[0], d_sc);
        LAUNCH_CHECK();
        const u32 nsel = c.read(&d_sc->selected);
        if (nsel == 0) throw HipError{hipErrorUnknown, "max_lcp: a live level selected nothing", (int)__LINE__};
        const int y = sort_pairs_u64_distinct(c, skeys, svals, nsel, 0, 32);
        const u32* sel = svals[y];
        const unsigned gs = wide ? cdiv((size_t)nsel * 64, 256) : cdiv(nsel, 256);
        if (wide) mlcp_apply_kernel<64><<<gs, 256, 0, s>>>(sel, nsel, L, n, phi, cur, fs.flen, fs.fsrc, stamp);
        else      mlcp_apply_kernel<1><<<gs, 256, 0, s>>>(sel, nsel, L, n, phi, cur, fs.flen, fs.fsrc, stamp);
        LAUNCH_CHECK();
        u64* bkeys[2] = { skeys[y ^ 1], skeys[y] };                // sel = svals[y] stays untouched while the pushes are collected
        if (wide) mlcp_push_kernel<64><<<gs, 256, 0, s>>>(sel, nsel, L, threshold, tbase, cur, stamp, prio, bkeys[0], d_sc);
        else      mlcp_push_kernel<1><<<gs, 256, 0, s>>>(sel, nsel, L, threshold, tbase, cur, stamp, prio, bkeys[0], d_sc);
        LAUNCH_CHECK();
        const u32 npush = c.read(&d_sc->npush);
        st->factors += nsel;
        tbase += nsel;
        if (npush) {
            if (pool_top + npush > pool_cap) throw HipError{hipErrorUnknown, "max_lcp: push pool exhausted", (int)__LINE__};
            u32* bvals[2] = { svals[0], svals[1] };               // values are not used; the records are (level, position) keys
            const int w = sort_pairs_u64_distinct(c, bkeys, bvals, npush, 0, 32 + (int)bits_for(L));
            mlcp_pool_kernel<<<cdiv(npush, 256), 256, 0, s>>>(bkeys[w], npush, pool + pool_top, d_segs, seg_cap, d_sc);
            LAUNCH_CHECK();
            c.read_n((const u32*)d_sc, (u32*)&h_sc, sizeof(LevelScalars) / sizeof(u32));
            const u32 nseg = h_sc.nseg;
            if (nseg > seg_cap) throw HipError{hipErrorUnknown, "max_lcp: too many push targets in one level", (int)__LINE__};
            if (h_segs.size() < nseg) h_segs.resize(nseg);
            if (nseg > SEG_INLINE) c.read_n(d_segs, h_segs.data(), nseg);
            else for (u32 j = 0; j < nseg; ++j) h_segs[j] = h_sc.segs[j];
            std::sort(h_segs.begin(), h_segs.begin() + nseg, [](const PushSeg& a, const PushSeg& b) { return a.start < b.start; });
            for (u32 j = 0; j < nseg; ++j) {
                const u32 end = (j + 1 < nseg) ? h_segs[j + 1].start : npush;
                const u32 tgt = h_segs[j].target;
                if (tgt >= L || tgt < threshold) throw HipError{hipErrorUnknown, "max_lcp: bad push target", (int)__LINE__};
                pushed_into[tgt].push_back(PoolSeg{(u32)pool_top + h_segs[j].start, end - h_segs[j].start});
            }
            pool_top += npush;
            st->pushes += npush;
        }
    }
    c.arena.release(mark);
    build_owner(c, n, fs);
}

}  // namespace tdc
