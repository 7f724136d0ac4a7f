// factorize_tiles.hip -- the LOW levels of lcpcomp::ArraysComp (compressors/lcpcomp/compress/ArraysComp.hpp:36-117),
// one text window per workgroup, all levels of the window inside ONE launch.
//
// The level loop of factorize.hip costs ~20 launches and 4-5 host round trips per level, and every level touches
// position-indexed arrays of the whole text.  But a level L only couples text positions at distance < L, so once the
// levels above `lcut` are done the remaining ones can be evaluated window by window from LDS:
//
//   * a workgroup loads the slice [w0, w1) of the global state (cur = working LCP, res = level whose list holds the
//     entry of a position, prio = list order) and runs the levels lcut .. threshold on it: collect the level's
//     entries, selection rounds (lexicographically-first maximal independent set among the live entries), encounter
//     values of the others, pushes (new residence + new priority), kills and truncations -- the same steps as the
//     global kernels, on bytes in LDS;
//   * everything outside the window is UNKNOWN.  [fl, fr) is the range of window positions whose state is still
//     exactly what the sequential algorithm would have.  An unknown factor of level L starts at an unknown position q
//     whose working value is still >= L; working values only ever decrease and the window holds an upper bound of them
//     everywhere (only the effects of certain factors are applied), so a border moves at level L exactly as far as such
//     a q exists within reach: on the left the factor covers up to q + L - 1, on the right it truncates down to
//     q - (L - 1).  (The worst case is L - 1 per level and side; on ordinary texts the borders move a few hundred
//     positions in total, because the high levels are sparse.)  An entry that could be affected by something unknown
//     (inside that reach, or next to such an entry with higher priority) gets the third selection state UNCERTAIN and
//     pushes the border past everything it could touch;
//   * a window is VALID iff its interior [a, b) is still inside [fl, fr) after the last level; only factors that start
//     in the interior are written.  The halo is a run-time value: the first attempt uses a small one, a window whose
//     known range shrank into its interior makes the pass retry with the largest one; if that fails too (or a fixed LDS
//     list overflows twice) the caller discards the results and runs the global level loop instead -- the global
//     state is never modified here.
//
// Inside a window the list order of pushed entries is a window-local rank (flag bit + counter): priorities are only
// ever compared between entries of one list at distance < L, i.e. inside one window, and a locally pushed entry follows
// every entry that was already in the list, exactly like the global priorities.
// Model: tests/models/position_space.py (factorize_tile), checked against the oracle on thousands of inputs.
#include "stages.hpp"
#include "prim.hpp"
#include "factorize_tiles.hpp"

#include <stdlib.h>

namespace tdc {

namespace {

#ifndef TDC_WIN_TW
#define TDC_WIN_TW 16384
#endif
#ifndef TDC_WIN_WPE
#define TDC_WIN_WPE 4
#endif
#ifndef TDC_WIN_TE
#define TDC_WIN_TE 512
#endif
#ifndef TDC_WIN_TP
#define TDC_WIN_TP 384
#endif
constexpr int TW = TDC_WIN_TW;       // window positions (a multiple of 64)
constexpr int TH_MAX = 2048;         // halo on either side: a run-time value (multiple of 4), at most this
#ifndef TDC_WIN_TT
#define TDC_WIN_TT 256
#endif
constexpr int TT = TDC_WIN_TT;       // threads per workgroup (the first TW / 64 of them own a 64-position chunk of the window: TT >= TW / 64)
constexpr int TCH = 64;              // consecutive window positions per thread in the dense passes (the first TW / 64 threads own a chunk)
constexpr int CPT = TW / (64 * TT) > 0 ? TW / (64 * TT) : 1;   // 64-position chunks per thread in the dense passes (thread t owns the chunks t * CPT ..)
static_assert(TW % 64 == 0 && (TW / 64 <= TT * CPT), "every chunk of the window needs an owner");
// Two sizes of the per-level LDS lists (alive entries / pushes per level and window).  The small one leaves 40 KB of LDS
// per workgroup, i.e. four workgroups per CU -- the kernel is latency bound, so its throughput follows the number of
// resident workgroups; the large one (one workgroup per CU) takes over if a level overflows the small lists, e.g. on
// texts with a random background, where a third of all positions sit in one level.
constexpr int TE_SMALL = TDC_WIN_TE, TP_SMALL = TDC_WIN_TP;
constexpr int TE_LARGE = 8192, TP_LARGE = 4096;
#ifndef TDC_WIN_TT_LARGE
#define TDC_WIN_TT_LARGE 1024
#endif
constexpr int TT_LARGE = TDC_WIN_TT_LARGE;     // threads of the large variant: one workgroup per CU, so it may as well fill the CU

// Position q of the window lives at byte PA(q) of the position-indexed LDS arrays.  A thread's dense passes read "its"
// 64-byte chunk with 8-byte loads; rotating the 16 words of chunk t by 2*(t>>2) words puts the 64 lanes' loads on 64
// different banks without any padding.
__device__ __forceinline__ int PA(int q) {
    const int t = q >> 6, w = (q >> 2) & 15;
    return (t << 6) + (((w + 2 * (t >> 2)) & 15) << 2) + (q & 3);
}
// byte offset of the k-th 8-byte word (k = 0..7) of thread t's chunk
__device__ __forceinline__ int PW(int t, int k) { return (t << 6) + (((2 * k + 2 * (t >> 2)) & 15) << 2); }
constexpr int BIG = 1 << 29;
// Entries are dealt round-robin to the four waves (entry i -> wave i mod 4): the phases of a level are latency chains, and a wave
// that takes 64 consecutive entries (so that waves without entries could skip a phase) was measured 10 % slower.
#define WDEAL_FIRST (lane * NWV + wv)
#define WDEAL_I0 0
#define WDEAL_OFF (lane * NWV + wv)

// An entry of the current level, packed so that a neighbour costs ONE LDS read:
//   [31:0] priority   [47:32] window position   [55:48] state (bit 7: priority is window-local)   [63:56] LCP value
enum : u32 { S_UND = 0, S_SEL = 1, S_REJ = 2, S_UNC = 3, S_STALE = 4, S_PUSH = 5, S_DROP = 6, S_MASK = 7, S_LOCAL = 0x80 };
// (all field accesses go through the 32-bit halves: 64-bit shifts are slow on this ISA)
__device__ __forceinline__ u32 e_hi(u64 e) { return (u32)(e >> 32); }
__device__ __forceinline__ int e_pos(u64 e) { return (int)(e_hi(e) & 0xFFFFu); }
__device__ __forceinline__ u32 e_state(u64 e) { return (e_hi(e) >> 16) & S_MASK; }
__device__ __forceinline__ u32 e_stbyte(u64 e) { return (e_hi(e) >> 16) & 0xFFu; }
__device__ __forceinline__ u32 e_val(u64 e) { return e_hi(e) >> 24; }
// list order: window-local priorities follow global ones
__device__ __forceinline__ bool e_before(u64 a, u64 b) {
    const u32 fa = e_hi(a) & 0x800000u, fb = e_hi(b) & 0x800000u;
    return (fa != fb) ? (fa < fb) : ((u32)a < (u32)b);
}
__device__ __forceinline__ u64 e_key(u64 e) { return ((u64)((e_hi(e) >> 23) & 1u) << 32) | (u32)e; }
// (entries are re-read after other threads changed them: lds_load / lds_store keep the accesses real ds_ instructions)
__device__ __forceinline__ void e_set_state(u64* ent, int i, u32 st) { lds_store((u8*)&ent[i] + 6, (u8)st); }
__device__ __forceinline__ void e_set_val(u64* ent, int i, u32 v) { lds_store((u8*)&ent[i] + 7, (u8)v); }
__device__ __forceinline__ u64 e_load(const u64* ent, int i) { return lds_load(&ent[i]); }

struct WinScalars { u32 fail; int min_margin; unsigned long long factors; u32 max_entries, max_pushes; unsigned long long prof[16]; };

// optional phase timing (compile with -DTDC_WIN_PROF): thread 0 of every workgroup sums s_memrealtime deltas per phase
#ifdef TDC_WIN_PROF
#define WPROF_DECL unsigned long long wp_t = __builtin_readcyclecounter(); unsigned long long wp_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
#define WPROF(i) do { const unsigned long long wp_n = __builtin_readcyclecounter(); wp_acc[i] += wp_n - wp_t; wp_t = wp_n; } while (0)
#define WPROF_CNT(i, v) do { wp_acc[i] += (v); } while (0)
#define WPROF_FLUSH do { if (tid == 0) for (int wi = 0; wi < 12; ++wi) atomicAdd(&sc->prof[wi], wp_acc[wi]); } while (0)
#else
#define WPROF_DECL
#define WPROF(i) do {} while (0)
#define WPROF_CNT(i, v) do {} while (0)
#define WPROF_FLUSH do {} while (0)
#endif

// Workgroup barrier that only waits for this wave's LDS traffic.  __syncthreads() also drains the vector-memory counter,
// i.e. every barrier behind a global store would cost a round trip to L2.  Global data that IS handed between threads
// (the window-local priorities) is ordered by the one full barrier per level.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KT, int TE, int TP, int WPE, bool WPHI>      // threads, list sizes; WPE = waves per SIMD the register budget is set for; WPHI: sources from a Phi array
__global__ __launch_bounds__(KT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void window_levels_kernel(const u32* __restrict__ cur_g, const u32* __restrict__ prio_g,
                                                            const u8* __restrict__ res_g, const u32* __restrict__ phi, size_t n,
                                                            u32 lcut, u32 threshold, u32 ntiles, u32 halo, u32* __restrict__ lprio_all,
                                                            u32* __restrict__ flen, u32* __restrict__ fsrc, WinScalars* __restrict__ sc) {
    // (the constants of the small variant, redefined for this instance's thread count: the large lists leave room for one workgroup per
    //  CU, which then runs 1 024 threads instead of 256)
    constexpr int TT = KT;
    constexpr int NWV = KT / 64;
    constexpr int CPT = TW / (64 * KT) > 0 ? TW / (64 * KT) : 1;
    static_assert(TW / 64 <= KT * CPT, "every chunk of the window needs an owner");
    __shared__ __attribute__((aligned(16))) u8 cur8[TW];
    __shared__ __attribute__((aligned(16))) u8 res8[TW];    // [5:0] list level, bit 6: factor start (then [5:0] = length), bit 7: priority is window-local
    __shared__ u64 ent[TE];
    __shared__ u64 pkey[TP];
    __shared__ unsigned short pidx[TP];
    __shared__ u32 wtot[NWV];
    __shared__ int s_und[3];
    __shared__ int s_npush, s_tl, s_tr;
    __shared__ u64 s_lvlmask;             // bit L: some window position resides in list L
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    u32* lprio = lprio_all + (size_t)blockIdx.x * TW;
    WPROF_DECL

    const size_t TH = halo, TI = (size_t)TW - 2 * (size_t)halo;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t a = (size_t)tile * TI;
        const size_t b = (a + TI < n) ? a + TI : n;
        const size_t w0 = (a >= (size_t)TH) ? a - TH : 0;
        const size_t w1 = (b + TH < n) ? b + TH : n;
        const int wl = (int)(w1 - w0);
        const int ia = (int)(a - w0), ib = (int)(b - w0);
        const int mid = wl / 2;
        __syncthreads();                                    // previous window's LDS is no longer read
        WPROF(0);
        if (tid == 0) { s_und[0] = 0; s_und[1] = 0; s_und[2] = 0; s_npush = 0; s_tl = -BIG; s_tr = BIG; s_lvlmask = 0; }
        __syncthreads();
        // ---- load the window: cur and residence as bytes (cur <= lcut everywhere once the levels above are done) ----
        u64 mymask = 0;
        for (int i = tid * 4; i < TW; i += TT * 4) {
            u32 cw = 0, rw = 0;
            const size_t gp = w0 + i;
            if (gp + 4 <= w1) {
                const uint4 cv = *(const uint4*)(cur_g + gp);                  // w0 and i are multiples of 4
                const u32 rv4 = *(const u32*)(res_g + gp);
                const u32 c4[4] = { cv.x, cv.y, cv.z, cv.w };
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    u32 rv = (rv4 >> (8 * k)) & 0xFFu;
                    if (rv > lcut) rv = 0;                  // entries of higher lists are gone (selected, dropped or pushed down)
                    cw |= (c4[k] > 255u ? 255u : c4[k]) << (8 * k);
                    rw |= rv << (8 * k);
                    mymask |= 1ull << rv;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (gp + k < w1) {
                        const u32 cv = cur_g[gp + k];
                        u32 rv = res_g[gp + k];
                        if (rv > lcut) rv = 0;
                        cw |= (cv > 255u ? 255u : cv) << (8 * k);
                        rw |= rv << (8 * k);
                        mymask |= 1ull << rv;
                    }
                }
            }
            *(u32*)&cur8[PA(i)] = cw;
            *(u32*)&res8[PA(i)] = rw;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mymask |= __shfl_xor(mymask, d, 64);
        if (lane == 0) atomicOr((unsigned long long*)&s_lvlmask, (unsigned long long)mymask);
        // known range [fl, fr) in window offsets.  The outermost lcut - 1 positions of a side that has unknown text behind it only
        // serve as upper bounds of what unknown factors can reach (see "borders" below): nothing outside the window reaches further.
        int fl = (w0 > 0) ? (int)lcut - 1 : -BIG;
        int fr = (w1 < n) ? wl - ((int)lcut - 1) : BIG;
        u32 local_base = 0;
        u32 nsel_interior = 0;
        bool failed = false;
        __syncthreads();
        WPROF(1);

        for (u32 L = lcut; L >= threshold && !failed; --L) {
            const int iL = (int)L;
            // ---- borders: an unknown factor of this level starts at an unknown position q whose working value is still >= L.  Working
            //      values only ever decrease and the window holds an upper bound of them everywhere (only the effects of certain
            //      factors were applied), so the borders move exactly as far as such a q exists: on the left it covers up to q + L - 1,
            //      on the right it truncates down to q - (L - 1).  (Every wave evaluates this for itself: two LDS reads, two ballots.)
            int dfl = fl, dfr = fr;
            if (fl > -BIG / 2) {
                const int q = fl - 1 - lane;
                const u64 mq = __ballot(lane < iL - 1 && q >= 0 && (u32)cur8[PA(q >= 0 ? q : 0)] >= L);
                if (mq) dfl = fl - 1 - __builtin_ctzll(mq) + iL;
            }
            if (fr < BIG / 2) {
                const int q = fr + lane;
                const u64 mq = __ballot(lane < iL - 1 && q < wl && (u32)cur8[PA(q < wl ? q : 0)] >= L);
                if (mq) dfr = fr + __builtin_ctzll(mq) - (iL - 1);
            }
            const u64 lvl_seen = lds_load(&s_lvlmask);
            if (((lvl_seen >> L) & 1ull) == 0) { fl = dfl; fr = dfr; continue; }   // nothing resides in list L
            const int lo = fl > 0 ? fl : 0, hi = fr < wl ? fr : wl;
            // ---- 1. collect the alive entries of list L in position order --------------------------------------
            // (branch-free per 8-position word: byte flags 0x80 for "resides in list L" and for "still alive", a shift cascade turns
            //  the flags into bits of the thread's 64-position mask; erased entries (:86) leave the list by a masked word store.
            //  Positions outside the known range are not collected; erasing a dead entry there is harmless -- the range only shrinks.)
            u64 amasks[CPT];
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                const int chunk = tid * CPT + cc;
                const int base = chunk * TCH;
                u64 amask = 0;
                if (base < TW && base < hi && base + TCH > lo) {
                    const u64 pat = (u64)L * 0x0101010101010101ull;
                    const u64 lo7 = 0x7F7F7F7F7F7F7F7Full, hi1 = 0x8080808080808080ull;
                    const u64 addc = (u64)(128u - threshold) * 0x0101010101010101ull;     // (c & 0x7F) + addc has bit 7 set iff (c & 0x7F) >= threshold
                    u64 rws[TCH / 8], cws[TCH / 8];
#pragma unroll
                    for (int k = 0; k < TCH / 8; ++k) { rws[k] = *(const u64*)&res8[PW(chunk, k)]; cws[k] = *(const u64*)&cur8[PW(chunk, k)]; }
#pragma unroll
                    for (int k = 0; k < TCH / 8; ++k) {
                        const u64 x = (rws[k] & lo7) ^ pat;
                        const u64 hit = ~(((x & lo7) + lo7) | x | lo7);            // 0x80 in every byte of x that is zero (exact)
                        const u64 alive = (((cws[k] & lo7) + addc) | cws[k]) & hi1;  // 0x80 where cur >= threshold
                        u64 ha = (hit & alive) >> 7;                                // flag bits at 0, 8, .., 56 -> bits 0..7
                        ha |= ha >> 7; ha |= ha >> 14; ha |= ha >> 28;
                        amask |= (ha & 0xFFull) << (8 * k);
                        const u64 he = (hit & ~alive) >> 7;                         // erased entries of this word
                        if (he) *(u64*)&res8[PW(chunk, k)] = rws[k] & ~((he << 8) - he);
                    }
                    const int rlo = lo - base, rhi = hi - base;                     // known range, relative to the chunk
                    if (rlo > 0) amask &= ~((1ull << rlo) - 1ull);
                    if (rhi < TCH) amask &= (1ull << rhi) - 1ull;
                }
                amasks[cc] = amask;
            }
            // the first priorities are requested before the barrier of the scan, so their latency overlaps it
            u32 pre0 = 0, pre1 = 0;
            {
                u64 mm = amasks[0];
                const int base = tid * CPT * TCH;
                if (mm) { pre0 = prio_g[w0 + base + __builtin_ctzll(mm)]; mm &= mm - 1; }   // (a window-local priority is read behind the full barrier below)
                if (mm) { pre1 = prio_g[w0 + base + __builtin_ctzll(mm)]; }
            }
            WPROF(2);
            u32 cnt = 0;
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) cnt += (u32)__popcll(amasks[cc]);
            const u32 inc = wave_inclusive_sum(cnt);
            if (lane == 63) wtot[wv] = inc;
            __syncthreads();                                // the level's one FULL barrier: lprio stores of earlier levels are complete
            u32 off = inc - cnt, total = 0;
#pragma unroll
            for (int k = 0; k < NWV; ++k) { const u32 t = wtot[k]; if (k < wv) off += t; total += t; }
            if (total > (u32)TE) { failed = true; break; }
            const int m = (int)total;
#ifdef TDC_WIN_PROF
            if (tid == 0) atomicMax(&sc->max_entries, total);
#endif
            int my_und = 0;
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                u64 amask = amasks[cc];
                const int base = (tid * CPT + cc) * TCH;
                for (int q = 0; amask; ++q) {
                    const int bit = __builtin_ctzll(amask);
                    amask &= amask - 1;
                    const int pos = base + bit;
                    const u32 cv = cur8[PA(pos)];
                    const u32 local = res8[PA(pos)] & S_LOCAL;
                    const u32 pr = local ? lprio[pos] : ((cc == 0 && q == 0) ? pre0 : (cc == 0 && q == 1) ? pre1 : prio_g[w0 + pos]);
                    const u32 st = (cv == L ? S_UND : S_STALE) | local;
                    ent[off] = ((u64)((cv << 24) | (st << 16) | (u32)pos) << 32) | pr;
                    if (cv == L) ++my_und;
                    ++off;
                }
            }
            my_und = wave_reduce_sum(my_und);
            if (lane == 0 && my_und) atomicAdd(&s_und[0], my_und);
            lds_barrier();
            WPROF(3);
            WPROF_CNT(8, 1); WPROF_CNT(9, m);
            if (m == 0) { fl = dfl; fr = dfr; continue; }

            // ---- 2. selection rounds (in place: a state only ever moves away from UNDECIDED) ---------------------
            // rotating counters: a round reads s_und[r], counts the entries it leaves undecided in s_und[r+1], clears s_und[r+2]
            int r = 0;
            for (int guard = 0; guard <= m; ++guard) {
                const int und = lds_load(&s_und[r]);
                if (und == 0) break;
                const int rn = (r + 1) % 3, rc = (r + 2) % 3;
                if (tid == 0) s_und[rc] = 0;
                for (int i = WDEAL_FIRST; i < m; i += TT) {
                    const u64 e = e_load(ent, i);
                    u64 fL = (i > 0) ? e_load(ent, i - 1) : 0ull;               // both direct neighbours are requested up front
                    u64 fR = (i + 1 < m) ? e_load(ent, i + 1) : 0ull;
                    if (e_state(e) != S_UND) continue;
                    const int p = e_pos(e);
                    bool hit = false, blocked = false, unc = false;
                    for (int j = i - 1; j >= 0; --j) {
                        const u64 f = (j == i - 1) ? fL : e_load(ent, j);
                        if (p - e_pos(f) >= iL) break;
                        const u32 s = e_state(f);
                        if (s == S_SEL) { hit = true; break; }
                        if ((s == S_UND || s == S_UNC) && e_before(f, e)) { if (s == S_UND) blocked = true; else unc = true; }
                    }
                    for (int j = i + 1; j < m && !hit; ++j) {
                        const u64 f = (j == i + 1) ? fR : e_load(ent, j);
                        if (e_pos(f) - p >= iL) break;
                        const u32 s = e_state(f);
                        if (s == S_SEL) { hit = true; break; }
                        if ((s == S_UND || s == S_UNC) && e_before(f, e)) { if (s == S_UND) blocked = true; else unc = true; }
                    }
                    const u32 keep = e_stbyte(e) & S_LOCAL;
                    if (hit) e_set_state(ent, i, S_REJ | keep);
                    else if (!blocked) {
                        const bool exposed = (p < dfl) || (p >= dfr);
                        e_set_state(ent, i, ((unc || exposed) ? S_UNC : S_SEL) | keep);
                    } else {                                    // (still undecided: one atomic for the lanes that are)
                        const u64 act = __ballot(true);
                        if (lane == __builtin_ctzll(act)) atomicAdd(&s_und[rn], (int)__popcll(act));
                    }
                }
                lds_barrier();
                r = rn;
                WPROF_CNT(10, 1);
            }
            WPROF(4);
            if (lds_load(&s_und[r]) != 0) { failed = true; break; }     // cannot happen (the best undecided entry always decides)

            // ---- 3. encounter values of the stale and the rejected entries, taint of the uncertain ones; the selected
            //         entries truncate the positions in front of them (their ranges are disjoint) and are written out ------
            for (int i0 = WDEAL_I0; i0 < m; i0 += TT) {
                const int i = i0 + WDEAL_OFF;
                const bool have = i < m;
                const u64 e = have ? e_load(ent, i) : 0ull;
                const u64 fL = (have && i > 0) ? e_load(ent, i - 1) : 0ull;
                const u64 fR = (have && i + 1 < m) ? e_load(ent, i + 1) : 0ull;
                const u32 s = have ? e_state(e) : (u32)S_DROP;
                const int p = e_pos(e);
                // selected entries: the lanes of the wave share the L positions in front of each of them; four entries per
                // batch (their ranges are disjoint), so the four reads are in flight together
                u64 selm = __ballot(s == S_SEL);
                while (selm) {
                    int qs[4];
                    u32 cv[4];
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) {
                        qs[b4] = -1;
                        if (selm) {
                            const int src = __builtin_ctzll(selm);
                            selm &= selm - 1;
                            const int q = __builtin_amdgcn_readlane(p, src) - 1 - lane;
                            if (lane < iL && q >= 0) qs[b4] = PA(q);
                        }
                    }
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) cv[b4] = (qs[b4] >= 0) ? cur8[qs[b4]] : 0u;
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) if (qs[b4] >= 0 && cv[b4] > (u32)lane + 1) cur8[qs[b4]] = (u8)(lane + 1);
                }
                if (s == S_SEL) continue;                   // (marked as a factor start in step 4, written out after the last level)
                if (s == S_UNC) {                           // its factor may or may not exist: everything it could touch is unknown
                    if (p < mid) atomicMax(&s_tl, p + iL); else atomicMin(&s_tr, p - (iL - 1));
                    continue;
                }
                if (s != S_STALE && s != S_REJ) continue;
                u32 v = e_val(e);
                bool uncertain = (p < dfl) || (p >= dfr);
                for (int j = i - 1; j >= 0; --j) {
                    const u64 f = (j == i - 1) ? fL : e_load(ent, j);
                    if (p - e_pos(f) >= iL) break;
                    const u32 t = e_state(f);
                    if ((t == S_SEL || t == S_UNC) && e_before(f, e)) { if (t == S_SEL) v = 0; else uncertain = true; }   // covered (:99-101)
                }
                for (int j = i + 1; j < m; ++j) {
                    const u64 f = (j == i + 1) ? fR : e_load(ent, j);
                    const int d = e_pos(f) - p;
                    if (d >= iL) break;
                    const u32 t = e_state(f);
                    if ((t == S_SEL || t == S_UNC) && e_before(f, e)) { if (t == S_SEL) { if ((u32)d < v) v = (u32)d; } else uncertain = true; }   // truncated (:103-109)
                }
                if (uncertain) {
                    if (p < mid) atomicMax(&s_tl, p + 1); else atomicMin(&s_tr, p);
                    e_set_state(ent, i, S_DROP);
                } else if (v >= threshold) {
                    // (the lanes that push in this step share one atomic: two thirds of all entry visits are pushes, and one LDS word
                    //  per push serialises; the level mask is only touched for a level it does not show yet)
                    const u64 act = __ballot(true);
                    const int leader = __builtin_ctzll(act);
                    int kb = 0;
                    if (lane == leader) kb = atomicAdd(&s_npush, (int)__popcll(act));
                    const int k = __builtin_amdgcn_readlane(kb, leader) + (int)__popcll(act & ((1ull << lane) - 1ull));
                    if (k < TP) { pkey[k] = e_key(e); pidx[k] = (unsigned short)i; }
                    e_set_val(ent, i, v);
                    e_set_state(ent, i, S_PUSH);
                    if (!((lvl_seen >> v) & 1ull)) atomicOr((unsigned long long*)&s_lvlmask, 1ull << v);
                } else e_set_state(ent, i, S_DROP);
            }
            lds_barrier();
            WPROF(5);
            const int npush = s_npush;
#ifdef TDC_WIN_PROF
            if (tid == 0) atomicMax(&sc->max_pushes, (u32)npush);
#endif
            if (npush > TP) { failed = true; break; }
            // ---- 4. kills; new residence / priority of the pushed entries, every other entry leaves the lists ------
            if (tid == 0) { s_und[0] = 0; s_und[1] = 0; s_und[2] = 0; }
            for (int i0 = WDEAL_I0; i0 < m; i0 += TT) {
                const int i = i0 + WDEAL_OFF;
                const bool have = i < m;
                const u64 e = have ? e_load(ent, i) : 0ull;
                const u32 s = have ? e_state(e) : (u32)S_PUSH;
                const int p = e_pos(e);
                u64 selm = __ballot(s == S_SEL);
                while (selm) {                              // kills (:99-101), one lane per covered position
                    const int src = __builtin_ctzll(selm);
                    selm &= selm - 1;
                    const int q = __builtin_amdgcn_readlane(p, src) + lane;
                    if (lane < iL && q < wl) cur8[PA(q)] = 0;
                }
                if (s != S_PUSH) res8[PA(p)] = (s == S_SEL) ? (u8)(0x40u | L) : (u8)0;
            }
            if (npush) {
                // rank by old priority: G lanes share one pushed entry
                int G = 1;
                while (G < 64 && npush * (G * 2) <= TT) G *= 2;
                const int per = TT / G;
                for (int k0 = 0; k0 < npush; k0 += per) {
                    const int k = k0 + tid / G;
                    const int sub = tid % G;
                    const bool act = k < npush;
                    const u64 key = act ? pkey[k] : 0ull;
                    u32 rank = 0;
                    if (act) {
#pragma unroll 8
                        for (int kk = sub; kk < npush; kk += G) rank += (pkey[kk] < key) ? 1u : 0u;
                    }
                    for (int d = G >> 1; d >= 1; d >>= 1) rank += __shfl_xor(rank, d, 64);
                    if (act && sub == 0) {
                        const int i = pidx[k];
                        const u64 e = e_load(ent, i);
                        const int p = e_pos(e);
                        lprio[p] = local_base + rank;
                        res8[PA(p)] = (u8)(e_val(e) | S_LOCAL);
                    }
                }
            }
            local_base += (u32)npush;
            int nfl = dfl, nfr = dfr;
            const int tl = s_tl, tr = s_tr;
            if (tl > nfl) nfl = tl;
            if (tr < nfr) nfr = tr;
            fl = nfl; fr = nfr;
            lds_barrier();
            WPROF(6);
            WPROF_CNT(11, npush);
            if (tid == 0) { s_npush = 0; s_tl = -BIG; s_tr = BIG; }
        }
        if (tid == 0 && !failed) {                          // smallest distance left between a known-range border and the interior
            int mg = BIG;
            if (fl > -BIG / 2) mg = ia - fl;
            if (fr < BIG / 2 && fr - ib < mg) mg = fr - ib;
            atomicMin(&sc->min_margin, mg);
        }
        if (failed || fl > ia || fr < ib) { if (tid == 0) atomicOr(&sc->fail, failed ? 2u : 1u); }   // 2: a fixed LDS list overflowed, 1: known range too small
        else {
            // ---- factor starts of the interior: (pos, Phi[pos], L)  (ArraysComp.hpp:91-96) ---------------------------
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                const int chunk = tid * CPT + cc;
                const int base = chunk * TCH;
                if (base < TW && base < ib && base + TCH > ia) {
#pragma unroll
                    for (int k = 0; k < TCH; k += 8) {
                        const u64 rw = *(const u64*)&res8[PW(chunk, k >> 3)];
                        u64 w = rw & 0x4040404040404040ull;
                        while (w) {
                            const int bb = __builtin_ctzll(w) >> 3;
                            w &= w - 1;
                            const int pos = base + k + bb;
                            if (pos < ia || pos >= ib) continue;
                            const size_t gp = w0 + pos;
                            flen[gp] = (u32)(rw >> (8 * bb)) & 0x3Fu;
                            if constexpr (WPHI) fsrc[gp] = phi[gp];     // (without a Phi array the sources are computed from SA[ISA[p] - 1] where they
                                                                        //  are needed: FactorSpace::src_prio, flatten.hip)
                            ++nsel_interior;
                        }
                    }
                }
            }
        }
        nsel_interior = wave_reduce_sum(nsel_interior);
        if (lane == 0 && nsel_interior) atomicAdd(&sc->factors, (unsigned long long)nsel_interior);
        WPROF(7);
    }
    WPROF_FLUSH;
}

// a failed pass leaves factors of the low levels behind: remove them (factors of the global levels are longer than lcut)
__global__ void window_cleanup_kernel(u32* __restrict__ flen, size_t n, u32 lcut) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && flen[p] <= lcut) flen[p] = 0;
}

}  // namespace

u32 window_levels_max_lcut() { return 63; }
u32 window_levels_window() { return (u32)TW; }
u32 window_levels_small_list() { return (u32)TE_SMALL; }
size_t window_levels_min_text() { return (size_t)4 * TW; }

int factorize_window_levels(Ctx& c, size_t n, const u32* cur, const u32* prio, const u8* res8, const u32* phi, u32 lcut, u32 threshold,
                            FactorSpace fs, u64* nfactors, bool start_large) {
    *nfactors = 0;
    if (lcut < threshold) return 0;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    // Attempts: (halo, list size).  The borders of the known range normally move a few hundred positions (they only move where an
    // unknown factor can exist), so the first attempt uses a small halo; a window whose known range shrank into its interior makes
    // the pass retry with the largest halo (the worst case of lcut = 63 needs 1 953 + the jumps), an overflowing per-level list
    // makes it retry with the large lists.
    u32 halo = (u32)c.window_halo & ~3u;
    if (halo < 2 * lcut + 64) halo = (2 * lcut + 64 + 3) & ~3u;
    if (halo > (u32)TH_MAX) halo = TH_MAX;
    bool large = c.window_large_lists != 0 || start_large;
    const u32 max_grid = 512u * TDC_WIN_WPE * (256 / TT);     // two rounds of resident workgroups
    u32* lprio = c.arena.get<u32>((size_t)max_grid * TW);
    WinScalars* d_sc = (WinScalars*)c.arena.alloc(sizeof(WinScalars));
    WinScalars h;
    int result = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const size_t ti = (size_t)TW - 2 * (size_t)halo;
        const u32 ntiles = cdiv(n, ti);
        const u32 grid = ntiles < max_grid ? ntiles : max_grid;
        HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(WinScalars), s));
        { const int big = BIG; HIP_TRY(hipMemcpyAsync(&d_sc->min_margin, &big, sizeof(int), hipMemcpyHostToDevice, s)); }
        {
            // per window position: cur (4) + residence (1); per text position: ~0.3 priority reads and the factor output
            Ctx::ProfScope prof(c, K_WINDOW_LEVELS, (u64)((double)n * TW / ti * 5) + (u64)n * 2);
            if (!large) {
                if (phi) window_levels_kernel<TT, TE_SMALL, TP_SMALL, TDC_WIN_WPE, true><<<grid, TT, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, lprio, fs.flen, fs.fsrc, d_sc);
                else window_levels_kernel<TT, TE_SMALL, TP_SMALL, TDC_WIN_WPE, false><<<grid, TT, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, lprio, fs.flen, fs.fsrc, d_sc);
            } else {
                if (phi) window_levels_kernel<TT_LARGE, TE_LARGE, TP_LARGE, TT_LARGE / 256, true><<<grid < 512u ? grid : 512u, TT_LARGE, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, lprio, fs.flen, fs.fsrc, d_sc);
                else window_levels_kernel<TT_LARGE, TE_LARGE, TP_LARGE, TT_LARGE / 256, false><<<grid < 512u ? grid : 512u, TT_LARGE, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, lprio, fs.flen, fs.fsrc, d_sc);
            }
            LAUNCH_CHECK();
        }
        h = c.read(d_sc);
#ifdef TDC_WIN_PROF
        {
            static const char* nm[12] = { "wait_prev", "load", "dense", "scan+write", "mis", "resolve", "apply", "tail", "levels", "entries", "rounds", "pushes" };
            for (int i = 0; i < 12; ++i) fprintf(stderr, "winprof %-10s %llu\n", nm[i], h.prof[i]);
            fprintf(stderr, "winprof windows %u grid %u attempt %d halo %u min_margin %d fail %u max_entries %u max_pushes %u\n", ntiles, grid, attempt,
                    halo, h.min_margin, h.fail, h.max_entries, h.max_pushes);
        }
#endif
        if (getenv("TDC_GPU_LEVEL_LOG")) fprintf(stderr, "window pass: attempt %d halo %u lists %s -> fail %u, smallest margin %d\n", attempt, halo, large ? "large" : "small", h.fail, h.min_margin);
        if (c.window_force_fail) h.fail |= 1u;                 // (tests: the pass is discarded as if a border had failed with the largest halo)
        result = (int)h.fail;
        if (!h.fail) break;
        window_cleanup_kernel<<<cdiv(n, 256), 256, 0, s>>>(fs.flen, n, lcut);     // forget the factors of the failed pass
        LAUNCH_CHECK();
        bool again = false;
        if ((h.fail & 2u) && !large) { large = true; again = true; }
        // (started with the large lists on a guess and a border failed: the retry keeps them -- they hold whatever the small ones do)
        if ((h.fail & 1u) && halo < (u32)TH_MAX && !c.window_force_fail) { halo = TH_MAX; again = true; }
        if (!again) break;
    }
    c.arena.release(mark);
    if (result == 0) *nfactors = h.factors;
    return result;
}

}  // namespace tdc
