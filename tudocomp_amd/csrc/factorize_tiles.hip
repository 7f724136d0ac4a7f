// factorize_tiles.hip -- the LOW levels of lcpcomp::ArraysComp (compressors/lcpcomp/compress/ArraysComp.hpp:36-117),
// one text window per workgroup, all levels of the window inside ONE launch.
//
// Round 5: the pass works on the factor SET, not on the emission order.  LCPCompressor sorts the factors by position before
// anything reads them (compressors/lzss/LZSSFactors.hpp:69-76), so the order in which ArraysComp.hpp:82-110 emits them is not
// observable; what IS observable is which of two conflicting entries of one level is selected first.  Call an entry NATURAL
// while its working value is still its PLCP value and TRUNCATED once a selected factor at p = x + cur[x] has cut it
// (:105-109).  For a truncated entry x with cur[x] = v:
//   * no position in (x, x + v) holds the value v (the factor at x + v cut everything there to less);
//   * no TRUNCATED entry of value v lies in (x - v, x) (its cutting factor would start inside (x, x + v) and would have cut
//     x further);
// so truncated entries of one level never meet each other, and the only entries they can lose against are NATURAL ones to
// their left -- originals of list v, which precede every pushed-down entry (:85-89 appends).  Consequences:
//   1. the relative order of pushed-down entries is irrelevant: no window-local priorities, no ranking of pushes;
//   2. so is the list an entry waits in: a cut entry moves to list cur[x] AT ONCE (eager push-down) -- the level of an entry
//      IS its working value, there are no stale entries, and an entry is visited once per value it is actually decided at
//      (the lazy formulation of rounds 1-4 spent 65 % of its entry visits on push-downs);
//   3. priorities (= ISA) are only ever compared between NATURAL entries of the same level within distance < L.
// Model: tests/models/position_space.py (factorize_eager, factorize_tile_eager), equal to the oracle's factor set on 12 000
// random / periodic / run-rich inputs with the truncated entries of every level visited in shuffled order.
//
// One state byte per window position in LDS:  [5:0] working value (0: dead)   bit 6: a factor of this length STARTS here
// bit 7: truncated.  Per level L (lcut .. threshold; levels without an entry are skipped through a 64-bit level mask):
//   1. dense scan of the bytes for value == L -> the level's entries in position order (32-bit words: position, state,
//      truncated); natural entries fetch their priority;
//   2. selection rounds: the lexicographically-first maximal independent set (natural before truncated, natural vs natural
//      by priority); an entry decides once every earlier entry within distance < L has;
//   3. the selected entries cut the L - 1 positions in front of them (value k | truncated at distance k) and kill the L
//      positions they cover; the start keeps L | bit 6.
// Everything outside the window is UNKNOWN; [fl, fr) is the range of window positions whose state is still exactly what
// the sequential algorithm would have.  An unknown factor of level L starts at an unknown position q whose working value
// is still >= L; the window holds an upper bound of the working values everywhere (only the effects of certain factors are
// applied), so a border moves at level L exactly as far as such a q exists within reach: on the left the factor covers up
// to q + L - 1, on the right it cuts down to q - (L - 1).  An entry that could be affected by something unknown (inside
// that reach, or behind an uncertain entry that precedes it) gets the state UNCERTAIN and pushes the border past everything
// it could touch.  A window is VALID iff its interior [a, b) is still inside [fl, fr) after the last level; only factors that
// start in the interior are written.  The halo is a run-time value: a window whose known range shrank into its interior
// makes the pass retry with the largest one; if that fails too (or a level overflows even the large entry list) the caller
// discards the results and runs the global level loop instead -- the global state is never modified here.
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
#define TDC_WIN_WPE 7
#endif
#ifndef TDC_WIN_LB
#define TDC_WIN_LB 4
#endif
#ifndef TDC_WIN_TE
#define TDC_WIN_TE 512
#endif
#ifndef TDC_WIN_TT
#define TDC_WIN_TT 256
#endif
constexpr int TW = TDC_WIN_TW;       // window positions (a multiple of 64)
constexpr int TH_MAX = 2048;         // halo on either side: a run-time value (multiple of 4), at most this
constexpr int TT = TDC_WIN_TT;       // threads per workgroup of the small variant
constexpr int TCH = 64;              // consecutive window positions per thread in the dense scan (thread t owns the chunks t * CPT ..)
// Two sizes of the per-level entry list.  The small one leaves ~21 KB of LDS per workgroup; the large one (one workgroup of
// 1 024 threads per CU) takes over if a level overflows it, e.g. on texts with a random background, where a third of all
// positions sit in one level.
constexpr int TE_SMALL = TDC_WIN_TE;
constexpr int TE_LARGE = 8192;
#ifndef TDC_WIN_TT_LARGE
#define TDC_WIN_TT_LARGE 1024
#endif
constexpr int TT_LARGE = TDC_WIN_TT_LARGE;

// Position q of the window lives at byte PA(q) of the state array.  A thread's dense scan reads "its" 64-byte chunk with
// 8-byte loads; rotating the 16 words of chunk t by 2*(t>>2) words puts the 64 lanes' loads on 64 different banks without
// any padding.
__device__ __forceinline__ int PA(int q) {
    const int t = q >> 6, w = (q >> 2) & 15;
    return (t << 6) + (((w + 2 * (t >> 2)) & 15) << 2) + (q & 3);
}
// byte offset of the k-th 8-byte word (k = 0..7) of thread t's chunk
__device__ __forceinline__ int PW(int t, int k) { return (t << 6) + (((2 * k + 2 * (t >> 2)) & 15) << 2); }
constexpr int BIG = 1 << 29;

// state byte
enum : u32 { B_VAL = 0x3Fu, B_START = 0x40u, B_TRUNC = 0x80u };
// upper bound of the working value of a position (a certain factor start is exactly dead)
__device__ __forceinline__ u32 b_ub(u32 b) { return (b & B_START) ? 0u : (b & B_VAL); }

// An entry of the current level: [15:0] window position   [17:16] state   [18] truncated
enum : u32 { S_UND = 0, S_SEL = 1, S_REJ = 2, S_UNC = 3 };
__device__ __forceinline__ int e_pos(u32 e) { return (int)(e & 0xFFFFu); }
__device__ __forceinline__ u32 e_state(u32 e) { return (e >> 16) & 3u; }
__device__ __forceinline__ u32 e_trunc(u32 e) { return (e >> 18) & 1u; }
// (entries are re-read after other threads changed them: lds_load / lds_store keep the accesses real ds_ instructions)
__device__ __forceinline__ void e_set_state(u32* ent, int i, u32 e, u32 st) { lds_store(&ent[i], (e & ~(3u << 16)) | (st << 16)); }

struct WinScalars { u32 fail; int min_margin; unsigned long long factors; u32 max_entries, pad; unsigned long long prof[16]; };

// optional phase timing (compile with -DTDC_WIN_PROF): thread 0 of every workgroup sums cycle-counter deltas per phase
#ifdef TDC_WIN_PROF
#define WPROF_DECL unsigned long long wp_t = __builtin_readcyclecounter(); unsigned long long wp_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
#define WPROF(i) do { const unsigned long long wp_n = __builtin_readcyclecounter(); wp_acc[i] += wp_n - wp_t; wp_t = wp_n; } while (0)
#define WPROF_CNT(i, v) do { wp_acc[i] += (v); } while (0)
#define WPROF_FLUSH do { if (tid == 0) for (int wi = 0; wi < 12; ++wi) atomicAdd(&sc->prof[wi], wp_acc[wi]); } while (0)
// executions of a code block by wave 0 (whatever lanes are active)
#define WTRIP(i) do { if (tid < 64) { const u64 wt_a = __ballot(true); if (lane == __builtin_ctzll(wt_a)) atomicAdd(&sc->prof[i], 1ull); } } while (0)
#else
#define WPROF_DECL
#define WPROF(i) do {} while (0)
#define WPROF_CNT(i, v) do {} while (0)
#define WPROF_FLUSH do {} while (0)
#define WTRIP(i) do {} while (0)
#endif

// Workgroup barrier that only waits for this wave's LDS traffic (__syncthreads() also drains the vector-memory counter; nothing
// in the level loop is handed between threads through global memory).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 0x80 in every byte of the 4-byte word w whose low 7 bits equal the byte of pat (pat: the level in every byte, bit 7 clear)
__device__ __forceinline__ u32 hit4(u32 w, u32 pat) {
    const u32 x = (w & 0x7F7F7F7Fu) ^ pat;               // a zero byte = a match; bytes are <= 0x7F, so the addition never carries
    return ~(x + 0x7F7F7F7Fu) & 0x80808080u;
}
template <int KT, int TE, int WPE, int WSRC>       // threads, entries per level; WPE = waves per SIMD the register budget is set for; WSRC: 0 no sources written, 1 from a Phi array, 2 SA[ISA[p] - 1] (`phi` is the suffix array then)
__global__ __launch_bounds__(KT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void window_eager_kernel(const u32* __restrict__ cur_g, const u32* __restrict__ prio_g,
                                                            const u8* __restrict__ res_g, const u32* __restrict__ phi, size_t n,
                                                            u32 lcut, u32 threshold, u32 ntiles, u32 halo,
                                                            u32* __restrict__ flen, u8* __restrict__ flen8, u32* __restrict__ fsrc, WinScalars* __restrict__ sc) {
    constexpr int NWV = KT / 64;
    constexpr int CPT = TW / (64 * KT) > 0 ? TW / (64 * KT) : 1;   // 64-position chunks per thread (the first TW / 64 / CPT threads own some)
    static_assert(TW % 64 == 0 && TW / 64 <= KT * CPT, "every chunk of the window needs an owner");
    static_assert(TW <= 65536, "16-bit window positions");
    __shared__ __attribute__((aligned(16))) u8 S[TW];
    __shared__ u32 ent[TE];
    __shared__ u32 pri[TE];               // priority (= ISA) of the natural entries
    __shared__ unsigned short selp[NWV][64];     // per wave: positions of the selected entries of the batch being applied
    __shared__ u32 wtot[NWV];
    __shared__ int s_und[3];
    __shared__ int s_tl, s_tr;
    __shared__ u64 s_lvlmask;             // bit L: some window position holds the value L
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
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
        if (tid == 0) { s_und[0] = 0; s_und[1] = 0; s_und[2] = 0; s_tl = -BIG; s_tr = BIG; s_lvlmask = 0; }
        __syncthreads();
        // ---- load the window: one state byte per position.  cur <= lcut everywhere once the levels above are done; an entry is
        //      natural iff the list it waits in (res: its PLCP value until a push moves it, and pushes write flagged values, see
        //      factorize.hip) is its working value -------------------------------------------------------------------------------
        u64 mymask = 0;
        constexpr int LIT = TW / (KT * 4);                 // 4-position groups per thread
        constexpr int LB = LIT < TDC_WIN_LB ? LIT : TDC_WIN_LB;   // a full window is loaded in batches of LB groups per thread: the loads of a batch are in flight together
        static_assert(TW % (KT * 4) == 0 && LIT % LB == 0, "window load");
        if (wl == TW) {
#pragma unroll 1
            for (int it0 = 0; it0 < LIT; it0 += LB) {
                uint4 cv[LB];
                u32 rv4[LB];
                u64 bmask = 0;
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const size_t gp = w0 + (size_t)((it0 + u) * KT * 4 + tid * 4);      // w0 and the offset are multiples of 4
                    cv[u] = *(const uint4*)(cur_g + gp);
                    rv4[u] = *(const u32*)(res_g + gp);
                }
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int i = (it0 + u) * KT * 4 + tid * 4;
                    const u32 c4[4] = { cv[u].x, cv[u].y, cv[u].z, cv[u].w };
                    u32 sw = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        u32 c = c4[k] > 63u ? 63u : c4[k];
                        if (c < threshold) c = 0;
                        const u32 rv = (rv4[u] >> (8 * k)) & 0xFFu;
                        const u32 bb = c ? (c | (rv != c ? B_TRUNC : 0u)) : 0u;
                        sw |= bb << (8 * k);
                        bmask |= 1ull << c;
                    }
                    *(u32*)&S[PA(i)] = sw;
                }
                mymask |= bmask;
            }
        } else {
            for (int i = tid * 4; i < TW; i += KT * 4) {
                u32 sw = 0;
                const size_t gp = w0 + i;
                if (gp + 4 <= w1) {
                    const uint4 cv = *(const uint4*)(cur_g + gp);                  // w0 and i are multiples of 4
                    const u32 rv4 = *(const u32*)(res_g + gp);
                    const u32 c4[4] = { cv.x, cv.y, cv.z, cv.w };
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        u32 c = c4[k] > 63u ? 63u : c4[k];
                        if (c < threshold) c = 0;
                        const u32 rv = (rv4 >> (8 * k)) & 0xFFu;
                        const u32 bb = c ? (c | (rv != c ? B_TRUNC : 0u)) : 0u;
                        sw |= bb << (8 * k);
                        mymask |= 1ull << c;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (gp + k < w1) {
                            u32 c = cur_g[gp + k];
                            c = c > 63u ? 63u : c;
                            if (c < threshold) c = 0;
                            const u32 rv = res_g[gp + k];
                            const u32 bb = c ? (c | (rv != c ? B_TRUNC : 0u)) : 0u;
                            sw |= bb << (8 * k);
                            mymask |= 1ull << c;
                            }
                    }
                }
                *(u32*)&S[PA(i)] = sw;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mymask |= __shfl_xor(mymask, d, 64);
        if (lane == 0) atomicOr((unsigned long long*)&s_lvlmask, (unsigned long long)mymask);
        // known range [fl, fr) in window offsets.  The outermost lcut - 1 positions of a side that has unknown text behind it only
        // serve as upper bounds of what unknown factors can reach (see "borders" below): nothing outside the window reaches further.
        int fl = (w0 > 0) ? (int)lcut - 1 : -BIG;
        int fr = (w1 < n) ? wl - ((int)lcut - 1) : BIG;
        u32 nsel_interior = 0;
        bool failed = false;
        __syncthreads();
        WPROF(1);

        for (u32 L = lcut; L >= threshold && !failed; --L) {
            const int iL = (int)L;
            // ---- borders: an unknown factor of this level starts at an unknown position q whose working value is still >= L.  Working
            //      values only ever decrease and the window holds an upper bound of them everywhere (only the effects of certain
            //      factors were applied), so the borders move exactly as far as such a q exists: on the left it covers up to q + L - 1,
            //      on the right it cuts down to q - (L - 1).  (Every wave evaluates this for itself: two LDS reads, two ballots.)
            int dfl = fl, dfr = fr;
            if (fl > -BIG / 2) {
                const int q = fl - 1 - lane;
                const u64 mq = __ballot(lane < iL - 1 && q >= 0 && b_ub(S[PA(q >= 0 ? q : 0)]) >= L);
                if (mq) dfl = fl - 1 - __builtin_ctzll(mq) + iL;
            }
            if (fr < BIG / 2) {
                const int q = fr + lane;
                const u64 mq = __ballot(lane < iL - 1 && q < wl && b_ub(S[PA(q < wl ? q : 0)]) >= L);
                if (mq) dfr = fr + __builtin_ctzll(mq) - (iL - 1);
            }
            const u64 lvl_seen = lds_load(&s_lvlmask);
            if (((lvl_seen >> L) & 1ull) == 0) { fl = dfl; fr = dfr; continue; }   // nothing holds the value L
            const int lo = fl > 0 ? fl : 0, hi = fr < wl ? fr : wl;
            // ---- 1. the entries of level L in position order: dense scan of the state bytes (value == L, no factor start) ----------
            u32 alo[CPT], ahi[CPT];
            u32 cnt = 0;
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                const int chunk = tid * CPT + cc;
                const int base = chunk * TCH;
                u32 ml = 0, mh = 0;
                if (base < TW && base < hi && base + TCH > lo) {
                    const u32 pat = L * 0x01010101u;
                    u64 ws[TCH / 8];
#pragma unroll
                    for (int k = 0; k < TCH / 8; ++k) ws[k] = *(const u64*)&S[PW(chunk, k)];
#pragma unroll
                    for (int k = 0; k < TCH / 8; ++k) {
                        // the eight flags of a word -> eight mask bits: two byte dot products (weights 1, 2, 4, 8 and 16 .. 128) give 128 * mask
                        const u32 f = __builtin_amdgcn_udot4(hit4((u32)(ws[k] >> 32), pat), 0x80402010u,
                                                             __builtin_amdgcn_udot4(hit4((u32)ws[k], pat), 0x08040201u, 0u, false), false);
                        const int kk = k & 3;
                        const u32 g = (kk == 0) ? (f >> 7) : (f << (8 * kk - 7));
                        if (k < 4) ml |= g; else mh |= g;
                    }
                    const int rlo = lo - base, rhi = hi - base;                     // known range, relative to the chunk
                    if (rlo > 0) { if (rlo >= 32) { ml = 0; mh &= ~((1u << (rlo - 32)) - 1u); } else ml &= ~((1u << rlo) - 1u); }
                    if (rhi < TCH) { if (rhi <= 32) { mh = 0; ml &= (rhi == 32) ? ~0u : ((1u << rhi) - 1u); } else mh &= (1u << (rhi - 32)) - 1u; }
                }
                alo[cc] = ml; ahi[cc] = mh;
                cnt += (u32)__popc(ml) + (u32)__popc(mh);
            }
            WPROF(2);
            const u32 inc = wave_inclusive_sum(cnt);
            if (lane == 63) wtot[wv] = inc;
            lds_barrier();
            u32 off = inc - cnt, total = 0;
#pragma unroll
            for (int k = 0; k < NWV; ++k) { const u32 t = wtot[k]; if (k < wv) off += t; total += t; }
            if (total > (u32)TE) { failed = true; break; }
            const int m = (int)total;
#ifdef TDC_WIN_PROF
            if (tid == 0) atomicMax(&sc->max_entries, total);
#endif
            if (tid == 0) { s_und[0] = m; s_und[1] = 0; s_und[2] = 0; }
            // (the first two priorities of a thread are requested together; a thread rarely holds more entries of one level.  Measured and
            //  rejected, profiles/r05_window.md: priorities fetched only for natural entries with a natural neighbour, in the first round;
            //  priorities requested one natural level ahead by the chunk's owner)
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                u64 amask = ((u64)ahi[cc] << 32) | alo[cc];
                const int base = (tid * CPT + cc) * TCH;
                int p0 = -1, p1 = -1;
                u32 t0 = 0, t1 = 0, r0 = 0, r1 = 0;
                if (amask) { p0 = base + __builtin_ctzll(amask); amask &= amask - 1; t0 = S[PA(p0)] >> 7; }
                if (amask) { p1 = base + __builtin_ctzll(amask); amask &= amask - 1; t1 = S[PA(p1)] >> 7; }
                if (p0 >= 0 && !t0) r0 = prio_g[w0 + p0];
                if (p1 >= 0 && !t1) r1 = prio_g[w0 + p1];
                if (p0 >= 0) { ent[off] = (u32)p0 | (t0 << 18); pri[off] = r0; ++off; }
                if (p1 >= 0) { ent[off] = (u32)p1 | (t1 << 18); pri[off] = r1; ++off; }
                while (amask) {
                    WTRIP(11);
                    const int pos = base + __builtin_ctzll(amask);
                    amask &= amask - 1;
                    const u32 t = S[PA(pos)] >> 7;
                    ent[off] = (u32)pos | (t << 18);
                    pri[off] = t ? 0u : prio_g[w0 + pos];
                    ++off;
                }
            }
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
                for (int i = lane * NWV + wv; i < m; i += KT) {       // (entries are dealt round-robin to the waves: every phase is a latency chain)
                    WTRIP(12);
                    const u32 e = lds_load(&ent[i]);
                    u32 fL = (i > 0) ? lds_load(&ent[i - 1]) : 0u;               // both direct neighbours are requested up front
                    u32 fR = (i + 1 < m) ? lds_load(&ent[i + 1]) : 0u;
                    if (e_state(e) != S_UND) continue;
                    const int p = e_pos(e);
                    const u32 et = e_trunc(e);
                    bool hit = false, blocked = false, unc = false;
                    const u32 mypri = pri[i];
                    for (int j = i - 1; j >= 0; --j) {
                        WTRIP(13);
                        const u32 f = (j == i - 1) ? fL : lds_load(&ent[j]);
                        if (p - e_pos(f) >= iL) break;
                        const u32 st = e_state(f);
                        if (st == S_SEL) { hit = true; break; }
                        if (st == S_UND || st == S_UNC) {
                            const bool before = e_trunc(f) ? false : (et ? true : pri[j] < mypri);     // natural before truncated, naturals by priority
                            if (before) { if (st == S_UND) blocked = true; else unc = true; }
                        }
                    }
                    for (int j = i + 1; j < m && !hit; ++j) {
                        WTRIP(14);
                        const u32 f = (j == i + 1) ? fR : lds_load(&ent[j]);
                        if (e_pos(f) - p >= iL) break;
                        const u32 st = e_state(f);
                        if (st == S_SEL) { hit = true; break; }
                        if (st == S_UND || st == S_UNC) {
                            const bool before = e_trunc(f) ? false : (et ? true : pri[j] < mypri);
                            if (before) { if (st == S_UND) blocked = true; else unc = true; }
                        }
                    }
                    if (hit) e_set_state(ent, i, e, S_REJ);
                    else if (!blocked) {
                        const bool exposed = (p < dfl) || (p >= dfr);
                        e_set_state(ent, i, e, (unc || exposed) ? S_UNC : S_SEL);
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

            // ---- 3. apply.  G = pow2 >= L lanes share one selected entry.  First the cuts of the L - 1 positions in front of every
            //         selected entry (the ranges of different entries are disjoint) and the taint of the uncertain entries, then --
            //         behind a barrier, a killed position stays dead whatever cut it -- the kills and the start marks ------------
            int G = 1;
            while (G < iL) G <<= 1;
            const int EPB = 64 / G;                          // entries per batch of a wave
            const int gi = lane / G, gj = lane % G;
            u64 newlv = 0;
            for (int pass = 0; pass < 2; ++pass) {
                for (int i0 = 0; i0 < m; i0 += KT) {
                    const int i = i0 + lane * NWV + wv;
                    const bool have = i < m;
                    const u32 e = have ? lds_load(&ent[i]) : 0u;
                    const u32 s = have ? e_state(e) : (u32)S_REJ;
                    const int p = e_pos(e);
                    if (pass == 0 && s == S_UNC) {               // its factor may or may not exist: everything it could touch is unknown
                        if (p < mid) atomicMax(&s_tl, p + iL); else atomicMin(&s_tr, p - (iL - 1));
                    }
                    const u64 selm = __ballot(s == S_SEL);
                    if (!selm) continue;
                    const int nsel = (int)__popcll(selm);
                    if (s == S_SEL) lds_store(&selp[wv][__popcll(selm & ((1ull << lane) - 1ull))], (unsigned short)p);
                    __builtin_amdgcn_wave_barrier();
                    for (int b0 = 0; b0 < nsel; b0 += EPB) {
                        WTRIP(15);
                        const int k = b0 + gi;
                        if (k < nsel && gj < iL) {
                            const int ps = (int)lds_load(&selp[wv][k]);
                            if (pass == 0) {
                                const int d = gj + 1, q = ps - d;                        // cut (:105-109): distance d = 1 .. L - 1
                                if (d < iL && q >= 0) {
                                    const u32 bb = S[PA(q)];
                                    if (!(bb & B_START) && (bb & B_VAL) > (u32)d) {
                                        const bool alive = (u32)d >= threshold;
                                        S[PA(q)] = alive ? (u8)(B_TRUNC | (u32)d) : (u8)0;
                                        if (alive) newlv |= 1ull << d;
                                    }
                                }
                            } else {
                                const int q = ps + gj;                                   // kill (:99-101); the start keeps its length
                                if (q < wl) S[PA(q)] = (gj == 0) ? (u8)(B_START | L) : (u8)0;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (pass == 0) {
                    // (new values only need a bit in the level mask where it does not show them yet)
                    newlv &= ~lvl_seen;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) newlv |= __shfl_xor(newlv, d, 64);
                    if (lane == 0 && newlv) atomicOr((unsigned long long*)&s_lvlmask, (unsigned long long)newlv);
                    lds_barrier();
                }
            }
            lds_barrier();
            WPROF(5);
            int nfl = dfl, nfr = dfr;
            const int tl = lds_load(&s_tl), tr = lds_load(&s_tr);
            if (tl > nfl) nfl = tl;
            if (tr < nfr) nfr = tr;
            fl = nfl; fr = nfr;
            if (tl > -BIG || tr < BIG) {                        // (rare: only then the taint words are reset, behind a barrier of their own)
                lds_barrier();
                if (tid == 0) { s_tl = -BIG; s_tr = BIG; }
            }
            WPROF(6);
        }
        if (tid == 0 && !failed) {                          // smallest distance left between a known-range border and the interior
            int mg = BIG;
            if (fl > -BIG / 2) mg = ia - fl;
            if (fr < BIG / 2 && fr - ib < mg) mg = fr - ib;
            atomicMin(&sc->min_margin, mg);
        }
        if (failed || fl > ia || fr < ib) { if (tid == 0) atomicOr(&sc->fail, failed ? 2u : 1u); }   // 2: the entry list overflowed, 1: known range too small
        else {
            // ---- factor starts of the interior: (pos, Phi[pos], L)  (ArraysComp.hpp:91-96) ---------------------------
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                const int chunk = tid * CPT + cc;
                const int base = chunk * TCH;
                if (base < TW && base < ib && base + TCH > ia) {
#pragma unroll
                    for (int k = 0; k < TCH; k += 8) {
                        const u64 rw = *(const u64*)&S[PW(chunk, k >> 3)];
                        u64 w = rw & 0x4040404040404040ull;
                        while (w) {
                            const int bb = __builtin_ctzll(w) >> 3;
                            w &= w - 1;
                            const int pos = base + k + bb;
                            if (pos < ia || pos >= ib) continue;
                            const size_t gp = w0 + pos;
                            if (flen8) flen8[gp] = (u8)((rw >> (8 * bb)) & 0x3Fu);      // (FactorSpace::flen8: the lengths as bytes)
                            else flen[gp] = (u32)(rw >> (8 * bb)) & 0x3Fu;
                            if constexpr (WSRC == 1) fsrc[gp] = phi[gp];    // (without a Phi array the sources are computed from SA[ISA[p] - 1] where they
                                                                            //  are needed: FactorSpace::src_prio, flatten.hip -- or here, WSRC == 2: an entry
                                                                            //  whose priority is no rank any more was pushed, its source was saved then)
                            else if constexpr (WSRC == 2) { const u32 r = prio_g[gp]; if (r < (u32)n) fsrc[gp] = r ? phi[r - 1] : phi[n - 1]; }
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
__global__ void window_cleanup_kernel(u32* __restrict__ flen, u8* __restrict__ flen8, size_t n, u32 lcut) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (flen8) { if (flen8[p] <= lcut) flen8[p] = 0; }
    else if (flen[p] <= lcut) flen[p] = 0;
}

}  // namespace

u32 window_levels_max_lcut() { return 63; }
u32 window_levels_window() { return (u32)TW; }
u32 window_levels_small_list() { return (u32)TE_SMALL; }
size_t window_levels_min_text() { return (size_t)4 * TW; }

int factorize_window_levels(Ctx& c, size_t n, const u32* cur, const u32* prio, const u8* res8, const u32* phi, u32 lcut, u32 threshold,
                            FactorSpace fs, u64* nfactors, bool start_large, const u32* src_sa) {
    *nfactors = 0;
    if (lcut < threshold) return 0;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    // Attempts: (halo, list size).  The borders of the known range normally move a few hundred positions (they only move where an
    // unknown factor can exist), so the first attempt uses a small halo; a window whose known range shrank into its interior makes
    // the pass retry with the largest one (the worst case of lcut = 63 needs 1 953 + the jumps), an overflowing entry list makes it
    // retry with the large one.
    u32 halo = (u32)c.window_halo & ~3u;
    if (halo < 2 * lcut + 64) halo = (2 * lcut + 64 + 3) & ~3u;
    if (halo > (u32)TH_MAX) halo = TH_MAX;
    bool large = c.window_large_lists != 0 || start_large;
    const u32 max_grid = 1u << 20;
    WinScalars* d_sc = (WinScalars*)c.arena.alloc(sizeof(WinScalars));
    WinScalars h;
    int result = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const size_t ti = (size_t)TW - 2 * (size_t)halo;
        const u32 ntiles = cdiv(n, ti);
        const u32 grid = ntiles < max_grid ? ntiles : max_grid;
        HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(WinScalars), s));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)&d_sc->min_margin, BIG, 1, s));       // (not a copy from pageable memory: that one drains the stream first)
        {
            // per window position: cur (4) + residence (1); per text position: ~0.1 priority reads and the factor output
            Ctx::ProfScope prof(c, K_WINDOW_LEVELS, (u64)((double)n * TW / ti * 5) + (u64)n * 2);
            if (!large) {
                if (phi) window_eager_kernel<TT, TE_SMALL, TDC_WIN_WPE, 1><<<grid, TT, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, fs.flen, fs.flen8, fs.fsrc, d_sc);
                else if (src_sa) window_eager_kernel<TT, TE_SMALL, TDC_WIN_WPE, 2><<<grid, TT, 0, s>>>(cur, prio, res8, src_sa, n, lcut, threshold, ntiles, halo, fs.flen, fs.flen8, fs.fsrc, d_sc);
                else window_eager_kernel<TT, TE_SMALL, TDC_WIN_WPE, 0><<<grid, TT, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, fs.flen, fs.flen8, fs.fsrc, d_sc);
            } else {
                if (phi) window_eager_kernel<TT_LARGE, TE_LARGE, TT_LARGE / 256, 1><<<grid, TT_LARGE, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, fs.flen, fs.flen8, fs.fsrc, d_sc);
                else if (src_sa) window_eager_kernel<TT_LARGE, TE_LARGE, TT_LARGE / 256, 2><<<grid, TT_LARGE, 0, s>>>(cur, prio, res8, src_sa, n, lcut, threshold, ntiles, halo, fs.flen, fs.flen8, fs.fsrc, d_sc);
                else window_eager_kernel<TT_LARGE, TE_LARGE, TT_LARGE / 256, 0><<<grid, TT_LARGE, 0, s>>>(cur, prio, res8, phi, n, lcut, threshold, ntiles, halo, fs.flen, fs.flen8, fs.fsrc, d_sc);
            }
            LAUNCH_CHECK();
        }
        h = c.read(d_sc);
#ifdef TDC_WIN_PROF
        {
            static const char* nm[16] = { "wait_prev", "load", "dense", "scan+write", "select", "apply", "borders", "tail", "levels", "entries", "rounds", "w0 collect3+", "w0 round body", "w0 walk left", "w0 walk right", "w0 apply batch" };
            for (int i = 0; i < 16; ++i) fprintf(stderr, "winprof %-10s %llu\n", nm[i], h.prof[i]);
            fprintf(stderr, "winprof windows %u grid %u attempt %d halo %u min_margin %d fail %u max_entries %u\n", ntiles, grid, attempt,
                    halo, h.min_margin, h.fail, h.max_entries);
        }
#endif
        if (c.level_log) fprintf(stderr, "window pass: attempt %d halo %u lists %s -> fail %u, smallest margin %d\n", attempt, halo, large ? "large" : "small", h.fail, h.min_margin);
        if (c.window_force_fail) h.fail |= 1u;                 // (tests: the pass is discarded as if a border had failed with the largest halo)
        result = (int)h.fail;
        if (!h.fail) break;
        window_cleanup_kernel<<<cdiv(n, 256), 256, 0, s>>>(fs.flen, fs.flen8, n, lcut);     // forget the factors of the failed pass
        LAUNCH_CHECK();
        bool again = false;
        if ((h.fail & 2u) && !large) { large = true; again = true; }
        // (started with the large list on a guess and a border failed: the retry keeps it -- it holds whatever the small one does)
        if ((h.fail & 1u) && halo < (u32)TH_MAX && !c.window_force_fail) { halo = TH_MAX; again = true; }
        if (!again) break;
    }
    c.arena.release(mark);
    if (result == 0) *nfactors = h.factors;
    return result;
}

}  // namespace tdc
