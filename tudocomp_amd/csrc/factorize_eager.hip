// factorize_eager.hip -- lcpcomp::ArraysComp (compressors/lcpcomp/compress/ArraysComp.hpp:36-117): a run of consecutive levels with FEW
// entries each inside ONE launch of ONE workgroup -- texts with long repeats have thousands of such levels (10^9 B of DNA: 4 000), and the
// level loop of factorize.hip pays a launch and a host round trip for every one of them.
//
// The loop can stay on the device because this pass works on the factor SET (see factorize_tiles.hip: the emission order of
// ArraysComp.hpp:82-110 is not observable, truncated entries of one level never meet, naturals precede them) and because of one more
// observation about the entries a selected factor cuts (:105-109).  q + PLCP[q] is non-decreasing in q, and no selected factor can lie
// between a cut position and the factor that cuts it, so a factor at p cuts a CONTIGUOUS run [q_h, p) of alive positions.  The run's
// values p - q fall towards p: its head q_h is decided first, at level p - q_h, and if it is selected it covers the whole run.  The
// tail only matters if the head gets covered by a factor that ends inside the run, at a position e whose predecessor e - 1 was alive
// with cur[e - 1] = cur[e] + 1 when that factor was selected; then e is the new head.  So per selected factor at most TWO entries
// change lists: the head of the run it cuts, and the position behind its end if that continues a run (right-head rule) -- where the
// lazy formulation pushes every position of every cut run down level by level (10^9 B of DNA: 4 140 pushes per level to select 4
// factors).  Model: tests/models/position_space.py::factorize_heads = the oracle's factor set on 12 120 inputs.
//
// The same holds for NATURAL entries: the body of a PLCP ramp (cur[q - 1] = cur[q] + 1) is decided after its predecessor, which covers
// it unless it is itself covered by a factor that ends right in front of q -- and then the right-head rule inserts q.  So the lists
// only ever hold RUN HEADS: alive positions whose predecessor does not continue their run (10^9 B of DNA: 5 000 - 18 000 natural
// candidates per level, a handful of heads).  Model: factorize_heads_only.
//
// Level L: the heads of the level (the segment of heads[] found by a dense pass when the phase started + the blocks inserted since;
// valid iff cur == L; natural iff the residence byte says so) -> sorted by position in LDS, duplicates dropped -> selection rounds
// (natural before truncated, naturals by ISA) -> cuts (a contiguous run; its last position = head), right heads, kills, insertions.  The workgroup gives up -- before it has changed anything of that level -- on a level it cannot hold; the
// host then rebuilds the lists of the lazy formulation from cur[] and continues there (factorize.hip).
#include "stages.hpp"
#include "prim.hpp"
#include "factorize_eager.hpp"

namespace tdc {

namespace {

#ifndef TDC_EAGER_ENT
#define TDC_EAGER_ENT 1024
#endif
constexpr int ENT = TDC_EAGER_ENT;        // threads of the one workgroup
constexpr u32 E_CAP = 4096;               // alive entries of one level
constexpr u32 E_SEL = 1024;               // selected factors of one level
constexpr u32 E_INS = 2 * E_SEL;          // list insertions of one level
constexpr u32 E_BLK_WORDS = 16, E_BLK_POS = 14;   // a block of inserted entries: next (index + 1), count, 14 positions
enum : u32 { ES_UND = 0, ES_SEL = 1, ES_REJ = 2 };

// ascending bitonic sort of a[0 .. np2) in LDS (np2 a power of two; the caller pads with maximal keys)
template <typename T>
__device__ __forceinline__ void lds_bitonic(T* a, u32 np2) {
    for (u32 k = 2; k <= np2; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = threadIdx.x; i < np2; i += ENT) {
                const u32 x = i ^ j;
                if (x > i) {
                    const T u = a[i], v = a[x];
                    const bool up = (i & k) == 0;
                    if ((u > v) == up) { a[i] = v; a[x] = u; }
                }
            }
            __syncthreads();
        }
    }
}

// the same for up to ENT elements by ranking: every thread counts the elements in front of its own (cnt broadcast reads) -- two
// barriers instead of log^2: the levels of a repeat-rich text hold a handful of entries each, and forty barriers of sixteen waves
// were most of a level's 12 microseconds.  tmp: as large as a; the sorted sequence is back in a[] on return.
template <typename T>
__device__ __forceinline__ void lds_ranksort(T* a, T* tmp, u32 cnt) {
    const u32 i = threadIdx.x;
    if (i < cnt) {
        const T k = a[i];
        u32 r = 0;
        for (u32 j = 0; j < cnt; ++j) { const T kj = a[j]; r += (kj < k || (kj == k && j < i)) ? 1u : 0u; }
        tmp[r] = k;
    }
    __syncthreads();
    if (i < cnt) a[i] = tmp[i];
    __syncthreads();
}

__global__ __launch_bounds__(ENT) void eager_levels_kernel(EagerParams P) {
    __shared__ u64 tmp64[ENT];                // rank sorts of up to ENT elements (keys: the first half as 32-bit words)
    __shared__ u32 key[E_CAP];                // position << 1 | truncated, sorted
    __shared__ u32 pri[E_CAP];
    __shared__ u8 st[E_CAP];
    __shared__ u32 sel[E_SEL];                // positions of the selected entries
    __shared__ u32 headq[E_SEL];              // leftmost position each of them cut
    __shared__ u64 ins[E_INS];                // level << 32 | position
    __shared__ u32 s_cnt, s_nsel, s_nins, s_und[2], s_next, s_fail;
    const u32 tid = threadIdx.x;
    const u32 thr = P.threshold;
    u32 L = P.L_from;
    u32 levels_done = 0;
    unsigned long long factors = 0;           // (thread 0 counts)
    u32 status = 0;
    unsigned long long t_lvl = P.dbg ? __builtin_readcyclecounter() : 0ull;
#ifdef TDC_EAGER_PROF
    unsigned long long ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tp = __builtin_readcyclecounter();
#define EPH(k) do { if (tid == 0) { const unsigned long long tn = __builtin_readcyclecounter(); ph[k] += tn - tp; tp = tn; } } while (0)
#else
#define EPH(k) do { } while (0)
#endif
    for (;;) {
        if (L < P.L_stop || L == 0) break;
        const u32 t0 = P.tstart[L], m1 = P.tend[L] - t0;
        u32 hb = P.head[L];
        if (m1 > P.raw_cap) { status = 1; break; }
        if (tid == 0) { s_cnt = 0; s_nsel = 0; s_nins = 0; s_fail = 0; }
        __syncthreads();
        // ---- entries: the heads of this level that still hold the value L (key: position << 1 | truncated) ---------------------------
        auto take = [&](u32 q, u32 r) {                                            // r = res8[q], requested together with cur[q]
            const bool natural = L < 255u ? r == L : r == 255u;                  // (res8_mark, factorize.hip)
            const u32 k = atomicAdd(&s_cnt, 1u);
            if (k < E_CAP) key[k] = (q << 1) | (natural ? 0u : 1u);
        };
        for (u32 i0 = 0; i0 < m1; i0 += ENT * 8) {
            u32 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const u32 i = i0 + (u32)u * ENT + tid; q[u] = i < m1 ? P.tcand[t0 + i] : NONE32; }
            u32 c[8], r8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { c[u] = q[u] != NONE32 ? P.cur[q[u]] : 0u; r8[u] = q[u] != NONE32 ? (u32)P.res8[q[u]] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (q[u] != NONE32 && c[u] == L) take(q[u], r8[u]);
        }
        for (u32 guard = 0; hb != 0 && guard < P.blk_cap; ++guard) {          // blocks inserted since the phase started (newest first)
            const u32* b = P.blk + (size_t)(hb - 1) * E_BLK_WORDS;
            const u32 nx = b[0], bc = b[1];
            const u32 qb = tid < E_BLK_POS ? b[2 + tid] : 0u;                     // (the block is one 64-byte line: link, count and positions in one round trip)
            if (tid < bc && tid < E_BLK_POS) {
                const u32 cq = P.cur[qb], rq = (u32)P.res8[qb];
                if (cq == L) take(qb, rq);
            }
            hb = nx;
        }
        __syncthreads();
        EPH(0);
        const u32 cnt = s_cnt;
        if (cnt > E_CAP) { status = 2; break; }                                // (nothing of this level has been touched)
        if (cnt == 0) {
            // ---- nothing alive here: look at up to ENT levels below at once (inside a long repeat every level holds one erased
            //      candidate: hundreds of thousands of levels with nothing to do) -------------------------------------------------
            ++levels_done;
            if (L == P.L_stop || L == 0) { L = L - 1; break; }
            if (tid == 0) s_next = 0;
            __syncthreads();
            const u32 span = (L - P.L_stop < (u32)ENT) ? L - P.L_stop : (u32)ENT;       // levels L - 1 .. L - span
            if (tid < span) {
                const u32 lv = L - 1 - tid;
                const u32 a0 = P.tstart[lv], am = P.tend[lv] - a0;
                bool alive = P.head[lv] != 0 || am > 4;
                for (u32 i = 0; i < am && i < 4 && !alive; ++i) alive = P.cur[P.tcand[a0 + i]] == lv;
                if (alive) atomicMax(&s_next, lv);
            }
            __syncthreads();
            const u32 nx = s_next;
            // (none of the span levels is alive: the next one to look at lies below all of them -- level L - ENT itself was probed too)
            if (nx) L = nx; else if (span == (u32)ENT) L = L - ENT - 1; else { L = P.L_stop - 1; break; }    // (level 0 never holds anything: threshold >= 1)
            __syncthreads();
            continue;
        }
        // ---- position order, duplicates out (an inserted position may also be a natural candidate: the natural copy sorts first) -------
        if (cnt <= (u32)ENT) lds_ranksort(key, (u32*)tmp64, cnt);
        else {
            u32 np2 = 1;
            while (np2 < cnt) np2 <<= 1;
            for (u32 i = cnt + tid; i < np2; i += ENT) key[i] = 0xFFFFFFFFu;
            __syncthreads();
            lds_bitonic(key, np2);
        }
        for (u32 i = tid; i < cnt; i += ENT) {
            const u32 k = key[i];
            const bool dup = i > 0 && (key[i - 1] >> 1) == (k >> 1);
            st[i] = dup ? (u8)ES_REJ : (u8)ES_UND;
            pri[i] = (k & 1u) ? 0u : P.prio[k >> 1];                                  // (a natural position still carries its ISA)
        }
        if (tid == 0) { s_und[0] = cnt; s_und[1] = 0; }
        __syncthreads();
        EPH(1);
        // ---- selection rounds: lexicographically-first maximal independent set (conflict: distance < L; natural before truncated,
        //      naturals by priority = ISA); an entry decides once every earlier entry within reach has ---------------------------------
        for (u32 round = 0; round <= cnt; ++round) {
            const u32 rd = round & 1u;
            if (s_und[rd] == 0) break;
            __syncthreads();
            if (tid == 0) s_und[rd ^ 1u] = 0;
            __syncthreads();
            for (u32 i = tid; i < cnt; i += ENT) {
                if (st[i] != ES_UND) continue;
                const u32 k = key[i], p = k >> 1, et = k & 1u, mypri = pri[i];
                bool hit = false, blocked = false;
                for (u32 j = i; j-- > 0;) {
                    const u32 f = key[j];
                    if (p - (f >> 1) >= L) break;
                    const u32 s = lds_load(&st[j]);
                    if (s == ES_SEL) { hit = true; break; }
                    if (s == ES_UND && !(f & 1u) && (et || pri[j] < mypri)) blocked = true;
                }
                for (u32 j = i + 1; j < cnt && !hit; ++j) {
                    const u32 f = key[j];
                    if ((f >> 1) - p >= L) break;
                    const u32 s = lds_load(&st[j]);
                    if (s == ES_SEL) { hit = true; break; }
                    if (s == ES_UND && !(f & 1u) && (et || pri[j] < mypri)) blocked = true;
                }
                if (hit) lds_store(&st[i], (u8)ES_REJ);
                else if (!blocked) lds_store(&st[i], (u8)ES_SEL);
                else atomicAdd(&s_und[rd ^ 1u], 1u);
            }
            __syncthreads();
        }
        for (u32 i = tid; i < cnt; i += ENT)
            if (st[i] == ES_SEL) { const u32 k = atomicAdd(&s_nsel, 1u); if (k < E_SEL) { sel[k] = key[i] >> 1; headq[k] = NONE32; } }
        __syncthreads();
        const u32 ns = s_nsel;
        EPH(2);
        if (ns > E_SEL) { status = 4; break; }                                 // (still nothing touched: the host takes the level)
        // ---- cuts (:105-109): distances 1 .. L - 1 in front of every selected entry.  The positions a factor cuts form ONE contiguous run
        //      that ends right in front of it (see the header): a wave walks leftwards in steps of 64 positions and stops at the first
        //      position it does not cut; the last one it did cut is the run's head -----------------------------------------------------
        if (L > 1) {
            const u32 wv = tid >> 6, lane = tid & 63u;
            for (u32 sI = wv; sI < ns; sI += ENT / 64) {
                const u32 p = sel[sI];
                const u32 span = (L - 1 < p) ? L - 1 : p;
                u32 ncut = 0;
                for (u32 d0 = 0; d0 < span; d0 += 64) {
                    const u32 d = d0 + lane + 1;
                    const bool in = d <= span;
                    const u32 q = in ? p - d : 0u;
                    const bool cut = in && P.cur[q] > d;
                    const u64 m = __ballot(cut);
                    const u32 run = (~m == 0ull) ? 64u : (u32)__builtin_ctzll(~m);          // leading lanes that cut
                    if (lane < run) {
                        P.cur[q] = d;
                        if (P.res8) P.res8[q] = (u8)(d <= 63u ? (0x80u | d) : 0u);           // a mark never equals the working value (res8_mark, factorize.hip)
                    }
                    ncut += run;
                    if (run < 64u) break;
                }
                if (lane == 0) headq[sI] = ncut ? p - ncut : NONE32;
            }
        }
        __syncthreads();
        EPH(3);
        // ---- the entries that change lists: the head of the cut run, and the position behind the factor if it continues a run whose
        //      previous position the factor covers (right-head rule) -- read after all cuts, before the kills ---------------------------
        for (u32 s = tid; s < ns; s += ENT) {
            const u32 p = sel[s];
            const u32 h = headq[s];
            if (h != NONE32) {
                const u32 c = P.cur[h];
                if (c >= thr) { const u32 k = atomicAdd(&s_nins, 1u); if (k < E_INS) ins[k] = ((u64)c << 32) | h; }
            }
            const size_t r = (size_t)p + L;
            if (r < P.n) {
                const u32 last = P.cur[r - 1], c = P.cur[r];
                if (c >= thr && last == c + 1) { const u32 k = atomicAdd(&s_nins, 1u); if (k < E_INS) ins[k] = ((u64)c << 32) | (u32)r; }
            }
        }
        __syncthreads();
        EPH(4);
        // ---- kills (:99-101) and the factors (:91-96) ------------------------------------------------------------------------------------
        if ((u64)ns * L < (1ull << 30)) {
            const u32 tot = ns * L;
            for (u32 w = tid; w < tot; w += ENT) {
                const u32 s = w / L, j = w - s * L;
                const size_t q = (size_t)sel[s] + j;
                if (q < P.n) P.cur[q] = 0;
            }
        } else {
            for (u32 s = 0; s < ns; ++s)
                for (u32 j = tid; j < L; j += ENT) { const size_t q = (size_t)sel[s] + j; if (q < P.n) P.cur[q] = 0; }
        }
        for (u32 s = tid; s < ns; s += ENT) {
            const u32 p = sel[s];
            if (P.flen8) { P.flen8[p] = (u8)(L < 255u ? L : 255u); if (L >= 255u) P.flen[p] = L; } else P.flen[p] = L;
            if (P.phi != P.fsrc) P.fsrc[p] = P.phi[p];
        }
        if (tid == 0) factors += ns;
        EPH(5);
        // ---- insertions: sorted by level, one thread per level appends its run to the level's newest block (or opens a new one) -------
        const u32 ni = s_nins;                                                   // (<= 2 ns <= E_INS)
        if (ni) {
            if (ni <= (u32)ENT) lds_ranksort(ins, tmp64, ni);
            else {
                u32 ip2 = 1;
                while (ip2 < ni) ip2 <<= 1;
                for (u32 i = ni + tid; i < ip2; i += ENT) ins[i] = ~0ull;
                __syncthreads();
                lds_bitonic(ins, ip2);
            }
            for (u32 i = tid; i < ni; i += ENT) {
                const u32 lv = (u32)(ins[i] >> 32);
                if (i > 0 && (u32)(ins[i - 1] >> 32) == lv) continue;            // not the first of its run
                u32 hbk = P.head[lv];
                u32 fill = hbk ? P.blk[(size_t)(hbk - 1) * E_BLK_WORDS + 1] : E_BLK_POS;
                for (u32 j = i; j < ni && (u32)(ins[j] >> 32) == lv; ++j) {
                    if (fill == E_BLK_POS) {
                        const u32 nb = atomicAdd(&P.ctl->nblk, 1u);
                        if (nb >= P.blk_cap) { s_fail = 1; break; }
                        u32* b = P.blk + (size_t)nb * E_BLK_WORDS;
                        b[0] = hbk; b[1] = 0;
                        hbk = nb + 1; fill = 0;
                    }
                    u32* b = P.blk + (size_t)(hbk - 1) * E_BLK_WORDS;
                    b[2 + fill] = (u32)ins[j];
                    b[1] = ++fill;
                }
                P.head[lv] = hbk;
            }
        }
        __syncthreads();
        EPH(6);
        ++levels_done;
        if (P.dbg && tid == 0) {                                                 // (TDC_GPU_LEVEL_LOG: entries, factors, cycles of every level)
            const unsigned long long t1 = __builtin_readcyclecounter();
            u32* g = P.dbg + (size_t)L * 4;
            g[0] = cnt; g[1] = ns; g[2] = (u32)(t1 - t_lvl); g[3] = m1;
            t_lvl = t1;
        }
        if (s_fail) { status = 3; L = L - 1; break; }                            // block pool exhausted: this level is done, the entries it could
                                                                                 // not insert are found again when the host rebuilds the lists from cur[]
        if (L == P.L_stop || L == 0) { L = L - 1; break; }
        L = L - 1;
    }
    __syncthreads();
#ifdef TDC_EAGER_PROF
    if (P.dbg && tid == 0) for (int k = 0; k < 8; ++k) P.dbg[k] = (u32)(ph[k] >> 8);     // (phase totals in units of 256 cycles: words of the unused levels 0 and 1)
#endif
    if (tid == 0) {
        P.ctl->level = L;                     // the next level to be processed (L_stop - 1: the phase is complete)
        P.ctl->status = status;
        P.ctl->levels_done = levels_done;     // levels looked at (runs without a head are skipped ENT at a time and count once)
        P.ctl->factors = factors;
    }
}

// Run heads of the levels (lo, hi]: alive positions whose predecessor does not continue their run.  cls[q] = 1 for them.
__global__ __launch_bounds__(256) void eager_heads_class_kernel(const u32* __restrict__ cur, size_t n, u32 lo, u32 hi, u8* __restrict__ cls) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const u32 c = cur[q];
    cls[q] = (c > lo && c <= hi && !(q > 0 && cur[q - 1] == c + 1)) ? 1 : 0;
}

// the lists of the lazy formulation from cur[]: every alive position of the levels (lo, hi] is an entry of list cur[q]
__global__ __launch_bounds__(256) void lazy_rebuild_class_kernel(const u32* __restrict__ cur, size_t n, u32 lo, u32 hi, u8* __restrict__ cls) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const u32 c = cur[q];
    cls[q] = (c > lo && c <= hi) ? 1 : 0;
}
// ... and their list order: truncated entries follow the natural ones (FactorSpace::src_prio).  Priorities are only ever compared inside
// one level, so a truncated entry takes prio_base + its index INSIDE its level's segment of the list (segstart[level]): the rebuild
// consumes as much priority space as its longest level holds entries, not as the whole list does (a text near 2^31 bytes
// with most positions alive used to run out of 32-bit priorities on its second rebuild)
__global__ __launch_bounds__(256) void lazy_rebuild_prio_kernel(const u32* __restrict__ list, size_t m, const u32* __restrict__ cur, const u8* __restrict__ res8,
                                                                size_t n, u32* __restrict__ prio, u32 prio_base, const u32* __restrict__ segstart,
                                                                const u32* __restrict__ src_sa, u32* __restrict__ fsrc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const u32 q = list[i];
    const u32 c = cur[q], r = res8[q];
    const bool natural = c < 255u ? r == c : r == 255u;
    const u32 pr = prio[q];
    if (pr >= (u32)n) return;                                    // pushed by an earlier lazy level: its source was saved then
    // no Phi array: the source of a factor at q is SA[ISA[q] - 1] (ds/PhiFromSA.hpp:35-45) -- saved now for EVERY listed entry, a push
    // of the lazy levels to come may overwrite its priority (= ISA) before it is selected
    if (src_sa) fsrc[q] = pr ? src_sa[pr - 1] : src_sa[n - 1];
    if (!natural) prio[q] = prio_base + ((u32)i - segstart[c]);
}

}  // namespace

u32 eager_levels_raw_cap() { return 32768; }
size_t eager_levels_block_bytes(size_t blocks) { return blocks * E_BLK_WORDS * sizeof(u32); }

void eager_levels_launch(Ctx& c, const EagerParams& P) {
    eager_levels_kernel<<<1, ENT, 0, c.stream>>>(P);
    LAUNCH_CHECK();
}
void eager_heads_class(Ctx& c, const u32* cur, size_t n, u32 lo, u32 hi, u8* cls) {
    eager_heads_class_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(cur, n, lo, hi, cls);
    LAUNCH_CHECK();
}
void lazy_rebuild_class(Ctx& c, const u32* cur, size_t n, u32 lo, u32 hi, u8* cls) {
    lazy_rebuild_class_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(cur, n, lo, hi, cls);
    LAUNCH_CHECK();
}
void lazy_rebuild_prio(Ctx& c, const u32* list, size_t m, const u32* cur, const u8* res8, size_t n, u32* prio, u32 prio_base, const u32* segstart,
                       const u32* src_sa, u32* fsrc) {
    if (!m) return;
    lazy_rebuild_prio_kernel<<<cdiv(m, 256), 256, 0, c.stream>>>(list, m, cur, res8, n, prio, prio_base, segstart, src_sa, fsrc);
    LAUNCH_CHECK();
}

}  // namespace tdc
