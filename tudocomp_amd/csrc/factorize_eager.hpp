// factorize_eager.hpp -- a run of consecutive small ArraysComp levels inside one launch (factorize_eager.hip), and the conversions
// between the eager and the lazy list formulation of factorize.hip (both rebuild their lists from cur[] and the residence marks).
#pragma once
#include "stages.hpp"

namespace tdc {

struct EagerCtl { u32 level; u32 status; u32 nblk; u32 levels_done; unsigned long long factors; };
struct EagerParams {
    const u32* tcand; const u32* tstart; const u32* tend;        // run heads that existed when the phase started: positions by level, [tstart, tend) per level
    u32* head;                                                   // per level: index + 1 of the newest block of entries inserted since (0: none)
    u32* blk; u32 blk_cap;                                       // blocks of 16 words: next (index + 1), count, 14 positions
    u32* cur; const u32* prio; const u32* phi; u32* flen; u8* flen8; u32* fsrc; u8* res8;     // flen8: FactorSpace::flen8 (nullable)
    size_t n; u32 threshold;
    u32 L_from, L_stop;                                          // levels L_from .. L_stop (inclusive), downwards
    u32 raw_cap;                                                 // longest head segment the workgroup reads
    u32* dbg;                                                    // optional (debugging): 4 words per level -- entries, factors, cycles, candidates
    EagerCtl* ctl;                                               // zeroed by the caller; out: next level (L_stop - 1: complete), status (0 ok, 1 a
                                                                 // segment above raw_cap, 2 more than 4096 entries alive, 3 block pool exhausted, 4 more
                                                                 // than 1024 factors in one level), levels processed, factors selected
};
u32 eager_levels_raw_cap();
size_t eager_levels_block_bytes(size_t blocks);
void eager_levels_launch(Ctx& c, const EagerParams& P);
// cls[q] = 1 for the run heads of the levels (lo, hi] (alive, predecessor not alive with the value cur[q] + 1)
void eager_heads_class(Ctx& c, const u32* cur, size_t n, u32 lo, u32 hi, u8* cls);
// cls[q] = 1 for every alive position of the levels (lo, hi]
void lazy_rebuild_class(Ctx& c, const u32* cur, size_t n, u32 lo, u32 hi, u8* cls);
// list[0 .. m): truncated entries that still carry their ISA as priority get prio_base + index (their source is saved first if src_sa)
void lazy_rebuild_prio(Ctx& c, const u32* list, size_t m, const u32* cur, const u8* res8, size_t n, u32* prio, u32 prio_base, const u32* segstart, const u32* src_sa, u32* fsrc);

}  // namespace tdc
