// factorize_tiles.hpp -- window-local evaluation of the low ArraysComp levels (factorize_tiles.hip).
#pragma once
#include "stages.hpp"

namespace tdc {

// Levels lcut .. threshold of the factorization, every text window inside one launch.  Inputs are the global state after
// the levels above lcut: cur (working LCP per position), prio (list order), res8 (level whose list holds the entry of a
// position; 0 = none; values above lcut are stale).  Nothing of it is modified.  Writes flen / fsrc at the factor
// starts and returns 0; returns a non-zero reason mask (flen cleaned of the low-level factors again) if some window could
// not be completed -- bit 0: the known range shrank into an interior (a lower lcut may still work), bit 1: a level had
// more entries than even the large LDS lists hold (the pass retries by itself with the large lists when the small ones
// overflow) -- in which case the caller runs the global level loop.
// phi may be NULL: fsrc[] is then left alone at the factor starts of this pass -- their sources are computed where they are needed
// (FactorSpace::src_prio: SA[ISA[p] - 1]; prio[p] is still ISA[p] unless a push at a global level overwrote it, and then fsrc[p] was saved).
int factorize_window_levels(Ctx& c, size_t n, const u32* cur, const u32* prio, const u8* res8, const u32* phi, u32 lcut, u32 threshold,
                             FactorSpace fs, u64* nfactors, bool start_large = false, const u32* src_sa = nullptr);   // src_sa (with phi == NULL): the pass writes fsrc[p] = SA[ISA[p] - 1] itself
u32 window_levels_window();         // positions per window
u32 window_levels_small_list();     // entries of one level the small per-level lists hold
u32 window_levels_max_lcut();       // one presence bit per level in a 64-bit mask
size_t window_levels_min_text();    // shorter texts stay on the global path

}  // namespace tdc
