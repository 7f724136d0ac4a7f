// arith.hip -- ArithmeticCoder::Encoder (coders/ArithmeticCoder.hpp:35-177) as the literal coder of the lzss token
// stream (BASELINE.json configs[2]: LCPCompressor + ArithmeticCoder; compress side only).
//
// Reference: a static model (normalised cumulative counts C[], :72-92) and a 64-bit interval [lower, upper).  Before a
// literal is coded, if upper - lower < min_range the encoder writes `lower` as a 64-bit word INTO THE SHARED BIT STREAM
// and resets the interval (:96-104); after the literal_count-th literal it writes `lower` and an all-ones word
// (:151-155, :169-176).  So a literal contributes 0, 64, or 64/128 more bits to the stream, at its own position.
//
// Device formulation.  The interval WIDTH evolves independently of `lower`, and a flush resets everything, so
//   nf(k) := the literal at which the first flush happens when the interval is reset at literal k
// is a function of the literals alone.  It is evaluated for every k in parallel (a few dozen steps each: a flush comes
// every 64 / H literals); the real flush positions are the orbit of literal 0 under nf (mark_orbit_u32, shared with the
// lzss_lcp parse); the flushed words are then recomputed per segment.  A degenerate model (a segment longer than
// ARITH_STEP_CAP literals) falls back to one sequential pass on the device.
#include "stages.hpp"
#include "prim.hpp"
#include "huffman_host.hpp"
#include "arith.hpp"

namespace tdc {

// ---- host: model + code book ---------------------------------------------------------------------------------------
bool arith_build_model(const u32 hist[256], ArithModel* m, HostBitWriter& hw) {
    u32 c[256];
    for (int i = 0; i < 256; ++i) c[i] = hist[i];
    u8 codebook_size = 0;                                        // uliteral_t: wraps at 256 like the reference
    if (c[0] != 0u) codebook_size++;                             // build_intervals :73-75
    u32 mn = 0xFFFFFFFFu;
    for (int i = 1; i <= 255; ++i) {                             // :78-84
        if (c[i] != 0u) { codebook_size++; if (c[i] < mn) mn = c[i]; }
        c[i] = c[i] + c[i - 1];
    }
    m->literal_count = c[254];                                   // :85 (cumulative count up to byte 254)
    for (int i = 0; i <= 255; ++i) c[i] = c[i] / mn;             // :88-90
    m->min_range = c[254];                                       // :91
    m->tot = c[255];
    for (int i = 0; i < 256; ++i) m->C[i] = c[i];
    hw.write_int(m->literal_count, 32);                          // writeCodebook :128-143
    hw.write_int(codebook_size, 8);
    if (c[0] != 0u) { hw.write_int(0, 8); hw.write_int(c[0], 32); }
    for (int i = 1; i <= 255; ++i) if (c[i] != c[i - 1]) { hw.write_int((u64)i, 8); hw.write_int(c[i], 32); }
    return m->tot != 0;                                          // tot == 0: the reference divides by zero (:110-113)
}

// ---- device ----------------------------------------------------------------------------------------------------------
constexpr u32 ARITH_STEP_CAP = 4096;

struct ArithDevModel { u32 C[256]; };

// setNewBounds without the flush check (:106-116)
__device__ __forceinline__ void arith_step(u64& lb, u64& ub, u32 v, const u32* __restrict__ C, u64 tot) {
    const u64 range = ub - lb;
    const u64 cv = C[v], cp = v ? C[v - 1] : 0;
    u64 offu, offl;
    if (range <= tot) { offu = (range * cv) / tot; offl = (range * cp) / tot; }
    else { const u64 q = range / tot; offu = q * cv; offl = q * cp; }
    ub = lb + offu;
    lb = lb + offl;                                              // v == 0: offl == 0 (the reference skips the update)
}

__global__ void lit_flag_kernel(const u32* __restrict__ owner, size_t n, u32* __restrict__ flag) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) flag[p] = (owner[p] == NONE32) ? 1u : 0u;
}
__global__ void lit_scatter_kernel(const u8* __restrict__ text, const u32* __restrict__ owner, const u32* __restrict__ litidx, size_t n,
                                   u8* __restrict__ lits) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && owner[p] == NONE32) lits[litidx[p]] = text[p];
}

// next[k] = first literal after k at which the encoder would flush if its interval were reset at k (or nlit)
__global__ __launch_bounds__(256) void arith_next_flush_kernel(const u8* __restrict__ lits, u32 nlit, ArithDevModel mdl, u64 tot,
                                                                u64 min_range, u32* __restrict__ next, u32* __restrict__ overflow) {
    __shared__ u32 C[256];
    C[threadIdx.x] = mdl.C[threadIdx.x];
    __syncthreads();
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nlit) return;
    u64 lb = 0, ub = ~0ull;
    u32 j = k, steps = 0;
    for (;;) {
        arith_step(lb, ub, lits[j], C, tot);
        ++j;
        if (j >= nlit) break;
        if (ub - lb < min_range) break;                          // literal j starts with a flush
        if (++steps > ARITH_STEP_CAP) { atomicOr(overflow, 1u); break; }
    }
    next[k] = j;
}

// one thread per segment start s (marked): the word flushed at next[s] and, if the literal_count-th literal lies in the
// segment, `lower` right after it
__global__ __launch_bounds__(256) void arith_segment_values_kernel(const u8* __restrict__ lits, u32 nlit, ArithDevModel mdl, u64 tot,
                                                                    const u32* __restrict__ next, const u8* __restrict__ mark,
                                                                    u32 lc_index, u64* __restrict__ fval, u64* __restrict__ pp_lb) {
    __shared__ u32 C[256];
    C[threadIdx.x] = mdl.C[threadIdx.x];
    __syncthreads();
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nlit || !mark[s]) return;
    const u32 e = next[s];
    u64 lb = 0, ub = ~0ull;
    for (u32 j = s; j < e; ++j) {
        arith_step(lb, ub, lits[j], C, tot);
        if (j == lc_index) *pp_lb = lb;
    }
    if (e < nlit) fval[e] = lb;
}

// fallback: the reference's loop, one thread (degenerate models with very long segments)
__global__ void arith_sequential_kernel(const u8* __restrict__ lits, u32 nlit, ArithDevModel mdl, u64 tot, u64 min_range, u32 lc_index,
                                        u8* __restrict__ mark, u64* __restrict__ fval, u64* __restrict__ pp_lb) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u64 lb = 0, ub = ~0ull;
    for (u32 j = 0; j < nlit; ++j) {
        if (ub - lb < min_range) { mark[j] = 1; fval[j] = lb; lb = 0; ub = ~0ull; }
        arith_step(lb, ub, lits[j], mdl.C, tot);
        if (j == lc_index) *pp_lb = lb;
    }
}
__global__ void arith_clear_first_kernel(u8* mark) { if (threadIdx.x == 0 && blockIdx.x == 0) mark[0] = 0; }

void arith_prepare(Ctx& c, const u8* text, size_t n, const u32* owner, const ArithModel& m, ArithPlan* plan) {
    hipStream_t s = c.stream;
    const unsigned gn = cdiv(n, 256);
    u32* litidx = c.arena.get<u32>(n);
    u32* d_cnt = c.arena.get<u32>(2);
    lit_flag_kernel<<<gn, 256, 0, s>>>(owner, n, litidx);
    LAUNCH_CHECK();
    exclusive_sum_u32(c, litidx, litidx, n, d_cnt);
    const u32 nlit = c.read(d_cnt);
    u8* lits = c.arena.get<u8>((size_t)nlit + 8);
    lit_scatter_kernel<<<gn, 256, 0, s>>>(text, owner, litidx, n, lits);
    LAUNCH_CHECK();
    u8* mark = c.arena.get<u8>((size_t)nlit + 8);
    u64* fval = c.arena.get<u64>((size_t)nlit + 1);
    u64* d_pp = c.arena.get<u64>(1);
    u32* next = c.arena.get<u32>((size_t)nlit + 1);
    u32* sc1 = c.arena.get<u32>((size_t)nlit + 1), *sc2 = c.arena.get<u32>((size_t)nlit + 1);
    HIP_TRY(hipMemsetAsync(d_cnt + 1, 0, sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(d_pp, 0, sizeof(u64), s));
    ArithDevModel dm;
    for (int i = 0; i < 256; ++i) dm.C[i] = m.C[i];
    const u32 lc_index = m.literal_count - 1;                    // literal_count >= 1: the sentinel is always a literal
    const unsigned gl = cdiv(nlit, 256);
    arith_next_flush_kernel<<<gl, 256, 0, s>>>(lits, nlit, dm, m.tot, m.min_range, next, d_cnt + 1);
    LAUNCH_CHECK();
    const u32 overflow = c.read(d_cnt + 1);
    if (!overflow) {
        mark_orbit_u32(c, next, nlit, mark, sc1, sc2);
        arith_segment_values_kernel<<<gl, 256, 0, s>>>(lits, nlit, dm, m.tot, next, mark, lc_index, fval, d_pp);
        LAUNCH_CHECK();
        arith_clear_first_kernel<<<1, 64, 0, s>>>(mark);         // literal 0 starts the first segment but is not a flush
        LAUNCH_CHECK();
    } else {
        HIP_TRY(hipMemsetAsync(mark, 0, nlit, s));
        arith_sequential_kernel<<<1, 64, 0, s>>>(lits, nlit, dm, m.tot, m.min_range, lc_index, mark, fval, d_pp);
        LAUNCH_CHECK();
    }
    plan->litidx = litidx; plan->amark = mark; plan->fval = fval; plan->lc_index = lc_index;
    plan->pp_lb = c.read(d_pp);
    plan->nlit = nlit;
    plan->sequential_fallback = overflow != 0;
}

}  // namespace tdc
