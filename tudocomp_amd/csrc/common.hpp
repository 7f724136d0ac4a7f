// common.hpp -- shared host/device helpers for the MI355X (gfx950) lcpcomp pipeline.
// Wave size is 64 on CDNA4; every wave-level idiom below is written for 64 lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

namespace tdc {

typedef uint8_t  u8;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint64_t u64;

constexpr u32 NONE32 = 0xFFFFFFFFu;
constexpr int WAVE = 64;

// Error codes of the C ABI (include/tdc_gpu.h)
enum {
    TDC_OK = 0,
    TDC_ERR_HIP = -1,          // a HIP runtime call failed (no device, OOM, launch failure)
    TDC_ERR_ARG = -2,          // bad argument
    TDC_ERR_NO_SENTINEL = -3,  // text does not end with the unique 0 (ds/TextDS.hpp:132-138)
    TDC_ERR_TOO_LARGE = -4,    // n >= 2^31 (reference limit, 32-bit len_t)
    TDC_ERR_OOM = -5,          // device arena too small
    TDC_ERR_UNSUPPORTED = -6,  // coder / option not built
    TDC_ERR_INTERNAL = -7,     // invariant violated (bug)
};

struct HipError { hipError_t e; const char* file; int line; };

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) throw ::tdc::HipError{_e, __FILE__, __LINE__};           \
    } while (0)

inline unsigned bits_for(u64 v) {   // util.hpp:194 of the reference: bits_for(0) == 1
    unsigned b = 0;
    if (v == 0) return 1;
    while (v) { ++b; v >>= 1; }
    return b;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// Device arena: one hipMalloc per context, bump allocation with mark/release.  All stages carve their
// position-space arrays out of it, so a compress call performs no hipMalloc/hipFree in steady state.
// ------------------------------------------------------------------------------------------------
struct Arena {
    char* base = nullptr;
    size_t size = 0;
    size_t top = 0;
    size_t high = 0;

    size_t top_hi = 0;          // bytes reserved at the END of the arena (alloc_top): scratch that must outlive bump-allocated neighbours

    void* alloc(size_t bytes) {
        size_t off = align_up(top, 256);
        if (off + bytes > size - top_hi) throw HipError{hipErrorOutOfMemory, "arena", (int)__LINE__};
        top = off + bytes;
        if (top > high) high = top;
        return base + off;
    }
    template <typename T> T* get(size_t count) { return (T*)alloc(count * sizeof(T)); }
    void* alloc_top(size_t bytes) {
        if (bytes + top_hi + 256 > size) throw HipError{hipErrorOutOfMemory, "arena (top)", (int)__LINE__};
        const size_t off = (size - top_hi - bytes) & ~(size_t)255;
        if (off < top) throw HipError{hipErrorOutOfMemory, "arena (top)", (int)__LINE__};
        top_hi = size - off;
        return base + off;
    }
    template <typename T> T* get_top(size_t count) { return (T*)alloc_top(count * sizeof(T)); }
    void release_top() { top_hi = 0; }
    size_t mark() const { return top; }
    void release(size_t m) { top = m; }
};

// Live per-kernel timing (bench.py's roofline): HIP events on the launching stream around every launch of the
// instrumented kernels, resolved after the stream has been synchronised.
enum KernelClass {
    K_RS_SCATTER_U64 = 0, K_RS_SCATTER_U32, K_RS_COUNT, K_SCAN,
    K_SA_RANK_SCATTER, K_SA_BUILD_KEYS, K_PHI, K_PLCP, K_CAND,
    K_LEVEL_INIT, K_MIS_ROUND, K_RESOLVE, K_PUSH, K_APPLY, K_POOL, K_SMALL_LEVEL, K_WINDOW_LEVELS,
    K_FLATTEN_ROUND, K_ENC_GAPS, K_ENC_HIST, K_ENC_TILE_BITS, K_ENC_PACK, K_EXTRACT, K_SS_LEAF, K_SA_LOCAL_SORT, K_WINDOW_SCATTER,
    K_WS_LEAF_SORT, K_WS_LEAF_COUNT, K_WS_RUN, K_FS_IMAGE,
    K_CLASS_COUNT
};

struct KernelProfile {
    double ms = 0;        // summed launch durations
    u64 launches = 0;
    u64 bytes = 0;        // summed algorithmic bytes
};

#ifdef __HIPCC__
// Small device -> host read-backs without a stream synchronisation: the words are stored straight into mapped host memory,
// followed by a sequence number the host spins on (a hipMemcpyAsync + hipStreamSynchronize pair costs several times the
// latency, and texts with long repeats need thousands of them).
static __global__ void publish_words_kernel(const uint32_t* __restrict__ src, uint32_t nwords, uint32_t* host_dst, uint32_t* host_flag,
                                            uint32_t seq) {
    for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x)
        __hip_atomic_store(&host_dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Arena arena;
    hipEvent_t ev[16] = {};
    void* pinned = nullptr;        // small pinned staging block for device->host scalars
    size_t pinned_size = 0;
    // page-locked scratch owned by the context (round 6: a hipHostMalloc / hipHostFree pair per call and every small copy from pageable
    // memory showed up as 0.3 - 0.8 ms of idle device time in the kernel trace of a step):
    void* pinned_tab = nullptr;    // the factorizer's gather-segment table (mapped: small levels read it in place); grown on demand
    size_t pinned_tab_size = 0;
    u8* pinned_hdr = nullptr;      // stream headers on their way to the device (64 KiB; rewritten only after the call's final synchronisation)
    static constexpr size_t PINNED_HDR = 65536;
    void* pinned_table(size_t bytes) {
        if (pinned_tab_size < bytes) {
            if (pinned_tab) { (void)hipHostFree(pinned_tab); pinned_tab = nullptr; pinned_tab_size = 0; }
            HIP_TRY(hipHostMalloc(&pinned_tab, bytes, hipHostMallocDefault));
            pinned_tab_size = bytes;
        }
        return pinned_tab;
    }
    static constexpr u32 ZC_SEG_OFF = 1040, ZC_WORDS = 1040 + 16384;  // words 1040 ..: up to 8192 (target, start) pairs of a one-workgroup level
    static constexpr u32 ZC_BLOCKS = 3;                                // block 0: read() / publish_*; blocks 1, 2: two one-workgroup levels in flight
    u32* zc_host = nullptr;        // mapped host block for publish_words_kernel: 1024 data words + the sequence word (+ the segment area)
    u32* zc_dev = nullptr;         // the same block as seen from the device
    u32 zc_seq = 0;
    int fast_read = 1;             // env TDC_GPU_FASTREAD=0: read-backs through hipMemcpyAsync + hipStreamSynchronize
    u32* d_err = nullptr;          // device error word (look-back timeouts etc.), checked at the end of every API call
    int sa_local_sort = 1;         // doubling rounds: sort whole runs inside 2048-element tiles locally (env TDC_GPU_SA_LOCAL=0 disables)
    int radix_waves = 4;           // waves per radix-sort workgroup for large inputs (env TDC_GPU_RADIX_WAVES = 4 | 8; no measurable difference)
    int plcp_samples = 1;          // PLCP: exact values at every 256th position first, as lower bounds for the chunks (env TDC_GPU_PLCP_SAMPLES=0: chunks start from 0)
    bool huff_ok = true;           // the start-up self-check of the host Huffman table passed (else coder=huff calls fail with TDC_GPU_ERR_INTERNAL)
    int flen_bytes = 1;            // the metric's path: factor lengths as bytes until build_owner (FactorSpace::flen8; env TDC_GPU_FLEN_BYTES=0: the dense u32 array)
    int eager_levels = 1;          // factorize: runs of small levels inside one launch (factorize_eager.hip; env TDC_GPU_EAGER=0: every level through the lazy loop)
    int level_purge = 1;           // factorize: the lists of the next 64 levels are purged in place before they are read (env TDC_GPU_LEVEL_PURGE=0 disables)
    int enc_rec = 1;               // with enc_early: the pack reads lengths and flattened sources from the records of the flatten stage (env TDC_GPU_ENC_REC=0: from flen[] / fsrc[])
    int owner_rem = 8;             // metric's path: owner words carry how far their factor still reaches, in at most this many bits (FactorSpace::owner_rem_bits; 0: plain ranks)
    int enc_early = 1;             // first half of the Huffman encoder next to the first flatten round (texts of 1 MiB and more; env TDC_GPU_ENC_EARLY=0: after the flatten stage, 2: for every text)
    int fs_pair = 1;               // fused scatter: two rows per workgroup (tiles of 8192 records; env TDC_GPU_FS_PAIR=0: one)
    int small_big = 1;             // factorize: one-workgroup levels with up to 4096 survivors run on a 512-thread instance of the kernel (env TDC_GPU_SMALL_BIG=0: multi-launch path above 2048)
    int small_pipeline = 1;        // factorize: the kernel of the next one-workgroup level is queued while the current one runs (env TDC_GPU_SMALL_PIPELINE=0 disables)
    int sa_pairs = 1;              // suffix array, doubling fall-back of the wide path: groups of exactly two suffixes are ordered from the text in one pass first (env TDC_GPU_SA_PAIRS=0 disables)
    int sa_stars = 1;              // suffix array, doubling fall-back of the wide path: every group is ordered against its smallest member from the text in one pass first (round 6; option sa_stars=0: the pair step)
    int sa_refine = 1;             // suffix array: small groups of the initial order are refined from the text before the first round (env TDC_GPU_SA_REFINE=0 disables)
    int sa_fused_init = 1;         // suffix array: pass 0 of the initial sort computes its keys from the text (env TDC_GPU_SA_FUSED_INIT=0: separate key kernel)
    int radix_lds = 1;             // radix scatter: reorder the tile in LDS before writing: 0 never, 1 always, 2 for 32-bit keys only
                                   // (measured: -33 % for u32 pairs; u64 pairs only gain together with xcd_remap: 21.1 -> 18.5 ms; env TDC_GPU_RADIX_LDS)
    int xcd_remap = 1;             // XCD-contiguous tile walk (env TDC_GPU_XCD_REMAP): 0 = only in the final pass of the bucketed scatter
                                   // (-1.2 ms of 7.3), 1 = also in the radix sort kernels (with the LDS-reordered scatter the 128-byte
                                   // pieces of neighbouring tiles then meet in one L2: -13 %), 2 = nowhere
    int window_src = 1;            // window pass without a Phi array: the kernel writes fsrc[p] = SA[ISA[p] - 1] at its factor starts itself -- its VALU-bound waves hide the two gathers that cost flatten_init 3.7 ms (window kernel + 1.1 ms: -2.5 ms per step; option window_src=0: the sources are computed in flatten_init)
    int window_lcut = 56;          // factorize: levels <= this run window-local in one launch (option window_lcut, 0 disables; factorize stage in the kernel trace of a step at 2e9 B: 40 / 44 / 48 / 52 / 56 / 60 / 63 -> 36.1 / 35.3 / 33.7 / 33.2 / 33.3 / 33.2 / 33.3 ms)
    int window_halo = 320;         // window pass: halo of the first attempt (option window_halo; a failed border retries with 2048.  With window_lcut 56 at 2e9 B: 256 / 320 / 384 -> factorize 32.8 / 33.1 / 33.3 ms with a smallest margin of 64 / 126 / 185 positions left: 320 keeps 40 % of the halo in reserve)
    size_t dec_seg = 0;            // decompression: bit positions per segment of the chain marking (0: 2^30; env TDC_GPU_DEC_SEG, tests)
    bool phi_lazy = true;          // lcpcomp(comp=arrays) behind the fused scatter: no Phi array, a factor's source is SA[ISA[p] - 1] (env TDC_GPU_PHI_LAZY=0: Phi as before)
    int dec_lean = 1;              // decompression, device parse: lean marking for streams of short tokens (env TDC_GPU_DEC_LEAN=0: the general marking for every stream; tests)
    int dec_parse = 1;             // decompression: token stream parsed on the device (env TDC_GPU_DEC_PARSE: 0 host parse, 1 streams >= 1 MiB, 2 always)
    int window_force_fail = 0;     // tests (env TDC_GPU_WINDOW_FORCE_FAIL=1): every window pass is reported as failed -> the global level loop takes the low levels
    int window_large_lists = 0;    // window pass: start with the large per-level lists (env TDC_GPU_WINDOW_LARGE=1; tests)
    int ssort = 1;                 // suffix array: splitter-partition sort (ssort.hip) instead of the 8-pass LSD sort for large inputs (env TDC_GPU_SSORT=0 disables)
    int ssort_levels = 0;          // force the number of partition levels of the splitter sort (env TDC_GPU_SSORT_LEVELS = 1..3; tests)
    int wsort = 1;                 // suffix array: wide-key (bit-packed, 2 x 64 bits) initial sort + text rounds + fused ISA / Phi / PLCP scatter (env TDC_GPU_WSORT=0: round-2 path)
    size_t wsort_min = (size_t)1 << 20;   // smallest text that takes it (env TDC_GPU_WSORT_MIN, >= 4096; tests)
    int wsort_syms = 0;            // cap on the symbols per key of the wide sort (env TDC_GPU_WSORT_SYMS; 0: as many as the key words hold) -- measurements only
    int wsort_kw = 0;              // key words: 0 = by alphabet (2 when a word holds fewer than 16 symbols), 1 | 2 forced (env TDC_GPU_WSORT_KW)
    int wsort_rounds = 24;         // most text rounds before the doubling fallback (env TDC_GPU_WSORT_ROUNDS; 0: straight to doubling)
    int wsort_cmax = 24;           // leaf sort: runs of tying records up to this length are ordered by counting, longer ones by the wave kernel (option wsort_cmax, 8 .. 64; leaf stage in the trace with wsort_pack 1024: 16 / 24 / 32 -> 49.4 / 49.2 / 49.7 ms)
    int wsort_overlap = 1;         // host-buffer calls: level 1 of the wide suffix sort runs chunk by chunk behind the upload (env TDC_GPU_WSORT_OVERLAP=0 disables)
    int wsort_prehist = 1;         // ... and the tile histograms of that pass are copied into the merging level's table, whose count pass skips those pieces (option wsort_prehist)
    bool wsort_count_nonheads = true;   // (transient, set by the caller of the wide sort) false: WSortStats::nonheads is not wanted -- the pass over the flags that counts it is skipped
    int sa_seg_bigcap = 65536;     // ... with room for this many groups of more than 1 024 members (more: the round sorts its list as a whole; tests: 1)
    int sa_seg_rounds = 2;         // text rounds: the groups are ordered in place (0: the list is sorted as a whole; 1: in place, behind a key pass of its own; 2: the in-place pass makes the records itself: 105.8 against 106.5 ms of suffix array; option sa_seg_rounds)
    int wsort_run_streams = 1;     // the four run kernels of a leaf stage side by side on the context's three streams (option wsort_run_streams)
    int wsort_predig_skip = 4;     // ... but not for the last chunks (their digits would only be ready after the last copy; option wsort_predig_skip; 0 / 2 / 3 / 4 / 6: 257.5 / 257.5 / 257.4 / 257.2 / 257.0 ms against 260.8 without; with the count pass that requests a row ahead: 4 against 6 = 247.1 against 247.7 ms, three alternating pairs)
    int wsort_predig = 1;          // with the overlapped level 1: the digits of level 2 are computed chunk by chunk behind the upload too, on the low-priority side stream (option wsort_predig; round 4 measured it on the main stream, where it lengthened the upload by more than it saved)
    int wsort_two = 0;             // wide sort of more than 235 M records: two partition levels of up to 1024 buckets instead of three of up to 256 (env TDC_GPU_WSORT_TWO)
    int wsort_leaf = 2048;         // leaf size the three-level wide sort aims at (env TDC_GPU_WSORT_LEAF: 1024 | 2048)
    int wsort_pack = 1024;         // leaf sort: leaves up to this size are packed into units of at most twice that (option wsort_pack: 1024 | 2048 | 4096; leaf stage in the kernel trace of a step at 2e9 B: 1024 -> 49.4 ms, 2048 -> 50.5 ms in six runs each -- with 1024 every unit fits a four-wave instance; round 6's A/B of whole steps had put the difference at 0.3 ms, inside its noise)
    int wsort_order = 1;           // wide sort with three levels: 1 = the widest level (256 buckets) first, where it runs behind the upload (env TDC_GPU_WSORT_ORDER)
    int wsort_fuse = 1;            // leaf sort: kernel A orders the short runs of its units itself instead of listing the unit for the counting kernel (env TDC_GPU_WSORT_FUSE)
    int wsort_small = 0;           // tests: 1 = every run of a leaf unit counts as "big" (the chunk iterations run everywhere) (env TDC_GPU_WSORT_SMALLRUN)
    int msd_partition = 1;         // bucketed scatter: MSD partition with atomic slots instead of two stable LSD passes (env TDC_GPU_MSD_PARTITION=0)
    int bucket_scatter = 1;        // big random scatters (rank, Phi) go through one radix partition by destination window (env TDC_GPU_BUCKET_SCATTER=0 disables)
    // (the options below used to be read from the environment wherever they were used; since round 6 every option is a field that only
    //  tdc_gpu_ctx_set_option() writes -- api.hip, one table -- and the shipped library never looks at TDC_GPU_* variables unless
    //  TDC_GPU_DEBUG_KNOBS=1 asks tdc_gpu_ctx_create() to apply them through that same function)
    int upload_tail_n = 3, upload_tail_pct = 60;   // the last upload_tail_n chunks of the overlapped upload shrink by this factor each (0.6, 0.36, 0.22: what is left behind the last copy is the device work of a small chunk)
    int upload_chunks = 16;        // chunks of the overlapped upload (4 .. 24: ev_copy[16 ..]; option upload_chunks)
    int flatten_steps = 1;         // flatten: chain steps per factor in the first round (0: unlimited; measured: 1,2,4,.. 8.5 ms; unlimited 11.2 ms)
    int flatten_growth = 8;        // ... and the factor the budget grows by per round (measured at 256 MiB: x2 8.4 ms, x4 7.4 ms, x8 7.0 ms)
    int sa_init_syms = 0;          // classic suffix sort: cap on the symbols of the initial key (0: as many as 64 bits hold; tuning)
    int dec_done = 1;              // decompression: final-bit mask in the pointer-jumping rounds (0: A/B switch)
    int dec_log = 0, level_log = 0, wsort_log = 0, eager_dump = 0, small_prof = 0, arena_log = 0;   // diagnostics on stderr

    // overlapped D2H of the compressed stream (end-to-end entry point with a caller buffer): while the pack kernel works on the
    // later tiles, the finished front part of the stream already travels to the host on a second stream
    // byte histogram of the text a pipeline call works on: one pass serves the sentinel check (count of 0 bytes) and the suffix array's
    // symbol codes; a host-buffer call accumulates it chunk by chunk behind the chunks of the upload
    u32 hist_cache[256] = {};
    const u8* hist_ptr = nullptr;
    size_t hist_n = 0;
    hipStream_t copy_stream = nullptr;
    hipStream_t aux_stream = nullptr;  // low-priority side stream: work that fills idle device time behind the upload (wsort.hip wsort_pre_chunk)
    hipEvent_t ev_aux[2] = {};
    hipEvent_t ev_copy[40] = {};
    struct WPre* wpre = nullptr;   // level 1 of the suffix sort done behind the upload (prim.hpp), owned by the API context
    u8* d2h_host = nullptr;        // destination (host) of the running call, or null
    size_t d2h_cap = 0;
    size_t d2h_done = 0;           // bytes of the stream already on their way when encode returns

    bool profiling = false;
    KernelProfile kprof[K_CLASS_COUNT];
    struct Pending { hipEvent_t a, b; int cls; u64 bytes; };
    Pending* pend = nullptr;
    int npend = 0, pend_cap = 0;
    hipEvent_t* ev_pool = nullptr;
    int ev_pool_size = 0, ev_pool_used = 0;

    // returns the index of a pending record (or -1 when profiling is off / the pool is exhausted)
    int prof_begin(int cls, u64 bytes) {
        if (!profiling || npend >= pend_cap || ev_pool_used + 2 > ev_pool_size) return -1;
        Pending& p = pend[npend];
        p.a = ev_pool[ev_pool_used++]; p.b = ev_pool[ev_pool_used++]; p.cls = cls; p.bytes = bytes;
        HIP_TRY(hipEventRecord(p.a, stream));
        return npend++;
    }
    void prof_end(int idx) { if (idx >= 0) HIP_TRY(hipEventRecord(pend[idx].b, stream)); }
    struct ProfScope {     // RAII: times everything enqueued on the stream between construction and destruction
        Ctx& c; int idx;
        ProfScope(Ctx& ctx, int cls, u64 bytes) : c(ctx), idx(ctx.prof_begin(cls, bytes)) {}
        ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c.pend[idx].b, c.stream); }
    };
    // call after a stream synchronisation
    void prof_collect() {
        for (int i = 0; i < npend; ++i) {
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, pend[i].a, pend[i].b));
            KernelProfile& k = kprof[pend[i].cls];
            k.ms += t; k.launches++; k.bytes += pend[i].bytes;
        }
        npend = 0; ev_pool_used = 0;
    }

    void ensure_arena(size_t bytes) {
        arena.top_hi = 0;
        if (arena.size >= bytes) { arena.top = 0; return; }
        if (arena.base) { HIP_TRY(hipFree(arena.base)); arena.base = nullptr; arena.size = 0; }
        HIP_TRY(hipMalloc((void**)&arena.base, bytes));
        if (arena_log) fprintf(stderr, "arena: %zu bytes at %p\n", bytes, (void*)arena.base);
        arena.size = bytes;
        arena.top = 0;
    }
    // read back a few scalars (stream-ordered, synchronous)
#ifdef __HIPCC__
    bool read_words_fast(const void* dptr, void* out, size_t bytes) {
        if (!fast_read || !zc_host || bytes == 0 || bytes > 4096 || (bytes & 3) || ((uintptr_t)dptr & 3)) return false;
        const u32 seq = ++zc_seq;
        publish_words_kernel<<<1, 64, 0, stream>>>((const u32*)dptr, (u32)(bytes / 4), zc_dev, zc_dev + 1024, seq);
        if (hipGetLastError() != hipSuccess) return false;
        volatile u32* flag = zc_host + 1024;
        for (u64 spins = 0;; ++spins) {
            if (__atomic_load_n((const u32*)flag, __ATOMIC_ACQUIRE) == seq) break;
            if ((spins & 0xFFFF) == 0xFFFF) {                    // every so often: has the stream failed?
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipSuccess && q != hipErrorNotReady) throw HipError{q, __FILE__, __LINE__};
            }
        }
        memcpy(out, zc_host, bytes);
        return true;
    }
    // for kernels that publish their result themselves (same protocol as publish_words_kernel): the kernel stores its words
    // at *dst (system scope), then `seq` at *flag; publish_wait spins on the flag and copies `bytes` (<= 4096) out
    bool publish_begin(u32** dst, u32** flag, u32* seq, u32 block = 0) {
        if (!fast_read || !zc_host) return false;
        *seq = ++zc_seq; *dst = zc_dev + (size_t)block * ZC_WORDS; *flag = *dst + 1024;
        return true;
    }
    // read-back under way: the words at dptr travel to block `block` of the mapped host area once the stream gets there; returns the
    // sequence number publish_wait() takes, 0 if the fast path is not available (the caller then reads synchronously)
    u32 publish_async(const void* dptr, size_t bytes, u32 block) {
        if (!fast_read || !zc_host || bytes == 0 || bytes > 4096 || (bytes & 3) || ((uintptr_t)dptr & 3) || block >= ZC_BLOCKS) return 0;
        u32 seq = ++zc_seq;
        if (seq == 0) seq = ++zc_seq;
        u32* dst = zc_dev + (size_t)block * ZC_WORDS;
        publish_words_kernel<<<1, 64, 0, stream>>>((const u32*)dptr, (u32)(bytes / 4), dst, dst + 1024, seq);
        return hipGetLastError() == hipSuccess ? seq : 0;
    }
    void publish_wait(u32 seq, void* out, size_t bytes, u32 block = 0) {
        volatile u32* flag = zc_host + (size_t)block * ZC_WORDS + 1024;
        for (u64 spins = 0;; ++spins) {
            if (__atomic_load_n((const u32*)flag, __ATOMIC_ACQUIRE) == seq) break;
            if ((spins & 0xFFFF) == 0xFFFF) {
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipSuccess && q != hipErrorNotReady) throw HipError{q, __FILE__, __LINE__};
            }
        }
        memcpy(out, zc_host + (size_t)block * ZC_WORDS, bytes);
    }
#else
    bool read_words_fast(const void*, void*, size_t) { return false; }
#endif
    template <typename T> T read(const T* dptr) {
        T v;
        if (read_words_fast(dptr, &v, sizeof(T))) return v;
        T* h = (T*)pinned;
        HIP_TRY(hipMemcpyAsync(h, dptr, sizeof(T), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return *h;
    }
    template <typename T> void read_n(const T* dptr, T* out, size_t count) {
        const size_t bytes = sizeof(T) * count;
        if (read_words_fast(dptr, out, bytes)) return;
        if (bytes <= pinned_size) {          // small read-backs go through the pinned block (pageable D2H copies are slow)
            HIP_TRY(hipMemcpyAsync(pinned, dptr, bytes, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            memcpy(out, pinned, bytes);
        } else {
            HIP_TRY(hipMemcpyAsync(out, dptr, bytes, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
    }
};

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one; observed behaviour, used for speed
// only, never for correctness).  Streaming kernels whose neighbouring tiles write neighbouring addresses give every XCD
// one contiguous range of tiles, so that the partial cache lines of consecutive tiles meet in ONE L2 and leave it as
// whole lines.  per_xcd = ceil(tiles / 8) (grid = 8 * per_xcd), 0 = identity mapping.
__device__ __forceinline__ u32 xcd_tile(u32 bid, u32 per_xcd) { return per_xcd ? (bid & 7u) * per_xcd + (bid >> 3) : bid; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Inclusive scan across the 64 lanes of a wave.
// 32-bit values: DPP row shifts inside the four rows of 16 lanes (Hillis-Steele), then the last lane of row 0 / 2 is broadcast into
// row 1 / 3 and lane 31 into the upper half -- six VALU instructions instead of six ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ u32 wave_inclusive_sum_u32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);    // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);    // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);    // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);    // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
    if constexpr (sizeof(T) == 4) return (T)wave_inclusive_sum_u32((u32)v);
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ u32 wave_inclusive_max(u32 v) {          // (0 is the identity: lanes without a source read 0)
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}
// Reductions: the inclusive scan's last lane, read back through a scalar register (every lane gets the result).
template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
    if constexpr (sizeof(T) == 4) return (T)__builtin_amdgcn_readlane((int)wave_inclusive_sum_u32((u32)v), 63);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ u32 wave_reduce_max(u32 v) { return (u32)__builtin_amdgcn_readlane((int)wave_inclusive_max(v), 63); }
__device__ __forceinline__ u32 wave_reduce_min(u32 v) { return ~wave_reduce_max(~v); }

// LDS accesses that must really happen (counters and tables shared between the lanes of a wave, re-read after another lane
// wrote them).  A `volatile` access through an ordinary pointer is compiled to a FLAT instruction with system-coherence bits
// plus a full wait (the address-space inference leaves volatile accesses alone): several times the cost of a ds_ instruction.
// The cast to the LDS address space keeps the access volatile AND a ds_read / ds_write.
template <typename T> __device__ __forceinline__ T lds_load(const T* p) { return *(const volatile __attribute__((address_space(3))) T*)p; }
template <typename T> __device__ __forceinline__ void lds_store(T* p, T v) { *(volatile __attribute__((address_space(3))) T*)p = v; }

// Wave-level match through LDS: returns the mask of the lanes of this wave that hold a valid key with the same digit.
// Every lane ORs its bit into the digit's slot of the wave's private table M (256 or 512 x u64, all zero on entry), reads the slot
// back and clears it again (zero on exit).  OR is order-independent, so the result does not depend on how the LDS unit serialises
// lanes that hit the same slot; LDS instructions of one wave execute in order.  The alternative -- one ballot per digit bit and
// five 64-bit-mask VALU operations per lane for each of them -- made every radix pass VALU-bound on CDNA (a wave64 instruction
// occupies a 16-lane SIMD for 4 cycles): ~46 VALU instructions per key against ~10 + 3 LDS operations.
__device__ __forceinline__ u64 wave_match_lds(unsigned long long* M, u32 d, bool valid, u64 lanebit) {
    const u32 d0 = __builtin_amdgcn_readfirstlane(d);
    if (__all(valid && d == d0)) return ~0ull;                   // whole row in one bin (high-order digits, long runs)
#ifdef TDC_MATCH_B64
    if (valid) atomicOr(&M[d], (unsigned long long)lanebit);
#else
    // (the lane's bit lives in one 32-bit half of the slot: a 4-byte LDS atomic occupies one bank instead of two)
    if (valid) atomicOr((unsigned*)&M[d] + (lane_id() >> 5), (unsigned)(lanebit >> (32 * (lane_id() >> 5))));
#endif
    __builtin_amdgcn_wave_barrier();
    const u64 peers = valid ? lds_load(&M[d]) : 0ull;
    __builtin_amdgcn_wave_barrier();
    if (valid) lds_store(&M[d], 0ull);
    return peers;
}

// The same with the frequent digits taken out first (round 6).  Lanes that hold the same digit hit the same LDS word with their atomic
// OR, and the LDS unit serialises them: in the passes over the high-order digits of a leaf (a handful of distinct values per row) one
// match costs 16 - 32 LDS cycles instead of 2 -- `SQ_LDS_BANK_CONFLICT` was 46 % of the LDS array cycles of ws_leaf_sort_kernel.  Up to
// PEEL times the digit of the first remaining lane is matched by a ballot instead (all of its lanes leave together, so the digits that
// reach the table never mix with peeled ones); a group of fewer than MING lanes ends the peeling (the rest of the row is diverse: few
// lanes per word).  Zero on entry, zero on exit, like wave_match_lds.
template <int PEEL, int MING>
__device__ __forceinline__ u64 wave_match_peel(unsigned long long* M, u32 d, bool valid, u64 lanebit) {
    u64 rem = __ballot(valid);
    u64 peers = 0;
    bool done = !valid;
#pragma unroll
    for (int it = 0; it < PEEL; ++it) {
        if (!rem) break;
        const int lead = __builtin_ctzll(rem);
        const u32 dl = (u32)__builtin_amdgcn_readlane((int)d, lead);
        const u64 grp = __ballot(!done && d == dl);
        if (__popcll(grp) < MING) break;
        if (!done && d == dl) { peers = grp; done = true; }
        rem &= ~grp;
    }
    if (rem) {                                                    // (uniform: the rest of the row through the table)
        if (!done) atomicOr((unsigned*)&M[d] + (lane_id() >> 5), (unsigned)(lanebit >> (32 * (lane_id() >> 5))));
        __builtin_amdgcn_wave_barrier();
        if (!done) peers = lds_load(&M[d]);
        __builtin_amdgcn_wave_barrier();
        if (!done) lds_store(&M[d], 0ull);
    }
    return peers;
}

// Block-wide exclusive sum for a block of NW waves (NW*64 threads).  `smem` must hold NW+1 values of T.
// Returns the exclusive prefix of `v` over the block in thread order; `total` receives the block sum.
template <typename T, int NW>
__device__ __forceinline__ T block_exclusive_sum(T v, T* smem, T& total) {
    const int lane = lane_id(), w = wave_id();
    T inc = wave_inclusive_sum(v);
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { T t = smem[i]; smem[i] = run; run += t; }
        smem[NW] = run;
    }
    __syncthreads();
    T res = smem[w] + inc - v;
    total = smem[NW];
    __syncthreads();
    return res;
}

// Block-wide inclusive max (same contract).
template <int NW>
__device__ __forceinline__ u32 block_inclusive_max(u32 v, u32* smem, u32& total) {
    const int lane = lane_id(), w = wave_id();
    u32 inc = wave_inclusive_max(v);
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { u32 t = smem[i]; smem[i] = run; run = max(run, t); }
        smem[NW] = run;
    }
    __syncthreads();
    u32 res = max(smem[w], inc);
    total = smem[NW];
    __syncthreads();
    return res;
}

// ---- workgroup bitonic sort of 2048 (key, value) pairs held in registers ------------------------------------------
// 256 threads x 8 elements; element index i = thread * 8 + r.  Compare-exchange partners at distance < 8 live in the
// same thread, at distance 8..256 in the same wave (lane xor, __shfl_xor, no barrier); only distances >= 512 go through
// LDS (3 barrier stages out of 66).  `xk` / `xv`: LDS scratch of 2048 entries each.  Ascending; keys must be distinct or
// ties may land in any order.
__device__ __forceinline__ void bitonic_cmpx(u64& ka, u32& va, u64& kb, u32& vb, bool up) {
    if ((ka > kb) == up) { const u64 tk = ka; ka = kb; kb = tk; const u32 tv = va; va = vb; vb = tv; }
}
template <u32 J>
__device__ __forceinline__ void bitonic_local(u64 (&k)[8], u32 (&v)[8], u32 t, u32 k2) {
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        constexpr u32 JJ = J;
        const u32 x = r ^ JJ;
        if (x > r) bitonic_cmpx(k[r], v[r], k[x], v[x], (((t * 8 + r) & k2) == 0));
    }
}
// `np2`: power of two >= number of real elements (the rest is padding with maximal keys): the network only has to
// merge up to blocks of np2 elements, which shortens it from 66 to log2(np2)*(log2(np2)+1)/2 stages.
__device__ inline void block_bitonic_sort_2048(u64 (&k)[8], u32 (&v)[8], u64* xk, u32* xv, u32 np2 = 2048) {
    const u32 t = threadIdx.x;
    for (u32 k2 = 2; k2 <= np2; k2 <<= 1) {
        for (u32 j = k2 >> 1; j > 0; j >>= 1) {
            if (j < 8) {                                   // same thread (register indices must be compile-time constants)
                if (j == 4) bitonic_local<4>(k, v, t, k2);
                else if (j == 2) bitonic_local<2>(k, v, t, k2);
                else bitonic_local<1>(k, v, t, k2);
            } else if (j < 512) {                          // same wave: partner lane = lane ^ (j / 8)
                const u32 lm = j >> 3;
                const bool lower = ((t & lm) == 0);        // this thread holds the lower index of every pair
#pragma unroll
                for (u32 r = 0; r < 8; ++r) {
                    const u64 ok = __shfl_xor(k[r], (int)lm, 64);
                    const u32 ov = __shfl_xor(v[r], (int)lm, 64);
                    const bool up = (((t * 8 + r) & k2) == 0);
                    // the pair (lower index i, upper index i ^ j): keep min at the lower index iff `up`
                    const bool take_other = lower ? ((k[r] > ok) == up) : ((ok > k[r]) == up);
                    if (take_other) { k[r] = ok; v[r] = ov; }
                }
            } else {                                       // across waves: through LDS
                __syncthreads();
#pragma unroll
                for (u32 r = 0; r < 8; ++r) { xk[t * 8 + r] = k[r]; xv[t * 8 + r] = v[r]; }
                __syncthreads();
                const bool lower = (((t * 8) & j) == 0);
#pragma unroll
                for (u32 r = 0; r < 8; ++r) {
                    const u32 i = t * 8 + r;
                    const u64 ok = xk[i ^ j];
                    const u32 ov = xv[i ^ j];
                    const bool up = ((i & k2) == 0);
                    const bool take_other = lower ? ((k[r] > ok) == up) : ((ok > k[r]) == up);
                    if (take_other) { k[r] = ok; v[r] = ov; }
                }
            }
        }
    }
    __syncthreads();
}

#endif  // __HIPCC__

}  // namespace tdc
