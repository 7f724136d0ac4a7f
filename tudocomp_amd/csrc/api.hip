// api.hip -- the C ABI of include/tdc_gpu.h on top of the stage functions (stages.hpp).
#include "../../include/tdc_gpu.h"
#include "stages.hpp"
#include "prim.hpp"
#include "huffman_host.hpp"

#include <new>
#include <string>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <ctype.h>

using namespace tdc;

struct tdc_gpu_ctx {
    Ctx c;
    WPre pre;                   // level 1 of the suffix sort behind the upload (compress_host); c.wpre points here
    std::string last_error;
    int last_decode_device = 0; // the last decompression parsed its token stream on the device
    const u8* kept = nullptr;   // tdc_gpu_lcpcomp_compress_keep: the stream of the last call, in the arena (until the next call)
    size_t kept_len = 0;
};

namespace {

struct ArgError { int code; const char* msg; };

// hipSetDevice is a per-thread setting of the embedding program: switch to the context's device for the duration of one
// API call only
struct DeviceGuard {
    int want, prev = -1;
    explicit DeviceGuard(int d) : want(d) {}
    hipError_t enter() {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        return prev == want ? hipSuccess : hipSetDevice(want);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

// malloc'd host buffer that is freed unless release()d: the output buffers handed to the caller are allocated before the
// last synchronisation, and an error surfacing there must not leak them
struct HostBuf {
    void* p = nullptr;
    explicit HostBuf(size_t bytes) : p(malloc(bytes ? bytes : 1)) { if (!p) throw std::bad_alloc(); }
    ~HostBuf() { free(p); }
    template <typename T> T* as() { return (T*)p; }
    template <typename T> T* release() { T* r = (T*)p; p = nullptr; return r; }
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
};

template <typename F>
int guarded(tdc_gpu_ctx* ctx, F&& f) {
    if (!ctx) return TDC_GPU_ERR_ARG;
    ctx->last_error.clear();
    ctx->kept = nullptr; ctx->kept_len = 0;      // (every call may reuse the arena)
    ctx->c.hist_ptr = nullptr;                   // the cached byte histogram belongs to ONE call (same address, other text: stale)
    DeviceGuard dg(ctx->c.device);               // the caller's current device is restored on every exit path
    try {
        HIP_TRY(dg.enter());
        f();
        if (ctx->c.d_err) {                      // device-side error word (e.g. a look-back that timed out)
            u32 e = 0;
            HIP_TRY(hipMemcpy(&e, ctx->c.d_err, sizeof(u32), hipMemcpyDeviceToHost));
            if (e) {
                HIP_TRY(hipMemset(ctx->c.d_err, 0, sizeof(u32)));
                throw HipError{hipErrorUnknown, "device-side error flag set (radix look-back timeout)", (int)e};
            }
        }
        return TDC_GPU_OK;
    } catch (const HipError& e) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s (%s:%d)", hipGetErrorString(e.e), e.file, e.line);
        ctx->last_error = buf;
        (void)hipGetLastError();
        if (e.e == hipErrorOutOfMemory) return TDC_GPU_ERR_OOM;
        if (e.e == hipErrorUnknown) return TDC_GPU_ERR_INTERNAL;
        if (e.e == hipErrorInvalidValue && e.line < 0) return TDC_GPU_ERR_UNSUPPORTED;
        return TDC_GPU_ERR_HIP;
    } catch (const ArgError& e) {
        ctx->last_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc&) {
        ctx->last_error = "host allocation failed";
        return TDC_GPU_ERR_OOM;
    } catch (...) {
        ctx->last_error = "unknown exception";
        return TDC_GPU_ERR_INTERNAL;
    }
}

// + 192 MiB: fixed-size scratch (the SLE coder's 2^24-entry k-mer table and its sort buffers are the largest)
size_t arena_need(size_t n) { return 112 * n + ((size_t)192 << 20); }
// (a context created with TDC_GPU_WSORT_SMALLRUN -- tests: every run of tying records is handed on -- needs ~48 B per byte more for the
//  hand-over lists; per context, not per process: other contexts of a test run keep the product's budget)
size_t arena_need(const Ctx& c, size_t n) { return arena_need(n) + (c.wsort_small ? 64 * n : 0) + (c.wsort_cmax < 16 ? 8 * n : 0); }

// public coder id (+ SLE's kmer option in bits 8..) -> coder id of encode_stream
int lcpcomp_enc_coder(int coder) {
    const int base = coder & 0xFF, k = coder >> 8;
    if (base == TDC_GPU_CODER_SLE) {
        if (k < 0 || k > 7) throw ArgError{TDC_GPU_ERR_ARG, "sle: kmer must be in 1..7"};       // SLECoder.hpp:12,86 (max_kmer = 7)
        return 3 | ((k ? k : 3) << 8);
    }
    if (k == 0 && base == TDC_GPU_CODER_HUFF) return 0;
    if (k == 0 && base == TDC_GPU_CODER_ARITH) return 1;
    if (k == 0 && base == TDC_GPU_CODER_ASCII) return 2;
    throw ArgError{TDC_GPU_ERR_UNSUPPORTED, "lcpcomp: coder must be huff, arithmetic, ascii or sle"};
}

// the context's arena for a call; a device that cannot hold it is reported with both numbers instead of a bare allocation failure
void reserve_arena(Ctx& c, size_t bytes) {
    if (c.arena.size < bytes) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr + c.arena.size < bytes) {
            static thread_local char msg[256];
            snprintf(msg, sizeof(msg), "device %d has %.1f GB free of %.1f GB, this call needs an arena of %.1f GB (112 bytes per text byte + 192 MiB): "
                     "use smaller blocks (tdc_gpu_arena_bytes)", c.device, (double)(fr + c.arena.size) / 1e9, (double)tot / 1e9, (double)bytes / 1e9);
            throw ArgError{TDC_GPU_ERR_OOM, msg};
        }
        (void)hipGetLastError();
    }
    c.ensure_arena(bytes);
}

void check_text_args(const void* text, size_t n) {
    if (!text) throw ArgError{TDC_GPU_ERR_ARG, "text is NULL"};
    if (n == 0) throw ArgError{TDC_GPU_ERR_NO_SENTINEL, "empty view: the text must end with a 0 sentinel"};
    if (n >= 0x7FFFFFFFull) throw ArgError{TDC_GPU_ERR_TOO_LARGE, "text length must be < 2^31 - 1 (32-bit len_t)"};
}

// Event marks are recorded while the pipeline is enqueued; the elapsed times are only read in finish(), after the
// stream has been synchronised (hipEventElapsedTime on a pending event returns hipErrorNotReady).
struct Events {
    Ctx& c;
    int used = 0;
    struct Span { float* dst; int a, b; };
    Span spans[16];
    int nspans = 0;
    explicit Events(Ctx& ctx) : c(ctx) {}
    int tick() { HIP_TRY(hipEventRecord(c.ev[used], c.stream)); return used++; }
    void span(float* dst, int a, int b) { if (dst) spans[nspans++] = Span{dst, a, b}; }
    void finish() {
        HIP_TRY(hipStreamSynchronize(c.stream));
        for (int i = 0; i < nspans; ++i) HIP_TRY(hipEventElapsedTime(spans[i].dst, c.ev[spans[i].a], c.ev[spans[i].b]));
        nspans = 0;
        c.prof_collect();
    }
};

struct DevArrays {
    u32 *sa = nullptr, *isa = nullptr, *phi = nullptr, *plcp = nullptr;
    FactorSpace fs;
    u32 maxlcp = 0;
    EncodeEarly* early = nullptr;           // first half of the encoder, run inside the flatten stage (run_factorize)
    DevArrays() = default;
    DevArrays(const DevArrays&) = delete;
    DevArrays& operator=(const DevArrays&) = delete;
    ~DevArrays() { encode_early_free(early); }
};

// c.stream points at another stream for the lifetime of the object (the stage functions enqueue on c.stream)
struct StreamSwap {
    Ctx& c; hipStream_t saved;
    StreamSwap(Ctx& ctx, hipStream_t other) : c(ctx), saved(ctx.stream) { c.stream = other; }
    ~StreamSwap() { c.stream = saved; }
};

// Checks that the 0 byte occurs exactly once, at n - 1 (ds/TextDS.hpp:132-138).  The count comes from the byte histogram of the text,
// which the suffix array needs anyway (one pass for both; a host-buffer call has accumulated it behind the upload already).
void validate_device_text(Ctx& c, const u8* d_text, size_t n) {
    if (!(c.hist_ptr == d_text && c.hist_n == n)) {
        const size_t mark = c.arena.mark();
        u32* d_hist = c.arena.get<u32>(256);
        HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), c.stream));
        text_histogram_add(c, d_text, n, d_hist);
        text_histogram_finish(c, d_text, n, d_hist);
        c.arena.release(mark);
    }
    u8 last = 1;
    HIP_TRY(hipMemcpyAsync(&last, d_text + n - 1, 1, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    const u32 zeros = c.hist_cache[0];
    if (last != 0) throw ArgError{TDC_GPU_ERR_NO_SENTINEL, "text does not end with a 0 sentinel"};
    if (zeros != 1) throw ArgError{TDC_GPU_ERR_ARG, "text contains 0 bytes besides the sentinel (escape the input first)"};
}

// SA -> ISA -> Phi -> PLCP  (TextDS::require, ds/TextDS.hpp:247-292)
// want_phi = false (lcpcomp with comp=arrays): where the fused scatter runs, Phi is not materialised -- 8-byte instead of 12-byte records
// through its two partition levels; the factorizer takes a factor's source from SA[ISA[p] - 1] (A.phi stays NULL)
void run_textds(Ctx& c, const u8* d_text, size_t n, DevArrays& A, tdc_gpu_stats* st, Events* ev, bool want_phi = true) {
    A.sa = c.arena.get<u32>(n);
    A.isa = c.arena.get<u32>(n);
    A.phi = nullptr;                                          // (taken behind the suffix array, and only where a Phi array is built)
    A.plcp = c.arena.get<u32>(n);
    u32* d_max = c.arena.get<u32>(1);
    SAStats ss;
    SAExtra ex;
    ex.lcp8 = c.arena.get<u8>(n + 64);                        // neighbour LCPs of the wide path (suffix_array.hip)
    const int e0 = ev ? ev->tick() : 0;
    build_suffix_array(c, d_text, n, A.sa, A.isa, &ss, &ex);
    const int e1 = ev ? ev->tick() : 0;
    int e2;
    if (!(ex.mode == 1 && !want_phi && c.phi_lazy)) A.phi = c.arena.get<u32>(n);
    if (ex.mode == 1) {                                       // ISA + Phi + PLCP in one scatter of the final suffix array
        build_isa_phi_plcp_fused(c, A.sa, ex.lcp8, n, A.isa, A.phi, A.plcp, d_max);
        e2 = ev ? ev->tick() : 0;
    } else {
        build_phi(c, A.sa, n, A.phi);
        e2 = ev ? ev->tick() : 0;
        build_plcp(c, d_text, n, A.phi, A.plcp, d_max);
    }
    const int e3 = ev ? ev->tick() : 0;
    A.maxlcp = c.read(d_max);
    if (st) {
        st->maxlcp = A.maxlcp;
        st->sa_rounds = ss.rounds; st->sa_init_syms = ss.init_syms; st->sa_sorted_elems = ss.sorted_elems;
        st->sa_key_words = ss.wide_kw; st->sa_text_rounds = ss.text_rounds; st->sa_mode = (uint32_t)ex.mode; st->sa_overlapped = ss.overlapped;
        st->sa_star_chains = (uint32_t)std::min<u64>(ss.star_chains, 0xFFFFFFFFull);
        if (ev) { ev->span(&st->ms_sa, e0, e1); ev->span(&st->ms_phi, e1, e2); ev->span(&st->ms_plcp, e2, e3); }
    }
}

// enc_coder >= 0 (with d_text): the stream is encoded next with this coder of encode_stream -- the first half of the encoder may run
// inside the flatten stage
void run_factorize(Ctx& c, size_t n, DevArrays& A, u32 threshold, int flatten, tdc_gpu_stats* st, Events* ev, int strategy = 0,
                   int enc_coder = -1, const u8* d_text = nullptr) {
    A.fs.flen = c.arena.get<u32>(n);
    A.fs.owner = c.arena.get<u32>(n);
    A.fs.fsrc = c.arena.get<u32>(n);
    A.fs.fpos = c.arena.get<u32>(n);
    A.fs.flenl = threshold >= 2 ? A.fs.fpos + (n + 1) / 2 : nullptr;    // (a factor covers >= threshold positions: at most n / 2 of them, the list of
                                                                         //  their lengths fits the upper half of the position list)
    A.fs.cls = c.arena.get<u8>(n + 64);                  // class bytes for the encoder (filled by build_owner)
    // the metric's path (lcpcomp(comp=arrays, coder=huff) with the encoder's first half inside the flatten stage: nothing reads the dense
    // flen[] array behind build_owner): the factor lengths travel as bytes until then
    const bool early_planned = strategy == TDC_GPU_COMP_ARRAYS && flatten && enc_coder == 0 && d_text && c.enc_early && c.enc_rec && c.copy_stream && c.huff_ok &&
                               n >= (c.enc_early >= 2 ? (size_t)1 : ((size_t)1 << 20)) && threshold >= 2;
    if (early_planned && c.flen_bytes) A.fs.flen8 = c.arena.get<u8>(n + 64);
    A.fs.want_owner_rem = early_planned ? (u32)c.owner_rem : 0u;   // (behind build_owner only the flatten rounds read owner[] on this path: the encoder reads cls[] and the records)
    FactorizeStats fz;
    FlattenStats fl;
    const int e0 = ev ? ev->tick() : 0;
    if (strategy == TDC_GPU_COMP_PLCPPEAKS) plcp_peaks_factorize(c, n, A.phi, A.plcp, threshold, A.fs, &fz.factors);
    else if (strategy == TDC_GPU_COMP_MAXLCP) factorize_max_lcp(c, n, A.isa, A.phi, A.plcp, A.maxlcp, threshold, A.fs, &fz);
    else if (strategy == TDC_GPU_COMP_HEAP) factorize_max_heap(c, n, A.sa, A.isa, A.plcp, A.maxlcp, threshold, A.fs, &fz);
    else factorize_arrays(c, n, A.sa, A.isa, A.phi, A.plcp, A.maxlcp, threshold, A.fs, &fz);
    const int e1 = ev ? ev->tick() : 0;
    // The first half of the Huffman encoder (gaps, literal histogram, code table, bits per tile and their scan: 4-5 ms of streaming
    // kernels and three host round trips at 2e9 B) reads positions, lengths and class bytes but no source, and the flatten rounds are
    // bound by the latency of their chains, not by bandwidth: it runs on the copy stream next to the first round.
    const bool early = flatten && enc_coder == 0 && d_text && c.enc_early && c.copy_stream && c.huff_ok && n >= (c.enc_early >= 2 ? (size_t)1 : ((size_t)1 << 20)) &&
                       A.fs.have_list && A.fs.have_cls && A.fs.flenl && A.fs.nfact > 0;
    if (!early) expand_flen8(c, n, A.fs);                  // (planned, but there is no factor list to run it on: everybody else reads the dense array)
    if (early) {
        A.early = encode_early_reserve(c, n, c.enc_rec ? A.fs.nfact : 0);
        HIP_TRY(hipEventRecord(c.ev_copy[0], c.stream));                  // the factors are in place
        // one step per round (the host never waits for the copy stream while a round needs it), the rest when the rounds are over
        bool waited = false;
        try {
            flatten_factors(c, n, A.fs, &fl, [&](int round) {
                StreamSwap sw(c, c.copy_stream);
                if (!waited) { HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_copy[0], 0)); waited = true; }
                if (round == 1 || round == 2) encode_early_run(c, d_text, n, A.fs, enc_coder, A.early, false);
                else if (round == 0) encode_early_run(c, d_text, n, A.fs, enc_coder, A.early, true);
            }, c.enc_rec ? encode_early_rec(A.early) : nullptr);
        } catch (...) {
            (void)hipStreamSynchronize(c.copy_stream);         // nothing of this call stays behind on the second stream
            throw;
        }
    } else if (flatten) {
        flatten_factors(c, n, A.fs, &fl);
    } else {
        materialize_sources(c, n, A.fs);                   // (no Phi array: the encoder reads fsrc[] at every factor start)
    }
    const int e2 = ev ? ev->tick() : 0;
    if (st) {
        st->factors = fz.factors; st->entries = fz.entries; st->pushes = fz.pushes;
        st->levels = fz.levels; st->mis_rounds = fz.rounds; st->small_levels = fz.small_levels; st->purges = fz.purges; st->window_pass = fz.window_pass; st->window_lcut = fz.window_lcut; st->eager_levels = fz.eager_levels; st->eager_phases = fz.eager_phases;
        st->num_flattened = fl.num_flattened; st->max_depth_lb = fl.max_depth_lb; st->flatten_rounds = fl.rounds;
        if (ev) { ev->span(&st->ms_factorize, e0, e1); ev->span(&st->ms_flatten, e1, e2); }
    }
}

// whole pipeline on a device-resident text; output written to *d_out (8-byte aligned, capacity out_cap); if *d_out is
// NULL the buffer is taken from the arena once the factorization scratch has been released
size_t run_pipeline(Ctx& c, const u8* d_text, size_t n, u32 threshold, int flatten, int coder, u8** d_out_io, size_t out_cap,
                    tdc_gpu_stats* st, Events& ev, int strategy = 0) {
    if (threshold == 0) throw ArgError{TDC_GPU_ERR_ARG, "threshold must be >= 1"};
    validate_device_text(c, d_text, n);
    DevArrays A;
    run_textds(c, d_text, n, A, st, &ev, strategy != TDC_GPU_COMP_ARRAYS);
    const int enc_coder = lcpcomp_enc_coder(coder);
    run_factorize(c, n, A, threshold, flatten, st, &ev, strategy, enc_coder, d_text);
    EncodeStats es;
    if (!*d_out_io) { out_cap = align_up(encode_bound_coder(n, enc_coder) + 16, 8); *d_out_io = c.arena.get<u8>(out_cap); }
    u8* d_out = *d_out_io;
    const int e0 = ev.tick();
    const size_t out_len = encode_stream(c, d_text, n, A.fs, enc_coder, d_out, out_cap, &es, A.early);
    const int e1 = ev.tick();
    HIP_TRY(hipStreamSynchronize(c.stream));
    if (st) {
        st->n = n; st->out_len = out_len;
        st->flen_min = es.flen_min; st->flen_max = es.flen_max; st->fdist_max = es.fdist_max; st->sigma = es.sigma;
        ev.span(&st->ms_encode, e0, e1);
        st->arena_bytes = c.arena.high;
    }
    return out_len;
}

void validate_factor_list(size_t n, const uint32_t* pos, const uint32_t* src, const uint32_t* len, size_t z) {
    uint64_t end = 0;
    for (size_t i = 0; i < z; ++i) {
        if (len[i] == 0) throw ArgError{TDC_GPU_ERR_ARG, "factor with length 0"};
        if (pos[i] < end) throw ArgError{TDC_GPU_ERR_ARG, "factors must be sorted by pos and must not overlap"};
        end = (uint64_t)pos[i] + len[i];
        if (end > n) throw ArgError{TDC_GPU_ERR_ARG, "factor exceeds the text"};
        if (src && (uint64_t)src[i] + len[i] > n) throw ArgError{TDC_GPU_ERR_ARG, "factor source exceeds the text"};
    }
}

}  // namespace

extern "C" {

const char* tdc_gpu_strerror(int status) {
    switch (status) {
        case TDC_GPU_OK: return "success";
        case TDC_GPU_ERR_HIP: return "HIP runtime error (is a gfx950 GPU visible?)";
        case TDC_GPU_ERR_ARG: return "invalid argument";
        case TDC_GPU_ERR_NO_SENTINEL: return "Expected a sentinel byte (0) at the end of the input text";
        case TDC_GPU_ERR_TOO_LARGE: return "input too large: text length must be < 2^31";
        case TDC_GPU_ERR_OOM: return "out of memory";
        case TDC_GPU_ERR_UNSUPPORTED: return "No implementation found for this coder/strategy";
        case TDC_GPU_ERR_INTERNAL: return "internal error";
        default: return "unknown status";
    }
}

const char* tdc_gpu_last_error(const tdc_gpu_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

void tdc_gpu_free(void* p) { free(p); }

// ---- options ---------------------------------------------------------------------------------------------------------------------
// Every switch of the library, in ONE table: name (the environment variable of the development aid is TDC_GPU_ + upper case), the
// field, and the values it accepts (out-of-range values are clamped the way the environment parser of rounds 1-5 did).  README.md
// lists them with the test that exercises each.
namespace {
struct OptionDef { const char* name; void (*set)(Ctx&, long); };
inline int clampi(long v, long lo, long hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }
const OptionDef OPTIONS[] = {
    { "fastread",         [](Ctx& c, long v) { c.fast_read = v ? 1 : 0; } },
    { "sa_local",         [](Ctx& c, long v) { c.sa_local_sort = (int)v; } },
    { "radix_waves",      [](Ctx& c, long v) { c.radix_waves = v == 8 ? 8 : 4; } },
    { "window_lcut",      [](Ctx& c, long v) { c.window_lcut = clampi(v, 0, 63); } },
    { "window_halo",      [](Ctx& c, long v) { c.window_halo = clampi(v, 0, 2048); } },
    { "dec_seg",          [](Ctx& c, long v) { c.dec_seg = v < 4096 ? 4096 : (v > (1l << 30) ? (size_t)1 << 30 : (size_t)v); } },
    { "dec_lean",         [](Ctx& c, long v) { c.dec_lean = v != 0; } },
    { "dec_parse",        [](Ctx& c, long v) { c.dec_parse = clampi(v, 0, 2); } },
    { "dec_done",         [](Ctx& c, long v) { c.dec_done = v != 0; } },
    { "dec_log",          [](Ctx& c, long v) { c.dec_log = v != 0; } },
    { "window_force_fail",[](Ctx& c, long v) { c.window_force_fail = v ? 1 : 0; } },
    { "window_large",     [](Ctx& c, long v) { c.window_large_lists = v ? 1 : 0; } },
    { "window_src",       [](Ctx& c, long v) { c.window_src = v ? 1 : 0; } },
    { "plcp_samples",     [](Ctx& c, long v) { c.plcp_samples = v != 0; } },
    { "small_pipeline",   [](Ctx& c, long v) { c.small_pipeline = v != 0; } },
    { "small_big",        [](Ctx& c, long v) { c.small_big = (int)v; } },
    { "small_prof",       [](Ctx& c, long v) { c.small_prof = v != 0; } },
    { "phi_lazy",         [](Ctx& c, long v) { c.phi_lazy = v != 0; } },
    { "fs_pair",          [](Ctx& c, long v) { c.fs_pair = v != 0; } },
    { "enc_early",        [](Ctx& c, long v) { c.enc_early = (int)v; } },
    { "owner_rem",        [](Ctx& c, long v) { c.owner_rem = (int)std::min<long>(std::max<long>(v, 0), 8); } },
    { "enc_rec",          [](Ctx& c, long v) { c.enc_rec = v != 0; } },
    { "level_purge",      [](Ctx& c, long v) { c.level_purge = v != 0; } },
    { "level_log",        [](Ctx& c, long v) { c.level_log = v != 0; } },
    { "eager",            [](Ctx& c, long v) { c.eager_levels = v != 0; } },
    { "eager_dump",       [](Ctx& c, long v) { c.eager_dump = v != 0; } },
    { "flen_bytes",       [](Ctx& c, long v) { c.flen_bytes = v != 0; } },
    { "flatten_steps",    [](Ctx& c, long v) { c.flatten_steps = v <= 0 ? (1 << 30) : clampi(v, 1, 1 << 30); } },
    { "flatten_growth",   [](Ctx& c, long v) { c.flatten_growth = clampi(v, 2, 1 << 20); } },
    { "sa_refine",        [](Ctx& c, long v) { c.sa_refine = v != 0; } },
    { "sa_pairs",         [](Ctx& c, long v) { c.sa_pairs = v != 0; } },
    { "sa_stars",         [](Ctx& c, long v) { c.sa_stars = v != 0; } },
    { "sa_fused_init",    [](Ctx& c, long v) { c.sa_fused_init = v != 0; } },
    { "sa_init_syms",     [](Ctx& c, long v) { c.sa_init_syms = clampi(v, 0, 64); } },
    { "radix_lds",        [](Ctx& c, long v) { c.radix_lds = (v >= 0 && v <= 2) ? (int)v : 2; } },
    { "xcd_remap",        [](Ctx& c, long v) { c.xcd_remap = (v >= 0 && v <= 2) ? (int)v : 0; } },
    { "bucket_scatter",   [](Ctx& c, long v) { c.bucket_scatter = v ? 1 : 0; } },
    { "ssort",            [](Ctx& c, long v) { c.ssort = v ? 1 : 0; } },
    { "ssort_levels",     [](Ctx& c, long v) { c.ssort_levels = (v >= 1 && v <= 3) ? (int)v : 0; } },
    { "msd_partition",    [](Ctx& c, long v) { c.msd_partition = v ? 1 : 0; } },
    { "wsort",            [](Ctx& c, long v) { c.wsort = v ? 1 : 0; } },
    { "wsort_min",        [](Ctx& c, long v) { c.wsort_min = v < 4096 ? 4096 : (size_t)v; } },
    { "wsort_syms",       [](Ctx& c, long v) { c.wsort_syms = (v >= 4 && v <= 64) ? (int)v : 0; } },
    { "wsort_kw",         [](Ctx& c, long v) { c.wsort_kw = (v == 1 || v == 2) ? (int)v : 0; } },
    { "wsort_rounds",     [](Ctx& c, long v) { c.wsort_rounds = clampi(v, 0, 100); } },
    { "wsort_smallrun",   [](Ctx& c, long v) { c.wsort_small = v ? 1 : 0; } },
    { "wsort_overlap",    [](Ctx& c, long v) { c.wsort_overlap = v ? 1 : 0; } },
    { "wsort_predig",     [](Ctx& c, long v) { c.wsort_predig = v ? 1 : 0; } },
    { "wsort_prehist",    [](Ctx& c, long v) { c.wsort_prehist = v ? 1 : 0; } },
    { "sa_seg_bigcap",    [](Ctx& c, long v) { c.sa_seg_bigcap = clampi(v, 0, 65536); } },
    { "sa_seg_rounds",    [](Ctx& c, long v) { c.sa_seg_rounds = clampi(v, 0, 2); } },
    { "wsort_run_streams",[](Ctx& c, long v) { c.wsort_run_streams = v != 0; } },
    { "wsort_predig_skip",[](Ctx& c, long v) { c.wsort_predig_skip = clampi(v, 0, 24); } },
    { "wsort_fuse",       [](Ctx& c, long v) { c.wsort_fuse = v ? 1 : 0; } },
    { "wsort_order",      [](Ctx& c, long v) { c.wsort_order = v ? 1 : 0; } },
    { "wsort_two",        [](Ctx& c, long v) { c.wsort_two = (v >= 0 && v <= 2) ? (int)v : 0; } },
    { "wsort_leaf",       [](Ctx& c, long v) { c.wsort_leaf = v == 1024 ? 1024 : 2048; } },
    { "wsort_pack",       [](Ctx& c, long v) { c.wsort_pack = (v == 1024 || v == 4096) ? (int)v : 2048; } },
    { "wsort_cmax",       [](Ctx& c, long v) { c.wsort_cmax = clampi(v, 8, 64); } },       // (the hand-over lists take 128 n / (cmax + 1) bytes: below 8 they outgrow the arena)
    { "wsort_log",        [](Ctx& c, long v) { c.wsort_log = v != 0; } },
    { "upload_chunks",    [](Ctx& c, long v) { c.upload_chunks = clampi(v, 4, 24); } },
    { "upload_tail_n",    [](Ctx& c, long v) { c.upload_tail_n = clampi(v, 0, 12); } },
    { "upload_tail_pct",  [](Ctx& c, long v) { c.upload_tail_pct = clampi(v, 30, 100); } },
    { "arena_log",        [](Ctx& c, long v) { c.arena_log = v != 0; } },
};
constexpr size_t NOPTIONS = sizeof(OPTIONS) / sizeof(OPTIONS[0]);
const OptionDef* find_option(const char* name) {
    if (!name) return nullptr;
    if (!strncasecmp(name, "TDC_GPU_", 8)) name += 8;
    for (size_t i = 0; i < NOPTIONS; ++i) if (!strcasecmp(name, OPTIONS[i].name)) return &OPTIONS[i];
    return nullptr;
}
// the ONE place that reads TDC_GPU_* variables (besides TDC_GPU_LIB of the Python loader, which picks the library file)
void apply_env_options(tdc_gpu_ctx* ctx) {
    const char* on = getenv("TDC_GPU_DEBUG_KNOBS");
    if (!on || atoi(on) == 0) return;
    for (size_t i = 0; i < NOPTIONS; ++i) {
        char var[64] = "TDC_GPU_";
        size_t k = 8;
        for (const char* q = OPTIONS[i].name; *q && k + 1 < sizeof(var); ++q) var[k++] = (char)toupper((unsigned char)*q);
        var[k] = 0;
        if (const char* m = getenv(var)) OPTIONS[i].set(ctx->c, atol(m));
    }
}
}  // namespace

int tdc_gpu_ctx_set_option(tdc_gpu_ctx* ctx, const char* name, long value) {
    if (!ctx) return TDC_GPU_ERR_ARG;
    const OptionDef* o = find_option(name);
    if (!o) return TDC_GPU_ERR_ARG;
    o->set(ctx->c, value);
    return TDC_GPU_OK;
}
int tdc_gpu_option_count(void) { return (int)NOPTIONS; }
const char* tdc_gpu_option_name(int i) { return (i >= 0 && (size_t)i < NOPTIONS) ? OPTIONS[i].name : nullptr; }

int tdc_gpu_ctx_create(int device, tdc_gpu_ctx** out) {
    if (!out) return TDC_GPU_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return TDC_GPU_ERR_HIP; }
    if (device < 0 || device >= count) return TDC_GPU_ERR_ARG;
    tdc_gpu_ctx* ctx = new (std::nothrow) tdc_gpu_ctx();
    if (!ctx) return TDC_GPU_ERR_OOM;
    // libstdc++ drift check: a C++ library whose heap / sort tie order differs from the reference build's would change every
    // Huffman stream silently (coders/HuffmanCoder.hpp:455 is an unstable std::sort).  Only what builds a Huffman table depends on
    // it: those calls fail (encode.hip), everything else -- other coders, lz78, decompression -- works
    ctx->c.huff_ok = huffman_selfcheck();
    ctx->c.device = device;
    ctx->c.wpre = &ctx->pre;
    DeviceGuard dg(device);                      // (the caller's current device is restored on every exit path)
    try {
        HIP_TRY(dg.enter());
        HIP_TRY(hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking));
        for (auto& e : ctx->c.ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipStreamCreateWithFlags(&ctx->c.copy_stream, hipStreamNonBlocking));
        {   // the side stream takes the lowest priority the device offers
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; (void)hipGetLastError(); }
            if (hipStreamCreateWithPriority(&ctx->c.aux_stream, hipStreamNonBlocking, lo) != hipSuccess) { ctx->c.aux_stream = nullptr; (void)hipGetLastError(); }
            for (auto& e : ctx->c.ev_aux) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        for (auto& e : ctx->c.ev_copy) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->c.pinned_size = 4096;
        HIP_TRY(hipHostMalloc(&ctx->c.pinned, ctx->c.pinned_size, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&ctx->c.pinned_hdr, Ctx::PINNED_HDR, hipHostMallocDefault));
        if (ctx->c.fast_read) {
            void* zc = nullptr;
            if (hipHostMalloc(&zc, (size_t)Ctx::ZC_WORDS * Ctx::ZC_BLOCKS * 4, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
                void* dv = nullptr;
                if (hipHostGetDevicePointer(&dv, zc, 0) == hipSuccess) { ctx->c.zc_host = (u32*)zc; ctx->c.zc_dev = (u32*)dv; memset(zc, 0, (size_t)Ctx::ZC_WORDS * Ctx::ZC_BLOCKS * 4); }
                else { (void)hipHostFree(zc); (void)hipGetLastError(); }
            } else (void)hipGetLastError();
        }
        HIP_TRY(hipMalloc((void**)&ctx->c.d_err, 256));
        HIP_TRY(hipMemset(ctx->c.d_err, 0, 256));
        // Development aid: with TDC_GPU_DEBUG_KNOBS=1 every TDC_GPU_<OPTION> variable of the environment is applied through
        // tdc_gpu_ctx_set_option().  Without it the library never reads an option from the environment: an embedding process cannot change
        // the algorithm by accident.
        apply_env_options(ctx);
    } catch (const HipError&) {
        (void)hipGetLastError();
        tdc_gpu_ctx_destroy(ctx);
        return TDC_GPU_ERR_HIP;
    }
    *out = ctx;
    return TDC_GPU_OK;
}

void tdc_gpu_ctx_destroy(tdc_gpu_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard dg(ctx->c.device);
    (void)dg.enter();
    if (ctx->c.stream) (void)hipStreamSynchronize(ctx->c.stream);
    if (ctx->c.arena.base) (void)hipFree(ctx->c.arena.base);
    if (ctx->c.pinned) (void)hipHostFree(ctx->c.pinned);
    if (ctx->c.pinned_hdr) (void)hipHostFree(ctx->c.pinned_hdr);
    if (ctx->c.pinned_tab) (void)hipHostFree(ctx->c.pinned_tab);
    if (ctx->c.zc_host) (void)hipHostFree(ctx->c.zc_host);
    if (ctx->c.d_err) (void)hipFree(ctx->c.d_err);
    for (auto& e : ctx->c.ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->c.ev_copy) if (e) (void)hipEventDestroy(e);
    if (ctx->c.copy_stream) { (void)hipStreamSynchronize(ctx->c.copy_stream); (void)hipStreamDestroy(ctx->c.copy_stream); }
    if (ctx->c.aux_stream) { (void)hipStreamSynchronize(ctx->c.aux_stream); (void)hipStreamDestroy(ctx->c.aux_stream); }
    for (auto& e : ctx->c.ev_aux) if (e) (void)hipEventDestroy(e);
    if (ctx->c.ev_pool) { for (int i = 0; i < ctx->c.ev_pool_size; ++i) if (ctx->c.ev_pool[i]) (void)hipEventDestroy(ctx->c.ev_pool[i]); free(ctx->c.ev_pool); }
    free(ctx->c.pend);
    if (ctx->c.stream) (void)hipStreamDestroy(ctx->c.stream);
    delete ctx;
}

int tdc_gpu_ctx_set_profiling(tdc_gpu_ctx* ctx, int enabled) {
    return guarded(ctx, [&] {
        Ctx& c = ctx->c;
        if (enabled && !c.ev_pool) {
            c.ev_pool_size = 16384; c.pend_cap = 8192;
            c.ev_pool = (hipEvent_t*)calloc(c.ev_pool_size, sizeof(hipEvent_t));
            c.pend = (Ctx::Pending*)calloc(c.pend_cap, sizeof(Ctx::Pending));
            if (!c.ev_pool || !c.pend) throw std::bad_alloc();
            for (int i = 0; i < c.ev_pool_size; ++i) HIP_TRY(hipEventCreate(&c.ev_pool[i]));
        }
        c.profiling = enabled != 0;
    });
}

void tdc_gpu_ctx_reset_profile(tdc_gpu_ctx* ctx) {
    if (!ctx) return;
    for (auto& k : ctx->c.kprof) k = KernelProfile();
}

const char* tdc_gpu_ctx_kernel_profile(const tdc_gpu_ctx* ctx, int idx, double* ms, uint64_t* launches, uint64_t* bytes) {
    static const char* names[K_CLASS_COUNT] = {
        "rs_scatter_kernel<u64>", "rs_scatter_kernel<u32>", "rs_count_kernel", "scan_kernels",
        "sa_groups_kernel", "sa_build_keys_kernel", "phi_kernel", "plcp_kernel", "cand_kernels",
        "level_init_kernel", "mis_round_kernel", "resolve_kernel", "push_kernel", "apply_kernel", "pool_kernels", "small_level_kernel", "window_levels_kernel",
        "flatten_round_kernel", "gaps_kernel", "literal_hist_kernel", "tile_bits_kernel", "pack_kernel", "extract_kernels",
        "ss_leaf_sort_kernel", "sa_local_sort_kernel", "window_scatter_kernels",
        "ws_leaf_sort_kernel", "ws_leaf_count_kernel", "ws_run_kernels", "fs_image_kernel" };
    if (!ctx || idx < 0 || idx >= K_CLASS_COUNT) return nullptr;
    const KernelProfile& k = ctx->c.kprof[idx];
    if (ms) *ms = k.ms;
    if (launches) *launches = k.launches;
    if (bytes) *bytes = k.bytes;
    return names[idx];
}

size_t tdc_gpu_arena_bytes(size_t n) { return arena_need(n); }

int tdc_gpu_device_memory(int device, size_t* free_bytes, size_t* total_bytes) {
    if (!free_bytes || !total_bytes) return TDC_GPU_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { (void)hipGetLastError(); return TDC_GPU_ERR_ARG; }
    DeviceGuard dg(device);
    if (dg.enter() != hipSuccess || hipMemGetInfo(free_bytes, total_bytes) != hipSuccess) { (void)hipGetLastError(); return TDC_GPU_ERR_HIP; }
    return TDC_GPU_OK;
}

int tdc_gpu_ctx_reserve(tdc_gpu_ctx* ctx, size_t n) {
    return guarded(ctx, [&] { reserve_arena(ctx->c, arena_need(ctx->c, n)); });
}

size_t tdc_gpu_lcpcomp_bound(size_t n) { return align_up(encode_bound(n) + 16, 8); }
size_t tdc_gpu_lcpcomp_bound_coder(size_t n, int coder) {
    try { return align_up(encode_bound_coder(n, lcpcomp_enc_coder(coder)) + 16, 8); } catch (...) { return 0; }
}

int tdc_gpu_lcpcomp_compress_dev(tdc_gpu_ctx* ctx, const void* d_text, size_t n, uint32_t threshold, int flatten, int coder,
                                 void* d_out, size_t out_cap, size_t* out_len, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] {
        (void)lcpcomp_enc_coder(coder);
        check_text_args(d_text, n);
        if (!d_out || !out_len || ((uintptr_t)d_out & 7)) throw ArgError{TDC_GPU_ERR_ARG, "d_out must be non-NULL and 8-byte aligned"};
        Ctx& c = ctx->c;
        if (stats) memset(stats, 0, sizeof(*stats));
        reserve_arena(c, arena_need(c, n));
        Events ev(c);
        const int e0 = ev.tick();
        u8* dst = (u8*)d_out;
        *out_len = run_pipeline(c, (const u8*)d_text, n, threshold, flatten, coder, &dst, out_cap, stats, ev);
        const int e1 = ev.tick();
        if (stats) ev.span(&stats->ms_total, e0, e1);
        ev.finish();
    });
}

namespace {
// Host buffers in, host buffer out: H2D, (escape,) the whole pipeline, D2H.  The output goes either into a malloc'd buffer
// (*ho.out) or into the caller's buffer ho.into of ho.cap bytes; copies from / to pinned memory (tdc_gpu_host_alloc) run at
// PCIe speed, pageable memory is staged by the runtime.
struct HostOut { uint8_t** out; uint8_t* into; size_t cap; size_t* out_len; bool keep = false; };   // keep: the stream stays on the device (ctx->kept)
void compress_host(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, bool raw, uint32_t threshold, int flatten, int coder, int comp,
                   HostOut ho, tdc_gpu_stats* stats) {
    (void)lcpcomp_enc_coder(coder);
    if (comp != TDC_GPU_COMP_ARRAYS && comp != TDC_GPU_COMP_PLCPPEAKS && comp != TDC_GPU_COMP_MAXLCP && comp != TDC_GPU_COMP_HEAP)
        throw ArgError{TDC_GPU_ERR_UNSUPPORTED, "lcpcomp: comp must be arrays, plcppeaks, max_lcp or heap"};
    if (!ho.out_len || (!ho.out && !ho.into && !ho.keep)) throw ArgError{TDC_GPU_ERR_ARG, "out/out_len is NULL"};
    if (raw) {
        if (!text && n) throw ArgError{TDC_GPU_ERR_ARG, "NULL argument"};
        if (n >= 0x7FFFFFFEull) throw ArgError{TDC_GPU_ERR_TOO_LARGE, "raw input too large: the escaped text must stay < 2^31 - 1 bytes"};
    } else {
        check_text_args(text, n);
        if (text[n - 1] != 0) throw ArgError{TDC_GPU_ERR_NO_SENTINEL, "text does not end with a 0 sentinel"};
    }
    Ctx& c = ctx->c;
    if (stats) memset(stats, 0, sizeof(*stats));
    // raw input: sized for a text without escapes first; the 0x00 / 0xFF bytes are counted on the device after the upload
    reserve_arena(c, raw ? arena_need(c, n + 1) + n + 64 : arena_need(c, n));
    Events ev(c);
    const int e0 = ev.tick();
    u8* d_text;
    size_t tn = n;
    if (raw) {
        u8* d_raw = c.arena.get<u8>(n + 64);
        if (n) HIP_TRY(hipMemcpyAsync(d_raw, text, n, hipMemcpyHostToDevice, c.stream));
        tn = n + count_escapes_device(c, d_raw, n) + 1;
        if (tn >= 0x7FFFFFFFull) throw ArgError{TDC_GPU_ERR_TOO_LARGE, "raw input too large: the escaped text must stay < 2^31 - 1 bytes"};
        if (c.arena.size < arena_need(c, tn) + n + 64) {                        // many escapes: a larger arena, upload once more
            HIP_TRY(hipStreamSynchronize(c.stream));
            reserve_arena(c, arena_need(c, tn) + n + 64);
            d_raw = c.arena.get<u8>(n + 64);
            if (n) HIP_TRY(hipMemcpyAsync(d_raw, text, n, hipMemcpyHostToDevice, c.stream));
        }
        d_text = c.arena.get<u8>(tn + 64);
        if (escape_device(c, d_raw, n, d_text) != tn) throw HipError{hipErrorUnknown, "escape: length mismatch", (int)__LINE__};
    } else {
        d_text = c.arena.get<u8>(n + 64);
        if (n >= ((size_t)1 << 26) && c.copy_stream) {
            // The upload in chunks on the copy stream.  Behind every chunk, on the compute stream: its byte histogram (sentinel check,
            // symbol codes) and -- texts that take the wide suffix sort -- level 1 of that sort for the chunk in front of it (a key reads
            // up to 64 bytes ahead), with the code map and the splitters taken from chunk 0 (prim.hpp WPre).  All copies are queued
            // first: the one host wait in between (the histogram of chunk 0) does not stall them.
            u32* d_hist = c.arena.get<u32>(256);
            HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), c.stream));
            HIP_TRY(hipEventRecord(c.ev_copy[8], c.stream));                   // (the copy stream starts behind whatever the compute stream did before)
            HIP_TRY(hipStreamWaitEvent(c.copy_stream, c.ev_copy[8], 0));
            const bool try_pre = c.wsort_overlap && c.wpre && wsort_applicable(c, n);
            const size_t CH = try_pre ? (size_t)c.upload_chunks : 8;      // (at most 24: ev_copy[16 ..])
            // Chunk boundaries (multiples of 4096).  With level 1 behind the copies the last three chunks shrink geometrically (0.6, 0.36,
            // 0.22 of the others): level 1 of a chunk runs 1.7 x as fast as its copy, so each of them is done before the next, shorter copy
            // ends, and what is left behind the last copy is the level 1 of a fifth of a chunk.
            std::vector<size_t> coff;
            {
                std::vector<double> w(CH, 1.0);
                if (try_pre) {                                   // (the last chunks shrink geometrically: options upload_tail_n / upload_tail_pct)
                    const size_t T = std::min<size_t>((size_t)c.upload_tail_n, CH - 1);
                    double f = 1.0;
                    for (size_t k = 0; k < T; ++k) { f *= (double)c.upload_tail_pct / 100.0; w[CH - T + k] = f; }
                }
                double tot = 0; for (double x : w) tot += x;
                coff.push_back(0);
                double acc = 0;
                for (size_t k = 0; k + 1 < CH; ++k) {
                    acc += w[k];
                    size_t o = ((size_t)((double)n * (acc / tot)) + 4095) & ~(size_t)4095;
                    if (o <= coff.back()) o = coff.back() + 4096;
                    if (o >= n) break;
                    coff.push_back(o);
                }
                coff.push_back(n);
            }
            const size_t nch = coff.size() - 1;
            if (nch > 24) throw HipError{hipErrorUnknown, "upload: more chunks than copy events (ev_copy[16 .. 39])", (int)__LINE__};
            size_t queued = 0;                                                  // copies handed to the copy stream so far
            auto queue_copies = [&](size_t upto) {                              // (a few chunks ahead of the compute stream's work, not all at once:
                for (; queued < nch && queued < upto; ++queued) {               //  the runtime batches what it is given in one go)
                    // (a copy takes the first 64 bytes of the next chunk along -- a key reads that far ahead --, so level 1 of a chunk
                    //  can start as soon as the chunk itself is there: behind the last copy one chunk's level 1 is left, not two)
                    //  -- and a copy starts behind the 64 bytes its predecessor delivered: no byte is written twice while level 1 reads it)
                    const size_t off = coff[queued] + (queued ? 64 : 0), end = std::min(coff[queued + 1] + 64, n);
                    if (end > off) HIP_TRY(hipMemcpyAsync(d_text + off, text + off, end - off, hipMemcpyHostToDevice, c.copy_stream));
                    HIP_TRY(hipEventRecord(c.ev_copy[16 + queued], c.copy_stream));
                    (void)hipStreamQuery(c.copy_stream);                        // (submit now)
                }
            };
            bool pre_on = false;
            for (size_t q = 0; q < nch; ++q) {
                const size_t off = coff[q], len = coff[q + 1] - off;
                queue_copies(q + 4);
                HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_copy[16 + q], 0));
                text_histogram_add(c, d_text + off, len, d_hist);
                if (q == 0 && try_pre) {
                    u32 h0[256];
                    c.read_n(d_hist, h0, 256);                                 // (waits for chunk 0 only; chunks 1 .. 3 are on their way)
                    pre_on = wsort_pre_begin(c, *c.wpre, d_text, n, coff.data(), (u32)nch, h0);
                    if (!pre_on) c.arena.release_top();
                }
                if (pre_on) wsort_pre_chunk(c, *c.wpre, (u32)q);
            }
            text_histogram_finish(c, d_text, n, d_hist);
            if (pre_on) {
                wsort_pre_finish(c, *c.wpre, c.hist_cache);
                if (!c.wpre->active) c.arena.release_top();                      // chunk 0 did not show every byte value: the classic order of things
            }
        } else HIP_TRY(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c.stream));
    }
    const int e1 = ev.tick();
    u8* d_out = nullptr;
    struct SinkGuard {       // on every exit path: no copy into the caller's buffer is still in flight, the sink is forgotten
        Ctx& c;
        ~SinkGuard() { if (c.d2h_done && c.copy_stream) (void)hipStreamSynchronize(c.copy_stream); c.d2h_host = nullptr; c.d2h_cap = 0; c.d2h_done = 0; }
    } sink_guard{c};
    c.d2h_host = ho.into; c.d2h_cap = ho.into ? ho.cap : 0; c.d2h_done = 0;      // the encoder may start the D2H while it still packs
    struct PreGuard { Ctx& c; ~PreGuard() { if (c.wpre) { c.wpre->active = false; c.wpre->begun = false; } if (c.aux_stream) (void)hipStreamSynchronize(c.aux_stream); c.arena.release_top(); } } pre_guard{c};   // (the side stream is idle by now unless the call failed half-way)
    const size_t len = run_pipeline(c, d_text, tn, threshold, flatten, coder, &d_out, 0, stats, ev, comp);
    const int e2 = ev.tick();
    *ho.out_len = len;
    if (ho.keep) {
        if (stats) { ev.span(&stats->ms_h2d, e0, e1); ev.span(&stats->ms_total, e0, e2); }
        ev.finish();
        ctx->kept = d_out; ctx->kept_len = len;
    } else if (ho.into) {
        if (len > ho.cap) throw ArgError{TDC_GPU_ERR_OOM, "output buffer too small (*out_len holds the required size)"};
        const size_t done = c.d2h_done <= len ? c.d2h_done : 0;
        HIP_TRY(hipMemcpyAsync(ho.into + done, d_out + done, len - done, hipMemcpyDeviceToHost, c.stream));
        if (done) {                                                          // the front part travels on the copy stream
            HIP_TRY(hipEventRecord(c.ev_copy[9], c.copy_stream));
            HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_copy[9], 0));
        }
        const int e3 = ev.tick();
        if (stats) { ev.span(&stats->ms_h2d, e0, e1); ev.span(&stats->ms_d2h, e2, e3); ev.span(&stats->ms_total, e0, e3); }
        ev.finish();
    } else {
        HostBuf h(len);
        HIP_TRY(hipMemcpyAsync(h.p, d_out, len, hipMemcpyDeviceToHost, c.stream));
        const int e3 = ev.tick();
        if (stats) { ev.span(&stats->ms_h2d, e0, e1); ev.span(&stats->ms_d2h, e2, e3); ev.span(&stats->ms_total, e0, e3); }
        ev.finish();
        *ho.out = h.release<uint8_t>();
    }
}
}  // namespace

int tdc_gpu_lcpcomp_compress(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                             uint8_t** out, size_t* out_len, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] { compress_host(ctx, text, n, false, threshold, flatten, coder, TDC_GPU_COMP_ARRAYS, HostOut{out, nullptr, 0, out_len}, stats); });
}

int tdc_gpu_lcpcomp_compress_comp(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                                  int comp, uint8_t** out, size_t* out_len, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] { compress_host(ctx, text, n, false, threshold, flatten, coder, comp, HostOut{out, nullptr, 0, out_len}, stats); });
}

int tdc_gpu_lcpcomp_compress_into(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                                  int comp, uint8_t* out, size_t out_cap, size_t* out_len, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] {
        if (!out) throw ArgError{TDC_GPU_ERR_ARG, "out is NULL"};
        compress_host(ctx, text, n, false, threshold, flatten, coder, comp, HostOut{nullptr, out, out_cap, out_len}, stats);
    });
}

int tdc_gpu_lcpcomp_compress_keep(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                                  int comp, size_t* out_len, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] {
        HostOut ho{nullptr, nullptr, 0, out_len};
        ho.keep = true;
        compress_host(ctx, text, n, false, threshold, flatten, coder, comp, ho, stats);
    });
}

int tdc_gpu_stream_fetch(tdc_gpu_ctx* ctx, uint8_t* dst, size_t cap, size_t* len) {
    if (!ctx) return TDC_GPU_ERR_ARG;
    const u8* kept = ctx->kept;
    const size_t kept_len = ctx->kept_len;
    const int rc = guarded(ctx, [&] {
        if (!kept) throw ArgError{TDC_GPU_ERR_ARG, "no stream is kept on this context (tdc_gpu_lcpcomp_compress_keep, and no other call since)"};
        if (len) *len = kept_len;
        if (!dst || cap < kept_len) throw ArgError{TDC_GPU_ERR_OOM, "destination too small (*len holds the stream length)"};
        if (kept_len) HIP_TRY(hipMemcpyAsync(dst, kept, kept_len, hipMemcpyDeviceToHost, ctx->c.stream));
        HIP_TRY(hipStreamSynchronize(ctx->c.stream));
    });
    ctx->kept = kept; ctx->kept_len = kept_len;      // (may be fetched again)
    return rc;
}

int tdc_gpu_stream_fetch_dev(tdc_gpu_ctx* ctx, void* d_dst, size_t cap, size_t* len) {
    if (!ctx) return TDC_GPU_ERR_ARG;
    const u8* kept = ctx->kept;
    const size_t kept_len = ctx->kept_len;
    const int rc = guarded(ctx, [&] {
        if (!kept) throw ArgError{TDC_GPU_ERR_ARG, "no stream is kept on this context (tdc_gpu_lcpcomp_compress_keep, and no other call since)"};
        if (len) *len = kept_len;
        if (!d_dst || cap < kept_len) throw ArgError{TDC_GPU_ERR_OOM, "destination too small (*len holds the stream length)"};
        if (kept_len) HIP_TRY(hipMemcpyAsync(d_dst, kept, kept_len, hipMemcpyDeviceToDevice, ctx->c.stream));
        HIP_TRY(hipStreamSynchronize(ctx->c.stream));
    });
    ctx->kept = kept; ctx->kept_len = kept_len;      // (may be fetched again)
    return rc;
}

int tdc_gpu_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return TDC_GPU_ERR_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return TDC_GPU_ERR_HIP; }
    return TDC_GPU_OK;
}
int tdc_gpu_host_unregister(void* p) {
    if (!p) return TDC_GPU_ERR_ARG;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return TDC_GPU_ERR_HIP; }
    return TDC_GPU_OK;
}

int tdc_gpu_lcpcomp_compress_raw(tdc_gpu_ctx* ctx, const uint8_t* data, size_t n, uint32_t threshold, int flatten, int coder,
                                 uint8_t** out, size_t* out_len, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] { compress_host(ctx, data, n, true, threshold, flatten, coder, TDC_GPU_COMP_ARRAYS, HostOut{out, nullptr, 0, out_len}, stats); });
}

int tdc_gpu_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count;
}

void* tdc_gpu_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void tdc_gpu_host_free(void* p) { if (p) (void)hipHostFree(p); }

namespace {
// shared front end of the two lzss_lcp entry points: text to the device, SA + ISA, factorization into position space
void run_lzss_lcp(Ctx& c, const uint8_t* text, size_t n, uint32_t threshold, u8** d_text_out, DevArrays& A, tdc_gpu_stats* st, Events& ev) {
    if (threshold == 0) throw ArgError{TDC_GPU_ERR_ARG, "threshold must be >= 1"};
    if (text[n - 1] != 0) throw ArgError{TDC_GPU_ERR_NO_SENTINEL, "text does not end with a 0 sentinel"};
    reserve_arena(c, arena_need(c, n));
    u8* d_text = c.arena.get<u8>(n + 64);
    HIP_TRY(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c.stream));
    validate_device_text(c, d_text, n);
    A.sa = c.arena.get<u32>(n);
    A.isa = c.arena.get<u32>(n);
    A.fs.flen = c.arena.get<u32>(n); A.fs.owner = c.arena.get<u32>(n); A.fs.fsrc = c.arena.get<u32>(n);
    SAStats ss;
    const int e0 = ev.tick();
    build_suffix_array(c, d_text, n, A.sa, A.isa, &ss);
    const int e1 = ev.tick();
    LzssStats ls;
    lzss_lcp_factorize(c, d_text, n, A.sa, A.isa, threshold, A.fs, &ls);
    const int e2 = ev.tick();
    if (st) {
        st->n = n; st->factors = ls.factors; st->sa_rounds = ss.rounds; st->sa_init_syms = ss.init_syms; st->sa_sorted_elems = ss.sorted_elems;
        ev.span(&st->ms_sa, e0, e1); ev.span(&st->ms_factorize, e1, e2);
    }
    *d_text_out = d_text;
}
}  // namespace

int tdc_gpu_lzss_lcp_compress(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int coder,
                              uint8_t** out, size_t* out_len, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] {
        if (coder != TDC_GPU_CODER_HUFF) throw ArgError{TDC_GPU_ERR_UNSUPPORTED, "lzss_lcp: only coder=huff is built"};
        check_text_args(text, n);
        if (!out || !out_len) throw ArgError{TDC_GPU_ERR_ARG, "out/out_len is NULL"};
        Ctx& c = ctx->c;
        if (stats) memset(stats, 0, sizeof(*stats));
        Events ev(c);
        DevArrays A;
        u8* d_text = nullptr;
        const int e0 = ev.tick();
        run_lzss_lcp(c, text, n, threshold, &d_text, A, stats, ev);
        const size_t cap = tdc_gpu_lcpcomp_bound(n);
        u8* d_out = c.arena.get<u8>(cap);
        EncodeStats es;
        const int e1 = ev.tick();
        const size_t len = encode_huff(c, d_text, n, A.fs, d_out, cap, &es);
        const int e2 = ev.tick();
        HostBuf h(len);
        HIP_TRY(hipMemcpyAsync(h.p, d_out, len, hipMemcpyDeviceToHost, c.stream));
        const int e3 = ev.tick();
        if (stats) {
            stats->out_len = len; stats->flen_min = es.flen_min; stats->flen_max = es.flen_max; stats->fdist_max = es.fdist_max;
            stats->sigma = es.sigma; stats->arena_bytes = c.arena.high;
            ev.span(&stats->ms_encode, e1, e2); ev.span(&stats->ms_d2h, e2, e3); ev.span(&stats->ms_total, e0, e3);
        }
        ev.finish();
        *out = h.release<uint8_t>(); *out_len = len;
    });
}

int tdc_gpu_lzss_lcp_factorize(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold,
                               uint32_t** pos, uint32_t** src, uint32_t** len, size_t* z) {
    return guarded(ctx, [&] {
        check_text_args(text, n);
        if (!pos || !src || !len || !z) throw ArgError{TDC_GPU_ERR_ARG, "output pointer is NULL"};
        Ctx& c = ctx->c;
        Events ev(c);
        DevArrays A;
        u8* d_text = nullptr;
        run_lzss_lcp(c, text, n, threshold, &d_text, A, nullptr, ev);
        u32* d_pos = c.arena.get<u32>(n), *d_src = c.arena.get<u32>(n), *d_len = c.arena.get<u32>(n);
        const size_t cnt = extract_factors(c, n, A.fs, d_pos, d_src, d_len, n);
        HostBuf hp(cnt * 4), hs(cnt * 4), hl(cnt * 4);
        if (cnt) {
            HIP_TRY(hipMemcpyAsync(hp.p, d_pos, cnt * 4, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(hs.p, d_src, cnt * 4, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(hl.p, d_len, cnt * 4, hipMemcpyDeviceToHost, c.stream));
        }
        ev.finish();
        *pos = hp.release<uint32_t>(); *src = hs.release<uint32_t>(); *len = hl.release<uint32_t>(); *z = cnt;
    });
}

int tdc_gpu_lz78_compress(tdc_gpu_ctx* ctx, const uint8_t* in, size_t n, int coder, uint8_t** out, size_t* out_len,
                          tdc_gpu_stats* stats) {
    return guarded(ctx, [&] {
        if (coder != TDC_GPU_CODER_GAMMA) throw ArgError{TDC_GPU_ERR_UNSUPPORTED, "lz78: only coder=gamma is built"};
        if ((!in && n) || !out || !out_len) throw ArgError{TDC_GPU_ERR_ARG, "NULL argument"};
        if (n >= 0xFFFFFFFFull) throw ArgError{TDC_GPU_ERR_TOO_LARGE, "lz78: input must be < 2^32 bytes"};
        Ctx& c = ctx->c;
        if (stats) memset(stats, 0, sizeof(*stats));
        std::vector<u32> ids;
        std::vector<u8> chars;
        bool high = false;
        const size_t z = lz78_parse_host(in, n, ids, chars, &high);
        if (high) throw ArgError{TDC_GPU_ERR_UNSUPPORTED,
            "lz78: the left-over phrase ends in a byte >= 0x80; the reference encodes it as a signed char (undefined shifts) -- not reproduced"};
        // arena: pairs (5 B each) + tile sums + worst-case output (2*33+2*9 bits = 84 bits < 11 B per pair)
        const size_t cap = align_up(z * 11 + 64, 8);
        reserve_arena(c, z * 5 + cap + ((size_t)64 << 20));
        Events ev(c);
        const int e0 = ev.tick();
        u32* d_ids = c.arena.get<u32>(z + 1);
        u8* d_chars = c.arena.get<u8>(z + 8);
        u8* d_out = c.arena.get<u8>(cap);
        if (z) {
            HIP_TRY(hipMemcpyAsync(d_ids, ids.data(), z * 4, hipMemcpyHostToDevice, c.stream));
            HIP_TRY(hipMemcpyAsync(d_chars, chars.data(), z, hipMemcpyHostToDevice, c.stream));
        }
        const int e1 = ev.tick();
        const size_t len = lz78_gamma_encode(c, d_ids, d_chars, z, d_out, cap);
        const int e2 = ev.tick();
        HostBuf h(len);
        HIP_TRY(hipMemcpyAsync(h.p, d_out, len, hipMemcpyDeviceToHost, c.stream));
        const int e3 = ev.tick();
        if (stats) {
            stats->n = n; stats->out_len = len; stats->factors = z; stats->arena_bytes = c.arena.high;
            ev.span(&stats->ms_h2d, e0, e1); ev.span(&stats->ms_encode, e1, e2); ev.span(&stats->ms_d2h, e2, e3); ev.span(&stats->ms_total, e0, e3);
        }
        ev.finish();
        *out = h.release<uint8_t>(); *out_len = len;
    });
}

int tdc_gpu_sort_pairs_u64(tdc_gpu_ctx* ctx, uint64_t* keys, uint32_t* vals, size_t n, int algo) {
    return guarded(ctx, [&] {
        if (!keys || !vals) throw ArgError{TDC_GPU_ERR_ARG, "keys/vals is NULL"};
        if (n == 0) return;
        if (n >= 0xFFFFFFFFull) throw ArgError{TDC_GPU_ERR_TOO_LARGE, "at most 2^32 - 2 pairs"};
        Ctx& c = ctx->c;
        reserve_arena(c, 64 * n + ((size_t)256 << 20));
        u64* k[2] = { c.arena.get<u64>(n), c.arena.get<u64>(n) };
        u32* v[2] = { c.arena.get<u32>(n), c.arena.get<u32>(n) };
        HIP_TRY(hipMemcpyAsync(k[0], keys, n * 8, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(hipMemcpyAsync(v[0], vals, n * 4, hipMemcpyHostToDevice, c.stream));
        int x;
        if (algo == 1) { SplitSortStats ss; x = splitter_sort_pairs_u64(c, k, v, n, nullptr, &ss); }
        else if (algo == 0) x = radix_sort_pairs_u64(c, k, v, n, 0, 64);
        else throw ArgError{TDC_GPU_ERR_ARG, "algo must be 0 (LSD radix) or 1 (splitter partition)"};
        HIP_TRY(hipMemcpyAsync(keys, k[x], n * 8, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(vals, v[x], n * 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    });
}

int tdc_gpu_suffix_array(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t* sa, uint32_t* isa) {
    return tdc_gpu_textds(ctx, text, n, sa, isa, nullptr, nullptr, nullptr, nullptr);
}

int tdc_gpu_textds(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t* sa, uint32_t* isa, uint32_t* phi,
                   uint32_t* plcp, uint32_t* lcp, uint32_t* maxlcp) {
    return guarded(ctx, [&] {
        check_text_args(text, n);
        if (text[n - 1] != 0) throw ArgError{TDC_GPU_ERR_NO_SENTINEL, "text does not end with a 0 sentinel"};
        Ctx& c = ctx->c;
        reserve_arena(c, arena_need(c, n));
        u8* d_text = c.arena.get<u8>(n + 64);
        HIP_TRY(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c.stream));
        validate_device_text(c, d_text, n);
        DevArrays A;
        if (!phi && !plcp && !lcp && !maxlcp) {
            A.sa = c.arena.get<u32>(n);
            A.isa = c.arena.get<u32>(n);
            build_suffix_array(c, d_text, n, A.sa, A.isa, nullptr);
        } else {
            run_textds(c, d_text, n, A, nullptr, nullptr);
        }
        if (sa) HIP_TRY(hipMemcpyAsync(sa, A.sa, n * 4, hipMemcpyDeviceToHost, c.stream));
        if (isa) HIP_TRY(hipMemcpyAsync(isa, A.isa, n * 4, hipMemcpyDeviceToHost, c.stream));
        if (phi) HIP_TRY(hipMemcpyAsync(phi, A.phi, n * 4, hipMemcpyDeviceToHost, c.stream));
        if (plcp) HIP_TRY(hipMemcpyAsync(plcp, A.plcp, n * 4, hipMemcpyDeviceToHost, c.stream));
        if (lcp) {
            u32* d_lcp = c.arena.get<u32>(n);
            build_lcp(c, A.sa, A.plcp, n, d_lcp);
            HIP_TRY(hipMemcpyAsync(lcp, d_lcp, n * 4, hipMemcpyDeviceToHost, c.stream));
        }
        if (maxlcp) *maxlcp = A.maxlcp;
        HIP_TRY(hipStreamSynchronize(c.stream));
    });
}

int tdc_gpu_lcpcomp_factorize(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                              uint32_t** pos, uint32_t** src, uint32_t** len, size_t* z, tdc_gpu_stats* stats) {
    return guarded(ctx, [&] {
        check_text_args(text, n);
        if (!pos || !src || !len || !z) throw ArgError{TDC_GPU_ERR_ARG, "output pointer is NULL"};
        if (threshold == 0) throw ArgError{TDC_GPU_ERR_ARG, "threshold must be >= 1"};
        if (text[n - 1] != 0) throw ArgError{TDC_GPU_ERR_NO_SENTINEL, "text does not end with a 0 sentinel"};
        Ctx& c = ctx->c;
        if (stats) memset(stats, 0, sizeof(*stats));
        reserve_arena(c, arena_need(c, n));
        Events ev(c);
        u8* d_text = c.arena.get<u8>(n + 64);
        HIP_TRY(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c.stream));
        validate_device_text(c, d_text, n);
        DevArrays A;
        run_textds(c, d_text, n, A, stats, &ev);
        run_factorize(c, n, A, threshold, flatten, stats, &ev);
        u32* d_pos = c.arena.get<u32>(n), *d_src = c.arena.get<u32>(n), *d_len = c.arena.get<u32>(n);
        const size_t cnt = extract_factors(c, n, A.fs, d_pos, d_src, d_len, n);
        HostBuf hp(cnt * 4), hs(cnt * 4), hl(cnt * 4);
        if (cnt) {
            HIP_TRY(hipMemcpyAsync(hp.p, d_pos, cnt * 4, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(hs.p, d_src, cnt * 4, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(hl.p, d_len, cnt * 4, hipMemcpyDeviceToHost, c.stream));
        }
        ev.finish();
        *pos = hp.release<uint32_t>(); *src = hs.release<uint32_t>(); *len = hl.release<uint32_t>(); *z = cnt;
        if (stats) { stats->n = n; stats->arena_bytes = c.arena.high; }
    });
}

int tdc_gpu_flatten(tdc_gpu_ctx* ctx, size_t n, const uint32_t* pos, uint32_t* src, const uint32_t* len, size_t z,
                    uint64_t* num_flattened, uint64_t* max_depth_lb) {
    return guarded(ctx, [&] {
        if (n == 0 || n >= 0x7FFFFFFFull) throw ArgError{TDC_GPU_ERR_ARG, "bad n"};
        if (z && (!pos || !src || !len)) throw ArgError{TDC_GPU_ERR_ARG, "factor arrays are NULL"};
        validate_factor_list(n, pos, src, len, z);
        Ctx& c = ctx->c;
        reserve_arena(c, arena_need(c, n));
        FactorSpace fs;
        fs.flen = c.arena.get<u32>(n); fs.owner = c.arena.get<u32>(n); fs.fsrc = c.arena.get<u32>(n);
        u32* d_pos = c.arena.get<u32>(z + 1), *d_src = c.arena.get<u32>(z + 1), *d_len = c.arena.get<u32>(z + 1);
        if (z) {
            HIP_TRY(hipMemcpyAsync(d_pos, pos, z * 4, hipMemcpyHostToDevice, c.stream));
            HIP_TRY(hipMemcpyAsync(d_src, src, z * 4, hipMemcpyHostToDevice, c.stream));
            HIP_TRY(hipMemcpyAsync(d_len, len, z * 4, hipMemcpyHostToDevice, c.stream));
        }
        scatter_factors(c, n, d_pos, d_src, d_len, z, fs);
        FlattenStats fl;
        flatten_factors(c, n, fs, &fl);
        const size_t cnt = extract_factors(c, n, fs, d_pos, d_src, nullptr, z + 1);
        if (cnt != z) throw HipError{hipErrorUnknown, "flatten: factor count changed", (int)__LINE__};
        if (z) HIP_TRY(hipMemcpyAsync(src, d_src, z * 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        if (num_flattened) *num_flattened = fl.num_flattened;
        if (max_depth_lb) *max_depth_lb = fl.max_depth_lb;
    });
}

int tdc_gpu_lcpcomp_decompress(tdc_gpu_ctx* ctx, const uint8_t* stream, size_t len, uint8_t** out, size_t* out_len,
                               uint64_t* factors, uint32_t* rounds) {
    return tdc_gpu_lcpcomp_decompress_coder(ctx, stream, len, TDC_GPU_CODER_HUFF, out, out_len, factors, rounds);
}

namespace {
void decompress_common(tdc_gpu_ctx* ctx, const uint8_t* stream, size_t len, int coder, DecodeOut& o, size_t* out_len, uint64_t* factors, uint32_t* rounds) {
    ctx->last_decode_device = 0;                             // (a failed call must not report the previous call's value)
    if (!stream || !out_len) throw ArgError{TDC_GPU_ERR_ARG, "NULL argument"};
    const int enc = lcpcomp_enc_coder(coder);
    if (enc == 1) throw ArgError{TDC_GPU_ERR_UNSUPPORTED, "lcpcomp(coder=arithmetic) streams cannot be decoded (neither can the reference)"};
    DecodeStats ds;
    size_t n = 0;
    try { n = decode_lzss(ctx->c, stream, len, enc, o, &ds); }
    catch (const StreamFormatError& e) { free(o.owned); o.owned = nullptr; throw ArgError{TDC_GPU_ERR_ARG, e.what}; }
    catch (...) { free(o.owned); o.owned = nullptr; throw; }
    *out_len = n;
    if (factors) *factors = ds.factors;
    if (rounds) *rounds = ds.rounds;
    ctx->last_decode_device = (int)ds.device_parse;
}
}  // namespace

int tdc_gpu_lcpcomp_decompress_coder(tdc_gpu_ctx* ctx, const uint8_t* stream, size_t len, int coder, uint8_t** out, size_t* out_len,
                                     uint64_t* factors, uint32_t* rounds) {
    return guarded(ctx, [&] {
        if (!out) throw ArgError{TDC_GPU_ERR_ARG, "NULL argument"};
        DecodeOut o;
        decompress_common(ctx, stream, len, coder, o, out_len, factors, rounds);
        *out = o.owned;
    });
}

int tdc_gpu_lcpcomp_decompress_into(tdc_gpu_ctx* ctx, const uint8_t* stream, size_t len, int coder, uint8_t* out, size_t out_cap,
                                    size_t* out_len, uint64_t* factors, uint32_t* rounds) {
    return guarded(ctx, [&] {
        if (!out) throw ArgError{TDC_GPU_ERR_ARG, "out is NULL"};
        DecodeOut o;
        o.into = out; o.cap = out_cap;
        decompress_common(ctx, stream, len, coder, o, out_len, factors, rounds);
    });
}

int tdc_gpu_ctx_last_decode_on_device(const tdc_gpu_ctx* ctx) { return ctx ? ctx->last_decode_device : 0; }

static int encode_entry(tdc_gpu_ctx* ctx, int coder, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                        const uint32_t* len, size_t z, uint8_t** out, size_t* out_len);
int tdc_gpu_encode_huff(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                        const uint32_t* len, size_t z, uint8_t** out, size_t* out_len) {
    return encode_entry(ctx, 0, text, n, pos, src, len, z, out, out_len);
}
int tdc_gpu_encode_arith(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                         const uint32_t* len, size_t z, uint8_t** out, size_t* out_len) {
    return encode_entry(ctx, 1, text, n, pos, src, len, z, out, out_len);
}
int tdc_gpu_encode_ascii(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                         const uint32_t* len, size_t z, uint8_t** out, size_t* out_len) {
    return encode_entry(ctx, 2, text, n, pos, src, len, z, out, out_len);
}
int tdc_gpu_encode_sle(tdc_gpu_ctx* ctx, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                       const uint32_t* len, size_t z, uint32_t kmer, uint8_t** out, size_t* out_len) {
    if (kmer > 7) return TDC_GPU_ERR_ARG;
    return encode_entry(ctx, 3 | ((int)(kmer ? kmer : 3) << 8), text, n, pos, src, len, z, out, out_len);
}
static int encode_entry(tdc_gpu_ctx* ctx, int coder, const uint8_t* text, size_t n, const uint32_t* pos, const uint32_t* src,
                        const uint32_t* len, size_t z, uint8_t** out, size_t* out_len) {
    return guarded(ctx, [&] {
        check_text_args(text, n);
        if (!out || !out_len) throw ArgError{TDC_GPU_ERR_ARG, "out/out_len is NULL"};
        if (z && (!pos || !src || !len)) throw ArgError{TDC_GPU_ERR_ARG, "factor arrays are NULL"};
        validate_factor_list(n, pos, src, len, z);
        Ctx& c = ctx->c;
        reserve_arena(c, arena_need(c, n));
        u8* d_text = c.arena.get<u8>(n + 64);
        HIP_TRY(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c.stream));
        FactorSpace fs;
        fs.flen = c.arena.get<u32>(n); fs.owner = c.arena.get<u32>(n); fs.fsrc = c.arena.get<u32>(n);
        u32* d_pos = c.arena.get<u32>(z + 1), *d_src = c.arena.get<u32>(z + 1), *d_len = c.arena.get<u32>(z + 1);
        if (z) {
            HIP_TRY(hipMemcpyAsync(d_pos, pos, z * 4, hipMemcpyHostToDevice, c.stream));
            HIP_TRY(hipMemcpyAsync(d_src, src, z * 4, hipMemcpyHostToDevice, c.stream));
            HIP_TRY(hipMemcpyAsync(d_len, len, z * 4, hipMemcpyHostToDevice, c.stream));
        }
        scatter_factors(c, n, d_pos, d_src, d_len, z, fs);
        const size_t cap = align_up(encode_bound_coder(n, coder) + 16, 8);
        u8* d_out = c.arena.get<u8>(cap);
        const size_t l = encode_stream(c, d_text, n, fs, coder, d_out, cap, nullptr);
        HostBuf h(l);
        HIP_TRY(hipMemcpyAsync(h.p, d_out, l, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        *out = h.release<uint8_t>(); *out_len = l;
    });
}

// ---- host-side helpers ------------------------------------------------------------------------------------
size_t tdc_escape(const uint8_t* in, size_t n, uint8_t* out) {
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t ch = in[i];
        if (ch == 0x00) { out[o++] = 0xFF; out[o++] = 0xFE; }
        else if (ch == 0xFF) { out[o++] = 0xFF; out[o++] = 0xFF; }
        else out[o++] = ch;
    }
    out[o++] = 0;
    return o;
}

size_t tdc_unescape(const uint8_t* in, size_t n, uint8_t* out) {
    size_t o = 0;
    if (n && in[n - 1] == 0) --n;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t ch = in[i];
        if (ch == 0xFF && i + 1 < n) { const uint8_t d = in[++i]; out[o++] = (d == 0xFE) ? 0x00 : d; }
        else out[o++] = ch;
    }
    return o;
}

int tdc_huffman_selfcheck(void) { return huffman_selfcheck() ? TDC_GPU_OK : TDC_GPU_ERR_INTERNAL; }

int tdc_huffman_table(const uint32_t counts[256], uint32_t* sigma, uint32_t* longest, uint8_t order[256],
                      uint8_t len_of[256], uint64_t code_of[256]) {
    if (!counts) return TDC_GPU_ERR_ARG;
    try {
        HuffTable t;
        build_huffman_table(counts, &t);
        if (sigma) *sigma = t.sigma;
        if (longest) *longest = t.longest;
        if (order) memcpy(order, t.order, 256);
        if (len_of) memcpy(len_of, t.len_of, 256);
        if (code_of) memcpy(code_of, t.code_of, 256 * sizeof(uint64_t));
    } catch (...) { return TDC_GPU_ERR_OOM; }
    return TDC_GPU_OK;
}

}  // extern "C"
