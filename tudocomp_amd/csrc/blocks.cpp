// blocks.cpp -- block mode of the C ABI (include/tdc_gpu.h): inputs above one-GPU size are cut into independent blocks, every
// block is a complete lcpcomp stream of its own (own sentinel / escaping, suffix array, factors, Huffman table), the blocks are
// spread over the visible devices -- one host thread and one context per device, blocks handed out from a shared counter -- and
// the per-block streams are framed in the container of SURVEY.md 8e:
//     "tdcgpu-blocks%" | u32 G | G x { u64 raw_len, u64 comp_len } | payload_0 | ... | payload_{G-1}        (little endian)
// The reference has no block mode (its 32-bit len_t caps an input at 2^31 - 1 bytes); every payload is
// byte-identical to what LCPCompressor::compress writes for that block alone (tudocomp_driver.cpp:231-276 with --raw).
// This file only orchestrates: it calls the C ABI's own entry points and owns no kernels.
#include "../../include/tdc_gpu.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <memory>
#include <thread>
#include <vector>

namespace {
const char MAGIC[] = "tdcgpu-blocks%";
constexpr size_t MAGIC_LEN = sizeof(MAGIC) - 1;
void put_u32(uint8_t* p, uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
void put_u64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
uint32_t get_u32(const uint8_t* p) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)p[i] << (8 * i); return v; }
uint64_t get_u64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }
}  // namespace

extern "C" {

size_t tdc_gpu_blocks_count(size_t n, size_t block_size) { return block_size ? (n + block_size - 1) / block_size : 0; }

int tdc_gpu_blocks_compress(const int* devices, int ndev, const uint8_t* data, size_t n, size_t block_size, uint32_t threshold, int flatten,
                            int coder, uint8_t** out, size_t* out_len, tdc_gpu_stats* per_block) {
    if (!out || !out_len || (!data && n) || ndev <= 0 || !devices) return TDC_GPU_ERR_ARG;
    if (block_size == 0 || block_size >= 0x7FFFFFFEull) return TDC_GPU_ERR_ARG;
    const size_t G = tdc_gpu_blocks_count(n, block_size);
    if (G > 0xFFFFFFFFull) return TDC_GPU_ERR_TOO_LARGE;
    // nothing may leave this function but a status code: every C++ failure (allocation, thread creation) is mapped to one, the workers
    // that did start are always joined, and the per-block streams are freed on every path
    try {
        std::vector<uint8_t*> streams(G, nullptr);
        struct FreeStreams { std::vector<uint8_t*>& s; ~FreeStreams() { for (uint8_t* p : s) tdc_gpu_free(p); } } free_streams{streams};
        std::vector<size_t> lens(G, 0);
        std::atomic<size_t> next(0);
        std::atomic<int> status(TDC_GPU_OK);
        auto fail = [&](int rc) { int ok = TDC_GPU_OK; status.compare_exchange_strong(ok, rc); };
        auto worker = [&](int device) noexcept {
            try {
                tdc_gpu_ctx* ctx = nullptr;
                int rc = tdc_gpu_ctx_create(device, &ctx);
                if (rc) { fail(rc); return; }
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= G || status.load() != TDC_GPU_OK) break;
                    const size_t off = k * block_size, len = (off + block_size <= n) ? block_size : n - off;
                    rc = tdc_gpu_lcpcomp_compress_raw(ctx, data + off, len, threshold, flatten, coder, &streams[k], &lens[k], per_block ? &per_block[k] : nullptr);
                    if (rc) { fail(rc); break; }
                }
                tdc_gpu_ctx_destroy(ctx);
            } catch (...) { fail(TDC_GPU_ERR_INTERNAL); }
        };
        // One worker (host thread + context) per listed device.  Where blocks outnumber the devices and a device has room for two
        // arenas, it gets a second worker: the upload of one block then overlaps the kernels of another (a context's calls are
        // synchronous; two contexts on one device run on their own streams).
        std::vector<int> slots(devices, devices + ndev);
        if (G > (size_t)ndev) {
            const size_t need = tdc_gpu_arena_bytes(block_size < n ? block_size + 1 : n + 1);
            for (int d = 0; d < ndev; ++d) {
                bool listed_before = false;
                for (int e = 0; e < d; ++e) listed_before = listed_before || devices[e] == devices[d];
                size_t free_b = 0, total_b = 0;
                if (!listed_before && tdc_gpu_device_memory(devices[d], &free_b, &total_b) == TDC_GPU_OK && free_b / 2 > need + (need >> 4)) slots.push_back(devices[d]);
            }
        }
        if (slots.size() > G) slots.resize(G ? G : 1);
        if (slots.size() <= 1 || G <= 1) worker(slots[0]);
        else {
            std::vector<std::thread> th;
            struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
            th.reserve(slots.size());
            for (size_t i = 0; i < slots.size(); ++i) {
                try { th.emplace_back(worker, slots[i]); }
                catch (...) { fail(TDC_GPU_ERR_INTERNAL); break; }            // (the workers already running are joined by the guard)
            }
        }
        int rc = status.load();
        size_t total = MAGIC_LEN + 4 + 16 * G;
        for (size_t k = 0; k < G; ++k) total += lens[k];
        // (owned until it is handed to the caller: an exception below -- the offset table, a copy thread -- must not leak gigabytes)
        struct Free { void operator()(uint8_t* p) const { free(p); } };
        std::unique_ptr<uint8_t, Free> blob_own;
        if (rc == TDC_GPU_OK) { blob_own.reset((uint8_t*)malloc(total ? total : 1)); if (!blob_own) rc = TDC_GPU_ERR_OOM; }
        if (rc == TDC_GPU_OK) {
            uint8_t* blob = blob_own.get();
            memcpy(blob, MAGIC, MAGIC_LEN);
            put_u32(blob + MAGIC_LEN, (uint32_t)G);
            size_t dir = MAGIC_LEN + 4, pay = dir + 16 * G;
            std::vector<size_t> at(G, 0);
            for (size_t k = 0; k < G; ++k) {
                const size_t off = k * block_size, len = (off + block_size <= n) ? block_size : n - off;
                put_u64(blob + dir, len); put_u64(blob + dir + 8, lens[k]); dir += 16;
                at[k] = pay; pay += lens[k];
            }
            // the payloads into place: a 2 GB block leaves 0.7 GB of stream, one thread copies that at ~8 GB/s -- several blocks are
            // copied side by side (the copies are independent).  Thread i owns the blocks i, i + nt, ..; if only `started` threads could
            // be created, the calling thread copies the blocks of the others -- never a block a running thread owns.
            const size_t nt = (total > ((size_t)1 << 20) && G > 1) ? (G < 8 ? G : 8) : 1;
            size_t started = 0;
            {
                std::vector<std::thread> ct;
                struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } cj{ct};
                if (nt > 1) {
                    try {
                        ct.reserve(nt);
                        for (size_t i = 0; i < nt; ++i) {
                            ct.emplace_back([&, i] { for (size_t k = i; k < G; k += nt) memcpy(blob + at[k], streams[k], lens[k]); });
                            ++started;
                        }
                    } catch (...) {}
                }
                for (size_t i = started; i < nt; ++i)
                    for (size_t k = i; k < G; k += nt) memcpy(blob + at[k], streams[k], lens[k]);
            }                                                                   // (copy threads joined here)
            *out = blob_own.release(); *out_len = total;
        }
        return rc;
    } catch (const std::bad_alloc&) {
        return TDC_GPU_ERR_OOM;
    } catch (...) {
        return TDC_GPU_ERR_INTERNAL;
    }
}

int tdc_gpu_blocks_decompress(tdc_gpu_ctx* ctx, const uint8_t* blob, size_t len, int coder, uint8_t** out, size_t* out_len) {
    if (!ctx || !blob || !out || !out_len) return TDC_GPU_ERR_ARG;
    if (len < MAGIC_LEN + 4 || memcmp(blob, MAGIC, MAGIC_LEN) != 0) return TDC_GPU_ERR_ARG;
    const size_t G = get_u32(blob + MAGIC_LEN);
    if (len < MAGIC_LEN + 4 + 16 * G) return TDC_GPU_ERR_ARG;
    size_t raw_total = 0, pay = MAGIC_LEN + 4 + 16 * G, at = pay;
    for (size_t k = 0; k < G; ++k) {
        const uint64_t raw = get_u64(blob + MAGIC_LEN + 4 + 16 * k), comp = get_u64(blob + MAGIC_LEN + 4 + 16 * k + 8);
        if (comp > len - at || raw >= 0x7FFFFFFEull) return TDC_GPU_ERR_ARG;
        raw_total += raw; at += comp;
    }
    if (at != len) return TDC_GPU_ERR_ARG;
    uint8_t* res = (uint8_t*)malloc(raw_total ? raw_total : 1);
    if (!res) return TDC_GPU_ERR_OOM;
    size_t o = 0;
    at = pay;
    for (size_t k = 0; k < G; ++k) {
        const uint64_t raw = get_u64(blob + MAGIC_LEN + 4 + 16 * k), comp = get_u64(blob + MAGIC_LEN + 4 + 16 * k + 8);
        uint8_t* text = nullptr; size_t tn = 0;
        const int rc = tdc_gpu_lcpcomp_decompress_coder(ctx, blob + at, comp, coder, &text, &tn, nullptr, nullptr);
        if (rc) { free(res); return rc; }
        std::vector<uint8_t> plain(tn + 1);
        const size_t pn = tdc_unescape(text, tn, plain.data());              // removes the block's input restrictions again
        tdc_gpu_free(text);
        if (pn != raw) { free(res); return TDC_GPU_ERR_ARG; }
        memcpy(res + o, plain.data(), pn);
        o += pn; at += comp;
    }
    *out = res; *out_len = raw_total;
    return TDC_GPU_OK;
}

}  // extern "C"
