// lz78.hip -- LZ78Compressor (compressors/LZ78Compressor.hpp:64-140) with EliasGammaCoder
// (coders/EliasGammaCoder.hpp:26-29, io/BitOStream.hpp:105-129); BASELINE.json configs[3].
//
// The LZ78 parse is inherently sequential (every step depends on the whole dictionary so far), so it stays on the
// host (option a): a hashed (parent, byte) -> child dictionary instead of the reference's trie
// back-ends -- all of them yield identical factor ids by contract (test/lz78_trie_tests.cpp:61-100).
// The coder side is data-parallel and runs on the GPU with the same cost / scan / pack scheme as encode.hip:
// gamma(v) = bits_for(v) zeros, "1", v in bits_for(v) bits; pair i contributes gamma(id_i) gamma(c_i).
#include "stages.hpp"
#include "prim.hpp"

#include <vector>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <new>

namespace tdc {

// ---- host: LZ78 parse ------------------------------------------------------------------------------------------
namespace {
struct PhraseTable {                 // open addressing: key = (parent << 8 | byte) + 1, value = child id
    struct Slot { u64 key; u32 val; u32 pad; };          // key and value in one 16-byte slot: a step down the trie is ONE cache miss
    // the table of a 1 GB input is gigabytes large and every step lands on a random page: huge pages where the kernel grants them
    struct Buf {
        Slot* p = nullptr; size_t n = 0;
        ~Buf() { free(p); }
        void alloc(size_t count) {
            free(p); p = nullptr; n = count;
            const size_t bytes = (count * sizeof(Slot) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            p = (Slot*)aligned_alloc((size_t)2 << 20, bytes);
            if (!p) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
            (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
            memset(p, 0, count * sizeof(Slot));
        }
        Slot& operator[](size_t i) { return p[i]; }
        const Slot& operator[](size_t i) const { return p[i]; }
        size_t size() const { return n; }
        void swap(Buf& o) { std::swap(p, o.p); std::swap(n, o.n); }
    } slots;
    u64 mask = 0;
    size_t used = 0;
    void init(size_t cap_pow2) { slots.alloc(cap_pow2); mask = cap_pow2 - 1; used = 0; }
    static u64 hash(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
    void grow() {
        Buf os; os.swap(slots);
        init((mask + 1) * 2);
        for (size_t i = 0; i < os.size(); ++i) if (os[i].key) insert(os[i].key, os[i].val);
    }
    void insert(u64 key, u32 val) {
        u64 h = hash(key) & mask;
        while (slots[h].key) h = (h + 1) & mask;
        slots[h].key = key; slots[h].val = val; ++used;
    }
    // returns child id or 0xFFFFFFFF
    u32 find(u64 key) const {
        u64 h = hash(key) & mask;
        while (slots[h].key) { if (slots[h].key == key) return slots[h].val; h = (h + 1) & mask; }
        return NONE32;
    }
};
}  // namespace

size_t lz78_parse_host(const u8* in, size_t n, std::vector<u32>& ids, std::vector<u8>& chars, bool* leftover_is_high) {
    ids.clear(); chars.clear();
    if (leftover_is_high) *leftover_is_high = false;
    PhraseTable tab;
    size_t cap = 1024;
    while (cap < n / 4 + 16) cap <<= 1;
    tab.init(cap);
    u32 next_id = 1;                                   // root = 0, ids in insertion order from 1 (LZ78Compressor.hpp:78-84)
    u32 node = 0, parent = 0;
    u8 c = 0;
    for (size_t i = 0; i < n; ++i) {                   // :97-121
        c = in[i];
        const u64 key = (((u64)node << 8) | c) + 1;
        const u32 child = tab.find(key);
        if (child == NONE32) {
            if (tab.used * 2 >= tab.mask) tab.grow();
            tab.insert(key, next_id++);
            ids.push_back(node); chars.push_back(c);   // encode(node.id(), Range(factor_count)); encode(c, literal_r)  :101-102
            parent = node = 0;
        } else { parent = node; node = child; }
    }
    if (node != 0) {                                   // :124-131 leftover phrase: (parent.id(), c)
        ids.push_back(parent); chars.push_back(c);
        if (leftover_is_high && c >= 0x80) *leftover_is_high = true;   // the reference passes a signed char here
    }
    return ids.size();
}

// ---- device: gamma coding of the (id, char) pairs --------------------------------------------------------------
constexpr int G_PER_THREAD = 8;
constexpr int G_TILE = 256 * G_PER_THREAD;

__device__ __forceinline__ u32 dev_bits_for(u32 v) { return v ? (32u - (u32)__builtin_clz(v)) : 1u; }
__device__ __forceinline__ u32 gamma_cost(u32 id, u32 ch) { return 2 * dev_bits_for(id) + 2 * dev_bits_for(ch) + 2; }

__global__ __launch_bounds__(256) void gamma_tile_bits_kernel(const u32* __restrict__ ids, const u8* __restrict__ chars, size_t z,
                                                               u64* __restrict__ tile_bits) {
    __shared__ u32 sm[4];
    const size_t i0 = (size_t)blockIdx.x * G_TILE + (size_t)threadIdx.x * G_PER_THREAD;
    u32 sum = 0;
#pragma unroll
    for (int j = 0; j < G_PER_THREAD; ++j) if (i0 + j < z) sum += gamma_cost(ids[i0 + j], chars[i0 + j]);
    sum = wave_reduce_sum(sum);
    if (lane_id() == 0) sm[wave_id()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tile_bits[blockIdx.x] = (u64)sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ void g_put_bits(u64* __restrict__ out, u64 bitpos, u64 val, u32 nbits) {   // see encode.hip put_bits
    const u64 w = bitpos >> 6;
    const u32 off = (u32)(bitpos & 63);
    const u32 avail = 64 - off;
    if (nbits <= avail) {
        const u64 x = (nbits == 64) ? val : (val << (avail - nbits));
        atomicOr((unsigned long long*)&out[w], (unsigned long long)__builtin_bswap64(x));
    } else {
        const u32 rem = nbits - avail;
        atomicOr((unsigned long long*)&out[w], (unsigned long long)__builtin_bswap64(val >> rem));
        atomicOr((unsigned long long*)&out[w + 1], (unsigned long long)__builtin_bswap64(val << (64 - rem)));
    }
}

__global__ __launch_bounds__(256) void gamma_pack_kernel(const u32* __restrict__ ids, const u8* __restrict__ chars, size_t z,
                                                          const u64* __restrict__ tile_off, u64* __restrict__ out) {
    __shared__ u32 sm[5];
    const size_t i0 = (size_t)blockIdx.x * G_TILE + (size_t)threadIdx.x * G_PER_THREAD;
    u32 id[G_PER_THREAD], ch[G_PER_THREAD];
    u32 sum = 0;
#pragma unroll
    for (int j = 0; j < G_PER_THREAD; ++j) {
        const bool v = i0 + j < z;
        id[j] = v ? ids[i0 + j] : 0u;
        ch[j] = v ? chars[i0 + j] : 0u;
        if (v) sum += gamma_cost(id[j], ch[j]);
    }
    u32 total;
    const u32 excl = block_exclusive_sum<u32, 4>(sum, sm, total);
    u64 pos = tile_off[blockIdx.x] + excl;
#pragma unroll
    for (int j = 0; j < G_PER_THREAD; ++j) {
        if (i0 + j < z) {
            // gamma(v): write_unary(bits_for(v)) = b zeros then a one, followed by v in b bits  ==  ((1 << b) | v) in 2b+1 bits
            const u32 b1 = dev_bits_for(id[j]);
            g_put_bits(out, pos + b1, (1ull << b1) | id[j], b1 + 1);      // the b1 leading zeros are already there (zeroed buffer)
            pos += 2 * b1 + 1;
            const u32 b2 = dev_bits_for(ch[j]);
            g_put_bits(out, pos + b2, (1ull << b2) | ch[j], b2 + 1);
            pos += 2 * b2 + 1;
        }
    }
}

__global__ void gamma_terminator_kernel(u8* out, u64 total_bits) {          // io/BitOStream.hpp:53-64
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u64 byte = total_bits >> 3;
    const u32 u = (u32)(total_bits & 7);
    if (u <= 5) out[byte] |= (u8)u;
    else out[byte + 1] = (u8)u;
}

// pairs (device) -> gamma bit stream in d_out (zeroed here); returns the stream length in bytes
size_t lz78_gamma_encode(Ctx& c, const u32* d_ids, const u8* d_chars, size_t z, u8* d_out, size_t out_cap) {
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    const unsigned tiles = z ? cdiv(z, G_TILE) : 0;
    u64 total_bits = 0;
    u64* tile_bits = nullptr;
    if (z) {
        tile_bits = c.arena.get<u64>(tiles + 1);
        u64* d_total = c.arena.get<u64>(1);
        gamma_tile_bits_kernel<<<tiles, 256, 0, s>>>(d_ids, d_chars, z, tile_bits);
        LAUNCH_CHECK();
        exclusive_sum_u64(c, tile_bits, tile_bits, tiles, d_total);
        total_bits = c.read(d_total);
    }
    const size_t out_len = (size_t)(total_bits >> 3) + ((total_bits & 7) <= 5 ? 1 : 2);
    const size_t padded = align_up(out_len + 8, 8);
    if (padded > out_cap) throw HipError{hipErrorOutOfMemory, "lz78: output buffer too small", (int)__LINE__};
    HIP_TRY(hipMemsetAsync(d_out, 0, padded, s));
    if (z) {
        gamma_pack_kernel<<<tiles, 256, 0, s>>>(d_ids, d_chars, z, tile_bits, (u64*)d_out);
        LAUNCH_CHECK();
    }
    gamma_terminator_kernel<<<1, 64, 0, s>>>(d_out, total_bits);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(s));
    c.arena.release(mark);
    return out_len;
}

}  // namespace tdc
