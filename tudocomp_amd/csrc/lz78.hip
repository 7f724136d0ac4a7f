// lz78.hip -- LZ78Compressor (compressors/LZ78Compressor.hpp:64-140) with EliasGammaCoder
// (coders/EliasGammaCoder.hpp:26-29, io/BitOStream.hpp:105-129); BASELINE.json configs[3].
//
// The LZ78 parse is inherently sequential (every step depends on the whole dictionary so far), so it stays on the
// host (option a; lz78_host.cpp).
// The coder side is data-parallel and runs on the GPU with the same cost / scan / pack scheme as encode.hip:
// gamma(v) = bits_for(v) zeros, "1", v in bits_for(v) bits; pair i contributes gamma(id_i) gamma(c_i).
#include "stages.hpp"
#include "prim.hpp"

#include <vector>
#include <stdlib.h>
#include <string.h>

namespace tdc {

// (the LZ78 parse itself: lz78_host.cpp)

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
