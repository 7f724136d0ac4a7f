// prim.hip -- device-wide scans and the stable LSD radix sort (gfx950, wave64).
//
// Radix sort layout (one pass = 8 key bits):
//   rs_count   : one workgroup per tile (4096 or 8192 keys) builds a 256-bin LDS histogram -> counts[digit][tile]
//   exclusive_sum over counts (digit-major) -> global start of every (digit, tile) run
//   rs_scatter : the same tiling; ranks inside a tile come from a wave-level match (8 ballots / key) plus
//                per-wave LDS counters, which keeps the sort stable without any LDS atomics.
// Keys are read twice and written once per pass; values are read and written once.
#include "prim.hpp"

namespace tdc {

// ------------------------------------------------------------------------------------------------
// scans
// ------------------------------------------------------------------------------------------------
constexpr int SNW = 4;          // waves per workgroup
constexpr int STILE = 4096;     // elements per workgroup: 4 rounds x 256 threads x 4 consecutive elements

template <typename T, int OP>   // OP 0: sum, OP 1: max (identity 0, unsigned)
__device__ __forceinline__ T op2(T a, T b) { return OP == 0 ? (T)(a + b) : (a > b ? a : b); }

template <typename T, int OP>
__device__ __forceinline__ T wave_inclusive_op(T v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= d) v = op2<T, OP>(v, o);
    }
    return v;
}

// exclusive prefix over the 256 threads of the block; total = block aggregate. smem: SNW+1 entries.
template <typename T, int OP>
__device__ __forceinline__ T block_exclusive_op(T v, T* smem, T& total) {
    const int lane = lane_id(), w = wave_id();
    T inc = wave_inclusive_op<T, OP>(v);
    T exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = 0;
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
#pragma unroll
        for (int i = 0; i < SNW; ++i) { T t = smem[i]; smem[i] = run; run = op2<T, OP>(run, t); }
        smem[SNW] = run;
    }
    __syncthreads();
    T res = op2<T, OP>(smem[w], exc);
    total = smem[SNW];
    __syncthreads();
    return res;
}

template <typename T>
__device__ __forceinline__ void load4(const T* __restrict__ in, size_t idx, size_t n, T (&v)[4]) {
    if (idx + 4 <= n) {
        const T* p = (const T*)__builtin_assume_aligned(in + idx, 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = p[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (idx + i < n) ? in[idx + i] : (T)0;
    }
}
template <typename T>
__device__ __forceinline__ void store4(T* __restrict__ out, size_t idx, size_t n, const T (&v)[4]) {
    if (idx + 4 <= n) {
        T* p = (T*)__builtin_assume_aligned(out + idx, 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (idx + i < n) out[idx + i] = v[i];
    }
}

template <typename T, int OP>
__global__ __launch_bounds__(256) void scan_reduce_kernel(const T* __restrict__ in, T* __restrict__ agg, size_t n) {
    __shared__ T sm[SNW];
    const size_t base = (size_t)blockIdx.x * STILE;
    T acc = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t idx = base + (size_t)r * 1024 + (size_t)threadIdx.x * 4;
        T v[4];
        load4(in, idx, n, v);
        acc = op2<T, OP>(acc, op2<T, OP>(op2<T, OP>(v[0], v[1]), op2<T, OP>(v[2], v[3])));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc = op2<T, OP>(acc, __shfl_xor(acc, d, 64));
    if (lane_id() == 0) sm[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T r = 0;
#pragma unroll
        for (int i = 0; i < SNW; ++i) r = op2<T, OP>(r, sm[i]);
        agg[blockIdx.x] = r;
    }
}

template <typename T, int OP, bool INCL>
__global__ __launch_bounds__(256) void scan_apply_kernel(const T* in, T* out, const T* __restrict__ block_prefix,
                                                          size_t n, T* d_total) {
    __shared__ T sm[SNW + 1];
    const size_t base = (size_t)blockIdx.x * STILE;
    T carry = block_prefix ? block_prefix[blockIdx.x] : (T)0;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const size_t idx = base + (size_t)r * 1024 + (size_t)threadIdx.x * 4;
        T v[4];
        load4(in, idx, n, v);
        T s[4];
        s[0] = v[0];
        s[1] = op2<T, OP>(s[0], v[1]);
        s[2] = op2<T, OP>(s[1], v[2]);
        s[3] = op2<T, OP>(s[2], v[3]);
        T total;
        const T excl = block_exclusive_op<T, OP>(s[3], sm, total);
        const T pre = op2<T, OP>(carry, excl);
        T o[4];
        if (INCL) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = op2<T, OP>(pre, s[i]);
        } else {
            o[0] = pre;
#pragma unroll
            for (int i = 1; i < 4; ++i) o[i] = op2<T, OP>(pre, s[i - 1]);
        }
        store4(out, idx, n, o);
        carry = op2<T, OP>(carry, total);
    }
    if (d_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *d_total = carry;
}

template <typename T, int OP, bool INCL>
static void scan_impl(Ctx& c, const T* in, T* out, size_t n, T* d_total) {
    if (n == 0) {
        if (d_total) HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(T), c.stream));
        return;
    }
    const unsigned nb = cdiv(n, STILE);
    if (nb == 1) {
        scan_apply_kernel<T, OP, INCL><<<1, 256, 0, c.stream>>>(in, out, nullptr, n, d_total);
        LAUNCH_CHECK();
        return;
    }
    const size_t mark = c.arena.mark();
    T* agg = c.arena.get<T>(nb);
    Ctx::ProfScope prof(c, K_SCAN, (u64)n * 3 * sizeof(T));    // two reads + one write of every element
    scan_reduce_kernel<T, OP><<<nb, 256, 0, c.stream>>>(in, agg, n);
    LAUNCH_CHECK();
    scan_impl<T, OP, false>(c, agg, agg, nb, nullptr);     // exclusive prefix of the block aggregates
    scan_apply_kernel<T, OP, INCL><<<nb, 256, 0, c.stream>>>(in, out, agg, n, d_total);
    LAUNCH_CHECK();
    c.arena.release(mark);
}

void exclusive_sum_u32(Ctx& c, const u32* in, u32* out, size_t n, u32* d_total) { scan_impl<u32, 0, false>(c, in, out, n, d_total); }
void exclusive_sum_u64(Ctx& c, const u64* in, u64* out, size_t n, u64* d_total) { scan_impl<u64, 0, false>(c, in, out, n, d_total); }
void inclusive_max_u32(Ctx& c, const u32* in, u32* out, size_t n) { scan_impl<u32, 1, true>(c, in, out, n, nullptr); }

// ------------------------------------------------------------------------------------------------
// selection (stream compaction by a class byte)
// ------------------------------------------------------------------------------------------------
constexpr int SEL_PER_THREAD = 8;
constexpr int SEL_TILE = 256 * SEL_PER_THREAD;

__device__ __forceinline__ u32 sel_load_mask(const u8* __restrict__ cls, u8 want, size_t k0, size_t m) {
    u32 mask = 0;
    if (k0 + SEL_PER_THREAD <= m) {
        const u64 w = *(const u64*)__builtin_assume_aligned(cls + k0, 8);
#pragma unroll
        for (int j = 0; j < SEL_PER_THREAD; ++j) mask |= (((u8)(w >> (8 * j)) == want) ? 1u : 0u) << j;
    } else {
        for (int j = 0; j < SEL_PER_THREAD; ++j) if (k0 + j < m && cls[k0 + j] == want) mask |= 1u << j;
    }
    return mask;
}

__global__ __launch_bounds__(256) void sel_count_kernel(const u8* __restrict__ cls, u8 want, size_t m, u32* __restrict__ tilecnt) {
    __shared__ u32 sm[4];
    const size_t k0 = (size_t)blockIdx.x * SEL_TILE + (size_t)threadIdx.x * SEL_PER_THREAD;
    u32 cnt = (k0 < m) ? (u32)__popc(sel_load_mask(cls, want, k0, m)) : 0u;
    cnt = wave_reduce_sum(cnt);
    if (lane_id() == 0) sm[wave_id()] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void sel_scatter_kernel(const u8* __restrict__ cls, u8 want, size_t m,
                                                           const u32* __restrict__ tileoff, const u32* __restrict__ srcA,
                                                           u32* __restrict__ outA, const u64* __restrict__ srcB,
                                                           u64* __restrict__ outB) {
    __shared__ u32 sm[5];
    const size_t k0 = (size_t)blockIdx.x * SEL_TILE + (size_t)threadIdx.x * SEL_PER_THREAD;
    const u32 mask = (k0 < m) ? sel_load_mask(cls, want, k0, m) : 0u;
    u32 total;
    u32 o = tileoff[blockIdx.x] + block_exclusive_sum<u32, 4>((u32)__popc(mask), sm, total);
    if (!mask) return;
#pragma unroll
    for (int j = 0; j < SEL_PER_THREAD; ++j) {
        if (mask & (1u << j)) {
            outA[o] = srcA ? srcA[k0 + j] : (u32)(k0 + j);      // srcA == nullptr: select the indices themselves
            if (srcB) outB[o] = srcB[k0 + j];
            ++o;
        }
    }
}

void select_by_class(Ctx& c, const u8* cls, u8 want, size_t m, const u32* srcA, u32* outA, const u64* srcB, u64* outB,
                     u32* d_count) {
    if (m == 0) { HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(u32), c.stream)); return; }
    const size_t mark = c.arena.mark();
    const unsigned tiles = cdiv(m, SEL_TILE);
    u32* tilecnt = c.arena.get<u32>(tiles);
    sel_count_kernel<<<tiles, 256, 0, c.stream>>>(cls, want, m, tilecnt);
    LAUNCH_CHECK();
    exclusive_sum_u32(c, tilecnt, tilecnt, tiles, d_count);
    sel_scatter_kernel<<<tiles, 256, 0, c.stream>>>(cls, want, m, tilecnt, srcA, outA, srcB, outB);
    LAUNCH_CHECK();
    c.arena.release(mark);
}

// ------------------------------------------------------------------------------------------------
// orbit marking (greedy parse chains)
// ------------------------------------------------------------------------------------------------
constexpr int ORB_TILE = 1024, ORB_SUPER = 1 << 20;

__global__ __launch_bounds__(256) void chain_exit1_kernel(const u32* __restrict__ next, size_t n, u32* __restrict__ exit1) {
    __shared__ u32 e[ORB_TILE];
    const size_t base = (size_t)blockIdx.x * ORB_TILE;
    const size_t tile_end = (base + ORB_TILE < n) ? base + ORB_TILE : n;
    for (int k = threadIdx.x; k < ORB_TILE; k += 256) e[k] = (base + k < n) ? next[base + k] : (u32)n;
    __syncthreads();
    // invariant: e[k] lies on k's chain and every chain position strictly between k and e[k] is inside the tile
    for (int round = 0; round < 10; ++round) {
        u32 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const u32 x = e[threadIdx.x + 256 * j]; v[j] = (x < tile_end) ? e[x - base] : x; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) e[threadIdx.x + 256 * j] = v[j];
        __syncthreads();
    }
    for (int k = threadIdx.x; k < ORB_TILE; k += 256) if (base + k < n) exit1[base + k] = e[k];
}

// one workgroup per super-tile, tiles from right to left; exit2 of a later tile is read back through L2 (agent-scope loads)
__global__ __launch_bounds__(256) void chain_exit2_kernel(const u32* __restrict__ exit1, size_t n, u32* exit2) {
    const size_t sbase = (size_t)blockIdx.x * ORB_SUPER;
    const size_t super_end = (sbase + ORB_SUPER < n) ? sbase + ORB_SUPER : n;
    const size_t ntile = (super_end - sbase + ORB_TILE - 1) / ORB_TILE;
    for (size_t tt = ntile; tt-- > 0;) {
        const size_t base = sbase + tt * ORB_TILE;
        for (int k = threadIdx.x; k < ORB_TILE; k += 256) {
            const size_t i = base + k;
            if (i < super_end) {
                const u32 x = exit1[i];
                const u32 y = (x >= super_end) ? x : __hip_atomic_load(&exit2[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&exit2[i], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();          // includes the wait for this tile's stores
    }
}

__global__ void chain_super_walk_kernel(const u32* __restrict__ exit2, size_t n, u32* __restrict__ super_entry) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    size_t e = 0, guard = 0;
    const size_t nsuper = (n + ORB_SUPER - 1) / ORB_SUPER;
    while (e < n && guard++ <= nsuper) { super_entry[e >> 20] = (u32)e; e = exit2[e]; }
}
__global__ void chain_tile_entries_kernel(const u32* __restrict__ exit1, size_t n, const u32* __restrict__ super_entry, size_t nsuper,
                                          u32* __restrict__ tile_entry) {
    const size_t sidx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nsuper) return;
    size_t e = super_entry[sidx];
    if (e == NONE32) return;
    const size_t super_end = (sidx + 1) * (size_t)ORB_SUPER < n ? (sidx + 1) * (size_t)ORB_SUPER : n;
    for (int guard = 0; e < super_end && guard <= ORB_SUPER / ORB_TILE; ++guard) { tile_entry[e >> 10] = (u32)e; e = exit1[e]; }
}
__global__ void chain_mark_kernel(const u32* __restrict__ next, size_t n, const u32* __restrict__ tile_entry, size_t ntiles,
                                  u8* __restrict__ mark) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    size_t e = tile_entry[t];
    if (e == NONE32) return;
    const size_t tile_end = (t + 1) * (size_t)ORB_TILE < n ? (t + 1) * (size_t)ORB_TILE : n;
    for (int guard = 0; e < tile_end && guard <= ORB_TILE; ++guard) { mark[e] = 1; e = next[e]; }
}


void mark_orbit_u32(Ctx& c, const u32* next, size_t n, u8* mark, u32* exit1, u32* exit2) {
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark0 = c.arena.mark();
    const size_t ntiles = (n + ORB_TILE - 1) / ORB_TILE, nsuper = (n + ORB_SUPER - 1) / ORB_SUPER;
    chain_exit1_kernel<<<(unsigned)ntiles, 256, 0, s>>>(next, n, exit1);
    LAUNCH_CHECK();
    chain_exit2_kernel<<<(unsigned)nsuper, 256, 0, s>>>(exit1, n, exit2);
    LAUNCH_CHECK();
    u32* super_entry = c.arena.get<u32>(nsuper);
    u32* tile_entry = c.arena.get<u32>(ntiles);
    HIP_TRY(hipMemsetAsync(super_entry, 0xFF, nsuper * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(tile_entry, 0xFF, ntiles * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(mark, 0, n, s));
    chain_super_walk_kernel<<<1, 64, 0, s>>>(exit2, n, super_entry);
    LAUNCH_CHECK();
    chain_tile_entries_kernel<<<cdiv(nsuper, 64), 64, 0, s>>>(exit1, n, super_entry, nsuper, tile_entry);
    LAUNCH_CHECK();
    chain_mark_kernel<<<cdiv(ntiles, 256), 256, 0, s>>>(next, n, tile_entry, ntiles, mark);
    LAUNCH_CHECK();
    c.arena.release(mark0);
}

// ------------------------------------------------------------------------------------------------
// fills
// ------------------------------------------------------------------------------------------------
__global__ void fill_u32_kernel(u32* p, size_t n, u32 v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
void fill_u32(Ctx& c, u32* p, size_t n, u32 v) {
    if (!n) return;
    unsigned g = cdiv(n, 256); if (g > 8192) g = 8192;
    fill_u32_kernel<<<g, 256, 0, c.stream>>>(p, n, v);
    LAUNCH_CHECK();
}
void fill_u8(Ctx& c, u8* p, size_t n, u8 v) {
    if (!n) return;
    HIP_TRY(hipMemsetAsync(p, v, n, c.stream));
}

// ------------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 16;               // keys per lane; wave w of a workgroup owns keys [w*1024, (w+1)*1024) of the tile

template <typename K, int NW>
__global__ __launch_bounds__(NW * 64) void rs_count_kernel(const K* __restrict__ keys, u32* __restrict__ counts, size_t n,
                                                            u32 numTiles, int shift, u32 dmask, u32 per_xcd) {
    __shared__ u32 hist[256];
    const u32 tile = xcd_tile(blockIdx.x, per_xcd);
    if (tile >= numTiles) return;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t tileBase = (size_t)tile * (NW * 64 * RS_ITEMS) + (size_t)wave_id() * (64 * RS_ITEMS) + lane_id();
#pragma unroll 4
    for (int j = 0; j < RS_ITEMS; ++j) {
        const size_t idx = tileBase + (size_t)j * 64;
        const bool valid = idx < n;
        const u32 d = valid ? (u32)((keys[idx] >> shift) & dmask) : 0u;
        const u32 d0 = __builtin_amdgcn_readfirstlane(d);
        if (__all(valid && d == d0)) {                 // whole wave in one bin: one LDS atomic instead of 64 colliding ones
            if (lane_id() == 0) atomicAdd(&hist[d0], 64u);
        } else if (valid) {
            atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) counts[(size_t)tile * 256 + threadIdx.x] = hist[threadIdx.x];      // tile-major: one coalesced KiB per tile
}

// counts[tile][digit] (tile-major) -> global start of every (digit, tile) run: the exclusive scan in digit-major order,
// evaluated column-wise so that every access is a coalesced row of 256 counters.
constexpr int CS_ROWS = 128;       // tiles per workgroup
__global__ __launch_bounds__(256) void rs_colsum_kernel(const u32* __restrict__ counts, u32 numTiles, u32* __restrict__ blocksum) {
    const u32 t0 = blockIdx.x * CS_ROWS;
    const u32 t1 = (t0 + CS_ROWS < numTiles) ? t0 + CS_ROWS : numTiles;
    u32 acc = 0;
#pragma unroll 8
    for (u32 t = t0; t < t1; ++t) acc += counts[(size_t)t * 256 + threadIdx.x];
    blocksum[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
// one workgroup of 1024 threads: per digit the exclusive prefix over the row blocks; dstart[d] = start of digit d.
// Thread (g, d) owns a quarter of the rows of column d (the walk down a column is a chain of dependent round trips: four
// shorter chains instead of one).
__global__ __launch_bounds__(1024) void rs_colbase_kernel(u32* __restrict__ blocksum, u32 numBlocks, u32* __restrict__ dstart) {
    __shared__ u32 part[4][256];
    __shared__ u32 cs[256];
    const u32 d = threadIdx.x & 255u, g = threadIdx.x >> 8;
    const u32 per = (numBlocks + 3) / 4;
    const u32 b0 = g * per < numBlocks ? g * per : numBlocks;
    const u32 b1 = b0 + per < numBlocks ? b0 + per : numBlocks;
    u32 acc = 0;
    u32 b = b0;
    for (; b + 8 <= b1; b += 8) {                           // eight independent loads in flight
        u32 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = blocksum[(size_t)(b + i) * 256 + d];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    for (; b < b1; ++b) acc += blocksum[(size_t)b * 256 + d];
    part[g][d] = acc;
    __syncthreads();
    u32 run = 0;
    for (u32 k = 0; k < g; ++k) run += part[k][d];
    if (g == 0) cs[d] = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    __syncthreads();
    if (threadIdx.x < 64) {                                 // start of every digit: exclusive scan over the 256 column sums
        const u32 l = threadIdx.x;
        const u32 c0 = cs[4 * l], c1 = cs[4 * l + 1], c2 = cs[4 * l + 2], c3 = cs[4 * l + 3];
        const u32 inc = wave_inclusive_sum(c0 + c1 + c2 + c3);
        const u32 ex = inc - (c0 + c1 + c2 + c3);
        dstart[4 * l] = ex; dstart[4 * l + 1] = ex + c0; dstart[4 * l + 2] = ex + c0 + c1; dstart[4 * l + 3] = ex + c0 + c1 + c2;
    }
    b = b0;
    for (; b + 8 <= b1; b += 8) {
        u32 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = blocksum[(size_t)(b + i) * 256 + d];
#pragma unroll
        for (int i = 0; i < 8; ++i) { blocksum[(size_t)(b + i) * 256 + d] = run; run += v[i]; }
    }
    for (; b < b1; ++b) { const u32 v = blocksum[(size_t)b * 256 + d]; blocksum[(size_t)b * 256 + d] = run; run += v; }
}
__global__ __launch_bounds__(256) void rs_colapply_kernel(u32* __restrict__ counts, u32 numTiles, const u32* __restrict__ blocksum,
                                                           const u32* __restrict__ dstart) {
    const u32 t0 = blockIdx.x * CS_ROWS;
    const u32 t1 = (t0 + CS_ROWS < numTiles) ? t0 + CS_ROWS : numTiles;
    u32 run = blocksum[(size_t)blockIdx.x * 256 + threadIdx.x] + dstart[threadIdx.x];
#pragma unroll 8
    for (u32 t = t0; t < t1; ++t) { const u32 v = counts[(size_t)t * 256 + threadIdx.x]; counts[(size_t)t * 256 + threadIdx.x] = run; run += v; }
}
static void radix_offsets(Ctx& c, u32* counts, u32 numTiles, u32* blocksum) {
    const u32 nb = cdiv(numTiles, CS_ROWS);
    Ctx::ProfScope prof(c, K_SCAN, (u64)numTiles * 256 * 3 * sizeof(u32));
    rs_colsum_kernel<<<nb, 256, 0, c.stream>>>(counts, numTiles, blocksum);
    LAUNCH_CHECK();
    rs_colbase_kernel<<<1, 1024, 0, c.stream>>>(blocksum, nb, blocksum + (size_t)nb * 256);
    LAUNCH_CHECK();
    rs_colapply_kernel<<<nb, 256, 0, c.stream>>>(counts, numTiles, blocksum, blocksum + (size_t)nb * 256);
    LAUNCH_CHECK();
}

template <typename K, int NW>
__global__ __launch_bounds__(NW * 64) void rs_scatter_kernel(const K* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                                              K* __restrict__ keys_out, u32* __restrict__ vals_out,
                                                              const u32* __restrict__ offsets, size_t n, u32 numTiles,
                                                              int shift, u32 dmask, u32 per_xcd) {
    __shared__ u32 wcnt[NW][256];     // per-wave running digit counts
    __shared__ u32 wbase[NW][256];    // global start of (wave, digit) run
    const int lane = lane_id(), w = wave_id();
    const u32 tile = xcd_tile(blockIdx.x, per_xcd);
    if (tile >= numTiles) return;
    for (int i = threadIdx.x; i < NW * 256; i += NW * 64) (&wcnt[0][0])[i] = 0;
    __syncthreads();

    K k[RS_ITEMS];
    u32 v[RS_ITEMS];
    u32 loc[RS_ITEMS];
    u32* mycnt = wcnt[w];
    const size_t tileBase = (size_t)tile * (NW * 64 * RS_ITEMS) + (size_t)w * (64 * RS_ITEMS) + lane;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const size_t idx = tileBase + (size_t)j * 64;
        const bool valid = idx < n;
        k[j] = valid ? keys_in[idx] : (K)0;
        v[j] = valid ? vals_in[idx] : 0u;
        const u32 d = (u32)((k[j] >> shift) & dmask);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const u32 prefix = lds_load(&mycnt[d]);
        const u32 rank = (u32)__popcll(peers & lt_mask);
        loc[j] = prefix + rank;
        if (valid && rank == 0) lds_store(&mycnt[d], prefix + (u32)__popcll(peers));   // leader = lowest valid lane of the group
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const u32 t = threadIdx.x;
        u32 run = offsets[(size_t)tile * 256 + t];
#pragma unroll
        for (int i = 0; i < NW; ++i) { wbase[i][t] = run; run += wcnt[i][t]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const size_t idx = tileBase + (size_t)j * 64;
        if (idx < n) {
            const u32 d = (u32)((k[j] >> shift) & dmask);
            const u32 dst = wbase[w][d] + loc[j];
            keys_out[dst] = k[j];
            vals_out[dst] = v[j];
        }
    }
}

// Variant that reorders the tile in LDS before writing: the direct scatter above issues one 8-byte write request per lane
// (64 different cache lines per store instruction); here consecutive lanes write consecutive elements of a digit's run, so a
// store instruction touches a handful of lines.  Same tiling, same stable ranks; the staging buffer holds the keys first and
// is reused for the values.
template <typename K, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void rs_scatter_lds_kernel(const K* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                                                  K* __restrict__ keys_out, u32* __restrict__ vals_out,
                                                                  const u32* __restrict__ offsets, size_t n, u32 numTiles,
                                                                  int shift, u32 dmask, u32 per_xcd) {
    constexpr int TILE = NW * 64 * RS_ITEMS;
    __shared__ u32 wcnt[NW][256];     // per-wave digit counts, then start of the (wave, digit) run inside the sorted tile
    __shared__ u32 gbase[256];        // global start of the digit's run minus its start inside the sorted tile
    __shared__ __align__(8) K stage[TILE];               // >= NW * 2 KB for both key widths
    __shared__ u32 scan_sm[NW + 1];
    const int lane = lane_id(), w = wave_id();
    const u32 tile = xcd_tile(blockIdx.x, per_xcd);
    if (tile >= numTiles) return;
    for (int i = threadIdx.x; i < NW * 256; i += NW * 64) { (&wcnt[0][0])[i] = 0; ((unsigned long long*)stage)[i] = 0; }
    __syncthreads();
    // the wave's lane-mask table of the LDS match: in the staging buffer, which is not needed before all ranks are known
    unsigned long long* M = (unsigned long long*)stage + w * 256;
    const u64 lanebit = 1ull << lane;

    K k[RS_ITEMS];
    u32 v[RS_ITEMS];
    u32 loc[RS_ITEMS];
    u32* mycnt = wcnt[w];
    const size_t tileBase = (size_t)tile * TILE + (size_t)w * (64 * RS_ITEMS) + lane;
    const u32 tileCount = (u32)(((size_t)(tile + 1) * TILE <= n) ? (size_t)TILE : n - (size_t)tile * TILE);
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const size_t idx = tileBase + (size_t)j * 64;
        const bool valid = idx < n;
        k[j] = valid ? keys_in[idx] : (K)0;
        v[j] = valid ? vals_in[idx] : 0u;
        const u32 d = (u32)((k[j] >> shift) & dmask);
        const u64 peers = wave_match_lds(M, d, valid, lanebit);
        const u32 prefix = lds_load(&mycnt[d]);
        const u32 rank = (u32)__popcll(peers & lt_mask);
        loc[j] = prefix + rank;
        if (valid && rank == 0) lds_store(&mycnt[d], prefix + (u32)__popcll(peers));
    }
    __syncthreads();
    {   // thread t = digit t: runs inside the sorted tile (exclusive scan over the digits), per-wave starts, global base
        const u32 t = threadIdx.x;
        u32 tot = 0;
        if (NW * 64 == 256 || t < 256) {
#pragma unroll
            for (int i = 0; i < NW; ++i) tot += wcnt[i][t];
        }
        u32 total;
        const u32 start = block_exclusive_sum<u32, NW>((NW * 64 == 256 || t < 256) ? tot : 0u, scan_sm, total);
        if (NW * 64 == 256 || t < 256) {
            u32 run = start;
#pragma unroll
            for (int i = 0; i < NW; ++i) { const u32 c = wcnt[i][t]; wcnt[i][t] = run; run += c; }
            gbase[t] = offsets[(size_t)tile * 256 + t] - start;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const size_t idx = tileBase + (size_t)j * 64;
        const u32 d = (u32)((k[j] >> shift) & dmask);
        loc[j] += wcnt[w][d];                                   // position inside the sorted tile
        if (idx < n) stage[loc[j]] = k[j];
    }
    __syncthreads();
    u32 dst[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 sp = (u32)r * (NW * 64) + threadIdx.x;
        dst[r] = 0xFFFFFFFFu;
        if (sp < tileCount) {
            const K key = stage[sp];
            dst[r] = gbase[(u32)((key >> shift) & dmask)] + sp;
            keys_out[dst[r]] = key;
        }
    }
    __syncthreads();
    u32* stage32 = (u32*)stage;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const size_t idx = tileBase + (size_t)j * 64;
        if (idx < n) stage32[loc[j]] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 sp = (u32)r * (NW * 64) + threadIdx.x;
        if (dst[r] != 0xFFFFFFFFu) vals_out[dst[r]] = stage32[sp];
    }
}

template <typename K, int NW>
static int radix_sort_pairs_nw(Ctx& c, K* keys[2], u32* vals[2], size_t n, int begin_bit, int end_bit) {
    const size_t mark = c.arena.mark();
    constexpr int TILE = NW * 64 * RS_ITEMS;
    const u32 numTiles = cdiv(n, TILE);
    u32* counts = c.arena.get<u32>((size_t)256 * numTiles);
    u32* blocksum = c.arena.get<u32>((size_t)256 * (cdiv(numTiles, CS_ROWS) + 1));     // + one row: start of every digit
    const u32 per_xcd = (c.xcd_remap == 1 && numTiles >= 64) ? cdiv(numTiles, 8) : 0u;
    const u32 grid = per_xcd ? 8 * per_xcd : numTiles;
    int cur = 0;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        const int bits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const u32 dmask = (1u << bits) - 1u;
        // algorithmic bytes: count reads every key once; scatter reads and writes every (key, value) pair once
        const int pc = c.prof_begin(K_RS_COUNT, (u64)n * sizeof(K));
        rs_count_kernel<K, NW><<<grid, NW * 64, 0, c.stream>>>(keys[cur], counts, n, numTiles, shift, dmask, per_xcd);
        LAUNCH_CHECK();
        c.prof_end(pc);
        radix_offsets(c, counts, numTiles, blocksum);
        const int ps = c.prof_begin(sizeof(K) == 8 ? K_RS_SCATTER_U64 : K_RS_SCATTER_U32, (u64)n * 2 * (sizeof(K) + sizeof(u32)));
        if (NW == 4 && (c.radix_lds == 1 || (c.radix_lds == 2 && sizeof(K) == 4)))
            rs_scatter_lds_kernel<K, NW><<<grid, NW * 64, 0, c.stream>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], counts, n,
                                                                      numTiles, shift, dmask, per_xcd);
        else
            rs_scatter_kernel<K, NW><<<grid, NW * 64, 0, c.stream>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], counts, n,
                                                                      numTiles, shift, dmask, per_xcd);
        LAUNCH_CHECK();
        c.prof_end(ps);
        cur ^= 1;
    }
    c.arena.release(mark);
    return cur;
}

template <typename K>
static int radix_sort_pairs(Ctx& c, K* keys[2], u32* vals[2], size_t n, int begin_bit, int end_bit) {
    if (n == 0 || end_bit <= begin_bit) return 0;
    // large inputs: 8 waves per workgroup (8192-key tiles: longer per-digit runs, half the count array);
    // small inputs: 4 waves (more workgroups to fill the 256 CUs)
    if (c.radix_waves == 8 && n >= ((size_t)1 << 22)) return radix_sort_pairs_nw<K, 8>(c, keys, vals, n, begin_bit, end_bit);
    return radix_sort_pairs_nw<K, 4>(c, keys, vals, n, begin_bit, end_bit);
}

int radix_sort_pairs_u64(Ctx& c, u64* keys[2], u32* vals[2], size_t n, int b, int e) { return radix_sort_pairs<u64>(c, keys, vals, n, b, e); }

// ---- pass 0 of the suffix array's initial sort, fed from the text ----------------------------------------------------
// key(i) = the k recoded bytes text[i .. i+k) as a k-digit number in base sigma (zeros behind the text), value(i) = i.
// The two kernels below are rs_count_kernel / rs_scatter_lds_kernel<u64, 4> with the loads replaced by that computation
// (the tile's recoded bytes are staged in LDS), so the (key, index) arrays are first written by the scatter of pass 0
// instead of being written, read by the count and read again by the scatter.
constexpr int GEN_HALO = 32;          // k <= 32
// recoded bytes of one tile (+ halo) into LDS; every thread fetches 16 consecutive bytes with one request (the loads of a
// workgroup are all in flight together instead of one round trip per loop iteration)
template <int NW>
__device__ __forceinline__ void gen_stage_tile(const TextKeyGen& g, size_t t0, const u8* __restrict__ code, u8* __restrict__ sy) {
    constexpr int TILE = NW * 64 * RS_ITEMS;
    static_assert(TILE == NW * 64 * 16, "one 16-byte piece per thread");
    const size_t p = t0 + (size_t)threadIdx.x * 16;
    u8 b[16];
    if (p + 16 <= g.n && (((size_t)g.text) & 15) == 0) {
        const uint4 v = *(const uint4*)(g.text + p);
        const u32 wv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (u8)(wv[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (p + i < g.n) ? g.text[p + i] : (u8)0;
    }
    u32 o[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i >> 2] |= (u32)((p + i < g.n) ? code[b[i]] : (u8)0) << (8 * (i & 3));
    *(uint4*)(sy + (size_t)threadIdx.x * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    if (threadIdx.x < GEN_HALO) {
        const size_t q = t0 + TILE + threadIdx.x;
        sy[TILE + threadIdx.x] = (q < g.n) ? code[g.text[q]] : (u8)0;
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void rs_gen_count_kernel(TextKeyGen g, u32* __restrict__ counts, u32 numTiles, u32 dmask, u32 per_xcd) {
    constexpr int TILE = NW * 64 * RS_ITEMS;
    __shared__ u32 hist[256];
    __shared__ u8 code[256];
    __shared__ __align__(16) u8 sy[TILE + GEN_HALO];
    const u32 tile = xcd_tile(blockIdx.x, per_xcd);
    if (tile >= numTiles) return;
    for (int i = threadIdx.x; i < 256; i += NW * 64) { hist[i] = 0; code[i] = g.code[i]; }
    __syncthreads();
    const size_t t0 = (size_t)tile * TILE;
    gen_stage_tile<NW>(g, t0, code, sy);
    __syncthreads();
    // a lane owns RS_ITEMS consecutive positions (pass 0 need not be stable, so the order inside the tile is free): the key of
    // the next position is the previous one minus its leading symbol, times sigma, plus the symbol that enters
    const int lb = wave_id() * (64 * RS_ITEMS) + lane_id() * RS_ITEMS;
    u32 acc = 0;                                          // the low 8 bits of the key survive 32-bit wrap-around
    for (int t = 0; t < g.k; ++t) acc = acc * g.sigma + sy[lb + t];
    const u32 top = (u32)g.top;
#pragma unroll 4
    for (int j = 0; j < RS_ITEMS; ++j) {
        const bool valid = t0 + lb + j < g.n;
        const u32 d = valid ? (acc & dmask) : 0u;
        acc = (acc - (u32)sy[lb + j] * top) * g.sigma + sy[lb + j + g.k];
        const u32 d0 = __builtin_amdgcn_readfirstlane(d);
        if (__all(valid && d == d0)) {
            if (lane_id() == 0) atomicAdd(&hist[d0], 64u);
        } else if (valid) {
            atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) counts[(size_t)tile * 256 + threadIdx.x] = hist[threadIdx.x];
}

template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void rs_gen_scatter_kernel(TextKeyGen g, u64* __restrict__ keys_out, u32* __restrict__ vals_out,
                                                                  const u32* __restrict__ offsets, u32 numTiles, u32 dmask, u32 per_xcd) {
    typedef u64 K;
    constexpr int TILE = NW * 64 * RS_ITEMS;
    __shared__ u32 wcnt[NW][256];
    __shared__ u32 gbase[256];
    __shared__ __align__(16) K stage[TILE];
    __shared__ u32 scan_sm[NW + 1];
    __shared__ u8 code[256];
    const int lane = lane_id(), w = wave_id();
    const u32 tile = xcd_tile(blockIdx.x, per_xcd);
    if (tile >= numTiles) return;
    const size_t n = g.n;
    for (int i = threadIdx.x; i < NW * 256; i += NW * 64) { (&wcnt[0][0])[i] = 0; ((unsigned long long*)stage)[1024 + i] = 0; }
    for (int i = threadIdx.x; i < 256; i += NW * 64) code[i] = g.code[i];
    __syncthreads();
    u8* sy = (u8*)stage;                                  // recoded bytes of the tile; dead before `stage` is written
    unsigned long long* M = (unsigned long long*)stage + 1024 + w * 256;     // lane-mask tables of the LDS match, behind the bytes
    const u64 lanebit = 1ull << lane;
    const size_t t0 = (size_t)tile * TILE;
    gen_stage_tile<NW>(g, t0, code, sy);
    __syncthreads();

    K k[RS_ITEMS];
    u32 loc[RS_ITEMS];
    u32* mycnt = wcnt[w];
    const int lb = w * (64 * RS_ITEMS) + lane * RS_ITEMS;  // a lane owns RS_ITEMS consecutive positions (see rs_gen_count_kernel)
    const size_t tileBase = t0 + lb;
    const u32 tileCount = (u32)(((size_t)(tile + 1) * TILE <= n) ? (size_t)TILE : n - t0);
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u64 key = 0;
    for (int j0 = 0; j0 < g.k; j0 += g.chunk) {
        const int len = (g.k - j0 < g.chunk) ? g.k - j0 : g.chunk;
        u32 acc = 0, scale = 1;
        for (int t = 0; t < len; ++t) { acc = acc * g.sigma + sy[lb + j0 + t]; scale *= g.sigma; }
        key = (j0 == 0) ? (u64)acc : key * scale + acc;
    }
    u8 lead[RS_ITEMS], enter[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) { lead[j] = sy[lb + j]; enter[j] = sy[lb + j + g.k]; }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const bool valid = tileBase + (size_t)j < n;
        k[j] = valid ? key : (K)0;
        const u32 d = (u32)(k[j] & dmask);
        const u64 peers = wave_match_lds(M, d, valid, lanebit);
        const u32 prefix = lds_load(&mycnt[d]);
        const u32 rank = (u32)__popcll(peers & lt_mask);
        loc[j] = prefix + rank;
        if (valid && rank == 0) lds_store(&mycnt[d], prefix + (u32)__popcll(peers));
        key = (key - (u64)lead[j] * g.top) * g.sigma + enter[j];
    }
    __syncthreads();
    {
        const u32 t = threadIdx.x;
        u32 tot = 0;
        if (NW * 64 == 256 || t < 256) {
#pragma unroll
            for (int i = 0; i < NW; ++i) tot += wcnt[i][t];
        }
        u32 total;
        const u32 start = block_exclusive_sum<u32, NW>((NW * 64 == 256 || t < 256) ? tot : 0u, scan_sm, total);
        if (NW * 64 == 256 || t < 256) {
            u32 run = start;
#pragma unroll
            for (int i = 0; i < NW; ++i) { const u32 c = wcnt[i][t]; wcnt[i][t] = run; run += c; }
            gbase[t] = offsets[(size_t)tile * 256 + t] - start;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const u32 d = (u32)(k[j] & dmask);
        loc[j] += wcnt[w][d];
        if (tileBase + (size_t)j < n) stage[loc[j]] = k[j];
    }
    __syncthreads();
    u32 dst[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 sp = (u32)r * (NW * 64) + threadIdx.x;
        dst[r] = 0xFFFFFFFFu;
        if (sp < tileCount) {
            const K key = stage[sp];
            dst[r] = gbase[(u32)(key & dmask)] + sp;
            keys_out[dst[r]] = key;
        }
    }
    __syncthreads();
    u32* stage32 = (u32*)stage;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const size_t idx = tileBase + (size_t)j;
        if (idx < n) stage32[loc[j]] = (u32)idx;           // value = text position
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 sp = (u32)r * (NW * 64) + threadIdx.x;
        if (dst[r] != 0xFFFFFFFFu) vals_out[dst[r]] = stage32[sp];
    }
}

int radix_sort_text_keys_u64(Ctx& c, const TextKeyGen& g, u64* keys[2], u32* vals[2], int end_bit) {
    const size_t n = g.n;
    if (n == 0) return 0;
    constexpr int NW = 4;
    constexpr int TILE = NW * 64 * RS_ITEMS;
    const size_t mark = c.arena.mark();
    const u32 numTiles = cdiv(n, TILE);
    u32* counts = c.arena.get<u32>((size_t)256 * numTiles);
    u32* blocksum = c.arena.get<u32>((size_t)256 * (cdiv(numTiles, CS_ROWS) + 1));
    const u32 per_xcd = (c.xcd_remap == 1 && numTiles >= 64) ? cdiv(numTiles, 8) : 0u;
    const u32 grid = per_xcd ? 8 * per_xcd : numTiles;
    const int bits = end_bit < 8 ? end_bit : 8;
    const u32 dmask = (1u << bits) - 1u;
    rs_gen_count_kernel<NW><<<grid, NW * 64, 0, c.stream>>>(g, counts, numTiles, dmask, per_xcd);
    LAUNCH_CHECK();
    radix_offsets(c, counts, numTiles, blocksum);
    rs_gen_scatter_kernel<NW><<<grid, NW * 64, 0, c.stream>>>(g, keys[0], vals[0], counts, numTiles, dmask, per_xcd);
    LAUNCH_CHECK();
    c.arena.release(mark);
    if (end_bit <= 8) return 0;
    return radix_sort_pairs_u64(c, keys, vals, n, 8, end_bit);
}

// ---- one-workgroup bitonic sort for tiny inputs -----------------------------------------------------------------
constexpr int SMALL_SORT_MAX = 2048;

__global__ __launch_bounds__(256) void small_sort_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                                          u64* __restrict__ keys_out, u32* __restrict__ vals_out, u32 n, u32 np2,
                                                          u64 keymask) {
    __shared__ u64 xk[SMALL_SORT_MAX];
    __shared__ u32 xv[SMALL_SORT_MAX];
    u64 k[8];
    u32 v[8];
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const u32 i = threadIdx.x * 8 + r;
        k[r] = (i < n) ? (keys_in[i] & keymask) : ~0ull;      // padding sorts to the end
        v[r] = (i < n) ? i : 0xFFFFFFFFu;                      // carry the input index: the full pair is re-read at the end
    }
    block_bitonic_sort_2048(k, v, xk, xv, np2);
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const u32 i = threadIdx.x * 8 + r;
        if (i < n) { keys_out[i] = keys_in[v[r]]; vals_out[i] = vals_in[v[r]]; }
    }
}

// ---- whole LSD radix sort of up to 8192 pairs in ONE workgroup and one launch ------------------------------------------
// The multi-launch sort costs five launches per 8-bit pass whatever the size; the push lists of mid-size factorization levels
// (a few thousand pairs, 45 key bits) are sorted thousands of times on texts with long repeats.  Same stable ranking as
// rs_scatter_lds_kernel; the pairs stay in registers, the tile is re-ordered through LDS after every pass.
constexpr int MID_NW = 8;
constexpr int MID_SORT_MAX = MID_NW * 64 * RS_ITEMS;      // 8192
__global__ __launch_bounds__(MID_NW * 64) void one_workgroup_radix_sort_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                                                                u64* __restrict__ keys_out, u32* __restrict__ vals_out, u32 n,
                                                                                int begin_bit, int end_bit) {
    constexpr int NW = MID_NW;
    __shared__ u32 wcnt[NW][256];
    __shared__ u64 stage[MID_SORT_MAX];
    __shared__ u32 scan_sm[NW + 1];
    const int lane = lane_id(), w = wave_id();
    u64 k[RS_ITEMS];
    u32 v[RS_ITEMS];
    u32 loc[RS_ITEMS];
    const u32 base = (u32)w * (64 * RS_ITEMS) + (u32)lane;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const u32 idx = base + (u32)j * 64;
        k[j] = (idx < n) ? keys_in[idx] : ~0ull;             // padding: largest digit in every pass, stays behind the real pairs (stable)
        v[j] = (idx < n) ? vals_in[idx] : 0u;
    }
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u32* mycnt = wcnt[w];
    u32* stage32 = (u32*)stage;
    unsigned long long* M = (unsigned long long*)stage + w * 256;     // lane-mask table of the LDS match (the staging buffer is idle while ranking)
    const u64 lanebit = 1ull << lane;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        const int bits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const u32 dmask = (1u << bits) - 1u;
        for (int i = threadIdx.x; i < NW * 256; i += NW * 64) { (&wcnt[0][0])[i] = 0; stage[i] = 0; }   // counters + lane-mask tables
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const u32 d = (u32)((k[j] >> shift) & dmask);
            const u64 peers = wave_match_lds(M, d, true, lanebit);
            const u32 prefix = lds_load(&mycnt[d]);
            const u32 rank = (u32)__popcll(peers & lt_mask);
            loc[j] = prefix + rank;
            if (rank == 0) lds_store(&mycnt[d], prefix + (u32)__popcll(peers));
        }
        __syncthreads();
        {
            const u32 t = threadIdx.x;
            u32 tot = 0;
            if (t < 256) {
#pragma unroll
                for (int i = 0; i < NW; ++i) tot += wcnt[i][t];
            }
            u32 total;
            const u32 start = block_exclusive_sum<u32, NW>(t < 256 ? tot : 0u, scan_sm, total);
            if (t < 256) {
                u32 run = start;
#pragma unroll
                for (int i = 0; i < NW; ++i) { const u32 c = wcnt[i][t]; wcnt[i][t] = run; run += c; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            loc[j] += wcnt[w][(u32)((k[j] >> shift) & dmask)];
            stage[loc[j]] = k[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) k[j] = stage[base + (u32)j * 64];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) stage32[loc[j]] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) v[j] = stage32[base + (u32)j * 64];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const u32 idx = base + (u32)j * 64;
        if (idx < n) { keys_out[idx] = k[j]; vals_out[idx] = v[j]; }
    }
}

int sort_pairs_u64_distinct(Ctx& c, u64* keys[2], u32* vals[2], size_t n, int b, int e) {
    if (n == 0 || e <= b) return 0;
    if (n > (size_t)SMALL_SORT_MAX && n <= (size_t)MID_SORT_MAX) {
        one_workgroup_radix_sort_kernel<<<1, MID_NW * 64, 0, c.stream>>>(keys[0], vals[0], keys[1], vals[1], (u32)n, b, e);
        LAUNCH_CHECK();
        return 1;
    }
    if (n > (size_t)SMALL_SORT_MAX || b != 0) return radix_sort_pairs<u64>(c, keys, vals, n, b, e);
    const u64 keymask = (e >= 64) ? ~0ull : ((1ull << e) - 1);
    u32 np2 = 8;
    while (np2 < n) np2 <<= 1;
    small_sort_kernel<<<1, 256, 0, c.stream>>>(keys[0], vals[0], keys[1], vals[1], (u32)n, np2, keymask);
    LAUNCH_CHECK();
    return 1;
}
int radix_sort_pairs_u32(Ctx& c, u32* keys[2], u32* vals[2], size_t n, int b, int e) { return radix_sort_pairs<u32>(c, keys, vals, n, b, e); }

// ---- bucketed scatter ------------------------------------------------------------------------------------------
// dst[idx[j]] = val[j] for m pairs with pairwise distinct idx < n_dst.  A direct scatter of 4-byte elements over a
// gigabyte touches one DRAM line per element; one stable radix partition by the top 8 bits of idx turns it into a
// streaming pass plus a scatter whose writes stay inside one n_dst/256 window per run (L2 / Infinity-Cache sized).
constexpr int WS_ITEMS = 4;
__global__ __launch_bounds__(256) void window_scatter_kernel(const u32* __restrict__ idx, const u32* __restrict__ val, size_t m,
                                                              u32* __restrict__ dst, u32 numTiles, u32 per_xcd) {
    const u32 tile = xcd_tile(blockIdx.x, per_xcd);
    if (tile >= numTiles) return;
    const size_t j0 = ((size_t)tile * 256 + threadIdx.x) * WS_ITEMS;
    u32 k[WS_ITEMS], v[WS_ITEMS];
    if ((((size_t)idx | (size_t)val) & 15) == 0) {               // the partitioned pairs; callers may pass offset views otherwise
        load4(idx, j0, m, k);
        load4(val, j0, m, v);
    } else {
#pragma unroll
        for (int r = 0; r < WS_ITEMS; ++r) { k[r] = (j0 + r < m) ? idx[j0 + r] : 0u; v[r] = (j0 + r < m) ? val[j0 + r] : 0u; }
    }
#pragma unroll
    for (int r = 0; r < WS_ITEMS; ++r) if (j0 + r < m) dst[k[r]] = v[r];
}

// Final pass when idx is a permutation of [0, m) (m = n_dst, or n_dst - 1 with only the last index missing): after the
// partition by the top 16 bits window w holds exactly the pairs [w * W, (w + 1) * W), W = 2^(bits - 16).  One workgroup per
// window: the values are scattered into an LDS image of the window, which then leaves as whole lines.
constexpr u32 WIMG_MAX = 8192;
__global__ __launch_bounds__(256) void window_image_kernel(const u32* __restrict__ idx, const u32* __restrict__ val, size_t m,
                                                            u32* __restrict__ dst, u32 W) {
    __shared__ u32 img[WIMG_MAX];
    const size_t base = (size_t)blockIdx.x * W;
    const size_t end = (base + W < m) ? base + W : m;
    for (size_t j = base + threadIdx.x; j < end; j += 256) img[idx[j] & (W - 1)] = val[j];
    __syncthreads();
    for (size_t q = base + threadIdx.x; q < end; q += 256) dst[q] = img[q - base];
}

void bucketed_scatter_u32(Ctx& c, const u32* idx, const u32* val, size_t m, u32* dst, size_t n_dst, u32* tmp_idx, u32* tmp_val,
                          u32* tmp_idx2, u32* tmp_val2, bool permutation) {
    if (m == 0) return;
    const int bits = (int)bits_for(n_dst ? n_dst - 1 : 0);
    const u32* k = idx;
    const u32* v = val;
    int part_bits = 16;                                          // index bits the partition has resolved
    if (c.msd_partition && bits > 16 && m >= ((size_t)1 << 20) && tmp_idx2 && tmp_val2) {
        // partition by the top 16 (18 for more than 2^29 destinations: the windows of the final pass then still fit its LDS image)
        // bits of idx: two MSD levels, no stability needed (ssort.hip)
        const int db = (bits - 16 > 13) ? 9 : 8;
        msd_partition_pairs_u32(c, idx, val, m, bits, db, tmp_idx2, tmp_val2, tmp_idx, tmp_val);
        k = tmp_idx2; v = tmp_val2;
        part_bits = 2 * db;
    } else if (bits > 8 && m >= ((size_t)1 << 20)) {
        // two stable 8-bit passes (low digit first) = partition by the top 16 bits: the writes of the final pass stay inside
        // windows of n_dst / 65536 elements, which L2 merges into whole lines
        const int top = bits > 16 ? bits - 16 : 0;
        const int mid = (bits - top > 8) ? top + 8 : bits;
        u32* ka[2] = { const_cast<u32*>(idx), tmp_idx };        // one pass each: [0] is only read
        u32* va[2] = { const_cast<u32*>(val), tmp_val };
        radix_sort_pairs<u32>(c, ka, va, m, top, mid);
        k = tmp_idx; v = tmp_val;
        if (mid < bits && tmp_idx2 && tmp_val2) {
            u32* kb[2] = { tmp_idx, tmp_idx2 };
            u32* vb[2] = { tmp_val, tmp_val2 };
            radix_sort_pairs<u32>(c, kb, vb, m, mid, bits);
            k = tmp_idx2; v = tmp_val2;
        }
    }
    if (permutation && k == tmp_idx2 && bits > part_bits && (1u << (bits - part_bits)) <= WIMG_MAX) {
        const u32 W = 1u << (bits - part_bits);
        Ctx::ProfScope prof(c, K_WINDOW_SCATTER, (u64)m * 12);      // read the partitioned pairs, write every destination word once
        window_image_kernel<<<cdiv(m, W), 256, 0, c.stream>>>(k, v, m, dst, W);
        LAUNCH_CHECK();
        return;
    }
    // every XCD walks one contiguous part of the partitioned pairs: all writes to a destination line then meet in ONE L2
    // (measured without this: WRITE_SIZE = 9x the destination bytes, every 4-byte store left its L2 as a partial line)
    const u32 numTiles = cdiv(m, 256 * WS_ITEMS);
    const u32 per_xcd = (c.xcd_remap != 2 && numTiles >= 64) ? cdiv(numTiles, 8) : 0u;
    Ctx::ProfScope prof(c, K_WINDOW_SCATTER, (u64)m * 12);
    window_scatter_kernel<<<per_xcd ? 8 * per_xcd : numTiles, 256, 0, c.stream>>>(k, v, m, dst, numTiles, per_xcd);
    LAUNCH_CHECK();
}

}  // namespace tdc
