// fused.hip -- ISA, Phi and PLCP from a final suffix array in ONE bucketed scatter (gfx950).
//   isa[sa[i]] = i                        ds/ISAFromSA.hpp:30-43
//   phi[sa[i]] = sa[i-1]                  ds/PhiFromSA.hpp:35-45     (phi[sa[0]] = sa[n-1])
//   plcp[sa[i]] = lcp(sa[i-1], sa[i])     ds/PLCPFromPhi.hpp:27-53   (the values, here handed over by the sort: suffix_array.hip)
// All three scatter along the same permutation i -> sa[i].  Round 2 ran two bucketed scatters of 8-byte pairs (first ranks, Phi) and
// recomputed PLCP from the text with one scattered read per position; here ONE record (sa[i], i, sa[i-1], lcp) travels
// through one two-level partition by destination window (prim.hip bucketed_scatter_u32 explains why a partition beats a direct
// scatter) and every window is written as whole lines from an LDS image, array by array.  The record is 12 bytes: behind the first
// level the top digit of the destination is implied by the bucket, and the LCP byte takes its place in the index word.
#include "stages.hpp"
#include "prim.hpp"

#include <type_traits>

namespace tdc {

constexpr int FS_TILE = 4096;        // = SS_TILE of ssort.hip (the row-block tables assume it)
constexpr int FS_ITEMS = 16;
constexpr u32 FS_WMAX = 8192;

struct FSLevel {
    const u32* idx_in; const void* rp_in; const u32* prev_in; const u8* lcp_in;   // rp: rank | predecessor << 32 (level 1: rank = index, prev_in = sa);
    u32* idx_out; void* rp_out; u8* lcp_out;                                      //     without Phi (WP = false) the rank alone, 4 bytes
    u32* counts; const u32* blk_seg; const u32* blk_start; const u32* seg_start;
    u32 nseg, R, per_xcd; int shift;
    int lsh;                                                  // FIRST: the LCP byte goes to the bits [lsh, lsh + 8) of the index word (the top digit leaves)
};
__device__ __forceinline__ bool fs_row(const FSLevel& P, u32 row, size_t& base, u32& cnt) {
    const u32 blk = row / P.R;
    if (blk >= P.blk_start[P.nseg]) return false;
    const u32 s = P.blk_seg[blk];
    const u64 t = (u64)(blk - P.blk_start[s]) * P.R + row % P.R;
    const u32 s0 = P.seg_start[s], s1 = P.seg_start[s + 1];
    const u64 off = t * FS_TILE;
    base = s0; cnt = 0;
    if (off >= (u64)(s1 - s0)) return true;
    base = (size_t)s0 + off;
    const u64 left = (u64)(s1 - s0) - off;
    cnt = left < FS_TILE ? (u32)left : (u32)FS_TILE;
    return true;
}

template <int DB>
__global__ __launch_bounds__(256) void fs_count_kernel(FSLevel P, u32 rows) {
    constexpr u32 D = 1u << DB;
    __shared__ u32 hist[D];
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd);
    if (row >= rows) return;
    size_t base; u32 cnt;
    if (!fs_row(P, row, base, cnt)) return;
    for (u32 i = threadIdx.x; i < D; i += 256) hist[i] = 0;
    if (cnt == 0) { for (u32 i = threadIdx.x; i < D; i += 256) P.counts[(size_t)row * D + i] = 0; return; }
    __syncthreads();
    const u32 lb = wave_id() * (64 * FS_ITEMS) + lane_id();
    const u32* ip = P.idx_in + base + lb;
    u32 kk[FS_ITEMS];
#pragma unroll
    for (int j = 0; j < FS_ITEMS; ++j) kk[j] = (lb + (u32)j * 64 < cnt) ? ip[j * 64] : 0u;
#pragma unroll
    for (int j = 0; j < FS_ITEMS; ++j) {
        const bool valid = lb + (u32)j * 64 < cnt;
        const u32 d = (kk[j] >> P.shift) & (D - 1);
        const u32 d0 = __builtin_amdgcn_readfirstlane(d);
        if (__all(valid && d == d0)) { if (lane_id() == 0) atomicAdd(&hist[d0], 64u); }
        else if (valid) atomicAdd(&hist[d], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < D; i += 256) P.counts[(size_t)row * D + i] = hist[i];
}

// FIRST: record j of the input is (sa[j + 1], j + 1, sa[j], lcp8[j + 1]) -- idx_in = sa + 1, prev_in = sa, lcp_in = lcp8 + 1, the rank
// is the index.  The tile is written stream by stream, each staged in LDS in bucket order first (whole runs per bucket leave as
// consecutive words).
#ifndef TDC_FS_WPE
#define TDC_FS_WPE 4
#endif
// PAIR = 2: a workgroup of 512 threads takes two consecutive rows of a row block as ONE tile of 8 192 records -- rows of a block lie in
// one segment, their inputs are adjacent and so are their runs of every digit in the output (the offsets are a running prefix inside
// the block), and the order inside a bucket does not matter here.  With 512 buckets a 4 096-record tile leaves 8 records per bucket:
// 32-byte runs of index words, i.e. partial lines (1.5 x the algorithmic write traffic in the PMC counters); twice the tile, twice the run.
template <int DB, bool FIRST, int PAIR, bool WP>
__global__ __launch_bounds__(256 * PAIR) __attribute__((amdgpu_waves_per_eu(TDC_FS_WPE, TDC_FS_WPE))) void fs_scatter_kernel(FSLevel P, u32 rows) {
    constexpr int NT = 256 * PAIR;
    constexpr u32 TILE = (u32)FS_TILE * PAIR;
    constexpr u32 D = 1u << DB;
    __shared__ u32 tcnt[D];
    __shared__ u32 gbase[D];
    __shared__ __align__(16) u64 stage[TILE];
    __shared__ u32 scan_sm[NT / 64 + 1];
    const int lane = lane_id();
    const u32 row = xcd_tile(blockIdx.x, P.per_xcd) * PAIR;
    if (row >= rows) return;
    size_t base; u32 cnt;
    if (!fs_row(P, row, base, cnt) || cnt == 0) return;
    if (PAIR == 2 && row + 1 < rows) {                          // (the second row of the pair: same block, its records follow directly)
        size_t base2; u32 cnt2;
        if (fs_row(P, row + 1, base2, cnt2) && cnt2) cnt += cnt2;
    }
    for (u32 i = threadIdx.x; i < D; i += NT) tcnt[i] = 0;
    __syncthreads();
    const u32 lb = wave_id() * (64 * FS_ITEMS) + lane;
    const u32* ip = P.idx_in + base + lb;
    u32 k[FS_ITEMS], pos[FS_ITEMS];
#pragma unroll
    for (int j = 0; j < FS_ITEMS; ++j) k[j] = (lb + (u32)j * 64 < cnt) ? ip[j * 64] : 0u;
    // every stream of the thread's records is requested up front: the staging phases below then never wait for global memory
    using RT = typename std::conditional<WP, u64, u32>::type;   // the second stream: rank (| predecessor)
    RT rp[FS_ITEMS];
    u32 lc[FS_ITEMS];
    const u8* lst = (const u8*)stage;                           // FIRST: the tile's LCP bytes, staged with 16-byte loads (sixteen byte loads per
    size_t lmis = 0;                                            // thread fetched 64 bytes per wave instruction)
    if (FIRST) {
        const u8* src = P.lcp_in + base;
        lmis = ((size_t)src) & 15;
        const uint4* a0 = (const uint4*)(src - lmis);           // (the array starts 16-byte aligned and is padded: no access outside it)
        const u32 nv = (cnt + (u32)lmis + 15u) / 16u;
        for (u32 i = threadIdx.x; i < nv; i += NT) ((uint4*)stage)[i] = a0[i];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < FS_ITEMS; ++j) {
        const u32 e = lb + (u32)j * 64;
        const bool valid = e < cnt;
        if (FIRST) { if constexpr (WP) rp[j] = (u64)(u32)(base + e + 1) | ((u64)(valid ? P.prev_in[base + e] : 0u) << 32); else rp[j] = (u32)(base + e + 1); }
        else rp[j] = valid ? ((const RT*)P.rp_in)[base + e] : (RT)0;
        lc[j] = (FIRST && valid) ? (u32)lst[lmis + e] : 0u;
    }
#pragma unroll
    for (int j = 0; j < FS_ITEMS; ++j) {
        const bool valid = lb + (u32)j * 64 < cnt;
        const u32 d = (k[j] >> P.shift) & (D - 1);
        const u32 d0 = __builtin_amdgcn_readfirstlane(d);
        u32 rank = 0;
        if (__all(valid && d == d0)) {
            if (lane == 0) rank = atomicAdd(&tcnt[d0], 64u);
            rank = __builtin_amdgcn_readfirstlane(rank) + (u32)lane;
        } else if (valid) rank = atomicAdd(&tcnt[d], 1u);
        pos[j] = rank;
    }
    __syncthreads();
    {
        const u32 t = threadIdx.x;
        constexpr u32 DPT = (D + NT - 1) / NT;                  // digits per thread
        u32 tot[DPT], sum = 0;
#pragma unroll
        for (u32 q = 0; q < DPT; ++q) { const u32 d = t * DPT + q; tot[q] = d < D ? tcnt[d] : 0u; sum += tot[q]; }
        u32 total;
        u32 start = block_exclusive_sum<u32, NT / 64>(sum, scan_sm, total);
#pragma unroll
        for (u32 q = 0; q < DPT; ++q) {
            const u32 d = t * DPT + q;
            if (d < D) {
                tcnt[d] = start;
                gbase[d] = P.counts[(size_t)row * D + d] - start;
            }
            start += tot[q];
        }
    }
    __syncthreads();
    u32 dst[FS_ITEMS];
    u32* stage32 = (u32*)stage;
    // stream 1: the destination index (FIRST: low bits + the LCP byte); its bucket gives every slot of the sorted tile its global address
    unsigned short* stage_d = (unsigned short*)(stage + TILE / 2);          // digits of the staged words: second half of the buffer
#pragma unroll
    for (int j = 0; j < FS_ITEMS; ++j) {
        if (lb + (u32)j * 64 < cnt) {
            const u32 d = (k[j] >> P.shift) & (D - 1);
            pos[j] += tcnt[d];
            stage32[pos[j]] = FIRST ? ((k[j] & ((1u << P.lsh) - 1u)) | (lc[j] << P.lsh)) : k[j];
            stage_d[pos[j]] = (unsigned short)d;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FS_ITEMS; ++r) {
        const u32 sp = (u32)r * NT + threadIdx.x;
        dst[r] = 0xFFFFFFFFu;
        if (sp < cnt) {
            dst[r] = gbase[stage_d[sp]] + sp;
            P.idx_out[dst[r]] = stage32[sp];
        }
    }
    __syncthreads();
    // stream 2: rank and predecessor as one 8-byte word (the rank alone without Phi)
    RT* stage2 = (RT*)stage;
#pragma unroll
    for (int j = 0; j < FS_ITEMS; ++j) if (lb + (u32)j * 64 < cnt) stage2[pos[j]] = rp[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FS_ITEMS; ++r) if (dst[r] != 0xFFFFFFFFu) ((RT*)P.rp_out)[dst[r]] = stage2[(u32)r * NT + threadIdx.x];
}

// After the partition by the top 2 * DB index bits, window w holds exactly the records of the positions [w * W, (w + 1) * W) (position
// n - 1 = sa[0] has no record).  One workgroup per window: each array is scattered into an LDS image of the window and leaves as whole
// lines.
constexpr int FS_IMG_T = 1024;       // threads of the image kernel: a window's records are loaded ONCE, eight per thread
template <bool WP>
__global__ __launch_bounds__(FS_IMG_T) void fs_image_kernel(const u32* __restrict__ idx, const void* __restrict__ rp_v, int lsh, size_t m, u32 W,
                                                             u32* __restrict__ isa, u32* __restrict__ phi, u32* __restrict__ plcp, u32* __restrict__ d_max) {
    __shared__ u32 img[FS_WMAX];
    __shared__ u32 smx[FS_IMG_T / 64];
    constexpr int R = FS_WMAX / FS_IMG_T;
    const size_t base = (size_t)blockIdx.x * W;
    const size_t end = (base + W < m) ? base + W : m;
    using RT = typename std::conditional<WP, u64, u32>::type;
    const RT* rp = (const RT*)rp_v;
    u32 ix[R];
    RT rr[R];
    u32 mx = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {                               // (round 2 of this kernel read idx three times and rp twice: 28 bytes per record)
        const size_t j = base + (size_t)r * FS_IMG_T + threadIdx.x;
        const bool have = j < end;
        ix[r] = have ? idx[j] : 0xFFFFFFFFu;
        rr[r] = have ? rp[j] : (RT)0;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (ix[r] != 0xFFFFFFFFu) img[ix[r] & (W - 1)] = (u32)rr[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) { const size_t q = base + (size_t)r * FS_IMG_T + threadIdx.x; if (q < end) isa[q] = img[q - base]; }
    __syncthreads();
    if constexpr (WP) {
#pragma unroll
        for (int r = 0; r < R; ++r) if (ix[r] != 0xFFFFFFFFu) img[ix[r] & (W - 1)] = (u32)(rr[r] >> 32);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) { const size_t q = base + (size_t)r * FS_IMG_T + threadIdx.x; if (q < end) phi[q] = img[q - base]; }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (ix[r] != 0xFFFFFFFFu) { const u32 l = (ix[r] >> lsh) & 0xFFu; img[ix[r] & (W - 1)] = l; mx = max(mx, l); }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) { const size_t q = base + (size_t)r * FS_IMG_T + threadIdx.x; if (q < end) plcp[q] = img[q - base]; }
    mx = wave_reduce_max(mx);
    if (lane_id() == 0) smx[wave_id()] = mx;
    __syncthreads();
    if (wave_id() == 0) {                                       // (the whole wave takes part in the reduction)
        u32 v = (lane_id() < FS_IMG_T / 64) ? smx[lane_id()] : 0u;
        v = wave_reduce_max(v);
        if (lane_id() == 0 && v) atomicMax(d_max, v);
    }
}

__global__ void fs_first_kernel(const u32* __restrict__ sa, size_t n, u32* __restrict__ isa, u32* __restrict__ phi, u32* __restrict__ plcp) {
    const u32 p = sa[0];
    isa[p] = 0; if (phi) phi[p] = sa[n - 1]; plcp[p] = 0;
}
__global__ void fs_direct_kernel(const u32* __restrict__ sa, const u8* __restrict__ lcp8, size_t n, u32* __restrict__ isa, u32* __restrict__ phi,
                                 u32* __restrict__ plcp, u32* __restrict__ d_max) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 l = 0;
    if (i < n) {
        const u32 p = sa[i];
        isa[p] = (u32)i;
        if (phi) phi[p] = (i == 0) ? sa[n - 1] : sa[i - 1];
        l = (i == 0) ? 0u : (u32)lcp8[i];
        plcp[p] = l;
    }
    l = wave_reduce_max(l);
    if (lane_id() == 0 && l) atomicMax(d_max, l);
}

void build_isa_phi_plcp_fused(Ctx& c, const u32* sa, const u8* lcp8, size_t n, u32* isa, u32* phi, u32* plcp, u32* d_maxlcp) {
    hipStream_t s = c.stream;
    HIP_TRY(hipMemsetAsync(d_maxlcp, 0, sizeof(u32), s));
    if (!n) return;
    const int bits = (int)bits_for(n - 1);
    if (n < ((size_t)1 << 20) || !c.bucket_scatter) {
        Ctx::ProfScope prof(c, K_PHI, (u64)n * 17);
        fs_direct_kernel<<<cdiv(n, 256), 256, 0, s>>>(sa, lcp8, n, isa, phi, plcp, d_maxlcp);
        LAUNCH_CHECK();
        return;
    }
    const size_t mark = c.arena.mark();
    const size_t m = n - 1;                                   // records of the slots 1 .. n-1; slot 0 (position n-1) has none
    const int db = (bits - 16 > 13) ? 9 : 8;
    const u32 D = 1u << db;
    u32* idx[2] = { c.arena.get<u32>(m), c.arena.get<u32>(m) };
    const bool wp = phi != nullptr;                           // phi == nullptr: ISA and PLCP only, 8-byte records (the factorizer then takes a factor's source from SA[ISA[p] - 1])
    void* rp[2] = { c.arena.alloc(m * (wp ? 8 : 4)), c.arena.alloc(m * (wp ? 8 : 4)) };
    const u32* seg_start = ss_first_segment(c, m);
    u32 nseg = 1;
    for (int l = 0; l < 2; ++l) {
        u32* nstart = c.arena.get<u32>((size_t)nseg * D + 1);
        const size_t lm = c.arena.mark();
        SegTables Tb;
        ss_level_tables(c, seg_start, nseg, m, D, Tb);
        FSLevel P;
        if (l == 0) { P.idx_in = sa + 1; P.rp_in = nullptr; P.prev_in = sa; P.lcp_in = lcp8 + 1; }
        else { P.idx_in = idx[0]; P.rp_in = rp[0]; P.prev_in = nullptr; P.lcp_in = nullptr; }
        P.idx_out = idx[l]; P.rp_out = rp[l]; P.lcp_out = nullptr;
        P.lsh = bits - db;
        P.counts = Tb.counts; P.blk_seg = Tb.blk_seg; P.blk_start = Tb.blk_start; P.seg_start = seg_start;
        P.nseg = nseg; P.R = Tb.R; P.shift = bits - db * (l + 1);
        const u32 rows = Tb.rows;
        P.per_xcd = (c.xcd_remap == 1 && rows >= 64) ? cdiv(rows, 8) : 0u;
        const u32 grid = P.per_xcd ? 8 * P.per_xcd : rows;
        {
            const int pc = c.prof_begin(K_RS_COUNT, (u64)m * 4);
            if (db == 9) fs_count_kernel<9><<<grid, 256, 0, s>>>(P, rows); else fs_count_kernel<8><<<grid, 256, 0, s>>>(P, rows);
            LAUNCH_CHECK();
            c.prof_end(pc);
        }
        ss_level_offsets(c, Tb, seg_start, nseg, D, nstart, m);
        {
            const int ps = c.prof_begin(K_RS_SCATTER_U32, (u64)m * (wp ? (l == 0 ? 5 + 12 : 24) : (l == 0 ? 5 + 8 : 16)));
            const bool pair = (Tb.R % 2) == 0 && c.fs_pair;             // (rows per block is a power of two: a pair never straddles a block)
            FSLevel P2 = P;
            const u32 groups = pair ? (rows + 1) / 2 : rows;
            P2.per_xcd = (c.xcd_remap == 1 && groups >= 64) ? cdiv(groups, 8) : 0u;
            const u32 grid2 = P2.per_xcd ? 8 * P2.per_xcd : groups;
            auto launch = [&](auto wpc) {
                constexpr bool WPC = decltype(wpc)::value;
                if (pair) {
                    if (l == 0) { if (db == 9) fs_scatter_kernel<9, true, 2, WPC><<<grid2, 512, 0, s>>>(P2, rows); else fs_scatter_kernel<8, true, 2, WPC><<<grid2, 512, 0, s>>>(P2, rows); }
                    else { if (db == 9) fs_scatter_kernel<9, false, 2, WPC><<<grid2, 512, 0, s>>>(P2, rows); else fs_scatter_kernel<8, false, 2, WPC><<<grid2, 512, 0, s>>>(P2, rows); }
                } else {
                    if (l == 0) { if (db == 9) fs_scatter_kernel<9, true, 1, WPC><<<grid, 256, 0, s>>>(P, rows); else fs_scatter_kernel<8, true, 1, WPC><<<grid, 256, 0, s>>>(P, rows); }
                    else { if (db == 9) fs_scatter_kernel<9, false, 1, WPC><<<grid, 256, 0, s>>>(P, rows); else fs_scatter_kernel<8, false, 1, WPC><<<grid, 256, 0, s>>>(P, rows); }
                }
            };
            if (wp) launch(std::true_type{}); else launch(std::false_type{});
            LAUNCH_CHECK();
            c.prof_end(ps);
        }
        c.arena.release(lm);
        seg_start = nstart;
        nseg *= D;
    }
    {
        const u32 W = 1u << (bits - 2 * db);
        if (W > FS_WMAX) throw HipError{hipErrorUnknown, "fused scatter: window larger than the LDS image", (int)__LINE__};
        Ctx::ProfScope prof(c, K_FS_IMAGE, (u64)m * (wp ? 24 : 16));          // 12 bytes in, 12 out (8 + 8 without Phi)
        if (wp) fs_image_kernel<true><<<cdiv(m, W), FS_IMG_T, 0, s>>>(idx[1], rp[1], bits - db, m, W, isa, phi, plcp, d_maxlcp);
        else fs_image_kernel<false><<<cdiv(m, W), FS_IMG_T, 0, s>>>(idx[1], rp[1], bits - db, m, W, isa, phi, plcp, d_maxlcp);
        LAUNCH_CHECK();
        fs_first_kernel<<<1, 1, 0, s>>>(sa, n, isa, phi, plcp);
        LAUNCH_CHECK();
    }
    c.arena.release(mark);
}

}  // namespace tdc
