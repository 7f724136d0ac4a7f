// flatten.hip -- factor list <-> position space, and lzss::FactorBuffer::flatten
// (compressors/lzss/LZSSFactors.hpp:79-132).
//
// flatten redirects the source of factor f through the factor s covering that source while the whole copy fits
// inside s (:106-119).  The reference does it sequentially in position order, so f sees the ALREADY FLATTENED source
// of an earlier factor (s.pos < f.pos) and the ORIGINAL source of a later one.  Device formulation: rounds; a factor
// advances along its chain as long as every earlier factor it touches is final, otherwise it waits for the next
// round.  A final source is published as ONE 32-bit word (NOT_DONE until then), so no flag/data ordering is needed.
#include "stages.hpp"
#include "prim.hpp"

#include <stdlib.h>

namespace tdc {

__global__ void fstart_flag_kernel(const u32* __restrict__ flen, const u32* __restrict__ owner, size_t n, u32* __restrict__ flag) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 o = owner[p];                                     // a factor starts where the covering factor changes
    flag[p] = (flen[p] != 0 && o != NONE32 && (p == 0 || owner[p - 1] != o)) ? 1u : 0u;
}
__global__ void fstart_scatter_kernel(const u32* __restrict__ flen, const u32* __restrict__ owner, const u32* __restrict__ fsrc,
                                      const u32* __restrict__ offs, size_t n, size_t cap, u32* __restrict__ pos,
                                      u32* __restrict__ src, u32* __restrict__ len) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u32 l = flen[p];
    const u32 ow = owner[p];
    if (l != 0 && ow != NONE32 && (p == 0 || owner[p - 1] != ow)) {
        const u32 o = offs[p];
        if (o < cap) {
            pos[o] = (u32)p;
            if (src) src[o] = fsrc[p];
            if (len) len[o] = l;
        }
    }
}

size_t extract_factors(Ctx& c, size_t n, FactorSpace fs, u32* pos, u32* src, u32* len, size_t cap) {
    if (n == 0) return 0;
    const size_t mark = c.arena.mark();
    u32* offs = c.arena.get<u32>(n);
    u32* d_total = c.arena.get<u32>(1);
    const unsigned gn = cdiv(n, 256);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * 12);
        fstart_flag_kernel<<<gn, 256, 0, c.stream>>>(fs.flen, fs.owner, n, offs);
        LAUNCH_CHECK();
    }
    exclusive_sum_u32(c, offs, offs, n, d_total);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * 12);
        fstart_scatter_kernel<<<gn, 256, 0, c.stream>>>(fs.flen, fs.owner, fs.fsrc, offs, n, cap, pos, src, len);
        LAUNCH_CHECK();
    }
    const size_t z = c.read(d_total);
    c.arena.release(mark);
    return z;
}

// ---- owner[] from the factor starts: owner[q] = index (in position order) of the factor covering q ------------------
__global__ void owner_flag_kernel(const u32* __restrict__ flen, size_t n, u32* __restrict__ flag, u32* __restrict__ owner) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flag[p] = flen[p] != 0 ? 1u : 0u;
    owner[p] = NONE32;
}
__global__ void owner_starts_kernel(const u32* __restrict__ flen, const u32* __restrict__ offs, size_t n, u32* __restrict__ pos) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && flen[p] != 0) pos[offs[p]] = (u32)p;
}
template <int G>      // G lanes per factor
__global__ void owner_fill_kernel(const u32* __restrict__ pos, size_t z, const u32* __restrict__ flen, size_t n, u32* __restrict__ owner) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const u32 sub = threadIdx.x % G;
    if (i >= z) return;
    const u32 p = pos[i], l = flen[p];
    for (u32 j = sub; j < l && (size_t)p + j < n; j += G) owner[p + j] = (u32)i;
}

void build_owner(Ctx& c, size_t n, FactorSpace& fs) {
    fs.have_list = false;
    if (n == 0) return;
    const size_t mark = c.arena.mark();
    u32* offs = c.arena.get<u32>(n);
    u32* pos = fs.fpos ? fs.fpos : c.arena.get<u32>(n);
    u32* d_total = c.arena.get<u32>(1);
    const unsigned gn = cdiv(n, 256);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * 12);
        owner_flag_kernel<<<gn, 256, 0, c.stream>>>(fs.flen, n, offs, fs.owner);
        LAUNCH_CHECK();
    }
    exclusive_sum_u32(c, offs, offs, n, d_total);
    {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * 8);
        owner_starts_kernel<<<gn, 256, 0, c.stream>>>(fs.flen, offs, n, pos);
        LAUNCH_CHECK();
    }
    const size_t z = c.read(d_total);
    if (fs.fpos) { fs.nfact = z; fs.have_list = true; }
    if (z) {
        Ctx::ProfScope prof(c, K_EXTRACT, (u64)n * 4 + (u64)z * 8);
        // factors are disjoint and in position order: consecutive lanes fill consecutive ranges
        if (z * 64 > n) owner_fill_kernel<8><<<cdiv(z * 8, 256), 256, 0, c.stream>>>(pos, z, fs.flen, n, fs.owner);
        else            owner_fill_kernel<64><<<cdiv(z * 64, 256), 256, 0, c.stream>>>(pos, z, fs.flen, n, fs.owner);
        LAUNCH_CHECK();
    }
    c.arena.release(mark);
}

__global__ void fspace_clear_kernel(size_t n, u32* __restrict__ flen, u32* __restrict__ owner) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flen[p] = 0;
    owner[p] = NONE32;
}
__global__ void fspace_scatter_kernel(const u32* __restrict__ pos, const u32* __restrict__ src, const u32* __restrict__ len,
                                      size_t z, size_t n, u32* __restrict__ flen, u32* __restrict__ owner, u32* __restrict__ fsrc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= z) return;
    const u32 p = pos[i], l = len[i];
    if ((size_t)p + l > n) return;        // validated on the host; never write out of bounds
    flen[p] = l;
    fsrc[p] = src[i];
    for (u32 j = 0; j < l; ++j) owner[p + j] = (u32)i;
}

void scatter_factors(Ctx& c, size_t n, const u32* pos, const u32* src, const u32* len, size_t z, FactorSpace fs) {
    if (n == 0) return;
    fspace_clear_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(n, fs.flen, fs.owner);
    LAUNCH_CHECK();
    if (z) {
        fspace_scatter_kernel<<<cdiv(z, 256), 256, 0, c.stream>>>(pos, src, len, z, n, fs.flen, fs.owner, fs.fsrc);
        LAUNCH_CHECK();
    }
}

// ------------------------------------------------------------------------------------------------------------
constexpr u32 NOT_DONE = 0xFFFFFFFFu;

struct FlattenScalars { u32 waiting; u32 num_flattened; u32 max_depth; u32 pad; };

// Everything a chain step needs about a factor lives in ONE 16-byte record, indexed by the factor's rank r in position order:
//   rec[r] = { pos, len, original source, final source (NOT_DONE until known) }
// (one scattered line per step besides owner[src]; separate arrays for (pos, len), the original and the final source cost a line
// each.  Finding the covering factor through a table of the first factor per 64-position block plus a scan of consecutive
// records -- no owner[] line at all -- was built and measured slower: 6.6 vs 5.7 ms, the scan is a chain of dependent loads.)
__global__ void flatten_init_kernel(const u32* __restrict__ fpos, size_t z, const u32* __restrict__ flen, const u32* __restrict__ orig,
                                    uint4* __restrict__ rec, u32* __restrict__ cursrc, u32* __restrict__ depth) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= z) return;
    const u32 p = fpos[i];
    const u32 o = orig[p];
    rec[i] = make_uint4(p, flen[p], o, NOT_DONE);
    cursrc[i] = o;
    depth[i] = 0;
}

// One round over the still-waiting factors (work[] holds their ranks; wcls[j] = 1 if still waiting).
__global__ __launch_bounds__(256) void flatten_round_kernel(const u32* __restrict__ work, u32 nwork, size_t n, const u32* __restrict__ owner,
                                                             uint4* rec, u32* __restrict__ cursrc, u32* __restrict__ depth, u8* __restrict__ wcls,
                                                             FlattenScalars* __restrict__ sc, u32 max_steps) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    bool counted = false;
    u32 dep = 0;
    if (j < nwork) {
        const u32 i = work ? work[j] : j;
        const uint4 me = rec[i];
        const u32 len = me.y;
        u32 src = cursrc[i];
        dep = depth[i];
        bool finished = false;
        for (u32 step = 0; step < max_steps; ++step) {      // a long chain continues in the next round, regrouped with its peers
            if ((size_t)src >= n || dep >= n) { finished = true; break; }   // :106 src < fmap.size()  (dep bound: no endless chains)
            const u32 r = owner[src];
            if (r == NONE32) { finished = true; break; }                    // :106 fmap[src] == 0
            const uint4 sr = rec[r];
            const u32 d = src - sr.x;
            if ((u64)d + len > sr.y) { finished = true; break; }            // :110 copy does not fit inside s
            u32 ssrc;
            if (r < i) {                                                    // earlier factor: needs its final source
                ssrc = sr.w;
                if (ssrc == NOT_DONE) break;                                // wait for the next round
            } else {
                ssrc = sr.z;                                                // later factor: still unflattened at this point
            }
            src = ssrc + d;                                                 // :111
            ++dep;
        }
        cursrc[i] = src;
        depth[i] = dep;
        wcls[j] = finished ? 0 : 1;
        if (finished) {
            ((u32*)&rec[i])[3] = dep ? src : me.z;                          // :122-124
            counted = dep != 0;
        }
    }
    // statistics: one atomic pair per wave
    const u64 b = __ballot(counted);
    const u32 mx = wave_reduce_max(counted ? dep : 0u);
    if (lane_id() == 0 && b) { atomicAdd(&sc->num_flattened, (u32)__popcll(b)); atomicMax(&sc->max_depth, mx); }
}

__global__ void flatten_iota_kernel(u32* __restrict__ a, u32 m) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) a[j] = j;
}

__global__ void flatten_commit_kernel(size_t z, const uint4* __restrict__ rec, u32* __restrict__ fsrc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= z) return;
    const uint4 q = rec[i];
    fsrc[q.x] = q.w;
}

void flatten_factors(Ctx& c, size_t n, FactorSpace fs, FlattenStats* st) {
    FlattenStats local;
    if (!st) st = &local;
    *st = FlattenStats();
    if (n == 0) return;
    hipStream_t s = c.stream;
    const size_t mark = c.arena.mark();
    u32* fpos = fs.have_list ? fs.fpos : c.arena.get<u32>(n);
    const size_t z = fs.have_list ? fs.nfact : extract_factors(c, n, fs, fpos, nullptr, nullptr, n);
    if (z == 0) { c.arena.release(mark); return; }
    uint4* rec = (uint4*)c.arena.alloc(z * sizeof(uint4));
    u32* cursrc = c.arena.get<u32>(z);
    u32* depth = c.arena.get<u32>(z);
    FlattenScalars* d_sc = (FlattenScalars*)c.arena.alloc(sizeof(FlattenScalars));
    HIP_TRY(hipMemsetAsync(d_sc, 0, sizeof(FlattenScalars), s));
    const unsigned gz = cdiv(z, 256);
    flatten_init_kernel<<<gz, 256, 0, s>>>(fpos, z, fs.flen, fs.fsrc, rec, cursrc, depth);
    LAUNCH_CHECK();
    // work lists of the still-waiting factors, compacted after every round
    u32* work[2] = { c.arena.get<u32>(z), c.arena.get<u32>(z) };
    u8* wcls = c.arena.get<u8>(z);
    u32* ident = c.arena.get<u32>(z);
    flatten_iota_kernel<<<gz, 256, 0, s>>>(ident, (u32)z);
    LAUNCH_CHECK();
    u32 waiting = (u32)z;
    int cur_w = -1;                               // -1: the identity list (first round)
    // steps per round: few in the first rounds (most chains are short; the lanes of a wave wait for the longest one),
    // doubling afterwards
    u32 max_steps = getenv("TDC_GPU_FLATTEN_STEPS") ? (u32)atoi(getenv("TDC_GPU_FLATTEN_STEPS")) : 1u;   // measured: 1,2,4,.. 8.5 ms; unlimited 11.2 ms
    if (max_steps == 0) max_steps = 1u << 30;
    // budget growth per round (measured at 256 MiB: x2 8.4 ms, x4 7.4 ms, x8 7.0 ms)
    u32 flat_growth = getenv("TDC_GPU_FLATTEN_GROWTH") ? (u32)atoi(getenv("TDC_GPU_FLATTEN_GROWTH")) : 8u;
    if (flat_growth < 2) flat_growth = 2;
    u32 stalled = 0;
    while (waiting) {
        {   // per waiting factor: record, cursrc, depth (24) + one chain step (block word + record: 20) + outputs (13)
            Ctx::ProfScope prof(c, K_FLATTEN_ROUND, (u64)waiting * 57);
            flatten_round_kernel<<<cdiv(waiting, 256), 256, 0, s>>>(cur_w < 0 ? nullptr : work[cur_w], waiting, n, fs.owner, rec,
                                                                    cursrc, depth, wcls, d_sc, max_steps);
            LAUNCH_CHECK();
        }
        const int nxt = cur_w < 0 ? 0 : (cur_w ^ 1);
        select_by_class(c, wcls, 1, waiting, cur_w < 0 ? ident : work[cur_w], work[nxt], nullptr, nullptr, &d_sc->waiting);
        const u32 now = c.read(&d_sc->waiting);
        st->rounds++;
        if (getenv("TDC_GPU_LEVEL_LOG")) fprintf(stderr, "flatten round %u: %u waiting -> %u\n", st->rounds, waiting, now);
        stalled = (now == waiting) ? stalled + 1 : 0;         // (a round with a small budget may finish nothing; never many in a row)
        if (now > waiting || (now == waiting && max_steps >= (1u << 30)) || stalled > 40)
            throw HipError{hipErrorUnknown, "flatten: rounds made no progress", (int)__LINE__};
        waiting = now;
        cur_w = nxt;
        if (max_steps < (1u << 30)) max_steps = (max_steps > (1u << 30) / flat_growth) ? (1u << 30) : max_steps * flat_growth;
    }
    flatten_commit_kernel<<<gz, 256, 0, s>>>(z, rec, fs.fsrc);
    LAUNCH_CHECK();
    FlattenScalars h = c.read(d_sc);
    st->num_flattened = h.num_flattened;
    st->max_depth_lb = h.max_depth;
    c.arena.release(mark);
}

}  // namespace tdc
