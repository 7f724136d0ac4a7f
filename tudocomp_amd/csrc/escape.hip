// escape.hip -- input restrictions "escape {0} + null-terminate" on the device.
// Replaces io::RestrictedBuffer::escape_with_iters + EscapeMap (io/RestrictedBuffer.hpp:43-74, io/EscapeMap.hpp:39-64):
//   0x00 -> FF FE,  0xFF -> FF FF,  every other byte unchanged, one 0x00 appended.
// flag (bytes that grow) -> exclusive scan -> scatter; KAT: test/tudocomp_tests.cpp:528-556.
#include "stages.hpp"
#include "prim.hpp"

namespace tdc {

__global__ void escape_flag_kernel(const u8* __restrict__ in, size_t n, u32* __restrict__ extra) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const u8 ch = in[i]; extra[i] = (ch == 0x00 || ch == 0xFF) ? 1u : 0u; }
}
__global__ void escape_scatter_kernel(const u8* __restrict__ in, size_t n, const u32* __restrict__ extra_before, u8* __restrict__ out,
                                      size_t out_len) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u8 ch = in[i];
        const size_t o = i + extra_before[i];
        if (ch == 0x00) { out[o] = 0xFF; out[o + 1] = 0xFE; }
        else if (ch == 0xFF) { out[o] = 0xFF; out[o + 1] = 0xFF; }
        else out[o] = ch;
    }
    if (i == 0) out[out_len - 1] = 0;             // the sentinel
}

__global__ void escape_count_kernel(const u8* __restrict__ in, size_t n, unsigned long long* __restrict__ cnt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32 local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const u8 ch = in[i]; local += (ch == 0x00 || ch == 0xFF) ? 1u : 0u; }
    local = wave_reduce_sum(local);
    if (lane_id() == 0 && local) atomicAdd(cnt, (unsigned long long)local);
}
// number of bytes the escaping adds (0x00 and 0xFF bytes of the input)
size_t count_escapes_device(Ctx& c, const u8* d_in, size_t n) {
    if (!n) return 0;
    const size_t mark = c.arena.mark();
    unsigned long long* d_cnt = (unsigned long long*)c.arena.alloc(sizeof(unsigned long long));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), c.stream));
    unsigned g = cdiv(n, 256 * 16); if (g > 4096) g = 4096; if (g == 0) g = 1;
    escape_count_kernel<<<g, 256, 0, c.stream>>>(d_in, n, d_cnt);
    LAUNCH_CHECK();
    u32 two[2];
    c.read_n((const u32*)d_cnt, two, 2);
    c.arena.release(mark);
    return (size_t)two[0] | ((size_t)two[1] << 32);
}

// d_out must hold the escaped length (n + #escapes + 1 bytes); returns the escaped length (incl. sentinel)
size_t escape_device(Ctx& c, const u8* d_in, size_t n, u8* d_out) {
    const size_t mark = c.arena.mark();
    u32* extra = c.arena.get<u32>(n + 1);
    u32* d_total = c.arena.get<u32>(1);
    size_t out_len = 1;
    if (n) {
        const unsigned g = cdiv(n, 256);
        escape_flag_kernel<<<g, 256, 0, c.stream>>>(d_in, n, extra);
        LAUNCH_CHECK();
        exclusive_sum_u32(c, extra, extra, n, d_total);
        out_len = n + c.read(d_total) + 1;
        escape_scatter_kernel<<<g, 256, 0, c.stream>>>(d_in, n, extra, d_out, out_len);
        LAUNCH_CHECK();
    } else {
        HIP_TRY(hipMemsetAsync(d_out, 0, 1, c.stream));
    }
    c.arena.release(mark);
    return out_len;
}

}  // namespace tdc
