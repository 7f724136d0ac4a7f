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

// d_out must hold 2*n + 1 bytes; returns the escaped length (incl. sentinel)
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
