// Development probe (not part of the library): does the bucket stride of a 512-way scatter matter on this GPU?
// Mimics the write pattern of fs_scatter_kernel (fused.hip): a tile of 8192 records leaves 16 consecutive records in each of 512
// buckets, 4-byte words into one array and 8-byte words into another; bucket b starts at b * stride records and all buckets fill at
// the same rate (every text position occurs once, so the buckets of a position-keyed partition have exactly 2^k records).
//   hipcc -O3 --offload-arch=gfx950 scatter_stride_probe.hip -o scatter_stride_probe && ./scatter_stride_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void scatter(const uint32_t* __restrict__ in4, const uint64_t* __restrict__ in8, uint32_t* __restrict__ out4,
                                               uint64_t* __restrict__ out8, size_t stride, uint32_t tiles) {
    const uint32_t tile = blockIdx.x;
    if (tile >= tiles) return;
    const size_t base = (size_t)tile * 8192;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t e = (uint32_t)i * 512u + threadIdx.x;
        const uint32_t b = e >> 4, w = e & 15u;
        const size_t o = (size_t)b * stride + (size_t)tile * 16 + w;
        out4[o] = in4[base + e];
        out8[o] = in8[base + e];
    }
}

int main() {
    const size_t m = 2000000000ull;
    const uint32_t tiles = (uint32_t)(m / 8192);
    const size_t strides[] = { (size_t)1 << 22, 3906251, ((size_t)1 << 22) + 32, ((size_t)1 << 22) + 1024, (size_t)1 << 22, 4000037 };
    const size_t cap = 512 * (((size_t)1 << 22) + 1024) + 8192;
    uint32_t *in4, *out4; uint64_t *in8, *out8;
    CK(hipMalloc(&in4, m * 4)); CK(hipMalloc(&in8, m * 8)); CK(hipMalloc(&out4, cap * 4)); CK(hipMalloc(&out8, cap * 8));
    CK(hipMemset(in4, 1, m * 4)); CK(hipMemset(in8, 2, m * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t st : strides) {
        if ((size_t)tiles * 16 > st) { printf("stride %zu too small\n", st); continue; }
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            scatter<<<tiles, 512>>>(in4, in8, out4, out8, st, tiles);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("stride %10zu records: %.2f ms  (%.2f TB/s of 24 B per record)\n", st, best, (double)m * 24 / best / 1e9);
        fflush(stdout);
    }
    return 0;
}
