// Development aid: rate of dependent random 32-byte-sector gathers as a function of the table size (is a table that fits the
// 256 MB Infinity Cache cheaper to gather from than an 8 GB one?).  hipcc -O3 --offload-arch=gfx950 gather_bench.hip -o gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// every thread: K independent gathers of one u32 from table[idx], idx pseudo-random; second mode: gather uint4 (16 B) as well
template <int K, bool SECOND>
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ t1, size_t n1, const uint4* __restrict__ t2, size_t n2, size_t items, uint32_t* out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * K;
    if (i >= items) return;
    uint32_t acc = 0;
    uint32_t v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = t1[mix(i + k) % n1];
    if (SECOND) {
        uint4 w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = t2[(mix(i + k + 77) ^ v[k]) % n2];
#pragma unroll
        for (int k = 0; k < K; ++k) acc += w[k].x + w[k].w;
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t items = 150000000;
    uint32_t* out; CK(hipMalloc(&out, 64));
    const size_t n2 = 150000000;           // 2.4 GB of 16-byte records
    uint4* t2; CK(hipMalloc(&t2, n2 * 16)); CK(hipMemset(t2, 1, n2 * 16));
    const size_t sizes_mb[] = { 16, 64, 128, 200, 285, 512, 1024, 2048, 8192 };
    for (size_t mb : sizes_mb) {
        const size_t n1 = mb * 1024 * 1024 / 4;
        uint32_t* t1; CK(hipMalloc(&t1, n1 * 4)); CK(hipMemset(t1, 1, n1 * 4));
        for (int second = 0; second < 2; ++second) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                if (second) gather<4, true><<<(items / 4 + 255) / 256, 256>>>(t1, n1, t2, n2, items, out);
                else gather<4, false><<<(items / 4 + 255) / 256, 256>>>(t1, n1, t2, n2, items, out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            printf("table %5zu MB %s: %7.3f ms for %zu items = %6.1f G gathers/s\n", mb, second ? "+ dependent 16 B record gather (2.4 GB)" : "only", best, items,
                   items / best / 1e6 * (second ? 2 : 1));
        }
        CK(hipFree(t1));
    }
    return 0;
}
