// Development aid: host-to-device rate of 2 GB of page-locked memory through one or several streams / chunk sizes
// hipcc -O3 --offload-arch=gfx950 h2d_bench.hip -o h2d_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t N = 2000000000;
    char* h; CK(hipHostMalloc((void**)&h, N, hipHostMallocDefault)); memset(h, 1, N);
    char* d; CK(hipMalloc((void**)&d, N));
    hipStream_t s[4]; for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    for (int ns = 1; ns <= 4; ++ns)
        for (size_t chunk : { (size_t)16 << 20, (size_t)125000000, N }) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                int k = 0;
                for (size_t a = 0; a < N; a += chunk, ++k) { const size_t b = a + chunk < N ? a + chunk : N; CK(hipMemcpyAsync(d + a, h + a, b - a, hipMemcpyHostToDevice, s[k % ns])); }
                for (int i = 0; i < ns; ++i) CK(hipStreamSynchronize(s[i]));
                const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (t < best) best = t;
            }
            printf("H2D %d stream(s), chunk %10zu: %7.2f ms = %5.1f GB/s\n", ns, chunk, best, N / best / 1e6);
        }
    // device-to-host for comparison
    for (int ns = 1; ns <= 2; ++ns) {
        double best = 1e9;
        const size_t chunk = (size_t)125000000;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            int k = 0;
            for (size_t a = 0; a < N; a += chunk, ++k) { const size_t b = a + chunk < N ? a + chunk : N; CK(hipMemcpyAsync(h + a, d + a, b - a, hipMemcpyDeviceToHost, s[k % ns])); }
            for (int i = 0; i < ns; ++i) CK(hipStreamSynchronize(s[i]));
            const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (t < best) best = t;
        }
        printf("D2H %d stream(s), chunk %10zu: %7.2f ms = %5.1f GB/s\n", ns, chunk, best, N / best / 1e6);
    }
    return 0;
}
