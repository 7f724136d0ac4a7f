// textds.hip -- Phi, PLCP and (debug) LCP arrays.
//   build_phi  : ds/PhiFromSA.hpp:35-45        phi[sa[i]] = sa[i-1], phi[sa[0]] = sa[n-1]
//   build_plcp : ds/PLCPFromPhi.hpp:27-53      plcp[i] = lcp(T[i..], T[phi[i]..]) for i < n-1 ; plcp[n-1] := 0
//   build_lcp  : ds/LCPFromPLCP.hpp:27-56      lcp[0] = 0, lcp[i] = plcp[sa[i]]
// The PLCP array is a function of the text alone, so the chunked evaluation below (each thread restarts the
// Phi-algorithm's carry l = 0 at the start of its chunk and then uses plcp[i+1] >= plcp[i] - 1) is bit-identical
// to the reference's sequential loop.
#include "stages.hpp"

namespace tdc {

__global__ void phi_kernel(const u32* __restrict__ sa, size_t n, u32* __restrict__ phi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 s = sa[i];
    phi[s] = (i == 0) ? sa[n - 1] : sa[i - 1];
}

void build_phi(Ctx& c, const u32* sa, size_t n, u32* phi) {
    if (!n) return;
    Ctx::ProfScope prof(c, K_PHI, (u64)n * 8);                  // read SA, scatter Phi
    phi_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(sa, n, phi);
    LAUNCH_CHECK();
}

constexpr int PLCP_CHUNK = 32;   // consecutive text positions per thread

__global__ __launch_bounds__(256) void plcp_kernel(const u8* __restrict__ text, size_t n, const u32* __restrict__ phi,
                                                    u32* __restrict__ plcp, u32* __restrict__ d_max) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t begin = t * PLCP_CHUNK;
    u32 mx = 0;
    if (begin < n) {
        size_t end = begin + PLCP_CHUNK;
        if (end > n) end = n;
        u32 l = 0;
        for (size_t i = begin; i < end; ++i) {
            if (i == n - 1) { plcp[i] = 0; break; }
            const size_t j = phi[i];
            // the sentinel T[n-1] is unique, so the comparison stops before either index leaves the text;
            // the explicit bounds only keep a corrupted Phi from faulting
            while (i + l < n && j + l < n && text[i + l] == text[j + l]) ++l;
            plcp[i] = l;
            mx = max(mx, l);
            if (l) --l;
        }
    }
    mx = wave_reduce_max(mx);
    if (lane_id() == 0 && mx) atomicMax(d_max, mx);
}

void build_plcp(Ctx& c, const u8* text, size_t n, const u32* phi, u32* plcp, u32* d_maxlcp) {
    HIP_TRY(hipMemsetAsync(d_maxlcp, 0, sizeof(u32), c.stream));
    if (!n) return;
    const size_t threads = (n + PLCP_CHUNK - 1) / PLCP_CHUNK;
    Ctx::ProfScope prof(c, K_PLCP, (u64)n * 10);                // Phi (4) + two text bytes + PLCP (4), SURVEY 8d
    plcp_kernel<<<cdiv(threads, 256), 256, 0, c.stream>>>(text, n, phi, plcp, d_maxlcp);
    LAUNCH_CHECK();
}

__global__ void lcp_kernel(const u32* __restrict__ sa, const u32* __restrict__ plcp, size_t n, u32* __restrict__ lcp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lcp[i] = (i == 0) ? 0u : plcp[sa[i]];
}

void build_lcp(Ctx& c, const u32* sa, const u32* plcp, size_t n, u32* lcp) {
    if (!n) return;
    lcp_kernel<<<cdiv(n, 256), 256, 0, c.stream>>>(sa, plcp, n, lcp);
    LAUNCH_CHECK();
}

}  // namespace tdc
