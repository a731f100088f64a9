// Micro-benchmark (GPU box): how fast does ONE work-group per compute unit drain a burst of global stores?
// 256 work-groups x 512 threads, each work-group writes BYTES_PER_WG contiguous-per-instruction (1 KiB per wave store),
// then signals completion (s_waitcnt vmcnt(0)); variants: plain / nontemporal stores, region stride.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(512) void burst(unsigned char *dst, size_t wg_stride, int kb_per_wg, int reps, unsigned long long *ticks, int spin) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long t = 0;
    for (int r = 0; r < reps; ++r) {
        unsigned char *base = dst + ((size_t)r * gridDim.x + blockIdx.x) * wg_stride;      // a fresh region per burst
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        // each wave instruction: 64 lanes x 16 B = 1 KiB contiguous; 8 waves interleave KiB-wise
        for (int i = wave; i < kb_per_wg; i += 8) {
            u4 v = {(unsigned)i, (unsigned)lane, (unsigned)r, 7u};
            u4 *p = (u4 *)(base + (size_t)i * 1024 + lane * 16);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t += __builtin_amdgcn_s_memtime() - t0;
        // some compute between bursts so that the bursts are separated
        float x = (float)tid;
        for (int k = 0; k < spin; ++k) x = x * 1.0001f + 0.5f;
        if (x == 12345.f) dst[0] = 1;
    }
    if (tid == 0) ticks[blockIdx.x] = t;
}
int main() {
    const int nwg = 256, reps = 24;
    unsigned char *d; unsigned long long *tk;
    const size_t total = (size_t)4 << 30;
    hipMalloc(&d, total); hipMalloc(&tk, nwg * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int kb : {512})
            for (int spin : {0, 20000, 80000}) {
                const size_t stride = (size_t)kb * 1024;
                for (int it = 0; it < 2; ++it) {
                    hipEventRecord(e0);
                    if (nt) hipLaunchKernelGGL(burst<1>, dim3(nwg), dim3(512), 0, 0, d, stride, kb, reps, tk, spin);
                    else hipLaunchKernelGGL(burst<0>, dim3(nwg), dim3(512), 0, 0, d, stride, kb, reps, tk, spin);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> h(nwg); hipMemcpy(h.data(), tk, nwg * 8, hipMemcpyDeviceToHost);
                double avg = 0; for (auto v : h) avg += (double)v / reps; avg /= nwg;
                // s_memtime ticks at 100 MHz
                printf("nt=%d %4d KB per WG per burst, spin %6d: burst %.0f memtime ticks; launch %.3f ms = %.1f us per (burst + spin); %.2f TB/s if all of it were the burst\n",
                       nt, kb, spin, avg, ms, ms * 1e3 / reps, nwg * kb * 1024.0 * reps / (ms * 1e-3) / 1e12);
            }
    return 0;
}
