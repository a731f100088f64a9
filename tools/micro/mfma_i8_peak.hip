// Micro-benchmark (GPU box): the dense int8 matrix pipe (v_mfma_i32_32x32x32_i8, nominal 2x the fp16 rate) under the package
// power limit, next to tools/micro/mfma_peak.hip -- would fixed-point digits (six int8 digit products per fp32-equivalent
// product at twice the rate = the same pipe time as three fp16 products) draw fewer joules per product?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
template <int ROT>
__global__ __launch_bounds__(512) void spin(const i4 *ops, int *out, int iters) {
    const int lane = threadIdx.x & 63;
    i4 a[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = ops[(2 * k) * 64 + lane]; b[k] = ops[(2 * k + 1) * 64 + lane]; }
    i16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = ROT ? 2 * u : 0;
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[k], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[k + 1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k + 1], b[k + 1], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k + 1], b[k], c3, 0, 0, 0);
        }
    }
    int s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123457) out[0] = s;
}
int main() {
    std::vector<int> h(32 * 64 * 4);
    i4 *d; int *o;
    (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&o, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind) {
        unsigned x = 12345;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = kind == 0 ? 0 : (int)(x ^ (x >> 13)); }      // zeros | random bytes
        (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int waves = 1; waves <= 2; ++waves) {
            const int iters = 40000, nthreads = 256 * waves, nwg = 256;
            float last = 0;
            for (int rep = 0; rep < 6; ++rep) {
                (void)hipEventRecord(e0);
                if (kind == 2) hipLaunchKernelGGL(spin<1>, dim3(nwg), dim3(nthreads), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(spin<0>, dim3(nwg), dim3(nthreads), 0, 0, d, o, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&last, e0, e1);
            }
            const double n_inst = (double)nwg * (nthreads / 64) * iters * 32.0;
            const double tops = n_inst * 2.0 * 32 * 32 * 32 / (last * 1e-3) / 1e12;
            printf("operands %-6s  %d wave(s) per SIMD: %.2f ms: %.0f TOP/s dense int8 (nominal 5000) = %.2f of nominal\n",
                   kind == 0 ? "zero" : kind == 1 ? "random" : "rotate", waves, last, tops, tops / 5000.0);
        }
    }
    return 0;
}
