// Micro-benchmark (GPU box): what does the dense fp16 matrix pipe deliver under the package power limit?
// Every SIMD runs WAVES waves that issue nothing but v_mfma_f32_32x32x16_f16 on register operands (four independent
// accumulators per wave).  Operands: random fp16 in [-1, 1) scaled like the library's planes, or all zeros.
// Prints TFLOP/s (2 * 32 * 32 * 16 flop per instruction) and the effective clock implied by the issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// ROT: every instruction reads operand registers different from the previous one's (16 A and 16 B fragments in rotation), as
// a GEMM does; else the same four fragments are re-used (only products and accumulators toggle).
template <int ROT>
__global__ __launch_bounds__(512) void spin(const h8 *ops, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = ops[(2 * k) * 64 + lane]; b[k] = ops[(2 * k + 1) * 64 + lane]; }
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = ROT ? 2 * u : 0;
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k + 1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k + 1], b[k + 1], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k + 1], b[k], c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 1.2345f) out[0] = s;
}
int main() {
    std::vector<_Float16> h(32 * 64 * 8);
    h8 *d; float *o;
    (void)hipMalloc(&d, h.size() * 2); (void)hipMalloc(&o, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int kind = 0; kind < 4; ++kind) {
        unsigned x = 12345;
        for (auto &v : h) {
            x = x * 1664525u + 1013904223u;
            const float r = ((x >> 8) & 0xffff) / 32768.0f - 1.0f;
            v = (_Float16)(kind == 0 ? 0.f : kind == 1 ? r * 0.01f : r * 4096.0f);      // zeros | small magnitudes | the planes' range, 4 fragments re-used | the same, 32 fragments in rotation
        }
        (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int waves = 1; waves <= 2; ++waves) {
            const int iters = 40000, nthreads = 256 * waves, nwg = 256;
            float best = 1e30f, last = 0;
            for (int rep = 0; rep < 6; ++rep) {         // ~100 ms of continuous load per setting: the power controller settles
                (void)hipEventRecord(e0);
                if (kind == 3) hipLaunchKernelGGL(spin<1>, dim3(nwg), dim3(nthreads), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(spin<0>, dim3(nwg), dim3(nthreads), 0, 0, d, o, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&last, e0, e1);
                if (rep >= 2 && last < best) best = last;
            }
            const double n_inst = (double)nwg * (nthreads / 64) * iters * 32.0;
            const double tf = n_inst * 2.0 * 32 * 32 * 16 / (last * 1e-3) / 1e12;
            // one instruction occupies a SIMD's matrix pipe for 32 cycles (8 passes x 4)
            const double ghz = n_inst * 32.0 / (1024.0) / (last * 1e-3) / 1e9;
            printf("operands %-6s  %d wave(s) per SIMD: last launch %.2f ms (best %.2f): %.0f TFLOP/s dense fp16 = %.2f busy-GHz per SIMD (2500 TFLOP/s = 2.4 GHz)\n",
                   kind == 0 ? "zero" : kind == 1 ? "small" : kind == 2 ? "random" : "rotate", waves, last, best, tf, ghz);
        }
    }
    return 0;
}
