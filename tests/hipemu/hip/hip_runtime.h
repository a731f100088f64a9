// TEST INFRASTRUCTURE -- never part of the product.
// A host-side stand-in for <hip/hip_runtime.h> that lets the unmodified kernel sources of patch2pix_amd/csrc be
// compiled with clang++ for x86 and executed on the CPU: every work-item is a fiber, work-groups run one after the
// other (several OS threads take a work-group each), wave64 collectives (__shfl_xor, the two MFMA shapes the
// kernels use) exchange operands through a per-wave mailbox.  It exists so that the index arithmetic of a kernel
// can be checked against the oracle and the golden vectors without a GPU; it says nothing about speed, and the
// library built from it (tests/hipemu/_build/libp2p_emu.so) is only ever loaded by tests/test_kernels_emulated.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __HIPEMU__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local          /* one copy per OS thread = per running work-group */

// ---- overrides of the two indirections in csrc/p2p_common.h
#define P2P_OPAQUE(v) asm volatile("" : "+r"(v))
#define P2P_OPAQUE_S(v) asm volatile("" : "+r"(v))
#define P2P_OPAQUE_V4(v) asm volatile("" : "+x"(v))
#define P2P_SWAP_ADJACENT(v) ((unsigned)__shfl_xor((int)(v), 1))
#define P2P_SWAP_PAIRS(v) ((unsigned)__shfl_xor((int)(v), 2))
#define P2P_LANE_ID() ((int)(hipemu::ctx().thread.x & 63))
#define P2P_DYN_SHARED(T, name) T *name = (T *)hipemu::dynamic_shared()
#define P2P_WAVE_SYNC() do { char z_ = 0; (void)hipemu::wave_gather(&z_, 1); } while (0)   /* the lanes are fibers: rendezvous */
// LDS-DMA (global_load_lds_dwordx4): the copy happens at once -- a kernel that is correct for ANY completion time before its
// counted wait is correct for this one; what the stand-in cannot see is a wait that is missing or counts wrongly
#define P2P_GLOBAL_LOAD_LDS16(gptr, lptr, imm) \
    memcpy((unsigned char *)(lptr) + (imm) + 16 * P2P_LANE_ID(), (const unsigned char *)(gptr) + (imm), 16)
#define P2P_WAIT_VMCNT(n) ((void)0)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace hipemu {
struct Idx { unsigned x, y, z; };
struct Ctx {                       // what a work-item knows about itself
    Idx thread, block;
    dim3 bdim, gdim;
};
Ctx &ctx();
void *dynamic_shared();
void sync_block();
// Wave64 mailbox: every live lane of the calling lane's wave deposits `bytes` bytes; returns the 64 deposits
// (slot = lane), valid until the lane's next collective.
const unsigned char *wave_gather(const void *mine, size_t bytes);
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);
}  // namespace hipemu

#define threadIdx (hipemu::ctx().thread)
#define blockIdx (hipemu::ctx().block)
#define blockDim (hipemu::ctx().bdim)
#define gridDim (hipemu::ctx().gdim)
#define __syncthreads() hipemu::sync_block()

// ---- host API subset used by the library
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef struct hipemu_stream *hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
// three "compute units" (HIPEMU_CUS overrides, read at the first query): launches of persistent work-groups walk several
// items each
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) {
    const char *e = getenv("HIPEMU_CUS");
    *v = e ? atoi(e) : 3;
    return hipSuccess;
}
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) {
    *p = (T *)aligned_alloc(256, (n + 255) & ~size_t(255));
    return *p ? hipSuccess : hipErrorUnknown;
}
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    hipemu::launch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })

// ---- device intrinsics
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
using std::max;
using std::min;
static inline int atomicMax(int *p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMin(int *p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline float unsafeAtomicAdd(float *p, float v) {
    unsigned old = __atomic_load_n((unsigned *)p, __ATOMIC_RELAXED), want;
    float cur;
    do {
        cur = __uint_as_float(old);
        want = __float_as_uint(cur + v);
    } while (!__atomic_compare_exchange_n((unsigned *)p, &old, want, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return cur;
}
#define __threadfence() __sync_synchronize()
#define __builtin_amdgcn_readfirstlane(x) (x)      /* only applied to wave-uniform values in these kernels */
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
#define __builtin_amdgcn_s_memtime() (0ull)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_barrier() hipemu::sync_block()

template <class T> static inline T __shfl_xor(T v, int mask) {
    const unsigned char *all = hipemu::wave_gather(&v, sizeof(T));
    T out;
    memcpy(&out, all + ((hipemu::ctx().thread.x & 63) ^ mask) * sizeof(T), sizeof(T));
    return out;
}

template <class T> static inline T __shfl_up(T v, int delta) {
    const unsigned char *all = hipemu::wave_gather(&v, sizeof(T));
    const int lane = hipemu::ctx().thread.x & 63;
    T out = v;
    if (lane >= delta) memcpy(&out, all + (lane - delta) * sizeof(T), sizeof(T));
    return out;
}

// v_mfma_f32_32x32x2_f32: D = A(32x2) B(2x32) + C.  Lane l holds A[l&31][l>>5], B[l>>5][l&31]; register r of
// C/D is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
static inline hipemu_f32x16 hipemu_mfma_32x32x2_f32(float a, float b, hipemu_f32x16 c) {
    struct { float a, b; } mine = {a, b};
    const unsigned char *all = hipemu::wave_gather(&mine, sizeof(mine));
    const int lane = hipemu::ctx().thread.x & 63, col = lane & 31, half = lane >> 5;
    auto A = [&](int row, int k) { float v; memcpy(&v, all + (row + 32 * k) * 8, 4); return v; };
    auto B = [&](int k, int cc) { float v; memcpy(&v, all + (cc + 32 * k) * 8 + 4, 4); return v; };
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        c[r] = fmaf(A(row, 1), B(1, col), fmaf(A(row, 0), B(0, col), c[r]));
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_32x32x2_f32((a), (b), (c))

// v_mfma_f32_16x16x4_f32: D = A(16x4) B(4x16) + C.  Lane l holds A[l&15][l>>4], B[l>>4][l&15]; register r of C/D is row
// 4*(l>>4) + r, column l&15.
typedef float hipemu_f32x4_ __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4_ hipemu_mfma_16x16x4_f32(float a, float b, hipemu_f32x4_ c) {
    struct { float a, b; } mine = {a, b};
    const unsigned char *all = hipemu::wave_gather(&mine, sizeof(mine));
    const int lane = hipemu::ctx().thread.x & 63, col = lane & 15, blk = lane >> 4;
    auto A = [&](int row, int k) { float v; memcpy(&v, all + (row + 16 * k) * 8, 4); return v; };
    auto B = [&](int k, int cc) { float v; memcpy(&v, all + (cc + 16 * k) * 8 + 4, 4); return v; };
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * blk + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(A(row, k), B(k, col), acc);
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4_f32((a), (b), (c))

// v_mfma_f32_32x32x16_bf16: A 32x16, B 16x32 in bf16, fp32 accumulate.  Lane l holds A[l&31][8*(l>>5) + i],
// B[8*(l>>5) + i][l&31], i = 0..7; C/D as above.
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c) {
    struct { unsigned short a[8], b[8]; } mine;
    memcpy(mine.a, &a, 16);
    memcpy(mine.b, &b, 16);
    const unsigned char *all = hipemu::wave_gather(&mine, sizeof(mine));
    const int lane = hipemu::ctx().thread.x & 63, col = lane & 31, half = lane >> 5;
    auto bf = [](unsigned short u) { return __uint_as_float((unsigned)u << 16); };
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            unsigned short ua, ub;
            memcpy(&ua, all + (row + 32 * (k >> 3)) * 32 + 2 * (k & 7), 2);
            memcpy(&ub, all + (col + 32 * (k >> 3)) * 32 + 16 + 2 * (k & 7), 2);
            acc += bf(ua) * bf(ub);      // products of bf16 are exact in fp32
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma_32x32x16_bf16((a), (b), (c))

// v_mfma_f32_16x16x32_bf16: A 16x32, B 32x16 in bf16, fp32 accumulate.  Lane l holds A[l&15][8*(l>>4) + i],
// B[8*(l>>4) + i][l&15], i = 0..7; register r of C/D is row 4*(l>>4) + r, column l&15.
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c) {
    struct { unsigned short a[8], b[8]; } mine;
    memcpy(mine.a, &a, 16);
    memcpy(mine.b, &b, 16);
    const unsigned char *all = hipemu::wave_gather(&mine, sizeof(mine));
    const int lane = hipemu::ctx().thread.x & 63, col = lane & 15, blk = lane >> 4;
    auto bf = [](unsigned short u) { return __uint_as_float((unsigned)u << 16); };
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * blk + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            unsigned short ua, ub;
            memcpy(&ua, all + (row + 16 * (k >> 3)) * 32 + 2 * (k & 7), 2);
            memcpy(&ub, all + (col + 16 * (k >> 3)) * 32 + 16 + 2 * (k & 7), 2);
            acc += bf(ua) * bf(ub);
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu_mfma_16x16x32_bf16((a), (b), (c))

// the fp16 forms of the two shapes (same lane layouts; products of fp16 are exact in fp32)
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x16 hipemu_mfma_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c) {
    struct { _Float16 a[8], b[8]; } mine;
    memcpy(mine.a, &a, 16);
    memcpy(mine.b, &b, 16);
    const unsigned char *all = hipemu::wave_gather(&mine, sizeof(mine));
    const int lane = hipemu::ctx().thread.x & 63, col = lane & 31, half = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 ua, ub;
            memcpy(&ua, all + (row + 32 * (k >> 3)) * 32 + 2 * (k & 7), 2);
            memcpy(&ub, all + (col + 32 * (k >> 3)) * 32 + 16 + 2 * (k & 7), 2);
            acc += (float)ua * (float)ub;
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu_mfma_32x32x16_f16((a), (b), (c))
static inline hipemu_f32x4 hipemu_mfma_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x4 c) {
    struct { _Float16 a[8], b[8]; } mine;
    memcpy(mine.a, &a, 16);
    memcpy(mine.b, &b, 16);
    const unsigned char *all = hipemu::wave_gather(&mine, sizeof(mine));
    const int lane = hipemu::ctx().thread.x & 63, col = lane & 15, blk = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * blk + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            _Float16 ua, ub;
            memcpy(&ua, all + (row + 16 * (k >> 3)) * 32 + 2 * (k & 7), 2);
            memcpy(&ub, all + (col + 16 * (k >> 3)) * 32 + 16 + 2 * (k & 7), 2);
            acc += (float)ua * (float)ub;
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu_mfma_16x16x32_f16((a), (b), (c))
