// Shared declarations for libp2p_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../include/p2p_hip.h"

// Timing experiments (kernel variants that drop or pin a part of the work, phase stamps; most of them produce WRONG results
// by design; switches whose question is settled are retired once their ablation is logged under profiles/) exist only behind
// -DP2P_EXPERIMENT: such a build reports P2P_VERSION_EXPERIMENT in p2p_version() and the
// Python binding refuses to load it unless P2P_ALLOW_EXPERIMENT=1 is set (tools/ab_variants.sh does).  A product build with
// one of the switches defined does not compile.
#if !defined(P2P_EXPERIMENT) &&                                                                                          \
    (defined(XF_PIN_W) || defined(XF_SAME_PATCH) || defined(XF_SKIP_P) || defined(XF_SKIP_C) || defined(XF_SKIP_FOLD) ||   \
     defined(XF_SKIP_CONV2) || defined(XF_SKIP_FC) || defined(XF_GRID_CAP) || defined(P2P_WINO_CHUNK) || defined(P2P_X3_TIMING) || \
     defined(NCF_TIMING) || defined(NCF_NOPRIO))
#error "experiment switches (XF_*, P2P_X3_TIMING, NCF_TIMING, NCF_NOPRIO, P2P_WINO_CHUNK) need -DP2P_EXPERIMENT: the library then identifies itself as an experiment build"
#endif
#define P2P_VERSION_EXPERIMENT 0x40000000

namespace p2p {

void set_error(const char *fmt, ...);

#define P2P_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            p2p::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return P2P_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define P2P_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            p2p::set_error(__VA_ARGS__);                                                 \
            return (code);                                                               \
        }                                                                                \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Two constructs the host-side kernel emulator of the test-suite (tests/hipemu) has to see through; for hipcc
// they expand to exactly the code they replace.
#ifndef P2P_OPAQUE                 // make a VGPR value opaque to the optimiser (stops hoisting of lane-only index math)
#define P2P_OPAQUE(v) asm volatile("" : "+v"(v))
#endif
#ifndef P2P_OPAQUE_S               // the same for a wave-uniform value in SGPRs (keeps an address computation on the scalar unit)
#define P2P_OPAQUE_S(v) asm volatile("" : "+s"(v))
#endif
#ifndef P2P_OPAQUE_V4              // the same for a 16-byte vector value (four VGPRs)
#define P2P_OPAQUE_V4(v) asm volatile("" : "+v"(v))
#endif
#ifndef P2P_SWAP_ADJACENT          // exchange a 32-bit value with the neighbouring lane (lane ^ 1): one DPP move, no LDS crossbar
#define P2P_SWAP_ADJACENT(v) ((unsigned)__builtin_amdgcn_mov_dpp((int)(v), 0xB1, 0xF, 0xF, true))      /* quad_perm [1,0,3,2] */
#endif
#ifndef P2P_SWAP_PAIRS             // the same with the lane two further (lane ^ 2)
#define P2P_SWAP_PAIRS(v) ((unsigned)__builtin_amdgcn_mov_dpp((int)(v), 0x4E, 0xF, 0xF, true))         /* quad_perm [2,3,0,1] */
#endif
#ifndef P2P_LANE_ID                // lane index inside the wave, recomputed from the hardware (v_mbcnt) instead of kept in a register
#define P2P_LANE_ID() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#endif
#ifndef P2P_DYN_SHARED             // the dynamic LDS allocation of a kernel, 16-byte aligned
#define P2P_DYN_SHARED(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif

#ifndef P2P_GLOBAL_LOAD_LDS16      // LDS-DMA: 16 bytes per lane, global `gptr + imm` (per lane) -> LDS `lptr + imm + 16 * lane` (lptr wave-uniform)
#define P2P_GLOBAL_LOAD_LDS16(gptr, lptr, imm)                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),           \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, (imm), 0)
#endif
#ifndef P2P_WAIT_VMCNT             // counted wait for this wave's vector-memory operations (LDS-DMA pieces included)
#define P2P_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif

#ifndef P2P_WAVE_SYNC              // hand-over of LDS data between the lanes of ONE wave (LDS operations of a wave execute in order)
#define P2P_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif

// "Done once per device" flags of the launchers (hipFuncSetAttribute for a kernel's dynamic LDS size, the compute-unit count): the
// work behind them is idempotent, so two host threads that meet on a device's first launch may both do it; the flag itself is an
// atomic with release / acquire ordering, so a thread that sees it set also sees what was stored before it (include/p2p_hip.h,
// threading note).
struct DeviceOnce {
    std::atomic<bool> flag[64];
    bool done(int dev) const { return dev >= 0 && dev < 64 && flag[dev].load(std::memory_order_acquire); }
    void set(int dev) { if (dev >= 0 && dev < 64) flag[dev].store(true, std::memory_order_release); }
};

// Check the launch that was just issued (asynchronous errors surface at the next sync).
static inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return P2P_EHIP;
    }
    return P2P_OK;
}

}  // namespace p2p

// opaque handles -------------------------------------------------------------------------------
struct p2p_ncn {
    float b2;                // scalar bias of layer 2 (same for both branches)
    unsigned char *wfused;   // both layers, both branches as fp16x2 MFMA fragments (consensus.hip), one device allocation
    int tile[3];             // forced (ta, tb, tc) of the fused kernel, 0 = automatic (p2p_ncn_set_tile: tests and sweeps)
};

struct p2p_regressor {
    int device;        // the device the handle was created on: every later allocation (another mode's weight stream) goes there
    float *dev;        // BatchNorm folds + FC layers (every mode)
    float *dev_p, *dev_h, *dev_w;   // the convolution weights in the stream order of the f32 / fp16x2 / fp16x2w kernels; packed on
                                    // the first selection of that mode (p2p_regressor_set_mode), null until then
    std::vector<float> conv1_w, conv2_w, bn1s_host, bn2s_host;   // host copies the packings are built from
    const float *wp1;  // f32: conv1 weights, MFMA-fragment order [8 waves][585 chunks][2][64 lanes][4]
    const float *wp2;  //      conv2 weights,                    [8][576][2][64][4]
    const float *wh1, *wh2;     // fp16x2: the same weights, scaled per output channel, split into two fp16 planes
    const float *bn1s_h, *bn2s_h;   // fp16x2: folded BN scales times the inverse of those weight (and activation) scales
    const float *ww2, *bn2s_w;      // fp16x2w: conv2 as Winograd-transformed filter blocks (regress_wino.hip) + its BN scale
    const float *wh1_w, *bn1s_w;    // fp16x2w: conv1's fp16x2 stream and folded BN1 scale inside this mode's allocation (dev_w)
    int mode;                   // P2P_REGRESS_F32 | P2P_REGRESS_FP16X2 | P2P_REGRESS_FP16X2W
    const float *bn1s, *bn1b;   // folded BN scale/shift [512]
    const float *bn2s, *bn2b;   // [512]
    const float *fc1t, *fc1b, *bnf1s, *bnf1b;   // fc1 as [128][512][4]; [512]
    const float *fc2t, *fc2b, *bnf2s, *bnf2b;   // fc2 as [128][256][4]; [256]
    const float *fc3, *fc3b;                    // [5][256]; [5]
    const float *fc1p, *fc2p;                   // fc1 / fc2 as B fragments of v_mfma_f32_16x16x4_f32: [k/16][n/16][lane 64][4]
};
