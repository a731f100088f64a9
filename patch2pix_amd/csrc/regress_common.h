// Pieces shared by the two fine-stage kernels (exact-f32 MFMA and split-bf16 MFMA).
#pragma once
#include "p2p_common.h"
#include <cstring>

namespace p2p {

constexpr int NT = 512;                 // threads per workgroup (8 waves)
constexpr int MAXB = 16;                // image pairs per launch

struct RegDev {
    const float *wp1, *wp2;             // f32 MFMA-fragment order (regress.hip)
    const float *ws1, *ws2;             // split-bf16 fragment order (regress_split.hip), viewed as 16-byte units
    const float *wx1, *wx2;             // three-plane bf16 fragment order (regress_x3.hip)
    const float *wh1, *wh2;             // two-plane fp16 fragment order, per-channel power-of-two scales (regress_h2.hip)
    const float *bn1s, *bn1b, *bn2s, *bn2b;
    const float *bn1s_h, *bn2s_h;       // BN scales with the fp16 operand scales of regress_h2.hip folded in
    const float *fc1t, *fc1b, *bnf1s, *bnf1b, *fc2t, *fc2b, *bnf2s, *bnf2b, *fc3, *fc3b;
};

struct ItemDev {
    const float *pyr[2][4];
    int H[2], W[2];
};

struct RegressArgs {
    ItemDev item[MAXB];
    int start[MAXB + 1];          // proposal range of each item in the concatenated arrays
    const int *dev_counts;        // optional: per item, how many of its slots hold a proposal (device memory; the
                                  // host then only knows the capacity start[b+1] - start[b])
    int nitems;
    const void *proposals;
    int is_float, n, nlevels;
    RegDev reg[2];
    float *matches[2], *probs[2], *raw[2];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// extent of pyramid level j for an image extent `dim`: the backbone's strided convolutions / pooling round UP
// (networks/resnet.py: 7x7 s2 p3, 3x3 s2 p1), while the gather clamps indices to dim // 2^j - 1 (networks/utils.py:22-23)
__device__ __forceinline__ int level_dim(int dim, int j) { return (dim + (1 << j) - 1) >> j; }

// cell index of patch row/col `p` (0..15) at pyramid level j, relative to the staged tile's origin:
// clamp(floor((origin+p)/2^j), 0, dim/2^j - 1) - clamp(floor(origin/2^j), ...)   (networks/utils.py:22-23)
__device__ __forceinline__ int patch_cell(int origin, int p, int j, int dim) {
    const int d = dim >> j;
    return clampi((origin + p) >> j, 0, d - 1) - clampi(origin >> j, 0, d - 1);
}

// FC 512->512->256->5 with folded BatchNorm1d + ReLU (networks/modules.py:89-99), then
// parse_regressor_out (networks/patch2pix.py:138-155; psize 16, ptype 'center').
// V: pooled conv features [512] in LDS; F1/F2 scratch [512]/[256]; misc[0..4] raw outputs,
// misc[8..11] current proposal (in: base of the offsets, out: regressed match).
__device__ __forceinline__ void fc_tail_parse(const RegDev &R, const ItemDev &I, const RegressArgs &args, int lvl, int prop,
                                              int tid, const float *V, float *F1, float *F2, float *misc) {
    P2P_OPAQUE(tid);      // keep the per-lane weight offsets from being hoisted out of the level loop (and spilled)
    {   // 512 outputs x 512 inputs: 32 weight loads (16 B each) in flight per thread
        const f32x4 *w = (const f32x4 *)R.fc1t + tid;
        float s = 0.f;
#pragma unroll 1
        for (int k0 = 0; k0 < 128; k0 += 32) {
            f32x4 wv[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) wv[i] = w[(k0 + i) * 512];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x4 x = *(const f32x4 *)(V + 4 * (k0 + i));
                s = fmaf(wv[i][0], x[0], s);
                s = fmaf(wv[i][1], x[1], s);
                s = fmaf(wv[i][2], x[2], s);
                s = fmaf(wv[i][3], x[3], s);
            }
        }
        s += R.fc1b[tid];
        F1[tid] = fmaxf(fmaf(s, R.bnf1s[tid], R.bnf1b[tid]), 0.f);
    }
    __syncthreads();
    if (tid < 256) {
        const f32x4 *w = (const f32x4 *)R.fc2t + tid;
        float s = 0.f;
#pragma unroll 1
        for (int k0 = 0; k0 < 128; k0 += 32) {
            f32x4 wv[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) wv[i] = w[(k0 + i) * 256];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x4 x = *(const f32x4 *)(F1 + 4 * (k0 + i));
                s = fmaf(wv[i][0], x[0], s);
                s = fmaf(wv[i][1], x[1], s);
                s = fmaf(wv[i][2], x[2], s);
                s = fmaf(wv[i][3], x[3], s);
            }
        }
        s += R.fc2b[tid];
        F2[tid] = fmaxf(fmaf(s, R.bnf2s[tid], R.bnf2b[tid]), 0.f);
    }
    __syncthreads();
    if (tid < 5) {
        const f32x4 *w = (const f32x4 *)(R.fc3 + tid * 256);
        float s = 0.f;
#pragma unroll 1
        for (int k0 = 0; k0 < 64; k0 += 16) {
            f32x4 wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = w[k0 + i];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 x = *(const f32x4 *)(F2 + 4 * (k0 + i));
                s = fmaf(wv[i][0], x[0], s);
                s = fmaf(wv[i][1], x[1], s);
                s = fmaf(wv[i][2], x[2], s);
                s = fmaf(wv[i][3], x[3], s);
            }
        }
        s += R.fc3b[tid];
        misc[tid] = s;
        if (args.raw[lvl]) args.raw[lvl][(size_t)prop * 5 + tid] = s;
        if (tid < 4) {
            const float off = 16.0f * tanhf(fmaxf(s, 0.f)) - 8.0f;
            float fm = misc[8 + tid] + off;
            const float hi = (float)((tid == 0) ? I.W[0] : (tid == 1) ? I.H[0] : (tid == 2) ? I.W[1] : I.H[1]);
            fm = fminf(fmaxf(fm, 0.f), hi);
            if (args.matches[lvl]) args.matches[lvl][(size_t)prop * 4 + tid] = fm;
            misc[8 + tid] = fm;       // becomes the next level's proposal (un-truncated)
        } else {
            if (args.probs[lvl]) args.probs[lvl][prop] = 1.0f / (1.0f + expf(-s));
        }
    }
    __syncthreads();
}

// regress_split.hip
constexpr int S1_SLABS = 4 + 9 * 2 * 16; // conv1: 4 slabs of level 0 (3 ch x 9 taps x 2 images, padded 54 -> 64), then per
                                        // (tap, image) 4 + 4 + 8 slabs of 16 channels of levels 1, 2, 3
constexpr int S2_SLABS = 9 * 32;        // conv2: 512 channels / 16 per tap
constexpr int SPF = 7;                  // weight prefetch distance in units = ring of 8 register buffers
constexpr int S1_UNITS = 2 * S1_SLABS;   // conv1 streams one n-tile at a time: unit = (slab, n-tile), 2 KiB per (wave, unit)
constexpr size_t WS1_FLOATS = (size_t)8 * (S1_UNITS + SPF) * 512;
constexpr int S2_UNITS = 2 * S2_SLABS;   // conv2 likewise: [tap][n-tile][32 slabs]
constexpr size_t WS2_FLOATS = (size_t)8 * (S2_UNITS + SPF) * 512;
void pack_split_weights(const float *conv1_w, const float *conv2_w, float *ws1, float *ws2);   // host
int launch_regress_split(const RegressArgs &a, int n, hipStream_t stream);
void split_conv1_index(int slab, int half, int j, int &ch, int &tap);   // K layout of conv1 shared by the bf16 kernels

// regress_x3.hip: unit = (slab of 16 K, n-tile), three bf16 planes = 3 KiB per (wave, unit); stream order [slab][n-tile]
constexpr int XPF = 8;                   // units the weight prefetch may run past the end of a stream
constexpr size_t WX1_FLOATS = (size_t)8 * (S1_UNITS + XPF) * 768;
constexpr size_t WX2_FLOATS = (size_t)8 * (S2_UNITS + XPF) * 768;
// t1 / t2 [512]: per-output-channel exponents the weights were scaled by (all zero for the bf16 planes)
void pack_x3_weights(const float *conv1_w, const float *conv2_w, float *wx1, float *wx2, int *t1, int *t2);      // host
int launch_regress_x3(const RegressArgs &a, int n, hipStream_t stream);

// regress_h2.hip: the same streams with two fp16 planes = 2 KiB per (wave, unit)
constexpr size_t WH1_FLOATS = (size_t)8 * (S1_UNITS + XPF) * 512;
constexpr size_t WH2_FLOATS = (size_t)8 * (S2_UNITS + XPF) * 512;
void pack_h2_weights(const float *conv1_w, const float *conv2_w, float *wh1, float *wh2, int *t1, int *t2);      // host
int launch_regress_h2(const RegressArgs &a, int n, hipStream_t stream);

// host-side bf16 helpers (round to nearest even)
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace p2p
