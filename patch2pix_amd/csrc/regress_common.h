// Pieces shared by the fine-stage kernels (exact-f32 MFMA, regress.hip; fp16x2, regress_h2.hip / regress_wino.hip).
#pragma once
#include "p2p_common.h"
#include <algorithm>
#include <cstring>

namespace p2p {

constexpr int NT = 512;                 // threads per workgroup (8 waves)
constexpr int MAXB = 16;                // image pairs per launch

struct RegDev {
    const float *wp1, *wp2;             // f32 MFMA-fragment order (regress.hip)
    const float *wh1, *wh2;             // two-plane fp16 fragment order, per-channel power-of-two scales (regress_h2.hip)
    const float *bn1s, *bn1b, *bn2s, *bn2b;
    const float *bn1s_h, *bn2s_h;       // BN scales with the fp16 operand scales of regress_h2.hip folded in
    const float *fc1t, *fc1b, *bnf1s, *bnf1b, *fc2t, *fc2b, *bnf2s, *bnf2b, *fc3, *fc3b;
    const float *fc1p, *fc2p;           // fc1 / fc2 in v_mfma_f32_16x16x4_f32 fragment order (fc_batch_parse)
    const float *ww2, *bn2s_w;          // conv2 as Winograd-transformed filter blocks + its BN scale (regress_wino.hip)
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
    float *ws;                    // regress_ws_floats(n) floats of scratch (kernels with the batched FC tail; else unused)
    // P2P_REGRESS_FP16X2W only (regress_h2_kernel<true>): the level and proposal range of this launch, the transformed
    // conv2 input it writes, the per-proposal inverse scales, the row blocks of the range
    unsigned char *wU;
    float *hinv;
    int lvl0, p0, p1, mblocks;
};

// scratch of the kernels whose FC tail is batched over a work-group's proposals (regress_h2.hip): the pooled
// convolution features V [level][n][512] and the un-truncated mid matches [n][4] the fine level starts from
constexpr int FC_ROWS = 16;             // proposals per FC batch = rows of a v_mfma_f32_16x16x4_f32 tile
// P2P_REGRESS_FP16X2W appends: the inverse H scale per proposal of a chunk, and the transformed conv2 input of a chunk of at
// most WINO_CHUNK (3328) proposals (16 positions x 16 tiles x 512 channels x 2 fp16 planes = 512 KiB per proposal), as the A
// blocks of wino_gemm_kernel: [position 16][row block of 8 proposals][K chunk 16][WINO_BLK bytes]
constexpr int WINO_BLK = 16384;         // [plane 2][row 128][32 K] fp16
#ifndef P2P_WINO_CHUNK
#define P2P_WINO_CHUNK 3328
#endif
// Proposals per conv1 -> GEMM round.  A call's n proposals are cut into ceil(units / (WINO_CHUNK / 256)) chunks of whole
// units of 256 proposals (= whole rounds of the persistent conv1 launch on 256 compute units, two rounds of the GEMM's
// 128-row x 128-column work-groups), sizes as even as the units allow: 6400 -> 3328 + 3072, 12800 -> 3328 + 3328 + 3072 + 3072
// (round 5's fixed chunk of 2048 left a last launch of 256 proposals = a half-empty GEMM round and a full set of fixed
// per-launch costs; caps measured in round 6, same box: 2048 13.51, 2560 13.50, 3328 13.40, 6656 13.43 ms per 6400 x 2 call).
constexpr int WINO_CHUNK = P2P_WINO_CHUNK;
static_assert(WINO_CHUNK % 256 == 0 && WINO_CHUNK >= 256, "chunks are whole units of 256 proposals");
static inline int wino_nchunks(size_t n) {
    const size_t units = (n + 255) / 256, per = WINO_CHUNK / 256;
    return (int)std::max<size_t>(1, (units + per - 1) / per);
}
// proposals [*p0, *p1) of chunk c (0 <= c < wino_nchunks(n))
static inline void wino_chunk_range(size_t n, int c, int *p0, int *p1) {
    const size_t units = (n + 255) / 256, nch = (size_t)wino_nchunks(n), base = units / nch, rem = units % nch;
    const size_t u0 = (size_t)c * base + std::min<size_t>((size_t)c, rem), u1 = u0 + base + ((size_t)c < rem ? 1 : 0);
    *p0 = (int)std::min(n, u0 * 256);
    *p1 = (int)std::min(n, u1 * 256);
}
static inline size_t regress_ws_base_floats(size_t n) { return ((2 * n * 512 + 31) & ~size_t(31)) + 4 * n + 32; }
static inline size_t wino_hinv_offset_floats(size_t n) { return (regress_ws_base_floats(n) + 63) & ~size_t(63); }
static inline size_t wino_chunk_rows(size_t n) {        // rows of the transformed-input buffer: no chunk is larger
    return (std::min(n, (size_t)WINO_CHUNK) + 7) & ~size_t(7);
}
static inline size_t wino_u_offset_floats(size_t n) { return (wino_hinv_offset_floats(n) + wino_chunk_rows(n) + 63) & ~size_t(63); }
static inline size_t regress_ws_floats(size_t n) {
    return wino_u_offset_floats(n) + (size_t)16 * (wino_chunk_rows(n) / 8) * 16 * (WINO_BLK / 4);
}

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

// P2P_REGRESS_FP16X2W walks the proposals that EXIST: compact index c (0, 1, ... in item order; with device-side counts the
// first dev_counts[i] slots of every item) -> slot in the concatenated arrays, or -1 past the last one.  Its scratch rows
// (the transformed conv2 input, the inverse scales) are indexed by c, so that empty slots cost no GEMM rows.
template <class A>
__device__ __forceinline__ int wino_slot(const A &a, int c) {
    if (!a.dev_counts) return c < a.n ? c : -1;
    int cum = 0;
    for (int it = 0; it < a.nitems; ++it) {
        const int cnt = max(0, min(a.dev_counts[it], a.start[it + 1] - a.start[it]));     // (-1 = "take the host path": no proposals)
        if (c < cum + cnt) return a.start[it] + (c - cum);
        cum += cnt;
    }
    return -1;
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

// ---- FC tail batched over the proposals of a (persistent) work-group ------------------------------------------------
// A load that must see what another wave of this work-group stored to global memory earlier in the same launch (the vector
// L1 is not guaranteed to reflect it): relaxed atomic load = cache-bypassing.
__device__ __forceinline__ float load_coherent(const float *p) {
    return __int_as_float(__atomic_load_n((const int *)p, __ATOMIC_RELAXED));
}
#define P2P_MFMA_F32_16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int FCST1 = 512 + 4, FCST2 = 256 + 4;          // LDS row strides (floats): rows 16 bytes apart modulo the 256-byte bank row
constexpr int FC_LDS_BYTES = FC_ROWS * (2 * FCST1 + FCST2) * 4;

// FC 512->512->256->5 with folded BatchNorm1d + ReLU (networks/modules.py:89-99) and parse_regressor_out
// (networks/patch2pix.py:138-155; psize 16, ptype 'center') for ALL proposals of this work-group at one level: proposal
// r of the group is first + r * gridDim.x, its pooled convolution features V[512] wait in wsV (global scratch, written by
// this work-group).  FC_ROWS proposals at a time are the 16 rows of v_mfma_f32_16x16x4_f32 tiles (exact fp32 products and
// accumulation): the 1.5 MB of fc1 / fc2 weights are streamed once per 16 proposals instead of once per proposal (the
// per-proposal tail took 5 % of the launch, all of it weight ingest), and a launch of a few hundred proposals -- one image
// pair at evaluation time -- no longer pays three serial tails per compute unit.
// K order inside a tile row: step (S, j) multiplies k = 16 S + 4 (lane >> 4) + j, so a lane's four A values of a super-step
// S are one aligned 16-byte LDS read and its four B values one 16-byte global load (weights packed to match, pack_fc_mfma).
// nextp: un-truncated matches of this level = the proposals of the next one (written when there is a next level).
__device__ __forceinline__ void fc_batch_parse(const RegDev &R, const RegressArgs &args, int lvl, const float *wsV,
                                               float *nextp, unsigned char *smb, int tid) {
    P2P_OPAQUE(tid);      // keep the per-lane offsets inside the level loop (nothing lane-dependent is hoisted and spilled)
    const int nwg = gridDim.x, first = blockIdx.x;
    const int cnt = (args.n - first + nwg - 1) / nwg;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kb = lane >> 4;
    float *Vs = (float *)smb, *F1s = Vs + FC_ROWS * FCST1, *F2s = F1s + FC_ROWS * FCST1;
#pragma unroll 1
    for (int r0 = 0; r0 < cnt; r0 += FC_ROWS) {
        {   // the rows' features: 16 x 512 floats, a thread moves 4 x 16 bytes of one row
            const int row = tid >> 5, c4 = tid & 31;
            const bool ok = r0 + row < cnt;
            const float *src = wsV + (size_t)(first + (r0 + row) * nwg) * 512;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 4 * (c4 + 32 * q);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = (f32x4){load_coherent(src + col), load_coherent(src + col + 1), load_coherent(src + col + 2), load_coherent(src + col + 3)};
                *(f32x4 *)(Vs + row * FCST1 + col) = v;
            }
        }
        __syncthreads();
        {   // fc1: this wave's 64 output channels = four 16-column tiles
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *w = R.fc1p + ((size_t)(wave * 4) * 64 + lane) * 4;       // [S 32][tile 32][lane 64][4]
            const float *a = Vs + l15 * FCST1 + 4 * kb;
#pragma unroll 2
            for (int S = 0; S < 32; ++S) {
                const f32x4 av = *(const f32x4 *)(a + 16 * S);
                f32x4 bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = *(const f32x4 *)(w + ((size_t)S * 32 + t) * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = P2P_MFMA_F32_16(av[j], bv[t][j], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = 16 * (4 * wave + t) + l15;
                const float b = R.fc1b[n], s = R.bnf1s[n], sh = R.bnf1b[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) F1s[(4 * kb + r) * FCST1 + n] = fmaxf(fmaf(acc[t][r] + b, s, sh), 0.f);
            }
        }
        __syncthreads();
        {   // fc2: 32 output channels per wave
            f32x4 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *w = R.fc2p + ((size_t)(wave * 2) * 64 + lane) * 4;       // [S 32][tile 16][lane 64][4]
            const float *a = F1s + l15 * FCST1 + 4 * kb;
#pragma unroll 2
            for (int S = 0; S < 32; ++S) {
                const f32x4 av = *(const f32x4 *)(a + 16 * S);
                f32x4 bv[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) bv[t] = *(const f32x4 *)(w + ((size_t)S * 16 + t) * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = P2P_MFMA_F32_16(av[j], bv[t][j], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = 16 * (2 * wave + t) + l15;
                const float b = R.fc2b[n], s = R.bnf2s[n], sh = R.bnf2b[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) F2s[(4 * kb + r) * FCST2 + n] = fmaxf(fmaf(acc[t][r] + b, s, sh), 0.f);
            }
        }
        __syncthreads();
        // fc3 (one fp32 fma chain in k order per output) + parse: wave w takes rows 2w and 2w + 1 (proposal and item are
        // wave-uniform: scalar reads of the launch arguments), lane o < 5 = output o
#pragma unroll 1
        for (int rr = 0; rr < FC_ROWS / 8; ++rr) {
            const int row = (FC_ROWS / 8) * wave + rr;
            if (r0 + row >= cnt) break;
            const int prop = first + (r0 + row) * nwg;
            int it = 0;
            while (it + 1 < args.nitems && prop >= args.start[it + 1]) ++it;
            if (args.dev_counts && prop - args.start[it] >= args.dev_counts[it]) continue;      // empty slot: outputs untouched
            const ItemDev &I = args.item[it];
            if (lane < 5) {
                const int o = lane;
                const f32x4 *w = (const f32x4 *)(R.fc3 + o * 256);
                const float *x = F2s + row * FCST2;
                float s = 0.f;
#pragma unroll 8
                for (int k = 0; k < 64; ++k) {
                    const f32x4 wv = w[k], xv = *(const f32x4 *)(x + 4 * k);
                    s = fmaf(wv[0], xv[0], s);
                    s = fmaf(wv[1], xv[1], s);
                    s = fmaf(wv[2], xv[2], s);
                    s = fmaf(wv[3], xv[3], s);
                }
                s += R.fc3b[o];
                if (args.raw[lvl]) args.raw[lvl][(size_t)prop * 5 + o] = s;
                if (o < 4) {
                    float base;       // the proposal the offsets are relative to, un-truncated (patch2pix.py:145)
                    if (lvl > 0) base = load_coherent(nextp + (size_t)prop * 4 + o);
                    else if (args.is_float) base = ((const float *)args.proposals)[(size_t)prop * 4 + o];
                    else base = (float)((const long long *)args.proposals)[(size_t)prop * 4 + o];
                    const float off = 16.0f * tanhf(fmaxf(s, 0.f)) - 8.0f;
                    const float hi = (float)((o == 0) ? I.W[0] : (o == 1) ? I.H[0] : (o == 2) ? I.W[1] : I.H[1]);
                    const float fm = fminf(fmaxf(base + off, 0.f), hi);
                    if (args.matches[lvl]) args.matches[lvl][(size_t)prop * 4 + o] = fm;
                    if (lvl + 1 < args.nlevels) nextp[(size_t)prop * 4 + o] = fm;
                } else {
                    if (args.probs[lvl]) args.probs[lvl][prop] = 1.0f / (1.0f + expf(-s));
                }
            }
        }
        __syncthreads();
    }
}

// fc weight [N][512] (row-major, torch Linear) -> B fragments of v_mfma_f32_16x16x4_f32 in the K order of fc_batch_parse:
// out[((S * (N / 16) + tile) * 64 + lane) * 4 + j] = W[16 tile + (lane & 15)][16 S + 4 (lane >> 4) + j]
static inline void pack_fc_mfma(const float *w, int N, float *out) {
    for (int S = 0; S < 32; ++S)
        for (int t = 0; t < N / 16; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    out[(((size_t)S * (N / 16) + t) * 64 + lane) * 4 + j] = w[(size_t)(16 * t + (lane & 15)) * 512 + 16 * S + 4 * (lane >> 4) + j];
}

// regress_h2.hip: the K axis of the two convolutions in slabs of 16
constexpr int S1_SLABS = 4 + 9 * 2 * 16; // conv1: 4 slabs of level 0 (3 ch x 9 taps x 2 images, padded 54 -> 64), then per
                                        // (tap, image) 4 + 4 + 8 slabs of 16 channels of levels 1, 2, 3
constexpr int S2_SLABS = 9 * 32;        // conv2: 512 channels / 16 per tap
constexpr int S1_UNITS = 2 * S1_SLABS;   // conv1 streams one n-tile at a time: unit = (slab, n-tile), 2 KiB per (wave, unit)
constexpr int S2_UNITS = 2 * S2_SLABS;   // conv2 likewise: [tap][n-tile][32 slabs]

// regress_h2.hip: unit = (slab of 16 K, n-tile), two fp16 planes = 2 KiB per (wave, unit); stream order [slab][n-tile]
constexpr int XPF = 8;                   // units the weight prefetch may run past the end of a stream
// t1 / t2 [512]: per-output-channel exponents the weights were scaled by
constexpr size_t WH1_FLOATS = (size_t)8 * (S1_UNITS + XPF) * 512;
constexpr size_t WH2_FLOATS = (size_t)8 * (S2_UNITS + XPF) * 512;
void pack_h2_weights(const float *conv1_w, const float *conv2_w, float *wh1, float *wh2, int *t1, int *t2);      // host
int launch_regress_h2(const RegressArgs &a, int n, hipStream_t stream);
int launch_regress_h2_conv1(const RegressArgs &a, int n, hipStream_t stream);   // conv1 -> transformed conv2 input (FP16X2W)

// regress_wino.hip: conv2 as Winograd F(2x2, 3x3) GEMMs; filter blocks [position 16][column block 4][K chunk 16][WINO_BLK]
constexpr size_t WW2_FLOATS = (size_t)16 * 4 * 16 * (WINO_BLK / 4);
void pack_wino_weights(const float *conv2_w, float *ww2, int *t2);                // host
int launch_regress_wino(RegressArgs a, int n, hipStream_t stream);

}  // namespace p2p
