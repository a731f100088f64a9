// Fine stage, fp32-equivalent arithmetic on the bf16 matrix cores (mode P2P_REGRESS_BF16X3, the default).
//
// Same algorithm and decomposition as regress.hip / regress_split.hip (one workgroup per proposal, 8 waves x 64
// output channels, patch deduplicated in LDS, weights streamed from L2 in consumption order).  Every fp32
// operand x of the two convolutions is represented EXACTLY as the sum of three bf16 numbers
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (3 x 8 = 24 significant bits)
// and a product is the six v_mfma_f32_32x32x16_bf16 whose order in 2^-8 is <= 2:
//     a*b ~= a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0),      fp32 accumulation.
// The dropped terms (a1*b2 + a2*b1 + a2*b2) are <= 2^-23 |a*b|, i.e. at the level of the rounding of one fp32
// product; measured against an fp64 evaluation this mode is as accurate as the exact-f32 MFMA kernel
// (tools/split_check.py, tests/test_gpu_parity.py).  Cost: 6 bf16 MFMA passes per unit of K instead of 32 fp32
// ones -> 2.65x the fp32-MFMA ceiling.
//
// What differs from regress_split.hip (2 planes, 3 products):
//   * LDS holds fp32 (de-duplicated tiles [cell][C], H[pixel][512]) -- three bf16 planes of H would need 203 KB.
//     The A fragments are read as fp32 (two ds_read_b128 per row and slab of 16 K), multiplied by the per-pixel
//     L2 scale (conv1; zero scale = zero padding) and split into the three planes in registers (40 VALU
//     operations per 8 values, v_cvt_pk_bf16_f32 based), once per slab for both n-tiles;
//   * because the scale is applied before the split there is no second accumulator set and no fold step;
//   * weights are pre-split at load time: unit = (slab, n-tile) = 3 planes x 1 KiB per wave, stream order
//     [slab][n-tile], prefetched two units ahead through a ring of four register buffers.
#include "regress_common.h"

#include <vector>

namespace p2p {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- LDS layout (bytes) --------------------------------------------------------------------------
// conv1 phase: levels 1-3 de-duplicated tiles [img][cell][C fp32 (+16 B pad)], level 0 raw [img][3][256] and the
// pre-scaled level-0 im2col block A0[64 px][64 K fp32 (+16 B pad)], K = img*32 + tap*3 + c (27 real per image).
// conv2 phase: H[65 px][512 fp32 (+16 B pad)] (row 64 = zeros = padding) overlays all of the above.
constexpr int XST1 = 64 * 4 + 16, XST3 = 128 * 4 + 16;           // bytes per cell of levels 1/2 and 3
constexpr int XNC1 = 81, XNC2 = 25, XNC3 = 9;
constexpr int XOFF1 = 0;
constexpr int XOFF2 = XOFF1 + XNC1 * XST1;
constexpr int XOFF3 = XOFF2 + XNC2 * XST1;
constexpr int XIMG = XOFF3 + XNC3 * XST3;                         // 33584
constexpr int XRAW0 = 2 * XIMG;                                   // float [2][3][256]
constexpr int XA0 = XRAW0 + 2 * 3 * 256 * 4;
constexpr int XA0ST = 64 * 4 + 16;
constexpr int XCONV1B = XA0 + 64 * XA0ST;                         // 90720
constexpr int XHPIX = 512 * 4 + 16;
constexpr int XUNION = 65 * XHPIX;                                // 134160
constexpr int XSM_SCALE = XUNION;                                 // float [2][256]
constexpr int XSM_V = XSM_SCALE + 512 * 4;
constexpr int XSM_F1 = XSM_V + 512 * 4;
constexpr int XSM_F2 = XSM_F1 + 512 * 4;
constexpr int XSM_MISC = XSM_F2 + 256 * 4;
constexpr int XSM_BYTES = XSM_MISC + 16 * 4;
static_assert(XCONV1B <= XUNION, "conv1 buffers must fit under H");
static_assert(XIMG % 16 == 0 && XA0 % 16 == 0 && XUNION % 16 == 0, "16-byte alignment of ds_read_b128");
static_assert(XSM_BYTES <= 160 * 1024, "LDS budget");

// two fp32 -> one dword of two bf16 (round to nearest even): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// 8 consecutive K values of one row (two 16-byte LDS reads), scaled by s, as three bf16x8 planes
__device__ __forceinline__ void split3(const f32x4 &xa, const f32x4 &xb, float s, f32x4 &p0, f32x4 &p1, f32x4 &p2) {
#ifdef XP_NOSPLIT                       // timing experiment (wrong results): how much of the split is hidden
    p0 = xa; p1 = xb; p2 = xa; return;
#endif
    const float x[8] = {xa[0] * s, xa[1] * s, xa[2] * s, xa[3] * s, xb[0] * s, xb[1] * s, xb[2] * s, xb[3] * s};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h = pk_bf16(x[2 * q], x[2 * q + 1]);
        const float r0 = x[2 * q] - __uint_as_float(h << 16);
        const float r1 = x[2 * q + 1] - __uint_as_float(h & 0xffff0000u);
        const unsigned m = pk_bf16(r0, r1);
        const float t0 = r0 - __uint_as_float(m << 16);
        const float t1 = r1 - __uint_as_float(m & 0xffff0000u);
        p0[q] = __uint_as_float(h);
        p1[q] = __uint_as_float(m);
        p2[q] = __uint_as_float(pk_bf16(t0, t1));
    }
}

#define XMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// raw fp32 A fragment (8 K values of this lane's row) of one m-tile: two 16-byte LDS reads
#define XLOADR(R, P) R[0] = *(const f32x4 *)(P); R[1] = *(const f32x4 *)((P) + 16);
// weights of the unit `AHEAD` units after the current stream position: 3 planes
#define XLOADB(BUF, AHEAD)                                                               \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) BUF[q_] = bp[(AHEAD) * 192 + q_ * 64];
// 12 MFMAs of one m-tile (planes SP) against the weights of both n-tiles: 6 products each, smallest terms first;
// the two accumulators alternate
#define XHALF(CU0, CU1, SP, BU0, BU1)                                                    \
    CU0 = XMFMA(SP[2], BU0[0], CU0); CU1 = XMFMA(SP[2], BU1[0], CU1);                    \
    CU0 = XMFMA(SP[1], BU0[1], CU0); CU1 = XMFMA(SP[1], BU1[1], CU1);                    \
    CU0 = XMFMA(SP[0], BU0[2], CU0); CU1 = XMFMA(SP[0], BU1[2], CU1);                    \
    CU0 = XMFMA(SP[1], BU0[0], CU0); CU1 = XMFMA(SP[1], BU1[0], CU1);                    \
    CU0 = XMFMA(SP[0], BU0[1], CU0); CU1 = XMFMA(SP[0], BU1[1], CU1);                    \
    CU0 = XMFMA(SP[0], BU0[0], CU0); CU1 = XMFMA(SP[0], BU1[0], CU1);
// Software pipeline inside a wave.  The matrix pipe takes 32 cycles per MFMA and a wave issues in order, so a wave
// that first splits a whole slab (88 VALU operations) and then issues its 24 MFMAs leaves the pipe idle while it --
// and the other wave of the SIMD, which runs the same code in step -- does VALU work (first version: pipe 55 % busy).
// Here every group of 12 MFMAs (one m-tile) carries the split of the OTHER m-tile's next fragment in its shadow:
//   phase A:  MFMAs of m-tile 0 (planes S0) || LDS read of the next slab's m-tile-0 fragment, split of R1 -> S1,
//             weight loads of the next slab's first unit
//   phase B:  MFMAs of m-tile 1 (planes S1) || LDS read of the next slab's m-tile-1 fragment, split of R0 -> S0,
//             weight loads of the next slab's second unit
// sched_group_barrier pins the interleave (1 MFMA, then up to 4 VALU; the loads at the head of the phase).
// Weights: (BC0, BC1) = this slab's two units, (BN0, BN1) = the next slab's, loaded one slab ahead (>= 768 matrix-pipe
// cycles) into the buffers the previous slab used.
#ifndef XP_VALU
#define XP_VALU 4                       // VALU operations placed behind each MFMA (tools/ab_variants.sh)
#endif
#define XPIPE()                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);             \
    _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, XP_VALU, 0); }
#define XSLAB(N0, N1, SC0, SC1, BC0, BC1, BN0, BN1, AH)                                   \
    { XLOADR(R0, N0) XLOADB(BN0, AH) split3(R1[0], R1[1], (SC1), S1[0], S1[1], S1[2]);                               \
      XHALF(acc00, acc01, S0, BC0, BC1) XPIPE() __builtin_amdgcn_sched_barrier(0);                                    \
      XLOADR(R1, N1) XLOADB(BN1, AH + 1) split3(R0[0], R0[1], (SC0), S0[0], S0[1], S0[2]);                            \
      XHALF(acc10, acc11, S1, BC0, BC1) XPIPE() __builtin_amdgcn_sched_barrier(0); }
// Two consecutive slabs = four units.  On entry S0 holds the planes of the first slab's m-tile 0, R1 the raw
// fragment of its m-tile 1 (XPRO), and B0/B1 the weights of its two units.  (N0,N1) = A addresses of the second
// slab, (M0,M1) = of the slab after that.
#define XSLAB2(N0, N1, M0, M1, SC0, SC1)                                                  \
    { XSLAB(N0, N1, SC0, SC1, B0, B1, B2, B3, 2) XSLAB(M0, M1, SC0, SC1, B2, B3, B0, B1, 4) bp += 4 * 192; }
// start of a run of slabs: fragments of its first slab
#define XPRO(P0, P1, SC0) { XLOADR(R0, P0) XLOADR(R1, P1) split3(R0[0], R0[1], (SC0), S0[0], S0[1], S0[2]); }

__global__ __launch_bounds__(NT, 2) void regress_x3_kernel(RegressArgs args) {
    P2P_DYN_SHARED(unsigned char, smb);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int prop = blockIdx.x;
    int it = 0;
    while (it + 1 < args.nitems && prop >= args.start[it + 1]) ++it;
    if (args.dev_counts && prop - args.start[it] >= args.dev_counts[it]) return;      // empty slot (whole work-group)
    const ItemDev &I = args.item[it];

    float *raw0 = (float *)(smb + XRAW0);
    float *scale = (float *)(smb + XSM_SCALE);
    float *V = (float *)(smb + XSM_V);
    float *F1 = (float *)(smb + XSM_F1);
    float *F2 = (float *)(smb + XSM_F2);
    float *misc = (float *)(smb + XSM_MISC);

    if (tid < 4) {
        float v;
        if (args.is_float) v = ((const float *)args.proposals)[prop * 4 + tid];
        else v = (float)((const long long *)args.proposals)[prop * 4 + tid];
        misc[8 + tid] = v;
    }
    __syncthreads();

    for (int lvl = 0; lvl < args.nlevels; ++lvl) {
        const RegDev &R_ = args.reg[lvl];
        // window origins (x, y) in image 1 / image 2 (networks/utils.py:8-19); scalars + selects, never an indexed array
        const int xa = (int)misc[8 + 0] - 8, ya = (int)misc[8 + 1] - 8;
        const int xb = (int)misc[8 + 2] - 8, yb = (int)misc[8 + 3] - 8;
#define XX0(img_) ((img_) ? xb : xa)
#define XY0(img_) ((img_) ? yb : ya)
        __syncthreads();
        // opaque copy of the thread id for the staging phases (keeps their lane-only index math inside the level loop)
        int tidv = tid;
        P2P_OPAQUE(tidv);

        // ------------------------------------------------------------ gather (networks/utils.py:4-36)
        {
            // two passes so that all ~40 scattered 4-byte loads of a thread are in flight together
            float g0[2][2], g1[2][11], g2[2][4], g3[2][3];
#pragma unroll
            for (int img = 0; img < 2; ++img) {
                const int Hh = I.H[img], Ww = I.W[img];
                {
                    const int r0 = clampi(XY0(img), 0, Hh - 1), c0 = clampi(XX0(img), 0, Ww - 1);
                    const float *src = I.pyr[img][0];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int e = tidv + k * NT;
                        const int c = e >> 8, rem = e & 255, r = rem >> 4, cc = rem & 15;
                        g0[img][k] = (e < 768) ? src[((size_t)c * Hh + min(r0 + r, Hh - 1)) * Ww + min(c0 + cc, Ww - 1)] : 0.f;
                    }
                }
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                    const int Cc = (j == 3) ? 128 : 64;
                    const int nk = (j == 1) ? 11 : (j == 2) ? 4 : 3;
                    const int Hj = Hh >> j, Wj = Ww >> j;                     // index clamp: dim // ds (networks/utils.py:22-23)
                    const int Ha = level_dim(Hh, j), Wa = level_dim(Ww, j);  // extent of the backbone's map
                    const int r0 = clampi(XY0(img) >> j, 0, Hj - 1);
                    const int c0 = clampi(XX0(img) >> j, 0, Wj - 1);
                    const float *src = I.pyr[img][j];
#pragma unroll
                    for (int k = 0; k < nk; ++k) {
                        const int e = tidv + k * NT;
                        const int c = e / (Rr * Rr);
                        const int rem = e - c * (Rr * Rr);
                        const int r = rem / Rr;
                        const int cc = rem - r * Rr;
                        const float v = (e < Cc * Rr * Rr)
                                            ? src[((size_t)c * Ha + min(r0 + r, Hj - 1)) * Wa + min(c0 + cc, Wj - 1)] : 0.f;
                        if (j == 1) g1[img][k] = v; else if (j == 2) g2[img][k] = v; else g3[img][k] = v;
                    }
                }
            }
#pragma unroll
            for (int img = 0; img < 2; ++img) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = tidv + k * NT;
                    if (e < 768) raw0[img * 768 + e] = g0[img][k];
                }
                unsigned char *tb = smb + img * XIMG;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                    const int Cc = (j == 3) ? 128 : 64;
                    const int nk = (j == 1) ? 11 : (j == 2) ? 4 : 3;
                    const int off = (j == 1) ? XOFF1 : (j == 2) ? XOFF2 : XOFF3;
                    const int st = (j == 3) ? XST3 : XST1;
#pragma unroll
                    for (int k = 0; k < nk; ++k) {
                        const int e = tidv + k * NT;
                        if (e < Cc * Rr * Rr) {
                            const int c = e / (Rr * Rr);
                            const int rem = e - c * (Rr * Rr);
                            *(float *)(tb + off + rem * st + c * 4) = (j == 1) ? g1[img][k] : (j == 2) ? g2[img][k] : g3[img][k];
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ per-pixel L2 scale (patch2pix.py:173-174)
        {
            const int img = tidv >> 8, pix = tidv & 255, py = pix >> 4, px = pix & 15;
            const unsigned char *tb = smb + img * XIMG;
            float ss = 0.f;
            {
                const float *p = raw0 + img * 768 + patch_cell(XY0(img), py, 0, I.H[img]) * 16 + patch_cell(XX0(img), px, 0, I.W[img]);
#pragma unroll
                for (int c = 0; c < 3; ++c) ss = fmaf(p[c * 256], p[c * 256], ss);
            }
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                const int Cc = (j == 3) ? 128 : 64;
                const int off = (j == 1) ? XOFF1 : (j == 2) ? XOFF2 : XOFF3;
                const int st = (j == 3) ? XST3 : XST1;
                const unsigned char *p = tb + off + (patch_cell(XY0(img), py, j, I.H[img]) * Rr + patch_cell(XX0(img), px, j, I.W[img])) * st;
                for (int c = 0; c < Cc; c += 4) {
                    const f32x4 v = *(const f32x4 *)(p + c * 4);
                    ss = fmaf(v[0], v[0], ss); ss = fmaf(v[1], v[1], ss); ss = fmaf(v[2], v[2], ss); ss = fmaf(v[3], v[3], ss);
                }
            }
            scale[tid] = 1.0f / sqrtf(ss + 1e-6f);
        }
        __syncthreads();

        // ------------------------------------------------------------ level-0 im2col block, pre-scaled
        for (int e = tidv; e < 64 * 64; e += NT) {
            const int m = e >> 6, kk = e & 63, img = kk >> 5, r = kk & 31;
            float v = 0.f;
            if (r < 27) {
                const int tap = r / 3, c = r - tap * 3, ky = tap / 3, kx = tap - ky * 3;
                const int py = 2 * (m >> 3) + ky - 1, px = 2 * (m & 7) + kx - 1;
                if (py >= 0 && px >= 0)
                    v = raw0[img * 768 + c * 256 + patch_cell(XY0(img), py, 0, I.H[img]) * 16 + patch_cell(XX0(img), px, 0, I.W[img])] *
                        scale[img * 256 + py * 16 + px];
            }
            *(float *)(smb + XA0 + m * XA0ST + kk * 4) = v;
        }
        __syncthreads();

        // ------------------------------------------------------------ conv1: 3x3, stride 2, pad 1
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        f32x4 B0[3], B1[3], B2[3], B3[3], R0[2], R1[2], S0[3], S1[3];
        {
            const f32x4 *bp = (const f32x4 *)R_.wx1 + (size_t)wave * (S1_UNITS + XPF) * 192 + lane;
            XLOADB(B0, 0) XLOADB(B1, 1)
            {   // level 0 of both images: 4 slabs of the pre-scaled block
                const unsigned char *p0 = smb + XA0 + l31 * XA0ST + half * 32;
                const unsigned char *p1 = p0 + 32 * XA0ST;
                XPRO(p0, p1, 1.0f)
                XSLAB2(p0 + 64, p1 + 64, p0 + 128, p1 + 128, 1.0f, 1.0f)
                XSLAB2(p0 + 192, p1 + 192, p0, p1, 1.0f, 1.0f)
            }
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                int pyc[2], pxc[2];
                bool ok[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int p = 32 * t + l31;
                    const int py = 2 * (p >> 3) + ky - 1, px = 2 * (p & 7) + kx - 1;
                    ok[t] = (py >= 0) && (px >= 0);
                    pyc[t] = max(py, 0);
                    pxc[t] = max(px, 0);
                }
#pragma unroll 1
                for (int img = 0; img < 2; ++img) {
                    // this lane's rows: LDS byte offsets of its cell per m-tile and level, and the pixel's scale
                    // (zero for the padding ring: the product is then exactly zero)
                    int ab[2][3];
                    float sc[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
#pragma unroll
                        for (int j = 1; j < 4; ++j) {
                            const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                            const int off = (j == 1) ? XOFF1 : (j == 2) ? XOFF2 : XOFF3;
                            const int st = (j == 3) ? XST3 : XST1;
                            const int cj = patch_cell(XY0(img), pyc[t], j, I.H[img]) * Rr + patch_cell(XX0(img), pxc[t], j, I.W[img]);
                            ab[t][j - 1] = img * XIMG + off + cj * st + half * 32;
                        }
                        sc[t] = ok[t] ? scale[img * 256 + pyc[t] * 16 + pxc[t]] : 0.f;
                    }
                    const unsigned char *a0 = smb + ab[0][0], *a1 = smb + ab[1][0];
                    const unsigned char *b0 = smb + ab[0][1], *b1 = smb + ab[1][1];
                    const unsigned char *c0 = smb + ab[0][2], *c1 = smb + ab[1][2];
                    XPRO(a0, a1, sc[0])
                    // level 1 (64 ch), level 2 (64 ch), level 3 (128 ch): slabs of 16 channels = 64 bytes
                    XSLAB2(a0 + 64, a1 + 64, a0 + 128, a1 + 128, sc[0], sc[1])
                    XSLAB2(a0 + 192, a1 + 192, b0, b1, sc[0], sc[1])
                    XSLAB2(b0 + 64, b1 + 64, b0 + 128, b1 + 128, sc[0], sc[1])
                    XSLAB2(b0 + 192, b1 + 192, c0, c1, sc[0], sc[1])
#pragma unroll 1
                    for (int g = 0; g < 4; ++g) {
                        const int gn = (g < 3) ? g + 1 : 3;
                        XSLAB2(c0 + g * 128 + 64, c1 + g * 128 + 64, c0 + gn * 128, c1 + gn * 128, sc[0], sc[1])
                    }
                }
            }
        }
        __syncthreads();   // all waves are done reading the conv1 operands

        // BN1 -> H[pixel][channel] fp32, plus the all-zero padding row
        {
            if (tid < XHPIX / 16) *(f32x4 *)(smb + 64 * XHPIX + tid * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R_.bn1s[n], b = R_.bn1b[n];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 &a = (t == 0) ? (u == 0 ? acc00 : acc01) : (u == 0 ? acc10 : acc11);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                        *(float *)(smb + p * XHPIX + n * 4) = fmaf(a[r], s, b);
                    }
                }
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ conv2: 3x3, stride 1, pad 1
        acc00 = (f32x16){0}; acc01 = (f32x16){0}; acc10 = (f32x16){0}; acc11 = (f32x16){0};
        {
            const f32x4 *bp = (const f32x4 *)R_.wx2 + (size_t)wave * (S2_UNITS + XPF) * 192 + lane;
            XLOADB(B0, 0) XLOADB(B1, 1)
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int oy = (l31 >> 3) + ky - 1, ox = (l31 & 7) + kx - 1;
                const bool okx = (ox >= 0) && (ox < 8);
                const bool ok0 = okx && (oy >= 0);
                const bool ok1 = okx && (oy + 4 < 8);
                const unsigned char *p0 = smb + (ok0 ? oy * 8 + ox : 64) * XHPIX + half * 32;
                const unsigned char *p1 = smb + (ok1 ? (oy + 4) * 8 + ox : 64) * XHPIX + half * 32;
                XPRO(p0, p1, 1.0f)
#pragma unroll 1
                for (int g = 0; g < 16; ++g) {       // 32 slabs of 16 channels = 64 bytes
                    const int gn = (g < 15) ? g + 1 : 15;
                    XSLAB2(p0 + g * 128 + 64, p1 + g * 128 + 64, p0 + gn * 128, p1 + gn * 128, 1.0f, 1.0f)
                }
            }
        }

        // BN2 -> ReLU -> max over the 8x8 outputs (BN before max: its scale may be negative)
        {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R_.bn2s[n], b = R_.bn2b[n];
                const f32x16 &aa = (u == 0) ? acc00 : acc01;
                const f32x16 &ab2 = (u == 0) ? acc10 : acc11;
                float m = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    m = fmaxf(m, fmaf(aa[r], s, b));
                    m = fmaxf(m, fmaf(ab2[r], s, b));
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                if (half == 0) V[n] = m;
            }
        }
        __syncthreads();

        fc_tail_parse(R_, I, args, lvl, prop, tid, V, F1, F2, misc);
    }
}

// --------------------------------------------------------------------------------------------------
// host side: the weight streams.  conv1: units 0-7 = level 0 ([4 slabs][n-tile]), then [tap][img][16 slabs][n-tile];
// conv2: [tap][32 slabs][n-tile].  A unit is [plane 3][lane 64][8 bf16]; K of a slab as in split_conv1_index.
// --------------------------------------------------------------------------------------------------
static void put3(uint16_t *d, size_t unit_base, int lane, int j, float v) {
    const uint16_t p0 = bf16_rne(v);
    const float r1 = v - bf16_to_f(p0);
    const uint16_t p1 = bf16_rne(r1);
    const uint16_t p2 = bf16_rne(r1 - bf16_to_f(p1));
    d[(unit_base + lane) * 8 + j] = p0;
    d[(unit_base + 64 + lane) * 8 + j] = p1;
    d[(unit_base + 128 + lane) * 8 + j] = p2;
}

void pack_x3_weights(const float *conv1_w, const float *conv2_w, float *wx1, float *wx2) {
    uint16_t *d1 = (uint16_t *)wx1, *d2 = (uint16_t *)wx2;
    for (int w = 0; w < 8; ++w)
        for (int slab = 0; slab < S1_SLABS; ++slab)
            for (int u = 0; u < 2; ++u) {
                const int unit = slab * 2 + u;
                const size_t base = ((size_t)w * (S1_UNITS + XPF) + unit) * 192;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        int ch, tap;
                        split_conv1_index(slab, lane >> 5, j, ch, tap);
                        put3(d1, base, lane, j, (ch < 0) ? 0.f : conv1_w[((size_t)n * 518 + ch) * 9 + tap]);
                    }
            }
    for (int w = 0; w < 8; ++w)
        for (int slab = 0; slab < S2_SLABS; ++slab) {
            const int tap = slab / 32, sin = slab % 32;
            for (int u = 0; u < 2; ++u) {
                const int unit = slab * 2 + u;
                const size_t base = ((size_t)w * (S2_UNITS + XPF) + unit) * 192;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        const int ch = sin * 16 + 8 * (lane >> 5) + j;
                        put3(d2, base, lane, j, conv2_w[((size_t)n * 512 + ch) * 9 + tap]);
                    }
            }
        }
}

int launch_regress_x3(const RegressArgs &a, int n, hipStream_t stream) {
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {false};
    if (dev < 64 && !attr_set[dev]) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)regress_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)XSM_BYTES));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(regress_x3_kernel, dim3(n), dim3(NT), XSM_BYTES, stream, a);
    return check_launch("regress_x3_kernel");
}

}  // namespace p2p
