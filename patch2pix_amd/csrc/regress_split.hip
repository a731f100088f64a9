// Fine stage, split-precision variant: the same algorithm and decomposition as regress.hip (one
// workgroup per proposal, 8 waves x 64 output channels, patch deduplicated in LDS, weights streamed
// from L2 in consumption order), but the two convolutions run on the bf16 matrix cores with every
// fp32 operand x represented as hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and
// three v_mfma_f32_32x32x16_bf16 per product:   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,
// accumulated in fp32.  That is 6 matrix-core cycles per unit of K instead of 32 for the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32), at an error of ~2^-16 per product: measured against an fp64
// evaluation the regressed coordinates move by <= ~2.5e-4 px (bar: 1e-3 px), see tests/test_gpu_parity.py.
//
// Differences to regress.hip that follow from the bf16 operand shape (8 consecutive K per lane):
//   * LDS tiles are channel-innermost ([cell][C] bf16, hi and lo planes) so that a lane's A fragment
//     is one ds_read_b128; cell / pixel strides are padded by 16 B, which makes the reads conflict-free;
//   * the per-pixel L2 scale cannot be multiplied into pre-split operands, so each (tap, image) K-range
//     accumulates unscaled into a second accumulator set that is folded in with one fma per element;
//   * zero padding is a dedicated all-zero cell / pixel row instead of a per-lane mask.
#include "regress_common.h"

#include <cstring>
#include <vector>

namespace p2p {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- LDS layout (bytes) --------------------------------------------------------------------------
// conv1 phase: levels 1-3 deduplicated tiles [img][plane][cell][C bf16 (+16 B pad)], level 0 as raw fp32
// [img][3][256] (it is not deduplicated: ds = 1) and, derived from it, a pre-scaled im2col block
// A0[plane][64 px][64 K bf16 (+16 B pad)] with K = img*32 + tap*3 + c (27 real per image).
// conv2 phase: H[plane][65 px][512 bf16 (+16 B pad)] overlays all of the above.
constexpr int ST1 = 144, ST3 = 272;                               // bytes per cell of levels 1/2 and 3
constexpr int NC1 = 81, NC2 = 25, NC3 = 9;                        // real cells; index NCj is the zero cell
constexpr int OFF1 = 0;
constexpr int OFF2 = OFF1 + (NC1 + 1) * ST1;                      // 11808
constexpr int OFF3 = OFF2 + (NC2 + 1) * ST1;                      // 15552
constexpr int PLANE = OFF3 + (NC3 + 1) * ST3;                     // 18272
constexpr int IMGB = 2 * PLANE;                                   // hi plane, lo plane
constexpr int TILESB = 2 * IMGB;                                  // 73088
constexpr int RAW0 = TILESB;                                      // float [2][3][256]
constexpr int A0OFF = RAW0 + 2 * 3 * 256 * 4;                     // 79232
constexpr int A0ST = 144;                                         // bytes per pixel row of A0 (64 bf16 + 16)
constexpr int A0PLANE = 64 * A0ST;                                // 9216
constexpr int CONV1B = A0OFF + 2 * A0PLANE;                       // 97664
constexpr int HPIX = 1040;                                        // bytes per pixel row of H (512 bf16 + 16)
constexpr int HPLANE = 65 * HPIX;                                 // 64 pixels + zero row
constexpr int UNIONB = 2 * HPLANE;                                // 135200
constexpr int SM_SCALE = UNIONB;                                  // float [2][256]
constexpr int SM_V = SM_SCALE + 512 * 4;
constexpr int SM_F1 = SM_V + 512 * 4;
constexpr int SM_F2 = SM_F1 + 512 * 4;
constexpr int SM_MISC = SM_F2 + 256 * 4;
constexpr int SM_BYTES = SM_MISC + 16 * 4;
static_assert(CONV1B <= UNIONB, "conv1 buffers must fit under H");
static_assert(PLANE % 16 == 0 && A0OFF % 16 == 0 && HPLANE % 16 == 0, "16-byte alignment of ds_read_b128");

__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// sum over the 8 bf16 pairs (hi + lo) of a fragment of (hi+lo)^2
__device__ __forceinline__ float sumsq8(const f32x4 &h, const f32x4 &l, float ss) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned hu = __float_as_uint(h[q]), lu = __float_as_uint(l[q]);
        const float a = __uint_as_float(hu << 16) + __uint_as_float(lu << 16);
        const float b = __uint_as_float(hu & 0xffff0000u) + __uint_as_float(lu & 0xffff0000u);
        ss = fmaf(a, a, ss);
        ss = fmaf(b, b, ss);
    }
    return ss;
}

#define BMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// A fragments of one slab: P0/P1 = this lane's byte address for m-tile 0/1, LO = distance to the lo plane
#define LOADA(BUF, P0, P1, LO)                                                          \
    BUF[0] = *(const f32x4 *)(P0); BUF[1] = *(const f32x4 *)((P0) + (LO));             \
    BUF[2] = *(const f32x4 *)(P1); BUF[3] = *(const f32x4 *)((P1) + (LO));

// One unit = one slab (16 K values) x one n-tile: 3 products x 2 m-tiles = 6 MFMAs, 2 KiB of weights
// per wave.  A = {m-tile0 hi, m-tile0 lo, m-tile1 hi, m-tile1 lo}, BQ = {hi, lo}.
#define SLAB_MFMA6(C0, C1, A, BQ)                              \
    C0 = BMFMA(A[0], BQ[0], C0);                               \
    C1 = BMFMA(A[2], BQ[0], C1);                               \
    C0 = BMFMA(A[0], BQ[1], C0);                               \
    C1 = BMFMA(A[2], BQ[1], C1);                               \
    C0 = BMFMA(A[1], BQ[0], C0);                               \
    C1 = BMFMA(A[3], BQ[0], C1);
#define LOADB2(BUF, UNIT_AHEAD)                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) BUF[q_] = bp[(UNIT_AHEAD) * 128 + q_ * 64];
// Eight consecutive units, software-pipelined by hand: while unit s runs on the matrix cores, the weight
// load for unit s+6 (ring of eight register buffers B0..B7) and the LDS reads for unit s+1 (A0_/A1_
// alternate) are in flight.  The load goes into the buffer that was consumed TWO units ago, never the one the
// matrix pipe has just read: a VMEM write to a register an in-flight MFMA sources costs ~pass-count wait states.  The caller pre-loads A0_ with the group's first unit.  (PA0,PA1) / (PB0,PB1)
// are the lane's A addresses of units 0-3 / 4-7 (STEP bytes apart), (NP0,NP1) the first unit of whatever
// follows.  The sched_barriers pin "loads first, then the MFMAs" and keep later units' loads from being
// hoisted (which would blow the register budget).
#define UNIT_(C0, C1, BLOAD, AHEAD, ANEXT, NA0, NA1, LO, ACUR, BCUR)                                      \
    LOADB2(BLOAD, AHEAD) LOADA(ANEXT, (NA0), (NA1), LO) __builtin_amdgcn_sched_barrier(0);                \
    SLAB_MFMA6(C0, C1, ACUR, BCUR) __builtin_amdgcn_sched_barrier(0);
#define GROUP4H(C0, C1, P0, P1, NP0, NP1, STEP, LO)                                                        \
    { UNIT_(C0, C1, B2, 2, A1_, (P0) + (STEP), (P1) + (STEP), LO, A0_, B0)                                 \
      UNIT_(C0, C1, B3, 3, A0_, (P0) + 2 * (STEP), (P1) + 2 * (STEP), LO, A1_, B1)                         \
      UNIT_(C0, C1, B0, 4, A1_, (P0) + 3 * (STEP), (P1) + 3 * (STEP), LO, A0_, B2)                         \
      UNIT_(C0, C1, B1, 5, A0_, (NP0), (NP1), LO, A1_, B3)                                                 \
      bp += 4 * 128; }
#define GROUP8H(C0, C1, PA0, PA1, PB0, PB1, NP0, NP1, STEP, LO)                                            \
    { UNIT_(C0, C1, B6, 6,  A1_, (PA0) + (STEP), (PA1) + (STEP), LO, A0_, B0)                              \
      UNIT_(C0, C1, B7, 7,  A0_, (PA0) + 2 * (STEP), (PA1) + 2 * (STEP), LO, A1_, B1)                      \
      UNIT_(C0, C1, B0, 8,  A1_, (PA0) + 3 * (STEP), (PA1) + 3 * (STEP), LO, A0_, B2)                      \
      UNIT_(C0, C1, B1, 9,  A0_, (PB0), (PB1), LO, A1_, B3)                                                \
      UNIT_(C0, C1, B2, 10, A1_, (PB0) + (STEP), (PB1) + (STEP), LO, A0_, B4)                              \
      UNIT_(C0, C1, B3, 11, A0_, (PB0) + 2 * (STEP), (PB1) + 2 * (STEP), LO, A1_, B5)                      \
      UNIT_(C0, C1, B4, 12, A1_, (PB0) + 3 * (STEP), (PB1) + 3 * (STEP), LO, A0_, B6)                      \
      UNIT_(C0, C1, B5, 13, A0_, (NP0), (NP1), LO, A1_, B7)                                                \
      bp += 8 * 128; }

#ifdef P2P_SPLIT_TIMING
#define STAMP(i) stamps[i] = __builtin_amdgcn_s_memtime();
#else
#define STAMP(i)
#endif

__global__ __launch_bounds__(NT, 2) void regress_split_kernel(RegressArgs args) {
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

    float *raw0 = (float *)(smb + RAW0);
    float *scale = (float *)(smb + SM_SCALE);
    float *V = (float *)(smb + SM_V);
    float *F1 = (float *)(smb + SM_F1);
    float *F2 = (float *)(smb + SM_F2);
    float *misc = (float *)(smb + SM_MISC);

    if (tid < 4) {
        float v;
        if (args.is_float) v = ((const float *)args.proposals)[prop * 4 + tid];
        else v = (float)((const long long *)args.proposals)[prop * 4 + tid];
        misc[8 + tid] = v;
    }
    __syncthreads();

    for (int lvl = 0; lvl < args.nlevels; ++lvl) {
        const RegDev &R = args.reg[lvl];
        // window origins (x, y) in image 1 / image 2; scalars + selects, never a runtime-indexed array
        const int xa = (int)misc[8 + 0] - 8, ya = (int)misc[8 + 1] - 8;
        const int xb = (int)misc[8 + 2] - 8, yb = (int)misc[8 + 3] - 8;
#define X0(img_) ((img_) ? xb : xa)
#define Y0(img_) ((img_) ? yb : ya)
        __syncthreads();
#ifdef P2P_SPLIT_TIMING
        unsigned long long stamps[12];
#endif
        STAMP(0)
        // Opaque copy of the thread id for the staging phases: their index arithmetic depends only on the
        // thread id, and without this the compiler hoists all of it out of the level loop and spills it
        // (scratch must stay at zero, see build.py).
        int tidv = tid;
        P2P_OPAQUE(tidv);

        // ------------------------------------------------------------ zero cells of the tiles
        if (tidv < 2 * 2 * 3) {
            const int j = tidv % 3 + 1, plane = (tidv / 3) & 1, img = tidv / 6;
            const int off = (j == 1) ? OFF1 + NC1 * ST1 : (j == 2) ? OFF2 + NC2 * ST1 : OFF3 + NC3 * ST3;
            const int nb = (j == 3) ? ST3 : ST1;
            unsigned char *z = smb + img * IMGB + plane * PLANE + off;
            for (int q = 0; q < nb; q += 16) *(f32x4 *)(z + q) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // ------------------------------------------------------------ gather (networks/utils.py:4-36)
#ifdef P2P_SPLIT_SKIP_GATHER
        if (args.n < 0)
#endif
        {
            // two passes so that all ~40 scattered 4-byte loads of a thread are in flight together:
            // first every address and load (both images, levels 0-3), then split + store
            float g0[2][2], g1[2][11], g2[2][4], g3[2][3];
#pragma unroll
            for (int img = 0; img < 2; ++img) {
                const int Hh = I.H[img], Ww = I.W[img];
                {
                    const int r0 = clampi(Y0(img), 0, Hh - 1), c0 = clampi(X0(img), 0, Ww - 1);
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
                    const int Hj = Hh >> j, Wj = Ww >> j;                 // index clamp: dim // ds (networks/utils.py:22-23)
                    const int Ha = level_dim(Hh, j), Wa = level_dim(Ww, j);  // extent of the backbone's map
                    const int r0 = clampi(Y0(img) >> j, 0, Hj - 1);
                    const int c0 = clampi(X0(img) >> j, 0, Wj - 1);
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
                unsigned char *tb = smb + img * IMGB;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                    const int Cc = (j == 3) ? 128 : 64;
                    const int nk = (j == 1) ? 11 : (j == 2) ? 4 : 3;
                    const int off = (j == 1) ? OFF1 : (j == 2) ? OFF2 : OFF3;
                    const int st = (j == 3) ? ST3 : ST1;
#pragma unroll
                    for (int k = 0; k < nk; ++k) {
                        const int e = tidv + k * NT;
                        if (e < Cc * Rr * Rr) {
                            const int c = e / (Rr * Rr);
                            const int rem = e - c * (Rr * Rr);
                            const float v = (j == 1) ? g1[img][k] : (j == 2) ? g2[img][k] : g3[img][k];
                            const unsigned short hi = f2bf(v);
                            const unsigned short lo = f2bf(v - bf2f(hi));
                            unsigned char *dst = tb + off + rem * st + c * 2;
                            *(unsigned short *)dst = hi;
                            *(unsigned short *)(dst + PLANE) = lo;
                        }
                    }
                }
            }
        }
        __syncthreads();
        STAMP(1)

        // ------------------------------------------------------------ per-pixel L2 scale (patch2pix.py:173-174)
        {
            const int img = tidv >> 8, pix = tidv & 255, py = pix >> 4, px = pix & 15;
            const unsigned char *tb = smb + img * IMGB;
            float ss = 0.f;
            {
                const float *p = raw0 + img * 768 + patch_cell(Y0(img), py, 0, I.H[img]) * 16 + patch_cell(X0(img), px, 0, I.W[img]);
#pragma unroll
                for (int c = 0; c < 3; ++c) ss = fmaf(p[c * 256], p[c * 256], ss);
            }
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                const int Cc = (j == 3) ? 128 : 64;
                const int off = (j == 1) ? OFF1 : (j == 2) ? OFF2 : OFF3;
                const int st = (j == 3) ? ST3 : ST1;
                const unsigned char *p = tb + off + (patch_cell(Y0(img), py, j, I.H[img]) * Rr + patch_cell(X0(img), px, j, I.W[img])) * st;
                for (int c = 0; c < Cc; c += 8) ss = sumsq8(*(const f32x4 *)(p + c * 2), *(const f32x4 *)(p + c * 2 + PLANE), ss);
            }
            scale[tid] = 1.0f / sqrtf(ss + 1e-6f);
        }
        __syncthreads();
        STAMP(2)

        // ------------------------------------------------------------ level-0 im2col block, pre-scaled and split
        for (int e = tidv; e < 64 * 64; e += NT) {
            const int m = e >> 6, kk = e & 63, img = kk >> 5, r = kk & 31;
            float v = 0.f;
            if (r < 27) {
                const int tap = r / 3, c = r - tap * 3, ky = tap / 3, kx = tap - ky * 3;
                const int py = 2 * (m >> 3) + ky - 1, px = 2 * (m & 7) + kx - 1;
                if (py >= 0 && px >= 0)
                    v = raw0[img * 768 + c * 256 + patch_cell(Y0(img), py, 0, I.H[img]) * 16 + patch_cell(X0(img), px, 0, I.W[img])] *
                        scale[img * 256 + py * 16 + px];
            }
            const unsigned short hi = f2bf(v);
            const unsigned short lo = f2bf(v - bf2f(hi));
            unsigned char *dst = smb + A0OFF + m * A0ST + kk * 2;
            *(unsigned short *)dst = hi;
            *(unsigned short *)(dst + A0PLANE) = lo;
        }
        __syncthreads();

        STAMP(3)
        // ------------------------------------------------------------ conv1: 3x3, stride 2, pad 1
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        {
            const f32x4 *bp = (const f32x4 *)R.ws1 + (size_t)wave * (S1_UNITS + SPF) * 128 + lane;
            // conv1 carries two accumulator sets (scaled sum + unscaled partial), so its weight ring is 4 deep
            f32x4 B0[2], B1[2], B2[2], B3[2], A0_[4], A1_[4];
#ifdef P2P_SPLIT_SKIP_CONV
            if (args.n < 0)     // never true: timing experiment without the MFMA loops
#endif
            {
            LOADB2(B0, 0) LOADB2(B1, 1)
            {   // level 0 of both images: 4 slabs, already scaled -> straight into the accumulators.  The
                // stream interleaves the two n-tiles per slab here so that the group has its 8 units.
                const unsigned char *p0 = smb + A0OFF + l31 * A0ST + half * 16;
                const unsigned char *p1 = p0 + 32 * A0ST;
                LOADA(A0_, p0, p1, A0PLANE)
                UNIT_(acc00, acc10, B2, 2, A1_, p0, p1, A0PLANE, A0_, B0)
                UNIT_(acc01, acc11, B3, 3, A0_, p0 + 32, p1 + 32, A0PLANE, A1_, B1)
                UNIT_(acc00, acc10, B0, 4, A1_, p0 + 32, p1 + 32, A0PLANE, A0_, B2)
                UNIT_(acc01, acc11, B1, 5, A0_, p0 + 64, p1 + 64, A0PLANE, A1_, B3)
                bp += 4 * 128;
                UNIT_(acc00, acc10, B2, 2, A1_, p0 + 64, p1 + 64, A0PLANE, A0_, B0)
                UNIT_(acc01, acc11, B3, 3, A0_, p0 + 96, p1 + 96, A0PLANE, A1_, B1)
                UNIT_(acc00, acc10, B0, 4, A1_, p0 + 96, p1 + 96, A0PLANE, A0_, B2)
                UNIT_(acc01, acc11, B1, 5, A0_, p0, p1, A0PLANE, A1_, B3)
                bp += 4 * 128;
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
                    // byte addresses of this lane's A fragments, per m-tile and level (zero cell when padding)
                    int ab[2][3];            // LDS byte offsets
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int j = 1; j < 4; ++j) {
                            const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                            const int off = (j == 1) ? OFF1 : (j == 2) ? OFF2 : OFF3;
                            const int st = (j == 3) ? ST3 : ST1;
                            const int cj = patch_cell(Y0(img), pyc[t], j, I.H[img]) * Rr + patch_cell(X0(img), pxc[t], j, I.W[img]);
                            ab[t][j - 1] = img * IMGB + off + (ok[t] ? cj : Rr * Rr) * st + half * 16;
                        }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f32x16 t0 = {0}, t1 = {0};
                        LOADA(A0_, smb + ab[0][0], smb + ab[1][0], PLANE)
                        // levels 1 and 2 (64 channels each), then level 3 (128 channels)
                        GROUP4H(t0, t1, smb + ab[0][0], smb + ab[1][0], smb + ab[0][1], smb + ab[1][1], 32, PLANE)
                        GROUP4H(t0, t1, smb + ab[0][1], smb + ab[1][1], smb + ab[0][2], smb + ab[1][2], 32, PLANE)
                        GROUP4H(t0, t1, smb + ab[0][2], smb + ab[1][2], smb + ab[0][2] + 128, smb + ab[1][2] + 128, 32, PLANE)
                        GROUP4H(t0, t1, smb + ab[0][2] + 128, smb + ab[1][2] + 128, smb + ab[0][0], smb + ab[1][0], 32, PLANE)
                        // fold the unscaled partial sums in with the scale of the source pixel of each row
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int px = 2 * (4 * half + (r & 3)) + kx - 1;
                            const int py0 = 2 * (r >> 2) + ky - 1, py1 = py0 + 8;
                            const float s0 = scale[img * 256 + max(py0, 0) * 16 + max(px, 0)];
                            const float s1 = scale[img * 256 + py1 * 16 + max(px, 0)];
                            if (u == 0) { acc00[r] = fmaf(s0, t0[r], acc00[r]); acc10[r] = fmaf(s1, t1[r], acc10[r]); }
                            else        { acc01[r] = fmaf(s0, t0[r], acc01[r]); acc11[r] = fmaf(s1, t1[r], acc11[r]); }
                        }
                    }
                }
            }
            }
        }
        STAMP(4)
        __syncthreads();   // all waves are done reading the conv1 operands

        // BN1 -> split -> H[pixel][channel] (bf16 hi / lo planes), plus the all-zero padding row
#ifdef P2P_SPLIT_SKIP_HWRITE
        if (args.n < 0)
#endif
        {
            if (tid < 2 * (HPIX / 16)) {
                const int plane = tid / (HPIX / 16), q = tid - plane * (HPIX / 16);
                *(f32x4 *)(smb + plane * HPLANE + 64 * HPIX + q * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R.bn1s[n], b = R.bn1b[n];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 &a = (t == 0) ? (u == 0 ? acc00 : acc01) : (u == 0 ? acc10 : acc11);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = fmaf(a[r], s, b);
                        const unsigned short hi = f2bf(v);
                        const unsigned short lo = f2bf(v - bf2f(hi));
                        const int p = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                        unsigned char *dst = smb + p * HPIX + n * 2;
                        *(unsigned short *)dst = hi;
                        *(unsigned short *)(dst + HPLANE) = lo;
                    }
                }
            }
        }
        __syncthreads();

        STAMP(5)
        // ------------------------------------------------------------ conv2: 3x3, stride 1, pad 1
        acc00 = (f32x16){0}; acc01 = (f32x16){0}; acc10 = (f32x16){0}; acc11 = (f32x16){0};
        {
            const f32x4 *bp = (const f32x4 *)R.ws2 + (size_t)wave * (S2_UNITS + SPF) * 128 + lane;
            f32x4 B0[2], B1[2], B2[2], B3[2], B4[2], B5[2], B6[2], B7[2], A0_[4], A1_[4];
            LOADB2(B0, 0) LOADB2(B1, 1) LOADB2(B2, 2) LOADB2(B3, 3) LOADB2(B4, 4) LOADB2(B5, 5)
#ifdef P2P_SPLIT_SKIP_CONV
            if (args.n < 0)
#endif
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int oy = (l31 >> 3) + ky - 1, ox = (l31 & 7) + kx - 1;
                const bool okx = (ox >= 0) && (ox < 8);
                const bool ok0 = okx && (oy >= 0);
                const bool ok1 = okx && (oy + 4 < 8);
                const unsigned char *p0 = smb + (ok0 ? oy * 8 + ox : 64) * HPIX + half * 16;
                const unsigned char *p1 = smb + (ok1 ? (oy + 4) * 8 + ox : 64) * HPIX + half * 16;
                // one n-tile at a time: [tap][n-tile][32 slabs]
                LOADA(A0_, p0, p1, HPLANE)
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
                    const int gn = (g < 3) ? g + 1 : 0;
                    GROUP8H(acc00, acc10, p0 + g * 256, p1 + g * 256, p0 + g * 256 + 128, p1 + g * 256 + 128,
                            p0 + gn * 256, p1 + gn * 256, 32, HPLANE)
                }
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
                    const int gn = (g < 3) ? g + 1 : 3;
                    GROUP8H(acc01, acc11, p0 + g * 256, p1 + g * 256, p0 + g * 256 + 128, p1 + g * 256 + 128,
                            p0 + gn * 256, p1 + gn * 256, 32, HPLANE)
                }
            }
        }

        STAMP(6)
        // BN2 -> ReLU -> max over the 8x8 outputs
        {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R.bn2s[n], b = R.bn2b[n];
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

        STAMP(7)
#ifdef P2P_SPLIT_SKIP_FC
        if (args.n < 0)
#endif
        fc_tail_parse(R, I, args, lvl, prop, tid, V, F1, F2, misc);
        STAMP(8)
#ifdef P2P_SPLIT_TIMING
        // raw[0] doubles as the stamp buffer in timing builds: workgroups < 64 record [prop][wave][8] phase lengths
        if (args.raw[0] && prop < 64 && lane == 0 && lvl == 0) {
            float *dbg = args.raw[0] + 5 * args.n + (prop * 8 + wave) * 8;
            for (int i = 0; i < 8; ++i) dbg[i] = (float)(stamps[i + 1] - stamps[i]);
        }
#endif
    }
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
// conv1 K layout (see the kernel): slabs 0-3 hold level 0 of both images with K = img*32 + tap*3 + c;
// slab 4 + (tap*2 + img)*16 + s holds 16 channels of level 1 (s 0-3), 2 (s 4-7) or 3 (s 8-15) of image `img`.
// Returns the channel (0..517) of the concatenated regressor input and the tap, or ch = -1 for padding.
void split_conv1_index(int slab, int half, int j, int &ch, int &tap) {
    if (slab < 4) {
        const int kk = slab * 16 + 8 * half + j, img = kk >> 5, r = kk & 31;
        if (r >= 27) { ch = -1; tap = 0; return; }
        tap = r / 3;
        ch = img * 259 + (r % 3);
        return;
    }
    const int q = slab - 4, s = q % 16, img = (q / 16) % 2;
    tap = q / 32;
    const int base = (s < 4) ? 3 + s * 16 : (s < 8) ? 67 + (s - 4) * 16 : 131 + (s - 8) * 16;
    ch = img * 259 + base + 8 * half + j;
}

void pack_split_weights(const float *conv1_w, const float *conv2_w, float *ws1, float *ws2) {
    uint16_t *d1 = (uint16_t *)ws1, *d2 = (uint16_t *)ws2;
    // conv1 stream order per wave: level 0 [4 slabs][u], then [tap][img][u][16 slabs]; 2 KiB units [plane][lane][8]
    for (int w = 0; w < 8; ++w)
        for (int slab = 0; slab < S1_SLABS; ++slab)
            for (int u = 0; u < 2; ++u) {
                int unit;
                if (slab < 4) unit = slab * 2 + u;      // level 0: the two n-tiles alternate
                else { const int q = slab - 4; unit = 8 + ((q / 16) * 2 + u) * 16 + (q % 16); }
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        int ch, tap;
                        split_conv1_index(slab, lane >> 5, j, ch, tap);
                        const float v = (ch < 0) ? 0.f : conv1_w[((size_t)n * 518 + ch) * 9 + tap];
                        const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f(hi));
                        const size_t base = (((size_t)w * (S1_UNITS + SPF) + unit) * 2) * 64;
                        d1[((base + lane) * 8) + j] = hi;
                        d1[((base + 64 + lane) * 8) + j] = lo;
                    }
            }
    // conv2 stream order per wave: [tap][u][32 slabs]
    for (int w = 0; w < 8; ++w)
        for (int slab = 0; slab < S2_SLABS; ++slab) {
            const int tap = slab / 32, sin = slab % 32;
            for (int u = 0; u < 2; ++u) {
                const int unit = (tap * 2 + u) * 32 + sin;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        const int ch = sin * 16 + 8 * (lane >> 5) + j;
                        const float v = conv2_w[((size_t)n * 512 + ch) * 9 + tap];
                        const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f(hi));
                        const size_t base = (((size_t)w * (S2_UNITS + SPF) + unit) * 2) * 64;
                        d2[((base + lane) * 8) + j] = hi;
                        d2[((base + 64 + lane) * 8) + j] = lo;
                    }
            }
        }
}

int launch_regress_split(const RegressArgs &a, int n, hipStream_t stream) {
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {false};
    if (dev < 64 && !attr_set[dev]) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)regress_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)SM_BYTES));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(regress_split_kernel, dim3(n), dim3(NT), SM_BYTES, stream, a);
    return check_launch("regress_split_kernel");
}

}  // namespace p2p
