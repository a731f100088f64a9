// Fine stage, split-precision variant: the same algorithm and decomposition as regress.hip (one
// workgroup per proposal, 8 waves x 64 output channels, patch deduplicated in LDS, weights streamed
// from L2 in consumption order), but the two convolutions run on the bf16 matrix cores with every
// fp32 operand x represented as hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and
// three v_mfma_f32_32x32x16_bf16 per product:   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,
// accumulated in fp32.  That is 6 matrix-core cycles per unit of K instead of 32 for the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32), at an error of ~2^-16 per product: measured against the fp64
// oracle the regressed coordinates move by <= ~2e-4 px (bar: 1e-3 px), see tests/test_gpu_parity.py.
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
constexpr int ST0 = 16, ST1 = 144, ST2 = 144, ST3 = 272;          // bytes per cell (C bf16 + 16 pad)
constexpr int NC0 = 256, NC1 = 81, NC2 = 25, NC3 = 9;             // real cells; index NCj is the zero cell
constexpr int OFF0 = 0;
constexpr int OFF1 = OFF0 + (NC0 + 1) * ST0;                      // 4112
constexpr int OFF2 = OFF1 + (NC1 + 1) * ST1;                      // 15920
constexpr int OFF3 = OFF2 + (NC2 + 1) * ST2;                      // 19664
constexpr int PLANE = 22400;                                      // >= OFF3 + (NC3+1)*ST3, multiple of 16
constexpr int IMGB = 2 * PLANE;                                   // hi plane, lo plane
constexpr int TILESB = 2 * IMGB;                                  // 89600
constexpr int HPIX = 1040;                                        // bytes per pixel row of H (512 bf16 + 16)
constexpr int HPLANE = 65 * HPIX;                                 // 64 pixels + zero row
constexpr int UNIONB = 2 * HPLANE;                                // 135200 (>= TILESB)
constexpr int SM_SCALE = UNIONB;                                  // float [2][256]
constexpr int SM_V = SM_SCALE + 512 * 4;
constexpr int SM_F1 = SM_V + 512 * 4;
constexpr int SM_F2 = SM_F1 + 512 * 4;
constexpr int SM_MISC = SM_F2 + 256 * 4;
constexpr int SM_BYTES = SM_MISC + 16 * 4;
static_assert(OFF3 + (NC3 + 1) * ST3 <= PLANE, "plane too small");
static_assert(TILESB <= UNIONB, "tiles must fit under H");

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

// one slab (16 K values): 3 products x 4 output tiles.  bq = {u0 hi, u0 lo, u1 hi, u1 lo}
#define SLAB_MFMA(C00, C01, C10, C11, AH0, AL0, AH1, AL1, BQ)  \
    C00 = BMFMA(AH0, BQ[0], C00);                              \
    C01 = BMFMA(AH0, BQ[2], C01);                              \
    C10 = BMFMA(AH1, BQ[0], C10);                              \
    C11 = BMFMA(AH1, BQ[2], C11);                              \
    C00 = BMFMA(AH0, BQ[1], C00);                              \
    C01 = BMFMA(AH0, BQ[3], C01);                              \
    C10 = BMFMA(AH1, BQ[1], C10);                              \
    C11 = BMFMA(AH1, BQ[3], C11);                              \
    C00 = BMFMA(AL0, BQ[0], C00);                              \
    C01 = BMFMA(AL0, BQ[2], C01);                              \
    C10 = BMFMA(AL1, BQ[0], C10);                              \
    C11 = BMFMA(AL1, BQ[2], C11);

__global__ __launch_bounds__(NT, 2) void regress_split_kernel(RegressArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int prop = blockIdx.x;
    int it = 0;
    while (it + 1 < args.nitems && prop >= args.start[it + 1]) ++it;
    const ItemDev &I = args.item[it];
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
        int x0[2], y0[2];
        x0[0] = (int)misc[8 + 0] - 8; y0[0] = (int)misc[8 + 1] - 8;
        x0[1] = (int)misc[8 + 2] - 8; y0[1] = (int)misc[8 + 3] - 8;
        __syncthreads();

        // ------------------------------------------------------------ clear tiles (zero cells, pad channels)
        for (int e = tid; e < TILESB / 16; e += NT) ((f32x4 *)smb)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();

        // ------------------------------------------------------------ gather + split (networks/utils.py:4-36)
        for (int img = 0; img < 2; ++img) {
            const int Hh = I.H[img], Ww = I.W[img];
            unsigned char *tb = smb + img * IMGB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int Rr = (j == 0) ? 16 : (j == 1) ? 9 : (j == 2) ? 5 : 3;
                const int Cc = (j == 0) ? 3 : (j == 3) ? 128 : 64;
                const int off = (j == 0) ? OFF0 : (j == 1) ? OFF1 : (j == 2) ? OFF2 : OFF3;
                const int st = (j == 0) ? ST0 : (j == 3) ? ST3 : ST1;
                const int Hj = Hh >> j, Wj = Ww >> j;
                const int r0 = clampi(y0[img] >> j, 0, Hj - 1);
                const int c0 = clampi(x0[img] >> j, 0, Wj - 1);
                const float *src = I.pyr[img][j];
                for (int e = tid; e < Cc * Rr * Rr; e += NT) {
                    const int c = e / (Rr * Rr);
                    const int rem = e - c * (Rr * Rr);
                    const int r = rem / Rr;
                    const int cc = rem - r * Rr;
                    const int sy = min(r0 + r, Hj - 1);
                    const int sx = min(c0 + cc, Wj - 1);
                    const float v = src[((size_t)c * Hj + sy) * Wj + sx];
                    const unsigned short hi = f2bf(v);
                    const unsigned short lo = f2bf(v - bf2f(hi));
                    unsigned char *dst = tb + off + rem * st + c * 2;
                    *(unsigned short *)dst = hi;
                    *(unsigned short *)(dst + PLANE) = lo;
                }
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ per-pixel L2 scale (patch2pix.py:173-174)
        {
            const int img = tid >> 8, pix = tid & 255, py = pix >> 4, px = pix & 15;
            const unsigned char *tb = smb + img * IMGB;
            float ss = 0.f;
            {
                const unsigned char *p = tb + OFF0 + (patch_cell(y0[img], py, 0, I.H[img]) * 16 + patch_cell(x0[img], px, 0, I.W[img])) * ST0;
                ss = sumsq8(*(const f32x4 *)p, *(const f32x4 *)(p + PLANE), ss);
            }
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                const int Cc = (j == 3) ? 128 : 64;
                const int off = (j == 1) ? OFF1 : (j == 2) ? OFF2 : OFF3;
                const int st = (j == 3) ? ST3 : ST1;
                const unsigned char *p = tb + off + (patch_cell(y0[img], py, j, I.H[img]) * Rr + patch_cell(x0[img], px, j, I.W[img])) * st;
                for (int c = 0; c < Cc; c += 8) ss = sumsq8(*(const f32x4 *)(p + c * 2), *(const f32x4 *)(p + c * 2 + PLANE), ss);
            }
            scale[tid] = 1.0f / sqrtf(ss + 1e-6f);
        }
        __syncthreads();

        // ------------------------------------------------------------ conv1: 3x3, stride 2, pad 1
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        {
            const f32x4 *bp = (const f32x4 *)R.ws1 + (size_t)wave * (S1_SLABS + SPF) * 256 + lane;
            f32x4 bc[4], bn[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { bc[q] = bp[q * 64]; bn[q] = bp[256 + q * 64]; }
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
                    // byte addresses of this lane's A fragments, per m-tile and level
                    int ab[2][4];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int tb = img * IMGB;
                        const int c0 = patch_cell(y0[img], pyc[t], 0, I.H[img]) * 16 + patch_cell(x0[img], pxc[t], 0, I.W[img]);
                        ab[t][0] = tb + OFF0 + ((ok[t] && half == 0) ? c0 : NC0) * ST0;
#pragma unroll
                        for (int j = 1; j < 4; ++j) {
                            const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                            const int off = (j == 1) ? OFF1 : (j == 2) ? OFF2 : OFF3;
                            const int st = (j == 3) ? ST3 : ST1;
                            const int cj = patch_cell(y0[img], pyc[t], j, I.H[img]) * Rr + patch_cell(x0[img], pxc[t], j, I.W[img]);
                            ab[t][j] = tb + off + (ok[t] ? cj : Rr * Rr) * st + half * 16;
                        }
                    }
                    f32x16 t00 = {0}, t01 = {0}, t10 = {0}, t11 = {0};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int nslab = (j == 0) ? 1 : (j == 3) ? 8 : 4;
                        const unsigned char *q0 = smb + ab[0][j], *q1 = smb + ab[1][j];
#pragma unroll 2
                        for (int kin = 0; kin < nslab; ++kin) {
                            const f32x4 ah0 = *(const f32x4 *)(q0 + kin * 32);
                            const f32x4 al0 = *(const f32x4 *)(q0 + kin * 32 + PLANE);
                            const f32x4 ah1 = *(const f32x4 *)(q1 + kin * 32);
                            const f32x4 al1 = *(const f32x4 *)(q1 + kin * 32 + PLANE);
                            f32x4 nn[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) nn[q] = bp[256 * SPF + q * 64];
                            bp += 256;
                            SLAB_MFMA(t00, t01, t10, t11, ah0, al0, ah1, al1, bc)
#pragma unroll
                            for (int q = 0; q < 4; ++q) { bc[q] = bn[q]; bn[q] = nn[q]; }
                        }
                    }
                    // fold the unscaled (tap, image) partial sums in with the per-pixel scale of their source pixel
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int py = 2 * (4 * t + (r >> 2)) + ky - 1, px = 2 * (4 * half + (r & 3)) + kx - 1;
                            const float sv = scale[img * 256 + max(py, 0) * 16 + max(px, 0)];
                            if (t == 0) { acc00[r] = fmaf(sv, t00[r], acc00[r]); acc01[r] = fmaf(sv, t01[r], acc01[r]); }
                            else        { acc10[r] = fmaf(sv, t10[r], acc10[r]); acc11[r] = fmaf(sv, t11[r], acc11[r]); }
                        }
                }
            }
        }
        __syncthreads();   // all waves are done reading the patch tiles
#ifdef P2P_DEBUG_SPLIT
        if (prop == 0 && lvl == 0 && tid == 0) {
            printf("DBG scale %g %g %g %g | %g %g\n", scale[0], scale[1], scale[17], scale[255], scale[256], scale[256 + 100]);
            printf("DBG acc00 (n=0; px 0,1,2,3) %g %g %g %g  acc01 (n=32) %g  acc10 (px32) %g\n", acc00[0], acc00[1], acc00[2], acc00[3], acc01[0], acc10[0]);
        }
        if (prop == 0 && lvl == 0 && tid == 33) printf("DBG lane33 acc00[0] (n=1, px 4) %g\n", acc00[0]);
#endif

        // BN1 -> split -> H[pixel][channel] (bf16 hi / lo planes), plus the all-zero padding row
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

        // ------------------------------------------------------------ conv2: 3x3, stride 1, pad 1
        acc00 = (f32x16){0}; acc01 = (f32x16){0}; acc10 = (f32x16){0}; acc11 = (f32x16){0};
        {
            const f32x4 *bp = (const f32x4 *)R.ws2 + (size_t)wave * (S2_SLABS + SPF) * 256 + lane;
            f32x4 bc[4], bn[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { bc[q] = bp[q * 64]; bn[q] = bp[256 + q * 64]; }
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int oy = (l31 >> 3) + ky - 1, ox = (l31 & 7) + kx - 1;
                const bool okx = (ox >= 0) && (ox < 8);
                const bool ok0 = okx && (oy >= 0);
                const bool ok1 = okx && (oy + 4 < 8);
                const unsigned char *p0 = smb + (ok0 ? oy * 8 + ox : 64) * HPIX + half * 16;
                const unsigned char *p1 = smb + (ok1 ? (oy + 4) * 8 + ox : 64) * HPIX + half * 16;
#pragma unroll 4
                for (int s = 0; s < 32; ++s) {
                    const f32x4 ah0 = *(const f32x4 *)(p0 + s * 32);
                    const f32x4 al0 = *(const f32x4 *)(p0 + s * 32 + HPLANE);
                    const f32x4 ah1 = *(const f32x4 *)(p1 + s * 32);
                    const f32x4 al1 = *(const f32x4 *)(p1 + s * 32 + HPLANE);
                    f32x4 nn[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) nn[q] = bp[256 * SPF + q * 64];
                    bp += 256;
                    SLAB_MFMA(acc00, acc01, acc10, acc11, ah0, al0, ah1, al1, bc)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { bc[q] = bn[q]; bn[q] = nn[q]; }
                }
            }
        }

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

        fc_tail_parse(R, I, args, lvl, prop, tid, V, F1, F2, misc);
    }
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
static uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// channel (0..517) of the concatenated regressor input for position (sin, half, j) of a conv1 slab of image `img`
static int split_conv1_channel(int img, int sin, int half, int j) {
    if (sin == 0) return (half == 0 && j < 3) ? img * 259 + j : -1;
    const int base = (sin < 5) ? 3 + (sin - 1) * 16 : (sin < 9) ? 67 + (sin - 5) * 16 : 131 + (sin - 9) * 16;
    return img * 259 + base + 8 * half + j;
}

void pack_split_weights(const float *conv1_w, const float *conv2_w, float *ws1, float *ws2) {
    uint16_t *d1 = (uint16_t *)ws1, *d2 = (uint16_t *)ws2;
    for (int w = 0; w < 8; ++w)
        for (int slab = 0; slab < S1_SLABS; ++slab) {
            const int tap = slab / 34, img = (slab % 34) / 17, sin = slab % 17;
            for (int u = 0; u < 2; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        const int ch = split_conv1_channel(img, sin, lane >> 5, j);
                        const float v = (ch < 0) ? 0.f : conv1_w[((size_t)n * 518 + ch) * 9 + tap];
                        const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f(hi));
                        const size_t base = ((((size_t)w * (S1_SLABS + SPF) + slab) * 2 + u) * 2) * 64;
                        d1[((base + lane) * 8) + j] = hi;
                        d1[((base + 64 + lane) * 8) + j] = lo;
                    }
        }
    for (int w = 0; w < 8; ++w)
        for (int slab = 0; slab < S2_SLABS; ++slab) {
            const int tap = slab / 32, sin = slab % 32;
            for (int u = 0; u < 2; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        const int ch = sin * 16 + 8 * (lane >> 5) + j;
                        const float v = conv2_w[((size_t)n * 512 + ch) * 9 + tap];
                        const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f(hi));
                        const size_t base = ((((size_t)w * (S2_SLABS + SPF) + slab) * 2 + u) * 2) * 64;
                        d2[((base + lane) * 8) + j] = hi;
                        d2[((base + 64 + lane) * 8) + j] = lo;
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
