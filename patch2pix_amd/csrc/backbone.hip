// Convolutions of the feature-pyramid producer (reference networks/resnet.py:26-60 BasicBlock, :125-157 forward_all:
// every 3x3 / 1x1 Conv2d + BatchNorm2d (eval) [+ skip] [+ ReLU] of layer1..layer3) as implicit GEMMs on the fp16 matrix
// cores, fp32-equivalent like the other 16-bit paths of the library.
//
// Activations live in HBM as fp32 NHWC together with the float bits of max |x| per image (written by the producing
// kernel with one atomicMax per wave).  A work-group computes an output tile [2 MT WM rows][16 columns] x [32 NT WN
// channels]; the K axis is walked in chunks of CK input channels: the halo tile of the chunk is read once (coalesced
// 32-byte pieces), scaled by the power of two that brings the image's largest value to [2^11, 2^12), split into two
// fp16 planes (x 2^s = h0 + h1 to within 2^-24) and staged in LDS [plane][pixel][CK (+16 B)] -- the pad makes the
// 16-byte fragment reads of 16 neighbouring pixels hit 16 different bank groups.  Inside a chunk every tap (dy, dx) is a
// shifted view of the staged tile (no im2col); per slab of 16 K values a wave issues MT x 2 ds_read_b128 (A: pixels),
// NT x 2 global_load_dwordx4 (B: weights, packed at load time in MFMA-fragment and consumption order per wave, every
// output channel scaled by a power of two into [2^11, 2^12) and split the same way) and MT x NT x 3
// v_mfma_f32_32x32x16_f16 (a0 b0 + a0 b1 + a1 b0).  Chunks are double buffered: the global loads of chunk c + 1 are in
// flight during the MFMAs of chunk c, one barrier per chunk.  Epilogue: acc x (BN scale / weight scale / activation
// scale) + BN shift [+ skip] [ReLU] -> fp32 NHWC (32 consecutive channels of a pixel per half wave), running max.
#include "p2p_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace p2p {

typedef _Float16 bh8 __attribute__((ext_vector_type(8)));
typedef _Float16 bh2 __attribute__((ext_vector_type(2)));
typedef float bf2 __attribute__((ext_vector_type(2)));
typedef unsigned bu2 __attribute__((ext_vector_type(2)));

struct ConvArgs {
    const float *x;          // [n][h][w][ci]
    const float *res;        // optional skip [n][ho][wo][co]
    float *y;                // [n][ho][wo][co]
    const int *xmax;         // [n] float bits of max |x| per image
    int *ymax;               // optional [n], zero on entry
    const unsigned char *wq; // packed weights
    const float *sc, *sh;    // [co] BN scale x 2^-(weight exponent), BN shift
    int n, h, w, ci, ho, wo, co, ks, stride, relu, ck, tiles_x, tiles_y;
};

#define BB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bh8, (a)), __builtin_bit_cast(bh8, (b)), (c), 0, 0, 0)

__device__ __forceinline__ unsigned bb_pk(float a, float b) {
    const bf2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bh2));
}
__device__ __forceinline__ float bb_lo(unsigned h) { return (float)__builtin_bit_cast(bh2, h)[0]; }
__device__ __forceinline__ float bb_hi(unsigned h) { return (float)__builtin_bit_cast(bh2, h)[1]; }
__device__ __forceinline__ int bb_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 8 consecutive channels x s -> two planes of 8 fp16
__device__ __forceinline__ void bb_split(const f32x4 &xa, const f32x4 &xb, float s, f32x4 &p0, f32x4 &p1) {
    const float x[8] = {xa[0] * s, xa[1] * s, xa[2] * s, xa[3] * s, xb[0] * s, xb[1] * s, xb[2] * s, xb[3] * s};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h = bb_pk(x[2 * q], x[2 * q + 1]);
        p0[q] = __uint_as_float(h);
        p1[q] = __uint_as_float(bb_pk(x[2 * q] - bb_lo(h), x[2 * q + 1] - bb_hi(h)));
    }
}

// MT m-tiles (32 pixels = 2 rows x 16 columns) x NT n-tiles (32 channels) per wave, WN waves along the channels (4 / WN
// along the rows), NIT staged 8-channel pieces per thread and chunk
// DB: slabs of weights in flight (ring of register sets; 3 for the small tiles of 3x3 convolutions, whose slabs are short)
template <int MT, int NT, int WN, int NIT, int DB>
__global__ __launch_bounds__(256, 2) void conv_kernel(ConvArgs a) {
    P2P_DYN_SHARED(unsigned char, sm);
    constexpr int WM = 4 / WN, TH = 2 * MT * WM, TW = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kb5 = lane >> 5;
    int g = blockIdx.x;
    const int tx = g % a.tiles_x; g /= a.tiles_x;
    const int ty = g % a.tiles_y;
    const int img = g / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int S = a.stride, KS = a.ks, PAD = KS >> 1;
    const int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    const int CK = a.ck, SPC = CK >> 4, gs = (CK == 32) ? 2 : 1, PS = CK * 2 + 16;
    const int PLB = (IH * IW * PS + 15) & ~15, BUFB = 2 * PLB;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const float *xi = a.x + (size_t)img * a.h * a.w * a.ci;

    // scale of this image's activations: max |x| < 2^(xeb - 126) -> x * 2^(138 - xeb) < 2^12
    const int xeb = (a.xmax[img] >> 23) & 0xff;
    const int sx = bb_clamp(138 - xeb, -100, 100);
    const float up = __int_as_float((127 + sx) << 23), down = __int_as_float((127 - sx) << 23);

    // staging: what this thread fetches per chunk (does not depend on the chunk but for the channel offset)
    int soff[NIT], sdst[NIT];
    unsigned sok = 0;                                         // bit it: inside the image; bit 8 + it: the piece exists
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = it * 256 + tid, px = e >> gs, gq = e & ((1 << gs) - 1);
        const int pxc = min(px, IH * IW - 1), py = pxc / IW, pxx = pxc - py * IW;
        const int iy = iy0 + py, ix = ix0 + pxx;
        soff[it] = (bb_clamp(iy, 0, a.h - 1) * a.w + bb_clamp(ix, 0, a.w - 1)) * a.ci + gq * 8;
        sdst[it] = pxc * PS + gq * 16;
        sok |= (unsigned)(iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) << it;
        sok |= (unsigned)(px < IH * IW) << (8 + it);
    }
    f32x4 xr[NIT][2];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const float *p = xi + soff[it] + c * CK;          // clamped address, always valid; zero padding chosen at the commit
            xr[it][0] = *(const f32x4 *)p;
            xr[it][1] = *(const f32x4 *)(p + 4);
        }
    };
    auto stage_commit = [&](int buf) {
        unsigned char *dst = sm + buf * BUFB;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 p0, p1;
            bb_split(xr[it][0], xr[it][1], ((sok >> it) & 1u) ? up : 0.f, p0, p1);
            if ((sok >> (8 + it)) & 1u) {
                *(f32x4 *)(dst + sdst[it]) = p0;
                *(f32x4 *)(dst + PLB + sdst[it]) = p1;
            }
        }
    };

    // A fragments: lane (pixel l31 of the m-tile, K block kb5)
    int abase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ry = (wm * MT + i) * 2 + (l31 >> 4), cx = l31 & 15;
        abase[i] = (ry * S * IW + cx * S) * PS + kb5 * 16;
    }
    // B operands: [chunk][tap][slab][n-tile of 32 channels][plane][lane][8 fp16]; a wave reads NT neighbouring n-tiles per slab
    const int nchunks = a.ci / CK, nsl = KS * KS * SPC;
    const int wstep = (a.co >> 5) * 2048;
    const unsigned char *wb = a.wq + (size_t)((blockIdx.y * WN + wn) * NT) * 2048 + lane * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr bool INPLACE = DB > 1 || MT * NT >= 8;
    f32x4 bq[DB][NT][2];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bq[d][j][0] = *(const f32x4 *)(wb + d * wstep + j * 2048);
            bq[d][j][1] = *(const f32x4 *)(wb + d * wstep + j * 2048 + 1024);
        }
    stage_load(0);
    stage_commit(0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) stage_load(c + 1);
        const unsigned char *buf = sm + (c & 1) * BUFB;
        f32x4 av[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            av[i][0] = *(const f32x4 *)(buf + abase[i]);
            av[i][1] = *(const f32x4 *)(buf + PLB + abase[i]);
        }
        int tyy = 0, txx = 0, sl = 0;
        for (int s = 0; s < nsl; s += DB) {                   // (DB divides the slabs of a chunk: host side)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                // pixel fragments of the next slab (the last one of the chunk re-reads itself)
                if (s + d + 1 < nsl) {
                    if (++sl == SPC) { sl = 0; if (++txx == KS) { txx = 0; ++tyy; } }
                }
                const int naoff = (tyy * IW + txx) * PS + sl * 32;
                f32x4 na[MT][2];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    na[i][0] = *(const f32x4 *)(buf + abase[i] + naoff);
                    na[i][1] = *(const f32x4 *)(buf + PLB + abase[i] + naoff);
                }
                wb += wstep;                                  // (DB slabs of zeros follow the last one)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (INPLACE) {
                    // a unit (n-tile) of weights is re-loaded in place right after its MFMAs, for the slab DB further on:
                    // the large tile has no registers for a second set (128 accumulators) and is 18 MFMAs ahead with
                    // DB = 1, the small ones keep three slabs in flight.  (sched_barrier: the scheduler would otherwise order
                    // the MFMAs by product and issue all the loads at the end of the slab.)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][j] = BB_MFMA(av[i][1], bq[d][j][0], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][j] = BB_MFMA(av[i][0], bq[d][j][1], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][j] = BB_MFMA(av[i][0], bq[d][j][0], acc[i][j]);
                        bq[d][j][0] = *(const f32x4 *)(wb + (DB - 1) * wstep + j * 2048);
                        bq[d][j][1] = *(const f32x4 *)(wb + (DB - 1) * wstep + j * 2048 + 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    // the weights of the next slab are loaded into a second register set before the MFMAs of this one
                    f32x4 nb[NT][2];
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        nb[j][0] = *(const f32x4 *)(wb + j * 2048);
                        nb[j][1] = *(const f32x4 *)(wb + j * 2048 + 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][j] = BB_MFMA(av[i][1], bq[0][j][0], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][j] = BB_MFMA(av[i][0], bq[0][j][1], acc[i][j]);
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][j] = BB_MFMA(av[i][0], bq[0][j][0], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NT; ++j) { bq[0][j][0] = nb[j][0]; bq[0][j][1] = nb[j][1]; }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) { av[i][0] = na[i][0]; av[i][1] = na[i][1]; }
            }
        }
        if (c + 1 < nchunks) stage_commit((c + 1) & 1);
        __syncthreads();
    }

    // epilogue.  D of an m-tile: row (r & 3) + 8 (r >> 2) + 4 kb5 = pixel, column l31 = channel
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = (blockIdx.y * WN + wn) * NT * 32 + j * 32 + l31;
        const float scj = a.sc[co] * down, shj = a.sh[co];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kb5;
                const int oy = oy0 + (wm * MT + i) * 2 + (m >> 4), ox = ox0 + (m & 15);
                if (oy < a.ho && ox < a.wo) {
                    const size_t o = (((size_t)img * a.ho + oy) * a.wo + ox) * a.co + co;
                    float v = fmaf(acc[i][j][r], scj, shj);
                    if (a.res) v += a.res[o];
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.y[o] = v;
                    vmax = fmaxf(vmax, fabsf(v));
                }
            }
    }
    if (a.ymax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
        int *dst = a.ymax + img;
        if (lane == 0 && __float_as_int(vmax) > *(volatile int *)dst) atomicMax(dst, __float_as_int(vmax));
    }
}

// ---- the stem: Conv2d(3, 64, 7, stride 2, padding 3) + BatchNorm2d + ReLU (reference resnet.py:101-103, 141-143) -----------
// image NCHW fp32 -> level 1 of the pyramid, NCHW fp32 (the layout the fine stage gathers from).  Here the CHANNELS are the
// rows of the MFMA (A = weights, kept in registers for the whole tile) and 32 neighbouring pixels of an output row its
// columns, so that a lane stores along x.  K = 7 rows ky x 8 columns kx (the eighth is zero) x 4 channels (the fourth is
// zero) = 14 slabs of 16: with the image patch staged as [plane][row][column][4 x fp16], the 8 K values of a lane are two
// neighbouring columns = one aligned 16-byte read, and the stride-2 walk of the 32 pixels makes a wave's reads contiguous.
struct StemArgs {
    const float *x;          // [n][3][h][w]
    float *y;                // [n][64][ho][wo]
    const int *xmax;         // [n]
    const unsigned char *wq; // [m-tile 2][slab 14][plane 2][lane 64][8 fp16]
    const float *sc, *sh;    // [64]
    int n, h, w, ho, wo, tiles_x, tiles_y;
};
constexpr int ST_TH = 8, ST_TW = 32, ST_IH = 2 * ST_TH + 5, ST_IW = 72, ST_PLB = ST_IH * ST_IW * 8;

__global__ __launch_bounds__(256, 2) void stem_kernel(StemArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[2 * ST_PLB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kb5 = lane >> 5;
    const int mt = wave & 1, rg = wave >> 1;                  // channel half, row group (4 rows)
    int g = blockIdx.x;
    const int tx = g % a.tiles_x; g /= a.tiles_x;
    const int ty = g % a.tiles_y;
    const int img = g / a.tiles_y;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const float *xi = a.x + (size_t)img * 3 * a.h * a.w;
    const int xeb = (a.xmax[img] >> 23) & 0xff;
    const int sx = bb_clamp(138 - xeb, -100, 100);
    const float up = __int_as_float((127 + sx) << 23), down = __int_as_float((127 - sx) << 23);

    // weights of this wave's 32 channels: 14 slabs x 2 planes
    f32x4 wv[14][2];
#pragma unroll
    for (int sl = 0; sl < 14; ++sl)
#pragma unroll
        for (int p = 0; p < 2; ++p) wv[sl][p] = *(const f32x4 *)(a.wq + (((mt * 14 + sl) * 2 + p) * 64 + lane) * 16);

    // the patch: rows iy0 ... iy0 + 20, columns ix0 ... ix0 + 68 (+ 3 columns of padding), three channels + a zero
    const size_t plane = (size_t)a.h * a.w;
    static_assert(ST_IH * ST_IW <= 6 * 256, "six staged pixels per thread");
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int e0 = it * 256 + tid, e = min(e0, ST_IH * ST_IW - 1);
        const int py = e / ST_IW, px = e - py * ST_IW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w && px < 69;
        const float *src = xi + (size_t)bb_clamp(iy, 0, a.h - 1) * a.w + bb_clamp(ix, 0, a.w - 1);
        const float v0 = src[0], v1 = src[plane], v2 = src[2 * plane];
        const float s = ok ? up : 0.f;
        const unsigned h01 = bb_pk(v0 * s, v1 * s), h2 = bb_pk(v2 * s, 0.f);
        const unsigned r01 = bb_pk(v0 * s - bb_lo(h01), v1 * s - bb_hi(h01)), r2 = bb_pk(v2 * s - bb_lo(h2), 0.f);
        if (e0 < ST_IH * ST_IW) {
            *(bu2 *)(xs + e * 8) = (bu2){h01, h2};
            *(bu2 *)(xs + ST_PLB + e * 8) = (bu2){r01, r2};
        }
    }
    __syncthreads();

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // B fragment of output row j of this wave: pixel l31, K block kb5 of slab (ky, half): patch row 2 (4 rg + j) + ky,
    // columns 2 l31 + 4 half + 2 kb5 and the next one
    const int bbase = ((2 * (4 * rg)) * ST_IW + 2 * l31 + 2 * kb5) * 8;
#pragma unroll
    for (int sl = 0; sl < 14; ++sl) {
        const int ky = sl >> 1, half = sl & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = bbase + ((2 * j + ky) * ST_IW + 4 * half) * 8;
            const f32x4 b0 = *(const f32x4 *)(xs + off), b1 = *(const f32x4 *)(xs + ST_PLB + off);
            acc[j] = BB_MFMA(wv[sl][1], b0, acc[j]);
            acc[j] = BB_MFMA(wv[sl][0], b1, acc[j]);
            acc[j] = BB_MFMA(wv[sl][0], b0, acc[j]);
        }
    }
    // D: row (r & 3) + 8 (r >> 2) + 4 kb5 = channel of the half, column l31 = pixel
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb5;
        const float scr = a.sc[ch] * down, shr = a.sh[ch];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + 4 * rg + j, ox = ox0 + l31;
            if (oy < a.ho && ox < a.wo)
                a.y[(((size_t)img * 64 + ch) * a.ho + oy) * a.wo + ox] = fmaxf(fmaf(acc[j][r], scr, shr), 0.f);
        }
    }
}

// ---- MaxPool2d(3, stride 2, padding 1) (reference resnet.py:104, 146): NCHW -> NHWC + max per image --------------------
constexpr int PL_TW = 32, PL_ROW = 2 * PL_TW + 3;            // 64 channels x 3 rows x (65 columns + pad: odd stride)

__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, int *ymax,
                                                            int c, int h, int w, int hp, int wp) {
    __shared__ float s[64 * 3 * PL_ROW];
    const int tid = threadIdx.x;
    const int px0 = blockIdx.x * PL_TW, py = blockIdx.y;
    const int img = blockIdx.z / (c >> 6), c0 = (blockIdx.z % (c >> 6)) << 6;
    const float *xi = x + ((size_t)img * c + c0) * h * w;
    // 49 loads per thread in two batches (25 + 24 in flight together: the kernel is bound by HBM latency x loads in flight; seven at a
    // time ran at 1.3 TB/s); clamped addresses, the store is what is conditional
#pragma unroll 1
    for (int b0 = 0; b0 < 49; b0 += 25) {
        float v[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            const int it = min(b0 + k, 48);
            const int e = it * 256 + tid, ec = min(e, 64 * 3 * 65 - 1);
            const int ch = ec / 195, rem = ec - ch * 195, r = rem / 65, col = rem - r * 65;
            const int iy = 2 * py - 1 + r, ix = 2 * px0 - 1 + col;
            v[k] = xi[((size_t)ch * h + bb_clamp(iy, 0, h - 1)) * w + bb_clamp(ix, 0, w - 1)];
        }
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            const int it = b0 + k;
            const int e = it * 256 + tid, ec = min(e, 64 * 3 * 65 - 1);
            const int ch = ec / 195, rem = ec - ch * 195, r = rem / 65, col = rem - r * 65;
            const int iy = 2 * py - 1 + r, ix = 2 * px0 - 1 + col;
            const bool ok = iy >= 0 && iy < h && ix >= 0 && ix < w;
            if (it < 49 && e < 64 * 3 * 65) s[(ch * 3 + r) * PL_ROW + col] = ok ? v[k] : -INFINITY;
        }
    }
    __syncthreads();
    const int ch = tid & 63;
    float vmax = 0.f;
    for (int q = tid >> 6; q < PL_TW; q += 4) {
        if (px0 + q >= wp) break;
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int d = 0; d < 3; ++d) m = fmaxf(m, s[(ch * 3 + r) * PL_ROW + 2 * q + d]);
        y[(((size_t)img * hp + py) * wp + px0 + q) * c + c0 + ch] = m;
        vmax = fmaxf(vmax, fabsf(m));
    }
    if (ymax) {      // one atomic per work-group (the 16 maxima of a batch share one cache line: 38 400 wave-level probes queued on it)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
        __syncthreads();                     // the tile in s is dead
        if ((tid & 63) == 0) s[tid >> 6] = vmax;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            int *dst = ymax + img;
            if (__float_as_int(m) > *(volatile int *)dst) atomicMax(dst, __float_as_int(m));
        }
    }
}

// ---- [n][hw][c] -> [n][c][hw] (the pyramid levels the matching stages read are NCHW) ----------------------------------------
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ x, float *__restrict__ y, int hw, int c) {
    __shared__ float s[64 * 65];
    const int tid = threadIdx.x, lo = tid & 63, hi = tid >> 6;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const float *xi = x + (size_t)blockIdx.z * hw * c;
    float *yi = y + (size_t)blockIdx.z * hw * c;
    for (int i = hi; i < 64; i += 4)
        if (p0 + i < hw && c0 + lo < c) s[i * 65 + lo] = xi[(size_t)(p0 + i) * c + c0 + lo];
    __syncthreads();
    for (int i = hi; i < 64; i += 4)
        if (c0 + i < c && p0 + lo < hw) yi[(size_t)(c0 + i) * hw + p0 + lo] = s[lo * 65 + i];
}

// ---- host side ------------------------------------------------------------------------------------------------
int launch_absmax(const float *x, size_t n, size_t stride, int pairs, int *out, size_t out_stride, hipStream_t stream);      // consensus.hip

struct ConvCfg { int mt, nt, wn, nit; };

static int conv_ck(int stride) { return stride == 1 ? 32 : 16; }      // input channels per staged chunk
static int conv_nit(const ConvCfg &c, int ks, int stride) {
    const int th = 2 * c.mt * (4 / c.wn), ih = (th - 1) * stride + ks, iw = 15 * stride + ks;
    return ceil_div(ih * iw * (conv_ck(stride) / 8), 256);
}

// Tile of a launch: [32 MT (4 / WN) pixels] x [32 NT WN channels].  Measured per layer shape and batch size
// (tools/conv_sweep.py, 60x80 maps = 37.5 tiles of 128 pixels per image): the small tile (64 pixels x 128 channels, three
// slabs of weights in flight, 152 registers = three work-groups per compute unit) wins up to ~500 tiles and for every 1x1
// and stride-2 layer; from there the 128 x 128 tile, and the 128 x 256 tile (24 MFMAs per 4 fragment reads and 8 weight
// loads, but two work-groups per compute unit and one slab in flight) only for launches of several rounds.
// p2p_conv_set_tile forces one per handle (experiments, the tile-independence tests).
static ConvCfg conv_cfg(int co, int ks, int stride, long tiles128, const int *forced) {
    static const ConvCfg cand256[] = {{2, 4, 2, 0}, {1, 4, 2, 0}, {2, 2, 2, 0}, {1, 2, 2, 0}};
    static const ConvCfg cand128[] = {{2, 2, 2, 0}, {1, 2, 2, 0}};
    static const ConvCfg cand64[] = {{1, 2, 1, 0}};
    const ConvCfg *cand = co % 256 == 0 ? cand256 : (co % 128 == 0 ? cand128 : cand64);
    const int ncand = co % 256 == 0 ? 4 : (co % 128 == 0 ? 2 : 1);
    ConvCfg best = cand[ncand - 1];
    if (ks == 3 && stride == 1) {
        if (co % 256 == 0 && tiles128 >= 512) best = tiles128 >= 1024 ? cand256[0] : cand256[2];
        if (co % 256 != 0 && co % 128 == 0 && tiles128 >= 1024) best = cand128[0];
    }
    if (forced && forced[0] > 0)         // (a tile the layer does not have is ignored)
        for (int i = 0; i < ncand; ++i)
            if (cand[i].mt == forced[0] && cand[i].nt == forced[1] && cand[i].wn == forced[2]) best = cand[i];
    best.nit = conv_nit(best, ks, stride);
    return best;
}

}  // namespace p2p

struct p2p_conv {
    unsigned char *wq;
    float *sc, *sh;          // one allocation behind wq
    int ci, co, ks, stride;
    int tile[3];             // forced (mt, nt, wn); 0 = by the launch size
};

using namespace p2p;

extern "C" int p2p_conv_set_tile(p2p_conv *cv, int mt, int nt, int wn) {
    P2P_REQUIRE(cv && mt >= 0 && nt >= 0 && wn >= 0, P2P_EINVAL, "p2p_conv_set_tile: bad argument");
    cv->tile[0] = mt; cv->tile[1] = nt; cv->tile[2] = wn;
    return P2P_OK;
}

extern "C" int p2p_conv_create(const float *weight, const p2p_bn_params *bn, int ci, int co, int ks, int stride, p2p_conv **out) {
    P2P_REQUIRE(weight && bn && out, P2P_EINVAL, "p2p_conv_create: null argument");
    P2P_REQUIRE((ks == 1 || ks == 3) && (stride == 1 || stride == 2) && ci % 32 == 0 && co % 64 == 0, P2P_EUNSUPPORTED,
                "p2p_conv_create: %dx%d stride %d, %d -> %d channels is outside the ResNet34 layers this library covers", ks, ks, stride, ci, co);
    P2P_REQUIRE(stride == 1 || co % 128 == 0, P2P_EUNSUPPORTED, "p2p_conv_create: stride 2 with %d output channels", co);
    const int ck = conv_ck(stride), nchunks = ci / ck, spc = ck / 16, nsl = ks * ks * spc, ntiles = co / 32;
    const size_t wbytes = ((size_t)nchunks * nsl + 3) * ntiles * 2048;      // + the slabs of zeros the prefetch runs into
    std::vector<unsigned char> h(wbytes + 2 * (size_t)co * 4, 0);
    float *sc = (float *)(h.data() + wbytes), *sh = sc + co;
    std::vector<int> sw(co);
    for (int o = 0; o < co; ++o) {
        float mx = 0.f;
        for (int i = 0; i < ci * ks * ks; ++i) mx = std::max(mx, std::fabs(weight[(size_t)o * ci * ks * ks + i]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { std::frexp(mx, &e); e = 12 - e; }
        sw[o] = e;
        const float inv = 1.0f / std::sqrt(bn->running_var[o] + 1e-5f);
        const float s = bn->weight[o] * inv;
        sc[o] = std::ldexp(s, -e);
        sh[o] = bn->bias[o] - bn->running_mean[o] * s;
    }
    {
        for (int c = 0; c < nchunks; ++c)
            for (int tap = 0; tap < ks * ks; ++tap)
                for (int sl = 0; sl < spc; ++sl)
                    for (int j = 0; j < ntiles; ++j) {
                        uint16_t *frag = (uint16_t *)(h.data() + ((((size_t)c * ks * ks + tap) * spc + sl) * ntiles + j) * 2048);
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e8 = 0; e8 < 8; ++e8) {
                                const int o = j * 32 + (lane & 31), i = c * ck + sl * 16 + (lane >> 5) * 8 + e8;
                                const float v = std::ldexp(weight[(((size_t)o * ci + i) * ks + tap / ks) * ks + tap % ks], sw[o]);
                                const _Float16 h0 = (_Float16)v;
                                const _Float16 h1 = (_Float16)(v - (float)h0);
                                frag[lane * 8 + e8] = __builtin_bit_cast(uint16_t, h0);
                                frag[512 + lane * 8 + e8] = __builtin_bit_cast(uint16_t, h1);
                            }
                    }
    }
    p2p_conv *cv = new p2p_conv{};
    cv->ci = ci; cv->co = co; cv->ks = ks; cv->stride = stride;
    if (hipMalloc((void **)&cv->wq, h.size()) != hipSuccess) {
        delete cv;
        set_error("p2p_conv_create: hipMalloc of %zu bytes failed", h.size());
        return P2P_ENOMEM;
    }
    if (hipMemcpy(cv->wq, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(cv->wq);
        delete cv;
        set_error("p2p_conv_create: upload failed");
        return P2P_EHIP;
    }
    cv->sc = (float *)(cv->wq + wbytes);
    cv->sh = cv->sc + co;
    *out = cv;
    return P2P_OK;
}

extern "C" void p2p_conv_destroy(p2p_conv *cv) {
    if (!cv) return;
    (void)hipFree(cv->wq);
    delete cv;
}

template <int MT, int NT, int WN, int NIT, int DB>
static int launch_conv(const ConvArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    static DeviceOnce attr_set;
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    if (!attr_set.done(dev)) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)conv_kernel<MT, NT, WN, NIT, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.set(dev);
    }
    hipLaunchKernelGGL((conv_kernel<MT, NT, WN, NIT, DB>), grid, dim3(256), lds, stream, a);
    return check_launch("conv_kernel");
}

extern "C" int p2p_conv_forward(const p2p_conv *cv, const float *x, const int *xmax, int n, int h, int w, const float *residual,
                                int relu, float *y, int *ymax, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(cv && x && xmax && y, P2P_EINVAL, "p2p_conv_forward: null argument");
    P2P_REQUIRE(n >= 1 && h >= 1 && w >= 1, P2P_EINVAL, "p2p_conv_forward: bad extents %d x %d x %d", n, h, w);
    const int pad = cv->ks / 2;
    const int ho = (h + 2 * pad - cv->ks) / cv->stride + 1, wo = (w + 2 * pad - cv->ks) / cv->stride + 1;
    const ConvCfg c = conv_cfg(cv->co, cv->ks, cv->stride, (long)n * ceil_div(ho, 8) * ceil_div(wo, 16), cv->tile);
    ConvArgs a{};
    a.x = x; a.res = residual; a.y = y; a.xmax = xmax; a.ymax = ymax; a.wq = cv->wq; a.sc = cv->sc; a.sh = cv->sh;
    a.n = n; a.h = h; a.w = w; a.ci = cv->ci; a.co = cv->co; a.ks = cv->ks; a.stride = cv->stride; a.relu = relu; a.ck = conv_ck(cv->stride);
    a.ho = ho; a.wo = wo;
    P2P_REQUIRE((size_t)n * h * w * cv->ci < ((size_t)1 << 31) && (size_t)n * a.ho * a.wo * cv->co < ((size_t)1 << 31), P2P_EUNSUPPORTED,
                "p2p_conv_forward: tensor of more than 2^31 elements");
    const int th = 2 * c.mt * (4 / c.wn);
    a.tiles_x = ceil_div(a.wo, 16); a.tiles_y = ceil_div(a.ho, th);
    const int ih = (th - 1) * cv->stride + cv->ks, iw = 15 * cv->stride + cv->ks;
    const size_t lds = (size_t)2 * 2 * ((ih * iw * (a.ck * 2 + 16) + 15) & ~15);
    P2P_REQUIRE(lds <= 160 * 1024, P2P_EUNSUPPORTED, "p2p_conv_forward: staging tile of %zu bytes", lds);
    const dim3 grid(a.tiles_x * a.tiles_y * n, cv->co / (32 * c.nt * c.wn));
    const bool deep = cv->ks == 3;          // 18 or 9 slabs per chunk: three in flight; 1x1 convolutions have 2 or 1
    if (c.mt == 2 && c.nt == 4 && c.wn == 2 && c.nit <= 3) return launch_conv<2, 4, 2, 3, 1>(a, grid, lds, stream);
    if (c.mt == 1 && c.nt == 4 && c.wn == 2 && c.nit <= 3) return launch_conv<1, 4, 2, 3, 1>(a, grid, lds, stream);
    if (c.mt == 2 && c.nt == 2 && c.wn == 2 && c.nit <= 3)
        return deep ? launch_conv<2, 2, 2, 3, 3>(a, grid, lds, stream) : launch_conv<2, 2, 2, 3, 1>(a, grid, lds, stream);
    if (c.mt == 2 && c.nt == 2 && c.wn == 2 && c.nit <= 5)
        return deep ? launch_conv<2, 2, 2, 5, 3>(a, grid, lds, stream) : launch_conv<2, 2, 2, 5, 1>(a, grid, lds, stream);
    if (c.mt == 1 && c.nt == 2 && c.wn == 2 && c.nit <= 3)
        return deep ? launch_conv<1, 2, 2, 3, 3>(a, grid, lds, stream) : launch_conv<1, 2, 2, 3, 1>(a, grid, lds, stream);
    if (c.mt == 1 && c.nt == 2 && c.wn == 1 && c.nit <= 3)
        return deep ? launch_conv<1, 2, 1, 3, 3>(a, grid, lds, stream) : launch_conv<1, 2, 1, 3, 1>(a, grid, lds, stream);
    set_error("p2p_conv_forward: no kernel instance (%d,%d,%d) with %d staged pieces per thread", c.mt, c.nt, c.wn, c.nit);
    return P2P_EUNSUPPORTED;
}

extern "C" int p2p_absmax_batch(const float *x, size_t count, int items, int *out, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(x && out && items >= 1 && count >= 1, P2P_EINVAL, "p2p_absmax_batch: bad argument");
    P2P_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)items * sizeof(int), stream));
    return launch_absmax(x, count, count, items, out, 1, stream);
}

struct p2p_stem {
    unsigned char *wq;
    float *sc, *sh;
};

extern "C" int p2p_stem_create(const float *weight, const p2p_bn_params *bn, p2p_stem **out) {
    P2P_REQUIRE(weight && bn && out, P2P_EINVAL, "p2p_stem_create: null argument");
    const size_t wbytes = 2 * 14 * 2 * 1024;
    std::vector<unsigned char> h(wbytes + 2 * 64 * 4, 0);
    float *sc = (float *)(h.data() + wbytes), *sh = sc + 64;
    for (int o = 0; o < 64; ++o) {
        float mx = 0.f;
        for (int i = 0; i < 147; ++i) mx = std::max(mx, std::fabs(weight[o * 147 + i]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { std::frexp(mx, &e); e = 12 - e; }
        const float inv = 1.0f / std::sqrt(bn->running_var[o] + 1e-5f);
        const float s = bn->weight[o] * inv;
        sc[o] = std::ldexp(s, -e);
        sh[o] = bn->bias[o] - bn->running_mean[o] * s;
        // A fragment: lane (row = channel o & 31, K block lane >> 5), k = slab * 16 + 8 (lane >> 5) + j = (ky, half, kx pair, c)
        for (int sl = 0; sl < 14; ++sl)
            for (int kb = 0; kb < 2; ++kb)
                for (int j = 0; j < 8; ++j) {
                    const int ky = sl >> 1, kx = 4 * (sl & 1) + 2 * kb + (j >> 2), c = j & 3;
                    float v = 0.f;
                    if (kx < 7 && c < 3) v = std::ldexp(weight[((o * 3 + c) * 7 + ky) * 7 + kx], e);
                    const _Float16 h0 = (_Float16)v;
                    const _Float16 h1 = (_Float16)(v - (float)h0);
                    uint16_t *frag = (uint16_t *)(h.data() + (((size_t)(o >> 5) * 14 + sl) * 2) * 1024);
                    const int lane = kb * 32 + (o & 31);
                    frag[lane * 8 + j] = __builtin_bit_cast(uint16_t, h0);
                    frag[512 + lane * 8 + j] = __builtin_bit_cast(uint16_t, h1);
                }
    }
    p2p_stem *st = new p2p_stem{};
    if (hipMalloc((void **)&st->wq, h.size()) != hipSuccess) {
        delete st;
        set_error("p2p_stem_create: hipMalloc failed");
        return P2P_ENOMEM;
    }
    if (hipMemcpy(st->wq, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(st->wq);
        delete st;
        set_error("p2p_stem_create: upload failed");
        return P2P_EHIP;
    }
    st->sc = (float *)(st->wq + wbytes);
    st->sh = st->sc + 64;
    *out = st;
    return P2P_OK;
}

extern "C" void p2p_stem_destroy(p2p_stem *st) {
    if (!st) return;
    (void)hipFree(st->wq);
    delete st;
}

extern "C" int p2p_stem_forward(const p2p_stem *st, const float *image, const int *imax, int n, int h, int w, float *y, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(st && image && imax && y && n >= 1 && h >= 1 && w >= 1, P2P_EINVAL, "p2p_stem_forward: bad argument");
    StemArgs a{};
    a.x = image; a.y = y; a.xmax = imax; a.wq = st->wq; a.sc = st->sc; a.sh = st->sh; a.n = n; a.h = h; a.w = w;
    a.ho = (h - 1) / 2 + 1; a.wo = (w - 1) / 2 + 1;
    a.tiles_x = ceil_div(a.wo, ST_TW); a.tiles_y = ceil_div(a.ho, ST_TH);
    hipLaunchKernelGGL(stem_kernel, dim3(a.tiles_x * a.tiles_y * n), dim3(256), 0, stream, a);
    return check_launch("stem_kernel");
}

extern "C" int p2p_maxpool_nhwc(const float *x, int n, int c, int h, int w, float *y, int *ymax, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(x && y && n >= 1 && h >= 1 && w >= 1, P2P_EINVAL, "p2p_maxpool_nhwc: bad argument");
    P2P_REQUIRE(c >= 64 && c % 64 == 0, P2P_EUNSUPPORTED, "p2p_maxpool_nhwc: %d channels (multiples of 64 only)", c);
    const int hp = (h - 1) / 2 + 1, wp = (w - 1) / 2 + 1;
    P2P_REQUIRE((long)n * (c / 64) <= 65535 && hp <= 65535, P2P_EUNSUPPORTED, "p2p_maxpool_nhwc: grid too large");
    hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3(ceil_div(wp, PL_TW), hp, n * (c / 64)), dim3(256), 0, stream, x, y, ymax, c, h, w, hp, wp);
    return check_launch("maxpool_nhwc_kernel");
}

extern "C" int p2p_nhwc_to_nchw(const float *x, int n, int h, int w, int c, float *y, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(x && y && n >= 1 && h >= 1 && w >= 1 && c >= 1, P2P_EINVAL, "p2p_nhwc_to_nchw: bad argument");
    P2P_REQUIRE(n <= 65535 && ceil_div(c, 64) <= 65535, P2P_EUNSUPPORTED, "p2p_nhwc_to_nchw: grid too large");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ceil_div(h * w, 64), ceil_div(c, 64), n), dim3(256), 0, stream, x, y, h * w, c);
    return check_launch("nhwc_to_nchw_kernel");
}
