// Coarse stage of the Patch2Pix matching path on gfx950 (reference networks/patch2pix.py:120-136,
// :340-375): L2-normalise -> 4-D correlation -> 4-D max-pool -> mutual matching -> neighbourhood
// consensus (two 3^4 convolutions, symmetric) -> mutual matching -> soft mutual-NN matches.
//
// Data layout in HBM (all fp32, A = image 1 cells, B = image 2 cells):
//   Fn      [pos'][C]     L2-normalised layer-3 features, position-major (K contiguous) and with
//                         positions re-ordered cell-major so that the k^2 positions of one pooling
//                         cell are adjacent: pos' = cell*k^2 + (i%k)*k + (j%k)
//   P, Y    [nA'][nB']    pooled correlation volume viewed as a matrix (row = A cell, col = B cell)
//   delta   [nA'][nB']    uint8 argmax code s = ((di*k+dj)*k+dk)*k+dl
//   H1      [32][nA'][nB'] hidden layer of the consensus net: 16 channels of the direct branch +
//                         16 of the transposed branch (evaluated with A/B-swapped taps, so the
//                         volume is never permuted)
// The full-resolution correlation (92 MB at 480x640, 1.5 GB at 960x1280) is never written: the
// pooling runs on the MFMA accumulators of the correlation GEMM.
//
// Batches: the reference's tensors carry a batch axis ([B,C,h,w] features of B equally sized pairs).  Every
// kernel takes the pair from blockIdx.z and a per-pair stride for each of its pointers, so a batch is
// ONE launch per kernel (B x the work-groups: the 30x40x30x40 volume of a single 480x640 pair does not
// fill 256 CUs in the consensus layers).
#include "p2p_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace p2p {

constexpr float MM_EPS = 1e-5f;

// monotone float <-> int key so that atomicMax(int) implements a float max (order independent)
__device__ __forceinline__ int f2key(float f) {
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }
constexpr int KEY_NEG_INF = (int)0xff800000 ^ 0x7fffffff;

// ------------------------------------------------------------------------------------------------
// 1. L2 normalise + transpose to position-major with cell-major position order (modules.py:6)
// ------------------------------------------------------------------------------------------------
constexpr int PREP_P = 16;    // positions per work-group (300 groups at 60x80: fills the chip)
// PLANES = 3: write the normalised features as three bf16 planes [plane][pos'][C] (exact: v = p0 + p1 + p2) for
// corr_pool_xn_kernel instead of fp32 [pos'][C] (PLANES = 0); PLANES = 2: two fp16 planes of v * 2^12 (|v| <= 1, so both
// planes stay in the normal range of fp16 and v * 2^12 = p0 + p1 to within 2^-24 |v * 2^12|).  The plane stride is hw * C elements.
constexpr float CORR_FP16_SCALE = 4096.0f;
template <int PLANES>
__global__ __launch_bounds__(256) void prep_kernel(const float *__restrict__ F, float *__restrict__ Fn, int C, int h,
                                                   int w, int k, size_t sF, size_t sFn) {
    F += blockIdx.z * sF;
    Fn += blockIdx.z * sFn;
    __shared__ float tile[256 * (PREP_P + 1)];   // [C <= 256][17]
    __shared__ float part[16][PREP_P];
    __shared__ float inv[PREP_P];
    const int hw = h * w;
    const int p0 = blockIdx.x * PREP_P;
    const int tid = threadIdx.x;
    // 4 threads x float4 cover the 16 positions of one channel row (64 contiguous bytes)
    for (int e = tid; e < C * 4; e += 256) {
        const int c = e >> 2, q = e & 3;
        const int pos = p0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pos + 3 < hw && (hw & 3) == 0) v = *(const f32x4 *)(F + (size_t)c * hw + pos);
        else
            for (int i = 0; i < 4; ++i) if (pos + i < hw) v[i] = F[(size_t)c * hw + pos + i];
        float *t = tile + c * (PREP_P + 1) + 4 * q;
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    }
    __syncthreads();
    {   // sum of squares: 16 channel groups x 16 positions, ascending channel order within a group
        const int p = tid & 15, g = tid >> 4;
        float ss = 0.f;
        for (int c = g; c < C; c += 16) { const float v = tile[c * (PREP_P + 1) + p]; ss = fmaf(v, v, ss); }
        part[g][p] = ss;
    }
    __syncthreads();
    if (tid < PREP_P) {
        float ss = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) ss += part[g][tid];
        inv[tid] = 1.0f / sqrtf(ss + 1e-6f);
    }
    __syncthreads();
    const int wc = w / k;
    for (int p = 0; p < PREP_P; ++p) {
        const int pos = p0 + p;
        if (pos >= hw) break;
        const int i = pos / w, j = pos - i * w;
        const int pp = ((i / k) * wc + (j / k)) * (k * k) + (i % k) * k + (j % k);
        for (int c = tid; c < C; c += 256) {
            const float v = tile[c * (PREP_P + 1) + p] * inv[p];
            if (PLANES == 0) {
                Fn[(size_t)pp * C + c] = v;
            } else if (PLANES == 2) {
                unsigned short *d = (unsigned short *)Fn + (size_t)pp * C + c;
                const float x = v * CORR_FP16_SCALE;
                const _Float16 h0 = (_Float16)x;
                d[0] = __builtin_bit_cast(unsigned short, h0);
                d[(size_t)hw * C] = __builtin_bit_cast(unsigned short, (_Float16)(x - (float)h0));
            } else {
                unsigned short *d = (unsigned short *)Fn + (size_t)pp * C + c;
                const size_t pl = (size_t)hw * C;
                const unsigned short p0 = __builtin_bit_cast(unsigned short, (__bf16)v);
                const float r1 = v - __uint_as_float((unsigned)p0 << 16);
                const unsigned short p1 = __builtin_bit_cast(unsigned short, (__bf16)r1);
                const float r2 = r1 - __uint_as_float((unsigned)p1 << 16);
                d[0] = p0;
                d[pl] = p1;
                d[2 * pl] = __builtin_bit_cast(unsigned short, (__bf16)r2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2. correlation GEMM (modules.py:41-53) with the 4-D max-pool (modules.py:11-34) in the epilogue
//    C[pA'][pB'] = sum_c FnA[pA'][c] * FnB[pB'][c];  128x128 tile, 4 waves x (2x2) 32x32x2 MFMA
// ------------------------------------------------------------------------------------------------
constexpr int CT = 128;      // tile edge
constexpr int CBK = 32;      // K per stage
constexpr int CLD = 36;      // LDS row stride (floats): 16-B aligned

template <int KS>
__global__ __launch_bounds__(256) void corr_pool_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                        int nA, int nB, int C, float *__restrict__ P,
                                                        uint8_t *__restrict__ delta, size_t sAB, size_t sP,
                                                        size_t sDelta) {
    A += blockIdx.z * sAB;
    B += blockIdx.z * sAB;
    P += blockIdx.z * sP;
    if (delta) delta += blockIdx.z * sDelta;
    __shared__ __attribute__((aligned(16))) float As[CT * CLD];
    __shared__ __attribute__((aligned(16))) float Bs[CT * CLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int rowA0 = blockIdx.y * CT, rowB0 = blockIdx.x * CT;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    const int lr = tid >> 3, lk = (tid & 7) * 4;    // loader: row lr (+32*i), floats lk..lk+3
    for (int k0 = 0; k0 < C; k0 += CBK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 32 * i;
            const int ra = min(rowA0 + r, nA - 1), rb = min(rowB0 + r, nB - 1);
            *(f32x4 *)(As + r * CLD + lk) = *(const f32x4 *)(A + (size_t)ra * C + k0 + lk);
            *(f32x4 *)(Bs + r * CLD + lk) = *(const f32x4 *)(B + (size_t)rb * C + k0 + lk);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < CBK; kk += 8) {
            // lane half h supplies k = kk + 4h + q for k-step q (same K permutation on both operands)
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *(const f32x4 *)(As + (wr * 64 + i * 32 + l31) * CLD + kk + 4 * half);
                b[i] = *(const f32x4 *)(Bs + (wc * 64 + i * 32 + l31) * CLD + kk + 4 * half);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // accumulator element r of lane: row = (r&3) + 8*(r>>2) + 4*half, col = l31
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tr = rowA0 + wr * 64 + i * 32, tc = rowB0 + wc * 64 + j * 32;
            if (KS == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = tr + (r & 3) + 8 * (r >> 2) + 4 * half, col = tc + l31;
                    if (row < nA && col < nB) P[(size_t)row * nB + col] = acc[i][j][r];
                }
            } else {
                // k = 2: a pooling cell is a 4x4 block: rows = (i,j) of A in regs 4g..4g+3, cols = (k,l)
                // of B in 4 adjacent lanes.  First maximum in the order s = row_in_cell*4 + col_in_cell.
                const int nAc = nA >> 2, nBc = nB >> 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float best = acc[i][j][4 * g];
                    int s = 0;
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        const float v = acc[i][j][4 * g + r];
                        if (v > best) { best = v; s = r; }
                    }
                    s = s * 4 + (lane & 3);
#pragma unroll
                    for (int m = 1; m <= 2; m <<= 1) {
                        const float ov = __shfl_xor(best, m);
                        const int os = __shfl_xor(s, m);
                        if (ov > best || (ov == best && os < s)) { best = ov; s = os; }
                    }
                    if ((lane & 3) == 0) {
                        const int crow = (tr >> 2) + 2 * g + half, ccol = (tc + l31) >> 2;
                        if (crow < nAc && ccol < nBc) {
                            P[(size_t)crow * nBc + ccol] = best;
                            if (delta) delta[(size_t)crow * nBc + ccol] = (uint8_t)s;
                        }
                    }
                }
            }
        }
}

// The same GEMM + pooling epilogue in fp32-equivalent arithmetic on the 16-bit matrix cores: both operands arrive as NPL
// planes (prep_kernel<NPL>).  NPL = 2 (P2P_CORR_MODE=fp16x2, the default): two fp16 planes of the features times 2^12,
// three v_mfma_f32_32x32x16_f16 per product (a0 b0 + a0 b1 + a1 b0; the dropped a1 b1 is <= 2^-24 of the product), the
// accumulators are scaled back by 2^-24 (exact) in the epilogue.  NPL = 3 (P2P_CORR_MODE=bf16x3): three bf16 planes, the
// six products of order <= 2 (see regress_x3.hip).  fp32 accumulation, no VALU in the loop: per K = 16 slab and wave
// 4 * NPL ds_read_b128 and 4 * (3 or 6) MFMAs.  The K loop is double buffered: the global loads of stage i + 1 are in
// flight while stage i is multiplied out of LDS, one barrier per stage.
// LDS: [stage 2][A|B][plane][128 rows][32 K 16-bit (+16 B pad)] = 80 KB (fp16x2) / 120 KB (bf16x3).
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cf16x8 __attribute__((ext_vector_type(8)));
constexpr int CX_ROW = 32 * 2 + 16;          // bytes per LDS row: 80 = 5 x 16 (odd multiple of 16 B: the 16 lanes of a ds_read_b128
                                             // service group land on 16 different 16-byte slots of the 256-byte bank row)
constexpr int CX_PLANE = CT * CX_ROW;        // 10240
template <int NPL> __device__ __forceinline__ f32x16 cx_mfma(const f32x4 &a, const f32x4 &b, const f32x16 &c) {
    if (NPL == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cf16x8, a), __builtin_bit_cast(cf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8, a), __builtin_bit_cast(cbf16x8, b), c, 0, 0, 0);
}

template <int KS, int NPL>
__global__ __launch_bounds__(256) void corr_pool_xn_kernel(const unsigned short *__restrict__ A, const unsigned short *__restrict__ B,
                                                           int nA, int nB, int C, float *__restrict__ P,
                                                           uint8_t *__restrict__ delta, size_t sAB, size_t sP, size_t sDelta) {
    constexpr int CX_MAT = NPL * CX_PLANE, CX_STAGE = 2 * CX_MAT;
    A += blockIdx.z * sAB * 2;               // sAB is in 4-byte words
    B += blockIdx.z * sAB * 2;
    P += blockIdx.z * sP;
    if (delta) delta += blockIdx.z * sDelta;
    P2P_DYN_SHARED(unsigned char, cx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int rowA0 = blockIdx.y * CT, rowB0 = blockIdx.x * CT;
    const size_t plA = (size_t)nA * C, plB = (size_t)nB * C;       // plane strides in elements

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    // loader: per plane and matrix 128 rows x 64 B = 512 16-byte pieces: thread -> pieces tid and tid + 256
    const int lrow = tid >> 2, lq = tid & 3;
    f32x4 va[NPL][2], vb[NPL][2];
    auto load = [&](int k0) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = lrow + 64 * i;
                const int ra = min(rowA0 + r, nA - 1), rb = min(rowB0 + r, nB - 1);
                va[pl][i] = *(const f32x4 *)(A + pl * plA + (size_t)ra * C + k0 + lq * 8);
                vb[pl][i] = *(const f32x4 *)(B + pl * plB + (size_t)rb * C + k0 + lq * 8);
            }
    };
    auto store = [&](unsigned char *st) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = lrow + 64 * i;
                *(f32x4 *)(st + pl * CX_PLANE + r * CX_ROW + lq * 16) = va[pl][i];
                *(f32x4 *)(st + CX_MAT + pl * CX_PLANE + r * CX_ROW + lq * 16) = vb[pl][i];
            }
    };
    load(0);
    store(cx);
    __syncthreads();
    const int nk = C / 32;
    for (int it = 0; it < nk; ++it) {
        const unsigned char *st = cx + (it & 1) * CX_STAGE;
        if (it + 1 < nk) load((it + 1) * 32);          // in flight while this stage is multiplied
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {     // two slabs of 16 K
            f32x4 a[2][NPL], b[2][NPL];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    a[i][pl] = *(const f32x4 *)(st + pl * CX_PLANE + (wr * 64 + i * 32 + l31) * CX_ROW + kk * 32 + half * 16);
                    b[i][pl] = *(const f32x4 *)(st + CX_MAT + pl * CX_PLANE + (wc * 64 + i * 32 + l31) * CX_ROW + kk * 32 + half * 16);
                }
            // smallest terms first; the four accumulators rotate
            constexpr int NT = (NPL == 3) ? 6 : 3;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int pa, pb;
                if (NPL == 3) {
                    pa = (t == 0) ? 2 : (t == 1 || t == 3) ? 1 : 0;
                    pb = (t == 0 || t == 3 || t == 5) ? 0 : (t == 1 || t == 4) ? 1 : 2;
                } else {
                    pa = (t == 0) ? 1 : 0;
                    pb = (t == 1) ? 1 : 0;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = cx_mfma<NPL>(a[i][pa], b[j][pb], acc[i][j]);
            }
        }
        if (it + 1 < nk) store(cx + ((it + 1) & 1) * CX_STAGE);     // the other stage: everybody left it at the last barrier
        __syncthreads();
    }
    if (NPL == 2) {                          // the planes carried 2^12 each
        constexpr float inv = 1.0f / (CORR_FP16_SCALE * CORR_FP16_SCALE);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
    }

    // accumulator element r of lane: row = (r&3) + 8*(r>>2) + 4*half, col = l31 (the epilogue of corr_pool_kernel)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tr = rowA0 + wr * 64 + i * 32, tc = rowB0 + wc * 64 + j * 32;
            if (KS == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = tr + (r & 3) + 8 * (r >> 2) + 4 * half, col = tc + l31;
                    if (row < nA && col < nB) P[(size_t)row * nB + col] = acc[i][j][r];
                }
            } else {
                const int nAc = nA >> 2, nBc = nB >> 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float best = acc[i][j][4 * g];
                    int s = 0;
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        const float v = acc[i][j][4 * g + r];
                        if (v > best) { best = v; s = r; }
                    }
                    s = s * 4 + (lane & 3);
#pragma unroll
                    for (int m = 1; m <= 2; m <<= 1) {
                        const float ov = __shfl_xor(best, m);
                        const int os = __shfl_xor(s, m);
                        if (ov > best || (ov == best && os < s)) { best = ov; s = os; }
                    }
                    if ((lane & 3) == 0) {
                        const int crow = (tr >> 2) + 2 * g + half, ccol = (tc + l31) >> 2;
                        if (crow < nAc && ccol < nBc) {
                            P[(size_t)crow * nBc + ccol] = best;
                            if (delta) delta[(size_t)crow * nBc + ccol] = (uint8_t)s;
                        }
                    }
                }
            }
        }
}

// arithmetic of the correlation GEMM: number of 16-bit planes per operand -- 2 = fp16x2 (default, fp32-equivalent), 3 = bf16x3
// (fp32-equivalent, P2P_CORR_MODE=bf16x3), 0 = the exact fp32 MFMA (P2P_CORR_MODE=f32)
static int corr_planes() {
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("P2P_CORR_MODE");
        mode = (e && !strcmp(e, "f32")) ? 0 : (e && !strcmp(e, "bf16x3")) ? 3 : 2;
    }
    return mode;
}

// ------------------------------------------------------------------------------------------------
// 3. row / column maxima of an [nA][nB] matrix (the two torch.max of ncn/model.py:165-166)
// ------------------------------------------------------------------------------------------------
// n keys = -inf, followed by one word = 0 (float bits of max |X| after the first mutual matching, see mm_apply_kernel)
__global__ void fill_keys_kernel(int *p, int n, size_t sKeys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) p[blockIdx.z * sKeys + i] = (i < n) ? KEY_NEG_INF : 0;
}

// column maxima: a thread owns one column over a 64-row chunk; chunks meet in an (order independent) atomicMax
// (X2: optional second addend of every value -- the fused consensus kernel leaves its two branches in two arrays)
__global__ __launch_bounds__(256) void colmax_kernel(const float *__restrict__ X, int nA, int nB, int *ckey, size_t sX,
                                                     size_t sKeys, const float *__restrict__ X2) {
    X += blockIdx.z * sX;
    if (X2) X2 += blockIdx.z * sX;
    ckey += blockIdx.z * sKeys;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= nB) return;
    const int r0 = blockIdx.y * 64, r1 = min(r0 + 64, nA);
    float cm = -INFINITY;
    if (X2) {
#pragma unroll 8
        for (int r = r0; r < r1; ++r) cm = fmaxf(cm, X[(size_t)r * nB + col] + X2[(size_t)r * nB + col]);
    } else {
#pragma unroll 8
        for (int r = r0; r < r1; ++r) cm = fmaxf(cm, X[(size_t)r * nB + col]);
    }
    atomicMax(&ckey[col], f2key(cm));
}

// row maxima: one wave per row
__global__ __launch_bounds__(256) void rowmax_kernel(const float *__restrict__ X, int nA, int nB, int *rkey, size_t sX,
                                                     size_t sKeys, const float *__restrict__ X2) {
    X += blockIdx.z * sX;
    if (X2) X2 += blockIdx.z * sX;
    rkey += blockIdx.z * sKeys;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nA) return;
    const float *x = X + (size_t)row * nB;
    float rm = -INFINITY;
    if (X2) {
        const float *x2 = X2 + (size_t)row * nB;
        for (int c = lane; c < nB; c += 64) rm = fmaxf(rm, x[c] + x2[c]);
    } else {
        for (int c = lane; c < nB; c += 64) rm = fmaxf(rm, x[c]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) rm = fmaxf(rm, __shfl_xor(rm, m));
    if (lane == 0) rkey[row] = f2key(rm);
}

// MutualMatching value (ncn/model.py:168-175): x * ((x / (max_over_B + eps)) * (x / (max_over_A + eps)))
__device__ __forceinline__ float mm_value(float x, float max_over_b, float max_over_a) {
    const float xa = x / (max_over_b + MM_EPS);
    const float xb = x / (max_over_a + MM_EPS);
    return x * (xa * xb);
}

// (in place when out == X)
__global__ __launch_bounds__(256) void mm_apply_kernel(const float *X, int nA, int nB, const int *__restrict__ rkey,
                                                       const int *__restrict__ ckey, float *out, size_t sX, size_t sKeys,
                                                       size_t sOut, float *__restrict__ zero, int *amax, const float *X2) {
    X += blockIdx.z * sX;
    if (X2) X2 += blockIdx.z * sX;
    rkey += blockIdx.z * sKeys;
    ckey += blockIdx.z * sKeys;
    out += blockIdx.z * sOut;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (i < (size_t)nA * nB) {
        const int r = (int)(i / nB), c = (int)(i - (size_t)r * nB);
        v = mm_value(X2 ? X[i] + X2[i] : X[i], key2f(rkey[r]), key2f(ckey[c]));
        out[i] = v;
        if (zero) zero[blockIdx.z * sX + i] = 0.f;      // same index space: clears the accumulation target of the consensus layers
    }
    if (amax) {                                     // largest magnitude of the volume (the fused consensus kernel scales its fp16 planes by it)
        float m = fabsf(v);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        // the word only grows: a (possibly stale) plain read first keeps nearly every wave off the atomic unit
        int *dst = amax + blockIdx.z * sKeys;
        if ((threadIdx.x & 63) == 0 && __float_as_int(m) > *(volatile int *)dst) atomicMax(dst, __float_as_int(m));
    }
}

// ------------------------------------------------------------------------------------------------
// 4. neighbourhood consensus (ncn/model.py:145-155; conv4d.py:12-74), 1 -> 16 -> 1 channels, both
//    symmetric branches.
// ------------------------------------------------------------------------------------------------
struct Vol { int d0, d1, d2, d3; };

// layer 1: X (mutual matching already applied) -> H1[32][nA][nB], bias + ReLU.
// Work-group: one a, L1_TB rows b, L1_Q consecutive B cells q = c*d3 + d.  A wave therefore stores runs of 256
// contiguous bytes per hidden channel (the 184 MB of H1 per 480x640 pair is the kernel's real cost), for any
// d3.  The input halo is staged as whole (c) rows: [3 a][L1_TB+2 b][nrc c][d3+2] floats.
constexpr int L1_TB = 4, L1_R = 2, L1_Q = 64 * L1_R;      // a thread owns L1_R cells 64 apart: one weight fetch feeds both
static int l1_rows(int d3) { return (L1_Q - 2) / d3 + 2 + 2; }      // c-rows L1_Q consecutive cells can touch, + halo

__global__ __launch_bounds__(256) void nc_layer1_kernel(const float *__restrict__ X, Vol v, const float *__restrict__ w1cat,
                                                        const float *__restrict__ b1cat, float *__restrict__ H1, size_t sWs) {
    P2P_DYN_SHARED(float, tile1);
    X += blockIdx.z * sWs;
    H1 += blockIdx.z * sWs;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nB = v.d2 * v.d3;
    const size_t nAB = (size_t)v.d0 * v.d1 * nB;
    const int nq = (nB + L1_Q - 1) / L1_Q, nbt = (v.d1 + L1_TB - 1) / L1_TB;
    int g = blockIdx.x;
    const int q0 = (g % nq) * L1_Q; g /= nq;
    const int b0 = (g % nbt) * L1_TB; g /= nbt;
    const int a = g;
    const int c_lo = q0 / v.d3 - 1, c_hi = min(q0 + L1_Q - 1, nB - 1) / v.d3 + 1;
    const int nrc = c_hi - c_lo + 1, W = v.d3 + 2;
    constexpr int HB = L1_TB + 2;
    // stage: one wave instruction per (a, b, c) row, lane = column (d = column - 1)
    for (int r = wave; r < 3 * HB * nrc; r += 4) {
        const int rc = r % nrc, db = (r / nrc) % HB, da = r / (nrc * HB);
        const int ia = a + da - 1, ib = b0 + db - 1, ic = c_lo + rc;
        const bool rok = ia >= 0 && ia < v.d0 && ib >= 0 && ib < v.d1 && ic >= 0 && ic < v.d2;
        const float *src = X + ((size_t)(ia * v.d1 + ib) * v.d2 + ic) * v.d3 - 1;
        for (int col = lane; col < W; col += 64)
            tile1[r * W + col] = (rok && col >= 1 && col <= v.d3) ? src[col] : 0.f;
    }
    __syncthreads();
    const int tb = wave, ib = b0 + tb;
    int cb[L1_R], boff[L1_R];
    bool active[L1_R];
#pragma unroll
    for (int h = 0; h < L1_R; ++h) {
        cb[h] = q0 + 64 * h + lane;
        const int c = cb[h] / v.d3, d = cb[h] - c * v.d3;
        active[h] = cb[h] < nB && ib < v.d1;
        boff[h] = active[h] ? (tb * nrc + (c - c_lo - 1)) * W + d : 0;      // inactive cells read cell 0 and store nothing
    }
    float acc[L1_R][32];
#pragma unroll
    for (int h = 0; h < L1_R; ++h)
#pragma unroll
        for (int o = 0; o < 32; ++o) acc[h][o] = 0.f;
#ifdef P2P_NC1_SKIP_COMPUTE
    for (int da = 0; da < (v.d0 < 0 ? 3 : 0); ++da)
#else
    for (int da = 0; da < 3; ++da)
#endif
        for (int db = 0; db < 3; ++db) {
            const float *tp = tile1 + (da * HB + db) * nrc * W;
            const float *wp = w1cat + (da * 3 + db) * 9 * 32;
#pragma unroll
            for (int dc = 0; dc < 3; ++dc)
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    float x[L1_R];
#pragma unroll
                    for (int h = 0; h < L1_R; ++h) x[h] = tp[boff[h] + dc * W + dd];
#pragma unroll
                    for (int o = 0; o < 32; ++o) {
                        const float w = wp[(dc * 3 + dd) * 32 + o];
#pragma unroll
                        for (int h = 0; h < L1_R; ++h) acc[h][o] = fmaf(x[h], w, acc[h][o]);
                    }
                }
        }
#pragma unroll
    for (int h = 0; h < L1_R; ++h)
        if (active[h]) {
            const size_t pos = (size_t)(a * v.d1 + ib) * nB + cb[h];
#pragma unroll
            for (int o = 0; o < 32; ++o) H1[o * nAB + pos] = fmaxf(acc[h][o] + b1cat[o], 0.f);
        }
}

// layer 2: Y = relu(b2 + sum_{c<16} W2*H1[c]) + relu(b2 + sum_{c<16} W2^T*H1[16+c])
//
// Work-group tile: ta x tb x tc x (8*tdr) outputs.  A thread owns a run of 8 consecutive outputs along the last
// axis at one (b, c) and MARCHES along the first axis: the a-slices of the hidden volume are staged one at a
// time (one channel per stage, (tb+2) x (tc+2) rows with halo, double buffered), and every 10-float LDS row
// read feeds the three output slices it contributes to (taps da = 0,1,2 -> outputs a+1, a, a-1): 72 FMAs per
// row read instead of 24, which is what takes the kernel off the LDS pipe.  Row stride 8*tdr+12 floats keeps
// ds_read_b128 16-B aligned and bank-conflict free for the rows-fastest thread order.
// The two symmetric branches only meet in the final sum of their ReLUs, so blockIdx.y = branch: each group
// convolves 16 channels and adds relu(b2 + sum) into Y with a hardware float atomic.  Y is zero beforehand and
// gets exactly two addends per cell, so the result does not depend on which branch arrives first.
struct NcTile { int tb, tc, tdr, rs, ta, nthreads; };
constexpr int NC_MAX_ITERS = 6;      // FULLROW staging: 16-byte loads per thread and stage
constexpr int NC_MAX_ROWS = 24;      // general staging: rows per wave and stage

// The nine staged rows (db, dc) around this thread's run, each applied to the three output slices it feeds.
// ALL = every slice is wanted (interior of the march): straight-line code, the 81 scalar weight loads of the
// channel are free to run ahead of their use.  Otherwise the unwanted slices are skipped with uniform branches.
template <bool ALL>
__device__ __forceinline__ void nc2_rows(const float *cb, const float *__restrict__ wch, int hc, int rs, bool useP, bool useC,
                                         bool useN, float (&aP)[8], float (&aC)[8], float (&aN)[8]) {
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
            const float *p = cb + (db * hc + dc) * rs;
            // three 16-byte reads (the last two floats of the third are not used): ds_read_b128 is conflict-free for this
            // thread order, while 4- or 8-byte reads of columns 8, 9 hit only 8 of the 32 banks per half wave (measured:
            // 55 % of the kernel's LDS cycles were bank conflicts)
            const f32x4 x0 = *(const f32x4 *)p, x1 = *(const f32x4 *)(p + 4), x2 = *(const f32x4 *)(p + 8);
            const float x[10] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1]};
            float w[9];                                  // [da][dd]
#pragma unroll
            for (int q = 0; q < 9; ++q) w[q] = wch[(db * 3 + dc) * 9 + q];
            if (ALL || useN) {
#pragma unroll
                for (int i = 0; i < 8; ++i) aN[i] = fmaf(x[i + 2], w[2], fmaf(x[i + 1], w[1], fmaf(x[i], w[0], aN[i])));
            }
            if (ALL || useC) {
#pragma unroll
                for (int i = 0; i < 8; ++i) aC[i] = fmaf(x[i + 2], w[5], fmaf(x[i + 1], w[4], fmaf(x[i], w[3], aC[i])));
            }
            if (ALL || useP) {
#pragma unroll
                for (int i = 0; i < 8; ++i) aP[i] = fmaf(x[i + 2], w[8], fmaf(x[i + 1], w[7], fmaf(x[i], w[6], aP[i])));
            }
        }
}

// FULLROW: the tile spans the whole last axis (one d-tile, d3 % 4 == 0): rows are staged with 16-byte
// loads, several rows per wave instruction.  Otherwise one wave instruction stages one row (+ halo columns).
template <bool FULLROW>
__global__ __launch_bounds__(256) void nc_layer2_kernel(const float *__restrict__ H1, Vol v, NcTile t,
                                                        const float *__restrict__ w2m, float b2,
                                                        float *__restrict__ Y, size_t sWs) {
    P2P_DYN_SHARED(float, tile2);      // [2][(tb+2)*(tc+2)][rs], then int rowoff[]
    H1 += blockIdx.z * sWs + (size_t)blockIdx.y * 16 * ((size_t)v.d0 * v.d1 * v.d2 * v.d3);      // this branch's 16 channels
    Y += blockIdx.z * sWs;
    w2m += blockIdx.y * 16 * 81;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int td = 8 * t.tdr;
    const int nd = (v.d3 + td - 1) / td, nc = (v.d2 + t.tc - 1) / t.tc, nb = (v.d1 + t.tb - 1) / t.tb;
    int g = blockIdx.x;
    const int d0 = (g % nd) * td; g /= nd;
    const int c0 = (g % nc) * t.tc; g /= nc;
    const int b0 = (g % nb) * t.tb; g /= nb;
    const int a0 = g * t.ta, a1 = min(a0 + t.ta, v.d0);       // this group's output slices [a0, a1)
    const size_t slice = (size_t)v.d1 * v.d2 * v.d3;
    const size_t nAB = (size_t)v.d0 * slice;
    const int hc = t.tc + 2, nrows = (t.tb + 2) * hc, ncol = td + 2;
    float *buf0 = tile2, *buf1 = tile2 + nrows * t.rs;
    int *rowoff = (int *)(tile2 + 2 * nrows * t.rs);
    // this thread's run of 8 outputs
    const int rc = tid % t.tc, rr = (tid / t.tc) % t.tdr, rb = tid / (t.tdr * t.tc);   // rows fastest: no LDS bank conflicts
    const bool active = rb < t.tb;
    const int myoff = (rb * hc + rc) * t.rs + 8 * rr;

    // source offset of every staged row inside an a-slice (identical for all slices and channels)
    for (int r = tid; r < nrows; r += blockDim.x) {
        const int dc = r % hc, db = r / hc;
        const int ib = b0 + db - 1, ic = c0 + dc - 1;
        rowoff[r] = (ib >= 0 && ib < v.d1 && ic >= 0 && ic < v.d2) ? (ib * v.d2 + ic) * v.d3 : -1;
        if (FULLROW) {      // the two halo columns are outside the volume: zero once, never overwritten
            buf0[r * t.rs] = 0.f; buf0[r * t.rs + v.d3 + 1] = 0.f;
            buf1[r * t.rs] = 0.f; buf1[r * t.rs + v.d3 + 1] = 0.f;
        }
    }
    // staging geometry
    const int lpr = max(v.d3 >> 2, 1), rpi = 64 / lpr;            // FULLROW: lanes per row, rows per wave instruction
    const int myr = lane / lpr, myq = lane - myr * lpr;
    const bool lane_ok = myr < rpi;
    const int niter = (nrows + nwaves * rpi - 1) / (nwaves * rpi);
    const int colid = d0 + lane - 1;                              // general: lane = column
    const bool colok = lane < ncol && colid >= 0 && colid < v.d3;

    // One stage = one a-slice of one hidden channel: fetched into registers while the previous stage is being
    // convolved out of LDS, then written to the other LDS buffer.
    f32x4 vq[NC_MAX_ITERS];
    float vs[NC_MAX_ROWS];
    auto fetch = [&](int s, int ch) {
        const float *src = H1 + (size_t)ch * nAB + (size_t)s * slice;
        if (FULLROW) {
#pragma unroll
            for (int i = 0; i < NC_MAX_ITERS; ++i) {
                const int r = (i * nwaves + wave) * rpi + myr;
                const int off = (i < niter && lane_ok && r < nrows) ? rowoff[r] : -1;
                vq[i] = (off >= 0) ? *(const f32x4 *)(src + off + 4 * myq) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int u = 0; u < NC_MAX_ROWS; ++u) {
                const int r = wave + nwaves * u;
                const int off = (r < nrows) ? rowoff[r] : -1;
                vs[u] = (off >= 0 && colok) ? src[off + colid] : 0.f;
            }
        }
    };
    auto commit = [&](float *buf) {
        if (FULLROW) {
#pragma unroll
            for (int i = 0; i < NC_MAX_ITERS; ++i) {
                const int r = (i * nwaves + wave) * rpi + myr;
                if (i < niter && lane_ok && r < nrows) {
                    float *dst = buf + r * t.rs + 1 + 4 * myq;
                    dst[0] = vq[i][0]; dst[1] = vq[i][1]; dst[2] = vq[i][2]; dst[3] = vq[i][3];
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NC_MAX_ROWS; ++u) {
                const int r = wave + nwaves * u;
                if (r < nrows && lane < ncol) buf[r * t.rs + lane] = vs[u];
            }
        }
    };

    // partial sums of the three output slices a staged slice s contributes to: P -> out[s-1], C -> out[s],
    // N -> out[s+1]; [run position]
    float accP[8], accC[8], accN[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) accP[i] = accC[i] = accN[i] = 0.f;

    const int s_first = max(a0 - 1, 0), s_last = min(a1, v.d0 - 1);    // slices outside the volume are all zero
    __syncthreads();        // rowoff / halo zeros are ready
    fetch(s_first, 0);
    commit(buf0);
    __syncthreads();
    int cur = 0;
    for (int s = a0 - 1; s <= a1; ++s) {
        if (s >= s_first && s <= s_last) {
            const bool useP = s - 1 >= a0, useC = s >= a0 && s < a1, useN = s + 1 < a1;
#pragma unroll 1
            for (int ch = 0; ch < 16; ++ch) {
                const bool more = ch < 15 || s < s_last;
#ifdef P2P_NC2_SKIP_STAGE
                if (more && v.d0 < 0) {
#else
                if (more) {
#endif
                    if (ch < 15) fetch(s, ch + 1);
                    else fetch(s + 1, 0);
                }
#ifdef P2P_NC2_SKIP_COMPUTE
                if (active && v.d0 < 0) {
#else
                if (active) {
#endif
                    const float *cb = (cur ? buf1 : buf0) + myoff;
                    const float *wch = w2m + ch * 81;
                    if (useP && useC && useN) nc2_rows<true>(cb, wch, hc, t.rs, true, true, true, accP, accC, accN);
                    else nc2_rows<false>(cb, wch, hc, t.rs, useP, useC, useN, accP, accC, accN);
                }
                if (more) commit(cur ? buf0 : buf1);
                __syncthreads();
                cur ^= 1;
            }
        }
        // every slice that feeds out[s-1] has been seen
        const int a = s - 1;
        if (active && a >= a0) {
            const int ib = b0 + rb, ic = c0 + rc;
            if (ib < v.d1 && ic < v.d2) {
                float *dst = Y + (((size_t)a * v.d1 + ib) * v.d2 + ic) * v.d3;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int id = d0 + 8 * rr + i;
                    if (id < v.d3) unsafeAtomicAdd(dst + id, fmaxf(accP[i] + b2, 0.f));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { accP[i] = accC[i]; accC[i] = accN[i]; accN[i] = 0.f; }
    }
}

static bool nc_fullrow(const Vol &v, const NcTile &t) {
    const int lpr = v.d3 / 4, nrows = (t.tb + 2) * (t.tc + 2), nwaves = t.nthreads / 64;
    return (v.d3 % 4 == 0) && (8 * t.tdr >= v.d3) && lpr >= 1 && lpr <= 64 &&
           ceil_div(nrows, nwaves * (64 / lpr)) <= NC_MAX_ITERS;
}
static size_t nc_lds_bytes(const NcTile &t) {
    const size_t nrows = (size_t)(t.tb + 2) * (t.tc + 2);
    return (2 * nrows * t.rs + nrows) * 4;
}

// Tile shape for this volume and batch.  Model: every configuration does the same useful work, split into
// `waves` wavefronts that each cost ta (FMA) + 0.85 (ta+2) (LDS reads, staging and barriers of the ta+2 slices
// they march through); the chip runs 1024 of them at a time and hides latency poorly below two per SIMD.
static NcTile pick_nc_tile(const Vol &v, int batch) {
    NcTile best{4, 6, 5, 52, 3, 128};
    if (const char *e = getenv("P2P_NC2_TILE")) {      // experiments: "tb,tc,tdr,ta,nthreads"
        NcTile t{};
        if (sscanf(e, "%d,%d,%d,%d,%d", &t.tb, &t.tc, &t.tdr, &t.ta, &t.nthreads) == 5) { t.rs = 8 * t.tdr + 12; return t; }
    }
    double best_cost = 1e300;
    const int tbs[] = {2, 3, 4, 5, 6, 8}, tcs[] = {4, 5, 6, 8, 10, 12, 15, 16}, tdrs[] = {2, 3, 4, 5, 6, 7};   // 8*tdr+2 columns must fit one wave
    const int tas[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 16, 20, 30};
    const int nts[] = {128, 256};
    for (int nthreads : nts)
        for (int tdr : tdrs)
            for (int tb : tbs)
                for (int tc : tcs) {
                    if (tb * tc * tdr > nthreads) continue;
                    NcTile t{tb, tc, tdr, 8 * tdr + 12, 1, nthreads};   // 52 words for 40-wide rows: conflict-free ds_read_b128 (bank model)
                    if (nc_lds_bytes(t) > 40000) continue;             // four groups per CU (160 KiB)
                    if (ceil_div((tb + 2) * (tc + 2), nthreads / 64) > NC_MAX_ROWS) continue;
                    for (int ta : tas) {
                        if (ta > v.d0 && ta != 1) continue;
                        const double groups = (double)ceil_div(v.d0, ta) * ceil_div(v.d1, tb) * ceil_div(v.d2, tc) * ceil_div(v.d3, 8 * tdr);
                        const double wps = groups * batch * 2 * (nthreads / 64) / 1024.0;      // waves per SIMD (two branches)
                        const double cost = (ta + 0.85 * (ta + 2)) * (wps > 1 ? wps : 1.0) * (wps < 2 ? 1.25 : 1.0);
                        if (cost < best_cost) { best_cost = cost; best = t; best.ta = ta; }
                    }
                }
    return best;
}

// ------------------------------------------------------------------------------------------------
// 5. matches (ncn/extract_ncmatches.py:6-94 twice; patch2pix.py:340-375)
// ------------------------------------------------------------------------------------------------
struct MatchArgs {
    const float *X;
    const uint8_t *delta;
    int hA, wA, hB, wB, ksize, upsample, center;
    long long *matches;
    float *scores;
    size_t sX, sM;      // per-pair strides: cells of the volume, rows of the match list
};
__device__ __forceinline__ MatchArgs match_args_of_pair(MatchArgs m, size_t z) {
    m.X += z * m.sX;
    if (m.delta) m.delta += z * m.sX;
    m.matches += z * m.sM * 4;
    m.scores += z * m.sM;
    return m;
}

__device__ __forceinline__ void emit_match(const MatchArgs &m, int out_row, int ra, int cb, float sum_exp) {
    int ia = ra / m.wA, ja = ra - ia * m.wA, ib = cb / m.wB, jb = cb - ib * m.wB;
    if (m.ksize > 1 && m.delta) {
        const int k = m.ksize;
        int s = m.delta[(size_t)ra * (m.hB * m.wB) + cb];
        const int dl = s % k; s /= k;
        const int dk = s % k; s /= k;
        const int dj = s % k; s /= k;
        ia = ia * k + s; ja = ja * k + dj; ib = ib * k + dk; jb = jb * k + dl;
    } else if (m.ksize > 1) {
        ia *= m.ksize; ja *= m.ksize; ib *= m.ksize; jb *= m.ksize;
    }
    const long long up = m.upsample, off = m.center ? m.upsample / 2 : 0;
    long long *o = m.matches + (size_t)out_row * 4;
    o[0] = up * ja + off; o[1] = up * ia + off; o[2] = up * jb + off; o[3] = up * ib + off;
    m.scores[out_row] = 1.0f / sum_exp;       // max of softmax = exp(0) / sum exp(x - max)
}

// direction B->A: one block per 8 columns, 32 interleaved row slices
__global__ __launch_bounds__(256) void match_cols_kernel(MatchArgs m_) {
    const MatchArgs m = match_args_of_pair(m_, blockIdx.z);
    __shared__ float smax[32][8];
    __shared__ int sarg[32][8];
    __shared__ float ssum[32][8];
    const int nA = m.hA * m.wA, nB = m.hB * m.wB;
    const int cs = threadIdx.x & 7, rs = threadIdx.x >> 3;
    const int col = blockIdx.x * 8 + cs;
    const bool ok = col < nB;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    if (ok)
        for (int r = rs; r < nA; r += 32) {
            const float v = m.X[(size_t)r * nB + col];
            if (v > best) { best = v; arg = r; }
        }
    smax[rs][cs] = best; sarg[rs][cs] = arg;
    __syncthreads();
    float gb = smax[0][cs];
    int ga = sarg[0][cs];
#pragma unroll
    for (int s = 1; s < 32; ++s) {
        const float v = smax[s][cs];
        const int a = sarg[s][cs];
        if (v > gb || (v == gb && a < ga)) { gb = v; ga = a; }
    }
    float sum = 0.f;
    if (ok)
        for (int r = rs; r < nA; r += 32) sum += expf(m.X[(size_t)r * nB + col] - gb);
    ssum[rs][cs] = sum;
    __syncthreads();
    if (rs == 0 && ok) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) t += ssum[s][cs];
        emit_match(m, col, ga, col, t);
    }
}

// direction A->B: one wave per row
__global__ __launch_bounds__(256) void match_rows_kernel(MatchArgs m_) {
    const MatchArgs m = match_args_of_pair(m_, blockIdx.z);
    const int nA = m.hA * m.wA, nB = m.hB * m.wB;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nA) return;
    const float *x = m.X + (size_t)row * nB;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < nB; c += 64) {
        const float v = x[c];
        if (v > best) { best = v; arg = c; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(best, s);
        const int oa = __shfl_xor(arg, s);
        if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    float sum = 0.f;
    for (int c = lane; c < nB; c += 64) sum += expf(x[c] - best);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s);
    if (lane == 0) emit_match(m, nB + row, row, arg, sum);
}

__global__ void delta_unpack_kernel(const uint8_t *__restrict__ delta, size_t n, int k, long long *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = delta[i];
    out[3 * n + i] = s % k; s /= k;
    out[2 * n + i] = s % k; s /= k;
    out[1 * n + i] = s % k; s /= k;
    out[i] = s;
}

// workspace carve-up shared by the size query and the launcher
struct CoarseWs {
    size_t fnA, fnB, P, Y, H1, keys, total;   // byte offsets
};
static CoarseWs coarse_ws(int C, int hA, int wA, int hB, int wB, int k) {
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t nA = (size_t)hA * wA, nB = (size_t)hB * wB;
    const size_t nAc = nA / (k * k), nBc = nB / (k * k);
    CoarseWs w;
    size_t off = 0;
    w.fnA = off; off += al(nA * C * 6);      // fp32 [pos'][C] or three bf16 planes
    w.fnB = off; off += al(nB * C * 6);
    w.P = off; off += al(nAc * nBc * 4);
    w.Y = off; off += al(nAc * nBc * 4);
    w.H1 = off; off += al(32 * nAc * nBc * 4);
    w.keys = off; off += al((2 * (nAc + nBc) + 1) * 4);      // row / column maxima of both mutual matchings + max |X|
    w.total = off;
    return w;
}

// consensus.hip
void pack_nc_fused(const float *w1, const float *b1, const float *w2, std::vector<unsigned char> &out);
int launch_nc_fused(const float *X, float *Y, float *Y2, size_t stride, int pairs, int d0, int d1, int d2, int d3,
                    const unsigned char *w_dev, float b2, const int *xmax, size_t xmax_stride, hipStream_t stream);
int launch_absmax(const float *x, size_t n, size_t stride, int pairs, int *out, size_t out_stride, hipStream_t stream);

bool nc_fused_fills_chip(int pairs, int d0, int d1, int d2, int d3);

// The two consensus layers inside p2p_coarse_forward: the fused kernel on the fp16 matrix cores (consensus.hip: hidden volume
// in LDS, 50x less HBM traffic) when the launch has enough work-groups to fill the chip -- batches of 480x640 pairs, any
// 960x1280 pair --, otherwise the two fp32 VALU kernels above with the hidden volume in HBM (a single 480x640 pair: 0.48 ms
// against 0.74 ms).  P2P_NC_MODE=fused / valu forces one of them.  Measurements: profiles/r03_ablation_log.txt.
static bool nc_fused(int pairs, int d0, int d1, int d2, int d3) {
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("P2P_NC_MODE");
        mode = (e && !strcmp(e, "fused")) ? 1 : (e && !strcmp(e, "valu")) ? 0 : 2;
    }
    return mode == 2 ? nc_fused_fills_chip(pairs, d0, d1, d2, d3) : mode == 1;
}

}  // namespace p2p

using namespace p2p;

extern "C" int p2p_ncn_create(const float *w1, const float *b1, const float *w2, const float *b2, p2p_ncn **out) {
    P2P_REQUIRE(w1 && b1 && w2 && b2 && out, P2P_EINVAL, "p2p_ncn_create: null argument");
    // stored layout (conv4d.py:119-120): w1s[da][o][ci=0][db][dc][dd], w2s[da][o=0][ci][db][dc][dd]
    std::vector<float> h(81 * 32 + 32 + 32 * 81, 0.f);
    float *w1cat = h.data(), *b1cat = w1cat + 81 * 32, *w2m = b1cat + 32;
    auto W1 = [&](int o, int da, int db, int dc, int dd) { return w1[(((da * 16 + o) * 3 + db) * 3 + dc) * 3 + dd]; };
    auto W2 = [&](int c, int da, int db, int dc, int dd) { return w2[(((da * 16 + c) * 3 + db) * 3 + dc) * 3 + dd]; };
    for (int da = 0; da < 3; ++da)
        for (int db = 0; db < 3; ++db)
            for (int dc = 0; dc < 3; ++dc)
                for (int dd = 0; dd < 3; ++dd) {
                    const int tap = ((da * 3 + db) * 3 + dc) * 3 + dd;
                    for (int o = 0; o < 16; ++o) {
                        w1cat[tap * 32 + o] = W1(o, da, db, dc, dd);
                        w1cat[tap * 32 + 16 + o] = W1(o, dc, dd, da, db);      // transposed branch
                        // layer 2 is consumed row by row of the staged slice: [channel][db][dc][da][dd]
                        const int m = ((db * 3 + dc) * 3 + da) * 3 + dd;
                        w2m[o * 81 + m] = W2(o, da, db, dc, dd);
                        w2m[(16 + o) * 81 + m] = W2(o, dc, dd, da, db);
                    }
                }
    for (int o = 0; o < 16; ++o) b1cat[o] = b1cat[16 + o] = b1[o];
    float *dev = nullptr;
    P2P_HIP_CHECK(hipMalloc(&dev, h.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(dev);
        set_error("hipMemcpy of consensus weights failed: %s", hipGetErrorString(e));
        return P2P_EHIP;
    }
    std::vector<unsigned char> wf;
    pack_nc_fused(w1, b1, w2, wf);
    unsigned char *wfd = nullptr;
    e = hipMalloc(&wfd, wf.size());
    if (e == hipSuccess) e = hipMemcpy(wfd, wf.data(), wf.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(dev);
        if (wfd) (void)hipFree(wfd);
        set_error("upload of the fused consensus weights failed: %s", hipGetErrorString(e));
        return P2P_EHIP;
    }
    p2p_ncn *n = new p2p_ncn();
    n->dev = dev; n->w1cat = dev; n->b1cat = dev + 81 * 32; n->w2m = dev + 81 * 32 + 32; n->b2 = b2[0];
    n->wfused = wfd;
    *out = n;
    return P2P_OK;
}

extern "C" void p2p_ncn_destroy(p2p_ncn *ncn) {
    if (!ncn) return;
    (void)hipFree(ncn->dev);
    (void)hipFree(ncn->wfused);
    delete ncn;
}

extern "C" size_t p2p_coarse_workspace_bytes(int channels, int hA, int wA, int hB, int wB, int ksize) {
    if (channels <= 0 || hA <= 0 || wA <= 0 || hB <= 0 || wB <= 0 || ksize < 1) return 0;
    return coarse_ws(channels, hA, wA, hB, wB, ksize).total;
}

extern "C" int p2p_coarse_forward_batch(const float *featA, const float *featB, int batch, int C, int hA, int wA, int hB,
                                        int wB, int ksize, const p2p_ncn *ncn, float *corr4d_out, uint8_t *delta_out,
                                        void *workspace, size_t workspace_bytes, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(featA && featB && ncn && corr4d_out && workspace, P2P_EINVAL, "p2p_coarse_forward: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535, P2P_EINVAL, "p2p_coarse_forward: batch %d out of range", batch);
    P2P_REQUIRE(ksize == 1 || ksize == 2, P2P_EUNSUPPORTED, "p2p_coarse_forward: ksize %d not supported (1 or 2)", ksize);
    P2P_REQUIRE(C > 0 && C % 32 == 0 && C <= 256, P2P_EUNSUPPORTED, "p2p_coarse_forward: channels %d (multiple of 32, <= 256)", C);
    P2P_REQUIRE(hA > 0 && wA > 0 && hB > 0 && wB > 0 && hA % ksize == 0 && wA % ksize == 0 && hB % ksize == 0 &&
                    wB % ksize == 0, P2P_EINVAL, "p2p_coarse_forward: feature map sizes must be positive multiples of ksize");
    const CoarseWs ws = coarse_ws(C, hA, wA, hB, wB, ksize);
    P2P_REQUIRE(workspace_bytes >= ws.total, P2P_ENOMEM, "p2p_coarse_forward: workspace %zu < %zu bytes (one pair)", workspace_bytes,
                ws.total);
    const int nA = hA * wA, nB = hB * wB, kk = ksize * ksize;
    const int nAc = nA / kk, nBc = nB / kk;
    const size_t nel = (size_t)nAc * nBc;
    const size_t sWs = ws.total / 4;        // every workspace buffer of pair z sits z * ws.total bytes further on
    const int per_launch = (int)std::min<size_t>(batch, workspace_bytes / ws.total);   // pairs the workspace holds at once
    const Vol v{hA / ksize, wA / ksize, hB / ksize, wB / ksize};

    for (int z0 = 0; z0 < batch; z0 += per_launch) {
        const unsigned nz = (unsigned)std::min(per_launch, batch - z0);
        const float *fA = featA + (size_t)z0 * C * nA, *fB = featB + (size_t)z0 * C * nB;
        float *out = corr4d_out + (size_t)z0 * nel;
        uint8_t *dout = delta_out ? delta_out + (size_t)z0 * nel : nullptr;
        char *base = (char *)workspace;
        float *fnA = (float *)(base + ws.fnA), *fnB = (float *)(base + ws.fnB);
        float *P = (float *)(base + ws.P), *Y = (float *)(base + ws.Y), *H1 = (float *)(base + ws.H1);
        int *rkey1 = (int *)(base + ws.keys), *ckey1 = rkey1 + nAc, *rkey2 = ckey1 + nBc, *ckey2 = rkey2 + nAc;

        const dim3 cgrid(ceil_div(nB, CT), ceil_div(nA, CT), nz);
        const int npl = corr_planes();
        if (npl) {
            int dev = 0;
            P2P_HIP_CHECK(hipGetDevice(&dev));
            static bool attr_set_dev[64] = {false};      // per device: a process may drive several GPUs
            const bool attr_set = dev < 64 && attr_set_dev[dev];
            const int lds = 2 * 2 * npl * CX_PLANE;      // two stages of [A|B][plane]
            if (!attr_set) {
                P2P_HIP_CHECK(hipFuncSetAttribute((const void *)corr_pool_xn_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * CX_PLANE));
                P2P_HIP_CHECK(hipFuncSetAttribute((const void *)corr_pool_xn_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * CX_PLANE));
                P2P_HIP_CHECK(hipFuncSetAttribute((const void *)corr_pool_xn_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 12 * CX_PLANE));
                P2P_HIP_CHECK(hipFuncSetAttribute((const void *)corr_pool_xn_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 12 * CX_PLANE));
                if (dev < 64) attr_set_dev[dev] = true;
            }
            const dim3 ga(ceil_div(nA, PREP_P), 1, nz), gb(ceil_div(nB, PREP_P), 1, nz);
            if (npl == 2) {
                hipLaunchKernelGGL(prep_kernel<2>, ga, dim3(256), 0, stream, fA, fnA, C, hA, wA, ksize, (size_t)C * nA, sWs);
                hipLaunchKernelGGL(prep_kernel<2>, gb, dim3(256), 0, stream, fB, fnB, C, hB, wB, ksize, (size_t)C * nB, sWs);
            } else {
                hipLaunchKernelGGL(prep_kernel<3>, ga, dim3(256), 0, stream, fA, fnA, C, hA, wA, ksize, (size_t)C * nA, sWs);
                hipLaunchKernelGGL(prep_kernel<3>, gb, dim3(256), 0, stream, fB, fnB, C, hB, wB, ksize, (size_t)C * nB, sWs);
            }
            const unsigned short *pa = (const unsigned short *)fnA, *pb = (const unsigned short *)fnB;
            uint8_t *dk = (ksize == 1) ? (uint8_t *)nullptr : dout;
            const size_t sd = (ksize == 1) ? (size_t)0 : nel;
            if (ksize == 1 && npl == 2)
                hipLaunchKernelGGL((corr_pool_xn_kernel<1, 2>), cgrid, dim3(256), lds, stream, pa, pb, nA, nB, C, P, dk, sWs, sWs, sd);
            else if (ksize == 1)
                hipLaunchKernelGGL((corr_pool_xn_kernel<1, 3>), cgrid, dim3(256), lds, stream, pa, pb, nA, nB, C, P, dk, sWs, sWs, sd);
            else if (npl == 2)
                hipLaunchKernelGGL((corr_pool_xn_kernel<2, 2>), cgrid, dim3(256), lds, stream, pa, pb, nA, nB, C, P, dk, sWs, sWs, sd);
            else
                hipLaunchKernelGGL((corr_pool_xn_kernel<2, 3>), cgrid, dim3(256), lds, stream, pa, pb, nA, nB, C, P, dk, sWs, sWs, sd);
        } else {
            hipLaunchKernelGGL(prep_kernel<0>, dim3(ceil_div(nA, PREP_P), 1, nz), dim3(256), 0, stream, fA, fnA, C, hA, wA, ksize,
                               (size_t)C * nA, sWs);
            hipLaunchKernelGGL(prep_kernel<0>, dim3(ceil_div(nB, PREP_P), 1, nz), dim3(256), 0, stream, fB, fnB, C, hB, wB, ksize,
                               (size_t)C * nB, sWs);
            if (ksize == 1)
                hipLaunchKernelGGL(corr_pool_kernel<1>, cgrid, dim3(256), 0, stream, fnA, fnB, nA, nB, C, P, (uint8_t *)nullptr, sWs,
                                   sWs, (size_t)0);
            else
                hipLaunchKernelGGL(corr_pool_kernel<2>, cgrid, dim3(256), 0, stream, fnA, fnB, nA, nB, C, P, dout, sWs, sWs, nel);
        }

        const int nkeys = 2 * (nAc + nBc);
        hipLaunchKernelGGL(fill_keys_kernel, dim3(ceil_div(nkeys + 1, 256), 1, nz), dim3(256), 0, stream, rkey1, nkeys, sWs);
        int *xmax = rkey1 + nkeys;
        const dim3 mgrid(ceil_div(nBc, 256), ceil_div(nAc, 64), nz);
        hipLaunchKernelGGL(colmax_kernel, mgrid, dim3(256), 0, stream, P, nAc, nBc, ckey1, sWs, sWs, (const float *)nullptr);
        hipLaunchKernelGGL(rowmax_kernel, dim3(ceil_div(nAc, 4), 1, nz), dim3(256), 0, stream, P, nAc, nBc, rkey1, sWs, sWs, (const float *)nullptr);

        // first mutual matching, in place on the pooled volume (also clears Y for layer 2's atomic adds)
        hipLaunchKernelGGL(mm_apply_kernel, dim3((unsigned)((nel + 255) / 256), 1, nz), dim3(256), 0, stream, P, nAc, nBc, rkey1,
                           ckey1, P, sWs, sWs, sWs, Y, xmax, (const float *)nullptr);
        const float *Y2 = nullptr;          // second addend of the consensus output (fused kernel: the transposed branch)
        if (nc_fused((int)nz, v.d0, v.d1, v.d2, v.d3)) {
            // the branches write relu(.) with plain stores into Y and into the (otherwise unused) head of the H1 region
            const int st = launch_nc_fused(P, Y, H1, sWs, (int)nz, v.d0, v.d1, v.d2, v.d3, ncn->wfused, ncn->b2, xmax, sWs, stream);
            if (st != P2P_OK) return st;
            Y2 = H1;
        } else {
            const int ntiles = v.d0 * ceil_div(v.d1, L1_TB) * ceil_div(nBc, L1_Q);
            const size_t lds1 = (size_t)3 * (L1_TB + 2) * l1_rows(v.d3) * (v.d3 + 2) * 4;
            P2P_REQUIRE(lds1 <= 64 * 1024, P2P_EUNSUPPORTED, "p2p_coarse_forward: pooled width %d too large for the consensus tile", v.d3);
            hipLaunchKernelGGL(nc_layer1_kernel, dim3(ntiles, 1, nz), dim3(256), lds1, stream, P, v, ncn->w1cat, ncn->b1cat, H1, sWs);
            const NcTile nt = pick_nc_tile(v, (int)nz);
            const int ntiles2 = ceil_div(v.d0, nt.ta) * ceil_div(v.d1, nt.tb) * ceil_div(v.d2, nt.tc) * ceil_div(v.d3, 8 * nt.tdr);
            const size_t lds2 = nc_lds_bytes(nt);
            if (nc_fullrow(v, nt))
                hipLaunchKernelGGL(nc_layer2_kernel<true>, dim3(ntiles2, 2, nz), dim3(nt.nthreads), lds2, stream, H1, v, nt, ncn->w2m,
                                   ncn->b2, Y, sWs);
            else
                hipLaunchKernelGGL(nc_layer2_kernel<false>, dim3(ntiles2, 2, nz), dim3(nt.nthreads), lds2, stream, H1, v, nt, ncn->w2m,
                                   ncn->b2, Y, sWs);
        }
        hipLaunchKernelGGL(colmax_kernel, mgrid, dim3(256), 0, stream, Y, nAc, nBc, ckey2, sWs, sWs, Y2);
        hipLaunchKernelGGL(rowmax_kernel, dim3(ceil_div(nAc, 4), 1, nz), dim3(256), 0, stream, Y, nAc, nBc, rkey2, sWs, sWs, Y2);
        hipLaunchKernelGGL(mm_apply_kernel, dim3((unsigned)((nel + 255) / 256), 1, nz), dim3(256), 0, stream, Y, nAc, nBc, rkey2,
                           ckey2, out, sWs, sWs, nel, (float *)nullptr, (int *)nullptr, Y2);
    }
    return check_launch("coarse_forward kernels");
}

extern "C" int p2p_coarse_forward(const float *featA, const float *featB, int C, int hA, int wA, int hB, int wB,
                                  int ksize, const p2p_ncn *ncn, float *corr4d_out, uint8_t *delta_out,
                                  void *workspace, size_t workspace_bytes, p2p_stream_t stream) {
    return p2p_coarse_forward_batch(featA, featB, 1, C, hA, wA, hB, wB, ksize, ncn, corr4d_out, delta_out, workspace,
                                    workspace_bytes, stream);
}

extern "C" int p2p_neigh_consensus_batch(const float *x, int batch, int hA, int wA, int hB, int wB, const p2p_ncn *ncn, float *y_out,
                                         void *workspace, size_t workspace_bytes, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(x && ncn && y_out && workspace, P2P_EINVAL, "p2p_neigh_consensus: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535 && hA > 0 && wA > 0 && hB > 0 && wB > 0, P2P_EINVAL, "p2p_neigh_consensus: bad sizes");
    P2P_REQUIRE(workspace_bytes >= (size_t)batch * sizeof(int), P2P_ENOMEM, "p2p_neigh_consensus: workspace of %zu bytes needed (4 per volume)",
                (size_t)batch * sizeof(int));
    const size_t nel = (size_t)hA * wA * hB * wB;
    int *xmax = (int *)workspace;
    P2P_HIP_CHECK(hipMemsetAsync(y_out, 0, (size_t)batch * nel * sizeof(float), stream));
    P2P_HIP_CHECK(hipMemsetAsync(xmax, 0, (size_t)batch * sizeof(int), stream));
    const int st = launch_absmax(x, nel, nel, batch, xmax, 1, stream);
    if (st != P2P_OK) return st;
    return launch_nc_fused(x, y_out, nullptr, nel, batch, hA, wA, hB, wB, ncn->wfused, ncn->b2, xmax, 1, stream);
}

extern "C" int p2p_delta_unpack(const uint8_t *delta, size_t n, int ksize, int64_t *out, p2p_stream_t stream) {
    P2P_REQUIRE(delta && out && ksize >= 1, P2P_EINVAL, "p2p_delta_unpack: bad argument");
    if (n == 0) return P2P_OK;
    hipLaunchKernelGGL(delta_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, n,
                       ksize, (long long *)out);
    return check_launch("delta_unpack_kernel");
}

extern "C" int p2p_coarse_matches_batch(const float *corr4d, const uint8_t *delta, int batch, int hA, int wA, int hB, int wB,
                                        int ksize, int upsample, int center, int64_t *matches_out, float *scores_out,
                                        p2p_stream_t stream) {
    P2P_REQUIRE(corr4d && matches_out && scores_out, P2P_EINVAL, "p2p_coarse_matches: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535, P2P_EINVAL, "p2p_coarse_matches: batch %d out of range", batch);
    P2P_REQUIRE(hA > 0 && wA > 0 && hB > 0 && wB > 0 && ksize >= 1, P2P_EINVAL, "p2p_coarse_matches: bad sizes");
    P2P_REQUIRE(ksize == 1 || delta, P2P_EINVAL, "p2p_coarse_matches: delta required when ksize > 1");
    const int nA = hA * wA, nB = hB * wB;
    MatchArgs m{corr4d, delta, hA, wA, hB, wB, ksize, upsample, center, (long long *)matches_out, scores_out,
                (size_t)nA * nB, (size_t)nA + nB};
    hipLaunchKernelGGL(match_cols_kernel, dim3(ceil_div(nB, 8), 1, batch), dim3(256), 0, (hipStream_t)stream, m);
    hipLaunchKernelGGL(match_rows_kernel, dim3(ceil_div(nA, 4), 1, batch), dim3(256), 0, (hipStream_t)stream, m);
    return check_launch("match kernels");
}

extern "C" int p2p_coarse_matches(const float *corr4d, const uint8_t *delta, int hA, int wA, int hB, int wB, int ksize,
                                  int upsample, int center, int64_t *matches_out, float *scores_out, p2p_stream_t stream) {
    return p2p_coarse_matches_batch(corr4d, delta, 1, hA, wA, hB, wB, ksize, upsample, center, matches_out, scores_out, stream);
}
