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
#include "p2p_common.h"

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
__global__ __launch_bounds__(256) void prep_kernel(const float *__restrict__ F, float *__restrict__ Fn, int C, int h,
                                                   int w, int k) {
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
        for (int c = tid; c < C; c += 256) Fn[(size_t)pp * C + c] = tile[c * (PREP_P + 1) + p] * inv[p];
    }
}

// ------------------------------------------------------------------------------------------------
// 2. correlation GEMM (modules.py:41-53) with the 4-D max-pool (modules.py:11-34) in the epilogue
//    C[pA'][pB'] = sum_c FnA[pA'][c] * FnB[pB'][c];  128x128 tile, 4 waves x (2x2) 32x32x2 MFMA
// ------------------------------------------------------------------------------------------------
constexpr int CT = 128;      // tile edge
constexpr int CBK = 32;      // K per stage
constexpr int CLD = 36;      // LDS row stride (floats): 16-B aligned, conflict-free for ds_read_b128

template <int KS>
__global__ __launch_bounds__(256) void corr_pool_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                        int nA, int nB, int C, float *__restrict__ P,
                                                        uint8_t *__restrict__ delta) {
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

// ------------------------------------------------------------------------------------------------
// 3. row / column maxima of an [nA][nB] matrix (the two torch.max of ncn/model.py:165-166)
// ------------------------------------------------------------------------------------------------
__global__ void fill_keys_kernel(int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = KEY_NEG_INF;
}

// column maxima: a thread owns one column over a 64-row chunk; chunks meet in an (order independent) atomicMax
__global__ __launch_bounds__(256) void colmax_kernel(const float *__restrict__ X, int nA, int nB, int *ckey) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= nB) return;
    const int r0 = blockIdx.y * 64, r1 = min(r0 + 64, nA);
    float cm = -INFINITY;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) cm = fmaxf(cm, X[(size_t)r * nB + col]);
    atomicMax(&ckey[col], f2key(cm));
}

// row maxima: one wave per row
__global__ __launch_bounds__(256) void rowmax_kernel(const float *__restrict__ X, int nA, int nB, int *rkey) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nA) return;
    const float *x = X + (size_t)row * nB;
    float rm = -INFINITY;
    for (int c = lane; c < nB; c += 64) rm = fmaxf(rm, x[c]);
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

__global__ __launch_bounds__(256) void mm_apply_kernel(const float *__restrict__ X, int nA, int nB,
                                                       const int *__restrict__ rkey, const int *__restrict__ ckey,
                                                       float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nA * nB) return;
    const int r = (int)(i / nB), c = (int)(i - (size_t)r * nB);
    out[i] = mm_value(X[i], key2f(rkey[r]), key2f(ckey[c]));
}

// ------------------------------------------------------------------------------------------------
// 4. neighbourhood consensus (ncn/model.py:145-155; conv4d.py:12-74), 1 -> 16 -> 1 channels, both
//    symmetric branches.  Work-group tile: 1 x 4 x 8 x 8 outputs with a one-cell halo in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int TB_ = 4, TC_ = 8, TD_ = 8;
constexpr int HB_ = TB_ + 2, HC_ = TC_ + 2, HD_ = TD_ + 2;
constexpr int HALO = 3 * HB_ * HC_ * HD_;      // 1800 cells

struct Vol { int d0, d1, d2, d3; };

__device__ __forceinline__ void tile_origin(const Vol &v, int &a, int &b0, int &c0, int &d0) {
    const int nd = (v.d3 + TD_ - 1) / TD_, nc = (v.d2 + TC_ - 1) / TC_, nb = (v.d1 + TB_ - 1) / TB_;
    int t = blockIdx.x;
    d0 = (t % nd) * TD_; t /= nd;
    c0 = (t % nc) * TC_; t /= nc;
    b0 = (t % nb) * TB_; t /= nb;
    a = t;
}

// layer 1: X (mutual matching applied on load) -> H1[32][nA][nB], bias + ReLU
__global__ __launch_bounds__(256) void nc_layer1_kernel(const float *__restrict__ X, Vol v,
                                                        const int *__restrict__ rkey, const int *__restrict__ ckey,
                                                        const float *__restrict__ w1cat, const float *__restrict__ b1cat,
                                                        float *__restrict__ H1) {
    __shared__ float tile[HALO];
    int a, b0, c0, d0;
    tile_origin(v, a, b0, c0, d0);
    const int tid = threadIdx.x;
    const int nB = v.d2 * v.d3;
    const size_t nAB = (size_t)v.d0 * v.d1 * nB;
    for (int e = tid; e < HALO; e += 256) {
        int t = e;
        const int dd = t % HD_; t /= HD_;
        const int dc = t % HC_; t /= HC_;
        const int db = t % HB_; t /= HB_;
        const int ia = a + t - 1, ib = b0 + db - 1, ic = c0 + dc - 1, id = d0 + dd - 1;
        float val = 0.f;
        if (ia >= 0 && ia < v.d0 && ib >= 0 && ib < v.d1 && ic >= 0 && ic < v.d2 && id >= 0 && id < v.d3) {
            const int ra = ia * v.d1 + ib, cb = ic * v.d3 + id;
            val = mm_value(X[(size_t)ra * nB + cb], key2f(rkey[ra]), key2f(ckey[cb]));
        }
        tile[e] = val;
    }
    __syncthreads();
    const int tb = tid >> 6, tc = (tid >> 3) & 7, td = tid & 7;
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = 0.f;
    for (int da = 0; da < 3; ++da)
        for (int db = 0; db < 3; ++db) {
            const float *tp = tile + ((da * HB_ + tb + db) * HC_ + tc) * HD_ + td;
            const float *wp = w1cat + (da * 3 + db) * 9 * 32;
#pragma unroll
            for (int dc = 0; dc < 3; ++dc)
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    const float x = tp[dc * HD_ + dd];
#pragma unroll
                    for (int o = 0; o < 32; ++o) acc[o] = fmaf(x, wp[(dc * 3 + dd) * 32 + o], acc[o]);
                }
        }
    const int ib = b0 + tb, ic = c0 + tc, id = d0 + td;
    if (ib < v.d1 && ic < v.d2 && id < v.d3) {
        const size_t pos = (size_t)(a * v.d1 + ib) * nB + ic * v.d3 + id;
#pragma unroll
        for (int o = 0; o < 32; ++o) H1[o * nAB + pos] = fmaxf(acc[o] + b1cat[o], 0.f);
    }
}

// layer 2: Y = relu(b2 + sum_{c<16} W2*H1[c]) + relu(b2 + sum_{c<16} W2^T*H1[16+c])
//
// Work-group tile: 1 x tb x tc x (8*tdr) outputs; a thread owns a run of 8 consecutive outputs along
// the last axis, so every 10-float LDS row read feeds 24 FMAs.  One hidden channel at a time is
// staged (3 x (tb+2) x (tc+2) rows with halo, row stride 8*tdr+4 floats so that ds_read_b128 stays
// 16-B aligned); at ~44 KB per work-group three of them share a CU and overlap each other's loads.
struct NcTile { int tb, tc, tdr, rs; };
constexpr int NC_MAX_ITERS = 12;      // FULLROW staging: wave instructions per channel stage (3*(tb+2)*(tc+2) rows / (4*rpi))

// FULLROW: the tile spans the whole last axis (one d-tile, d3 % 4 == 0): rows are staged with 16-byte
// loads, several rows per wave instruction, all of a stage's loads in flight at once.
template <bool FULLROW>
__global__ __launch_bounds__(256) void nc_layer2_kernel(const float *__restrict__ H1, Vol v, NcTile t,
                                                        const float *__restrict__ w2cat, float b2,
                                                        float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float tile2[];      // [3][tb+2][tc+2][rs]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int td = 8 * t.tdr;
    const int nd = (v.d3 + td - 1) / td, nc = (v.d2 + t.tc - 1) / t.tc, nb = (v.d1 + t.tb - 1) / t.tb;
    int g = blockIdx.x;
    const int d0 = (g % nd) * td; g /= nd;
    const int c0 = (g % nc) * t.tc; g /= nc;
    const int b0 = (g % nb) * t.tb; g /= nb;
    const int a = g;
    const int nB = v.d2 * v.d3;
    const size_t nAB = (size_t)v.d0 * v.d1 * nB;
    const int hb = t.tb + 2, hc = t.tc + 2, ncol = td + 2;
    const int nrows = 3 * hb * hc;
    // this thread's run of 8 outputs
    const int rc = tid % t.tc, rr = (tid / t.tc) % t.tdr, rb = tid / (t.tdr * t.tc);   // rows fastest: fewer LDS bank conflicts
    const bool active = rb < t.tb;
    float out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = 0.f;

    // source offset of every staged row (identical for all 32 channels): computed once
    int *rowoff = (int *)(tile2 + nrows * t.rs);
    for (int r = tid; r < nrows; r += 256) {
        const int dc = r % hc, db = (r / hc) % hb, da = r / (hc * hb);
        const int ia = a + da - 1, ib = b0 + db - 1, ic = c0 + dc - 1;
        const bool rok = ia >= 0 && ia < v.d0 && ib >= 0 && ib < v.d1 && ic >= 0 && ic < v.d2;
        rowoff[r] = rok ? ((ia * v.d1 + ib) * v.d2 + ic) * v.d3 : -1;
    }
    const int colid = d0 + lane - 1;
    const bool colok = lane < ncol && colid >= 0 && colid < v.d3;
    // FULLROW staging geometry: lpr lanes x float4 per row, rpi rows per wave instruction
    const int lpr = v.d3 >> 2, rpi = 64 / max(lpr, 1);
    const int myr = lane / max(lpr, 1), myq = lane - myr * lpr;
    const bool lane_ok = myr < rpi;
    const int niter = (nrows + 4 * rpi - 1) / (4 * rpi);
    if (FULLROW) {      // the two halo columns are outside the volume: zero once, never overwritten
        for (int r = tid; r < nrows; r += 256) { tile2[r * t.rs] = 0.f; tile2[r * t.rs + v.d3 + 1] = 0.f; }
    }
    const int slab_c = t.rs, slab_b = hc * t.rs, slab_a = hb * hc * t.rs;
    const float *mybase = tile2 + (rb * hc + rc) * t.rs + 8 * rr;

    // FULLROW: the rows of hidden channel ch+1 are fetched into registers (16-byte loads, all in flight) while
    // channel ch is convolved out of LDS, and written to LDS after the barrier that ends that compute.
    f32x4 vals[NC_MAX_ITERS];
    auto fetch = [&](int chn) {
        const float *src = H1 + (size_t)chn * nAB;
#pragma unroll
        for (int i = 0; i < NC_MAX_ITERS; ++i) {
            const int r = (i * 4 + wave) * rpi + myr;
            const int off = (i < niter && lane_ok && r < nrows) ? rowoff[r] : -1;
            vals[i] = (off >= 0) ? *(const f32x4 *)(src + off + 4 * myq) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    if (FULLROW) {
        __syncthreads();        // rowoff / halo zeros are ready
        fetch(0);
    }
    for (int branch = 0; branch < 2; ++branch) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int ch = 0; ch < 16; ++ch) {
            const float *src = H1 + (size_t)(branch * 16 + ch) * nAB;
            __syncthreads();
            if (FULLROW) {
#pragma unroll
                for (int i = 0; i < NC_MAX_ITERS; ++i) {
                    const int r = (i * 4 + wave) * rpi + myr;
                    if (i < niter && lane_ok && r < nrows) {
                        float *dst = tile2 + r * t.rs + 1 + 4 * myq;
                        dst[0] = vals[i][0]; dst[1] = vals[i][1]; dst[2] = vals[i][2]; dst[3] = vals[i][3];
                    }
                }
            } else {
            // stage one channel: each wave copies whole rows (coalesced along the last axis), eight
            // rows in flight per wave so that the loads overlap
            for (int r0 = wave; r0 < nrows; r0 += 32) {
                float v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + 4 * u;
                    const int off = (r < nrows) ? rowoff[r] : -1;
                    v8[u] = (off >= 0 && colok) ? src[off + colid] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + 4 * u;
                    if (r < nrows && lane < ncol) tile2[r * t.rs + lane] = v8[u];
                }
            }
            }
            __syncthreads();
            if (FULLROW && branch * 16 + ch + 1 < 32) fetch(branch * 16 + ch + 1);
            if (active) {
                const float *wch = w2cat + (branch * 16 + ch) * 81;
                for (int da = 0; da < 3; ++da)
                    for (int db = 0; db < 3; ++db) {
#pragma unroll
                        for (int dc = 0; dc < 3; ++dc) {
                            const float *p = mybase + da * slab_a + db * slab_b + dc * slab_c;
                            const f32x4 x0 = *(const f32x4 *)p, x1 = *(const f32x4 *)(p + 4);
                            const float x8 = p[8], x9 = p[9];
                            const float x[10] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x8, x9};
                            const float *w = wch + (da * 3 + db) * 9 + dc * 3;
                            const float w0 = w[0], w1 = w[1], w2 = w[2];
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                acc[i] = fmaf(x[i + 2], w2, fmaf(x[i + 1], w1, fmaf(x[i], w0, acc[i])));
                        }
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] += fmaxf(acc[i] + b2, 0.f);
    }
    if (active) {
        const int ib = b0 + rb, ic = c0 + rc;
        if (ib < v.d1 && ic < v.d2) {
            float *dst = Y + ((size_t)(a * v.d1 + ib) * v.d2 + ic) * v.d3;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int id = d0 + 8 * rr + i;
                if (id < v.d3) dst[id] = out[i];
            }
        }
    }
}

// tile shape with the least padding for this volume (<= 256 runs, <= 52 KB of LDS so 3 groups fit a CU)
static NcTile pick_nc_tile(const Vol &v) {
    NcTile best{4, 8, 4, 44};
    double best_eff = -1;
    const int tbs[] = {2, 3, 4, 5, 6, 8}, tcs[] = {4, 5, 6, 8, 10, 12, 15, 16}, tdrs[] = {2, 3, 4, 5, 6, 7};   // 8*tdr+2 columns must fit one wave
    for (int tdr : tdrs)
        for (int tb : tbs)
            for (int tc : tcs) {
                if (tb * tc * tdr > 256) continue;
                const int rs = 8 * tdr + 12;   // 52 words for 40-wide rows: best of the multiples of 4 for ds_read_b128 (bank model)
                const size_t lds = (size_t)3 * (tb + 2) * (tc + 2) * (rs + 1) * 4;
                if (lds > 54000) continue;                 // three groups per CU (160 KiB)
                const double groups = (double)v.d0 * ceil_div(v.d1, tb) * ceil_div(v.d2, tc) * ceil_div(v.d3, 8 * tdr);
                const double useful = (double)v.d0 * v.d1 * v.d2 * v.d3;
                // padding efficiency x halo efficiency (staged cells per useful output)
                const double eff = useful / (groups * 256 * 8) *
                                   ((double)tb * tc * 8 * tdr / ((tb + 2) * (tc + 2) * (8 * tdr + 2)));
                if (eff > best_eff) { best_eff = eff; best = NcTile{tb, tc, tdr, rs}; }
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
};

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
__global__ __launch_bounds__(256) void match_cols_kernel(MatchArgs m) {
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
__global__ __launch_bounds__(256) void match_rows_kernel(MatchArgs m) {
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
    w.fnA = off; off += al(nA * C * 4);
    w.fnB = off; off += al(nB * C * 4);
    w.P = off; off += al(nAc * nBc * 4);
    w.Y = off; off += al(nAc * nBc * 4);
    w.H1 = off; off += al(32 * nAc * nBc * 4);
    w.keys = off; off += al(2 * (nAc + nBc) * 4);
    w.total = off;
    return w;
}

}  // namespace p2p

using namespace p2p;

extern "C" int p2p_ncn_create(const float *w1, const float *b1, const float *w2, const float *b2, p2p_ncn **out) {
    P2P_REQUIRE(w1 && b1 && w2 && b2 && out, P2P_EINVAL, "p2p_ncn_create: null argument");
    // stored layout (conv4d.py:119-120): w1s[da][o][ci=0][db][dc][dd], w2s[da][o=0][ci][db][dc][dd]
    std::vector<float> h(81 * 32 + 32 + 32 * 81, 0.f);
    float *w1cat = h.data(), *b1cat = w1cat + 81 * 32, *w2cat = b1cat + 32;
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
                        w2cat[o * 81 + tap] = W2(o, da, db, dc, dd);
                        w2cat[(16 + o) * 81 + tap] = W2(o, dc, dd, da, db);
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
    p2p_ncn *n = new p2p_ncn();
    n->dev = dev; n->w1cat = dev; n->b1cat = dev + 81 * 32; n->w2cat = dev + 81 * 32 + 32; n->b2 = b2[0];
    *out = n;
    return P2P_OK;
}

extern "C" void p2p_ncn_destroy(p2p_ncn *ncn) {
    if (!ncn) return;
    (void)hipFree(ncn->dev);
    delete ncn;
}

extern "C" size_t p2p_coarse_workspace_bytes(int channels, int hA, int wA, int hB, int wB, int ksize) {
    if (channels <= 0 || hA <= 0 || wA <= 0 || hB <= 0 || wB <= 0 || ksize < 1) return 0;
    return coarse_ws(channels, hA, wA, hB, wB, ksize).total;
}

extern "C" int p2p_coarse_forward(const float *featA, const float *featB, int C, int hA, int wA, int hB, int wB,
                                  int ksize, const p2p_ncn *ncn, float *corr4d_out, uint8_t *delta_out,
                                  void *workspace, size_t workspace_bytes, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(featA && featB && ncn && corr4d_out && workspace, P2P_EINVAL, "p2p_coarse_forward: null argument");
    P2P_REQUIRE(ksize == 1 || ksize == 2, P2P_EUNSUPPORTED, "p2p_coarse_forward: ksize %d not supported (1 or 2)", ksize);
    P2P_REQUIRE(C > 0 && C % 32 == 0 && C <= 256, P2P_EUNSUPPORTED, "p2p_coarse_forward: channels %d (multiple of 32, <= 256)", C);
    P2P_REQUIRE(hA > 0 && wA > 0 && hB > 0 && wB > 0 && hA % ksize == 0 && wA % ksize == 0 && hB % ksize == 0 &&
                    wB % ksize == 0, P2P_EINVAL, "p2p_coarse_forward: feature map sizes must be positive multiples of ksize");
    const CoarseWs ws = coarse_ws(C, hA, wA, hB, wB, ksize);
    P2P_REQUIRE(workspace_bytes >= ws.total, P2P_ENOMEM, "p2p_coarse_forward: workspace %zu < %zu bytes", workspace_bytes, ws.total);
    char *base = (char *)workspace;
    float *fnA = (float *)(base + ws.fnA), *fnB = (float *)(base + ws.fnB);
    float *P = (float *)(base + ws.P), *Y = (float *)(base + ws.Y), *H1 = (float *)(base + ws.H1);
    const int nA = hA * wA, nB = hB * wB, kk = ksize * ksize;
    const int nAc = nA / kk, nBc = nB / kk;
    int *rkey1 = (int *)(base + ws.keys), *ckey1 = rkey1 + nAc, *rkey2 = ckey1 + nBc, *ckey2 = rkey2 + nAc;

    hipLaunchKernelGGL(prep_kernel, dim3(ceil_div(nA, PREP_P)), dim3(256), 0, stream, featA, fnA, C, hA, wA, ksize);
    hipLaunchKernelGGL(prep_kernel, dim3(ceil_div(nB, PREP_P)), dim3(256), 0, stream, featB, fnB, C, hB, wB, ksize);
    const dim3 cgrid(ceil_div(nB, CT), ceil_div(nA, CT));
    if (ksize == 1)
        hipLaunchKernelGGL(corr_pool_kernel<1>, cgrid, dim3(256), 0, stream, fnA, fnB, nA, nB, C, P, (uint8_t *)nullptr);
    else
        hipLaunchKernelGGL(corr_pool_kernel<2>, cgrid, dim3(256), 0, stream, fnA, fnB, nA, nB, C, P, delta_out);

    const int nkeys = 2 * (nAc + nBc);
    hipLaunchKernelGGL(fill_keys_kernel, dim3(ceil_div(nkeys, 256)), dim3(256), 0, stream, rkey1, nkeys);
    const dim3 mgrid(ceil_div(nBc, 256), ceil_div(nAc, 64));
    hipLaunchKernelGGL(colmax_kernel, mgrid, dim3(256), 0, stream, P, nAc, nBc, ckey1);
    hipLaunchKernelGGL(rowmax_kernel, dim3(ceil_div(nAc, 4)), dim3(256), 0, stream, P, nAc, nBc, rkey1);

    Vol v{hA / ksize, wA / ksize, hB / ksize, wB / ksize};
    const int ntiles = v.d0 * ceil_div(v.d1, TB_) * ceil_div(v.d2, TC_) * ceil_div(v.d3, TD_);
    hipLaunchKernelGGL(nc_layer1_kernel, dim3(ntiles), dim3(256), 0, stream, P, v, rkey1, ckey1, ncn->w1cat, ncn->b1cat, H1);
    const NcTile nt = pick_nc_tile(v);
    const int ntiles2 = v.d0 * ceil_div(v.d1, nt.tb) * ceil_div(v.d2, nt.tc) * ceil_div(v.d3, 8 * nt.tdr);
    const size_t lds2 = (size_t)3 * (nt.tb + 2) * (nt.tc + 2) * (nt.rs + 1) * 4;
    const int lpr = v.d3 / 4, rows2 = 3 * (nt.tb + 2) * (nt.tc + 2);
    const bool fullrow = (v.d3 % 4 == 0) && (8 * nt.tdr >= v.d3) && lpr >= 1 && lpr <= 64 &&
                         ceil_div(rows2, 4 * (64 / lpr)) <= NC_MAX_ITERS;
    if (fullrow)
        hipLaunchKernelGGL(nc_layer2_kernel<true>, dim3(ntiles2), dim3(256), lds2, stream, H1, v, nt, ncn->w2cat, ncn->b2, Y);
    else
        hipLaunchKernelGGL(nc_layer2_kernel<false>, dim3(ntiles2), dim3(256), lds2, stream, H1, v, nt, ncn->w2cat, ncn->b2, Y);
    hipLaunchKernelGGL(colmax_kernel, mgrid, dim3(256), 0, stream, Y, nAc, nBc, ckey2);
    hipLaunchKernelGGL(rowmax_kernel, dim3(ceil_div(nAc, 4)), dim3(256), 0, stream, Y, nAc, nBc, rkey2);
    const size_t nel = (size_t)nAc * nBc;
    hipLaunchKernelGGL(mm_apply_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, stream, Y, nAc, nBc, rkey2, ckey2,
                       corr4d_out);
    return check_launch("coarse_forward kernels");
}

extern "C" int p2p_delta_unpack(const uint8_t *delta, size_t n, int ksize, int64_t *out, p2p_stream_t stream) {
    P2P_REQUIRE(delta && out && ksize >= 1, P2P_EINVAL, "p2p_delta_unpack: bad argument");
    if (n == 0) return P2P_OK;
    hipLaunchKernelGGL(delta_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, n,
                       ksize, (long long *)out);
    return check_launch("delta_unpack_kernel");
}

extern "C" int p2p_coarse_matches(const float *corr4d, const uint8_t *delta, int hA, int wA, int hB, int wB, int ksize,
                                  int upsample, int center, int64_t *matches_out, float *scores_out, p2p_stream_t stream) {
    P2P_REQUIRE(corr4d && matches_out && scores_out, P2P_EINVAL, "p2p_coarse_matches: null argument");
    P2P_REQUIRE(hA > 0 && wA > 0 && hB > 0 && wB > 0 && ksize >= 1, P2P_EINVAL, "p2p_coarse_matches: bad sizes");
    P2P_REQUIRE(ksize == 1 || delta, P2P_EINVAL, "p2p_coarse_matches: delta required when ksize > 1");
    MatchArgs m{corr4d, delta, hA, wA, hB, wB, ksize, upsample, center, (long long *)matches_out, scores_out};
    const int nA = hA * wA, nB = hB * wB;
    hipLaunchKernelGGL(match_cols_kernel, dim3(ceil_div(nB, 8)), dim3(256), 0, (hipStream_t)stream, m);
    hipLaunchKernelGGL(match_rows_kernel, dim3(ceil_div(nA, 4)), dim3(256), 0, (hipStream_t)stream, m);
    return check_launch("match kernels");
}
