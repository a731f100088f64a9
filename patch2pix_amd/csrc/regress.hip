// Fine stage of the Patch2Pix matching path on gfx950: one workgroup per proposal runs
//   patch gather (4 pyramid levels, both images) -> per-pixel L2 normalisation ->
//   conv 3x3 s2 (518->512) -> BN -> conv 3x3 s1 (512->512) -> BN -> ReLU -> 8x8 max ->
//   FC 512 -> 512 -> 256 -> 5 -> tanh/sigmoid parse -> clamp
// and, when a second regressor is given, feeds its own result through that one as well.
//
// Reference semantics: networks/utils.py:4-36 (gather), networks/patch2pix.py:157-184 (mini batch),
// networks/modules.py:56-112 (FeatRegressNet), networks/patch2pix.py:138-155 (parse).
//
// Mapping to CDNA4
//   * a proposal is a 64-row (8x8 output pixels) implicit GEMM against 512 output channels; the
//     8 waves of the workgroup each own 64 output channels (2x2 tiles of v_mfma_f32_32x32x2_f32),
//     so the 16x16x259x2 patch is never written to HBM and the 512x8x8 intermediate lives in LDS;
//   * the A operand comes from LDS: the patch is stored *deduplicated* (levels 1-3 are nearest
//     neighbour up-samplings, so a 16x16 window only touches 9x9 / 5x5 / 3x3 distinct cells) and
//     the per-pixel L2 scale is applied when the fragment is read;
//   * the B operand (weights) is streamed straight from L2 into VGPRs in a layout packed at load
//     time in exact consumption order: one global_load_dwordx4 per lane feeds four k-steps;
//   * fp32 MFMA is bit-identical to an fma chain, so results match an fp32 CPU evaluation to
//     round-off of the (fixed, documented) summation order: taps outer, channels inner.
#include "regress_common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace p2p {

constexpr int K1_CHUNKS_PER_TAP = 65;   // 1 (level-0 of both images, 6 ch padded to 8) + 2*(8+8+16)
constexpr int K1_CHUNKS = 9 * K1_CHUNKS_PER_TAP;
constexpr int K2_CHUNKS_PER_TAP = 64;   // 512 channels / 8
constexpr int K2_CHUNKS = 9 * K2_CHUNKS_PER_TAP;
constexpr int PF = 2;                   // weight prefetch distance (chunks); buffers are padded by PF chunks
constexpr int WP1_FLOATS = 8 * (K1_CHUNKS + PF) * 2 * 64 * 4;
constexpr int WP2_FLOATS = 8 * (K2_CHUNKS + PF) * 2 * 64 * 4;


// LDS carve-up (floats)
constexpr int TILE_L0 = 0, TILE_L1 = 768, TILE_L2 = 5952, TILE_L3 = 7552, TILE_IMG = 8704;
constexpr int HSTRIDE = 68;                       // row stride of the conv1 output H[c][64 px]
constexpr int LDS_UNION = 512 * HSTRIDE;          // max(2*TILE_IMG, 512*HSTRIDE)
constexpr int LDS_SCALE = LDS_UNION;              // [2][256] per-pixel 1/||f||
constexpr int LDS_V = LDS_SCALE + 512;            // [512] pooled conv features
constexpr int LDS_F1 = LDS_V + 512;               // [512]
constexpr int LDS_F2 = LDS_F1 + 512;              // [256]
constexpr int LDS_MISC = LDS_F2 + 256;            // [16] raw outputs / current proposal
constexpr int LDS_FLOATS = LDS_MISC + 16;
constexpr size_t LDS_BYTES = size_t(LDS_FLOATS) * 4;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// One chunk = 8 K values = 4 k-steps for both m-tiles and both n-tiles (16 MFMAs).
#define P2P_CHUNK_MFMA(A0, A1)                                   \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {              \
        acc00 = MFMA(A0[q], b0[q], acc00);                       \
        acc01 = MFMA(A0[q], b1[q], acc01);                       \
        acc10 = MFMA(A1[q], b0[q], acc10);                       \
        acc11 = MFMA(A1[q], b1[q], acc11);                       \
    }

__global__ __launch_bounds__(NT, 2) void regress_kernel(RegressArgs args) {
    P2P_DYN_SHARED(float, smem);
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

    float *tiles = smem;
    float *Hbuf = smem;
    float *scale = smem + LDS_SCALE;
    float *V = smem + LDS_V;
    float *F1 = smem + LDS_F1;
    float *F2 = smem + LDS_F2;
    float *misc = smem + LDS_MISC;

    // current proposal (float coordinates; exact for the int64 coarse matches)
    if (tid < 4) {
        float v;
        if (args.is_float) v = ((const float *)args.proposals)[prop * 4 + tid];
        else v = (float)((const long long *)args.proposals)[prop * 4 + tid];
        misc[8 + tid] = v;
    }
    __syncthreads();

    for (int lvl = 0; lvl < args.nlevels; ++lvl) {
        const RegDev &R = args.reg[lvl];
        // ---------------------------------------------------------------- proposal geometry
        // x, y = trunc(match) (networks/utils.py:19); window origin = centre - 8 (:8-15)
        int x0[2], y0[2];
        x0[0] = (int)misc[8 + 0] - 8; y0[0] = (int)misc[8 + 1] - 8;
        x0[1] = (int)misc[8 + 2] - 8; y0[1] = (int)misc[8 + 3] - 8;
        __syncthreads();   // everyone has read misc / finished with the previous level's LDS

        // ---------------------------------------------------------------- gather (dedup tiles)
        for (int img = 0; img < 2; ++img) {
            const int Hh = I.H[img], Ww = I.W[img];
            float *t = tiles + img * TILE_IMG;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int Rr = (j == 0) ? 16 : (j == 1) ? 9 : (j == 2) ? 5 : 3;
                const int Cc = (j == 0) ? 3 : (j == 3) ? 128 : 64;
                const int off = (j == 0) ? TILE_L0 : (j == 1) ? TILE_L1 : (j == 2) ? TILE_L2 : TILE_L3;
                const int Hj = Hh >> j, Wj = Ww >> j;                 // index clamp: dim // ds (networks/utils.py:22-23)
                    const int Ha = level_dim(Hh, j), Wa = level_dim(Ww, j);  // extent of the backbone's map
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
                    t[off + e] = src[((size_t)c * Ha + sy) * Wa + sx];
                }
            }
        }
        __syncthreads();

        // ---------------------------------------------------------------- per-pixel L2 scale
        {
            const int img = tid >> 8, pix = tid & 255, py = pix >> 4, px = pix & 15;
            const float *t = tiles + img * TILE_IMG;
            float ss = 0.f;
            {
                const float *p = t + TILE_L0 + patch_cell(y0[img], py, 0, I.H[img]) * 16 + patch_cell(x0[img], px, 0, I.W[img]);
#pragma unroll
                for (int c = 0; c < 3; ++c) { float v = p[c * 256]; ss = fmaf(v, v, ss); }
            }
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                const int Cc = (j == 3) ? 128 : 64;
                const int off = (j == 1) ? TILE_L1 : (j == 2) ? TILE_L2 : TILE_L3;
                const float *p = t + off + patch_cell(y0[img], py, j, I.H[img]) * Rr + patch_cell(x0[img], px, j, I.W[img]);
                for (int c = 0; c < Cc; ++c) { float v = p[c * Rr * Rr]; ss = fmaf(v, v, ss); }
            }
            scale[tid] = 1.0f / sqrtf(ss + 1e-6f);
        }
        __syncthreads();

        // ---------------------------------------------------------------- conv1: 3x3, stride 2, pad 1
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        {
            // weight stream: b* = current chunk, c* = next chunk, the loads issued below are two ahead
            const f32x4 *bp = (const f32x4 *)R.wp1 + (size_t)wave * (K1_CHUNKS + PF) * 128 + lane;
            f32x4 b0 = bp[0], b1 = bp[64], c0 = bp[128], c1 = bp[192];
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                // per-lane source pixel of both m-tiles for this tap
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
                float a0[4], a1[4];
                // --- chunk 0: level 0 of both images; k = 2q+half -> (img0 c0,c1,c2, img1 c0,c1,c2, 0, 0)
                {
                    f32x4 n0 = bp[256], n1 = bp[320];
                    bp += 128;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int pofs = pyc[t] * 16 + pxc[t];
                        const float s0 = ok[t] ? scale[pofs] : 0.f;
                        const float s1 = ok[t] ? scale[256 + pofs] : 0.f;
                        const float *t0 = tiles + TILE_L0 + patch_cell(y0[0], pyc[t], 0, I.H[0]) * 16 +
                                          patch_cell(x0[0], pxc[t], 0, I.W[0]);
                        const float *t1 = tiles + TILE_IMG + TILE_L0 + patch_cell(y0[1], pyc[t], 0, I.H[1]) * 16 +
                                          patch_cell(x0[1], pxc[t], 0, I.W[1]);
                        float v0 = t0[half * 256] * s0;                                   // img0 c0 | c1
                        float v1 = half ? t1[0] * s1 : t0[512] * s0;                      // img0 c2 | img1 c0
                        float v2 = t1[(1 + half) * 256] * s1;                             // img1 c1 | c2
                        if (t == 0) { a0[0] = v0; a0[1] = v1; a0[2] = v2; a0[3] = 0.f; }
                        else        { a1[0] = v0; a1[1] = v1; a1[2] = v2; a1[3] = 0.f; }
                    }
                    P2P_CHUNK_MFMA(a0, a1)
                    b0 = c0; b1 = c1; c0 = n0; c1 = n1;
                }
                // --- chunks 1..64: (img, level 1..3) segments, 8 channels per chunk
                for (int img = 0; img < 2; ++img) {
                    const float s0 = ok[0] ? scale[img * 256 + pyc[0] * 16 + pxc[0]] : 0.f;
                    const float s1 = ok[1] ? scale[img * 256 + pyc[1] * 16 + pxc[1]] : 0.f;
#pragma unroll
                    for (int j = 1; j < 4; ++j) {
                        const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                        const int CS = Rr * Rr;
                        const int nchunk = (j == 3) ? 16 : 8;
                        const int off = (j == 1) ? TILE_L1 : (j == 2) ? TILE_L2 : TILE_L3;
                        const float *base = tiles + img * TILE_IMG + off + half * CS;
                        const float *p0 = base + patch_cell(y0[img], pyc[0], j, I.H[img]) * Rr +
                                          patch_cell(x0[img], pxc[0], j, I.W[img]);
                        const float *p1 = base + patch_cell(y0[img], pyc[1], j, I.H[img]) * Rr +
                                          patch_cell(x0[img], pxc[1], j, I.W[img]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { a0[q] = p0[2 * q * CS] * s0; a1[q] = p1[2 * q * CS] * s1; }
                        for (int ch = 0; ch < nchunk; ++ch) {
                            f32x4 n0 = bp[256], n1 = bp[320];
                            bp += 128;
                            // A fragments of the next chunk (re-reads the last one at the segment end)
                            const int chn = min(ch + 1, nchunk - 1);
                            float an0[4], an1[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                an0[q] = p0[(chn * 8 + 2 * q) * CS];
                                an1[q] = p1[(chn * 8 + 2 * q) * CS];
                            }
                            P2P_CHUNK_MFMA(a0, a1)
#pragma unroll
                            for (int q = 0; q < 4; ++q) { a0[q] = an0[q] * s0; a1[q] = an1[q] * s1; }
                            b0 = c0; b1 = c1; c0 = n0; c1 = n1;
                        }
                    }
                }
            }
        }
        __syncthreads();   // all waves are done reading the patch tiles

        // BN1 (folded scale/shift) and spill H[c][px] to LDS for conv2's A operand
        {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R.bn1s[n], b = R.bn1b[n];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 &a = (t == 0) ? (u == 0 ? acc00 : acc01) : (u == 0 ? acc10 : acc11);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
                        v[0] = fmaf(a[4 * g + 0], s, b);
                        v[1] = fmaf(a[4 * g + 1], s, b);
                        v[2] = fmaf(a[4 * g + 2], s, b);
                        v[3] = fmaf(a[4 * g + 3], s, b);
                        *(f32x4 *)(Hbuf + n * HSTRIDE + 32 * t + 8 * g + 4 * half) = v;
                    }
                }
            }
        }
        __syncthreads();

        // ---------------------------------------------------------------- conv2: 3x3, stride 1, pad 1
        acc00 = (f32x16){0}; acc01 = (f32x16){0}; acc10 = (f32x16){0}; acc11 = (f32x16){0};
        {
            const f32x4 *bp = (const f32x4 *)R.wp2 + (size_t)wave * (K2_CHUNKS + PF) * 128 + lane;
            f32x4 b0 = bp[0], b1 = bp[64], c0 = bp[128], c1 = bp[192];
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const float *p0, *p1;
                float m0, m1;
                {
                    const int oy = (l31 >> 3) + ky - 1, ox = (l31 & 7) + kx - 1;
                    const bool okx = (ox >= 0) && (ox < 8);
                    const bool ok0 = okx && (oy >= 0);                 // m-tile 0: rows 0..3 (+ky-1 <= 4)
                    const bool ok1 = okx && (oy + 4 < 8);              // m-tile 1: rows 4..7 (+ky-1 >= 3)
                    m0 = ok0 ? 1.f : 0.f;
                    m1 = ok1 ? 1.f : 0.f;
                    p0 = Hbuf + half * HSTRIDE + (ok0 ? oy * 8 + ox : 0);
                    p1 = Hbuf + half * HSTRIDE + (ok1 ? (oy + 4) * 8 + ox : 0);
                }
                float a0[4], a1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { a0[q] = p0[2 * q * HSTRIDE] * m0; a1[q] = p1[2 * q * HSTRIDE] * m1; }
                for (int ch = 0; ch < K2_CHUNKS_PER_TAP; ++ch) {
                    f32x4 n0 = bp[256], n1 = bp[320];
                    bp += 128;
                    const int chn = min(ch + 1, K2_CHUNKS_PER_TAP - 1);
                    float an0[4], an1[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        an0[q] = p0[(chn * 8 + 2 * q) * HSTRIDE];
                        an1[q] = p1[(chn * 8 + 2 * q) * HSTRIDE];
                    }
                    P2P_CHUNK_MFMA(a0, a1)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a0[q] = an0[q] * m0; a1[q] = an1[q] * m1; }
                    b0 = c0; b1 = c1; c0 = n0; c1 = n1;
                }
            }
        }

        // BN2 -> ReLU -> max over the 8x8 outputs (BN before max: its scale may be negative)
        {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R.bn2s[n], b = R.bn2b[n];
                const f32x16 &aa = (u == 0) ? acc00 : acc01;
                const f32x16 &ab = (u == 0) ? acc10 : acc11;
                float m = 0.f;                                  // ReLU folded into the max
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    m = fmaxf(m, fmaf(aa[r], s, b));
                    m = fmaxf(m, fmaf(ab[r], s, b));
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                if (half == 0) V[n] = m;
            }
        }
        __syncthreads();

        // ---------------------------------------------------------------- FC tail + parse
        fc_tail_parse(R, I, args, lvl, prop, tid, V, F1, F2, misc);
    }
}

// --------------------------------------------------------------------------------------------------
// host side: weight packing
// --------------------------------------------------------------------------------------------------

// channel (0..517) of the concatenated [im1 259 | im2 259] regressor input for conv1 K index r (0..519)
// inside one tap; -1 for the two padding slots.
static int conv1_channel_of(int r) {
    if (r < 8) {
        if (r >= 6) return -1;
        return (r / 3) * 259 + (r % 3);
    }
    const int img = (r - 8) / 256, cc = (r - 8) % 256;
    return img * 259 + 3 + cc;
}

static void fold_bn(const p2p_bn_params &bn, int n, float *scale, float *shift) {
    for (int i = 0; i < n; ++i) {
        const float inv = 1.0f / std::sqrt(bn.running_var[i] + 1e-5f);
        const float a = bn.weight[i] * inv;
        scale[i] = a;
        shift[i] = bn.bias[i] - bn.running_mean[i] * a;
    }
}

}  // namespace p2p

using namespace p2p;

// Arithmetic of the two convolutions: new regressors start in P2P_REGRESS_DEFAULT, p2p_regressor_set_mode selects another
// mode per handle (the library reads no environment variables).  Only the weight stream of the mode in use is packed and
// uploaded; another mode's is built on its first selection.
static int upload(const std::vector<float> &h, float **dev, const char *what) {
    *dev = nullptr;
    P2P_HIP_CHECK(hipMalloc(dev, h.size() * sizeof(float)));
    hipError_t e = hipMemcpy(*dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(*dev);
        *dev = nullptr;
        set_error("hipMemcpy of %s failed: %s", what, hipGetErrorString(e));
        return P2P_EHIP;
    }
    return P2P_OK;
}

// pack + upload the convolution weights in the stream order of `mode`'s kernel (once per handle and mode)
static int ensure_mode(p2p_regressor *r, int mode) {
    auto al = [](size_t n) { return (n + 63) & ~size_t(63); };
    const float *c1 = r->conv1_w.data(), *c2 = r->conv2_w.data();
    if (mode == P2P_REGRESS_FP16X2W && !r->dev_w) {
        // conv2 as Winograd filter blocks + conv1's fp16x2 stream (the same stream the direct mode runs, packed here on its own:
        // the direct mode's conv2 stream -- 9.6 MB per regressor -- is neither packed nor uploaded for this mode)
        const size_t ob2 = al(WW2_FLOATS), o1 = ob2 + 512, ob1 = o1 + al(WH1_FLOATS);
        std::vector<float> h(ob1 + 512, 0.f);
        std::vector<int> t1(512), t2(512);
        pack_wino_weights(c2, &h[0], t2.data());
        pack_h2_weights(c1, nullptr, &h[o1], nullptr, t1.data(), nullptr);
        for (int n = 0; n < 512; ++n) {
            h[ob2 + n] = std::ldexp(r->bn2s_host[n], -t2[n]);
            h[ob1 + n] = std::ldexp(r->bn1s_host[n], -12 - t1[n]);      // conv1 accumulates 2^12 x 2^t1[n] x the true sum
        }
        const int st = upload(h, &r->dev_w, "the Winograd filter blocks and conv1's stream");
        if (st != P2P_OK) return st;
        r->ww2 = r->dev_w; r->bn2s_w = r->dev_w + ob2; r->wh1_w = r->dev_w + o1; r->bn1s_w = r->dev_w + ob1;
    } else if (mode == P2P_REGRESS_FP16X2 && !r->dev_h) {
        const size_t o1 = 0, o2 = al(WH1_FLOATS), ob1 = o2 + al(WH2_FLOATS), ob2 = ob1 + 512;
        std::vector<float> h(ob2 + 512, 0.f);
        std::vector<int> t1(512), t2(512);
        pack_h2_weights(c1, c2, &h[o1], &h[o2], t1.data(), t2.data());
        // conv1 accumulates 2^12 (activations) x 2^t1[n] (weights) x the true sum, conv2 2^t2[n] x (the per-proposal scale of H,
        // undone in the kernel) x the true sum: exact powers of two folded into the BatchNorm scales
        for (int n = 0; n < 512; ++n) {
            h[ob1 + n] = std::ldexp(r->bn1s_host[n], -12 - t1[n]);
            h[ob2 + n] = std::ldexp(r->bn2s_host[n], -t2[n]);
        }
        const int st = upload(h, &r->dev_h, "the fp16x2 weight streams");
        if (st != P2P_OK) return st;
        r->wh1 = r->dev_h + o1; r->wh2 = r->dev_h + o2; r->bn1s_h = r->dev_h + ob1; r->bn2s_h = r->dev_h + ob2;
    } else if (mode == P2P_REGRESS_F32 && !r->dev_p) {
        const size_t o_wp1 = 0, o_wp2 = al(WP1_FLOATS);
        std::vector<float> h(o_wp2 + al(WP2_FLOATS), 0.f);
        // conv1: Wp1[w][kc][u][lane][q] = W1[n = 64w+32u+(lane&31)][channel(kidx)][tap],
        //        kidx = 8*(kc % 65) + 2q + (lane>>5), tap = kc / 65
        for (int w = 0; w < 8; ++w)
            for (int kc = 0; kc < K1_CHUNKS; ++kc) {
                const int tap = kc / K1_CHUNKS_PER_TAP, kin = kc % K1_CHUNKS_PER_TAP;
                for (int u = 0; u < 2; ++u)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int q = 0; q < 4; ++q) {
                            const int n = 64 * w + 32 * u + (lane & 31);
                            const int ch = conv1_channel_of(8 * kin + 2 * q + (lane >> 5));
                            const size_t dst = o_wp1 + ((((size_t)w * (K1_CHUNKS + PF) + kc) * 2 + u) * 64 + lane) * 4 + q;
                            h[dst] = (ch < 0) ? 0.f : c1[((size_t)n * 518 + ch) * 9 + tap];
                        }
            }
        // conv2: kidx = 8*(kc % 64) + 2q + (lane>>5) is the input channel, tap = kc / 64
        for (int w = 0; w < 8; ++w)
            for (int kc = 0; kc < K2_CHUNKS; ++kc) {
                const int tap = kc / K2_CHUNKS_PER_TAP, kin = kc % K2_CHUNKS_PER_TAP;
                for (int u = 0; u < 2; ++u)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int q = 0; q < 4; ++q) {
                            const int n = 64 * w + 32 * u + (lane & 31);
                            const int ch = 8 * kin + 2 * q + (lane >> 5);
                            const size_t dst = o_wp2 + ((((size_t)w * (K2_CHUNKS + PF) + kc) * 2 + u) * 64 + lane) * 4 + q;
                            h[dst] = c2[((size_t)n * 512 + ch) * 9 + tap];
                        }
            }
        const int st = upload(h, &r->dev_p, "the f32 weight streams");
        if (st != P2P_OK) return st;
        r->wp1 = r->dev_p + o_wp1; r->wp2 = r->dev_p + o_wp2;
    }
    return P2P_OK;
}

extern "C" int p2p_regressor_set_mode(p2p_regressor *reg, int mode) {
    P2P_REQUIRE(reg, P2P_EINVAL, "p2p_regressor_set_mode: null handle");
    P2P_REQUIRE(mode == P2P_REGRESS_F32 || mode == P2P_REGRESS_FP16X2 || mode == P2P_REGRESS_FP16X2W, P2P_EINVAL,
                "p2p_regressor_set_mode: unknown mode %d", mode);
    // the weight stream of a mode is allocated on the HANDLE's device, whatever the caller's current device is
    int cur = 0;
    P2P_HIP_CHECK(hipGetDevice(&cur));
    if (cur != reg->device) P2P_HIP_CHECK(hipSetDevice(reg->device));
    const int st = ensure_mode(reg, mode);
    if (cur != reg->device) P2P_HIP_CHECK(hipSetDevice(cur));
    if (st != P2P_OK) return st;
    reg->mode = mode;
    return P2P_OK;
}

extern "C" int p2p_regressor_get_mode(const p2p_regressor *reg) { return reg ? reg->mode : P2P_EINVAL; }

extern "C" int p2p_regressor_create(const p2p_regressor_params *p, p2p_regressor **out) {
    P2P_REQUIRE(p && out, P2P_EINVAL, "p2p_regressor_create: null argument");
    const float *const need[] = {p->conv1_w, p->conv2_w, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, p->fc3_w, p->fc3_b,
                                 p->bn1.weight, p->bn1.bias, p->bn1.running_mean, p->bn1.running_var,
                                 p->bn2.weight, p->bn2.bias, p->bn2.running_mean, p->bn2.running_var,
                                 p->bnf1.weight, p->bnf1.bias, p->bnf1.running_mean, p->bnf1.running_var,
                                 p->bnf2.weight, p->bnf2.bias, p->bnf2.running_mean, p->bnf2.running_var};
    for (const float *q : need) P2P_REQUIRE(q, P2P_EINVAL, "p2p_regressor_create: null weight pointer");

    // everything but the convolution weights (which are packed per arithmetic mode, ensure_mode): one device allocation
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
    const size_t o_bn1s = take(512), o_bn1b = take(512), o_bn2s = take(512), o_bn2b = take(512);
    const size_t o_fc1t = take(512 * 512), o_fc1b = take(512), o_bnf1s = take(512), o_bnf1b = take(512);
    const size_t o_fc2t = take(256 * 512), o_fc2b = take(256), o_bnf2s = take(256), o_bnf2b = take(256);
    const size_t o_fc3 = take(5 * 256), o_fc3b = take(8);
    const size_t o_fc1p = take(512 * 512), o_fc2p = take(256 * 512);
    std::vector<float> h(off, 0.f);
    fold_bn(p->bn1, 512, &h[o_bn1s], &h[o_bn1b]);
    fold_bn(p->bn2, 512, &h[o_bn2s], &h[o_bn2b]);
    fold_bn(p->bnf1, 512, &h[o_bnf1s], &h[o_bnf1b]);
    fold_bn(p->bnf2, 256, &h[o_bnf2s], &h[o_bnf2b]);
    // fc weights as [k/4][out][4] so that a wave reads 1 KiB contiguous per step (per-proposal tail of the f32 kernel)
    for (int o = 0; o < 512; ++o)
        for (int k = 0; k < 512; ++k) h[o_fc1t + ((size_t)(k / 4) * 512 + o) * 4 + (k & 3)] = p->fc1_w[(size_t)o * 512 + k];
    for (int o = 0; o < 256; ++o)
        for (int k = 0; k < 512; ++k) h[o_fc2t + ((size_t)(k / 4) * 256 + o) * 4 + (k & 3)] = p->fc2_w[(size_t)o * 512 + k];
    pack_fc_mfma(p->fc1_w, 512, &h[o_fc1p]);      // the same two layers as MFMA fragments (batched tail of the fp16x2 kernel)
    pack_fc_mfma(p->fc2_w, 256, &h[o_fc2p]);
    for (int i = 0; i < 512; ++i) h[o_fc1b + i] = p->fc1_b[i];
    for (int i = 0; i < 256; ++i) h[o_fc2b + i] = p->fc2_b[i];
    for (int i = 0; i < 5 * 256; ++i) h[o_fc3 + i] = p->fc3_w[i];
    for (int i = 0; i < 5; ++i) h[o_fc3b + i] = p->fc3_b[i];

    float *dev = nullptr;
    int st = upload(h, &dev, "the regressor's BatchNorm / FC parameters");
    if (st != P2P_OK) return st;
    p2p_regressor *r = new p2p_regressor();
    r->device = 0;
    (void)hipGetDevice(&r->device);
    r->dev = dev;
    r->dev_p = r->dev_h = r->dev_w = nullptr;
    r->ww2 = r->bn2s_w = r->wp1 = r->wp2 = r->wh1 = r->wh2 = r->bn1s_h = r->bn2s_h = r->wh1_w = r->bn1s_w = nullptr;
    r->conv1_w.assign(p->conv1_w, p->conv1_w + (size_t)512 * 518 * 9);      // host copies: another mode's stream is packed on demand
    r->conv2_w.assign(p->conv2_w, p->conv2_w + (size_t)512 * 512 * 9);
    r->bn1s_host.assign(&h[o_bn1s], &h[o_bn1s] + 512);
    r->bn2s_host.assign(&h[o_bn2s], &h[o_bn2s] + 512);
    r->bn1s = dev + o_bn1s; r->bn1b = dev + o_bn1b; r->bn2s = dev + o_bn2s; r->bn2b = dev + o_bn2b;
    r->fc1t = dev + o_fc1t; r->fc1b = dev + o_fc1b; r->bnf1s = dev + o_bnf1s; r->bnf1b = dev + o_bnf1b;
    r->fc2t = dev + o_fc2t; r->fc2b = dev + o_fc2b; r->bnf2s = dev + o_bnf2s; r->bnf2b = dev + o_bnf2b;
    r->fc3 = dev + o_fc3; r->fc3b = dev + o_fc3b;
    r->fc1p = dev + o_fc1p; r->fc2p = dev + o_fc2p;
    r->mode = P2P_REGRESS_DEFAULT;
    st = ensure_mode(r, r->mode);
    if (st != P2P_OK) {
        p2p_regressor_destroy(r);
        return st;
    }
    *out = r;
    return P2P_OK;
}

extern "C" void p2p_regressor_destroy(p2p_regressor *reg) {
    if (!reg) return;
    (void)hipFree(reg->dev);
    if (reg->dev_p) (void)hipFree(reg->dev_p);
    if (reg->dev_h) (void)hipFree(reg->dev_h);
    if (reg->dev_w) (void)hipFree(reg->dev_w);
    delete reg;
}

static RegDev to_dev(const p2p_regressor *r) {
    RegDev d;
    d.ww2 = r->ww2; d.bn2s_w = r->bn2s_w; d.wp1 = r->wp1; d.wp2 = r->wp2; d.wh1 = r->wh1; d.wh2 = r->wh2;
    d.bn1s_h = r->bn1s_h; d.bn2s_h = r->bn2s_h; d.bn1s = r->bn1s; d.bn1b = r->bn1b; d.bn2s = r->bn2s; d.bn2b = r->bn2b;
    d.fc1t = r->fc1t; d.fc1b = r->fc1b; d.bnf1s = r->bnf1s; d.bnf1b = r->bnf1b;
    d.fc2t = r->fc2t; d.fc2b = r->fc2b; d.bnf2s = r->bnf2s; d.bnf2b = r->bnf2b; d.fc3 = r->fc3; d.fc3b = r->fc3b;
    d.fc1p = r->fc1p; d.fc2p = r->fc2p;
    if (r->mode == P2P_REGRESS_FP16X2W) { d.wh1 = r->wh1_w; d.bn1s_h = r->bn1s_w; }      // conv1's stream of this mode's allocation
    return d;
}

// counts: per item, the number of slots in the concatenated arrays (host memory); dev_counts (optional, device
// memory, indexed like counts): how many of those slots hold a proposal -- the remaining work-groups exit at once.
static int regress_batch_impl(const p2p_regressor *reg1, const p2p_regressor *reg2, int nitems,
                              const p2p_pyramid *im1, const p2p_pyramid *im2, const int *counts, const int *dev_counts,
                              const void *proposals, int is_float,
                              float *matches1, float *probs1, float *raw1,
                              float *matches2, float *probs2, float *raw2, void *workspace, size_t workspace_bytes,
                              p2p_stream_t stream) {
    P2P_REQUIRE(reg1 && im1 && im2 && counts, P2P_EINVAL, "p2p_regress: null argument");
    P2P_REQUIRE(nitems >= 0, P2P_EINVAL, "p2p_regress: negative item count");
    long long total = 0;
    for (int i = 0; i < nitems; ++i) {
        P2P_REQUIRE(counts[i] >= 0, P2P_EINVAL, "p2p_regress: negative proposal count");
        total += counts[i];
    }
    if (total == 0) return P2P_OK;
    P2P_REQUIRE(total < (1ll << 31), P2P_EINVAL, "p2p_regress: too many proposals");
    P2P_REQUIRE(proposals, P2P_EINVAL, "p2p_regress: null proposals");
    P2P_REQUIRE(reg2 ? (matches2 && probs2) : (matches1 && probs1), P2P_EINVAL, "p2p_regress: missing output buffers");
    P2P_REQUIRE(!reg2 || reg2->mode == reg1->mode, P2P_EINVAL, "p2p_regress: the two regressors use different arithmetic modes");
    for (int i = 0; i < nitems; ++i) {
        const p2p_pyramid *im[2] = {im1 + i, im2 + i};
        for (int s = 0; s < 2; ++s) {
            P2P_REQUIRE(im[s]->height >= 8 && im[s]->width >= 8 && im[s]->height < 32768 && im[s]->width < 32768,
                        P2P_EINVAL, "p2p_regress: item %d image %d size %dx%d must be within [8, 32767]", i, s + 1,
                        im[s]->height, im[s]->width);
            for (int j = 0; j < 4; ++j) P2P_REQUIRE(im[s]->level[j], P2P_EINVAL, "p2p_regress: null pyramid level");
        }
    }
    const bool batched_fc = reg1->mode == P2P_REGRESS_FP16X2 || reg1->mode == P2P_REGRESS_FP16X2W;
    if (batched_fc) {
        int most = 0;
        for (int i0 = 0; i0 < nitems; i0 += MAXB) {
            int n = 0;
            for (int b = i0; b < nitems && b < i0 + MAXB; ++b) n += counts[b];
            most = std::max(most, n);
        }
        // the direct mode only parks the pooled features and the next level's proposals; the Winograd mode also the
        // transformed conv2 input of a chunk (p2p_regress_workspace_bytes_mode)
        const size_t need = (reg1->mode == P2P_REGRESS_FP16X2W ? regress_ws_floats((size_t)most) : regress_ws_base_floats((size_t)most)) * sizeof(float);
        P2P_REQUIRE(workspace && ((uintptr_t)workspace & 127) == 0, P2P_EINVAL,
                    "p2p_regress: a 128-byte aligned workspace of %zu bytes (p2p_regress_workspace_bytes_mode) is needed", need);
        // too small: P2P_ENOMEM like every other workspace check of the library (a caller may grow the buffer and retry)
        P2P_REQUIRE(workspace_bytes >= need, P2P_ENOMEM,
                    "p2p_regress: workspace of %zu bytes (p2p_regress_workspace_bytes_mode) needed, got %zu", need, workspace_bytes);
    }
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static DeviceOnce attr_set;
    if (!attr_set.done(dev)) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)regress_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)LDS_BYTES));
        attr_set.set(dev);
    }
    // launches of at most MAXB items; outputs/proposals are indexed by the global proposal number
    int first_prop = 0;
    for (int i0 = 0; i0 < nitems; i0 += MAXB) {
        const int nb = (nitems - i0 < MAXB) ? nitems - i0 : MAXB;
        RegressArgs a;
        int n = 0;
        for (int b = 0; b < nb; ++b) {
            const p2p_pyramid *im[2] = {im1 + i0 + b, im2 + i0 + b};
            for (int s = 0; s < 2; ++s) {
                for (int j = 0; j < 4; ++j) a.item[b].pyr[s][j] = im[s]->level[j];
                a.item[b].H[s] = im[s]->height;
                a.item[b].W[s] = im[s]->width;
            }
            a.start[b] = n;
            n += counts[i0 + b];
        }
        for (int b = nb; b <= MAXB; ++b) a.start[b] = n;
        for (int b = nb; b < MAXB; ++b) a.item[b] = a.item[0];
        a.nitems = nb;
        a.dev_counts = dev_counts ? dev_counts + i0 : nullptr;
        a.is_float = is_float; a.n = n; a.nlevels = reg2 ? 2 : 1;
        a.proposals = is_float ? (const void *)((const float *)proposals + (size_t)first_prop * 4)
                               : (const void *)((const long long *)proposals + (size_t)first_prop * 4);
        a.reg[0] = to_dev(reg1);
        a.reg[1] = reg2 ? to_dev(reg2) : a.reg[0];
        auto adv = [&](float *p, int cols) { return p ? p + (size_t)first_prop * cols : nullptr; };
        a.matches[0] = adv(matches1, 4); a.probs[0] = adv(probs1, 1); a.raw[0] = adv(raw1, 5);
        a.matches[1] = adv(matches2, 4); a.probs[1] = adv(probs2, 1); a.raw[1] = adv(raw2, 5);
        a.ws = (float *)workspace;      // launches of one call are ordered on the stream: they may share the scratch
        if (n > 0) {
            int st;
            a.wU = nullptr; a.hinv = nullptr; a.lvl0 = 0; a.p0 = 0; a.p1 = n; a.mblocks = 0;
            if (reg1->mode == P2P_REGRESS_FP16X2W) {
                st = launch_regress_wino(a, n, (hipStream_t)stream);
            } else if (reg1->mode == P2P_REGRESS_FP16X2) {
                st = launch_regress_h2(a, n, (hipStream_t)stream);
            } else {
                hipLaunchKernelGGL(regress_kernel, dim3(n), dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
                st = check_launch("regress_kernel");
            }
            if (st != P2P_OK) return st;
        }
        first_prop += n;
    }
    return P2P_OK;
}

extern "C" int p2p_regress_batch(const p2p_regressor *reg1, const p2p_regressor *reg2, int nitems,
                                 const p2p_pyramid *im1, const p2p_pyramid *im2, const int *counts,
                                 const void *proposals, int is_float,
                                 float *matches1, float *probs1, float *raw1,
                                 float *matches2, float *probs2, float *raw2, void *workspace, size_t workspace_bytes,
                                 p2p_stream_t stream) {
    return regress_batch_impl(reg1, reg2, nitems, im1, im2, counts, nullptr, proposals, is_float, matches1, probs1, raw1,
                              matches2, probs2, raw2, workspace, workspace_bytes, stream);
}

extern "C" size_t p2p_regress_workspace_bytes(int n) {
    return n > 0 ? regress_ws_floats((size_t)n) * sizeof(float) : 0;
}

extern "C" size_t p2p_regress_workspace_bytes_mode(int n, int mode) {
    if (n <= 0 || mode == P2P_REGRESS_F32) return 0;
    return (mode == P2P_REGRESS_FP16X2W ? regress_ws_floats((size_t)n) : regress_ws_base_floats((size_t)n)) * sizeof(float);
}

extern "C" int p2p_regress_batch_dev(const p2p_regressor *reg1, const p2p_regressor *reg2, int nitems,
                                     const p2p_pyramid *im1, const p2p_pyramid *im2, const int *dev_counts, int stride,
                                     const void *proposals, int is_float,
                                     float *matches1, float *probs1, float *raw1,
                                     float *matches2, float *probs2, float *raw2, void *workspace, size_t workspace_bytes,
                                     p2p_stream_t stream) {
    P2P_REQUIRE(dev_counts && stride >= 1 && nitems >= 0 && nitems <= 4096, P2P_EINVAL, "p2p_regress_batch_dev: bad argument");
    std::vector<int> cap(nitems, stride);
    return regress_batch_impl(reg1, reg2, nitems, im1, im2, cap.data(), dev_counts, proposals, is_float, matches1, probs1,
                              raw1, matches2, probs2, raw2, workspace, workspace_bytes, stream);
}

extern "C" int p2p_regress(const p2p_regressor *reg1, const p2p_regressor *reg2,
                           const p2p_pyramid *im1, const p2p_pyramid *im2,
                           const void *proposals, int is_float, int n,
                           float *matches1, float *probs1, float *raw1,
                           float *matches2, float *probs2, float *raw2, void *workspace, size_t workspace_bytes,
                           p2p_stream_t stream) {
    P2P_REQUIRE(n >= 0, P2P_EINVAL, "p2p_regress: negative proposal count");
    return p2p_regress_batch(reg1, reg2, 1, im1, im2, &n, proposals, is_float, matches1, probs1, raw1, matches2, probs2,
                             raw2, workspace, workspace_bytes, stream);
}
