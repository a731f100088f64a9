// Fine stage, arithmetic P2P_REGRESS_FP16X2W: the second convolution of FeatRegressNet (reference networks/modules.py:80-84:
// Conv2d(512, 512, 3, stride 1, padding 1) on the 8 x 8 map, then BatchNorm2d + ReLU + MaxPool2d(8)) as Winograd F(2x2, 3x3).
//
// Why.  The one-launch kernel (regress_h2.hip) keeps one proposal per work-group, so every compute unit streams the whole
// 9.4 MB of conv2 weights per proposal through its 64 B/clk vector-memory path and issues the convolution's full 27.6 k
// matrix-core passes -- at the package power limit (profiles/r04_power_probe.txt) only fewer passes or fewer bytes help.
// Here conv2 is 16 GEMMs (one per position (i, j) of the 4 x 4 transformed tile) whose M axis runs over the 2 x 2 output
// tiles of ALL proposals: 2.25x fewer passes, and the transformed filters are shared by 8 proposals through LDS.
//
//   U_p[row][k] = (B^T d B)[i][j]   d = the 4 x 4 window (stride 2, zero ring) of H = BN1(conv1) around tile (ty, tx), channel k;
//                                   row = 16 * proposal + 4 * ty + tx; written by regress_h2_kernel<true> (regress_h2.hip),
//                                   scaled per proposal so that |U| < 2^13, as two fp16 planes
//   W_p[k][n]   = (G g G^T)[i][j]   g = the 3 x 3 filter of (output n, input k); computed in fp64 at pack time, scaled per
//                                   output channel to [2^11, 2^12), two fp16 planes
//   M_p = U_p W_p                   three v_mfma_f32_32x32x16_f16 per product (a1 b0 + a0 b1 + a0 b0), fp32 accumulation
//   Y[a][b] += A^T[a][i] A^T[b][j] M_p   (coefficients 0, +-1: exact)      the 2 x 2 outputs of the tile
//   V[proposal][n] = max(0, max over tiles and (a, b) of bn2(Y))            (BN before the max: its scale may be negative)
//
// wino_gemm_kernel.  Work-group = 128 rows (8 proposals x 16 tiles) x 128 output channels, 8 waves (two per SIMD: issuing an
// LDS-DMA piece stalls a wave for ~60-100 cycles, longer than an MFMA lasts, so a single wave per SIMD left the matrix pipe
// 45 % busy; its sibling now issues meanwhile), each a 64 x 32 tile = 2 x 1 MFMA tiles: M (32) + Y (128) accumulators.
// K = 16 positions x 512 channels is walked in 256 stages of 32 channels; a stage's operands are two 16 KB blocks
// (A: U rows, B: filters) whose GLOBAL layout is the LDS image -- [plane 2][row 128][4 pieces of 16 B], piece q of a row
// stored at slot q ^ ((row >> 2) & 3), which puts the 16 lanes of every ds_read_b128 service group on 16 different bank
// slots -- so they are copied by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into a ring of four
// stages, three in flight, with counted s_waitcnt vmcnt and ONE raw s_barrier per stage (in the middle of the stage, so
// that the fragments of the next stage's first slab are prefetched behind the second slab's MFMAs).
// The four work-groups that share a row block (output-channel blocks 0-3) get consecutive slots on the same XCD, so U is
// fetched from HBM once and re-read from that XCD's L2.
#include "regress_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace p2p {

typedef _Float16 we8 __attribute__((ext_vector_type(8)));
#define WMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(we8, (a)), __builtin_bit_cast(we8, (b)), (c), 0, 0, 0)

constexpr int WNT = 512;                 // threads per work-group: 8 waves, two per SIMD
constexpr int WSTAGE = 2 * WINO_BLK;     // A block + B block
constexpr int WRING = 4;
constexpr int WLDS = WRING * WSTAGE;     // 128 KB
static_assert(WINO_BLK == 2 * 128 * 64, "block = [plane 2][row 128][32 K of fp16]");

struct WinoArgs {
    const unsigned char *U;      // [position 16][row block][K chunk 16][WINO_BLK]
    const unsigned char *Wt;     // [position 16][column block 4][K chunk 16][WINO_BLK]
    const float *bn2s, *bn2b;    // folded BatchNorm of conv2 (scale with the filter exponents folded in), [512]
    const float *hinv;           // per proposal of the chunk: inverse of the power of two its H was scaled by
    float *V;                    // pooled features [n][512] of the level
    int mblocks;                 // row blocks of the chunk
    int p0, n;                   // first (compact) proposal index of the chunk; slots of the launch
    // compact index -> slot (wino_slot): the item table of the launch
    const int *dev_counts;
    int nitems;
    int start[MAXB + 1];
};

// one stage = 4 LDS-DMA pieces of 1 KiB per wave: wave w copies bytes [2048 w, 2048 w + 2048) of the A and of the B block
#define WISSUE(ST, SLOT)                                                                                               \
    {                                                                                                                  \
        const int st_ = (ST) < 256 ? (ST) : 255;          /* past the end: reload the last stage into a free slot */  \
        const int p_ = st_ >> 4, kc_ = st_ & 15;                                                                       \
        const unsigned char *ga_ = a.U + ((size_t)(p_ * a.mblocks + mb) * 16 + kc_) * WINO_BLK + woff;                  \
        const unsigned char *gb_ = a.Wt + ((size_t)((p_ * 4 + nb) * 16 + kc_)) * WINO_BLK + woff;                       \
        unsigned char *la_ = smb + (SLOT) * WSTAGE + wave * 2048;                                                      \
        P2P_GLOBAL_LOAD_LDS16(ga_ + lane16, la_, 0); P2P_GLOBAL_LOAD_LDS16(ga_ + lane16, la_, 1024);                   \
        P2P_GLOBAL_LOAD_LDS16(gb_ + lane16, la_ + WINO_BLK, 0); P2P_GLOBAL_LOAD_LDS16(gb_ + lane16, la_ + WINO_BLK, 1024); \
    }
// fragments of slab S (0, 1) of ring slot SLOT: A[m-tile][plane], B[plane]
#define WREAD(FA, FB, SLOT, S)                                                                                         \
    {                                                                                                                  \
        const unsigned char *pa_ = smb + (SLOT) * WSTAGE + ((S) ? aoff1 : aoff0);                                      \
        const unsigned char *pb_ = smb + (SLOT) * WSTAGE + WINO_BLK + ((S) ? boff1 : boff0);                            \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                             \
            FA[0][q_] = *(const f32x4 *)(pa_ + q_ * 8192);                                                             \
            FA[1][q_] = *(const f32x4 *)(pa_ + 2048 + q_ * 8192);                                                      \
            FB[q_] = *(const f32x4 *)(pb_ + q_ * 8192);                                                                \
        }                                                                                                              \
    }
// 6 MFMAs of a slab; smallest terms first; the two accumulators alternate
#define WSLAB_(FA, FB, C0, C1)                                                                                         \
    M0 = WMFMA(FA[0][1], FB[0], C0); M1 = WMFMA(FA[1][1], FB[0], C1);                                                  \
    M0 = WMFMA(FA[0][0], FB[1], M0); M1 = WMFMA(FA[1][0], FB[1], M1);                                                  \
    M0 = WMFMA(FA[0][0], FB[0], M0); M1 = WMFMA(FA[1][0], FB[0], M1);
#define WSLAB(FA, FB) WSLAB_(FA, FB, M0, M1)
#define WSLABZ(FA, FB) WSLAB_(FA, FB, zero16, zero16)

// The compiler folds a software prefetch back into read -> wait -> MFMA chains unless the interleave is pinned: one LDS read
// (and, in the half of a stage that issues the next DMA, LDS-DMA pieces) behind each MFMA of a slab.
#define WPIPE_A()                                                                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }        \
    __builtin_amdgcn_sched_barrier(0);
#define WPIPE_B()                                                                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);          \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }                                                           \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }        \
    __builtin_amdgcn_sched_barrier(0);

__global__ __launch_bounds__(WNT, 1) void wino_gemm_kernel(WinoArgs a) {
    P2P_DYN_SHARED(unsigned char, smb);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    // work-group g runs on XCD g mod 8: the four column blocks of a row block are neighbours there
    const int g = blockIdx.x, xcd = g & 7, q = g >> 3;
    const int nb = q & 3, mb = (q >> 2) * 8 + xcd;
    if (mb >= a.mblocks || wino_slot(a, a.p0 + mb * 8) < 0) return;      // no row block, or none of its proposals exists
    const int wm = wave >> 2, wn = wave & 3;          // this wave's tile: rows [64 wm, +64), columns [32 wn, +32)
    const unsigned lane16 = lane * 16, woff = wave * 2048;
    // fragment addresses inside a block: row r, piece (2 slab + half) ^ ((r >> 2) & 3)
    const int sw = (l31 >> 2) & 3;
    const unsigned aoff0 = ((wm * 64 + l31) * 4 + ((0 + half) ^ sw)) * 16, aoff1 = ((wm * 64 + l31) * 4 + ((2 + half) ^ sw)) * 16;
    const unsigned boff0 = ((wn * 32 + l31) * 4 + ((0 + half) ^ sw)) * 16, boff1 = ((wn * 32 + l31) * 4 + ((2 + half) ^ sw)) * 16;

    const f32x16 zero16 = {0};
    f32x16 Y[4][2];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab)
#pragma unroll
        for (int t = 0; t < 2; ++t) Y[ab][t] = zero16;
    f32x4 XA[2][2], XB[2], YA[2][2], YB[2];

    WISSUE(0, 0) WISSUE(1, 1) WISSUE(2, 2)
    P2P_WAIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    WREAD(XA, XB, 0, 0)
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int p = 0; p < 16; ++p) {
        f32x16 M0, M1;
#pragma unroll
        for (int kc = 0; kc < 16; ++kc) {
            const int slot = kc & 3;
            // slab 0 (fragments X), behind it the reads of slab 1 (fragments Y)
            WREAD(YA, YB, slot, 1)
            if (kc == 0) { WSLABZ(XA, XB) } else { WSLAB(XA, XB) }
            WPIPE_A()
            // stage st + 1 has landed (this wave's pieces; the barrier extends that to everybody's) and every wave is
            // past stage st - 1, whose slot stage st + 3 overwrites
            P2P_WAIT_VMCNT(4);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            WISSUE(p * 16 + kc + 3, (kc + 3) & 3)
            // slab 1, behind it the reads of the next stage's slab 0
            WREAD(XA, XB, (kc + 1) & 3, 0)
            WSLAB(YA, YB)
            WPIPE_B()
        }
        // Y[a][b] += A^T[a][i] A^T[b][j] M,   A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
        {
            const int i = p >> 2, j = p & 3;
            const float ca[2] = {(i < 3) ? 1.f : 0.f, (i == 0) ? 0.f : (i == 1) ? 1.f : -1.f};
            const float cb[2] = {(j < 3) ? 1.f : 0.f, (j == 0) ? 0.f : (j == 1) ? 1.f : -1.f};
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const float c = ca[ab >> 1] * cb[ab & 1];
                Y[ab][0] += c * M0; Y[ab][1] += c * M1;
            }
        }
    }
    P2P_WAIT_VMCNT(0);

    // BN2 -> ReLU -> max over the 16 tiles x 4 outputs of each proposal.  Register r of an accumulator is row
    // (r & 3) + 8 (r >> 2) + 4 half of its m-tile: r < 8 belongs to the m-tile's first proposal, r >= 8 to its second.
    const int n = nb * 128 + wn * 32 + l31;
    const float s = a.bn2s[n], b = a.bn2b[n];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            const int pl = mb * 8 + wm * 4 + 2 * t + gq;          // proposal inside the chunk
            const int slot = wino_slot(a, a.p0 + pl);
            const float hs = s * a.hinv[pl];
            float m = 0.f;
#pragma unroll
            for (int ab = 0; ab < 4; ++ab)
#pragma unroll
                for (int r = 0; r < 8; ++r) m = fmaxf(m, fmaf(Y[ab][t][8 * gq + r], hs, b));
            m = fmaxf(m, __shfl_xor(m, 32));
            if (half == 0 && slot >= 0) a.V[(size_t)slot * 512 + n] = m;
        }
}

// The GEMM copies whole 128-row blocks of the transformed input.  Rows of a block that no proposal owns (the tail of a chunk whose
// size is no multiple of 8; with device-side counts, everything behind the last proposal of a partly filled block) are zeros,
// never uninitialised memory: the GEMM's stores are masked, so their products were never used, but NaN bit patterns went through
// the matrix cores.  One work-group per row block, before the conv1 launch of the chunk; blocks without such rows exit at once.
__global__ __launch_bounds__(WNT) void wino_zero_tail_kernel(WinoArgs a, unsigned char *U, float *hinv, int p1) {
    const int mb = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wino_slot(a, a.p0 + mb * 8) < 0) return;          // an empty block: the GEMM skips it
    for (int j = 1; j < 8; ++j) {
        const int c = a.p0 + mb * 8 + j;
        if (c < p1 && wino_slot(a, c) >= 0) continue;
        // the 16 rows of proposal j of the block: lane = (tile, piece of 8 channels), wave w the K chunks 2 w and 2 w + 1
        const unsigned rr = (unsigned)j * 16u + ((unsigned)lane >> 2);
        const unsigned inblk = (rr * 4u + (((unsigned)lane & 3u) ^ ((rr >> 2) & 3u))) * 16u;
        const size_t pstride = (size_t)a.mblocks * (16u * WINO_BLK);
        for (int i2 = 0; i2 < 2; ++i2) {
            unsigned char *ub = U + (size_t)(((unsigned)mb * 16u + (unsigned)(wave * 2 + i2)) * (unsigned)WINO_BLK + inblk);
            for (int pos = 0; pos < 16; ++pos) {
                *(uint4 *)(ub + pos * pstride) = make_uint4(0u, 0u, 0u, 0u);
                *(uint4 *)(ub + pos * pstride + WINO_BLK / 2) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (tid == 0) hinv[mb * 8 + j] = 0.f;
    }
}

// FC tail of a level as its own launch (the one-launch kernel runs it at the end of a work-group's share): same work
// distribution, same code (fc_batch_parse, regress_common.h).
__global__ __launch_bounds__(NT, 2) void regress_fc_kernel(RegressArgs args, int lvl) {
    P2P_DYN_SHARED(unsigned char, smb);
    fc_batch_parse(args.reg[lvl], args, lvl, args.ws + (size_t)lvl * args.n * 512,
                   args.ws + ((2 * (size_t)args.n * 512 + 31) & ~(size_t)31), smb, threadIdx.x);
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
static uint16_t w_e(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }
static float w_e2f(uint16_t e) { return (float)__builtin_bit_cast(_Float16, e); }

// transformed filters of conv2 as the B blocks of wino_gemm_kernel; t2[n] = the exponent output channel n was scaled by
void pack_wino_weights(const float *conv2_w, float *out, int *t2) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<double> wt((size_t)16 * 512 * 512);            // [p][n][k]
    std::vector<double> mx(512, 0.0);
    for (int n = 0; n < 512; ++n)
        for (int k = 0; k < 512; ++k) {
            const float *gk = conv2_w + ((size_t)n * 512 + k) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * gk[0 * 3 + b] + G[i][1] * gk[1 * 3 + b] + G[i][2] * gk[2 * 3 + b];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double v = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    wt[((size_t)(i * 4 + j) * 512 + n) * 512 + k] = v;
                    mx[n] = std::max(mx[n], std::fabs(v));
                }
        }
    for (int n = 0; n < 512; ++n) {
        t2[n] = 0;
        if (mx[n] > 0.0 && std::isfinite(mx[n])) {
            int e;
            std::frexp(mx[n], &e);
            t2[n] = 12 - e;
        }
    }
    uint16_t *d = (uint16_t *)out;
    for (int p = 0; p < 16; ++p)
        for (int n = 0; n < 512; ++n) {
            const int nb = n >> 7, col = n & 127;
            for (int k = 0; k < 512; ++k) {
                const int kc = k >> 5, qq = (k >> 3) & 3, e = k & 7;
                const float v = (float)std::ldexp(wt[((size_t)p * 512 + n) * 512 + k], t2[n]);
                const uint16_t h0 = w_e(v), h1 = w_e(v - w_e2f(h0));
                const size_t blk = ((size_t)(p * 4 + nb) * 16 + kc) * (WINO_BLK / 2);       // in fp16 elements
                const size_t in = (size_t)(col * 4 + (qq ^ ((col >> 2) & 3))) * 8 + e;
                d[blk + in] = h0;
                d[blk + 128 * 32 + in] = h1;
            }
        }
}

// Both levels of a launch: per level, chunks of whole units of 256 proposals (wino_nchunks / wino_chunk_range, at most WINO_CHUNK
// each) run conv1 -> U (regress_h2_kernel<true>) and the
// GEMMs (U -> V); then the level's FC tail, whose matches are the next level's proposals.
int launch_regress_wino(RegressArgs a, int n, hipStream_t stream) {
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static DeviceOnce attr_set;
    if (!attr_set.done(dev)) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)wino_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WLDS));
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)regress_fc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FC_LDS_BYTES));
        attr_set.set(dev);
    }
    int ncu = 0;
    P2P_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    ncu = std::max(ncu, 1);
    P2P_REQUIRE(a.ws, P2P_EINVAL, "%s: the scratch buffer is missing", "launch_regress_wino");
    unsigned char *wsU = (unsigned char *)(a.ws + wino_u_offset_floats((size_t)n));
    float *hinv = a.ws + wino_hinv_offset_floats((size_t)n);
    for (int lvl = 0; lvl < a.nlevels; ++lvl) {
        for (int ch = 0, nch = wino_nchunks((size_t)n); ch < nch; ++ch) {
            int p0, p1;
            wino_chunk_range((size_t)n, ch, &p0, &p1);
            const int cn = p1 - p0, mblocks = (cn + 7) / 8;
            if (cn <= 0) continue;
            a.lvl0 = lvl; a.p0 = p0; a.p1 = p0 + cn; a.wU = wsU; a.hinv = hinv; a.mblocks = mblocks;
            WinoArgs w;
            w.U = wsU; w.Wt = (const unsigned char *)a.reg[lvl].ww2; w.bn2s = a.reg[lvl].bn2s_w; w.bn2b = a.reg[lvl].bn2b;
            w.hinv = hinv; w.V = a.ws + (size_t)lvl * n * 512; w.mblocks = mblocks; w.p0 = p0; w.n = n;
            w.dev_counts = a.dev_counts; w.nitems = a.nitems;
            for (int b = 0; b <= MAXB; ++b) w.start[b] = a.start[b];
            int st = P2P_OK;
            if (a.dev_counts || (cn & 7)) {      // rows of a partly filled row block that no proposal owns: zeros (host counts: the last block only)
                const int mb0 = a.dev_counts ? 0 : mblocks - 1;
                WinoArgs wz = w;
                wz.p0 = p0 + mb0 * 8;
                hipLaunchKernelGGL(wino_zero_tail_kernel, dim3(mblocks - mb0), dim3(WNT), 0, stream, wz,
                                   wsU + (size_t)mb0 * 16 * WINO_BLK, hinv + mb0 * 8, p0 + cn);
                st = check_launch("wino_zero_tail_kernel");
                if (st != P2P_OK) return st;
            }
            st = launch_regress_h2_conv1(a, cn, stream);
            if (st != P2P_OK) return st;
            hipLaunchKernelGGL(wino_gemm_kernel, dim3(((mblocks + 7) / 8) * 32), dim3(WNT), WLDS, stream, w);
            st = check_launch("wino_gemm_kernel");
            if (st != P2P_OK) return st;
        }
        hipLaunchKernelGGL(regress_fc_kernel, dim3(std::min(n, ncu)), dim3(NT), FC_LDS_BYTES, stream, a, lvl);
        const int st = check_launch("regress_fc_kernel");
        if (st != P2P_OK) return st;
    }
    return P2P_OK;
}

}  // namespace p2p
