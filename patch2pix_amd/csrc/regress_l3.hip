// Fine stage, arithmetic P2P_REGRESS_FP16X2W: level 3 of the first convolution of FeatRegressNet as batched GEMMs
// (reference networks/utils.py:4-36: the gather of pyramid level 3; networks/modules.py:76-87: Conv2d(518, 512, 3, stride 2,
// padding 1), of whose 518 input channels 2 x 128 are level 3).
//
// Why.  Level 3 (stride 8) of a 16 x 16 patch is 3 x 3 distinct cells per image, so conv1's level-3 K range is multiplied
// once per CELL, not per pixel (regress_h2.hip header): T3[tap][img][cell][n] = sum_ch cell[ch] W1[n][img, level 3, ch][tap],
// folded into the 8 x 8 output pixels afterwards (acc[pixel][n] += scale[pixel] T3[tap][img][cell(pixel, tap)][n]).  Inside the
// one-proposal-per-work-group kernel those are 16-row MFMA tiles with 9 live rows, and every compute unit streams the 4.6 MB
// of level-3 weights per proposal: 49 % of conv1's weight bytes and 40 % of its matrix-core cycles.  Here the 9 cells of ALL
// proposals are the rows of 18 GEMMs (tap, image) [rows] x [128 channels] x [512 outputs] whose filters are shared by 14
// proposals through LDS (tiles filled 16 / 16); regress_h2_kernel<true> reads its T3 rows from global memory instead
// (331 776 bytes per proposal and level, written once, read once).
//
//   patch_prep_kernel (regress_h2.hip) gathers the 3 x 3 x 128 level-3 values of every (proposal, image) exactly like
//                     select_local_patch_feats does (same clamps) and writes the cells x 2^(138 - biased exponent of the largest
//                     magnitude) (largest value in [2^11, 2^12)) as two fp16 planes into the A blocks of the GEMM
//                     (row = 9 * proposal + cell), in the block layout that IS the LDS image (regress_wino.hip header).
//   l3_gemm_kernel    work-group = 128 rows x one image x three taps x all 512 outputs: the A rows (K = 128: 64 KB) stay in
//                     LDS, the filters stream through a ring of four 16 KB LDS-DMA stages (three in flight, counted vmcnt
//                     waits, one raw barrier per stage, like wino_gemm_kernel); per (tap, column block of 128) 4 stages of
//                     2 x 6 v_mfma_f32_32x32x16_f16 per wave (three products per fp32 product), then the 64 x 32 tile of
//                     every wave goes to T3 as 128-byte row segments.  The kernel is bound by those writes (4.25 GB per
//                     6400 x 2 proposals), not by its 0.8 PFLOP of fp16 products.
#include "regress_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace p2p {

typedef _Float16 le8 __attribute__((ext_vector_type(8)));
#define LMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(le8, (a)), __builtin_bit_cast(le8, (b)), (c), 0, 0, 0)

// --------------------------------------------------------------------------------------------------
constexpr int LNT = 512;                 // 8 waves, two per SIMD
constexpr int LRING = 4;
constexpr int L3_TAPS_PER_WG = 3;        // a work-group walks 3 taps x 4 column blocks = 12 passes of 4 stages
constexpr int L3_PASSES = L3_TAPS_PER_WG * 4;
constexpr int L3_LDS = 4 * WINO_BLK + LRING * WINO_BLK;      // A rows (K = 128) + the filter ring: 128 KB

struct L3GemmArgs {
    const unsigned char *A3;     // [img 2][rowblocks][K chunk 4][WINO_BLK]
    const unsigned char *W;      // [img 2][tap 9][column block 4][K chunk 4][WINO_BLK]
    float *T3;                   // [cn][step 18][wave 8][cell 9][64]
    int rowblocks, cn, c0;
    // compact index -> slot (wino_slot): the item table of the launch
    const int *dev_counts;
    int nitems, n;
    int start[MAXB + 1];
};

// stage S of this work-group's filter stream -> ring slot SLOT: 2 LDS-DMA pieces of 1 KiB per wave
#define LISSUE(S, SLOT)                                                                                                \
    {                                                                                                                  \
        const int s_ = (S) < 4 * L3_PASSES ? (S) : 4 * L3_PASSES - 1;      /* past the end: reload the last stage */     \
        const unsigned char *gb_ = wbase + (size_t)s_ * WINO_BLK + woff;                                               \
        unsigned char *lb_ = smb + 4 * WINO_BLK + (SLOT) * WINO_BLK + wave * 2048;                                      \
        P2P_GLOBAL_LOAD_LDS16(gb_ + lane16, lb_, 0); P2P_GLOBAL_LOAD_LDS16(gb_ + lane16, lb_, 1024);                   \
    }
// fragments of slab S (0, 1) of K chunk KC (A) / ring slot SLOT (B): A[m-tile][plane], B[plane]
#define LREAD(FA, FB, KC, SLOT, S)                                                                                     \
    {                                                                                                                  \
        const unsigned char *pa_ = smb + (KC) * WINO_BLK + ((S) ? aoff1 : aoff0);                                      \
        const unsigned char *pb_ = smb + 4 * WINO_BLK + (SLOT) * WINO_BLK + ((S) ? boff1 : boff0);                      \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                             \
            FA[0][q_] = *(const f32x4 *)(pa_ + q_ * 8192);                                                             \
            FA[1][q_] = *(const f32x4 *)(pa_ + 2048 + q_ * 8192);                                                      \
            FB[q_] = *(const f32x4 *)(pb_ + q_ * 8192);                                                                \
        }                                                                                                              \
    }
#define LSLAB_(FA, FB, C0, C1)                                                                                         \
    M0 = LMFMA(FA[0][1], FB[0], C0); M1 = LMFMA(FA[1][1], FB[0], C1);                                                  \
    M0 = LMFMA(FA[0][0], FB[1], M0); M1 = LMFMA(FA[1][0], FB[1], M1);                                                  \
    M0 = LMFMA(FA[0][0], FB[0], M0); M1 = LMFMA(FA[1][0], FB[0], M1);
#define LPIPE_A()                                                                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }        \
    __builtin_amdgcn_sched_barrier(0);
#define LPIPE_B()                                                                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);          \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }                                                           \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }        \
    __builtin_amdgcn_sched_barrier(0);

__global__ __launch_bounds__(LNT, 1) void l3_gemm_kernel(L3GemmArgs a) {
    P2P_DYN_SHARED(unsigned char, smb);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    // block order: all row blocks of one (image, tap group) are neighbours in time, so its 768 KB of filters stay in the L2s
    const int combo = blockIdx.x / a.rowblocks, rb = blockIdx.x - combo * a.rowblocks;
    const int img = combo / (9 / L3_TAPS_PER_WG), tg = combo - img * (9 / L3_TAPS_PER_WG);
    const int cfirst = (rb * 128) / 9;               // the proposal of the block's first row; none after a missing one exists
    if (cfirst >= a.cn || wino_slot(a, a.c0 + cfirst) < 0) return;
    const int wm = wave >> 2, wn = wave & 3;          // this wave's tile: rows [64 wm, +64), columns [32 wn, +32)
    const unsigned lane16 = lane * 16, woff = wave * 2048;
    const int sw = (l31 >> 2) & 3;
    const unsigned aoff0 = ((wm * 64 + l31) * 4 + ((0 + half) ^ sw)) * 16, aoff1 = ((wm * 64 + l31) * 4 + ((2 + half) ^ sw)) * 16;
    const unsigned boff0 = ((wn * 32 + l31) * 4 + ((0 + half) ^ sw)) * 16, boff1 = ((wn * 32 + l31) * 4 + ((2 + half) ^ sw)) * 16;
    const unsigned char *wbase = a.W + (size_t)((img * 9 + tg * L3_TAPS_PER_WG) * 16) * WINO_BLK;

    // T3 offsets (in floats) of this lane's 32 accumulator rows: register r of m-tile t is row 64 wm + 32 t + (r & 3) +
    // 8 (r >> 2) + 4 half of the block = GEMM row R = 9 * proposal + cell; 0xffffffff beyond the round's proposals
    unsigned rowoff[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned R = (unsigned)rb * 128u + (unsigned)(wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half);
            const unsigned c = R / 9u, cell = R - 9u * c;
            rowoff[t][r] = (c < (unsigned)a.cn) ? c * (unsigned)L3_T3_FLOATS + cell * 64u : 0xffffffffu;
        }

    {   // the block's rows, K = 128: four 16 KB blocks, 8 pieces per wave
        const unsigned char *ga = a.A3 + (size_t)(img * a.rowblocks + rb) * 4 * WINO_BLK + wave * 8192 + lane16;
        unsigned char *la = smb + wave * 8192;
        P2P_GLOBAL_LOAD_LDS16(ga, la, 0); P2P_GLOBAL_LOAD_LDS16(ga, la, 1024);
        P2P_GLOBAL_LOAD_LDS16(ga, la, 2048); P2P_GLOBAL_LOAD_LDS16(ga, la, 3072);
        P2P_GLOBAL_LOAD_LDS16(ga + 4096, la + 4096, 0); P2P_GLOBAL_LOAD_LDS16(ga + 4096, la + 4096, 1024);
        P2P_GLOBAL_LOAD_LDS16(ga + 4096, la + 4096, 2048); P2P_GLOBAL_LOAD_LDS16(ga + 4096, la + 4096, 3072);
    }
    LISSUE(0, 0) LISSUE(1, 1) LISSUE(2, 2)
    P2P_WAIT_VMCNT(4);
    __builtin_amdgcn_s_barrier();
    f32x4 XA[2][2], XB[2], YA[2][2], YB[2];
    LREAD(XA, XB, 0, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 zero16 = {0};

#pragma unroll 1
    for (int pass = 0; pass < L3_PASSES; ++pass) {
        f32x16 M0, M1;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            // slab 0 (fragments X), behind it the reads of slab 1 (fragments Y)
            LREAD(YA, YB, kc, kc, 1)
            if (kc == 0) { LSLAB_(XA, XB, zero16, zero16) } else { LSLAB_(XA, XB, M0, M1) }
            LPIPE_A()
            // stage s + 1 has landed (this wave's pieces; the barrier extends that to everybody's) and every wave is past stage
            // s - 1, whose slot stage s + 3 overwrites.  (Outstanding T3 stores of the previous pass count too: loads return in
            // order among themselves, so "at most 2 outstanding" still implies every piece before the last two has landed.)
            P2P_WAIT_VMCNT(2);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            LISSUE(pass * 4 + kc + 3, (kc + 3) & 3)
            // slab 1, behind it the reads of the next stage's slab 0
            LREAD(XA, XB, (kc + 1) & 3, (kc + 1) & 3, 0)
            LSLAB_(YA, YB, M0, M1)
            LPIPE_B()
        }
        // the 64 x 32 tile -> T3[proposal][step][n >> 6][cell][n & 63]: a wave instruction stores two rows of 128 bytes
        const int tap = tg * L3_TAPS_PER_WG + (pass >> 2), nb = pass & 3;
        const unsigned n = (unsigned)(nb * 128 + wn * 32 + l31);
        float *out = a.T3 + ((unsigned)(tap * 2 + img) * 8u + (n >> 6)) * 576u + (n & 63u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (rowoff[0][r] != 0xffffffffu) out[rowoff[0][r]] = M0[r];
            if (rowoff[1][r] != 0xffffffffu) out[rowoff[1][r]] = M1[r];
        }
    }
    P2P_WAIT_VMCNT(0);
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
static uint16_t l_e(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }
static float l_e2f(uint16_t e) { return (float)__builtin_bit_cast(_Float16, e); }

// level-3 weights of conv1 as the B blocks of l3_gemm_kernel, scaled per output channel by 2^t1[n] like the rest of conv1
// (pack_h2_weights): block (img, tap, column block, K chunk) = [plane 2][column 128][4 pieces of 8 channels, XOR-swizzled]
void pack_l3_weights(const float *conv1_w, const int *t1, float *out) {
    uint16_t *d = (uint16_t *)out;
    for (int img = 0; img < 2; ++img)
        for (int tap = 0; tap < 9; ++tap)
            for (int n = 0; n < 512; ++n) {
                const int nb = n >> 7, col = n & 127;
                for (int k = 0; k < 128; ++k) {
                    const int kc = k >> 5, qq = (k >> 3) & 3, e = k & 7;
                    const float v = std::ldexp(conv1_w[((size_t)n * 518 + img * 259 + 131 + k) * 9 + tap], t1[n]);
                    const uint16_t h0 = l_e(v), h1 = l_e(v - l_e2f(h0));
                    const size_t blk = ((size_t)((img * 9 + tap) * 4 + nb) * 4 + kc) * (WINO_BLK / 2);       // in fp16 elements
                    const size_t in = (size_t)(col * 4 + (qq ^ ((col >> 2) & 3))) * 8 + e;
                    d[blk + in] = h0;
                    d[blk + 128 * 32 + in] = h1;
                }
            }
}

// T3 (and the metadata) of the compact proposals [c0, c1) of level a.lvl0
int launch_regress_l3(const RegressArgs &a, int c0, int c1, const unsigned char *A3, float *T3, hipStream_t stream) {
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {false};          // (idempotent per device; see the threading note of include/p2p_hip.h)
    if (dev >= 64 || !attr_set[dev]) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)l3_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L3_LDS));
        if (dev < 64) attr_set[dev] = true;
    }
    const int cn = c1 - c0;
    if (cn <= 0) return P2P_OK;
    const int rowblocks = (9 * cn + 127) / 128;
    L3GemmArgs g;
    g.A3 = A3; g.W = (const unsigned char *)a.reg[a.lvl0].wl3; g.T3 = T3; g.rowblocks = rowblocks; g.cn = cn; g.c0 = c0;
    g.dev_counts = a.dev_counts; g.nitems = a.nitems; g.n = a.n;
    for (int b = 0; b <= MAXB; ++b) g.start[b] = a.start[b];
    hipLaunchKernelGGL(l3_gemm_kernel, dim3(rowblocks * 2 * (9 / L3_TAPS_PER_WG)), dim3(LNT), L3_LDS, stream, g);
    return check_launch("l3_gemm_kernel");
}

}  // namespace p2p
