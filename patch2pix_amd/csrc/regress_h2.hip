// Fine stage (reference networks/patch2pix.py:157-218, networks/modules.py:56-112, networks/utils.py:4-36), the default
// arithmetic P2P_REGRESS_FP16X2: fp32-equivalent on the fp16 matrix cores in THREE products per fp32 product.
//
// Decomposition.  A proposal is an implicit GEMM with M = 64 (8 x 8 output pixels), N = 512 output channels, K = 518 * 9
// (conv1, stride 2) then 512 * 9 (conv2).  One 512-thread work-group holds one proposal at a time (145 KB of LDS), wave w
// owns output channels [64 w, 64 w + 64).  Work-groups are PERSISTENT: the launch has one per compute unit, each walks its
// share of the proposals; per regressor level it runs the two convolutions of each of its proposals (pooled features
// V[512] -> a scratch buffer in global memory), then the FC tail of ALL of them as batches of 16 rows on the exact-fp32
// matrix path (fc_batch_parse, regress_common.h: the 1.5 MB of FC weights are streamed once per 16 proposals), whose
// regressed matches are the next level's proposals (patch2pix.py:259-272).
//
// Arithmetic.  Every fp32 operand x of the two convolutions is represented as the sum of two fp16 numbers
//     x * 2^s = h0 + h1,   h0 = fp16(x * 2^s), h1 = fp16(x * 2^s - h0)          (2 x 11 = 22 bits + the sign of h1)
// whose error is <= 2^-24 |x| (two half-ulp roundings of 11-bit significands) PROVIDED neither plane leaves the normal
// range of fp16 (2^-14 ... 65504).  That is what the power-of-two scales 2^s are for -- they are exact, and every one of
// them is undone exactly by a later multiplication:
//   * activations of conv1: the per-pixel L2-normalised level-0 values (|v| <= 1) times 2^12; the cells of levels 1, 2 and
//     3 times 2^e of their image, e = 12 + floor(log2(smallest per-pixel scale)), so that every component is <= 2^12 --
//     their partial sums are multiplied by scale[pixel] * 2^(12 - e) instead of scale[pixel];
//   * weights: per output channel the power of two that brings the largest weight to [2^11, 2^12) (at pack time; folded
//     into the BatchNorm scale that follows);
//   * H = BN1(conv1), the input of conv2: per proposal the power of two that brings max |H| to [2^12, 2^13) (a
//     work-group reduction on the accumulators); BN2's scale is multiplied by its inverse.
// Elements more than 2^17 below the largest of their tensor lose relative precision (their second plane becomes an fp16
// subnormal, absolute error 2^-25 on the scaled value = 2^-37 of the largest), which is far below the rounding of the
// fp32 accumulation.  A product is
//     a * b ~= a0*b0 + (a0*b1 + a1*b0)        (the dropped a1*b1 is <= 2^-24 |a*b|),   fp32 accumulation
// = 3 v_mfma_f32_32x32x16_f16.  Measured against an fp64 evaluation it is as accurate as the exact-f32 MFMA kernel
// (tools/margin_sweep.py); ceiling 2500 / 3 = 833 TFLOP/s of algorithmic fp32 work.
//
// The kernel runs at the package power limit (effective clock 1.7-2.0 GHz; with all-zero weights the same instruction
// stream runs at 2.4 GHz), so it is organised to issue as little as possible (profiles/r0*_ablation_log.txt):
//   * conv1, levels 2 and 3 of the patch (192 of 259 channels).  They are nearest-neighbour up-samplings (4x4 / 8x8 pixels
//     per cell) and a level-3 cell is determined by the level-2 cell, so the 64 output pixels of a tap touch <= 25 distinct
//     (level-2 cell, level-3 parent) rows.  Their K-range is multiplied ONCE PER CELL ROW from tiles that were split into
//     the two planes once, when they were gathered; the per-pixel L2 scale and the cell -> pixel expansion are applied when
//     the partial sum T[cell row][n] is folded into the accumulators (acc[pixel][n] += scale[pixel] * T[row(pixel)][n],
//     through LDS fold buffers).  Level 3 (3 x 3 cells) runs on 16-row tiles (v_mfma_f32_16x16x32_f16) into its own fold
//     buffer T3, level 2 on a 32-row tile into T2: -24.8 % MFMAs against pixel rows.
//   * conv1, level 1 (no de-duplication possible at stride 2: a tap's 64 pixels touch 64 different cells): since round 6 also
//     from planes split once per proposal (cells x 2^e); the tap's pixel rows are multiplied unscaled into temporaries, one
//     n-tile at a time, and the per-pixel scale is applied when they are added to the accumulators (XQSTEP / XPFOLD below).
//     Level 0 (3 channels, 4 slabs per proposal): a pre-scaled fp32 im2col block, split in registers (40 VALU operations
//     per 8 values), software-pipelined against the MFMAs of the other m-tile.
//   * conv2: its input H = BN1(conv1) is split into the planes ONCE per proposal, by the BN1 pass, straight into the
//     four 128-channel K-chunks conv2 walks (two fp16 planes are as many bytes as fp32): no conversion passes, no
//     work-group barrier inside conv2.  The conv2 loop has no VALU work at all.
//   * weights are pre-split at load time: unit = (slab of 16 K, n-tile) = 2 planes x 1 KiB per wave, stream order =
//     consumption order per wave, addressed as scalar base + 16 * lane (one VGPR for the whole stream), prefetched three
//     slabs ahead through a ring of eight register buffers in BOTH convolutions: conv1 streams 9.3 MB per proposal
//     against 141 k cycles of MFMA issue, i.e. it needs the full 64 B/clk of the compute unit's vector-memory path and
//     ~100 KB in flight (a four-unit ring, one slab ahead, ran its cell range at a third of that).
//   * in conv2 (MFMA-bound) the halves of the work-group (waves 0-3 / 4-7; wave w shares its SIMD with wave w + 4) take
//     turns on the matrix pipe between bare s_barriers; in conv1 (ingest-bound) they run free in a staggered order
//     (waves 0-3: pixel range, cell range, fold; waves 4-7: cell range, fold, pixel range), so that every wave's loads
//     are in flight all the time and a fold sits beside the partner's MFMAs.
//
// Two instantiations.  regress_h2_kernel<false> is the kernel described above (mode P2P_REGRESS_FP16X2: both levels, both
// convolutions and the FC tail in ONE persistent launch).  regress_h2_kernel<true> is the first launch of the default mode
// P2P_REGRESS_FP16X2W (regress_wino.hip): one level, a chunk of the proposals, gather + conv1 exactly as above, and then --
// instead of conv2 -- the Winograd input transform B^T d B of H = BN1(conv1), written to global memory in the block layout
// the GEMM kernel copies into LDS; the gather of the work-group's next proposal is in flight during that write-out.
#define XNPL 2                          // planes per operand; XUB = bytes of a weight unit per wave (XNPL x 1 KiB)
#include "regress_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace p2p {

// XNPL planes per operand, XNPROD MFMA products per fp32 product, XUB bytes of a weight unit (XNPL x 1 KiB per wave)
typedef _Float16 xe8 __attribute__((ext_vector_type(8)));
typedef _Float16 xe2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int XUB = XNPL * 1024;

// ---- LDS layout (bytes) --------------------------------------------------------------------------
// conv1 phase, per image: level 1 as two fp16 planes [plane][9 grid rows (+112 B)][9 cells][64 ch (+16 B pad)]; levels 2 and 3 as
// two fp16 planes [plane][25 cells + 432 B of zeros][64 ch (+16 B)] and [plane][9 cells + 576 B of zeros][128 ch (+32 B)] ("Bank
// slots" below).  Then level 0 raw [img][3][256] and one
// shared region that is, in turn: the fp32 copies of levels 1-3 the scale pass reads (and the planes are converted from); the pre-scaled level-0 im2col
// block A0[64 px][64 K fp32 (+16 B)] (K = img*32 + tap*3 + c, 27 real per image); the fold buffers
// T2[8 waves][25 level-2 rows][64 n] and T3[8 waves][9 level-3 rows][64 n] fp32 (the two n-tiles of a wave interleaved).  Then the fold table [img][17][17] of {scale, T row offset} (row/column 0 =
// the zero padding ring of the convolution) and the per-pixel scale [2][256].
//
// Bank slots.  ds_read_b128 is served in four groups of 16 lanes (lanes {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same
// + 32); a group is conflict-free when its lanes hit 16 different 16-byte slots of the 256-byte bank row.  All A-fragment
// reads have "lane & 31 = tile row, lane >> 5 = K half" (16-row tiles: lane & 15, lane >> 4), and the tile rows of a group
// are 16 different values mod 16, so every layout below makes the slot a bijection of (row mod 16):
//   level 1: the 9 x 9 cell grid with 9-slot cells and a row pitch of 8 (mod 16) slots: the 4 y x 4 x pixels of a service group
//            land on slots {0,9,2,11} + {0,12,4,8} = 16 different ones (a plain [81] array put pairs of lanes on one slot);
//   level 2: cell stride 9 slots (odd); level 3 (16-row tiles, K quarter = lane >> 4): cell stride 2 slots (mod 16), so that
//            rows 0-3/12-15 of K quarter q take the even and rows 4-11 of quarter q + 1 the odd slots;
//   the dead rows of a cell tile (and, in conv2, taps outside the 8 x 8 map) read zeros from the piece of a zero area that
//   keeps them on the slot of their virtual row -- one shared zero cell collided with a live row in half of the groups.
// level 1 (round 6): two fp16 planes like levels 2 / 3 -- [plane][9 grid rows][9 cells][64 ch (+16 B)]; a cell is 9 slots, a grid
// row 88 slots = 8 (mod 16): the 16 lanes of a ds_read_b128 service group (4 y x 4 x of the tap's pixels) land on 16 different slots
// ({0,9,2,11} + {0,8,..} ...: see "Bank slots"), as the fp32 grid of rounds 4-5 did with its 17-slot cells
constexpr int XST1 = 64 * 2 + 16;                                 // bytes per level-1 cell (9 slots)
constexpr int XRP1 = 9 * XST1 + 112;                              // level-1 grid row: 88 slots = 8 (mod 16)
constexpr int XPL1 = 9 * XRP1;                                    // plane stride
constexpr int YST2 = 64 * 2 + 16, YNC2 = 25, YPL2 = (YNC2 + 3) * YST2;  // level 2: cell stride, cells (+ 432 B of zeros), plane stride
constexpr int YST3 = 128 * 2 + 32, YNC3 = 9, YPL3 = (YNC3 + 2) * YST3;  // level 3: 18 slots per cell (+ 576 B of zeros)
constexpr int XOFF1 = 0;
constexpr int YOFF2 = XOFF1 + XNPL * XPL1;
constexpr int YOFF3 = YOFF2 + XNPL * YPL2;
constexpr int XIMG = YOFF3 + XNPL * YPL3;
constexpr int XRAW0 = 2 * XIMG;                                   // float [2][3][256]
constexpr int XSHARED = XRAW0 + 2 * 3 * 256 * 4;
constexpr int XTMP2ST = 64 * 4 + 16, XTMP3ST = 128 * 4 + 16;      // fp32 copy of levels 2/3: [25][272] then [9][528]
constexpr int XTMP3 = YNC2 * XTMP2ST, XTMPIMG = XTMP3 + YNC3 * XTMP3ST;
constexpr int XTMP1 = 2 * XTMPIMG, XTMP1ST = 64 * 4 + 16, XTMP1IMG = 81 * XTMP1ST;   // fp32 copy of level 1: [img][81 cells][64 ch (+16 B:
                                                                  // the gather's lanes walk cells, 17 slots apart)], behind levels 2 / 3
constexpr int XA0ST = 64 * 4 + 16;
// Exclusive turns of the two waves of a SIMD on the matrix pipe (XPP below): conv2 (MFMA-bound, 43 B/clk of weights) gains 3 %
// from them; conv1 -- whose weight stream needs every wave's loads in flight all the time -- lost 5 % of the launch to them
// (profiles/r04_ablation_log.txt), so its halves run free in a staggered order, every wave with a level-2 fold buffer of its own.
constexpr int XTROW = 64 * 4;                                     // one fold-buffer row: 64 output channels of a wave, fp32
constexpr int XTROWS = 25, XT2N = 8;      // one level-2 buffer per wave, only the 25 rows that are read back
constexpr int XTW = XTROWS * XTROW;                               // level-2 fold buffer
constexpr int XSHR = XT2N * XTW + 8 * 9 * XTROW;                  // + T3[8 waves][9 level-3 rows]
constexpr int XTAB = XSHARED + XSHR;
constexpr int XTABIMG = 17 * 17 * 8;
constexpr int XSM_SCALE = XTAB + 2 * XTABIMG + 16;                // float [2][256]
constexpr int XCONV1B = XSM_SCALE + 512 * 4;
// conv2 phase.  H = BN1(conv1), [64 px][512 ch], is conv2's A operand, walked in four K-chunks of 128 input channels; a
// chunk's planes are [plane][66 px][128 ch 16-bit (+16 B)]; rows 64-65 = zeros = the padding ring of the convolution.
// The 272-byte row puts the 16 lanes of every ds_read_b128 service group (rows {0-3,12-15,20-27}, ... : 16 different
// values mod 16) on 16 different 16-byte slots of the 256-byte bank row.  A lane whose tap falls outside the 8x8 map reads
// zeros from the slot its virtual row would have had (piece (row + K half) & 15 of the zero rows): sending all of them
// to ONE zero row put them on the slot of a lane with a real row -- 27 % of conv2's LDS cycles were such 2-way conflicts.
constexpr int HST = 128 * 2 + 16, HPL = 66 * HST;                 // 272, 17952
// two fp16 planes are as many bytes as fp32: ALL four chunks are written as planes by the BN1 pass (no conversion passes,
// no work-group barriers inside conv2)
constexpr int HCHUNK = XNPL * HPL;                                // 35904: planes of chunk c start at c * HCHUNK
constexpr int XCONV2B = 4 * HCHUNK;
// both phases, then the FC batch of the level (fc_batch_parse) over the whole allocation
constexpr int XSM_MISC = (XCONV1B > XCONV2B) ? XCONV1B : XCONV2B;  // [16] floats: 8-11 the proposal, 12-14 the fp16 scale reductions
#ifdef P2P_X3_TIMING
constexpr int XSM_BYTES = XSM_MISC + 16 * 4 + 8 * 16 * 4;
#else
constexpr int XSM_BYTES = XSM_MISC + 16 * 4;
#endif
static_assert(FC_LDS_BYTES <= XSM_MISC, "the FC batch stages its rows over the convolution buffers");
static_assert(XTMP1 + 2 * XTMP1IMG <= XSHR && 64 * XA0ST <= XSHR, "the shared region is sized by the fold buffers");
static_assert(XPL1 % 16 == 0 && XTMP1 % 16 == 0 && XIMG % 16 == 0 && YOFF2 % 16 == 0 && YOFF3 % 16 == 0 && YPL2 % 16 == 0 && YPL3 % 16 == 0 && XSHARED % 16 == 0 &&
              XTAB % 16 == 0 && XSM_SCALE % 16 == 0 && HPL % 16 == 0 && HCHUNK % 16 == 0 && XCONV2B % 16 == 0,
              "16-byte alignment of ds_read_b128");
static_assert(XSM_BYTES <= 160 * 1024, "LDS budget");

// two fp32 -> one dword of two fp16 (round to nearest even): v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pk_e(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, xe2));
}
__device__ __forceinline__ unsigned short f2e(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float e2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ float pk_lo(unsigned h) { return (float)__builtin_bit_cast(xe2, h)[0]; }
__device__ __forceinline__ float pk_hi(unsigned h) { return (float)__builtin_bit_cast(xe2, h)[1]; }
// v = p0 + p1 (+ p2): the planes of one value, stored XPLST bytes apart
__device__ __forceinline__ void store_planes(unsigned char *dst, int plane_stride, float v) {
    const unsigned short p0 = f2e(v);
    const float r1 = v - e2f(p0);
    const unsigned short p1 = f2e(r1);
    *(unsigned short *)dst = p0;
    *(unsigned short *)(dst + plane_stride) = p1;
}

// 8 consecutive K values of one row (two 16-byte LDS reads), scaled by s, as XNPL planes of 8 elements
__device__ __forceinline__ void splitn(const f32x4 &xa, const f32x4 &xb, float s, f32x4 (&p)[XNPL]) {
    const float x[8] = {xa[0] * s, xa[1] * s, xa[2] * s, xa[3] * s, xb[0] * s, xb[1] * s, xb[2] * s, xb[3] * s};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h = pk_e(x[2 * q], x[2 * q + 1]);
        const float r0 = x[2 * q] - pk_lo(h);
        const float r1 = x[2 * q + 1] - pk_hi(h);
        const unsigned m = pk_e(r0, r1);
        p[0][q] = __uint_as_float(h);
        p[1][q] = __uint_as_float(m);
    }
}

#ifdef XF_PIN_W                         // timing experiment (wrong results): the weight stream never advances (always cache hits)
#define XWADV(N)
#else
#define XWADV(N) wb += (N) * XUB;
#endif
#ifdef P2P_X3_TIMING                    // phase lengths in s_memtime ticks -> args.raw[0] (tools/x3_timing.py)
// per-wave counters in LDS (14 more live SGPRs spill): [wave][16] unsigned behind the misc block
#define XTL_() ((unsigned *)(smb + XSM_MISC + 64) + wave * 16)
#define XT_DECL
#define XT_START { const unsigned n_ = (unsigned)__builtin_amdgcn_s_memtime(); if (P2P_LANE_ID() < 16) XTL_()[P2P_LANE_ID()] = (P2P_LANE_ID() == 15) ? n_ : 0u; }
#define XT(i) { const unsigned n_ = (unsigned)__builtin_amdgcn_s_memtime(); if (P2P_LANE_ID() == 0) { unsigned *x_ = XTL_(); x_[i] += n_ - x_[15]; x_[15] = n_; } }
#if P2P_X3_TIMING >= 2                  // also the three ranges inside a conv1 step (stamps inside the hot loop: spills)
#define XTL(i) XT(i)
#else
#define XTL(i)
#endif
#else
#define XT_DECL
#define XT_START
#define XT(i)
#define XTL(i)
#endif
#define XMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(xe8, (a)), __builtin_bit_cast(xe8, (b)), (c), 0, 0, 0)

// raw fp32 A fragment (8 K values of this lane's row) of one m-tile: two 16-byte LDS reads
#define XLOADR(R, P) R[0] = *(const f32x4 *)(P); R[1] = *(const f32x4 *)((P) + 16);
// pre-split A fragment of one m-tile: one 16-byte LDS read per plane, PL = plane stride
#define XLOADP(S, P, PL) { _Pragma("unroll") for (int q_ = 0; q_ < XNPL; ++q_) S[q_] = *(const f32x4 *)((P) + q_ * (PL)); }
// weights of the unit `AHEAD` units after the current stream position: 3 planes
// (wb = wave-uniform stream position, kept in SGPRs; wlane = 16 * lane: one VGPR addresses every weight load)
#define XLOADB(BUF, AHEAD)                                                               \
    { unsigned wo_ = (AHEAD) * XUB; P2P_OPAQUE_S(wo_);        /* opaque: keeps "+ AHEAD units" on the scalar side */  \
      _Pragma("unroll") for (int q_ = 0; q_ < XNPL; ++q_) BUF[q_] = *(const f32x4 *)((wb + wo_) + wlane + q_ * 1024); }
// 12 MFMAs of one m-tile (planes SP) against the weights of both n-tiles: 6 products each, smallest terms first;
// the two accumulators alternate.  XHALFZ starts the two accumulators from zero.
#define XNPROD 3
#define XHALF_(C0IN, C1IN, CU0, CU1, SP, BU0, BU1)                                       \
    CU0 = XMFMA(SP[1], BU0[0], C0IN); CU1 = XMFMA(SP[1], BU1[0], C1IN);                  \
    CU0 = XMFMA(SP[0], BU0[1], CU0); CU1 = XMFMA(SP[0], BU1[1], CU1);                    \
    CU0 = XMFMA(SP[0], BU0[0], CU0); CU1 = XMFMA(SP[0], BU1[0], CU1);
#define XNM (2 * XNPROD)                /* MFMAs of one m-tile against both n-tiles of a slab */
#define XHALF(CU0, CU1, SP, BU0, BU1) XHALF_(CU0, CU1, CU0, CU1, SP, BU0, BU1)
#define XHALFZ(CU0, CU1, SP, BU0, BU1) XHALF_(zero16, zero16, CU0, CU1, SP, BU0, BU1)

// ---- slabs whose A operand is split in registers (conv1, levels 0 and 1) ---------------------------
// Software pipeline inside a wave.  The matrix pipe takes 32 cycles per MFMA and a wave issues in order, so a wave
// that first splits a whole slab and then issues its 24 MFMAs leaves the pipe idle while it -- and the other wave of
// the SIMD, which runs the same code in step -- does VALU work.  Here every group of 12 MFMAs (one m-tile) carries
// the split of the OTHER m-tile's next fragment in its shadow:
//   phase A:  MFMAs of m-tile 0 (planes S0) || LDS read of the next slab's m-tile-0 fragment, split of R1 -> S1,
//             weight loads of the next slab's first unit
//   phase B:  MFMAs of m-tile 1 (planes S1) || LDS read of the next slab's m-tile-1 fragment, split of R0 -> S0,
//             weight loads of the next slab's second unit
// sched_group_barrier pins the interleave (1 MFMA, then up to 4 VALU; the loads at the head of the phase).
// Weights: (BC0, BC1) = this slab's two units, (BN0, BN1) = the next slab's, loaded one slab ahead (>= 768 matrix-pipe
// cycles) into the buffers the previous slab used.
#define XPIPE(NDS)                                                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0); __builtin_amdgcn_sched_group_barrier(0x020, XNPL, 0);        \
    _Pragma("unroll") for (int g_ = 0; g_ < XNM; ++g_) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
#define XPIPE0()                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x020, XNPL, 0);                                                             \
    _Pragma("unroll") for (int g_ = 0; g_ < XNM; ++g_) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
#define XSLAB(N0, N1, SC0, SC1, BC0, BC1, BN0, BN1, AH)                                   \
    { XLOADR(R0, N0) XLOADB(BN0, AH) splitn(R1[0], R1[1], (SC1), S1);                               \
      XHALF(acc00, acc01, S0, BC0, BC1) XPIPE(2) __builtin_amdgcn_sched_barrier(0);                                   \
      XLOADR(R1, N1) XLOADB(BN1, (AH) + 1) splitn(R0[0], R0[1], (SC0), S0);                          \
      XHALF(acc10, acc11, S1, BC0, BC1) XPIPE(2) __builtin_amdgcn_sched_barrier(0); }
// Weights: a ring of EIGHT units = four slabs (B0 ... B7), every slab loads the two units of the slab THREE slabs ahead into
// the buffers the previous slab just freed (AH = 6, 8, 10, 12 for the four slabs of a group, then the stream advances by 8
// units).  conv1 streams 9.3 MB of weights per proposal against 141 k cycles of MFMA issue: it needs the full 64 B/clk of
// the compute unit's vector-memory path, i.e. ~100 KB in flight per compute unit at the ~1500-cycle latency of a loaded L2
// (a ring of four units, one slab ahead, left 4 KB per wave in flight: the cell range ran at a third of that rate).
// Every range of conv1 is a whole number of groups (level 0: one, pixel range: one, cell range: three), so each starts at
// ring position 0 with units 0-5 of its first group already in flight.
#define XGROUP4(SL0, SL1, SL2, SL3) { SL0 SL1 SL2 SL3 XWADV(8) }
// The last slab of such a run: nothing to read or split for a next slab.
#define XSLABEND(SC1, BC0, BC1, BN0, BN1, AH)                                            \
    { XLOADB(BN0, AH) splitn(R1[0], R1[1], (SC1), S1);                                              \
      XHALF(acc00, acc01, S0, BC0, BC1) XPIPE0() __builtin_amdgcn_sched_barrier(0);                                    \
      XLOADB(BN1, (AH) + 1)                                                                                           \
      XHALF(acc10, acc11, S1, BC0, BC1) XPIPE0() __builtin_amdgcn_sched_barrier(0); }
// start of a run of slabs: fragments of its first slab
#define XPRO(P0, P1, SC0) { XLOADR(R0, P0) XLOADR(R1, P1) splitn(R0[0], R0[1], (SC0), S0); }

// ---- conv1, level 1 (round 6): pixel rows of the pre-split cell planes -------------------------------------------
// Until round 5 the level-1 operand (cell x per-pixel scale) was split in registers inside the loop: 80 VALU operations per 12
// MFMAs kept the range at 0.63 of its MFMA issue.  Now the cells are split ONCE per proposal (x 2^e of their image, like levels
// 2 / 3), a tap's 64 pixel rows are multiplied unscaled into two temporaries (m-tile 0 / 1), ONE n-tile at a time, and the
// per-pixel scale is applied when they are added to the accumulators: acc[pixel][n] += table[pixel(tap)] * tp[pixel][n] (the
// fold table's entry, zero for the padding ring).  Step j of a (tap, image) range consumes weight unit j = (n-tile j / 4,
// slab j % 4) and loads unit j + 6 (the ring of eight); the A fragments of a slab are read twice (once per n-tile).
#define XQHALF_(C0IN, C1IN, CU0, CU1, A0, A1, BU)                                        \
    CU0 = XMFMA(A0[1], BU[0], C0IN); CU1 = XMFMA(A1[1], BU[0], C1IN);                    \
    CU0 = XMFMA(A0[0], BU[1], CU0); CU1 = XMFMA(A1[0], BU[1], CU1);                      \
    CU0 = XMFMA(A0[0], BU[0], CU0); CU1 = XMFMA(A1[0], BU[0], CU1);
#define XQHALF(CU0, CU1, A0, A1, BU) XQHALF_(CU0, CU1, CU0, CU1, A0, A1, BU)
#define XQHALFZ(CU0, CU1, A0, A1, BU) XQHALF_(zero16, zero16, CU0, CU1, A0, A1, BU)
#define XQPIPE()                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 2 * XNPL; ++g_) {                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, XNPL, 0);          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#define XQSTEP(HALF, AC0, AC1, AN0, AN1, NP0, NP1, BC, BN, AH)                            \
    { XLOADP(AN0, NP0, XPL1) XLOADP(AN1, NP1, XPL1) XLOADB(BN, AH)                                                    \
      HALF(tp0, tp1, AC0, AC1, BC) XQPIPE() __builtin_amdgcn_sched_barrier(0); }
// acc (m-tile 0 / 1 of one n-tile) += table entry of the register's pixel x tp
#define XPFOLD(ACC0, ACC1)                                                               \
    { _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                             \
          const float e0_ = *(const float *)(tabq + ((2 * (r_ >> 2)) * 17 + 2 * (r_ & 3)) * 8);                       \
          const float e1_ = *(const float *)(tabq + ((8 + 2 * (r_ >> 2)) * 17 + 2 * (r_ & 3)) * 8);                   \
          ACC0[r_] = fmaf(e0_, tp0[r_], ACC0[r_]); ACC1[r_] = fmaf(e1_, tp1[r_], ACC1[r_]); } }

// ---- slabs whose A operand was split beforehand ------------------------------------------------------
#define XLPIPE(NDS, NMFMA)                                                               \
    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0); __builtin_amdgcn_sched_group_barrier(0x020, 2 * XNPL, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
// conv1, levels 2 + 3: one m-tile of cell rows -> T0/T1.  SC = planes of this slab, SN <- planes of the next (NP, NPL).
#define XCSLAB(HALF, SC, SN, NP, NPL, BC0, BC1, BN0, BN1, AH)                             \
    { XLOADP(SN, NP, NPL) XLOADB(BN0, AH) XLOADB(BN1, (AH) + 1)                                                       \
      HALF(t0, t1, SC, BC0, BC1) XLPIPE(XNPL, XNM) __builtin_amdgcn_sched_barrier(0); }
// Level 3 (9 cells per image) on 16-row tiles, v_mfma_f32_16x16x32_f16 (A: lane l = row l & 15, K block
// l >> 4; B: column l & 15; D: rows 4 * (l >> 4) + r, column l & 15).  One "pseudo-slab" = one K step of 32 channels
// against two 16-column n-tiles = 12 MFMAs of 16 cycles and two weight units.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define X16MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xe8, (a)), __builtin_bit_cast(xe8, (b)), (c), 0, 0, 0)
#define X16HALF_(C0IN, C1IN, CU0, CU1, SP, BU0, BU1)                                     \
    CU0 = X16MFMA(SP[1], BU0[0], C0IN); CU1 = X16MFMA(SP[1], BU1[0], C1IN);              \
    CU0 = X16MFMA(SP[0], BU0[1], CU0); CU1 = X16MFMA(SP[0], BU1[1], CU1);                \
    CU0 = X16MFMA(SP[0], BU0[0], CU0); CU1 = X16MFMA(SP[0], BU1[0], CU1);
#define X16HALF(CU0, CU1, SP, BU0, BU1) X16HALF_(CU0, CU1, CU0, CU1, SP, BU0, BU1)
#define X16HALFZ(CU0, CU1, SP, BU0, BU1) X16HALF_(zero4, zero4, CU0, CU1, SP, BU0, BU1)
// LOADNEXT: the LDS reads this pseudo-slab carries for a later one (or nothing)
#define X16SLAB(HALF, UA, UB, SC, LOADNEXT, BC0, BC1, BN0, BN1, AH)                       \
    { LOADNEXT XLOADB(BN0, AH) XLOADB(BN1, (AH) + 1)                                                                  \
      HALF(UA, UB, SC, BC0, BC1) __builtin_amdgcn_sched_barrier(0); }
// conv2: both m-tiles from the planes (AC0, AC1); (AN0, AN1) <- the next slab's (addresses NP0, NP1)
// Two waves of a SIMD that both issue MFMAs back to back get ~57 % of the matrix pipe between them, one wave alone
// 85 % (measured); so the two halves of the work-group take turns, two slabs (48 MFMAs) at a time: XPP() = the two
// barriers that end a wave's turn and its partner's (a bare s_barrier: outstanding loads stay in flight).
#define XPB() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define XPP() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
// (the MFMAs lead: the first ones issue as soon as the turn starts, the loads for later slabs follow in their shadow -- one load
// behind every MFMA measured 2-3 % faster than two bursts at the head; four slabs per turn, global loads first, alternating
// LDS / global loads, one load per two MFMAs: no better; profiles/r03_ablation_log.txt, r04_ablation_log.txt)
#define XHPIPE()                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                                \
    _Pragma("unroll") for (int g_ = 0; g_ < 2 * XNPL; ++g_) {                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }       \
    _Pragma("unroll") for (int g_ = 0; g_ < 2 * XNPL; ++g_) {                                                        \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }       \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * XNPROD - 4 * XNPL - 2, 0);
// MFMA order in which consecutive instructions share one operand and the four accumulators rotate (0.5 % faster than m-tile
// after m-tile)
#define XQUAD(AC0, AC1, BC0, BC1, P, Q)                                                  \
    acc00 = XMFMA(AC0[P], BC0[Q], acc00); acc01 = XMFMA(AC0[P], BC1[Q], acc01);                                       \
    acc11 = XMFMA(AC1[P], BC1[Q], acc11); acc10 = XMFMA(AC1[P], BC0[Q], acc10);
#define XHMFMAS(AC0, AC1, BC0, BC1)                                                      \
    XQUAD(AC0, AC1, BC0, BC1, 1, 0) XQUAD(AC0, AC1, BC0, BC1, 0, 1) XQUAD(AC0, AC1, BC0, BC1, 0, 0)
#define XHSLAB(AC0, AC1, AN0, AN1, NP0, NP1, BC0, BC1, BN0, BN1, AH)                       \
    { XLOADP(AN0, NP0, HPL) XLOADP(AN1, NP1, HPL) XLOADB(BN0, AH) XLOADB(BN1, (AH) + 1)                                \
      XHMFMAS(AC0, AC1, BC0, BC1)                                                                                     \
      XHPIPE() __builtin_amdgcn_sched_barrier(0); }

// Persistent work-groups: the launch has min(n, compute units) of them (one fits a compute unit), work-group g owns the
// proposals g, g + G, g + 2G, ...  Per level it runs the two convolutions of each of its proposals (pooled features V[512]
// -> global scratch), then the FC tail of ALL of them as one batch (fc_batch_parse: weights streamed once per 16
// proposals), whose regressed matches are the next level's proposals (patch2pix.py:259-272).
// WINO (arithmetic P2P_REGRESS_FP16X2W, regress_wino.hip): the launch covers ONE level (args.lvl0) and the proposals
// [args.p0, args.p1); a work-group runs conv1 only and leaves the Winograd-transformed input of conv2 (B^T d B of every
// 2 x 2 output tile, two fp16 planes) in the global buffer wino_gemm_kernel reads; conv2 and the FC tail are other launches.
template <bool WINO>
__global__ __launch_bounds__(NT, 2) void regress_h2_kernel(RegressArgs args) {
    P2P_DYN_SHARED(unsigned char, smb);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x;

    float *raw0 = (float *)(smb + XRAW0);
    float *scale = (float *)(smb + XSM_SCALE);
    float *misc = (float *)(smb + XSM_MISC);
    // scratch (re-derived from the launch arguments where it is used: two more live 64-bit pointers across the convolution
    // phases spill): pooled features V [level][n][512], then the un-truncated matches of the previous level [n][4]
#define XWS_V(lvl_) (args.ws + (size_t)(lvl_) * args.n * 512)
#define XWS_NEXTP() (args.ws + ((2 * (size_t)args.n * 512 + 31) & ~(size_t)31))

#pragma unroll 1
    for (int lvl = WINO ? args.lvl0 : 0; lvl < (WINO ? args.lvl0 + 1 : args.nlevels); ++lvl) {
        const RegDev &R_ = args.reg[lvl];
        // WINO: the gather of proposal i + 1 is issued before the transform + write-out of proposal i (its ~40 scattered loads
        // per thread land in these registers while that phase runs) and committed to LDS at the top of its own iteration
        float gn0[2][2], gn1[2][11], gn2[2][4], gn3[2][3];
        bool pre = false;                    // gn* hold the gather of the proposal the next iteration starts with
        auto gather_loads = [&](int xa_, int ya_, int xb_, int yb_, const ItemDev &J) {
            int tv = wave * 64 + P2P_LANE_ID();
            P2P_OPAQUE(tv);
#pragma unroll
            for (int img = 0; img < 2; ++img) {
                const int Hh = J.H[img], Ww = J.W[img], x0 = img ? xb_ : xa_, y0 = img ? yb_ : ya_;
                {
                    const int r0 = clampi(y0, 0, Hh - 1), c0 = clampi(x0, 0, Ww - 1);
                    const float *src = J.pyr[img][0];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int e = tv + k * NT;
                        const int c = e >> 8, rem = e & 255, r = rem >> 4, cc = rem & 15;
                        gn0[img][k] = (e < 768) ? src[((size_t)c * Hh + min(r0 + r, Hh - 1)) * Ww + min(c0 + cc, Ww - 1)] : 0.f;
                    }
                }
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                    const int Cc = (j == 3) ? 128 : 64;
                    const int nk = (j == 1) ? 11 : (j == 2) ? 4 : 3;
                    const int Hj = Hh >> j, Wj = Ww >> j;
                    const int Ha = level_dim(Hh, j), Wa = level_dim(Ww, j);
                    const int r0 = clampi(y0 >> j, 0, Hj - 1);
                    const int c0 = clampi(x0 >> j, 0, Wj - 1);
                    const float *src = J.pyr[img][j];
#pragma unroll
                    for (int k = 0; k < nk; ++k) {
                        const int e = tv + k * NT;
                        const int c = e / (Rr * Rr);
                        const int rem = e - c * (Rr * Rr);
                        const int r = rem / Rr;
                        const int cc = rem - r * Rr;
                        const float v = (e < Cc * Rr * Rr)
                                            ? src[((size_t)c * Ha + min(r0 + r, Hj - 1)) * Wa + min(c0 + cc, Wj - 1)] : 0.f;
                        if (j == 1) gn1[img][k] = v; else if (j == 2) gn2[img][k] = v; else gn3[img][k] = v;
                    }
                }
            }
        };
        auto gather_commit = [&]() {
            int tv = wave * 64 + P2P_LANE_ID();
            P2P_OPAQUE(tv);
#pragma unroll
            for (int img = 0; img < 2; ++img) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = tv + k * NT;
                    if (e < 768) raw0[img * 768 + e] = gn0[img][k];
                }
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                    const int Cc = (j == 3) ? 128 : 64;
                    const int nk = (j == 1) ? 11 : (j == 2) ? 4 : 3;
#pragma unroll
                    for (int k = 0; k < nk; ++k) {
                        const int e = tv + k * NT;
                        if (e < Cc * Rr * Rr) {
                            const int c = e / (Rr * Rr);
                            const int rem = e - c * (Rr * Rr);
                            const float v = (j == 1) ? gn1[img][k] : (j == 2) ? gn2[img][k] : gn3[img][k];
                            if (j == 1) *(float *)(smb + XSHARED + XTMP1 + img * XTMP1IMG + rem * XTMP1ST + c * 4) = v;
                            else *(float *)(smb + XSHARED + img * XTMPIMG + ((j == 2) ? rem * XTMP2ST : XTMP3 + rem * XTMP3ST) + c * 4) = v;
                        }
                    }
                }
            }
        };
#pragma unroll 1
      for (int cprop = (WINO ? args.p0 : 0) + blockIdx.x; cprop < (WINO ? args.p1 : args.n); cprop += nwg) {
        // WINO: cprop is the compact index of a proposal that exists (its scratch rows), prop its slot; else they coincide
        const int prop = WINO ? wino_slot(args, cprop) : cprop;
        if (WINO && prop < 0) break;
        int it = 0;
        while (it + 1 < args.nitems && prop >= args.start[it + 1]) ++it;
        if (!WINO && args.dev_counts && prop - args.start[it] >= args.dev_counts[it]) continue;      // empty slot (whole work-group)
        const ItemDev &I = args.item[it];
        int tq = tid;                        // WINO: an opaque copy -- the lane's load addresses are otherwise hoisted out of the loop and spilled
        if constexpr (WINO) {
            tq = wave * 64 + P2P_LANE_ID();
            P2P_OPAQUE(tq);
        }
        if (tq < 4) {
            float v;
            if (lvl > 0) v = load_coherent(XWS_NEXTP() + (size_t)prop * 4 + tq);
            else if (args.is_float) v = ((const float *)args.proposals)[(size_t)prop * 4 + tq];
            else v = (float)((const long long *)args.proposals)[(size_t)prop * 4 + tq];
#ifdef XF_SAME_PATCH                    // timing experiment (wrong results): every proposal gathers the same (cache-resident) patch
            v = 100.f + 16.f * tq;
#endif
            misc[8 + tq] = v;
        }
        if constexpr (WINO) {                // the proposal this work-group takes next (its gather is prefetched): coordinates -> misc[4..7]
            const int nprop = (cprop + nwg < args.p1) ? wino_slot(args, cprop + nwg) : -1;
            if (nprop >= 0 && tq >= 4 && tq < 8) {
                const int q = tq - 4;
                float v;
                if (lvl > 0) v = load_coherent(XWS_NEXTP() + (size_t)nprop * 4 + q);
                else if (args.is_float) v = ((const float *)args.proposals)[(size_t)nprop * 4 + q];
                else v = (float)((const long long *)args.proposals)[(size_t)nprop * 4 + q];
                misc[4 + q] = v;
            }
        }
        // per-level reductions behind the power-of-two operand scales: misc[12 + img] = smallest per-pixel L2 scale of the
        // image (float bits, atomic min), misc[14] = largest |H| (float bits, atomic max)
        if (tq >= 64 && tq < 67) ((int *)misc)[12 + tq - 64] = (tq < 66) ? 0x7f7fffff : 0;
        __syncthreads();
        // window origins (x, y) in image 1 / image 2 (networks/utils.py:8-19); scalars + selects, never an indexed array
        int moff = 8;                // opaque: the LDS address of misc is otherwise materialised before the loop and spilled
        P2P_OPAQUE(moff);
        const int xa = (int)misc[moff + 0] - 8, ya = (int)misc[moff + 1] - 8;
        const int xb = (int)misc[moff + 2] - 8, yb = (int)misc[moff + 3] - 8;
#define XX0(img_) ((img_) ? xb : xa)
#define XY0(img_) ((img_) ? yb : ya)
        // opaque copy of the thread id for the staging phases (keeps their lane-only index math inside the level loop)
        int tidv = wave * 64 + P2P_LANE_ID();       // re-derived per level: not even the thread id is kept in a VGPR across it
        P2P_OPAQUE(tidv);
        // lane coordinates derived from the opaque copy: nothing lane-dependent is loop-invariant for the compiler, so
        // nothing is hoisted out of the level loop and kept (or spilled) across its high-pressure phases
        const int half = (tidv >> 5) & 1, l31 = tidv & 31;
        XT_DECL XT_START
        // waves 4-7 are the younger wave of their SIMD and lose the issue arbitration on every MFMA segment (they were
        // ~20 % slower between barriers): static priority for that half, no per-segment flips
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);

        // ------------------------------------------------------------ gather (networks/utils.py:4-36)
        if constexpr (WINO) {
            if (!pre) gather_loads(xa, ya, xb, yb, I);       // the work-group's first proposal: nothing was prefetched
            gather_commit();
        } else {
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
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                    const int Cc = (j == 3) ? 128 : 64;
                    const int nk = (j == 1) ? 11 : (j == 2) ? 4 : 3;
#pragma unroll
                    for (int k = 0; k < nk; ++k) {
                        const int e = tidv + k * NT;
                        if (e < Cc * Rr * Rr) {
                            const int c = e / (Rr * Rr);
                            const int rem = e - c * (Rr * Rr);
                            const float v = (j == 1) ? g1[img][k] : (j == 2) ? g2[img][k] : g3[img][k];
                            if (j == 1) {
                                *(float *)(smb + XSHARED + XTMP1 + img * XTMP1IMG + rem * XTMP1ST + c * 4) = v;
                            } else {
                                // fp32 copy for the scale pass + the planes (exact: v = p0 + p1 + p2)
                                *(float *)(smb + XSHARED + img * XTMPIMG + ((j == 2) ? rem * XTMP2ST : XTMP3 + rem * XTMP3ST) + c * 4) = v;
                            }
                        }
                    }
                }
            }
        }
        if (tidv < 2 * XNPL * (3 * YST2 + 2 * YST3) / 16) {  // the zero areas: the dead rows of the cell tiles multiply zeros
            const int per = (3 * YST2 + 2 * YST3) / 16;
            const int im = tidv / (XNPL * per), pl = (tidv / per) % XNPL, q = tidv % per;
            float zf = 0.f;
            P2P_OPAQUE(zf);
            unsigned char *z = smb + im * XIMG + ((q < 3 * YST2 / 16) ? YOFF2 + pl * YPL2 + YNC2 * YST2 + q * 16
                                                                      : YOFF3 + pl * YPL3 + YNC3 * YST3 + (q - 3 * YST2 / 16) * 16);
            *(f32x4 *)z = (f32x4){zf, zf, zf, zf};
        }
        __syncthreads();
        XT(0)

        // ------------------------------------------------------------ per-pixel L2 scale (patch2pix.py:173-174) + fold table
        float sc_keep;
        int c23_keep;
        {
            const int img = tidv >> 8, pix = tidv & 255, py = pix >> 4, px = pix & 15;
            float ss = 0.f;
            {
                const float *p = raw0 + img * 768 + patch_cell(XY0(img), py, 0, I.H[img]) * 16 + patch_cell(XX0(img), px, 0, I.W[img]);
#pragma unroll
                for (int c = 0; c < 3; ++c) ss = fmaf(p[c * 256], p[c * 256], ss);
            }
            const int c2 = patch_cell(XY0(img), py, 2, I.H[img]) * 5 + patch_cell(XX0(img), px, 2, I.W[img]);
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const int Rr = (j == 1) ? 9 : (j == 2) ? 5 : 3;
                const int Cc = (j == 3) ? 128 : 64;
                const int cjy = patch_cell(XY0(img), py, j, I.H[img]), cjx = patch_cell(XX0(img), px, j, I.W[img]);
                const int cj = cjy * Rr + cjx;
                const unsigned char *p = (j == 1) ? smb + XSHARED + XTMP1 + img * XTMP1IMG + cj * XTMP1ST
                                                  : smb + XSHARED + img * XTMPIMG + ((j == 2) ? cj * XTMP2ST : XTMP3 + cj * XTMP3ST);
                for (int c = 0; c < Cc; c += 4) {
                    const f32x4 v = *(const f32x4 *)(p + c * 4);
                    ss = fmaf(v[0], v[0], ss); ss = fmaf(v[1], v[1], ss); ss = fmaf(v[2], v[2], ss); ss = fmaf(v[3], v[3], ss);
                }
            }
            const float sc = 1.0f / sqrtf(ss + 1e-6f);
            // fp16 operands carry power-of-two scales (header): per-pixel-normalised values x 2^12; the fold table entry
            // needs the image's smallest scale and is written after the barrier
            scale[tidv] = sc * 4096.0f;
            atomicMin((int *)misc + 12 + img, __float_as_int(sc));
            sc_keep = sc;
            c23_keep = c2 * XTROW | (patch_cell(XY0(img), py, 3, I.H[img]) * 3 + patch_cell(XX0(img), px, 3, I.W[img])) * XTROW << 16;
            if (tidv < 2 * 33) {     // ring = the zero padding of conv1: scale 0
                const int im = tidv / 33, q = tidv - im * 33;
                const int idx = (q < 17) ? q : (q - 16) * 17;
                float zf = 0.f;      // opaque, or the constant is hoisted out of the level loop and spilled
                P2P_OPAQUE(zf);
                *(f32x2 *)(smb + XTAB + im * XTABIMG + idx * 8) = (f32x2){zf, zf};
            }
        }
        __syncthreads();
        {   // the image's power-of-two scale 2^e, e = 12 + floor(log2(smallest per-pixel scale)): every cell component
            // times 2^e is <= 2^12 (|c| * scale[p] <= 1 for the pixels p of its cell), the fold multiplies by scale[p] * 2^(12 - e)
            const int img = tidv >> 8, pix = tidv & 255, py = pix >> 4, px = pix & 15;
            const int eb = clampi((((const int *)misc)[12 + img] >> 23) & 0xff, 13, 240);
            *(f32x2 *)(smb + XTAB + img * XTABIMG + ((py + 1) * 17 + px + 1) * 8) =
                (f32x2){sc_keep * __int_as_float((254 - eb) << 23), __int_as_float(c23_keep)};
            // planes of levels 1, 2 and 3 from their fp32 copies.  Level 1: 2 x 81 x 32 channel PAIRS (one 4-byte store per plane)
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const int e = tidv + k * NT;
                if (e < 2 * 2592) {
                    const int im = (e >= 2592), r = e - im * 2592;
                    const int ebi = clampi((((const int *)misc)[12 + im] >> 23) & 0xff, 13, 240);
                    const float mul = __int_as_float((ebi + 12) << 23);
                    const int cell = r >> 5, c2 = (r & 31) * 2, cy = cell / 9, cx = cell - 9 * cy;
                    const f32x2 v = *(const f32x2 *)(smb + XSHARED + XTMP1 + im * XTMP1IMG + cell * XTMP1ST + c2 * 4);
                    const unsigned h = pk_e(v[0] * mul, v[1] * mul);
                    unsigned char *d = smb + im * XIMG + XOFF1 + cy * XRP1 + cx * XST1 + c2 * 2;
                    *(unsigned *)d = h;
                    *(unsigned *)(d + XPL1) = pk_e(v[0] * mul - pk_lo(h), v[1] * mul - pk_hi(h));
                }
            }
            // levels 2 and 3: 2 x (25 x 64 + 9 x 128) values
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const int e = tidv + k * NT;
                if (e < 2 * 2752) {
                    const int im = (e >= 2752), r = e - im * 2752;
                    const int ebi = clampi((((const int *)misc)[12 + im] >> 23) & 0xff, 13, 240);
                    const float mul = __int_as_float((ebi + 12) << 23);
                    const bool l2 = r < 1600;
                    const int cell = l2 ? r >> 6 : (r - 1600) >> 7, c = l2 ? r & 63 : (r - 1600) & 127;
                    const float v = *(const float *)(smb + XSHARED + im * XTMPIMG + (l2 ? cell * XTMP2ST : XTMP3 + cell * XTMP3ST) + c * 4);
                    store_planes(smb + im * XIMG + (l2 ? YOFF2 + cell * YST2 : YOFF3 + cell * YST3) + c * 2, l2 ? YPL2 : YPL3, v * mul);
                }
            }
        }
        __syncthreads();       // the fp32 copy is dead: its region becomes the level-0 block
        XT(1)

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
            *(float *)(smb + XSHARED + m * XA0ST + kk * 4) = v;
        }
        __syncthreads();
        XT(2)

        // ------------------------------------------------------------ conv1: 3x3, stride 2, pad 1
        // zeros from an opaque register: a literal zero gets tied to some long-lived zero of the prologue and spilled
#define XZERO16(A_) { float z_ = 0.f; P2P_OPAQUE(z_); _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) A_[i_] = z_; }
        f32x16 acc00, acc01, acc10, acc11;
        XZERO16(acc00) XZERO16(acc01) XZERO16(acc10) XZERO16(acc11)
        f32x4 B0[XNPL], B1[XNPL], B2[XNPL], B3[XNPL], B4[XNPL], B5[XNPL], B6[XNPL], B7[XNPL], S0[XNPL], S1[XNPL];
        const f32x16 zero16 = {0};
        {
            f32x4 R0[2], R1[2];
            const unsigned char *wb = (const unsigned char *)R_.wh1 + (size_t)wave * (S1_UNITS + XPF) * XUB;
            const unsigned wlane = (tidv & 63) * 16;
            XLOADB(B0, 0) XLOADB(B1, 1) XLOADB(B2, 2) XLOADB(B3, 3) XLOADB(B4, 4) XLOADB(B5, 5)
            {   // level 0 of both images: 4 slabs of the pre-scaled block
                const unsigned char *p0 = smb + XSHARED + l31 * XA0ST + half * 32;
                const unsigned char *p1 = p0 + 32 * XA0ST;
                XPRO(p0, p1, 1.0f)
                XGROUP4(XSLAB(p0 + 64, p1 + 64, 1.0f, 1.0f, B0, B1, B6, B7, 6),
                        XSLAB(p0 + 128, p1 + 128, 1.0f, 1.0f, B2, B3, B0, B1, 8),
                        XSLAB(p0 + 192, p1 + 192, 1.0f, 1.0f, B4, B5, B2, B3, 10),
                        XSLAB(p0, p1, 1.0f, 1.0f, B6, B7, B4, B5, 12))
            }
            __syncthreads();   // the im2col block is dead: its region becomes the fold buffers
            XT(3)
            // this lane's row of the cell tile: level-2 cell l31 (rows >= 25 are never read back) and its level-3 parent
            const int c2y = (l31 < 25) ? l31 / 5 : 0, c2x = (l31 < 25) ? l31 - 5 * (l31 / 5) : 0;
            // T2 (level-2 rows): under the turn protocol shared by wave w and w + 4 (their folds never overlap: G1's is over
            // before e3, G0's runs between e3 and e4), else one per wave; T3 (9 level-3 rows) is private
            // fold buffers of this wave, rows of 64 floats = its 64 output channels with the two n-tiles INTERLEAVED (channel n of
            // the wave at float 2 (n & 31) + (n >> 5)): a lane's pair (n, n + 32) is one 8-byte LDS access (round 6: the fold read
            // them as two 4-byte words, 5 LDS instructions per row and step instead of 3)
            float *Tw = (float *)(smb + XSHARED + (wave & (XT2N - 1)) * XTW) + 2 * l31;
            float *T3w = (float *)(smb + XSHARED + XT2N * XTW + wave * (9 * XTROW));
            const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
            // The K-ranges (tap, image) are walked in 18 steps.  A step = the pixel slabs of level 1 (P), the cell slabs
            // of levels 2 + 3 (C) and the fold (F).  Waves 0-3 run P(i) C(i) F(i); waves 4-7 -- each shares its SIMD with
            // one of waves 0-3 -- run C(i) F(i) P(i) (their weight stream is packed in that order), so that a fold, which
            // issues no MFMA for ~1500 cycles, sits beside the other wave's MFMA-dense cell range instead of beside
            // its fold.  In loop form: iteration it does P(it - stagger) then C(it) F(it).
            // The two halves also take turns on the matrix pipe (see XPP): per step, waves 0-3 run P | - | C | F and
            // waves 4-7 - | C | F | P between the same four barriers, so a fold always sits beside the partner's MFMAs.
            const int stagger = wave >> 2;
#pragma unroll 1
            for (int it = 0; it < 18 + stagger; ++it) {
                const int pi = it - stagger;
                if (pi >= 0) {      // ---- P(pi): level 1 (64 ch), pixel rows of the pre-split cell planes
                    const int tap = pi >> 1, img = pi & 1;
                    const int ky = tap / 3, kx = tap - ky * 3;
                    // this lane's pixel rows (LDS byte offsets per m-tile); a pixel of the padding ring reads some valid cell: its
                    // table entry is zero
                    int ab[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int p = 32 * t + l31;
                        const int pyc = max(2 * (p >> 3) + ky - 1, 0), pxc = max(2 * (p & 7) + kx - 1, 0);
                        ab[t] = img * XIMG + XOFF1 + patch_cell(XY0(img), pyc, 1, I.H[img]) * XRP1 +
                                patch_cell(XX0(img), pxc, 1, I.W[img]) * XST1 + half * 16;
                    }
                    const unsigned char *a0 = smb + ab[0], *a1 = smb + ab[1];
                    const unsigned char *tabq = smb + XTAB + img * XTABIMG + (ky * 17 + kx + 8 * half) * 8;
#ifdef XF_SKIP_P                        // timing experiments (wrong results): XF_SKIP_P / _C / _FOLD / _CONV2 drop one part
                    XWADV(8) (void)a0; (void)a1; (void)tabq;
#else
                    {
                        f32x16 tp0, tp1;
                        XLOADP(S0, a0, XPL1) XLOADP(S1, a1, XPL1)
                        XQSTEP(XQHALFZ, S0, S1, R0, R1, a0 + 32, a1 + 32, B0, B6, 6)
                        XQSTEP(XQHALF, R0, R1, S0, S1, a0 + 64, a1 + 64, B1, B7, 7)
                        XQSTEP(XQHALF, S0, S1, R0, R1, a0 + 96, a1 + 96, B2, B0, 8)
                        XQSTEP(XQHALF, R0, R1, S0, S1, a0, a1, B3, B1, 9)
                        XPFOLD(acc00, acc10)
                        XQSTEP(XQHALFZ, S0, S1, R0, R1, a0 + 32, a1 + 32, B4, B2, 10)
                        XQSTEP(XQHALF, R0, R1, S0, S1, a0 + 64, a1 + 64, B5, B3, 11)
                        XQSTEP(XQHALF, S0, S1, R0, R1, a0 + 96, a1 + 96, B6, B4, 12)
                        XQSTEP(XQHALF, R0, R1, S0, S1, a0 + 96, a1 + 96, B7, B5, 13)
                        XPFOLD(acc01, acc11)
                        XWADV(8)
                    }
#endif
                    XTL(4)
                }
                if (it < 18) {      // ---- C(it), F(it): levels 2 (64 ch) + 3 (128 ch), cell rows, pre-split planes
                    const int tap = it >> 1, img = it & 1;
                    const int ky = tap / 3, kx = tap - ky * 3;
                    // the lane's cell row.  Level-3 cell of level-2 cell c (per axis, absolute indices):
                    // min(c >> 1, dim/8 - 1), see the header; the staged tiles start at the clamped origins.
                    int a2, a3;
                    {
                        const int Hh = I.H[img], Ww = I.W[img];
                        const int by2 = clampi(XY0(img) >> 2, 0, (Hh >> 2) - 1), bx2 = clampi(XX0(img) >> 2, 0, (Ww >> 2) - 1);
                        const int by3 = clampi(XY0(img) >> 3, 0, (Hh >> 3) - 1), bx3 = clampi(XX0(img) >> 3, 0, (Ww >> 3) - 1);
                        const int c3y = clampi(min((by2 + c2y) >> 1, (Hh >> 3) - 1) - by3, 0, 2);
                        const int c3x = clampi(min((bx2 + c2x) >> 1, (Ww >> 3) - 1) - bx3, 0, 2);
                        // dead rows 25-31: the zero area, at the piece of the slot row l31 would have (9 slots per cell)
                        a2 = img * XIMG + YOFF2 + ((l31 < 25) ? (c2y * 5 + c2x) * YST2 + half * 16
                                                              : YNC2 * YST2 + ((((l31 - YNC2) * 9 + half) & 15) << 4));
                        a3 = img * XIMG + YOFF3 + ((l31 < 25) ? c3y * 3 + c3x : YNC3) * YST3 + half * 16;
                    }
                    const unsigned char *q2 = smb + a2, *q3 = smb + a3;
                    f32x16 t0, t1;
#if defined(XF_SKIP_C)
                    t0 = acc00; t1 = acc01; XWADV(24) (void)q2; (void)q3;
#else
                    {
                        (void)q3;
                        // level 3 first: 4 K steps of 32 channels x 4 n-tiles of 16 columns, rows = the 9 cells
                        const int l16 = (tidv & 15), kb = (tidv >> 4) & 3;
                        // dead rows 9-15: the zero area, at the piece of the slot row l16 would have (2 slots per cell, mod 16)
                        const unsigned char *q3r = smb + img * XIMG + YOFF3 + ((l16 < 9) ? l16 * YST3 + kb * 16
                                                                                         : YNC3 * YST3 + (((2 * (l16 - YNC3) + kb) & 15) << 4));
                        f32x4v u0, u1, u2, u3;
                        XLOADP(S0, q3r, YPL3)
                        XGROUP4(X16SLAB(X16HALFZ, u0, u1, S0, XLOADP(S1, q3r + 64, YPL3), B0, B1, B6, B7, 6),
                                X16SLAB(X16HALFZ, u2, u3, S0, , B2, B3, B0, B1, 8),
                                X16SLAB(X16HALF, u0, u1, S1, XLOADP(S0, q3r + 128, YPL3), B4, B5, B2, B3, 10),
                                X16SLAB(X16HALF, u2, u3, S1, , B6, B7, B4, B5, 12))
                        XGROUP4(X16SLAB(X16HALF, u0, u1, S0, XLOADP(S1, q3r + 192, YPL3), B0, B1, B6, B7, 6),
                                X16SLAB(X16HALF, u2, u3, S0, , B2, B3, B0, B1, 8),
                                X16SLAB(X16HALF, u0, u1, S1, XLOADP(S0, q2, YPL2), B4, B5, B2, B3, 10),
                                X16SLAB(X16HALF, u2, u3, S1, , B6, B7, B4, B5, 12))
                        // T3[row = 4 * kb + r][column 16 * nt + l16 -> float 2 (column & 31) + (column >> 5)]: n-tiles 0 / 2 and 1 / 3 pair up
                        P2P_WAVE_SYNC();
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (4 * kb + r < 9) {
                                f32x2 *d = (f32x2 *)(T3w + (4 * kb + r) * 64) + l16;
                                d[0] = (f32x2){u0[r], u2[r]}; d[16] = (f32x2){u1[r], u3[r]};
                            }
                        // level 2: 4 slabs of 16 channels, rows = level-2 cells
                        XGROUP4(XCSLAB(XHALFZ, S0, S1, q2 + 32, YPL2, B0, B1, B6, B7, 6),
                                XCSLAB(XHALF, S1, S0, q2 + 64, YPL2, B2, B3, B0, B1, 8),
                                XCSLAB(XHALF, S0, S1, q2 + 96, YPL2, B4, B5, B2, B3, 10),
                                XCSLAB(XHALF, S1, S0, q2 + 96, YPL2, B6, B7, B4, B5, 12))
                    }
#endif
                    XTL(5)
#ifndef XF_SKIP_FOLD
                    // fold: acc[pixel][n] += scale[pixel] * T[cell row of the pixel][n]
                    P2P_WAVE_SYNC();            // the wave's previous fold has read T
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2);              // + 4 * half
                        if ((r < 12 || half == 0) && row + 4 * half < XTROWS) {
                            *(f32x2 *)(Tw + (row + 4 * half) * 64) = (f32x2){t0[r], t1[r]};
                        }
                    }
                    P2P_WAVE_SYNC();
                    {
                        const unsigned char *tabp = smb + XTAB + img * XTABIMG + (ky * 17 + kx + 8 * half) * 8;
                        const unsigned char *Tr = (const unsigned char *)Tw;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const f32x2 e = *(const f32x2 *)(tabp + ((8 * t + 2 * (r >> 2)) * 17 + 2 * (r & 3)) * 8);
                                const int offs = __float_as_int(e[1]);
                                const f32x2 g = *(const f32x2 *)(Tr + (offs & 0xffff));
                                const f32x2 h3 = *(const f32x2 *)((const unsigned char *)(T3w + 2 * l31) + (offs >> 16));
                                const float v0 = g[0] + h3[0], v1 = g[1] + h3[1];
                                if (t == 0) { acc00[r] = fmaf(e[0], v0, acc00[r]); acc01[r] = fmaf(e[0], v1, acc01[r]); }
                                else        { acc10[r] = fmaf(e[0], v0, acc10[r]); acc11[r] = fmaf(e[0], v1, acc11[r]); }
                                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // four rows in flight, not all 32
                            }
                    }
#else
                    acc00 += t0; acc01 += t1;
#endif
                    XTL(6)
                }
            }
        }
        __syncthreads();   // all waves are done reading the conv1 operands
        XT(7)

        if constexpr (WINO) {
            // BN1 -> H (fp32, scaled per proposal) -> LDS [pixel 64 + one row of zeros][512 ch (+16 B)] -> B^T d B of the 16
            // tiles (4 x 4 windows at stride 2 over the zero-padded 8 x 8 map) -> two fp16 planes -> the A blocks of
            // wino_gemm_kernel: [position 16][row block][K chunk 16][plane 2][row 128][4 pieces of 8 ch, XOR-swizzled].
            constexpr int HWST = 512 * 4 + 16;
            static_assert(65 * HWST <= XSM_MISC, "H as fp32 fits the convolution buffers");
            {
                float mx = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = wave * 64 + u * 32 + l31;
                    const float s = R_.bn1s_h[n], b = R_.bn1b[n];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x16 &a = (t == 0) ? (u == 0 ? acc00 : acc01) : (u == 0 ? acc10 : acc11);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(fmaf(a[r], s, b)));
                    }
                }
                atomicMax((int *)misc + 14, __float_as_int(mx));
            }
            if (tidv < HWST / 16) {
                float zf = 0.f;
                P2P_OPAQUE(zf);
                *(f32x4 *)(smb + 64 * HWST + tidv * 16) = (f32x4){zf, zf, zf, zf};
            }
            __syncthreads();
            // |H| * hmul in [2^10, 2^11): the transform grows a value at most fourfold, |U| < 2^13
            const int eb = clampi((((const int *)misc)[14] >> 23) & 0xff, 20, 250);
            const float hmul = __int_as_float((264 - eb) << 23);
            if (tidv == 0) args.hinv[cprop - args.p0] = __int_as_float((eb - 10) << 23);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R_.bn1s_h[n], b = R_.bn1b[n];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 &a = (t == 0) ? (u == 0 ? acc00 : acc01) : (u == 0 ? acc10 : acc11);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                        *(float *)(smb + p * HWST + n * 4) = fmaf(a[r], s, b) * hmul;
                    }
                }
            }
            __syncthreads();
            XT(8)
            // the next proposal's gather: in flight during the transform and the write-out below (its slot is looked up again:
            // one value less alive across conv1)
            const int nprop2 = (cprop + nwg < args.p1) ? wino_slot(args, cprop + nwg) : -1;
            pre = nprop2 >= 0;
            if (pre) {
                int nit = 0;
                while (nit + 1 < args.nitems && nprop2 >= args.start[nit + 1]) ++nit;
                int mo2 = 4;
                P2P_OPAQUE(mo2);
                gather_loads((int)misc[mo2 + 0] - 8, (int)misc[mo2 + 1] - 8, (int)misc[mo2 + 2] - 8, (int)misc[mo2 + 3] - 8, args.item[nit]);
            } else {
                // (defined on this path too: otherwise the values committed at the top of this iteration count as live across
                // conv1 -- the compiler does not see that `pre == false` makes the next iteration reload them -- 40 registers)
                float zf = 0.f;
                P2P_OPAQUE(zf);
#pragma unroll
                for (int im = 0; im < 2; ++im) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) gn0[im][k] = zf;
#pragma unroll
                    for (int k = 0; k < 11; ++k) gn1[im][k] = zf;
#pragma unroll
                    for (int k = 0; k < 4; ++k) gn2[im][k] = zf;
#pragma unroll
                    for (int k = 0; k < 3; ++k) gn3[im][k] = zf;
                }
            }
            // item = K chunk of 32 channels: lane = (tile, 8 channels), two passes of 4 channels; a lane's 8 values of a
            // (position, plane) are ONE 16-byte store and a wave's store covers the proposal's 16 rows of a block = 1 KiB
            // contiguous (8-byte stores in 64-byte runs cost 2 ms of a 14.5 ms launch in write bandwidth)
            int tve = P2P_LANE_ID();             // a fresh opaque lane id: values derived from the loop's tidv would be kept across conv1
            P2P_OPAQUE(tve);
            const int lq = tve & 63, oct = lq & 3, tile = lq >> 2, ty = tile >> 2, tx = tile & 3;
            const unsigned pl = (unsigned)(cprop - args.p0);
            const unsigned pstride = (unsigned)args.mblocks * (16u * WINO_BLK);
            const unsigned rr = (pl & 7u) * 16u + (unsigned)tile;
            const unsigned inblk = (rr * 4u + ((unsigned)oct ^ ((rr >> 2) & 3u))) * 16u;
            // LDS offset of window element (aa, bb) (outside the map: the zero row); recomputed where it is used -- a table of the 16
            // offsets would be 16 more live registers beside the window, the first pass's planes and the prefetched gather
            auto pxo = [&](int aa, int bb) {
                const int y = 2 * ty + aa - 1, x = 2 * tx + bb - 1;
                return (((unsigned)y < 8u && (unsigned)x < 8u) ? y * 8 + x : 64) * HWST + oct * 32;
            };
#pragma unroll 1
            for (int i2 = 0; i2 < 2; ++i2) {
                const int kc = wave * 2 + i2;
                unsigned char *ub = args.wU + (size_t)(((pl >> 3) * 16u + (unsigned)kc) * (unsigned)WINO_BLK + inblk);
                unsigned keep[16][4];                             // first pass: {h0a, h0b, h1a, h1b} of channels 0-3
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    f32x4 d[4][4];
#pragma unroll
                    for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) d[aa][bb] = *(const f32x4 *)(smb + pxo(aa, bb) + kc * 128 + hh * 16);
                    // B^T d B,  B^T = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]; one row of B^T d at a time
                    // (the whole 4 x 4 intermediate beside the window, the first pass's planes and the prefetched gather of
                    // the next proposal does not fit 256 registers)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        f32x4 tr[4];
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb)
                            tr[bb] = (ii == 0) ? d[0][bb] - d[2][bb] : (ii == 1) ? d[1][bb] + d[2][bb]
                                   : (ii == 2) ? d[2][bb] - d[1][bb] : d[1][bb] - d[3][bb];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const f32x4 v = (jj == 0) ? tr[0] - tr[2] : (jj == 1) ? tr[1] + tr[2]
                                          : (jj == 2) ? tr[2] - tr[1] : tr[1] - tr[3];
                            const unsigned h0a = pk_e(v[0], v[1]), h0b = pk_e(v[2], v[3]);
                            const unsigned h1a = pk_e(v[0] - pk_lo(h0a), v[1] - pk_hi(h0a));
                            const unsigned h1b = pk_e(v[2] - pk_lo(h0b), v[3] - pk_hi(h0b));
                            unsigned *kp = keep[ii * 4 + jj];
                            if (hh == 0) {
                                kp[0] = h0a; kp[1] = h0b; kp[2] = h1a; kp[3] = h1b;
                            } else {
                                unsigned char *dst = ub + (size_t)(ii * 4 + jj) * pstride;
                                *(uint4 *)dst = make_uint4(kp[0], kp[1], h0a, h0b);
                                *(uint4 *)(dst + WINO_BLK / 2) = make_uint4(kp[2], kp[3], h1a, h1b);
                            }
                        }
                    }
                }
            }
            XT(9)
        } else {
        // BN1 -> H.  Two planes: every wave writes the planes of its 64 channels into its chunk (wave >> 1).  Three planes:
        // chunk 0 (channels of waves 0, 1) as planes, the other chunks wait as fp32
        {
            if (tidv < 4 * XNPL * (2 * HST / 16)) {  // the two all-zero padding rows of every plane (of every chunk)
                const int ch = tidv / (XNPL * (2 * HST / 16)), pq = tidv - ch * (XNPL * (2 * HST / 16));
                const int pl = pq / (2 * HST / 16), q = pq - pl * (2 * HST / 16);
                float zf = 0.f;
                P2P_OPAQUE(zf);
                if (HCHUNK != 0 || ch == 0) *(f32x4 *)(smb + ch * HCHUNK + pl * HPL + 64 * HST + q * 16) = (f32x4){zf, zf, zf, zf};
            }
            const int chunk = wave >> 1;
            const int hv = half, lv = l31;
            // H is scaled by the power of two that brings its largest magnitude to [2^12, 2^13) before it is split
            {
                float mx = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = wave * 64 + u * 32 + lv;
                    const float s = R_.bn1s_h[n], b = R_.bn1b[n];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x16 &a = (t == 0) ? (u == 0 ? acc00 : acc01) : (u == 0 ? acc10 : acc11);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(fmaf(a[r], s, b)));
                    }
                }
                atomicMax((int *)misc + 14, __float_as_int(mx));
            }
            __syncthreads();
            const float hmul = __int_as_float((266 - clampi((((const int *)misc)[14] >> 23) & 0xff, 20, 250)) << 23);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + lv;
                const int cc = (wave & 1) * 64 + u * 32 + lv;                // channel inside the chunk
                const float s = R_.bn1s_h[n], b = R_.bn1b[n];
                // channels (cc, cc + 1) sit in adjacent lanes: the even lane stores the pair's first plane, the odd lane its
                // second plane -- one 4-byte store per value instead of two 2-byte ones
                const bool odd = lv & 1;
                unsigned char *dplane = smb + chunk * HCHUNK + (odd ? HPL : 0) + 4 * hv * HST + (cc & ~1) * 2;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 &a = (t == 0) ? (u == 0 ? acc00 : acc01) : (u == 0 ? acc10 : acc11);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = 32 * t + (r & 3) + 8 * (r >> 2);       // + 4 * half
                        const float v = fmaf(a[r], s, b) * hmul;
                        const unsigned short h0 = f2e(v), h1 = f2e(v - e2f(h0));
                        const unsigned mine = odd ? h1 : h0, give = odd ? h0 : h1;   // keep the half of my plane, hand the other to the partner
                        const unsigned got = P2P_SWAP_ADJACENT(give);
                        *(unsigned *)(dplane + p * HST) = odd ? (got | mine << 16) : (mine | got << 16);
                    }
                }
            }
        }

        XT(8)
        // ------------------------------------------------------------ conv2: 3x3, stride 1, pad 1, four K-chunks
        XZERO16(acc00) XZERO16(acc01) XZERO16(acc10) XZERO16(acc11)
        {
            const unsigned char *wb = (const unsigned char *)R_.wh2 + (size_t)wave * (S2_UNITS + XPF) * XUB;
            const unsigned wlane = (tidv & 63) * 16;
            f32x4 A00[XNPL], A01[XNPL], A10[XNPL], A11[XNPL];   // [buffer][m-tile][plane]
            // weights: ring of 8 units = 4 slabs, loaded THREE slabs ahead (a third of the stream misses L2 and comes
            // from the Infinity Cache: ~1 us, more than the ~0.8 us one slab of both waves of the SIMD lasts)
            XLOADB(B0, 0) XLOADB(B1, 1) XLOADB(B2, 2) XLOADB(B3, 3) XLOADB(B4, 4) XLOADB(B5, 5)
#pragma unroll 1
            for (int chunk = 0; chunk < 4; ++chunk) {
                if (chunk == 0) __syncthreads();       // H is complete; the four chunks need no further hand-over
                XT(9)
                // A addresses of a tap: pixel rows of the two m-tiles (outside the 8x8 map: the zero rows, at the 16-byte
                // piece that keeps the lane on the bank slot of its virtual row)
                auto rows = [&](int tap, const unsigned char *&p0, const unsigned char *&p1) {
                    const int ky = tap / 3, kx = tap - ky * 3;
                    const int oy = (l31 >> 3) + ky - 1, ox = (l31 & 7) + kx - 1;
                    const bool okx = (ox >= 0) && (ox < 8);
                    const bool ok0 = okx && (oy >= 0);
                    const bool ok1 = okx && (oy + 4 < 8);
                    const int vrow = oy * 8 + ox;
                    const int zoff = 64 * HST + (((vrow + half) & 15) << 4);
                    p0 = smb + chunk * HCHUNK + (ok0 ? vrow * HST + half * 16 : zoff);
                    p1 = smb + chunk * HCHUNK + (ok1 ? (vrow + 32) * HST + half * 16 : zoff);
                };
                const unsigned char *p0, *p1;
                rows(0, p0, p1);
                XLOADP(A00, p0, HPL) XLOADP(A01, p1, HPL)
                if (wave >= 4) __builtin_amdgcn_s_barrier();        // waves 4-7 take the second turn
#ifdef XF_SKIP_CONV2
                if (args.n < 0)
#endif
#pragma unroll 1
                for (int tap = 0; tap < 9; ++tap) {
                    const unsigned char *n0, *n1;       // first slab of the next tap (of this chunk)
                    rows(min(tap + 1, 8), n0, n1);
                    // 8 slabs of 16 channels = 32 bytes per plane; slab j of a group of four loads the units of slab j + 3
                    XHSLAB(A00, A01, A10, A11, p0 + 32, p1 + 32, B0, B1, B6, B7, 6)
                    XHSLAB(A10, A11, A00, A01, p0 + 64, p1 + 64, B2, B3, B0, B1, 8)
                    XPP()
                    XHSLAB(A00, A01, A10, A11, p0 + 96, p1 + 96, B4, B5, B2, B3, 10)
                    XHSLAB(A10, A11, A00, A01, p0 + 128, p1 + 128, B6, B7, B4, B5, 12)
                    XPP()
                    XWADV(8)
                    XHSLAB(A00, A01, A10, A11, p0 + 160, p1 + 160, B0, B1, B6, B7, 6)
                    XHSLAB(A10, A11, A00, A01, p0 + 192, p1 + 192, B2, B3, B0, B1, 8)
                    XPP()
                    XHSLAB(A00, A01, A10, A11, p0 + 224, p1 + 224, B4, B5, B2, B3, 10)
                    XHSLAB(A10, A11, A00, A01, n0, n1, B6, B7, B4, B5, 12)
                    XPP()
                    XWADV(8)
                    p0 = n0; p1 = n1;
                }
                if (wave < 4) __builtin_amdgcn_s_barrier();         // every wave has executed the same number of barriers
                XT(10)
            }
        }

        // BN2 -> ReLU -> max over the 8x8 outputs (BN before max: its scale may be negative)
        {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = wave * 64 + u * 32 + l31;
                const float s = R_.bn2s_h[n] * __int_as_float((clampi((((const int *)misc)[14] >> 23) & 0xff, 20, 250) - 12) << 23), b = R_.bn2b[n];
                const f32x16 &aa = (u == 0) ? acc00 : acc01;
                const f32x16 &ab2 = (u == 0) ? acc10 : acc11;
                float m = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    m = fmaxf(m, fmaf(aa[r], s, b));
                    m = fmaxf(m, fmaf(ab2[r], s, b));
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                if (half == 0) XWS_V(lvl)[(size_t)prop * 512 + n] = m;       // pooled features: the FC batch of the level reads them back
            }
        }
        }
        __syncthreads();       // every wave is done with the proposal's LDS (the next gather overwrites it)
        XT(11)
#ifdef P2P_X3_TIMING
        // raw[0] doubles as the stamp buffer in timing builds: work-groups < 64 record [group][wave][16] phase lengths of the
        // middle proposal of their share
        if (args.raw[0] && blockIdx.x < 64 && prop / nwg == (args.n / nwg) / 2 && (tidv & 63) == 0 && lvl == 0) {
            float *dbg = args.raw[0] + 5 * args.n + (blockIdx.x * 8 + wave) * 16;
#pragma unroll
            for (int i = 0; i < 13; ++i) dbg[i] = (float)XTL_()[i];
        }
#endif
      }
        // ------------------------------------------------------------ FC tail of all this work-group's proposals
        __builtin_amdgcn_s_setprio(0);
        if constexpr (!WINO) {
        __threadfence();       // the V rows written above are visible to the (cache-bypassing) loads of the batch
        __syncthreads();
#ifndef XF_SKIP_FC                      // timing experiment (wrong results)
        fc_batch_parse(R_, args, lvl, XWS_V(lvl), XWS_NEXTP(), smb, tid);
#endif
        __threadfence();       // ... and the regressed matches to the next level's proposal loads
        __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------------------------------
// host side: the weight streams.  conv1: units 0-7 = level 0 ([4 slabs][n-tile]), then [tap][img][16 slabs][n-tile]
// (waves 4-7: the 12 cell slabs of a step before its 4 pixel slabs);
// conv2: [chunk of 128 input channels][tap][8 slabs][n-tile].  A unit is [plane 2][lane 64][8 fp16]; K of a conv1
// slab as in split_conv1_index.
// --------------------------------------------------------------------------------------------------
// K layout of conv1: slab (16 K) -> (input channel of cat(f1, f2), tap) of the value lane half `half`, element j holds.
// Slabs 0-3 = level 0 of both images (K = img * 32 + tap * 3 + c, 27 real per image), then per (tap, image) 16 slabs:
// 4 of level 1 (64 ch), 4 of level 2 (64 ch), 8 of level 3 (128 ch).
static void split_conv1_index(int slab, int half, int j, int &ch, int &tap) {
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

static uint16_t host_e(float v) {
    return __builtin_bit_cast(uint16_t, (_Float16)v);
}
static float host_e2f(uint16_t e) {
    return (float)__builtin_bit_cast(_Float16, e);
}
static void putn(uint16_t *d, size_t unit_base, int lane, int j, float v) {
    const uint16_t p0 = host_e(v);
    const float r1 = v - host_e2f(p0);
    const uint16_t p1 = host_e(r1);
    d[(unit_base + lane) * 8 + j] = p0;
    d[(unit_base + 64 + lane) * 8 + j] = p1;
}

// fp16 planes: every output channel's weights are scaled by the power of two 2^t[n] that brings the largest of them into
// [2^11, 2^12) (exact; undone in the folded BatchNorm scale).
static void channel_exponents(const float *w, int rows, int per_row, int *t) {
    for (int n = 0; n < rows; ++n) {
        t[n] = 0;
        float mx = 0.f;
        for (int k = 0; k < per_row; ++k) mx = std::max(mx, std::fabs(w[(size_t)n * per_row + k]));
        if (mx > 0.f && std::isfinite(mx)) {
            int e;
            std::frexp(mx, &e);          // mx = m * 2^e, m in [0.5, 1)
            t[n] = 12 - e;
        }
    }
}

void pack_h2_weights(const float *conv1_w, const float *conv2_w, float *wx1, float *wx2, int *t1, int *t2) {
    // conv2_w == nullptr: conv1's stream only (P2P_REGRESS_FP16X2W runs conv2 from the Winograd blocks of regress_wino.hip)
    uint16_t *d1 = (uint16_t *)wx1, *d2 = (uint16_t *)wx2;
    channel_exponents(conv1_w, 512, 518 * 9, t1);
    if (conv2_w) channel_exponents(conv2_w, 512, 512 * 9, t2);
    auto W1 = [&](int n, int ch, int tap) { return std::ldexp(conv1_w[((size_t)n * 518 + ch) * 9 + tap], t1[n]); };
    auto W2 = [&](int n, int ch, int tap) { return std::ldexp(conv2_w[((size_t)n * 512 + ch) * 9 + tap], t2[n]); };
    for (int w = 0; w < 8; ++w)
        for (int pos = 0; pos < S1_SLABS; ++pos) {
            // stream position -> canonical slab (split_conv1_index): waves 4-7 walk every (tap, image) step as
            // [12 cell slabs of levels 2 + 3][4 pixel slabs of level 1], waves 0-3 the other way round
            int slab = pos;
            if (w >= 4 && pos >= 4) {
                const int step = (pos - 4) / 16, j = (pos - 4) % 16;
                slab = 4 + step * 16 + ((j < 12) ? 4 + j : j - 12);
            }
            // inside a step the cell range is walked as [level 3: 8 pseudo-slabs in 16x16x32 order][level 2: 4 slabs]
            if (slab >= 4) {
                const int step = (slab - 4) / 16, sc = (slab - 4) % 16;       // canonical: 0-3 level 1, 4-7 level 2, 8-15 level 3
                const int rel = (w >= 4) ? (pos - 4) % 16 : (pos - 4) % 16 - 4;   // position inside the cell range (waves 0-3: after P)
                if (sc >= 4) {
                    if (rel < 8) {          // pseudo-slab rel: K step rel / 2, n-tiles 2 * (rel & 1) + u
                        const int ks = rel / 2, img = step & 1, tap = step >> 1;
                        for (int u = 0; u < 2; ++u) {
                            const size_t base = ((size_t)w * (S1_UNITS + XPF) + pos * 2 + u) * (XNPL * 64);
                            const int nt = 2 * (rel & 1) + u;
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int n = 64 * w + 16 * nt + (lane & 15);
                                    const int ch = img * 259 + 131 + 32 * ks + 8 * (lane >> 4) + j;
                                    putn(d1, base, lane, j, W1(n, ch, tap));
                                }
                        }
                        continue;
                    }
                    slab = 4 + step * 16 + 4 + (rel - 8);       // level-2 slab rel - 8
                }
            }
            // level 1 (canonical slabs 0-3 of a step): consumed one n-tile at a time -- unit 4 u + s of the range's 8 units is
            // (n-tile u, slab s); everything else: units 2 pos + u
            const bool lvl1 = slab >= 4 && (slab - 4) % 16 < 4;
            const int s_in = lvl1 ? (slab - 4) % 16 : 0;
            for (int u = 0; u < 2; ++u) {
                const int unit = lvl1 ? (pos - s_in) * 2 + 4 * u + s_in : pos * 2 + u;
                const size_t base = ((size_t)w * (S1_UNITS + XPF) + unit) * (XNPL * 64);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        int ch, tap;
                        split_conv1_index(slab, lane >> 5, j, ch, tap);
                        putn(d1, base, lane, j, (ch < 0) ? 0.f : W1(n, ch, tap));
                    }
            }
        }
    for (int w = 0; conv2_w && w < 8; ++w)
        for (int slab = 0; slab < S2_SLABS; ++slab) {
            const int chunk = slab / 72, tap = (slab % 72) / 8, sin = slab % 8;
            for (int u = 0; u < 2; ++u) {
                const int unit = slab * 2 + u;
                const size_t base = ((size_t)w * (S2_UNITS + XPF) + unit) * (XNPL * 64);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 64 * w + 32 * u + (lane & 31);
                        const int ch = chunk * 128 + sin * 16 + 8 * (lane >> 5) + j;
                        putn(d2, base, lane, j, W2(n, ch, tap));
                    }
            }
        }
}

static int launch_h2(const RegressArgs &a, int n, bool wino, hipStream_t stream) {
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static DeviceOnce attr_set;
    static std::atomic<int> cus[64];
    int ncu_dev = (dev >= 0 && dev < 64 && attr_set.done(dev)) ? cus[dev].load(std::memory_order_relaxed) : 0;
    if (ncu_dev <= 0) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)regress_h2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)XSM_BYTES));
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)regress_h2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)XSM_BYTES));
        P2P_HIP_CHECK(hipDeviceGetAttribute(&ncu_dev, hipDeviceAttributeMultiprocessorCount, dev));
        if (dev >= 0 && dev < 64) { cus[dev].store(ncu_dev, std::memory_order_relaxed); attr_set.set(dev); }
    }
    // persistent work-groups: one fits a compute unit (LDS), each walks its share of the proposals
#ifdef XF_GRID_CAP                       // power experiment: only this many work-groups (= busy compute units)
    const int ncu = std::min(ncu_dev, XF_GRID_CAP);
#else
    const int ncu = ncu_dev;
#endif
    P2P_REQUIRE(a.ws, P2P_EINVAL, "%s: the scratch buffer is missing", "regress_h2_kernel");
    if (wino) hipLaunchKernelGGL(regress_h2_kernel<true>, dim3(std::min(n, std::max(ncu, 1))), dim3(NT), XSM_BYTES, stream, a);
    else hipLaunchKernelGGL(regress_h2_kernel<false>, dim3(std::min(n, std::max(ncu, 1))), dim3(NT), XSM_BYTES, stream, a);
    return check_launch("regress_h2_kernel");
}

int launch_regress_h2(const RegressArgs &a, int n, hipStream_t stream) { return launch_h2(a, n, false, stream); }
// conv1 of the proposals [a.p0, a.p1) of level a.lvl0 -> the transformed conv2 input (regress_wino.hip); n = a.p1 - a.p0
int launch_regress_h2_conv1(const RegressArgs &a, int n, hipStream_t stream) { return launch_h2(a, n, true, stream); }

}  // namespace p2p
