// Fine stage, fp32-equivalent arithmetic on the bf16 matrix cores (mode P2P_REGRESS_BF16X3, the default).
//
// Same decomposition as regress.hip / regress_split.hip (one workgroup per proposal, 8 waves x 64 output channels,
// patch de-duplicated in LDS, weights streamed from L2 in consumption order).  Every fp32 operand x of the two
// convolutions is represented EXACTLY as the sum of three bf16 numbers
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (3 x 8 = 24 significant bits)
// and a product is the six v_mfma_f32_32x32x16_bf16 whose order in 2^-8 is <= 2:
//     a*b ~= a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0),      fp32 accumulation.
// The dropped terms (a1*b2 + a2*b1 + a2*b2) are <= 2^-23 |a*b|, i.e. at the level of the rounding of one fp32
// product; measured against an fp64 evaluation this mode is as accurate as the exact-f32 MFMA kernel
// (tools/split_check.py, tests/test_gpu_parity.py).  Cost: 6 bf16 MFMA passes per unit of K instead of 32 fp32
// ones -> 2.65x the fp32-MFMA ceiling.
//
// The kernel runs at the package power limit (measured: 1.32 kW, shader clock ~1.95 GHz; with all-zero weights the same
// instruction stream runs at 2.4 GHz), so it is organised to issue as little as possible -- and so that the two waves
// of a SIMD never issue MFMAs at the same time (profiles/r02_ablation_log.txt):
//   * conv1, levels 2 and 3 of the patch (192 of 259 channels).  They are nearest-neighbour up-samplings (4x4 /
//     8x8 pixels per cell) and a level-3 cell is determined by the level-2 cell, so the 64 output pixels of a
//     tap touch <= 25 distinct (level-2 cell, level-3 parent) rows.  Their K-range is therefore multiplied ONCE
//     PER CELL ROW (one m-tile of 32 rows instead of two m-tiles of pixels: half the MFMAs) from tiles that
//     were split into the three bf16 planes ONCE, when they were gathered; the per-pixel L2 scale and the
//     cell -> pixel expansion are applied when the partial sum T[cell row][n] is folded into the accumulators
//     (acc[pixel][n] += scale[pixel] * T[row(pixel)][n], through LDS fold buffers): -18.6 % MFMAs.  Level 3 (128 of
//     those channels) has only 3 x 3 cells per image: it runs on 16-row tiles (v_mfma_f32_16x16x32_bf16, rows = the 9
//     cells) into its own fold buffer T3, level 2 on a 32-row tile into T2: -24.8 % MFMAs in total.  (T2 is shared by
//     the two waves of a SIMD pair -- their folds never overlap under the turn protocol -- which is what pays for T3.)
//   * conv1, levels 0 and 1 (no de-duplication possible at stride 2): fp32 in LDS, scaled and split in registers
//     (40 VALU operations per 8 values), software-pipelined against the MFMAs of the other m-tile.
//   * conv2: its input H = BN1(conv1) is split into the three planes ONCE per proposal (not once per tap and
//     wave = 72 times).  Three planes of H are 196 KB, so conv2 runs in four K-chunks of 128 input channels: the
//     planes of one chunk (52 KB) are in LDS, the other chunks wait as fp32 (3 x 33 KB) and are converted by the
//     whole work-group between chunks.  The conv2 loop has no VALU work at all.
//   * weights are pre-split at load time: unit = (slab of 16 K, n-tile) = 3 planes x 1 KiB per wave, stream order =
//     consumption order per wave, addressed as scalar base + 16 * lane (one VGPR for the whole stream), prefetched one
//     slab ahead through a ring of four register buffers in conv1 and three slabs ahead through eight in conv2.
//   * the halves of the work-group (waves 0-3 / 4-7; wave w shares its SIMD with wave w + 4) take turns on the matrix
//     pipe between bare s_barriers: two MFMA-dense waves on one SIMD got 57 % of the pipe, one wave alone 85 %.
#define XNPL 3
#define XN_FP16 0
#define XN_KERNEL regress_x3_kernel
#define XN_LAUNCH launch_regress_x3
#define XN_PACK pack_x3_weights
#define XN_NAME "regress_x3_kernel"
#define XN_W1 wx1
#define XN_W2 wx2
#define XN_BN1S bn1s
#define XN_BN2S bn2s
#include "regress_xn_impl.h"
