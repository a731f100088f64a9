// Fine stage, fp32-equivalent arithmetic on the fp16 matrix cores in THREE products per fp32 product
// (mode P2P_REGRESS_FP16X2).
//
// Same kernel body as regress_x3.hip (regress_xn_impl.h).  Every fp32 operand x of the two convolutions is represented as
// the sum of two fp16 numbers
//     x * 2^s = h0 + h1,   h0 = fp16(x * 2^s), h1 = fp16(x * 2^s - h0)          (2 x 11 = 22 bits + the sign of h1)
// whose error is <= 2^-24 |x| (two half-ulp roundings of 11-bit significands) PROVIDED neither plane leaves the normal
// range of fp16 (2^-14 ... 65504).  That is what the power-of-two scales 2^s are for -- they are exact, and every one of
// them is undone exactly by a later multiplication:
//   * activations of conv1: the per-pixel L2-normalised patch values (|v| <= 1) times 2^12; the de-duplicated cell rows
//     of levels 2 and 3 times 2^e of their image, e = 12 + floor(log2(smallest per-pixel scale)), so that every component
//     is <= 2^12 -- the fold multiplies by scale[pixel] * 2^(12 - e) instead of scale[pixel];
//   * weights: per output channel the power of two that brings the largest weight to [2^11, 2^12) (at pack time; folded
//     into the BatchNorm scale that follows);
//   * H = BN1(conv1), the input of conv2: per proposal the power of two that brings max |H| to [2^12, 2^13) (a
//     work-group reduction on the accumulators); BN2's scale is multiplied by its inverse.
// Elements more than 2^17 below the largest of their tensor lose relative precision (their second plane becomes an fp16
// subnormal, absolute error 2^-25 on the scaled value = 2^-37 of the largest), which is far below the rounding of the
// fp32 accumulation.  A product is
//     a * b ~= a0*b0 + (a0*b1 + a1*b0)        (the dropped a1*b1 is <= 2^-24 |a*b|),   fp32 accumulation
// = 3 v_mfma_f32_32x32x16_f16 instead of the 6 bf16 MFMAs of the bf16x3 mode: half the matrix-core work per proposal
// (the kernel sits at the package power limit, so that is what counts), two thirds of the weight stream and of the LDS
// operand reads.  Measured against an fp64 evaluation it is as accurate as the exact-f32 MFMA kernel
// (tools/margin_sweep.py); ceiling 2500 / 3 = 833 TFLOP/s of algorithmic fp32 work.
#define XNPL 2
#define XN_FP16 1
#define XN_KERNEL regress_h2_kernel
#define XN_LAUNCH launch_regress_h2
#define XN_PACK pack_h2_weights
#define XN_NAME "regress_h2_kernel"
#define XN_W1 wh1
#define XN_W2 wh2
#define XN_BN1S bn1s_h
#define XN_BN2S bn2s_h
#include "regress_xn_impl.h"
